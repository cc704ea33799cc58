// Fused 1x1 expand + depthwise kxk: the first two thirds of an inverted-residual block in one launch.
//
// reference ops replaced: InvertedResidual's expand ConvBNActivation followed by the depthwise ConvBNActivation
//   (mobilenetv3.py:72-84), _extra_block's 1x1 + depthwise (ssd_mobilenetv3.py:39-50); BN folded at plan time.
//
// Why: the expanded activation is the largest tensor of every block (3-6x the block's input) and the unfused path moves it
// through HBM twice (written by the 1x1, read back by the depthwise). Here it only ever exists as fp16 tiles in LDS.
//
// One 512-thread workgroup = one OHxOW output tile of one image and a run of 64-channel chunks of the expanded tensor
// (grid.z splits the chunks so that small maps still fill the chip; the input region is tiny and re-staged per workgroup):
//   1. the input halo region (IH x IW pixels x cin, NHWC fp16) is staged in LDS once;
//   2. per chunk: E[pixel][64] = act1(X[pixel][:] . W1[chunk][:] + b1) on the matrix cores (v_mfma_f32_32x32x16_f16, weights as
//      the A operand straight from L2 into registers, pixels as the B operand from LDS), rounded to fp16 into LDS exactly
//      like the unfused path rounds it into HBM; pixels outside the image become 0 (the depthwise zero-pads E, not X);
//   3. per chunk: depthwise kxk over the LDS tile (16-byte reads, fp32 accumulate), act2, fp16 NHWC store -- 128 contiguous
//      bytes per pixel and chunk -- plus optional per-tile channel sums for the squeeze-excitation pool (fixed order).
// The halo recompute of step 2 (1.1x .. 1.9x) rides on MFMA throughput the path does not otherwise use.
#include <stdlib.h>
#include <algorithm>

#include "common.h"

#ifdef DN_DEV_STAMPS
static long long* g_xd_stamps = nullptr;     // dev build only (tools/probe_expdw.py): per-workgroup phase stamps
extern "C" __attribute__((visibility("default"))) void dn_debug_expdw_stamps(void* dev_ptr) { g_xd_stamps = (long long*)dev_ptr; }
#define XD_STAMP(k) do { if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 16 + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
// dev build only (python -m demonet_amd.build --stamps): finer stamps inside the SECOND chunk of the run (the steady state), slots 5.. A stamp is a
// global store: a phase that contains an `s_waitcnt vmcnt(0)` also waits for the previous stamp's acknowledgement -- read the LDS-only phases
// (depthwise) at face value and the others as upper bounds.
#define XD_STAMP2(k) do { if (a.stamps && threadIdx.x == 0 && c0 == c_begin + 64) a.stamps[(size_t)blockIdx.x * 16 + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
constexpr long long* g_xd_stamps = nullptr;
#define XD_STAMP(k) do { } while (0)
#define XD_STAMP2(k) do { } while (0)
#endif

namespace {

constexpr int EW = 72;          // halfs per E row in LDS: 64 channels + 8 pad (144 B = 9 x 16 B). The pad is real storage: a LAST chunk of
                                // exactly 72 channels (the 24 -> 72 -> 24 block) is taken whole, columns 64..71 holding its last 8 channels
constexpr int XKS = 8;          // max full 16-deep K steps of the expand (cin <= 128)
constexpr int NT = 512;         // threads per workgroup: 8 waves; two workgroups per CU give 4 waves per SIMD to hide LDS/L2 latency

template <int K, int S, int OH, int OW>
struct ExpDwGeom {
    static constexpr int IH = (OH - 1) * S + K, IW = (OW - 1) * S + K, NPIX = IH * IW;
    static constexpr int RT = (NPIX + 31) / 32, ROWS = RT * 32;
};

// one uniform switch per 16 / 8 values instead of a select chain per element (these kernels are VALU-issue bound: tools/valu.sh)
template <typename V, int N>
__device__ __forceinline__ void act_n(V& v, int act) {
    if (act == DN_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = dn_relu(v[e]);
    } else if (act == DN_ACT_RELU6) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = dn_relu6(v[e]);
    } else if (act == DN_ACT_HSWISH) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = v[e] * dn_relu6(v[e] + 3.f) * (1.f / 6.f);
    }
}

// Instruction diet (the kernel is VALU-issue bound, tools/valu.sh: a wave64 VALU instruction holds its SIMD for 4 cycles):
//   * 3-D grid [8 XCD groups x tiles_x][tiles_y (x chunk split)][image slot]: no integer division to find the tile;
//   * XW8 > 0: the X row width is a compile-time constant (XW8 16-byte chunks), so staging divides by constants only;
//   * the expand's bias rides in the reduction: column `cin` of every X row holds 1.0 for pixels inside the image (0 outside, 0 in
//     the row padding) and the weight fragment holds (hi, lo) = the fp32 bias split into two fp16 values there
//     (|b - hi - lo| <= 2^-22 |b|). A pixel outside the image then expands to act(0) = 0 -- exactly the zero padding the
//     depthwise needs -- without a per-element inside test, bias add or select;
//   * activations: one uniform switch per accumulator (act_n);
//   * every global load is unconditional with a clamped address (a predicated load costs a branch and a conservative wait).
// ONE (round 3): the workgroup's run is exactly one chunk (cexp == 72 taken whole by the wide variant): the chunk loop is not a loop, so the
// next-chunk operands and the loop-carried liveness of the expand fragments go away -- the 72-channel block spilled 64 B per lane with them.
template <int K, int S, int OH, int OW, int KSM, bool EXP, bool PROJ, int XW8, bool WD, bool ONE = false>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(KSM <= 6 ? 4 : 2, ONE ? 8 : 4))) void expdw_kernel(ExpDwArgs a, int tiles_x, int tiles_y, int zsplit) {
    using G = ExpDwGeom<K, S, OH, OW>;
    constexpr int IW = G::IW, NPIX = G::NPIX, RT = G::RT, ROWS = G::ROWS;
    constexpr bool WIDE = WD && PROJ && EXP && KSM <= 2;    // may take a last chunk of 72 channels whole (WD: cexp % 64 == 8 -- the extra
                                                            // fragments cost 32 registers, which a 64-channel block would only spill)
    static_assert(K * K * 9 <= NT, "one 16-byte piece of the chunk's depthwise weights per thread");
    extern __shared__ __attribute__((aligned(16))) half_t lds[];
    const int XW = XW8 > 0 ? XW8 * 8 : a.xw;                // halfs per X row: round16(cin) + 8
    half_t* Xs = lds;                                       // [ROWS][XW] + 8 zero halfs (the last K step of the last row reads past its end)
    half_t* Es = Xs + ROWS * XW + 8;                        // [ROWS][EW]
    half_t* Wd = Es + (EXP ? ROWS * EW : 0);                // [K*K][72]   depthwise weights of the chunk
    float* Bd = reinterpret_cast<float*>(Wd + K * K * EW);  // [72]        depthwise bias of the chunk
    float* Ps = Bd + EW;                                    // [NT/64][64] pooled-sum scratch (one row per wave)
    half_t* Ds = reinterpret_cast<half_t*>(Ps + NT / 64 * 64);   // [DROWS][EW] depthwise output of the chunk (PROJ only)
    constexpr int DROWS = (OH * OW + 31) / 32 * 32;
    float* B3s = reinterpret_cast<float*>(Ds + (PROJ ? DROWS * EW : 0));   // [256] project bias (PROJ only)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    int n, tx, ty, zblk = 0;
    if (a.xq > 0) {                                         // XCD grouping (common.h): blockIdx.x % 8 = the group of a.xq images
        n = (blockIdx.x & 7) * a.xq + blockIdx.z;
        tx = blockIdx.x >> 3;
    } else {
        n = blockIdx.z;
        tx = blockIdx.x;
    }
    if (n >= a.n) return;
    ty = blockIdx.y;
    if (zsplit > 1) { zblk = ty / tiles_y; ty -= zblk * tiles_y; }
    const int tile = ty * tiles_x + tx, tiles = tiles_x * tiles_y;
    const int oy0 = ty * OH, ox0 = tx * OW;
    const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;
    const int cin = a.cin, cexp = a.cexp;
    const int KSF = cin >> 4;                               // full K steps; then ONE last step: the K tail (cin % 16 == 8) and the bias columns
    const int c_begin = zblk * a.chunks_per_wg * 64;
    const int c_end = min(cexp, c_begin + a.chunks_per_wg * 64);

    // Per-chunk operands from global memory, requested one phase ahead of their use: this wave's A fragments of the expand
    // (channel tile t = wave & 1; with a 72-wide chunk also those of the third tile) and this thread's 16-byte piece of the
    // chunk's depthwise weights / bias (written to LDS at the top of the chunk). Unconditional loads, clamped addresses.
    const int t = wave & 1;
    half8 wf[KSM], wl, wf2[WIDE ? KSM : 1], wl2;
    float b1v = 0.f, b1v2 = 0.f;
    uint4 wdreg = make_uint4(0, 0, 0, 0);
    float bdreg = 0.f;
    const int kb = KSF * 16 + hh * 8;                       // this lane's columns of the last step
    const int kcl = min(kb, cin - 8);
    auto request_chunk = [&](int c0, bool wide) {
        if constexpr (EXP) {
            const int ch = min(c0 + t * 32 + r, cexp - 1);
            const half_t* wrow = a.w1 + (size_t)ch * cin;
#pragma unroll
            for (int ks = 0; ks < KSM; ++ks)
                if (ks < KSF) wf[ks] = *reinterpret_cast<const half8*>(wrow + ks * 16 + hh * 8);
            wl = *reinterpret_cast<const half8*>(wrow + kcl);
            b1v = a.b1[ch];
            if constexpr (WIDE) {
                if (wide) {
                    const int ch2 = min(c0 + 64 + r, cexp - 1);
                    const half_t* wrow2 = a.w1 + (size_t)ch2 * cin;
#pragma unroll
                    for (int ks = 0; ks < KSM; ++ks)
                        if (ks < KSF) wf2[ks] = *reinterpret_cast<const half8*>(wrow2 + ks * 16 + hh * 8);
                    wl2 = *reinterpret_cast<const half8*>(wrow2 + kcl);
                    b1v2 = a.b1[ch2];
                }
            }
        }
        const int wrow_i = tid / 9, wc8 = tid - wrow_i * 9;
        wdreg = *reinterpret_cast<const uint4*>(a.wd + (size_t)min(wrow_i, K * K - 1) * cexp + min(c0 + wc8 * 8, cexp - 8));
        bdreg = a.bd[min(c0 + min(tid, EW - 1), cexp - 1)];
    };
    // last-step fragment: data columns as loaded, the bias pair at column cin, zero beyond
    auto last_frag = [&](const half8& w, float b) {
        const half_t hi = (half_t)b;
        const half_t lo = (half_t)(b - (float)hi);
        const half8 bw = {hi, lo, 0, 0, 0, 0, 0, 0};
        const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        return kb < cin ? w : (kb == cin ? bw : zero8);
    };
    half8 w3f[PROJ ? (WIDE ? 5 : 4) : 1];

    XD_STAMP(0);
    // ---- 1. input region -> LDS (zeros outside the image, in the K padding and in the row padding; 1.0 in column cin inside the image)
    {
        const int c8 = cin >> 3, xc8 = XW >> 3;
        const half_t* xin = a.x + (size_t)n * a.H * a.W * cin;
        if (tid == 0) *reinterpret_cast<uint4*>(&Xs[ROWS * XW]) = make_uint4(0, 0, 0, 0);
        for (int i0 = tid; i0 == tid || i0 < ROWS * xc8; i0 += NT * 4) {     // (every thread runs the first pass: it requests the chunk operands)
            uint4 v[4];
            int dst[4];
            bool ok[4], one[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = i0 + NT * u;
                const int pix = idx / xc8, q = idx - pix * xc8;
                const int py = pix / IW, px = pix - py * IW;
                const int gy = iy0 + py, gx = ix0 + px;
                const bool inside = idx < ROWS * xc8 && pix < NPIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                dst[u] = idx < ROWS * xc8 ? pix * XW + q * 8 : -1;
                ok[u] = inside && q < c8;
                one[u] = EXP && inside && q == c8;
                v[u] = *reinterpret_cast<const uint4*>(xin + (ok[u] ? ((size_t)gy * a.W + gx) * cin + q * 8 : (size_t)0));
            }
            float b3reg = 0.f;
            if (i0 == tid) {
                request_chunk(c_begin, WIDE && c_end - c_begin == EW);       // behind the first batch of region loads (memory returns in order)
                if constexpr (PROJ) b3reg = a.b3[min(tid & 255, a.cout - 1)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (dst[u] >= 0) *reinterpret_cast<uint4*>(&Xs[dst[u]]) = ok[u] ? v[u] : make_uint4(one[u] ? 0x3c003c00u : 0u, 0, 0, 0);
            if constexpr (PROJ) { if (i0 == tid && tid < 256) B3s[tid] = tid < a.cout ? b3reg : 0.f; }
        }
    }
    __syncthreads();
    XD_STAMP(1);

    // project stage (PROJ): wave = one (32-pixel row tile, 32-channel tile) unit of the [OH*OW][cout] output; its accumulator
    // lives across the chunk loop (the projection sums over all expanded channels)
    constexpr int PRT = DROWS / 32;
    const int pct = PROJ ? (a.cout + 31) >> 5 : 0;
    const int prt = wave % PRT, pc = wave / PRT;          // launcher guarantees PRT * pct <= NT / 64
    const bool punit = PROJ && pc < pct;
    floatx16 pacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) pacc[e] = 0.f;
    for (int c0 = c_begin; c0 < c_end;) {
        const bool wide = WIDE && c_end - c0 == EW;         // the last 72 channels in one go
        const int cw = wide ? EW : 64, ng = cw >> 3;
        {
            XD_STAMP2(5);
            const int wrow_i = tid / 9, wc8 = tid - wrow_i * 9;
            const bool wok = tid < K * K * 9 && wc8 < ng && c0 + wc8 * 8 < cexp;
            if (tid < K * K * 9) *reinterpret_cast<uint4*>(&Wd[wrow_i * EW + wc8 * 8]) = wok ? wdreg : make_uint4(0, 0, 0, 0);
            if (tid < EW) Bd[tid] = (tid < cw && c0 + tid < cexp) ? bdreg : 0.f;
        }
        XD_STAMP2(6);
        // ---- 2. expand on the matrix cores. A = weight rows (channels), B = pixel rows: the accumulator then holds, per lane,
        //         pixel (lane & 31) and channels 8g + 4*(lane >> 5) .. +3 in registers 4g .. 4g+3.
        //         wave w: channel tile t = w & 1 of the chunk, row tiles (w >> 1), (w >> 1) + 4, ...
        if constexpr (EXP) {
            const half8 wlast = last_frag(wl, b1v);
            for (int rt = wave >> 1; rt < RT && c0 + t * 32 < cexp; rt += NT / 128) {
                floatx16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
                const half_t* xrow = &Xs[(rt * 32 + r) * XW + hh * 8];
#pragma unroll
                for (int ks = 0; ks < KSM; ++ks)
                    if (ks < KSF) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks], *reinterpret_cast<const half8*>(xrow + ks * 16), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlast, *reinterpret_cast<const half8*>(xrow + KSF * 16), acc, 0, 0, 0);
                asm volatile("s_nop 7\n\ts_nop 3" : "+v"(acc));      /* matrix result -> activation switch: every branch path gets its 12 wait states (pwdirect.hip act16) */
                act_n<floatx16, 16>(acc, a.act1);
                half_t* erow = &Es[(rt * 32 + r) * EW + t * 32 + 4 * hh];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    half4 hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) hv[e] = (half_t)acc[4 * g + e];
                    *reinterpret_cast<half4*>(erow + 8 * g) = hv;
                }
            }
            if constexpr (WIDE) {
                if (wide) {                                 // third channel tile: only its first 8 channels exist (columns 64..71)
                    const half8 wlast2 = last_frag(wl2, b1v2);
                    for (int rt = wave; rt < RT; rt += NT / 64) {
                        floatx16 acc;
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
                        const half_t* xrow = &Xs[(rt * 32 + r) * XW + hh * 8];
#pragma unroll
                        for (int ks = 0; ks < KSM; ++ks)
                            if (ks < KSF) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf2[ks], *reinterpret_cast<const half8*>(xrow + ks * 16), acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlast2, *reinterpret_cast<const half8*>(xrow + KSF * 16), acc, 0, 0, 0);
                        asm volatile("s_nop 7\n\ts_nop 3" : "+v"(acc));
                        float v4[4] = {acc[0], acc[1], acc[2], acc[3]};
                        act_n<float[4], 4>(v4, a.act1);
                        half4 hv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) hv[e] = (half_t)v4[e];
                        *reinterpret_cast<half4*>(&Es[(rt * 32 + r) * EW + 64 + 4 * hh]) = hv;
                    }
                }
            }
        }
        XD_STAMP2(7);
        const int cnext = c0 + cw;
        if constexpr (!ONE) { if (cnext < c_end) request_chunk(cnext, WIDE && c_end - cnext == EW); }       // lands under the depthwise stage
        if constexpr (PROJ) {
            const int co = min(pc * 32 + r, a.cout - 1);
#pragma unroll
            for (int ks = 0; ks < (WIDE ? 5 : 4); ++ks)
                w3f[ks] = *reinterpret_cast<const half8*>(a.w3 + (size_t)co * cexp + min(c0 + ks * 16 + hh * 8, cexp - 8));
        }
        __syncthreads();
        if (c0 == c_begin) XD_STAMP(2);
        XD_STAMP2(8);

        // ---- 3. depthwise over the LDS tile: item = (output pixel, 8-channel group)
        float psum[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) psum[e] = 0.f;
        for (int item = tid; item < OH * OW * ng; item += NT) {
            const int opix = (WIDE && wide) ? item / 9 : item >> 3;
            const int cg = item - opix * ng;                 // (8 groups: the same group in every pass -- the pooled sums below rely on it)
            if (c0 + cg * 8 >= cexp) continue;
            const int oy = opix / OW, ox = opix - oy * OW;
            float acc[8];
            {
                const float4 b0 = *reinterpret_cast<const float4*>(&Bd[cg * 8]), b1 = *reinterpret_cast<const float4*>(&Bd[cg * 8 + 4]);
                acc[0] = b0.x; acc[1] = b0.y; acc[2] = b0.z; acc[3] = b0.w; acc[4] = b1.x; acc[5] = b1.y; acc[6] = b1.z; acc[7] = b1.w;
            }
            // without an expand stage the depthwise reads the staged input itself (zero outside the image = its zero padding)
            const int SW = EXP ? EW : XW;
            const half_t* ebase = (EXP ? Es : Xs) + ((oy * S) * IW + ox * S) * SW + cg * 8;
            // one kernel row in flight at a time (fully unrolled, hipcc hoists all K*K tile and weight reads: 350 VGPRs for 5x5)
#pragma unroll 1
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const uint4 ev = *reinterpret_cast<const uint4*>(ebase + (ky * IW + kx) * SW);
                    const uint4 wv = *reinterpret_cast<const uint4*>(&Wd[(ky * K + kx) * EW + cg * 8]);
                    fma_mix_h8(acc, ev, wv);
                }
            const int gy = oy0 + oy, gx = ox0 + ox;
            if (PROJ || (gy < a.Ho && gx < a.Wo)) {
                act_n<float[8], 8>(acc, a.act2);
                half8 hv;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    psum[e] += acc[e];
                    hv[e] = (half_t)acc[e];
                }
                if constexpr (PROJ) *reinterpret_cast<half8*>(&Ds[opix * EW + cg * 8]) = hv;     // B operand rows of the projection
                else *reinterpret_cast<half8*>(a.out + (((size_t)n * a.Ho + gy) * a.Wo + gx) * cexp + c0 + cg * 8) = hv;
            }
        }
        XD_STAMP2(9);
        if (a.pool) {
            // per-tile channel sums in a fixed order: the 8 lanes of a wave that share a channel group, then the waves
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = psum[e];
                v += __shfl_xor(v, 8);
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                psum[e] = v;
            }
            if (lane < 8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) Ps[wave * 64 + lane * 8 + e] = psum[e];
            }
            __syncthreads();
            if (tid < 64 && c0 + tid < cexp) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < NT / 64; ++q) s += Ps[q * 64 + tid];
                a.pool[((size_t)n * tiles + tile) * cexp + c0 + tid] = s;
            }
        }
        if constexpr (PROJ) {
            // channel groups of a partial last chunk that the depthwise stage skipped must read as zero
            if (c0 + cw > cexp)
                for (int item = tid; item < DROWS * 8; item += NT)
                    if (c0 + (item & 7) * 8 >= cexp) *reinterpret_cast<uint4*>(&Ds[(item >> 3) * EW + (item & 7) * 8]) = make_uint4(0, 0, 0, 0);
            XD_STAMP2(10);
            __syncthreads();
            if (punit) {
                const bool cok = pc * 32 + r < a.cout;
                const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < (WIDE ? 5 : 4); ++ks) {
                    if (ks == 4 && !wide) break;
                    const bool ok = cok && c0 + ks * 16 + hh * 8 < cexp && ks * 16 + hh * 8 < cw;
                    const half8 w = ok ? w3f[ks] : zero8;
                    half8 df = *reinterpret_cast<const half8*>(&Ds[(prt * 32 + r) * EW + ks * 16 + hh * 8]);
                    if (ks == 4) df = hh ? zero8 : df;      // columns 72..79 are the next row
                    pacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, df, pacc, 0, 0, 0);
                }
            }
        }
        XD_STAMP2(11);
        __syncthreads();        // Es / Wd / Bd / Ps / Ds are rewritten by the next chunk (and by the staged output tile)
        if (c0 == c_begin) XD_STAMP(3);
        XD_STAMP2(12);
        if constexpr (ONE) break;
        c0 = cnext;
    }
    if constexpr (PROJ) {
        // lane = output pixel (row tile prt, row r), registers 4g..4g+3 = channels pc*32 + 8g + 4hh .. +3.
        // The tile leaves through LDS (Os, over the dead E / depthwise buffers) as 16-byte chunks that are consecutive along a tile
        // row: written straight from this layout every store instruction puts 8-byte pieces at the pixel stride (48 B for the
        // 24-channel maps), the L2 evicts the half-written lines under the input stream and the 80 x 80 outputs cost 4 - 10x their
        // size in HBM writes (profiles/r02_hbm_traffic.json before this: 97 MB written for a 9.8 MB output).
        half_t* Os = Es;
        const bool staged = a.stage_out != 0;                   // uniform: the tile fits the dead buffers (launcher)
        const int opix = prt * 32 + r;
        const int oy = opix / OW, ox = opix - oy * OW;
        const int gy = oy0 + oy, gx = ox0 + ox;
        if (punit && opix < OH * OW && (staged || (gy < a.Ho && gx < a.Wo))) {
            half_t* orow = staged ? Os + opix * a.cout : a.out + (((size_t)n * a.Ho + gy) * a.Wo + gx) * a.cout;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = pc * 32 + 8 * g + 4 * hh;
                if (c < a.cout) {           // cout % 8 == 0: the 4-channel group is entirely in range
                    const float4 b = *reinterpret_cast<const float4*>(&B3s[c]);
                    float v[4] = {pacc[4 * g + 0] + b.x, pacc[4 * g + 1] + b.y, pacc[4 * g + 2] + b.z, pacc[4 * g + 3] + b.w};
                    if (a.has_res) {        // residual (stride 1, cout == cin): the block's input at the same pixel
                        const half4 rr = *reinterpret_cast<const half4*>(&Xs[((oy * S + a.pad) * IW + ox * S + a.pad) * XW + c]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)rr[e];
                    }
                    half4 hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) hv[e] = (half_t)v[e];
                    *reinterpret_cast<half4*>(orow + c) = hv;
                }
            }
        }
        if (staged) {
            __syncthreads();
            const int oc8 = a.cout >> 3;
            for (int i = tid; i < OH * OW * oc8; i += NT) {
                const int px = (int)fd_div((unsigned)i, a.fd_oc8), c8 = i - px * oc8;
                const int py = px / OW, pxx = px - py * OW;
                const int y = oy0 + py, x = ox0 + pxx;
                if (y < a.Ho && x < a.Wo)
                    *reinterpret_cast<uint4*>(a.out + (((size_t)n * a.Ho + y) * a.Wo + x) * a.cout + c8 * 8) =
                        *reinterpret_cast<const uint4*>(Os + px * a.cout + c8 * 8);
            }
        }
    }
    XD_STAMP(4);
}

// ---- persistent single-chunk form (round 4) --------------------------------------------------------------------------------------
// The blocks of the large maps whose expanded width is ONE chunk (16 -> 64 -> 24 at 160 x 160 stride 2, 24 -> 72 -> 24 at 80 x 80) are the
// largest launches of the family: 6 400 / 3 200 workgroups that live 4.6 us each, two per CU -- 1.2 us of it the input region's memory
// round trip, every workgroup re-requesting the same weights. Here 512 workgroups (two per CU, as many as are resident) WALK the tiles:
//   * weights, biases and the depthwise weights' LDS copy are set up once per workgroup;
//   * the input region of the NEXT tile is copied by LDS-DMA (per-lane source: a pixel outside the image reads a zero block, the bias
//     column a block of ones -- the staging of expdw_kernel without its registers) as soon as the current tile's expand stage has read
//     X (the residual pixels are taken into registers during that stage), and lands under the depthwise / project / store stages;
//   * four barriers per tile instead of five plus a launch ramp.
// Same arithmetic at the same rounding points as expdw_kernel<..., ONE>: outputs are bit-identical (test_expdw_persistent_bit_identical).
__device__ const uint4 g_xd_const[2] = {{0u, 0u, 0u, 0u}, {0x3c003c00u, 0u, 0u, 0u}};      // what a chunk outside the image / the bias column of a pixel inside it reads

template <int K, int S, int OH, int OW, int KSM, int XW8, bool WD>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void expdw_one_kernel(ExpDwArgs a, int tiles_x, int tiles_y, int wpg, int dty, int dtx) {
    using G = ExpDwGeom<K, S, OH, OW>;
    constexpr int IW = G::IW, NPIX = G::NPIX, RT = G::RT, ROWS = G::ROWS;
    constexpr bool WIDE = WD;
    constexpr int XW = XW8 * 8, XC8 = XW8;
    static_assert(XW8 > 0 && KSM <= 2, "block shapes with a compile-time X row and at most two full K steps");
    extern __shared__ __attribute__((aligned(16))) half_t lds[];
    half_t* Xs = lds;                                       // [ROWS][XW] + 8 zero halfs
    half_t* Es = Xs + ROWS * XW + 8;                        // [ROWS][EW]
    half_t* Wd = Es + ROWS * EW;                            // [K*K][72]
    float* Bd = reinterpret_cast<float*>(Wd + K * K * EW);  // [72]
    half_t* Ds = reinterpret_cast<half_t*>(Bd + EW + NT / 64 * 64);   // [DROWS][EW]   (same offsets as expdw_kernel: one LDS size for both)
    constexpr int DROWS = (OH * OW + 31) / 32 * 32;
    float* B3s = reinterpret_cast<float*>(Ds + DROWS * EW); // [256]
    half_t* Os = reinterpret_cast<half_t*>(B3s + 256);      // [OH * OW][cout]: the finished tile, stored to memory during the NEXT tile

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int cin = a.cin, cexp = a.cexp;
    const int KSF = cin >> 4;
    const int tiles = tiles_x * tiles_y;
    // ---- this workgroup's tiles: XCD group g walks the tiles of its images, workgroup j of the group takes tiles j, j + wpg, ...
    int grp = 0, j = blockIdx.x, img0 = 0, nimg = a.n;
    if (a.xq > 0) { grp = blockIdx.x & 7; j = blockIdx.x >> 3; img0 = grp * a.xq; nimg = min(a.xq, a.n - img0); }
    if (nimg <= 0) return;
    int img = j / tiles, tl = j - img * tiles;
    int ty = tl / tiles_x, tx = tl - ty * tiles_x;
    if (img >= nimg) return;

    // ---- once per workgroup: weights (registers), depthwise weights / biases (LDS)
    const int t = wave & 1;
    half8 wf[KSM], wl, wf2[WIDE ? KSM : 1], wl2;
    float b1v, b1v2 = 0.f;
    const int kb = KSF * 16 + hh * 8;
    const int kcl = min(kb, cin - 8);
    {
        const int ch = min(t * 32 + r, cexp - 1);
        const half_t* wrow = a.w1 + (size_t)ch * cin;
#pragma unroll
        for (int ks = 0; ks < KSM; ++ks)
            if (ks < KSF) wf[ks] = *reinterpret_cast<const half8*>(wrow + ks * 16 + hh * 8);
        wl = *reinterpret_cast<const half8*>(wrow + kcl);
        b1v = a.b1[ch];
        if constexpr (WIDE) {
            const int ch2 = min(64 + r, cexp - 1);
            const half_t* wrow2 = a.w1 + (size_t)ch2 * cin;
#pragma unroll
            for (int ks = 0; ks < KSM; ++ks)
                if (ks < KSF) wf2[ks] = *reinterpret_cast<const half8*>(wrow2 + ks * 16 + hh * 8);
            wl2 = *reinterpret_cast<const half8*>(wrow2 + kcl);
            b1v2 = a.b1[ch2];
        }
    }
    auto last_frag = [&](const half8& w, float b) {
        const half_t hi = (half_t)b;
        const half_t lo = (half_t)(b - (float)hi);
        const half8 bw = {hi, lo, 0, 0, 0, 0, 0, 0};
        const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        return kb < cin ? w : (kb == cin ? bw : zero8);
    };
    const half8 wlast = last_frag(wl, b1v);
    half8 wlast2 = wlast;
    if constexpr (WIDE) wlast2 = last_frag(wl2, b1v2);
    const int cw = WIDE ? EW : 64, ng = cw >> 3;
    {
        const int wrow_i = tid / 9, wc8 = tid - wrow_i * 9;
        const bool wok = tid < K * K * 9 && wc8 < ng && wc8 * 8 < cexp;
        if (tid < K * K * 9) {
            const uint4 v = *reinterpret_cast<const uint4*>(a.wd + (size_t)min(wrow_i, K * K - 1) * cexp + min(wc8 * 8, cexp - 8));
            *reinterpret_cast<uint4*>(&Wd[wrow_i * EW + wc8 * 8]) = wok ? v : make_uint4(0, 0, 0, 0);
        }
        if (tid < EW) Bd[tid] = (tid < cw && tid < cexp) ? a.bd[min(tid, cexp - 1)] : 0.f;
        if (tid < 256) B3s[tid] = tid < a.cout ? a.b3[tid] : 0.f;
        if (tid == 0) *reinterpret_cast<uint4*>(&Xs[ROWS * XW]) = make_uint4(0, 0, 0, 0);
        // channel groups beyond cexp are never written by the depthwise stage and must read as zero in the projection
        if (cw > cexp)
            for (int item = tid; item < DROWS * ng; item += NT)
                if ((item % ng) * 8 >= cexp) *reinterpret_cast<uint4*>(&Ds[(item / ng) * EW + (item % ng) * 8]) = make_uint4(0, 0, 0, 0);
    }
    constexpr int PRT = DROWS / 32;
    const int pct = (a.cout + 31) >> 5;
    const int prt = wave % PRT, pc = wave / PRT;
    const bool punit = pc < pct;
    half8 w3f[WIDE ? 5 : 4];
    {
        const int co = min(pc * 32 + r, a.cout - 1);
#pragma unroll
        for (int ks = 0; ks < (WIDE ? 5 : 4); ++ks)
            w3f[ks] = *reinterpret_cast<const half8*>(a.w3 + (size_t)co * cexp + min(ks * 16 + hh * 8, cexp - 8));
    }

    // ---- LDS-DMA of a tile's input region: chunk idx = (pixel of the halo region, 16-byte piece of its X row), two instructions per wave
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    constexpr int NCHK = ROWS * XC8;                       // 16-byte chunks of the region (a multiple of 32)
    constexpr int DPW = (NCHK + NT - 1) / NT;               // DMA instructions per wave
    int cpy[DPW], cpx[DPW], cq[DPW];
#pragma unroll
    for (int u = 0; u < DPW; ++u) {
        const int idx = (wave * DPW + u) * 64 + lane;
        const int pix = idx / XC8;
        cq[u] = idx - pix * XC8;
        cpy[u] = pix / IW; cpx[u] = pix - cpy[u] * IW;
        if (idx >= NCHK || pix >= NPIX) cpy[u] = -100000;   // rows of the padding: always "outside"
    }
    const int c8 = cin >> 3;
    auto stage_x = [&](int im, int ty_, int tx_) {
        const half_t* xin = a.x + (size_t)(img0 + im) * a.H * a.W * cin;
        const int iy0 = ty_ * OH * S - a.pad, ix0 = tx_ * OW * S - a.pad;
#pragma unroll
        for (int u = 0; u < DPW; ++u) {
            if ((wave * DPW + u) * 64 >= NCHK) break;       // (uniform: the last wave's second instruction may be beyond the region)
            const int gy = iy0 + cpy[u], gx = ix0 + cpx[u];
            const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            const void* src = (inside && cq[u] < c8) ? (const void*)(xin + ((size_t)gy * a.W + gx) * cin + cq[u] * 8)
                                                      : (const void*)&g_xd_const[(inside && cq[u] == c8) ? 1 : 0];
            const unsigned dst = lds0 + (unsigned)((wave * DPW + u) * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
    };
    stage_x(img, ty, tx);

    const int oc8 = a.cout >> 3;
    // the finished tile of the PREVIOUS iteration leaves Os here (row-contiguous 16-byte chunks): issued in front of the next region's DMA, so
    // that the wait for that DMA at the top of the loop has nothing younger behind it
    int pn = -1, poy0 = 0, pox0 = 0;
    auto store_prev = [&]() {
        if (pn < 0) return;
        for (int i = tid; i < OH * OW * oc8; i += NT) {
            const int px = (int)fd_div((unsigned)i, a.fd_oc8), c8o = i - px * oc8;
            const int py = px / OW, pxx = px - py * OW;
            const int y = poy0 + py, x = pox0 + pxx;
            if (y < a.Ho && x < a.Wo)
                *reinterpret_cast<uint4*>(a.out + (((size_t)pn * a.Ho + y) * a.Wo + x) * a.cout + c8o * 8) =
                    *reinterpret_cast<const uint4*>(Os + px * a.cout + c8o * 8);
        }
    };
    while (true) {
        const int n = img0 + img;
        const int oy0 = ty * OH, ox0 = tx * OW;
        // this tile's region has landed (every wave waits for its own DMA instructions; the only memory operations behind them in the queue
        // would be the previous tile's stores, and those were issued in front of the DMA)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- expand on the matrix cores (expdw_kernel, stage 2)
        for (int rt = wave >> 1; rt < RT && t * 32 < cexp; rt += NT / 128) {
            floatx16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            const half_t* xrow = &Xs[(rt * 32 + r) * XW + hh * 8];
#pragma unroll
            for (int ks = 0; ks < KSM; ++ks)
                if (ks < KSF) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks], *reinterpret_cast<const half8*>(xrow + ks * 16), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlast, *reinterpret_cast<const half8*>(xrow + KSF * 16), acc, 0, 0, 0);
            asm volatile("s_nop 7\n\ts_nop 3" : "+v"(acc));      /* matrix result -> activation switch: every branch path gets its 12 wait states (pwdirect.hip act16) */
                act_n<floatx16, 16>(acc, a.act1);
            half_t* erow = &Es[(rt * 32 + r) * EW + t * 32 + 4 * hh];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                half4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = (half_t)acc[4 * g + e];
                *reinterpret_cast<half4*>(erow + 8 * g) = hv;
            }
        }
        if constexpr (WIDE) {
            for (int rt = wave; rt < RT; rt += NT / 64) {
                floatx16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
                const half_t* xrow = &Xs[(rt * 32 + r) * XW + hh * 8];
#pragma unroll
                for (int ks = 0; ks < KSM; ++ks)
                    if (ks < KSF) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf2[ks], *reinterpret_cast<const half8*>(xrow + ks * 16), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlast2, *reinterpret_cast<const half8*>(xrow + KSF * 16), acc, 0, 0, 0);
                asm volatile("s_nop 7\n\ts_nop 3" : "+v"(acc));
                        float v4[4] = {acc[0], acc[1], acc[2], acc[3]};
                act_n<float[4], 4>(v4, a.act1);
                half4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = (half_t)v4[e];
                *reinterpret_cast<half4*>(&Es[(rt * 32 + r) * EW + 64 + 4 * hh]) = hv;
            }
        }
        // the residual pixels of this wave's output unit leave X now: the region is overwritten under the stages below
        const int opix = prt * 32 + r;
        const int poy = opix / OW, pox = opix - poy * OW;
        half4 res[4];
        if (a.has_res && punit) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = pc * 32 + 8 * g + 4 * hh;
                res[g] = *reinterpret_cast<const half4*>(&Xs[((min(poy, OH - 1) * S + a.pad) * IW + pox * S + a.pad) * XW + min(c, a.cout - 4)]);
            }
        }
        __syncthreads();
        // ---- the previous tile's output, then the next tile's region
        store_prev();
        int nimg_i = img, nty = ty + dty, ntx = tx + dtx;
        if (ntx >= tiles_x) { ntx -= tiles_x; ++nty; }
        while (nty >= tiles_y) { nty -= tiles_y; ++nimg_i; }
        const bool more = nimg_i < nimg;
        if (more) stage_x(nimg_i, nty, ntx);
        // ---- depthwise over the LDS tile (expdw_kernel, stage 3)
        for (int item = tid; item < OH * OW * ng; item += NT) {
            const int op = WIDE ? item / 9 : item >> 3;
            const int cg = item - op * ng;
            if (cg * 8 >= cexp) continue;
            const int oy = op / OW, ox = op - oy * OW;
            float acc[8];
            {
                const float4 b0 = *reinterpret_cast<const float4*>(&Bd[cg * 8]), b1 = *reinterpret_cast<const float4*>(&Bd[cg * 8 + 4]);
                acc[0] = b0.x; acc[1] = b0.y; acc[2] = b0.z; acc[3] = b0.w; acc[4] = b1.x; acc[5] = b1.y; acc[6] = b1.z; acc[7] = b1.w;
            }
            const half_t* ebase = Es + ((oy * S) * IW + ox * S) * EW + cg * 8;
#pragma unroll 1
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const uint4 ev = *reinterpret_cast<const uint4*>(ebase + (ky * IW + kx) * EW);
                    const uint4 wv = *reinterpret_cast<const uint4*>(&Wd[(ky * K + kx) * EW + cg * 8]);
                    fma_mix_h8(acc, ev, wv);
                }
            act_n<float[8], 8>(acc, a.act2);
            half8 hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = (half_t)acc[e];
            *reinterpret_cast<half8*>(&Ds[op * EW + cg * 8]) = hv;
        }
        __syncthreads();            // (also: every thread's reads of Os for the previous tile's stores are complete)
        // ---- project (expdw_kernel, PROJ); the tile goes to Os and leaves during the next iteration
        floatx16 pacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) pacc[e] = 0.f;
        if (punit) {
            const bool cok = pc * 32 + r < a.cout;
            const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < (WIDE ? 5 : 4); ++ks) {
                const bool ok = cok && ks * 16 + hh * 8 < cexp && ks * 16 + hh * 8 < cw;
                const half8 w = ok ? w3f[ks] : zero8;
                half8 df = *reinterpret_cast<const half8*>(&Ds[(prt * 32 + r) * EW + ks * 16 + hh * 8]);
                if (ks == 4) df = hh ? zero8 : df;
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, df, pacc, 0, 0, 0);
            }
            if (opix < OH * OW) {
                half_t* orow = Os + opix * a.cout;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = pc * 32 + 8 * g + 4 * hh;
                    if (c < a.cout) {
                        const float4 b = *reinterpret_cast<const float4*>(&B3s[c]);
                        float v[4] = {pacc[4 * g + 0] + b.x, pacc[4 * g + 1] + b.y, pacc[4 * g + 2] + b.z, pacc[4 * g + 3] + b.w};
                        if (a.has_res) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += (float)res[g][e];
                        }
                        half4 hv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) hv[e] = (half_t)v[e];
                        *reinterpret_cast<half4*>(orow + c) = hv;
                    }
                }
            }
        }
        pn = n; poy0 = oy0; pox0 = ox0;
        if (!more) break;
        img = nimg_i; ty = nty; tx = ntx;
    }
    __syncthreads();
    store_prev();
}

template <int K, int S, int OH, int OW, int KSM, bool EXP, bool PROJ, int XW8, bool WD>
int launch_kw(const ExpDwArgs& a, int tiles_x, int tiles_y, int zsplit, size_t lds, hipStream_t s) {
    const dim3 grid(a.xq > 0 ? 8 * tiles_x : tiles_x, tiles_y * zsplit, a.xq > 0 ? a.xq : a.n);
    // (instantiated for the block shapes that take it: 3x3 on the 8 x 8 stride-2 / 8 x 16 stride-1 tiles of the large maps, cin 16 / 24 / 32)
    if constexpr (PROJ && EXP && KSM <= 2 && (XW8 == 3 || XW8 == 5) && K == 3 && OH == 8 && OW == (S == 1 ? 16 : 8)) {
        // persistent single-chunk form: the workgroups that are resident (two per CU) walk the tiles. Needs the staged output tile inside the dead E
        // buffer (no barrier between the project stage and its write) and a stride of whole tiles per step.
        const size_t lds_p = lds + (size_t)OH * OW * a.cout * sizeof(half_t);        // + the tile's own output buffer
        if ((WD ? a.cexp == EW : a.cexp <= 64) && !a.pool && !a.stamps && a.cout % 8 == 0 && a.cout <= 256 && lds_p <= 80 * 1024 &&
            fd_ok((unsigned long long)OH * OW * (a.cout >> 3), (unsigned)(a.cout >> 3)) && dn_knob("DN_EXPDW_PERSIST", 1)) {
            const int tiles = tiles_x * tiles_y;
            const int groups = a.xq > 0 ? 8 : 1;
            const long per_group = (long)(a.xq > 0 ? a.xq : a.n) * tiles;
            const int wpg = (int)std::min<long>(per_group, 512 / groups);       // two resident workgroups per CU
            DN_HIP_CHECK(dn_allow_big_lds(reinterpret_cast<const void*>(expdw_one_kernel<K, S, OH, OW, KSM, XW8, WD>)));
            dn_note_kernel("expdw_one_kernel<%d,%d,%d,%d,%d>", K, S, OH, OW, KSM);
            ExpDwArgs b = a;
            b.fd_oc8 = fastdiv((unsigned)(a.cout >> 3));
            hipLaunchKernelGGL((expdw_one_kernel<K, S, OH, OW, KSM, XW8, WD>), dim3(groups * wpg), dim3(NT), lds_p, s, b, tiles_x, tiles_y, wpg, wpg / tiles_x, wpg % tiles_x);
            return DN_OK;
        }
    }
    if constexpr (PROJ && EXP && KSM <= 2) {
        if ((WD ? a.cexp == EW : a.cexp <= 64) && dn_knob("DN_EXPDW_ONE", 1)) {          // the whole expanded width is one chunk (72 channels: the wide variant)
            DN_HIP_CHECK(dn_allow_big_lds(reinterpret_cast<const void*>(expdw_kernel<K, S, OH, OW, KSM, EXP, PROJ, XW8, WD, true>)));
            dn_note_kernel("expdw_kernel<%d,%d,%d,%d,%d,%s,%s>", K, S, OH, OW, KSM, EXP ? "true" : "false", PROJ ? "true" : "false");
            hipLaunchKernelGGL((expdw_kernel<K, S, OH, OW, KSM, EXP, PROJ, XW8, WD, true>), grid, dim3(NT), lds, s, a, tiles_x, tiles_y, zsplit);
            return DN_OK;
        }
    }
    if constexpr (WD) {
        return DN_E_UNSUPPORTED;        // (launch_k only picks the wide variant together with the single-chunk form)
    } else {
        DN_HIP_CHECK(dn_allow_big_lds(reinterpret_cast<const void*>(expdw_kernel<K, S, OH, OW, KSM, EXP, PROJ, XW8, WD>)));
        dn_note_kernel("expdw_kernel<%d,%d,%d,%d,%d,%s,%s>", K, S, OH, OW, KSM, EXP ? "true" : "false", PROJ ? "true" : "false");
        hipLaunchKernelGGL((expdw_kernel<K, S, OH, OW, KSM, EXP, PROJ, XW8, WD>), grid, dim3(NT), lds, s, a, tiles_x, tiles_y, zsplit);
        return DN_OK;
    }
}
template <int K, int S, int OH, int OW, int KSM, bool EXP, bool PROJ, int XW8>
int launch_k(const ExpDwArgs& a, int tiles_x, int tiles_y, int zsplit, size_t lds, hipStream_t s) {
    if constexpr (PROJ && EXP && KSM <= 2) {
        // the wide variant (a last chunk of 72 channels taken whole) exists for the one shape that uses it: the whole expanded width IS that chunk
        // (24 -> 72 -> 24). As a tail of a longer chunk run (cexp = 136, 200 ... with cin <= 40) it was never dispatched by the model zoo and every
        // such instantiation spilled 36 - 76 B per lane at the 128-register cap: not instantiated any more.
        if (a.cexp == EW && dn_knob("DN_EXPDW_ONE", 1)) return launch_kw<K, S, OH, OW, KSM, EXP, PROJ, XW8, true>(a, tiles_x, tiles_y, zsplit, lds, s);
    }
    return launch_kw<K, S, OH, OW, KSM, EXP, PROJ, XW8, false>(a, tiles_x, tiles_y, zsplit, lds, s);
}

template <int K, int S, int OH, int OW>
int launch_t(const ExpDwArgs& a0, hipStream_t s) {
    using G = ExpDwGeom<K, S, OH, OW>;
    ExpDwArgs a = a0;
    const bool proj = a.w3 != nullptr, exp = a.w1 != nullptr;
    constexpr int DROWS = (OH * OW + 31) / 32 * 32;
    const size_t lds0 = ((size_t)G::ROWS * a.xw + 8 + (exp ? (size_t)G::ROWS * EW : 0) + K * K * EW + (proj ? (size_t)DROWS * EW : 0)) * sizeof(half_t) +
                       (EW + NT / 64 * 64 + (proj ? 256 : 0)) * sizeof(float);
    const size_t lds = lds0;
    DN_REQUIRE(lds0 <= 160 * 1024, "expand+depthwise: LDS %zu B exceeds 160 KB", lds0);
    // split the 64-channel chunks over grid.y until there are enough workgroups to fill the chip a few times over
    // (not with a project stage: it sums over all chunks inside the workgroup)
    const int tiles_x = dn_cdiv(a.Wo, OW), tiles_y = dn_cdiv(a.Ho, OH), tiles = tiles_x * tiles_y, chunks = dn_cdiv(a.cexp, 64);
    const int want = 1024;
    int cpw = chunks;
    while (!proj && cpw > 1 && (long)tiles * a.n * dn_cdiv(chunks, cpw) < want) --cpw;
    a.chunks_per_wg = cpw;
    {   // dev: DN_EXPDW_STAMP_SEL = 10 * H + stride stamps only the launches of that shape (every launch writes the same buffer)
        const int sel = g_xd_stamps ? dn_knob("DN_EXPDW_STAMP_SEL", 0) : 0;
        a.stamps = (sel == 0 || sel == 10 * a.H + S) ? g_xd_stamps : nullptr;
    }
    if (proj) {
        // the output tile is staged over the E .. depthwise-output buffers (everything between the X rows and the project bias)
        const size_t avail = ((exp ? (size_t)G::ROWS * EW : 0) + K * K * EW + (size_t)DROWS * EW) * sizeof(half_t) + (EW + NT / 64 * 64) * sizeof(float);
        a.stage_out = (size_t)OH * OW * a.cout * sizeof(half_t) <= avail && a.cout % 8 == 0 &&
                      fd_ok((unsigned long long)OH * OW * (a.cout >> 3), (unsigned)(a.cout >> 3));
        a.fd_oc8 = fastdiv((unsigned)(a.cout >> 3));
    }
    const int zsplit = dn_cdiv(chunks, cpw);
    DN_REQUIRE((long)tiles_y * zsplit <= 65535 && a.n <= 65535, "expand+depthwise: grid too large");
    if (proj) DN_REQUIRE((DROWS / 32) * dn_cdiv(a.cout, 32) <= NT / 64, "expand+depthwise+project: %d output units for %d waves", (DROWS / 32) * dn_cdiv(a.cout, 32), NT / 64);
    // the A fragments of the expand live in registers: 2 full K steps + the tail/bias step cover cin <= 40 (fewer registers -> more
    // waves), else 8. The block shapes of the backbones (cin 16 / 24 + project) get compile-time X row widths.
    if (!exp) return proj ? launch_k<K, S, OH, OW, 2, false, true, 0>(a, tiles_x, tiles_y, zsplit, lds, s) : DN_E_UNSUPPORTED;
    if (proj) {
        if (a.xw == 24) return launch_k<K, S, OH, OW, 2, true, true, 3>(a, tiles_x, tiles_y, zsplit, lds, s);
        if (a.xw == 40 && a.cin <= 40) return launch_k<K, S, OH, OW, 2, true, true, 5>(a, tiles_x, tiles_y, zsplit, lds, s);
        if (a.xw == 56) return launch_k<K, S, OH, OW, 2, true, true, 7>(a, tiles_x, tiles_y, zsplit, lds, s);        // cin 40
        if (a.cin <= 40) return launch_k<K, S, OH, OW, 2, true, true, 0>(a, tiles_x, tiles_y, zsplit, lds, s);
        if (a.xw == 88) return launch_k<K, S, OH, OW, 5, true, true, 11>(a, tiles_x, tiles_y, zsplit, lds, s);       // cin 80: the 20 x 20 blocks
        if (a.cin <= 88) return launch_k<K, S, OH, OW, 5, true, true, 0>(a, tiles_x, tiles_y, zsplit, lds, s);
        if (a.xw == 104) return launch_k<K, S, OH, OW, 6, true, true, 13>(a, tiles_x, tiles_y, zsplit, lds, s);      // cin 96
        return launch_k<K, S, OH, OW, XKS, true, true, 0>(a, tiles_x, tiles_y, zsplit, lds, s);
    }
    if (a.xw == 56) return launch_k<K, S, OH, OW, 2, true, false, 7>(a, tiles_x, tiles_y, zsplit, lds, s);          // cin 40
    if (a.xw == 40 && a.cin <= 40) return launch_k<K, S, OH, OW, 2, true, false, 5>(a, tiles_x, tiles_y, zsplit, lds, s);   // cin 24 / 32
    if (a.cin <= 40) return launch_k<K, S, OH, OW, 2, true, false, 0>(a, tiles_x, tiles_y, zsplit, lds, s);
    if (a.xw == 88) return launch_k<K, S, OH, OW, 5, true, false, 11>(a, tiles_x, tiles_y, zsplit, lds, s);         // cin 80 (squeeze-excitation blocks of the 20 x 20 / 10 x 10 maps)
    return launch_k<K, S, OH, OW, XKS, true, false, 0>(a, tiles_x, tiles_y, zsplit, lds, s);
}

template <int K, int S>
int launch_ks(const ExpDwArgs& a, int oh, int ow, hipStream_t s) {
    if (oh == 4) return launch_t<K, S, 4, 8>(a, s);
    if (oh == 8 && ow == 8) return launch_t<K, S, 8, 8>(a, s);
    if (oh == 8) {
        if constexpr (S == 1) return launch_t<K, S, 8, 16>(a, s);
        else return launch_t<K, S, 8, 8>(a, s);
    }
    if constexpr (S == 1) { if (oh == 10) return launch_t<K, S, 10, 10>(a, s); }
    if (ow == 10) return launch_t<K, S, 5, 10>(a, s);
    return launch_t<K, S, 5, 5>(a, s);
}

}  // namespace

void expdw_tile(int Ho, int Wo, int stride, int* oh, int* ow, int proj_cout = 0) {
    // dev knob: DN_EXPDW_TILE = 48 / 88 forces the 4 x 8 / 8 x 8 output tile on the large maps (tile-size experiments)
    const int force = 0;
    if (force && Wo >= 32 && Ho >= 32) { *oh = force / 10; *ow = force % 10; return; }
    if (Wo <= 5 && Ho <= 5) { *oh = 5; *ow = 5; }
    else if (Wo <= 10) { *oh = 5; *ow = 10; }
    else if (Wo < 32 && Wo % 16 != 0) {                     // 19x19 / 20x20 maps: 10-wide tiles waste nothing
        if (stride == 1) { *oh = 10; *ow = 10; } else { *oh = 5; *ow = 10; }       // (measured again in round 4 on the V2 model at 300 x 300: 5 x 10 tiles there are 2.4 % slower, although the 10 x 10 variants with cin >= 64 spill 36 - 44 B per lane)
        // with a project stage the (pixel tile x channel tile) units must fit the 8 waves: 100 pixels x 80 channels = 12 units, 50 x 80 = 6
        if (proj_cout > 0 && ((*oh * *ow + 31) / 32) * ((proj_cout + 31) / 32) > NT / 64) { *oh = 5; *ow = 10; }
    }
    else { *oh = 8; *ow = (stride == 1) ? 16 : 8; }
}

int expdw_tiles_per_image(int Ho, int Wo, int stride) {
    int oh, ow;
    expdw_tile(Ho, Wo, stride, &oh, &ow);
    return dn_cdiv(Ho, oh) * dn_cdiv(Wo, ow);
}

bool expdw_supported(int cin, int cexp, int k, int stride) {
    return cin % 8 == 0 && cin <= 16 * XKS && cexp % 8 == 0 && (k == 3 || k == 5) && (stride == 1 || stride == 2);
}

// the project stage keeps one MFMA accumulator per wave: (pixels of the tile / 32) x (cout / 32) units must fit the 8 waves
bool expdw_project_supported(int cexp, int cout, int Ho, int Wo, int stride) {
    int oh, ow;
    expdw_tile(Ho, Wo, stride, &oh, &ow, cout);
    return cexp % 8 == 0 && cout % 8 == 0 && ((oh * ow + 31) / 32) * ((cout + 31) / 32) <= NT / 64;
}

int launch_expdw(const ExpDwArgs& a0, hipStream_t s) {
    ExpDwArgs a = a0;
    DN_REQUIRE(expdw_supported(a.cin, a.cexp, a.k, a.stride), "expand+depthwise: unsupported cin=%d cexp=%d k=%d stride=%d", a.cin, a.cexp,
               a.k, a.stride);
    DN_REQUIRE(a.w1 || a.cin == a.cexp, "depthwise+project without expand needs cexp == cin");
    DN_REQUIRE(a.w1 || a.w3, "expand+depthwise: neither an expand nor a project stage (use the depthwise kernel)");
    DN_REQUIRE(!a.w3 || (!a.pool && expdw_project_supported(a.cexp, a.cout, a.Ho, a.Wo, a.stride)), "expand+depthwise+project: unsupported cout=%d / tile", a.cout);
    DN_REQUIRE(!a.has_res || (a.w3 && a.stride == 1 && a.cout == a.cin), "expand+depthwise+project: residual needs stride 1 and cout == cin");
    DN_REQUIRE(a.n > 0 && a.H > 0 && a.W > 0 && a.Ho > 0 && a.Wo > 0, "expand+depthwise: empty problem");
    a.xw = ((a.cin + 15) / 16) * 16 + 8;
    int oh, ow;
    expdw_tile(a.Ho, a.Wo, a.stride, &oh, &ow, a.w3 ? a.cout : 0);
    if (a.k == 3 && a.stride == 1) return launch_ks<3, 1>(a, oh, ow, s);
    if (a.k == 3 && a.stride == 2) return launch_ks<3, 2>(a, oh, ow, s);
    if (a.k == 5 && a.stride == 1) return launch_ks<5, 1>(a, oh, ow, s);
    return launch_ks<5, 2>(a, oh, ow, s);
}
