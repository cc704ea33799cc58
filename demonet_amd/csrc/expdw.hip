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

#include "common.h"

static long long* g_xd_stamps = nullptr;     // dev hook (tools/probe_expdw.py): per-workgroup phase stamps
extern "C" __attribute__((visibility("default"))) void dn_debug_expdw_stamps(void* dev_ptr) { g_xd_stamps = (long long*)dev_ptr; }
#define XD_STAMP(k) do { if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)

namespace {

constexpr int EW = 72;          // halfs per E row in LDS: 64 channels + 8 pad (144 B = 9 x 16 B)
constexpr int XKS = 8;          // max 16-deep K steps of the expand (cin <= 128)
constexpr int NT = 512;         // threads per workgroup: 8 waves; two workgroups per CU give 4 waves per SIMD to hide LDS/L2 latency

template <int K, int S, int OH, int OW>
struct ExpDwGeom {
    static constexpr int IH = (OH - 1) * S + K, IW = (OW - 1) * S + K, NPIX = IH * IW;
    static constexpr int RT = (NPIX + 31) / 32, ROWS = RT * 32;
};

template <int K, int S, int OH, int OW, int KSM, bool EXP, bool PROJ>
__global__ __launch_bounds__(NT) void expdw_kernel(ExpDwArgs a, int tiles, int zsplit) {
    using G = ExpDwGeom<K, S, OH, OW>;
    constexpr int IW = G::IW, NPIX = G::NPIX, RT = G::RT, ROWS = G::ROWS;
    static_assert(K * K * 8 <= NT, "one 16-byte piece of the chunk's depthwise weights per thread");
    extern __shared__ __attribute__((aligned(16))) half_t lds[];
    const int XW = a.xw;                                    // halfs per X row: round16(cin) + 8
    half_t* Xs = lds;                                       // [ROWS][XW]
    half_t* Es = Xs + ROWS * XW;                            // [ROWS][EW]
    half_t* Wd = Es + (EXP ? ROWS * EW : 0);                // [K*K][64]   depthwise weights of the chunk
    float* Bd = reinterpret_cast<float*>(Wd + K * K * 64);  // [64]        depthwise bias of the chunk
    float* Ps = Bd + 64;                                    // [NT/64][64] pooled-sum scratch (one row per wave)
    half_t* Ds = reinterpret_cast<half_t*>(Ps + NT / 64 * 64);   // [DROWS][EW] depthwise output of the chunk (PROJ only)
    constexpr int DROWS = (OH * OW + 31) / 32 * 32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    // flat 1-D grid [image slot][chunk split z][tile]; XCD grouping (common.h): the workgroups with equal (index % 8) share the
    // images of one group of a.xq images
    int n, rem;
    {
        const int per_image = tiles * zsplit;
        if (a.xq > 0) {
            const int g = blockIdx.x & 7, w = blockIdx.x >> 3;
            const int j = w / per_image;
            rem = w - j * per_image;
            n = g * a.xq + j;
        } else {
            n = blockIdx.x / per_image;
            rem = blockIdx.x - n * per_image;
        }
        if (n >= a.n) return;
    }
    const int zblk = rem / tiles, tile = rem - zblk * tiles;
    const int tiles_x = (a.Wo + OW - 1) / OW;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int oy0 = ty * OH, ox0 = tx * OW;
    const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;
    const int cin = a.cin, cexp = a.cexp;
    const int KS = (cin + 15) >> 4;
    const int c_begin = zblk * a.chunks_per_wg * 64;
    const int c_end = min(cexp, c_begin + a.chunks_per_wg * 64);

    // Per-chunk operands that come from global memory, requested one phase ahead of their use so that their latency hides
    // under the staging / depthwise work: this wave's A fragments (channel tile t = wave & 1) and bias of the expand, and
    // this thread's 16-byte piece of the chunk's depthwise weights / bias (written to LDS at the top of the chunk).
    const int t = wave & 1;
    half8 wf[KSM];
    float4 bv[4];
    uint4 wdreg = make_uint4(0, 0, 0, 0);
    float bdreg = 0.f;
    auto request_chunk = [&](int c0) {
        if constexpr (EXP) {
            const int ch = c0 + t * 32 + r;
#pragma unroll
            for (int ks = 0; ks < KSM; ++ks) {
                const int k = ks * 16 + hh * 8;
                half8 w = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ks < KS && ch < cexp && k < cin) w = *reinterpret_cast<const half8*>(a.w1 + (size_t)ch * cin + k);
                wf[ks] = w;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = c0 + t * 32 + 8 * g + 4 * hh;
                bv[g] = (c < cexp) ? *reinterpret_cast<const float4*>(a.b1 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        wdreg = make_uint4(0, 0, 0, 0);
        if (tid < K * K * 8 && c0 + (tid & 7) * 8 < cexp)
            wdreg = *reinterpret_cast<const uint4*>(a.wd + (size_t)(tid >> 3) * cexp + c0 + (tid & 7) * 8);
        bdreg = (tid < 64 && c0 + tid < cexp) ? a.bd[c0 + tid] : 0.f;
    };

    XD_STAMP(0);
    // ---- 1. input region -> LDS (zeros outside the image, in the K padding and in the row padding)
    {
        const int c8 = cin >> 3, xc8 = XW >> 3;
        const half_t* xin = a.x + (size_t)n * a.H * a.W * cin;
        for (int i0 = tid; i0 < ROWS * xc8; i0 += NT * 4) {
            uint4 v[4];
            int dst[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = i0 + NT * u;
                const int pix = idx / xc8, q = idx - pix * xc8;
                const int py = pix / IW, px = pix - py * IW;
                const int gy = iy0 + py, gx = ix0 + px;
                v[u] = make_uint4(0, 0, 0, 0);
                dst[u] = idx < ROWS * xc8 ? pix * XW + q * 8 : -1;
                if (idx < ROWS * xc8 && pix < NPIX && q < c8 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                    v[u] = *reinterpret_cast<const uint4*>(xin + ((size_t)gy * a.W + gx) * cin + q * 8);
            }
            if (i0 == tid) request_chunk(c_begin);      // behind the first batch of region loads (memory returns in order)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (dst[u] >= 0) *reinterpret_cast<uint4*>(&Xs[dst[u]]) = v[u];
        }
    }
    // which of this lane's MFMA pixels (row r of every row tile) lie inside the image: bit rt
    unsigned inside_bits = 0;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int pix = rt * 32 + r;
        const int py = pix / IW, px = pix - py * IW;
        const int gy = iy0 + py, gx = ix0 + px;
        if (pix < NPIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) inside_bits |= 1u << rt;
    }
    static_assert(RT <= 32, "row-tile mask is 32 bits");
    __syncthreads();
    XD_STAMP(1);

    const int cg = tid & 7;                 // this thread's 8-channel group in the depthwise stage (NT % 8 == 0)
    // project stage (PROJ): wave = one (32-pixel row tile, 32-channel tile) unit of the [OH*OW][cout] output; its accumulator
    // lives across the chunk loop (the projection sums over all expanded channels)
    constexpr int PRT = DROWS / 32;
    const int pct = PROJ ? (a.cout + 31) >> 5 : 0;
    const int prt = wave % PRT, pc = wave / PRT;          // launcher guarantees PRT * pct <= NT / 64
    const bool punit = PROJ && pc < pct;
    floatx16 pacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) pacc[e] = 0.f;
    for (int c0 = c_begin; c0 < c_end; c0 += 64) {
        if (tid < K * K * 8) *reinterpret_cast<uint4*>(&Wd[(tid >> 3) * 64 + (tid & 7) * 8]) = wdreg;
        if (tid < 64) Bd[tid] = bdreg;
        // ---- 2. expand on the matrix cores. A = weight rows (channels), B = pixel rows: the accumulator then holds, per lane,
        //         pixel (lane & 31) and channels 8g + 4*(lane >> 5) .. +3 in registers 4g .. 4g+3.
        //         wave w: channel tile t = w & 1 of the chunk, row tiles (w >> 1), (w >> 1) + 4, ...
        for (int rt = wave >> 1; EXP && rt < RT && c0 + t * 32 < cexp; rt += NT / 128) {
            floatx16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            const half_t* xrow = &Xs[(rt * 32 + r) * XW + hh * 8];
#pragma unroll
            for (int ks = 0; ks < KSM; ++ks) {
                if (ks < KS) {
                    const half8 xf = *reinterpret_cast<const half8*>(xrow + ks * 16);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks], xf, acc, 0, 0, 0);
                }
            }
            const bool inside = (inside_bits >> rt) & 1u;
            half_t* erow = &Es[(rt * 32 + r) * EW + t * 32 + 4 * hh];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                half4 hv = {0, 0, 0, 0};
                if (inside) {
                    hv[0] = (half_t)dn_act(acc[4 * g + 0] + bv[g].x, a.act1);
                    hv[1] = (half_t)dn_act(acc[4 * g + 1] + bv[g].y, a.act1);
                    hv[2] = (half_t)dn_act(acc[4 * g + 2] + bv[g].z, a.act1);
                    hv[3] = (half_t)dn_act(acc[4 * g + 3] + bv[g].w, a.act1);
                }
                *reinterpret_cast<half4*>(erow + 8 * g) = hv;
            }
        }
        if (c0 + 64 < c_end) request_chunk(c0 + 64);       // lands under the depthwise stage
        __syncthreads();
        if (c0 == c_begin) XD_STAMP(2);

        // ---- 3. depthwise over the LDS tile: item = (output pixel, 8-channel group)
        float psum[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) psum[e] = 0.f;
        const bool cvalid = c0 + cg * 8 < cexp;
        for (int item = tid; cvalid && item < OH * OW * 8; item += NT) {
            const int opix = item >> 3;
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
                    const uint4 wv = *reinterpret_cast<const uint4*>(&Wd[(ky * K + kx) * 64 + cg * 8]);
                    fma_mix_h8(acc, ev, wv);
                }
            const int gy = oy0 + oy, gx = ox0 + ox;
            if (PROJ || (gy < a.Ho && gx < a.Wo)) {
                half8 hv;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = dn_act(acc[e], a.act2);
                    psum[e] += v;
                    hv[e] = (half_t)v;
                }
                if constexpr (PROJ) *reinterpret_cast<half8*>(&Ds[opix * EW + cg * 8]) = hv;     // B operand rows of the projection
                else *reinterpret_cast<half8*>(a.out + (((size_t)n * a.Ho + gy) * a.Wo + gx) * cexp + c0 + cg * 8) = hv;
            }
        }
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
            if (c0 + 64 > cexp)
                for (int item = tid; item < DROWS * 8; item += NT)
                    if (c0 + (item & 7) * 8 >= cexp) *reinterpret_cast<uint4*>(&Ds[(item >> 3) * EW + (item & 7) * 8]) = make_uint4(0, 0, 0, 0);
            __syncthreads();
            if (punit) {
                const int co = pc * 32 + r;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int k = c0 + ks * 16 + hh * 8;
                    half8 w = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (co < a.cout && k < cexp) w = *reinterpret_cast<const half8*>(a.w3 + (size_t)co * cexp + k);
                    const half8 df = *reinterpret_cast<const half8*>(&Ds[(prt * 32 + r) * EW + ks * 16 + hh * 8]);
                    pacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, df, pacc, 0, 0, 0);
                }
            }
        }
        __syncthreads();        // Es / Wd / Bd / Ps / Ds are rewritten by the next chunk
        if (c0 == c_begin) XD_STAMP(3);
    }
    if constexpr (PROJ) {
        // lane = output pixel (row tile prt, row r), registers 4g..4g+3 = channels pc*32 + 8g + 4hh .. +3
        const int opix = prt * 32 + r;
        const int oy = opix / OW, ox = opix - oy * OW;
        const int gy = oy0 + oy, gx = ox0 + ox;
        if (punit && opix < OH * OW && gy < a.Ho && gx < a.Wo) {
            half_t* orow = a.out + (((size_t)n * a.Ho + gy) * a.Wo + gx) * a.cout;
            const half_t* xrow = a.x + (((size_t)n * a.H + gy) * a.W + gx) * cin;      // residual (stride 1, cout == cin)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = pc * 32 + 8 * g + 4 * hh;
                if (c < a.cout) {           // cout % 8 == 0: the 4-channel group is entirely in range
                    const float4 b = *reinterpret_cast<const float4*>(a.b3 + c);
                    float v[4] = {pacc[4 * g + 0] + b.x, pacc[4 * g + 1] + b.y, pacc[4 * g + 2] + b.z, pacc[4 * g + 3] + b.w};
                    if (a.has_res) {
                        const half4 rr = *reinterpret_cast<const half4*>(xrow + c);
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
    }
    XD_STAMP(4);
}

template <int K, int S, int OH, int OW, int KSM, bool EXP, bool PROJ>
int launch_k(const ExpDwArgs& a, dim3 grid3, size_t lds, hipStream_t s) {
    const int tiles = grid3.x, zsplit = grid3.z;
    const dim3 grid((unsigned)tiles * zsplit * (a.xq > 0 ? 8 * a.xq : a.n));
    DN_HIP_CHECK(dn_allow_big_lds(reinterpret_cast<const void*>(expdw_kernel<K, S, OH, OW, KSM, EXP, PROJ>)));
    dn_note_kernel("expdw_kernel<%d,%d,%d,%d,%d,%s,%s>", K, S, OH, OW, KSM, EXP ? "true" : "false", PROJ ? "true" : "false");
    hipLaunchKernelGGL((expdw_kernel<K, S, OH, OW, KSM, EXP, PROJ>), grid, dim3(NT), lds, s, a, tiles, zsplit);
    return DN_OK;
}

template <int K, int S, int OH, int OW>
int launch_t(const ExpDwArgs& a0, hipStream_t s) {
    using G = ExpDwGeom<K, S, OH, OW>;
    ExpDwArgs a = a0;
    const bool proj = a.w3 != nullptr, exp = a.w1 != nullptr;
    constexpr int DROWS = (OH * OW + 31) / 32 * 32;
    const size_t lds = ((size_t)G::ROWS * a.xw + (exp ? (size_t)G::ROWS * EW : 0) + K * K * 64 + (proj ? (size_t)DROWS * EW : 0)) * sizeof(half_t) +
                       (64 + NT / 64 * 64) * sizeof(float);
    DN_REQUIRE(lds <= 160 * 1024, "expand+depthwise: LDS %zu B exceeds 160 KB", lds);
    // split the 64-channel chunks over grid.z until there are enough workgroups to fill the chip a few times over
    // (not with a project stage: it sums over all chunks inside the workgroup)
    const int tiles = dn_cdiv(a.Ho, OH) * dn_cdiv(a.Wo, OW), chunks = dn_cdiv(a.cexp, 64);
    const int want = dn_knob("DN_EXPDW_WGS", 1024);
    int cpw = chunks;
    while (!proj && cpw > 1 && (long)tiles * a.n * dn_cdiv(chunks, cpw) < want) --cpw;
    a.chunks_per_wg = cpw;
    a.stamps = g_xd_stamps;
    const dim3 grid(tiles, a.n, dn_cdiv(chunks, cpw));
    if (proj) DN_REQUIRE((DROWS / 32) * dn_cdiv(a.cout, 32) <= NT / 64, "expand+depthwise+project: %d output units for %d waves", (DROWS / 32) * dn_cdiv(a.cout, 32), NT / 64);
    // the A fragments of the expand live in registers: 2 K steps cover cin <= 32 (fewer registers -> more waves), else 8
    if (!exp) return proj ? launch_k<K, S, OH, OW, 2, false, true>(a, grid, lds, s) : DN_E_UNSUPPORTED;
    if (proj) {
        if (a.cin <= 32) return launch_k<K, S, OH, OW, 2, true, true>(a, grid, lds, s);
        return launch_k<K, S, OH, OW, XKS, true, true>(a, grid, lds, s);
    }
    if (a.cin <= 32) return launch_k<K, S, OH, OW, 2, true, false>(a, grid, lds, s);
    return launch_k<K, S, OH, OW, XKS, true, false>(a, grid, lds, s);
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

void expdw_tile(int Ho, int Wo, int stride, int* oh, int* ow) {
    // dev knob: DN_EXPDW_TILE = 48 / 88 forces the 4 x 8 / 8 x 8 output tile on the large maps (tile-size experiments)
    const int force = dn_knob("DN_EXPDW_TILE", 0);
    if (force && Wo >= 32 && Ho >= 32) { *oh = force / 10; *ow = force % 10; return; }
    if (Wo <= 5 && Ho <= 5) { *oh = 5; *ow = 5; }
    else if (Wo <= 10) { *oh = 5; *ow = 10; }
    else if (Wo < 32 && Wo % 16 != 0) {                     // 19x19 / 20x20 maps: 10-wide tiles waste nothing
        if (stride == 1) { *oh = 10; *ow = 10; } else { *oh = 5; *ow = 10; }
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
    expdw_tile(Ho, Wo, stride, &oh, &ow);
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
    expdw_tile(a.Ho, a.Wo, a.stride, &oh, &ow);
    if (a.k == 3 && a.stride == 1) return launch_ks<3, 1>(a, oh, ow, s);
    if (a.k == 3 && a.stride == 2) return launch_ks<3, 2>(a, oh, ow, s);
    if (a.k == 5 && a.stride == 1) return launch_ks<5, 1>(a, oh, ow, s);
    return launch_ks<5, 2>(a, oh, ow, s);
}
