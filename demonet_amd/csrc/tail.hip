// The "extras" tail of the backbone in ONE launch: a run of tiny 1x1 / depthwise layers whose maps have at most 32 output
// pixels (5x5, 3x3, 2x2, 1x1 in the SSDLite models: ssd_mobilenetv3.py:39-54,112-116; backbone.py:70-111 for the V2 hub model).
//
// As separate launches every one of these layers costs a dependent kernel boundary plus two exposed memory round trips
// (~8 us) for microseconds of work: 11 launches per forward. Here one 512-thread workgroup per image walks the run with the
// activations resident in LDS (ping-pong buffers); only the weights stream in from L2 (shared by all images' workgroups) and
// the pyramid features go out to HBM for the heads. (Touching the whole weight range first, to turn the per-layer HBM misses
// into L2 hits, was measured slower: the up-front wait costs more than the misses it removes.)
//   1x1: M <= 32 pixels is one MFMA row tile. Wave w owns the 32-channel output tiles w, w+8, ...; per tile it streams the
//        weight rows as A fragments straight from L2 (16 B per lane, 8 K-steps per batch, double-buffered) against the
//        pixel rows in LDS (B fragments). fp32 accumulate, bias + activation, fp16 result to LDS (and HBM).
//   depthwise: thread = (output pixel, 8-channel group), v_fma_mix_f32 over the taps, LDS -> LDS.
// Rounding points are those of the separate kernels (fp16 activations, fp32 accumulation), so the features agree with the
// launch-per-layer path.
#include "common.h"

#ifdef DN_DEV_STAMPS
static long long* g_tail_stamps = nullptr;     // dev build only (tools/probe_tail.py): per-workgroup stamps [n][32]
extern "C" __attribute__((visibility("default"))) void dn_debug_tail_stamps(void* dev_ptr) { g_tail_stamps = (long long*)dev_ptr; }
#define TL_STAMP(k) do { if (a.stamps && threadIdx.x == 0) a.stamps[blockIdx.x * 32 + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
constexpr long long* g_tail_stamps = nullptr;
#define TL_STAMP(k) do { } while (0)
#endif

namespace {

constexpr int TT = 512;             // threads per workgroup
constexpr int NWV = TT / 64;        // waves
constexpr int PART_SLICES = 8;      // max K slices per output tile

__global__ __launch_bounds__(TT) void tail_kernel(TailArgs a, int nimg) {
    extern __shared__ __attribute__((aligned(16))) half_t lds[];
    half_t* buf[2] = {lds, lds + a.buf_halfs};          // ping (input, odd outputs) / pong (even outputs): sized separately
    float* part = reinterpret_cast<float*>(lds + a.buf_halfs + a.buf2_halfs);
    // the op table lives in kernel-argument memory: read through the scalar cache, field by field, every first touch is a
    // memory round trip on the critical path. Copy it to LDS once.
    __shared__ TailOp ops_sh[TAIL_MAX_OPS];
    for (int i = threadIdx.x; i < (int)(sizeof(TailOp) * TAIL_MAX_OPS / 4); i += TT)
        reinterpret_cast<int*>(ops_sh)[i] = reinterpret_cast<const int*>(a.op)[i];       // [NWV][1024] fp32 partial tiles of the K slices
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    // XCD grouping (common.h): workgroup b serves image (b % 8) * xq + b / 8 -- the XCD that produced this image's input
    const int n = a.xq > 0 ? (int)(blockIdx.x & 7) * a.xq + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (n >= nimg) return;
    TL_STAMP(0);

    // stage the first op's input [pixels][cin] (NHWC fp16, contiguous per image)
    {
        // LDS rows are padded by 8 halfs (16 B): a row stride that is a multiple of 128 B would put every pixel's fragment
        // in the same banks
        const TailOp& o = a.op[0];
        const int c8 = o.cin / 8, chunks = o.hin * o.win * c8;
        const uint4* src = reinterpret_cast<const uint4*>(a.in0 + (size_t)n * a.in0_stride);
        for (int i0 = tid; i0 < chunks; i0 += TT * 4) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = src[min(i0 + TT * u, chunks - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + TT * u;
                if (i < chunks) *reinterpret_cast<uint4*>(buf[0] + (size_t)(i / c8) * (o.cin + 8) + (i % c8) * 8) = v[u];
            }
        }
    }
    __syncthreads();
    TL_STAMP(1);

    int cur = 0;
    for (int oi = 0; oi < a.count; ++oi) {
        const TailOp o = ops_sh[oi];
        const half_t* x = buf[cur];
        half_t* y = buf[cur ^ 1];
        half_t* yg = o.out ? o.out + (size_t)n * o.out_stride : nullptr;     // also materialised in HBM (pyramid feature)
        const int M = o.hout * o.wout;
        if (o.type == DN_OP_PW) {
            // work unit = (32-channel output tile, K slice). The layers have few output tiles and long K (512 -> 128 is four
            // tiles of 32 K-steps), and a wave's weight stream is a chain of dependent L2 round trips: splitting K over the
            // waves turns K/128 round trips into one. Slices meet in LDS (fp32 partial tiles).
            const int K = o.cin, N = o.cout;
            const int KS = (K + 15) >> 4;
            const int ntiles = (N + 31) >> 5;
            int nsl = NWV / ntiles;
            const int maxsl = (KS + 7) >> 3;
            if (nsl > maxsl) nsl = maxsl;
            if (nsl < 1) nsl = 1;
            if (nsl > PART_SLICES) nsl = PART_SLICES;
            while (nsl & (nsl - 1)) nsl &= nsl - 1;                 // power of two: divides the wave count, rounds stay tile-aligned
            const int ksl = (KS + nsl - 1) / nsl;                   // K-steps per slice
            const half_t* wrow0 = a.weights + o.w_off;
            const float* bias = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.weights) + o.b_off);
            for (int u0 = 0; u0 < ntiles * nsl; u0 += NWV) {
                const int unit = u0 + wave;
                const bool live = unit < ntiles * nsl;
                const int ct = live ? unit / nsl : 0, sl = live ? unit - ct * nsl : 0;
                const int ks_begin = sl * ksl, ks_end = min(KS, ks_begin + ksl);
                const half_t* wfrag = wrow0 + (size_t)ct * KS * 512 + lane * 8;      // fragment-major copy: 1 KB per (tile, K step)
                floatx16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
                half8 wa[8], wb[8];
                auto load8 = [&](half8 (&dst)[8], int ks0) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        // wave-uniform condition only (a per-lane predicate costs a branch and a wait per load); rows beyond N are
                        // clamped -- their outputs are never stored
                        half8 w = {0, 0, 0, 0, 0, 0, 0, 0};
                        if (live && ks0 + u < ks_end) w = *reinterpret_cast<const half8*>(wfrag + (size_t)(ks0 + u) * 512);
                        dst[u] = w;
                    }
                };
                auto mul8 = [&](const half8 (&w)[8], int ks0) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (ks0 + u < ks_end) {
                            // pixel rows beyond M are clamped (their outputs are never stored): unconditional LDS read
                            const half8 xf = *reinterpret_cast<const half8*>(x + (size_t)min(r, M - 1) * (K + 8) + (ks0 + u) * 16 + hh * 8);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[u], xf, acc, 0, 0, 0);
                        }
                    }
                };
                float4 bq[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = ct * 32 + 8 * g + 4 * hh;
                    bq[g] = (c < N) ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);     // with the weights, not after the MFMAs
                }
                load8(wa, ks_begin);
                for (int ks0 = ks_begin; ks0 < ks_end; ks0 += 16) {
                    if (ks0 + 8 < ks_end) load8(wb, ks0 + 8);
                    mul8(wa, ks0);
                    if (ks0 + 16 < ks_end) load8(wa, ks0 + 16);
                    if (ks0 + 8 < ks_end) mul8(wb, ks0 + 8);
                }
                // lane = pixel r, registers 4g..4g+3 = channels ct*32 + 8g + 4hh .. +3
                if (nsl == 1) {
                    if (live && r < M) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int c = ct * 32 + 8 * g + 4 * hh;
                            if (c < N) {            // cout % 8 == 0
                                const float4 b = bq[g];
                                half4 hv;
                                float t4[4] = {acc[4 * g + 0] + b.x, acc[4 * g + 1] + b.y, acc[4 * g + 2] + b.z, acc[4 * g + 3] + b.w};
                                dn_act_n<float[4], 4>(t4, o.act);
                                hv[0] = (half_t)t4[0]; hv[1] = (half_t)t4[1]; hv[2] = (half_t)t4[2]; hv[3] = (half_t)t4[3];
                                *reinterpret_cast<half4*>(y + (size_t)r * (N + 8) + c) = hv;
                                if (yg) *reinterpret_cast<half4*>(yg + (size_t)r * N + c) = hv;
                            }
                        }
                    }
                } else {
                    // partial tile -> LDS [wave][16][64 lanes] (conflict-free), then the slices of a tile are added in order
                    float* pw = part + (size_t)wave * 1024;
#pragma unroll
                    for (int e = 0; e < 16; ++e) pw[e * 64 + lane] = acc[e];
                    __syncthreads();
                    for (int idx = tid; idx < NWV / nsl * 1024; idx += TT) {
                        const int tl = idx >> 10, e = (idx >> 6) & 15, ln = idx & 63;      // local tile, register, lane
                        const int ct2 = (u0 / nsl) + tl;
                        const int px = ln & 31, c = ct2 * 32 + 8 * (e >> 2) + 4 * (ln >> 5) + (e & 3);
                        if (ct2 < ntiles && px < M && c < N) {
                            float v = bias[c];
                            for (int q = 0; q < nsl; ++q) v += part[(size_t)(tl * nsl + q) * 1024 + e * 64 + ln];
                            const half_t hv = (half_t)dn_act(v, o.act);
                            y[(size_t)px * (N + 8) + c] = hv;
                            if (yg) yg[(size_t)px * N + c] = hv;
                        }
                    }
                    __syncthreads();
                }
            }
        } else {        // DN_OP_DW
            const int C = o.cin, C8 = C >> 3, KK = o.k;
            const half_t* wts = a.weights + o.w_off;        // [k*k][c]
            const float* bias = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.weights) + o.b_off);
            for (int item = tid; item < M * C8; item += TT) {
                const int cg = item % C8, p = item / C8;
                const int oy = p / o.wout, ox = p - oy * o.wout;
                // all tap weights and the bias are requested together, before the first use (a load inside the tap loop would
                // be one exposed round trip per tap)
                uint4 wv[9];
                if (KK == 3) {
#pragma unroll
                    for (int tp = 0; tp < 9; ++tp) wv[tp] = *reinterpret_cast<const uint4*>(wts + (size_t)tp * C + cg * 8);
                }
                float acc[8];
                {
                    const float4 b0 = *reinterpret_cast<const float4*>(bias + cg * 8), b1 = *reinterpret_cast<const float4*>(bias + cg * 8 + 4);
                    acc[0] = b0.x; acc[1] = b0.y; acc[2] = b0.z; acc[3] = b0.w; acc[4] = b1.x; acc[5] = b1.y; acc[6] = b1.z; acc[7] = b1.w;
                }
                if (KK == 3) {
#pragma unroll
                    for (int tp = 0; tp < 9; ++tp) {
                        const int ky = tp / 3, kx = tp - ky * 3;
                        const int iy = oy * o.stride - o.pad + ky, ix = ox * o.stride - o.pad + kx;
                        if (iy >= 0 && iy < o.hin && ix >= 0 && ix < o.win) {
                            const uint4 ev = *reinterpret_cast<const uint4*>(x + ((size_t)iy * o.win + ix) * (C + 8) + cg * 8);
                            fma_mix_h8(acc, ev, wv[tp]);
                        }
                    }
                } else {
                    for (int ky = 0; ky < KK; ++ky) {
                        const int iy = oy * o.stride - o.pad + ky;
                        if (iy < 0 || iy >= o.hin) continue;
                        for (int kx = 0; kx < KK; ++kx) {
                            const int ix = ox * o.stride - o.pad + kx;
                            if (ix < 0 || ix >= o.win) continue;
                            const uint4 ev = *reinterpret_cast<const uint4*>(x + ((size_t)iy * o.win + ix) * (C + 8) + cg * 8);
                            const uint4 wq = *reinterpret_cast<const uint4*>(wts + (size_t)(ky * KK + kx) * C + cg * 8);
                            fma_mix_h8(acc, ev, wq);
                        }
                    }
                }
                half8 hv;
                dn_act_n<float[8], 8>(acc, o.act);
#pragma unroll
                for (int e = 0; e < 8; ++e) hv[e] = (half_t)acc[e];
                *reinterpret_cast<half8*>(y + (size_t)p * (C + 8) + cg * 8) = hv;
                if (yg) *reinterpret_cast<half8*>(yg + (size_t)p * C + cg * 8) = hv;
            }
        }
        __syncthreads();
        TL_STAMP(2 + oi);
        cur ^= 1;
    }
}

}  // namespace

bool tail_op_supported(const dn_op_desc& o, int hin, int win, int hout, int wout) {
    if (o.head || o.residual >= 0 || o.se >= 0 || o.pool >= 0) return false;
    if (o.type == DN_OP_PW) return hout * wout <= 32 && o.cin % 16 == 0 && o.cout % 8 == 0 && o.cin <= 2048 && o.cout <= 2048 && o.w2_off >= 0;
    if (o.type == DN_OP_DW) return hout * wout <= 32 && hin * win <= 128 && o.cin % 8 == 0 && o.dil == 1 && (o.k == 3 || o.k == 5);
    return false;
}

int launch_tail(const TailArgs& a0, int n, hipStream_t s) {
    TailArgs a = a0;
    DN_REQUIRE(a.count >= 1 && a.count <= TAIL_MAX_OPS, "tail: %d ops", a.count);
    size_t ping = (size_t)a.op[0].hin * a.op[0].win * (a.op[0].cin + 8), pong = 0;
    for (int i = 0; i < a.count; ++i) {
        const TailOp& o = a.op[i];
        const size_t out = (size_t)o.hout * o.wout * (o.cout + 8);
        size_t& dst = (i & 1) ? ping : pong;        // op i reads buf[i & 1] and writes the other one
        if (out > dst) dst = out;
    }
    a.buf_halfs = (int)((ping + 7) & ~(size_t)7);
    a.buf2_halfs = (int)((pong + 7) & ~(size_t)7);
    const size_t lds = ((size_t)a.buf_halfs + a.buf2_halfs) * sizeof(half_t) + (size_t)NWV * 1024 * sizeof(float);
    DN_REQUIRE(lds <= 156 * 1024, "tail: activations need %zu B of LDS", lds);
    DN_HIP_CHECK(dn_allow_big_lds(reinterpret_cast<const void*>(tail_kernel), 156 * 1024));     // + the static op table
    a.stamps = g_tail_stamps;
    dn_note_kernel("tail_kernel");
    hipLaunchKernelGGL(tail_kernel, dim3(a.xq > 0 ? 8 * a.xq : n), dim3(TT), lds, s, a, n);
    return DN_OK;
}
