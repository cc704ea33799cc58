// The 10 x 10 stage of the MobileNetV3 backbone in ONE launch: a GROUP of workgroups per image, channel-split, exchanging through memory.
//
// reference ops replaced (mobilenetv3.py:61-99 InvertedResidual with SqueezeExcitation :22-37; the block table :198-214, rows 13 - 15 with
// reduce_divider 2; ssd_mobilenetv3.py:104-108 feature taps): the projection that closes the stride-2 block (672 -> 80, SE-scaled), the
// two 80 -> 480 -> 80 blocks with 5x5 depthwise + squeeze-excitation + residual, and the 80 -> 480 expansion that is pyramid feature 1.
// As launches: pw_xs | (pw_direct, dw_kernel, se_fc8, pw_xs) x 2 | pw_direct = 10 dependent launches of 11 - 16 us each for 6 400 pixels of
// work per 64 images (130 us of a 0.85 ms forward): every one of them is a launch + two exposed memory round trips. One workgroup per
// image with the map resident in LDS was built in round 2 (trunk.hip) and lost: a whole image on ONE compute unit turns every phase into a
// latency chain and idles three quarters of the chip.
//
// CDNA4 mapping. An image is worked on by COOP_G = 4 workgroups of 512 threads (256 workgroups for 64 images: one per compute unit), each
// owning a SLICE OF THE EXPANDED CHANNELS (whole 32-channel MFMA tiles: 480 = 15 tiles -> 4 + 4 + 4 + 3). Per block:
//   expand     x[M][80] (LDS, all of it in every workgroup) x W1[:, slice] on MFMA (A = fragment-major weights straight from L2), bias +
//              hard-swish, fp16 -> E[M][slice] in LDS. The expansion is channel-parallel: no exchange.
//   depthwise  5x5 over E in LDS -> D[M][slice] (weights of the slice staged in LDS, taps in (ky, kx) order, v_fma_mix_f32): depthwise is
//              per channel: no exchange. Then the slice's channel means.
//   exchange 1 the means (128 floats per workgroup) -> memory, group barrier, every workgroup reads all 480.
//   SE         fc1 (480 -> 120) by every workgroup (115 KB of fp16 weights through L2 each), fc2 only for the own slice; D scaled in place
//              with the `(half)(x * s)` rounding of the launch-per-layer path.
//   project    D[M][slice] x W3[slice, :] = a K-SLICE of the projection: fp32 partial sums [M][80] -> memory.
//   exchange 2 group barrier; every workgroup adds the four partial tiles in slice order (fixed: deterministic), + bias + residual, rounds to
//              fp16: the next block's x, again complete in every workgroup's LDS.
// Two exchanges per block, one for the leading projection, none for the final expansion (each workgroup writes its channel slice of
// feature 1). A group barrier = agent-scope write-through stores (8-byte relaxed atomics: no cache-flushing fence, depthwise.hip's
// hand-over spelled out), every storing thread's `s_waitcnt vmcnt(0)`, a workgroup barrier, ONE relaxed atomic increment of the step's
// counter, and a polling loop of thread 0 with s_sleep; the data is read back with agent-scope atomic loads. The four workgroups of an
// image have consecutive positions in their XCD's dispatch order (flat index = 8 * (4 * image_in_group + slice) + XCD), so a group
// is dispatched together and normally shares an L2; correctness does not depend on it. Groups are independent, a group's members never wait
// for another group, and a kernel's resident workgroups are a prefix of its dispatch order: at most one group per launch can be waiting for
// members that have no slot yet, and every other resident group runs to completion and frees slots -- no deadlock with other launches on the chip.
// Counters are zeroed by the stem launch of every forward (the squeeze-excitation counters' range).
//
// Rounding points are those of the separate launches (fp16 activations, fp32 accumulation, fp16 SE product); sums are taken in another order
// (K-slices of the projection, slice partials of the means), so results agree with that path to fp32 rounding, not bit for bit.
#include <algorithm>

#include "common.h"

#ifdef DN_DEV_STAMPS
static long long* g_coop_stamps = nullptr;     // dev build only (tools/probe_coop.py): per-workgroup stamps [wg][32]
extern "C" __attribute__((visibility("default"))) void dn_debug_coop_stamps(void* dev_ptr) { g_coop_stamps = (long long*)dev_ptr; }
#define CO_STAMP(k) do { if (a.stamps && threadIdx.x == 0) a.stamps[blockIdx.x * 32 + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
constexpr long long* g_coop_stamps = nullptr;
#define CO_STAMP(k) do { } while (0)
#endif

namespace {

constexpr int CT = 512;              // threads per workgroup
constexpr int CW = CT / 64;          // waves

__device__ __forceinline__ void st8(float* p, float a, float b) {
    unsigned long long v = ((unsigned long long)__float_as_uint(b) << 32) | (unsigned long long)__float_as_uint(a);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 ld8(const float* p) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
}

// group barrier `step` of this image: every member has written (write-through) what the others read next
__device__ __forceinline__ void group_barrier(unsigned* ctr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's write-through stores are acknowledged ...
    __syncthreads();                                      // ... all of the workgroup's are, before its arrival is published
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)COOP_G) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
}

// tiles [t0, t0 + nt) of `total` 32-channel tiles for slice g of COOP_G
__device__ __forceinline__ void slice_of(int total, int g, int& t0, int& nt) {
    const int base = total / COOP_G, rem = total % COOP_G;
    t0 = g * base + min(g, rem);
    nt = base + (g < rem ? 1 : 0);
}

// E[px][c - 32 t0] = act(x[px][:] . W[c][:] + b[c]) for the channel tiles [t0, t0 + nt) of a 1x1 conv; x: LDS [M][cin + 8]; optionally also to memory.
// A wave owns the (channel tile, pixel tile) units wave, wave + 8 (at most two: <= 4 x 4 units); the A fragments of BOTH are requested before the
// first use -- one exposed L2 round trip per call instead of one per unit and batch (the launch is a chain of such round trips).
constexpr int EXP_KS = 8;            // K steps of an expansion (cin <= 128)
struct ExpA { half8 f[2][EXP_KS]; };
template <int NK>
__device__ __forceinline__ void expand_prefetch(ExpA& A, const CoopPw& w, const int M, const int t0, const int nt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int KS = w.cin >> 4, npt = (M + 31) >> 5, nunits = nt * npt;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int unit = min(wave + CW * q, nunits - 1), ct = t0 + unit / npt;
        const half_t* wf = w.wfrag + ((size_t)ct * KS) * 512 + lane * 8;
#pragma unroll
        for (int u = 0; u < NK; ++u) A.f[q][u] = *reinterpret_cast<const half8*>(wf + (size_t)min(u, KS - 1) * 512);
    }
}
template <int NK>
__device__ __forceinline__ void expand_slice(const ExpA& A, const CoopPw& w, const half_t* __restrict__ x, half_t* __restrict__ e, const int est, const int M, const int t0,
                                             const int nt, half_t* __restrict__ gout) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    const int KS = w.cin >> 4, xst = w.cin + 8, npt = (M + 31) >> 5, nunits = nt * npt;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int unit = wave + CW * q;
        if (unit >= nunits) break;
        const int ctl = unit / npt, pt = unit - ctl * npt, ct = t0 + ctl;
        const half_t* xr = x + (size_t)min(pt * 32 + r, M - 1) * xst + hh * 8;
        floatx16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int u = 0; u < NK; ++u)
            if (u < KS) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.f[q][u], *reinterpret_cast<const half8*>(xr + u * 16), acc, 0, 0, 0);
        const int px = pt * 32 + r;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int c = ct * 32 + 8 * gq + 4 * hh;
            const float4 b = *reinterpret_cast<const float4*>(w.bias + min(c, w.cout - 4));
            float t4[4] = {acc[4 * gq] + b.x, acc[4 * gq + 1] + b.y, acc[4 * gq + 2] + b.z, acc[4 * gq + 3] + b.w};
            dn_act_n<float[4], 4>(t4, w.act);
            half4 hv;
            hv[0] = (half_t)t4[0]; hv[1] = (half_t)t4[1]; hv[2] = (half_t)t4[2]; hv[3] = (half_t)t4[3];
            if (px < M && c < w.cout) {
                *reinterpret_cast<half4*>(e + (size_t)px * est + (c - t0 * 32)) = hv;
                if (gout) *reinterpret_cast<half4*>(gout + (size_t)px * w.cout + c) = hv;
            }
        }
    }
}

// partial[px][co] = sum over the K slice [32 t0, 32 (t0 + nt)) of d[px][k] * W[co][k], fp32, to memory (write-through); d: LDS [M][dst].
// Units (output-channel tile, pixel tile): at most 2 per wave (<= 4 x 4), every A fragment of both requested up front.
constexpr int PRJ_KS = 12;           // K steps of a projection slice (<= 192 channels)
struct PrjA { half8 f[2][PRJ_KS]; };
template <int NK>
__device__ __forceinline__ void project_prefetch(PrjA& A, const CoopPw& w, const int M, const int t0, const int nt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int KS = w.cin >> 4, npt = (M + 31) >> 5, nct = (w.cout + 31) >> 5, ks_lo = 2 * t0, nks = 2 * nt, nunits = nct * npt;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int unit = min(wave + CW * q, nunits - 1), ct = unit / npt;
        const half_t* wf = w.wfrag + ((size_t)ct * KS + ks_lo) * 512 + lane * 8;
#pragma unroll
        for (int u = 0; u < NK; ++u) A.f[q][u] = *reinterpret_cast<const half8*>(wf + (size_t)min(u, nks - 1) * 512);
    }
}
template <int NK>
__device__ __forceinline__ void project_partial(const PrjA& A, const CoopPw& w, const half_t* __restrict__ d, const int dst, const int M, const int t0, const int nt,
                                                float* __restrict__ part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    const int npt = (M + 31) >> 5, nct = (w.cout + 31) >> 5, nks = 2 * nt, nunits = nct * npt;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int unit = wave + CW * q;
        if (unit >= nunits) break;
        const int ct = unit / npt, pt = unit - ct * npt;
        const half_t* dr = d + (size_t)min(pt * 32 + r, M - 1) * dst + hh * 8;
        floatx16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int u = 0; u < NK; ++u)
            if (u < nks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.f[q][u], *reinterpret_cast<const half8*>(dr + u * 16), acc, 0, 0, 0);
        const int px = pt * 32 + r;
        if (px < M) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int c = ct * 32 + 8 * gq + 4 * hh;
                if (c < w.cout) {         // cout % 4 == 0
                    st8(part + (size_t)px * w.cout + c, acc[4 * gq], acc[4 * gq + 1]);
                    st8(part + (size_t)px * w.cout + c + 2, acc[4 * gq + 2], acc[4 * gq + 3]);
                }
            }
        }
    }
}

// x[px][c] = fp16(bias[c] + part_0 + part_1 + part_2 + part_3 (+ x[px][c])): the four K-slice partials in slice order. Every load of a thread's
// (up to RED_IT) items is requested before the first use: the partials come from memory (agent-scope loads), one round trip instead of one per item.
constexpr int RED_IT = 5;            // items of four channels per thread: 128 pixels x 80 channels / 4 / 512 threads
__device__ __forceinline__ void reduce_partials(const CoopPw& w, const float* __restrict__ part_img, const size_t part_stride, half_t* __restrict__ x, const int xst,
                                                const int M, const bool residual) {
    const int c4n = w.cout >> 2, nitems = M * c4n;
    float2 p[RED_IT][COOP_G][2];
#pragma unroll
    for (int it = 0; it < RED_IT; ++it) {
        const int item = min((int)threadIdx.x + CT * it, nitems - 1);
        const int px = item / c4n, c = (item - px * c4n) * 4;
#pragma unroll
        for (int q = 0; q < COOP_G; ++q) {
            const float* src = part_img + (size_t)q * part_stride + (size_t)px * w.cout + c;
            p[it][q][0] = ld8(src);
            p[it][q][1] = ld8(src + 2);
        }
    }
#pragma unroll
    for (int it = 0; it < RED_IT; ++it) {
        const int item = (int)threadIdx.x + CT * it;
        if (item >= nitems) break;
        const int px = item / c4n, c = (item - px * c4n) * 4;
        const float4 b = *reinterpret_cast<const float4*>(w.bias + c);
        float v[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int q = 0; q < COOP_G; ++q) { v[0] += p[it][q][0].x; v[1] += p[it][q][0].y; v[2] += p[it][q][1].x; v[3] += p[it][q][1].y; }
        half4* xp = reinterpret_cast<half4*>(x + (size_t)px * xst + c);
        if (residual) {
            const half4 old = *xp;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)old[e];
        }
        half4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[e] = (half_t)dn_act(v[e], w.act);
        *xp = hv;
    }
}

template <int EK>      // K steps of the expansions: 5 (80 input channels) or 8
__global__ __launch_bounds__(CT) void coop_kernel(CoopArgs a) {
    extern __shared__ __attribute__((aligned(16))) half_t cl[];
    // workgroup -> (image, slice): the members of a group are consecutive in their XCD's dispatch order
    int img, g;
    {
        const int flat = blockIdx.x;
        if (a.xq > 0) {
            const int xcd = flat & 7, j = flat >> 3;
            img = xcd * a.xq + j / COOP_G;
            g = j % COOP_G;
        } else {
            img = flat / COOP_G;
            g = flat % COOP_G;
        }
        if (img >= a.n) return;             // (a whole group leaves together)
    }
    const int tid = threadIdx.x;
    const int M = a.H * a.W;
    const int xst = a.cx + 8, bst = a.slice_max + 8;
    half_t* const X = cl;                                    // [M][cx + 8]        block input / output, complete
    half_t* const E = X + (size_t)M * xst;                   // [M][slice_max + 8] expanded slice
    half_t* const D = E + (size_t)M * bst;                   // [M][slice_max + 8] depthwise output slice (also the staged input of the leading projection)
    float* const fm = reinterpret_cast<float*>(D + (size_t)M * bst);      // [cmax] channel means of the image, then hidden / partial scratch
    float* const fh = fm + a.cmax;                           // [CW][sqmax] fc1 slices, [sqmax] hidden
    float* const fs = fh + (CW + 1) * a.sqmax;               // [CW][slice_max] fc2 slices -> [slice_max] scale
    half_t* const WD = reinterpret_cast<half_t*>(fs + (CW + 1) * a.slice_max);  // [k * k][slice_max] depthwise weights of the slice
    float* const means = a.scratch + (size_t)img * a.scratch_per_image;  // [cmax]
    float* const parts = means + a.cmax;                                 // [COOP_G][M][cx]
    const size_t part_stride = (size_t)M * a.cx;
    unsigned* const ctr = a.counters + img;
    int step = 0;
    CO_STAMP(0);

    // Weights never depend on activations: every weight fetch below is ISSUED one or two phases before its use (the fetches of a phase used to
    // sit at its start: the launch was a chain of ~25 exposed L2 round trips of 2 - 4 us; as launches the same chain costs a dispatch each).
    ExpA exA;
    PrjA pjA;
    // ---- leading projection: d0[img][M][cin] (memory) scaled by the squeeze-excitation vector s0[img][cin], K-slice g
    {
        int t0, nt;
        slice_of(a.p0.cin >> 5, g, t0, nt);
        project_prefetch<PRJ_KS>(pjA, a.p0, M, t0, nt);
        const int c0 = t0 * 32, cw = nt * 32, c8 = cw >> 3;
        const half_t* src = a.d0 + (size_t)img * M * a.p0.cin + c0;
        const float* sc = a.s0 + (size_t)img * a.p0.cin + c0;
        for (int item = tid; item < M * c8; item += CT) {
            const int px = item / c8, cg = item - px * c8;
            const half8 v = *reinterpret_cast<const half8*>(src + (size_t)px * a.p0.cin + cg * 8);
            const float4 s_lo = *reinterpret_cast<const float4*>(sc + cg * 8), s_hi = *reinterpret_cast<const float4*>(sc + cg * 8 + 4);
            const float s8[8] = {s_lo.x, s_lo.y, s_lo.z, s_lo.w, s_hi.x, s_hi.y, s_hi.z, s_hi.w};
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)v[e] * s8[e]);
            *reinterpret_cast<half8*>(D + (size_t)px * bst + cg * 8) = o;
        }
        __syncthreads();
        CO_STAMP(1);
        project_partial<PRJ_KS>(pjA, a.p0, D, bst, M, t0, nt, parts + (size_t)g * part_stride);
        CO_STAMP(2);
        {   // the first block's expansion weights: in flight under the group barrier
            int e0, en;
            const CoopPw& nx = a.nblocks > 0 ? a.blk[0].ex : a.last;
            slice_of(nx.cout >> 5, g, e0, en);
            expand_prefetch<EK>(exA, nx, M, e0, en);
        }
        group_barrier(ctr + (size_t)(step++) * a.counter_stride);
        CO_STAMP(3);
        reduce_partials(a.p0, parts, part_stride, X, xst, M, false);
        __syncthreads();
        CO_STAMP(4);
    }

    for (int bi = 0; bi < a.nblocks; ++bi) {
        const CoopBlock& B = a.blk[bi];
        int t0, nt;
        slice_of(B.ex.cout >> 5, g, t0, nt);
        const int c0 = t0 * 32, cw = nt * 32, c8 = cw >> 3, KK = B.k * B.k;
        // depthwise weights of the slice -> LDS (requested first: in flight under the expansion)
        for (int i = tid; i < KK * c8; i += CT) {
            const int tp = i / c8, cg = i - tp * c8;
            *reinterpret_cast<uint4*>(WD + (size_t)tp * a.slice_max + cg * 8) = *reinterpret_cast<const uint4*>(B.wd + (size_t)tp * B.ex.cout + c0 + cg * 8);
        }
        expand_slice<EK>(exA, B.ex, X, E, bst, M, t0, nt, nullptr);
        __syncthreads();
        CO_STAMP(5 + 10 * bi);
        // depthwise k x k, stride 1: item = (pixel, 8-channel group); taps in (ky, kx) order like dw_kernel
        for (int item = tid; item < M * c8; item += CT) {
            const int cg = item % c8, px = item / c8;
            const int oy = px / a.W, ox = px - oy * a.W;
            float acc[8];
            {
                const float4 b0 = *reinterpret_cast<const float4*>(B.bd + c0 + cg * 8), b1 = *reinterpret_cast<const float4*>(B.bd + c0 + cg * 8 + 4);
                acc[0] = b0.x; acc[1] = b0.y; acc[2] = b0.z; acc[3] = b0.w; acc[4] = b1.x; acc[5] = b1.y; acc[6] = b1.z; acc[7] = b1.w;
            }
            // the taps inside the map only (zero padding contributes nothing), in (ky, kx) order; a row's taps are read before its multiply-adds
            const int ky0 = max(0, B.pad - oy), ky1 = min(B.k, a.H + B.pad - oy), kx0 = max(0, B.pad - ox), kx1 = min(B.k, a.W + B.pad - ox);
            for (int ky = ky0; ky < ky1; ++ky) {
                const half_t* er = E + (ptrdiff_t)((oy - B.pad + ky) * a.W + (ox - B.pad)) * bst + cg * 8;
                const half_t* wr = WD + (size_t)(ky * B.k) * a.slice_max + cg * 8;
                uint4 ev[5], wv[5];
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const int kx = min(kx0 + u, kx1 - 1);
                    ev[u] = *reinterpret_cast<const uint4*>(er + (size_t)kx * bst);
                    wv[u] = *reinterpret_cast<const uint4*>(wr + (size_t)kx * a.slice_max);
                }
#pragma unroll
                for (int u = 0; u < 5; ++u)
                    if (kx0 + u < kx1) fma_mix_h8(acc, ev[u], wv[u]);
            }
            dn_act_n<float[8], 8>(acc, B.act_dw);
            half8 hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = (half_t)acc[e];
            *reinterpret_cast<half8*>(D + (size_t)px * bst + cg * 8) = hv;
        }
        // squeeze-excitation weights of this thread: fc1 (wave = slice of the channels, lane = output pair), fc2 (thread = channel pair of the slice x
        // slice of the hidden units): requested now, used behind the exchange of the means
        const int C = B.ex.cout, sq = B.sq, sq2 = sq >> 1;
        const int wv1 = tid >> 6, jp = tid & 63;
        const int per = (C + CW - 1) / CW, k0 = wv1 * per, k1 = min(C, k0 + per);
        constexpr int FC1N = 64, FC2N = 16;
        half2_t w1r[FC1N];
#pragma unroll
        for (int u = 0; u < FC1N; ++u) w1r[u] = *reinterpret_cast<const half2_t*>(B.w1t + (size_t)min(k0 + u, k1 - 1) * sq + 2 * min(jp, sq2 - 1));
        const int cp = tid & 63, js = tid >> 6;                 // fc2: 64 channel pairs (slices of <= 128 channels) x 8 slices of the hidden units
        const int jper = (sq + CW - 1) / CW, j0 = js * jper, j1 = min(sq, j0 + jper);
        half2_t w2r[FC2N];
#pragma unroll
        for (int u = 0; u < FC2N; ++u) w2r[u] = *reinterpret_cast<const half2_t*>(B.w2t + (size_t)min(j0 + u, max(j1, j0 + 1) - 1) * C + c0 + 2 * min(cp, (cw >> 1) - 1));
        __syncthreads();
        CO_STAMP(6 + 10 * bi);
        // channel means of the slice (pixels in order: deterministic) -> memory -> all means
        if (tid < cw) {
            float sacc = 0.f;
            for (int px = 0; px < M; ++px) sacc += (float)D[(size_t)px * bst + tid];
            __hip_atomic_store(means + c0 + tid, sacc * B.inv_pixels, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        CO_STAMP(7 + 10 * bi);
        group_barrier(ctr + (size_t)(step++) * a.counter_stride);
        CO_STAMP(8 + 10 * bi);
        for (int c = tid; c < C; c += CT) fm[c] = __hip_atomic_load(means + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        // squeeze-excitation: fc1 + ReLU (all of it, every workgroup), fc2 + hard-sigmoid for the slice (mobilenetv3.py:31-36)
        {
            float h0 = 0.f, h1 = 0.f;
#pragma unroll
            for (int u = 0; u < FC1N; ++u)
                if (k0 + u < k1) {
                    h0 += fm[k0 + u] * (float)w1r[u][0];
                    h1 += fm[k0 + u] * (float)w1r[u][1];
                }
            if (jp < sq2) { fh[wv1 * a.sqmax + 2 * jp] = h0; fh[wv1 * a.sqmax + 2 * jp + 1] = h1; }
            // this block's projection weights (slices of <= 128 channels), now that the fc1 registers are free: in flight under fc2 and the scaling
            project_prefetch<8>(pjA, B.pj, M, t0, nt);
            __syncthreads();
            if (tid < sq) {
                float v = B.b1[tid];
                for (int q = 0; q < CW; ++q) v += fh[q * a.sqmax + tid];
                fh[CW * a.sqmax + tid] = fmaxf(v, 0.f);
            }
            __syncthreads();
            CO_STAMP(9 + 10 * bi);
            const float* hid = fh + CW * a.sqmax;
            float v0 = 0.f, v1 = 0.f;
#pragma unroll
            for (int u = 0; u < FC2N; ++u)
                if (j0 + u < j1) {
                    v0 += hid[j0 + u] * (float)w2r[u][0];
                    v1 += hid[j0 + u] * (float)w2r[u][1];
                }
            if (2 * cp < cw) { fs[js * a.slice_max + 2 * cp] = v0; fs[js * a.slice_max + 2 * cp + 1] = v1; }
            __syncthreads();
            if (tid < cw) {
                float v = B.b2[c0 + tid];
                for (int q = 0; q < CW; ++q) v += fs[q * a.slice_max + tid];
                fs[CW * a.slice_max + tid] = dn_relu6(v + 3.f) * (1.f / 6.f);
            }
            __syncthreads();
            const float* scl = fs + CW * a.slice_max;
            for (int item = tid; item < M * c8; item += CT) {
                const int px = item / c8, cg = item - px * c8;
                half8* dp = reinterpret_cast<half8*>(D + (size_t)px * bst + cg * 8);
                half8 v = *dp;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * scl[cg * 8 + e]);
                *dp = v;
            }
            __syncthreads();
        }
        CO_STAMP(10 + 10 * bi);
        project_partial<8>(pjA, B.pj, D, bst, M, t0, nt, parts + (size_t)g * part_stride);
        CO_STAMP(11 + 10 * bi);
        {   // the next expansion's weights: in flight under the group barrier and the reduction
            int e0, en;
            const CoopPw& nx = bi + 1 < a.nblocks ? a.blk[bi + 1].ex : a.last;
            slice_of(nx.cout >> 5, g, e0, en);
            expand_prefetch<EK>(exA, nx, M, e0, en);
        }
        group_barrier(ctr + (size_t)(step++) * a.counter_stride);
        CO_STAMP(12 + 10 * bi);
        reduce_partials(B.pj, parts, part_stride, X, xst, M, B.has_res != 0);
        __syncthreads();
        CO_STAMP(13 + 10 * bi);
    }

    // ---- final expansion: every workgroup writes its channel slice of the feature map
    {
        int t0, nt;
        slice_of(a.last.cout >> 5, g, t0, nt);
        expand_slice<EK>(exA, a.last, X, E, bst, M, t0, nt, a.out_last + (size_t)img * M * a.last.cout);
    }
    CO_STAMP(30);
}

}  // namespace

bool coop_supported(const CoopArgs& a) {
    const int M = a.H * a.W;
    if (M < 1 || M > 128 || a.nblocks < 0 || a.nblocks > COOP_MAX_BLOCKS) return false;
    auto pw_ok = [](const CoopPw& w) { return w.wfrag && w.bias && w.cin % 16 == 0 && w.cout % 8 == 0; };
    if (!pw_ok(a.p0) || !pw_ok(a.last) || a.p0.cin % 32 != 0 || (a.p0.cin >> 5) < COOP_G || a.last.cout % 32 != 0 || (a.last.cout >> 5) < COOP_G) return false;
    if (a.p0.cout != a.cx || a.last.cin != a.cx || a.cx % 16 != 0) return false;
    for (int i = 0; i < a.nblocks; ++i) {
        const CoopBlock& b = a.blk[i];
        if (!pw_ok(b.ex) || !pw_ok(b.pj) || b.ex.cin != a.cx || b.pj.cout != a.cx || b.ex.cout != b.pj.cin || b.ex.cout % 32 != 0 || (b.ex.cout >> 5) < COOP_G) return false;
        if (b.ex.cin > 16 * EXP_KS || (b.k != 3 && b.k != 5) || b.pad != (b.k - 1) / 2 || b.sq % 2 != 0 || b.sq > 2 * 64 || b.sq > a.sqmax || b.ex.cout > a.cmax) return false;
        if (!b.wd || !b.bd || !b.w1t || !b.b1 || !b.w2t || !b.b2) return false;
        if (dn_cdiv(b.ex.cout / 32, COOP_G) > 4 || dn_cdiv(b.ex.cout, CW) > 64 || dn_cdiv(b.sq, CW) > 16) return false;      // register-resident weight slices
    }
    const int npt = (M + 31) / 32;
    int ex_tiles = dn_cdiv(a.last.cout / 32, COOP_G);           // channel tiles of an expansion slice: <= 2 units per wave
    for (int i = 0; i < a.nblocks; ++i) ex_tiles = std::max(ex_tiles, dn_cdiv(a.blk[i].ex.cout / 32, COOP_G));
    if (a.last.cin > 16 * EXP_KS || a.slice_max > 16 * PRJ_KS || M * (a.cx / 4) > RED_IT * CT || npt * ((a.cx + 31) / 32) > 2 * CW || npt * ex_tiles > 2 * CW) return false;
    return a.slice_max % 32 == 0 && a.slice_max <= CT && coop_lds_bytes(a) <= 156 * 1024;
}

size_t coop_lds_bytes(const CoopArgs& a) {
    const size_t M = (size_t)a.H * a.W;
    size_t b = (M * (a.cx + 8) + 2 * M * (a.slice_max + 8)) * sizeof(half_t);
    b += ((size_t)a.cmax + (size_t)(CW + 1) * a.sqmax + (size_t)(CW + 1) * a.slice_max) * sizeof(float);
    b += (size_t)25 * a.slice_max * sizeof(half_t);
    return (b + 15) & ~(size_t)15;
}

int launch_coop(const CoopArgs& a, hipStream_t s) {
    DN_REQUIRE(coop_supported(a), "cooperative 10 x 10 stage: configuration not supported");
    DN_REQUIRE(a.scratch && a.counters && a.d0 && a.s0 && a.out_last, "cooperative 10 x 10 stage: null buffer");
    const bool k5 = a.cx <= 80;
    DN_HIP_CHECK(dn_allow_big_lds(k5 ? reinterpret_cast<const void*>(coop_kernel<5>) : reinterpret_cast<const void*>(coop_kernel<8>), 156 * 1024));
    dn_note_kernel("coop_kernel");
    CoopArgs b = a;
    b.stamps = g_coop_stamps;
    const int slots = a.xq > 0 ? 8 * a.xq : a.n;
    if (k5) hipLaunchKernelGGL(coop_kernel<5>, dim3(slots * COOP_G), dim3(CT), coop_lds_bytes(a), s, b);
    else hipLaunchKernelGGL(coop_kernel<8>, dim3(slots * COOP_G), dim3(CT), coop_lds_bytes(a), s, b);
    return DN_OK;
}
