// Depthwise kxk convolution (+folded BN bias, activation), squeeze-excite pooling / FCs, stem conv.
//
// reference ops replaced: depthwise ConvBNActivation (mobilenetv3.py:81, ssd_mobilenetv3.py:31,48),
//   SqueezeExcitation (mobilenetv3.py:22-40), stem conv (mobilenetv3.py:141) with the input normalisation of
//   transform.py:129-138 applied on load.
// All of these are HBM-bound streaming kernels: NHWC fp16, 16-byte (8-channel) vectors per lane,
// channel index fastest across lanes so every wave instruction touches whole 128-B lines, fp32 accumulation.
#include <algorithm>

#include "common.h"

#ifdef DN_DEV_STAMPS
static long long* g_se_stamps = nullptr;     // dev build only (tools/probe_se.py)
extern "C" __attribute__((visibility("default"))) void dn_debug_se_stamps(void* dev_ptr) { g_se_stamps = (long long*)dev_ptr; }
#else
constexpr long long* g_se_stamps = nullptr;
#endif

namespace {

__device__ const uint4 g_dw_zero16 = {0u, 0u, 0u, 0u};      // what an out-of-image tap reads (pointer select instead of a predicated load)

// each thread: TW consecutive output pixels of one row x 8 channels; sliding input window kept in registers.
// Instruction diet (tools/valu.sh: these launches are VALU-issue bound): divisions by launch invariants through FastDiv, the image
// from a 2-D grid, unconditional loads (an out-of-image tap reads a zero line through a selected pointer -- a predicated load
// costs a branch and a conservative wait), one uniform activation switch per output pixel, pooled sums only in the POOL variant.
// Squeeze-excitation FCs (mobilenetv3.py:31-36: mean -> fc1 + ReLU -> fc2 + Hardsigmoid) computed by the LAST workgroup of an
// image's pooling launch instead of a launch of their own (a dependent launch of 32 workgroups cost 10 - 15 us, five times per
// chain). 256 threads; a thread owns 8 adjacent outputs (one 16-byte load per weight row) and a slice of the reduction axis, the
// slices are combined through LDS in a fixed order; loads in batches of 16 rows. sh: >= c + sq + 2048 floats.
__device__ __forceinline__ void dw_se_tail(const DwArgs& a, const int n, const int nblk, float* sh) {
    const int c = a.c, sq = a.se_sq, tid = threadIdx.x;
    float* mean = sh;               // [c]
    float* z = sh + c;              // [sq]
    float* part = z + sq;           // [256][8]
    {   // pooled mean: thread = channel (c <= 1024: up to 4 per thread), all partial rows of the image in a fixed order
        const float* p0 = a.pool + (size_t)n * nblk * c;
        for (int ic = tid; ic < c; ic += 256) {
            float t = 0.f;
            for (int b0 = 0; b0 < nblk; b0 += 16) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = __hip_atomic_load(&p0[(size_t)min(b0 + u, nblk - 1) * c + ic], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int u = 0; u < 16; ++u) t += (b0 + u < nblk) ? v[u] : 0.f;
            }
            mean[ic] = t * a.se_inv;
        }
    }
    __syncthreads();
    auto fc = [&](const half_t* wt, const float* bias, const float* in, int nin, int nout, float* out, bool hsig) {
        // out[j] = f(bias[j] + sum_i wt[i][j] * in[i]); thread = (8-output group og, K slice ks)
        const int OG = (nout + 7) >> 3;
        int ogp = 1;
        while (ogp < OG) ogp <<= 1;                     // groups padded to a power of two <= 256
        const int KS = 256 / ogp;
        const int og = tid & (ogp - 1), ks = tid / ogp;
        const int per = (nin + KS - 1) / KS, i0 = ks * per, i1 = min(nin, i0 + per);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (og < OG && i0 < i1) {
            const half_t* wp = wt + (size_t)og * 8;
            for (int ib = i0; ib < i1; ib += 16) {
                uint4 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const uint4*>(wp + (size_t)min(ib + u, i1 - 1) * nout);
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const float m = (ib + u < i1) ? in[ib + u] : 0.f;
                    const half8 h = *reinterpret_cast<const half8*>(&v[u]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += (float)h[e] * m;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) part[tid * 8 + e] = acc[e];
        __syncthreads();
        for (int j = tid; j < nout; j += 256) {
            float t = bias[j];
            const int g = j >> 3, e = j & 7;
            for (int q = 0; q < KS; ++q) t += part[(q * ogp + g) * 8 + e];
            out[j] = hsig ? fminf(fmaxf(t + 3.f, 0.f), 6.f) * (1.f / 6.f) : fmaxf(t, 0.f);
        }
        __syncthreads();
    };
    fc(a.se_w1t, a.se_b1, mean, c, sq, z, false);
    fc(a.se_w2t, a.se_b2, z, sq, c, a.se_scale + (size_t)n * c, true);
}

// The same for the SMALL squeeze-excitations (c <= 128, squeeze <= 32, <= 32 partial rows: the 40 x 40 blocks) with every global load of
// the tail -- the partial rows, both FC weight slices of this thread, the biases -- requested TOGETHER right after the ticket: the tail
// is the exposed end of its launch (it starts when the image's other workgroups are done), and the phase-by-phase form above made
// four dependent memory round trips of it (~10 us per launch, three launches per chain). Same sums in the same order: bit-identical.
__device__ __forceinline__ void dw_se_tail_small(const DwArgs& a, const int n, const int nblk, float* sh) {
    const int c = a.c, sq = a.se_sq, tid = threadIdx.x;
    float* mean = sh;               // [c]
    float* z = sh + c;              // [sq]
    float* part = z + sq;           // [256][8]
    // thread = (8-output group og, K slice ks) of each FC, as in dw_se_tail::fc; at these sizes a slice is at most two rows
    const int OG1 = (sq + 7) >> 3, OG2 = (c + 7) >> 3;
    int ogp1 = 1, ogp2 = 1;
    while (ogp1 < OG1) ogp1 <<= 1;
    while (ogp2 < OG2) ogp2 <<= 1;
    const int KS1 = 256 / ogp1, KS2 = 256 / ogp2;
    const int og1 = tid & (ogp1 - 1), ks1 = tid / ogp1, per1 = (c + KS1 - 1) / KS1, i0 = ks1 * per1, i1 = min(c, i0 + per1);
    const int og2 = tid & (ogp2 - 1), ks2 = tid / ogp2, per2 = (sq + KS2 - 1) / KS2, j0 = ks2 * per2, j1 = min(sq, j0 + per2);
    const bool act1 = og1 < OG1 && i0 < i1, act2 = og2 < OG2 && j0 < j1;
    float pv[32];
    {
        const float* p0 = a.pool + (size_t)n * nblk * c + min(tid, c - 1);
#pragma unroll
        for (int b = 0; b < 32; ++b) pv[b] = __hip_atomic_load(&p0[(size_t)min(b, nblk - 1) * c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint4 w1[2], w2[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        w1[u] = *reinterpret_cast<const uint4*>(a.se_w1t + (size_t)min(i0 + u, c - 1) * sq + min(og1, OG1 - 1) * 8);
        w2[u] = *reinterpret_cast<const uint4*>(a.se_w2t + (size_t)min(j0 + u, sq - 1) * c + min(og2, OG2 - 1) * 8);
    }
    const float b1v = a.se_b1[min(tid, sq - 1)], b2v = a.se_b2[min(tid, c - 1)];
    if (tid < c) {
        float t = 0.f;
#pragma unroll
        for (int b = 0; b < 32; ++b) t += (b < nblk) ? pv[b] : 0.f;
        mean[tid] = t * a.se_inv;
    }
    __syncthreads();
    auto fc = [&](const uint4 (&w)[2], bool act, int r0, int r1, const float* in, int ogp, int KS, int nout, float bias, float* out, bool hsig) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (act) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float m = (r0 + u < r1) ? in[r0 + u] : 0.f;
                const half8 h = *reinterpret_cast<const half8*>(&w[u]);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)h[e] * m;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) part[tid * 8 + e] = acc[e];
        __syncthreads();
        if (tid < nout) {
            float t = bias;
            const int g = tid >> 3, e = tid & 7;
            for (int q = 0; q < KS; ++q) t += part[(q * ogp + g) * 8 + e];
            out[tid] = hsig ? fminf(fmaxf(t + 3.f, 0.f), 6.f) * (1.f / 6.f) : fmaxf(t, 0.f);
        }
        __syncthreads();
    };
    fc(w1, act1, i0, i1, mean, ogp1, KS1, sq, b1v, z, false);
    fc(w2, act2, j0, j1, z, ogp2, KS2, c, b2v, a.se_scale + (size_t)n * c, true);
}
// (dw_se_tail_small keeps TWO weight rows per thread and FC: the K slices of its thread map must be at most two rows long. c <= 128 and
//  sq <= 32 imply that -- 256 / pow2(ceil(sq / 8)) >= 64 slices for fc1, >= 16 for fc2 -- and the condition is spelled out rather than implied.)
__device__ __forceinline__ bool dw_se_tail_is_small(int c, int sq, int nblk) {
    if (!(c <= 128 && sq <= 32 && nblk <= 32)) return false;
    int ogp1 = 1, ogp2 = 1;
    while (ogp1 < ((sq + 7) >> 3)) ogp1 <<= 1;
    while (ogp2 < ((c + 7) >> 3)) ogp2 <<= 1;
    const int ks1 = 256 / ogp1, ks2 = 256 / ogp2;
    return (c + ks1 - 1) / ks1 <= 2 && (sq + ks2 - 1) / ks2 <= 2;
}

// POOL: 0 = none, 1 = per-workgroup channel sums for the squeeze-excitation, 2 = sums + the FCs in the image's last workgroup
//   (dw_se_tail: its 16-row load batches need 114 registers, which capped EVERY pooling launch while the code was compiled into all of them).
// MODE bit 2: software-pipelined kernel rows (below); 0: the batched rows.
template <int K, int S, int TW, int POOL, int MODE>
__device__ __forceinline__ void dw_body(const DwArgs& a, const int bx, const int nblocks, const int n) {
    constexpr int NIN = (TW - 1) * S + K;
    extern __shared__ float red[];            // [256][8], only when pooling
    const unsigned C8 = a.c >> 3;
    const unsigned idx0 = bx * 256 + threadIdx.x;
    const unsigned q1 = fd_div(idx0, a.fd_c8);
    const unsigned cg = idx0 - q1 * C8;
    // strip position q1 -> (output row, strip). Plain: row-major. Row blocks (rb_log2 > 0, DN_DW_RB): q1 walks down RB rows before it moves
    // to the next strip, so the ~256 / C8 strip positions of a workgroup form a RB-row block whose input rows overlap -- the K-fold
    // vertical re-read of every input row is then served by the CU's own L1 instead of L2.
    unsigned oy, xs;
    if (a.rb_log2 == 0) {
        oy = fd_div(q1, a.fd_xs);
        xs = q1 - oy * a.fd_xs.d;
    } else {
        const unsigned blk = fd_div(q1, a.fd_rbxs), within = q1 - blk * a.fd_rbxs.d;
        xs = within >> a.rb_log2;
        oy = (blk << a.rb_log2) + (within & ((1u << a.rb_log2) - 1u));
    }
    const bool valid = (int)oy < a.ho;
    if (!valid && POOL == 0) return;
    const int ox0 = xs * TW;
    const unsigned c0 = cg * 8;
    float psum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (valid) {

    float acc[TW][8];
    {
        const float4 b0 = *reinterpret_cast<const float4*>(a.bias + c0);
        const float4 b1 = *reinterpret_cast<const float4*>(a.bias + c0 + 4);
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            acc[t][0] = b0.x; acc[t][1] = b0.y; acc[t][2] = b0.z; acc[t][3] = b0.w;
            acc[t][4] = b1.x; acc[t][5] = b1.y; acc[t][6] = b1.z; acc[t][7] = b1.w;
        }
    }
    const int ix0 = ox0 * S - a.pad;
    typedef const __attribute__((address_space(1))) half8* gp8;     // explicit global pointers: the selected pointer must not degrade to a flat load
    const gp8 zero = (gp8)(&g_dw_zero16);
    // RP kernel rows are requested together before their first use: hipcc otherwise waits for each row's loads before
    // issuing the next row's (vmcnt(0) per row), i.e. K dependent memory round trips per thread. 3x3 takes all rows at once;
    // 5x5 two at a time (all five would need 260 VGPRs of staging).
    constexpr int RP = (K == 3) ? 3 : 2;
    const half_t* const wbase = a.w + c0;
    const half_t* const xbase = a.x + (size_t)n * a.h * a.w_ * a.c + c0;
    if constexpr ((MODE & 4) != 0) {
        // Software-pipelined rows (MODE bit 2, DN_DW_PIPE): two register sets; row ky + 1 is requested BEFORE the multiply-adds of row ky
        // are issued, so only the first row's round trip is exposed (the batched form exposes ceil(K / RP) of them: three for 5x5).
        half8 wv[2][K];
        half8 xin[2][NIN];
        auto load_row = [&](const int ky, const int buf) {
            const int iy = (int)oy * S - a.pad + ky;
            const bool yok = iy >= 0 && iy < a.h;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) wv[buf][kx] = *reinterpret_cast<const half8*>(wbase + (unsigned)((ky * K + kx) * a.c));
            const half_t* rowp = xbase + (unsigned)((yok ? iy : 0) * a.w_ * a.c);
#pragma unroll
            for (int i = 0; i < NIN; ++i) {
                const int ix = ix0 + i;
                const bool ok = yok && ix >= 0 && ix < a.w_;
                xin[buf][i] = *(ok ? (gp8)(rowp + (unsigned)(ix * a.c)) : zero);
            }
        };
        load_row(0, 0);
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            if (ky + 1 < K) load_row(ky + 1, (ky + 1) & 1);
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
                    fma_mix_h8(acc[t], *reinterpret_cast<const uint4*>(&xin[ky & 1][t * S + kx]), *reinterpret_cast<const uint4*>(&wv[ky & 1][kx]));
        }
    } else {
#pragma unroll
    for (int ky0 = 0; ky0 < K; ky0 += RP) {
        half8 wv[RP][K];
        half8 xin[RP][NIN];
#pragma unroll
        for (int rr = 0; rr < RP; ++rr) {
            const int ky = ky0 + rr;
            if (ky >= K) continue;
            const int iy = (int)oy * S - a.pad + ky;
            const bool yok = iy >= 0 && iy < a.h;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) wv[rr][kx] = *reinterpret_cast<const half8*>(wbase + (unsigned)((ky * K + kx) * a.c));
            const half_t* rowp = xbase + (unsigned)((yok ? iy : 0) * a.w_ * a.c);
#pragma unroll
            for (int i = 0; i < NIN; ++i) {
                const int ix = ix0 + i;
                const bool ok = yok && ix >= 0 && ix < a.w_;
                xin[rr][i] = *(ok ? (gp8)(rowp + (unsigned)(ix * a.c)) : zero);
            }
        }
#pragma unroll
        for (int rr = 0; rr < RP; ++rr) {
            if (ky0 + rr >= K) continue;
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
                    fma_mix_h8(acc[t], *reinterpret_cast<const uint4*>(&xin[rr][t * S + kx]), *reinterpret_cast<const uint4*>(&wv[rr][kx]));
        }
    }
    }
    half_t* orow = a.out + ((size_t)(n * a.ho + oy) * a.wo) * a.c + c0;
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        if (ox0 + t >= a.wo) break;
        dn_act_n<float[8], 8>(acc[t], a.act);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if constexpr (POOL != 0) psum[e] += acc[t][e];
            o[e] = (half_t)acc[t][e];
        }
        *reinterpret_cast<half8*>(orow + (unsigned)((ox0 + t) * a.c)) = o;
    }
    }   // valid
    if constexpr (POOL != 0) {
        // SE squeeze (mobilenetv3.py:32 adaptive_avg_pool2d) fused as deterministic per-workgroup partial sums:
        // threads with equal channel group sit C8 apart; thread t < C8 adds them in a fixed order.
#pragma unroll
        for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = psum[e];
        __syncthreads();
        if (threadIdx.x < C8) {
            const unsigned i1 = bx * 256 + threadIdx.x;
            const unsigned cgp = i1 - fd_div(i1, a.fd_c8) * C8;
            float t8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (unsigned u = threadIdx.x; u < 256; u += C8)
#pragma unroll
                for (int e = 0; e < 8; ++e) t8[e] += red[u * 8 + e];
            float* dst = a.pool + ((size_t)n * nblocks + bx) * a.c + cgp * 8;
            if constexpr (POOL == 2) {
                // published for the last workgroup of the image (below): device-scope atomic stores (sc1) write through to the
                // coherence point themselves, so no cache-flushing fence is needed to make them visible ...
#pragma unroll
                for (int e = 0; e < 8; ++e) __hip_atomic_store(&dst[e], t8[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // ... but THIS thread must see them acknowledged before the barrier that precedes the ticket. A workgroup-scope fence
                // compiles to NO wait here (non-tgsplit mode: the waves of a workgroup share an L1; ISA before this line was
                // `global_store_dword ... sc1` x 8 -> s_barrier -> global_atomic_add), so the ticket of thread 0 -- another wave, another
                // L2 channel -- could become visible before these stores and the last workgroup would read stale rows.
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) dst[e] = t8[e];
            }
        }
        if constexpr (POOL == 2) {
            // Last workgroup of the image: ticket by a relaxed device-scope atomic. NO device-scope fence: on gfx950 a release at agent
            // scope writes back the XCD's whole L2 (measured: +20 - 100 us per launch, once per workgroup). Instead: the partial sums
            // were stored with device-scope (write-through) atomics and every storing thread has waited for their acknowledgement
            // (s_waitcnt vmcnt(0) above); the barrier orders those waits before the ticket of thread 0; the last workgroup reads the
            // rows back with device-scope atomic loads issued after its ticket has returned (the barrier below). This is the
            // hardware's ordering, spelled out -- the formal-model release would be the flushing fence.
            __shared__ int s_last;
            __syncthreads();
            if (threadIdx.x == 0)
                s_last = __hip_atomic_fetch_add(&a.se_counter[n], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nblocks - 1);
            __syncthreads();
            if (s_last) {
                if (dw_se_tail_is_small(a.c, a.se_sq, nblocks)) dw_se_tail_small(a, n, nblocks, red);
                else dw_se_tail(a, n, nblocks, red);
                if (threadIdx.x == 0) a.se_counter[n] = 0u;
            }
        }
    }
}

template <int K, int S, int TW, int POOL, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((MODE & 4) ? (K == 3 ? 4 : 3) : 1))) void dw_kernel(DwArgs a, int nblocks) {
    int img, bx;
    if (!xcd_image_of2(a.xq, a.n, img, bx)) return;
    dw_body<K, S, TW, POOL, MODE>(a, bx, nblocks, img);
}

// Grouped launch: up to 12 independent depthwise problems of the same (k, stride) and batch in ONE launch (the head
// depthwise convs of all pyramid levels, both heads). blockIdx.x is flat over the problems' per-image block counts (x 8 with the
// XCD grouping: every problem's range starts at a multiple of 8), blockIdx.y is the image slot.
struct DwGroup {
    int count;
    int start[13];
    int nblocks[12];        // workgroups per image
    DwArgs a[12];
};

template <int K, int S, int TW, int MODE = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((MODE & 4) && K == 3 ? 4 : 1))) void dw_group_kernel(DwGroup g) {
    int p = 0;
#pragma unroll
    for (int i = 1; i < 12; ++i)
        if (i < g.count && (int)blockIdx.x >= g.start[i]) p = i;
    const DwArgs& a = g.a[p];
    const int rel = blockIdx.x - g.start[p];
    int img, bx;
    if (a.xq > 0) { img = (rel & 7) * a.xq + blockIdx.y; bx = rel >> 3; }
    else { img = blockIdx.y; bx = rel; }
    if (img >= a.n) return;
    dw_body<K, S, TW, 0, MODE>(a, bx, g.nblocks[p], img);
}

template <int K, int S, int TW>
bool dw_fill(DwArgs& a) {
    const unsigned c8 = a.c / 8, xs = (a.wo + TW - 1) / TW;
    a.fd_c8 = fastdiv(c8);
    a.fd_xs = fastdiv(xs);
    // row blocks: the largest power of two <= DN_DW_RB that divides the output height (the thread count, and with it the number of
    // pooled partial rows the plan sized, stays what it is)
    int rb = 0;
    for (int want = 1; (2 << rb) <= want && a.ho % (2 << rb) == 0; ) ++rb;
    a.rb_log2 = rb;
    a.fd_rbxs = fastdiv(xs << rb);
    const unsigned long long threads = (unsigned long long)a.ho * xs * c8 + 256;
    return fd_ok(threads, c8) && fd_ok(threads / c8 + 1, xs << rb) && (unsigned long long)a.h * a.w_ * a.c < 0x80000000ull;
}

template <int K, int S, int TW>
int launch_dw_group(const DwArgs* arr, int count, hipStream_t s) {
    DwGroup g{};
    g.count = count;
    int acc = 0;
    for (int i = 0; i < count; ++i) {
        g.a[i] = arr[i];
        DN_REQUIRE((dw_fill<K, S, TW>(g.a[i])), "depthwise group: problem %d outside the index range of the kernel", i);
        DN_REQUIRE(!arr[i].pool && arr[i].n == arr[0].n && arr[i].xq == arr[0].xq, "depthwise group: problems must share the batch and have no pooled output");
        g.start[i] = acc;
        g.nblocks[i] = dn_cdiv((long)arr[i].ho * ((arr[i].wo + TW - 1) / TW) * (arr[i].c / 8), 256);
        acc += g.nblocks[i] * (arr[i].xq > 0 ? 8 : 1);
    }
    g.start[count] = acc;
    dn_note_kernel("dw_group_kernel<%d,%d,%d>", K, S, TW);
    hipLaunchKernelGGL((dw_group_kernel<K, S, TW, 4>), dim3(acc, arr[0].xq > 0 ? arr[0].xq : arr[0].n), dim3(256), 0, s, g);
    return DN_OK;
}

template <int K, int S, int TW>
int launch_dw(const DwArgs& a0, hipStream_t s) {
    DwArgs a = a0;
    DN_REQUIRE((dw_fill<K, S, TW>(a)), "depthwise: %d x %d x %d outside the index range of the kernel", a.ho, a.wo, a.c);
    const long threads = (long)a.ho * ((a.wo + TW - 1) / TW) * (a.c / 8);       // per image
    dn_note_kernel("dw_kernel<%d,%d,%d>", K, S, TW);
    const int nblocks = dn_cdiv(threads, 256);
    if (a.se_scale) DN_REQUIRE(a.pool && a.se_counter && depthwise_se_tail_supported(a.c, a.se_sq), "depthwise: squeeze-excitation tail needs the pooled output and c <= 1024, squeeze <= 256, both multiples of 8");
    const size_t pool_lds = a.se_scale ? (size_t)(a.c + a.se_sq + 2048) * 4 : (size_t)256 * 8 * 4;
    const dim3 grid = xcd_grid2(nblocks, a.xq, a.n);
    const int cls = a.pool ? (a.se_scale ? 4 : 2) : 1;      // launch class: 1 = no pooling, 2 = pooled sums, 4 = pooled sums + SE tail
    // software-pipelined rows, default on, stress-tested. 5x5 (DN_DW_PIPE): one exposed round trip instead of three, 29 -> 25.7 us per 40 x 40 launch. 3x3 (DN_DW_PIPE3): the batched form already takes its three rows in one round trip, but two row sets instead of three are 127 registers instead of 140 -- 4 waves per SIMD: the head group 36.4 -> 33.8 us.
    // (Round 2 - 3 also carried a ONE-row-at-a-time form, 70 - 116 registers: with it a forward's result depended on what else ran on the chip; cause not found in
    //  two rounds of hunting (profiles/r03_dw_rows_hunt.txt), deleted in round 4 -- the poison test of tests/test_gpu_pipeline.py and the in-flight stress test guard what is left.)
    const bool pipe = K == 5 ? (7 & cls) != 0 : (7 & cls) != 0;
    const size_t lds = a.pool ? pool_lds : 0;
    if (pipe) {
        if (a.pool && a.se_scale) hipLaunchKernelGGL((dw_kernel<K, S, TW, 2, 4>), grid, dim3(256), lds, s, a, nblocks);
        else if (a.pool) hipLaunchKernelGGL((dw_kernel<K, S, TW, 1, 4>), grid, dim3(256), lds, s, a, nblocks);
        else hipLaunchKernelGGL((dw_kernel<K, S, TW, 0, 4>), grid, dim3(256), lds, s, a, nblocks);
        return DN_OK;
    }
    if (a.pool && a.se_scale) hipLaunchKernelGGL((dw_kernel<K, S, TW, 2, 0>), grid, dim3(256), lds, s, a, nblocks);
    else if (a.pool) hipLaunchKernelGGL((dw_kernel<K, S, TW, 1, 0>), grid, dim3(256), lds, s, a, nblocks);
    else hipLaunchKernelGGL((dw_kernel<K, S, TW, 0, 0>), grid, dim3(256), lds, s, a, nblocks);
    return DN_OK;
}

template <int K, int S, int TW>
int dw_blocks(const DwArgs& a) { return dn_cdiv((long)a.ho * ((a.wo + TW - 1) / TW) * (a.c / 8), 256); }

// ---- SE FCs: (sum of partials)/pixels -> fc1(+b) -> ReLU -> fc2(+b) -> Hardsigmoid   (mobilenetv3.py:31-36) ------------
// One 1024-thread workgroup per image. Both weight matrices are stored TRANSPOSED and in fp16 at plan time (w1t [c][sq], w2t [sq][c];
// fp32 accumulation -- the rounding of a weight is below the fp16 rounding of the activations the scale multiplies),
// so a thread owns one output and walks a slice of the reduction axis with loads that are coalesced across the wave and
// all independent -- no cross-lane reductions (the earlier wave-per-output form spent most of its time in 6-step shuffle
// trees), one LDS combine of the K-slices per FC. Loads are issued in batches of 16 before their first use: a plain
// `t += w[i] * x[i]` loop waits one full L2/HBM latency per iteration on this chip.
__global__ __launch_bounds__(1024) void se_fc_kernel(const float* __restrict__ partial, int nblk, const unsigned* __restrict__ w1t,
                                                   const float* __restrict__ b1, const unsigned* __restrict__ w2t,
                                                   const float* __restrict__ b2, float* __restrict__ scale,
                                                   int c, int sq, float inv_pixels, long long* __restrict__ stamps, int nimg, int xq) {
#ifdef DN_DEV_STAMPS
#define SE_STAMP(k) do { if (stamps && threadIdx.x == 0) stamps[blockIdx.x * 8 + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SE_STAMP(k) do { } while (0)
#endif
    // w1t: fc1 weight transposed [c][sq] fp16, read as [c][sq/2] half pairs; w2t: fc2 weight transposed [sq][c] fp16 as
    // [sq][c/2] pairs. A thread owns TWO adjacent outputs (one 4-byte load per weight row) and a K-slice.
    extern __shared__ float sh[];      // mean[c], z[sq], part[2][1024]
    float* mean = sh;
    float* z = sh + c;
    float* part = z + sq;
    int n, unused;
    if (!xcd_image_of(blockIdx.x, 1, xq, nimg, n, unused)) return;
    const int tid = threadIdx.x;
    SE_STAMP(0);
    const int sq2 = sq >> 1, c2 = c >> 1;
    // (round 3: exact, not wave-rounded, pair counts -- 84 pairs give 12 K slices of 56 rows where 128 gave 8 of 84: the 672-channel
    // blocks walk their slices in two batches of loads per FC instead of three)
    const int JP = sq2;                               // fc1: output pair j, slice r1 of the c inputs
    const int KS1 = 1024 / JP;
    const int j = tid % JP, r1 = tid / JP;
    const int per1 = (c + KS1 - 1) / KS1;
    const int i0 = r1 * per1, i1 = min(c, i0 + per1);
    const bool act1 = j < sq2 && r1 < KS1;
    const int CP = c2;                                // fc2: output pair i, slice r2 of the sq inputs
    const int KS2 = max(1, 1024 / CP);
    const int i = tid % CP, r2 = tid / CP;
    const int per2 = (sq + KS2 - 1) / KS2;
    const int j0 = r2 * per2, j1 = min(sq, j0 + per2);
    const bool act2 = i < c2 && r2 < KS2;
    constexpr int FB = 32;                            // weight rows in flight per thread
    // (Measured and dropped: requesting the next phase's first weight rows early -- before or right after the loads the
    // current phase waits for. Memory returns in order; both variants were slower than the plain phase-by-phase form.)
    // pooled mean: thread (channel, row slice r) sums every RS-th partial row; slices combined through LDS
    {
        const int CPm = (c + 63) & ~63;
        const int RS = max(1, min(1024 / CPm, nblk));
        const int ic = tid % CPm, r = tid / CPm;
        float t = 0.f;
        if (ic < c && r < RS) {
            const float* p = partial + (size_t)n * nblk * c + ic;
            for (int b0 = r; b0 < nblk; b0 += RS * 16) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = p[(size_t)min(b0 + u * RS, nblk - 1) * c];
#pragma unroll
                for (int u = 0; u < 16; ++u) t += (b0 + u * RS < nblk) ? v[u] : 0.f;
            }
        }
        part[tid] = t;
        __syncthreads();
        if (tid < c) {
            float m = 0.f;
            for (int q = 0; q < RS; ++q) m += part[q * CPm + tid];
            mean[tid] = m * inv_pixels;
        }
        __syncthreads();
    }
    SE_STAMP(1);
    auto lo = [](unsigned w) { return (float)__builtin_bit_cast(half_t, (unsigned short)(w & 0xffffu)); };
    auto hi = [](unsigned w) { return (float)__builtin_bit_cast(half_t, (unsigned short)(w >> 16)); };
    // fc1: z[j] = relu(b1[j] + sum_i w1t[i][j] * mean[i])
    {
        float t0 = 0.f, t1 = 0.f;
        if (act1) {
            for (int ib = i0; ib < i1; ib += FB) {
                unsigned v[FB];
#pragma unroll
                for (int u = 0; u < FB; ++u) v[u] = w1t[(size_t)min(ib + u, i1 - 1) * sq2 + j];     // clamped, not predicated: a predicated load costs a branch and a wait each
#pragma unroll
                for (int u = 0; u < FB; ++u) {
                    const float m = (ib + u < i1) ? mean[ib + u] : 0.f;
                    t0 += lo(v[u]) * m;
                    t1 += hi(v[u]) * m;
                }
            }
        }
        part[tid] = t0;
        part[1024 + tid] = t1;
        __syncthreads();
        if (tid < sq) {
            float a = b1[tid];
            const float* pp = part + (tid & 1) * 1024 + (tid >> 1);
            for (int q = 0; q < KS1; ++q) a += pp[q * JP];
            z[tid] = fmaxf(a, 0.f);
        }
        __syncthreads();
    }
    SE_STAMP(2);
    // fc2: scale[i] = hardsigmoid(b2[i] + sum_j w2t[j][i] * z[j])
    {
        float t0 = 0.f, t1 = 0.f;
        if (act2) {
            for (int jb = j0; jb < j1; jb += FB) {
                unsigned v[FB];
#pragma unroll
                for (int u = 0; u < FB; ++u) v[u] = w2t[(size_t)min(jb + u, j1 - 1) * c2 + i];
#pragma unroll
                for (int u = 0; u < FB; ++u) {
                    const float zz = (jb + u < j1) ? z[jb + u] : 0.f;
                    t0 += lo(v[u]) * zz;
                    t1 += hi(v[u]) * zz;
                }
            }
        }
        part[tid] = t0;
        part[1024 + tid] = t1;
        __syncthreads();
        if (tid < c) {
            float a = b2[tid];
            const float* pp = part + (tid & 1) * 1024 + (tid >> 1);
            for (int q = 0; q < KS2; ++q) a += pp[q * CP];
            scale[(size_t)n * c + tid] = fminf(fmaxf(a + 3.f, 0.f), 6.f) * (1.f / 6.f);
        }
    }
    SE_STAMP(3);
}

// Round 3: the same three phases with every global load 16 bytes wide and ONE round trip per phase where the sizes allow (c % 8 == 0,
// sq % 8 == 0 -- every squeeze-excitation of the zoo). A thread owns 8 adjacent outputs of an FC and a slice of its reduction axis: 14 loads
// per thread and FC for the 672 / 168 blocks (the 4-byte form walked 56 rows in two batches of 32), the pooled partial rows are read as
// float4 by (4 channels, row slice) threads in one batch, the first fc1 batch is requested together with the partial rows and the first fc2
// batch before the fc1 slices are combined, so the kernel is three exposed round trips instead of six to seven. Weights are fp16 operands
// of v_fma_mix_f32 against the fp32 mean / z (no conversions). Sums run in a different order than se_fc_kernel: equal up to fp32 rounding.
__device__ __forceinline__ void se_fma8(float (&acc)[8], const unsigned __attribute__((ext_vector_type(4))) & w, float m) {
    const unsigned ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[2 * i]) : "v"(ww[i]), "v"(m));
        asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[2 * i + 1]) : "v"(ww[i]), "v"(m));
    }
}

typedef float se_f4 __attribute__((ext_vector_type(4)));
typedef unsigned se_u4 __attribute__((ext_vector_type(4)));
// 16-byte load from a uniform base + 32-bit byte offset, as an asm volatile statement: these keep their program order among themselves (LLVM sinks
// ordinary loads to their first use -- the partial rows would land BEHIND the weight requests -- and a sched_barrier only binds the machine
// scheduler), and the waits for them are written by hand below (the compiler does not count asm loads).
__device__ __forceinline__ void se_load16(se_u4& v, const void* base, unsigned off) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory");
}
__device__ __forceinline__ void se_load16(se_f4& v, const void* base, unsigned off) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory");
}
// "at most N vector-memory operations still in flight" with the registers the wait protects tied to it (their uses cannot move above it)
template <int FB>
__device__ __forceinline__ void se_wait_w(se_u4 (&a)[FB]) {       // every weight row of the batch has arrived
    static_assert(FB == 14 || FB == 15, "operand lists below");
    if constexpr (FB == 14)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
                     "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]));
    else
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
                     "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]));
}
template <int FB, int PB, bool ALL>
__device__ __forceinline__ void se_wait_p(se_f4 (&a)[PB]) {       // the partial rows have arrived; !ALL: the FB weight rows requested behind them stay in flight
    static_assert(PB == 4 || PB == 6, "operand lists below");
    if constexpr (PB == 6) {
        if constexpr (ALL) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]));
        else if constexpr (FB == 14) asm volatile("s_waitcnt vmcnt(14)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]));
        else asm volatile("s_waitcnt vmcnt(15)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]));
    } else {
        if constexpr (ALL) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
        else if constexpr (FB == 14) asm volatile("s_waitcnt vmcnt(14)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
        else asm volatile("s_waitcnt vmcnt(15)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
    }
}

template <int FB, int PB>       // 16-byte weight rows / partial rows in flight per thread: (14, 6), or (15, 4) for the 960 / 240 blocks (29 - 30 rows per slice)
__global__ __launch_bounds__(1024) void se_fc8_kernel(const float* __restrict__ partial, int nblk, const half_t* __restrict__ w1t,
                                                    const float* __restrict__ b1, const half_t* __restrict__ w2t,
                                                    const float* __restrict__ b2, float* __restrict__ scale,
                                                    int c, int sq, float inv_pixels, long long* __restrict__ stamps, int nimg, int xq, int rot_step) {
    extern __shared__ float sh[];      // mean[c], z[sq], part[max(RS * c, KS1 * sq, KS2 * c)]
    float* mean = sh;
    float* z = sh + c;
    float* part = z + sq;
    int n, unused;
    if (!xcd_image_of(blockIdx.x, 1, xq, nimg, n, unused)) return;
    const int tid = threadIdx.x;
    SE_STAMP(0);
    // roles
    const int C4 = c >> 2, RS = max(1, min(1024 / C4, nblk));
    const int ic4 = tid % C4, rr = tid / C4;
    const bool actm = rr < RS;
    const int G1 = sq >> 3, KS1 = max(1, min(1024 / G1, c >> 3));          // >= 8 rows per slice; KS1 * sq <= 8192 floats of LDS
    // (slice -> thread assignment rotated by the image: 64 workgroups walking the same weight rows in the same order at the same moment all
    //  queue on the same L2 channel; the slices themselves, and the order they are combined in, do not change)
    const int rot = rot_step * n;
    const int j8 = tid % G1, r1s = tid / G1;
    const bool act1 = r1s < KS1;
    const int r1 = act1 ? (r1s + rot) % KS1 : r1s;
    const int per1 = (c + KS1 - 1) / KS1;
    const int i0 = min(r1 * per1, c), i1 = min(c, i0 + per1);
    const int G2 = c >> 3, KS2 = max(1, min(1024 / G2, sq >> 3));
    const int i8 = tid % G2, r2s = tid / G2;
    const bool act2 = r2s < KS2;
    const int r2 = act2 ? (r2s + rot) % KS2 : r2s;
    const int per2 = (sq + KS2 - 1) / KS2;
    const int j0 = min(r2 * per2, sq), j1 = min(sq, j0 + per2);
    // (the two biases too: a compiler-generated load in front of the asm loads gets a full wait before the first of them)
    float b1v, b2v;
    asm volatile("global_load_dword %0, %1, %2" : "=v"(b1v) : "v"((unsigned)min(tid, sq - 1) * 4u), "s"(b1) : "memory");
    asm volatile("global_load_dword %0, %1, %2" : "=v"(b2v) : "v"((unsigned)min(tid, c - 1) * 4u), "s"(b2) : "memory");
    // ---- requests: the partial rows of this thread's (4 channels, slice), then the first batch of its fc1 weight rows. Unconditional, clamped.
    const float* pbase = partial + (size_t)n * nblk * c;
    const unsigned pcol = (unsigned)min(ic4, C4 - 1) * 16u, prs = (unsigned)C4 * 16u;
    const int rrc = min(rr, RS - 1);
    se_f4 pv[PB];
#pragma unroll
    for (int u = 0; u < PB; ++u) se_load16(pv[u], pbase, pcol + (unsigned)min(rrc + u * RS, nblk - 1) * prs);
    const unsigned w1col = (unsigned)min(j8, G1 - 1) * 16u, w1rs = (unsigned)G1 * 16u;
    se_u4 wv[FB];
#pragma unroll
    for (int u = 0; u < FB; ++u) se_load16(wv[u], w1t, w1col + (unsigned)min(i0 + u, c - 1) * w1rs);
    // ---- pooled mean
    {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        se_wait_p<FB, PB, false>(pv);
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const float ok = (rrc + u * RS < nblk) ? 1.f : 0.f;
            t.x = fmaf(pv[u].x, ok, t.x); t.y = fmaf(pv[u].y, ok, t.y); t.z = fmaf(pv[u].z, ok, t.z); t.w = fmaf(pv[u].w, ok, t.w);
        }
        for (int bb = PB * RS; bb < nblk; bb += PB * RS) {         // more partial rows than one batch covers (uniform bound)
#pragma unroll
            for (int u = 0; u < PB; ++u) se_load16(pv[u], pbase, pcol + (unsigned)min(rrc + bb + u * RS, nblk - 1) * prs);
            se_wait_p<FB, PB, true>(pv);
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const float ok = (rrc + bb + u * RS < nblk) ? 1.f : 0.f;
                t.x = fmaf(pv[u].x, ok, t.x); t.y = fmaf(pv[u].y, ok, t.y); t.z = fmaf(pv[u].z, ok, t.z); t.w = fmaf(pv[u].w, ok, t.w);
            }
        }
        if (actm) *reinterpret_cast<float4*>(&part[rr * c + ic4 * 4]) = t;
        __syncthreads();
        if (tid < c) {
            float m = 0.f;
            for (int q = 0; q < RS; ++q) m += part[q * c + tid];
            mean[tid] = m * inv_pixels;
        }
        __syncthreads();
    }
    SE_STAMP(1);
    const unsigned w2col = (unsigned)min(i8, G2 - 1) * 16u, w2rs = (unsigned)G2 * 16u;
    // ---- fc1: z[j] = relu(b1[j] + sum_i w1t[i][j] * mean[i])
    {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        se_wait_w<FB>(wv);
        asm volatile("" : "+v"(b1v), "+v"(b2v));        // (their uses stay below the wait as well)
#pragma unroll
        for (int u = 0; u < FB; ++u) se_fma8(acc, wv[u], (i0 + u < i1) ? mean[i0 + u] : 0.f);
        for (int bb = FB; bb < per1; bb += FB) {                    // (uniform bound)
#pragma unroll
            for (int u = 0; u < FB; ++u) se_load16(wv[u], w1t, w1col + (unsigned)min(i0 + bb + u, c - 1) * w1rs);
            se_wait_w<FB>(wv);
#pragma unroll
            for (int u = 0; u < FB; ++u) se_fma8(acc, wv[u], (i0 + bb + u < i1) ? mean[i0 + bb + u] : 0.f);
        }
        // the first fc2 batch is on its way while the slices are combined
#pragma unroll
        for (int u = 0; u < FB; ++u) se_load16(wv[u], w2t, w2col + (unsigned)min(j0 + u, sq - 1) * w2rs);
        if (act1) {
            *reinterpret_cast<float4*>(&part[r1 * sq + j8 * 8]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            *reinterpret_cast<float4*>(&part[r1 * sq + j8 * 8 + 4]) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
        __syncthreads();
        if (tid < sq) {
            float a = b1v;                             // (requested first, long arrived: se_wait_w above waited for everything)
            for (int q = 0; q < KS1; ++q) a += part[q * sq + tid];
            z[tid] = dn_relu(a);
        }
        __syncthreads();
    }
    SE_STAMP(2);
    // ---- fc2: scale[i] = hardsigmoid(b2[i] + sum_j w2t[j][i] * z[j])
    {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        se_wait_w<FB>(wv);
#pragma unroll
        for (int u = 0; u < FB; ++u) se_fma8(acc, wv[u], (j0 + u < j1) ? z[j0 + u] : 0.f);
        for (int bb = FB; bb < per2; bb += FB) {
#pragma unroll
            for (int u = 0; u < FB; ++u) se_load16(wv[u], w2t, w2col + (unsigned)min(j0 + bb + u, sq - 1) * w2rs);
            se_wait_w<FB>(wv);
#pragma unroll
            for (int u = 0; u < FB; ++u) se_fma8(acc, wv[u], (j0 + bb + u < j1) ? z[j0 + bb + u] : 0.f);
        }
        if (act2) {
            *reinterpret_cast<float4*>(&part[r2 * c + i8 * 8]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            *reinterpret_cast<float4*>(&part[r2 * c + i8 * 8 + 4]) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
        __syncthreads();
        if (tid < c) {
            float a = b2v;
            for (int q = 0; q < KS2; ++q) a += part[q * c + tid];
            scale[(size_t)n * c + tid] = dn_relu6(a + 3.f) * (1.f / 6.f);
        }
    }
    SE_STAMP(3);
}

// ---- stem: dense kxk conv on the NCHW fp32 image, normalisation on load, NHWC fp16 out ------------------
// One thread per output pixel, all COUT channels. The weights are wave-uniform: indexing the kernel-argument pointer
// with compile-time offsets makes hipcc fetch them with s_load (scalar cache) and feed them as SGPR operands of
// v_fmac -- no LDS, no 400-register weight image (the first version needed 256 VGPRs and spilled for COUT >= 32).
template <int COUT, int K>
__global__ __launch_bounds__(256) void stem_kernel(StemArgs a, int nblocks) {
    const float* __restrict__ wts = a.w;
    const float* __restrict__ bias = a.bias;
    if (a.zero_u32 && blockIdx.x == 0)       // the chain's squeeze-excitation counters (DwArgs::se_counter): cleared once per forward, ahead of every pooling launch
        for (int i = threadIdx.x; i < a.zero_count; i += 256) a.zero_u32[i] = 0u;
    int n, bx;
    if (!xcd_image_of(blockIdx.x, nblocks, a.xq, a.n, n, bx)) return;
    int idx = bx * 256 + threadIdx.x;
    const int ox = idx % a.wo;
    const int oy = idx / a.wo;
    if (oy >= a.ho) return;
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = bias[o];
    // only one (channel, ky) row of taps is unrolled at a time: K*COUT weights live in SGPRs per step. Fully unrolling
    // makes hipcc hoist all K*K*3*COUT scalar loads and spill SGPRs through thousands of v_readlane.
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        const float* plane = a.img + ((size_t)n * 3 + c) * a.h * a.w_;
        const float mean = a.mean[c], inv = a.inv_std[c];
#pragma unroll 1
        for (int ky = 0; ky < K; ++ky) {
            const int iy = oy * a.stride - a.pad + ky;
            const bool yok = iy >= 0 && iy < a.h;
            const float* prow = plane + (yok ? iy : 0) * a.w_;
            const float* wp = wts + (c * K + ky) * K * COUT;
            float v[K];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int ix = ox * a.stride - a.pad + kx;
                float t = 0.f;      // zero padding is applied to the NORMALISED image (transform then conv)
                if (yok && ix >= 0 && ix < a.w_) t = (prow[ix] - mean) * inv;
                v[kx] = t;
            }
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
#pragma unroll
                for (int o = 0; o < COUT; ++o) acc[o] += v[kx] * wp[kx * COUT + o];
        }
    }
    half_t* op = a.out + ((size_t)(n * a.ho + oy) * a.wo + ox) * COUT;
    dn_act_n<float[COUT], COUT>(acc, a.act);      // one uniform switch for the thread's COUT values (the branch-free dn_act costs ~9 instructions per value)
#pragma unroll
    for (int o8 = 0; o8 < COUT / 8; ++o8) {
        half8 hv;
#pragma unroll
        for (int e = 0; e < 8; ++e) hv[e] = (half_t)acc[o8 * 8 + e];
        *reinterpret_cast<half8*>(op + o8 * 8) = hv;
    }
}

// 3x3 stride 2 pad 1 on an even-width image (every model here). The three taps of pixel ox are input columns 2ox-1, 2ox, 2ox+1;
// lanes hold consecutive ox, so one aligned float2 per lane covers whole 128-B lines across the wave and the left tap is the
// previous lane's second element. All nine (channel, row) loads are requested before the first use -- the generic kernel
// above pays one exposed memory round trip per (channel, row) step because its loop must not be unrolled (SGPR pressure);
// here the loads are hoisted and the normalised taps wait in LDS for the (still not unrolled) weight loop.
template <int COUT>
__global__ __launch_bounds__(256) void stem3s2_kernel(StemArgs a, int nblocks) {
    __shared__ float taps[27][256];     // this thread's 27 normalised taps, parked in its own LDS column between the phases
    const float* __restrict__ wts = a.w;
    const float* __restrict__ bias = a.bias;
    if (a.zero_u32 && blockIdx.x == 0)       // the chain's squeeze-excitation counters (DwArgs::se_counter): cleared once per forward, ahead of every pooling launch
        for (int i = threadIdx.x; i < a.zero_count; i += 256) a.zero_u32[i] = 0u;
    int n, bx;
    if (!xcd_image_of(blockIdx.x, nblocks, a.xq, a.n, n, bx)) return;
    const int idx = bx * 256 + threadIdx.x;
    const int ox = idx % a.wo;
    int oy = idx / a.wo;
    const bool live = oy < a.ho;        // no early return: the shuffle needs every lane; dead threads redo the last row
    if (!live) oy = a.ho - 1;
    {
        float2 p[3][3];
        float left[3][3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy * 2 - 1 + ky;
                const bool yok = iy >= 0 && iy < a.h;
                const float* prow = a.img + (((size_t)n * 3 + c) * a.h + (yok ? iy : 0)) * a.w_;
                p[c][ky] = yok ? *reinterpret_cast<const float2*>(prow + 2 * ox) : make_float2(0.f, 0.f);
                left[c][ky] = ((threadIdx.x & 63) == 0 && ox > 0 && yok) ? prow[2 * ox - 1] : 0.f;
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float mean = a.mean[c], inv = a.inv_std[c];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy * 2 - 1 + ky;
                const bool yok = iy >= 0 && iy < a.h;
                float l = __shfl_up(p[c][ky].y, 1);
                if ((threadIdx.x & 63) == 0) l = left[c][ky];
                const int st = (c * 3 + ky) * 3;
                taps[st + 0][threadIdx.x] = (yok && ox > 0) ? (l - mean) * inv : 0.f;      // zero padding of the NORMALISED image
                taps[st + 1][threadIdx.x] = yok ? (p[c][ky].x - mean) * inv : 0.f;
                taps[st + 2][threadIdx.x] = yok ? (p[c][ky].y - mean) * inv : 0.f;
            }
        }
    }
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = bias[o];
    // one (channel, row) step at a time: K*COUT weights live in SGPRs per step (see stem_kernel)
#pragma unroll 1
    for (int st = 0; st < 9; ++st) {
        const float* wp = wts + st * 3 * COUT;
        const float v0 = taps[st * 3 + 0][threadIdx.x], v1 = taps[st * 3 + 1][threadIdx.x], v2 = taps[st * 3 + 2][threadIdx.x];
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] += v0 * wp[o];
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] += v1 * wp[COUT + o];
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] += v2 * wp[2 * COUT + o];
    }
    if (!live) return;
    half_t* op = a.out + ((size_t)(n * a.ho + oy) * a.wo + ox) * COUT;
    dn_act_n<float[COUT], COUT>(acc, a.act);      // one uniform switch for the thread's COUT values (the branch-free dn_act costs ~9 instructions per value)
#pragma unroll
    for (int o8 = 0; o8 < COUT / 8; ++o8) {
        half8 hv;
#pragma unroll
        for (int e = 0; e < 8; ++e) hv[e] = (half_t)acc[o8 * 8 + e];
        *reinterpret_cast<half8*>(op + o8 * 8) = hv;
    }
}

// 3x3 stride-1 stem with 64 output channels (VGG conv1_1, ssd_vgg16.py:33 via torchvision vgg16 features[0]) on the fp32 matrix
// cores: K = 27 taps padded to 28 = 14 steps of v_mfma_f32_32x32x2_f32, exact fp32 products and accumulation like the VALU
// kernel above, which spends 1728 v_fmac per pixel here (26 TFLOP/s, 1 TB/s). A = weights (lane: channel lane & 31, tap
// 2 i + (lane >> 5)), kept in registers; B = the normalised image taps, each lane loading its own (pixel lane & 31, tap) value
// straight from the planar fp32 image (32 consecutive pixels per tap: whole 128-B lines). No LDS on the input side. The 32 x 64
// result tile goes through a per-wave LDS slab so that every lane writes 16-byte row-contiguous chunks of the NHWC fp16 output.
__global__ __launch_bounds__(256) void stem_mfma64_kernel(StemArgs a, int tiles_per_image, int tiles_per_wave, int nblocks) {
    if (a.zero_u32 && blockIdx.x == 0)       // the chain's squeeze-excitation counters (DwArgs::se_counter): cleared once per forward, ahead of every pooling launch
        for (int i = threadIdx.x; i < a.zero_count; i += 256) a.zero_u32[i] = 0u;
    __shared__ __attribute__((aligned(16))) half_t slab[4][32 * 72];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    int n, bx;
    if (!xcd_image_of(blockIdx.x, nblocks, a.xq, a.n, n, bx)) return;
    const int HW = a.h * a.w_, OHW = a.ho * a.wo;
    // this lane's tap of K step i: t = 2 i + hh = (c * 3 + ky) * 3 + kx; t == 27 is the zero pad
    float wa[14][2];
    int toff[14];               // offset of the tap relative to the output pixel's own position in plane 0
    int tky[14], tkx[14];
    float tmean[14], tinv[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) {
        const int t = 2 * i + hh;
        const int tc = min(t, 26);
        const int c = tc / 9, ky = (tc - c * 9) / 3, kx = tc - c * 9 - ky * 3;
        wa[i][0] = t < 27 ? a.w[tc * 64 + r] : 0.f;
        wa[i][1] = t < 27 ? a.w[tc * 64 + 32 + r] : 0.f;
        toff[i] = c * HW + (ky - a.pad) * a.w_ + (kx - a.pad);
        tky[i] = t < 27 ? ky - a.pad : 1 << 20;         // the pad tap is always "outside"
        tkx[i] = kx - a.pad;
        tmean[i] = a.mean[c];
        tinv[i] = a.inv_std[c];
    }
    float4 bq[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[j][g] = *reinterpret_cast<const float4*>(a.bias + j * 32 + 8 * g + 4 * hh);
    const float* img = a.img + (size_t)n * 3 * HW;
    half_t* outp = a.out + (size_t)n * OHW * 64;
    half_t* sl = slab[wave];
    const int tile0 = (bx * 4 + wave) * tiles_per_wave;
    for (int tt = 0; tt < tiles_per_wave; ++tt) {
        const int tile = tile0 + tt;
        if (tile >= tiles_per_image) break;           // wave-uniform
        const int p = min(tile * 32 + r, OHW - 1);     // pixels beyond the image: computed, never stored
        const int oy = p / a.wo, ox = p - oy * a.wo;
        const int base = oy * a.w_ + ox;
        float v[14];
#pragma unroll
        for (int i = 0; i < 14; ++i) {
            const int iy = oy + tky[i], ix = ox + tkx[i];
            const bool ok = iy >= 0 && iy < a.h && ix >= 0 && ix < a.w_;
            v[i] = img[ok ? base + toff[i] : 0];
            // zero padding is applied to the NORMALISED image (transform then conv)
            v[i] = ok ? (v[i] - tmean[i]) * tinv[i] : 0.f;
        }
        floatx16 acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
        for (int i = 0; i < 14; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[i][0], v[i], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[i][1], v[i], acc1, 0, 0, 0);
        }
        // lane = pixel r, registers 4g..4g+3 = channels 32 j + 8 g + 4 hh .. +3
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            half4 h0, h1;
            const float4 b0 = bq[0][g], b1 = bq[1][g];
            float t8[8] = {acc0[4 * g + 0] + b0.x, acc0[4 * g + 1] + b0.y, acc0[4 * g + 2] + b0.z, acc0[4 * g + 3] + b0.w,
                           acc1[4 * g + 0] + b1.x, acc1[4 * g + 1] + b1.y, acc1[4 * g + 2] + b1.z, acc1[4 * g + 3] + b1.w};
            dn_act_n<float[8], 8>(t8, a.act);
#pragma unroll
            for (int e = 0; e < 4; ++e) { h0[e] = (half_t)t8[e]; h1[e] = (half_t)t8[4 + e]; }
            *reinterpret_cast<half4*>(sl + r * 72 + 8 * g + 4 * hh) = h0;
            *reinterpret_cast<half4*>(sl + r * 72 + 32 + 8 * g + 4 * hh) = h1;
        }
        // the slab is private to the wave: its own LDS operations are ordered, no barrier
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ch = lane + 64 * u;              // 16-B chunk of the 32 x 128-B tile
            const int row = ch >> 3, q = ch & 7;
            const uint4 val = *reinterpret_cast<const uint4*>(sl + row * 72 + q * 8);
            const int pp = tile * 32 + row;
            if (pp < OHW) *reinterpret_cast<uint4*>(outp + (size_t)pp * 64 + q * 8) = val;
        }
    }
}

// The same tiles, products and accumulation order as stem_mfma64_kernel (bit-identical output), scheduled as a pipeline inside the wave.
// The kernel above runs request -> wait -> normalise -> 28 dependent-pair MFMAs -> epilogue -> stores strictly in sequence per tile, and its
// costs ADD (ablations on the 512 x 512 batch, 16 images per launch: 236 us = vector work 49 + matrix chain ~95 + stores ~55 + taps ~45;
// matrix pipe 0.35 busy). Here the wave's TPW tiles are straight-line code and every step t is ONE scheduling region that holds
//   the taps of tile t + 2 (requests),  the MFMA chain of tile t,  bias / ReLU / fp16 / slab / stores of tile t - 1,  the wait for and the
//   normalisation of tile t + 1
// so that each memory round trip has a whole chain to complete in and the vector work of three tiles sits between one tile's MFMAs.
// pad 1, ReLU and output predication by the buffer's range keep the step free of branches; border handling is a 6-bit mask test per tap
// (borders the tap must not cross & borders the pixel touches); pixel coordinates advance by addition.
// Requests, stores and waits are `asm volatile`: hipcc sinks a plain load to its first use (behind the chain it is meant to overlap), and
// asm statements keep their order -- requests(t + 2), stores(t - 1), wait(t + 1) -- so the wait's count is known: loads and stores retire in
// order, and younger than the taps of tile t + 1 are the stores of t - 2, the requests of t + 2 and the stores of t - 1 (tests/test_isa_lint.py
// checks the compiled code: nothing touches a requested register before the wait that retires it).
template <int TPW>
__global__ __launch_bounds__(256) void stem_mfma64p_kernel(StemArgs a, int nblocks) {
    if (a.zero_u32 && blockIdx.x == 0)
        for (int i = threadIdx.x; i < a.zero_count; i += 256) a.zero_u32[i] = 0u;
    __shared__ __attribute__((aligned(16))) half_t slab[4][32 * 72];
    __shared__ __attribute__((aligned(16))) float sbias[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    int n, bx;
    if (!xcd_image_of(blockIdx.x, nblocks, a.xq, a.n, n, bx)) return;
    if (threadIdx.x < 64) sbias[threadIdx.x] = a.bias[threadIdx.x];
    const int HW = a.h * a.w_;                 // stride 1, pad 1: output and input pixel indices coincide
    float wa[14][2];
    int toff4[14];              // byte offset of the tap relative to the pixel's own position in plane 0
    unsigned need[14];          // borders this tap must not cross: 1 top, 2 bottom, 4 left, 8 right; 16: the zero pad of K = 27 -> 28; 32: a pixel of the image
    float tmean[14], tinv[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) {
        const int t = 2 * i + hh;
        const int tc = min(t, 26);
        const int c = tc / 9, ky = (tc - c * 9) / 3, kx = tc - c * 9 - ky * 3;
        wa[i][0] = t < 27 ? a.w[tc * 64 + r] : 0.f;
        wa[i][1] = t < 27 ? a.w[tc * 64 + 32 + r] : 0.f;
        toff4[i] = 4 * (c * HW + (ky - 1) * a.w_ + (kx - 1));
        need[i] = (ky == 0 ? 1u : 0u) | (ky == 2 ? 2u : 0u) | (kx == 0 ? 4u : 0u) | (kx == 2 ? 8u : 0u) | (t < 27 ? 0u : 16u) | 32u;
        tmean[i] = a.mean[c];
        tinv[i] = a.inv_std[c];
    }
    const __amdgpu_buffer_rsrc_t irs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.img + (size_t)n * 3 * HW), 0, 3 * HW * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)n * HW * 64, 0, HW * 128, 0x00020000);
    half_t* sl = slab[wave];
    const int tile0 = (bx * 4 + wave) * TPW;
    __syncthreads();            // sbias

    // the next tile to request: pixel index (x 4: the byte offset in plane 0), coordinates
    int pu4 = (tile0 * 32 + r) * 4;
    int oy = (tile0 * 32 + r) / a.wo, ox = tile0 * 32 + r - oy * a.wo;
    float raw[2][14];           // taps in flight: tile t + 2 into raw[t & 1] while raw[(t + 1) & 1] waits to be normalised
    unsigned viol[2];           // borders the lane's pixel touches, per requested tile
    float v[2][14];             // normalised taps: tile t in v[t & 1]
    auto request = [&](const int b) {
        viol[b] = (oy == 0 ? 1u : 0u) | (oy == a.h - 1 ? 2u : 0u) | (ox == 0 ? 4u : 0u) | (ox == a.w_ - 1 ? 8u : 0u) | 16u |
                  (oy >= a.h ? 32u : 0u);       // pixels beyond the image (last tiles): every tap "outside", nothing stored
#pragma unroll
        for (int i = 0; i < 14; ++i) {
            const int off = (need[i] & viol[b]) == 0u ? pu4 + toff4[i] : (int)0x80000000;      // out of range: no access
            asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(raw[b][i]) : "v"(off), "s"(irs) : "memory");
        }
        pu4 += 128;
        ox += 32;
        const bool wrap = ox >= a.wo;           // wo >= 32 (launcher): at most one row per step
        ox -= wrap ? a.wo : 0;
        oy += wrap ? 1 : 0;
    };
    // `younger`: vector-memory operations issued behind the requests of this buffer (they may stay in flight)
    auto normalise = [&](const int b, float (&vd)[14], const int younger) {
        float(&q)[14] = raw[b];
#define DN_STEM_TIE "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]), "+v"(q[8]), "+v"(q[9]), \
                    "+v"(q[10]), "+v"(q[11]), "+v"(q[12]), "+v"(q[13])
        switch (younger) {
        case 22: asm volatile("s_waitcnt vmcnt(22)" : DN_STEM_TIE : : "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" : DN_STEM_TIE : : "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" : DN_STEM_TIE : : "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" : DN_STEM_TIE : : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" : DN_STEM_TIE : : "memory"); break;
        }
#undef DN_STEM_TIE
#pragma unroll
        for (int i = 0; i < 14; ++i) vd[i] = (need[i] & viol[b]) == 0u ? (q[i] - tmean[i]) * tinv[i] : 0.f;      // zero padding is applied to the NORMALISED image
    };
    // one quarter of a finished tile's epilogue: lane = pixel r, registers 4g..4g+3 = channels 32 j + 8 g + 4 hh .. +3
    auto epi_quarter = [&](const floatx16& acc0, const floatx16& acc1, const int g) {
        half4 h0, h1;
        const float4 b0 = *reinterpret_cast<const float4*>(sbias + 8 * g + 4 * hh), b1 = *reinterpret_cast<const float4*>(sbias + 32 + 8 * g + 4 * hh);
        const float t8[8] = {acc0[4 * g + 0] + b0.x, acc0[4 * g + 1] + b0.y, acc0[4 * g + 2] + b0.z, acc0[4 * g + 3] + b0.w,
                             acc1[4 * g + 0] + b1.x, acc1[4 * g + 1] + b1.y, acc1[4 * g + 2] + b1.z, acc1[4 * g + 3] + b1.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { h0[e] = (half_t)dn_relu(t8[e]); h1[e] = (half_t)dn_relu(t8[4 + e]); }
        *reinterpret_cast<half4*>(sl + r * 72 + 8 * g + 4 * hh) = h0;
        *reinterpret_cast<half4*>(sl + r * 72 + 32 + 8 * g + 4 * hh) = h1;
    };
    auto epi_store = [&](const int tile) {      // the slab is private to the wave: its own LDS operations are ordered, no barrier
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ch = lane + 64 * u;              // 16-B chunk of the 32 x 128-B tile
            const int row = ch >> 3, q = ch & 7;
            typedef unsigned __attribute__((ext_vector_type(4))) u4;
            const u4 val = *reinterpret_cast<const u4*>(sl + row * 72 + q * 8);
            const int off = (tile * 32 + row) * 128 + q * 16;
            // (s_nop: a store of more than 8 bytes reads its data registers a cycle after it issues; hipcc keeps writers of those registers
            //  away from the stores it knows, an asm statement is opaque to it)
            asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(val), "v"(off), "s"(ors) : "memory");       // rows beyond the image: out of range, dropped
        }
    };
    floatx16 acc[2][2];
    auto step = [&](const int tt) {
        const int p = tt & 1;
        floatx16 &c0 = acc[p][0], &c1 = acc[p][1];
        if (tt + 2 < TPW) request(p);
#pragma unroll
        for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; }
#pragma unroll
        for (int i = 0; i < 14; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[i][0], v[p][i], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[i][1], v[p][i], c1, 0, 0, 0);
            if (tt > 0 && i >= 1 && i <= 10 && (i % 3) == 1) epi_quarter(acc[p ^ 1][0], acc[p ^ 1][1], (i - 1) / 3);
            if (tt > 0 && i == 11) epi_store(tile0 + tt - 1);
            if (tt + 1 < TPW && i == 12) normalise(p ^ 1, v[p ^ 1], (tt >= 2 ? 4 : 0) + (tt + 2 < TPW ? 14 : 0) + (tt >= 1 ? 4 : 0));
        }
        __builtin_amdgcn_sched_barrier(0);          // one scheduling region per step
    };
    request(0);
    if (TPW > 1) request(1);
    normalise(0, v[0], TPW > 1 ? 14 : 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) step(tt);
#pragma unroll
    for (int g = 0; g < 4; ++g) epi_quarter(acc[(TPW - 1) & 1][0], acc[(TPW - 1) & 1][1], g);
    epi_store(tile0 + TPW - 1);
}

// ---- 3x3 stem on the fp16 matrix cores with fp32-grade products: every operand as the sum of two fp16 numbers --------------------------
// Measured (tools/mfma_valu_lab.hip, profiles/r06_lab_mfma_valu.txt): v_mfma_f32_32x32x2_f32 and vector instructions do NOT overlap on
// a SIMD -- not inside a wave, not between waves (one matrix-only + one vector-only wave per SIMD take the SUM of their times; the fp32
// matrix rate equals the packed-fp32 vector rate: the same multipliers) -- while the fp16 matrix instructions run in the shadow of vector work.
// So the fp32 stems pay 1 792 matrix cycles per 32 pixels x 64 channels ON TOP of their vector work (stem_mfma64*_kernel), or 27 x COUT vector
// multiply-adds per pixel (stem3s2_kernel). Here x = xh + xl and w = wh + wl with xh = fp16(x), xl = fp16(x - xh) (x - xh is exact in fp32; the
// pair carries 22+ bits of x, the error of the sum is below 2^-23 |x| until xl goes subnormal, then below 3e-8 absolute), and
//     w x  ~  wl xh + wh xl + wh xh      (the dropped wl xl is below 2^-22 |w x|)
// as three v_mfma_f32_32x32x16_f16 per 16 taps, exact fp16 products, fp32 accumulation: the sum differs from the fp32 kernels' by a few units of
// fp32 rounding, i.e. the fp16 output differs in the last place in ~0.1 % of the values (tests/test_gpu_model.py measures it). 12 x 32 matrix
// cycles per 32 x 64 tile instead of 28 x 64, and they overlap with the vector work. The bias rides in a spare K slot (x = 1).
// Weights and bias are scaled by a power of two (StemArgs::w_scale, chosen by the plan from their largest magnitude; the sums are scaled back
// in the epilogue): small weights -- 0.003 on average in the VGG stem, against inputs up to 130 -- would otherwise have a SUBNORMAL low half
// with a handful of bits (measured: differences of 3e-5 to the fp32 kernel instead of 1e-6).
// K slots: lane (r = lane & 31, kg = lane >> 5) holds 16 of the 32: element e of the lane is tap 15 kg + e -- rows (channel, ky) 0..4 for
// kg = 0, rows 5..8 + the bias slot + zero pads for kg = 1.
// Taps: a lane requests ONE value per row -- its pixel's centre column (stride 2: the aligned pair of columns 2 ox, 2 ox + 1) -- whole 128-byte
// lines per instruction, and takes the kx = 0 / kx = 2 taps from its neighbours' registers (DPP wave_shr:1 / wave_shl:1). The outer lanes of
// the 32 have no neighbour: a tile is NV = 30 (stride 2: 31) output pixels, lanes 1..NV, the outer lanes only feed them. (First form: three
// taps per lane and row as one 12-byte request -- 54 cycles of the texture-address path per instruction, as slow as the stores: 143 us per 16
// images of 512 x 512, 85 without the requests OR without the stores.)
// Pipeline per wave: straight-line over TPW tiles; tile t + 2 is requested (asm: hipcc would sink the loads to their use) when tile t starts;
// the hand-written wait for tile t's taps leaves the stores of tiles t - 2 and t - 1 in flight (loads and stores retire in order, so a wait for
// a request is also a wait for every older store: with the request only one tile ahead the launch ran at the pace of its store round trips --
// 135 us per 16 images of 512 x 512, 84 without the requests OR without the stores; tests/test_isa_lint.py checks the counts).
template <int S> struct StemTaps { typedef float type; };            // what a lane requests per row: the centre column,
template <> struct StemTaps<2> { typedef float type __attribute__((ext_vector_type(2))); };     // stride 2: the pair (2 ox, 2 ox + 1)

template <int COUT, int S, int ACT, int TPW, int D, bool TOUCH>
__global__ __launch_bounds__(256) void stem_split_kernel(StemArgs a, int nblocks) {
    constexpr int MT = (COUT + 31) / 32;         // 32-channel tiles (16 channels: the upper half of the one tile is zero weights)
    constexpr int SW = COUT + 8;                 // slab row stride (halves)
    constexpr int NV = S == 1 ? 30 : 31;         // output pixels per tile: lanes 1..NV
    constexpr int NS = (NV * COUT / 8 + 63) / 64;        // store instructions (16 bytes per lane) per tile
    if (a.zero_u32 && blockIdx.x == 0)
        for (int i = threadIdx.x; i < a.zero_count; i += 256) a.zero_u32[i] = 0u;
    __shared__ __attribute__((aligned(16))) half_t slab[4][32 * SW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, kg = lane >> 5;
    int n, bx;
    if (!xcd_image_of(blockIdx.x, nblocks, a.xq, a.n, n, bx)) return;
    const int HW = a.h * a.w_, OHW = a.ho * a.wo;
    // weights: A fragments (channel r of tile mt, this lane's 16 K slots), split
    half8 ah[MT][2], al[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ch = 32 * mt + r;
            const int tap = 15 * kg + e;
            const bool is_tap = kg ? e < 12 : e < 15, is_bias = kg && e == 12;
            float wv = 0.f;
            if (ch < COUT) wv = is_tap ? a.w[min(tap, 26) * COUT + ch] : is_bias ? a.bias[ch] : 0.f;
            wv *= a.w_scale;            // (a power of two: exact) the low halves of all but negligible weights stay in fp16's normal range
            const half_t h = (half_t)wv;
            ah[mt][e >> 3][e & 7] = h;
            al[mt][e >> 3][e & 7] = (half_t)(wv - (float)h);
        }
    // the lane's five rows q (row = 5 kg + q = channel * 3 + ky; row 9 does not exist): byte offset of the row's centre tap relative to the
    // pixel's own centre tap in plane 0, ky - 1, normalisation constants
    int rowoff4[5], kyq[5];
    float qmean[5], qinv[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int row = min(5 * kg + q, 8);
        const int c = row / 3, ky = row - 3 * c;
        rowoff4[q] = 4 * (c * HW + (ky - 1) * a.w_);
        kyq[q] = 5 * kg + q < 9 ? ky - 1 : (1 << 20);          // the missing row is always "outside"
        qmean[q] = a.mean[c];
        qinv[q] = a.inv_std[c];
    }
    const __amdgpu_buffer_rsrc_t irs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.img + (size_t)n * 3 * HW), 0, 3 * HW * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)n * OHW * COUT, 0, OHW * COUT * 2, 0x00020000);
    half_t* sl = slab[wave];
    const int tile0 = (bx * 4 + wave) * TPW;
    typedef typename StemTaps<S>::type fS;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));

    // the next tile to request: this lane's pixel tile * NV + r - 1 (lane 0: the pixel in front of the tile; it may be pixel -1)
    int oy, ox;
    {
        const int p0 = tile0 * NV + r - 1;
        oy = p0 < 0 ? -1 : p0 / a.wo;
        ox = p0 - oy * a.wo;            // (pixel -1 = row -1, last column)
    }
    fS raw[D][5];               // tile t's taps in raw[t % D]: requested D tiles ahead, so that a tile's stores have D more tiles to retire in
    int noy[D], nox[D];         // the lane's pixel of the tile in raw[b]
    auto request = [&](const int b) {
        noy[b] = oy; nox[b] = ox;
        const int base4 = 4 * S * (oy * a.w_ + ox);
        const bool pix = (unsigned)oy < (unsigned)a.ho;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int iy = S * oy + kyq[q];
            const int off = ((unsigned)iy < (unsigned)a.h && pix) ? base4 + rowoff4[q] : (int)0x80000000;      // out of range: no access, zeros
            if (S == 1) asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(raw[b][q]) : "v"(off), "s"(irs) : "memory");
            else asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(raw[b][q]) : "v"(off), "s"(irs) : "memory");
        }
        ox += NV;
        const bool wrap = ox >= a.wo;           // wo >= 32 (launcher): at most one row per step
        ox -= wrap ? a.wo : 0;
        oy += wrap ? 1 : 0;
    };
    auto lane_below = [](const float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false)); };     // wave_shr:1
    auto lane_above = [](const float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, false)); };     // wave_shl:1
    half8 xh[2], xl[2];
    // `younger`: vector-memory operations issued behind this buffer's requests -- they may stay in flight (loads and stores retire in order)
    auto normalise_split = [&](const int b, const int younger) {
        fS(&w)[5] = raw[b];
#define DN_STEM_WAIT(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]) : : "memory"); break;
        switch (younger) {
            DN_STEM_WAIT(1) DN_STEM_WAIT(2) DN_STEM_WAIT(3) DN_STEM_WAIT(4) DN_STEM_WAIT(5) DN_STEM_WAIT(6) DN_STEM_WAIT(7) DN_STEM_WAIT(8)
            DN_STEM_WAIT(9) DN_STEM_WAIT(10) DN_STEM_WAIT(11) DN_STEM_WAIT(12) DN_STEM_WAIT(13) DN_STEM_WAIT(14) DN_STEM_WAIT(15) DN_STEM_WAIT(16)
            DN_STEM_WAIT(17) DN_STEM_WAIT(18) DN_STEM_WAIT(19) DN_STEM_WAIT(20) DN_STEM_WAIT(21) DN_STEM_WAIT(22) DN_STEM_WAIT(23) DN_STEM_WAIT(24)
            DN_STEM_WAIT(25) DN_STEM_WAIT(26) DN_STEM_WAIT(27) DN_STEM_WAIT(28) DN_STEM_WAIT(29) DN_STEM_WAIT(30) DN_STEM_WAIT(31)
            default: asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]) : : "memory"); break;
        }
#undef DN_STEM_WAIT
        static_assert(NS == 1 || NS == 2 || NS == 4, "stem_split_kernel: store count");
        const bool left = nox[b] > 0, right = S == 2 || nox[b] + 1 < a.w_;
        float x[16];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const bool yok = (unsigned)(S * noy[b] + kyq[q]) < (unsigned)a.h && (unsigned)noy[b] < (unsigned)a.ho;
            float t0, t1, t2;           // taps kx = 0, 1, 2: columns S ox - 1, S ox, S ox + 1
            if constexpr (S == 1) { t0 = lane_below(w[q]); t1 = w[q]; t2 = lane_above(w[q]); }
            else { t0 = lane_below(w[q].y); t1 = w[q].x; t2 = w[q].y; }
            // zero padding is applied to the NORMALISED image (transform, then conv)
            const float n0 = (t0 - qmean[q]) * qinv[q], n1 = (t1 - qmean[q]) * qinv[q], n2 = (t2 - qmean[q]) * qinv[q];
            x[3 * q + 0] = (yok && left) ? n0 : 0.f;
            x[3 * q + 1] = yok ? n1 : 0.f;
            x[3 * q + 2] = (yok && right) ? n2 : 0.f;
        }
        x[15] = 0.f;
        x[12] = kg ? 1.0f : x[12];              // the bias slot
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const half_t h = (half_t)x[e];
            xh[e >> 3][e & 7] = h;
            xl[e >> 3][e & 7] = (half_t)(x[e] - (float)h);
        }
    };
    // Every line of the image the wave's tiles will ask for is touched once up front (two requests per lane into a register nobody reads: the
    // oldest vector-memory operations of the wave, so no wait ever depends on them): behind a chip-wide stream of stores a first touch
    // takes longer than the two tiles a request runs ahead.
    float sink[2] = {0.f, 0.f};         // (kept allocated until the first wait has retired the two requests: tile loop)
    if (TOUCH) {
        const int p0 = tile0 * NV - 1;                                   // the wave's first pixel
        const int y0 = p0 < 0 ? 0 : p0 / a.wo, x0 = p0 < 0 ? 0 : p0 - y0 * a.wo;
        const int span = (TPW * NV + 2) * S * 4;                         // bytes of one input row the wave's pixels cover (rows wrap: the touch is approximate)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int l = lane + 64 * u, row = l >> 3, seg = l & 7;      // 9 (channel, ky) rows x 8 segments
            const int c = row / 3, ky = row - 3 * c;
            const int iy = S * y0 + ky - 1;
            const int off = (row < 9 && (unsigned)iy < (unsigned)a.h && seg * ((span + 7) / 8) < span) ? 4 * (c * HW + iy * a.w_ + S * x0) + seg * ((span + 7) / 8) : (int)0x80000000;
            asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(sink[u]) : "v"(off), "s"(irs) : "memory");
        }
    }
    // order of the vector-memory operations: R(0) .. R(D - 1) | tile 0: R(D) S(0) | tile 1: R(D + 1) S(1) | ...  -- behind R(k) when tile k starts:
    // the requests k + 1 .. k + D - 1 that exist and the stores of the min(k, D) tiles before it
    static_assert(D >= 1 && 5 * (D - 1) + NS * D <= 31, "stem_split_kernel: wait count");
#pragma unroll
    for (int k = 0; k < D && k < TPW; ++k) request(k);
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        normalise_split(tt % D, 5 * ((tt + D - 1 < TPW - 1 ? tt + D - 1 : TPW - 1) - tt) + NS * (tt < D ? tt : D));
        if (tt == 0) asm volatile("" : "+v"(sink[0]), "+v"(sink[1]) : : "memory");      // the touch requests are older than tile 0's taps: retired by now
        if (tt + D < TPW) request(tt % D);
        __builtin_amdgcn_sched_barrier(0);          // the requests stay in front of the work they overlap with
        floatx16 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][e] = 0.f;
#pragma unroll
            for (int k = 0; k < 2; ++k) {              // the small terms first
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt][k], xh[k], acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt][k], xl[k], acc[mt], 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt][k], xh[k], acc[mt], 0, 0, 0);
        }
        // lane = pixel r, registers 4g..4g+3 of tile mt = channels 32 mt + 8 g + 4 kg .. +3
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < (COUT >= 32 ? 4 : 2); ++g) {
                float t4[4] = {acc[mt][4 * g + 0] * a.w_unscale, acc[mt][4 * g + 1] * a.w_unscale, acc[mt][4 * g + 2] * a.w_unscale,
                               acc[mt][4 * g + 3] * a.w_unscale};
                dn_act_n<float[4], 4>(t4, ACT);
                half4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = (half_t)t4[e];
                *reinterpret_cast<half4*>(sl + r * SW + 32 * mt + 8 * g + 4 * kg) = hv;
            }
        // the slab is private to the wave: its own LDS operations are ordered, no barrier. Slab rows 1..NV are the tile's pixels.
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int ch = lane + 64 * u;                   // 16-B chunk of the NV x (2 COUT)-byte tile
            const int row = ch / (COUT / 8), q = ch % (COUT / 8);
            const u4 val = *reinterpret_cast<const u4*>(sl + min(row + 1, 31) * SW + q * 8);
            const int off = row >= NV ? (int)0x80000000 : ((tile0 + tt) * NV + row) * (COUT * 2) + q * 16;
            // (s_nop: a store of more than 8 bytes reads its data registers a cycle after it issues; hipcc keeps writers of those registers
            //  away from the stores it knows, an asm statement is opaque to it)
            asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(val), "v"(off), "s"(ors) : "memory");       // pixels beyond the image: out of range, dropped
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int COUT, int K>
int launch_stem_t(const StemArgs& a, hipStream_t s) {
    const int images = a.xq > 0 ? 8 * a.xq : a.n;      // image slots of the launch (XCD grouping: 8 groups of xq)
    if (K == 3 && a.pad == 1 && a.split_ok && dn_knob("DN_STEM_SPLIT", 1) && a.wo >= 32 && (long)3 * a.h * a.w_ < (1L << 28) &&
        (long)a.ho * a.wo * COUT < (1L << 29)) {
        // requests run 3 tiles ahead and the wave touches its lines up front for the 64-channel stem (1 GB of stores per forward: a first touch of an
        // image line behind that stream outlasts two tiles; 16 images of 512 x 512, same box: 145 us -> 126 with the touch, 128 with 3 tiles
        // ahead, 121 - 126 with both, 123 with 4), 2 tiles ahead and no touch for the narrow ones (17.6 / 36.3 us; with the touch 19.1 / 37.2).
        // tiles per wave: 8 for the 64-channel stem (the weights' split -- 64 values per lane -- once per 8 tiles), 4 for the narrow ones (their
        // launches are small: more, shorter waves; measured 4 / 8 / 16: 17.5 / 19.3 / 19.0 us for 16 channels, 36.2 / 38.0 / 38.4 for 32,
        // 147.5 / 144.8 / 144.6 for 64)
        constexpr int TPW = COUT >= 64 ? 8 : 4;
        const int nv = a.stride == 1 ? 30 : 31;
        const int tiles = dn_cdiv((long)a.ho * a.wo, nv);
        const int nblocks = dn_cdiv(tiles, 4 * TPW);
        const dim3 grid(nblocks * images);
#define DN_STEM_SPLIT_CASE(C, S_, ACT_)                                                                                                    \
        if (COUT == C && a.stride == S_ && a.act == ACT_ && (S_ == 1 ? (a.ho == a.h && a.wo == a.w_) : ((a.w_ & 1) == 0 && 2 * a.wo == a.w_ && a.ho == (a.h + 1) / 2))) { \
            dn_note_kernel("stem_split_kernel<%d,%d>", C, S_);                                                                             \
            hipLaunchKernelGGL((stem_split_kernel<C, S_, ACT_, (C >= 64 ? 8 : 4), (C >= 64 ? 3 : 2), (C >= 64)>), grid, dim3(256), 0, s, a, nblocks); \
            return DN_OK;                                                                                                                  \
        }
        DN_STEM_SPLIT_CASE(64, 1, DN_ACT_RELU)
        DN_STEM_SPLIT_CASE(16, 2, DN_ACT_HSWISH)
        DN_STEM_SPLIT_CASE(32, 2, DN_ACT_RELU6)
#undef DN_STEM_SPLIT_CASE
    }
    if (K == 3 && a.stride == 2 && a.pad == 1 && (a.w_ & 1) == 0 && 2 * a.wo == a.w_) {
        dn_note_kernel("stem3s2_kernel<%d>", COUT);
        const int nblocks = dn_cdiv((long)a.ho * a.wo, 256);
        hipLaunchKernelGGL((stem3s2_kernel<COUT>), dim3(nblocks * images), dim3(256), 0, s, a, nblocks);
        return DN_OK;
    }
    const int mf = dn_knob("DN_STEM_MFMA", 1);
    if (mf && K == 3 && COUT == 64 && a.stride == 1 && (long)3 * a.h * a.w_ < (1L << 30)) {
        const int tiles = dn_cdiv((long)a.ho * a.wo, 32), per_wave = 8;
        if (dn_knob("DN_STEM_PIPE", 1) && a.pad == 1 && a.act == DN_ACT_RELU && a.ho == a.h && a.wo == a.w_ && a.wo >= 32 && (long)a.h * a.w_ < (1L << 24)) {
            dn_note_kernel("stem_mfma64p_kernel");
            const int nblocks = dn_cdiv(tiles, 4 * per_wave);
            hipLaunchKernelGGL((stem_mfma64p_kernel<8>), dim3(nblocks * images), dim3(256), 0, s, a, nblocks);
            return DN_OK;
        }
        dn_note_kernel("stem_mfma64_kernel");
        const int nblocks = dn_cdiv(tiles, 4 * per_wave);
        hipLaunchKernelGGL(stem_mfma64_kernel, dim3(nblocks * images), dim3(256), 0, s, a, tiles, per_wave, nblocks);
        return DN_OK;
    }
    dn_note_kernel("stem_kernel<%d,%d>", COUT, K);
    const int nblocks = dn_cdiv((long)a.ho * a.wo, 256);
    hipLaunchKernelGGL((stem_kernel<COUT, K>), dim3(nblocks * images), dim3(256), 0, s, a, nblocks);
    return DN_OK;
}

}  // namespace

int launch_depthwise(const DwArgs& a, hipStream_t s) {
    DN_REQUIRE(a.c % 8 == 0, "depthwise: c=%d must be a multiple of 8", a.c);
    DN_REQUIRE(!a.pool || a.c / 8 <= 256, "depthwise: pooled channel groups %d > 256", a.c / 8);
    if (a.k == 3 && a.stride == 1) return launch_dw<3, 1, 4>(a, s);
    if (a.k == 3 && a.stride == 2) return launch_dw<3, 2, 2>(a, s);
    if (a.k == 5 && a.stride == 1) return launch_dw<5, 1, 4>(a, s);
    if (a.k == 5 && a.stride == 2) return launch_dw<5, 2, 2>(a, s);
    dn_set_error("depthwise: unsupported k=%d stride=%d", a.k, a.stride);
    return DN_E_UNSUPPORTED;
}

int launch_depthwise_group(const DwArgs* arr, int count, hipStream_t s) {
    DN_REQUIRE(count >= 1 && count <= 12, "depthwise group: %d problems", count);
    for (int i = 0; i < count; ++i) {
        DN_REQUIRE(arr[i].c % 8 == 0 && !arr[i].pool && arr[i].k == arr[0].k && arr[i].stride == arr[0].stride && arr[i].n == arr[0].n,
                   "depthwise group: problem %d is not compatible", i);
    }
    if (arr[0].k == 3 && arr[0].stride == 1) return launch_dw_group<3, 1, 4>(arr, count, s);
    if (arr[0].k == 3 && arr[0].stride == 2) return launch_dw_group<3, 2, 2>(arr, count, s);
    if (arr[0].k == 5 && arr[0].stride == 1) return launch_dw_group<5, 1, 4>(arr, count, s);
    return launch_dw_group<5, 2, 2>(arr, count, s);
}

int depthwise_pool_blocks(const DwArgs& a) {
    if (a.k == 3 && a.stride == 1) return dw_blocks<3, 1, 4>(a);
    if (a.k == 3 && a.stride == 2) return dw_blocks<3, 2, 2>(a);
    if (a.k == 5 && a.stride == 1) return dw_blocks<5, 1, 4>(a);
    return dw_blocks<5, 2, 2>(a);
}

bool depthwise_se_tail_supported(int c, int squeeze) { return c % 8 == 0 && squeeze % 8 == 0 && c <= 1024 && squeeze <= 256 && c >= 8 && squeeze >= 8; }

int launch_se_fc(const float* partial, int nblk, const void* w1t, const float* b1, const void* w2t, const float* b2, float* scale,
                 int n, int c, int squeeze, int pool_pixels, hipStream_t s, int xq) {
    DN_REQUIRE(c <= 1024 && squeeze <= 256 && c % 2 == 0 && squeeze % 2 == 0, "se: c=%d squeeze=%d outside the kernel's range (even, <= 1024 / 256)", c, squeeze);
    if (c % 8 == 0 && squeeze % 8 == 0 && c >= 8 && squeeze >= 8 && dn_knob("DN_SE_FC8", 1)) {
        const int C4 = c >> 2, RS = std::max(1, std::min(1024 / C4, nblk));
        const int KS1 = std::max(1, std::min(1024 / (squeeze >> 3), c >> 3)), KS2 = std::max(1, std::min(1024 / (c >> 3), squeeze >> 3));
        const size_t pf = (size_t)std::max(std::max(RS * c, KS1 * squeeze), KS2 * c);
        dn_note_kernel("se_fc8_kernel");
        const int per1 = dn_cdiv(c, KS1), per2 = dn_cdiv(squeeze, KS2);
        const bool wide = (per1 > 14 || per2 > 14) && nblk <= 4 * RS;          // two batches of 15 instead of three of 14
        auto k = wide ? se_fc8_kernel<15, 4> : se_fc8_kernel<14, 6>;
        hipLaunchKernelGGL(k, dim3(xq > 0 ? 8 * xq : n), dim3(1024), (size_t)(c + squeeze + pf) * sizeof(float), s, partial, nblk,
                           reinterpret_cast<const half_t*>(w1t), b1, reinterpret_cast<const half_t*>(w2t), b2, scale, c, squeeze,
                           1.0f / (float)pool_pixels, g_se_stamps, n, xq, 5);
        return DN_OK;
    }
    dn_note_kernel("se_fc_kernel");
    hipLaunchKernelGGL(se_fc_kernel, dim3(xq > 0 ? 8 * xq : n), dim3(1024), (size_t)(c + squeeze + 2048) * sizeof(float), s, partial, nblk,
                       reinterpret_cast<const unsigned*>(w1t), b1, reinterpret_cast<const unsigned*>(w2t), b2, scale, c, squeeze,
                       1.0f / (float)pool_pixels, g_se_stamps, n, xq);
    return DN_OK;
}

int launch_stem(const StemArgs& a, hipStream_t s) {
    if (a.k == 3 && a.cout == 16) return launch_stem_t<16, 3>(a, s);
    if (a.k == 3 && a.cout == 32) return launch_stem_t<32, 3>(a, s);
    if (a.k == 3 && a.cout == 64) return launch_stem_t<64, 3>(a, s);
    dn_set_error("stem: unsupported k=%d cout=%d", a.k, a.cout);
    return DN_E_UNSUPPORTED;
}
