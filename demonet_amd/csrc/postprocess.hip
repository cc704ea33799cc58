// SSD post-process on the GPU: softmax -> box decode -> clip -> per-class (score > thr, top-k) -> hard NMS
// -> global top-D by score -> rescale to the original image size.
//
// reference ops replaced: SSD.postprocess_detections (generalized_ssd.py:351-397), BoxCoder.decode_single
//   (_utils.py:187-224), torchvision clip_boxes_to_image / batched_nms (third-party; per-class, IoU > thr),
//   GeneralizedRCNNTransform.postprocess / resize_boxes (transform.py:228-247,278-292).
// The reference runs 90 python iterations per image (mask, topk, index) and then NMS over up to 27 000 boxes.
// Here: three launches for the whole batch.
//   P1 softmax_decode : 64-anchor tiles through LDS; scores written class-major ([n][K-1][A]) so that P2 streams
//                       one contiguous column per (image, class); boxes decoded in fp32 in the reference's op order.
//   P2 select_nms     : one 256-thread workgroup per (image, class): column in LDS, 4x8-bit radix select of the
//                       top-k threshold, ordered compaction, bitonic sort on (score, anchor) keys, 64x64-bit IoU
//                       mask matrix in LDS, wave-serial greedy reduce.
//   P3 merge          : one 1024-thread workgroup per image: radix select of the D-th score over all kept
//                       candidates, ordered compaction, sort, gather + rescale.
// All ordering decisions are integer/compare work on the fp32 scores, with the canonical tie-break of the oracle
// (score desc, class asc, anchor asc): bit-exact indices whenever scores/boxes agree.
// This file is compiled with -ffp-contract=off: decode and IoU must round like the reference (no FMA fusion).
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include "common.h"
#include "post_math.h"

#ifdef DN_DEV_STAMPS
static long long* g_pp_stamps = nullptr;     // dev build only: per-workgroup phase stamps of select_nms (tools/probe_pp_stamps.py)
extern "C" __attribute__((visibility("default"))) void dn_debug_pp_stamps(void* dev_ptr) { g_pp_stamps = (long long*)dev_ptr; }
#else
constexpr long long* g_pp_stamps = nullptr;
#endif

namespace {

#ifdef DN_DEV_STAMPS
#define PP_STAMP(k) do { if (stamps && tid == 0) stamps[(size_t)blockIdx.x * 16 + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PP_STAMP(k) do { } while (0)
#endif

constexpr int HSHIFT = DN_PP_HSHIFT;                     // score histogram: float bits 30..19 (8 exponent + 4 mantissa bits)
constexpr int HBINS = DN_PP_HBINS;                       // bins kept: the top 256 (scores down to 2^-16); anything lower shares bin 0

// ------------------------------------------------------------------------------------------------------------
// P1
// ------------------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------------------
// Cut-off (the body of tau_kernel; since round 5 also run by the LAST softmax tile of an image when the head launch produced the scores of
// the large levels, TauArgs::tickets): only candidates that can reach the global top-D matter. tau = lower edge of the highest histogram
// bin whose suffix count reaches `want` (a multiple of D). Greedy NMS restricted to the score >= tau prefix of each class is exact for
// that prefix; if at least D boxes survive, every box with score < tau ranks below them and cannot appear in the output. Otherwise
// needFull[n] triggers the full path.
// ------------------------------------------------------------------------------------------------------------
struct TauArgs {
    const unsigned* phist; int tiles, hb0, clamped; unsigned want;
    unsigned* tauKey; int* needFull; int nimg, xq, nb, Km1; int* order; int* fbcnt; HistRows hr;
};
template <bool AGENT>     // AGENT: the extra rows (softmax tiles of THIS launch, other workgroups) are read with agent-scope loads
__device__ __forceinline__ void tau_body(const TauArgs& t, const int n, unsigned* __restrict__ part) {
    const int tid = threadIdx.x;
    const int nb = t.nb, Km1 = t.Km1;
    const HistRows& hr = t.hr;
    unsigned s = 0;
    if (hr.levels == 0) {
        // thread t owns bin t: sum of the per-workgroup rows of softmax_decode_kernel (fixed order, 16 loads in flight)
        const unsigned* h = t.phist + ((size_t)n * t.tiles << 8) + tid;
        for (int t0 = 0; t0 < t.tiles; t0 += 16) {
            unsigned v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = h[(size_t)min(t0 + u, t.tiles - 1) << 8];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += (t0 + u < t.tiles) ? v[u] : 0u;
        }
    } else {
        // rows of head_fused_kernel's epilogue: per level, the 32-pixel half tiles that touch this image, in slots sbase .. (HistRows). Exactly
        // the slots counted here were written in this forward; integer sums, so the order does not matter.
        const unsigned* h = t.phist + (size_t)n * hr.rows_per_image * 256 + tid;
        for (int l = 0; l < hr.levels; ++l) {
            const int nl = hr.grouped[l] ? n - (n / t.xq) * t.xq : n;                // image index inside its XCD group's pixel range
            const int cnt = (((nl + 1) * hr.hw[l] - 1) >> 5) - ((nl * hr.hw[l]) >> 5) + 1;
            const unsigned* hl = h + (size_t)hr.sbase[l] * 256;
            for (int t0 = 0; t0 < cnt; t0 += 8) {
                unsigned v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = hl[(size_t)min(t0 + u, cnt - 1) << 8];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += (t0 + u < cnt) ? v[u] : 0u;
            }
        }
        for (int q = 0; q < hr.extra_rows; ++q) {
            const unsigned* e = h + ((size_t)(hr.extra_base + q) << 8);
            s += AGENT ? __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *e;
        }
    }
    part[tid] = s;
    __syncthreads();
    unsigned above = 0;
    for (int q = tid + 1; q < nb; ++q) above += part[q];
    // the unique thread where the suffix count (from the top bin down) crosses `want`: tau = lower edge of its bin. Bin 0 of a
    // clamped table also holds every lower score -> tau 0 (take everything above the score threshold).
    if (tid < nb && above < t.want && above + s >= t.want) t.tauKey[n] = (t.clamped && tid == 0) ? 0u : (unsigned)(t.hb0 + tid) << HSHIFT;
    // classes by descending count of passing scores (ties: ascending class): the per-class workgroups of the next launch are
    // dispatched in this order, the heavy ones first, so the launch does not end on a late-started heavy class. No counts
    // (the row has no room): identity.
    if (nb + Km1 <= HBINS) {
        if (tid >= nb && tid < nb + Km1) {
            int rank = 0;
            for (int q = nb; q < nb + Km1; ++q) rank += (part[q] > s || (part[q] == s && q < tid)) ? 1 : 0;
            t.order[(size_t)n * Km1 + rank] = tid - nb;
        }
    } else {
        for (int c = tid; c < Km1; c += 256) t.order[(size_t)n * Km1 + c] = c;
    }
    if (tid == 0) {
        unsigned total = 0;
        for (int q = 0; q < nb; ++q) total += part[q];
        if (total < t.want) t.tauKey[n] = 0u;    // fewer passing scores than wanted: take everything
        t.needFull[n] = 0;
        t.fbcnt[n] = 0;                          // ticket of the fused fallback merge (select_nms_kernel)
    }
}

// One tile of 64 anchors [a0, a0 + na) of image n: softmax over the classes, scores out class-major, score histogram + per-class counts of the
// passing scores ADDED to lhist (the caller zeroes and flushes it), boxes decoded. tile: [64][K] floats + rowsum[64] of LDS. Ends with its LDS
// reads done only after the caller's next barrier.
template <bool PERM>      // PERM: scores stored anchor-major within a level (PostLevels); false: canonical order, no index code at all
__device__ __forceinline__ void softmax_decode_tile(float* __restrict__ tile, float* __restrict__ rowsum, unsigned* __restrict__ lhist, int* __restrict__ pidx,
                                                    const float* __restrict__ logits, const float* __restrict__ reg, const float* __restrict__ anchors,
                                                    float* __restrict__ scoresT, float4* __restrict__ boxes, const int A, const int K, const float img_w,
                                                    const float img_h, const float score_thr, const int hb0, const int nb, const int ccb,
                                                    long long* __restrict__ stamps, const int n, const int a0, const PostLevels& lv) {
    const int tid = threadIdx.x;
    const int na = min(64, A - a0);
    constexpr bool ident = !PERM;
    if (!ident && tid < 64) pidx[tid] = post_perm(lv, min(a0 + tid, A - 1));      // (visible after the barrier behind the tile load)
    const float* src = logits + ((size_t)n * A + a0) * K;
    const int total = na * K;
    {
        // 23 KB per workgroup: 8 independent 8-byte loads in flight per thread (rows of K floats are only 8-byte aligned when
        // A*K is even; odd products fall back to 4-byte loads). A plain copy loop exposes one memory round trip per iteration.
        const bool pair_ok = ((reinterpret_cast<size_t>(src) & 7) == 0) && ((total & 1) == 0);
        if (pair_ok) {
            const float2* s2 = reinterpret_cast<const float2*>(src);
            float2* t2 = reinterpret_cast<float2*>(tile);
            const int n2 = total >> 1;
            for (int i0 = tid; i0 < n2; i0 += 256 * 8) {
                float2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = s2[min(i0 + 256 * u, n2 - 1)];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (i0 + 256 * u < n2) t2[i0 + 256 * u] = v[u];
            }
        } else {
            for (int i0 = tid; i0 < total; i0 += 256 * 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[min(i0 + 256 * u, total - 1)];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (i0 + 256 * u < total) tile[i0 + 256 * u] = v[u];
            }
        }
    }
    __syncthreads();
    PP_STAMP(9);
    {
        const int row = tid >> 2, sub = tid & 3;
        const float sm = pp_softmax_row(tile + row * K, K, sub, row < na);
        if (sub == 0) rowsum[row] = pp_row_rcp(sm);       // (the reciprocal: scores are e * (1 / sum), post_math.h)
    }
    __syncthreads();
    PP_STAMP(10);
    const int Km1 = K - 1;
    for (int idx = tid; idx < Km1 * 64; idx += 256) {
        const int k = 1 + (idx >> 6), a = idx & 63;
        bool pass = false;
        if (a < na) {
            const float sc = pp_score(tile[a * K + k], rowsum[a]);
            scoresT[((size_t)n * Km1 + (k - 1)) * A + (ident ? a0 + a : pidx[a])] = sc;
            pass = sc > score_thr;
            if (pass) atomicAdd(&lhist[pp_hist_bin(sc, hb0, nb)], 1u);
        }
        // a wave's 64 lanes hold the 64 anchors of ONE class per iteration, and no other wave or iteration of this tile sees that class
        const unsigned long long m = __ballot(pass);
        if (ccb >= 0 && a == 0) lhist[ccb + k - 1] += (unsigned)__popcll(m);
    }
    if (tid < na) {
        const int a = a0 + tid;
        const float4 rg = reinterpret_cast<const float4*>(reg)[(size_t)n * A + a];
        const float4 an = reinterpret_cast<const float4*>(anchors)[a];
        boxes[(size_t)n * A + a] = pp_decode_box(rg, an, img_w, img_h);
    }
}

template <bool PERM>
__global__ __launch_bounds__(256) void softmax_decode_kernel(const float* __restrict__ logits, const float* __restrict__ reg,
                                                            const float* __restrict__ anchors, float* __restrict__ scoresT,
                                                            float4* __restrict__ boxes, int A, int K, float img_w, float img_h,
                                                            float score_thr, unsigned* __restrict__ phist, int hb0, int nb,
                                                            long long* __restrict__ stamps, int nimg, int tiles, int xq, PostLevels lv,
                                                            int a_base, int row_stride, int row_base, unsigned* __restrict__ tickets, TauArgs tau) {
    // the launch covers the anchors [a_base, a_base + 64 tiles) of every image; a tile's histogram goes to row row_base + tile of the image's row_stride rows
    extern __shared__ float tile[];            // [64][K] then rowsum[64] then hist[HBINS]
    float* rowsum = tile + 64 * K;
    unsigned* lhist = reinterpret_cast<unsigned*>(rowsum + 64);
    __shared__ int pidx[64];                   // stored (anchor-major within the level) index of the tile's anchors
    // only bins [hb0, hb0 + nb) can be hit: scores lie in (score_thr, 1] (161 bins for score_thr = 0.001; nb <= HBINS)
    // bins [nb, nb + K-1), when they fit the 256-entry row, count each class's passing scores: tau_kernel ranks the classes by them
    const int ccb = (nb + K - 1 <= HBINS) ? nb : -1;
    if (threadIdx.x < nb + (ccb >= 0 ? K - 1 : 0)) lhist[threadIdx.x] = 0u;
    [[maybe_unused]] const int tid = threadIdx.x;       // (the stamp macro of dev builds)
    int n, atile;                              // flat grid [image slot][anchor tile] (XCD grouping: common.h)
    if (!xcd_image_of(blockIdx.x, tiles, xq, nimg, n, atile)) return;
    PP_STAMP(8);
    softmax_decode_tile<PERM>(tile, rowsum, lhist, pidx, logits, reg, anchors, scoresT, boxes, A, K, img_w, img_h, score_thr, hb0, nb, ccb, stamps, n, a_base + atile * 64, lv);
    __syncthreads();
    PP_STAMP(11);
    // this workgroup's histogram goes to its own row; tau_kernel adds the rows. (Device-scope atomics into one per-image table
    // made a few workgroups per launch wait 15-25 us on the hot bins: the kernel's whole tail.)
    unsigned* const rowp = phist + (((size_t)n * row_stride + row_base + atile) << 8) + threadIdx.x;
    const unsigned rowv = (threadIdx.x < nb + (ccb >= 0 ? K - 1 : 0)) ? lhist[threadIdx.x] : 0u;
    if (!tickets) {
        *rowp = rowv;
    } else {
        // The cut-off of the image in the LAST of its softmax tiles to finish (round 5: one dependent launch less behind the head launch): the row
        // goes out write-through (agent-scope store), every storing thread waits for its acknowledgement, the barrier orders those waits before
        // thread 0's relaxed ticket, and the workgroup that draws the last ticket reads the other tiles' rows with agent-scope loads -- the
        // hand-over of depthwise.hip's squeeze-excitation tail, no cache-flushing fence. The head launch's rows come from an EARLIER launch.
        // Counters: zeroed by the stem launch of every forward, left at zero here.
        __hip_atomic_store(rowp, rowv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __shared__ int s_last;
        __syncthreads();
        if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(&tickets[n], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(tiles - 1);
        __syncthreads();
        if (s_last) {
            tau_body<true>(tau, n, lhist);          // (lhist: 256 words of LDS nobody needs any more)
            if (threadIdx.x == 0) tickets[n] = 0u;
        }
    }
    PP_STAMP(12);
}

// ------------------------------------------------------------------------------------------------------------
// block-level helpers shared by P2 and P3
// ------------------------------------------------------------------------------------------------------------
// Finds the radix digit where the count of larger keys crosses `need` (scan of 256 bins from the top by wave 0).
// hist[] must be complete (barrier before). Results via sh[0] = digit, sh[1] = remaining need within that bin.
__device__ __forceinline__ void radix_pick_digit(const unsigned* hist, unsigned need, unsigned* sh) {
    if (threadIdx.x < 64) {
        const int l = threadIdx.x;
        // lane l owns bins 255-4l .. 252-4l (descending)
        unsigned h0 = hist[255 - 4 * l], h1 = hist[254 - 4 * l], h2 = hist[253 - 4 * l], h3 = hist[252 - 4 * l];
        unsigned s = h0 + h1 + h2 + h3;
        unsigned incl = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            unsigned t = __shfl_up(incl, d);
            if (l >= d) incl += t;
        }
        const unsigned long long ball = __ballot(incl >= need);
        const int first = __ffsll((long long)ball) - 1;     // ball != 0 because total >= need
        if (l == first) {
            unsigned before = incl - s;      // count in bins above this lane's
            unsigned digit, rem;
            if (before + h0 >= need) { digit = 255 - 4 * l; rem = need - before; }
            else if (before + h0 + h1 >= need) { digit = 254 - 4 * l; rem = need - before - h0; }
            else if (before + h0 + h1 + h2 >= need) { digit = 253 - 4 * l; rem = need - before - h0 - h1; }
            else { digit = 252 - 4 * l; rem = need - before - h0 - h1 - h2; }
            sh[0] = digit;
            sh[1] = rem;
            sh[2] = hist[digit];        // size of the chosen bin (rem == size: the whole bin is taken, lower digits need no pass)
        }
    }
}

// Descending bitonic sort of N (power of two) 64-bit keys in LDS by a workgroup of NT threads.
template <int NT>
__device__ __forceinline__ void bitonic_sort_desc(unsigned long long* v, int N) {
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < N; i += NT) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = v[i], b = v[ixj];
                    const bool desc = ((i & k) == 0);
                    if (desc ? (a < b) : (a > b)) { v[i] = b; v[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// Descending sort of M unique 64-bit keys by counting ranks (no barriers inside): v -> tmp -> v. M <= capacity of both.
template <int NT>
__device__ __forceinline__ void rank_sort_desc(unsigned long long* v, unsigned long long* tmp, int M) {
    for (int e = threadIdx.x; e < M; e += NT) {
        const unsigned long long key = v[e];
        int rank = 0;
#pragma unroll 8
        for (int u = 0; u < M; ++u) rank += (v[u] > key) ? 1 : 0;
        tmp[rank] = key;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < M; e += NT) v[e] = tmp[e];
    __syncthreads();
}

// Exclusive, thread-ordered block scan of two flags. wtot = LDS scratch [2][NT/64]. Returns totals via references.
template <int NT>
__device__ __forceinline__ void block_scan2(bool f0, bool f1, unsigned* wtot, unsigned& ex0, unsigned& ex1,
                                            unsigned& tot0, unsigned& tot1) {
    constexpr int NWV = NT / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long b0 = __ballot(f0), b1 = __ballot(f1);
    const unsigned long long lower = (1ull << lane) - 1ull;
    const unsigned p0 = __popcll(b0 & lower), p1 = __popcll(b1 & lower);
    if (lane == 0) { wtot[wave] = __popcll(b0); wtot[NWV + wave] = __popcll(b1); }
    __syncthreads();
    unsigned o0 = 0, o1 = 0, t0 = 0, t1 = 0;
#pragma unroll
    for (int w = 0; w < NWV; ++w) {
        const unsigned c0 = wtot[w], c1 = wtot[NWV + w];
        if (w < wave) { o0 += c0; o1 += c1; }
        t0 += c0; t1 += c1;
    }
    ex0 = o0 + p0; ex1 = o1 + p1; tot0 = t0; tot1 = t1;
    __syncthreads();
}

// ---- NMS phases shared by the fast and the full per-class kernels -------------------------------------------------
template <int NW, int NWV = 4>
__device__ __forceinline__ void nms_mask_phase(const float4* __restrict__ cbox, const float* __restrict__ carea,
                                               unsigned long long* __restrict__ mask, int M, float nms_thr) {
    // 6. IoU mask: mask[i][w] bit j-64w set iff j > i and IoU(i, j) > nms_thr  (strict >, float32, inter/(a_i+a_j-inter)).
    //    Lane = column j (box in registers); two rows per iteration in packed fp32 (v_pk_*); the row boxes are LDS
    //    broadcasts; one __ballot per row yields the 64-bit word. The division is only executed when some lane is within
    //    2^-20 (relative) of the threshold: outside that band  inter > thr*union*(1+2^-20)  implies RN(inter/union) > thr
    //    and  inter < thr*union*(1-2^-20)  implies RN(inter/union) <= thr, so the result is identical to always dividing.
    const int nwords = (M + 63) >> 6;
    const int tid = threadIdx.x;
    {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const int lane = tid & 63, wave = tid >> 6;
        int item = 0;
        for (int w = 0; w < nwords; ++w) {
            const int j = 64 * w + lane;
            const float4 bj = cbox[j];          // j < MC always; rows >= M hold zero boxes
            const float aj = carea[j];
            const bool jvalid = j < M;
            const f2 jx1 = {bj.x, bj.x}, jy1 = {bj.y, bj.y}, jx2 = {bj.z, bj.z}, jy2 = {bj.w, bj.w}, ja = {aj, aj};
            const f2 zero = {0.f, 0.f};
            for (int rb = 0; rb <= w; ++rb, ++item) {
                if ((item % NWV) != wave) continue;
                const int iend = min(M, 64 * (rb + 1));
                const bool diag = (rb == w);
                // software-pipelined: the next row pair's boxes are read from LDS before this pair's math. A heavy class (M ~ topk)
                // runs this loop ~130 times per wave and each pass is a chain LDS read -> packed math -> ballot; with one wave per
                // SIMD there is nothing else to hide the read latency behind.
                float4 nb0 = cbox[64 * rb], nb1 = cbox[64 * rb + 1];
                float na0 = carea[64 * rb], na1 = carea[64 * rb + 1];
                for (int i = 64 * rb; i < iend; i += 2) {
                    const float4 b0 = nb0, b1 = nb1;      // i+1 <= MC-1; a row >= M is a zero box -> inter 0
                    const f2 ia = {na0, na1};
                    {
                        const int in = min(i + 2, 64 * rb + 62);     // stays inside this 64-row block (< MC)
                        nb0 = cbox[in]; nb1 = cbox[in + 1];
                        na0 = carea[in]; na1 = carea[in + 1];
                    }
                    const f2 ix1 = {b0.x, b1.x}, iy1 = {b0.y, b1.y}, ix2 = {b0.z, b1.z}, iy2 = {b0.w, b1.w};
                    const f2 xx1 = __builtin_elementwise_max(ix1, jx1), yy1 = __builtin_elementwise_max(iy1, jy1);
                    const f2 xx2 = __builtin_elementwise_min(ix2, jx2), yy2 = __builtin_elementwise_min(iy2, jy2);
                    const f2 iw = __builtin_elementwise_max(zero, xx2 - xx1), ih = __builtin_elementwise_max(zero, yy2 - yy1);
                    const f2 inter = iw * ih;
                    const f2 uni = (ia + ja) - inter;
                    const f2 t = uni * nms_thr;
                    const f2 hi = t * 1.00000095367431640625f, lo = t * 0.99999904632568359375f;
                    bool s0 = inter.x > hi.x, s1 = inter.y > hi.y;
                    const bool m0 = !s0 && !(inter.x < lo.x), m1 = !s1 && !(inter.y < lo.y);
                    if (m0 || m1) {                                   // borderline (incl. 0/0): decide exactly
                        if (m0) s0 = (inter.x / uni.x) > nms_thr;
                        if (m1) s1 = (inter.y / uni.y) > nms_thr;
                    }
                    s0 = s0 && jvalid;
                    s1 = s1 && jvalid;
                    if (diag) { s0 = s0 && (j > i); s1 = s1 && (j > i + 1); }
                    const unsigned long long w0 = __ballot(s0), w1 = __ballot(s1);
                    if (lane == 0) {
                        mask[i * NW + w] = w0;
                        if (i + 1 < iend) mask[(i + 1) * NW + w] = w1;
                    }
                }
            }
        }
    }
}

// greedy reduce by ONE wave (threadIdx.x < 64), 64 candidates at a time, then ordered compaction of the survivors
template <int NW>
__device__ __forceinline__ void nms_serial_phase(const unsigned long long* cand, const unsigned long long* mask,
                                                 unsigned long long* removed, int M, float* keptScoreOut, int* keptAnchorOut,
                                                 int* keptCountOut) {
    // 7. greedy reduce by wave 0, 64 candidates at a time; 8. compact kept candidates in order
    {
        const int lane = threadIdx.x;
        const int nwords = (M + 63) >> 6;
        int kept_before = 0;
        for (int c = 0; c < nwords; ++c) {
            const int i = 64 * c + lane;
            const unsigned long long rowbits = (i < M) ? mask[i * NW + c] : 0ull;
            const unsigned lo = (unsigned)rowbits, hi = (unsigned)(rowbits >> 32);
            const unsigned long long rem0 = removed[c];
            // wave-uniform state in SGPRs: the 64-step dependency chain runs on the scalar unit
            unsigned rem_lo = __builtin_amdgcn_readfirstlane((unsigned)rem0);
            unsigned rem_hi = __builtin_amdgcn_readfirstlane((unsigned)(rem0 >> 32));
            const int nvalid = min(64, M - 64 * c);
            // visit only the candidates that are still alive when their turn comes (find-first-set over the not-removed mask):
            // a suppressed candidate costs nothing, a kept one ORs its row into the removed set. Same order, same result as
            // testing l = 0..nvalid-1 one by one.
            const unsigned vmask_lo = nvalid >= 32 ? 0xFFFFFFFFu : ((1u << nvalid) - 1u);
            const unsigned vmask_hi = nvalid >= 64 ? 0xFFFFFFFFu : (nvalid > 32 ? ((1u << (nvalid - 32)) - 1u) : 0u);
            // Fixed-point form first: removed = external | OR of the rows of the not-removed candidates. The mask is strictly upper
            // triangular (a row only suppresses later candidates), so bit l of any fixed point is determined by the bits below l: the
            // fixed point is unique and IS the greedy result; iteration t fixes the candidates of dependency depth < t. With little
            // suppression inside a chunk (random boxes: depth 2 - 3) that is a few wave-wide ORs instead of 64 dependent
            // readlane steps (15 -> 4 us for a class with top-k candidates); deep chains fall through to the sequential walk.
            bool settled = false;
            {
                unsigned cur_lo = rem_lo, cur_hi = rem_hi;
#pragma unroll 1
                for (int it = 0; it < 6; ++it) {
                    const bool alive = lane < nvalid && !(((lane < 32 ? cur_lo : cur_hi) >> (lane & 31)) & 1u);
                    unsigned o_lo = alive ? lo : 0u, o_hi = alive ? hi : 0u;
#pragma unroll
                    for (int d = 32; d > 0; d >>= 1) {
                        o_lo |= __shfl_xor(o_lo, d);
                        o_hi |= __shfl_xor(o_hi, d);
                    }
                    const unsigned n_lo = __builtin_amdgcn_readfirstlane(o_lo) | rem_lo, n_hi = __builtin_amdgcn_readfirstlane(o_hi) | rem_hi;
                    if (n_lo == cur_lo && n_hi == cur_hi) { settled = true; break; }
                    cur_lo = n_lo;
                    cur_hi = n_hi;
                }
                if (settled) { rem_lo = cur_lo; rem_hi = cur_hi; }
            }
            unsigned done_lo = 0u, done_hi = 0u;          // candidates already visited
            while (!settled) {
                const unsigned cur_lo = ~(rem_lo | done_lo) & vmask_lo, cur_hi = ~(rem_hi | done_hi) & vmask_hi;
                if ((cur_lo | cur_hi) == 0u) break;
                const int l = cur_lo ? __builtin_ctz(cur_lo) : 32 + __builtin_ctz(cur_hi);
                rem_lo |= __builtin_amdgcn_readlane(lo, l);
                rem_hi |= __builtin_amdgcn_readlane(hi, l);
                if (l < 32) done_lo |= 1u << l; else done_hi |= 1u << (l - 32);
            }
            const unsigned long long rem = ((unsigned long long)rem_hi << 32) | rem_lo;
            const bool kept = (lane < nvalid) && !((rem >> lane) & 1ull);
            // propagate suppression by the kept candidates of this chunk to the later chunks
            if (kept) {
                for (int w = c + 1; w < nwords; ++w) {
                    const unsigned long long m = mask[i * NW + w];
                    if (m) atomicOr(&removed[w], m);
                }
            }
            const unsigned long long kb = __ballot(kept);
            if (kept) {
                const int pos = kept_before + __popcll(kb & ((1ull << lane) - 1ull));
                const unsigned long long kv = cand[i];
                keptScoreOut[pos] = __uint_as_float((unsigned)(kv >> 32));
                keptAnchorOut[pos] = (int)(0xFFFFFFFFu - (unsigned)(kv & 0xFFFFFFFFull));
            }
            kept_before += __popcll(kb);
            __threadfence_block();
        }
        if (lane == 0) *keptCountOut = kept_before;
    }
}

// ------------------------------------------------------------------------------------------------------------
// P2: per (image, class)
// ------------------------------------------------------------------------------------------------------------
template <int NW, bool PERM>   // 64-candidate words: candidate capacity MC = 64*NW >= topk; PERM: the column is stored anchor-major within a level
__device__ __forceinline__ void select_nms_one(const float* __restrict__ scoresT, const float4* __restrict__ boxes,
                                               int A, int Km1, float score_thr, float nms_thr, int topk,
                                               float* __restrict__ keptScore, int* __restrict__ keptAnchor,
                                               int* __restrict__ keptCount, long long* stamps, PostLevels lv, const int n, const int cls) {
    constexpr int MC = 64 * NW;
    constexpr int SORTN = (NW <= 1) ? 64 : (NW <= 2) ? 128 : (NW <= 4) ? 256 : 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // carve (all offsets multiples of 16 B)
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(smem);                 // [SORTN]
    unsigned long long* mask = cand + SORTN;                                                // [MC][NW]
    unsigned long long* removed = mask + MC * NW;                                           // [NW] (+pad to 8)
    float4* cbox = reinterpret_cast<float4*>(removed + 8);                                  // [MC]
    float* carea = reinterpret_cast<float*>(cbox + MC);                                     // [MC]
    unsigned* hist = reinterpret_cast<unsigned*>(carea + MC);                               // [256]
    unsigned* sh = hist + 256;                                                              // [16] scalars / wave totals
    unsigned* key = sh + 16;                                                                // [A]

    const int tid = threadIdx.x;
    const float* col = scoresT + ((size_t)n * Km1 + cls) * A;

    PP_STAMP(0);
    // 1. keys + count of passing scores
    if (tid < 16) sh[tid] = 0;
    __syncthreads();
    unsigned local = 0;
    for (int ap = tid; ap < A; ap += 256) {
        const float s = col[ap];
        const unsigned k = (s > score_thr) ? __float_as_uint(s) : 0u;     // strict > (generalized_ssd.py:371)
        key[PERM ? post_canon(lv, ap) : ap] = k;                                       // the column is stored anchor-major within a level: back to the canonical order
        local += (k != 0u);
    }
    {
        unsigned w = local;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) w += __shfl_xor(w, d);
        if ((tid & 63) == 0) atomicAdd(&sh[8], w);
    }
    __syncthreads();
    const unsigned cnt = sh[8];
    const size_t obase = ((size_t)n * Km1 + cls) * topk;
    if (cnt == 0) {
        if (tid == 0) keptCount[(size_t)n * Km1 + cls] = 0;
        return;
    }
    PP_STAMP(1);
    // 2. threshold key T: keep all keys > T and `quota` keys == T (lowest anchors first)
    unsigned T = 0, quota = 0;
    if (cnt > (unsigned)topk) {
        unsigned prefix = 0, need = topk;
#pragma unroll
        for (int shift = 24; shift >= 0; shift -= 8) {
            hist[tid] = 0;
            __syncthreads();
            for (int a = tid; a < A; a += 256) {
                const unsigned k = key[a];
                if (k != 0u && (shift == 24 || (k >> (shift + 8)) == (prefix >> (shift + 8)))) atomicAdd(&hist[(k >> shift) & 255u], 1u);
            }
            __syncthreads();
            radix_pick_digit(hist, need, sh);
            __syncthreads();
            prefix |= sh[0] << shift;
            need = sh[1];
            __syncthreads();
        }
        T = prefix;
        quota = need;
    }
    const int M = (int)min(cnt, (unsigned)topk);
    PP_STAMP(2);
    // 3. ordered compaction -> cand[] (keys > T first region, then == T in ascending anchor order)
    {
        unsigned base_gt = 0, base_eq = 0;
        const unsigned g_total = (cnt > (unsigned)topk) ? (unsigned)topk - quota : cnt;
        for (int a0 = 0; a0 < A; a0 += 256) {
            const int a = a0 + tid;
            const unsigned k = (a < A) ? key[a] : 0u;
            const bool gt = (k > T);
            const bool eq = (T != 0u) && (k == T);
            unsigned e0, e1, t0, t1;
            block_scan2<256>(gt, eq, sh, e0, e1, t0, t1);
            const unsigned long long kv = ((unsigned long long)k << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)a);
            if (gt) cand[base_gt + e0] = kv;
            if (eq && base_eq + e1 < quota) cand[g_total + base_eq + e1] = kv;
            base_gt += t0;
            base_eq += t1;
        }
    }
    for (int i = M + tid; i < SORTN; i += 256) cand[i] = 0ull;
    __syncthreads();
    PP_STAMP(3);
    // 4. sort by (score desc, anchor asc): keys are unique (anchor index in the low word) -> rank by counting
    rank_sort_desc<256>(cand, mask, M);
    PP_STAMP(4);
    // 5. gather boxes
    for (int i = tid; i < MC; i += 256) {
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < M) {
            const unsigned a = 0xFFFFFFFFu - (unsigned)(cand[i] & 0xFFFFFFFFull);
            b = boxes[(size_t)n * A + a];
        }
        cbox[i] = b;
        carea[i] = (b.z - b.x) * (b.w - b.y);
    }
    if (tid < 8) removed[tid] = 0ull;
    __syncthreads();
    PP_STAMP(5);
    nms_mask_phase<NW>(cbox, carea, mask, M, nms_thr);
    __syncthreads();
    PP_STAMP(6);
    if (tid < 64) nms_serial_phase<NW>(cand, mask, removed, M, keptScore + obase, keptAnchor + obase, keptCount + (size_t)n * Km1 + cls);
    PP_STAMP(7);
}


// What the merge needs besides the per-class survivor lists (one struct: the fallback launch carries it too)
struct MergeArgs {
    const float* scale_xy; float* oboxes; float* oscores; long long* olabels; int* ocounts; int* oanchor; float* opacked;
    const unsigned* tauKey; int* needFull; int D;
};
constexpr int MERGE_LCAP = 3072;
// LDS of the merge, carved from one buffer (the stand-alone launch owns a static one; the fused fallback launch lends the selection's dynamic LDS)
constexpr int MERGE_LDS = 2 * 512 * 8 + MERGE_LCAP * (8 + 4 + 2) + 256 * 4 + 256 + 256 * 4 + 1040;

template <int MT>
__device__ __forceinline__ void merge_body(char* __restrict__ mlds, const float* __restrict__ keptScore, const int* __restrict__ keptAnchor,
                                           const int* __restrict__ keptCount, const float4* __restrict__ boxes, int A, int Km1, int topk,
                                           const MergeArgs& g, int mode, int n);

template <int NW, bool PERM>
__global__ __launch_bounds__(256) void select_nms_kernel(const float* __restrict__ scoresT, const float4* __restrict__ boxes,
                                                        int A, int Km1, float score_thr, float nms_thr, int topk,
                                                        float* __restrict__ keptScore, int* __restrict__ keptAnchor,
                                                        int* __restrict__ keptCount, const int* __restrict__ needFull,
                                                        long long* stamps, int nimg, int xq, PostLevels lv, int islots, MergeArgs mg, int* __restrict__ fbcnt) {
    // islots == 0: flat grid [image slot][class], one (image, class) per workgroup; cls 0..Km1-1 (label = cls + 1).
    // islots > 0 (the fallback behind the cut-off pass, usually with no flagged image at all): grid [islots][class], a workgroup walks the images
    // slot, slot + islots, ... and works on the flagged ones -- 8 x 90 workgroups that look at 8 flags each instead of 64 x 90 that look at one.
    if (islots > 0) {
        const int slot = blockIdx.x / Km1, cls1 = blockIdx.x - slot * Km1;
        bool any = false;
        for (int n1 = slot; n1 < nimg; n1 += islots) any |= needFull[n1] != 0;
        if (!any) return;
        __shared__ int s_lastm;
        for (int n1 = slot; n1 < nimg; n1 += islots) {
            if (!needFull[n1]) continue;
            select_nms_one<NW, PERM>(scoresT, boxes, A, Km1, score_thr, nms_thr, topk, keptScore, keptAnchor, keptCount, stamps, lv, n1, cls1);
            __syncthreads();                    // the next image reuses the workgroup's LDS
            if (fbcnt) {
                // The image's merge in the same launch: the workgroup that finishes the image's LAST class merges it (one launch less per forward; the
                // path is rare, so the device-scope fences -- an L2 write-back / invalidate each on this chip -- are affordable here). The counter is
                // zeroed by tau_kernel in every forward and left at zero.
                __threadfence();
                __syncthreads();                // write, fence, BARRIER, ticket: every wave's release has completed before thread 0 publishes the arrival
                if (threadIdx.x == 0) s_lastm = atomicAdd(&fbcnt[n1], 1) == Km1 - 1;
                __syncthreads();
                if (s_lastm) {
                    __threadfence();
                    extern __shared__ __attribute__((aligned(16))) char dynlds[];      // the selection is done with it (launcher: >= MERGE_LDS bytes)
                    merge_body<256>(dynlds, keptScore, keptAnchor, keptCount, boxes, A, Km1, topk, mg, 1, n1);
                    if (threadIdx.x == 0) fbcnt[n1] = 0;
                }
                __syncthreads();
            }
        }
        return;
    }
    int n, cls;
    if (!xcd_image_of(blockIdx.x, Km1, xq, nimg, n, cls)) return;
    if (needFull && !needFull[n]) return;      // the fast path already produced this image's result
    select_nms_one<NW, PERM>(scoresT, boxes, A, Km1, score_thr, nms_thr, topk, keptScore, keptAnchor, keptCount, stamps, lv, n, cls);
}

// the cut-off as a launch of its own (one workgroup per image)
__global__ __launch_bounds__(256) void tau_kernel(TauArgs t) {
    __shared__ unsigned part[256];
    int n, unused;
    if (!xcd_image_of(blockIdx.x, 1, t.xq, t.nimg, n, unused)) return;
    tau_body<false>(t, n, part);
}

// Fast path of P2: same semantics as select_nms_kernel restricted to keys >= tau. A class with more than topk scores >= tau keeps
// its topk highest (the per-class cap of generalized_ssd.py:376: every score >= tau outranks every score below it, so the top-k
// of the class is the top-k of this set), as long as the set fits the CAP-entry list; beyond that needFull[n] hands the image to
// the full kernel. (With 24 732 anchors -- ssd512 -- one dominant class regularly has more than topk = 400 scores above the
// cut-off: without the in-kernel cap every image of that model went through the full path, 3.0 of its 11.1 ms per batch.)
template <int NW, int FT, bool PERM>
__global__ __launch_bounds__(FT) void select_nms_fast_kernel(const float* __restrict__ scoresT, const float4* __restrict__ boxes,
                                                             int A, int Km1, float score_thr, float nms_thr, int topk,
                                                             const unsigned* __restrict__ tauKey, int* __restrict__ needFull,
                                                             float* __restrict__ keptScore, int* __restrict__ keptAnchor,
                                                             int* __restrict__ keptCount, long long* __restrict__ stamps, int nimg, int xq, PostLevels lv,
                                                             const int* __restrict__ order) {
    constexpr int MC = 64 * NW;
    constexpr int CAP = (64 * NW * NW < 2048) ? 64 * NW * NW : 2048;      // candidate list (>= MC); the sort scratch aliases the IoU mask
    static_assert(CAP >= MC && CAP <= MC * NW, "sort scratch must fit the mask array");
    __shared__ unsigned long long cand[CAP];
    __shared__ unsigned long long mask[MC * NW];
    unsigned long long* tmp = mask;             // only used by the rank sort, before the mask phase writes the array
    __shared__ unsigned long long removed[8];
    __shared__ __attribute__((aligned(16))) float4 cbox[MC];
    __shared__ float carea[MC];
    __shared__ unsigned cnt_sh;
    const int tid = threadIdx.x;
    int n, cls;
    if (order) {
        // flat grid [rank][image slot]: every image's heaviest class first (tau_kernel's order), the light ones fill in behind
        const int flat = blockIdx.x;
        int rank;
        if (xq > 0) {
            const int w = flat >> 3;
            rank = w / xq;
            n = (flat & 7) * xq + (w - rank * xq);
        } else {
            rank = flat / nimg;
            n = flat - rank * nimg;
        }
        if (n >= nimg) return;
        cls = order[(size_t)n * Km1 + rank];
    } else if (!xcd_image_of(blockIdx.x, Km1, xq, nimg, n, cls)) return;      // flat grid [image slot][class]
    const float* col = scoresT + ((size_t)n * Km1 + cls) * A;
    const unsigned tau = tauKey[n];
    PP_STAMP(0);
    if (tid == 0) cnt_sh = 0;
    if (tid < 8) removed[tid] = 0ull;
    __syncthreads();
    // the scan is pure latency (13 KB per workgroup for A = 3234): keep 8 independent loads in flight per thread instead of
    // one dependent round trip per FT anchors
    for (int a0 = 0; a0 < A; a0 += FT * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int a = a0 + u * FT + tid;
            v[u] = (a < A) ? col[a] : -1.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned k = __float_as_uint(v[u]);
            if (v[u] > score_thr && k >= tau) {          // out-of-range lanes are re-checked below (a < A)
                const int ap = a0 + u * FT + tid;
                if (ap < A) {
                    const unsigned pos = atomicAdd(&cnt_sh, 1u);
                    const int a = PERM ? post_canon(lv, ap) : ap;       // stored -> canonical anchor index (tie-break, box lookup, output)
                    if (pos < (unsigned)CAP) cand[pos] = ((unsigned long long)k << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)a);
                }
            }
        }
    }
    __syncthreads();
    PP_STAMP(1);
    const unsigned cnt = cnt_sh;
    const size_t obase = ((size_t)n * Km1 + cls) * topk;
    if (cnt > (unsigned)CAP) {                  // does not fit the list -> whole image goes through the full path
        if (tid == 0) { needFull[n] = 1; keptCount[(size_t)n * Km1 + cls] = 0; }
        return;
    }
    if (cnt == 0) {
        if (tid == 0) keptCount[(size_t)n * Km1 + cls] = 0;
        PP_STAMP(2); PP_STAMP(3); PP_STAMP(4); PP_STAMP(5);
        return;
    }
    rank_sort_desc<FT>(cand, tmp, (int)cnt);   // unique keys: order is the canonical (score desc, anchor asc)
    const int M = min((int)cnt, topk);          // per-class cap: the first topk of the sorted list
    PP_STAMP(2);
    for (int i = tid; i < MC; i += FT) {
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < M) b = boxes[(size_t)n * A + (0xFFFFFFFFu - (unsigned)(cand[i] & 0xFFFFFFFFull))];
        cbox[i] = b;
        carea[i] = (b.z - b.x) * (b.w - b.y);
    }
    __syncthreads();
    PP_STAMP(3);
    nms_mask_phase<NW, FT / 64>(cbox, carea, mask, M, nms_thr);
    __syncthreads();
    PP_STAMP(4);
    if (tid < 64) nms_serial_phase<NW>(cand, mask, removed, M, keptScore + obase, keptAnchor + obase, keptCount + (size_t)n * Km1 + cls);
    PP_STAMP(5);
}

// ------------------------------------------------------------------------------------------------------------
// P3: per image
// ------------------------------------------------------------------------------------------------------------
template <int MT>       // threads per workgroup: 1024 sorts fastest, 256 is scheduled at once beside the other chain's kernels (a 1024-thread
                        // workgroup waits for 16 free wave slots on ONE compute unit)
__device__ __forceinline__ void merge_body(char* __restrict__ mlds, const float* __restrict__ keptScore, const int* __restrict__ keptAnchor,
                                           const int* __restrict__ keptCount, const float4* __restrict__ boxes, int A, int Km1, int topk,
                                           const MergeArgs& g, int mode, int n) {
    // mode 0: after the fast per-class pass (may raise needFull); mode 1: after the full pass (only flagged images);
    // mode 2: unconditional (fast path disabled)
    const float* __restrict__ scale_xy = g.scale_xy;
    float* __restrict__ oboxes = g.oboxes; float* __restrict__ oscores = g.oscores; long long* __restrict__ olabels = g.olabels;
    int* __restrict__ ocounts = g.ocounts; int* __restrict__ oanchor = g.oanchor; float* __restrict__ opacked = g.opacked;
    const unsigned* __restrict__ tauKey = g.tauKey; int* __restrict__ needFull = g.needFull;
    const int D = g.D;
    unsigned long long* fin = reinterpret_cast<unsigned long long*>(mlds);           // [512]
    unsigned long long* fin2 = fin + 512;                                            // [512]
    unsigned long long* lkey = fin2 + 512;                                           // [MERGE_LCAP]
    int* lanc = reinterpret_cast<int*>(lkey + MERGE_LCAP);                           // [MERGE_LCAP]
    unsigned short* llab = reinterpret_cast<unsigned short*>(lanc + MERGE_LCAP);     // [MERGE_LCAP]
    unsigned* hist = reinterpret_cast<unsigned*>(llab + MERGE_LCAP);                 // [256]
    unsigned* sh = hist + 256;                                                       // [40] (64 reserved)
    int* ccount = reinterpret_cast<int*>(sh + 64);                                   // [256]
    int* cpre = ccount + 256;                                                        // [257]
    const int tid = threadIdx.x;
    const int F = Km1 * topk;
    const float* ks = keptScore + (size_t)n * F;
    const int* ka = keptAnchor + (size_t)n * F;
    if (tid < 40) sh[tid] = 0;
    for (int c = tid; c < Km1; c += MT) ccount[c] = keptCount[(size_t)n * Km1 + c];
    __syncthreads();
    if (tid < 64) {
        unsigned t = 0;
        for (int c = tid; c < Km1; c += 64) t += ccount[c];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) t += __shfl_xor(t, d);
        if (tid == 0) sh[8] = t;
    }
    __syncthreads();
    const unsigned total = sh[8];
    if (mode == 0 && total < (unsigned)D && tauKey[n] != 0u) {
        // fewer survivors than outputs while candidates below tau were skipped: redo this image with the full path
        if (tid == 0) needFull[n] = 1;
        return;
    }
    // survivors live in per-class slots [c][0..ccount[c]); walk them through the class prefix sums (e -> class by binary
    // search) instead of scanning all Km1*topk slots: `total` is a few thousand, the slot array 27k.
    if (tid < 64) {
        // exclusive scan of ccount over classes, one wave (Km1 <= 256)
        int run = 0;
        for (int c0 = 0; c0 < Km1; c0 += 64) {
            const int c = c0 + tid;
            const int v = (c < Km1) ? ccount[c] : 0;
            int inc = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(inc, d);
                if (tid >= d) inc += o;
            }
            if (c < Km1) cpre[c] = run + inc - v;
            run += __shfl(inc, 63);
        }
        if (tid == 0) cpre[Km1] = run;
    }
    __syncthreads();
    auto slot_of = [&](int e) {
        int lo = 0, hi = Km1;               // largest c with cpre[c] <= e
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (cpre[mid] <= e) lo = mid; else hi = mid;
        }
        return lo * topk + (e - cpre[lo]);
    };
    // The survivors' (score, slot) keys are fetched ONCE into LDS when they fit (a few thousand at most with the cut-off path): the
    // four radix passes and the compaction then run out of LDS instead of repeating a binary search and a dependent global load
    // per entry and pass (the kernel was five exposed memory round trips long: 20 us per launch at any batch size).
    // (round 3: the survivor's anchor and label are fetched in the same round trip and kept beside the key, whose low word is then the survivor's
    //  index e -- monotone in the slot f, so ties break exactly as before -- : the output gather is one dependent load shorter)
    constexpr int LCAP = MERGE_LCAP;
    const bool in_lds = total <= (unsigned)LCAP;
    if (in_lds) {
        for (int e = tid; e < (int)total; e += MT) {
            const unsigned f = (unsigned)slot_of(e);
            lkey[e] = ((unsigned long long)__float_as_uint(ks[f]) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)e);
            lanc[e] = ka[f];
            llab[e] = (unsigned short)(f / (unsigned)topk + 1u);
        }
        __syncthreads();
    }
    unsigned T = 0, quota = 0;
    if (total > (unsigned)D) {
        unsigned prefix = 0, need = D;
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            for (int e = tid; e < (int)total; e += MT) {
                const unsigned k = in_lds ? (unsigned)(lkey[e] >> 32) : __float_as_uint(ks[slot_of(e)]);
                if (shift == 24 || (k >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(k >> shift) & 255u], 1u);
            }
            __syncthreads();
            radix_pick_digit(hist, need, sh);
            __syncthreads();
            prefix |= sh[0] << shift;
            need = sh[1];
            const bool whole = sh[1] == sh[2];
            __syncthreads();
            if (whole && shift > 0) {        // the whole bin is taken: every key >= prefix (lower digits zero) is in, nothing is left to split
                prefix -= 1u;
                need = 0;
                break;
            }
        }
        T = prefix;
        quota = need;
    }
    const int M = (int)min(total, (unsigned)D);
    {
        unsigned base_gt = 0, base_eq = 0;
        const unsigned g_total = (total > (unsigned)D) ? (unsigned)D - quota : total;
        for (int e0i = 0; e0i < (int)total; e0i += MT) {
            const int e = e0i + tid;
            unsigned k = 0, f = 0;
            if (e < (int)total) {
                if (in_lds) {
                    const unsigned long long kv0 = lkey[e];
                    k = (unsigned)(kv0 >> 32);
                    f = (unsigned)e;            // (the key's low word: the survivor index)
                } else {
                    f = (unsigned)slot_of(e);
                    k = __float_as_uint(ks[f]);
                }
            }
            const bool gt = (k > T);
            const bool eq = (T != 0u) && (k == T);
            unsigned e0, e1, t0, t1;
            block_scan2<MT>(gt, eq, sh, e0, e1, t0, t1);
            const unsigned long long kv = ((unsigned long long)k << 32) | (unsigned long long)(0xFFFFFFFFu - f);
            if (gt) fin[base_gt + e0] = kv;
            if (eq && base_eq + e1 < quota) fin[g_total + base_eq + e1] = kv;
            base_gt += t0;
            base_eq += t1;
        }
    }
    __syncthreads();
    rank_sort_desc<MT>(fin, fin2, M);
    float sx = 1.f, sy = 1.f;
    if (scale_xy) { sx = scale_xy[2 * n]; sy = scale_xy[2 * n + 1]; }
    for (int i = tid; i < D; i += MT) {
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        float s = 0.f;
        long long lab = 0;
        int anc = -1;
        if (i < M) {
            const unsigned long long kv = fin[i];
            const unsigned f = 0xFFFFFFFFu - (unsigned)(kv & 0xFFFFFFFFull);
            s = __uint_as_float((unsigned)(kv >> 32));
            if (in_lds) { lab = (long long)llab[f]; anc = lanc[f]; }
            else { lab = (long long)(f / (unsigned)topk) + 1; anc = ka[f]; }
            b = boxes[(size_t)n * A + anc];
            b.x *= sx; b.z *= sx; b.y *= sy; b.w *= sy;        // resize_boxes (transform.py:286-291)
        }
        reinterpret_cast<float4*>(oboxes)[(size_t)n * D + i] = b;
        oscores[(size_t)n * D + i] = s;
        olabels[(size_t)n * D + i] = lab;
        if (oanchor) oanchor[(size_t)n * D + i] = anc;
        if (opacked) {
            float* pr = opacked + ((size_t)n * (D + 1) + i) * 6;
            pr[0] = b.x; pr[1] = b.y; pr[2] = b.z; pr[3] = b.w; pr[4] = s; pr[5] = (float)lab;
        }
    }
    if (tid == 0) {
        ocounts[n] = M;
        if (opacked) {
            float* pr = opacked + ((size_t)n * (D + 1) + D) * 6;
            pr[0] = (float)M; pr[1] = pr[2] = pr[3] = pr[4] = pr[5] = 0.f;
        }
    }
}

template <int MT>
__global__ __launch_bounds__(MT) void merge_kernel(const float* __restrict__ keptScore, const int* __restrict__ keptAnchor,
                                                    const int* __restrict__ keptCount, const float4* __restrict__ boxes, int A, int Km1, int topk,
                                                    MergeArgs g, int mode, int nimg, int xq) {
    __shared__ __attribute__((aligned(16))) char mlds[MERGE_LDS];
    int n, unused;
    if (!xcd_image_of(blockIdx.x, 1, xq, nimg, n, unused)) return;
    if (mode == 0 && g.needFull[n]) return;
    if (mode == 1 && !g.needFull[n]) return;
    merge_body<MT>(mlds, keptScore, keptAnchor, keptCount, boxes, A, Km1, topk, g, mode, n);
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

template <int NW>
size_t p2_lds_bytes(int A) {
    constexpr int MC = 64 * NW;
    constexpr int SORTN = (NW <= 1) ? 64 : (NW <= 2) ? 128 : (NW <= 4) ? 256 : 512;
    return (size_t)SORTN * 8 + (size_t)MC * NW * 8 + 64 + (size_t)MC * 16 + (size_t)MC * 4 + 1024 + 64 + (size_t)A * 4;
}

template <int NW>
int launch_p2(const PostArgs& a, const float* scoresT, const float4* boxes, float* keptScore, int* keptAnchor, int* keptCount,
              const int* needFull, const MergeArgs& mg, int* fbcnt, hipStream_t s) {
    // (fbcnt: the image's merge runs in this launch, in the selection's LDS: the request covers both)
    const size_t lds = fbcnt ? std::max<size_t>(p2_lds_bytes<NW>(a.A), MERGE_LDS) : p2_lds_bytes<NW>(a.A);
    if (lds > 160 * 1024 - 64) {
        dn_set_error("postprocess: %d anchors need %zu B of LDS (> 160 KiB)", a.A, lds);
        return DN_E_UNSUPPORTED;
    }
    const bool perm = !(a.lv.n == 1 && a.lv.aloc[0] == 1);
    // behind the cut-off pass (needFull given) only flagged images are worked on -- usually none: a small grid whose workgroups walk the images
    const int islots = needFull ? std::min(a.n, 8) : 0;
    if (perm) {
        DN_HIP_CHECK(dn_allow_big_lds(reinterpret_cast<const void*>(select_nms_kernel<NW, true>), 160 * 1024 - 64));      // (minus the kernel's static LDS: the ticket flag)
        hipLaunchKernelGGL((select_nms_kernel<NW, true>), dim3((a.K - 1) * (islots > 0 ? islots : xcd_image_slots(a.xq, a.n))), dim3(256), lds, s, scoresT, boxes, a.A, a.K - 1,
                           a.score_thresh, a.nms_thresh, a.topk, keptScore, keptAnchor, keptCount, needFull, g_pp_stamps, a.n, a.xq, a.lv, islots, mg, islots > 0 ? fbcnt : nullptr);
        return DN_OK;
    }
    DN_HIP_CHECK(dn_allow_big_lds(reinterpret_cast<const void*>(select_nms_kernel<NW, false>), 160 * 1024 - 64));      // (minus the kernel's static LDS: the ticket flag)
    hipLaunchKernelGGL((select_nms_kernel<NW, false>), dim3((a.K - 1) * (islots > 0 ? islots : xcd_image_slots(a.xq, a.n))), dim3(256), lds, s, scoresT, boxes, a.A, a.K - 1,
                       a.score_thresh, a.nms_thresh, a.topk, keptScore, keptAnchor, keptCount, needFull, g_pp_stamps, a.n, a.xq, a.lv, islots, mg, islots > 0 ? fbcnt : nullptr);
    return DN_OK;
}

template <int NW>
int launch_p2_fast(const PostArgs& a, const float* scoresT, const float4* boxes, const unsigned* tauKey, int* needFull,
                   float* keptScore, int* keptAnchor, int* keptCount, const int* order, hipStream_t s) {
    // 512 threads: the column scan is one batch of loads for A <= 4096, and a heavy class (up to topk candidates) spreads its
    // IoU-mask rows over 8 waves instead of 4 -- the kernel's duration is the lifetime of its heaviest workgroups
    const bool perm = !(a.lv.n == 1 && a.lv.aloc[0] == 1);
    const dim3 grid((a.K - 1) * xcd_image_slots(a.xq, a.n));
#define DN_P2F(FT, PERM) hipLaunchKernelGGL((select_nms_fast_kernel<NW, FT, PERM>), grid, dim3(FT), 0, s, scoresT, boxes, a.A, a.K - 1, a.score_thresh, \
                                            a.nms_thresh, a.topk, tauKey, needFull, keptScore, keptAnchor, keptCount, g_pp_stamps, a.n, a.xq, a.lv, order)
    if (perm) DN_P2F(512, true); else DN_P2F(512, false);
#undef DN_P2F
    return DN_OK;
}

}  // namespace

PostBuffers post_buffers(void* ws, int n, int A, int K, int topk) {
    const size_t Km1 = K - 1;
    PostBuffers b;
    unsigned char* p = reinterpret_cast<unsigned char*>(ws);
    b.scoresT = reinterpret_cast<float*>(p);
    p += align256((size_t)n * Km1 * A * 4);
    b.boxes = reinterpret_cast<float4*>(p);
    p += align256((size_t)n * A * 16);
    b.keptScore = reinterpret_cast<float*>(p);
    p += align256((size_t)n * Km1 * topk * 4);
    b.keptAnchor = reinterpret_cast<int*>(p);
    p += align256((size_t)n * Km1 * topk * 4);
    b.keptCount = reinterpret_cast<int*>(p);
    p += align256((size_t)n * Km1 * 4);
    b.tiles = dn_cdiv(A, 64);
    b.phist = reinterpret_cast<unsigned*>(p);                              // [n][tiles][HBINS], then tauKey[n], needFull[n]
    b.tauKey = b.phist + (size_t)n * b.tiles * HBINS;
    b.needFull = reinterpret_cast<int*>(b.tauKey + n);
    b.order = b.needFull + n;                                              // [n][K-1] classes, heaviest first (tau_kernel)
    b.fbcnt = b.order + (size_t)n * Km1;                                   // [n] tickets of the fused fallback merge (zeroed by tau_kernel)
    return b;
}

size_t postprocess_ws_bytes(int n, int A, int K, int topk, int dets) {
    (void)dets;
    const size_t Km1 = K - 1;
    return align256((size_t)n * Km1 * A * 4) + align256((size_t)n * A * 16) + 2 * align256((size_t)n * Km1 * topk * 4) +
           align256((size_t)n * Km1 * 4) + align256((size_t)n * dn_cdiv(A, 64) * HBINS * 4 + (size_t)n * (12 + 4 * Km1));
}

void post_hist_range(float score_thresh, int* hb0_out, int* nb_out, int* clamped_out) {
    // histogram bins that scores in (score_thresh, 1] can reach (float bits >> HSHIFT is monotone for positive floats)
    unsigned thr_bits, one_bits;
    const float t = score_thresh > 0.f ? score_thresh : 0.f, one = 1.0f;
    memcpy(&thr_bits, &t, 4);
    memcpy(&one_bits, &one, 4);
    const int top = (int)(one_bits >> HSHIFT);
    const int hb_thr = (int)(thr_bits >> HSHIFT);
    const int hb0 = hb_thr > top + 1 - HBINS ? hb_thr : top + 1 - HBINS;
    *hb0_out = hb0;
    *nb_out = top + 1 - hb0;
    *clamped_out = hb_thr < hb0;
}

// DN_PP_FAST=0 disables the cut-off fast path (A/B and tests of the full path); DN_PP_WANT overrides the multiple of D.
static int pp_env(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

int launch_postprocess(const PostArgs& a0, hipStream_t s, hipEvent_t* ev) {
    PostArgs a = a0;
    DN_REQUIRE(a.n > 0 && a.A > 0 && a.K >= 2, "postprocess: bad sizes n=%d A=%d K=%d", a.n, a.A, a.K);
    DN_REQUIRE(a.topk >= 1 && a.topk <= 512, "postprocess: topk_candidates=%d outside [1,512]", a.topk);
    DN_REQUIRE(a.dets >= 1 && a.dets <= 512, "postprocess: detections_per_img=%d outside [1,512]", a.dets);
    DN_REQUIRE(a.K - 1 <= 256, "postprocess: more than 256 foreground classes");
    const size_t need = postprocess_ws_bytes(a.n, a.A, a.K, a.topk, a.dets);
    if (a.ws_bytes < need) {
        dn_set_error("postprocess: workspace %zu B < required %zu B", a.ws_bytes, need);
        return DN_E_WORKSPACE;
    }
    if (a.lv.n == 1 && a.lv.off[1] == 0) { a.lv.off[1] = a.A; a.lv.hw[0] = a.A; a.lv.aloc[0] = 1; }      // no level table given: one level, stored == canonical
    {
        bool ok = a.lv.n >= 1 && a.lv.n <= 8 && a.lv.off[0] == 0 && a.lv.off[a.lv.n] == a.A;
        for (int l = 0; ok && l < a.lv.n; ++l) ok = a.lv.hw[l] >= 1 && a.lv.aloc[l] >= 1 && a.lv.off[l + 1] - a.lv.off[l] == a.lv.hw[l] * a.lv.aloc[l];
        DN_REQUIRE(ok, "postprocess: the level table does not cover the %d anchors", a.A);
    }
    const size_t Km1 = a.K - 1;
    const PostBuffers pb = post_buffers(a.ws, a.n, a.A, a.K, a.topk);
    float* const scoresT = pb.scoresT; float4* const boxes = pb.boxes;
    float* const keptScore = pb.keptScore; int* const keptAnchor = pb.keptAnchor; int* const keptCount = pb.keptCount;
    const int tiles = pb.tiles;
    unsigned* const phist = pb.phist; unsigned* const tauKey = pb.tauKey;
    int* const needFull = pb.needFull; int* const order = pb.order; int* const fbcnt = pb.fbcnt;
    HistRows hrows = a.scores_ready ? a.hrows : HistRows{};
    const bool small = a.scores_ready && a.small_first >= 0 && a.small_first < a.A;      // the levels the head launch left in logit form
    const int small_tiles = small ? dn_cdiv(a.A - a.small_first, 64) : 0;
    if (a.scores_ready) {
        DN_REQUIRE(hrows.levels >= 1 && hrows.levels <= 8 && a.lv.n == 1 && a.lv.aloc[0] == 1, "postprocess: scores from the head launch need a histogram-row table");
        hrows.extra_base = hrows.rows_per_image;
        hrows.extra_rows = small_tiles;
        hrows.rows_per_image += small_tiles;
        DN_REQUIRE(hrows.rows_per_image <= tiles, "postprocess: %d histogram rows per image, %d available", hrows.rows_per_image, tiles);
        DN_REQUIRE(!small || (a.logits && a.reg), "postprocess: the small levels arrive as logits: null logits / regressions");
    }

    const int fast = dn_knob("DN_PP_FAST", 1);
    // candidates per image kept by the cut-off, as a multiple of D. Any value is exact (too few survivors -> device-side
    // fallback to the full kernel for that image); 4 leaves a 4x margin for NMS suppression and keeps the heaviest per-class
    // workgroups (the kernel's tail) short: 8 -> 4 is -2.5 % on the step, 2 another -2 % but with no margin.
    const int want_mult = dn_knob("DN_PP_WANT", 4);
    long long* labels = reinterpret_cast<long long*>(a.labels);
    const int nw = (a.topk + 63) / 64;
    const MergeArgs mg{a.scale_xy, a.boxes, a.scores, labels, a.counts, a.kept_anchor, a.packed, tauKey, needFull, a.dets};

    if (ev) (void)hipEventRecord(ev[0], s);
    int hb0, nb, clamped;
    post_hist_range(a.score_thresh, &hb0, &nb, &clamped);
    const size_t lds1 = (size_t)(64 * a.K + 64) * sizeof(float) + (size_t)HBINS * sizeof(unsigned);
    const int slots = xcd_image_slots(a.xq, a.n);
    const TauArgs targs{phist, tiles, hb0, clamped, (unsigned)(want_mult * a.dets), tauKey, needFull, a.n, a.xq, nb, (int)(a.K - 1), order, fbcnt, hrows};
    // the cut-off inside the softmax launch of the small levels (its last tile per image): when that launch exists and the caller gave counters
    const bool fold_tau = fast && a.scores_ready && small_tiles > 0 && a.tickets != nullptr && (a.lv.n == 1 && a.lv.aloc[0] == 1) && dn_knob("DN_PP_FOLD_TAU", 1) != 0;
    {
        // every anchor (plain path), or the anchors [small_first, A) the fused head launch left in logit form: a few tiles per image, their histogram
        // rows behind the head launch's rows
        const long long* const stp = nullptr;       // (dev builds stamp the selection kernels; this launch is not stamped)
        const int t1 = a.scores_ready ? small_tiles : tiles, abase = a.scores_ready ? a.small_first : 0;
        const int rstride = a.scores_ready ? hrows.rows_per_image : tiles, rbase = a.scores_ready ? hrows.extra_base : 0;
        if (t1 > 0) {
            if (!(a.lv.n == 1 && a.lv.aloc[0] == 1))
                hipLaunchKernelGGL(softmax_decode_kernel<true>, dim3(t1 * slots), dim3(256), lds1, s, a.logits, a.reg, a.anchors,
                                   scoresT, boxes, a.A, a.K, a.img_w, a.img_h, a.score_thresh, phist, hb0, nb, const_cast<long long*>(stp),
                                   a.n, t1, a.xq, a.lv, abase, rstride, rbase, (unsigned*)nullptr, TauArgs{});
            else
                hipLaunchKernelGGL(softmax_decode_kernel<false>, dim3(t1 * slots), dim3(256), lds1, s, a.logits, a.reg, a.anchors,
                                   scoresT, boxes, a.A, a.K, a.img_w, a.img_h, a.score_thresh, phist, hb0, nb, const_cast<long long*>(stp),
                                   a.n, t1, a.xq, a.lv, abase, rstride, rbase, fold_tau ? a.tickets : (unsigned*)nullptr, targs);
        }
    }
    if (ev) (void)hipEventRecord(ev[1], s);
    int rc = DN_OK;
    if (fast && !fold_tau) hipLaunchKernelGGL(tau_kernel, dim3(slots), dim3(256), 0, s, targs);
    if (fast) {
        const int* ord = dn_knob("DN_PP_ORDER", 1) ? order : nullptr;      // heaviest classes first (0: class order)
        if (nw <= 1) rc = launch_p2_fast<1>(a, scoresT, boxes, tauKey, needFull, keptScore, keptAnchor, keptCount, ord, s);
        else if (nw <= 2) rc = launch_p2_fast<2>(a, scoresT, boxes, tauKey, needFull, keptScore, keptAnchor, keptCount, ord, s);
        else if (nw <= 4) rc = launch_p2_fast<4>(a, scoresT, boxes, tauKey, needFull, keptScore, keptAnchor, keptCount, ord, s);
        else if (nw <= 5) rc = launch_p2_fast<5>(a, scoresT, boxes, tauKey, needFull, keptScore, keptAnchor, keptCount, ord, s);
        else rc = launch_p2_fast<8>(a, scoresT, boxes, tauKey, needFull, keptScore, keptAnchor, keptCount, ord, s);
        if (ev) (void)hipEventRecord(ev[2], s);
        hipLaunchKernelGGL(merge_kernel<1024>, dim3(slots), dim3(1024), 0, s, keptScore, keptAnchor, keptCount, boxes, a.A, (int)Km1, a.topk, mg, 0, a.n, a.xq);
    }
    else if (ev) (void)hipEventRecord(ev[2], s);
    if (ev) (void)hipEventRecord(ev[3], s);
    const int* flag = fast ? needFull : nullptr;
    // behind the cut-off pass: ONE launch redoes the flagged images (usually none) with the full kernel and merges each in the workgroup that
    // finishes its last class (DN_PP_FUSE_FALLBACK, default 1; 0: a second merge launch as in rounds 1 - 3)
    const bool fuse_fb = fast && dn_knob("DN_PP_FUSE_FALLBACK", 1) != 0;
    int* fb = fuse_fb ? fbcnt : nullptr;
    if (nw <= 1) rc = launch_p2<1>(a, scoresT, boxes, keptScore, keptAnchor, keptCount, flag, mg, fb, s);
    else if (nw <= 2) rc = launch_p2<2>(a, scoresT, boxes, keptScore, keptAnchor, keptCount, flag, mg, fb, s);
    else if (nw <= 4) rc = launch_p2<4>(a, scoresT, boxes, keptScore, keptAnchor, keptCount, flag, mg, fb, s);
    else if (nw <= 5) rc = launch_p2<5>(a, scoresT, boxes, keptScore, keptAnchor, keptCount, flag, mg, fb, s);
    else rc = launch_p2<8>(a, scoresT, boxes, keptScore, keptAnchor, keptCount, flag, mg, fb, s);
    if (rc != DN_OK) return rc;
    if (!fuse_fb) {
        // the merge after the full pass: with the fast path on it only works for flagged images (usually none): 256 threads, scheduled at once
        if (fast)
            hipLaunchKernelGGL(merge_kernel<256>, dim3(slots), dim3(256), 0, s, keptScore, keptAnchor, keptCount, boxes, a.A, (int)Km1, a.topk, mg, 1, a.n, a.xq);
        else
            hipLaunchKernelGGL(merge_kernel<1024>, dim3(slots), dim3(1024), 0, s, keptScore, keptAnchor, keptCount, boxes, a.A, (int)Km1, a.topk, mg, fast ? 1 : 2, a.n, a.xq);
    }
    if (ev) (void)hipEventRecord(ev[4], s);
    DN_HIP_CHECK(hipGetLastError());
    return DN_OK;
}
