// Arithmetic of the first post-process step -- softmax over the classes (generalized_ssd.py:354), BoxCoder.decode_single + clip
// (_utils.py:187-224, generalized_ssd.py:362-363), the score-histogram bin -- shared by softmax_decode_kernel (postprocess.hip) and the
// epilogue of head_fused_kernel (headfuse.hip). Both must produce the same bits from the same logits, whatever the contraction setting of
// the file that includes this (fp32, the reference's op order, no FMA fusion: every function switches contraction off for its body).
#pragma once
#include "common.h"

constexpr float PP_BBOX_XFORM_CLIP = 4.135166556742356f;   // log(1000/16), _utils.py:135

__device__ __forceinline__ float4 pp_decode_box(const float4 rg, const float4 an, const float img_w, const float img_h) {
#pragma clang fp contract(off)
    const float w = an.z - an.x, h = an.w - an.y;
    const float cx = an.x + 0.5f * w, cy = an.y + 0.5f * h;
    const float dx = rg.x / 10.f, dy = rg.y / 10.f;
    const float dw = fminf(rg.z / 5.f, PP_BBOX_XFORM_CLIP), dh = fminf(rg.w / 5.f, PP_BBOX_XFORM_CLIP);
    const float pcx = dx * w + cx, pcy = dy * h + cy;
    const float pw = expf(dw) * w, ph = expf(dh) * h;
    float4 b;
    b.x = pcx - 0.5f * pw;
    b.y = pcy - 0.5f * ph;
    b.z = pcx + 0.5f * pw;
    b.w = pcy + 0.5f * ph;
    b.x = fminf(fmaxf(b.x, 0.f), img_w);
    b.z = fminf(fmaxf(b.z, 0.f), img_w);
    b.y = fminf(fmaxf(b.y, 0.f), img_h);
    b.w = fminf(fmaxf(b.w, 0.f), img_h);
    return b;
}

// exp(d) for d <= 0 (softmax arguments x - max): the reduction of the device library's expf, instruction for instruction -- product with log2(e)
// split into a rounded head and an fma tail, exp2 of the fraction, ldexp by the integer part, zero below the float range -- without its overflow
// branch (d <= 0 cannot overflow) and therefore with the same bits as expf(d).
__device__ __forceinline__ float pp_exp_nonpos(const float d) {
#pragma clang fp contract(off)
    const float t = d * 1.44269502162933349609375f;                                // 0x3fb8aa3b
    float lo = __builtin_fmaf(d, 1.44269502162933349609375f, -t);
    lo = __builtin_fmaf(d, __builtin_bit_cast(float, 0x32a5705fu), lo);
    const float n = __builtin_rintf(t);
    const float f = (t - n) + lo;
    const float p = __builtin_amdgcn_exp2f(f);
    const float r = __builtin_ldexpf(p, (int)n);
    return __builtin_bit_cast(float, 0xc2ce8ed0u) > d ? 0.f : r;                     // (below -103.28 expf returns 0)
}

// One row of K logits in LDS (contiguous), worked on by FOUR adjacent lanes (sub = lane & 3): the logits are replaced by exp(x - max) and
// the row sum is returned to all four lanes. Lane `sub` walks k = sub, sub + 4, ...; the partial maxima / sums meet through two
// xor-shuffles (1, then 2): that order is part of the result's bits.
__device__ __forceinline__ float pp_softmax_row(float* __restrict__ row, const int K, const int sub, const bool valid) {
#pragma clang fp contract(off)
    // Batches of eight elements: the eight LDS reads, then the eight exponentials, are independent (the fused head epilogue runs at two
    // waves per SIMD: a read-use-read chain per element exposed every LDS round trip); only the additions form a chain, in ascending k --
    // the order of the plain loop `for (k = sub; k < K; k += 4) sm += exp(row[k] - mx)`, so the sum has the same bits.
    float mx = -INFINITY;
    if (valid)
        for (int k0 = sub; k0 < K; k0 += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = row[min(k0 + 4 * u, K - 1)];      // (a clamped index re-reads an element of this row: harmless for a maximum)
#pragma unroll
            for (int u = 0; u < 8; ++u) mx = fmaxf(mx, v[u]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 1));
    mx = fmaxf(mx, __shfl_xor(mx, 2));
    float sm = 0.f;
    if (valid)
        for (int k0 = sub; k0 < K; k0 += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = row[min(k0 + 4 * u, K - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = pp_exp_nonpos(v[u] - mx);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k0 + 4 * u < K) {
                    row[k0 + 4 * u] = v[u];
                    sm += v[u];
                }
        }
    sm += __shfl_xor(sm, 1);
    sm += __shfl_xor(sm, 2);
    return sm;
}

// score = exp(x - max) * (1 / sum): ONE correctly rounded division per row, one multiplication per class. (Rounds 1 - 4 divided per class: the
// quotient and this product differ by at most one unit in the last place, 6e-8 relative -- against the 1e-7 the reference's own vectorised expf
// already differs from this one's, and a 2e-6 tolerance on the scores; the ten instructions of an IEEE division per score were a third of the
// instruction count of the launch.) Both kernels use these two functions: same bits.
__device__ __forceinline__ float pp_row_rcp(const float rowsum) {
#pragma clang fp contract(off)
    return 1.0f / rowsum;
}
__device__ __forceinline__ float pp_score(const float e, const float row_rcp) {
#pragma clang fp contract(off)
    return e * row_rcp;
}

// histogram bin of a passing score: float bits >> HSHIFT, relative to the first reachable bin, clamped into [0, nb)
__device__ __forceinline__ int pp_hist_bin(const float sc, const int hb0, const int nb) {
    return min(max((int)(__float_as_uint(sc) >> DN_PP_HSHIFT) - hb0, 0), nb - 1);
}
