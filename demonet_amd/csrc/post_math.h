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

// One row of K logits in LDS (contiguous), worked on by FOUR adjacent lanes (sub = lane & 3): the logits are replaced by exp(x - max) and
// the row sum is returned to all four lanes. Lane `sub` walks k = sub, sub + 4, ...; the partial maxima / sums meet through two
// xor-shuffles (1, then 2): that order is part of the result's bits.
__device__ __forceinline__ float pp_softmax_row(float* __restrict__ row, const int K, const int sub, const bool valid) {
#pragma clang fp contract(off)
    float mx = -INFINITY;
    if (valid)
        for (int k = sub; k < K; k += 4) mx = fmaxf(mx, row[k]);
    mx = fmaxf(mx, __shfl_xor(mx, 1));
    mx = fmaxf(mx, __shfl_xor(mx, 2));
    float sm = 0.f;
    if (valid)
        for (int k = sub; k < K; k += 4) {
            const float e = expf(row[k] - mx);
            row[k] = e;
            sm += e;
        }
    sm += __shfl_xor(sm, 1);
    sm += __shfl_xor(sm, 2);
    return sm;
}

__device__ __forceinline__ float pp_score(const float e, const float rowsum) {
#pragma clang fp contract(off)
    return e / rowsum;
}

// histogram bin of a passing score: float bits >> HSHIFT, relative to the first reachable bin, clamped into [0, nb)
__device__ __forceinline__ int pp_hist_bin(const float sc, const int hb0, const int nb) {
    return min(max((int)(__float_as_uint(sc) >> DN_PP_HSHIFT) - hb0, 0), nb - 1);
}
