// Pointwise (1x1) convolution for short reductions (cin <= 256), register-direct: no LDS, no barriers.
//
// reference ops replaced: the expand / project ConvBNActivation 1x1 convs of InvertedResidual (mobilenetv3.py:72-95, BN folded),
//   the 1x1 convs of _extra_block (ssd_mobilenetv3.py:39-54) and of the V2 ExtraBlocks (backbone.py:100-111) -- the same layers
//   pointwise.hip's tiled kernel serves; this file is only a different schedule for them.
//
// Why: on these layers the tiled kernel is not bound by HBM or by the matrix cores but by VALU issue (tools/valu.sh: ~400 vector
// instructions per wave for 1024 outputs -- staging addresses, guards, the LDS round trips of operands and of the output tile, a
// bias add and an activation select per element; a wave64 VALU instruction holds its SIMD for 4 cycles). Here a wave owns 32
// pixels x (32 * TC) channels and
//   * both MFMA operands go from global memory / L2 straight into registers in fragment order (lane (r, hh) of
//     v_mfma_f32_32x32x16_f16 needs 16 contiguous bytes of row r: one global_load_dwordx4 with an immediate offset per K step,
//     every load of the wave requested before the first use -- ONE exposed memory round trip);
//   * the bias rides in the reduction: two extra K columns hold (1, 1) on the pixel side and (hi, lo) = the fp32 bias split into
//     two fp16 values on the weight side (|b - hi - lo| <= 2^-22 |b|), so the accumulator starts life as conv + bias;
//   * the activation is ONE uniform switch per 16-value accumulator, not a select per element;
//   * v_permlane32_swap turns the accumulator layout (4 channels per lane and register group) into 8-channel runs: the tile leaves as
//     two 16-byte stores per lane, each instruction writing 32 contiguous bytes per pixel, without an LDS transpose.
// The residual (fp16, NHWC) is requested up front in the accumulator layout and added in fp32 before the single rounding
// (mobilenetv3.py:97-99 computes act(conv + bias) + x).
#include <algorithm>

#include "common.h"

__device__ const uint4 g_pwdw_zero16 = {0u, 0u, 0u, 0u};      // what a tap outside the image reads

namespace {

typedef unsigned uint2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void act16(floatx16& v, int act) {
    // `v` comes straight from a matrix instruction and the first thing that happens to it is a scalar branch on `act`: hipcc's hazard recognizer gives the
    // straight path its 12 wait states but undercounts some of the branch paths (5 - 7: tools/mfma_hazard_scan.py --all). Twelve explicit wait states in
    // front of the switch cover every path (the matrix instruction runs under them; other waves issue meanwhile).
    asm volatile("s_nop 7\n\ts_nop 3" : "+v"(v));       // (tied to the accumulator: otherwise hipcc schedules it in front of the last matrix instruction)
    if (act == DN_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = dn_relu(v[e]);
    } else if (act == DN_ACT_RELU6) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = dn_relu6(v[e]);
    } else if (act == DN_ACT_HSWISH) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = v[e] * dn_relu6(v[e] + 3.f) * (1.f / 6.f);
    }
}

// KSF: full 16-deep K steps (cin >> 4, exact: no guards in the load / MFMA sequences); TC: 32-channel tiles per wave.
// Workgroup = 4 waves = (4 >> wc_log) row tiles of 32 pixels x (1 << wc_log) channel blocks of 32 * TC channels.
template <int KSF, int TC>
__global__ __launch_bounds__(256) void pw_direct_kernel(PwArgs a, int tiles, int wc_log) {
    constexpr int KSM = KSF > 0 ? KSF : 1;
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wc = wave & ((1 << wc_log) - 1), wp = wave >> wc_log;
    const int BP = 32 * (4 >> wc_log), BC = (32 * TC) << wc_log;
    const int K = a.cin, NC = a.cout;
    // row tile / channel block of this workgroup; XCD grouping as in pointwise.hip (pw_tile_rows): the workgroups with equal
    // (flat index % 8) own the rows of one group of a.xq images
    int m0, mend, by;
    {
        const int flat = blockIdx.x;
        if (a.xq > 0) {
            const int g = flat & 7, w = flat >> 3;
            by = w / tiles;
            const int t = w - by * tiles;
            const int r0 = g * a.xq * a.hw;
            mend = min(a.m, r0 + a.xq * a.hw);
            m0 = r0 + t * BP;
        } else {
            by = flat / tiles;
            m0 = (flat - by * tiles) * BP;
            mend = a.m;
        }
    }
    const int mrow0 = m0 + wp * 32;
    const int n_base = by * BC + wc * 32 * TC;
    if (mrow0 >= mend || n_base >= NC) return;            // wave-uniform
    const int row = mrow0 + r;
    const int rowc = min(row, mend - 1);
    // KSF full 16-deep steps; then ONE last step: the K tail (cin % 16 == 8) and the bias columns

    // ---- every load of the wave, up front
    const half_t* xp = a.x + (size_t)rowc * K + hh * 8;
    const half_t* wpt[TC];
    int nrow[TC];
#pragma unroll
    for (int i = 0; i < TC; ++i) {
        nrow[i] = min(n_base + i * 32 + r, NC - 1);
        wpt[i] = a.w + (size_t)nrow[i] * K + hh * 8;
    }
    half8 xf[KSM], wf[TC][KSM];
#pragma unroll
    for (int ks = 0; ks < KSF; ++ks) {
        xf[ks] = *reinterpret_cast<const half8*>(xp + ks * 16);
#pragma unroll
        for (int i = 0; i < TC; ++i) wf[i][ks] = *reinterpret_cast<const half8*>(wpt[i] + ks * 16);
    }
    // squeeze-excitation scale of the wave's image (a 32-row tile never straddles two images here: hw % 32 == 0, pw_direct_supported),
    // requested with the operands: s[k] for this lane's columns of every step
    const bool has_se = a.se != nullptr;
    float4 sv[KSM][2], svl[2];
    const float* sp = has_se ? a.se + (size_t)(mrow0 / a.hw) * K + hh * 8 : nullptr;
    if (has_se) {
#pragma unroll
        for (int ks = 0; ks < KSF; ++ks) {
            sv[ks][0] = *reinterpret_cast<const float4*>(sp + ks * 16);
            sv[ks][1] = *reinterpret_cast<const float4*>(sp + ks * 16 + 4);
        }
    }
    // last step: this lane's columns kb .. kb+7; kb < K: data (address clamped, value selected below); kb == K: the bias columns
    const int kb = KSF * 16 + hh * 8;
    const int kcl = min(kb, K - 8) - hh * 8;
    half8 xl = *reinterpret_cast<const half8*>(xp + kcl);
    if (has_se) {
        svl[0] = *reinterpret_cast<const float4*>(sp + kcl);
        svl[1] = *reinterpret_cast<const float4*>(sp + kcl + 4);
    }
    half8 wl[TC];
    float bl[TC];
#pragma unroll
    for (int i = 0; i < TC; ++i) {
        wl[i] = *reinterpret_cast<const half8*>(wpt[i] + kcl);
        bl[i] = a.bias[nrow[i]];
    }
    uint2 rres[TC][4];
    const bool has_res = a.residual != nullptr;
    if (has_res) {
        const half_t* rp = a.residual + (size_t)rowc * NC;
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                rres[i][g] = *reinterpret_cast<const uint2*>(rp + min(n_base + i * 32 + 8 * g + 4 * hh, NC - 4));
    }
    if (has_se) {
        // (half)((float)x * s): the rounding of the tiled kernel's squeeze-excitation path (pointwise.hip load_into)
        auto scale8 = [](half8 v, const float4& s0, const float4& s1) {
            const float s[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * s[e]);
            return v;
        };
#pragma unroll
        for (int ks = 0; ks < KSF; ++ks) xf[ks] = scale8(xf[ks], sv[ks][0], sv[ks][1]);
        xl = scale8(xl, svl[0], svl[1]);
    }
    {
        const bool data = kb < K, bcol = kb == K;
        const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        const half8 ones = {(half_t)1.f, (half_t)1.f, 0, 0, 0, 0, 0, 0};
        xl = data ? xl : (bcol ? ones : zero8);
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const half_t hi = (half_t)bl[i];
            const half_t lo = (half_t)(bl[i] - (float)hi);
            const half8 bw = {hi, lo, 0, 0, 0, 0, 0, 0};
            wl[i] = data ? wl[i] : (bcol ? bw : zero8);
        }
    }

    // ---- the reduction
    floatx16 acc[TC];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSF; ++ks) {
#pragma unroll
        for (int i = 0; i < TC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i][ks], xf[ks], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < TC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[i], xl, acc[i], 0, 0, 0);

    // ---- epilogue: lane = pixel r; registers 4g .. 4g+3 of tile i = channels n_base + 32 i + 8g + 4hh .. +3
    half_t* orow = reinterpret_cast<half_t*>(a.out) + (size_t)row * NC + hh * 8;
#pragma unroll
    for (int i = 0; i < TC; ++i) {
        const int nt = n_base + i * 32;
        if (nt >= NC) break;                               // wave-uniform
        act16(acc[i], a.act);
        if (has_res) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const half4 rr = __builtin_bit_cast(half4, rres[i][g]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][4 * g + e] += (float)rr[e];
            }
        }
        uint2v p[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            half4 hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) hv[e] = (half_t)acc[i][4 * g + e];
            p[g] = __builtin_bit_cast(uint2v, hv);
        }
        // lanes r / r + 32 hold the two halves of every 8-channel group of pixel r: swap so that lane r ends up with channels
        // 0..15 and lane r + 32 with channels 16..31 of the tile (v_permlane32_swap: upper half of the first operand <-> lower half of the second)
        uint4 lo4, hi4;
        {
            // (round 3: groups (0, 1) and (2, 3) are paired, so that lane r ends up with channels 0..7 / 16..23 and lane r + 32 with 8..15 /
            // 24..31: each store instruction then writes 32 CONTIGUOUS bytes per pixel -- whole sectors -- instead of two 16-byte pieces
            // 32 bytes apart)
            const uint2v s0 = __builtin_amdgcn_permlane32_swap(p[0][0], p[1][0], false, false);
            const uint2v s1 = __builtin_amdgcn_permlane32_swap(p[0][1], p[1][1], false, false);
            const uint2v s2 = __builtin_amdgcn_permlane32_swap(p[2][0], p[3][0], false, false);
            const uint2v s3 = __builtin_amdgcn_permlane32_swap(p[2][1], p[3][1], false, false);
            lo4 = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            hi4 = make_uint4(s2[0], s3[0], s2[1], s3[1]);
        }
        const int c0 = nt + hh * 8;
        if (row < mend) {
            if (c0 < NC) *reinterpret_cast<uint4*>(orow + nt) = lo4;
            if (c0 + 16 < NC) *reinterpret_cast<uint4*>(orow + nt + 16) = hi4;
        }
    }
}

// Streaming variant for the WIDE expansions (round 3): 112 -> 672 and 80 -> 480 on the 20 x 20 maps. pw_direct_kernel<7, 2> needs 186
// registers for "every operand of the wave up front" (2 waves per SIMD) and puts 8 800 waves through the chip in 4.3 rounds (24 us for a
// 40 MB layer). Here a wave keeps its 32 pixel rows (the B fragments) for a RUN of channel tiles and streams the weight tiles through two
// register sets: tile t + 1 is requested before the matrix instructions of tile t are issued, so after the first tile no round trip is
// exposed, and 800 pixel tiles x 3 channel runs = 2 400 waves fit the chip at once (~150 registers, 3 waves per SIMD). The four waves of a
// workgroup walk the same weight tiles at the same time (three of the four reads hit L1). Per-tile arithmetic, bias-in-the-reduction and
// the permlane epilogue are pw_direct_kernel's: the outputs are bit-identical to it. No squeeze-excitation scale, no residual (the
// expansions have neither).
// PX (round 4): 32-pixel tiles per wave. The four waves of a workgroup walk the same weight tiles, and every wave pulls them through the CU's
// L1 by itself: 13 M cache accesses (850 MB) for a 40 MB layer, and with the weight requests taken out the 112 -> 672 launch drops from 20.8 to
// 15.3 us (tools/time_pw.py on an ablated build) -- the L1's 64 B per clock, not L2 or HBM, is what the stream costs. With two pixel tiles per
// wave a weight fragment feeds two matrix instructions: half the L1 traffic per output.
template <int KSF, int PX>
__global__ __launch_bounds__(256) void pw_stream_kernel(PwArgs a, int tiles, int tiles_per_run) {
    constexpr int KSM = KSF > 0 ? KSF : 1;
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K = a.cin, NC = a.cout;
    int m0, mend, by;
    {
        const int flat = blockIdx.x;
        if (a.xq > 0) {
            const int g = flat & 7, w = flat >> 3;
            by = w / tiles;
            const int t = w - by * tiles;
            const int r0 = g * a.xq * a.hw;
            mend = min(a.m, r0 + a.xq * a.hw);
            m0 = r0 + t * (128 * PX);
        } else {
            by = flat / tiles;
            m0 = (flat - by * tiles) * (128 * PX);
            mend = a.m;
        }
    }
    const int mrow0 = m0 + wave * (32 * PX);
    const int ctiles = (NC + 31) >> 5;
    const int ct0 = by * tiles_per_run, ct1 = min(ctiles, ct0 + tiles_per_run);
    if (mrow0 >= mend || ct0 >= ct1) return;               // wave-uniform
    const int kb = KSF * 16 + hh * 8;
    const int kcl = min(kb, K - 8) - hh * 8;
    const bool data = kb < K, bcol = kb == K;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    half8 wf[2][KSM], wl[2];
    float bl[2];
    auto request = [&](const int ct, const int buf) {
        const int nrow = min(ct * 32 + r, NC - 1);
        const half_t* wp = a.w + (size_t)nrow * K + hh * 8;
#pragma unroll
        for (int ks = 0; ks < KSF; ++ks) wf[buf][ks] = *reinterpret_cast<const half8*>(wp + ks * 16);
        wl[buf] = *reinterpret_cast<const half8*>(wp + kcl);
        bl[buf] = a.bias[nrow];
    };
    half8 xf[PX][KSM], xl[PX];
    int row[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        row[j] = mrow0 + 32 * j + r;
        const half_t* xp = a.x + (size_t)min(row[j], mend - 1) * K + hh * 8;
#pragma unroll
        for (int ks = 0; ks < KSF; ++ks) xf[j][ks] = *reinterpret_cast<const half8*>(xp + ks * 16);
        xl[j] = *reinterpret_cast<const half8*>(xp + kcl);
    }
    request(ct0, 0);
    {
        const half8 ones = {(half_t)1.f, (half_t)1.f, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < PX; ++j) xl[j] = data ? xl[j] : (bcol ? ones : zero8);
    }
    auto tile = [&](const int ct, const int buf) {
        half8 wlast;
        {
            const half_t hi = (half_t)bl[buf];
            const half_t lo = (half_t)(bl[buf] - (float)hi);
            const half8 bw = {hi, lo, 0, 0, 0, 0, 0, 0};
            wlast = data ? wl[buf] : (bcol ? bw : zero8);
        }
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            floatx16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KSF; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[buf][ks], xf[j][ks], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlast, xl[j], acc, 0, 0, 0);
            act16(acc, a.act);
            uint2v p[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                half4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = (half_t)acc[4 * g + e];
                p[g] = __builtin_bit_cast(uint2v, hv);
            }
            const uint2v s0 = __builtin_amdgcn_permlane32_swap(p[0][0], p[1][0], false, false);
            const uint2v s1 = __builtin_amdgcn_permlane32_swap(p[0][1], p[1][1], false, false);
            const uint2v s2 = __builtin_amdgcn_permlane32_swap(p[2][0], p[3][0], false, false);
            const uint2v s3 = __builtin_amdgcn_permlane32_swap(p[2][1], p[3][1], false, false);
            const uint4 lo4 = make_uint4(s0[0], s1[0], s0[1], s1[1]), hi4 = make_uint4(s2[0], s3[0], s2[1], s3[1]);
            const int nt = ct * 32, c0 = nt + hh * 8;
            half_t* orow = reinterpret_cast<half_t*>(a.out) + (size_t)row[j] * NC + hh * 8;
            if (row[j] < mend) {
                if (c0 < NC) *reinterpret_cast<uint4*>(orow + nt) = lo4;
                if (c0 + 16 < NC) *reinterpret_cast<uint4*>(orow + nt + 16) = hi4;
            }
        }
    };
    // (the requests are UNCONDITIONAL, with a clamped tile index: behind a branch the compiler cannot count which loads are outstanding at
    // the join and waits for all of them -- vmcnt(0) before the first matrix instruction, i.e. no overlap at all; the price is one
    // redundant tile request at the end of a run)
    // (and pinned in front of the matrix instructions: left alone the scheduler sinks the next tile's loads below the current tile's MFMAs)
    for (int ct = ct0; ct < ct1; ct += 2) {
        request(min(ct + 1, ct1 - 1), 1);
        __builtin_amdgcn_sched_barrier(0);
        tile(ct, 0);
        if (ct + 1 < ct1) {
            request(min(ct + 2, ct1 - 1), 0);
            __builtin_amdgcn_sched_barrier(0);
            tile(ct + 1, 1);
        }
    }
}

template <int KSF, int PX>
int launch_stream_t(const PwArgs& a, int runs, int tiles_per_run, hipStream_t s) {
    const int tiles = a.xq > 0 ? dn_cdiv((long)a.xq * a.hw, 128 * PX) : dn_cdiv(a.m, 128 * PX);
    const dim3 grid((unsigned)(a.xq > 0 ? 8 * tiles : tiles) * runs);
    dn_note_kernel("pw_stream_kernel<%d,%d>", KSF, PX);
    hipLaunchKernelGGL((pw_stream_kernel<KSF, PX>), grid, dim3(256), 0, s, a, tiles, tiles_per_run);
    return DN_OK;
}

// Depthwise 3x3 (stride 1, pad 1) + the 1x1 projection behind it in ONE launch, register-direct: the first block of the MobileNets
// (mobilenetv3.py:61-99 with expanded == input channels: depthwise -> project (+ residual); mobilenetv2.py:57-84 with expand_ratio 1).
// Lane (r, hh) of the projection's B fragment holds 8 channels of ONE pixel -- exactly what a depthwise thread produces -- so the
// depthwise is computed straight into the fragment: nine 16-byte taps per lane and K step (neighbouring pixels, same channels; the
// wave covers 32 consecutive pixels of a row, the taps overlap in L1), bias first, taps in (ky, kx) order through v_fma_mix_f32,
// activation, ONE rounding to fp16 -- the arithmetic of dw_kernel, bit for bit -- and the value never exists in HBM: on the 160 x 160
// map that is 52 MB written and 52 MB read back per 64 images, and the residual (the block's input) is the centre tap's line.
// The projection (bias in the reduction, permlane epilogue) is pw_direct_kernel<KSF, 1>'s.
struct PwDwArgs {
    PwArgs pw;                       // x = the depthwise INPUT [n][h][w][cin]; cin == depthwise channels; residual: null or == x
    const half_t* wd; const float* bd;      // depthwise weights [9][cin] fp16, bias [cin] fp32
    int h, w, act_dw;
    float inv_hw;                    // 1 / (h w): image of a row by a corrected float quotient
    FastDiv fd_w;
};

template <int KSF>
__global__ __launch_bounds__(256) void pw_dw_direct_kernel(PwDwArgs q, int tiles) {
    const PwArgs& a = q.pw;
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K = a.cin, NC = a.cout;
    int m0, mend;
    {
        const int flat = blockIdx.x;
        if (a.xq > 0) {
            const int g = flat & 7, t = flat >> 3;
            const int r0 = g * a.xq * a.hw;
            mend = min(a.m, r0 + a.xq * a.hw);
            m0 = r0 + t * 128;
        } else {
            m0 = flat * 128;
            mend = a.m;
        }
    }
    const int mrow0 = m0 + wave * 32;
    if (mrow0 >= mend) return;                             // wave-uniform
    const int row = mrow0 + r;
    const int rowc = min(row, mend - 1);
    // position of this lane's pixel inside its image: the wave's first row by a corrected float quotient (wave-uniform), then at most
    // one wrap for the lane (hw >= 32)
    int rem;
    {
        int q0 = (int)((float)mrow0 * q.inv_hw);
        int r0 = mrow0 - q0 * a.hw;
        while (r0 < 0) r0 += a.hw;                            // (the float quotient is off by at most a few units even for 2^31 rows)
        while (r0 >= a.hw) r0 -= a.hw;
        rem = r0 + (rowc - mrow0);
        if (rem >= a.hw) rem -= a.hw;
    }
    const int py = (int)fd_div((unsigned)rem, q.fd_w), px = rem - py * q.w;
    // the projection's operands that do not depend on the depthwise: requested first
    const int nrow = min(r, NC - 1);
    const half_t* wpt = a.w + (size_t)nrow * K + hh * 8;
    half8 wf[KSF];
#pragma unroll
    for (int ks = 0; ks < KSF; ++ks) wf[ks] = *reinterpret_cast<const half8*>(wpt + ks * 16);
    const int kb = KSF * 16 + hh * 8;                      // == K + 8 hh: hh = 0 carries the bias columns, hh = 1 zeros
    const float bl = a.bias[nrow];
    uint2 rres[4];
    const bool has_res = a.residual != nullptr;
    if (has_res) {
        const half_t* rp = a.residual + (size_t)rowc * NC;
#pragma unroll
        for (int g = 0; g < 4; ++g) rres[g] = *reinterpret_cast<const uint2*>(rp + min(8 * g + 4 * hh, NC - 4));
    }
    // ---- depthwise into the B fragments
    typedef const __attribute__((address_space(1))) half8* gp8;
    const gp8 zero = (gp8)(&g_pwdw_zero16);
    const half_t* xpix = a.x + (size_t)rowc * K + hh * 8;           // this lane's channels of its own pixel
    half8 xf[KSF];
#pragma unroll
    for (int ks = 0; ks < KSF; ++ks) {
        // The launch is bound by the vector-memory pipe (22 wave-wide loads per wave for two MFMAs), and the depthwise weights / bias are the
        // same for every pixel: both 8-channel halves of a tap come through the scalar cache (uniform address, constant address space) and
        // the lane picks its half with selects -- nine vector loads less per lane.
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
        typedef float f4v __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(4))) u4v* cp4;
        typedef const __attribute__((address_space(4))) f4v* cpf4;
        half8 xin[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = py + ky - 1, ix = px + kx - 1;
                const bool ok = iy >= 0 && iy < q.h && ix >= 0 && ix < q.w;
                xin[ky * 3 + kx] = *(ok ? (gp8)(xpix + ks * 16 + ((ky - 1) * q.w + (kx - 1)) * K) : zero);
            }
        float acc[8];
        {
            const cpf4 bp = (cpf4)(q.bd + ks * 16);
            const f4v c0 = bp[0], c1 = bp[1], c2 = bp[2], c3 = bp[3];
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[e] = hh ? c2[e] : c0[e]; acc[4 + e] = hh ? c3[e] : c1[e]; }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // this tap's weights: both halves sit in scalar registers, the lane's half is selected right before its use (nine selected
            // fragments held at once cost 32 registers and three waves of occupancy)
            const cp4 wp = (cp4)(q.wd + t * K + ks * 16);
            const u4v w0 = wp[0], w1 = wp[1];
            const uint4 wsel = make_uint4(hh ? w1[0] : w0[0], hh ? w1[1] : w0[1], hh ? w1[2] : w0[2], hh ? w1[3] : w0[3]);
            fma_mix_h8(acc, *reinterpret_cast<const uint4*>(&xin[t]), wsel);
        }
        dn_act_n<float[8], 8>(acc, q.act_dw);
#pragma unroll
        for (int e = 0; e < 8; ++e) xf[ks][e] = (half_t)acc[e];
    }
    // ---- the projection: KSF full steps, then the bias step (pw_direct_kernel's last step with kb >= K)
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const half8 ones = {(half_t)1.f, (half_t)1.f, 0, 0, 0, 0, 0, 0};
    const half_t bhi = (half_t)bl;
    const half8 bw = {bhi, (half_t)(bl - (float)bhi), 0, 0, 0, 0, 0, 0};
    const bool bcol = kb == K;
    floatx16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSF; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks], xf[ks], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bcol ? bw : zero8, bcol ? ones : zero8, acc, 0, 0, 0);
    act16(acc, a.act);
    if (has_res) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const half4 rr = __builtin_bit_cast(half4, rres[g]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * g + e] += (float)rr[e];
        }
    }
    uint2v p[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        half4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[e] = (half_t)acc[4 * g + e];
        p[g] = __builtin_bit_cast(uint2v, hv);
    }
    uint4 lo4, hi4;
    {
        const uint2v s0 = __builtin_amdgcn_permlane32_swap(p[0][0], p[1][0], false, false);
        const uint2v s1 = __builtin_amdgcn_permlane32_swap(p[0][1], p[1][1], false, false);
        const uint2v s2 = __builtin_amdgcn_permlane32_swap(p[2][0], p[3][0], false, false);
        const uint2v s3 = __builtin_amdgcn_permlane32_swap(p[2][1], p[3][1], false, false);
        lo4 = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        hi4 = make_uint4(s2[0], s3[0], s2[1], s3[1]);
    }
    half_t* orow = reinterpret_cast<half_t*>(a.out) + (size_t)row * NC + hh * 8;
    const int c0 = hh * 8;
    if (row < mend) {
        if (c0 < NC) *reinterpret_cast<uint4*>(orow) = lo4;
        if (c0 + 16 < NC) *reinterpret_cast<uint4*>(orow + 16) = hi4;
    }
}


// Weight-stationary variant for the EXPANSIONS (round 6): cin <= 128, no squeeze-excitation scale, no residual -- 24 -> 72 on the 80 x 80 map,
// 40 -> 120, 80 -> 480, 112 -> 672 (mobilenetv3.py:76 expand ConvBNActivation) and the V2 expansions that run as launches of their own.
// What the lab (tools/pw_lab.hip, ablations + in-kernel stamps, LAB_NOTEBOOK R6) found about pw_stream_kernel on 112 -> 672 (17.2 us by rocprofv3):
// 6.5 us remain with every memory operation removed, the weight requests cost 4.5, the stores 5.1, the x requests 3.2 -- and they ADD, because (a) a
// wave's fragment loads are 32-byte pieces of 32 different rows per instruction (32 tag lookups each; with 8 - 12 waves per CU the lines leave L1
// between the K steps that use them: 13 M L1 accesses for a 40 MB layer), (b) loads and stores retire in order, so the wait for the next weight
// tile is also a wait for the previous tile's stores, (c) one round of workgroups runs load | compute | store in lockstep over the chip.
// Here
//   * a wave keeps the A fragments of its RUN of RT channel tiles in registers for its whole life (staged once per workgroup through LDS from
//     the fragment-major copy: 1 KB contiguous per LDS-DMA instruction; the bias joins as the (hi, lo) columns of the last K step);
//   * it walks over 32-row pixel tiles: a tile of x is 64 cin CONTIGUOUS bytes, fetched by LDS-DMA as whole cache lines into the wave's own LDS
//     buffer and read back as B fragments with ds_read_b128 -- the next tile's DMA is in flight during the matrix + epilogue work of this one;
//   * no vector-memory load is waited for inside a tile, and the wait at the end of a tile is COUNTED: this tile's stores (buffer stores with an
//     out-of-range offset for masked lanes, so their number is fixed) stay in flight behind it;
//   * the tile leaves as 64-byte pieces (v_permlane32_swap + v_permlane16_swap: four lanes per pixel row and store instruction).
// Arithmetic per output = pw_direct_kernel's (same K order, bias in the reduction, one rounding): bit-identical to it (test_pointwise_wstat_*).
__device__ __forceinline__ void pw_glds16(const void* gsrc, unsigned dst) {
    // LDS-DMA behind the compiler's back (headfuse.hip: as a builtin every later ds_read would wait for vmcnt(0)); M0 saved and restored
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

template <int KS1, int RT>
__global__ __launch_bounds__(256) void pw_wstat_kernel(PwArgs a, int runs, int wg_per_group) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pws_lds[];
    constexpr int KSF = KS1 - 1;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K = a.cin, NC = a.cout;
    const int ksw = (K + 15) >> 4;                       // K steps of the fragment-major copy (a tail step holds zeros in its upper half)
    // XCD grouping as everywhere (pw_tile_rows): the workgroups with equal (flat index % 8) own the rows of one group of a.xq images
    int r0, mend, w;
    if (a.xq > 0) {
        const int g = blockIdx.x & 7;
        w = blockIdx.x >> 3;
        r0 = g * a.xq * a.hw;
        mend = min(a.m, r0 + a.xq * a.hw);
    } else {
        w = blockIdx.x; r0 = 0; mend = a.m;
    }
    const int run = w % runs, wgq = w / runs;
    const int Q = (wg_per_group / runs) * 4;             // pixel tiles between two tiles of one wave
    const int n_units = mend > r0 ? (mend - r0 + 31) >> 5 : 0;
    const int ctiles = (NC + 31) >> 5;
    const int ct0 = run * RT;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)pws_lds;
    const unsigned xoff = (unsigned)(RT * ksw * 1024 + wave * ksw * 1024);          // this wave's x buffer: ksw pieces of 1 KB >= 64 K bytes
    auto dma_x = [&](int u) {
        const long row0 = (long)r0 + (long)u * 32;
        const unsigned char* src = reinterpret_cast<const unsigned char*>(a.x + row0 * K);
        const unsigned lim = (unsigned)min((long)(a.m - row0) * K * 2 - 16, 0x7fffffffL);           // the last 16 bytes of the tensor
#pragma unroll 1
        for (int p = 0; p < ksw; ++p) pw_glds16(src + min((unsigned)(p * 1024 + lane * 16), lim), lds0 + xoff + p * 1024);
    };
    int u = wgq * 4 + wave;
    if (u < n_units) dma_x(u);
    {
        const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.wfrag) + (size_t)ct0 * ksw * 1024 + lane * 16;
        const int pieces = max(min(RT, ctiles - ct0), 0) * ksw;
#pragma unroll 1
        for (int p = wave; p < pieces; p += 4) pw_glds16(wsrc + (size_t)p * 1024, lds0 + p * 1024);
    }
    float bl[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) bl[t] = a.bias[min((ct0 + t) * 32 + r, NC - 1)];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int kb = KSF * 16 + hh * 8;                    // first column of this lane in the last step: < K data, == K the bias columns, > K nothing
    const bool data = kb < K, bcol = kb == K;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    half8 wf[RT][KS1];
    {
        const half8* wl = reinterpret_cast<const half8*>(pws_lds) + lane;
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int tt = min(t, max(ctiles - ct0 - 1, 0));                          // (a run's tiles beyond the last: a valid tile again, never stored)
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) wf[t][ks] = wl[(tt * ksw + min(ks, ksw - 1)) * 64];
            const half_t hi = (half_t)bl[t];
            const half8 bw = {hi, (half_t)(bl[t] - (float)hi), 0, 0, 0, 0, 0, 0};
            wf[t][KSF] = data ? wf[t][KSF] : (bcol ? bw : zero8);
        }
    }
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)((unsigned)a.m * (unsigned)NC * 2u), 0x00020000);
    const half8 ones = {(half_t)1.f, (half_t)1.f, 0, 0, 0, 0, 0, 0};
    const unsigned xrd = xoff + (unsigned)r * (unsigned)K * 2u;                       // this lane's row in the wave's x buffer
    const unsigned xlast = (unsigned)min(kb, K - 8) * 2u;
    const int colq = 8 * ((lane >> 5) + 2 * ((lane >> 4) & 1));
    const int act = a.act;
    for (; u < n_units; u += Q) {
        half8 xf[KS1];
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) xf[ks] = *reinterpret_cast<const half8*>(pws_lds + xrd + (ks < KSF ? (unsigned)(ks * 32 + hh * 16) : xlast));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (u + Q < n_units) dma_x(u + Q);
        xf[KSF] = data ? xf[KSF] : (bcol ? ones : zero8);
        const int rr = r0 + u * 32 + (lane & 15);
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            floatx16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[t][ks], xf[ks], acc, 0, 0, 0);
            act16(acc, act);
            uint2v p[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                half4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = (half_t)acc[4 * g + e];
                p[g] = __builtin_bit_cast(uint2v, hv);
            }
            // permlane32: lane r <- channels 0..7 / 16..23, lane r + 32 <- 8..15 / 24..31 of pixel r (pw_direct_kernel); permlane16 on top: rows 0..15 of
            // the tile end up in one register set with FOUR lanes per row (64 contiguous bytes per row and store instruction), rows 16..31 in the other
            const uint2v s0 = __builtin_amdgcn_permlane32_swap(p[0][0], p[1][0], false, false);
            const uint2v s1 = __builtin_amdgcn_permlane32_swap(p[0][1], p[1][1], false, false);
            const uint2v s2 = __builtin_amdgcn_permlane32_swap(p[2][0], p[3][0], false, false);
            const uint2v s3 = __builtin_amdgcn_permlane32_swap(p[2][1], p[3][1], false, false);
            const uint2v q0 = __builtin_amdgcn_permlane16_swap(s0[0], s2[0], false, false);
            const uint2v q1 = __builtin_amdgcn_permlane16_swap(s1[0], s3[0], false, false);
            const uint2v q2 = __builtin_amdgcn_permlane16_swap(s0[1], s2[1], false, false);
            const uint2v q3 = __builtin_amdgcn_permlane16_swap(s1[1], s3[1], false, false);
            const u32x4 lo4 = {q0[0], q1[0], q2[0], q3[0]}, hi4 = {q0[1], q1[1], q2[1], q3[1]};
            const int col = (ct0 + t) * 32 + colq;
            const unsigned off = ((unsigned)rr * (unsigned)NC + (unsigned)col) * 2u;
            const bool okc = col < NC;
            __builtin_amdgcn_raw_buffer_store_b128(lo4, ors, (okc && rr < mend) ? off : 0x80000000u, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(hi4, ors, (okc && rr + 16 < mend) ? off + 32u * (unsigned)NC : 0x80000000u, 0, 0);
        }
        // the next tile of x must have landed; this tile's 2 RT stores stay in flight (always issued: masked lanes carry an out-of-range offset)
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * RT) : "memory");
    }
}

template <int KS1, int RT>
int launch_wstat_t(const PwArgs& a, hipStream_t s) {
    const int ctiles = dn_cdiv(a.cout, 32), runs = dn_cdiv(ctiles, RT);
    const int groups = a.xq > 0 ? 8 : 1;
    const long rows = a.xq > 0 ? (long)a.xq * a.hw : a.m;
    const int units = dn_cdiv(rows, 32);
    // two workgroups per compute unit (188 registers at KS1 RT = 24: two waves per SIMD), fewer when there are not that many pixel tiles
    int wpg = std::min(512 / groups, dn_cdiv(units, 4) * runs);
    wpg = std::max(runs, wpg / runs * runs);
    const int ksw = dn_cdiv(a.cin, 16);
    const int lds = (RT + 4) * ksw * 1024;
    dn_note_kernel("pw_wstat_kernel<%d,%d>", KS1, RT);
    hipLaunchKernelGGL((pw_wstat_kernel<KS1, RT>), dim3((unsigned)(groups * wpg)), dim3(256), lds, s, a, runs, wpg);
    return DN_OK;
}

static bool pw_wstat_supported_(const PwArgs& a) {
    return dn_knob("DN_PW_WSTAT", 1) != 0 && a.wfrag && !a.se && !a.residual && a.cin >= 16 && a.cin <= 128 && a.cin % 8 == 0 && a.m >= 3200 &&
           (long)a.m * a.cout * 2 < 0x7fffffffL && a.cout >= 64;
}

static int launch_pw_wstat_(const PwArgs& a, hipStream_t s) {
    const int ks1 = a.cin / 16 + 1, ctiles = dn_cdiv(a.cout, 32);
    // tiles per run: as many as 96 registers of A fragments allow, least padding of the last run first
    int rt = 2, waste = 1 << 30;
    for (int c = 4; c >= 2; --c) {
        if (ks1 * c > 24) continue;
        const int w = dn_cdiv(ctiles, c) * c - ctiles;
        if (w < waste) { waste = w; rt = c; }
    }
    switch (ks1 * 10 + rt) {
#define DN_PWS_CASE(k, t) case k * 10 + t: return launch_wstat_t<k, t>(a, s);
        DN_PWS_CASE(2, 2) DN_PWS_CASE(2, 3) DN_PWS_CASE(2, 4) DN_PWS_CASE(3, 2) DN_PWS_CASE(3, 3) DN_PWS_CASE(3, 4) DN_PWS_CASE(4, 2) DN_PWS_CASE(4, 3) DN_PWS_CASE(4, 4)
        DN_PWS_CASE(5, 2) DN_PWS_CASE(5, 3) DN_PWS_CASE(5, 4) DN_PWS_CASE(6, 2) DN_PWS_CASE(6, 3) DN_PWS_CASE(6, 4) DN_PWS_CASE(7, 2) DN_PWS_CASE(7, 3)
        DN_PWS_CASE(8, 2) DN_PWS_CASE(8, 3) DN_PWS_CASE(9, 2)
#undef DN_PWS_CASE
    }
    return DN_E_UNSUPPORTED;
}

template <int KSF, int TC>
int launch_t(const PwArgs& a, int wc_log, hipStream_t s) {
    const int BP = 32 * (4 >> wc_log), BC = (32 * TC) << wc_log;
    const int tiles = a.xq > 0 ? dn_cdiv((long)a.xq * a.hw, BP) : dn_cdiv(a.m, BP);
    const dim3 grid((unsigned)(a.xq > 0 ? 8 * tiles : tiles) * dn_cdiv(a.cout, BC));
    dn_note_kernel("pw_direct_kernel<%d,%d>", KSF, TC);
    hipLaunchKernelGGL((pw_direct_kernel<KSF, TC>), grid, dim3(256), 0, s, a, tiles, wc_log);
    return DN_OK;
}

}  // namespace

bool pw_dw_direct_supported(const PwArgs& a, const DwArgs& d) {
    return dn_knob("DN_PW_DW", 1) != 0 && pw_direct_supported(a) && !a.se && (a.cin == 16 || a.cin == 32) && a.cout <= 32 && d.c == a.cin && d.k == 3 &&
           d.stride == 1 && d.pad == 1 && !d.pool && d.ho == d.h && d.wo == d.w_ && a.hw == d.h * d.w_ && (!a.residual || (a.residual == d.x && a.cout == a.cin)) &&
           a.hw >= 32 && fd_ok((unsigned long long)a.hw, (unsigned)d.w_);
}

int launch_pw_dw_direct(const PwArgs& a, const DwArgs& d, hipStream_t s) {
    DN_REQUIRE(pw_dw_direct_supported(a, d), "depthwise + pointwise (direct): unsupported cin=%d cout=%d", a.cin, a.cout);
    PwDwArgs q{};
    q.pw = a;
    q.pw.x = d.x;
    q.wd = d.w; q.bd = d.bias; q.h = d.h; q.w = d.w_; q.act_dw = d.act;
    q.inv_hw = 1.0f / (float)a.hw; q.fd_w = fastdiv((unsigned)d.w_);
    const int tiles = a.xq > 0 ? dn_cdiv((long)a.xq * a.hw, 128) : dn_cdiv(a.m, 128);
    const dim3 grid((unsigned)(a.xq > 0 ? 8 * tiles : tiles));
    dn_note_kernel("pw_dw_direct_kernel<%d>", a.cin / 16);
    if (a.cin == 16) hipLaunchKernelGGL((pw_dw_direct_kernel<1>), grid, dim3(256), 0, s, q, tiles);
    else hipLaunchKernelGGL((pw_dw_direct_kernel<2>), grid, dim3(256), 0, s, q, tiles);
    return DN_OK;
}

bool pw_direct_supported(const PwArgs& a) {
    // squeeze-excitation scaled inputs: only where a 32-row tile lies inside one image (the 40 x 40 maps) and the reduction is short
    if (a.se && !(a.hw % 32 == 0 && a.cin <= 128)) return false;
    return dn_knob("DN_PW_DIRECT", 1) != 0 && a.cv_k == 1 && !a.out_fp32 && !a.sef_part && !a.w_b && a.cin % 8 == 0 && a.cin >= 8 &&
           a.cin <= 256 && a.cout % 8 == 0 && a.cout >= 8 && !(a.act >> 8) && a.out_img_stride == 0 && a.out_base == 0;
}

int launch_pw_direct(const PwArgs& a, hipStream_t s) {
    DN_REQUIRE(pw_direct_supported(a), "pointwise (direct): unsupported cin=%d cout=%d", a.cin, a.cout);
    // One 32-channel tile per wave (TC = 1) on the narrow layers: these launches are latency-bound, and twice the waves with half the
    // registers overlap their single memory round trip better. Wide expansions (cout >= DN_PW_DIRECT_TC2, short reductions) take two
    // tiles per wave: every wave re-reads its x rows once per channel tile, and at 21 tiles (112 -> 672) that is most of the traffic
    // (measured: threshold 400 -> batch 64 1.115 -> 1.107 ms, batch 32 0.79 -> 0.77 ms; 200 and 600 in between).
    // the expansions (short reduction, wide output, no scale, no residual): weight-stationary waves over LDS-DMA'd pixel tiles (round 6)
    if (pw_wstat_supported_(a)) return launch_pw_wstat_(a, s);
    const int ctiles = dn_cdiv(a.cout, 32);
    const int ksf = a.cin >> 4;
    // wide expansions with enough rows: the streaming variant (pw_stream_kernel) -- channel runs sized so that all waves are resident at once
    if (dn_knob("DN_PW_STREAM", 1) && !a.se && !a.residual && ksf >= 4 && ksf <= 8 && ctiles >= 12 && a.m >= 12800) {
        const int px = 2;                   // 32-pixel tiles per wave
        const long ptiles = dn_cdiv(a.m, 32 * px);
        int runs = (int)std::max(1L, std::min((long)ctiles, (long)2800 / ptiles));
        const int per = dn_cdiv(ctiles, runs);
        runs = dn_cdiv(ctiles, per);
        switch (ksf * 10 + px) {
            case 42: return launch_stream_t<4, 2>(a, runs, per, s);
            case 52: return launch_stream_t<5, 2>(a, runs, per, s);
            case 62: return launch_stream_t<6, 2>(a, runs, per, s);
            case 72: return launch_stream_t<7, 2>(a, runs, per, s);
            case 82: return launch_stream_t<8, 2>(a, runs, per, s);
        }
    }
    if (ksf <= 8 && a.cout >= 400) {
        const int wc_log = ctiles <= 2 ? 0 : ctiles <= 4 ? 1 : 2;
        switch (ksf) {
#define DN_PWD_CASE2(k) case k: return launch_t<k, 2>(a, wc_log, s);
            DN_PWD_CASE2(0) DN_PWD_CASE2(1) DN_PWD_CASE2(2) DN_PWD_CASE2(3) DN_PWD_CASE2(4) DN_PWD_CASE2(5) DN_PWD_CASE2(6) DN_PWD_CASE2(7) DN_PWD_CASE2(8)
#undef DN_PWD_CASE2
        }
    }
    const int wc_log = ctiles <= 1 ? 0 : ctiles <= 2 ? 1 : 2;
    switch (ksf) {
#define DN_PWD_CASE(k) case k: return launch_t<k, 1>(a, wc_log, s);
        DN_PWD_CASE(0) DN_PWD_CASE(1) DN_PWD_CASE(2) DN_PWD_CASE(3) DN_PWD_CASE(4) DN_PWD_CASE(5) DN_PWD_CASE(6) DN_PWD_CASE(7) DN_PWD_CASE(8)
        DN_PWD_CASE(9) DN_PWD_CASE(10) DN_PWD_CASE(11) DN_PWD_CASE(12) DN_PWD_CASE(13) DN_PWD_CASE(14) DN_PWD_CASE(15) DN_PWD_CASE(16)
#undef DN_PWD_CASE
    }
    return DN_E_UNSUPPORTED;
}
