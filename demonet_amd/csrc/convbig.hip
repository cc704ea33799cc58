// Dense kxk convolution as an implicit GEMM with a 256x256 workgroup tile -- the MFMA-bound layers of the VGG models
// (ssd_vgg16.py:30-109: conv3_x .. fc7; 256..1024 output channels, cin % 64 == 0).
//
// Why a second kernel beside pointwise.hip's CONV path: there a wave owns a 64x64 tile and reads 1 KB of LDS per MFMA, i.e.
// 128 B/clk per CU at full matrix rate -- the LDS port saturates together with the matrix pipe (440-560 TFLOP/s measured).
// Here a wave owns 128 pixels x 128 channels (16 accumulator tiles = 256 AGPRs): 512 B of LDS per MFMA, each weight / pixel
// fragment feeds four MFMAs. One workgroup (4 waves, one per SIMD) per CU; with a single wave per SIMD nothing hides a wait
// but the wave's own MFMAs, so every LDS read and every staging instruction is placed between MFMAs
// (__builtin_amdgcn_sched_group_barrier) and the only barrier of a stage sits where its waits have long been satisfied.
//   K runs over (ky, kx, cin) in 64-deep stages (a stage lies inside one tap); the pixel tile of a stage is the tap-shifted NHWC
//   rows (taps outside the image read a block of zeros kept behind the weights), the weight tile 256 rows of [cout][ky][kx][cin].
//   Operands are swapped as in pointwise.hip (A = weights, B = pixels): lane = pixel, 4 consecutive channels per register
//   group; the epilogue stages the tile in LDS and writes 16-byte row-contiguous chunks.
// Staging is LDS-DMA (global_load_lds_dwordx4): a register-staged version of the same loop (global_load -> ds_write_b128, 32-deep
// stages) spent a quarter of its time in the ds_write_b128 issue (~13 cycles per wave-instruction) and ran level with this one;
// the DMA form needs no staging registers, which is what makes the 64-deep stage (half the barriers) fit.
// Measured ablations of conv_glds_kernel's loop (conv3_2 of ssd300_vgg16, TFLOP/s): as is 830; without the staging traffic 1330 --
// the 256x256x64 stage moves 64 KB from L2 for 8.4 MFLOP (128 FLOP/B), so at 830 TFLOP/s the CUs pull 6.5 TB/s out of L2, and the 9
// taps re-read every input row. conv_halo_kernel (below) stages each input row once per channel slice instead: 940 TFLOP/s.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#ifdef DN_DEV_STAMPS
static long long* g_patch_stamps = nullptr;     // dev build only (tools/probe_patch.py): per-workgroup phase stamps [workgroup][8]
static size_t g_patch_stamps_used = 0;          // every launch takes its own [workgroups][8] region behind the previous one's
extern "C" __attribute__((visibility("default"))) void dn_debug_patch_stamps(void* dev_ptr) { g_patch_stamps = (long long*)dev_ptr; g_patch_stamps_used = 0; }
static long long* patch_stamps_take(size_t workgroups) {
    if (!g_patch_stamps) return nullptr;
    long long* p = g_patch_stamps + g_patch_stamps_used;
    g_patch_stamps_used += workgroups * 8;
    return p;
}
#define CP_STAMP(k) do { if (stamps && threadIdx.x == 0) stamps[(size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#define CP_STAMP_IF(c, k) do { if (c) CP_STAMP(k); } while (0)
#else
#define CP_STAMP_IF(c, k) do { } while (0)
static long long* patch_stamps_take(size_t) { return nullptr; }
#define CP_STAMP(k) do { } while (0)
#endif

namespace {

constexpr int BP = 256, BC = 256;
constexpr int OROW = BC + 8;

// bias + activation, tile -> LDS [BP][BC+8] halfs (over the stage buffers; the caller has passed a barrier), then 16-B
// row-contiguous stores
template <int TP, int TC>
__device__ __forceinline__ void conv_epilogue(floatx16 (&acc)[TC][TP], const PwArgs& a, half_t* lds, const float* bsh, int m0, int n0) {
    constexpr int BPt = 64 * TP, BCt = 64 * TC, ORW = BCt + 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave >> 1, wc = wave & 1;
    const int r = lane & 31, hh = lane >> 5;
    const int M = a.m, NC = a.cout;
    half_t* ot = lds;
    // per channel tile: the lane's 16 bias values first (ONE wait for four LDS reads), and one uniform branch on the activation around everything:
    // per 4-value chunk the round-4 form read its bias, waited, and walked the activation switch -- 16 TC dependent LDS round trips per tile
    // (tools/probe_patch.py)
    auto emit = [&](auto actf) {
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            float4 bq[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const float4*>(&bsh[(wc * TC + i) * 32 + 8 * g + 4 * hh]);
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int prow = (wp * TP + j) * 32 + r;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = (wc * TC + i) * 32 + 8 * g + 4 * hh;
                    const float4 bv = bq[g];
                    half4 hv;
                    hv[0] = (half_t)actf(acc[i][j][4 * g + 0] + bv.x); hv[1] = (half_t)actf(acc[i][j][4 * g + 1] + bv.y);
                    hv[2] = (half_t)actf(acc[i][j][4 * g + 2] + bv.z); hv[3] = (half_t)actf(acc[i][j][4 * g + 3] + bv.w);
                    *reinterpret_cast<half4*>(&ot[prow * ORW + cl]) = hv;
                }
            }
        }
    };
    const int act = a.act;
    if (act == DN_ACT_RELU) emit([](float v) { return dn_relu(v); });
    else if (act == DN_ACT_NONE) emit([](float v) { return v; });
    else emit([act](float v) { return dn_act(v, act); });
    __syncthreads();
    if constexpr (TP == 4 && TC == 4) if (a.pool_out) {        // (compiled into the 256 x 256 tile only: in the 512 x 128 one it spilled the main loop)
        // fused MaxPool2d(2, 2) (ssd_vgg16.py:34-37 via torchvision vgg16 features): the tile is BPt / W whole image rows starting at an even row
        // (launcher: BPt % 2W == 0, H W % BPt == 0), so every 2 x 2 window lies inside it; the conv output itself never reaches HBM
        const int W = a.cv_w, Wp = W >> 1, HWp = (a.cv_h >> 1) * Wp;
        const int img = m0 / a.hw, y0 = (m0 - img * a.hw) / W;
        half_t* pout = a.pool_out + ((size_t)img * HWp + (size_t)(y0 >> 1) * Wp) * NC + n0;
        for (int c = tid; c < (BPt / 4) * (BCt / 8); c += 256) {
            const int pp = c / (BCt / 8), ch = c - pp * (BCt / 8);
            const int pr = pp / Wp, pc = pp - pr * Wp;
            const int p00 = 2 * pr * W + 2 * pc;
            const half8 v0 = *reinterpret_cast<const half8*>(&ot[p00 * ORW + ch * 8]);
            const half8 v1 = *reinterpret_cast<const half8*>(&ot[(p00 + 1) * ORW + ch * 8]);
            const half8 v2 = *reinterpret_cast<const half8*>(&ot[(p00 + W) * ORW + ch * 8]);
            const half8 v3 = *reinterpret_cast<const half8*>(&ot[(p00 + W + 1) * ORW + ch * 8]);
            half8 mx;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const half_t m01 = v0[e] > v1[e] ? v0[e] : v1[e], m23 = v2[e] > v3[e] ? v2[e] : v3[e];
                mx[e] = m01 > m23 ? m01 : m23;
            }
            *reinterpret_cast<half8*>(pout + (size_t)(pr * Wp + pc) * NC + ch * 8) = mx;
        }
        if (!a.out) return;         // (a conv output with other readers -- conv4_3 feeds the L2-norm level -- is written as well)
    }
    half_t* outp = reinterpret_cast<half_t*>(a.out);
#pragma unroll 4
    for (int c = tid; c < BPt * (BCt / 8); c += 256) {
        const int row = c / (BCt / 8), ch = c % (BCt / 8);
        const int m = m0 + row;
        if (m < M) *reinterpret_cast<uint4*>(outp + (size_t)m * NC + n0 + ch * 8) = *reinterpret_cast<const uint4*>(&ot[row * ORW + ch * 8]);
    }
}

// Head outputs (generalized_ssd.py:60-74): fp32 rows of the [anchor][class] / [anchor][4] arrays, addressed
// out_base + image * out_img_stride + (pixel in image) * cout + channel. The accumulator layout has every lane on a different row:
// the tile goes through LDS in two halves of BPt / 2 pixel rows ([rows][BCt + 4] floats) and leaves as row-contiguous float2 runs.
template <int TP, int TC>
__device__ __forceinline__ void conv_epilogue_fp32(floatx16 (&acc)[TC][TP], const PwArgs& a, half_t* lds, const float* bsh, int m0, int n0) {
    constexpr int BPt = 64 * TP, BCt = 64 * TC, FROW = BCt + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave >> 1, wc = wave & 1;
    const int r = lane & 31, hh = lane >> 5;
    const int M = a.m, NC1 = a.cout, NCT = a.cout + a.cout_b;
    float* ot = reinterpret_cast<float*>(lds);
    auto pair_aligned = [](const void* p, int nc, long base, long stride) {
        return ((nc | (int)(base & 1) | (int)(stride & 1)) & 1) == 0 && (reinterpret_cast<size_t>(p) & 7) == 0;
    };
    const bool pair1 = pair_aligned(a.out, a.cout, a.out_base, a.out_img_stride);
    const bool pair2 = a.cout_b > 0 && pair_aligned(a.out_b, a.cout_b, a.out_b_base, a.out_b_img_stride);
    for (int half = 0; half < 2; ++half) {
        if (wp == half) {
            // (bias per channel tile in one batch, one activation branch: see conv_epilogue)
            auto emit = [&](auto actf) {
#pragma unroll
                for (int i = 0; i < TC; ++i) {
                    float4 bq[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const float4*>(&bsh[(wc * TC + i) * 32 + 8 * g + 4 * hh]);
#pragma unroll
                    for (int j = 0; j < TP; ++j) {
                        const int prow = j * 32 + r;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int cl = (wc * TC + i) * 32 + 8 * g + 4 * hh;
                            const float4 bv = bq[g];
                            float4 v;
                            v.x = actf(acc[i][j][4 * g + 0] + bv.x); v.y = actf(acc[i][j][4 * g + 1] + bv.y);
                            v.z = actf(acc[i][j][4 * g + 2] + bv.z); v.w = actf(acc[i][j][4 * g + 3] + bv.w);
                            *reinterpret_cast<float4*>(&ot[prow * FROW + cl]) = v;
                        }
                    }
                }
            };
            const int act = a.act;
            if (act == DN_ACT_NONE) emit([](float v) { return v; });
            else emit([act](float v) { return dn_act(v, act); });
        }
        __syncthreads();
        constexpr int PPR = BCt / 2;
#pragma unroll 4
        for (int c = tid; c < (BPt / 2) * PPR; c += 256) {
            const int row = c / PPR, cp = (c - row * PPR) * 2;
            const int m = m0 + half * (BPt / 2) + row, n = n0 + cp;
            if (m >= M || n >= NCT) continue;
            const int img = m / a.hw;
            const float2 v = *reinterpret_cast<const float2*>(&ot[row * FROW + cp]);
            // columns [0, cout) belong to the first head, [cout, cout + cout_b) to the second (cout is even: a pair never straddles)
            const bool second = n >= NC1;
            const int nc = second ? a.cout_b : NC1, nn = second ? n - NC1 : n;
            float* o = (second ? reinterpret_cast<float*>(a.out_b) + (size_t)a.out_b_base + (size_t)img * a.out_b_img_stride
                               : reinterpret_cast<float*>(a.out) + (size_t)a.out_base + (size_t)img * a.out_img_stride) +
                       (size_t)(m - img * a.hw) * nc + nn;
            if ((second ? pair2 : pair1) && nn + 1 < nc) {
                *reinterpret_cast<float2*>(o) = v;
            } else {
                o[0] = v.x;
                if (nn + 1 < nc) o[1] = v.y;
            }
        }
        __syncthreads();
    }
}

// An LDS-DMA wave-instruction writes 1 KB linearly (lane l -> base + 16 l): the stage image is rows of 128 B without padding, 8 rows
// per instruction, and the bank-conflict swizzle goes on the SOURCE: position p of row r holds K-chunk p ^ ((r >> 1) & 7); the
// fragment reads apply the same XOR (conflict-free for the 16-lane groups of ds_read_b128).
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr int GK = 64;                      // K per stage
constexpr int GSTAGE = (BP + BC) * GK;      // halfs per stage buffer (64 KB)

// HEAD: the 1x1 class head of a large pyramid level with fp32 [anchor][class] output (SSDLite, generalized_ssd.py:60-74): any
// cin % 8 == 0 (the chunks of the last 64-deep stage beyond cin are fetched from the zero block on both sides), weight rows beyond the
// last channel repeat the last row and are never stored, a second head on the same input may follow in the channel range (cout_b).
template <bool HEAD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_glds_kernel(PwArgs a) {
    extern __shared__ __attribute__((aligned(16))) half_t lds_raw[];
    float* bsh = reinterpret_cast<float*>(lds_raw);
    half_t* lds = lds_raw + 2 * BC;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave >> 1, wc = wave & 1;
    const int r = lane & 31, hh = lane >> 5;
    const int m0 = blockIdx.x * BP, n0 = blockIdx.y * BC;
    const int M = a.m, K = a.cin, CIN = a.cv_cin;
    const int NC = HEAD ? a.cout + a.cout_b : a.cout;
    const int KT = HEAD ? (K + GK - 1) / GK : K / GK;

    floatx16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // loader geometry: wave w fills rows w*64 .. w*64+63 of the pixel tile and of the weight tile, 8 rows per instruction:
    // lane -> row j*8 + (lane >> 3), position lane & 7, source chunk = position ^ ((row >> 1) & 7)
    const int lrow = lane >> 3, lpos = lane & 7;
    const char* xbase = reinterpret_cast<const char*>(a.x);
    const char* zeros = reinterpret_cast<const char*>(a.zeros);
    int xoff[8];
    unsigned vmask[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = wave * 64 + j * 8 + lrow;
        const int chunk = lpos ^ ((row >> 1) & 7);
        const int m = min(m0 + row, M - 1);           // rows beyond M: computed, never stored
        const int img = m / a.hw, rem = m - img * a.hw;
        const int oy = rem / a.cv_wo, ox = rem - oy * a.cv_wo;
        const int iy0 = oy * a.cv_stride - a.cv_pad, ix0 = ox * a.cv_stride - a.cv_pad;
        xoff[j] = (((img * a.cv_h + iy0) * a.cv_w + ix0) * CIN + chunk * 8) * 2;
        unsigned mk = 0;
        for (int ky = 0; ky < a.cv_k; ++ky)
            for (int kx = 0; kx < a.cv_k; ++kx) {
                const int iy = iy0 + ky * a.cv_dil, ix = ix0 + kx * a.cv_dil;
                if (iy >= 0 && iy < a.cv_h && ix >= 0 && ix < a.cv_w) mk |= 1u << (ky * a.cv_k + kx);
            }
        vmask[j] = mk;
    }
    // weights: row (n0 + w*64 + j*8 + lrow), chunk as above: even / odd j differ by chunk ^ 4
    const char* wbase = reinterpret_cast<const char*>(a.w) + (HEAD ? 0 : ((size_t)(n0 + wave * 64) * K) * 2);
    const int chE = lpos ^ ((lrow >> 1) & 3), chO = lpos ^ (4 + ((lrow >> 1) & 3));       // K chunk of this lane in an even / odd row group
    const unsigned woffE = (unsigned)((lrow * K + chE * 8) * 2);
    const unsigned woffO = (unsigned)((lrow * K + chO * 8) * 2);
    long wro[HEAD ? 8 : 1];                 // HEAD: byte offset of this lane's weight row (clamped to the last channel; second head behind the first)
    if constexpr (HEAD) {
        const long wdelta = a.w_b ? reinterpret_cast<const char*>(a.w_b) - reinterpret_cast<const char*>(a.w) : 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = min(n0 + wave * 64 + j * 8 + lrow, NC - 1);
            wro[j] = (row < a.cout ? (long)row * K * 2 : wdelta + (long)(row - a.cout) * K * 2) + ((j & 1) ? chO : chE) * 16;
        }
    }

    int ld_c0 = 0, ld_tap = 0, ld_ky = 0, ld_kx = 0, ld_k0 = 0;     // loader position, advanced without divisions
    // a stage = 16 LDS-DMA instructions per wave (8 pixel-row groups, 8 weight-row groups), issued in three parts so that they
    // can sit between the MFMAs of three K steps
    int st_toff = 0, st_tap = 0, st_k0 = 0, st_c0 = 0;
    auto stage_begin = [&]() {
        st_toff = (((ld_ky * a.cv_dil) * a.cv_w + ld_kx * a.cv_dil) * CIN + ld_c0) * 2;
        st_tap = ld_tap;
        st_k0 = ld_k0;
        st_c0 = ld_c0;
        ld_k0 += GK * 2;
        ld_c0 += GK;
        const int wrap = ld_c0 == CIN;
        ld_c0 = wrap ? 0 : ld_c0;
        ld_tap += wrap;
        ld_kx += wrap;
        const int wrapx = ld_kx == a.cv_k;
        ld_kx = wrapx ? 0 : ld_kx;
        ld_ky += wrapx;
    };
    auto issue_x = [&](int b, int j) {
        half_t* dst = lds + b * GSTAGE + wave * 64 * GK;             // wave-uniform
        bool ok = (vmask[j] >> st_tap) & 1u;
        if constexpr (HEAD) ok = ok && st_c0 + ((j & 1) ? chO : chE) * 8 < CIN;      // (1x1 only: K == CIN)
        const char* p = ok ? xbase + (long)(xoff[j] + st_toff) : zeros;
        __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(dst + j * 8 * GK), 16, 0, 0);
    };
    auto issue_w = [&](int b, int j) {
        half_t* dst = lds + b * GSTAGE + wave * 64 * GK;
        const char* p;
        if constexpr (HEAD) p = st_c0 + ((j & 1) ? chO : chE) * 8 < K ? wbase + (wro[j] + (long)st_k0) : zeros;
        else p = wbase + (size_t)(j * 8) * K * 2 + (((j & 1) ? woffO : woffE) + (unsigned)st_k0);
        __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(dst + BP * GK + j * 8 * GK), 16, 0, 0);
    };
    auto issue_part = [&](int b, int part) {        // part 0: x0-5, part 1: x6-7 w0-3, part 2: w4-7 (6 + 6 + 4)
        if (part == 0) { issue_x(b, 0); issue_x(b, 1); issue_x(b, 2); issue_x(b, 3); issue_x(b, 4); issue_x(b, 5); }
        if (part == 1) { issue_x(b, 6); issue_x(b, 7); issue_w(b, 0); issue_w(b, 1); issue_w(b, 2); issue_w(b, 3); }
        if (part == 2) { issue_w(b, 4); issue_w(b, 5); issue_w(b, 6); issue_w(b, 7); }
    };

    if constexpr (HEAD) { const int n = n0 + tid; bsh[tid] = n < a.cout ? a.bias[n] : n < NC ? a.bias_b[n - a.cout] : 0.f; }
    else bsh[tid] = a.bias[n0 + tid];

    // fragment addresses: row base + ((2 ks + hh) ^ ((r >> 1) & 7)) * 16 bytes
    const int sw = (r >> 1) & 7;
    const half_t* xrow = lds + ((wp * 4) * 32 + r) * GK;
    const half_t* wrow = lds + (BP + (wc * 4) * 32 + r) * GK;
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = ((2 * ks + hh) ^ sw) * 8;
    half8 xfa[4], wfa[4], xfb[4], wfb[4];
    auto read_frags = [&](half8 (&xf)[4], half8 (&wf)[4], int b, int ks) {
#pragma unroll
        for (int j = 0; j < 4; ++j) xf[j] = *reinterpret_cast<const half8*>(xrow + b * GSTAGE + j * 32 * GK + koff[ks]);
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const half8*>(wrow + b * GSTAGE + i * 32 * GK + koff[ks]);
    };
    auto mfma16 = [&](const half8 (&xf)[4], const half8 (&wf)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    };

#define CG_PHASE(nvm)                                                 \
    do {                                                               \
        for (int u_ = 0; u_ < 8; ++u_) {                               \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);         \
        }                                                              \
        for (int u_ = 0; u_ < (nvm); ++u_) {                           \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);         \
        }                                                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 8 - (nvm), 0);     \
        __builtin_amdgcn_sched_barrier(0);                             \
    } while (0)

    // One barrier per stage, before the MFMAs of the last K step:
    //   K steps 0..2: MFMAs | fragment reads of the next K step | the LDS-DMA of stage kt+1 into the other buffer
    //   wait for the DMA, barrier (everyone is done reading this buffer; the other one is complete)
    //   K step 3: MFMAs | fragment reads of K step 0 of stage kt+1
    stage_begin();
    issue_part(0, 0); issue_part(0, 1); issue_part(0, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    read_frags(xfa, wfa, 0, 0);
    int kt = 0;
    for (; kt + 1 < KT; ++kt) {
        const int b = kt & 1;
        stage_begin();
        read_frags(xfb, wfb, b, 1);
        issue_part(b ^ 1, 0);
        mfma16(xfa, wfa);
        CG_PHASE(6);
        read_frags(xfa, wfa, b, 2);
        issue_part(b ^ 1, 1);
        mfma16(xfb, wfb);
        CG_PHASE(6);
        read_frags(xfb, wfb, b, 3);
        issue_part(b ^ 1, 2);
        mfma16(xfa, wfa);
        CG_PHASE(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        read_frags(xfa, wfa, b ^ 1, 0);
        mfma16(xfb, wfb);
        CG_PHASE(0);
    }
    {
        const int b = kt & 1;
        read_frags(xfb, wfb, b, 1);
        mfma16(xfa, wfa);
        read_frags(xfa, wfa, b, 2);
        mfma16(xfb, wfb);
        read_frags(xfb, wfb, b, 3);
        mfma16(xfa, wfa);
        mfma16(xfb, wfb);
    }
    __syncthreads();

    if constexpr (HEAD) conv_epilogue_fp32<4, 4>(acc, a, lds, bsh, m0, n0);
    else conv_epilogue<4, 4>(acc, a, lds, bsh, m0, n0);
}


// ---- 3x3-style convs with stride 1 and "same" padding: the pixel tile of all k*k taps from ONE staged run of input rows ---------
// The 256 output pixels of a tile are consecutive in the flattened (image, y, x) order, and with stride 1 / pad = dil (k-1)/2 tap
// (ky, kx) of output pixel p reads input pixel p + (ky dil - pad) W + (kx dil - pad): the k*k pixel tiles are the same run of
// 256 + 2 (pad W + pad) input rows read at k*k row shifts. conv_glds_kernel fetches every shifted copy from L2 (9 x 32 KB of pixels
// beside 9 x 32 KB of weights per 64 channels); here the run is staged once per 32-channel slice (26 KB at W = 75) and the taps
// shift the fragment reads instead -- 40 % less L2 -> LDS traffic, which is what bounds that kernel. Rows that a tap must not see
// (image borders: the neighbour in the flattened order is another row's or image's pixel) are zeroed in the pixel fragment
// registers (lane = pixel) by a per-pixel tap mask.
//   K order: 32-channel slice outer, tap inner; a stage = two consecutive (slice, tap) half-stages = 64 deep. 9 stages = 18
//   half-stages = two slices = 64 input channels form the loop body, unrolled so that every buffer index, tap shift and DMA
//   piece is static. Pixel runs are double-buffered per slice parity, the weight stages double-buffered per stage.
// The weights are then most of the traffic, and their bytes per flop fall with the PIXEL tile only: a wave tile of TP pixel tiles x
// TC channel tiles (workgroup 64 TP pixels x 64 TC channels) is instantiated as 4 x 4 (256 x 256) and 8 x 2 (512 x 128).
constexpr int HK = 32;                      // K per half-stage
// capacity of a staged run in rows (2 runs + 2 x 2 weight half-stages + 1 KB <= 152 KB of LDS). Unused DMA slots still cost an issue
// each: 512 x 128 keeps the 832 rows its layers need (W <= 159; with 960 it spills and loses 8 %)
constexpr int halo_run_rows(int tp, int tc) { return tc == 4 ? 704 : tp == 8 ? 832 : 960; }

template <int KSZ, int TP, int TC, bool HEAD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_halo_kernel(PwArgs a) {
    constexpr int NT = KSZ * KSZ;           // taps; the unrolled body below is written for NT == 9
    static_assert(NT == 9, "3x3 only");
    constexpr int BPt = 64 * TP, BCt = 64 * TC;
    constexpr int A_ROWS = halo_run_rows(TP, TC);   // rows of a staged run (pixel tile + halo on both sides), upper bound
    constexpr int A_BUF = A_ROWS * HK;      // halfs
    constexpr int B_HALF = BCt * HK;
    constexpr int B_STAGE = 2 * B_HALF;
    constexpr int APW = (A_ROWS / 16 + 3) / 4;      // DMA pieces of a run per wave, APS per stage over a 3-stage window
    constexpr int APS = (APW + 2) / 3;
    constexpr int NM = TP * TC;             // MFMAs of a phase
    extern __shared__ __attribute__((aligned(16))) half_t lds_raw[];
    float* bsh = reinterpret_cast<float*>(lds_raw);
    half_t* lds = lds_raw + 2 * BCt;
    half_t* Abuf = lds;                     // [2][A_BUF]
    half_t* Bbuf = lds + 2 * A_BUF;         // [2][B_STAGE]; behind it 1 KB nobody reads: destination of the DMA slots a short run does not need

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 1, wc = wave & 1;
    const int r = lane & 31, hh = lane >> 5;
    // Tile of this workgroup. Workgroups are dispatched in linear order, workgroup b to XCD b % 8. An XCD walks a CONTIGUOUS range of
    // pixel tiles, the channel tiles of one pixel tile next to each other: neighbouring pixel tiles share their halo rows (W + 1 rows
    // on each side of 256: a third to a half of a run) and the channel tiles share the whole run in that XCD's L2. In the plain (x, y)
    // order neighbouring pixel tiles sit on different XCDs and the second channel tile of a pixel tile runs a full round later: every
    // XCD fetched its halos, and every channel tile its rows, from HBM again (PMC: 2.7 x the input bytes).
    int bx, by;
    {
        const int gx = gridDim.x, gy = gridDim.y, id = blockIdx.x + gx * blockIdx.y, G = gx & ~7;
        if (a.cv_plain_order) {
            bx = blockIdx.x;
            by = blockIdx.y;
        } else if (id < G * gy) {
            const int k = id >> 3;
            by = k % gy;
            bx = (id & 7) * (G >> 3) + k / gy;
        } else {
            const int rem = id - G * gy, wd = gx - G;
            bx = G + rem % wd;
            by = rem / wd;
        }
    }
    const int m0 = bx * BPt, n0 = by * BCt;
    const int M = a.m, K = a.cin, CIN = a.cv_cin;
    const int NC = HEAD ? a.cout + a.cout_b : a.cout;       // head kernel: the channels of a second head on the same input follow
    const int W = a.cv_w, pad = a.cv_pad, dil = a.cv_dil;
    const int HALO = pad * W + pad;
    const int NPIECE = (BPt + 2 * HALO + 15) >> 4;      // 16-row DMA pieces of a run (<= 4 APW)
    const int NIT = CIN / 64;

    floatx16 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- loader. A DMA piece = 16 rows of 64 B; lane -> row lane >> 2, position lane & 3, source chunk = position ^ ((row >> 2) & 3).
    // The DMA goes out from inline asm, behind the compiler's back: as a builtin, hipcc drains EVERY outstanding piece (s_waitcnt vmcnt(0))
    // in front of the stage barrier, and the youngest piece had been issued a few dozen cycles before it -- every stage waited one full L2
    // round trip with the matrix pipe idle (one wave per SIMD: nothing else runs). Here the wait in front of a stage's barrier is COUNTED:
    // it retires the weight pieces of the next stage (issued three and four phases earlier) and leaves the run pieces of this stage in
    // flight until the next stage's wait. Addresses are raw-buffer offsets (one VGPR per lane, the tap / slice part in an SGPR): rows of
    // the run outside the tensor are out of range, return 0 and need no clamp (the fragment masks zero them anyway).
    const int lrow = lane >> 2, lchunk = (lane & 3) ^ ((lane >> 4) & 3);
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)lds;
    const unsigned dummy_dst = lds0 + (2 * A_BUF + 2 * B_STAGE) * 2;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(a.x)), 0,
                                                                            (int)((unsigned)M * (unsigned)CIN * 2u), 0x00020000);
    const unsigned xvo = ((unsigned)(m0 - HALO + wave * 16 + lrow) * (unsigned)CIN + (unsigned)lchunk * 8u) * 2u;     // piece 0 of this wave
    const unsigned xstep = 128u * (unsigned)CIN;                                                                     // 64 rows on
    // weight rows beyond the last channel (last channel tile of a head): the last row again, its outputs are never stored
    const char* wlo = reinterpret_cast<const char*>(a.w);
    if (HEAD && a.w_b && reinterpret_cast<const char*>(a.w_b) < wlo) wlo = reinterpret_cast<const char*>(a.w_b);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wlo), 0, -1, 0x00020000);
    unsigned wvo[TC];
#pragma unroll
    for (int j = 0; j < TC; ++j) {
        const int row = min(n0 + wave * (BCt / 4) + j * 16 + lrow, NC - 1);
        const bool first = row < a.cout;
        const unsigned base = (unsigned)((first ? reinterpret_cast<const char*>(a.w) : reinterpret_cast<const char*>(a.w_b)) - wlo);
        wvo[j] = base + (unsigned)(first ? row : row - a.cout) * (unsigned)K * 2u + (unsigned)lchunk * 16u;
    }
    auto dma16 = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned voff, unsigned soff, unsigned dst) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" : : "v"(voff), "s"(rs), "s"(dst), "s"(soff) : "memory");
    };
    auto issue_a = [&](int ab, int slice, int i, bool live) {    // piece i of this wave, 32-channel slice `slice` -> run buffer ab
        // The DMA schedule is static (APW slots per wave and slice); a slot beyond the run, or a slice that does not exist, lands in the
        // dummy block instead of branching around the instruction.
        const int pc = wave + 4 * i;
        const bool real = live && pc < NPIECE;                   // wave-uniform
        // (a slot that is not needed asks for an offset beyond the buffer: no memory access, zeros into the dummy block)
        dma16(xrs, real ? xvo + (unsigned)i * xstep : 0xfffffff0u, (unsigned)slice * 64u,
              real ? lds0 + (unsigned)(ab * A_BUF + pc * 16 * HK) * 2u : dummy_dst);
    };
    auto issue_b = [&](int bb, int hf, int slice, int tap, int j) {      // 16 weight rows j of this wave's BCt / 4, half-stage (slice, tap)
        dma16(wrs, wvo[j], (unsigned)(tap * CIN + slice * 32) * 2u,
              lds0 + (unsigned)(2 * A_BUF + bb * B_STAGE + hf * B_HALF + (wave * (BCt / 4) + j * 16) * HK) * 2u);
    };

    if (tid < BCt) {
        const int n = n0 + tid;
        bsh[tid] = n < a.cout ? a.bias[n] : (HEAD && n < NC) ? a.bias_b[n - a.cout] : 0.f;
    }

    // ---- fragment geometry
    unsigned vm[TP];                        // tap-validity mask of this lane's TP pixels (pixel tiles jt)
#pragma unroll
    for (int jt = 0; jt < TP; ++jt) {
        const int m = m0 + (wp * TP + jt) * 32 + r;
        unsigned mk = 0;
        if (m < M) {
            const int rem = m % a.hw;
            const int oy = rem / W, ox = rem - oy * W;
            for (int ky = 0; ky < KSZ; ++ky)
                for (int kx = 0; kx < KSZ; ++kx) {
                    const int iy = oy + ky * dil - pad, ix = ox + kx * dil - pad;
                    if (iy >= 0 && iy < a.cv_h && ix >= 0 && ix < W) mk |= 1u << (ky * KSZ + kx);
                }
        }
        vm[jt] = mk;
    }
    const int kbw0 = ((wc * TC) * 32 + r) * HK + ((hh ^ ((r >> 2) & 3)) * 8);             // weight fragment, K step 0 / 1 of a half-stage
    const int kbw1 = ((wc * TC) * 32 + r) * HK + (((2 + hh) ^ ((r >> 2) & 3)) * 8);
    half8 xf0[TP], wf0[TC], xf1[TP], wf1[TC];
    // fragment addresses of half-stage (run buffer ab, tap, weight buffer bb / half hf), K step ks
    auto x_addr = [&](int ab, int tap, int ks) -> const half_t* {
        const int ky = tap / KSZ, kx = tap - ky * KSZ;
        const int rs = r + HALO + (ky * dil - pad) * W + (kx * dil - pad);             // run row of pixel tile 0, wave row 0
        const int sw = (rs >> 2) & 3;
        return Abuf + ab * A_BUF + (wp * (BPt / 2) + rs) * HK + (((2 * ks + hh) ^ sw) * 8);
    };
    auto w_addr = [&](int bb, int hf, int ks) -> const half_t* { return Bbuf + bb * B_STAGE + hf * B_HALF + (ks ? kbw1 : kbw0); };
    auto mask_frag = [&](half8& xf, int j, int tap) {
        const unsigned mk = 0u - ((vm[j] >> tap) & 1u);
        uint4 v = *reinterpret_cast<uint4*>(&xf);
        v.x &= mk; v.y &= mk; v.z &= mk; v.w &= mk;
        xf = *reinterpret_cast<half8*>(&v);
    };
    // One phase: the TP x TC MFMAs of fragments (xc, wc). Between them, in the source order (a scheduling barrier closes every MFMA slot):
    // from slot R0 on the TP + TC fragment reads of the NEXT phase (xn from xa, wn from wb), then the phase's DMA pieces (dma(k), ND of
    // them, one per slot), and in the last TP slots the tap masks of the pixel fragments just read. SYNC: slot R0 opens with the stage's
    // counted wait and barrier (what the reads of this phase need: the next stage's weights and, at a run switch, the next run).
    auto phase = [&](const half8 (&xc)[TP], const half8 (&wc)[TC], half8 (&xn)[TP], half8 (&wn)[TC], const half_t* xa, const half_t* wb,
                     int mtap, auto r0, auto nwait, int nd, auto&& dma) {
        constexpr int R0 = decltype(r0)::value, NW = decltype(nwait)::value;
        constexpr int D0 = R0 + TP + TC < NM ? R0 + TP + TC : NM - 1;
#pragma unroll
        for (int u = 0; u < NM; ++u) {
            if constexpr (NW >= 0) {
                if (u == R0) {
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(NW) : "memory");
                    __builtin_amdgcn_s_barrier();
                }
            }
            acc[u / TP][u % TP] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc[u / TP], xc[u % TP], acc[u / TP][u % TP], 0, 0, 0);
            const int q = u - R0;
            if (q >= 0 && q < TP) xn[q] = *reinterpret_cast<const half8*>(xa + q * 32 * HK);
            else if (q >= TP && q < TP + TC) wn[q - TP] = *reinterpret_cast<const half8*>(wb + (q - TP) * 32 * HK);
#pragma unroll
            for (int k = 0; k < nd; ++k)
                if ((D0 + k < NM ? D0 + k : NM - 1) == u) dma(k);
            if (u >= NM - TP) mask_frag(xn[u - (NM - TP)], u - (NM - TP), mtap);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using I_ = std::integral_constant<int, 0>;
    using NoSync = std::integral_constant<int, -1>;

    // ---- prologue: run of slice 0, weights of stage 0 and the first half of stage 1 (what P3 of a stage -1 would have requested)
#pragma unroll
    for (int i = 0; i < APW; ++i) issue_a(0, 0, i, true);
#pragma unroll
    for (int j = 0; j < TC; ++j) { issue_b(0, 0, 0, 0, j); issue_b(0, 1, 0, 1, j); }
#pragma unroll
    for (int j = 0; j < TC; ++j) issue_b(1, 0, 0, 2, j);
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(TC) : "memory");
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TP; ++j) { xf0[j] = *reinterpret_cast<const half8*>(x_addr(0, 0, 0) + j * 32 * HK); mask_frag(xf0[j], j, 0); }
#pragma unroll
    for (int i = 0; i < TC; ++i) wf0[i] = *reinterpret_cast<const half8*>(w_addr(0, 0, 0) + i * 32 * HK);

    // One loop iteration = slices 2 it (run buffer 0) and 2 it + 1 (run buffer 1) = 18 half-stages h = 0..17 = 9 stages.
    // Stage s: half-stages h0 = 2 s, h1 = 2 s + 1; phases P0 (h0, K step 0) P1 (h0, 1) P2 (h1, 0) P3 (h1, 1); every phase reads the
    // fragments of the next one. Weight buffers alternate per stage. DMA: P3 of stage s - 1 (behind its barrier: the buffer stage s - 1
    // used is free) requests the first half-stage of stage s + 1, P0 of stage s the second; P1 two..five pieces of a pixel run (stages
    // 0-2: slice 2 it + 1 -> run buffer 1, stages 5-7: slice 2 it + 2 -> run buffer 0; a run is complete one stage before its first read).
    // The wait in front of the barrier of stage s (P3) retires everything but the run pieces of its own P1.
    for (int it = 0; it < NIT; ++it) {
        const int sA = 2 * it, sB = 2 * it + 1;
        const int itn = min(it + 1, NIT - 1);
        auto stage = [&](auto sc) {
            constexpr int S = decltype(sc)::value;
            constexpr int H0 = 2 * S, H1 = 2 * S + 1;
            constexpr int T0 = H0 % NT, T1 = H1 % NT, AB0 = H0 / NT, AB1 = H1 / NT;
            constexpr int S1 = (S + 1) % 9, S2 = (S + 2) % 9;                // next stage, the one after
            constexpr int N1H0 = 2 * S1, N1H1 = 2 * S1 + 1, N2H0 = 2 * S2;
            const int bb = (it + S) & 1, nb = bb ^ 1;                        // weight buffer of this / the next stage (9 stages per iteration)
            const int it1 = (S + 1 >= 9) ? itn : it, it2 = (S + 2 >= 9) ? itn : it;  // (beyond the last stage: its own slices again, never read)
            constexpr int AQ = S <= 2 ? S : S >= 5 && S <= 7 ? S - 5 : -1;   // run pieces requested in this stage: AQ * APS ..
            constexpr int NA = AQ < 0 ? 0 : (APW - APS * AQ < 0 ? 0 : APW - APS * AQ > APS ? APS : APW - APS * AQ);
            // P0
            phase(xf0, wf0, xf1, wf1, x_addr(AB0, T0, 1), w_addr(bb, 0, 1), T0, I_{}, NoSync{}, TC,
                  [&](int j) { issue_b(nb, 1, 2 * it1 + N1H1 / NT, N1H1 % NT, j); });
            // P1
            phase(xf1, wf1, xf0, wf0, x_addr(AB1, T1, 0), w_addr(bb, 1, 0), T1, I_{}, NoSync{}, NA, [&](int u) {
                if constexpr (S <= 2) issue_a(1, sB, APS * AQ + u, true);
                else if constexpr (S >= 5 && S <= 7) issue_a(0, sA + 2, APS * AQ + u, it + 1 < NIT);
            });
            // P2
            phase(xf0, wf0, xf1, wf1, x_addr(AB1, T1, 1), w_addr(bb, 1, 1), T1, I_{}, NoSync{}, 0, [&](int) {});
            // P3: two MFMAs, the counted wait and the barrier, then the next stage's first fragments and the stage after's first weights
            phase(xf1, wf1, xf0, wf0, x_addr(N1H0 / NT, N1H0 % NT, 0), w_addr(nb, 0, 0), N1H0 % NT, std::integral_constant<int, (NM >= 16 ? 2 : 1)>{},
                  std::integral_constant<int, NA>{}, TC, [&](int j) { issue_b(bb, 0, 2 * it2 + N2H0 / NT, N2H0 % NT, j); });
        };
        stage(std::integral_constant<int, 0>{});
        stage(std::integral_constant<int, 1>{});
        stage(std::integral_constant<int, 2>{});
        stage(std::integral_constant<int, 3>{});
        stage(std::integral_constant<int, 4>{});
        stage(std::integral_constant<int, 5>{});
        stage(std::integral_constant<int, 6>{});
        stage(std::integral_constant<int, 7>{});
        stage(std::integral_constant<int, 8>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the requests of the stages that do not exist
    __syncthreads();
    if constexpr (HEAD) conv_epilogue_fp32<TP, TC>(acc, a, lds, bsh, m0, n0);
    else conv_epilogue<TP, TC>(acc, a, lds, bsh, m0, n0);
}

}  // namespace

namespace {

// Result of a 16 x 16 block (conv_patch_kernel, conv_patch_resident_kernel) -> LDS [256 pixels][64 + 8] halfs: bias, activation, fp16. The bias comes
// in registers and the activation is ONE uniform branch around all 64 values of the lane (round 5: per 4-value chunk the old form read its bias
// from LDS, waited, and walked the activation switch -- sixteen dependent LDS round trips per block, 1.6 us of a 7 us block, tools/probe_patch.py).
template <typename F>
__device__ __forceinline__ void patch_emit_t(const floatx16 (&acc)[2][2], const float4 (&bq)[2][4], half_t* ot, int wave, int r, int hh, F actf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int prow = (4 * wave + 2 * j) * 16 + r;           // pixel index inside the block
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = bq[i][g];
                half4 hv;
                hv[0] = (half_t)actf(acc[i][j][4 * g + 0] + bv.x); hv[1] = (half_t)actf(acc[i][j][4 * g + 1] + bv.y);
                hv[2] = (half_t)actf(acc[i][j][4 * g + 2] + bv.z); hv[3] = (half_t)actf(acc[i][j][4 * g + 3] + bv.w);
                *reinterpret_cast<half4*>(&ot[prow * 72 + i * 32 + 8 * g + 4 * hh]) = hv;
            }
    }
}
__device__ __forceinline__ void patch_emit(const floatx16 (&acc)[2][2], const float4 (&bq)[2][4], half_t* ot, int wave, int r, int hh, int act) {
    if (act == DN_ACT_RELU) patch_emit_t(acc, bq, ot, wave, r, hh, [](float v) { return dn_relu(v); });
    else if (act == DN_ACT_NONE) patch_emit_t(acc, bq, ot, wave, r, hh, [](float v) { return v; });
    else patch_emit_t(acc, bq, ot, wave, r, hh, [act](float v) { return dn_act(v, act); });
}

// ---- 64 input channels at full resolution (VGG conv1_2, conv2_1: 300^2 .. 512^2 maps) -----------------------------------------------
// The flattened run of conv_halo_kernel needs 2 W + 2 halo rows -- too many for W >= 300 -- and with cin = 64 the whole K range of a
// pixel is 128 B. Here a workgroup owns a 16 x 16 block of output pixels x 64 output channels: the 18 x 18 input patch (all 64
// channels, 41 KB) is staged ONCE, zero-filled outside the image, so no tap needs a mask; the nine taps are nine row shifts of the
// fragment reads inside the patch; only the weights stream: 18 stages per slice of (one tap, 32 input channels) = 4 KB, three buffers,
// requested two stages ahead with a counted wait. 53760 B of LDS and 144 registers per lane: THREE workgroups per CU (round 5; 58 KB
// and two before -- a workgroup lives 12.9 us for 1.9 us of matrix work: 2.5 us waiting for its patch, 2 + 0.9 us writing its result, and
// the tap phases of two workgroups overlap only 58 % of the time, tools/probe_patch.py). Staging by LDS-DMA with the row swizzles.
constexpr int PT = 16, PP = PT + 2;             // output block edge, patch edge
constexpr int PATCH_ROWS = PP * PP;             // 324 pixels (the last 8-row DMA piece moves 4 rows)
constexpr int PATCH_HALFS = PATCH_ROWS * 64;
constexpr int WST_HALFS = 64 * 32;              // one weight stage: 64 output channels x 32 input channels of one tap
constexpr int WST_BUFS = 3;                     // stages in LDS: the one being read and the two after it (requested two stages ahead; 9 deep with two
                                                // workgroups per CU measured slower: the bound is the L2 -> LDS traffic itself, not its latency)
constexpr int PATCH_LDS = (PATCH_HALFS + WST_BUFS * WST_HALFS) * 2;            // 53760 B = 42 allocation granules of 1280 B: THREE workgroups per CU
                                                                                // (measured, tools/lds_occ.hip: 53760 -> 3 resident, 54016 -> 2)

__global__ __launch_bounds__(256, 3) void conv_patch_kernel(PwArgs a, long long* __restrict__ stamps) {      // (three waves per SIMD: at most 168 registers)
    extern __shared__ __attribute__((aligned(16))) half_t lds_raw[];
    half_t* patch = lds_raw;                                    // [PATCH_ROWS][64], swizzled 16-B chunks
    half_t* wbuf = patch + PATCH_HALFS;                         // [WST_BUFS][64][32], swizzled 16-B chunks
    const float* bsh = nullptr;                                 // [64] bias: no room of its own -- parked in the weight buffer the last stage leaves free
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int H = a.cv_h, W = a.cv_w, NC = a.cout;
    const int tiles_x = (W + PT - 1) / PT;
    const int ty0 = (blockIdx.x / tiles_x) * PT, tx0 = (blockIdx.x % tiles_x) * PT;
    const int n0 = blockIdx.y * 64, img = blockIdx.z;
    const int CIN = a.cv_cin, NS = CIN / 64;                   // 64-channel slices: the patch is re-staged per slice, the accumulators stay
    const char* xbase = reinterpret_cast<const char*>(a.x) + (size_t)img * H * W * CIN * 2;
    const char* zeros = reinterpret_cast<const char*>(a.zeros);
    const char* wbase = reinterpret_cast<const char*>(a.w) + (size_t)n0 * 9 * CIN * 2;

    // Swizzles (round 5, after SQ_LDS_BANK_CONFLICT showed 49 % of this kernel's LDS cycles were conflicts): ds_read_b128 serves the lane groups
    // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32) from 64 banks = sixteen 16-B slots (MI355X_MICROARCH.md, LDS). A B fragment's group is 8 pixels
    // of one patch row and 8 of the next (columns {0-3, 12-15} | {4-11} or the other way round, shifted by kx): slot = 8 (column & 1) + position,
    // so the position is the chunk XOR (COLUMN >> 1) & 7 -- by column, not by pixel index, whose rows are 18 apart and collided.
    // patch DMA: an instruction moves 8 rows of 128 B; lane -> row lane >> 3, position lane & 7, source chunk = position ^ ((column >> 1) & 7)
    const int lrow = lane >> 3, lpos = lane & 7;
    // patch pieces: 41 per workgroup, wave w takes pieces w, w + 4, ...
    auto issue_patch = [&](int slice) {
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            const int pc = wave + 4 * i;                            // wave-uniform
            if (pc * 8 >= PATCH_ROWS) break;
            const int q = pc * 8 + lrow;
            const int py = q / PP, px = q - py * PP;
            const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int chunk = lpos ^ ((px >> 1) & 7);
            const char* p = ok ? xbase + ((size_t)iy * W + ix) * CIN * 2 + slice * 128 + chunk * 16 : zeros;
            if (q < PATCH_ROWS) __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(patch + pc * 8 * 64), 16, 0, 0);     // (the last piece: 4 rows)
        }
    };
    // weight stage st = (slice, tap, K half): 64 rows of 64 B = four instructions of 16 rows, ONE per wave (so every wave's vmcnt counts stages);
    // lane -> row lane >> 2, position lane & 3, source chunk = position ^ ((row >> 2) & 3) (64-B rows: slot = 4 (row & 3) + position, and the rows
    // of a read group with equal row & 3 are {0, 12, 20, 24} / {4, 8, 16, 28} + ..)
    const int wrow = wave * 16 + (lane >> 2), wchunk = (lane & 3) ^ ((wrow >> 2) & 3);
    const char* wlane = wbase + (size_t)wrow * 9 * CIN * 2 + wchunk * 16;
    auto issue_w = [&](int st, int b) {
        const int sl = st / 18, rem = st - sl * 18;
        const int tap = rem >> 1, kh = rem & 1;
        __builtin_amdgcn_global_load_lds((gptr_t)(wlane + ((size_t)tap * CIN + sl * 64 + kh * 32) * 2), (lptr_t)(wbuf + b * WST_HALFS + wave * 16 * 32), 16, 0, 0);
    };
    CP_STAMP(0);
    issue_patch(0);
#pragma unroll
    for (int q = 0; q + 1 < WST_BUFS; ++q) issue_w(q, q);
    const float bias_r = tid < 64 ? a.bias[n0 + tid] : 0.f;

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // wave w: block rows 4 w .. 4 w + 3; pixel tile j = rows 4 w + 2 j, + 1; lane -> (row r >> 4, column r & 15)
    int q0[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) q0[j] = (4 * wave + 2 * j + (r >> 4)) * PP + (r & 15);       // patch pixel of tap (0, 0)
    const int wsw = (r >> 2) & 3;
    const int xcol = r & 15;                                     // patch column of tap kx = 0

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    CP_STAMP(1);
    // 18 stages of 8 MFMAs per slice. A stage's weights were requested two stages ahead; the buffer requested now was read in the stage before
    // this one, which every wave left through the barrier in front of this stage.
    const int T = NS * 18;
    int b = 0;                                                   // buffer of stage st
    for (int st = 0; st < T; ++st) {
        const int b2 = b >= 1 ? b - 1 : WST_BUFS - 1;            // buffer of stage st - 1 = of stage st + WST_BUFS - 1
        if (st + WST_BUFS - 1 < T) issue_w(st + WST_BUFS - 1, b2);
        if (st + 1 == T) {                                       // (the buffer of stage T - 2: read by nobody any more, published by this stage's barrier)
            float* bw = reinterpret_cast<float*>(wbuf + b2 * WST_HALFS);
            if (tid < 64) bw[tid] = bias_r;
            bsh = bw;
        }
        const int rem = st % 18;
        const int tap = rem >> 1, kh = rem & 1;
        const int ky = tap / 3, kx = tap - ky * 3;
        const half_t* wb = wbuf + b * WST_HALFS;
#pragma unroll
        for (int ksl = 0; ksl < 2; ++ksl) {
            const int ks = kh * 2 + ksl;
            half8 xf[2], wf[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = q0[j] + ky * PP + kx;
                xf[j] = *reinterpret_cast<const half8*>(patch + q * 64 + (((2 * ks + hh) ^ (((xcol + kx) >> 1) & 7)) * 8));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) wf[i] = *reinterpret_cast<const half8*>(wb + (i * 32 + r) * 32 + (((2 * ksl + hh) ^ wsw) * 8));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
        if (rem == 17 && st + 1 < T) {
            // everyone is done with this slice's patch only after the barrier: restage it behind one
            __syncthreads();
            issue_patch(st / 18 + 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (st + WST_BUFS - 1 < T) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WST_BUFS - 2) : "memory");    // stage st + 1 has landed (this wave's part), the later ones may be in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        b = b + 1 < WST_BUFS ? b + 1 : 0;
    }
    CP_STAMP(2);
    // epilogue: block -> LDS [256 pixels][64 + 8] halfs over the patch, then 16-B chunks: a block row is 2 KB contiguous in NHWC
    half_t* ot = patch;
    {
        float4 bq[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) bq[i][g] = *reinterpret_cast<const float4*>(&bsh[i * 32 + 8 * g + 4 * hh]);
        patch_emit(acc, bq, ot, wave, r, hh, a.act);
    }
    __syncthreads();
    CP_STAMP(3);
    if (a.pool_out) {
        // fused MaxPool2d(kernel 2, stride 2) (ssd_vgg16.py:34-37 via torchvision vgg16 features): the 16 x 16 block pools to 8 x 8,
        // block origins are multiples of 16 and H, W are even, so every window lies inside one block. The conv output itself
        // never reaches HBM (it is 4x the pooled map: conv1_2 of ssd512 alone is 537 MB written + read back per 16 images).
        const int HP = H >> 1, WP = W >> 1;
        half_t* pout = a.pool_out + (size_t)img * HP * WP * NC;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c = tid + 256 * u;                        // 64 pooled pixels x 8 chunks
            const int pp = c >> 3, ch = c & 7;
            const int py = pp >> 3, pxx = pp & 7;
            const int oy = (ty0 >> 1) + py, ox = (tx0 >> 1) + pxx;
            if (oy < HP && ox < WP) {
                const int p00 = (2 * py) * PT + 2 * pxx;
                const half8 v0 = *reinterpret_cast<const half8*>(&ot[p00 * 72 + ch * 8]);
                const half8 v1 = *reinterpret_cast<const half8*>(&ot[(p00 + 1) * 72 + ch * 8]);
                const half8 v2 = *reinterpret_cast<const half8*>(&ot[(p00 + PT) * 72 + ch * 8]);
                const half8 v3 = *reinterpret_cast<const half8*>(&ot[(p00 + PT + 1) * 72 + ch * 8]);
                half8 mx;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const half_t m01 = v0[e] > v1[e] ? v0[e] : v1[e], m23 = v2[e] > v3[e] ? v2[e] : v3[e];
                    mx[e] = m01 > m23 ? m01 : m23;
                }
                *reinterpret_cast<half8*>(pout + ((size_t)oy * WP + ox) * NC + n0 + ch * 8) = mx;
            }
        }
        CP_STAMP(4);
        return;
    }
    half_t* outp = reinterpret_cast<half_t*>(a.out) + (size_t)img * H * W * NC;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int c = tid + 256 * u;                            // 16-B chunk: 256 pixels x 8 chunks
        const int px = c >> 3, ch = c & 7;
        const int oy = ty0 + (px >> 4), ox = tx0 + (px & 15);
        if (oy < H && ox < W)
            *reinterpret_cast<uint4*>(outp + ((size_t)oy * W + ox) * NC + n0 + ch * 8) = *reinterpret_cast<const uint4*>(&ot[px * 72 + ch * 8]);
    }
    CP_STAMP(4);
}

// ---- the same block with 64 input channels, weights RESIDENT (round 5; VGG conv1_2 and conv2_1) ---------------------------------------------
// What bounded conv_patch_kernel was never the matrix pipe (0.31) nor the LDS array (30 % busy, SQ_LDS_IDX_ACTIVE) but everything around the
// multiplication: every 16 x 16 block streamed its nine taps (72 KB) next to its 41 KB patch -- 3.7 GB of L2 -> LDS traffic for conv1_2 of ssd512 --
// behind one barrier per tap, waited for its patch, and wrote its result through LDS: a workgroup lived 12.9 us for 1.9 us of matrix work
// (tools/probe_patch.py). With cin = 64 the nine taps of 64 output channels are 72 KB: ONE workgroup per CU keeps them in LDS for its whole life
// and walks over blocks (persistent, 256 workgroups). Per block and wave: 36 steps of four MFMAs with no barrier and no wait on memory, and
// riding in the same instruction stream (a) the LDS-DMA requests of the NEXT block's patch into the other patch buffer (one piece per step:
// ~60 cycles of issue under 96 cycles of MFMA), (b) the PREVIOUS block's result straight from its accumulator copy to memory -- bias, ReLU, fp16,
// the 2 x 2 max-pool across lanes (v_permlane16_swap for the row pair, a quad DPP for the column pair) and 8-byte stores whose out-of-image lanes
// carry an out-of-range buffer offset -- no LDS staging, no branch, one barrier per block. The fragments of step s + 1 are requested behind the
// FIRST MFMA of step s (sched_group_barrier; the order the compiler picks on its own leaves a read 32 cycles of cover: 3.8 vs 2.5 us per block).
// Blocks are dealt out so that the 32 workgroups of an XCD work on 32 consecutive blocks (neighbours share their halo in that XCD's L2).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// maximum of two packed fp16 pairs that are known to be ordinary numbers (the results of a ReLU): the instruction itself -- through
// __builtin_elementwise_max every operand is canonicalised first (three v_pk_max_f16 for one)
__device__ __forceinline__ unsigned pk_max_f16(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int RW_HALFS = 9 * 64 * 64;           // resident weights [tap][64 rows][64], rows swizzled by (row >> 1) & 7
constexpr int RP_HALFS = 328 * 64;              // a patch buffer: 41 whole DMA pieces of 8 rows (rows 324.. are never read)
constexpr int RES_LDS = (RW_HALFS + 2 * RP_HALFS) * 2;             // 157696 B

template <bool POOL>
__global__ __launch_bounds__(256) void conv_patch_resident_kernel(PwArgs a, int tiles_x, int tiles_per_image, int ntiles, long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) half_t lds_raw[];
    half_t* wres = lds_raw;
    half_t* pbuf = wres + RW_HALFS;                             // [2][328][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int H = a.cv_h, W = a.cv_w, NC = a.cout;
    const int n0 = blockIdx.y * 64;
    // blocks of this workgroup: XCD x = blockIdx.x & 7 owns [x N / 8, (x + 1) N / 8), its workgroup `slot` of `per` takes every per-th of them
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
    const int lo = (int)((long)xcd * ntiles / 8), hi = (int)((long)(xcd + 1) * ntiles / 8);
    const int first = lo + slot;
    if (first >= hi) return;
    const int nmine = (hi - first + per - 1) / per;
    const char* zeros = reinterpret_cast<const char*>(a.zeros);
    const char* wbase = reinterpret_cast<const char*>(a.w) + (size_t)n0 * 9 * 64 * 2;
    const int lrow = lane >> 3, lpos = lane & 7;
    struct Blk { const char* xbase; int ty1, tx1, img, ty0, tx0; };
    auto blk = [&](int t) {
        const int img = t / tiles_per_image, tt = t - img * tiles_per_image;
        const int by = tt / tiles_x, bx = tt - by * tiles_x;
        return Blk{reinterpret_cast<const char*>(a.x) + (size_t)img * H * W * 128, by * PT - 1, bx * PT - 1, img, by * PT, bx * PT};
    };
    // patch piece i of this wave = rows 8 (wave + 4 i) .. of the patch; the lane's pixel (row, column) inside the patch is recomputed per request
    // (a few VALU operations in an MFMA gap; 22 registers otherwise)
    auto issue_piece = [&](const Blk& B, half_t* patch, int i) {
        const int q = min((wave + 4 * i) * 8 + lrow, PATCH_ROWS - 1);
        const int py = (q * 3641) >> 16;                            // q / 18 for q < 324
        const int px = q - py * PP;
        const int iy = B.ty1 + py, ix = B.tx1 + px;
        const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const int chunk = lpos ^ ((px >> 1) & 7);                  // (the column swizzle: see conv_patch_kernel)
        const char* p = ok ? B.xbase + (size_t)(unsigned)((iy * W + ix) * 128 + chunk * 16) : zeros;
        __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(patch + (wave + 4 * i) * 8 * 64), 16, 0, 0);
    };
    CP_STAMP(0);
    // the weights, once: 72 pieces of 8 rows x 128 B
#pragma unroll 1
    for (int pc = wave; pc < 72; pc += 4) {
        const int tap = pc >> 3, row = (pc & 7) * 8 + lrow;
        const int chunk = lpos ^ ((row >> 1) & 7);
        __builtin_amdgcn_global_load_lds((gptr_t)(wbase + (((size_t)row * 9 + tap) * 64) * 2 + chunk * 16), (lptr_t)(wres + (size_t)pc * 8 * 64), 16, 0, 0);
    }
    {
        const Blk B0 = blk(first);
#pragma unroll
        for (int i = 0; i < 10; ++i) issue_piece(B0, pbuf, i);
        if (wave == 0) issue_piece(B0, pbuf, 10);
    }
    float4 bq[2][4];                                             // this lane's bias values, for the workgroup's whole life
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[i][g] = *reinterpret_cast<const float4*>(a.bias + n0 + i * 32 + 8 * g + 4 * hh);
    // wave w: block rows 4 w .. 4 w + 3; pixel tile j = rows 4 w + 2 j, + 1; lane -> (row r >> 4, column r & 15)
    int xoff[2];                                                 // byte offset of the lane's patch pixel of tap (0, 0)
#pragma unroll
    for (int j = 0; j < 2; ++j) xoff[j] = ((4 * wave + 2 * j + (r >> 4)) * PP + (r & 15)) * 128;
    const int xcol = r & 15;
    int xsw[3];                                                  // chunk swizzle of the lane's column under tap column kx
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) xsw[kx] = ((xcol + kx) >> 1) & 7;
    const int woff = r * 128, wsw = (r >> 1) & 7;
    // the output as a buffer: a store whose offset lies beyond it is dropped (lanes outside the image, and the block "before the first")
    const int HO = POOL ? H >> 1 : H, WO = POOL ? W >> 1 : W;
    const unsigned out_bytes = (unsigned)(a.m / a.hw) * (unsigned)HO * (unsigned)WO * (unsigned)NC * 2u;
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(POOL ? (void*)a.pool_out : a.out, 0, (int)out_bytes, 0x00020000);
    const int vsel = (r & 1) + 2 * (r >> 4);                     // POOL: which of the four channel groups g this lane of a 2 x 2 window stores
    unsigned vmask[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        vmask[q] = vsel == q ? 0xFFFFFFFFu : 0u;
        asm volatile("" : "+v"(vmask[q]));                       // (opaque: keeps the AND / OR form below)
    }

    // Result of one block from its accumulators to memory (see the header), cut into 20 units that ride in 20 MFMA steps of the next block: unit
    // (c, g), c = 2 j + i < 4, g < 4: bias, ReLU, fp16 of four channels and (POOL) their 2 x 2 maximum, (!POOL) their store; unit (c, 4): (POOL) the
    // store. Every unit -- like a patch request -- comes in four PARTS, one behind each MFMA of its step: the wave issues in order, so what shall run
    // under an MFMA has to stand between it and the next one. `live` false: every offset out of range.
    unsigned pk[4][4][2];
    struct Rider { float t[4]; unsigned u[2]; unsigned off; int iy, ix, px; const char* p; };
    auto emit_part = [&](Rider& R, const floatx16 (&acc)[2][2], const Blk& B, bool live, auto cc, auto gc, auto pc) {
        constexpr int c = decltype(cc)::value, g = decltype(gc)::value, part = decltype(pc)::value;
        constexpr int j = c >> 1, i = c & 1;
        if constexpr (g < 4) {
            if constexpr (part == 0) {
                const float4 bv = bq[i][g];
                R.t[0] = dn_relu(acc[i][j][4 * g + 0] + bv.x); R.t[1] = dn_relu(acc[i][j][4 * g + 1] + bv.y);
                R.t[2] = dn_relu(acc[i][j][4 * g + 2] + bv.z); R.t[3] = dn_relu(acc[i][j][4 * g + 3] + bv.w);
            } else if constexpr (part == 1) {
                half2_t h0, h1;
                h0[0] = (half_t)R.t[0]; h0[1] = (half_t)R.t[1]; h1[0] = (half_t)R.t[2]; h1[1] = (half_t)R.t[3];
                pk[c][g][0] = __builtin_bit_cast(unsigned, h0); pk[c][g][1] = __builtin_bit_cast(unsigned, h1);
                if constexpr (POOL) {
                    // MaxPool2d(2, 2) (ssd_vgg16.py:34-37 via torchvision vgg16 features): the window = lanes {2 c, 2 c + 1} x {row 0, row 1 of the
                    // tile's row pair} = lanes l, l ^ 1, l ^ 16, l ^ 17 of the 32-lane half
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const auto sw = __builtin_amdgcn_permlane16_swap(pk[c][g][e], pk[c][g][e], false, false);   // {own rows 0 0 2 2, other rows 1 1 3 3}
                        R.u[e] = pk_max_f16(sw[0], sw[1]);
                    }
                } else {
                    const int oy = B.ty0 + 4 * wave + 2 * j + (r >> 4), ox = B.tx0 + xcol;
                    const bool ok = live && oy < HO && ox < WO;
                    R.off = ok ? (unsigned)(((B.img * HO + oy) * WO + ox) * NC + n0 + i * 32 + 8 * g + 4 * hh) * 2u : 0xFFFFFFF0u;
                }
            } else if constexpr (part == 2) {
                if constexpr (POOL) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const unsigned nb = (unsigned)__builtin_amdgcn_mov_dpp((int)R.u[e], 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
                        pk[c][g][e] = pk_max_f16(R.u[e], nb);
                    }
                }
            } else if constexpr (!POOL) {
                u32x2 v; v[0] = pk[c][g][0]; v[1] = pk[c][g][1];
                __builtin_amdgcn_raw_buffer_store_b64(v, ors, R.off, 0, 0);
            }
        } else if constexpr (POOL) {
            if constexpr (part == 0) {
                // the four lanes of a window hold the same maxima: lane `vsel` stores channel group g = vsel
                // (masks, not selects: the compiler turns a select chain over pk[c][0..3] into an indexed load from a scratch copy)
                R.u[0] = pk[c][0][0] & vmask[0]; R.u[1] = pk[c][0][1] & vmask[0];
#pragma unroll
                for (int q = 1; q < 4; ++q) { R.u[0] |= pk[c][q][0] & vmask[q]; R.u[1] |= pk[c][q][1] & vmask[q]; }
            } else if constexpr (part == 1) {
                const int oy = (B.ty0 >> 1) + 2 * wave + j, ox = (B.tx0 >> 1) + (xcol >> 1);
                const bool ok = live && oy < HO && ox < WO;
                R.off = ok ? (unsigned)(((B.img * HO + oy) * WO + ox) * NC + n0 + i * 32 + 8 * vsel + 4 * hh) * 2u : 0xFFFFFFF0u;
            } else if constexpr (part == 3) {
                u32x2 v; v[0] = R.u[0]; v[1] = R.u[1];
                __builtin_amdgcn_raw_buffer_store_b64(v, ors, R.off, 0, 0);
            }
        }
    };
    // a request of the next patch in the same four parts
    auto piece_part = [&](Rider& R, const Blk& B, half_t* patch, int i, auto pc) {
        constexpr int part = decltype(pc)::value;
        if constexpr (part == 0) {
            const int q = min((wave + 4 * i) * 8 + lrow, PATCH_ROWS - 1);
            const int py = (q * 3641) >> 16;                            // q / 18 for q < 324
            R.px = q - py * PP;
            R.iy = B.ty1 + py; R.ix = B.tx1 + R.px;
        } else if constexpr (part == 1) {
            const bool ok = (unsigned)R.iy < (unsigned)H && (unsigned)R.ix < (unsigned)W;
            const int chunk = lpos ^ ((R.px >> 1) & 7);
            R.off = ok ? (unsigned)((R.iy * W + R.ix) * 128 + chunk * 16) : 0xFFFFFFFFu;
        } else if constexpr (part == 2) {
            R.p = R.off != 0xFFFFFFFFu ? B.xbase + (size_t)R.off : zeros;
        } else {
            __builtin_amdgcn_global_load_lds((gptr_t)R.p, (lptr_t)(patch + (wave + 4 * i) * 8 * 64), 16, 0, 0);
        }
    };

    floatx16 prev[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) prev[i][j][e] = 0.f;
    Blk Bprev = blk(first);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    CP_STAMP(1);

    for (int k = 0; k < nmine; ++k) {
        const int t = first + k * per;
        const Blk Bcur = blk(t);
        const Blk Bnext = blk(k + 1 < nmine ? t + per : t);     // (no next block: this one again, into the buffer nobody reads)
        const char* pb = reinterpret_cast<const char*>(pbuf + (k & 1) * RP_HALFS);
        half_t* pnext = pbuf + ((k + 1) & 1) * RP_HALFS;
        const char* wb = reinterpret_cast<const char*>(wres);
        CP_STAMP_IF(k == 2, 3);
        if (wave == 0) issue_piece(Bnext, pnext, 10);            // the 41st piece (the other 40: one per wave in each of the first ten steps)
        floatx16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        half8 xf[2][2], wf[2][2];
        auto frags = [&](int st, int bf) {
            const int tap = st >> 2, ks = st & 3;
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                xf[bf][j] = *reinterpret_cast<const half8*>(pb + xoff[j] + (ky * PP + kx) * 128 + (((2 * ks + hh) ^ xsw[kx]) * 16));
#pragma unroll
            for (int i = 0; i < 2; ++i)
                wf[bf][i] = *reinterpret_cast<const half8*>(wb + (tap * 64 + i * 32) * 128 + woff + (((2 * ks + hh) ^ wsw) * 16));
        };
        const bool live = k > 0;
        frags(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        static_for<36>([&](auto stc) {
            constexpr int st = decltype(stc)::value;
            constexpr int bf = st & 1;
            Rider R;
            // the step's rider: one request of the next patch, or one unit of the previous block's result, a part behind each MFMA
            auto rider = [&](auto pc) {
                if constexpr (st < 10) piece_part(R, Bnext, pnext, st, pc);
                else if constexpr (st < 30) emit_part(R, prev, Bprev, live, std::integral_constant<int, (st - 10) / 5>{}, std::integral_constant<int, (st - 10) % 5>{}, pc);
            };
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[bf][0], xf[bf][0], acc[0][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (st + 1 < 36) frags(st + 1, bf ^ 1);            // 128 cycles of MFMA in front of their use
            rider(std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[bf][0], xf[bf][1], acc[0][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            rider(std::integral_constant<int, 1>{});
            __builtin_amdgcn_sched_barrier(0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[bf][1], xf[bf][0], acc[1][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            rider(std::integral_constant<int, 2>{});
            __builtin_amdgcn_sched_barrier(0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[bf][1], xf[bf][1], acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            rider(std::integral_constant<int, 3>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        CP_STAMP_IF(k == 2, 4);
        // the next block's patch has landed and everyone is done reading this block's patch. The wave's stores of the previous block's result
        // (4 with the pool, 16 without: every unit issues its store, out of range or not) are all YOUNGER than its patch requests (steps 0 - 9
        // against 10 - 29; loads and stores retire in order), so the count leaves them in flight: with vmcnt(0) -- or __syncthreads(), whose
        // release fence waits for vmcnt(0) too -- every block ended with a wait for the acknowledgement of stores issued 0.3 - 2 us earlier
        // (conv2_1, 16 stores per block: 164 -> 156 - 158 us per 16 images of 512 x 512; conv1_2 with its 4 pooled stores: level).
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(POOL ? 4 : 16) : "memory");
        __builtin_amdgcn_s_barrier();
        CP_STAMP_IF(k == 2, 5);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) prev[i][j] = acc[i][j];
        Bprev = Bcur;
    }
    static_for<20>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        Rider R;
        static_for<4>([&](auto pc) { emit_part(R, prev, Bprev, true, std::integral_constant<int, u / 5>{}, std::integral_constant<int, u % 5>{}, pc); });
    });
    CP_STAMP(2);
}

bool patch_resident_shape(const PwArgs& a) {
    const long out_bytes = (long)(a.m / a.hw) * (a.pool_out ? (a.cv_h >> 1) * (a.cv_w >> 1) : a.cv_h * a.cv_w) * a.cout * 2;
    return a.cv_cin == 64 && a.act == DN_ACT_RELU && out_bytes < 0xFFFFFFF0L && (long)a.cv_h * a.cv_w * 128 < (1L << 31) && dn_knob("DN_PATCH_RESIDENT", 1) != 0;
}

int launch_patch_resident(const PwArgs& a, hipStream_t s) {
    const int tiles_x = dn_cdiv(a.cv_w, PT), tiles = tiles_x * dn_cdiv(a.cv_h, PT);
    const long ntiles = (long)tiles * (a.m / a.hw);
    DN_REQUIRE(ntiles < (1L << 30), "conv_patch_resident_kernel: %ld blocks", ntiles);
    const int groups = a.cout / 64;
    int g = (256 / groups) & ~7;                                  // one workgroup per CU in all (256 CUs), a multiple of the 8 XCDs per channel group
    if (g < 8) g = 8;
    long long* st = patch_stamps_take((size_t)g * groups);
    if (a.pool_out) {
        DN_HIP_CHECK(dn_allow_big_lds(reinterpret_cast<const void*>(conv_patch_resident_kernel<true>), 160 * 1024));
        dn_note_kernel("conv_patch_resident_kernel<pool>");
        hipLaunchKernelGGL(conv_patch_resident_kernel<true>, dim3(g, groups), dim3(256), RES_LDS, s, a, tiles_x, tiles, (int)ntiles, st);
    } else {
        DN_HIP_CHECK(dn_allow_big_lds(reinterpret_cast<const void*>(conv_patch_resident_kernel<false>), 160 * 1024));
        dn_note_kernel("conv_patch_resident_kernel");
        hipLaunchKernelGGL(conv_patch_resident_kernel<false>, dim3(g, groups), dim3(256), RES_LDS, s, a, tiles_x, tiles, (int)ntiles, st);
    }
    return DN_OK;
}

int patch_max_cin() { return 128; }

bool patch_shape(const PwArgs& a) {
    const int on = 1;
    return on && a.zeros && !a.out_fp32 && !a.residual && !a.se && a.cv_k == 3 && a.cv_stride == 1 && a.cv_pad == 1 && a.cv_dil == 1 &&
           a.cv_ho == a.cv_h && a.cv_wo == a.cv_w && a.cv_cin % 64 == 0 && a.cv_cin <= patch_max_cin() && a.cout % 64 == 0 && a.m / a.hw <= 65535;
}

int launch_patch(const PwArgs& a, hipStream_t s) {
    if (patch_resident_shape(a)) return launch_patch_resident(a, s);
    const size_t lds = PATCH_LDS;
    DN_HIP_CHECK(dn_allow_big_lds(reinterpret_cast<const void*>(conv_patch_kernel)));
    dn_note_kernel("conv_patch_kernel");
    const int tiles = dn_cdiv(a.cv_w, PT) * dn_cdiv(a.cv_h, PT);
    hipLaunchKernelGGL(conv_patch_kernel, dim3(tiles, a.cout / 64, a.m / a.hw), dim3(256), lds, s, a, patch_stamps_take((size_t)tiles * (a.cout / 64) * (a.m / a.hw)));
    return DN_OK;
}
}  // namespace

namespace {
bool halo_shape(const PwArgs& a) {
    const int halo = dn_knob("DN_CONV_HALO", 1);
    return halo && a.zeros && a.cv_k == 3 && a.cv_stride == 1 && a.cv_pad == a.cv_dil && a.cv_ho == a.cv_h && a.cv_wo == a.cv_w &&
           a.cv_cin % 64 == 0 && (long)a.m * a.cv_cin * 2 < (1L << 31) && (long)(a.cout + a.cout_b) * a.cin * 2 < (1L << 32);
}
int halo_rows(const PwArgs& a) { return 2 * (a.cv_pad * a.cv_w + a.cv_pad); }

template <int TP, int TC, bool HEAD>
int launch_halo(const PwArgs& a, hipStream_t s, const char* name) {
    constexpr int BPt = 64 * TP, BCt = 64 * TC;
    const size_t st = (size_t)2 * halo_run_rows(TP, TC) * HK + (size_t)2 * 2 * BCt * HK + 512;      // runs, weight stages, dummy block
    const size_t ot = HEAD ? (size_t)2 * (BPt / 2) * (BCt + 4) : (size_t)BPt * (BCt + 8);            // epilogue tile, in halfs
    const size_t lds = (st > ot ? st : ot) * sizeof(half_t) + BCt * sizeof(float);
    DN_HIP_CHECK(dn_allow_big_lds(reinterpret_cast<const void*>(conv_halo_kernel<3, TP, TC, HEAD>)));
    dn_note_kernel(name);
    PwArgs b = a;
    b.cv_plain_order = 0;
    hipLaunchKernelGGL((conv_halo_kernel<3, TP, TC, HEAD>), dim3(dn_cdiv(a.m, BPt), dn_cdiv(a.cout + (HEAD ? a.cout_b : 0), BCt)), dim3(256), lds, s, b);
    return DN_OK;
}

// which tile a fp16-output 3x3 stride-1 conv runs on: 0 none, 1 = 256 x 256, 2 = 512 x 128, 3 = 256 x 128
int halo_variant(const PwArgs& a) {
    if (a.out_fp32 || a.residual || a.se || !halo_shape(a)) return 0;
    const int hr = halo_rows(a);
    const int force = 0;      // dev knob: prefer the 512 x 128 (2) / 256 x 128 (3) tile where it applies
    if (force == 2 && a.cout % 128 == 0 && a.cv_cin >= 128 && 512 + hr <= halo_run_rows(8, 2)) return 2;
    if (force == 3 && a.cout % 128 == 0 && a.cv_cin >= 128 && 256 + hr <= halo_run_rows(4, 2)) return 3;
    if (a.cout % 256 == 0 && 256 + hr <= halo_run_rows(4, 4)) return 1;
    if (a.cout % 128 == 0 && a.cv_cin >= 128 && 512 + hr <= halo_run_rows(8, 2)) return 2;
    if (a.cout % 128 == 0 && a.cv_cin >= 128 && 256 + hr <= halo_run_rows(4, 2)) return 3;      // (one 64-channel iteration: measured level with the 128x128 tile)
    // (a 256 x 64 tile for the 64-channel layer conv1_2: 330 vs 388 TFLOP/s for the 128 x 64 tile of pointwise.hip -- four MFMAs per K
    // step cannot cover the fragment masks and reads)
    return 0;
}
}  // namespace

bool conv_big_supported(const PwArgs& a) {
    if (halo_variant(a) || patch_shape(a)) return true;
    return a.zeros && a.cv_cin % GK == 0 && a.cout % BC == 0 && !a.out_fp32 && !a.residual && !a.se && a.cv_k * a.cv_k <= 32 && a.cin >= 2 * GK &&
           (long)a.m / a.hw * a.cv_h * a.cv_w * a.cv_cin * 2 < (1L << 31) && (long)a.cout * a.cin * 2 < (1L << 32);
}

int launch_conv_big(const PwArgs& a, hipStream_t s) {
    // 64 input channels: always the patch kernel; 128: where the run kernel would need its 256 x 128 tile (maps wider than 159; measured:
    // conv2_2 of ssd512 882 vs 568 TFLOP/s, while the 512 x 128 run tile of the 150-wide map keeps the faster step)
    if (patch_shape(a) && (a.cv_cin <= 64 || halo_variant(a) == 0 || halo_variant(a) == 3)) return launch_patch(a, s);
    switch (halo_variant(a)) {
        case 1: return launch_halo<4, 4, false>(a, s, "conv_halo_kernel<3,4,4>");
        case 2: return launch_halo<8, 2, false>(a, s, "conv_halo_kernel<3,8,2>");
        case 3: return launch_halo<4, 2, false>(a, s, "conv_halo_kernel<3,4,2>");
        default: break;
    }
    DN_REQUIRE(a.cout % BC == 0, "conv: cout=%d not a multiple of 256 on the 256x256 tile", a.cout);
    const size_t otile = (size_t)BP * OROW, st = (size_t)2 * GSTAGE;
    const size_t lds = (st > otile ? st : otile) * sizeof(half_t) + BC * sizeof(float);
    DN_HIP_CHECK(dn_allow_big_lds(reinterpret_cast<const void*>(conv_glds_kernel<false>)));
    dn_note_kernel("conv_glds_kernel");
    hipLaunchKernelGGL(conv_glds_kernel<false>, dim3(dn_cdiv(a.m, BP), a.cout / BC), dim3(256), lds, s, a);
    return DN_OK;
}

// Dense 3x3 heads with fp32 outputs (SSDHead, generalized_ssd.py:77-92) on the run-staged 256x256 tile; channel tiles beyond cout
// compute on the last weight row and are not stored.
bool conv_patch_pool_ok(int cin, int cout, int h, int w) {
    if (!dn_knob("DN_CONV_BIG", 1) || !dn_knob("DN_CONV_POOL", 1) || (h & 1) || (w & 1)) return false;
    static const half_t dummy_zero[8] = {};
    PwArgs a{};
    a.cv_k = 3; a.cv_stride = 1; a.cv_pad = 1; a.cv_dil = 1; a.cv_h = a.cv_ho = h; a.cv_w = a.cv_wo = w; a.cv_cin = cin;
    a.zeros = dummy_zero; a.residual = nullptr; a.se = nullptr; a.out_fp32 = 0;
    a.hw = h * w; a.m = a.hw; a.cin = 9 * cin; a.cout = cout;
    // the patch kernel is what launch_conv_big picks when the run-staged 512 x 128 tile does not apply; a larger batch can only
    // take the run-staged tiles away (their 2 GB addressing limit), never the patch kernel
    return patch_shape(a) && (cin <= 64 || halo_variant(a) == 0 || halo_variant(a) == 3) && (long)dn_cdiv(a.m, 256) * dn_cdiv(cout, 256) >= 40;
}

// ... or, for the layers of the run-staged 256 x 256 tile (conv3_3 of ssd512: 256 channels on 128 x 128), in the epilogue of conv_halo_kernel<3,4,4>:
// the tile must be whole row pairs of one image (256 % 2W == 0, H W % 256 == 0)
static bool conv_halo_pool_geometry(int cin, int cout, int h, int w) {
    if ((h & 1) || (w & 1)) return false;
    static const half_t dummy_zero[8] = {};
    PwArgs a{};
    a.cv_k = 3; a.cv_stride = 1; a.cv_pad = 1; a.cv_dil = 1; a.cv_h = a.cv_ho = h; a.cv_w = a.cv_wo = w; a.cv_cin = cin;
    a.zeros = dummy_zero; a.residual = nullptr; a.se = nullptr; a.out_fp32 = 0;
    a.hw = h * w; a.m = a.hw; a.cin = 9 * cin; a.cout = cout;
    return halo_variant(a) == 1 && !(patch_shape(a) && cin <= 64) && 256 % (2 * w) == 0 && (h * w) % 256 == 0 &&
           (long)dn_cdiv(a.m, 256) * dn_cdiv(cout, 256) >= 40;
}
// (the knobs decide at plan creation; the launch only checks the geometry, so a knob flipped between dn_create and dn_forward cannot strand a fused pair)
bool conv_halo_pool_ok(int cin, int cout, int h, int w) {
    return dn_knob("DN_CONV_BIG", 1) && dn_knob("DN_CONV_POOL", 1) && dn_knob("DN_CONV_HALO_POOL", 1) && conv_halo_pool_geometry(cin, cout, h, w);
}
bool conv_pool_ok(int cin, int cout, int h, int w) { return conv_patch_pool_ok(cin, cout, h, w) || conv_halo_pool_ok(cin, cout, h, w); }

int launch_conv_pool(const PwArgs& a, hipStream_t s) {
    DN_REQUIRE(a.pool_out && !(a.cv_h & 1) && !(a.cv_w & 1), "conv + max-pool: needs the pooled output and an even map");
    // by geometry alone: the patch kernel never writes the full-resolution map (a pair whose map has other readers is only fused when the run-staged
    // tile takes it: plan.hip), so a.out -- non-null for EVERY tensor with DN_WS_REUSE=0 -- must not steer the choice
    if (patch_shape(a) && (a.cv_cin <= 64 || !conv_halo_pool_geometry(a.cv_cin, a.cout, a.cv_h, a.cv_w))) return launch_patch(a, s);
    DN_REQUIRE(conv_halo_pool_geometry(a.cv_cin, a.cout, a.cv_h, a.cv_w), "conv + max-pool: geometry not supported by the patch kernel or the run-staged tile");
    // the run-staged tile addresses its input with 31-bit byte offsets: a batch beyond that goes in image ranges
    const int n = a.m / a.hw;
    const long per_img = (long)a.hw * a.cv_cin * 2;
    const int step = (int)std::max<long>(1, ((1L << 31) - 1) / per_img);
    for (int i0 = 0; i0 < n; i0 += step) {
        PwArgs b = a;
        const int cnt = std::min(step, n - i0);
        b.m = cnt * a.hw;
        b.x = a.x + (size_t)i0 * a.hw * a.cv_cin;
        b.pool_out = a.pool_out + (size_t)i0 * (a.hw / 4) * a.cout;
        if (a.out) b.out = reinterpret_cast<half_t*>(a.out) + (size_t)i0 * a.hw * a.cout;
        DN_REQUIRE(halo_variant(b) == 1, "conv + max-pool: the run-staged 256 x 256 tile does not take %d x %d x %d", a.cv_h, a.cv_w, a.cv_cin);
        const int rc = launch_halo<4, 4, false>(b, s, "conv_halo_kernel<3,4,4>");
        if (rc != DN_OK) return rc;
    }
    return DN_OK;
}

// Tile of a dense fp32 head launch: 0 none, 1 = 256 px x 256 ch, 2 = 512 x 128, 3 = 256 x 128. The class + box channels of a level (380 / 570
// for 91 classes) fill 74 % of two / three 256-channel tiles but 99 % / 89 % of three / five 128-channel tiles, and the 512 x 128 run tile
// runs level with the 256 x 256 one per FLOP (conv3 .. conv5 of ssd512_vgg16: 1140-1330 vs 1150-1320 TFLOP/s): the idle channels were a
// quarter of the head launches' time. A level too small for half the chip in 512-pixel tiles (the 16 x 16 level of ssd512 at batch 32: 80
// workgroups) takes 256 x 128 tiles -- twice the workgroups at half the work each.
static int head_variant(const PwArgs& a) {
    const int on = dn_knob("DN_CONV_HEAD_BIG", 1);
    const int minwg = 40;
    if (!on || !a.out_fp32 || a.residual || a.se || !halo_shape(a) || (a.cout & 1)) return 0;
    const int hr = halo_rows(a), nc = a.cout + a.cout_b;
    const int c256 = dn_cdiv(nc, 256), c128 = dn_cdiv(nc, 128);
    const long p256 = dn_cdiv(a.m, 256), p512 = dn_cdiv(a.m, 512);
    const int narrow = dn_knob("DN_CONV_HEAD_NARROW", 1);
    if (narrow && c128 * 128 < c256 * 256 && a.cv_cin >= 128) {
        if (512 + hr <= halo_run_rows(8, 2) && p512 * c128 >= 128) return 2;
        if (256 + hr <= halo_run_rows(4, 2) && p256 * c128 >= 2 * minwg) return 3;      // (40 workgroups of the 8 x 8 level: slower than the group launch)
    }
    const int tiles = dn_cdiv(a.cout, 256);
    if (256 + hr <= halo_run_rows(4, 4) && a.cout * 10 >= tiles * 256 * 6 && p256 * tiles >= minwg) return 1;      // at most 40 % of the channel tiles idle
    return 0;
}

bool conv_head_big_supported(const PwArgs& a) { return head_variant(a) != 0; }

int launch_conv_head_big(const PwArgs& a, hipStream_t s) {
    switch (head_variant(a)) {
        case 1: return launch_halo<4, 4, true>(a, s, "conv_halo_kernel<3,4,4,head>");
        case 2: return launch_halo<8, 2, true>(a, s, "conv_halo_kernel<3,8,2,head>");
        case 3: return launch_halo<4, 2, true>(a, s, "conv_halo_kernel<3,4,2,head>");
        default: break;
    }
    dn_set_error("dense head: not supported by the run-staged tiles");
    return DN_E_UNSUPPORTED;
}
