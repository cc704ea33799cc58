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
// Measured ablations of the loop (conv3_2 of ssd300_vgg16, TFLOP/s): as is 830; without the staging traffic 1330 -- the
// 256x256x64 stage moves 64 KB from L2 for 8.4 MFLOP (128 FLOP/B), so at 830 TFLOP/s the CUs pull 6.5 TB/s out of L2; the 9 taps
// re-read every input row from L2, which a spatially tiled input patch in LDS would remove (next step).
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BP = 256, BC = 256;
constexpr int OROW = BC + 8;

// An LDS-DMA wave-instruction writes 1 KB linearly (lane l -> base + 16 l): the stage image is rows of 128 B without padding, 8 rows
// per instruction, and the bank-conflict swizzle goes on the SOURCE: position p of row r holds K-chunk p ^ ((r >> 1) & 7); the
// fragment reads apply the same XOR (conflict-free for the 16-lane groups of ds_read_b128).
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr int GK = 64;                      // K per stage
constexpr int GSTAGE = (BP + BC) * GK;      // halfs per stage buffer (64 KB)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_glds_kernel(PwArgs a) {
    extern __shared__ __attribute__((aligned(16))) half_t lds_raw[];
    float* bsh = reinterpret_cast<float*>(lds_raw);
    half_t* lds = lds_raw + 2 * BC;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave >> 1, wc = wave & 1;
    const int r = lane & 31, hh = lane >> 5;
    const int m0 = blockIdx.x * BP, n0 = blockIdx.y * BC;
    const int M = a.m, K = a.cin, NC = a.cout, CIN = a.cv_cin;
    const int KT = K / GK;

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
    const char* wbase = reinterpret_cast<const char*>(a.w) + ((size_t)(n0 + wave * 64) * K) * 2;
    const unsigned woffE = (unsigned)((lrow * K + (lpos ^ ((lrow >> 1) & 3)) * 8) * 2);
    const unsigned woffO = (unsigned)((lrow * K + (lpos ^ (4 + ((lrow >> 1) & 3))) * 8) * 2);

    int ld_c0 = 0, ld_tap = 0, ld_ky = 0, ld_kx = 0, ld_k0 = 0;     // loader position, advanced without divisions
    // a stage = 16 LDS-DMA instructions per wave (8 pixel-row groups, 8 weight-row groups), issued in three parts so that they
    // can sit between the MFMAs of three K steps
    int st_toff = 0, st_tap = 0, st_k0 = 0;
    auto stage_begin = [&]() {
        st_toff = (((ld_ky * a.cv_dil) * a.cv_w + ld_kx * a.cv_dil) * CIN + ld_c0) * 2;
        st_tap = ld_tap;
        st_k0 = ld_k0;
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
        const bool ok = (vmask[j] >> st_tap) & 1u;
        const char* p = ok ? xbase + (long)(xoff[j] + st_toff) : zeros;
        __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(dst + j * 8 * GK), 16, 0, 0);
    };
    auto issue_w = [&](int b, int j) {
        half_t* dst = lds + b * GSTAGE + wave * 64 * GK;
        const char* p = wbase + (size_t)(j * 8) * K * 2 + (((j & 1) ? woffO : woffE) + (unsigned)st_k0);
        __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(dst + BP * GK + j * 8 * GK), 16, 0, 0);
    };
    auto issue_part = [&](int b, int part) {        // part 0: x0-5, part 1: x6-7 w0-3, part 2: w4-7 (6 + 6 + 4)
        if (part == 0) { issue_x(b, 0); issue_x(b, 1); issue_x(b, 2); issue_x(b, 3); issue_x(b, 4); issue_x(b, 5); }
        if (part == 1) { issue_x(b, 6); issue_x(b, 7); issue_w(b, 0); issue_w(b, 1); issue_w(b, 2); issue_w(b, 3); }
        if (part == 2) { issue_w(b, 4); issue_w(b, 5); issue_w(b, 6); issue_w(b, 7); }
    };

    bsh[tid] = a.bias[n0 + tid];

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

    // epilogue: bias + activation, tile -> LDS [BP][BC+8] halfs (over the stage buffers), then 16-B row-contiguous stores
    half_t* ot = lds;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int prow = (wp * 4 + j) * 32 + r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = (wc * 4 + i) * 32 + 8 * g + 4 * hh;
                const float4 bv = *reinterpret_cast<const float4*>(&bsh[cl]);
                half4 hv;
                hv[0] = (half_t)dn_act(acc[i][j][4 * g + 0] + bv.x, a.act);
                hv[1] = (half_t)dn_act(acc[i][j][4 * g + 1] + bv.y, a.act);
                hv[2] = (half_t)dn_act(acc[i][j][4 * g + 2] + bv.z, a.act);
                hv[3] = (half_t)dn_act(acc[i][j][4 * g + 3] + bv.w, a.act);
                *reinterpret_cast<half4*>(&ot[prow * OROW + cl]) = hv;
            }
        }
    }
    __syncthreads();
    half_t* outp = reinterpret_cast<half_t*>(a.out);
#pragma unroll 4
    for (int c = tid; c < BP * (BC / 8); c += 256) {
        const int row = c >> 5, ch = c & 31;
        const int m = m0 + row;
        if (m < M) *reinterpret_cast<uint4*>(outp + (size_t)m * NC + n0 + ch * 8) = *reinterpret_cast<const uint4*>(&ot[row * OROW + ch * 8]);
    }
}

}  // namespace

bool conv_big_supported(const PwArgs& a) {
    return a.zeros && a.cv_cin % GK == 0 && a.cout % BC == 0 && !a.out_fp32 && !a.residual && !a.se && a.cv_k * a.cv_k <= 32 &&
           a.cin >= 2 * GK && (long)a.m / a.hw * a.cv_h * a.cv_w * a.cv_cin * 2 < (1L << 31) && (long)a.cout * a.cin * 2 < (1L << 32);
}

int launch_conv_big(const PwArgs& a, hipStream_t s) {
    const size_t otile = (size_t)BP * OROW, st = (size_t)2 * GSTAGE;
    const size_t lds = (st > otile ? st : otile) * sizeof(half_t) + BC * sizeof(float);
    static bool attr = false;
    if (!attr) {
        DN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_glds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    dn_note_kernel("conv_glds_kernel");
    hipLaunchKernelGGL(conv_glds_kernel, dim3(dn_cdiv(a.m, BP), a.cout / BC), dim3(256), lds, s, a);
    return DN_OK;
}
