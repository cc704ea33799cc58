// Dense kxk convolution as an implicit GEMM with a 256x256 workgroup tile -- the MFMA-bound layers of the VGG models
// (ssd_vgg16.py:30-109: conv3_x .. fc7; 256..1024 output channels).
//
// Why a second kernel beside pointwise.hip's CONV path: there a wave owns a 64x64 tile and reads 1 KB of LDS per MFMA, i.e.
// 128 B/clk per CU at full matrix rate -- the LDS port saturates together with the matrix pipe (440-560 TFLOP/s measured).
// Here a wave owns 128 pixels x 128 channels (16 accumulator tiles = 256 registers): 512 B of LDS per MFMA, and each weight /
// pixel fragment feeds four MFMAs. One workgroup (4 waves, one per SIMD) per CU; with a single wave per SIMD nothing hides a
// wait but the wave's own MFMAs, so a K stage is laid out as straight-line code: all fragment reads of the stage first, the
// global loads two stages ahead, the LDS writes of the next stage between the two MFMA blocks.
//   K runs over (ky, kx, cin) in 32-deep stages (cin % 32 == 0: a stage lies inside one tap); the pixel tile of a stage is the
//   tap-shifted NHWC rows (zeros outside the image), the weight tile 256 rows of [cout][ky][kx][cin].
//   Operands are swapped as in pointwise.hip (A = weights, B = pixels): lane = pixel, 4 consecutive channels per register
//   group; the epilogue stages the tile in LDS and writes 16-byte row-contiguous chunks.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BP = 256, BC = 256, BK = 32;
constexpr int LROW = BK + 8;            // halfs per LDS row (80 B: odd number of 16-B slots)
constexpr int STAGE = (BP + BC) * LROW; // halfs per LDS stage buffer
constexpr int OROW = BC + 8;
#ifndef CB_SCHED
#define CB_SCHED 1
#endif

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_big_kernel(PwArgs a) {
    extern __shared__ __attribute__((aligned(16))) half_t lds_raw[];     // bias[BC] floats, then 2 stage buffers / the output tile
    float* bsh = reinterpret_cast<float*>(lds_raw);
    half_t* lds = lds_raw + 2 * BC;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave >> 1, wc = wave & 1;
    const int r = lane & 31, hh = lane >> 5;
    const int m0 = blockIdx.x * BP, n0 = blockIdx.y * BC;
    const int M = a.m, K = a.cin, NC = a.cout, CIN = a.cv_cin;
    const int KT = K / BK;

    floatx16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // staging geometry: thread -> 16-B chunk q of rows (tid >> 2) + 64 i, i < 4, of the pixel tile and of the weight tile
    const int q = tid & 3, row0 = tid >> 2;
    // addresses as uniform base + 32-bit per-lane byte offsets (saddr loads: one register per row instead of a 64-bit pointer)
    int xoff[4];                // tap (0,0) offset of the row's output pixel (may lie outside the image: see vmask)
    unsigned vmask[4];          // bit t: tap t of this pixel reads inside the image
    unsigned woff[4];
    const char* xbase = reinterpret_cast<const char*>(a.x);
    const char* zeros = reinterpret_cast<const char*>(a.zeros);
    const char* wbase = reinterpret_cast<const char*>(a.w) + (size_t)n0 * K * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = min(m0 + row0 + 64 * i, M - 1);           // rows beyond M: computed, never stored
        const int img = m / a.hw, rem = m - img * a.hw;
        const int oy = rem / a.cv_wo, ox = rem - oy * a.cv_wo;
        const int iy0 = oy * a.cv_stride - a.cv_pad, ix0 = ox * a.cv_stride - a.cv_pad;
        xoff[i] = (((img * a.cv_h + iy0) * a.cv_w + ix0) * CIN + q * 8) * 2;
        unsigned mk = 0;
        for (int ky = 0; ky < a.cv_k; ++ky)
            for (int kx = 0; kx < a.cv_k; ++kx) {
                const int iy = iy0 + ky * a.cv_dil, ix = ix0 + kx * a.cv_dil;
                if (iy >= 0 && iy < a.cv_h && ix >= 0 && ix < a.cv_w) mk |= 1u << (ky * a.cv_k + kx);
            }
        vmask[i] = mk;
        woff[i] = (unsigned)(((row0 + 64 * i) * K + q * 8) * 2);
    }

    // loader position (stage index -> tap, channel offset), advanced without divisions
    int ld_c0 = 0, ld_tap = 0, ld_ky = 0, ld_kx = 0, ld_k0 = 0;
    uint4 gx0, gx1, gx2, gx3, gw0, gw1, gw2, gw3;       // the stage in flight (named registers: arrays captured by the lambdas ended up in scratch)
    auto load_stage = [&]() {
        const int toff = (((ld_ky * a.cv_dil) * a.cv_w + ld_kx * a.cv_dil) * CIN + ld_c0) * 2;       // wave-uniform, bytes
        // unpredicated: a tap outside the image reads 16 zero bytes kept behind the weights (a select or mask on the loaded data
        // would make the wave wait for the load where the compiler places it)
        auto ldx = [&](int i) {
            const bool ok = (vmask[i] >> ld_tap) & 1u;
            const char* p = ok ? xbase + (long)(xoff[i] + toff) : zeros;
            return *reinterpret_cast<const uint4*>(p);
        };
        gx0 = ldx(0); gx1 = ldx(1); gx2 = ldx(2); gx3 = ldx(3);
        gw0 = *reinterpret_cast<const uint4*>(wbase + (woff[0] + (unsigned)ld_k0));
        gw1 = *reinterpret_cast<const uint4*>(wbase + (woff[1] + (unsigned)ld_k0));
        gw2 = *reinterpret_cast<const uint4*>(wbase + (woff[2] + (unsigned)ld_k0));
        gw3 = *reinterpret_cast<const uint4*>(wbase + (woff[3] + (unsigned)ld_k0));
        ld_k0 += BK * 2;
        ld_c0 += BK;
        const int wrap = ld_c0 == CIN;
        ld_c0 = wrap ? 0 : ld_c0;
        ld_tap += wrap;
        ld_kx += wrap;
        const int wrapx = ld_kx == a.cv_k;
        ld_kx = wrapx ? 0 : ld_kx;
        ld_ky += wrapx;
    };
    auto store_stage = [&](int b) {
        half_t* base = lds + b * STAGE + row0 * LROW + q * 8;
        *reinterpret_cast<uint4*>(base) = gx0;
        *reinterpret_cast<uint4*>(base + 64 * LROW) = gx1;
        *reinterpret_cast<uint4*>(base + 128 * LROW) = gx2;
        *reinterpret_cast<uint4*>(base + 192 * LROW) = gx3;
        *reinterpret_cast<uint4*>(base + (BP + 0) * LROW) = gw0;
        *reinterpret_cast<uint4*>(base + (BP + 64) * LROW) = gw1;
        *reinterpret_cast<uint4*>(base + (BP + 128) * LROW) = gw2;
        *reinterpret_cast<uint4*>(base + (BP + 192) * LROW) = gw3;
    };

    bsh[tid] = a.bias[n0 + tid];        // BC == 256 == threads; visible after the first barrier

    const half_t* xrow = lds + ((wp * 4) * 32 + r) * LROW + hh * 8;
    const half_t* wrow = lds + (BP + (wc * 4) * 32 + r) * LROW + hh * 8;
    half8 xf0[4], wf0[4], xf1[4], wf1[4];       // fragments of K step 0 / 1 of the current stage
    auto read_frags = [&](half8 (&xf)[4], half8 (&wf)[4], int b, int ks) {
#pragma unroll
        for (int j = 0; j < 4; ++j) xf[j] = *reinterpret_cast<const half8*>(xrow + b * STAGE + j * 32 * LROW + ks * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const half8*>(wrow + b * STAGE + i * 32 * LROW + ks * 16);
    };
    auto mfma16 = [&](const half8 (&xf)[4], const half8 (&wf)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    };

    // One barrier per stage, in the middle:
    //   block A: MFMAs of K step 0 | read the fragments of K step 1 | write stage kt+1 (in registers) into the other LDS buffer
    //   barrier (everyone is done reading this buffer and writing the other one)
    //   block B: MFMAs of K step 1 | read the fragments of K step 0 of stage kt+1 | request stage kt+2 from memory
    // so no LDS or memory wait is exposed between MFMAs except at the barrier itself.
    load_stage();
    store_stage(0);
    load_stage();
    __syncthreads();
    read_frags(xf0, wf0, 0, 0);

#define CB_INTERLEAVE_A()                                              \
    do {                                                               \
        for (int u_ = 0; u_ < 8; ++u_) {                               \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);         \
        }                                                              \
        for (int u_ = 0; u_ < 8; ++u_) {                               \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         \
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);         \
        }                                                              \
    } while (0)
#define CB_INTERLEAVE_B()                                              \
    do {                                                               \
        for (int u_ = 0; u_ < 8; ++u_) {                               \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);         \
        }                                                              \
        for (int u_ = 0; u_ < 8; ++u_) {                               \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);         \
        }                                                              \
    } while (0)

    constexpr bool sched = CB_SCHED;
    auto stage_full = [&](int b) {      // steady state: stages kt+1 and kt+2 exist
        read_frags(xf1, wf1, b, 1);
        store_stage(b ^ 1);
        mfma16(xf0, wf0);
        if (sched) CB_INTERLEAVE_A();
        __syncthreads();
        read_frags(xf0, wf0, b ^ 1, 0);
        load_stage();
        mfma16(xf1, wf1);
        if (sched) { CB_INTERLEAVE_B(); __builtin_amdgcn_sched_barrier(0); }
    };
    auto stage_tail = [&](int kt) {
        const int b = kt & 1;
        read_frags(xf1, wf1, b, 1);
        if (kt + 1 < KT) store_stage(b ^ 1);
        mfma16(xf0, wf0);
        __syncthreads();
        if (kt + 1 < KT) read_frags(xf0, wf0, b ^ 1, 0);
        if (kt + 2 < KT) load_stage();
        mfma16(xf1, wf1);
    };
    int kt = 0;
    for (; kt + 3 < KT; kt += 2) {
        stage_full(0);
        stage_full(1);
    }
    for (; kt < KT; ++kt) stage_tail(kt);
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
    return a.zeros && a.cv_cin % 32 == 0 && a.cout % 256 == 0 && !a.out_fp32 && !a.residual && !a.se && a.cv_k * a.cv_k <= 32 && a.cin >= 4 * BK &&
           (long)a.m / a.hw * a.cv_h * a.cv_w * a.cv_cin * 2 < (1L << 31) && (long)a.cout * a.cin * 2 < (1L << 32);
}

int launch_conv_big(const PwArgs& a, hipStream_t s) {
    const size_t stage_halfs = (size_t)2 * STAGE, otile = (size_t)BP * OROW;
    const size_t lds = (stage_halfs > otile ? stage_halfs : otile) * sizeof(half_t) + BC * sizeof(float);
    static bool attr = false;
    if (!attr) {
        DN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    dn_note_kernel("conv_big_kernel");
    hipLaunchKernelGGL(conv_big_kernel, dim3(dn_cdiv(a.m, BP), a.cout / BC), dim3(256), lds, s, a);
    return DN_OK;
}
