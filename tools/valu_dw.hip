// dev probe: the depthwise inner loop (one kernel row: TW = 4 outputs x K = 5 taps x 8 channels = 160 multiply-adds per lane) built from
// different instruction mixes, operands in registers only. Reports SIMD-cycles per row step per wave at 2.4 GHz, 4 waves per SIMD.
//   build: hipcc -O3 --offload-arch=gfx950 -o tools/valu_dw.bin tools/valu_dw.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 512;
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const uint4* in) {
    uint4 xv[8], wv[5];
    float acc[4][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) xv[i] = in[(threadIdx.x + i) & 63];
#pragma unroll
    for (int i = 0; i < 5; ++i) wv[i] = in[64 + ((threadIdx.x + i) & 63)];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
    for (int it = 0; it < ITER; ++it) {
        if constexpr (MODE == 0) {              // v_fma_mix_f32, as the kernels do today
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) {
                    const unsigned ee[4] = {xv[t + kx].x, xv[t + kx].y, xv[t + kx].z, xv[t + kx].w}, ww[4] = {wv[kx].x, wv[kx].y, wv[kx].z, wv[kx].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(acc[t][2 * i]) : "v"(ee[i]), "v"(ww[i]));
                        asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc[t][2 * i + 1]) : "v"(ee[i]), "v"(ww[i]));
                    }
                }
        }
        if constexpr (MODE == 1 || MODE == 2) {  // convert once per row, then fp32 FMAs (1: v_fma_f32, 2: v_pk_fma_f32)
            float xf[8][8], wf[5][8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned ee[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(xf[i][2 * j]) : "v"(ee[j]));
                    asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(xf[i][2 * j + 1]) : "v"(ee[j]));
                }
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const unsigned ee[4] = {wv[i].x, wv[i].y, wv[i].z, wv[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(wf[i][2 * j]) : "v"(ee[j]));
                    asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(wf[i][2 * j + 1]) : "v"(ee[j]));
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) {
                    if constexpr (MODE == 1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[t][e]) : "v"(xf[t + kx][e]), "v"(wf[kx][e]));
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            f2 c = {acc[t][e], acc[t][e + 1]}, x = {xf[t + kx][e], xf[t + kx][e + 1]}, w = {wf[kx][e], wf[kx][e + 1]};
                            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c) : "v"(x), "v"(w));
                            acc[t][e] = c.x; acc[t][e + 1] = c.y;
                        }
                    }
                }
        }
        if constexpr (MODE == 3) {               // weights pre-converted (fp32 in LDS/registers): only x is converted, v_pk_fma_f32
            float xf[8][8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned ee[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(xf[i][2 * j]) : "v"(ee[j]));
                    asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(xf[i][2 * j + 1]) : "v"(ee[j]));
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int kx = 0; kx < 5; ++kx)
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        f2 c = {acc[t][e], acc[t][e + 1]}, x = {xf[t + kx][e], xf[t + kx][e + 1]};
                        f2 w = {__uint_as_float((&wv[kx].x)[e / 2]), __uint_as_float((&wv[(kx + 1) % 5].x)[e / 2])};
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c) : "v"(x), "v"(w));
                        acc[t][e] = c.x; acc[t][e + 1] = c.y;
                    }
        }
        if constexpr (MODE == 4) {               // v_dot2_f32_f16 on tap pairs: 3 pairs stand for 5 taps -> 0.6 x the instructions
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const unsigned ee[4] = {xv[t + kx].x, xv[t + kx].y, xv[t + kx].z, xv[t + kx].w}, e2[4] = {xv[t + kx + 1].x, xv[t + kx + 1].y, xv[t + kx + 1].z, xv[t + kx + 1].w};
                    const unsigned ww[4] = {wv[kx].x, wv[kx].y, wv[kx].z, wv[kx].w}, w2[4] = {wv[kx + 1].x, wv[kx + 1].y, wv[kx + 1].z, wv[kx + 1].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(acc[t][2 * i]) : "v"(ee[i]), "v"(ww[i]));
                        asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(acc[t][2 * i + 1]) : "v"(e2[i]), "v"(w2[i]));
                    }
                }
        }
        if constexpr (MODE == 5) {
            // v_dot2_f32_f16 as it could really be used (round 5): a register holds two CHANNELS of one pixel, a dot2 needs two TAPS of one channel,
            // so pixel pairs (0,1) (2,3) (4,5) (6,7) are transposed with v_perm_b32 (2 per register pair: 32 per row step) and every output uses
            // two pairs + one single tap: even outputs pairs (kx 0,1) (kx 2,3) + tap 4, odd outputs tap 0 + pairs (kx 1,2) (kx 3,4).
            // 32 v_perm + 64 v_dot2 + 32 v_fma_mix = 128 instructions for the 160 multiply-adds (weights pre-paired: free).
            unsigned plo[4][4], phi[4][4];       // [pair][register]: (x_i.c_even, x_{i+1}.c_even), (x_i.c_odd, x_{i+1}.c_odd)
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                const unsigned a[4] = {xv[2 * pr].x, xv[2 * pr].y, xv[2 * pr].z, xv[2 * pr].w}, b[4] = {xv[2 * pr + 1].x, xv[2 * pr + 1].y, xv[2 * pr + 1].z, xv[2 * pr + 1].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(plo[pr][i]) : "v"(b[i]), "v"(a[i]), "s"(0x05040100u));
                    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(phi[pr][i]) : "v"(b[i]), "v"(a[i]), "s"(0x07060302u));
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int p0 = (t + 1) >> 1;            // first pair of this output: t = 0 -> pairs 0, 1; 1 -> 1, 2; 2 -> 1, 2; 3 -> 2, 3
                const int single = (t & 1) ? t : t + 4; // the unpaired tap's pixel
                const unsigned sx[4] = {xv[single].x, xv[single].y, xv[single].z, xv[single].w};
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const unsigned ww[4] = {wv[2 * q].x, wv[2 * q].y, wv[2 * q].z, wv[2 * q].w}, w2[4] = {wv[2 * q + 1].x, wv[2 * q + 1].y, wv[2 * q + 1].z, wv[2 * q + 1].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(acc[t][2 * i]) : "v"(plo[p0 + q][i]), "v"(ww[i]));
                        asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(acc[t][2 * i + 1]) : "v"(phi[p0 + q][i]), "v"(w2[i]));
                    }
                }
                const unsigned w4[4] = {wv[4].x, wv[4].y, wv[4].z, wv[4].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(acc[t][2 * i]) : "v"(sx[i]), "v"(w4[i]));
                    asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc[t][2 * i + 1]) : "v"(sx[i]), "v"(w4[i]));
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += acc[t][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
int run(const char* name, float* out, uint4* in, int w) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 256 * w;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, in);
    CHECK(hipEventRecord(e0, 0));
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, in);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 4;
    printf("%-52s %d waves/SIMD  %7.3f ms  %7.0f SIMD-cycles per row step (160 MACs/lane) per wave\n", name, w, ms, 1024.0 * ms * 1e-3 * 2.4e9 / ((double)grid * 4 * ITER));
    return 0;
}

int main() {
    float* out; uint4* in;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    CHECK(hipMalloc(&in, 128 * 16));
    std::vector<unsigned> h(512, 0x3c003c00u);
    CHECK(hipMemcpy(in, h.data(), 512 * 4, hipMemcpyHostToDevice));
    for (int w : {2, 4}) {
        if (run<0>("160 v_fma_mix_f32 (today)", out, in, w)) return 1;
        if (run<1>("104 v_cvt + 160 v_fma_f32", out, in, w)) return 1;
        if (run<2>("104 v_cvt + 80 v_pk_fma_f32", out, in, w)) return 1;
        if (run<3>("64 v_cvt (x only) + 80 v_pk_fma_f32 (fp32 weights)", out, in, w)) return 1;
        if (run<4>("96 v_dot2_f32_f16 (tap pairs, 3 for 5)", out, in, w)) return 1;
        if (run<5>("32 v_perm + 64 v_dot2 + 32 v_fma_mix (NHWC registers)", out, in, w)) return 1;
    }
    return 0;
}
