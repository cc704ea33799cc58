// dev lab: do independent vector instructions issue in the shadow of fp32 MFMAs (v_mfma_f32_32x32x2_f32, 16 passes)?
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_lab.hip -o tools/_mfma_valu_lab && tools/_mfma_valu_lab
// Every wave runs ITER x { MFMA (chain 0), NV vector ops, MFMA (chain 1), NV vector ops }; all instructions are asm volatile (order as written).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int NV, int KIND, bool MF>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    floatx16 c0, c1;
    for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; }
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    float x[8];
    for (int e = 0; e < 8; ++e) x[e] = a + e;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (MF) {
                if (h == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
            }
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[j & 7]) : "v"(b));
                else if (KIND == 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[j & 7]) : "v"(b));
                else asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x[j & 7]) : "v"(b));
            }
        }
    }
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e];
    for (int e = 0; e < 8; ++e) s += x[e];
    if (s == 12345.678f) out[0] = s;
}
// fp16 matrix instruction (8 passes) instead of the fp32 one
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
template <int NV>
__global__ __launch_bounds__(256) void kh(float* out, int iters) {
    floatx16 c0, c1;
    for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; }
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 1e-3f + e); b[e] = (_Float16)(1.0f + e); }
    float x[8];
    for (int e = 0; e < 8; ++e) x[e] = threadIdx.x + e;
    float bb = 1.0f + blockIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[j & 7]) : "v"(bb));
        }
    }
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e];
    for (int e = 0; e < 8; ++e) s += x[e];
    if (s == 12345.678f) out[0] = s;
}
// waves 0, 1 of a workgroup: matrix chain only; waves 2, 3: vector ops only. 512-thread workgroups: every SIMD holds one wave of each kind.
template <int NV, int WHICH>
__global__ __launch_bounds__(512) void ksplit(float* out, int iters) {
    floatx16 c0, c1;
    for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; }
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    float x[8];
    for (int e = 0; e < 8; ++e) x[e] = a + e;
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (WHICH & 1)
        for (int it = 0; it < iters; ++it) {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
        }
    } else {
        if (WHICH & 2)
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 2 * NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[j & 7]) : "v"(b));
        }
    }
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e];
    for (int e = 0; e < 8; ++e) s += x[e];
    if (s == 12345.678f) out[0] = s;
}
template <typename F>
void time_it(const char* name, F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(4000);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-60s %8.1f us  = %.1f ns per half-step\n", name, ms * 1e3, ms * 1e6 / 8000.0);
}
template <int NV, int KIND, bool MF>
void run(const char* name, int wg_per_cu, float* out) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, KIND, MF>), dim3(256 * wg_per_cu), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, KIND, MF>), dim3(256 * wg_per_cu), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: wg_per_cu waves, each 2 * iters half-steps
    printf("%-28s waves/SIMD %d  NV %2d  %8.1f us   %.1f ns per half-step per SIMD (MFMA alone = 64 cycles = 26.7 ns at 2.4 GHz)\n", name, wg_per_cu, NV, ms * 1e3,
           ms * 1e6 / (2.0 * iters * wg_per_cu));
}
int main() {
    float* out; hipMalloc(&out, 4);
    for (int w = 1; w <= 2; ++w) {
        run<0, 0, true>("mfma only", w, out);
        run<4, 0, true>("mfma + 4 fma", w, out);
        run<8, 0, true>("mfma + 8 fma", w, out);
        run<12, 0, true>("mfma + 12 fma", w, out);
        run<16, 0, true>("mfma + 16 fma", w, out);
        run<8, 0, false>("8 fma only", w, out);
        run<16, 0, false>("16 fma only", w, out);
        run<8, 1, true>("mfma + 8 cndmask", w, out);
        run<8, 1, false>("8 cndmask only", w, out);
        run<8, 2, true>("mfma + 8 cvt_pk", w, out);
        run<8, 2, false>("8 cvt_pk only", w, out);
    }
    float* o = out;
    time_it("fp16 32x32x16 only, 1 wave/SIMD", [&](int it) { hipLaunchKernelGGL((kh<0>), dim3(256), dim3(256), 0, 0, o, it); });
    time_it("fp16 32x32x16 + 4 fma, 1 wave/SIMD", [&](int it) { hipLaunchKernelGGL((kh<4>), dim3(256), dim3(256), 0, 0, o, it); });
    time_it("fp16 32x32x16 + 8 fma, 1 wave/SIMD", [&](int it) { hipLaunchKernelGGL((kh<8>), dim3(256), dim3(256), 0, 0, o, it); });
    time_it("fp16 32x32x16 + 8 fma, 2 waves/SIMD (per wave pair)", [&](int it) { hipLaunchKernelGGL((kh<8>), dim3(512), dim3(256), 0, 0, o, it); });
    time_it("split: matrix waves only (1 per SIMD)", [&](int it) { hipLaunchKernelGGL((ksplit<8, 1>), dim3(256), dim3(512), 0, 0, o, it); });
    time_it("split: vector waves only (8 fma per half-step)", [&](int it) { hipLaunchKernelGGL((ksplit<8, 2>), dim3(256), dim3(512), 0, 0, o, it); });
    time_it("split: both (one matrix + one vector wave per SIMD)", [&](int it) { hipLaunchKernelGGL((ksplit<8, 3>), dim3(256), dim3(512), 0, 0, o, it); });
    time_it("split: vector waves only (16 fma per half-step)", [&](int it) { hipLaunchKernelGGL((ksplit<16, 2>), dim3(256), dim3(512), 0, 0, o, it); });
    time_it("split: both (16 fma per half-step)", [&](int it) { hipLaunchKernelGGL((ksplit<16, 3>), dim3(256), dim3(512), 0, 0, o, it); });
    return 0;
}
