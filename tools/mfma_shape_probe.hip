// dev probe: sustained fp16 MFMA rate of the two dense shapes on random operands (the chip lowers its clock under matrix load, and the clock it
// holds depends on the shape: MI355X_MICROARCH.md, DVFS give-back (7)). One wave per SIMD (256 threads x 256 CUs x 4 rounds), operands in registers.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_shape_probe.hip -o /tmp/mfma_shape_probe && /tmp/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void probe(const half8* __restrict__ in, float* __restrict__ out, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(tid * 8 + i) & 65535]; b[i] = in[(tid * 8 + 4 + i) & 65535]; }
    float s = 0.f;
    if (SHAPE == 32) {
        floatx16 acc[4][4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    } else {
        floatx4 acc[8][8] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[j & 3], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    }
    out[tid] = s;
}

int main() {
    std::vector<_Float16> h(65536 * 8);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.1f);
    half8* din; float* dout;
    hipMalloc(&din, h.size() * 2); hipMalloc(&dout, 256 * 1024 * 4 * 4);
    hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int shape : {32, 16, 32, 16}) {
        const int iters = 4000, grid = 256 * 4;
        // flops per wave-iteration: 32-shape 16 x 32768, 16-shape 64 x 16384 -- both 1 MFLOP... (2*M*N*K)
        const double flops = (double)grid * 4 * iters * (shape == 32 ? 16.0 * 2 * 32 * 32 * 16 : 64.0 * 2 * 16 * 16 * 32);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int k = 0; k < 20; ++k) {
                if (shape == 32) hipLaunchKernelGGL(probe<32>, dim3(grid), dim3(256), 0, 0, din, dout, iters);
                else hipLaunchKernelGGL(probe<16>, dim3(grid), dim3(256), 0, 0, din, dout, iters);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("shape %dx%d: %.1f TFLOP/s (%.2f ms)\n", shape, shape, flops * 20 / (ms * 1e-3) / 1e12, ms);
        }
    }
    return 0;
}
