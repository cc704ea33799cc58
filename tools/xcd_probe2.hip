// dev probe: is the XCD of workgroup 0 the same for consecutive kernels of DIFFERENT shapes (threads, LDS, registers) in one
// graph / stream?   hipcc --offload-arch=gfx950 -O3 tools/xcd_probe2.hip -o /tmp/xcd_probe2 && /tmp/xcd_probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15; }

template <int T>
__global__ __launch_bounds__(T) void rec_kernel(int* xcc, int G, float* sink) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) xcc[blockIdx.x] = xcc_id();
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    if (lds[(threadIdx.x + 1) % T] == -1.f) sink[0] = 1.f;
}

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int* xcc; float* sink;
    const int L = 16;
    CK(hipMalloc(&xcc, L * 8192 * 4)); CK(hipMalloc(&sink, 64));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(rec_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(rec_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(rec_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int grids[L], kinds[L];
    for (int mode = 0; mode < 2; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        if (mode) CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l < L; ++l) {
            const int G = (l % 4 == 0) ? 800 : (l % 4 == 1) ? 3000 : (l % 4 == 2) ? 264 : 32;
            grids[l] = G; kinds[l] = l % 3;
            if (l % 3 == 0) hipLaunchKernelGGL(rec_kernel<256>, dim3(G), dim3(256), 20 * 1024, s, xcc + l * 8192, G, sink);
            else if (l % 3 == 1) hipLaunchKernelGGL(rec_kernel<512>, dim3(G), dim3(512), 70 * 1024, s, xcc + l * 8192, G, sink);
            else hipLaunchKernelGGL(rec_kernel<1024>, dim3(G), dim3(1024), 8 * 1024, s, xcc + l * 8192, G, sink);
        }
        if (mode) { CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0)); }
        for (int rep = 0; rep < (mode ? 3 : 1); ++rep) {
            if (mode) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            std::vector<int> h(L * 8192);
            CK(hipMemcpy(h.data(), xcc, h.size() * 4, hipMemcpyDeviceToHost));
            printf("%s rep %d:", mode ? "graph" : "eager", rep);
            for (int l = 0; l < L; ++l) {
                const int x0 = h[l * 8192];
                int bad = 0;
                for (int b = 0; b < grids[l]; ++b) bad += h[l * 8192 + b] != (b + x0) % 8;
                printf(" [k%d G=%d x0=%d bad=%d]", kinds[l], grids[l], x0, bad);
            }
            printf("\n");
        }
    }
    return 0;
}
