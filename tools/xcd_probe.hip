// dev probe (not part of the library): does a workgroup's XCD follow blockIdx % 8 consistently across the launches of a chain,
// and what does a consumer kernel gain when it reads what the SAME XCD's previous kernel wrote (L2 hit) instead of another XCD's?
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_probe.hip -o gpurun_out/xcd_probe && gpurun_out/xcd_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15; }

__global__ __launch_bounds__(256) void producer(uint4* buf, int per_block16, int G, int shift, unsigned salt, int* xcc) {
    const int region = (blockIdx.x + shift) % G;
    if (threadIdx.x == 0 && xcc) xcc[blockIdx.x] = xcc_id();
    uint4* p = buf + (size_t)region * per_block16;
    for (int i = threadIdx.x; i < per_block16; i += 256) p[i] = make_uint4(salt + i, region, 1, 2);
}

__global__ __launch_bounds__(256) void consumer(const uint4* buf, int per_block16, int G, int shift, unsigned* sink, int* xcc) {
    const int region = (blockIdx.x + shift) % G;
    if (threadIdx.x == 0 && xcc) xcc[blockIdx.x] = xcc_id();
    const uint4* p = buf + (size_t)region * per_block16;
    unsigned s = 0;
    for (int i0 = threadIdx.x; i0 < per_block16; i0 += 256 * 8) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[min(i0 + 256 * u, per_block16 - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (s == 0x12345u) sink[blockIdx.x] = s;
}

__global__ void empty_kernel(int* x) { if (x && threadIdx.x == 9999) x[0] = 1; }

int main() {
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const size_t BYTES = 256u << 20;
    uint4* buf; unsigned* sink; int* xcc;
    CK(hipMalloc(&buf, BYTES)); CK(hipMalloc(&sink, 1 << 20)); CK(hipMalloc(&xcc, 64 * 4096 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    // ---- 1. placement: XCC id of every block over a sequence of launches (grids that are / are not multiples of 8)
    for (int G : {512, 800, 1001, 200}) {
        const int L = 12;
        for (int l = 0; l < L; ++l) hipLaunchKernelGGL(producer, dim3(G), dim3(256), 0, s, buf, 16, G, 0, 1u, xcc + l * 4096);
        CK(hipStreamSynchronize(s));
        std::vector<int> h(L * 4096);
        CK(hipMemcpy(h.data(), xcc, h.size() * 4, hipMemcpyDeviceToHost));
        printf("G=%d:", G);
        for (int l = 0; l < L; ++l) {
            const int r0 = (h[l * 4096] + 8) % 8;
            int bad = 0;
            for (int b = 0; b < G; ++b) bad += h[l * 4096 + b] != (b + r0) % 8;
            printf(" [x0=%d bad=%d]", r0, bad);
        }
        printf("\n");
    }
    // the same inside a graph with two parallel branches (as the forward runs)
    {
        const int G = 512, L = 10;
        hipGraph_t g; hipGraphExec_t ge;
        hipEvent_t ef, ej; CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        CK(hipEventRecord(ef, s)); CK(hipStreamWaitEvent(s2, ef, 0));
        for (int l = 0; l < L; ++l) {
            hipLaunchKernelGGL(producer, dim3(G), dim3(256), 0, s, buf, 64, G, 0, 1u, xcc + l * 4096);
            hipLaunchKernelGGL(producer, dim3(G + 8 * (l % 3)), dim3(256), 0, s2, buf + (64 << 20) / 16, 64, G + 8 * (l % 3), 0, 1u, xcc + (L + l) * 4096);
        }
        CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s, ej, 0));
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
            std::vector<int> h(2 * L * 4096);
            CK(hipMemcpy(h.data(), xcc, h.size() * 4, hipMemcpyDeviceToHost));
            printf("graph 2 branches rep %d:", rep);
            for (int l = 0; l < 2 * L; ++l) {
                const int Gl = l < L ? G : G + 8 * ((l - L) % 3);
                const int r0 = h[l * 4096];
                int bad = 0;
                for (int b = 0; b < Gl; ++b) bad += h[l * 4096 + b] != (b + r0) % 8;
                printf(" [x0=%d bad=%d]", r0, bad);
            }
            printf("\n");
        }
    }

    // ---- 2. launch floor: chain of empty kernels in a graph
    for (int G : {1, 256, 2048}) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l < 100; ++l) hipLaunchKernelGGL(empty_kernel, dim3(G), dim3(256), 0, s, (int*)nullptr);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int rep = 0; rep < 10; ++rep) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty chain G=%d: %.2f us per dependent launch\n", G, ms * 1000.f / 1000.f);
    }

    // ---- 3. consumer time vs placement of its data: graph of 40 (producer, consumer) pairs
    for (int per_block_kb : {8, 16, 32, 64, 128}) {
        for (int G : {256, 512, 1024, 2048}) {
            const int per16 = per_block_kb * 1024 / 16;
            if ((size_t)G * per_block_kb * 1024 > BYTES) continue;
            float t[4];
            int shifts[4] = {0, 8, 1, 4};
            for (int v = 0; v < 4; ++v) {
                hipGraph_t g; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                for (int l = 0; l < 40; ++l) {
                    hipLaunchKernelGGL(producer, dim3(G), dim3(256), 0, s, buf, per16, G, 0, (unsigned)l, (int*)nullptr);
                    hipLaunchKernelGGL(consumer, dim3(G), dim3(256), 0, s, buf, per16, G, shifts[v], sink, (int*)nullptr);
                }
                CK(hipStreamEndCapture(s, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
                CK(hipEventRecord(e0, s));
                for (int rep = 0; rep < 5; ++rep) CK(hipGraphLaunch(ge, s));
                CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                t[v] = ms * 1000.f / 200.f;
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            }
            printf("pair(P+C) %4d KB/block x %4d blocks (%6.1f MB): same block %.2f us | same XCD other block %.2f | next XCD %.2f | +4 XCD %.2f\n",
                   per_block_kb, G, G * per_block_kb / 1024.0, t[0], t[1], t[2], t[3]);
        }
    }
    return 0;
}
