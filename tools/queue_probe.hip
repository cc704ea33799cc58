// dev probe: do the parallel branches of a hipGraph (sub-batch chains) pay the per-launch cost concurrently or one after another?
//   hipcc --offload-arch=gfx950 -O3 tools/queue_probe.hip -o /tmp/queue_probe && /tmp/queue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin_kernel(long long ticks, int* sink) {          // ticks of the 100 MHz clock
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
    if (sink && threadIdx.x == 9999) sink[0] = 1;
}

int main() {
    hipStream_t s[8];
    for (int i = 0; i < 8; ++i) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    hipEvent_t ef, ej[8], e0, e1;
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
    for (int i = 0; i < 8; ++i) CK(hipEventCreateWithFlags(&ej[i], hipEventDisableTiming));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int L = 60;
    for (int us10 : {0, 30, 80}) {                 // kernel body: 0, 3, 8 us of spinning
        for (int G : {1, 64, 512}) {
            for (int B : {1, 2, 3, 4, 6, 8}) {
                hipGraph_t g; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
                CK(hipEventRecord(ef, s[0]));
                for (int b = 1; b < B; ++b) CK(hipStreamWaitEvent(s[b], ef, 0));
                for (int l = 0; l < L; ++l)
                    for (int b = 0; b < B; ++b) hipLaunchKernelGGL(spin_kernel, dim3(G), dim3(256), 0, s[b], (long long)us10 * 10, (int*)nullptr);
                for (int b = 1; b < B; ++b) { CK(hipEventRecord(ej[b], s[b])); CK(hipStreamWaitEvent(s[0], ej[b], 0)); }
                CK(hipStreamEndCapture(s[0], &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CK(hipGraphLaunch(ge, s[0])); CK(hipStreamSynchronize(s[0]));
                CK(hipEventRecord(e0, s[0]));
                for (int rep = 0; rep < 10; ++rep) CK(hipGraphLaunch(ge, s[0]));
                CK(hipEventRecord(e1, s[0])); CK(hipStreamSynchronize(s[0]));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("body %.1f us, grid %4d, %d branches x %d kernels: %.1f us per graph = %.2f us per chain step (%.2f us per kernel overall)\n",
                       us10 / 10.0, G, B, L, ms * 100.f, ms * 100.f / L, ms * 100.f / (L * B));
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            }
        }
    }
    // ---- eager launches on B streams (no graph): does the per-launch cost overlap across queues?
    for (int us10 : {0, 30, 80}) {
        for (int B : {1, 2, 4}) {
            for (int rep = 0; rep < 2; ++rep) {
                for (int b = 0; b < B; ++b) CK(hipStreamSynchronize(s[b]));
                CK(hipEventRecord(e0, s[0]));
                for (int b = 1; b < B; ++b) { CK(hipEventRecord(ej[b], s[0])); CK(hipStreamWaitEvent(s[b], ej[b], 0)); }
                for (int l = 0; l < 10 * L; ++l)
                    for (int b = 0; b < B; ++b) hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s[b], (long long)us10 * 10, (int*)nullptr);
                for (int b = 1; b < B; ++b) { CK(hipEventRecord(ej[b], s[b])); CK(hipStreamWaitEvent(s[0], ej[b], 0)); }
                CK(hipEventRecord(e1, s[0])); CK(hipStreamSynchronize(s[0]));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) printf("eager body %.1f us, %d streams x %d kernels: %.2f us per chain step (%.2f us per kernel overall)\n", us10 / 10.0, B, 10 * L,
                                ms * 1000.f / (10 * L), ms * 1000.f / (10 * L * B));
            }
        }
    }
    // ---- single-branch graphs launched alternately on 2 streams (consecutive forwards on two execution contexts)
    for (int us10 : {0, 30, 80}) {
        hipGraph_t g; hipGraphExec_t ge[2];
        for (int q = 0; q < 2; ++q) {
            CK(hipStreamBeginCapture(s[q], hipStreamCaptureModeThreadLocal));
            for (int l = 0; l < L; ++l) hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s[q], (long long)us10 * 10, (int*)nullptr);
            CK(hipStreamEndCapture(s[q], &g));
            CK(hipGraphInstantiate(&ge[q], g, nullptr, nullptr, 0));
            CK(hipGraphDestroy(g));
        }
        for (int mode = 0; mode < 2; ++mode) {      // 0: both on one stream, 1: alternating streams
            for (int q = 0; q < 2; ++q) CK(hipStreamSynchronize(s[q]));
            CK(hipEventRecord(e0, s[0]));
            CK(hipEventRecord(ej[1], s[0])); CK(hipStreamWaitEvent(s[1], ej[1], 0));
            for (int rep = 0; rep < 20; ++rep) CK(hipGraphLaunch(ge[rep & 1], s[mode ? (rep & 1) : 0]));
            CK(hipEventRecord(ej[2], s[1])); CK(hipStreamWaitEvent(s[0], ej[2], 0));
            CK(hipEventRecord(e1, s[0])); CK(hipStreamSynchronize(s[0]));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("20 single-chain graphs of %d kernels (body %.1f us) %s: %.1f us per graph\n", L, us10 / 10.0, mode ? "alternating on 2 streams" : "on one stream", ms * 1000.f / 20);
        }
    }
    return 0;
}
