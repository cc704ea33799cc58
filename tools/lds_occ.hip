// dev: how many workgroups of 256 threads REALLY share a CU for a given dynamic LDS size on gfx950? Every workgroup bumps a per-CU counter (XCC_ID, HW_ID),
// spins ~30 us, records the maximum it saw, and leaves.   hipcc --offload-arch=gfx950 tools/lds_occ.hip -o tools/_lds_occ   (git-ignored; run it on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(256) void k(int* cnt, int* mx, float* o) {
    extern __shared__ float s[];
    s[threadIdx.x] = o[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const int cu = ((xcc & 15) << 8) | ((hw >> 8) & 255);      // cu_id, sh_id, se_id
        const int now = atomicAdd(&cnt[cu], 1) + 1;
        atomicMax(&mx[cu], now);
        const long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < 3000) { atomicMax(&mx[cu], __hip_atomic_load(&cnt[cu], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
        atomicAdd(&cnt[cu], -1);
    }
    __syncthreads();
    o[threadIdx.x] = s[255 - threadIdx.x];
}
int main() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int *cnt, *mx; float* o;
    (void)hipMalloc(&cnt, 4096 * 4); (void)hipMalloc(&mx, 4096 * 4); (void)hipMalloc(&o, 4096);
    (void)hipMemset(o, 0, 4096);
    for (int b : {32768, 40704, 40960, 41216, 45056, 49152, 51200, 52224, 53248, 53760, 54016, 54272, 54613, 58880, 65536, 80000, 81920}) {
        (void)hipMemset(cnt, 0, 4096 * 4); (void)hipMemset(mx, 0, 4096 * 4);
        hipLaunchKernelGGL(k, dim3(256 * 12), dim3(256), b, 0, cnt, mx, o);
        (void)hipDeviceSynchronize();
        std::vector<int> h(4096);
        (void)hipMemcpy(h.data(), mx, 4096 * 4, hipMemcpyDeviceToHost);
        int cus = 0, top = 0; long sum = 0;
        for (int v : h) if (v) { ++cus; top = std::max(top, v); sum += v; }
        int api = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, k, 256, b);
        printf("%6d B: %d CUs seen, max resident %d, mean of per-CU max %.2f (occupancy API says %d)\n", b, cus, top, cus ? (double)sum / cus : 0.0, api);
    }
    return 0;
}
