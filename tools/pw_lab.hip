// dev tool (round 6): stand-alone laboratory for the 1x1-conv launches -- ablations of the round-5 streaming kernel, store-pattern and
// load-pattern micro-benchmarks, and candidate schedules, timed the way the forward sees them: the input was just written by a producer launch
// (L2 / Infinity-Cache resident), the output goes to one of R rotating buffers (never cache-resident when it is written).
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/pw_lab.hip -o tools/_pw_lab
//   tools/_pw_lab M K N act [iters]          (prints one line per variant: mean us over iters launches, HIP events around each launch)
// Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel durations the review quotes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <string>
#include <algorithm>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned uint2v __attribute__((ext_vector_type(2)));

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_RELU6 = 2, ACT_HSWISH = 3 };
__device__ __forceinline__ float dn_relu(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 3.4028234663852886e38f); }
__device__ __forceinline__ float dn_relu6(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 6.f); }
__device__ __forceinline__ void act16(floatx16& v, int act) {
    if (act == ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = dn_relu(v[e]);
    } else if (act == ACT_RELU6) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = dn_relu6(v[e]);
    } else if (act == ACT_HSWISH) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = v[e] * dn_relu6(v[e] + 3.f) * (1.f / 6.f);
    }
}

struct Args {
    const half_t* x; const half_t* w; const half_t* wfb; const float* bias; half_t* out;
    int m, cin, cout, act;
    long long* stamps;
};

// ---------------------------------------------------------------------------------------------------------------------------------
// producer: writes x (what the previous layer's launch does in the forward)
__global__ void k_fill(half_t* x, size_t n8, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n8; i += stride) {
        half8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            unsigned h = (unsigned)(i * 8 + e) * 2654435761u + seed;
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            v[e] = (half_t)(((int)(h & 1023) - 512) * (1.f / 512.f));
        }
        reinterpret_cast<half8*>(x)[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// round-5 streaming kernel (pwdirect.hip pw_stream_kernel) with ablation bits: 1 = no weight loads, 2 = no stores, 4 = no x loads, 8 = no MFMA,
// 16 = 64-byte store pieces (permlane16 on top of permlane32)
template <int KSF, int PX, int ABL>
__global__ __launch_bounds__(256) void k_base(Args a, int tiles, int tiles_per_run) {
    constexpr int KSM = KSF > 0 ? KSF : 1;
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K = a.cin, NC = a.cout;
    const int flat = blockIdx.x;
    const int by = flat / tiles;
    const int m0 = (flat - by * tiles) * (128 * PX);
    const int mend = a.m;
    const int mrow0 = m0 + wave * (32 * PX);
    const int ctiles = (NC + 31) >> 5;
    const int ct0 = by * tiles_per_run, ct1 = min(ctiles, ct0 + tiles_per_run);
    if (mrow0 >= mend || ct0 >= ct1) return;
    const int kb = KSF * 16 + hh * 8;
    const int kcl = min(kb, K - 8) - hh * 8;
    const bool data = kb < K, bcol = kb == K;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    half8 wf[2][KSM], wl[2];
    float bl[2];
    auto request = [&](const int ct, const int buf) {
        const int nrow = min(ct * 32 + r, NC - 1);
        const half_t* wp = a.w + (size_t)nrow * K + hh * 8;
        if (ABL & 1) {
#pragma unroll
            for (int ks = 0; ks < KSF; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) wf[buf][ks][e] = (half_t)(float)((lane + ks + e + ct) & 7);
#pragma unroll
            for (int e = 0; e < 8; ++e) wl[buf][e] = (half_t)(float)((lane + e) & 3);
            bl[buf] = (float)(lane & 15);
        } else {
#pragma unroll
            for (int ks = 0; ks < KSF; ++ks) wf[buf][ks] = *reinterpret_cast<const half8*>(wp + ks * 16);
            wl[buf] = *reinterpret_cast<const half8*>(wp + kcl);
            bl[buf] = a.bias[nrow];
        }
    };
    half8 xf[PX][KSM], xl[PX];
    int row[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        row[j] = mrow0 + 32 * j + r;
        const half_t* xp = a.x + (size_t)min(row[j], mend - 1) * K + hh * 8;
        if (ABL & 4) {
#pragma unroll
            for (int ks = 0; ks < KSF; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) xf[j][ks][e] = (half_t)(float)((lane * 3 + ks + e + j) & 7);
#pragma unroll
            for (int e = 0; e < 8; ++e) xl[j][e] = (half_t)(float)((lane + e) & 3);
        } else {
#pragma unroll
            for (int ks = 0; ks < KSF; ++ks) xf[j][ks] = *reinterpret_cast<const half8*>(xp + ks * 16);
            xl[j] = *reinterpret_cast<const half8*>(xp + kcl);
        }
    }
    request(ct0, 0);
    {
        const half8 ones = {(half_t)1.f, (half_t)1.f, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < PX; ++j) xl[j] = data ? xl[j] : (bcol ? ones : zero8);
    }
    auto tile = [&](const int ct, const int buf) {
        half8 wlast;
        {
            const half_t hi = (half_t)bl[buf];
            const half_t lo = (half_t)(bl[buf] - (float)hi);
            const half8 bw = {hi, lo, 0, 0, 0, 0, 0, 0};
            wlast = data ? wl[buf] : (bcol ? bw : zero8);
        }
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            floatx16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            if (ABL & 8) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = (float)wf[buf][e % KSM][e & 7] + (float)xf[j][e % KSM][e & 7] + (float)wlast[e & 7] * (float)xl[j][e & 7];
            } else {
#pragma unroll
                for (int ks = 0; ks < KSF; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[buf][ks], xf[j][ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlast, xl[j], acc, 0, 0, 0);
            }
            act16(acc, a.act);
            uint2v p[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                half4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = (half_t)acc[4 * g + e];
                p[g] = __builtin_bit_cast(uint2v, hv);
            }
            const uint2v s0 = __builtin_amdgcn_permlane32_swap(p[0][0], p[1][0], false, false);
            const uint2v s1 = __builtin_amdgcn_permlane32_swap(p[0][1], p[1][1], false, false);
            const uint2v s2 = __builtin_amdgcn_permlane32_swap(p[2][0], p[3][0], false, false);
            const uint2v s3 = __builtin_amdgcn_permlane32_swap(p[2][1], p[3][1], false, false);
            uint4 lo4 = make_uint4(s0[0], s1[0], s0[1], s1[1]), hi4 = make_uint4(s2[0], s3[0], s2[1], s3[1]);
            const int nt = ct * 32;
            if (ABL & 16) {
                // rows 0..15 of the tile in lo4 (four lanes per row: 64 contiguous bytes), rows 16..31 in hi4
                const uint2v t0 = __builtin_amdgcn_permlane16_swap(lo4.x, hi4.x, false, false);
                const uint2v t1 = __builtin_amdgcn_permlane16_swap(lo4.y, hi4.y, false, false);
                const uint2v t2 = __builtin_amdgcn_permlane16_swap(lo4.z, hi4.z, false, false);
                const uint2v t3 = __builtin_amdgcn_permlane16_swap(lo4.w, hi4.w, false, false);
                lo4 = make_uint4(t0[0], t1[0], t2[0], t3[0]);
                hi4 = make_uint4(t0[1], t1[1], t2[1], t3[1]);
                const int rr = mrow0 + 32 * j + (lane & 15);
                const int col = nt + 8 * ((lane >> 5) + 2 * ((lane >> 4) & 1));
                half_t* o0 = a.out + (size_t)rr * NC + col;
                bool ok0 = rr < mend && col < NC, ok1 = rr + 16 < mend && col < NC;
                if (ABL & 2) { ok0 = ok0 && acc[0] == 12345.678f; ok1 = ok1 && acc[1] == 12345.678f; }
                if (ok0) *reinterpret_cast<uint4*>(o0) = lo4;
                if (ok1) *reinterpret_cast<uint4*>(o0 + (size_t)16 * NC) = hi4;
            } else {
                const int c0 = nt + hh * 8;
                half_t* orow = a.out + (size_t)row[j] * NC + hh * 8;
                bool ok = row[j] < mend;
                if (ABL & 2) ok = ok && acc[0] == 12345.678f;
                if (ok) {
                    if (c0 < NC) *reinterpret_cast<uint4*>(orow + nt) = lo4;
                    if (c0 + 16 < NC) *reinterpret_cast<uint4*>(orow + nt + 16) = hi4;
                }
            }
        }
    };
    for (int ct = ct0; ct < ct1; ct += 2) {
        request(min(ct + 1, ct1 - 1), 1);
        __builtin_amdgcn_sched_barrier(0);
        tile(ct, 0);
        if (ct + 1 < ct1) {
            request(min(ct + 2, ct1 - 1), 0);
            __builtin_amdgcn_sched_barrier(0);
            tile(ct + 1, 1);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// store-only patterns: the grid and per-wave instruction counts of k_base<.., 2>; every wave writes its 64 rows x run of channel tiles.
//   PAT 0: 32 rows x 32 B per instruction (round 5)   PAT 1: 16 rows x 64 B   PAT 2: 8 rows x 128 B   PAT 3: linear 1 KB per instruction
template <int PAT>
__global__ __launch_bounds__(256) void k_store(half_t* out, int m, int NC, int tiles, int tiles_per_run, unsigned v, long long* stamps) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    struct Stamp { long long* p; long long t0; __device__ ~Stamp() { const long long t2 = __builtin_amdgcn_s_memrealtime(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t3 = __builtin_amdgcn_s_memrealtime(); if ((threadIdx.x & 63) == 0) { p[0] = t0; p[1] = t0; p[2] = t2; p[3] = t3; } } }
        st{stamps + ((size_t)blockIdx.x * 4 + wave) * 4, (long long)__builtin_amdgcn_s_memrealtime()};
    const int flat = blockIdx.x;
    const int by = flat / tiles;
    const int m0 = (flat - by * tiles) * 256 + wave * 64;
    const int ctiles = (NC + 31) >> 5;
    const int ct0 = by * tiles_per_run, ct1 = min(ctiles, ct0 + tiles_per_run);
    if (m0 >= m || ct0 >= ct1) return;
    const uint4 val = make_uint4(v + lane, v, v ^ lane, v);
    if (PAT == 3) {
        // the same byte count as a linear sweep: wave w of the grid writes bytes [w * S, (w + 1) * S)
        const size_t total = (size_t)m * NC * 2;
        const size_t nw = (size_t)gridDim.x * 4;
        const size_t per = (total / nw) & ~(size_t)1023;
        unsigned char* p = reinterpret_cast<unsigned char*>(out) + ((size_t)flat * 4 + wave) * per + lane * 16;
        for (size_t o = 0; o < per; o += 1024) *reinterpret_cast<uint4*>(p + o) = val;
        return;
    }
    for (int ct = ct0; ct < ct1; ++ct) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rb = m0 + 32 * j;
            if (PAT == 0) {
                const int r = lane & 31, hh = lane >> 5;
                half_t* o = out + (size_t)(rb + r) * NC + ct * 32 + hh * 8;
                if (rb + r < m) { *reinterpret_cast<uint4*>(o) = val; *reinterpret_cast<uint4*>(o + 16) = val; }
            } else if (PAT == 1) {
                const int rr = rb + (lane & 15);
                half_t* o = out + (size_t)rr * NC + ct * 32 + 8 * ((lane >> 5) + 2 * ((lane >> 4) & 1));
                if (rr < m) *reinterpret_cast<uint4*>(o) = val;
                if (rr + 16 < m) *reinterpret_cast<uint4*>(o + (size_t)16 * NC) = val;
            } else {
                // two channel tiles at a time: 8 lanes per row = 128 B (ct must be even-aligned in pairs; odd tail tile handled as PAT 1)
                if ((ct - ct0) & 1) continue;
                if (ct + 1 < ct1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int rr = rb + 8 * q + (lane >> 3);
                        half_t* o = out + (size_t)rr * NC + ct * 32 + 8 * (lane & 7);
                        if (rr < m) *reinterpret_cast<uint4*>(o) = val;
                    }
                } else {
                    const int rr = rb + (lane & 15);
                    half_t* o = out + (size_t)rr * NC + ct * 32 + 8 * ((lane >> 5) + 2 * ((lane >> 4) & 1));
                    if (rr < m) *reinterpret_cast<uint4*>(o) = val;
                    if (rr + 16 < m) *reinterpret_cast<uint4*>(o + (size_t)16 * NC) = val;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// candidate 1: the run's weight tiles (fragment-major, bias step baked in: [tile][KS1][64 lanes][8]) staged ONCE per workgroup into LDS by
// LDS-DMA (1 KB contiguous per instruction) and read by all four waves with ds_read_b128; x fragments straight into registers as before;
// stores as 64-byte pieces. KS1 = K steps including the tail / bias step.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// FLAGS: 1 = per-wave stamps (s_memrealtime, 100 MHz) into a.stamps[wave][3], 2 = no stores
template <int KS1, int PX, int ST64, int FLAGS>
__global__ __launch_bounds__(256) void k_ldsw(Args a, int tiles, int tiles_per_run) {
    long long t0 = 0, t1 = 0;
    if (FLAGS & 1) t0 = __builtin_amdgcn_s_memrealtime();
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K = a.cin, NC = a.cout;
    const int flat = blockIdx.x;
    const int by = flat / tiles;
    const int m0 = (flat - by * tiles) * (128 * PX);
    const int mend = a.m;
    const int mrow0 = m0 + wave * (32 * PX);
    const int ctiles = (NC + 31) >> 5;
    const int ct0 = by * tiles_per_run, ct1 = min(ctiles, ct0 + tiles_per_run);
    const int nt_run = ct1 - ct0;
    // weights of the run -> LDS (all four waves take part, also waves without rows)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    {
        const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.wfb) + (size_t)ct0 * KS1 * 1024 + lane * 16;
        const int pieces = nt_run * KS1;
        for (int p = wave; p < pieces; p += 4) glds16(wsrc + (size_t)p * 1024, lds0 + p * 1024);
    }
    // x fragments
    constexpr int KSF = KS1 - 1;
    const int kb = KSF * 16 + hh * 8;
    const int kcl = min(kb, K - 8) - hh * 8;
    const bool data = kb < K, bcol = kb == K;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    half8 xf[PX][KS1];
    int row[PX];
    const bool live = mrow0 < mend;
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        row[j] = mrow0 + 32 * j + r;
        const half_t* xp = a.x + (size_t)min(row[j], mend - 1) * K + hh * 8;
#pragma unroll
        for (int ks = 0; ks < KSF; ++ks) xf[j][ks] = *reinterpret_cast<const half8*>(xp + ks * 16);
        xf[j][KSF] = *reinterpret_cast<const half8*>(xp + kcl);
    }
    {
        const half8 ones = {(half_t)1.f, (half_t)1.f, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < PX; ++j) xf[j][KSF] = data ? xf[j][KSF] : (bcol ? ones : zero8);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (FLAGS & 1) t1 = __builtin_amdgcn_s_memrealtime();
    if (!live) return;
    const half8* wl = reinterpret_cast<const half8*>(lds) + lane;
    for (int t = 0; t < nt_run; ++t) {
        half8 wf[KS1];
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) wf[ks] = wl[(t * KS1 + ks) * 64];
        const int nt = (ct0 + t) * 32;
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            floatx16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks], xf[j][ks], acc, 0, 0, 0);
            act16(acc, a.act);
            uint2v p[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                half4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = (half_t)acc[4 * g + e];
                p[g] = __builtin_bit_cast(uint2v, hv);
            }
            const uint2v s0 = __builtin_amdgcn_permlane32_swap(p[0][0], p[1][0], false, false);
            const uint2v s1 = __builtin_amdgcn_permlane32_swap(p[0][1], p[1][1], false, false);
            const uint2v s2 = __builtin_amdgcn_permlane32_swap(p[2][0], p[3][0], false, false);
            const uint2v s3 = __builtin_amdgcn_permlane32_swap(p[2][1], p[3][1], false, false);
            uint4 lo4 = make_uint4(s0[0], s1[0], s0[1], s1[1]), hi4 = make_uint4(s2[0], s3[0], s2[1], s3[1]);
            if (ST64) {
                const uint2v t0 = __builtin_amdgcn_permlane16_swap(lo4.x, hi4.x, false, false);
                const uint2v t1 = __builtin_amdgcn_permlane16_swap(lo4.y, hi4.y, false, false);
                const uint2v t2 = __builtin_amdgcn_permlane16_swap(lo4.z, hi4.z, false, false);
                const uint2v t3 = __builtin_amdgcn_permlane16_swap(lo4.w, hi4.w, false, false);
                lo4 = make_uint4(t0[0], t1[0], t2[0], t3[0]);
                hi4 = make_uint4(t0[1], t1[1], t2[1], t3[1]);
                const int rr = mrow0 + 32 * j + (lane & 15);
                const int col = nt + 8 * ((lane >> 5) + 2 * ((lane >> 4) & 1));
                half_t* o0 = a.out + (size_t)rr * NC + col;
                bool ok0 = rr < mend && col < NC, ok1 = rr + 16 < mend && col < NC;
                if (FLAGS & 2) { ok0 = ok0 && acc[0] == 12345.678f; ok1 = ok1 && acc[1] == 12345.678f; }
                if (ok0) *reinterpret_cast<uint4*>(o0) = lo4;
                if (ok1) *reinterpret_cast<uint4*>(o0 + (size_t)16 * NC) = hi4;
            } else {
                const int c0 = nt + hh * 8;
                half_t* orow = a.out + (size_t)row[j] * NC + hh * 8;
                if (row[j] < mend) {
                    if (c0 < NC) *reinterpret_cast<uint4*>(orow + nt) = lo4;
                    if (c0 + 16 < NC) *reinterpret_cast<uint4*>(orow + nt + 16) = hi4;
                }
            }
        }
    }
    if (FLAGS & 1) {
        const long long t2 = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t3 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) { long long* sp = a.stamps + ((size_t)blockIdx.x * 4 + wave) * 4; sp[0] = t0; sp[1] = t1; sp[2] = t2; sp[3] = t3; }
    }
}

__global__ __launch_bounds__(256) void k_null(Args a, int tiles, int tiles_per_run) {
    if (a.m < 0) a.out[threadIdx.x] = (half_t)0.f;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// candidate 2 ("pws"): persistent waves, weights of the wave's channel run STATIONARY IN REGISTERS (RT tiles x KS1 fragments, staged once per
// workgroup through LDS), x tiles of PXU x 32 rows fetched as CONTIGUOUS bytes by LDS-DMA (whole cache lines; the row-strided fragment
// loads of the round-5 kernels touch 32 lines per instruction and thrash L1), fragments read back with ds_read_b128, next tile's DMA in
// flight during the matrix / epilogue work, 64-byte store pieces through buffer stores (out-of-range offset instead of a branch).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int ACT>
__device__ __forceinline__ void act16t(floatx16& v) {
    if (ACT == ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = dn_relu(v[e]);
    } else if (ACT == ACT_RELU6) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = dn_relu6(v[e]);
    } else if (ACT == ACT_HSWISH) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = v[e] * dn_relu6(v[e] + 3.f) * (1.f / 6.f);
    }
}

template <int KS1, int RT, int PXU, int ACT, int FLAGS>
__global__ __launch_bounds__(256) void k_pws(Args a, int runs, int n_units) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int KSF = KS1 - 1;
    long long t0 = 0, t1 = 0, t2 = 0, ta = 0, tb = 0, tc = 0, ca = 0, cc = 0;
    if (FLAGS & 1) t0 = __builtin_amdgcn_s_memrealtime();
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K = a.cin, NC = a.cout;
    const int run = blockIdx.x % runs, wgq = blockIdx.x / runs;
    const int Q = (gridDim.x / runs) * 4;
    const int ctiles = (NC + 31) >> 5;
    const int ct0 = run * RT;
    const int xpieces = (PXU * K + 15) >> 4;                         // 1 KB LDS-DMA pieces of one x unit (PXU * 32 rows * 2K bytes)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned xb = lds0 + RT * KS1 * 1024 + wave * xpieces * 1024;      // this wave's x buffer
    const unsigned char* xsrc = reinterpret_cast<const unsigned char*>(a.x);
    const unsigned xbytes = (unsigned)a.m * (unsigned)K * 2u;         // (the lab's tensors are < 4 GB)
    const unsigned unit_bytes = PXU * 64u * (unsigned)K;
    auto dma_x = [&](int u) {
        const unsigned base = (unsigned)u * unit_bytes + lane * 16u;
        for (int p = 0; p < xpieces; ++p) glds16(xsrc + min(base + p * 1024u, xbytes - 16u), xb + p * 1024);
    };
    int u = wgq * 4 + wave;
    if (u < n_units) dma_x(u);
    half8 wf[RT][KS1];
    if (FLAGS & 8) {
        const half8* wg = reinterpret_cast<const half8*>(a.wfb) + (size_t)ct0 * KS1 * 64 + lane;
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) wf[t][ks] = wg[(min(t, ctiles - ct0 - 1) * KS1 + ks) * 64];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.wfb) + (size_t)ct0 * KS1 * 1024 + lane * 16;
        const int pieces = min(RT, ctiles - ct0) * KS1;
        for (int p = wave; p < pieces; p += 4) glds16(wsrc + (size_t)p * 1024, lds0 + p * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const half8* wl = reinterpret_cast<const half8*>(lds) + lane;
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) wf[t][ks] = wl[(t * KS1 + ks) * 64];
    }
    if (FLAGS & 1) t1 = __builtin_amdgcn_s_memrealtime();
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)((unsigned)a.m * (unsigned)NC * 2u), 0x00020000);
    const int kb = KSF * 16 + hh * 8;
    const bool data = kb < K, bcol = kb == K;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const half8 ones = {(half_t)1.f, (half_t)1.f, 0, 0, 0, 0, 0, 0};
    const unsigned xrd = (unsigned)(RT * KS1 * 1024 + wave * xpieces * 1024) + (unsigned)r * (unsigned)K * 2u + hh * 16u;      // byte offset of this lane's row in its x buffer
    const int colq = 8 * ((lane >> 5) + 2 * ((lane >> 4) & 1));
    for (; u < n_units; u += Q) {
        half8 xf[PXU][KS1];
#pragma unroll
        for (int j = 0; j < PXU; ++j)
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const unsigned o = xrd + (unsigned)j * 64u * (unsigned)K + (ks < KSF ? ks * 32u : (unsigned)(min(kb, K - 8) - hh * 8) * 2u);
                xf[j][ks] = *reinterpret_cast<const half8*>(lds + o);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if ((FLAGS & 1) && ta == 0) { ta = __builtin_amdgcn_s_memrealtime(); ca = __builtin_amdgcn_s_memtime(); }
        if (u + Q < n_units) dma_x(u + Q);
        if ((FLAGS & 1) && tb == 0) tb = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int j = 0; j < PXU; ++j) xf[j][KSF] = data ? xf[j][KSF] : (bcol ? ones : zero8);
        const int row0 = (u * PXU) * 32;
#pragma unroll
        for (int j = 0; j < PXU; ++j) {
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                floatx16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[t][ks], xf[j][ks], acc, 0, 0, 0);
                act16t<ACT>(acc);
                uint2v p[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    half4 hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) hv[e] = (half_t)acc[4 * g + e];
                    p[g] = __builtin_bit_cast(uint2v, hv);
                }
                const uint2v s0 = __builtin_amdgcn_permlane32_swap(p[0][0], p[1][0], false, false);
                const uint2v s1 = __builtin_amdgcn_permlane32_swap(p[0][1], p[1][1], false, false);
                const uint2v s2 = __builtin_amdgcn_permlane32_swap(p[2][0], p[3][0], false, false);
                const uint2v s3 = __builtin_amdgcn_permlane32_swap(p[2][1], p[3][1], false, false);
                const uint2v q0 = __builtin_amdgcn_permlane16_swap(s0[0], s2[0], false, false);
                const uint2v q1 = __builtin_amdgcn_permlane16_swap(s1[0], s3[0], false, false);
                const uint2v q2 = __builtin_amdgcn_permlane16_swap(s0[1], s2[1], false, false);
                const uint2v q3 = __builtin_amdgcn_permlane16_swap(s1[1], s3[1], false, false);
                const u32x4 lo4 = {q0[0], q1[0], q2[0], q3[0]}, hi4 = {q0[1], q1[1], q2[1], q3[1]};
                const int rr = row0 + 32 * j + (lane & 15);
                const int col = (ct0 + t) * 32 + colq;
                const unsigned off = ((unsigned)rr * (unsigned)NC + (unsigned)col) * 2u;
                const bool okc = col < NC && !(FLAGS & 2);
                __builtin_amdgcn_raw_buffer_store_b128(lo4, ors, (okc && rr < a.m) ? off : 0x80000000u, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(hi4, ors, (okc && rr + 16 < a.m) ? off + 32u * (unsigned)NC : 0x80000000u, 0, 0);
            }
        }
        if ((FLAGS & 1) && tc == 0) { tc = __builtin_amdgcn_s_memrealtime(); cc = __builtin_amdgcn_s_memtime(); }
        // the next unit's x tile must have landed; with FLAGS & 4 this unit's 2 RT PXU stores (always issued: buffer stores) stay in flight
        if (FLAGS & 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * RT * PXU) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((FLAGS & 1) && t2 == 0) t2 = __builtin_amdgcn_s_memrealtime();
    }
    if (FLAGS & 1) {
        const long long t3 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) { long long* sp = a.stamps + ((size_t)blockIdx.x * 4 + wave) * 4; sp[0] = t0; sp[1] = t1; sp[2] = t2; sp[3] = t3;
            long long* sq = a.stamps + (1 << 19) + ((size_t)blockIdx.x * 4 + wave) * 4; sq[0] = ta - t1; sq[1] = tb - ta; sq[2] = tc - tb; sq[3] = (cc - ca) * 1000 / max(tc - ta, 1LL); }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// candidate 3 ("kproj"): projections (long K, few channels) -- lab copy of demonet_amd/csrc/pwproj.hip with ablation flags
struct KArgs { const half_t* x; const half_t* wfrag; const float* bias; const half_t* residual; const float* se; half_t* out; int m, cin, cout, hw, act, xq; long long* stamps; };
constexpr int KP_KC = 64, KP_IMGS = 3;
__device__ __forceinline__ void kp_glds16(const void* gsrc, unsigned dst) {
#ifdef LAB_M0_NOSAVE
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(dst) : "memory");
#else
    glds16(gsrc, dst);
#endif
}
template <int N> __device__ __forceinline__ void kp_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// FL: 1 = stamps, 2 = weight pieces from one hot line, 4 = x pieces from one hot line, 8 = no MFMA
template <int NTL, int D, int FL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_kproj(KArgs a, int wg_per_group) {
    long long tw = 0, tc = 0, t0 = __builtin_amdgcn_s_memrealtime();
    extern __shared__ __attribute__((aligned(16))) unsigned char kp_lds[];
    constexpr int PPW = NTL + 4;                            // DMA instructions per wave and stage: NTL weight pieces + its 4 x pieces
    constexpr int STAGE = (NTL * 4 + 16) * 1024;            // bytes: [NTL tiles][4 K steps] fragments of 1 KB, then [4 waves][4 pieces] of x
    static_assert((D - 1) * PPW <= 63, "vmcnt is six bits");
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K = a.cin, NC = a.cout;
    const int KS = K >> 4;                                  // 16-deep K steps (K % 16 == 0)
    const int nst = (KS + 3) >> 2;                          // stages
    int r0, mend, w;
    if (a.xq > 0) {
        const int g = blockIdx.x & 7;
        w = blockIdx.x >> 3;
        r0 = g * a.xq * a.hw;
        mend = min(a.m, r0 + a.xq * a.hw);
    } else {
        w = blockIdx.x; r0 = 0; mend = a.m;
    }
    const int wrow0 = r0 + w * 128;                         // the workgroup's first row
    if (wrow0 >= mend) return;                              // (uniform over the workgroup)
    const int row0 = wrow0 + wave * 32;                     // the wave's first row
    const bool live = row0 < mend;                          // a wave without rows still copies its share of the weights
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)kp_lds;
    float* const tab = reinterpret_cast<float*>(kp_lds + D * STAGE);             // [KP_IMGS][K] squeeze-excitation scales (with a.se), then [32 NTL] bias
    float* const bsh = tab + (a.se ? KP_IMGS * K : 0);

    // ---- the DMA of one stage: this wave's x rows (4 pieces of 8 rows x 128 B) and NTL of the stage's 4 NTL weight fragments
    const int xr = min(row0 + (lane >> 3), mend - 1);       // piece p covers rows 8 p .. 8 p + 7 of the wave's tile: this lane's row of piece 0
    const unsigned char* const xbase = reinterpret_cast<const unsigned char*>(a.x);
    const unsigned char* const wbase = reinterpret_cast<const unsigned char*>(a.wfrag);
    const int part = (lane & 7) ^ (lane >> 3);              // the 16-byte part of the row's 128-byte run this lane copies (XOR swizzle by the row)
    const int ctiles = (NC + 31) >> 5;
    // share q (0 .. 3) of a stage's DMA: x piece q and the weight pieces q, q + 4, ... of this wave -- issued behind K step q of the stage in
    // flight, so that an instruction that holds the vector-memory issue path for ~100 cycles sits under that step's matrix instructions
    auto stage_dma_share = [&](int st, int q) {
        const unsigned dst = lds0 + (unsigned)((st % D) * STAGE);
        if (st < nst) {
            const int k0 = st * KP_KC + part * 8;           // first K column of this lane's part; beyond K (the last stage of K % 64 == 32): a valid dummy
            const int kc = min(k0, K - 8);
            const long row = min(xr + 8 * q, mend - 1);
            kp_glds16((FL & 4) ? (const void*)(wbase + lane * 16) : (const void*)(xbase + (row * K + kc) * 2), dst + (unsigned)(NTL * 4096 + (wave * 4 + q) * 1024));
#pragma unroll
            for (int i = q; i < NTL; i += 4) {
                const int piece = wave * NTL + i;           // (tile, step) = (piece / 4, piece % 4)
                const int t = min(piece >> 2, ctiles - 1), ks = min(st * 4 + (piece & 3), KS - 1);
                kp_glds16((FL & 2) ? wbase + lane * 16 : wbase + ((size_t)(t * KS + ks) * 64 + lane) * 16, dst + (unsigned)(piece * 1024));
            }
        } else {
            // behind the last stage: the same number of instructions (the counted wait below relies on it), one hot line, a ring slot nobody reads any more
            kp_glds16(wbase + lane * 16, dst + (unsigned)(q * 1024));
#pragma unroll
            for (int i = q; i < NTL; i += 4) kp_glds16(wbase + lane * 16, dst + (unsigned)((4 + i) * 1024));
        }
    };
#pragma unroll
    for (int st = 0; st < D - 1; ++st)
#pragma unroll
        for (int q = 0; q < 4; ++q) stage_dma_share(st, q);

    // ---- tables: squeeze-excitation scales of the workgroup's images, bias
    const int img_first = wrow0 / a.hw;
    const int nimg_all = (a.m + a.hw - 1) / a.hw;
    if (a.se) {
        for (int i = threadIdx.x; i < KP_IMGS * K; i += 256) {
            const int im = i / K, k = i - im * K;
            tab[i] = a.se[(size_t)min(img_first + im, nimg_all - 1) * K + k];
        }
    }
    for (int i = threadIdx.x; i < 32 * NTL; i += 256) bsh[i] = i < NC ? a.bias[i] : 0.f;
    // residual rows in the accumulator layout (pw_direct_kernel), requested now
    const int row = row0 + r;
    const int rowc = min(row, mend - 1);
    uint2 rres[NTL][4];
    const bool has_res = a.residual != nullptr;
    if (has_res) {
        const half_t* rp = a.residual + (size_t)rowc * NC;
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) rres[t][g] = *reinterpret_cast<const uint2*>(rp + min(t * 32 + 8 * g + 4 * hh, NC - 4));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (compiler-visible loads must not sit between the DMA stages and their counted waits)
    }
    const float* const srow = tab + (a.se ? (rowc / a.hw - img_first) * K + hh * 8 : 0);

    floatx16 acc[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    // this lane's B fragment of K step s of a stage: row r of the wave's tile = piece r / 8, row r % 8 of it; part 2 s + hh, swizzled
    const unsigned xrd = (unsigned)(NTL * 4096 + wave * 4096 + (r >> 3) * 1024 + (r & 7) * 128);
    const bool has_se = a.se != nullptr;
    for (int st = 0; st < nst; ++st) {
        // stage st has landed (this wave's pieces: all but those of the D - 2 younger stages; the other waves': the barrier)
        long long ta = (FL & 1) ? __builtin_amdgcn_s_memrealtime() : 0;
        kp_wait<(D - 2) * PPW>();
        __syncthreads();
        long long tb = (FL & 1) ? __builtin_amdgcn_s_memrealtime() : 0;
        tw += tb - ta;
        // (the DMA of stage st + D - 1 goes into the slot stage st - 1 was read from: every wave is past it)
        const unsigned char* sb = kp_lds + (st % D) * STAGE;
        const int steps = live ? min(4, KS - st * 4) : 0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < steps) {
                half8 bf = *reinterpret_cast<const half8*>(sb + xrd + (unsigned)(((2 * s + hh) ^ (r & 7)) * 16));
                if (has_se) {
                    // (half)((float)x * s) in one instruction per value: v_fma_mixlo / mixhi_f16 multiply in fp32 and round once to fp16 -- the
                    // rounding of every other 1x1 kernel's scale8 (8 instructions per fragment instead of 20: the scaling is this wave's vector work)
                    const float4 s0 = *reinterpret_cast<const float4*>(srow + st * KP_KC + s * 16);
                    const float4 s1 = *reinterpret_cast<const float4*>(srow + st * KP_KC + s * 16 + 4);
                    const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                    u32x4 xb = __builtin_bit_cast(u32x4, bf), ob;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        unsigned d = 0;
                        asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "+v"(d) : "v"(xb[e]), "v"(sv[2 * e]));
                        asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(xb[e]), "v"(sv[2 * e + 1]));
                        ob[e] = d;
                    }
                    bf = __builtin_bit_cast(half8, ob);
                }
#pragma unroll
                for (int t = 0; t < NTL; ++t) {
                    const half8 af = *reinterpret_cast<const half8*>(sb + (unsigned)((t * 4 + s) * 1024 + lane * 16));
                    if (FL & 8) acc[t][0] += (float)af[0] * (float)bf[0]; else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[t], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            stage_dma_share(st + D - 1, s);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (FL & 1) { const long long t1 = __builtin_amdgcn_s_memrealtime(); if (lane == 0) { long long* sp = a.stamps + ((size_t)blockIdx.x * 4 + wave) * 4; sp[0] = t0; sp[1] = t0 + tw; sp[2] = t1; sp[3] = t1; } }
    if (!live) return;
    // ---- epilogue
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)((unsigned)a.m * (unsigned)NC * 2u), 0x00020000);
    const int colq = 8 * ((lane >> 5) + 2 * ((lane >> 4) & 1));
    const int rr = row0 + (lane & 15);
    const int act = a.act;
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        if (t * 32 >= NC) break;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b = *reinterpret_cast<const float4*>(&bsh[t * 32 + 8 * g + 4 * hh]);
            acc[t][4 * g] += b.x; acc[t][4 * g + 1] += b.y; acc[t][4 * g + 2] += b.z; acc[t][4 * g + 3] += b.w;
        }
        act16(acc[t], act);
        if (has_res) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const half4 q = __builtin_bit_cast(half4, rres[t][g]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t][4 * g + e] += (float)q[e];
            }
        }
        uint2v p[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            half4 hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) hv[e] = (half_t)acc[t][4 * g + e];
            p[g] = __builtin_bit_cast(uint2v, hv);
        }
        const uint2v s0 = __builtin_amdgcn_permlane32_swap(p[0][0], p[1][0], false, false);
        const uint2v s1 = __builtin_amdgcn_permlane32_swap(p[0][1], p[1][1], false, false);
        const uint2v s2 = __builtin_amdgcn_permlane32_swap(p[2][0], p[3][0], false, false);
        const uint2v s3 = __builtin_amdgcn_permlane32_swap(p[2][1], p[3][1], false, false);
        const uint2v q0 = __builtin_amdgcn_permlane16_swap(s0[0], s2[0], false, false);
        const uint2v q1 = __builtin_amdgcn_permlane16_swap(s1[0], s3[0], false, false);
        const uint2v q2 = __builtin_amdgcn_permlane16_swap(s0[1], s2[1], false, false);
        const uint2v q3 = __builtin_amdgcn_permlane16_swap(s1[1], s3[1], false, false);
        const u32x4 lo4 = {q0[0], q1[0], q2[0], q3[0]}, hi4 = {q0[1], q1[1], q2[1], q3[1]};
        const int col = t * 32 + colq;
        const unsigned off = ((unsigned)rr * (unsigned)NC + (unsigned)col) * 2u;
        const bool okc = col < NC;
        __builtin_amdgcn_raw_buffer_store_b128(lo4, ors, (okc && rr < mend) ? off : 0x80000000u, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(hi4, ors, (okc && rr + 16 < mend) ? off + 32u * (unsigned)NC : 0x80000000u, 0, 0);
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// host
static std::vector<half_t> pack_frag_bias(const std::vector<half_t>& w, const std::vector<float>& b, int N, int K) {
    // [tile][KS1][lane = hh * 32 + r][8]; step s < KSF: columns s * 16 + hh * 8 ..; last step: kb = KSF * 16 + hh * 8: < K data, == K bias (hi, lo), else zero
    const int nt = (N + 31) / 32, KSF = K / 16, KS1 = KSF + 1;
    std::vector<half_t> o((size_t)nt * KS1 * 64 * 8, (half_t)0.f);
    for (int t = 0; t < nt; ++t)
        for (int s = 0; s < KS1; ++s)
            for (int l = 0; l < 64; ++l) {
                const int r = l & 31, hh = l >> 5, n = std::min(t * 32 + r, N - 1);
                half_t* d = &o[(((size_t)t * KS1 + s) * 64 + l) * 8];
                const int kb = s * 16 + hh * 8;
                if (kb < K) for (int e = 0; e < 8; ++e) d[e] = w[(size_t)n * K + kb + e];
                else if (kb == K) { const half_t hi = (half_t)b[n]; d[0] = hi; d[1] = (half_t)(b[n] - (float)hi); }
            }
    return o;
}

struct Timer {
    hipEvent_t e0, e1;
    Timer() { CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); }
};

int main(int argc, char** argv) {
    if (argc < 5) { printf("usage: pw_lab M K N act [iters]\n"); return 1; }
    const int M = atoi(argv[1]), K = atoi(argv[2]), N = atoi(argv[3]), act = atoi(argv[4]);
    const int iters = argc > 5 ? atoi(argv[5]) : 30;
    const int R = 8;
    printf("== pw_lab M=%d K=%d N=%d act=%d  (x %.1f MB, out %.1f MB, w %.2f MB; roofline at 8 TB/s %.2f us)\n", M, K, N, act, M * K * 2e-6, M * (double)N * 2e-6,
           K * N * 2e-6, (M * (double)(K + N) * 2 + K * N * 2) / 8e6);
    std::vector<half_t> hw((size_t)N * K);
    std::vector<float> hb(N);
    srand(1);
    for (auto& v : hw) v = (half_t)((rand() % 2001 - 1000) * (0.1f / 1000.f));
    for (auto& v : hb) v = (rand() % 2001 - 1000) * (1.f / 1000.f);
    std::vector<half_t> hwfb = pack_frag_bias(hw, hb, N, K);
    half_t *dx, *dw, *dwfb, *dout[R], *dref;
    float* db;
    CK(hipMalloc(&dx, (size_t)M * K * 2));
    CK(hipMalloc(&dw, hw.size() * 2)); CK(hipMalloc(&dwfb, hwfb.size() * 2)); CK(hipMalloc(&db, N * 4));
    for (int i = 0; i < R; ++i) CK(hipMalloc(&dout[i], (size_t)M * N * 2));
    CK(hipMalloc(&dref, (size_t)M * N * 2));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dwfb, hwfb.data(), hwfb.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    Timer T;
    const int KSF = K / 16;
    const int ctiles = (N + 31) / 32;
    auto runs_for = [&](int px, int cap, int& per) {
        const long ptiles = (M + 32 * px - 1) / (32 * px);
        int runs = (int)std::max(1L, std::min((long)ctiles, (long)cap / ptiles));
        per = (ctiles + runs - 1) / runs;
        return (ctiles + per - 1) / per;
    };
    std::vector<half_t> href((size_t)M * N), hout((size_t)M * N);
    bool have_ref = false;
    auto bench = [&](const char* name, auto launch, bool check) {
        // warm-up
        for (int i = 0; i < 3; ++i) { hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, dx, (size_t)M * K / 8, 7u); launch(dout[i % R]); }
        CK(hipStreamSynchronize(s));
        double tot = 0, mn = 1e9;
        for (int i = 0; i < iters; ++i) {
            hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, dx, (size_t)M * K / 8, 7u);
            CK(hipEventRecord(T.e0, s));
            launch(dout[i % R]);
            CK(hipEventRecord(T.e1, s));
            CK(hipEventSynchronize(T.e1));
            float ms; CK(hipEventElapsedTime(&ms, T.e0, T.e1));
            tot += ms; mn = std::min(mn, (double)ms);
        }
        CK(hipGetLastError());
        const double us = tot / iters * 1e3;
        std::string verdict = "";
        if (check) {
            CK(hipMemcpy(hout.data(), dout[(iters - 1) % R], hout.size() * 2, hipMemcpyDeviceToHost));
            if (!have_ref) { href = hout; have_ref = true; verdict = "  (reference)"; }
            else {
                size_t bad = 0;
                for (size_t i = 0; i < hout.size(); ++i) if (memcmp(&hout[i], &href[i], 2)) ++bad;
                verdict = bad ? "  MISMATCH " + std::to_string(bad) : "  bit-identical";
            }
        }
        printf("%-44s %8.2f us  (min %7.2f)  %6.2f TB/s%s\n", name, us, mn * 1e3, (M * (double)(K + N) * 2 + K * N * 2) / us / 1e6, verdict.c_str());
        fflush(stdout);
    };
    long long* dstamps; const size_t nst = (size_t)1 << 20;
    CK(hipMalloc(&dstamps, nst * 8)); CK(hipMemset(dstamps, 0, nst * 8));
    Args a{dx, dw, dwfb, db, nullptr, M, K, N, act, dstamps};
    auto stamp_report_fwd = 0; (void)stamp_report_fwd;
    auto stamp_report = [&](int nwaves, bool h2 = false) {
        std::vector<long long> h((size_t)nwaves * 4);
        CK(hipMemcpy(h.data(), dstamps, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> v[4]; long long base = -1;
        for (int w = 0; w < nwaves; ++w) if (h[w * 4] && (base < 0 || h[w * 4] < base)) base = h[w * 4];
        for (int w = 0; w < nwaves; ++w) if (h[w * 4]) for (int k = 0; k < 4; ++k) v[k].push_back((h[w * 4 + k] - base) * 0.01);
        const char* nm[4] = {"wave start", "operands ready", "first unit done", "wave done"};
        for (int k = 0; k < 4; ++k) { std::sort(v[k].begin(), v[k].end()); const size_t n = v[k].size(); if (!n) continue;
            printf("      %-18s min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f us   (%zu waves)\n", nm[k], v[k][0], v[k][n / 10], v[k][n / 2], v[k][n * 9 / 10], v[k][n - 1], n); }
        if (h2) {
            std::vector<long long> g((size_t)nwaves * 4);
            CK(hipMemcpy(g.data(), dstamps + (1 << 19), g.size() * 8, hipMemcpyDeviceToHost));
            const char* nm2[4] = {"x fragments read", "next DMA issued", "unit issued", "MHz*10 in unit"};
            for (int k = 0; k < 4; ++k) { std::vector<double> q; for (int w = 0; w < nwaves; ++w) if (h[w * 4]) q.push_back(g[w * 4 + k] * (k < 3 ? 0.01 : 1.0));
                std::sort(q.begin(), q.end()); const size_t n = q.size(); if (!n) continue;
                printf("      first unit: %-18s min %7.2f  p10 %7.2f  median %7.2f  p90 %7.2f  max %7.2f\n", nm2[k], q[0], q[n / 10], q[n / 2], q[n * 9 / 10], q[n - 1]); }
        }
        CK(hipMemset(dstamps, 0, nst * 8));
    };
    // CPU check of the reference on a few rows
    auto cpu_check = [&]() {
        std::vector<half_t> hx((size_t)M * K);
        CK(hipMemcpy(hx.data(), dx, hx.size() * 2, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int t = 0; t < 64; ++t) {
            const int m = (int)(((long)t * 7919 * 131) % M), n = (t * 37) % N;
            double acc = hb[n];
            for (int k = 0; k < K; ++k) acc += (double)(float)hx[(size_t)m * K + k] * (double)(float)hw[(size_t)n * K + k];
            double v = acc;
            if (act == ACT_RELU) v = std::max(v, 0.0); else if (act == ACT_RELU6) v = std::min(std::max(v, 0.0), 6.0);
            else if (act == ACT_HSWISH) v = v * std::min(std::max(v + 3.0, 0.0), 6.0) / 6.0;
            worst = std::max(worst, fabs(v - (double)(float)href[(size_t)m * N + n]) / (1.0 + fabs(v)));
        }
        printf("reference vs fp64 on 64 samples: max rel err %.3e\n", worst);
    };

#define BASE(KSFv, PXv, ABLv, nm) if (KSF == KSFv) { int per; const int runs = runs_for(PXv, 2800, per); const int tiles = (M + 128 * PXv - 1) / (128 * PXv); \
        bench(nm, [&](half_t* o) { Args b = a; b.out = o; hipLaunchKernelGGL((k_base<KSFv, PXv, ABLv>), dim3(tiles * runs), dim3(256), 0, s, b, tiles, per); }, (ABLv & ~16) == 0); }
#define BASES(KSFv) BASE(KSFv, 2, 0, "base stream<" #KSFv ",2> (round 5)") if (getenv("LAB_ABL")) { BASES2(KSFv) }
#define BASES2(KSFv) BASE(KSFv, 2, 16, "base + 64-B store pieces") BASE(KSFv, 2, 1, "base, no weight loads") \
        BASE(KSFv, 2, 2, "base, no stores") BASE(KSFv, 2, 4, "base, no x loads") BASE(KSFv, 2, 3, "base, no weight loads, no stores") \
        BASE(KSFv, 2, 5, "base, no weight loads, no x loads") BASE(KSFv, 2, 7, "base, MFMA + epilogue VALU only") BASE(KSFv, 2, 15, "base, launch + VALU only") \
        BASE(KSFv, 1, 0, "base stream<" #KSFv ",1>")

    if (K >= 256) {
        // projection lab: fragment-major weights WITHOUT the bias step (plan.py fragment_major), no SE / residual (the ablations are about the streams)
        const int nt = (N + 31) / 32, KS = K / 16;
        std::vector<half_t> wf((size_t)nt * KS * 64 * 8, (half_t)0.f);
        for (int t = 0; t < nt; ++t) for (int ks = 0; ks < KS; ++ks) for (int l = 0; l < 64; ++l) { const int r = l & 31, hh = l >> 5, n = t * 32 + r;
            if (n < N) for (int e = 0; e < 8; ++e) wf[(((size_t)t * KS + ks) * 64 + l) * 8 + e] = hw[(size_t)n * K + ks * 16 + hh * 8 + e]; }
        half_t* dwf; CK(hipMalloc(&dwf, wf.size() * 2)); CK(hipMemcpy(dwf, wf.data(), wf.size() * 2, hipMemcpyHostToDevice));
        const int hwp = 400;
#define KPROJ(NTLv, Dv, FLv, nm) if ((N + 31) / 32 <= NTLv && (NTLv == 3 || (N + 31) / 32 > (NTLv == 4 ? 3 : NTLv == 6 ? 4 : 6))) { const int wpg = (M + 127) / 128; \
        const size_t ldsb = (size_t)Dv * (NTLv * 4 + 16) * 1024 + 32 * NTLv * 4; \
        CK(hipFuncSetAttribute((const void*)k_kproj<NTLv, Dv, FLv>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        bench(nm, [&](half_t* o) { KArgs b{dx, dwf, db, nullptr, nullptr, o, M, K, N, hwp, act, 0, dstamps}; hipLaunchKernelGGL((k_kproj<NTLv, Dv, FLv>), dim3(wpg), dim3(256), ldsb, s, b, wpg); }, false); \
        if (FLv & 1) stamp_report(wpg * 4); }
#define KPROJS(NTLv, Dv) KPROJ(NTLv, Dv, 0, "kproj") KPROJ(NTLv, Dv, 1, "kproj stamps (operands ready = time in waits)") KPROJ(NTLv, Dv, 2, "kproj, weights from one hot line") \
        KPROJ(NTLv, Dv, 4, "kproj, x from one hot line") KPROJ(NTLv, Dv, 6, "kproj, both from one hot line") KPROJ(NTLv, Dv, 8, "kproj, no MFMA") KPROJ(NTLv, Dv, 14, "kproj, launch + DMA issue + LDS only")
        KPROJS(3, 4) KPROJS(4, 4) KPROJS(6, 3) KPROJS(8, 3)
        return 0;
    }
    { int per; const int runs = runs_for(2, 2800, per); const int tiles = (M + 255) / 256;
      bench("null kernel, same grid", [&](half_t* o) { Args b = a; b.out = o; hipLaunchKernelGGL(k_null, dim3(tiles * runs), dim3(256), 0, s, b, tiles, per); }, false); }
    BASES(1) BASES(2) BASES(4) BASES(5) BASES(7)
    if (have_ref) cpu_check();
    {
        int per; const int runs = runs_for(2, 2800, per); const int tiles = (M + 255) / 256;
        bench("store only, 32 rows x 32 B / instr", [&](half_t* o) { hipLaunchKernelGGL((k_store<0>), dim3(tiles * runs), dim3(256), 0, s, o, M, N, tiles, per, 1u, dstamps); }, false); stamp_report(tiles * runs * 4);
        bench("store only, 16 rows x 64 B / instr", [&](half_t* o) { hipLaunchKernelGGL((k_store<1>), dim3(tiles * runs), dim3(256), 0, s, o, M, N, tiles, per, 1u, dstamps); }, false); stamp_report(tiles * runs * 4);
        bench("store only, 8 rows x 128 B / instr", [&](half_t* o) { hipLaunchKernelGGL((k_store<2>), dim3(tiles * runs), dim3(256), 0, s, o, M, N, tiles, per, 1u, dstamps); }, false); stamp_report(tiles * runs * 4);
        bench("store only, linear 1 KB / instr", [&](half_t* o) { hipLaunchKernelGGL((k_store<3>), dim3(tiles * runs), dim3(256), 0, s, o, M, N, tiles, per, 1u, dstamps); }, false); stamp_report(tiles * runs * 4);
    }
#define LDSW(KS1v, PXv, STv, cap, nm) if (KSF + 1 == KS1v) { int per; const int runs = runs_for(PXv, cap, per); const int tiles = (M + 128 * PXv - 1) / (128 * PXv); \
        const int ldsb = per * KS1v * 1024; CK(hipFuncSetAttribute((const void*)k_ldsw<KS1v, PXv, STv, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        CK(hipFuncSetAttribute((const void*)k_ldsw<KS1v, PXv, STv, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        CK(hipFuncSetAttribute((const void*)k_ldsw<KS1v, PXv, STv, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        char nm2[128]; snprintf(nm2, sizeof nm2, "%s runs=%d per=%d lds=%dK", nm, runs, per, ldsb / 1024); \
        bench(nm2, [&](half_t* o) { Args b = a; b.out = o; hipLaunchKernelGGL((k_ldsw<KS1v, PXv, STv, 0>), dim3(tiles * runs), dim3(256), ldsb, s, b, tiles, per); }, true); \
        snprintf(nm2, sizeof nm2, "%s runs=%d per=%d NO STORES", nm, runs, per); \
        bench(nm2, [&](half_t* o) { Args b = a; b.out = o; hipLaunchKernelGGL((k_ldsw<KS1v, PXv, STv, 2>), dim3(tiles * runs), dim3(256), ldsb, s, b, tiles, per); }, false); \
        snprintf(nm2, sizeof nm2, "%s runs=%d per=%d stamps", nm, runs, per); \
        bench(nm2, [&](half_t* o) { Args b = a; b.out = o; hipLaunchKernelGGL((k_ldsw<KS1v, PXv, STv, 1>), dim3(tiles * runs), dim3(256), ldsb, s, b, tiles, per); }, false); \
        stamp_report(tiles * runs * 4); }
#define LDSWS(KS1v) LDSW(KS1v, 2, 1, 2800, "ldsw px2 st64") LDSW(KS1v, 1, 1, 2800, "ldsw px1 st64") LDSW(KS1v, 1, 1, 5600, "ldsw px1 st64")
    if (getenv("LAB_LDSW")) { LDSWS(2) LDSWS(3) LDSWS(5) LDSWS(6) LDSWS(8) }

#define PWS1(KS1v, RTv, PXUv, ACTv, wpc, FL, nm) if (KSF + 1 == KS1v && act == ACTv) { const int runs = (ctiles + RTv - 1) / RTv; \
        const int n_units = (M + 32 * PXUv - 1) / (32 * PXUv); \
        int nwg = std::min((wpc) * 256, (n_units + 3) / 4 * runs); nwg = std::max(runs, nwg / runs * runs); \
        const int xp = (PXUv * K + 15) / 16; const int ldsb = RTv * KS1v * 1024 + 4 * xp * 1024; \
        CK(hipFuncSetAttribute((const void*)k_pws<KS1v, RTv, PXUv, ACTv, FL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        CK(hipFuncSetAttribute((const void*)k_pws<KS1v, RTv, PXUv, ACTv, FL | 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        CK(hipFuncSetAttribute((const void*)k_pws<KS1v, RTv, PXUv, ACTv, FL | 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        char nm2[128]; snprintf(nm2, sizeof nm2, "%s RT=%d PXU=%d wg=%d (%d/CU)", nm, RTv, PXUv, nwg, wpc); \
        bench(nm2, [&](half_t* o) { Args b = a; b.out = o; hipLaunchKernelGGL((k_pws<KS1v, RTv, PXUv, ACTv, FL>), dim3(nwg), dim3(256), ldsb, s, b, runs, n_units); }, true); \
        snprintf(nm2, sizeof nm2, "%s RT=%d PXU=%d wg=%d NO STORES", nm, RTv, PXUv, nwg); \
        bench(nm2, [&](half_t* o) { Args b = a; b.out = o; hipLaunchKernelGGL((k_pws<KS1v, RTv, PXUv, ACTv, FL | 2>), dim3(nwg), dim3(256), ldsb, s, b, runs, n_units); }, false); \
        snprintf(nm2, sizeof nm2, "%s RT=%d PXU=%d wg=%d stamps", nm, RTv, PXUv, nwg); \
        bench(nm2, [&](half_t* o) { Args b = a; b.out = o; hipLaunchKernelGGL((k_pws<KS1v, RTv, PXUv, ACTv, FL | 1>), dim3(nwg), dim3(256), ldsb, s, b, runs, n_units); }, false); \
        stamp_report(nwg * 4, true); }
#define PWS(KS1v, RTv, PXUv, ACTv) PWS1(KS1v, RTv, PXUv, ACTv, 2, 4, "pws counted") PWS1(KS1v, RTv, PXUv, ACTv, 3, 4, "pws counted")
    PWS(8, 3, 1, 0) PWS(8, 3, 2, 3)
    PWS(8, 3, 1, 3) PWS(6, 3, 1, 3) PWS(6, 5, 1, 3) PWS(3, 4, 1, 3) PWS(3, 4, 2, 3) PWS(2, 3, 2, 1) PWS(2, 3, 1, 1) PWS(2, 3, 4, 1)
    return 0;
}
