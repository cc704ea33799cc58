// A run of inverted-residual blocks on the small maps (20 x 20 and 10 x 10) in ONE launch: one workgroup per image walks the
// blocks with the block input resident in LDS; nothing but the weights (L2, shared by all images) and the pyramid feature moves.
//
// reference ops replaced: InvertedResidual (mobilenetv3.py:61-99: expand 1x1 -> depthwise kxk -> SqueezeExcitation (:22-40) ->
//   project 1x1 (+ residual)) for features.0.8 .. features.1.2 of the SSDLite backbone, incl. the C4 block that the extractor
//   splits at its expansion layer (ssd_mobilenetv3.py:104-108: the expanded 672 x 20 x 20 map is pyramid feature 0).
//
// Why: as separate launches these 8 blocks are 31 dependent kernels of 5-20 us each, every one far from filling the chip (a 20 x 20
// map of 32 images is 12 800 pixels) and every one paying the launch and two memory round trips; the chain's cost does not depend
// on the batch size (T(n) = 570 us + 12.6 us * n). Here a block is a loop over 64-channel chunks of the expanded
// tensor inside one workgroup:
//   expand   E[pixel][64] = act(X . W1 + b1) on the matrix cores, weights streamed as MFMA A fragments from the fragment-major copy
//            (1 KB contiguous per wave load), rounded to fp16 into LDS exactly where the unfused path rounds it into HBM;
//   depthwise over the LDS tile (the whole image is resident: no halo, out-of-image taps read a zero row), computed by each lane
//            directly in the B-fragment layout of the projection MFMA (lane = pixel, 8 consecutive channels), rounded to fp16;
//   project  accumulators live in registers across the chunks; bias (+ residual from the resident input) at the end, the result
//            overwrites the input in LDS.
// Blocks with squeeze-excitation need the pooled mean of the WHOLE depthwise output before the projection can start: their first
// pass parks the depthwise output in a per-image global scratch (read back by the same CU: L2 hits, no cross-workgroup traffic)
// and accumulates the channel sums; the two FCs run in the workgroup; the second pass scales and projects.
// Rounding points are those of the separate kernels (fp16 E, fp16 D, fp16 D*s, fp32 accumulation), so the results agree with the
// launch-per-layer path to the fp16 tolerance (the pooled sums are added in a different order).
#include "common.h"

static long long* g_trunk_stamps = nullptr;     // dev hook: per-workgroup stamps [n][64]
extern "C" __attribute__((visibility("default"))) void dn_debug_trunk_stamps(void* dev_ptr) { g_trunk_stamps = (long long*)dev_ptr; }
#define TK_STAMP(k) do { if (a.stamps && threadIdx.x == 0 && (k) < 64) a.stamps[(size_t)blockIdx.x * 64 + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)

namespace {

constexpr int TT = 512;             // threads per workgroup
constexpr int NWV = TT / 64;        // waves
constexpr int XS = 120;             // halfs per X row: up to 112 channels + 8 (15 x 16 B: odd -> conflict-free b128 reads)
constexpr int ES = 72;              // halfs per E row: 64-channel chunk + 8
constexpr int MAXPX = 400;          // pixels of the largest map (20 x 20)
constexpr int CW = 64;              // chunk width (expanded channels)
constexpr int MAXC = 672, MAXSQ = 168;

__device__ __forceinline__ float hsig(float v) { return fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f); }

__global__ __launch_bounds__(TT) void trunk_kernel(TrunkArgs a, int nimg) {
    extern __shared__ __attribute__((aligned(16))) half_t lds[];
    half_t* X = lds;                                            // [MAXPX][XS]       block input / output
    half_t* E = X + MAXPX * XS;                                 // [MAXPX + 1][ES]   expanded chunk; row MAXPX stays zero
    half_t* Wd = E + (MAXPX + 1) * ES;                          // [25][CW]          depthwise weights of the chunk
    float* Bd = reinterpret_cast<float*>(Wd + 25 * CW);         // [CW]              depthwise bias of the chunk
    float* vec = Bd + CW;                                       // [MAXC]            pooled sums -> mean -> SE scale
    float* zv = vec + MAXC;                                     // [MAXSQ]           FC1 output
    float* red = zv + MAXSQ;                                    // [NWV][CW]         per-wave partial channel sums
    float* fcs = reinterpret_cast<float*>(E);                   // [2][TT]           FC K-slice partials (E is idle during the FCs)
    __shared__ TrunkBlock blk_sh[TRUNK_MAX_BLOCKS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    for (int i = tid; i < (int)(sizeof(TrunkBlock) * TRUNK_MAX_BLOCKS / 4); i += TT)
        reinterpret_cast<int*>(blk_sh)[i] = reinterpret_cast<const int*>(a.blk)[i];
    // XCD grouping (common.h): workgroup b serves image (b % 8) * xq + b / 8
    const int n = a.xq > 0 ? (int)(blockIdx.x & 7) * a.xq + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (n >= nimg) return;
    TK_STAMP(0);
    if (tid < ES / 8) *reinterpret_cast<uint4*>(E + MAXPX * ES + tid * 8) = make_uint4(0, 0, 0, 0);
    {
        // stage the first block's input [pixels][cin] (NHWC fp16, contiguous per image)
        const int c8 = a.cin0 / 8, chunks = a.px0 * c8;
        const uint4* src = reinterpret_cast<const uint4*>(a.in0 + (size_t)n * a.in0_stride);
        for (int i0 = tid; i0 < chunks; i0 += TT * 4) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = src[min(i0 + TT * u, chunks - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + TT * u;
                if (i < chunks) *reinterpret_cast<uint4*>(X + (size_t)(i / c8) * XS + (i % c8) * 8) = v[u];
            }
        }
    }
    __syncthreads();
    TK_STAMP(1);
    const unsigned char* const Wb = reinterpret_cast<const unsigned char*>(a.weights);
    int stamp = 2;
    for (int bi = 0; bi < a.count; ++bi) {
        const TrunkBlock B = blk_sh[bi];
        const int npi = B.hin * B.hin, npo = B.hout * B.hout;
        const int RT = (npi + 31) >> 5, ORT = (npo + 31) >> 5;
        const int KS1 = (B.cin + 15) >> 4, KS3 = (B.cexp + 15) >> 4;
        const int CT3 = (B.cout + 31) >> 5;                      // <= 4
        const half_t* w1f = reinterpret_cast<const half_t*>(Wb + B.w1f_off);
        const float* b1 = reinterpret_cast<const float*>(Wb + B.b1_off);
        const half_t* wd = reinterpret_cast<const half_t*>(Wb + B.wd_off);
        const float* bd = reinterpret_cast<const float*>(Wb + B.bd_off);
        const half_t* w3f = reinterpret_cast<const half_t*>(Wb + B.w3f_off);
        const float* b3 = reinterpret_cast<const float*>(Wb + B.b3_off);
        half_t* dsc = a.dscratch + (size_t)n * a.dscratch_stride + B.dsc_off;      // [npo][cexp] depthwise output of an SE block (a slice of
                                                                                   // its own per block: every address is written once and read once per launch)
        half_t* feat = B.feat_out ? B.feat_out + (size_t)n * B.feat_stride : nullptr;
        floatx16 pacc[2][4];
        const int ect = wave & 1;                                // expand: this wave's 32-channel tile of the chunk (units wave, wave + 8, ...)
        const int kk = B.k * B.k;
        // per chunk: depthwise weights / bias -> LDS, expand -> E (LDS), optional feature write. Ends with E complete (barrier).
        auto expand_chunk = [&](int c0) {
            for (int i = tid; i < kk * (CW / 8); i += TT) {
                const int tap = i / (CW / 8), q = i - tap * (CW / 8);
                uint4 v = make_uint4(0, 0, 0, 0);
                if (c0 + q * 8 < B.cexp) v = *reinterpret_cast<const uint4*>(wd + (size_t)tap * B.cexp + c0 + q * 8);
                *reinterpret_cast<uint4*>(Wd + tap * CW + q * 8) = v;
            }
            if (tid < CW) Bd[tid] = (c0 + tid < B.cexp) ? bd[c0 + tid] : 0.f;
            // E[pixel][chunk] on the matrix cores (A = weights, B = pixels: lane = pixel, 4 consecutive channels per register group)
            if (c0 + ect * 32 < B.cexp) {
                half8 wf[7];
                const half_t* wp = w1f + ((size_t)((c0 >> 5) + ect) * KS1) * 512 + lane * 8;
#pragma unroll
                for (int ks = 0; ks < 7; ++ks) {
                    half8 w = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (ks < KS1) w = *reinterpret_cast<const half8*>(wp + (size_t)ks * 512);
                    wf[ks] = w;
                }
                float4 bv[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = c0 + ect * 32 + 8 * g + 4 * hh;
                    bv[g] = (c < B.cexp) ? *reinterpret_cast<const float4*>(b1 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                for (int rt = wave >> 1; rt < RT; rt += NWV / 2) {
                    floatx16 acc;
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
                    const int px = min(rt * 32 + r, npi - 1);       // rows beyond the map: computed, never stored
                    const half_t* xrow = X + (size_t)px * XS + hh * 8;
#pragma unroll
                    for (int ks = 0; ks < 7; ++ks)
                        if (ks < KS1) {
                            const half8 xf = *reinterpret_cast<const half8*>(xrow + ks * 16);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks], xf, acc, 0, 0, 0);
                        }
                    if (rt * 32 + r < npi) {
                        half_t* erow = E + (size_t)(rt * 32 + r) * ES + ect * 32 + 4 * hh;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            half4 hv;
                            float t4[4] = {acc[4 * g + 0] + bv[g].x, acc[4 * g + 1] + bv[g].y, acc[4 * g + 2] + bv[g].z, acc[4 * g + 3] + bv[g].w};
                            dn_act_n<float[4], 4>(t4, B.act1);
                            hv[0] = (half_t)t4[0]; hv[1] = (half_t)t4[1]; hv[2] = (half_t)t4[2]; hv[3] = (half_t)t4[3];
                            *reinterpret_cast<half4*>(erow + 8 * g) = hv;
                        }
                    }
                }
            }
            __syncthreads();
            if (feat) {
                // the expanded map is a pyramid feature: this chunk's channels of every pixel -> HBM (128 B per pixel)
                const int cw8 = min(CW, B.cexp - c0) >> 3;
                for (int i = tid; i < npi * cw8; i += TT) {
                    const int px = i / cw8, q = i - px * cw8;
                    *reinterpret_cast<uint4*>(feat + (size_t)px * B.cexp + c0 + q * 8) = *reinterpret_cast<const uint4*>(E + (size_t)px * ES + q * 8);
                }
            }
        };
        // depthwise of (row tile ort, 16-channel step ks of chunk c0) in the projection's B-fragment layout: lane = (output pixel r,
        // channel group 2 ks + hh); returns the activated fp32 values (zero for lanes outside the map / beyond cexp)
        auto depthwise = [&](int c0, int ort, int ks, float (&v)[8]) {
            const int op = ort * 32 + r;
            const bool pvalid = op < npo;
            const int oy = op / B.hout, ox = op - oy * B.hout;
            const int iy0 = oy * B.stride - B.pad, ix0 = ox * B.stride - B.pad;
            const int cg = 2 * ks + hh;
            const bool cvalid = c0 + cg * 8 < B.cexp;
            float acc[8];
            {
                const float4 q0 = *reinterpret_cast<const float4*>(&Bd[cg * 8]), q1 = *reinterpret_cast<const float4*>(&Bd[cg * 8 + 4]);
                acc[0] = q0.x; acc[1] = q0.y; acc[2] = q0.z; acc[3] = q0.w; acc[4] = q1.x; acc[5] = q1.y; acc[6] = q1.z; acc[7] = q1.w;
            }
#pragma unroll 1
            for (int ky = 0; ky < B.k; ++ky) {
                const int iy = iy0 + ky;
                const bool yok = pvalid && iy >= 0 && iy < B.hin;
#pragma unroll 1
                for (int kx = 0; kx < B.k; ++kx) {
                    const int ix = ix0 + kx;
                    const int row = (yok && ix >= 0 && ix < B.hin) ? iy * B.hin + ix : MAXPX;      // zero row outside the map
                    const uint4 ev = *reinterpret_cast<const uint4*>(E + (size_t)row * ES + cg * 8);
                    const uint4 wv = *reinterpret_cast<const uint4*>(Wd + (ky * B.k + kx) * CW + cg * 8);
                    fma_mix_h8(acc, ev, wv);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (pvalid && cvalid) ? dn_act(acc[e], B.act2) : 0.f;
        };
        if (!B.has_se) {
            // ---------------- expand -> depthwise -> project, chunk by chunk; the projection accumulates over the chunks
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int e = 0; e < 16; ++e) pacc[t][c][e] = 0.f;
            for (int c0 = 0; c0 < B.cexp; c0 += CW) {
                expand_chunk(c0);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int ort = wave + t * NWV;
                    if (ort >= ORT) break;                           // wave-uniform
                    for (int ks = 0; ks < 4; ++ks) {
                        if (c0 + ks * 16 >= B.cexp) break;           // wave-uniform
                        const int kstep = (c0 >> 4) + ks;
                        half8 w3[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {                // requested ahead of the depthwise arithmetic
                            half8 w = {0, 0, 0, 0, 0, 0, 0, 0};
                            if (c < CT3) w = *reinterpret_cast<const half8*>(w3f + ((size_t)c * KS3 + kstep) * 512 + lane * 8);
                            w3[c] = w;
                        }
                        float v[8];
                        depthwise(c0, ort, ks, v);
                        half8 df;
#pragma unroll
                        for (int e = 0; e < 8; ++e) df[e] = (half_t)v[e];
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (c < CT3) pacc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3[c], df, pacc[t][c], 0, 0, 0);
                    }
                }
                __syncthreads();        // E / Wd / Bd are rewritten by the next chunk
                if (stamp < 60) { TK_STAMP(stamp); ++stamp; }
            }
        } else {
            // ---------------- pass A: expand -> depthwise -> park the fp16 result + channel sums, chunk by chunk
            for (int i = tid; i < B.cexp; i += TT) vec[i] = 0.f;
            for (int c0 = 0; c0 < B.cexp; c0 += CW) {
                expand_chunk(c0);
                for (int ks = 0; ks < 4; ++ks) {
                    if (c0 + ks * 16 >= B.cexp) break;               // wave-uniform
                    float psum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    for (int ort = wave; ort < ORT; ort += NWV) {
                        float v[8];
                        depthwise(c0, ort, ks, v);
                        half8 df;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { df[e] = (half_t)v[e]; psum[e] += v[e]; }
                        const int op = ort * 32 + r, cb = c0 + (2 * ks + hh) * 8;
                        if (op < npo && cb < B.cexp) *reinterpret_cast<half8*>(dsc + (size_t)op * B.cexp + cb) = df;
                    }
                    // channel sums: the 32 pixel lanes of the wave (fixed shuffle tree), then the waves (fixed order, below)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float q = psum[e];
                        q += __shfl_xor(q, 1); q += __shfl_xor(q, 2); q += __shfl_xor(q, 4); q += __shfl_xor(q, 8); q += __shfl_xor(q, 16);
                        if (r == 0) red[wave * CW + (2 * ks + hh) * 8 + e] = q;
                    }
                }
                __syncthreads();
                if (tid < CW && c0 + tid < B.cexp) {
                    float q = 0.f;
#pragma unroll
                    for (int w = 0; w < NWV; ++w) q += red[w * CW + tid];
                    vec[c0 + tid] = q;
                }
                __syncthreads();        // E / Wd / Bd / red are rewritten by the next chunk
                if (stamp < 60) { TK_STAMP(stamp); ++stamp; }
            }
            // ---------------- squeeze-excitation FCs (mobilenetv3.py:31-36) on the pooled means, in the workgroup
            const int c = B.cexp, sq = B.sq;
            const unsigned* w1t = reinterpret_cast<const unsigned*>(Wb + B.se_w1t_off);     // [c][sq/2] half pairs
            const unsigned* w2t = reinterpret_cast<const unsigned*>(Wb + B.se_w2t_off);     // [sq][c/2]
            const float* sb1 = reinterpret_cast<const float*>(Wb + B.se_b1_off);
            const float* sb2 = reinterpret_cast<const float*>(Wb + B.se_b2_off);
            const float inv = 1.0f / (float)npo;
            for (int i = tid; i < c; i += TT) vec[i] *= inv;
            __syncthreads();
            auto lo = [](unsigned w) { return (float)__builtin_bit_cast(half_t, (unsigned short)(w & 0xffffu)); };
            auto hi = [](unsigned w) { return (float)__builtin_bit_cast(half_t, (unsigned short)(w >> 16)); };
            constexpr int FB = 16;
            {
                // fc1: thread = (output pair j, K slice): sq/2 <= 84 pairs -> 128-wide layout, 4 slices of the c inputs
                const int sq2 = sq >> 1;
                const int j = tid & 127, sl = tid >> 7;
                const int per = (c + 3) >> 2, i0 = sl * per, i1 = min(c, i0 + per);
                float t0 = 0.f, t1 = 0.f;
                if (j < sq2) {
                    for (int ib = i0; ib < i1; ib += FB) {
                        unsigned v[FB];
#pragma unroll
                        for (int u = 0; u < FB; ++u) v[u] = w1t[(size_t)min(ib + u, i1 - 1) * sq2 + j];
#pragma unroll
                        for (int u = 0; u < FB; ++u) {
                            const float m = (ib + u < i1) ? vec[ib + u] : 0.f;
                            t0 += lo(v[u]) * m;
                            t1 += hi(v[u]) * m;
                        }
                    }
                }
                fcs[tid] = t0;
                fcs[TT + tid] = t1;
                __syncthreads();
                if (tid < sq) {
                    float q = sb1[tid];
                    const float* pp = fcs + (tid & 1) * TT + (tid >> 1);
#pragma unroll
                    for (int w = 0; w < 4; ++w) q += pp[w * 128];
                    zv[tid] = fmaxf(q, 0.f);
                }
                __syncthreads();
            }
            {
                // fc2: thread = output pair i (c/2 <= 336 pairs), all sq inputs
                const int c2 = c >> 1;
                for (int i = tid; i < c2; i += TT) {
                    float t0 = 0.f, t1 = 0.f;
                    for (int jb = 0; jb < sq; jb += FB) {
                        unsigned v[FB];
#pragma unroll
                        for (int u = 0; u < FB; ++u) v[u] = w2t[(size_t)min(jb + u, sq - 1) * c2 + i];
#pragma unroll
                        for (int u = 0; u < FB; ++u) {
                            const float zz = (jb + u < sq) ? zv[jb + u] : 0.f;
                            t0 += lo(v[u]) * zz;
                            t1 += hi(v[u]) * zz;
                        }
                    }
                    fcs[2 * i] = hsig(t0 + sb2[2 * i]);
                    fcs[2 * i + 1] = hsig(t1 + sb2[2 * i + 1]);
                }
                __syncthreads();
                for (int i = tid; i < c; i += TT) vec[i] = fcs[i];
                __syncthreads();
            }
            if (stamp < 60) { TK_STAMP(stamp); ++stamp; }
            // ---------------- pass B: scale the parked depthwise output (fp16 product, as the projection kernels round it) and project.
            // Every lane reads back exactly what it stored itself; the wait makes sure those stores have left the wave.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                    for (int e = 0; e < 16; ++e) pacc[t][cc][e] = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ort = wave + t * NWV;
                if (ort >= ORT) break;
                const int op = min(ort * 32 + r, npo - 1);
                const half_t* drow = dsc + (size_t)op * c + hh * 8;
                for (int ks0 = 0; ks0 < KS3; ks0 += 4) {
                    half8 dv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        half8 d = {0, 0, 0, 0, 0, 0, 0, 0};
                        if (ks0 + u < KS3 && (ks0 + u) * 16 + hh * 8 < c) d = *reinterpret_cast<const half8*>(drow + (ks0 + u) * 16);
                        dv[u] = d;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (ks0 + u >= KS3) break;
                        const int cb = (ks0 + u) * 16 + hh * 8;
                        half8 df;
#pragma unroll
                        for (int e = 0; e < 8; ++e) df[e] = (cb + e < c) ? (half_t)((float)dv[u][e] * vec[cb + e]) : (half_t)0.f;
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc)
                            if (cc < CT3) {
                                const half8 w = *reinterpret_cast<const half8*>(w3f + ((size_t)cc * KS3 + ks0 + u) * 512 + lane * 8);
                                pacc[t][cc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, df, pacc[t][cc], 0, 0, 0);
                            }
                    }
                }
            }
        }
        __syncthreads();        // every wave is done reading X (expand) before the output overwrites it
        // ---------------- block output: bias (+ residual from the resident input), one fp16 rounding, back into X
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ort = wave + t * NWV;
            if (ort >= ORT) break;
            const int op = ort * 32 + r;
            if (op < npo) {
                half_t* xrow = X + (size_t)op * XS;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
                    if (cc < CT3) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int co = cc * 32 + 8 * g + 4 * hh;
                            if (co < B.cout) {              // cout % 8 == 0: the 4-channel group is entirely in range
                                const float4 bq = *reinterpret_cast<const float4*>(b3 + co);
                                float v[4] = {pacc[t][cc][4 * g + 0] + bq.x, pacc[t][cc][4 * g + 1] + bq.y, pacc[t][cc][4 * g + 2] + bq.z,
                                              pacc[t][cc][4 * g + 3] + bq.w};
                                if (B.has_res) {
                                    const half4 rr = *reinterpret_cast<const half4*>(xrow + co);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] += (float)rr[e];
                                }
                                half4 hv;
#pragma unroll
                                for (int e = 0; e < 4; ++e) hv[e] = (half_t)v[e];
                                *reinterpret_cast<half4*>(xrow + co) = hv;
                            }
                        }
                    }
            }
        }
        __syncthreads();
        if (stamp < 60) { TK_STAMP(stamp); ++stamp; }
    }
    // final block output -> HBM
    {
        const int c8 = a.cout_last / 8, chunks = a.px_last * c8;
        uint4* dst = reinterpret_cast<uint4*>(a.out + (size_t)n * a.out_stride);
        for (int i = tid; i < chunks; i += TT) dst[i] = *reinterpret_cast<const uint4*>(X + (size_t)(i / c8) * XS + (i % c8) * 8);
    }
    TK_STAMP(63);
}

}  // namespace

bool trunk_block_supported(int cin, int cexp, int cout, int k, int stride, int hin, int hout, int sq) {
    return cin % 16 == 0 && cin <= 112 && cexp % 8 == 0 && cexp <= MAXC && cout % 8 == 0 && cout <= 128 - 16 && (k == 3 || k == 5) &&
           (stride == 1 || stride == 2) && hin * hin <= MAXPX && hout * hout <= 64 * NWV && sq <= MAXSQ && sq % 2 == 0 && cexp % 2 == 0;
}

size_t trunk_lds_bytes() {
    return ((size_t)MAXPX * XS + (size_t)(MAXPX + 1) * ES + 25 * CW) * sizeof(half_t) + ((size_t)CW + MAXC + MAXSQ + NWV * CW) * sizeof(float);
}

int launch_trunk(const TrunkArgs& a0, int n, hipStream_t s) {
    TrunkArgs a = a0;
    DN_REQUIRE(a.count >= 1 && a.count <= TRUNK_MAX_BLOCKS, "trunk: %d blocks", a.count);
    const size_t lds = trunk_lds_bytes();
    DN_REQUIRE(lds + sizeof(TrunkBlock) * TRUNK_MAX_BLOCKS <= 160 * 1024, "trunk: %zu B of LDS", lds);
    DN_HIP_CHECK(dn_allow_big_lds(reinterpret_cast<const void*>(trunk_kernel), (int)lds));
    a.stamps = g_trunk_stamps;
    dn_note_kernel("trunk_kernel");
    hipLaunchKernelGGL(trunk_kernel, dim3(a.xq > 0 ? 8 * a.xq : n), dim3(TT), lds, s, a, n);
    return DN_OK;
}
