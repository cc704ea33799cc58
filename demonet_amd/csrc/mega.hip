// Per-image "tail" kernel: one 512-thread workgroup owns ONE image and runs a whole run of consecutive small ops
// (pointwise / depthwise / squeeze-excite at <= 20x20 resolution) back to back, with only __syncthreads() between
// layers.
//
// Why: at batch 64 the low-resolution tail of SSDLite (everything after the last stride-2 stage, the extras and the
// heads of levels 1..5; mobilenetv3.py:198-214 rows 8-15, ssd_mobilenetv3.py:39-54,65-95) is ~65 launches of a few
// hundred workgroups each. In-kernel stamps showed each of them paying 1 us per dependent memory round trip plus a
// grid fill/drain; the data itself (<= 0.5 MB per image per tensor) never leaves L2. Images are independent
// (BN in eval mode, per-image post-process), so a workgroup can walk the layers of its own image without any
// inter-workgroup synchronisation: every tensor address is written once and read afterwards by the same CU, stores are
// drained by the barrier's fence, nothing is ever stale.
//
// GEMMs: v_mfma_f32_32x32x16_f16, A = weight rows, B = pixel rows, both read straight from global/L2 as 16-byte
// fragments (K-contiguous in NHWC and in the [cout][cin] weight layout), 2x2 register blocking per wave, 4 k-steps of
// loads issued before their MFMAs. Depthwise: thread = (pixel row, 8 channels). SE: pooled sums are exact per-image
// sums in a fixed order (deterministic), FCs lane-parallel.
#include "common.h"

namespace {

constexpr int MEGA_THREADS = 512;
constexpr int MEGA_WAVES = MEGA_THREADS / 64;

__device__ __forceinline__ void mega_pw(const MegaOp& o, const half_t* __restrict__ x, const half_t* __restrict__ w,
                                        const float* __restrict__ bias, const half_t* res, const float* se, void* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int HW = o.hout * o.wout, K = o.cin, N = o.cout;
    const int KS = (K + 15) >> 4;
    const int MT = (HW + 31) >> 5, NT = (N + 31) >> 5;
    const int MP = (MT + 1) >> 1, NP = (NT + 1) >> 1;
    for (int item = wave; item < MP * NP; item += MEGA_WAVES) {
        const int mp = item / NP, np = item - mp * NP;
        int m[2], n[2];
        bool mv[2], nv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            m[j] = (2 * mp + j) * 32 + r;
            mv[j] = m[j] < HW;
            n[j] = (2 * np + j) * 32 + r;
            nv[j] = n[j] < N;
        }
        floatx16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int ks0 = 0; ks0 < KS; ks0 += 4) {
            half8 xf[2][4], wf[2][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = (ks0 + u) * 16 + hh * 8;
                const bool kv = k < K;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    half8 t = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (kv && mv[j]) t = *reinterpret_cast<const half8*>(x + (size_t)m[j] * K + k);
                    xf[j][u] = t;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    half8 t = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (kv && nv[i]) t = *reinterpret_cast<const half8*>(w + (size_t)n[i] * K + k);
                    wf[i][u] = t;
                }
            }
            if (se) {        // SE scale on the input channels (only when the producer could not rescale in place)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = (ks0 + u) * 16 + hh * 8;
                    if (k < K) {
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int e = 0; e < 8; ++e) xf[j][u][e] = (half_t)((float)xf[j][u][e] * se[k + e]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (ks0 + u < KS) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i][u], xf[j][u], acc[i][j], 0, 0, 0);
                }
            }
        }
        // epilogue: residual loads first (no load may sit behind a possibly-aliasing store), then math + stores
        half4 resv[2][2][4];
        if (res) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c0 = (2 * np + i) * 32 + 8 * g + 4 * hh;
                        half4 rv = {0, 0, 0, 0};
                        if (mv[j] && c0 < N) rv = *reinterpret_cast<const half4*>(res + (size_t)m[j] * N + c0);
                        resv[j][i][g] = rv;
                    }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = (2 * np + i) * 32 + 8 * g + 4 * hh;
                if (c0 >= N) continue;
                float bv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = (c0 + e < N) ? bias[c0 + e] : 0.f;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (!mv[j]) continue;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = dn_act(acc[i][j][4 * g + e] + bv[e], o.act);
                    if (o.out_fp32) {
                        float* op = reinterpret_cast<float*>(out) + (size_t)m[j] * N + c0;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (c0 + e < N) op[e] = v[e];
                    } else {
                        if (res) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += (float)resv[j][i][g][e];
                        }
                        half4 hv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) hv[e] = (half_t)v[e];
                        *reinterpret_cast<half4*>(reinterpret_cast<half_t*>(out) + (size_t)m[j] * N + c0) = hv;
                    }
                }
            }
        }
    }
}

template <int K>
__device__ __forceinline__ void mega_dw_k(const MegaOp& o, const half_t* __restrict__ x, const half_t* __restrict__ w,
                                          const float* __restrict__ bias, half_t* __restrict__ out, float* pool_rows, float* red) {
    const int C = o.cin, C8 = C >> 3;
    const int R = MEGA_THREADS / C8;                // pixel rows of threads (C8 <= 512)
    const int cg = threadIdx.x % C8, prow = threadIdx.x / C8;
    const bool active = prow < R;
    const int c0 = cg * 8;
    const int P = o.hout * o.wout;
    float psum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (active) {
        float b[8];
        {
            const float4 b0 = *reinterpret_cast<const float4*>(bias + c0), b1 = *reinterpret_cast<const float4*>(bias + c0 + 4);
            b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
        }
        for (int p = prow; p < P; p += R) {
            const int oy = p / o.wout, ox = p - oy * o.wout;
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = b[e];
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int iy = oy * o.stride - o.pad + ky;
                if (iy < 0 || iy >= o.hin) continue;
                half8 xv[K], wv[K];
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const int ix = ox * o.stride - o.pad + kx;
                    half8 t = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (ix >= 0 && ix < o.win) t = *reinterpret_cast<const half8*>(x + ((size_t)iy * o.win + ix) * C + c0);
                    xv[kx] = t;
                    wv[kx] = *reinterpret_cast<const half8*>(w + (size_t)(ky * K + kx) * C + c0);
                }
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += (float)xv[kx][e] * (float)wv[kx][e];
            }
            half8 hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = dn_act(acc[e], o.act);
                psum[e] += v;
                hv[e] = (half_t)v;
            }
            *reinterpret_cast<half8*>(out + (size_t)p * C + c0) = hv;
        }
    }
    if (o.pool >= 0) {
        // exact per-image channel sums in a fixed order; stored as partial-sum row 0 (rows 1.. zero) so that both the
        // stand-alone se_fc_kernel and mega_se read them the same way
#pragma unroll
        for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = psum[e];
        __syncthreads();
        if (threadIdx.x < C8) {
            float t8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int rr = 0; rr < R; ++rr)
#pragma unroll
                for (int e = 0; e < 8; ++e) t8[e] += red[(rr * C8 + threadIdx.x) * 8 + e];
#pragma unroll
            for (int e = 0; e < 8; ++e) pool_rows[c0 + e] = t8[e];
            for (int b2 = 1; b2 < o.pool_rows; ++b2)
#pragma unroll
                for (int e = 0; e < 8; ++e) pool_rows[(size_t)b2 * C + c0 + e] = 0.f;
        }
    }
}

// (sum of partial rows)/pixels -> fc1 -> ReLU -> fc2 -> Hardsigmoid (mobilenetv3.py:31-36); scale written to global;
// optionally the producing depthwise output is rescaled in place (x *= s, rounded to fp16 exactly like the on-load
// scaling of the stand-alone pointwise kernel) so that the following in-group projection needs no per-fragment multiply.
__device__ __forceinline__ void mega_se(const MegaOp& o, const float* __restrict__ partial, const float* __restrict__ w1,
                                        const float* __restrict__ b1, const float* __restrict__ w2,
                                        const float* __restrict__ b2, float* __restrict__ scale, half_t* rescale_x,
                                        float* sh) {
    constexpr int MAXC64 = 16, MAXS64 = 4;
    const int c = o.cin, sq = o.squeeze;
    float* mean = sh;
    float* z = sh + c;
    float* sc = z + sq;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_pixels = 1.0f / (float)o.pool_pixels;
    for (int i = threadIdx.x; i < c; i += MEGA_THREADS) {
        float t = 0.f;
        for (int b = 0; b < o.pool_rows; ++b) t += partial[(size_t)b * c + i];
        mean[i] = t * inv_pixels;
    }
    __syncthreads();
    for (int j0 = wave * 4; j0 < sq; j0 += MEGA_WAVES * 4) {
        float wv[4][MAXC64];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < MAXC64; ++q) {
                const int i = lane + 64 * q;
                wv[u][q] = (j0 + u < sq && i < c) ? w1[(size_t)(j0 + u) * c + i] : 0.f;
            }
        float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < MAXC64; ++q) {
            const int i = lane + 64 * q;
            const float mm = (i < c) ? mean[i] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] += wv[u][q] * mm;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) t[u] += __shfl_xor(t[u], d);
            if (lane == 0 && j0 + u < sq) z[j0 + u] = fmaxf(t[u] + b1[j0 + u], 0.f);
        }
    }
    __syncthreads();
    for (int i0 = wave * 16; i0 < c; i0 += MEGA_WAVES * 16) {
        float wv[16][MAXS64];
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int q = 0; q < MAXS64; ++q) {
                const int j = lane + 64 * q;
                wv[u][q] = (i0 + u < c && j < sq) ? w2[(size_t)(i0 + u) * sq + j] : 0.f;
            }
        float zz[MAXS64];
#pragma unroll
        for (int q = 0; q < MAXS64; ++q) zz[q] = (lane + 64 * q < sq) ? z[lane + 64 * q] : 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < MAXS64; ++q) t += wv[u][q] * zz[q];
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) t += __shfl_xor(t, d);
            if (lane == 0 && i0 + u < c) {
                const float s = fminf(fmaxf(t + b2[i0 + u] + 3.f, 0.f), 6.f) * (1.f / 6.f);
                sc[i0 + u] = s;
                scale[i0 + u] = s;
            }
        }
    }
    if (rescale_x) {
        __syncthreads();
        const int C8 = c >> 3;
        const int total = o.pool_pixels * C8;
        for (int idx = threadIdx.x; idx < total; idx += MEGA_THREADS) {
            const int cg = idx % C8;
            half8 v = *reinterpret_cast<half8*>(rescale_x + (size_t)idx * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * sc[cg * 8 + e]);
            *reinterpret_cast<half8*>(rescale_x + (size_t)idx * 8) = v;
        }
    }
}

__global__ __launch_bounds__(MEGA_THREADS) void mega_kernel(const MegaOp* __restrict__ ops, int first, int count,
                                                           unsigned char* __restrict__ ws, const unsigned char* __restrict__ wts) {
    extern __shared__ __attribute__((aligned(16))) float msh[];      // dw reduction [512][8] | SE vectors
    const int img = blockIdx.x;
    for (int q = first; q < first + count; ++q) {
        const MegaOp o = ops[q];
        const half_t* x = reinterpret_cast<const half_t*>(ws + o.x_off + (size_t)img * o.x_stride);
        void* out = ws + o.out_off + (size_t)img * o.out_stride;
        if (o.type == DN_OP_PW) {
            const half_t* res = o.res_off >= 0 ? reinterpret_cast<const half_t*>(ws + o.res_off + (size_t)img * o.res_stride) : nullptr;
            const float* se = o.se_off >= 0 ? reinterpret_cast<const float*>(ws + o.se_off + (size_t)img * o.se_stride) : nullptr;
            mega_pw(o, x, reinterpret_cast<const half_t*>(wts + o.w_off), reinterpret_cast<const float*>(wts + o.b_off), res, se, out);
        } else if (o.type == DN_OP_DW) {
            float* pool = o.pool >= 0 ? reinterpret_cast<float*>(ws + o.pool_off + (size_t)img * o.pool_stride) : nullptr;
            const half_t* w = reinterpret_cast<const half_t*>(wts + o.w_off);
            const float* b = reinterpret_cast<const float*>(wts + o.b_off);
            if (o.k == 3) mega_dw_k<3>(o, x, w, b, reinterpret_cast<half_t*>(out), pool, msh);
            else mega_dw_k<5>(o, x, w, b, reinterpret_cast<half_t*>(out), pool, msh);
        } else if (o.type == DN_OP_SE) {
            half_t* rx = o.res_off >= 0 ? reinterpret_cast<half_t*>(ws + o.res_off + (size_t)img * o.res_stride) : nullptr;
            mega_se(o, reinterpret_cast<const float*>(x), reinterpret_cast<const float*>(wts + o.w_off),
                    reinterpret_cast<const float*>(wts + o.b_off), reinterpret_cast<const float*>(wts + o.w2_off),
                    reinterpret_cast<const float*>(wts + o.b2_off), reinterpret_cast<float*>(out), rx, msh);
        }
        __syncthreads();        // includes the fence that drains this layer's stores before the next layer reads them
    }
}

}  // namespace

int launch_mega(const MegaOp* ops_dev, int first, int count, int n, unsigned char* ws, const unsigned char* wts, hipStream_t s) {
    dn_note_kernel("mega_kernel");
    hipLaunchKernelGGL(mega_kernel, dim3(n), dim3(MEGA_THREADS), MEGA_THREADS * 8 * sizeof(float), s, ops_dev, first, count, ws, wts);
    return DN_OK;
}
