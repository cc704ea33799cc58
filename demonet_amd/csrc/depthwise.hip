// Depthwise kxk convolution (+folded BN bias, activation), squeeze-excite pooling / FCs, stem conv.
//
// reference ops replaced: depthwise ConvBNActivation (mobilenetv3.py:81, ssd_mobilenetv3.py:31,48),
//   SqueezeExcitation (mobilenetv3.py:22-40), stem conv (mobilenetv3.py:141) with the input normalisation of
//   transform.py:129-138 applied on load.
// All of these are HBM-bound streaming kernels: NHWC fp16, 16-byte (8-channel) vectors per lane,
// channel index fastest across lanes so every wave instruction touches whole 128-B lines, fp32 accumulation.
#include "common.h"

namespace {

// each thread: TW consecutive output pixels of one row x 8 channels; sliding input window kept in registers
template <int K, int S, int TW>
__global__ __launch_bounds__(256) void dw_kernel(DwArgs a) {
    constexpr int NIN = (TW - 1) * S + K;
    const int C8 = a.c >> 3;
    const int XS = (a.wo + TW - 1) / TW;
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(idx % C8);
    idx /= C8;
    const int xs = (int)(idx % XS);
    idx /= XS;
    const int oy = (int)(idx % a.ho);
    const int n = (int)(idx / a.ho);
    if (n >= a.n) return;
    const int ox0 = xs * TW;
    const int c0 = cg * 8;

    float acc[TW][8];
    {
        const float4 b0 = *reinterpret_cast<const float4*>(a.bias + c0);
        const float4 b1 = *reinterpret_cast<const float4*>(a.bias + c0 + 4);
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            acc[t][0] = b0.x; acc[t][1] = b0.y; acc[t][2] = b0.z; acc[t][3] = b0.w;
            acc[t][4] = b1.x; acc[t][5] = b1.y; acc[t][6] = b1.z; acc[t][7] = b1.w;
        }
    }
    const int ix0 = ox0 * S - a.pad;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        const int iy = oy * S - a.pad + ky;
        if (iy < 0 || iy >= a.h) continue;
        half8 wv[K];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) wv[kx] = *reinterpret_cast<const half8*>(a.w + (size_t)(ky * K + kx) * a.c + c0);
        half8 xin[NIN];
        const half_t* rowp = a.x + ((size_t)(n * a.h + iy) * a.w_) * a.c + c0;
#pragma unroll
        for (int i = 0; i < NIN; ++i) {
            const int ix = ix0 + i;
            half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ix >= 0 && ix < a.w_) v = *reinterpret_cast<const half8*>(rowp + (size_t)ix * a.c);
            xin[i] = v;
        }
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[t][e] += (float)xin[t * S + kx][e] * (float)wv[kx][e];
    }
    half_t* orow = a.out + ((size_t)(n * a.ho + oy) * a.wo) * a.c + c0;
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        if (ox0 + t >= a.wo) break;
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)dn_act(acc[t][e], a.act);
        *reinterpret_cast<half8*>(orow + (size_t)(ox0 + t) * a.c) = o;
    }
}

template <int K, int S, int TW>
int launch_dw(const DwArgs& a, hipStream_t s) {
    const long threads = (long)a.n * a.ho * ((a.wo + TW - 1) / TW) * (a.c / 8);
    hipLaunchKernelGGL((dw_kernel<K, S, TW>), dim3(dn_cdiv(threads, 256)), dim3(256), 0, s, a);
    return DN_OK;
}

// ---- SE: per-(image, channel) sums over the spatial map; deterministic tree order -----------------------
__global__ __launch_bounds__(256) void se_pool_kernel(const half_t* __restrict__ x, float* __restrict__ sums, int hw, int c) {
    __shared__ float red[16][16][8];
    const int C8 = c >> 3;
    const int cgl = threadIdx.x & 15, slot = threadIdx.x >> 4;
    const int cg = blockIdx.x * 16 + cgl;
    const int n = blockIdx.y;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (cg < C8) {
        const half_t* p = x + (size_t)n * hw * c + cg * 8;
        for (int px = slot; px < hw; px += 16) {
            const half8 v = *reinterpret_cast<const half8*>(p + (size_t)px * c);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[slot][cgl][e] = s[e];
    __syncthreads();
    if (slot == 0 && cg < C8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = 0.f;
            for (int k = 0; k < 16; ++k) t += red[k][cgl][e];
            sums[(size_t)n * c + cg * 8 + e] = t;
        }
    }
}

// fc1 (+bias) -> ReLU -> fc2 (+bias) -> Hardsigmoid = relu6(x+3)/6      (mobilenetv3.py:31-36)
// w1t: [c][squeeze], w2t: [squeeze][c] (transposed at plan time so lanes read consecutive addresses)
__global__ __launch_bounds__(256) void se_fc_kernel(const float* __restrict__ sums, const float* __restrict__ w1t,
                                                   const float* __restrict__ b1, const float* __restrict__ w2t,
                                                   const float* __restrict__ b2, float* __restrict__ scale,
                                                   int c, int sq, float inv_pixels) {
    extern __shared__ float sh[];      // mean[c] then z[sq]
    float* mean = sh;
    float* z = sh + c;
    const int n = blockIdx.x;
    for (int i = threadIdx.x; i < c; i += 256) mean[i] = sums[(size_t)n * c + i] * inv_pixels;
    __syncthreads();
    for (int j = threadIdx.x; j < sq; j += 256) {
        float t = b1[j];
        for (int i = 0; i < c; ++i) t += w1t[(size_t)i * sq + j] * mean[i];
        z[j] = fmaxf(t, 0.f);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < c; i += 256) {
        float t = b2[i];
        for (int j = 0; j < sq; ++j) t += w2t[(size_t)j * c + i] * z[j];
        scale[(size_t)n * c + i] = fminf(fmaxf(t + 3.f, 0.f), 6.f) * (1.f / 6.f);
    }
}

// ---- stem: dense kxk conv on the NCHW fp32 image, normalisation on load, NHWC fp16 out ------------------
template <int COUT, int K>
__global__ __launch_bounds__(256) void stem_kernel(StemArgs a) {
    __shared__ float wsh[K * K * 3 * COUT];
    __shared__ float bsh[COUT];
    for (int i = threadIdx.x; i < K * K * 3 * COUT; i += 256) wsh[i] = a.w[i];
    for (int i = threadIdx.x; i < COUT; i += 256) bsh[i] = a.bias[i];
    __syncthreads();
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int ox = (int)(idx % a.wo);
    idx /= a.wo;
    const int oy = (int)(idx % a.ho);
    const int n = (int)(idx / a.ho);
    if (n >= a.n) return;
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = bsh[o];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* plane = a.img + ((size_t)n * 3 + c) * a.h * a.w_;
        const float mean = a.mean[c], sd = a.std[c];
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int iy = oy * a.stride - a.pad + ky;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int ix = ox * a.stride - a.pad + kx;
                float v = 0.f;      // zero padding is applied to the NORMALISED image (transform then conv)
                if (iy >= 0 && iy < a.h && ix >= 0 && ix < a.w_) v = (plane[(size_t)iy * a.w_ + ix] - mean) / sd;
                const float* wp = &wsh[((c * K + ky) * K + kx) * COUT];
#pragma unroll
                for (int o = 0; o < COUT; ++o) acc[o] += v * wp[o];
            }
        }
    }
    half_t* op = a.out + ((size_t)(n * a.ho + oy) * a.wo + ox) * COUT;
#pragma unroll
    for (int o8 = 0; o8 < COUT / 8; ++o8) {
        half8 hv;
#pragma unroll
        for (int e = 0; e < 8; ++e) hv[e] = (half_t)dn_act(acc[o8 * 8 + e], a.act);
        *reinterpret_cast<half8*>(op + o8 * 8) = hv;
    }
}

template <int COUT, int K>
int launch_stem_t(const StemArgs& a, hipStream_t s) {
    const long threads = (long)a.n * a.ho * a.wo;
    hipLaunchKernelGGL((stem_kernel<COUT, K>), dim3(dn_cdiv(threads, 256)), dim3(256), 0, s, a);
    return DN_OK;
}

}  // namespace

int launch_depthwise(const DwArgs& a, hipStream_t s) {
    DN_REQUIRE(a.c % 8 == 0, "depthwise: c=%d must be a multiple of 8", a.c);
    if (a.k == 3 && a.stride == 1) return launch_dw<3, 1, 4>(a, s);
    if (a.k == 3 && a.stride == 2) return launch_dw<3, 2, 2>(a, s);
    if (a.k == 5 && a.stride == 1) return launch_dw<5, 1, 4>(a, s);
    if (a.k == 5 && a.stride == 2) return launch_dw<5, 2, 2>(a, s);
    dn_set_error("depthwise: unsupported k=%d stride=%d", a.k, a.stride);
    return DN_E_UNSUPPORTED;
}

int launch_se_pool(const half_t* x, float* sums, int n, int hw, int c, hipStream_t s) {
    hipLaunchKernelGGL(se_pool_kernel, dim3(dn_cdiv(c / 8, 16), n), dim3(256), 0, s, x, sums, hw, c);
    return DN_OK;
}

int launch_se_fc(const float* sums, const float* w1t, const float* b1, const float* w2t, const float* b2, float* scale,
                 int n, int c, int squeeze, int pool_pixels, hipStream_t s) {
    hipLaunchKernelGGL(se_fc_kernel, dim3(n), dim3(256), (size_t)(c + squeeze) * sizeof(float), s, sums, w1t, b1, w2t, b2,
                       scale, c, squeeze, 1.0f / (float)pool_pixels);
    return DN_OK;
}

int launch_stem(const StemArgs& a, hipStream_t s) {
    if (a.k == 3 && a.cout == 16) return launch_stem_t<16, 3>(a, s);
    if (a.k == 3 && a.cout == 32) return launch_stem_t<32, 3>(a, s);
    if (a.k == 3 && a.cout == 64) return launch_stem_t<64, 3>(a, s);
    dn_set_error("stem: unsupported k=%d cout=%d", a.k, a.cout);
    return DN_E_UNSUPPORTED;
}
