// Fused inverted-residual block:  [1x1 expand + BN + act] -> depthwise kxk + BN + act -> [1x1 project + BN (+ residual)]
// in ONE kernel, the expanded tensor never leaving the CU.
//
// reference ops replaced: InvertedResidual.forward (mobilenetv3.py:61-99, mobilenetv2.py:62-100, backbone.py:81-119) and
// _extra_block (ssd_mobilenetv3.py:39-54) -- expand ConvBNActivation, depthwise ConvBNActivation, project ConvBNActivation,
// `result += input`. For SE blocks the project cannot be fused (the squeeze needs the whole map): the kernel then stops
// after the depthwise stage, writes its output and the per-tile pooled sums (SqueezeExcitation, mobilenetv3.py:31-33).
//
// Why: layer by layer, the expanded tensor (4-6x the block input) is written and read twice; at batch 64 the 160^2/80^2
// blocks alone move ~1 GB that way (40.3 MB/img layer-isolated). Here a workgroup owns a TH x TW tile of
// OUTPUT pixels of one image:
//   1. the input tile with its halo is staged once in LDS (all loads in flight together);
//   2. the expanded channels are produced 32 at a time: MFMA (A = 32 weight rows, B = input pixels) -> BN/act -> LDS,
//      zeroed outside the image (the depthwise conv pads the EXPANDED tensor with zeros);
//   3. the depthwise stage reads that 32-channel slab from LDS (16-B vectors, fp32 accumulate) and leaves its output in
//      LDS as the B operand of
//   4. the projection MFMA, whose accumulators (<= 4 32x32 tiles per wave) sum over the 32-channel chunks -- the chunk
//      loop IS the projection's K loop, in the same k order as the unfused kernel;
//   5. epilogue: bias, residual taken from the staged input tile (no second read), fp16, LDS-staged row-contiguous
//      16-byte stores.
// Rounding points are the same as layer by layer (expanded and depthwise activations are fp16 in both).
#include "common.h"

static long long* g_fu_stamps = nullptr;     // dev hook: per-workgroup phase stamps
extern "C" __attribute__((visibility("default"))) void dn_debug_fused_stamps(void* dev_ptr) { g_fu_stamps = (long long*)dev_ptr; }
#define FU_STAMP(k) do { if (a.stamps && threadIdx.x == 0) a.stamps[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)

namespace {

constexpr int FT = 256;        // threads

template <int TH, int TW, int K, int MAXT>
__global__ __launch_bounds__(FT, 3) void fused_kernel(FusedArgs a) {
    constexpr int P = TH * TW;
    constexpr int PP = (P + 31) / 32 * 32;
    constexpr int MTO = PP / 32;
    constexpr int KK = K * K;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int n = blockIdx.z;
    const int oy0 = blockIdx.y * TH, ox0 = blockIdx.x * TW;
    const int S = a.stride;
    const int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
    const int IHW = IH * IW, IHWP = (IHW + 31) / 32 * 32;
    const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;
    const bool has_exp = a.w1 != nullptr, has_proj = a.w3 != nullptr;
    const int CH = a.ch;                        // channels per chunk: 32 (16 when the depthwise has only 16 channels)
    const int EW = CH + 8;                      // slab row width (halfs): odd number of 16-B slots
    const int XW = a.xw;                        // staged input row width (halfs)
    const int NTO = (a.cout + 31) >> 5;
    const int OW = NTO * 32 + 8;

    // LDS carve (mirrored by fused_lds): input tile | expanded slab (aliased by the output staging) | depthwise slab |
    // depthwise weights x2 | depthwise bias x2 | validity bytes | pooling scratch
    half_t* Xs = reinterpret_cast<half_t*>(smem);
    half_t* Es = Xs + (size_t)IHWP * XW;
    const size_t es_halfs = has_exp ? (size_t)IHWP * EW : 0;
    const size_t os_halfs = has_proj ? (size_t)PP * OW : 0;
    half_t* Os = Es;
    half_t* Ds = Es + (es_halfs > os_halfs ? es_halfs : os_halfs);
    half_t* W1s = Ds + (has_proj ? (size_t)PP * EW : 0);            // [2][32][XW]   expand weights of a chunk
    const size_t w1_halfs = has_exp ? (size_t)32 * XW : 0;
    half_t* W3s = W1s + 2 * w1_halfs;                               // [2][NTO*32][EW] project weights of a chunk
    const size_t w3_halfs = has_proj ? (size_t)NTO * 32 * EW : 0;
    half_t* Wds = W3s + 2 * w3_halfs;                               // [2][KK][32]
    float* bds = reinterpret_cast<float*>(Wds + 2 * KK * 32);       // [2][32] depthwise bias
    float* b1s = bds + 64;                                          // [2][32] expand bias
    unsigned char* valid = reinterpret_cast<unsigned char*>(b1s + 64);
    float* red = reinterpret_cast<float*>(valid + IHWP + ((16 - (IHWP & 15)) & 15));

    const int KS1 = (a.cin + 15) >> 4;
    const int CH8 = CH >> 3;
    const int npairs = MTO * NTO;

    // ---- weight prefetch: global -> a few registers per thread (one chunk ahead) -> the LDS buffers of that chunk
    const int X8 = XW >> 3;                                         // 16-B chunks per expand-weight row (last = pad)
    const int Q3 = CH8 + 1;                                         // 16-B chunks per project-weight row (last = pad)
    const int n1 = has_exp ? 32 * X8 : 0;                           // <= 512 (cin <= 120)
    const int n3 = has_proj ? NTO * 32 * Q3 : 0;                    // <= 480 (cout <= 96)
    uint4 pw1[2], pw3[2], pwd;
    float4 pbd;
    auto fetch = [&](int c0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c = tid + u * FT;
            uint4 t = make_uint4(0, 0, 0, 0);
            if (c < n1) {
                const int row = c / X8, q = c - row * X8;
                if (row < CH && c0 + row < a.cexp && q * 8 < a.cin) t = *reinterpret_cast<const uint4*>(a.w1 + (size_t)(c0 + row) * a.cin + q * 8);
            }
            pw1[u] = t;
            uint4 t3 = make_uint4(0, 0, 0, 0);
            if (c < n3) {
                const int row = c / Q3, q = c - row * Q3;
                if (row < a.cout && q < CH8 && c0 + q * 8 < a.cexp) t3 = *reinterpret_cast<const uint4*>(a.w3 + (size_t)row * a.cexp + c0 + q * 8);
            }
            pw3[u] = t3;
        }
        pwd = make_uint4(0, 0, 0, 0);
        pbd = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < KK * CH8) {
            const int tap = tid / CH8, cg = tid - tap * CH8;
            if (c0 + cg * 8 < a.cexp) pwd = *reinterpret_cast<const uint4*>(a.wd + (size_t)tap * a.cexp + c0 + cg * 8);
        }
        if (tid < CH / 4 && c0 + tid * 4 < a.cexp) pbd = *reinterpret_cast<const float4*>(a.bd + c0 + tid * 4);
        // threads 64.. carry the expand bias (a different wave than the depthwise bias: no extra register)
        if (has_exp && tid >= 64 && tid < 64 + CH / 4 && c0 + (tid - 64) * 4 < a.cexp)
            pbd = *reinterpret_cast<const float4*>(a.b1 + c0 + (tid - 64) * 4);
    };
    auto stash = [&](int bf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c = tid + u * FT;
            if (c < n1) *reinterpret_cast<uint4*>(&W1s[(size_t)bf * w1_halfs + (size_t)(c / X8) * XW + (c % X8) * 8]) = pw1[u];
            if (c < n3) *reinterpret_cast<uint4*>(&W3s[(size_t)bf * w3_halfs + (size_t)(c / Q3) * EW + (c % Q3) * 8]) = pw3[u];
        }
        if (tid < KK * CH8) {
            const int tap = tid / CH8, cg = tid - tap * CH8;
            *reinterpret_cast<uint4*>(&Wds[(size_t)bf * KK * 32 + tap * CH + cg * 8]) = pwd;
        }
        if (tid < CH / 4) *reinterpret_cast<float4*>(&bds[bf * 32 + tid * 4]) = pbd;
        if (has_exp && tid >= 64 && tid < 64 + CH / 4) *reinterpret_cast<float4*>(&b1s[bf * 32 + (tid - 64) * 4]) = pbd;
    };
    FU_STAMP(0);
    fetch(0);

    // ---- 1. stage the input tile (+halo), zero outside the image / beyond cin / beyond IHW
    {
        const int XC = XW >> 3;                 // 16-B chunks per staged row (the last one is padding)
        const int total = IHWP * XC;
        const half_t* xin = a.x + (size_t)n * a.H * a.W * a.cin;
        for (int c0 = 0; c0 < total; c0 += FT * 4) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u * FT + tid;
                uint4 t = make_uint4(0, 0, 0, 0);
                if (c < total) {
                    const int row = c / XC, q = c - row * XC;
                    const int ly = row / IW, lx = row - ly * IW;
                    const int iy = iy0 + ly, ix = ix0 + lx;
                    if (row < IHW && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && q * 8 < a.cin)
                        t = *reinterpret_cast<const uint4*>(xin + ((size_t)iy * a.W + ix) * a.cin + q * 8);
                }
                v[u] = t;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u * FT + tid;
                if (c < total) {
                    const int row = c / XC, q = c - row * XC;
                    *reinterpret_cast<uint4*>(&Xs[(size_t)row * XW + q * 8]) = v[u];
                }
            }
        }
        for (int row = tid; row < IHWP; row += FT) {
            const int ly = row / IW, lx = row - ly * IW;
            const int iy = iy0 + ly, ix = ix0 + lx;
            valid[row] = (row < IHW && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? 1 : 0;
        }
    }
    stash(0);
    __syncthreads();
    FU_STAMP(1);

    floatx16 acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    float psum[8];

    int buf = 0;
    for (int c0 = 0; c0 < a.cexp; c0 += CH, buf ^= 1) {
        const bool more = c0 + CH < a.cexp;
        if (more) fetch(c0 + CH);               // in flight during this chunk's compute
        // ---- 2. expand this chunk: Es[pixel][0..CH) = act1(Xs . W1[c0..c0+CH)^T + b1) * valid
        if (has_exp) {
            // no global load inside a chunk: s_waitcnt vmcnt is in-order, a load here would also wait for the
            // next chunk's weight prefetch issued above
            float4 bv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (8 * g + 4 * hh < CH) t = *reinterpret_cast<const float4*>(&b1s[buf * 32 + 8 * g + 4 * hh]);
                bv[g] = t;
            }
            const half_t* w1b = W1s + (size_t)buf * w1_halfs + (size_t)r * XW + hh * 8;
            for (int mt = wave; mt < IHWP / 32; mt += 4) {
                floatx16 e16;
#pragma unroll
                for (int e = 0; e < 16; ++e) e16[e] = 0.f;
                const half_t* xr = &Xs[(size_t)(mt * 32 + r) * XW + hh * 8];
                for (int ks = 0; ks < KS1; ++ks) {
                    const half8 wf = *reinterpret_cast<const half8*>(w1b + ks * 16);
                    const half8 xf = *reinterpret_cast<const half8*>(xr + ks * 16);
                    e16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xf, e16, 0, 0, 0);
                }
                const float vm = valid[mt * 32 + r] ? 1.f : 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = 8 * g + 4 * hh;
                    if (cl < CH) {
                        half4 hv;
                        hv[0] = (half_t)(dn_act(e16[4 * g + 0] + bv[g].x, a.act1) * vm);
                        hv[1] = (half_t)(dn_act(e16[4 * g + 1] + bv[g].y, a.act1) * vm);
                        hv[2] = (half_t)(dn_act(e16[4 * g + 2] + bv[g].z, a.act1) * vm);
                        hv[3] = (half_t)(dn_act(e16[4 * g + 3] + bv[g].w, a.act1) * vm);
                        *reinterpret_cast<half4*>(&Es[(size_t)(mt * 32 + r) * EW + cl]) = hv;
                    }
                }
            }
            __syncthreads();
            if (c0 == 0) FU_STAMP(2);
        }
        // ---- 3. depthwise on the slab -> Ds (and global / pooled sums when there is no projection)
        {
            const half_t* Sb = has_exp ? Es : (Xs + c0);
            const int SW = has_exp ? EW : XW;
            const half_t* wb = Wds + (size_t)buf * KK * 32;
            const float* bb = bds + buf * 32;
#pragma unroll
            for (int e = 0; e < 8; ++e) psum[e] = 0.f;
            for (int it = tid; it < PP * CH8; it += FT) {
                const int cg = it % CH8, p = it / CH8;
                const int oy = p / TW, ox = p - oy * TW;
                const int cglob = c0 + cg * 8;
                const bool pix_ok = p < P && (oy0 + oy) < a.Ho && (ox0 + ox) < a.Wo;
                half8 o8 = {0, 0, 0, 0, 0, 0, 0, 0};
                if (pix_ok && cglob < a.cexp) {
                    float d[8];
                    {
                        const float4 b0 = *reinterpret_cast<const float4*>(bb + cg * 8), b1 = *reinterpret_cast<const float4*>(bb + cg * 8 + 4);
                        d[0] = b0.x; d[1] = b0.y; d[2] = b0.z; d[3] = b0.w; d[4] = b1.x; d[5] = b1.y; d[6] = b1.z; d[7] = b1.w;
                    }
                    const half_t* s0 = Sb + (size_t)((oy * S) * IW + ox * S) * SW + cg * 8;
#pragma unroll 1
                    for (int ky = 0; ky < K; ++ky) {            // one tap row at a time: bounds the live LDS reads
#pragma unroll
                        for (int kx = 0; kx < K; ++kx) {
                            const half8 xv = *reinterpret_cast<const half8*>(s0 + (size_t)(ky * IW + kx) * SW);
                            const half8 wv = *reinterpret_cast<const half8*>(wb + (ky * K + kx) * CH + cg * 8);
#pragma unroll
                            for (int e = 0; e < 8; ++e) d[e] += (float)xv[e] * (float)wv[e];
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float v = dn_act(d[e], a.act2);
                        psum[e] += v;
                        o8[e] = (half_t)v;
                    }
                    if (!has_proj)
                        *reinterpret_cast<half8*>(a.out + (((size_t)n * a.Ho + oy0 + oy) * a.Wo + ox0 + ox) * a.cexp + cglob) = o8;
                }
                if (has_proj) *reinterpret_cast<half8*>(&Ds[(size_t)p * EW + cg * 8]) = o8;
            }
            if (a.pool) {
                // per-tile channel sums in a fixed order: lanes with equal channel group sit CH8 apart inside a wave ->
                // butterfly over the lane bits above CH8, then the 4 wave partials are added in wave order
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = psum[e];
                    for (int d = 32; d >= CH8; d >>= 1) v += __shfl_xor(v, d);
                    psum[e] = v;
                }
                if (lane < CH8) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) red[(wave * 4 + lane) * 8 + e] = psum[e];
                }
                __syncthreads();
                if (tid < CH8 * 8) {
                    const int cg = tid >> 3, e = tid & 7;
                    const float t = ((red[(0 * 4 + cg) * 8 + e] + red[(1 * 4 + cg) * 8 + e]) + red[(2 * 4 + cg) * 8 + e]) + red[(3 * 4 + cg) * 8 + e];
                    const int cglob = c0 + cg * 8;
                    if (cglob < a.cexp)
                        a.pool[((size_t)n * gridDim.y * gridDim.x + blockIdx.y * gridDim.x + blockIdx.x) * a.cexp + cglob + e] = t;
                }
            }
            if (c0 == 0) FU_STAMP(3);
            if (more) stash(buf ^ 1);           // other buffers: last read during the previous chunk
            __syncthreads();
            if (c0 == 0) FU_STAMP(4);
        }
        // ---- 4. projection partial sums over this chunk
        if (has_proj) {
            const half_t* w3b = W3s + (size_t)buf * w3_halfs;
#pragma unroll
            for (int t = 0; t < MAXT; ++t) {
                const int pi = wave + 4 * t;
                if (pi < npairs) {
                    const int mt = pi / NTO, nt = pi - mt * NTO;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        if (ks * 16 < CH) {
                            const half8 wf3 = *reinterpret_cast<const half8*>(w3b + (size_t)(nt * 32 + r) * EW + ks * 16 + hh * 8);
                            const half8 df = *reinterpret_cast<const half8*>(&Ds[(size_t)(mt * 32 + r) * EW + ks * 16 + hh * 8]);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf3, df, acc[t], 0, 0, 0);
                        }
                    }
                }
            }
            if (!has_exp) __syncthreads();      // the next chunk's depthwise overwrites Ds
        }
    }

    FU_STAMP(5);
    if (!has_proj) return;
    // ---- 5. epilogue: bias, residual from the staged input tile, fp16 -> Os (aliases the slab) -> row-contiguous stores
    if (has_exp) __syncthreads();               // every wave is done with Es/Ds before Os (== Es) is overwritten
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const int pi = wave + 4 * t;
        if (pi < npairs) {
            const int mt = pi / NTO, nt = pi - mt * NTO;
            const int p = mt * 32 + r;
            const int oy = p / TW, ox = p - oy * TW;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = nt * 32 + 8 * g + 4 * hh;
                float v[4] = {acc[t][4 * g + 0], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
                if (c < a.cout) {
                    const float4 b = *reinterpret_cast<const float4*>(a.b3 + c);
                    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = dn_act(v[e], a.act3);
                    if (a.has_res && p < P) {
                        const half4 rv = *reinterpret_cast<const half4*>(&Xs[(size_t)((oy + a.pad) * IW + ox + a.pad) * XW + c]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
                    }
                }
                half4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = (half_t)v[e];
                *reinterpret_cast<half4*>(&Os[(size_t)p * OW + c]) = hv;
            }
        }
    }
    __syncthreads();
    {
        const int OC8 = a.cout >> 3;
        for (int it = tid; it < P * OC8; it += FT) {
            const int q = it % OC8, p = it / OC8;
            const int oy = p / TW, ox = p - oy * TW;
            if ((oy0 + oy) < a.Ho && (ox0 + ox) < a.Wo)
                *reinterpret_cast<uint4*>(a.out + (((size_t)n * a.Ho + oy0 + oy) * a.Wo + ox0 + ox) * a.cout + q * 8) =
                    *reinterpret_cast<const uint4*>(&Os[(size_t)p * OW + q * 8]);
        }
    }
    FU_STAMP(6);
}

template <int TH, int TW>
size_t fused_lds(const FusedArgs& a) {
    constexpr int P = TH * TW, PP = (P + 31) / 32 * 32;
    const int IH = (TH - 1) * a.stride + a.k, IW = (TW - 1) * a.stride + a.k;
    const int IHWP = (IH * IW + 31) / 32 * 32;
    const int EW = a.ch + 8, NTO = (a.cout + 31) / 32, OW = NTO * 32 + 8;
    const size_t es = a.w1 ? (size_t)IHWP * EW : 0, os = a.w3 ? (size_t)PP * OW : 0;
    size_t h = (size_t)IHWP * a.xw + (es > os ? es : os);
    if (a.w3) h += (size_t)PP * EW;
    if (a.w1) h += (size_t)2 * 32 * a.xw;
    if (a.w3) h += (size_t)2 * NTO * 32 * EW;
    h += (size_t)2 * a.k * a.k * 32;
    size_t b = h * 2 + 128 * 4 + IHWP + 16;
    if (a.pool) b += (size_t)16 * 8 * 4;
    return b;
}

template <int TH, int TW, int K, int MAXT>
int launch_tkm(const FusedArgs& a, hipStream_t s, size_t lds) {
    static bool attr = false;
    if (!attr) {
        DN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_kernel<TH, TW, K, MAXT>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    dim3 grid(dn_cdiv(a.Wo, TW), dn_cdiv(a.Ho, TH), a.n);
    dn_note_kernel("fused_kernel<%d,%d,%d,%d>", TH, TW, K, MAXT);
    hipLaunchKernelGGL((fused_kernel<TH, TW, K, MAXT>), grid, dim3(FT), lds, s, a);
    return DN_OK;
}

template <int TH, int TW, int K>
int launch_tk(const FusedArgs& a, hipStream_t s, size_t lds) {
    constexpr int PP = (TH * TW + 31) / 32 * 32;
    const int npairs = a.w3 ? (PP / 32) * ((a.cout + 31) / 32) : 0;
    if (npairs <= 4) return launch_tkm<TH, TW, K, 1>(a, s, lds);
    if (npairs <= 8) return launch_tkm<TH, TW, K, 2>(a, s, lds);
    return launch_tkm<TH, TW, K, 4>(a, s, lds);
}

template <int TH, int TW>
int launch_t(const FusedArgs& a, hipStream_t s) {
    const size_t lds = fused_lds<TH, TW>(a);
    if (lds > 160 * 1024) {
        dn_set_error("fused: tile %dx%d needs %zu B of LDS", TH, TW, lds);
        return DN_E_UNSUPPORTED;
    }
    return a.k == 3 ? launch_tk<TH, TW, 3>(a, s, lds) : launch_tk<TH, TW, 5>(a, s, lds);
}

}  // namespace

// Tile choice: as large as possible (less halo recompute) while the LDS footprint still admits ~3 workgroups per CU.
// Depends only on the op geometry, so the plan can size the pooled partial-sum tensors (rows = tiles per image).
void fused_tile(int Ho, int Wo, int* th, int* tw) {
    if (Ho >= 40) { *th = 8; *tw = 8; }
    else if (Ho >= 20) { *th = 5; *tw = 20; }
    else if (Ho >= 10) { *th = 5; *tw = 10; }
    else { *th = 8; *tw = 8; }
}

int fused_tiles_per_image(int Ho, int Wo) {
    int th, tw;
    fused_tile(Ho, Wo, &th, &tw);
    return dn_cdiv(Ho, th) * dn_cdiv(Wo, tw);
}

int launch_fused(const FusedArgs& a0, hipStream_t s) {
    FusedArgs a = a0;
    DN_REQUIRE(a.cin % 8 == 0 && a.cexp % 8 == 0 && (!a.w3 || a.cout % 8 == 0), "fused: channel counts must be multiples of 8");
    DN_REQUIRE(!a.w1 || a.cin <= 120, "fused: expand cin=%d > 120", a.cin);
    DN_REQUIRE(a.k == 3 || a.k == 5, "fused: k=%d", a.k);
    DN_REQUIRE(!a.w3 || a.cout <= 96, "fused: project cout=%d > 96", a.cout);
    a.ch = (a.cexp <= 16) ? 16 : 32;
    a.stamps = g_fu_stamps;
    a.xw = (a.w1 ? (a.cin + 15) / 16 * 16 : a.cin) + 8;
    int th, tw;
    fused_tile(a.Ho, a.Wo, &th, &tw);
    const int PP = (th * tw + 31) / 32 * 32;
    if (a.w3 && (PP / 32) * ((a.cout + 31) / 32) > 16) {
        dn_set_error("fused: too many projection tiles for tile %dx%d, cout %d", th, tw, a.cout);
        return DN_E_UNSUPPORTED;
    }
    if (th == 8 && tw == 8) return launch_t<8, 8>(a, s);
    if (th == 5 && tw == 20) return launch_t<5, 20>(a, s);
    return launch_t<5, 10>(a, s);
}
