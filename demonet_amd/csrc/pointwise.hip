// Pointwise (1x1) convolution == GEMM  out[m][n] = act(sum_k x[m][k] * w[n][k] + bias[n]) (+ residual)
//
// reference ops replaced: every 1x1 ConvBNActivation / SE-scaled projection / SSDLite head 1x1 conv
//   (mobilenetv3.py:76,88 ; ssd_mobilenetv3.py:35,44,52 ; BN folded into w/bias at plan time).
//
// CDNA4 mapping. Activations are NHWC fp16, so x rows are K-contiguous and so are the rows of the weight
// matrix in its native [cout][cin] layout: both are directly the per-lane 8-half fragments of
// v_mfma_f32_32x32x16_f16. The operands are SWAPPED (A := weight tile, B := pixel tile) so that the 32x32
// accumulator has the pixel on the lane and 4 consecutive output channels in consecutive registers:
// the epilogue then writes channel-contiguous vectors straight into NHWC rows without a transpose.
// Tiles are staged through LDS (rows padded to 80 B: conflict-free ds_read_b128), register-staged double
// buffering with one barrier per 32-deep K step. These GEMMs are HBM-bound (AI 13..300 FLOP/B):
// the design goal is to read x once, write out once and keep the weight tile L2-resident.
#include <stdlib.h>

#include "common.h"

// dev hooks: exported and compiled into the kernels only by the dev build (python -m demonet_amd.build --stamps, -DDN_DEV_STAMPS)
#ifdef DN_DEV_STAMPS
static int g_pw_tile = 0;                   // tools/tune_pw.py: force a tile variant for every launch
extern "C" __attribute__((visibility("default"))) void dn_debug_pw_tile(int t) { g_pw_tile = t; }
static long long* g_pw_stamps = nullptr;     // tools/probe_pw_stamps.py: per-workgroup phase stamps
extern "C" __attribute__((visibility("default"))) void dn_debug_pw_stamps(void* dev_ptr) { g_pw_stamps = (long long*)dev_ptr; }
#define PW_STAMP(k) do { if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
constexpr int g_pw_tile = 0;
constexpr long long* g_pw_stamps = nullptr;
#define PW_STAMP(k) do { } while (0)
#endif

// row tiles of BP pixels: per XCD group of a.xq images when the grouping is on, else of the whole problem
static inline int pw_row_tiles(const PwArgs& a, int BP) { return a.xq > 0 ? dn_cdiv((long)a.xq * a.hw, BP) : dn_cdiv(a.m, BP); }

namespace {


// CONV = true: implicit GEMM for a dense kxk convolution (VGG path, SSDHead 3x3 heads; ssd_vgg16.py, generalized_ssd.py:77-92):
// K runs over (ky, kx, cin) with the weight stored [cout][ky][kx][cin]; each 32-deep K stage lies inside one tap
// (cin % 32 == 0), so the pixel tile of a stage is the NHWC rows of the tap-shifted input pixels (zeros outside).
// XCD affinity (common.h, xcd_group_rows): with a.xq > 0 the workgroups of a launch with equal (flat index % 8) -- one XCD --
// own the rows of one contiguous group of a.xq images, in every kernel of the chain, so that a layer reads what the same XCD's
// L2 has just written. flat = workgroup index within its problem, tiles = row tiles per group (plain mapping: per problem).
template <int BP>
__device__ __forceinline__ bool pw_tile_rows(const PwArgs& a, int flat, int tiles, int& m0, int& mend, int& by) {
    if (a.xq > 0) {
        const int g = flat & 7, w = flat >> 3;
        by = w / tiles;
        const int t = w - by * tiles;
        const int r0 = g * a.xq * a.hw;
        mend = min(a.m, r0 + a.xq * a.hw);
        m0 = r0 + t * BP;
        return m0 < mend;
    }
    by = flat / tiles;
    m0 = (flat - by * tiles) * BP;
    mend = a.m;
    return true;
}

template <int BP, int BC, int WP, int WC, bool CONV, int BK = 32, int PF = 1, bool SEF = false, bool FK = false>
__device__ __forceinline__ void pw_body(PwArgs a, const int m0, const int mend, const int by) {
    static_assert(WP * WC == 4, "4 waves per workgroup");
    constexpr int LDS_ROW = BK + 8;     // halfs per LDS row: BK data + 8 pad -> odd number of 16-B slots (conflict-free b128)
    constexpr int CPR = BK / 8;         // 16-B chunks per row per stage (any multiple of 2: divisions by a constant)
    constexpr int TP = BP / WP / 32;    // 32-pixel MFMA tiles per wave
    constexpr int TC = BC / WC / 32;    // 32-channel MFMA tiles per wave
    constexpr int NX = (BP * CPR + 255) / 256;  // 16-B chunks of the pixel tile per thread per stage (last may be partial)
    constexpr int NW = (BC * CPR + 255) / 256;  // 16-B chunks of the weight tile per thread per stage
    constexpr int NWc = NW;
    extern __shared__ __attribute__((aligned(16))) half_t lds_raw[];      // bias[BC] floats, then [1|2][(BP + BC) * LDS_ROW] halfs
    float* bsh = reinterpret_cast<float*>(lds_raw);
    constexpr int SEF_FLOATS = SEF ? 832 : 0;           // scale[2][128], mean[2][128], z[2][32], fc1 partials[2][4][32]
    float* sef = bsh + BC;
    // FK with an SE scale vector (a.se): the scales of the tile's (at most two) images sit in LDS, [2][K] floats behind the bias
    const int FKSE_FLOATS = (FK && a.se) ? 2 * a.cin : 0;
    float* sesc = bsh + BC + SEF_FLOATS;
    half_t* lds_dyn = lds_raw + 2 * BC + 2 * SEF_FLOATS + 2 * FKSE_FLOATS;
    half_t (*lds)[(BP + BC) * LDS_ROW] = reinterpret_cast<half_t (*)[(BP + BC) * LDS_ROW]>(lds_dyn);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wp = wave / WC, wc = wave % WC;
    const int r = lane & 31, hh = lane >> 5;
    const int n0 = by * BC;
    const int M = mend, K = a.cin, NC = a.cout;      // rows beyond the group's end belong to another workgroup
    const int dbg = a.act >> 8;            // probe-only knobs: 1 = skip stores, 2 = skip global loads of x
    a.act &= 0xff;

    PW_STAMP(0);
    floatx16 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    uint4 sx[NX], sw[NWc];
    int cvy[NX], cvx[NX];               // CONV: top-left input coordinate of the output pixel of each staged row
    long cvo[NX];                       //       and the element offset of that (possibly out-of-image) pixel's chunk in x
    if constexpr (CONV) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int c = tid + 256 * i;
            const int m = m0 + c / CPR;
            const int img = m / a.hw, rem = m - img * a.hw;
            const int oy = rem / a.cv_wo, ox = rem - oy * a.cv_wo;
            cvy[i] = oy * a.cv_stride - a.cv_pad;
            cvx[i] = ox * a.cv_stride - a.cv_pad;
            cvo[i] = (((long)img * a.cv_h + cvy[i]) * a.cv_w + cvx[i]) * a.cv_cin + (c % CPR) * 8;
        }
    }

    auto load_into = [&](uint4 (&sx)[NX], uint4 (&sw)[NWc], int k0) {
        int dy = 0, dx = 0;
        long soff = 0;                  // CONV: element offset of this stage's tap and channel slice (uniform)
        if constexpr (CONV) {
            // per K stage and uniform: mul-hi divisions (k0 and cv_cin are multiples of 32) -- the hardware-division sequences were a
            // good part of a stage of the small dense convs, which run one wave per SIMD with nothing to hide them behind
            const int tap = (int)fd_div((unsigned)k0 >> 5, a.fd_cin32);
            const int tap_c0 = k0 - tap * a.cv_cin;
            const int ky = (int)fd_div((unsigned)tap, a.fd_k);
            dy = ky * a.cv_dil;
            dx = (tap - ky * a.cv_k) * a.cv_dil;
            soff = ((long)dy * a.cv_w + dx) * a.cv_cin + tap_c0;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int c = tid + 256 * i;
            const int row = c / CPR, q = c - row * CPR;
            const int m = (row < BP) ? m0 + row : M, k = k0 + q * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if constexpr (CONV) {
                const int iy = cvy[i] + dy, ix = cvx[i] + dx;
                if (m < M && k < K && iy >= 0 && iy < a.cv_h && ix >= 0 && ix < a.cv_w)
                    v = *reinterpret_cast<const uint4*>(a.x + (cvo[i] + soff));
            } else if (m < M && k < K && !(dbg & 2)) {
                v = *reinterpret_cast<const uint4*>(a.x + (size_t)m * K + k);
                if (a.se) {
                    const float* sp = a.se + (size_t)(m / a.hw) * K + k;
                    half8 hv = *reinterpret_cast<half8*>(&v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) hv[e] = (half_t)((float)hv[e] * sp[e]);
                    v = *reinterpret_cast<uint4*>(&hv);
                }
            }
            sx[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int c = tid + 256 * i;
            const int row = c / CPR, q = c - row * CPR;
            const int n = n0 + row, k = k0 + q * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < BC && n < NC && k < K) v = *reinterpret_cast<const uint4*>(a.w + (size_t)n * K + k);
            sw[i] = v;
        }
    };
    // FK (round 3, template flag set by the launcher when every problem of the launch has K % BK == 0, no SE-scaled staging and arrays below 2 GB):
    // the K loop without bound tests -- chunk byte offsets formed once, rows beyond the problem CLAMPED instead of predicated (their products land in
    // rows / columns the epilogue never stores), a stage = NX + NW unconditional loads at offset + 2 k0. The three-stage ring is spelled out as NAMED
    // register sets: as arrays hipcc kept its weight half in scratch memory with a run-time slot index (88 -> 167 us).
    unsigned xo[NX], wo[NWc];
    if constexpr (FK) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int c = tid + 256 * i;
            const int row = c / CPR, q = c - row * CPR;
            xo[i] = ((unsigned)min(m0 + min(row, BP - 1), M - 1) * (unsigned)K + (unsigned)q * 8u) * 2u;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int c = tid + 256 * i;
            const int row = c / CPR, q = c - row * CPR;
            wo[i] = ((unsigned)min(n0 + min(row, BC - 1), NC - 1) * (unsigned)K + (unsigned)q * 8u) * 2u;
        }
    }
    // (plain ext-vector values, not HIP's uint4 struct: its copies are memcpy calls on allocas, and with the lifetime markers of the inlined lambdas
    //  SROA left the weight sets in scratch memory -- 88 -> 167 us)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto load_fast = [&](u32x4 (&sx)[NX], u32x4 (&sw)[NWc], int k0) __attribute__((always_inline)) {
        const char* xb = reinterpret_cast<const char*>(a.x);
        const char* wb = reinterpret_cast<const char*>(a.w);
        const unsigned ko = (unsigned)k0 * 2u;
#pragma unroll
        for (int i = 0; i < NX; ++i) sx[i] = *reinterpret_cast<const u32x4*>(xb + (xo[i] + ko));
#pragma unroll
        for (int i = 0; i < NW; ++i) sw[i] = *reinterpret_cast<const u32x4*>(wb + (wo[i] + ko));
    };
    int se_split = 0x7fffffff;              // FK + a.se: first row of the tile that belongs to the second image
    auto store_fast = [&](const u32x4 (&sx)[NX], const u32x4 (&sw)[NWc], int b, int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int c = tid + 256 * i;
            const int row = c / CPR, q = c - row * CPR;
            u32x4 v = sx[i];
            if (a.se) {
                // SE scale applied as the rows are staged: (half)((float)x * s), the rounding of the general path (load_into) -- there the
                // product sat right behind the load and every stage waited for its own memory round trip
                const float* sp = sesc + (min(row, BP - 1) >= se_split ? K : 0) + k0 + q * 8;
                const float4 s0 = *reinterpret_cast<const float4*>(sp), s1 = *reinterpret_cast<const float4*>(sp + 4);
                half8 hv = *reinterpret_cast<half8*>(&v);
                hv[0] = (half_t)((float)hv[0] * s0.x); hv[1] = (half_t)((float)hv[1] * s0.y); hv[2] = (half_t)((float)hv[2] * s0.z); hv[3] = (half_t)((float)hv[3] * s0.w);
                hv[4] = (half_t)((float)hv[4] * s1.x); hv[5] = (half_t)((float)hv[5] * s1.y); hv[6] = (half_t)((float)hv[6] * s1.z); hv[7] = (half_t)((float)hv[7] * s1.w);
                v = *reinterpret_cast<u32x4*>(&hv);
            }
            // (unconditional: a chunk beyond the tile was LOADED from the clamped row, so it rewrites that row's piece with the same bytes -- no branch)
            *reinterpret_cast<u32x4*>(&lds[b][min(row, BP - 1) * LDS_ROW + q * 8]) = v;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int c = tid + 256 * i;
            const int row = c / CPR, q = c - row * CPR;
            *reinterpret_cast<u32x4*>(&lds[b][(BP + min(row, BC - 1)) * LDS_ROW + q * 8]) = sw[i];
        }
    };
    int sef_split = 0x7fffffff;             // SEF: first row of the tile that belongs to the second image
    auto store_from = [&](const uint4 (&sx)[NX], const uint4 (&sw)[NWc], int b, int k0 = 0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int c = tid + 256 * i;
            const int row = c / CPR, q = c - row * CPR;
            uint4 v = sx[i];
            if constexpr (SEF) {
                // SE scale applied as the rows are staged: (half)((float)x * s), the rounding of the launch-per-op path (load_into)
                const float* sp = sef + (row >= sef_split ? 128 : 0) + min(k0 + q * 8, 120);
                half8 hv = *reinterpret_cast<half8*>(&v);
#pragma unroll
                for (int e = 0; e < 8; ++e) hv[e] = (half_t)((float)hv[e] * sp[e]);
                v = *reinterpret_cast<uint4*>(&hv);
            }
            if (row < BP) *reinterpret_cast<uint4*>(&lds[b][row * LDS_ROW + q * 8]) = v;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int c = tid + 256 * i;
            const int row = c / CPR, q = c - row * CPR;
            if (row < BC) *reinterpret_cast<uint4*>(&lds[b][(BP + row) * LDS_ROW + q * 8]) = sw[i];
        }
    };

    const int KT = (K + BK - 1) / BK;
    if (tid < BC) bsh[tid] = (n0 + tid < NC) ? a.bias[n0 + tid] : 0.f;      // visible after the first barrier below
    // residual rows of the output tile: requested now as row-contiguous 16-B chunks, consumed after the K loop (one exposed
    // memory round trip less per workgroup than loading them in the epilogue)
    constexpr int CPRO = BC / 8;                    // 16-B chunks per tile row
    constexpr int NCH = (BP * CPRO + 255) / 256;
    uint4 rv[NCH];
    const bool staged_res = a.residual && !a.out_fp32;
    if (staged_res) {
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int c = tid + 256 * u;
            const int row = c / CPRO, ch = c % CPRO;
            const int m = m0 + row, n = n0 + ch * 8;
            rv[u] = (row < BP && m < M && n < NC) ? *reinterpret_cast<const uint4*>(a.residual + (size_t)m * NC + n) : make_uint4(0, 0, 0, 0);
        }
    }
    auto mfma_stage = [&](int b, int kt) {
        const int ksteps = min(BK / 16, (K - kt * BK + 15) >> 4);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            if (ks < ksteps) {
                half8 xf[TP], wf[TC];
#pragma unroll
                for (int j = 0; j < TP; ++j)
                    xf[j] = *reinterpret_cast<const half8*>(&lds[b][((wp * TP + j) * 32 + r) * LDS_ROW + ks * 16 + hh * 8]);
#pragma unroll
                for (int i = 0; i < TC; ++i)
                    wf[i] = *reinterpret_cast<const half8*>(&lds[b][(BP + (wc * TC + i) * 32 + r) * LDS_ROW + ks * 16 + hh * 8]);
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TP; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
            }
        }
    };
    auto mfma_full = [&](int b) __attribute__((always_inline)) {       // every K step of the stage exists: no tail test between the steps
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            half8 xf[TP], wf[TC];
#pragma unroll
            for (int j = 0; j < TP; ++j)
                xf[j] = *reinterpret_cast<const half8*>(&lds[b][((wp * TP + j) * 32 + r) * LDS_ROW + ks * 16 + hh * 8]);
#pragma unroll
            for (int i = 0; i < TC; ++i)
                wf[i] = *reinterpret_cast<const half8*>(&lds[b][(BP + (wc * TC + i) * 32 + r) * LDS_ROW + ks * 16 + hh * 8]);
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
    };
    if constexpr (FK && (PF == 3 || PF == 4) && !SEF) {
        u32x4 x0[NX], x1[NX], x2[NX], x3[PF == 4 ? NX : 1], w0[NWc], w1[NWc], w2[NWc], w3[PF == 4 ? NWc : 1];
        load_fast(x0, w0, 0);
        if (KT > 1) load_fast(x1, w1, BK);
        if (KT > 2) load_fast(x2, w2, 2 * BK);
        if constexpr (PF == 4) { if (KT > 3) load_fast(x3, w3, 3 * BK); }
        if (a.se) {
            const int img0 = m0 / a.hw, img_last = (min(m0 + BP, M) - 1) / a.hw;       // (the launcher checks hw >= BP: at most two images per tile)
            se_split = (img0 + 1) * a.hw - m0;
            for (int t = tid; t < 2 * K; t += 256) {
                const int im = t >= K ? 1 : 0;
                sesc[t] = (img0 + im <= img_last) ? a.se[(size_t)(img0 + im) * K + (t - im * K)] : 0.f;
            }
            __syncthreads();
        }
        store_fast(x0, w0, 0, 0);
        __syncthreads();
        PW_STAMP(1);
        // stage kt sits in LDS buffer kt & 1; set (kt + 1) % PF holds stage kt + 1, set kt % PF is free for stage kt + PF
#define PW_FK_STAGE(KT_, XN, WN, XF, WF)                                                            \
        if ((KT_) < KT) {                                                                            \
            mfma_full((KT_) & 1);                                                                    \
            if ((KT_) + 1 < KT) store_fast(XN, WN, ((KT_) + 1) & 1, ((KT_) + 1) * BK);                \
            if ((KT_) + PF < KT) load_fast(XF, WF, ((KT_) + PF) * BK);                               \
            __syncthreads();                                                                         \
        }
        if constexpr (PF == 3) {
            for (int kt0 = 0; kt0 < KT; kt0 += 3) {
                PW_FK_STAGE(kt0, x1, w1, x0, w0)
                PW_FK_STAGE(kt0 + 1, x2, w2, x1, w1)
                PW_FK_STAGE(kt0 + 2, x0, w0, x2, w2)
            }
        } else {
            for (int kt0 = 0; kt0 < KT; kt0 += 4) {
                PW_FK_STAGE(kt0, x1, w1, x0, w0)
                PW_FK_STAGE(kt0 + 1, x2, w2, x1, w1)
                PW_FK_STAGE(kt0 + 2, x3, w3, x2, w2)
                PW_FK_STAGE(kt0 + 3, x0, w0, x3, w3)
            }
        }
#undef PW_FK_STAGE
    } else if constexpr (PF > 1) {
        // Register ring of PF stages: every stage's global loads are requested PF stages before their use. For short K
        // (KT <= PF) that is everything up front -- ONE exposed memory round trip per workgroup instead of one per 32-deep
        // stage; for long K the latency hides behind PF stages of MFMA work. The LDS double buffer and its barriers stay.
        uint4 px[PF][NX], pwt[PF][NWc];
#pragma unroll
        for (int st = 0; st < PF; ++st)
            if (st < KT) load_into(px[st], pwt[st], st * BK);
        if constexpr (SEF) {
            // ---- squeeze-excitation FCs of this tile's (at most two) images, while the tile loads are in flight
            //      (mobilenetv3.py:31-36: mean -> fc1 + ReLU -> fc2 + Hardsigmoid; c <= 128, squeeze <= 32)
            float* s_scale = sef; float* s_mean = sef + 256; float* s_z = sef + 512; float* s_p = sef + 576;
            const int C = K, sq = a.sef_sq, nblk = a.sef_nblk;
            const int img0 = m0 / a.hw, img_last = (min(m0 + BP, M) - 1) / a.hw;
            sef_split = (img0 + 1) * a.hw - m0;
            const int im = tid >> 7;
            const bool live = img0 + im <= img_last;
            {
                const int c = tid & 127;
                float t = 0.f;
                if (c < C && live) {
                    const float* pp = a.sef_part + (size_t)(img0 + im) * nblk * C + c;
                    for (int b0 = 0; b0 < nblk; b0 += 8) {
                        float v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = pp[(size_t)min(b0 + u, nblk - 1) * C];
#pragma unroll
                        for (int u = 0; u < 8; ++u) t += (b0 + u < nblk) ? v[u] : 0.f;
                    }
                }
                s_mean[im * 128 + c] = t * a.sef_inv;
            }
            __syncthreads();
            {
                const int sl = (tid >> 5) & 3, j = tid & 31;
                const int per = (C + 3) >> 2, i0 = sl * per, i1 = min(C, i0 + per);
                float t = 0.f;
                if (j < sq && live) {
                    for (int ib = i0; ib < i1; ib += 8) {
                        half_t v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = a.sef_w1t[(size_t)min(ib + u, i1 - 1) * sq + j];
#pragma unroll
                        for (int u = 0; u < 8; ++u) t += (ib + u < i1) ? (float)v[u] * s_mean[im * 128 + ib + u] : 0.f;
                    }
                }
                s_p[(im * 4 + sl) * 32 + j] = t;
            }
            __syncthreads();
            if (tid < 64) {
                const int i2 = tid >> 5, j = tid & 31;
                float z = 0.f;
                if (j < sq) z = fmaxf(a.sef_b1[j] + s_p[(i2 * 4 + 0) * 32 + j] + s_p[(i2 * 4 + 1) * 32 + j] + s_p[(i2 * 4 + 2) * 32 + j] + s_p[(i2 * 4 + 3) * 32 + j], 0.f);
                s_z[i2 * 32 + j] = z;
            }
            __syncthreads();
            {
                const int i = tid & 127;
                float t = 0.f;
                if (i < C && live) {
                    t = a.sef_b2[i];
                    for (int jb = 0; jb < sq; jb += 8) {
                        half_t v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = a.sef_w2t[(size_t)min(jb + u, sq - 1) * C + i];
#pragma unroll
                        for (int u = 0; u < 8; ++u) t += (jb + u < sq) ? (float)v[u] * s_z[im * 32 + jb + u] : 0.f;
                    }
                    t = fminf(fmaxf(t + 3.f, 0.f), 6.f) * (1.f / 6.f);
                }
                s_scale[im * 128 + i] = t;
            }
            __syncthreads();
        }
        store_from(px[0], pwt[0], 0, 0);
        __syncthreads();
        PW_STAMP(1);
        for (int kt0 = 0; kt0 < KT; kt0 += PF) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const int kt = kt0 + j;
                if (kt < KT) {
                    mfma_stage(kt & 1, kt);
                    // slot (j+1)%PF holds stage kt+1; slot j (stage kt, already in LDS) is free for stage kt+PF
                    if (kt + 1 < KT) store_from(px[(j + 1) % PF], pwt[(j + 1) % PF], (kt + 1) & 1, (kt + 1) * BK);
                    if (kt + PF < KT) load_into(px[j], pwt[j], (kt + PF) * BK);
                    __syncthreads();
                }
            }
        }
    } else {
        load_into(sx, sw, 0);
        store_from(sx, sw, 0);
        __syncthreads();
        PW_STAMP(1);
        for (int kt = 0; kt < KT; ++kt) {
            const int b = kt & 1;
            if (kt + 1 < KT) load_into(sx, sw, (kt + 1) * BK);
            mfma_stage(b, kt);
            if (kt + 1 < KT) store_from(sx, sw, b ^ 1);
            __syncthreads();
        }
    }

    PW_STAMP(2);
    // epilogue: lane holds pixel (lane&31) of each pixel tile; registers 4g..4g+3 = channels 8g+4h..+3 of each channel tile.
    // Two phases so that no load sits behind a possibly-aliasing store: (1) all residual loads, (2) math + stores.
    // Bias comes from LDS (staged before the K loop).
    if (!a.out_fp32 && !(dbg & 1)) {
        // fp16 outputs: stage the finished tile in LDS ([BP][BC+8] halfs, reusing the K-loop buffers) and write it out as
        // row-contiguous 16-byte chunks -> every wave store covers whole 128-B lines; the residual is read the same way.
        constexpr int OROW = BC + 8;
        // (the launcher sizes the dynamic LDS as max(K-loop buffers, output tile))
        // (the last K-loop iteration ended with a barrier: nobody reads the staging buffers any more)
        if (staged_res) {
            // residual layers stage the tile in fp32: the chunk loop below then adds the residual (requested before the K loop, already
            // in this chunk layout) in fp32 and rounds ONCE -- act(conv + bias) + x as mobilenetv3.py:97-99 computes it
            constexpr int FROW = BC + 4;
            float* otf = reinterpret_cast<float*>(lds_dyn);
            dn_tile_emit<TP, TC>(acc, bsh, wc, hh, a.act, [&](int i, int j, int g, const float4& v) {
                *reinterpret_cast<float4*>(&otf[((wp * TP + j) * 32 + r) * FROW + (wc * TC + i) * 32 + 8 * g + 4 * hh]) = v;
            });
            __syncthreads();
#pragma unroll
            for (int u = 0; u < NCH; ++u) {
                const int c = tid + 256 * u;
                const int row = c / CPRO, ch = c % CPRO;
                const int m = m0 + row, n = n0 + ch * 8;
                if (row < BP && m < M && n < NC) {
                    const float4 lo = *reinterpret_cast<const float4*>(&otf[row * FROW + ch * 8]);
                    const float4 hi = *reinterpret_cast<const float4*>(&otf[row * FROW + ch * 8 + 4]);
                    const half8 rr = *reinterpret_cast<const half8*>(&rv[u]);
                    half8 hv;
                    hv[0] = (half_t)(lo.x + (float)rr[0]); hv[1] = (half_t)(lo.y + (float)rr[1]);
                    hv[2] = (half_t)(lo.z + (float)rr[2]); hv[3] = (half_t)(lo.w + (float)rr[3]);
                    hv[4] = (half_t)(hi.x + (float)rr[4]); hv[5] = (half_t)(hi.y + (float)rr[5]);
                    hv[6] = (half_t)(hi.z + (float)rr[6]); hv[7] = (half_t)(hi.w + (float)rr[7]);
                    *reinterpret_cast<half8*>(reinterpret_cast<half_t*>(a.out) + (size_t)m * NC + n) = hv;
                }
            }
            PW_STAMP(3);
            return;
        }
        half_t* ot = lds_dyn;
        dn_tile_emit<TP, TC>(acc, bsh, wc, hh, a.act, [&](int i, int j, int g, const float4& v) {
            half4 hv;
            hv[0] = (half_t)v.x; hv[1] = (half_t)v.y; hv[2] = (half_t)v.z; hv[3] = (half_t)v.w;
            *reinterpret_cast<half4*>(&ot[((wp * TP + j) * 32 + r) * OROW + (wc * TC + i) * 32 + 8 * g + 4 * hh]) = hv;
        });
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int c = tid + 256 * u;
            const int row = c / CPRO, ch = c % CPRO;
            const int m = m0 + row, n = n0 + ch * 8;
            if (row < BP && m < M && n < NC)
                *reinterpret_cast<uint4*>(reinterpret_cast<half_t*>(a.out) + (size_t)m * NC + n) = *reinterpret_cast<const uint4*>(&ot[row * OROW + ch * 8]);
        }
        PW_STAMP(3);
        return;
    }
    if (a.out_fp32 && !(dbg & 1)) {
        // fp32 head outputs: rows of the [anchor][class] arrays are NC floats long at arbitrary 8-byte alignment, and the
        // accumulator layout has every lane on a different row -> direct stores touch each 128-B line 8 times. Stage the
        // tile in LDS ([BP][BC+4] floats over the K-loop buffers) and write row-contiguous float2 runs instead.
        constexpr int FROW = BC + 4;
        float* ot = reinterpret_cast<float*>(lds_dyn);
        dn_tile_emit<TP, TC>(acc, bsh, wc, hh, a.act, [&](int i, int j, int g, const float4& v) {
            *reinterpret_cast<float4*>(&ot[((wp * TP + j) * 32 + r) * FROW + (wc * TC + i) * 32 + 8 * g + 4 * hh]) = v;
        });
        __syncthreads();
        float* outp = reinterpret_cast<float*>(a.out);
        const bool pair_ok = ((NC | (int)(a.out_base & 1) | (int)(a.out_img_stride & 1)) & 1) == 0 &&
                             (reinterpret_cast<size_t>(outp) & 7) == 0;       // every row start 8-byte aligned
        constexpr int PPR = BC / 2;                     // float2 per tile row
        const int img0 = m0 / a.hw, rem0 = m0 - img0 * a.hw;
        constexpr int RG = 256 / PPR;                   // row groups: thread (rg, j) owns column pair j of rows rg, rg + RG, ...
        if (a.hw >= RG && RG >= 1) {
            // Round 3 instruction diet: the element loop used to pay a division, the image wrap and a 64-bit address per float2 (~25 vector
            // instructions each, ~600 per thread and tile); with a fixed column pair per thread the row walks by increments.
            const int rg = tid / PPR, j = tid - rg * PPR;
            const int n = n0 + 2 * j;
            if (rg < RG && n < NC) {
                int row = rg;
                int pix = rem0 + rg, img = img0;
                while (pix >= a.hw) { pix -= a.hw; ++img; }
                float* o = outp + (size_t)a.out_base + (size_t)img * a.out_img_stride + (size_t)pix * NC + n;
                const long wrap = a.out_img_stride - (long)a.hw * NC;
                const bool two_ok = pair_ok && n + 1 < NC;
                const float* src = &ot[row * FROW + 2 * j];
                for (; row < BP && m0 + row < M; row += RG) {
                    const float2 v = *reinterpret_cast<const float2*>(src);
                    if (two_ok) {
                        *reinterpret_cast<float2*>(o) = v;
                    } else {
                        o[0] = v.x;
                        if (n + 1 < NC) o[1] = v.y;
                    }
                    src += RG * FROW;
                    o += (size_t)RG * NC;
                    pix += RG;
                    if (pix >= a.hw) { pix -= a.hw; o += wrap; }      // (hw >= RG: at most one image boundary per step)
                }
            }
            PW_STAMP(3);
            return;
        }
        const bool two = a.hw >= BP;
#pragma unroll 4
        for (int c = tid; c < BP * PPR; c += 256) {
            const int row = c / PPR, cp = (c - row * PPR) * 2;
            const int m = m0 + row, n = n0 + cp;
            if (m >= M || n >= NC) continue;
            int img, pix;
            if (two) { const int t = rem0 + row; const bool wrap = t >= a.hw; img = img0 + (wrap ? 1 : 0); pix = wrap ? t - a.hw : t; }
            else { img = m / a.hw; pix = m - img * a.hw; }
            float* o = outp + (size_t)a.out_base + (size_t)img * a.out_img_stride + (size_t)pix * NC + n;
            const float2 v = *reinterpret_cast<const float2*>(&ot[row * FROW + cp]);
            if (pair_ok && n + 1 < NC) {
                *reinterpret_cast<float2*>(o) = v;
            } else {
                o[0] = v.x;
                if (n + 1 < NC) o[1] = v.y;
            }
        }
        PW_STAMP(3);
        return;
    }
    size_t obase[TP];
    bool mvalid[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int m = m0 + (wp * TP + j) * 32 + r;
        mvalid[j] = m < M;
        if (a.out_fp32) {
            const int img = m / a.hw;
            obase[j] = (size_t)a.out_base + (size_t)img * a.out_img_stride + (size_t)(m - img * a.hw) * NC;
        } else {
            obase[j] = (size_t)m * NC;
        }
    }
    half4 resv[TP][TC][4];
    if (a.residual) {
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = n0 + (wc * TC + i) * 32 + 8 * g + 4 * hh;
                    half4 rv = {0, 0, 0, 0};
                    if (mvalid[j] && c0 < NC) rv = *reinterpret_cast<const half4*>(a.residual + obase[j] + c0);
                    resv[j][i][g] = rv;
                }
    }
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        if (!mvalid[j]) continue;
#pragma unroll
        for (int i = 0; i < TC; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = (wc * TC + i) * 32 + 8 * g + 4 * hh;      // channel within the block tile
                const int c0 = n0 + cl;
                if (c0 >= NC) continue;
                const float4 bv = *reinterpret_cast<const float4*>(&bsh[cl]);
                float v[4] = {acc[i][j][4 * g + 0] + bv.x, acc[i][j][4 * g + 1] + bv.y, acc[i][j][4 * g + 2] + bv.z,
                              acc[i][j][4 * g + 3] + bv.w};
                dn_act_n<float[4], 4>(v, a.act);
                if (dbg & 1) {
                    if (v[0] == 123.456f) reinterpret_cast<float*>(a.out)[0] = v[1];
                } else if (a.out_fp32) {
                    float* o = reinterpret_cast<float*>(a.out) + obase[j] + c0;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c0 + e < NC) o[e] = v[e];
                } else {
                    // cout is a multiple of 4 for every fp16 layer -> the 4-channel group is entirely in range
                    if (a.residual) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)resv[j][i][g][e];
                    }
                    half4 hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) hv[e] = (half_t)v[e];
                    *reinterpret_cast<half4*>(reinterpret_cast<half_t*>(a.out) + obase[j] + c0) = hv;
                }
            }
        }
    }
    PW_STAMP(3);
}

template <int BP, int BC, int WP, int WC, bool CONV, int BK = 32, int PF = 1, bool SEF = false, bool FK = false>
__global__ __launch_bounds__(256) void pw_kernel(PwArgs a, int tiles) {
    int m0, mend, by;
    if (!pw_tile_rows<BP>(a, blockIdx.x, tiles, m0, mend, by)) return;
    pw_body<BP, BC, WP, WC, CONV, BK, PF, SEF, FK>(a, m0, mend, by);
}

// Grouped launch: up to 12 independent GEMMs (e.g. the class-head 1x1 convs of all pyramid levels) in ONE launch.
// The fixed cost of a launch chain (~9 us per dependent launch at any batch size) dominates the small levels; here
// their workgroups simply ride along with level 0's. blockIdx.x is flat; start[] are prefix sums of workgroup counts.
struct PwGroup {
    int count;
    int start[13];
    int gx[12];
    PwArgs a[12];
};

template <int BP, int BC, int WP, int WC, bool CONV, int BK = 32, int PF = 1, bool FK = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((FK && BP == 128 && BC == 96) ? 3 : 1))) void pw_group_kernel(PwGroup g) {
    int p = 0;
#pragma unroll
    for (int i = 1; i < 12; ++i)
        if (i < g.count && (int)blockIdx.x >= g.start[i]) p = i;
    int m0, mend, by;       // (start[] are multiples of 8 when the XCD grouping is on: local % 8 == blockIdx.x % 8)
    if (!pw_tile_rows<BP>(g.a[p], blockIdx.x - g.start[p], g.gx[p], m0, mend, by)) return;
    pw_body<BP, BC, WP, WC, CONV, BK, PF, false, FK>(g.a[p], m0, mend, by);
}

// ---------------------------------------------------------------------------------------------------------------
// X-stationary variant for the mid/small layers (M <= ~130k rows, any K <= 1024, any N).
// The tiled kernel above pays one exposed memory latency per 32-deep K stage; with K up to 672 and few workgroups that
// is 20+ us of pure latency. Here a workgroup stages its whole [BP][K] pixel strip in LDS once (every load in flight
// together), then each wave walks its own 32-channel tiles streaming the weight rows straight from L2 into MFMA
// A-fragments (16 B per lane, 256 contiguous bytes per weight row per 8 k-steps), double-buffered so the next chunk's
// loads fly under the current chunk's MFMAs. No barrier after the staging one.
// ---------------------------------------------------------------------------------------------------------------
template <int BP>
__global__ __launch_bounds__(256) void pw_xs_kernel(PwArgs a, int tiles) {
    constexpr int TP = BP / 32;
    constexpr int KC = 8;                       // k-steps (of 16) per weight chunk
    extern __shared__ __attribute__((aligned(16))) half_t xs[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int K = a.cin, NC = a.cout;
    const int KS = (K + 15) >> 4;
    const int KP = KS * 16 + 8;                 // row stride (halfs): odd number of 16-B slots -> conflict-free b128 reads
    int m0, M, by_unused;                       // this workgroup's rows [m0, min(m0 + BP, M)) (XCD grouping: pw_tile_rows)
    if (!pw_tile_rows<BP>(a, blockIdx.x, tiles, m0, M, by_unused)) return;

    // ---- stage the pixel strip (zero-filled beyond M / K), 8 independent 16-B loads per thread per batch
    const int CPR = KS * 2;                     // 16-B chunks per row (incl. zero padding up to KS*16)
    const int total = BP * CPR;
    // (round 3: the SE scales of a chunk are requested TOGETHER with the chunk, clamped and unconditional -- they used to be loaded behind the wait for the
    //  batch's rows: a second dependent round trip per batch)
    const bool has_se = a.se != nullptr;
    for (int c0 = 0; c0 < total; c0 += 256 * 8) {
        uint4 v[8];
        float4 s0[8], s1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + u * 256 + tid;
            const int row = c / CPR, q = c - row * CPR;
            const int m = m0 + row, k = q * 8;
            uint4 t = make_uint4(0, 0, 0, 0);
            if (c < total && m < M && k < K) t = *reinterpret_cast<const uint4*>(a.x + (size_t)m * K + k);
            v[u] = t;
            if (has_se) {
                const float* sp = a.se + (size_t)(min(m, M - 1) / a.hw) * K + min(k, K - 8);
                s0[u] = *reinterpret_cast<const float4*>(sp);
                s1[u] = *reinterpret_cast<const float4*>(sp + 4);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + u * 256 + tid;
            if (c < total) {
                const int row = c / CPR, q = c - row * CPR;
                uint4 t = v[u];
                if (has_se) {
                    const int m = m0 + row, k = q * 8;
                    if (m < M && k < K) {
                        half8 hv = *reinterpret_cast<half8*>(&t);
                        hv[0] = (half_t)((float)hv[0] * s0[u].x); hv[1] = (half_t)((float)hv[1] * s0[u].y); hv[2] = (half_t)((float)hv[2] * s0[u].z); hv[3] = (half_t)((float)hv[3] * s0[u].w);
                        hv[4] = (half_t)((float)hv[4] * s1[u].x); hv[5] = (half_t)((float)hv[5] * s1[u].y); hv[6] = (half_t)((float)hv[6] * s1[u].z); hv[7] = (half_t)((float)hv[7] * s1[u].w);
                        t = *reinterpret_cast<uint4*>(&hv);
                    }
                }
                *reinterpret_cast<uint4*>(&xs[row * KP + q * 8]) = t;
            }
        }
    }
    __syncthreads();

    size_t obase[TP];
    bool mvalid[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int m = m0 + j * 32 + r;
        mvalid[j] = m < M;
        if (a.out_fp32) {
            const int img = m / a.hw;
            obase[j] = (size_t)a.out_base + (size_t)img * a.out_img_stride + (size_t)(m - img * a.hw) * NC;
        } else {
            obase[j] = (size_t)m * NC;
        }
    }

    // weights come from the fragment-major copy (plan.py): [channel tile][K step][lane][8 halfs], zero rows beyond cout -- every
    // wave-wide load is 1 KB contiguous and needs no per-lane predicate (K is zero-padded to a multiple of 16 in the copy, as in the strip)
    auto load_w = [&](half8 (&wf)[KC], int nt, int ks0) {
        const half_t* wr = a.wfrag + (size_t)nt * KS * 512 + lane * 8;
#pragma unroll
        for (int u = 0; u < KC; ++u) {
            half8 t = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ks0 + u < KS) t = *reinterpret_cast<const half8*>(wr + (size_t)(ks0 + u) * 512);
            wf[u] = t;
        }
    };

    const int NT = (NC + 31) >> 5;
    half8 wa[KC], wb[KC];
    if (wave < NT) load_w(wa, wave, 0);
    for (int nt = wave; nt < NT; nt += 4) {
        // bias / residual of this channel tile: issued now, consumed after the K loop
        float4 bv[4];
        half4 resv[TP][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = nt * 32 + 8 * g + 4 * hh;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c0 + 3 < NC) t = *reinterpret_cast<const float4*>(a.bias + c0);
            else {
                if (c0 < NC) t.x = a.bias[c0];
                if (c0 + 1 < NC) t.y = a.bias[c0 + 1];
                if (c0 + 2 < NC) t.z = a.bias[c0 + 2];
            }
            bv[g] = t;
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                half4 rv = {0, 0, 0, 0};
                if (a.residual && mvalid[j] && c0 < NC) rv = *reinterpret_cast<const half4*>(a.residual + obase[j] + c0);
                resv[j][g] = rv;
            }
        }
        floatx16 acc[TP];
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

        for (int ks0 = 0; ks0 < KS; ks0 += 2 * KC) {
            // chunk A is loaded; prefetch chunk B (next KC k-steps, or the next tile's first chunk)
            const bool moreB = ks0 + KC < KS;
            if (moreB) load_w(wb, nt, ks0 + KC);
            else if (nt + 4 < NT) load_w(wb, nt + 4, 0);
#pragma unroll
            for (int u = 0; u < KC; ++u) {
                if (ks0 + u < KS) {
#pragma unroll
                    for (int j = 0; j < TP; ++j) {
                        const half8 xf = *reinterpret_cast<const half8*>(&xs[(j * 32 + r) * KP + (ks0 + u) * 16 + hh * 8]);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[u], xf, acc[j], 0, 0, 0);
                    }
                }
            }
            if (moreB) {
                const bool moreA = ks0 + 2 * KC < KS;
                if (moreA) load_w(wa, nt, ks0 + 2 * KC);
                else if (nt + 4 < NT) load_w(wa, nt + 4, 0);
#pragma unroll
                for (int u = 0; u < KC; ++u) {
                    if (ks0 + KC + u < KS) {
#pragma unroll
                        for (int j = 0; j < TP; ++j) {
                            const half8 xf = *reinterpret_cast<const half8*>(&xs[(j * 32 + r) * KP + (ks0 + KC + u) * 16 + hh * 8]);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[u], xf, acc[j], 0, 0, 0);
                        }
                    }
                }
            } else {
                // the prefetched chunk (next tile's first) sits in wb: make it the A buffer of the next tile
#pragma unroll
                for (int u = 0; u < KC; ++u) wa[u] = wb[u];
            }
        }
        // epilogue
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            if (!mvalid[j]) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = nt * 32 + 8 * g + 4 * hh;
                if (c0 >= NC) continue;
                float v[4] = {acc[j][4 * g + 0] + bv[g].x, acc[j][4 * g + 1] + bv[g].y, acc[j][4 * g + 2] + bv[g].z,
                              acc[j][4 * g + 3] + bv[g].w};
                dn_act_n<float[4], 4>(v, a.act);
                if (a.out_fp32) {
                    float* o = reinterpret_cast<float*>(a.out) + obase[j] + c0;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c0 + e < NC) o[e] = v[e];
                } else {
                    if (a.residual) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)resv[j][g][e];
                    }
                    half4 hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) hv[e] = (half_t)v[e];
                    *reinterpret_cast<half4*>(reinterpret_cast<half_t*>(a.out) + obase[j] + c0) = hv;
                }
            }
        }
    }
}

template <int BP>
int launch_xs(const PwArgs& a, hipStream_t s) {
    const int KS = (a.cin + 15) / 16;
    const size_t lds = (size_t)BP * (KS * 16 + 8) * sizeof(half_t);
    DN_HIP_CHECK(dn_allow_big_lds(reinterpret_cast<const void*>(pw_xs_kernel<BP>)));
    dn_note_kernel("pw_xs_kernel<%d>", BP);
    const int tiles = pw_row_tiles(a, BP);
    hipLaunchKernelGGL((pw_xs_kernel<BP>), dim3(a.xq > 0 ? 8 * tiles : tiles), dim3(256), lds, s, a, tiles);
    return DN_OK;
}

template <int BP, int BC, int WP, int WC, bool CONV, int BK, int PF = 1, bool SEF = false>
int launch_bk(const PwArgs& a, hipStream_t s, int nbuf) {
    const int tiles = pw_row_tiles(a, BP);
    dim3 grid((a.xq > 0 ? 8 * tiles : tiles) * dn_cdiv(a.cout, BC));
    size_t halfs = (size_t)nbuf * (BP + BC) * (BK + 8);
    const size_t otile = (a.out_fp32 || a.residual) ? (size_t)2 * BP * (BC + 4) : (size_t)BP * (BC + 8);     // epilogue staging tile, in halfs (fp32 for head rows and for residual layers)
    if (otile > halfs) halfs = otile;
    const size_t lds = halfs * sizeof(half_t) + BC * sizeof(float) + (SEF ? 832 * sizeof(float) : 0);
    if (lds > 64 * 1024) DN_HIP_CHECK(dn_allow_big_lds(reinterpret_cast<const void*>(pw_kernel<BP, BC, WP, WC, CONV, BK, PF, SEF>)));
    dn_note_kernel(SEF ? "pw_kernel<%d,%d,%d,%d,%s,%d,%d,true>" : PF > 1 ? "pw_kernel<%d,%d,%d,%d,%s,%d,%d>" : "pw_kernel<%d,%d,%d,%d,%s,%d>", BP, BC, WP, WC,
                   CONV ? "true" : "false", BK, PF);
    if constexpr (!CONV && !SEF && (PF == 3 || PF == 4)) {
        // the bound-test-free K loop (pw_body FK): whole 32-deep stages, no SE-scaled staging, 32-bit byte offsets
        if (dn_knob("DN_PW_FASTK", 1) && a.cin % BK == 0 && (!a.se || a.hw >= BP) && !(a.act >> 8) && (size_t)a.m * a.cin < (1u << 30) && (size_t)a.cout * a.cin < (1u << 30)) {
            const size_t lds_fk = lds + (a.se ? (size_t)2 * a.cin * sizeof(float) : 0);      // + the SE scales of the tile's two images
            if (lds_fk > 64 * 1024) DN_HIP_CHECK(dn_allow_big_lds(reinterpret_cast<const void*>(pw_kernel<BP, BC, WP, WC, CONV, BK, PF, SEF, true>)));
            hipLaunchKernelGGL((pw_kernel<BP, BC, WP, WC, CONV, BK, PF, SEF, true>), grid, dim3(256), lds_fk, s, a, tiles);
            return DN_OK;
        }
    }
    hipLaunchKernelGGL((pw_kernel<BP, BC, WP, WC, CONV, BK, PF, SEF>), grid, dim3(256), lds, s, a, tiles);
    return DN_OK;
}

// 32-deep double-buffered K staging. (Measured and dropped: a single exact-K stage for K <= 128 -- no load/compute overlap,
// slower; 64-deep double buffer -- lost to occupancy.)
template <int BP, int BC, int WP, int WC, bool CONV>
int launch_cfg(const PwArgs& a, hipStream_t s) {
    const_cast<PwArgs&>(a).stamps = g_pw_stamps;
    if constexpr (CONV && BP == 128 && BC == 128) {
        // MFMA-bound dense convolutions (VGG): 64-deep stages halve the barriers per MFMA; the tile is register-limited to two
        // workgroups per CU either way, and 2 x 74 KB of LDS fit
        const int bk64 = 1;
        if (bk64 && a.cv_cin % 64 == 0) return launch_bk<BP, BC, WP, WC, CONV, 64>(a, s, 2);
    }
    return launch_bk<BP, BC, WP, WC, CONV, 32>(a, s, 2);
}


}  // namespace

// Tile choice: these GEMMs are latency/HBM-bound, not MFMA-bound, so what matters is (a) enough workgroups to fill
// 256 CUs several times over and (b) not re-reading x for many channel tiles. Prefer the largest tile that still
// gives >= ~1500 workgroups, else fall back to smaller tiles.
template <bool CONV>
int launch_select(const PwArgs& a, hipStream_t s) {
    auto wgs = [&](int bp, int bc) { return (long)dn_cdiv(a.m, bp) * dn_cdiv(a.cout, bc); };
    switch (g_pw_tile) {
        case 1: return launch_cfg<256, 32, 4, 1, CONV>(a, s);
        case 2: return launch_cfg<128, 32, 4, 1, CONV>(a, s);
        case 3: return launch_cfg<128, 64, 4, 1, CONV>(a, s);
        case 4: return launch_cfg<64, 64, 2, 2, CONV>(a, s);
        case 5: return launch_cfg<128, 128, 2, 2, CONV>(a, s);
        case 6: return launch_cfg<64, 128, 2, 2, CONV>(a, s);
        case 7: if constexpr (!CONV) return launch_cfg<128, 96, 4, 1, CONV>(a, s); break;
        default: break;
    }
    if constexpr (!CONV) {
        // 1x1 convs with a thin side (cin < 256 or cout < 128) are HBM/latency-bound: tools/tune_pw.py over every layer shape
        // of the model shows the small tiles (most workgroups, fewest registers: 64 VGPRs -> 8 waves/SIMD) winning or tying
        // everywhere, 128x32 when there is a single channel tile. The big tiles only pay off for MFMA-bound shapes.
        if (a.cin < 256 || a.cout < 128) {
            const int shortk = 1;
            if (shortk && a.cin > 32 && a.cin <= 128) {       // 2..4 K stages: all loads up front
                const_cast<PwArgs&>(a).stamps = g_pw_stamps;
                if constexpr (!CONV) {
                    if (a.sef_part) return launch_bk<64, 64, 2, 2, CONV, 32, 4, true>(a, s, 2);     // squeeze-excitation folded in (pw_se_fold_supported)
                }
                if (a.cout <= 32) return launch_bk<128, 32, 4, 1, CONV, 32, 4>(a, s, 2);
                return launch_bk<64, 64, 2, 2, CONV, 32, 4>(a, s, 2);
            }
            if (shortk && a.cin > 128) {                      // long K on a thin layer: loads 4 stages ahead
                const_cast<PwArgs&>(a).stamps = g_pw_stamps;
                if (a.cout <= 32) return launch_bk<128, 32, 4, 1, CONV, 32, 4>(a, s, 2);
                return launch_bk<64, 64, 2, 2, CONV, 32, 4>(a, s, 2);
            }
            if (a.cout <= 32) return launch_cfg<128, 32, 4, 1, CONV>(a, s);
            return launch_cfg<64, 64, 2, 2, CONV>(a, s);
        }
    }
    if constexpr (CONV) {
        const int big = dn_knob("DN_CONV_BIG", 1);
        const int bigmin = 40;       // measured on both VGG models: 40 < 90 < 200; the sub-batch chains fill the chip together
        if (big && conv_big_supported(a) && wgs(256, 256) >= bigmin) return launch_conv_big(a, s);
    }
    if (a.cout <= 32) {
        if (wgs(256, 32) >= 1500) return launch_cfg<256, 32, 4, 1, CONV>(a, s);
        return launch_cfg<128, 32, 4, 1, CONV>(a, s);
    }
    if (a.cout <= 64) {
        if (wgs(128, 64) >= 1500) return launch_cfg<128, 64, 4, 1, CONV>(a, s);
        return launch_cfg<64, 64, 2, 2, CONV>(a, s);
    }
    if constexpr (CONV) {
        // small dense convs (the extras of the VGG models: <= 16 x 16 maps, K = 9 cin up to 4608): a few workgroups walking 70 - 140
        // K stages, each an exposed memory round trip with the plain double buffer (60 - 130 us per layer for < 1 us of MFMA work):
        // request the stages 4 ahead through the register ring of the short-K pointwise variant
        if (a.cout > 32 && wgs(64, 64) < 512) return launch_bk<64, 64, 2, 2, CONV, 32, 4>(a, s, 2);
    }
    if constexpr (!CONV) {
        // a 1x1 layer with few workgroups and a long K (the first extras layer: 480 -> 256 on 10 x 10) is a chain of exposed round trips with the plain
        // double buffer: stages requested 4 ahead on the bound-test-free loop. Measured (round 3): the launch 18.8 -> 14 us, one forward at a time -5 us,
        // with three forwards in flight 0.2 - 0.3 % slower in three of three pairs: opt-in
        if (dn_knob("DN_PW_LONGK_PF", 0) && a.cin % 32 == 0 && a.cin >= 256 && a.cout > 32 && wgs(64, 64) < 1500) {
            const_cast<PwArgs&>(a).stamps = g_pw_stamps;
            return launch_bk<64, 64, 2, 2, CONV, 32, 4>(a, s, 2);
        }
    }
    const int t128 = 300;      // min workgroups for the 128x128 tile of the MFMA-bound dense convs (measured on the VGG models)
    if (wgs(128, 128) >= (CONV ? t128 : 1500)) return launch_cfg<128, 128, 2, 2, CONV>(a, s);
    if (wgs(128, 64) >= 1500 || a.cout % 128 > 64 || a.cout % 128 == 0) {
        if (wgs(64, 128) >= 600) return launch_cfg<64, 128, 2, 2, CONV>(a, s);
    }
    return launch_cfg<64, 64, 2, 2, CONV>(a, s);
}

namespace {
template <int BP, int BC, int WP, int WC, bool CONV, int BK = 32, int GPF_ = 0>
int launch_group_cfg(const PwArgs* arr, int count, hipStream_t s) {
    // head 1x1 convs have long K (21 stages at level 0): loads run 3 stages ahead (the dense-conv heads of the VGG models are
    // MFMA-bound at 184 VGPRs and keep the plain double buffer -- except the small levels, below)
    constexpr int GPF = GPF_ ? GPF_ : (CONV ? 1 : 3);
    PwGroup g{};
    g.count = count;
    int acc = 0;
    bool all_xq = true;         // XCD grouping needs every problem's first workgroup at a multiple of 8: all problems or none
    for (int i = 0; i < count; ++i) all_xq &= arr[i].xq > 0;
    for (int i = 0; i < count; ++i) {
        g.a[i] = arr[i];
        if (!all_xq) g.a[i].xq = 0;
        g.a[i].stamps = g_pw_stamps;        // dev hook (null unless a probe set it)
        g.start[i] = acc;
        g.gx[i] = pw_row_tiles(g.a[i], BP);
        const int ctiles = dn_cdiv(arr[i].cout, BC);
        acc += (g.a[i].xq > 0 ? 8 * g.gx[i] : g.gx[i]) * ctiles;
    }
    g.start[count] = acc;
    size_t halfs = (size_t)2 * (BP + BC) * (BK + 8);
    bool any_fp32 = false;
    for (int i = 0; i < count; ++i) any_fp32 |= arr[i].out_fp32 != 0 || arr[i].residual != nullptr;
    size_t otile = any_fp32 ? (size_t)2 * BP * (BC + 4) : (size_t)BP * (BC + 8);
    if (otile > halfs) halfs = otile;
    const size_t lds = halfs * sizeof(half_t) + BC * sizeof(float);
    dn_note_kernel(GPF > 1 ? "pw_group_kernel<%d,%d,%d,%d,%s,%d,%d>" : "pw_group_kernel<%d,%d,%d,%d,%s,%d>", BP, BC, WP, WC, CONV ? "true" : "false", BK, GPF);
    if constexpr (!CONV && GPF == 3) {
        // every problem on the bound-test-free K loop (whole 32-deep stages, no SE-scaled staging, 32-bit byte offsets): the instantiation that holds nothing else
        bool fk = dn_knob("DN_PW_FASTK", 1) != 0;
        for (int i = 0; i < count; ++i)
            fk &= arr[i].cin % BK == 0 && !arr[i].se && !(arr[i].act >> 8) && (size_t)arr[i].m * arr[i].cin < (1u << 30) && (size_t)arr[i].cout * arr[i].cin < (1u << 30);
        if (fk) {
            if (lds > 64 * 1024) DN_HIP_CHECK(dn_allow_big_lds(reinterpret_cast<const void*>(pw_group_kernel<BP, BC, WP, WC, CONV, BK, GPF, true>)));
            hipLaunchKernelGGL((pw_group_kernel<BP, BC, WP, WC, CONV, BK, GPF, true>), dim3(acc), dim3(256), lds, s, g);
            return DN_OK;
        }
    }
    if (lds > 64 * 1024) DN_HIP_CHECK(dn_allow_big_lds(reinterpret_cast<const void*>(pw_group_kernel<BP, BC, WP, WC, CONV, BK, GPF>)));
    hipLaunchKernelGGL((pw_group_kernel<BP, BC, WP, WC, CONV, BK, GPF>), dim3(acc), dim3(256), lds, s, g);
    return DN_OK;
}
}  // namespace

// All problems must be of the same kind (pointwise or implicit-GEMM conv); the tile is chosen for the widest one.
int launch_pointwise_group(const PwArgs* arr, int count, bool conv, hipStream_t s) {
    DN_REQUIRE(count >= 1 && count <= 12, "pointwise group: %d problems", count);
    int maxc = 0;
    long wg128 = 0;
    for (int i = 0; i < count; ++i) {
        DN_REQUIRE(arr[i].cin % 8 == 0 && arr[i].m > 0, "pointwise group: bad problem %d", i);
        DN_REQUIRE(!conv || arr[i].cv_cin % 32 == 0, "conv group: cin=%d must be a multiple of 32", arr[i].cv_cin);
        if (arr[i].cout > maxc) maxc = arr[i].cout;
        wg128 += (long)dn_cdiv(arr[i].m, 128) * dn_cdiv(arr[i].cout, 128);
    }
    if (maxc <= 32 && conv && wg128 < 256) return launch_group_cfg<128, 32, 4, 1, true, 32, 3>(arr, count, s);
    if (maxc <= 32) return conv ? launch_group_cfg<128, 32, 4, 1, true>(arr, count, s) : launch_group_cfg<128, 32, 4, 1, false>(arr, count, s);
    if (maxc <= 64) return conv ? launch_group_cfg<64, 64, 2, 2, true>(arr, count, s) : launch_group_cfg<64, 64, 2, 2, false>(arr, count, s);
    if (wg128 >= 1500 && conv) {
        // MFMA-bound dense-conv heads: 64-deep stages as in launch_cfg (half the barriers per MFMA)
        const int bk64 = 1;
        bool all64 = bk64 != 0;
        for (int i = 0; i < count; ++i) all64 &= arr[i].cv_cin % 64 == 0;
        if (all64) return launch_group_cfg<128, 128, 2, 2, true, 64>(arr, count, s);
    }
    if (wg128 >= 1500 && !conv) {
        // 96-wide channel tiles (a wave = 32 pixels x 96 channels) where they pad less: the 546 class channels of the SSDLite heads are
        // 6 x 96 = 576 columns instead of 5 x 128 = 640 -- the head launch is the longest full-chip launch of a forward (batch 64, three
        // forwards in flight: 0.789 -> 0.775 ms; 128 x 192 tiles 0.808, 64 x 192 level)
        double c96 = 0, c128 = 0;
        for (int i = 0; i < count; ++i) {
            c96 += (double)arr[i].m * dn_cdiv(arr[i].cout, 96) * 96;
            c128 += (double)arr[i].m * dn_cdiv(arr[i].cout, 128) * 128;
        }
        if (c96 <= 0.95 * c128) return launch_group_cfg<128, 96, 4, 1, false>(arr, count, s);
    }
    if (wg128 >= 1500) return conv ? launch_group_cfg<128, 128, 2, 2, true>(arr, count, s) : launch_group_cfg<128, 128, 2, 2, false>(arr, count, s);
    // the dense heads of the small levels (a few dozen workgroups, 72 - 144 K stages): latency-bound, stages requested 3 ahead
    if (conv && wg128 < 256) return launch_group_cfg<64, 128, 2, 2, true, 32, 3>(arr, count, s);
    return conv ? launch_group_cfg<64, 128, 2, 2, true>(arr, count, s) : launch_group_cfg<64, 128, 2, 2, false>(arr, count, s);
}

// the squeeze-excitation of a projection can be computed in the projection kernel's prologue (SEF variant of the 64 x 64 tile)
bool pw_se_fold_supported(int cin, int cout, int squeeze, int hw) {
    return dn_knob("DN_SE_FOLD", 1) != 0 && cin > 32 && cin <= 128 && cin % 8 == 0 && squeeze <= 32 && hw >= 64 &&
           (cin < 256 || cout < 128);
}

int launch_pointwise(const PwArgs& a, hipStream_t s) {
    DN_REQUIRE(!a.sef_part || (!a.se && pw_se_fold_supported(a.cin, a.cout, a.sef_sq, a.hw)), "pointwise: squeeze-excitation fold not supported for this shape");
    DN_REQUIRE(a.cin % 8 == 0, "pointwise: cin=%d must be a multiple of 8", a.cin);
    DN_REQUIRE(a.out_fp32 || a.cout % 4 == 0, "pointwise: fp16 cout=%d must be a multiple of 4", a.cout);
    DN_REQUIRE(a.m > 0 && a.hw > 0, "pointwise: empty problem");
    if (!g_pw_tile && pw_direct_supported(a)) return launch_pw_direct(a, s);
    const int xs_mode = dn_knob("DN_PW_XS", 1);
    if (g_pw_tile == 8 && a.wfrag) return launch_xs<32>(a, s);
    if (xs_mode && !g_pw_tile && !a.sef_part && a.wfrag && a.cin % 16 == 0 && a.cin <= 1024 && a.cout <= 160 && !(a.act >> 8) &&      // (K % 16 == 8: measured slower than the tiled kernel)
        ((a.cin >= 64 && a.m <= 8192) || (a.cin >= 160 && a.m <= 16384))) {
        // measured (tools/tune_pw.py): the strip kernel wins where the tiled kernel cannot fill the chip -- M <= ~8k rows, or
        // M <= ~16k rows when K is long (the tiled kernel pays one exposed round trip per 32-deep K stage)
        return launch_xs<32>(a, s);
    }
    return launch_select<false>(a, s);
}

PwArgs conv_to_pw(const ConvArgs& c) {
    PwArgs a;
    a.cv_k = c.k; a.cv_stride = c.stride; a.cv_pad = c.pad; a.cv_dil = c.dil; a.cv_h = c.h; a.cv_w = c.w_;
    a.cv_ho = c.ho; a.cv_wo = c.wo; a.cv_cin = c.cin;
    a.fd_cin32 = fastdiv((unsigned)(c.cin >> 5));
    a.fd_k = fastdiv((unsigned)c.k);
    a.x = c.x; a.w = c.w; a.bias = c.bias; a.residual = nullptr; a.se = nullptr; a.out = c.out; a.zeros = c.zeros;
    a.hw = c.ho * c.wo;
    a.m = c.n * a.hw;
    a.cin = c.k * c.k * c.cin;
    a.cout = c.cout; a.act = c.act; a.out_fp32 = c.out_fp32; a.out_img_stride = c.out_img_stride; a.out_base = c.out_base;
    a.xq = c.xq;
    return a;
}

int launch_conv(const ConvArgs& c, hipStream_t s) {
    DN_REQUIRE(c.cin % 32 == 0, "conv: cin=%d must be a multiple of 32 (implicit-GEMM K stage within one tap)", c.cin);
    DN_REQUIRE(c.out_fp32 || c.cout % 4 == 0, "conv: fp16 cout=%d must be a multiple of 4", c.cout);
    PwArgs a;
    a.cv_k = c.k; a.cv_stride = c.stride; a.cv_pad = c.pad; a.cv_dil = c.dil; a.cv_h = c.h; a.cv_w = c.w_;
    a.cv_ho = c.ho; a.cv_wo = c.wo; a.cv_cin = c.cin;
    a.fd_cin32 = fastdiv((unsigned)(c.cin >> 5));
    a.fd_k = fastdiv((unsigned)c.k);
    a.x = c.x; a.w = c.w; a.bias = c.bias; a.residual = nullptr; a.se = nullptr; a.out = c.out; a.zeros = c.zeros;
    a.hw = c.ho * c.wo;
    a.m = c.n * a.hw;
    a.cin = c.k * c.k * c.cin;
    a.cout = c.cout; a.act = c.act; a.out_fp32 = c.out_fp32; a.out_img_stride = c.out_img_stride; a.out_base = c.out_base;
    a.xq = c.xq;
    DN_REQUIRE(a.m > 0, "conv: empty problem");
    if (c.k == 1 && c.stride == 1 && c.pad == 0 && !c.out_fp32) {
        // a 1x1 dense conv IS a pointwise conv: unless it is big enough for the 256 x 256-tile kernel, the pointwise path serves it
        // (register-direct kernel up to cin = 256, 4-stage prefetch ring beyond; the implicit-GEMM body walks it one exposed stage at a time)
        const long wg256 = (long)dn_cdiv(a.m, 256) * dn_cdiv(a.cout, 256);
        if (!(dn_knob("DN_CONV_BIG", 1) && conv_big_supported(a) && wg256 >= 40)) {
            PwArgs b;
            b.x = c.x; b.w = c.w; b.bias = c.bias; b.residual = nullptr; b.se = nullptr; b.out = c.out;
            b.hw = a.hw; b.m = a.m; b.cin = c.cin; b.cout = c.cout; b.act = c.act; b.out_fp32 = 0; b.out_img_stride = 0; b.out_base = 0;
            b.xq = c.xq;
            return launch_pointwise(b, s);
        }
    }
    return launch_select<true>(a, s);
}


