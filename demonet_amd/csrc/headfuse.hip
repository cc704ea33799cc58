// SSDLite prediction heads of ALL pyramid levels, both heads, in ONE launch: depthwise 3x3 + BN + ReLU6 computed inside the operand
// staging of the 1x1 GEMM behind it (round 4).
//
// reference ops replaced: `_prediction_block` (ssd_mobilenetv3.py:27-36: depthwise 3x3 ConvBNActivation -> 1x1 conv) of the class and
// the box head of every level, and the view / permute / reshape / cat of generalized_ssd.py:60-74 (the fp32 rows leave in final layout).
// Before: dw_group_kernel wrote both heads' depthwise outputs (83 MB per 64 images) for pw_group_kernel to read back, 128 x 96 tiles
// re-reading their pixel rows once per channel tile.
//
// CDNA4 mapping. A workgroup (4 waves) owns 64 consecutive pixels of the flattened (image, y, x) order and ALL output channels of both
// heads (546 + 24 -> 18 + 1 channel tiles of 32): wave w accumulates channel tiles w, w + 4, ... x two 32-pixel tiles (<= 160 fp32
// accumulator registers), so a pixel's depthwise values are computed ONCE per head. K (the feature channels) is walked in chunks of 32:
//   1. the chunk's input rows -- the tile's 64 pixels plus W + 1 pixels either side, ONE run that holds all nine taps of every pixel --
//      go to LDS by LDS-DMA, pixel-major with the four 8-channel groups of a pixel XOR-swizzled over its four 16-byte slots (16 whole
//      cache lines per DMA instruction, conflict-free tap reads);
//   2. depthwise: thread = (8-channel group = wave, pixel = lane) reads its nine taps once (addresses formed once per workgroup; a tap
//      outside the image reads a zero slot, which is the zero padding), and runs both heads over them: bias first, taps in (ky, kx)
//      order through v_fma_mix_f32 (fp16 operands, fp32 accumulate), ReLU6, ONE rounding to fp16 -- the arithmetic of dw_kernel bit for
//      bit. The 8-channel group is wave-uniform, so a head's depthwise weights and bias are the same for the wave's 64 pixels: the class
//      head's come through the scalar cache into SGPR operands (44 SGPRs, requested one phase ahead -- both heads that way would need 88
//      live at once), the box head's ride in the LDS-DMA of the x run (704 B per chunk) and are read as wave-wide broadcasts;
//   3. the fp16 values go to LDS in the same plane layout, where they ARE the B fragments of v_mfma_f32_32x32x16_f16 (lane = pixel,
//      8 consecutive k per half wave); the A fragments come straight from L2 out of the fragment-major weight copy (1 KB per wave load,
//      requested before the depthwise phase). Accumulation order = pw_group_kernel's (one accumulator per output, K ascending in 16-deep
//      steps, bias added in fp32 at the end): logits and box regressions are bit-identical to the two-launch path.
// One barrier per chunk (B tiles and x runs double-buffered). Epilogue: straight from the accumulators, 32 contiguous bytes per pixel and store.
//
// Round 5, SM = true: the first launch of the post-process in the epilogue (generalized_ssd.py:354 softmax, :362-363 decode + clip). The
// workgroup holds every class and box channel of its 64 pixels, so each (pixel, anchor) row of K logits is complete in it: per 32-pixel half
// tile the accumulators (+ bias) go to LDS as [pixel][channel] fp32 rows (70 KB for 6 x 91 classes: the main loop's buffers are dead by then),
// four lanes per row compute max / exp / sum exactly as softmax_decode_kernel does (post_math.h: same code, same bits), the scores leave
// class-major -- a half tile's 32 x A_l rows are consecutive canonical anchors, 768-byte runs per class -- the boxes are decoded from the 4 A_l
// regression channels, and the score histogram (+ per-class counts) of every (half tile, image) pair goes to a row of its own that tau_kernel
// adds up (HistRows; no atomics in memory, nothing to clear). The fp32 logits of these levels (73 of 79 MB per 64 images) are neither written nor
// read back, and softmax_decode_kernel's launch is gone. Only the levels with >= 32 pixels per image take this epilogue (HeadFuseLevel::sm): a
// workgroup with it lives ~85 us at batch 64 whatever its level, the launch is 543 workgroups on 512 residency slots, and late starters with the
// epilogue were a 65 us tail. The small levels (7 % of the anchors) keep the plain logit epilogue, are launched last, and get their softmax in
// the cut-off launch that follows (tau_kernel, TauSmall). SM = false (dn_forward_heads, DN_HEAD_SOFTMAX=0) writes every level's logits.
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "post_math.h"

namespace {

constexpr int HF_P = 64;            // pixels per workgroup
constexpr int HF_RUNP = 128;        // run pixels per plane (64 + 2 (W + 1) <= 128: W <= 31)
constexpr int HF_ZOFF = 4 * HF_RUNP * 16;         // zero slot inside an x buffer
constexpr int HF_WOFF = HF_ZOFF + 64;             // 1 KB slot: the box head's depthwise weights of the chunk [4 groups][9][8] halfs, its bias [32] floats, the class head's bias [32] floats
constexpr int HF_XS = HF_WOFF + 1024;             // bytes per x buffer: four planes + the zero slot + the weight slot
constexpr int HF_BS = 4 * HF_P * 16;              // bytes per B tile (one head, one buffer)
constexpr int HF_LDS = 2 * HF_XS + 4 * HF_BS;     // 18 560 + 16 384 B
constexpr int HF_NP = 2;            // softmax epilogue: a 32-pixel half tile of a level with >= 32 pixels per image touches at most 2 images
// LDS of the softmax epilogue: [32][nc0] + [32][nc1] fp32 rows, row sum / row offset / anchor tables of 256 entries, HF_NP histograms, 256 part bytes
constexpr int hf_post_tabs(int nc0, int nc1) {       // (at least the main loop's buffers: what follows -- the biases -- is written while the loop runs)
    const int t = 32 * (nc0 + nc1) * 4 + 3 * 256 * 4 + HF_NP * 256 * 4 + 256;
    return t > HF_LDS ? t : HF_LDS;
}
// ... + both heads' biases (each padded to a multiple of 4 floats), copied in at the start of the workgroup behind everything the main loop uses
constexpr int hf_post_lds(int nc0, int nc1) { return hf_post_tabs(nc0, nc1) + (((nc0 + 3) & ~3) + ((nc1 + 3) & ~3)) * 4; }

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));      // a head row starts at an 8-byte boundary
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) u32x4* cp_u4;
typedef const __attribute__((address_space(4))) u32x16* cp_u16;

struct HfLevel {
    const half_t* x;                 // [n][H][W][C]
    const half_t* wd;                // the class head's depthwise weights, group-major [C / 8][9 taps][8] fp16
    const unsigned char* wslot;      // [C / 32][1 KB]: the box head's depthwise weights + bias and the class head's bias of every chunk (HeadFuseLevel::wslot)
    const half_t* wf[2];             // 1x1 weights, fragment-major [ceil(nc/32)][C/16][2][32][8] (plan.py::fragment_major)
    const float* bias[2];            // [nc]
    float* out[2];                   // logits / box regressions, element offset of image 0 already added
    unsigned out_img_stride[2];      // elements between images
    int nc[2];
    int H, W, C, hw, m;              // m = n * hw
    int act;                         // depthwise activation
    int tiles;                       // 64-pixel tiles per XCD group; 0: this level uses the plain mapping (tiles of the whole level)
    int ct0;                         // channel tiles of the class head; the box head is tile ct0
    int aoff, aloc, sbase;           // softmax epilogue: first anchor of the level, anchors per location, first histogram-row slot (HistRows)
    int sm;                          // this level's workgroups run the softmax epilogue (SM instantiations; levels with >= 32 pixels per image)
};
struct HfGroup {
    int count, xq;
    int start[9];
    long long* stamps;               // dev builds only (-DDN_DEV_STAMPS, tools/probe_headfuse.py): per-workgroup phase cycle sums
    HeadPost post;                   // SM instantiations only
    HfLevel lv[8];
};
#ifdef DN_DEV_STAMPS
#define HF_T(var) const long long var = (long long)__builtin_amdgcn_s_memtime()
#else
#define HF_T(var)
#endif

// acc[0..7] += x[0..7] * w[0..7]: fp16 x in a VGPR quad, fp16 w in an SGPR quad (uniform per wave), fp32 accumulate
__device__ __forceinline__ void hf_fma8(float (&acc)[8], const u32x4& x, const u32x4& w) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(acc[2 * i]) : "v"(x[i]), "s"(w[i]));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc[2 * i + 1]) : "v"(x[i]), "s"(w[i]));
    }
}

__device__ __forceinline__ void hf_fma8v(float (&acc)[8], const u32x4& x, const u32x4& w) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(acc[2 * i]) : "v"(x[i]), "v"(w[i]));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc[2 * i + 1]) : "v"(x[i]), "v"(w[i]));
    }
}

// Softmax rows AND scores in one pass (round 6). Until then a half tile's rows were walked twice: pp_softmax_row replaced the logits in LDS by exp(x - max)
// (two reads and a write per element), a barrier, then hf_scores read every element again, multiplied, stored (in-kernel stamps: 34 k + 35 k of a
// workgroup's 150 k cycles). Here the four lanes of a row keep their 23 exponentials in registers: one LDS read per element, the row's maximum and sum
// through the same two xor-shuffles in the same order (pp_softmax_row's arithmetic, bit for bit), then score = e * (1 / sum) straight to memory -- a store
// instruction writes four classes x 16 consecutive anchors (64 contiguous bytes each). The per-class counts of passing scores, ballots in hf_scores (a
// class belonged to one wave there), are LDS atomics here like the histogram bins: passing scores are a few per cent.
__device__ __forceinline__ void hf_softmax_scores(const HeadPost& P, const float* __restrict__ lg, const unsigned* __restrict__ rowoff, const unsigned* __restrict__ ranc,
                                                  const unsigned char* __restrict__ rpart, unsigned* __restrict__ lhist, const int nrows, const int img_first,
                                                  const int te, const int K, const int ccb) {
#pragma clang fp contract(off)
    const int Km1 = K - 1, sub = te & 3;
    char* const sbase = reinterpret_cast<char*>(P.scoresT);
    const unsigned step4 = 4u * (unsigned)P.A;              // bytes between class k and class k + 1
    for (int rb = 0; rb < nrows; rb += 64) {
        const int row = rb + (te >> 2);
        const bool valid = row < nrows;
        const float* rp = lg + (valid ? rowoff[row] : 0u);
        float mx = -INFINITY, sm = 0.f;
        float v[24];                                        // element 8 b + u <-> class k = sub + 32 b + 4 u (pp_softmax_row's batches)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[8 * b + u] = rp[min(sub + 32 * b + 4 * u, K - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) mx = fmaxf(mx, v[8 * b + u]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 1));
        mx = fmaxf(mx, __shfl_xor(mx, 2));
#pragma unroll
        for (int b = 0; b < 3; ++b) {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[8 * b + u] = pp_exp_nonpos(v[8 * b + u] - mx);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (sub + 32 * b + 4 * u < K) sm += v[8 * b + u];
        }
        sm += __shfl_xor(sm, 1);
        sm += __shfl_xor(sm, 2);
        if (!valid) continue;
        const float rinv = pp_row_rcp(sm);
        // (the row's tables are read HERE, behind the exponentials: held across them they cost the one register the 256-register budget does not have)
        int rw = row;
        asm volatile("" : "+v"(rw));
        const unsigned q = rpart[rw];
        const unsigned hb = q * 256u;
        const unsigned ob4 = (((unsigned)(img_first + (int)q) * (unsigned)Km1) * (unsigned)P.A + ranc[rw]) * 4u + (unsigned)(sub - 1) * step4;     // class sub - 1 (sub = 0: the background, skipped)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = sub + 32 * b + 4 * u;
                if (k >= 1 && k < K) {
                    const float sc = pp_score(v[8 * b + u], rinv);
                    *reinterpret_cast<float*>(sbase + (unsigned)(ob4 + (unsigned)(8 * b + u) * 4u * step4)) = sc;       // (32-bit sum first: for sub = 0 ob4 is "class -1" modulo 2^32)
                    if (sc > P.score_thr) {
                        atomicAdd(&lhist[hb + (unsigned)pp_hist_bin(sc, P.hb0, P.nb)], 1u);
                        if (ccb >= 0) atomicAdd(&lhist[hb + (unsigned)(ccb + k - 1)], 1u);
                    }
                }
            }
    }
}

template <int TCW, bool SM>
__global__ __launch_bounds__(256, 2) void head_fused_kernel(HfGroup g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hf_lds[];
    int p = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i)
        if (i < g.count && (int)blockIdx.x >= g.start[i]) p = i;
    const HfLevel& L = g.lv[p];
    const int flat = blockIdx.x - g.start[p];
    int m0, mend;
    if (g.xq > 0 && L.tiles > 0) {
        const int grp = flat & 7, t = flat >> 3;
        const int r0 = grp * g.xq * L.hw;
        mend = min(L.m, r0 + g.xq * L.hw);
        m0 = r0 + t * HF_P;
    } else {
        m0 = flat * HF_P;
        mend = L.m;
    }
    if (m0 >= mend) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int C = L.C, W = L.W, H = L.H;
    const int NCH = C >> 5;

    // ---- per-workgroup constants of the depthwise phase: the nine tap addresses of (group = wave, pixel = lane)
    // (three 10-bit slot numbers per register: nine registers held across the loop were what spilled -- and a scratch reload is a VMEM
    //  operation that waits in order behind the A-fragment requests)
    unsigned tap3[3] = {0u, 0u, 0u};
    {
        const int mp = min(m0 + lane, mend - 1);
        const int img = mp / L.hw, rem = mp - img * L.hw;
        const int y = rem / W, x = rem - y * W;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = y + ky - 1, ix = x + kx - 1;
                const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
                const int P = lane + (W + 1) + (ky - 1) * W + (kx - 1);          // run pixel of the tap
                const unsigned slot = ok ? (unsigned)(4 * P + (wave ^ ((P >> 2) & 3))) : (unsigned)(HF_ZOFF / 16);
                tap3[ky] |= slot << (10 * kx);
            }
    }
    auto tap_addr = [&](int t) -> unsigned { return ((tap3[t / 3] >> (10 * (t % 3))) & 1023u) * 16u; };
    // x run in LDS: [run pixel][4 slots of 16 B], the 8-channel group g of run pixel P in slot g ^ ((P >> 2) & 3) -- a wave's tap read (64 consecutive
    // pixels, one group) then touches every bank once, and one LDS-DMA instruction covers 16 whole pixels = 16 cache lines (as four planes it was 64
    // lines per instruction: the launch was bound by L1 tag lookups). Wave w fills run pixels [32 w, 32 w + 32) with two instructions; run pixels are
    // clamped into the level (a tap that is used never reads a clamped pixel).
    unsigned xoff[2];                // byte offsets from L.x (the launcher checks the level is < 2 GB)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int P = (2 * wave + i) * 16 + (lane >> 2);
        const int mp = min(max(m0 - (W + 1) + P, 0), L.m - 1);
        xoff[i] = ((unsigned)mp * (unsigned)C + (unsigned)(((lane & 3) ^ ((P >> 2) & 3)) * 8)) * 2u;
    }
    if (tid < 8) *reinterpret_cast<unsigned*>(hf_lds + (tid >> 2) * HF_XS + HF_ZOFF + (tid & 3) * 4) = 0u;
    // LDS-DMA behind the compiler's back (inline asm): as a builtin, hipcc orders every later ds_read behind the DMA's LDS write with
    // s_waitcnt vmcnt(0) -- the taps of chunk c would wait for the run of chunk c + 1. The DMA is published by the counted wait + barrier
    // at the end of the depthwise phase (below); its target buffer was last read before the previous barrier.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)hf_lds;
    auto glds16 = [&](const void* gsrc, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
    };
    const unsigned aoff = (unsigned)lane * 16u;
    const unsigned char* const xbase = reinterpret_cast<const unsigned char*>(L.x);
    auto stage_x = [&](int c) {
        const unsigned dst = lds0 + (c & 1) * HF_XS;
        glds16(xbase + (xoff[0] + (unsigned)c * 64u), dst + wave * 2048);
        glds16(xbase + (xoff[1] + (unsigned)c * 64u), dst + wave * 2048 + 1024);
        // the chunk's weight slot (832 B of its 1 KB in the blob: box-head depthwise weights + bias, class-head bias), one instruction of wave 3
        if (wave == 3) glds16(L.wslot + ((unsigned)c * 1024u + aoff), dst + HF_WOFF);
    };

    // channel tiles of this wave: wave + 4 i. Tiles i < TCW - 1 are class tiles; the last one is a class tile, the box tile or nothing.
    const int tlast = wave + 4 * (TCW - 1);
    const bool last_reg = tlast == L.ct0, last_on = tlast <= L.ct0;
    const int KS = C >> 4;
    // A fragments: [tile][K step][lane][8] halfs. Buffer loads: descriptor of the head's weight copy (SGPRs), tile + K-step offset as the scalar
    // offset, lane * 16 B as the ONE vector offset of every load (as global loads hipcc kept a 64-bit address pair per tile alive: 10 registers)
    const __amdgpu_buffer_rsrc_t ars0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(L.wf[0]), 0, L.ct0 * KS * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t ars1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(L.wf[1]), 0, KS * 1024, 0x00020000);
    unsigned atile[TCW];             // byte offset of the tile's first K step (uniform)
#pragma unroll
    for (int i = 0; i < TCW; ++i) {
        const int t = wave + 4 * i;
        const bool reg = (i == TCW - 1) && last_reg;
        atile[i] = reg ? 0u : (unsigned)(min(t, L.ct0 - 1) * KS) * 1024u;
    }
    auto load_a = [&](int i, int kstep) -> half8 {
        const u32x4 v = ((i == TCW - 1) && last_reg) ? __builtin_amdgcn_raw_buffer_load_b128(ars1, aoff, atile[i] + (unsigned)kstep * 1024u, 0)
                                                     : __builtin_amdgcn_raw_buffer_load_b128(ars0, aoff, atile[i] + (unsigned)kstep * 1024u, 0);
        return __builtin_bit_cast(half8, v);
    };
    const unsigned blast = last_reg ? (unsigned)HF_BS : 0u;       // B tile of the last channel tile: the box head's or the class head's

    floatx16 acc[TCW][2];
#pragma unroll
    for (int i = 0; i < TCW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // the class head's depthwise weights of a chunk: 144 contiguous bytes of the group-major copy -> three scalar loads (uniform address)
    u32x16 w0a, w0b;
    u32x4 w0c;
    auto load_w0 = [&](int c) {
        const unsigned char* wp = reinterpret_cast<const unsigned char*>(L.wd) + (size_t)(c * 4 + wave) * 144;
        w0a = *(cp_u16)(wp);
        w0b = *(cp_u16)(wp + 64);
        w0c = *(cp_u4)(wp + 128);
    };
    auto w0tap = [&](int t) -> u32x4 {
        u32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int d = 4 * t + e; r[e] = d < 16 ? w0a[d] : (d < 32 ? w0b[d - 16] : w0c[d - 32]); }
        return r;
    };

#ifdef DN_DEV_STAMPS
    long long st_dw = 0, st_wait = 0, st_mm = 0;
    const long long st_begin = (long long)__builtin_amdgcn_s_memtime();
    const long long st_begin_rt = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (SM) if (L.sm) {
        // both heads' biases -> LDS (the epilogue adds them to 64 x 570 accumulators; read from memory there, their latency was exposed in front of
        // its first barrier). Published by the barrier below, far from the buffers of the main loop.
        float* const lb = reinterpret_cast<float*>(hf_lds + hf_post_tabs(L.nc[0], L.nc[1]));
        const int p0 = (L.nc[0] + 3) & ~3;
        for (int i = tid; i < L.nc[0]; i += 256) lb[i] = L.bias[0][i];
        if (tid < L.nc[1]) lb[p0 + tid] = L.bias[1][tid];
    }
    stage_x(0);
    load_w0(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    HF_T(st_loop);

    unsigned char* const bb = hf_lds + 2 * HF_XS;
    for (int c = 0; c < NCH; ++c) {
        const int PAR = c & 1;
        HF_T(st0);
        // this chunk's A fragments, first K step: requested now, consumed behind the depthwise phase
        half8 af0[TCW], af1[TCW];
#pragma unroll
        for (int i = 0; i < TCW; ++i) af0[i] = load_a(i, 2 * c);
        // ---- depthwise of both heads over the chunk's 8-channel group `wave`, pixel `lane`: kernel row by kernel row (three taps and the
        //      box head's three weight rows in registers at a time)
        {
            const unsigned char* xb = hf_lds + PAR * HF_XS;
            const unsigned char* wb = xb + HF_WOFF + wave * 144;
            // (opaque per iteration: hipcc otherwise hoists the nine unpacked tap addresses out of the loop -- nine registers again)
            asm volatile("" : "+v"(tap3[0]), "+v"(tap3[1]), "+v"(tap3[2]));
            const float4 bl4 = *reinterpret_cast<const float4*>(xb + HF_WOFF + 576 + wave * 32);
            const float4 bh4 = *reinterpret_cast<const float4*>(xb + HF_WOFF + 576 + wave * 32 + 16);
            const float4 cl4 = *reinterpret_cast<const float4*>(xb + HF_WOFF + 704 + wave * 32);
            const float4 ch4 = *reinterpret_cast<const float4*>(xb + HF_WOFF + 704 + wave * 32 + 16);
            float a0[8] = {cl4.x, cl4.y, cl4.z, cl4.w, ch4.x, ch4.y, ch4.z, ch4.w}, a1[8] = {bl4.x, bl4.y, bl4.z, bl4.w, bh4.x, bh4.y, bh4.z, bh4.w};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                u32x4 tap[3], wv[3];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    tap[kx] = *reinterpret_cast<const u32x4*>(xb + tap_addr(ky * 3 + kx));
                    wv[kx] = *reinterpret_cast<const u32x4*>(wb + (ky * 3 + kx) * 16);
                }
                // The LDS-DMA of the next chunk goes out HERE, not at the top of the loop: hipcc waits for "every load older than this chunk's A
                // requests" before it reuses their registers for the first LDS reads (a loop-header scoreboard merge) -- a wait that would
                // otherwise sit on the DMA it cannot see.
                if (ky == 0 && c + 1 < NCH) stage_x(c + 1);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) hf_fma8(a0, tap[kx], w0tap(ky * 3 + kx));      // class head: weights in SGPRs
                if (ky == 2) {
                    // second K step's A fragments: requested as soon as the last kernel row's operands are in registers
#pragma unroll
                    for (int i = 0; i < TCW; ++i) af1[i] = load_a(i, 2 * c + 1);
                }
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) hf_fma8v(a1, tap[kx], wv[kx]);                  // box head: weights broadcast from LDS
                // keep the next row's 24 registers of LDS reads behind this row's multiply-adds: hoisted (hipcc does, for latency) they spill
                // the tap addresses, and a scratch reload is a VMEM operation that waits in order behind the A-fragment requests
                __builtin_amdgcn_sched_barrier(0);
            }
            unsigned char* bout = bb + PAR * 2 * HF_BS + wave * (HF_P * 16) + lane * 16;
            half8 o0, o1;
#pragma unroll
            for (int e = 0; e < 8; ++e) { o0[e] = (half_t)dn_relu6(a0[e]); o1[e] = (half_t)dn_relu6(a1[e]); }
            *reinterpret_cast<half8*>(bout) = o0;
            *reinterpret_cast<half8*>(bout + HF_BS) = o1;
        }
        HF_T(st1);
        // the LDS-DMA of chunk c + 1 is older than the TCW loads requested behind it (the second K step's A fragments): all but those are complete
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(TCW) : "memory");
        __syncthreads();            // B tiles of chunk c written, x run + box-head weights of chunk c + 1 landed
        HF_T(st2);
        load_w0(min(c + 1, NCH - 1));       // next chunk's scalar operands: in flight under the matrix phase
        // ---- GEMM over the chunk: two 16-deep steps
        {
            const unsigned char* bc = bb + PAR * 2 * HF_BS;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                half8 bf[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const half8*>(bc + (unsigned)((2 * s + hh) * (HF_P * 16) + (32 * j + r) * 16));
#pragma unroll
                for (int i = 0; i < TCW - 1; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(s ? af1[i] : af0[i], bf[j], acc[i][j], 0, 0, 0);
                if (last_on) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const half8 bl = *reinterpret_cast<const half8*>(bc + blast + (unsigned)((2 * s + hh) * (HF_P * 16) + (32 * j + r) * 16));
                        acc[TCW - 1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(s ? af1[TCW - 1] : af0[TCW - 1], bl, acc[TCW - 1][j], 0, 0, 0);
                    }
                }
            }
        }
#ifdef DN_DEV_STAMPS
        asm volatile("" : "+v"(acc[0][0]));
        const long long st3 = (long long)__builtin_amdgcn_s_memtime();
        st_dw += st1 - st0; st_wait += st2 - st1; st_mm += st3 - st2;
#endif
    }
#ifdef DN_DEV_STAMPS
    const long long st_end_loop = (long long)__builtin_amdgcn_s_memtime();
#endif
    // the lane's coordinates for the epilogues. SM instantiations take them from threadIdx.x again, opaque to value numbering: otherwise hipcc keeps
    // the main loop's copies alive across it for the epilogue -- registers the 256-register budget at TCW = 5 does not have (a spill; the lint
    // refuses scratch in this kernel)
    int te = threadIdx.x;
    if constexpr (SM) asm volatile("" : "+v"(te));
    const int lane_e = SM ? (te & 63) : lane, r_e = SM ? (lane_e & 31) : r, hh_e = SM ? (lane_e >> 5) : hh;
    if constexpr (SM) if (L.sm) {
        // ---- softmax + decode epilogue (file header). Lane (r, hh) holds pixel r of each pixel tile and channels 8 g + 4 hh .. + 3 of each channel tile.
        const HeadPost& P = g.post;
        const int lane = lane_e, r = r_e, hh = hh_e;
        const int K = P.K, Km1 = K - 1, AL = L.aloc, NC0 = L.nc[0], NC1 = L.nc[1], hw = L.hw;
        float* const lg = reinterpret_cast<float*>(hf_lds);                        // [32][NC0] logits, then exp(x - max)
        float* const rgt = lg + 32 * NC0;                                           // [32][NC1] box regressions
        float* const rowsum = rgt + 32 * NC1;                                       // [256]
        unsigned* const rowoff = reinterpret_cast<unsigned*>(rowsum + 256);         // [256] pixel * NC0 + anchor * K
        unsigned* const ranc = rowoff + 256;                                        // [256] canonical anchor index of the row
        unsigned* const lhist = ranc + 256;                                         // [HF_NP][256]
        unsigned char* const rpart = reinterpret_cast<unsigned char*>(lhist + HF_NP * 256);      // [256] the row's image - first image of the half tile
        const int ccb = (P.nb + Km1 <= 256) ? P.nb : -1;                            // per-class counts of passing scores in the free bins [nb, nb + K - 1)
        const bool grouped = g.xq > 0 && L.tiles > 0;
        const int img_g0 = grouped ? (flat & 7) * g.xq : 0;                         // first image / pixel of this workgroup's XCD group (plain mapping: 0)
        const int r0 = img_g0 * hw;
        __syncthreads();            // every wave is done with the x runs and B tiles: the LDS belongs to the epilogue now
#ifdef DN_DEV_STAMPS
        long long st_w = 0, st_s = 0, st_p = 0;
#endif
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int mj = m0 + 32 * j;
            if (mj >= mend) break;
            HF_T(se0);
            const int npx = min(32, mend - mj), nrows = npx * AL;
            const int img_first = mj / hw, img_last = (mj + npx - 1) / hw;
            // (W) accumulators + bias -> [pixel][channel] rows (float2 pieces: a row starts at an 8-byte boundary, nc is even)
#pragma unroll
            for (int i = 0; i < TCW; ++i) {
                const bool reg = (i == TCW - 1) && last_reg;
                if (i == TCW - 1 && !last_on) break;
                const int ct = reg ? 0 : wave + 4 * i;
                const int nc = reg ? NC1 : NC0;
                const float* bias = reinterpret_cast<const float*>(hf_lds + hf_post_tabs(NC0, NC1)) + (reg ? ((NC0 + 3) & ~3) : 0);      // (LDS copy)
                float* drow = (reg ? rgt : lg) + r * nc;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int ch = 32 * ct + 8 * gq + 4 * hh;
                    // (a group of four is one 16-byte LDS read; the pad behind a head's last channel is never stored)
                    const float4 b4 = *reinterpret_cast<const float4*>(bias + min(ch, ((nc + 3) & ~3) - 4));
                    const float v[4] = {acc[i][j][4 * gq] + b4.x, acc[i][j][4 * gq + 1] + b4.y, acc[i][j][4 * gq + 2] + b4.z, acc[i][j][4 * gq + 3] + b4.w};
                    if (ch + 4 <= nc) {
                        *reinterpret_cast<float2*>(drow + ch) = make_float2(v[0], v[1]);
                        *reinterpret_cast<float2*>(drow + ch + 2) = make_float2(v[2], v[3]);
                    } else if (ch + 2 <= nc) {
                        *reinterpret_cast<float2*>(drow + ch) = make_float2(v[0], v[1]);
                    }
                }
            }
            for (int t = te; t < HF_NP * 256; t += 256) lhist[t] = 0u;
            // row tables: row = pixel * AL + anchor -- consecutive rows are consecutive canonical anchors
            int bx_px = 0, bx_a = 0, bx_img = 0, bx_anc = 0;
            if (te < nrows) {
                bx_px = te / AL; bx_a = te - bx_px * AL;
                const int m = mj + bx_px;
                bx_img = m / hw;
                bx_anc = L.aoff + (m - bx_img * hw) * AL + bx_a;
                rowoff[te] = (unsigned)(bx_px * NC0 + bx_a * K);
                ranc[te] = (unsigned)bx_anc;
                rpart[te] = (unsigned char)(bx_img - img_first);
            }
            __syncthreads();
            HF_T(se1);
            // (B) decode_single + clip of the row's box (_utils.py:187-224): the anchor load is in flight under the softmax
            if (te < nrows) {
                const float4 rg4 = *reinterpret_cast<const float4*>(rgt + bx_px * NC1 + bx_a * 4);
                const float4 an = reinterpret_cast<const float4*>(P.anchors)[bx_anc];
                P.boxes[(size_t)bx_img * P.A + bx_anc] = pp_decode_box(rg4, an, P.img_w, P.img_h);
            }
            // (S + P) softmax rows and scores in one pass: four adjacent lanes per row, 64 rows per pass (hf_softmax_scores)
            hf_softmax_scores(P, lg, rowoff, ranc, rpart, lhist, nrows, img_first, te, K, ccb);
            __syncthreads();
            HF_T(se2);
            // the rows of this half tile's (at most two) images: slot = half tiles of the range between the image's first one and this one
#pragma unroll
            for (int qq = 0; qq < HF_NP; ++qq) {
                const int img = img_first + qq;
                if (img <= img_last) {
                    const int slot = ((mj - r0) >> 5) - (((img - img_g0) * hw) >> 5);
                    P.hrows[((size_t)img * P.rows_per_image + L.sbase + slot) * 256 + te] = (te < P.nb + (ccb >= 0 ? Km1 : 0)) ? lhist[qq * 256 + te] : 0u;
                }
            }
            __syncthreads();        // (the next half tile rewrites the rows and tables)
#ifdef DN_DEV_STAMPS
            const long long se3 = (long long)__builtin_amdgcn_s_memtime();
            st_w += se1 - se0; st_s += se2 - se1; st_p += se3 - se2;
#endif
        }
#ifdef DN_DEV_STAMPS
        if (g.stamps && te == 0) {
            long long* sp = g.stamps + (size_t)blockIdx.x * 16;
            sp[8] = st_begin_rt;
            sp[9] = (long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
            sp[10] = (long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
            sp[0] = st_loop - st_begin; sp[1] = st_dw; sp[2] = st_wait; sp[3] = st_mm; sp[4] = st_end_loop - st_loop;
            sp[5] = (long long)__builtin_amdgcn_s_memtime() - st_end_loop; sp[6] = NCH; sp[7] = (long long)__builtin_amdgcn_s_memrealtime();
            sp[11] = st_w; sp[12] = st_s; sp[13] = st_p;
        }
#endif
        return;
    }
    // ---- epilogue: straight from the accumulators. Lane (r_e, hh_e) holds pixel r_e of each pixel tile and channels 8 g + 4 hh_e .. + 3 of each channel
    // tile in registers 4 g .. 4 g + 3: one 16-byte store per (tile, g) -- the two half waves write adjacent pieces, 32 contiguous bytes per
    // pixel and instruction. (Through a per-wave LDS slab as row-contiguous float2 runs it was 80 dependent LDS round trips per wave: 20 000
    // cycles of a 100 000-cycle workgroup.) Rows are 8-byte aligned only (nc * 4 B = 2184): global_store_dwordx4 needs dword alignment.
    unsigned ro[2][2];
    bool rok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + 32 * j + r_e;
        const int img = m / L.hw, pix = m - img * L.hw;
        rok[j] = m < mend;
#pragma unroll
        for (int h = 0; h < 2; ++h) ro[h][j] = (unsigned)img * L.out_img_stride[h] + (unsigned)pix * (unsigned)L.nc[h];
    }
#pragma unroll
    for (int i = 0; i < TCW; ++i) {
        const bool reg = (i == TCW - 1) && last_reg;
        if (i == TCW - 1 && !last_on) break;
        const int ct = reg ? 0 : wave + 4 * i;
        const int nc = reg ? L.nc[1] : L.nc[0];
        const float* bias = reg ? L.bias[1] : L.bias[0];
        float* outp = reg ? L.out[1] : L.out[0];
        float bv[16];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[4 * gq + e] = bias[min(32 * ct + 8 * gq + 4 * hh_e + e, nc - 1)];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float* orow = outp + (reg ? ro[1][j] : ro[0][j]);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int ch = 32 * ct + 8 * gq + 4 * hh_e;
                floatx4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * gq + e] + bv[4 * gq + e];
                if (rok[j]) {
                    if (ch + 4 <= nc) *reinterpret_cast<f32x4_a8*>(orow + ch) = v;
                    else if (ch + 2 <= nc) *reinterpret_cast<float2*>(orow + ch) = make_float2(v[0], v[1]);       // (nc even, nc % 4 == 2: the last pair)
                }
            }
        }
    }
#ifdef DN_DEV_STAMPS
    if (g.stamps && tid == 0) {
        long long* sp = g.stamps + (size_t)blockIdx.x * 16;
        sp[8] = st_begin_rt;
        sp[9] = (long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));       // HW_REG_HW_ID
        sp[10] = (long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));     // HW_REG_XCC_ID
        sp[0] = st_loop - st_begin; sp[1] = st_dw; sp[2] = st_wait; sp[3] = st_mm; sp[4] = st_end_loop - st_loop;
        sp[5] = (long long)__builtin_amdgcn_s_memtime() - st_end_loop; sp[6] = NCH; sp[7] = (long long)__builtin_amdgcn_s_memrealtime();
    }
#endif
}

}  // namespace

// ---- host side -------------------------------------------------------------------------------------------------------------------
bool head_fused_level_supported(const HeadFuseLevel& l) {
    if (l.C % 32 != 0 || l.C < 32 || l.W > 31 || l.W < 1 || l.H < 1 || l.act != DN_ACT_RELU6) return false;
    if (l.nc[1] > 32 || l.nc[0] < 1 || l.nc[1] < 1 || !l.wdg || !l.wslot) return false;
    for (int h = 0; h < 2; ++h) {
        // rows leave as float2 runs: every row start 8-byte aligned
        if ((l.nc[h] & 1) || (l.out_img_stride[h] & 1) || (l.out_base[h] & 1) || (reinterpret_cast<size_t>(l.out[h]) & 7)) return false;
        if ((unsigned long long)l.n * (unsigned long long)l.out_img_stride[h] >= 0xffffffffull) return false;
        if (!l.wf[h] || !l.bias[h]) return false;
    }
    const int tiles = dn_cdiv(l.nc[0], 32) + 1;
    return dn_cdiv(tiles, 4) <= 5 && (unsigned long long)l.n * l.H * l.W * l.C < 0x80000000ull;
}

#ifdef DN_DEV_STAMPS
static long long* g_hf_stamps = nullptr;
extern "C" __attribute__((visibility("default"))) void dn_debug_hf_stamps(void* dev_ptr) { g_hf_stamps = (long long*)dev_ptr; }
#endif
static int g_head_fused_launches = 0, g_head_softmax_launches = 0;
extern "C" __attribute__((visibility("default"))) int dn_debug_head_fused_launches() { return g_head_fused_launches; }      // tests: the path was taken
extern "C" __attribute__((visibility("default"))) int dn_debug_head_softmax_launches() { return g_head_softmax_launches; }  // ... with the softmax / decode epilogue

bool head_fused_post_supported(const HeadFuseLevel* lv, int count, const HeadPost& post) {
    if (!post.scoresT || !post.boxes || !post.hrows || !post.anchors || post.K < 2 || post.A < 1 || post.nb < 1 || post.nb > 256) return false;
    int rows = 0, nsm = 0;
    for (int i = 0; i < count; ++i) {
        const HeadFuseLevel& l = lv[i];
        if (!l.sm) continue;
        ++nsm;
        if (l.H * l.W < 32) return false;                                                                       // (a half tile of such a level spans more than two images)
        if (l.aloc < 1 || l.aloc > 8 || l.nc[0] != l.aloc * post.K || l.nc[1] != 4 * l.aloc) return false;       // 32 pixels x aloc rows fit the 256-entry tables
        if (hf_post_lds(l.nc[0], l.nc[1]) > 80 * 1024) return false;                                            // two workgroups per CU stay resident
        if (l.sbase != rows) return false;
        rows += hist_rows_slots(l.H * l.W);
        if ((unsigned long long)l.n * (post.K - 1) * post.A >= 0x3fffffffull) return false;       // byte offsets of the scores in 32 bits
    }
    return nsm > 0 && rows <= post.rows_per_image;      // (rows behind these belong to the softmax tiles of the logit-writing levels)
}

int launch_head_fused(const HeadFuseLevel* lv, int count, int xq, hipStream_t s, const HeadPost* post) {
    DN_REQUIRE(count >= 1 && count <= 8, "fused heads: %d levels", count);
    HfGroup g{};
    g.count = count; g.xq = xq;
    int lds = HF_LDS;
    if (post) {
        DN_REQUIRE(head_fused_post_supported(lv, count, *post), "fused heads: softmax epilogue not supported for these levels");
        g.post = *post;
        for (int i = 0; i < count; ++i)
            if (lv[i].sm) lds = std::max(lds, hf_post_lds(lv[i].nc[0], lv[i].nc[1]));
    }
    // Launch order: the levels with the softmax epilogue first, longest reduction first (their workgroups live longest); the small, logit-writing
    // levels last -- they are the ones that may start late (543 workgroups on 512 residency slots at batch 64) and they are short.
    int ord[8];
    for (int i = 0; i < count; ++i) ord[i] = i;
    if (post)
        std::stable_sort(ord, ord + count, [&](int x, int y) {
            if (lv[x].sm != lv[y].sm) return lv[x].sm > lv[y].sm;
            return lv[x].sm ? lv[x].C > lv[y].C : false;
        });
    int acc = 0, tcw = 0;
    for (int i = 0; i < count; ++i) {
        const HeadFuseLevel& l = lv[ord[i]];
        DN_REQUIRE(head_fused_level_supported(l), "fused heads: level %d unsupported (C=%d W=%d nc=%d/%d)", ord[i], l.C, l.W, l.nc[0], l.nc[1]);
        HfLevel& d = g.lv[i];
        d.x = l.x; d.wd = l.wdg; d.wslot = l.wslot;
        for (int h = 0; h < 2; ++h) {
            d.wf[h] = l.wf[h]; d.bias[h] = l.bias[h];
            d.out[h] = l.out[h] + l.out_base[h];
            d.out_img_stride[h] = (unsigned)l.out_img_stride[h];
            d.nc[h] = l.nc[h];
        }
        d.H = l.H; d.W = l.W; d.C = l.C; d.hw = l.H * l.W; d.m = l.n * d.hw; d.act = l.act;
        d.ct0 = dn_cdiv(l.nc[0], 32);
        d.aoff = l.aoff; d.aloc = l.aloc; d.sbase = l.sbase; d.sm = post ? l.sm : 0;
        const int t4 = dn_cdiv(d.ct0 + 1, 4);
        DN_REQUIRE(i == 0 || t4 == tcw, "fused heads: levels differ in channel tiles per wave (%d vs %d)", t4, tcw);
        tcw = t4;
        // XCD grouping (a group of xq images per workgroup residue mod 8) only where a group fills whole tiles: on the small maps the ragged last
        // tile of every group is most of the level (5 x 5: 32 workgroups instead of 25) and every workgroup beyond the 512 resident ones starts
        // a second round of the launch
        const bool grouped = head_fused_grouped(xq, d.hw);      // (xq * hw >= 4 tiles)
        d.tiles = grouped ? dn_cdiv((long)xq * d.hw, HF_P) : 0;
        g.start[i] = acc;
        acc += grouped ? d.tiles * 8 : dn_cdiv(d.m, HF_P);
    }
    g.start[count] = acc;
#ifdef DN_DEV_STAMPS
    g.stamps = g_hf_stamps;
#endif
    dn_note_kernel(post ? "head_fused_kernel<%d,softmax>" : "head_fused_kernel<%d>", tcw);
    ++g_head_fused_launches;
    if (post) ++g_head_softmax_launches;
    const dim3 grid(acc), block(256);
#define HF_GO(T)                                                                                                            \
    if (post) {                                                                                                             \
        DN_HIP_CHECK(dn_allow_big_lds(reinterpret_cast<const void*>(head_fused_kernel<T, true>), 160 * 1024));               \
        hipLaunchKernelGGL((head_fused_kernel<T, true>), grid, block, lds, s, g);                                           \
    } else {                                                                                                                \
        if (lds > 64 * 1024) DN_HIP_CHECK(dn_allow_big_lds(reinterpret_cast<const void*>(head_fused_kernel<T, false>), 160 * 1024)); \
        hipLaunchKernelGGL((head_fused_kernel<T, false>), grid, block, lds, s, g);                                          \
    }
    switch (tcw) {
        case 1: HF_GO(1); break;
        case 2: HF_GO(2); break;
        case 3: HF_GO(3); break;
        case 4: HF_GO(4); break;
        case 5: HF_GO(5); break;
        default: DN_REQUIRE(false, "fused heads: %d channel tiles per wave", tcw);
    }
#undef HF_GO
    return DN_OK;
}
