// Shared device/host helpers for the gfx950 SSD library. CDNA4 only: wave64, MFMA, no portability shims.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/demonet_hip.h"

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

void dn_set_error(const char* fmt, ...);
// label of the kernel the last launcher call on this thread enqueued (bench.py's roofline names; matches rocprofv3's kernel names)
void dn_note_kernel(const char* fmt, ...);
const char* dn_last_kernel();

#define DN_HIP_CHECK(expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            dn_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return DN_E_HIP;                                                                 \
        }                                                                                    \
    } while (0)

#define DN_REQUIRE(cond, ...)                                                                \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            dn_set_error(__VA_ARGS__);                                                       \
            return DN_E_INVALID;                                                             \
        }                                                                                    \
    } while (0)

// ReLU / ReLU6 in ONE instruction each. fmaxf / fminf make hipcc quiet a possible signalling NaN first (`v_max_f32 v, v, v`) whenever the operand does
// not come from an arithmetic instruction it knows -- MFMA results, inline-asm v_fma_mix_f32 sums -- i.e. two instructions per value in epilogues
// that are bound by vector-instruction issue (round 3: 28 000 such pairs in the expdw object alone). Same values for every non-NaN input.
// (round 6) ReLU is v_med3_f32(v, 0, FLT_MAX), a BUILTIN: until then it was an inline-asm `v_max_f32`, and an asm statement is opaque to hipcc's hazard
// recognizer -- gfx950 has no interlock between a matrix instruction's result write and a vector instruction that reads it (software keeps passes + 4
// wait states), so wherever the asm landed right behind an MFMA it could read stale accumulators (tools/mfma_hazard_scan.py; tests/test_isa_lint.py).
// FLT_MAX, not +inf: med3(v, 0, inf) is folded into max(v, 0), and fmaxf on a value hipcc does not know to be canonical costs a second instruction
// (`v_max_f32 v, v, v` first: the reason the asm existed). Same value as max(v, 0) for every finite input, -0.0 included (both give +0.0); +inf becomes
// FLT_MAX, which the fp16 stores of every caller round back to +inf.
__device__ __forceinline__ float dn_relu(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 3.4028234663852886e38f); }
__device__ __forceinline__ float dn_relu6(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 6.f); }

__device__ __forceinline__ float dn_act(float v, int act) {
    // Hardswish x*relu6(x+3)/6, ReLU6 clamp [0,6] (mobilenetv3.py:72, ssd_mobilenetv3.py:31).
    // Branch-free: `act` is wave-uniform (a kernel argument), so the bounds and the selector are scalar selects computed once, and
    // every value costs a fixed handful of VALU instructions. The obvious if-chain compiles to three scalar compare-and-branch pairs
    // PER VALUE (the epilogue of the 64 x 64 pointwise tile alone carried ~300 branch instructions, ~0.5 us of a 3.8 us workgroup).
    const float lo = (act == DN_ACT_RELU || act == DN_ACT_RELU6) ? 0.f : -INFINITY;
    const float hi = (act == DN_ACT_RELU6) ? 6.f : INFINITY;
    const float clamped = fminf(fmaxf(v, lo), hi);
    const float hs = v * fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);
    // (identity returns v itself: fmaxf / fminf would turn a NaN into -inf and hide a numerical failure from the NaN checks)
    return act == DN_ACT_HSWISH ? hs : (act == DN_ACT_NONE ? v : clamped);
}

// acc[0..7] += e[0..7] * w[0..7] with fp16 operands and fp32 accumulation in ONE instruction per element (v_fma_mix_f32).
// Left to itself hipcc converts every half to fp32 first (2 v_cvt per multiply-add): the depthwise kernels are VALU-heavy and
// that triples their instruction count.
__device__ __forceinline__ void fma_mix_h8(float (&acc)[8], const uint4& e, const uint4& w) {
    const unsigned ee[4] = {e.x, e.y, e.z, e.w}, ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(acc[2 * i]) : "v"(ee[i]), "v"(ww[i]));
        asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc[2 * i + 1]) : "v"(ee[i]), "v"(ww[i]));
    }
}

static inline int dn_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Environment knob, read on EVERY call -- never latched in a static -- so that a test can A/B two settings in one process
// (a new plan / the next eager launch sees the new value; a captured hipGraph keeps what it was captured with).
int dn_knob(const char* name, int dflt);
// hipFuncAttributeMaxDynamicSharedMemorySize = 160 KB, once per (device, kernel): the attribute is per device, so a process-wide
// flag would leave the second GPU of a process at the 64 KB default.
hipError_t dn_allow_big_lds(const void* kernel, int bytes = 160 * 1024);     // (kernels with static LDS pass 160 KB minus that)

// XCD affinity. Measured on MI355X (tools/xcd_probe.hip): workgroup b of a dispatch runs on XCD (b + x0) % 8 with x0 fixed per
// queue -- also for the parallel branches of a hipGraph -- and a kernel that reads what the SAME XCD's previous kernel wrote is
// served from that XCD's 4 MB L2 (2 - 2.5x faster on 8 - 32 MB hand-offs than reading another XCD's output through the fabric).
// So every kernel of the chain gives the workgroups with (flat index % 8) == g the images of group g = [g*q, (g+1)*q), q =
// ceil(n / 8): a layer then reads what its own XCD has just written. Speed only -- results never depend on the placement.
// Below 8 images there are fewer groups than XCDs and the plain mapping (all XCDs share every image) is kept.
int xcd_images_per_group(int n);       // 0: grouping off (n < 8 or DN_XCD=0)
// device side: flat workgroup index of a [8 groups][xq images][per_image] launch -> image and index within the image; false for
// the slots of images beyond n (n % 8 != 0). Plain mapping (xq == 0): flat = image * per_image + index.
__device__ __forceinline__ bool xcd_image_of(int flat, int per_image, int xq, int n, int& img, int& idx) {
    if (xq > 0) {
        const int g = flat & 7, w = flat >> 3;
        const int j = w / per_image;
        idx = w - j * per_image;
        img = g * xq + j;
        return img < n;
    }
    img = flat / per_image;
    idx = flat - img * per_image;
    return img < n;
}
static inline int xcd_image_slots(int xq, int n) { return xq > 0 ? 8 * xq : n; }     // image slots of such a launch
// 2-D form of the same mapping without the integer division (a runtime division costs ~25 VALU instructions, and the small
// kernels of the chain are VALU-issue bound): grid.x = [per_image indices] x 8 interleaved groups, grid.y = image slot of the
// group. Workgroups are dispatched x-fastest, so (blockIdx.x % 8) is still the XCD.
static inline dim3 xcd_grid2(int per_image, int xq, int n) { return xq > 0 ? dim3(8u * per_image, xq) : dim3(per_image, n); }
__device__ __forceinline__ bool xcd_image_of2(int xq, int n, int& img, int& idx) {
    if (xq > 0) {
        img = (blockIdx.x & 7) * xq + blockIdx.y;
        idx = blockIdx.x >> 3;
    } else {
        img = blockIdx.y;
        idx = blockIdx.x;
    }
    return img < n;
}
// Division by a launch-invariant divisor: q = umulhi(x, m), m = ceil(2^32 / d); exact for x * d * d < 2^32 (fd_ok).
struct FastDiv { unsigned d, m; };
static inline FastDiv fastdiv(unsigned d) { return FastDiv{d, d > 1 ? (unsigned)((0x100000000ull + d - 1) / d) : 0u}; }
static inline bool fd_ok(unsigned long long xmax, unsigned d) { return xmax * d * d < 0x100000000ull; }
__device__ __forceinline__ unsigned fd_div(unsigned x, FastDiv f) { return f.d > 1 ? __umulhi(x, f.m) : x; }
// activation of N values behind ONE uniform switch (a select chain per element costs ~8 VALU instructions each)
template <typename V, int N>
__device__ __forceinline__ void dn_act_n(V& v, int act) {
    if (act == DN_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = dn_relu(v[e]);
    } else if (act == DN_ACT_RELU6) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = dn_relu6(v[e]);
    } else if (act == DN_ACT_HSWISH) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = v[e] * dn_relu6(v[e] + 3.f) * (1.f / 6.f);
    }
}

// Epilogue of an MFMA tile: sink(i, j, g, act(acc[i][j][4 g ..] + bias)) for every 4-channel chunk the lane holds (channel tile i of TC, pixel
// tile j of TP; registers 4 g .. 4 g + 3 = channels 32 (wc TC + i) + 8 g + 4 hh ..). Per channel tile the lane's 16 bias values are read from LDS
// in ONE batch, and the activation is ONE uniform branch around the whole tile (round 5: the per-chunk form -- read the bias, wait, walk the
// activation switch -- made 4 TC TP dependent LDS round trips per tile; same arithmetic, bit-identical results).
template <int TP, int TC, typename ACC, typename SINK>
__device__ __forceinline__ void dn_tile_emit(const ACC& acc, const float* bsh, int wc, int hh, int act, SINK&& sink) {
    auto run = [&](auto actf) {
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            float4 bq[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const float4*>(&bsh[(wc * TC + i) * 32 + 8 * g + 4 * hh]);
#pragma unroll
            for (int j = 0; j < TP; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 v;
                    v.x = actf(acc[i][j][4 * g + 0] + bq[g].x); v.y = actf(acc[i][j][4 * g + 1] + bq[g].y);
                    v.z = actf(acc[i][j][4 * g + 2] + bq[g].z); v.w = actf(acc[i][j][4 * g + 3] + bq[g].w);
                    sink(i, j, g, v);
                }
        }
    };
    if (act == DN_ACT_RELU) run([](float v) { return dn_relu(v); });
    else if (act == DN_ACT_RELU6) run([](float v) { return dn_relu6(v); });
    else if (act == DN_ACT_HSWISH) run([](float v) { return v * dn_relu6(v + 3.f) * (1.f / 6.f); });
    else run([](float v) { return v; });
}

// launchers implemented by the per-kernel translation units (used by plan.hip and by the single-op C entry points)
struct PwArgs {
    // implicit-GEMM geometry (dense kxk conv); pointwise uses k=1: x rows are then simply [m][cin]
    int cv_k = 1, cv_stride = 1, cv_pad = 0, cv_dil = 1, cv_h = 0, cv_w = 0, cv_ho = 0, cv_wo = 0, cv_cin = 0;
    FastDiv fd_cin32{1, 0}, fd_k{1, 0};     // conv_to_pw: (k0 / 32) / (cv_cin / 32) and tap / cv_k without a hardware division per K stage
    const half_t* zeros = nullptr;   // optional: >= 16 zero bytes on the device (dense convs)
    int cv_plain_order = 0;          // dev knob DN_CONV_PLAIN_ORDER: conv_halo_kernel tiles in plain (x, y) order instead of XCD-grouped
    // optional second head on the same input (convbig.hip, head kernel): output channels [cout, cout + cout2) use these
    const half_t* w_b = nullptr; const float* bias_b = nullptr; void* out_b = nullptr;
    int cout_b = 0; long out_b_img_stride = 0, out_b_base = 0;
    const half_t* wfrag = nullptr;   // optional: the weights in MFMA-fragment order (dn_op_desc::w2_off), used by the strip kernel
    // squeeze-excitation folded into the projection (pointwise.hip, SEF variant): instead of `se`, the pooled partial sums of the
    // producing depthwise launch and the two FC weight sets; every workgroup computes the scale vector of its (at most two) images
    const float* sef_part = nullptr; int sef_nblk = 0, sef_sq = 0; float sef_inv = 0.f;
    const half_t* sef_w1t = nullptr; const half_t* sef_w2t = nullptr; const float* sef_b1 = nullptr; const float* sef_b2 = nullptr;
    half_t* pool_out = nullptr;      // conv_patch_kernel / conv_halo_kernel<3,4,4>: write the 2x2 / stride-2 max-pooled map [n][h/2][w/2][cout] instead of `out`
    const half_t* x;        // [m][cin]
    const half_t* w;        // [cout][cin]
    const float* bias;      // [cout]
    const half_t* residual; // [m][cout] or null
    const float* se;        // [m/hw][cin] or null
    void* out;
    int m, cin, cout, hw, act, out_fp32;
    long out_img_stride;    // elements between images in `out`
    long out_base;          // element offset of image 0 (head ops: level offset * columns)
    long long* stamps = nullptr;   // dev-only phase stamps
    int xq = 0;             // XCD grouping: images per group (0: plain mapping); see xcd_images_per_group
};
constexpr int DN_PP_HSHIFT = 19;      // score histogram: float bits 30..19 (8 exponent + 4 mantissa bits)
constexpr int DN_PP_HBINS = 256;      // bins kept: the top 256 (scores down to 2^-16); anything lower shares bin 0
int launch_pointwise(const PwArgs& a, hipStream_t s);
// register-direct schedule for short reductions (pwdirect.hip)
bool pw_direct_supported(const PwArgs& a);
int launch_pw_direct(const PwArgs& a, hipStream_t s);
bool pw_se_fold_supported(int cin, int cout, int squeeze, int hw);
int launch_pointwise_group(const PwArgs* arr, int count, bool conv, hipStream_t s);
// 256x256-tile implicit GEMM for the MFMA-bound dense convs (convbig.hip)
bool conv_big_supported(const PwArgs& a);
int launch_conv_big(const PwArgs& a, hipStream_t s);
// 3x3 "same" conv followed by MaxPool2d(2, 2) in one launch (the 16 x 16 output block of the patch kernel pools to 8 x 8 in its
// epilogue): true where launch_conv_big would take the patch kernel for this geometry at every batch size
bool conv_patch_pool_ok(int cin, int cout, int h, int w);
bool conv_pool_ok(int cin, int cout, int h, int w);             // conv 3x3 + MaxPool2d(2, 2) in one launch: the patch kernel or the run-staged 256 x 256 tile
bool conv_halo_pool_ok(int cin, int cout, int h, int w);        // ... the run-staged tile's form: may also write the conv output itself (a.out set)
int launch_conv_pool(const PwArgs& a, hipStream_t s);          // a.pool_out set
bool conv_head_big_supported(const PwArgs& a);
int launch_conv_head_big(const PwArgs& a, hipStream_t s);
struct DwArgs;
bool pw_dw_direct_supported(const PwArgs& a, const DwArgs& d);      // depthwise 3x3 + the 1x1 behind it in one register-direct launch (pwdirect.hip)
int launch_pw_dw_direct(const PwArgs& a, const DwArgs& d, hipStream_t s);
// SSDLite heads, depthwise 3x3 inside the 1x1 GEMM's operand staging: every level, both heads, one launch (headfuse.hip).
// Index 0 = class head, 1 = box head; both read the level's feature map x.
// ... and, since round 5, the first launch of the post-process in its epilogue (generalized_ssd.py:354,362-363: softmax over the classes,
// decode_single + clip): a workgroup holds ALL class and box channels of its 64 pixels, so every (pixel, anchor) row is complete in it.
// The logits / regressions are then not written at all; what leaves is what softmax_decode_kernel (postprocess.hip) would have produced
// from them, bit for bit: class-major scores, decoded boxes, score-histogram rows.
struct HeadPost {
    float* scoresT = nullptr;        // [n][K-1][A]; null: plain head launch (fp32 logits / regressions to HeadFuseLevel::out)
    float4* boxes = nullptr;         // [n][A]
    unsigned* hrows = nullptr;       // [n][rows_per_image][256] histogram rows: one per (32-pixel half tile, image) pair, see HistRows
    const float* anchors = nullptr;  // [A][4]
    int A = 0, K = 0, rows_per_image = 0, hb0 = 0, nb = 0;
    float img_w = 0.f, img_h = 0.f, score_thr = 0.f;
};
// Which histogram rows belong to an image (written by head_fused_kernel, added up by tau_kernel). Level l owns slots [sbase[l], sbase[l] + S_l)
// of an image's rows_per_image rows; the 32-pixel half tiles of the level that touch the image fill slots 0, 1, ... in order (an image of hw
// pixels is touched by at most (hw + 30) / 32 + 1 of them). levels == 0: the layout of softmax_decode_kernel ([n][tiles][256], every row used).
struct HistRows {
    int levels = 0, rows_per_image = 0;
    int hw[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sbase[8] = {0, 0, 0, 0, 0, 0, 0, 0}, grouped[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int extra_base = 0, extra_rows = 0;      // + rows [extra_base, extra_base + extra_rows) of every image, all of them used (the softmax_decode_kernel tiles of the small levels)
};
inline int hist_rows_slots(int hw) { return (hw + 30) / 32 + 1; }
// XCD grouping of a fused-head level (a group of xq images per workgroup residue mod 8) only where a group fills whole tiles
inline bool head_fused_grouped(int xq, int hw) { return xq > 0 && (long)xq * hw >= 256; }

struct HeadFuseLevel {
    const half_t* x;                 // [n][H][W][C]
    const half_t* wdg;               // the class head's depthwise 3x3 weights in GROUP-major order [C / 8][9 taps][8] fp16 (dn_op_desc::w2_off of its DW op)
    const unsigned char* wslot;      // [C / 32] slots of 1 KB: the box head's group-major depthwise weights of the chunk (576 B), its bias [32] fp32, the class head's bias [32] fp32 (dn_op_desc::b2_off of the class head's DW op)
    const half_t* wf[2]; const float* bias[2];    // 1x1 weights in fragment-major order (dn_op_desc::w2_off), bias [nc]
    float* out[2]; long out_img_stride[2], out_base[2];     // fp32 head arrays: elements between images, element offset of image 0's level rows
    int nc[2];                       // output channels (anchors per location x classes, x 4)
    int n, H, W, C, act;             // act: the depthwise activation
    int aoff = 0, aloc = 0, sbase = 0;      // with a HeadPost: first anchor of the level, anchors per location, first histogram-row slot
    int sm = 0;                             // with a HeadPost: 1 = this level's workgroups run the softmax / decode epilogue, 0 = they write logits
};
bool head_fused_level_supported(const HeadFuseLevel& l);
bool head_fused_post_supported(const HeadFuseLevel* lv, int count, const HeadPost& post);
int launch_head_fused(const HeadFuseLevel* lv, int count, int xq, hipStream_t s, const HeadPost* post = nullptr);

struct DwArgs {
    const half_t* x; const half_t* w; const float* bias; half_t* out;
    int n, h, w_, c, k, stride, pad, act, ho, wo;
    float* pool = nullptr;      // optional: [n][blocks][c] fp32 per-workgroup sums of the outputs (SE squeeze)
    int xq = 0;                 // XCD grouping: images per group (0: plain mapping)
    FastDiv fd_c8{1, 0}, fd_xs{1, 0};      // filled by the launcher: divisions by c / 8 and by the x strips per row
    int rb_log2 = 0; FastDiv fd_rbxs{1, 0};   // row blocks (depthwise.hip dw_body): log2 of the rows per block, division by strips * rows per block
    // squeeze-excitation FCs in the tail of the pooling launch (depthwise.hip, dw_se_tail): the workgroup of an image that finishes
    // last turns the partial sums into scale[n][c]. se_counter: one zero-initialised unsigned per image, left at zero again.
    const half_t* se_w1t = nullptr; const float* se_b1 = nullptr;      // fc1 transposed [c][sq] fp16, bias [sq]
    const half_t* se_w2t = nullptr; const float* se_b2 = nullptr;      // fc2 transposed [sq][c] fp16, bias [c]
    float* se_scale = nullptr; unsigned* se_counter = nullptr;
    int se_sq = 0; float se_inv = 0.f;                                 // squeeze width, 1 / pooled pixels
    int pool_rows = 0;                                                 // rows per image the plan sized `pool` for (0: the launcher's own choice)
};
bool depthwise_se_tail_supported(int c, int squeeze);
int launch_depthwise(const DwArgs& a, hipStream_t s);
int launch_depthwise_group(const DwArgs* arr, int count, hipStream_t s);
int depthwise_pool_blocks(const DwArgs& a);       // workgroups per image == partial-sum rows per image

struct StemArgs {
    const float* img;       // [n][3][h][w] fp32 (raw, 0..1)
    const float* w;         // [k*k*3][cout] fp32, tap index = (c*k + ky)*k + kx
    const float* bias;      // [cout]
    half_t* out;            // [n][ho][wo][cout]
    int n, h, w_, cout, k, stride, pad, act, ho, wo;
    float mean[3], inv_std[3];
    int xq = 0;             // XCD grouping: images per group (0: plain mapping)
    unsigned* zero_u32 = nullptr; int zero_count = 0;      // optional: words the first workgroup clears (the chain's SE counters)
    int split_ok = 0;       // weights, bias and the normalised input range are finite and fit the split-fp16 matrix kernel (plan.hip checks the host copy)
    float w_scale = 1.f, w_unscale = 1.f;      // 2^s and 2^-s: the split kernel's weights / bias are scaled so that the largest magnitude is in [2^13, 2^15)
};
int launch_stem(const StemArgs& a, hipStream_t s);

int launch_se_fc(const float* partial, int nblk, const void* w1t, const float* b1, const void* w2t, const float* b2, float* scale,
                 int n, int c, int squeeze, int pool_pixels, hipStream_t s, int xq = 0);

struct ConvArgs {
    const half_t* x; const half_t* w; const float* bias; void* out;
    const half_t* zeros = nullptr;      // >= 16 zero bytes on the device (optional; enables convbig.hip)
    int n, h, w_, cin, cout, k, stride, pad, dil, act, ho, wo, out_fp32;
    long out_img_stride, out_base;
    int xq = 0;                         // XCD grouping: images per group (0: plain mapping)
};
int launch_conv(const ConvArgs& a, hipStream_t s);
PwArgs conv_to_pw(const ConvArgs& c);
int launch_poison(hipStream_t s);     // DN_POISON: NaN patterns into every LDS byte and vector register (dense.hip)
int launch_maxpool(const half_t* x, half_t* out, int n, int h, int w, int c, int k, int stride, int pad, int ho, int wo,
                   hipStream_t s);
int launch_l2norm(const half_t* x, const float* scale, half_t* out, long pixels, int c, hipStream_t s);
// bilinear (align_corners=False) resize of NCHW fp32 planes; also writes scale_xy[n][2] = (w/ow, h/oh) in fp32
int launch_u8hwc_to_planar(const unsigned char* in, float* out, float* scale_xy, int n, int h, int w, int oh, int ow, hipStream_t s);
int launch_resize_bilinear(const float* in, float* out, float* scale_xy, int n, int h, int w, int oh, int ow, hipStream_t s);

// fused inverted-residual block (fused.hip): [expand 1x1] -> depthwise -> [project 1x1 (+residual)]
// run of tiny layers (<= 32 output pixels each) executed by one workgroup per image with the activations in LDS (tail.hip)
constexpr int TAIL_MAX_OPS = 16;
struct TailOp {
    int type, cin, cout, k, stride, pad, hin, win, hout, wout, act;
    long w_off;             // halfs from TailArgs::weights
    long b_off;             // bytes from TailArgs::weights
    half_t* out;            // non-null: the output is also written to HBM (per-image stride out_stride halfs)
    long out_stride;
};
struct TailArgs {
    int count, buf_halfs, buf2_halfs;
    const half_t* in0; long in0_stride;     // first op's input [n][pixels][cin], per-image stride in halfs
    const half_t* weights;
    long long* stamps;                      // dev-only
    int xq;                                 // XCD grouping: images per group (0: plain mapping)
    TailOp op[TAIL_MAX_OPS];
};
int launch_tail(const TailArgs& a, int n, hipStream_t s);
bool tail_op_supported(const dn_op_desc& o, int hin, int win, int hout, int wout);

// expand 1x1 + depthwise kxk in one launch (expdw.hip)
struct ExpDwArgs {
    const half_t* x; half_t* out; float* pool;     // pool (optional): [n][tiles][cexp] fp32 per-tile channel sums
    const half_t* w1; const float* b1;             // expand [cexp][cin]; null: no expand stage (cexp == cin, the depthwise reads x)
    const half_t* wd; const float* bd;             // depthwise [k*k][cexp]
    const half_t* w3; const float* b3;             // project [cout][cexp]; null: stop after the depthwise stage (out has cexp channels)
    int n, H, W, Ho, Wo, cin, cexp, cout, k, stride, pad, act1, act2, has_res;   // has_res: out += x (stride 1, cout == cin)
    int xw, chunks_per_wg;                         // filled by the launcher
    int xq;                                        // XCD grouping: images per group (0: plain mapping)
    long long* stamps;                             // dev-only
    int stage_out = 0; FastDiv fd_oc8 = {1, 0};    // filled by the launcher: the projected tile leaves through LDS as row-contiguous 16-byte chunks
};
int launch_expdw(const ExpDwArgs& a, hipStream_t s);
int expdw_tiles_per_image(int Ho, int Wo, int stride);
bool expdw_supported(int cin, int cexp, int k, int stride);
bool expdw_project_supported(int cexp, int cout, int Ho, int Wo, int stride);


// Pyramid levels of the anchor axis. The class-major score array of the post-process is stored ANCHOR-MAJOR WITHIN A LEVEL:
//   canonical (the reference's order, generalized_ssd.py:66-74):  a  = off[l] + pixel * aloc[l] + anchor
//   stored:                                                       a' = off[l] + anchor * hw[l] + pixel
// so that a head GEMM tile (consecutive pixels of one anchor) writes runs of consecutive floats per class (round 3: softmax in the head
// launch's epilogue). Every consumer turns a' back into a before an index enters a key, a box lookup or an output.
struct PostLevels {
    int n = 1;
    int off[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};     // off[n] = A
    int hw[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    int aloc[8] = {1, 1, 1, 1, 1, 1, 1, 1};
};
__device__ __forceinline__ int post_level_of(const PostLevels& lv, int a) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) l += (i < lv.n && a >= lv.off[i]) ? 1 : 0;
    return l;
}
__device__ __forceinline__ bool post_identity(const PostLevels& lv) { return lv.n == 1 && lv.aloc[0] == 1; }      // uniform: stored == canonical
__device__ __forceinline__ int post_canon(const PostLevels& lv, int ap) {      // stored -> canonical
    if (post_identity(lv)) return ap;
    const int l = post_level_of(lv, ap), r = ap - lv.off[l];
    const int anc = r / lv.hw[l], pix = r - anc * lv.hw[l];
    return lv.off[l] + pix * lv.aloc[l] + anc;
}
__device__ __forceinline__ int post_perm(const PostLevels& lv, int a) {        // canonical -> stored
    if (post_identity(lv)) return a;
    const int l = post_level_of(lv, a), r = a - lv.off[l];
    const int pix = r / lv.aloc[l], anc = r - pix * lv.aloc[l];
    return lv.off[l] + anc * lv.hw[l] + pix;
}

struct PostArgs {
    const float* logits; const float* reg; const float* anchors;
    int n, A, K;
    float img_h, img_w; const float* scale_xy;
    float score_thresh, nms_thresh; int topk, dets;
    float* boxes; float* scores; int64_t* labels; int32_t* counts; int32_t* kept_anchor;
    float* packed = nullptr;    // optional [n][dets+1][6] fp32: rows (x1,y1,x2,y2,score,label), row `dets` = (count,0,..)
    void* ws; size_t ws_bytes;
    int xq = 0;                 // XCD grouping: images per group (0: plain mapping)
    PostLevels lv;              // default: one level with one anchor per location (stored order == canonical order)
    // scores, boxes and histogram rows already written by the fused head launch (HeadPost): the softmax / decode launch is skipped and
    // tau_kernel reads the rows through this table
    bool scores_ready = false;
    HistRows hrows;
    int small_first = -1;       // with scores_ready: first anchor still in logit form (the levels below 32 pixels per image); -1 / A: none
    unsigned* tickets = nullptr; // with small levels: one zeroed counter per image (left at zero) -> the cut-off runs in the last softmax tile of the image, no tau launch
};
// where launch_postprocess keeps its arrays inside the workspace it is given (the fused head launch writes the first three itself)
struct PostBuffers {
    float* scoresT; float4* boxes; float* keptScore; int* keptAnchor; int* keptCount;
    unsigned* phist; unsigned* tauKey; int* needFull; int* order; int* fbcnt;
    int tiles;
};
PostBuffers post_buffers(void* ws, int n, int A, int K, int topk);
size_t postprocess_ws_bytes(int n, int A, int K, int topk, int dets);
void post_hist_range(float score_thresh, int* hb0, int* nb, int* clamped);
int launch_postprocess(const PostArgs& a, hipStream_t s, hipEvent_t* ev /* optional [5] phase boundaries: softmax+decode | cut-off + selection | merge | fallback */);
