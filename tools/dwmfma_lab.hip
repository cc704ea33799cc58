// LAB ARTIFACT (round 6, not built into the library): the formulation is verified -- wired in behind a knob it passed eight op-level cases against
// the fp32 reference and against dw_kernel (40 x 40 x 120 5 x 5, 20 x 20 x 480 / 672 3 x 3, 10 x 10 x 480 5 x 5, ragged 21 x 19 and 33 x 47, 5 x 5 maps, 9
// images) on the first run -- and this decomposition is 2.5 - 4 x SLOWER than dw_kernel (64 images: 40 x 40 x 120 k5: 61 us against 21.4; 20 x 20 x 480 k3:
// 47 against 12.0; 10 x 10 x 960 k5: 28 against 15). Ablations (40 x 40 x 120): without the staging loads 36, without the matrix phase 50, without
// the stores 44, without loads + matrix phase 25, with nothing 8.7. What costs is the MEMORY side of "one workgroup = one image x 8 channels": every
// lane's 16 bytes sit 2 C bytes apart, so a load or store instruction touches 64 lines for 1 KB (dw_kernel's lanes are consecutive channel groups:
// whole lines), and the C / 8 workgroups of an image fetch every line C / 8 times. The next form needs >= 32 channels per workgroup (64-byte runs) and
// therefore spatial tiles (32 planes of a 20 x 40 halo tile = 51 KB), pooled sums and the stride-2 band. To try it again: add the file to
// demonet_amd/build.py (VGPR-form MFMA flags), declare depthwise_mfma_supported / launch_depthwise_mfma in common.h and call them from launch_depthwise.
//
// Depthwise k x k, stride 1, on the fp16 matrix cores: each kernel row as a banded (Toeplitz) matrix along W.
//
// demonet/models/mobilenetv3.py:81 (InvertedResidual's depthwise ConvBNActivation), ssd_mobilenetv3.py:31,48.
//
// Why: the depthwise launches are bound by vector issue -- v_fma_mix_f32 goes at 4.3 - 5 cycles per wave, 25 of them per output and 8 channels for
// 5 x 5 -- and a SIMD runs fp16 matrix instructions in the shadow of vector work (tools/mfma_valu_lab.hip). A depthwise conv has no reduction over
// channels, but per channel c and kernel row ky
//     out[oy][ox] += sum_ix  T[ox][ix] * xpad[oy + ky][ix],      T[ox][ix] = w[ky][ix - ox][c]  for 0 <= ix - ox < k, else 0
// is a matrix product: M = 16 output columns, K = 32 padded input columns (16 + k - 1 used), N = 16 output rows -- one v_mfma_f32_16x16x32_f16 per
// (channel, ky, 16 x 16 block) with exact fp16 products and fp32 accumulation, 5 of them where the vector form spends 16 x 16 x 25 / 64 = 100
// multiply-adds. The price is the layout: the B operand wants 8 consecutive columns of ONE channel per lane, the tensors are NHWC.
//   * a workgroup = one image x 8 channels (the 16 bytes a lane loads per pixel); the image goes to LDS channel-major, zero-padded, 2-byte writes
//     (lanes = consecutive pixels: conflict-free), rows of WP halves with WP * 2 = 16 (mod 128) so that the sixteen 16-byte row reads of a B fragment
//     fall into different banks;
//   * A fragments (the band) are gathered per (channel, ky) from the workgroup's 25 x 8 weights in LDS: lane (m, kg) reads w[ky][8 kg + j - m][c],
//     out-of-band positions point into a zero block -- eight 2-byte LDS reads with immediate offsets, no vector work;
//   * a wave owns up to MAXB of the image's 16 x 16 blocks and keeps their accumulators for all 8 channels (4 registers per channel and block), so
//     that the epilogue has the 8 channels of a pixel in one lane: bias, activation, fp16, one 16-byte NHWC store per pixel.
// Summation order differs from dw_kernel's (tap order there, the matrix instruction's K order here): results agree to fp32 rounding, not bit for bit.
#include "common.h"

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int K, int MAXB>
__global__ __launch_bounds__(256) void dwm_kernel(DwArgs a, int nrb, int nxb, int HP, int WP, int abl) {       // abl: the ablation bits of the numbers above (DN_DWM_ABL)
    extern __shared__ __attribute__((aligned(16))) half_t lds[];
    constexpr int P = (K - 1) / 2;
    constexpr int WT = K * K * 8;                   // weights [ky][kx][8 channels], then as many zeros
    half_t* wsh = lds;
    half_t* planes = lds + 2 * WT;                  // [8][HP][WP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int slab = blockIdx.x, img = blockIdx.y;
    const int H = a.h, W = a.w_, C = a.c;
    const int plane_halfs = HP * WP;

    // ---- zero the planes (padding, and the reach of the last blocks' windows), stage the weights
    {
        uint4* z = reinterpret_cast<uint4*>(lds);
        const int n16 = (2 * WT + 8 * plane_halfs) / 8;
        for (int i = tid; i < ((abl & 8) ? 1 : n16); i += blockDim.x) z[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    if (tid < K * K) *reinterpret_cast<uint4*>(wsh + tid * 8) = *reinterpret_cast<const uint4*>(a.w + (size_t)tid * C + slab * 8);
    // ---- the image, channel-major: thread = pixel, its 8 channels to 8 planes
    {
        const half_t* xb = a.x + (size_t)img * H * W * C + slab * 8;
        for (int p0 = tid; p0 < ((abl & 1) ? 0 : H * W); p0 += 4 * blockDim.x) {
            half8 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = min(p0 + u * (int)blockDim.x, H * W - 1);
                v[u] = *reinterpret_cast<const half8*>(xb + (size_t)p * C);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = p0 + u * (int)blockDim.x;
                if (p < H * W) {
                    const int y = p / W, x = p - y * W;
                    half_t* d = planes + (y + P) * WP + x + P;
#pragma unroll
                    for (int c = 0; c < 8; ++c) d[c * plane_halfs] = v[u][c];
                }
            }
        }
    }
    __syncthreads();

    // ---- this lane's band: A element j of a fragment is w[ky][8 kg + j - m] (byte offset into wsh; out of band: the zero block behind the weights)
    const int m = lane & 15, kg = lane >> 4;
    int aoff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int d = 8 * kg + j - m;
        aoff[j] = (d >= 0 && d < K) ? d * 16 : WT * 2;
    }
    const char* wbytes = reinterpret_cast<const char*>(wsh);

    floatx4 acc[MAXB][8];
#pragma unroll
    for (int b = 0; b < MAXB; ++b)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[b][c] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int nblk = nrb * nxb;
    // B fragment of block b, row shift ky: lane (n = lane & 15, kg) reads plane[c][oy0 + n + ky][ox0 + 8 kg .. + 7] (padded coordinates; rows beyond
    // the padded image -- outputs that do not exist -- re-read its last row). Byte offsets inside a plane, once per lane.
    int boff[MAXB][K];
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
        const int blk = min(wave + b * nw, nblk - 1);
        const int rb = blk / nxb, xb = blk - rb * nxb;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) boff[b][ky] = (min(rb * 16 + m + ky, HP - 1) * WP + xb * 16 + 8 * kg) * 2;
    }
    const char* pbytes = reinterpret_cast<const char*>(planes);
#pragma unroll
    for (int c = 0; c < ((abl & 2) ? 0 : 8); ++c) {
        half8 A[K];
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int j = 0; j < 8; ++j) A[ky][j] = *reinterpret_cast<const half_t*>(wbytes + aoff[j] + ky * K * 16 + c * 2);
        // all row reads of the channel in one batch (one exposed LDS round trip per channel instead of one per matrix instruction)
        half8 B[MAXB][K];
        const char* pc = pbytes + c * plane_halfs * 2;
#pragma unroll
        for (int b = 0; b < MAXB; ++b)
#pragma unroll
            for (int ky = 0; ky < K; ++ky) B[b][ky] = *reinterpret_cast<const half8*>(pc + boff[b][ky]);
#pragma unroll
        for (int b = 0; b < MAXB; ++b)
#pragma unroll
            for (int ky = 0; ky < K; ++ky) acc[b][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ky], B[b][ky], acc[b][c], 0, 0, 0);
    }

    // ---- epilogue: lane (n = lane & 15 -> row, rg = lane >> 4) holds columns ox0 + 4 rg + e of its row, 8 channels each
    float bias[8];
    {
        const float4 b0 = *reinterpret_cast<const float4*>(a.bias + slab * 8), b1 = *reinterpret_cast<const float4*>(a.bias + slab * 8 + 4);
        bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w; bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
    }
    half_t* ob = a.out + (size_t)img * H * W * C + slab * 8;
    auto emit = [&](auto actf) {
#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
            const int blk = wave + b * nw;
            const int rb = blk / nxb, xb = blk - rb * nxb;
            const int oy = rb * 16 + m;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ox = xb * 16 + 4 * kg + e;
                half8 o;
#pragma unroll
                for (int c = 0; c < 8; ++c) o[c] = (half_t)actf(acc[b][c][e] + bias[c]);
                if (blk < nblk && oy < H && ox < W && !(abl & 4)) *reinterpret_cast<half8*>(ob + ((size_t)oy * W + ox) * C) = o;
            }
        }
    };
    if (a.act == DN_ACT_RELU) emit([](float v) { return dn_relu(v); });
    else if (a.act == DN_ACT_RELU6) emit([](float v) { return dn_relu6(v); });
    else if (a.act == DN_ACT_HSWISH) emit([](float v) { return v * dn_relu6(v + 3.f) * (1.f / 6.f); });
    else emit([](float v) { return v; });
}

}  // namespace

bool depthwise_mfma_supported(const DwArgs& a) {
    if (!(a.stride == 1 && (a.k == 3 || a.k == 5) && a.pad == (a.k - 1) / 2 && a.c % 8 == 0 && a.ho == a.h && a.wo == a.w_ && !a.pool)) return false;
    const int nrb = dn_cdiv(a.h, 16), nxb = dn_cdiv(a.w_, 16);
    return nrb * nxb <= 12 && a.n <= 65535;
}

int launch_depthwise_mfma(const DwArgs& a, hipStream_t s) {
    const int nrb = dn_cdiv(a.h, 16), nxb = dn_cdiv(a.w_, 16), nblk = nrb * nxb;
    const int HP = a.h + a.k - 1;
    const int WP = 16 * nxb + 16 + 8;               // the last block's window reaches 32 columns from its first; + 8: rows 16 bytes apart mod 128
    const int waves = std::min(4, nblk);
    const size_t lds = ((size_t)2 * a.k * a.k * 8 + (size_t)8 * HP * WP) * 2;
    dn_note_kernel("dwm_kernel<%d>", a.k);
    const dim3 grid(a.c / 8, a.n);
#define DN_DWM(K_, MB_)                                                                                              \
    do {                                                                                                              \
        DN_HIP_CHECK(dn_allow_big_lds(reinterpret_cast<const void*>(dwm_kernel<K_, MB_>), 160 * 1024));               \
        hipLaunchKernelGGL((dwm_kernel<K_, MB_>), grid, dim3(64 * waves), lds, s, a, nrb, nxb, HP, WP, dn_knob("DN_DWM_ABL", 0));               \
    } while (0)
    const int maxb = dn_cdiv(nblk, waves);
    if (a.k == 5) { if (maxb <= 1) DN_DWM(5, 1); else if (maxb == 2) DN_DWM(5, 2); else DN_DWM(5, 3); }
    else { if (maxb <= 1) DN_DWM(3, 1); else if (maxb == 2) DN_DWM(3, 2); else DN_DWM(3, 3); }
#undef DN_DWM
    return DN_OK;
}
