// pointwise_fwd.hip -- point-wise (kernel-size-1) convolution with the PREVIOUS layer's BatchNorm +
// ReLU applied to its input on the fly, on the exact-fp32 MFMA, gfx950.
//
//   y[b, o, p] = sum_i W[o, i] * act_i(x[b, i, p])        x (B, Cin, P), y (B, Cout, P)
//   act_i(v)   = relu?(v * sc_i + sh_i),  sc_i = invstd_i * gamma_i,  sh_i = beta_i - mean_i * sc_i
//                (the arithmetic of bn_apply_kernel, csrc/bn_act.hip), or the identity
//
// This is one [BN -> ReLU -> Conv2d 1x1] step of the reference's shared MLPs (pointnet2_batch/
// pointnet2_modules.py:86-92, pointnet2_stack/pointnet2_modules.py:33-40, voxel_pool_modules.py:41-
// 52) without ever writing the activated tensor: the layer input is the previous layer's PRE-BN
// output.  The same kernel with W read transposed and the identity activation is the data gradient
// of the layer (dX = W^T dY).
//
// As a GEMM it is skinny -- Cout, Cin <= 64, up to 15.7 M columns at config c3 -- and HBM-bound
// (4*(Cin + Cout) bytes per column against 2*Cin*Cout flops: 5..16 flop/byte); the library runs these
// shapes at 1.2-1.6 TB/s.  Here:
//   * the B operand of v_mfma_f32_32x32x2_f32 (k = channel pair, n = column) is exactly the layout
//     of a row-major (channel, column) tensor, so x goes global -> registers -> MFMA with no LDS
//     staging: lane (l, h) loads a float4 = columns 4l..4l+3 of channel 2k+h (512 contiguous bytes
//     per half-wave), and component j of the float4 feeds the j-th of four interleaved 32-column
//     sub-tiles; the D registers of the four sub-tiles then hold columns 4l..4l+3 again, so y is
//     written with 16-byte stores;
//   * W (<= 32 KB) and the per-channel (sc, sh) sit in LDS, read conflict-free as the A operand;
//   * every wave owns whole 128-column tiles and software-pipelines them: the loads of the next
//     16-channel slab are in flight while the MFMAs of the current one run.
#include <type_traits>

#include "common.hpp"
#include "payload.hpp"

namespace mgar {

typedef float __attribute__((ext_vector_type(16))) f32x16;

constexpr int PF_COLS = 128;  // columns per wave tile
// channels per slab (2 per MFMA k-step): 16, or 8 for the 64-output-channel instance so that its 128 accumulator
// registers + two slab buffers stay under 256 VGPRs (2 waves per SIMD instead of 1)
template <int OB> struct PfSlab { static constexpr int KC = OB == 1 ? 16 : 8; };

template <typename T>
struct PfArgs {
    const T *x;                                 // payload: float or bf16_t (fp32 accumulation on the fp32 MFMA either way)
    const float *w;
    const float *mean, *invstd, *gamma, *beta;  // input activation (mean == nullptr: identity)
    T *y;
    int B, Cin, Cout, P;
    int w_rs, w_cs;  // W[o, i] = w[o * w_rs + i * w_cs]
    int relu;
    // optional (P % 128 == 0): per (output channel, 128-column tile) the tile's (mean, M2) of y, in the layout of
    // bn_partial_kernel with chunk = 128 -- stats[(o * ntiles + tile) * 2 + {0, 1}], tile = b * (P / 128) + tile of the
    // sample -- so that the BatchNorm that follows needs no pass over y for its statistics (csrc/bn_act.hip, bn_finalize_kernel)
    float *stats;
};

template <int OB, typename T, bool STATS = false>
__global__ __launch_bounds__(256, 2) void pointwise_fwd_kernel(PfArgs<T> a) {
    constexpr int PF_KC = PfSlab<OB>::KC, KS = PF_KC / 2;
    constexpr int WLD = OB == 1 ? 32 : 96;  // LDS row stride of W: the two half-waves hit disjoint banks
    extern __shared__ float lds[];          // [nkc*16][WLD] weights, then [nkc*16][4] (sc, sh, mu, -)
    const int nkc = (a.Cin + PF_KC - 1) / PF_KC;
    float *wl = lds;
    float *act = lds + (size_t)nkc * PF_KC * WLD;
    for (int e = threadIdx.x; e < nkc * PF_KC * WLD; e += 256) {
        const int ch = e / WLD, o = e - ch * WLD;
        wl[e] = (ch < a.Cin && o < a.Cout) ? a.w[(size_t)o * a.w_rs + (size_t)ch * a.w_cs] : 0.f;
    }
    for (int ch = threadIdx.x; ch < nkc * PF_KC; ch += 256) {
        float sc = 1.f, sh = 0.f, mu = 0.f;   // act(v) = (v - mu) * sc + sh: the arithmetic of bn_apply_kernel
        if (a.mean && ch < a.Cin) {
            sc = a.invstd[ch] * (a.gamma ? a.gamma[ch] : 1.f);
            sh = a.beta ? a.beta[ch] : 0.f;
            mu = a.mean[ch];
        }
        *reinterpret_cast<float4 *>(act + 4 * ch) = make_float4(sc, sh, mu, 0.f);
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l = lane & 31, h = lane >> 5;
    const int tiles_per_b = (a.P + PF_COLS - 1) / PF_COLS;
    const long long ntiles = (long long)a.B * tiles_per_b;
    const long long wave_id = (long long)blockIdx.x * 4 + wave, nwaves = (long long)gridDim.x * 4;
    if (wave_id >= ntiles) return;
    const long long my_tiles = (ntiles - wave_id + nwaves - 1) / nwaves;
    const long long steps = my_tiles * nkc;
    const bool has_act = a.mean != nullptr, relu = a.relu != 0;

    f32x16 acc[OB][4];
#pragma unroll
    for (int ob = 0; ob < OB; ++ob)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ob][j][r] = 0.f;

    // slab loads of step s (tile s / nkc of this wave, channel slab s % nkc)
    auto issue = [&](long long s, float4 (&buf)[KS]) {
        const long long t = wave_id + (s / nkc) * nwaves;
        const int kc = (int)(s % nkc);
        const int b = (int)(t / tiles_per_b);
        const int p = (int)(t - (long long)b * tiles_per_b) * PF_COLS + 4 * l;
        const T *src = a.x + ((size_t)b * a.Cin + kc * PF_KC + h) * a.P + p;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int ch = kc * PF_KC + 2 * ks + h;
            buf[ks] = (ch < a.Cin && p < a.P) ? Payload<T>::ld4(src + (size_t)2 * ks * a.P)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto compute = [&](long long s, float4 (&buf)[KS]) {
        const int kc = (int)(s % nkc);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int ch = kc * PF_KC + 2 * ks + h;
            float4 v = buf[ks];
            if (has_act) {
                const float4 ac = *reinterpret_cast<const float4 *>(act + 4 * ch);
                const float sc = ac.x, sh = ac.y, mu = ac.z;
                v.x = (v.x - mu) * sc + sh; v.y = (v.y - mu) * sc + sh; v.z = (v.z - mu) * sc + sh; v.w = (v.w - mu) * sc + sh;
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (ch >= a.Cin) v = make_float4(0.f, 0.f, 0.f, 0.f);   // padding channels stay zero (W is zero there too)
            }
#pragma unroll
            for (int ob = 0; ob < OB; ++ob) {
                const float wa = wl[ch * WLD + ob * 32 + l];
                acc[ob][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, v.x, acc[ob][0], 0, 0, 0);
                acc[ob][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, v.y, acc[ob][1], 0, 0, 0);
                acc[ob][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, v.z, acc[ob][2], 0, 0, 0);
                acc[ob][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, v.w, acc[ob][3], 0, 0, 0);
            }
        }
        if (kc == nkc - 1) {  // tile finished: D[row = (r&3) + 8*(r>>2) + 4*h][col = l] of sub-tile j = column 4l + j
            const long long t = wave_id + (s / nkc) * nwaves;
            const int b = (int)(t / tiles_per_b);
            const int p = (int)(t - (long long)b * tiles_per_b) * PF_COLS + 4 * l;
            if (STATS) {   // the 128 columns of a channel live in the 32 lanes of one half-wave x 4 sub-tiles: exact two-pass
#pragma unroll
                for (int ob = 0; ob < OB; ++ob)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v0 = acc[ob][0][r], v1 = acc[ob][1][r], v2 = acc[ob][2][r], v3 = acc[ob][3][r];
                        float sum = (v0 + v1) + (v2 + v3);
#pragma unroll
                        for (int d = 1; d < 32; d <<= 1) sum += __shfl_xor(sum, d, 32);
                        const float mean = sum * (1.f / PF_COLS);
                        const float d0 = v0 - mean, d1 = v1 - mean, d2 = v2 - mean, d3 = v3 - mean;
                        float m2 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
#pragma unroll
                        for (int d = 1; d < 32; d <<= 1) m2 += __shfl_xor(m2, d, 32);
                        const int o = ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (l == 0 && o < a.Cout)
                            *reinterpret_cast<float2 *>(a.stats + ((size_t)o * ntiles + t) * 2) = make_float2(mean, m2);
                    }
            }
#pragma unroll
            for (int ob = 0; ob < OB; ++ob)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (o < a.Cout && p < a.P)
                        Payload<T>::st4(a.y + ((size_t)b * a.Cout + o) * a.P + p,
                                        make_float4(acc[ob][0][r], acc[ob][1][r], acc[ob][2][r], acc[ob][3][r]));
                    acc[ob][0][r] = 0.f; acc[ob][1][r] = 0.f; acc[ob][2][r] = 0.f; acc[ob][3][r] = 0.f;
                }
        }
    };

    float4 buf0[KS], buf1[KS];
    issue(0, buf0);
    for (long long s = 0; s < steps; s += 2) {
        if (s + 1 < steps) issue(s + 1, buf1);
        compute(s, buf0);
        if (s + 1 < steps) {
            if (s + 2 < steps) issue(s + 2, buf0);
            compute(s + 1, buf1);
        }
    }
}

template <int OB, typename T, bool STATS = false>
static void launch_pf(const PfArgs<T> &a, hipStream_t st) {
    constexpr int WLD = OB == 1 ? 32 : 96, PF_KC = PfSlab<OB>::KC;
    const int nkc = (a.Cin + PF_KC - 1) / PF_KC;
    const int lds = nkc * PF_KC * (WLD + 4) * (int)sizeof(float);
    static int attr_lds = 0;
    if (lds > 65536 && lds > attr_lds) {
        (void)hipFuncSetAttribute((const void *)pointwise_fwd_kernel<OB, T, STATS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_lds = lds;
    }
    const long long ntiles = (long long)a.B * ((a.P + PF_COLS - 1) / PF_COLS);
    long long wgs = (ntiles + 3) / 4;
    if (wgs > 2048) wgs = 2048;   // 8 workgroups per CU; waves stride over the tiles
    hipLaunchKernelGGL((pointwise_fwd_kernel<OB, T, STATS>), dim3((unsigned)wgs), dim3(256), lds, st, a);
}

}  // namespace mgar

using namespace mgar;

template <typename T>
static int pointwise_conv_fwd_impl(const T *x, int B, int Cin, int P, const float *w, int w_row_stride, int w_col_stride, int Cout,
                                   const float *in_mean, const float *in_invstd, const float *in_gamma, const float *in_beta,
                                   int in_relu, T *y, void *stream, float *out_stats = nullptr) {
    MGAR_REQUIRE(B >= 0 && Cin >= 0 && Cout >= 0 && P >= 0, "pointwise_conv_fwd: negative size");
    if ((long long)B * P == 0 || Cout == 0) return MGAR_OK;
    MGAR_REQUIRE(x && w && y, "pointwise_conv_fwd: null pointer");
    MGAR_REQUIRE(in_mean == nullptr || in_invstd != nullptr, "pointwise_conv_fwd: in_mean without in_invstd");
    if (Cout > 64 || Cin > 256 || Cin == 0 || (P & 3) != 0) {
        set_error("pointwise_conv_fwd: needs 1 <= Cin <= 256, Cout <= 64 and P % 4 == 0 (use the library GEMM otherwise)");
        return MGAR_EUNSUPPORTED;
    }
    MGAR_REQUIRE(out_stats == nullptr || P % PF_COLS == 0, "pointwise_conv_fwd: output statistics need P % 128 == 0");
    PfArgs<T> a{x, w, in_mean, in_invstd, in_gamma, in_beta, y, B, Cin, Cout, P, w_row_stride, w_col_stride, in_relu, out_stats};
    hipStream_t st = (hipStream_t)stream;
    KtScope kt(KT_POINTWISE_FWD, st, (double)sizeof(T) * B * P * (Cin + Cout), 2.0 * (double)B * P * Cin * Cout);
    if (out_stats) {   // Cout <= 32 (checked by the entry point): the 64-channel instance has no registers left for it
        if constexpr (std::is_same<T, float>::value) launch_pf<1, float, true>(a, st);
    } else if (Cout <= 32) launch_pf<1, T>(a, st);
    else launch_pf<2, T>(a, st);
    return check_launch("pointwise_conv_fwd: launch failed");
}

#define PF_API extern "C" __attribute__((visibility("default")))
PF_API int mgar_pointwise_conv_fwd(const float *x, int B, int Cin, int P, const float *w, int w_row_stride, int w_col_stride, int Cout,
                                   const float *in_mean, const float *in_invstd, const float *in_gamma, const float *in_beta,
                                   int in_relu, float *y, void *stream) {
    return pointwise_conv_fwd_impl<float>(x, B, Cin, P, w, w_row_stride, w_col_stride, Cout, in_mean, in_invstd, in_gamma, in_beta,
                                          in_relu, y, stream);
}
// bf16 payload (x, y address bf16 elements; W and the BatchNorm vectors stay fp32)
PF_API int mgar_pointwise_conv_fwd_bf16(const void *x, int B, int Cin, int P, const float *w, int w_row_stride, int w_col_stride,
                                        int Cout, const float *in_mean, const float *in_invstd, const float *in_gamma,
                                        const float *in_beta, int in_relu, void *y, void *stream) {
    return pointwise_conv_fwd_impl<bf16_t>((const bf16_t *)x, B, Cin, P, w, w_row_stride, w_col_stride, Cout, in_mean, in_invstd,
                                           in_gamma, in_beta, in_relu, (bf16_t *)y, stream);
}

// pointwise_conv_fwd that also leaves the BatchNorm statistics partials of its OUTPUT: out_stats (Cout, B * P / 128, 2) floats =
// per (channel, 128-column tile) the tile's mean and sum of squared deviations, the chunk format mgar_bn_stats_from_partials
// finalizes (chunk = 128).  P % 128 == 0, Cout <= 32.  fp32.  Saves the BatchNorm's own pass over y (4 * B * Cout * P bytes).
PF_API int mgar_pointwise_conv_fwd_stats(const float *x, int B, int Cin, int P, const float *w, int w_row_stride, int w_col_stride,
                                         int Cout, const float *in_mean, const float *in_invstd, const float *in_gamma,
                                         const float *in_beta, int in_relu, float *y, float *out_stats, void *stream) {
    MGAR_REQUIRE(out_stats, "pointwise_conv_fwd_stats: null pointer");
    if (Cout > 32) {
        set_error("pointwise_conv_fwd_stats: Cout <= 32 (run mgar_pointwise_conv_fwd and the BatchNorm's own statistics pass)");
        return MGAR_EUNSUPPORTED;
    }
    return pointwise_conv_fwd_impl<float>(x, B, Cin, P, w, w_row_stride, w_col_stride, Cout, in_mean, in_invstd, in_gamma, in_beta,
                                          in_relu, y, stream, out_stats);
}
