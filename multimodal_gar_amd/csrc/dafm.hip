// dafm.hip -- Distance-Aware Fusion Module attention core (fwd + bwd) for gfx950.
//
// Replaces the op chain of the reference's FusionAttention_mat.forward
// (model/gat_model.py:487-491 for the R branch, :503-505 for the L branch):
//     E   = softmax(-(De / sigma), dim=1)
//     Att = softmax(Q K^T * E / sqrt(out_dim), dim=1)
//     out = Att V
// batched over S scenes (the reference loops over scenes in Python, gat_model.py:1396, and
// launches ~10 tiny kernels per scene per branch).  The Q/K/V projections and the FFN are
// plain dense GEMMs and stay on the library GEMM path; this kernel fuses everything between
// them, so the (n, n) logits never exist in HBM except for the saved Att.
//
// Shape: n <= 128 actors, D = 512.  One WAVE per query row i:
//   lanes along j   : logits  s_ij = q_i . k_j  (q_i arrives as wave-uniform scalars, each lane
//                     streams its own k_j row), E and the two row-softmaxes as DPP reductions;
//   lanes along d   : out_i = sum_j att_ij v_j  (att_ij broadcast with v_readlane, v_j rows read
//                     coalesced).
// Weight-read bound at these sizes; the point of the kernel is launch/HBM
// round-trip removal, not FLOPs.
#include "common.hpp"

namespace mgar {

constexpr int DAFM_JPL = MGAR_DAFM_MAX_N / kWave;  // j's per lane (2)
typedef const float __attribute__((address_space(4))) *cfloat_p;

__device__ __forceinline__ int scene_of_row(int row, int S, const int *__restrict__ scene_off) {
    int lo = 0, hi = S - 1;  // largest s with scene_off[s] <= row
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (scene_off[mid] <= row) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// grid: ceil(total_rows / 4) workgroups of 4 waves
__global__ __launch_bounds__(256) void dafm_fwd_kernel(int S, int total_rows, int D, const int *__restrict__ scene_off,
                                                       const int *__restrict__ de_off, const float *__restrict__ q,
                                                       const float *__restrict__ k, const float *__restrict__ v,
                                                       const float *__restrict__ de, float inv_sigma, float scale,
                                                       float *__restrict__ att, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= total_rows) return;
    const int s = scene_of_row(row, S, scene_off);
    const int r0 = scene_off[s], n = scene_off[s + 1] - r0, i = row - r0;
    const float *de_row = de + de_off[s] + (size_t)i * n;
    float *att_row = att + de_off[s] + (size_t)i * n;

    // ---- E row and raw logits, lanes along j ----
    float x[DAFM_JPL], dot[DAFM_JPL];
    float xmax = -__builtin_inff();
#pragma unroll
    for (int t = 0; t < DAFM_JPL; ++t) {
        const int j = lane + t * kWave;
        x[t] = j < n ? -(de_row[j] * inv_sigma) : -__builtin_inff();
        xmax = fmaxf(xmax, x[t]);
        dot[t] = 0.f;
    }
    xmax = wave_max(xmax);
    cfloat_p qi = (cfloat_p)(q + (size_t)row * D);
    for (int d0 = 0; d0 < D; d0 += 16) {
        float qc[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) qc[u] = qi[d0 + u];
#pragma unroll
        for (int t = 0; t < DAFM_JPL; ++t) {
            const int j = lane + t * kWave;
            if (j < n) {
                const float4 *kj = reinterpret_cast<const float4 *>(k + (size_t)(r0 + j) * D + d0);
#pragma unroll
                for (int u4 = 0; u4 < 4; ++u4) {
                    const float4 kk = kj[u4];
                    dot[t] += qc[u4 * 4 + 0] * kk.x + qc[u4 * 4 + 1] * kk.y + qc[u4 * 4 + 2] * kk.z + qc[u4 * 4 + 3] * kk.w;
                }
            }
        }
    }
    float esum = 0.f, ex[DAFM_JPL];
#pragma unroll
    for (int t = 0; t < DAFM_JPL; ++t) { ex[t] = (lane + t * kWave) < n ? __expf(x[t] - xmax) : 0.f; esum += ex[t]; }
    esum = wave_sum(esum);
    float lg[DAFM_JPL], lmax = -__builtin_inff();
#pragma unroll
    for (int t = 0; t < DAFM_JPL; ++t) {
        lg[t] = (lane + t * kWave) < n ? dot[t] * (ex[t] / esum) * scale : -__builtin_inff();
        lmax = fmaxf(lmax, lg[t]);
    }
    lmax = wave_max(lmax);
    float psum = 0.f, p[DAFM_JPL];
#pragma unroll
    for (int t = 0; t < DAFM_JPL; ++t) { p[t] = (lane + t * kWave) < n ? __expf(lg[t] - lmax) : 0.f; psum += p[t]; }
    psum = wave_sum(psum);
#pragma unroll
    for (int t = 0; t < DAFM_JPL; ++t) {
        p[t] /= psum;
        if (lane + t * kWave < n) att_row[lane + t * kWave] = p[t];
    }

    // ---- out_i = sum_j att_ij v_j, lanes along d (D/64 floats per lane, 8 at D = 512) ----
    for (int d0 = lane * 4; d0 < D; d0 += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < n; ++j) {
            const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, j < kWave ? p[0] : p[1]), j & 63));
            const float4 vv = *reinterpret_cast<const float4 *>(v + (size_t)(r0 + j) * D + d0);
            acc.x += a * vv.x; acc.y += a * vv.y; acc.z += a * vv.z; acc.w += a * vv.w;
        }
        *reinterpret_cast<float4 *>(out + (size_t)row * D + d0) = acc;
    }
}

// Backward, pass A (one wave per query row i):
//   dAtt_ij = gO_i . v_j ;  dS_ij = att_ij (dAtt_ij - sum_j' att_ij' dAtt_ij') ;
//   G_ij = dS_ij * E_ij * scale  (= dL/d(q_i . k_j), written to gmat) ;  dq_i = sum_j G_ij k_j
__global__ __launch_bounds__(256) void dafm_bwd_rows_kernel(int S, int total_rows, int D, const int *__restrict__ scene_off,
                                                            const int *__restrict__ de_off, const float *__restrict__ k,
                                                            const float *__restrict__ v, const float *__restrict__ de,
                                                            float inv_sigma, float scale, const float *__restrict__ att,
                                                            const float *__restrict__ grad_out, float *__restrict__ gmat,
                                                            float *__restrict__ grad_q) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= total_rows) return;
    const int s = scene_of_row(row, S, scene_off);
    const int r0 = scene_off[s], n = scene_off[s + 1] - r0, i = row - r0;
    const size_t mo = de_off[s] + (size_t)i * n;
    float x[DAFM_JPL], a[DAFM_JPL], dot[DAFM_JPL];
    float xmax = -__builtin_inff();
#pragma unroll
    for (int t = 0; t < DAFM_JPL; ++t) {
        const int j = lane + t * kWave;
        x[t] = j < n ? -(de[mo + j] * inv_sigma) : -__builtin_inff();
        a[t] = j < n ? att[mo + j] : 0.f;
        xmax = fmaxf(xmax, x[t]);
        dot[t] = 0.f;
    }
    xmax = wave_max(xmax);
    float esum = 0.f, ex[DAFM_JPL];
#pragma unroll
    for (int t = 0; t < DAFM_JPL; ++t) { ex[t] = (lane + t * kWave) < n ? __expf(x[t] - xmax) : 0.f; esum += ex[t]; }
    esum = wave_sum(esum);
    cfloat_p gi = (cfloat_p)(grad_out + (size_t)row * D);
    for (int d0 = 0; d0 < D; d0 += 16) {
        float gc[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) gc[u] = gi[d0 + u];
#pragma unroll
        for (int t = 0; t < DAFM_JPL; ++t) {
            const int j = lane + t * kWave;
            if (j < n) {
                const float4 *vj = reinterpret_cast<const float4 *>(v + (size_t)(r0 + j) * D + d0);
#pragma unroll
                for (int u4 = 0; u4 < 4; ++u4) {
                    const float4 vv = vj[u4];
                    dot[t] += gc[u4 * 4 + 0] * vv.x + gc[u4 * 4 + 1] * vv.y + gc[u4 * 4 + 2] * vv.z + gc[u4 * 4 + 3] * vv.w;
                }
            }
        }
    }
    float rsum = 0.f;
#pragma unroll
    for (int t = 0; t < DAFM_JPL; ++t) rsum += a[t] * dot[t];
    rsum = wave_sum(rsum);
    float g[DAFM_JPL];
#pragma unroll
    for (int t = 0; t < DAFM_JPL; ++t) {
        g[t] = a[t] * (dot[t] - rsum) * (ex[t] / esum) * scale;
        if (lane + t * kWave < n) gmat[mo + lane + t * kWave] = g[t];
    }
    for (int d0 = lane * 4; d0 < D; d0 += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < n; ++j) {
            const float gj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, j < kWave ? g[0] : g[1]), j & 63));
            const float4 kk = *reinterpret_cast<const float4 *>(k + (size_t)(r0 + j) * D + d0);
            acc.x += gj * kk.x; acc.y += gj * kk.y; acc.z += gj * kk.z; acc.w += gj * kk.w;
        }
        *reinterpret_cast<float4 *>(grad_q + (size_t)row * D + d0) = acc;
    }
}

// Backward, pass B (one wave per key row j):  dk_j = sum_i G_ij q_i ;  dv_j = sum_i att_ij gO_i
__global__ __launch_bounds__(256) void dafm_bwd_cols_kernel(int S, int total_rows, int D, const int *__restrict__ scene_off,
                                                            const int *__restrict__ de_off, const float *__restrict__ q,
                                                            const float *__restrict__ att, const float *__restrict__ grad_out,
                                                            const float *__restrict__ gmat, float *__restrict__ grad_k,
                                                            float *__restrict__ grad_v) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= total_rows) return;
    const int s = scene_of_row(row, S, scene_off);
    const int r0 = scene_off[s], n = scene_off[s + 1] - r0, j = row - r0;
    cfloat_p G = (cfloat_p)(gmat + de_off[s]);
    cfloat_p A = (cfloat_p)(att + de_off[s]);
    for (int d0 = lane * 4; d0 < D; d0 += 256) {
        float4 ak = make_float4(0.f, 0.f, 0.f, 0.f), av = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = 0; i < n; ++i) {
            const float gij = G[(size_t)i * n + j], aij = A[(size_t)i * n + j];
            const float4 qq = *reinterpret_cast<const float4 *>(q + (size_t)(r0 + i) * D + d0);
            const float4 go = *reinterpret_cast<const float4 *>(grad_out + (size_t)(r0 + i) * D + d0);
            ak.x += gij * qq.x; ak.y += gij * qq.y; ak.z += gij * qq.z; ak.w += gij * qq.w;
            av.x += aij * go.x; av.y += aij * go.y; av.z += aij * go.z; av.w += aij * go.w;
        }
        *reinterpret_cast<float4 *>(grad_k + (size_t)row * D + d0) = ak;
        *reinterpret_cast<float4 *>(grad_v + (size_t)row * D + d0) = av;
    }
}

}  // namespace mgar

using namespace mgar;

extern "C" __attribute__((visibility("default"))) int mgar_dafm_attn_fwd(int S, int total_rows, int D, const int *scene_off,
                                                                        const int *de_off, const float *q, const float *k,
                                                                        const float *v, const float *de, float sigma,
                                                                        float scale, float *att, float *out, void *stream) {
    MGAR_REQUIRE(S >= 0 && total_rows >= 0 && D > 0 && D % 64 == 0, "dafm_attn_fwd: bad sizes (D must be a multiple of 64)");
    MGAR_REQUIRE(sigma != 0.f, "dafm_attn_fwd: sigma == 0");
    if (S == 0 || total_rows == 0) return MGAR_OK;
    MGAR_REQUIRE(scene_off && de_off && q && k && v && de && att && out, "dafm_attn_fwd: null pointer");
    // attention core only (the projections / FFN are library GEMMs): q, k, v read + out written (4 * rows * D floats), De read and
    // Att written (rows * n floats each, n = rows / S actors per scene); QK^T and Att V: 4 * rows * n * D flop
    const double n_avg = (double)total_rows / S;
    KtScope kt(KT_DAFM_FWD, (hipStream_t)stream, 4.0 * (4.0 * total_rows * D + 2.0 * total_rows * n_avg), 4.0 * total_rows * n_avg * D);
    hipLaunchKernelGGL(dafm_fwd_kernel, dim3(ceil_div(total_rows, 4)), dim3(256), 0, (hipStream_t)stream, S, total_rows, D,
                       scene_off, de_off, q, k, v, de, 1.0f / sigma, scale, att, out);
    return check_launch("dafm_attn_fwd: launch failed");
}

extern "C" __attribute__((visibility("default"))) int mgar_dafm_attn_bwd(int S, int total_rows, int D, const int *scene_off,
                                                                        const int *de_off, const float *q, const float *k,
                                                                        const float *v, const float *de, float sigma,
                                                                        float scale, const float *att, const float *grad_out,
                                                                        float *gmat, float *grad_q, float *grad_k,
                                                                        float *grad_v, void *stream) {
    MGAR_REQUIRE(S >= 0 && total_rows >= 0 && D > 0 && D % 64 == 0, "dafm_attn_bwd: bad sizes (D must be a multiple of 64)");
    MGAR_REQUIRE(sigma != 0.f, "dafm_attn_bwd: sigma == 0");
    if (S == 0 || total_rows == 0) return MGAR_OK;
    MGAR_REQUIRE(scene_off && de_off && q && k && v && de && att && grad_out && gmat && grad_q && grad_k && grad_v,
                 "dafm_attn_bwd: null pointer");
    // q, k, v, grad_out read + 3 gradients written (7 * rows * D), De / Att read + the (rows, n) scratch written and read;
    // dAtt = gO V^T, dq = dS K, dk = dS^T Q, dv = Att^T gO: 8 * rows * n * D flop
    const double n_avg = (double)total_rows / S;
    KtScope kt(KT_DAFM_BWD, (hipStream_t)stream, 4.0 * (7.0 * total_rows * D + 4.0 * total_rows * n_avg), 8.0 * total_rows * n_avg * D);
    hipLaunchKernelGGL(dafm_bwd_rows_kernel, dim3(ceil_div(total_rows, 4)), dim3(256), 0, (hipStream_t)stream, S, total_rows,
                       D, scene_off, de_off, k, v, de, 1.0f / sigma, scale, att, grad_out, gmat, grad_q);
    hipLaunchKernelGGL(dafm_bwd_cols_kernel, dim3(ceil_div(total_rows, 4)), dim3(256), 0, (hipStream_t)stream, S, total_rows,
                       D, scene_off, de_off, q, att, grad_out, gmat, grad_k, grad_v);
    return check_launch("dafm_attn_bwd: launch failed");
}
