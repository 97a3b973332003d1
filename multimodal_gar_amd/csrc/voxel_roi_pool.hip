// voxel_roi_pool.hip -- fused Voxel-RoI pooling (one scale of NeighborVoxelSAModuleMSG) for gfx950.
//
// Replaces the torch op chain of the reference's
//   pcdet/ops/pointnet2/pointnet2_stack/voxel_pool_modules.py:86-126
// between `mlps_in` and `mlps_out`:
//   voxel_query -> group features -> zero empty balls -> group xyz -> subtract the query -> zero ->
//   mlps_pos = Conv2d(3 -> C, no bias) + BatchNorm2d (train: statistics over ALL M*nsample columns) ->
//   add -> ReLU -> max over nsample
// which materialises two (C, M, nsample) tensors and a (3, M, nsample) one and passes over them ~8 times (1.7 GB each per
// scale at config c3).  Here nothing of size M*nsample*C is ever written:
//
//   * the position branch is LINEAR in the relative coordinates r (3 numbers per neighbour): p_c = w_c . r.  Its BatchNorm
//     statistics therefore follow from the first and second moments of r over all columns,
//         mean_c = w_c . E[r],   var_c = w_c^T Cov(r) w_c,
//     9 sums that `vrp_moments_kernel` takes from the index tensor alone (4 B per column + cached xyz gathers);
//   * `vrp_fwd_kernel`: lanes along the C <= 32 channels, a half-wave per query: every neighbour is ONE contiguous
//     128-byte gather of the projected voxel features, the position term is 3 FMAs on values the lane already holds,
//     ReLU and the max over nsample happen in registers; the (C, M) channel-major result (what the following Conv1d
//     consumes) + 1-byte arg-max leave through an LDS transpose as 256-byte runs;
//   * backward: only the arg-max neighbour of each (query, channel) receives gradient, and the BatchNorm backward of the
//     position branch needs, besides the moments, just five per-channel sums over those M*C entries -- the weight
//     gradient is assembled in closed form (see vrp_bwd_finalize_kernel).
// Empty neighbourhoods (voxel_query wrote idx[m][0] = -1): features and relative coordinates are zero, as the reference
// zeroes them (voxel_pool_modules.py:101,106); they count in the statistics.
#include "common.hpp"
#include "payload.hpp"

namespace mgar {

constexpr int VP_THREADS = 256;
constexpr int VP_Q = 64;      // queries per workgroup of the forward / backward kernels
constexpr int VP_MAXC = 32;   // channels = lanes of a half-wave

// ---- moments of r over all columns: partial[blk][9] = sum r (3), sum r r^T (xx, xy, xz, yy, yz, zz) ----------------
__device__ __forceinline__ double block_sum_f64(double v, double *scratch) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < VP_THREADS / 64; ++i) t += scratch[i];
    return t;
}

__global__ __launch_bounds__(VP_THREADS) void vrp_moments_kernel(long long total, int nsample, const float *__restrict__ xyz,
                                                                 const float *__restrict__ new_xyz, const int *__restrict__ idx,
                                                                 double *__restrict__ partial) {
    __shared__ double scratch[VP_THREADS / 64];
    float s[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long long e = (long long)blockIdx.x * VP_THREADS + threadIdx.x; e < total; e += (long long)gridDim.x * VP_THREADS) {
        const long long m = e / nsample;
        if (idx[m * nsample] < 0) continue;   // empty neighbourhood: r = 0
        const int j = idx[e];
        const float rx = xyz[(size_t)j * 3 + 0] - new_xyz[m * 3 + 0];
        const float ry = xyz[(size_t)j * 3 + 1] - new_xyz[m * 3 + 1];
        const float rz = xyz[(size_t)j * 3 + 2] - new_xyz[m * 3 + 2];
        s[0] += rx; s[1] += ry; s[2] += rz;
        s[3] += rx * rx; s[4] += rx * ry; s[5] += rx * rz; s[6] += ry * ry; s[7] += ry * rz; s[8] += rz * rz;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const double t = block_sum_f64((double)s[k], scratch);
        if (threadIdx.x == 0) partial[(size_t)blockIdx.x * 9 + k] = t;
    }
}

// one workgroup: moments[0..2] = E[r], moments[3..8] = Cov(r) (biased; xx, xy, xz, yy, yz, zz), moments[9] = n;
// per channel: mean / invstd of p_c = w_c . r and the running-statistics update of the BatchNorm2d
__global__ __launch_bounds__(64) void vrp_stats_finalize_kernel(const double *__restrict__ partial, int nblk, double n,
                                                                const float *__restrict__ w_pos, int C, float eps, float momentum,
                                                                double *__restrict__ moments, float *__restrict__ mean,
                                                                float *__restrict__ invstd, float *__restrict__ running_mean,
                                                                float *__restrict__ running_var,
                                                                long long *__restrict__ num_batches_tracked) {
    __shared__ double mo[10];
    if (threadIdx.x < 9) {
        double t = 0.0;
        for (int b = 0; b < nblk; ++b) t += partial[(size_t)b * 9 + threadIdx.x];   // fixed order: reproducible
        mo[threadIdx.x] = t / n;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double ex = mo[0], ey = mo[1], ez = mo[2];
        mo[3] -= ex * ex; mo[4] -= ex * ey; mo[5] -= ex * ez; mo[6] -= ey * ey; mo[7] -= ey * ez; mo[8] -= ez * ez;
        mo[9] = n;
        for (int k = 0; k < 10; ++k) moments[k] = mo[k];
        if (num_batches_tracked) *num_batches_tracked += 1;
    }
    __syncthreads();
    const int c = threadIdx.x;
    if (c >= C) return;
    const double w0 = w_pos[c * 3 + 0], w1 = w_pos[c * 3 + 1], w2 = w_pos[c * 3 + 2];
    const double m = w0 * mo[0] + w1 * mo[1] + w2 * mo[2];
    double var = w0 * (mo[3] * w0 + mo[4] * w1 + mo[5] * w2) + w1 * (mo[4] * w0 + mo[6] * w1 + mo[7] * w2) +
                 w2 * (mo[5] * w0 + mo[7] * w1 + mo[8] * w2);
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(n > 1.0 ? var * n / (n - 1.0) : var);
}

// ---- forward ---------------------------------------------------------------------------------------------------------
// grid ceil(M / VP_Q).  feats (N, ld_f) row-major: the OUTPUT of mlps_in (Conv1d + BatchNorm1d) on all voxels.
template <typename T>
__global__ __launch_bounds__(VP_THREADS) void vrp_fwd_kernel(int M, int nsample, int C, const float *__restrict__ xyz,
                                                             const float *__restrict__ new_xyz, const T *__restrict__ feats,
                                                             int ld_f, const int *__restrict__ idx,
                                                             const float *__restrict__ w_pos, const float *__restrict__ mean,
                                                             const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, T *__restrict__ pooled,
                                                             unsigned char *__restrict__ arg) {
    __shared__ float tile[VP_Q][VP_MAXC + 1];
    __shared__ unsigned char targ[VP_Q][VP_MAXC + 4];
    const int c = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const bool live = c < C;
    const float w0 = live ? w_pos[c * 3 + 0] : 0.f, w1 = live ? w_pos[c * 3 + 1] : 0.f, w2 = live ? w_pos[c * 3 + 2] : 0.f;
    const float sc = live ? invstd[c] * (gamma ? gamma[c] : 1.f) : 0.f;
    const float mu = live ? mean[c] : 0.f, sh = live && beta ? beta[c] : 0.f;
    const int m0 = blockIdx.x * VP_Q;
    for (int q = hw; q < VP_Q; q += VP_THREADS / 32) {
        const int m = m0 + q;
        if (m >= M) break;
        const int *id = idx + (size_t)m * nsample;
        float best, qx = new_xyz[(size_t)m * 3], qy = new_xyz[(size_t)m * 3 + 1], qz = new_xyz[(size_t)m * 3 + 2];
        int bi = 0;
        if (id[0] < 0) {
            best = (0.f - mu) * sc + sh;            // every column of an empty neighbourhood: BN(0) + 0
        } else {
            best = -__builtin_inff();
#pragma unroll 4
            for (int s = 0; s < nsample; ++s) {
                const int j = id[s];
                const float rx = xyz[(size_t)j * 3] - qx, ry = xyz[(size_t)j * 3 + 1] - qy, rz = xyz[(size_t)j * 3 + 2] - qz;
                const float p = w0 * rx + w1 * ry + w2 * rz;
                const float f = live ? Payload<T>::ld(feats + (size_t)j * ld_f + c) : 0.f;
                const float v = f + ((p - mu) * sc + sh);
                if (v > best) { best = v; bi = s; }
            }
        }
        tile[q][c] = fmaxf(best, 0.f);
        targ[q][c] = (unsigned char)bi;
    }
    __syncthreads();
    const int nq = min(VP_Q, M - m0);
    for (int e = threadIdx.x; e < C * VP_Q; e += VP_THREADS) {   // channel-major (C, M): 64 consecutive queries per channel
        const int ch = e / VP_Q, q = e - ch * VP_Q;
        if (q < nq) {
            Payload<T>::st(pooled + (size_t)ch * M + m0 + q, tile[q][ch]);
            arg[(size_t)ch * M + m0 + q] = targ[q][ch];
        }
    }
}

// ---- backward --------------------------------------------------------------------------------------------------------
// g = dpooled where pooled > 0 (ReLU).  Every (query, channel) sends g to the feature row of its arg-max neighbour and
// adds to five per-channel sums: S0 = sum g, S1 = sum g * phat, S2..4 = sum g * r   (phat = (w_c . r - mean_c) * invstd_c).
// partial[blk][C][5].  grid ceil(M / VP_Q)
template <typename T>
__global__ __launch_bounds__(VP_THREADS) void vrp_bwd_kernel(int M, int nsample, int C, const float *__restrict__ xyz,
                                                             const float *__restrict__ new_xyz, const int *__restrict__ idx,
                                                             const float *__restrict__ w_pos, const float *__restrict__ mean,
                                                             const float *__restrict__ invstd, const T *__restrict__ dpooled,
                                                             const T *__restrict__ pooled, const unsigned char *__restrict__ arg,
                                                             float *__restrict__ dfeats, int ld_f, float *__restrict__ partial) {
    __shared__ float tile[VP_Q][VP_MAXC + 1];
    __shared__ unsigned char targ[VP_Q][VP_MAXC + 4];
    __shared__ float red[VP_THREADS / 32][VP_MAXC][5];
    const int m0 = blockIdx.x * VP_Q;
    const int nq = min(VP_Q, M - m0);
    for (int e = threadIdx.x; e < C * VP_Q; e += VP_THREADS) {   // coalesced reads of the channel-major gradient
        const int ch = e / VP_Q, q = e - ch * VP_Q;
        float g = 0.f;
        unsigned char a = 0;
        if (q < nq) {
            const size_t o = (size_t)ch * M + m0 + q;
            g = Payload<T>::ld(pooled + o) > 0.f ? Payload<T>::ld(dpooled + o) : 0.f;
            a = arg[o];
        }
        tile[q][ch] = g;
        targ[q][ch] = a;
    }
    __syncthreads();
    const int c = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const bool live = c < C;
    const float w0 = live ? w_pos[c * 3 + 0] : 0.f, w1 = live ? w_pos[c * 3 + 1] : 0.f, w2 = live ? w_pos[c * 3 + 2] : 0.f;
    const float mu = live ? mean[c] : 0.f, is = live ? invstd[c] : 0.f;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
    for (int q = hw; q < nq; q += VP_THREADS / 32) {
        const int m = m0 + q;
        const float g = live ? tile[q][c] : 0.f;
        const int *id = idx + (size_t)m * nsample;
        float rx = 0.f, ry = 0.f, rz = 0.f;
        if (id[0] >= 0) {
            const int a = live ? (int)targ[q][c] : 0;     // lanes beyond C hold no arg-max (their tile columns were never written)
            const int j = id[a < nsample ? a : 0];
            rx = xyz[(size_t)j * 3] - new_xyz[(size_t)m * 3];
            ry = xyz[(size_t)j * 3 + 1] - new_xyz[(size_t)m * 3 + 1];
            rz = xyz[(size_t)j * 3 + 2] - new_xyz[(size_t)m * 3 + 2];
            if (live && g != 0.f && dfeats) atomicAdd(dfeats + (size_t)j * ld_f + c, g);
        }
        const float ph = ((w0 * rx + w1 * ry + w2 * rz) - mu) * is;
        s0 += g; s1 += g * ph; s2 += g * rx; s3 += g * ry; s4 += g * rz;
    }
    red[hw][c][0] = s0; red[hw][c][1] = s1; red[hw][c][2] = s2; red[hw][c][3] = s3; red[hw][c][4] = s4;
    __syncthreads();
    for (int e = threadIdx.x; e < C * 5; e += VP_THREADS) {
        const int ch = e / 5, k = e - ch * 5;
        float t = 0.f;
        for (int h = 0; h < VP_THREADS / 32; ++h) t += red[h][ch][k];
        partial[((size_t)blockIdx.x * C + ch) * 5 + k] = t;
    }
}

// d gamma, d beta, d w_pos (C, 3) from the per-block sums and the moments.  With n = M*nsample columns,
//   gbar = S0 / n, gp = S1 / n,   sum_cols phat * r = invstd * n * Cov . w
//   d w_c = gamma_c * invstd_c * ( S2 - gbar * n * E[r] - gp * invstd_c * n * Cov . w_c )
// 1 024 lanes = 32 slices x 32 channels: a lane sums every 32nd block partial of its channel, the slices are then added in
// slice order (fixed order: reproducible).  (One lane per channel walking all M / 64 partials took 3.9 ms at config c3.)
__global__ __launch_bounds__(1024) void vrp_bwd_finalize_kernel(const float *__restrict__ partial, int nblk, int C,
                                                                const double *__restrict__ moments, const float *__restrict__ w_pos,
                                                                const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                                int train_stats, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                                float *__restrict__ dw_pos) {
    __shared__ double red[32][VP_MAXC][5];
    const int c = threadIdx.x & 31, sl = threadIdx.x >> 5;
    double s[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (c < C)
        for (int b = sl; b < nblk; b += 32)
            for (int k = 0; k < 5; ++k) s[k] += (double)partial[((size_t)b * C + c) * 5 + k];
    for (int k = 0; k < 5; ++k) red[sl][c][k] = s[k];
    __syncthreads();
    if (sl != 0 || c >= C) return;
    for (int k = 0; k < 5; ++k) {
        double t = 0.0;
        for (int q = 0; q < 32; ++q) t += red[q][c][k];
        s[k] = t;
    }
    if (dbeta) dbeta[c] = (float)s[0];
    if (dgamma) dgamma[c] = (float)s[1];
    const double g = gamma ? (double)gamma[c] : 1.0, is = invstd[c];
    double d[3] = {s[2], s[3], s[4]};
    if (train_stats) {
        const double w0 = w_pos[c * 3], w1 = w_pos[c * 3 + 1], w2 = w_pos[c * 3 + 2];
        const double cw[3] = {moments[3] * w0 + moments[4] * w1 + moments[5] * w2, moments[4] * w0 + moments[6] * w1 + moments[7] * w2,
                              moments[5] * w0 + moments[7] * w1 + moments[8] * w2};
        for (int k = 0; k < 3; ++k) d[k] -= s[0] * moments[k] + s[1] * is * cw[k];   // gbar * n = S0, gp * n = S1
    }
    for (int k = 0; k < 3; ++k) dw_pos[c * 3 + k] = (float)(g * is * d[k]);
}

static int vrp_blocks(int M) { return ceil_div(M, VP_Q); }
static int vrp_moment_blocks(long long total) {
    long long b = (total + VP_THREADS * 8 - 1) / (VP_THREADS * 8);
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace mgar

using namespace mgar;

#define VP_API extern "C" __attribute__((visibility("default")))

// workspace sizes: doubles for the statistics pass, floats for the backward
VP_API long long mgar_voxel_roi_pool_stats_workspace_doubles(int M, int nsample) {
    if (M < 0 || nsample < 0) return MGAR_EINVAL;
    return (long long)vrp_moment_blocks((long long)M * nsample) * 9;
}
VP_API long long mgar_voxel_roi_pool_bwd_workspace_floats(int M, int C) {
    if (M < 0 || C < 0) return MGAR_EINVAL;
    return (long long)vrp_blocks(M) * C * 5;
}

VP_API int mgar_voxel_roi_pool_stats(int M, int nsample, int C, const float *xyz, const float *new_xyz, const int *idx,
                                     const float *w_pos, float eps, float momentum, double *workspace, double *moments, float *mean,
                                     float *invstd, float *running_mean, float *running_var, long long *num_batches_tracked,
                                     void *stream) {
    MGAR_REQUIRE(M >= 0 && nsample >= 1 && C >= 1 && C <= VP_MAXC, "voxel_roi_pool_stats: bad sizes (1 <= C <= 32)");
    if (M == 0) return MGAR_OK;
    MGAR_REQUIRE(xyz && new_xyz && idx && w_pos && workspace && moments && mean && invstd, "voxel_roi_pool_stats: null pointer");
    const long long total = (long long)M * nsample;
    const int nblk = vrp_moment_blocks(total);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(vrp_moments_kernel, dim3(nblk), dim3(VP_THREADS), 0, st, total, nsample, xyz, new_xyz, idx, workspace);
    hipLaunchKernelGGL(vrp_stats_finalize_kernel, dim3(1), dim3(64), 0, st, workspace, nblk, (double)total, w_pos, C, eps, momentum,
                       moments, mean, invstd, running_mean, running_var, num_batches_tracked);
    return check_launch("voxel_roi_pool_stats: launch failed");
}

template <typename T>
static int vrp_fwd_impl(int M, int nsample, int C, const float *xyz, const float *new_xyz, const T *feats, int ld_f, const int *idx,
                        const float *w_pos, const float *mean, const float *invstd, const float *gamma, const float *beta, T *pooled,
                        unsigned char *arg, void *stream) {
    MGAR_REQUIRE(M >= 0 && nsample >= 1 && nsample <= 255 && C >= 1 && C <= VP_MAXC && ld_f >= C,
                 "voxel_roi_pool_fwd: bad sizes (1 <= C <= 32, nsample <= 255, ld >= C)");
    if (M == 0) return MGAR_OK;
    MGAR_REQUIRE(xyz && new_xyz && feats && idx && w_pos && mean && invstd && pooled && arg, "voxel_roi_pool_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    // minimum traffic: the index tensor, the queries, the (C, M) result + arg-max (the voxel rows are cache-resident gathers)
    KtScope kt(KT_VOXEL_ROI_POOL_FWD, st, (double)M * (4.0 * nsample + 12.0 + C * (sizeof(T) + 1.0)));
    hipLaunchKernelGGL(vrp_fwd_kernel<T>, dim3(vrp_blocks(M)), dim3(VP_THREADS), 0, st, M, nsample, C, xyz, new_xyz, feats, ld_f, idx,
                       w_pos, mean, invstd, gamma, beta, pooled, arg);
    return check_launch("voxel_roi_pool_fwd: launch failed");
}

VP_API int mgar_voxel_roi_pool_fwd(int M, int nsample, int C, const float *xyz, const float *new_xyz, const float *feats, int ld_f,
                                   const int *idx, const float *w_pos, const float *mean, const float *invstd, const float *gamma,
                                   const float *beta, float *pooled, unsigned char *arg, void *stream) {
    return vrp_fwd_impl<float>(M, nsample, C, xyz, new_xyz, feats, ld_f, idx, w_pos, mean, invstd, gamma, beta, pooled, arg, stream);
}
VP_API int mgar_voxel_roi_pool_fwd_bf16(int M, int nsample, int C, const float *xyz, const float *new_xyz, const void *feats, int ld_f,
                                        const int *idx, const float *w_pos, const float *mean, const float *invstd,
                                        const float *gamma, const float *beta, void *pooled, unsigned char *arg, void *stream) {
    return vrp_fwd_impl<bf16_t>(M, nsample, C, xyz, new_xyz, (const bf16_t *)feats, ld_f, idx, w_pos, mean, invstd, gamma, beta,
                                (bf16_t *)pooled, arg, stream);
}

VP_API int mgar_voxel_roi_pool_bwd(int M, int nsample, int C, const float *xyz, const float *new_xyz, const int *idx,
                                   const float *w_pos, const float *mean, const float *invstd, const float *gamma,
                                   const double *moments, int train_stats, const float *dpooled, const float *pooled,
                                   const unsigned char *arg, float *workspace, float *dfeats, int ld_f, float *dgamma, float *dbeta,
                                   float *dw_pos, void *stream) {
    MGAR_REQUIRE(M >= 0 && nsample >= 1 && C >= 1 && C <= VP_MAXC && (dfeats == nullptr || ld_f >= C), "voxel_roi_pool_bwd: bad sizes");
    if (M == 0) return MGAR_OK;
    MGAR_REQUIRE(xyz && new_xyz && idx && w_pos && mean && invstd && dpooled && pooled && arg && workspace && dw_pos &&
                     (moments || !train_stats), "voxel_roi_pool_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = vrp_blocks(M);
    { KtScope kt(KT_VOXEL_ROI_POOL_BWD, st, (double)M * (4.0 * nsample + 12.0 + C * 13.0));
    hipLaunchKernelGGL(vrp_bwd_kernel<float>, dim3(nblk), dim3(VP_THREADS), 0, st, M, nsample, C, xyz, new_xyz, idx, w_pos, mean, invstd,
                       dpooled, pooled, arg, dfeats, ld_f, workspace);
    }
    hipLaunchKernelGGL(vrp_bwd_finalize_kernel, dim3(1), dim3(1024), 0, st, workspace, nblk, C, moments, w_pos, invstd, gamma, train_stats,
                       dgamma, dbeta, dw_pos);
    return check_launch("voxel_roi_pool_bwd: launch failed");
}
