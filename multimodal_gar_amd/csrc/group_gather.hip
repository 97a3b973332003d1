// group_gather.hip -- grouping / gathering of point features (fwd + bwd), batch and
// stack layouts, for gfx950.
//
// Replaces  pointnet2_batch/src/group_points_gpu.cu:14-92, sampling_gpu.cu:15-90
//           pointnet2_stack/src/group_points_gpu.cu:15-125
//
// These are pure data movement (HBM-bound; algorithmic bytes in DESIGN.md):
//   batch fwd : out[b,c,p,s] = points[b,c,idx[b,p,s]]
//       One thread per (p,s) output column; it reads its index ONCE and walks a chunk of
//       channels, so idx is not re-read per channel as in the reference's
//       (npoints*nsample, c, b) grid.  Stores are coalesced along (p,s); the gathers hit
//       one (b,c) row of N floats (64 KB at N=16384), which stays in L2.
//   batch bwd : the reference scatters with one global atomicAdd per element.  Here one
//       workgroup owns one (b,c) row of grad_points: it accumulates the whole row in LDS
//       (ds_add_f32; N <= 36864 floats fit the 160 KB LDS) and adds it to the caller's
//       buffer with plain coalesced read-modify-write.  No global atomics, and the only
//       run-to-run variation left is the order of LDS adds inside one row.
//       Rows too long for LDS fall back to global float atomics.
//   stack fwd : out[m,c,s] = features[start_b + idx[m,s], c]  -- feature rows are
//       contiguous (C floats), the output is its transpose per query; a workgroup stages
//       the [nsample][C] gather of a few queries in LDS (row reads coalesced along c) and
//       writes the (c,s) image coalesced.
//   stack bwd : lanes run along c, so each wave-instruction of atomics covers whole
//       contiguous 4*C-byte row segments (the full-rate shape for memory-side float
//       atomics on MI355X).
#include "common.hpp"

namespace mgar {

constexpr int GG_THREADS = 256;
constexpr int GG_CCHUNK = 8;  // channels walked per thread in the batch forward

// ------------------------------- batch forward ------------------------------------
// grid: (ceil(cols/256), ceil(c/GG_CCHUNK), b); cols = npoints*nsample (nsample = 1: gather)
__global__ __launch_bounds__(GG_THREADS) void group_batch_fwd_kernel(int c, int n, int cols,
                                                                     const float *__restrict__ points,
                                                                     const int *__restrict__ idx,
                                                                     float *__restrict__ out) {
    const int col = blockIdx.x * GG_THREADS + threadIdx.x;
    if (col >= cols) return;
    const int bs = blockIdx.z;
    const int c0 = blockIdx.y * GG_CCHUNK;
    const int c1 = min(c0 + GG_CCHUNK, c);
    const int k = idx[(size_t)bs * cols + col];
    const float *src = points + ((size_t)bs * c + c0) * n + k;
    float *dst = out + ((size_t)bs * c + c0) * cols + col;
#pragma unroll 4
    for (int ci = c0; ci < c1; ++ci) {
        *dst = *src;
        src += n;
        dst += cols;
    }
}

// ------------------------------- batch backward -----------------------------------
// grid: (c, b); LDS: n floats
__global__ __launch_bounds__(1024) void group_batch_bwd_lds_kernel(int c, int n, int cols,
                                                                   const float *__restrict__ grad_out,
                                                                   const int *__restrict__ idx,
                                                                   float *__restrict__ grad_points) {
    extern __shared__ float row[];
    const int ci = blockIdx.x, bs = blockIdx.y;
    for (int i = threadIdx.x; i < n; i += blockDim.x) row[i] = 0.f;
    __syncthreads();
    const float *g = grad_out + ((size_t)bs * c + ci) * cols;
    const int *id = idx + (size_t)bs * cols;
    for (int e = threadIdx.x; e < cols; e += blockDim.x) atomicAdd(&row[id[e]], g[e]);
    __syncthreads();
    float *dst = grad_points + ((size_t)bs * c + ci) * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = row[i];
        if (v != 0.f) dst[i] += v;  // accumulate into the caller's (zero-filled) buffer
    }
}

__global__ __launch_bounds__(GG_THREADS) void group_batch_bwd_atomic_kernel(int c, int n, int cols,
                                                                            const float *__restrict__ grad_out,
                                                                            const int *__restrict__ idx,
                                                                            float *__restrict__ grad_points) {
    const int col = blockIdx.x * GG_THREADS + threadIdx.x;
    if (col >= cols) return;
    const int ci = blockIdx.y, bs = blockIdx.z;
    const int k = idx[(size_t)bs * cols + col];
    atomicAdd(grad_points + ((size_t)bs * c + ci) * n + k, grad_out[((size_t)bs * c + ci) * cols + col]);
}

// ------------------------------- stack forward ------------------------------------
// One workgroup handles GS_Q queries.  Phase 1: lanes along c read feature rows
// (coalesced) into LDS tile[q][s][c+pad]; phase 2: lanes along (c,s) write out[m,c,s].
constexpr int GS_QMAX = 64;       // queries per workgroup: chosen so a tile is ~4K floats
constexpr int GS_TILE_FLOATS = 4096;

static inline int gs_queries(int per_q) {
    int q = GS_TILE_FLOATS / (per_q > 0 ? per_q : 1);
    return q < 1 ? 1 : (q > GS_QMAX ? GS_QMAX : q);
}

__global__ __launch_bounds__(GG_THREADS) void group_stack_fwd_kernel(int GS_Q, int B, int M, int C, int nsample,
                                                                     const float *__restrict__ features,
                                                                     const int *__restrict__ features_batch_cnt,
                                                                     const int *__restrict__ idx,
                                                                     const int *__restrict__ idx_batch_cnt,
                                                                     float *__restrict__ out) {
    extern __shared__ float tile[];  // [GS_Q][nsample][C+1]
    __shared__ int seg_start[GS_QMAX];
    const int m0 = blockIdx.x * GS_Q;
    const int nq = min(GS_Q, M - m0);
    const int CP = C + 1;
    const int per_q = nsample * C;
    if (threadIdx.x < nq) seg_start[threadIdx.x] = find_segment(m0 + threadIdx.x, B, idx_batch_cnt, features_batch_cnt).b_start;
    __syncthreads();
    // phase 1
    for (int e = threadIdx.x; e < nq * per_q; e += GG_THREADS) {
        const int ql = e / per_q, r = e - ql * per_q;
        const int s = r / C, ci = r - s * C;
        const int k = idx[(size_t)(m0 + ql) * nsample + s];
        tile[(ql * nsample + s) * CP + ci] = features[((size_t)seg_start[ql] + k) * C + ci];
    }
    __syncthreads();
    // phase 2
    float *dst = out + (size_t)m0 * per_q;
    for (int e = threadIdx.x; e < nq * per_q; e += GG_THREADS) {
        const int ql = e / per_q, r = e - ql * per_q;
        const int ci = r / nsample, s = r - ci * nsample;
        dst[e] = tile[(ql * nsample + s) * CP + ci];
    }
}

// ------------------------------- stack backward -----------------------------------
// lanes along c inside a (m, s) pair: each group of C lanes adds one contiguous row.
__global__ __launch_bounds__(GG_THREADS) void group_stack_bwd_kernel(int GS_Q, int B, int M, int C, int nsample,
                                                                     const float *__restrict__ grad_out,
                                                                     const int *__restrict__ idx,
                                                                     const int *__restrict__ idx_batch_cnt,
                                                                     const int *__restrict__ features_batch_cnt,
                                                                     float *__restrict__ grad_features) {
    extern __shared__ float tile[];  // [GS_Q][C][nsample+1]
    __shared__ int seg_start[GS_QMAX];
    const int m0 = blockIdx.x * GS_Q;
    const int nq = min(GS_Q, M - m0);
    const int SP = nsample + 1;
    const int per_q = nsample * C;
    if (threadIdx.x < nq) seg_start[threadIdx.x] = find_segment(m0 + threadIdx.x, B, idx_batch_cnt, features_batch_cnt).b_start;
    const float *src = grad_out + (size_t)m0 * per_q;
    for (int e = threadIdx.x; e < nq * per_q; e += GG_THREADS) {  // coalesced read of (c,s) images
        const int ql = e / per_q, r = e - ql * per_q;
        const int ci = r / nsample, s = r - ci * nsample;
        tile[(ql * C + ci) * SP + s] = src[e];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nq * per_q; e += GG_THREADS) {
        const int ql = e / per_q, r = e - ql * per_q;
        const int s = r / C, ci = r - s * C;
        const int k = idx[(size_t)(m0 + ql) * nsample + s];
        atomicAdd(grad_features + ((size_t)seg_start[ql] + k) * C + ci, tile[(ql * C + ci) * SP + s]);
    }
}

static int launch_group_batch_fwd(int b, int c, int n, int cols, const float *points, const int *idx, float *out,
                                  hipStream_t st) {
    if (b == 0 || c == 0 || cols == 0) return MGAR_OK;
    dim3 grid(ceil_div(cols, GG_THREADS), ceil_div(c, GG_CCHUNK), b);
    hipLaunchKernelGGL(group_batch_fwd_kernel, grid, dim3(GG_THREADS), 0, st, c, n, cols, points, idx, out);
    return check_launch("group/gather fwd: launch failed");
}

constexpr int GG_LDS_MAX_FLOATS = 36864;  // 144 KB of the 160 KB LDS

static int launch_group_batch_bwd(int b, int c, int n, int cols, const float *grad_out, const int *idx,
                                  float *grad_points, hipStream_t st) {
    if (b == 0 || c == 0 || cols == 0) return MGAR_OK;
    if (n <= GG_LDS_MAX_FLOATS) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void *)group_batch_bwd_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                GG_LDS_MAX_FLOATS * (int)sizeof(float));
            attr_set = true;
        }
        const int threads = cols >= 4096 ? 1024 : 256;
        hipLaunchKernelGGL(group_batch_bwd_lds_kernel, dim3(c, b), dim3(threads), (size_t)n * sizeof(float), st, c, n,
                           cols, grad_out, idx, grad_points);
    } else {
        dim3 grid(ceil_div(cols, GG_THREADS), c, b);
        hipLaunchKernelGGL(group_batch_bwd_atomic_kernel, grid, dim3(GG_THREADS), 0, st, c, n, cols, grad_out, idx,
                           grad_points);
    }
    return check_launch("group/gather bwd: launch failed");
}

}  // namespace mgar

using namespace mgar;

extern "C" __attribute__((visibility("default"))) int mgar_group_points_batch(int b, int c, int n, int npoints, int nsample, const float *points,
                                       const int *idx, float *out, void *stream) {
    MGAR_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0, "group_points_batch: negative size");
    MGAR_REQUIRE(b <= 65535, "group_points_batch: b > 65535");
    if ((long long)b * c * npoints * nsample == 0) return MGAR_OK;
    MGAR_REQUIRE(points && idx && out, "group_points_batch: null pointer");
    return launch_group_batch_fwd(b, c, n, npoints * nsample, points, idx, out, (hipStream_t)stream);
}

extern "C" __attribute__((visibility("default"))) int mgar_group_points_grad_batch(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                                            const int *idx, float *grad_points, void *stream) {
    MGAR_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0, "group_points_grad_batch: negative size");
    MGAR_REQUIRE(b <= 65535 && c <= 65535, "group_points_grad_batch: b or c > 65535");
    if ((long long)b * c * npoints * nsample == 0) return MGAR_OK;
    MGAR_REQUIRE(grad_out && idx && grad_points, "group_points_grad_batch: null pointer");
    return launch_group_batch_bwd(b, c, n, npoints * nsample, grad_out, idx, grad_points, (hipStream_t)stream);
}

extern "C" __attribute__((visibility("default"))) int mgar_gather_points_batch(int b, int c, int n, int npoints, const float *points, const int *idx,
                                        float *out, void *stream) {
    MGAR_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0, "gather_points_batch: negative size");
    MGAR_REQUIRE(b <= 65535, "gather_points_batch: b > 65535");
    if ((long long)b * c * npoints == 0) return MGAR_OK;
    MGAR_REQUIRE(points && idx && out, "gather_points_batch: null pointer");
    return launch_group_batch_fwd(b, c, n, npoints, points, idx, out, (hipStream_t)stream);
}

extern "C" __attribute__((visibility("default"))) int mgar_gather_points_grad_batch(int b, int c, int n, int npoints, const float *grad_out, const int *idx,
                                             float *grad_points, void *stream) {
    MGAR_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0, "gather_points_grad_batch: negative size");
    MGAR_REQUIRE(b <= 65535 && c <= 65535, "gather_points_grad_batch: b or c > 65535");
    if ((long long)b * c * npoints == 0) return MGAR_OK;
    MGAR_REQUIRE(grad_out && idx && grad_points, "gather_points_grad_batch: null pointer");
    return launch_group_batch_bwd(b, c, n, npoints, grad_out, idx, grad_points, (hipStream_t)stream);
}

extern "C" __attribute__((visibility("default"))) int mgar_group_points_stack(int B, int M, int C, int nsample, const float *features,
                                       const int *features_batch_cnt, const int *idx, const int *idx_batch_cnt,
                                       float *out, void *stream) {
    MGAR_REQUIRE(B >= 0 && M >= 0 && C >= 0 && nsample >= 0, "group_points_stack: negative size");
    if ((long long)M * C * nsample == 0 || B == 0) return MGAR_OK;
    MGAR_REQUIRE(features && features_batch_cnt && idx && idx_batch_cnt && out, "group_points_stack: null pointer");
    const int GS_Q = gs_queries(nsample * C);
    const size_t lds = (size_t)GS_Q * nsample * (C + 1) * sizeof(float);
    if (lds > 64 * 1024) {
        set_error("group_points_stack: nsample*C too large for the LDS tile");
        return MGAR_EUNSUPPORTED;
    }
    hipLaunchKernelGGL(group_stack_fwd_kernel, dim3(ceil_div(M, GS_Q)), dim3(GG_THREADS), lds, (hipStream_t)stream,
                       GS_Q, B, M, C, nsample, features, features_batch_cnt, idx, idx_batch_cnt, out);
    return check_launch("group_points_stack: launch failed");
}

extern "C" __attribute__((visibility("default"))) int mgar_group_points_grad_stack(int B, int M, int C, int N, int nsample, const float *grad_out,
                                            const int *idx, const int *idx_batch_cnt, const int *features_batch_cnt,
                                            float *grad_features, void *stream) {
    MGAR_REQUIRE(B >= 0 && M >= 0 && C >= 0 && N >= 0 && nsample >= 0, "group_points_grad_stack: negative size");
    if ((long long)M * C * nsample == 0 || B == 0) return MGAR_OK;
    MGAR_REQUIRE(grad_out && features_batch_cnt && idx && idx_batch_cnt && grad_features,
                 "group_points_grad_stack: null pointer");
    const int GS_Q = gs_queries(nsample * C);
    const size_t lds = (size_t)GS_Q * C * (nsample + 1) * sizeof(float);
    if (lds > 64 * 1024) {
        set_error("group_points_grad_stack: nsample*C too large for the LDS tile");
        return MGAR_EUNSUPPORTED;
    }
    hipLaunchKernelGGL(group_stack_bwd_kernel, dim3(ceil_div(M, GS_Q)), dim3(GG_THREADS), lds, (hipStream_t)stream,
                       GS_Q, B, M, C, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features);
    return check_launch("group_points_grad_stack: launch failed");
}
