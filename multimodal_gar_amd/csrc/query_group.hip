// query_group.hip -- fused "query and group": relative xyz + grouped features written ONCE, in
// the layout the shared MLP's GEMM consumes.
//
// Replaces the torch op chains of the reference's
//   pointnet2_batch/pointnet2_utils.py:241-264  QueryAndGroup.forward  (dense batches)
//   pointnet2_stack/pointnet2_utils.py:123-159  QueryAndGroup.forward  (stacked batches)
// which, after ball_query, run: transpose xyz -> group xyz -> subtract centre -> (stack: zero the
// empty balls) -> group features -> (stack: zero) -> cat -> (stack: permute for the conv).  Every
// one of those is a full pass over the grouped tensor (1.7 GB per scale for the RoI-grid lift at
// config c3); the concatenation alone re-reads and re-writes all of it.  Here one kernel reads the
// neighbour indices and writes the final (3 + C)-channel tensor exactly once.
//
//   batch : out (B, 3+C, M, ns), same layout the reference's module returns.
//   stack : out (3+C, M*ns) CHANNEL-MAJOR -- what `mlp(new_features.permute(1,0,2).unsqueeze(0))`
//           (pointnet2_stack/pointnet2_modules.py:95-96) needs, so that permute copy disappears too.
//           Feature rows are gathered with lanes along c (one contiguous 128-byte piece per
//           neighbour) into an LDS tile and written with lanes along the columns (coalesced);
//           the backward does the same in reverse and adds whole contiguous row pieces with
//           float atomics (the full-rate shape on MI355X).
// The stack kernels consume the RAW ball-query output (idx[row][0] == -1 marks an empty ball,
// pointnet2_stack/src/ball_query_gpu.cu:65) so the host needs no mask / fix-up passes either.
#include "common.hpp"
#include "payload.hpp"

namespace mgar {

// ------------------------------------------------------------------------------------------
// batch
// ------------------------------------------------------------------------------------------
constexpr int QG_CCHUNK = 8;

// grid: (ceil(cols/256), 1 + ceil(c/QG_CCHUNK), b); blockIdx.y == 0 writes the 3 xyz rows.
// rel_out (may be NULL) and y_out are addressed with their own batch strides so that they can be
// two tensors, or rows 0..2 / 3.. of one (b, 3+c, npoints, nsample) tensor.
// wx (c, 3), optional: y[c] += wx[c] . rel  -- the xyz half of a first MLP layer whose feature
// half was applied to the un-grouped features beforehand ("project, then group").
// T = payload type of features / rel_out / y_out (float or bf16_t); xyz, new_xyz, wx and the arithmetic are fp32.
template <typename T>
__global__ __launch_bounds__(256) void qg_batch_fwd_kernel(int c, int n, int npoints, int nsample,
                                                           const float *__restrict__ xyz,
                                                           const float *__restrict__ new_xyz,
                                                           const T *__restrict__ features,
                                                           const float *__restrict__ wx,
                                                           const int *__restrict__ idx, T *__restrict__ rel_out,
                                                           size_t rel_bstride, T *__restrict__ y_out, size_t y_bstride) {
    const int cols = npoints * nsample;
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= cols) return;
    const int bs = blockIdx.z;
    const int k = idx[(size_t)bs * cols + col];
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    if (blockIdx.y == 0 || wx) {
        const int p = col / nsample;
        const float *q = new_xyz + ((size_t)bs * npoints + p) * 3;
        const float *s = xyz + ((size_t)bs * n + k) * 3;
        r0 = s[0] - q[0]; r1 = s[1] - q[1]; r2 = s[2] - q[2];
    }
    if (blockIdx.y == 0) {
        if (rel_out) {
            T *dst = rel_out + (size_t)bs * rel_bstride + col;
            Payload<T>::st(dst, r0);
            Payload<T>::st(dst + (size_t)cols, r1);
            Payload<T>::st(dst + (size_t)2 * cols, r2);
        }
        return;
    }
    const int c0 = (blockIdx.y - 1) * QG_CCHUNK;
    const int c1 = min(c0 + QG_CCHUNK, c);
    const T *src = features + ((size_t)bs * c + c0) * n + k;
    T *dst = y_out + (size_t)bs * y_bstride + (size_t)c0 * cols + col;
#pragma unroll 4
    for (int ci = c0; ci < c1; ++ci) {
        float v = Payload<T>::ld(src);
        if (wx) v += wx[ci * 3 + 0] * r0 + wx[ci * 3 + 1] * r1 + wx[ci * 3 + 2] * r2;
        Payload<T>::st(dst, v);
        src += n;
        dst += cols;
    }
}

// grid: (c, b); one workgroup owns one (b, c) row of grad_features, accumulated in LDS
__global__ __launch_bounds__(1024) void qg_batch_bwd_lds_kernel(int c, int n, int cols, const float *__restrict__ grad_y,
                                                                size_t y_bstride, const int *__restrict__ idx,
                                                                float *__restrict__ grad_features) {
    extern __shared__ float row[];
    const int ci = blockIdx.x, bs = blockIdx.y;
    for (int i = threadIdx.x; i < n; i += blockDim.x) row[i] = 0.f;
    __syncthreads();
    const float *g = grad_y + (size_t)bs * y_bstride + (size_t)ci * cols;
    const int *id = idx + (size_t)bs * cols;
    for (int e = threadIdx.x; e < cols; e += blockDim.x) atomicAdd(&row[id[e]], g[e]);
    __syncthreads();
    float *dst = grad_features + ((size_t)bs * c + ci) * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = row[i];
        if (v != 0.f) dst[i] += v;
    }
}

__global__ __launch_bounds__(256) void qg_batch_bwd_atomic_kernel(int c, int n, int cols, const float *__restrict__ grad_y,
                                                                  size_t y_bstride, const int *__restrict__ idx,
                                                                  float *__restrict__ grad_features) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= cols) return;
    const int ci = blockIdx.y, bs = blockIdx.z;
    atomicAdd(grad_features + ((size_t)bs * c + ci) * n + idx[(size_t)bs * cols + col],
              grad_y[(size_t)bs * y_bstride + (size_t)ci * cols + col]);
}

// ------------------------------------------------------------------------------------------
// stack
// ------------------------------------------------------------------------------------------
constexpr int QS_COLS = 128;   // columns (query, sample) per workgroup
constexpr int QS_CH = 32;      // channels per LDS pass (one 128-byte piece per neighbour)

struct QsTile {
    int src_row[QS_COLS];          // global feature row of each column, -1 = empty ball
    float rel[QS_COLS][3];         // neighbour xyz relative to the query (0 for an empty ball)
    float tile[QS_COLS][QS_CH + 1];
};

// common prologue: resolve the tile's columns to global source rows.  The tile's queries are
// consecutive, so one wave-uniform segment search (for the first query) serves every column that
// is still inside that sample; only columns past its end search again.
__device__ __forceinline__ int qs_prologue(QsTile &t, int B, int M, int nsample, const int *__restrict__ idx,
                                           const int *__restrict__ q_cnt, const int *__restrict__ p_cnt, int &col0) {
    __shared__ int seg_q_end, seg_p_start;
    const long long total = (long long)M * nsample;
    col0 = blockIdx.x * QS_COLS;
    const int ncol = (int)min((long long)QS_COLS, total - col0);
    if (threadIdx.x == 0) {
        const int m0 = col0 / nsample;
        const Segment sg = find_segment(m0, B, q_cnt, p_cnt);
        seg_q_end = sg.a_start + q_cnt[sg.bs];
        seg_p_start = sg.b_start;
    }
    __syncthreads();
    for (int cl = threadIdx.x; cl < ncol; cl += blockDim.x) {
        const int col = col0 + cl;
        const int m = col / nsample;
        const int k = idx[col];
        const int first = idx[(size_t)m * nsample];
        const int p_start = m < seg_q_end ? seg_p_start : find_segment(m, B, q_cnt, p_cnt).b_start;
        t.src_row[cl] = first < 0 ? -1 : p_start + k;
    }
    __syncthreads();
    return ncol;
}

template <typename T>
__global__ __launch_bounds__(256) void qg_stack_fwd_kernel(int B, int M, int C, int nsample, const float *__restrict__ xyz,
                                                           const int *__restrict__ xyz_batch_cnt,
                                                           const float *__restrict__ new_xyz,
                                                           const int *__restrict__ new_xyz_batch_cnt,
                                                           const T *__restrict__ features, int ld,
                                                           const float *__restrict__ wx, const int *__restrict__ idx,
                                                           T *__restrict__ rel_out, T *__restrict__ y_out) {
    __shared__ QsTile t;
    int col0;
    const int ncol = qs_prologue(t, B, M, nsample, idx, new_xyz_batch_cnt, xyz_batch_cnt, col0);
    const size_t ms = (size_t)M * nsample;
    // rows of rel_out: neighbour xyz relative to the query, zero for an empty ball
    for (int e = threadIdx.x; e < ncol * 3; e += 256) {
        const int r = e / ncol, cl = e - r * ncol;
        const int src = t.src_row[cl];
        const int m = (col0 + cl) / nsample;
        const float v = src < 0 ? 0.f : xyz[(size_t)src * 3 + r] - new_xyz[(size_t)m * 3 + r];
        t.rel[cl][r] = v;
        if (rel_out) Payload<T>::st(rel_out + (size_t)r * ms + col0 + cl, v);
    }
    __syncthreads();
    // rows of y_out: features (+ wx . rel), QS_CH channels per pass through the LDS tile
    for (int c0 = 0; c0 < C; c0 += QS_CH) {
        const int nch = min(QS_CH, C - c0);
        for (int e = threadIdx.x; e < ncol * QS_CH; e += 256) {      // lanes along c
            const int cl = e / QS_CH, ci = e - cl * QS_CH;
            const int src = t.src_row[cl];
            float v = 0.f;
            if (src >= 0 && ci < nch) {
                v = Payload<T>::ld(features + (size_t)src * ld + c0 + ci);
                if (wx) {
                    const float *w = wx + (size_t)(c0 + ci) * 3;
                    v += w[0] * t.rel[cl][0] + w[1] * t.rel[cl][1] + w[2] * t.rel[cl][2];
                }
            }
            t.tile[cl][ci] = v;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < nch * QS_COLS; e += 256) {     // lanes along the columns
            const int ci = e / QS_COLS, cl = e - ci * QS_COLS;
            if (cl < ncol) Payload<T>::st(y_out + (size_t)(c0 + ci) * ms + col0 + cl, t.tile[cl][ci]);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void qg_stack_bwd_kernel(int B, int M, int C, int nsample, const float *__restrict__ grad_y,
                                                           const int *__restrict__ idx,
                                                           const int *__restrict__ new_xyz_batch_cnt,
                                                           const int *__restrict__ xyz_batch_cnt,
                                                           float *__restrict__ grad_features, int ld) {
    __shared__ QsTile t;
    int col0;
    const int ncol = qs_prologue(t, B, M, nsample, idx, new_xyz_batch_cnt, xyz_batch_cnt, col0);
    const size_t ms = (size_t)M * nsample;
    for (int c0 = 0; c0 < C; c0 += QS_CH) {
        const int nch = min(QS_CH, C - c0);
        for (int e = threadIdx.x; e < nch * QS_COLS; e += 256) {     // coalesced reads of the gradient rows
            const int ci = e / QS_COLS, cl = e - ci * QS_COLS;
            if (cl < ncol) t.tile[cl][ci] = grad_y[(size_t)(c0 + ci) * ms + col0 + cl];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < ncol * QS_CH; e += 256) {      // lanes along c: contiguous atomic pieces
            const int cl = e / QS_CH, ci = e - cl * QS_CH;
            const int src = t.src_row[cl];
            if (src >= 0 && ci < nch) atomicAdd(grad_features + (size_t)src * ld + c0 + ci, t.tile[cl][ci]);
        }
        __syncthreads();
    }
}

// ---- stack backward without float atomics: inverted index + owner-computes ------------------------------------------------
// The atomic scatter above runs at the chip's float-atomic rate (~1.3 TB/s of added bytes; 1.7 ms per scale of the RoI-grid
// lift at config c3) and its sums depend on the arrival order.  Here every source row is summed by ONE owner in a fixed order:
//   1. qg_inv_count_kernel : source row of every column (col_src) + per-row reference counts (integer atomics: exact, cheap);
//   2. an exclusive scan of the counts (the caller's cumsum) -> offsets;
//   3. qg_inv_fill_kernel  : list[offsets[row] + cursor++] = column  (order inside a row's list is arbitrary here);
//   4. qg_stack_bwd_rows_kernel: a half-wave per source row sorts its list (rank sort: the summation order becomes the
//      column order, whatever step 3 did), then adds the listed rows of the ROW-MAJOR gradient g_t (M*nsample, C) -- one
//      contiguous 4*C-byte read per entry -- and stores its feature-gradient row once.  Rows nobody references are not
//      touched (the caller zero-fills).
__global__ __launch_bounds__(256) void qg_inv_count_kernel(int B, int M, int nsample, const int *__restrict__ idx,
                                                           const int *__restrict__ q_cnt, const int *__restrict__ p_cnt,
                                                           int *__restrict__ col_src, int *__restrict__ counts) {
    __shared__ int seg_q_end, seg_p_start;
    const long long total = (long long)M * nsample;
    const long long col0 = (long long)blockIdx.x * 256;
    if (threadIdx.x == 0) {
        const Segment sg = find_segment((int)(col0 / nsample), B, q_cnt, p_cnt);
        seg_q_end = sg.a_start + q_cnt[sg.bs];
        seg_p_start = sg.b_start;
    }
    __syncthreads();
    const long long col = col0 + threadIdx.x;
    if (col >= total) return;
    const int m = (int)(col / nsample);
    int src = -1;
    if (idx[(size_t)m * nsample] >= 0) {
        const int p_start = m < seg_q_end ? seg_p_start : find_segment(m, B, q_cnt, p_cnt).b_start;
        src = p_start + idx[col];
        atomicAdd(counts + src, 1);
    }
    col_src[col] = src;
}

__global__ __launch_bounds__(256) void qg_inv_fill_kernel(long long total, const int *__restrict__ col_src,
                                                          const int *__restrict__ offsets, int *__restrict__ cursor,
                                                          int *__restrict__ list) {
    const long long col = (long long)blockIdx.x * 256 + threadIdx.x;
    if (col >= total) return;
    const int src = col_src[col];
    if (src < 0) return;
    list[offsets[src] + atomicAdd(cursor + src, 1)] = (int)col;
}

constexpr int QR_CAP = 512;           // list entries a half-wave sorts in LDS at once (8 half-waves: 16 KB per workgroup)
constexpr int QR_PER_LANE = QR_CAP / 32;

__device__ __forceinline__ void qr_wave_sync() {            // LDS hand-over between the lanes of one wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// buf[0, len) ascending, len <= QR_CAP, by one half-wave.  The entries are distinct (column numbers), so the ranks are a
// permutation; every lane keeps its entries in registers while it counts, then all write back in place.
__device__ __forceinline__ void qr_rank_sort(int *buf, int len, int lane) {
    int e[QR_PER_LANE], r[QR_PER_LANE];
#pragma unroll
    for (int i = 0; i < QR_PER_LANE; ++i) {
        const int p = i * 32 + lane;
        e[i] = p < len ? buf[p] : 0x7fffffff;
        r[i] = 0;
    }
    for (int j = 0; j < len; ++j) {
        const int v = buf[j];                                // broadcast read
#pragma unroll
        for (int i = 0; i < QR_PER_LANE; ++i) r[i] += v < e[i];
    }
    qr_wave_sync();
#pragma unroll
    for (int i = 0; i < QR_PER_LANE; ++i)
        if (i * 32 + lane < len) buf[r[i]] = e[i];
    qr_wave_sync();
}

// grid ceil(N / 8): 8 half-waves per workgroup, one source row each.  No workgroup barrier: the half-waves are independent.
// The summation order is a fixed function of the row's SORTED column list, whatever order the fill kernel left:
//   L <= QR_CAP : one LDS sort, four interleaved partial sums;
//   longer      : chunks of QR_CAP sorted in LDS and written back, then a merge over the chunk heads (sequential sum); the head
//                 positions live in LDS, or for more than QR_CAP chunks in the row's own stretch of `scratch` (col_src, which
//                 nobody reads any more): a single row with > 262 144 references is slow (~1 us per reference) but correct.
__global__ __launch_bounds__(256) void qg_stack_bwd_rows_kernel(int N, int C, const int *__restrict__ offsets, int *list,
                                                                int *scratch, const float *__restrict__ g_t,
                                                                float *__restrict__ grad_features, int ld) {
    __shared__ int lds[8][QR_CAP];
    const int hw = threadIdx.x >> 5, lane = threadIdx.x & 31;
    const int row = blockIdx.x * 8 + hw;
    if (row >= N) return;
    const int start = offsets[row], L = offsets[row + 1] - start;
    if (L == 0) return;
    int *buf = lds[hw];
    int *lst = list + start;
    const bool c0 = lane < C, c1 = lane + 32 < C;
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};     // channels lane and lane + 32
    if (L <= QR_CAP) {
        for (int i = lane; i < L; i += 32) buf[i] = lst[i];
        qr_wave_sync();
        qr_rank_sort(buf, L, lane);
        int k = 0;
        for (; k + 4 <= L; k += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float *g = g_t + (size_t)buf[k + u] * C;
                if (c0) a0[u] += g[lane];
                if (c1) a1[u] += g[lane + 32];
            }
        }
        for (; k < L; ++k) {
            const float *g = g_t + (size_t)buf[k] * C;
            if (c0) a0[0] += g[lane];
            if (c1) a1[0] += g[lane + 32];
        }
    } else {
        const int nchunk = (L + QR_CAP - 1) / QR_CAP;
        for (int ch = 0; ch < nchunk; ++ch) {
            const int base = ch * QR_CAP, len = min(QR_CAP, L - base);
            for (int i = lane; i < len; i += 32) buf[i] = lst[base + i];
            qr_wave_sync();
            qr_rank_sort(buf, len, lane);
            for (int i = lane; i < len; i += 32) lst[base + i] = buf[i];
            qr_wave_sync();
        }
        __threadfence();
        int *heads = nchunk <= QR_CAP ? buf : scratch + start;              // head position of chunk ch: touched by lane ch % 32 only
        for (int ch = lane; ch < nchunk; ch += 32) heads[ch] = 0;
        int which = -1, cur = 0x7fffffff;
        auto lane_min = [&]() {
            cur = 0x7fffffff;
            which = -1;
            for (int ch = lane; ch < nchunk; ch += 32) {
                const int pos = heads[ch], len = min(QR_CAP, L - ch * QR_CAP);
                if (pos < len) {
                    const int v = lst[ch * QR_CAP + pos];
                    if (v < cur) {
                        cur = v;
                        which = ch;
                    }
                }
            }
        };
        lane_min();
        for (int k = 0; k < L; ++k) {
            int best = cur;
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) best = min(best, __shfl_xor(best, d, 32));
            if (cur == best && which >= 0) {
                heads[which] += 1;
                lane_min();
            }
            // compensated (Kahan) sum: a row this long would otherwise lose ~sqrt(L) ulps to the running total
            const float *g = g_t + (size_t)best * C;
            if (c0) {
                const float y = g[lane] - a0[1], t = a0[0] + y;
                a0[1] = (t - a0[0]) - y;
                a0[0] = t;
            }
            if (c1) {
                const float y = g[lane + 32] - a1[1], t = a1[0] + y;
                a1[1] = (t - a1[0]) - y;
                a1[0] = t;
            }
        }
        a0[1] = a1[1] = 0.f;
    }
    float *dst = grad_features + (size_t)row * ld;
    if (c0) dst[lane] = (a0[0] + a0[1]) + (a0[2] + a0[3]);
    if (c1) dst[lane + 32] = (a1[0] + a1[1]) + (a1[2] + a1[3]);
}

constexpr int QG_LDS_MAX_FLOATS = 36864;

}  // namespace mgar

using namespace mgar;

#define QG_API extern "C" __attribute__((visibility("default")))

template <typename T>
static int qg_batch_fwd(int b, int c, int n, int npoints, int nsample, const float *xyz, const float *new_xyz,
                        const T *features, const float *wx, const int *idx, T *rel_out, size_t rel_bstride,
                        T *y_out, size_t y_bstride, void *stream, const char *what) {
    MGAR_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0, "query_group (batch) fwd: negative size");
    MGAR_REQUIRE(b <= 65535, "query_group (batch) fwd: b > 65535");
    if ((long long)b * npoints * nsample == 0) return MGAR_OK;
    MGAR_REQUIRE(xyz && new_xyz && idx && (features || c == 0) && (y_out || c == 0) && (rel_out || y_out),
                 "query_group (batch) fwd: null pointer");
    dim3 grid(ceil_div((long long)npoints * nsample, 256), 1 + ceil_div(c, QG_CCHUNK), b);
    KtScope kt(KT_QUERY_GROUP_FWD, (hipStream_t)stream, (double)b * npoints * nsample * (4.0 + (double)sizeof(T) * (c + 3)));   // idx + the grouped tensor written once (source rows: cache-resident)
    hipLaunchKernelGGL(qg_batch_fwd_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, c, n, npoints, nsample, xyz, new_xyz,
                       features, wx, idx, rel_out, rel_bstride, y_out, y_bstride);
    return check_launch(what);
}

static int qg_batch_bwd(int b, int c, int n, int npoints, int nsample, const float *grad_y, size_t y_bstride, const int *idx,
                        float *grad_features, void *stream, const char *what) {
    MGAR_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0, "query_group (batch) bwd: negative size");
    MGAR_REQUIRE(b <= 65535 && c <= 65535, "query_group (batch) bwd: b or c > 65535");
    const int cols = npoints * nsample;
    if ((long long)b * c * cols == 0) return MGAR_OK;
    MGAR_REQUIRE(grad_y && idx && grad_features, "query_group (batch) bwd: null pointer");
    KtScope kt(KT_QUERY_GROUP_BWD, (hipStream_t)stream, (double)b * cols * (4.0 + 4.0 * c));
    if (n <= QG_LDS_MAX_FLOATS) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void *)qg_batch_bwd_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      QG_LDS_MAX_FLOATS * (int)sizeof(float));
            attr_set = true;
        }
        hipLaunchKernelGGL(qg_batch_bwd_lds_kernel, dim3(c, b), dim3(cols >= 4096 ? 1024 : 256), (size_t)n * sizeof(float),
                           (hipStream_t)stream, c, n, cols, grad_y, y_bstride, idx, grad_features);
    } else {
        hipLaunchKernelGGL(qg_batch_bwd_atomic_kernel, dim3(ceil_div(cols, 256), c, b), dim3(256), 0, (hipStream_t)stream, c, n,
                           cols, grad_y, y_bstride, idx, grad_features);
    }
    return check_launch(what);
}

QG_API int mgar_query_group_batch_fwd(int b, int c, int n, int npoints, int nsample, const float *xyz, const float *new_xyz,
                                      const float *features, const int *idx, float *out, void *stream) {
    const size_t cols = (size_t)npoints * nsample;
    return qg_batch_fwd<float>(b, c, n, npoints, nsample, xyz, new_xyz, features, nullptr, idx, out, (3 + c) * cols,
                               out ? out + 3 * cols : nullptr, (3 + c) * cols, stream, "query_group_batch_fwd: launch failed");
}

QG_API int mgar_query_group_batch_bwd(int b, int c, int n, int npoints, int nsample, const float *grad_out, const int *idx,
                                      float *grad_features, void *stream) {
    const size_t cols = (size_t)npoints * nsample;
    return qg_batch_bwd(b, c, n, npoints, nsample, grad_out ? grad_out + 3 * cols : nullptr, (3 + c) * cols, idx, grad_features,
                        stream, "query_group_batch_bwd: launch failed");
}

QG_API int mgar_query_group_proj_batch_fwd(int b, int c, int n, int npoints, int nsample, const float *xyz,
                                           const float *new_xyz, const float *zf, const float *wx, const int *idx,
                                           float *rel_out, float *y_out, void *stream) {
    MGAR_REQUIRE(wx && zf && y_out, "query_group_proj_batch_fwd: null pointer");
    const size_t cols = (size_t)npoints * nsample;
    return qg_batch_fwd<float>(b, c, n, npoints, nsample, xyz, new_xyz, zf, wx, idx, rel_out, 3 * cols, y_out, (size_t)c * cols,
                               stream, "query_group_proj_batch_fwd: launch failed");
}

QG_API int mgar_query_group_proj_batch_bwd(int b, int c, int n, int npoints, int nsample, const float *grad_y, const int *idx,
                                           float *grad_zf, void *stream) {
    return qg_batch_bwd(b, c, n, npoints, nsample, grad_y, (size_t)c * npoints * nsample, idx, grad_zf, stream,
                        "query_group_proj_batch_bwd: launch failed");
}

template <typename T>
static int qg_stack_fwd(int B, int M, int C, int nsample, const float *xyz, const int *xyz_batch_cnt, const float *new_xyz,
                        const int *new_xyz_batch_cnt, const T *features, int ld, const float *wx, const int *idx,
                        T *rel_out, T *y_out, void *stream, const char *what) {
    MGAR_REQUIRE(B >= 0 && M >= 0 && C >= 0 && nsample >= 0, "query_group (stack) fwd: negative size");
    const long long total = (long long)M * nsample;
    if (B == 0 || total == 0) return MGAR_OK;
    MGAR_REQUIRE(xyz && xyz_batch_cnt && new_xyz && new_xyz_batch_cnt && idx && (features || C == 0) && (y_out || C == 0) &&
                     (rel_out || y_out), "query_group (stack) fwd: null pointer");
    KtScope kt(KT_QUERY_GROUP_FWD, (hipStream_t)stream, (double)total * (4.0 + (double)sizeof(T) * (C + 3)));
    hipLaunchKernelGGL(qg_stack_fwd_kernel<T>, dim3(ceil_div(total, QS_COLS)), dim3(256), 0, (hipStream_t)stream, B, M, C, nsample,
                       xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, ld, wx, idx, rel_out, y_out);
    return check_launch(what);
}

static int qg_stack_bwd(int B, int M, int C, int nsample, const float *grad_y, const int *idx, const int *new_xyz_batch_cnt,
                        const int *xyz_batch_cnt, float *grad_features, int ld, void *stream, const char *what) {
    MGAR_REQUIRE(B >= 0 && M >= 0 && C >= 0 && nsample >= 0 && ld >= C, "query_group (stack) bwd: negative size or ld < C");
    const long long total = (long long)M * nsample;
    if (B == 0 || total == 0 || C == 0) return MGAR_OK;
    MGAR_REQUIRE(grad_y && idx && new_xyz_batch_cnt && xyz_batch_cnt && grad_features, "query_group (stack) bwd: null pointer");
    KtScope kt(KT_QUERY_GROUP_BWD, (hipStream_t)stream, (double)total * (4.0 + 4.0 * C));
    hipLaunchKernelGGL(qg_stack_bwd_kernel, dim3(ceil_div(total, QS_COLS)), dim3(256), 0, (hipStream_t)stream, B, M, C, nsample,
                       grad_y, idx, new_xyz_batch_cnt, xyz_batch_cnt, grad_features, ld);
    return check_launch(what);
}

QG_API int mgar_query_group_stack_fwd(int B, int M, int C, int nsample, const float *xyz, const int *xyz_batch_cnt,
                                      const float *new_xyz, const int *new_xyz_batch_cnt, const float *features, const int *idx,
                                      float *out, void *stream) {
    const size_t ms = (size_t)M * nsample;
    return qg_stack_fwd<float>(B, M, C, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, C, nullptr, idx, out,
                               out ? out + 3 * ms : nullptr, stream, "query_group_stack_fwd: launch failed");
}

QG_API int mgar_query_group_stack_bwd(int B, int M, int C, int nsample, const float *grad_out, const int *idx,
                                      const int *new_xyz_batch_cnt, const int *xyz_batch_cnt, float *grad_features, void *stream) {
    return qg_stack_bwd(B, M, C, nsample, grad_out ? grad_out + 3 * (size_t)M * nsample : nullptr, idx, new_xyz_batch_cnt,
                        xyz_batch_cnt, grad_features, C, stream, "query_group_stack_bwd: launch failed");
}

QG_API int mgar_query_group_proj_stack_fwd(int B, int M, int C, int nsample, const float *xyz, const int *xyz_batch_cnt,
                                           const float *new_xyz, const int *new_xyz_batch_cnt, const float *zf, int zf_ld,
                                           const float *wx, const int *idx, float *rel_out, float *y_out, void *stream) {
    MGAR_REQUIRE(wx && zf && y_out && zf_ld >= C, "query_group_proj_stack_fwd: null pointer or zf_ld < C");
    return qg_stack_fwd<float>(B, M, C, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, zf, zf_ld, wx, idx, rel_out, y_out,
                               stream, "query_group_proj_stack_fwd: launch failed");
}

QG_API int mgar_query_group_proj_stack_bwd(int B, int M, int C, int nsample, const float *grad_y, const int *idx,
                                           const int *new_xyz_batch_cnt, const int *xyz_batch_cnt, float *grad_zf, int zf_ld,
                                           void *stream) {
    return qg_stack_bwd(B, M, C, nsample, grad_y, idx, new_xyz_batch_cnt, xyz_batch_cnt, grad_zf, zf_ld, stream,
                        "query_group_proj_stack_bwd: launch failed");
}

// ---- bf16 payload twins of the forward entry points: features / zf / out / rel_out / y_out address bf16 elements; xyz, new_xyz,
// wx and every index stay fp32 / int32 (index parity must not depend on the payload type) ----
QG_API int mgar_query_group_batch_fwd_bf16(int b, int c, int n, int npoints, int nsample, const float *xyz, const float *new_xyz,
                                           const void *features, const int *idx, void *out, void *stream) {
    const size_t cols = (size_t)npoints * nsample;
    bf16_t *o = (bf16_t *)out;
    return qg_batch_fwd<bf16_t>(b, c, n, npoints, nsample, xyz, new_xyz, (const bf16_t *)features, nullptr, idx, o, (3 + c) * cols,
                                o ? o + 3 * cols : nullptr, (3 + c) * cols, stream, "query_group_batch_fwd_bf16: launch failed");
}
QG_API int mgar_query_group_proj_batch_fwd_bf16(int b, int c, int n, int npoints, int nsample, const float *xyz,
                                                const float *new_xyz, const void *zf, const float *wx, const int *idx,
                                                void *rel_out, void *y_out, void *stream) {
    MGAR_REQUIRE(wx && zf && y_out, "query_group_proj_batch_fwd_bf16: null pointer");
    const size_t cols = (size_t)npoints * nsample;
    return qg_batch_fwd<bf16_t>(b, c, n, npoints, nsample, xyz, new_xyz, (const bf16_t *)zf, wx, idx, (bf16_t *)rel_out, 3 * cols,
                                (bf16_t *)y_out, (size_t)c * cols, stream, "query_group_proj_batch_fwd_bf16: launch failed");
}
QG_API int mgar_query_group_stack_fwd_bf16(int B, int M, int C, int nsample, const float *xyz, const int *xyz_batch_cnt,
                                           const float *new_xyz, const int *new_xyz_batch_cnt, const void *features,
                                           const int *idx, void *out, void *stream) {
    const size_t ms = (size_t)M * nsample;
    bf16_t *o = (bf16_t *)out;
    return qg_stack_fwd<bf16_t>(B, M, C, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, (const bf16_t *)features, C, nullptr,
                                idx, o, o ? o + 3 * ms : nullptr, stream, "query_group_stack_fwd_bf16: launch failed");
}
QG_API int mgar_query_group_proj_stack_fwd_bf16(int B, int M, int C, int nsample, const float *xyz, const int *xyz_batch_cnt,
                                                const float *new_xyz, const int *new_xyz_batch_cnt, const void *zf, int zf_ld,
                                                const float *wx, const int *idx, void *rel_out, void *y_out, void *stream) {
    MGAR_REQUIRE(wx && zf && y_out && zf_ld >= C, "query_group_proj_stack_fwd_bf16: null pointer or zf_ld < C");
    return qg_stack_fwd<bf16_t>(B, M, C, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, (const bf16_t *)zf, zf_ld, wx, idx,
                                (bf16_t *)rel_out, (bf16_t *)y_out, stream, "query_group_proj_stack_fwd_bf16: launch failed");
}

// ---- deterministic stack backward (no float atomics): see qg_stack_bwd_rows_kernel ---------------------------------------------
// Step 1: col_src (M*nsample) int32 = global source row of every column (-1: empty ball), counts (N) += references
// (counts zero-filled by the caller).  Step 3: list (sum counts) from offsets = exclusive scan of counts (N + 1 entries) and a
// zero-filled cursor (N).  Step 4: grad_features (N, ld) rows with references are written, the others left alone; g_t is the
// ROW-MAJOR gradient (M*nsample, C), C <= 64; `list` may come back reordered (long rows are sorted in place) and `scratch`
// (M*nsample ints: pass col_src, which is dead by then) overwritten.
QG_API int mgar_query_group_stack_inverse_count(int B, int M, int nsample, const int *idx, const int *new_xyz_batch_cnt,
                                                const int *xyz_batch_cnt, int *col_src, int *counts, void *stream) {
    MGAR_REQUIRE(B >= 0 && M >= 0 && nsample >= 0, "query_group_stack_inverse_count: negative size");
    const long long total = (long long)M * nsample;
    if (B == 0 || total == 0) return MGAR_OK;
    MGAR_REQUIRE(idx && new_xyz_batch_cnt && xyz_batch_cnt && col_src && counts, "query_group_stack_inverse_count: null pointer");
    hipLaunchKernelGGL(qg_inv_count_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, B, M, nsample, idx,
                       new_xyz_batch_cnt, xyz_batch_cnt, col_src, counts);
    return check_launch("query_group_stack_inverse_count: launch failed");
}
QG_API int mgar_query_group_stack_inverse_fill(long long total, const int *col_src, const int *offsets, int *cursor, int *list,
                                               void *stream) {
    MGAR_REQUIRE(total >= 0, "query_group_stack_inverse_fill: negative size");
    if (total == 0) return MGAR_OK;
    MGAR_REQUIRE(col_src && offsets && cursor && list, "query_group_stack_inverse_fill: null pointer");
    hipLaunchKernelGGL(qg_inv_fill_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, total, col_src, offsets, cursor,
                       list);
    return check_launch("query_group_stack_inverse_fill: launch failed");
}
QG_API int mgar_query_group_stack_bwd_rows(int N, int C, long long total, const int *offsets, int *list, int *scratch,
                                           const float *g_t, float *grad_features, int ld, void *stream) {
    MGAR_REQUIRE(N >= 0 && C >= 1 && ld >= C && total >= 0, "query_group_stack_bwd_rows: bad sizes");
    if (C > 64) {
        set_error("query_group_stack_bwd_rows: C <= 64");
        return MGAR_EUNSUPPORTED;
    }
    if (N == 0 || total == 0) return MGAR_OK;
    MGAR_REQUIRE(offsets && list && scratch && g_t && grad_features, "query_group_stack_bwd_rows: null pointer");
    KtScope kt(KT_QUERY_GROUP_BWD, (hipStream_t)stream, (double)total * (4.0 + 4.0 * C));
    hipLaunchKernelGGL(qg_stack_bwd_rows_kernel, dim3(ceil_div(N, 8)), dim3(256), 0, (hipStream_t)stream, N, C, offsets, list, scratch,
                       g_t, grad_features, ld);
    return check_launch("query_group_stack_bwd_rows: launch failed");
}
