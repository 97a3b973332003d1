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
    // 16-byte loads, two of them in flight per operand (round 3): the scalar version waited for one 4-byte index and one 4-byte
    // gradient per LDS atomic and ran at the pace of the memory latency (0.7 ms per launch at level 1 of config c3)
    if ((cols & 3) == 0 && ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(id)) & 15) == 0) {
        const int q = cols >> 2;
        const float4 *g4 = reinterpret_cast<const float4 *>(g);
        const int4 *i4 = reinterpret_cast<const int4 *>(id);
        int e = threadIdx.x;
        for (; e + (int)blockDim.x < q; e += 2 * blockDim.x) {
            const int4 ia = i4[e], ib = i4[e + blockDim.x];
            const float4 ga = g4[e], gb = g4[e + blockDim.x];
            atomicAdd(&row[ia.x], ga.x); atomicAdd(&row[ia.y], ga.y); atomicAdd(&row[ia.z], ga.z); atomicAdd(&row[ia.w], ga.w);
            atomicAdd(&row[ib.x], gb.x); atomicAdd(&row[ib.y], gb.y); atomicAdd(&row[ib.z], gb.z); atomicAdd(&row[ib.w], gb.w);
        }
        if (e < q) {
            const int4 ia = i4[e];
            const float4 ga = g4[e];
            atomicAdd(&row[ia.x], ga.x); atomicAdd(&row[ia.y], ga.y); atomicAdd(&row[ia.z], ga.z); atomicAdd(&row[ia.w], ga.w);
        }
    } else {
        for (int e = threadIdx.x; e < cols; e += blockDim.x) atomicAdd(&row[id[e]], g[e]);
    }
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
                                                           T *__restrict__ rel_out, T *__restrict__ y_out,
                                                           float *__restrict__ stats) {
    __shared__ QsTile t;
    __shared__ float red[8][QS_CH];
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
        if (stats) {
            // BatchNorm statistics partials of y for the layer that follows: per channel the (mean, M2) of this tile's 128
            // columns (total % 128 == 0), exact two-pass over the LDS tile; thread = (channel ci, segment of 16 columns).
            // Layout: stats[(channel * ntiles + tile) * 2 + {0, 1}] = the chunk format of bn_finalize_kernel, chunk = 128.
            const int ci = threadIdx.x & (QS_CH - 1), seg = threadIdx.x / QS_CH;
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < QS_COLS / 8; ++j) sum += t.tile[seg * (QS_COLS / 8) + j][ci];
            red[seg][ci] = sum;
            __syncthreads();
            float tot = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) tot += red[k][ci];
            const float mean = tot * (1.f / QS_COLS);
            float m2 = 0.f;
#pragma unroll
            for (int j = 0; j < QS_COLS / 8; ++j) {
                const float d = t.tile[seg * (QS_COLS / 8) + j][ci] - mean;
                m2 += d * d;
            }
            __syncthreads();
            red[seg][ci] = m2;
            __syncthreads();
            if (seg == 0 && ci < nch) {
                float q = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) q += red[k][ci];
                *reinterpret_cast<float2 *>(stats + ((size_t)(c0 + ci) * gridDim.x + blockIdx.x) * 2) = make_float2(mean, q);
            }
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

// ---- stack backward without atomics: inverted index (stable counting sort) + owner-computes -----------------------------------
// The atomic scatter above runs at the chip's device-scope atomic REQUEST rate (~8 G requests/s: 1.7 ms per scale of the
// RoI-grid lift at config c3, 1.3 TB/s of added bytes) and its sums depend on the arrival order.  An inverted index built with
// global integer atomics would pay that same request rate twice.  Here the index is a STABLE COUNTING SORT of the
// (source row, column) pairs with its histograms in LDS, one workgroup per WINDOW of QW_ROWS source rows of one sample:
//   qg_inv_window_count_kernel  how many columns of a sample fall into each of its windows (where a window's lists start);
//   qg_inv_window_sort_kernel   the workgroup of a window reads all columns of its sample; each of its 16 waves owns a
//                        contiguous run of them and counts the window's rows it meets in its own LDS histogram; an exclusive
//                        scan over the waves and the rows turns the histograms into cursors; every wave walks its run again
//                        and places its columns, lanes with the same row taking consecutive slots in lane order.  "Run order,
//                        then column order inside the run" IS ascending column order: list = the columns of every source
//                        row, ascending.  It also emits WORK ITEMS (row, begin, length <= QR_PART, parts of the row): a point
//                        inside many balls is shared between several workers along fixed cuts;
//   qg_stack_bwd_rows_*  a half-wave per work item adds the listed rows of the ROW-MAJOR gradient g_t (M*nsample, C) -- one
//                        contiguous 4*C-byte read per entry -- in a fixed order and stores its row once (one-part rows
//                        straight into the feature gradient, parts of longer rows into a side buffer); it also forms its
//                        share of d wx = sum_col g_t[col] (x) (xyz[row] - new_xyz[query]), so the relative coordinates
//                        need neither be stored by the forward nor streamed again by a weight-gradient pass;
//   qg_stack_bwd_combine_kernel  adds the parts of the long rows in part order.
// Every sum has a fixed order: the result is bit-reproducible.  Rows nobody references are not written (caller zero-fills).
constexpr int QW_THREADS = 1024, QW_WAVES = QW_THREADS / 64;
constexpr int QW_ROWS = 1024;          // source rows per window: QW_WAVES histograms of QW_ROWS ints = 64 KB LDS
constexpr int QW_LOADS = 16;           // 64-column steps whose index loads are in flight together
constexpr int QR_PART = 256;           // list entries per work item

struct QiScan {
    int wave_tot[QW_WAVES];
    int total;
};

// exclusive scan of v over the QW_THREADS threads of the workgroup (thread order), total to all; barriers inside
__device__ __forceinline__ int qi_block_excl_scan(int v, QiScan &sc, int &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) sc.wave_tot[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int w = 0; w < QW_WAVES; ++w) {
            const int t = sc.wave_tot[w];
            sc.wave_tot[w] = run;
            run += t;
        }
        sc.total = run;
    }
    __syncthreads();
    total = sc.total;
    const int r = inc - v + sc.wave_tot[wave];
    __syncthreads();
    return r;
}

__device__ __forceinline__ void qi_wave_sync() {            // LDS hand-over between the lanes of one wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Row (window-local) of sample-local column c; -1: empty ball / another window / no such column.  All 64 lanes call it with
// consecutive c (c - lane a multiple of 64): when nsample divides 64 the first slot of the lane's query sits in an earlier
// lane of the same step and comes by shuffle instead of a second load.
__device__ __forceinline__ int qi_key(const int *__restrict__ idx, int col0, int c, int end, int nsample, bool ns_div64, int win0,
                                      int nb) {
    const int lane = threadIdx.x & 63;
    const int v = c < end ? idx[col0 + c] : -1;
    const int first = ns_div64 ? __shfl(v, lane - lane % nsample, 64) : (c < end ? idx[col0 + c - c % nsample] : -1);
    const int k = v - win0;
    return (c < end && first >= 0 && k >= 0 && k < nb) ? k : -1;
}

// ---- level 1: split a sample's columns by WINDOW (stable), so that the workgroup of a window reads its own columns only ----
constexpr int QP_GROUPS = 8, QP_WAVES = 4, QP_RUNS = QP_GROUPS * QP_WAVES;   // column runs per sample (one wave each)
constexpr int QP_WINDOWS = 256;        // windows per sample the split supports (262 144 source rows per sample)

struct QpRun {
    int col0, beg, end, ord0, nwin, np, run;
};
// the run of this wave: sample blockIdx.y, run blockIdx.x * QP_WAVES + wave; 256 threads
__device__ __forceinline__ QpRun qp_run(int nsample, const int *__restrict__ q_cnt, const int *__restrict__ p_cnt) {
    __shared__ int s_q0, s_ord0;
    const int b = blockIdx.y;
    if (threadIdx.x == 0) s_q0 = s_ord0 = 0;
    __syncthreads();
    int q = 0, w = 0;
    for (int i = threadIdx.x; i < b; i += 256) {
        q += q_cnt[i];
        w += (p_cnt[i] + QW_ROWS - 1) / QW_ROWS;
    }
    if (q) atomicAdd(&s_q0, q);
    if (w) atomicAdd(&s_ord0, w);
    __syncthreads();
    QpRun r;
    r.col0 = s_q0 * nsample;
    r.ord0 = s_ord0;
    r.np = p_cnt[b];
    r.nwin = (r.np + QW_ROWS - 1) / QW_ROWS;
    r.run = blockIdx.x * QP_WAVES + (threadIdx.x >> 6);
    const int ncols = q_cnt[b] * nsample;
    const int per = ((ncols + QP_RUNS - 1) / QP_RUNS + 63) & ~63;
    r.beg = min(r.run * per, ncols);
    r.end = min(r.beg + per, ncols);
    return r;
}

// rank of the lane among the lanes with the same key (key < 0: not taking part; keys < 2^BITS), size of that group, first lane
// of it.  Few lanes: one register-only round per distinct key; many: the set of equal lanes bit by bit (BITS ballots).
template <int BITS>
__device__ __forceinline__ void qi_rank(int key, int lane, int &rank, int &group, bool &leader) {
    const unsigned long long valid = __ballot(key >= 0);
    rank = group = 0;
    leader = false;
    if (__popcll(valid) <= 8) {
        unsigned long long rem = valid;
        while (rem) {
            const int first = __ffsll((long long)rem) - 1;
            const int v = __builtin_amdgcn_readlane(key, first);
            const unsigned long long same = __ballot(key == v);
            if (key == v) {
                rank = __popcll(same & ((1ull << lane) - 1ull));
                group = __popcll(same);
                leader = lane == first;
            }
            rem &= ~same;
        }
        return;
    }
    unsigned long long same = valid;
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
        const bool bit = (key >> b) & 1;
        const unsigned long long bal = __ballot(key >= 0 && bit);
        same &= bit ? bal : ~bal;
    }
    if (key >= 0) {
        rank = __popcll(same & ((1ull << lane) - 1ull));
        group = __popcll(same);
        leader = rank == 0;
    }
}

// grid (QP_GROUPS, B), 256 threads: rc[(window ordinal) * QP_RUNS + run] = columns of the run whose row lies in that window
__global__ __launch_bounds__(256) void qg_inv_split_count_kernel(int nsample, const int *__restrict__ idx, const int *__restrict__ q_cnt,
                                                                 const int *__restrict__ p_cnt, int *__restrict__ rc) {
    __shared__ int cnt[QP_WAVES][QP_WINDOWS];
    const QpRun r = qp_run(nsample, q_cnt, p_cnt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = lane; i < QP_WINDOWS; i += 64) cnt[wave][i] = 0;
    __syncthreads();
    const bool ns_div64 = 64 % nsample == 0;
    for (int c0 = r.beg; c0 < r.end; c0 += 64 * QW_LOADS) {
        int k[QW_LOADS];
#pragma unroll
        for (int u = 0; u < QW_LOADS; ++u) k[u] = qi_key(idx, r.col0, c0 + u * 64 + lane, r.end, nsample, ns_div64, 0, r.np);
#pragma unroll
        for (int u = 0; u < QW_LOADS; ++u)
            if (k[u] >= 0) atomicAdd(&cnt[wave][k[u] / QW_ROWS], 1);     // integer LDS atomics: order-free
    }
    __syncthreads();
    for (int i = lane; i < r.nwin; i += 64) rc[(size_t)(r.ord0 + i) * QP_RUNS + r.run] = cnt[wave][i];
}

// grid (B), 256 threads: thread t = window t of the sample: rc -> exclusive over the runs; wbase = first list slot of the
// window; wtab[window ordinal] = (first source row, rows, first list slot, columns) for the workgroup that sorts the window;
// head[0] = number of windows, head[1] = 0 (the counter of the rows cut into several work items)
__global__ __launch_bounds__(256) void qg_inv_split_scan_kernel(int B, int nsample, const int *__restrict__ q_cnt,
                                                                const int *__restrict__ p_cnt, int *__restrict__ rc,
                                                                int *__restrict__ wbase, int4 *__restrict__ wtab,
                                                                int *__restrict__ head) {
    __shared__ int s_q0, s_p0, s_ord0, wave_tot[4];
    const int b = blockIdx.x;
    if (threadIdx.x == 0) s_q0 = s_p0 = s_ord0 = 0;
    __syncthreads();
    int q = 0, pp = 0, w = 0;
    for (int i = threadIdx.x; i < b; i += 256) {
        q += q_cnt[i];
        pp += p_cnt[i];
        w += (p_cnt[i] + QW_ROWS - 1) / QW_ROWS;
    }
    if (q) atomicAdd(&s_q0, q);
    if (pp) atomicAdd(&s_p0, pp);
    if (w) atomicAdd(&s_ord0, w);
    __syncthreads();
    const int np = p_cnt[b], nwin = (np + QW_ROWS - 1) / QW_ROWS, t = threadIdx.x;
    if (t == 0 && b == B - 1) {
        head[0] = s_ord0 + nwin;
        head[1] = 0;
    }
    int total = 0;
    if (t < nwin) {
        int *row = rc + (size_t)(s_ord0 + t) * QP_RUNS;
        int v[QP_RUNS];
#pragma unroll
        for (int i = 0; i < QP_RUNS; ++i) v[i] = row[i];
#pragma unroll
        for (int i = 0; i < QP_RUNS; ++i) {
            row[i] = total;
            total += v[i];
        }
    }
    const int lane = t & 63, wave = t >> 6;
    int inc = total;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int u = __shfl_up(inc, d, 64);
        if (lane >= d) inc += u;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    int before = 0;
    for (int i = 0; i < wave; ++i) before += wave_tot[i];
    if (t < nwin) {
        const int base = s_q0 * nsample + before + inc - total;
        wbase[s_ord0 + t] = base;
        wtab[s_ord0 + t] = make_int4(s_p0 + t * QW_ROWS, min(QW_ROWS, np - t * QW_ROWS), base, total);
    }
}

// grid (QP_GROUPS, B), 256 threads: the stable split: wcol / wkey[slot] = column / row inside its window, window by window
__global__ __launch_bounds__(256) void qg_inv_split_fill_kernel(int nsample, const int *__restrict__ idx, const int *__restrict__ q_cnt,
                                                                const int *__restrict__ p_cnt, const int *__restrict__ rc,
                                                                const int *__restrict__ wbase, int *__restrict__ wcol,
                                                                int *__restrict__ wkey) {
    __shared__ int cursor[QP_WAVES][QP_WINDOWS];
    const QpRun r = qp_run(nsample, q_cnt, p_cnt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int *my = cursor[wave];
    for (int i = lane; i < r.nwin; i += 64) my[i] = wbase[r.ord0 + i] + rc[(size_t)(r.ord0 + i) * QP_RUNS + r.run];
    __syncthreads();
    const bool ns_div64 = 64 % nsample == 0;
    for (int c0 = r.beg; c0 < r.end; c0 += 64 * QW_LOADS) {
        int k[QW_LOADS];
#pragma unroll
        for (int u = 0; u < QW_LOADS; ++u) k[u] = qi_key(idx, r.col0, c0 + u * 64 + lane, r.end, nsample, ns_div64, 0, r.np);
#pragma unroll
        for (int u = 0; u < QW_LOADS; ++u) {
            const int win = k[u] >= 0 ? k[u] / QW_ROWS : -1;
            int rank, group;
            bool leader;
            qi_rank<8>(win, lane, rank, group, leader);
            if (win >= 0) {                                              // in-order LDS: all read the cursor, then the leaders advance it
                const int pos = my[win];
                wcol[pos + rank] = r.col0 + c0 + u * 64 + lane;
                wkey[pos + rank] = k[u] - win * QW_ROWS;
                if (leader) my[win] = pos + group;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- level 2: grid (upper bound of the number of windows: ceil(N / QW_ROWS) + B), QW_THREADS threads -------------------------------
// The workgroup of a window sorts the window's columns (wcol, rows wkey) by row, stably.  multi_rows: the rows cut into several
// work items (head[1] of them; at most total / QR_PART).
// items: int4 (row, begin, length, parts of the row), zero-filled by the caller; the items of a window start at slot
// (first row of the window) + floor(first list slot of the window / QR_PART) + (ordinal of the window): an upper bound of
// what the windows before it can need.  row_item[row] = first item of the row, -1 if nobody references it.
__global__ __launch_bounds__(QW_THREADS) void qg_inv_window_sort_kernel(const int4 *__restrict__ wtab, int *__restrict__ head,
                                                                        const int *__restrict__ wcol, const int *__restrict__ wkey,
                                                                        int *__restrict__ list, int4 *__restrict__ items,
                                                                        int *__restrict__ row_item, int *__restrict__ multi_rows) {
    __shared__ int hist[QW_WAVES][QW_ROWS];
    __shared__ QiScan sc;
    if ((int)blockIdx.x >= head[0]) return;                              // workgroup-uniform
    const int4 wt = wtab[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = wt.x, nb = wt.y, base = wt.z, ncols = wt.w;
    for (int i = threadIdx.x; i < QW_WAVES * QW_ROWS; i += QW_THREADS) (&hist[0][0])[i] = 0;
    __syncthreads();
    const int per = ((ncols + QW_WAVES - 1) / QW_WAVES + 63) & ~63;      // columns per wave, whole 64-column steps
    const int beg = min(wave * per, ncols), end = min(beg + per, ncols);
    int *my = hist[wave];
    // 1. histogram of this wave's run (integer LDS atomics: the counts are order-free)
    for (int c0 = beg; c0 < end; c0 += 64 * QW_LOADS) {
        int k[QW_LOADS];
#pragma unroll
        for (int u = 0; u < QW_LOADS; ++u) {
            const int c = c0 + u * 64 + lane;
            k[u] = c < end ? wkey[base + c] : -1;
        }
#pragma unroll
        for (int u = 0; u < QW_LOADS; ++u)
            if (k[u] >= 0) atomicAdd(my + k[u], 1);
    }
    __syncthreads();
    // 2. per row: exclusive scan over the waves; then over the rows (thread t owns row t); cursors; work items
    const int k = threadIdx.x;
    int cnt = 0;
    if (k < nb) {
#pragma unroll
        for (int w = 0; w < QW_WAVES; ++w) {
            const int t = hist[w][k];
            hist[w][k] = cnt;
            cnt += t;
        }
    }
    int total, itotal;
    const int off = qi_block_excl_scan(cnt, sc, total);
    const int parts = (cnt + QR_PART - 1) / QR_PART;
    const int ioff = qi_block_excl_scan(parts, sc, itotal);
    if (k < nb) {
        const int rbeg = base + off, row = row0 + k;
        const int first = row0 + base / QR_PART + (int)blockIdx.x + ioff;
#pragma unroll
        for (int w = 0; w < QW_WAVES; ++w) hist[w][k] += rbeg;
        row_item[row] = parts ? first : -1;
        for (int q = 0; q < parts; ++q) items[first + q] = make_int4(row, rbeg + q * QR_PART, min(QR_PART, cnt - q * QR_PART), parts);
        if (parts > 1) multi_rows[atomicAdd(head + 1, 1)] = row;          // a set: the order of this list does not matter
    }
    __syncthreads();
    // 3. stable fill
    for (int c0 = beg; c0 < end; c0 += 64 * QW_LOADS) {
        int kk[QW_LOADS], cc[QW_LOADS];
#pragma unroll
        for (int u = 0; u < QW_LOADS; ++u) {
            const int c = c0 + u * 64 + lane;
            kk[u] = c < end ? wkey[base + c] : -1;
            cc[u] = c < end ? wcol[base + c] : 0;
        }
#pragma unroll
        for (int u = 0; u < QW_LOADS; ++u) {
            int rank, group;
            bool leader;
            qi_rank<10>(kk[u], lane, rank, group, leader);
            if (kk[u] >= 0) {                                            // in-order LDS: all read the cursor, then the leaders advance it
                const int pos = my[kk[u]];
                list[pos + rank] = cc[u];
                if (leader) my[kk[u]] = pos + group;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// What a half-wave needs of one round of <= 32 list entries: its column and the query centre relative to the source row
struct QrRound {
    int col;
    float rx, ry, rz;
};
__device__ __forceinline__ QrRound qr_load_round(const int *__restrict__ list, int begin, int ne, int lane, bool want_rel,
                                                 const float *__restrict__ new_xyz, int nsample, float px, float py, float pz) {
    QrRound r = {0, 0.f, 0.f, 0.f};
    if (lane < ne) {
        r.col = list[begin + lane];
        if (want_rel) {
            const float *q = new_xyz + (size_t)(r.col / nsample) * 3;
            r.rx = px - q[0];
            r.ry = py - q[1];
            r.rz = pz - q[2];
        }
    }
    return r;
}

// grid ceil(n_items / 8), 256 threads: a half-wave per work item.  C % 4 == 0: 8 lanes read one row as float4 (channels 4q..4q+3
// and, for C > 32, 32+4q..), so one instruction of the half-wave fetches 4 rows and a round of 32 entries is 8 independent
// loads per lane.  Entry e of a round belongs to lane group e & 3; the groups are added as (g0 + g1) + (g2 + g3).
template <bool WIDE>                   // WIDE: C > 32, a second float4 per lane
__global__ __launch_bounds__(256) void qg_stack_bwd_rows_vec_kernel(int n_items, int C, int nsample, const int4 *__restrict__ items,
                                                                    const int *__restrict__ list, const float *__restrict__ g_t,
                                                                    const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                                    float *__restrict__ grad_features, int ld,
                                                                    float *__restrict__ part_rows, float *__restrict__ wx_part) {
    __shared__ float wsum[8][64][3];
    const int hw = threadIdx.x >> 5, lane = threadIdx.x & 31, grp = lane >> 3, q = lane & 7;
    const int item = blockIdx.x * 8 + hw;
    const int4 it = item < n_items ? items[item] : make_int4(0, 0, 0, 0);
    const int row = it.x, begin = it.y, len = it.z, parts = it.w;
    const bool h0 = 4 * q < C, h1 = WIDE && 32 + 4 * q < C;              // which of its two float4 exist
    float4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    float w0[4][3], w1[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int d = 0; d < 3; ++d) w0[i][d] = w1[i][d] = 0.f;
    float px = 0.f, py = 0.f, pz = 0.f;
    const bool want_rel = wx_part != nullptr;
    if (len > 0 && want_rel) {
        px = xyz[(size_t)row * 3];
        py = xyz[(size_t)row * 3 + 1];
        pz = xyz[(size_t)row * 3 + 2];
    }
    QrRound nxt = qr_load_round(list, begin, min(32, len), lane, want_rel, new_xyz, nsample, px, py, pz);
    for (int e0 = 0; e0 < len; e0 += 32) {
        const int ne = min(32, len - e0);
        const QrRound cur = nxt;
        if (e0 + 32 < len) nxt = qr_load_round(list, begin + e0 + 32, min(32, len - e0 - 32), lane, want_rel, new_xyz, nsample, px, py, pz);
#pragma unroll
        for (int uh = 0; uh < 8; uh += 4) {                                  // 4 row loads per lane (16 rows per half-wave) in flight
            if (uh * 4 >= ne) break;
            float4 v0[4], v1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = (uh + u) * 4 + grp;
                const int col = __shfl(cur.col, e, 32);
                const float4 *g = reinterpret_cast<const float4 *>(g_t + (size_t)col * C);
                v0[u] = (e < ne && h0) ? g[q] : make_float4(0.f, 0.f, 0.f, 0.f);
                if (WIDE) v1[u] = (e < ne && h1) ? g[8 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0.x += v0[u].x, a0.y += v0[u].y, a0.z += v0[u].z, a0.w += v0[u].w;
                if (WIDE) a1.x += v1[u].x, a1.y += v1[u].y, a1.z += v1[u].z, a1.w += v1[u].w;
                if (want_rel) {
                    const int e = (uh + u) * 4 + grp;
                    const float r[3] = {__shfl(cur.rx, e, 32), __shfl(cur.ry, e, 32), __shfl(cur.rz, e, 32)};   // zero past ne
                    const float g0[4] = {v0[u].x, v0[u].y, v0[u].z, v0[u].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int d = 0; d < 3; ++d) w0[i][d] += g0[i] * r[d];
                    if (WIDE) {
                        const float g1[4] = {v1[u].x, v1[u].y, v1[u].z, v1[u].w};
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int d = 0; d < 3; ++d) w1[i][d] += g1[i] * r[d];
                    }
                }
            }
        }
    }
    // (g0 + g1) + (g2 + g3): lane groups 8 and 16 apart
    auto fold = [&](float v) {
        v += __shfl_xor(v, 8, 32);
        v += __shfl_xor(v, 16, 32);
        return v;
    };
    a0.x = fold(a0.x), a0.y = fold(a0.y), a0.z = fold(a0.z), a0.w = fold(a0.w);
    if (WIDE) a1.x = fold(a1.x), a1.y = fold(a1.y), a1.z = fold(a1.z), a1.w = fold(a1.w);
    if (len > 0 && grp == 0) {
        float *dst = parts == 1 ? grad_features + (size_t)row * ld : part_rows + (size_t)item * C;
        if (h0) *reinterpret_cast<float4 *>(dst + 4 * q) = a0;
        if (h1) *reinterpret_cast<float4 *>(dst + 32 + 4 * q) = a1;
    }
    if (!want_rel) return;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float s0 = fold(w0[i][d]), s1 = WIDE ? fold(w1[i][d]) : 0.f;
            if (grp == 0) {
                wsum[hw][4 * q + i][d] = s0;
                if (WIDE) wsum[hw][32 + 4 * q + i][d] = s1;
            }
        }
    __syncthreads();
    for (int e = threadIdx.x; e < C * 3; e += 256) {         // the workgroup's share of d wx (C, 3), half-waves in order
        const int c = e / 3, d = e - c * 3;
        float t = 0.f;
#pragma unroll
        for (int h = 0; h < 8; ++h) t += wsum[h][c][d];
        wx_part[(size_t)blockIdx.x * C * 3 + e] = t;
    }
}

// Short lists (a few entries per row: many source rows, e.g. every point of the cloud): 8 lanes per WORK ITEM, 32 items per
// workgroup; a lane group reads its rows as float4 (C % 4 == 0) one after the other, 4 loads in flight, plain sequential sum.
template <bool WIDE>
__global__ __launch_bounds__(256) void qg_stack_bwd_rows_grp_kernel(int n_items, int C, int nsample, const int4 *__restrict__ items,
                                                                    const int *__restrict__ list, const float *__restrict__ g_t,
                                                                    const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                                    float *__restrict__ grad_features, int ld,
                                                                    float *__restrict__ part_rows, float *__restrict__ wx_part) {
    __shared__ float wsum[4][64][3];
    const int grp = threadIdx.x >> 3, q = threadIdx.x & 7, wave = threadIdx.x >> 6;
    const int item = blockIdx.x * 32 + grp;
    const int4 it = item < n_items ? items[item] : make_int4(0, 0, 0, 0);
    const int row = it.x, begin = it.y, len = it.z, parts = it.w;
    const bool h0 = 4 * q < C, h1 = WIDE && 32 + 4 * q < C;
    float4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    float w0[4][3], w1[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int d = 0; d < 3; ++d) w0[i][d] = w1[i][d] = 0.f;
    const bool want_rel = wx_part != nullptr;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (len > 0 && want_rel) {
        px = xyz[(size_t)row * 3];
        py = xyz[(size_t)row * 3 + 1];
        pz = xyz[(size_t)row * 3 + 2];
    }
    for (int e0 = 0; e0 < len; e0 += 8) {                     // group-uniform trip count
        const int ne = min(8, len - e0);
        int col_l = 0;
        float rx_l = 0.f, ry_l = 0.f, rz_l = 0.f;
        if (q < ne) {
            col_l = list[begin + e0 + q];
            if (want_rel) {
                const float *c = new_xyz + (size_t)(col_l / nsample) * 3;
                rx_l = px - c[0];
                ry_l = py - c[1];
                rz_l = pz - c[2];
            }
        }
#pragma unroll
        for (int uh = 0; uh < 8; uh += 4) {
            if (uh >= ne) break;
            float4 v0[4], v1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = uh + u;
                const int col = __shfl(col_l, e, 8);
                const float4 *g = reinterpret_cast<const float4 *>(g_t + (size_t)col * C);
                v0[u] = (e < ne && h0) ? g[q] : make_float4(0.f, 0.f, 0.f, 0.f);
                if (WIDE) v1[u] = (e < ne && h1) ? g[8 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0.x += v0[u].x, a0.y += v0[u].y, a0.z += v0[u].z, a0.w += v0[u].w;
                if (WIDE) a1.x += v1[u].x, a1.y += v1[u].y, a1.z += v1[u].z, a1.w += v1[u].w;
                if (want_rel) {
                    const int e = uh + u;
                    const float r[3] = {__shfl(rx_l, e, 8), __shfl(ry_l, e, 8), __shfl(rz_l, e, 8)};           // zero past ne
                    const float g0[4] = {v0[u].x, v0[u].y, v0[u].z, v0[u].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int d = 0; d < 3; ++d) w0[i][d] += g0[i] * r[d];
                    if (WIDE) {
                        const float g1[4] = {v1[u].x, v1[u].y, v1[u].z, v1[u].w};
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int d = 0; d < 3; ++d) w1[i][d] += g1[i] * r[d];
                    }
                }
            }
        }
    }
    if (len > 0) {
        float *dst = parts == 1 ? grad_features + (size_t)row * ld : part_rows + (size_t)item * C;
        if (h0) *reinterpret_cast<float4 *>(dst + 4 * q) = a0;
        if (h1) *reinterpret_cast<float4 *>(dst + 32 + 4 * q) = a1;
    }
    if (!want_rel) return;
    // d wx: the 8 groups of a wave as ((g0 + g1) + (g2 + g3)) + ((g4 + g5) + (g6 + g7)), then the 4 waves in order
    auto fold = [&](float v) {
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        return v;
    };
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float s0 = fold(w0[i][d]), s1 = WIDE ? fold(w1[i][d]) : 0.f;
            if ((threadIdx.x & 63) < 8) {
                wsum[wave][4 * q + i][d] = s0;
                if (WIDE) wsum[wave][32 + 4 * q + i][d] = s1;
            }
        }
    __syncthreads();
    for (int e = threadIdx.x; e < C * 3; e += 256) {
        const int c = e / 3, d = e - c * 3;
        wx_part[(size_t)blockIdx.x * C * 3 + e] = ((wsum[0][c][d] + wsum[1][c][d]) + wsum[2][c][d]) + wsum[3][c][d];
    }
}

// the same for any C <= 64 (lanes = channels c and c + 32, one row per instruction)
__global__ __launch_bounds__(256) void qg_stack_bwd_rows_kernel(int n_items, int C, int nsample, const int4 *__restrict__ items,
                                                                const int *__restrict__ list, const float *__restrict__ g_t,
                                                                const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                                float *__restrict__ grad_features, int ld,
                                                                float *__restrict__ part_rows, float *__restrict__ wx_part) {
    __shared__ float wsum[8][64][3];
    const int hw = threadIdx.x >> 5, lane = threadIdx.x & 31;
    const int item = blockIdx.x * 8 + hw;
    const int4 it = item < n_items ? items[item] : make_int4(0, 0, 0, 0);
    const int row = it.x, begin = it.y, len = it.z, parts = it.w;
    const bool c0 = lane < C, c1 = lane + 32 < C;
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
    float w0[3] = {0.f, 0.f, 0.f}, w1[3] = {0.f, 0.f, 0.f};
    float px = 0.f, py = 0.f, pz = 0.f;
    const bool want_rel = wx_part != nullptr;
    if (len > 0 && want_rel) {
        px = xyz[(size_t)row * 3];
        py = xyz[(size_t)row * 3 + 1];
        pz = xyz[(size_t)row * 3 + 2];
    }
    for (int e0 = 0; e0 < len; e0 += 32) {
        const int ne = min(32, len - e0);
        const QrRound cur = qr_load_round(list, begin + e0, ne, lane, want_rel, new_xyz, nsample, px, py, pz);
        auto step = [&](int e, int u) {
            const int col = __shfl(cur.col, e, 32);
            const float *g = g_t + (size_t)col * C;
            const float g0 = c0 ? g[lane] : 0.f, g1 = c1 ? g[lane + 32] : 0.f;
            a0[u] += g0;
            a1[u] += g1;
            if (want_rel) {
                const float rx = __shfl(cur.rx, e, 32), ry = __shfl(cur.ry, e, 32), rz = __shfl(cur.rz, e, 32);
                w0[0] += g0 * rx;
                w0[1] += g0 * ry;
                w0[2] += g0 * rz;
                w1[0] += g1 * rx;
                w1[1] += g1 * ry;
                w1[2] += g1 * rz;
            }
        };
        int e = 0;
        for (; e + 4 <= ne; e += 4) {
            step(e, 0);
            step(e + 1, 1);
            step(e + 2, 2);
            step(e + 3, 3);
        }
        for (; e < ne; ++e) step(e, 0);
    }
    if (len > 0) {
        float *dst = parts == 1 ? grad_features + (size_t)row * ld : part_rows + (size_t)item * C;
        if (c0) dst[lane] = (a0[0] + a0[1]) + (a0[2] + a0[3]);
        if (c1) dst[lane + 32] = (a1[0] + a1[1]) + (a1[2] + a1[3]);
    }
    if (!want_rel) return;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        wsum[hw][lane][d] = w0[d];
        wsum[hw][lane + 32][d] = w1[d];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < C * 3; e += 256) {
        const int c = e / 3, d = e - c * 3;
        float t = 0.f;
#pragma unroll
        for (int h = 0; h < 8; ++h) t += wsum[h][c][d];
        wx_part[(size_t)blockIdx.x * C * 3 + e] = t;
    }
}

// grid ceil(max_multi / 8), 256 threads: a half-wave per row that was cut into several work items
__global__ __launch_bounds__(256) void qg_stack_bwd_combine_kernel(int C, const int *__restrict__ head, const int *__restrict__ multi_rows,
                                                                   const int *__restrict__ row_item, const int4 *__restrict__ items,
                                                                   const float *__restrict__ part_rows, float *__restrict__ grad_features,
                                                                   int ld) {
    const int i = blockIdx.x * 8 + (threadIdx.x >> 5), lane = threadIdx.x & 31;
    if (i >= head[1]) return;
    const int row = multi_rows[i];
    const int first = row_item[row];
    const int parts = items[first].w;
    float s0 = 0.f, s1 = 0.f;
    for (int q = 0; q < parts; ++q) {
        const float *src = part_rows + (size_t)(first + q) * C;
        if (lane < C) s0 += src[lane];
        if (lane + 32 < C) s1 += src[lane + 32];
    }
    if (lane < C) grad_features[(size_t)row * ld + lane] = s0;
    if (lane + 32 < C) grad_features[(size_t)row * ld + lane + 32] = s1;
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
                        T *rel_out, T *y_out, void *stream, const char *what, float *out_stats = nullptr) {
    MGAR_REQUIRE(B >= 0 && M >= 0 && C >= 0 && nsample >= 0, "query_group (stack) fwd: negative size");
    const long long total = (long long)M * nsample;
    if (B == 0 || total == 0) return MGAR_OK;
    MGAR_REQUIRE(xyz && xyz_batch_cnt && new_xyz && new_xyz_batch_cnt && idx && (features || C == 0) && (y_out || C == 0) &&
                     (rel_out || y_out), "query_group (stack) fwd: null pointer");
    MGAR_REQUIRE(out_stats == nullptr || total % QS_COLS == 0, "query_group (stack) fwd: output statistics need M * nsample % 128 == 0");
    KtScope kt(KT_QUERY_GROUP_FWD, (hipStream_t)stream, (double)total * (4.0 + (double)sizeof(T) * (C + 3)));
    hipLaunchKernelGGL(qg_stack_fwd_kernel<T>, dim3(ceil_div(total, QS_COLS)), dim3(256), 0, (hipStream_t)stream, B, M, C, nsample,
                       xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, ld, wx, idx, rel_out, y_out, out_stats);
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

// The same, also leaving the BatchNorm statistics partials of y_out: out_stats (C, M * nsample / 128, 2) floats = per (channel,
// 128-column tile) the tile's mean and sum of squared deviations (chunk format of mgar_bn_stats_from_partials, chunk = 128).
// M * nsample % 128 == 0.  fp32.
QG_API int mgar_query_group_proj_stack_fwd_stats(int B, int M, int C, int nsample, const float *xyz, const int *xyz_batch_cnt,
                                                 const float *new_xyz, const int *new_xyz_batch_cnt, const float *zf, int zf_ld,
                                                 const float *wx, const int *idx, float *rel_out, float *y_out, float *out_stats,
                                                 void *stream) {
    MGAR_REQUIRE(wx && zf && y_out && out_stats && zf_ld >= C, "query_group_proj_stack_fwd_stats: null pointer or zf_ld < C");
    return qg_stack_fwd<float>(B, M, C, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, zf, zf_ld, wx, idx, rel_out, y_out,
                               stream, "query_group_proj_stack_fwd_stats: launch failed", out_stats);
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

// ---- stack backward without atomics: see qg_inv_index_kernel / qg_stack_bwd_rows_kernel ------------------------------------------
// Capacity (in items of 4 ints) of the work-item array for B samples, N source rows, total = M * nsample columns.
QG_API long long mgar_query_group_stack_inverse_items(int B, int N, long long total) {
    if (B < 0 || N < 0 || total < 0) return -1;
    return (long long)N + total / QR_PART + (N + QW_ROWS - 1) / QW_ROWS + B + 1;
}
// Ints the index occupies besides list / items / row_item: a head (window count, multi-part row count), per window its first
// list slot, table entry and per-run offsets, the rows cut into several items, and the window-split copy of the columns.
// The first 2 + total / QR_PART + 1 ints (head, multi_rows) are read again by mgar_query_group_stack_bwd_rows.
static long long qi_nwin_ub(int B, int N) { return (long long)(N + QW_ROWS - 1) / QW_ROWS + B; }
QG_API long long mgar_query_group_stack_inverse_workspace_ints(int B, int N, long long total) {
    if (B < 0 || N < 0 || total < 0) return -1;
    return 4 + (total / QR_PART + 1) + qi_nwin_ub(B, N) * (1 + 4 + QP_RUNS) + 2 * total + 4;
}
// idx: raw ball-query result (M, nsample).  list (M*nsample ints), items (capacity above, ZERO-FILLED by the caller), row_item (N),
// workspace (size above).  No sample may hold more than QP_WINDOWS * QW_ROWS = 262 144 source rows (not checked: the counts
// live on the device) -- the caller guarantees it.
QG_API int mgar_query_group_stack_inverse_index(int B, int M, int nsample, int N, const int *idx, const int *new_xyz_batch_cnt,
                                                const int *xyz_batch_cnt, int *workspace, int *list, int *items, int *row_item,
                                                void *stream) {
    MGAR_REQUIRE(B >= 0 && M >= 0 && nsample >= 1 && N >= 0, "query_group_stack_inverse_index: bad sizes");
    MGAR_REQUIRE((long long)M * nsample < (1ll << 31) - 64 * QW_LOADS, "query_group_stack_inverse_index: M * nsample >= 2^31");
    MGAR_REQUIRE(B <= 65535, "query_group_stack_inverse_index: B > 65535");
    if (B == 0 || N == 0) return MGAR_OK;
    MGAR_REQUIRE(new_xyz_batch_cnt && xyz_batch_cnt && workspace && list && items && row_item,
                 "query_group_stack_inverse_index: null pointer");
    MGAR_REQUIRE(M == 0 || idx, "query_group_stack_inverse_index: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const long long nwin_ub = qi_nwin_ub(B, N), total = (long long)M * nsample;
    int *head = workspace, *multi_rows = head + 4, *wbase = multi_rows + (total / QR_PART + 1);
    int *wtab = wbase + nwin_ub;
    wtab += (4 - ((wtab - workspace) & 3)) & 3;                          // int4 entries: 16-byte aligned (the caller's buffer is)
    int *rc = wtab + 4 * nwin_ub, *wcol = rc + nwin_ub * QP_RUNS, *wkey = wcol + total;
    // the index columns once read per pass (3 x 4 B), the window-split copy written and read (2 x 8 B), the list written (4 B)
    KtScope kt(KT_QG_INDEX, st, 32.0 * (double)total);
    hipLaunchKernelGGL(qg_inv_split_count_kernel, dim3(QP_GROUPS, B), dim3(256), 0, st, nsample, idx, new_xyz_batch_cnt, xyz_batch_cnt, rc);
    hipLaunchKernelGGL(qg_inv_split_scan_kernel, dim3(B), dim3(256), 0, st, B, nsample, new_xyz_batch_cnt, xyz_batch_cnt, rc, wbase,
                       reinterpret_cast<int4 *>(wtab), head);
    hipLaunchKernelGGL(qg_inv_split_fill_kernel, dim3(QP_GROUPS, B), dim3(256), 0, st, nsample, idx, new_xyz_batch_cnt, xyz_batch_cnt, rc,
                       wbase, wcol, wkey);
    hipLaunchKernelGGL(qg_inv_window_sort_kernel, dim3((unsigned)nwin_ub), dim3(QW_THREADS), 0, st, reinterpret_cast<const int4 *>(wtab),
                       head, wcol, wkey, list, reinterpret_cast<int4 *>(items), row_item, multi_rows);
    return check_launch("query_group_stack_inverse_index: launch failed");
}
// g_t: ROW-MAJOR gradient (M*nsample, C), C <= 64.  grad_zf (N, ld): rows with references are written, the others left alone.
// workspace: the index's (its head and multi-part row list).  part_rows: n_items * C floats of scratch.  wx_part: ceil(n_items / 8) * C * 3 floats, ZERO-FILLED by the caller: its sum over the
// first axis is d wx (C, 3) = sum_col g_t[col] (x) (xyz[row] - new_xyz[col / nsample]); NULL (with xyz, new_xyz) to skip it.
QG_API int mgar_query_group_stack_bwd_rows(int n_items, int N, int C, int nsample, const int *workspace, const int *items,
                                           const int *row_item, const int *list, const float *g_t, const float *xyz,
                                           const float *new_xyz, float *grad_zf, int ld, float *part_rows, float *wx_part,
                                           long long total, void *stream) {
    MGAR_REQUIRE(n_items >= 0 && N >= 0 && C >= 1 && ld >= C && nsample >= 1 && total >= 0, "query_group_stack_bwd_rows: bad sizes");
    if (C > 64) {
        set_error("query_group_stack_bwd_rows: C <= 64");
        return MGAR_EUNSUPPORTED;
    }
    if (n_items == 0 || N == 0) return MGAR_OK;
    MGAR_REQUIRE(workspace && items && row_item && list && g_t && grad_zf && part_rows, "query_group_stack_bwd_rows: null pointer");
    MGAR_REQUIRE(!wx_part || (xyz && new_xyz), "query_group_stack_bwd_rows: wx_part needs xyz and new_xyz");
    hipStream_t st = (hipStream_t)stream;
    {
        KtScope kt(KT_QUERY_GROUP_BWD, st, (double)total * (4.0 + 4.0 * C));
        const bool vec = C % 4 == 0 && ld % 4 == 0 && ((uintptr_t)g_t | (uintptr_t)grad_zf | (uintptr_t)part_rows) % 16 == 0;
        const bool short_lists = total < 32ll * N;            // on average: a lane group of 8 per item then beats a half-wave
        const int4 *it4 = reinterpret_cast<const int4 *>(items);
        if (vec && short_lists && C > 32)
            hipLaunchKernelGGL(qg_stack_bwd_rows_grp_kernel<true>, dim3(ceil_div(n_items, 32)), dim3(256), 0, st, n_items, C, nsample, it4,
                               list, g_t, xyz, new_xyz, grad_zf, ld, part_rows, wx_part);
        else if (vec && short_lists)
            hipLaunchKernelGGL(qg_stack_bwd_rows_grp_kernel<false>, dim3(ceil_div(n_items, 32)), dim3(256), 0, st, n_items, C, nsample, it4,
                               list, g_t, xyz, new_xyz, grad_zf, ld, part_rows, wx_part);
        else if (vec && C > 32)
            hipLaunchKernelGGL(qg_stack_bwd_rows_vec_kernel<true>, dim3(ceil_div(n_items, 8)), dim3(256), 0, st, n_items, C, nsample, it4,
                               list, g_t, xyz, new_xyz, grad_zf, ld, part_rows, wx_part);
        else if (vec)
            hipLaunchKernelGGL(qg_stack_bwd_rows_vec_kernel<false>, dim3(ceil_div(n_items, 8)), dim3(256), 0, st, n_items, C, nsample, it4,
                               list, g_t, xyz, new_xyz, grad_zf, ld, part_rows, wx_part);
        else
            hipLaunchKernelGGL(qg_stack_bwd_rows_kernel, dim3(ceil_div(n_items, 8)), dim3(256), 0, st, n_items, C, nsample, it4, list, g_t,
                               xyz, new_xyz, grad_zf, ld, part_rows, wx_part);
        hipLaunchKernelGGL(qg_stack_bwd_combine_kernel, dim3(ceil_div(total / QR_PART + 1, 8)), dim3(256), 0, st, C, workspace,
                           workspace + 4, row_item, reinterpret_cast<const int4 *>(items), part_rows, grad_zf, ld);
    }
    return check_launch("query_group_stack_bwd_rows: launch failed");
}
