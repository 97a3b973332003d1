// ball_query.hip -- ball query (batch + stack layouts) for gfx950.
//
// Replaces  pcdet/ops/pointnet2/pointnet2_batch/src/ball_query_gpu.cu:15-51   (batch)
//           pcdet/ops/pointnet2/pointnet2_stack/src/ball_query_gpu.cu:16-66   (stack)
//
// Reference shape: one thread per query, every thread walks the whole cloud with
// 12-byte-strided per-thread loads.  Here:
//   * one lane per query, 256 queries per workgroup, all of one cloud;
//   * the cloud is a wave-UNIFORM stream: the point index k is a scalar, so the
//     coordinates arrive through the scalar cache into SGPRs (s_load_dwordxN, one
//     instruction per ~5 points) and feed the VALU as scalar operands -- no per-lane
//     global loads, no LDS broadcast traffic in the scan loop;
//   * hits are appended to a per-workgroup LDS image [slot][query] (stride 257 so the
//     transposed write-out is bank-conflict free) and written out once, coalesced, with
//     the reference's "pad the row with the first hit" rule applied on the way out;
//   * a workgroup leaves the scan as soon as all of its 256 queries are full.
// The scan is VALU-bound: 7 VALU instructions per (query, point) pair.
#include "common.hpp"

namespace mgar {

constexpr int BQ_THREADS = 256;
constexpr int BQ_ROW_STRIDE = BQ_THREADS + 1;
constexpr int BQ_CHUNK = 16;  // points fetched into SGPRs per step (48 dwords)

typedef const float __attribute__((address_space(4))) *cfloat_p;

template <bool STACK>
__global__ __launch_bounds__(BQ_THREADS) void ball_query_kernel(
    int B, int n_batch, int m_batch, float radius2, int nsample, const float *__restrict__ new_xyz,
    const int *__restrict__ new_xyz_batch_cnt, const float *__restrict__ xyz, const int *__restrict__ xyz_batch_cnt,
    int *__restrict__ idx) {
    extern __shared__ int lds[];  // [nsample][257] hit rows, then [256] counts
    int *rows = lds;
    int *cnts = lds + nsample * BQ_ROW_STRIDE;

    // ---- which cloud / which 256-query tile does this workgroup own? (wave-uniform) ----
    int q0, q_end, p_start, n;
    if (STACK) {
        int g = blockIdx.x, qs = 0, ps = 0, bs = 0;
        bool found = false;
        for (; bs < B; ++bs) {
            const int mi = new_xyz_batch_cnt[bs];
            const int nb = (mi + BQ_THREADS - 1) / BQ_THREADS;
            if (g < nb) { found = true; break; }
            g -= nb;
            qs += mi;
            ps += xyz_batch_cnt[bs];
        }
        if (!found) return;
        q0 = qs + g * BQ_THREADS;
        q_end = qs + new_xyz_batch_cnt[bs];
        p_start = ps;
        n = xyz_batch_cnt[bs];
    } else {
        const int bs = blockIdx.y;
        q0 = bs * m_batch + blockIdx.x * BQ_THREADS;
        q_end = (bs + 1) * m_batch;
        p_start = bs * n_batch;
        n = n_batch;
    }

    const int tid = threadIdx.x;
    const int q = q0 + tid;
    const bool valid = q < q_end;
    // A lane whose row is full (or that has no query) is parked by making its query x
    // infinite: d2 becomes +inf/NaN and "d2 < r2" is false from then on, so the scan needs
    // no per-point "cnt < nsample" test.
    float qx = __builtin_inff(), qy = 0.f, qz = 0.f;
    if (valid) {
        qx = new_xyz[(size_t)q * 3 + 0];
        qy = new_xyz[(size_t)q * 3 + 1];
        qz = new_xyz[(size_t)q * 3 + 2];
    }
    int cnt = 0;

    // scalar (constant address space) view of this cloud's points
    cfloat_p P = (cfloat_p)(xyz + (size_t)p_start * 3);

    // Scan.  A chunk of BQ_CHUNK points is fetched into SGPRs first (the loads sit in one
    // basic block so they issue back to back), then tested.  The append path is behind a
    // wave-uniform branch on the hit ballot: the common "nobody hit" case costs 7 VALU.
    const int waddr = tid;
    int k0 = 0;
    for (; k0 + BQ_CHUNK <= n; k0 += BQ_CHUNK) {
        float c[BQ_CHUNK * 3];
#pragma unroll
        for (int i = 0; i < BQ_CHUNK * 3; ++i) c[i] = P[k0 * 3 + i];
#pragma unroll
        for (int j = 0; j < BQ_CHUNK; ++j) {
            const float d2 = d2_of(qx - c[j * 3 + 0], qy - c[j * 3 + 1], qz - c[j * 3 + 2]);
            if (d2 < radius2) {  // rare path
                // the asm marker makes the backend keep the s_cbranch_execz over this block
                // (it drops the branch for short blocks, whose VALU then issues with EXEC=0)
                asm volatile("; append hit" ::: "memory");
                rows[cnt * BQ_ROW_STRIDE + waddr] = k0 + j;
                if (++cnt == nsample) qx = __builtin_inff();
            }
        }
        // a wave leaves once all of its 64 queries are parked (no barrier needed here)
        if (__builtin_amdgcn_ballot_w64(qx != __builtin_inff()) == 0ull) { k0 = n; break; }
    }
    for (int k = k0; k < n; ++k) {  // tail (< BQ_CHUNK points)
        const float d2 = d2_of(qx - P[k * 3 + 0], qy - P[k * 3 + 1], qz - P[k * 3 + 2]);
        if (d2 < radius2) {
            rows[cnt * BQ_ROW_STRIDE + waddr] = k;
            if (++cnt == nsample) qx = __builtin_inff();
        }
    }

    cnts[tid] = valid ? cnt : -1;
    __syncthreads();

    // ---- coalesced write-out: lanes run along (query, slot) of the output rows ----
    const int nq = min(BQ_THREADS, q_end - q0);
    const int total = nq * nsample;
    int *out = idx + (size_t)q0 * nsample;
    for (int e = tid; e < total; e += BQ_THREADS) {
        const int ql = e / nsample, s = e - ql * nsample;
        const int c = cnts[ql];
        if (c > 0) {
            out[e] = rows[(s < c ? s : 0) * BQ_ROW_STRIDE + ql];
        } else if (STACK && s == 0) {
            out[e] = -1;  // pointnet2_stack/src/ball_query_gpu.cu:65
        }
        // batch layout: an empty ball leaves the caller's row untouched
    }
}

// ---- several radii in one scan (multi-scale grouping) ---------------------------------------------
// The scales of an MSG module query the SAME centres against the SAME cloud with different (radius,
// nsample).  One scan computes every squared distance once and tests it against all radii: the common
// "no hit for any radius" case costs 7 VALU for the distance + 1 compare against the lane's largest
// still-open radius, instead of 7 + 1 per scale.  Each radius keeps its own LDS rows, count and
// parking (its r^2 becomes -1 once its row is full); the results are exactly those of the
// single-radius kernel, scale by scale.
constexpr int BQ_MAX_RADII = 4;
struct BqMulti {
    float radius2[BQ_MAX_RADII];
    int nsample[BQ_MAX_RADII];
    int *idx[BQ_MAX_RADII];
};

template <bool STACK, int NR>
__global__ __launch_bounds__(BQ_THREADS) void ball_query_multi_kernel(int B, int n_batch, int m_batch, BqMulti A,
                                                                      const float *__restrict__ new_xyz,
                                                                      const int *__restrict__ new_xyz_batch_cnt,
                                                                      const float *__restrict__ xyz,
                                                                      const int *__restrict__ xyz_batch_cnt) {
    extern __shared__ int lds[];  // per radius: [nsample_r][257] hit rows; then [NR][256] counts
    int *rows[NR];
    int off = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) { rows[r] = lds + off; off += A.nsample[r] * BQ_ROW_STRIDE; }
    int *cnts = lds + off;

    int q0, q_end, p_start, n;
    if (STACK) {
        int g = blockIdx.x, qs = 0, ps = 0, bs = 0;
        bool found = false;
        for (; bs < B; ++bs) {
            const int mi = new_xyz_batch_cnt[bs];
            const int nb = (mi + BQ_THREADS - 1) / BQ_THREADS;
            if (g < nb) { found = true; break; }
            g -= nb;
            qs += mi;
            ps += xyz_batch_cnt[bs];
        }
        if (!found) return;
        q0 = qs + g * BQ_THREADS;
        q_end = qs + new_xyz_batch_cnt[bs];
        p_start = ps;
        n = xyz_batch_cnt[bs];
    } else {
        const int bs = blockIdx.y;
        q0 = bs * m_batch + blockIdx.x * BQ_THREADS;
        q_end = (bs + 1) * m_batch;
        p_start = bs * n_batch;
        n = n_batch;
    }
    const int tid = threadIdx.x;
    const int q = q0 + tid;
    const bool valid = q < q_end;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    float r2[NR];   // per lane: -1 once the row of that radius is full (or the lane has no query)
    int cnt[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) { r2[r] = valid ? A.radius2[r] : -1.f; cnt[r] = 0; }
    if (valid) {
        qx = new_xyz[(size_t)q * 3 + 0];
        qy = new_xyz[(size_t)q * 3 + 1];
        qz = new_xyz[(size_t)q * 3 + 2];
    }
    float rmax = -1.f;   // largest still-open radius^2 of this lane
#pragma unroll
    for (int r = 0; r < NR; ++r) rmax = fmaxf(rmax, r2[r]);

    cfloat_p P = (cfloat_p)(xyz + (size_t)p_start * 3);
    auto test = [&](float d2, int k) {
        if (d2 < rmax) {  // rare path (wave-uniform branch on the ballot)
            asm volatile("; append hit" ::: "memory");
#pragma unroll
            for (int r = 0; r < NR; ++r)
                if (d2 < r2[r]) {
                    rows[r][cnt[r] * BQ_ROW_STRIDE + tid] = k;
                    if (++cnt[r] == A.nsample[r]) r2[r] = -1.f;
                }
            rmax = -1.f;
#pragma unroll
            for (int r = 0; r < NR; ++r) rmax = fmaxf(rmax, r2[r]);
        }
    };
    int k0 = 0;
    for (; k0 + BQ_CHUNK <= n; k0 += BQ_CHUNK) {
        float c[BQ_CHUNK * 3];
#pragma unroll
        for (int i = 0; i < BQ_CHUNK * 3; ++i) c[i] = P[k0 * 3 + i];
#pragma unroll
        for (int j = 0; j < BQ_CHUNK; ++j) test(d2_of(qx - c[j * 3 + 0], qy - c[j * 3 + 1], qz - c[j * 3 + 2]), k0 + j);
        if (__builtin_amdgcn_ballot_w64(rmax >= 0.f) == 0ull) { k0 = n; break; }   // every row of every lane is full
    }
    for (int k = k0; k < n; ++k) test(d2_of(qx - P[k * 3 + 0], qy - P[k * 3 + 1], qz - P[k * 3 + 2]), k);

#pragma unroll
    for (int r = 0; r < NR; ++r) cnts[r * BQ_THREADS + tid] = valid ? cnt[r] : -1;
    __syncthreads();
    const int nq = min(BQ_THREADS, q_end - q0);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int ns = A.nsample[r];
        const int total = nq * ns;
        int *out = A.idx[r] + (size_t)q0 * ns;
        for (int e = tid; e < total; e += BQ_THREADS) {
            const int ql = e / ns, sl = e - ql * ns;
            const int c = cnts[r * BQ_THREADS + ql];
            if (c > 0) {
                out[e] = rows[r][(sl < c ? sl : 0) * BQ_ROW_STRIDE + ql];
            } else if (STACK && sl == 0) {
                out[e] = -1;
            }
        }
    }
}

template <bool STACK>
static int bq_multi_launch(int B, int n, int m, int grid_x, int grid_y, int nr, const float *radii, const int *nsamples,
                           int *const *idx, const float *new_xyz, const int *new_cnt, const float *xyz, const int *xyz_cnt,
                           double bytes, double flops, hipStream_t st, const char *what) {
    BqMulti A;
    int rows = 0;
    for (int r = 0; r < BQ_MAX_RADII; ++r) {
        A.radius2[r] = r < nr ? radii[r] * radii[r] : -1.f;
        A.nsample[r] = r < nr ? nsamples[r] : 0;
        A.idx[r] = r < nr ? idx[r] : nullptr;
        rows += A.nsample[r];
    }
    const size_t lds = ((size_t)rows * BQ_ROW_STRIDE + (size_t)nr * BQ_THREADS) * sizeof(int);
    if (lds > 160 * 1024) {
        set_error("ball_query_multi: the nsample rows of all radii do not fit LDS");
        return MGAR_EUNSUPPORTED;
    }
    KtScope kt(KT_BALL_QUERY, st, bytes, flops);
#define BQM(NR)                                                                                                              \
    {                                                                                                                        \
        static bool attr_set = false;                                                                                        \
        if (!attr_set) {                                                                                                     \
            (void)hipFuncSetAttribute((const void *)ball_query_multi_kernel<STACK, NR>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      160 * 1024);                                                                           \
            attr_set = true;                                                                                                 \
        }                                                                                                                    \
        hipLaunchKernelGGL((ball_query_multi_kernel<STACK, NR>), dim3(grid_x, grid_y), dim3(BQ_THREADS), lds, st, B, n, m, A,  \
                           new_xyz, new_cnt, xyz, xyz_cnt);                                                                  \
    }
    if (nr == 2) BQM(2) else if (nr == 3) BQM(3) else BQM(4)
#undef BQM
    return check_launch(what);
}

static int bq_check(int nsample) {
    if (nsample < 1 || nsample > MGAR_MAX_NSAMPLE) {
        set_error("ball_query: nsample outside [1, MGAR_MAX_NSAMPLE]");
        return MGAR_EUNSUPPORTED;
    }
    static bool attr_set = false;  // rows for nsample > 63 need more than the default 64 KB
    if (!attr_set) {
        const int max_lds = (MGAR_MAX_NSAMPLE * BQ_ROW_STRIDE + BQ_THREADS) * (int)sizeof(int);
        (void)hipFuncSetAttribute((const void *)ball_query_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
        (void)hipFuncSetAttribute((const void *)ball_query_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
        attr_set = true;
    }
    return MGAR_OK;
}

}  // namespace mgar

using namespace mgar;

extern "C" __attribute__((visibility("default"))) int mgar_ball_query_batch(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                                     const float *xyz, int *idx, void *stream) {
    MGAR_REQUIRE(b >= 0 && n >= 0 && m >= 0, "ball_query_batch: negative size");
    if (int e = bq_check(nsample)) return e;
    if (b == 0 || m == 0) return MGAR_OK;
    MGAR_REQUIRE(new_xyz && xyz && idx, "ball_query_batch: null pointer");
    MGAR_REQUIRE(b <= 65535, "ball_query_batch: b > 65535");
    const size_t lds = (size_t)(nsample * BQ_ROW_STRIDE + BQ_THREADS) * sizeof(int);
    dim3 grid(ceil_div(m, BQ_THREADS), b);
    KtScope kt(KT_BALL_QUERY, (hipStream_t)stream, (double)b * (12.0 * n + 12.0 * m + 4.0 * m * nsample), 8.0 * b * (double)m * n);   // <= m*n pair tests of 8 flop
    hipLaunchKernelGGL(ball_query_kernel<false>, grid, dim3(BQ_THREADS), lds, (hipStream_t)stream, b, n, m,
                       radius * radius, nsample, new_xyz, (const int *)nullptr, xyz, (const int *)nullptr, idx);
    return check_launch("ball_query_batch: launch failed");
}

extern "C" __attribute__((visibility("default"))) int mgar_ball_query_stack(int B, int M, float radius, int nsample, const float *new_xyz,
                                     const int *new_xyz_batch_cnt, const float *xyz, const int *xyz_batch_cnt,
                                     int *idx, void *stream) {
    MGAR_REQUIRE(B >= 0 && M >= 0, "ball_query_stack: negative size");
    if (int e = bq_check(nsample)) return e;
    if (B == 0 || M == 0) return MGAR_OK;
    MGAR_REQUIRE(new_xyz && xyz && idx && new_xyz_batch_cnt && xyz_batch_cnt, "ball_query_stack: null pointer");
    const size_t lds = (size_t)(nsample * BQ_ROW_STRIDE + BQ_THREADS) * sizeof(int);
    // sum_i ceil(M_i/256) <= ceil(M/256) + B; surplus workgroups exit at once.  The counts
    // live on the device, so sizing the grid exactly would cost a host sync.
    dim3 grid(ceil_div(M, BQ_THREADS) + B);
    KtScope kt(KT_BALL_QUERY, (hipStream_t)stream, 12.0 * M + 4.0 * (double)M * nsample);   // + 12 N: N lives on the device
    hipLaunchKernelGGL(ball_query_kernel<true>, grid, dim3(BQ_THREADS), lds, (hipStream_t)stream, B, 0, 0,
                       radius * radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx);
    return check_launch("ball_query_stack: launch failed");
}

// Several (radius, nsample) pairs against the same centres and cloud in one scan: idx[r] (b,m,nsample[r])
// gets exactly what mgar_ball_query_batch(radius[r], nsample[r]) writes.  2 <= nr <= 4.
extern "C" __attribute__((visibility("default"))) int mgar_ball_query_multi_batch(int b, int n, int m, int nr, const float *radii,
                                                                                 const int *nsamples, const float *new_xyz,
                                                                                 const float *xyz, int *const *idx, void *stream) {
    MGAR_REQUIRE(b >= 0 && n >= 0 && m >= 0, "ball_query_multi_batch: negative size");
    MGAR_REQUIRE(nr >= 2 && nr <= BQ_MAX_RADII && radii && nsamples && idx, "ball_query_multi_batch: 2 <= nr <= 4 radii");
    for (int r = 0; r < nr; ++r)
        if (int e = bq_check(nsamples[r])) return e;
    if (b == 0 || m == 0) return MGAR_OK;
    MGAR_REQUIRE(new_xyz && xyz, "ball_query_multi_batch: null pointer");
    for (int r = 0; r < nr; ++r) MGAR_REQUIRE(idx[r], "ball_query_multi_batch: null idx pointer");
    MGAR_REQUIRE(b <= 65535, "ball_query_multi_batch: b > 65535");
    double bytes = (double)b * (12.0 * n + 12.0 * m);
    for (int r = 0; r < nr; ++r) bytes += 4.0 * b * m * nsamples[r];
    return bq_multi_launch<false>(b, n, m, ceil_div(m, BQ_THREADS), b, nr, radii, nsamples, idx, new_xyz, nullptr, xyz, nullptr, bytes, 8.0 * b * (double)m * n,
                                  (hipStream_t)stream, "ball_query_multi_batch: launch failed");
}

extern "C" __attribute__((visibility("default"))) int mgar_ball_query_multi_stack(int B, int M, int nr, const float *radii,
                                                                                 const int *nsamples, const float *new_xyz,
                                                                                 const int *new_xyz_batch_cnt, const float *xyz,
                                                                                 const int *xyz_batch_cnt, int *const *idx,
                                                                                 void *stream) {
    MGAR_REQUIRE(B >= 0 && M >= 0, "ball_query_multi_stack: negative size");
    MGAR_REQUIRE(nr >= 2 && nr <= BQ_MAX_RADII && radii && nsamples && idx, "ball_query_multi_stack: 2 <= nr <= 4 radii");
    for (int r = 0; r < nr; ++r)
        if (int e = bq_check(nsamples[r])) return e;
    if (B == 0 || M == 0) return MGAR_OK;
    MGAR_REQUIRE(new_xyz && xyz && new_xyz_batch_cnt && xyz_batch_cnt, "ball_query_multi_stack: null pointer");
    for (int r = 0; r < nr; ++r) MGAR_REQUIRE(idx[r], "ball_query_multi_stack: null idx pointer");
    double bytes = 12.0 * M;
    for (int r = 0; r < nr; ++r) bytes += 4.0 * M * nsamples[r];
    return bq_multi_launch<true>(B, 0, 0, ceil_div(M, BQ_THREADS) + B, 1, nr, radii, nsamples, idx, new_xyz, new_xyz_batch_cnt, xyz,
                                 xyz_batch_cnt, bytes, 0.0, (hipStream_t)stream, "ball_query_multi_stack: launch failed");
}
