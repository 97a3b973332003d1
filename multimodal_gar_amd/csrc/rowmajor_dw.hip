// rowmajor_dw.hip -- weight gradient of a Linear layer over STACKED (row-major) operands on the
// exact-fp32 MFMA, gfx950.
//
//   dW[o][i] = sum_n A[n][o] * F[n][i]        A (N, Co) leading dimension lda, F (N, Ci) ldf
//
// This is the backward-weights of the "project" half of project-then-group on the stacked layout
// (zf = features W_f^T over the N source points of the RoI-grid pooling,
// pointnet2_stack/pointnet2_modules.py:33-40 / roi_heads point-grid pooling): Co x Ci = 96 x 128 with
// N = 1.97 M reduction rows at config c3.  The library launches 21 workgroups for it (no split-K):
// 2.3 ms per scale.  Row-major operands ARE the MFMA operand layout when the reduction runs over
// rows -- A-operand lane (o, k) = A[n0 + k][o], B-operand lane (k, i) = F[n0 + k][i], 128 contiguous
// bytes per half-wave -- so both stream global -> registers -> v_mfma_f32_32x32x2_f32 with no LDS.
// Every wave owns a strided set of 16-row slabs, keeps the whole Co x Ci accumulator in registers,
// and prefetches the next slab while the MFMAs of the current one run.  Workgroup partials go to a
// workspace and are summed in a fixed order (csrc/pointwise_dw.hip's reduce): deterministic.
#include "common.hpp"

namespace mgar {

typedef float __attribute__((ext_vector_type(16))) f32x16;

// rows per slab = 2 * KS (KS MFMA k-steps): 16, or 8 where the Co x Ci accumulator leaves fewer registers for
// the two prefetch buffers (a <3,4> instance with 16-row slabs spilled 81 VGPRs to scratch)
template <int OB, int IB> struct RdSlab { static constexpr int KS = OB * IB > 6 ? 4 : 8; };

template <int OB, int IB>
__global__ __launch_bounds__(256) void rowmajor_dw_kernel(const float *__restrict__ A, int lda, const float *__restrict__ F, int ldf,
                                                          long long N, int Co, int Ci, float *__restrict__ partial) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l = lane & 31, h = lane >> 5;
    constexpr int KS = RdSlab<OB, IB>::KS, RD_ROWS = 2 * KS;
    const long long nslab = (N + RD_ROWS - 1) / RD_ROWS;
    const long long wave_id = (long long)blockIdx.x * 4 + wave, nwaves = (long long)gridDim.x * 4;

    f32x16 acc[OB][IB];
#pragma unroll
    for (int a = 0; a < OB; ++a)
#pragma unroll
        for (int c = 0; c < IB; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    auto issue = [&](long long slab, float (&av)[KS][OB], float (&fv)[KS][IB]) {
        const long long n0 = slab * RD_ROWS + h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const long long n = n0 + 2 * ks;
            const bool ok = n < N;
#pragma unroll
            for (int a = 0; a < OB; ++a) av[ks][a] = (ok && a * 32 + l < Co) ? A[(size_t)n * lda + a * 32 + l] : 0.f;
#pragma unroll
            for (int c = 0; c < IB; ++c) fv[ks][c] = (ok && c * 32 + l < Ci) ? F[(size_t)n * ldf + c * 32 + l] : 0.f;
        }
    };
    auto compute = [&](float (&av)[KS][OB], float (&fv)[KS][IB]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int a = 0; a < OB; ++a)
#pragma unroll
                for (int c = 0; c < IB; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks][a], fv[ks][c], acc[a][c], 0, 0, 0);
    };

    float a0[KS][OB], f0[KS][IB], a1[KS][OB], f1[KS][IB];
    long long slab = wave_id;
    if (slab < nslab) issue(slab, a0, f0);
    while (slab < nslab) {
        const long long s1 = slab + nwaves, s2 = s1 + nwaves;
        if (s1 < nslab) issue(s1, a1, f1);
        compute(a0, f0);
        if (s1 < nslab) {
            if (s2 < nslab) issue(s2, a0, f0);
            compute(a1, f1);
        }
        slab = s2;
    }
    // every wave stores its own partial block: slot = workgroup * 4 + wave
#pragma unroll
    for (int a = 0; a < OB; ++a)
#pragma unroll
        for (int c = 0; c < IB; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int i = c * 32 + l;
                if (o < Co && i < Ci) partial[((size_t)wave_id * Co + o) * Ci + i] = acc[a][c][r];
            }
}

// same fixed-order reduce as csrc/pointwise_dw.hip: 1024 threads = 64 outputs x 16 slices of the partials, 8 loads in flight
constexpr int RDR_SLICES = 16;
__global__ __launch_bounds__(64 * RDR_SLICES) void rowmajor_dw_reduce_kernel(const float *__restrict__ partial, int np, int n_out,
                                                                             float *__restrict__ dw) {
    __shared__ float part[RDR_SLICES][64];
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
    const int per = (np + RDR_SLICES - 1) / RDR_SLICES, g0 = min(slice * per, np), g1 = min(g0 + per, np);
    float s = 0.f;
    if (e < n_out) {
        int g = g0;
        for (; g + 8 <= g1; g += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(g + u) * n_out + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; g < g1; ++g) s += partial[(size_t)g * n_out + e];
    }
    part[slice][threadIdx.x & 63] = s;
    __syncthreads();
    if (slice == 0 && e < n_out) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < RDR_SLICES; ++i) t += part[i][threadIdx.x];
        dw[e] = t;
    }
}

static int rd_workgroups(long long N) {
    const long long nslab = (N + 15) / 16;
    long long wgs = (nslab + 4 * 8 - 1) / (4 * 8);  // at least 8 slabs per wave
    if (wgs > 256) wgs = 256;                        // one workgroup per CU: the accumulator fills the register file
    if (wgs < 1) wgs = 1;
    return (int)wgs;
}

template <int OB, int IB>
static void launch_rd(const float *A, int lda, const float *F, int ldf, long long N, int Co, int Ci, float *partial, hipStream_t st) {
    hipLaunchKernelGGL((rowmajor_dw_kernel<OB, IB>), dim3(rd_workgroups(N)), dim3(256), 0, st, A, lda, F, ldf, N, Co, Ci, partial);
}

}  // namespace mgar

using namespace mgar;

extern "C" __attribute__((visibility("default"))) int mgar_rowmajor_dw_workspace_floats(long long N, int Co, int Ci) {
    if (N <= 0 || Co <= 0 || Ci <= 0) return 0;
    return rd_workgroups(N) * 4 * Co * Ci;
}

extern "C" __attribute__((visibility("default"))) int mgar_rowmajor_dw(const float *a, int lda, const float *f, int ldf, long long N,
                                                                      int Co, int Ci, float *workspace, float *dw, void *stream) {
    MGAR_REQUIRE(N >= 0 && Co >= 0 && Ci >= 0 && lda >= Co && ldf >= Ci, "rowmajor_dw: bad sizes");
    if (Co == 0 || Ci == 0) return MGAR_OK;
    MGAR_REQUIRE(dw, "rowmajor_dw: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (N == 0) {
        (void)hipMemsetAsync(dw, 0, sizeof(float) * Co * Ci, st);
        return check_launch("rowmajor_dw: memset failed");
    }
    MGAR_REQUIRE(a && f && workspace, "rowmajor_dw: null pointer");
    const int ob = ceil_div(Co, 32), ib = ceil_div(Ci, 32);
    if (ob > 3 || ib > 4) {
        set_error("rowmajor_dw: needs Co <= 96 and Ci <= 128 (use the library GEMM otherwise)");
        return MGAR_EUNSUPPORTED;
    }
#define RD_CASE(O, I) if (ob == O && ib == I) launch_rd<O, I>(a, lda, f, ldf, N, Co, Ci, workspace, st);
    {
    KtScope kt(KT_ROWMAJOR_DW, st, 4.0 * (double)N * (Co + Ci), 2.0 * (double)N * Co * Ci);
    RD_CASE(1, 1) RD_CASE(1, 2) RD_CASE(1, 3) RD_CASE(1, 4)
    RD_CASE(2, 1) RD_CASE(2, 2) RD_CASE(2, 3) RD_CASE(2, 4)
    RD_CASE(3, 1) RD_CASE(3, 2) RD_CASE(3, 3) RD_CASE(3, 4)
    }
#undef RD_CASE
    const int np = rd_workgroups(N) * 4, n_out = Co * Ci;
    hipLaunchKernelGGL(rowmajor_dw_reduce_kernel, dim3(ceil_div(n_out, 64)), dim3(64 * RDR_SLICES), 0, st, workspace, np, n_out, dw);
    return check_launch("rowmajor_dw: launch failed");
}
