// voxel_query.hip -- neighbour-voxel query (stack layout) for gfx950.
//
// Replaces  pcdet/ops/pointnet2/pointnet2_stack/src/voxel_query_gpu.cu:10-113
//
// One lane per query.  Each query walks the (2z+1)(2y+1)(2x+1) cells around its voxel in
// the reference's dz, dy, dx ascending order, looks the cell up in the dense
// point_indices (B,Z,Y,X) table and tests the voxel centre with d2 <= r^2 (NON-strict,
// voxel_query_gpu.cu:65).  This is latency-bound dependent gathers (4 B table lookup, then
// 12 B xyz on occupied cells), so the kernel is organised to keep many loads in flight:
// a whole x-row of the neighbourhood (2x+1 consecutive table entries) is fetched before any
// of it is consumed, and occupancy is high (no LDS in the scan, < 48 VGPRs).
// The hit rows are staged in LDS [slot][query] and written out coalesced with the
// first-hit padding, exactly like ball_query.hip.  A lane stops appending at nsample hits
// (the reference keeps scanning but can no longer change its row).
#include "common.hpp"
#include "voxel_hash.hpp"

namespace mgar {

constexpr int VQ_THREADS = 256;
constexpr int VQ_ROW_STRIDE = VQ_THREADS + 1;
constexpr int VQ_MAX_XROW = 17;  // x_range <= 8 handled with the row prefetch

// HASH = false: point_indices is the reference's dense (B, R1, R2, R3) table (common_utils.py:244-252);
// HASH = true : the same lookups go through the voxel hash table (table_keys / table_vals / mask) -- same cells, same
//               order, same results, without the dense table (80 MB per sample at the shipped 2000 x 2000 x 40 grid).
template <bool HASH>
__global__ __launch_bounds__(VQ_THREADS) void voxel_query_kernel(int M, int R1, int R2, int R3, int nsample,
                                                                 float radius2, int z_range, int y_range, int x_range,
                                                                 const float *__restrict__ new_xyz,
                                                                 const float *__restrict__ xyz,
                                                                 const int *__restrict__ new_coords,
                                                                 const int *__restrict__ point_indices,
                                                                 const long long *__restrict__ table_keys,
                                                                 const int *__restrict__ table_vals, int mask,
                                                                 int *__restrict__ idx) {
    extern __shared__ int lds[];
    int *rows = lds;
    int *cnts = lds + nsample * VQ_ROW_STRIDE;
    const int tid = threadIdx.x;
    const int q0 = blockIdx.x * VQ_THREADS;
    const int q = q0 + tid;
    int cnt = 0;
    if (q < M) {
        const float nx = new_xyz[(size_t)q * 3 + 0], ny = new_xyz[(size_t)q * 3 + 1], nz = new_xyz[(size_t)q * 3 + 2];
        const int b = new_coords[(size_t)q * 4 + 0], cz = new_coords[(size_t)q * 4 + 1];
        const int cy = new_coords[(size_t)q * 4 + 2], cx = new_coords[(size_t)q * 4 + 3];
        const int x_lo = max(cx - x_range, 0), x_hi = min(cx + x_range, R3 - 1);
        const int z_lo = max(cz - z_range, 0), z_hi = min(cz + z_range, R1 - 1);
        const int y_lo = max(cy - y_range, 0), y_hi = min(cy + y_range, R2 - 1);
        const bool prefetch_rows = (2 * x_range + 1) <= VQ_MAX_XROW;
        for (int z = z_lo; z <= z_hi && cnt < nsample; ++z) {
            for (int y = y_lo; y <= y_hi && cnt < nsample; ++y) {
                const int *cell = HASH ? nullptr : point_indices + (((size_t)b * R1 + z) * R2 + y) * R3;
                auto at = [&](int x) -> int {
                    return HASH ? sph_find(table_keys, table_vals, mask, sph_key(b, z, y, x, R1, R2, R3)) : cell[x];
                };
                if (prefetch_rows) {
                    int nb[VQ_MAX_XROW];
#pragma unroll
                    for (int i = 0; i < VQ_MAX_XROW; ++i) {
                        const int x = x_lo + i;
                        nb[i] = x <= x_hi ? at(x) : -1;
                    }
#pragma unroll
                    for (int i = 0; i < VQ_MAX_XROW; ++i) {
                        const int k = nb[i];
                        if (k < 0 || cnt >= nsample) continue;
                        const float d2 = d2_of(xyz[(size_t)k * 3 + 0] - nx, xyz[(size_t)k * 3 + 1] - ny,
                                               xyz[(size_t)k * 3 + 2] - nz);
                        if (d2 > radius2) continue;
                        rows[cnt * VQ_ROW_STRIDE + tid] = k;
                        ++cnt;
                    }
                } else {
                    for (int x = x_lo; x <= x_hi && cnt < nsample; ++x) {
                        const int k = at(x);
                        if (k < 0) continue;
                        const float d2 = d2_of(xyz[(size_t)k * 3 + 0] - nx, xyz[(size_t)k * 3 + 1] - ny,
                                               xyz[(size_t)k * 3 + 2] - nz);
                        if (d2 > radius2) continue;
                        rows[cnt * VQ_ROW_STRIDE + tid] = k;
                        ++cnt;
                    }
                }
            }
        }
    }
    cnts[tid] = cnt;
    __syncthreads();
    const int nq = min(VQ_THREADS, M - q0);
    const int total = nq * nsample;
    int *out = idx + (size_t)q0 * nsample;
    for (int e = tid; e < total; e += VQ_THREADS) {
        const int ql = e / nsample, s = e - ql * nsample;
        const int c = cnts[ql];
        if (c > 0) out[e] = rows[(s < c ? s : 0) * VQ_ROW_STRIDE + ql];
        else if (s == 0) out[e] = -1;  // voxel_query_gpu.cu:88
    }
}

}  // namespace mgar

using namespace mgar;

static int voxel_query_launch(bool hash, int M, int R1, int R2, int R3, int nsample, float radius, int z_range, int y_range, int x_range,
                              const float *new_xyz, const float *xyz, const int *new_coords, const int *point_indices,
                              const long long *table_keys, const int *table_vals, int capacity, int *idx, void *stream) {
    MGAR_REQUIRE(M >= 0 && R1 > 0 && R2 > 0 && R3 > 0, "voxel_query: bad grid size");
    MGAR_REQUIRE(z_range >= 0 && y_range >= 0 && x_range >= 0, "voxel_query: negative range");
    if (nsample < 1 || nsample > MGAR_MAX_NSAMPLE) {
        set_error("voxel_query: nsample outside [1, MGAR_MAX_NSAMPLE]");
        return MGAR_EUNSUPPORTED;
    }
    if (M == 0) return MGAR_OK;
    MGAR_REQUIRE(new_xyz && xyz && new_coords && idx, "voxel_query: null pointer");
    MGAR_REQUIRE(hash ? (table_keys && table_vals && capacity > 0 && (capacity & (capacity - 1)) == 0) : point_indices != nullptr,
                 "voxel_query: missing lookup table");
    const size_t lds = (size_t)(nsample * VQ_ROW_STRIDE + VQ_THREADS) * sizeof(int);
    static bool attr_set = false;
    if (!attr_set) {
        const int max_lds = (MGAR_MAX_NSAMPLE * VQ_ROW_STRIDE + VQ_THREADS) * (int)sizeof(int);
        (void)hipFuncSetAttribute((const void *)voxel_query_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
        (void)hipFuncSetAttribute((const void *)voxel_query_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
        attr_set = true;
    }
    const double cells = (2.0 * z_range + 1) * (2.0 * y_range + 1) * (2.0 * x_range + 1);
    // 12 M + 16 M + 4 M nsample + 4 K per query of table lookups (hit xyz gathers are data dependent: not counted)
    KtScope kt(KT_VOXEL_QUERY, (hipStream_t)stream, (double)M * (28.0 + 4.0 * nsample + 4.0 * cells));
    if (hash)
        hipLaunchKernelGGL(voxel_query_kernel<true>, dim3(ceil_div(M, VQ_THREADS)), dim3(VQ_THREADS), lds, (hipStream_t)stream, M, R1, R2,
                           R3, nsample, radius * radius, z_range, y_range, x_range, new_xyz, xyz, new_coords, (const int *)nullptr,
                           table_keys, table_vals, capacity - 1, idx);
    else
        hipLaunchKernelGGL(voxel_query_kernel<false>, dim3(ceil_div(M, VQ_THREADS)), dim3(VQ_THREADS), lds, (hipStream_t)stream, M, R1, R2,
                           R3, nsample, radius * radius, z_range, y_range, x_range, new_xyz, xyz, new_coords, point_indices,
                           (const long long *)nullptr, (const int *)nullptr, 0, idx);
    return check_launch("voxel_query: launch failed");
}

extern "C" __attribute__((visibility("default"))) int mgar_voxel_query_stack(int M, int R1, int R2, int R3, int nsample, float radius, int z_range,
                                      int y_range, int x_range, const float *new_xyz, const float *xyz,
                                      const int *new_coords, const int *point_indices, int *idx, void *stream) {
    return voxel_query_launch(false, M, R1, R2, R3, nsample, radius, z_range, y_range, x_range, new_xyz, xyz, new_coords, point_indices,
                              nullptr, nullptr, 0, idx, stream);
}

// The same query with the voxel -> row lookups served by the hash table of mgar_voxel_hash_build instead of the dense
// (B, R1, R2, R3) table: identical results (tests/test_sparse_conv_gpu.py), O(active voxels) memory.
extern "C" __attribute__((visibility("default"))) int mgar_voxel_query_hash_stack(int M, int R1, int R2, int R3, int nsample, float radius,
                                      int z_range, int y_range, int x_range, const float *new_xyz, const float *xyz,
                                      const int *new_coords, const long long *table_keys, const int *table_vals, int capacity,
                                      int *idx, void *stream) {
    return voxel_query_launch(true, M, R1, R2, R3, nsample, radius, z_range, y_range, x_range, new_xyz, xyz, new_coords, nullptr,
                              table_keys, table_vals, capacity, idx, stream);
}
