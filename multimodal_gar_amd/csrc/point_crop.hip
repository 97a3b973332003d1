// point_crop.hip -- per-actor point crop for gfx950: points_in_boxes and roipoint_pool3d.
//
// Replaces
//   pcdet/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:313-334   points_in_boxes_kernel
//   pcdet/ops/roipoint_pool3d/src/roipoint_pool3d_kernel.cu:38-166    assign_pts_to_box3d, get_pooled_idx, roipool3d_forward,
//                                                                      roipool3dLauncher
// (the box test of both: :15-36 lidar_to_local_coords / check_pt_in_box3d, MARGIN 1e-5).
//
// The reference's crop materialises a (B, N, M) int matrix of box membership (252 MB at config c3), cudaMalloc's and frees
// two scratch buffers per call, and then lets ONE THREAD per box walk all N points serially to collect the first S hits.
// Here a workgroup owns one (box, sample): its 256 lanes test 256 consecutive points at a time and append the hits in
// index order with wave ballots + a 4-entry LDS prefix (the same ordered compaction as ball_query.hip), stop as soon as S
// points are found, duplicate cyclically (idx[k] = idx[k % cnt], roipoint_pool3d_kernel.cu:89-96) and write the
// (S, 3 + C) crop with coalesced stores.  No scratch in HBM, no allocation, no serial walk.
//
// Arithmetic of the box test as the reference writes it: |z - cz| against dz / 2.0 and |local| against d / 2.0 + MARGIN are
// DOUBLE comparisons of float values; the rotation is cos / sin of the negated heading in float with un-contracted float
// products.  cos / sin are evaluated in double and rounded to float, which agrees with a correctly rounded float libm (the
// oracle's, pinned to the reference's CPU build in tests/test_point_crop_cpu.py) wherever that one is.
#include "common.hpp"

namespace mgar {

struct CropBox {
    float cx, cy, cz, hz;     // centre, half height (dz / 2 is exact in float)
    float cosa, sina;         // of -heading
    double tx, ty;            // dx / 2.0 + MARGIN, dy / 2.0 + MARGIN in double, MARGIN = 1e-5f
};

__device__ __forceinline__ CropBox crop_box(const float *__restrict__ b) {
    CropBox c;
    c.cx = b[0]; c.cy = b[1]; c.cz = b[2];
    c.hz = b[5] * 0.5f;
    const double a = -(double)b[6];
    c.cosa = (float)cos(a);
    c.sina = (float)sin(a);
    c.tx = (double)b[3] / 2.0 + (double)1e-5f;
    c.ty = (double)b[4] / 2.0 + (double)1e-5f;
    return c;
}

__device__ __forceinline__ bool in_crop_box(const CropBox &c, float x, float y, float z) {
    if (fabsf(z - c.cz) > c.hz) return false;
    const float sx = x - c.cx, sy = y - c.cy;
    const float lx = sx * c.cosa + sy * (-c.sina);
    const float ly = sx * c.sina + sy * c.cosa;
    return ((double)fabsf(lx) < c.tx) && ((double)fabsf(ly) < c.ty);
}

constexpr int PC_THREADS = 256;
constexpr int PC_MAX_BOXES = 512;

// grid (ceil(P / 256), B): first box (ascending) holding each point, -1 = none
__global__ __launch_bounds__(PC_THREADS) void points_in_boxes_kernel(int boxes_num, int pts_num, const float *__restrict__ boxes,
                                                                     const float *__restrict__ pts, int *__restrict__ out) {
    __shared__ CropBox sb[PC_MAX_BOXES];
    const int bs = blockIdx.y;
    for (int k = threadIdx.x; k < boxes_num; k += PC_THREADS) sb[k] = crop_box(boxes + ((size_t)bs * boxes_num + k) * 7);
    __syncthreads();
    const int p = blockIdx.x * PC_THREADS + threadIdx.x;
    if (p >= pts_num) return;
    const float *pt = pts + ((size_t)bs * pts_num + p) * 3;
    const float x = pt[0], y = pt[1], z = pt[2];
    int hit = -1;
    for (int k = 0; k < boxes_num; ++k)
        if (in_crop_box(sb[k], x, y, z)) { hit = k; break; }
    out[(size_t)bs * pts_num + p] = hit;
}

// grid (M, B): one workgroup per (box, sample).  LDS: idx[S] ints
__global__ __launch_bounds__(PC_THREADS) void roipoint_pool3d_kernel(int pts_num, int boxes_num, int C, int S,
                                                                     const float *__restrict__ xyz, const float *__restrict__ boxes3d,
                                                                     const float *__restrict__ feat, float *__restrict__ pooled,
                                                                     int *__restrict__ empty_flag) {
    extern __shared__ int sel[];              // [S] indices of the selected points
    __shared__ int wave_cnt[PC_THREADS / 64];
    __shared__ int total;
    const int m = blockIdx.x, bs = blockIdx.y;
    const CropBox box = crop_box(boxes3d + ((size_t)bs * boxes_num + m) * 7);
    const float *pts = xyz + (size_t)bs * pts_num * 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    for (int p0 = 0; p0 < pts_num; p0 += PC_THREADS) {
        const int base = total;               // uniform: written behind the barrier at the end of the previous chunk
        if (base >= S) break;
        const int p = p0 + threadIdx.x;
        bool in = false;
        if (p < pts_num) in = in_crop_box(box, pts[(size_t)p * 3], pts[(size_t)p * 3 + 1], pts[(size_t)p * 3 + 2]);
        const unsigned long long mask = __ballot(in);
        const int rank = __popcll(mask & ((1ULL << lane) - 1ULL));
        if (lane == 0) wave_cnt[wave] = __popcll(mask);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (in && off + rank < S) sel[off + rank] = p;          // hits in index order: the first S inside points
        __syncthreads();
        if (threadIdx.x == 0) total = base + wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    const int cnt = min(total, S);
    if (cnt == 0) {
        if (threadIdx.x == 0) empty_flag[(size_t)bs * boxes_num + m] = 1;   // the crop rows stay as the caller zero-filled them
        return;
    }
    for (int k = cnt + threadIdx.x; k < S; k += PC_THREADS) sel[k] = sel[k % cnt];   // k % cnt < cnt: reads original entries only
    __syncthreads();
    const int row = 3 + C;
    float *dst = pooled + (((size_t)bs * boxes_num + m) * S) * row;
    const float *f = feat + (size_t)bs * pts_num * C;
    for (int e = threadIdx.x; e < S * row; e += PC_THREADS) {
        const int s = e / row, j = e - s * row;
        const int src = sel[s];
        dst[e] = j < 3 ? pts[(size_t)src * 3 + j] : f[(size_t)src * C + (j - 3)];
    }
}

}  // namespace mgar

using namespace mgar;

#define PC_API extern "C" __attribute__((visibility("default")))

PC_API int mgar_points_in_boxes(int batch_size, int boxes_num, int pts_num, const float *boxes, const float *pts,
                                int *box_idx_of_points, void *stream) {
    MGAR_REQUIRE(batch_size >= 0 && boxes_num >= 0 && pts_num >= 0, "points_in_boxes: negative size");
    if (boxes_num > PC_MAX_BOXES || batch_size > 65535) {
        set_error("points_in_boxes: at most 512 boxes per sample, 65535 samples");
        return MGAR_EUNSUPPORTED;
    }
    if ((long long)batch_size * pts_num == 0) return MGAR_OK;
    MGAR_REQUIRE(pts && box_idx_of_points && (boxes || boxes_num == 0), "points_in_boxes: null pointer");
    // points read once, boxes once, one int per point written; batch * pts * boxes box tests
    KtScope kt(KT_POINTS_IN_BOXES, (hipStream_t)stream, (double)batch_size * (16.0 * pts_num + 28.0 * boxes_num));
    hipLaunchKernelGGL(points_in_boxes_kernel, dim3(ceil_div(pts_num, PC_THREADS), batch_size), dim3(PC_THREADS), 0, (hipStream_t)stream,
                       boxes_num, pts_num, boxes, pts, box_idx_of_points);
    return check_launch("points_in_boxes: launch failed");
}

PC_API int mgar_roipoint_pool3d_fwd(int batch_size, int pts_num, int boxes_num, int feature_in_len, int sampled_pts_num,
                                    const float *xyz, const float *boxes3d, const float *pts_feature, float *pooled_features,
                                    int *pooled_empty_flag, void *stream) {
    MGAR_REQUIRE(batch_size >= 0 && pts_num >= 0 && boxes_num >= 0 && feature_in_len >= 0 && sampled_pts_num >= 1,
                 "roipoint_pool3d_fwd: bad sizes");
    if (sampled_pts_num > 32768 || batch_size > 65535) {
        set_error("roipoint_pool3d_fwd: at most 32768 sampled points per box, 65535 samples");
        return MGAR_EUNSUPPORTED;
    }
    if ((long long)batch_size * boxes_num == 0) return MGAR_OK;
    MGAR_REQUIRE(xyz && boxes3d && pooled_features && pooled_empty_flag && (pts_feature || feature_in_len == 0),
                 "roipoint_pool3d_fwd: null pointer");
    const size_t lds = (size_t)sampled_pts_num * sizeof(int);
    static size_t attr_lds = 0;
    if (lds > 65536 && lds > attr_lds) {
        (void)hipFuncSetAttribute((const void *)roipoint_pool3d_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    // every (box, sample) workgroup scans its cloud (12 B per point, cache-resident across the boxes of a sample: counted once per
    // sample) and writes its crop
    KtScope kt(KT_ROIPOINT_POOL, (hipStream_t)stream, (double)batch_size * (12.0 * pts_num + 28.0 * boxes_num +
                                                          (double)boxes_num * sampled_pts_num * 4.0 * (3 + feature_in_len) * 2.0));
    hipLaunchKernelGGL(roipoint_pool3d_kernel, dim3(boxes_num, batch_size), dim3(PC_THREADS), lds, (hipStream_t)stream, pts_num, boxes_num,
                       feature_in_len, sampled_pts_num, xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag);
    return check_launch("roipoint_pool3d_fwd: launch failed");
}
