/*
 * mgar_ops.h -- C ABI of libmgar_hip.so, the MI355X (gfx950) implementation of the
 * MGAR-net hot-path operators.
 *
 * This is the drop-in boundary: every entry point below replaces one function of the
 * reference's pybind11 tables (file:line cited per function, relative to
 * /root/reference/) or one third-party op the reference calls on the hot path.
 * INTEGRATION.md shows the reference-side binding for each.
 *
 * Conventions:
 *   - plain C, raw DEVICE pointers + sizes, no torch / hip types in the signatures
 *     (`stream` is a hipStream_t passed as void*; NULL = the default stream);
 *   - the CALLER allocates every output and scratch buffer, exactly as the reference's
 *     Python wrappers do; the library allocates nothing, keeps nothing, never
 *     synchronises the host, and is re-entrant;
 *   - buffers are contiguous float32 / int32 on the device the stream belongs to;
 *   - return value: 0 on success, a negative MGAR_E* code otherwise.  The reference
 *     prints and exit(-1)s on failure (e.g. pointnet2_batch/src/ball_query_gpu.cu:68-72);
 *     this library never exits the process.
 */
#ifndef MGAR_OPS_H
#define MGAR_OPS_H

#ifdef __cplusplus
extern "C" {
#endif

#define MGAR_OK 0
#define MGAR_EINVAL (-1)      /* null pointer / negative size / inconsistent arguments      */
#define MGAR_ELAUNCH (-2)     /* hipGetLastError() != hipSuccess after the launch            */
#define MGAR_EUNSUPPORTED (-3)/* size outside what the kernel was built for (e.g. nsample)   */

#define MGAR_MAX_NSAMPLE 128   /* ball/voxel query: rows are staged in LDS                   */

/* Library identity: ABI version (bumped on any signature change) and a static
 * description string of the last error on the calling thread. */
int mgar_abi_version(void);
const char *mgar_last_error(void);

/* Optional per-kernel timing for roofline reports (bench.py): while enabled, each instrumented
 * launcher brackets its main kernel with two HIP events on the launch stream and notes the launch's
 * algorithmic bytes / flops.  mgar_ktimer_read waits for the events of kernel
 * `id` (0 <= id < mgar_ktimer_count()) and returns totals since the last reset. */
int mgar_ktimer_enable(int on);
int mgar_ktimer_count(void);
const char *mgar_ktimer_name(int id);
int mgar_ktimer_read(int id, double *total_ms, long long *launches, double *total_bytes, double *total_flops,
                     int reset);
/* adds `flops` to kernel id's total while the timers are on: for stacked-layout launches whose per-sample counts
 * live on the device (the library cannot know sum_i M_i * N_i), the instrumenting caller supplies the work */
int mgar_ktimer_add_flops(int id, double flops);
int mgar_ktimer_add_bytes(int id, double bytes);   /* the same for algorithmic bytes (sparse convolutions: pair counts live with the caller) */

/* ======================= pointnet2_batch: (B, N, 3) / (B, C, N) ======================= */

/* ball_query_wrapper   pcdet/ops/pointnet2/pointnet2_batch/src/pointnet2_api.cpp:11
 * kernel               pointnet2_batch/src/ball_query_gpu.cu:15-51
 * new_xyz (b,m,3), xyz (b,n,3) -> idx (b,m,nsample).  First nsample indices k (ascending)
 * with d2 < radius^2, row padded with the first hit; rows of empty balls are NOT written
 * (the caller zero-fills idx, pointnet2_batch/pointnet2_utils.py:218). */
int mgar_ball_query_batch(int b, int n, int m, float radius, int nsample,
                          const float *new_xyz, const float *xyz, int *idx, void *stream);

/* Multi-scale grouping: nr (2..4) (radius, nsample) pairs against the same centres and cloud in ONE
 * scan (each squared distance is computed once); idx[r] receives exactly what the single-radius entry
 * point writes for (radii[r], nsamples[r]).  radii / nsamples / idx are HOST arrays of nr entries (idx:
 * device pointers).  Stack variant: rows of empty balls start with -1 as in mgar_ball_query_stack. */
int mgar_ball_query_multi_batch(int b, int n, int m, int nr, const float *radii, const int *nsamples,
                                const float *new_xyz, const float *xyz, int *const *idx, void *stream);
int mgar_ball_query_multi_stack(int B, int M, int nr, const float *radii, const int *nsamples, const float *new_xyz,
                                const int *new_xyz_batch_cnt, const float *xyz, const int *xyz_batch_cnt,
                                int *const *idx, void *stream);

/* group_points_wrapper / group_points_grad_wrapper   pointnet2_api.cpp:13-14
 * kernels              pointnet2_batch/src/group_points_gpu.cu:53-72, :14-31
 * points (b,c,n), idx (b,npoints,nsample) -> out (b,c,npoints,nsample);
 * grad: grad_out (b,c,npoints,nsample) accumulated into caller-zeroed grad_points (b,c,n). */
int mgar_group_points_batch(int b, int c, int n, int npoints, int nsample,
                            const float *points, const int *idx, float *out, void *stream);
int mgar_group_points_grad_batch(int b, int c, int n, int npoints, int nsample,
                                 const float *grad_out, const int *idx, float *grad_points, void *stream);

/* gather_points_wrapper / gather_points_grad_wrapper   pointnet2_api.cpp:16-17
 * kernels              pointnet2_batch/src/sampling_gpu.cu:15-31, :53-70 */
int mgar_gather_points_batch(int b, int c, int n, int npoints,
                             const float *points, const int *idx, float *out, void *stream);
int mgar_gather_points_grad_batch(int b, int c, int n, int npoints,
                                  const float *grad_out, const int *idx, float *grad_points, void *stream);

/* farthest_point_sampling_wrapper   pointnet2_api.cpp:19
 * kernel               pointnet2_batch/src/sampling_gpu.cu:101-216 (launcher :218-259)
 * points (b,n,3), temp (b,n) pre-filled by the caller (1e10) -> idx (b,m).
 * temp is updated in place to the final min-distances, like the reference.
 * Ties are resolved exactly as the reference's block_size = min(2^floor(log2 n),1024)
 * strided scan + shared-memory tree would (FPS tie rule). */
int mgar_fps_batch(int b, int n, int m, const float *points, float *temp, int *idx, void *stream);

/* The same sampling with spatial pruning (same indices, bit for bit): `perm` (b,n) lists every cloud's
 * point indices in a spatially coherent order -- any permutation is correct, a Morton order is fast:
 * the 64*ceil(n/1024) points of a wave then form a compact cluster and the wave skips the distance
 * update of a round whenever the new sample is farther from its bounding box than its current
 * maximum distance.  mgar_morton_codes writes 30-bit Morton codes (b,n) for the caller to sort.
 * Needs 1024 <= n <= 16384 (MGAR_EUNSUPPORTED otherwise: use mgar_fps_batch). */
int mgar_morton_codes(int b, int n, const float *points, int *codes, void *stream);
int mgar_fps_batch_perm(int b, int n, int m, const float *points, float *temp, const int *perm, int *idx,
                        void *stream);
/* The same for clouds of 16 385 .. 65 536 points (BASELINE config c5), which do not fit the register file: pruning at the
 * granularity of 256-point units; the running minima stay in registers, the coordinates are re-ordered once into `workspace`
 * (mgar_fps_batch_buckets_workspace_floats(b, n) floats, 16-byte aligned) and only the units a new sample can still lower are
 * re-read.  Same indices and final temp as mgar_fps_batch, bit for bit, for any permutation. */
long long mgar_fps_batch_buckets_workspace_floats(int b, int n);
int mgar_fps_batch_buckets(int b, int n, int m, const float *points, float *temp, const int *perm, float *workspace, int *idx,
                           void *stream);

/* Ball query through a uniform cell grid (csrc/ball_query_grid.hip; round 3): the rows of mgar_ball_query_batch / _stack, bit
 * for bit, for queries that rarely fill their balls (the RoI-grid points: every scan of the plain kernel runs to the end of
 * the cloud).  mgar_point_grid_build bins the points of B clouds once (batch layout: n_batch > 0, xyz_batch_cnt NULL; stack
 * layout: n_batch 0, xyz_batch_cnt (B) on the device; n_total = rows of xyz; cell = cell edge, enlarged per cloud until the
 * grid has at most 32 768 cells) into `workspace` (mgar_point_grid_workspace_bytes(B, n_total) bytes, 16-byte aligned), which
 * the queries take as `grid` for as long as xyz is unchanged.  nsample <= 64 (MGAR_EUNSUPPORTED above: use the scan kernels). */
long long mgar_point_grid_workspace_bytes(int B, long long n_total);
int mgar_point_grid_build(int B, int n_batch, long long n_total, const float *xyz, const int *xyz_batch_cnt, float cell,
                          void *workspace, void *stream);
int mgar_ball_query_grid_batch(int b, int n, int m, float radius, int nsample, const float *new_xyz, const void *grid, int *idx,
                               void *stream);
int mgar_ball_query_grid_stack(int B, int M, long long n_total, float radius, int nsample, const float *new_xyz,
                               const int *new_xyz_batch_cnt, const void *grid, int *idx, void *stream);
/* three_nn of mgar_three_nn_batch / _stack (same dist2 / idx, bit for bit -- the reference's strict-'<' cascade keeps the three
 * smallest (d2, index) pairs) through a grid built over the KNOWN points with mgar_point_grid_build (cell <= 0: the library sizes
 * the cells at about -cell, default 4, points each): shells of cells around the query until the third neighbour is closer than
 * anything unvisited.  m_total = rows of `known`. */
int mgar_three_nn_grid_batch(int b, int n, int m, const float *unknown, const void *grid, float *dist2, int *idx, void *stream);
int mgar_three_nn_grid_stack(int B, int N, long long m_total, const float *unknown, const int *unknown_batch_cnt, const void *grid,
                             float *dist2, int *idx, void *stream);

/* three_nn_wrapper   pointnet2_api.cpp:21;  kernel interpolate_gpu.cu:16-59
 * unknown (b,n,3), known (b,m,3) -> dist2 (b,n,3) squared distances, idx (b,n,3). */
int mgar_three_nn_batch(int b, int n, int m, const float *unknown, const float *known,
                        float *dist2, int *idx, void *stream);

/* three_interpolate_wrapper / _grad_wrapper   pointnet2_api.cpp:22-23
 * kernels              pointnet2_batch/src/interpolate_gpu.cu:84-104, :127-149
 * points (b,c,m), idx/weight (b,n,3) -> out (b,c,n); grad accumulates into grad_points (b,c,m). */
int mgar_three_interpolate_batch(int b, int c, int m, int n, const float *points, const int *idx,
                                 const float *weight, float *out, void *stream);
int mgar_three_interpolate_grad_batch(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                      const float *weight, float *grad_points, void *stream);

/* three_interpolate backward through an inverted index (no atomics).  `list` holds, cloud after
 * cloud, the 3n (known point j, unknown point u, weight) entries of that cloud sorted by j with a
 * STABLE sort (so ascending (u, k) inside one j; this fixes the summation order), 2 ints per entry:
 * list[2e] = (j << 16) | u, list[2e+1] = bit pattern of the fp32 weight.  Needs n <= 36864 and
 * m <= 65535 (MGAR_EUNSUPPORTED otherwise: use mgar_three_interpolate_grad_batch).  grad_points
 * (b,c,m) must be ZEROED by the caller: known points with entries are overwritten, the others are
 * left alone.  Same result as mgar_three_interpolate_grad_batch on a zeroed buffer up to fp32
 * summation order. */
int mgar_three_interpolate_grad_sorted_batch(int b, int c, int n, int m, const float *grad_out, const int *list,
                                             float *grad_points, void *stream);
/* Forward with out a CHANNEL SLICE of a wider (b, c_total, n) tensor (samples out_bstride >= c * n elements apart): the
 * interpolated half of the decoder's torch.cat([interpolated, skip]) is written where the concatenation wants it. */
int mgar_three_interpolate_batch_into(int b, int c, int m, int n, const float *points, const int *idx, const float *weight,
                                      float *out, long long out_bstride, void *stream);
/* out (b, c, n) += three_interpolate(points, idx, weight): the caller pre-fills out.  "Project, then interpolate": the first layer of
 * a feature-propagation MLP is linear and so is the interpolation, W [interp(f) ; skip] = interp(W_a f) + W_b skip
 * (pointnet2_batch/pointnet2_modules.py:139-150) -- the known features are projected on the coarse level first. */
int mgar_three_interpolate_batch_add(int b, int c, int m, int n, const float *points, const int *idx, const float *weight,
                                     float *out, void *stream);
/* Both with grad_out a CHANNEL SLICE of a wider (b, c_total, n) tensor: consecutive samples are grad_out_bstride >= c * n
 * elements apart.  The decoder (reference PointnetFPModule, pointnet2_batch/pointnet2_modules.py:139-148) concatenates the
 * interpolated features with the skip features; the gradient of that torch.cat hands this op a slice, which the reference's
 * wrapper first copies (grad_out.contiguous()): up to 2 GB per launch at config c3. */
int mgar_three_interpolate_grad_batch_strided(int b, int c, int n, int m, const float *grad_out, long long grad_out_bstride,
                                              const int *idx, const float *weight, float *grad_points, void *stream);
int mgar_three_interpolate_grad_sorted_batch_strided(int b, int c, int n, int m, const float *grad_out,
                                                     long long grad_out_bstride, const int *list, float *grad_points,
                                                     void *stream);

/* ============== pointnet2_stack: (N1+N2+..., 3|C) + per-sample counts ================= */

/* ball_query_wrapper   pcdet/ops/pointnet2/pointnet2_stack/src/pointnet2_api.cpp:13
 * kernel               pointnet2_stack/src/ball_query_gpu.cu:16-66
 * As the batch version, local indices; an empty ball writes idx[row][0] = -1 only. */
int mgar_ball_query_stack(int B, int M, float radius, int nsample,
                          const float *new_xyz, const int *new_xyz_batch_cnt,
                          const float *xyz, const int *xyz_batch_cnt, int *idx, void *stream);

/* voxel_query_wrapper   pointnet2_stack/src/pointnet2_api.cpp:14
 * kernel               pointnet2_stack/src/voxel_query_gpu.cu:10-89
 * new_coords (M,4) [b,z,y,x]; point_indices (B,R1,R2,R3) of global row ids or -1.
 * Cells are visited dz,dy,dx ascending; accept d2 <= radius^2; idx[row][0] = -1 if none. */
int mgar_voxel_query_stack(int M, int R1, int R2, int R3, int nsample, float radius,
                           int z_range, int y_range, int x_range,
                           const float *new_xyz, const float *xyz, const int *new_coords,
                           const int *point_indices, int *idx, void *stream);

/* The same query with the voxel -> row lookups served by the hash table of mgar_voxel_hash_build (below) instead of the
 * dense (B, R1, R2, R3) table of generate_voxel2pinds (pcdet/utils/common_utils.py:244-252; 80 MB per sample at the
 * shipped grid): same cells in the same order, identical idx. */
int mgar_voxel_query_hash_stack(int M, int R1, int R2, int R3, int nsample, float radius,
                                int z_range, int y_range, int x_range,
                                const float *new_xyz, const float *xyz, const int *new_coords,
                                const long long *table_keys, const int *table_vals, int capacity, int *idx,
                                void *stream);

/* stack_farthest_point_sampling_wrapper   pointnet2_stack/src/pointnet2_api.cpp:17
 * kernel               pointnet2_stack/src/sampling_gpu.cu:188-319 (always 1024 threads)
 * points (N,3), temp (N), xyz_batch_cnt (batch_size), num_sampled_points (batch_size)
 * -> idx (sum m_i) GLOBAL row ids. */
int mgar_fps_stack(int batch_size, int N, const float *points, float *temp, const int *xyz_batch_cnt,
                   int *idx, const int *num_sampled_points, void *stream);

/* group_points_wrapper / group_points_grad_wrapper   pointnet2_stack/src/pointnet2_api.cpp:19-20
 * kernels              pointnet2_stack/src/group_points_gpu.cu:71-102, :15-45
 * features (N,C) row-major, idx (M,nsample) local -> out (M,C,nsample). */
int mgar_group_points_stack(int B, int M, int C, int nsample, const float *features,
                            const int *features_batch_cnt, const int *idx, const int *idx_batch_cnt,
                            float *out, void *stream);
int mgar_group_points_grad_stack(int B, int M, int C, int N, int nsample, const float *grad_out,
                                 const int *idx, const int *idx_batch_cnt, const int *features_batch_cnt,
                                 float *grad_features, void *stream);

/* three_nn_wrapper   pointnet2_stack/src/pointnet2_api.cpp:22; kernel interpolate_gpu.cu:16-75
 * idx are GLOBAL rows of `known`. */
int mgar_three_nn_stack(int batch_size, int N, int M, const float *unknown, const int *unknown_batch_cnt,
                        const float *known, const int *known_batch_cnt, float *dist2, int *idx, void *stream);

/* three_interpolate_wrapper / _grad_wrapper   pointnet2_stack/src/pointnet2_api.cpp:23-24
 * kernels              pointnet2_stack/src/interpolate_gpu.cu:107-126, :151-172
 * features (M,C), idx/weight (N,3) -> out (N,C). */
int mgar_three_interpolate_stack(int N, int C, const float *features, const int *idx, const float *weight,
                                 float *out, void *stream);
int mgar_three_interpolate_grad_stack(int N, int C, const float *grad_out, const int *idx, const float *weight,
                                      float *grad_features, void *stream);

/* ===================== fused query-and-group ==================== */

/* The torch op chain of QueryAndGroup.forward after ball_query
 * (pointnet2_batch/pointnet2_utils.py:241-264: transpose, group xyz, subtract centre, group
 * features, cat) as one kernel.  xyz (b,n,3), new_xyz (b,npoints,3), features (b,c,n) or NULL
 * with c = 0, idx (b,npoints,nsample) -> out (b, 3+c, npoints, nsample): rows 0..2 are the
 * neighbour coordinates relative to the query, rows 3.. the grouped features.
 * bwd: the feature rows of grad_out (b, 3+c, npoints, nsample) are accumulated into the
 * caller-zeroed grad_features (b,c,n); xyz receives no gradient (as in the reference, where
 * xyz never requires grad). */
int mgar_query_group_batch_fwd(int b, int c, int n, int npoints, int nsample, const float *xyz,
                               const float *new_xyz, const float *features, const int *idx, float *out,
                               void *stream);
int mgar_query_group_batch_bwd(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                               const int *idx, float *grad_features, void *stream);

/* Stacked layout (pointnet2_stack/pointnet2_utils.py:123-159 + the permute of
 * pointnet2_stack/pointnet2_modules.py:95): idx (M,nsample) is the RAW output of
 * mgar_ball_query_stack (idx[row][0] == -1 marks an empty ball, whose columns are zero-filled);
 * out is CHANNEL-MAJOR (3+C, M*nsample), the layout the shared MLP's GEMM consumes.
 * bwd accumulates the feature rows of grad_out (3+C, M*nsample) into caller-zeroed
 * grad_features (N,C). */
int mgar_query_group_stack_fwd(int B, int M, int C, int nsample, const float *xyz, const int *xyz_batch_cnt,
                               const float *new_xyz, const int *new_xyz_batch_cnt, const float *features,
                               const int *idx, float *out, void *stream);
int mgar_query_group_stack_bwd(int B, int M, int C, int nsample, const float *grad_out, const int *idx,
                               const int *new_xyz_batch_cnt, const int *xyz_batch_cnt, float *grad_features,
                               void *stream);

/* "Project, then group": the first shared-MLP layer is linear, so
 *   W [rel_xyz ; feat_j] = W_xyz rel_xyz + (W_f feat)_j
 * i.e. the feature half can be applied to the N un-grouped points (zf = W_f features, a small GEMM)
 * and the kernel gathers zf rows and adds the xyz half on the fly.  The (3+C)-channel grouped
 * tensor of QueryAndGroup never exists and the first-layer GEMM shrinks from M*nsample to N columns.
 * zf: (b,c,n) / (N,C) pre-projected features; wx: (c,3) row-major;
 * rel_out (may be NULL): (b,3,npoints,nsample) / (3, M*nsample); y_out: (b,c,npoints,nsample) /
 * (C, M*nsample) = the first layer's pre-BatchNorm output.  bwd scatters grad_y into the
 * caller-zeroed grad_zf; d wx = grad_y . rel^T is a small GEMM left to the caller. */
int mgar_query_group_proj_batch_fwd(int b, int c, int n, int npoints, int nsample, const float *xyz,
                                    const float *new_xyz, const float *zf, const float *wx, const int *idx,
                                    float *rel_out, float *y_out, void *stream);
int mgar_query_group_proj_batch_bwd(int b, int c, int n, int npoints, int nsample, const float *grad_y,
                                    const int *idx, float *grad_zf, void *stream);
/* stack layout: zf / grad_zf rows have leading dimension zf_ld >= C floats, so the projections of
 * several scales (multi-scale grouping) can live side by side in one (N, sum C_k) matrix produced by
 * one GEMM over the shared features. */
int mgar_query_group_proj_stack_fwd(int B, int M, int C, int nsample, const float *xyz, const int *xyz_batch_cnt,
                                    const float *new_xyz, const int *new_xyz_batch_cnt, const float *zf, int zf_ld,
                                    const float *wx, const int *idx, float *rel_out, float *y_out, void *stream);
/* ..._fwd that also leaves the statistics partials of y_out for the BatchNorm that follows: out_stats (C, M*nsample/128, 2),
 * chunk = 128 (M * nsample % 128 == 0). */
int mgar_query_group_proj_stack_fwd_stats(int B, int M, int C, int nsample, const float *xyz, const int *xyz_batch_cnt,
                                          const float *new_xyz, const int *new_xyz_batch_cnt, const float *zf, int zf_ld,
                                          const float *wx, const int *idx, float *rel_out, float *y_out, float *out_stats,
                                          void *stream);
int mgar_query_group_proj_stack_bwd(int B, int M, int C, int nsample, const float *grad_y, const int *idx,
                                    const int *new_xyz_batch_cnt, const int *xyz_batch_cnt, float *grad_zf,
                                    int zf_ld, void *stream);
/* The same gradient WITHOUT atomics and bit-reproducible (replaces the scatter of the reference's
 * group_points_grad_kernel_stack, pcdet/ops/pointnet2/pointnet2_stack/src/group_points_gpu.cu:15-46, for C <= 64):
 *   1. inverse_index: a stable counting sort (histograms in LDS, no global atomics) of the (source row, column) pairs of the raw
 *      ball-query result idx (M, nsample): list = the columns of every source row in ascending order; items = work items
 *      int[4] (row, begin, length <= 256, parts of the row) -- capacity mgar_query_group_stack_inverse_items(), ZERO-FILLED
 *      by the caller; row_item[row] = the row's first item, -1 if no column references it; workspace:
 *      mgar_query_group_stack_inverse_workspace_ints(B, N, M * nsample) ints.  M * nsample < 2^31; at most 262 144 source
 *      rows per sample (the caller guarantees it: the counts live on the device).
 *   2. bwd_rows (workspace = the one inverse_index filled: part of the index): grad_zf[row][0..C) = sum of g_t[col][0..C)
 *      over the row's columns, in list order; g_t is the ROW-MAJOR
 *      gradient (M * nsample, C).  Rows nobody references are not written (hand in zeros).  part_rows: n_items * C floats of
 *      scratch.  wx_part (optional, with xyz / new_xyz): ceil(n_items / 8) * C * 3 floats, ZERO-FILLED; summed over the first
 *      axis they are d wx (C, 3) = sum_col g_t[col] (x) (xyz[row] - new_xyz[col / nsample]) -- the weight gradient of the
 *      relative-coordinate half of the folded first layer, so the forward need not store rel_out for the backward. */
long long mgar_query_group_stack_inverse_items(int B, int N, long long total);
long long mgar_query_group_stack_inverse_workspace_ints(int B, int N, long long total);
int mgar_query_group_stack_inverse_index(int B, int M, int nsample, int N, const int *idx, const int *new_xyz_batch_cnt,
                                         const int *xyz_batch_cnt, int *workspace, int *list, int *items, int *row_item,
                                         void *stream);
int mgar_query_group_stack_bwd_rows(int n_items, int N, int C, int nsample, const int *workspace, const int *items,
                                    const int *row_item, const int *list, const float *g_t, const float *xyz,
                                    const float *new_xyz, float *grad_zf, int zf_ld, float *part_rows, float *wx_part,
                                    long long total, void *stream);

/* Weight gradient of that projection on the exact-fp32 MFMA, stacked (row-major) operands:
 *   dw[o][i] = sum_n a[n*lda + o] * f[n*ldf + i]      a (N,Co) = grad_zf, f (N,Ci) = features
 * dw (Co,Ci) is OVERWRITTEN; partial sums go through `workspace`
 * (mgar_rowmajor_dw_workspace_floats(N,Co,Ci) floats) and are added in a fixed order.
 * Needs Co <= 96, Ci <= 128 (MGAR_EUNSUPPORTED otherwise). */
int mgar_rowmajor_dw_workspace_floats(long long N, int Co, int Ci);
int mgar_rowmajor_dw(const float *a, int lda, const float *f, int ldf, long long N, int Co, int Ci,
                     float *workspace, float *dw, void *stream);

/* ========== fused BatchNorm(train) + ReLU + max-over-nsample ========== */

/* What follows the 1x1 convolution in every shared MLP of the SA / FP / RoI-pool modules:
 * nn.BatchNorm2d -> nn.ReLU [-> F.max_pool2d over nsample]
 * (pointnet2_batch/pointnet2_modules.py:37-45; pointnet2_stack/pointnet2_modules.py:96-104).
 * Activations are x (B, C, P) contiguous, P = npoint*nsample columns.
 * workspace: caller-allocated floats, mgar_bn_workspace_floats(B, C, P) of them.
 *   train_stats : per-channel mean / invstd = 1/sqrt(var_biased + eps) of x; if running_* are
 *                 non-NULL they receive the usual momentum update (unbiased variance) and the int64
 *                 counter num_batches_tracked (may be NULL) is incremented.
 *   act_fwd     : y = [relu](x * gamma*invstd + beta - mean*gamma*invstd)   (gamma/beta may be NULL)
 *   act_maxpool_fwd : out (B,C,M) = max_s [relu](bn(x[b,c,m,s])), arg (B,C,M) uint8 = first arg-max,
 *                 xarg (B,C,M, may be NULL) = x at the arg-max (lets the backward reduction read
 *                 coalesced arrays; act_maxpool_bwd gathers from x when it is NULL)
 *   act_bwd     : dx, dgamma, dbeta of y = [relu](bn_train(x)) given dy (all fully written)
 *   act_maxpool_bwd : the same when y was reduced by act_maxpool_fwd (dpool, pooled, arg) */
int mgar_bn_workspace_floats(int B, int C, int P);
int mgar_bn_train_stats(const float *x, int B, int C, int P, float eps, float momentum, float *workspace,
                        float *mean, float *invstd, float *running_mean, float *running_var,
                        long long *num_batches_tracked, void *stream);
/* Per-sample statistics: each of the G samples of x (G,C,P) is normalised with ITS OWN batch statistics
 * (mean / invstd have G*C entries) -- what the reference computes when it sends G clips through a train-mode
 * BatchNorm one at a time (I3D, model/gat_model.py:1048) -- and the running statistics receive the G momentum
 * updates in sample order.  workspace: mgar_bn_workspace_floats(1, G*C, P) floats.  Forward only. */
int mgar_bn_train_stats_grouped(const float *x, int G, int C, int P, float eps, float momentum, float *workspace,
                                float *mean, float *invstd, float *running_mean, float *running_var,
                                long long *num_batches_tracked, void *stream);
int mgar_bn_act_fwd_grouped(const float *x, int G, int C, int P, const float *mean, const float *invstd,
                            const float *gamma, const float *beta, int relu, float *y, void *stream);
int mgar_bn_act_fwd(const float *x, int B, int C, int P, const float *mean, const float *invstd,
                    const float *gamma, const float *beta, int relu, float *y, void *stream);
/* ---- channels-last (NDHWC) forward kernels for the frozen I3D between its convolutions (csrc/channels_last.hpp) ----
 * MIOpen's composable-kernel convolutions work in NDHWC and wrap every NCDHW call in two transposes; with the activations
 * kept channels-last the BatchNorm3d + ReLU (reference model/backbone.py:61-97), MaxPool3dSamePadding (:99-131) and the
 * Inception concatenation (:227-260) must read / write that layout.  A tensor is (S * R rows, C): S samples, R = T*H*W
 * positions, channel innermost; C % 4 == 0, C <= 1024.  per_sample: every sample normalised with its own statistics
 * (mean / invstd have S * C entries), running statistics updated once per sample in order.
 *   bn_cl_train_stats  workspace: mgar_bn_cl_workspace_floats(S, R, C, per_sample) floats
 *   bn_cl_act_fwd      y[row * ldy + c]: y may point at a COLUMN SLICE of a wider (rows, ldy) tensor
 *   bn_act_fwd_to_cl   the same arithmetic reading NCDHW x (S, C, R) and writing NDHWC (where the layout changes: the stem)
 *   maxpool3d_same_fwd_cl  x (N, T, H, W, C) -> y (N, ceil(T/st), ceil(H/sh), ceil(W/sw), C), TF "same" zero padding */
long long mgar_bn_cl_workspace_floats(int S, int R, int C, int per_sample);
int mgar_bn_cl_train_stats(const float *x, int S, int R, int C, int per_sample, float eps, float momentum, float *workspace,
                           float *mean, float *invstd, float *running_mean, float *running_var,
                           long long *num_batches_tracked, void *stream);
int mgar_bn_cl_act_fwd(const float *x, int S, int R, int C, int per_sample, const float *mean, const float *invstd,
                       const float *gamma, const float *beta, int relu, float *y, int ldy, void *stream);
/* BatchNorm1d(train) [+ ReLU] BACKWARD over row-major x, dy (rows, C) fp32 -- the sparse trunk's (N_active, C) features
 * (pcdet/models/backbones_3d/spconv_backbone.py:8-27 post_act_block: BatchNorm1d + ReLU on .features); forward = the two entry
 * points above with S = 1.  dx (rows, C), dgamma / dbeta (C) fully written.  workspace: mgar_bn_rows_bwd_workspace_floats. */
long long mgar_bn_rows_bwd_workspace_floats(int rows, int C);
int mgar_bn_rows_bwd(const float *dy, const float *x, int rows, int C, const float *mean, const float *invstd, const float *gamma,
                     const float *beta, int relu, float *workspace, float *dgamma, float *dbeta, float *dx, void *stream);
int mgar_bn_act_fwd_to_cl(const float *x, int S, int C, int R, int per_sample, const float *mean, const float *invstd,
                          const float *gamma, const float *beta, int relu, float *y, int ldy, void *stream);
int mgar_maxpool3d_same_fwd_cl(const float *x, int N, int T, int H, int W, int C, int kt, int kh, int kw, int st, int sh, int sw,
                               float *y, void *stream);
/* Training statistics + apply [+ ReLU] in ONE launch for small channels: at most 16 384 elements per channel (B * P, or P with
 * per_sample statistics) and P % 4 == 0, MGAR_EUNSUPPORTED otherwise.  What mgar_bn_train_stats[_grouped] followed by
 * mgar_bn_act_fwd_into computes (forward only; the launch-bound one-clip step has ~45 such BatchNorms in the I3D).
 * mean / invstd: optional outputs (C, or B * C with per_sample); workspace: B * C floats, needed with per_sample + running
 * statistics; y_bstride < 0: y contiguous. */
int mgar_bn_act_small(const float *x, int B, int C, int P, int per_sample, float eps, float momentum, const float *gamma,
                      const float *beta, int relu, float *workspace, float *mean, float *invstd, float *running_mean,
                      float *running_var, long long *num_batches_tracked, float *y, long long y_bstride, void *stream);
/* bn_act_fwd (stats_per_sample = 0) / bn_act_fwd_grouped (1) with y a CHANNEL SLICE of a wider (B, C_total, P) tensor:
 * consecutive samples of y are y_bstride >= C * P elements apart.  The branches of an Inception module (reference
 * model/backbone.py:227-260: torch.cat of four branch outputs) write straight into the concatenated tensor. */
int mgar_bn_act_fwd_into(const float *x, int B, int C, int P, const float *mean, const float *invstd, const float *gamma,
                         const float *beta, int relu, int stats_per_sample, float *y, long long y_bstride, void *stream);
int mgar_bn_act_maxpool_fwd(const float *x, int B, int C, int M, int nsample, const float *mean,
                            const float *invstd, const float *gamma, const float *beta, int relu, float *out,
                            unsigned char *arg, float *xarg, void *stream);
int mgar_bn_act_bwd(const float *dy, const float *x, int B, int C, int P, const float *mean, const float *invstd,
                    const float *gamma, const float *beta, int relu, float *workspace, float *dgamma,
                    float *dbeta, float *dx, void *stream);
/* BatchNorm training statistics from partials a PRODUCER kernel left behind (mgar_pointwise_conv_fwd_stats,
 * mgar_query_group_proj_stack_fwd_stats): partial (C, nchunk, 2) = per (channel, chunk of `chunk` consecutive elements of the
 * channel in (b, p) order) the chunk's mean and sum of squared deviations; n = elements per channel.  The finalize of
 * mgar_bn_train_stats (merge in double, running statistics) without its pass over the tensor (4 * B * C * P bytes). */
long long mgar_bn_stats_from_partials_workspace_floats(int nchunk, int C);
int mgar_bn_stats_from_partials(const float *partial, int nchunk, int C, long long n, int chunk, float eps, float momentum,
                                float *workspace, float *mean, float *invstd, float *running_mean, float *running_var,
                                long long *num_batches_tracked, void *stream);
/* mgar_bn_act_maxpool_bwd with dpool read IN PLACE from a strided tensor: element (b, c, m) at dpool[b*sb + c*sc + m*sm]
 * -- a channel slice of a wider (B, C_total, M) tensor (gradient of the torch.cat over the scales of an SA module,
 * reference pointnet2_modules.py:55) or the transposed view of (M, C_total) rows -- instead of a copy first. */
int mgar_bn_act_maxpool_bwd_strided(const float *dpool, long long sb, long long sc, long long sm, const float *pooled,
                                    const unsigned char *arg, const float *x, const float *xarg, int B, int C, int M,
                                    int nsample, const float *mean, const float *invstd, const float *gamma, int relu,
                                    float *workspace, float *dgamma, float *dbeta, float *dx, void *stream);
/* mgar_bn_act_bwd without its reduction: coef (2 C) = per channel {mean dz, mean dz xhat} from mgar_pointwise_conv_dw_bnbwd;
 * rowmajor != 0: dx_t (B * P, C) as mgar_bn_act_bwd_rowmajor (C <= 64). */
int mgar_bn_act_bwd_apply(const float *dy, const float *x, int B, int C, int P, const float *mean, const float *invstd,
                          const float *gamma, const float *beta, int relu, const float *coef, int rowmajor, float *dx,
                          void *stream);
/* mgar_bn_act_bwd with the input gradient written ROW-MAJOR, dx_t (B*P, C), C <= 64: the layout the atomic-free stack
 * grouping backward (mgar_query_group_stack_bwd_rows) gathers from. */
int mgar_bn_act_bwd_rowmajor(const float *dy, const float *x, int B, int C, int P, const float *mean, const float *invstd,
                             const float *gamma, const float *beta, int relu, float *workspace, float *dgamma,
                             float *dbeta, float *dx_t, void *stream);
int mgar_bn_act_maxpool_bwd(const float *dpool, const float *pooled, const unsigned char *arg, const float *x,
                            const float *xarg, int B, int C, int M, int nsample, const float *mean, const float *invstd,
                            const float *gamma, int relu, float *workspace, float *dgamma, float *dbeta,
                            float *dx, void *stream);

/* Weight gradient of a point-wise convolution on the exact-fp32 MFMA:
 *   dw[o][i] = sum_b sum_p dy[b,o,p] * x[b,i,p]       x (B,Cin,P), dy (B,Cout,P), dw (Cout,Cin)
 * (backward-weights of the Conv2d 1x1 layers of the shared MLPs, pointnet2_batch/
 * pointnet2_modules.py:86-92).  dw is OVERWRITTEN.  Workgroup partials go through `workspace`
 * (mgar_pointwise_dw_workspace_floats(B,Cin,Cout,P) floats) and are summed in a fixed order, so the
 * result is reproducible run to run. */
int mgar_pointwise_dw_workspace_floats(int B, int Cin, int Cout, int P);
int mgar_pointwise_conv_dw(const float *x, const float *dy, int B, int Cin, int Cout, int P, float *workspace,
                           float *dw, void *stream);

/* The same with the X operand activated on the fly: x := relu?(bn(x)) with the given BatchNorm
 * statistics / affine (in_gamma, in_beta may be NULL = 1, 0; in_mean NULL = no activation), i.e. X
 * is the PRE-BN output of the previous layer. */
int mgar_pointwise_conv_dw_act(const float *x, const float *dy, int B, int Cin, int Cout, int P,
                               const float *in_mean, const float *in_invstd, const float *in_gamma,
                               const float *in_beta, int in_relu, float *workspace, float *dw, void *stream);

/* One [BatchNorm -> ReLU -> Conv 1x1] step of the shared MLPs (pointnet2_batch/pointnet2_modules.py:
 * 86-92, pointnet2_stack/pointnet2_modules.py:33-40) without materialising the activated tensor:
 *   y[b,o,p] = sum_i w[o*w_row_stride + i*w_col_stride] * act_i(x[b,i,p])
 *   act_i(v) = relu?(v * sc_i + sh_i), sc_i = in_invstd[i]*in_gamma[i], sh_i = in_beta[i] - in_mean[i]*sc_i
 * (in_mean NULL = identity; in_gamma / in_beta NULL = 1 / 0).  With w read transposed (strides
 * swapped) and no activation it is the data gradient dX = W^T dY.  x (B,Cin,P), y (B,Cout,P).
 * Needs Cout <= 64, 1 <= Cin <= 256, P % 4 == 0 (MGAR_EUNSUPPORTED otherwise). */
int mgar_pointwise_conv_fwd(const float *x, int B, int Cin, int P, const float *w, int w_row_stride,
                            int w_col_stride, int Cout, const float *in_mean, const float *in_invstd,
                            const float *in_gamma, const float *in_beta, int in_relu, float *y, void *stream);
/* Weight gradient of [BatchNorm -> ReLU -> conv 1x1] AND the reduction of that BatchNorm's backward in ONE pass over x (the
 * layer's pre-BatchNorm input, B x Cin x P) and dy (gradient of the conv output): the kernel accumulates, on the matrix cores,
 * A[o][i] = sum dy m_i and B[o][i] = sum dy m_i xhat_i (m = ReLU mask, xhat = normalised x); then dW = gamma B + beta A,
 * dbeta_i = sum_o W A, dgamma_i = sum_o W B, coef = {dbeta, dgamma} / (B * P) for mgar_bn_act_bwd_apply.  Replaces
 * mgar_pointwise_conv_dw_act + the reduction pass of mgar_bn_act_bwd (8 * B * Cin * P bytes).  Cin, Cout <= 64. */
int mgar_pointwise_dw_bnbwd_workspace_floats(int B, int Cin, int Cout, int P);
int mgar_pointwise_conv_dw_bnbwd(const float *x, const float *dy, const float *w, int B, int Cin, int Cout, int P,
                                 const float *in_mean, const float *in_invstd, const float *in_gamma, const float *in_beta,
                                 int in_relu, float *workspace, float *dw, float *dgamma, float *dbeta, float *coef, void *stream);
/* The same, also leaving the statistics partials of y for the BatchNorm that follows: out_stats (Cout, B * P / 128, 2),
 * chunk = 128 (mgar_bn_stats_from_partials).  P % 128 == 0, Cout <= 32 (MGAR_EUNSUPPORTED otherwise). */
int mgar_pointwise_conv_fwd_stats(const float *x, int B, int Cin, int P, const float *w, int w_row_stride, int w_col_stride,
                                  int Cout, const float *in_mean, const float *in_invstd, const float *in_gamma,
                                  const float *in_beta, int in_relu, float *y, float *out_stats, void *stream);

/* MaxPool3dSamePadding.forward of the reference's I3D (model/backbone.py:99-131): zero "same"
 * padding + max pooling without materialising the padded tensor.  x (NC, T, H, W) -> y (NC, ceil(T/st),
 * ceil(H/sh), ceil(W/sw)).  Forward only (I3D is frozen in MGAR-net). */
int mgar_maxpool3d_same_fwd(const float *x, int NC, int T, int H, int W, int kt, int kh, int kw, int st, int sh,
                            int sw, float *y, void *stream);
/* The same windows with the maximum taken over their VALID elements only (the padding does not take part): for pooling a
 * PRE-BatchNorm tensor.  With gamma > 0 in every channel, maxpool_same(relu(bn(x))) == relu(bn(maxpool_valid(x))) bit for bit
 * (monotone per channel, relu >= 0 absorbs the zero padding), so a Unit3D that feeds a MaxPool3dSamePadding
 * (model/backbone.py:305-313: Conv3d_1a_7x7 -> MaxPool3d_2a_3x3, Conv3d_2c_3x3 -> MaxPool3d_3a_3x3) normalises the pooled
 * tensor instead of the full one. */
int mgar_maxpool3d_valid_fwd(const float *x, int NC, int T, int H, int W, int kt, int kh, int kw, int st, int sh, int sw, float *y,
                             void *stream);

/* The first convolution of Inception-I3D (model/backbone.py:305-307 ``Conv3d_1a_7x7``: Unit3D(3 -> 64, kernel [7,7,7],
 * stride (2,2,2), TF "same" padding :168-172, no bias) as a direct implicit GEMM on the fp32 MFMA: padding handled in the
 * kernel (no padded copy), NCDHW in and out (no layout transposes).  x (N, 3, T, H, W), w (64, 3, 7, 7, 7) ->
 * y (N, 64, ceil(T/2), ceil(H/2), ceil(W/2)).  w_packed: caller-allocated scratch of
 * mgar_stem_conv3d_workspace_floats() floats.  Forward only (I3D is frozen in MGAR-net). */
int mgar_stem_conv3d_workspace_floats(void);
int mgar_stem_conv3d_fwd(const float *x, int N, int T, int H, int W, const float *w, float *w_packed, float *y,
                         void *stream);
/* fp32 with W % 4 == 0 runs the minimal-filtering variant (the stride-2 row convolution split by column parity into a
 * 3-tap and a 4-tap stride-1 convolution, Winograd F(2, 3) on the 3-tap parts: 10 instead of 14 multiply-accumulates per
 * (kt, c, kh) and output pair); 0 switches it off (A/B tests), default 1. */
int mgar_stem_conv3d_set_minimal_filtering(int on);

/* The 3x3x3, stride-1, "same"-padded convolutions of Inception-I3D (model/backbone.py:311-312 ``Conv3d_2c_3x3`` and the
 * ``Conv3d_0b_3x3`` units of every InceptionModule :215-236; Unit3D :134-206 = pad + nn.Conv3d(bias=False)) on the fp32 MFMA
 * with the Winograd F(2, 3) identity along W (2/3 of the direct convolution's multiply-accumulates), NCDHW in and out,
 * padding handled in the kernel.  x (N, Cin, D, H, W), w (Cout, Cin, 3, 3, 3) -> y (N, Cout, D, H, W).  Cin and W must be
 * even.  w_packed: caller-allocated scratch of mgar_conv3d_k3_workspace_floats(Cin, Cout) floats (the transformed filter,
 * rewritten on every call).  Forward only (I3D is frozen in MGAR-net). */
long long mgar_conv3d_k3_workspace_floats(int Cin, int Cout);
int mgar_conv3d_k3_set_lds_pad(int bytes);   /* diagnostics: extra dynamic LDS per workgroup (occupancy experiments), default 0 */
int mgar_conv3d_k3_fwd(const float *x, int N, int Cin, int D, int H, int W, const float *w, int Cout, float *w_packed, float *y,
                       void *stream);

/* A timed gap on a stream: one wave sleeping `us` (0 .. 1000) microseconds.  Scheduling aid of the two-stream clip model
 * (workload.ClipModel.forward): the RGB side stream starts a few microseconds after the level-1 FPS launch. */
int mgar_delay_us(int us, void *stream);

/* ===================== third-party ops on the hot path ================================ */

/* torchvision.ops.roi_align (call site model/gat_model.py:1056-1057, sg_model.py:96-97).
 * input (N,C,H,W), rois (K,5) [batch_index,x1,y1,x2,y2] -> out (K,C,ph,pw).
 * sampling_ratio <= 0 = adaptive ceil(roi/pooled); aligned = 0 is the reference's mode.
 * bwd accumulates into caller-zeroed grad_input (N,C,H,W). */
int mgar_roi_align_fwd(const float *input, int N, int C, int H, int W, const float *rois, int K,
                       int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio, int aligned,
                       float *out, void *stream);
int mgar_roi_align_bwd(const float *grad_out, int N, int C, int H, int W, const float *rois, int K,
                       int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio, int aligned,
                       float *grad_input, void *stream);

/* DAFM distance-aware attention core (model/gat_model.py:487-491 and :503-505), batched
 * over S scenes of n_s <= MGAR_DAFM_MAX_N actors each (scene_off: (S+1) row offsets).
 *   E   = softmax(-De/sigma, dim=1)
 *   Att = softmax((Q K^T * E) * scale, dim=1)
 *   out = Att V
 * q,k,v,out: (rows, D) row-major, D a multiple of 64; de: per scene a dense (n_s, n_s)
 * block stored at de + de_off[s].  att (same layout as de) is saved for the backward.
 * total_rows = scene_off[S] is passed by the host (it sizes the grid). */
#define MGAR_DAFM_MAX_N 128
int mgar_dafm_attn_fwd(int S, int total_rows, int D, const int *scene_off, const int *de_off, const float *q, const float *k,
                       const float *v, const float *de, float sigma, float scale, float *att, float *out,
                       void *stream);
/* bwd: gmat is caller-allocated scratch with the layout of att (it receives
 * dL/d(QK^T)); grad_q / grad_k / grad_v are fully written (no need to zero them). */
int mgar_dafm_attn_bwd(int S, int total_rows, int D, const int *scene_off, const int *de_off, const float *q, const float *k,
                       const float *v, const float *de, float sigma, float scale, const float *att,
                       const float *grad_out, float *gmat, float *grad_q, float *grad_k, float *grad_v,
                       void *stream);

/* GATv2 edge-softmax + aggregate (torch_geometric.nn.GATv2Conv arithmetic; call site
 * model/gat_model.py:1082-1094).  Edges are given in CSR form grouped by TARGET node:
 * rowptr (n_nodes+1), col (E) = source node of each incoming edge (self loops included
 * by the caller).  xl, xr: (n_nodes, H*C) = W_l x, W_r x;  att: (H, C).
 *   e_ij   = att_h . leaky_relu(xl_j + xr_i, slope)
 *   alpha  = softmax_j(e_ij)          (saved, (E,H))
 *   out_i  = sum_j alpha_ij * xl_j    (n_nodes, H*C)
 * edge_scale (E,H), may be NULL: multiplies alpha in the aggregation only (PyG applies
 * dropout to alpha in training mode: pass mask/(1-p)); alpha is saved un-scaled.
 * bwd (round 3: no float atomics, every sum in a fixed order -> bit-reproducible gradients): besides the by-target CSR it
 * takes the same edges indexed BY SOURCE node -- src_rowptr (n_nodes+1), src_edge (E) = edge id (position in col / alpha) of
 * every outgoing edge of a node, ascending, src_dst (E) = that edge's target node -- and a workspace of
 * mgar_gatv2_bwd_workspace_floats(n_nodes, H, C, E) floats.  grad_xl, grad_xr, grad_att are fully written.
 */
/* C must be a multiple of 64 (lanes run along the channel axis). */
int mgar_gatv2_fwd(int n_nodes, int H, int C, const int *rowptr, const int *col, const float *xl,
                   const float *xr, const float *att, float slope, const float *edge_scale, float *alpha,
                   float *out, void *stream);
int mgar_gatv2_bwd(int n_nodes, int H, int C, const int *rowptr, const int *col, const int *src_rowptr, const int *src_edge,
                   const int *src_dst, const float *xl, const float *xr, const float *att, float slope, const float *edge_scale,
                   const float *alpha, const float *grad_out, float *workspace, float *grad_xl, float *grad_xr, float *grad_att,
                   void *stream);
long long mgar_gatv2_bwd_workspace_floats(int n_nodes, int H, int C, int n_edges);

/* ===================== fused Voxel-RoI pooling ====================
 * One scale of NeighborVoxelSAModuleMSG between mlps_in and mlps_out
 * (pcdet/ops/pointnet2/pointnet2_stack/voxel_pool_modules.py:86-126; queries from voxel_query_gpu.cu:10-89):
 *     pooled[c, m] = max_s relu( feats[idx[m,s], c] + BN_pos( w_pos[c] . (xyz[idx[m,s]] - new_xyz[m]) ) )
 * idx (M, nsample) is the RAW output of mgar_voxel_query_stack (GLOBAL voxel rows; idx[m][0] == -1 marks an empty
 * neighbourhood, whose features and relative coordinates are zero as in the reference, :101,:106).  feats (N, ld_f) is the
 * row-major output of mlps_in on all voxels; pooled / arg are CHANNEL-MAJOR (C, M) (what mlps_out's Conv1d consumes);
 * 1 <= C <= 32.  BN_pos is the module's BatchNorm2d(C) after the bias-free Conv2d(3, C): in training mode its batch
 * statistics over all M*nsample columns come from mgar_voxel_roi_pool_stats (the position branch is linear in the
 * relative coordinates, so first and second moments of those suffice): moments[10] doubles = E[r] (3), Cov(r) (6: xx, xy,
 * xz, yy, yz, zz), n; mean / invstd (C) feed _fwd and _bwd; running statistics get the usual momentum update.
 * bwd: dfeats (N, ld_f) is ACCUMULATED into (caller zero-fills; may be NULL), dgamma / dbeta (C) and dw_pos (C, 3) are
 * written; train_stats = 0 treats mean / invstd as constants (eval-mode BatchNorm). */
long long mgar_voxel_roi_pool_stats_workspace_doubles(int M, int nsample);
long long mgar_voxel_roi_pool_bwd_workspace_floats(int M, int C);
int mgar_voxel_roi_pool_stats(int M, int nsample, int C, const float *xyz, const float *new_xyz, const int *idx,
                              const float *w_pos, float eps, float momentum, double *workspace, double *moments,
                              float *mean, float *invstd, float *running_mean, float *running_var,
                              long long *num_batches_tracked, void *stream);
int mgar_voxel_roi_pool_fwd(int M, int nsample, int C, const float *xyz, const float *new_xyz, const float *feats,
                            int ld_f, const int *idx, const float *w_pos, const float *mean, const float *invstd,
                            const float *gamma, const float *beta, float *pooled, unsigned char *arg, void *stream);
int mgar_voxel_roi_pool_bwd(int M, int nsample, int C, const float *xyz, const float *new_xyz, const int *idx,
                            const float *w_pos, const float *mean, const float *invstd, const float *gamma,
                            const double *moments, int train_stats, const float *dpooled, const float *pooled,
                            const unsigned char *arg, float *workspace, float *dfeats, int ld_f, float *dgamma,
                            float *dbeta, float *dw_pos, void *stream);

/* ============ sparse 3-D convolution: trunk of Voxel R-CNN ====================
 * Replaces the third-party spconv calls of pcdet/models/backbones_3d/spconv_backbone.py:69-170 (SubMConv3d /
 * SparseConv3d) and the dense voxel -> row table of pcdet/utils/common_utils.py:235-252 as a lookup structure.
 * Voxel coordinates are (N, 4) int32 [b, z, y, x].
 *
 * Hash table: table_keys (capacity) int64 pre-filled with -1, table_vals (capacity) int32 pre-filled with INT_MAX by the
 * caller; capacity a power of two >= 2 N.  _lookup writes the row stored for each coordinate or -1. */
int mgar_voxel_hash_build(int N, const int *coords, int Z, int Y, int X, long long *table_keys, int *table_vals,
                          int capacity, void *stream);
int mgar_voxel_hash_lookup(int M, const int *coords, int Z, int Y, int X, const long long *table_keys,
                           const int *table_vals, int capacity, int *rows, void *stream);
/* Rulebook of one convolution.  geom: 15 HOST ints {kz,ky,kx, sz,sy,sx, pz,py,px, Zi,Yi,Xi, Zo,Yo,Xo}; K = kz*ky*kx
 * offsets in z-major order (the order of spconv's (C_out, kz, ky, kx, C_in) weight).
 *   inverse = 0: site_coords = OUTPUT sites, table = hash of the INPUT sites -> nbr (n_sites, K): input row at
 *                o * stride - pad + k, or -1;
 *   inverse = 1: site_coords = INPUT sites, table = hash of the OUTPUT sites -> nbr (n_sites, K): output row reached
 *                through offset k, or -1 (the data gradient gathers over this table). */
int mgar_spconv_rulebook(int n_sites, const int *site_coords, const int *geom, const long long *table_keys,
                         const int *table_vals, int capacity, int inverse, int *nbr, void *stream);
/* Candidate output sites of a strided convolution: keys (n_in, K) int64 = linear ((b*Zo + z)*Yo + y)*Xo + x key of the
 * output site input i reaches through offset k, or -1.  The caller keeps the non-negative keys and makes them unique
 * (ascending = the order of the output rows). */
int mgar_spconv_output_keys(int n_in, const int *in_coords, const int *geom, long long *keys, void *stream);
/* out (No, Cout) = sum_k in[nbr[:, k]] . w[k], w (K, Cin, Cout) row-major, fp32 on the exact-fp32 MFMA; rows with
 * nbr == -1 contribute nothing; out is fully written.  flip_k != 0 reads w[K-1-k] (data gradient of a submanifold
 * convolution over its own forward table).  Cin, Cout <= 128 (MGAR_EUNSUPPORTED otherwise). */
int mgar_spconv_gather_gemm(int No, int K, int Cin, int Cout, const float *in, const int *nbr, const float *w,
                            int flip_k, float *out, void *stream);
/* Weight gradient: partial (mgar_spconv_dw_chunks(No), K, Cin, Cout) is written; dW = its sum over the first axis
 * (left to the caller: a fixed-order reduction, reproducible). */
int mgar_spconv_dw_chunks(int No);
/* The weight gradient over PAIR LISTS (the neighbour table compacted per offset): no work on sites without a neighbour, 64-pair
 * tiles software-pipelined.  pair_i / pair_o (P) int32: input / output row of every (offset, site) pair, grouped by offset,
 * ascending output row inside an offset; items (n_items, 4) int32 {offset, first pair, end pair, 0} with at most
 * mgar_spconv_pair_chunk() pairs each, grouped by ascending offset; item_start (K + 1) int32: first item of every offset.
 * partial (n_items, Cin, Cout) scratch; dw (K, Cin, Cout) fully written, summed in item order (reproducible).
 * C_in, C_out powers of two <= 128 (MGAR_EUNSUPPORTED otherwise: mgar_spconv_dw). */
int mgar_spconv_set_register_gather(int on);   /* A/B switch of mgar_spconv_gather_gemm: 1 (default) = the register-gather kernel
                                               * (spconv_os_kernel) where it applies, 0 = always the LDS-staged kernel */
int mgar_spconv_pair_chunk(void);
/* The pair lists themselves, built on the device from the neighbour table nbr (No, K) (round 3; replaces six torch passes):
 * mgar_spconv_pairs_count fills blk (K, mgar_spconv_pairs_blocks(No)) int32 (per-row-block counts, scanned per offset) and total (K)
 * int32; the caller reads total (one host synchronisation per rulebook), forms offset_start (K) int64 = exclusive sums, allocates
 * pair_i / pair_o (sum total) and calls mgar_spconv_pairs_fill: offset k's pairs land at [offset_start[k], +total[k]), ascending
 * output row (stable, deterministic). */
int mgar_spconv_pairs_blocks(int No);
int mgar_spconv_pairs_count(int No, int K, const int *nbr, int *blk, int *total, void *stream);
int mgar_spconv_pairs_fill(int No, int K, const int *nbr, const int *blk, const long long *offset_start, int *pair_i, int *pair_o,
                           void *stream);
/* Forward / data gradient over the same pair lists: for k = 0 .. K-1 in order, dst[pair_dst[p], :] += src[pair_src[p], :] . w[k]
 * over the pairs of offset k (one launch per offset: under one offset a destination row occurs once, so the update needs no
 * atomics and the order of the sum over k is fixed).  dst (rows, Cd) ZERO-FILLED by the caller; w (K, Cs, Cd) row-major;
 * item_start_host: the (K + 1) item offsets as a HOST array.  Forward: (src, pair_src, pair_dst, w) = (in, pair_i, pair_o, W);
 * data gradient: (dout, pair_o, pair_i, W_k^T) -- no inverse table.  C_s a power of two, C_s, C_d <= 128. */
int mgar_spconv_pairs_gemm(int K, int Cs, int Cd, const float *src, const int *pair_src, const int *pair_dst, const int *items,
                           const int *item_start_host, const float *w, float *dst, void *stream);
int mgar_spconv_pairs_dw(int n_items, int K, int Cin, int Cout, const float *in, const float *dout, const int *pair_i,
                         const int *pair_o, const int *items, const int *item_start, float *partial, float *dw, void *stream);
int mgar_spconv_dw(int No, int K, int Cin, int Cout, const float *in, const int *nbr, const float *dout,
                   float *partial, void *stream);

/* ===================== per-actor point crop ========================
 * points_in_boxes_gpu   pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:98-116; kernel roiaware_pool3d_kernel.cu:313-334
 * boxes (B, N, 7) [x, y, z, dx, dy, dz, heading], pts (B, P, 3) -> box_idx_of_points (B, P): index of the FIRST box
 * containing the point, -1 = background.  Every entry is written. */
int mgar_points_in_boxes(int batch_size, int boxes_num, int pts_num, const float *boxes, const float *pts,
                         int *box_idx_of_points, void *stream);
/* roipool3d_gpu / roipool3dLauncher   pcdet/ops/roipoint_pool3d/src/roipoint_pool3d.cpp:23-51, roipoint_pool3d_kernel.cu:138-166
 * xyz (B, N, 3), boxes3d (B, M, 7), pts_feature (B, N, C) -> pooled_features (B, M, S, 3 + C): the first S points (in
 * index order) inside each box, cyclically repeated when fewer; pooled_empty_flag (B, M) = 1 for a box without points,
 * whose rows are left untouched.  Both outputs are ZERO-FILLED by the caller (roipoint_pool3d_utils.py:50-51).  Unlike
 * the reference this neither allocates nor uses an O(B*N*M) scratch. */
int mgar_roipoint_pool3d_fwd(int batch_size, int pts_num, int boxes_num, int feature_in_len, int sampled_pts_num,
                             const float *xyz, const float *boxes3d, const float *pts_feature, float *pooled_features,
                             int *pooled_empty_flag, void *stream);

/* ============================ input preparation ============================
 * dataloader.py:47-49   transforms.Resize(image_size) + ToTensor() + Normalize(mean, std) of every stitched frame.  Resize on
 * a PIL image is Pillow's Image.resize(size, BILINEAR) (third-party, un-vendored; src/libImaging/Resample.c restated in
 * oracle/oracle.py::pil_bilinear_resize and pinned against Pillow itself).
 *
 * mgar_image_resample_ksize / _coeffs: HOST helpers -- the per-axis tables of Pillow's precompute_coeffs +
 * normalize_coeffs_8bpc for the bilinear filter: bounds (out_size, 2) int [first tap, taps], kk (out_size, ksize) int, 22-bit
 * fixed point; in_size == out_size gives the unit table (ksize 1), i.e. the pass Pillow skips.  ksize returns the row length. */
int mgar_image_resample_ksize(int in_size, int out_size);
int mgar_image_resample_coeffs(int in_size, int out_size, int *bounds, int *kk);
/* src (frames, in_h, in_w, 3) uint8 in HBM; xbounds / xkk / ybounds / ykk the DEVICE copies of the tables above for
 * (in_w -> out_w) and (in_h -> out_h); lut (3, 256) float32: lut[c][v] = the normalised value of byte v in channel c, as the
 * caller's float32 arithmetic gives it.  Resamples exactly as Pillow does (horizontal pass to bytes, then vertical) and writes
 * dst[f * dst_frame_stride + c * dst_channel_stride + y * out_w + x] (element strides): dst_kind 0 float32, 1 bfloat16;
 * dst_kind 2 writes the resampled BYTES, (frames, out_h, out_w, 3), and ignores lut and the strides.
 * MGAR_EUNSUPPORTED when one tile's source rows do not fit LDS (down-scaling beyond roughly 20x). */
int mgar_image_resize_normalize_u8(int frames, int in_h, int in_w, int out_h, int out_w, const unsigned char *src,
                                   const int *xbounds, const int *xkk, const int *ybounds, const int *ykk, const float *lut,
                                   void *dst, long long dst_frame_stride, long long dst_channel_stride, int dst_kind,
                                   void *stream);
/* dataloader.py:119-128 load_pc (upper / lower velodyne moved to the base frame, upper first) followed by the x / y range mask
 * of mask_points_and_boxes_outside_range (pcdet/datasets/processor/data_processor.py:78-84, common_utils.py:60-63), in one
 * ordered compaction.  upper (n_upper, C), lower (n_lower, C) float32 rows [x, y, z, features...]; tf_upper / tf_lower HOST
 * arrays of 12 floats, row-major [R | t]; xy_range HOST array [x_min, y_min, x_max, y_max] (bounds inclusive);
 * workspace mgar_velodyne_merge_crop_workspace_ints() ints; out (n_upper + n_lower, C) capacity, rows [0, *count) written in
 * input order; count one int in HBM. */
long long mgar_velodyne_merge_crop_workspace_ints(int n_upper, int n_lower);
int mgar_velodyne_merge_crop(int n_upper, int n_lower, int C, const float *upper, const float *lower, const float *tf_upper,
                             const float *tf_lower, const float *xy_range, int *workspace, float *out, int *count,
                             void *stream);

/* ============================ bf16 feature payloads (BASELINE configs c2, c5) ============================
 * The reference's kernels are fp32 + int32 only.  For the bf16 configurations SURVEY.md section 8 keeps coordinates,
 * distances, indices and BatchNorm statistics in fp32 / int32 and stores only FEATURE PAYLOADS (and runs the GEMMs) in
 * bf16.  Every entry point below is the twin of the fp32 entry point of the same name without the suffix: same
 * arguments, same semantics, but the pointers named in the comment address bf16 elements (torch.bfloat16 storage,
 * passed as void*).  Arithmetic is fp32 inside the kernels (elements are widened on load and rounded to nearest even
 * on store), so index outputs are bit-identical to the fp32 path by construction and features agree with it to
 * bf16 rounding (2^-9 relative per stored tensor).  Query ops (ball / voxel query, FPS, three_nn) have no twin: they
 * never touch a payload. */
/* features, out */
int mgar_query_group_batch_fwd_bf16(int b, int c, int n, int npoints, int nsample, const float *xyz, const float *new_xyz,
                                    const void *features, const int *idx, void *out, void *stream);
int mgar_query_group_stack_fwd_bf16(int B, int M, int C, int nsample, const float *xyz, const int *xyz_batch_cnt,
                                    const float *new_xyz, const int *new_xyz_batch_cnt, const void *features,
                                    const int *idx, void *out, void *stream);
/* zf, rel_out, y_out (wx stays fp32) */
int mgar_query_group_proj_batch_fwd_bf16(int b, int c, int n, int npoints, int nsample, const float *xyz,
                                         const float *new_xyz, const void *zf, const float *wx, const int *idx,
                                         void *rel_out, void *y_out, void *stream);
int mgar_query_group_proj_stack_fwd_bf16(int B, int M, int C, int nsample, const float *xyz, const int *xyz_batch_cnt,
                                         const float *new_xyz, const int *new_xyz_batch_cnt, const void *zf, int zf_ld,
                                         const float *wx, const int *idx, void *rel_out, void *y_out, void *stream);
/* x (statistics, affine, running statistics: fp32) */
int mgar_bn_train_stats_bf16(const void *x, int B, int C, int P, float eps, float momentum, float *workspace,
                             float *mean, float *invstd, float *running_mean, float *running_var,
                             long long *num_batches_tracked, void *stream);
int mgar_bn_train_stats_grouped_bf16(const void *x, int G, int C, int P, float eps, float momentum, float *workspace,
                                     float *mean, float *invstd, float *running_mean, float *running_var,
                                     long long *num_batches_tracked, void *stream);
/* x, y */
int mgar_bn_act_fwd_bf16(const void *x, int B, int C, int P, const float *mean, const float *invstd,
                         const float *gamma, const float *beta, int relu, void *y, void *stream);
int mgar_three_interpolate_batch_into_bf16(int b, int c, int m, int n, const void *points, const int *idx, const float *weight,
                                           void *out, long long out_bstride, void *stream);
int mgar_three_interpolate_batch_add_bf16(int b, int c, int m, int n, const void *points, const int *idx, const float *weight,
                                          void *out, void *stream);
int mgar_bn_cl_train_stats_bf16(const void *x, int S, int R, int C, int per_sample, float eps, float momentum, float *workspace,
                                float *mean, float *invstd, float *running_mean, float *running_var,
                                long long *num_batches_tracked, void *stream);
int mgar_bn_cl_act_fwd_bf16(const void *x, int S, int R, int C, int per_sample, const float *mean, const float *invstd,
                            const float *gamma, const float *beta, int relu, void *y, int ldy, void *stream);
int mgar_bn_act_fwd_to_cl_bf16(const void *x, int S, int C, int R, int per_sample, const float *mean, const float *invstd,
                               const float *gamma, const float *beta, int relu, void *y, int ldy, void *stream);
int mgar_maxpool3d_same_fwd_cl_bf16(const void *x, int N, int T, int H, int W, int C, int kt, int kh, int kw, int st, int sh,
                                    int sw, void *y, void *stream);
int mgar_bn_act_small_bf16(const void *x, int B, int C, int P, int per_sample, float eps, float momentum, const float *gamma,
                           const float *beta, int relu, float *workspace, float *mean, float *invstd, float *running_mean,
                           float *running_var, long long *num_batches_tracked, void *y, long long y_bstride, void *stream);
int mgar_bn_act_fwd_into_bf16(const void *x, int B, int C, int P, const float *mean, const float *invstd, const float *gamma,
                              const float *beta, int relu, int stats_per_sample, void *y, long long y_bstride, void *stream);
int mgar_bn_act_fwd_grouped_bf16(const void *x, int G, int C, int P, const float *mean, const float *invstd,
                                 const float *gamma, const float *beta, int relu, void *y, void *stream);
/* x, out, xarg */
int mgar_bn_act_maxpool_fwd_bf16(const void *x, int B, int C, int M, int nsample, const float *mean,
                                 const float *invstd, const float *gamma, const float *beta, int relu, void *out,
                                 unsigned char *arg, void *xarg, void *stream);
/* dy, x, dx (dgamma, dbeta fp32) */
int mgar_bn_act_bwd_bf16(const void *dy, const void *x, int B, int C, int P, const float *mean, const float *invstd,
                         const float *gamma, const float *beta, int relu, float *workspace, float *dgamma,
                         float *dbeta, void *dx, void *stream);
/* dpool, pooled, x, xarg, dx */
int mgar_bn_act_maxpool_bwd_bf16(const void *dpool, const void *pooled, const unsigned char *arg, const void *x,
                                 const void *xarg, int B, int C, int M, int nsample, const float *mean, const float *invstd,
                                 const float *gamma, int relu, float *workspace, float *dgamma, float *dbeta,
                                 void *dx, void *stream);
/* x, y (w and the BatchNorm vectors fp32; fp32 accumulation) */
int mgar_pointwise_conv_fwd_bf16(const void *x, int B, int Cin, int P, const float *w, int w_row_stride,
                                 int w_col_stride, int Cout, const float *in_mean, const float *in_invstd,
                                 const float *in_gamma, const float *in_beta, int in_relu, void *y, void *stream);
/* points / features, out (weight fp32) */
int mgar_three_interpolate_batch_bf16(int b, int c, int m, int n, const void *points, const int *idx,
                                      const float *weight, void *out, void *stream);
int mgar_three_interpolate_stack_bf16(int N, int C, const void *features, const int *idx, const float *weight,
                                      void *out, void *stream);
/* x, y */
int mgar_maxpool3d_same_fwd_bf16(const void *x, int NC, int T, int H, int W, int kt, int kh, int kw, int st, int sh,
                                 int sw, void *y, void *stream);
int mgar_maxpool3d_valid_fwd_bf16(const void *x, int NC, int T, int H, int W, int kt, int kh, int kw, int st, int sh, int sw, void *y,
                                  void *stream);
/* input, out (rois fp32) */
int mgar_roi_align_fwd_bf16(const void *input, int N, int C, int H, int W, const float *rois, int K,
                            int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio, int aligned,
                            void *out, void *stream);
/* x, y (w and the accumulation fp32) */
int mgar_stem_conv3d_fwd_bf16(const void *x, int N, int T, int H, int W, const float *w, float *w_packed, void *y,
                              void *stream);
/* feats, pooled */
int mgar_voxel_roi_pool_fwd_bf16(int M, int nsample, int C, const float *xyz, const float *new_xyz, const void *feats,
                                 int ld_f, const int *idx, const float *w_pos, const float *mean, const float *invstd,
                                 const float *gamma, const float *beta, void *pooled, unsigned char *arg, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MGAR_OPS_H */
