/*
 * mgar_oracle.c -- CPU restatement of the reference's pointnet2 CUDA kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under multimodal_gar_amd/ may import, link
 * or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and there only as the checker / the timed CPU baseline.
 *
 * Each function follows one reference kernel loop-for-loop (same scan order, same
 * strict / non-strict comparisons, same tie rules, same reduction-tree geometry).
 * Citations are file:line relative to /root/reference/pcdet/ops/pointnet2/.
 *
 * PARITY STATUS: "parity unpinned" by reference tests -- the reference ships no
 * tests, golden vectors or a CPU path for these kernels, and its
 * CUDA sources cannot be built here (no nvcc, no GPU).  The restatement is pinned
 * instead by independent brute-force numpy definitions (tests/test_oracle_cpu.py)
 * and by committed golden fixtures generated from it (tests/golden/).
 *
 * Floating-point convention.  The reference is compiled by nvcc with its default
 * -fmad=true, i.e. a*a + b*b + c*c is contracted.  The contraction LLVM's DAG
 * combiner (used by NVVM and by hipcc alike) picks for
 *     (dx*dx + dy*dy) + dz*dz
 * is  fma(dz, dz, fma(dx, dx, dy*dy)).  Both this oracle and the HIP kernels spell
 * that out with explicit fmaf() and are built with -ffp-contract=off so neither
 * compiler is free to choose differently.  Same for the 3-term interpolation sum.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

static inline float orc_d2(float dx, float dy, float dz) {
    return fmaf(dz, dz, fmaf(dx, dx, dy * dy));
}

/* w0*p0 + w1*p1 + w2*p2 under the same contraction rule */
static inline float orc_dot3(float w0, float p0, float w1, float p1, float w2, float p2) {
    return fmaf(w2, p2, fmaf(w0, p0, w1 * p1));
}

/* ------------------------------------------------------------------ */
/* pointnet2_batch/src/cuda_utils.h:10-14  opt_n_threads               */
/* ------------------------------------------------------------------ */
ORC_API void orc_set_threads(int t) {
#ifdef _OPENMP
    omp_set_num_threads(t > 0 ? t : 1);
#else
    (void)t;
#endif
}

ORC_API int orc_opt_n_threads(int work_size) {
    const int pow_2 = (int)(log((double)work_size) / log(2.0));
    int v = 1 << pow_2;
    if (v > 1024) v = 1024;
    if (v < 1) v = 1;
    return v;
}

/* ------------------------------------------------------------------ */
/* ball query, batch: pointnet2_batch/src/ball_query_gpu.cu:15-51      */
/* idx rows of empty balls are left untouched (caller zero-fills).     */
/* ------------------------------------------------------------------ */
ORC_API void orc_ball_query_batch(int b, int n, int m, float radius, int nsample,
                                  const float *new_xyz, const float *xyz, int *idx) {
    const float radius2 = radius * radius;
    for (int bs = 0; bs < b; ++bs) {
        const float *P = xyz + (size_t)bs * n * 3;
#pragma omp parallel for schedule(static)
        for (int pt = 0; pt < m; ++pt) {
            const float *q = new_xyz + ((size_t)bs * m + pt) * 3;
            int *row = idx + ((size_t)bs * m + pt) * nsample;
            const float nx = q[0], ny = q[1], nz = q[2];
            int cnt = 0;
            for (int k = 0; k < n; ++k) {
                const float x = P[k * 3 + 0], y = P[k * 3 + 1], z = P[k * 3 + 2];
                const float d2 = orc_d2(nx - x, ny - y, nz - z);
                if (d2 < radius2) {
                    if (cnt == 0)
                        for (int l = 0; l < nsample; ++l) row[l] = k;
                    row[cnt] = k;
                    ++cnt;
                    if (cnt >= nsample) break;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* ball query, stack: pointnet2_stack/src/ball_query_gpu.cu:16-66      */
/* segment search :27-35, idx[0] = -1 for an empty ball :65            */
/* ------------------------------------------------------------------ */
ORC_API void orc_ball_query_stack(int B, int M, float radius, int nsample,
                                  const float *new_xyz, const int *new_xyz_batch_cnt,
                                  const float *xyz, const int *xyz_batch_cnt, int *idx) {
    const float radius2 = radius * radius;
#pragma omp parallel for schedule(static)
    for (int pt = 0; pt < M; ++pt) {
        int bs_idx = 0, pt_cnt = new_xyz_batch_cnt[0];
        for (int k = 1; k < B; k++) {
            if (pt < pt_cnt) break;
            pt_cnt += new_xyz_batch_cnt[k];
            bs_idx = k;
        }
        int start = 0;
        for (int k = 0; k < bs_idx; k++) start += xyz_batch_cnt[k];
        const float *q = new_xyz + (size_t)pt * 3;
        const float *P = xyz + (size_t)start * 3;
        int *row = idx + (size_t)pt * nsample;
        const float nx = q[0], ny = q[1], nz = q[2];
        const int n = xyz_batch_cnt[bs_idx];
        int cnt = 0;
        for (int k = 0; k < n; ++k) {
            const float x = P[k * 3 + 0], y = P[k * 3 + 1], z = P[k * 3 + 2];
            const float d2 = orc_d2(nx - x, ny - y, nz - z);
            if (d2 < radius2) {
                if (cnt == 0)
                    for (int l = 0; l < nsample; ++l) row[l] = k;
                row[cnt] = k;
                ++cnt;
                if (cnt >= nsample) break;
            }
        }
        if (cnt == 0) row[0] = -1;
    }
}

/* ------------------------------------------------------------------ */
/* farthest point sampling: one simulated thread block per cloud.      */
/* pointnet2_batch/src/sampling_gpu.cu:93-98 (__update), :101-216      */
/* (kernel), :218-259 (launcher: block_size = opt_n_threads(n)).       */
/* dists/dists_i model the __shared__ arrays; the tree is run literally */
/* ------------------------------------------------------------------ */
static void orc_fps_one(int n, int m, int block_size, const float *dataset, float *temp,
                        int *idxs, int idx_offset, float *dists, int *dists_i) {
    if (m <= 0) return;
    int old = 0;
    idxs[0] = old + idx_offset;
    for (int j = 1; j < m; j++) {
        const float x1 = dataset[old * 3 + 0];
        const float y1 = dataset[old * 3 + 1];
        const float z1 = dataset[old * 3 + 2];
        for (int tid = 0; tid < block_size; ++tid) {
            int besti = 0;
            float best = -1;
            for (int k = tid; k < n; k += block_size) {
                const float x2 = dataset[k * 3 + 0], y2 = dataset[k * 3 + 1], z2 = dataset[k * 3 + 2];
                const float d = orc_d2(x2 - x1, y2 - y1, z2 - z1);
                const float d2 = fminf(d, temp[k]);
                temp[k] = d2;
                besti = d2 > best ? k : besti;
                best = d2 > best ? d2 : best;
            }
            dists[tid] = best;
            dists_i[tid] = besti;
        }
        for (int s = block_size / 2; s >= 1; s >>= 1) {
            for (int tid = 0; tid < s; ++tid) {
                const float v1 = dists[tid], v2 = dists[tid + s];
                const int i1 = dists_i[tid], i2 = dists_i[tid + s];
                dists[tid] = fmaxf(v1, v2);
                dists_i[tid] = v2 > v1 ? i2 : i1;
            }
        }
        old = dists_i[0];
        idxs[j] = old + idx_offset;
    }
}

ORC_API void orc_fps_batch(int b, int n, int m, const float *dataset, float *temp, int *idxs) {
    const int bs = orc_opt_n_threads(n);
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < b; ++i) {
        float dists[1024];
        int dists_i[1024];
        orc_fps_one(n, m, bs, dataset + (size_t)i * n * 3, temp + (size_t)i * n, idxs + (size_t)i * m, 0,
                    dists, dists_i);
    }
}

/* pointnet2_stack/src/sampling_gpu.cu:188-319: always 1024 threads, per-segment m,
 * global indices (+xyz_batch_start_idx), idxs[0] = segment start even when m == 0
 * is NOT guarded in the reference (it writes idxs[0] unconditionally, :228); we keep
 * that write only when m > 0 so an m == 0 segment cannot clobber its neighbour --
 * the reference's Python never passes m == 0 (pointnet2_utils.py:191-225).        */
ORC_API void orc_fps_stack(int batch_size, const float *dataset, float *temp, const int *xyz_batch_cnt,
                           int *idxs, const int *num_sampled_points) {
    float *dists = (float *)malloc(sizeof(float) * 1024);
    int *dists_i = (int *)malloc(sizeof(int) * 1024);
    int start = 0, ostart = 0;
    for (int i = 0; i < batch_size; ++i) {
        const int n = xyz_batch_cnt[i], m = num_sampled_points[i];
        orc_fps_one(n, m, 1024, dataset + (size_t)start * 3, temp + start, idxs + ostart, start, dists, dists_i);
        start += n;
        ostart += m;
    }
    free(dists);
    free(dists_i);
}

/* ------------------------------------------------------------------ */
/* gather: pointnet2_batch/src/sampling_gpu.cu:15-31, grad :53-70       */
/* ------------------------------------------------------------------ */
ORC_API void orc_gather_points(int b, int c, int n, int m, const float *points, const int *idx, float *out) {
    for (int bs = 0; bs < b; ++bs)
        for (int ci = 0; ci < c; ++ci)
            for (int pt = 0; pt < m; ++pt)
                out[((size_t)bs * c + ci) * m + pt] = points[((size_t)bs * c + ci) * n + idx[(size_t)bs * m + pt]];
}

ORC_API void orc_gather_points_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                    float *grad_points) {
    for (int bs = 0; bs < b; ++bs)
        for (int ci = 0; ci < c; ++ci)
            for (int pt = 0; pt < m; ++pt)
                grad_points[((size_t)bs * c + ci) * n + idx[(size_t)bs * m + pt]] +=
                    grad_out[((size_t)bs * c + ci) * m + pt];
}

/* ------------------------------------------------------------------ */
/* group, batch: pointnet2_batch/src/group_points_gpu.cu:53-72 / 14-31  */
/* ------------------------------------------------------------------ */
ORC_API void orc_group_points_batch(int b, int c, int n, int npoints, int nsample, const float *points,
                                    const int *idx, float *out) {
    for (int bs = 0; bs < b; ++bs)
        for (int ci = 0; ci < c; ++ci)
            for (int pt = 0; pt < npoints; ++pt)
                for (int s = 0; s < nsample; ++s) {
                    const int k = idx[((size_t)bs * npoints + pt) * nsample + s];
                    out[(((size_t)bs * c + ci) * npoints + pt) * nsample + s] = points[((size_t)bs * c + ci) * n + k];
                }
}

ORC_API void orc_group_points_grad_batch(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                                         const int *idx, float *grad_points) {
    for (int bs = 0; bs < b; ++bs)
        for (int ci = 0; ci < c; ++ci)
            for (int pt = 0; pt < npoints; ++pt)
                for (int s = 0; s < nsample; ++s) {
                    const int k = idx[((size_t)bs * npoints + pt) * nsample + s];
                    grad_points[((size_t)bs * c + ci) * n + k] +=
                        grad_out[(((size_t)bs * c + ci) * npoints + pt) * nsample + s];
                }
}

/* ------------------------------------------------------------------ */
/* group, stack: pointnet2_stack/src/group_points_gpu.cu:71-102 / 15-45 */
/* ------------------------------------------------------------------ */
static int orc_segment_of(int pt, int B, const int *cnt) {
    int bs_idx = 0, pt_cnt = cnt[0];
    for (int k = 1; k < B; k++) {
        if (pt < pt_cnt) break;
        pt_cnt += cnt[k];
        bs_idx = k;
    }
    return bs_idx;
}

ORC_API void orc_group_points_stack(int B, int M, int C, int nsample, const float *features,
                                    const int *features_batch_cnt, const int *idx, const int *idx_batch_cnt,
                                    float *out) {
    for (int pt = 0; pt < M; ++pt) {
        const int bs_idx = orc_segment_of(pt, B, idx_batch_cnt);
        int start = 0;
        for (int k = 0; k < bs_idx; k++) start += features_batch_cnt[k];
        const float *F = features + (size_t)start * C;
        for (int ci = 0; ci < C; ++ci)
            for (int s = 0; s < nsample; ++s)
                out[((size_t)pt * C + ci) * nsample + s] = F[(size_t)idx[(size_t)pt * nsample + s] * C + ci];
    }
}

ORC_API void orc_group_points_grad_stack(int B, int M, int C, int N, int nsample, const float *grad_out,
                                         const int *idx, const int *idx_batch_cnt, const int *features_batch_cnt,
                                         float *grad_features) {
    (void)N;
    for (int pt = 0; pt < M; ++pt) {
        const int bs_idx = orc_segment_of(pt, B, idx_batch_cnt);
        int start = 0;
        for (int k = 0; k < bs_idx; k++) start += features_batch_cnt[k];
        for (int ci = 0; ci < C; ++ci)
            for (int s = 0; s < nsample; ++s)
                grad_features[((size_t)start + idx[(size_t)pt * nsample + s]) * C + ci] +=
                    grad_out[((size_t)pt * C + ci) * nsample + s];
    }
}

/* ------------------------------------------------------------------ */
/* three_nn: pointnet2_batch/src/interpolate_gpu.cu:16-59               */
/* best1..3 are doubles initialised to 1e40 in the reference and stored */
/* to a float output: an unfilled slot comes out as +inf, index 0.      */
/* ------------------------------------------------------------------ */
static void orc_three_nn_one(const float *u, const float *known, int m, int idx_offset, float *dist2, int *idx) {
    const float ux = u[0], uy = u[1], uz = u[2];
    double best1 = 1e40, best2 = 1e40, best3 = 1e40;
    int besti1 = 0, besti2 = 0, besti3 = 0;
    for (int k = 0; k < m; ++k) {
        const float x = known[k * 3 + 0], y = known[k * 3 + 1], z = known[k * 3 + 2];
        const float d = orc_d2(ux - x, uy - y, uz - z);
        if (d < best1) {
            best3 = best2; besti3 = besti2;
            best2 = best1; besti2 = besti1;
            best1 = d; besti1 = k;
        } else if (d < best2) {
            best3 = best2; besti3 = besti2;
            best2 = d; besti2 = k;
        } else if (d < best3) {
            best3 = d; besti3 = k;
        }
    }
    dist2[0] = (float)best1; dist2[1] = (float)best2; dist2[2] = (float)best3;
    idx[0] = besti1 + idx_offset; idx[1] = besti2 + idx_offset; idx[2] = besti3 + idx_offset;
}

ORC_API void orc_three_nn_batch(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                                int *idx) {
    for (int bs = 0; bs < b; ++bs)
#pragma omp parallel for schedule(static)
        for (int pt = 0; pt < n; ++pt)
            orc_three_nn_one(unknown + ((size_t)bs * n + pt) * 3, known + (size_t)bs * m * 3, m, 0,
                             dist2 + ((size_t)bs * n + pt) * 3, idx + ((size_t)bs * n + pt) * 3);
}

/* pointnet2_stack/src/interpolate_gpu.cu:16-75 (global indices :72-74) */
ORC_API void orc_three_nn_stack(int batch_size, int N, const float *unknown, const int *unknown_batch_cnt,
                                const float *known, const int *known_batch_cnt, float *dist2, int *idx) {
    for (int pt = 0; pt < N; ++pt) {
        const int bs_idx = orc_segment_of(pt, batch_size, unknown_batch_cnt);
        int start = 0;
        for (int k = 0; k < bs_idx; k++) start += known_batch_cnt[k];
        orc_three_nn_one(unknown + (size_t)pt * 3, known + (size_t)start * 3, known_batch_cnt[bs_idx], start,
                         dist2 + (size_t)pt * 3, idx + (size_t)pt * 3);
    }
}

/* ------------------------------------------------------------------ */
/* three_interpolate batch: interpolate_gpu.cu:84-104, grad :127-149    */
/* ------------------------------------------------------------------ */
ORC_API void orc_three_interpolate_batch(int b, int c, int m, int n, const float *points, const int *idx,
                                         const float *weight, float *out) {
    for (int bs = 0; bs < b; ++bs)
        for (int ci = 0; ci < c; ++ci) {
            const float *P = points + ((size_t)bs * c + ci) * m;
            for (int pt = 0; pt < n; ++pt) {
                const float *w = weight + ((size_t)bs * n + pt) * 3;
                const int *id = idx + ((size_t)bs * n + pt) * 3;
                out[((size_t)bs * c + ci) * n + pt] = orc_dot3(w[0], P[id[0]], w[1], P[id[1]], w[2], P[id[2]]);
            }
        }
}

ORC_API void orc_three_interpolate_grad_batch(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                              const float *weight, float *grad_points) {
    for (int bs = 0; bs < b; ++bs)
        for (int ci = 0; ci < c; ++ci) {
            float *G = grad_points + ((size_t)bs * c + ci) * m;
            for (int pt = 0; pt < n; ++pt) {
                const float *w = weight + ((size_t)bs * n + pt) * 3;
                const int *id = idx + ((size_t)bs * n + pt) * 3;
                const float g = grad_out[((size_t)bs * c + ci) * n + pt];
                G[id[0]] += g * w[0];
                G[id[1]] += g * w[1];
                G[id[2]] += g * w[2];
            }
        }
}

/* stack: pointnet2_stack/src/interpolate_gpu.cu:107-126, grad :151-172 */
ORC_API void orc_three_interpolate_stack(int N, int channels, const float *features, const int *idx,
                                         const float *weight, float *out) {
    for (int pt = 0; pt < N; ++pt) {
        const float *w = weight + (size_t)pt * 3;
        const int *id = idx + (size_t)pt * 3;
        for (int ci = 0; ci < channels; ++ci)
            out[(size_t)pt * channels + ci] =
                orc_dot3(w[0], features[(size_t)id[0] * channels + ci], w[1], features[(size_t)id[1] * channels + ci],
                         w[2], features[(size_t)id[2] * channels + ci]);
    }
}

ORC_API void orc_three_interpolate_grad_stack(int N, int channels, const float *grad_out, const int *idx,
                                              const float *weight, float *grad_features) {
    for (int pt = 0; pt < N; ++pt) {
        const float *w = weight + (size_t)pt * 3;
        const int *id = idx + (size_t)pt * 3;
        for (int ci = 0; ci < channels; ++ci) {
            const float g = grad_out[(size_t)pt * channels + ci];
            grad_features[(size_t)id[0] * channels + ci] += g * w[0];
            grad_features[(size_t)id[1] * channels + ci] += g * w[1];
            grad_features[(size_t)id[2] * channels + ci] += g * w[2];
        }
    }
}

/* ------------------------------------------------------------------ */
/* voxel query: pointnet2_stack/src/voxel_query_gpu.cu:10-89            */
/* dz,dy,dx ascending; accept d2 <= r2 (":65 if (dist2 > radius2) continue") */
/* ------------------------------------------------------------------ */
ORC_API void orc_voxel_query(int M, int R1, int R2, int R3, int nsample, float radius, int z_range, int y_range,
                             int x_range, const float *new_xyz, const float *xyz, const int *new_coords,
                             const int *point_indices, int *idx) {
    const float radius2 = radius * radius;
#pragma omp parallel for schedule(static)
    for (int pt = 0; pt < M; ++pt) {
        const float nx = new_xyz[(size_t)pt * 3 + 0], ny = new_xyz[(size_t)pt * 3 + 1], nz = new_xyz[(size_t)pt * 3 + 2];
        const int *co = new_coords + (size_t)pt * 4;
        int *row = idx + (size_t)pt * nsample;
        const int batch_idx = co[0], cz = co[1], cy = co[2], cx = co[3];
        int cnt = 0;
        for (int dz = -z_range; dz <= z_range; ++dz) {
            const int z_coord = cz + dz;
            if (z_coord < 0 || z_coord >= R1) continue;
            for (int dy = -y_range; dy <= y_range; ++dy) {
                const int y_coord = cy + dy;
                if (y_coord < 0 || y_coord >= R2) continue;
                for (int dx = -x_range; dx <= x_range; ++dx) {
                    const int x_coord = cx + dx;
                    if (x_coord < 0 || x_coord >= R3) continue;
                    const size_t index = (((size_t)batch_idx * R1 + z_coord) * R2 + y_coord) * R3 + x_coord;
                    const int nb = point_indices[index];
                    if (nb < 0) continue;
                    const float xp = xyz[(size_t)nb * 3 + 0], yp = xyz[(size_t)nb * 3 + 1], zp = xyz[(size_t)nb * 3 + 2];
                    const float dist2 = orc_d2(xp - nx, yp - ny, zp - nz);
                    if (dist2 > radius2) continue;
                    if (cnt < nsample) {
                        if (cnt == 0)
                            for (int l = 0; l < nsample; ++l) row[l] = nb;
                        row[cnt] = nb;
                        ++cnt;
                    }
                }
            }
        }
        if (cnt == 0) row[0] = -1;
    }
}

/* ------------------------------------------------------------------ */
/* RoIAlign forward, torchvision.ops.roi_align semantics (third party,  */
/* torchvision==0.17.2 per reference requirements.txt:422; call site    */
/* model/gat_model.py:1056-1057: aligned=False, sampling_ratio=-1).     */
/* Restated from the published algorithm (Detectron RoIAlign):          */
/*   roi = box*scale; w,h = max(end-start, 1); bin = size/pooled;       */
/*   grid = ceil(size/pooled) samples per bin axis; bilinear with the   */
/*   "y < -1 or y > H -> 0", "y <= 0 -> 0", clamp-to-last-pixel rules.  */
/* rois: (K,5) [batch, x1,y1,x2,y2]                                      */
/* ------------------------------------------------------------------ */
static float orc_bilinear(const float *img, int H, int W, float y, float x) {
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.f;
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
    if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
    const float ly = y - y_low, lx = x - x_low, hy = 1.f - ly, hx = 1.f - lx;
    const float v1 = img[y_low * W + x_low], v2 = img[y_low * W + x_high];
    const float v3 = img[y_high * W + x_low], v4 = img[y_high * W + x_high];
    const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

ORC_API void orc_roi_align_fwd(const float *input, int N, int C, int H, int W, const float *rois, int K,
                               int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio, int aligned,
                               float *out) {
    (void)N;
    const float offset = aligned ? 0.5f : 0.f;
    for (int k = 0; k < K; ++k) {
        const float *r = rois + (size_t)k * 5;
        const int bi = (int)r[0];
        const float x1 = r[1] * spatial_scale - offset, y1 = r[2] * spatial_scale - offset;
        const float x2 = r[3] * spatial_scale - offset, y2 = r[4] * spatial_scale - offset;
        float rw = x2 - x1, rh = y2 - y1;
        if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
        const float bin_h = rh / (float)pooled_h, bin_w = rw / (float)pooled_w;
        const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / pooled_h);
        const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / pooled_w);
        const float count = fmaxf((float)(gh * gw), 1.f);
        for (int c = 0; c < C; ++c) {
            const float *img = input + ((size_t)bi * C + c) * H * W;
            for (int ph = 0; ph < pooled_h; ++ph)
                for (int pw = 0; pw < pooled_w; ++pw) {
                    float acc = 0.f;
                    for (int iy = 0; iy < gh; ++iy) {
                        const float y = y1 + ph * bin_h + (iy + .5f) * bin_h / (float)gh;
                        for (int ix = 0; ix < gw; ++ix) {
                            const float x = x1 + pw * bin_w + (ix + .5f) * bin_w / (float)gw;
                            acc += orc_bilinear(img, H, W, y, x);
                        }
                    }
                    out[(((size_t)k * C + c) * pooled_h + ph) * pooled_w + pw] = acc / count;
                }
        }
    }
}


/* ======================= per-actor point crop ==========================
 * pcdet/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:18-37 (lidar_to_local_coords, check_pt_in_box3d), :313-334
 * (points_in_boxes_kernel) and pcdet/ops/roipoint_pool3d/src/roipoint_pool3d_kernel.cu:15-135 (assign_pts_to_box3d,
 * get_pooled_idx, roipool3d_forward).  Arithmetic as written there: the z test and the half-extent tests compare in
 * double (dz / 2.0, dx / 2.0 + MARGIN with a float MARGIN), the rotation uses float cos / sin of -heading and
 * un-contracted float products.  The box test is PINNED against the reference's own CPU build of the same test
 * (roiaware_pool3d.cpp:118-168, MARGIN 1e-2 there; oracle/_ref, tests/test_point_crop_cpu.py). */
static int orc_pt_in_box3d(const float *pt, const float *box3d, float margin, float *local_x, float *local_y) {
    const float x = pt[0], y = pt[1], z = pt[2];
    const float cx = box3d[0], cy = box3d[1], cz = box3d[2];
    const float dx = box3d[3], dy = box3d[4], dz = box3d[5], rz = box3d[6];
    if (fabsf(z - cz) > dz / 2.0) return 0;
    const float shift_x = x - cx, shift_y = y - cy;
    const float cosa = cosf(-rz), sina = sinf(-rz);
    *local_x = shift_x * cosa + shift_y * (-sina);
    *local_y = shift_x * sina + shift_y * cosa;
    return (fabs(*local_x) < dx / 2.0 + margin) & (fabs(*local_y) < dy / 2.0 + margin);
}

/* (N boxes, P points) 0/1 matrix with an explicit margin: the shape of the reference's points_in_boxes_cpu */
ORC_API void orc_points_in_boxes_mask(int boxes_num, int pts_num, const float *boxes, const float *pts, float margin, int *out) {
    float lx, ly;
    for (int i = 0; i < boxes_num; ++i)
        for (int j = 0; j < pts_num; ++j) out[(size_t)i * pts_num + j] = orc_pt_in_box3d(pts + (size_t)j * 3, boxes + (size_t)i * 7, margin, &lx, &ly);
}

/* points_in_boxes_kernel: index of the FIRST box (ascending) holding each point, -1 = background (caller pre-fills) */
ORC_API void orc_points_in_boxes(int batch_size, int boxes_num, int pts_num, const float *boxes, const float *pts,
                                 int *box_idx_of_points) {
    float lx, ly;
    for (int b = 0; b < batch_size; ++b)
        for (int p = 0; p < pts_num; ++p)
            for (int k = 0; k < boxes_num; ++k)
                if (orc_pt_in_box3d(pts + ((size_t)b * pts_num + p) * 3, boxes + ((size_t)b * boxes_num + k) * 7, 1e-5f, &lx, &ly)) {
                    box_idx_of_points[(size_t)b * pts_num + p] = k;
                    break;
                }
}

/* roipool3dLauncher: the three kernels in sequence.  pooled_features (B, M, S, 3 + C) and pooled_empty_flag (B, M) are
 * pre-zeroed by the caller (roipoint_pool3d_utils.py:50-51); rows of empty boxes are left untouched. */
ORC_API void orc_roipoint_pool3d(int batch_size, int pts_num, int boxes_num, int feature_in_len, int sampled_pts_num,
                                 const float *xyz, const float *boxes3d, const float *pts_feature, float *pooled_features,
                                 int *pooled_empty_flag) {
    float lx, ly;
    int *pts_idx = (int *)malloc(sizeof(int) * (size_t)(sampled_pts_num > 0 ? sampled_pts_num : 1));
    for (int b = 0; b < batch_size; ++b)
        for (int m = 0; m < boxes_num; ++m) {
            const float *box = boxes3d + ((size_t)b * boxes_num + m) * 7;
            int cnt = 0;
            for (int k = 0; k < pts_num && cnt < sampled_pts_num; ++k)
                if (orc_pt_in_box3d(xyz + ((size_t)b * pts_num + k) * 3, box, 1e-5f, &lx, &ly)) pts_idx[cnt++] = k;
            if (cnt == 0) {
                pooled_empty_flag[(size_t)b * boxes_num + m] = 1;
                continue;
            }
            for (int k = cnt; k < sampled_pts_num; ++k) pts_idx[k] = pts_idx[k % cnt];
            for (int s = 0; s < sampled_pts_num; ++s) {
                float *dst = pooled_features + (((size_t)b * boxes_num + m) * sampled_pts_num + s) * (3 + feature_in_len);
                const int src = pts_idx[s];
                for (int j = 0; j < 3; ++j) dst[j] = xyz[((size_t)b * pts_num + src) * 3 + j];
                for (int j = 0; j < feature_in_len; ++j) dst[3 + j] = pts_feature[((size_t)b * pts_num + src) * feature_in_len + j];
            }
        }
    free(pts_idx);
}
