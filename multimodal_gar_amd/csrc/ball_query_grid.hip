// ball_query_grid.hip -- ball query through a uniform cell grid (batch + stack layouts) for gfx950.
//
// Same results as csrc/ball_query.hip (reference pointnet2_batch/src/ball_query_gpu.cu:15-51, pointnet2_stack/src/
// ball_query_gpu.cu:16-66), bit for bit: the first nsample indices k in ASCENDING order with d2 < r2, the row padded with
// its first hit, untouched (batch) / idx[0] = -1 (stack) when the ball is empty.
//
// The scan kernel tests every (query, point) pair -- 7 VALU each, at the vector peak already -- which is the right thing
// where rows fill up and the scan stops early (the trunk's FPS centres).  The RoI-grid queries rarely fill their balls:
// every scan runs to the end of the cloud, 1.1 ms per radius at config c3 and 26 ms at c5 (128 actors x 65 536 points).
// Here the points of a cloud are binned once into cells of edge h (counting sort with integer atomics; the order inside a
// cell is irrelevant), a query visits only the cells its ball can touch, and keeps the nsample SMALLEST hit indices in a
// sorted register list (min / max insertion, 2 VALU per slot and hit) -- "first nsample in index order" is a property of
// the hit SET, so any visiting order gives the reference's row.  ~130 candidates per query instead of 16 384 at c3.
//   * the distance test is the same expression on the same operands (d2_of(q - p), strict <);
//   * the cell range of a query is computed from q -+ r(1 + 1e-5) with the same monotone fp32 map that bins the points,
//     so no point inside the ball can fall outside the visited range;
//   * cells are x-fastest: the cells (cz, cy, x0..x1) of a query are ONE contiguous run of the sorted point array.
#include "common.hpp"

namespace mgar {

constexpr int PG_MAX_CELLS = 32768;   // cells per cloud (the cell edge grows until the grid fits)
constexpr int PG_GEOM_INTS = 16;      // per cloud: lo[3], inv_h, dim[3], ncell, pstart, n, h
constexpr int BQG_THREADS = 256;
constexpr int BQG_ROW_STRIDE = BQG_THREADS + 1;

struct PgLayout {   // byte offsets inside the workspace (all 16-byte aligned)
    size_t geom, cell_start, cursor, cell_of, sorted, total;
};
__host__ __device__ inline size_t pg_align(size_t v) { return (v + 15) & ~(size_t)15; }
static PgLayout pg_layout(int B, long long n_total) {
    PgLayout l;
    size_t o = 0;
    l.geom = o; o = pg_align(o + (size_t)B * PG_GEOM_INTS * 4);
    l.cell_start = o; o = pg_align(o + (size_t)B * (PG_MAX_CELLS + 1) * 4);
    l.cursor = o; o = pg_align(o + (size_t)B * PG_MAX_CELLS * 4);
    l.cell_of = o; o = pg_align(o + (size_t)n_total * 4);
    l.sorted = o; o = pg_align(o + (size_t)n_total * 16);
    l.total = o;
    return l;
}

struct PgGeom {
    float lo[3];
    float inv_h;
    int dim[3];
    int ncell, pstart, n;
    float h;
    int pad[5];
};
static_assert(sizeof(PgGeom) == PG_GEOM_INTS * 4, "PgGeom layout");

// the one map from a coordinate to a cell index (monotone non-decreasing in v: fp32 subtract, multiply by a positive constant, floor)
__device__ __forceinline__ int pg_cell(float v, float lo, float inv_h) {
    const float t = floorf((v - lo) * inv_h);
    return (int)fminf(fmaxf(t, -1048576.f), 1048576.f);
}

// K1: per cloud -- start row, bounding box, grid dimensions; zero the cell counters.  grid (B), block 1024
__global__ __launch_bounds__(1024) void pg_geom_kernel(int n_batch, const float *__restrict__ xyz, const int *__restrict__ xyz_batch_cnt,
                                                       float cell, PgGeom *__restrict__ geom, int *__restrict__ cursor) {
    __shared__ float red[6][16];
    const int b = blockIdx.x;
    int pstart, n;
    if (xyz_batch_cnt) {
        pstart = 0;
        for (int i = 0; i < b; ++i) pstart += xyz_batch_cnt[i];
        n = xyz_batch_cnt[b];
    } else {
        pstart = b * n_batch;
        n = n_batch;
    }
    const float *P = xyz + (size_t)pstart * 3;
    const float inf = __builtin_inff();
    float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
    for (int k = threadIdx.x; k < n; k += 1024)
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = P[k * 3 + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float l = -wave_max(-lo[a]), h = wave_max(hi[a]);
        if (lane == 0) { red[a][wave] = l; red[3 + a][wave] = h; }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = inf, h = -inf;
        for (int w = 0; w < 16; ++w) { l = fminf(l, red[a][w]); h = fmaxf(h, red[3 + a][w]); }
        lo[a] = n > 0 ? l : 0.f;
        hi[a] = n > 0 ? h : 0.f;
    }
    // cell edge: the caller's, enlarged until the grid has at most PG_MAX_CELLS cells (dim = floor(ext / h) + 1 per axis:
    // every point's cell index is then inside [0, dim) without clamping)
    float h = cell;
    if (!(h > 0.f)) {
        // automatic: ~|cell| (default 4) points per cell of the occupied box; flat or degenerate boxes fall back on their largest extent
        const float target = cell < 0.f ? -cell : 4.f;
        const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
        const float emax = fmaxf(fmaxf(ex, ey), fmaxf(ez, 1e-6f));
        const float vol = fmaxf(ex, 1e-3f * emax) * fmaxf(ey, 1e-3f * emax) * fmaxf(ez, 1e-3f * emax);
        h = cbrtf(vol * target / (float)(n > 0 ? n : 1));
        h = fmaxf(h, 1e-6f * emax);
    }
    int d[3];
    for (int it = 0; it < 200; ++it) {
        const float ih = 1.f / h;
        long long c = 1;
#pragma unroll
        for (int a = 0; a < 3; ++a) { d[a] = pg_cell(hi[a], lo[a], ih) + 1; c *= d[a]; }   // cell(v) <= cell(hi) = dim - 1 for every point v
        if (c <= PG_MAX_CELLS) break;
        h *= 1.25f;
    }
    const int ncell = d[0] * d[1] * d[2];
    if (threadIdx.x == 0) {
        PgGeom g;
        g.lo[0] = lo[0]; g.lo[1] = lo[1]; g.lo[2] = lo[2];
        g.inv_h = 1.f / h; g.h = h;
        g.dim[0] = d[0]; g.dim[1] = d[1]; g.dim[2] = d[2];
        g.ncell = ncell; g.pstart = pstart; g.n = n;
        for (int i = 0; i < 5; ++i) g.pad[i] = 0;
        geom[b] = g;
    }
    int *cnt = cursor + (size_t)b * PG_MAX_CELLS;
    for (int c = threadIdx.x; c < ncell; c += 1024) cnt[c] = 0;
}

// cloud of global point row p (stack layout: binary search over the clouds' start rows)
__device__ __forceinline__ int pg_cloud_of(long long p, int B, int n_batch, const PgGeom *geom) {
    if (n_batch > 0) return (int)(p / n_batch);
    int lo = 0, hi = B - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (geom[mid].pstart <= p) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// K2: cell of every point + histogram.  grid over all points, block 256
__global__ __launch_bounds__(256) void pg_count_kernel(long long n_total, int B, int n_batch, const float *__restrict__ xyz,
                                                       const PgGeom *__restrict__ geom, int *__restrict__ cursor, int *__restrict__ cell_of) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_total) return;
    const int b = pg_cloud_of(p, B, n_batch, geom);
    const PgGeom g = geom[b];
    if (p >= (long long)g.pstart + g.n) return;   // stack layout: rows beyond the last cloud's count
    const int cx = pg_cell(xyz[p * 3 + 0], g.lo[0], g.inv_h), cy = pg_cell(xyz[p * 3 + 1], g.lo[1], g.inv_h);
    const int cz = pg_cell(xyz[p * 3 + 2], g.lo[2], g.inv_h);
    const int c = (cz * g.dim[1] + cy) * g.dim[0] + cx;
    cell_of[p] = c;
    atomicAdd(cursor + (size_t)b * PG_MAX_CELLS + c, 1);
}

// K3: exclusive scan of a cloud's histogram -> cell_start (ncell + 1), cursor = cell_start.  grid (B), block 1024
__global__ __launch_bounds__(1024) void pg_scan_kernel(const PgGeom *__restrict__ geom, int *__restrict__ cursor, int *__restrict__ cell_start) {
    __shared__ int part[1024];
    const int b = blockIdx.x, t = threadIdx.x;
    const int ncell = geom[b].ncell;
    int *cnt = cursor + (size_t)b * PG_MAX_CELLS;
    int *start = cell_start + (size_t)b * (PG_MAX_CELLS + 1);
    constexpr int PER = PG_MAX_CELLS / 1024;   // 32 consecutive cells per thread
    int v[PER], s = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) { const int c = t * PER + i; v[i] = c < ncell ? cnt[c] : 0; s += v[i]; }
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan of the 1 024 partial sums
        const int add = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    int run = part[t] - s;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = t * PER + i;
        if (c < ncell) { start[c] = run; cnt[c] = run; }
        run += v[i];
    }
    if (t == 1023) start[ncell] = part[1023];
}

// K4: scatter {x, y, z, k} into cell order (k = index inside the cloud).  grid over all points, block 256
__global__ __launch_bounds__(256) void pg_scatter_kernel(long long n_total, int B, int n_batch, const float *__restrict__ xyz,
                                                         const PgGeom *__restrict__ geom, const int *__restrict__ cell_of,
                                                         int *__restrict__ cursor, float4 *__restrict__ sorted) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_total) return;
    const int b = pg_cloud_of(p, B, n_batch, geom);
    const int pstart = geom[b].pstart, n = geom[b].n;
    if (p >= (long long)pstart + n) return;
    const int pos = atomicAdd(cursor + (size_t)b * PG_MAX_CELLS + cell_of[p], 1);
    sorted[(size_t)pstart + pos] = make_float4(xyz[p * 3 + 0], xyz[p * 3 + 1], xyz[p * 3 + 2], __int_as_float((int)(p - pstart)));
}

// ---- the query --------------------------------------------------------------------------------------------------------
template <bool STACK, int NS>
__global__ __launch_bounds__(BQG_THREADS) void ball_query_grid_kernel(int B, int m_batch, float radius, float radius2, int nsample,
                                                                      const float *__restrict__ new_xyz,
                                                                      const int *__restrict__ new_xyz_batch_cnt,
                                                                      const PgGeom *__restrict__ geom, const int *__restrict__ cell_start,
                                                                      const float4 *__restrict__ sorted, int *__restrict__ idx) {
    extern __shared__ int lds[];  // [nsample][257] rows, then [256] counts (coalesced write-out, as csrc/ball_query.hip)
    int *rows = lds;
    int *cnts = lds + nsample * BQG_ROW_STRIDE;
    int q0, q_end, bs;
    if (STACK) {
        int g = blockIdx.x, qs = 0;
        bool found = false;
        for (bs = 0; bs < B; ++bs) {
            const int mi = new_xyz_batch_cnt[bs];
            const int nb = (mi + BQG_THREADS - 1) / BQG_THREADS;
            if (g < nb) { found = true; break; }
            g -= nb;
            qs += mi;
        }
        if (!found) return;
        q0 = qs + g * BQG_THREADS;
        q_end = qs + new_xyz_batch_cnt[bs];
    } else {
        bs = blockIdx.y;
        q0 = bs * m_batch + blockIdx.x * BQG_THREADS;
        q_end = (bs + 1) * m_batch;
    }
    const PgGeom g = geom[bs];
    const int *__restrict__ start = cell_start + (size_t)bs * (PG_MAX_CELLS + 1);
    const float4 *__restrict__ pts = sorted + g.pstart;
    const int tid = threadIdx.x;
    const int q = q0 + tid;
    const bool valid = q < q_end;
    int list[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) list[s] = 0x7FFFFFFF;
    int cnt = 0;
    if (valid && g.n > 0) {
        const float qx = new_xyz[(size_t)q * 3 + 0], qy = new_xyz[(size_t)q * 3 + 1], qz = new_xyz[(size_t)q * 3 + 2];
        const float rr = radius * 1.00001f + 1e-30f;   // the cell range may only be too wide, never too narrow
        const int x0 = max(pg_cell(qx - rr, g.lo[0], g.inv_h), 0), x1 = min(pg_cell(qx + rr, g.lo[0], g.inv_h), g.dim[0] - 1);
        const int y0 = max(pg_cell(qy - rr, g.lo[1], g.inv_h), 0), y1 = min(pg_cell(qy + rr, g.lo[1], g.inv_h), g.dim[1] - 1);
        const int z0 = max(pg_cell(qz - rr, g.lo[2], g.inv_h), 0), z1 = min(pg_cell(qz + rr, g.lo[2], g.inv_h), g.dim[2] - 1);
        auto hit = [&](const float4 v) {
            const float d2 = d2_of(qx - v.x, qy - v.y, qz - v.z);
            if (d2 < radius2) {
                int nv = __float_as_int(v.w);
                ++cnt;
                if (nv < list[NS - 1]) {   // once the list is full of small indices most hits stop here
#pragma unroll
                    for (int s = 0; s < NS; ++s) {   // sorted insertion: list keeps the NS smallest indices, ascending
                        const int a = list[s];
                        list[s] = min(a, nv);
                        nv = max(a, nv);
                    }
                }
            }
        };
        // the (cz, cy) rows of the cell range, flattened; every row is ONE contiguous run [start(x0), start(x1 + 1)) of the
        // sorted points.  The next row's two cell_start loads are issued before the current run is walked, and a run is walked
        // four points at a time (independent 16-byte loads in flight), so a lane is not a chain of dependent cache misses.
        const int ny = y1 - y0 + 1, rows_n = (x0 <= x1 && y0 <= y1 && z0 <= z1) ? ny * (z1 - z0 + 1) : 0;
        const float4 far = make_float4(__builtin_inff(), 0.f, 0.f, 0.f);   // d2 = inf: never a hit
        int ps = 0, pe = 0;
        if (rows_n > 0) {
            const int base = (z0 * g.dim[1] + y0) * g.dim[0];
            ps = start[base + x0]; pe = start[base + x1 + 1];
        }
        for (int t = 0; t < rows_n; ++t) {
            int ns_ = 0, ne_ = 0;
            if (t + 1 < rows_n) {
                const int tz = (t + 1) / ny, ty = (t + 1) - tz * ny;
                const int base = ((z0 + tz) * g.dim[1] + (y0 + ty)) * g.dim[0];
                ns_ = start[base + x0]; ne_ = start[base + x1 + 1];
            }
            for (int p = ps; p < pe; p += 4) {
                const float4 v0 = pts[p];
                const float4 v1 = p + 1 < pe ? pts[p + 1] : far;
                const float4 v2 = p + 2 < pe ? pts[p + 2] : far;
                const float4 v3 = p + 3 < pe ? pts[p + 3] : far;
                hit(v0); hit(v1); hit(v2); hit(v3);
            }
            ps = ns_; pe = ne_;
        }
    }
    const int c = min(cnt, nsample);
#pragma unroll
    for (int s = 0; s < NS; ++s)
        if (s < c) rows[s * BQG_ROW_STRIDE + tid] = list[s];
    cnts[tid] = valid ? c : -1;
    __syncthreads();
    const int nq = min(BQG_THREADS, q_end - q0);
    const int total = nq * nsample;
    int *out = idx + (size_t)q0 * nsample;
    for (int e = tid; e < total; e += BQG_THREADS) {
        const int ql = e / nsample, s = e - ql * nsample;
        const int cc = cnts[ql];
        if (cc > 0) {
            out[e] = rows[(s < cc ? s : 0) * BQG_ROW_STRIDE + ql];
        } else if (STACK && s == 0) {
            out[e] = -1;
        }
    }
}

// ---- three nearest neighbours through the grid --------------------------------------------------------------------------------
// three_nn of csrc/interpolate.hip (reference pointnet2_batch/src/interpolate_gpu.cu:16-59, stack :16-81), bit for bit: the
// reference's strict-'<' cascade in index order keeps, among equal distances, the EARLIEST index, i.e. its result is the three
// smallest (d2, k) pairs in lexicographic order -- a property of the candidate SET, so the cells may be visited in any order as
// long as every point that could enter the top three is seen.  Shells of cells of growing Chebyshev radius s around the query's
// cell are walked until the third-best distance is strictly smaller than the distance from the query to the boundary of the cube
// of shells 0..s (every unvisited point lies outside that cube); missing neighbours keep (inf, 0) as the reference (1e40 -> inf).
__device__ __forceinline__ void tn_consider(float d, int k, float &b1, float &b2, float &b3, int &i1, int &i2, int &i3) {
    if (d < b3 || (d == b3 && k < i3)) {
        if (d < b2 || (d == b2 && k < i2)) {
            b3 = b2; i3 = i2;
            if (d < b1 || (d == b1 && k < i1)) { b2 = b1; i2 = i1; b1 = d; i1 = k; }
            else { b2 = d; i2 = k; }
        } else { b3 = d; i3 = k; }
    }
}

template <bool STACK>
__global__ __launch_bounds__(BQG_THREADS) void three_nn_grid_kernel(int B, int n_batch, const float *__restrict__ unknown,
                                                                    const int *__restrict__ unknown_batch_cnt,
                                                                    const PgGeom *__restrict__ geom, const int *__restrict__ cell_start,
                                                                    const float4 *__restrict__ sorted, float *__restrict__ dist2,
                                                                    int *__restrict__ idx) {
    int u0, u_end, bs;
    if (STACK) {
        int g = blockIdx.x, us = 0;
        bool found = false;
        for (bs = 0; bs < B; ++bs) {
            const int ni = unknown_batch_cnt[bs];
            const int nb = (ni + BQG_THREADS - 1) / BQG_THREADS;
            if (g < nb) { found = true; break; }
            g -= nb;
            us += ni;
        }
        if (!found) return;
        u0 = us + g * BQG_THREADS;
        u_end = us + unknown_batch_cnt[bs];
    } else {
        bs = blockIdx.y;
        u0 = bs * n_batch + blockIdx.x * BQG_THREADS;
        u_end = (bs + 1) * n_batch;
    }
    const int u = u0 + threadIdx.x;
    if (u >= u_end) return;
    const PgGeom g = geom[bs];
    const int *__restrict__ start = cell_start + (size_t)bs * (PG_MAX_CELLS + 1);
    const float4 *__restrict__ pts = sorted + g.pstart;
    const float ux = unknown[(size_t)u * 3 + 0], uy = unknown[(size_t)u * 3 + 1], uz = unknown[(size_t)u * 3 + 2];
    const float inf = __builtin_inff();
    float b1 = inf, b2 = inf, b3 = inf;
    int i1 = 0, i2 = 0, i3 = 0;
    if (g.n > 0) {
        const int dx = g.dim[0], dy = g.dim[1], dz = g.dim[2];
        const int cx = min(max(pg_cell(ux, g.lo[0], g.inv_h), 0), dx - 1), cy = min(max(pg_cell(uy, g.lo[1], g.inv_h), 0), dy - 1);
        const int cz = min(max(pg_cell(uz, g.lo[2], g.inv_h), 0), dz - 1);
        auto run = [&](int base, int xa, int xb) {   // cells base + xa .. base + xb: one contiguous run of the sorted points
            const int e = start[base + xb + 1];
            for (int p = start[base + xa]; p < e; ++p) {
                const float4 v = pts[p];
                tn_consider(d2_of(ux - v.x, uy - v.y, uz - v.z), __float_as_int(v.w), b1, b2, b3, i1, i2, i3);
            }
        };
        const int smax = max(max(max(cx, dx - 1 - cx), max(cy, dy - 1 - cy)), max(cz, dz - 1 - cz));
        for (int s = 0; s <= smax; ++s) {
            const int x0 = max(cx - s, 0), x1 = min(cx + s, dx - 1), y0 = max(cy - s, 0), y1 = min(cy + s, dy - 1);
            const int z0 = max(cz - s, 0), z1 = min(cz + s, dz - 1);
            for (int z = z0; z <= z1; ++z) {
                const bool zface = (z == cz - s) || (z == cz + s);
                for (int y = y0; y <= y1; ++y) {
                    const int base = (z * dy + y) * dx;
                    if (zface || y == cy - s || y == cy + s) {
                        run(base, x0, x1);                                   // a whole row of the shell's z / y faces
                    } else {
                        if (cx - s >= 0) run(base, cx - s, cx - s);          // the two x end cells
                        if (cx + s <= dx - 1) run(base, cx + s, cx + s);
                    }
                }
            }
            // every unvisited point lies outside the cube of cells [c - s, c + s]: along at least one axis it is beyond the cube's
            // face, unless the cube already reaches the end of the grid on that side
            float bound = inf;
            const float q[3] = {ux, uy, uz};
            const int c[3] = {cx, cy, cz}, dim[3] = {dx, dy, dz};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if (c[a] - s > 0) bound = fminf(bound, q[a] - (g.lo[a] + (float)(c[a] - s) * g.h));
                if (c[a] + s < dim[a] - 1) bound = fminf(bound, (g.lo[a] + (float)(c[a] + s + 1) * g.h) - q[a]);
            }
            if (bound == inf) break;                                         // the cube covers the grid
            // the cell map floor((v - lo) * inv_h) and these face coordinates lo + c * h round differently: by at most ~2e-7 * dim
            // cells (relative rounding of a value <= dim), so a margin of 1e-6 * dim cells is safe
            bound -= g.h * (1e-6f * (float)max(dx, max(dy, dz)) + 1e-5f);
            if (bound > 0.f && b3 < bound * bound) break;
        }
    }
    const int off = STACK ? g.pstart : 0;   // the stack op returns global rows (interpolate_gpu.cu:72-74)
    dist2[(size_t)u * 3 + 0] = b1; dist2[(size_t)u * 3 + 1] = b2; dist2[(size_t)u * 3 + 2] = b3;
    idx[(size_t)u * 3 + 0] = i1 + off; idx[(size_t)u * 3 + 1] = i2 + off; idx[(size_t)u * 3 + 2] = i3 + off;
}

}  // namespace mgar

using namespace mgar;

#define BQG_API extern "C" __attribute__((visibility("default")))

BQG_API long long mgar_point_grid_workspace_bytes(int B, long long n_total) {
    if (B < 0 || n_total < 0) return -1;
    return (long long)pg_layout(B, n_total).total;
}

// Bins the points of B clouds into cells of edge `cell` (enlarged per cloud until it has at most 32 768 cells); cell <= 0: chosen
// per cloud so that a cell of the occupied box holds about -cell (0: four) points.
// Batch layout: n_batch > 0 points per cloud, xyz_batch_cnt == NULL; stack layout: n_batch == 0, xyz_batch_cnt (B) on the device.
// n_total = rows of xyz.  workspace: mgar_point_grid_workspace_bytes(B, n_total) bytes, 16-byte aligned; it is what
// mgar_ball_query_grid_* take, valid for as long as xyz is unchanged.
BQG_API int mgar_point_grid_build(int B, int n_batch, long long n_total, const float *xyz, const int *xyz_batch_cnt, float cell,
                                  void *workspace, void *stream) {
    MGAR_REQUIRE(B >= 0 && n_batch >= 0 && n_total >= 0 && cell == cell, "point_grid_build: bad sizes");
    if (B == 0 || n_total == 0) return MGAR_OK;
    MGAR_REQUIRE(xyz && workspace && (n_batch > 0 || xyz_batch_cnt), "point_grid_build: null pointer");
    MGAR_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "point_grid_build: workspace must be 16-byte aligned");
    MGAR_REQUIRE(n_batch == 0 || (long long)B * n_batch == n_total, "point_grid_build: batch layout needs n_total == B * n");
    MGAR_REQUIRE(n_total / 256 < 2147483647LL, "point_grid_build: too many points");
    hipStream_t st = (hipStream_t)stream;
    const PgLayout l = pg_layout(B, n_total);
    char *ws = (char *)workspace;
    PgGeom *geom = (PgGeom *)(ws + l.geom);
    int *cell_start = (int *)(ws + l.cell_start), *cursor = (int *)(ws + l.cursor), *cell_of = (int *)(ws + l.cell_of);
    float4 *sorted = (float4 *)(ws + l.sorted);
    // points read twice (bounding box, binning) + 4 B cell id written and read + the 16-byte sorted record written
    KtScope kt(KT_POINT_GRID, st, (double)n_total * (24.0 + 8.0 + 16.0));
    const int blocks = (int)((n_total + 255) / 256);
    hipLaunchKernelGGL(pg_geom_kernel, dim3(B), dim3(1024), 0, st, n_batch, xyz, n_batch > 0 ? nullptr : xyz_batch_cnt, cell, geom, cursor);
    hipLaunchKernelGGL(pg_count_kernel, dim3(blocks), dim3(256), 0, st, n_total, B, n_batch, xyz, geom, cursor, cell_of);
    hipLaunchKernelGGL(pg_scan_kernel, dim3(B), dim3(1024), 0, st, geom, cursor, cell_start);
    hipLaunchKernelGGL(pg_scatter_kernel, dim3(blocks), dim3(256), 0, st, n_total, B, n_batch, xyz, geom, cell_of, cursor, sorted);
    return check_launch("point_grid_build: launch failed");
}

template <bool STACK>
static int bqg_launch(int B, int m_batch, long long m_total, long long n_total, float radius, int nsample, const float *new_xyz,
                      const int *new_cnt, const void *workspace, int *idx, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const PgLayout l = pg_layout(B, n_total);
    const char *ws = (const char *)workspace;
    const PgGeom *geom = (const PgGeom *)(ws + l.geom);
    const int *cell_start = (const int *)(ws + l.cell_start);
    const float4 *sorted = (const float4 *)(ws + l.sorted);
    const size_t lds = ((size_t)nsample * BQG_ROW_STRIDE + BQG_THREADS) * sizeof(int);
    dim3 grid = STACK ? dim3((unsigned)((m_total + BQG_THREADS - 1) / BQG_THREADS + B)) : dim3(ceil_div(m_batch, BQG_THREADS), B);
    // 12 N + 12 M + 4 M nsample; the pair tests depend on the data (credited by the caller)
    KtScope kt(KT_BALL_QUERY_GRID, st, 12.0 * (double)n_total + (double)m_total * (12.0 + 4.0 * nsample));
#define BQG_GO(NS)                                                                                                             \
    do {                                                                                                                       \
        if (lds > 65536) (void)hipFuncSetAttribute((const void *)ball_query_grid_kernel<STACK, NS>,                             \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
        hipLaunchKernelGGL((ball_query_grid_kernel<STACK, NS>), grid, dim3(BQG_THREADS), lds, st, B, m_batch, radius, radius * radius, \
                           nsample, new_xyz, new_cnt, geom, cell_start, sorted, idx);                                           \
    } while (0)
    if (nsample <= 16) BQG_GO(16);
    else if (nsample <= 32) BQG_GO(32);
    else BQG_GO(64);
#undef BQG_GO
    return check_launch("ball_query_grid: launch failed");
}

// ball query of mgar_ball_query_batch / _stack through the grid of mgar_point_grid_build (same xyz, same B / n).  nsample <= 64.
BQG_API int mgar_ball_query_grid_batch(int b, int n, int m, float radius, int nsample, const float *new_xyz, const void *grid, int *idx,
                                       void *stream) {
    MGAR_REQUIRE(b >= 0 && n >= 0 && m >= 0 && radius >= 0.f, "ball_query_grid_batch: bad sizes");
    if (nsample < 1 || nsample > 64) { set_error("ball_query_grid: nsample outside [1, 64] (use mgar_ball_query_batch)"); return MGAR_EUNSUPPORTED; }
    if (b == 0 || m == 0) return MGAR_OK;
    MGAR_REQUIRE(new_xyz && idx && (grid || n == 0), "ball_query_grid_batch: null pointer");
    if (n == 0) return MGAR_OK;
    return bqg_launch<false>(b, m, (long long)b * m, (long long)b * n, radius, nsample, new_xyz, nullptr, grid, idx, stream);
}
BQG_API int mgar_ball_query_grid_stack(int B, int M, long long n_total, float radius, int nsample, const float *new_xyz,
                                       const int *new_xyz_batch_cnt, const void *grid, int *idx, void *stream) {
    MGAR_REQUIRE(B >= 0 && M >= 0 && n_total >= 0 && radius >= 0.f, "ball_query_grid_stack: bad sizes");
    if (nsample < 1 || nsample > 64) { set_error("ball_query_grid: nsample outside [1, 64] (use mgar_ball_query_stack)"); return MGAR_EUNSUPPORTED; }
    if (B == 0 || M == 0) return MGAR_OK;
    MGAR_REQUIRE(new_xyz && new_xyz_batch_cnt && idx, "ball_query_grid_stack: null pointer");
    if (n_total == 0) {   // every ball is empty
        set_error("ball_query_grid_stack: empty point set (use mgar_ball_query_stack)");
        return MGAR_EUNSUPPORTED;
    }
    MGAR_REQUIRE(grid, "ball_query_grid_stack: null grid");
    return bqg_launch<true>(B, 0, M, n_total, radius, nsample, new_xyz, new_xyz_batch_cnt, grid, idx, stream);
}

// three_nn of mgar_three_nn_batch / _stack through a grid built over the KNOWN points (mgar_point_grid_build(known ...); cell <= 0
// lets the library size the cells).  dist2 (.., 3) squared distances, idx (.., 3) (stack: global rows), as the scan kernels.
BQG_API int mgar_three_nn_grid_batch(int b, int n, int m, const float *unknown, const void *grid, float *dist2, int *idx, void *stream) {
    MGAR_REQUIRE(b >= 0 && n >= 0 && m >= 0, "three_nn_grid_batch: negative size");
    if (b == 0 || n == 0) return MGAR_OK;
    MGAR_REQUIRE(unknown && dist2 && idx && grid && m > 0, "three_nn_grid_batch: null pointer or no known points (use mgar_three_nn_batch)");
    hipStream_t st = (hipStream_t)stream;
    const PgLayout l = pg_layout(b, (long long)b * m);
    const char *ws = (const char *)grid;
    KtScope kt(KT_THREE_NN_GRID, st, (double)b * (12.0 * n + 12.0 * m + 24.0 * n));
    hipLaunchKernelGGL(three_nn_grid_kernel<false>, dim3(ceil_div(n, BQG_THREADS), b), dim3(BQG_THREADS), 0, st, b, n, unknown, (const int *)nullptr,
                       (const PgGeom *)(ws + l.geom), (const int *)(ws + l.cell_start), (const float4 *)(ws + l.sorted), dist2, idx);
    return check_launch("three_nn_grid_batch: launch failed");
}
BQG_API int mgar_three_nn_grid_stack(int B, int N, long long m_total, const float *unknown, const int *unknown_batch_cnt, const void *grid,
                                     float *dist2, int *idx, void *stream) {
    MGAR_REQUIRE(B >= 0 && N >= 0 && m_total >= 0, "three_nn_grid_stack: negative size");
    if (B == 0 || N == 0) return MGAR_OK;
    MGAR_REQUIRE(unknown && unknown_batch_cnt && dist2 && idx && grid && m_total > 0,
                 "three_nn_grid_stack: null pointer or no known points (use mgar_three_nn_stack)");
    hipStream_t st = (hipStream_t)stream;
    const PgLayout l = pg_layout(B, m_total);
    const char *ws = (const char *)grid;
    KtScope kt(KT_THREE_NN_GRID, st, 12.0 * N + 12.0 * (double)m_total + 24.0 * N);
    hipLaunchKernelGGL(three_nn_grid_kernel<true>, dim3(ceil_div(N, BQG_THREADS) + B), dim3(BQG_THREADS), 0, st, B, 0, unknown, unknown_batch_cnt,
                       (const PgGeom *)(ws + l.geom), (const int *)(ws + l.cell_start), (const float4 *)(ws + l.sorted), dist2, idx);
    return check_launch("three_nn_grid_stack: launch failed");
}
