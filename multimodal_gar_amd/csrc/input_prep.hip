// input_prep.hip -- the per-clip input preparation of the JRDB loader on gfx950.
//
// Replaces, for a whole clip at once and with the decoded bytes already in HBM,
//   dataloader.py:47-49     transforms.Resize(image_size) + ToTensor() + Normalize(mean, std) on every stitched frame
//                           (Resize on a PIL image = Pillow's Image.resize(size, BILINEAR): third-party, un-vendored;
//                           algorithm restated in oracle/oracle.py::pil_bilinear_resize, pinned against Pillow itself)
//   dataloader.py:119-128   load_pc: upper / lower velodyne clouds moved to the base frame and concatenated
//   pcdet/datasets/processor/data_processor.py:78-84 + pcdet/utils/common_utils.py:60-63
//                           mask_points_and_boxes_outside_range: x / y range mask, points kept in order
//
// The reference does this per frame on the host (Pillow's C loops, then three float32 passes of torch over the frame) and
// ships float32 frames over PCIe.  Here the uint8 frames are uploaded (a quarter of the bytes) and ONE kernel per clip
// reads every source byte once, resamples in Pillow's exact 22-bit fixed point (horizontal pass into LDS as bytes, vertical
// pass out of LDS), maps the byte through a 3 x 256 float table that holds ((v / 255) - mean) / std evaluated exactly as
// torch does, and writes the network's layout directly (frame and channel strides are arguments: (T, 3, H, W) as the
// loader returns it, or (3, T, H, W) as the I3D trunk wants it -- no permute pass).  HBM-bound byte work: no MFMA.
//
// The range crop is an ordered stream compaction: per-1024-point counts (ballot + popcount), one scan of those counts,
// then every point writes itself at its rank.  Output order = input order, as boolean-mask indexing gives.
#include <math.h>
#include "common.hpp"
#include "payload.hpp"

namespace mgar {

constexpr int IP_THREADS = 256;
constexpr int IP_TW = 128;                   // output columns per workgroup
constexpr int IP_BITS = 32 - 8 - 2;          // Pillow's PRECISION_BITS for 8-bit channels
constexpr int IP_LDS_BUDGET = 60 * 1024;

// pixel (8 bits) x weight (<= 2^22, never negative for the triangle filter): a 24-bit multiply-add runs at full rate, the
// 32-bit v_mul_lo_u32 the plain expression compiles to at a quarter of it
__device__ __forceinline__ int tap(int acc, unsigned pixel, int weight) {
    return acc + (int)__umul24(pixel, (unsigned)weight);
}

__device__ __forceinline__ int clip8(int acc) {
    const int v = acc >> IP_BITS;            // arithmetic shift, then clamp: Pillow's clip8 lookup
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

struct ResizeArgs {
    const unsigned char *src;                // (frames, in_h, in_w, 3)
    const int *xb, *xk, *yb, *yk;            // bounds (out, 2) [first tap, taps], weights (out, ksize)
    const float *lut;                        // (3, 256)
    void *dst;
    long long dst_fs, dst_cs;                // element strides of frame and channel (fp32 / bf16 output)
    long long src_bytes;
    int in_h, in_w, out_h, out_w, xksize, yksize;
    int th, rows_cap, raw_stride;            // output rows per workgroup; LDS row capacity; LDS bytes per staged source row
    int vec4;                                // four neighbouring outputs of a row can go out as one 16-byte (8-byte bf16) store
    double sx, supx, sy, supy;               // scale and filter support per axis, as precompute_coeffs forms them
};

// [first tap, first tap + taps) of output index i: the bounds row of the coefficient table, recomputed (same double expressions
// as resample_coeffs_host) so that a workgroup knows its tile's source extent without a dependent load
__device__ __forceinline__ void tap_range(int i, int in_size, int out_size, double scale, double support, int &lo, int &hi) {
    if (in_size == out_size) { lo = i; hi = i + 1; return; }
    const double center = (i + 0.5) * scale;
    lo = (int)(center - support + 0.5);
    if (lo < 0) lo = 0;
    hi = (int)(center + support + 0.5);
    if (hi > in_size) hi = in_size;
}

constexpr int IP_KX = 8;                     // horizontal taps a thread keeps in registers (ksize <= 8: down-scaling below 3.5x)

// OUT: 0 float32 planes, 1 bf16 planes, 2 the resampled bytes themselves, (frames, out_h, out_w, 3)
// KXR: the horizontal weights of a thread's column live in IP_KX registers (else they are read from the LDS copy)
//
// LDS: lut[768] | ytab[th][2 + yksize] | xtab[IP_TW][xksize] (only !KXR) | raw[rows_cap][raw_stride] | mid[rows_cap][3][IP_TW]
template <int OUT, bool KXR>
__global__ __launch_bounds__(IP_THREADS) void image_resize_normalize_kernel(ResizeArgs a) {
    extern __shared__ unsigned char lds[];
    float *lut = reinterpret_cast<float *>(lds);
    int *ytab = reinterpret_cast<int *>(lds + 768 * sizeof(float));
    int *xtab = ytab + a.th * (2 + a.yksize);
    unsigned char *raw = reinterpret_cast<unsigned char *>(xtab + (KXR ? 0 : IP_TW * a.xksize));
    unsigned char *mid = raw + (size_t)a.rows_cap * a.raw_stride;
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * IP_TW, y0 = blockIdx.y * a.th, f = blockIdx.z;
    const int y1 = min(y0 + a.th, a.out_h);
    const int tw = min(IP_TW, a.out_w - x0);
    int r0, r1, c0, c1, lo, hi;
    tap_range(y0, a.in_h, a.out_h, a.sy, a.supy, r0, hi);                    // bounds grow with the index: the tile's rows are
    tap_range(y1 - 1, a.in_h, a.out_h, a.sy, a.supy, lo, r1);                // [first row's first tap, last row's last tap)
    tap_range(x0, a.in_w, a.out_w, a.sx, a.supx, c0, hi);
    tap_range(x0 + tw - 1, a.in_w, a.out_w, a.sx, a.supx, lo, c1);
    const int nrows = r1 - r0, span = (c1 - c0) * 3;
    if (nrows > a.rows_cap || span + 8 > a.raw_stride) return;                // cannot happen for the launcher's plan
    // every global load of the tile is issued before the first wait: the column's bounds and weights, the tables, the rows
    const int x = tid % IP_TW;
    int xmin = 0, cnt = 0, kr[IP_KX];
    if (x < tw) {
        xmin = a.xb[2 * (x0 + x)] - c0;
        cnt = a.xb[2 * (x0 + x) + 1];
        if (KXR) {
#pragma unroll
            for (int t = 0; t < IP_KX; ++t) kr[t] = t < a.xksize ? a.xk[(size_t)(x0 + x) * a.xksize + t] : 0;   // 0 beyond cnt
        }
    }
    if (OUT != 2)
        for (int k = tid; k < 768; k += IP_THREADS) lut[k] = a.lut[k];
    const int yrow = 2 + a.yksize;
    for (int k = tid; k < (y1 - y0) * yrow; k += IP_THREADS) {
        const int yy = k / yrow, j = k - yy * yrow;
        ytab[k] = j < 2 ? a.yb[2 * (y0 + yy) + j] - (j == 0 ? r0 : 0) : a.yk[(size_t)(y0 + yy) * a.yksize + (j - 2)];
    }
    if (!KXR)
        for (int k = tid; k < tw * a.xksize; k += IP_THREADS) xtab[k] = a.xk[(size_t)x0 * a.xksize + k];
    // 1. the tile's source bytes, once: aligned 4-byte words, sixteen rows' loads in flight before the first LDS store.  The
    // word that would cross the end of the buffer is read 1..3 bytes earlier and shifted (no branch, no byte past the end).
    // 1. the tile's source bytes, once, as aligned 4-byte words.  A wave takes every fourth row and five 64-word pieces of it at
    // a time, so twenty loads are in flight per lane before the first LDS store.  Addresses are 32-bit offsets from `ab`, the
    // frame's start rounded down to a word; the word that would cross the end of the buffer is read 1..3 bytes earlier and
    // shifted (no byte past the end is touched), words wholly past it read the last word (their bytes are never used).
    const size_t frame_bytes = (size_t)a.in_h * a.in_w * 3;
    const unsigned fmis = (unsigned)((f * frame_bytes) & 3);                   // src is word aligned
    const unsigned char *ab = a.src + (f * frame_bytes - fmis);
    const long long room = a.src_bytes - (long long)(f * frame_bytes - fmis) - 4;   // largest offset a whole word may start at
    const int lim = room > 0x7ffffff0ll ? 0x7ffffff0 : (int)room;
    const unsigned tile_rel = fmis + (unsigned)((r0 * a.in_w + c0) * 3);       // the tile's first byte, relative to ab
    const unsigned pitch = (unsigned)a.in_w * 3;
    const int words_cap = a.raw_stride >> 2;
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int NW = IP_THREADS / 64, RQ = 4, KQ = 5;                        // rows x word pieces in flight per lane
    if (lim >= 0) {
        for (int rb = 0; rb < nrows; rb += NW * RQ)
            for (int kb = 0; kb < words_cap; kb += 64 * KQ) {
                uint32_t v[RQ][KQ];
#pragma unroll
                for (int u = 0; u < RQ; ++u) {
                    const int r = min(rb + wave + NW * u, nrows - 1);
                    const unsigned al = (tile_rel + (unsigned)r * pitch) & ~3u;
#pragma unroll
                    for (int q = 0; q < KQ; ++q) {
                        const int t = (int)al + 4 * (kb + 64 * q + lane);
                        const int o = min(t, lim);
                        const int over = min(t - o, 3);
                        uint32_t w;
                        __builtin_memcpy(&w, ab + (unsigned)o, 4);
                        v[u][q] = w >> (8 * over);
                    }
                }
#pragma unroll
                for (int u = 0; u < RQ; ++u) {
                    const int r = rb + wave + NW * u;
#pragma unroll
                    for (int q = 0; q < KQ; ++q) {
                        const int k = kb + 64 * q + lane;
                        if (r < nrows && k < words_cap) reinterpret_cast<uint32_t *>(raw + (size_t)r * a.raw_stride)[k] = v[u][q];
                    }
                }
            }
    } else {                                                                   // fewer than 4 bytes left: the last 1 x 1 frame
        if (tid < 4 && (long long)(f * frame_bytes - fmis) + tid < a.src_bytes) raw[tid] = ab[tid];
    }
    __syncthreads();
    // 2. horizontal pass -> bytes; a thread keeps its column (IP_THREADS is a multiple of IP_TW) and its weights.  LDS issue is
    // what bounds this kernel, so the 3 x 8 source bytes of a pixel are fetched as seven aligned words and realigned in
    // registers (v_alignbyte) instead of 24 byte reads with 3-way bank conflicts.
    if (x < tw) {
        const int *kx = xtab + x * a.xksize;
        for (int r = tid / IP_TW; r < nrows; r += IP_THREADS / IP_TW) {
            const unsigned at = (unsigned)r * a.raw_stride + ((tile_rel + (unsigned)r * pitch) & 3u) + xmin * 3;
            int a0 = 1 << (IP_BITS - 1), a1 = a0, a2 = a0;
            if (KXR) {
                const uint32_t *pw = reinterpret_cast<const uint32_t *>(raw + (at & ~3u));
                uint32_t w[7], d[6];
#pragma unroll
                for (int i = 0; i < 7; ++i) w[i] = pw[i];      // taps beyond cnt carry weight 0; their bytes are inside the LDS block
#pragma unroll
                for (int i = 0; i < 6; ++i) d[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], at & 3u);
#pragma unroll
                for (int t = 0; t < IP_KX; ++t) {
                    a0 = tap(a0, (d[(3 * t) >> 2] >> (8 * ((3 * t) & 3))) & 0xffu, kr[t]);
                    a1 = tap(a1, (d[(3 * t + 1) >> 2] >> (8 * ((3 * t + 1) & 3))) & 0xffu, kr[t]);
                    a2 = tap(a2, (d[(3 * t + 2) >> 2] >> (8 * ((3 * t + 2) & 3))) & 0xffu, kr[t]);
                }
            } else {
                const unsigned char *p = raw + at;
                for (int t = 0; t < cnt; ++t) {
                    const int k = kx[t];
                    a0 = tap(a0, p[3 * t], k);
                    a1 = tap(a1, p[3 * t + 1], k);
                    a2 = tap(a2, p[3 * t + 2], k);
                }
            }
            unsigned char *m = mid + (size_t)r * 3 * IP_TW + x;
            m[0] = (unsigned char)clip8(a0);
            m[IP_TW] = (unsigned char)clip8(a1);
            m[2 * IP_TW] = (unsigned char)clip8(a2);
        }
    }
    __syncthreads();
    // 3. vertical pass, table, store: a thread owns four neighbouring columns of one output row (one LDS word per tap and
    // channel, 16-byte stores)
    constexpr int QW = IP_TW / 4;
    for (int item = tid; item < (y1 - y0) * QW; item += IP_THREADS) {
        const int yy = item / QW, xq = item - yy * QW, xa = 4 * xq;
        if (xa >= tw) continue;
        const int *yt = ytab + yy * yrow;
        const int ymin = yt[0], ycnt = yt[1], y = y0 + yy;
        const uint32_t *m32 = reinterpret_cast<const uint32_t *>(mid + (size_t)ymin * 3 * IP_TW) + xq;
        int acc[3][4];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[c][i] = 1 << (IP_BITS - 1);
        for (int t = 0; t < ycnt; ++t) {
            const int k = yt[2 + t];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const uint32_t word = m32[(3 * t + c) * QW];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[c][i] = tap(acc[c][i], (word >> (8 * i)) & 0xffu, k);
            }
        }
        const int live = min(4, tw - xa);
        if (OUT == 2) {
            unsigned char *o = reinterpret_cast<unsigned char *>(a.dst) + (((size_t)f * a.out_h + y) * a.out_w + x0 + xa) * 3;
            for (int i = 0; i < live; ++i)
#pragma unroll
                for (int c = 0; c < 3; ++c) o[3 * i + c] = (unsigned char)clip8(acc[c][i]);
        } else {
            const size_t at = (size_t)f * a.dst_fs + (size_t)y * a.out_w + x0 + xa;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = lut[c * 256 + clip8(acc[c][i])];
                const size_t e = at + (size_t)c * a.dst_cs;
                if (OUT == 0) {
                    float *dp = reinterpret_cast<float *>(a.dst) + e;
                    if (a.vec4 && live == 4) *reinterpret_cast<float4 *>(dp) = make_float4(o[0], o[1], o[2], o[3]);
                    else for (int i = 0; i < live; ++i) dp[i] = o[i];
                } else {
                    bf16_t *dp = reinterpret_cast<bf16_t *>(a.dst) + e;
                    if (a.vec4 && live == 4) Payload<bf16_t>::st4(dp, make_float4(o[0], o[1], o[2], o[3]));
                    else for (int i = 0; i < live; ++i) Payload<bf16_t>::st(dp + i, o[i]);
                }
            }
        }
    }
}

// Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter (support 1), in the same double arithmetic.
static void resample_coeffs_host(int in_size, int out_size, int ksize, int *bounds, int *kk) {
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale, ss = 1.0 / filterscale;
    double *w = new double[ksize];
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            double v = (x + xmin - center + 0.5) * ss;
            if (v < 0.0) v = -v;
            w[x] = v < 1.0 ? 1.0 - v : 0.0;
            ww += w[x];
        }
        for (int x = 0; x < xmax; ++x)
            if (ww != 0.0) w[x] /= ww;
        for (int x = xmax; x < ksize; ++x) w[x] = 0.0;
        for (int x = 0; x < ksize; ++x)
            kk[(size_t)xx * ksize + x] = w[x] < 0 ? (int)(-0.5 + w[x] * (1 << IP_BITS)) : (int)(0.5 + w[x] * (1 << IP_BITS));
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    delete[] w;
}

static int resample_ksize(int in_size, int out_size) {
    if (in_size == out_size) return 1;
    const double scale = (double)in_size / (double)out_size;
    return (int)ceil(scale < 1.0 ? 1.0 : scale) * 2 + 1;
}

// ---- velodyne merge + range crop ------------------------------------------------------------------------------------
constexpr int VM_THREADS = 1024;

struct MergeArgs {
    const float *upper, *lower;
    int nu, nl, C;
    float tf[2][12];                         // row-major [R | t] of the upper and the lower sensor
    float x0, y0, x1, y1;
};

__device__ __forceinline__ bool vm_point(const MergeArgs &a, int i, float &x, float &y, float &z, const float *&src) {
    const int s = i >= a.nu;
    src = (s ? a.lower + (size_t)(i - a.nu) * a.C : a.upper + (size_t)i * a.C);
    const float px = src[0], py = src[1], pz = src[2];
    const float *t = a.tf[s];
    x = ((t[0] * px + t[1] * py) + t[2] * pz) + t[3];
    y = ((t[4] * px + t[5] * py) + t[6] * pz) + t[7];
    z = ((t[8] * px + t[9] * py) + t[10] * pz) + t[11];
    return x >= a.x0 && x <= a.x1 && y >= a.y0 && y <= a.y1;
}

__global__ __launch_bounds__(VM_THREADS) void velodyne_count_kernel(MergeArgs a, int *__restrict__ block_cnt) {
    __shared__ int wsum[VM_THREADS / kWave];
    const int i = blockIdx.x * VM_THREADS + threadIdx.x;
    float x, y, z;
    const float *src;
    const bool keep = i < a.nu + a.nl && vm_point(a, i, x, y, z, src);
    const unsigned long long m = __ballot(keep);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int w = 0; w < VM_THREADS / kWave; ++w) s += wsum[w];
        block_cnt[blockIdx.x] = s;
    }
}

// one workgroup: exclusive scan of the block counts in place; total -> block_cnt[nblocks] and *count
__global__ __launch_bounds__(VM_THREADS) void velodyne_scan_kernel(int nblocks, int *__restrict__ block_cnt, int *__restrict__ count) {
    __shared__ int part[VM_THREADS];
    const int per = (nblocks + VM_THREADS - 1) / VM_THREADS;
    const int b0 = threadIdx.x * per, b1 = min(b0 + per, nblocks);
    int s = 0;
    for (int b = b0; b < b1; ++b) s += block_cnt[b];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < VM_THREADS; d <<= 1) {
        const int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - s;
    for (int b = b0; b < b1; ++b) {
        const int c = block_cnt[b];
        block_cnt[b] = run;
        run += c;
    }
    if (threadIdx.x == VM_THREADS - 1) {
        block_cnt[nblocks] = part[threadIdx.x];
        *count = part[threadIdx.x];
    }
}

__global__ __launch_bounds__(VM_THREADS) void velodyne_write_kernel(MergeArgs a, const int *__restrict__ block_start, float *__restrict__ out) {
    __shared__ int wsum[VM_THREADS / kWave];
    const int i = blockIdx.x * VM_THREADS + threadIdx.x;
    float x, y, z;
    const float *src = nullptr;
    const bool keep = i < a.nu + a.nl && vm_point(a, i, x, y, z, src);
    const unsigned long long m = __ballot(keep);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) wsum[w] = __popcll(m);
    __syncthreads();
    if (!keep) return;
    int at = block_start[blockIdx.x] + __popcll(m & ((1ull << lane) - 1ull));
    for (int k = 0; k < w; ++k) at += wsum[k];
    float *o = out + (size_t)at * a.C;
    o[0] = x; o[1] = y; o[2] = z;
    for (int c = 3; c < a.C; ++c) o[c] = src[c];
}

}  // namespace mgar

using namespace mgar;

#define MGAR_API __attribute__((visibility("default")))

extern "C" {

MGAR_API int mgar_image_resample_ksize(int in_size, int out_size) {
    if (in_size <= 0 || out_size <= 0) return MGAR_EINVAL;
    return resample_ksize(in_size, out_size);
}

MGAR_API int mgar_image_resample_coeffs(int in_size, int out_size, int *bounds, int *kk) {
    MGAR_REQUIRE(in_size > 0 && out_size > 0 && bounds && kk, "mgar_image_resample_coeffs: bad arguments");
    if (in_size == out_size) {               // Pillow skips the pass; one unit tap reproduces the byte exactly
        for (int x = 0; x < out_size; ++x) {
            bounds[2 * x] = x;
            bounds[2 * x + 1] = 1;
            kk[x] = 1 << IP_BITS;
        }
        return MGAR_OK;
    }
    resample_coeffs_host(in_size, out_size, resample_ksize(in_size, out_size), bounds, kk);
    return MGAR_OK;
}

MGAR_API int mgar_image_resize_normalize_u8(int frames, int in_h, int in_w, int out_h, int out_w, const unsigned char *src,
                                            const int *xbounds, const int *xkk, const int *ybounds, const int *ykk,
                                            const float *lut, void *dst, long long dst_frame_stride,
                                            long long dst_channel_stride, int dst_kind, void *stream) {
    MGAR_REQUIRE(frames >= 0 && in_h > 0 && in_w > 0 && out_h > 0 && out_w > 0, "mgar_image_resize_normalize_u8: bad sizes");
    MGAR_REQUIRE(dst_kind >= 0 && dst_kind <= 2, "mgar_image_resize_normalize_u8: dst_kind is 0 (f32), 1 (bf16) or 2 (u8)");
    if (frames == 0) return MGAR_OK;
    MGAR_REQUIRE(src && xbounds && xkk && ybounds && ykk && dst && (lut || dst_kind == 2),
                 "mgar_image_resize_normalize_u8: null pointer");
    MGAR_REQUIRE((reinterpret_cast<uintptr_t>(src) & 3) == 0, "mgar_image_resize_normalize_u8: src must be 4-byte aligned");
    MGAR_REQUIRE((long long)in_h * in_w * 3 < (1ll << 31) && (long long)out_h * out_w < (1ll << 31) && frames <= 65535,
                 "mgar_image_resize_normalize_u8: frame too large");
    ResizeArgs a;
    a.src = src; a.xb = xbounds; a.xk = xkk; a.yb = ybounds; a.yk = ykk; a.lut = lut; a.dst = dst;
    a.dst_fs = dst_frame_stride; a.dst_cs = dst_channel_stride;
    a.src_bytes = (long long)frames * in_h * in_w * 3;
    a.in_h = in_h; a.in_w = in_w; a.out_h = out_h; a.out_w = out_w;
    a.xksize = resample_ksize(in_w, out_w);
    a.yksize = resample_ksize(in_h, out_h);
    const bool kxr = a.xksize <= IP_KX;
    // LDS plan from the filter geometry alone (the tables live on the device): a tile of IP_TW columns touches at most
    // ceil(IP_TW * scale) + ksize source columns, th rows touch at most ceil(th * scale) + ksize source rows
    const double sx = (double)in_w / out_w, sy = (double)in_h / out_h;
    const int span_cap = (int)fmin((double)in_w, ceil(IP_TW * sx) + a.xksize + 1) * 3;
    a.raw_stride = (span_cap + 8 + 3) & ~3;                        // + the alignment slack of the first and the last word
    size_t lds = 0;
    int th = 16;
    for (;; th >>= 1) {
        a.rows_cap = (int)fmin((double)in_h, ceil(th * sy) + a.yksize + 1);
        lds = 768 * sizeof(float) + sizeof(int) * ((size_t)th * (2 + a.yksize) + (kxr ? 0 : (size_t)IP_TW * a.xksize))
              + (size_t)a.rows_cap * (a.raw_stride + 3 * IP_TW);
        if (lds <= (size_t)IP_LDS_BUDGET) break;
        if (th == 1) {
            set_error("mgar_image_resize_normalize_u8: down-scaling factor too large for one LDS tile");
            return MGAR_EUNSUPPORTED;
        }
    }
    a.th = th;
    a.vec4 = dst_kind != 2 && out_w % 4 == 0 && dst_frame_stride % 4 == 0 && dst_channel_stride % 4 == 0
             && reinterpret_cast<uintptr_t>(dst) % 16 == 0;
    a.sx = sx; a.supx = sx < 1.0 ? 1.0 : sx;                       // precompute_coeffs: scale, support = 1.0 * max(scale, 1)
    a.sy = sy; a.supy = sy < 1.0 ? 1.0 : sy;
    dim3 grid(ceil_div(out_w, IP_TW), ceil_div(out_h, th), frames);
    MGAR_REQUIRE(grid.y <= 65535, "mgar_image_resize_normalize_u8: too many row tiles");
    hipStream_t st = (hipStream_t)stream;
    const double out_bytes = dst_kind == 0 ? 4.0 : (dst_kind == 1 ? 2.0 : 1.0);
    KtScope kt(KT_IMAGE_PREP, st, (double)frames * ((double)in_h * in_w * 3 + (double)out_h * out_w * 3 * out_bytes));
#define IP_LAUNCH(O) \
    do { if (kxr) image_resize_normalize_kernel<O, true><<<grid, IP_THREADS, lds, st>>>(a); \
         else image_resize_normalize_kernel<O, false><<<grid, IP_THREADS, lds, st>>>(a); } while (0)
    if (dst_kind == 0) IP_LAUNCH(0);
    else if (dst_kind == 1) IP_LAUNCH(1);
    else IP_LAUNCH(2);
#undef IP_LAUNCH
    return check_launch("mgar_image_resize_normalize_u8: launch failed");
}

MGAR_API long long mgar_velodyne_merge_crop_workspace_ints(int n_upper, int n_lower) {
    if (n_upper < 0 || n_lower < 0) return MGAR_EINVAL;
    return (long long)ceil_div((long long)n_upper + n_lower, VM_THREADS) + 2;
}

MGAR_API int mgar_velodyne_merge_crop(int n_upper, int n_lower, int C, const float *upper, const float *lower,
                                      const float *tf_upper, const float *tf_lower, const float *xy_range, int *workspace,
                                      float *out, int *count, void *stream) {
    MGAR_REQUIRE(n_upper >= 0 && n_lower >= 0 && C >= 3, "mgar_velodyne_merge_crop: bad sizes");
    MGAR_REQUIRE(tf_upper && tf_lower && xy_range && workspace && count, "mgar_velodyne_merge_crop: null pointer");
    MGAR_REQUIRE((long long)n_upper + n_lower < (1ll << 30), "mgar_velodyne_merge_crop: too many points");
    hipStream_t st = (hipStream_t)stream;
    const int n = n_upper + n_lower;
    if (n == 0) {
        hipError_t e = hipMemsetAsync(count, 0, sizeof(int), st);
        if (e != hipSuccess) { set_error("mgar_velodyne_merge_crop: memset failed"); return MGAR_ELAUNCH; }
        return MGAR_OK;
    }
    MGAR_REQUIRE((upper || n_upper == 0) && (lower || n_lower == 0) && out, "mgar_velodyne_merge_crop: null pointer");
    MergeArgs a;
    a.upper = upper; a.lower = lower; a.nu = n_upper; a.nl = n_lower; a.C = C;
    for (int k = 0; k < 12; ++k) { a.tf[0][k] = tf_upper[k]; a.tf[1][k] = tf_lower[k]; }   // HOST arrays (12 floats each)
    a.x0 = xy_range[0]; a.y0 = xy_range[1]; a.x1 = xy_range[2]; a.y1 = xy_range[3];       // HOST array [x0, y0, x1, y1]
    const int nblocks = ceil_div(n, VM_THREADS);
    velodyne_count_kernel<<<nblocks, VM_THREADS, 0, st>>>(a, workspace);
    velodyne_scan_kernel<<<1, VM_THREADS, 0, st>>>(nblocks, workspace, count);
    velodyne_write_kernel<<<nblocks, VM_THREADS, 0, st>>>(a, workspace, out);
    return check_launch("mgar_velodyne_merge_crop: launch failed");
}

}  // extern "C"
