// interpolate.hip -- three nearest neighbours + inverse-distance interpolation
// (fwd + bwd), batch and stack layouts, for gfx950.
//
// Replaces  pointnet2_batch/src/interpolate_gpu.cu:16-168
//           pointnet2_stack/src/interpolate_gpu.cu:16-194
//
// three_nn: one lane per unknown point; the known cloud is a wave-uniform stream through
// the scalar cache (same structure as ball_query.hip).  The reference's strict-'<' cascade
// is kept verbatim so equal distances resolve to the earlier index; it sits behind a
// wave-uniform branch ("does any lane improve its 3rd best?"), so the steady state is the
// 7-VALU distance evaluation + 1 compare per pair.
//
// three_interpolate: bandwidth-bound gathers.  Batch layout (B,C,M): a thread owns one
// output point and walks the channels (idx/weight read once, not once per channel).
// Stack layout (M,C): lanes run along c so every access is a contiguous feature row;
// the backward's float atomics therefore cover whole row segments.
#include "common.hpp"
#include "payload.hpp"

namespace mgar {

constexpr int TN_THREADS = 256;
constexpr int TN_CHUNK = 16;
typedef const float __attribute__((address_space(4))) *cfloat_p;

template <bool STACK>
__global__ __launch_bounds__(TN_THREADS) void three_nn_kernel(int B, int n_batch, int m_batch,
                                                              const float *__restrict__ unknown,
                                                              const int *__restrict__ unknown_batch_cnt,
                                                              const float *__restrict__ known,
                                                              const int *__restrict__ known_batch_cnt,
                                                              float *__restrict__ dist2, int *__restrict__ idx) {
    int u0, u_end, k_start, m;
    if (STACK) {
        int g = blockIdx.x, us = 0, ks = 0, bs = 0;
        bool found = false;
        for (; bs < B; ++bs) {
            const int ni = unknown_batch_cnt[bs];
            const int nb = (ni + TN_THREADS - 1) / TN_THREADS;
            if (g < nb) { found = true; break; }
            g -= nb;
            us += ni;
            ks += known_batch_cnt[bs];
        }
        if (!found) return;
        u0 = us + g * TN_THREADS;
        u_end = us + unknown_batch_cnt[bs];
        k_start = ks;
        m = known_batch_cnt[bs];
    } else {
        const int bs = blockIdx.y;
        u0 = bs * n_batch + blockIdx.x * TN_THREADS;
        u_end = (bs + 1) * n_batch;
        k_start = bs * m_batch;
        m = m_batch;
    }
    const int u = u0 + threadIdx.x;
    const bool valid = u < u_end;
    // lanes without a point carry NaN: every "d < best" is false, so they never trigger
    // the wave-uniform update branch
    float ux = __builtin_nanf(""), uy = 0.f, uz = 0.f;
    if (valid) {
        ux = unknown[(size_t)u * 3 + 0];
        uy = unknown[(size_t)u * 3 + 1];
        uz = unknown[(size_t)u * 3 + 2];
    }
    // reference: double best = 1e40 stored to float -> +inf for unfilled slots
    float best1 = __builtin_inff(), best2 = __builtin_inff(), best3 = __builtin_inff();
    int besti1 = 0, besti2 = 0, besti3 = 0;
    cfloat_p K = (cfloat_p)(known + (size_t)k_start * 3);

    auto consider = [&](float d, int k) {
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
    };

    int k0 = 0;
    for (; k0 + TN_CHUNK <= m; k0 += TN_CHUNK) {
        float c[TN_CHUNK * 3];
#pragma unroll
        for (int i = 0; i < TN_CHUNK * 3; ++i) c[i] = K[k0 * 3 + i];
#pragma unroll
        for (int j = 0; j < TN_CHUNK; ++j) {
            const float d = d2_of(ux - c[j * 3 + 0], uy - c[j * 3 + 1], uz - c[j * 3 + 2]);
            if (d < best3) {  // d < best1 or d < best2 imply d < best3 (best1 <= best2 <= best3)
                asm volatile("; top-3 update" ::: "memory");  // keep the execz skip branch
                consider(d, k0 + j);
            }
        }
    }
    for (int k = k0; k < m; ++k) {
        const float d = d2_of(ux - K[k * 3 + 0], uy - K[k * 3 + 1], uz - K[k * 3 + 2]);
        consider(d, k);
    }
    if (valid) {
        const int off = STACK ? k_start : 0;  // stack op returns global rows (interpolate_gpu.cu:72-74)
        dist2[(size_t)u * 3 + 0] = best1; dist2[(size_t)u * 3 + 1] = best2; dist2[(size_t)u * 3 + 2] = best3;
        idx[(size_t)u * 3 + 0] = besti1 + off; idx[(size_t)u * 3 + 1] = besti2 + off; idx[(size_t)u * 3 + 2] = besti3 + off;
    }
}

// ------------------------- three_interpolate, batch layout -------------------------
constexpr int TI_CCHUNK = 8;
constexpr int TI_LDS_MAX_FLOATS = 36864;   // 144 KB of the 160 KB LDS

// Forward, LDS-staged: a workgroup stages CH channel rows of the KNOWN features (CH * m floats) in
// LDS with coalesced loads and then gathers from LDS.  The direct version below is bound by the
// texture-address path (3 random 4-byte gathers per output: 0.65 TB/s measured at c = 256,
// n = 16384); LDS serves the same random reads ~10x faster.  grid (ceil(c/CH), b)
// T = payload type of points / out (float or bf16_t); the LDS image, the weights and the arithmetic are fp32.
template <typename T>
__global__ __launch_bounds__(1024) void three_interp_batch_fwd_lds_kernel(int c, int m, int n, int CH,
                                                                          const T *__restrict__ points,
                                                                          const int *__restrict__ idx,
                                                                          const float *__restrict__ weight,
                                                                          T *__restrict__ out, size_t out_bs, int accumulate) {
    extern __shared__ float rows[];  // [CH][m]
    const int c0 = blockIdx.x * CH, bs = blockIdx.y;
    const int nch = min(CH, c - c0);
    const T *src = points + ((size_t)bs * c + c0) * m;
    for (int i = threadIdx.x; i < nch * m; i += blockDim.x) rows[i] = Payload<T>::ld(src + i);
    __syncthreads();
    T *dst = out + (size_t)bs * out_bs + (size_t)c0 * n;
    for (int pt = threadIdx.x; pt < n; pt += blockDim.x) {
        const size_t o = ((size_t)bs * n + pt) * 3;
        const int i0 = idx[o], i1 = idx[o + 1], i2 = idx[o + 2];
        const float w0 = weight[o], w1 = weight[o + 1], w2 = weight[o + 2];
        for (int ch = 0; ch < nch; ++ch) {
            const float *r = rows + (size_t)ch * m;
            float v = dot3_of(w0, r[i0], w1, r[i1], w2, r[i2]);
            if (accumulate) v += Payload<T>::ld(dst + (size_t)ch * n + pt);   // out = out + interpolation (see mgar_three_interpolate_batch_add)
            Payload<T>::st(dst + (size_t)ch * n + pt, v);
        }
    }
}

// Backward through an inverted index: `list` holds, per cloud, its 3n (known point j, unknown point
// u, weight) entries sorted by j (stable, so ascending (u, k) inside one j), packed as
// int2{(j << 16) | u, weight bits}.  A workgroup stages CH rows of grad_out (CH * n floats) in LDS;
// its waves walk contiguous runs of the sorted entries 64 at a time (coalesced), gather from LDS and
// reduce runs of equal j with a segmented wave scan, so every known point is written exactly once --
// no atomics (LDS float atomics run at ~0.3 lanes/clk/CU here: 8.2 ms for the level-1 FP module of
// config c3) and no per-point list walk (list lengths are very skewed: FPS puts few known points into
// dense clusters, so some lists hold hundreds of entries).  Segments that cross a wave boundary
// go through 2 records per wave in LDS and are merged in wave order by one thread per channel, so the
// summation order -- and with it the result -- is fixed.  Known points nobody references are not
// written: the caller hands in a zeroed grad_points.  grid (ceil(c/CH), b)
// one step of the segmented scan: fetch (key, v) from the DPP source lane, add where the keys agree
template <int CH, int CTRL, int ROW_MASK>
__device__ __forceinline__ void seg_scan_step(int key, float (&v)[CH], bool has_src) {
    const int ku = __builtin_amdgcn_update_dpp(-1, key, CTRL, ROW_MASK, 0xF, false);
    const bool take = has_src && ku == key;
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
        const float vu = dpp_get<CTRL, ROW_MASK>(0.f, v[ch]);
        if (take) v[ch] = vu + v[ch];
    }
}

template <int CH>
__global__ __launch_bounds__(1024) void three_interp_batch_bwd_sorted_kernel(int c, int n, int m,
                                                                             const float *__restrict__ grad_out, size_t go_bs,
                                                                             const int2 *__restrict__ list,
                                                                             float *__restrict__ grad_points) {
    extern __shared__ float rows[];  // [CH][n]
    __shared__ int rec_key[32];
    __shared__ float rec_sum[32][CH];
    const int c0 = blockIdx.x * CH, bs = blockIdx.y;
    const int nch = min(CH, c - c0);
    const float *src = grad_out + (size_t)bs * go_bs + (size_t)c0 * n;
    if ((n & 3) == 0) {
        for (int i = threadIdx.x * 4; i < nch * n; i += blockDim.x * 4)
            *reinterpret_cast<float4 *>(rows + i) = *reinterpret_cast<const float4 *>(src + i);
    } else {
        for (int i = threadIdx.x; i < nch * n; i += blockDim.x) rows[i] = src[i];
    }
    if (threadIdx.x < 32) rec_key[threadIdx.x] = -1;
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int E = 3 * n, chunks = (E + 63) >> 6, cpw = (chunks + nwaves - 1) / nwaves;
    const int2 *L = list + (size_t)bs * E;
    float *dst = grad_points + ((size_t)bs * c + c0) * m;

    // one lane delivers a finished (key, sums): the wave's first delivery goes to its head record
    auto deliver = [&](int key, const float (&sum)[CH], bool to_head) {
        if (key >= m) return;  // padding lanes behind the last entry
        if (to_head) {
            rec_key[2 * wave] = key;
#pragma unroll
            for (int ch = 0; ch < CH; ++ch) rec_sum[2 * wave][ch] = sum[ch];
        } else {
#pragma unroll
            for (int ch = 0; ch < CH; ++ch)
                if (ch < nch) dst[(size_t)ch * m + key] = sum[ch];
        }
    };

    int carry_key = -1;  // wave-uniform: the segment still open at the end of the previous chunk
    float carry[CH];
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) carry[ch] = 0.f;
    bool first_pending = true;
    const int q1 = min((wave + 1) * cpw, chunks);
    constexpr int PF = 4;  // chunks fetched per round trip: the walk is latency-bound otherwise
    for (int qb = wave * cpw; qb < q1; qb += PF) {
        int2 pf[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int e = (qb + k) * 64 + lane;
            pf[k] = (qb + k < q1 && e < E) ? L[e] : make_int2(m << 16, 0);
        }
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            if (qb + k >= q1) break;
            const int2 en = pf[k];
            const int key = (int)((unsigned)en.x >> 16);
            const int u = en.x & 0xffff;
            const float w = __int_as_float(en.y);
            float v[CH];
#pragma unroll
            for (int ch = 0; ch < CH; ++ch) v[ch] = (ch < nch && key < m) ? rows[(size_t)ch * n + u] * w : 0.f;
            // segmented inclusive scan on DPP (keys are sorted: an equal key d lanes back => same segment):
            // row_shr 1/2/4/8 inside rows of 16, then lane 15 / lane 31 broadcasts across rows
            seg_scan_step<CH, 0x111, 0xF>(key, v, (lane & 15) >= 1);
            seg_scan_step<CH, 0x112, 0xF>(key, v, (lane & 15) >= 2);
            seg_scan_step<CH, 0x114, 0xF>(key, v, (lane & 15) >= 4);
            seg_scan_step<CH, 0x118, 0xF>(key, v, (lane & 15) >= 8);
            seg_scan_step<CH, 0x142, 0xA>(key, v, (lane & 16) != 0);
            seg_scan_step<CH, 0x143, 0xC>(key, v, lane >= 32);
            const int key_next = __shfl_down(key, 1);
            const bool is_tail = lane < 63 && key_next != key;
            const int key0 = __builtin_amdgcn_readfirstlane(key);
            if (carry_key >= 0 && key0 != carry_key) {  // the open segment ended exactly at the chunk border
                if (lane == 0) deliver(carry_key, carry, first_pending);
                first_pending = false;
                carry_key = -1;
            }
            if (carry_key >= 0 && key == carry_key && (is_tail || lane == 63)) {
#pragma unroll
                for (int ch = 0; ch < CH; ++ch) v[ch] = carry[ch] + v[ch];
            }
            const unsigned long long tails = __ballot(is_tail);
            if (tails) {
                const int head_lane = first_pending ? (int)__builtin_ctzll(tails) : -1;
                if (is_tail) deliver(key, v, lane == head_lane);
                first_pending = false;
            }
            carry_key = __builtin_amdgcn_readlane(key, 63);
#pragma unroll
            for (int ch = 0; ch < CH; ++ch) carry[ch] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[ch]), 63));
        }
    }
    if (carry_key >= 0 && carry_key < m && lane == 0) {
        rec_key[2 * wave + 1] = carry_key;
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) rec_sum[2 * wave + 1][ch] = carry[ch];
    }
    __syncthreads();
    if (threadIdx.x < nch) {  // merge the wave-border records in wave order
        const int ch = threadIdx.x;
        int cur = -1;
        float sum = 0.f;
        for (int r = 0; r < 2 * nwaves; ++r) {
            const int k = rec_key[r];
            if (k < 0) continue;
            if (k == cur) {
                sum = sum + rec_sum[r][ch];
            } else {
                if (cur >= 0) dst[(size_t)ch * m + cur] = sum;
                cur = k;
                sum = rec_sum[r][ch];
            }
        }
        if (cur >= 0) dst[(size_t)ch * m + cur] = sum;
    }
}

// direct forward (rows that do not fit LDS).  grid (ceil(n/256), ceil(c/TI_CCHUNK), b)
template <typename T>
__global__ __launch_bounds__(256) void three_interp_batch_fwd_kernel(int c, int m, int n,
                                                                     const T *__restrict__ points,
                                                                     const int *__restrict__ idx,
                                                                     const float *__restrict__ weight,
                                                                     T *__restrict__ out, size_t out_bs, int accumulate) {
    const int pt = blockIdx.x * 256 + threadIdx.x;
    if (pt >= n) return;
    const int bs = blockIdx.z;
    const int c0 = blockIdx.y * TI_CCHUNK, c1 = min(c0 + TI_CCHUNK, c);
    const size_t o = ((size_t)bs * n + pt) * 3;
    const int i0 = idx[o], i1 = idx[o + 1], i2 = idx[o + 2];
    const float w0 = weight[o], w1 = weight[o + 1], w2 = weight[o + 2];
    const T *src = points + ((size_t)bs * c + c0) * m;
    T *dst = out + (size_t)bs * out_bs + (size_t)c0 * n + pt;
    for (int ci = c0; ci < c1; ++ci) {
        float v = dot3_of(w0, Payload<T>::ld(src + i0), w1, Payload<T>::ld(src + i1), w2, Payload<T>::ld(src + i2));
        if (accumulate) v += Payload<T>::ld(dst);
        Payload<T>::st(dst, v);
        src += m;
        dst += n;
    }
}

// One workgroup owns CH consecutive (b, c) rows of grad_points (CH * m floats in LDS): idx and weight
// (24 B per point) are read once per CH channels instead of once per channel -- at FP level 1
// (n = 16384, c = 256, 120 clouds) the per-channel version re-read 12 GB of idx/weight from L2.
// grid (ceil(c / CH), b)
__global__ __launch_bounds__(1024) void three_interp_batch_bwd_lds_kernel(int c, int n, int m, int CH,
                                                                          const float *__restrict__ grad_out, size_t go_bs,
                                                                          const int *__restrict__ idx,
                                                                          const float *__restrict__ weight,
                                                                          float *__restrict__ grad_points) {
    extern __shared__ float rows[];  // [CH][m]
    const int c0 = blockIdx.x * CH, bs = blockIdx.y;
    const int nch = min(CH, c - c0);
    for (int i = threadIdx.x; i < nch * m; i += blockDim.x) rows[i] = 0.f;
    __syncthreads();
    const float *g = grad_out + (size_t)bs * go_bs + (size_t)c0 * n;
    for (int pt = threadIdx.x; pt < n; pt += blockDim.x) {
        const size_t o = ((size_t)bs * n + pt) * 3;
        const int i0 = idx[o], i1 = idx[o + 1], i2 = idx[o + 2];
        const float w0 = weight[o], w1 = weight[o + 1], w2 = weight[o + 2];
        for (int ch = 0; ch < nch; ++ch) {
            const float gv = g[(size_t)ch * n + pt];
            float *r = rows + (size_t)ch * m;
            atomicAdd(&r[i0], gv * w0);
            atomicAdd(&r[i1], gv * w1);
            atomicAdd(&r[i2], gv * w2);
        }
    }
    __syncthreads();
    float *dst = grad_points + ((size_t)bs * c + c0) * m;
    for (int i = threadIdx.x; i < nch * m; i += blockDim.x) {
        const float v = rows[i];
        if (v != 0.f) dst[i] += v;
    }
}

__global__ __launch_bounds__(256) void three_interp_batch_bwd_atomic_kernel(int c, int n, int m,
                                                                            const float *__restrict__ grad_out, size_t go_bs,
                                                                            const int *__restrict__ idx,
                                                                            const float *__restrict__ weight,
                                                                            float *__restrict__ grad_points) {
    const int pt = blockIdx.x * 256 + threadIdx.x;
    if (pt >= n) return;
    const int ci = blockIdx.y, bs = blockIdx.z;
    const size_t o = ((size_t)bs * n + pt) * 3;
    const float gv = grad_out[(size_t)bs * go_bs + (size_t)ci * n + pt];
    float *G = grad_points + ((size_t)bs * c + ci) * m;
    atomicAdd(G + idx[o + 0], gv * weight[o + 0]);
    atomicAdd(G + idx[o + 1], gv * weight[o + 1]);
    atomicAdd(G + idx[o + 2], gv * weight[o + 2]);
}

// ------------------------- three_interpolate, stack layout -------------------------
// flat index e = pt*C + ci, lanes along c
template <typename T>
__global__ __launch_bounds__(256) void three_interp_stack_fwd_kernel(long long total, int C,
                                                                     const T *__restrict__ features,
                                                                     const int *__restrict__ idx,
                                                                     const float *__restrict__ weight,
                                                                     T *__restrict__ out) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long pt = e / C;
        const int ci = (int)(e - pt * C);
        const int i0 = idx[pt * 3], i1 = idx[pt * 3 + 1], i2 = idx[pt * 3 + 2];
        const float w0 = weight[pt * 3], w1 = weight[pt * 3 + 1], w2 = weight[pt * 3 + 2];
        Payload<T>::st(out + e, dot3_of(w0, Payload<T>::ld(features + (size_t)i0 * C + ci), w1,
                                        Payload<T>::ld(features + (size_t)i1 * C + ci), w2,
                                        Payload<T>::ld(features + (size_t)i2 * C + ci)));
    }
}

__global__ __launch_bounds__(256) void three_interp_stack_bwd_kernel(long long total, int C,
                                                                     const float *__restrict__ grad_out,
                                                                     const int *__restrict__ idx,
                                                                     const float *__restrict__ weight,
                                                                     float *__restrict__ grad_features) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long pt = e / C;
        const int ci = (int)(e - pt * C);
        const float gv = grad_out[e];
        atomicAdd(grad_features + (size_t)idx[pt * 3 + 0] * C + ci, gv * weight[pt * 3 + 0]);
        atomicAdd(grad_features + (size_t)idx[pt * 3 + 1] * C + ci, gv * weight[pt * 3 + 1]);
        atomicAdd(grad_features + (size_t)idx[pt * 3 + 2] * C + ci, gv * weight[pt * 3 + 2]);
    }
}

}  // namespace mgar

using namespace mgar;

extern "C" __attribute__((visibility("default"))) int mgar_three_nn_batch(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                                   int *idx, void *stream) {
    MGAR_REQUIRE(b >= 0 && n >= 0 && m >= 0, "three_nn_batch: negative size");
    MGAR_REQUIRE(b <= 65535, "three_nn_batch: b > 65535");
    if (b == 0 || n == 0) return MGAR_OK;
    MGAR_REQUIRE(unknown && dist2 && idx && (known || m == 0), "three_nn_batch: null pointer");
    dim3 grid(ceil_div(n, TN_THREADS), b);
    KtScope kt(KT_THREE_NN, (hipStream_t)stream, (double)b * (12.0 * n + 12.0 * m + 24.0 * n), 8.0 * b * (double)n * m);
    hipLaunchKernelGGL(three_nn_kernel<false>, grid, dim3(TN_THREADS), 0, (hipStream_t)stream, b, n, m, unknown,
                       (const int *)nullptr, known, (const int *)nullptr, dist2, idx);
    return check_launch("three_nn_batch: launch failed");
}

extern "C" __attribute__((visibility("default"))) int mgar_three_nn_stack(int batch_size, int N, int M, const float *unknown, const int *unknown_batch_cnt,
                                   const float *known, const int *known_batch_cnt, float *dist2, int *idx,
                                   void *stream) {
    MGAR_REQUIRE(batch_size >= 0 && N >= 0 && M >= 0, "three_nn_stack: negative size");
    if (batch_size == 0 || N == 0) return MGAR_OK;
    MGAR_REQUIRE(unknown && dist2 && idx && unknown_batch_cnt && known_batch_cnt && (known || M == 0),
                 "three_nn_stack: null pointer");
    dim3 grid(ceil_div(N, TN_THREADS) + batch_size);
    KtScope kt(KT_THREE_NN, (hipStream_t)stream, 12.0 * N + 12.0 * M + 24.0 * N);
    hipLaunchKernelGGL(three_nn_kernel<true>, grid, dim3(TN_THREADS), 0, (hipStream_t)stream, batch_size, 0, 0, unknown,
                       unknown_batch_cnt, known, known_batch_cnt, dist2, idx);
    return check_launch("three_nn_stack: launch failed");
}

template <typename T>
static int three_interpolate_batch_impl(int b, int c, int m, int n, const T *points, const int *idx, const float *weight, T *out,
                                        void *stream, long long out_bstride = -1, int accumulate = 0) {
    MGAR_REQUIRE(b >= 0 && c >= 0 && m >= 0 && n >= 0, "three_interpolate_batch: negative size");
    if (out_bstride < 0) out_bstride = (long long)c * n;
    MGAR_REQUIRE(out_bstride >= (long long)c * n, "three_interpolate_batch: output batch stride smaller than a sample");
    const size_t out_bs = (size_t)out_bstride;
    MGAR_REQUIRE(b <= 65535, "three_interpolate_batch: b > 65535");
    if ((long long)b * c * n == 0) return MGAR_OK;
    MGAR_REQUIRE(points && idx && weight && out, "three_interpolate_batch: null pointer");
    // minimum traffic: the source rows once, the outputs once, idx + weight (24 B per unknown point)
    KtScope kt(KT_THREE_INTERP_FWD, (hipStream_t)stream, (double)b * (24.0 * n + (double)sizeof(T) * c * ((double)m + n)));
    if (m <= TI_LDS_MAX_FLOATS) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void *)three_interp_batch_fwd_lds_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      TI_LDS_MAX_FLOATS * (int)sizeof(float));
            attr_set = true;
        }
        int ch = TI_LDS_MAX_FLOATS / m;
        ch = ch > 8 ? 8 : ch;
        while (ch > 1 && (long long)b * ceil_div(c, ch) < 512) ch >>= 1;
        hipLaunchKernelGGL(three_interp_batch_fwd_lds_kernel<T>, dim3(ceil_div(c, ch), b), dim3(n >= 4096 ? 1024 : 256),
                           (size_t)ch * m * sizeof(float), (hipStream_t)stream, c, m, n, ch, points, idx, weight, out, out_bs, accumulate);
    } else {
        dim3 grid(ceil_div(n, 256), ceil_div(c, TI_CCHUNK), b);
        hipLaunchKernelGGL(three_interp_batch_fwd_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, c, m, n, points, idx,
                           weight, out, out_bs, accumulate);
    }
    return check_launch("three_interpolate_batch: launch failed");
}

extern "C" __attribute__((visibility("default"))) int mgar_three_interpolate_batch(int b, int c, int m, int n, const float *points, const int *idx,
                                            const float *weight, float *out, void *stream) {
    return three_interpolate_batch_impl<float>(b, c, m, n, points, idx, weight, out, stream);
}
// bf16 payload: points / out address bf16 elements; idx int32, weight fp32
extern "C" __attribute__((visibility("default"))) int mgar_three_interpolate_batch_bf16(int b, int c, int m, int n, const void *points, const int *idx,
                                            const float *weight, void *out, void *stream) {
    return three_interpolate_batch_impl<bf16_t>(b, c, m, n, (const bf16_t *)points, idx, weight, (bf16_t *)out, stream);
}

// out (b, c, n) += interpolation: the caller pre-fills out.  "Project, then interpolate" (round 3): the first shared-MLP layer of
// a feature-propagation module is linear and the interpolation is linear per channel, so
//     W [interp(f) ; skip] = interp(W_a f) + W_b skip
// -- the known features are projected on the COARSE level (m columns instead of n), the skip term is one GEMM with K = C_skip, and
// this kernel adds the interpolation of the projected features into it (reference pointnet2_batch/pointnet2_modules.py:139-150).
extern "C" __attribute__((visibility("default"))) int mgar_three_interpolate_batch_add(int b, int c, int m, int n, const float *points, const int *idx,
                                            const float *weight, float *out, void *stream) {
    return three_interpolate_batch_impl<float>(b, c, m, n, points, idx, weight, out, stream, -1, 1);
}
extern "C" __attribute__((visibility("default"))) int mgar_three_interpolate_batch_add_bf16(int b, int c, int m, int n, const void *points, const int *idx,
                                            const float *weight, void *out, void *stream) {
    return three_interpolate_batch_impl<bf16_t>(b, c, m, n, (const bf16_t *)points, idx, weight, (bf16_t *)out, stream, -1, 1);
}

// out a CHANNEL SLICE of a wider (b, c_total, n) tensor (samples out_bstride >= c * n elements apart): the decoder's
// torch.cat([interpolated, skip]) without the pass that copies the interpolated half
extern "C" __attribute__((visibility("default"))) int mgar_three_interpolate_batch_into(int b, int c, int m, int n, const float *points, const int *idx,
                                            const float *weight, float *out, long long out_bstride, void *stream) {
    return three_interpolate_batch_impl<float>(b, c, m, n, points, idx, weight, out, stream, out_bstride);
}
extern "C" __attribute__((visibility("default"))) int mgar_three_interpolate_batch_into_bf16(int b, int c, int m, int n, const void *points, const int *idx,
                                            const float *weight, void *out, long long out_bstride, void *stream) {
    return three_interpolate_batch_impl<bf16_t>(b, c, m, n, (const bf16_t *)points, idx, weight, (bf16_t *)out, stream, out_bstride);
}

// grad_out_bstride: elements between consecutive samples of grad_out (>= c * n): grad_out may be a channel slice of a wider
// (b, c_total, n) tensor -- the gradient of the decoder's torch.cat([interpolated, skip]) -- read in place, no copy
extern "C" __attribute__((visibility("default"))) int mgar_three_interpolate_grad_batch_strided(int b, int c, int n, int m, const float *grad_out,
                                                 long long grad_out_bstride, const int *idx,
                                                 const float *weight, float *grad_points, void *stream) {
    MGAR_REQUIRE(b >= 0 && c >= 0 && m >= 0 && n >= 0, "three_interpolate_grad_batch: negative size");
    MGAR_REQUIRE(grad_out_bstride >= (long long)c * n, "three_interpolate_grad_batch: grad_out batch stride smaller than a sample");
    const size_t go_bs = (size_t)grad_out_bstride;
    MGAR_REQUIRE(b <= 65535 && c <= 65535, "three_interpolate_grad_batch: b or c > 65535");
    if ((long long)b * c * n == 0) return MGAR_OK;
    MGAR_REQUIRE(grad_out && idx && weight && grad_points, "three_interpolate_grad_batch: null pointer");
    if (m <= TI_LDS_MAX_FLOATS) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void *)three_interp_batch_bwd_lds_kernel,
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      TI_LDS_MAX_FLOATS * (int)sizeof(float));
            attr_set = true;
        }
        const int threads = n >= 4096 ? 1024 : 256;
        int ch = TI_LDS_MAX_FLOATS / (m > 0 ? m : 1);
        ch = ch > 16 ? 16 : (ch < 1 ? 1 : ch);
        // keep enough workgroups in flight: at least ~2 per CU
        while (ch > 1 && (long long)b * ceil_div(c, ch) < 512) ch >>= 1;
        hipLaunchKernelGGL(three_interp_batch_bwd_lds_kernel, dim3(ceil_div(c, ch), b), dim3(threads),
                           (size_t)ch * m * sizeof(float), (hipStream_t)stream, c, n, m, ch, grad_out, go_bs, idx, weight, grad_points);
    } else {
        dim3 grid(ceil_div(n, 256), c, b);
        hipLaunchKernelGGL(three_interp_batch_bwd_atomic_kernel, grid, dim3(256), 0, (hipStream_t)stream, c, n, m,
                           grad_out, go_bs, idx, weight, grad_points);
    }
    return check_launch("three_interpolate_grad_batch: launch failed");
}
extern "C" __attribute__((visibility("default"))) int mgar_three_interpolate_grad_batch(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                                 const float *weight, float *grad_points, void *stream) {
    return mgar_three_interpolate_grad_batch_strided(b, c, n, m, grad_out, (long long)c * n, idx, weight, grad_points, stream);
}

template <typename T>
static int three_interpolate_stack_impl(int N, int C, const T *features, const int *idx, const float *weight, T *out, void *stream) {
    MGAR_REQUIRE(N >= 0 && C >= 0, "three_interpolate_stack: negative size");
    const long long total = (long long)N * C;
    if (total == 0) return MGAR_OK;
    MGAR_REQUIRE(features && idx && weight && out, "three_interpolate_stack: null pointer");
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(three_interp_stack_fwd_kernel<T>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, total, C,
                       features, idx, weight, out);
    return check_launch("three_interpolate_stack: launch failed");
}
extern "C" __attribute__((visibility("default"))) int mgar_three_interpolate_stack(int N, int C, const float *features, const int *idx, const float *weight,
                                            float *out, void *stream) {
    return three_interpolate_stack_impl<float>(N, C, features, idx, weight, out, stream);
}
extern "C" __attribute__((visibility("default"))) int mgar_three_interpolate_stack_bf16(int N, int C, const void *features, const int *idx,
                                            const float *weight, void *out, void *stream) {
    return three_interpolate_stack_impl<bf16_t>(N, C, (const bf16_t *)features, idx, weight, (bf16_t *)out, stream);
}

extern "C" __attribute__((visibility("default"))) int mgar_three_interpolate_grad_stack(int N, int C, const float *grad_out, const int *idx,
                                                 const float *weight, float *grad_features, void *stream) {
    MGAR_REQUIRE(N >= 0 && C >= 0, "three_interpolate_grad_stack: negative size");
    const long long total = (long long)N * C;
    if (total == 0) return MGAR_OK;
    MGAR_REQUIRE(grad_out && idx && weight && grad_features, "three_interpolate_grad_stack: null pointer");
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(three_interp_stack_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, total, C,
                       grad_out, idx, weight, grad_features);
    return check_launch("three_interpolate_grad_stack: launch failed");
}

template <int CH>
static void launch_bwd_sorted(int b, int c, int n, int m, const float *grad_out, size_t go_bs, const int *list, float *grad_points,
                              hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)mgar::three_interp_batch_bwd_sorted_kernel<CH>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, mgar::TI_LDS_MAX_FLOATS * (int)sizeof(float));
        attr_set = true;
    }
    hipLaunchKernelGGL((mgar::three_interp_batch_bwd_sorted_kernel<CH>), dim3(ceil_div(c, CH), b), dim3(n >= 1024 ? 1024 : 256),
                       (size_t)CH * n * sizeof(float), st, c, n, m, grad_out, go_bs, reinterpret_cast<const int2 *>(list), grad_points);
}

extern "C" __attribute__((visibility("default"))) int mgar_three_interpolate_grad_sorted_batch_strided(int b, int c, int n, int m,
                                                                                                       const float *grad_out,
                                                                                                       long long grad_out_bstride,
                                                                                                       const int *list,
                                                                                                       float *grad_points, void *stream) {
    MGAR_REQUIRE(b >= 0 && c >= 0 && m >= 0 && n >= 0, "three_interpolate_grad_sorted_batch: negative size");
    MGAR_REQUIRE(grad_out_bstride >= (long long)c * n, "three_interpolate_grad_sorted_batch: grad_out batch stride smaller than a sample");
    MGAR_REQUIRE((n & 3) != 0 || (grad_out_bstride % 4 == 0 && (uintptr_t)grad_out % 16 == 0),
                 "three_interpolate_grad_sorted_batch: grad_out slice not 16-byte aligned");
    const size_t go_bs = (size_t)grad_out_bstride;
    MGAR_REQUIRE(b <= 65535, "three_interpolate_grad_sorted_batch: b > 65535");
    if ((long long)b * c * n == 0 || m == 0) return MGAR_OK;
    MGAR_REQUIRE(grad_out && list && grad_points, "three_interpolate_grad_sorted_batch: null pointer");
    if (n > TI_LDS_MAX_FLOATS || m > 65535) {
        set_error("three_interpolate_grad_sorted_batch: needs n <= 36864 (LDS row) and m <= 65535 (packed entry)");
        return MGAR_EUNSUPPORTED;
    }
    int ch = TI_LDS_MAX_FLOATS / n;
    ch = ch >= 8 ? 8 : (ch >= 4 ? 4 : (ch >= 2 ? 2 : 1));
    while (ch > 1 && (long long)b * ceil_div(c, ch) < 512) ch >>= 1;
    hipStream_t st = (hipStream_t)stream;
    KtScope kt(KT_THREE_INTERP_BWD, st, (double)b * (24.0 * n + 4.0 * c * m + 4.0 * c * n));   // grad_out once, grad_points once, the 3n-entry list
    if (ch == 8) launch_bwd_sorted<8>(b, c, n, m, grad_out, go_bs, list, grad_points, st);
    else if (ch == 4) launch_bwd_sorted<4>(b, c, n, m, grad_out, go_bs, list, grad_points, st);
    else if (ch == 2) launch_bwd_sorted<2>(b, c, n, m, grad_out, go_bs, list, grad_points, st);
    else launch_bwd_sorted<1>(b, c, n, m, grad_out, go_bs, list, grad_points, st);
    return check_launch("three_interpolate_grad_sorted_batch: launch failed");
}
extern "C" __attribute__((visibility("default"))) int mgar_three_interpolate_grad_sorted_batch(int b, int c, int n, int m,
                                                                                               const float *grad_out,
                                                                                               const int *list,
                                                                                               float *grad_points, void *stream) {
    return mgar_three_interpolate_grad_sorted_batch_strided(b, c, n, m, grad_out, (long long)c * n, list, grad_points, stream);
}
