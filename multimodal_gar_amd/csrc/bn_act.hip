// bn_act.hip -- BatchNorm (training statistics) + ReLU + max-over-nsample for the shared MLPs of
// the set-abstraction / RoI-pooling modules, forward and backward, for gfx950.
//
// Replaces the per-layer torch chain of the reference's shared MLPs
//   [Conv2d 1x1 -> BatchNorm2d -> ReLU] x k -> F.max_pool2d(kernel=[1, nsample])
//   (pointnet2_batch/pointnet2_modules.py:37-45, :86-95; pointnet2_stack/pointnet2_modules.py:60-66,
//    :96-104; voxel_pool_modules.py:44-58,:108-125)
// for everything after the 1x1 convolution (which stays a library GEMM).
//
// Why a kernel: the activations are (B, C, P) with FEW channels (16..128) and HUGE P (up to
// 3.3 M columns per sample).  MIOpen's spatial BatchNorm parallelises over channels, so C = 32
// keeps 32 of 256 CUs busy (2.0 ms for a 0.42 GB tensor = 0.6 TB/s measured); ReLU and the max
// over nsample are two more full passes.  Here
//   * statistics: every (channel, 64 K-element chunk) is its own workgroup -> thousands of
//     workgroups, fp32 partial sums, combined in double;
//   * apply: one streaming pass y = relu(x * scale_c + shift_c), 16 B per lane;
//   * the last layer never materialises y: relu(bn(x)) is reduced over nsample on the fly and only
//     (B, C, M) maxima + 1-byte arg-max are written;
//   * backward: one reduction pass (d_beta, d_gamma), one streaming pass for dx; after a fused
//     max-pool the reduction touches only the arg-max elements.
// All kernels are HBM-bound streaming passes: bytes per element are listed at each kernel.
#include "common.hpp"
#include "payload.hpp"

namespace mgar {

constexpr int BN_THREADS = 256;
constexpr int BN_CHUNK = 65536;  // most elements of one channel reduced by one workgroup (bn_chunk() shrinks it for small inputs)

__device__ __forceinline__ float block_sum(float v, float *scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < BN_THREADS / 64; ++i) t += scratch[i];
    return t;
}

// element e of channel c (0 <= e < B*P) lives at ((e / P) * C + c) * P + e % P
__device__ __forceinline__ size_t chan_off(long long e, int c, int C, int P) {
    const long long b = e / P;
    return ((size_t)b * C + c) * P + (size_t)(e - b * P);
}

// ---- statistics: 4 B read per element -------------------------------------------------------
// grid (nchunk, C).  Cancellation-safe: every workgroup shifts its chunk by a chunk-local pivot (the mean of the
// chunk's first <= 256 elements -- one per thread, so a single outlier moves it by 1/256 of itself) and accumulates
// sum / sum of squares of (x - pivot) in fp32: the squares are of the order of the variance, not of mean^2.
// partial[(c * nchunk + chunk) * 2 + {0,1}] = chunk mean, chunk M2 = sum (x - chunk mean)^2; bn_finalize_kernel merges
// the chunks with Chan's parallel formula in double.  (With |mean| >> std a plain sum / sum-of-squares loses the
// variance digits already in the fp32 partials: ADVICE r1.)
template <typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_partial_kernel(const T *__restrict__ x, int B, int C, int P, int chunk,
                                                                float *__restrict__ partial) {
    __shared__ float scratch[BN_THREADS / 64];
    const int c = blockIdx.y;
    const long long n = (long long)B * P;
    const long long e0 = (long long)blockIdx.x * chunk;
    const long long e1 = min(e0 + chunk, n);
    const long long npiv = min((long long)BN_THREADS, e1 - e0);
    float pv = threadIdx.x < npiv ? Payload<T>::ld(x + chan_off(e0 + threadIdx.x, c, C, P)) : 0.f;
    const float pivot = block_sum(pv, scratch) / (float)npiv;
    float s = 0.f, q = 0.f;
    if ((P & 3) == 0) {
#pragma unroll 4
        for (long long e = e0 + (long long)threadIdx.x * 4; e < e1; e += BN_THREADS * 4) {
            float4 v = Payload<T>::ld4(x + chan_off(e, c, C, P));
            v.x -= pivot; v.y -= pivot; v.z -= pivot; v.w -= pivot;
            s += (v.x + v.y) + (v.z + v.w);
            q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
    } else {
        for (long long e = e0 + threadIdx.x; e < e1; e += BN_THREADS) {
            const float v = Payload<T>::ld(x + chan_off(e, c, C, P)) - pivot;
            s += v;
            q += v * v;
        }
    }
    s = block_sum(s, scratch);
    q = block_sum(q, scratch);
    if (threadIdx.x == 0) {
        const double nk = (double)(e1 - e0), sd = (double)s;
        double m2 = (double)q - sd * sd / nk;
        if (m2 < 0.0) m2 = 0.0;
        partial[((size_t)c * gridDim.x + blockIdx.x) * 2 + 0] = (float)((double)pivot + sd / nk);
        partial[((size_t)c * gridDim.x + blockIdx.x) * 2 + 1] = (float)m2;
    }
}

// one wave per channel: merge the chunk (count, mean, M2) triples in double (Chan et al.: M2 = sum M2_k +
// sum n_k (mean_k - mean)^2; lanes stride over the chunks, fixed-order shuffle tree), write mean / invstd, update
// running stats.  chunk = elements per chunk (the last one holds the remainder of n).
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__global__ __launch_bounds__(64) void bn_finalize_kernel(const float *__restrict__ partial, int nchunk, int C, double n, int chunk,
                                                         float eps, float momentum, float *__restrict__ mean,
                                                         float *__restrict__ invstd, float *__restrict__ running_mean,
                                                         float *__restrict__ running_var,
                                                         long long *__restrict__ num_batches_tracked,
                                                         float *__restrict__ var_out) {
    const int c = blockIdx.x;
    auto count = [&](int i) { return i + 1 < nchunk ? (double)chunk : n - (double)chunk * (nchunk - 1); };
    double s = 0.0;
    for (int i = threadIdx.x; i < nchunk; i += 64) s += count(i) * (double)partial[((size_t)c * nchunk + i) * 2 + 0];
    const double m = wave_sum_f64(s) / n;
    double q = 0.0;
    for (int i = threadIdx.x; i < nchunk; i += 64) {
        const double d = (double)partial[((size_t)c * nchunk + i) * 2 + 0] - m;
        q += (double)partial[((size_t)c * nchunk + i) * 2 + 1] + count(i) * d * d;
    }
    q = wave_sum_f64(q);
    if (threadIdx.x != 0) return;
    double var = q / n;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (var_out) var_out[c] = (float)var;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(n > 1.0 ? var * n / (n - 1.0) : var);
    if (num_batches_tracked && c == 0) *num_batches_tracked += 1;
}

// ---- small channels: statistics + apply in ONE launch (round 2) ---------------------------------------------------------
// When a whole channel (n = B * P <= BN_SMALL_MAX elements) fits the registers of one workgroup, the three launches
// (partial, finalize, apply) collapse into one: read x once (up to 16 float4 per thread), exact two-pass mean / M2 in
// registers, running statistics, apply, write y.  At one clip per rank ~45 of the I3D's BatchNorms are this small and the
// step is launch-bound (profiles/README.md, round 2).  P % 4 == 0.  grid (rows): rows = C, or G * C with per-sample
// statistics (then B = 1 per row).  y may be a channel slice of a wider tensor (y_bstride, as bn_apply_kernel).
constexpr int BN_SMALL_V = 16, BN_SMALL_MAX = BN_THREADS * 4 * BN_SMALL_V;   // 16 384 elements per channel
template <bool RELU, typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_small_fused_kernel(const T *__restrict__ x, int B, int C, int P, int per_sample,
                                                                    float eps, float momentum, const float *__restrict__ gamma,
                                                                    const float *__restrict__ beta, float *__restrict__ mean_out,
                                                                    float *__restrict__ invstd_out, float *__restrict__ var_out,
                                                                    float *__restrict__ running_mean, float *__restrict__ running_var,
                                                                    long long *__restrict__ num_batches_tracked, T *__restrict__ y,
                                                                    long long y_bstride) {
    __shared__ float scratch[BN_THREADS / 64];
    const int row = blockIdx.x;                          // per_sample: row = g * C + c, its P elements are contiguous
    const int c = per_sample ? row % C : row;
    const int nb = per_sample ? 1 : B;
    const long long n = (long long)nb * P;
    float4 v[BN_SMALL_V];
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < BN_SMALL_V; ++u) {
        const long long e = ((long long)u * BN_THREADS + threadIdx.x) * 4;
        if (e < n) {
            const size_t o = per_sample ? (size_t)row * P + e : chan_off(e, c, C, P);
            v[u] = Payload<T>::ld4(x + o);
            s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
        } else {
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float mu = block_sum(s, scratch) / (float)n;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < BN_SMALL_V; ++u) {
        const long long e = ((long long)u * BN_THREADS + threadIdx.x) * 4;
        if (e < n) {
            const float dx = v[u].x - mu, dy = v[u].y - mu, dz = v[u].z - mu, dw = v[u].w - mu;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
    __syncthreads();                                     // block_sum reuses its scratch
    const float var = fmaxf(block_sum(q, scratch) / (float)n, 0.f);
    const float is = (float)(1.0 / sqrt((double)var + (double)eps));
    if (threadIdx.x == 0) {
        if (mean_out) mean_out[row] = mu;
        if (invstd_out) invstd_out[row] = is;
        if (var_out) var_out[row] = var;                 // per-sample statistics: the running update walks the samples in order
        if (!per_sample) {
            if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
            if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (n > 1 ? var * (float)n / (float)(n - 1) : var);
            if (num_batches_tracked && c == 0) *num_batches_tracked += 1;
        }
    }
    const float sc = is * (gamma ? gamma[c] : 1.f), sh = beta ? beta[c] : 0.f;
#pragma unroll
    for (int u = 0; u < BN_SMALL_V; ++u) {
        const long long e = ((long long)u * BN_THREADS + threadIdx.x) * 4;
        if (e < n) {
            float4 r = v[u];
            r.x = (r.x - mu) * sc + sh; r.y = (r.y - mu) * sc + sh; r.z = (r.z - mu) * sc + sh; r.w = (r.w - mu) * sc + sh;
            if (RELU) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
            size_t o;
            if (per_sample) {
                o = (size_t)(row / C) * y_bstride + (size_t)c * P + e;
            } else {
                const long long b = e / P;
                o = (size_t)b * y_bstride + (size_t)c * P + (size_t)(e - b * P);
            }
            Payload<T>::st4(y + o, r);
        }
    }
}

// ---- apply: 4 B read + 4 B written per element ----------------------------------------------
// grid (B*C rows, ceil(P / (256*4*BN_APPLY_V)))
constexpr int BN_APPLY_V = 4;  // float4 per thread

template <bool RELU, typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(const T *__restrict__ x, int C, int P,
                                                              const float *__restrict__ mean, const float *__restrict__ invstd,
                                                              const float *__restrict__ gamma, const float *__restrict__ beta,
                                                              T *__restrict__ y, int stats_per_row, long long y_bstride) {
    const int row = blockIdx.x;
    const int c = row % C;
    const int sidx = stats_per_row ? row : c;   // per-sample statistics: mean / invstd have one entry per (sample, channel)
    // y = (x - mean) * sc + beta, not x * sc + (beta - mean * sc): with |mean| >> std the second form rounds at the
    // magnitude of mean * sc (x - mean is exact for x within a factor 2 of mean)
    const float sc = invstd[sidx] * (gamma ? gamma[c] : 1.f);
    const float mu = mean[sidx], sh = beta ? beta[c] : 0.f;
    const T *xr = x + (size_t)row * P;
    T *yr = y + (size_t)(row / C) * y_bstride + (size_t)c * P;   // y_bstride = C * P, or more: y is a channel slice of a wider tensor
    const int base = blockIdx.y * (BN_THREADS * 4 * BN_APPLY_V);
    if ((P & 3) == 0) {
#pragma unroll
        for (int u = 0; u < BN_APPLY_V; ++u) {
            const int p = base + (u * BN_THREADS + threadIdx.x) * 4;
            if (p < P) {
                float4 v = Payload<T>::ld4(xr + p);
                v.x = (v.x - mu) * sc + sh; v.y = (v.y - mu) * sc + sh; v.z = (v.z - mu) * sc + sh; v.w = (v.w - mu) * sc + sh;
                if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                Payload<T>::st4(yr + p, v);
            }
        }
    } else {
        for (int p = base + threadIdx.x; p < min(P, base + BN_THREADS * 4 * BN_APPLY_V); p += BN_THREADS) {
            float v = (Payload<T>::ld(xr + p) - mu) * sc + sh;
            Payload<T>::st(yr + p, RELU ? fmaxf(v, 0.f) : v);
        }
    }
}

// ---- apply + max over nsample: 4 B read per element, 5 B written per GROUP ---------------------
// x (rows = B*C, M, NS) -> out (rows, M), arg (rows, M) uint8 (first arg-max).
// Lanes run along the flat (m, s) index, one float4 per lane (fully coalesced 16-byte reads); the
// NS/4 lanes that share a group combine their (value, index) pairs with an xor butterfly on the DPP
// crossbar (quad_perm, row_half_mirror, row_mirror).  grid (rows, ceil(M*NS/4 / 256))
__device__ __forceinline__ void max_pair(float &v, int &i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}

template <int CTRL>
__device__ __forceinline__ void dpp_max_step(float &v, int &i) {
    const float ov = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
    const int oi = __builtin_amdgcn_update_dpp(0, i, CTRL, 0xF, 0xF, false);
    max_pair(v, i, ov, oi);
}

constexpr int BN_MAX_V = 4;   // float4 per thread: four independent 16-byte loads in flight

template <bool RELU, int G, typename T>   // G = NS / 4 lanes per group: 1, 2, 4, 8 or 16
__global__ __launch_bounds__(BN_THREADS) void bn_max_vec_kernel(const T *__restrict__ x, int C, int M, int NS,
                                                                const float *__restrict__ mean, const float *__restrict__ invstd,
                                                                const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                T *__restrict__ out, unsigned char *__restrict__ arg,
                                                                T *__restrict__ xarg) {
    const int row = blockIdx.x;
    const int c = row % C;
    const float sc = invstd[c] * (gamma ? gamma[c] : 1.f);
    const float mu = mean[c], sh = beta ? beta[c] : 0.f;
    const long long nq = (long long)M * G;
    const T *xr = x + (size_t)row * M * NS;
    float4 v[BN_MAX_V];
    long long qs[BN_MAX_V];
#pragma unroll
    for (int u = 0; u < BN_MAX_V; ++u) {   // float4 index inside the row; whole groups are live or dead together
        qs[u] = ((long long)blockIdx.y * BN_MAX_V + u) * BN_THREADS + threadIdx.x;
        v[u] = qs[u] < nq ? Payload<T>::ld4(xr + qs[u] * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < BN_MAX_V; ++u) {
        const long long q = qs[u];
        const bool live = q < nq;
        float best = -__builtin_inff();
        int bi = 0;
        if (live) {
            const int s0 = (int)(q % G) * 4;
            const float a[4] = {(v[u].x - mu) * sc + sh, (v[u].y - mu) * sc + sh, (v[u].z - mu) * sc + sh, (v[u].w - mu) * sc + sh};
#pragma unroll
            for (int w = 0; w < 4; ++w)
                if (a[w] > best) { best = a[w]; bi = s0 + w; }
        }
        if (G >= 2) dpp_max_step<0xB1>(best, bi);    // quad_perm [1,0,3,2]  (xor 1)
        if (G >= 4) dpp_max_step<0x4E>(best, bi);    // quad_perm [2,3,0,1]  (xor 2)
        if (G >= 8) dpp_max_step<0x141>(best, bi);   // row_half_mirror      (acts as xor 4 once quads agree)
        if (G >= 16) dpp_max_step<0x140>(best, bi);  // row_mirror           (acts as xor 8)
        if (live && (threadIdx.x & (G - 1)) == 0) {
            const long long m = q / G;
            Payload<T>::st(out + (size_t)row * M + m, RELU ? fmaxf(best, 0.f) : best);
            arg[(size_t)row * M + m] = (unsigned char)bi;
            // the pre-BN value at the arg-max (the line was just read: an L1/L2 hit), so that the backward
            // reduction reads three coalesced (B,C,M) arrays instead of gathering one element per group
            if (xarg) xarg[(size_t)row * M + m] = xr[m * NS + bi];   // a copy: no conversion
        }
    }
}

// generic fallback (NS not in {4, 8, 16, 32, 64}): one thread per group.  grid (rows, ceil(M/256))
template <bool RELU, typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_max_kernel(const T *__restrict__ x, int C, int M, int NS,
                                                            const float *__restrict__ mean, const float *__restrict__ invstd,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            T *__restrict__ out, unsigned char *__restrict__ arg,
                                                            T *__restrict__ xarg) {
    const int row = blockIdx.x;
    const int m = blockIdx.y * BN_THREADS + threadIdx.x;
    if (m >= M) return;
    const int c = row % C;
    const float sc = invstd[c] * (gamma ? gamma[c] : 1.f);
    const float mu = mean[c], sh = beta ? beta[c] : 0.f;
    const T *xr = x + ((size_t)row * M + m) * NS;
    float best = -__builtin_inff();
    int bi = 0;
    for (int s = 0; s < NS; ++s) {
        const float a = (Payload<T>::ld(xr + s) - mu) * sc + sh;
        if (a > best) { best = a; bi = s; }
    }
    Payload<T>::st(out + (size_t)row * M + m, RELU ? fmaxf(best, 0.f) : best);
    arg[(size_t)row * M + m] = (unsigned char)bi;
    if (xarg) xarg[(size_t)row * M + m] = xr[bi];
}

// ---- backward reduction: 8 B read per element ---------------------------------------------------
// partial[(c*nchunk + chunk)*2 + {0,1}] = sum dz, sum dz * xhat    with dz = dy * [relu active]
template <bool RELU, typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_partial_kernel(const T *__restrict__ dy, const T *__restrict__ x,
                                                                    int B, int C, int P, const float *__restrict__ mean,
                                                                    const float *__restrict__ invstd,
                                                                    const float *__restrict__ gamma,
                                                                    const float *__restrict__ beta, float *__restrict__ partial,
                                                                    int chunk) {
    __shared__ float scratch[BN_THREADS / 64];
    const int c = blockIdx.y;
    const float mu = mean[c], is = invstd[c];
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const long long n = (long long)B * P;
    const long long e0 = (long long)blockIdx.x * chunk;
    const long long e1 = min(e0 + chunk, n);
    float s = 0.f, q = 0.f;
    const float sc = is * g;
    auto acc = [&](float xv, float d) {
        const float xh = (xv - mu) * is;
        if (RELU && !((xv - mu) * sc + b > 0.f)) d = 0.f;   // the forward's expression, bit for bit
        s += d;
        q += d * xh;
    };
    if ((P & 3) == 0) {
        for (long long e = e0 + (long long)threadIdx.x * 4; e < e1; e += BN_THREADS * 4) {
            const size_t o = chan_off(e, c, C, P);
            const float4 xv = Payload<T>::ld4(x + o);
            const float4 dv = Payload<T>::ld4(dy + o);
            acc(xv.x, dv.x); acc(xv.y, dv.y); acc(xv.z, dv.z); acc(xv.w, dv.w);
        }
    } else {
        for (long long e = e0 + threadIdx.x; e < e1; e += BN_THREADS) {
            const size_t o = chan_off(e, c, C, P);
            acc(Payload<T>::ld(x + o), Payload<T>::ld(dy + o));
        }
    }
    s = block_sum(s, scratch);
    q = block_sum(q, scratch);
    if (threadIdx.x == 0) {
        partial[((size_t)c * gridDim.x + blockIdx.x) * 2 + 0] = s;
        partial[((size_t)c * gridDim.x + blockIdx.x) * 2 + 1] = q;
    }
}

// where element (b, c, m) of the pooled gradient lives: contiguous (C*M, M, 1); a channel slice of a wider (B, C_total, M) tensor
// (C_total*M, M, 1); or a transposed view of (M, C_total) rows (-, 1, C_total) -- read in place instead of copied first
struct DpoolStrides {
    long long b, c, m;
};

// after a fused max-pool only the arg-max element of every group carries gradient:
// 13 B read per GROUP.  grid (nchunk over B*M, C)
template <bool RELU, typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_max_bwd_partial_kernel(const T *__restrict__ dpool,
                                                                        const T *__restrict__ pooled,
                                                                        const unsigned char *__restrict__ arg,
                                                                        const T *__restrict__ x,
                                                                        const T *__restrict__ xarg, int B, int C, int M, int NS,
                                                                        const float *__restrict__ mean,
                                                                        const float *__restrict__ invstd,
                                                                        float *__restrict__ partial, DpoolStrides ds, int chunk) {
    __shared__ float scratch[BN_THREADS / 64];
    const int c = blockIdx.y;
    const float mu = mean[c], is = invstd[c];
    const long long n = (long long)B * M;
    const long long e0 = (long long)blockIdx.x * chunk;
    const long long e1 = min(e0 + chunk, n);
    float s = 0.f, q = 0.f;
    for (long long e = e0 + threadIdx.x; e < e1; e += BN_THREADS) {
        const size_t o = chan_off(e, c, C, M);
        const long long eb = e / M;
        float d = Payload<T>::ld(dpool + eb * ds.b + c * ds.c + (e - eb * M) * ds.m);
        if (RELU && !(Payload<T>::ld(pooled + o) > 0.f)) d = 0.f;
        const float xh = ((xarg ? Payload<T>::ld(xarg + o) : Payload<T>::ld(x + o * NS + arg[o])) - mu) * is;
        s += d;
        q += d * xh;
    }
    s = block_sum(s, scratch);
    q = block_sum(q, scratch);
    if (threadIdx.x == 0) {
        partial[((size_t)c * gridDim.x + blockIdx.x) * 2 + 0] = s;
        partial[((size_t)c * gridDim.x + blockIdx.x) * 2 + 1] = q;
    }
}

// d_beta[c] = sum dz ; d_gamma[c] = sum dz*xhat ; coef[c] = {mean dz, mean dz*xhat}.  One wave per channel.
__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(const float *__restrict__ partial, int nchunk, int C, double n,
                                                             float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                             float *__restrict__ coef) {
    const int c = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < nchunk; i += 64) {
        s += (double)partial[((size_t)c * nchunk + i) * 2 + 0];
        q += (double)partial[((size_t)c * nchunk + i) * 2 + 1];
    }
    s = wave_sum_f64(s);
    q = wave_sum_f64(q);
    if (threadIdx.x != 0) return;
    if (dbeta) dbeta[c] = (float)s;
    if (dgamma) dgamma[c] = (float)q;
    coef[2 * c + 0] = (float)(s / n);
    coef[2 * c + 1] = (float)(q / n);
}

// ---- backward apply: dx = gamma*invstd * (dz - mean(dz) - xhat * mean(dz*xhat)) ----------------
// 8 B read + 4 B written per element.  grid (ceil(P/(256*4)), B*C)
template <bool RELU, typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_kernel(const T *__restrict__ dy, const T *__restrict__ x, int C,
                                                                  int P, const float *__restrict__ mean,
                                                                  const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                                  const float *__restrict__ beta, const float *__restrict__ coef,
                                                                  T *__restrict__ dx) {
    const int row = blockIdx.x;
    const int c = row % C;
    const float mu = mean[c], is = invstd[c];
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float m0 = coef[2 * c], m1 = coef[2 * c + 1];
    const float k = is * g;
    const size_t ro = (size_t)row * P;
    auto one = [&](float xv, float d) {
        const float xh = (xv - mu) * is;
        if (RELU && !((xv - mu) * k + b > 0.f)) d = 0.f;    // k = invstd * gamma as in the forward
        return k * (d - m0 - xh * m1);
    };
    if ((P & 3) == 0) {
        const int p0 = (blockIdx.y * BN_THREADS + threadIdx.x) * 4;
        if (p0 >= P) return;
        const float4 xv = Payload<T>::ld4(x + ro + p0);
        const float4 dv = Payload<T>::ld4(dy + ro + p0);
        float4 r;
        r.x = one(xv.x, dv.x); r.y = one(xv.y, dv.y); r.z = one(xv.z, dv.z); r.w = one(xv.w, dv.w);
        Payload<T>::st4(dx + ro + p0, r);
    } else {
        const int p0 = blockIdx.y * BN_THREADS + threadIdx.x;
        if (p0 < P) Payload<T>::st(dx + ro + p0, one(Payload<T>::ld(x + ro + p0), Payload<T>::ld(dy + ro + p0)));
    }
}

// The same with a ROW-MAJOR result dx_t (B*P, C): the layout the stacked query-and-group backward gathers row by row
// (csrc/query_group.hip, qg_stack_bwd_rows_kernel).  A workgroup owns 64 columns of all C <= 64 channels: coalesced
// channel-major reads, an LDS transpose, coalesced row-major writes.  grid (ceil(P / 64), B)
template <bool RELU>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_t_kernel(const float *__restrict__ dy, const float *__restrict__ x, int C,
                                                                    int P, const float *__restrict__ mean,
                                                                    const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                                    const float *__restrict__ beta, const float *__restrict__ coef,
                                                                    float *__restrict__ dx_t) {
    __shared__ float tile[64][65];
    const int b = blockIdx.y, p0 = blockIdx.x * 64;
    const int np = min(64, P - p0);
    for (int e = threadIdx.x; e < C * 64; e += BN_THREADS) {
        const int c = e >> 6, pl = e & 63;
        if (pl >= np) continue;
        const size_t o = ((size_t)b * C + c) * P + p0 + pl;
        const float mu = mean[c], is = invstd[c];
        const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
        const float k = is * g;
        const float xv = x[o];
        float d = dy[o];
        const float xh = (xv - mu) * is;
        if (RELU && !((xv - mu) * k + bt > 0.f)) d = 0.f;
        tile[pl][c] = k * (d - coef[2 * c] - xh * coef[2 * c + 1]);
    }
    __syncthreads();
    float *dst = dx_t + ((size_t)b * P + p0) * C;
    for (int e = threadIdx.x; e < np * C; e += BN_THREADS) dst[e] = tile[e / C][e % C];
}

// after a fused max-pool: 4 B read + 4 B written per element (+ 9 B per group, cached).
// lanes run along the flat (m, s) index, 4 elements per lane.  grid (ceil(M*NS/(256*4)), B*C)
template <bool RELU, typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_max_bwd_apply_kernel(const T *__restrict__ dpool,
                                                                      const T *__restrict__ pooled,
                                                                      const unsigned char *__restrict__ arg,
                                                                      const T *__restrict__ x, int C, int M, int NS,
                                                                      const float *__restrict__ mean,
                                                                      const float *__restrict__ invstd,
                                                                      const float *__restrict__ gamma,
                                                                      const float *__restrict__ coef, T *__restrict__ dx,
                                                                      DpoolStrides ds) {
    const int row = blockIdx.x;
    const int c = row % C;
    const float mu = mean[c], is = invstd[c];
    const float k = (gamma ? gamma[c] : 1.f) * is;
    const float m0 = coef[2 * c], m1 = coef[2 * c + 1];
    const long long P = (long long)M * NS;
    const size_t ro = (size_t)row * P;
    const size_t go = (size_t)row * M;
    const int step = (NS & 3) == 0 ? 4 : 1;
    const long long p0 = ((long long)blockIdx.y * BN_THREADS + threadIdx.x) * step;
    if (p0 >= P) return;
    const int m = (int)(p0 / NS), s0 = (int)(p0 - (long long)m * NS);
    float d = Payload<T>::ld(dpool + (long long)(row / C) * ds.b + c * ds.c + m * ds.m);
    if (RELU && !(Payload<T>::ld(pooled + go + m) > 0.f)) d = 0.f;
    const int a = arg[go + m];
    if (step == 4) {
        const float4 xv = Payload<T>::ld4(x + ro + p0);
        float4 r;
        r.x = k * ((s0 + 0 == a ? d : 0.f) - m0 - (xv.x - mu) * is * m1);
        r.y = k * ((s0 + 1 == a ? d : 0.f) - m0 - (xv.y - mu) * is * m1);
        r.z = k * ((s0 + 2 == a ? d : 0.f) - m0 - (xv.z - mu) * is * m1);
        r.w = k * ((s0 + 3 == a ? d : 0.f) - m0 - (xv.w - mu) * is * m1);
        Payload<T>::st4(dx + ro + p0, r);
    } else {
        Payload<T>::st(dx + ro + p0, k * ((s0 == a ? d : 0.f) - m0 - (Payload<T>::ld(x + ro + p0) - mu) * is * m1));
    }
}

// chunk of the FORWARD statistics pass: shrink it until ~2048 workgroups exist (the I3D layers of a
// single clip have as few as 14 400 elements per channel)
static inline int bn_chunk(int B, int C, long long P) {
    const long long n = (long long)B * P;
    int chunk = BN_CHUNK;
    while (chunk > 4096 && ((n + chunk - 1) / chunk) * C < 2048) chunk >>= 1;
    return chunk;
}
static inline int bn_nchunk_fwd(int B, int C, long long P) {
    const int chunk = bn_chunk(B, C, P);
    return (int)(((long long)B * P + chunk - 1) / chunk);
}
static inline int bn_nchunk(int B, int P) { return (int)(((long long)B * P + BN_CHUNK - 1) / BN_CHUNK); }
// the BACKWARD reductions shrink their chunk the same way (round 2: at one clip per rank the fixed 65 536-element chunk left
// 32 workgroups walking 61 440 elements each: 47-84 us per launch).  n = elements per channel of THIS reduction (B*P, or B*M
// after a max-pool); never more chunks than mgar_bn_workspace_floats() reserves for the layer's (B, C, P).
static inline int bn_chunk_bwd(int C, long long n) {
    int chunk = BN_CHUNK;
    while (chunk > 4096 && ((n + chunk - 1) / chunk) * C < 2048) chunk >>= 1;
    return chunk;
}

}  // namespace mgar

using namespace mgar;

#define BN_API extern "C" __attribute__((visibility("default")))

BN_API int mgar_bn_workspace_floats(int B, int C, int P) {
    if (B < 0 || C < 0 || P < 0) return MGAR_EINVAL;
    int nc = bn_nchunk_fwd(B, C, P) > bn_nchunk(B, P) ? bn_nchunk_fwd(B, C, P) : bn_nchunk(B, P);
    const int small = 4096 / (C > 0 ? C : 1) + 2;   // bound of the chunk count of a reduction over fewer elements (bn_chunk_bwd)
    if (nc < small) nc = small;
    return 2 * C * (nc > 0 ? nc : 1) + 2 * C;   // [partials | 2*C: bwd coefficients, or C: variances of the grouped statistics]
}

static int bn_sizes_ok(int B, int C, long long P) { return B >= 0 && C >= 0 && P >= 0 && (long long)B * C <= 2147483647LL; }

// ---- implementations, templated on the payload type (float / bf16_t); statistics, affine and gradients of the affine
// are always fp32 ----
template <typename T>
static int bn_train_stats_impl(const T *x, int B, int C, int P, float eps, float momentum, float *workspace, float *mean,
                               float *invstd, float *running_mean, float *running_var, long long *num_batches_tracked,
                               void *stream) {
    MGAR_REQUIRE(bn_sizes_ok(B, C, P), "bn_train_stats: bad sizes");
    if ((long long)B * C * P == 0) return MGAR_OK;
    MGAR_REQUIRE(x && workspace && mean && invstd, "bn_train_stats: null pointer");
    MGAR_REQUIRE(C <= 65535, "bn_train_stats: C > 65535");
    const int chunk = bn_chunk(B, C, P), nchunk = bn_nchunk_fwd(B, C, P);
    hipStream_t st = (hipStream_t)stream;
    { KtScope kt(KT_BN_STATS, st, (double)sizeof(T) * B * C * P);
    hipLaunchKernelGGL(bn_partial_kernel<T>, dim3(nchunk, C), dim3(BN_THREADS), 0, st, x, B, C, P, chunk, workspace);
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, st, workspace, nchunk, C, (double)B * P, chunk, eps,
                       momentum, mean, invstd, running_mean, running_var, num_batches_tracked, (float *)nullptr);
    return check_launch("bn_train_stats: launch failed");
}

template <typename T>
static int bn_act_fwd_impl(const T *x, int B, int C, int P, const float *mean, const float *invstd, const float *gamma,
                           const float *beta, int relu, T *y, int stats_per_row, void *stream, long long y_bstride = -1) {
    MGAR_REQUIRE(bn_sizes_ok(B, C, P), "bn_act_fwd: bad sizes");
    if (y_bstride < 0) y_bstride = (long long)C * P;
    MGAR_REQUIRE(y_bstride >= (long long)C * P, "bn_act_fwd: output batch stride smaller than a sample");
    MGAR_REQUIRE((P & 3) != 0 || ((y_bstride * (long long)sizeof(T)) % 16 == 0 && (uintptr_t)y % 16 == 0),
                 "bn_act_fwd: output slice not 16-byte aligned");
    if ((long long)B * C * P == 0) return MGAR_OK;
    MGAR_REQUIRE(x && y && mean && invstd, "bn_act_fwd: null pointer");
    MGAR_REQUIRE((long long)P <= 65535LL * BN_THREADS * 4, "bn_act_fwd: P too large");
    dim3 grid(B * C, ceil_div(P, BN_THREADS * 4 * BN_APPLY_V));
    hipStream_t st = (hipStream_t)stream;
    { KtScope kt(KT_BN_APPLY, st, 2.0 * sizeof(T) * (double)B * C * P);
    if (relu) hipLaunchKernelGGL((bn_apply_kernel<true, T>), grid, dim3(BN_THREADS), 0, st, x, C, P, mean, invstd, gamma, beta, y, stats_per_row, y_bstride);
    else hipLaunchKernelGGL((bn_apply_kernel<false, T>), grid, dim3(BN_THREADS), 0, st, x, C, P, mean, invstd, gamma, beta, y, stats_per_row, y_bstride);
    }
    return check_launch("bn_act_fwd: launch failed");
}

// ---- per-sample ("grouped") statistics: G samples, each normalised with its OWN batch statistics -------------
// = what the reference computes when it pushes G clips through a train-mode BatchNorm one at a time (I3D: one pass
// per clip), but as one launch over the (G, C, P) tensor.  The running statistics receive the G momentum updates
// in sample order.
__global__ void bn_running_update_grouped_kernel(const float *__restrict__ mean, const float *__restrict__ var, int G, int C,
                                                 double n, float momentum, float *__restrict__ running_mean,
                                                 float *__restrict__ running_var, long long *__restrict__ num_batches_tracked) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && num_batches_tracked) *num_batches_tracked += G;
    if (c >= C) return;
    float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 0.f;
    for (int g = 0; g < G; ++g) {
        rm = (1.f - momentum) * rm + momentum * mean[(size_t)g * C + c];
        const double v = (double)var[(size_t)g * C + c];
        rv = (1.f - momentum) * rv + momentum * (float)(n > 1.0 ? v * n / (n - 1.0) : v);
    }
    if (running_mean) running_mean[c] = rm;
    if (running_var) running_var[c] = rv;
}

template <typename T>
static int bn_train_stats_grouped_impl(const T *x, int G, int C, int P, float eps, float momentum, float *workspace,
                                       float *mean, float *invstd, float *running_mean, float *running_var,
                                       long long *num_batches_tracked, void *stream) {
    MGAR_REQUIRE(bn_sizes_ok(G, C, P), "bn_train_stats_grouped: bad sizes");
    if ((long long)G * C * P == 0) return MGAR_OK;
    MGAR_REQUIRE(x && workspace && mean && invstd, "bn_train_stats_grouped: null pointer");
    const int rows = G * C;   // every (sample, channel) row is its own "channel" of a batch of one
    MGAR_REQUIRE(rows <= 65535, "bn_train_stats_grouped: G*C > 65535");
    const int chunk = bn_chunk(1, rows, P), nchunk = bn_nchunk_fwd(1, rows, P);
    hipStream_t st = (hipStream_t)stream;
    float *var = workspace + (size_t)2 * rows * nchunk;   // biased variances, rows floats (inside mgar_bn_workspace_floats(1, G*C, P))
    { KtScope kt(KT_BN_STATS, st, (double)sizeof(T) * G * C * P);
    hipLaunchKernelGGL(bn_partial_kernel<T>, dim3(nchunk, rows), dim3(BN_THREADS), 0, st, x, 1, rows, P, chunk, workspace);
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(rows), dim3(64), 0, st, workspace, nchunk, rows, (double)P, chunk, eps, momentum, mean,
                       invstd, (float *)nullptr, (float *)nullptr, (long long *)nullptr, var);
    if (running_mean || running_var || num_batches_tracked)
        hipLaunchKernelGGL(bn_running_update_grouped_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, st, mean, var, G, C, (double)P,
                           momentum, running_mean, running_var, num_batches_tracked);
    return check_launch("bn_train_stats_grouped: launch failed");
}

template <typename T>
static int bn_act_maxpool_fwd_impl(const T *x, int B, int C, int M, int nsample, const float *mean, const float *invstd,
                                   const float *gamma, const float *beta, int relu, T *out, unsigned char *arg,
                                   T *xarg, void *stream) {
    MGAR_REQUIRE(bn_sizes_ok(B, C, (long long)M * nsample) && nsample >= 1 && nsample <= 255, "bn_act_maxpool_fwd: bad sizes");
    if ((long long)B * C * M == 0) return MGAR_OK;
    MGAR_REQUIRE(x && out && arg && mean && invstd, "bn_act_maxpool_fwd: null pointer");
    MGAR_REQUIRE((long long)M <= 65535LL * BN_THREADS, "bn_act_maxpool_fwd: M too large");
    hipStream_t st = (hipStream_t)stream;
    KtScope kt(KT_BN_MAX, st, (double)B * C * M * ((double)sizeof(T) * nsample + sizeof(T) + 1.0));
#define BN_MAX_VEC(G)                                                                                                  \
    {                                                                                                                  \
        dim3 gv(B * C, ceil_div((long long)M * (G), BN_THREADS * BN_MAX_V));                                                      \
        if (relu) hipLaunchKernelGGL((bn_max_vec_kernel<true, G, T>), gv, dim3(BN_THREADS), 0, st, x, C, M, nsample, mean, invstd, gamma, beta, out, arg, xarg); \
        else hipLaunchKernelGGL((bn_max_vec_kernel<false, G, T>), gv, dim3(BN_THREADS), 0, st, x, C, M, nsample, mean, invstd, gamma, beta, out, arg, xarg);     \
    }
    if (nsample == 4) BN_MAX_VEC(1)
    else if (nsample == 8) BN_MAX_VEC(2)
    else if (nsample == 16) BN_MAX_VEC(4)
    else if (nsample == 32) BN_MAX_VEC(8)
    else if (nsample == 64) BN_MAX_VEC(16)
    else {
        dim3 grid(B * C, ceil_div(M, BN_THREADS));
        if (relu) hipLaunchKernelGGL((bn_max_kernel<true, T>), grid, dim3(BN_THREADS), 0, st, x, C, M, nsample, mean, invstd, gamma, beta, out, arg, xarg);
        else hipLaunchKernelGGL((bn_max_kernel<false, T>), grid, dim3(BN_THREADS), 0, st, x, C, M, nsample, mean, invstd, gamma, beta, out, arg, xarg);
    }
#undef BN_MAX_VEC
    return check_launch("bn_act_maxpool_fwd: launch failed");
}

template <typename T>
static int bn_act_bwd_impl(const T *dy, const T *x, int B, int C, int P, const float *mean, const float *invstd,
                           const float *gamma, const float *beta, int relu, float *workspace, float *dgamma, float *dbeta,
                           T *dx, void *stream) {
    MGAR_REQUIRE(bn_sizes_ok(B, C, P), "bn_act_bwd: bad sizes");
    if ((long long)B * C * P == 0) return MGAR_OK;
    MGAR_REQUIRE(dy && x && mean && invstd && workspace && dx, "bn_act_bwd: null pointer");
    MGAR_REQUIRE(C <= 65535 && (long long)P <= 65535LL * BN_THREADS * ((P & 3) == 0 ? 4 : 1), "bn_act_bwd: C > 65535 or P too large");
    const int chunk = bn_chunk_bwd(C, (long long)B * P), nchunk = (int)(((long long)B * P + chunk - 1) / chunk);
    float *coef = workspace + (size_t)2 * C * nchunk;
    hipStream_t st = (hipStream_t)stream;
    { KtScope kt(KT_BN_BWD_REDUCE, st, 2.0 * sizeof(T) * (double)B * C * P);
    if (relu) hipLaunchKernelGGL((bn_bwd_partial_kernel<true, T>), dim3(nchunk, C), dim3(BN_THREADS), 0, st, dy, x, B, C, P, mean, invstd, gamma, beta, workspace, chunk);
    else hipLaunchKernelGGL((bn_bwd_partial_kernel<false, T>), dim3(nchunk, C), dim3(BN_THREADS), 0, st, dy, x, B, C, P, mean, invstd, gamma, beta, workspace, chunk);
    }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, st, workspace, nchunk, C, (double)B * P, dgamma, dbeta, coef);
    dim3 grid(B * C, ceil_div(P, BN_THREADS * ((P & 3) == 0 ? 4 : 1)));
    { KtScope kt(KT_BN_BWD_APPLY, st, 3.0 * sizeof(T) * (double)B * C * P);
    if (relu) hipLaunchKernelGGL((bn_bwd_apply_kernel<true, T>), grid, dim3(BN_THREADS), 0, st, dy, x, C, P, mean, invstd, gamma, beta, coef, dx);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<false, T>), grid, dim3(BN_THREADS), 0, st, dy, x, C, P, mean, invstd, gamma, beta, coef, dx);
    }
    return check_launch("bn_act_bwd: launch failed");
}

template <typename T>
static int bn_act_maxpool_bwd_impl(const T *dpool, const T *pooled, const unsigned char *arg, const T *x,
                                   const T *xarg, int B, int C, int M, int nsample, const float *mean, const float *invstd,
                                   const float *gamma, int relu, float *workspace, float *dgamma, float *dbeta, T *dx,
                                   void *stream, long long dp_bs = -1, long long dp_cs = -1, long long dp_ms = 1) {
    MGAR_REQUIRE(bn_sizes_ok(B, C, (long long)M * nsample) && nsample >= 1 && nsample <= 255, "bn_act_maxpool_bwd: bad sizes");
    const DpoolStrides ds{dp_bs < 0 ? (long long)C * M : dp_bs, dp_cs < 0 ? (long long)M : dp_cs, dp_ms};
    MGAR_REQUIRE(ds.b >= 0 && ds.c >= 1 && ds.m >= 1, "bn_act_maxpool_bwd: bad dpool strides");
    if ((long long)B * C * M == 0) return MGAR_OK;
    MGAR_REQUIRE(dpool && pooled && arg && x && mean && invstd && workspace && dx, "bn_act_maxpool_bwd: null pointer");
    // (the apply kernel's second grid dimension counts 256-thread blocks of 4 elements when nsample % 4 == 0, of 1 otherwise)
    MGAR_REQUIRE(C <= 65535 && (long long)M * nsample <= 65535LL * BN_THREADS * ((nsample & 3) == 0 ? 4 : 1),
                 "bn_act_maxpool_bwd: C > 65535 or M*nsample too large");
    const int chunk = bn_chunk_bwd(C, (long long)B * M), nchunk = (int)(((long long)B * M + chunk - 1) / chunk);
    float *coef = workspace + (size_t)2 * C * nchunk;
    hipStream_t st = (hipStream_t)stream;
    { KtScope kt(KT_BN_MAX_BWD_REDUCE, st, (3.0 * sizeof(T) + 1.0) * (double)B * C * M);
    if (relu) hipLaunchKernelGGL((bn_max_bwd_partial_kernel<true, T>), dim3(nchunk, C), dim3(BN_THREADS), 0, st, dpool, pooled, arg, x, xarg, B, C, M, nsample, mean, invstd, workspace, ds, chunk);
    else hipLaunchKernelGGL((bn_max_bwd_partial_kernel<false, T>), dim3(nchunk, C), dim3(BN_THREADS), 0, st, dpool, pooled, arg, x, xarg, B, C, M, nsample, mean, invstd, workspace, ds, chunk);
    }
    // the means are over ALL B*M*nsample elements of the channel, not only the arg-max ones
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, st, workspace, nchunk, C, (double)B * M * nsample, dgamma, dbeta, coef);
    dim3 grid(B * C, ceil_div((long long)M * nsample, BN_THREADS * ((nsample & 3) == 0 ? 4 : 1)));
    { KtScope kt(KT_BN_MAX_BWD_APPLY, st, (double)B * C * M * (2.0 * sizeof(T) * nsample + 2.0 * sizeof(T) + 1.0));
    if (relu) hipLaunchKernelGGL((bn_max_bwd_apply_kernel<true, T>), grid, dim3(BN_THREADS), 0, st, dpool, pooled, arg, x, C, M, nsample, mean, invstd, gamma, coef, dx, ds);
    else hipLaunchKernelGGL((bn_max_bwd_apply_kernel<false, T>), grid, dim3(BN_THREADS), 0, st, dpool, pooled, arg, x, C, M, nsample, mean, invstd, gamma, coef, dx, ds);
    }
    return check_launch("bn_act_maxpool_bwd: launch failed");
}

// ---- C ABI: fp32 payload (include/mgar_ops.h) and the bf16-payload twins (suffix _bf16; same arguments, the payload
// pointers -- x, y, out, xarg, dy, dx, dpool, pooled -- address bf16 elements) ----
#define BN_BOTH(NAME, PARAMS_F, PARAMS_B, CALL_F, CALL_B)      \
    BN_API int NAME PARAMS_F { return CALL_F; }                  \
    BN_API int NAME##_bf16 PARAMS_B { return CALL_B; }
typedef const bf16_t *cbf;
typedef bf16_t *mbf;

BN_BOTH(mgar_bn_train_stats,
        (const float *x, int B, int C, int P, float eps, float momentum, float *workspace, float *mean, float *invstd,
         float *running_mean, float *running_var, long long *num_batches_tracked, void *stream),
        (const void *x, int B, int C, int P, float eps, float momentum, float *workspace, float *mean, float *invstd,
         float *running_mean, float *running_var, long long *num_batches_tracked, void *stream),
        bn_train_stats_impl<float>(x, B, C, P, eps, momentum, workspace, mean, invstd, running_mean, running_var, num_batches_tracked, stream),
        bn_train_stats_impl<bf16_t>((cbf)x, B, C, P, eps, momentum, workspace, mean, invstd, running_mean, running_var, num_batches_tracked, stream))
BN_BOTH(mgar_bn_train_stats_grouped,
        (const float *x, int G, int C, int P, float eps, float momentum, float *workspace, float *mean, float *invstd,
         float *running_mean, float *running_var, long long *num_batches_tracked, void *stream),
        (const void *x, int G, int C, int P, float eps, float momentum, float *workspace, float *mean, float *invstd,
         float *running_mean, float *running_var, long long *num_batches_tracked, void *stream),
        bn_train_stats_grouped_impl<float>(x, G, C, P, eps, momentum, workspace, mean, invstd, running_mean, running_var, num_batches_tracked, stream),
        bn_train_stats_grouped_impl<bf16_t>((cbf)x, G, C, P, eps, momentum, workspace, mean, invstd, running_mean, running_var, num_batches_tracked, stream))
BN_BOTH(mgar_bn_act_fwd,
        (const float *x, int B, int C, int P, const float *mean, const float *invstd, const float *gamma, const float *beta,
         int relu, float *y, void *stream),
        (const void *x, int B, int C, int P, const float *mean, const float *invstd, const float *gamma, const float *beta,
         int relu, void *y, void *stream),
        bn_act_fwd_impl<float>(x, B, C, P, mean, invstd, gamma, beta, relu, y, 0, stream),
        bn_act_fwd_impl<bf16_t>((cbf)x, B, C, P, mean, invstd, gamma, beta, relu, (mbf)y, 0, stream))
BN_BOTH(mgar_bn_act_fwd_grouped,
        (const float *x, int G, int C, int P, const float *mean, const float *invstd, const float *gamma, const float *beta,
         int relu, float *y, void *stream),
        (const void *x, int G, int C, int P, const float *mean, const float *invstd, const float *gamma, const float *beta,
         int relu, void *y, void *stream),
        bn_act_fwd_impl<float>(x, G, C, P, mean, invstd, gamma, beta, relu, y, 1, stream),
        bn_act_fwd_impl<bf16_t>((cbf)x, G, C, P, mean, invstd, gamma, beta, relu, (mbf)y, 1, stream))
// Statistics + apply in one launch for small channels (B * P, or P with per-sample statistics, <= 16 384; P % 4 == 0): what
// mgar_bn_train_stats[_grouped] followed by mgar_bn_act_fwd_into computes, forward only.  mean / invstd (rows) optional
// outputs; workspace: rows floats (per-sample statistics only: the biased variances for the running update).
template <typename T>
static int bn_act_small_impl(const T *x, int B, int C, int P, int per_sample, float eps, float momentum, const float *gamma,
                             const float *beta, int relu, float *workspace, float *mean, float *invstd, float *running_mean,
                             float *running_var, long long *num_batches_tracked, T *y, long long y_bstride, void *stream) {
    MGAR_REQUIRE(bn_sizes_ok(B, C, P), "bn_act_small: bad sizes");
    if ((long long)B * C * P == 0) return MGAR_OK;
    const long long n = per_sample ? P : (long long)B * P;
    if (n > BN_SMALL_MAX || (P & 3) != 0) {
        set_error("bn_act_small: needs at most 16384 elements per channel and P % 4 == 0");
        return MGAR_EUNSUPPORTED;
    }
    MGAR_REQUIRE(x && y, "bn_act_small: null pointer");
    if (y_bstride < 0) y_bstride = (long long)C * P;
    MGAR_REQUIRE(y_bstride >= (long long)C * P && (y_bstride * (long long)sizeof(T)) % 16 == 0 && (uintptr_t)y % 16 == 0,
                 "bn_act_small: bad output slice");
    const int rows = per_sample ? B * C : C;
    const bool track = running_mean || running_var || num_batches_tracked;
    MGAR_REQUIRE(!(per_sample && track) || (workspace && mean), "bn_act_small: per-sample running update needs workspace and mean");
    hipStream_t st = (hipStream_t)stream;
    float *var = per_sample && track ? workspace : nullptr;
    if (relu) hipLaunchKernelGGL((bn_small_fused_kernel<true, T>), dim3(rows), dim3(BN_THREADS), 0, st, x, B, C, P, per_sample, eps, momentum, gamma, beta, mean, invstd, var, running_mean, running_var, num_batches_tracked, y, y_bstride);
    else hipLaunchKernelGGL((bn_small_fused_kernel<false, T>), dim3(rows), dim3(BN_THREADS), 0, st, x, B, C, P, per_sample, eps, momentum, gamma, beta, mean, invstd, var, running_mean, running_var, num_batches_tracked, y, y_bstride);
    if (per_sample && track)
        hipLaunchKernelGGL(bn_running_update_grouped_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, st, mean, var, B, C, (double)P, momentum,
                           running_mean, running_var, num_batches_tracked);
    return check_launch("bn_act_small: launch failed");
}
BN_BOTH(mgar_bn_act_small,
        (const float *x, int B, int C, int P, int per_sample, float eps, float momentum, const float *gamma, const float *beta, int relu,
         float *workspace, float *mean, float *invstd, float *running_mean, float *running_var, long long *num_batches_tracked,
         float *y, long long y_bstride, void *stream),
        (const void *x, int B, int C, int P, int per_sample, float eps, float momentum, const float *gamma, const float *beta, int relu,
         float *workspace, float *mean, float *invstd, float *running_mean, float *running_var, long long *num_batches_tracked,
         void *y, long long y_bstride, void *stream),
        bn_act_small_impl<float>(x, B, C, P, per_sample, eps, momentum, gamma, beta, relu, workspace, mean, invstd, running_mean, running_var, num_batches_tracked, y, y_bstride, stream),
        bn_act_small_impl<bf16_t>((cbf)x, B, C, P, per_sample, eps, momentum, gamma, beta, relu, workspace, mean, invstd, running_mean, running_var, num_batches_tracked, (mbf)y, y_bstride, stream))
// the same with y a CHANNEL SLICE of a wider (B, C_total, P) tensor: sample b of the result starts y_bstride elements after
// sample b - 1 (an Inception module's branches write straight into the concatenated output: no torch.cat pass)
BN_BOTH(mgar_bn_act_fwd_into,
        (const float *x, int B, int C, int P, const float *mean, const float *invstd, const float *gamma, const float *beta,
         int relu, int stats_per_sample, float *y, long long y_bstride, void *stream),
        (const void *x, int B, int C, int P, const float *mean, const float *invstd, const float *gamma, const float *beta,
         int relu, int stats_per_sample, void *y, long long y_bstride, void *stream),
        bn_act_fwd_impl<float>(x, B, C, P, mean, invstd, gamma, beta, relu, y, stats_per_sample != 0, stream, y_bstride),
        bn_act_fwd_impl<bf16_t>((cbf)x, B, C, P, mean, invstd, gamma, beta, relu, (mbf)y, stats_per_sample != 0, stream, y_bstride))
BN_BOTH(mgar_bn_act_maxpool_fwd,
        (const float *x, int B, int C, int M, int nsample, const float *mean, const float *invstd, const float *gamma,
         const float *beta, int relu, float *out, unsigned char *arg, float *xarg, void *stream),
        (const void *x, int B, int C, int M, int nsample, const float *mean, const float *invstd, const float *gamma,
         const float *beta, int relu, void *out, unsigned char *arg, void *xarg, void *stream),
        bn_act_maxpool_fwd_impl<float>(x, B, C, M, nsample, mean, invstd, gamma, beta, relu, out, arg, xarg, stream),
        bn_act_maxpool_fwd_impl<bf16_t>((cbf)x, B, C, M, nsample, mean, invstd, gamma, beta, relu, (mbf)out, arg, (mbf)xarg, stream))
BN_BOTH(mgar_bn_act_bwd,
        (const float *dy, const float *x, int B, int C, int P, const float *mean, const float *invstd, const float *gamma,
         const float *beta, int relu, float *workspace, float *dgamma, float *dbeta, float *dx, void *stream),
        (const void *dy, const void *x, int B, int C, int P, const float *mean, const float *invstd, const float *gamma,
         const float *beta, int relu, float *workspace, float *dgamma, float *dbeta, void *dx, void *stream),
        bn_act_bwd_impl<float>(dy, x, B, C, P, mean, invstd, gamma, beta, relu, workspace, dgamma, dbeta, dx, stream),
        bn_act_bwd_impl<bf16_t>((cbf)dy, (cbf)x, B, C, P, mean, invstd, gamma, beta, relu, workspace, dgamma, dbeta, (mbf)dx, stream))
BN_BOTH(mgar_bn_act_maxpool_bwd,
        (const float *dpool, const float *pooled, const unsigned char *arg, const float *x, const float *xarg, int B, int C,
         int M, int nsample, const float *mean, const float *invstd, const float *gamma, int relu, float *workspace,
         float *dgamma, float *dbeta, float *dx, void *stream),
        (const void *dpool, const void *pooled, const unsigned char *arg, const void *x, const void *xarg, int B, int C,
         int M, int nsample, const float *mean, const float *invstd, const float *gamma, int relu, float *workspace,
         float *dgamma, float *dbeta, void *dx, void *stream),
        bn_act_maxpool_bwd_impl<float>(dpool, pooled, arg, x, xarg, B, C, M, nsample, mean, invstd, gamma, relu, workspace, dgamma, dbeta, dx, stream),
        bn_act_maxpool_bwd_impl<bf16_t>((cbf)dpool, (cbf)pooled, arg, (cbf)x, (cbf)xarg, B, C, M, nsample, mean, invstd, gamma, relu, workspace, dgamma, dbeta, (mbf)dx, stream))

// bn_act_bwd with the input gradient written ROW-MAJOR: dx_t (B*P, C) instead of dx (B, C, P).  fp32, C <= 64.
BN_API int mgar_bn_act_bwd_rowmajor(const float *dy, const float *x, int B, int C, int P, const float *mean, const float *invstd,
                                    const float *gamma, const float *beta, int relu, float *workspace, float *dgamma, float *dbeta,
                                    float *dx_t, void *stream) {
    MGAR_REQUIRE(bn_sizes_ok(B, C, P), "bn_act_bwd_rowmajor: bad sizes");
    if (C > 64) {
        set_error("bn_act_bwd_rowmajor: C <= 64");
        return MGAR_EUNSUPPORTED;
    }
    if ((long long)B * C * P == 0) return MGAR_OK;
    MGAR_REQUIRE(dy && x && mean && invstd && workspace && dx_t, "bn_act_bwd_rowmajor: null pointer");
    MGAR_REQUIRE(B <= 65535, "bn_act_bwd_rowmajor: B > 65535");
    const int chunk = bn_chunk_bwd(C, (long long)B * P), nchunk = (int)(((long long)B * P + chunk - 1) / chunk);
    float *coef = workspace + (size_t)2 * C * nchunk;
    hipStream_t st = (hipStream_t)stream;
    { KtScope kt(KT_BN_BWD_REDUCE, st, 8.0 * (double)B * C * P);
    if (relu) hipLaunchKernelGGL((bn_bwd_partial_kernel<true, float>), dim3(nchunk, C), dim3(BN_THREADS), 0, st, dy, x, B, C, P, mean, invstd, gamma, beta, workspace, chunk);
    else hipLaunchKernelGGL((bn_bwd_partial_kernel<false, float>), dim3(nchunk, C), dim3(BN_THREADS), 0, st, dy, x, B, C, P, mean, invstd, gamma, beta, workspace, chunk);
    }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, st, workspace, nchunk, C, (double)B * P, dgamma, dbeta, coef);
    dim3 grid(ceil_div(P, 64), B);
    { KtScope kt(KT_BN_BWD_APPLY, st, 12.0 * (double)B * C * P);
    if (relu) hipLaunchKernelGGL(bn_bwd_apply_t_kernel<true>, grid, dim3(BN_THREADS), 0, st, dy, x, C, P, mean, invstd, gamma, beta, coef, dx_t);
    else hipLaunchKernelGGL(bn_bwd_apply_t_kernel<false>, grid, dim3(BN_THREADS), 0, st, dy, x, C, P, mean, invstd, gamma, beta, coef, dx_t);
    }
    return check_launch("bn_act_bwd_rowmajor: launch failed");
}

// bn_act_maxpool_bwd with dpool read IN PLACE from a strided tensor: element (b, c, m) at dpool[b * sb + c * sc + m * sm].
// Covers the two layouts the pooled gradient arrives in: a channel slice of a wider (B, C_total, M) tensor (the gradient
// of the torch.cat over the scales of an SA module) and the transposed view of (M, C_total) rows (the RoI-grid lift's
// consumer works on rows).  fp32.
BN_API int mgar_bn_act_maxpool_bwd_strided(const float *dpool, long long sb, long long sc, long long sm, const float *pooled,
                                           const unsigned char *arg, const float *x, const float *xarg, int B, int C, int M,
                                           int nsample, const float *mean, const float *invstd, const float *gamma, int relu,
                                           float *workspace, float *dgamma, float *dbeta, float *dx, void *stream) {
    MGAR_REQUIRE(sb >= 0 && sc >= 1 && sm >= 1, "bn_act_maxpool_bwd_strided: bad strides");
    return bn_act_maxpool_bwd_impl<float>(dpool, pooled, arg, x, xarg, B, C, M, nsample, mean, invstd, gamma, relu, workspace, dgamma,
                                          dbeta, dx, stream, sb, sc, sm);
}

// Merge groups of G consecutive chunk partials (mean, M2) of a channel into one partial each (Chan et al., in double): the
// output has the same format with chunk' = G * chunk.  grid (ceil(nchunk / G), C), one wave; G = 64 * BN_MERGE_PER_LANE.
constexpr int BN_MERGE_PER_LANE = 4, BN_MERGE_G = 64 * BN_MERGE_PER_LANE;
__global__ __launch_bounds__(64) void bn_merge_partials_kernel(const float *__restrict__ in, int nchunk, double n, int chunk,
                                                              float *__restrict__ out) {
    const int c = blockIdx.y, g = blockIdx.x, ngroup = gridDim.x;
    const int i0 = g * BN_MERGE_G, i1 = min(i0 + BN_MERGE_G, nchunk);
    auto count = [&](int i) { return i + 1 < nchunk ? (double)chunk : n - (double)chunk * (nchunk - 1); };
    float2 v[BN_MERGE_PER_LANE];
    double cnt[BN_MERGE_PER_LANE], s = 0.0, tot = 0.0;
#pragma unroll
    for (int k = 0; k < BN_MERGE_PER_LANE; ++k) {
        const int i = i0 + k * 64 + (int)threadIdx.x;
        const bool ok = i < i1;
        v[k] = ok ? *reinterpret_cast<const float2 *>(in + ((size_t)c * nchunk + i) * 2) : make_float2(0.f, 0.f);
        cnt[k] = ok ? count(i) : 0.0;
        s += cnt[k] * (double)v[k].x;
        tot += cnt[k];
    }
    s = wave_sum_f64(s);
    tot = wave_sum_f64(tot);
    const double m = s / tot;
    double q = 0.0;
#pragma unroll
    for (int k = 0; k < BN_MERGE_PER_LANE; ++k) {
        const double d = (double)v[k].x - m;
        q += (double)v[k].y + cnt[k] * d * d;
    }
    q = wave_sum_f64(q);
    if (threadIdx.x == 0) *reinterpret_cast<float2 *>(out + ((size_t)c * ngroup + g) * 2) = make_float2((float)m, (float)q);
}

// BatchNorm training statistics from partials a PRODUCER kernel left behind (mgar_pointwise_conv_fwd_stats,
// mgar_query_group_proj_stack_fwd_stats): partial (C, nchunk, 2) = per (channel, chunk of `chunk` consecutive elements of the
// channel in (b, p) order) the chunk's mean and sum of squared deviations; n = B * P elements per channel = nchunk * chunk.
// Same finalize (Chan merge in double, running statistics) as mgar_bn_train_stats, without its pass over x.
// workspace: mgar_bn_stats_from_partials_workspace_floats(nchunk, C) floats (many partials are first merged in groups).
BN_API long long mgar_bn_stats_from_partials_workspace_floats(int nchunk, int C) {
    if (nchunk < 0 || C < 0) return -1;
    return nchunk > 4 * BN_MERGE_G ? 2ll * C * ((nchunk + BN_MERGE_G - 1) / BN_MERGE_G) : 0;
}
BN_API int mgar_bn_stats_from_partials(const float *partial, int nchunk, int C, long long n, int chunk, float eps, float momentum,
                                       float *workspace, float *mean, float *invstd, float *running_mean, float *running_var,
                                       long long *num_batches_tracked, void *stream) {
    MGAR_REQUIRE(nchunk >= 0 && C >= 0 && n >= 0 && chunk >= 1, "bn_stats_from_partials: bad sizes");
    if (C == 0 || n == 0) return MGAR_OK;
    MGAR_REQUIRE(partial && mean && invstd, "bn_stats_from_partials: null pointer");
    MGAR_REQUIRE((long long)nchunk * chunk >= n && (long long)(nchunk - 1) * chunk < n, "bn_stats_from_partials: nchunk * chunk does not cover n");
    MGAR_REQUIRE(C <= 65535, "bn_stats_from_partials: C > 65535");
    if (nchunk > 4 * BN_MERGE_G) {   // one wave per channel would walk them all: merge groups of 256 first (same partial format)
        MGAR_REQUIRE(workspace, "bn_stats_from_partials: null workspace");
        MGAR_REQUIRE((long long)chunk * BN_MERGE_G <= 2147483647LL, "bn_stats_from_partials: chunk too large");
        const int ngroup = (nchunk + BN_MERGE_G - 1) / BN_MERGE_G;
        hipLaunchKernelGGL(bn_merge_partials_kernel, dim3(ngroup, C), dim3(64), 0, (hipStream_t)stream, partial, nchunk, (double)n, chunk,
                           workspace);
        partial = workspace;
        nchunk = ngroup;
        chunk *= BN_MERGE_G;
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, partial, nchunk, C, (double)n, chunk, eps, momentum,
                       mean, invstd, running_mean, running_var, num_batches_tracked, (float *)nullptr);
    return check_launch("bn_stats_from_partials: launch failed");
}

// BatchNorm [+ ReLU] backward, APPLY ONLY: the reduction {mean dz, mean dz xhat} per channel comes in `coef` (2 C floats) from
// the kernel that already had the operands in hand (mgar_pointwise_conv_dw_bnbwd).  rowmajor != 0: dx_t (B * P, C), C <= 64.
BN_API int mgar_bn_act_bwd_apply(const float *dy, const float *x, int B, int C, int P, const float *mean, const float *invstd,
                                 const float *gamma, const float *beta, int relu, const float *coef, int rowmajor, float *dx,
                                 void *stream) {
    MGAR_REQUIRE(bn_sizes_ok(B, C, P), "bn_act_bwd_apply: bad sizes");
    if ((long long)B * C * P == 0) return MGAR_OK;
    MGAR_REQUIRE(dy && x && mean && invstd && coef && dx, "bn_act_bwd_apply: null pointer");
    hipStream_t st = (hipStream_t)stream;
    KtScope kt(KT_BN_BWD_APPLY, st, 12.0 * (double)B * C * P);
    if (rowmajor) {
        if (C > 64) {
            set_error("bn_act_bwd_apply: rowmajor needs C <= 64");
            return MGAR_EUNSUPPORTED;
        }
        MGAR_REQUIRE(B <= 65535, "bn_act_bwd_apply: B > 65535");
        dim3 grid(ceil_div(P, 64), B);
        if (relu) hipLaunchKernelGGL(bn_bwd_apply_t_kernel<true>, grid, dim3(BN_THREADS), 0, st, dy, x, C, P, mean, invstd, gamma, beta, coef, dx);
        else hipLaunchKernelGGL(bn_bwd_apply_t_kernel<false>, grid, dim3(BN_THREADS), 0, st, dy, x, C, P, mean, invstd, gamma, beta, coef, dx);
    } else {
        MGAR_REQUIRE(C <= 65535 && (long long)P <= 65535LL * BN_THREADS * ((P & 3) == 0 ? 4 : 1), "bn_act_bwd_apply: C > 65535 or P too large");
        dim3 grid(B * C, ceil_div(P, BN_THREADS * ((P & 3) == 0 ? 4 : 1)));
        if (relu) hipLaunchKernelGGL((bn_bwd_apply_kernel<true, float>), grid, dim3(BN_THREADS), 0, st, dy, x, C, P, mean, invstd, gamma, beta, coef, dx);
        else hipLaunchKernelGGL((bn_bwd_apply_kernel<false, float>), grid, dim3(BN_THREADS), 0, st, dy, x, C, P, mean, invstd, gamma, beta, coef, dx);
    }
    return check_launch("bn_act_bwd_apply: launch failed");
}

// the channels-last (NDHWC) variants of the forward kernels, for the frozen I3D between its convolutions
#include "channels_last.hpp"
