// channels_last.hpp -- included at the end of bn_act.hip (same translation unit: it launches bn_finalize_kernel) -- the frozen I3D's BatchNorm(train statistics) + ReLU and "same" max-pooling on NDHWC (channels-last)
// activations, forward only, gfx950.
//
// Why: MIOpen runs the I3D's 3-D convolutions on composable-kernel kernels that work in NDHWC; on NCDHW tensors it wraps every
// one of them in two `batched_transpose` launches (46 launches, 4.97 ms per step at config c3, on the critical path of the
// step; the 64 -> 192 3x3x3 convolution alone: 22.98 ms NCDHW vs 20.04 ms NDHWC, tools/probe_ndhwc.py).  With the activations
// kept channels-last between the stem and the RoI crop, the convolutions need no adapter -- provided the kernels of THIS
// library that sit between them (reference model/backbone.py:61-131, 227-260: Unit3D's BatchNorm3d + ReLU,
// MaxPool3dSamePadding, the Inception concatenation) read and write that layout too.  Same arithmetic as csrc/bn_act.hip /
// csrc/maxpool3d.hip; only the indexing differs: a tensor is (rows = N*T*H*W, C) with the channel innermost.
//
//   bn_cl_partial_kernel   per (sample | all) and channel: chunk (mean, M2) partials in the format of bn_finalize_kernel
//   bn_cl_apply_kernel     y = relu?((x - mean) * invstd * gamma + beta); y may be a COLUMN SLICE of a wider (rows, ldy) tensor
//                          (the Inception concatenation: no torch.cat pass)
//   bn_to_cl_apply_kernel  the same, reading NCDHW and writing NDHWC (the stem's BatchNorm is where the layout changes)
//   maxpool3d_cl_kernel    TF-"same" zero-padded max pooling, one thread per (output position, 4 channels)
namespace mgar {

constexpr int CL_THREADS = 256;

// ---- statistics ----------------------------------------------------------------------------------------------------------
// x (S * R rows, C), C % 4 == 0, C <= 1024.  grid (nchunk, S): chunk k of sample s = rows [k * chunk, ...) of that sample.
// Thread (j = float4 channel group, rr = row lane): C / 4 groups, 256 / (C / 4) rows in flight.  Cancellation-safe like
// bn_partial_kernel: sums of (x - pivot), pivot = the chunk's first row.  partial[((s * C + c) * nchunk + k) * 2 + {0, 1}] =
// chunk mean, chunk M2 (bn_finalize_kernel with S * C "channels", n = R, chunk rows per chunk).
template <typename T>
__global__ __launch_bounds__(CL_THREADS) void bn_cl_partial_kernel(const T *__restrict__ x, int R, int C, int chunk,
                                                                   float *__restrict__ partial) {
    __shared__ float4 red_s[CL_THREADS], red_q[CL_THREADS];
    const int ng = C >> 2, rpar = CL_THREADS / ng;
    const int j = threadIdx.x % ng, rr = threadIdx.x / ng;
    const int s = blockIdx.y, k = blockIdx.x, nchunk = gridDim.x;
    const int r0 = k * chunk, r1 = min(r0 + chunk, R);
    const T *base = x + ((size_t)s * R) * C + 4 * j;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f), sq = sum, piv = sum;
    if (rr < rpar) {
        piv = Payload<T>::ld4(base + (size_t)r0 * C);
        for (int r = r0 + rr; r < r1; r += rpar) {
            float4 v = Payload<T>::ld4(base + (size_t)r * C);
            v.x -= piv.x; v.y -= piv.y; v.z -= piv.z; v.w -= piv.w;
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            sq.x += v.x * v.x; sq.y += v.y * v.y; sq.z += v.z * v.z; sq.w += v.w * v.w;
        }
    }
    red_s[threadIdx.x] = sum;
    red_q[threadIdx.x] = sq;
    __syncthreads();
    if (threadIdx.x < ng) {                                  // rr == 0 threads: add the row lanes in order
        for (int q = 1; q < rpar; ++q) {
            const float4 a = red_s[q * ng + j], b = red_q[q * ng + j];
            sum.x += a.x; sum.y += a.y; sum.z += a.z; sum.w += a.w;
            sq.x += b.x; sq.y += b.y; sq.z += b.z; sq.w += b.w;
        }
        const double nk = (double)(r1 - r0);
        const float sv[4] = {sum.x, sum.y, sum.z, sum.w}, qv[4] = {sq.x, sq.y, sq.z, sq.w}, pv[4] = {piv.x, piv.y, piv.z, piv.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double sd = (double)sv[i];
            double m2 = (double)qv[i] - sd * sd / nk;
            if (m2 < 0.0) m2 = 0.0;
            float *dst = partial + (((size_t)s * C + 4 * j + i) * nchunk + k) * 2;
            dst[0] = (float)((double)pv[i] + sd / nk);
            dst[1] = (float)m2;
        }
    }
}

// ---- apply -----------------------------------------------------------------------------------------------------------------
// statistics index = (per_sample ? sample * C : 0) + c.  y[row * ldy + c] (ldy >= C: y points at the first column of the slice)
template <bool RELU, typename T>
__global__ __launch_bounds__(CL_THREADS) void bn_cl_apply_kernel(const T *__restrict__ x, long long rows, int R, int C, int per_sample,
                                                                 const float *__restrict__ mean, const float *__restrict__ invstd,
                                                                 const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                 T *__restrict__ y, int ldy) {
    const int ng = C >> 2;
    const long long total = rows * ng;
    for (long long e = (long long)blockIdx.x * CL_THREADS + threadIdx.x; e < total; e += (long long)gridDim.x * CL_THREADS) {
        const long long row = e / ng;
        const int c = (int)(e - row * ng) * 4;
        const int si = (per_sample ? (int)(row / R) * C : 0) + c;
        const float4 mu = *reinterpret_cast<const float4 *>(mean + si), is = *reinterpret_cast<const float4 *>(invstd + si);
        const float4 g = gamma ? *reinterpret_cast<const float4 *>(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 b = beta ? *reinterpret_cast<const float4 *>(beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 v = Payload<T>::ld4(x + row * C + c);
        v.x = (v.x - mu.x) * (is.x * g.x) + b.x; v.y = (v.y - mu.y) * (is.y * g.y) + b.y;
        v.z = (v.z - mu.z) * (is.z * g.z) + b.z; v.w = (v.w - mu.w) * (is.w * g.w) + b.w;
        if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        Payload<T>::st4(y + row * ldy + c, v);
    }
}

// NCDHW in (S, C, R), NDHWC out (S * R rows, ldy): a workgroup transposes a 64-position x 64-channel tile through LDS.
// grid (ceil(R / 64), ceil(C / 64), S)
template <bool RELU, typename T>
__global__ __launch_bounds__(CL_THREADS) void bn_to_cl_apply_kernel(const T *__restrict__ x, int C, int R, int per_sample,
                                                                    const float *__restrict__ mean, const float *__restrict__ invstd,
                                                                    const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                    T *__restrict__ y, int ldy) {
    __shared__ float tile[64][65];
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64, s = blockIdx.z;
    for (int e = threadIdx.x; e < 64 * 64; e += CL_THREADS) {       // lanes along the positions: coalesced NCDHW reads
        const int cl = e >> 6, pl = e & 63, c = c0 + cl, p = p0 + pl;
        float v = 0.f;
        if (c < C && p < R) {
            const int si = (per_sample ? s * C : 0) + c;
            v = (Payload<T>::ld(x + ((size_t)s * C + c) * R + p) - mean[si]) * (invstd[si] * (gamma ? gamma[c] : 1.f)) + (beta ? beta[c] : 0.f);
            if (RELU) v = fmaxf(v, 0.f);
        }
        tile[pl][cl] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 64; e += CL_THREADS) {       // lanes along the channels: coalesced NDHWC writes
        const int pl = e >> 6, cl = e & 63, c = c0 + cl, p = p0 + pl;
        if (c < C && p < R) Payload<T>::st(y + ((size_t)s * R + p) * ldy + c, tile[pl][cl]);
    }
}

// ---- max pooling, TF "same" zero padding (reference model/backbone.py:99-131), NDHWC -----------------------------------------
struct PoolCl {
    int T, H, W, To, Ho, Wo, C;
    int kt, kh, kw, st, sh, sw, pt, ph, pw;   // p* = FRONT padding
};
template <typename T>
__global__ __launch_bounds__(CL_THREADS) void maxpool3d_cl_kernel(const T *__restrict__ x, long long total, PoolCl g, T *__restrict__ y) {
    const int ng = g.C >> 2;
    for (long long e = (long long)blockIdx.x * CL_THREADS + threadIdx.x; e < total; e += (long long)gridDim.x * CL_THREADS) {
        const int c = (int)(e % ng) * 4;
        long long r = e / ng;
        const int wo = (int)(r % g.Wo); r /= g.Wo;
        const int ho = (int)(r % g.Ho); r /= g.Ho;
        const int to = (int)(r % g.To);
        const long long n = r / g.To;
        const int t0 = to * g.st - g.pt, h0 = ho * g.sh - g.ph, w0 = wo * g.sw - g.pw;
        const int t1 = t0 + g.kt, h1 = h0 + g.kh, w1 = w0 + g.kw;
        const bool pad = t0 < 0 || h0 < 0 || w0 < 0 || t1 > g.T || h1 > g.H || w1 > g.W;
        const float init = pad ? 0.f : -__builtin_inff();             // the zero padding takes part in the max
        float4 best = make_float4(init, init, init, init);
        const T *base = x + (size_t)n * g.T * g.H * g.W * g.C + c;
        for (int t = max(t0, 0); t < min(t1, g.T); ++t)
            for (int h = max(h0, 0); h < min(h1, g.H); ++h)
                for (int w = max(w0, 0); w < min(w1, g.W); ++w) {
                    const float4 v = Payload<T>::ld4(base + (((size_t)t * g.H + h) * g.W + w) * g.C);
                    best.x = fmaxf(best.x, v.x); best.y = fmaxf(best.y, v.y); best.z = fmaxf(best.z, v.z); best.w = fmaxf(best.w, v.w);
                }
        Payload<T>::st4(y + (((size_t)(n * g.To + to) * g.Ho + ho) * g.Wo + wo) * g.C + c, best);
    }
}

// rows per statistics chunk: at least 256, and ~2048 workgroups in all
static inline int cl_chunk_rows(int S, int R) {
    long long want = ((long long)S * R + 2047) / 2048;
    if (want < 256) want = 256;
    if (want > R) want = R;
    return (int)want;
}

// ---- backward for ROW-MAJOR activations (round 3: the sparse trunk's (N_active, C) features) ------------------------------
// BatchNorm1d (train statistics) [+ ReLU] over the rows of x (rows, C): the same two passes as csrc/bn_act.hip, indexed like the
// forward kernels above.  partial[(c * nchunk + k) * 2 + {0, 1}] = sum dz, sum dz * xhat over chunk k (bn_bwd_finalize_kernel's
// format); dz = dy * [relu active], the mask being the forward's expression bit for bit.
template <bool RELU>
__global__ __launch_bounds__(CL_THREADS) void bn_rows_bwd_partial_kernel(const float *__restrict__ dy, const float *__restrict__ x, int R, int C,
                                                                         int chunk, const float *__restrict__ mean,
                                                                         const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                                         const float *__restrict__ beta, float *__restrict__ partial) {
    __shared__ float4 red_s[CL_THREADS], red_q[CL_THREADS];
    const int ng = C >> 2, rpar = CL_THREADS / ng;
    const int j = threadIdx.x % ng, rr = threadIdx.x / ng;
    const int k = blockIdx.x, nchunk = gridDim.x;
    const int r0 = k * chunk, r1 = min(r0 + chunk, R);
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f), sq = sum;
    if (rr < rpar) {
        const int c = 4 * j;
        const float4 mu = *reinterpret_cast<const float4 *>(mean + c), is = *reinterpret_cast<const float4 *>(invstd + c);
        const float4 g = gamma ? *reinterpret_cast<const float4 *>(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 b = beta ? *reinterpret_cast<const float4 *>(beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = r0 + rr; r < r1; r += rpar) {
            const float4 xv = *reinterpret_cast<const float4 *>(x + (size_t)r * C + c);
            float4 d = *reinterpret_cast<const float4 *>(dy + (size_t)r * C + c);
            if (RELU) {
                if (!((xv.x - mu.x) * (is.x * g.x) + b.x > 0.f)) d.x = 0.f;
                if (!((xv.y - mu.y) * (is.y * g.y) + b.y > 0.f)) d.y = 0.f;
                if (!((xv.z - mu.z) * (is.z * g.z) + b.z > 0.f)) d.z = 0.f;
                if (!((xv.w - mu.w) * (is.w * g.w) + b.w > 0.f)) d.w = 0.f;
            }
            sum.x += d.x; sum.y += d.y; sum.z += d.z; sum.w += d.w;
            sq.x += d.x * ((xv.x - mu.x) * is.x); sq.y += d.y * ((xv.y - mu.y) * is.y);
            sq.z += d.z * ((xv.z - mu.z) * is.z); sq.w += d.w * ((xv.w - mu.w) * is.w);
        }
    }
    red_s[threadIdx.x] = sum;
    red_q[threadIdx.x] = sq;
    __syncthreads();
    if (threadIdx.x < ng) {                                  // rr == 0 threads: add the row lanes in order
        for (int q = 1; q < rpar; ++q) {
            const float4 a = red_s[q * ng + j], b = red_q[q * ng + j];
            sum.x += a.x; sum.y += a.y; sum.z += a.z; sum.w += a.w;
            sq.x += b.x; sq.y += b.y; sq.z += b.z; sq.w += b.w;
        }
        const float sv[4] = {sum.x, sum.y, sum.z, sum.w}, qv[4] = {sq.x, sq.y, sq.z, sq.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float *dst = partial + ((size_t)(4 * j + i) * nchunk + k) * 2;
            dst[0] = sv[i];
            dst[1] = qv[i];
        }
    }
}

// dx = gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat)); 8 B read + 4 B written per element
template <bool RELU>
__global__ __launch_bounds__(CL_THREADS) void bn_rows_bwd_apply_kernel(const float *__restrict__ dy, const float *__restrict__ x, long long rows,
                                                                       int C, const float *__restrict__ mean, const float *__restrict__ invstd,
                                                                       const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                       const float *__restrict__ coef, float *__restrict__ dx) {
    const int ng = C >> 2;
    const long long total = rows * ng;
    for (long long e = (long long)blockIdx.x * CL_THREADS + threadIdx.x; e < total; e += (long long)gridDim.x * CL_THREADS) {
        const long long row = e / ng;
        const int c = (int)(e - row * ng) * 4;
        const float4 mu = *reinterpret_cast<const float4 *>(mean + c), is = *reinterpret_cast<const float4 *>(invstd + c);
        const float4 g = gamma ? *reinterpret_cast<const float4 *>(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 b = beta ? *reinterpret_cast<const float4 *>(beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 xv = *reinterpret_cast<const float4 *>(x + row * C + c);
        float4 d = *reinterpret_cast<const float4 *>(dy + row * C + c);
        float4 r;
        auto one = [&](float xs, float ds, float m, float i, float gg, float bb, float m0, float m1) {
            const float k = i * gg;
            if (RELU && !((xs - m) * k + bb > 0.f)) ds = 0.f;
            return k * (ds - m0 - (xs - m) * i * m1);
        };
        r.x = one(xv.x, d.x, mu.x, is.x, g.x, b.x, coef[2 * c + 0], coef[2 * c + 1]);
        r.y = one(xv.y, d.y, mu.y, is.y, g.y, b.y, coef[2 * c + 2], coef[2 * c + 3]);
        r.z = one(xv.z, d.z, mu.z, is.z, g.z, b.z, coef[2 * c + 4], coef[2 * c + 5]);
        r.w = one(xv.w, d.w, mu.w, is.w, g.w, b.w, coef[2 * c + 6], coef[2 * c + 7]);
        *reinterpret_cast<float4 *>(dx + row * C + c) = r;
    }
}

}  // namespace mgar

#define CL_API extern "C" __attribute__((visibility("default")))

static bool cl_shape_ok(int S, int R, int C) { return S >= 0 && R >= 0 && C >= 0 && C % 4 == 0 && C <= 1024; }

// floats of workspace for mgar_bn_cl_train_stats (partials + the per-sample variances)
CL_API long long mgar_bn_cl_workspace_floats(int S, int R, int C, int per_sample) {
    if (!cl_shape_ok(S, R, C) || S == 0 || R == 0) return 0;
    const int Ss = per_sample ? S : 1;
    const long long Rs = per_sample ? R : (long long)S * R;
    if (Rs > 2147483647LL) return -1;
    const int chunk = cl_chunk_rows(Ss, (int)Rs), nchunk = (int)((Rs + chunk - 1) / chunk);
    return 2ll * Ss * C * nchunk + (long long)Ss * C;
}

template <typename T>
static int bn_cl_train_stats_impl(const T *x, int S, int R, int C, int per_sample, float eps, float momentum, float *workspace,
                                  float *mean, float *invstd, float *running_mean, float *running_var,
                                  long long *num_batches_tracked, void *stream) {
    MGAR_REQUIRE(cl_shape_ok(S, R, C), "bn_cl_train_stats: needs C % 4 == 0, C <= 1024");
    if ((long long)S * R * C == 0) return MGAR_OK;
    MGAR_REQUIRE(x && workspace && mean && invstd, "bn_cl_train_stats: null pointer");
    const int Ss = per_sample ? S : 1;
    const long long Rl = per_sample ? R : (long long)S * R;
    MGAR_REQUIRE(Rl <= 2147483647LL && (long long)Ss * C <= 65535, "bn_cl_train_stats: too many rows or S * C > 65535");
    const int Rs = (int)Rl, chunk = cl_chunk_rows(Ss, Rs), nchunk = (Rs + chunk - 1) / chunk;
    hipStream_t st = (hipStream_t)stream;
    float *var = workspace + (size_t)2 * Ss * C * nchunk;
    {
        KtScope kt(KT_BN_STATS, st, (double)sizeof(T) * S * R * C);
        hipLaunchKernelGGL(bn_cl_partial_kernel<T>, dim3(nchunk, Ss), dim3(CL_THREADS), 0, st, x, Rs, C, chunk, workspace);
    }
    if (per_sample) {
        hipLaunchKernelGGL(bn_finalize_kernel, dim3(Ss * C), dim3(64), 0, st, workspace, nchunk, Ss * C, (double)Rs, chunk, eps, momentum,
                           mean, invstd, (float *)nullptr, (float *)nullptr, (long long *)nullptr, var);
        if (running_mean || running_var || num_batches_tracked)
            hipLaunchKernelGGL(bn_running_update_grouped_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, st, mean, var, S, C, (double)R,
                               momentum, running_mean, running_var, num_batches_tracked);
    } else {
        hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, st, workspace, nchunk, C, (double)Rs, chunk, eps, momentum, mean,
                           invstd, running_mean, running_var, num_batches_tracked, (float *)nullptr);
    }
    return check_launch("bn_cl_train_stats: launch failed");
}

template <typename T>
static int bn_cl_act_fwd_impl(const T *x, int S, int R, int C, int per_sample, const float *mean, const float *invstd,
                              const float *gamma, const float *beta, int relu, T *y, int ldy, void *stream) {
    MGAR_REQUIRE(cl_shape_ok(S, R, C) && ldy >= C && ldy % 4 == 0, "bn_cl_act_fwd: needs C % 4 == 0, C <= 1024, ldy >= C, ldy % 4 == 0");
    if ((long long)S * R * C == 0) return MGAR_OK;
    MGAR_REQUIRE(x && y && mean && invstd, "bn_cl_act_fwd: null pointer");
    MGAR_REQUIRE(((uintptr_t)y * 1) % (4 * sizeof(T)) == 0, "bn_cl_act_fwd: output slice not aligned to 4 elements");
    const long long rows = (long long)S * R, total = rows * (C / 4);
    long long blocks = (total + CL_THREADS - 1) / CL_THREADS;
    if (blocks > 16384) blocks = 16384;
    hipStream_t st = (hipStream_t)stream;
    KtScope kt(KT_BN_APPLY, st, 2.0 * sizeof(T) * (double)S * R * C);
    if (relu) hipLaunchKernelGGL((bn_cl_apply_kernel<true, T>), dim3((unsigned)blocks), dim3(CL_THREADS), 0, st, x, rows, R, C, per_sample, mean, invstd, gamma, beta, y, ldy);
    else hipLaunchKernelGGL((bn_cl_apply_kernel<false, T>), dim3((unsigned)blocks), dim3(CL_THREADS), 0, st, x, rows, R, C, per_sample, mean, invstd, gamma, beta, y, ldy);
    return check_launch("bn_cl_act_fwd: launch failed");
}

template <typename T>
static int bn_act_fwd_to_cl_impl(const T *x, int S, int C, int R, int per_sample, const float *mean, const float *invstd,
                                 const float *gamma, const float *beta, int relu, T *y, int ldy, void *stream) {
    MGAR_REQUIRE(S >= 0 && C >= 0 && R >= 0 && ldy >= C, "bn_act_fwd_to_cl: bad sizes");
    if ((long long)S * R * C == 0) return MGAR_OK;
    MGAR_REQUIRE(x && y && mean && invstd && S <= 65535, "bn_act_fwd_to_cl: null pointer or S > 65535");
    dim3 grid(ceil_div(R, 64), ceil_div(C, 64), S);
    hipStream_t st = (hipStream_t)stream;
    KtScope kt(KT_BN_APPLY, st, 2.0 * sizeof(T) * (double)S * R * C);
    if (relu) hipLaunchKernelGGL((bn_to_cl_apply_kernel<true, T>), grid, dim3(CL_THREADS), 0, st, x, C, R, per_sample, mean, invstd, gamma, beta, y, ldy);
    else hipLaunchKernelGGL((bn_to_cl_apply_kernel<false, T>), grid, dim3(CL_THREADS), 0, st, x, C, R, per_sample, mean, invstd, gamma, beta, y, ldy);
    return check_launch("bn_act_fwd_to_cl: launch failed");
}

template <typename T>
static int maxpool3d_cl_impl(const T *x, int N, int T_, int H, int W, int C, int kt, int kh, int kw, int st_, int sh, int sw, T *y,
                             void *stream) {
    MGAR_REQUIRE(N >= 0 && T_ >= 0 && H >= 0 && W >= 0 && C >= 0 && C % 4 == 0, "maxpool3d_same_fwd_cl: bad sizes (C % 4 == 0)");
    MGAR_REQUIRE(kt >= 1 && kh >= 1 && kw >= 1 && st_ >= 1 && sh >= 1 && sw >= 1, "maxpool3d_same_fwd_cl: bad kernel / stride");
    if ((long long)N * T_ * H * W * C == 0) return MGAR_OK;
    MGAR_REQUIRE(x && y, "maxpool3d_same_fwd_cl: null pointer");
    auto same = [](int size, int k, int s) { return size % s == 0 ? (k - s > 0 ? k - s : 0) : (k - size % s > 0 ? k - size % s : 0); };
    PoolCl g{T_, H, W, (T_ + st_ - 1) / st_, (H + sh - 1) / sh, (W + sw - 1) / sw, C, kt, kh, kw, st_, sh, sw,
             same(T_, kt, st_) / 2, same(H, kh, sh) / 2, same(W, kw, sw) / 2};
    const long long total = (long long)N * g.To * g.Ho * g.Wo * (C / 4);
    long long blocks = (total + CL_THREADS - 1) / CL_THREADS;
    if (blocks > 32768) blocks = 32768;
    hipStream_t st = (hipStream_t)stream;
    KtScope kt_(KT_MAXPOOL3D, st, (double)sizeof(T) * N * C * ((double)T_ * H * W + (double)g.To * g.Ho * g.Wo));
    hipLaunchKernelGGL(maxpool3d_cl_kernel<T>, dim3((unsigned)blocks), dim3(CL_THREADS), 0, st, x, total, g, y);
    return check_launch("maxpool3d_same_fwd_cl: launch failed");
}

// ---- C ABI (include/mgar_ops.h); _bf16 twins: x / y address bf16 elements, statistics stay fp32 --------------------------------
CL_API int mgar_bn_cl_train_stats(const float *x, int S, int R, int C, int per_sample, float eps, float momentum, float *workspace,
                                  float *mean, float *invstd, float *running_mean, float *running_var,
                                  long long *num_batches_tracked, void *stream) {
    return bn_cl_train_stats_impl<float>(x, S, R, C, per_sample, eps, momentum, workspace, mean, invstd, running_mean, running_var,
                                         num_batches_tracked, stream);
}
CL_API int mgar_bn_cl_train_stats_bf16(const void *x, int S, int R, int C, int per_sample, float eps, float momentum, float *workspace,
                                       float *mean, float *invstd, float *running_mean, float *running_var,
                                       long long *num_batches_tracked, void *stream) {
    return bn_cl_train_stats_impl<bf16_t>((const bf16_t *)x, S, R, C, per_sample, eps, momentum, workspace, mean, invstd, running_mean,
                                          running_var, num_batches_tracked, stream);
}
CL_API int mgar_bn_cl_act_fwd(const float *x, int S, int R, int C, int per_sample, const float *mean, const float *invstd,
                              const float *gamma, const float *beta, int relu, float *y, int ldy, void *stream) {
    return bn_cl_act_fwd_impl<float>(x, S, R, C, per_sample, mean, invstd, gamma, beta, relu, y, ldy, stream);
}
CL_API int mgar_bn_cl_act_fwd_bf16(const void *x, int S, int R, int C, int per_sample, const float *mean, const float *invstd,
                                   const float *gamma, const float *beta, int relu, void *y, int ldy, void *stream) {
    return bn_cl_act_fwd_impl<bf16_t>((const bf16_t *)x, S, R, C, per_sample, mean, invstd, gamma, beta, relu, (bf16_t *)y, ldy, stream);
}
CL_API int mgar_bn_act_fwd_to_cl(const float *x, int S, int C, int R, int per_sample, const float *mean, const float *invstd,
                                 const float *gamma, const float *beta, int relu, float *y, int ldy, void *stream) {
    return bn_act_fwd_to_cl_impl<float>(x, S, C, R, per_sample, mean, invstd, gamma, beta, relu, y, ldy, stream);
}
CL_API int mgar_bn_act_fwd_to_cl_bf16(const void *x, int S, int C, int R, int per_sample, const float *mean, const float *invstd,
                                      const float *gamma, const float *beta, int relu, void *y, int ldy, void *stream) {
    return bn_act_fwd_to_cl_impl<bf16_t>((const bf16_t *)x, S, C, R, per_sample, mean, invstd, gamma, beta, relu, (bf16_t *)y, ldy, stream);
}
CL_API int mgar_maxpool3d_same_fwd_cl(const float *x, int N, int T, int H, int W, int C, int kt, int kh, int kw, int st, int sh, int sw,
                                      float *y, void *stream) {
    return maxpool3d_cl_impl<float>(x, N, T, H, W, C, kt, kh, kw, st, sh, sw, y, stream);
}
CL_API int mgar_maxpool3d_same_fwd_cl_bf16(const void *x, int N, int T, int H, int W, int C, int kt, int kh, int kw, int st, int sh,
                                           int sw, void *y, void *stream) {
    return maxpool3d_cl_impl<bf16_t>((const bf16_t *)x, N, T, H, W, C, kt, kh, kw, st, sh, sw, (bf16_t *)y, stream);
}

// BatchNorm1d(train) [+ ReLU] backward over ROW-MAJOR x, dy (rows, C) fp32 (the sparse trunk's features): dx (rows, C), dgamma,
// dbeta (C).  workspace: mgar_bn_rows_bwd_workspace_floats(rows, C) floats.  C % 4 == 0, C <= 1024.
CL_API long long mgar_bn_rows_bwd_workspace_floats(int rows, int C) {
    if (!cl_shape_ok(1, rows, C)) return -1;
    const int chunk = cl_chunk_rows(1, rows), nchunk = (rows + chunk - 1) / chunk;
    return 2ll * C * (nchunk > 0 ? nchunk : 1) + 2ll * C;
}
CL_API int mgar_bn_rows_bwd(const float *dy, const float *x, int rows, int C, const float *mean, const float *invstd, const float *gamma,
                            const float *beta, int relu, float *workspace, float *dgamma, float *dbeta, float *dx, void *stream) {
    MGAR_REQUIRE(cl_shape_ok(1, rows, C), "bn_rows_bwd: needs C % 4 == 0, C <= 1024");
    if ((long long)rows * C == 0) return MGAR_OK;
    MGAR_REQUIRE(dy && x && mean && invstd && workspace && dx, "bn_rows_bwd: null pointer");
    const int chunk = cl_chunk_rows(1, rows), nchunk = (rows + chunk - 1) / chunk;
    hipStream_t st = (hipStream_t)stream;
    float *coef = workspace + (size_t)2 * C * nchunk;
    {
        KtScope kt(KT_BN_BWD_REDUCE, st, 8.0 * (double)rows * C);
        if (relu) hipLaunchKernelGGL(bn_rows_bwd_partial_kernel<true>, dim3(nchunk), dim3(CL_THREADS), 0, st, dy, x, rows, C, chunk, mean, invstd, gamma, beta, workspace);
        else hipLaunchKernelGGL(bn_rows_bwd_partial_kernel<false>, dim3(nchunk), dim3(CL_THREADS), 0, st, dy, x, rows, C, chunk, mean, invstd, gamma, beta, workspace);
    }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, st, workspace, nchunk, C, (double)rows, dgamma, dbeta, coef);
    const long long total = (long long)rows * (C / 4);
    long long blocks = (total + CL_THREADS - 1) / CL_THREADS;
    if (blocks > 16384) blocks = 16384;
    {
        KtScope kt(KT_BN_BWD_APPLY, st, 12.0 * (double)rows * C);
        if (relu) hipLaunchKernelGGL(bn_rows_bwd_apply_kernel<true>, dim3((unsigned)blocks), dim3(CL_THREADS), 0, st, dy, x, (long long)rows, C, mean, invstd, gamma, beta, coef, dx);
        else hipLaunchKernelGGL(bn_rows_bwd_apply_kernel<false>, dim3((unsigned)blocks), dim3(CL_THREADS), 0, st, dy, x, (long long)rows, C, mean, invstd, gamma, beta, coef, dx);
    }
    return check_launch("bn_rows_bwd: launch failed");
}
