// maxpool3d.hip -- 3-D max pooling with TensorFlow-"same" zero padding, forward only, for gfx950.
//
// Replaces MaxPool3dSamePadding.forward of the reference's I3D (model/backbone.py:99-131), i.e.
//     x = F.pad(x, same_pads)            # materialises a zero-padded copy of the activation
//     x = nn.MaxPool3d(kernel, stride)(x)
// Ten of these run per clip (3 strided pools + the 3x3x3 stride-1 pool of every Inception block).
// On ROCm the pad is a strided-copy kernel over the whole activation and the pooling kernel
// (max_pool3d_with_indices_single_out_frame) also produces an index tensor nobody reads: 21 ms +
// ~10 ms per c3 step.  I3D is frozen in MGAR-net (I3D_FREEZE), so only the forward is needed.
// The window is read straight from the un-padded input (rows stay in L1/L2), and "the window
// touches the zero padding" is folded in as max(., 0).  HBM-bound: input read once, output written
// once.  Main kernel: one thread per FOUR consecutive outputs of a row, walking the input slices t once
// with the last three 2-D pooled quads in registers (see maxpool3d_same_vec_kernel); per input row it
// loads one aligned 16-byte vector (two for stride 2) plus the one or two edge elements and forms the four
// horizontal maxima in registers.  A one-thread-per-output kernel covers the other geometries.
#include "common.hpp"
#include "payload.hpp"

namespace mgar {

struct Pool3dGeom {
    int T, H, W, To, Ho, Wo;
    int kt, kh, kw, st, sh, sw, pt, ph, pw;  // p* = FRONT padding of each axis
    int pad_zero;                            // 1: the zero padding takes part in the max (TF "same" on an activation); 0: max over the
                                             // window's VALID elements only (pooling a PRE-BatchNorm tensor, see mgar_maxpool3d_valid_fwd)
};

template <typename T>   // payload type: float or bf16_t (max of widened values is exact: bf16 in, bf16 out loses nothing)
__global__ __launch_bounds__(256) void maxpool3d_same_kernel(const T *__restrict__ x, long long total, Pool3dGeom g,
                                                             T *__restrict__ y) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int wo = (int)(e % g.Wo);
        long long r = e / g.Wo;
        const int ho = (int)(r % g.Ho); r /= g.Ho;
        const int to = (int)(r % g.To);
        const long long nc = r / g.To;
        const int t0 = to * g.st - g.pt, h0 = ho * g.sh - g.ph, w0 = wo * g.sw - g.pw;
        const int t1 = t0 + g.kt, h1 = h0 + g.kh, w1 = w0 + g.kw;
        const bool pad = t0 < 0 || h0 < 0 || w0 < 0 || t1 > g.T || h1 > g.H || w1 > g.W;
        float best = (pad && g.pad_zero) ? 0.f : -__builtin_inff();  // zero padding takes part in the max
        const T *base = x + (size_t)nc * g.T * g.H * g.W;
        for (int t = max(t0, 0); t < min(t1, g.T); ++t)
            for (int h = max(h0, 0); h < min(h1, g.H); ++h) {
                const T *row = base + ((size_t)t * g.H + h) * g.W;
                for (int w = max(w0, 0); w < min(w1, g.W); ++w) best = fmaxf(best, Payload<T>::ld(row + w));
            }
        Payload<T>::st(y + e, best);
    }
}

// kw == 3 and (SW == 1, front pad 1) or (SW == 2, front pad 0); W % 4 == 0, Wo % 4 == 0; kt <= 3.
// One thread owns a quad of output columns (ho, 4 wo) of one (n, c) and walks the input slices t once: per slice
// it forms the 2-D pooled quad (kh rows x [one aligned 16-byte load (two for stride 2) + the edge elements]) and
// keeps the last three of them in registers; an output slice is emitted when its last input slice has been seen.
// Every input element is loaded once per (kh x kw) window it belongs to -- 9 loads per output quad instead of 27
// for the 3x3x3 pools, and the t-reuse no longer depends on the L2.  grid (ceil(Ho * Wo/4 / 256), NC)
template <int SW, typename T>
__global__ __launch_bounds__(256) void maxpool3d_same_vec_kernel(const T *__restrict__ x, Pool3dGeom g, T *__restrict__ y) {
    const int nq = g.Wo >> 2;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int ho = q / nq, wq = q - ho * nq;
    if (ho >= g.Ho) return;
    const size_t nc = blockIdx.y;
    const int h0 = ho * g.sh - g.ph, h1 = h0 + g.kh;
    const bool pad_h = h0 < 0 || h1 > g.H;
    const float ninf = -__builtin_inff();
    const T *base = x + nc * g.T * g.H * g.W;
    const int wb = wq * 4 * SW;  // first aligned input column of this quad
    const bool pad_l = SW == 1 && wb == 0;
    const bool pad_r = SW == 1 ? (wb + 4 >= g.W) : (wb + 8 >= g.W);
    float4 p0 = make_float4(ninf, ninf, ninf, ninf), p1 = p0, p2 = p0;   // 2-D pooled quads of slices t-2, t-1, t
    int to = 0;
    for (int t = 0; t < g.T; ++t) {
        p0 = p1; p1 = p2;
        float b0 = ninf, b1 = ninf, b2 = ninf, b3 = ninf;
        for (int h = max(h0, 0); h < min(h1, g.H); ++h) {
            const T *row = base + ((size_t)t * g.H + h) * g.W + wb;
            if (SW == 1) {  // outputs j = 0..3 cover inputs wb + j - 1 .. wb + j + 1
                const float4 v = Payload<T>::ld4(row);
                const float lft = wb > 0 ? Payload<T>::ld(row - 1) : ninf;
                const float rgt = wb + 4 < g.W ? Payload<T>::ld(row + 4) : ninf;
                b0 = fmaxf(b0, fmaxf(fmaxf(lft, v.x), v.y));
                b1 = fmaxf(b1, fmaxf(fmaxf(v.x, v.y), v.z));
                b2 = fmaxf(b2, fmaxf(fmaxf(v.y, v.z), v.w));
                b3 = fmaxf(b3, fmaxf(fmaxf(v.z, v.w), rgt));
            } else {        // outputs j cover inputs wb + 2j .. wb + 2j + 2
                const float4 v = Payload<T>::ld4(row);
                const float4 u = Payload<T>::ld4(row + 4);
                const float rgt = wb + 8 < g.W ? Payload<T>::ld(row + 8) : ninf;
                b0 = fmaxf(b0, fmaxf(fmaxf(v.x, v.y), v.z));
                b1 = fmaxf(b1, fmaxf(fmaxf(v.z, v.w), u.x));
                b2 = fmaxf(b2, fmaxf(fmaxf(u.x, u.y), u.z));
                b3 = fmaxf(b3, fmaxf(fmaxf(u.z, u.w), rgt));
            }
        }
        p2 = make_float4(b0, b1, b2, b3);
        // emit every output slice whose window ends at (or is clipped to) input slice t
        while (to < g.To) {
            const int t0 = to * g.st - g.pt, t1 = t0 + g.kt;
            if (min(t1, g.T) - 1 != t) break;
            const int lo = max(t0, 0);                     // window = slices lo .. t  (at most 3)
            float4 r = p2;
            if (lo <= t - 1) { r.x = fmaxf(r.x, p1.x); r.y = fmaxf(r.y, p1.y); r.z = fmaxf(r.z, p1.z); r.w = fmaxf(r.w, p1.w); }
            if (lo <= t - 2) { r.x = fmaxf(r.x, p0.x); r.y = fmaxf(r.y, p0.y); r.z = fmaxf(r.z, p0.z); r.w = fmaxf(r.w, p0.w); }
            const bool pad = pad_h || t0 < 0 || t1 > g.T;  // zero padding takes part in the max wherever the window leaves the input
            if (g.pad_zero) {
                if (pad || pad_l) r.x = fmaxf(r.x, 0.f);
                if (pad) { r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); }
                if (pad || pad_r) r.w = fmaxf(r.w, 0.f);
            }
            Payload<T>::st4(y + ((nc * g.To + to) * g.Ho + ho) * g.Wo + wq * 4, r);
            ++to;
        }
    }
}

}  // namespace mgar

using namespace mgar;

template <typename T_>
static int maxpool3d_same_fwd_impl(const T_ *x, int NC, int T, int H, int W, int kt, int kh, int kw, int st, int sh, int sw, T_ *y,
                                   void *stream, int pad_zero = 1) {
    MGAR_REQUIRE(NC >= 0 && T > 0 && H > 0 && W > 0 && kt > 0 && kh > 0 && kw > 0 && st > 0 && sh > 0 && sw > 0,
                 "maxpool3d_same_fwd: bad sizes");
    if (NC == 0) return MGAR_OK;
    MGAR_REQUIRE(x && y, "maxpool3d_same_fwd: null pointer");
    auto front = [](int size, int k, int s) {   // model/backbone.py:101-105, :123-128
        const int total = size % s == 0 ? (k - s > 0 ? k - s : 0) : (k - size % s > 0 ? k - size % s : 0);
        return total / 2;
    };
    Pool3dGeom g{T, H, W, (T + st - 1) / st, (H + sh - 1) / sh, (W + sw - 1) / sw, kt, kh, kw, st, sh, sw,
                 front(T, kt, st), front(H, kh, sh), front(W, kw, sw), pad_zero};
    KtScope ktimer(KT_MAXPOOL3D, (hipStream_t)stream, (double)sizeof(T_) * NC * ((double)T * H * W + (double)g.To * g.Ho * g.Wo));
    const bool vec1 = kw == 3 && sw == 1 && g.pw == 1, vec2 = kw == 3 && sw == 2 && g.pw == 0 && W % 2 == 0;
    if ((vec1 || vec2) && kt <= 3 && W % 4 == 0 && g.Wo % 4 == 0 && NC <= 65535 && (!vec2 || W >= 8)) {
        dim3 grid(ceil_div(g.Ho * (g.Wo / 4), 256), NC);
        if (vec1) hipLaunchKernelGGL((maxpool3d_same_vec_kernel<1, T_>), grid, dim3(256), 0, (hipStream_t)stream, x, g, y);
        else hipLaunchKernelGGL((maxpool3d_same_vec_kernel<2, T_>), grid, dim3(256), 0, (hipStream_t)stream, x, g, y);
        return check_launch("maxpool3d_same_fwd: launch failed");
    }
    const long long total = (long long)NC * g.To * g.Ho * g.Wo;
    const int blocks = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    hipLaunchKernelGGL(maxpool3d_same_kernel<T_>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, total, g, y);
    return check_launch("maxpool3d_same_fwd: launch failed");
}

extern "C" __attribute__((visibility("default"))) int mgar_maxpool3d_same_fwd(const float *x, int NC, int T, int H, int W, int kt, int kh,
                                                                             int kw, int st, int sh, int sw, float *y, void *stream) {
    return maxpool3d_same_fwd_impl<float>(x, NC, T, H, W, kt, kh, kw, st, sh, sw, y, stream);
}
extern "C" __attribute__((visibility("default"))) int mgar_maxpool3d_same_fwd_bf16(const void *x, int NC, int T, int H, int W, int kt,
                                                                                  int kh, int kw, int st, int sh, int sw, void *y,
                                                                                  void *stream) {
    return maxpool3d_same_fwd_impl<bf16_t>((const bf16_t *)x, NC, T, H, W, kt, kh, kw, st, sh, sw, (bf16_t *)y, stream);
}

// The same windows, maximum over the VALID elements only (no zero padding in the max): pooling a PRE-BatchNorm tensor.  With
// gamma > 0, relu(bn(.)) is monotone non-decreasing per channel -- in fp32 too: subtract, multiply by a positive constant, add
// and max(., 0) are all monotone -- and relu(.) >= 0 makes the zero padding a no-op, so
//     maxpool_same(relu(bn(x))) == relu(bn(maxpool_valid(x)))     bit for bit,
// and the BatchNorm + ReLU pass runs over the pooled tensor (1/4 ... 1/8 of the elements) instead of the full one.
extern "C" __attribute__((visibility("default"))) int mgar_maxpool3d_valid_fwd(const float *x, int NC, int T, int H, int W, int kt, int kh,
                                                                              int kw, int st, int sh, int sw, float *y, void *stream) {
    return maxpool3d_same_fwd_impl<float>(x, NC, T, H, W, kt, kh, kw, st, sh, sw, y, stream, 0);
}
extern "C" __attribute__((visibility("default"))) int mgar_maxpool3d_valid_fwd_bf16(const void *x, int NC, int T, int H, int W, int kt,
                                                                                   int kh, int kw, int st, int sh, int sw, void *y,
                                                                                   void *stream) {
    return maxpool3d_same_fwd_impl<bf16_t>((const bf16_t *)x, NC, T, H, W, kt, kh, kw, st, sh, sw, (bf16_t *)y, stream, 0);
}
