// maxpool3d.hip -- 3-D max pooling with TensorFlow-"same" zero padding, forward only, for gfx950.
//
// Replaces MaxPool3dSamePadding.forward of the reference's I3D (model/backbone.py:99-131), i.e.
//     x = F.pad(x, same_pads)            # materialises a zero-padded copy of the activation
//     x = nn.MaxPool3d(kernel, stride)(x)
// Ten of these run per clip (3 strided pools + the 3x3x3 stride-1 pool of every Inception block).
// On ROCm the pad is a strided-copy kernel over the whole activation and the pooling kernel
// (max_pool3d_with_indices_single_out_frame) also produces an index tensor nobody reads: 21 ms +
// ~10 ms per c3 step.  I3D is frozen in MGAR-net (I3D_FREEZE), so only the forward is needed.
// One thread per output element, w fastest; the window is read straight from the un-padded input
// (rows stay in L1/L2), and "the window touches the zero padding" is folded in as max(., 0).
// HBM-bound: input read once, output written once.
#include "common.hpp"

namespace mgar {

struct Pool3dGeom {
    int T, H, W, To, Ho, Wo;
    int kt, kh, kw, st, sh, sw, pt, ph, pw;  // p* = FRONT padding of each axis
};

__global__ __launch_bounds__(256) void maxpool3d_same_kernel(const float *__restrict__ x, long long total, Pool3dGeom g,
                                                             float *__restrict__ y) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int wo = (int)(e % g.Wo);
        long long r = e / g.Wo;
        const int ho = (int)(r % g.Ho); r /= g.Ho;
        const int to = (int)(r % g.To);
        const long long nc = r / g.To;
        const int t0 = to * g.st - g.pt, h0 = ho * g.sh - g.ph, w0 = wo * g.sw - g.pw;
        const int t1 = t0 + g.kt, h1 = h0 + g.kh, w1 = w0 + g.kw;
        const bool pad = t0 < 0 || h0 < 0 || w0 < 0 || t1 > g.T || h1 > g.H || w1 > g.W;
        float best = pad ? 0.f : -__builtin_inff();  // zero padding takes part in the max
        const float *base = x + (size_t)nc * g.T * g.H * g.W;
        for (int t = max(t0, 0); t < min(t1, g.T); ++t)
            for (int h = max(h0, 0); h < min(h1, g.H); ++h) {
                const float *row = base + ((size_t)t * g.H + h) * g.W;
                for (int w = max(w0, 0); w < min(w1, g.W); ++w) best = fmaxf(best, row[w]);
            }
        y[e] = best;
    }
}

}  // namespace mgar

using namespace mgar;

extern "C" __attribute__((visibility("default"))) int mgar_maxpool3d_same_fwd(const float *x, int NC, int T, int H, int W,
                                                                             int kt, int kh, int kw, int st, int sh, int sw,
                                                                             float *y, void *stream) {
    MGAR_REQUIRE(NC >= 0 && T > 0 && H > 0 && W > 0 && kt > 0 && kh > 0 && kw > 0 && st > 0 && sh > 0 && sw > 0,
                 "maxpool3d_same_fwd: bad sizes");
    if (NC == 0) return MGAR_OK;
    MGAR_REQUIRE(x && y, "maxpool3d_same_fwd: null pointer");
    auto front = [](int size, int k, int s) {   // model/backbone.py:101-105, :123-128
        const int total = size % s == 0 ? (k - s > 0 ? k - s : 0) : (k - size % s > 0 ? k - size % s : 0);
        return total / 2;
    };
    Pool3dGeom g{T, H, W, (T + st - 1) / st, (H + sh - 1) / sh, (W + sw - 1) / sw, kt, kh, kw, st, sh, sw,
                 front(T, kt, st), front(H, kh, sh), front(W, kw, sw)};
    const long long total = (long long)NC * g.To * g.Ho * g.Wo;
    const int blocks = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    hipLaunchKernelGGL(maxpool3d_same_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, total, g, y);
    return check_launch("maxpool3d_same_fwd: launch failed");
}
