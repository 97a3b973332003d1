// roi_align.hip -- RoIAlign forward / backward for gfx950.
//
// Replaces the third-party op the reference calls at model/gat_model.py:1056-1057 and
// model/sg_model.py:96-97 (torchvision.ops.roi_align, torchvision==0.17.2 per
// requirements.txt:422; aligned=False, sampling_ratio=-1 at the call sites).
// Arithmetic restated from the published RoIAlign definition (Detectron / torchvision
// semantics): box * spatial_scale, w/h clamped to >= 1 (aligned=False), bin = size/pooled,
// adaptive ceil(size/pooled) samples per bin axis, bilinear sampling with the
// out-of-range (< -1 or > size) -> 0 and clamp-to-border rules, mean over samples.
//
// One thread per output element (k, c, ph, pw), pw fastest: the threads of a wave share
// the RoI and mostly the channel plane, so the 4 bilinear taps of neighbouring bins fall in
// the same few cache lines; the (N,C,H,W) map of one clip (12 MB at 832x45x80) stays in L2 /
// Infinity Cache across its RoIs.  Gather-bound; the backward scatters with float atomics.
#include "common.hpp"
#include "payload.hpp"

namespace mgar {

struct RoiGeom {
    int batch;
    float x1, y1, bin_w, bin_h;
    int gw, gh;
    float inv_count;
};

__device__ __forceinline__ RoiGeom roi_geom(const float *__restrict__ r, float scale, int ph, int pw, int sampling_ratio,
                                            int aligned) {
    RoiGeom g;
    const float off = aligned ? 0.5f : 0.f;
    g.batch = (int)r[0];
    g.x1 = r[1] * scale - off;
    g.y1 = r[2] * scale - off;
    float rw = r[3] * scale - off - g.x1, rh = r[4] * scale - off - g.y1;
    if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
    g.bin_w = rw / (float)pw;
    g.bin_h = rh / (float)ph;
    g.gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)ph);
    g.gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)pw);
    g.inv_count = 1.f / fmaxf((float)(g.gh * g.gw), 1.f);
    return g;
}

struct Taps {
    int o1, o2, o3, o4;
    float w1, w2, w3, w4;
    bool ok;
};

__device__ __forceinline__ Taps bilinear_taps(int H, int W, float y, float x) {
    Taps t;
    t.ok = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
    if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
    const float ly = y - y_low, lx = x - x_low, hy = 1.f - ly, hx = 1.f - lx;
    t.o1 = y_low * W + x_low; t.o2 = y_low * W + x_high; t.o3 = y_high * W + x_low; t.o4 = y_high * W + x_high;
    t.w1 = hy * hx; t.w2 = hy * lx; t.w3 = ly * hx; t.w4 = ly * lx;
    return t;
}

template <typename T>   // payload type of input / out; box geometry, tap weights and the accumulation are fp32
__global__ __launch_bounds__(256) void roi_align_fwd_kernel(long long total, const T *__restrict__ input, int C, int H,
                                                            int W, const float *__restrict__ rois, int PH, int PW,
                                                            float scale, int sampling_ratio, int aligned,
                                                            T *__restrict__ out) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int pw = (int)(e % PW);
        const int ph = (int)((e / PW) % PH);
        const int c = (int)((e / ((long long)PW * PH)) % C);
        const int k = (int)(e / ((long long)PW * PH * C));
        const RoiGeom g = roi_geom(rois + (size_t)k * 5, scale, PH, PW, sampling_ratio, aligned);
        const T *img = input + ((size_t)g.batch * C + c) * H * W;
        float acc = 0.f;
        for (int iy = 0; iy < g.gh; ++iy) {
            const float y = g.y1 + ph * g.bin_h + (iy + .5f) * g.bin_h / (float)g.gh;
            for (int ix = 0; ix < g.gw; ++ix) {
                const float x = g.x1 + pw * g.bin_w + (ix + .5f) * g.bin_w / (float)g.gw;
                const Taps t = bilinear_taps(H, W, y, x);
                if (t.ok) acc += t.w1 * Payload<T>::ld(img + t.o1) + t.w2 * Payload<T>::ld(img + t.o2) + t.w3 * Payload<T>::ld(img + t.o3) +
                                 t.w4 * Payload<T>::ld(img + t.o4);
            }
        }
        Payload<T>::st(out + e, acc * g.inv_count);
    }
}

__global__ __launch_bounds__(256) void roi_align_bwd_kernel(long long total, const float *__restrict__ grad_out, int C,
                                                            int H, int W, const float *__restrict__ rois, int PH, int PW,
                                                            float scale, int sampling_ratio, int aligned,
                                                            float *__restrict__ grad_input) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int pw = (int)(e % PW);
        const int ph = (int)((e / PW) % PH);
        const int c = (int)((e / ((long long)PW * PH)) % C);
        const int k = (int)(e / ((long long)PW * PH * C));
        const RoiGeom g = roi_geom(rois + (size_t)k * 5, scale, PH, PW, sampling_ratio, aligned);
        float *img = grad_input + ((size_t)g.batch * C + c) * H * W;
        const float go = grad_out[e] * g.inv_count;
        for (int iy = 0; iy < g.gh; ++iy) {
            const float y = g.y1 + ph * g.bin_h + (iy + .5f) * g.bin_h / (float)g.gh;
            for (int ix = 0; ix < g.gw; ++ix) {
                const float x = g.x1 + pw * g.bin_w + (ix + .5f) * g.bin_w / (float)g.gw;
                const Taps t = bilinear_taps(H, W, y, x);
                if (t.ok) {
                    atomicAdd(img + t.o1, go * t.w1);
                    atomicAdd(img + t.o2, go * t.w2);
                    atomicAdd(img + t.o3, go * t.w3);
                    atomicAdd(img + t.o4, go * t.w4);
                }
            }
        }
    }
}

}  // namespace mgar

using namespace mgar;

static int roi_args_ok(int N, int C, int H, int W, int K, int ph, int pw) {
    return N >= 0 && C >= 0 && H > 0 && W > 0 && K >= 0 && ph > 0 && pw > 0;
}

template <typename T>
static int roi_align_fwd_impl(const T *input, int N, int C, int H, int W, const float *rois, int K, int pooled_h, int pooled_w,
                              float spatial_scale, int sampling_ratio, int aligned, T *out, void *stream) {
    MGAR_REQUIRE(roi_args_ok(N, C, H, W, K, pooled_h, pooled_w), "roi_align_fwd: bad sizes");
    const long long total = (long long)K * C * pooled_h * pooled_w;
    if (total == 0) return MGAR_OK;
    MGAR_REQUIRE(input && rois && out, "roi_align_fwd: null pointer");
    const int blocks = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    // the feature map read once + the RoIs + the crops written
    KtScope kt(KT_ROI_ALIGN_FWD, (hipStream_t)stream, sizeof(T) * ((double)N * C * H * W + (double)total) + 20.0 * K);
    hipLaunchKernelGGL(roi_align_fwd_kernel<T>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, total, input, C, H, W, rois,
                       pooled_h, pooled_w, spatial_scale, sampling_ratio, aligned, out);
    return check_launch("roi_align_fwd: launch failed");
}
extern "C" __attribute__((visibility("default"))) int mgar_roi_align_fwd(const float *input, int N, int C, int H, int W,
                                                                        const float *rois, int K, int pooled_h,
                                                                        int pooled_w, float spatial_scale,
                                                                        int sampling_ratio, int aligned, float *out,
                                                                        void *stream) {
    return roi_align_fwd_impl<float>(input, N, C, H, W, rois, K, pooled_h, pooled_w, spatial_scale, sampling_ratio, aligned, out, stream);
}
// bf16 payload: input / out address bf16 elements, rois stay fp32
extern "C" __attribute__((visibility("default"))) int mgar_roi_align_fwd_bf16(const void *input, int N, int C, int H, int W,
                                                                             const float *rois, int K, int pooled_h,
                                                                             int pooled_w, float spatial_scale,
                                                                             int sampling_ratio, int aligned, void *out,
                                                                             void *stream) {
    return roi_align_fwd_impl<bf16_t>((const bf16_t *)input, N, C, H, W, rois, K, pooled_h, pooled_w, spatial_scale, sampling_ratio,
                                      aligned, (bf16_t *)out, stream);
}

extern "C" __attribute__((visibility("default"))) int mgar_roi_align_bwd(const float *grad_out, int N, int C, int H, int W,
                                                                        const float *rois, int K, int pooled_h,
                                                                        int pooled_w, float spatial_scale,
                                                                        int sampling_ratio, int aligned,
                                                                        float *grad_input, void *stream) {
    MGAR_REQUIRE(roi_args_ok(N, C, H, W, K, pooled_h, pooled_w), "roi_align_bwd: bad sizes");
    const long long total = (long long)K * C * pooled_h * pooled_w;
    if (total == 0) return MGAR_OK;
    MGAR_REQUIRE(grad_out && rois && grad_input, "roi_align_bwd: null pointer");
    const int blocks = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    KtScope kt(KT_ROI_ALIGN_BWD, (hipStream_t)stream, 4.0 * ((double)total + 4.0 * (double)total) + 20.0 * K);   // crops' gradient read + <= 4 taps per sample written
    hipLaunchKernelGGL(roi_align_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, total, grad_out, C, H, W,
                       rois, pooled_h, pooled_w, spatial_scale, sampling_ratio, aligned, grad_input);
    return check_launch("roi_align_bwd: launch failed");
}
