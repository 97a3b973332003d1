// errors.hip -- per-thread last-error string, ABI version and the optional per-kernel timers of
// libmgar_hip.so.
#include "common.hpp"

#include <mutex>
#include <vector>

namespace mgar {
static thread_local const char *g_last_error = "";
void set_error(const char *msg) { g_last_error = msg; }

int g_kt_on = 0;
namespace {
struct KtLaunch { hipEvent_t a, b; };
struct KtSlot {
    std::vector<KtLaunch> pending;
    hipEvent_t open = nullptr;
    double ms = 0.0, bytes = 0.0, flops = 0.0;
    long long launches = 0;
};
KtSlot g_kt[KT_COUNT];
std::mutex g_kt_mu;
const char *const kKtNames[KT_COUNT] = {
    "fps_kernel", "ball_query_kernel", "three_nn_kernel", "three_interp_fwd", "three_interp_bwd", "query_group_fwd",
    "query_group_bwd", "bn_partial_kernel", "bn_apply_kernel", "bn_max_vec_kernel", "bn_bwd_partial_kernel",
    "bn_bwd_apply_kernel", "bn_max_bwd_partial_kernel", "bn_max_bwd_apply_kernel", "pointwise_fwd_kernel",
    "pointwise_dw_kernel", "rowmajor_dw_kernel", "maxpool3d_same_kernel", "voxel_roi_pool_fwd", "voxel_roi_pool_bwd", "stem_conv3d_kernel", "query_group_inverse_index",
    "image_resize_normalize",
    "dafm_attn_fwd", "dafm_attn_bwd", "gatv2_fwd", "gatv2_bwd", "roi_align_fwd", "roi_align_bwd", "voxel_query_kernel", "points_in_boxes_kernel",
    "roipoint_pool3d_kernel", "spconv_index", "spconv_gemm", "spconv_dw", "point_grid_build", "ball_query_grid_kernel", "three_nn_grid_kernel", "conv3d_wino_kernel"};
}  // namespace

void kt_begin(int id, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_kt_mu);
    KtSlot &s = g_kt[id];
    if (hipEventCreate(&s.open) != hipSuccess) { s.open = nullptr; return; }
    (void)hipEventRecord(s.open, st);
}
void kt_end(int id, hipStream_t st, double bytes, double flops) {
    std::lock_guard<std::mutex> lk(g_kt_mu);
    KtSlot &s = g_kt[id];
    if (!s.open) return;
    hipEvent_t b;
    if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(s.open); s.open = nullptr; return; }
    (void)hipEventRecord(b, st);
    s.pending.push_back({s.open, b});
    s.open = nullptr;
    s.bytes += bytes;
    s.flops += flops;
    s.launches += 1;
}
}  // namespace mgar

using namespace mgar;

extern "C" __attribute__((visibility("default"))) int mgar_abi_version(void) { return 12; }
extern "C" __attribute__((visibility("default"))) const char *mgar_last_error(void) { return mgar::g_last_error; }

extern "C" __attribute__((visibility("default"))) int mgar_ktimer_enable(int on) {
    g_kt_on = on ? 1 : 0;
    return MGAR_OK;
}
extern "C" __attribute__((visibility("default"))) int mgar_ktimer_count(void) { return KT_COUNT; }
extern "C" __attribute__((visibility("default"))) const char *mgar_ktimer_name(int id) {
    return id >= 0 && id < KT_COUNT ? kKtNames[id] : "";
}
// Adds flops to kernel `id`'s total: for launches whose work the library cannot know (stacked layouts keep their
// per-sample counts on the device), the instrumenting caller supplies it.
extern "C" __attribute__((visibility("default"))) int mgar_ktimer_add_flops(int id, double flops) {
    MGAR_REQUIRE(id >= 0 && id < KT_COUNT, "ktimer_add_flops: bad kernel id");
    if (!g_kt_on) return MGAR_OK;
    std::lock_guard<std::mutex> lk(g_kt_mu);
    g_kt[id].flops += flops;
    return MGAR_OK;
}
extern "C" __attribute__((visibility("default"))) int mgar_ktimer_add_bytes(int id, double bytes) {
    MGAR_REQUIRE(id >= 0 && id < KT_COUNT, "ktimer_add_bytes: bad kernel id");
    if (!g_kt_on) return MGAR_OK;
    std::lock_guard<std::mutex> lk(g_kt_mu);
    g_kt[id].bytes += bytes;
    return MGAR_OK;
}
// Waits for the recorded events of kernel `id`, adds their elapsed times, returns the totals since the
// last reset (reset != 0 clears them afterwards).
extern "C" __attribute__((visibility("default"))) int mgar_ktimer_read(int id, double *total_ms, long long *launches,
                                                                      double *total_bytes, double *total_flops, int reset) {
    MGAR_REQUIRE(id >= 0 && id < KT_COUNT, "ktimer_read: bad kernel id");
    std::lock_guard<std::mutex> lk(g_kt_mu);
    KtSlot &s = g_kt[id];
    for (KtLaunch &l : s.pending) {
        float ms = 0.f;
        if (hipEventSynchronize(l.b) == hipSuccess && hipEventElapsedTime(&ms, l.a, l.b) == hipSuccess) s.ms += ms;
        (void)hipEventDestroy(l.a);
        (void)hipEventDestroy(l.b);
    }
    s.pending.clear();
    if (total_ms) *total_ms = s.ms;
    if (launches) *launches = s.launches;
    if (total_bytes) *total_bytes = s.bytes;
    if (total_flops) *total_flops = s.flops;
    if (reset) { s.ms = s.bytes = s.flops = 0.0; s.launches = 0; }
    return MGAR_OK;
}

// ---- a timed gap on a stream ------------------------------------------------------------------------------------------------
// One wave that sleeps for `us` microseconds (constant 100 MHz wall clock; bounded, no memory polling).  The clip model puts
// it at the head of the RGB side stream, behind an event recorded just before the level-1 FPS launch: a 1024-thread workgroup
// per cloud needs half a CU at once, and once the I3D stem's tens of thousands of workgroups are being dispatched the freed
// slots go to the stem's next workgroups -- measured, the sampling then starts when the stem ENDS (4.6 -> 13.9 ms, and the whole
// LiDAR branch behind it).  A few microseconds of head start let its workgroups become resident first.
namespace mgar {
__global__ void delay_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace mgar
extern "C" __attribute__((visibility("default"))) int mgar_delay_us(int us, void *stream) {
    MGAR_REQUIRE(us >= 0 && us <= 1000, "delay_us: 0 .. 1000 microseconds");
    if (us == 0) return MGAR_OK;
    hipLaunchKernelGGL(mgar::delay_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)us * 100);
    return check_launch("delay_us: launch failed");
}
