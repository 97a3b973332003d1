// common.hpp -- shared helpers for the gfx950 kernels of libmgar_hip.so.
// Written for CDNA4 only: 64-wide wavefronts, 160 KB LDS / CU, no dual-backend macros.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mgar_ops.h"

namespace mgar {

constexpr int kWave = 64;

// Squared distance with the contraction pinned (see oracle/mgar_oracle.c header and
// DESIGN.md "floating-point convention"): fma(dz,dz, fma(dx,dx, dy*dy)).
// The library is compiled with -ffp-contract=off so nothing else is fused.
__device__ __forceinline__ float d2_of(float dx, float dy, float dz) {
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy));
}
__device__ __forceinline__ float dot3_of(float w0, float p0, float w1, float p1, float w2, float p2) {
    return __builtin_fmaf(w2, p2, __builtin_fmaf(w0, p0, w1 * p1));
}

// ---- wave-wide float reductions on the DPP crossbar (no LDS, no ds_bpermute) -----------
// row_shr 1,2,4,8 leave each 16-lane row's total in its lane 15; row_bcast15/31 chain the
// rows; the wave total sits in lane 63 and is returned wave-uniform through v_readlane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_get(float identity, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity),
                                                                 __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_get<0x111, 0xF>(0.f, v);
    v += dpp_get<0x112, 0xF>(0.f, v);
    v += dpp_get<0x114, 0xF>(0.f, v);
    v += dpp_get<0x118, 0xF>(0.f, v);
    v += dpp_get<0x142, 0xA>(0.f, v);
    v += dpp_get<0x143, 0xC>(0.f, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    const float ninf = -__builtin_inff();
    v = fmaxf(v, dpp_get<0x111, 0xF>(ninf, v));
    v = fmaxf(v, dpp_get<0x112, 0xF>(ninf, v));
    v = fmaxf(v, dpp_get<0x114, 0xF>(ninf, v));
    v = fmaxf(v, dpp_get<0x118, 0xF>(ninf, v));
    v = fmaxf(v, dpp_get<0x142, 0xA>(ninf, v));
    v = fmaxf(v, dpp_get<0x143, 0xC>(ninf, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// A raw buffer resource over `bytes` bytes at a wave-uniform address (the address is pinned to scalar registers: a descriptor the
// compiler cannot prove uniform costs a waterfall loop around every load).  Loads at an offset >= bytes return 0.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_buffer(const void *p, int bytes) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

void set_error(const char *msg);

// ---- optional per-kernel timing (bench.py's roofline table) --------------------------------------
// When switched on through mgar_ktimer_enable(), a launcher brackets ONE kernel launch with two HIP
// events recorded on the launch stream and notes the launch's algorithmic bytes / flops
// (SURVEY.md section 8d figures); mgar_ktimer_read() resolves the events.
// Off by default: one predictable branch per launch.
enum KernelId {
    KT_FPS = 0, KT_BALL_QUERY, KT_THREE_NN, KT_THREE_INTERP_FWD, KT_THREE_INTERP_BWD, KT_QUERY_GROUP_FWD, KT_QUERY_GROUP_BWD,
    KT_BN_STATS, KT_BN_APPLY, KT_BN_MAX, KT_BN_BWD_REDUCE, KT_BN_BWD_APPLY, KT_BN_MAX_BWD_REDUCE, KT_BN_MAX_BWD_APPLY,
    KT_POINTWISE_FWD, KT_POINTWISE_DW, KT_ROWMAJOR_DW, KT_MAXPOOL3D, KT_VOXEL_ROI_POOL_FWD, KT_VOXEL_ROI_POOL_BWD, KT_STEM_CONV, KT_QG_INDEX, KT_IMAGE_PREP,
    KT_DAFM_FWD, KT_DAFM_BWD, KT_GATV2_FWD, KT_GATV2_BWD, KT_ROI_ALIGN_FWD, KT_ROI_ALIGN_BWD, KT_VOXEL_QUERY, KT_POINTS_IN_BOXES,
    KT_ROIPOINT_POOL, KT_SPCONV_INDEX, KT_SPCONV_GEMM, KT_SPCONV_DW, KT_POINT_GRID, KT_BALL_QUERY_GRID, KT_THREE_NN_GRID, KT_CONV3D_WINO, KT_COUNT
};
extern int g_kt_on;
void kt_begin(int id, hipStream_t st);
void kt_end(int id, hipStream_t st, double bytes, double flops);
struct KtScope {  // RAII: events around the launches issued while it is alive
    int id; hipStream_t st; double bytes, flops;
    KtScope(int id_, hipStream_t st_, double bytes_, double flops_ = 0.0) : id(id_), st(st_), bytes(bytes_), flops(flops_) {
        if (g_kt_on) kt_begin(id, st);
    }
    ~KtScope() { if (g_kt_on) kt_end(id, st, bytes, flops); }
};

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(what);
        return MGAR_ELAUNCH;
    }
    return MGAR_OK;
}

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

#define MGAR_REQUIRE(cond, msg)        \
    do {                               \
        if (!(cond)) {                 \
            ::mgar::set_error(msg);    \
            return MGAR_EINVAL;        \
        }                              \
    } while (0)

// Find the segment of a stacked layout that `pt` falls in and the prefix sums before it.
// cnt_a: counts that define the segments of pt's own array; cnt_b: counts of the companion
// array whose start offset is wanted.  Mirrors the per-thread search of
// pointnet2_stack/src/ball_query_gpu.cu:27-35 but is only ever run wave-uniformly.
struct Segment {
    int bs;       // segment index
    int a_start;  // start row of the segment in array a
    int b_start;  // start row of the segment in array b
};

__device__ __forceinline__ Segment find_segment(int pt, int B, const int *__restrict__ cnt_a,
                                                const int *__restrict__ cnt_b) {
    Segment s{0, 0, 0};
    int acc = cnt_a[0];
    for (int k = 1; k < B; ++k) {
        if (pt < acc) break;
        s.a_start = acc;
        s.b_start += cnt_b[k - 1];
        acc += cnt_a[k];
        s.bs = k;
    }
    return s;
}

}  // namespace mgar
