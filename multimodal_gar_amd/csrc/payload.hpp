// payload.hpp -- feature-payload element types of the gfx950 kernels: float and bf16.
//
// the bf16 configurations keep coordinates, distances, indices and BatchNorm statistics
// in fp32 / int32 and use bf16 only for FEATURE PAYLOADS (and the GEMMs).  Every payload kernel is a template on the
// storage type T; arithmetic is always fp32 -- a bf16 element is widened on load (a 16-bit shift) and narrowed on
// store with v_cvt_pk_bf16_f32 (round to nearest even), so index outputs cannot depend on the payload type.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mgar {

struct bf16_t { uint16_t bits; };   // storage only (same layout as torch.bfloat16)

typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));
typedef float f32x2_v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {   // one v_cvt_pk_bf16_f32
    const f32x2_v v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_v));
}
__device__ __forceinline__ float bf16_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

template <typename T> struct Payload;

template <> struct Payload<float> {
    static constexpr bool is_bf16 = false;
    static constexpr int bytes = 4;
    static __device__ __forceinline__ float ld(const float *p) { return *p; }
    static __device__ __forceinline__ void st(float *p, float v) { *p = v; }
    // 4 consecutive elements, p aligned to 4 elements
    static __device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
    static __device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
};

template <> struct Payload<bf16_t> {
    static constexpr bool is_bf16 = true;
    static constexpr int bytes = 2;
    static __device__ __forceinline__ float ld(const bf16_t *p) { return __builtin_bit_cast(float, (uint32_t)p->bits << 16); }
    static __device__ __forceinline__ void st(bf16_t *p, float v) { p->bits = (uint16_t)(pack_bf16x2(v, 0.f) & 0xffffu); }
    static __device__ __forceinline__ float4 ld4(const bf16_t *p) {
        const uint2 u = *reinterpret_cast<const uint2 *>(p);
        return make_float4(bf16_lo(u.x), bf16_hi(u.x), bf16_lo(u.y), bf16_hi(u.y));
    }
    static __device__ __forceinline__ void st4(bf16_t *p, float4 v) {
        *reinterpret_cast<uint2 *>(p) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    }
};

}  // namespace mgar
