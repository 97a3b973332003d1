// stem_conv.hip -- the first convolution of Inception-I3D (Conv3d_1a_7x7: 3 -> 64 channels, 7x7x7, stride 2, TF-"same"
// padding) as a direct implicit GEMM on the exact-fp32 MFMA, for gfx950.
//
// Reference: model/backbone.py:134-206 (Unit3D: dynamic "same" padding + nn.Conv3d(bias=False)), instance :305-307
// (``Conv3d_1a_7x7``, kernel [7,7,7], stride (2,2,2)); the convolution itself is torch / cuDNN there (MIOpen here).
//
// Why a kernel: with C_in = 3 the library's implicit-GEMM convolution runs at 39 % of the fp32 MFMA peak and is the
// largest single kernel of a training step (31.6 of 268 ms at config c3: 29.1 ms CK kernel + NCDHW <-> NDHWC transposes
// of its 1.3 GB input and 4.7 GB output + the F.pad copy of the asymmetric "same" padding).  Here:
//   * D[co][position] = sum_k W[co][k] * patch[k][position] with M = 64 output channels as the MFMA rows and 32
//     consecutive output columns (wo) as the MFMA columns, so every accumulator register is 128 contiguous bytes of the
//     NCDHW output: no transposes, coalesced stores;
//   * K = 3 * 7 * 7 * 7 = 1029 is walked as 21 (kt, c_in) slabs of 49 (kh, kw) taps (padded to 50 = 25 k-steps of
//     v_mfma_f32_32x32x2_f32); a slab needs a (21 rows x 69 columns) input patch for the workgroup's 8 x 32 output tile
//     and a 50 x 64 weight block: both are staged in LDS, double-buffered, the next slab's global loads in flight under
//     the current slab's 100 MFMAs per wave; slabs whose input plane lies in the temporal padding are skipped;
//   * the input patch is stored de-interleaved by column parity, so the stride-2 reads of a stride-2 convolution
//     (column 2 wo + kw) are unit-stride in LDS: conflict-free;
//   * padding is handled by the patch loader (zeros): no padded copy of the input.
// Work per workgroup: 256 outputs x 64 channels; 4 waves x (2 channel blocks x 2 output rows) accumulators.
// fp32 accumulation in tap order (kt, c_in, kh, kw): the result is an fp32 FMA chain like the library's, in a different
// order (parity test: 1e-5 relative).  T = payload type of input / output (float or bf16_t); weights fp32.
#include "common.hpp"
#include "payload.hpp"

namespace mgar {

typedef float __attribute__((ext_vector_type(16))) f32x16;

constexpr int SCV_K = 7, SCV_S = 2, SCV_CIN = 3, SCV_COUT = 64;
constexpr int SCV_TH = 8, SCV_TW = 32;                                   // output tile (ho x wo) per workgroup
constexpr int SCV_IH = SCV_S * (SCV_TH - 1) + SCV_K;                      // 21 input rows
constexpr int SCV_IW = SCV_S * (SCV_TW - 1) + SCV_K;                      // 69 input columns
constexpr int SCV_IWH = (SCV_IW + 1) / 2;                                 // 35 per column parity
constexpr int SCV_TAPS = 50;                                              // 49 (kh, kw) taps + 1 zero tap
constexpr int SCV_IN_FLOATS = SCV_IH * 2 * SCV_IWH;                       // 1470
constexpr int SCV_W_FLOATS = SCV_TAPS * SCV_COUT;                         // 3200
constexpr int SCV_IN_PER_THREAD = (SCV_IH * SCV_IW + 255) / 256;          // 6
constexpr int SCV_W_PER_THREAD = SCV_W_FLOATS / 4 / 256 + 1;              // 4 float4 (800 float4 over 256 threads)

struct StemArgs {
    int N, T, H, W, To, Ho, Wo, pt, ph, pw;   // front paddings of the TF "same" mode
};

// wp: weights packed as [kt][c_in][tap (50)][co (64)], tap = kh * 7 + kw, tap 49 = 0
template <typename T>
__global__ __launch_bounds__(256, 2) void stem_conv3d_kernel(const T *__restrict__ x, const float *__restrict__ wp, StemArgs a,
                                                             T *__restrict__ y) {
    __shared__ float s_in[2][SCV_IN_FLOATS];
    __shared__ __attribute__((aligned(16))) float s_w[2][SCV_W_FLOATS];
    const int tiles_w = (a.Wo + SCV_TW - 1) / SCV_TW;
    const int bx = blockIdx.x % tiles_w, by = blockIdx.x / tiles_w;
    const int to = blockIdx.y, n = blockIdx.z;
    const int wo0 = bx * SCV_TW, ho0 = by * SCV_TH;
    const int hbase = SCV_S * ho0 - a.ph, wbase = SCV_S * wo0 - a.pw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l = lane & 31, h = lane >> 5;

    // the (kt, c) slabs whose input plane exists: kt in [kt_lo, kt_hi)
    const int t0 = SCV_S * to - a.pt;
    const int kt_lo = max(0, -t0), kt_hi = min(SCV_K, a.T - t0);
    const int nslab = max(0, kt_hi - kt_lo) * SCV_CIN;

    f32x16 acc[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

    float pin[SCV_IN_PER_THREAD];
    float4 pw4[SCV_W_PER_THREAD];
    auto fetch = [&](int slab) {                       // global -> registers
        const int kt = kt_lo + slab / SCV_CIN, c = slab % SCV_CIN;
        const T *plane = x + (((size_t)n * SCV_CIN + c) * a.T + (t0 + kt)) * a.H * a.W;
#pragma unroll
        for (int u = 0; u < SCV_IN_PER_THREAD; ++u) {
            const int e = threadIdx.x + u * 256;
            const int ir = e / SCV_IW, ic = e - ir * SCV_IW;
            const int hh = hbase + ir, ww = wbase + ic;
            pin[u] = (e < SCV_IH * SCV_IW && hh >= 0 && hh < a.H && ww >= 0 && ww < a.W) ? Payload<T>::ld(plane + (size_t)hh * a.W + ww) : 0.f;
        }
        const float4 *wsrc = reinterpret_cast<const float4 *>(wp + (size_t)(kt * SCV_CIN + c) * SCV_W_FLOATS);
#pragma unroll
        for (int u = 0; u < SCV_W_PER_THREAD; ++u) {
            const int e = threadIdx.x + u * 256;
            pw4[u] = e < SCV_W_FLOATS / 4 ? wsrc[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stash = [&](int buf) {                        // registers -> LDS (input de-interleaved by column parity)
#pragma unroll
        for (int u = 0; u < SCV_IN_PER_THREAD; ++u) {
            const int e = threadIdx.x + u * 256;
            const int ir = e / SCV_IW, ic = e - ir * SCV_IW;
            if (e < SCV_IH * SCV_IW) s_in[buf][(ir * 2 + (ic & 1)) * SCV_IWH + (ic >> 1)] = pin[u];
        }
#pragma unroll
        for (int u = 0; u < SCV_W_PER_THREAD; ++u) {
            const int e = threadIdx.x + u * 256;
            if (e < SCV_W_FLOATS / 4) reinterpret_cast<float4 *>(s_w[buf])[e] = pw4[u];
        }
    };

    if (nslab > 0) {
        fetch(0);
        stash(0);
    }
    __syncthreads();
    for (int slab = 0; slab < nslab; ++slab) {
        const int buf = slab & 1;
        if (slab + 1 < nslab) fetch(slab + 1);         // in flight under the MFMAs below
        const float *ti = s_in[buf], *tw = s_w[buf];
#pragma unroll 5
        for (int s = 0; s < SCV_TAPS / 2; ++s) {
            const int tap = 2 * s + h;
            int kh = tap / SCV_K, kw = tap - kh * SCV_K;
            if (tap >= SCV_K * SCV_K) { kh = 0; kw = 0; }                  // the zero tap: any valid address
            const float wa0 = tw[tap * SCV_COUT + l], wa1 = tw[tap * SCV_COUT + 32 + l];
            const int col = (kw & 1) * SCV_IWH + l + (kw >> 1);
            const float b0 = ti[(SCV_S * (2 * wave + 0) + kh) * 2 * SCV_IWH + col];
            const float b1 = ti[(SCV_S * (2 * wave + 1) + kh) * 2 * SCV_IWH + col];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa0, b0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa1, b0, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa0, b1, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa1, b1, acc[1][1], 0, 0, 0);
        }
        if (slab + 1 < nslab) stash(buf ^ 1);          // the other buffer: its last readers passed the barrier of slab - 1
        __syncthreads();
    }
    // D[row = co][col = l = wo]: every accumulator register is 32 consecutive output columns of one channel
    const int wo = wo0 + l;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int ho = ho0 + 2 * wave + nb;
        if (ho >= a.Ho || wo >= a.Wo) continue;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                Payload<T>::st(y + ((((size_t)n * SCV_COUT + co) * a.To + to) * a.Ho + ho) * a.Wo + wo, acc[mb][nb][r]);
            }
    }
}

// (64, 3, 7, 7, 7) -> [kt][c][tap 50][co 64]
__global__ void stem_pack_weights_kernel(const float *__restrict__ w, float *__restrict__ wp) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= SCV_K * SCV_CIN * SCV_W_FLOATS) return;
    const int co = e % SCV_COUT, tap = (e / SCV_COUT) % SCV_TAPS, c = (e / SCV_W_FLOATS) % SCV_CIN, kt = e / (SCV_W_FLOATS * SCV_CIN);
    float v = 0.f;
    if (tap < SCV_K * SCV_K) {
        const int kh = tap / SCV_K, kw = tap - kh * SCV_K;
        v = w[((((size_t)co * SCV_CIN + c) * SCV_K + kt) * SCV_K + kh) * SCV_K + kw];
    }
    wp[e] = v;
}

// ---- bf16 payloads: the same convolution on the bf16 MFMA -------------------------------------------------------------------
// (configurations c2 / c5: feature payloads AND GEMM operands bf16, fp32 accumulation.)  v_mfma_f32_32x32x16_bf16 takes 8
// consecutive k per lane and half-wave: with k ordered (kh, kw) and kw padded 7 -> 8, a lane's B fragment is 8 CONSECUTIVE input
// columns 2 wo .. 2 wo + 7 of input row 2 ho + kh -- four aligned 4-byte LDS reads of the bf16 patch, no de-interleaving -- and
// its A fragment 16 contiguous bytes of the weights packed [k-step][co][kh parity][kw].  A (kt, c_in) slab is 4 k-steps (kh pairs
// (0,1) (2,3) (4,5) (6,-)) x 4 accumulators = 16 MFMAs per wave instead of 100 fp32 ones; same tile, same double-buffered
// pipeline, same epilogue.  Weights are rounded to bf16 once per call (what a bf16 convolution does).
typedef __bf16 __attribute__((ext_vector_type(8))) bf16x8;
typedef unsigned int __attribute__((ext_vector_type(4))) u32x4;
constexpr int SCB_COLS = 72;                                              // patch row pitch in bf16 (69 columns + pad)
constexpr int SCB_IN_HALVES = SCV_IH * SCB_COLS;                          // 1512
constexpr int SCB_W_HALVES = 4 * SCV_COUT * 16;                           // 4096 per slab: [step][co][h][8]

__global__ __launch_bounds__(256, 2) void stem_conv3d_bf16_kernel(const bf16_t *__restrict__ x, const u32x4 *__restrict__ wp, StemArgs a,
                                                                  bf16_t *__restrict__ y) {
    __shared__ __attribute__((aligned(16))) uint16_t s_in[2][SCB_IN_HALVES];
    __shared__ __attribute__((aligned(16))) uint16_t s_w[2][SCB_W_HALVES];
    const int tiles_w = (a.Wo + SCV_TW - 1) / SCV_TW;
    const int bx = blockIdx.x % tiles_w, by = blockIdx.x / tiles_w;
    const int to = blockIdx.y, n = blockIdx.z;
    const int wo0 = bx * SCV_TW, ho0 = by * SCV_TH;
    const int hbase = SCV_S * ho0 - a.ph, wbase = SCV_S * wo0 - a.pw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l = lane & 31, h = lane >> 5;
    const int t0 = SCV_S * to - a.pt;
    const int kt_lo = max(0, -t0), kt_hi = min(SCV_K, a.T - t0);
    const int nslab = max(0, kt_hi - kt_lo) * SCV_CIN;
    f32x16 acc[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
    for (int e = threadIdx.x; e < 2 * SCB_IN_HALVES; e += 256) s_in[0][e] = 0;   // the pad columns of both buffers stay zero

    uint32_t pin[SCV_IN_PER_THREAD];                      // one bf16 each (kept in 32-bit registers)
    u32x4 pwa, pwb;
    auto fetch = [&](int slab) {
        const int kt = kt_lo + slab / SCV_CIN, c = slab % SCV_CIN;
        const bf16_t *plane = x + (((size_t)n * SCV_CIN + c) * a.T + (t0 + kt)) * a.H * a.W;
#pragma unroll
        for (int u = 0; u < SCV_IN_PER_THREAD; ++u) {
            const int e = threadIdx.x + u * 256;
            const int ir = e / SCV_IW, ic = e - ir * SCV_IW;
            const int hh = hbase + ir, ww = wbase + ic;
            pin[u] = (e < SCV_IH * SCV_IW && hh >= 0 && hh < a.H && ww >= 0 && ww < a.W) ? (uint32_t)plane[(size_t)hh * a.W + ww].bits : 0u;
        }
        const u32x4 *wsrc = wp + (size_t)(kt * SCV_CIN + c) * (SCB_W_HALVES / 8);
        pwa = wsrc[threadIdx.x];
        pwb = wsrc[threadIdx.x + 256];
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < SCV_IN_PER_THREAD; ++u) {
            const int e = threadIdx.x + u * 256;
            const int ir = e / SCV_IW, ic = e - ir * SCV_IW;
            if (e < SCV_IH * SCV_IW) s_in[buf][ir * SCB_COLS + ic] = (uint16_t)pin[u];
        }
        reinterpret_cast<u32x4 *>(s_w[buf])[threadIdx.x] = pwa;
        reinterpret_cast<u32x4 *>(s_w[buf])[threadIdx.x + 256] = pwb;
    };
    __syncthreads();
    if (nslab > 0) {
        fetch(0);
        stash(0);
    }
    __syncthreads();
    for (int slab = 0; slab < nslab; ++slab) {
        const int buf = slab & 1;
        if (slab + 1 < nslab) fetch(slab + 1);
        const uint16_t *ti = s_in[buf], *tw = s_w[buf];
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int kh = min(2 * st + h, SCV_K - 1);                     // kh = 7 carries zero weights: any valid row
            const u32x4 wa0 = *reinterpret_cast<const u32x4 *>(tw + ((st * SCV_COUT + l) * 2 + h) * 8);
            const u32x4 wa1 = *reinterpret_cast<const u32x4 *>(tw + ((st * SCV_COUT + 32 + l) * 2 + h) * 8);
            const uint32_t *r0 = reinterpret_cast<const uint32_t *>(ti + (SCV_S * (2 * wave + 0) + kh) * SCB_COLS) + l;
            const uint32_t *r1 = reinterpret_cast<const uint32_t *>(ti + (SCV_S * (2 * wave + 1) + kh) * SCB_COLS) + l;
            const u32x4 b0 = {r0[0], r0[1], r0[2], r0[3]};
            const u32x4 b1 = {r1[0], r1[1], r1[2], r1[3]};
            const bf16x8 fa0 = __builtin_bit_cast(bf16x8, wa0), fa1 = __builtin_bit_cast(bf16x8, wa1);
            const bf16x8 fb0 = __builtin_bit_cast(bf16x8, b0), fb1 = __builtin_bit_cast(bf16x8, b1);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb0, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb1, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb1, acc[1][1], 0, 0, 0);
        }
        if (slab + 1 < nslab) stash(buf ^ 1);
        __syncthreads();
    }
    const int wo = wo0 + l;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int ho = ho0 + 2 * wave + nb;
        if (ho >= a.Ho || wo >= a.Wo) continue;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                Payload<bf16_t>::st(y + ((((size_t)n * SCV_COUT + co) * a.To + to) * a.Ho + ho) * a.Wo + wo, acc[mb][nb][r]);
            }
    }
}

// (64, 3, 7, 7, 7) fp32 -> bf16 [kt][c][step 4][co 64][h 2][kw 8]: kh = 2 step + h, zero for kh = 7 or kw = 7
__global__ void stem_pack_weights_bf16_kernel(const float *__restrict__ w, uint16_t *__restrict__ wp) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= SCV_K * SCV_CIN * SCB_W_HALVES) return;
    const int kw = e & 7, hh = (e >> 3) & 1, co = (e >> 4) % SCV_COUT, st = (e / (16 * SCV_COUT)) & 3;
    const int c = (e / SCB_W_HALVES) % SCV_CIN, kt = e / (SCB_W_HALVES * SCV_CIN);
    const int kh = 2 * st + hh;
    float v = 0.f;
    if (kh < SCV_K && kw < SCV_K) v = w[((((size_t)co * SCV_CIN + c) * SCV_K + kt) * SCV_K + kh) * SCV_K + kw];
    wp[e] = (uint16_t)(pack_bf16x2(v, 0.f) & 0xffffu);
}

// ---- fp32, W % 4 == 0: minimal filtering along W ------------------------------------------------------------------------
// The exact-fp32 MFMA is the bound of this kernel (one 32x32x2 instruction = 64 cycles for one pair of operand registers; LDS,
// VALU and L2 idle), so multiply-accumulates are what to save.  A stride-2, 7-tap row convolution splits by column parity
// into a 3-tap stride-1 convolution of the odd columns (taps w1, w3, w5) and a 4-tap one of the even columns (w0, w2, w4, w6)
// (front padding 2 for even W: input column 2 wo + kw - 2).  For an output PAIR (2p, 2p+1) each 3-tap part is Winograd
// F(2, 3) -- 4 products instead of 6, transforms made of additions and one halving only, as in csrc/conv3d_wino.hip -- and
// the fourth even tap is taken directly; its two products ride in the accumulators of components 0 and 3 (y[2p] = m0 + m1 +
// m2, y[2p+1] = m1 - m2 - m3: m0 enters only y[2p], m3 only y[2p+1]).  10 products per (kt, c, kh) and output pair instead
// of 14: 36 instead of 50 MFMAs per slab, channel block and 64 outputs.
//   d^o = odd-column samples  x[2 (2p - 1 + j) + 1 - 2 + ...], d^e = even-column samples, j = 0..3 (d^e also j = 4):
//   m0 += G0o (d0o - d2o) + G0e (d0e - d2e) + w6 d3e          m1 += G1o (d1o + d2o) + G1e (d1e + d2e)
//   m3 += G3o (d1o - d3o) + G3e (d1e - d3e) - w6 d4e          m2 += G2o (d2o - d1o) + G2e (d2e - d1e)
//   G = (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2) of (w1, w3, w5) resp. (w0, w2, w4).
// Tile: 4 output rows x 64 output columns x 64 channels per workgroup; a wave = one row = 32 pairs x 2 channel blocks x 4
// components (128 accumulator registers).  The two k of an MFMA are the two parities (half-wave 0 reads the odd-column patch,
// half-wave 1 the even-column one); the direct taps of two kh share an instruction.  Per (kt, c) slab: 13 x 134 input patch,
// de-interleaved by parity, and a 4096-float weight block in LDS, double-buffered; global loads of the next slab are issued
// before the 72 MFMAs of the current one and stored after them (no guards in the staging loops: a guarded LDS store lets the
// compiler sink the load down to it); operands of step s + 1 are read from LDS before the MFMAs of step s.
typedef float __attribute__((ext_vector_type(4))) f32x4;
typedef float __attribute__((ext_vector_type(2))) f32x2;
constexpr int SW_TH = 4, SW_TWO = 64;                                     // output tile: rows x columns
constexpr int SW_IH = SCV_S * (SW_TH - 1) + SCV_K;                        // 13 input rows
constexpr int SW_ICOLS = 134, SW_PITCH = 68;                              // patch columns (67 per parity), LDS pitch per parity row
constexpr int SW_IN_PER_THREAD = (SW_IH * SW_ICOLS + 255) / 256;          // 7
constexpr int SW_IN_FLOATS = (SW_IH + 1) * 2 * SW_PITCH;                  // 14 rows allocated: the padded staging slots land in row 13
constexpr int SW_W_MAIN = SCV_K * 2 * SCV_COUT * 4;                       // [kh][parity][co][t] = 3584
constexpr int SW_W_FLOATS = SW_W_MAIN + 8 * SCV_COUT;                     // + direct tap [kh 8 (7 + a zero row)][co] = 4096
constexpr int SW_W_PER_THREAD = SW_W_FLOATS / 4 / 256;                    // 4 x f32x4

__global__ __launch_bounds__(256, 2) void stem_conv3d_wino_kernel(const float *__restrict__ x, const float *__restrict__ wp, StemArgs a,
                                                                  float *__restrict__ y) {
    __shared__ __attribute__((aligned(16))) float s_in[2][SW_IN_FLOATS];
    __shared__ __attribute__((aligned(16))) float s_w[2][SW_W_FLOATS];
    const int tiles_w = (a.Wo + SW_TWO - 1) / SW_TWO;
    const int bx = blockIdx.x % tiles_w, by = blockIdx.x / tiles_w;
    const int to = blockIdx.y, n = blockIdx.z;
    const int wo0 = bx * SW_TWO, ho0 = by * SW_TH;
    const int hbase = SCV_S * ho0 - a.ph, cbase = SCV_S * wo0 - 2;        // first input row / column of the patch (pw = 2)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = lane & 31, h = lane >> 5;

    const int t0 = SCV_S * to - a.pt;
    const int kt_lo = max(0, -t0), kt_hi = min(SCV_K, a.T - t0);
    const int nslab = max(0, kt_hi - kt_lo) * SCV_CIN;

    // staging slots of this thread: offset inside an input plane (-1: padding -> zero) and LDS position
    int off[SW_IN_PER_THREAD], pos[SW_IN_PER_THREAD];
#pragma unroll
    for (int u = 0; u < SW_IN_PER_THREAD; ++u) {
        const int e = threadIdx.x + u * 256;
        const int ir = e / SW_ICOLS, ic = e - ir * SW_ICOLS;
        const int hh = hbase + ir, ww = cbase + ic;
        off[u] = (ir < SW_IH && hh >= 0 && hh < a.H && ww >= 0 && ww < a.W) ? 4 * (hh * a.W + ww) : -1;   // byte offset; -1: reads 0
        pos[u] = (ir * 2 + 1 - (ic & 1)) * SW_PITCH + (ic >> 1);           // parity row 0 = odd columns, 1 = even columns
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][t][r] = 0.f;

    float pin[SW_IN_PER_THREAD];
    f32x4 pw4[SW_W_PER_THREAD];
    auto fetch = [&](int slab) {                       // global -> registers; nothing here waits for a load
        const int kt = kt_lo + slab / SCV_CIN, c = slab % SCV_CIN;
        // the plane through a raw buffer resource: the padding's offset (-1) is out of range and reads 0 -- no select, no 64-bit
        // address arithmetic in the staging path
        const __amdgpu_buffer_rsrc_t plane = uniform_buffer(x + (((size_t)n * SCV_CIN + c) * a.T + (t0 + kt)) * a.H * a.W, 4 * a.H * a.W);
#pragma unroll
        for (int u = 0; u < SW_IN_PER_THREAD; ++u) pin[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(plane, off[u], 0, 0));
        const f32x4 *wsrc = reinterpret_cast<const f32x4 *>(wp + (size_t)(kt * SCV_CIN + c) * SW_W_FLOATS);
#pragma unroll
        for (int u = 0; u < SW_W_PER_THREAD; ++u) pw4[u] = wsrc[threadIdx.x + u * 256];
    };
    auto stash = [&](int buf) {                        // registers -> LDS
#pragma unroll
        for (int u = 0; u < SW_IN_PER_THREAD; ++u) s_in[buf][pos[u]] = pin[u];
#pragma unroll
        for (int u = 0; u < SW_W_PER_THREAD; ++u) reinterpret_cast<f32x4 *>(s_w[buf])[threadIdx.x + u * 256] = pw4[u];
    };

    if (nslab > 0) {
        fetch(0);
        stash(0);
    }
    __syncthreads();
    // lane constants: its parity row of patch row 2 * wave (+ kh), at pair l; the even-column row for the direct taps
    const int in_lane = (SCV_S * wave * 2 + h) * SW_PITCH + 2 * l;
    const int ex_lane = (SCV_S * wave * 2 + 1) * SW_PITCH + 2 * l + 3;
    for (int slab = 0; slab < nslab; ++slab) {
        const int buf = slab & 1;
        fetch(min(slab + 1, nslab - 1));               // in flight under the MFMAs below (the last slab re-loads itself: no branch)
        __builtin_amdgcn_sched_barrier(0);
        const float *ti = s_in[buf] + in_lane, *te = s_in[buf] + ex_lane;
        const f32x4 *tw = reinterpret_cast<const f32x4 *>(s_w[buf]) + h * SCV_COUT + l;
        const float *tx = s_w[buf] + SW_W_MAIN + h * SCV_COUT + l;
        f32x2 xa[2], xb[2];
        f32x4 g[2][2];
        auto read_main = [&](int kh, int slot) {
            const float *p = ti + kh * 2 * SW_PITCH;
            xa[slot] = *reinterpret_cast<const f32x2 *>(p);
            xb[slot] = *reinterpret_cast<const f32x2 *>(p + 2);
            g[slot][0] = tw[kh * 2 * SCV_COUT];
            g[slot][1] = tw[kh * 2 * SCV_COUT + 32];
        };
        float e3[2], e4[2], gx[2][2];
        auto read_extra = [&](int i, int slot) {        // direct taps of kh = 2 i + h (kh = 7: zero weights, row 6 again)
            const float *p = te + min(2 * i + h, SCV_K - 1) * 2 * SW_PITCH;
            e3[slot] = p[0];
            e4[slot] = p[1];
            gx[slot][0] = tx[i * 2 * SCV_COUT];
            gx[slot][1] = tx[i * 2 * SCV_COUT + 32];
        };
        read_main(0, 0);
#pragma unroll
        for (int kh = 0; kh < SCV_K; ++kh) {
            const int slot = kh & 1;
            if (kh + 1 < SCV_K) read_main(kh + 1, slot ^ 1);
            else read_extra(0, 0);
            const float u0 = xa[slot].x - xb[slot].x, u1 = xa[slot].y + xb[slot].x, u2 = xb[slot].x - xa[slot].y, u3 = xa[slot].y - xb[slot].y;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                acc[mb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[slot][mb].x, u0, acc[mb][0], 0, 0, 0);
                acc[mb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[slot][mb].y, u1, acc[mb][1], 0, 0, 0);
                acc[mb][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[slot][mb].z, u2, acc[mb][2], 0, 0, 0);
                acc[mb][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[slot][mb].w, u3, acc[mb][3], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int slot = i & 1;
            if (i + 1 < 4) read_extra(i + 1, slot ^ 1);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                acc[mb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(gx[slot][mb], e3[slot], acc[mb][0], 0, 0, 0);
                acc[mb][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(gx[slot][mb], -e4[slot], acc[mb][3], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        stash(buf ^ 1);                                // the other buffer: its last readers passed the barrier of slab - 1
        __syncthreads();
    }
    // output transform + store: register r of a lane = (co = 32 mb + (r & 3) + 8 (r >> 2) + 4 h, pair l)
    const int ho = ho0 + wave, wo = wo0 + 2 * l;
    if (ho < a.Ho && wo < a.Wo) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float m0 = acc[mb][0][r], m1 = acc[mb][1][r], m2 = acc[mb][2][r], m3 = acc[mb][3][r];
                f32x2 o;
                o.x = (m0 + m1) + m2;
                o.y = (m1 - m2) - m3;
                *reinterpret_cast<f32x2 *>(y + ((((size_t)n * SCV_COUT + co) * a.To + to) * a.Ho + ho) * a.Wo + wo) = o;
            }
    }
}

// (64, 3, 7, 7, 7) -> [kt][c][ [kh 7][parity 2: odd taps (w1,w3,w5) / even taps (w0,w2,w4)][co 64][t 4] | [kh 8][co 64]: w6, row 7 = 0 ]
__global__ void stem_pack_weights_wino_kernel(const float *__restrict__ w, float *__restrict__ wp) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= SCV_K * SCV_CIN * SW_W_FLOATS) return;
    const int slab = e / SW_W_FLOATS, r = e - slab * SW_W_FLOATS;
    const int kt = slab / SCV_CIN, c = slab - kt * SCV_CIN;
    float v = 0.f;
    if (r < SW_W_MAIN) {
        const int t = r & 3, co = (r >> 2) % SCV_COUT, par = (r / (4 * SCV_COUT)) & 1, kh = r / (8 * SCV_COUT);
        const float *src = w + ((((size_t)co * SCV_CIN + c) * SCV_K + kt) * SCV_K + kh) * SCV_K;
        const float g0 = par ? src[0] : src[1], g1 = par ? src[2] : src[3], g2 = par ? src[4] : src[5];
        v = t == 0 ? g0 : (t == 3 ? g2 : (t == 1 ? 0.5f * ((g0 + g2) + g1) : 0.5f * ((g0 + g2) - g1)));
    } else {
        const int q = r - SW_W_MAIN, co = q % SCV_COUT, kh = q / SCV_COUT;
        if (kh < SCV_K) v = w[((((size_t)co * SCV_CIN + c) * SCV_K + kt) * SCV_K + kh) * SCV_K + 6];
    }
    wp[e] = v;
}

static int g_stem_minimal_filtering = 1;

template <typename T>
static int stem_conv_impl(const T *x, int N, int Tn, int H, int W, const float *w, float *w_packed, T *y, void *stream) {
    MGAR_REQUIRE(N >= 0 && Tn > 0 && H > 0 && W > 0, "stem_conv3d_fwd: bad sizes");
    if (N == 0) return MGAR_OK;
    MGAR_REQUIRE(x && w && w_packed && y, "stem_conv3d_fwd: null pointer");
    auto front = [](int size) {   // TF "same" for kernel 7, stride 2 (model/backbone.py:168-172)
        const int total = size % SCV_S == 0 ? SCV_K - SCV_S : SCV_K - size % SCV_S;
        return (total > 0 ? total : 0) / 2;
    };
    StemArgs a{N, Tn, H, W, (Tn + 1) / 2, (H + 1) / 2, (W + 1) / 2, front(Tn), front(H), front(W)};
    MGAR_REQUIRE(a.To <= 65535 && N <= 65535, "stem_conv3d_fwd: T or N too large");
    MGAR_REQUIRE((long long)4 * H * W < (1ll << 31), "stem_conv3d_fwd: frame too large for 32-bit plane offsets");
    hipStream_t st = (hipStream_t)stream;
    const int tiles = ((a.Wo + SCV_TW - 1) / SCV_TW) * ((a.Ho + SCV_TH - 1) / SCV_TH);
    const double outs = (double)N * a.To * a.Ho * a.Wo;
    if (Payload<T>::is_bf16) {                                     // bf16 payloads: operands bf16 on the bf16 MFMA
        hipLaunchKernelGGL(stem_pack_weights_bf16_kernel, dim3(ceil_div(SCV_K * SCV_CIN * SCB_W_HALVES, 256)), dim3(256), 0, st, w,
                           reinterpret_cast<uint16_t *>(w_packed));
        KtScope kt(KT_STEM_CONV, st, (double)sizeof(T) * ((double)N * SCV_CIN * Tn * H * W + outs * SCV_COUT),
                   2.0 * outs * SCV_COUT * SCV_CIN * SCV_K * SCV_K * SCV_K);
        hipLaunchKernelGGL(stem_conv3d_bf16_kernel, dim3(tiles, a.To, N), dim3(256), 0, st, reinterpret_cast<const bf16_t *>(x),
                           reinterpret_cast<const u32x4 *>(w_packed), a, reinterpret_cast<bf16_t *>(y));
        return check_launch("stem_conv3d_fwd: launch failed");
    }
    if (!Payload<T>::is_bf16 && g_stem_minimal_filtering && W % 4 == 0) {      // even W (front padding 2) and whole output pairs
        hipLaunchKernelGGL(stem_pack_weights_wino_kernel, dim3(ceil_div(SCV_K * SCV_CIN * SW_W_FLOATS, 256)), dim3(256), 0, st, w, w_packed);
        const int wtiles = ((a.Wo + SW_TWO - 1) / SW_TWO) * ((a.Ho + SW_TH - 1) / SW_TH);
        // flops: the MFMA work issued = 10/14 of the direct convolution's (bench.py prices the kernel against the MFMA peak)
        KtScope kt(KT_STEM_CONV, st, (double)sizeof(T) * ((double)N * SCV_CIN * Tn * H * W + outs * SCV_COUT),
                   2.0 * outs * SCV_COUT * SCV_CIN * SCV_K * SCV_K * 5.0);
        hipLaunchKernelGGL(stem_conv3d_wino_kernel, dim3(wtiles, a.To, N), dim3(256), 0, st, reinterpret_cast<const float *>(x),
                           (const float *)w_packed, a, reinterpret_cast<float *>(y));
        return check_launch("stem_conv3d_fwd: launch failed");
    }
    hipLaunchKernelGGL(stem_pack_weights_kernel, dim3(ceil_div(SCV_K * SCV_CIN * SCV_W_FLOATS, 256)), dim3(256), 0, st, w, w_packed);
    {
        KtScope kt(KT_STEM_CONV, st, (double)sizeof(T) * ((double)N * SCV_CIN * Tn * H * W + outs * SCV_COUT),
                   2.0 * outs * SCV_COUT * SCV_CIN * SCV_K * SCV_K * SCV_K);
        hipLaunchKernelGGL(stem_conv3d_kernel<T>, dim3(tiles, a.To, N), dim3(256), 0, st, x, (const float *)w_packed, a, y);
    }
    return check_launch("stem_conv3d_fwd: launch failed");
}

}  // namespace mgar

using namespace mgar;

#define SCV_API extern "C" __attribute__((visibility("default")))

// x (N, 3, T, H, W), w (64, 3, 7, 7, 7) -> y (N, 64, ceil(T/2), ceil(H/2), ceil(W/2)); w_packed: caller-allocated scratch of
// mgar_stem_conv3d_workspace_floats() floats (the weights re-laid out for the kernel, rewritten on every call).
SCV_API int mgar_stem_conv3d_workspace_floats(void) { return SCV_K * SCV_CIN * (SW_W_FLOATS > SCV_W_FLOATS ? SW_W_FLOATS : SCV_W_FLOATS); }
// A/B switch (tests, tools): 0 = the direct kernel for fp32 too; default 1 = minimal filtering along W where W % 4 == 0
SCV_API int mgar_stem_conv3d_set_minimal_filtering(int on) { g_stem_minimal_filtering = on ? 1 : 0; return MGAR_OK; }
SCV_API int mgar_stem_conv3d_fwd(const float *x, int N, int T, int H, int W, const float *w, float *w_packed, float *y, void *stream) {
    return stem_conv_impl<float>(x, N, T, H, W, w, w_packed, y, stream);
}
SCV_API int mgar_stem_conv3d_fwd_bf16(const void *x, int N, int T, int H, int W, const float *w, float *w_packed, void *y, void *stream) {
    return stem_conv_impl<bf16_t>((const bf16_t *)x, N, T, H, W, w, w_packed, (bf16_t *)y, stream);
}
