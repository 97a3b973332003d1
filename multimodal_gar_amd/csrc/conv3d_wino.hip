// conv3d_wino.hip -- the 3x3x3, stride-1, "same"-padded convolutions of Inception-I3D (Conv3d_2c_3x3 and the Branch_1 /
// Branch_2 ``Conv3d_0b_3x3`` units of every Mixed block) on the exact-fp32 MFMA of gfx950, NCDHW in and out, with the
// Winograd F(2, 3) minimal-filtering identity along W.
//
// Reference: model/backbone.py:134-206 (Unit3D: dynamic "same" padding + nn.Conv3d(bias=False)); instances :311-312
// (Conv3d_2c_3x3, 64 -> 192) and :215-236 (InceptionModule b1b / b2b).  The convolution itself is torch / cuDNN there and
// MIOpen in a plain PyTorch-ROCm run.
//
// Why a kernel: the nine 3x3x3 convolutions are 52 ms of a 213 ms training step at config c3 (the largest, 64 -> 192 over
// 8 x 8 x 180 x 320, 23.5 ms at 0.66 of the fp32 MFMA peak, plus the NCDHW <-> NDHWC adapters MIOpen wraps around its
// kernels).  fp32 on the matrix cores is v_mfma_f32_32x32x2_f32: 4096 FLOP in 64 cycles per SIMD, i.e. one pair of operand
// registers per 64 cycles -- the MFMA pipe is the bound and everything else (LDS, VALU, L2) idles.  That is exactly where
// minimal filtering pays: for two adjacent outputs of a row,
//     y[2p]   = m0 + m1 + m2,   y[2p+1] = m1 - m2 - m3,      m_t = sum_{ci,kd,kh} G_t[co][ci][kd][kh] * U_t[ci][d+kd][h+kh][p]
//     U = (x0 - x2, x1 + x2, x2 - x1, x1 - x3) of the four inputs x[2p-1 .. 2p+2]
//     G = (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2) of the three taps along W
// so 4 multiply-accumulates replace 6: two thirds of the MFMA work of the direct convolution, for four VALU
// additions per four MFMAs.  Only additions, subtractions and one halving enter the transforms (no large constants as in
// F(4, 3)); the measured error against an fp64 convolution is that of the library's direct kernel (0.1 - 1.0e-6 of the output scale for both)
// (tests/test_conv3d_wino_gpu.py writes both to gpurun_out/parity_margins.txt).
//
// Layout of the GEMMs: D[co][pair] -- the MFMA rows are 32 output channels, the columns 32 output PAIRS (a PW x BH patch of
// pairs, PW * BH = 32), k runs over (c_in, kd, kh) with the two k of one instruction being the two input channels of a
// channel pair.  A wave owns 2 channel blocks x 1 pair block x 4 Winograd components = 8 accumulators (128 registers); the 4
// waves of a workgroup take 4 pair blocks stacked along H and share the transformed filter.  Per channel pair the workgroup
// stages in LDS the 3 x (TH + 2) x (TW + 2) input halo tile of both channels (zeros outside the volume: the padding costs
// nothing) and the 9 x 2 x 64 x 4 transformed filter block, double-buffered, the next pair's global loads in flight under
// the current pair's 72 MFMAs per wave.  A lane's operands per k-step are two 8-byte LDS reads of the input row (-> 4
// components) and two 16-byte reads of the filter; an accumulator register after the output transform is 32 consecutive
// pairs = up to 256 contiguous bytes of one NCDHW output row.
//
// Workgroups are numbered so that the channel groups of one spatial tile and the tiles next to it run on the same XCD
// (shared L2: the halo re-reads and the 2-3 channel groups' re-reads of the input hit there).
#include "common.hpp"

namespace mgar {

typedef float __attribute__((ext_vector_type(16))) f32x16;
typedef float __attribute__((ext_vector_type(4))) f32x4;
typedef float __attribute__((ext_vector_type(2))) f32x2;

constexpr int CW_CG = 64;                          // output channels per workgroup (2 MFMA row blocks)
constexpr int CW_W_FLOATS = 9 * 2 * CW_CG * 4;     // transformed filter block of one channel pair: [kd*3+kh][half][co][t]
constexpr int CW_W_PER_THREAD = (CW_W_FLOATS / 4 + 255) / 256;   // 5 float4 (1152 over 256 threads)

template <int PW>
struct CwGeom {
    static constexpr int BH = 32 / PW;             // rows of a wave's pair block
    static constexpr int TW = 2 * PW, TH = 4 * BH; // outputs per workgroup tile (w, h)
    static constexpr int IW = TW + 2, IH = TH + 2; // input halo tile
    static constexpr int PLANE = IH * IW, PER_CI = 3 * PLANE, IN_FLOATS = 2 * PER_CI;
    static constexpr int IN_PER_THREAD = (IN_FLOATS + 255) / 256;
};

struct CwArgs {
    int N, Cin, D, H, W, Cout, ncg, cg0, tiles_w, tiles_h, per_xcd;   // this launch: channel groups cg0 .. cg0 + ncg - 1
    long long total;
};

// wt: [cg][c_in pair][kd*3+kh][half][co 64][t 4] (conv3d_wino_filter_kernel)
// NMB: 32-channel blocks per workgroup that exist (2; 1 for a tail group of <= 32 channels, launched separately)
template <int PW, int NMB>
__global__ __launch_bounds__(256, 2) void conv3d_wino_kernel(const float *__restrict__ x, const float *__restrict__ wt, CwArgs a,
                                                             float *__restrict__ y) {
    typedef CwGeom<PW> G;
    // (padded to a whole number of elements per thread: the staging loops carry no guards -- a guarded store lets the compiler
    // sink the global load down to it, behind the MFMA block, and the prefetch is gone)
    __shared__ __attribute__((aligned(16))) float s_in[2][G::IN_PER_THREAD * 256];
    __shared__ __attribute__((aligned(16))) float s_w[2][CW_W_PER_THREAD * 256 * 4];

    // XCD-aware numbering: hardware workgroup b runs on XCD b % 8; logical ids are contiguous per XCD
    const long long logical = (long long)(blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
    if (logical >= a.total) return;
    int rest = (int)logical;
    const int cg = a.cg0 + rest % a.ncg; rest /= a.ncg;
    const int bx = rest % a.tiles_w; rest /= a.tiles_w;
    const int by = rest % a.tiles_h; rest /= a.tiles_h;
    const int d = rest % a.D, n = rest / a.D;
    const int w0 = bx * G::TW, h0 = by * G::TH;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = lane & 31, half = lane >> 5;
    const int pw = l % PW, bh = wave * G::BH + l / PW;

    const int HW = a.H * a.W;
    const long long DHW = (long long)a.D * HW;
    // per-thread gather offsets of the halo tile relative to the first channel of a pair.  The tile is read through a raw buffer
    // resource over the two channels of the pair: an offset outside its range (the padding: -1 = 0xffffffff) returns 0 from the
    // hardware's bounds check, so the staging path has no address arithmetic and no select (vector instructions do not overlap
    // the MFMAs here: every one of them is MFMA time)
    int off[G::IN_PER_THREAD];
#pragma unroll
    for (int u = 0; u < G::IN_PER_THREAD; ++u) {
        const int e = threadIdx.x + u * 256;
        const int c = e / G::PER_CI, r0 = e - c * G::PER_CI;
        const int dz = r0 / G::PLANE, r1 = r0 - dz * G::PLANE;
        const int iy = r1 / G::IW, ix = r1 - iy * G::IW;
        const int dd = d - 1 + dz, hh = h0 - 1 + iy, ww = w0 - 1 + ix;
        const bool ok = e < G::IN_FLOATS && dd >= 0 && dd < a.D && hh >= 0 && hh < a.H && ww >= 0 && ww < a.W;
        off[u] = ok ? 4 * ((int)(c * DHW) + dd * HW + hh * a.W + ww) : -1;   // BYTE offset; -1 = beyond the buffer's range -> reads 0
    }
    const float *xin = x + (long long)n * a.Cin * DHW;
    const int pair_bytes = (int)(8 * DHW);             // two channels
    const f32x4 *wsrc = reinterpret_cast<const f32x4 *>(wt) + (long long)cg * (a.Cin / 2) * (CW_W_FLOATS / 4);

    f32x16 acc[NMB][4];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][t][r] = 0.f;

    float pin[G::IN_PER_THREAD];
    f32x4 pw4[CW_W_PER_THREAD];    // (native vector types: arrays of HIP's float4 structs are not split into registers here)
    auto fetch = [&](int j) {                          // global -> registers: channel pair j
        const __amdgpu_buffer_rsrc_t src = uniform_buffer(xin + (long long)(2 * j) * DHW, pair_bytes);
#pragma unroll
        for (int u = 0; u < G::IN_PER_THREAD; ++u)     // nothing here waits for a load
            pin[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(src, off[u], 0, 0));

        const f32x4 *ws = wsrc + (long long)j * (CW_W_FLOATS / 4);
#pragma unroll
        for (int u = 0; u < CW_W_PER_THREAD; ++u) {
            const int e = threadIdx.x + u * 256;
            pw4[u] = ws[e < CW_W_FLOATS / 4 ? e : threadIdx.x];
        }
    };
    auto stash = [&](int buf) {                        // registers -> LDS
#pragma unroll
        for (int u = 0; u < G::IN_PER_THREAD; ++u) {
            s_in[buf][threadIdx.x + u * 256] = pin[u];
        }
#pragma unroll
        for (int u = 0; u < CW_W_PER_THREAD; ++u) {
            reinterpret_cast<f32x4 *>(s_w[buf])[threadIdx.x + u * 256] = pw4[u];
        }
    };

    const int npair = a.Cin / 2;
    fetch(0);
    stash(0);
    __syncthreads();
    const int in_lane = half * G::PER_CI + bh * G::IW + 2 * pw;     // x[2p-1] of output row bh, tap (kd, kh) = (0, 0)
    for (int j = 0; j < npair; ++j) {
        const int buf = j & 1;
        // the next pair's global loads are in flight under the MFMAs below (the last iteration re-loads its own pair: one
        // unconditional straight-line body, so that nothing merges the load and the LDS store into one early block)
        fetch(min(j + 1, npair - 1));
        __builtin_amdgcn_sched_barrier(0);
        const float *ti = s_in[buf] + in_lane;
        const f32x4 *tw = reinterpret_cast<const f32x4 *>(s_w[buf]) + half * CW_CG + l;
        // operands of k-step kk + 1 are read from LDS before the MFMAs of k-step kk are issued
        f32x2 xa[2], xb[2];
        f32x4 g[2][NMB];
        auto read_operands = [&](int kk, int slot) {
            const int kd = kk / 3, kh = kk - 3 * kd;
            const float *p = ti + kd * G::PLANE + kh * G::IW;
            xa[slot] = *reinterpret_cast<const f32x2 *>(p);
            xb[slot] = *reinterpret_cast<const f32x2 *>(p + 2);
#pragma unroll
            for (int mb = 0; mb < NMB; ++mb) g[slot][mb] = tw[kk * 2 * CW_CG + 32 * mb];
        };
        read_operands(0, 0);
#pragma unroll
        for (int kk = 0; kk < 9; ++kk) {
            const int slot = kk & 1;
            if (kk + 1 < 9) read_operands(kk + 1, slot ^ 1);
            const float u0 = xa[slot].x - xb[slot].x, u1 = xa[slot].y + xb[slot].x, u2 = xb[slot].x - xa[slot].y, u3 = xa[slot].y - xb[slot].y;
#pragma unroll
            for (int mb = 0; mb < NMB; ++mb) {
                acc[mb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[slot][mb].x, u0, acc[mb][0], 0, 0, 0);
                acc[mb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[slot][mb].y, u1, acc[mb][1], 0, 0, 0);
                acc[mb][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[slot][mb].z, u2, acc[mb][2], 0, 0, 0);
                acc[mb][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[slot][mb].w, u3, acc[mb][3], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        stash(buf ^ 1);                                // the other buffer: its last readers passed the barrier of pair j - 1
        __syncthreads();
    }

    // output transform + store: register r of a lane is (co = 32 mb + (r & 3) + 8 (r >> 2) + 4 half, pair l)
    const int ho = h0 + bh, wo = w0 + 2 * pw;
    if (ho < a.H && wo < a.W) {
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cg * CW_CG + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co >= a.Cout) continue;
                const float m0 = acc[mb][0][r], m1 = acc[mb][1][r], m2 = acc[mb][2][r], m3 = acc[mb][3][r];
                f32x2 o;
                o.x = (m0 + m1) + m2;
                o.y = (m1 - m2) - m3;
                *reinterpret_cast<f32x2 *>(y + (((long long)n * a.Cout + co) * a.D + d) * HW + (long long)ho * a.W + wo) = o;
            }
        }
    }
}

// w (Cout, Cin, 3, 3, 3) -> wt [cg][c_in pair][kd*3+kh][half][co 64][t 4]; channels beyond Cout are zero
__global__ void conv3d_wino_filter_kernel(const float *__restrict__ w, int Cin, int Cout, long long total, float *__restrict__ wt) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    long long rest = e;
    const int col = (int)(rest % CW_CG); rest /= CW_CG;
    const int half = (int)(rest % 2); rest /= 2;
    const int kk = (int)(rest % 9); rest /= 9;
    const int j = (int)(rest % (Cin / 2));
    const int cg = (int)(rest / (Cin / 2));
    const int co = cg * CW_CG + col, ci = 2 * j + half;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (co < Cout) {
        const float *src = w + (((long long)co * Cin + ci) * 9 + kk) * 3;
        const float g0 = src[0], g1 = src[1], g2 = src[2];
        g.x = g0;
        g.y = 0.5f * ((g0 + g2) + g1);
        g.z = 0.5f * ((g0 + g2) - g1);
        g.w = g2;
    }
    reinterpret_cast<float4 *>(wt)[e] = g;
}

// Extra dynamic LDS per workgroup (bytes): > 20 KB leaves one workgroup per CU (tools/overlap_probe.py measures how much of
// a concurrent HBM-bound stream then runs beside this MFMA-bound kernel).  0 by default.
static int g_cw_lds_pad = 0;

template <int PW>
static void cw_launch(const float *x, const float *wt, CwArgs a, float *y, hipStream_t st) {
    typedef CwGeom<PW> G;
    a.tiles_w = ceil_div(a.W, G::TW);
    a.tiles_h = ceil_div(a.H, G::TH);
    const int groups = ceil_div(a.Cout, CW_CG);
    const int tail = (a.Cout % CW_CG != 0 && a.Cout % CW_CG <= 32) ? 1 : 0;      // a last group with one 32-channel block
    const long long tiles = (long long)a.tiles_w * a.tiles_h * a.D * a.N;
    if (groups - tail > 0) {
        a.cg0 = 0; a.ncg = groups - tail;
        a.total = tiles * a.ncg;
        a.per_xcd = (int)((a.total + 7) / 8);
        hipLaunchKernelGGL((conv3d_wino_kernel<PW, 2>), dim3(a.per_xcd * 8), dim3(256), g_cw_lds_pad, st, x, wt, a, y);
    }
    if (tail) {
        a.cg0 = groups - 1; a.ncg = 1;
        a.total = tiles;
        a.per_xcd = (int)((a.total + 7) / 8);
        hipLaunchKernelGGL((conv3d_wino_kernel<PW, 1>), dim3(a.per_xcd * 8), dim3(256), g_cw_lds_pad, st, x, wt, a, y);
    }
}

}  // namespace mgar

using namespace mgar;

#define CW_API extern "C" __attribute__((visibility("default")))

// floats of the transformed-filter scratch mgar_conv3d_k3_fwd needs (rewritten on every call)
CW_API long long mgar_conv3d_k3_workspace_floats(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    return (long long)ceil_div(Cout, CW_CG) * CW_CG * Cin * 36;
}

CW_API int mgar_conv3d_k3_set_lds_pad(int bytes) {
    MGAR_REQUIRE(bytes >= 0 && bytes <= 90 * 1024, "conv3d_k3_set_lds_pad: 0 .. 92160 bytes");
    g_cw_lds_pad = bytes;
    return MGAR_OK;
}

// x (N, Cin, D, H, W) fp32 NCDHW, w (Cout, Cin, 3, 3, 3) -> y (N, Cout, D, H, W): stride 1, zero padding 1 on every side.
// Cin and W must be even (every I3D instance is); anything else is MGAR_EINVAL and the caller keeps the library convolution.
CW_API int mgar_conv3d_k3_fwd(const float *x, int N, int Cin, int D, int H, int W, const float *w, int Cout, float *w_packed, float *y,
                              void *stream) {
    MGAR_REQUIRE(N >= 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0, "conv3d_k3_fwd: bad sizes");
    MGAR_REQUIRE(Cin % 2 == 0 && W % 2 == 0, "conv3d_k3_fwd: Cin and W must be even");
    if (N == 0) return MGAR_OK;
    MGAR_REQUIRE(x && w && w_packed && y, "conv3d_k3_fwd: null pointer");
    MGAR_REQUIRE((long long)8 * D * H * W < (1ll << 31), "conv3d_k3_fwd: volume too large for 32-bit tile offsets");
    hipStream_t st = (hipStream_t)stream;
    CwArgs a{N, Cin, D, H, W, Cout, ceil_div(Cout, CW_CG), 0, 0, 0, 0, 0};
    const long long wtotal = (long long)a.ncg * (Cin / 2) * 9 * 2 * CW_CG;      // float4 elements
    MGAR_REQUIRE((long long)a.ncg * ceil_div(W, 16) * ceil_div(H, 4) * D * N < (1ll << 30), "conv3d_k3_fwd: too many tiles");
    hipLaunchKernelGGL(conv3d_wino_filter_kernel, dim3(ceil_div(wtotal, 256)), dim3(256), 0, st, w, Cin, Cout, wtotal, w_packed);
    // the tile shape that wastes the fewest outputs (ties: the widest rows)
    auto padded = [&](int pw) { return (long long)ceil_div(W, 2 * pw) * 2 * pw * ceil_div(H, 128 / pw) * (128 / pw); };
    int best = 32;
    if (padded(16) < padded(best)) best = 16;
    if (padded(8) < padded(best)) best = 8;
    const double outs = (double)N * D * H * W;
    {
        // bytes: input + output once; flops: the MFMA work actually issued = 2/3 of the direct convolution's 2 * 27 * Cin * Cout per output
        KtScope kt(KT_CONV3D_WINO, st, 4.0 * outs * (Cin + Cout), 2.0 * outs * Cout * Cin * 18.0);
        if (best == 32) cw_launch<32>(x, w_packed, a, y, st);
        else if (best == 16) cw_launch<16>(x, w_packed, a, y, st);
        else cw_launch<8>(x, w_packed, a, y, st);
    }
    return check_launch("conv3d_k3_fwd: launch failed");
}
