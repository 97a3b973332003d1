// pointwise_dw.hip -- weight gradient of a point-wise (kernel-size-1) convolution on MFMA, gfx950.
//
//   dW[o][i] = sum_b sum_p dY[b, o, p] * X[b, i, p]        X (B, Cin, P), dY (B, Cout, P)
//
// This is the backward-weights of the Conv2d 1x1 layers of the reference's shared MLPs
// (pointnet2_batch/pointnet2_modules.py:86-92 and the stack / voxel-pool equivalents).  As a GEMM
// it is degenerate: M x N = Cout x Cin is tiny (16..128) and K = B*P is up to 15.7 M columns.  The
// library picks a 16x32x512 macro-tile for it and reaches 0.9 TB/s on the two streamed operands
// (6.5 ms per call at config c3, 39 ms per step).  Here the two operands are streamed exactly once
// through LDS and multiplied on the matrix cores with the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32:
// bit-for-bit an fp32 fma chain, so no precision is given up):
//   * a workgroup walks 128-column tiles (grid-stride), stages dY and X rows in LDS with coalesced
//     16-byte loads (row stride 129 floats => the transposed operand reads are bank-conflict free);
//   * each of its 4 waves takes 32 columns of the tile = 16 MFMA k-steps (K = 2 columns each) for
//     every 32x32 output block; the Cout x Cin accumulator lives in registers for the whole kernel;
//   * one flush at the end: every workgroup stores its partial Cout x Cin block to a workspace and a
//     second small kernel sums the partials in workgroup order -- deterministic, and without the
//     ~0.5 ms that 2048 workgroups x (Cout*Cin) float atomics on the SAME 1-4 K addresses cost
//     (same-address atomics serialise in L2; measured as the size-independent floor of this kernel).
// HBM-bound: 4*(Cin + Cout) bytes per column; MFMA time is ~4x below the streaming time.
#include "common.hpp"

namespace mgar {

typedef float __attribute__((ext_vector_type(16))) f32x16;

constexpr int DW_TP = 128;            // columns per tile
constexpr int DW_LD = DW_TP + 1;      // LDS row stride (floats)

// optional activation of the X operand on the fly: x := relu?(x * sc_i + sh_i) with the BatchNorm
// arithmetic of bn_apply_kernel (csrc/bn_act.hip) -- X is then the PRE-BN output of the previous
// layer and the activated tensor is never materialised (see csrc/pointwise_fwd.hip)
struct DwAct {
    const float *mean, *invstd, *gamma, *beta;
    int relu;
    // pair != 0 ("dW and the BatchNorm backward reduction in one pass"): the kernel runs with 2 * Cin VIRTUAL input channels,
    //   virtual i <  Cin : m_i       = [relu'd pre-activation of channel i is positive] (1 where there is no ReLU)
    //   virtual i >= Cin : m_i * xhat_i,  xhat = (x - mean) * invstd
    // so its result is [A | B] with A[o][i] = sum_p dy[o,p] m_i[p], B[o][i] = sum_p dy[o,p] m_i[p] xhat_i[p]; from these
    //   dW = gamma_i B + beta_i A,   sum_p dz_i = sum_o W[o][i] A[o][i],   sum_p dz_i xhat_i = sum_o W[o][i] B[o][i]
    // (dz = (W^T dy) m: what bn_bwd_partial_kernel would need a pass over W^T dy and x for).  Cin here = REAL channels.
    int pair;
};

// OB x IB output blocks of 32 x 32 per workgroup (OB * IB <= 8)
template <int OB, int IB>
__global__ __launch_bounds__(256) void pointwise_dw_kernel(const float *__restrict__ x, const float *__restrict__ dy, int B, int Cin,
                                                           int Cout, int P, DwAct act, float *__restrict__ partial) {
    extern __shared__ float lds[];                 // [(OB + IB) * 32][DW_LD]
    __shared__ float act_sc[IB * 32], act_sh[IB * 32], act_mu[IB * 32];   // (x - mu) * sc + sh, as bn_apply_kernel
    __shared__ float act_is[IB * 32];                                     // pair mode: invstd of the row's real channel
    float *sy = lds;                               // dY rows of this workgroup's output blocks
    float *sx = lds + OB * 32 * DW_LD;             // X rows of this workgroup's input blocks
    const int o_base = blockIdx.y * OB * 32, i_base = blockIdx.z * IB * 32;
    const bool has_act = act.mean != nullptr;
    const bool pair = act.pair != 0;
    const int CinV = pair ? 2 * Cin : Cin;         // (virtual) input channels = columns of the result
    if (has_act) {
        for (int r = threadIdx.x; r < IB * 32; r += 256) {
            const int chv = i_base + r, ch = (pair && chv >= Cin) ? chv - Cin : chv;
            float sc = 1.f, sh = 0.f, mu = 0.f, is = 0.f;
            if (chv < CinV) {
                sc = act.invstd[ch] * (act.gamma ? act.gamma[ch] : 1.f);
                sh = act.beta ? act.beta[ch] : 0.f;
                mu = act.mean[ch];
                is = act.invstd[ch];
            }
            act_sc[r] = sc;
            act_sh[r] = sh;
            act_mu[r] = mu;
            act_is[r] = is;
        }
        __syncthreads();
    }
    // pair mode: what a staged X element becomes (kind = second half of the virtual channels)
    auto pair_of = [&](float v, int r, bool kind) {
        const float mu = act_mu[r];
        const bool on = !act.relu || (v - mu) * act_sc[r] + act_sh[r] > 0.f;      // the forward's expression, bit for bit
        return on ? (kind ? (v - mu) * act_is[r] : 1.f) : 0.f;
    };
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles_per_b = (P + DW_TP - 1) / DW_TP;
    const long long total_tiles = (long long)B * tiles_per_b;
    const bool vec = (P & 3) == 0;

    f32x16 acc[OB][IB];
#pragma unroll
    for (int a = 0; a < OB; ++a)
#pragma unroll
        for (int c = 0; c < IB; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    // register prefetch (vector path): the global loads of tile t+1 are in flight while the MFMAs of
    // tile t run; the LDS tile is single-buffered
    constexpr int NV = (OB + IB) * 32 * (DW_TP / 4) / 256;   // float4 per thread and tile
    float4 pre[NV];
    auto fetch = [&](long long tile) {
        const int b = (int)(tile / tiles_per_b);
        const int p0 = (int)(tile - (long long)b * tiles_per_b) * DW_TP;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int e = threadIdx.x + u * 256;
            const int row = e / (DW_TP / 4), c4 = (e - row * (DW_TP / 4)) * 4;
            const bool is_y = row < OB * 32;
            const int chv = is_y ? o_base + row : i_base + row - OB * 32;
            const int ch = (!is_y && pair && chv >= Cin) ? chv - Cin : chv;       // the real row to read
            const int cmax = is_y ? Cout : Cin;
            pre[u] = (chv < (is_y ? Cout : CinV) && p0 + c4 < P)
                         ? *reinterpret_cast<const float4 *>((is_y ? dy : x) + ((size_t)b * cmax + ch) * P + p0 + c4)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (vec && (long long)blockIdx.x < total_tiles) fetch(blockIdx.x);

    for (long long tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const int b = (int)(tile / tiles_per_b);
        const int p0 = (int)(tile - (long long)b * tiles_per_b) * DW_TP;
        // ---- stage (OB + IB) * 32 rows x 128 columns, zero-padded ----
        if (vec) {
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int e = threadIdx.x + u * 256;
                const int row = e / (DW_TP / 4), c4 = (e - row * (DW_TP / 4)) * 4;
                const bool is_y = row < OB * 32;
                const int ch = is_y ? o_base + row : i_base + row - OB * 32;
                float4 v = pre[u];
                if (has_act && !is_y && ch < CinV && p0 + c4 < P) {   // padding stays zero
                    const int r = row - OB * 32;
                    if (pair) {
                        const bool kind = ch >= Cin;
                        v.x = pair_of(v.x, r, kind); v.y = pair_of(v.y, r, kind); v.z = pair_of(v.z, r, kind); v.w = pair_of(v.w, r, kind);
                    } else {
                        const float sc = act_sc[r], sh = act_sh[r], mu = act_mu[r];
                        v.x = (v.x - mu) * sc + sh; v.y = (v.y - mu) * sc + sh; v.z = (v.z - mu) * sc + sh; v.w = (v.w - mu) * sc + sh;
                        if (act.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    }
                }
                float *d = lds + row * DW_LD + c4;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        } else {
            for (int e = threadIdx.x; e < (OB + IB) * 32 * DW_TP; e += 256) {
                const int row = e / DW_TP, cc = e - row * DW_TP;
                const bool is_y = row < OB * 32;
                const int chv = is_y ? o_base + row : i_base + row - OB * 32;
                const int ch = (!is_y && pair && chv >= Cin) ? chv - Cin : chv;
                const int cmax = is_y ? Cout : Cin;
                float v = 0.f;
                if (chv < (is_y ? Cout : CinV) && p0 + cc < P) {
                    v = (is_y ? dy : x)[((size_t)b * cmax + ch) * P + p0 + cc];
                    if (has_act && !is_y) {
                        if (pair) {
                            v = pair_of(v, row - OB * 32, chv >= Cin);
                        } else {
                            v = (v - act_mu[row - OB * 32]) * act_sc[row - OB * 32] + act_sh[row - OB * 32];
                            if (act.relu) v = fmaxf(v, 0.f);
                        }
                    }
                }
                lds[row * DW_LD + cc] = v;
            }
        }
        __syncthreads();
        if (vec && tile + gridDim.x < total_tiles) fetch(tile + gridDim.x);
        // ---- this wave's 32 columns: 16 k-steps of 2 columns ----
        const int colw = wave * 32 + (lane >> 5);
#pragma unroll 4
        for (int ks = 0; ks < 16; ++ks) {
            const int cc = colw + 2 * ks;
            float af[OB], bf[IB];
#pragma unroll
            for (int a = 0; a < OB; ++a) af[a] = sy[(a * 32 + (lane & 31)) * DW_LD + cc];   // A[i = o][k = column]
#pragma unroll
            for (int c = 0; c < IB; ++c) bf[c] = sx[(c * 32 + (lane & 31)) * DW_LD + cc];   // B[k = column][j = i]
#pragma unroll
            for (int a = 0; a < OB; ++a)
#pragma unroll
                for (int c = 0; c < IB; ++c)
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[c], acc[a][c], 0, 0, 0);
        }
        __syncthreads();
    }
    // ---- combine the 4 waves in LDS (wave order), then wave 0 stores the workgroup's partial block ----
    // (the staging area is free now: the loop ends with a barrier; 3 * OB*IB * 1024 floats fit in it)
    if (wave > 0) {
#pragma unroll
        for (int a = 0; a < OB; ++a)
#pragma unroll
            for (int c = 0; c < IB; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) lds[(((wave - 1) * OB * IB + a * IB + c) * 16 + r) * 64 + lane] = acc[a][c][r];
    }
    __syncthreads();
    if (wave > 0) return;
    // D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31]
#pragma unroll
    for (int a = 0; a < OB; ++a)
#pragma unroll
        for (int c = 0; c < IB; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[a][c][r];
#pragma unroll
                for (int w = 0; w < 3; ++w) v += lds[((w * OB * IB + a * IB + c) * 16 + r) * 64 + lane];
                const int o = o_base + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int i = i_base + c * 32 + (lane & 31);
                if (o < Cout && i < CinV) partial[((size_t)blockIdx.x * Cout + o) * CinV + i] = v;
            }
}

// dw[e] = sum over the gx partials in a fixed order: 1024 threads = 64 outputs x 16 slices of gx, every slice summed in
// workgroup order with 8 loads in flight, then the slices in order.  (Round 1 ran 4 slices of up to 512 dependent loads:
// 52 us per launch, 19 launches per step, for a few KB of output.)
constexpr int DWR_SLICES = 16;
__global__ __launch_bounds__(64 * DWR_SLICES) void pointwise_dw_reduce_kernel(const float *__restrict__ partial, int gx, int n_out,
                                                                              float *__restrict__ dw) {
    __shared__ float part[DWR_SLICES][64];
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
    const int per = (gx + DWR_SLICES - 1) / DWR_SLICES, g0 = min(slice * per, gx), g1 = min(g0 + per, gx);
    float s = 0.f;
    if (e < n_out) {
        int g = g0;
        for (; g + 8 <= g1; g += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(g + u) * n_out + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; g < g1; ++g) s += partial[(size_t)g * n_out + e];
    }
    part[slice][threadIdx.x & 63] = s;
    __syncthreads();
    if (slice == 0 && e < n_out) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < DWR_SLICES; ++i) t += part[i][threadIdx.x];
        dw[e] = t;
    }
}

// pair mode epilogue: ab (Cout, 2 Cin) = [A | B] -> dW (Cout, Cin) = gamma_i B + beta_i A, and per input channel
// dbeta_i = sum_o W[o][i] A[o][i] (= sum dz), dgamma_i = sum_o W[o][i] B[o][i] (= sum dz xhat), coef = their means over n.
// One thread per input channel; the sums over o (<= 64 terms) in double.
__global__ void pointwise_dw_pair_finalize_kernel(const float *__restrict__ ab, const float *__restrict__ w, int Cin, int Cout, double n,
                                                  const float *__restrict__ gamma, const float *__restrict__ beta,
                                                  float *__restrict__ dw, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                  float *__restrict__ coef) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cin) return;
    const float g = gamma ? gamma[i] : 1.f, b = beta ? beta[i] : 0.f;
    double s = 0.0, q = 0.0;
    for (int o = 0; o < Cout; ++o) {
        const float a = ab[(size_t)o * 2 * Cin + i], bb = ab[(size_t)o * 2 * Cin + Cin + i], wv = w[(size_t)o * Cin + i];
        if (dw) dw[(size_t)o * Cin + i] = g * bb + b * a;
        s += (double)wv * (double)a;
        q += (double)wv * (double)bb;
    }
    if (dbeta) dbeta[i] = (float)s;
    if (dgamma) dgamma[i] = (float)q;
    coef[2 * i + 0] = (float)(s / n);
    coef[2 * i + 1] = (float)(q / n);
}

static int dw_grid_x(int B, int Cin, int Cout, int P, int ob, int ib) {
    const long long tiles = (long long)B * ((P + DW_TP - 1) / DW_TP);
    const int gy = ceil_div(Cout, ob * 32), gz = ceil_div(Cin, ib * 32);
    long long gx = 2048 / (gy * gz);                 // ~8 workgroups per CU in total ...
    if (gx > (tiles + 3) / 4) gx = (tiles + 3) / 4;  // ... each with at least 4 tiles to stream
    if (gx < 1) gx = 1;
    return (int)gx;
}
static void dw_blocks(int Cin, int Cout, int &ob, int &ib) {
    ob = Cout <= 32 ? 1 : 2;
    ib = Cin <= 32 ? 1 : (Cin <= 64 ? 2 : 4);        // OB * IB <= 8
}

template <int OB, int IB>
static void launch_dw(const float *x, const float *dy, int B, int Cin, int Cout, int P, const DwAct &act, float *partial,
                      hipStream_t st) {
    static bool attr_set = false;
    const int lds = (OB + IB) * 32 * DW_LD * (int)sizeof(float);
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)pointwise_dw_kernel<OB, IB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int CinV = act.pair ? 2 * Cin : Cin;
    const int gy = ceil_div(Cout, OB * 32), gz = ceil_div(CinV, IB * 32);
    const int gx = dw_grid_x(B, CinV, Cout, P, OB, IB);
    hipLaunchKernelGGL((pointwise_dw_kernel<OB, IB>), dim3((unsigned)gx, gy, gz), dim3(256), lds, st, x, dy, B, Cin, Cout, P, act, partial);
}

}  // namespace mgar

using namespace mgar;

extern "C" __attribute__((visibility("default"))) int mgar_pointwise_dw_workspace_floats(int B, int Cin, int Cout, int P) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || P <= 0) return 0;
    int ob, ib;
    dw_blocks(Cin, Cout, ob, ib);
    return dw_grid_x(B, Cin, Cout, P, ob, ib) * Cout * Cin;   // <= 2048 * 64 * 256
}

extern "C" __attribute__((visibility("default"))) int mgar_pointwise_conv_dw_act(const float *x, const float *dy, int B, int Cin,
                                                                                int Cout, int P, const float *in_mean,
                                                                                const float *in_invstd, const float *in_gamma,
                                                                                const float *in_beta, int in_relu,
                                                                                float *workspace, float *dw, void *stream) {
    MGAR_REQUIRE(B >= 0 && Cin >= 0 && Cout >= 0 && P >= 0, "pointwise_conv_dw: negative size");
    if (Cin == 0 || Cout == 0) return MGAR_OK;
    MGAR_REQUIRE(dw, "pointwise_conv_dw: null pointer");
    MGAR_REQUIRE(in_mean == nullptr || in_invstd != nullptr, "pointwise_conv_dw: in_mean without in_invstd");
    hipStream_t st = (hipStream_t)stream;
    if ((long long)B * P == 0) {
        (void)hipMemsetAsync(dw, 0, sizeof(float) * Cout * Cin, st);
        return check_launch("pointwise_conv_dw: memset failed");
    }
    MGAR_REQUIRE(x && dy && workspace, "pointwise_conv_dw: null pointer");
    const DwAct act{in_mean, in_invstd, in_gamma, in_beta, in_relu, 0};
    int ob, ib;
    dw_blocks(Cin, Cout, ob, ib);
    {
    KtScope kt(KT_POINTWISE_DW, st, 4.0 * (double)B * P * (Cin + Cout), 2.0 * (double)B * P * Cin * Cout);
    if (ob == 1 && ib == 1) launch_dw<1, 1>(x, dy, B, Cin, Cout, P, act, workspace, st);
    else if (ob == 1 && ib == 2) launch_dw<1, 2>(x, dy, B, Cin, Cout, P, act, workspace, st);
    else if (ob == 1 && ib == 4) launch_dw<1, 4>(x, dy, B, Cin, Cout, P, act, workspace, st);
    else if (ob == 2 && ib == 1) launch_dw<2, 1>(x, dy, B, Cin, Cout, P, act, workspace, st);
    else if (ob == 2 && ib == 2) launch_dw<2, 2>(x, dy, B, Cin, Cout, P, act, workspace, st);
    else launch_dw<2, 4>(x, dy, B, Cin, Cout, P, act, workspace, st);
    }
    const int gx = dw_grid_x(B, Cin, Cout, P, ob, ib), n_out = Cout * Cin;
    hipLaunchKernelGGL(pointwise_dw_reduce_kernel, dim3(ceil_div(n_out, 64)), dim3(64 * DWR_SLICES), 0, st, workspace, gx, n_out, dw);
    return check_launch("pointwise_conv_dw: launch failed");
}

extern "C" __attribute__((visibility("default"))) int mgar_pointwise_conv_dw(const float *x, const float *dy, int B, int Cin,
                                                                            int Cout, int P, float *workspace, float *dw,
                                                                            void *stream) {
    return mgar_pointwise_conv_dw_act(x, dy, B, Cin, Cout, P, nullptr, nullptr, nullptr, nullptr, 0, workspace, dw, stream);
}

// ---- weight gradient of [BatchNorm -> ReLU -> conv 1x1] AND the reduction of that BatchNorm's backward, in one pass ----------
// x (B, Cin, P): the layer's PRE-BatchNorm input; dy (B, Cout, P): gradient of the conv output; w (Cout, Cin) row-major.
// -> dw (Cout, Cin); dgamma, dbeta (Cin); coef (2 Cin) = {mean dz, mean dz xhat} for mgar_bn_act_bwd_apply[_rowmajor],
// where dz = (W^T dy) [relu active].  Replaces mgar_pointwise_conv_dw_act + the reduction pass of mgar_bn_act_bwd over W^T dy
// and x (8 * B * Cin * P bytes).  Cin <= 64, Cout <= 64.  workspace: mgar_pointwise_dw_bnbwd_workspace_floats(...) floats.
extern "C" __attribute__((visibility("default"))) int mgar_pointwise_dw_bnbwd_workspace_floats(int B, int Cin, int Cout, int P) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || P <= 0) return 0;
    int ob, ib;
    dw_blocks(2 * Cin, Cout, ob, ib);
    return dw_grid_x(B, 2 * Cin, Cout, P, ob, ib) * Cout * 2 * Cin + Cout * 2 * Cin;
}
extern "C" __attribute__((visibility("default"))) int mgar_pointwise_conv_dw_bnbwd(const float *x, const float *dy, const float *w, int B,
                                                                                  int Cin, int Cout, int P, const float *in_mean,
                                                                                  const float *in_invstd, const float *in_gamma,
                                                                                  const float *in_beta, int in_relu, float *workspace,
                                                                                  float *dw, float *dgamma, float *dbeta, float *coef,
                                                                                  void *stream) {
    MGAR_REQUIRE(B >= 0 && Cin >= 1 && Cout >= 1 && P >= 0, "pointwise_conv_dw_bnbwd: bad sizes");
    if (Cin > 64 || Cout > 64) {
        set_error("pointwise_conv_dw_bnbwd: Cin <= 64 and Cout <= 64");
        return MGAR_EUNSUPPORTED;
    }
    MGAR_REQUIRE((long long)B * P > 0, "pointwise_conv_dw_bnbwd: empty input");
    MGAR_REQUIRE(x && dy && w && in_mean && in_invstd && workspace && coef, "pointwise_conv_dw_bnbwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const DwAct act{in_mean, in_invstd, in_gamma, in_beta, in_relu, 1};
    int ob, ib;
    dw_blocks(2 * Cin, Cout, ob, ib);
    const int gx = dw_grid_x(B, 2 * Cin, Cout, P, ob, ib), n_out = Cout * 2 * Cin;
    float *ab = workspace + (size_t)gx * n_out;
    {
        KtScope kt(KT_POINTWISE_DW, st, 4.0 * (double)B * P * (Cin + Cout), 4.0 * (double)B * P * Cin * Cout);
        if (ob == 1 && ib == 1) launch_dw<1, 1>(x, dy, B, Cin, Cout, P, act, workspace, st);
        else if (ob == 1 && ib == 2) launch_dw<1, 2>(x, dy, B, Cin, Cout, P, act, workspace, st);
        else if (ob == 1 && ib == 4) launch_dw<1, 4>(x, dy, B, Cin, Cout, P, act, workspace, st);
        else if (ob == 2 && ib == 1) launch_dw<2, 1>(x, dy, B, Cin, Cout, P, act, workspace, st);
        else if (ob == 2 && ib == 2) launch_dw<2, 2>(x, dy, B, Cin, Cout, P, act, workspace, st);
        else launch_dw<2, 4>(x, dy, B, Cin, Cout, P, act, workspace, st);
    }
    hipLaunchKernelGGL(pointwise_dw_reduce_kernel, dim3(ceil_div(n_out, 64)), dim3(64 * DWR_SLICES), 0, st, workspace, gx, n_out, ab);
    hipLaunchKernelGGL(pointwise_dw_pair_finalize_kernel, dim3(ceil_div(Cin, 64)), dim3(64), 0, st, ab, w, Cin, Cout, (double)B * P, in_gamma,
                       in_beta, dw, dgamma, dbeta, coef);
    return check_launch("pointwise_conv_dw_bnbwd: launch failed");
}
