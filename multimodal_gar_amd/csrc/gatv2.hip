// gatv2.hip -- GATv2 edge-softmax + neighbourhood aggregation (fwd + bwd) for gfx950.
//
// Replaces the message-passing core of torch_geometric.nn.GATv2Conv, which the reference
// calls over the fully connected actor graph (model/gat_model.py:1019, :1082-1094;
// sg_model.py:59, :122-134).  torch_geometric is an unpinned third-party dependency that is
// not in the reference tree; the arithmetic is restated from the published GATv2 layer:
//     e_ij   = a_h . LeakyReLU_slope(x_l[j] + x_r[i])         (j -> i, per head h)
//     alpha  = softmax_j(e_ij)  over the incoming edges of i (self loop included)
//     out_i  = sum_j alpha_ij x_l[j]                           (per head; heads are averaged
//                                                              and the bias added by the caller)
// PyG materialises x_l[j] + x_r[i] for every edge ((N^2) x H x C floats: 268 MB at N = 128).
// Here nothing per-edge is materialised except alpha (E x H floats):
//   one WAVE per (target node i, head h), lanes along the channel axis (C/64 per lane);
//   per incoming edge: fused add + LeakyReLU + dot with a_h, a DPP wave reduction, then an
//   online pass for the softmax statistics; a second sweep normalises and accumulates out_i.
// Edges arrive in CSR form grouped by target node (rowptr, col = source node).
#include "common.hpp"

namespace mgar {

// The raw logits make a round trip through the alpha buffer inside one wave (lane 0 stores,
// all lanes reload).  Both sides use agent-scope relaxed atomics (sc1: served by L2, never
// by this CU's vector L1) and the wave drains its stores in between, so the reload cannot
// see a stale line.
__device__ __forceinline__ void st_l2(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_l2(const float *p) {
    return __hip_atomic_load(const_cast<float *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int CPL>
__global__ __launch_bounds__(256) void gatv2_fwd_kernel(int n_nodes, int H, const int *__restrict__ rowptr,
                                                        const int *__restrict__ col, const float *__restrict__ xl,
                                                        const float *__restrict__ xr, const float *__restrict__ att,
                                                        float slope, const float *__restrict__ edge_scale,
                                                        float *__restrict__ alpha, float *__restrict__ out) {
    constexpr int C = CPL * kWave;
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_nodes * H) return;
    const int i = w / H, h = w - i * H;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    float a[CPL], r[CPL], acc[CPL];
#pragma unroll
    for (int u = 0; u < CPL; ++u) {
        a[u] = att[(size_t)h * C + u * kWave + lane];
        r[u] = xr[((size_t)i * H + h) * C + u * kWave + lane];
        acc[u] = 0.f;
    }
    // sweep 1: raw logits -> alpha buffer, running max
    float emax = -__builtin_inff();
    for (int e = e0; e < e1; ++e) {
        const int j = col[e];
        const float *lj = xl + ((size_t)j * H + h) * C;
        float part = 0.f;
#pragma unroll
        for (int u = 0; u < CPL; ++u) {
            const float z = lj[u * kWave + lane] + r[u];
            part += a[u] * (z > 0.f ? z : slope * z);
        }
        const float lg = wave_sum(part);
        emax = fmaxf(emax, lg);
        if (lane == 0) st_l2(alpha + (size_t)e * H + h, lg);
    }
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): the logit stores have reached L2
    float denom = 0.f;
    for (int e = e0 + lane; e < e1; e += kWave) denom += __expf(ld_l2(alpha + (size_t)e * H + h) - emax);
    denom = wave_sum(denom);
    const float inv = denom > 0.f ? 1.f / denom : 0.f;
    // sweep 2: normalise and aggregate
    for (int e = e0; e < e1; ++e) {
        const int j = col[e];
        const float al = __expf(ld_l2(alpha + (size_t)e * H + h) - emax) * inv;
        const float *lj = xl + ((size_t)j * H + h) * C;
        const float am = edge_scale ? al * edge_scale[(size_t)e * H + h] : al;
#pragma unroll
        for (int u = 0; u < CPL; ++u) acc[u] += am * lj[u * kWave + lane];
        if (lane == 0) st_l2(alpha + (size_t)e * H + h, al);
    }
#pragma unroll
    for (int u = 0; u < CPL; ++u) out[((size_t)i * H + h) * C + u * kWave + lane] = acc[u];
}

// Backward, without float atomics (round 3: every sum has a fixed order, gradients are bit-reproducible).
//   dalpha_ij = gO_i . x_l[j] ;  de_ij = alpha_ij (m_ij dalpha_ij - sum_j' alpha_ij' m_ij' dalpha_ij')
//   dz_ijc = de_ij a_c LeakyReLU'(z_ijc) ;  grad_xr[i] = sum_j dz_ij ;
//   grad_xl[j] = sum_i alpha_ij m_ij gO_i + dz_ij ;  grad_att[h] = sum_ij de_ij LeakyReLU(z_ij)
// Pass 1 (gatv2_bwd_target_kernel): one wave per (target i, head h) walks i's incoming edges in CSR order: de_ij goes to
//   a scratch (E, H), grad_xr[i] is complete, the (i, h) share of grad_att goes to att_part (n, H, C).
// Pass 2 (gatv2_bwd_source_kernel): one wave per (source j, head h) walks j's OUTGOING edges in the order of the
//   by-source index (src_rowptr / src_edge / src_dst, ascending edge id) and owns grad_xl[j, h, :]; the waves of the first
//   n_att blocks' worth then add up att_part over i in ascending order.
template <int CPL>
__global__ __launch_bounds__(256) void gatv2_bwd_target_kernel(int n_nodes, int H, const int *__restrict__ rowptr,
                                                               const int *__restrict__ col, const float *__restrict__ xl,
                                                               const float *__restrict__ xr, const float *__restrict__ att,
                                                               float slope, const float *__restrict__ edge_scale,
                                                               const float *__restrict__ alpha,
                                                               const float *__restrict__ grad_out, float *__restrict__ de_out,
                                                               float *__restrict__ grad_xr, float *__restrict__ att_part) {
    constexpr int C = CPL * kWave;
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_nodes * H) return;
    const int i = w / H, h = w - i * H;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    float a[CPL], r[CPL], go[CPL], gxr[CPL], gatt[CPL];
#pragma unroll
    for (int u = 0; u < CPL; ++u) {
        const size_t o = ((size_t)i * H + h) * C + u * kWave + lane;
        a[u] = att[(size_t)h * C + u * kWave + lane];
        r[u] = xr[o];
        go[u] = grad_out[o];
        gxr[u] = 0.f;
        gatt[u] = 0.f;
    }
    float s = 0.f;  // sum_j alpha_ij m_ij dalpha_ij
    for (int e = e0; e < e1; ++e) {
        const float *lj = xl + ((size_t)col[e] * H + h) * C;
        float part = 0.f;
#pragma unroll
        for (int u = 0; u < CPL; ++u) part += go[u] * lj[u * kWave + lane];
        const float m = edge_scale ? edge_scale[(size_t)e * H + h] : 1.f;
        s += alpha[(size_t)e * H + h] * m * wave_sum(part);
    }
    for (int e = e0; e < e1; ++e) {
        const float *lj = xl + ((size_t)col[e] * H + h) * C;
        float lv[CPL], part = 0.f;
#pragma unroll
        for (int u = 0; u < CPL; ++u) { lv[u] = lj[u * kWave + lane]; part += go[u] * lv[u]; }
        const float al = alpha[(size_t)e * H + h];
        const float m = edge_scale ? edge_scale[(size_t)e * H + h] : 1.f;
        const float de = al * (m * wave_sum(part) - s);
        if (lane == 0) de_out[(size_t)e * H + h] = de;
#pragma unroll
        for (int u = 0; u < CPL; ++u) {
            const float z = lv[u] + r[u];
            gxr[u] += de * a[u] * (z > 0.f ? 1.f : slope);
            gatt[u] += de * (z > 0.f ? z : slope * z);
        }
    }
#pragma unroll
    for (int u = 0; u < CPL; ++u) {
        const size_t o = ((size_t)i * H + h) * C + u * kWave + lane;
        grad_xr[o] = gxr[u];
        att_part[o] = gatt[u];
    }
}

template <int CPL>
__global__ __launch_bounds__(256) void gatv2_bwd_source_kernel(int n_nodes, int H, const int *__restrict__ src_rowptr,
                                                               const int *__restrict__ src_edge, const int *__restrict__ src_dst,
                                                               const float *__restrict__ xl, const float *__restrict__ xr,
                                                               const float *__restrict__ att, float slope,
                                                               const float *__restrict__ edge_scale,
                                                               const float *__restrict__ alpha, const float *__restrict__ de_in,
                                                               const float *__restrict__ grad_out,
                                                               const float *__restrict__ att_part, float *__restrict__ grad_xl,
                                                               float *__restrict__ grad_att) {
    constexpr int C = CPL * kWave;
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_nodes * H) return;
    const int j = w / H, h = w - j * H;
    float a[CPL], l[CPL], acc[CPL];
#pragma unroll
    for (int u = 0; u < CPL; ++u) {
        a[u] = att[(size_t)h * C + u * kWave + lane];
        l[u] = xl[((size_t)j * H + h) * C + u * kWave + lane];
        acc[u] = 0.f;
    }
    for (int q = src_rowptr[j]; q < src_rowptr[j + 1]; ++q) {
        const int e = src_edge[q], i = src_dst[q];
        const float al = alpha[(size_t)e * H + h];
        const float m = edge_scale ? edge_scale[(size_t)e * H + h] : 1.f;
        const float de = de_in[(size_t)e * H + h];
        const size_t o = ((size_t)i * H + h) * C;
#pragma unroll
        for (int u = 0; u < CPL; ++u) {
            const float z = l[u] + xr[o + u * kWave + lane];
            acc[u] += al * m * grad_out[o + u * kWave + lane] + de * a[u] * (z > 0.f ? 1.f : slope);
        }
    }
#pragma unroll
    for (int u = 0; u < CPL; ++u) grad_xl[((size_t)j * H + h) * C + u * kWave + lane] = acc[u];
    if (j == 0) {   // the H waves of node 0 also fold att_part over the nodes, ascending: grad_att[h, :]
#pragma unroll
        for (int u = 0; u < CPL; ++u) {
            float t = 0.f;
            for (int i = 0; i < n_nodes; ++i) t += att_part[((size_t)i * H + h) * C + u * kWave + lane];
            grad_att[(size_t)h * C + u * kWave + lane] = t;
        }
    }
}

}  // namespace mgar

using namespace mgar;

#define GAT_DISPATCH(KERNEL, ...)                                                                         \
    switch (C / 64) {                                                                                     \
        case 1: hipLaunchKernelGGL(KERNEL<1>, grid, dim3(256), 0, st, __VA_ARGS__); break;                \
        case 2: hipLaunchKernelGGL(KERNEL<2>, grid, dim3(256), 0, st, __VA_ARGS__); break;                \
        case 4: hipLaunchKernelGGL(KERNEL<4>, grid, dim3(256), 0, st, __VA_ARGS__); break;                \
        case 8: hipLaunchKernelGGL(KERNEL<8>, grid, dim3(256), 0, st, __VA_ARGS__); break;                \
        case 16: hipLaunchKernelGGL(KERNEL<16>, grid, dim3(256), 0, st, __VA_ARGS__); break;              \
        default: set_error("gatv2: C must be 64, 128, 256, 512 or 1024"); return MGAR_EUNSUPPORTED;       \
    }

extern "C" __attribute__((visibility("default"))) int mgar_gatv2_fwd(int n_nodes, int H, int C, const int *rowptr,
                                                                    const int *col, const float *xl, const float *xr,
                                                                    const float *att, float slope,
                                                                    const float *edge_scale, float *alpha, float *out,
                                                                    void *stream) {
    MGAR_REQUIRE(n_nodes >= 0 && H > 0 && C > 0, "gatv2_fwd: bad sizes");
    if (n_nodes == 0) return MGAR_OK;
    MGAR_REQUIRE(rowptr && col && xl && xr && att && alpha && out, "gatv2_fwd: null pointer");
    if (C % 64 != 0) { set_error("gatv2_fwd: C must be a multiple of 64"); return MGAR_EUNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(ceil_div((long long)n_nodes * H, 4));
    // minimum bytes: x_l, x_r read and out written once (3 * n * H * C floats; the per-edge re-reads of x_l rows are cache hits) +
    // alpha (E, H) -- E is on the device, bounded here by n^2; SURVEY.md section 8d edge phase: 6 * E * H * C flop
    KtScope kt(KT_GATV2_FWD, st, 4.0 * (3.0 * n_nodes * H * C + (double)n_nodes * n_nodes * H), 6.0 * n_nodes * n_nodes * (double)H * C);
    GAT_DISPATCH(gatv2_fwd_kernel, n_nodes, H, rowptr, col, xl, xr, att, slope, edge_scale, alpha, out);
    return check_launch("gatv2_fwd: launch failed");
}

extern "C" __attribute__((visibility("default"))) int mgar_gatv2_bwd(int n_nodes, int H, int C, const int *rowptr,
                                                                    const int *col, const int *src_rowptr,
                                                                    const int *src_edge, const int *src_dst, const float *xl,
                                                                    const float *xr, const float *att, float slope,
                                                                    const float *edge_scale, const float *alpha,
                                                                    const float *grad_out, float *workspace, float *grad_xl,
                                                                    float *grad_xr, float *grad_att, void *stream) {
    MGAR_REQUIRE(n_nodes >= 0 && H > 0 && C > 0, "gatv2_bwd: bad sizes");
    if (n_nodes == 0) return MGAR_OK;
    MGAR_REQUIRE(rowptr && col && src_rowptr && src_edge && src_dst && xl && xr && att && alpha && grad_out && workspace && grad_xl &&
                     grad_xr && grad_att,
                 "gatv2_bwd: null pointer");
    if (C % 64 != 0) { set_error("gatv2_bwd: C must be a multiple of 64"); return MGAR_EUNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(ceil_div((long long)n_nodes * H, 4));
    // workspace: att_part (n, H, C) floats, then de (E, H) floats (E is not known here: the caller sized it with
    // mgar_gatv2_bwd_workspace_floats)
    float *att_part = workspace, *de = workspace + (size_t)n_nodes * H * C;
    {
        // one wave per (node, head): 2 sweeps over the incoming rows of x_l (cache-resident), alpha / edge_scale / de per edge
        // x_l, x_r, grad_out read, grad_x_l, grad_x_r, att_part written (6 * n * H * C floats) + alpha / de per edge (E <= n^2);
        // per edge and head: 2 dots + dz + the aggregation terms ~ 12 * C flop
        KtScope kt(KT_GATV2_BWD, st, 4.0 * (6.0 * n_nodes * H * C + 3.0 * n_nodes * n_nodes * H), 12.0 * n_nodes * n_nodes * (double)H * C);
        GAT_DISPATCH(gatv2_bwd_target_kernel, n_nodes, H, rowptr, col, xl, xr, att, slope, edge_scale, alpha, grad_out, de, grad_xr, att_part);
        GAT_DISPATCH(gatv2_bwd_source_kernel, n_nodes, H, src_rowptr, src_edge, src_dst, xl, xr, att, slope, edge_scale, alpha, de, grad_out,
                     att_part, grad_xl, grad_att);
    }
    return check_launch("gatv2_bwd: launch failed");
}

extern "C" __attribute__((visibility("default"))) long long mgar_gatv2_bwd_workspace_floats(int n_nodes, int H, int C, int n_edges) {
    if (n_nodes < 0 || H <= 0 || C <= 0 || n_edges < 0) return -1;
    return (long long)n_nodes * H * C + (long long)n_edges * H;
}
