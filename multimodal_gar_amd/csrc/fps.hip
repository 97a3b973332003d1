// fps.hip -- farthest point sampling (batch + stack layouts) for gfx950.
//
// Replaces  pcdet/ops/pointnet2/pointnet2_batch/src/sampling_gpu.cu:101-259
//           pcdet/ops/pointnet2/pointnet2_stack/src/sampling_gpu.cu:188-348
//
// The reference keeps the cloud and the running min-distance array `temp` in global
// memory and re-reads both (20 B/point) in every one of the m-1 serial rounds, then
// reduces through a 10-step shared-memory tree with a __syncthreads per step.
//
// MI355X design: one workgroup per cloud (the rounds are a serial dependency chain), and
//   * the whole cloud AND temp live in VGPRs for the entire kernel: up to 16 points per
//     lane x 1024 lanes (4 VGPRs per point); HBM is touched once on the way in and once
//     on the way out (12N + 4N + 4N + 4M bytes per cloud);
//   * the per-round arg-max is a DPP (row_shr / row_bcast) reduction inside each wave, one
//     32-byte LDS slot per wave, ONE barrier per round (slots are double-buffered), then a
//     16-lane DPP reduction of the slots; the winner's coordinates travel with its key so
//     no lane ever goes back to memory for the next round's reference point.
//
// Tie rule (bit-exact index parity).  The reference's result depends on its launch
// geometry: thread tid scans k = tid, tid+bs, ... keeping the FIRST maximum (strict >),
// and the tree "__update" keeps the lower slot unless the upper is strictly greater
// (sampling_gpu.cu:93-98).  Unrolling the tree shows the survivor among equal maxima is
// the candidate with the smallest  bitreverse_{log2 bs}(k mod bs), then the smallest k.
// So every point gets the priority  p(k) = bitrev(k mod bs) * ceil(n/bs) + k / bs, lanes
// visit their points in ascending p, and all reductions take max over (value, -p): any
// reduction order then gives the reference's answer.  bs = min(2^floor(log2 n), 1024)
// for the batch op (computed on the host with the reference's own double-precision
// expression, cuda_utils.h:10-14) and 1024 for the stack op.
#include <cmath>
#include "common.hpp"

namespace mgar {

template <int N> struct VecOf {
    typedef float __attribute__((ext_vector_type(N))) f;
    typedef int __attribute__((ext_vector_type(N))) i;
};

// ---- DPP helpers (gfx9 encodings) -------------------------------------------------
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_umax64(unsigned long long key) {
    const unsigned lo = (unsigned)key, hi = (unsigned)(key >> 32);
    // lanes without a valid source (or outside ROW_MASK) receive 0 = the identity of umax
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, CTRL, ROW_MASK, 0xF, false);
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, CTRL, ROW_MASK, 0xF, false);
    const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
    return o > key ? o : key;
}

// max over each 16-lane row, result in lane 15 of the row
__device__ __forceinline__ unsigned long long row_umax64(unsigned long long key) {
    key = dpp_umax64<DPP_ROW_SHR1, 0xF>(key);
    key = dpp_umax64<DPP_ROW_SHR2, 0xF>(key);
    key = dpp_umax64<DPP_ROW_SHR4, 0xF>(key);
    key = dpp_umax64<DPP_ROW_SHR8, 0xF>(key);
    return key;
}

// max over the wave, returned wave-uniform
__device__ __forceinline__ unsigned long long wave_umax64(unsigned long long key) {
    key = row_umax64(key);
    key = dpp_umax64<DPP_ROW_BCAST15, 0xA>(key);
    key = dpp_umax64<DPP_ROW_BCAST31, 0xC>(key);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)key, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(key >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

// Single-instruction min / max.  fminf()/fmaxf() make hipcc emit a canonicalising
// v_max_f32 x,x in front of every v_min/v_max (IEEE mode sNaN quieting): +1 VALU per
// point per round in a loop that is VALU-bound.  Inputs here are never NaN.
__device__ __forceinline__ float vmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// order-preserving float -> uint
__device__ __forceinline__ unsigned ordered_bits(float v) {
    const unsigned b = __float_as_uint(v);
    return b ^ ((unsigned)((int)b >> 31) | 0x80000000u);
}

struct FpsArgs {
    int stack;           // 0 = batch layout, 1 = stacked layout
    int n_batch, m_batch, bs_log2_batch;
    const float *points;
    float *temp;
    const int *xyz_batch_cnt;
    int *idx;
    const int *num_sampled;
};

struct alignas(32) FpsSlot {
    unsigned key_lo, key_hi;
    float x, y, z;
    int k;
    int pad0, pad1;
};

// slot s of thread t  ->  point index (ascending priority inside a thread); see header.
//   R = max(bs/T, 1) residues per thread, L = ceil(n/bs) points per residue
__device__ __forceinline__ int slot_to_k(int t, int s, int T, int bs_log2, int R_log2, int L) {
    const int ap = s / L, j = s - ap * L;
    const int a = R_log2 ? (int)(__brev((unsigned)ap) >> (32 - R_log2)) : 0;
    return t + a * T + (j << bs_log2);
}

template <int T, int PPT>
__global__ __launch_bounds__(T) void fps_kernel(FpsArgs A) {
    constexpr int NW = T / kWave;
    __shared__ FpsSlot slots[2][16];

    const int cloud = blockIdx.x;
    int start, ostart, n, m, bs_log2, idx_off;
    if (A.stack) {
        start = 0; ostart = 0;
        for (int i = 0; i < cloud; ++i) { start += A.xyz_batch_cnt[i]; ostart += A.num_sampled[i]; }
        n = A.xyz_batch_cnt[cloud];
        m = A.num_sampled[cloud];
        bs_log2 = 10;
        idx_off = start;  // stack op writes global row ids (sampling_gpu.cu:313-315)
    } else {
        start = cloud * A.n_batch; ostart = cloud * A.m_batch;
        n = A.n_batch; m = A.m_batch; bs_log2 = A.bs_log2_batch;
        idx_off = 0;
    }
    if (m <= 0 || n <= 0) return;

    const float *__restrict__ P = A.points + (size_t)start * 3;
    float *__restrict__ temp = A.temp + start;
    int *__restrict__ out = A.idx + ostart;

    const int tid = threadIdx.x;
    const int bs = 1 << bs_log2;
    const int L = (n + bs - 1) >> bs_log2;
    int R_log2 = 0;
    while ((T << R_log2) < bs) ++R_log2;
    const int used = L << R_log2;       // slots that can hold a real point (<= PPT by dispatch)
    const bool lane_active = tid < bs;  // T > bs only for tiny clouds
    // Priority of slot s of this lane is  prio_base + s  (see header: the bit-reversed
    // residue of k = tid + a*T splits into bitrev(tid) and the slot's residue rank).
    const int t_log2 = (T < bs ? __builtin_ctz((unsigned)T) : bs_log2);
    const unsigned rev_t = t_log2 ? (__brev((unsigned)tid) >> (32 - t_log2)) : 0u;
    const unsigned prio_base = (rev_t << R_log2) * (unsigned)L;

    typename VecOf<PPT>::f X, Y, Z, D;
    typename VecOf<PPT>::i K;
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int k = slot_to_k(tid, s, T, bs_log2, R_log2, L);
        const bool ok = lane_active && s < used && k < n;
        K[s] = ok ? k : -1;
        X[s] = ok ? P[k * 3 + 0] : 0.f;
        Y[s] = ok ? P[k * 3 + 1] : 0.f;
        Z[s] = ok ? P[k * 3 + 2] : 0.f;
        D[s] = ok ? temp[k] : -1.f;  // padding can never beat a real candidate (d2 >= 0)
    }

    float x1 = P[0], y1 = P[1], z1 = P[2];
    if (tid == 0) out[0] = idx_off;

    const int lane = tid & 63, wave = tid >> 6;

    for (int j = 1; j < m; ++j) {
        // ---- per-lane scan in ascending priority, strict '>' keeps the first maximum ----
        float best = -1.f;
        int bslot = 0;
        // whole-vector arithmetic: the backend pairs the lanes of the register arrays into
        // v_pk_add/mul/fma_f32 (2 points per instruction); same operations, same rounding as d2_of
        const typename VecOf<PPT>::f dx = X - x1, dy = Y - y1, dz = Z - z1;
        const typename VecOf<PPT>::f dd = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const float d2 = vmin(dd[s], D[s]);
            D[s] = d2;
            bslot = d2 > best ? s : bslot;
            best = vmax(best, d2);
        }
        const unsigned long long mykey = ((unsigned long long)ordered_bits(best) << 32) |
                                         (unsigned long long)(0xFFFFFFFFu - (prio_base + (unsigned)bslot));

        // ---- wave arg-max; the unique winning lane publishes key + coordinates ----
        const unsigned long long wkey = wave_umax64(mykey);
        FpsSlot *buf = slots[j & 1];
        if (mykey == wkey) {  // exactly one lane: priorities are unique
            FpsSlot sl;
            sl.key_lo = (unsigned)wkey; sl.key_hi = (unsigned)(wkey >> 32);
            // one active lane => its slot number is wave-uniform: index the register
            // arrays through an SGPR instead of a 15-deep select chain per array
            const int us = __builtin_amdgcn_readfirstlane(bslot);
            sl.x = X[us]; sl.y = Y[us]; sl.z = Z[us];
            sl.k = K[us]; sl.pad0 = 0; sl.pad1 = 0;
            buf[wave] = sl;
        }
        int win_k;
        if (NW > 1) {
            __syncthreads();
            // ---- 16-lane reduction over the per-wave slots (every row does the same) ----
            const int w = lane & 15;
            const FpsSlot sl = buf[w < NW ? w : 0];
            const unsigned long long skey = w < NW ? (((unsigned long long)sl.key_hi << 32) | sl.key_lo) : 0ull;
            const unsigned long long rkey = row_umax64(skey);
            const unsigned rlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)rkey, 15);
            const unsigned rhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(rkey >> 32), 15);
            const unsigned long long gkey = ((unsigned long long)rhi << 32) | rlo;
            const unsigned long long hit = __builtin_amdgcn_ballot_w64(skey == gkey && w < NW);
            const int src = __builtin_ctzll(hit);  // first lane holding the winning slot
            x1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sl.x), src));
            y1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sl.y), src));
            z1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sl.z), src));
            win_k = __builtin_amdgcn_readlane(sl.k, src);
        } else {
            // single wave: the slot written above is already the global winner; LDS ops of
            // one wave execute in order, so the read below sees the write
            __builtin_amdgcn_wave_barrier();
            const FpsSlot sl = buf[0];
            x1 = sl.x; y1 = sl.y; z1 = sl.z; win_k = sl.k;
        }
        if (tid == 0) out[j] = win_k + idx_off;
    }

    // ---- temp is an in/out argument of the reference op: hand the final values back ----
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        if (K[s] >= 0) temp[K[s]] = D[s];
    }
}

// ---- spatially pruned variant -------------------------------------------------------------------
// Same result, fewer distance updates.  The host hands in `perm`, the points of each cloud in Morton
// order, so the 64 * PPT points of a wave form a compact cluster with a small bounding box.  A new
// sample s can lower the running distance D[p] of a point only if |s - p|^2 < D[p]; with
// wave_max = max D over the wave (cached: D never grows) and dmin2 = squared distance from s to the
// wave's box, dmin2 >= wave_max proves that nothing in the wave changes -- the wave skips the update
// and re-publishes its cached candidate.  After a few hundred samples the reach sqrt(max D) is a
// fraction of the scene and 1-3 of the 16 waves are active per round; the distance update was 58 % of
// a round (measured: 0.95 of 1.63 us at N = 16 384).  Conservative by a 1e-5 margin on dmin2, so
// rounding can only cause a superfluous update, never a missed one.
// Priorities can no longer be derived from (lane, slot): each slot carries p(k); slots are sorted by
// priority inside a lane at set-up (bitonic network in registers) so the strict '>' scan still keeps
// the first maximum, and the 64-bit keys (value, ~p) make every reduction order give the reference's
// answer, exactly as above.
template <int PPT>
__global__ __launch_bounds__(1024) void fps_pruned_kernel(FpsArgs A, const int *__restrict__ perm) {
    constexpr int NW = 16;
    __shared__ FpsSlot slots[2][16];
    __shared__ FpsSlot cached[16];   // each wave's last candidate (re-published while the wave is skipped)
    const int cloud = blockIdx.x;
    const int n = A.n_batch, m = A.m_batch;
    if (m <= 0 || n <= 0) return;
    const int L = (n + 1023) >> 10;  // bs = 1024 (dispatch guarantees n >= 1024)
    const float *__restrict__ P = A.points + (size_t)cloud * n * 3;
    float *__restrict__ temp = A.temp + (size_t)cloud * n;
    int *__restrict__ out = A.idx + (size_t)cloud * m;
    const int *__restrict__ pm = perm + (size_t)cloud * n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    typename VecOf<PPT>::f X, Y, Z, D;
    typename VecOf<PPT>::i PR;   // priority p(k); k itself is recovered from it (p is a bijection): saves 16 VGPRs
    constexpr int NOPT = 0x7FFFFFFF;
    auto k_of = [&](int pr) { const int hi = pr / L; return (int)(__brev((unsigned)hi) >> 22) + (pr - hi * L) * 1024; };
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int q = tid * PPT + s;
        const bool ok = q < n;
        const int k = ok ? pm[q] : -1;
        X[s] = ok ? P[k * 3 + 0] : 0.f;
        Y[s] = ok ? P[k * 3 + 1] : 0.f;
        Z[s] = ok ? P[k * 3 + 2] : 0.f;
        D[s] = ok ? temp[k] : -1.f;
        PR[s] = ok ? (int)(__brev((unsigned)(k & 1023)) >> 22) * L + (k >> 10) : NOPT;
    }
    // sort the lane's slots by ascending priority
#pragma unroll
    for (int size = 2; size <= PPT; size <<= 1)
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1)
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const int j = i ^ stride;
                if (j > i) {
                    const bool up = (i & size) == 0;
                    const bool sw = (PR[i] > PR[j]) == up;
                    const float tx = X[i], ty = Y[i], tz = Z[i], td = D[i];
                    const int tp = PR[i];
                    X[i] = sw ? X[j] : tx; X[j] = sw ? tx : X[j];
                    Y[i] = sw ? Y[j] : ty; Y[j] = sw ? ty : Y[j];
                    Z[i] = sw ? Z[j] : tz; Z[j] = sw ? tz : Z[j];
                    D[i] = sw ? D[j] : td; D[j] = sw ? td : D[j];
                    PR[i] = sw ? PR[j] : tp; PR[j] = sw ? tp : PR[j];
                }
            }
    // bounding box of the wave's real points
    const float inf = __builtin_inff();
    float lx0 = inf, ly0 = inf, lz0 = inf, lx1 = -inf, ly1 = -inf, lz1 = -inf;
#pragma unroll
    for (int s = 0; s < PPT; ++s)
        if (PR[s] != NOPT) {
            lx0 = fminf(lx0, X[s]); lx1 = fmaxf(lx1, X[s]);
            ly0 = fminf(ly0, Y[s]); ly1 = fmaxf(ly1, Y[s]);
            lz0 = fminf(lz0, Z[s]); lz1 = fmaxf(lz1, Z[s]);
        }
    const float bx0 = -wave_max(-lx0), by0 = -wave_max(-ly0), bz0 = -wave_max(-lz0);
    const float bx1 = wave_max(lx1), by1 = wave_max(ly1), bz1 = wave_max(lz1);

    float x1 = P[0], y1 = P[1], z1 = P[2];
    if (tid == 0) out[0] = 0;

    float c_best = inf;   // max D over the wave as of its last update (wave-uniform); +inf = never updated
    bool other_stale = false;   // slots[] is double-buffered: after an update the other buffer still holds the old record

    for (int j = 1; j < m; ++j) {
        FpsSlot *buf = slots[j & 1];
        const float ex = fmaxf(fmaxf(bx0 - x1, x1 - bx1), 0.f), ey = fmaxf(fmaxf(by0 - y1, y1 - by1), 0.f);
        const float ez = fmaxf(fmaxf(bz0 - z1, z1 - bz1), 0.f);
        const float dmin2 = (ex * ex + ey * ey + ez * ez) * 0.99999f;
        if (j == 1 || dmin2 < c_best) {  // wave-uniform: some point of this wave may move (round 1: everybody publishes)
            float best = -1.f;
            int bslot = 0, bprio = 0x7FFFFFFF;
            const typename VecOf<PPT>::f dx = X - x1, dy = Y - y1, dz = Z - z1;
            const typename VecOf<PPT>::f dd = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
#pragma unroll
            for (int s = 0; s < PPT; ++s) {
                const float d2 = vmin(dd[s], D[s]);
                D[s] = d2;
                const bool gt = d2 > best;
                bslot = gt ? s : bslot;
                bprio = gt ? PR[s] : bprio;
                best = vmax(best, d2);
            }
            const unsigned long long mykey = ((unsigned long long)ordered_bits(best) << 32) |
                                             (unsigned long long)(0xFFFFFFFFu - (unsigned)bprio);
            const unsigned long long wkey = wave_umax64(mykey);
            if (mykey == wkey) {  // exactly one lane: priorities are unique
                FpsSlot sl;
                sl.key_lo = (unsigned)wkey; sl.key_hi = (unsigned)(wkey >> 32);
                const int us = __builtin_amdgcn_readfirstlane(bslot);
                sl.x = X[us]; sl.y = Y[us]; sl.z = Z[us];
                sl.k = k_of(PR[us]); sl.pad0 = __float_as_int(best); sl.pad1 = 0;
                buf[wave] = sl;
                cached[wave] = sl;
            }
            // the wave maximum is the high word of the key: invert ordered_bits
            const unsigned ob = (unsigned)(wkey >> 32);
            c_best = __uint_as_float(ob ^ ((ob >> 31) ? 0x80000000u : 0xFFFFFFFFu));
            other_stale = true;
        } else if (other_stale) {  // first skipped round after an update: bring the other buffer up to date, once
            if (lane == 0) buf[wave] = cached[wave];
            other_stale = false;
        }
        __syncthreads();
        // ---- 16-lane reduction over the per-wave slots (every row does the same) ----
        const int w = lane & 15;
        const FpsSlot sl = buf[w];
        const unsigned long long skey = ((unsigned long long)sl.key_hi << 32) | sl.key_lo;
        const unsigned long long rkey = row_umax64(skey);
        const unsigned rlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)rkey, 15);
        const unsigned rhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(rkey >> 32), 15);
        const unsigned long long gkey = ((unsigned long long)rhi << 32) | rlo;
        const unsigned long long hit = __builtin_amdgcn_ballot_w64(skey == gkey);
        const int src = __builtin_ctzll(hit);
        x1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sl.x), src));
        y1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sl.y), src));
        z1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sl.z), src));
        const int win_k = __builtin_amdgcn_readlane(sl.k, src);
        if (tid == 0) out[j] = win_k;
    }
#pragma unroll
    for (int s = 0; s < PPT; ++s)
        if (PR[s] != NOPT) temp[k_of(PR[s])] = D[s];
    (void)NW;
}

// Morton codes (10 bits per axis, one common scale = the cloud's largest extent) of every point of
// every cloud; the host sorts them to obtain `perm`.  grid (b), block 1024
__device__ __forceinline__ unsigned spread3(unsigned v) {
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__global__ __launch_bounds__(1024) void morton_codes_kernel(const float *__restrict__ xyz, int n, int *__restrict__ codes) {
    __shared__ float red[6][16];
    const float *P = xyz + (size_t)blockIdx.x * n * 3;
    const float inf = __builtin_inff();
    float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
    for (int k = threadIdx.x; k < n; k += 1024)
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = P[k * 3 + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float l = -wave_max(-lo[a]), h = wave_max(hi[a]);
        if (lane == 0) { red[a][wave] = l; red[3 + a][wave] = h; }
    }
    __syncthreads();
    float ext = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = inf, h = -inf;
        for (int w = 0; w < 16; ++w) { l = fminf(l, red[a][w]); h = fmaxf(h, red[3 + a][w]); }
        lo[a] = l;
        ext = fmaxf(ext, h - l);
    }
    const float scale = ext > 0.f ? 1023.f / ext : 0.f;
    for (int k = threadIdx.x; k < n; k += 1024) {
        unsigned q[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float t = (P[k * 3 + a] - lo[a]) * scale;
            q[a] = (unsigned)fminf(fmaxf(t, 0.f), 1023.f);
        }
        codes[(size_t)blockIdx.x * n + k] = (int)(spread3(q[0]) | (spread3(q[1]) << 1) | (spread3(q[2]) << 2));
    }
}

// Fallback for clouds that do not fit the register file (n > 16384): same reduction
// machinery, but coordinates and temp are streamed from memory every round like the
// reference does.  Correctness path only.
__global__ __launch_bounds__(1024) void fps_stream_kernel(FpsArgs A) {
    constexpr int NW = 16;
    __shared__ FpsSlot slots[2][16];
    const int cloud = blockIdx.x;
    int start, ostart, n, m, bs_log2, idx_off;
    if (A.stack) {
        start = 0; ostart = 0;
        for (int i = 0; i < cloud; ++i) { start += A.xyz_batch_cnt[i]; ostart += A.num_sampled[i]; }
        n = A.xyz_batch_cnt[cloud]; m = A.num_sampled[cloud]; bs_log2 = 10; idx_off = start;
    } else {
        start = cloud * A.n_batch; ostart = cloud * A.m_batch;
        n = A.n_batch; m = A.m_batch; bs_log2 = A.bs_log2_batch; idx_off = 0;
    }
    if (m <= 0 || n <= 0) return;
    const float *__restrict__ P = A.points + (size_t)start * 3;
    float *__restrict__ temp = A.temp + start;
    int *__restrict__ out = A.idx + ostart;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bs = 1 << bs_log2;
    const int L = (n + bs - 1) >> bs_log2;
    float x1 = P[0], y1 = P[1], z1 = P[2];
    if (tid == 0) out[0] = idx_off;
    for (int j = 1; j < m; ++j) {
        float best = -1.f, bx = 0.f, by = 0.f, bz = 0.f;
        int bk = 0;
        if (tid < bs) {
            for (int k = tid; k < n; k += bs) {  // bs == T here (n > 16384 => bs = 1024)
                const float x2 = P[k * 3 + 0], y2 = P[k * 3 + 1], z2 = P[k * 3 + 2];
                const float d = d2_of(x2 - x1, y2 - y1, z2 - z1);
                const float d2 = vmin(d, temp[k]);
                temp[k] = d2;
                if (d2 > best) { best = d2; bk = k; bx = x2; by = y2; bz = z2; }
            }
        }
        const unsigned prio = (__brev((unsigned)(bk & (bs - 1))) >> (32 - bs_log2)) * (unsigned)L + (unsigned)(bk >> bs_log2);
        const unsigned long long mykey = ((unsigned long long)ordered_bits(best) << 32) | (unsigned long long)(0xFFFFFFFFu - prio);
        const unsigned long long wkey = wave_umax64(mykey);
        FpsSlot *buf = slots[j & 1];
        if (mykey == wkey) {
            FpsSlot sl;
            sl.key_lo = (unsigned)wkey; sl.key_hi = (unsigned)(wkey >> 32);
            sl.x = bx; sl.y = by; sl.z = bz; sl.k = bk; sl.pad0 = 0; sl.pad1 = 0;
            buf[wave] = sl;
        }
        __syncthreads();
        const int w = lane & 15;
        FpsSlot sl = buf[w < NW ? w : 0];
        unsigned long long skey = ((unsigned long long)sl.key_hi << 32) | sl.key_lo;
        unsigned long long rkey = row_umax64(skey);
        const unsigned rlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)rkey, 15);
        const unsigned rhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(rkey >> 32), 15);
        const unsigned long long gkey = ((unsigned long long)rhi << 32) | rlo;
        const int src = __builtin_ctzll(__builtin_amdgcn_ballot_w64(skey == gkey));
        x1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sl.x), src));
        y1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sl.y), src));
        z1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sl.z), src));
        const int win_k = __builtin_amdgcn_readlane(sl.k, src);
        if (tid == 0) out[j] = win_k + idx_off;
    }
}

// Clouds of up to 65 536 points (BASELINE config c5): the coordinates do not fit the register file, but the running minimum
// distances do (64 per lane).  The coordinates are streamed every sample, as the reference does -- but as fully coalesced
// 16-byte loads into a double-buffered LDS stage of 4 096 points (the next chunk in flight while the current one is scanned),
// from which every lane reads its own points (stride 3 words: conflict-free).  fps_stream_kernel's 4-byte loads at a 12-byte
// lane stride spend 28 us per sample at n = 65 536 in the texture path alone.  Same scan order, strict comparison, tie rule
// and reduction; temp is read once and written back at the end.  Needs n % 4 == 0 (16-byte aligned clouds).
constexpr int FS_CHUNK = 4096;                                            // points per LDS stage
template <int PPT>
__global__ __launch_bounds__(1024) void fps_stream_reg_kernel(FpsArgs A) {
    constexpr int NW = 16, NCH = PPT / 4;                                 // chunks of 4 096 points = 4 per lane
    __shared__ FpsSlot slots[2][16];
    __shared__ __attribute__((aligned(16))) float stage[2][FS_CHUNK * 3];
    const int cloud = blockIdx.x;
    const int start = cloud * A.n_batch, ostart = cloud * A.m_batch;      // batch flavour only (bs = 1024)
    const int n = A.n_batch, m = A.m_batch, bs_log2 = 10;
    if (m <= 0 || n <= 0) return;
    const float *__restrict__ P = A.points + (size_t)start * 3;
    const float4 *__restrict__ P4 = reinterpret_cast<const float4 *>(P);
    const int n4 = n * 3 / 4;                                             // float4 words of the cloud
    float *__restrict__ temp = A.temp + start;
    int *__restrict__ out = A.idx + ostart;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bs = 1 << bs_log2;
    const int L = (n + bs - 1) >> bs_log2;
    float t[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int k = tid + i * 1024;
        t[i] = k < n ? temp[k] : -1.f;                                    // -1: never larger than `best`
    }
    float x1 = P[0], y1 = P[1], z1 = P[2];
    if (tid == 0) out[0] = 0;
    float4 pre[3];                                                        // the chunk in flight: 3 072 float4 over 1 024 lanes
    int tv = tid;                                                         // re-materialised per sample (see below)
    auto fetch = [&](int c) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int e = c * (FS_CHUNK * 3 / 4) + tv + q * 1024;
            pre[q] = e < n4 ? P4[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    fetch(0);
    for (int j = 1; j < m; ++j) {
        float best = -1.f, bx = 0.f, by = 0.f, bz = 0.f;
        int bk = 0;
        asm volatile("" : "+v"(tv));       // keeps the 3 * NCH load addresses from being hoisted out of this loop (they would spill)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            float4 *sg = reinterpret_cast<float4 *>(stage[c & 1]);
#pragma unroll
            for (int q = 0; q < 3; ++q) sg[tid + q * 1024] = pre[q];
            __syncthreads();                                              // also: every wave is done with the other buffer's previous chunk
            fetch(c + 1 < NCH ? c + 1 : 0);                              // in flight while this chunk is scanned
            const float *sf = stage[c & 1];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = tid + (4 * c + i) * 1024;
                const float px = sf[(tid + i * 1024) * 3 + 0], py = sf[(tid + i * 1024) * 3 + 1], pz = sf[(tid + i * 1024) * 3 + 2];
                const float d = d2_of(px - x1, py - y1, pz - z1);
                const float d2 = k < n ? vmin(d, t[4 * c + i]) : -1.f;
                t[4 * c + i] = d2;
                if (d2 > best) { best = d2; bk = k; bx = px; by = py; bz = pz; }
            }
        }
        const unsigned prio = (__brev((unsigned)(bk & (bs - 1))) >> (32 - bs_log2)) * (unsigned)L + (unsigned)(bk >> bs_log2);
        const unsigned long long mykey = ((unsigned long long)ordered_bits(best) << 32) | (unsigned long long)(0xFFFFFFFFu - prio);
        const unsigned long long wkey = wave_umax64(mykey);
        FpsSlot *buf = slots[j & 1];
        if (mykey == wkey) {
            FpsSlot sl;
            sl.key_lo = (unsigned)wkey; sl.key_hi = (unsigned)(wkey >> 32);
            sl.x = bx; sl.y = by; sl.z = bz; sl.k = bk; sl.pad0 = 0; sl.pad1 = 0;
            buf[wave] = sl;
        }
        __syncthreads();
        const int w = lane & 15;
        FpsSlot sl = buf[w < NW ? w : 0];
        unsigned long long skey = ((unsigned long long)sl.key_hi << 32) | sl.key_lo;
        unsigned long long rkey = row_umax64(skey);
        const unsigned rlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)rkey, 15);
        const unsigned rhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(rkey >> 32), 15);
        const unsigned long long gkey = ((unsigned long long)rhi << 32) | rlo;
        const int src = __builtin_ctzll(__builtin_amdgcn_ballot_w64(skey == gkey));
        x1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sl.x), src));
        y1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sl.y), src));
        z1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sl.z), src));
        const int win_k = __builtin_amdgcn_readlane(sl.k, src);
        if (tid == 0) out[j] = win_k;
    }
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int k = tid + i * 1024;
        if (k < n) temp[k] = t[i];
    }
}

// ---- clouds of 16 385 .. 65 536 points with bucketed pruning (round 3; BASELINE config c5) ------------------------------
// fps_stream_reg_kernel re-reads the whole cloud for every sample (254 of 473 ms of the c5 step).  Here the pruning of
// fps_pruned_kernel is applied at the granularity of UNITS of 256 points (Morton order: a unit is a compact cluster):
//   * the running minimum distances of all points stay in registers (4 * NU per lane, NU units per wave, 16 waves);
//   * coordinates do NOT fit on chip: they live in a workspace as float4 {x, y, z, p} in sorted order (p = the point's
//     tie-break priority, from which k is recovered) -- a unit is 4 KB contiguous, a lane's 4 points one 64-byte run;
//   * every unit keeps its bounding box and its current maximum distance in the registers of lane u of its wave, and its
//     candidate record {key, x, y, z, k} in LDS.  Per sample a wave tests its NU boxes in one vector step (lane u <-> unit u);
//     only units the new sample can still lower are loaded (4 float4 per lane), updated and re-reduced; a wave with no such
//     unit re-publishes its cached candidate without touching anything.
// After the first few hundred samples 1-3 of the 256 units of a cloud are live per sample, so a sample costs the latency of one
// 4 KB read from L2 / MALL + one unit update + the two reductions instead of a pass over 768 KB.
// Exactness: same distance expression, running minimum and (value, ~p) keys as the other kernels; inside a lane the 4 points of
// a unit are ordered by ascending p (prep kernel), so the strict '>' scan keeps the reference's survivor; the pruning test is
// conservative by the same 1e-5 margin.  Any permutation gives the exact result (tests: Morton, random).
constexpr int FB_UNIT = 256;            // points per unit = 64 lanes x 4
constexpr int FB_NOPT = 0x7FFFFFFF;     // priority of a padding slot

struct alignas(32) FbCand {
    unsigned key_lo, key_hi;
    float x, y, z;
    int k;
    float best;
    int pad;
};

// perm (b, n) -> sorted (b, cap) float4 {x, y, z, p}, groups of 4 consecutive positions ordered by ascending p; padding p = NOPT
__global__ __launch_bounds__(256) void fps_bucket_prep_kernel(const float *__restrict__ points, const int *__restrict__ perm, int n,
                                                              int cap, float4 *__restrict__ sorted) {
    const int cloud = blockIdx.y;
    const int g = blockIdx.x * 256 + threadIdx.x;   // group of 4 positions
    if (g * 4 >= cap) return;
    const int L = (n + 1023) >> 10;
    const float *P = points + (size_t)cloud * n * 3;
    const int *pm = perm + (size_t)cloud * n;
    float4 v[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int q = g * 4 + s;
        if (q < n) {
            const int k = pm[q];
            const int pr = (int)(__brev((unsigned)(k & 1023)) >> 22) * L + (k >> 10);
            v[s] = make_float4(P[k * 3 + 0], P[k * 3 + 1], P[k * 3 + 2], __int_as_float(pr));
        } else {
            v[s] = make_float4(0.f, 0.f, 0.f, __int_as_float(FB_NOPT));
        }
    }
    // 4-element sorting network on p (ascending)
    auto cswap = [&](int i, int j) {
        if (__float_as_int(v[i].w) > __float_as_int(v[j].w)) { const float4 t = v[i]; v[i] = v[j]; v[j] = t; }
    };
    cswap(0, 1); cswap(2, 3); cswap(0, 2); cswap(1, 3); cswap(1, 2);
    float4 *out = sorted + (size_t)cloud * cap + (size_t)g * 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) out[s] = v[s];
}

template <int NU>
__global__ __launch_bounds__(1024) void fps_bucket_kernel(FpsArgs A, const float4 *__restrict__ sorted, int cap) {
    constexpr int NW = 16;
    __shared__ FpsSlot slots[2][16];
    __shared__ FpsSlot cached[16];
    __shared__ FbCand cand[NW][NU];
    const int cloud = blockIdx.x;
    const int n = A.n_batch, m = A.m_batch;
    if (m <= 0 || n <= 0) return;
    const int L = (n + 1023) >> 10;
    const float *__restrict__ P = A.points + (size_t)cloud * n * 3;
    float *__restrict__ temp = A.temp + (size_t)cloud * n;
    int *__restrict__ out = A.idx + (size_t)cloud * m;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // this wave's units (a wave-uniform base: scalar registers) and this lane's 4 points inside a unit (a 32-bit lane offset)
    const float4 *__restrict__ wbase = sorted + (size_t)cloud * cap + (size_t)wave * NU * FB_UNIT;
    const float4 *__restrict__ mine = wbase + lane * 4;
    auto k_of = [&](int pr) { const int hi = pr / L; return (int)(__brev((unsigned)hi) >> 22) + (pr - hi * L) * 1024; };

    float D[NU * 4];
    const float inf = __builtin_inff();
    // per-unit data in lane u: bounding box + cached maximum
    float bx0 = inf, by0 = inf, bz0 = inf, bx1 = -inf, by1 = -inf, bz1 = -inf, cmax = inf;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        float lx0 = inf, ly0 = inf, lz0 = inf, lx1 = -inf, ly1 = -inf, lz1 = -inf;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float4 v = mine[u * FB_UNIT + s];
            const int pr = __float_as_int(v.w);
            const bool ok = pr != FB_NOPT;
            D[u * 4 + s] = ok ? temp[k_of(pr)] : -1.f;
            if (ok) {
                lx0 = fminf(lx0, v.x); lx1 = fmaxf(lx1, v.x);
                ly0 = fminf(ly0, v.y); ly1 = fmaxf(ly1, v.y);
                lz0 = fminf(lz0, v.z); lz1 = fmaxf(lz1, v.z);
            }
        }
        const float ux0 = -wave_max(-lx0), uy0 = -wave_max(-ly0), uz0 = -wave_max(-lz0);
        const float ux1 = wave_max(lx1), uy1 = wave_max(ly1), uz1 = wave_max(lz1);
        if (lane == u) { bx0 = ux0; by0 = uy0; bz0 = uz0; bx1 = ux1; by1 = uy1; bz1 = uz1; }
    }

    float x1 = P[0], y1 = P[1], z1 = P[2];
    if (tid == 0) out[0] = 0;
    bool other_stale = false;

    for (int j = 1; j < m; ++j) {
        FpsSlot *buf = slots[j & 1];
        // lane u tests unit u's box (lanes >= NU never vote)
        const float ex = fmaxf(fmaxf(bx0 - x1, x1 - bx1), 0.f), ey = fmaxf(fmaxf(by0 - y1, y1 - by1), 0.f);
        const float ez = fmaxf(fmaxf(bz0 - z1, z1 - bz1), 0.f);
        const float dmin2 = (ex * ex + ey * ey + ez * ez) * 0.99999f;
        const bool live = lane < NU && (j == 1 || dmin2 < cmax);
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(live);
        if (mask != 0ull) {  // wave-uniform
            unsigned lo = (unsigned)lane * 4u;
            asm volatile("" : "+v"(lo));   // re-materialised per sample: otherwise 2 * NU 64-bit addresses are hoisted out of the loop and spill
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if ((mask >> u) & 1ull) {  // wave-uniform
                    float4 v[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) v[s] = wbase[u * FB_UNIT + lo + s];
                    float best = -1.f, wx = 0.f, wy = 0.f, wz = 0.f;
                    int wp = FB_NOPT;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const float d = d2_of(v[s].x - x1, v[s].y - y1, v[s].z - z1);
                        // a padding slot holds -1 and zero coordinates: min(d, -1) = -1 keeps it out for ever
                        const float d2 = vmin(d, D[u * 4 + s]);
                        D[u * 4 + s] = d2;
                        const bool gt = d2 > best;
                        wx = gt ? v[s].x : wx; wy = gt ? v[s].y : wy; wz = gt ? v[s].z : wz;
                        wp = gt ? __float_as_int(v[s].w) : wp;
                        best = vmax(best, d2);
                    }
                    const unsigned long long mykey = ((unsigned long long)ordered_bits(best) << 32) |
                                                     (unsigned long long)(0xFFFFFFFFu - (unsigned)wp);
                    const unsigned long long wkey = wave_umax64(mykey);
                    const unsigned ob = (unsigned)(wkey >> 32);
                    const float umax = __uint_as_float(ob ^ ((ob >> 31) ? 0x80000000u : 0xFFFFFFFFu));
                    if (mykey == wkey) {  // exactly one lane (priorities are unique; an all-padding unit: lane with the largest ~p)
                        FbCand c;
                        c.key_lo = (unsigned)wkey; c.key_hi = (unsigned)(wkey >> 32);
                        c.x = wx; c.y = wy; c.z = wz;
                        c.k = wp == FB_NOPT ? 0 : k_of(wp);
                        c.best = umax; c.pad = 0;
                        cand[wave][u] = c;
                    }
                    if (lane == u) cmax = umax;
                }
            }
            // wave candidate = best of the NU unit candidates (the untouched ones are still valid: D never grows, and a unit
            // whose box the sample cannot reach keeps its maximum)
            __builtin_amdgcn_wave_barrier();
            const int w = lane & 15;
            const FbCand c = cand[wave][w < NU ? w : 0];
            const unsigned long long ckey = w < NU ? (((unsigned long long)c.key_hi << 32) | c.key_lo) : 0ull;
            const unsigned long long rkey = row_umax64(ckey);
            const unsigned rlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)rkey, 15);
            const unsigned rhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(rkey >> 32), 15);
            const unsigned long long gkey = ((unsigned long long)rhi << 32) | rlo;
            if (lane < 16 && ckey == gkey && w < NU) {  // one lane (keys are unique)
                FpsSlot sl;
                sl.key_lo = c.key_lo; sl.key_hi = c.key_hi; sl.x = c.x; sl.y = c.y; sl.z = c.z; sl.k = c.k; sl.pad0 = 0; sl.pad1 = 0;
                buf[wave] = sl;
                cached[wave] = sl;
            }
            other_stale = true;
        } else if (other_stale) {
            if (lane == 0) buf[wave] = cached[wave];
            other_stale = false;
        }
        __syncthreads();
        const int w = lane & 15;
        const FpsSlot sl = buf[w];
        const unsigned long long skey = ((unsigned long long)sl.key_hi << 32) | sl.key_lo;
        const unsigned long long rkey = row_umax64(skey);
        const unsigned rlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)rkey, 15);
        const unsigned rhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(rkey >> 32), 15);
        const unsigned long long gkey = ((unsigned long long)rhi << 32) | rlo;
        const unsigned long long hit = __builtin_amdgcn_ballot_w64(skey == gkey);
        const int src = __builtin_ctzll(hit);
        x1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sl.x), src));
        y1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sl.y), src));
        z1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sl.z), src));
        const int win_k = __builtin_amdgcn_readlane(sl.k, src);
        if (tid == 0) out[j] = win_k;
    }
    // temp is an in/out argument: hand the final running minima back
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int pr = __float_as_int(mine[u * FB_UNIT + s].w);
            if (pr != FB_NOPT) temp[k_of(pr)] = D[u * 4 + s];
        }
}

template <int T>
static bool launch_fps_t(int ppt_needed, int nclouds, const FpsArgs &A, hipStream_t st) {
    if (ppt_needed <= 1) hipLaunchKernelGGL((fps_kernel<T, 1>), dim3(nclouds), dim3(T), 0, st, A);
    else if (ppt_needed <= 2) hipLaunchKernelGGL((fps_kernel<T, 2>), dim3(nclouds), dim3(T), 0, st, A);
    else if (ppt_needed <= 4) hipLaunchKernelGGL((fps_kernel<T, 4>), dim3(nclouds), dim3(T), 0, st, A);
    else if (ppt_needed <= 8) hipLaunchKernelGGL((fps_kernel<T, 8>), dim3(nclouds), dim3(T), 0, st, A);
    else if (ppt_needed <= 16) hipLaunchKernelGGL((fps_kernel<T, 16>), dim3(nclouds), dim3(T), 0, st, A);
    else return false;
    return true;
}

// n_max: an upper bound of the points in any one cloud; bs_log2: log2 of the reference's
// block size (it fixes the tie rule, not our launch geometry).
static int fps_dispatch(int nclouds, int n_max, int bs_log2, const FpsArgs &A, hipStream_t st) {
    const int bs = 1 << bs_log2;
    const int L = (n_max + bs - 1) / bs;
    // our workgroup: the largest of {1024, 256, 64} lanes that does not exceed bs (so every
    // lane owns whole residues), but never fewer than 64
    const int T = bs >= 1024 ? 1024 : (bs >= 256 ? 256 : 64);
    const int R = bs > T ? bs / T : 1;
    const int need = L * R;  // register slots per lane
    // algorithmic bytes 12N + 4M per cloud, (M-1)*N pair evaluations of 8 flop
    const double tot_n = A.stack ? (double)n_max : (double)nclouds * A.n_batch, tot_m = A.stack ? tot_n / 4 : (double)nclouds * A.m_batch;
    KtScope kt(KT_FPS, st, 12.0 * tot_n + 4.0 * tot_m, A.stack ? 0.0 : 8.0 * nclouds * ((double)A.m_batch - 1) * A.n_batch);
    bool ok;
    if (T == 1024) ok = launch_fps_t<1024>(need, nclouds, A, st);
    else if (T == 256) ok = launch_fps_t<256>(need, nclouds, A, st);
    else ok = launch_fps_t<64>(need, nclouds, A, st);
    if (!ok) {
        // does not fit the register file: only possible with bs == 1024 and n_max > 16384
        if (!A.stack && n_max <= 32768 && n_max % 4 == 0 && (reinterpret_cast<uintptr_t>(A.points) & 15) == 0)
            hipLaunchKernelGGL(fps_stream_reg_kernel<32>, dim3(nclouds), dim3(1024), 0, st, A);
        else if (!A.stack && n_max <= 65536 && n_max % 4 == 0 && (reinterpret_cast<uintptr_t>(A.points) & 15) == 0)
            hipLaunchKernelGGL(fps_stream_reg_kernel<64>, dim3(nclouds), dim3(1024), 0, st, A);
        else hipLaunchKernelGGL(fps_stream_kernel, dim3(nclouds), dim3(1024), 0, st, A);
    }
    return check_launch("fps: launch failed");
}

}  // namespace mgar

using namespace mgar;

extern "C" __attribute__((visibility("default"))) int mgar_fps_batch(int b, int n, int m, const float *points, float *temp, int *idx, void *stream) {
    MGAR_REQUIRE(b >= 0 && n >= 0 && m >= 0, "fps_batch: negative size");
    if (b == 0 || m == 0) return MGAR_OK;
    MGAR_REQUIRE(n > 0, "fps_batch: m > 0 samples requested from an empty cloud");
    MGAR_REQUIRE(points && temp && idx, "fps_batch: null pointer");
    // the reference's block size: pointnet2_batch/src/cuda_utils.h:10-14, same expression
    const int pow_2 = (int)(std::log(static_cast<double>(n)) / std::log(2.0));
    int bs = 1 << pow_2;
    bs = bs > 1024 ? 1024 : (bs < 1 ? 1 : bs);
    int bs_log2 = 0;
    while ((1 << bs_log2) < bs) ++bs_log2;
    FpsArgs A{0, n, m, bs_log2, points, temp, nullptr, idx, nullptr};
    return fps_dispatch(b, n, bs_log2, A, (hipStream_t)stream);
}

extern "C" __attribute__((visibility("default"))) int mgar_fps_stack(int batch_size, int N, const float *points, float *temp, const int *xyz_batch_cnt,
                              int *idx, const int *num_sampled_points, void *stream) {
    MGAR_REQUIRE(batch_size >= 0 && N >= 0, "fps_stack: negative size");
    if (batch_size == 0 || N == 0) return MGAR_OK;
    MGAR_REQUIRE(points && temp && idx && xyz_batch_cnt && num_sampled_points, "fps_stack: null pointer");
    FpsArgs A{1, 0, 0, 10, points, temp, xyz_batch_cnt, idx, num_sampled_points};
    // per-cloud sizes live on the device; N (their sum) bounds every one of them
    return fps_dispatch(batch_size, N, 10, A, (hipStream_t)stream);
}

// Morton codes of every point (sort them per cloud to obtain the `perm` of mgar_fps_batch_perm).
extern "C" __attribute__((visibility("default"))) int mgar_morton_codes(int b, int n, const float *points, int *codes, void *stream) {
    MGAR_REQUIRE(b >= 0 && n >= 0, "morton_codes: negative size");
    if ((long long)b * n == 0) return MGAR_OK;
    MGAR_REQUIRE(points && codes, "morton_codes: null pointer");
    hipLaunchKernelGGL(morton_codes_kernel, dim3(b), dim3(1024), 0, (hipStream_t)stream, points, n, codes);
    return check_launch("morton_codes: launch failed");
}

// mgar_fps_batch with spatial pruning: perm (b, n) lists each cloud's point indices in a spatially
// coherent order (any permutation gives the same, exact result; a Morton order makes it fast).
extern "C" __attribute__((visibility("default"))) int mgar_fps_batch_perm(int b, int n, int m, const float *points, float *temp, const int *perm,
                                                                         int *idx, void *stream) {
    MGAR_REQUIRE(b >= 0 && n >= 0 && m >= 0, "fps_batch_perm: negative size");
    if (b == 0 || m == 0) return MGAR_OK;
    MGAR_REQUIRE(n > 0, "fps_batch_perm: m > 0 samples requested from an empty cloud");
    MGAR_REQUIRE(points && temp && idx && perm, "fps_batch_perm: null pointer");
    if (n < 1024 || n > 16384) {
        set_error("fps_batch_perm: needs 1024 <= n <= 16384 (use mgar_fps_batch)");
        return MGAR_EUNSUPPORTED;
    }
    FpsArgs A{0, n, m, 10, points, temp, nullptr, idx, nullptr};
    hipStream_t st = (hipStream_t)stream;
    KtScope kt(KT_FPS, st, (double)b * (12.0 * n + 4.0 * m), 8.0 * b * ((double)m - 1) * n);
    const int ppt = (n + 1023) / 1024;
    if (ppt <= 2) hipLaunchKernelGGL((fps_pruned_kernel<2>), dim3(b), dim3(1024), 0, st, A, perm);
    else if (ppt <= 4) hipLaunchKernelGGL((fps_pruned_kernel<4>), dim3(b), dim3(1024), 0, st, A, perm);
    else if (ppt <= 8) hipLaunchKernelGGL((fps_pruned_kernel<8>), dim3(b), dim3(1024), 0, st, A, perm);
    else hipLaunchKernelGGL((fps_pruned_kernel<16>), dim3(b), dim3(1024), 0, st, A, perm);
    return check_launch("fps_batch_perm: launch failed");
}

// Clouds of 16 385 .. 65 536 points, any permutation `perm` (a Morton order is fast): see fps_bucket_kernel.
// workspace: mgar_fps_batch_buckets_workspace_floats(b, n) floats, 16-byte aligned.
extern "C" __attribute__((visibility("default"))) long long mgar_fps_batch_buckets_workspace_floats(int b, int n) {
    if (b < 0 || n < 0) return -1;
    const int cap = n <= 32768 ? 32768 : 65536;
    return (long long)b * cap * 4;
}
extern "C" __attribute__((visibility("default"))) int mgar_fps_batch_buckets(int b, int n, int m, const float *points, float *temp,
                                                                            const int *perm, float *workspace, int *idx, void *stream) {
    MGAR_REQUIRE(b >= 0 && n >= 0 && m >= 0, "fps_batch_buckets: negative size");
    if (b == 0 || m == 0) return MGAR_OK;
    MGAR_REQUIRE(points && temp && idx && perm && workspace, "fps_batch_buckets: null pointer");
    if (n <= 16384 || n > 65536 || (reinterpret_cast<uintptr_t>(workspace) & 15)) {
        set_error("fps_batch_buckets: needs 16384 < n <= 65536 and a 16-byte aligned workspace (smaller clouds: mgar_fps_batch_perm)");
        return MGAR_EUNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    const int cap = n <= 32768 ? 32768 : 65536;
    FpsArgs A{0, n, m, 10, points, temp, nullptr, idx, nullptr};
    KtScope kt(KT_FPS, st, (double)b * (12.0 * n + 4.0 * m), 8.0 * b * ((double)m - 1) * n);
    float4 *sorted = reinterpret_cast<float4 *>(workspace);
    hipLaunchKernelGGL(fps_bucket_prep_kernel, dim3(ceil_div(cap / 4, 256), b), dim3(256), 0, st, points, perm, n, cap, sorted);
    if (cap == 32768) hipLaunchKernelGGL((fps_bucket_kernel<8>), dim3(b), dim3(1024), 0, st, A, sorted, cap);
    else hipLaunchKernelGGL((fps_bucket_kernel<16>), dim3(b), dim3(1024), 0, st, A, sorted, cap);
    return check_launch("fps_batch_buckets: launch failed");
}
