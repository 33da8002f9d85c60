// bf16 prefix-LM attention with the streamed operands of a WHOLE HEAD resident in LDS (K3 / K3b of reference
// models/transformer_decoder.py:138-153 and its autograd, pre-rotated q,k), for the shapes the configs run:
// S <= ~320 rows and head_dim <= 128 (local/real_config: S = 315, d = 100; nocat: S = 238, d = 128).
//
// Why: at these sequence lengths a workgroup of the tiled kernels (attention_fast.hip) lives for 7-10 short steps, each
// behind a barrier and a register-staged tile hand-over, and the counters (profiles/r03_attn_d100_pmc.json) show what that
// costs: a wave lives 41k cycles for 3.2k cycles of MFMA work, waits 58 % of its life and only 1.5 waves per SIMD are
// resident on average.  Here ONE workgroup owns one (batch, head):
//   * its K and V (forward, dQ) or Q and dO (dK/dV) rows are loaded into LDS ONCE (each element of the head is fetched once
//     per kernel: algorithmic traffic), zero-padded to the tile grid; ONE barrier;
//   * after that no wave waits for another: every wave walks its own 16-row groups (stationary rows as register
//     fragments straight from global) over the resident operand, tile by tile, with no barrier and no hand-over;
//   * the 16-row groups are handed out at run time, heaviest first (one LDS atomic per group), to up to 16 waves;
//   * head_dim 100 keeps 112 columns: 7 output column tiles instead of the 8 of a 128-wide padding, and an LDS row pitch
//     of 112 elements (the smallest odd multiple of 16 that holds the columns; 144 for d <= 128): b128 and transposing
//     reads are bank-conflict free, and K + V of S = 315 fit the CU's 160 KiB;
//   * tiles wholly visible to a group run a loop of their own with no mask arithmetic (48 vector instructions for 15 MFMAs);
//   * heads are dealt to the XCDs in contiguous runs (neighbouring heads share 128-byte lines of the projection row);
//   * the dQ kernel computes delta = rowsum(dO * O) for its rows itself (no kernel of its own).
// Measured at B32 H16 d100 S315 (profiles/r03_*): forward 66 -> 42 us, backward 210 -> 152 us; in-kernel stamps
// (benchmarks/attn_res_stamps.py): a workgroup lives 18 us, 5.4 of them in the load phase (nothing else runs on its CU:
// K + V fill the LDS), the rest at about 600 cycles per 16 x 32 score tile per SIMD against 240 of MFMA work.
// Arithmetic, masks and lane maps are those of attention_fast.hip (S^T = K.Q^T keeps the query on the lane, P^T / dS^T go to
// the second product from the accumulator registers; stablemax as max(1/(1+|x|), 1+x)).
#include "attention_frag.h"
#include <stdlib.h>
#include <algorithm>

namespace {
template <int COLS_> struct Geo {
    static constexpr int COLS = COLS_;                     // head columns kept, a multiple of 16 (zero beyond head_dim)
    static constexpr int KS = (COLS + 31) / 32;            // 32-column K-steps of a score; when COLS is not a multiple of 32 the last
                                                           // one runs 16 columns into the NEXT LDS row: finite resident data times
                                                           // the stationary fragment's zeros (the buffer ends in a zeroed slack row)
    static constexpr int PITCH = ((COLS / 16) & 1) ? COLS : COLS + 16;     // elements: 16 x odd
    static constexpr int NDV = COLS / 16;                  // 16-column output tiles
    static constexpr int V4 = PITCH / 4;                   // 8-byte pieces per LDS row
    // waves per workgroup (= register budget 512 / waves-per-SIMD): forward and dQ fit 128 registers; dK/dV keeps 2 fragment
    // sets and 2 accumulator sets
    static constexpr int Q_WAVES = 16;
    static constexpr int KV_WAVES = COLS <= 64 ? 16 : (COLS <= 112 ? 12 : 8);
};
constexpr int RT = 32;                                     // rows per tile step

// Heads are dealt to the XCDs in contiguous runs (workgroup ids congruent mod 8 share an XCD and its L2): a head's rows are
// 2 d bytes at a stride of the whole projection row, so neighbouring heads share 128-byte lines -- they now meet in ONE L2
// at about the same time instead of being fetched by eight.
__device__ __forceinline__ bool head_of_block(int BH, int& bh) {
    const int per = (BH + 7) >> 3;
    bh = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    return (int)(blockIdx.x >> 3) < per && bh < BH;
}
// 16-row groups are handed out at run time, heaviest first (LDS counter, one atomic per group): a wave that finishes a short
// prompt group takes the next one, no wave is bound to a fixed share of an uneven mask
__device__ __forceinline__ int next_group(int* counter, int lane) {
    int t = 0;
    if (lane == 0) t = atomicAdd(counter, 1);
    return __builtin_amdgcn_readfirstlane(t);
}

// rows [0, Spad) x PITCH of TWO head slices into LDS: rows >= S and columns >= d are zero (a masked score still multiplies
// what sits there).  8-byte pieces; every load of both operands is issued before the first LDS store, so the workgroup pays
// the memory latency once (issued in batches of 8 with the stores between them, the load phase was 17 of the forward
// kernel's 53 us: four latencies in a row per workgroup).
template <typename G>
__device__ __forceinline__ void load_heads(bf16* __restrict__ dst_a, const bf16* __restrict__ ga, long lda, bf16* __restrict__ dst_b,
                                           const bf16* __restrict__ gb, long ldb, int S, int d, int Spad) {
    constexpr int U = 12;
    const int total = Spad * G::V4, nt = blockDim.x;
    for (int base = threadIdx.x; base < total; base += nt * U) {
        uint2 va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int idx = base + u * nt;
            const int row = idx / G::V4, c = (idx - row * G::V4) * 4;
            va[u] = make_uint2(0u, 0u); vb[u] = make_uint2(0u, 0u);
            if (idx < total && row < S && c < d) {
                va[u] = *reinterpret_cast<const uint2*>(ga + ((long)row * lda + c));
                vb[u] = *reinterpret_cast<const uint2*>(gb + ((long)row * ldb + c));
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int idx = base + u * nt;
            if (idx < total) {
                const int row = idx / G::V4, c = (idx - row * G::V4) * 4;
                *reinterpret_cast<uint2*>(&dst_a[row * G::PITCH + c]) = va[u];
                *reinterpret_cast<uint2*>(&dst_b[row * G::PITCH + c]) = vb[u];
            }
        }
    }
}
// the stationary fragments of one row (B operands): KS x 8 columns per lane, zero beyond head_dim
template <typename G> struct RowFrag {
    bf16x8_t f[G::KS];
    __device__ __forceinline__ void load(const bf16* __restrict__ g, long ld, int row, int S, int d, int fg) {
#pragma unroll
        for (int ks = 0; ks < G::KS; ks++) f[ks] = frag_global(g, ld, row, ks * 32 + fg * 8, S, d);
    }
};
// acc += T[rows r0 + fr][:] . frag over the head columns (T resident in LDS): the A operand streams from LDS
template <typename G>
__device__ __forceinline__ f32x4_t dot_rows(const bf16* lt, int r0, const RowFrag<G>& b, int fr, int fg) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    const bf16* row = lt + (r0 + fr) * G::PITCH;
#pragma unroll
    for (int ks = 0; ks < G::KS; ks++)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(row + ks * 32 + fg * 8), b.f[ks], acc, 0, 0, 0);
    return acc;
}

// ------------------------------------------------------------------------------------------------ forward
template <typename G, bool STABLE>
__global__ __launch_bounds__(64 * G::Q_WAVES) void attn_fwd_r_k(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int counter;
    const int Spad = (p.S + RT - 1) / RT * RT;
    bf16* lk = reinterpret_cast<bf16*>(smem);
    bf16* lv = lk + Spad * G::PITCH;
    const int lane = threadIdx.x & 63;
    const int fr = lane & 15, fg = lane >> 4, qq = fr >> 2, pp = fr & 3;
    int bh;
    if (!head_of_block(p.B * p.H, bh)) return;
    const int b = bh / p.H, h = bh % p.H;
    if (threadIdx.x == 0) counter = 0;
    if (threadIdx.x < G::PITCH / 4) *reinterpret_cast<uint2*>(&lv[Spad * G::PITCH + 4 * threadIdx.x]) = make_uint2(0u, 0u);   // slack row
    const long head_off = (long)b * p.S * p.ld + (long)h * p.d;
    const bf16* qg = (const bf16*)p.q + head_off;
#ifdef RES_LAB
    long long* stamps = (long long*)p.cos_t + ((long)bh * 16 + (threadIdx.x >> 6)) * 8;
    int nstamp = 0;
#define STAMP() do { if (p.vec_ok == 5 && lane == 0 && nstamp < 8) stamps[nstamp++] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMP() do {} while (0)
#endif
    STAMP();
#ifdef RES_LAB                                              // lab modes: 2 load only, 3 compute only, 4 launch + barrier only, 5 stamps
    if (p.vec_ok != 3 && p.vec_ok != 4)
#endif
    load_heads<G>(lk, (const bf16*)p.k + head_off, p.ld, lv, (const bf16*)p.v + head_off, p.ld, p.S, p.d, Spad);
    __syncthreads();
    STAMP();
#ifdef RES_LAB
    if (p.vec_ok == 2 || p.vec_ok == 4) return;
#endif
    const int ngroups = (p.S + 15) / 16;
    for (;;) {
        const int t = next_group(&counter, lane);
        if (t >= ngroups) break;
        const int g = ngroups - 1 - t;                             // later queries see more keys: heaviest first
        const int qw0 = 16 * g, qi = qw0 + fr;
        RowFrag<G> qf;
        qf.load(qg, p.ld, qi, p.S, p.d, fg);
        const int lim = last_key(qi, p.P, p.S);
        const int kend = last_key(min(qw0 + 15, p.S - 1), p.P, p.S) + 1;
        const int lim0 = last_key(qw0, p.P, p.S);                  // keys up to here are visible to every query of the group
        f32x4_t oacc[G::NDV];
#pragma unroll
        for (int n = 0; n < G::NDV; n++) oacc[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        float m_run = -INFINITY, l_run = 0.f;
        auto tile = [&](int k0, auto fc) {
            constexpr bool FULL = decltype(fc)::value;             // every key of the tile visible to every query of the group
            f32x4_t sacc[2];
            sacc[0] = dot_rows<G>(lk, k0, qf, fr, fg);
            sacc[1] = dot_rows<G>(lk, k0 + 16, qf, fr, fg);
            const int rel = lim - k0 - 4 * fg;                     // key 16t + r of this lane is visible iff 16t + r <= rel
            float pv[8];
            if (STABLE) {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float s = sacc[e / 4][e % 4];
                    const float v = fmaxf(rcpf(fmaf(fabsf(s), p.scale, 1.0f)), fmaf(s, p.scale, 1.0f));
                    pv[e] = (FULL || 16 * (e / 4) + (e % 4) <= rel) ? v : 0.f;
                }
            } else {
                float tmax = -INFINITY;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float x = sacc[e / 4][e % 4] * p.scale;
                    pv[e] = (FULL || 16 * (e / 4) + (e % 4) <= rel) ? x : -INFINITY;
                    tmax = fmaxf(tmax, pv[e]);
                }
                tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
                const float m_new = fmaxf(m_run, tmax);
                const float alpha = (m_new == -INFINITY) ? 1.f : __expf(m_run - m_new);
                l_run *= alpha;
#pragma unroll
                for (int n = 0; n < G::NDV; n++) oacc[n] *= alpha;
#pragma unroll
                for (int e = 0; e < 8; e++) pv[e] = (m_new == -INFINITY) ? 0.f : __expf(pv[e] - m_new);
                m_run = m_new;
            }
#pragma unroll
            for (int e = 0; e < 8; e++) l_run += pv[e];
            const bf16x8_t pf = pack8(pv);
#pragma unroll
            for (int n = 0; n < G::NDV; n++)
                oacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(lv, G::PITCH, k0, 16 * n, fg, qq, pp), pf, oacc[n], 0, 0, 0);
        };
        // tiles wholly visible to the group first (no mask arithmetic), then the ones the mask cuts: two loops, not a branch
        // in one (with both bodies in one loop the register allocator kept both address sets live and spilled)
        const int kfull = min(((lim0 + 1) / RT) * RT, kend);
#pragma unroll 1
        for (int k0 = 0; k0 < kfull; k0 += RT) tile(k0, std::true_type{});
#pragma unroll 1
        for (int k0 = kfull; k0 < kend; k0 += RT) tile(k0, std::false_type{});
        float l = l_run;
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / (STABLE ? fmaxf(l, 1e-9f) : l);
        if (qi < p.S) {
            bf16* og = (bf16*)p.o + ((long)b * p.S + qi) * p.ld_o + (long)h * p.d;
#pragma unroll
            for (int n = 0; n < G::NDV; n++) {
                const int dv = 16 * n + 4 * fg;
                if (dv < p.d) {
                    const float o4[4] = {oacc[n][0] * inv, oacc[n][1] * inv, oacc[n][2] * inv, oacc[n][3] * inv};
                    VecIO<bf16, 4>::store(og + dv, o4);
                }
            }
            if (p.lse && fg == 0) p.lse[((long)b * p.H + h) * p.S + qi] = STABLE ? l : (m_run + logf(l));
        }
        STAMP();
    }
    STAMP();
}

// ------------------------------------------------------------------------------------------------ backward: dQ
// K and V resident; a wave's 16 queries keep q and dO as fragments:  dq^T += K^T . dS^T over the key tiles,
// dS = (dP - delta) / L * s'(x) (stablemax) or p (dP - delta) (softmax), dP = dO.V^T
template <typename G, bool STABLE>
__global__ __launch_bounds__(64 * G::Q_WAVES) void attn_bwd_dq_r_k(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int counter;
    const int Spad = (p.S + RT - 1) / RT * RT;
    bf16* lk = reinterpret_cast<bf16*>(smem);
    bf16* lv = lk + Spad * G::PITCH;
    const int lane = threadIdx.x & 63;
    const int fr = lane & 15, fg = lane >> 4, qq = fr >> 2, pp = fr & 3;
    int bh;
    if (!head_of_block(p.B * p.H, bh)) return;
    const int b = bh / p.H, h = bh % p.H;
    if (threadIdx.x == 0) counter = 0;
    if (threadIdx.x < G::PITCH / 4) *reinterpret_cast<uint2*>(&lv[Spad * G::PITCH + 4 * threadIdx.x]) = make_uint2(0u, 0u);   // slack row
    const long head_off = (long)b * p.S * p.ld + (long)h * p.d;
    const bf16* qg = (const bf16*)p.q + head_off;
    const bf16* dog = (const bf16*)p.dO + (long)b * p.S * p.ld_o + (long)h * p.d;
    load_heads<G>(lk, (const bf16*)p.k + head_off, p.ld, lv, (const bf16*)p.v + head_off, p.ld, p.S, p.d, Spad);
    __syncthreads();
    const int ngroups = (p.S + 15) / 16;
    for (;;) {
        const int t = next_group(&counter, lane);
        if (t >= ngroups) break;
        const int g = ngroups - 1 - t;                             // later queries see more keys: heaviest first
        const int qw0 = 16 * g, qi = qw0 + fr;
        RowFrag<G> qf, gf;
        qf.load(qg, p.ld, qi, p.S, p.d, fg);
        gf.load(dog, p.ld_o, qi, p.S, p.d, fg);
        const int lim = last_key(qi, p.P, p.S);
        const int kend = last_key(min(qw0 + 15, p.S - 1), p.P, p.S) + 1;
        const int lim0 = last_key(qw0, p.P, p.S);
        float c0, c1;                                      // stablemax: 1/L and -delta/L;  softmax: lse and delta
        {
            // delta = rowsum(dO * O) of the group's rows, here instead of in a kernel of its own (17 us at the config shape):
            // the lane's 8 columns of every 32 against the same columns of O, summed over the four lanes of a row; written
            // out for the dK/dV kernel, which runs after this one
            float dlt = 0.f;
            const bf16* og = (const bf16*)p.o + (long)b * p.S * p.ld_o + (long)h * p.d;
#pragma unroll
            for (int ks = 0; ks < G::KS; ks++) {
                const bf16x8_t of = frag_global(og, p.ld_o, qi, ks * 32 + fg * 8, p.S, p.d);
#pragma unroll
                for (int e = 0; e < 8; e++) dlt = fmaf(bf2f((unsigned short)of[e]), bf2f((unsigned short)gf.f[ks][e]), dlt);
            }
            dlt += __shfl_xor(dlt, 16);
            dlt += __shfl_xor(dlt, 32);
            const long stat = ((long)b * p.H + h) * p.S + min(qi, p.S - 1);
            if (fg == 0 && qi < p.S) const_cast<float*>(p.delta)[stat] = dlt;
            const float lse = p.lse[stat];
            if (STABLE) { c0 = 1.0f / fmaxf(lse, 1e-9f); c1 = -dlt * c0; }
            else { c0 = lse; c1 = dlt; }
        }
        f32x4_t dqa[G::NDV];
#pragma unroll
        for (int n = 0; n < G::NDV; n++) dqa[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        auto tile = [&](int k0, auto fc) {
            constexpr bool FULL = decltype(fc)::value;
            f32x4_t sa[2], da[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                sa[t] = dot_rows<G>(lk, k0 + 16 * t, qf, fr, fg);
                da[t] = dot_rows<G>(lv, k0 + 16 * t, gf, fr, fg);
            }
            const int rel = lim - k0 - 4 * fg;
            float e8[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int t = e / 4, rr = e % 4;
                const float s = sa[t][rr];
                float ds;
                if (STABLE) {                              // s'(x) = x < 0 ? 1/(1-x)^2 : 1 = rcp(1 + max(-x, 0))^2
                    const float w = rcpf(fmaf(fmaxf(-s, 0.f), p.scale, 1.0f));
                    ds = fmaf(da[t][rr], c0, c1) * w * w;
                } else {
                    ds = __expf(fmaf(s, p.scale, -c0)) * (da[t][rr] - c1);
                }
                e8[e] = (FULL || 16 * t + rr <= rel) ? ds : 0.f;
            }
            const bf16x8_t dsf = pack8(e8);
#pragma unroll
            for (int n = 0; n < G::NDV; n++)
                dqa[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(lk, G::PITCH, k0, 16 * n, fg, qq, pp), dsf, dqa[n], 0, 0, 0);
        };
        // tiles wholly visible to the group first (no mask arithmetic), then the ones the mask cuts: two loops, not a branch
        // in one (with both bodies in one loop the register allocator kept both address sets live and spilled)
        const int kfull = min(((lim0 + 1) / RT) * RT, kend);
#pragma unroll 1
        for (int k0 = 0; k0 < kfull; k0 += RT) tile(k0, std::true_type{});
#pragma unroll 1
        for (int k0 = kfull; k0 < kend; k0 += RT) tile(k0, std::false_type{});
        if (qi < p.S) {
            bf16* dq = (bf16*)p.dq + ((long)b * p.S + qi) * p.ld_d + (long)h * p.d;
#pragma unroll
            for (int n = 0; n < G::NDV; n++) {
                const int dv = 16 * n + 4 * fg;
                if (dv < p.d) {
                    const float o4[4] = {dqa[n][0] * p.scale, dqa[n][1] * p.scale, dqa[n][2] * p.scale, dqa[n][3] * p.scale};
                    VecIO<bf16, 4>::store(dq + dv, o4);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// Q and dO resident, with two statistics per query row (stablemax {1/L, -delta/L}, softmax {lse, delta}; rows >= S carry
// {0, 0} / {+inf, 0}: their p and dS are exactly zero).  A wave's 16 keys keep k and v as fragments:
//   dv^T += dO^T . P,  dk^T += Q^T . dS  over the query tiles that see the keys.
template <typename G, bool STABLE>
__global__ __launch_bounds__(64 * G::KV_WAVES) void attn_bwd_dkv_r_k(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Spad = (p.S + RT - 1) / RT * RT;
    bf16* lq = reinterpret_cast<bf16*>(smem);
    bf16* lg = lq + Spad * G::PITCH;
    float* lstat = reinterpret_cast<float*>(lg + (Spad + 1) * G::PITCH);       // [2][Spad], behind the slack row
    __shared__ int counter;
    const int lane = threadIdx.x & 63;
    const int fr = lane & 15, fg = lane >> 4, qq = fr >> 2, pp = fr & 3;
    int bh;
    if (!head_of_block(p.B * p.H, bh)) return;
    const int b = bh / p.H, h = bh % p.H;
    if (threadIdx.x == 0) counter = 0;
    if (threadIdx.x < G::PITCH / 4) *reinterpret_cast<uint2*>(&lg[Spad * G::PITCH + 4 * threadIdx.x]) = make_uint2(0u, 0u);   // slack row
    const long head_off = (long)b * p.S * p.ld + (long)h * p.d;
    const bf16* kg = (const bf16*)p.k + head_off;
    const bf16* vg = (const bf16*)p.v + head_off;
    const long statbase = ((long)b * p.H + h) * p.S;
    load_heads<G>(lq, (const bf16*)p.q + head_off, p.ld, lg, (const bf16*)p.dO + (long)b * p.S * p.ld_o + (long)h * p.d, p.ld_o, p.S, p.d, Spad);
    for (int j = threadIdx.x; j < Spad; j += blockDim.x) {
        const float lse = p.lse[statbase + min(j, p.S - 1)], dlt = p.delta[statbase + min(j, p.S - 1)];
        float s0, s1;
        if (STABLE) { s0 = j < p.S ? 1.0f / fmaxf(lse, 1e-9f) : 0.f; s1 = -dlt * s0; }
        else { s0 = j < p.S ? lse : INFINITY; s1 = j < p.S ? dlt : 0.f; }
        lstat[j] = s0; lstat[Spad + j] = s1;
    }
    __syncthreads();
    const int ngroups = (p.S + 15) / 16;
    for (;;) {
        const int g = next_group(&counter, lane);                  // earlier keys are seen by more queries: heaviest first
        if (g >= ngroups) break;
        const int kw0 = 16 * g, kj = kw0 + fr;
        RowFrag<G> kf, vf;
        kf.load(kg, p.ld, kj, p.S, p.d, fg);
        vf.load(vg, p.ld, kj, p.S, p.d, fg);
        const int qmin = first_query(kj, p.P);
        const int qstart = (first_query(kw0, p.P) / RT) * RT;                  // first query tile that sees any key of the group
        const int qfull = first_query(min(kw0 + 15, p.S - 1), p.P);            // queries from here on see every key of the group
        f32x4_t dka[G::NDV], dva[G::NDV];
#pragma unroll
        for (int n = 0; n < G::NDV; n++) { dka[n] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dva[n] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
        auto tile = [&](int q0, auto fc) {
            constexpr bool FULL = decltype(fc)::value;     // every query of the tile sees every key of the group
            f32x4_t sa[2], da[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                sa[t] = dot_rows<G>(lq, q0 + 16 * t, kf, fr, fg);
                da[t] = dot_rows<G>(lg, q0 + 16 * t, vf, fr, fg);
            }
            const int rel = qmin - q0 - 4 * fg;            // query 16t + r of this lane sees the key iff 16t + r >= rel
            float p8[8], d8[8];
#pragma unroll
            for (int tt = 0; tt < 2; tt++) {
                const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(&lstat[q0 + 16 * tt + 4 * fg]);
                const f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(&lstat[Spad + q0 + 16 * tt + 4 * fg]);
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const float s = sa[tt][rr];
                    float pr, ds;
                    if (STABLE) {
                        const float sx = fmaxf(rcpf(fmaf(fabsf(s), p.scale, 1.0f)), fmaf(s, p.scale, 1.0f));
                        const float w = fminf(sx, 1.0f);
                        pr = sx * s0[rr];
                        ds = fmaf(da[tt][rr], s0[rr], s1[rr]) * w * w;
                    } else {
                        pr = __expf(fmaf(s, p.scale, -s0[rr]));
                        ds = pr * (da[tt][rr] - s1[rr]);
                    }
                    const bool ok = FULL || 16 * tt + rr >= rel;
                    p8[4 * tt + rr] = ok ? pr : 0.f; d8[4 * tt + rr] = ok ? ds : 0.f;
                }
            }
            const bf16x8_t pf = pack8(p8), dsf = pack8(d8);
#pragma unroll
            for (int n = 0; n < G::NDV; n++) {
                dva[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(lg, G::PITCH, q0, 16 * n, fg, qq, pp), pf, dva[n], 0, 0, 0);
                dka[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(lq, G::PITCH, q0, 16 * n, fg, qq, pp), dsf, dka[n], 0, 0, 0);
            }
        };
        // query tiles the mask cuts first, then the ones that see every key of the group: two loops, not a branch in one
        const int qcut = min((qfull + RT - 1) / RT * RT, Spad);
#pragma unroll 1
        for (int q0 = qstart; q0 < qcut; q0 += RT) tile(q0, std::false_type{});
#pragma unroll 1
        for (int q0 = max(qcut, qstart); q0 < p.S; q0 += RT) tile(q0, std::true_type{});
        if (kj < p.S) {
            bf16* dk = (bf16*)p.dk + ((long)b * p.S + kj) * p.ld_d + (long)h * p.d;
            bf16* dv = (bf16*)p.dv + ((long)b * p.S + kj) * p.ld_d + (long)h * p.d;
#pragma unroll
            for (int n = 0; n < G::NDV; n++) {
                const int c = 16 * n + 4 * fg;
                if (c < p.d) {
                    const float k4[4] = {dka[n][0] * p.scale, dka[n][1] * p.scale, dka[n][2] * p.scale, dka[n][3] * p.scale};
                    const float v4[4] = {dva[n][0], dva[n][1], dva[n][2], dva[n][3]};
                    VecIO<bf16, 4>::store(dk + c, k4);
                    VecIO<bf16, 4>::store(dv + c, v4);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ host
constexpr size_t LDS_LIMIT = 160 * 1024;
int pitch_of(int d) {
    if (d <= 32) return Geo<32>::PITCH;
    if (d <= 64) return Geo<64>::PITCH;
    if (d <= 96) return Geo<96>::PITCH;
    if (d <= 112) return Geo<112>::PITCH;
    return Geo<128>::PITCH;
}
size_t lds_need(const AttnP& p, bool stats) {
    const size_t Spad = (size_t)cdiv(p.S, RT) * RT;
    return (2 * Spad + 1) * pitch_of(p.d) * sizeof(bf16) + (stats ? 2 * Spad * sizeof(float) : 0);      // + the zeroed slack row
}
int waves_for(int S, int maxw) {
    static const int env = getenv("MM_ATTN_RES_WAVES") ? atoi(getenv("MM_ATTN_RES_WAVES")) : 0;       // A/B knob
    if (env > 0) maxw = std::min(maxw, env);
    return (int)std::min<long>(maxw, cdiv(S, 16));
}
unsigned grid_for(const AttnP& p) { return (unsigned)(8 * cdiv((long)p.B * p.H, 8)); }
template <typename K> int raise_lds(K kern, size_t bytes) {
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) MM_FAIL(MM_ERR_LAUNCH, "attention (resident): cannot raise dynamic LDS to %zu bytes: %s", bytes, hipGetErrorString(e));
    }
    return MM_OK;
}
template <typename G> int fwd_go(const AttnP& p_, int mode, hipStream_t st) {
    AttnP p = p_;
#ifdef RES_LAB                                              // benchmarks/attn_res_stamps.py; the library build has none of this
    static const int dbg = getenv("MM_ATTN_RES_DBG") ? atoi(getenv("MM_ATTN_RES_DBG")) : 1;
    p.vec_ok = dbg;
#endif
    const size_t lds = lds_need(p, false);
    const dim3 grid(grid_for(p)), block((unsigned)(64 * waves_for(p.S, G::Q_WAVES)));
    if (mode == MM_ATTN_STABLEMAX) {
        if (int rc = raise_lds(attn_fwd_r_k<G, true>, lds)) return rc;
        hipLaunchKernelGGL((attn_fwd_r_k<G, true>), grid, block, lds, st, p);
    } else {
        if (int rc = raise_lds(attn_fwd_r_k<G, false>, lds)) return rc;
        hipLaunchKernelGGL((attn_fwd_r_k<G, false>), grid, block, lds, st, p);
    }
    MM_LAUNCH_CHECK();
    return MM_OK;
}
template <typename G> int bwd_go(const AttnP& p, int mode, hipStream_t st) {
    const size_t lds_q = lds_need(p, false), lds_kv = lds_need(p, true);
    const dim3 grid(grid_for(p)), block((unsigned)(64 * waves_for(p.S, G::Q_WAVES))), block_kv((unsigned)(64 * waves_for(p.S, G::KV_WAVES)));
    if (mode == MM_ATTN_STABLEMAX) {
        if (int rc = raise_lds(attn_bwd_dq_r_k<G, true>, lds_q)) return rc;
        hipLaunchKernelGGL((attn_bwd_dq_r_k<G, true>), grid, block, lds_q, st, p);
        if (int rc = raise_lds(attn_bwd_dkv_r_k<G, true>, lds_kv)) return rc;
        hipLaunchKernelGGL((attn_bwd_dkv_r_k<G, true>), grid, block_kv, lds_kv, st, p);
    } else {
        if (int rc = raise_lds(attn_bwd_dq_r_k<G, false>, lds_q)) return rc;
        hipLaunchKernelGGL((attn_bwd_dq_r_k<G, false>), grid, block, lds_q, st, p);
        if (int rc = raise_lds(attn_bwd_dkv_r_k<G, false>, lds_kv)) return rc;
        hipLaunchKernelGGL((attn_bwd_dkv_r_k<G, false>), grid, block_kv, lds_kv, st, p);
    }
    MM_LAUNCH_CHECK();
    return MM_OK;
}
#define RES_PICK_(GO_)                                                     \
    do {                                                                   \
        if (p.d <= 32) return GO_<Geo<32>>(p, mode, st);                   \
        if (p.d <= 64) return GO_<Geo<64>>(p, mode, st);                   \
        if (p.d <= 96) return GO_<Geo<96>>(p, mode, st);                   \
        if (p.d <= 112) return GO_<Geo<112>>(p, mode, st);                 \
        return GO_<Geo<128>>(p, mode, st);                                 \
    } while (0)
}  // namespace

// whole-head-resident forms: pre-rotated bf16, 8-byte pieces, head_dim <= 128, both resident operands (+ statistics) within the
// CU's LDS, and enough heads to fill the chip one workgroup per CU at a time
bool attn_res_ok(const AttnP& p) {
    return p.d % 4 == 0 && p.d >= 4 && p.d <= 128 && p.ld % 4 == 0 && p.ld_o % 4 == 0 && p.S >= 64 && lds_need(p, true) <= LDS_LIMIT &&
           (long)p.B * p.H >= 128;
}
int attn_res_fwd(const AttnP& p, int mode, hipStream_t st) { RES_PICK_(fwd_go); }
// computes delta = rowsum(dO * O) itself (in the dQ kernel): the caller does NOT launch attn_delta_k first
int attn_res_bwd(const AttnP& p, int mode, hipStream_t st) { RES_PICK_(bwd_go); }
