// bf16 fast forms of K3 / K3b (prefix-LM attention, reference models/transformer_decoder.py:138-153 and its autograd)
// for pre-rotated q,k (mm_rope_inplace ran first), head_dim <= 192, 8-byte aligned rows.
//
// Same transposed-product scheme as attention.hip (S^T = K.Q^T keeps the query on the lane; P^T is the B operand of the
// second product straight from the accumulator registers), restructured around what the counters of the previous
// generation showed (profiles/r02_attn_pmc.json: 13 vector instructions per MFMA, every LDS fragment feeding ONE MFMA,
// two barriers per 32-key tile):
//   * a wave owns NG groups of 16 rows (NG = 2: 32 queries / keys per wave, 128 per workgroup), so every fragment read
//     from LDS feeds NG MFMAs: LDS bytes per flop halve;
//   * streamed tiles are double-buffered in LDS and TWO further tiles are in flight in registers: the tile for step i+1 is
//     written at the END of step i (its global loads were issued two steps earlier), ONE barrier separates the steps
//     (the waves spent 62 % of their cycles waiting with one tile in flight);
//   * no load sits behind a branch: rows past the end are clamped to the last row (their scores are masked, or their
//     statistics are zero), columns past d are clamped and zeroed by a select;
//   * stablemax needs no compare: s(x) = max(1/(1+|x|), 1+x)  [x<0: (1+x)(1-x) <= 1;  x>=0: 1/(1+x) <= 1 <= 1+x], with
//     the 1/sqrt(d) scale folded into the two fmas; the reference's "+1e-9" (transformer_decoder.py:54) is below half an
//     ulp of 1-x >= 1 in f32 and changes nothing.  s'(x) = min(s(x),1)^2.  5-6 vector instructions per score.
//   * the mask is one compare against a per-lane limit (last visible key of a query / first query that sees a key);
//     tiles wholly visible to the wave skip it.
#include "attention_frag.h"
#include <stdlib.h>

namespace {
// ------------------------------------------------------------------------------------------------ forward
template <int DP, int KT, int NG, bool STABLE>
__global__ __launch_bounds__(256) void attn_fwd3_k(const AttnP p) {
    typedef Tile<DP, KT> TL;
    constexpr int PITCH = TL::PITCH, NKS = DP / 32, NDV = DP / 16, NT = KT / 16, NKK = KT / 32, TILE = KT * PITCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* lds = reinterpret_cast<bf16*>(smem);                     // [2 buffers][K tile | V tile]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: conditions on the wave's rows become scalar branches
    const int fr = lane & 15, fg = lane >> 4, qq = fr >> 2, pp = fr & 3;
    int rb, bh;
    if (!map_block((p.S + 64 * NG - 1) / (64 * NG), p.B * p.H, rb, bh)) return;
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = rb * (64 * NG), qw0 = q0 + wave * (16 * NG);
    const long head_off = (long)b * p.S * p.ld + (long)h * p.d;
    const bf16* qg = (const bf16*)p.q + head_off;
    const bf16* kg = (const bf16*)p.k + head_off;
    const bf16* vg = (const bf16*)p.v + head_off;
    const int ld = (int)p.ld;
    const int kend_blk = last_key(min(q0 + 64 * NG - 1, p.S - 1), p.P, p.S) + 1;
    const int kend_w = qw0 < p.S ? last_key(min(qw0 + 16 * NG - 1, p.S - 1), p.P, p.S) + 1 : 0;
    const int lim_w0 = qw0 < p.S ? last_key(qw0, p.P, p.S) : -1;   // keys up to here are visible to every query of the wave
    const int ntiles = (kend_blk + KT - 1) / KT;

    TL rk, rv, rk2, rv2;            // two streamed tiles in flight in registers (tile i+1 and tile i+2 during step i)
    rk.load(kg, ld, 0, p.S, p.d); rv.load(vg, ld, 0, p.S, p.d);
    bf16x8_t qf[NG][NKS];
    int lim[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const int qi = qw0 + 16 * g + fr;
        lim[g] = last_key(qi, p.P, p.S);
#pragma unroll
        for (int ks = 0; ks < NKS; ks++) qf[g][ks] = frag_global(qg, p.ld, qi, ks * 32 + fg * 8, p.S, p.d);
    }
    f32x4_t oacc[NG][NDV];
    float m_run[NG], l_run[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) {
        m_run[g] = -INFINITY; l_run[g] = 0.f;
#pragma unroll
        for (int n = 0; n < NDV; n++) oacc[g][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    rk.store(lds); rv.store(lds + TILE);
    if (ntiles > 1) { rk.load(kg, ld, KT, p.S, p.d); rv.load(vg, ld, KT, p.S, p.d); }
    if (ntiles > 2) { rk2.load(kg, ld, 2 * KT, p.S, p.d); rv2.load(vg, ld, 2 * KT, p.S, p.d); }
    __syncthreads();
    // end of step i: the register set holding tile i+1 goes to the LDS buffer step i-1 read (every wave is past that
    // barrier) and is refilled with tile i+3, so each global load has two whole steps to land
    auto advance = [&](int i, TL& ak, TL& av) {
        if (i + 1 < ntiles) {
            bf16* nb = lds + ((i + 1) & 1) * 2 * TILE;
            ak.store(nb); av.store(nb + TILE);
            if (i + 3 < ntiles) { ak.load(kg, ld, (i + 3) * KT, p.S, p.d); av.load(vg, ld, (i + 3) * KT, p.S, p.d); }
        }
        __syncthreads();
    };

    auto step = [&](int i) {
        const int k0 = i * KT;
        const bf16* lk = lds + (i & 1) * 2 * TILE;
        const bf16* lv = lk + TILE;
        if (k0 < kend_w) {
            f32x4_t sacc[NG][NT];
#pragma unroll
            for (int t = 0; t < NT; t++) {
#pragma unroll
                for (int g = 0; g < NG; g++) sacc[g][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NKS; ks++) {
                    const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(&lk[(16 * t + fr) * PITCH + ks * 32 + fg * 8]);
#pragma unroll
                    for (int g = 0; g < NG; g++) sacc[g][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[g][ks], sacc[g][t], 0, 0, 0);
                }
            }
            const bool full = k0 + KT - 1 <= lim_w0;           // wave-uniform
            bf16x8_t pf[NG][NKK];
#pragma unroll
            for (int g = 0; g < NG; g++) {
                const int rel = lim[g] - k0 - 4 * fg;          // key 16t + r of this lane is visible iff 16t + r <= rel
                float pv[4 * NT];
                if (STABLE) {
                    auto fill = [&](auto fc) {
                        constexpr bool FULL = decltype(fc)::value;
#pragma unroll
                        for (int t = 0; t < NT; t++)
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const float s = sacc[g][t][r];
                                const float v = fmaxf(rcpf(fmaf(fabsf(s), p.scale, 1.0f)), fmaf(s, p.scale, 1.0f));
                                pv[4 * t + r] = (FULL || 16 * t + r <= rel) ? v : 0.f;
                            }
                    };
                    if (full) fill(std::true_type{}); else fill(std::false_type{});
                } else {
                    float tmax = -INFINITY;
#pragma unroll
                    for (int t = 0; t < NT; t++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const float x = sacc[g][t][r] * p.scale;
                            pv[4 * t + r] = (full || 16 * t + r <= rel) ? x : -INFINITY;
                            tmax = fmaxf(tmax, pv[4 * t + r]);
                        }
                    tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
                    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
                    const float m_new = fmaxf(m_run[g], tmax);
                    const float alpha = (m_new == -INFINITY) ? 1.f : __expf(m_run[g] - m_new);
                    l_run[g] *= alpha;
#pragma unroll
                    for (int n = 0; n < NDV; n++) oacc[g][n] *= alpha;
#pragma unroll
                    for (int e = 0; e < 4 * NT; e++) pv[e] = (m_new == -INFINITY) ? 0.f : __expf(pv[e] - m_new);
                    m_run[g] = m_new;
                }
#pragma unroll
                for (int kk = 0; kk < NKK; kk++) {
                    float e8[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) { e8[e] = pv[8 * kk + e]; l_run[g] += e8[e]; }
                    pf[g][kk] = pack8(e8);
                }
            }
#pragma unroll
            for (int kk = 0; kk < NKK; kk++) {
                if (k0 + 32 * kk < kend_w) {                   // wave-uniform: later keys are masked for every query here
#pragma unroll
                    for (int n = 0; n < NDV; n++) {
                        const bf16x8_t vf = frag_tr(lv, PITCH, 32 * kk, 16 * n, fg, qq, pp);
#pragma unroll
                        for (int g = 0; g < NG; g++) oacc[g][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[g][kk], oacc[g][n], 0, 0, 0);
                    }
                }
            }
        }
    };
    for (int i = 0; i < ntiles; i += 2) {
        step(i); advance(i, rk, rv);
        if (i + 1 < ntiles) { step(i + 1); advance(i + 1, rk2, rv2); }
    }
#pragma unroll
    for (int g = 0; g < NG; g++) {
        float l = l_run[g];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / (STABLE ? fmaxf(l, 1e-9f) : l);
        const int qi = qw0 + 16 * g + fr;
        if (qi < p.S) {
            bf16* og = (bf16*)p.o + ((long)b * p.S + qi) * p.ld_o + (long)h * p.d;
#pragma unroll
            for (int n = 0; n < NDV; n++) {
                const int dv = 16 * n + 4 * fg;
                if (dv < p.d) {
                    const float o4[4] = {oacc[g][n][0] * inv, oacc[g][n][1] * inv, oacc[g][n][2] * inv, oacc[g][n][3] * inv};
                    VecIO<bf16, 4>::store(og + dv, o4);
                }
            }
            if (p.lse && fg == 0) p.lse[((long)b * p.H + h) * p.S + qi] = STABLE ? l : (m_run[g] + logf(l));
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward: dQ
// dq^T += K^T . dS^T over key tiles;  dS = (dP - delta) / L * s'(x) (stablemax) or p (dP - delta) (softmax), dP = dO.V^T
template <int DP, int KT, int NG, bool STABLE>
__global__ __launch_bounds__(256) void attn_bwd_dq3_k(const AttnP p) {
    typedef Tile<DP, KT> TL;
    constexpr int PITCH = TL::PITCH, NKS = DP / 32, NDV = DP / 16, NT = KT / 16, NKK = KT / 32, TILE = KT * PITCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* lds = reinterpret_cast<bf16*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: conditions on the wave's rows become scalar branches
    const int fr = lane & 15, fg = lane >> 4, qq = fr >> 2, pp = fr & 3;
    int rb, bh;
    if (!map_block((p.S + 64 * NG - 1) / (64 * NG), p.B * p.H, rb, bh)) return;
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = rb * (64 * NG), qw0 = q0 + wave * (16 * NG);
    const long head_off = (long)b * p.S * p.ld + (long)h * p.d;
    const bf16* qg = (const bf16*)p.q + head_off;
    const bf16* kg = (const bf16*)p.k + head_off;
    const bf16* vg = (const bf16*)p.v + head_off;
    const bf16* dog = (const bf16*)p.dO + (long)b * p.S * p.ld_o + (long)h * p.d;
    const int ld = (int)p.ld;
    const int kend_blk = last_key(min(q0 + 64 * NG - 1, p.S - 1), p.P, p.S) + 1;
    const int kend_w = qw0 < p.S ? last_key(min(qw0 + 16 * NG - 1, p.S - 1), p.P, p.S) + 1 : 0;
    const int lim_w0 = qw0 < p.S ? last_key(qw0, p.P, p.S) : -1;
    const int ntiles = (kend_blk + KT - 1) / KT;

    TL rk, rv, rk2, rv2;            // two streamed tiles in flight in registers (tile i+1 and tile i+2 during step i)
    rk.load(kg, ld, 0, p.S, p.d); rv.load(vg, ld, 0, p.S, p.d);
    bf16x8_t qf[NG][NKS], gf[NG][NKS];
    int lim[NG];
    float c0[NG], c1[NG];       // stablemax: 1/L and -delta/L;  softmax: lse and delta
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const int qi = qw0 + 16 * g + fr;
        lim[g] = last_key(qi, p.P, p.S);
#pragma unroll
        for (int ks = 0; ks < NKS; ks++) {
            qf[g][ks] = frag_global(qg, p.ld, qi, ks * 32 + fg * 8, p.S, p.d);
            gf[g][ks] = frag_global(dog, p.ld_o, qi, ks * 32 + fg * 8, p.S, p.d);
        }
        const long stat = ((long)b * p.H + h) * p.S + min(qi, p.S - 1);
        const float lse = p.lse[stat], dlt = p.delta[stat];
        if (STABLE) { c0[g] = 1.0f / fmaxf(lse, 1e-9f); c1[g] = -dlt * c0[g]; }
        else { c0[g] = lse; c1[g] = dlt; }
    }
    f32x4_t dqa[NG][NDV];
#pragma unroll
    for (int g = 0; g < NG; g++)
#pragma unroll
        for (int n = 0; n < NDV; n++) dqa[g][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    rk.store(lds); rv.store(lds + TILE);
    if (ntiles > 1) { rk.load(kg, ld, KT, p.S, p.d); rv.load(vg, ld, KT, p.S, p.d); }
    if (ntiles > 2) { rk2.load(kg, ld, 2 * KT, p.S, p.d); rv2.load(vg, ld, 2 * KT, p.S, p.d); }
    __syncthreads();
    // end of step i: the register set holding tile i+1 goes to the LDS buffer step i-1 read (every wave is past that
    // barrier) and is refilled with tile i+3, so each global load has two whole steps to land
    auto advance = [&](int i, TL& ak, TL& av) {
        if (i + 1 < ntiles) {
            bf16* nb = lds + ((i + 1) & 1) * 2 * TILE;
            ak.store(nb); av.store(nb + TILE);
            if (i + 3 < ntiles) { ak.load(kg, ld, (i + 3) * KT, p.S, p.d); av.load(vg, ld, (i + 3) * KT, p.S, p.d); }
        }
        __syncthreads();
    };

    auto step = [&](int i) {
        const int k0 = i * KT;
        const bf16* lk = lds + (i & 1) * 2 * TILE;
        const bf16* lv = lk + TILE;
        if (k0 < kend_w) {
            f32x4_t sa[NG][NT], da[NG][NT];
#pragma unroll
            for (int t = 0; t < NT; t++) {
#pragma unroll
                for (int g = 0; g < NG; g++) { sa[g][t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; da[g][t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int ks = 0; ks < NKS; ks++) {
                    const int off = (16 * t + fr) * PITCH + ks * 32 + fg * 8;
                    const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(&lk[off]);
                    const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(&lv[off]);
#pragma unroll
                    for (int g = 0; g < NG; g++) {
                        sa[g][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[g][ks], sa[g][t], 0, 0, 0);
                        da[g][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, gf[g][ks], da[g][t], 0, 0, 0);
                    }
                }
            }
            const bool full = k0 + KT - 1 <= lim_w0;
            bf16x8_t dsf[NG][NKK];
#pragma unroll
            for (int g = 0; g < NG; g++) {
                const int rel = lim[g] - k0 - 4 * fg;
                auto fill = [&](auto fc) {
                    constexpr bool FULL = decltype(fc)::value;
#pragma unroll
                    for (int kk = 0; kk < NKK; kk++) {
                        float e8[8];
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            const int t = 2 * kk + e / 4, r = e % 4;
                            const float s = sa[g][t][r];
                            float ds;
                            if (STABLE) {      // s'(x) = x < 0 ? 1/(1-x)^2 : 1 = rcp(1 + max(-x, 0))^2
                                const float w = rcpf(fmaf(fmaxf(-s, 0.f), p.scale, 1.0f));
                                ds = fmaf(da[g][t][r], c0[g], c1[g]) * w * w;
                            } else {
                                ds = __expf(fmaf(s, p.scale, -c0[g])) * (da[g][t][r] - c1[g]);
                            }
                            e8[e] = (FULL || 16 * t + r <= rel) ? ds : 0.f;
                        }
                        dsf[g][kk] = pack8(e8);
                    }
                };
                if (full) fill(std::true_type{}); else fill(std::false_type{});
            }
#pragma unroll
            for (int kk = 0; kk < NKK; kk++) {
                if (k0 + 32 * kk < kend_w) {
#pragma unroll
                    for (int n = 0; n < NDV; n++) {
                        const bf16x8_t kt = frag_tr(lk, PITCH, 32 * kk, 16 * n, fg, qq, pp);
#pragma unroll
                        for (int g = 0; g < NG; g++) dqa[g][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, dsf[g][kk], dqa[g][n], 0, 0, 0);
                    }
                }
            }
        }
    };
    for (int i = 0; i < ntiles; i += 2) {
        step(i); advance(i, rk, rv);
        if (i + 1 < ntiles) { step(i + 1); advance(i + 1, rk2, rv2); }
    }
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const int qi = qw0 + 16 * g + fr;
        if (qi < p.S) {
            bf16* dq = (bf16*)p.dq + ((long)b * p.S + qi) * p.ld_d + (long)h * p.d;
#pragma unroll
            for (int n = 0; n < NDV; n++) {
                const int dv = 16 * n + 4 * fg;
                if (dv < p.d) {
                    const float o4[4] = {dqa[g][n][0] * p.scale, dqa[g][n][1] * p.scale, dqa[g][n][2] * p.scale, dqa[g][n][3] * p.scale};
                    VecIO<bf16, 4>::store(dq + dv, o4);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// a wave owns NG x 16 keys (K, V fragments stationary in registers) and streams query tiles (Q, dO):
//   dv^T += dO^T . P,  dk^T += Q^T . dS.   Row statistics of the tile's queries sit in LDS next to it: stablemax {1/L, -delta/L},
//   softmax {lse, delta}; queries >= S carry {0, 0} / {+inf, 0}, which makes their p and dS exactly zero (no row mask).
template <int DP, int QT, int NG, bool STABLE>
__global__ __launch_bounds__(256) void attn_bwd_dkv3_k(const AttnP p) {
    typedef Tile<DP, QT> TL;
    constexpr int PITCH = TL::PITCH, NKS = DP / 32, NDV = DP / 16, NT = QT / 16, NKK = QT / 32, TILE = QT * PITCH;
    constexpr int BUF = 2 * TILE + 4 * QT;                                    // elements (bf16): Q tile, dO tile, 2 x QT floats
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* lds = reinterpret_cast<bf16*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: conditions on the wave's rows become scalar branches
    const int fr = lane & 15, fg = lane >> 4, qq = fr >> 2, pp = fr & 3;
    int rb, bh;
    if (!map_block((p.S + 64 * NG - 1) / (64 * NG), p.B * p.H, rb, bh)) return;
    const int b = bh / p.H, h = bh % p.H;
    const int kb0 = rb * (64 * NG), kw0 = kb0 + wave * (16 * NG);
    const long head_off = (long)b * p.S * p.ld + (long)h * p.d;
    const bf16* qg = (const bf16*)p.q + head_off;
    const bf16* kg = (const bf16*)p.k + head_off;
    const bf16* vg = (const bf16*)p.v + head_off;
    const bf16* dog = (const bf16*)p.dO + (long)b * p.S * p.ld_o + (long)h * p.d;
    const int ld = (int)p.ld, ldo = (int)p.ld_o;
    const long statbase = ((long)b * p.H + h) * p.S;
    const int qstart_blk = (first_query(kb0, p.P) / QT) * QT;
    const int qstart_w = first_query(kw0, p.P);                               // first query that sees any key of this wave
    const int qfull_w = first_query(min(kw0 + 16 * NG - 1, p.S - 1), p.P);     // queries from here on see every key of the wave
    const bool wave_live = kw0 < p.S;
    const int ntiles = (p.S - qstart_blk + QT - 1) / QT;

    struct Set { TL q, g; float s0, s1; };          // one streamed tile in registers: Q rows, dO rows, two statistics per row
    Set ra, rb2;                                     // two in flight (tile i+1 and tile i+2 during step i)
    auto load_set = [&](Set& r, int qbase) {
        r.q.load(qg, ld, qbase, p.S, p.d); r.g.load(dog, ldo, qbase, p.S, p.d);
        r.s0 = 0.f; r.s1 = 0.f;
        if (tid < QT) {
            const int qi = qbase + tid;
            const float lse = p.lse[statbase + min(qi, p.S - 1)], dlt = p.delta[statbase + min(qi, p.S - 1)];
            if (STABLE) { r.s0 = qi < p.S ? 1.0f / fmaxf(lse, 1e-9f) : 0.f; r.s1 = -dlt * r.s0; }
            else { r.s0 = qi < p.S ? lse : INFINITY; r.s1 = qi < p.S ? dlt : 0.f; }
        }
    };
    auto store_set = [&](const Set& r, bf16* nb) {
        r.q.store(nb); r.g.store(nb + TILE);
        float* ls = reinterpret_cast<float*>(nb + 2 * TILE);
        if (tid < QT) { ls[tid] = r.s0; ls[QT + tid] = r.s1; }
    };
    load_set(ra, qstart_blk);
    bf16x8_t kf[NG][NKS], vf[NG][NKS];
    int qmin[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const int kj = kw0 + 16 * g + fr;
        qmin[g] = first_query(kj, p.P);
#pragma unroll
        for (int ks = 0; ks < NKS; ks++) {
            kf[g][ks] = frag_global(kg, p.ld, kj, ks * 32 + fg * 8, p.S, p.d);
            vf[g][ks] = frag_global(vg, p.ld, kj, ks * 32 + fg * 8, p.S, p.d);
        }
    }
    f32x4_t dka[NG][NDV], dva[NG][NDV];
#pragma unroll
    for (int g = 0; g < NG; g++)
#pragma unroll
        for (int n = 0; n < NDV; n++) { dka[g][n] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dva[g][n] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    store_set(ra, lds);
    if (ntiles > 1) load_set(ra, qstart_blk + QT);
    if (ntiles > 2) load_set(rb2, qstart_blk + 2 * QT);
    __syncthreads();
    auto advance = [&](int i, Set& r) {
        if (i + 1 < ntiles) {
            store_set(r, lds + ((i + 1) & 1) * BUF);
            if (i + 3 < ntiles) load_set(r, qstart_blk + (i + 3) * QT);
        }
        __syncthreads();
    };

    auto step = [&](int i) {
        const int q0 = qstart_blk + i * QT;
        const bf16* lq = lds + (i & 1) * BUF;
        const bf16* lg = lq + TILE;
        const float* lstat = reinterpret_cast<const float*>(lq + 2 * TILE);
        if (wave_live && q0 + QT > qstart_w) {
            f32x4_t sa[NG][NT], da[NG][NT];
#pragma unroll
            for (int t = 0; t < NT; t++) {
#pragma unroll
                for (int g = 0; g < NG; g++) { sa[g][t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; da[g][t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int ks = 0; ks < NKS; ks++) {
                    const int off = (16 * t + fr) * PITCH + ks * 32 + fg * 8;
                    const bf16x8_t qf = *reinterpret_cast<const bf16x8_t*>(&lq[off]);
                    const bf16x8_t gf = *reinterpret_cast<const bf16x8_t*>(&lg[off]);
#pragma unroll
                    for (int g = 0; g < NG; g++) {
                        sa[g][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf[g][ks], sa[g][t], 0, 0, 0);
                        da[g][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, vf[g][ks], da[g][t], 0, 0, 0);
                    }
                }
            }
            const bool full = q0 >= qfull_w;                   // wave-uniform
            bf16x8_t pf[NG][NKK], dsf[NG][NKK];
#pragma unroll
            for (int kk = 0; kk < NKK; kk++) {
                f32x4_t s0[2], s1[2];                          // statistics of queries 16t + 4fg .. +3, t = 2kk, 2kk+1
#pragma unroll
                for (int tt = 0; tt < 2; tt++) {
                    s0[tt] = *reinterpret_cast<const f32x4_t*>(&lstat[16 * (2 * kk + tt) + 4 * fg]);
                    s1[tt] = *reinterpret_cast<const f32x4_t*>(&lstat[QT + 16 * (2 * kk + tt) + 4 * fg]);
                }
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    const int rel = qmin[g] - q0 - 4 * fg;     // query 16t + r of this lane sees the key iff 16t + r >= rel
                    auto fill = [&](auto fc) {
                        constexpr bool FULL = decltype(fc)::value;
                        float p8[8], d8[8];
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            const int tt = e / 4, t = 2 * kk + tt, r = e % 4;
                            const float s = sa[g][t][r];
                            float pr, ds;
                            if (STABLE) {
                                const float sx = fmaxf(rcpf(fmaf(fabsf(s), p.scale, 1.0f)), fmaf(s, p.scale, 1.0f));
                                const float w = fminf(sx, 1.0f);
                                pr = sx * s0[tt][r];
                                ds = fmaf(da[g][t][r], s0[tt][r], s1[tt][r]) * w * w;
                            } else {
                                pr = __expf(fmaf(s, p.scale, -s0[tt][r]));
                                ds = pr * (da[g][t][r] - s1[tt][r]);
                            }
                            const bool ok = FULL || 16 * t + r >= rel;
                            p8[e] = ok ? pr : 0.f; d8[e] = ok ? ds : 0.f;
                        }
                        pf[g][kk] = pack8(p8); dsf[g][kk] = pack8(d8);
                    };
                    if (full) fill(std::true_type{}); else fill(std::false_type{});
                }
            }
#pragma unroll
            for (int kk = 0; kk < NKK; kk++) {
                if (q0 + 32 * kk + 32 > qstart_w) {            // wave-uniform: earlier queries see none of this wave's keys
#pragma unroll
                    for (int n = 0; n < NDV; n++) {
                        const bf16x8_t gt = frag_tr(lg, PITCH, 32 * kk, 16 * n, fg, qq, pp);
                        const bf16x8_t qt = frag_tr(lq, PITCH, 32 * kk, 16 * n, fg, qq, pp);
#pragma unroll
                        for (int g = 0; g < NG; g++) {
                            dva[g][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt, pf[g][kk], dva[g][n], 0, 0, 0);
                            dka[g][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, dsf[g][kk], dka[g][n], 0, 0, 0);
                        }
                    }
                }
            }
        }
    };
    for (int i = 0; i < ntiles; i += 2) {
        step(i); advance(i, ra);
        if (i + 1 < ntiles) { step(i + 1); advance(i + 1, rb2); }
    }
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const int kj = kw0 + 16 * g + fr;
        if (kj < p.S) {
            bf16* dk = (bf16*)p.dk + ((long)b * p.S + kj) * p.ld_d + (long)h * p.d;
            bf16* dv = (bf16*)p.dv + ((long)b * p.S + kj) * p.ld_d + (long)h * p.d;
#pragma unroll
            for (int n = 0; n < NDV; n++) {
                const int c = 16 * n + 4 * fg;
                if (c < p.d) {
                    const float k4[4] = {dka[g][n][0] * p.scale, dka[g][n][1] * p.scale, dka[g][n][2] * p.scale, dka[g][n][3] * p.scale};
                    const float v4[4] = {dva[g][n][0], dva[g][n][1], dva[g][n][2], dva[g][n][3]};
                    VecIO<bf16, 4>::store(dk + c, k4);
                    VecIO<bf16, 4>::store(dv + c, v4);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ host
template <typename K> int set_lds(K kern, size_t bytes) {
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) MM_FAIL(MM_ERR_LAUNCH, "attention: cannot raise dynamic LDS to %zu bytes: %s", bytes, hipGetErrorString(e));
    }
    return MM_OK;
}

int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }

template <int DP, int KT, int NG>
int fwd_launch(const AttnP& p, int mode, hipStream_t st) {
    const size_t lds = (size_t)2 * 2 * KT * (DP + FP) * 2;
    const dim3 grid(grid_blocks((int)cdiv(p.S, 64 * NG), p.B * p.H));
    if (mode == MM_ATTN_STABLEMAX) {
        if (int rc = set_lds(attn_fwd3_k<DP, KT, NG, true>, lds)) return rc;
        hipLaunchKernelGGL((attn_fwd3_k<DP, KT, NG, true>), grid, dim3(256), lds, st, p);
    } else {
        if (int rc = set_lds(attn_fwd3_k<DP, KT, NG, false>, lds)) return rc;
        hipLaunchKernelGGL((attn_fwd3_k<DP, KT, NG, false>), grid, dim3(256), lds, st, p);
    }
    MM_LAUNCH_CHECK();
    return MM_OK;
}

template <int DP, int KT, int NG>
int dq_launch(const AttnP& p, int mode, hipStream_t st) {
    const size_t lds = (size_t)2 * 2 * KT * (DP + FP) * 2;
    const dim3 grid(grid_blocks((int)cdiv(p.S, 64 * NG), p.B * p.H));
    if (mode == MM_ATTN_STABLEMAX) {
        if (int rc = set_lds(attn_bwd_dq3_k<DP, KT, NG, true>, lds)) return rc;
        hipLaunchKernelGGL((attn_bwd_dq3_k<DP, KT, NG, true>), grid, dim3(256), lds, st, p);
    } else {
        if (int rc = set_lds(attn_bwd_dq3_k<DP, KT, NG, false>, lds)) return rc;
        hipLaunchKernelGGL((attn_bwd_dq3_k<DP, KT, NG, false>), grid, dim3(256), lds, st, p);
    }
    MM_LAUNCH_CHECK();
    return MM_OK;
}

template <int DP, int QT, int NG>
int dkv_launch(const AttnP& p, int mode, hipStream_t st) {
    const size_t lds = (size_t)2 * (2 * QT * (DP + FP) * 2 + 2 * QT * sizeof(float));
    const dim3 grid(grid_blocks((int)cdiv(p.S, 64 * NG), p.B * p.H));
    if (mode == MM_ATTN_STABLEMAX) {
        if (int rc = set_lds(attn_bwd_dkv3_k<DP, QT, NG, true>, lds)) return rc;
        hipLaunchKernelGGL((attn_bwd_dkv3_k<DP, QT, NG, true>), grid, dim3(256), lds, st, p);
    } else {
        if (int rc = set_lds(attn_bwd_dkv3_k<DP, QT, NG, false>, lds)) return rc;
        hipLaunchKernelGGL((attn_bwd_dkv3_k<DP, QT, NG, false>), grid, dim3(256), lds, st, p);
    }
    MM_LAUNCH_CHECK();
    return MM_OK;
}
}  // namespace

bool attn_fast_ok(const AttnP& p) {
    // 32-bit element offsets inside one batch element's rows; whole 8-byte pieces
    return p.d % 4 == 0 && p.d >= 4 && p.d <= 192 && p.ld % 4 == 0 && p.ld_o % 4 == 0 && (long)p.S * p.ld < (1L << 30);
}

// Tile / group choice per head-dim class (only the combinations that fit the register file are instantiated):
//   d <= 64 : 64-row streamed tiles, NG = 2;   d <= 128: 32-row tiles, NG = 1 or 2;   d <= 192: 32-row tiles, NG = 1.
// MM_ATTN_NG / MM_ATTN_NG_DQ / MM_ATTN_NG_DKV (1|2) and MM_ATTN_KT (32|64, d <= 64 only) override the defaults for A/B runs.
#define ATTN_PICK_(LAUNCH_)                                                                               \
    do {                                                                                                  \
        if (p.d <= 32) {                                                                                  \
            if (ng == 1) return kt == 32 ? LAUNCH_<32, 32, 1>(p, mode, st) : LAUNCH_<32, 64, 1>(p, mode, st); \
            return kt == 32 ? LAUNCH_<32, 32, 2>(p, mode, st) : LAUNCH_<32, 64, 2>(p, mode, st);          \
        }                                                                                                 \
        if (p.d <= 64) {                                                                                  \
            if (ng == 1) return kt == 32 ? LAUNCH_<64, 32, 1>(p, mode, st) : LAUNCH_<64, 64, 1>(p, mode, st); \
            return kt == 32 ? LAUNCH_<64, 32, 2>(p, mode, st) : LAUNCH_<64, 64, 2>(p, mode, st);          \
        }                                                                                                 \
        if (p.d <= 128) return ng == 1 ? LAUNCH_<128, 32, 1>(p, mode, st) : LAUNCH_<128, 32, 2>(p, mode, st); \
        return LAUNCH_<192, 32, 1>(p, mode, st);                                                          \
    } while (0)

int attn_fast_fwd(const AttnP& p, int mode, hipStream_t st) {
    static const int ng_env = env_int("MM_ATTN_NG", 0), kt_env = env_int("MM_ATTN_KT", 0);
    const int ng = ng_env ? ng_env : (p.S > 64 ? 2 : 1);
    const int kt = kt_env ? kt_env : 64;
    ATTN_PICK_(fwd_launch);
}

int attn_fast_bwd(const AttnP& p, int mode, hipStream_t st) {
    static const int ng_env = env_int("MM_ATTN_NG_DQ", 0), kt_env = env_int("MM_ATTN_KT", 0), ngkv_env = env_int("MM_ATTN_NG_DKV", 0);
    const int kt = kt_env ? kt_env : 64;
    {
        const int ng = ng_env ? ng_env : (p.S > 64 ? 2 : 1);
        auto go = [&]() -> int { ATTN_PICK_(dq_launch); };
        const int rc = go();
        if (rc != MM_OK) return rc;
    }
    const int ng = ngkv_env ? ngkv_env : (p.S > 64 && p.d <= 64 ? 2 : 1);
    ATTN_PICK_(dkv_launch);
}
