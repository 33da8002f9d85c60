// bf16 prefix-LM attention for head_dim = 512 (configs/real_config_mup.yaml: E2048 H4), pre-rotated q,k: K3 / K3b of
// reference models/transformer_decoder.py:138-153 and its autograd, same transposed-product scheme and lane maps as
// attention_fast.hip (S^T = K.Q^T keeps the query on the lane; P^T / dS^T go to the second product straight from the
// accumulator registers).
//
// What is different at d = 512: the stationary state of 16 rows no longer fits a quarter of the register file --
//   dQ kernel   : q and dO fragments 2 x 64 registers, dq^T accumulators 128
//   dK/dV kernel: k and v fragments 2 x 64, dk^T and dv^T accumulators 2 x 128
//   forward     : q fragments 64, o^T accumulators 128
// so a workgroup is 4 waves, ONE per SIMD, each with the whole 512-entry register file (accumulators in the AGPR half),
// 16 stationary rows per wave, 64 per workgroup.  With one wave per SIMD nothing hides a register-staged tile load, and four
// 32-row tiles in flight would cost 128 more registers, so the streamed tiles (K,V or Q,dO: 32 rows x 1 KiB each) go
// global -> LDS by DMA (`global_load_lds_dwordx4`: one wave-instruction = one 1 KiB row, scalar row base + lane offset, no
// VGPR staging), double-buffered: the tile for step i+1 is requested at the top of step i and waited for (vmcnt(0) + one
// barrier) at the top of step i+1.  LDS: 2 buffers x 2 operands x 32 rows x 1056 B = 132 KiB (+ row statistics).
// The kernels this replaces at d = 512 (attention.hip: register-staged, single-buffered, two barriers per tile, epilogue
// through LDS) ran 1191 us per layer backward at B32 H4 S315; DESIGN.md 4 (K3 wide) has the measured figures of these.
#include "attention_frag.h"
#include <stdlib.h>

namespace {
constexpr int WD = 512;                        // head_dim served (rows of exactly 1 KiB: one DMA instruction per row)
constexpr int WP = WD + FP;                    // LDS row pitch in elements (1056 B: b128 and tr reads conflict-free)
constexpr int WT = 32;                         // streamed rows per tile
constexpr int WTILE = WT * WP;                 // elements per operand tile
constexpr int WNKS = WD / 32, WNDV = WD / 16;
constexpr int WSTAT_MAX = 3072;                // row statistics of one head kept in LDS by the dK/dV kernel (2 floats per row)
constexpr size_t WLDS_TILES = (size_t)2 * 2 * WTILE * sizeof(bf16);

typedef __attribute__((address_space(3))) void* lds_vp;

// one streamed row (512 bf16) global -> LDS: lane l's 16 bytes land at lds_addr + 16 l.  `row` is wave-uniform.
__device__ __forceinline__ void dma_row(const bf16* row, unsigned lds_addr, unsigned lane16) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(lane16), "s"(row) : "memory", "m0");
}
__device__ __forceinline__ const bf16* uniform_ptr(const bf16* q) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)q), hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)q >> 32));
    return (const bf16*)(((uintptr_t)hi << 32) | lo);
}
// tile rows row0 .. row0+31 of two operands (a: K or Q, b: V or dO) into buffer `buf`; rows past S-1 are clamped (their
// scores are masked / their statistics are zero).  64 rows over 4 waves: 16 DMA instructions per wave.
__device__ __forceinline__ void stage_pair(const bf16* a, long lda, const bf16* b, long ldb, int row0, int S, unsigned lds0, int buf,
                                           int wave, unsigned lane16) {
    const unsigned base = lds0 + (unsigned)(buf * 2 * WTILE * 2);
#pragma unroll
    for (int i = 0; i < WT / 4; i++) {
        const int r = wave + 4 * i;
        const int gr = min(row0 + r, S - 1);
        dma_row(a + (long)gr * lda, base + (unsigned)(r * WP * 2), lane16);
        dma_row(b + (long)gr * ldb, base + (unsigned)((WTILE + r * WP) * 2), lane16);
    }
}
__device__ __forceinline__ void wait_tile() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}


// ---- MFMA with the register file of every operand chosen by hand ---------------------------------------------------------
// Left to the compiler, these kernels either copied all 128 loop-carried accumulators AGPR -> VGPR -> AGPR around every tile
// step (AGPR form: 714 v_accvgpr moves in the dQ kernel), or (VGPR form) filled the 256 VGPRs with stationary state and read
// every LDS fragment just before its use, two at a time, with the LDS latency exposed 48 times a step.  Here the stationary
// fragments (B operands of the scores phase, loop-invariant) live in AGPRs by constraint, which leaves VGPRs for the streamed
// fragments: they are read a chunk ahead (double-buffered in registers) behind sched_barriers.  The compiler does not know these
// instructions, so the waits its hazard recognizer would add after an MFMA are written out (s_nop) where a VALU instruction
// or an accumulator read follows one.
__device__ __forceinline__ void mfma_v0(f32x4_t& acc, const bf16x8_t a, const bf16x8_t& b_acc) {      // acc(VGPR) = a . b
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b_acc));
}
__device__ __forceinline__ void mfma_v(f32x4_t& acc, const bf16x8_t a, const bf16x8_t& b_acc) {       // acc(VGPR) += a . b
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b_acc));
}
// The OUTPUT accumulators are loop-carried: a value that flows through a loop phi is given the VGPR class whatever asm
// constraints its uses carry (an "+a" accumulator was copied VGPR -> AGPR -> VGPR around every chunk: 256 moves per step), so
// they stay in VGPRs through the compiler's own VGPR-form MFMA (Makefile: -amdgpu-mfma-vgpr-form=1), 128 of the 256.
__device__ __forceinline__ void mfma_out(f32x4_t& acc, const bf16x8_t a, const bf16x8_t b) {          // acc(VGPR, loop-carried) += a . b
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
// a stationary fragment becomes ONE aligned 4-register AGPR tuple here, once (assembled from two 8-byte loads it otherwise sat
// in scattered registers and every use rebuilt the tuple with four v_accvgpr_mov)
__device__ __forceinline__ void pin_acc(bf16x8_t& f) { asm volatile("" : "+a"(f)); }
// MFMA result -> VALU read: 24 wait states, written out because the hazard recognizer does not see an MFMA inside asm.  The
// accumulators pass THROUGH the statement, so no instruction that reads them can be scheduled above it.
__device__ __forceinline__ void mfma_drain(f32x4_t (&sa)[2], f32x4_t (&da)[2]) {
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(sa[0]), "+v"(sa[1]), "+v"(da[0]), "+v"(da[1]));
}

// Scores phase of one 32-row streamed tile: sa[t] = A_tile[16t..][:] . Af (and da[t] = B_tile . Bf when TWO), the tile's
// fragments read two 32-column slices (8 or 4 ds_read_b128) ahead of the MFMAs that use them.  `piece(c)` is called once per
// slice pair, c = 0..7: the caller issues the next tile's DMA rows there, spread over the phase instead of in one burst.
template <bool TWO, typename PIECE>
__device__ __forceinline__ void scores_phase(const bf16* la, const bf16* lb, const bf16x8_t (&Af)[WNKS], const bf16x8_t (&Bf)[TWO ? WNKS : 1],
                                             f32x4_t (&sa)[2], f32x4_t (&da)[2], int fr, int fg, PIECE&& piece) {
    bf16x8_t fa[2][2][2], fb[2][2][2];                 // [register buffer][slice of the pair][16-row group]
    auto rd = [&](int c, int buf) {
#pragma unroll
        for (int kk = 0; kk < 2; kk++)
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int off = (16 * t + fr) * WP + (2 * c + kk) * 32 + fg * 8;
                fa[buf][kk][t] = *reinterpret_cast<const bf16x8_t*>(&la[off]);
                if (TWO) fb[buf][kk][t] = *reinterpret_cast<const bf16x8_t*>(&lb[off]);
            }
    };
    rd(0, 0);
#pragma unroll
    for (int c = 0; c < WNKS / 2; c++) {
        if (c + 1 < WNKS / 2) rd(c + 1, (c + 1) & 1);
        piece(c);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; kk++)
#pragma unroll
            for (int t = 0; t < 2; t++) {
                if (c == 0 && kk == 0) {
                    mfma_v0(sa[t], fa[c & 1][kk][t], Af[2 * c + kk]);
                    if (TWO) mfma_v0(da[t], fb[c & 1][kk][t], Bf[TWO ? 2 * c + kk : 0]);
                } else {
                    mfma_v(sa[t], fa[c & 1][kk][t], Af[2 * c + kk]);
                    if (TWO) mfma_v(da[t], fb[c & 1][kk][t], Bf[TWO ? 2 * c + kk : 0]);
                }
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}
// Output phase: acc[n] += src_tile^T[16n.., 32 rows] . bfrag for the 32 column groups of the head, transposed fragments read
// eight groups ahead.  `first8` holds the first eight, which the caller read before its vector work (tr_chunk).
__device__ __forceinline__ void tr_chunk(const bf16* lsrc, int j, bf16x8_t (&ft)[8], int fg, int qq, int pp) {
#pragma unroll
    for (int n = 0; n < 8; n++) ft[n] = frag_tr(lsrc, WP, 0, 16 * (8 * j + n), fg, qq, pp);
}
__device__ __forceinline__ void output_phase(const bf16* lsrc, bf16x8_t (&ft0)[8], const bf16x8_t bfrag, f32x4_t (&acc)[WNDV], int fg, int qq, int pp) {
    bf16x8_t ft1[8];
#pragma unroll
    for (int j = 0; j < WNDV / 8; j++) {
        bf16x8_t (&cur)[8] = (j & 1) ? ft1 : ft0;
        bf16x8_t (&nxt)[8] = (j & 1) ? ft0 : ft1;
        if (j + 1 < WNDV / 8) tr_chunk(lsrc, j + 1, nxt, fg, qq, pp);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < 8; n++) mfma_out(acc[8 * j + n], cur[n], bfrag);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// one DMA row pair of the next tile (rows r = wave + 4c of both operands), c = 0..7
__device__ __forceinline__ void stage_piece(const bf16* a, long lda, const bf16* b, long ldb, int row0, int S, unsigned lds0, int buf,
                                            int wave, unsigned lane16, int c) {
    const unsigned base = lds0 + (unsigned)(buf * 2 * WTILE * 2);
    const int r = wave + 4 * c;
    const int gr = min(row0 + r, S - 1);
    dma_row(a + (long)gr * lda, base + (unsigned)(r * WP * 2), lane16);
    dma_row(b + (long)gr * ldb, base + (unsigned)((WTILE + r * WP) * 2), lane16);
}

// ------------------------------------------------------------------------------------------------ forward
template <bool STABLE>
__global__ __launch_bounds__(256) void attn_fwd_w_k(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* lds = reinterpret_cast<bf16*>(smem);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_vp)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane16 = (unsigned)lane * 16u;
    const int fr = lane & 15, fg = lane >> 4, qq = fr >> 2, pp = fr & 3;
    int rb, bh;
    if (!map_block((p.S + 63) / 64, p.B * p.H, rb, bh)) return;
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = rb * 64, qw0 = q0 + wave * 16;
    const long head_off = (long)b * p.S * p.ld + (long)h * p.d;
    const bf16* qg = uniform_ptr((const bf16*)p.q + head_off);       // scalar bases: the DMA row addresses are scalar arithmetic
    const bf16* kg = uniform_ptr((const bf16*)p.k + head_off);
    const bf16* vg = uniform_ptr((const bf16*)p.v + head_off);
    const int kend_blk = last_key(min(q0 + 63, p.S - 1), p.P, p.S) + 1;
    const int kend_w = qw0 < p.S ? last_key(min(qw0 + 15, p.S - 1), p.P, p.S) + 1 : 0;
    const int lim_w0 = qw0 < p.S ? last_key(qw0, p.P, p.S) : -1;
    const int ntiles = (kend_blk + WT - 1) / WT;

    stage_pair(kg, p.ld, vg, p.ld, 0, p.S, lds0, 0, wave, lane16);
    const int qi = qw0 + fr;
    const int lim = last_key(qi, p.P, p.S);
    bf16x8_t qf[WNKS], none[1];
#pragma unroll
    for (int ks = 0; ks < WNKS; ks++) { qf[ks] = frag_global(qg, p.ld, qi, ks * 32 + fg * 8, p.S, p.d); pin_acc(qf[ks]); }
    f32x4_t oacc[WNDV];
#pragma unroll
    for (int n = 0; n < WNDV; n++) oacc[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    for (int i = 0; i < ntiles; i++) {
        wait_tile();                                   // tile i landed; every wave is done with the other buffer
        const bool more = i + 1 < ntiles;
        const int k0 = i * WT;
        const bf16* lk = lds + (i & 1) * 2 * WTILE;
        const bf16* lv = lk + WTILE;
        auto piece = [&](int c) { if (more) stage_piece(kg, p.ld, vg, p.ld, (i + 1) * WT, p.S, lds0, (i + 1) & 1, wave, lane16, c); };
        if (k0 < kend_w) {
            f32x4_t sacc[2], unused[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
            scores_phase<false>(lk, lv, qf, none, sacc, unused, fr, fg, piece);
            bf16x8_t ft0[8];
            tr_chunk(lv, 0, ft0, fg, qq, pp);          // in flight under the drain and the vector work
            mfma_drain(sacc, unused);
            const bool full = k0 + WT - 1 <= lim_w0;
            const int rel = lim - k0 - 4 * fg;
            float pv[8];
            if (STABLE) {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float s = sacc[e / 4][e % 4];
                    const float v = fmaxf(rcpf(fmaf(fabsf(s), p.scale, 1.0f)), fmaf(s, p.scale, 1.0f));
                    pv[e] = (full || 16 * (e / 4) + (e % 4) <= rel) ? v : 0.f;
                }
            } else {
                float tmax = -INFINITY;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float x = sacc[e / 4][e % 4] * p.scale;
                    pv[e] = (full || 16 * (e / 4) + (e % 4) <= rel) ? x : -INFINITY;
                    tmax = fmaxf(tmax, pv[e]);
                }
                tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
                const float m_new = fmaxf(m_run, tmax);
                const float alpha = (m_new == -INFINITY) ? 1.f : __expf(m_run - m_new);
                l_run *= alpha;
#pragma unroll
                for (int n = 0; n < WNDV; n++) oacc[n] *= alpha;
#pragma unroll
                for (int e = 0; e < 8; e++) pv[e] = (m_new == -INFINITY) ? 0.f : __expf(pv[e] - m_new);
                m_run = m_new;
            }
#pragma unroll
            for (int e = 0; e < 8; e++) l_run += pv[e];
            const bf16x8_t pf = pack8(pv);
            output_phase(lv, ft0, pf, oacc, fg, qq, pp);
        } else {
#pragma unroll
            for (int c = 0; c < WT / 4; c++) piece(c);
        }
    }
    float l = l_run;
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / (STABLE ? fmaxf(l, 1e-9f) : l);
    if (qi < p.S) {
        bf16* og = (bf16*)p.o + ((long)b * p.S + qi) * p.ld_o + (long)h * p.d;
#pragma unroll
        for (int n = 0; n < WNDV; n++) {
            const float o4[4] = {oacc[n][0] * inv, oacc[n][1] * inv, oacc[n][2] * inv, oacc[n][3] * inv};
            VecIO<bf16, 4>::store(og + 16 * n + 4 * fg, o4);
        }
        if (p.lse && fg == 0) p.lse[((long)b * p.H + h) * p.S + qi] = STABLE ? l : (m_run + logf(l));
    }
}

// ------------------------------------------------------------------------------------------------ backward: dQ
// dq^T += K^T . dS^T over key tiles;  dS = (dP - delta) / L * s'(x) (stablemax) or p (dP - delta) (softmax), dP = dO.V^T
template <bool STABLE>
__global__ __launch_bounds__(256) void attn_bwd_dq_w_k(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* lds = reinterpret_cast<bf16*>(smem);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_vp)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane16 = (unsigned)lane * 16u;
    const int fr = lane & 15, fg = lane >> 4, qq = fr >> 2, pp = fr & 3;
    int rb, bh;
    if (!map_block((p.S + 63) / 64, p.B * p.H, rb, bh)) return;
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = rb * 64, qw0 = q0 + wave * 16;
    const long head_off = (long)b * p.S * p.ld + (long)h * p.d;
    const bf16* qg = uniform_ptr((const bf16*)p.q + head_off);       // scalar bases: the DMA row addresses are scalar arithmetic
    const bf16* kg = uniform_ptr((const bf16*)p.k + head_off);
    const bf16* vg = uniform_ptr((const bf16*)p.v + head_off);
    const bf16* dog = uniform_ptr((const bf16*)p.dO + (long)b * p.S * p.ld_o + (long)h * p.d);
    const int kend_blk = last_key(min(q0 + 63, p.S - 1), p.P, p.S) + 1;
    const int kend_w = qw0 < p.S ? last_key(min(qw0 + 15, p.S - 1), p.P, p.S) + 1 : 0;
    const int lim_w0 = qw0 < p.S ? last_key(qw0, p.P, p.S) : -1;
    const int ntiles = (kend_blk + WT - 1) / WT;

    stage_pair(kg, p.ld, vg, p.ld, 0, p.S, lds0, 0, wave, lane16);
    const int qi = qw0 + fr;
    const int lim = last_key(qi, p.P, p.S);
    bf16x8_t qf[WNKS], gf[WNKS];
#pragma unroll
    for (int ks = 0; ks < WNKS; ks++) {
        qf[ks] = frag_global(qg, p.ld, qi, ks * 32 + fg * 8, p.S, p.d);
        gf[ks] = frag_global(dog, p.ld_o, qi, ks * 32 + fg * 8, p.S, p.d);
        pin_acc(qf[ks]); pin_acc(gf[ks]);
    }
    float c0, c1;                                      // stablemax: 1/L and -delta/L;  softmax: lse and delta
    {
        const long stat = ((long)b * p.H + h) * p.S + min(qi, p.S - 1);
        const float lse = p.lse[stat], dlt = p.delta[stat];
        if (STABLE) { c0 = 1.0f / fmaxf(lse, 1e-9f); c1 = -dlt * c0; }
        else { c0 = lse; c1 = dlt; }
    }
    f32x4_t dqa[WNDV];
#pragma unroll
    for (int n = 0; n < WNDV; n++) dqa[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int i = 0; i < ntiles; i++) {
        wait_tile();
        const bool more = i + 1 < ntiles;
        const int k0 = i * WT;
        const bf16* lk = lds + (i & 1) * 2 * WTILE;
        const bf16* lv = lk + WTILE;
        auto piece = [&](int c) { if (more) stage_piece(kg, p.ld, vg, p.ld, (i + 1) * WT, p.S, lds0, (i + 1) & 1, wave, lane16, c); };
        if (k0 < kend_w) {
            f32x4_t sa[2], da[2];
            scores_phase<true>(lk, lv, qf, gf, sa, da, fr, fg, piece);
            bf16x8_t ft0[8];
            tr_chunk(lk, 0, ft0, fg, qq, pp);          // in flight under the drain and the vector work
            mfma_drain(sa, da);
            const bool full = k0 + WT - 1 <= lim_w0;
            const int rel = lim - k0 - 4 * fg;
            float e8[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int t = e / 4, r = e % 4;
                const float s = sa[t][r];
                float ds;
                if (STABLE) {                          // s'(x) = x < 0 ? 1/(1-x)^2 : 1 = rcp(1 + max(-x, 0))^2
                    const float w = rcpf(fmaf(fmaxf(-s, 0.f), p.scale, 1.0f));
                    ds = fmaf(da[t][r], c0, c1) * w * w;
                } else {
                    ds = __expf(fmaf(s, p.scale, -c0)) * (da[t][r] - c1);
                }
                e8[e] = (full || 16 * t + r <= rel) ? ds : 0.f;
            }
            const bf16x8_t dsf = pack8(e8);
            output_phase(lk, ft0, dsf, dqa, fg, qq, pp);
        } else {
#pragma unroll
            for (int c = 0; c < WT / 4; c++) piece(c);
        }
    }
    if (qi < p.S) {
        bf16* dq = (bf16*)p.dq + ((long)b * p.S + qi) * p.ld_d + (long)h * p.d;
#pragma unroll
        for (int n = 0; n < WNDV; n++) {
            const float o4[4] = {dqa[n][0] * p.scale, dqa[n][1] * p.scale, dqa[n][2] * p.scale, dqa[n][3] * p.scale};
            VecIO<bf16, 4>::store(dq + 16 * n + 4 * fg, o4);
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// a wave owns 16 keys (fragments stationary in registers) and streams query tiles (Q, dO):
//   dv^T += dO^T . P,  dk^T += Q^T . dS.
// k + v fragments (128 registers) and BOTH accumulator sets (256) leave too little for the step's own fragments (the fused
// form spilled 145 registers), so the two gradients are two ROLES of one launch (blockIdx.y): the dV role keeps k and dv^T
// (S, then P^T.dO: 64 MFMAs per tile), the dK role keeps k, v and dk^T (S and dP, then dS^T.Q: 96 MFMAs per tile).
// The row statistics of every query the workgroup will see sit in LDS from the start: stablemax {1/L, -delta/L}, softmax
// {lse, delta}; queries >= S carry {0, 0} / {+inf, 0}, which makes their p and dS exactly zero (the clamped rows the DMA
// delivered for them contribute nothing).
template <bool STABLE, bool ROLE_DK>
__device__ __forceinline__ void attn_bwd_dkv_w_body(const AttnP& p, char* smem) {
    bf16* lds = reinterpret_cast<bf16*>(smem);
    float* lstat = reinterpret_cast<float*>(smem + WLDS_TILES);                // [2][npad]
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_vp)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lane16 = (unsigned)lane * 16u;
    const int fr = lane & 15, fg = lane >> 4, qq = fr >> 2, pp = fr & 3;
    int rb, bh;
    if (!map_block((p.S + 63) / 64, p.B * p.H, rb, bh)) return;
    const int b = bh / p.H, h = bh % p.H;
    const int kb0 = rb * 64, kw0 = kb0 + wave * 16;
    const long head_off = (long)b * p.S * p.ld + (long)h * p.d;
    const bf16* qg = uniform_ptr((const bf16*)p.q + head_off);       // scalar bases: the DMA row addresses are scalar arithmetic
    const bf16* kg = uniform_ptr((const bf16*)p.k + head_off);
    const bf16* vg = uniform_ptr((const bf16*)p.v + head_off);
    const bf16* dog = uniform_ptr((const bf16*)p.dO + (long)b * p.S * p.ld_o + (long)h * p.d);
    const long statbase = ((long)b * p.H + h) * p.S;
    const int qstart_blk = (first_query(kb0, p.P) / WT) * WT;
    const int qstart_w = first_query(kw0, p.P);                               // first query that sees any key of this wave
    const int qfull_w = first_query(min(kw0 + 15, p.S - 1), p.P);              // queries from here on see every key of the wave
    const bool wave_live = kw0 < p.S;
    const int ntiles = (p.S - qstart_blk + WT - 1) / WT;
    const int npad = ntiles * WT;

    stage_pair(qg, p.ld, dog, p.ld_o, qstart_blk, p.S, lds0, 0, wave, lane16);
    for (int j = tid; j < npad; j += 256) {
        const int qi = qstart_blk + j;
        const float lse = p.lse[statbase + min(qi, p.S - 1)], dlt = p.delta[statbase + min(qi, p.S - 1)];
        float s0, s1;
        if (STABLE) { s0 = qi < p.S ? 1.0f / fmaxf(lse, 1e-9f) : 0.f; s1 = -dlt * s0; }
        else { s0 = qi < p.S ? lse : INFINITY; s1 = qi < p.S ? dlt : 0.f; }
        lstat[j] = s0; lstat[npad + j] = s1;
    }
    const int kj = kw0 + fr;
    const int qmin = first_query(kj, p.P);
    bf16x8_t kf[WNKS], vf[ROLE_DK ? WNKS : 1];
#pragma unroll
    for (int ks = 0; ks < WNKS; ks++) {
        kf[ks] = frag_global(kg, p.ld, kj, ks * 32 + fg * 8, p.S, p.d);
        pin_acc(kf[ks]);
        if (ROLE_DK) { vf[ks] = frag_global(vg, p.ld, kj, ks * 32 + fg * 8, p.S, p.d); pin_acc(vf[ks]); }
    }
    f32x4_t acc[WNDV];
#pragma unroll
    for (int n = 0; n < WNDV; n++) acc[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int i = 0; i < ntiles; i++) {
        wait_tile();                                   // (the first one also publishes lstat)
        const bool more = i + 1 < ntiles;
        const int q0 = qstart_blk + i * WT;
        const bf16* lq = lds + (i & 1) * 2 * WTILE;
        const bf16* lg = lq + WTILE;
        auto piece = [&](int c) {
            if (more) stage_piece(qg, p.ld, dog, p.ld_o, qstart_blk + (i + 1) * WT, p.S, lds0, (i + 1) & 1, wave, lane16, c);
        };
        if (wave_live && q0 + WT > qstart_w) {
            f32x4_t sa[2], da[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
            scores_phase<ROLE_DK>(lq, lg, kf, vf, sa, da, fr, fg, piece);
            const bf16* lsrc = ROLE_DK ? lq : lg;
            bf16x8_t ft0[8];
            tr_chunk(lsrc, 0, ft0, fg, qq, pp);        // in flight under the drain and the vector work
            mfma_drain(sa, da);
            const bool full = q0 >= qfull_w;                   // wave-uniform
            const int rel = qmin - q0 - 4 * fg;                // query 16t + r of this lane sees the key iff 16t + r >= rel
            float e8[8];
#pragma unroll
            for (int tt = 0; tt < 2; tt++) {
                const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(&lstat[i * WT + 16 * tt + 4 * fg]);
                f32x4_t s1 = f32x4_t{0.f, 0.f, 0.f, 0.f};
                if (ROLE_DK) s1 = *reinterpret_cast<const f32x4_t*>(&lstat[npad + i * WT + 16 * tt + 4 * fg]);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float s = sa[tt][r];
                    float val;
                    if (STABLE) {
                        if (ROLE_DK) {                         // s'(x) = rcp(1 + max(-x, 0))^2
                            const float w = rcpf(fmaf(fmaxf(-s, 0.f), p.scale, 1.0f));
                            val = fmaf(da[tt][r], s0[r], s1[r]) * w * w;
                        } else {
                            val = fmaxf(rcpf(fmaf(fabsf(s), p.scale, 1.0f)), fmaf(s, p.scale, 1.0f)) * s0[r];
                        }
                    } else {
                        const float pr = __expf(fmaf(s, p.scale, -s0[r]));
                        val = ROLE_DK ? pr * (da[tt][r] - s1[r]) : pr;
                    }
                    e8[4 * tt + r] = (full || 16 * tt + r >= rel) ? val : 0.f;
                }
            }
            const bf16x8_t bf = pack8(e8);                     // P (dV role) or dS (dK role), queries on the k axis
            output_phase(lsrc, ft0, bf, acc, fg, qq, pp);
        } else {
#pragma unroll
            for (int c = 0; c < WT / 4; c++) piece(c);
        }
    }
    if (kj < p.S) {
        bf16* dst = (bf16*)(ROLE_DK ? p.dk : p.dv) + ((long)b * p.S + kj) * p.ld_d + (long)h * p.d;
        const float sc = ROLE_DK ? p.scale : 1.0f;
#pragma unroll
        for (int n = 0; n < WNDV; n++) {
            const float o4[4] = {acc[n][0] * sc, acc[n][1] * sc, acc[n][2] * sc, acc[n][3] * sc};
            VecIO<bf16, 4>::store(dst + 16 * n + 4 * fg, o4);
        }
    }
}
template <bool STABLE>
__global__ __launch_bounds__(256) void attn_bwd_dkv_w_k(const AttnP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (blockIdx.y == 0) attn_bwd_dkv_w_body<STABLE, true>(p, smem);           // the longer role first
    else attn_bwd_dkv_w_body<STABLE, false>(p, smem);
}

template <typename K> int raise_lds(K kern, size_t bytes) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) MM_FAIL(MM_ERR_LAUNCH, "attention (head_dim 512): cannot raise dynamic LDS to %zu bytes: %s", bytes, hipGetErrorString(e));
    return MM_OK;
}
}  // namespace

// head_dim exactly 512 (one DMA instruction per row), pre-rotated, 16-byte aligned rows, 32-bit offsets inside a head's rows
bool attn_wide_ok(const AttnP& p) {
    return p.d == WD && p.ld % 8 == 0 && p.ld_o % 8 == 0 && (long)p.S * p.ld < (1L << 30) && p.S <= WSTAT_MAX - WT &&
           ((uintptr_t)p.q | (uintptr_t)p.k | (uintptr_t)p.v) % 16 == 0;
}

int attn_wide_fwd(const AttnP& p, int mode, hipStream_t st) {
    const dim3 grid(grid_blocks((int)cdiv(p.S, 64), p.B * p.H));
    if (mode == MM_ATTN_STABLEMAX) {
        if (int rc = raise_lds(attn_fwd_w_k<true>, WLDS_TILES)) return rc;
        hipLaunchKernelGGL((attn_fwd_w_k<true>), grid, dim3(256), WLDS_TILES, st, p);
    } else {
        if (int rc = raise_lds(attn_fwd_w_k<false>, WLDS_TILES)) return rc;
        hipLaunchKernelGGL((attn_fwd_w_k<false>), grid, dim3(256), WLDS_TILES, st, p);
    }
    MM_LAUNCH_CHECK();
    return MM_OK;
}

int attn_wide_bwd(const AttnP& p, int mode, hipStream_t st) {
    const dim3 grid(grid_blocks((int)cdiv(p.S, 64), p.B * p.H));
    const dim3 grid_kv(grid.x, 2);                                              // y: 0 = dK role, 1 = dV role
    const size_t lds_kv = WLDS_TILES + (size_t)2 * (cdiv(p.S, WT) * WT) * sizeof(float);
    if (mode == MM_ATTN_STABLEMAX) {
        if (int rc = raise_lds(attn_bwd_dq_w_k<true>, WLDS_TILES)) return rc;
        hipLaunchKernelGGL((attn_bwd_dq_w_k<true>), grid, dim3(256), WLDS_TILES, st, p);
        if (int rc = raise_lds(attn_bwd_dkv_w_k<true>, lds_kv)) return rc;
        hipLaunchKernelGGL((attn_bwd_dkv_w_k<true>), grid_kv, dim3(256), lds_kv, st, p);
    } else {
        if (int rc = raise_lds(attn_bwd_dq_w_k<false>, WLDS_TILES)) return rc;
        hipLaunchKernelGGL((attn_bwd_dq_w_k<false>), grid, dim3(256), WLDS_TILES, st, p);
        if (int rc = raise_lds(attn_bwd_dkv_w_k<false>, lds_kv)) return rc;
        hipLaunchKernelGGL((attn_bwd_dkv_w_k<false>), grid_kv, dim3(256), lds_kv, st, p);
    }
    MM_LAUNCH_CHECK();
    return MM_OK;
}
