// K1 / K1b: depthwise Conv1d(k=7, pad=3) fused with channels-last LayerNorm, forward and backward.
// Replaces reference models/convnext.py:38-40 and its autograd.
//
// Data layout in HBM: activations channels-last [B][L][C]; a row (b,l) is C contiguous elements.
// HBM-bound: algorithmic bytes per call = B*L*C*(s_in + s_out).
//
// "strip" kernels (C/VEC in {8,16,32,64} lanes per row): a group of LPR lanes owns a strip of TL
// consecutive positions of one sample and slides a 7-row window down it IN REGISTERS, so every input
// element is loaded from memory once (plus a 6-row halo per strip).  Loads/stores are 16 B per lane,
// LayerNorm sums are DPP reductions inside the LPR-lane group.  64/LPR strips share a wave.
// "wide" kernels (any C): one 256-thread workgroup per strip, channels spread over the block,
// cross-wave sums through LDS.
#include "common.h"
#include <stdlib.h>

#ifndef RINGB
#define RINGB 8
#endif
#ifndef RINGF
#define RINGF 8          // 7 taps + (RINGF-7) rows of prefetch; the walk is unrolled RINGF times so slot indices are static
#endif

// ------------------------------------------------------------------------------------------------
// forward, channels-last input, strip kernel
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int LPR>
__global__ __launch_bounds__(256) void dwconv7_ln_fwd_strip(const T* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ cb, const float* __restrict__ lw,
                                                           const float* __restrict__ lb, T* __restrict__ y,
                                                           float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                           int B, int L, int TL, int sps, float eps) {
    constexpr int SPW = 64 / LPR;
    constexpr int C = VEC * LPR;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int li = lane % LPR;
    long strip = ((long)blockIdx.x * 4 + wave) * SPW + lane / LPR;
    const long nstrips = (long)B * sps;
    const bool live = strip < nstrips;
    if (!live) strip = nstrips - 1;           // keep every lane running (DPP needs full EXEC); stores are masked
    const int b = (int)(strip / sps);
    const int l0 = (int)(strip % sps) * TL;
    const int l1 = min(l0 + TL, L);
    const int c0 = li * VEC;
    const T* xb = x + (long)b * L * C + c0;
    T* yb = y + (long)b * L * C + c0;

    float wt[7][VEC], bias[VEC], g[VEC], be[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) {
#pragma unroll
        for (int t = 0; t < 7; t++) wt[t][v] = w[(c0 + v) * 7 + t];
        bias[v] = cb[c0 + v]; g[v] = lw[c0 + v]; be[v] = lb[c0 + v];
    }
    float win[RINGF][VEC];
    auto load_row = [&](int l, float (&dst)[VEC]) {
        if (l >= 0 && l < L) VecIO<T, VEC>::load(xb + (long)l * C, dst);
        else {
#pragma unroll
            for (int v = 0; v < VEC; v++) dst[v] = 0.f;
        }
    };
#pragma unroll
    for (int s = 0; s < RINGF - 1; s++) load_row(l0 - 3 + s, win[s]);      // rows l0-3 .. l0+RINGF-5

    const float invC = 1.0f / (float)C;
    for (int base = l0; base < l1; base += RINGF) {
#pragma unroll
        for (int i = 0; i < RINGF; i++) {
            const int l = base + i;
            load_row(l + RINGF - 4, win[(i + RINGF - 1) % RINGF]);            // prefetch RINGF-7 rows ahead
            float acc[VEC];
            float s1 = 0.f;
#pragma unroll
            for (int v = 0; v < VEC; v++) {
                float a = bias[v];
#pragma unroll
                for (int t = 0; t < 7; t++) a = fmaf(wt[t][v], win[(i + t) % RINGF][v], a);
                acc[v] = a; s1 += a;
            }
            const float mu = group_sum<LPR>(s1) * invC;
            float s2 = 0.f;
#pragma unroll
            for (int v = 0; v < VEC; v++) { acc[v] -= mu; s2 = fmaf(acc[v], acc[v], s2); }
            const float rs = rsqrtf(group_sum<LPR>(s2) * invC + eps);
            if (live && l < l1) {
                float o[VEC];
#pragma unroll
                for (int v = 0; v < VEC; v++) o[v] = fmaf(acc[v] * rs, g[v], be[v]);
                VecIO<T, VEC>::store_nt(yb + (long)l * C, o);
                if (mean_out && li == 0) { mean_out[(long)b * L + l] = mu; rstd_out[(long)b * L + l] = rs; }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward, channels-last input, wide kernel: block = 256 threads per strip, CH chunks of VEC per thread
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int CH>
__global__ __launch_bounds__(256) void dwconv7_ln_fwd_wide(const T* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ cb, const float* __restrict__ lw,
                                                          const float* __restrict__ lb, T* __restrict__ y,
                                                          float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                          int B, int C, int L, int TL, int sps, float eps) {
    __shared__ float red[2][4];
    const int b = blockIdx.x / sps;
    const int l0 = (blockIdx.x % sps) * TL;
    const int l1 = min(l0 + TL, L);
    const T* xb = x + (long)b * L * C;
    T* yb = y + (long)b * L * C;
    int cidx[CH]; bool cok[CH];
    float wt[CH][7][VEC], bias[CH][VEC], g[CH][VEC], be[CH][VEC];
#pragma unroll
    for (int k = 0; k < CH; k++) {
        cidx[k] = (k * 256 + threadIdx.x) * VEC;
        cok[k] = cidx[k] < C;
#pragma unroll
        for (int v = 0; v < VEC; v++) {
            const int c = cok[k] ? cidx[k] + v : 0;
#pragma unroll
            for (int t = 0; t < 7; t++) wt[k][t][v] = w[c * 7 + t];
            bias[k][v] = cb[c]; g[k][v] = lw[c]; be[k][v] = lb[c];
        }
    }
    const float invC = 1.0f / (float)C;
    for (int l = l0; l < l1; l++) {
        float acc[CH][VEC];
        float s1 = 0.f;
#pragma unroll
        for (int k = 0; k < CH; k++) {
#pragma unroll
            for (int v = 0; v < VEC; v++) acc[k][v] = bias[k][v];
            if (cok[k]) {
#pragma unroll
                for (int t = 0; t < 7; t++) {
                    const int ll = l + t - 3;
                    if (ll >= 0 && ll < L) {
                        float r[VEC];
                        VecIO<T, VEC>::load(xb + (long)ll * C + cidx[k], r);
#pragma unroll
                        for (int v = 0; v < VEC; v++) acc[k][v] = fmaf(wt[k][t][v], r[v], acc[k][v]);
                    }
                }
#pragma unroll
                for (int v = 0; v < VEC; v++) s1 += acc[k][v];
            }
        }
        const float mu = block_sum<4>(s1, red[0]) * invC;
        float s2 = 0.f;
#pragma unroll
        for (int k = 0; k < CH; k++)
            if (cok[k]) {
#pragma unroll
                for (int v = 0; v < VEC; v++) { acc[k][v] -= mu; s2 = fmaf(acc[k][v], acc[k][v], s2); }
            }
        const float rs = rsqrtf(block_sum<4>(s2, red[1]) * invC + eps);
#pragma unroll
        for (int k = 0; k < CH; k++)
            if (cok[k]) {
                float o[VEC];
#pragma unroll
                for (int v = 0; v < VEC; v++) o[v] = fmaf(acc[k][v] * rs, g[k][v], be[k][v]);
                VecIO<T, VEC>::store(yb + (long)l * C + cidx[k], o);
            }
        if (mean_out && threadIdx.x == 0) { mean_out[(long)b * L + l] = mu; rstd_out[(long)b * L + l] = rs; }
    }
}

// ------------------------------------------------------------------------------------------------
// forward, channel-major input [B][C][L] (nn.Conv1d's layout): LDS-staged transpose.
// The block loads a [C][TL+6] slab with loads coalesced along L, then each wave walks positions with
// lanes over channels: 7 LDS reads per output, LayerNorm by wave/block sums, stores coalesced along C.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dwconv7_ln_fwd_bcl(const T* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ cb, const float* __restrict__ lw,
                                                         const float* __restrict__ lb, T* __restrict__ y,
                                                         float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                         int B, int C, int L, int TL, int sps, int pitch, float eps) {
    extern __shared__ __attribute__((aligned(16))) float slab[];     // [C][pitch], pitch odd >= TL+6
    const int b = blockIdx.x / sps;
    const int l0 = (blockIdx.x % sps) * TL;
    const int l1 = min(l0 + TL, L);
    const int span = TL + 6;
    const T* xb = x + (long)b * C * L;
    for (int idx = threadIdx.x; idx < C * span; idx += 256) {
        const int c = idx / span, j = idx % span;
        const int l = l0 - 3 + j;
        slab[c * pitch + j] = (l >= 0 && l < L) ? to_f(xb[(long)c * L + l]) : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float invC = 1.0f / (float)C;
    T* yb = y + (long)b * L * C;
    for (int l = l0 + wave; l < l1; l += 4) {          // one wave per position
        const int j = l - l0;
        float s1 = 0.f;
        for (int c = lane; c < C; c += 64) {
            float a = cb[c];
#pragma unroll
            for (int t = 0; t < 7; t++) a = fmaf(w[c * 7 + t], slab[c * pitch + j + t], a);
            s1 += a;
        }
        const float mu = wave_sum(s1) * invC;
        float s2 = 0.f;
        for (int c = lane; c < C; c += 64) {
            float a = cb[c];
#pragma unroll
            for (int t = 0; t < 7; t++) a = fmaf(w[c * 7 + t], slab[c * pitch + j + t], a);
            a -= mu; s2 = fmaf(a, a, s2);
        }
        const float rs = rsqrtf(wave_sum(s2) * invC + eps);
        for (int c = lane; c < C; c += 64) {
            float a = cb[c];
#pragma unroll
            for (int t = 0; t < 7; t++) a = fmaf(w[c * 7 + t], slab[c * pitch + j + t], a);
            yb[(long)l * C + c] = from_f<T>(fmaf((a - mu) * rs, lw[c], lb[c]));
        }
        if (mean_out && lane == 0) { mean_out[(long)b * L + l] = mu; rstd_out[(long)b * L + l] = rs; }
    }
}

// ------------------------------------------------------------------------------------------------
// backward, channels-last, strip kernel.  Per strip walks l = l0-3 .. l1+2:
//   conv(l) from the x ring -> xhat -> LayerNorm backward -> dconv(l) into a second ring,
//   parameter-gradient partials for l in [l0,l1), and dx(l-3) = sum_t w[t]*dconv(l-t).
// Partials go to LDS, are summed over the block's strips, and one row per block is written to the
// workspace; dwconv7_ln_bwd_reduce adds the rows in a fixed order (deterministic).
// Workspace row layout: [7][C] dw (tap-major), [C] db, [C] dln_w, [C] dln_b  = 10*C floats.
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int LPR>
__global__ __launch_bounds__(256) void dwconv7_ln_bwd_strip(const T* __restrict__ x, const T* __restrict__ dy,
                                                           const T* __restrict__ dres, const float* __restrict__ w,
                                                           const float* __restrict__ cb, const float* __restrict__ lw,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           T* __restrict__ dx, float* __restrict__ partial,
                                                           int B, int L, int TL, int sps) {
    constexpr int SPW = 64 / LPR;
    constexpr int C = VEC * LPR;
    constexpr int NP = 10;                       // partial vectors per channel
    __shared__ float acc_lds[4 * SPW][NP * C];   // 4*SPW strips * 10*C floats = 40 KiB for every (VEC,LPR)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int li = lane % LPR;
    const int sub = lane / LPR;
    long strip = ((long)blockIdx.x * 4 + wave) * SPW + sub;
    const long nstrips = (long)B * sps;
    const bool live = strip < nstrips;
    if (!live) strip = nstrips - 1;
    const int b = (int)(strip / sps);
    const int l0 = (int)(strip % sps) * TL;
    const int l1 = min(l0 + TL, L);
    const int c0 = li * VEC;
    const long rowbase = (long)b * L;
    const T* xb = x + rowbase * C + c0;
    const T* dyb = dy + rowbase * C + c0;
    T* dxb = dx + rowbase * C + c0;

    float wt[7][VEC], bias[VEC], g[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) {
#pragma unroll
        for (int t = 0; t < 7; t++) wt[t][v] = w[(c0 + v) * 7 + t];
        bias[v] = cb[c0 + v]; g[v] = lw[c0 + v];
    }
    float pdw[7][VEC], pdb[VEC], pdg[VEC], pdbeta[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) {
#pragma unroll
        for (int t = 0; t < 7; t++) pdw[t][v] = 0.f;
        pdb[v] = 0.f; pdg[v] = 0.f; pdbeta[v] = 0.f;
    }
    float xw[RINGB][VEC], dc[RINGB][VEC];
    auto load_x = [&](int l, float (&dst)[VEC]) {
        if (l >= 0 && l < L) VecIO<T, VEC>::load_nt(xb + (long)l * C, dst);
        else {
#pragma unroll
            for (int v = 0; v < VEC; v++) dst[v] = 0.f;
        }
    };
    // the walk starts at l = l0-3: x ring slots 0..6 hold rows l0-6 .. l0
#pragma unroll
    for (int s = 0; s < RINGB - 1; s++) load_x(l0 - 6 + s, xw[s]);
#pragma unroll
    for (int s = 0; s < RINGB; s++)
#pragma unroll
        for (int v = 0; v < VEC; v++) dc[s][v] = 0.f;

    const float invC = 1.0f / (float)C;
    const int lend = l1 + 3;
    for (int base = l0 - 3; base < lend; base += RINGB) {
#pragma unroll
        for (int i = 0; i < RINGB; i++) {
            const int l = base + i;                         // row whose dconv is produced now
            load_x(l + RINGB - 4, xw[(i + RINGB - 1) % RINGB]);   // prefetch RINGB-7 rows ahead
            const bool inrow = (l >= 0 && l < L && l < lend);
            // rows outside the image are computed on zeros (mu = rs = 0 -> dconv = 0): control flow stays
            // uniform, so the DPP sums always run with full EXEC
            float gy[VEC], xh[VEC], gg[VEC], d[VEC];
            float mu = 0.f, rs = 0.f;
            if (inrow) {
                VecIO<T, VEC>::load_nt(dyb + (long)l * C, gy);
                mu = mean[rowbase + l]; rs = rstd[rowbase + l];
            } else {
#pragma unroll
                for (int v = 0; v < VEC; v++) gy[v] = 0.f;
            }
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int v = 0; v < VEC; v++) {
                float a = bias[v];
#pragma unroll
                for (int t = 0; t < 7; t++) a = fmaf(wt[t][v], xw[(i + t) % RINGB][v], a);
                xh[v] = (a - mu) * rs;
                gg[v] = gy[v] * g[v];
                s1 += gg[v]; s2 = fmaf(gg[v], xh[v], s2);
            }
            const float m1 = group_sum<LPR>(s1) * invC;
            const float m2 = group_sum<LPR>(s2) * invC;
#pragma unroll
            for (int v = 0; v < VEC; v++) d[v] = rs * (gg[v] - m1 - xh[v] * m2);
            if (l >= l0 && l < l1) {
#pragma unroll
                for (int v = 0; v < VEC; v++) {
                    pdg[v] = fmaf(gy[v], xh[v], pdg[v]);
                    pdbeta[v] += gy[v];
                    pdb[v] += d[v];
#pragma unroll
                    for (int t = 0; t < 7; t++) pdw[t][v] = fmaf(d[v], xw[(i + t) % RINGB][v], pdw[t][v]);
                }
            }
#pragma unroll
            for (int v = 0; v < VEC; v++) dc[i][v] = d[v];  // dconv ring slot of row l is i (RINGB == unroll)
            // dx(l-3) = sum_t w[t] * dconv(l-3-t+3) = sum_t w[t] * dconv(l-t)
            const int lo = l - 3;
            if (live && lo >= l0 && lo < l1) {
                float o[VEC];
                if (dres) VecIO<T, VEC>::load_nt(dres + (rowbase + lo) * C + c0, o);
                else {
#pragma unroll
                    for (int v = 0; v < VEC; v++) o[v] = 0.f;
                }
#pragma unroll
                for (int v = 0; v < VEC; v++) {
                    float a = o[v];
#pragma unroll
                    for (int t = 0; t < 7; t++) a = fmaf(wt[t][v], dc[(i - t + RINGB) % RINGB][v], a);
                    o[v] = a;
                }
                VecIO<T, VEC>::store_nt(dxb + (long)lo * C, o);
            }
        }
    }
    // ---- block-level sum of the parameter partials
    float* mine = acc_lds[wave * SPW + sub];
#pragma unroll
    for (int v = 0; v < VEC; v++) {
        const float z = live ? 1.f : 0.f;
#pragma unroll
        for (int t = 0; t < 7; t++) mine[t * C + c0 + v] = pdw[t][v] * z;
        mine[7 * C + c0 + v] = pdb[v] * z;
        mine[8 * C + c0 + v] = pdg[v] * z;
        mine[9 * C + c0 + v] = pdbeta[v] * z;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < NP * C; idx += 256) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4 * SPW; k++) s += acc_lds[k][idx];
        partial[(long)blockIdx.x * NP * C + idx] = s;
    }
}

// backward, wide kernel (any C): block per strip, threads own channels; no LDS reduction of partials needed.
template <typename T, int VEC, int CH>
__global__ __launch_bounds__(256) void dwconv7_ln_bwd_wide(const T* __restrict__ x, const T* __restrict__ dy,
                                                          const T* __restrict__ dres, const float* __restrict__ w,
                                                          const float* __restrict__ cb, const float* __restrict__ lw,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          T* __restrict__ dx, float* __restrict__ partial,
                                                          int B, int C, int L, int TL, int sps) {
    __shared__ float red[2][4];
    const int b = blockIdx.x / sps;
    const int l0 = (blockIdx.x % sps) * TL;
    const int l1 = min(l0 + TL, L);
    const long rowbase = (long)b * L;
    const T* xb = x + rowbase * C;
    const T* dyb = dy + rowbase * C;
    T* dxb = dx + rowbase * C;
    const float invC = 1.0f / (float)C;
    int cidx[CH]; bool cok[CH];
    float wt[CH][7][VEC], bias[CH][VEC], g[CH][VEC];
    float pdw[CH][7][VEC], pdb[CH][VEC], pdg[CH][VEC], pdbeta[CH][VEC];
    float dc[CH][7][VEC];          // dconv of rows l-6 .. l (slot = (row mod 7)), small TL so a plain shift is fine
#pragma unroll
    for (int k = 0; k < CH; k++) {
        cidx[k] = (k * 256 + threadIdx.x) * VEC;
        cok[k] = cidx[k] < C;
#pragma unroll
        for (int v = 0; v < VEC; v++) {
            const int c = cok[k] ? cidx[k] + v : 0;
#pragma unroll
            for (int t = 0; t < 7; t++) { wt[k][t][v] = w[c * 7 + t]; pdw[k][t][v] = 0.f; dc[k][t][v] = 0.f; }
            bias[k][v] = cb[c]; g[k][v] = lw[c];
            pdb[k][v] = 0.f; pdg[k][v] = 0.f; pdbeta[k][v] = 0.f;
        }
    }
    for (int l = l0 - 3; l < l1 + 3; l++) {
        const bool inrow = (l >= 0 && l < L);
        const bool own = (l >= l0 && l < l1);
        float xh[CH][VEC], gg[CH][VEC], gy[CH][VEC], xr[CH][7][VEC];
        float s1 = 0.f, s2 = 0.f;
        float mu = 0.f, rs = 0.f;
        if (inrow) { mu = mean[rowbase + l]; rs = rstd[rowbase + l]; }
#pragma unroll
        for (int k = 0; k < CH; k++) {
#pragma unroll
            for (int v = 0; v < VEC; v++) { xh[k][v] = 0.f; gg[k][v] = 0.f; gy[k][v] = 0.f; }
            if (inrow && cok[k]) {
                float a[VEC];
#pragma unroll
                for (int v = 0; v < VEC; v++) a[v] = bias[k][v];
#pragma unroll
                for (int t = 0; t < 7; t++) {
                    const int ll = l + t - 3;
                    if (ll >= 0 && ll < L) VecIO<T, VEC>::load(xb + (long)ll * C + cidx[k], xr[k][t]);
                    else {
#pragma unroll
                        for (int v = 0; v < VEC; v++) xr[k][t][v] = 0.f;
                    }
#pragma unroll
                    for (int v = 0; v < VEC; v++) a[v] = fmaf(wt[k][t][v], xr[k][t][v], a[v]);
                }
                VecIO<T, VEC>::load(dyb + (long)l * C + cidx[k], gy[k]);
#pragma unroll
                for (int v = 0; v < VEC; v++) {
                    xh[k][v] = (a[v] - mu) * rs;
                    gg[k][v] = gy[k][v] * g[k][v];
                    s1 += gg[k][v]; s2 = fmaf(gg[k][v], xh[k][v], s2);
                }
            }
        }
        const float m1 = block_sum<4>(s1, red[0]) * invC;
        const float m2 = block_sum<4>(s2, red[1]) * invC;
#pragma unroll
        for (int k = 0; k < CH; k++) {
            // shift the dconv history: slot t holds dconv(l - t) after this update
#pragma unroll
            for (int t = 6; t > 0; t--)
#pragma unroll
                for (int v = 0; v < VEC; v++) dc[k][t][v] = dc[k][t - 1][v];
#pragma unroll
            for (int v = 0; v < VEC; v++)
                dc[k][0][v] = (inrow && cok[k]) ? rs * (gg[k][v] - m1 - xh[k][v] * m2) : 0.f;
            if (own && cok[k]) {
#pragma unroll
                for (int v = 0; v < VEC; v++) {
                    pdg[k][v] = fmaf(gy[k][v], xh[k][v], pdg[k][v]);
                    pdbeta[k][v] += gy[k][v];
                    pdb[k][v] += dc[k][0][v];
#pragma unroll
                    for (int t = 0; t < 7; t++) pdw[k][t][v] = fmaf(dc[k][0][v], xr[k][t][v], pdw[k][t][v]);
                }
            }
            const int lo = l - 3;
            if (lo >= l0 && lo < l1 && cok[k]) {
                float o[VEC];
                if (dres) VecIO<T, VEC>::load(dres + (rowbase + lo) * C + cidx[k], o);
                else {
#pragma unroll
                    for (int v = 0; v < VEC; v++) o[v] = 0.f;
                }
#pragma unroll
                for (int v = 0; v < VEC; v++) {
                    float a = o[v];
#pragma unroll
                    for (int t = 0; t < 7; t++) a = fmaf(wt[k][t][v], dc[k][t][v], a);
                    o[v] = a;
                }
                VecIO<T, VEC>::store(dxb + (long)lo * C + cidx[k], o);
            }
        }
    }
    float* prow = partial + (long)blockIdx.x * 10 * C;
#pragma unroll
    for (int k = 0; k < CH; k++)
        if (cok[k]) {
#pragma unroll
            for (int v = 0; v < VEC; v++) {
                const int c = cidx[k] + v;
#pragma unroll
                for (int t = 0; t < 7; t++) prow[t * C + c] = pdw[k][t][v];
                prow[7 * C + c] = pdb[k][v];
                prow[8 * C + c] = pdg[k][v];
                prow[9 * C + c] = pdbeta[k][v];
            }
        }
}

// second stage: fixed-order sum over workspace rows; writes dw[C][7], db, dln_w, dln_b
__global__ void dwconv7_ln_bwd_reduce(const float* __restrict__ partial, int nrows, int C, float* __restrict__ dw,
                                      float* __restrict__ db, float* __restrict__ dlw, float* __restrict__ dlb) {
    __shared__ float red[4];
    const int idx = blockIdx.x;              // one block per output scalar column (10*C blocks)
    float s = 0.f;
    for (int r = threadIdx.x; r < nrows; r += 256) s += partial[(long)r * 10 * C + idx];
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) {
        const int kind = idx / C, c = idx % C;
        if (kind < 7) dw[c * 7 + kind] = s;
        else if (kind == 7) db[c] = s;
        else if (kind == 8) dlw[c] = s;
        else dlb[c] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int pick_tl(int64_t B, int64_t L, int strips_per_unit, int tl_min, int tl_max) {
    // aim for >= 2048 scheduling units (waves or blocks); fewer, longer strips when the batch is large
    int64_t tl = (B * L) / ((int64_t)2048 * strips_per_unit);
    if (tl < tl_min) tl = tl_min;
    if (tl > tl_max) tl = tl_max;
    if (tl > L) tl = L;
    return (int)tl;
}

template <typename T, int VEC>
static int launch_fwd_strip(const void* x, const float* w, const float* b, const float* lw, const float* lb, void* y,
                            float* mean, float* rstd, int64_t B, int64_t C, int64_t L, float eps, hipStream_t st) {
    const int lpr = (int)(C / VEC);
    const int spw = 64 / lpr;
    static int tlmax = getenv("MM_K1_TLMAX") ? atoi(getenv("MM_K1_TLMAX")) : 64;
    const int TL = pick_tl(B, L, spw, 4, tlmax);
    const int sps = (int)cdiv(L, TL);
    const int64_t waves = cdiv(B * sps, spw);
    const dim3 grid((unsigned)cdiv(waves, 4));
#define LAUNCH_(LPR_) hipLaunchKernelGGL((dwconv7_ln_fwd_strip<T, VEC, LPR_>), grid, dim3(256), 0, st, (const T*)x, w, b, lw, lb, \
                                         (T*)y, mean, rstd, (int)B, (int)L, TL, sps, eps)
    switch (lpr) {
        case 8: LAUNCH_(8); break;
        case 16: LAUNCH_(16); break;
        case 32: LAUNCH_(32); break;
        case 64: LAUNCH_(64); break;
        default: MM_FAIL(MM_ERR_UNSUPPORTED, "dwconv7_ln strip kernel: C/VEC=%d unsupported", lpr);
    }
#undef LAUNCH_
    MM_LAUNCH_CHECK();
    return MM_OK;
}

template <typename T>
static int launch_fwd_wide(const void* x, const float* w, const float* b, const float* lw, const float* lb, void* y,
                           float* mean, float* rstd, int64_t B, int64_t C, int64_t L, float eps, hipStream_t st) {
    const int TL = pick_tl(B, L, 1, 1, 16);
    const int sps = (int)cdiv(L, TL);
    const dim3 grid((unsigned)(B * sps));
#define LAUNCH_(VEC_, CH_) hipLaunchKernelGGL((dwconv7_ln_fwd_wide<T, VEC_, CH_>), grid, dim3(256), 0, st, (const T*)x, w, b, lw, \
                                              lb, (T*)y, mean, rstd, (int)B, (int)C, (int)L, TL, sps, eps)
    if (C % 4 == 0 && C <= 1024) LAUNCH_(4, 1);
    else if (C % 4 == 0 && C <= 2048) LAUNCH_(4, 2);
    else if (C % 4 == 0 && C <= 4096) LAUNCH_(4, 4);
    else if (C <= 1024) LAUNCH_(1, 4);
    else MM_FAIL(MM_ERR_UNSUPPORTED, "dwconv7_ln: C=%lld unsupported (need C%%4==0 and C<=4096, or C<=1024)", (long long)C);
#undef LAUNCH_
    MM_LAUNCH_CHECK();
    return MM_OK;
}

static bool strip_ok(int64_t C, int vec) {
    if (C % vec) return false;
    const int64_t lpr = C / vec;
    return lpr == 8 || lpr == 16 || lpr == 32 || lpr == 64;
}

extern "C" int mm_dwconv7_ln_fwd(const void* x, const float* w, const float* b, const float* ln_w, const float* ln_b,
                                 void* y, float* mean, float* rstd, int64_t B, int64_t C, int64_t L, float eps,
                                 int dtype, int in_layout, void* stream) {
    MM_CHECK(x && w && b && ln_w && ln_b && y, "dwconv7_ln_fwd: null pointer");
    MM_CHECK(B > 0 && C > 0 && L > 0, "dwconv7_ln_fwd: empty shape B=%lld C=%lld L=%lld", (long long)B, (long long)C, (long long)L);
    MM_CHECK((mean == nullptr) == (rstd == nullptr), "dwconv7_ln_fwd: mean and rstd must both be given or both NULL");
    MM_CHECK(dtype == MM_F32 || dtype == MM_BF16, "dwconv7_ln_fwd: bad dtype %d", dtype);
    MM_CHECK(B * L * C < (int64_t)1 << 40, "dwconv7_ln_fwd: tensor too large");
    hipStream_t st = (hipStream_t)stream;
    if (in_layout == MM_LAYOUT_BCL) {
        // slab [C][pitch] floats in LDS, <= 96 KiB
        int TL = 64;
        while (TL > 1 && (size_t)C * (size_t)((TL + 6) | 1) * 4 > 96 * 1024) TL >>= 1;
        if (TL > L) TL = (int)L;
        const int pitch = (TL + 6) | 1;
        const size_t lds = (size_t)C * pitch * 4;
        MM_CHECK(lds <= 160 * 1024, "dwconv7_ln_fwd(BCL): C=%lld too large for the LDS slab", (long long)C);
        const int sps = (int)cdiv(L, TL);
        const dim3 grid((unsigned)(B * sps));
        if (dtype == MM_F32) {
            if (lds > 64 * 1024) hipFuncSetAttribute((const void*)dwconv7_ln_fwd_bcl<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(dwconv7_ln_fwd_bcl<float>, grid, dim3(256), lds, st, (const float*)x, w, b, ln_w, ln_b,
                               (float*)y, mean, rstd, (int)B, (int)C, (int)L, TL, sps, pitch, eps);
        } else {
            if (lds > 64 * 1024) hipFuncSetAttribute((const void*)dwconv7_ln_fwd_bcl<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(dwconv7_ln_fwd_bcl<bf16>, grid, dim3(256), lds, st, (const bf16*)x, w, b, ln_w, ln_b,
                               (bf16*)y, mean, rstd, (int)B, (int)C, (int)L, TL, sps, pitch, eps);
        }
        MM_LAUNCH_CHECK();
        return MM_OK;
    }
    MM_CHECK(in_layout == MM_LAYOUT_BLC, "dwconv7_ln_fwd: bad layout %d", in_layout);
    if (dtype == MM_F32) {
        if (strip_ok(C, 4)) return launch_fwd_strip<float, 4>(x, w, b, ln_w, ln_b, y, mean, rstd, B, C, L, eps, st);
        return launch_fwd_wide<float>(x, w, b, ln_w, ln_b, y, mean, rstd, B, C, L, eps, st);
    }
    if (strip_ok(C, 8)) return launch_fwd_strip<bf16, 8>(x, w, b, ln_w, ln_b, y, mean, rstd, B, C, L, eps, st);
    return launch_fwd_wide<bf16>(x, w, b, ln_w, ln_b, y, mean, rstd, B, C, L, eps, st);
}

// ---- backward host side
struct BwdPlan { bool strip; int vec, lpr, TL, sps; int64_t nblocks; };
static BwdPlan plan_bwd(int64_t B, int64_t C, int64_t L, int dtype) {
    BwdPlan p{};
    const int vec = 4;   // both dtypes use VEC=4 in backward (register budget: two rings + 10 partial vectors)
    p.vec = vec;
    if (strip_ok(C, vec) && C / vec >= 16) {
        p.strip = true; p.lpr = (int)(C / vec);
        const int spw = 64 / p.lpr;
        p.TL = pick_tl(B, L, spw, 8, 64);
        p.sps = (int)cdiv(L, p.TL);
        p.nblocks = cdiv(cdiv(B * p.sps, spw), 4);
    } else {
        p.strip = false;
        p.TL = pick_tl(B, L, 1, 4, 32);
        p.sps = (int)cdiv(L, p.TL);
        p.nblocks = B * p.sps;
    }
    (void)dtype;
    return p;
}

extern "C" size_t mm_dwconv7_ln_bwd_workspace(int64_t B, int64_t C, int64_t L) {
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    const BwdPlan p = plan_bwd(B, C, L, MM_F32);
    return (size_t)p.nblocks * 10 * (size_t)C * sizeof(float);
}

template <typename T>
static int launch_bwd(const BwdPlan& p, const void* x, const void* dy, const void* dres, const float* w, const float* b,
                      const float* lw, const float* mean, const float* rstd, void* dx, float* partial, int64_t B,
                      int64_t C, int64_t L, hipStream_t st) {
    const dim3 grid((unsigned)p.nblocks);
    if (p.strip) {
#define LAUNCH_(LPR_) hipLaunchKernelGGL((dwconv7_ln_bwd_strip<T, 4, LPR_>), grid, dim3(256), 0, st, (const T*)x, (const T*)dy, \
                                         (const T*)dres, w, b, lw, mean, rstd, (T*)dx, partial, (int)B, (int)L, p.TL, p.sps)
        switch (p.lpr) {
            case 16: LAUNCH_(16); break;
            case 32: LAUNCH_(32); break;
            case 64: LAUNCH_(64); break;
            default: MM_FAIL(MM_ERR_UNSUPPORTED, "dwconv7_ln_bwd: lpr=%d", p.lpr);
        }
#undef LAUNCH_
    } else {
#define LAUNCH_(VEC_, CH_) hipLaunchKernelGGL((dwconv7_ln_bwd_wide<T, VEC_, CH_>), grid, dim3(256), 0, st, (const T*)x, (const T*)dy, \
                                              (const T*)dres, w, b, lw, mean, rstd, (T*)dx, partial, (int)B, (int)C, (int)L, p.TL, p.sps)
        if (C % 4 == 0 && C <= 1024) LAUNCH_(4, 1);
        else if (C % 4 == 0 && C <= 2048) LAUNCH_(4, 2);
        else if (C <= 256) LAUNCH_(1, 1);
        else if (C <= 512) LAUNCH_(1, 2);
        else MM_FAIL(MM_ERR_UNSUPPORTED, "dwconv7_ln_bwd: C=%lld unsupported (need C%%4==0 and C<=2048, or C<=512)", (long long)C);
#undef LAUNCH_
    }
    MM_LAUNCH_CHECK();
    return MM_OK;
}

extern "C" int mm_dwconv7_ln_bwd(const void* x, const void* dy, const void* dres, const float* w, const float* b,
                                 const float* ln_w, const float* mean, const float* rstd, void* dx, float* dw, float* db,
                                 float* dln_w, float* dln_b, void* workspace, size_t workspace_bytes, int64_t B,
                                 int64_t C, int64_t L, int dtype, void* stream) {
    MM_CHECK(x && dy && w && b && ln_w && mean && rstd && dx && dw && db && dln_w && dln_b && workspace,
             "dwconv7_ln_bwd: null pointer");
    MM_CHECK(B > 0 && C > 0 && L > 0, "dwconv7_ln_bwd: empty shape");
    MM_CHECK(dtype == MM_F32 || dtype == MM_BF16, "dwconv7_ln_bwd: bad dtype %d", dtype);
    MM_CHECK(workspace_bytes >= mm_dwconv7_ln_bwd_workspace(B, C, L), "dwconv7_ln_bwd: workspace too small (%zu < %zu)",
             workspace_bytes, mm_dwconv7_ln_bwd_workspace(B, C, L));
    hipStream_t st = (hipStream_t)stream;
    const BwdPlan p = plan_bwd(B, C, L, dtype);
    float* partial = (float*)workspace;
    int rc = (dtype == MM_F32) ? launch_bwd<float>(p, x, dy, dres, w, b, ln_w, mean, rstd, dx, partial, B, C, L, st)
                               : launch_bwd<bf16>(p, x, dy, dres, w, b, ln_w, mean, rstd, dx, partial, B, C, L, st);
    if (rc != MM_OK) return rc;
    hipLaunchKernelGGL(dwconv7_ln_bwd_reduce, dim3((unsigned)(10 * C)), dim3(256), 0, st, partial, (int)p.nblocks, (int)C,
                       dw, db, dln_w, dln_b);
    MM_LAUNCH_CHECK();
    return MM_OK;
}
