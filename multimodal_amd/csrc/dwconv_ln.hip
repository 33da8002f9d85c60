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

static __device__ __forceinline__ void IO_f4(const float* p, float (&v)[4]) {          // 16-byte parameter load (L1/L2 resident)
    const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}

template <int VEC> static __device__ __forceinline__ void IO_fv(const float* p, float (&v)[VEC]) {   // VEC = 4 or 8 parameters
#pragma unroll
    for (int i = 0; i < VEC; i += 4) {
        const float4 t = *reinterpret_cast<const float4*>(p + i); v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w;
    }
}

// ------------------------------------------------------------------------------------------------
// forward, channels-last input, strip kernel
//
// Row pipeline per strip: the 7-row window lives unpacked in an 8-slot register ring (slot of row r = (r - l0 + 3) mod 8,
// static after unrolling by 8); QD further rows are IN FLIGHT in a queue of packed 16-byte registers.  No branch guards a
// load: row indices are clamped into the sample and rows outside it are zeroed by a select when they enter the window
// (a guarded load makes the compiler wait for it inside its branch -- measured: every row load was followed by
// s_waitcnt vmcnt(0), i.e. no prefetch at all).  Straight-line loads let it count: vmcnt(QD-1).
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int LPR, int QD>
__global__ __launch_bounds__(256) void dwconv7_ln_fwd_strip(const T* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ cb, const float* __restrict__ lw,
                                                           const float* __restrict__ lb, T* __restrict__ y,
                                                           float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                           int B, int L, int TL, int sps, float eps) {
    constexpr int SPW = 64 / LPR;
    constexpr int C = VEC * LPR;
    constexpr int RING = 8;
    static_assert(RING % QD == 0, "queue depth must divide the ring");
    typedef VecIO<T, VEC> IO;
    typedef typename IO::raw_t raw_t;
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
    auto fetch = [&](int l) -> raw_t { return IO::load_raw(xb + (long)min(max(l, 0), L - 1) * C); };   // always in bounds
    auto enter = [&](raw_t r, int l, float (&dst)[VEC]) {                   // unpack; rows outside the sample are zero padding
        IO::unpack(r, dst);
        const bool pad = (l < 0) | (l >= L);
#pragma unroll
        for (int v = 0; v < VEC; v++) dst[v] = pad ? 0.f : dst[v];
    };
    float win[RING][VEC];
    raw_t q[QD];
    {
        raw_t head[6];
#pragma unroll
        for (int s = 0; s < 6; s++) head[s] = fetch(l0 - 3 + s);            // rows l0-3 .. l0+2
#pragma unroll
        for (int s = 0; s < QD; s++) q[s] = fetch(l0 + 3 + s);              // rows l0+3 .. in flight
#pragma unroll
        for (int s = 0; s < 6; s++) enter(head[s], l0 - 3 + s, win[s]);
    }
    const float invC = 1.0f / (float)C;
    for (int base = l0; base < l1; base += RING) {
#pragma unroll
        for (int i = 0; i < RING; i++) {
            const int l = base + i;
            enter(q[i % QD], l + 3, win[(i + 6) % RING]);                   // row l+3 completes the window of row l
            q[i % QD] = fetch(l + 3 + QD);
            float acc[VEC];
            float s1 = 0.f;
#pragma unroll
            for (int v = 0; v < VEC; v++) {
                float a = bias[v];
#pragma unroll
                for (int t = 0; t < 7; t++) a = fmaf(wt[t][v], win[(i + t) % RING][v], a);
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
                IO::store_nt(yb + (long)l * C, o);
                if (mean_out && li == 0) { mean_out[(long)b * L + l] = mu; rstd_out[(long)b * L + l] = rs; }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward, channels-last input, wide kernel (any C up to NT*CH*4): one NT-thread workgroup per strip, a thread owns CH
// chunks of 4 channels.  Same row pipeline as the strip kernel (8-slot register window, QD packed rows in flight, no
// branch guards a load): every input row is loaded ONCE per strip.  LayerNorm needs sums over the whole row, i.e. over the
// workgroup: rows are finished in groups of R = 4, so the two cross-wave reductions (mean, then centred variance) cost two
// barriers per 4 rows instead of per row; the per-wave partials of the 4 rows travel through LDS as one float4.
// ------------------------------------------------------------------------------------------------
template <typename T, int NT, int CH, int QD, int VEC = 4>
__global__ __launch_bounds__(NT) void dwconv7_ln_fwd_wide(const T* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ cb, const float* __restrict__ lw,
                                                         const float* __restrict__ lb, T* __restrict__ y,
                                                         float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                         int B, int C, int L, int TL, int sps, float eps) {
    constexpr int RING = 8, R = 4, NW = NT / 64;        // VEC: channels per chunk (8 = one 16-byte load of bf16)
    static_assert(RING % QD == 0, "queue depth must divide the ring");
    typedef VecIO<T, VEC> IO;
    typedef typename IO::raw_t raw_t;
    __shared__ __attribute__((aligned(16))) float red[2][2][NW][R];      // [parity of the group][mean | M2][wave][row]
    const int b = blockIdx.x / sps;
    const int l0 = (blockIdx.x % sps) * TL;
    const int l1 = min(l0 + TL, L);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T* xb = x + (long)b * L * C;
    T* yb = y + (long)b * L * C;
    int cidx[CH]; bool cok[CH];
    float wt[CH][7][VEC], bias[CH][VEC];
    int mine = 0;                                 // channels this thread really owns
#pragma unroll
    for (int k = 0; k < CH; k++) {
        const int c = (k * NT + threadIdx.x) * VEC;
        cok[k] = c < C;
        cidx[k] = cok[k] ? c : 0;                 // idle threads shadow chunk 0 (valid addresses), contribute nothing
        mine += cok[k] ? VEC : 0;
#pragma unroll
        for (int v = 0; v < VEC; v++)
#pragma unroll
            for (int t = 0; t < 7; t++) wt[k][t][v] = w[(cidx[k] + v) * 7 + t];
        IO_fv<VEC>(cb + cidx[k], bias[k]);
    }
    const float n_w = wave_sum_u((float)mine);    // channels of this wave
    const float inv_nw = n_w > 0.f ? 1.0f / n_w : 0.f;
    auto fetch = [&](int l, int k) -> raw_t { return IO::load_raw(xb + (long)min(max(l, 0), L - 1) * C + cidx[k]); };
    auto enter = [&](raw_t r, int l, float (&dst)[VEC]) {
        IO::unpack(r, dst);
        const bool pad = (l < 0) | (l >= L);
#pragma unroll
        for (int v = 0; v < VEC; v++) dst[v] = pad ? 0.f : dst[v];
    };
    float win[CH][RING][VEC];
    raw_t q[CH][QD];
#pragma unroll
    for (int k = 0; k < CH; k++) {
        raw_t head[6];
#pragma unroll
        for (int s = 0; s < 6; s++) head[s] = fetch(l0 - 3 + s, k);
#pragma unroll
        for (int s = 0; s < QD; s++) q[k][s] = fetch(l0 + 3 + s, k);
#pragma unroll
        for (int s = 0; s < 6; s++) enter(head[s], l0 - 3 + s, win[k][s]);
    }
    // lane j < NW: the channel count of wave j (wave j owns channels [(k*NT + j*64)*4, +256) of each chunk k); 0 elsewhere
    float cntl = 0.f;
    static_assert(NW <= 8, "the cross-wave step reduces over at most 8 lanes");
    if (lane < NW) {
        int n = 0;
#pragma unroll
        for (int k = 0; k < CH; k++) n += max(0, min(C - (k * NT + lane * 64) * VEC, 64 * VEC));
        cntl = (float)n;
    }
    const float invC = 1.0f / (float)C;
    float acc[R][CH][VEC];
    int grp = 0;
    for (int base = l0; base < l1; base += RING) {
#pragma unroll
        for (int i = 0; i < RING; i++) {
            const int l = base + i;
            const int r = i % R;
#pragma unroll
            for (int k = 0; k < CH; k++) {
                enter(q[k][i % QD], l + 3, win[k][(i + 6) % RING]);
                q[k][i % QD] = fetch(l + 3 + QD, k);
            }
            __builtin_amdgcn_sched_barrier(0);    // the scheduler would sink the loads to their use (QD rows later): pin them here
#pragma unroll
            for (int k = 0; k < CH; k++) {
#pragma unroll
                for (int v = 0; v < VEC; v++) {
                    float a = bias[k][v];
#pragma unroll
                    for (int t = 0; t < 7; t++) a = fmaf(wt[k][t][v], win[k][(i + t) % RING][v], a);
                    acc[r][k][v] = a;
                }
            }
            if (r != R - 1) continue;
            // ---- the group's 4 rows.  Per wave: mean and centred sum of squares over the wave's own channels (two-pass,
            // DPP only); across the waves the (count, mean, M2) triples are combined exactly (M2 = sum M2_w + sum n_w
            // (mean_w - mean)^2): two-pass accuracy with ONE barrier per 4 rows.
            float g[CH][VEC], be[CH][VEC];
#pragma unroll
            for (int k = 0; k < CH; k++) { IO_fv<VEC>(lw + cidx[k], g[k]); IO_fv<VEC>(lb + cidx[k], be[k]); }   // issued before the barrier
            float mw[R], m2w[R];
#pragma unroll
            for (int rr = 0; rr < R; rr++) {
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < CH; k++)
#pragma unroll
                    for (int v = 0; v < VEC; v++) t += cok[k] ? acc[rr][k][v] : 0.f;
                mw[rr] = wave_sum_u(t) * inv_nw;                 // wave-uniform (SGPR)
            }
#pragma unroll
            for (int rr = 0; rr < R; rr++) {
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < CH; k++)
#pragma unroll
                    for (int v = 0; v < VEC; v++) { const float dlt = acc[rr][k][v] - mw[rr]; t = cok[k] ? fmaf(dlt, dlt, t) : t; }
                m2w[rr] = wave_sum_u(t);
            }
            const int par = grp & 1;
            grp++;
            if (lane == 0) {
                *reinterpret_cast<float4*>(red[par][0][wave]) = make_float4(mw[0], mw[1], mw[2], mw[3]);
                *reinterpret_cast<float4*>(red[par][1][wave]) = make_float4(m2w[0], m2w[1], m2w[2], m2w[3]);
            }
            __syncthreads();
            // lane j < NW holds wave j's (mean, M2) of the 4 rows; 3 DPP steps over those lanes give the row statistics
            const int jw = lane < NW ? lane : 0;
            float4 pm = *reinterpret_cast<const float4*>(red[par][0][jw]);
            float4 p2 = *reinterpret_cast<const float4*>(red[par][1][jw]);
            if (lane >= NW) { pm = make_float4(0.f, 0.f, 0.f, 0.f); p2 = pm; }
            float mu[R], var[R];
            mu[0] = first8_sum_u(cntl * pm.x) * invC; mu[1] = first8_sum_u(cntl * pm.y) * invC;
            mu[2] = first8_sum_u(cntl * pm.z) * invC; mu[3] = first8_sum_u(cntl * pm.w) * invC;
            { float dlt;
              dlt = pm.x - mu[0]; var[0] = first8_sum_u(fmaf(cntl * dlt, dlt, p2.x));
              dlt = pm.y - mu[1]; var[1] = first8_sum_u(fmaf(cntl * dlt, dlt, p2.y));
              dlt = pm.z - mu[2]; var[2] = first8_sum_u(fmaf(cntl * dlt, dlt, p2.z));
              dlt = pm.w - mu[3]; var[3] = first8_sum_u(fmaf(cntl * dlt, dlt, p2.w)); }
#pragma unroll
            for (int rr = 0; rr < R; rr++) {
                const int lr = l - (R - 1) + rr;
                const float rs = rsqrtf(var[rr] * invC + eps);
                if (lr < l1) {
#pragma unroll
                    for (int k = 0; k < CH; k++)
                        if (cok[k]) {
                            float o[VEC];
#pragma unroll
                            for (int v = 0; v < VEC; v++) o[v] = fmaf((acc[rr][k][v] - mu[rr]) * rs, g[k][v], be[k][v]);
                            IO::store_nt(yb + (long)lr * C + cidx[k], o);
                        }
                    if (mean_out && threadIdx.x == 0) { mean_out[(long)b * L + lr] = mu[rr]; rstd_out[(long)b * L + lr] = rs; }
                }
            }
        }
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
// The arithmetic is written on explicit pairs of channels (f2_t): the backend issues v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32
// on aligned register pairs directly.  Left to the SLP vectoriser the same loop held 126 v_mov_b32 per 8 rows that only
// assembled pairs it had chosen across taps, and 80 selects for the zero padding (now done on the packed row before unpacking).
// Registers: 186 (bf16) / 229 (f32) VGPRs, two waves per SIMD.  A cap at 168 or 128 (three / four waves) makes the compiler spill
// the rings to scratch inside the loop: 420 us (bf16) and 1570 us (f32) against 231 / 339 at B8192 C64 L250.
typedef float f2_t __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ f2_t fma2(f2_t a, f2_t b, f2_t c) { return __builtin_elementwise_fma(a, b, c); }
static __device__ __forceinline__ f2_t bc2(float s) { return (f2_t){s, s}; }

template <typename T, int VEC, int LPR, int QD, bool HAS_RES>
__global__ __launch_bounds__(256) void dwconv7_ln_bwd_strip(const T* __restrict__ x, const T* __restrict__ dy,
                                                           const T* __restrict__ dres, const float* __restrict__ w,
                                                           const float* __restrict__ cb, const float* __restrict__ lw,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           T* __restrict__ dx, float* __restrict__ partial,
                                                           int B, int L, int TL, int sps) {
    static_assert(VEC == 4, "two channel pairs per lane");
    constexpr int SPW = 64 / LPR;
    constexpr int C = VEC * LPR;
    constexpr int NV = VEC / 2;                  // channel pairs per lane
    constexpr int NP = 10;                       // partial vectors per channel
    constexpr int RING = 8;
    static_assert(RING % QD == 0, "queue depth must divide the ring");
    typedef VecIO<T, VEC> IO;
    typedef typename IO::raw_t raw_t;
    constexpr int RAWN = sizeof(raw_t) / 4;      // dwords of a packed row piece
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
    // Addresses: a wave-uniform base (the sample of the wave's first strip: scalar registers) plus a 32-bit byte offset per
    // lane -- the strips of a wave lie in at most SPW + 1 consecutive samples, so the offset always fits.  The loads take the
    // scalar-base form; per load the lane computes one v_med3 (row clamp) and one multiply-add instead of a 64-bit address
    // (the 64-bit form cost about 21 vector instructions per row, a sixth of the loop).
    const int bw = __builtin_amdgcn_readfirstlane(b);
    const long ubase = (long)bw * L;
    const char* xu = (const char*)(x + ubase * C);
    const char* dyu = (const char*)(dy + ubase * C);
    const char* dru = HAS_RES ? (const char*)(dres + ubase * C) : nullptr;
    const char* muu = (const char*)(mean + ubase);
    const char* rsu = (const char*)(rstd + ubase);
    char* dxu = (char*)(dx + ubase * C);
    const unsigned relrow = (unsigned)((b - bw) * L);              // rows from the uniform base to this lane's sample
    const unsigned cbytes = (unsigned)(c0 * sizeof(T));
    const int Lm1 = L - 1;
    auto rowof = [&](int l) -> unsigned {                           // clamp into the sample, then rows from the uniform base
        int r;
        asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(l), "s"(Lm1));
        return relrow + (unsigned)r;
    };
    auto ldrow = [&](const char* base, unsigned row) -> raw_t {
        return IO::load_raw_nt((const T*)(base + (row * (unsigned)(C * sizeof(T)) + cbytes)));
    };
    auto ldstat = [&](const char* base, unsigned row) -> float { return *(const float*)(base + row * 4u); };

    f2_t wt[7][NV], bias[NV], g[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) {
#pragma unroll
        for (int t = 0; t < 7; t++) wt[t][v] = (f2_t){w[(c0 + 2 * v) * 7 + t], w[(c0 + 2 * v + 1) * 7 + t]};
        bias[v] = (f2_t){cb[c0 + 2 * v], cb[c0 + 2 * v + 1]};
        g[v] = (f2_t){lw[c0 + 2 * v], lw[c0 + 2 * v + 1]};
    }
    f2_t pdw[7][NV], pdb[NV], pdg[NV], pdbeta[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) {
#pragma unroll
        for (int t = 0; t < 7; t++) pdw[t][v] = bc2(0.f);
        pdb[v] = bc2(0.f); pdg[v] = bc2(0.f); pdbeta[v] = bc2(0.f);
    }
    // Row pipeline as in the forward kernel: no branch guards a load.  Row indices are clamped into the sample, rows
    // outside it are zeroed by selects when they are consumed; x, dy, (mean, rstd) and dres each keep QD rows in flight.
    auto widen = [&](raw_t r, f2_t (&dst)[NV]) {
        float tmp[VEC];
        IO::unpack(r, tmp);
#pragma unroll
        for (int v = 0; v < NV; v++) dst[v] = (f2_t){tmp[2 * v], tmp[2 * v + 1]};
    };
    auto enter = [&](raw_t r, bool pad, f2_t (&dst)[NV]) {          // zero padding on the packed row: all-zero bits are 0.0
#pragma unroll
        for (int k = 0; k < RAWN; k++) r[k] = pad ? decltype(r[k] + 0)(0) : r[k];
        widen(r, dst);
    };
    f2_t xw[RING][NV], dc[RING][NV];
    raw_t qx[QD], qdy[QD], qdr[QD];
    float qmu[QD], qrs[QD];
    const int lw0 = l0 - 3;                      // first row of the walk
    {
        raw_t head[6];
#pragma unroll
        for (int s = 0; s < 6; s++) head[s] = ldrow(xu, rowof(lw0 - 3 + s));                         // rows l0-6 .. l0-1
#pragma unroll
        for (int s = 0; s < QD; s++) {
            qx[s] = ldrow(xu, rowof(lw0 + 3 + s));
            const unsigned rdy = rowof(lw0 + s);
            qdy[s] = ldrow(dyu, rdy);
            qmu[s] = ldstat(muu, rdy);
            qrs[s] = ldstat(rsu, rdy);
            if (HAS_RES) qdr[s] = ldrow(dru, rowof(lw0 - 3 + s));
        }
#pragma unroll
        for (int s = 0; s < 6; s++) { const int r = lw0 - 3 + s; enter(head[s], (r < 0) | (r >= L), xw[s]); }
    }
#pragma unroll
    for (int s = 0; s < RING; s++)
#pragma unroll
        for (int v = 0; v < NV; v++) dc[s][v] = bc2(0.f);

    const float invC = 1.0f / (float)C;
    const int lend = l1 + 3;
    for (int base = lw0; base < lend; base += RING) {
#pragma unroll
        for (int i = 0; i < RING; i++) {
            const int l = base + i;                         // row whose dconv is produced now
            const int k = i % QD;
            enter(qx[k], (l + 3 < 0) | (l + 3 >= L), xw[(i + 6) % RING]);      // row l+3 completes the window of row l
            qx[k] = ldrow(xu, rowof(l + 3 + QD));
            // rows outside the image are computed on zeros (mu = rs = 0 -> dconv = 0): control flow stays
            // uniform, so the DPP sums always run with full EXEC
            const bool inrow = (l >= 0) & (l < L) & (l < lend);
            f2_t gy[NV], xh[NV], gg[NV], d[NV];
            enter(qdy[k], !inrow, gy);
            const float mu = inrow ? qmu[k] : 0.f, rs = inrow ? qrs[k] : 0.f;
            const unsigned rdy = rowof(l + QD);
            qdy[k] = ldrow(dyu, rdy);
            qmu[k] = ldstat(muu, rdy);
            qrs[k] = ldstat(rsu, rdy);
#pragma unroll
            for (int v = 0; v < NV; v++) {
                f2_t a = bias[v];
#pragma unroll
                for (int t = 0; t < 7; t++) a = fma2(wt[t][v], xw[(i + t) % RING][v], a);
                xh[v] = (a - bc2(mu)) * bc2(rs);
                gg[v] = gy[v] * g[v];
            }
            // the lane's 4 terms of each row sum: pairs first, then the two halves
            const f2_t p1 = gg[0] + gg[1];
            const f2_t p2 = fma2(gg[1], xh[1], gg[0] * xh[0]);
            const float m1 = group_sum<LPR>(p1.x + p1.y) * invC;
            const float m2 = group_sum<LPR>(p2.x + p2.y) * invC;
#pragma unroll
            for (int v = 0; v < NV; v++) d[v] = bc2(rs) * ((gg[v] - bc2(m1)) - xh[v] * bc2(m2));
            if (l >= l0 && l < l1) {
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    pdg[v] = fma2(gy[v], xh[v], pdg[v]);
                    pdbeta[v] += gy[v];
                    pdb[v] += d[v];
#pragma unroll
                    for (int t = 0; t < 7; t++) pdw[t][v] = fma2(d[v], xw[(i + t) % RING][v], pdw[t][v]);
                }
            }
#pragma unroll
            for (int v = 0; v < NV; v++) dc[i][v] = d[v];  // dconv ring slot of row l is i (RING == unroll)
            // dx(l-3) = sum_t w[t] * dconv(l-3-t+3) = sum_t w[t] * dconv(l-t)
            const int lo = l - 3;
            f2_t o[NV];
            if (HAS_RES) {
                widen(qdr[k], o);
                qdr[k] = ldrow(dru, rowof(lo + QD));
            } else {
#pragma unroll
                for (int v = 0; v < NV; v++) o[v] = bc2(0.f);
            }
            if (live && lo >= l0 && lo < l1) {
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    f2_t a = o[v];
#pragma unroll
                    for (int t = 0; t < 7; t++) a = fma2(wt[t][v], dc[(i - t + RING) % RING][v], a);
                    o[v] = a;
                }
                const float of[VEC] = {o[0].x, o[0].y, o[1].x, o[1].y};
                IO::store_nt((T*)(dxu + ((relrow + (unsigned)lo) * (unsigned)(C * sizeof(T)) + cbytes)), of);
            }
        }
    }
    // ---- block-level sum of the parameter partials
    float* mine = acc_lds[wave * SPW + sub];
    const float z = live ? 1.f : 0.f;
#pragma unroll
    for (int v = 0; v < NV; v++) {
        const int c = c0 + 2 * v;
#pragma unroll
        for (int t = 0; t < 7; t++) { mine[t * C + c] = pdw[t][v].x * z; mine[t * C + c + 1] = pdw[t][v].y * z; }
        mine[7 * C + c] = pdb[v].x * z; mine[7 * C + c + 1] = pdb[v].y * z;
        mine[8 * C + c] = pdg[v].x * z; mine[8 * C + c + 1] = pdg[v].y * z;
        mine[9 * C + c] = pdbeta[v].x * z; mine[9 * C + c + 1] = pdbeta[v].y * z;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < NP * C; idx += 256) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4 * SPW; k++) s += acc_lds[k][idx];
        partial[(long)blockIdx.x * NP * C + idx] = s;
    }
}

// backward, wide kernel (C up to NT*4): one NT-thread workgroup per strip, a thread owns 4 channels; same row pipeline as
// the strip kernel (register rings for x and dconv, QD packed rows of x / dy / dres in flight, no guarded loads).  The two
// LayerNorm-backward row sums are independent of each other, so a group of R = 4 rows needs ONE barrier: per-wave sums
// (DPP, wave-uniform) -> LDS -> 3 DPP steps over the first NW lanes.  mean/rstd of a row are uniform: scalar loads.
#ifndef K1W_PREG
#define K1W_PREG 2
#endif
#ifndef K1W_QD_BF16
#define K1W_QD_BF16 4          // rows in flight per tensor: a row step of the wide kernel takes about 1 us, HBM latency under load more
#endif
#define K1W_QD(T) (sizeof(T) == 2 ? K1W_QD_BF16 : 2)
template <typename T, int NT, int QD, bool HAS_RES>
__global__ __launch_bounds__(NT) void dwconv7_ln_bwd_wide(const T* __restrict__ x, const T* __restrict__ dy,
                                                         const T* __restrict__ dres, const float* __restrict__ w,
                                                         const float* __restrict__ cb, const float* __restrict__ lw,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         T* __restrict__ dx, float* __restrict__ partial,
                                                         int B, int C, int L, int TL, int sps) {
    constexpr int VEC = 4, NV = 2, RING = 8, R = 2, NW = NT / 64;   // R = 2: a row's x window (7 rows) must still be in the 8-slot
                                                                      // ring when the group is finished (weight gradient)
    static_assert(RING % QD == 0 && NW <= 8, "queue depth must divide the ring; at most 8 waves");
    typedef VecIO<T, VEC> IO;
    typedef typename IO::raw_t raw_t;
    constexpr int RAWN = sizeof(raw_t) / 4;
    __shared__ __attribute__((aligned(16))) float red[2][NW][4];         // [parity of the group][wave][s1 row0, s1 row1, s2 row0, s2 row1]
    const int b = blockIdx.x / sps;
    const int l0 = (blockIdx.x % sps) * TL;
    const int l1 = min(l0 + TL, L);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long rowbase = (long)b * L;
    const int c = threadIdx.x * VEC;
    const bool cok = c < C;
    const int cidx = cok ? c : 0;                 // idle threads shadow chunk 0 (valid addresses and finite values); what they
                                                  // compute is never stored and is masked out of the row sums
    // row addresses are uniform (scalar registers); the only per-thread part is the 32-bit channel offset cidx
    const T* xb = x + rowbase * C;
    const T* dyb = dy + rowbase * C;
    const T* drb = HAS_RES ? dres + rowbase * C : nullptr;
    T* dxb = dx + rowbase * C;
    const float* mub = mean + rowbase;
    const float* rsb = rstd + rowbase;
    const float invC = 1.0f / (float)C;

    // arithmetic on explicit channel pairs (see the strip kernel)
    f2_t wt[7][NV], bias[NV], g[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) {
#pragma unroll
        for (int t = 0; t < 7; t++) wt[t][v] = (f2_t){w[(cidx + 2 * v) * 7 + t], w[(cidx + 2 * v + 1) * 7 + t]};
        bias[v] = (f2_t){cb[cidx + 2 * v], cb[cidx + 2 * v + 1]};
        g[v] = (f2_t){lw[cidx + 2 * v], lw[cidx + 2 * v + 1]};
    }
    // The 10 parameter-gradient partial vectors of this thread's 4 channels.  Until round 3 they had to live in LDS (thread-private
    // 16-byte slots, no synchronisation): in registers (40 of them) the SLP-paired kernel exceeded 256 VGPRs at 8 waves per
    // workgroup, and the spill reloads the compiler then placed in the loop each waited for ALL outstanding loads (vmcnt(0)).
    // PREG (the build default, K1W_PREG = 2): since the pair-arithmetic rewrite the kernel needs 152 (bf16) / 190 (f32) VGPRs, so the
    // partials fit in registers again (186-196 / 224-242, no spills) and the LDS read-modify-write -- 11 KB per wave and row, a
    // quarter of a row's time -- is gone: 338 -> 307 us (bf16), 348 -> 331 us (f32) at B2048 C1600 L31.  K1W_PREG = 0 keeps the LDS
    // slots (1: bf16 only in registers).
    constexpr bool PREG = K1W_PREG && (sizeof(T) == 2 || K1W_PREG == 2);
    extern __shared__ __attribute__((aligned(16))) mm_f4_t part[];       // [10][NT]: dw taps 0..6, db, dln_w, dln_b
    f2_t preg[PREG ? 10 : 1][NV];
    auto padd = [&](int which, const f2_t (&a)[NV]) {
        if constexpr (PREG) {
#pragma unroll
            for (int v = 0; v < NV; v++) preg[which][v] += a[v];
        } else {
            part[which * NT + threadIdx.x] += __builtin_shufflevector(a[0], a[1], 0, 1, 2, 3);
        }
    };
    if constexpr (PREG) {
#pragma unroll
        for (int j = 0; j < 10; j++)
#pragma unroll
            for (int v = 0; v < NV; v++) preg[j][v] = bc2(0.f);
    } else {
#pragma unroll
        for (int j = 0; j < 10; j++) part[j * NT + threadIdx.x] = (mm_f4_t){0.f, 0.f, 0.f, 0.f};
    }
    auto clampl = [&](int l) -> long { return (long)min(max(l, 0), L - 1); };
    auto widen = [&](raw_t r, f2_t (&dst)[NV]) {
        float tmp[VEC];
        IO::unpack(r, tmp);
#pragma unroll
        for (int v = 0; v < NV; v++) dst[v] = (f2_t){tmp[2 * v], tmp[2 * v + 1]};
    };
    auto enter = [&](raw_t r, bool pad, f2_t (&dst)[NV]) {          // zero padding on the packed row (pad is uniform)
#pragma unroll
        for (int k = 0; k < RAWN; k++) r[k] = pad ? decltype(r[k] + 0)(0) : r[k];
        widen(r, dst);
    };
    f2_t xw[RING][NV], dc[RING][NV];
    raw_t qx[QD], qdy[QD], qdr[QD];
    float qmu[QD], qrs[QD];                      // uniform (scalar loads), fetched QD rows ahead like the rest
    const int lw0 = l0 - 3;                      // first row of the walk
    {
        raw_t head[6];
#pragma unroll
        for (int s = 0; s < 6; s++) head[s] = IO::load_raw_nt(xb + clampl(lw0 - 3 + s) * C + cidx);
#pragma unroll
        for (int s = 0; s < QD; s++) {
            qx[s] = IO::load_raw_nt(xb + clampl(lw0 + 3 + s) * C + cidx);
            qdy[s] = IO::load_raw_nt(dyb + clampl(lw0 + s) * C + cidx);
            qmu[s] = mub[clampl(lw0 + s)];
            qrs[s] = rsb[clampl(lw0 + s)];
            if (HAS_RES) qdr[s] = IO::load_raw_nt(drb + clampl(lw0 - 3 + s) * C + cidx);
        }
#pragma unroll
        for (int s = 0; s < 6; s++) { const int r = lw0 - 3 + s; enter(head[s], (r < 0) | (r >= L), xw[s]); }
    }
#pragma unroll
    for (int s = 0; s < RING; s++)
#pragma unroll
        for (int v = 0; v < NV; v++) dc[s][v] = bc2(0.f);

    const int lend = l1 + 3;
    f2_t gg[R][NV], xh[R][NV], ores[R][NV];
    float rsr[R];
    f2_t acc8[NV], acc9[NV];
    int grp = 0;
    for (int base = lw0; base < lend; base += RING) {
#pragma unroll
        for (int i = 0; i < RING; i++) {
            const int l = base + i;
            const int r = i % R, k = i % QD;
            enter(qx[k], (l + 3 < 0) | (l + 3 >= L), xw[(i + 6) % RING]);
            qx[k] = IO::load_raw_nt(xb + clampl(l + 3 + QD) * C + cidx);
            const bool inrow = (l >= 0) & (l < L) & (l < lend);          // uniform
            f2_t gy[NV];
            enter(qdy[k], !inrow, gy);
            qdy[k] = IO::load_raw_nt(dyb + clampl(l + QD) * C + cidx);
            if (HAS_RES) {
                widen(qdr[k], ores[r]);                                  // dres of row l-3 (used if that row is ours)
                qdr[k] = IO::load_raw_nt(drb + clampl(l - 3 + QD) * C + cidx);
            } else {
#pragma unroll
                for (int v = 0; v < NV; v++) ores[r][v] = bc2(0.f);
            }
            __builtin_amdgcn_sched_barrier(0);    // keep the loads here: the scheduler would sink them to their use
            const float mu = inrow ? qmu[k] : 0.f;
            const float rs = inrow ? qrs[k] : 0.f;
            qmu[k] = mub[clampl(l + QD)];                                // uniform address: scalar loads
            qrs[k] = rsb[clampl(l + QD)];
            rsr[r] = rs;
            const bool own = (l >= l0) & (l < l1);                       // uniform
#pragma unroll
            for (int v = 0; v < NV; v++) {
                f2_t a = bias[v];
#pragma unroll
                for (int t = 0; t < 7; t++) a = fma2(wt[t][v], xw[(i + t) % RING][v], a);
                xh[r][v] = (a - bc2(mu)) * bc2(rs);
                gg[r][v] = gy[v] * g[v];
            }
            // parameter-gradient terms of the group's rows are summed in registers and added to the LDS partials ONCE per
            // group: a read-modify-write of all 10 partial vectors per row made the LDS stores the bottleneck (1.3k cycles
            // per row and workgroup: 0.25 of HBM on the wide shapes)
            {   // rows outside [l0, l1) contribute nothing: dy scaled by a uniform 0 / 1 (one multiply per pair, no selects, no
                // branch whose join would cost register copies)
                const float ow = own ? 1.f : 0.f;
#pragma unroll
                for (int v = 0; v < NV; v++) {
                    const f2_t gyo = gy[v] * bc2(ow);
                    if (r == 0) { acc8[v] = gyo * xh[r][v]; acc9[v] = gyo; }
                    else { acc8[v] = fma2(gyo, xh[r][v], acc8[v]); acc9[v] += gyo; }
                }
            }
            if (r != R - 1) continue;
            // ---- the group's 2 rows: both row sums with one barrier
            float s1[R], s2[R];
#pragma unroll
            for (int rr = 0; rr < R; rr++) {
                const f2_t p1 = gg[rr][0] + gg[rr][1];
                const f2_t p2 = fma2(gg[rr][1], xh[rr][1], gg[rr][0] * xh[rr][0]);
                s1[rr] = wave_sum_u(cok ? p1.x + p1.y : 0.f);
                s2[rr] = wave_sum_u(cok ? p2.x + p2.y : 0.f);
            }
            const int par = grp & 1;
            grp++;
            if (lane == 0) *reinterpret_cast<float4*>(red[par][wave]) = make_float4(s1[0], s1[1], s2[0], s2[1]);
            __syncthreads();
            float4 ps = *reinterpret_cast<const float4*>(red[par][lane < NW ? lane : 0]);
            if (lane >= NW) ps = make_float4(0.f, 0.f, 0.f, 0.f);
            float m1[R], m2[R];
            m1[0] = first8_sum_u(ps.x) * invC; m1[1] = first8_sum_u(ps.y) * invC;
            m2[0] = first8_sum_u(ps.z) * invC; m2[1] = first8_sum_u(ps.w) * invC;
#pragma unroll
            for (int rr = 0; rr < R; rr++) {
                const int ii = i - (R - 1) + rr;            // unroll index of this row (static)
                const int lr = l - (R - 1) + rr;
#pragma unroll
                for (int v = 0; v < NV; v++)                // dconv ring slot of row lr is ii (RING == unroll)
                    dc[ii][v] = bc2(rsr[rr]) * ((gg[rr][v] - bc2(m1[rr])) - xh[rr][v] * bc2(m2[rr]));
                // dx(lr-3) = sum_t w[t] * dconv(lr-t)
                const int lo = lr - 3;
                if (cok && lo >= l0 && lo < l1) {
                    f2_t o[NV];
#pragma unroll
                    for (int v = 0; v < NV; v++) {
                        f2_t a = ores[rr][v];
#pragma unroll
                        for (int t = 0; t < 7; t++) a = fma2(wt[t][v], dc[(ii - t + RING) % RING][v], a);
                        o[v] = a;
                    }
                    const float of[VEC] = {o[0].x, o[0].y, o[1].x, o[1].y};
                    IO::store_nt(dxb + (long)lo * C + cidx, of);
                }
            }
            {   // both rows' parameter terms (rows outside [l0, l1) contribute nothing), one LDS update per partial vector
                const int i0 = i - 1, i1 = i;                // unroll indices of the group's rows (static; R == 2)
                const int la = l - 1, lb = l;
                const float oa = ((la >= l0) & (la < l1)) ? 1.f : 0.f, ob = ((lb >= l0) & (lb < l1)) ? 1.f : 0.f;   // uniform
                if (oa + ob > 0.f) {
                    f2_t da[NV], db_[NV], acc[NV];
#pragma unroll
                    for (int v = 0; v < NV; v++) { da[v] = bc2(oa) * dc[i0][v]; db_[v] = bc2(ob) * dc[i1][v]; acc[v] = da[v] + db_[v]; }
                    padd(7, acc);
#pragma unroll
                    for (int t = 0; t < 7; t++) {
#pragma unroll
                        for (int v = 0; v < NV; v++) acc[v] = fma2(da[v], xw[(i0 + t) % RING][v], db_[v] * xw[(i1 + t) % RING][v]);
                        padd(t, acc);
                    }
                    padd(8, acc8);
                    padd(9, acc9);
                }
            }
        }
    }
    if (cok) {
        float* prow = partial + (long)blockIdx.x * 10 * C;
#pragma unroll
        for (int j = 0; j < 10; j++) {
            if constexpr (PREG) *reinterpret_cast<mm_f4_t*>(prow + j * C + c) = __builtin_shufflevector(preg[j][0], preg[j][1], 0, 1, 2, 3);
            else *reinterpret_cast<mm_f4_t*>(prow + j * C + c) = part[j * NT + threadIdx.x];
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

// Two-stage, coalesced form of the reduction above for many partial rows (one per workgroup of the backward kernel: 2048 at
// B = 2048): a block sums ROWS_PER_SPLIT rows of 64 consecutive columns -- a wave reads 256 contiguous bytes per row --
// into out[split][column]; the column-per-block kernel above read one float per 64-KiB stride and took 231 us of the 556 us
// backward at B2048 C1600 (profiles/r03_k1_wide_pmc.txt).  Fixed summation order: deterministic.
constexpr int RED_ROWS_PER_SPLIT = 128;
__global__ __launch_bounds__(256) void dwconv7_ln_bwd_reduce_rows(const float* __restrict__ partial, int nrows, int ncols,
                                                                float* __restrict__ out) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * RED_ROWS_PER_SPLIT, r1 = min(r0 + RED_ROWS_PER_SPLIT, nrows);
    float s[8];
#pragma unroll
    for (int u = 0; u < 8; u++) s[u] = 0.f;
    if (col < ncols) {
        for (int r = r0 + rl; r < r1; r += 32) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int rr = r + 4 * u;
                s[u] += rr < r1 ? partial[(long)rr * ncols + col] : 0.f;
            }
        }
    }
    red[rl][cl] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (rl == 0 && col < ncols) out[(long)blockIdx.y * ncols + col] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}
// final stage: one thread per output scalar over `nrows` (few) rows, consecutive threads on consecutive columns
__global__ __launch_bounds__(256) void dwconv7_ln_bwd_final(const float* __restrict__ partial, int nrows, int C, float* __restrict__ dw,
                                                          float* __restrict__ db, float* __restrict__ dlw, float* __restrict__ dlb) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 10 * C) return;
    float s = 0.f;
    for (int r = 0; r < nrows; r++) s += partial[(long)r * 10 * C + idx];
    const int kind = idx / C, c = idx % C;
    if (kind < 7) dw[c * 7 + kind] = s;
    else if (kind == 7) db[c] = s;
    else if (kind == 8) dlw[c] = s;
    else dlb[c] = s;
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
    static int qd = getenv("MM_K1_QD") ? atoi(getenv("MM_K1_QD")) : (sizeof(T) == 2 ? 8 : 4);   // rows in flight per strip (measured best)
#define LAUNCH_Q(LPR_, QD_) hipLaunchKernelGGL((dwconv7_ln_fwd_strip<T, VEC, LPR_, QD_>), grid, dim3(256), 0, st, (const T*)x, w, b, \
                                               lw, lb, (T*)y, mean, rstd, (int)B, (int)L, TL, sps, eps)
#define LAUNCH_(LPR_) do { if (qd == 2) LAUNCH_Q(LPR_, 2); else if (qd == 8) LAUNCH_Q(LPR_, 8); else LAUNCH_Q(LPR_, 4); } while (0)
    switch (lpr) {
        case 8: LAUNCH_(8); break;
        case 16: LAUNCH_(16); break;
        case 32: LAUNCH_(32); break;
        case 64: LAUNCH_(64); break;
        default: MM_FAIL(MM_ERR_UNSUPPORTED, "dwconv7_ln strip kernel: C/VEC=%d unsupported", lpr);
    }
#undef LAUNCH_
#undef LAUNCH_Q
    MM_LAUNCH_CHECK();
    return MM_OK;
}

template <typename T>
static int launch_fwd_wide(const void* x, const float* w, const float* b, const float* lw, const float* lb, void* y,
                           float* mean, float* rstd, int64_t B, int64_t C, int64_t L, float eps, hipStream_t st) {
    MM_CHECK(C % 4 == 0 && C <= 4096, "dwconv7_ln: C=%lld unsupported (need C %% 4 == 0 and C <= 4096)", (long long)C);
    // whole samples per workgroup when the batch alone fills the chip (no halo rows at all), shorter strips otherwise
    const int TL = pick_tl(B, L, 1, 4, 64);
    const int sps = (int)cdiv(L, TL);
    const dim3 grid((unsigned)(B * sps));
#define LAUNCH_(NT_, CH_, QD_) hipLaunchKernelGGL((dwconv7_ln_fwd_wide<T, NT_, CH_, QD_>), grid, dim3(NT_), 0, st, (const T*)x, w, b, \
                                                  lw, lb, (T*)y, mean, rstd, (int)B, (int)C, (int)L, TL, sps, eps)
    static int variant = getenv("MM_K1W_VARIANT") ? atoi(getenv("MM_K1W_VARIANT")) : 0;      // A/B knob
    // bf16: a thread owns 8 consecutive channels (one 16-byte load per row; the per-row LayerNorm reduction steps, which do not
    // depend on the channel count, are shared by 8 elements instead of 4): 175 -> 130 us (two 4-channel chunks) -> see
    // profiles/r03_k1_all_shapes.txt at B2048 C1600 L31
    if constexpr (sizeof(T) == 2) {
        if (C % 8 == 0 && C <= 2048 && variant != 4) {
            hipLaunchKernelGGL((dwconv7_ln_fwd_wide<T, 256, 1, 4, 8>), grid, dim3(256), 0, st, (const T*)x, w, b, lw, lb, (T*)y, mean,
                               rstd, (int)B, (int)C, (int)L, TL, sps, eps);
            MM_LAUNCH_CHECK();
            return MM_OK;
        }
    }
    if (C <= 1024) { if (variant == 1) LAUNCH_(256, 1, 8); else LAUNCH_(256, 1, 4); }
    else if (C <= 2048) {
        if (variant == 1) LAUNCH_(512, 1, 8);
        else if (variant == 2 || sizeof(T) == 2) LAUNCH_(256, 2, 4);
        else LAUNCH_(512, 1, 4);
    } else LAUNCH_(512, 2, 2);
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
    // the strip kernel addresses a wave's rows by 32-bit byte offsets from the wave's first sample: its strips lie in at most
    // spw + 1 consecutive samples, which must span less than 4 GiB (L < 500 000 at C = 256 in f32); longer samples take the wide kernel
    const bool lanes_ok = strip_ok(C, vec) && C / vec >= 16;
    const bool fits32 = lanes_ok && (64 / (C / vec) + 1) * L * C * (int64_t)4 < ((int64_t)1 << 32);   // same for both dtypes: one plan
    if (lanes_ok && fits32) {
        p.strip = true; p.lpr = (int)(C / vec);
        const int spw = 64 / p.lpr;
        // strip length cap: 64 rows in f32; 128 in bf16, where the 12 halo rows of x and 6 of dy a strip re-reads cost
        // relatively more (B8192 C64 L250: 231 -> 214 us; f32: 339 either way).  Shorter strips lose in both (32: 294 / 374 us).
        static const int tlmax_env = getenv("MM_K1B_TLMAX") ? atoi(getenv("MM_K1B_TLMAX")) : 0;   // A/B knob
        const int tlmax = tlmax_env > 0 ? tlmax_env : (dtype == MM_BF16 ? 128 : 64);
        p.TL = pick_tl(B, L, spw, 8, tlmax);
        p.sps = (int)cdiv(L, p.TL);
        p.nblocks = cdiv(cdiv(B * p.sps, spw), 4);
    } else {
        p.strip = false;
        p.TL = pick_tl(B, L, 1, 8, 64);
        p.sps = (int)cdiv(L, p.TL);
        p.nblocks = B * p.sps;
    }
    return p;
}

extern "C" size_t mm_dwconv7_ln_bwd_workspace(int64_t B, int64_t C, int64_t L) {
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    const BwdPlan p = plan_bwd(B, C, L, MM_F32);
    const size_t splits = (size_t)cdiv(p.nblocks, RED_ROWS_PER_SPLIT);           // rows of the reduction's second stage
    return ((size_t)p.nblocks + splits) * 10 * (size_t)C * sizeof(float);
}

template <typename T>
static int launch_bwd(const BwdPlan& p, const void* x, const void* dy, const void* dres, const float* w, const float* b,
                      const float* lw, const float* mean, const float* rstd, void* dx, float* partial, int64_t B,
                      int64_t C, int64_t L, hipStream_t st) {
    const dim3 grid((unsigned)p.nblocks);
    if (p.strip) {
        static int qd = getenv("MM_K1B_QD") ? atoi(getenv("MM_K1B_QD")) : 2;     // rows in flight per tensor per strip (4: one wave per SIMD, slower)
#define LAUNCH_Q(LPR_, QD_, RES_) hipLaunchKernelGGL((dwconv7_ln_bwd_strip<T, 4, LPR_, QD_, RES_>), grid, dim3(256), 0, st, (const T*)x, \
                                                     (const T*)dy, (const T*)dres, w, b, lw, mean, rstd, (T*)dx, partial, (int)B, (int)L, \
                                                     p.TL, p.sps)
#define LAUNCH_(LPR_) do { if (dres) { if (qd == 2) LAUNCH_Q(LPR_, 2, true); else LAUNCH_Q(LPR_, 4, true); } \
                           else { if (qd == 2) LAUNCH_Q(LPR_, 2, false); else LAUNCH_Q(LPR_, 4, false); } } while (0)
        switch (p.lpr) {
            case 16: LAUNCH_(16); break;
            case 32: LAUNCH_(32); break;
            case 64: LAUNCH_(64); break;
            default: MM_FAIL(MM_ERR_UNSUPPORTED, "dwconv7_ln_bwd: lpr=%d", p.lpr);
        }
#undef LAUNCH_
#undef LAUNCH_Q
    } else {
        MM_CHECK(C % 4 == 0 && C <= 2048, "dwconv7_ln_bwd: C=%lld unsupported (need C %% 4 == 0 and C <= 2048)", (long long)C);
#define LAUNCH_W(NT_, RES_) do { \
            const size_t lds_ = (K1W_PREG && (sizeof(T) == 2 || K1W_PREG == 2)) ? 0 : (size_t)10 * NT_ * sizeof(float4);   /* bf16: partials in registers */ \
            static bool attr_ = false;                 /* 80 KiB at NT = 512: above the default dynamic-LDS limit */ \
            if (!attr_) { (void)hipFuncSetAttribute((const void*)dwconv7_ln_bwd_wide<T, NT_, K1W_QD(T), RES_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_); attr_ = true; } \
            hipLaunchKernelGGL((dwconv7_ln_bwd_wide<T, NT_, K1W_QD(T), RES_>), grid, dim3(NT_), lds_, st, (const T*)x, (const T*)dy, \
                               (const T*)dres, w, b, lw, mean, rstd, (T*)dx, partial, (int)B, (int)C, (int)L, p.TL, p.sps); } while (0)
        if (C <= 1024) { if (dres) LAUNCH_W(256, true); else LAUNCH_W(256, false); }
        else { if (dres) LAUNCH_W(512, true); else LAUNCH_W(512, false); }
#undef LAUNCH_W
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
    if (p.nblocks <= 64) {                     // few partial rows (the config batch): one pass, a block per output scalar
        hipLaunchKernelGGL(dwconv7_ln_bwd_reduce, dim3((unsigned)(10 * C)), dim3(256), 0, st, partial, (int)p.nblocks, (int)C,
                           dw, db, dln_w, dln_b);
    } else {
        const int splits = (int)cdiv(p.nblocks, RED_ROWS_PER_SPLIT);
        float* part2 = partial + (size_t)p.nblocks * 10 * C;
        hipLaunchKernelGGL(dwconv7_ln_bwd_reduce_rows, dim3((unsigned)cdiv(10 * C, 64), (unsigned)splits), dim3(256), 0, st, partial,
                           (int)p.nblocks, (int)(10 * C), part2);
        hipLaunchKernelGGL(dwconv7_ln_bwd_final, dim3((unsigned)cdiv(10 * C, 256)), dim3(256), 0, st, part2, splits, (int)C, dw, db,
                           dln_w, dln_b);
    }
    MM_LAUNCH_CHECK();
    return MM_OK;
}
