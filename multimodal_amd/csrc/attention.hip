// K3 / K3b / K4: prefix-LM attention with partial RoPE, softmax or stablemax, on the matrix cores.
// Replaces reference models/transformer_decoder.py:129-153 (RoPE, scores, mask, stablemax|SDPA, @v), the
// [B,H,S,S] boolean mask of :295-298 (evaluated in-kernel from indices) and their autograd; K4 is the
// KV-cached decode step the reference does not have (train_autoregressive.py:97-101 recomputes everything).
//
// Layout: q,k,v,o are [B][S][H][d] (row stride ld elements), i.e. the nn.Linear outputs as they are.
// One workgroup = one (b,h) and 64 queries (4 waves x 16).  K/V tiles of KT keys are staged in LDS and
// shared by the 4 waves.  Products are computed TRANSPOSED so the query index sits on the lane for both:
//   S^T[key][q]  = mfma(K rows, Q rows)           (lane = q, registers = 4 keys)
//   O^T[dv][q]  += mfma(V^T (ds_read_b64_tr_b16), P^T straight from the S^T accumulator registers)
// hence the per-query softmax/stablemax state never crosses lanes except for one 4-lane sum, and no
// [S,S] score tensor ever exists in HBM (the reference materialises ~7 of them).
#include "common.h"
#include <type_traits>
#include <stdlib.h>

#include "attention.h"

template <typename T> struct AT;
template <> struct AT<bf16> { static constexpr int KT = 32, PAD = 8, KSTEP = 32; };
template <> struct AT<float> { static constexpr int KT = 16, PAD = 2, KSTEP = 4; };

// stage NROWS rows (row0..) of a [S][d] head slice into LDS [NROWS][pitch], zero padded to DP columns.
// Two phases -- all global loads first (up to 8 vectors per thread in flight), then the LDS writes -- so the
// load latencies overlap instead of serialising (the one-phase loop cost ~1 us per vector).
template <typename T, int DP, int NROWS, int NT>
__device__ __forceinline__ void stage_rows(T* lds, int pitch, const T* __restrict__ g, long ld, int row0,
                                           int S, int d, bool vec_ok) {
    constexpr int V4 = DP / 4, TOT = NROWS * V4, IT = (TOT + NT - 1) / NT;
    constexpr int CH = IT < 8 ? IT : 8;
    struct alignas(16) Vec { T e[4]; };
    for (int it0 = 0; it0 < IT; it0 += CH) {
        Vec v[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const int idx = threadIdx.x + (it0 + j) * NT;
            const int r = idx / V4, c = (idx % V4) * 4;
            const int row = row0 + r;
            if (idx < TOT && row < S && c < d) {
                const T* src = g + (long)row * ld + c;
                if (vec_ok && c + 4 <= d) {
                    if constexpr (sizeof(T) == 2) *reinterpret_cast<uint2*>(v[j].e) = *reinterpret_cast<const uint2*>(src);
                    else *reinterpret_cast<float4*>(v[j].e) = *reinterpret_cast<const float4*>(src);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; q++) v[j].e[q] = (c + q < d) ? src[q] : T(0.0f);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; q++) v[j].e[q] = T(0.0f);
            }
        }
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const int idx = threadIdx.x + (it0 + j) * NT;
            if (idx < TOT) {
                const int r = idx / V4, c = (idx % V4) * 4;
                if constexpr (sizeof(T) == 2) *reinterpret_cast<uint2*>(&lds[r * pitch + c]) = *reinterpret_cast<uint2*>(v[j].e);
                else {
#pragma unroll
                    for (int q = 0; q < 4; q++) lds[r * pitch + c + q] = v[j].e[q];
                }
            }
        }
    }
}

// NeoX half rotation of the first `rot` columns of rows staged in LDS; row r sits at absolute position row0+r.
// inverse=true applies the transpose (for gradients).
template <typename T>
__device__ __forceinline__ void rope_rows(T* lds, int pitch, int row0, int nrows, int S, int rot,
                                          const float* __restrict__ cos_t, const float* __restrict__ sin_t, bool inverse) {
    const int half = rot >> 1;
    for (int idx = threadIdx.x; idx < nrows * half; idx += blockDim.x) {
        const int r = idx / half, i = idx % half;
        const int pos = row0 + r;
        if (pos < S) {
            const float c = cos_t[(long)pos * half + i];
            float s = sin_t[(long)pos * half + i];
            if (inverse) s = -s;
            const float a = to_f(lds[r * pitch + i]), b = to_f(lds[r * pitch + i + half]);
            lds[r * pitch + i] = from_f<T>(a * c - b * s);
            lds[r * pitch + i + half] = from_f<T>(b * c + a * s);
        }
    }
}

__device__ __forceinline__ bool attn_allowed(int qi, int kj, int P, int S) {
    return kj < S && (kj <= qi || (qi < P && kj < P));
}
__device__ __forceinline__ float stable_s_f(float x) { return x < 0.f ? 1.0f / ((1.0f - x) + 1e-9f) : x + 1.0f; }
// bf16 kernels: hardware reciprocal (1 ulp) instead of the IEEE division sequence -- far below bf16 rounding
__device__ __forceinline__ float stable_s_fast(float x) { return x < 0.f ? __builtin_amdgcn_rcpf((1.0f - x) + 1e-9f) : x + 1.0f; }

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <typename T, int DP, bool STABLE>
__global__ __launch_bounds__(256) void prefix_attn_fwd_k(const AttnP p) {
    constexpr bool BF = sizeof(T) == 2;
    constexpr int KT = AT<T>::KT, PITCH = DP + AT<T>::PAD, KSTEP = AT<T>::KSTEP;
    constexpr int NKS = DP / KSTEP;          // k-steps of the QK^T product
    constexpr int NDV = DP / 16;             // output dv tiles
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* lk = reinterpret_cast<T*>(smem);
    T* lv = lk + KT * PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int q0 = blockIdx.x * 64;
    const long head_off = (long)b * p.S * p.ld + (long)h * p.d;
    const T* qg = (const T*)p.q + head_off;
    const T* kg = (const T*)p.k + head_off;
    const T* vg = (const T*)p.v + head_off;

    // ---- Q fragments of this wave's 16 queries (rotated), staged through LDS one wave at a time
    typedef typename std::conditional<BF, bf16x8_t, float>::type Frag;
    Frag qf[NKS];
    for (int w = 0; w < 4; w++) {
        stage_rows<T, DP, 16, 256>(lk, PITCH, qg, p.ld, q0 + 16 * w, p.S, p.d, p.vec_ok);
        __syncthreads();
        if (p.rot > 0) { rope_rows<T>(lk, PITCH, q0 + 16 * w, 16, p.S, p.rot, p.cos_t, p.sin_t, false); __syncthreads(); }
        if (wave == w) {
#pragma unroll
            for (int ks = 0; ks < NKS; ks++) {
                if constexpr (BF) qf[ks] = *reinterpret_cast<const bf16x8_t*>(&lk[fr * PITCH + ks * 32 + fg * 8]);
                else qf[ks] = lk[fr * PITCH + ks * 4 + fg];
            }
        }
        __syncthreads();
    }

    const int qw0 = q0 + 16 * wave;                 // first query of this wave
    const int qi = qw0 + fr;                        // this lane's query
    auto key_end = [&](int qlo, int qhi) {          // one past the last key any query in [qlo,qhi] may see
        int e = min(qhi, p.S - 1) + 1;
        if (qlo < p.P) e = max(e, p.P);
        return min(e, p.S);
    };
    const int kend_blk = key_end(q0, q0 + 63);
    const int kend_w = (qw0 < p.S) ? key_end(qw0, qw0 + 15) : 0;

    f32x4_t oacc[NDV];
#pragma unroll
    for (int n = 0; n < NDV; n++) oacc[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    for (int k0 = 0; k0 < kend_blk; k0 += KT) {
        stage_rows<T, DP, KT, 256>(lk, PITCH, kg, p.ld, k0, p.S, p.d, p.vec_ok);
        stage_rows<T, DP, KT, 256>(lv, PITCH, vg, p.ld, k0, p.S, p.d, p.vec_ok);
        __syncthreads();
        if (p.rot > 0) { rope_rows<T>(lk, PITCH, k0, KT, p.S, p.rot, p.cos_t, p.sin_t, false); __syncthreads(); }
        if (k0 < kend_w) {                           // wave-uniform
            if constexpr (BF) {
                f32x4_t sacc[2];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    sacc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < NKS; ks++) {
                        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(&lk[(16 * t + fr) * PITCH + ks * 32 + fg * 8]);
                        sacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], sacc[t], 0, 0, 0);
                    }
                }
                float pv[8];
                float tmax = -INFINITY;
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int kj = k0 + 16 * t + 4 * fg + r;
                        const float x = sacc[t][r] * p.scale;
                        const bool ok = attn_allowed(qi, kj, p.P, p.S);
                        if (STABLE) pv[4 * t + r] = ok ? stable_s_f(x) : 0.f;
                        else { pv[4 * t + r] = ok ? x : -INFINITY; tmax = fmaxf(tmax, pv[4 * t + r]); }
                    }
                if (!STABLE) {
                    tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
                    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
                    const float m_new = fmaxf(m_run, tmax);
                    const float alpha = (m_new == -INFINITY) ? 1.f : __expf(m_run - m_new);
                    l_run *= alpha;
#pragma unroll
                    for (int n = 0; n < NDV; n++) oacc[n] *= alpha;
#pragma unroll
                    for (int e = 0; e < 8; e++) pv[e] = (m_new == -INFINITY) ? 0.f : __expf(pv[e] - m_new);
                    m_run = m_new;
                }
                bf16x8_t pf;
#pragma unroll
                for (int e = 0; e < 8; e++) { l_run += pv[e]; pf[e] = (short)f2bf(pv[e]); }
                typedef bf16x4_t __attribute__((address_space(3))) * lds4p;
                const int qq = fr >> 2, pp = fr & 3;
#pragma unroll
                for (int n = 0; n < NDV; n++) {
                    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(&lv[(4 * fg + qq) * PITCH + 16 * n + 4 * pp]));
                    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(&lv[(16 + 4 * fg + qq) * PITCH + 16 * n + 4 * pp]));
                    const bf16x8_t vf = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    oacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, oacc[n], 0, 0, 0);
                }
            } else {
                f32x4_t sacc = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NKS; ks++)
                    sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(lk[fr * PITCH + ks * 4 + fg], qf[ks], sacc, 0, 0, 0);
                float pv[4];
                float tmax = -INFINITY;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int kj = k0 + 4 * fg + r;
                    const float x = sacc[r] * p.scale;
                    const bool ok = attn_allowed(qi, kj, p.P, p.S);
                    if (STABLE) pv[r] = ok ? stable_s_f(x) : 0.f;
                    else { pv[r] = ok ? x : -INFINITY; tmax = fmaxf(tmax, pv[r]); }
                }
                if (!STABLE) {
                    tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
                    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
                    const float m_new = fmaxf(m_run, tmax);
                    const float alpha = (m_new == -INFINITY) ? 1.f : expf(m_run - m_new);
                    l_run *= alpha;
#pragma unroll
                    for (int n = 0; n < NDV; n++) oacc[n] *= alpha;
#pragma unroll
                    for (int e = 0; e < 4; e++) pv[e] = (m_new == -INFINITY) ? 0.f : expf(pv[e] - m_new);
                    m_run = m_new;
                }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    l_run += pv[r];
#pragma unroll
                    for (int n = 0; n < NDV; n++)
                        oacc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(lv[(4 * fg + r) * PITCH + 16 * n + fr], pv[r], oacc[n], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // ---- finish: row sum over the 4 lanes that share a query, normalise, store
    l_run += __shfl_xor(l_run, 16);
    l_run += __shfl_xor(l_run, 32);
    const float denom = STABLE ? fmaxf(l_run, 1e-9f) : l_run;
    const float inv = 1.0f / denom;
    if (qi < p.S) {
        T* og = (T*)p.o + ((long)b * p.S + qi) * p.ld_o + (long)h * p.d;
#pragma unroll
        for (int n = 0; n < NDV; n++) {
            const int dv = 16 * n + 4 * fg;
#pragma unroll
            for (int e = 0; e < 4; e++) if (dv + e < p.d) og[dv + e] = from_f<T>(oacc[n][e] * inv);
        }
        if (p.lse && fg == 0) p.lse[((long)b * p.H + h) * p.S + qi] = STABLE ? l_run : (m_run + logf(l_run));
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 fast path (q,k already rotated by mm_rope_inplace, 8-byte aligned rows): Q fragments come straight
// from global memory (no LDS round trips in the prologue), key tiles are KT = 64 (32 for d > 256) keys, and
// the next tile's global loads are issued BEFORE the current tile is multiplied (register prefetch), so a
// tile costs one barrier pair and no exposed load latency.
// ------------------------------------------------------------------------------------------------
// LDS row padding of the fast kernels: a row stride of DP/2 + 8 dwords is an odd multiple of 8 banks, so the 8 rows x 32 bytes
// of a ds_read_b64_tr_b16 half-wave cover all 64 banks once (with + 8 elements the rows overlapped pairwise: 40 % conflict
// cycles measured), and the ds_read_b128 fragment reads stay conflict-free
constexpr int FPAD = 16;
template <int DP, int NROWS>
struct TileRegs { static constexpr int IT = NROWS * (DP / 4) / 256; uint2 v[IT]; };

template <int DP, int NROWS>
__device__ __forceinline__ void tile_load(TileRegs<DP, NROWS>& r, const bf16* __restrict__ g, long ld, int row0, int S, int d) {
    constexpr int V4 = DP / 4;
    if constexpr (256 % V4 == 0) {
        // a thread keeps its column and walks rows rr0, rr0+RS, ...: every load is (wave-uniform row base) + (one fixed
        // 32-bit per-lane offset), i.e. the scalar-base addressing form with no per-load vector address arithmetic
        constexpr int RS = 256 / V4;
        const int rr0 = threadIdx.x / V4, c = (threadIdx.x % V4) * 4;
        const unsigned voff = (unsigned)(rr0 * ld + c) * 2u;
        const char* base = reinterpret_cast<const char*>(g) + (long)row0 * ld * 2;
        const bool col_ok = c < d;
#pragma unroll
        for (int j = 0; j < TileRegs<DP, NROWS>::IT; j++) {
            const int row = row0 + rr0 + j * RS;
            if (col_ok && row < S) r.v[j] = *reinterpret_cast<const uint2*>(base + (long)(j * RS) * ld * 2 + voff);
            else r.v[j] = make_uint2(0u, 0u);
        }
    } else {
#pragma unroll
        for (int j = 0; j < TileRegs<DP, NROWS>::IT; j++) {
            const int idx = threadIdx.x + j * 256;
            const int rr = idx / V4, c = (idx % V4) * 4;
            const int row = row0 + rr;
            if (row < S && c < d) r.v[j] = *reinterpret_cast<const uint2*>(g + (long)row * ld + c);
            else r.v[j] = make_uint2(0u, 0u);
        }
    }
}
template <int DP, int NROWS>
__device__ __forceinline__ void tile_store(const TileRegs<DP, NROWS>& r, bf16* lds, int pitch) {
    constexpr int V4 = DP / 4;
#pragma unroll
    for (int j = 0; j < TileRegs<DP, NROWS>::IT; j++) {
        const int idx = threadIdx.x + j * 256;
        const int rr = idx / V4, c = (idx % V4) * 4;
        *reinterpret_cast<uint2*>(&lds[rr * pitch + c]) = r.v[j];
    }
}
// 8 consecutive bf16 of one row as an MFMA fragment, straight from global (two 8-byte loads; zero beyond d / S)
__device__ __forceinline__ bf16x8_t frag_load(const bf16* __restrict__ g, long ld, int row, int col, int S, int d) {
    uint2 a = make_uint2(0u, 0u), b = make_uint2(0u, 0u);
    if (row < S) {
        const bf16* src = g + (long)row * ld + col;
        if (col < d) a = *reinterpret_cast<const uint2*>(src);
        if (col + 4 < d) b = *reinterpret_cast<const uint2*>(src + 4);
    }
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    const u4 w = {a.x, a.y, b.x, b.y};
    return __builtin_bit_cast(bf16x8_t, w);
}

#ifndef FWD_WAVES
#define FWD_WAVES 1
#endif
template <int DP, int KT, bool STABLE>
__global__ __launch_bounds__(256, (DP <= 128 ? FWD_WAVES : 1)) void prefix_attn_fwd2_k(const AttnP p) {
    typedef bf16 T;
    constexpr int PITCH = DP + FPAD;
    constexpr int NKS = DP / 32, NDV = DP / 16, NT16 = KT / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* lk = reinterpret_cast<T*>(smem);
    T* lv = lk + KT * PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int q0 = blockIdx.x * 64;
    const long head_off = (long)b * p.S * p.ld + (long)h * p.d;
    const T* qg = (const T*)p.q + head_off;
    const T* kg = (const T*)p.k + head_off;
    const T* vg = (const T*)p.v + head_off;
    const int qw0 = q0 + 16 * wave;
    const int qi = qw0 + fr;
    auto key_end = [&](int qlo, int qhi) {
        int e = min(qhi, p.S - 1) + 1;
        if (qlo < p.P) e = max(e, p.P);
        return min(e, p.S);
    };
    const int kend_blk = key_end(q0, q0 + 63);
    const int kend_w = (qw0 < p.S) ? key_end(qw0, qw0 + 15) : 0;

    TileRegs<DP, KT> rk, rv;
    tile_load<DP, KT>(rk, kg, p.ld, 0, p.S, p.d);
    tile_load<DP, KT>(rv, vg, p.ld, 0, p.S, p.d);
    bf16x8_t qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) qf[ks] = frag_load(qg, p.ld, qi, ks * 32 + fg * 8, p.S, p.d);

    f32x4_t oacc[NDV];
#pragma unroll
    for (int n = 0; n < NDV; n++) oacc[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    typedef bf16x4_t __attribute__((address_space(3))) * lds4p;
    const int qq = fr >> 2, pp = fr & 3;

    for (int k0 = 0; k0 < kend_blk; k0 += KT) {
        tile_store<DP, KT>(rk, lk, PITCH);
        tile_store<DP, KT>(rv, lv, PITCH);
        __syncthreads();
        if (k0 + KT < kend_blk) {                       // prefetch: in flight while this tile is multiplied
            tile_load<DP, KT>(rk, kg, p.ld, k0 + KT, p.S, p.d);
            tile_load<DP, KT>(rv, vg, p.ld, k0 + KT, p.S, p.d);
        }
        if (k0 < kend_w) {
            f32x4_t sacc[NT16];
#pragma unroll
            for (int t = 0; t < NT16; t++) {
                sacc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NKS; ks++) {
                    const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(&lk[(16 * t + fr) * PITCH + ks * 32 + fg * 8]);
                    sacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], sacc[t], 0, 0, 0);
                }
            }
            float pv[4 * NT16];
            float tmax = -INFINITY;
            // interior tiles (every key visible to every query of the wave) skip the per-element mask arithmetic
            const bool full = k0 + KT <= p.S && ((k0 + KT - 1 <= qw0) || (qw0 + 15 < p.P && k0 + KT <= p.P));
            auto fill = [&](auto fc) {
                constexpr bool FULL = decltype(fc)::value;
#pragma unroll
                for (int t = 0; t < NT16; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int kj = k0 + 16 * t + 4 * fg + r;
                        const float x = sacc[t][r] * p.scale;
                        const bool ok = FULL || attn_allowed(qi, kj, p.P, p.S);
                        if (STABLE) pv[4 * t + r] = ok ? stable_s_fast(x) : 0.f;
                        else { pv[4 * t + r] = ok ? x : -INFINITY; tmax = fmaxf(tmax, pv[4 * t + r]); }
                    }
            };
            if (full) fill(std::true_type{}); else fill(std::false_type{});
            if (!STABLE) {
                tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
                const float m_new = fmaxf(m_run, tmax);
                const float alpha = (m_new == -INFINITY) ? 1.f : __expf(m_run - m_new);
                l_run *= alpha;
#pragma unroll
                for (int n = 0; n < NDV; n++) oacc[n] *= alpha;
#pragma unroll
                for (int e = 0; e < 4 * NT16; e++) pv[e] = (m_new == -INFINITY) ? 0.f : __expf(pv[e] - m_new);
                m_run = m_new;
            }
#pragma unroll
            for (int kk = 0; kk < KT / 32; kk++) {
                bf16x8_t pf;
#pragma unroll
                for (int e = 0; e < 8; e++) { l_run += pv[8 * kk + e]; pf[e] = (short)f2bf(pv[8 * kk + e]); }
                if (k0 + 32 * kk < kend_w) {             // wave-uniform: later keys are masked for every query here
#pragma unroll
                    for (int n = 0; n < NDV; n++) {
                        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(&lv[(32 * kk + 4 * fg + qq) * PITCH + 16 * n + 4 * pp]));
                        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(&lv[(32 * kk + 16 + 4 * fg + qq) * PITCH + 16 * n + 4 * pp]));
                        const bf16x8_t vf = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        oacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, oacc[n], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }
    l_run += __shfl_xor(l_run, 16);
    l_run += __shfl_xor(l_run, 32);
    const float inv = 1.0f / (STABLE ? fmaxf(l_run, 1e-9f) : l_run);
    if (qi < p.S) {
        T* og = (T*)p.o + ((long)b * p.S + qi) * p.ld_o + (long)h * p.d;
#pragma unroll
        for (int n = 0; n < NDV; n++) {
            const int dv = 16 * n + 4 * fg;
            if (dv < p.d) {                              // d % 4 == 0 on this path: whole 4-vectors only
                const float o4[4] = {oacc[n][0] * inv, oacc[n][1] * inv, oacc[n][2] * inv, oacc[n][3] * inv};
                VecIO<T, 4>::store(og + dv, o4);
            }
        }
        if (p.lse && fg == 0) p.lse[((long)b * p.H + h) * p.S + qi] = STABLE ? l_run : (m_run + logf(l_run));
    }
}

template <int DP, int KT>
static int fwd2_launch(const AttnP& p, int mode, hipStream_t st) {
    const size_t lds = (size_t)2 * KT * (DP + FPAD) * 2;
    const dim3 grid((unsigned)cdiv(p.S, 64), (unsigned)(p.B * p.H));
    if (mode == MM_ATTN_STABLEMAX) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)prefix_attn_fwd2_k<DP, KT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((prefix_attn_fwd2_k<DP, KT, true>), grid, dim3(256), lds, st, p);
    } else {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)prefix_attn_fwd2_k<DP, KT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((prefix_attn_fwd2_k<DP, KT, false>), grid, dim3(256), lds, st, p);
    }
    MM_LAUNCH_CHECK();
    return MM_OK;
}
static int fwd2_dispatch(const AttnP& p, int mode, hipStream_t st) {
    const int d = p.d;
    if (d <= 32) return fwd2_launch<32, 64>(p, mode, st);
    if (d <= 64) return fwd2_launch<64, 64>(p, mode, st);
    // 32-key tiles from d = 128 up: fewer live registers -> 3 waves per SIMD instead of 2 (measured +13 % at d = 128)
    static const bool t64 = getenv("MM_ATTN_TILE64") != nullptr;                 // A/B switch
    if (d <= 128) return t64 ? fwd2_launch<128, 64>(p, mode, st) : fwd2_launch<128, 32>(p, mode, st);
    if (d <= 192) return t64 ? fwd2_launch<192, 64>(p, mode, st) : fwd2_launch<192, 32>(p, mode, st);
    if (d <= 256) return fwd2_launch<256, 32>(p, mode, st);
    if (d <= 512) return fwd2_launch<512, 32>(p, mode, st);
    MM_FAIL(MM_ERR_UNSUPPORTED, "prefix_attn: head_dim %d > 512", d);
}

// ------------------------------------------------------------------------------------------------
// host dispatch (forward)
// ------------------------------------------------------------------------------------------------
template <typename T, int DP>
static int fwd_launch(const AttnP& p, int mode, hipStream_t st) {
    constexpr int PITCH = DP + AT<T>::PAD;
    const size_t lds = (size_t)2 * AT<T>::KT * PITCH * sizeof(T);
    const dim3 grid((unsigned)cdiv(p.S, 64), (unsigned)(p.B * p.H));
    if (mode == MM_ATTN_STABLEMAX) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)prefix_attn_fwd_k<T, DP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((prefix_attn_fwd_k<T, DP, true>), grid, dim3(256), lds, st, p);
    } else {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)prefix_attn_fwd_k<T, DP, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((prefix_attn_fwd_k<T, DP, false>), grid, dim3(256), lds, st, p);
    }
    MM_LAUNCH_CHECK();
    return MM_OK;
}

template <typename T>
static int fwd_dispatch(const AttnP& p, int mode, hipStream_t st) {
    const int d = p.d;
    if (d <= 32) return fwd_launch<T, 32>(p, mode, st);
    if (d <= 64) return fwd_launch<T, 64>(p, mode, st);
    if (d <= 128) return fwd_launch<T, 128>(p, mode, st);
    if (d <= 192) return fwd_launch<T, 192>(p, mode, st);
    if (d <= 256) return fwd_launch<T, 256>(p, mode, st);
    if (d <= 512) return fwd_launch<T, 512>(p, mode, st);
    MM_FAIL(MM_ERR_UNSUPPORTED, "prefix_attn: head_dim %d > 512", d);
}

static int attn_check(const char* who, const void* q, const void* k, const void* v, const float* cos_t, const float* sin_t,
                      int64_t B, int64_t H, int64_t S, int64_t d, int64_t rot, int64_t prompt_len, int64_t ld, int mode, int dtype) {
    MM_CHECK(q && k && v, "%s: null pointer", who);
    MM_CHECK(B > 0 && H > 0 && S > 0 && d > 0, "%s: empty shape B=%lld H=%lld S=%lld d=%lld", who, (long long)B, (long long)H, (long long)S, (long long)d);
    MM_CHECK(rot >= 0 && rot <= d && rot % 2 == 0, "%s: rot=%lld must be even and <= d", who, (long long)rot);
    MM_CHECK(rot == 0 || (cos_t && sin_t), "%s: cos/sin tables required when rot > 0", who);
    MM_CHECK(prompt_len >= 0 && prompt_len <= S, "%s: prompt_len=%lld out of range", who, (long long)prompt_len);
    MM_CHECK(ld >= H * d, "%s: ld=%lld smaller than H*d", who, (long long)ld);
    MM_CHECK(mode == MM_ATTN_SOFTMAX || mode == MM_ATTN_STABLEMAX, "%s: bad mode %d", who, mode);
    MM_CHECK(dtype == MM_F32 || dtype == MM_BF16, "%s: bad dtype %d", who, dtype);
    MM_CHECK(B * H < 65536 * 32 && S < (1 << 24), "%s: shape too large", who);
    return MM_OK;
}

static bool attn_vec_ok(const void* a, const void* b, const void* c, int64_t ld, int64_t d, int dtype) {
    const int esz = dtype == MM_BF16 ? 2 : 4;
    const uintptr_t al = 4 * esz;
    return ld % 4 == 0 && d % 4 == 0 && ((uintptr_t)a) % al == 0 && ((uintptr_t)b) % al == 0 && ((uintptr_t)c) % al == 0;
}

extern "C" int mm_prefix_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const float* cos_t,
                                  const float* sin_t, int64_t B, int64_t H, int64_t S, int64_t d, int64_t rot,
                                  int64_t prompt_len, int64_t ld_qkv, int mode, int dtype, void* stream) {
    int rc = attn_check("prefix_attn_fwd", q, k, v, cos_t, sin_t, B, H, S, d, rot, prompt_len, ld_qkv, mode, dtype);
    if (rc != MM_OK) return rc;
    MM_CHECK(o, "prefix_attn_fwd: null output");
    AttnP p{};
    p.q = q; p.k = k; p.v = v; p.o = o; p.lse = lse; p.cos_t = cos_t; p.sin_t = sin_t;
    p.B = (int)B; p.H = (int)H; p.S = (int)S; p.d = (int)d; p.rot = (int)rot; p.P = (int)prompt_len;
    p.ld = ld_qkv; p.ld_o = H * d; p.scale = (float)(1.0 / sqrt((double)d));
    p.vec_ok = attn_vec_ok(q, k, v, ld_qkv, d, dtype);
    hipStream_t st = (hipStream_t)stream;
    static const bool force_generic = getenv("MM_ATTN_GENERIC") != nullptr;     // A/B switch for debugging
    static const bool force_v2 = getenv("MM_ATTN_V2") != nullptr;               // A/B switch: the previous fast generation
    static const bool no_wide = getenv("MM_ATTN_NO_WIDE") != nullptr;           // A/B switch: head_dim 512 on the staged kernels
    static const bool no_res = getenv("MM_ATTN_NO_RES") != nullptr;             // A/B switch: tiled kernels instead of the resident ones
    if (!force_generic && dtype == MM_BF16 && rot == 0 && p.vec_ok && ((uintptr_t)o) % 8 == 0) {
        if (!force_v2 && !no_res && attn_res_ok(p)) return attn_res_fwd(p, mode, st);        // whole head resident in LDS
        if (!force_v2 && attn_fast_ok(p)) return attn_fast_fwd(p, mode, st);
        if (!force_v2 && !no_wide && attn_wide_ok(p)) return attn_wide_fwd(p, mode, st);     // head_dim 512
        return fwd2_dispatch(p, mode, st);
    }
    return dtype == MM_BF16 ? fwd_dispatch<bf16>(p, mode, st) : fwd_dispatch<float>(p, mode, st);
}


// ------------------------------------------------------------------------------------------------
// backward.  P is recomputed from q,k and the saved row normaliser (no [S,S] tensor is stored):
//   softmax:   p = exp(x - lse),            dS = p * (dP - delta)
//   stablemax: p = s(x) / L,                dS = (dP - delta) / L * s'(x),  s'(x) = x<0 ? s(x)^2 : 1
// with dP = dO.V^T and delta = rowsum(dO * O).  Two kernels, no atomics, deterministic:
//   bwd_dq : one wave per 16 queries, streams key tiles:   dq^T += K^T . dS^T
//   bwd_dkv: one wave per 16 keys,    streams query tiles: dk^T += Q^T . dS,  dv^T += dO^T . P
// Every operand tile lives in LDS; transposed operands come from ds_read_b64_tr_b16 (bf16) or plain
// strided ds_read_b32 (f32).  Gradients w.r.t. the un-rotated q,k are obtained by the inverse rotation
// in the epilogue.
// ------------------------------------------------------------------------------------------------
__global__ void attn_delta_k(const void* __restrict__ o, const void* __restrict__ d_o, float* __restrict__ delta, long rows,
                             int H, int S, int d, long ld_o, long ld_d, int is_bf16) {
    // rows = B*S*H; 16 lanes per row; delta layout [B][H][S]
    const int lane = threadIdx.x & 63, li = lane & 15;
    long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const bool live = row < rows;
    if (!live) row = rows - 1;
    const int h = (int)(row % H);
    const long bs = row / H;                  // b*S + s
    float acc = 0.f;
    if (is_bf16 == 2) {                          // bf16, 8-byte aligned rows: 4 elements per lane and load
        const bf16* op = (const bf16*)o + bs * ld_o + (long)h * d;
        const bf16* gp = (const bf16*)d_o + bs * ld_d + (long)h * d;
        for (int c = li * 4; c < d; c += 64) {
            float a[4], g[4];
            VecIO<bf16, 4>::load(op + c, a);
            VecIO<bf16, 4>::load(gp + c, g);
#pragma unroll
            for (int j = 0; j < 4; j++) acc = fmaf(a[j], g[j], acc);
        }
    } else
    for (int c = li; c < d; c += 16) {
        float a, g;
        if (is_bf16) { a = to_f(((const bf16*)o)[bs * ld_o + (long)h * d + c]); g = to_f(((const bf16*)d_o)[bs * ld_d + (long)h * d + c]); }
        else { a = ((const float*)o)[bs * ld_o + (long)h * d + c]; g = ((const float*)d_o)[bs * ld_d + (long)h * d + c]; }
        acc = fmaf(a, g, acc);
    }
    acc = group_sum<16>(acc);
    if (live && li == 0) { const long b = bs / S, s = bs % S; delta[(b * H + h) * S + s] = acc; }
}

template <int DP>
__device__ __forceinline__ void unrope_store(float* stg, int pitch, int row0, int nrows, int S, int d, int rot,
                                             const float* __restrict__ cos_t, const float* __restrict__ sin_t, bool do_rope,
                                             void* __restrict__ out, long ld, bool is_bf16) {
    // stg: [nrows][pitch] f32 gradients w.r.t. ROTATED rows; apply R^T then store rows row0.. of [S][d] (stride ld)
    const int half = rot >> 1;
    if (do_rope) {
        for (int idx = threadIdx.x; idx < nrows * half; idx += blockDim.x) {
            const int r = idx / half, i = idx % half;
            const int pos = row0 + r;
            if (pos < S) {
                const float c = cos_t[(long)pos * half + i], s = sin_t[(long)pos * half + i];
                const float a = stg[r * pitch + i], b = stg[r * pitch + i + half];
                stg[r * pitch + i] = a * c + b * s;
                stg[r * pitch + i + half] = b * c - a * s;
            }
        }
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < nrows * d; idx += blockDim.x) {
        const int r = idx / d, c = idx % d;
        const int pos = row0 + r;
        if (pos < S) {
            if (is_bf16) ((bf16*)out)[(long)pos * ld + c] = __float2bfloat16(stg[r * pitch + c]);
            else ((float*)out)[(long)pos * ld + c] = stg[r * pitch + c];
        }
    }
}

template <typename T, int DP, bool STABLE>
__global__ __launch_bounds__(64) void prefix_attn_bwd_dq_k(const AttnP p) {
    constexpr bool BF = sizeof(T) == 2;
    constexpr int KT = AT<T>::KT, PITCH = DP + AT<T>::PAD, KSTEP = AT<T>::KSTEP;
    constexpr int NKS = DP / KSTEP, NDV = DP / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* lq = reinterpret_cast<T*>(smem);
    T* ldo = lq + 16 * PITCH;
    T* lk = ldo + 16 * PITCH;
    T* lv = lk + KT * PITCH;
    const int lane = threadIdx.x, fr = lane & 15, fg = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int q0 = blockIdx.x * 16;
    const long head_off = (long)b * p.S * p.ld + (long)h * p.d;
    const T* qg = (const T*)p.q + head_off;
    const T* kg = (const T*)p.k + head_off;
    const T* vg = (const T*)p.v + head_off;
    const T* dog = (const T*)p.dO + (long)b * p.S * p.ld_o + (long)h * p.d;
    stage_rows<T, DP, 16, 64>(lq, PITCH, qg, p.ld, q0, p.S, p.d, p.vec_ok);
    stage_rows<T, DP, 16, 64>(ldo, PITCH, dog, p.ld_o, q0, p.S, p.d, p.vec_ok);
    __syncthreads();
    if (p.rot > 0) { rope_rows<T>(lq, PITCH, q0, 16, p.S, p.rot, p.cos_t, p.sin_t, false); __syncthreads(); }
    const int qi = q0 + fr;
    const bool qok = qi < p.S;
    const long stat = ((long)b * p.H + h) * p.S + (qok ? qi : 0);
    const float lse = p.lse[stat], dlt = p.delta[stat];
    const float invL = STABLE ? 1.0f / fmaxf(lse, 1e-9f) : 0.f;
    int kend = min(q0 + 15, p.S - 1) + 1;
    if (q0 < p.P) kend = max(kend, p.P);
    kend = min(kend, p.S);

    f32x4_t dqa[NDV];
#pragma unroll
    for (int n = 0; n < NDV; n++) dqa[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < kend; k0 += KT) {
        stage_rows<T, DP, KT, 64>(lk, PITCH, kg, p.ld, k0, p.S, p.d, p.vec_ok);
        stage_rows<T, DP, KT, 64>(lv, PITCH, vg, p.ld, k0, p.S, p.d, p.vec_ok);
        __syncthreads();
        if (p.rot > 0) { rope_rows<T>(lk, PITCH, k0, KT, p.S, p.rot, p.cos_t, p.sin_t, false); __syncthreads(); }
        if constexpr (BF) {
            f32x4_t sa[2], da[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                sa[t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; da[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NKS; ks++) {
                    const int off = ks * 32 + fg * 8;
                    const bf16x8_t qf = *reinterpret_cast<const bf16x8_t*>(&lq[fr * PITCH + off]);
                    const bf16x8_t gf = *reinterpret_cast<const bf16x8_t*>(&ldo[fr * PITCH + off]);
                    const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(&lk[(16 * t + fr) * PITCH + off]);
                    const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(&lv[(16 * t + fr) * PITCH + off]);
                    sa[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, sa[t], 0, 0, 0);
                    da[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, gf, da[t], 0, 0, 0);
                }
            }
            bf16x8_t dsf;
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int kj = k0 + 16 * t + 4 * fg + r;
                    const float x = sa[t][r] * p.scale;
                    float ds = 0.f;
                    if (qok && attn_allowed(qi, kj, p.P, p.S)) {
                        if (STABLE) { const float s = stable_s_f(x); ds = (da[t][r] - dlt) * invL * (x < 0.f ? s * s : 1.f); }
                        else { const float pr = __expf(x - lse); ds = pr * (da[t][r] - dlt); }
                    }
                    dsf[4 * t + r] = (short)f2bf(ds);
                }
            typedef bf16x4_t __attribute__((address_space(3))) * lds4p;
            const int qq = fr >> 2, pp = fr & 3;
#pragma unroll
            for (int n = 0; n < NDV; n++) {
                const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(&lk[(4 * fg + qq) * PITCH + 16 * n + 4 * pp]));
                const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(&lk[(16 + 4 * fg + qq) * PITCH + 16 * n + 4 * pp]));
                const bf16x8_t kt = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                dqa[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, dsf, dqa[n], 0, 0, 0);
            }
        } else {
            f32x4_t sa = f32x4_t{0.f, 0.f, 0.f, 0.f}, da = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ks++) {
                const int off = ks * 4 + fg;
                sa = __builtin_amdgcn_mfma_f32_16x16x4f32(lk[fr * PITCH + off], lq[fr * PITCH + off], sa, 0, 0, 0);
                da = __builtin_amdgcn_mfma_f32_16x16x4f32(lv[fr * PITCH + off], ldo[fr * PITCH + off], da, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int kj = k0 + 4 * fg + r;
                const float x = sa[r] * p.scale;
                float ds = 0.f;
                if (qok && attn_allowed(qi, kj, p.P, p.S)) {
                    if (STABLE) { const float s = stable_s_f(x); ds = (da[r] - dlt) * invL * (x < 0.f ? s * s : 1.f); }
                    else { const float pr = expf(x - lse); ds = pr * (da[r] - dlt); }
                }
#pragma unroll
                for (int n = 0; n < NDV; n++)
                    dqa[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(lk[(4 * fg + r) * PITCH + 16 * n + fr], ds, dqa[n], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // epilogue: dq^T -> LDS [16][DP+1] f32 -> inverse rotation -> global
    float* stg = reinterpret_cast<float*>(lk);
    constexpr int SP = DP + 1;
#pragma unroll
    for (int n = 0; n < NDV; n++)
#pragma unroll
        for (int e = 0; e < 4; e++) stg[fr * SP + 16 * n + 4 * fg + e] = dqa[n][e] * p.scale;
    __syncthreads();
    unrope_store<DP>(stg, SP, q0, 16, p.S, p.d, p.rot, p.cos_t, p.sin_t, p.rot > 0,
                     (void*)((T*)p.dq + (long)b * p.S * p.ld_d + (long)h * p.d), p.ld_d, BF);
}

template <typename T, int DP, bool STABLE>
__global__ __launch_bounds__(64) void prefix_attn_bwd_dkv_k(const AttnP p) {
    constexpr bool BF = sizeof(T) == 2;
    constexpr int QT = AT<T>::KT, PITCH = DP + AT<T>::PAD, KSTEP = AT<T>::KSTEP;
    constexpr int NKS = DP / KSTEP, NDV = DP / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* lk = reinterpret_cast<T*>(smem);
    T* lv = lk + 16 * PITCH;
    T* lq = lv + 16 * PITCH;
    T* ldo = lq + QT * PITCH;
    float* lstat = reinterpret_cast<float*>(ldo + QT * PITCH);      // [2][QT]: lse, delta
    const int lane = threadIdx.x, fr = lane & 15, fg = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int kb0 = blockIdx.x * 16;
    const long head_off = (long)b * p.S * p.ld + (long)h * p.d;
    const T* qg = (const T*)p.q + head_off;
    const T* kg = (const T*)p.k + head_off;
    const T* vg = (const T*)p.v + head_off;
    const T* dog = (const T*)p.dO + (long)b * p.S * p.ld_o + (long)h * p.d;
    stage_rows<T, DP, 16, 64>(lk, PITCH, kg, p.ld, kb0, p.S, p.d, p.vec_ok);
    stage_rows<T, DP, 16, 64>(lv, PITCH, vg, p.ld, kb0, p.S, p.d, p.vec_ok);
    __syncthreads();
    if (p.rot > 0) { rope_rows<T>(lk, PITCH, kb0, 16, p.S, p.rot, p.cos_t, p.sin_t, false); __syncthreads(); }
    const int kj = kb0 + fr;                         // this lane's key
    const long statbase = ((long)b * p.H + h) * p.S;
    // first query tile that can see any of these keys: prompt keys are seen by every query
    const int qstart = (kb0 < p.P) ? 0 : (kb0 / QT) * QT;

    f32x4_t dka[NDV], dva[NDV];
#pragma unroll
    for (int n = 0; n < NDV; n++) { dka[n] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dva[n] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }

    for (int q0 = qstart; q0 < p.S; q0 += QT) {
        stage_rows<T, DP, QT, 64>(lq, PITCH, qg, p.ld, q0, p.S, p.d, p.vec_ok);
        stage_rows<T, DP, QT, 64>(ldo, PITCH, dog, p.ld_o, q0, p.S, p.d, p.vec_ok);
        if (lane < QT) {
            const int qi = q0 + lane;
            lstat[lane] = qi < p.S ? p.lse[statbase + qi] : 1.f;
            lstat[QT + lane] = qi < p.S ? p.delta[statbase + qi] : 0.f;
        }
        __syncthreads();
        if (p.rot > 0) { rope_rows<T>(lq, PITCH, q0, QT, p.S, p.rot, p.cos_t, p.sin_t, false); __syncthreads(); }
        if constexpr (BF) {
            // S[query][key]: A = Q rows, B = K rows  -> lane = key, registers = 4 queries (x2 tiles)
            f32x4_t sa[2], da[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                sa[t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; da[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NKS; ks++) {
                    const int off = ks * 32 + fg * 8;
                    const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(&lk[fr * PITCH + off]);
                    const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(&lv[fr * PITCH + off]);
                    const bf16x8_t qf = *reinterpret_cast<const bf16x8_t*>(&lq[(16 * t + fr) * PITCH + off]);
                    const bf16x8_t gf = *reinterpret_cast<const bf16x8_t*>(&ldo[(16 * t + fr) * PITCH + off]);
                    sa[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf, sa[t], 0, 0, 0);
                    da[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, vf, da[t], 0, 0, 0);
                }
            }
            bf16x8_t pf, dsf;
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int ql = 16 * t + 4 * fg + r;
                    const int qi = q0 + ql;
                    const float x = sa[t][r] * p.scale;
                    float pr = 0.f, ds = 0.f;
                    if (qi < p.S && attn_allowed(qi, kj, p.P, p.S)) {
                        const float lse = lstat[ql], dlt = lstat[QT + ql];
                        if (STABLE) {
                            const float s = stable_s_f(x), invL = 1.0f / fmaxf(lse, 1e-9f);
                            pr = s * invL; ds = (da[t][r] - dlt) * invL * (x < 0.f ? s * s : 1.f);
                        } else { pr = __expf(x - lse); ds = pr * (da[t][r] - dlt); }
                    }
                    pf[4 * t + r] = (short)f2bf(pr); dsf[4 * t + r] = (short)f2bf(ds);
                }
            typedef bf16x4_t __attribute__((address_space(3))) * lds4p;
            const int qq = fr >> 2, pp = fr & 3;
#pragma unroll
            for (int n = 0; n < NDV; n++) {
                const bf16x4_t glo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(&ldo[(4 * fg + qq) * PITCH + 16 * n + 4 * pp]));
                const bf16x4_t ghi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(&ldo[(16 + 4 * fg + qq) * PITCH + 16 * n + 4 * pp]));
                const bf16x8_t gt = bf16x8_t{glo[0], glo[1], glo[2], glo[3], ghi[0], ghi[1], ghi[2], ghi[3]};
                dva[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt, pf, dva[n], 0, 0, 0);
                const bf16x4_t qlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(&lq[(4 * fg + qq) * PITCH + 16 * n + 4 * pp]));
                const bf16x4_t qhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(&lq[(16 + 4 * fg + qq) * PITCH + 16 * n + 4 * pp]));
                const bf16x8_t qt = bf16x8_t{qlo[0], qlo[1], qlo[2], qlo[3], qhi[0], qhi[1], qhi[2], qhi[3]};
                dka[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, dsf, dka[n], 0, 0, 0);
            }
        } else {
            f32x4_t sa = f32x4_t{0.f, 0.f, 0.f, 0.f}, da = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ks++) {
                const int off = ks * 4 + fg;
                sa = __builtin_amdgcn_mfma_f32_16x16x4f32(lq[fr * PITCH + off], lk[fr * PITCH + off], sa, 0, 0, 0);
                da = __builtin_amdgcn_mfma_f32_16x16x4f32(ldo[fr * PITCH + off], lv[fr * PITCH + off], da, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int ql = 4 * fg + r;
                const int qi = q0 + ql;
                const float x = sa[r] * p.scale;
                float pr = 0.f, ds = 0.f;
                if (qi < p.S && attn_allowed(qi, kj, p.P, p.S)) {
                    const float lse = lstat[ql], dlt = lstat[QT + ql];
                    if (STABLE) {
                        const float s = stable_s_f(x), invL = 1.0f / fmaxf(lse, 1e-9f);
                        pr = s * invL; ds = (da[r] - dlt) * invL * (x < 0.f ? s * s : 1.f);
                    } else { pr = expf(x - lse); ds = pr * (da[r] - dlt); }
                }
#pragma unroll
                for (int n = 0; n < NDV; n++) {
                    dva[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(ldo[ql * PITCH + 16 * n + fr], pr, dva[n], 0, 0, 0);
                    dka[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(lq[ql * PITCH + 16 * n + fr], ds, dka[n], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // epilogue through LDS (f32 [16][DP+1]); the Q/dO tile region is free now
    float* stg = reinterpret_cast<float*>(lq);
    constexpr int SP = DP + 1;
#pragma unroll
    for (int n = 0; n < NDV; n++)
#pragma unroll
        for (int e = 0; e < 4; e++) stg[fr * SP + 16 * n + 4 * fg + e] = dva[n][e];
    __syncthreads();
    unrope_store<DP>(stg, SP, kb0, 16, p.S, p.d, 0, p.cos_t, p.sin_t, false,
                     (void*)((T*)p.dv + (long)b * p.S * p.ld_d + (long)h * p.d), p.ld_d, BF);
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NDV; n++)
#pragma unroll
        for (int e = 0; e < 4; e++) stg[fr * SP + 16 * n + 4 * fg + e] = dka[n][e] * p.scale;
    __syncthreads();
    unrope_store<DP>(stg, SP, kb0, 16, p.S, p.d, p.rot, p.cos_t, p.sin_t, p.rot > 0,
                     (void*)((T*)p.dk + (long)b * p.S * p.ld_d + (long)h * p.d), p.ld_d, BF);
}


// ------------------------------------------------------------------------------------------------
// bf16 backward, 4 waves per workgroup: the streamed tiles (K/V for dQ, Q/dO for dK/dV) are staged once
// per 64 stationary rows and shared by the 4 waves; each wave keeps its 16 stationary rows as MFMA
// B-fragments in registers.  Same arithmetic and lane maps as the one-wave kernels above.
// ------------------------------------------------------------------------------------------------
template <int DP, bool STABLE>
__global__ __launch_bounds__(256) void prefix_attn_bwd_dq4_k(const AttnP p) {
    typedef bf16 T;
    constexpr int KT = 32, PITCH = DP + 8;
    constexpr int NKS = DP / 32, NDV = DP / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* lk = reinterpret_cast<T*>(smem);
    T* lv = lk + KT * PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int q0 = blockIdx.x * 64;
    const long head_off = (long)b * p.S * p.ld + (long)h * p.d;
    const T* qg = (const T*)p.q + head_off;
    const T* kg = (const T*)p.k + head_off;
    const T* vg = (const T*)p.v + head_off;
    const T* dog = (const T*)p.dO + (long)b * p.S * p.ld_o + (long)h * p.d;
    bf16x8_t qf[NKS], gf[NKS];
    for (int w = 0; w < 4; w++) {
        stage_rows<T, DP, 16, 256>(lk, PITCH, qg, p.ld, q0 + 16 * w, p.S, p.d, p.vec_ok);
        stage_rows<T, DP, 16, 256>(lv, PITCH, dog, p.ld_o, q0 + 16 * w, p.S, p.d, p.vec_ok);
        __syncthreads();
        if (p.rot > 0) { rope_rows<T>(lk, PITCH, q0 + 16 * w, 16, p.S, p.rot, p.cos_t, p.sin_t, false); __syncthreads(); }
        if (wave == w) {
#pragma unroll
            for (int ks = 0; ks < NKS; ks++) {
                qf[ks] = *reinterpret_cast<const bf16x8_t*>(&lk[fr * PITCH + ks * 32 + fg * 8]);
                gf[ks] = *reinterpret_cast<const bf16x8_t*>(&lv[fr * PITCH + ks * 32 + fg * 8]);
            }
        }
        __syncthreads();
    }
    const int qw0 = q0 + 16 * wave;
    const int qi = qw0 + fr;
    const bool qok = qi < p.S;
    const long stat = ((long)b * p.H + h) * p.S + (qok ? qi : 0);
    const float lse = p.lse[stat], dlt = p.delta[stat];
    const float invL = STABLE ? 1.0f / fmaxf(lse, 1e-9f) : 0.f;
    auto key_end = [&](int qlo, int qhi) {
        int e = min(qhi, p.S - 1) + 1;
        if (qlo < p.P) e = max(e, p.P);
        return min(e, p.S);
    };
    const int kend_blk = key_end(q0, q0 + 63);
    const int kend_w = (qw0 < p.S) ? key_end(qw0, qw0 + 15) : 0;

    f32x4_t dqa[NDV];
#pragma unroll
    for (int n = 0; n < NDV; n++) dqa[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < kend_blk; k0 += KT) {
        stage_rows<T, DP, KT, 256>(lk, PITCH, kg, p.ld, k0, p.S, p.d, p.vec_ok);
        stage_rows<T, DP, KT, 256>(lv, PITCH, vg, p.ld, k0, p.S, p.d, p.vec_ok);
        __syncthreads();
        if (p.rot > 0) { rope_rows<T>(lk, PITCH, k0, KT, p.S, p.rot, p.cos_t, p.sin_t, false); __syncthreads(); }
        if (k0 < kend_w) {
            f32x4_t sa[2], da[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                sa[t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; da[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NKS; ks++) {
                    const int off = ks * 32 + fg * 8;
                    const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(&lk[(16 * t + fr) * PITCH + off]);
                    const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(&lv[(16 * t + fr) * PITCH + off]);
                    sa[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], sa[t], 0, 0, 0);
                    da[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, gf[ks], da[t], 0, 0, 0);
                }
            }
            bf16x8_t dsf;
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int kj = k0 + 16 * t + 4 * fg + r;
                    const float x = sa[t][r] * p.scale;
                    float ds = 0.f;
                    if (qok && attn_allowed(qi, kj, p.P, p.S)) {
                        if (STABLE) { const float sx = stable_s_f(x); ds = (da[t][r] - dlt) * invL * (x < 0.f ? sx * sx : 1.f); }
                        else { const float pr = __expf(x - lse); ds = pr * (da[t][r] - dlt); }
                    }
                    dsf[4 * t + r] = (short)f2bf(ds);
                }
            typedef bf16x4_t __attribute__((address_space(3))) * lds4p;
            const int qq = fr >> 2, pp = fr & 3;
#pragma unroll
            for (int n = 0; n < NDV; n++) {
                const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(&lk[(4 * fg + qq) * PITCH + 16 * n + 4 * pp]));
                const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(&lk[(16 + 4 * fg + qq) * PITCH + 16 * n + 4 * pp]));
                const bf16x8_t kt = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                dqa[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, dsf, dqa[n], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    float* stg = reinterpret_cast<float*>(lk);
    constexpr int SP = DP + 1;
    for (int w = 0; w < 4; w++) {
        if (wave == w) {
#pragma unroll
            for (int n = 0; n < NDV; n++)
#pragma unroll
                for (int e = 0; e < 4; e++) stg[fr * SP + 16 * n + 4 * fg + e] = dqa[n][e] * p.scale;
        }
        __syncthreads();
        unrope_store<DP>(stg, SP, q0 + 16 * w, 16, p.S, p.d, p.rot, p.cos_t, p.sin_t, p.rot > 0,
                         (void*)((T*)p.dq + (long)b * p.S * p.ld_d + (long)h * p.d), p.ld_d, true);
        __syncthreads();
    }
}

template <int DP, bool STABLE>
__global__ __launch_bounds__(256) void prefix_attn_bwd_dkv4_k(const AttnP p) {
    typedef bf16 T;
    constexpr int QT = 32, PITCH = DP + 8;
    constexpr int NKS = DP / 32, NDV = DP / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* lq = reinterpret_cast<T*>(smem);
    T* ldo = lq + QT * PITCH;
    float* lstat = reinterpret_cast<float*>(ldo + QT * PITCH);       // [2][QT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int kb0 = blockIdx.x * 64;
    const long head_off = (long)b * p.S * p.ld + (long)h * p.d;
    const T* qg = (const T*)p.q + head_off;
    const T* kg = (const T*)p.k + head_off;
    const T* vg = (const T*)p.v + head_off;
    const T* dog = (const T*)p.dO + (long)b * p.S * p.ld_o + (long)h * p.d;
    bf16x8_t kf[NKS], vf[NKS];
    for (int w = 0; w < 4; w++) {
        stage_rows<T, DP, 16, 256>(lq, PITCH, kg, p.ld, kb0 + 16 * w, p.S, p.d, p.vec_ok);
        stage_rows<T, DP, 16, 256>(ldo, PITCH, vg, p.ld, kb0 + 16 * w, p.S, p.d, p.vec_ok);
        __syncthreads();
        if (p.rot > 0) { rope_rows<T>(lq, PITCH, kb0 + 16 * w, 16, p.S, p.rot, p.cos_t, p.sin_t, false); __syncthreads(); }
        if (wave == w) {
#pragma unroll
            for (int ks = 0; ks < NKS; ks++) {
                kf[ks] = *reinterpret_cast<const bf16x8_t*>(&lq[fr * PITCH + ks * 32 + fg * 8]);
                vf[ks] = *reinterpret_cast<const bf16x8_t*>(&ldo[fr * PITCH + ks * 32 + fg * 8]);
            }
        }
        __syncthreads();
    }
    const int kw0 = kb0 + 16 * wave;
    const int kj = kw0 + fr;
    const long statbase = ((long)b * p.H + h) * p.S;
    const int qstart_blk = (kb0 < p.P) ? 0 : (kb0 / QT) * QT;
    const int qstart_w = (kw0 < p.P) ? 0 : kw0;        // first query that can see any key of this wave
    const bool wave_live = kw0 < p.S;

    f32x4_t dka[NDV], dva[NDV];
#pragma unroll
    for (int n = 0; n < NDV; n++) { dka[n] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dva[n] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }

    for (int q0 = qstart_blk; q0 < p.S; q0 += QT) {
        stage_rows<T, DP, QT, 256>(lq, PITCH, qg, p.ld, q0, p.S, p.d, p.vec_ok);
        stage_rows<T, DP, QT, 256>(ldo, PITCH, dog, p.ld_o, q0, p.S, p.d, p.vec_ok);
        if (tid < QT) {
            const int qi = q0 + tid;
            lstat[tid] = qi < p.S ? p.lse[statbase + qi] : 1.f;
            lstat[QT + tid] = qi < p.S ? p.delta[statbase + qi] : 0.f;
        }
        __syncthreads();
        if (p.rot > 0) { rope_rows<T>(lq, PITCH, q0, QT, p.S, p.rot, p.cos_t, p.sin_t, false); __syncthreads(); }
        if (wave_live && q0 + QT > qstart_w) {
            f32x4_t sa[2], da[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                sa[t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; da[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NKS; ks++) {
                    const int off = ks * 32 + fg * 8;
                    const bf16x8_t qf = *reinterpret_cast<const bf16x8_t*>(&lq[(16 * t + fr) * PITCH + off]);
                    const bf16x8_t gf = *reinterpret_cast<const bf16x8_t*>(&ldo[(16 * t + fr) * PITCH + off]);
                    sa[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf[ks], sa[t], 0, 0, 0);
                    da[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, vf[ks], da[t], 0, 0, 0);
                }
            }
            bf16x8_t pf, dsf;
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int ql = 16 * t + 4 * fg + r;
                    const int qi = q0 + ql;
                    const float x = sa[t][r] * p.scale;
                    float pr = 0.f, ds = 0.f;
                    if (qi < p.S && attn_allowed(qi, kj, p.P, p.S)) {
                        const float lse = lstat[ql], dlt = lstat[QT + ql];
                        if (STABLE) {
                            const float sx = stable_s_f(x), invL = 1.0f / fmaxf(lse, 1e-9f);
                            pr = sx * invL; ds = (da[t][r] - dlt) * invL * (x < 0.f ? sx * sx : 1.f);
                        } else { pr = __expf(x - lse); ds = pr * (da[t][r] - dlt); }
                    }
                    pf[4 * t + r] = (short)f2bf(pr); dsf[4 * t + r] = (short)f2bf(ds);
                }
            typedef bf16x4_t __attribute__((address_space(3))) * lds4p;
            const int qq = fr >> 2, pp = fr & 3;
#pragma unroll
            for (int n = 0; n < NDV; n++) {
                const bf16x4_t glo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(&ldo[(4 * fg + qq) * PITCH + 16 * n + 4 * pp]));
                const bf16x4_t ghi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(&ldo[(16 + 4 * fg + qq) * PITCH + 16 * n + 4 * pp]));
                const bf16x8_t gt = bf16x8_t{glo[0], glo[1], glo[2], glo[3], ghi[0], ghi[1], ghi[2], ghi[3]};
                dva[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt, pf, dva[n], 0, 0, 0);
                const bf16x4_t qlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(&lq[(4 * fg + qq) * PITCH + 16 * n + 4 * pp]));
                const bf16x4_t qhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(&lq[(16 + 4 * fg + qq) * PITCH + 16 * n + 4 * pp]));
                const bf16x8_t qt = bf16x8_t{qlo[0], qlo[1], qlo[2], qlo[3], qhi[0], qhi[1], qhi[2], qhi[3]};
                dka[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, dsf, dka[n], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    float* stg = reinterpret_cast<float*>(lq);
    constexpr int SP = DP + 1;
    for (int w = 0; w < 4; w++) {
        if (wave == w) {
#pragma unroll
            for (int n = 0; n < NDV; n++)
#pragma unroll
                for (int e = 0; e < 4; e++) stg[fr * SP + 16 * n + 4 * fg + e] = dva[n][e];
        }
        __syncthreads();
        unrope_store<DP>(stg, SP, kb0 + 16 * w, 16, p.S, p.d, 0, p.cos_t, p.sin_t, false,
                         (void*)((T*)p.dv + (long)b * p.S * p.ld_d + (long)h * p.d), p.ld_d, true);
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int n = 0; n < NDV; n++)
#pragma unroll
                for (int e = 0; e < 4; e++) stg[fr * SP + 16 * n + 4 * fg + e] = dka[n][e] * p.scale;
        }
        __syncthreads();
        unrope_store<DP>(stg, SP, kb0 + 16 * w, 16, p.S, p.d, p.rot, p.cos_t, p.sin_t, p.rot > 0,
                         (void*)((T*)p.dk + (long)b * p.S * p.ld_d + (long)h * p.d), p.ld_d, true);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 backward fast path (q,k already rotated: rot == 0; 8-byte aligned rows), same structure as
// prefix_attn_fwd2_k: stationary fragments straight from global memory, streamed tiles of 64 rows (32 for d > 256)
// held in registers and prefetched one tile ahead, gradients stored from the accumulators (a lane owns 4 consecutive
// head-dim columns of one row): no LDS prologue / epilogue passes.
// ------------------------------------------------------------------------------------------------
#ifndef DQ_WAVES
#define DQ_WAVES 1
#endif
template <int DP, int KT, bool STABLE>
__global__ __launch_bounds__(256, (DP <= 128 ? DQ_WAVES : 1)) void prefix_attn_bwd_dq5_k(const AttnP p) {
    typedef bf16 T;
    constexpr int PITCH = DP + FPAD;
    constexpr int NKS = DP / 32, NDV = DP / 16, NT16 = KT / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* lk = reinterpret_cast<T*>(smem);
    T* lv = lk + KT * PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int q0 = blockIdx.x * 64;
    const long head_off = (long)b * p.S * p.ld + (long)h * p.d;
    const T* qg = (const T*)p.q + head_off;
    const T* kg = (const T*)p.k + head_off;
    const T* vg = (const T*)p.v + head_off;
    const T* dog = (const T*)p.dO + (long)b * p.S * p.ld_o + (long)h * p.d;
    const int qw0 = q0 + 16 * wave;
    const int qi = qw0 + fr;
    const bool qok = qi < p.S;
    auto key_end = [&](int qlo, int qhi) {
        int e = min(qhi, p.S - 1) + 1;
        if (qlo < p.P) e = max(e, p.P);
        return min(e, p.S);
    };
    const int kend_blk = key_end(q0, q0 + 63);
    const int kend_w = (qw0 < p.S) ? key_end(qw0, qw0 + 15) : 0;

    TileRegs<DP, KT> rk, rv;
    tile_load<DP, KT>(rk, kg, p.ld, 0, p.S, p.d);
    tile_load<DP, KT>(rv, vg, p.ld, 0, p.S, p.d);
    bf16x8_t qf[NKS], gf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) {
        qf[ks] = frag_load(qg, p.ld, qi, ks * 32 + fg * 8, p.S, p.d);
        gf[ks] = frag_load(dog, p.ld_o, qi, ks * 32 + fg * 8, p.S, p.d);
    }
    const long stat = ((long)b * p.H + h) * p.S + (qok ? qi : 0);
    const float lse = p.lse[stat], dlt = p.delta[stat];
    const float invL = STABLE ? 1.0f / fmaxf(lse, 1e-9f) : 0.f;

    f32x4_t dqa[NDV];
#pragma unroll
    for (int n = 0; n < NDV; n++) dqa[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    typedef bf16x4_t __attribute__((address_space(3))) * lds4p;
    const int qq = fr >> 2, pp = fr & 3;

    for (int k0 = 0; k0 < kend_blk; k0 += KT) {
        tile_store<DP, KT>(rk, lk, PITCH);
        tile_store<DP, KT>(rv, lv, PITCH);
        __syncthreads();
        if (k0 + KT < kend_blk) {
            tile_load<DP, KT>(rk, kg, p.ld, k0 + KT, p.S, p.d);
            tile_load<DP, KT>(rv, vg, p.ld, k0 + KT, p.S, p.d);
        }
        if (k0 < kend_w) {
            f32x4_t sa[NT16], da[NT16];
#pragma unroll
            for (int t = 0; t < NT16; t++) {
                sa[t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; da[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NKS; ks++) {
                    const int off = ks * 32 + fg * 8;
                    const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(&lk[(16 * t + fr) * PITCH + off]);
                    const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(&lv[(16 * t + fr) * PITCH + off]);
                    sa[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], sa[t], 0, 0, 0);
                    da[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, gf[ks], da[t], 0, 0, 0);
                }
            }
            const bool full = k0 + KT <= p.S && qw0 + 15 < p.S && ((k0 + KT - 1 <= qw0) || (qw0 + 15 < p.P && k0 + KT <= p.P));
#pragma unroll
            for (int kk = 0; kk < KT / 32; kk++) {
                bf16x8_t dsf;
                auto fill = [&](auto fc) {
                    constexpr bool FULL = decltype(fc)::value;
#pragma unroll
                    for (int tt = 0; tt < 2; tt++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int t = 2 * kk + tt;
                            const int kj = k0 + 16 * t + 4 * fg + r;
                            const float x = sa[t][r] * p.scale;
                            float ds = 0.f;
                            if (FULL || (qok && attn_allowed(qi, kj, p.P, p.S))) {
                                if (STABLE) { const float sx = stable_s_fast(x); ds = (da[t][r] - dlt) * invL * (x < 0.f ? sx * sx : 1.f); }
                                else { const float pr = __expf(x - lse); ds = pr * (da[t][r] - dlt); }
                            }
                            dsf[4 * tt + r] = (short)f2bf(ds);
                        }
                };
                if (full) fill(std::true_type{}); else fill(std::false_type{});
                if (k0 + 32 * kk < kend_w) {             // wave-uniform: later keys are masked for every query here
#pragma unroll
                    for (int n = 0; n < NDV; n++) {
                        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(&lk[(32 * kk + 4 * fg + qq) * PITCH + 16 * n + 4 * pp]));
                        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(&lk[(32 * kk + 16 + 4 * fg + qq) * PITCH + 16 * n + 4 * pp]));
                        const bf16x8_t kt = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        dqa[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, dsf, dqa[n], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }
    if (qok) {
        T* dq = (T*)p.dq + ((long)b * p.S + qi) * p.ld_d + (long)h * p.d;
#pragma unroll
        for (int n = 0; n < NDV; n++) {
            const int dv = 16 * n + 4 * fg;
            if (dv < p.d) {
                const float o4[4] = {dqa[n][0] * p.scale, dqa[n][1] * p.scale, dqa[n][2] * p.scale, dqa[n][3] * p.scale};
                VecIO<T, 4>::store(dq + dv, o4);
            }
        }
    }
}

#ifndef DKV_WAVES
#define DKV_WAVES 1
#endif
template <int DP, int QT, bool STABLE>
__global__ __launch_bounds__(256, (DP <= 128 ? DKV_WAVES : 1)) void prefix_attn_bwd_dkv5_k(const AttnP p) {
    typedef bf16 T;
    constexpr int PITCH = DP + FPAD;
    constexpr int NKS = DP / 32, NDV = DP / 16, NT16 = QT / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* lq = reinterpret_cast<T*>(smem);
    T* ldo = lq + QT * PITCH;
    float* lstat = reinterpret_cast<float*>(ldo + QT * PITCH);       // [2][QT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int kb0 = blockIdx.x * 64;
    const long head_off = (long)b * p.S * p.ld + (long)h * p.d;
    const T* qg = (const T*)p.q + head_off;
    const T* kg = (const T*)p.k + head_off;
    const T* vg = (const T*)p.v + head_off;
    const T* dog = (const T*)p.dO + (long)b * p.S * p.ld_o + (long)h * p.d;
    const int kw0 = kb0 + 16 * wave;
    const int kj = kw0 + fr;
    const long statbase = ((long)b * p.H + h) * p.S;
    const int qstart_blk = (kb0 < p.P) ? 0 : (kb0 / QT) * QT;
    const int qstart_w = (kw0 < p.P) ? 0 : kw0;        // first query that can see any key of this wave
    const bool wave_live = kw0 < p.S;

    TileRegs<DP, QT> rq, rg;
    tile_load<DP, QT>(rq, qg, p.ld, qstart_blk, p.S, p.d);
    tile_load<DP, QT>(rg, dog, p.ld_o, qstart_blk, p.S, p.d);
    bf16x8_t kf[NKS], vf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) {
        kf[ks] = frag_load(kg, p.ld, kj, ks * 32 + fg * 8, p.S, p.d);
        vf[ks] = frag_load(vg, p.ld, kj, ks * 32 + fg * 8, p.S, p.d);
    }
    float st_l = 1.f, st_d = 0.f;                       // this thread's slot of the next tile's row statistics
    if (tid < QT) {
        const int qi = qstart_blk + tid;
        st_l = qi < p.S ? p.lse[statbase + qi] : 1.f; st_d = qi < p.S ? p.delta[statbase + qi] : 0.f;
        if (STABLE) st_l = 1.0f / fmaxf(st_l, 1e-9f);
    }

    f32x4_t dka[NDV], dva[NDV];
#pragma unroll
    for (int n = 0; n < NDV; n++) { dka[n] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dva[n] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    typedef bf16x4_t __attribute__((address_space(3))) * lds4p;
    const int qq = fr >> 2, pp = fr & 3;

    for (int q0 = qstart_blk; q0 < p.S; q0 += QT) {
        tile_store<DP, QT>(rq, lq, PITCH);
        tile_store<DP, QT>(rg, ldo, PITCH);
        if (tid < QT) { lstat[tid] = st_l; lstat[QT + tid] = st_d; }
        __syncthreads();
        if (q0 + QT < p.S) {
            tile_load<DP, QT>(rq, qg, p.ld, q0 + QT, p.S, p.d);
            tile_load<DP, QT>(rg, dog, p.ld_o, q0 + QT, p.S, p.d);
            if (tid < QT) {
                const int qi = q0 + QT + tid;
                st_l = qi < p.S ? p.lse[statbase + qi] : 1.f; st_d = qi < p.S ? p.delta[statbase + qi] : 0.f;
                if (STABLE) st_l = 1.0f / fmaxf(st_l, 1e-9f);
            }
        }
        if (wave_live && q0 + QT > qstart_w) {
            f32x4_t sa[NT16], da[NT16];
#pragma unroll
            for (int t = 0; t < NT16; t++) {
                sa[t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; da[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NKS; ks++) {
                    const int off = ks * 32 + fg * 8;
                    const bf16x8_t qf = *reinterpret_cast<const bf16x8_t*>(&lq[(16 * t + fr) * PITCH + off]);
                    const bf16x8_t gf = *reinterpret_cast<const bf16x8_t*>(&ldo[(16 * t + fr) * PITCH + off]);
                    sa[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf[ks], sa[t], 0, 0, 0);
                    da[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, vf[ks], da[t], 0, 0, 0);
                }
            }
            const bool full = kw0 + 15 < p.S && q0 + QT <= p.S && ((kw0 + 15 <= q0) || (q0 + QT <= p.P && kw0 + 15 < p.P));
#pragma unroll
            for (int kk = 0; kk < QT / 32; kk++) {
                bf16x8_t pf, dsf;
                auto fill = [&](auto fc) {
                    constexpr bool FULL = decltype(fc)::value;
#pragma unroll
                    for (int tt = 0; tt < 2; tt++) {
                        const int t = 2 * kk + tt;
                        // row statistics of queries 16t+4fg .. +3: lstat holds 1/L (stablemax) or lse (softmax), then delta
                        const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(&lstat[16 * t + 4 * fg]);
                        const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(&lstat[QT + 16 * t + 4 * fg]);
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int qi = q0 + 16 * t + 4 * fg + r;
                            const float x = sa[t][r] * p.scale;
                            float pr = 0.f, ds = 0.f;
                            if (FULL || (qi < p.S && attn_allowed(qi, kj, p.P, p.S))) {
                                if (STABLE) {
                                    const float sx = stable_s_fast(x), invL = l4[r];
                                    pr = sx * invL; ds = (da[t][r] - d4[r]) * invL * (x < 0.f ? sx * sx : 1.f);
                                } else { pr = __expf(x - l4[r]); ds = pr * (da[t][r] - d4[r]); }
                            }
                            pf[4 * tt + r] = (short)f2bf(pr); dsf[4 * tt + r] = (short)f2bf(ds);
                        }
                    }
                };
                if (full) fill(std::true_type{}); else fill(std::false_type{});
                if (q0 + 32 * kk + 32 > qstart_w) {      // wave-uniform: earlier queries see none of this wave's keys
#pragma unroll
                    for (int n = 0; n < NDV; n++) {
                        const bf16x4_t glo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(&ldo[(32 * kk + 4 * fg + qq) * PITCH + 16 * n + 4 * pp]));
                        const bf16x4_t ghi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(&ldo[(32 * kk + 16 + 4 * fg + qq) * PITCH + 16 * n + 4 * pp]));
                        const bf16x8_t gt = bf16x8_t{glo[0], glo[1], glo[2], glo[3], ghi[0], ghi[1], ghi[2], ghi[3]};
                        dva[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt, pf, dva[n], 0, 0, 0);
                        const bf16x4_t qlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(&lq[(32 * kk + 4 * fg + qq) * PITCH + 16 * n + 4 * pp]));
                        const bf16x4_t qhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(&lq[(32 * kk + 16 + 4 * fg + qq) * PITCH + 16 * n + 4 * pp]));
                        const bf16x8_t qt = bf16x8_t{qlo[0], qlo[1], qlo[2], qlo[3], qhi[0], qhi[1], qhi[2], qhi[3]};
                        dka[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, dsf, dka[n], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }
    if (kj < p.S) {
        T* dk = (T*)p.dk + ((long)b * p.S + kj) * p.ld_d + (long)h * p.d;
        T* dv = (T*)p.dv + ((long)b * p.S + kj) * p.ld_d + (long)h * p.d;
#pragma unroll
        for (int n = 0; n < NDV; n++) {
            const int c = 16 * n + 4 * fg;
            if (c < p.d) {
                const float k4[4] = {dka[n][0] * p.scale, dka[n][1] * p.scale, dka[n][2] * p.scale, dka[n][3] * p.scale};
                const float v4[4] = {dva[n][0], dva[n][1], dva[n][2], dva[n][3]};
                VecIO<T, 4>::store(dk + c, k4);
                VecIO<T, 4>::store(dv + c, v4);
            }
        }
    }
}

template <int DP, int KT, int QT = KT>
static int bwd_launch5(const AttnP& p, int mode, hipStream_t st) {
    constexpr int PITCH = DP + FPAD;
    constexpr int RT = KT > QT ? KT : QT;
    const size_t lds = (size_t)2 * RT * PITCH * 2 + 2 * RT * sizeof(float);
    const dim3 grid((unsigned)cdiv(p.S, 64), (unsigned)(p.B * p.H));
#define SETATTR_(K_) if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)K_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
    if (mode == MM_ATTN_STABLEMAX) {
        SETATTR_((prefix_attn_bwd_dq5_k<DP, KT, true>));
        hipLaunchKernelGGL((prefix_attn_bwd_dq5_k<DP, KT, true>), grid, dim3(256), lds, st, p);
        SETATTR_((prefix_attn_bwd_dkv5_k<DP, QT, true>));
        hipLaunchKernelGGL((prefix_attn_bwd_dkv5_k<DP, QT, true>), grid, dim3(256), lds, st, p);
    } else {
        SETATTR_((prefix_attn_bwd_dq5_k<DP, KT, false>));
        hipLaunchKernelGGL((prefix_attn_bwd_dq5_k<DP, KT, false>), grid, dim3(256), lds, st, p);
        SETATTR_((prefix_attn_bwd_dkv5_k<DP, QT, false>));
        hipLaunchKernelGGL((prefix_attn_bwd_dkv5_k<DP, QT, false>), grid, dim3(256), lds, st, p);
    }
#undef SETATTR_
    MM_LAUNCH_CHECK();
    return MM_OK;
}
static int bwd_dispatch5(const AttnP& p, int mode, hipStream_t st) {
    const int d = p.d;
    if (d <= 32) return bwd_launch5<32, 64>(p, mode, st);
    if (d <= 64) return bwd_launch5<64, 64>(p, mode, st);
    // 32-row streamed tiles from d = 128 up: the dK/dV kernel drops below 256 registers (2 waves per SIMD instead of
    // 1: 1.29x on the whole backward), the dQ kernel reaches 3 waves
    static const bool t64 = getenv("MM_ATTN_TILE64") != nullptr;                 // A/B switch
    if (d <= 128) return t64 ? bwd_launch5<128, 64>(p, mode, st) : bwd_launch5<128, 32, 32>(p, mode, st);
    if (d <= 192) return t64 ? bwd_launch5<192, 64>(p, mode, st) : bwd_launch5<192, 32, 32>(p, mode, st);
    MM_FAIL(MM_ERR_UNSUPPORTED, "prefix_attn_bwd: head_dim %d > 512", d);
}

template <int DP>
static int bwd_launch4(const AttnP& p, int mode, hipStream_t st) {
    constexpr int PITCH = DP + 8;
    const size_t lds = (size_t)2 * 32 * PITCH * 2 + 2 * 32 * sizeof(float);
    const dim3 grid((unsigned)cdiv(p.S, 64), (unsigned)(p.B * p.H));
#define SETATTR_(K_) if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)K_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
    if (mode == MM_ATTN_STABLEMAX) {
        SETATTR_((prefix_attn_bwd_dq4_k<DP, true>));
        hipLaunchKernelGGL((prefix_attn_bwd_dq4_k<DP, true>), grid, dim3(256), lds, st, p);
        SETATTR_((prefix_attn_bwd_dkv4_k<DP, true>));
        hipLaunchKernelGGL((prefix_attn_bwd_dkv4_k<DP, true>), grid, dim3(256), lds, st, p);
    } else {
        SETATTR_((prefix_attn_bwd_dq4_k<DP, false>));
        hipLaunchKernelGGL((prefix_attn_bwd_dq4_k<DP, false>), grid, dim3(256), lds, st, p);
        SETATTR_((prefix_attn_bwd_dkv4_k<DP, false>));
        hipLaunchKernelGGL((prefix_attn_bwd_dkv4_k<DP, false>), grid, dim3(256), lds, st, p);
    }
#undef SETATTR_
    MM_LAUNCH_CHECK();
    return MM_OK;
}
static int bwd_dispatch4(const AttnP& p, int mode, hipStream_t st) {
    const int d = p.d;
    if (d <= 32) return bwd_launch4<32>(p, mode, st);
    if (d <= 64) return bwd_launch4<64>(p, mode, st);
    if (d <= 128) return bwd_launch4<128>(p, mode, st);
    if (d <= 192) return bwd_launch4<192>(p, mode, st);
    if (d <= 256) return bwd_launch4<256>(p, mode, st);
    if (d <= 512) return bwd_launch4<512>(p, mode, st);
    MM_FAIL(MM_ERR_UNSUPPORTED, "prefix_attn_bwd: head_dim %d > 512", d);
}

template <typename T, int DP>
static int bwd_launch(const AttnP& p, int mode, hipStream_t st) {
    constexpr int PITCH = DP + AT<T>::PAD, KT = AT<T>::KT;
    const size_t lds_q = (size_t)(32 + 2 * KT) * PITCH * sizeof(T);
    size_t lds_kv = (size_t)(32 + 2 * KT) * PITCH * sizeof(T) + 2 * KT * sizeof(float);
    // the f32 epilogue staging tile [16][DP+1] must fit in the streamed-tile region
    const size_t need = (size_t)16 * (DP + 1) * sizeof(float);
    static_assert((size_t)2 * AT<T>::KT * (DP + AT<T>::PAD) * sizeof(T) >= (size_t)16 * (DP + 1) * sizeof(float), "staging tile does not fit");
    (void)need;
    const dim3 gq((unsigned)cdiv(p.S, 16), (unsigned)(p.B * p.H));
#define SETATTR_(K_, BYTES_) if ((BYTES_) > 64 * 1024) (void)hipFuncSetAttribute((const void*)K_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES_))
    if (mode == MM_ATTN_STABLEMAX) {
        SETATTR_((prefix_attn_bwd_dq_k<T, DP, true>), lds_q);
        hipLaunchKernelGGL((prefix_attn_bwd_dq_k<T, DP, true>), gq, dim3(64), lds_q, st, p);
        SETATTR_((prefix_attn_bwd_dkv_k<T, DP, true>), lds_kv);
        hipLaunchKernelGGL((prefix_attn_bwd_dkv_k<T, DP, true>), gq, dim3(64), lds_kv, st, p);
    } else {
        SETATTR_((prefix_attn_bwd_dq_k<T, DP, false>), lds_q);
        hipLaunchKernelGGL((prefix_attn_bwd_dq_k<T, DP, false>), gq, dim3(64), lds_q, st, p);
        SETATTR_((prefix_attn_bwd_dkv_k<T, DP, false>), lds_kv);
        hipLaunchKernelGGL((prefix_attn_bwd_dkv_k<T, DP, false>), gq, dim3(64), lds_kv, st, p);
    }
#undef SETATTR_
    MM_LAUNCH_CHECK();
    return MM_OK;
}

template <typename T>
static int bwd_dispatch(const AttnP& p, int mode, hipStream_t st) {
    const int d = p.d;
    if (d <= 32) return bwd_launch<T, 32>(p, mode, st);
    if (d <= 64) return bwd_launch<T, 64>(p, mode, st);
    if (d <= 128) return bwd_launch<T, 128>(p, mode, st);
    if (d <= 192) return bwd_launch<T, 192>(p, mode, st);
    if (d <= 256) return bwd_launch<T, 256>(p, mode, st);
    if (d <= 512) return bwd_launch<T, 512>(p, mode, st);
    MM_FAIL(MM_ERR_UNSUPPORTED, "prefix_attn_bwd: head_dim %d > 512", d);
}

extern "C" size_t mm_prefix_attn_bwd_workspace(int64_t B, int64_t H, int64_t S) {
    return (B > 0 && H > 0 && S > 0) ? (size_t)(B * H * S) * sizeof(float) : 0;
}

extern "C" int mm_prefix_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                                  void* dq, void* dk, void* dv, const float* cos_t, const float* sin_t, void* workspace,
                                  size_t workspace_bytes, int64_t B, int64_t H, int64_t S, int64_t d, int64_t rot,
                                  int64_t prompt_len, int64_t ld_qkv, int64_t ld_dqkv, int mode, int dtype, void* stream) {
    int rc = attn_check("prefix_attn_bwd", q, k, v, cos_t, sin_t, B, H, S, d, rot, prompt_len, ld_qkv, mode, dtype);
    if (rc != MM_OK) return rc;
    MM_CHECK(o && d_o && lse && dq && dk && dv && workspace, "prefix_attn_bwd: null pointer");
    MM_CHECK(ld_dqkv >= H * d, "prefix_attn_bwd: ld_dqkv too small");
    MM_CHECK(workspace_bytes >= mm_prefix_attn_bwd_workspace(B, H, S), "prefix_attn_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    AttnP p{};
    p.q = q; p.k = k; p.v = v; p.o = (void*)o; p.lse = (float*)lse; p.dO = d_o; p.dq = dq; p.dk = dk; p.dv = dv;
    p.delta = (const float*)workspace; p.cos_t = cos_t; p.sin_t = sin_t;
    p.B = (int)B; p.H = (int)H; p.S = (int)S; p.d = (int)d; p.rot = (int)rot; p.P = (int)prompt_len;
    p.ld = ld_qkv; p.ld_o = H * d; p.ld_d = ld_dqkv; p.scale = (float)(1.0 / sqrt((double)d));
    p.vec_ok = attn_vec_ok(q, k, v, ld_qkv, d, dtype) && attn_vec_ok(d_o, d_o, d_o, H * d, d, dtype);
    static const bool force_generic = getenv("MM_ATTN_GENERIC") != nullptr;     // A/B switch for debugging
    const bool al8 = (((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) % 8) == 0 && ld_dqkv % 4 == 0;
    // measured: the fast form wins up to d = 192 (1.05-1.5x); beyond that its register footprint loses to the staged kernels
    static const bool force_v2 = getenv("MM_ATTN_V2") != nullptr;               // A/B switch: the previous fast generation
    static const bool no_res = getenv("MM_ATTN_NO_RES") != nullptr;             // A/B switch: tiled kernels instead of the resident ones
    const bool fast = !force_generic && dtype == MM_BF16 && rot == 0 && p.vec_ok && al8 && d <= 192;
    // whole head resident in LDS; its dQ kernel computes delta = rowsum(dO * O) itself (8-byte aligned o / dO rows)
    if (fast && !force_v2 && !no_res && attn_res_ok(p) && (((uintptr_t)o | (uintptr_t)d_o) % 8) == 0) return attn_res_bwd(p, mode, st);
    const long rows = B * S * H;
    hipLaunchKernelGGL(attn_delta_k, dim3((unsigned)cdiv(rows, 16)), dim3(256), 0, st, o, d_o, (float*)workspace, rows, (int)H, (int)S,
                       (int)d, (long)(H * d), (long)(H * d),
                       dtype == MM_BF16 ? ((d % 4 == 0 && (((uintptr_t)o | (uintptr_t)d_o) % 8) == 0) ? 2 : 1) : 0);
    MM_LAUNCH_CHECK();
    if (fast) {
        if (!force_v2 && attn_fast_ok(p)) return attn_fast_bwd(p, mode, st);
        return bwd_dispatch5(p, mode, st);
    }
    static const bool no_wide = getenv("MM_ATTN_NO_WIDE") != nullptr;           // A/B switch: head_dim 512 on the staged kernels
    if (!force_generic && !force_v2 && !no_wide && dtype == MM_BF16 && rot == 0 && p.vec_ok && al8 && attn_wide_ok(p) &&
        ((uintptr_t)d_o) % 16 == 0)
        return attn_wide_bwd(p, mode, st);                                      // head_dim 512: DMA-staged, one wave per SIMD
    return dtype == MM_BF16 ? bwd_dispatch4(p, mode, st) : bwd_dispatch<float>(p, mode, st);
}

// ------------------------------------------------------------------------------------------------
// K4: decode step against the KV cache.  One wave per (b,h).  16 lanes cover one cache row (coalesced
// 16-byte loads), 4 keys per wave instruction; scores go through LDS, so K and V are each read once.
// HBM-bound: algorithmic bytes = 2 * (pos+1) * d * sizeof(T) per (b,h).
// ------------------------------------------------------------------------------------------------
struct DecP {
    const void* q; const void* kn; const void* vn; void* kc; void* vc; void* o;
    const float* cos_t; const float* sin_t;
    int B, H, d, rot, pos, Smax; long ld; float scale; int mode;
    const int* pos_dev;          // if set: the position is read from device memory (captured HIP graphs), clamped to Smax-1
};

template <typename T, int VEC, int CH>
__global__ __launch_bounds__(256) void decode_attn_k(const DecP p) {
    // LDS sized by the launch for THIS cache length and head_dim ([4 waves][Smax scores | d | d floats]): with the static
    // worst case (1024 scores, d = 512: 32 KiB) five workgroups held 160 KiB and nothing else could share the CU
    extern __shared__ __attribute__((aligned(16))) float k4_lds[];
    const int dpad = (p.d + 3) & ~3;
    float* sc_w = k4_lds + (threadIdx.x >> 6) * (p.Smax + 2 * dpad);
    float* rq_w = sc_w + p.Smax;
    float* rk_w = rq_w + dpad;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    (void)wave;
    const int li = lane & 15, grp = lane >> 4;
    const int bh = blockIdx.x * 4 + wave;
    if (bh >= p.B * p.H) return;                       // wave-uniform
    const int b = bh / p.H, h = bh % p.H;
    const int d = p.d, half = p.rot >> 1, pos = p.pos_dev ? min(max(*p.pos_dev, 0), p.Smax - 1) : p.pos;
    const T* qg = (const T*)p.q + (long)b * p.ld + (long)h * d;
    const T* kng = (const T*)p.kn + (long)b * p.ld + (long)h * d;
    const T* vng = (const T*)p.vn + (long)b * p.ld + (long)h * d;
    T* kc = (T*)p.kc + (long)bh * p.Smax * d;
    T* vc = (T*)p.vc + (long)bh * p.Smax * d;
    // ---- rotate q and k_new at position pos (through LDS: the partner element lives in another lane)
    for (int c = lane; c < d; c += 64) { rq_w[c] = to_f(qg[c]); rk_w[c] = to_f(kng[c]); }
    __builtin_amdgcn_s_waitcnt(0xC07F);                // lgkmcnt(0): own LDS writes landed (single wave owns its rows)
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < half; i += 64) {
        const float c = p.cos_t[(long)pos * half + i], s = p.sin_t[(long)pos * half + i];
        const float a = rq_w[i], bb = rq_w[i + half];
        rq_w[i] = a * c - bb * s; rq_w[i + half] = bb * c + a * s;
        const float a2 = rk_w[i], b2 = rk_w[i + half];
        rk_w[i] = a2 * c - b2 * s; rk_w[i + half] = b2 * c + a2 * s;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    // registers: this lane's slice of q (rounded to T like the prefill kernel does), k_new, v_new
    float qv[CH][VEC], knv[CH][VEC], vnv[CH][VEC];
    bool cok[CH];
#pragma unroll
    for (int k = 0; k < CH; k++) {
        const int c = (k * 16 + li) * VEC;
        cok[k] = c < d;
#pragma unroll
        for (int v = 0; v < VEC; v++) {
            const bool ok = c + v < d;
            qv[k][v] = ok ? to_f(from_f<T>(rq_w[c + v])) : 0.f;
            knv[k][v] = ok ? to_f(from_f<T>(rk_w[c + v])) : 0.f;
            vnv[k][v] = ok ? to_f(vng[c + v]) : 0.f;
        }
    }
    // append to the cache (group 0 writes)
    if (grp == 0) {
#pragma unroll
        for (int k = 0; k < CH; k++)
            if (cok[k]) {
                const int c = (k * 16 + li) * VEC;
                VecIO<T, VEC>::store(kc + (long)pos * d + c, knv[k]);
                VecIO<T, VEC>::store(vc + (long)pos * d + c, vnv[k]);
            }
    }
    // ---- pass 1: scores of cached keys 0..pos-1 (4 per iteration), the new key from registers.  The cache rows of PF
    // iterations are requested ahead and stay packed in registers (a wave has only its own stream to hide HBM latency
    // with: one 1-KiB request in flight per wave is ~5 MB on the chip, Little's law at ~1 us gives the 4.8 TB/s this
    // kernel used to reach).  Rows past the end are clamped to a valid row and masked, so no branch guards a load.
    typedef typename VecIO<T, VEC>::raw_t raw_t;
    constexpr int PF = CH >= 4 ? 2 : (CH == 1 ? 8 : 4);              // wide heads: fewer rows ahead, the row itself is many registers
    const int nit = (pos + 3) >> 2;
    auto issue = [&](const T* base, int it, raw_t (&dst)[CH]) {
        const int j = min(it * 4 + grp, pos - 1);
#pragma unroll
        for (int k = 0; k < CH; k++)
            if (cok[k]) dst[k] = VecIO<T, VEC>::load_raw(base + (long)j * d + (k * 16 + li) * VEC);
    };
    raw_t buf[PF][CH];
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < PF; u++)
        if (u < nit) issue(kc, u, buf[u]);
    for (int it0 = 0; it0 < nit; it0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int it = it0 + u;
            if (it < nit) {                                  // wave-uniform
                const int j = it * 4 + grp;
                float dot = 0.f;
#pragma unroll
                for (int k = 0; k < CH; k++)
                    if (cok[k]) {
                        float kv[VEC];
                        VecIO<T, VEC>::unpack(buf[u][k], kv);
#pragma unroll
                        for (int v = 0; v < VEC; v++) dot = fmaf(qv[k][v], kv[v], dot);
                    }
                if (it + PF < nit) issue(kc, it + PF, buf[u]);
                dot = group_sum<16>(dot) * p.scale;
                if (j < pos) { if (li == 0) sc_w[j] = dot; mx = fmaxf(mx, dot); }
            }
        }
    }
    float dn = 0.f;
#pragma unroll
    for (int k = 0; k < CH; k++)
#pragma unroll
        for (int v = 0; v < VEC; v++) dn = fmaf(qv[k][v], knv[k][v], dn);
    dn = group_sum<16>(dn) * p.scale;                   // every group computes the same value
    mx = fmaxf(mx, dn);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    // ---- pass 2: weights and P.V
    const bool stable = p.mode == MM_ATTN_STABLEMAX;
    float oacc[CH][VEC];
    float lsum = 0.f;
#pragma unroll
    for (int k = 0; k < CH; k++)
#pragma unroll
        for (int v = 0; v < VEC; v++) oacc[k][v] = 0.f;
#pragma unroll
    for (int u = 0; u < PF; u++)
        if (u < nit) issue(vc, u, buf[u]);
    for (int it0 = 0; it0 < nit; it0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int it = it0 + u;
            if (it < nit) {                                  // wave-uniform
                const int j = it * 4 + grp;
                const bool ok = j < pos;
                const float x = sc_w[ok ? j : 0];
                const float w = ok ? (stable ? stable_s_f(x) : __expf(x - mx)) : 0.f;
                lsum += w;
#pragma unroll
                for (int k = 0; k < CH; k++)
                    if (cok[k]) {
                        float vv[VEC];
                        VecIO<T, VEC>::unpack(buf[u][k], vv);
#pragma unroll
                        for (int v = 0; v < VEC; v++) oacc[k][v] = fmaf(w, vv[v], oacc[k][v]);
                    }
                if (it + PF < nit) issue(vc, it + PF, buf[u]);
            }
        }
    }
    if (grp == 0) {                                      // the new token's own key/value
        const float w = stable ? stable_s_f(dn) : __expf(dn - mx);
        lsum += w;
#pragma unroll
        for (int k = 0; k < CH; k++)
#pragma unroll
            for (int v = 0; v < VEC; v++) oacc[k][v] = fmaf(w, vnv[k][v], oacc[k][v]);
    }
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
    const float inv = 1.0f / (stable ? fmaxf(lsum, 1e-9f) : lsum);
    T* og = (T*)p.o + (long)b * p.H * d + (long)h * d;
#pragma unroll
    for (int k = 0; k < CH; k++) {
        float r[VEC];
#pragma unroll
        for (int v = 0; v < VEC; v++) {
            float t = oacc[k][v];
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            r[v] = t * inv;
        }
        if (grp == 0 && cok[k]) VecIO<T, VEC>::store(og + (k * 16 + li) * VEC, r);
    }
}

// K4, one pass (bf16): the K row and the V row of a position are in flight together and every 16-lane group keeps its
// own running softmax (max, sum, output), merged across the four groups at the end; stablemax needs no max at all.
// No score round trip through LDS and no K-pass -> V-pass boundary: with all B*H waves resident and of equal length,
// that boundary (and the start-up) was chip-wide idle HBM time.  The walk is branch-free: every iteration issues its
// loads (rows past the end clamp to the last cached row and weigh 0; columns past d clamp to column 0 against a zero q),
// so the compiler counts the loads in flight exactly (behind `if (more rows)` it fell back to vmcnt(0) per iteration);
// the cache base is wave-uniform (scalar registers) and a row's offset a 32-bit lane value.
// benchmarks/k4_pattern.hip: the cache walk alone reads at 6.0-6.1 TB/s; the two-pass kernel reached 4.8-5.0.
template <typename T, int VEC, int CH, bool STABLE>
__global__ __launch_bounds__(256) void decode_attn1_k(const DecP p) {
    extern __shared__ __attribute__((aligned(16))) float k4_lds[];          // [4 waves][d | d floats]: the rotation's partner exchange
    const int dpad = (p.d + 3) & ~3;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* rq_w = k4_lds + wave * (2 * dpad);
    float* rk_w = rq_w + dpad;
    const int li = lane & 15, grp = lane >> 4;
    const int bh = blockIdx.x * 4 + wave;
    if (bh >= p.B * p.H) return;                       // wave-uniform
    const int b = bh / p.H, h = bh % p.H;
    const int d = p.d, half = p.rot >> 1, pos = p.pos_dev ? min(max(*p.pos_dev, 0), p.Smax - 1) : p.pos;
    const T* qg = (const T*)p.q + (long)b * p.ld + (long)h * d;
    const T* kng = (const T*)p.kn + (long)b * p.ld + (long)h * d;
    const T* vng = (const T*)p.vn + (long)b * p.ld + (long)h * d;
    T* kc = (T*)p.kc + (long)bh * p.Smax * d;
    T* vc = (T*)p.vc + (long)bh * p.Smax * d;
    typedef typename VecIO<T, VEC>::raw_t raw_t;
#ifndef K4_PF
#define K4_PF 4
#endif
    constexpr int PF = CH >= 4 ? 1 : (CH == 2 ? 2 : K4_PF);           // positions ahead (a K row and a V row each)
    constexpr int DMAX = CH * VEC * 16, NQ = (DMAX + 63) / 64, NH = (DMAX / 2 + 63) / 64;      // the launch picks CH with d <= DMAX
    static_assert(DMAX <= 512, "head_dim <= 512");
    bool cok[CH];
    unsigned colb[CH];                                                 // byte offset of this lane's chunk k inside a row (0 if past d)
#pragma unroll
    for (int k = 0; k < CH; k++) { const int c = (k * 16 + li) * VEC; cok[k] = c < d; colb[k] = cok[k] ? c * (unsigned)sizeof(T) : 0u; }
    // ---- the token's own vectors first (the waits are in issue order), then the first cache rows, which depend on nothing
    float qraw[NQ], kraw[NQ];                                          // lane + 64 * i < d
#pragma unroll
    for (int i = 0; i < NQ; i++) { const int c = lane + 64 * i; if (c < d) { qraw[i] = to_f(qg[c]); kraw[i] = to_f(kng[c]); } }
    raw_t vnr[CH];
#pragma unroll
    for (int k = 0; k < CH; k++) vnr[k] = VecIO<T, VEC>::load_raw((const T*)((const char*)vng + colb[k]));
    float cr[NH], sr[NH];                                              // lane + 64 * i < half
#pragma unroll
    for (int i = 0; i < NH; i++) { const int c = lane + 64 * i; if (c < half) { cr[i] = p.cos_t[(long)pos * half + c]; sr[i] = p.sin_t[(long)pos * half + c]; } }
    const int nit = (pos + 3) >> 2;
    const unsigned rowb = (unsigned)d * (unsigned)sizeof(T);          // Smax * d * sizeof(T) <= 1024 * 512 * 4: 32-bit offsets
    auto issue = [&](int it, raw_t (&dk)[CH], raw_t (&dv)[CH]) {
        const unsigned r = (unsigned)min(it * 4 + grp, pos - 1) * rowb;
#pragma unroll
        for (int k = 0; k < CH; k++) {
            // non-temporal: the rows are read once per step; as ordinary loads they also evicted the step's weights from the
            // last-level cache (K4 alone 177 -> 164 us, the whole decode step 0.97 -> 0.88 ms)
            dk[k] = __builtin_nontemporal_load((const raw_t*)((const char*)kc + (r + colb[k])));
            dv[k] = __builtin_nontemporal_load((const raw_t*)((const char*)vc + (r + colb[k])));
        }
    };
    raw_t bk[PF][CH], bv[PF][CH];
    if (nit > 0) {                                       // pos = 0: no cached row exists to clamp to
#pragma unroll
        for (int u = 0; u < PF; u++) issue(u, bk[u], bv[u]);
    }
    // ---- rotate q and k_new at position pos (through LDS: the partner element lives in another lane)
#pragma unroll
    for (int i = 0; i < NQ; i++) { const int c = lane + 64 * i; if (c < d) { rq_w[c] = qraw[i]; rk_w[c] = kraw[i]; } }
    __builtin_amdgcn_s_waitcnt(0xC07F);                // lgkmcnt(0): own LDS writes landed (single wave owns its rows)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < NH; i++) {
        const int c = lane + 64 * i;
        if (c < half) {
            const float a = rq_w[c], bb = rq_w[c + half];
            rq_w[c] = a * cr[i] - bb * sr[i]; rq_w[c + half] = bb * cr[i] + a * sr[i];
            const float a2 = rk_w[c], b2 = rk_w[c + half];
            rk_w[c] = a2 * cr[i] - b2 * sr[i]; rk_w[c + half] = b2 * cr[i] + a2 * sr[i];
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    // registers: this lane's slice of q (rounded to T like the prefill kernel does; 0 past d), k_new, v_new
    float qv[CH][VEC], knv[CH][VEC], vnv[CH][VEC];
#pragma unroll
    for (int k = 0; k < CH; k++) {
        const int c = cok[k] ? (k * 16 + li) * VEC : 0;                // d % VEC == 0 (launch): a chunk is inside the row or past it
        VecIO<T, VEC>::unpack(vnr[k], vnv[k]);
#pragma unroll
        for (int v = 0; v < VEC; v += 4) {
            const mm_f4_t tq = *reinterpret_cast<const mm_f4_t*>(rq_w + c + v), tk = *reinterpret_cast<const mm_f4_t*>(rk_w + c + v);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                qv[k][v + e] = cok[k] ? to_f(from_f<T>(tq[e])) : 0.f;
                knv[k][v + e] = to_f(from_f<T>(tk[e]));
            }
        }
    }
    if (grp == 0) {                                      // append to the cache
#pragma unroll
        for (int k = 0; k < CH; k++)
            if (cok[k]) {
                const int c = (k * 16 + li) * VEC;
                VecIO<T, VEC>::store(kc + (long)pos * d + c, knv[k]);
                VecIO<T, VEC>::store(vc + (long)pos * d + c, vnv[k]);
            }
    }
    // ---- the walk.  Softmax in base 2: x2 = q.k * scale * log2(e), weights exp2(x2 - running max).
    const float sc = STABLE ? p.scale : p.scale * 1.44269504088896f;
    float oacc[CH][VEC];
    float lsum = 0.f, m = -1e30f;
#pragma unroll
    for (int k = 0; k < CH; k++)
#pragma unroll
        for (int v = 0; v < VEC; v++) oacc[k][v] = 0.f;
    auto take = [&](bool ok, float x, const float (&vv)[CH][VEC]) {
        float w;
        if (STABLE) {
            w = ok ? stable_s_fast(x) : 0.f;
        } else {
            const float mn = ok ? fmaxf(m, x) : m;
            const float al = __builtin_amdgcn_exp2f(m - mn);
            w = ok ? __builtin_amdgcn_exp2f(x - mn) : 0.f;
            m = mn; lsum *= al;
#pragma unroll
            for (int k = 0; k < CH; k++)
#pragma unroll
                for (int v = 0; v < VEC; v++) oacc[k][v] *= al;
        }
        lsum += w;
#pragma unroll
        for (int k = 0; k < CH; k++)
#pragma unroll
            for (int v = 0; v < VEC; v++) oacc[k][v] = fmaf(w, vv[k][v], oacc[k][v]);
    };
    {                                                    // the new token's own key/value: one more position of group 0, taken first (its registers die here)
        float dn = 0.f;
#pragma unroll
        for (int k = 0; k < CH; k++)
#pragma unroll
            for (int v = 0; v < VEC; v++) dn = fmaf(qv[k][v], knv[k][v], dn);
        take(grp == 0, group_sum<16>(dn) * sc, vnv);
    }
    for (int it0 = 0; it0 < nit; it0 += PF) {            // iterations past nit (the last round) re-read the last row and weigh 0
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int it = it0 + u;
            float kv[CH][VEC], vv[CH][VEC];
#pragma unroll
            for (int k = 0; k < CH; k++) { VecIO<T, VEC>::unpack(bk[u][k], kv[k]); VecIO<T, VEC>::unpack(bv[u][k], vv[k]); }
            issue(it + PF, bk[u], bv[u]);
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k < CH; k++)
#pragma unroll
                for (int v = 0; v < VEC; v++) dot = fmaf(qv[k][v], kv[k][v], dot);
            take(it * 4 + grp < pos, group_sum<16>(dot) * sc, vv);
        }
    }
    float f = 1.f;
    if (!STABLE) {
        float M = fmaxf(m, __shfl_xor(m, 16));
        M = fmaxf(M, __shfl_xor(M, 32));
        f = __builtin_amdgcn_exp2f(m - M);               // a group without positions: m = -1e30, f = 0
    }
    lsum *= f;
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
    const float inv = 1.0f / (STABLE ? fmaxf(lsum, 1e-9f) : lsum);
    T* og = (T*)p.o + (long)b * p.H * d + (long)h * d;
#pragma unroll
    for (int k = 0; k < CH; k++) {
        float r[VEC];
#pragma unroll
        for (int v = 0; v < VEC; v++) {
            float t = oacc[k][v] * f;
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            r[v] = t * inv;
        }
        if (grp == 0 && cok[k]) VecIO<T, VEC>::store(og + (k * 16 + li) * VEC, r);
    }
}

template <typename T>
__global__ void kv_cache_fill_k(const T* __restrict__ k, const T* __restrict__ v, T* __restrict__ kc, T* __restrict__ vc,
                                const float* __restrict__ cos_t, const float* __restrict__ sin_t, int B, int H, int S, int d,
                                int rot, int Smax, long ld) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)B * S * H * d;
    if (idx >= total) return;
    const int c = (int)(idx % d);
    const int h = (int)((idx / d) % H);
    const int s = (int)((idx / ((long)d * H)) % S);
    const int b = (int)(idx / ((long)d * H * S));
    const long src = ((long)b * S + s) * ld + (long)h * d;
    const long dst = (((long)b * H + h) * Smax + s) * d + c;
    const int half = rot >> 1;
    float kv = to_f(k[src + c]);
    if (c < rot) {
        const int i = c < half ? c : c - half;
        const float cs = cos_t[(long)s * half + i], sn = sin_t[(long)s * half + i];
        const float other = to_f(k[src + (c < half ? c + half : c - half)]);
        kv = c < half ? kv * cs - other * sn : kv * cs + other * sn;
    }
    kc[dst] = from_f<T>(kv);
    vc[dst] = v[src + c];
}

// rot == 0 (keys already rotated by mm_rope_inplace), d % 8 == 0, 16-byte aligned rows: a pure layout change
// [B][S][H][d] -> [B][H][Smax][d], 16 bytes per lane both ways (the element-wise kernel above moved 0.8 TB/s)
__global__ __launch_bounds__(256) void kv_cache_fill_v_k(const bf16* __restrict__ k, const bf16* __restrict__ v, bf16* __restrict__ kc,
                                                          bf16* __restrict__ vc, int H, int S, int d8, int Smax, long ld, long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;          // over [B][S][H][d/8]
    if (idx >= total) return;
    const int c = (int)(idx % d8);
    const long t = idx / d8;
    const int h = (int)(t % H);
    const long bs = t / H;
    const int s = (int)(bs % S);
    const long b = bs / S;
    const long src = bs * ld + (long)h * d8 * 8 + c * 8;
    const long dst = ((b * H + h) * Smax + s) * (long)d8 * 8 + c * 8;
    *reinterpret_cast<mm_u4_t*>(kc + dst) = *reinterpret_cast<const mm_u4_t*>(k + src);
    *reinterpret_cast<mm_u4_t*>(vc + dst) = *reinterpret_cast<const mm_u4_t*>(v + src);
}

extern "C" int mm_kv_cache_fill(const void* k, const void* v, void* kcache, void* vcache, const float* cos_t,
                                const float* sin_t, int64_t B, int64_t H, int64_t S, int64_t d, int64_t rot, int64_t Smax,
                                int64_t ld_qkv, int dtype, void* stream) {
    MM_CHECK(k && v && kcache && vcache, "kv_cache_fill: null pointer");
    MM_CHECK(B > 0 && H > 0 && S > 0 && d > 0 && S <= Smax, "kv_cache_fill: bad shape (S=%lld Smax=%lld)", (long long)S, (long long)Smax);
    MM_CHECK(rot >= 0 && rot <= d && rot % 2 == 0 && (rot == 0 || (cos_t && sin_t)), "kv_cache_fill: bad rot/tables");
    MM_CHECK(ld_qkv >= H * d, "kv_cache_fill: ld too small");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MM_BF16 && rot == 0 && d % 8 == 0 && ld_qkv % 8 == 0 &&
        (((uintptr_t)k | (uintptr_t)v | (uintptr_t)kcache | (uintptr_t)vcache) % 16) == 0) {
        const long tot8 = B * S * H * (d / 8);
        hipLaunchKernelGGL(kv_cache_fill_v_k, dim3((unsigned)cdiv(tot8, 256)), dim3(256), 0, st, (const bf16*)k, (const bf16*)v,
                           (bf16*)kcache, (bf16*)vcache, (int)H, (int)S, (int)(d / 8), (int)Smax, (long)ld_qkv, tot8);
        MM_LAUNCH_CHECK();
        return MM_OK;
    }
    const long total = B * S * H * d;
    const dim3 grid((unsigned)cdiv(total, 256));
    if (dtype == MM_BF16) hipLaunchKernelGGL(kv_cache_fill_k<bf16>, grid, dim3(256), 0, st, (const bf16*)k, (const bf16*)v, (bf16*)kcache, (bf16*)vcache, cos_t, sin_t, (int)B, (int)H, (int)S, (int)d, (int)rot, (int)Smax, (long)ld_qkv);
    else if (dtype == MM_F32) hipLaunchKernelGGL(kv_cache_fill_k<float>, grid, dim3(256), 0, st, (const float*)k, (const float*)v, (float*)kcache, (float*)vcache, cos_t, sin_t, (int)B, (int)H, (int)S, (int)d, (int)rot, (int)Smax, (long)ld_qkv);
    else MM_FAIL(MM_ERR_ARG, "kv_cache_fill: bad dtype %d", dtype);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

static int decode_launch(const void* q, const void* k_new, const void* v_new, void* kcache, void* vcache, void* o,
                         const float* cos_t, const float* sin_t, int64_t B, int64_t H, int64_t d, int64_t rot,
                         int64_t pos, const int32_t* pos_dev, int64_t Smax, int64_t ld_qkv, int mode, int dtype, void* stream) {
    MM_CHECK(q && k_new && v_new && kcache && vcache && o, "decode_attn: null pointer");
    MM_CHECK(B > 0 && H > 0 && d > 0 && d <= 512, "decode_attn: bad shape (d=%lld, max 512)", (long long)d);
    MM_CHECK(pos >= 0 && pos < Smax && Smax <= 1024, "decode_attn: pos=%lld Smax=%lld (Smax <= 1024)", (long long)pos, (long long)Smax);
    MM_CHECK(rot >= 0 && rot <= d && rot % 2 == 0 && (rot == 0 || (cos_t && sin_t)), "decode_attn: bad rot/tables");
    MM_CHECK(ld_qkv >= H * d, "decode_attn: ld too small");
    MM_CHECK(mode == MM_ATTN_SOFTMAX || mode == MM_ATTN_STABLEMAX, "decode_attn: bad mode %d", mode);
    DecP p{};
    p.q = q; p.kn = k_new; p.vn = v_new; p.kc = kcache; p.vc = vcache; p.o = o; p.cos_t = cos_t; p.sin_t = sin_t;
    p.B = (int)B; p.H = (int)H; p.d = (int)d; p.rot = (int)rot; p.pos = (int)pos; p.Smax = (int)Smax; p.ld = ld_qkv;
    p.scale = (float)(1.0 / sqrt((double)d)); p.mode = mode; p.pos_dev = (const int*)pos_dev;
    const dim3 grid((unsigned)cdiv(B * H, 4));
    hipStream_t st = (hipStream_t)stream;
    // vector width: the cache rows are d elements apart, so VEC must divide d
    const size_t lds = (size_t)4 * (Smax + 2 * ((d + 3) & ~3)) * sizeof(float);      // <= 4 * (1024 + 1024) * 4 = 32 KiB
#define L_(T_, VEC_, CH_) hipLaunchKernelGGL((decode_attn_k<T_, VEC_, CH_>), grid, dim3(256), lds, st, p)
    // bf16: the one-pass kernel (LDS only for the rotation); f32 keeps the two-pass kernel and its exact op order
    static const bool two_pass = getenv("MM_K4_TWO_PASS") != nullptr;          // A/B switch
    const size_t lds1 = (size_t)4 * 2 * ((d + 3) & ~3) * sizeof(float);
#define L1_(VEC_, CH_) do { if (mode == MM_ATTN_STABLEMAX) hipLaunchKernelGGL((decode_attn1_k<bf16, VEC_, CH_, true>), grid, dim3(256), lds1, st, p); \
                            else hipLaunchKernelGGL((decode_attn1_k<bf16, VEC_, CH_, false>), grid, dim3(256), lds1, st, p); } while (0)
    if (dtype == MM_BF16 && !two_pass) {
        const bool a16 = ((uintptr_t)kcache % 16 == 0) && ((uintptr_t)vcache % 16 == 0) && ((uintptr_t)o % 16 == 0);
        if (d % 8 == 0 && a16) { if (d <= 128) L1_(8, 1); else if (d <= 256) L1_(8, 2); else L1_(8, 4); }
        else if (d % 4 == 0) { if (d <= 64) L1_(4, 1); else if (d <= 128) L1_(4, 2); else if (d <= 256) L1_(4, 4); else L1_(4, 8); }
        else MM_FAIL(MM_ERR_UNSUPPORTED, "decode_attn: head_dim %lld must be a multiple of 4", (long long)d);
    } else if (dtype == MM_BF16) {
        const bool a16 = ((uintptr_t)kcache % 16 == 0) && ((uintptr_t)vcache % 16 == 0) && ((uintptr_t)o % 16 == 0);
        if (d % 8 == 0 && a16) { if (d <= 128) L_(bf16, 8, 1); else if (d <= 256) L_(bf16, 8, 2); else L_(bf16, 8, 4); }
        else if (d % 4 == 0) { if (d <= 64) L_(bf16, 4, 1); else if (d <= 128) L_(bf16, 4, 2); else if (d <= 256) L_(bf16, 4, 4); else L_(bf16, 4, 8); }
        else MM_FAIL(MM_ERR_UNSUPPORTED, "decode_attn: head_dim %lld must be a multiple of 4", (long long)d);
    } else if (dtype == MM_F32) {
        if (d % 4 == 0) { if (d <= 64) L_(float, 4, 1); else if (d <= 128) L_(float, 4, 2); else if (d <= 256) L_(float, 4, 4); else L_(float, 4, 8); }
        else MM_FAIL(MM_ERR_UNSUPPORTED, "decode_attn: head_dim %lld must be a multiple of 4", (long long)d);
    } else MM_FAIL(MM_ERR_ARG, "decode_attn: bad dtype %d", dtype);
#undef L_
#undef L1_
    MM_LAUNCH_CHECK();
    return MM_OK;
}

extern "C" int mm_decode_attn(const void* q, const void* k_new, const void* v_new, void* kcache, void* vcache, void* o,
                              const float* cos_t, const float* sin_t, int64_t B, int64_t H, int64_t d, int64_t rot,
                              int64_t pos, int64_t Smax, int64_t ld_qkv, int mode, int dtype, void* stream) {
    return decode_launch(q, k_new, v_new, kcache, vcache, o, cos_t, sin_t, B, H, d, rot, pos, nullptr, Smax, ld_qkv, mode, dtype, stream);
}
// The same step with the position in device memory (*pos_dev, clamped to [0, Smax-1]): one captured HIP graph of a decode
// step can be replayed for every position while a device-side counter advances.
extern "C" int mm_decode_attn_dev(const void* q, const void* k_new, const void* v_new, void* kcache, void* vcache, void* o,
                                  const float* cos_t, const float* sin_t, int64_t B, int64_t H, int64_t d, int64_t rot,
                                  const int32_t* pos_dev, int64_t Smax, int64_t ld_qkv, int mode, int dtype, void* stream) {
    MM_CHECK(pos_dev, "decode_attn_dev: null position pointer");
    return decode_launch(q, k_new, v_new, kcache, vcache, o, cos_t, sin_t, B, H, d, rot, 0, pos_dev, Smax, ld_qkv, mode, dtype, stream);
}

// ------------------------------------------------------------------------------------------------
// RoPE pre-pass: rotate (or un-rotate, inverse=1) the first `rot` dims of every head of x [B][S][H][d] in place.
// Lets the attention kernels run with rot = 0: q,k are rotated ONCE per layer instead of once per tile,
// and the backward un-rotates dq,dk once.  HBM traffic: 2 * B*S*H*rot elements.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void rope_inplace_k(T* __restrict__ x, const float* __restrict__ cos_t, const float* __restrict__ sin_t, long total,
                               int S, int H, int d, int half, long ld, int inverse) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int i = (int)(idx % half);
    const int h = (int)((idx / half) % H);
    const long bs = idx / ((long)half * H);
    const int s = (int)(bs % S);
    T* row = x + bs * ld + (long)h * d;
    const float c = cos_t[(long)s * half + i];
    float sn = sin_t[(long)s * half + i];
    if (inverse) sn = -sn;
    const float a = to_f(row[i]), b = to_f(row[i + half]);
    row[i] = from_f<T>(a * c - b * sn);
    row[i + half] = from_f<T>(b * c + a * sn);
}
// The same rotation with the row on blockIdx.x: row base, position and table row are block-uniform (scalar registers), a
// thread's (head, pair) comes from one 32-bit division.  The flat kernel above spends three 64-bit divisions per pair on
// its index (36.8 us for 64 MB of traffic at the config shape).
template <typename T>
__global__ __launch_bounds__(256) void rope_inplace_row_k(T* __restrict__ x, const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                         int S, int per_row, int d, int half, long ld, int inverse) {
    const int t = blockIdx.y * 256 + threadIdx.x;                  // over [H][half] of this row
    if (t >= per_row) return;
    const long bs = blockIdx.x;
    const int s = (int)(bs % S);
    const int h = t / half, i = t - h * half;
    T* row = x + bs * ld + (long)h * d;
    const float c = cos_t[(long)s * half + i];
    float sn = sin_t[(long)s * half + i];
    if (inverse) sn = -sn;
    const float a = to_f(row[i]), b = to_f(row[i + half]);
    row[i] = from_f<T>(a * c - b * sn);
    row[i + half] = from_f<T>(b * c + a * sn);
}

extern "C" int mm_rope_inplace(void* x, const float* cos_t, const float* sin_t, int64_t B, int64_t S, int64_t H, int64_t d,
                               int64_t rot, int64_t ld, int inverse, int dtype, void* stream) {
    MM_CHECK(x && cos_t && sin_t, "rope_inplace: null pointer");
    MM_CHECK(B > 0 && S > 0 && H > 0 && d > 0 && rot > 0 && rot <= d && rot % 2 == 0 && ld >= H * d, "rope_inplace: bad shape");
    const int half = (int)(rot / 2);
    const long total = B * S * H * half;
    hipStream_t st = (hipStream_t)stream;
    static const bool flat = getenv("MM_ROPE_FLAT") != nullptr;               // A/B switch: the flat-index kernel
    if (!flat && B * S < (1L << 31) && H * half <= 65535L * 256 && (dtype == MM_BF16 || dtype == MM_F32)) {
        const dim3 grid2((unsigned)(B * S), (unsigned)cdiv(H * half, 256));
        if (dtype == MM_BF16) hipLaunchKernelGGL(rope_inplace_row_k<bf16>, grid2, dim3(256), 0, st, (bf16*)x, cos_t, sin_t, (int)S, (int)(H * half), (int)d, half, (long)ld, inverse);
        else hipLaunchKernelGGL(rope_inplace_row_k<float>, grid2, dim3(256), 0, st, (float*)x, cos_t, sin_t, (int)S, (int)(H * half), (int)d, half, (long)ld, inverse);
        MM_LAUNCH_CHECK();
        return MM_OK;
    }
    const dim3 grid((unsigned)cdiv(total, 256));
    if (dtype == MM_BF16) hipLaunchKernelGGL(rope_inplace_k<bf16>, grid, dim3(256), 0, st, (bf16*)x, cos_t, sin_t, total, (int)S, (int)H, (int)d, half, (long)ld, inverse);
    else if (dtype == MM_F32) hipLaunchKernelGGL(rope_inplace_k<float>, grid, dim3(256), 0, st, (float*)x, cos_t, sin_t, total, (int)S, (int)H, (int)d, half, (long)ld, inverse);
    else MM_FAIL(MM_ERR_ARG, "rope_inplace: bad dtype %d", dtype);
    MM_LAUNCH_CHECK();
    return MM_OK;
}
