// Shared by the GEMM kernels: launch parameters and the fused epilogue (bias, pre-activation save, activation or
// activation-derivative multiply, LayerScale gamma, DropPath row scale, residual), 4 consecutive columns per call.
#pragma once
#include "common.h"

struct GemmP {
    const void* A; const void* B; void* out;
    int M, N, K;
    long lda, ldb, ldo;
    const float* bias; void* pre; const void* mul; int act;
    const float* gamma; const float* rowscale; int rows_per_scale;
    const void* res;
    int a_vec, b_vec, o_vec;       // 16-byte path legal for A / B loads, out-family accesses
    int k_per_split;               // K range per blockIdx.z
    float* partial;                // split-K slabs [z][M][N] (f32) or NULL
    int tiles_m, tiles_n;
    int stagger;                   // large-tile kernel: the two waves of a SIMD run half a K step apart
};

// TO = bf16: the result is rounded to bf16, the short GELU of common.h is exact to that rounding; TO = float: exact erf
template <typename TO> __device__ __forceinline__ float act_fwd(int act, float v) {
    if (act == 1) return sizeof(TO) == 2 ? gelu_fast(v) : gelu_erf(v);
    if (act == 2) { const float r = fmaxf(v, 0.f); return r * r; }
    return v;
}
template <typename TO> __device__ __forceinline__ float act_bwd(int act, float pre) {
    if (act == 1) return sizeof(TO) == 2 ? gelu_fast_grad(pre) : gelu_erf_grad(pre);
    if (act == 2) return 2.f * fmaxf(pre, 0.f);
    return 1.f;
}


// v[0..3] = accumulators of row m, columns n..n+3.  FAST: the out family is 4-element accessible and N % 4 == 0.
template <typename TO, bool FAST>
__device__ __forceinline__ void gemm_epilogue4(const GemmP& p, float (&v)[4], long m, long n, float rscale, int z) {
    TO* __restrict__ out = (TO*)p.out;
    if (p.partial) {
        float* dst = p.partial + ((long)z * p.M + m) * p.N + n;
#pragma unroll
        for (int e = 0; e < 4; e++) if (n + e < p.N) dst[e] = v[e];
        return;
    }
    const bool full = FAST || (p.o_vec && n + 4 <= p.N);
    const long off = m * p.ldo + n;
    if (p.bias) {
#pragma unroll
        for (int e = 0; e < 4; e++) if (n + e < p.N) v[e] += p.bias[n + e];
    }
    if (p.pre) {
        TO* pre = (TO*)p.pre;
        if (full) VecIO<TO, 4>::store(pre + off, v);
        else { for (int e = 0; e < 4; e++) if (n + e < p.N) pre[off + e] = from_f<TO>(v[e]); }
    }
    if (p.mul) {
        const TO* mul = (const TO*)p.mul;
        float h[4] = {0.f, 0.f, 0.f, 0.f};
        if (full) VecIO<TO, 4>::load(mul + off, h);
        else { for (int e = 0; e < 4; e++) if (n + e < p.N) h[e] = to_f(mul[off + e]); }
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] *= act_bwd<TO>(p.act, h[e]);
    } else if (p.act) {
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = act_fwd<TO>(p.act, v[e]);
    }
    if (p.gamma) {
#pragma unroll
        for (int e = 0; e < 4; e++) if (n + e < p.N) v[e] *= p.gamma[n + e];
    }
    if (p.rowscale) {
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] *= rscale;
    }
    if (p.res) {
        const TO* res = (const TO*)p.res;
        float h[4] = {0.f, 0.f, 0.f, 0.f};
        if (full) VecIO<TO, 4>::load(res + off, h);
        else { for (int e = 0; e < 4; e++) if (n + e < p.N) h[e] = to_f(res[off + e]); }
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] += h[e];
    }
    if (full) VecIO<TO, 4>::store(out + off, v);
    else { for (int e = 0; e < 4; e++) if (n + e < p.N) out[off + e] = from_f<TO>(v[e]); }
}

// 8 consecutive columns of one row (row-major epilogue of the large-tile kernel).  vec: the out family (out, pre,
// mul, res) may be accessed 16 bytes at a time here and n + 8 <= N.
template <typename TO> __device__ __forceinline__ void io8_load(const TO* q, float (&v)[8]);
template <> __device__ __forceinline__ void io8_load<bf16>(const bf16* q, float (&v)[8]) { VecIO<bf16, 8>::load(q, v); }
template <> __device__ __forceinline__ void io8_load<float>(const float* q, float (&v)[8]) {
    float a[4], b[4];
    VecIO<float, 4>::load(q, a); VecIO<float, 4>::load(q + 4, b);
#pragma unroll
    for (int e = 0; e < 4; e++) { v[e] = a[e]; v[4 + e] = b[e]; }
}
template <typename TO> __device__ __forceinline__ void io8_store(TO* q, const float (&v)[8]);
template <> __device__ __forceinline__ void io8_store<bf16>(bf16* q, const float (&v)[8]) { VecIO<bf16, 8>::store(q, v); }
template <> __device__ __forceinline__ void io8_store<float>(float* q, const float (&v)[8]) {
    const float a[4] = {v[0], v[1], v[2], v[3]}, b[4] = {v[4], v[5], v[6], v[7]};
    VecIO<float, 4>::store(q, a); VecIO<float, 4>::store(q + 4, b);
}
template <typename TO>
__device__ __forceinline__ void gemm_epilogue8(const GemmP& p, float (&v)[8], long m, long n, bool vec) {
    const long off = m * p.ldo + n;
    const int lim = (int)min((long)8, (long)p.N - n);
    if (p.bias) {
#pragma unroll
        for (int e = 0; e < 8; e++) if (e < lim) v[e] += p.bias[n + e];
    }
    if (p.pre) {
        TO* pre = (TO*)p.pre;
        if (vec) io8_store<TO>(pre + off, v);
        else { for (int e = 0; e < lim; e++) pre[off + e] = from_f<TO>(v[e]); }
    }
    if (p.mul) {
        const TO* mul = (const TO*)p.mul;
        float h[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (vec) io8_load<TO>(mul + off, h);
        else { for (int e = 0; e < lim; e++) h[e] = to_f(mul[off + e]); }
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] *= act_bwd<TO>(p.act, h[e]);
    } else if (p.act) {
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = act_fwd<TO>(p.act, v[e]);
    }
    if (p.gamma) {
#pragma unroll
        for (int e = 0; e < 8; e++) if (e < lim) v[e] *= p.gamma[n + e];
    }
    if (p.rowscale) {
        const float rscale = p.rowscale[m / p.rows_per_scale];
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] *= rscale;
    }
    if (p.res) {
        const TO* res = (const TO*)p.res;
        float h[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (vec) io8_load<TO>(res + off, h);
        else { for (int e = 0; e < lim; e++) h[e] = to_f(res[off + e]); }
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] += h[e];
    }
    TO* __restrict__ out = (TO*)p.out;
    if (vec) io8_store<TO>(out + off, v);
    else { for (int e = 0; e < lim; e++) out[off + e] = from_f<TO>(v[e]); }
}
