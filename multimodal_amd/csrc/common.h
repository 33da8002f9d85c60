// Shared device/host helpers for libmmhip (gfx950 / CDNA4 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mmhip.h"

typedef __hip_bfloat16 bf16;
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA A/B fragment, 8 bf16
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// ---------------------------------------------------------------- errors
void mm_set_error(const char* fmt, ...);
#define MM_FAIL(code, ...) do { mm_set_error(__VA_ARGS__); return (code); } while (0)
#define MM_CHECK(cond, ...) do { if (!(cond)) MM_FAIL(MM_ERR_ARG, __VA_ARGS__); } while (0)
#define MM_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) MM_FAIL(MM_ERR_LAUNCH, "%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); } while (0)

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- scalar conversion
__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {           // round-to-nearest-even, NaN stays NaN
    return __builtin_bit_cast(unsigned short, __float2bfloat16(f));
}
__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(bf16 v) { return __bfloat162float(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float v) { return __float2bfloat16(v); }

// ---------------------------------------------------------------- vector load/store of VEC elements -> float[VEC]
template <typename T, int VEC> struct VecIO;
typedef __attribute__((ext_vector_type(4))) float mm_f4_t;
typedef __attribute__((ext_vector_type(4))) unsigned mm_u4_t;
typedef __attribute__((ext_vector_type(2))) unsigned mm_u2_t;
template <> struct VecIO<float, 4> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
        float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
    // raw form: keep the 16 loaded bytes packed until they are needed (several rows in flight per lane)
    typedef mm_f4_t raw_t;
    static __device__ __forceinline__ raw_t load_raw(const float* p) { return *reinterpret_cast<const mm_f4_t*>(p); }
    static __device__ __forceinline__ raw_t load_raw_nt(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const mm_f4_t*>(p)); }
    static __device__ __forceinline__ void unpack(raw_t t, float (&v)[4]) { v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
    // non-temporal forms for once-touched streams (measured on K1: +4% fwd stores, +10% bwd loads in f32; bf16 forward
    // stores 111 -> 90 us at B8192 C64 L250, round 3 -- until then the bf16 forms were aliases of the plain ones)
    static __device__ __forceinline__ void load_nt(const float* p, float (&v)[4]) {
        mm_f4_t t = __builtin_nontemporal_load(reinterpret_cast<const mm_f4_t*>(p));
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
    static __device__ __forceinline__ void store_nt(float* p, const float (&v)[4]) {
        mm_f4_t t = {v[0], v[1], v[2], v[3]};
        __builtin_nontemporal_store(t, reinterpret_cast<mm_f4_t*>(p)); }
};
template <> struct VecIO<float, 2> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[2]) {
        float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y; }
    static __device__ __forceinline__ void store(float* p, const float (&v)[2]) {
        *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]); }
};
template <> struct VecIO<float, 1> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[1]) { v[0] = *p; }
    static __device__ __forceinline__ void store(float* p, const float (&v)[1]) { *p = v[0]; }
};
template <> struct VecIO<bf16, 8> {
    typedef mm_u4_t raw_t;
    static __device__ __forceinline__ raw_t load_raw(const bf16* p) { return *reinterpret_cast<const mm_u4_t*>(p); }
    static __device__ __forceinline__ raw_t load_raw_nt(const bf16* p) { return __builtin_nontemporal_load(reinterpret_cast<const mm_u4_t*>(p)); }
    static __device__ __forceinline__ void unpack(raw_t t, float (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 4; i++) { v[2 * i] = __uint_as_float(t[i] << 16); v[2 * i + 1] = __uint_as_float(t[i] & 0xffff0000u); }
    }
    static __device__ __forceinline__ void load(const bf16* p, float (&v)[8]) {
        uint4 t = *reinterpret_cast<const uint4*>(p);
        unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; i++) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    }
    static __device__ __forceinline__ void store(bf16* p, const float (&v)[8]) {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; i++) w[i] = (unsigned)f2bf(v[2 * i]) | ((unsigned)f2bf(v[2 * i + 1]) << 16);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    static __device__ __forceinline__ void load_nt(const bf16* p, float (&v)[8]) { unpack(load_raw_nt(p), v); }
    static __device__ __forceinline__ void store_nt(bf16* p, const float (&v)[8]) {
        mm_u4_t w;
#pragma unroll
        for (int i = 0; i < 4; i++) w[i] = (unsigned)f2bf(v[2 * i]) | ((unsigned)f2bf(v[2 * i + 1]) << 16);
        __builtin_nontemporal_store(w, reinterpret_cast<mm_u4_t*>(p));
    }
};
template <> struct VecIO<bf16, 4> {
    typedef mm_u2_t raw_t;
    static __device__ __forceinline__ raw_t load_raw(const bf16* p) { return *reinterpret_cast<const mm_u2_t*>(p); }
    static __device__ __forceinline__ raw_t load_raw_nt(const bf16* p) { return __builtin_nontemporal_load(reinterpret_cast<const mm_u2_t*>(p)); }
    static __device__ __forceinline__ void unpack(raw_t t, float (&v)[4]) {
        v[0] = __uint_as_float(t[0] << 16); v[1] = __uint_as_float(t[0] & 0xffff0000u);
        v[2] = __uint_as_float(t[1] << 16); v[3] = __uint_as_float(t[1] & 0xffff0000u);
    }
    static __device__ __forceinline__ void load(const bf16* p, float (&v)[4]) {
        uint2 t = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    }
    static __device__ __forceinline__ void store(bf16* p, const float (&v)[4]) {
        uint2 t;
        t.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
        t.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
        *reinterpret_cast<uint2*>(p) = t;
    }
    static __device__ __forceinline__ void load_nt(const bf16* p, float (&v)[4]) { unpack(load_raw_nt(p), v); }
    static __device__ __forceinline__ void store_nt(bf16* p, const float (&v)[4]) {
        mm_u2_t t;
        t[0] = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
        t[1] = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
        __builtin_nontemporal_store(t, reinterpret_cast<mm_u2_t*>(p));
    }
};
template <> struct VecIO<bf16, 2> {
    static __device__ __forceinline__ void load(const bf16* p, float (&v)[2]) {
        unsigned t = *reinterpret_cast<const unsigned*>(p);
        v[0] = __uint_as_float(t << 16); v[1] = __uint_as_float(t & 0xffff0000u); }
    static __device__ __forceinline__ void store(bf16* p, const float (&v)[2]) {
        *reinterpret_cast<unsigned*>(p) = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16); }
};
template <> struct VecIO<bf16, 1> {
    static __device__ __forceinline__ void load(const bf16* p, float (&v)[1]) { v[0] = __bfloat162float(*p); }
    static __device__ __forceinline__ void store(bf16* p, const float (&v)[1]) { *p = __float2bfloat16(v[0]); }
};

// ---------------------------------------------------------------- cross-lane sums (wave64, DPP for the in-row steps)
// old = 0 with bound_ctrl set: every control used here reads a valid lane, so the value is that of a plain DPP move, and in
// this form the compiler folds the move into the add that follows it (v_add_f32_dpp: one instruction for two -- the strip K1b
// loop held 64 separate v_mov_b32_dpp per 8 rows)
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
#ifdef MM_DPP_PLAIN_MOV          // A/B: the separate-move form (benchmarks/build_variant.sh plainmov dwconv_ln -DMM_DPP_PLAIN_MOV)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
#else
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
#endif
}
// Sum over aligned groups of G consecutive lanes (G in 1,2,4,8,16,32,64); every lane of the group gets the sum.
// All 64 lanes must be active.
template <int G> __device__ __forceinline__ float group_sum(float v) {
    if (G >= 2) v += dpp_f<0xB1>(v);        // quad_perm [1,0,3,2]
    if (G >= 4) v += dpp_f<0x4E>(v);        // quad_perm [2,3,0,1]
    if (G >= 8) v += dpp_f<0x141>(v);       // row_half_mirror
    if (G >= 16) v += dpp_f<0x140>(v);      // row_mirror
    if (G >= 32) v += __shfl_xor(v, 16);
    if (G >= 64) v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }

// Wave-wide sum returned as a wave-UNIFORM value (it lives in an SGPR): 4 DPP steps inside each row of 16 lanes, then
// row_bcast:15 / row_bcast:31 carry the row totals into lane 63, which v_readlane reads.  No LDS permute, no wait; the
// callers below only need the total on one lane (to post it) or as a scalar operand.  All 64 lanes must be active.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_f_rows(float v) {      // rows outside ROW_MASK get 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_u(float v) {
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    v += dpp_f<0x140>(v);                  // every lane: the sum of its row of 16
    v += dpp_f_rows<0x142, 0xA>(v);        // row_bcast:15 -> rows 1 and 3 add the total of the row before them
    v += dpp_f_rows<0x143, 0xC>(v);        // row_bcast:31 -> rows 2 and 3 add lane 31 (= rows 0+1): lane 63 holds the wave total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// Sum over the first 8 lanes of a wave (lanes 8..63 may hold anything), uniform result.
__device__ __forceinline__ float first8_sum_u(float v) {
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);                  // row_half_mirror: lanes 0..7 of a row now hold the sum of those 8 lanes
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// Block-wide sum for blockDim.x == NW*64; red must hold NW floats; contains ONE __syncthreads().
template <int NW> __device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; i++) t += red[i];
    return t;
}

// exact-erf GELU (nn.GELU() default, reference models/convnext.py:30) and its derivative
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// The same function for results that are rounded to bf16 (relative step 2^-9): Phi(x) through erfc(|x|/sqrt 2) =
// poly(t) * exp(-x^2/2), t = 1 / (1 + p |x|/sqrt 2) (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 on erf; taken on the erfc side
// so the negative tail does not cancel).  14 instructions, two of them transcendental, against the library erff's ~50 with
// branches: in a GEMM epilogue, where two waves per SIMD are resident, the exact form took as long as the product itself
// (pwconv1 + GELU 310 us against 168 us for pwconv2 with the same flops at 7936 x 1600 x 6400).  ex = exp(-x^2/2) is returned for
// the derivative, which needs the same exponential: gelu'(x) = Phi(x) + x * ex / sqrt(2 pi).
__device__ __forceinline__ float gelu_phi_fast(float x, float& ex) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    ex = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
    const float half_erfc = 0.5f * (poly * t) * ex;
    return x < 0.f ? half_erfc : 1.0f - half_erfc;
}
__device__ __forceinline__ float gelu_fast(float x) { float ex; return x * gelu_phi_fast(x, ex); }
__device__ __forceinline__ float gelu_fast_grad(float x) {
    float ex;
    const float phi = gelu_phi_fast(x, ex);
    return fmaf(x * 0.39894228040143267794f, ex, phi);
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
