// Device helpers shared by the bf16 fast attention kernels (attention_fast.hip: head_dim <= 192; attention_wide.hip:
// head_dim 512): fragment loads, packing, the prefix-LM mask limits and the XCD-aware block map.
#pragma once
#include "attention.h"
#include <type_traits>

namespace {
constexpr int FP = 16;      // LDS row padding (elements): row stride = odd multiple of 8 banks -> tr and b128 reads conflict-free
typedef bf16x4_t __attribute__((address_space(3))) * lds4p;

__device__ __forceinline__ float rcpf(float x) { return __builtin_amdgcn_rcpf(x); }

// Streamed [ROWS][DP] tile of a [S][d] head slice: 8-byte pieces, thread -> piece idx = tid + 256 j.
template <int DP, int ROWS> struct Tile {
    static constexpr int V4 = DP / 4, IT = ROWS * V4 / 256, PITCH = DP + FP;
    static_assert(ROWS * V4 % 256 == 0, "tile must be a whole number of 256-thread passes");
    uint2 v[IT];
    // rows >= S are clamped to row S-1, columns >= d read column d-4 and become zero
    __device__ __forceinline__ void load(const bf16* __restrict__ g, int ld, int row0, int S, int d) {
#pragma unroll
        for (int j = 0; j < IT; j++) {
            const int idx = threadIdx.x + 256 * j;
            const int rr = idx / V4, c = (idx % V4) * 4;
            const int row = min(row0 + rr, S - 1);
            const uint2 t = *reinterpret_cast<const uint2*>(g + (row * ld + min(c, d - 4)));
            v[j].x = c < d ? t.x : 0u; v[j].y = c < d ? t.y : 0u;
        }
    }
    __device__ __forceinline__ void store(bf16* lds) const {
#pragma unroll
        for (int j = 0; j < IT; j++) {
            const int idx = threadIdx.x + 256 * j;
            const int rr = idx / V4, c = (idx % V4) * 4;
            *reinterpret_cast<uint2*>(&lds[rr * PITCH + c]) = v[j];
        }
    }
};

// 8 consecutive bf16 of one row as an MFMA fragment straight from global (two 8-byte loads; zero beyond d / S)
__device__ __forceinline__ bf16x8_t frag_global(const bf16* __restrict__ g, long ld, int row, int col, int S, int d) {
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
// transposed fragment: columns c0 + fr of rows r0 + {4fg..4fg+3, 16+4fg..16+4fg+3} (the k order the S^T accumulators have)
__device__ __forceinline__ bf16x8_t frag_tr(const bf16* lds, int pitch, int r0, int c0, int fg, int qq, int pp) {
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(&lds[(r0 + 4 * fg + qq) * pitch + c0 + 4 * pp]));
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(&lds[(r0 + 16 + 4 * fg + qq) * pitch + c0 + 4 * pp]));
    return bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ unsigned pack2(float a, float b) {      // one v_cvt_pk_bf16_f32 (round to nearest even)
    typedef float f2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f2_t{a, b}, b2_t));
}
__device__ __forceinline__ bf16x8_t pack8(const float (&v)[8]) {
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    const u4 w = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
    return __builtin_bit_cast(bf16x8_t, w);
}
// last key visible to query qi (prefix-LM mask of transformer_decoder.py:295-298): prompt rows see the whole prompt
__device__ __forceinline__ int last_key(int qi, int P, int S) { return min(qi < P ? max(P - 1, qi) : qi, S - 1); }
// first query that sees key kj: every query sees a prompt key
__device__ __forceinline__ int first_query(int kj, int P) { return kj < P ? 0 : kj; }

// 1-D grid, XCD-aware: workgroups are dealt round-robin over the 8 XCDs, so ids congruent mod 8 share an L2.  The row blocks of
// one (b,h) get consecutive ids on ONE XCD: the head's K/V (or Q/dO) stream is fetched from the fabric once and then hit in
// that L2 (measured with the 2-D grid: 32 % L2 hits, every row block of a head on a different XCD).
__device__ __forceinline__ bool map_block(int nrb, int BH, int& rb, int& bh) {
    const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
    bh = (j / nrb) * 8 + xcd; rb = j % nrb;
    return bh < BH;
}
inline unsigned grid_blocks(int nrb, int BH) { return (unsigned)(nrb * 8 * ((BH + 7) / 8)); }

}  // namespace
