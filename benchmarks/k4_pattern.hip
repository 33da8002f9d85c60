// Access-pattern probe for K4 (no product code): what HBM read rate does the KV-cache walk itself allow?
// Same cache layout as the decode kernel ([B*H][Smax][d] bf16, K and V), same row count, no attention arithmetic.
//   hipcc --offload-arch=gfx950 -O3 -o benchmarks/bin/k4_pattern benchmarks/k4_pattern.hip
//   MODE 0: linear read of the same bytes (grid-stride, 16 B per lane): the ceiling
//   MODE 1: K4's mapping: wave per (b,h), 4 rows per instruction, PF requests ahead; K pass, then V pass
//   MODE 2: one pass, K row and V row of the same position in flight together (PF/2 each)
//   MODE 3: as 1, but the four waves of a workgroup walk ONE (b,h) (16 rows = 4 KiB contiguous per workgroup instruction)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void eat(u32x4& acc, const u32x4& v) { acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }

__global__ __launch_bounds__(256) void linear_k(const u32x4* __restrict__ src, long n16, unsigned* out) {
    u32x4 acc = {0, 0, 0, 0};
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < 8; u++) eat(acc, v[u]);
    }
    for (; i < n16; i += stride) eat(acc, src[i]);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}

template <int MODE, int PF>
__global__ __launch_bounds__(256) void walk_k(const char* __restrict__ kc, const char* __restrict__ vc, int BH, int Smax, int pos,
                                              unsigned* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, grp = lane >> 4;
    constexpr int RPI = MODE == 3 ? 16 : 4;                     // rows per iteration of the walking unit
    const int bh = MODE == 3 ? blockIdx.x : blockIdx.x * 4 + wave;
    if (bh >= BH) return;
    const int row0 = MODE == 3 ? wave * 4 + grp : grp;
    const char* kb = kc + (long)bh * Smax * 256 + li * 16;
    const char* vb = vc + (long)bh * Smax * 256 + li * 16;
    const int nit = (pos + RPI - 1) / RPI;
    u32x4 acc = {0, 0, 0, 0};
    auto ld = [&](const char* base, int it) {
        const int j = min(it * RPI + row0, pos - 1);
        return *reinterpret_cast<const u32x4*>(base + (long)j * 256);
    };
    if (MODE == 2) {
        constexpr int P2 = PF / 2;
        u32x4 bk[P2], bv[P2];
#pragma unroll
        for (int u = 0; u < P2; u++) if (u < nit) { bk[u] = ld(kb, u); bv[u] = ld(vb, u); }
        for (int it0 = 0; it0 < nit; it0 += P2) {
#pragma unroll
            for (int u = 0; u < P2; u++) {
                const int it = it0 + u;
                if (it < nit) {
                    eat(acc, bk[u]); eat(acc, bv[u]);
                    if (it + P2 < nit) { bk[u] = ld(kb, it + P2); bv[u] = ld(vb, it + P2); }
                }
            }
        }
    } else {
        u32x4 buf[PF];
        for (int pass = 0; pass < 2; pass++) {
            const char* base = pass ? vb : kb;
#pragma unroll
            for (int u = 0; u < PF; u++) if (u < nit) buf[u] = ld(base, u);
            for (int it0 = 0; it0 < nit; it0 += PF) {
#pragma unroll
                for (int u = 0; u < PF; u++) {
                    const int it = it0 + u;
                    if (it < nit) {
                        eat(acc, buf[u]);
                        if (it + PF < nit) buf[u] = ld(base, it + PF);
                    }
                }
            }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}


// MODE 5: the attention arithmetic on the one-pass walk (online softmax / stablemax per 16-lane group, merged at the end).
template <int N> __device__ __forceinline__ float dppf(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), N, 0xf, 0xf, false));
}
__device__ __forceinline__ float sum16(float v) { v += dppf<0xB1>(v); v += dppf<0x4E>(v); v += dppf<0x141>(v); v += dppf<0x140>(v); return v; }
__device__ __forceinline__ void unpack8(const u32x4& t, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; i++) { v[2 * i] = __uint_as_float(t[i] << 16); v[2 * i + 1] = __uint_as_float(t[i] & 0xffff0000u); }
}
template <bool STABLE, int P2, int WPB>
__global__ __launch_bounds__(64 * WPB) void attn1_k(const char* __restrict__ kc, const char* __restrict__ vc, const unsigned short* __restrict__ q,
                                               unsigned short* __restrict__ o, int BH, int Smax, int pos, float scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, grp = lane >> 4;
    const int bh = blockIdx.x * WPB + wave;
    if (bh >= BH) return;
    const char* kb = kc + (long)bh * Smax * 256 + li * 16;
    const char* vb = vc + (long)bh * Smax * 256 + li * 16;
    const int nit = (pos + 3) >> 2;
    auto ld = [&](const char* base, int it) {
        const int j = min(it * 4 + grp, pos - 1);
        return *reinterpret_cast<const u32x4*>(base + (long)j * 256);
    };
    u32x4 bk[P2], bv[P2];
#pragma unroll
    for (int u = 0; u < P2; u++) if (u < nit) { bk[u] = ld(kb, u); bv[u] = ld(vb, u); }
    float qv[8];
    { u32x4 t = *reinterpret_cast<const u32x4*>(q + (long)bh * 128 + li * 8); unpack8(t, qv); }
    const float sc2 = STABLE ? scale : scale * 1.44269504f;
#pragma unroll
    for (int i = 0; i < 8; i++) qv[i] *= sc2;
    float oacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, l = 0.f, m = -1e30f;
    for (int it0 = 0; it0 < nit; it0 += P2) {
#pragma unroll
        for (int u = 0; u < P2; u++) {
            const int it = it0 + u;
            if (it < nit) {
                const bool ok = it * 4 + grp < pos;
                float kv[8], vv[8];
                unpack8(bk[u], kv); unpack8(bv[u], vv);
                if (it + P2 < nit) { bk[u] = ld(kb, it + P2); bv[u] = ld(vb, it + P2); }
                float dot = 0.f;
#pragma unroll
                for (int i = 0; i < 8; i++) dot = fmaf(qv[i], kv[i], dot);
                const float x = sum16(dot);
                float w;
                if (STABLE) {
                    w = ok ? (x < 0.f ? __builtin_amdgcn_rcpf(1.0f - x) : x + 1.0f) : 0.f;
                } else {
                    const float mn = ok ? fmaxf(m, x) : m;
                    const float al = __builtin_amdgcn_exp2f(m - mn);
                    w = ok ? __builtin_amdgcn_exp2f(x - mn) : 0.f;
                    m = mn; l *= al;
#pragma unroll
                    for (int i = 0; i < 8; i++) oacc[i] *= al;
                }
                l += w;
#pragma unroll
                for (int i = 0; i < 8; i++) oacc[i] = fmaf(w, vv[i], oacc[i]);
            }
        }
    }
    float f = 1.f;
    if (!STABLE) {
        float M = fmaxf(m, __shfl_xor(m, 16)); M = fmaxf(M, __shfl_xor(M, 32));
        f = __builtin_amdgcn_exp2f(m - M);
    }
    l *= f; l += __shfl_xor(l, 16); l += __shfl_xor(l, 32);
    const float inv = 1.0f / fmaxf(l, 1e-9f);
    unsigned r[4];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        float t = oacc[i] * f; t += __shfl_xor(t, 16); t += __shfl_xor(t, 32); t *= inv;
        const unsigned b = __float_as_uint(t) >> 16;
        if (i & 1) r[i >> 1] |= b << 16; else r[i >> 1] = b;
    }
    if (grp == 0) *reinterpret_cast<u32x4*>(o + (long)bh * 128 + li * 8) = u32x4{r[0], r[1], r[2], r[3]};
}

int main(int argc, char** argv) {
    const int BH = argc > 1 ? atoi(argv[1]) : 8192, Smax = 238, pos = argc > 2 ? atoi(argv[2]) : 238;
    const long bytes = (long)BH * Smax * 256;
    char *kc, *vc; unsigned* out;
    CHECK(hipMalloc(&kc, 2 * bytes)); vc = kc + bytes; CHECK(hipMalloc(&out, 64));       // one allocation: the linear read walks both
    CHECK(hipMemset(kc, 1, bytes)); CHECK(hipMemset(vc, 2, bytes));
    // a second pair of caches so consecutive launches do not find their rows in the 256 MiB last-level cache
    char *kc2, *vc2;
    CHECK(hipMalloc(&kc2, 2 * bytes)); vc2 = kc2 + bytes;
    CHECK(hipMemset(kc2, 3, bytes)); CHECK(hipMemset(vc2, 4, bytes));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const double algo = 2.0 * BH * pos * 256;
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; i++) launch(i & 1);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        const int n = 20;
        for (int i = 0; i < n; i++) launch(i & 1);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %8.1f us  %7.0f GB/s  (%.3f of 8 TB/s)\n", name, ms * 1e3 / n, algo / (ms * 1e-3 / n) / 1e9, algo / (ms * 1e-3 / n) / 8e12);
    };
    run("linear read, 2048 blocks", [&](int s) {
        hipLaunchKernelGGL(linear_k, dim3(2048), dim3(256), 0, 0, (const u32x4*)(s ? kc2 : kc), (long)(algo / 16), out); });
    run("linear read, 8192 blocks", [&](int s) {
        hipLaunchKernelGGL(linear_k, dim3(8192), dim3(256), 0, 0, (const u32x4*)(s ? kc2 : kc), (long)(algo / 16), out); });
#define W_(MODE, PF, GRID) run("walk mode " #MODE " PF " #PF, [&](int s) { \
        hipLaunchKernelGGL((walk_k<MODE, PF>), dim3(GRID), dim3(256), 0, 0, s ? kc2 : kc, s ? vc2 : vc, BH, Smax, pos, out); })
    W_(1, 8, BH / 4); W_(1, 4, BH / 4); W_(1, 16, BH / 4);
    W_(2, 8, BH / 4); W_(2, 16, BH / 4);
    W_(3, 8, BH); W_(3, 4, BH); W_(3, 2, BH);
    unsigned short *q, *o;
    CHECK(hipMalloc(&q, (long)BH * 256)); CHECK(hipMalloc(&o, (long)BH * 256)); CHECK(hipMemset(q, 0x3c, (long)BH * 256));
#define A_(ST, P2, WPB) run("attn one pass " #ST " P2=" #P2 " waves/block " #WPB, [&](int s) { \
        hipLaunchKernelGGL((attn1_k<ST, P2, WPB>), dim3(BH / WPB), dim3(64 * WPB), 0, 0, s ? kc2 : kc, s ? vc2 : vc, q, o, BH, Smax, pos, 0.088f); })
    A_(false, 4, 4); A_(false, 2, 4); A_(false, 3, 4); A_(true, 4, 4); A_(true, 2, 4); A_(false, 4, 1); A_(false, 4, 2); A_(false, 4, 8);
    CHECK(hipDeviceSynchronize());
    return 0;
}
