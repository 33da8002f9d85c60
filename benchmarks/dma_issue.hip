// Issue-cost probe (no product code): one wave per SIMD runs bf16 MFMAs back to back; every 8 MFMAs it also issues one
// 1 KiB load in one of several forms.  Cycles per MFMA above 16 are what the load costs the wave.
//   hipcc --offload-arch=gfx950 -O3 -o benchmarks/bin/dma_issue benchmarks/dma_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0: MFMA only; 1: global_load_lds, 64-bit lane address; 2: global_load_lds, scalar base + 32-bit lane offset;
// 3: global_load_dwordx4 into registers (4 sets in rotation); 4: mode 2 + two ds_read_b128 per 8 MFMAs
// PER: MFMAs between two loads.  ROWS: 8 -> a piece is 8 rows x 128 B, 16 -> 16 rows x 64 B (stride `ld` bytes)
template <int MODE, int PER, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void probe_k(const char* __restrict__ src, long ld, int iters, float* out, long per_cu_bytes) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) ((float*)smem)[i] = (float)(i & 7) - 3.f;
    __syncthreads();
    bf16x8 a, b;
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < 8; ++i) {
        h = h * 1664525u + 1013904223u; a[i] = (__bf16)((float)(int)(h >> 16 & 255) / 128.f - 1.f);
        h = h * 1664525u + 1013904223u; b[i] = (__bf16)((float)(int)(h >> 16 & 255) / 128.f - 1.f);
    }
    f32x4 acc[PER];
    for (int j = 0; j < PER; ++j) acc[j] = f32x4{0, 0, 0, 0};
    const char* base = src + (long)blockIdx.x * per_cu_bytes;                       // this CU's slice
    const unsigned voff = (unsigned)((lane >> 3) * ld + (lane & 7) * 16);            // 8 rows x 128 B
    u32x4 stg[4] = {};
    bf16x8 f0 = a, f1 = b;
    unsigned char* dst = smem + 32768 + wave * 1024;
    long pos = wave * 8 * ld;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < PER; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(MODE == 4 ? f0 : a, MODE == 4 ? f1 : b, acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 1) __builtin_amdgcn_global_load_lds((glb_vp)(base + pos + (long)voff), (lds_vp)(dst + (it & 7) * 4096), 16, 0, 0);
        if (MODE == 2 || MODE == 4) {
            const char* sb = base + pos;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)sb), hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)sb >> 32));
            __builtin_amdgcn_global_load_lds((glb_vp)((const char*)(((uintptr_t)hi << 32) | lo) + voff), (lds_vp)(dst + (it & 7) * 4096), 16, 0, 0);
        }
        if (MODE == 3) stg[it & 3] = *(const u32x4*)(base + pos + (long)voff);
        if (MODE == 4) {
            f0 = *(const bf16x8*)(smem + ((it & 15) * 1024) + lane * 16);
            f1 = *(const bf16x8*)(smem + 16384 + ((it & 15) * 1024) + lane * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
        pos += (long)WAVES * 8 * ld;
        if (pos + 8 * ld > per_cu_bytes) pos = wave * 8 * ld;
        if (MODE != 0 && (it & 7) == 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0;
    for (int j = 0; j < PER; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    s += (float)(stg[0][0] + stg[1][1] + stg[2][2] + stg[3][3]) + (float)f0[0];
    if (s == 12345.f) out[0] = s;
}

template <int MODE, int PER, int WAVES>
static void run(const char* name, const char* src, long ld, long per_cu, float* out) {
    const int iters = 4000;
    auto k = probe_k<MODE, PER, WAVES>;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<<<256, 64 * WAVES, 98304, 0>>>(src, ld, iters, out, per_cu); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) k<<<256, 64 * WAVES, 98304, 0>>>(src, ld, iters, out, per_cu);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / 5;
    const double mf = (double)iters * PER * (WAVES / 4.0);                       // MFMAs per SIMD
    const double tf = 16.0 * 16 * 32 * 2 * iters * PER * WAVES * 256 / (us * 1e-6) / 1e12;
    printf("%-64s %8.1f us  %6.1f ns per SIMD-MFMA  %7.1f TFLOP/s  load rate %.1f B/clk/CU at 2.1 GHz\n", name, us, us * 1e3 / mf, tf,
           MODE ? 1024.0 * iters * WAVES / (us * 1e-6 * 2.1e9) : 0.0);
    fflush(stdout);
}

int main() {
    const long per_cu = 1 << 20;                  // 1 MiB per CU: 256 MiB in all (beyond L2, inside the Infinity Cache)
    char* src; float* out;
    CHECK(hipMalloc(&src, per_cu * 256 + 65536)); CHECK(hipMemset(src, 1, per_cu * 256 + 65536)); CHECK(hipMalloc(&out, 4096));
    const long ld = 12800;                        // row pitch of a K = 6400 bf16 operand
    run<0, 8, 4>("1 wave/SIMD: mfma only", src, ld, per_cu, out);
    run<1, 8, 4>("1 wave/SIMD: + global_load_lds (64-bit lane address) per 8 mfma", src, ld, per_cu, out);
    run<2, 8, 4>("1 wave/SIMD: + global_load_lds (scalar base + lane offset) per 8", src, ld, per_cu, out);
    run<3, 8, 4>("1 wave/SIMD: + global_load_dwordx4 to registers per 8 mfma", src, ld, per_cu, out);
    run<4, 8, 4>("1 wave/SIMD: + DMA (scalar base) + 2 ds_read_b128 per 8 mfma", src, ld, per_cu, out);
    run<2, 4, 4>("1 wave/SIMD: + DMA (scalar base) per 4 mfma", src, ld, per_cu, out);
    run<2, 16, 4>("1 wave/SIMD: + DMA (scalar base) per 16 mfma", src, ld, per_cu, out);
    run<0, 8, 8>("2 waves/SIMD: mfma only", src, ld, per_cu, out);
    run<2, 8, 8>("2 waves/SIMD: + DMA (scalar base) per 8 mfma each", src, ld, per_cu, out);
    run<4, 8, 8>("2 waves/SIMD: + DMA + 2 ds_read_b128 per 8 mfma each", src, ld, per_cu, out);
    return 0;
}
