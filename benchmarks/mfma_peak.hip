// Ceiling probe (no product code): sustained bf16 MFMA rate of the chip with nothing else in the way, and with the LDS
// fragment traffic of a GEMM main loop beside it.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// NACC independent accumulators per wave, iters rounds: iters * NACC MFMAs of 16x16x32
template <int NACC>
__global__ __launch_bounds__(512) void pure_k(float* out, int iters) {
    bf16x8 a, b;
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;           // random operands: power depends on the data
    for (int i = 0; i < 8; ++i) {
        h = h * 1664525u + 1013904223u; a[i] = (__bf16)((float)(int)(h >> 16 & 255) / 128.f - 1.f);
        h = h * 1664525u + 1013904223u; b[i] = (__bf16)((float)(int)(h >> 16 & 255) / 128.f - 1.f);
    }
    f32x4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
    }
    float s = 0;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (s == 12345.f) out[0] = s;
}

// the same with LDS fragment reads: per round RA + RB ds_read_b128 feed RA*RB MFMAs (a wave tile of RA x RB blocks, K=32)
template <int RA, int RB>
__global__ __launch_bounds__(512) void lds_k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 65536 / 2; i += blockDim.x) {
        unsigned h = (i + blockIdx.x * 977) * 2654435761u;
        ((__bf16*)smem)[i] = (__bf16)((float)(int)(h >> 16 & 255) / 128.f - 1.f);
    }
    __syncthreads();
    f32x4 acc[RA * RB];
    for (int j = 0; j < RA * RB; ++j) acc[j] = f32x4{0, 0, 0, 0};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = smem + wave * 4096 + lane * 16;
    for (int it = 0; it < iters; ++it) {
        bf16x8 a[RA], b[RB];
        const char* p = base;
#pragma unroll
        for (int i = 0; i < RA; ++i) a[i] = *(const bf16x8*)(p + i * 1024);
#pragma unroll
        for (int i = 0; i < RB; ++i) b[i] = *(const bf16x8*)(p + 32768 + i * 1024);
#pragma unroll
        for (int i = 0; i < RA; ++i)
#pragma unroll
            for (int j = 0; j < RB; ++j) acc[i * RB + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i * RB + j], 0, 0, 0);
        asm volatile("" ::: "memory");
    }
    float s = 0;
    for (int j = 0; j < RA * RB; ++j) s += acc[j][0];
    if (s == 12345.f) out[0] = s;
}

template <typename F>
static void run(const char* name, F launch, double flops_per_launch, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %9.1f us per launch  %7.1f TFLOP/s (%.3f of 2500)\n", name, ms * 1e3 / reps, flops_per_launch * reps / (ms * 1e-3) / 1e12,
           flops_per_launch * reps / (ms * 1e-3) / 2.5e15);
    fflush(stdout);
}

int main() {
    float* out; CHECK(hipMalloc(&out, 4096));
    const double F = 16.0 * 16 * 32 * 2;
    for (int wg : {256, 512}) {                 // 1 or 2 waves per SIMD... (256 threads = 4 waves = 1 per SIMD)
        for (int iters : {2000, 60000}) {       // ~0.2 ms and ~6 ms: does the rate hold once power management reacts?
            char nm[96];
            snprintf(nm, sizeof nm, "pure mfma, 8 acc, %d waves/SIMD, %d rounds", wg / 256, iters);
            run(nm, [&] { pure_k<8><<<256, wg, 0, 0>>>(out, iters); }, F * 8 * iters * (wg / 64) * 256.0, iters > 10000 ? 20 : 200);
        }
    }
    {
        char nm[96];
        snprintf(nm, sizeof nm, "pure mfma, 8 acc, 2 waves/SIMD, 2 WG/CU");
        run(nm, [&] { pure_k<8><<<512, 512, 0, 0>>>(out, 20000); }, F * 8 * 20000 * 8 * 512.0, 50);
    }
    CHECK(hipFuncSetAttribute((const void*)lds_k<7, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipFuncSetAttribute((const void*)lds_k<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipFuncSetAttribute((const void*)lds_k<8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    run("mfma + LDS fragments 7x5 (12 reads / 35)", [&] { lds_k<7, 5><<<256, 512, 65536, 0>>>(out, 4000); }, F * 35 * 4000 * 8 * 256.0, 100);
    run("mfma + LDS fragments 4x4 (8 reads / 16)", [&] { lds_k<4, 4><<<256, 512, 65536, 0>>>(out, 8000); }, F * 16 * 8000 * 8 * 256.0, 100);
    run("mfma + LDS fragments 8x4 (12 reads / 32)", [&] { lds_k<8, 4><<<256, 512, 65536, 0>>>(out, 4000); }, F * 32 * 4000 * 8 * 256.0, 100);
    run("mfma + LDS fragments 7x5, long (6 ms)", [&] { lds_k<7, 5><<<256, 512, 65536, 0>>>(out, 100000); }, F * 35 * 100000 * 8 * 256.0, 10);
    return 0;
}
