// Lab harness for the large-tile GEMM kernels (no product code): builds gemm_big.hip alone with ONE tile shape, times it on a
// shape from the command line and prints the cycle stamps of workgroup 0 (where a K step's time goes, wave by wave).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DGEMM_LAB -DGEMM_LAB_TM=7 -DGEMM_LAB_TN=5 -DGEMM_LAB_KT0=16 \
//         -I multimodal_amd/csrc -I include -o benchmarks/bin/gemm_lab_75 benchmarks/gemm_lab.hip
//   gemm_lab M N K [tn]      (tn: weight-gradient form, both operands K-major)
#include "../multimodal_amd/csrc/gemm_big.hip"
#include <stdarg.h>
#include <stdio.h>
#include <vector>
#include <math.h>

void mm_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill_k(bf16* p, long n, unsigned seed) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (bf16)((float)(int)(h & 1023) / 512.f - 1.f);
    }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 10080, N = argc > 2 ? atoi(argv[2]) : 1600, K = argc > 3 ? atoi(argv[3]) : 6400;
    const bool tn = argc > 4 && argv[4][0] == 't';
    bf16 *A, *B, *C; float* ws;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
    const size_t ws_bytes = tn ? (size_t)8 * M * N * 4 : 0;
    if (tn) CK(hipMalloc(&ws, ws_bytes)); else ws = nullptr;
    fill_k<<<4096, 256>>>(A, (long)M * K, 1u); fill_k<<<4096, 256>>>(B, (long)N * K, 7u);
    CK(hipDeviceSynchronize());
    GemmP p = {};
    p.A = A; p.B = B; p.out = C; p.M = M; p.N = N; p.K = K;
    p.lda = tn ? M : K; p.ldb = tn ? N : K; p.ldo = N; p.a_vec = p.b_vec = 1; p.rows_per_scale = 1;
    auto go = [&]() { GemmP q = p; int rc = tn ? gemm_big_tn_launch(q, MM_BF16, 0, ws, ws_bytes) : gemm_big_launch(q, MM_BF16, 0, nullptr, 0); if (rc) exit(2); };
    go(); CK(hipDeviceSynchronize());
    // check a few entries against a host dot product
    std::vector<bf16> hA((size_t)M * K), hB((size_t)N * K), hC((size_t)M * N);
    CK(hipMemcpy(hA.data(), A, hA.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hB.data(), B, hB.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hC.data(), C, hC.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < 64; t++) {
        const int m = (int)((t * 7919L + 13) % M), n = (int)((t * 104729L + 5) % N);
        double s = 0;
        for (int k = 0; k < K; k++) s += tn ? (double)(float)hA[(size_t)k * M + m] * (float)hB[(size_t)k * N + n] : (double)(float)hA[(size_t)m * K + k] * (float)hB[(size_t)n * K + k];
        worst = fmax(worst, fabs(s - (float)hC[(size_t)m * N + n]) / (fabs(s) + 1.0));
    }
    printf("check: worst relative error of 64 entries %.2e %s\n", worst, worst < 2e-2 ? "ok" : "WRONG");
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 30;
    for (int r = 0; r < 3; r++) go();
    CK(hipEventRecord(e0)); for (int r = 0; r < reps; r++) go(); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("%d x %d x %d %s tile %dx%d: %.1f us  %.3f PF/s\n", M, N, K, tn ? "TN" : "NT", 32 * GEMM_LAB_TM, 64 * GEMM_LAB_TN, us, 2.0 * M * N * K / us / 1e9);
    unsigned long long st[8][16][8];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof st));
    // per wave: average over the 16 stamped steps of (slot s+1 - slot s) and the step period
    for (int w = 0; w < 8; w++) {
        printf("wave %d:", w);
        double per = 0; int np = 0;
        for (int k = 1; k < 16; k++) if (st[w][k][0] > st[w][k - 1][0] && st[w][k - 1][0] && st[w][k][0] - st[w][k - 1][0] < 10000000ull) { per += (double)(st[w][k][0] - st[w][k - 1][0]); np++; }
        for (int s = 0; s < 7; s++) {
            double d = 0; int n = 0;
            for (int k = 0; k < 16; k++) if (st[w][k][s + 1] > st[w][k][s] && st[w][k][s] && st[w][k][s + 1] - st[w][k][s] < 10000000ull) { d += (double)(st[w][k][s + 1] - st[w][k][s]); n++; }
            if (n) printf("  s%d->s%d %6.0f", s, s + 1, d / n);
        }
        printf("  | period %6.0f cycles\n", np ? per / np : 0.0);
    }
    return 0;
}
