// hipBLASLt algorithm sweep for the decoder's weight-gradient products (bf16 operands, f32 result, reduction over the tokens):
// the heuristic's first choice against every algorithm the library holds for the type combination (getAllAlgos).
//   hipcc --offload-arch=gfx950 -O2 -o benchmarks/bin/blaslt_probe benchmarks/blaslt_probe.cpp -lhipblaslt
//   hipcc --offload-arch=gfx950 -O2 -DAS_LIB -shared -fPIC -o benchmarks/bin/libblaslt_probe.so benchmarks/blaslt_probe.cpp -lhipblaslt
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <hipblaslt/hipblaslt-ext.hpp>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>
#define CK(x) do { auto e_ = (x); if ((int)e_ != 0) { printf("%s -> %d (line %d)\n", #x, (int)e_, __LINE__); exit(1); } } while (0)

// run_probe: the sweep, callable from a process that already holds a hipBLASLt (PyTorch bundles its own build under the same
// SONAME; a shared object built with -DAS_LIB and loaded after `import torch` binds to THAT one: benchmarks/blaslt_probe.py)
extern "C" int run_probe(int64_t K) {
    struct Shape { const char* name; int64_t m, n; bool tn; } shapes[] = {
        {"dW1   [6400 x 1600]", 1600, 6400, true}, {"dW2   [1600 x 6400]", 6400, 1600, true},
        {"dWqkv [4800 x 1600]", 1600, 4800, true}, {"dWo   [1600 x 1600]", 1600, 1600, true}};
    hipblasLtHandle_t h; CK(hipblasLtCreate(&h));
    void *A, *B, *C, *ws; const size_t wsz = 256u << 20;
    CK(hipMalloc(&A, K * 6400 * 2)); CK(hipMalloc(&B, K * 6400 * 2)); CK(hipMalloc(&C, 6400 * 6400 * 4)); CK(hipMalloc(&ws, wsz));
    {   // normal(0,1) bf16 operands: constant data draws less power and times 20 % faster than a training step's tensors do
        std::vector<uint16_t> hbuf(K * 6400);
        uint64_t st = 0x9E3779B97F4A7C15ull;
        auto fill = [&](void* dst) {
            for (auto& v : hbuf) {
                float acc = 0.f;
                for (int j = 0; j < 4; j++) { st = st * 6364136223846793005ull + 1442695040888963407ull; acc += (float)(st >> 40) * (1.f / 16777216.f) - 0.5f; }
                float f = acc * 1.7320508f; uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16);
            }
            CK(hipMemcpy(dst, hbuf.data(), hbuf.size() * 2, hipMemcpyHostToDevice));
        };
        fill(A); fill(B);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& s : shapes) {
        // column-major: C[m x n] = A[m x k] * B[n x k]^T   (A = x as [K_in x tokens], B = dy as [N_out x tokens])
        hipblasLtMatmulDesc_t desc; CK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
        hipblasOperation_t opA = HIPBLAS_OP_N, opB = HIPBLAS_OP_T;
        CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opA, sizeof(opA)));
        CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opB, sizeof(opB)));
        hipblasLtMatrixLayout_t la, lb, lc;
        CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, s.m, K, s.m));
        CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, s.n, K, s.n));
        CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_32F, s.m, s.n, s.m));
        const float alpha = 1.f, beta = 0.f;
        auto time_algo = [&](hipblasLtMatmulAlgo_t& algo, size_t need) -> float {
            if (need > wsz) return -1.f;
            for (int i = 0; i < 2; i++)
                if (hipblasLtMatmul(h, desc, &alpha, A, la, B, lb, &beta, C, lc, C, lc, &algo, ws, wsz, 0) != HIPBLAS_STATUS_SUCCESS) return -1.f;
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            const int n = 8;
            for (int i = 0; i < n; i++) hipblasLtMatmul(h, desc, &alpha, A, la, B, lb, &beta, C, lc, C, lc, &algo, ws, wsz, 0);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            return ms * 1e3f / n;
        };
        hipblasLtMatmulPreference_t pref; CK(hipblasLtMatmulPreferenceCreate(&pref));
        CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof(wsz)));
        hipblasLtMatmulHeuristicResult_t heur[8]; int got = 0;
        CK(hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, lc, lc, pref, 8, heur, &got));
        const double flop = 2.0 * s.m * s.n * K;
        printf("%s  heuristic:", s.name);
        for (int i = 0; i < got; i++) { float t = time_algo(heur[i].algo, heur[i].workspaceSize); printf(" %.0f", t); }
        printf(" us (first = the default, %.2f PF/s)\n", got ? flop / time_algo(heur[0].algo, heur[0].workspaceSize) / 1e9 : 0.0);
        std::vector<hipblasLtMatmulHeuristicResult_t> all;
        CK(hipblaslt_ext::getAllAlgos(h, hipblaslt_ext::GemmType::HIPBLASLT_GEMM, opA, opB, HIP_R_16BF, HIP_R_16BF, HIP_R_32F, HIP_R_32F,
                                      HIPBLAS_COMPUTE_32F, all));
        std::vector<std::pair<float, int>> res;
        int supported = 0;
        for (size_t i = 0; i < all.size(); i++) {
            size_t need = 0;
            if (hipblaslt_ext::matmulIsAlgoSupported(h, desc, &alpha, la, lb, &beta, lc, lc, all[i].algo, need) != HIPBLAS_STATUS_SUCCESS) continue;
            supported++;
            float t = time_algo(all[i].algo, need);
            if (t > 0) res.push_back({t, (int)i});
        }
        std::sort(res.begin(), res.end());
        printf("    %zu algorithms, %d support the problem; fastest:", all.size(), supported);
        for (size_t i = 0; i < res.size() && i < 5; i++) {
            std::string nm = hipblaslt_ext::getKernelNameFromAlgo(h, all[res[i].second].algo);
            printf("\n      %.0f us (%.2f PF/s) idx %d %s", res[i].first, flop / res[i].first / 1e9, hipblaslt_ext::getIndexFromAlgo(all[res[i].second].algo), nm.substr(0, 110).c_str());
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
#ifndef AS_LIB
int main(int argc, char** argv) { return run_probe(argc > 1 ? atol(argv[1]) : 10080); }
#endif
