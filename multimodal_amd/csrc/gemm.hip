// MFMA GEMM family with fused epilogues:  out[M][N] = epi( A[M][K] . B[N][K]^T ).
// Replaces nn.Linear / F.linear on the hot path (reference models/convnext.py:41,:43,:80,:134) and the
// matmuls autograd derives from them (dX = dY.W, dW = dY^T.X).
//
// bf16: v_mfma_f32_16x16x32_bf16, f32 accumulate.   f32: v_mfma_f32_16x16x4_f32 (exact f32 FMA chains).
// Tile BM x BN per 256-thread workgroup (4 waves as 2x2); operands are staged global -> registers -> LDS
// (16-byte loads), the next K tile is fetched while the current one is multiplied.
// Operand layouts: K-contiguous ([rows][K]) tiles are kept [rows][BK] in LDS and read with ds_read_b128;
// K-major ([K][rows]) bf16 tiles are kept [BK][rows] and read with ds_read_b64_tr_b16 (hardware transpose),
// so dX and dW products need no transposed copies in HBM.
// The MFMA is issued as (B-fragment, A-fragment): D's register index then runs along N, so every lane
// owns 4 consecutive output columns and the epilogue moves 8/16 bytes per lane.
#include "common.h"
#include <mutex>
#include "gemm_epi.h"
#include <stdlib.h>

template <typename T> struct Mma;
template <> struct Mma<bf16> {
    static constexpr int BK = 64, EPV = 8;
    static constexpr int PAD = 8;
};
template <> struct Mma<float> {
    static constexpr int BK = 16, EPV = 4;
    static constexpr int PAD = 1;
};

// ---- guarded EPV-wide global load into raw storage ------------------------------------------------
template <typename T, int E, bool FAST>
__device__ __forceinline__ void load_vec(const T* __restrict__ base, bool vec_ok, long slow, long slow_lim, long fast,
                                         long fast_lim, long ld, T (&out)[E]) {
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    if constexpr (FAST) {       // host guarantees: 16-byte aligned rows and a contiguous extent that is a multiple of E
        if (slow < slow_lim && fast < fast_lim) *reinterpret_cast<u4*>(&out[0]) = *reinterpret_cast<const u4*>(base + slow * ld + fast);
        else *reinterpret_cast<u4*>(&out[0]) = u4{0u, 0u, 0u, 0u};
    } else if (slow < slow_lim && vec_ok && fast + E <= fast_lim) {
        *reinterpret_cast<u4*>(&out[0]) = *reinterpret_cast<const u4*>(base + slow * ld + fast);
    } else {
#pragma unroll
        for (int j = 0; j < E; j++)
            out[j] = (slow < slow_lim && fast + j < fast_lim) ? base[slow * ld + fast + j] : T(0.0f);
    }
}

template <typename T, typename TO, int BM, int BN, bool AKM, bool BKM, bool FAST>
#ifndef GEMM_MIN_WAVES
#define GEMM_MIN_WAVES 2
#endif
__global__ __launch_bounds__(256, GEMM_MIN_WAVES) void gemm_k(const GemmP p) {
    constexpr int BK = Mma<T>::BK, EPV = Mma<T>::EPV, PAD = Mma<T>::PAD;
    constexpr bool IS_BF16 = sizeof(T) == 2;
    // LDS images.  bf16: K-contiguous -> [ROWS][BK+PAD];  K-major -> [BK][ROWS+PAD].  f32: always [ROWS][BK+1].
    constexpr int A_ELEMS = (IS_BF16 && AKM) ? BK * (BM + PAD) : BM * (BK + PAD);
    constexpr int B_ELEMS = (IS_BF16 && BKM) ? BK * (BN + PAD) : BN * (BK + PAD);
    __shared__ __attribute__((aligned(16))) T lds_a[A_ELEMS];
    __shared__ __attribute__((aligned(16))) T lds_b[B_ELEMS];
    constexpr int WM = BM / 2, WN = BN / 2;          // wave tile
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr int NVA = BM * BK / EPV / 256, NVB = BN * BK / EPV / 256;
    static_assert(NVA >= 1 && NVB >= 1, "tile too small for 256 threads");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware, bijective remap of the linear block id: each XCD (ids congruent mod 8) gets a contiguous
    // run of tiles, so neighbouring tiles that share an operand panel share an L2
    int wg;
    {
        const int nwg = p.tiles_m * p.tiles_n, id = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int tile_m = wg / p.tiles_n, tile_n = wg % p.tiles_n;
    const long m0 = (long)tile_m * BM, n0 = (long)tile_n * BN;
    const int kbeg = blockIdx.z * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const T* __restrict__ A = (const T*)p.A;
    const T* __restrict__ B = (const T*)p.B;

    alignas(16) T ra[NVA][EPV];
    alignas(16) T rb[NVB][EPV];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NVA; i++) {
            const int v = tid + 256 * i;
            if (!AKM) { const int row = v / (BK / EPV), kc = v % (BK / EPV);
                load_vec<T, EPV, FAST>(A, p.a_vec, m0 + row, p.M, k0 + kc * EPV, kend, p.lda, ra[i]); }
            else { const int k = v / (BM / EPV), rc = v % (BM / EPV);
                load_vec<T, EPV, FAST>(A, p.a_vec, k0 + k, kend, m0 + rc * EPV, p.M, p.lda, ra[i]); }
        }
#pragma unroll
        for (int i = 0; i < NVB; i++) {
            const int v = tid + 256 * i;
            if (!BKM) { const int row = v / (BK / EPV), kc = v % (BK / EPV);
                load_vec<T, EPV, FAST>(B, p.b_vec, n0 + row, p.N, k0 + kc * EPV, kend, p.ldb, rb[i]); }
            else { const int k = v / (BN / EPV), rc = v % (BN / EPV);
                load_vec<T, EPV, FAST>(B, p.b_vec, k0 + k, kend, n0 + rc * EPV, p.N, p.ldb, rb[i]); }
        }
    };
    auto stash = [&]() {
        typedef __attribute__((ext_vector_type(4))) unsigned u4;
#pragma unroll
        for (int i = 0; i < NVA; i++) {
            const int v = tid + 256 * i;
            if constexpr (IS_BF16) {
                if (!AKM) { const int row = v / (BK / EPV), kc = v % (BK / EPV);
                    *reinterpret_cast<u4*>(&lds_a[row * (BK + PAD) + kc * EPV]) = *reinterpret_cast<u4*>(&ra[i][0]); }
                else { const int k = v / (BM / EPV), rc = v % (BM / EPV);
                    *reinterpret_cast<u4*>(&lds_a[k * (BM + PAD) + rc * EPV]) = *reinterpret_cast<u4*>(&ra[i][0]); }
            } else {
                if (!AKM) { const int row = v / (BK / EPV), kc = v % (BK / EPV);
#pragma unroll
                    for (int j = 0; j < EPV; j++) lds_a[row * (BK + PAD) + kc * EPV + j] = ra[i][j]; }
                else { const int k = v / (BM / EPV), rc = v % (BM / EPV);
#pragma unroll
                    for (int j = 0; j < EPV; j++) lds_a[(rc * EPV + j) * (BK + PAD) + k] = ra[i][j]; }
            }
        }
#pragma unroll
        for (int i = 0; i < NVB; i++) {
            const int v = tid + 256 * i;
            if constexpr (IS_BF16) {
                if (!BKM) { const int row = v / (BK / EPV), kc = v % (BK / EPV);
                    *reinterpret_cast<u4*>(&lds_b[row * (BK + PAD) + kc * EPV]) = *reinterpret_cast<u4*>(&rb[i][0]); }
                else { const int k = v / (BN / EPV), rc = v % (BN / EPV);
                    *reinterpret_cast<u4*>(&lds_b[k * (BN + PAD) + rc * EPV]) = *reinterpret_cast<u4*>(&rb[i][0]); }
            } else {
                if (!BKM) { const int row = v / (BK / EPV), kc = v % (BK / EPV);
#pragma unroll
                    for (int j = 0; j < EPV; j++) lds_b[row * (BK + PAD) + kc * EPV + j] = rb[i][j]; }
                else { const int k = v / (BN / EPV), rc = v % (BN / EPV);
#pragma unroll
                    for (int j = 0; j < EPV; j++) lds_b[(rc * EPV + j) * (BK + PAD) + k] = rb[i][j]; }
            }
        }
    };

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;           // fragment row, k group
    fetch(kbeg);
    stash();
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = k0 + BK < kend;
        if (more) fetch(k0 + BK);
        if constexpr (IS_BF16) {
            typedef bf16x4_t __attribute__((address_space(3))) * lds4p;
#pragma unroll
            for (int ks = 0; ks < BK / 32; ks++) {
                bf16x8_t fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; i++) {
                    const int r0 = wm * WM + i * 16;
                    if (!AKM) fa[i] = *reinterpret_cast<const bf16x8_t*>(&lds_a[(r0 + fr) * (BK + PAD) + ks * 32 + fg * 8]);
                    else {
                        const int q = fr >> 2, pp = fr & 3;
                        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (lds4p)(&lds_a[(ks * 32 + fg * 8 + q) * (BM + PAD) + r0 + 4 * pp]));
                        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (lds4p)(&lds_a[(ks * 32 + fg * 8 + 4 + q) * (BM + PAD) + r0 + 4 * pp]));
                        fa[i] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    }
                }
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    const int r0 = wn * WN + j * 16;
                    if (!BKM) fb[j] = *reinterpret_cast<const bf16x8_t*>(&lds_b[(r0 + fr) * (BK + PAD) + ks * 32 + fg * 8]);
                    else {
                        const int q = fr >> 2, pp = fr & 3;
                        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (lds4p)(&lds_b[(ks * 32 + fg * 8 + q) * (BN + PAD) + r0 + 4 * pp]));
                        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (lds4p)(&lds_b[(ks * 32 + fg * 8 + 4 + q) * (BN + PAD) + r0 + 4 * pp]));
                        fb[j] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < BK / 4; ks++) {
                float fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; i++) fa[i] = lds_a[(wm * WM + i * 16 + fr) * (BK + PAD) + ks * 4 + fg];
#pragma unroll
                for (int j = 0; j < TN; j++) fb[j] = lds_b[(wn * WN + j * 16 + fr) * (BK + PAD) + ks * 4 + fg];
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
        if (more) { stash(); __syncthreads(); }
    }

    // ---- epilogue: lane owns columns n..n+3 of row m for each (i,j) tile
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const long m = m0 + wm * WM + i * 16 + fr;
        if (m >= p.M) continue;
        const float rscale = p.rowscale ? p.rowscale[m / p.rows_per_scale] : 1.f;
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const long n = n0 + wn * WN + j * 16 + fg * 4;
            if (n >= p.N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            gemm_epilogue4<TO, FAST>(p, v, m, n, rscale, blockIdx.z);
        }
    }
}

// split-K second stage: out[m][n] = sum_z partial[z][m][n] (+ bias), fixed order
template <typename TO>
__global__ void splitk_reduce_k(const float* __restrict__ partial, int splits, long MN, int N, long ldo,
                                const float* __restrict__ bias, TO* __restrict__ out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= MN) return;
    float s = 0.f;
    for (int z = 0; z < splits; z++) s += partial[(long)z * MN + idx];
    const long m = idx / N; const int n = (int)(idx % N);
    if (bias) s += bias[n];
    out[m * ldo + n] = from_f<TO>(s);
}

template <typename T, typename TO, int BM, int BN, bool FAST>
static void launch_layout2(const GemmP& p, int akm, int bkm, dim3 grid, hipStream_t st) {
    if (!akm && !bkm) hipLaunchKernelGGL((gemm_k<T, TO, BM, BN, false, false, FAST>), grid, dim3(256), 0, st, p);
    else if (!akm && bkm) hipLaunchKernelGGL((gemm_k<T, TO, BM, BN, false, true, FAST>), grid, dim3(256), 0, st, p);
    else if (akm && bkm) hipLaunchKernelGGL((gemm_k<T, TO, BM, BN, true, true, FAST>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_k<T, TO, BM, BN, true, false, FAST>), grid, dim3(256), 0, st, p);
}
template <typename T, typename TO, int BM, int BN>
static void launch_layout(const GemmP& p, int akm, int bkm, dim3 grid, hipStream_t st) {
    // FAST: every operand row is 16-byte aligned and its contiguous extent is a whole number of 16-byte vectors,
    // and the output family can be accessed 4 elements at a time -> no scalar paths are compiled in
    constexpr int EPV = 16 / (int)sizeof(T);
    const long a_ext = akm ? p.M : p.K, b_ext = bkm ? p.N : p.K;
    const bool fast = p.a_vec && p.b_vec && a_ext % EPV == 0 && b_ext % EPV == 0 && (p.partial || (p.o_vec && p.N % 4 == 0));
    if (fast) launch_layout2<T, TO, BM, BN, true>(p, akm, bkm, grid, st);
    else launch_layout2<T, TO, BM, BN, false>(p, akm, bkm, grid, st);
}

template <typename T, typename TO>
static void launch_tile(GemmP& p, int akm, int bkm, int splits, hipStream_t st) {
    // big tile when it still fills the chip, else 64x64
    const long t128 = cdiv(p.M, 128) * cdiv(p.N, 128) * splits;
    if (t128 >= 256) {
        p.tiles_m = (int)cdiv(p.M, 128); p.tiles_n = (int)cdiv(p.N, 128);
        launch_layout<T, TO, 128, 128>(p, akm, bkm, dim3((unsigned)(p.tiles_m * p.tiles_n), 1, (unsigned)splits), st);
    } else {
        p.tiles_m = (int)cdiv(p.M, 64); p.tiles_n = (int)cdiv(p.N, 64);
        launch_layout<T, TO, 64, 64>(p, akm, bkm, dim3((unsigned)(p.tiles_m * p.tiles_n), 1, (unsigned)splits), st);
    }
}

bool gemm_big_applicable(const GemmP& p, int dtype, int a_kmajor, int b_kmajor, int splits);
int gemm_big_launch(GemmP& p, int out_dtype, hipStream_t st, void* ws, size_t ws_bytes);
bool gemm_big_tn_applicable(const GemmP& p, int dtype, int a_kmajor, int b_kmajor, int splits);
int gemm_big_tn_launch(GemmP& p, int out_dtype, hipStream_t st, void* ws, size_t ws_bytes);

extern "C" int mm_gemm(const mm_gemm_args* a, void* stream) {
    MM_CHECK(a && a->A && a->B && a->out, "gemm: null pointer");
    MM_CHECK(a->M > 0 && a->N > 0 && a->K > 0, "gemm: empty shape M=%lld N=%lld K=%lld", (long long)a->M, (long long)a->N, (long long)a->K);
    MM_CHECK(a->M < (1LL << 31) && a->N < (1LL << 31) && a->K < (1LL << 31), "gemm: dimension too large");
    MM_CHECK(a->dtype == MM_F32 || a->dtype == MM_BF16, "gemm: bad dtype %d", a->dtype);
    MM_CHECK(a->out_dtype == MM_F32 || a->out_dtype == MM_BF16, "gemm: bad out_dtype %d", a->out_dtype);
    MM_CHECK(a->act >= 0 && a->act <= 2, "gemm: bad act %d", a->act);
    MM_CHECK(a->lda >= (a->a_kmajor ? a->M : a->K) && a->ldb >= (a->b_kmajor ? a->N : a->K) && a->ldo >= a->N,
             "gemm: leading dimension smaller than the row length");
    MM_CHECK(!a->rowscale || a->rows_per_scale > 0, "gemm: rowscale needs rows_per_scale > 0");
    const int esz = a->dtype == MM_BF16 ? 2 : 4;
    const int epv = 16 / esz;
    GemmP p{};
    p.A = a->A; p.B = a->B; p.out = a->out;
    p.M = (int)a->M; p.N = (int)a->N; p.K = (int)a->K;
    p.lda = a->lda; p.ldb = a->ldb; p.ldo = a->ldo;
    p.bias = a->bias; p.pre = a->pre; p.mul = a->mul; p.act = a->act;
    p.gamma = a->gamma; p.rowscale = a->rowscale; p.rows_per_scale = (int)a->rows_per_scale;
    p.res = a->res;
    p.a_vec = (a->lda % epv == 0) && (((uintptr_t)a->A) % 16 == 0);
    p.b_vec = (a->ldb % epv == 0) && (((uintptr_t)a->B) % 16 == 0);
    const int osz = a->out_dtype == MM_BF16 ? 2 : 4;
    auto al = [&](const void* q) { return q == nullptr || ((uintptr_t)q) % (4 * osz) == 0; };
    p.o_vec = (a->ldo % 4 == 0) && al(a->out) && al(a->pre) && al(a->mul) && al(a->res);
    int splits = a->split_k > 1 ? a->split_k : 1;
    const int bk = a->dtype == MM_BF16 ? 64 : 16;
    if (splits > 1) {
        MM_CHECK(!a->pre && !a->mul && !a->act && !a->gamma && !a->rowscale && !a->res, "gemm: split_k supports only the bias epilogue");
        MM_CHECK(a->workspace && a->workspace_bytes >= (size_t)splits * a->M * a->N * sizeof(float), "gemm: split_k workspace too small");
        int64_t per = cdiv(cdiv(a->K, splits), bk) * bk;
        splits = (int)cdiv(a->K, per);
        p.k_per_split = (int)per;
        p.partial = splits > 1 ? (float*)a->workspace : nullptr;
    } else {
        p.k_per_split = p.K;
    }
    hipStream_t st = (hipStream_t)stream;
    static const bool no256 = getenv("MM_GEMM_NO256") != nullptr;          // A/B switch for benchmarks
    // split_k <= 1 with a workspace given: the large-tile path may split K on its own for skinny, deep-K problems
    if (!no256 && gemm_big_applicable(p, a->dtype, a->a_kmajor, a->b_kmajor, splits))
        return gemm_big_launch(p, a->out_dtype, st, splits <= 1 ? a->workspace : nullptr, splits <= 1 ? a->workspace_bytes : 0);
    // weight-gradient form (both operands K-major): the large-tile kernel with its own K split when the caller left one open
    if (!no256 && !a->pre && !a->mul && !a->act && !a->gamma && !a->rowscale && !a->res &&
        gemm_big_tn_applicable(p, a->dtype, a->a_kmajor, a->b_kmajor, splits))
        return gemm_big_tn_launch(p, a->out_dtype, st, a->workspace, a->workspace_bytes);
    if (a->dtype == MM_BF16) {
        if (a->out_dtype == MM_BF16) launch_tile<bf16, bf16>(p, a->a_kmajor, a->b_kmajor, splits, st);
        else launch_tile<bf16, float>(p, a->a_kmajor, a->b_kmajor, splits, st);
    } else {
        MM_CHECK(a->out_dtype == MM_F32, "gemm: f32 operands need an f32 output");
        launch_tile<float, float>(p, a->a_kmajor, a->b_kmajor, splits, st);
    }
    MM_LAUNCH_CHECK();
    if (p.partial) {
        const long MN = (long)a->M * a->N;
        const dim3 grid((unsigned)cdiv(MN, 256));
        if (a->out_dtype == MM_BF16)
            hipLaunchKernelGGL(splitk_reduce_k<bf16>, grid, dim3(256), 0, st, p.partial, splits, MN, p.N, p.ldo, p.bias, (bf16*)p.out);
        else
            hipLaunchKernelGGL(splitk_reduce_k<float>, grid, dim3(256), 0, st, p.partial, splits, MN, p.N, p.ldo, p.bias, (float*)p.out);
        MM_LAUNCH_CHECK();
    }
    return MM_OK;
}

// ------------------------------------------------------------------------------------------------
// column sums (bias gradients): two-stage, fixed order.  workspace [nchunks][N] floats.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void colsum_stage1_k(const T* __restrict__ x, float* __restrict__ part, int M, int N, long ld, int rows_per_chunk) {
    const int n = blockIdx.x * 256 + threadIdx.x;                 // scalar form: any N / ld / alignment
    if (n >= N) return;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    float s = 0.f;
    for (int r = r0; r < r1; r++) s += to_f(x[(long)r * ld + n]);
    part[(long)blockIdx.y * N + n] = s;
}
// vector form: a block covers 32*VEC columns x one row chunk; 8 row groups of 32 lanes, each lane 16 B per row
template <typename T, int VEC>
__global__ __launch_bounds__(256) void colsum_stage1v_k(const T* __restrict__ x, float* __restrict__ part, int M, int N, long ld,
                                                        int rows_per_chunk) {
    __shared__ float red[8][32 * VEC + 1];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int n0 = (blockIdx.x * 32 + cl) * VEC;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    float s[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) s[j] = 0.f;
    if (n0 < N) {
#pragma unroll 4
        for (int r = r0 + rg; r < r1; r += 8) {
            float v[VEC];
            VecIO<T, VEC>::load(x + (long)r * ld + n0, v);
#pragma unroll
            for (int j = 0; j < VEC; j++) s[j] += v[j];
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; j++) red[rg][cl * VEC + j] = s[j];
    __syncthreads();
    for (int c = threadIdx.x; c < 32 * VEC; c += 256) {
        const int n = blockIdx.x * 32 * VEC + c;
        if (n < N) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 8; g++) t += red[g][c];
            part[(long)blockIdx.y * N + n] = t;
        }
    }
}
// out[n] = sum_c part[c][n]: a block covers 64 columns, 4 chunk groups
__global__ __launch_bounds__(256) void colsum_stage2_k(const float* __restrict__ part, float* __restrict__ out, int nchunks, int N) {
    // 16 columns x 16 chunk groups per block: with up to 160 chunks a 4-way split left 40 dependent loads per thread (15-25 us for
    // a few hundred KB, on the backward's critical path after every relu^2 / LayerNorm backward); fixed summation order
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int n = blockIdx.x * 16 + cl;
    float s = 0.f;
    if (n < N)
        for (int c = g; c < nchunks; c += 16) s += part[(long)c * N + n];
    red[g][cl] = s;
    __syncthreads();
    if (g == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) t += red[k][cl];
        out[n] = t;
    }
}
static int colsum_chunks(int64_t M) { int64_t c = cdiv(M, 64); if (c > 160) c = 160; if (c < 1) c = 1; return (int)c; }
extern "C" size_t mm_colsum_workspace(int64_t M, int64_t N) {
    if (M <= 0 || N <= 0) return 0;
    return (size_t)colsum_chunks(M) * (size_t)N * sizeof(float);
}
extern "C" int mm_colsum(const void* x, float* out, void* workspace, size_t workspace_bytes, int64_t M, int64_t N,
                         int64_t ld, int dtype, void* stream) {
    MM_CHECK(x && out && workspace, "colsum: null pointer");
    MM_CHECK(M > 0 && N > 0 && ld >= N, "colsum: bad shape");
    MM_CHECK(workspace_bytes >= mm_colsum_workspace(M, N), "colsum: workspace too small");
    const int chunks = colsum_chunks(M);
    const int rpc = (int)cdiv(M, chunks);
    const int nchunks = (int)cdiv(M, rpc);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)cdiv(N, 256), (unsigned)nchunks);
    const int esz = dtype == MM_BF16 ? 2 : 4;
    const bool vec_ok = (N % (16 / esz)) == 0 && ((ld * esz) % 16) == 0 && ((uintptr_t)x % 16) == 0;
    if (dtype == MM_F32) {
        if (vec_ok) hipLaunchKernelGGL((colsum_stage1v_k<float, 4>), dim3((unsigned)cdiv(N, 128), (unsigned)nchunks), dim3(256), 0, st, (const float*)x, (float*)workspace, (int)M, (int)N, (long)ld, rpc);
        else hipLaunchKernelGGL(colsum_stage1_k<float>, grid, dim3(256), 0, st, (const float*)x, (float*)workspace, (int)M, (int)N, (long)ld, rpc);
    } else if (dtype == MM_BF16) {
        if (vec_ok) hipLaunchKernelGGL((colsum_stage1v_k<bf16, 8>), dim3((unsigned)cdiv(N, 256), (unsigned)nchunks), dim3(256), 0, st, (const bf16*)x, (float*)workspace, (int)M, (int)N, (long)ld, rpc);
        else hipLaunchKernelGGL(colsum_stage1_k<bf16>, grid, dim3(256), 0, st, (const bf16*)x, (float*)workspace, (int)M, (int)N, (long)ld, rpc);
    } else MM_FAIL(MM_ERR_ARG, "colsum: bad dtype %d", dtype);
    MM_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_stage2_k, dim3((unsigned)cdiv(N, 16)), dim3(256), 0, st, (const float*)workspace, out, nchunks, (int)N);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

// ------------------------------------------------------------------------------------------------
// mm_colsum for callers that cannot allocate or hand over a workspace (code under HIP graph capture, serving loops):
// the [nchunks][N] partial sums live in a file-scope device array.  Each GPU holds its own copy of the array at its own
// address: the address is looked up once per device (hipGetSymbolAddress, with that device current) and that pointer --
// never the array's name, which in host code is only a stand-in -- is what the kernels receive.
// ------------------------------------------------------------------------------------------------
#define MM_COLSUM_NOALLOC_MAX_N 8192                       /* 4 * d_model of the widest config (E = 2048) */
#define MM_COLSUM_NOALLOC_MAX_CHUNKS 160                   /* colsum_chunks()'s cap */
static __device__ __align__(256) float scratch[MM_COLSUM_NOALLOC_MAX_CHUNKS * MM_COLSUM_NOALLOC_MAX_N];   // 5 MiB per GPU
#define MM_MAX_DEVICES 64
static float* g_scratch_ptr[MM_MAX_DEVICES];               // per device; written once, under g_scratch_mu
static std::mutex g_scratch_mu;

static int colsum_scratch_for_current_device(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) MM_FAIL(MM_ERR_LAUNCH, "colsum_noalloc: hipGetDevice failed: %s", hipGetErrorString(e));
    MM_CHECK(dev >= 0 && dev < MM_MAX_DEVICES, "colsum_noalloc: device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        float* ptr = nullptr;
        e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr)
            MM_FAIL(MM_ERR_LAUNCH, "colsum_noalloc: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return MM_OK;
}

extern "C" int mm_colsum_noalloc(const void* x, float* out, int64_t M, int64_t N, int64_t ld, int dtype, void* stream) {
    MM_CHECK(x && out, "colsum_noalloc: null pointer");
    MM_CHECK(M > 0 && N > 0 && ld >= N, "colsum_noalloc: bad shape");
    MM_CHECK(N <= MM_COLSUM_NOALLOC_MAX_N,
             "colsum_noalloc: N=%lld exceeds the built-in scratch (max N = %d); use mm_colsum with a caller workspace",
             (long long)N, MM_COLSUM_NOALLOC_MAX_N);
    MM_CHECK(mm_colsum_workspace(M, N) <= sizeof(float) * MM_COLSUM_NOALLOC_MAX_CHUNKS * MM_COLSUM_NOALLOC_MAX_N,
             "colsum_noalloc: shape needs more than the built-in scratch");
    float* ptr = nullptr;
    int rc = colsum_scratch_for_current_device(&ptr);
    if (rc != MM_OK) return rc;
    // stage 1 writes every partial that stage 2 reads ([nchunks][N], dense), so nothing has to start at zero and the
    // previous call's contents are harmless; two calls that overlap on one GPU (two streams) race on the array.
    return mm_colsum(x, out, ptr, sizeof(float) * MM_COLSUM_NOALLOC_MAX_CHUNKS * MM_COLSUM_NOALLOC_MAX_N, M, N, ld, dtype, stream);
}

// ------------------------------------------------------------------------------------------------
// relu^2 backward fused with the bias gradient of the Linear in front of it (models/transformer_decoder.py:161):
//   dh = da * 2*max(h,0)  (dh may alias da),   colsum[n] = sum_m dh[m][n]  (of the values as stored)
// Same block shape and two-stage fixed-order reduction as mm_colsum: the [M][N] tensor is read once instead of twice.
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(256) void relu2_bwd_colsum_k(const T* __restrict__ h, const T* da, T* dh, float* __restrict__ part,
                                                          int M, int N, int rows_per_chunk) {
    __shared__ float red[8][32 * VEC + 1];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int n0 = (blockIdx.x * 32 + cl) * VEC;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    float s[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) s[j] = 0.f;
    if (n0 < N) {
#pragma unroll 4
        for (int r = r0 + rg; r < r1; r += 8) {
            float hv[VEC], gv[VEC];
            VecIO<T, VEC>::load(h + (long)r * N + n0, hv);
            VecIO<T, VEC>::load(da + (long)r * N + n0, gv);
#pragma unroll
            for (int j = 0; j < VEC; j++) {
                gv[j] = to_f(from_f<T>(gv[j] * (2.f * fmaxf(hv[j], 0.f))));
                s[j] += gv[j];
            }
            VecIO<T, VEC>::store(dh + (long)r * N + n0, gv);
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; j++) red[rg][cl * VEC + j] = s[j];
    __syncthreads();
    for (int c = threadIdx.x; c < 32 * VEC; c += 256) {
        const int n = blockIdx.x * 32 * VEC + c;
        if (n < N) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 8; g++) t += red[g][c];
            part[(long)blockIdx.y * N + n] = t;
        }
    }
}
extern "C" int mm_relu2_bwd_colsum(const void* h, const void* da, void* dh, float* colsum, void* workspace,
                                   size_t workspace_bytes, int64_t M, int64_t N, int dtype, void* stream) {
    MM_CHECK(h && da && dh && colsum && workspace, "relu2_bwd_colsum: null pointer");
    MM_CHECK(M > 0 && N > 0, "relu2_bwd_colsum: bad shape");
    MM_CHECK(workspace_bytes >= mm_colsum_workspace(M, N), "relu2_bwd_colsum: workspace too small");
    const int esz = dtype == MM_BF16 ? 2 : 4;
    MM_CHECK(dtype == MM_BF16 || dtype == MM_F32, "relu2_bwd_colsum: bad dtype %d", dtype);
    MM_CHECK((N % (16 / esz)) == 0 && ((((uintptr_t)h) | ((uintptr_t)da) | ((uintptr_t)dh)) % 16) == 0,
             "relu2_bwd_colsum: needs 16-byte aligned rows (N=%lld)", (long long)N);
    const int chunks = colsum_chunks(M);
    const int rpc = (int)cdiv(M, chunks);
    const int nchunks = (int)cdiv(M, rpc);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MM_F32)
        hipLaunchKernelGGL((relu2_bwd_colsum_k<float, 4>), dim3((unsigned)cdiv(N, 128), (unsigned)nchunks), dim3(256), 0, st,
                           (const float*)h, (const float*)da, (float*)dh, (float*)workspace, (int)M, (int)N, rpc);
    else
        hipLaunchKernelGGL((relu2_bwd_colsum_k<bf16, 8>), dim3((unsigned)cdiv(N, 256), (unsigned)nchunks), dim3(256), 0, st,
                           (const bf16*)h, (const bf16*)da, (bf16*)dh, (float*)workspace, (int)M, (int)N, rpc);
    MM_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_stage2_k, dim3((unsigned)cdiv(N, 16)), dim3(256), 0, st, (const float*)workspace, colsum, nchunks, (int)N);
    MM_LAUNCH_CHECK();
    return MM_OK;
}
