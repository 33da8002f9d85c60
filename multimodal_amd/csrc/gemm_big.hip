// Large-tile bf16 GEMM for the decoder projections:  out[M][N] = epi( A[M][K] . B[N][K]^T ), both operands K-contiguous.
// One 512-thread workgroup (8 waves as 2(M) x 4(N)) owns a BM x BN output tile, BM = 32*TM, BN = 64*TN (a wave
// accumulates TM x TN MFMA 16x16x32 tiles); the K loop moves 64 columns a step.
//   * operands go global -> LDS directly (global_load_lds, 16 bytes per lane, no VGPR staging), two LDS stages:
//     the loads of tile k+1 are in flight while tile k is multiplied; one barrier per K tile;
//   * LDS image: rows of 128 bytes, the 16-byte chunk c of row r sits in slot c ^ ((r>>1)&7).  A ds_read_b128
//     wave-instruction is served in 16-lane groups (16 rows, one chunk): rows r, r^1 fall in the two halves of a
//     256-byte bank row and the 8 row pairs in 8 different slots -> conflict-free.  The DMA writes LDS linearly
//     (wave base + lane*16), so the permutation is applied to each lane's SOURCE address;
//   * rows past M / N are clamped to the last row on load (their products are never stored); K must be a
//     multiple of 64 (the caller falls back to gemm_k otherwise);
//   * epilogue through LDS: accumulators are transposed 32 rows a pass through a per-wave window and leave in a
//     rolled loop, 8 consecutive columns per lane (16-byte global accesses, compact code);
//   * the tile shape is chosen per problem from a small set by wave quantisation (tiles vs 256 CUs).
// Tile order: XCD-aware bijective remap, N-fastest, so the 8 XCD-private L2s each keep a band of A rows.
#include "common.h"
#include "gemm_epi.h"
#include <stdlib.h>
#include <utility>
#include <algorithm>
#include <type_traits>

#ifndef GEMM_BIG_STAGGER
#define GEMM_BIG_STAGGER 1
#endif
// Lab build (benchmarks/gemm_lab.hip includes this file with GEMM_LAB defined): one tile shape, bf16 only, and cycle stamps
// of workgroup 0 kept in LDS behind the operand stages and dumped at the end.  In the library STAMP() is empty.
#ifdef GEMM_LAB
#ifndef GEMM_LAB_TTM
#define GEMM_LAB_TTM 8
#define GEMM_LAB_TTN 4
#endif
#define LAB_LDS 8192
__device__ unsigned long long g_stamp[8][16][8];
#define STAMP(kt_, slot_) do { if (blockIdx.x == 0 && (kt_) >= GEMM_LAB_KT0 && (kt_) < GEMM_LAB_KT0 + 16 && lane == 0) \
    reinterpret_cast<unsigned long long*>(smem + LAB_BASE)[(wave * 16 + (kt_) - GEMM_LAB_KT0) * 8 + (slot_)] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMP_DUMP() do { __syncthreads(); if (blockIdx.x == 0) for (int i_ = threadIdx.x; i_ < 1024; i_ += blockDim.x) \
    (&g_stamp[0][0][0])[i_] = reinterpret_cast<unsigned long long*>(smem + LAB_BASE)[i_]; __syncthreads(); } while (0)
#else
#define LAB_LDS 0
#define STAMP(kt_, slot_) do { } while (0)
#define STAMP_DUMP() do { } while (0)
#endif
namespace {
constexpr bool STAGGER = GEMM_BIG_STAGGER != 0;
constexpr int BK = 64;
constexpr int kNumCU = 256;
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

// 8 consecutive output-type elements kept packed in registers until used
template <typename TO> struct Raw8;
template <> struct Raw8<bf16> {
    mm_u4_t r;
    __device__ __forceinline__ void load(const bf16* q) { r = *reinterpret_cast<const mm_u4_t*>(q); }
    __device__ __forceinline__ void unpack(float (&v)[8]) const { VecIO<bf16, 8>::unpack(r, v); }
};
template <> struct Raw8<float> {
    mm_f4_t a, b;
    __device__ __forceinline__ void load(const float* q) { a = *reinterpret_cast<const mm_f4_t*>(q); b = *reinterpret_cast<const mm_f4_t*>(q + 4); }
    __device__ __forceinline__ void unpack(float (&v)[8]) const {
#pragma unroll
        for (int e = 0; e < 4; e++) { v[e] = a[e]; v[4 + e] = b[e]; }
    }
};

template <int TM, int TN, int WN> struct BigTile {
    static constexpr int NW = 2 * WN;                                  // waves: 2 along M x WN along N
    static constexpr int BM = 32 * TM, BN = 16 * TN * WN;
    static constexpr int STAGE_BYTES = (BM + BN) * BK * 2;
    static constexpr int EPI_ROW = TN * 64 + 16;                       // padded f32 row of a wave's window (bytes)
    static constexpr int EPI_WAVE = 16 * EPI_ROW;                      // 16 rows a pass; 8 windows fit in ONE stage
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
    static constexpr int PIECES = (BM + BN) / 8;                       // 1 KiB DMA pieces (8 rows) per stage
    static constexpr int PPW = (PIECES + NW - 1) / NW;                 // pieces per wave (the last may be absent)
};

// ---- epilogue through LDS: a wave transposes its TM x TN accumulator tiles 16 rows a pass through a private window
// `wl` (padded f32 rows of EPI_ROW bytes) and leaves 8 consecutive columns per lane.  Fast form (16-byte accessible out
// family, N % 8 == 0, at most one of mul/res, no gamma/rowscale): the bias is read once, and the mul/res operand of
// pass p+1 is requested before pass p is processed, so no global-load latency sits between the passes.
// mw, nw: first row / column of the wave's tile.
template <typename TO, int TM, int TN, int EPI_ROW>
__device__ __forceinline__ void big_epilogue(const GemmP& p, f32x4_t (&acc)[TM][TN], unsigned char* wl, long mw, long nw, int lane) {
    const int fr = lane & 15, fg = lane >> 4;
    constexpr int LPR = TN * 2;                        // lanes per row (8 columns each)
    if (p.partial) {                                    // split-K: raw accumulators into this split's slab [M][N] (f32)
        float* slab = p.partial + (long)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int pass = 0; pass < TM; pass++) {
#pragma unroll
            for (int j = 0; j < TN; j++)
                *reinterpret_cast<f32x4_t*>(wl + fr * EPI_ROW + ((j * 4 + fg) << 4)) = acc[pass][j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
            for (int idx = lane; idx < 16 * TN * 4; idx += 64) {
                const int row = idx / (TN * 4), c4 = idx % (TN * 4);
                const long m = mw + pass * 16 + row, n = nw + c4 * 4;
                if (m < p.M && n < p.N) {
                    const f32x4_t v = *reinterpret_cast<const f32x4_t*>(wl + row * EPI_ROW + c4 * 16);
                    if (n + 4 <= p.N && (p.N & 3) == 0) *reinterpret_cast<f32x4_t*>(slab + m * p.N + n) = v;
                    else for (int e = 0; e < 4 && n + e < p.N; e++) slab[m * p.N + n + e] = v[e];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        return;
    }
    const bool fast = p.o_vec && (p.N & 7) == 0 && !(p.mul && p.res) && !p.gamma && !p.rowscale;   // block-uniform
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    if (fast) {
        // lane -> fixed column group (so the bias is read once), rows r0 + RPI*it of the pass
        constexpr int CG = LPR <= 8 ? 8 : (LPR <= 16 ? 16 : 32), RPI = 64 / CG, NIT2 = 16 / RPI;
        const TO* extra = (const TO*)(p.mul ? p.mul : p.res);
        const int c8 = lane % CG, r0 = lane / CG;
        const long n = nw + c8 * 8;
        const bool on = c8 < LPR && n < p.N;
        float bias[8];
#pragma unroll
        for (int e = 0; e < 8; e++) bias[e] = (on && p.bias) ? p.bias[n + e] : 0.f;
        Raw8<TO> nxt[NIT2];
#pragma unroll
        for (int it = 0; it < NIT2; it++) {
            nxt[it] = Raw8<TO>{};
            const long m = mw + r0 + RPI * it;
            if (extra && on && m < p.M) nxt[it].load(extra + m * p.ldo + n);
        }
#pragma unroll
        for (int pass = 0; pass < TM; pass++) {
            Raw8<TO> cur[NIT2];
#pragma unroll
            for (int it = 0; it < NIT2; it++) {
                cur[it] = nxt[it];
                const long m = mw + (pass + 1) * 16 + r0 + RPI * it;
                if (pass + 1 < TM && extra && on && m < p.M) nxt[it].load(extra + m * p.ldo + n);
            }
#pragma unroll
            for (int j = 0; j < TN; j++)
                *reinterpret_cast<f32x4_t*>(wl + fr * EPI_ROW + ((j * 4 + fg) << 4)) = acc[pass][j];
            wave_sync();
#pragma unroll
            for (int it = 0; it < NIT2; it++) {
                const int row = r0 + RPI * it;
                const long m = mw + pass * 16 + row;
                if (on && m < p.M) {
                    const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(wl + row * EPI_ROW + c8 * 32);
                    const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(wl + row * EPI_ROW + c8 * 32 + 16);
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    float ex[8];
                    cur[it].unpack(ex);
                    const long off = m * p.ldo + n;
#pragma unroll
                    for (int e = 0; e < 8; e++) v[e] += bias[e];
                    if (p.pre) io8_store<TO>((TO*)p.pre + off, v);
                    if (p.mul) {
#pragma unroll
                        for (int e = 0; e < 8; e++) v[e] *= act_bwd<TO>(p.act, ex[e]);
                    } else if (p.act) {
#pragma unroll
                        for (int e = 0; e < 8; e++) v[e] = act_fwd<TO>(p.act, v[e]);
                    }
                    if (p.res) {
#pragma unroll
                        for (int e = 0; e < 8; e++) v[e] += ex[e];
                    }
                    io8_store<TO>((TO*)p.out + off, v);
                }
            }
            wave_sync();
        }
    } else {
        const bool vec_all = p.o_vec != 0;
#pragma unroll
        for (int pass = 0; pass < TM; pass++) {
#pragma unroll
            for (int j = 0; j < TN; j++)
                *reinterpret_cast<f32x4_t*>(wl + fr * EPI_ROW + ((j * 4 + fg) << 4)) = acc[pass][j];
            wave_sync();
#pragma unroll 1
            for (int idx = lane; idx < 16 * LPR; idx += 64) {
                const int row = idx / LPR, c8 = idx % LPR;
                const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(wl + row * EPI_ROW + c8 * 32);
                const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(wl + row * EPI_ROW + c8 * 32 + 16);
                const long m = mw + pass * 16 + row, n = nw + c8 * 8;
                if (m < p.M && n < p.N) {
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    gemm_epilogue8<TO>(p, v, m, n, vec_all && n + 8 <= p.N);
                }
            }
            wave_sync();
        }
    }
}

template <typename TO, int TM, int TN, int WN>
__global__ __launch_bounds__(128 * WN, 1) void gemm_big_k(const GemmP p) {
    typedef BigTile<TM, TN, WN> G;
    constexpr int BM = G::BM, BN = G::BN;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    // split-K (skinny problems with a deep K): blockIdx.y owns K steps [kt0, kt0 + nk) and leaves a raw f32 slab
    const int kt0 = p.partial ? blockIdx.y * (p.k_per_split / BK) : 0;
    const int nk = p.partial ? min(p.K / BK - kt0, p.k_per_split / BK) : p.K / BK;
    const int nwg = p.tiles_m * p.tiles_n;

    // ---- fragment read offsets (bytes inside a stage)
    const int fr = lane & 15, fg = lane >> 4;
    int offA[2], offB[2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
        const int slot = ((ks * 4 + fg) ^ (fr >> 1)) << 4;
        offA[ks] = (wr * (BM / 2) + fr) * 128 + slot;
        offB[ks] = (BM + wc * (BN / WN) + fr) * 128 + slot;
    }

    // ---- staging: piece q = wave + NW*i covers rows 8q .. 8q+7 of the stacked [A rows ; B rows] image
    // The source of a piece is a scalar base (A or B: a piece never straddles them; advanced by the K step) + a 32-bit lane
    // offset (host: M * lda and N * ldb bytes below 4 GiB), and the piece goes out through inline asm in that form: no vector
    // ALU work and no branch per piece (the builtin took a 64-bit lane address: an add and a select pair per piece).
    unsigned voff[G::PPW];
    auto tile_origin = [&](int id, long& m0, long& n0) {              // XCD-aware bijective remap, N fastest
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
        m0 = (long)(wg / p.tiles_n) * BM; n0 = (long)(wg % p.tiles_n) * BN;
    };
    auto set_sources = [&](long m0, long n0) {
#pragma unroll
        for (int i = 0; i < G::PPW; i++) {
            const int q = wave + G::NW * i;
            const bool isA = q * 8 < BM;                               // wave-uniform (BM % 8 == 0)
            const int lrow = q * 8 + (lane >> 3) - (isA ? 0 : BM);     // operand-local row (the swizzle uses it)
            const int chunk = (lane & 7) ^ ((lrow >> 1) & 7);
            long gr = (isA ? m0 : n0) + lrow;
            const long lim = (isA ? p.M : p.N) - 1;
            if (gr > lim) gr = lim;
            voff[i] = (unsigned)(gr * (isA ? p.lda : p.ldb) * 2 + chunk * 16);
        }
    };
    auto uniform = [](const char* q) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)q), hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)q >> 32));
        return (const char*)(((uintptr_t)hi << 32) | lo);
    };
    auto stage = [&](int kt, int buf) {
        const unsigned lds0 = (unsigned)(uintptr_t)(lds_vp)(smem + buf * G::STAGE_BYTES);
        const char* sA = uniform((const char*)p.A + (long)(kt0 + kt) * BK * 2);
        const char* sB = uniform((const char*)p.B + (long)(kt0 + kt) * BK * 2);
#pragma unroll
        for (int i = 0; i < G::PPW; i++) {
            const int q = wave + G::NW * i;
            if (q < G::PIECES)                                         // wave-uniform
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds0 + q * 1024), "v"(voff[i]),
                             "s"(q * 8 < BM ? sA : sB) : "memory", "m0");
        }
    };
    // ---- one output tile per workgroup (a persistent variant with next-tile prefetch measured no faster and cost
    // ~50 VGPRs: the epilogue's stores are counted by the same vmcnt the next tile's first wait needs)
    long m0, n0;
    tile_origin(blockIdx.x, m0, n0);
    set_sources(m0, n0);
    stage(0, 0);
    {
        f32x4_t acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

        // The two waves that share a SIMD (w and w + NW/2) run half a K step apart: in lockstep all eight waves read their
        // fragments (192 KiB per step: ~0.37 us of LDS time) and THEN all multiply (~1 us), the step time was the sum of the
        // two; staggered, one wave's fragment reads sit under its partner's MFMAs.  The late half keeps the fragments of a
        // tile's second half step in registers across the barrier (read before the stage is handed back to the DMA).
        bf16x8_t fa[TM], fb[TN];
        auto rd = [&](const unsigned char* ls, int ks) {
#pragma unroll
            for (int j = 0; j < TN; j++) fb[j] = *reinterpret_cast<const bf16x8_t*>(ls + offB[ks] + j * 2048);
#pragma unroll
            for (int i = 0; i < TM; i++) fa[i] = *reinterpret_cast<const bf16x8_t*>(ls + offA[ks] + i * 2048);
            // all fragment reads of the half step are issued before its first MFMA (left alone the compiler alternates two
            // reads, a full lgkmcnt(0) wait and ten MFMAs, exposing the LDS latency five times a half step): 4-5 %
            __builtin_amdgcn_sched_barrier(0);
        };
        auto mm = [&]() {
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        };
        const bool late = STAGGER && p.stagger && wave >= G::NW / 2;       // scalar
        // Where the DMA pieces of the next tile are issued matters: a piece holds its wave for 60-190 cycles while the address
        // path takes it (cycle stamps, benchmarks/gemm_lab.hip), 8-9 pieces a third of a step's MFMA time.  Issued by all
        // eight waves right after the barrier the matrix pipe idles meanwhile; the late wave of an 8-wave tile therefore
        // issues its pieces AFTER its held MFMAs, i.e. under the early wave's first MFMAs (which issued its own under these):
        // 4-9 % on the decoder shapes.  (One piece between rows of MFMAs measured slower, as did the 4-wave tiles this way.)
        constexpr bool LATE_DMA_AFTER = WN == 4;
        constexpr int LAB_BASE = G::LDS_BYTES;
        if (!late) {
            for (int kt = 0; kt < nk; kt++) {
                STAMP(kt, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                STAMP(kt, 1);
                __syncthreads();                           // K tile kt landed; everyone is done with the other stage
                STAMP(kt, 2);
                if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
                STAMP(kt, 3);
                const unsigned char* ls = smem + (kt & 1) * G::STAGE_BYTES;
                rd(ls, 0); mm();
                STAMP(kt, 4);
                rd(ls, 1); mm();
                STAMP(kt, 5);
            }
        } else {
            for (int kt = 0; kt < nk; kt++) {
                STAMP(kt, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                STAMP(kt, 1);
                __syncthreads();
                STAMP(kt, 2);
                const unsigned char* ls = smem + (kt & 1) * G::STAGE_BYTES;
                if (!LATE_DMA_AFTER && kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
                if (kt > 0) mm();                          // second half step of tile kt-1 (fragments held in registers)
                STAMP(kt, 3);
                if (LATE_DMA_AFTER && kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
                STAMP(kt, 4);
                rd(ls, 0); mm();
                STAMP(kt, 5);
                rd(ls, 1);                                 // held across the next barrier
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // in registers before the stage can be overwritten
                STAMP(kt, 6);
            }
            if (nk > 0) mm();
        }
        STAMP_DUMP();
        __syncthreads();                                   // every wave is done with both operand stages
        big_epilogue<TO, TM, TN, G::EPI_ROW>(p, acc, smem + G::STAGE_BYTES + wave * G::EPI_WAVE, m0 + wr * (BM / 2), n0 + wc * (BN / WN), lane);
    }
}

#ifdef GEMM_LAB   // experimental main loops live outside the product source: benchmarks/gemm_lab_loops.inc (lab builds only)
#include "../../benchmarks/gemm_lab_loops.inc"
#endif  // GEMM_LAB

// split-K second stage with the FULL fused epilogue: sum the slabs in a fixed order, then bias / act / mul / gamma / rowscale /
// residual exactly as the one-pass kernels do (4 consecutive columns per thread)
template <typename TO>
__global__ __launch_bounds__(256) void big_splitk_epilogue_k(GemmP p, int splits) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    const long nq = (p.N + 3) / 4;
    if (q >= (long)p.M * nq) return;
    const long m = q / nq, n = (q % nq) * 4;
    const float* src = p.partial + m * p.N + n;
    const long slab = (long)p.M * p.N;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const bool vec = (p.N & 3) == 0;
    for (int z = 0; z < splits; z++) {
        if (vec) { const f32x4_t t = *reinterpret_cast<const f32x4_t*>(src + z * slab); v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3]; }
        else for (int e = 0; e < 4 && n + e < p.N; e++) v[e] += src[z * slab + e];
    }
    const float rscale = p.rowscale ? p.rowscale[m / p.rows_per_scale] : 1.f;
    p.partial = nullptr;
    gemm_epilogue4<TO, false>(p, v, m, n, rscale, 0);
}

template <typename TO, int TM, int TN, int WN>
int launch_big(const GemmP& p, hipStream_t st, int splits = 1) {
    typedef BigTile<TM, TN, WN> G;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_big_k<TO, TM, TN, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES + LAB_LDS);
        if (e != hipSuccess) MM_FAIL(MM_ERR_LAUNCH, "gemm_big: cannot raise dynamic LDS to %d bytes: %s", G::LDS_BYTES, hipGetErrorString(e));
        attr_set = true;
    }
    const int tiles = p.tiles_m * p.tiles_n;
    const dim3 grid((unsigned)tiles, (unsigned)splits), block(128 * WN);
#ifdef GEMM_LAB
    // main-loop variant of the 8-wave shapes (MM_GEMM_BIG_LOOP): two (default) | pipe | ring | reg
    static const char* loop_env = getenv("MM_GEMM_BIG_LOOP");
    static const int loop = !loop_env ? 1 : (loop_env[0] == 'p' && loop_env[1] == '3' ? 4 : loop_env[0] == 'p' ? 0 : (loop_env[0] == 'r' && loop_env[1] == 'i' ? 2 : (loop_env[0] == 'r' ? 3 : 1)));
#define LOOP_(KERNEL_, BYTES_)                                                                                                          \
    do {                                                                                                                               \
        static bool attr_ = false;                                                                                                     \
        if (!attr_) {                                                                                                                  \
            hipError_t e = hipFuncSetAttribute((const void*)KERNEL_<TO, TM, TN, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (BYTES_) + LAB_LDS); \
            if (e != hipSuccess) MM_FAIL(MM_ERR_LAUNCH, #KERNEL_ ": cannot raise dynamic LDS to %d bytes: %s", (int)(BYTES_), hipGetErrorString(e)); \
            attr_ = true;                                                                                                              \
        }                                                                                                                              \
        hipLaunchKernelGGL((KERNEL_<TO, TM, TN, WN>), grid, block, (BYTES_) + LAB_LDS, st, p);                                          \
    } while (0)
    if constexpr (WN == 4) {
        const bool off32 = (size_t)p.M * p.lda * 2 < (1ull << 32) && (size_t)p.N * p.ldb * 2 < (1ull << 32);
        if (loop == 0 && off32) LOOP_(gemm_pipe_k, G::LDS_BYTES);
        else if (loop == 4 && off32) { if constexpr (TN == 4) LOOP_(gemm_pipe3_k, (2 * G::BM * 128 + 3 * G::BN * 128)); }
        else if (loop == 2) LOOP_(gemm_ring_k, (RingTile<TM, TN, WN>::LDS_BYTES));
        else if (loop == 3 && off32) LOOP_(gemm_reg_k, G::LDS_BYTES);
        else hipLaunchKernelGGL((gemm_big_k<TO, TM, TN, WN>), grid, block, G::LDS_BYTES + LAB_LDS, st, p);
    } else
#undef LOOP_
#endif
        hipLaunchKernelGGL((gemm_big_k<TO, TM, TN, WN>), grid, block, G::LDS_BYTES + LAB_LDS, st, p);
    MM_LAUNCH_CHECK();
    if (p.partial) {
        const long nq = (long)p.M * ((p.N + 3) / 4);
        hipLaunchKernelGGL(big_splitk_epilogue_k<TO>, dim3((unsigned)cdiv(nq, 256)), dim3(256), 0, st, p, splits);
        MM_LAUNCH_CHECK();
    }
    return MM_OK;
}


// ------------------------------------------------------------------------------------------------
// Weight-gradient form:  out[M][N] = sum_k A[k][M] . B[k][N]  -- BOTH operands K-major (dW = dY^T X: the reduction runs over
// the tokens, the slow index of both tensors).  Same skeleton as gemm_big_k (8 waves 2 x 4, 64-deep K steps, operands global
// -> LDS by DMA into two stages, one barrier per step, epilogue through per-wave LDS windows), with
//   * LDS image [64 k-rows][BM | BN columns]: a DMA piece (1 KiB) is linear in LDS, each lane's SOURCE address is the one its
//     16 bytes belong to; k-rows past K read a zero page, columns past M / N are clamped (never stored);
//   * fragments by ds_read_b64_tr_b16 (hardware transpose: a lane gets 4 consecutive k of one column); the 8 k-rows x 32 bytes
//     of a half-wave read must fall in 8 different 32-byte bank groups: the 32-byte chunk index of a row is XOR-ed with a
//     function of the row (3 bits for a 512-byte row pitch, 2 bits for 640 bytes: the pitch itself already alternates);
//   * deep K, few tiles: K is split over blockIdx.y by the host so that one round of workgroups covers the chip; the raw f32
//     slabs are summed in a fixed order by big_splitk_epilogue_k (deterministic).
// ------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) unsigned g_zero_page[8] = {0, 0, 0, 0, 0, 0, 0, 0};

template <int BX> __device__ __forceinline__ int tn_swz(int row) {          // BX columns: row pitch 2*BX bytes
    if constexpr ((BX / 16) % 8 == 0) return (row & 3) | (((row >> 3) & 1) << 2);
    else { static_assert((BX / 16) % 8 == 4, "unsupported pitch"); return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }
}

template <typename TO, int TM, int TN, int WN>
__global__ __launch_bounds__(128 * WN, 1) void gemm_big_tn_k(const GemmP p) {
    typedef BigTile<TM, TN, WN> G;
    constexpr int BM = G::BM, BN = G::BN;
    constexpr int A_BYTES = BK * BM * 2, PA = 2 * BM, PB = 2 * BN;             // row pitches in bytes
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WN, wc = wave % WN;
    const int kt0 = p.partial ? blockIdx.y * (p.k_per_split / BK) : 0;
    const int ksteps = (p.K + BK - 1) / BK;
    const int nk = p.partial ? min(ksteps - kt0, p.k_per_split / BK) : ksteps;
    const int nwg = p.tiles_m * p.tiles_n;
    long m0, n0;
    {
        const int id = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
        m0 = (long)(wg / p.tiles_n) * BM; n0 = (long)(wg % p.tiles_n) * BN;
    }
    // ---- staging: piece q = wave + NW*i covers bytes [1024 q, 1024 q + 1024) of the stage image [A rows | B rows]; the source is
    // a scalar base (A or B: a piece does not straddle them) + a 32-bit lane offset (host: K * lda and K * ldb bytes below 4 GiB)
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    unsigned voff[G::PPW];
    auto piece_row = [&](int q) {                                             // k-row (inside a stage) of this lane's 16 bytes of piece q
        int off = q * 1024 + lane * 16;
        const bool isA = q * 1024 < A_BYTES;
        if (!isA) off -= A_BYTES;
        return off / (isA ? PA : PB);
    };
#pragma unroll
    for (int i = 0; i < G::PPW; i++) {
        const int q = swave + G::NW * i;
        int off = q * 1024 + lane * 16;
        const bool isA = q * 1024 < A_BYTES;                                  // wave-uniform (A_BYTES % 1024 == 0)
        if (!isA) off -= A_BYTES;
        const int pitch = isA ? PA : PB;
        const int row = off / pitch, c16 = (off % pitch) >> 4;
        const int sw = isA ? tn_swz<BM>(row) : tn_swz<BN>(row);
        const int lc16 = (((c16 >> 1) ^ sw) << 1) | (c16 & 1);               // logical 16-byte chunk held at this LDS place
        long col = (isA ? m0 : n0) + lc16 * 8;
        const long lim = (isA ? p.M : p.N) - 8;
        if (col > lim) col = lim;
        voff[i] = (unsigned)(((long)(kt0 * BK + row) * (isA ? p.lda : p.ldb) + col) * 2);
    }
    auto uniform = [](const char* q) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)q), hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)q >> 32));
        return (const char*)(((uintptr_t)hi << 32) | lo);
    };
    // fetched once: behind a memory clobber the compiler re-loads the symbol's address (s_load + lgkmcnt(0): ~170 cycles) per piece
    const char* const zero_page = uniform(reinterpret_cast<const char*>(g_zero_page));
    auto stage = [&](int kt, int buf, int part = 0) {                        // part 0: all pieces, 1 / 2: first / second half
        unsigned char* base = smem + buf * G::STAGE_BYTES;
        const char* sA = uniform((const char*)p.A + (long)kt * BK * p.lda * 2);
        const char* sB = uniform((const char*)p.B + (long)kt * BK * p.ldb * 2);
        const bool tail = (kt0 + kt + 1) * BK > p.K;                          // scalar: only the last K step of a problem
#pragma unroll
        for (int i = 0; i < G::PPW; i++) {
            const int q = swave + G::NW * i;
            if (part == 1 && i >= G::PPW / 2) continue;
            if (part == 2 && i < G::PPW / 2) continue;
            if (q < G::PIECES) {                                             // wave-uniform
                const unsigned lds_addr = (unsigned)(uintptr_t)(lds_vp)(base + q * 1024);
                if (!tail) {                     // every k-row of this stage exists: scalar base + lane offset, no vector ALU work
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff[i]),
                                 "s"(q * 1024 < A_BYTES ? sA : sB) : "memory", "m0");
                    continue;
                }
                const char* s = (q * 1024 < A_BYTES ? sA : sB) + voff[i];
                if ((kt0 + kt) * BK + piece_row(q) >= p.K) s = zero_page;
                // Issued through inline asm on purpose: behind the builtin the compiler orders every later ds_read_b64_tr_b16
                // (an intrinsic it cannot disambiguate from the DMA's LDS write) after the DMA with s_waitcnt vmcnt(0), i.e.
                // each K step waited for the loads it had just issued.  Ordering is ours anyway: vmcnt(0) + barrier per step.
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(s) : "memory", "m0");
            }
        }
    };
    // ---- fragment read offsets: lane (fr, fg) reads k-rows fg*8 + qq (+4) of a 32-row half step, 8 bytes at column 4*pp
    const int fr = lane & 15, fg = lane >> 4, qq = fr >> 2, pp = fr & 3;
    // both reads of a fragment (rows r and r+4) share the swizzle value (it ignores bit 2 of the row), as do the two half steps
    const int swA = tn_swz<BM>(fg * 8 + qq), swB = tn_swz<BN>(fg * 8 + qq);
    const int rowoff_a = (fg * 8 + qq) * PA + pp * 8, rowoff_b = A_BYTES + (fg * 8 + qq) * PB + pp * 8;
    const int cbase_a = wr * (BM / 32), cbase_b = wc * (BN / WN / 16);          // first 32-byte column chunk of the wave's tile

    stage(0, 0);
    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    typedef bf16x4_t __attribute__((address_space(3))) * lds4p;
    constexpr int LAB_BASE = G::LDS_BYTES;
    bf16x8_t fa[TM], fb[TN];
#define TN_RD(ls, ks)                                                                                                  \
    _Pragma("unroll") for (int j = 0; j < TN; j++) {                                                                   \
        const unsigned char* a0 = (ls) + rowoff_b + (ks) * 32 * PB + (((cbase_b + j) ^ swB) << 5);                     \
        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(a0));                                      \
        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(a0 + 4 * PB));                             \
        fb[j] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                                      \
    }                                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < TM; i++) {                                                                   \
        const unsigned char* a0 = (ls) + rowoff_a + (ks) * 32 * PA + (((cbase_a + i) ^ swA) << 5);                     \
        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(a0));                                      \
        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4p)(a0 + 4 * PA));                             \
        fa[i] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                                      \
    }
#define TN_MM()                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < TM; i++)                                                                     \
        _Pragma("unroll") for (int j = 0; j < TN; j++)                                                                 \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    // the two waves of a SIMD run half a K step apart; the late one issues its DMA pieces after its held MFMAs (gemm_big_k)
#ifdef GEMM_LAB
    const bool late = p.stagger && __builtin_amdgcn_readfirstlane(wave) >= G::NW / 2;
#else
    constexpr bool late = false;                 // staggered waves: faster on the 256 x 256 tile alone, slower on the 320-wide ones (registers)
#endif
    if (!late) {
        for (int kt = 0; kt < nk; kt++) {
            STAMP(kt, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STAMP(kt, 1);
            __syncthreads();
            STAMP(kt, 2);
            const bool halves = p.stagger == 2;
            if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1, halves ? 1 : 0);
            STAMP(kt, 3);
            const unsigned char* ls = smem + (kt & 1) * G::STAGE_BYTES;
            TN_RD(ls, 0) TN_MM()
            if (halves && kt + 1 < nk) stage(kt + 1, (kt + 1) & 1, 2);
            STAMP(kt, 4);
            TN_RD(ls, 1) TN_MM()
            STAMP(kt, 5);
        }
    } else {
        for (int kt = 0; kt < nk; kt++) {
            STAMP(kt, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STAMP(kt, 1);
            __syncthreads();
            STAMP(kt, 2);
            const unsigned char* ls = smem + (kt & 1) * G::STAGE_BYTES;
            if (kt > 0) { TN_MM() }
            STAMP(kt, 3);
            const bool halves = p.stagger == 2;
            if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1, halves ? 1 : 0);
            STAMP(kt, 4);
            TN_RD(ls, 0) TN_MM()
            if (halves && kt + 1 < nk) stage(kt + 1, (kt + 1) & 1, 2);
            STAMP(kt, 5);
            TN_RD(ls, 1)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            STAMP(kt, 6);
        }
        if (nk > 0) { TN_MM() }
    }
#undef TN_RD
#undef TN_MM
    STAMP_DUMP();
    __syncthreads();
    big_epilogue<TO, TM, TN, G::EPI_ROW>(p, acc, smem + G::STAGE_BYTES + wave * G::EPI_WAVE, m0 + wr * (BM / 2), n0 + wc * (BN / WN), lane);
}

template <typename TO, int TM, int TN>
int launch_big_tn(const GemmP& p, hipStream_t st, int splits) {
    typedef BigTile<TM, TN, 4> G;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_big_tn_k<TO, TM, TN, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES + LAB_LDS);
        if (e != hipSuccess) MM_FAIL(MM_ERR_LAUNCH, "gemm_big_tn: cannot raise dynamic LDS to %d bytes: %s", G::LDS_BYTES, hipGetErrorString(e));
        attr_set = true;
    }
    const int tiles = p.tiles_m * p.tiles_n;
    hipLaunchKernelGGL((gemm_big_tn_k<TO, TM, TN, 4>), dim3((unsigned)tiles, (unsigned)splits), dim3(512), G::LDS_BYTES + LAB_LDS, st, p);
    MM_LAUNCH_CHECK();
    if (p.partial) {
        const long nq = (long)p.M * ((p.N + 3) / 4);
        hipLaunchKernelGGL(big_splitk_epilogue_k<TO>, dim3((unsigned)cdiv(nq, 256)), dim3(256), 0, st, p, splits);
        MM_LAUNCH_CHECK();
    }
    return MM_OK;
}

struct Shape { int tm, tn, wn; };
// 8-wave shapes (256x256 class) for problems that fill the chip with them, 4-wave shapes (128x128, 64x128; two or three
// workgroups per CU) for skinny problems -- M of about 1000 rows: the encoder's stage-3 products at the config batch and the
// decode-step products -- where the big tiles would leave most CUs idle.
constexpr Shape kShapes[] = {{8, 4, 4}, {7, 4, 4}, {8, 5, 4}, {7, 5, 4}, {4, 4, 2}, {2, 4, 2}};
constexpr int kNumShapes = 6;

// Estimated time of a tile shape: rounds of (workgroups resident per CU) x (K steps x MFMA work + fixed prologue/epilogue).
double shape_cost(const GemmP& p, Shape s, bool allow_split = false) {
    const long tiles = cdiv(p.M, 32 * s.tm) * cdiv(p.N, 16 * s.tn * s.wn);
    const int per_cu = s.wn == 4 ? 1 : (s.tm == 4 ? 2 : 3);            // by LDS: 2 x (BM + BN) x 128 bytes per workgroup
    const long rounds = cdiv(tiles, (long)kNumCU * per_cu);
    // us per K step: 1.62 measured on the 256x256 shape (8 waves = two per SIMD sharing a matrix pipe).  A 4-wave workgroup
    // puts one wave on each SIMD; c co-resident workgroups give c waves per SIMD.  A step cannot beat its load latency.
    const long conc = s.wn == 4 ? 2 : std::min<long>(per_cu, cdiv(tiles, kNumCU));
    // measured alone on a CU: 0.65 us per step for the 128x128 shape, 0.52 for 64x128 (load latency, not MFMA work)
    const double floor_us = s.wn == 4 ? 0.40 : 0.38 + 0.017 * s.tm * s.tn;
    const double iter = std::max(floor_us, 1.62 * (s.tm * s.tn) / 32.0 * conc / 2.0);
    const double fixed = s.wn == 4 ? 12.0 : 7.0;
    if (s.wn != 4 && allow_split) {                          // split-K fills the chip: K steps divide, a reduce pass is added
        const long sp = std::max<long>(1, std::min<long>({8, (long)kNumCU * per_cu / std::max<long>(tiles, 1), (long)(p.K / BK) / 8}));
        if (sp > 1) {
            const long c2 = std::min<long>(per_cu, cdiv(tiles * sp, kNumCU));
            const double it2 = std::max(floor_us, 1.62 * (s.tm * s.tn) / 32.0 * c2 / 2.0);
            return cdiv(tiles * sp, (long)kNumCU * per_cu) * (cdiv(p.K / BK, sp) * it2 + fixed) + 4.0 + (double)sp * p.M * p.N * 8.0 / 4e6;
        }
    }
    return rounds * ((p.K / BK) * iter + fixed);
}
}  // namespace

// Is the large-tile kernel applicable (bf16, both operands K-contiguous, 16-byte aligned rows, K % 64 == 0) and worth it?
bool gemm_big_applicable(const GemmP& p, int dtype, int a_kmajor, int b_kmajor, int splits) {
    if (dtype != MM_BF16 || a_kmajor || b_kmajor || splits > 1 || p.partial) return false;
    if (!p.a_vec || !p.b_vec || p.K % BK != 0) return false;
    if ((size_t)p.M * p.lda * 2 >= (1ull << 32) || (size_t)p.N * p.ldb * 2 >= (1ull << 32)) return false;      // 32-bit lane offsets
    // 256x256-class tiles from a quarter of the chip up; the 4-wave shapes for skinny problems from about 1000 x 1000 x 1000
    // (measured: below that gemm_k's 64x64 tiles are as fast -- 13 us at 1984 x 1024 x 256 either way -- and the
    // one-K-step products of the first encoder stages are slower here)
    static const long min_mn = getenv("MM_GEMM_BIG_MIN_MN") ? atol(getenv("MM_GEMM_BIG_MIN_MN")) : 256L * 256 * 12;
    if ((long)p.M * p.N >= 256L * 256 * 64) return true;
    return (long)p.M * p.N >= min_mn && p.M >= 64 && p.N >= 1024 && p.K >= 1024;
}

// Split-K for a 4-wave shape: when its tiles leave most of the chip idle and K is deep, cut K so that about
// kNumCU * per_cu workgroups exist; each split must keep >= 8 K steps.  Needs splits * M * N floats of caller workspace.
static int pick_splits(const GemmP& p, Shape s, size_t ws_bytes) {
    if (s.wn == 4) return 1;
    const long tiles = cdiv(p.M, 32 * s.tm) * cdiv(p.N, 16 * s.tn * s.wn);
    const int per_cu = s.tm == 4 ? 2 : 3;
    long sp = std::min<long>({8, (long)kNumCU * per_cu / std::max<long>(tiles, 1), (long)(p.K / BK) / 8});
    while (sp > 1 && (size_t)sp * p.M * p.N * sizeof(float) > ws_bytes) sp--;
    return (int)std::max<long>(sp, 1);
}

int gemm_big_launch(GemmP& p, int out_dtype, hipStream_t st, void* ws, size_t ws_bytes) {
    // 8-wide access to the out family: 16-byte aligned bases, row stride a multiple of 16 bytes
    const int osz = out_dtype == MM_BF16 ? 2 : 4;
    auto al = [&](const void* q) { return q == nullptr || ((uintptr_t)q) % 16 == 0; };
    p.o_vec = (p.ldo * osz) % 16 == 0 && al(p.out) && al(p.pre) && al(p.mul) && al(p.res);
    int best = 0;
    static const char* force = getenv("MM_GEMM_BIG_SHAPE");              // "0".."5": benchmark override
    if (force) best = atoi(force) % kNumShapes;
    else
        for (int i = 1; i < kNumShapes; i++)
            if (shape_cost(p, kShapes[i], ws != nullptr) < shape_cost(p, kShapes[best], ws != nullptr) * 0.97) best = i;
#ifdef GEMM_LAB
    const Shape s = {GEMM_LAB_TM, GEMM_LAB_TN, 4};
#else
    const Shape s = kShapes[best];
#endif
    static const char* fstag = getenv("MM_GEMM_BIG_STAGGER");            // "0" / "1": benchmark override
    p.stagger = fstag ? atoi(fstag) : 1;                          // measured: 1-3 % on every routed shape, 4-9 % with the late DMA issue
    p.tiles_m = (int)cdiv(p.M, 32 * s.tm); p.tiles_n = (int)cdiv(p.N, 16 * s.tn * s.wn);
    static const char* fsp = getenv("MM_GEMM_BIG_SPLITS");               // benchmark override
    int splits = ws ? (fsp ? atoi(fsp) : pick_splits(p, s, ws_bytes)) : 1;
    if (s.wn == 4 || (size_t)splits * p.M * p.N * sizeof(float) > ws_bytes) splits = 1;
    if (splits > 1) {
        const int steps = (int)cdiv(p.K / BK, splits);
        p.k_per_split = steps * BK;
        splits = (int)cdiv(p.K / BK, steps);
        p.partial = splits > 1 ? (float*)ws : nullptr;
    }
    if (splits <= 1) { splits = 1; p.partial = nullptr; p.k_per_split = p.K; }
#ifdef GEMM_LAB
    return launch_big<bf16, GEMM_LAB_TM, GEMM_LAB_TN, 4>(p, st, splits);
#else
#define GO_(TM_, TN_, WN_) return out_dtype == MM_BF16 ? launch_big<bf16, TM_, TN_, WN_>(p, st, splits) : launch_big<float, TM_, TN_, WN_>(p, st, splits)
    switch (best) {
        case 0: GO_(8, 4, 4);
        case 1: GO_(7, 4, 4);
        case 2: GO_(8, 5, 4);
        case 3: GO_(7, 5, 4);
        case 4: GO_(4, 4, 2);
        default: GO_(2, 4, 2);
    }
#undef GO_
#endif
}

// ---- weight-gradient form (both operands K-major): applicability, shape / split choice, launch
bool gemm_big_tn_applicable(const GemmP& p, int dtype, int a_kmajor, int b_kmajor, int splits) {
    static const bool off = getenv("MM_GEMM_NO_TN") != nullptr;              // A/B switch
    if (off || dtype != MM_BF16 || !a_kmajor || !b_kmajor || splits > 1 || p.partial) return false;
    if (!p.a_vec || !p.b_vec || p.M % 8 || p.N % 8 || p.M < 256 || p.N < 256 || p.K < 1024) return false;
    if ((size_t)p.K * p.lda * 2 >= (1ull << 32) || (size_t)p.K * p.ldb * 2 >= (1ull << 32)) return false;    // 32-bit lane offsets
    return (long)p.M * p.N >= 256L * 256 * 12;
}

int gemm_big_tn_launch(GemmP& p, int out_dtype, hipStream_t st, void* ws, size_t ws_bytes) {
    const int osz = out_dtype == MM_BF16 ? 2 : 4;
    auto al = [&](const void* q) { return q == nullptr || ((uintptr_t)q) % 16 == 0; };
    p.o_vec = (p.ldo * osz) % 16 == 0 && al(p.out) && al(p.pre) && al(p.mul) && al(p.res);
    // staggered waves (as in gemm_big_k) measured faster on the 256 x 256 tile alone and slower over the shapes the cost model
    // picks (registers: the 320-wide tiles): off unless asked for
    static const char* fstag = getenv("MM_GEMM_TN_STAGGER");               // "0" / "1" / "2": benchmark override
    p.stagger = fstag ? atoi(fstag) : 0;
    struct Cand { int tm, tn; };
    static const Cand cands[3] = {{8, 4}, {8, 5}, {10, 4}};                   // 256x256, 256x320, 320x256
    const long ksteps = cdiv(p.K, BK);
    double best_cost = 1e30; int best = 0, best_sp = 1;
    static const char* fshape = getenv("MM_GEMM_TN_SHAPE");
    static const char* fsplit = getenv("MM_GEMM_TN_SPLITS");
    for (int c = 0; c < 3; c++) {
        if (fshape && atoi(fshape) != c) continue;
        const long tiles = cdiv(p.M, 32 * cands[c].tm) * cdiv(p.N, 64 * cands[c].tn);
        static const double kStepUs[3] = {2.1, 2.5, 2.4};                     // measured us per K step of each shape
        for (int sp = 1; sp <= 8; sp++) {
            if (fsplit && atoi(fsplit) != sp) continue;
            if (sp > 1 && ((size_t)sp * p.M * p.N * sizeof(float) > ws_bytes || ksteps / sp < 8)) continue;
            const long rounds = cdiv(tiles * sp, kNumCU);
            double cost = rounds * (cdiv(ksteps, sp) * kStepUs[c] + 9.0);
            if (sp > 1) cost += 5.0 + (double)(sp + 1) * p.M * p.N * 4.0 / 4.5e6;   // slab writes are in the tile time; the reduce pass
            if (cost < best_cost) { best_cost = cost; best = c; best_sp = sp; }
        }
    }
    if (best_cost > 1e29) MM_FAIL(MM_ERR_ARG, "gemm_big_tn: no admissible shape / split");
    (void)ws;
#ifdef GEMM_LAB
    const Cand s = {GEMM_LAB_TTM, GEMM_LAB_TTN};
#else
    const Cand s = cands[best];
#endif
    p.tiles_m = (int)cdiv(p.M, 32 * s.tm); p.tiles_n = (int)cdiv(p.N, 64 * s.tn);
    int splits = best_sp;
    if (splits > 1) {
        const int steps = (int)cdiv(ksteps, splits);
        p.k_per_split = steps * BK;
        splits = (int)cdiv(ksteps, steps);
        p.partial = splits > 1 ? (float*)ws : nullptr;
    }
    if (splits <= 1) { splits = 1; p.partial = nullptr; p.k_per_split = (int)(ksteps * BK); }
#ifdef GEMM_LAB
    return launch_big_tn<bf16, GEMM_LAB_TTM, GEMM_LAB_TTN>(p, st, splits);
#else
#define GO_(TM_, TN_) return out_dtype == MM_BF16 ? launch_big_tn<bf16, TM_, TN_>(p, st, splits) : launch_big_tn<float, TM_, TN_>(p, st, splits)
    switch (best) {
        case 0: GO_(8, 4);
        case 1: GO_(8, 5);
        default: GO_(10, 4);
    }
#undef GO_
#endif
}
