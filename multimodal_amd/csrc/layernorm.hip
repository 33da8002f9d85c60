// Row LayerNorm (forward/backward) and the stem Conv1d(1,C0,k4,s4)+LayerNorm.
// Replaces nn.LayerNorm at reference models/convnext.py:70,:78,:131 and
// models/transformer_decoder.py:122,:160,:306, and the stem conv at models/convnext.py:68.
// HBM-bound: one read of x, one write of y; one wave per row, row held in registers.
#include "common.h"
#include <stdlib.h>

// ------------------------------------------------------------------------------------------------
// forward: wave per row, CH chunks of VEC per lane (C <= 64*VEC*CH)
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int CH>
__global__ __launch_bounds__(256) void layernorm_fwd_k(const T* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ b, T* __restrict__ y,
                                                      float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                      long rows_out, int Lin, int Lout, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long row_o = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row_o >= rows_out) return;                       // wave-uniform
    const long g = row_o / Lout, l = row_o % Lout;
    const T* xr = x + (g * Lin + l) * C;
    T* yr = y + row_o * C;
    float v[CH][VEC];
    float s1 = 0.f;
#pragma unroll
    for (int k = 0; k < CH; k++) {
        const int c = (k * 64 + lane) * VEC;
        if (c < C) {
            VecIO<T, VEC>::load(xr + c, v[k]);
#pragma unroll
            for (int i = 0; i < VEC; i++) s1 += v[k][i];
        } else {
#pragma unroll
            for (int i = 0; i < VEC; i++) v[k][i] = 0.f;
        }
    }
    const float invC = 1.0f / (float)C;
    const float mu = wave_sum(s1) * invC;
    float s2 = 0.f;
#pragma unroll
    for (int k = 0; k < CH; k++) {
        const int c = (k * 64 + lane) * VEC;
        if (c < C) {
#pragma unroll
            for (int i = 0; i < VEC; i++) { v[k][i] -= mu; s2 = fmaf(v[k][i], v[k][i], s2); }
        }
    }
    const float rs = rsqrtf(wave_sum(s2) * invC + eps);
#pragma unroll
    for (int k = 0; k < CH; k++) {
        const int c = (k * 64 + lane) * VEC;
        if (c < C) {
            float o[VEC];
#pragma unroll
            for (int i = 0; i < VEC; i++) o[i] = fmaf(v[k][i] * rs, w[c + i], b[c + i]);
            VecIO<T, VEC>::store(yr + c, o);
        }
    }
    if (mean_out && lane == 0) { mean_out[row_o] = mu; rstd_out[row_o] = rs; }
}

// ------------------------------------------------------------------------------------------------
// backward: each wave walks rows (grid-stride), keeps per-lane dw/db partials for its channels;
// block partials -> workspace [nblocks][2][C]; layernorm_bwd_reduce sums them in a fixed order.
// Rows l >= Lout of a group (dropped by the forward) get dx = dres or 0.
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, int CH>
__global__ __launch_bounds__(256) void layernorm_bwd_k(const T* __restrict__ x, const T* __restrict__ dy,
                                                      const T* __restrict__ dres, const float* __restrict__ w,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      T* __restrict__ dx, float* __restrict__ partial, long rows_in,
                                                      int Lin, int Lout, int C) {
    __shared__ float acc_lds[4][2 * 64 * VEC * CH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float pw[CH][VEC], pb[CH][VEC], wv[CH][VEC];
#pragma unroll
    for (int k = 0; k < CH; k++) {
        const int c = (k * 64 + lane) * VEC;
#pragma unroll
        for (int i = 0; i < VEC; i++) { pw[k][i] = 0.f; pb[k][i] = 0.f; wv[k][i] = (c + i < C) ? w[c + i] : 0.f; }
    }
    const float invC = 1.0f / (float)C;
    for (long row = (long)blockIdx.x * 4 + wave; row < rows_in; row += (long)gridDim.x * 4) {
        const long g = row / Lin, l = row % Lin;
        T* dxr = dx + row * C;
        if (l >= Lout) {                                  // dropped row: gradient is only the residual
#pragma unroll
            for (int k = 0; k < CH; k++) {
                const int c = (k * 64 + lane) * VEC;
                if (c < C) {
                    float o[VEC];
                    if (dres) VecIO<T, VEC>::load(dres + row * C + c, o);
                    else {
#pragma unroll
                        for (int i = 0; i < VEC; i++) o[i] = 0.f;
                    }
                    VecIO<T, VEC>::store(dxr + c, o);
                }
            }
            continue;
        }
        const long row_o = g * Lout + l;
        const float mu = mean[row_o], rs = rstd[row_o];
        float xh[CH][VEC], gg[CH][VEC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const int c = (k * 64 + lane) * VEC;
            if (c < C) {
                float xv[VEC], gy[VEC];
                VecIO<T, VEC>::load(x + row * C + c, xv);
                VecIO<T, VEC>::load(dy + row_o * C + c, gy);
#pragma unroll
                for (int i = 0; i < VEC; i++) {
                    xh[k][i] = (xv[i] - mu) * rs;
                    gg[k][i] = gy[i] * wv[k][i];
                    s1 += gg[k][i]; s2 = fmaf(gg[k][i], xh[k][i], s2);
                    pw[k][i] = fmaf(gy[i], xh[k][i], pw[k][i]);
                    pb[k][i] += gy[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < VEC; i++) { xh[k][i] = 0.f; gg[k][i] = 0.f; }
            }
        }
        const float m1 = wave_sum(s1) * invC, m2 = wave_sum(s2) * invC;
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const int c = (k * 64 + lane) * VEC;
            if (c < C) {
                float o[VEC];
                if (dres) VecIO<T, VEC>::load(dres + row * C + c, o);
                else {
#pragma unroll
                    for (int i = 0; i < VEC; i++) o[i] = 0.f;
                }
#pragma unroll
                for (int i = 0; i < VEC; i++) o[i] += rs * (gg[k][i] - m1 - xh[k][i] * m2);
                VecIO<T, VEC>::store(dxr + c, o);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < CH; k++)
#pragma unroll
        for (int i = 0; i < VEC; i++) {
            const int slot = (k * 64 + lane) * VEC + i;
            acc_lds[wave][slot] = pw[k][i];
            acc_lds[wave][64 * VEC * CH + slot] = pb[k][i];
        }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 2 * 64 * VEC * CH; idx += 256) {
        const int kind = idx / (64 * VEC * CH), c = idx % (64 * VEC * CH);
        if (c < C) partial[((long)blockIdx.x * 2 + kind) * C + c] =
            acc_lds[0][idx] + acc_lds[1][idx] + acc_lds[2][idx] + acc_lds[3][idx];
    }
}

// Wide rows (C > 512): one BLOCK walks rows, R at a time; a thread owns VEC*CH fixed channels, so the dw/db partials
// need no cross-thread step.  The 3*R*CH 16-byte loads of a pass are issued together and stay packed in registers
// (bf16 rows: 4 VGPRs per load) until used; each row is unpacked once for the sums and once for dx.
// SUMS: also accumulate the column sums of the residual-gradient input (dres) and of the produced dx -- the bias gradients of
// the two linears on either side of the LayerNorm in a pre-LN residual block -- as partial kinds 2 and 3.
template <typename T, int VEC, int CH, int R, bool SUMS>
__global__ __launch_bounds__(256) void layernorm_bwd_blk_k(const T* __restrict__ x, const T* __restrict__ dy,
                                                          const T* __restrict__ dres, const float* __restrict__ w,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          T* __restrict__ dx, float* __restrict__ partial, long rows_in,
                                                          int Lin, int Lout, int C) {
    typedef typename VecIO<T, VEC>::raw_t raw_t;
    __shared__ float red[2][R][2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NK = SUMS ? 4 : 2;
    float pw[CH][VEC], pb[CH][VEC], wv[CH][VEC], pr[SUMS ? CH : 1][VEC], px[SUMS ? CH : 1][VEC];
#pragma unroll
    for (int k = 0; k < CH; k++) {
        const int c = (k * 256 + tid) * VEC;
#pragma unroll
        for (int i = 0; i < VEC; i++) {
            pw[k][i] = 0.f; pb[k][i] = 0.f; wv[k][i] = (c + i < C) ? w[c + i] : 0.f;
            if (SUMS) { pr[k][i] = 0.f; px[k][i] = 0.f; }
        }
    }
    const float invC = 1.0f / (float)C;
    int par = 0;
    for (long row0 = (long)blockIdx.x * R; row0 < rows_in; row0 += (long)gridDim.x * R, par ^= 1) {
        raw_t rx[R][CH], rg[R][CH], rr[R][CH];
        float mu[R], rs[R];
        bool live[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const long row = row0 + r;
            const long g = row / Lin, l = row % Lin;
            live[r] = row < rows_in && l < Lout;                               // block-uniform
            const long row_o = g * Lout + l;
#pragma unroll
            for (int k = 0; k < CH; k++) {
                const int c = (k * 256 + tid) * VEC;
                rx[r][k] = raw_t{}; rg[r][k] = raw_t{}; rr[r][k] = raw_t{};
                if (c < C) {
                    if (live[r]) { rx[r][k] = VecIO<T, VEC>::load_raw(x + row * C + c); rg[r][k] = VecIO<T, VEC>::load_raw(dy + row_o * C + c); }
                    if (dres && row < rows_in) rr[r][k] = VecIO<T, VEC>::load_raw(dres + row * C + c);
                }
            }
            mu[r] = live[r] ? mean[row_o] : 0.f;
            rs[r] = live[r] ? rstd[row_o] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < CH; k++) {
                float xv[VEC], gy[VEC];
                VecIO<T, VEC>::unpack(rx[r][k], xv);
                VecIO<T, VEC>::unpack(rg[r][k], gy);
#pragma unroll
                for (int i = 0; i < VEC; i++) {
                    const float h = (xv[i] - mu[r]) * rs[r];                  // dead row / channel: rs = 0 or gy = w = 0
                    const float g = gy[i] * wv[k][i];
                    s1 += g; s2 = fmaf(g, h, s2);
                    pw[k][i] = fmaf(gy[i], h, pw[k][i]);
                    pb[k][i] += gy[i];
                }
            }
            s1 = wave_sum(s1); s2 = wave_sum(s2);
            if (lane == 0) { red[par][r][0][wave] = s1; red[par][r][1][wave] = s2; }
        }
        __syncthreads();      // the other parity's slots are rewritten only after every thread passed this barrier
#pragma unroll
        for (int r = 0; r < R; r++) {
            const long row = row0 + r;
            if (row < rows_in) {
                const float m1 = ((red[par][r][0][0] + red[par][r][0][1]) + (red[par][r][0][2] + red[par][r][0][3])) * invC;
                const float m2 = ((red[par][r][1][0] + red[par][r][1][1]) + (red[par][r][1][2] + red[par][r][1][3])) * invC;
#pragma unroll
                for (int k = 0; k < CH; k++) {
                    const int c = (k * 256 + tid) * VEC;
                    if (c < C) {
                        float xv[VEC], gy[VEC], o[VEC];
                        VecIO<T, VEC>::unpack(rx[r][k], xv);
                        VecIO<T, VEC>::unpack(rg[r][k], gy);
                        VecIO<T, VEC>::unpack(rr[r][k], o);
#pragma unroll
                        for (int i = 0; i < VEC; i++) {
                            const float h = (xv[i] - mu[r]) * rs[r];
                            if (SUMS) pr[k][i] += o[i];
                            o[i] += rs[r] * (gy[i] * wv[k][i] - m1 - h * m2);
                            if (SUMS) px[k][i] += o[i];
                        }
                        VecIO<T, VEC>::store(dx + row * C + c, o);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < CH; k++) {
        const int c = (k * 256 + tid) * VEC;
#pragma unroll
        for (int i = 0; i < VEC; i++)
            if (c + i < C) {
                partial[((long)blockIdx.x * NK + 0) * C + c + i] = pw[k][i];
                partial[((long)blockIdx.x * NK + 1) * C + c + i] = pb[k][i];
                if (SUMS) {
                    partial[((long)blockIdx.x * NK + 2) * C + c + i] = pr[k][i];
                    partial[((long)blockIdx.x * NK + 3) * C + c + i] = px[k][i];
                }
            }
    }
}

// out0[c] = sum_r partial[r][c], out1[c] = sum_r partial[r][C + c]  (width = 2*C); a block covers 16 columns with
// 16 row groups (64-byte coalesced reads, 8 loads in flight per thread), fixed summation order.
__global__ __launch_bounds__(256) void rows_reduce_k(const float* __restrict__ partial, int nrows, int width,
                                                     float* __restrict__ out0, float* __restrict__ out1, int C,
                                                     float* __restrict__ out2 = nullptr, float* __restrict__ out3 = nullptr) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int idx = blockIdx.x * 16 + cl;
    float s = 0.f;
    if (idx < width) {
        int r = g;
        for (; r + 7 * 16 < nrows; r += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = partial[(long)(r + u * 16) * width + idx];
#pragma unroll
            for (int u = 0; u < 8; u++) s += v[u];
        }
        for (; r < nrows; r += 16) s += partial[(long)r * width + idx];
    }
    red[g][cl] = s;
    __syncthreads();
    if (g == 0 && idx < width) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) t += red[k][cl];
        if (idx < C) out0[idx] = t; else if (idx < 2 * C) out1[idx - C] = t;
        else if (idx < 3 * C) out2[idx - 2 * C] = t; else out3[idx - 3 * C] = t;
    }
}

template <typename T>
static int ln_fwd_launch(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd, int64_t G,
                         int64_t Lin, int64_t Lout, int64_t C, float eps, hipStream_t st) {
    const long rows_out = G * Lout;
    const dim3 grid((unsigned)cdiv(rows_out, 4));
#define L_(VEC_, CH_) hipLaunchKernelGGL((layernorm_fwd_k<T, VEC_, CH_>), grid, dim3(256), 0, st, (const T*)x, w, b, (T*)y, mean, \
                                         rstd, rows_out, (int)Lin, (int)Lout, (int)C, eps)
    if (C % 4 == 0) {
        if (C <= 256) L_(4, 1); else if (C <= 512) L_(4, 2); else if (C <= 1024) L_(4, 4);
        else if (C <= 2048) L_(4, 8); else if (C <= 4096) L_(4, 16);
        else MM_FAIL(MM_ERR_UNSUPPORTED, "layernorm: C=%lld > 4096", (long long)C);
    } else {
        if (C <= 64) L_(1, 1); else if (C <= 256) L_(1, 4); else if (C <= 1024) L_(1, 16);
        else MM_FAIL(MM_ERR_UNSUPPORTED, "layernorm: C=%lld (not a multiple of 4) > 1024", (long long)C);
    }
#undef L_
    MM_LAUNCH_CHECK();
    return MM_OK;
}

extern "C" int mm_layernorm_fwd(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd,
                                int64_t G, int64_t Lin, int64_t Lout, int64_t C, float eps, int dtype, void* stream) {
    MM_CHECK(x && w && b && y, "layernorm_fwd: null pointer");
    MM_CHECK(G > 0 && Lin > 0 && Lout > 0 && Lout <= Lin && C > 0, "layernorm_fwd: bad shape G=%lld Lin=%lld Lout=%lld C=%lld",
             (long long)G, (long long)Lin, (long long)Lout, (long long)C);
    MM_CHECK((mean == nullptr) == (rstd == nullptr), "layernorm_fwd: mean/rstd must both be given or both NULL");
    if (dtype == MM_F32) return ln_fwd_launch<float>(x, w, b, y, mean, rstd, G, Lin, Lout, C, eps, (hipStream_t)stream);
    if (dtype == MM_BF16) return ln_fwd_launch<bf16>(x, w, b, y, mean, rstd, G, Lin, Lout, C, eps, (hipStream_t)stream);
    MM_FAIL(MM_ERR_ARG, "layernorm_fwd: bad dtype %d", dtype);
}

#define LN_BWD_R 4
static bool ln_bwd_wide(int64_t C) { return C > 512 && C % 4 == 0; }
static int64_t ln_bwd_blocks(int64_t rows, int64_t C) {
    int64_t nb = ln_bwd_wide(C) ? cdiv(rows, 4) : cdiv(rows, 4 * 4);        // wide: 4 rows per block pass
    static const int64_t wide_cap = getenv("MM_LN_CAP") ? atoi(getenv("MM_LN_CAP")) : 512;
    const int64_t cap = ln_bwd_wide(C) ? wide_cap : 1024;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return nb;
}
extern "C" size_t mm_layernorm_bwd_workspace(int64_t rows, int64_t C) {
    if (rows <= 0 || C <= 0) return 0;
    return (size_t)ln_bwd_blocks(rows, C) * 4 * (size_t)C * sizeof(float);      // 4: room for the two optional column sums
}

template <typename T>
static int ln_bwd_launch(const void* x, const void* dy, const void* dres, const float* w, const float* mean,
                         const float* rstd, void* dx, float* partial, int64_t nb, int64_t G, int64_t Lin, int64_t Lout,
                         int64_t C, hipStream_t st, bool sums = false) {
    const long rows_in = G * Lin;
#define L_(VEC_, CH_) hipLaunchKernelGGL((layernorm_bwd_k<T, VEC_, CH_>), dim3((unsigned)nb), dim3(256), 0, st, (const T*)x, \
                                         (const T*)dy, (const T*)dres, w, mean, rstd, (T*)dx, partial, rows_in, (int)Lin, (int)Lout, (int)C)
#define W1_(VEC_, CH_, S_) hipLaunchKernelGGL((layernorm_bwd_blk_k<T, VEC_, CH_, LN_BWD_R, S_>), dim3((unsigned)nb), dim3(256), 0, st, (const T*)x, \
                                         (const T*)dy, (const T*)dres, w, mean, rstd, (T*)dx, partial, rows_in, (int)Lin, (int)Lout, (int)C)
#define W_(VEC_, CH_) do { if (sums) W1_(VEC_, CH_, true); else W1_(VEC_, CH_, false); } while (0)
    if (ln_bwd_wide(C)) {
        bool done = false;
        if constexpr (sizeof(T) == 2) {
            if (C % 8 == 0) {
                if (C <= 2048) W_(8, 1); else if (C <= 4096) W_(8, 2);
                else MM_FAIL(MM_ERR_UNSUPPORTED, "layernorm_bwd: C=%lld > 4096", (long long)C);
                done = true;
            }
        }
        if (!done) {
            if (C <= 1024) W_(4, 1); else if (C <= 2048) W_(4, 2); else if (C <= 4096) W_(4, 4);
            else MM_FAIL(MM_ERR_UNSUPPORTED, "layernorm_bwd: C=%lld > 4096", (long long)C);
        }
    } else if (C % 4 == 0) {
        if (C <= 256) L_(4, 1); else L_(4, 2);
    } else {
        if (C <= 64) L_(1, 1); else if (C <= 256) L_(1, 4); else if (C <= 1024) L_(1, 16);
        else MM_FAIL(MM_ERR_UNSUPPORTED, "layernorm_bwd: C=%lld (not a multiple of 4) > 1024", (long long)C);
    }
#undef L_
#undef W_
#undef W1_
    MM_LAUNCH_CHECK();
    return MM_OK;
}

extern "C" int mm_layernorm_bwd(const void* x, const void* dy, const void* dres, const float* w, const float* mean,
                                const float* rstd, void* dx, float* dw, float* db, void* workspace,
                                size_t workspace_bytes, int64_t G, int64_t Lin, int64_t Lout, int64_t C, int dtype,
                                void* stream) {
    MM_CHECK(x && dy && w && mean && rstd && dx && dw && db && workspace, "layernorm_bwd: null pointer");
    MM_CHECK(G > 0 && Lin > 0 && Lout > 0 && Lout <= Lin && C > 0, "layernorm_bwd: bad shape");
    const int64_t rows = G * Lin;
    MM_CHECK(workspace_bytes >= mm_layernorm_bwd_workspace(rows, C), "layernorm_bwd: workspace too small");
    const int64_t nb = ln_bwd_blocks(rows, C);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (dtype == MM_F32) rc = ln_bwd_launch<float>(x, dy, dres, w, mean, rstd, dx, (float*)workspace, nb, G, Lin, Lout, C, st);
    else if (dtype == MM_BF16) rc = ln_bwd_launch<bf16>(x, dy, dres, w, mean, rstd, dx, (float*)workspace, nb, G, Lin, Lout, C, st);
    else MM_FAIL(MM_ERR_ARG, "layernorm_bwd: bad dtype %d", dtype);
    if (rc != MM_OK) return rc;
    hipLaunchKernelGGL(rows_reduce_k, dim3((unsigned)cdiv(2 * C, 16)), dim3(256), 0, st, (const float*)workspace, (int)nb, (int)(2 * C), dw,
                       db, (int)C, (float*)nullptr, (float*)nullptr);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

// The same, for a pre-LN residual block  y = x + f(LN(x)):  dx = dres + LN'(dy), and additionally
//   dres_sum[c] = sum_rows dres[row][c]   (bias gradient of the linear that produced the block's output gradient path)
//   dx_sum[c]   = sum_rows dx[row][c]     (bias gradient of the linear that produced x)
// so the two bias gradients cost no pass of their own.  Wide rows only (C > 512, C % 4 == 0, Lin == Lout).
extern "C" int mm_layernorm_bwd_sums(const void* x, const void* dy, const void* dres, const float* w, const float* mean,
                                     const float* rstd, void* dx, float* dw, float* db, float* dres_sum, float* dx_sum,
                                     void* workspace, size_t workspace_bytes, int64_t rows, int64_t C, int dtype, void* stream) {
    MM_CHECK(x && dy && dres && w && mean && rstd && dx && dw && db && dres_sum && dx_sum && workspace, "layernorm_bwd_sums: null pointer");
    MM_CHECK(rows > 0 && ln_bwd_wide(C), "layernorm_bwd_sums: needs rows > 0, C > 512, C %% 4 == 0 (C=%lld)", (long long)C);
    MM_CHECK(workspace_bytes >= mm_layernorm_bwd_workspace(rows, C), "layernorm_bwd_sums: workspace too small");
    const int64_t nb = ln_bwd_blocks(rows, C);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (dtype == MM_F32) rc = ln_bwd_launch<float>(x, dy, dres, w, mean, rstd, dx, (float*)workspace, nb, 1, rows, rows, C, st, true);
    else if (dtype == MM_BF16) rc = ln_bwd_launch<bf16>(x, dy, dres, w, mean, rstd, dx, (float*)workspace, nb, 1, rows, rows, C, st, true);
    else MM_FAIL(MM_ERR_ARG, "layernorm_bwd_sums: bad dtype %d", dtype);
    if (rc != MM_OK) return rc;
    hipLaunchKernelGGL(rows_reduce_k, dim3((unsigned)cdiv(4 * C, 16)), dim3(256), 0, st, (const float*)workspace, (int)nb, (int)(4 * C), dw,
                       db, (int)C, dres_sum, dx_sum);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

// ------------------------------------------------------------------------------------------------
// stem: y[b][l][c] = LN_c( sum_k w[c][k] * x[b][4l+k] + bias[c] ), C0 = 4*LPR channels, LPR lanes per row
// ------------------------------------------------------------------------------------------------
template <typename T, int LPR>
__global__ __launch_bounds__(256) void stem_ln_fwd_k(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ cb, const float* __restrict__ lw,
                                                    const float* __restrict__ lb, T* __restrict__ y,
                                                    float* __restrict__ mean_out, float* __restrict__ rstd_out, long rows,
                                                    int Lin, int Lo, float eps) {
    constexpr int C = 4 * LPR;
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int li = lane % LPR;
    long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const bool live = row < rows;
    if (!live) row = rows - 1;
    const long b = row / Lo, l = row % Lo;
    const float4 xin = *reinterpret_cast<const float4*>(x + b * Lin + 4 * l);
    const int c0 = li * 4;
    float a[4];
    float s1 = 0.f;
#pragma unroll
    for (int v = 0; v < 4; v++) {
        const float4 wv = *reinterpret_cast<const float4*>(w + (c0 + v) * 4);
        a[v] = fmaf(wv.w, xin.w, fmaf(wv.z, xin.z, fmaf(wv.y, xin.y, fmaf(wv.x, xin.x, cb[c0 + v]))));
        s1 += a[v];
    }
    const float invC = 1.0f / (float)C;
    const float mu = group_sum<LPR>(s1) * invC;
    float s2 = 0.f;
#pragma unroll
    for (int v = 0; v < 4; v++) { a[v] -= mu; s2 = fmaf(a[v], a[v], s2); }
    const float rs = rsqrtf(group_sum<LPR>(s2) * invC + eps);
    if (live) {
        float o[4];
#pragma unroll
        for (int v = 0; v < 4; v++) o[v] = fmaf(a[v] * rs, lw[c0 + v], lb[c0 + v]);
        VecIO<T, 4>::store(y + row * C + c0, o);
        if (mean_out && li == 0) { mean_out[row] = mu; rstd_out[row] = rs; }
    }
}

// backward of the stem; partial row layout: [4][C] dw (tap-major), [C] db, [C] dln_w, [C] dln_b = 7*C
template <typename T, int LPR>
__global__ __launch_bounds__(256) void stem_ln_bwd_k(const float* __restrict__ x, const T* __restrict__ dy,
                                                    const float* __restrict__ w, const float* __restrict__ cb,
                                                    const float* __restrict__ lw, const float* __restrict__ mean,
                                                    const float* __restrict__ rstd, float* __restrict__ dx,
                                                    float* __restrict__ partial, long rows, int Lin, int Lo) {
    constexpr int C = 4 * LPR;
    constexpr int RPW = 64 / LPR;
    __shared__ float acc_lds[4 * RPW][7 * C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane % LPR, sub = lane / LPR;
    const int c0 = li * 4;
    float wv[4][4], bias[4], g[4];
    float pdw[4][4], pdb[4], pdg[4], pdbeta[4];
#pragma unroll
    for (int v = 0; v < 4; v++) {
#pragma unroll
        for (int k = 0; k < 4; k++) { wv[v][k] = w[(c0 + v) * 4 + k]; pdw[v][k] = 0.f; }
        bias[v] = cb[c0 + v]; g[v] = lw[c0 + v]; pdb[v] = 0.f; pdg[v] = 0.f; pdbeta[v] = 0.f;
    }
    const float invC = 1.0f / (float)C;
    const long stride = (long)gridDim.x * 4 * RPW;
    const long first = ((long)blockIdx.x * 4 + wave) * RPW + sub;
    const long niter = (rows + stride - 1) / stride;      // uniform trip count: DPP sums need full EXEC
    for (long it = 0; it < niter; it++) {
        long row = first + it * stride;
        const bool live = row < rows;
        if (!live) row = rows - 1;
        const long b = row / Lo, l = row % Lo;
        const float4 xin = *reinterpret_cast<const float4*>(x + b * Lin + 4 * l);
        const float xk[4] = {xin.x, xin.y, xin.z, xin.w};
        float gy[4];
        VecIO<T, 4>::load(dy + row * C + c0, gy);
        const float mu = mean[row], rs = live ? rstd[row] : 0.f;
        float xh[4], gg[4], d[4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            if (!live) gy[v] = 0.f;
            const float a = fmaf(wv[v][3], xk[3], fmaf(wv[v][2], xk[2], fmaf(wv[v][1], xk[1], fmaf(wv[v][0], xk[0], bias[v]))));
            xh[v] = (a - mu) * rs;
            gg[v] = gy[v] * g[v];
            s1 += gg[v]; s2 = fmaf(gg[v], xh[v], s2);
        }
        const float m1 = group_sum<LPR>(s1) * invC, m2 = group_sum<LPR>(s2) * invC;
        float dxk[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int v = 0; v < 4; v++) {
            d[v] = rs * (gg[v] - m1 - xh[v] * m2);
            pdg[v] = fmaf(gy[v], xh[v], pdg[v]); pdbeta[v] += gy[v]; pdb[v] += d[v];
#pragma unroll
            for (int k = 0; k < 4; k++) { pdw[v][k] = fmaf(d[v], xk[k], pdw[v][k]); dxk[k] = fmaf(wv[v][k], d[v], dxk[k]); }
        }
        if (dx) {
#pragma unroll
            for (int k = 0; k < 4; k++) dxk[k] = group_sum<LPR>(dxk[k]);
            if (live && li == 0) *reinterpret_cast<float4*>(dx + b * Lin + 4 * l) = make_float4(dxk[0], dxk[1], dxk[2], dxk[3]);
        }
    }
    float* mine = acc_lds[wave * RPW + sub];
#pragma unroll
    for (int v = 0; v < 4; v++) {
#pragma unroll
        for (int k = 0; k < 4; k++) mine[k * C + c0 + v] = pdw[v][k];
        mine[4 * C + c0 + v] = pdb[v];
        mine[5 * C + c0 + v] = pdg[v];
        mine[6 * C + c0 + v] = pdbeta[v];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 7 * C; idx += 256) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4 * RPW; k++) s += acc_lds[k][idx];
        partial[(long)blockIdx.x * 7 * C + idx] = s;
    }
}

__global__ void stem_bwd_reduce_k(const float* __restrict__ partial, int nrows, int C, float* __restrict__ dw,
                                  float* __restrict__ db, float* __restrict__ dlw, float* __restrict__ dlb) {
    __shared__ float red[4];
    const int idx = blockIdx.x;
    float s = 0.f;
    for (int r = threadIdx.x; r < nrows; r += 256) s += partial[(long)r * 7 * C + idx];
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) {
        const int kind = idx / C, c = idx % C;
        if (kind < 4) dw[c * 4 + kind] = s;
        else if (kind == 4) db[c] = s;
        else if (kind == 5) dlw[c] = s;
        else dlb[c] = s;
    }
}

static bool stem_ok(int64_t C0) { return C0 == 32 || C0 == 64 || C0 == 128 || C0 == 256; }

extern "C" int mm_stem_ln_fwd(const float* x, const float* w, const float* b, const float* ln_w, const float* ln_b, void* y,
                              float* mean, float* rstd, int64_t B, int64_t Lin, int64_t C0, float eps, int dtype,
                              void* stream) {
    MM_CHECK(x && w && b && ln_w && ln_b && y, "stem_ln_fwd: null pointer");
    MM_CHECK(B > 0 && Lin >= 4, "stem_ln_fwd: bad shape B=%lld Lin=%lld", (long long)B, (long long)Lin);
    MM_CHECK(Lin % 4 == 0, "stem_ln_fwd: Lin=%lld must be a multiple of 4 (16-byte row loads)", (long long)Lin);
    MM_CHECK((mean == nullptr) == (rstd == nullptr), "stem_ln_fwd: mean/rstd must both be given or both NULL");
    if (!stem_ok(C0)) MM_FAIL(MM_ERR_UNSUPPORTED, "stem_ln_fwd: C0=%lld unsupported (32,64,128,256)", (long long)C0);
    const int64_t Lo = (Lin - 4) / 4 + 1;
    const long rows = B * Lo;
    const int lpr = (int)(C0 / 4), rpw = 64 / lpr;
    const dim3 grid((unsigned)cdiv(cdiv(rows, rpw), 4));
    hipStream_t st = (hipStream_t)stream;
#define L_(T_, LPR_) hipLaunchKernelGGL((stem_ln_fwd_k<T_, LPR_>), grid, dim3(256), 0, st, x, w, b, ln_w, ln_b, (T_*)y, mean, rstd,                                         rows, (int)Lin, (int)Lo, eps)
#define D_(T_) switch (lpr) { case 8: L_(T_, 8); break; case 16: L_(T_, 16); break; case 32: L_(T_, 32); break; default: L_(T_, 64); }
    if (dtype == MM_F32) { D_(float) } else if (dtype == MM_BF16) { D_(bf16) } else MM_FAIL(MM_ERR_ARG, "stem_ln_fwd: bad dtype %d", dtype);
#undef D_
#undef L_
    MM_LAUNCH_CHECK();
    return MM_OK;
}

static int64_t stem_bwd_blocks(int64_t rows, int rpw) {
    int64_t nb = cdiv(rows, (int64_t)4 * rpw * 8);
    if (nb > 512) nb = 512;
    if (nb < 1) nb = 1;
    return nb;
}
extern "C" size_t mm_stem_ln_bwd_workspace(int64_t B, int64_t Lin, int64_t C0) {
    if (B <= 0 || Lin < 4 || !stem_ok(C0)) return 0;
    const int64_t Lo = (Lin - 4) / 4 + 1;
    return (size_t)stem_bwd_blocks(B * Lo, (int)(64 / (C0 / 4))) * 7 * (size_t)C0 * sizeof(float);
}

extern "C" int mm_stem_ln_bwd(const float* x, const void* dy, const float* w, const float* b, const float* ln_w,
                              const float* mean, const float* rstd, float* dx, float* dw, float* db, float* dln_w,
                              float* dln_b, void* workspace, size_t workspace_bytes, int64_t B, int64_t Lin, int64_t C0,
                              int dtype, void* stream) {
    MM_CHECK(x && dy && w && b && ln_w && mean && rstd && dw && db && dln_w && dln_b && workspace, "stem_ln_bwd: null pointer");
    MM_CHECK(B > 0 && Lin >= 4 && Lin % 4 == 0, "stem_ln_bwd: bad shape");
    if (!stem_ok(C0)) MM_FAIL(MM_ERR_UNSUPPORTED, "stem_ln_bwd: C0=%lld unsupported", (long long)C0);
    MM_CHECK(workspace_bytes >= mm_stem_ln_bwd_workspace(B, Lin, C0), "stem_ln_bwd: workspace too small");
    const int64_t Lo = (Lin - 4) / 4 + 1;
    const long rows = B * Lo;
    const int lpr = (int)(C0 / 4), rpw = 64 / lpr;
    const int64_t nb = stem_bwd_blocks(rows, rpw);
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
#define L_(T_, LPR_) hipLaunchKernelGGL((stem_ln_bwd_k<T_, LPR_>), dim3((unsigned)nb), dim3(256), 0, st, x, (const T_*)dy, w, b, ln_w,                                         mean, rstd, dx, partial, rows, (int)Lin, (int)Lo)
#define D_(T_) switch (lpr) { case 8: L_(T_, 8); break; case 16: L_(T_, 16); break; case 32: L_(T_, 32); break; default: L_(T_, 64); }
    if (dtype == MM_F32) { D_(float) } else if (dtype == MM_BF16) { D_(bf16) } else MM_FAIL(MM_ERR_ARG, "stem_ln_bwd: bad dtype %d", dtype);
#undef D_
#undef L_
    MM_LAUNCH_CHECK();
    hipLaunchKernelGGL(stem_bwd_reduce_k, dim3((unsigned)(7 * C0)), dim3(256), 0, st, partial, (int)nb, (int)C0, dw, db, dln_w, dln_b);
    MM_LAUNCH_CHECK();
    return MM_OK;
}
