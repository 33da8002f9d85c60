// Small fused elementwise / reduction pieces around the GEMMs.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// LayerScale + DropPath backward (autograd of reference models/convnext.py:45-47):
//   d_z[m][c] = dout[m][c] * gamma[c] * s[m/rps]
//   dgamma[c] = sum_m dout[m][c] * s * z[m][c]          (z = pwconv2 output incl. bias)
//   db2[c]    = sum_m d_z[m][c]
// Two-stage, fixed-order reduction: workspace [nchunks][2][C].
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void layerscale_bwd_k(const T* __restrict__ dout, const T* __restrict__ z, const float* __restrict__ gamma,
                                 const float* __restrict__ rowscale, int rps, T* __restrict__ dz, float* __restrict__ part,
                                 int M, int C, int rows_per_chunk) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    const float g = gamma[c];
    float sg = 0.f, sb = 0.f;
    for (int r = r0; r < r1; r++) {
        const float s = rowscale ? rowscale[r / rps] : 1.f;
        const float d = to_f(dout[(long)r * C + c]) * s;
        sg = fmaf(d, to_f(z[(long)r * C + c]), sg);
        const float v = d * g;
        sb += v;
        dz[(long)r * C + c] = from_f<T>(v);
    }
    part[((long)blockIdx.y * 2 + 0) * C + c] = sg;
    part[((long)blockIdx.y * 2 + 1) * C + c] = sb;
}
__global__ void pair_reduce_k(const float* __restrict__ part, int nchunks, int C, float* __restrict__ o0, float* __restrict__ o1) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 2 * C) return;
    const int kind = idx / C, c = idx % C;
    float s = 0.f;
    for (int k = 0; k < nchunks; k++) s += part[((long)k * 2 + kind) * C + c];
    (kind ? o1 : o0)[c] = s;
}
static int ls_chunks(int64_t M) { int64_t c = cdiv(M, 32); if (c > 512) c = 512; if (c < 1) c = 1; return (int)c; }
extern "C" size_t mm_layerscale_bwd_workspace(int64_t M, int64_t C) {
    return (M > 0 && C > 0) ? (size_t)ls_chunks(M) * 2 * (size_t)C * sizeof(float) : 0;
}
extern "C" int mm_layerscale_bwd(const void* dout, const void* z, const float* gamma, const float* rowscale,
                                 int64_t rows_per_scale, void* dz, float* dgamma, float* dbias, void* workspace,
                                 size_t workspace_bytes, int64_t M, int64_t C, int dtype, void* stream) {
    MM_CHECK(dout && z && gamma && dz && dgamma && dbias && workspace, "layerscale_bwd: null pointer");
    MM_CHECK(M > 0 && C > 0, "layerscale_bwd: empty shape");
    MM_CHECK(!rowscale || rows_per_scale > 0, "layerscale_bwd: rowscale needs rows_per_scale > 0");
    MM_CHECK(workspace_bytes >= mm_layerscale_bwd_workspace(M, C), "layerscale_bwd: workspace too small");
    const int chunks = ls_chunks(M);
    const int rpc = (int)cdiv(M, chunks);
    const int nchunks = (int)cdiv(M, rpc);
    const dim3 grid((unsigned)cdiv(C, 256), (unsigned)nchunks);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MM_F32)
        hipLaunchKernelGGL(layerscale_bwd_k<float>, grid, dim3(256), 0, st, (const float*)dout, (const float*)z, gamma, rowscale,
                           (int)rows_per_scale, (float*)dz, (float*)workspace, (int)M, (int)C, rpc);
    else if (dtype == MM_BF16)
        hipLaunchKernelGGL(layerscale_bwd_k<bf16>, grid, dim3(256), 0, st, (const bf16*)dout, (const bf16*)z, gamma, rowscale,
                           (int)rows_per_scale, (bf16*)dz, (float*)workspace, (int)M, (int)C, rpc);
    else MM_FAIL(MM_ERR_ARG, "layerscale_bwd: bad dtype %d", dtype);
    MM_LAUNCH_CHECK();
    hipLaunchKernelGGL(pair_reduce_k, dim3((unsigned)cdiv(2 * C, 256)), dim3(256), 0, st, (const float*)workspace, nchunks, (int)C, dgamma, dbias);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

// ------------------------------------------------------------------------------------------------
// argmax over the last dim (train_autoregressive.py:102); first maximal index wins, like torch.argmax
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void argmax_rows_k(const T* __restrict__ x, long* __restrict__ out, long rows, int V, long ld) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int c = lane; c < V; c += 64) {
        const float v = to_f(x[row * ld + c]);
        if (v > best || (v == best && c < bi)) { best = v; bi = c; }   // NaN never wins, as a max of finite logits
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off); const int oi = __shfl_xor(bi, off);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) out[row] = bi == 0x7fffffff ? 0 : bi;
}
extern "C" int mm_argmax_rows(const void* logits, int64_t* out, int64_t rows, int64_t V, int64_t ld, int dtype, void* stream) {
    MM_CHECK(logits && out, "argmax_rows: null pointer");
    MM_CHECK(rows > 0 && V > 0 && ld >= V, "argmax_rows: bad shape");
    const dim3 grid((unsigned)cdiv(rows, 4));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MM_F32) hipLaunchKernelGGL(argmax_rows_k<float>, grid, dim3(256), 0, st, (const float*)logits, (long*)out, (long)rows, (int)V, (long)ld);
    else if (dtype == MM_BF16) hipLaunchKernelGGL(argmax_rows_k<bf16>, grid, dim3(256), 0, st, (const bf16*)logits, (long*)out, (long)rows, (int)V, (long)ld);
    else MM_FAIL(MM_ERR_ARG, "argmax_rows: bad dtype %d", dtype);
    MM_LAUNCH_CHECK();
    return MM_OK;
}
