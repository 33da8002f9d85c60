// Small fused elementwise / reduction pieces around the GEMMs.
#include "common.h"
#include <algorithm>

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
__global__ __launch_bounds__(256) void pair_reduce_k(const float* __restrict__ part, int nchunks, int C, float* __restrict__ o0,
                                                     float* __restrict__ o1) {
    __shared__ float red[4][64];                                   // 64 outputs x 4 chunk groups per block
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + cl;
    const int kind = idx / C, c = idx % C;
    float s = 0.f;
    if (idx < 2 * C)
        for (int k = g; k < nchunks; k += 4) s += part[((long)k * 2 + kind) * C + c];
    red[g][cl] = s;
    __syncthreads();
    if (g == 0 && idx < 2 * C) (kind ? o1 : o0)[c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}
// bf16, C % 8 == 0, C <= 2048: a lane owns 8 consecutive channels (16-byte accesses), C/8 lanes cover a row and the block's
// 256 / (C/8) row lanes walk the chunk's rows together; the row lanes' sums meet in LDS in a fixed order.  (The scalar kernel above
// -- a thread per channel, 32 rows one after the other, 2-byte accesses, 64 of 256 threads alive at C = 64 -- took 18-20 us per
// ConvNeXt block on the encoder backward's critical path for 1.5 MB of traffic.)
__global__ __launch_bounds__(256) void layerscale_bwd_v_k(const bf16* __restrict__ dout, const bf16* __restrict__ z,
                                                         const float* __restrict__ gamma, const float* __restrict__ rowscale, int rps,
                                                         bf16* __restrict__ dz, float* __restrict__ part, int M, int C, int rows_per_chunk) {
    __shared__ float red[256][17];                                  // [thread][sg 0..7 | sb 0..7], padded
    const int cpl = C >> 3, rp = 256 / cpl;
    const int cg = threadIdx.x % cpl, rl = threadIdx.x / cpl;
    const bool live = rl < rp;
    const int r0 = blockIdx.x * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    float g[8], sg[8], sb[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { g[j] = gamma[cg * 8 + j]; sg[j] = 0.f; sb[j] = 0.f; }
    if (live) {
        for (int r = r0 + rl; r < r1; r += rp) {
            const float s = rowscale ? rowscale[r / rps] : 1.f;
            float d[8], zz[8], v[8];
            VecIO<bf16, 8>::load(dout + (long)r * C + cg * 8, d);
            VecIO<bf16, 8>::load(z + (long)r * C + cg * 8, zz);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float dd = d[j] * s;
                sg[j] = fmaf(dd, zz[j], sg[j]);
                v[j] = dd * g[j];
                sb[j] += v[j];
            }
            VecIO<bf16, 8>::store(dz + (long)r * C + cg * 8, v);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) { red[threadIdx.x][j] = sg[j]; red[threadIdx.x][8 + j] = sb[j]; }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 2 * C; idx += 256) {         // idx -> (kind, channel): sums over the row lanes in order
        const int kind = idx / C, c = idx % C;
        float a = 0.f;
        for (int k = 0; k < rp; k++) a += red[k * cpl + (c >> 3)][kind * 8 + (c & 7)];
        part[((long)blockIdx.x * 2 + kind) * C + c] = a;
    }
}
static int ls_chunks(int64_t M) { int64_t c = cdiv(M, 4); if (c > 512) c = 512; if (c < 1) c = 1; return (int)c; }
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
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MM_BF16 && C % 8 == 0 && C <= 2048 && (((uintptr_t)dout | (uintptr_t)z | (uintptr_t)dz) % 16) == 0) {
        const int rp = 256 / (int)(C / 8);                          // rows a block walks together
        const int target = chunks < 64 ? chunks : 64;               // few chunks: the second stage walks them one after the other
        const int rpc = (int)cdiv(cdiv(M, target), rp) * rp;        // a chunk is whole passes of the block
        const int nchunks = (int)cdiv(M, rpc);                      // <= chunks: the workspace holds them
        hipLaunchKernelGGL(layerscale_bwd_v_k, dim3((unsigned)nchunks), dim3(256), 0, st, (const bf16*)dout, (const bf16*)z, gamma, rowscale,
                           (int)rows_per_scale, (bf16*)dz, (float*)workspace, (int)M, (int)C, rpc);
        MM_LAUNCH_CHECK();
        hipLaunchKernelGGL(pair_reduce_k, dim3((unsigned)cdiv(2 * C, 64)), dim3(256), 0, st, (const float*)workspace, nchunks, (int)C, dgamma, dbias);
        MM_LAUNCH_CHECK();
        return MM_OK;
    }
    const int rpc = (int)cdiv(M, chunks);
    const int nchunks = (int)cdiv(M, rpc);
    const dim3 grid((unsigned)cdiv(C, 256), (unsigned)nchunks);
    if (dtype == MM_F32)
        hipLaunchKernelGGL(layerscale_bwd_k<float>, grid, dim3(256), 0, st, (const float*)dout, (const float*)z, gamma, rowscale,
                           (int)rows_per_scale, (float*)dz, (float*)workspace, (int)M, (int)C, rpc);
    else if (dtype == MM_BF16)
        hipLaunchKernelGGL(layerscale_bwd_k<bf16>, grid, dim3(256), 0, st, (const bf16*)dout, (const bf16*)z, gamma, rowscale,
                           (int)rows_per_scale, (bf16*)dz, (float*)workspace, (int)M, (int)C, rpc);
    else MM_FAIL(MM_ERR_ARG, "layerscale_bwd: bad dtype %d", dtype);
    MM_LAUNCH_CHECK();
    hipLaunchKernelGGL(pair_reduce_k, dim3((unsigned)cdiv(2 * C, 64)), dim3(256), 0, st, (const float*)workspace, nchunks, (int)C, dgamma, dbias);
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

// ------------------------------------------------------------------------------------------------
// OrthoGrad projection + AdamW, multi-tensor, two passes over the parameters (reference ortho_grad.py:60-99
// followed by torch.optim.AdamW.step):
//   pass 1  per chunk: partial (w.w, w.g, g.g)        -> finalize per tensor: proj = w.g/(w.w+eps),
//           |g_orth|^2 = g.g - 2 proj w.g + proj^2 w.w, scale = |g| / (|g_orth| + eps)   (rescale)
//   pass 2  g <- (g - proj w) * scale;  AdamW update of p, m, v with that g.
//   The closed form of |g_orth|^2 cancels when g is nearly parallel to w (tiny tensors, scalars): finalize flags a
//   tensor whose |g_orth|^2 < 1% of |g|^2 and og_exact_k then sums (g - proj w)^2 element by element for the flagged
//   tensors only (its blocks of every other tensor exit at once); pass 2 takes the scale from those sums.
// Tables (device memory, built once by the caller): descs[n_tensors], chunk_tensor[n_chunks],
// chunk_off[n_chunks], tensor_chunk_start[n_tensors+1].  Deterministic (fixed-order sums, f64 finalize).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void og_reduce_k(const mm_param_desc* __restrict__ descs, const int* __restrict__ chunk_tensor,
                                                  const long* __restrict__ chunk_off, int chunk_elems, float* __restrict__ part) {
    __shared__ float red[3][4];
    const mm_param_desc d = descs[chunk_tensor[blockIdx.x]];
    const long off = chunk_off[blockIdx.x];
    const long end = min(d.n, off + (long)chunk_elems);
    float ww = 0.f, wg = 0.f, gg = 0.f;
    long i0 = off;
    if (((((uintptr_t)d.p) | ((uintptr_t)d.g)) & 15) == 0) {          // 16 B per lane; chunk offsets are multiples of 4
        const long vend = off + ((end - off) & ~3L);
#pragma unroll 4
        for (long i = off + threadIdx.x * 4; i < vend; i += 1024) {
            const float4 w = *(const float4*)(d.p + i), g = *(const float4*)(d.g + i);
            ww = fmaf(w.x, w.x, ww); wg = fmaf(w.x, g.x, wg); gg = fmaf(g.x, g.x, gg);
            ww = fmaf(w.y, w.y, ww); wg = fmaf(w.y, g.y, wg); gg = fmaf(g.y, g.y, gg);
            ww = fmaf(w.z, w.z, ww); wg = fmaf(w.z, g.z, wg); gg = fmaf(g.z, g.z, gg);
            ww = fmaf(w.w, w.w, ww); wg = fmaf(w.w, g.w, wg); gg = fmaf(g.w, g.w, gg);
        }
        i0 = vend;
    }
    for (long i = i0 + threadIdx.x; i < end; i += 256) {
        const float w = d.p[i], g = d.g[i];
        ww = fmaf(w, w, ww); wg = fmaf(w, g, wg); gg = fmaf(g, g, gg);
    }
    ww = wave_sum(ww); wg = wave_sum(wg); gg = wave_sum(gg);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ww; red[1][threadIdx.x >> 6] = wg; red[2][threadIdx.x >> 6] = gg; }
    __syncthreads();
    if (threadIdx.x < 3) part[(long)blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}
__global__ void og_finalize_k(const float* __restrict__ part, const int* __restrict__ tcs, int n_tensors, float ortho_eps,
                              int rescale, float* __restrict__ coef, int* __restrict__ flag) {
    // one wave per tensor: lanes stride over the tensor's chunk partials, fixed-order butterfly in double
    const int t = blockIdx.x;
    double ww = 0, wg = 0, gg = 0;
    for (int c = tcs[t] + threadIdx.x; c < tcs[t + 1]; c += 64) { ww += part[(long)c * 3]; wg += part[(long)c * 3 + 1]; gg += part[(long)c * 3 + 2]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ww += __shfl_xor(ww, o); wg += __shfl_xor(wg, o); gg += __shfl_xor(gg, o); }
    if (threadIdx.x != 0) return;
    // the sums are rounded to float and combined in float arithmetic in the reference's order (ortho_grad.py:69-77), so
    // that a tensor whose sums are exact (a scalar parameter) gets bit-identical coefficients
    const float proj = (float)wg / ((float)ww + ortho_eps);
    const float gnorm = sqrtf((float)gg);
    float scale = 1.f;
    int fl = 0;
    if (rescale) {
        double go2 = gg - 2.0 * (double)proj * wg + (double)proj * (double)proj * ww;
        if (go2 < 0.01 * gg) fl = 1;
        if (go2 < 0) go2 = 0;
        scale = gnorm / (sqrtf((float)go2) + ortho_eps);
    }
    coef[3 * t] = proj; coef[3 * t + 1] = scale; coef[3 * t + 2] = gnorm;
    flag[t] = fl;
}
__global__ __launch_bounds__(256) void og_exact_k(const mm_param_desc* __restrict__ descs, const int* __restrict__ chunk_tensor,
                                                 const long* __restrict__ chunk_off, int chunk_elems, const float* __restrict__ coef,
                                                 const int* __restrict__ flag, float* __restrict__ part2) {
    __shared__ float red[4];
    const int t = chunk_tensor[blockIdx.x];
    if (!flag[t]) return;
    const mm_param_desc d = descs[t];
    const long off = chunk_off[blockIdx.x];
    const long end = min(d.n, off + (long)chunk_elems);
    const float proj = coef[3 * t];
    float s = 0.f;
    for (long i = off + threadIdx.x; i < end; i += 256) {
        const float go = d.g[i] - proj * d.p[i];
        s = fmaf(go, go, s);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part2[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void og_adamw_k(const mm_param_desc* __restrict__ descs, const int* __restrict__ chunk_tensor,
                                                 const long* __restrict__ chunk_off, int chunk_elems, const float* __restrict__ coef,
                                                 const int* __restrict__ flag, const float* __restrict__ part2,
                                                 const int* __restrict__ tcs, float ortho_eps,
                                                 float lr, float b1, float b2, float eps, float wd, float bc1, float bc2s,
                                                 const float* __restrict__ hyper) {
    if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2s = hyper[2]; }      // step-dependent scalars from device memory (HIP graphs)
    const int blk = blockIdx.x;          // (walking the chunks in reverse, to meet the reduce pass's last reads in the last-level cache: 2 % alone, nothing in the step)
    const int t = chunk_tensor[blk];
    const mm_param_desc d = descs[t];
    const long off = chunk_off[blk];
    const long end = min(d.n, off + (long)chunk_elems);
    const float proj = coef ? coef[3 * t] : 0.f;
    float scale = coef ? coef[3 * t + 1] : 1.f;
    if (coef && flag[t]) {                     // ill-conditioned closed form: |g_orth| from the element-wise sums
        double go2 = 0;
        for (int c = tcs[t]; c < tcs[t + 1]; c++) go2 += part2[c];
        scale = coef[3 * t + 2] / (sqrtf((float)go2) + ortho_eps);
    }
    const float step_size = lr / bc1;
    const float decay = 1.f - lr * wd, rb1 = 1.f - b1, rb2 = 1.f - b2;
    auto upd = [&](float& w, float& g, float& m, float& v) {
        if (coef) g = (g - proj * w) * scale;
        w *= decay;
        m = b1 * m + rb1 * g;
        v = b2 * v + rb2 * g * g;
        const float denom = sqrtf(v) / bc2s + eps;
        w = w - step_size * (m / denom);
    };
    long i0 = off;
    if (((((uintptr_t)d.p) | ((uintptr_t)d.g) | ((uintptr_t)d.m) | ((uintptr_t)d.v)) & 15) == 0 && (((uintptr_t)d.w16) & 7) == 0) {
        const long vend = off + ((end - off) & ~3L);
#pragma unroll 2
        for (long i = off + threadIdx.x * 4; i < vend; i += 1024) {
            // once-touched streams: non-temporal (the weights and their bf16 copies are read again by the next forward)
            float w[4], g[4], m[4], v[4];
            VecIO<float, 4>::load(d.p + i, w); VecIO<float, 4>::load_nt(d.g + i, g);
            VecIO<float, 4>::load_nt(d.m + i, m); VecIO<float, 4>::load_nt(d.v + i, v);
#pragma unroll
            for (int j = 0; j < 4; j++) upd(w[j], g[j], m[j], v[j]);
            if (coef) VecIO<float, 4>::store_nt(d.g + i, g);
            VecIO<float, 4>::store_nt(d.m + i, m); VecIO<float, 4>::store_nt(d.v + i, v);
            VecIO<float, 4>::store(d.p + i, w);
            if (d.w16) {
                VecIO<bf16, 4>::store((bf16*)d.w16 + i, w);
            }
        }
        i0 = vend;
    }
    for (long i = i0 + threadIdx.x; i < end; i += 256) {
        float w = d.p[i], g = d.g[i], m = d.m[i], v = d.v[i];
        upd(w, g, m, v);
        if (coef) d.g[i] = g;
        d.m[i] = m; d.v[i] = v; d.p[i] = w;
        if (d.w16) ((bf16*)d.w16)[i] = __float2bfloat16(w);
    }
}
extern "C" size_t mm_orthograd_adamw_workspace(int64_t n_tensors, int64_t n_chunks) {
    return (n_tensors > 0 && n_chunks > 0) ? (size_t)(4 * n_chunks + 4 * n_tensors) * sizeof(float) : 0;
}
static int og_launch(const mm_param_desc* descs, const int32_t* chunk_tensor, const int64_t* chunk_off,
                     const int32_t* tensor_chunk_start, int64_t n_tensors, int64_t n_chunks, int64_t chunk_elems,
                     float lr, float beta1, float beta2, float adam_eps, float weight_decay, float bc1, float bc2s,
                     const float* hyper, float ortho_eps, int rescale, int do_ortho, void* workspace, size_t workspace_bytes,
                     void* stream) {
    MM_CHECK(descs && chunk_tensor && chunk_off && tensor_chunk_start && workspace, "orthograd_adamw: null pointer");
    MM_CHECK(n_tensors > 0 && n_chunks > 0 && chunk_elems > 0, "orthograd_adamw: bad sizes");
    MM_CHECK(workspace_bytes >= mm_orthograd_adamw_workspace(n_tensors, n_chunks), "orthograd_adamw: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
    float* part2 = part + 3 * n_chunks;
    float* coef = part2 + n_chunks;
    int* flag = (int*)(coef + 3 * n_tensors);
    if (do_ortho) {
        hipLaunchKernelGGL(og_reduce_k, dim3((unsigned)n_chunks), dim3(256), 0, st, descs, (const int*)chunk_tensor, (const long*)chunk_off, (int)chunk_elems, part);
        MM_LAUNCH_CHECK();
        hipLaunchKernelGGL(og_finalize_k, dim3((unsigned)n_tensors), dim3(64), 0, st, part, (const int*)tensor_chunk_start, (int)n_tensors, ortho_eps, rescale, coef, flag);
        MM_LAUNCH_CHECK();
        if (rescale) {
            hipLaunchKernelGGL(og_exact_k, dim3((unsigned)n_chunks), dim3(256), 0, st, descs, (const int*)chunk_tensor, (const long*)chunk_off, (int)chunk_elems,
                               (const float*)coef, (const int*)flag, part2);
            MM_LAUNCH_CHECK();
        }
    }
    hipLaunchKernelGGL(og_adamw_k, dim3((unsigned)n_chunks), dim3(256), 0, st, descs, (const int*)chunk_tensor, (const long*)chunk_off, (int)chunk_elems,
                       do_ortho ? coef : nullptr, (const int*)flag, (const float*)part2, (const int*)tensor_chunk_start, ortho_eps, lr, beta1, beta2, adam_eps, weight_decay, bc1, bc2s, hyper);
    MM_LAUNCH_CHECK();
    return MM_OK;
}
extern "C" int mm_orthograd_adamw(const mm_param_desc* descs, const int32_t* chunk_tensor, const int64_t* chunk_off,
                                  const int32_t* tensor_chunk_start, int64_t n_tensors, int64_t n_chunks, int64_t chunk_elems,
                                  float lr, float beta1, float beta2, float adam_eps, float weight_decay, int64_t step,
                                  float ortho_eps, int rescale, int do_ortho, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    MM_CHECK(step >= 1, "orthograd_adamw: step must be >= 1");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    return og_launch(descs, chunk_tensor, chunk_off, tensor_chunk_start, n_tensors, n_chunks, chunk_elems, lr, beta1, beta2,
                     adam_eps, weight_decay, bc1, bc2s, nullptr, ortho_eps, rescale, do_ortho, workspace, workspace_bytes, stream);
}
// The same update with the step-dependent scalars read from device memory: hyper[0] = lr, hyper[1] = 1 - beta1^step,
// hyper[2] = sqrt(1 - beta2^step).  A HIP graph that captured this launch stays valid while the learning-rate schedule and
// the step count move: the host rewrites the three floats before each replay.
extern "C" int mm_orthograd_adamw_dev(const mm_param_desc* descs, const int32_t* chunk_tensor, const int64_t* chunk_off,
                                      const int32_t* tensor_chunk_start, int64_t n_tensors, int64_t n_chunks,
                                      int64_t chunk_elems, const float* hyper, float beta1, float beta2, float adam_eps,
                                      float weight_decay, float ortho_eps, int rescale, int do_ortho, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    MM_CHECK(hyper, "orthograd_adamw_dev: null hyper");
    return og_launch(descs, chunk_tensor, chunk_off, tensor_chunk_start, n_tensors, n_chunks, chunk_elems, 0.f, beta1, beta2,
                     adam_eps, weight_decay, 1.f, 1.f, hyper, ortho_eps, rescale, do_ortho, workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------------
// relu(h)^2 of the decoder MLP (reference models/transformer_decoder.py:161) and its backward
//   fwd: a = max(h,0)^2          bwd: dh = da * 2*max(h,0)       (dh may alias da)
// ------------------------------------------------------------------------------------------------
template <typename T, int VEC, bool BWD>
__global__ __launch_bounds__(256) void relu2_k(const T* __restrict__ h, const T* da, T* out, long nvec, long n) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
        float hv[VEC], ov[VEC];
        VecIO<T, VEC>::load(h + i * VEC, hv);
        if (BWD) VecIO<T, VEC>::load(da + i * VEC, ov);
#pragma unroll
        for (int j = 0; j < VEC; j++) {
            const float r = fmaxf(hv[j], 0.f);
            ov[j] = BWD ? ov[j] * (2.f * r) : r * r;
        }
        VecIO<T, VEC>::store(out + i * VEC, ov);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)                       // ragged tail (n % VEC elements)
        for (long i = nvec * VEC; i < n; i++) {
            const float r = fmaxf(to_f(h[i]), 0.f);
            out[i] = from_f<T>(BWD ? to_f(da[i]) * (2.f * r) : r * r);
        }
}
template <bool BWD>
static int relu2_launch(const void* h, const void* da, void* out, int64_t n, int dtype, hipStream_t st) {
    const bool al = (((uintptr_t)h | (uintptr_t)out | (uintptr_t)da) & 15) == 0;
    if (dtype == MM_F32) {
        const long nvec = al ? n / 4 : 0;
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(std::max<long>(nvec, 1), 256), 256 * 16));
        if (al) hipLaunchKernelGGL((relu2_k<float, 4, BWD>), dim3(grid), dim3(256), 0, st, (const float*)h, (const float*)da, (float*)out, nvec, (long)n);
        else hipLaunchKernelGGL((relu2_k<float, 1, BWD>), dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 4096)), dim3(256), 0, st, (const float*)h, (const float*)da, (float*)out, (long)n, (long)n);
    } else if (dtype == MM_BF16) {
        const long nvec = al ? n / 8 : 0;
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(std::max<long>(nvec, 1), 256), 256 * 16));
        if (al) hipLaunchKernelGGL((relu2_k<bf16, 8, BWD>), dim3(grid), dim3(256), 0, st, (const bf16*)h, (const bf16*)da, (bf16*)out, nvec, (long)n);
        else hipLaunchKernelGGL((relu2_k<bf16, 1, BWD>), dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 4096)), dim3(256), 0, st, (const bf16*)h, (const bf16*)da, (bf16*)out, (long)n, (long)n);
    } else MM_FAIL(MM_ERR_ARG, "relu2: bad dtype %d", dtype);
    MM_LAUNCH_CHECK();
    return MM_OK;
}
extern "C" int mm_relu2_fwd(const void* h, void* a, int64_t n, int dtype, void* stream) {
    MM_CHECK(h && a && n > 0, "relu2_fwd: null pointer or empty");
    return relu2_launch<false>(h, nullptr, a, n, dtype, (hipStream_t)stream);
}
extern "C" int mm_relu2_bwd(const void* h, const void* da, void* dh, int64_t n, int dtype, void* stream) {
    MM_CHECK(h && da && dh && n > 0, "relu2_bwd: null pointer or empty");
    return relu2_launch<true>(h, da, dh, n, dtype, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// out[c][r] (bf16 or f32, row stride ldo) = in[r][c] (f32 master weight [R][C]): the transposed compute-dtype
// operand copy that turns dX = dY . W into a K-contiguous product.  64 x 64 tiles through LDS.
// ------------------------------------------------------------------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void cast_transpose_k(const float* __restrict__ in, TO* __restrict__ out, int R, int C, long ldo) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int r = r0 + ty + 4 * i, c = c0 + tx;
        tile[ty + 4 * i][tx] = (r < R && c < C) ? in[(long)r * C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int c = c0 + ty + 4 * i, r = r0 + tx;
        if (c < C && r < R) out[(long)c * ldo + r] = from_f<TO>(tile[tx][ty + 4 * i]);
    }
}
// bf16 output, 16-byte accesses on both sides: float4 loads along C, 8 packed bf16 per store along R
__global__ __launch_bounds__(256) void cast_transpose_v_k(const float* __restrict__ in, bf16* __restrict__ out, int R, int C, long ldo) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int q = threadIdx.x & 15, rr = threadIdx.x >> 4;             // 16 float4 per tile row, 16 rows per pass
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r = r0 + rr + 16 * i, c = c0 + q * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R && c < C) v = *reinterpret_cast<const float4*>(in + (long)r * C + c);     // C % 4 == 0: all or nothing
        tile[rr + 16 * i][q * 4 + 0] = v.x; tile[rr + 16 * i][q * 4 + 1] = v.y;
        tile[rr + 16 * i][q * 4 + 2] = v.z; tile[rr + 16 * i][q * 4 + 3] = v.w;
    }
    __syncthreads();
    const int ch = threadIdx.x & 7, cc = threadIdx.x >> 3;              // 8 chunks of 8 rows per output row, 32 output rows per pass
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int c = c0 + cc + 32 * i, r = r0 + ch * 8;
        if (c < C && r < R) {                                           // R % 8 == 0: all or nothing
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = tile[ch * 8 + j][cc + 32 * i];
            VecIO<bf16, 8>::store(out + (long)c * ldo + r, v);
        }
    }
}
extern "C" int mm_cast_transpose(const float* in, void* out, int64_t R, int64_t C, int64_t ldo, int out_dtype, void* stream) {
    MM_CHECK(in && out && R > 0 && C > 0 && ldo >= R, "cast_transpose: bad arguments");
    const dim3 grid((unsigned)cdiv(C, 64), (unsigned)cdiv(R, 64));
    hipStream_t st = (hipStream_t)stream;
    const bool vec = out_dtype == MM_BF16 && C % 4 == 0 && R % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0;
    if (vec) hipLaunchKernelGGL(cast_transpose_v_k, grid, dim3(256), 0, st, in, (bf16*)out, (int)R, (int)C, (long)ldo);
    else if (out_dtype == MM_BF16) hipLaunchKernelGGL(cast_transpose_k<bf16>, grid, dim3(256), 0, st, in, (bf16*)out, (int)R, (int)C, (long)ldo);
    else if (out_dtype == MM_F32) hipLaunchKernelGGL(cast_transpose_k<float>, grid, dim3(256), 0, st, in, (float*)out, (int)R, (int)C, (long)ldo);
    else MM_FAIL(MM_ERR_ARG, "cast_transpose: bad dtype %d", out_dtype);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

// Multi-tensor form: ONE launch refreshes every transposed bf16 operand copy after an optimizer step (40 launches of 6-25 us
// inside backward before: the small ones are latency bound).  descs[t] = one [R][C] weight (f32 master, or its bf16 operand copy
// when the model keeps one: half the read bytes, same values) and its [C][ldo] destination; tile_desc[b] = tensor of block b.
// Contract (checked by the caller that builds the table): C % 8 == 0, R % 8 == 0, ldo % 8 == 0, 16-byte aligned bases.
__global__ __launch_bounds__(256) void cast_transpose_multi_k(const mm_transpose_desc* __restrict__ descs, const int* __restrict__ tile_desc) {
    __shared__ float tile[64][65];
    const mm_transpose_desc d = descs[tile_desc[blockIdx.x]];
    const int R = (int)d.R, C = (int)d.C;
    const int tcols = (C + 63) >> 6, t = blockIdx.x - (int)d.tile0;
    const int r0 = (t / tcols) * 64, c0 = (t % tcols) * 64;
    if (d.src_dtype == MM_BF16) {
        const bf16* in = (const bf16*)d.src;
        const int q = threadIdx.x & 7, rr = threadIdx.x >> 3;             // 8 chunks of 8 bf16 per tile row, 32 rows per pass
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int r = r0 + rr + 32 * i, c = c0 + q * 8;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (r < R && c < C) VecIO<bf16, 8>::load(in + (long)r * C + c, v);
#pragma unroll
            for (int j = 0; j < 8; j++) tile[rr + 32 * i][q * 8 + j] = v[j];
        }
    } else {
        const float* in = (const float*)d.src;
        const int q = threadIdx.x & 15, rr = threadIdx.x >> 4;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = r0 + rr + 16 * i, c = c0 + q * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < R && c < C) v = *reinterpret_cast<const float4*>(in + (long)r * C + c);
            tile[rr + 16 * i][q * 4 + 0] = v.x; tile[rr + 16 * i][q * 4 + 1] = v.y;
            tile[rr + 16 * i][q * 4 + 2] = v.z; tile[rr + 16 * i][q * 4 + 3] = v.w;
        }
    }
    __syncthreads();
    bf16* out = (bf16*)d.dst;
    const int ch = threadIdx.x & 7, cc = threadIdx.x >> 3;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int c = c0 + cc + 32 * i, r = r0 + ch * 8;
        if (c < C && r < R) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = tile[ch * 8 + j][cc + 32 * i];
            VecIO<bf16, 8>::store(out + (long)c * d.ldo + r, v);
        }
    }
}
extern "C" int mm_cast_transpose_multi(const mm_transpose_desc* descs, const int32_t* tile_desc, int64_t n_tiles, void* stream) {
    MM_CHECK(descs && tile_desc && n_tiles > 0, "cast_transpose_multi: bad arguments");
    hipLaunchKernelGGL(cast_transpose_multi_k, dim3((unsigned)n_tiles), dim3(256), 0, (hipStream_t)stream, descs, (const int*)tile_desc);
    MM_LAUNCH_CHECK();
    return MM_OK;
}

// ------------------------------------------------------------------------------------------------
// DropPath factor per sample (reference models/convnext.py:13-17): out[i] = floor(keep + u[i]) / keep, u = the
// torch.rand draw.  One launch instead of ATen's add, floor and div.
// ------------------------------------------------------------------------------------------------
__global__ void droppath_scale_k(const float* __restrict__ u, float* __restrict__ out, int n, float keep) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = floorf(keep + u[i]) / keep;
}
extern "C" int mm_droppath_scale(const float* u, float* out, int64_t n, float keep, void* stream) {
    MM_CHECK(u && out && n > 0 && keep > 0.f && keep <= 1.f, "droppath_scale: bad arguments");
    hipLaunchKernelGGL(droppath_scale_k, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, u, out, (int)n, keep);
    MM_LAUNCH_CHECK();
    return MM_OK;
}
