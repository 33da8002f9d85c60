"""Thin Python wrappers over the C ABI (include/mmhip.h): shape/dtype/contiguity checks live here,
PyTorch only provides device memory and the current stream.  No fallbacks: CPU tensors raise."""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Tuple

import torch

from . import _lib as L

Tensor = torch.Tensor


def _chk(t: Tensor, name: str, dtype=None) -> Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor (the hot path has no CPU fallback)")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


def _f32(t: Tensor, name: str) -> Tensor:
    return _chk(t, name, torch.float32)


_ws_cache = {}


def workspace(nbytes: int, device, tag: str = "ws") -> Tensor:
    """Grow-only byte workspace per (device, tag, current stream); the caller owns it, the library never allocates.  Keyed by
    stream: kernels of two streams may run beside each other (generate()'s two half-batch loops) and must not share scratch."""
    key = (device, tag, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


# --------------------------------------------------------------------------- K1
def dwconv7_ln_fwd(x: Tensor, w: Tensor, b: Tensor, ln_w: Tensor, ln_b: Tensor, eps: float = 1e-5,
                   layout: int = L.LAYOUT_BLC, save_stats: bool = False):
    """x: [B,L,C] (BLC) or [B,C,L] (BCL) -> y [B,L,C] (+ mean, rstd [B*L])."""
    _chk(x, "x")
    if layout == L.LAYOUT_BLC:
        B, Ln, Cc = x.shape
    else:
        B, Cc, Ln = x.shape
    w2 = _f32(w, "dwconv.weight").view(Cc, 7)
    y = torch.empty((B, Ln, Cc), dtype=x.dtype, device=x.device)
    mean = rstd = None
    if save_stats:
        mean = torch.empty(B * Ln, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
    L.check(L.lib().mm_dwconv7_ln_fwd(L.ptr(x), L.ptr(w2), L.ptr(_f32(b, "b")), L.ptr(_f32(ln_w, "ln_w")),
                                      L.ptr(_f32(ln_b, "ln_b")), L.ptr(y), L.ptr(mean), L.ptr(rstd), B, Cc, Ln, eps,
                                      L.dtype_code(x), layout, L.cur_stream()), "mm_dwconv7_ln_fwd")
    return y, mean, rstd


def dwconv7_ln_bwd(x: Tensor, dy: Tensor, dres: Optional[Tensor], w: Tensor, b: Tensor, ln_w: Tensor, mean: Tensor,
                   rstd: Tensor):
    _chk(x, "x"); _chk(dy, "dy", x.dtype)
    B, Ln, Cc = x.shape
    dx = torch.empty_like(x)
    dw = torch.empty((Cc, 1, 7), dtype=torch.float32, device=x.device)
    db = torch.empty(Cc, dtype=torch.float32, device=x.device)
    dlw = torch.empty_like(db)
    dlb = torch.empty_like(db)
    nbytes = L.lib().mm_dwconv7_ln_bwd_workspace(B, Cc, Ln)
    ws = workspace(nbytes, x.device)
    if dres is not None:
        _chk(dres, "dres", x.dtype)
    L.check(L.lib().mm_dwconv7_ln_bwd(L.ptr(x), L.ptr(dy), L.ptr(dres), L.ptr(_f32(w, "w").view(Cc, 7)), L.ptr(b),
                                      L.ptr(ln_w), L.ptr(mean), L.ptr(rstd), L.ptr(dx), L.ptr(dw), L.ptr(db), L.ptr(dlw),
                                      L.ptr(dlb), L.ptr(ws), nbytes, B, Cc, Ln, L.dtype_code(x), L.cur_stream()),
            "mm_dwconv7_ln_bwd")
    return dx, dw, db, dlw, dlb


# --------------------------------------------------------------------------- stem
def stem_ln_fwd(x: Tensor, w: Tensor, b: Tensor, ln_w: Tensor, ln_b: Tensor, out_dtype, eps: float = 1e-5,
                save_stats: bool = False):
    """x: f32 [B,Lin] -> y [B,Lo,C0]."""
    _f32(x, "ir")
    B, Lin = x.shape
    C0 = w.shape[0]
    Lo = (Lin - 4) // 4 + 1
    y = torch.empty((B, Lo, C0), dtype=out_dtype, device=x.device)
    mean = rstd = None
    if save_stats:
        mean = torch.empty(B * Lo, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
    L.check(L.lib().mm_stem_ln_fwd(L.ptr(x), L.ptr(_f32(w, "w").view(C0, 4)), L.ptr(_f32(b, "b")), L.ptr(_f32(ln_w, "ln_w")),
                                   L.ptr(_f32(ln_b, "ln_b")), L.ptr(y), L.ptr(mean), L.ptr(rstd), B, Lin, C0, eps,
                                   L.dtype_code(y), L.cur_stream()), "mm_stem_ln_fwd")
    return y, mean, rstd


def stem_ln_bwd(x: Tensor, dy: Tensor, w: Tensor, b: Tensor, ln_w: Tensor, mean: Tensor, rstd: Tensor, need_dx: bool):
    B, Lin = x.shape
    C0 = w.shape[0]
    _chk(dy, "dy")
    dx = torch.zeros_like(x) if need_dx else None
    dw = torch.empty((C0, 1, 4), dtype=torch.float32, device=x.device)
    db = torch.empty(C0, dtype=torch.float32, device=x.device)
    dlw = torch.empty_like(db)
    dlb = torch.empty_like(db)
    nbytes = L.lib().mm_stem_ln_bwd_workspace(B, Lin, C0)
    ws = workspace(nbytes, x.device)
    L.check(L.lib().mm_stem_ln_bwd(L.ptr(x), L.ptr(dy), L.ptr(w.view(C0, 4)), L.ptr(b), L.ptr(ln_w), L.ptr(mean), L.ptr(rstd),
                                   L.ptr(dx), L.ptr(dw), L.ptr(db), L.ptr(dlw), L.ptr(dlb), L.ptr(ws), nbytes, B, Lin, C0,
                                   L.dtype_code(dy), L.cur_stream()), "mm_stem_ln_bwd")
    return dx, dw, db, dlw, dlb


# --------------------------------------------------------------------------- LayerNorm
def layernorm_fwd(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5, save_stats: bool = False,
                  drop_tail_to: Optional[int] = None):
    """x: [..., C] -> LN over C.  With drop_tail_to=Lout and x [G,Lin,C]: y is [G,Lout,C] (rows l>=Lout dropped)."""
    _chk(x, "x")
    Cc = x.shape[-1]
    if drop_tail_to is None:
        rows = x.numel() // Cc
        G, Lin, Lout = 1, rows, rows
        y = torch.empty_like(x)
    else:
        G, Lin, _ = x.shape
        Lout = drop_tail_to
        y = torch.empty((G, Lout, Cc), dtype=x.dtype, device=x.device)
    mean = rstd = None
    if save_stats:
        mean = torch.empty(G * Lout, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
    L.check(L.lib().mm_layernorm_fwd(L.ptr(x), L.ptr(_f32(w, "ln.weight")), L.ptr(_f32(b, "ln.bias")), L.ptr(y), L.ptr(mean),
                                     L.ptr(rstd), G, Lin, Lout, Cc, eps, L.dtype_code(x), L.cur_stream()), "mm_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(x: Tensor, dy: Tensor, dres: Optional[Tensor], w: Tensor, mean: Tensor, rstd: Tensor,
                  drop_tail_to: Optional[int] = None):
    _chk(x, "x"); _chk(dy, "dy", x.dtype)
    Cc = x.shape[-1]
    if drop_tail_to is None:
        rows = x.numel() // Cc
        G, Lin, Lout = 1, rows, rows
    else:
        G, Lin, _ = x.shape
        Lout = drop_tail_to
    dx = torch.empty_like(x)
    dw = torch.empty(Cc, dtype=torch.float32, device=x.device)
    db = torch.empty_like(dw)
    nbytes = L.lib().mm_layernorm_bwd_workspace(G * Lin, Cc)
    ws = workspace(nbytes, x.device)
    if dres is not None:
        _chk(dres, "dres", x.dtype)
    L.check(L.lib().mm_layernorm_bwd(L.ptr(x), L.ptr(dy), L.ptr(dres), L.ptr(w), L.ptr(mean), L.ptr(rstd), L.ptr(dx), L.ptr(dw),
                                     L.ptr(db), L.ptr(ws), nbytes, G, Lin, Lout, Cc, L.dtype_code(x), L.cur_stream()),
            "mm_layernorm_bwd")
    return dx, dw, db


def layernorm_bwd_sums_ok(Cc: int) -> bool:
    return Cc > 512 and Cc % 4 == 0


def layernorm_bwd_sums(x: Tensor, dy: Tensor, dres: Tensor, w: Tensor, mean: Tensor, rstd: Tensor):
    """layernorm_bwd with a residual gradient, plus the column sums of dres and of dx (two bias gradients for free)."""
    _chk(x, "x"); _chk(dy, "dy", x.dtype); _chk(dres, "dres", x.dtype)
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    dx = torch.empty_like(x)
    out = torch.empty((4, Cc), dtype=torch.float32, device=x.device)
    nbytes = L.lib().mm_layernorm_bwd_workspace(rows, Cc)
    ws = workspace(nbytes, x.device)
    L.check(L.lib().mm_layernorm_bwd_sums(L.ptr(x), L.ptr(dy), L.ptr(dres), L.ptr(w), L.ptr(mean), L.ptr(rstd), L.ptr(dx),
                                          L.ptr(out[0]), L.ptr(out[1]), L.ptr(out[2]), L.ptr(out[3]), L.ptr(ws), nbytes, rows, Cc,
                                          L.dtype_code(x), L.cur_stream()), "mm_layernorm_bwd_sums")
    return dx, out[0], out[1], out[2], out[3]


# --------------------------------------------------------------------------- GEMM
ACT_NONE, ACT_GELU, ACT_RELU2 = 0, 1, 2


def gemm(A: Tensor, B: Tensor, M: int, N: int, K: int, *, a_kmajor: bool = False, b_kmajor: bool = False,
         out_dtype=None, bias: Optional[Tensor] = None, pre: Optional[Tensor] = None, mul: Optional[Tensor] = None,
         act: int = ACT_NONE, gamma: Optional[Tensor] = None, rowscale: Optional[Tensor] = None, rows_per_scale: int = 0,
         res: Optional[Tensor] = None, out: Optional[Tensor] = None, split_k: int = 1, ws_tag: str = "gemm") -> Tensor:
    """out[M,N] = epilogue(A[M,K] @ B[N,K]^T).  A is [M,K] (or [K,M] if a_kmajor); B is [N,K] (or [K,N] if b_kmajor).
    Both operands are 2-D views with unit stride in the last dim."""
    for t, nm in ((A, "A"), (B, "B")):
        if not t.is_cuda or t.dim() != 2 or t.stride(1) != 1:
            raise RuntimeError(f"gemm: {nm} must be a 2-D GPU tensor with unit inner stride")
    if A.dtype != B.dtype:
        raise TypeError(f"gemm: operand dtypes differ ({A.dtype} vs {B.dtype})")
    exp_a = (K, M) if a_kmajor else (M, K)
    exp_b = (K, N) if b_kmajor else (N, K)
    if tuple(A.shape) != exp_a or tuple(B.shape) != exp_b:
        raise RuntimeError(f"gemm: shapes {tuple(A.shape)} / {tuple(B.shape)} do not match M={M} N={N} K={K}")
    out_dtype = out_dtype or A.dtype
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=A.device)
    ldo = out.stride(0)
    for t, nm in ((pre, "pre"), (mul, "mul"), (res, "res")):
        if t is not None and (t.dtype != out.dtype or t.stride(-1) != 1 or t.stride(-2) != ldo or t.numel() != M * N):
            raise RuntimeError(f"gemm: {nm} must look like out ([M,N] {out.dtype}, ld {ldo})")
    a = L.GemmArgs()
    a.A, a.B, a.out = L.ptr(A), L.ptr(B), L.ptr(out)
    a.M, a.N, a.K = M, N, K
    a.lda, a.ldb, a.ldo = A.stride(0), B.stride(0), ldo
    a.a_kmajor, a.b_kmajor = int(a_kmajor), int(b_kmajor)
    a.dtype, a.out_dtype = L.dtype_code(A), L.dtype_code(out)
    a.bias = L.ptr(_f32(bias, "bias")) if bias is not None else None
    a.pre = L.ptr(pre); a.mul = L.ptr(mul); a.act = act
    a.gamma = L.ptr(_f32(gamma, "gamma")) if gamma is not None else None
    a.rowscale = L.ptr(_f32(rowscale, "rowscale")) if rowscale is not None else None
    a.rows_per_scale = rows_per_scale
    a.res = L.ptr(res)
    a.split_k = split_k
    if split_k > 1:
        nbytes = split_k * M * N * 4
        ws = workspace(nbytes, A.device, ws_tag)       # a caller on a second stream passes its own tag
        a.workspace, a.workspace_bytes = L.ptr(ws), nbytes
    elif M * N <= (1 << 22) and K >= 2048 and not a_kmajor and not b_kmajor and A.dtype == torch.bfloat16:
        # skinny and deep: let the library split K on its own (full epilogue kept); 8 slabs cover every choice it makes
        nbytes = 8 * M * N * 4
        ws = workspace(nbytes, A.device, ws_tag + "_auto")
        a.workspace, a.workspace_bytes = L.ptr(ws), nbytes
    elif a_kmajor and b_kmajor and A.dtype == torch.bfloat16 and M >= 256 and N >= 256 and K >= 1024 and M * N >= 786432:
        # weight-gradient form on the large-tile kernel: it splits K itself (few tiles, deep reduction); up to 8 slabs, 256 MiB
        nbytes = max(1, min(8, (256 << 20) // (M * N * 4))) * M * N * 4
        ws = workspace(nbytes, A.device, ws_tag + "_tn")
        a.workspace, a.workspace_bytes = L.ptr(ws), nbytes
    L.check(L.lib().mm_gemm(C.byref(a), L.cur_stream()), "mm_gemm")
    return out


def colsum(x: Tensor, M: int, N: int, tag: str = "ws") -> Tensor:
    if not x.is_cuda or x.stride(-1) != 1:
        raise RuntimeError("colsum: expected a GPU tensor with unit inner stride")
    x2 = x.reshape(M, N) if x.is_contiguous() else x
    out = torch.empty(N, dtype=torch.float32, device=x.device)
    nbytes = L.lib().mm_colsum_workspace(M, N)
    ws = workspace(nbytes, x.device, tag)          # a caller on a second stream passes its own tag (own buffer)
    L.check(L.lib().mm_colsum(L.ptr(x2), L.ptr(out), L.ptr(ws), nbytes, M, N, x2.stride(0), L.dtype_code(x), L.cur_stream()),
            "mm_colsum")
    return out


def colsum_noalloc(x: Tensor, out: Tensor) -> Tensor:
    """out[n] = sum_m x[m][n] through mm_colsum_noalloc: no workspace argument, nothing allocated by the call (out is
    the caller's).  For callers under HIP graph capture / serving loops; N <= 8192."""
    if not x.is_cuda or x.dim() != 2 or x.stride(1) != 1:
        raise RuntimeError("colsum_noalloc: expected a 2-D GPU tensor with unit inner stride")
    M, N = x.shape
    if not out.is_cuda or out.dtype != torch.float32 or out.numel() != N or not out.is_contiguous():
        raise RuntimeError("colsum_noalloc: out must be a contiguous float32 GPU tensor of N elements")
    L.check(L.lib().mm_colsum_noalloc(L.ptr(x), L.ptr(out), M, N, x.stride(0), L.dtype_code(x), L.cur_stream()),
            "mm_colsum_noalloc")
    return out


# --------------------------------------------------------------------------- attention
_rope_cache = {}


def rope_tables(inv_freq: Tensor, seq_len: int):
    """cos/sin [S, rot/2] f32, computed as the reference does (transformer_decoder.py:18-22); the second half of
    the reference's table repeats the first (emb = cat(freqs, freqs)), so only rot/2 columns are kept."""
    key = (inv_freq.data_ptr(), inv_freq.numel(), inv_freq.device, seq_len, inv_freq._version)
    hit = _rope_cache.get(key)
    if hit is None:
        t = torch.arange(seq_len, device=inv_freq.device)
        freqs = torch.einsum("i,j->ij", t, inv_freq.float())
        # the entry keeps inv_freq alive: a freed buffer's address could otherwise come back with other contents
        hit = (freqs.cos().contiguous(), freqs.sin().contiguous(), inv_freq)
        if len(_rope_cache) > 64:
            _rope_cache.clear()
        _rope_cache[key] = hit
    return hit[0], hit[1]


def _qkv_geom(q: Tensor, k: Tensor, v: Tensor, H: int):
    """q,k,v: [B,S,E] views with unit inner stride and a common row stride (e.g. slices of one [B,S,3E] tensor)."""
    B, S, E = q.shape
    for t in (q, k, v):
        if not t.is_cuda or t.shape != q.shape or t.dtype != q.dtype or t.stride(2) != 1 or t.stride(1) != q.stride(1) \
                or (B > 1 and t.stride(0) != S * q.stride(1)):
            raise RuntimeError("attention: q,k,v must be GPU [B,S,E] views with unit inner stride and equal row strides")
    if E % H:
        raise RuntimeError("attention: E not divisible by H")
    return B, S, E, E // H, q.stride(1)


def prefix_attn_fwd(q: Tensor, k: Tensor, v: Tensor, H: int, rot: int, prompt_len: int, cos_t: Tensor, sin_t: Tensor,
                    stablemax: bool, save_lse: bool = False):
    B, S, E, d, ld = _qkv_geom(q, k, v, H)
    o = torch.empty((B, S, E), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=q.device) if save_lse else None
    L.check(L.lib().mm_prefix_attn_fwd(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(o), L.ptr(lse), L.ptr(cos_t), L.ptr(sin_t),
                                       B, H, S, d, rot, prompt_len, ld, int(stablemax), L.dtype_code(q), L.cur_stream()),
            "mm_prefix_attn_fwd")
    return o, lse


def prefix_attn_bwd(q: Tensor, k: Tensor, v: Tensor, o: Tensor, d_o: Tensor, lse: Tensor, H: int, rot: int,
                    prompt_len: int, cos_t: Tensor, sin_t: Tensor, stablemax: bool):
    """Returns dqkv [B,S,3,E] (so dq,dk,dv are views with row stride 3E, ready for a fused QKV weight gradient)."""
    B, S, E, d, ld = _qkv_geom(q, k, v, H)
    _chk(o, "o", q.dtype); _chk(d_o, "dO", q.dtype)
    dqkv = torch.empty((B, S, 3, E), dtype=q.dtype, device=q.device)
    nbytes = L.lib().mm_prefix_attn_bwd_workspace(B, H, S)
    ws = workspace(nbytes, q.device, "attn")
    L.check(L.lib().mm_prefix_attn_bwd(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(o), L.ptr(d_o), L.ptr(lse),
                                       dqkv.data_ptr(), dqkv.data_ptr() + E * dqkv.element_size(),
                                       dqkv.data_ptr() + 2 * E * dqkv.element_size(), L.ptr(cos_t), L.ptr(sin_t), L.ptr(ws),
                                       nbytes, B, H, S, d, rot, prompt_len, ld, 3 * E, int(stablemax), L.dtype_code(q),
                                       L.cur_stream()), "mm_prefix_attn_bwd")
    return dqkv


def kv_cache_fill(k: Tensor, v: Tensor, kcache: Tensor, vcache: Tensor, H: int, rot: int, cos_t: Tensor, sin_t: Tensor):
    B, S, E = k.shape
    d = E // H
    Smax = kcache.shape[2]
    for t in (k, v):
        if t.stride(2) != 1 or t.stride(1) != k.stride(1) or (B > 1 and t.stride(0) != S * k.stride(1)):
            raise RuntimeError("kv_cache_fill: k,v must be [B,S,E] views whose batch stride is S * row stride")
    L.check(L.lib().mm_kv_cache_fill(L.ptr(k), L.ptr(v), L.ptr(kcache), L.ptr(vcache), L.ptr(cos_t), L.ptr(sin_t), B, H, S, d,
                                     rot, Smax, k.stride(1), L.dtype_code(k), L.cur_stream()), "mm_kv_cache_fill")


def decode_attn(q: Tensor, k_new: Tensor, v_new: Tensor, kcache: Tensor, vcache: Tensor, H: int, rot: int, pos: int,
                cos_t: Tensor, sin_t: Tensor, stablemax: bool) -> Tensor:
    """q,k_new,v_new: [B,E] views (row stride ld); caches [B,H,Smax,d]; returns o [B,E]."""
    B, E = q.shape
    d = E // H
    o = torch.empty((B, E), dtype=q.dtype, device=q.device)
    if isinstance(pos, Tensor):            # int32 device scalar: position advanced on the device (captured decode step)
        if pos.dtype != torch.int32 or pos.numel() != 1 or not pos.is_cuda:
            raise RuntimeError("decode_attn: a device position must be one int32 element on the GPU")
        L.check(L.lib().mm_decode_attn_dev(L.ptr(q), L.ptr(k_new), L.ptr(v_new), L.ptr(kcache), L.ptr(vcache), L.ptr(o),
                                           L.ptr(cos_t), L.ptr(sin_t), B, H, d, rot, L.ptr(pos), kcache.shape[2], q.stride(0),
                                           int(stablemax), L.dtype_code(q), L.cur_stream()), "mm_decode_attn_dev")
        return o
    L.check(L.lib().mm_decode_attn(L.ptr(q), L.ptr(k_new), L.ptr(v_new), L.ptr(kcache), L.ptr(vcache), L.ptr(o), L.ptr(cos_t),
                                   L.ptr(sin_t), B, H, d, rot, pos, kcache.shape[2], q.stride(0), int(stablemax),
                                   L.dtype_code(q), L.cur_stream()), "mm_decode_attn")
    return o


# --------------------------------------------------------------------------- small pieces
def layerscale_bwd(dout: Tensor, z: Tensor, gamma: Tensor, rowscale: Optional[Tensor], rows_per_scale: int):
    _chk(dout, "dout"); _chk(z, "z", dout.dtype)
    M, Cc = dout.shape
    dz = torch.empty_like(dout)
    dgamma = torch.empty(Cc, dtype=torch.float32, device=dout.device)
    dbias = torch.empty_like(dgamma)
    nbytes = L.lib().mm_layerscale_bwd_workspace(M, Cc)
    ws = workspace(nbytes, dout.device)
    L.check(L.lib().mm_layerscale_bwd(L.ptr(dout), L.ptr(z), L.ptr(_f32(gamma, "gamma")), L.ptr(rowscale), rows_per_scale,
                                      L.ptr(dz), L.ptr(dgamma), L.ptr(dbias), L.ptr(ws), nbytes, M, Cc, L.dtype_code(dout),
                                      L.cur_stream()), "mm_layerscale_bwd")
    return dz, dgamma, dbias


def argmax_rows(logits: Tensor) -> Tensor:
    """logits [rows, V] (unit inner stride) -> int64 [rows]."""
    if not logits.is_cuda or logits.dim() != 2 or logits.stride(1) != 1:
        raise RuntimeError("argmax_rows: expected a 2-D GPU tensor with unit inner stride")
    out = torch.empty(logits.shape[0], dtype=torch.int64, device=logits.device)
    L.check(L.lib().mm_argmax_rows(L.ptr(logits), L.ptr(out), logits.shape[0], logits.shape[1], logits.stride(0),
                                   L.dtype_code(logits), L.cur_stream()), "mm_argmax_rows")
    return out


def relu2_fwd(h: Tensor) -> Tensor:
    """relu(h)**2 (the decoder MLP activation)."""
    _chk(h, "h")
    a = torch.empty_like(h)
    L.check(L.lib().mm_relu2_fwd(L.ptr(h), L.ptr(a), h.numel(), L.dtype_code(h), L.cur_stream()), "mm_relu2_fwd")
    return a


def relu2_bwd_(h: Tensor, da: Tensor) -> Tensor:
    """da <- da * 2*relu(h), in place."""
    _chk(h, "h"); _chk(da, "da", h.dtype)
    if da.numel() != h.numel():
        raise RuntimeError("relu2_bwd: shape mismatch")
    L.check(L.lib().mm_relu2_bwd(L.ptr(h), L.ptr(da), L.ptr(da), h.numel(), L.dtype_code(h), L.cur_stream()), "mm_relu2_bwd")
    return da


def relu2_bwd_colsum_(h: Tensor, da: Tensor):
    """da <- da * 2*relu(h) in place over [M, N]; also returns the column sums of the result (the bias gradient)."""
    _chk(h, "h"); _chk(da, "da", h.dtype)
    if da.shape != h.shape or h.dim() != 2:
        raise RuntimeError("relu2_bwd_colsum: expected two [M, N] tensors")
    M, N = h.shape
    out = torch.empty(N, dtype=torch.float32, device=h.device)
    nbytes = L.lib().mm_colsum_workspace(M, N)
    ws = workspace(nbytes, h.device)
    L.check(L.lib().mm_relu2_bwd_colsum(L.ptr(h), L.ptr(da), L.ptr(da), L.ptr(out), L.ptr(ws), nbytes, M, N, L.dtype_code(h),
                                        L.cur_stream()), "mm_relu2_bwd_colsum")
    return da, out


def droppath_scale(u: Tensor, keep: float) -> Tensor:
    """floor(keep + u) / keep per element of the float32 draw u (any shape) -> flat float32 [u.numel()]."""
    _f32(u, "u")
    out = torch.empty(u.numel(), dtype=torch.float32, device=u.device)
    L.check(L.lib().mm_droppath_scale(L.ptr(u), L.ptr(out), u.numel(), float(keep), L.cur_stream()), "mm_droppath_scale")
    return out


def cast_transpose(w: Tensor, out: Tensor) -> Tensor:
    """out[:, :] (view [C, R] with unit inner stride, bf16 or f32) = w[R, C].T in out's dtype."""
    _f32(w, "w")
    R, Cc = w.shape
    if tuple(out.shape) != (Cc, R) or out.stride(1) != 1 or not out.is_cuda:
        raise RuntimeError("cast_transpose: out must be a GPU [C, R] view with unit inner stride")
    L.check(L.lib().mm_cast_transpose(L.ptr(w), L.ptr(out), R, Cc, out.stride(0), L.dtype_code(out), L.cur_stream()),
            "mm_cast_transpose")
    return out


def rope_inplace(x: Tensor, H: int, rot: int, cos_t: Tensor, sin_t: Tensor, inverse: bool = False) -> None:
    """x: [B,S,E] view (unit inner stride, batch stride S*row stride), rotated in place."""
    if rot == 0:
        return
    B, S, E = x.shape
    if not x.is_cuda or x.stride(2) != 1 or (B > 1 and x.stride(0) != S * x.stride(1)):
        raise RuntimeError("rope_inplace: expected a GPU [B,S,E] view with unit inner stride")
    L.check(L.lib().mm_rope_inplace(L.ptr(x), L.ptr(cos_t), L.ptr(sin_t), B, S, H, E // H, rot, x.stride(1), int(inverse),
                                    L.dtype_code(x), L.cur_stream()), "mm_rope_inplace")
