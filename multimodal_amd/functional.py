"""Autograd nodes of the hot path.  Each node's forward and backward are sequences of C-ABI calls
(multimodal_amd/ops.py); PyTorch only owns the tensors.  Activations are channels-last [B, L, C]
inside the encoder.  Compute dtype: float32 (parity mode) or bfloat16 (throughput mode); parameters
always stay float32 masters and are cast per call when the compute dtype is bfloat16."""
from __future__ import annotations

from contextlib import contextmanager
from typing import Optional

import torch

from . import _lib as L
from . import ops

Tensor = torch.Tensor

_compute_dtype = torch.float32


def get_compute_dtype():
    return _compute_dtype


def set_compute_dtype(dtype) -> None:
    global _compute_dtype
    if dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("compute dtype must be torch.float32 or torch.bfloat16")
    _compute_dtype = dtype


@contextmanager
def compute_dtype(dtype):
    prev = get_compute_dtype()
    set_compute_dtype(dtype)
    try:
        yield
    finally:
        set_compute_dtype(prev)


def _w(p: Tensor, dtype) -> Tensor:
    """GEMM operand copy of a float32 master weight in the compute dtype."""
    p = p.detach()
    return p if p.dtype == dtype else p.to(dtype)


def _split_k(M: int, N: int, K: int) -> int:
    """Weight-gradient GEMMs reduce over tokens (K large) into small [M,N] outputs: split K so the chip fills."""
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    if tiles >= 128 or K < 1024:
        return 1
    return int(max(1, min(32, 256 // tiles, K // 512)))


# --------------------------------------------------------------------------------------------------
class StemLN(torch.autograd.Function):
    """Conv1d(1,C0,k4,s4) + LayerNorm, [B,Lin] f32 -> [B,Lo,C0] (reference models/convnext.py:67-72)."""

    @staticmethod
    def forward(ctx, ir, w, b, lw, lb, dtype):
        need = any(ctx.needs_input_grad)
        y, mean, rstd = ops.stem_ln_fwd(ir, w.detach(), b.detach(), lw.detach(), lb.detach(), dtype, save_stats=need)
        if need:
            ctx.save_for_backward(ir, w, b, lw, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        ir, w, b, lw, mean, rstd = ctx.saved_tensors
        dx, dw, db, dlw, dlb = ops.stem_ln_bwd(ir, dy.contiguous(), w.detach(), b.detach(), lw.detach(), mean, rstd,
                                               need_dx=ctx.needs_input_grad[0])
        return dx, dw, db, dlw, dlb, None


class ConvNeXtBlock(torch.autograd.Function):
    """One ConvNeXtBlock1D on channels-last activations (reference models/convnext.py:36-48):
    K1 dwconv7+LN, then pwconv1+GELU and pwconv2+LayerScale+DropPath+residual as two MFMA GEMMs with
    fused epilogues.  dp_scale: per-sample DropPath factor mask/keep [B] f32, or None."""

    @staticmethod
    def forward(ctx, x, dw_w, dw_b, ln_w, ln_b, w1, b1, w2, b2, gamma, dp_scale):
        B, Ln, Cc = x.shape
        M = B * Ln
        train = any(ctx.needs_input_grad)
        dt = x.dtype
        xn, mean, rstd = ops.dwconv7_ln_fwd(x, dw_w.detach(), dw_b.detach(), ln_w.detach(), ln_b.detach(), save_stats=train)
        w1c, w2c = _w(w1, dt), _w(w2, dt)
        h = torch.empty((M, 4 * Cc), dtype=dt, device=x.device) if train else None
        a = ops.gemm(xn.view(M, Cc), w1c, M, 4 * Cc, Cc, bias=b1.detach(), pre=h, act=ops.ACT_GELU)
        z = torch.empty((M, Cc), dtype=dt, device=x.device) if train else None
        g = gamma.detach() if gamma is not None else None
        out = ops.gemm(a, w2c, M, Cc, 4 * Cc, bias=b2.detach(), pre=z, gamma=g, rowscale=dp_scale, rows_per_scale=Ln,
                       res=x.view(M, Cc))
        if train:
            ctx.save_for_backward(x, dw_w, dw_b, ln_w, mean, rstd, xn, h, a, z, w1, w2, gamma, dp_scale)
        return out.view(B, Ln, Cc)

    @staticmethod
    def backward(ctx, dout):
        x, dw_w, dw_b, ln_w, mean, rstd, xn, h, a, z, w1, w2, gamma, dp_scale = ctx.saved_tensors
        B, Ln, Cc = x.shape
        M = B * Ln
        dt = x.dtype
        dout = dout.contiguous().view(M, Cc)
        gvec = gamma.detach() if gamma is not None else torch.ones(Cc, dtype=torch.float32, device=x.device)
        dz, dgamma, db2 = ops.layerscale_bwd(dout, z, gvec, dp_scale, Ln)
        w1c, w2c = _w(w1, dt), _w(w2, dt)
        dw2 = ops.gemm(dz, a, Cc, 4 * Cc, M, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32,
                       split_k=_split_k(Cc, 4 * Cc, M))
        dh = ops.gemm(dz, w2c, M, 4 * Cc, Cc, b_kmajor=True, mul=h, act=ops.ACT_GELU)
        db1 = ops.colsum(dh, M, 4 * Cc)
        dw1 = ops.gemm(dh, xn.view(M, Cc), 4 * Cc, Cc, M, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32,
                       split_k=_split_k(4 * Cc, Cc, M))
        dxn = ops.gemm(dh, w1c, M, Cc, 4 * Cc, b_kmajor=True)
        dx, ddw, ddb, dlw, dlb = ops.dwconv7_ln_bwd(x, dxn.view(B, Ln, Cc), dout.view(B, Ln, Cc), dw_w.detach(),
                                                    dw_b.detach(), ln_w.detach(), mean, rstd)
        return dx, ddw, ddb, dlw, dlb, dw1, db1, dw2, db2, (dgamma if gamma is not None else None), None


class LNLinear(torch.autograd.Function):
    """LayerNorm over C followed by a Linear on `fold` consecutive rows folded together:
       fold=2, drop_tail: the downsample LN + Conv1d(k=2,s=2) (models/convnext.py:75-82), weight [Cout, fold*C]
                          (the conv weight permuted to tap-major);
       fold=1: the final norm + per-position head (models/convnext.py:130-134).
    x: [B,L,C] -> [B, L//fold, Cout]."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w, b, fold):
        B, Ln, Cc = x.shape
        Lk = (Ln // fold) * fold
        train = any(ctx.needs_input_grad)
        dt = x.dtype
        xn, mean, rstd = ops.layernorm_fwd(x, ln_w.detach(), ln_b.detach(), save_stats=train,
                                           drop_tail_to=(Lk if Lk != Ln else None))
        M, K = B * (Lk // fold), fold * Cc
        N = w.shape[0]
        out = ops.gemm(xn.view(M, K), _w(w, dt), M, N, K, bias=b.detach())
        if train:
            ctx.save_for_backward(x, ln_w, mean, rstd, xn, w)
            ctx.fold = fold
        return out.view(B, Lk // fold, N)

    @staticmethod
    def backward(ctx, dout):
        x, ln_w, mean, rstd, xn, w = ctx.saved_tensors
        fold = ctx.fold
        B, Ln, Cc = x.shape
        Lk = (Ln // fold) * fold
        M, K = B * (Lk // fold), fold * Cc
        N = w.shape[0]
        dt = x.dtype
        dout = dout.contiguous().view(M, N)
        db = ops.colsum(dout, M, N)
        dw = ops.gemm(dout, xn.view(M, K), N, K, M, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32,
                      split_k=_split_k(N, K, M))
        dxn = ops.gemm(dout, _w(w, dt), M, K, N, b_kmajor=True)
        dx, dlw, dlb = ops.layernorm_bwd(x, dxn.view(B, Lk, Cc), None, ln_w.detach(), mean, rstd,
                                         drop_tail_to=(Lk if Lk != Ln else None))
        return dx, dlw, dlb, dw, db, None


class LayerNorm(torch.autograd.Function):
    """nn.LayerNorm over the last dim (models/transformer_decoder.py:122,:160,:306)."""

    @staticmethod
    def forward(ctx, x, w, b):
        train = any(ctx.needs_input_grad)
        x = x.contiguous()
        y, mean, rstd = ops.layernorm_fwd(x, w.detach(), b.detach(), save_stats=train)
        if train:
            ctx.save_for_backward(x, w, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        dx, dw, db = ops.layernorm_bwd(x, dy.contiguous(), None, w.detach(), mean, rstd)
        return dx, dw, db


class PrefixAttention(torch.autograd.Function):
    """K3: partial RoPE + prefix-LM masked softmax/stablemax attention on the fused projection
    qkv [B,S,3,E] (models/transformer_decoder.py:129-155; mask of :295-298).  Returns [B,S,E]."""

    @staticmethod
    def forward(ctx, qkv, cos_t, sin_t, H, rot, prompt_len, stablemax):
        train = ctx.needs_input_grad[0]
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        # q,k are rotated ONCE, in place (qkv is the projection's own output; nothing else reads it un-rotated)
        ops.rope_inplace(q, H, rot, cos_t, sin_t)
        ops.rope_inplace(k, H, rot, cos_t, sin_t)
        o, lse = ops.prefix_attn_fwd(q, k, v, H, 0, prompt_len, cos_t, sin_t, stablemax, save_lse=train)
        if train:
            ctx.save_for_backward(qkv, o, lse, cos_t, sin_t)
            ctx.cfg = (H, rot, prompt_len, stablemax)
        return o

    @staticmethod
    def backward(ctx, d_o):
        qkv, o, lse, cos_t, sin_t = ctx.saved_tensors
        H, rot, prompt_len, stablemax = ctx.cfg
        dqkv = ops.prefix_attn_bwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], o, d_o.contiguous(), lse, H, 0, prompt_len,
                                   cos_t, sin_t, stablemax)          # saved q,k are already rotated
        ops.rope_inplace(dqkv[:, :, 0], H, rot, cos_t, sin_t, inverse=True)
        ops.rope_inplace(dqkv[:, :, 1], H, rot, cos_t, sin_t, inverse=True)
        return dqkv, None, None, None, None, None, None
