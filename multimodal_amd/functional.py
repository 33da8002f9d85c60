"""Autograd nodes of the hot path.  Each node's forward and backward are sequences of C-ABI calls
(multimodal_amd/ops.py); PyTorch only owns the tensors.  Activations are channels-last [B, L, C]
inside the encoder.  Compute dtype: float32 (parity mode) or bfloat16 (throughput mode); parameters
always stay float32 masters and are cast per call when the compute dtype is bfloat16."""
from __future__ import annotations

import os
import weakref
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


# Which kernels run the decoder's dense products.  "auto" (default): per product, whichever measured faster on MI355X -- the
# library (hipBLASLt through torch.mm/addmm) for the short-K products (q/k/v, d_model -> d_ffn, their dX at N = d_ffn, all
# weight gradients), libmmhip for the N = d_model products with their fused epilogues (DESIGN.md 4 has the table).  "own":
# EVERY product on libmmhip (large-tile kernel for K-contiguous operands, gemm_k's transposed-read tiles for the weight
# gradients, relu^2 and its derivative in GEMM epilogues): the complete hot path without the library, kept tested and timed
# (bench.py: train_own_gemms) so that the per-product choice stays a measured one.
_gemm_backend = os.environ.get("MM_GEMM_BACKEND", "auto")
# the out-projection dX product on libmmhip in the "auto" back end too: 67.7 vs 76.2 us alone, 25.10 vs 25.15 ms per step (two pairs)
_DO_OWN = os.environ.get("MM_DO_OWN", "1") == "1"


def set_gemm_backend(name: str) -> None:
    global _gemm_backend
    if name not in ("auto", "own"):
        raise ValueError("gemm backend must be 'auto' or 'own'")
    _gemm_backend = name


def get_gemm_backend() -> str:
    return _gemm_backend


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


# Weight / bias gradients are off the critical dX chain.  For the layers whose backward runs LAST (the first decoder
# layers) they can be issued on a side stream: they then overlap the encoder's backward, which consists of many small,
# latency-bound launches that leave most of the chip idle.  The autograd engine joins the side stream at the end of the
# backward pass (queue_callback), and dp / the optimizer join it before they touch gradients.
_side_streams = {}
_join_queued = False


def _runs_beside(main: torch.cuda.Stream, cand: torch.cuda.Stream, cycles: int = 400_000) -> bool:
    """Do kernels on `cand` run while kernels on `main` run?  HIP maps streams onto a few hardware queues (4 a process on
    the target boxes) in creation order: two streams that land on the same queue are serialised.  Measured: after
    torch.distributed has created RCCL's streams the first new stream shared the main stream's queue and the whole
    parameter-gradient overlap was gone (34.3 instead of 28.4 ms per step, no two kernels ever concurrent in the trace)."""
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(main):
        torch.cuda._sleep(1000)                    # warm the spin kernel
        e0.record()
        torch.cuda._sleep(cycles)
        e1.record()                                # one spin alone
        cand.wait_event(e1)
        with torch.cuda.stream(cand):
            torch.cuda._sleep(cycles)
            done = torch.cuda.Event()
            done.record()
        torch.cuda._sleep(cycles)
        main.wait_event(done)
        e2.record()                                # two spins: beside each other or one after the other
    e2.synchronize()
    return e1.elapsed_time(e2) < 1.5 * e0.elapsed_time(e1)


def side_stream(device) -> torch.cuda.Stream:
    s = _side_streams.get(device)
    if s is None:
        main = torch.cuda.current_stream(device)
        s = torch.cuda.Stream(device=device)
        if os.environ.get("MM_SIDE_STREAM_PROBE", "1") == "1" and not torch.cuda.is_current_stream_capturing():
            tried = [s]                            # keep the rejected ones alive so that new streams get other queues
            for _ in range(6):
                if _runs_beside(main, s):
                    break
                s = torch.cuda.Stream(device=device)
                tried.append(s)
            else:
                s = tried[0]                       # none found (one hardware queue?): correctness does not depend on it
            _rejected_streams[device] = [t for t in tried if t is not s]
        _side_streams[device] = s
    return s


_rejected_streams = {}
_comm_streams = {}
_flushers = []          # weak references to OrthoGradAdamW.flush_deferred of optimizers with updates deferred into the next forward


def register_flusher(bound_method) -> None:
    _flushers.append(weakref.WeakMethod(bound_method))


def flush_deferred_updates() -> None:
    """Complete every optimizer update still deferred into 'the next forward' (optim.OrthoGradAdamW.defer_into_next_forward) and
    make the current stream wait for it: for code that reads parameters without running the model's forward (generate())."""
    for r in list(_flushers):
        f = r()
        if f is None:
            _flushers.remove(r)
        else:
            f(wait=True)


def comm_stream(device) -> torch.cuda.Stream:
    """The stream gradient all-reduces are issued on (dp.DataParallel): probed like the side stream, against BOTH the main and
    the side stream, so that with 4 hardware queues compute, parameter gradients and RCCL's kernels each get their own."""
    s = _comm_streams.get(device)
    if s is None:
        main, side = main_stream(device), side_stream(device)
        s = torch.cuda.Stream(device=device)
        if os.environ.get("MM_SIDE_STREAM_PROBE", "1") == "1" and not torch.cuda.is_current_stream_capturing():
            tried = [s]
            for _ in range(8):
                if _runs_beside(main, s) and _runs_beside(side, s):
                    break
                s = torch.cuda.Stream(device=device)
                tried.append(s)
            else:
                s = tried[0]
            _rejected_streams.setdefault(device, []).extend(t for t in tried if t is not s)
        _comm_streams[device] = s
    return s


_main_of_side = {}


def enter_side(device):
    """(main, side) for a node that is about to issue work on the side stream; remembers which stream is 'main'."""
    main = main_stream(device)
    side = side_stream(device)
    _main_of_side[device] = main
    side.wait_stream(main)
    return main, side


def main_stream(device) -> torch.cuda.Stream:
    """The stream the model runs on: the current stream, unless the caller is itself inside a side-stream section
    (a gradient-ready callback fired from there), in which case the stream that section was entered from."""
    cur = torch.cuda.current_stream(device)
    s = _side_streams.get(device)
    if s is not None and cur == s:
        return _main_of_side.get(device, torch.cuda.default_stream(device))
    return cur


_keepalive = []      # tensors the side stream still reads (kept referenced until the main stream has joined it)


def _order_accumulation(pairs, main, side) -> None:
    """pairs: (parameter, gradient this node returns for it, produced on the side stream).  Autograd keeps a returned
    gradient as it is when the parameter's .grad is unset (no kernel), but ACCUMULATES into an existing .grad
    (`p.grad += g`, gradient accumulation, a second backward, zero_grad(set_to_none=False)) on the MAIN stream, and the
    engine orders that only after the node's own main-stream work.  In that case the main stream must wait for the side
    stream here, before the node returns."""
    if any(g is not None and p.grad is not None for p, g in pairs):
        main.wait_stream(side)


def _new_graph_root() -> None:
    """A forward pass in training mode starts: any earlier backward is over, whether it completed or raised, so a join
    flag left set by a backward that raised (its end-of-backward callback never ran) must not suppress the next one."""
    global _join_queued
    _join_queued = False


def join_side_stream() -> None:
    """Make the current stream wait for every side-stream gradient launch issued so far."""
    global _join_queued
    _join_queued = False
    # memory released from here on returns to the main stream's pool, and the main stream now waits for the side
    # stream before running anything new: no reuse can overtake a side-stream reader
    _keepalive.clear()
    for dev, s in _side_streams.items():
        main_stream(dev).wait_stream(s)



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
        w1c, w2c = cached_operand((w1,), dt), cached_operand((w2,), dt)
        h = torch.empty((M, 4 * Cc), dtype=dt, device=x.device) if train else None
        a = ops.gemm(xn.view(M, Cc), w1c, M, 4 * Cc, Cc, bias=b1.detach(), pre=h, act=ops.ACT_GELU)
        z = torch.empty((M, Cc), dtype=dt, device=x.device) if train else None
        g = gamma.detach() if gamma is not None else None
        out = ops.gemm(a, w2c, M, Cc, 4 * Cc, bias=b2.detach(), pre=z, gamma=g, rowscale=dp_scale, rows_per_scale=Ln,
                       res=x.view(M, Cc))
        if train:
            _new_graph_root()
            ctx.save_for_backward(x, dw_w, dw_b, ln_w, mean, rstd, xn, h, a, z, w1, w2, gamma, dp_scale)
            ctx.b1 = b1
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
        w1c, w2c = cached_operand((w1,), dt), cached_operand((w2,), dt)
        dh = ops.gemm(dz, w2c, M, 4 * Cc, Cc, b_kmajor=True, mul=h, act=ops.ACT_GELU)
        dxn = ops.gemm(dh, w1c, M, Cc, 4 * Cc, b_kmajor=True)        # last use of the block's weights (see DecoderLayer)
        # weight / bias gradients on the side stream (see DecoderLayer): the dX chain below does not wait for them
        global _join_queued
        main, side = enter_side(x.device)
        with torch.cuda.stream(side):
            # the two weight gradients go straight into their gradient-bucket slots when dp.GradBuckets marked them fresh (as the
            # decoder's do, _weight_grad): finish() then has nothing to copy for them
            s2, s1 = _fresh_slot(w2), _fresh_slot(w1)
            dw2 = ops.gemm(dz, a, Cc, 4 * Cc, M, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, out=s2,
                           split_k=_split_k(Cc, 4 * Cc, M), ws_tag="gemm_side")
            db1 = ops.colsum(dh, M, 4 * Cc, "side")
            dw1 = ops.gemm(dh, xn.view(M, Cc), 4 * Cc, Cc, M, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, out=s1,
                           split_k=_split_k(4 * Cc, Cc, M), ws_tag="gemm_side")
            if s2 is not None:
                dw2 = _slot_written(w2, s2)
            if s1 is not None:
                dw1 = _slot_written(w1, s1)
        _keepalive.append((dz, dh, a, xn))     # incl. saved activations (freed when this node is released)
        for t in (dw2, db1, dw1):
            if t is not None:
                t.record_stream(main)
        _order_accumulation(((w2, dw2), (ctx.b1, db1), (w1, dw1)), main, side)
        if not _join_queued:
            _join_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(join_side_stream)
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


class Linear(torch.autograd.Function):
    """x W^T + b on libmmhip for the "own" GEMM backend (final `out` projection transformer_decoder.py:307, memory_proj
    :259): forward with the bias in the epilogue, dX on the transposed operand copy, dW through _weight_grad, db a column sum."""

    @staticmethod
    def forward(ctx, x, w, b):
        dt = x.dtype
        K, N = w.shape[1], w.shape[0]
        x2 = x.reshape(-1, K).contiguous()
        y = ops.gemm(x2, cached_operand((w,), dt), x2.shape[0], N, K, bias=b.detach())
        ctx.save_for_backward(x2, w)
        ctx.b = b
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dt = x2.dtype
        K, N = w.shape[1], w.shape[0]
        dy2 = dy.reshape(-1, N).contiguous()
        M = dy2.shape[0]
        dx = ops.gemm(dy2, cached_operand((w,), dt, transpose=True), M, K, N).view(ctx.xshape) if ctx.needs_input_grad[0] else None
        db = ops.colsum(dy2, M, N)
        dw = _weight_grad(w, dy2, x2)
        return dx, dw, db


class Embedding(torch.autograd.Function):
    """nn.Embedding lookup in the compute dtype (models/transformer_decoder.py:256,:272).  Backward: the weight gradient
    is onehot(idx)^T . dY as ONE library GEMM (exact products, float32 sums, deterministic) instead of ATen's sort-based
    scatter kernels (0.25 ms per step at the config shape); it lands in the parameter's bucket slot when one is fresh."""

    @staticmethod
    def forward(ctx, weight, idx, dtype):
        ctx.save_for_backward(idx, weight)
        return weight.detach()[idx].to(dtype)

    @staticmethod
    def backward(ctx, dy):
        idx, weight = ctx.saved_tensors
        V, E = weight.shape
        rows = idx.numel()
        dy2 = dy.contiguous().view(rows, E)
        onehot = torch.zeros((rows, V), dtype=dy2.dtype, device=dy2.device)
        onehot.scatter_(1, idx.reshape(rows, 1), 1.0)
        return _weight_grad(weight, onehot, dy2), None, None


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


# --------------------------------------------------------------------------------------------------
# Decoder layer as ONE autograd node.
# --------------------------------------------------------------------------------------------------
_operand_cache = {}


def cached_operand(params, dtype, transpose: bool = False) -> Tensor:
    """Compute-dtype GEMM operand of one float32 master parameter, or of several stacked along dim 0 (the fused
    q/k/v projection).  Rebuilt only when a parameter's version counter or storage moves; OrthoGradAdamW bumps the
    counters of everything it updates, so the copy is made once per optimizer step instead of once per use."""
    if len(params) == 1 and params[0].dtype == dtype and not transpose:
        return params[0].detach()
    key = tuple(id(p) for p in params) + (dtype, transpose)
    stamp = tuple((p._version, p.data_ptr()) for p in params)
    ent = _operand_cache.get(key)
    if ent is not None and ent[0] == stamp and all(r() is p for r, p in zip(ent[2], params)):
        return ent[1]
    shape = (sum(p.shape[0] for p in params),) + tuple(params[0].shape[1:])
    if transpose:                      # [in, out] copy: makes dX = dY . W a K-contiguous (NT) product as well
        shape = (shape[1], shape[0])
    dev = params[0].device
    buf = ent[1] if ent is not None and tuple(ent[1].shape) == shape and ent[1].device == dev else \
        torch.empty(shape, dtype=dtype, device=dev)
    off = 0
    with torch.no_grad():
        for p in params:
            if transpose:
                ops.cast_transpose(p.detach(), buf[:, off: off + p.shape[0]])
            else:
                buf[off: off + p.shape[0]].copy_(p)
            off += p.shape[0]
    _operand_cache[key] = (stamp, buf, tuple(weakref.ref(p) for p in params))
    if not transpose and dtype == torch.bfloat16 and all(p.dtype == torch.float32 for p in params):
        off = 0
        for p in params:                       # contiguous per-parameter row blocks: the optimizer can refresh them
            _shadow_index[id(p)] = (key, off)
            off += p.shape[0]
    return buf


# ---- transposed operand copies, refreshed in one launch ------------------------------------------------------------------
_tr_table = {}          # (addresses) -> device descriptor table of mm_cast_transpose_multi
_TR_MULTI = os.environ.get("MM_TRANSPOSE_MULTI", "1") == "1"      # A/B: 0 = each copy rebuilt on first use inside backward


def refresh_transposed_operands() -> int:
    """Called by OrthoGradAdamW after an update: every STALE transposed bf16 operand copy whose parameters live on the current
    device is rebuilt by ONE mm_cast_transpose_multi launch (from the bf16 operand copy when the model keeps one, else from
    the f32 master) and re-stamped.  Returns the number of copies refreshed.  Copies it cannot take (odd sizes, other dtypes)
    stay stale and are rebuilt on use by cached_operand, as before."""
    if not _TR_MULTI:
        return 0
    import numpy as np
    todo = []
    for key, ent in _operand_cache.items():
        if key[-1] is not True or key[-2] != torch.bfloat16:
            continue
        ps = [r() for r in ent[2]]
        if any(p is None for p in ps) or not ps[0].is_cuda or ps[0].device != ent[1].device:
            continue
        stamp = tuple((p._version, p.data_ptr()) for p in ps)
        if stamp == ent[0]:
            continue
        buf = ent[1]
        ok = buf.data_ptr() % 16 == 0 and buf.stride(0) % 8 == 0 and all(
            p.dim() == 2 and p.dtype == torch.float32 and p.is_contiguous() and p.shape[0] % 8 == 0 and p.shape[1] % 8 == 0
            and p.data_ptr() % 16 == 0 for p in ps)
        if ok and tuple(buf.shape) == (ps[0].shape[1], sum(p.shape[0] for p in ps)):
            todo.append((key, ent, ps, stamp))
    if not todo:
        return 0
    dev = todo[0][1][1].device
    todo = [t for t in todo if t[1][1].device == dev]
    rows, tiles = [], 0
    for key, ent, ps, stamp in todo:
        off = 0
        for p in ps:
            sh = shadow_of(p)
            src = sh if (sh is not None and sh.is_contiguous() and sh.data_ptr() % 16 == 0) else p
            R, Cc = p.shape
            rows.append((src.data_ptr(), ent[1].data_ptr() + 2 * off, R, Cc, ent[1].stride(0), tiles,
                         1 if src.dtype == torch.bfloat16 else 0))
            tiles += ((R + 63) // 64) * ((Cc + 63) // 64)
            off += R
    tkey = tuple(r[:2] + (r[6],) for r in rows)
    tab = _tr_table.get(dev)
    if tab is None or tab[0] != tkey:
        if torch.cuda.is_current_stream_capturing():
            return 0                                   # no table yet: leave the copies to cached_operand
        host = np.array(rows, dtype=np.int64)
        host[:, 6] = np.where(host[:, 6] == 1, ops.L.MM_BF16, ops.L.MM_F32)
        tile_desc = np.concatenate([np.full(((r[2] + 63) // 64) * ((r[3] + 63) // 64), i, dtype=np.int32) for i, r in enumerate(rows)])
        tab = (tkey, torch.from_numpy(host.view(np.uint8).reshape(-1).copy()).to(dev), torch.from_numpy(tile_desc).to(dev), tiles)
        _tr_table[dev] = tab
    ops.L.check(ops.L.lib().mm_cast_transpose_multi(ops.L.ptr(tab[1]), ops.L.ptr(tab[2]), tab[3], ops.L.cur_stream()),
                "mm_cast_transpose_multi")
    for key, ent, ps, stamp in todo:
        _operand_cache[key] = (stamp, ent[1], ent[2])
    return len(todo)


_shadow_index = {}


def shadow_of(p: Tensor) -> Optional[Tensor]:
    """The bf16 operand copy of parameter p inside the cache (a contiguous view of p's shape), or None."""
    hit = _shadow_index.get(id(p))
    if hit is None:
        return None
    key, off = hit
    ent = _operand_cache.get(key)
    if ent is None or not any(r() is p for r in ent[2]) or ent[1].device != p.device:
        return None
    return ent[1][off: off + p.shape[0]]


def shadows_written(params) -> None:
    """OrthoGradAdamW refreshed the copies of `params` while updating them: re-stamp the cache entries all of whose
    parameters were refreshed, so the next forward takes the copies as current."""
    done = set(id(p) for p in params)
    for key in set(_shadow_index[i][0] for i in done if i in _shadow_index):
        ent = _operand_cache.get(key)
        if ent is None:
            continue
        ps = [r() for r in ent[2]]
        if all(q is not None and id(q) in done for q in ps):
            _operand_cache[key] = (tuple((q._version, q.data_ptr()) for q in ps), ent[1], ent[2])


_DW_SMALL_OWN = os.environ.get("MM_DW_SMALL_OWN", "0") == "1"
_DW_OWN = os.environ.get("MM_DW_OWN", "0")


def _mm_f32(a: Tensor, b: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """a @ b with a float32 result (bf16 operands accumulate and land in f32: weight gradients never see bf16).
    Called as _mm_f32(dy2.t(), x2): the weight-gradient product dW[n][k] = sum_m dy2[m][n] x2[m][k]."""
    # "auto": the library.  MM_DW_SMALL_OWN=1 sends small outputs under a long reduction (the square out-projection: 49 tiles
    # of 256 x 256) to the large-tile weight-gradient kernel, which splits K over the chip: 81 vs 102 us alone, but INSIDE the
    # step its 250 512-thread workgroups crowd the main stream's kernels (25.70 vs 25.49 ms per step, three A/B pairs): off.
    small = _DW_SMALL_OWN and a.dtype == torch.bfloat16 and a.shape[0] * b.shape[1] <= (1 << 22) and b.shape[0] >= 4096 and \
        a.shape[0] >= 256 and b.shape[1] >= 256 and a.shape[0] % 8 == 0 and b.shape[1] % 8 == 0
    # MM_DW_OWN (A/B): "ffn" = the two d_model x d_ffn weight gradients on the libmmhip tile (240 / 218 us alone against the
    # library's 265 / 231 since the DMA of that kernel goes out through inline asm), "all" = every large one
    if _DW_OWN != "0" and a.dtype == torch.bfloat16 and b.shape[0] >= 4096 and a.shape[0] % 8 == 0 and b.shape[1] % 8 == 0:
        big = a.shape[0] * b.shape[1] >= (1 << 23)
        small = small or _DW_OWN == "all" or (_DW_OWN == "ffn" and big) or (_DW_OWN == "ffn+o" and (big or a.shape[0] == b.shape[1]))
    if _gemm_backend == "own" or small:
        dy2 = a.t()                               # [rows, n] with unit inner stride again
        n, k, rows = dy2.shape[1], b.shape[1], b.shape[0]
        tag = "gemm_side" if torch.cuda.current_stream(b.device) == _side_streams.get(b.device) else "gemm"
        return ops.gemm(dy2, b, n, k, rows, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, out=out,
                        split_k=_split_k(n, k, rows), ws_tag=tag)
    if a.dtype == torch.float32:
        return torch.mm(a, b, out=out) if out is not None else torch.mm(a, b)
    if out is not None:
        return torch.mm(a, b, out_dtype=torch.float32, out=out)
    return torch.mm(a, b, out_dtype=torch.float32)


def _linear_res(x2: Tensor, w: Tensor, b: Tensor, res: Tensor) -> Tensor:
    """res + x2 W^T + b.  bf16: the large-tile libmmhip GEMM with bias and residual fused in its epilogue (these are
    the N = d_model products, where its tile choice beats the library's); f32: library GEMM + in-place add."""
    dt = x2.dtype
    if dt == torch.bfloat16 or _gemm_backend == "own":
        return ops.gemm(x2, cached_operand((w,), dt), x2.shape[0], w.shape[0], w.shape[1], bias=b.detach(), res=res)
    return torch.addmm(b.detach(), x2, w.detach().t()).add_(res)


def _fresh_slot(p: Tensor) -> Optional[Tensor]:
    """The parameter's gradient-bucket slot if a gradient may be WRITTEN there now (dp.GradBuckets.zero() marked it fresh and
    nothing has been accumulated since), else None."""
    g = getattr(p, "_mm_view", None)
    if g is not None and getattr(p, "_mm_fresh", False) and p.grad is None and g.dtype == torch.float32 and g.shape == p.shape:
        return g
    return None


def _slot_written(p: Tensor, g: Tensor) -> None:
    """The gradient of p has been produced in its bucket slot g: point .grad at it, tell the bucket; the autograd node returns
    None for this parameter."""
    p.grad = g
    p._mm_fresh = False
    ready = getattr(p, "_mm_ready", None)
    if ready is not None:
        ready(p)
    return None


def _weight_grad(p: Tensor, dy2: Tensor, x2: Tensor) -> Optional[Tensor]:
    """dW = dy2^T @ x2 (float32).  When dp.GradBuckets.zero() has marked the parameter fresh and nothing has been
    accumulated yet, the GEMM writes the parameter's bucket slot directly, .grad is pointed at it and the bucket is told
    the gradient is ready; the node then returns None for this parameter.  Otherwise the gradient goes back to autograd."""
    g = getattr(p, "_mm_view", None)
    if g is not None and getattr(p, "_mm_fresh", False) and p.grad is None and g.dtype == torch.float32 and g.shape == p.shape:
        _mm_f32(dy2.t(), x2, out=g)
        p.grad = g
        p._mm_fresh = False
        ready = getattr(p, "_mm_ready", None)
        if ready is not None:
            ready(p)
        return None
    return _mm_f32(dy2.t(), x2)


def _stacked_weight_grads(params, dy2: Tensor, x2: Tensor):
    """Weight gradients of several projections that share an input (q/k/v): ONE GEMM into their adjacent bucket slots
    when dp.GradBuckets laid them out that way and all are fresh; else one GEMM each."""
    slots = [getattr(p, "_mm_slot", None) for p in params]
    n = params[0].numel()
    if all(s is not None for s in slots) and all(getattr(p, "_mm_fresh", False) and p.grad is None for p in params) \
            and all(s[0] is slots[0][0] and s[1] == slots[0][1] + i * n and p.numel() == n
                    for i, (s, p) in enumerate(zip(slots, params))):
        flat, off = slots[0]
        out = flat[off: off + len(params) * n].view(len(params) * params[0].shape[0], params[0].shape[1])
        _mm_f32(dy2.t(), x2, out=out)
        for p in params:
            p.grad = p._mm_view
            p._mm_fresh = False
            ready = getattr(p, "_mm_ready", None)
            if ready is not None:
                ready(p)
        return [None] * len(params)
    r = params[0].shape[0]
    return [_weight_grad(p, dy2[:, i * r: (i + 1) * r], x2) for i, p in enumerate(params)]


# The library GEMMs are stream-K kernels: by default their workgroups wait for each other's partial tiles whenever the
# library decides to split tiles across workgroups.  Two such launches from two streams can starve each other -- at
# B*S = 16380 rows and above (K of the weight-gradient products) the main-stream and side-stream GEMMs deadlocked
# reproducibly, while at the config batch (10080 rows: the dW products happen to be one workgroup per tile) they never
# did.  multimodal_amd/__init__.py therefore sets TENSILE_STREAMK_DATA_PARALLEL=1 (read by hipBLASLt on every call):
# whole tiles per workgroup, no inter-workgroup waits, so any launch can always finish whatever runs beside it.  Measured:
# no loss at batch 32, +12 % at batch 256.  If the user overrides that variable, side-stream weight gradients are limited
# to the row range that was validated with stream-K splitting active.
def _defer_rows_limit() -> int:
    v = os.environ.get("MM_DEFER_MAX_ROWS")
    if v is not None:
        return int(v)
    return (1 << 62) if os.environ.get("TENSILE_STREAMK_DATA_PARALLEL") == "1" else 12288


class DecoderLayer(torch.autograd.Function):
    """DecoderPromptLayer.forward (reference models/transformer_decoder.py:115-166) as one node:
         xa = LN1(x); qkv = xa Wqkv^T + b; RoPE(q,k) in place; o = K3(q,k,v); x1 = x + o Wo^T + bo
         xf = LN2(x1); h = xf W1^T + b1; a = relu(h)^2; out = x1 + a W2^T + b2
    Dense projections are plain library GEMMs (hipBLASLt through torch.mm/addmm) on cached bf16 operands; LayerNorm,
    RoPE, attention, relu^2 and the bias-gradient column sums are libmmhip kernels; residual gradients are folded into
    the LayerNorm backward kernels; weight gradients are produced in float32."""

    @staticmethod
    def forward(ctx, x, cos_t, sin_t, H, rot, prompt_len, stablemax, defer_dw, ln1w, ln1b, qw, qb, kw, kb, vw, vb, ow, ob,
                ln2w, ln2b, w1, b1, w2, b2):
        B, S, D = x.shape
        M = B * S
        dt = x.dtype
        train = any(ctx.needs_input_grad)
        x = x.contiguous()
        xa, mean1, rstd1 = ops.layernorm_fwd(x, ln1w.detach(), ln1b.detach(), save_stats=train)
        own = _gemm_backend == "own"
        if own:
            qkv = ops.gemm(xa.view(M, D), cached_operand((qw, kw, vw), dt), M, 3 * D, D,
                           bias=cached_operand((qb, kb, vb), torch.float32))
        else:
            qkv = torch.addmm(cached_operand((qb, kb, vb), dt), xa.view(M, D), cached_operand((qw, kw, vw), dt).t())
        qkv = qkv.view(B, S, 3, D)
        ops.rope_inplace(qkv.view(B, S, 3 * D)[:, :, :2 * D], 2 * H, rot, cos_t, sin_t)      # q and k in one launch
        o, lse = ops.prefix_attn_fwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], H, 0, prompt_len, cos_t, sin_t, stablemax,
                                     save_lse=train)
        x1 = _linear_res(o.view(M, D), ow, ob, x.view(M, D))
        xf, mean2, rstd2 = ops.layernorm_fwd(x1, ln2w.detach(), ln2b.detach(), save_stats=train)
        if own:                               # bias, the saved pre-activation and relu^2 in the product's epilogue
            h = torch.empty((M, w1.shape[0]), dtype=dt, device=x.device)
            a = ops.gemm(xf, cached_operand((w1,), dt), M, w1.shape[0], D, bias=b1.detach(), pre=h, act=ops.ACT_RELU2)
        else:
            h = torch.addmm(cached_operand((b1,), dt), xf, cached_operand((w1,), dt).t())
            a = ops.relu2_fwd(h)
        out = _linear_res(a, w2, b2, x1)
        if train:
            ctx.save_for_backward(x, mean1, rstd1, xa, qkv, o, lse, x1, mean2, rstd2, xf, h, a, cos_t, sin_t,
                                  ln1w, qw, kw, vw, ow, ln2w, w1, w2)
            ctx.cfg = (H, rot, prompt_len, stablemax, bool(defer_dw))
            ctx.biases = (qb, kb, vb, ob, b1, b2)
            _new_graph_root()
        return out.view(B, S, D)

    @staticmethod
    def backward(ctx, dout):
        (x, mean1, rstd1, xa, qkv, o, lse, x1, mean2, rstd2, xf, h, a, cos_t, sin_t,
         ln1w, qw, kw, vw, ow, ln2w, w1, w2) = ctx.saved_tensors
        H, rot, prompt_len, stablemax, defer_dw = ctx.cfg
        B, S, D = x.shape
        M = B * S
        dt = x.dtype
        F4 = h.shape[1]
        dout = dout.contiguous().view(M, D)
        # ---- the dX chain (critical path)
        own = _gemm_backend == "own"
        esz = 2 if dt == torch.bfloat16 else 4
        if own:                          # the relu^2 derivative in the product's epilogue; the bias gradient is a column sum
            dh = ops.gemm(dout, cached_operand((w2,), dt, transpose=True), M, F4, D, mul=h, act=ops.ACT_RELU2)
            db1 = None
        else:
            if dt == torch.bfloat16:
                # dh = dout . W2 with W2 [d_model][d_ffn]: as written the library runs an NN product (reduction index strided
                # in W2) at 0.37 PF/s (553 us at the config batch); on a transposed operand copy it is an NT product at 1.0 PF/s
                dh = torch.mm(dout, cached_operand((w2,), dt, transpose=True).t())
            else:
                dh = torch.mm(dout, w2.detach())
            if (F4 * esz) % 16 == 0:
                _, db1 = ops.relu2_bwd_colsum_(h, dh)                 # bias gradient of ffn_w1 in the same pass
            else:
                ops.relu2_bwd_(h, dh)
                db1 = None
        if dt == torch.bfloat16 or own:  # N = d_model, K = d_ffn: NT product on the transposed operand copy
            dxf = ops.gemm(dh, cached_operand((w1,), dt, transpose=True), M, D, F4)
        else:
            dxf = torch.mm(dh, w1.detach())
        if ops.layernorm_bwd_sums_ok(D):
            # + residual gradient; column sums of dout and dx1 = bias gradients of ffn_w2 and of the out projection
            dx1, dln2w, dln2b, db2, dbo = ops.layernorm_bwd_sums(x1, dxf, dout, ln2w.detach(), mean2, rstd2)
        else:
            dx1, dln2w, dln2b = ops.layernorm_bwd(x1, dxf, dout, ln2w.detach(), mean2, rstd2)  # + residual gradient
            db2 = dbo = None
        if own or (_DO_OWN and dt == torch.bfloat16):
            d_o = ops.gemm(dx1, cached_operand((ow,), dt, transpose=True), M, D, D).view(B, S, D)
        else:
            d_o = torch.mm(dx1, cached_operand((ow,), dt)).view(B, S, D)
        dqkv = ops.prefix_attn_bwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], o, d_o, lse, H, 0, prompt_len, cos_t, sin_t,
                                   stablemax)                                               # saved q,k are rotated
        ops.rope_inplace(dqkv.view(B, S, 3 * D)[:, :, :2 * D], 2 * H, rot, cos_t, sin_t, inverse=True)
        dqkv2 = dqkv.view(M, 3 * D)
        xa2 = xa.view(M, D)
        # last use of this layer's weights: issued before the parameter gradients, because their completion may trigger
        # the optimizer update of these very weights (dp.DataParallel.attach_optimizer)
        if dt == torch.bfloat16 or own:  # N = d_model: NT product on the transposed stacked copy (large-tile kernel)
            dxa = ops.gemm(dqkv2, cached_operand((qw, kw, vw), dt, transpose=True), M, D, 3 * D)
        else:
            dxa = torch.mm(dqkv2, cached_operand((qw, kw, vw), dt))

        # ---- parameter gradients (off the critical path)
        def param_grads(tag, db1=db1, db2=db2, dbo=dbo):
            if db2 is None:
                db2 = ops.colsum(dout, M, D, tag)
            dw2 = _weight_grad(w2, dout, a)
            if db1 is None:
                db1 = ops.colsum(dh, M, F4, tag)
            dw1 = _weight_grad(w1, dh, xf)
            if dbo is None:
                dbo = ops.colsum(dx1, M, D, tag)
            dwo = _weight_grad(ow, dx1, o.view(M, D))
            dbqkv = ops.colsum(dqkv2, M, 3 * D, tag)
            dqw, dkw, dvw = _stacked_weight_grads((qw, kw, vw), dqkv2, xa2)
            return db2, dw2, db1, dw1, dbo, dwo, dbqkv, dqw, dkw, dvw

        if defer_dw and M <= _defer_rows_limit():
            global _join_queued
            main, side = enter_side(x.device)
            with torch.cuda.stream(side):
                grads = param_grads("side")
            # everything the side stream still reads: this call's temporaries and the saved activations, whose memory
            # the allocator would otherwise hand back to the main stream as soon as this node is released
            _keepalive.append((dout, dh, dx1, dqkv, a, xf, o, xa))
            for t in grads:
                if t is not None:
                    t.record_stream(main)
            g_db2, g_dw2, g_db1, g_dw1, g_dbo, g_dwo, g_dbqkv, g_dqw, g_dkw, g_dvw = grads
            qb, kb, vb, ob, b1, b2 = ctx.biases
            _order_accumulation(((w2, g_dw2), (b2, g_db2), (w1, g_dw1), (b1, g_db1), (ow, g_dwo), (ob, g_dbo), (qw, g_dqw),
                                 (kw, g_dkw), (vw, g_dvw), (qb, g_dbqkv), (kb, g_dbqkv), (vb, g_dbqkv)), main, side)
            if not _join_queued:
                _join_queued = True
                torch.autograd.Variable._execution_engine.queue_callback(join_side_stream)
        else:
            grads = param_grads("ws")
        db2, dw2, db1, dw1, dbo, dwo, dbqkv, dqw, dkw, dvw = grads
        dx, dln1w, dln1b = ops.layernorm_bwd(x, dxa.view(B, S, D), dx1.view(B, S, D), ln1w.detach(), mean1, rstd1)
        return (dx, None, None, None, None, None, None, None, dln1w, dln1b, dqw, dbqkv[:D], dkw, dbqkv[D:2 * D], dvw,
                dbqkv[2 * D:], dwo, dbo, dln2w, dln2b, dw1, db1, dw2, db2)
