"""KV-cached greedy decode (new API; the reference re-runs the full decoder every step,
train_autoregressive.py:95-116).  Legality: prompt rows never see target rows, target rows are
causal, RoPE positions are absolute, so K/V of earlier positions do not depend on later tokens.

Prefill = one prefix-LM pass (K3) over [memory ; nmr ; BOS]; every further token is one decode step per layer:
LayerNorm -> fused QKV projection (M = batch rows) -> K4 (rotate, append to cache, attend) -> out projection ->
MLP.  Batch-sharded across GPUs with no collective: each rank owns its rows and its cache."""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch

from . import functional as FN
from . import ops


class KVCache:
    def __init__(self, n_layers: int, B: int, H: int, Smax: int, d: int, dtype, device):
        self.k = [torch.empty((B, H, Smax, d), dtype=dtype, device=device) for _ in range(n_layers)]
        self.v = [torch.empty((B, H, Smax, d), dtype=dtype, device=device) for _ in range(n_layers)]
        self.Smax = Smax


def _layer_weights(layer, dt):
    """GEMM operands of one layer from the operand cache (functional.cached_operand: built once per parameter version,
    refreshed by the optimizer -- nothing is re-cast or re-concatenated per generate() call).  Weights in the compute
    dtype, biases as float32 (the libmmhip epilogues add them to the float32 accumulator)."""
    co = FN.cached_operand
    return dict(
        wqkv=co((layer.q.weight, layer.k.weight, layer.v.weight), dt),
        bqkv=co((layer.q.bias, layer.k.bias, layer.v.bias), torch.float32),
        bqkv16=co((layer.q.bias, layer.k.bias, layer.v.bias), dt),
        wo=co((layer.out.weight,), dt), bo=layer.out.bias.detach(),
        w1=co((layer.ffn_w1.weight,), dt), b1=layer.ffn_w1.bias.detach(), b116=co((layer.ffn_w1.bias,), dt),
        w2=co((layer.ffn_w2.weight,), dt), b2=layer.ffn_w2.bias.detach(), b216=co((layer.ffn_w2.bias,), dt))


_DECODE_GEMM = os.environ.get("MM_DECODE_GEMM", "auto")    # A/B knob: "mm" = every product on libmmhip, "lib" = none


def _lin(x2, w, b, b16=None, act=0, res=None, out_dtype=None, square=False):
    """x2 [M,K] @ w[N,K]^T + b (+ relu^2) (+ res).  At decode shapes (M = batch rows, about 1000) the products are short
    and latency-bound; measured on MI355X (bf16, M = 1024): the libmmhip GEMM with its fused bias + residual epilogue wins
    for the square E x E product (13.0 vs 15.0 us), the library wins for N = 3E / 4E and K = 4E (13.5-21.5 vs 22-45 us),
    so those run there with the bias in the library's epilogue and relu^2 / the residual as one libmmhip / in-place pass.
    b is the float32 bias (libmmhip epilogue), b16 its compute-dtype copy from the operand cache (library epilogue)."""
    use_mm = _DECODE_GEMM == "mm" or (_DECODE_GEMM == "auto" and square and x2.dtype == torch.bfloat16)
    if use_mm:
        return ops.gemm(x2, w, x2.shape[0], w.shape[0], w.shape[1], bias=b, act=act, res=res, out_dtype=out_dtype)
    y = torch.addmm(b16 if b16 is not None else b.to(x2.dtype), x2, w.t())
    if act == ops.ACT_RELU2:
        y = ops.relu2_fwd(y)
    if res is not None:
        y.add_(res)
    return y if out_dtype in (None, y.dtype) else y.to(out_dtype)


@torch.no_grad()
def greedy_generate(model, nmr_tokens: Optional[torch.Tensor], ir_data: torch.Tensor, max_len: int = 128, bos_id: int = 12,
                    eos_id: int = 13, early_stop: bool = True, sync_every: int = 8,
                    graph: Optional[bool] = None) -> torch.Tensor:
    """graph=True: replay the decode step from ONE captured HIP graph (position, output column and current ids live in device
    memory and advance inside the graph).  Same kernels, same ids as the eager loop (tested), but OFF by default: on this
    stack (ROCm 7.2) replaying the ~45-node step costs more than issuing it -- nocat shape, 80 steps: B=1024 549k vs 643k
    tokens/s eager, B=128 181k vs 178k -- the decode loop is not host-bound (unlike the small-config train step, where
    graph.CapturedTrainStep pays)."""
    dec = model.decoder
    dt = FN.get_compute_dtype()
    was_training = model.training
    model.eval()
    dev = ir_data.device
    B = nmr_tokens.size(0) if nmr_tokens is not None else ir_data.size(0)
    steps = min(max_len, dec.max_seq_length)                                   # train_autoregressive.py:91
    memory = model.encoder(None, ir_data, None)                                # :83
    prompt = dec.build_prompt(memory, nmr_tokens, dt)
    P = prompt.shape[1]
    H, E = dec.num_heads, dec.embed_dim
    d = E // H
    Smax = P + steps
    if Smax > dec.max_seq_length + 256:
        Smax = dec.max_seq_length + 256            # the reference's length guard (:249) would raise beyond this
        steps = Smax - P
    cache = KVCache(len(dec.layers), B, H, Smax, d, dt, dev)
    weights = [_layer_weights(l, dt) for l in dec.layers]
    w_out, b_out = FN.cached_operand((dec.out.weight,), dt), dec.out.bias.detach()
    b_out16 = FN.cached_operand((dec.out.bias,), dt)
    emb = dec.smiles_embed.weight

    tokens = torch.full((B, steps + 1), bos_id, dtype=torch.long, device=dev)
    # ---- prefill over [prompt ; BOS]
    x = torch.cat([prompt, emb[tokens[:, :1]].to(dt)], dim=1)                  # [B, P+1, E]
    S0 = P + 1
    for li, layer in enumerate(dec.layers):
        w = weights[li]
        cos_t, sin_t = layer.rotary_emb.half_tables(Smax)
        M0 = B * S0
        xa, _, _ = ops.layernorm_fwd(x, layer.attn_norm.weight, layer.attn_norm.bias)
        # the big prefill products stay on the library where it wins (N = 3E, 4E: DESIGN.md 4), the N = E products with
        # their bias + residual run on the large-tile kernel, as in the training forward
        qkv = torch.addmm(w["bqkv16"], xa.view(M0, E), w["wqkv"].t()).view(B, S0, 3, E)
        ops.kv_cache_fill(qkv[:, :, 1], qkv[:, :, 2], cache.k[li], cache.v[li], H, layer.rotary_ndims, cos_t, sin_t)
        a, _ = ops.prefix_attn_fwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], H, layer.rotary_ndims, P, cos_t, sin_t,
                                   layer.use_stablemax)
        x = _lin(a.view(M0, E), w["wo"], w["bo"], res=x.view(M0, E), square=True).view(B, S0, E)
        xf, _, _ = ops.layernorm_fwd(x, layer.mlp_norm.weight, layer.mlp_norm.bias)
        u = ops.relu2_fwd(torch.addmm(w["b116"], xf.view(M0, E), w["w1"].t()))
        x = _lin(u, w["w2"], w["b2"], res=x.view(M0, E), square=True).view(B, S0, E)      # prefill: N = E on the large-tile kernel
    xt, _, _ = ops.layernorm_fwd(x[:, -1].contiguous(), dec.final_norm.weight, dec.final_norm.bias)
    nxt = ops.argmax_rows(_lin(xt, w_out, b_out, b_out16, out_dtype=torch.float32))     # :102
    tokens[:, 1] = nxt
    finished = nxt == eos_id
    n_done = 1
    # ---- decode steps
    def decode_step(cur, pos):
        """One token for every row: cur [B] ids at position pos-1 -> argmax ids at pos (pos: int, or int32 device scalar)."""
        h = emb[cur].to(dt)                                                    # [B, E]
        for li, layer in enumerate(dec.layers):
            w = weights[li]
            cos_t, sin_t = layer.rotary_emb.half_tables(Smax)
            # LN, QKV (+bias), K4, out-proj (+bias +residual fused), LN, W1 (+bias) relu^2, W2 (+bias) += residual; operands
            # and bias copies come from the operand cache (no per-call casts)
            xa, _, _ = ops.layernorm_fwd(h, layer.attn_norm.weight, layer.attn_norm.bias)
            qkv = _lin(xa, w["wqkv"], w["bqkv"], w["bqkv16"])                  # [B, 3E]
            a = ops.decode_attn(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], cache.k[li], cache.v[li], H,
                                layer.rotary_ndims, pos, cos_t, sin_t, layer.use_stablemax)
            h = _lin(a, w["wo"], w["bo"], res=h, square=True)
            xf, _, _ = ops.layernorm_fwd(h, layer.mlp_norm.weight, layer.mlp_norm.bias)
            u = _lin(xf, w["w1"], w["b1"], w["b116"], act=ops.ACT_RELU2)
            h = _lin(u, w["w2"], w["b2"], w["b216"], res=h)
        xt, _, _ = ops.layernorm_fwd(h, dec.final_norm.weight, dec.final_norm.bias)
        return ops.argmax_rows(_lin(xt, w_out, b_out, b_out16, out_dtype=torch.float32))

    use_graph = bool(graph)
    if use_graph and steps > 2:
        cur_s = nxt.clone()
        pos_dev = torch.tensor([P + 1], dtype=torch.int32, device=dev)
        col_dev = torch.tensor([2], dtype=torch.long, device=dev)

        def body():
            new = decode_step(cur_s, pos_dev)
            tokens.index_copy_(1, col_dev, new[:, None])
            finished.logical_or_(new == eos_id)
            cur_s.copy_(new)
            pos_dev.add_(1)
            col_dev.add_(1)

        cur_stream = torch.cuda.current_stream(dev)
        warm = torch.cuda.Stream(device=dev)
        warm.wait_stream(cur_stream)
        with torch.cuda.stream(warm):
            body()                                                             # t = 1, eagerly (library warm-up before capture)
        n_done = 2
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(warm):          # bare capture: torch.cuda.graph() would also synchronize, collect garbage
            g.capture_begin()                  # and empty the allocator cache on every generate() call
            try:
                body()
            finally:
                g.capture_end()
        cur_stream.wait_stream(warm)
        for t in range(2, steps):
            if early_stop and t % sync_every == 0 and bool(finished.all()):    # :113 (checked every few steps)
                break
            g.replay()
            n_done = t + 1
    else:
        for t in range(1, steps):
            if early_stop and t % sync_every == 0 and bool(finished.all()):    # :113 (checked every few steps)
                break
            nxt = decode_step(nxt, P + t)
            tokens[:, t + 1] = nxt
            finished |= nxt == eos_id
            n_done = t + 1
    if was_training:
        model.train()
    return tokens[:, : n_done + 1]


def ids_to_lists(tokens: torch.Tensor, eos_id: int = 13) -> List[List[int]]:
    """Per row: BOS + the ids appended while the row was unfinished (the reference's `generated_sequences`,
    train_autoregressive.py:86-110): everything up to and including the first EOS."""
    out = []
    for row in tokens.tolist():
        seq = [row[0]]
        for tok in row[1:]:
            seq.append(tok)
            if tok == eos_id:
                break
        out.append(seq)
    return out


def ids_to_strings(tokens: torch.Tensor, tokenizer, eos_id: int = 13) -> List[str]:
    """train_autoregressive.py:118-130: cut at the first EOS, drop BOS, tokenizer.decode."""
    res = []
    for seq in ids_to_lists(tokens, eos_id):
        if eos_id in seq:
            seq = seq[: seq.index(eos_id)]
        res.append(tokenizer.decode(seq[1:]))
    return res
