"""KV-cached greedy decode (new API; the reference re-runs the full decoder every step,
train_autoregressive.py:95-116).  Legality: prompt rows never see target rows, target rows are
causal, RoPE positions are absolute, so K/V of earlier positions do not depend on later tokens.

Prefill = one prefix-LM pass (K3) over [memory ; nmr ; BOS]; every further token is one decode step per layer:
LayerNorm -> fused QKV projection (M = batch rows) -> K4 (rotate, append to cache, attend) -> out projection ->
MLP.  Batch-sharded across GPUs with no collective: each rank owns its rows and its cache."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import functional as FN
from . import ops


class KVCache:
    def __init__(self, n_layers: int, B: int, H: int, Smax: int, d: int, dtype, device):
        self.k = [torch.empty((B, H, Smax, d), dtype=dtype, device=device) for _ in range(n_layers)]
        self.v = [torch.empty((B, H, Smax, d), dtype=dtype, device=device) for _ in range(n_layers)]
        self.Smax = Smax


def _layer_weights(layer, dt):
    return dict(
        wqkv=torch.cat((layer.q.weight, layer.k.weight, layer.v.weight), 0).to(dt).contiguous(),
        bqkv=torch.cat((layer.q.bias, layer.k.bias, layer.v.bias), 0).to(dt).contiguous(),
        wo=layer.out.weight.to(dt), bo=layer.out.bias.to(dt),
        w1=layer.ffn_w1.weight.to(dt), b1=layer.ffn_w1.bias.to(dt),
        w2=layer.ffn_w2.weight.to(dt), b2=layer.ffn_w2.bias.to(dt))


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
    w_out, b_out = dec.out.weight.to(dt), dec.out.bias.to(dt)
    emb = dec.smiles_embed.weight

    tokens = torch.full((B, steps + 1), bos_id, dtype=torch.long, device=dev)
    # ---- prefill over [prompt ; BOS]
    x = torch.cat([prompt, emb[tokens[:, :1]].to(dt)], dim=1)                  # [B, P+1, E]
    S0 = P + 1
    for li, layer in enumerate(dec.layers):
        w = weights[li]
        cos_t, sin_t = layer.rotary_emb.half_tables(Smax)
        xa, _, _ = ops.layernorm_fwd(x, layer.attn_norm.weight, layer.attn_norm.bias)
        qkv = F.linear(xa, w["wqkv"], w["bqkv"]).view(B, S0, 3, E)
        ops.kv_cache_fill(qkv[:, :, 1], qkv[:, :, 2], cache.k[li], cache.v[li], H, layer.rotary_ndims, cos_t, sin_t)
        a, _ = ops.prefix_attn_fwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], H, layer.rotary_ndims, P, cos_t, sin_t,
                                   layer.use_stablemax)
        x = F.linear(a, w["wo"], w["bo"]) + x
        xf, _, _ = ops.layernorm_fwd(x, layer.mlp_norm.weight, layer.mlp_norm.bias)
        x = x + F.linear(F.relu(F.linear(xf, w["w1"], w["b1"])).square(), w["w2"], w["b2"])
    xt, _, _ = ops.layernorm_fwd(x[:, -1].contiguous(), dec.final_norm.weight, dec.final_norm.bias)
    nxt = ops.argmax_rows(F.linear(xt, w_out, b_out).float())                  # :102
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
            xa, _, _ = ops.layernorm_fwd(h, layer.attn_norm.weight, layer.attn_norm.bias)
            qkv = F.linear(xa, w["wqkv"], w["bqkv"])                           # [B, 3E]
            a = ops.decode_attn(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], cache.k[li], cache.v[li], H,
                                layer.rotary_ndims, pos, cos_t, sin_t, layer.use_stablemax)
            h = F.linear(a, w["wo"], w["bo"]) + h
            xf, _, _ = ops.layernorm_fwd(h, layer.mlp_norm.weight, layer.mlp_norm.bias)
            h = h + F.linear(F.relu(F.linear(xf, w["w1"], w["b1"])).square(), w["w2"], w["b2"])
        xt, _, _ = ops.layernorm_fwd(h, dec.final_norm.weight, dec.final_norm.bias)
        return ops.argmax_rows(F.linear(xt, w_out, b_out).float())

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
