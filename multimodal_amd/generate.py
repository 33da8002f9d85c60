"""KV-cached greedy decode (new API; the reference re-runs the full decoder every step,
train_autoregressive.py:95-116).  Legality: prompt rows never see target rows, target rows are
causal, RoPE positions are absolute, so K/V of earlier positions do not depend on later tokens.

Prefill = one prefix-LM pass (K3) over [memory ; nmr ; BOS]; every further token is one decode step per layer:
LayerNorm -> fused QKV projection (M = batch rows) -> K4 (rotate, append to cache, attend) -> out projection ->
MLP.  Batch-sharded across GPUs with no collective: each rank owns its rows and its cache."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import torch

from . import _lib as L
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
_DECODE_SPLIT = int(os.environ.get("MM_DECODE_SPLIT", "2"))     # A/B knob: independent decode loops a large batch is cut into (1 = none)
_extra_streams: dict = {}


class KVCacheView:
    """Batch slice of a KVCache (contiguous: the batch is the leading dimension)."""

    def __init__(self, k, v, Smax):
        self.k, self.v, self.Smax = k, v, Smax


def _lin(x2, w, b, b16=None, act=0, res=None, out_dtype=None, square=False):
    """x2 [M,K] @ w[N,K]^T + b (+ relu^2) (+ res).  At decode shapes (M = batch rows, about 1000) the products are short
    and latency-bound; measured on MI355X (bf16, M = 1024): the libmmhip GEMM with its fused bias + residual epilogue wins
    for the square E x E product (13.0 vs 15.0 us), the library wins for N = 3E / 4E and K = 4E (13.5-21.5 vs 22-45 us),
    so those run there with the bias in the library's epilogue and relu^2 / the residual as one libmmhip / in-place pass.
    b is the float32 bias (libmmhip epilogue), b16 its compute-dtype copy from the operand cache (library epilogue)."""
    # `square`: this product runs on libmmhip in bf16 (the E x E products, and d_model -> d_ffn with its relu^2)
    use_mm = _DECODE_GEMM == "mm" or (_DECODE_GEMM == "auto" and square and x2.dtype == torch.bfloat16)
    if use_mm:
        return ops.gemm(x2, w, x2.shape[0], w.shape[0], w.shape[1], bias=b, act=act, res=res, out_dtype=out_dtype)
    y = torch.addmm(b16 if b16 is not None else b.to(x2.dtype), x2, w.t())
    if act == ops.ACT_RELU2:
        y = ops.relu2_fwd(y)
    if res is not None:
        y.add_(res)
    return y if out_dtype in (None, y.dtype) else y.to(out_dtype)


class _StepPlan:
    """One decode step for a fixed batch as a flat list of PREPARED launches.  The step is about 40 short kernels (13-25 us
    products at M = batch rows, 140 us of K4 per layer); issued through the checking wrappers of ops.py (allocation, shape
    checks, pointer conversion: about 30 us of Python per call) the loop was host-bound at 1.37 ms per step for 1.0 ms of
    kernels.  Here every buffer is allocated once, every C-ABI argument list is built once (only the K4 position changes),
    the library products write into preallocated outputs, and the token ids live in a [steps+1, B] table whose row t+1 is
    the argmax kernel's output AND the next step's input: no per-step copies, casts or bookkeeping kernels.
    Per product (measured, bf16, M = 1024, benchmarks/decode_bench.py gemm): fused q/k/v on the library (13.4 vs 17.4 us), the
    square out-projection with bias + residual on libmmhip (13.1 vs 15.3), d_model -> d_ffn with bias + relu^2 on libmmhip
    (18.8 vs 15.2 + 5), d_ffn -> d_model on the library + one in-place add (21.4 vs 25.4).  float32: library products."""

    def __init__(self, dec, weights, cache, w_out, b_out, b_out16, emb_dt, ids, B, dt, dev, Smax):
        lib = L.lib()
        self.stream = L.cur_stream()
        st = self.stream
        E, H = dec.embed_dim, dec.num_heads
        d = E // H
        bf = dt == torch.bfloat16
        code = L.MM_BF16 if bf else L.MM_F32
        new = lambda *shape, dtype=dt: torch.empty(shape, dtype=dtype, device=dev)
        x0, x1, xa, qkv, att, xf = new(B, E), new(B, E), new(B, E), new(B, 3 * E), new(B, E), new(B, E)
        F4 = dec.layers[0].ffn_w1.weight.shape[0]
        u = new(B, F4)
        V = w_out.shape[0]
        logits = new(B, V)
        self.keep = [x0, x1, xa, qkv, att, xf, u, logits, weights, w_out, b_out, b_out16, emb_dt, ids, cache]
        self.x0, self.emb, self.ids = x0, emb_dt, ids
        # K-split workspace of the skinny, deep products (both have K = E): ONE buffer owned by this plan, sized for the
        # largest of them before any argument list takes its address (a grow-only shared workspace would be replaced -- and
        # the first product's slabs written into freed memory -- when a later product asks for more)
        needs = [8 * B * n * 4 for n in (E, F4) if bf and B * n <= (1 << 22) and E >= 2048]
        ws = torch.empty(max(needs), dtype=torch.uint8, device=dev) if needs else None
        self.keep.append(ws)
        calls = []                              # (callable, args) ; args None -> plain Python callable
        self.k4 = []                            # indices of the K4 entries (their position argument changes per step)

        def c(name, *args):
            calls.append((getattr(lib, name), name, list(args)))

        def gemm(a, w, out, bias=None, act=0, res=None):
            g = L.GemmArgs()
            g.A, g.B, g.out = a.data_ptr(), w.data_ptr(), out.data_ptr()
            g.M, g.N, g.K = a.shape[0], w.shape[0], w.shape[1]
            g.lda, g.ldb, g.ldo = a.stride(0), w.stride(0), out.stride(0)
            g.dtype, g.out_dtype = code, code
            g.bias = bias.data_ptr() if bias is not None else None
            g.act = act
            g.res = res.data_ptr() if res is not None else None
            g.split_k = 1
            if bf and g.M * g.N <= (1 << 22) and g.K >= 2048:      # skinny and deep: room for the library's own K split
                assert ws is not None and ws.numel() >= 8 * g.M * g.N * 4
                g.workspace, g.workspace_bytes = ws.data_ptr(), 8 * g.M * g.N * 4
            self.keep.append(g)
            calls.append((lib.mm_gemm, "mm_gemm", [C.byref(g), st]))

        def addmm(bias, a, w, out):
            wt = w.t()
            calls.append((lambda: torch.addmm(bias, a, wt, out=out), None, None))

        def ln(x, norm, y):
            c("mm_layernorm_fwd", x.data_ptr(), norm.weight.data_ptr(), norm.bias.data_ptr(), y.data_ptr(), None, None, 1, B, B, E,
              float(norm.eps), code, st)

        for li, layer in enumerate(dec.layers):
            w = weights[li]
            cos_t, sin_t = layer.rotary_emb.half_tables(Smax)
            self.keep += [cos_t, sin_t]
            ln(x0, layer.attn_norm, xa)
            addmm(w["bqkv16"], xa, w["wqkv"], qkv)
            self.k4.append(len(calls))
            c("mm_decode_attn", qkv.data_ptr(), qkv.data_ptr() + E * qkv.element_size(), qkv.data_ptr() + 2 * E * qkv.element_size(),
              cache.k[li].data_ptr(), cache.v[li].data_ptr(), att.data_ptr(), cos_t.data_ptr(), sin_t.data_ptr(), B, H, d,
              layer.rotary_ndims, 0, Smax, 3 * E, int(layer.use_stablemax), code, st)
            if bf:
                gemm(att, w["wo"], x1, bias=w["bo"], res=x0)
            else:
                addmm(w["bo"], att, w["wo"], x1)
                calls.append((lambda: x1.add_(x0), None, None))
            ln(x1, layer.mlp_norm, xf)
            if bf:
                gemm(xf, w["w1"], u, bias=w["b1"], act=ops.ACT_RELU2)
            else:
                addmm(w["b116"], xf, w["w1"], u)
                c("mm_relu2_fwd", u.data_ptr(), u.data_ptr(), u.numel(), code, st)
            addmm(w["b216"], u, w["w2"], x0)
            calls.append((lambda: x0.add_(x1), None, None))
        ln(x0, dec.final_norm, xa)
        addmm(b_out16, xa, w_out, logits)
        self.argmax = len(calls)
        c("mm_argmax_rows", logits.data_ptr(), 0, B, V, logits.stride(0), code, st)
        self.calls = calls
        self.POS = 12                           # index of `pos` in mm_decode_attn's argument list
        self.id_bytes = ids.stride(0) * ids.element_size()
        self.ids_ptr = ids.data_ptr()

    def step(self, t: int, pos: int) -> None:
        """ids[t] (at sequence position pos - 1) -> ids[t + 1]."""
        torch.index_select(self.emb, 0, self.ids[t], out=self.x0)
        calls = self.calls
        for i in self.k4:
            calls[i][2][self.POS] = pos
        calls[self.argmax][2][1] = self.ids_ptr + (t + 1) * self.id_bytes
        for fn, name, args in calls:
            if args is None:
                fn()
            else:
                rc = fn(*args)
                if rc or L._TRACE:
                    L.check(rc, name)


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
    FN.flush_deferred_updates()        # this loop reads the weights directly, not through model.forward (optim: deferred updates)
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
    for layer in dec.layers:
        layer.rotary_emb.half_tables(Smax)         # cached tables exist before any other stream asks for them

    def prefill(pr, kcs, vcs):
        """Pass over the PROMPT rows `pr` [b, P, E] on the current stream (prompt positions see each other and nothing else, so
        they do not depend on what follows): fills the rows' cache slices (kcs / vcs: per-layer [b, H, Smax, d]).  The BOS
        token is not part of it -- it is decode step 0 at position P -- and of the last layer only the keys and values of the
        prompt are computed: nothing reads that layer's outputs at prompt positions (its attention, out-projection and MLP over
        B*P rows were a fifth of the prefill's products)."""
        nb = pr.shape[0]
        x = pr.contiguous()
        S0 = P
        M0 = nb * S0
        for li, layer in enumerate(dec.layers):
            w = weights[li]
            cos_t, sin_t = layer.rotary_emb.half_tables(Smax)
            xa, _, _ = ops.layernorm_fwd(x, layer.attn_norm.weight, layer.attn_norm.bias)
            rot = layer.rotary_ndims
            if li == len(dec.layers) - 1:
                kv = torch.addmm(w["bqkv16"][E:], xa.view(M0, E), w["wqkv"][E:].t()).view(nb, S0, 2, E)
                if dt == torch.bfloat16 and rot > 0:
                    ops.rope_inplace(kv.view(nb, S0, 2 * E)[:, :, :E], H, rot, cos_t, sin_t)
                    rot = 0
                ops.kv_cache_fill(kv[:, :, 0], kv[:, :, 1], kcs[li], vcs[li], H, rot, cos_t, sin_t)
                break
            # the big prefill products stay on the library where it wins (N = 3E, 4E: DESIGN.md 4), the N = E products with
            # their bias + residual run on the large-tile kernel, as in the training forward
            qkv = torch.addmm(w["bqkv16"], xa.view(M0, E), w["wqkv"].t()).view(nb, S0, 3, E)
            if dt == torch.bfloat16 and rot > 0:   # rotate q,k once in place: cache fill becomes a copy, K3 runs its fast form
                ops.rope_inplace(qkv.view(nb, S0, 3 * E)[:, :, :2 * E], 2 * H, rot, cos_t, sin_t)
                rot = 0
            ops.kv_cache_fill(qkv[:, :, 1], qkv[:, :, 2], kcs[li], vcs[li], H, rot, cos_t, sin_t)
            a, _ = ops.prefix_attn_fwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], H, rot, P, cos_t, sin_t, layer.use_stablemax)
            x = _lin(a.view(M0, E), w["wo"], w["bo"], res=x.view(M0, E), square=True).view(nb, S0, E)
            xf, _, _ = ops.layernorm_fwd(x, layer.mlp_norm.weight, layer.mlp_norm.bias)
            u = ops.relu2_fwd(torch.addmm(w["b116"], xf.view(M0, E), w["w1"].t()))     # fused on libmmhip it measured slower (prefill 27.5 vs 25 ms)
            x = _lin(u, w["w2"], w["b2"], res=x.view(M0, E), square=True).view(nb, S0, E)  # prefill: N = E on the large-tile kernel

    use_graph = bool(graph)
    nsplit = _DECODE_SPLIT if (_DECODE_SPLIT > 1 and not use_graph and B >= 128 * _DECODE_SPLIT and B % _DECODE_SPLIT == 0) else 1
    n_done = 0
    # the prefill runs once for the whole batch (MFMA-bound: cutting it in two only made its products smaller -- measured
    # 741k against 753k tokens/s with the halves prefilled on their own streams)
    prefill(prompt, cache.k, cache.v)
    finished = torch.zeros(B, dtype=torch.bool, device=dev)
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
            u = _lin(xf, w["w1"], w["b1"], w["b116"], act=ops.ACT_RELU2, square=True)      # libmmhip: bias + relu^2 fused
            h = _lin(u, w["w2"], w["b2"], w["b216"], res=h)
        xt, _, _ = ops.layernorm_fwd(h, dec.final_norm.weight, dec.final_norm.bias)
        return ops.argmax_rows(_lin(xt, w_out, b_out, b_out16, out_dtype=torch.float32))

    if use_graph and steps > 1:
        cur_s = torch.full((B,), bos_id, dtype=torch.long, device=dev)
        pos_dev = torch.tensor([P], dtype=torch.int32, device=dev)
        col_dev = torch.tensor([1], dtype=torch.long, device=dev)

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
            body()                                                             # t = 0 (the BOS token), eagerly: library warm-up before capture
        n_done = 1
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(warm):          # bare capture: torch.cuda.graph() would also synchronize, collect garbage
            g.capture_begin()                  # and empty the allocator cache on every generate() call
            try:
                body()
            finally:
                g.capture_end()
        cur_stream.wait_stream(warm)
        for t in range(1, steps):
            if early_stop and t % sync_every == 0 and bool(finished.all()):    # :113 (checked every few steps)
                break
            g.replay()
            n_done = t + 1
    else:
        # Rows are independent, so a large batch is cut into TWO halves that run their decode loops free of each other, one on
        # the current stream and one on the side stream (no event between them until the end).  A step alternates an
        # HBM-bound kernel (K4: the cache read, half of a step's time) with short latency-bound products (M = batch rows);
        # two loops out of phase put one half's cache read beside the other half's products (K4's LDS is sized per launch
        # for that: with its static 32 KiB five of its workgroups filled a CU's LDS and no product could move in).  Same
        # kernels on the same rows: the ids are those of the single loop (tests/test_decode_plan_gpu.py).
        emb_dt = FN.cached_operand((emb,), dt)
        main = torch.cuda.current_stream(dev)
        streams = [main] + ([FN.side_stream(dev)] if nsplit >= 2 else [])
        while len(streams) < nsplit:
            key = (dev, len(streams))
            if key not in _extra_streams:
                _extra_streams[key] = torch.cuda.Stream(device=dev)
            streams.append(_extra_streams[key])
        hb = B // nsplit
        parts = []
        for j, st in enumerate(streams):
            st.wait_stream(main)                                                 # the prefill's cache rows and first ids
            with torch.cuda.stream(st):
                view = KVCacheView([kc[j * hb: (j + 1) * hb] for kc in cache.k], [vc[j * hb: (j + 1) * hb] for vc in cache.v], Smax)
                ids = torch.empty((steps + 1, hb), dtype=torch.long, device=dev)   # row t+1: argmax output and next input
                ids[0] = bos_id
                parts.append((st, ids, _StepPlan(dec, weights, view, w_out, b_out, b_out16, emb_dt, ids, hb, dt, dev, Smax)))
        for t in range(steps):                                                   # step 0: the BOS token at position P
            if early_stop and t > 0 and t % sync_every == 0 and all(bool((ids[1: t + 1] == eos_id).any(0).all()) for _, ids, _ in parts):   # :113
                break
            for st, ids, plan in parts:
                with torch.cuda.stream(st):
                    plan.step(t, P + t)
            n_done = t + 1
        for j, (st, ids, plan) in enumerate(parts):
            with torch.cuda.stream(st):
                tokens[j * hb: (j + 1) * hb, 1: n_done + 1] = ids[1: n_done + 1].t()
            main.wait_stream(st)
            for tns in (ids, tokens):
                tns.record_stream(st)
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
