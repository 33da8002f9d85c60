"""CPU (PyTorch fp32) restatement of the reference hot path, written as pure
functions over a flat ``state_dict`` (oracle package: TEST INFRASTRUCTURE ONLY;
the product under ``multimodal_amd/`` never imports it).

Parity: PINNED by ``tests/golden/*.npz`` (generated from the reference itself
by ``oracle/gen_golden.py``); see ``tests/test_oracle_golden.py``.

Every function cites the reference lines it follows (paths relative to the
reference repository root).  The arithmetic is ATen's (torch CPU kernels), which
is also what the reference runs on ("third-party arithmetic").
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

BOS_ID, EOS_ID, PAD_ID, UNK_ID = 12, 13, 0, 11   # [CLS], [SEP], [PAD], [UNK] rows of vocab.txt


@dataclass
class Cfg:
    embed_dim: int = 768
    num_heads: int = 8
    num_layers: int = 6
    max_seq_length: int = 512
    max_nmr_length: int = 128
    max_memory_length: int = 128
    use_stablemax: bool = False
    smiles_vocab_size: int = 591
    nmr_vocab_size: int = 1000
    depths: Sequence[int] = (3, 3, 6, 3)          # models/spectral_encoder.py:21
    drop_path_rate: float = 0.1                    # models/spectral_encoder.py:23
    dims: Sequence[int] = field(default=None)      # models/spectral_encoder.py:22

    def __post_init__(self):
        if self.dims is None:
            self.dims = (64, 128, 256, self.embed_dim)


# --------------------------------------------------------------------------
# encoder: models/convnext.py, models/spectral_encoder.py
# --------------------------------------------------------------------------

def drop_path_rates(cfg: Cfg) -> List[float]:
    """models/convnext.py:85 -- linspace(0, rate, sum(depths)); block 0 -> Identity (:34)."""
    return [x.item() for x in torch.linspace(0, cfg.drop_path_rate, sum(cfg.depths))]


def drop_path_scale(rate: float, batch: int, dtype=torch.float32) -> Tensor:
    """models/convnext.py:13-17 -- draws the same torch.rand the reference draws and
    returns the per-sample factor floor(keep + U) / keep, shape [B,1,1]."""
    keep = 1 - rate
    r = keep + torch.rand((batch, 1, 1), dtype=dtype)
    r.floor_()
    return r / keep


def convnext_block(sd: SD, pre: str, x: Tensor, dp_scale: Optional[Tensor] = None) -> Tensor:
    """models/convnext.py:36-48.  x: [B,C,L] -> [B,C,L].  dp_scale: [B,1,1] or None."""
    c = x.shape[1]
    y = F.conv1d(x, sd[pre + "dwconv.weight"], sd[pre + "dwconv.bias"], padding=3, groups=c)  # :38
    y = y.transpose(1, 2)                                                                    # :39
    y = F.layer_norm(y, (c,), sd[pre + "norm.weight"], sd[pre + "norm.bias"], 1e-5)          # :40
    y = F.linear(y, sd[pre + "pwconv1.weight"], sd[pre + "pwconv1.bias"])                    # :41
    y = F.gelu(y)                                                                            # :42 (erf)
    y = F.linear(y, sd[pre + "pwconv2.weight"], sd[pre + "pwconv2.bias"])                    # :43
    y = sd[pre + "gamma"] * y                                                                # :45
    y = y.transpose(1, 2)                                                                    # :46
    if dp_scale is not None:
        # x.div(keep) * mask  (:17); dp_scale already is mask/keep, applied in that order
        y = y * dp_scale
    return x + y                                                                             # :47


def convnext1d(sd: SD, pre: str, x: Tensor, cfg: Cfg, training: bool = False,
               taps: Optional[dict] = None) -> Tensor:
    """models/convnext.py:120-141 with keep_sequence=True, regression=True.
    x: [B,1,Lin] -> [B, L/32.., E].  ``taps`` collects intermediate activations."""
    b = x.shape[0]
    rates = drop_path_rates(cfg)
    # stem :67-72
    y = F.conv1d(x, sd[pre + "downsample_layers.0.0.weight"], sd[pre + "downsample_layers.0.0.bias"], stride=4)
    y = F.layer_norm(y.permute(0, 2, 1), (cfg.dims[0],), sd[pre + "downsample_layers.0.2.weight"],
                     sd[pre + "downsample_layers.0.2.bias"], 1e-5).permute(0, 2, 1)
    if taps is not None:
        taps["stem"] = y
    blk = 0
    for s in range(4):
        if s > 0:   # :75-82
            y = F.layer_norm(y.permute(0, 2, 1), (cfg.dims[s - 1],), sd[pre + f"downsample_layers.{s}.1.weight"],
                             sd[pre + f"downsample_layers.{s}.1.bias"], 1e-5).permute(0, 2, 1)
            y = F.conv1d(y, sd[pre + f"downsample_layers.{s}.3.weight"], sd[pre + f"downsample_layers.{s}.3.bias"],
                         stride=2)
            if taps is not None:
                taps[f"down{s}"] = y
        for j in range(cfg.depths[s]):
            scale = None
            if training and rates[blk] > 0.0:          # :11, :34
                scale = drop_path_scale(rates[blk], b, y.dtype)
            y = convnext_block(sd, pre + f"stages.{s}.{j}.", y, scale)
            if taps is not None:
                taps[f"stage{s}.{j}"] = y
            blk += 1
    y = y.transpose(1, 2)                                                                    # :130
    y = F.layer_norm(y, (cfg.dims[-1],), sd[pre + "norm.weight"], sd[pre + "norm.bias"], 1e-5)   # :131
    return F.linear(y, sd[pre + "head.weight"], sd[pre + "head.bias"])                       # :134


def spectral_encoder(sd: SD, ir: Optional[Tensor], cfg: Cfg, training: bool = False,
                     taps: Optional[dict] = None, pre: str = "encoder.") -> Optional[Tensor]:
    """models/spectral_encoder.py:31-50 -- only ``ir_data`` is used."""
    if ir is None:
        return None
    if ir.dim() == 2:
        ir = ir.unsqueeze(1)                                                                 # :42-43
    return convnext1d(sd, pre + "ir_backbone.", ir, cfg, training, taps)


# --------------------------------------------------------------------------
# decoder: models/transformer_decoder.py
# --------------------------------------------------------------------------

def rotary_tables(rot_dim: int, seq_len: int, inv_freq: Optional[Tensor] = None):
    """models/transformer_decoder.py:9, :18-22.  Returns cos, sin of shape [S, rot_dim]."""
    if inv_freq is None:
        inv_freq = 1.0 / (10000 ** (torch.arange(0, rot_dim, 2).float() / rot_dim))
    t = torch.arange(seq_len)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def _half_turn(x: Tensor) -> Tensor:
    """models/transformer_decoder.py:26-28."""
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), -1)


def apply_partial_rope(q: Tensor, k: Tensor, rot: int, inv_freq: Optional[Tensor] = None):
    """models/transformer_decoder.py:129-135 + :33-35.  q,k: [B,H,S,d]; first ``rot`` dims rotate."""
    s = q.shape[-2]
    cos, sin = rotary_tables(rot, s, inv_freq)
    qr, qp = q[..., :rot], q[..., rot:]
    kr, kp = k[..., :rot], k[..., rot:]
    qr = qr * cos + _half_turn(qr) * sin
    kr = kr * cos + _half_turn(kr) * sin
    return torch.cat((qr, qp), -1), torch.cat((kr, kp), -1)


def stable_s(x: Tensor, epsilon: float = 1e-9) -> Tensor:
    """models/transformer_decoder.py:53-57 (same fp32 operation order)."""
    return torch.where(x < 0, 1.0 / (1.0 - x + epsilon), x + 1.0)


def stablemax(x: Tensor, dim: int = -1, epsilon: float = 1e-9) -> Tensor:
    """models/transformer_decoder.py:66-75."""
    s = stable_s(x, epsilon)
    return s / s.sum(dim=dim, keepdim=True).clamp_min(epsilon)


def prefix_lm_mask(batch: int, prompt_len: int, tgt_len: int) -> Tensor:
    """models/transformer_decoder.py:295-297 (before the per-head repeat). bool [B,S,S], True = keep."""
    s = prompt_len + tgt_len
    m = torch.ones(batch, s, s).tril().bool()
    m[:, :prompt_len, :prompt_len] = True
    return m


def attention_core(q: Tensor, k: Tensor, v: Tensor, mask: Tensor, use_stablemax: bool) -> Tensor:
    """models/transformer_decoder.py:138-153.  q,k,v: [B,H,S,d]; mask bool [B,H|1,S,S]."""
    if use_stablemax:
        scores = (q @ k.transpose(-2, -1)) / math.sqrt(q.shape[-1])                          # :140
        scores = scores.masked_fill(~mask, float("-inf"))                                    # :144
        return stablemax(scores, dim=-1) @ v                                                 # :148-150
    return F.scaled_dot_product_attention(q, k, v, attn_mask=mask, is_causal=False)          # :153


def decoder_layer(sd: SD, pre: str, x: Tensor, mask: Tensor, nhead: int, use_stablemax: bool,
                  taps: Optional[dict] = None) -> Tensor:
    """models/transformer_decoder.py:116-165 (all three dropouts are p=0.0, :111-113)."""
    b, s, e = x.shape
    d = e // nhead
    rot = int(d * 0.5)                                                                       # :94
    xa = F.layer_norm(x, (e,), sd[pre + "attn_norm.weight"], sd[pre + "attn_norm.bias"], 1e-5)   # :122
    q = F.linear(xa, sd[pre + "q.weight"], sd[pre + "q.bias"]).view(b, s, nhead, d).transpose(1, 2)
    k = F.linear(xa, sd[pre + "k.weight"], sd[pre + "k.bias"]).view(b, s, nhead, d).transpose(1, 2)
    v = F.linear(xa, sd[pre + "v.weight"], sd[pre + "v.bias"]).view(b, s, nhead, d).transpose(1, 2)
    q, k = apply_partial_rope(q, k, rot, sd.get(pre + "rotary_emb.inv_freq"))
    a = attention_core(q, k, v, mask, use_stablemax)
    if taps is not None:
        taps[pre + "attn"] = a
    a = a.transpose(1, 2).contiguous().view(b, s, e)                                         # :155
    x = F.linear(a, sd[pre + "out.weight"], sd[pre + "out.bias"]) + x                        # :156-157
    xf = F.layer_norm(x, (e,), sd[pre + "mlp_norm.weight"], sd[pre + "mlp_norm.bias"], 1e-5)  # :160
    xf = F.relu(F.linear(xf, sd[pre + "ffn_w1.weight"], sd[pre + "ffn_w1.bias"])).square()   # :161
    xf = F.linear(xf, sd[pre + "ffn_w2.weight"], sd[pre + "ffn_w2.bias"])                    # :162
    return x + xf                                                                            # :163


def smiles_decoder(sd: SD, tgt: Tensor, memory: Tensor, nmr_tokens: Optional[Tensor], cfg: Cfg,
                   taps: Optional[dict] = None, pre: str = "decoder.") -> Tensor:
    """models/transformer_decoder.py:220-311.  Returns logits [B,T,V]."""
    b, t = tgt.shape
    if memory.size(1) > cfg.max_memory_length:                                               # :235-238
        memory = memory[:, : cfg.max_memory_length]
    if nmr_tokens is not None and nmr_tokens.size(1) > cfg.max_nmr_length:                   # :240-243
        nmr_tokens = nmr_tokens[:, : cfg.max_nmr_length]
    plen = memory.size(1) + (nmr_tokens.size(1) if nmr_tokens is not None else 0)            # :246
    if t + plen > cfg.max_seq_length + 256:                                                  # :249-253
        raise ValueError(
            f"Combined sequence length ({t} + {plen} = {t + plen}) "
            f"exceeds maximum length ({cfg.max_seq_length + 256})")
    x = F.embedding(tgt, sd[pre + "smiles_embed.weight"])                                    # :256
    if (pre + "memory_proj.weight") in sd:                                                   # :199, :259
        memory = F.linear(memory, sd[pre + "memory_proj.weight"], sd[pre + "memory_proj.bias"])
    prompt = memory
    if nmr_tokens is not None:
        prompt = torch.cat([memory, F.embedding(nmr_tokens, sd[pre + "nmr_embed.weight"])], dim=1)  # :272-282
    m = prompt.shape[1]
    x = torch.cat([prompt, x], dim=1)                                                        # :292
    mask = prefix_lm_mask(b, m, t)[:, None]                                                  # :295-298 (broadcast == repeat)
    for i in range(cfg.num_layers):
        x = decoder_layer(sd, pre + f"layers.{i}.", x, mask, cfg.num_heads, cfg.use_stablemax, taps)
        if taps is not None:
            taps[f"layer{i}"] = x
    xt = F.layer_norm(x[:, m:], (cfg.embed_dim,), sd[pre + "final_norm.weight"],
                      sd[pre + "final_norm.bias"], 1e-5)                                     # :305-306
    return F.linear(xt, sd[pre + "out.weight"], sd[pre + "out.bias"])                        # :307


def model_forward(sd: SD, cfg: Cfg, nmr_tokens: Optional[Tensor], ir: Tensor, target_seq: Tensor,
                  training: bool = False, taps: Optional[dict] = None) -> Tensor:
    """models/multimodal_to_smiles.py:101-108 (``target_mask`` is ignored there, :74)."""
    memory = spectral_encoder(sd, ir, cfg, training, taps)
    if taps is not None:
        taps["memory"] = memory
    return smiles_decoder(sd, target_seq, memory, nmr_tokens, cfg, taps)


def train_loss(sd: SD, cfg: Cfg, nmr_tokens: Tensor, ir: Tensor, target_tokens: Tensor,
               training: bool = True) -> Tensor:
    """train_autoregressive.py:968-974 with criterion :672-674 (ignore_index=[PAD]=0)."""
    logits = model_forward(sd, cfg, nmr_tokens, ir, target_tokens[:, :-1], training)
    return F.cross_entropy(logits.reshape(-1, logits.size(-1)), target_tokens[:, 1:].reshape(-1),
                           ignore_index=PAD_ID)


# --------------------------------------------------------------------------
# greedy decode: train_autoregressive.py:51-132
# --------------------------------------------------------------------------

@torch.no_grad()
def greedy_decode_ids(sd: SD, cfg: Cfg, nmr_tokens: Optional[Tensor], ir: Tensor, max_len: int = 128,
                      early_stop: bool = True) -> List[List[int]]:
    """The reference loop: encoder once (:83), then FULL decoder recompute per step (:97-101),
    argmax of the last position (:102), per-row EOS bookkeeping (:105-110), stop when all rows
    are finished (:113).  Returns, per row, the ids appended while the row was unfinished,
    starting with BOS (the ``generated_sequences`` of :86-88)."""
    bsz = nmr_tokens.size(0) if nmr_tokens is not None else ir.size(0)                       # :73-77
    cur = torch.full((bsz, 1), BOS_ID, dtype=torch.long)                                     # :80
    memory = spectral_encoder(sd, ir, cfg, training=False)                                   # :83
    seqs = [[BOS_ID] for _ in range(bsz)]
    done = [False] * bsz
    for _ in range(min(max_len, cfg.max_seq_length)):                                        # :91,:95
        logits = smiles_decoder(sd, cur, memory, nmr_tokens, cfg)
        nxt = logits[:, -1:].argmax(dim=-1)                                                  # :102
        for i in range(bsz):
            if not done[i]:
                tok = int(nxt[i].item())
                seqs[i].append(tok)
                if tok == EOS_ID:
                    done[i] = True
        if early_stop and all(done):                                                         # :113
            break
        cur = torch.cat([cur, nxt], dim=1)                                                   # :116
    return seqs


def ids_to_string(seq: List[int], id_to_token: Dict[int, str]) -> str:
    """train_autoregressive.py:120-130 + models/smiles_tokenizer.py:77-82: cut at the first EOS,
    drop BOS, join tokens with single spaces."""
    if EOS_ID in seq:
        seq = seq[: seq.index(EOS_ID)]
    toks = [id_to_token.get(i, "[UNK]") for i in seq[1:]]
    return " ".join(toks).replace(" ##", "").strip()


# --------------------------------------------------------------------------
# tokenizer / indexing: models/smiles_tokenizer.py
# --------------------------------------------------------------------------

# models/smiles_tokenizer.py:10 (a regular expression is data: it IS the tokenisation rule)
SMILES_PATTERN = r"""(\[[^\]]+]|Br?|Cl?|N|O|S|P|F|I|b|c|n|o|s|p|\(|\)|\.|=|#|-|\+|\\|\/|:|~|@|\?|>>?|\*|\$|\%[0-9]{2}|[0-9])"""
_smiles_re = re.compile(SMILES_PATTERN)


def read_vocab(path: str) -> Dict[str, int]:
    """models/smiles_tokenizer.py:13-21: one token per line, index = line number."""
    with open(path, "r", encoding="utf-8") as fh:
        return {line.rstrip("\n"): i for i, line in enumerate(fh.readlines())}


def encode_smiles(text: str, vocab: Dict[str, int], max_seq_length: int = 512) -> List[int]:
    """models/smiles_tokenizer.py:69-75, :84-85: regex split, truncate to max-2, [CLS] ids [SEP]."""
    toks = _smiles_re.findall(text)[: max_seq_length - 2]
    return [BOS_ID] + [vocab.get(t, UNK_ID) for t in toks] + [EOS_ID]


# --------------------------------------------------------------------------
# CrossAttentionBlock: models/cross_attention.py:20-46 -- dead code in the reference (cannot be
# imported: models/rope.py is missing).  Restated from the source text, use_rope=False branch.
# PARITY UNPINNED for this function: no output of the reference exists for it.
# --------------------------------------------------------------------------

def cross_attention_block(sd: SD, pre: str, query: Tensor, key_value: Tensor, num_heads: int) -> Tensor:
    b, tq, c = query.shape
    tkv = key_value.shape[1]
    d = c // num_heads
    q = F.linear(query, sd[pre + "linear_q.weight"], sd[pre + "linear_q.bias"]).view(b, tq, num_heads, d).permute(0, 2, 1, 3)
    k = F.linear(key_value, sd[pre + "linear_k.weight"], sd[pre + "linear_k.bias"]).view(b, tkv, num_heads, d).permute(0, 2, 1, 3)
    v = F.linear(key_value, sd[pre + "linear_v.weight"], sd[pre + "linear_v.bias"]).view(b, tkv, num_heads, d).permute(0, 2, 1, 3)
    a = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0)          # :42 with dropout=0.0
    a = a.permute(0, 2, 1, 3).contiguous().view(b, tq, c)
    out = query + F.linear(a, sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"])        # :44-45
    return F.layer_norm(out, (c,), sd[pre + "norm.weight"], sd[pre + "norm.bias"], 1e-5)     # :46


# --------------------------------------------------------------------------
# OrthoGrad + AdamW: ortho_grad.py:48-99 (next-row N3; used by the DP equivalence tests)
# --------------------------------------------------------------------------

@torch.no_grad()
def orthograd_project_(params: Sequence[Tensor], eps: float = 1e-30, rescale: bool = True) -> None:
    """ortho_grad.py:60-80, in place on ``p.grad``."""
    for p in params:
        if p.grad is None:
            continue
        w = p.view(-1)
        g = p.grad.view(-1)
        proj = torch.dot(w, g) / (torch.dot(w, w) + eps)
        g_orth = g - proj * w
        if rescale:
            g_orth = g_orth * (g.norm(2) / (g_orth.norm(2) + eps))
        p.grad.copy_(g_orth.view_as(p.grad))
