"""CrossAttentionBlock: API-compatible with reference models/cross_attention.py:6-46 (dead code there: it imports a
missing `.rope`).  Built on the same fused attention kernel (K3) with an empty causal part: every query row is a
"prompt" row of a [kv ; q] sequence is NOT needed -- cross attention is plain full attention of Tq queries over Tkv
keys, which K3 computes when q and kv have equal length; shorter query sets are padded, longer ones go through in chunks.
PARITY UNPINNED: the reference class cannot be imported, no reference output exists (oracle restates the text)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as FN
from .. import ops


class CrossAttentionBlock(nn.Module):
    def __init__(self, embed_dim, num_heads=4, dropout=0.1, use_rope=False, rope_freqs=None):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.dropout = nn.Dropout(dropout)
        self.linear_q = nn.Linear(embed_dim, embed_dim)
        self.linear_k = nn.Linear(embed_dim, embed_dim)
        self.linear_v = nn.Linear(embed_dim, embed_dim)
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        self.norm = nn.LayerNorm(embed_dim)
        self.use_rope, self.rope_freqs = use_rope, rope_freqs

    def forward(self, query, key_value):
        if self.use_rope and self.rope_freqs is not None:
            raise NotImplementedError("the reference's rope module (models/rope.py) does not exist; use_rope=False only")
        dt = FN.get_compute_dtype()
        B, Tq, C = query.shape
        Tkv = key_value.shape[1]
        q = F.linear(query.to(dt), self.linear_q.weight.to(dt), self.linear_q.bias.to(dt))
        k = F.linear(key_value.to(dt), self.linear_k.weight.to(dt), self.linear_k.bias.to(dt))
        v = F.linear(key_value.to(dt), self.linear_v.weight.to(dt), self.linear_v.bias.to(dt))
        if self.dropout.p != 0.0:
            # The class's own default (dropout=0.1).  The reference passes dropout_p=self.dropout.p to SDPA in EVERY mode
            # (:42), i.e. it drops attention probabilities in eval too, and applies the module dropout to the projected output
            # (:45, train mode only).  Dropping probabilities needs the [B,H,Tq,Tkv] matrix, which the fused kernel never
            # forms, so this (dead-code, parity-unpinned, random by construction) case forms it with plain tensor products and
            # torch's generator; the dropout-free case below stays on the fused kernel.
            H, d = self.num_heads, C // self.num_heads
            qh, kh, vh = (t.view(B, -1, H, d).transpose(1, 2) for t in (q, k, v))
            prob = torch.softmax((qh.float() @ kh.float().transpose(-1, -2)) * (d ** -0.5), dim=-1)
            prob = F.dropout(prob, self.dropout.p, training=True)
            a = (prob.to(dt) @ vh).transpose(1, 2).reshape(B, Tq, C)
            o = self.dropout(F.linear(a, self.out_proj.weight.to(dt), self.out_proj.bias.to(dt)))
            return FN.LayerNorm.apply(query.to(dt) + o, self.norm.weight, self.norm.bias)
        # K3 computes full attention of S rows over S keys when every row is a prompt row (prompt_len = S).  Keys must not be
        # padded (a padded key row would be seen), queries may: the queries go through in chunks of at most Tkv rows, each
        # padded to Tkv (the padded query rows are computed and dropped).
        S = Tkv
        zero = torch.zeros((S, 0), dtype=torch.float32, device=q.device)
        parts = []
        for t0 in range(0, Tq, S):
            n = min(S, Tq - t0)
            qkv = torch.zeros((B, S, 3, C), dtype=dt, device=q.device)
            qkv[:, :n, 0] = q[:, t0: t0 + n]
            qkv[:, :, 1] = k
            qkv[:, :, 2] = v
            parts.append(FN.PrefixAttention.apply(qkv, zero, zero, self.num_heads, 0, S, False)[:, :n])
        a = parts[0] if len(parts) == 1 else torch.cat(parts, 1)
        out = query.to(dt) + F.linear(a, self.out_proj.weight.to(dt), self.out_proj.bias.to(dt))
        return FN.LayerNorm.apply(out, self.norm.weight, self.norm.bias)
