"""state_dict key -> shape map of MultiModalToSMILESModel (oracle package: test
infrastructure only).  Mirrors the constructors at models/convnext.py:63-102,
models/transformer_decoder.py:93-109,:195-218 -- the compatibility contract of
SURVEY.md 8b(2): 297 keys at 8 layers."""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from .restate import Cfg


def state_shapes(cfg: Cfg) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    bb = "encoder.ir_backbone."
    d = list(cfg.dims)
    s[bb + "downsample_layers.0.0.weight"] = (d[0], 1, 4)
    s[bb + "downsample_layers.0.0.bias"] = (d[0],)
    s[bb + "downsample_layers.0.2.weight"] = (d[0],)
    s[bb + "downsample_layers.0.2.bias"] = (d[0],)
    for i in (1, 2, 3):
        s[bb + f"downsample_layers.{i}.1.weight"] = (d[i - 1],)
        s[bb + f"downsample_layers.{i}.1.bias"] = (d[i - 1],)
        s[bb + f"downsample_layers.{i}.3.weight"] = (d[i], d[i - 1], 2)
        s[bb + f"downsample_layers.{i}.3.bias"] = (d[i],)
    for st in range(4):
        c = d[st]
        for j in range(cfg.depths[st]):
            p = bb + f"stages.{st}.{j}."
            s[p + "gamma"] = (c,)
            s[p + "dwconv.weight"] = (c, 1, 7)
            s[p + "dwconv.bias"] = (c,)
            s[p + "norm.weight"] = (c,)
            s[p + "norm.bias"] = (c,)
            s[p + "pwconv1.weight"] = (4 * c, c)
            s[p + "pwconv1.bias"] = (4 * c,)
            s[p + "pwconv2.weight"] = (c, 4 * c)
            s[p + "pwconv2.bias"] = (c,)
    s[bb + "norm.weight"] = (d[3],)
    s[bb + "norm.bias"] = (d[3],)
    s[bb + "head.weight"] = (cfg.embed_dim, d[3])
    s[bb + "head.bias"] = (cfg.embed_dim,)
    e = cfg.embed_dim
    dec = "decoder."
    s[dec + "smiles_embed.weight"] = (cfg.smiles_vocab_size, e)
    s[dec + "nmr_embed.weight"] = (cfg.nmr_vocab_size, e)
    rot = int((e // cfg.num_heads) * 0.5)
    for i in range(cfg.num_layers):
        p = dec + f"layers.{i}."
        s[p + "rotary_emb.inv_freq"] = ((rot + 1) // 2,)
        for nm in ("q", "k", "v", "out"):
            s[p + nm + ".weight"] = (e, e)
            s[p + nm + ".bias"] = (e,)
        s[p + "ffn_w1.weight"] = (4 * e, e)
        s[p + "ffn_w1.bias"] = (4 * e,)
        s[p + "ffn_w2.weight"] = (e, 4 * e)
        s[p + "ffn_w2.bias"] = (e,)
        for nm in ("attn_norm", "mlp_norm"):
            s[p + nm + ".weight"] = (e,)
            s[p + nm + ".bias"] = (e,)
    s[dec + "final_norm.weight"] = (e,)
    s[dec + "final_norm.bias"] = (e,)
    s[dec + "out.weight"] = (cfg.smiles_vocab_size, e)
    s[dec + "out.bias"] = (cfg.smiles_vocab_size,)
    return s


def formula_state_dict(cfg: Cfg, seed: int = 0) -> Dict[str, torch.Tensor]:
    from . import fill
    sd = fill.make_state_dict(state_shapes(cfg), seed)
    rot = int((cfg.embed_dim // cfg.num_heads) * 0.5)
    inv = 1.0 / (10000 ** (torch.arange(0, rot, 2).float() / rot))     # transformer_decoder.py:9
    for i in range(cfg.num_layers):
        sd[f"decoder.layers.{i}.rotary_emb.inv_freq"] = inv.clone()
    return sd


def decoder_state_shapes(cfg: Cfg, memory_dim: int) -> Dict[str, Tuple[int, ...]]:
    """Keys of a stand-alone SMILESDecoder (no 'decoder.' prefix); memory_proj is a Linear when memory_dim differs from
    embed_dim (models/transformer_decoder.py:199)."""
    s = {k[len("decoder."):]: v for k, v in state_shapes(cfg).items() if k.startswith("decoder.")}
    if memory_dim != cfg.embed_dim:
        s["memory_proj.weight"] = (cfg.embed_dim, memory_dim)
        s["memory_proj.bias"] = (cfg.embed_dim,)
    return s


def decoder_formula_state_dict(cfg: Cfg, memory_dim: int, seed: int = 0) -> Dict[str, torch.Tensor]:
    from . import fill
    sd = fill.make_state_dict(decoder_state_shapes(cfg, memory_dim), seed)
    rot = int((cfg.embed_dim // cfg.num_heads) * 0.5)
    inv = 1.0 / (10000 ** (torch.arange(0, rot, 2).float() / rot))
    for i in range(cfg.num_layers):
        sd[f"layers.{i}.rotary_emb.inv_freq"] = inv.clone()
    return sd
