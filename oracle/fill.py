"""Closed-form, RNG-free weights and inputs shared by the golden generator, the
oracle tests and the GPU parity tests (oracle package: test infrastructure only).

The same filler is applied to the reference model (in ``gen_golden.py``, build
container only) and to the oracle / product models (everywhere), so no weights
are stored in the fixtures.  Values come from a splitmix64 hash of
(tensor ordinal, element index): platform independent, full rank, no torch RNG.
"""
from __future__ import annotations

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def hash_uniform(n: int, stream: int) -> np.ndarray:
    """n float64 values in [-1, 1), a pure function of (stream, index)."""
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = (np.uint64(stream) * np.uint64(0xD1342543DE82EF95)) & _M64
    z = _splitmix64(idx + base)
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 / (1 << 53)) - 1.0


def _scale_for(key: str, shape) -> tuple[float, float]:
    """(offset, amplitude) per parameter kind; chosen so activations stay O(1)
    and the LayerScale branch (gamma) is observable (reference init is 1e-6)."""
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "gamma":
        return 0.6, 0.3
    if "norm" in key or key.endswith(("downsample_layers.0.2.weight", "downsample_layers.0.2.bias")) \
            or any(key.endswith(f"downsample_layers.{i}.1.{leaf}") for i in (1, 2, 3)):
        return (1.0, 0.2) if leaf == "weight" else (0.0, 0.1)
    if "embed" in key:
        return 0.0, 0.7
    if leaf == "bias":
        return 0.0, 0.1
    if leaf == "weight":
        fan_in = 1
        for s in shape[1:]:
            fan_in *= int(s)
        return 0.0, 1.7 / max(1.0, fan_in) ** 0.5
    return 0.0, 0.5


def fill_state_dict_(module: torch.nn.Module, seed: int = 0) -> None:
    """In-place formula fill of every floating-point parameter (buffers such as
    ``rotary_emb.inv_freq`` keep their constructor values)."""
    sd = module.state_dict()
    with torch.no_grad():
        for ordinal, key in enumerate(sorted(sd.keys())):
            t = sd[key]
            if key.endswith("inv_freq") or not t.is_floating_point():
                continue
            off, amp = _scale_for(key, tuple(t.shape))
            vals = off + amp * hash_uniform(t.numel(), seed * 100003 + ordinal + 1)
            t.copy_(torch.from_numpy(vals.astype(np.float32)).view_as(t))


def make_state_dict(shapes: dict, seed: int = 0) -> dict:
    """Formula-filled tensors for a {key: shape} map (same values as
    ``fill_state_dict_`` on a module with exactly these keys)."""
    out = {}
    for ordinal, key in enumerate(sorted(shapes.keys())):
        shp = tuple(shapes[key])
        if key.endswith("inv_freq"):
            continue
        n = int(np.prod(shp)) if len(shp) else 1
        off, amp = _scale_for(key, shp)
        vals = off + amp * hash_uniform(n, seed * 100003 + ordinal + 1)
        out[key] = torch.from_numpy(vals.astype(np.float32)).reshape(shp)
    return out


def synth_ir(batch: int, length: int = 1000, seed: int = 0) -> torch.Tensor:
    """IR spectra in [0, 1) (reference data is min-max normalised:
    create_tokenized_dataset_faster.py:96-104)."""
    v = (hash_uniform(batch * length, 7_000_000 + seed) + 1.0) * 0.5
    # smooth a little so that the signal looks like a spectrum, not white noise
    v = v.reshape(batch, length)
    k = np.array([0.25, 0.5, 0.25])
    v = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, v)
    return torch.from_numpy(v.astype(np.float32))


def synth_tokens(batch: int, length: int, lo: int, hi: int, seed: int = 0) -> torch.Tensor:
    v = (hash_uniform(batch * length, 9_000_000 + seed) + 1.0) * 0.5
    t = lo + np.floor(v * (hi - lo)).astype(np.int64)
    return torch.from_numpy(np.clip(t, lo, hi - 1)).reshape(batch, length)
