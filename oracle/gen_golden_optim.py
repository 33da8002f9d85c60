#!/usr/bin/env python3
"""Golden vectors for the optimizer of the train step (oracle package: test infrastructure only).

Runs ONLY in the build container: imports the reference's own ``ortho_grad.OrthoGrad`` (pure torch), wraps
``torch.optim.AdamW`` exactly as train_autoregressive.py:685-704 does, steps it on closed-form parameters and
gradients (``oracle/fill.hash_uniform``; nothing random, nothing stored but outputs) and writes
``tests/golden/optim.npz``: the parameters after every step.

    python oracle/gen_golden_optim.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("MM_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

from oracle import fill  # noqa: E402

# shapes: vector, matrix, scalar, conv weight, a tensor longer than one 65536-element chunk (ragged tail), one
# that never gets a gradient, and one whose gradient is identically zero
SHAPES = [(7,), (33, 17), (1,), (64, 1, 7), (300, 257), (5, 3), (40,)]
NO_GRAD = 5
ZERO_GRAD = 6
STEPS = 3
VARIANTS = {
    # name: (OrthoGrad eps, rescale, lr, betas, weight_decay).  The reference shares ONE 'eps' entry per param group
    # between OrthoGrad and the AdamW it wraps (ortho_grad.py:42-46: AdamW's own default never overrides it)
    "cfg": (1e-8, True, 1e-2, (0.9, 0.999), 0.01),          # configs/local_config.yaml optimizer section, larger lr
    "norescale": (1e-8, False, 3e-3, (0.8, 0.95), 0.1),
    "eps30": (1e-30, True, 1e-3, (0.9, 0.999), 0.0),
}


def params_and_grads():
    ps = [torch.from_numpy((0.3 * fill.hash_uniform(int(np.prod(s)), 500 + i)).astype(np.float32)).reshape(s)
          for i, s in enumerate(SHAPES)]
    gs = [[None if i == NO_GRAD else
           torch.zeros(s) if i == ZERO_GRAD else
           torch.from_numpy((fill.hash_uniform(int(np.prod(s)), 900 + 31 * t + i) * (0.5 + t)).astype(np.float32)).reshape(s)
           for i, s in enumerate(SHAPES)] for t in range(STEPS)]
    return ps, gs


def main():
    sys.path.insert(0, REF)
    from ortho_grad import OrthoGrad  # reference class
    store = {}
    for name, (eps, rescale, lr, betas, wd) in VARIANTS.items():
        ps, gs = params_and_grads()
        ps = [torch.nn.Parameter(p.clone()) for p in ps]
        opt = OrthoGrad(ps, base_optimizer_cls=torch.optim.AdamW, eps=eps, rescale=rescale, lr=lr, betas=betas,
                        weight_decay=wd)
        for t in range(STEPS):
            for p, g in zip(ps, gs[t]):
                p.grad = None if g is None else g.clone()
            opt.step()
            for i, p in enumerate(ps):
                store[f"{name}.step{t}.p{i}"] = p.detach().numpy().copy()
            if t == 0:
                for i, p in enumerate(ps):
                    if p.grad is not None:
                        store[f"{name}.step0.g{i}"] = p.grad.numpy().copy()      # the projected gradient left in .grad
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "optim.npz"), **store)
    print("wrote", len(store), "arrays")


if __name__ == "__main__":
    main()
