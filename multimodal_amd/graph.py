"""Whole train step as ONE HIP graph: forward + loss + backward + gradient completion + optimizer are captured once and
replayed per step, so the host issues one launch instead of ~1300 (the small configs are launch-bound --
nocat 11.9 -> 9.0 ms, test_config 14.5 -> 13.2 ms per step; the E1600 config is GPU-bound and gains nothing).

What makes the captured step valid across steps:
  * inputs live in static tensors the caller copies each batch into;
  * the optimizer is OrthoGradAdamW(capturable=True): learning rate and bias corrections are read from device memory,
    which advance() rewrites before every replay (the LR schedule keeps working);
  * DropPath draws come from torch's graph-safe Philox state (a fresh offset per replay);
  * the side stream is forked from and joined back into the capture stream by the step itself.
Single process only: a collective inside the captured region is not attempted here."""
from __future__ import annotations

from typing import Callable, Optional

import torch


class CapturedTrainStep:
    def __init__(self, step_fn: Callable[[], torch.Tensor], optimizer, warmup: int = 3,
                 before_warmup_step: Optional[Callable[[int], None]] = None):
        """step_fn(): one full eager step on STATIC input tensors (zero_grad, forward, loss, backward, dp.finish(),
        optimizer.step()), returning the loss tensor; it must not load the batch itself (a copy inside it would be captured
        and replayed).  `warmup` eager steps run first, on a side stream -- they are real steps;
        before_warmup_step(i) lets the caller load batch i / set the learning rate for each of them."""
        if not getattr(optimizer, "capturable", False):
            raise ValueError("CapturedTrainStep needs OrthoGradAdamW(capturable=True)")
        self.optimizer = optimizer
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for i in range(max(1, warmup)):
                if before_warmup_step is not None:
                    before_warmup_step(i)
                step_fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = step_fn()

    def __call__(self) -> torch.Tensor:
        self.optimizer.advance()
        self.graph.replay()
        return self.loss
