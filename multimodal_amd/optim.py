"""OrthoGrad + AdamW as one fused multi-tensor HIP pass (reference ortho_grad.py:19-99 wrapped around
torch.optim.AdamW, as train_autoregressive.py:685-704 builds it).  State (exp_avg, exp_avg_sq, step) follows
torch.optim.AdamW's math; parameters stay ordinary nn.Parameters."""
from __future__ import annotations

import ctypes as C
from typing import Iterable

import numpy as np
import torch

from . import _lib as L
from . import functional as FN
from . import ops

CHUNK = 1 << 16


class ParamDesc(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_int64),
                ("w16", C.c_void_p)]


class OrthoGradAdamW(torch.optim.Optimizer):
    """capturable=True keeps the step-dependent scalars (learning rate, bias corrections) in device memory, so a HIP graph
    that captured step() stays valid across steps: advance() (called by step() itself outside capture, by
    graph.CapturedTrainStep before each replay) moves the step count and uploads the scalars."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                 ortho=True, ortho_eps=1e-30, rescale=True, capturable=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, ortho=ortho, ortho_eps=ortho_eps,
                        rescale=rescale)
        super().__init__(params, defaults)
        self._tables = {}
        self._step = 0
        self.capturable = bool(capturable)
        self._hyper = None            # [n_groups, 4] float32 on the device: lr, 1-b1^t, sqrt(1-b2^t), pad
        self._hyper_host = None

    def advance(self):
        """Open the next step: count it and (capturable) upload lr and the bias corrections of every group."""
        self._step += 1
        if not self.capturable:
            return
        R = 8       # ring of pinned staging rows: the host may run steps ahead of the copies it has queued
        if self._hyper is None:
            dev = self.param_groups[0]["params"][0].device
            self._hyper_host = torch.zeros(R, len(self.param_groups), 4, dtype=torch.float32).pin_memory()
            self._hyper_ev = [None] * R
            self._hyper = torch.zeros(len(self.param_groups), 4, dtype=torch.float32, device=dev)
        slot = self._step % R
        if self._hyper_ev[slot] is not None:
            self._hyper_ev[slot].synchronize()      # the copy queued R steps ago has read this row
        row = self._hyper_host[slot]
        for gi, g in enumerate(self.param_groups):
            b1, b2 = g["betas"]
            row[gi, 0] = float(g["lr"])
            row[gi, 1] = 1.0 - b1 ** self._step
            row[gi, 2] = (1.0 - b2 ** self._step) ** 0.5
        self._hyper.copy_(row, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._hyper_ev[slot] = ev

    @staticmethod
    def _key(plist, shadows):
        return tuple((p.data_ptr(), p.grad.data_ptr(), 0 if s is None else s.data_ptr()) for p, s in zip(plist, shadows))

    def _build(self, slot, plist, shadows):
        dev = plist[0].device
        descs = (ParamDesc * len(plist))()
        ct, co, tcs = [], [], [0]
        for t, p in enumerate(plist):
            st = self.state[p]
            if "exp_avg" not in st:
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            descs[t].p, descs[t].g = p.data_ptr(), p.grad.data_ptr()
            descs[t].m, descs[t].v = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            descs[t].n = p.numel()
            descs[t].w16 = None if shadows[t] is None else shadows[t].data_ptr()
            nch = (p.numel() + CHUNK - 1) // CHUNK
            ct += [t] * nch
            co += [i * CHUNK for i in range(nch)]
            tcs.append(tcs[-1] + nch)
        raw = np.frombuffer(bytes(descs), dtype=np.uint8)
        tab = dict(
            key=self._key(plist, shadows),
            descs=torch.from_numpy(raw.copy()).to(dev),
            ct=torch.tensor(ct, dtype=torch.int32, device=dev), co=torch.tensor(co, dtype=torch.int64, device=dev),
            tcs=torch.tensor(tcs, dtype=torch.int32, device=dev), n_tensors=len(plist), n_chunks=len(ct))
        self._tables[slot] = tab
        return tab

    def _run(self, slot, group, plist):
        """One OrthoGrad+AdamW launch sequence over plist with group's hyper-parameters, on the current stream."""
        for p in plist:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()):
                raise RuntimeError("OrthoGradAdamW: float32 contiguous GPU parameters and gradients only")
        # bf16 GEMM-operand copies the model keeps of these parameters (functional.cached_operand): the update
        # refreshes them in the same pass, so no cast kernels run in the next forward
        shadows = [FN.shadow_of(p) for p in plist]
        tab = self._tables.get(slot)
        if tab is None or tab["key"] != self._key(plist, shadows):
            tab = self._build(slot, plist, shadows)
        nbytes = L.lib().mm_orthograd_adamw_workspace(tab["n_tensors"], tab["n_chunks"])
        ws = ops.workspace(nbytes, plist[0].device, "optim")
        b1, b2 = group["betas"]
        if self.capturable:
            gi = next(i for i, g in enumerate(self.param_groups) if g is group)
            L.check(L.lib().mm_orthograd_adamw_dev(L.ptr(tab["descs"]), L.ptr(tab["ct"]), L.ptr(tab["co"]), L.ptr(tab["tcs"]),
                                                   tab["n_tensors"], tab["n_chunks"], CHUNK, self._hyper[gi].data_ptr(), b1, b2,
                                                   group["eps"], group["weight_decay"], group["ortho_eps"],
                                                   int(group["rescale"]), int(group["ortho"]), L.ptr(ws), nbytes,
                                                   L.cur_stream()), "mm_orthograd_adamw_dev")
            torch.autograd.graph.increment_version(plist)
            FN.shadows_written([p for p, s in zip(plist, shadows) if s is not None])
            return
        L.check(L.lib().mm_orthograd_adamw(L.ptr(tab["descs"]), L.ptr(tab["ct"]), L.ptr(tab["co"]), L.ptr(tab["tcs"]),
                                           tab["n_tensors"], tab["n_chunks"], CHUNK, group["lr"], b1, b2, group["eps"],
                                           group["weight_decay"], self._step, group["ortho_eps"], int(group["rescale"]),
                                           int(group["ortho"]), L.ptr(ws), nbytes, L.cur_stream()), "mm_orthograd_adamw")
        # the kernel writes through raw pointers: tell autograd (and functional.cached_operand) the values moved
        torch.autograd.graph.increment_version(plist)
        FN.shadows_written([p for p, s in zip(plist, shadows) if s is not None])

    # ---- optimizer-in-backward (driven by dp.DataParallel.attach_optimizer): OrthoGrad and AdamW are per-tensor, so a
    # bucket of parameters can be updated as soon as its gradients are final, while backward continues on other layers
    def begin_step(self):
        self.advance()
        self._begun = True
        self._done = set()
        self._group_of = {id(p): gi for gi, g in enumerate(self.param_groups) for p in g["params"]}

    @torch.no_grad()
    def step_subset(self, params, tag):
        """Update `params` now (their .grad must be final), on the current stream.  begin_step() opened the step."""
        if not getattr(self, "_begun", False):
            raise RuntimeError("OrthoGradAdamW.step_subset outside begin_step()/step()")
        by_group = {}
        for p in params:
            gi = self._group_of.get(id(p))
            if gi is not None and p.grad is not None and id(p) not in self._done:
                by_group.setdefault(gi, []).append(p)
        for gi, plist in by_group.items():
            self._run(("subset", tag, gi), self.param_groups[gi], plist)
            self._done.update(id(p) for p in plist)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        begun = getattr(self, "_begun", False)
        capturing = self.capturable and torch.cuda.is_current_stream_capturing()
        if not begun and not capturing:          # under capture the replaying caller advances (graph.CapturedTrainStep)
            self.advance()
        done = self._done if begun else ()
        FN.join_side_stream()
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None and id(p) not in done]
            if plist:
                self._run(("all" if not begun else "rest", gi), group, plist)
        self._begun = False
        self._done = set()
        return loss


def OrthoGrad(params, base_optimizer_cls=torch.optim.AdamW, eps: float = 1e-30, rescale: bool = True, **base_optimizer_args):
    """Constructor with the reference's signature (ortho_grad.py:20-46) for train scripts that build
    ``OrthoGrad(model.parameters(), base_optimizer_cls=optim.AdamW, eps=..., rescale=..., lr=..., ...)``
    (train_autoregressive.py:698-704).  The reference keeps ONE 'eps' entry per parameter group, shared by the
    projection and the wrapped AdamW (the base optimizer's own default never replaces it), and so does this.
    Only AdamW is fused here; any other base class is an error, not a fallback."""
    if base_optimizer_cls is not torch.optim.AdamW:
        raise NotImplementedError("multimodal_amd.optim.OrthoGrad: only base_optimizer_cls=torch.optim.AdamW is implemented")
    extra = set(base_optimizer_args) - {"lr", "betas", "weight_decay"}
    if extra:
        raise TypeError(f"multimodal_amd.optim.OrthoGrad: unsupported AdamW arguments {sorted(extra)}")
    return OrthoGradAdamW(params, lr=base_optimizer_args.get("lr", 1e-3), betas=base_optimizer_args.get("betas", (0.9, 0.999)),
                          eps=eps, weight_decay=base_optimizer_args.get("weight_decay", 1e-2), ortho=True, ortho_eps=eps,
                          rescale=rescale)
