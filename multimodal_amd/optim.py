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
        self._step = 0                # steps opened so far (torch AdamW's per-parameter `step` = _step - _skipped[p])
        self._skipped = {}            # id(p) -> steps in which p had no gradient (torch does not count those for p)
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

    def _key(self, plist, shadows):
        """The static part of a descriptor table: parameter, moment and operand-copy addresses (load_state_dict replaces
        the moment tensors; .to()/.data= move parameters).  Gradient addresses are patched separately (_patch_grads)."""
        out = []
        for p, s in zip(plist, shadows):
            st = self.state.get(p, {})
            m, v = st.get("exp_avg"), st.get("exp_avg_sq")
            out.append((p.data_ptr(), 0 if m is None else m.data_ptr(), 0 if v is None else v.data_ptr(),
                        0 if s is None else s.data_ptr()))
        return tuple(out)

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
        host = np.frombuffer(bytes(descs), dtype=np.int64).reshape(len(plist), 6).copy()     # p, g, m, v, n, w16
        tab = dict(
            key=self._key(plist, shadows), host=host, gkey=tuple(p.grad.data_ptr() for p in plist),
            descs=torch.from_numpy(host.view(np.uint8).reshape(-1).copy()).to(dev),
            ct=torch.tensor(ct, dtype=torch.int32, device=dev), co=torch.tensor(co, dtype=torch.int64, device=dev),
            tcs=torch.tensor(tcs, dtype=torch.int32, device=dev), n_tensors=len(plist), n_chunks=len(ct))
        self._tables[slot] = tab
        return tab

    @staticmethod
    def _patch_grads(tab, plist):
        """Without gradient buckets autograd hands over fresh gradient tensors every step: only the g column of the
        descriptor table moves.  It is patched on the host copy and uploaded with ONE asynchronous copy from pinned
        memory (the caching host allocator keeps the staging block until the copy has run) -- no table rebuild, no
        synchronous pageable uploads."""
        gkey = tuple(p.grad.data_ptr() for p in plist)
        if gkey == tab["gkey"]:
            return
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("OrthoGradAdamW: gradient addresses moved under graph capture (use dp.DataParallel's "
                               "gradient buckets, whose slots are fixed)")
        tab["host"][:, 1] = gkey
        stage = torch.from_numpy(tab["host"].view(np.uint8).reshape(-1)).pin_memory()
        tab["descs"].copy_(stage, non_blocking=True)
        tab["gkey"] = gkey

    def _run(self, slot, group, plist, step=None):
        """One OrthoGrad+AdamW launch sequence over plist with group's hyper-parameters, on the current stream."""
        for p in plist:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()):
                raise RuntimeError("OrthoGradAdamW: float32 contiguous GPU parameters and gradients only")
        # bf16 GEMM-operand copies the model keeps of these parameters (functional.cached_operand): the update
        # refreshes them in the same pass, so no cast kernels run in the next forward
        shadows = [FN.shadow_of(p) for p in plist]
        for p in plist:                 # moments exist before the key is taken (their addresses are part of it)
            st = self.state[p]
            if "exp_avg" not in st:
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        tab = self._tables.get(slot)
        if tab is None or tab["key"] != self._key(plist, shadows):
            tab = self._build(slot, plist, shadows)
        else:
            self._patch_grads(tab, plist)
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
                                           group["weight_decay"], int(self._step if step is None else step),
                                           group["ortho_eps"], int(group["rescale"]),
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
            for j, (stp, sub) in enumerate(self._by_step(plist)):
                self._run(("subset", tag, gi, j), self.param_groups[gi], sub, stp)
            self._done.update(id(p) for p in plist)

    # ---- deferral into the next forward: the updates of the decoder layers (95 % of the parameters: 2 ms of HBM-bound work)
    # run on the side stream at the top of the NEXT forward, beside the encoder's ~150 small latency-bound kernels, each layer's
    # forward waiting only for its own update.  Same kernels, same inputs, same order per tensor: the parameters after step k
    # are bit-identical to the plain optimizer's; they just become so when the next forward (or flush_deferred) asks for them.
    _HYPER = ("lr", "betas", "eps", "weight_decay", "ortho", "ortho_eps", "rescale")

    def defer_into_next_forward(self, model, layer_modules):
        """Call once after construction.  `layer_modules`: the modules whose parameters may be updated late, in forward order
        (model.decoder.layers).  Everything else is updated inside step() as before.  Hooks on `model` launch / await the
        updates; state_dict() of the model or of this optimizer, zero_grad() of this optimizer and flush_deferred() complete
        them first.  Gradients must not be zeroed IN PLACE between step() and the next forward (DataParallel.zero_grad and
        zero_grad(set_to_none=True) only drop the references, which step() keeps).  Code that reads the deferred parameters
        through `parameters()` between step() and the next forward (an EMA, a weight-norm log) calls flush_deferred(wait=True)
        first: a raw tensor read cannot be hooked."""
        self._defer_layers = [[p for p in m.parameters()] for m in layer_modules]
        self._defer_ids = set(id(p) for ps in self._defer_layers for p in ps)
        self._pending = None
        self._defer_events = {}
        # every module that holds a deferred parameter -- the model, the decoder, the layers, their children -- completes the
        # pending updates before its state_dict() is read; every strict ancestor of the layers launches them at the top of
        # its forward (a loop may call model.decoder directly) and joins what is left when its forward returns
        layer_ids = set(id(m) for m in layer_modules)
        inside = set(id(sub) for m in layer_modules for sub in m.modules())
        for mod in model.modules():
            if not any(id(p) in self._defer_ids for p in mod.parameters()):
                continue
            mod.register_state_dict_pre_hook(lambda *a, **k: self.flush_deferred(wait=True))
            if id(mod) not in inside:
                mod.register_forward_pre_hook(lambda m, a: self.flush_deferred())
                mod.register_forward_hook(lambda m, a, o: self._await_deferred(None))
        for i, m in enumerate(layer_modules):
            m.register_forward_pre_hook(lambda mod, a, i=i: self._await_deferred(i))
        FN.register_flusher(self.flush_deferred)        # generate() reads the weights without a forward of `model`

    def _await_deferred(self, i):
        evs = getattr(self, "_defer_events", None)
        if not evs:
            return
        keys = [i] if i is not None else list(evs.keys())
        for k in keys:
            ev = evs.pop(k, None)
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)

    @torch.no_grad()
    def flush_deferred(self, wait=False):
        """Launch the pending deferred updates now (side stream; events per layer).  wait=True: the current stream also waits
        for all of them (before reading parameters outside a forward: checkpoints, evaluation through other entry points)."""
        pend = getattr(self, "_pending", None)
        if pend:
            self._pending = None
            dev = next(p for items in pend for p, _ in items).device
            if not dev.type == "cuda":
                raise RuntimeError("OrthoGradAdamW: deferred updates need GPU parameters")
            main, side = FN.enter_side(dev)            # after everything the main stream was given: gradients are final
            now = [{k: g[k] for k in self._HYPER} for g in self.param_groups]
            for g, h in zip(self.param_groups, self._pending_hyper):      # the hyper-parameters step() was called with
                g.update(h)
            with torch.cuda.stream(side):
                for i, items in enumerate(pend):
                    if not items:
                        continue
                    ps = [p for p, _ in items]
                    kept = [p.grad for p in ps]
                    for p, g in items:
                        p.grad = g
                    self.step_subset(ps, ("deferred", i))
                    for p, g in zip(ps, kept):
                        p.grad = g
                    ev = torch.cuda.Event()
                    ev.record(side)
                    self._defer_events[i] = ev
                FN.refresh_transposed_operands()
                ev = torch.cuda.Event()
                ev.record(side)
                self._defer_events["transposed"] = ev
            for g, h in zip(self.param_groups, now):
                g.update(h)
            self._begun = False
            self._done = set()
        if wait:
            self._await_deferred(None)

    def zero_grad(self, set_to_none: bool = True):
        self.flush_deferred(wait=True)                  # an in-place zero must not reach gradients still to be applied
        super().zero_grad(set_to_none=set_to_none)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if getattr(self, "_pending", None):
            # a step with no forward since the last one: the older update is still pending (and left the step open).  Apply
            # it first, in order, then open the new step
            self.flush_deferred()
        begun = getattr(self, "_begun", False)
        capturing = self.capturable and torch.cuda.is_current_stream_capturing()
        if getattr(self, "_defer_ids", None) and not begun and not capturing and not self.capturable:
            FN.join_side_stream()
            self.begin_step()
            early, late_missing = [], []
            for group in self.param_groups:
                for p in group["params"]:
                    if p.grad is None:
                        self._skipped[id(p)] = self._skipped.get(id(p), 0) + 1
                    elif id(p) not in self._defer_ids:
                        early.append(p)
            self.step_subset(early, "early")
            self._pending = [[(p, p.grad) for p in ps if p.grad is not None] for ps in self._defer_layers]
            self._pending_hyper = [{k: g[k] for k in self._HYPER} for g in self.param_groups]   # a scheduler may move lr before the flush
            if not any(self._pending):
                self._pending = None
                self._begun = False
                self._done = set()
            FN.refresh_transposed_operands()
            return loss
        if not begun and not capturing:          # under capture the replaying caller advances (graph.CapturedTrainStep)
            self.advance()
        done = self._done if begun else ()
        FN.join_side_stream()
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None and id(p) not in done]
            for j, (stp, sub) in enumerate(self._by_step(plist)):
                self._run(("all" if not begun else "rest", gi, j), group, sub, stp)
            for p in group["params"]:                # torch.optim.AdamW advances `step` only for parameters it updates
                if p.grad is None and id(p) not in done:
                    self._skipped[id(p)] = self._skipped.get(id(p), 0) + 1
        self._begun = False
        self._done = set()
        FN.refresh_transposed_operands()             # the transposed bf16 copies of what was just updated, in one launch
        return loss

    def _by_step(self, plist):
        """[(step, parameters)] -- one entry unless some parameter sat out earlier steps (it then lags, as in torch)."""
        groups = {}
        for p in plist:
            groups.setdefault(self._step - self._skipped.get(id(p), 0), []).append(p)
        if self.capturable and (len(groups) > 1 or any(k != self._step for k in groups)):
            raise RuntimeError("OrthoGradAdamW(capturable=True): every parameter must receive a gradient every step")
        return sorted(groups.items())

    # ---- checkpoint layout of the reference (ortho_grad.py:111-129: the wrapper's own state is empty, the wrapped AdamW's
    # state_dict sits under 'base_optimizer' with torch's per-parameter {'step', 'exp_avg', 'exp_avg_sq'})
    def state_dict(self):
        self.flush_deferred(wait=True)
        sd = super().state_dict()
        order = [p for g in self.param_groups for p in g["params"]]
        state = {}
        for idx, st in sd["state"].items():
            st = dict(st)
            st["step"] = torch.tensor(float(self._step - self._skipped.get(id(order[idx]), 0)), dtype=torch.float32)
            state[idx] = st
        groups = sd["param_groups"]
        return {"state": {}, "param_groups": groups,
                "base_optimizer": {"state": state, "param_groups": [dict(g) for g in groups]}}

    def load_state_dict(self, state_dict):
        """Accepts this class's state_dict() and the reference's OrthoGrad(AdamW) checkpoints (same layout; its groups
        carry one 'eps' for both the projection and AdamW and no 'ortho*' keys).  The caller's dict is not modified.
        Pending deferred updates are completed first (on the moments they were computed for), not on the loaded ones."""
        self.flush_deferred(wait=True)
        base = state_dict.get("base_optimizer")
        if base is not None:
            state, groups = base["state"], (state_dict.get("param_groups") or base["param_groups"])
        else:
            state, groups = state_dict["state"], state_dict["param_groups"]
        if len(groups) != len(self.param_groups):
            raise ValueError("loaded state dict has a different number of parameter groups")
        merged = []
        for cur, saved in zip(self.param_groups, groups):
            g = {k: v for k, v in cur.items() if k != "params"}
            for k in ("lr", "betas", "eps", "weight_decay", "rescale", "ortho", "ortho_eps"):
                if k in saved:
                    g[k] = tuple(saved[k]) if k == "betas" else saved[k]
            if "ortho_eps" not in saved and "eps" in saved:
                g["ortho_eps"] = saved["eps"]
            g["params"] = list(saved["params"])
            merged.append(g)
        steps, clean = {}, {}
        for idx, st in state.items():
            st = dict(st)
            stp = st.pop("step", None)
            steps[int(idx)] = 0 if stp is None else int(float(stp))
            clean[idx] = {k: v for k, v in st.items() if k in ("exp_avg", "exp_avg_sq")}
        super().load_state_dict({"state": clean, "param_groups": merged})
        order = [p for g in self.param_groups for p in g["params"]]
        flat_ids = [i for g in merged for i in g["params"]]
        self._step = max(steps.values(), default=0)
        self._skipped = {}
        for pos, idx in enumerate(flat_ids):
            if int(idx) in steps and steps[int(idx)] != self._step:
                self._skipped[id(order[pos])] = self._step - steps[int(idx)]
        # every cached device table points at the replaced moment tensors / an older step
        self._tables.clear()
        self._begun = False
        self._done = set()


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
