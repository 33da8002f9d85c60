"""Data-parallel training: one process per GPU, replicated weights, ONE exchange step -- the gradient all-reduce
on RCCL over xGMI (torch.distributed backend "nccl" is RCCL on ROCm).  The reference is single-device
(train_autoregressive.py:608); equivalence target = a single-process step on the concatenated global batch
(SURVEY.md 8e).

Gradients live permanently inside a few large flat buckets (p.grad are views), so a bucket is reduced in place
with no packing copies; a bucket's all-reduce is launched from an autograd hook as soon as its last gradient has
been accumulated, on RCCL's own stream, overlapping the rest of backward.  Buckets are filled in reverse
parameter order (the order backward produces gradients).  Bucket size defaults to 128 MiB: xGMI is
point-to-point (7 links x ~153 GB/s), large messages keep every link busy and 288 GB of HBM makes them free."""
from __future__ import annotations

import os
from contextlib import contextmanager
from typing import List, Optional

import torch
import torch.distributed as dist

from . import functional as FN


class GradBuckets:
    """p.grad become views into a few flat float buffers.  Every slot starts on a 256-byte boundary (GEMMs write
    weight gradients straight into them).  `adjacent`: tuples of parameters whose slots must follow one another in
    the given order, so that one GEMM can produce their stacked gradient (the fused q/k/v projection)."""

    ALIGN = 64          # elements
    # zero() does not touch the buffers: it unsets every .grad.  Autograd then keeps the gradient tensor it is handed
    # (no accumulate launch), nodes that produce a whole weight gradient in one GEMM write the slot directly
    # (functional._weight_grad), and gather() completes a bucket before it is reduced / the optimizer runs: one
    # multi-tensor copy for the gradients left outside, one multi-tensor zero for the slots nobody wrote.

    def __init__(self, params: List[torch.nn.Parameter], bucket_bytes: int = 128 << 20, comm_dtype=None, adjacent=()):
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []           # dicts: flat, params, pending, work
        group_of = {}
        for g in adjacent:
            g = tuple(p for p in g if p.requires_grad)
            for p in g:
                group_of[p] = g
        cur, cur_n, seen = [], 0, set()
        for p in reversed(self.params):
            if p in seen:
                continue
            for q in group_of.get(p, (p,)):
                seen.add(q)
                cur.append(q)
                cur_n += -(-q.numel() // self.ALIGN) * self.ALIGN
            if cur_n * 4 >= bucket_bytes:
                self._make(cur)
                cur, cur_n = [], 0
        if cur:
            self._make(cur)
        self.comm_dtype = comm_dtype

    def _make(self, plist):
        n = sum(-(-p.numel() // self.ALIGN) * self.ALIGN for p in plist)
        flat = torch.zeros(n, dtype=plist[0].dtype, device=plist[0].device)
        off = 0
        for p in plist:
            p._mm_view = flat[off: off + p.numel()].view_as(p)
            p.grad = p._mm_view
            p._mm_slot = (flat, off)
            off += -(-p.numel() // self.ALIGN) * self.ALIGN
        self.buckets.append(dict(flat=flat, params=list(plist), pending=len(plist), work=None, fired=set()))

    def zero(self):
        for b in self.buckets:
            for p in b["params"]:
                p._mm_fresh = True
                p.grad = None

    def gather(self, b):
        """Complete a bucket (or a list of buckets, in ONE multi-tensor copy): copy in the gradients autograd left outside
        it, zero the slots that received nothing, and point every .grad at its slot."""
        src, dst, zero = [], [], []
        for p in (b["params"] if isinstance(b, dict) else [p for bb in b for p in bb["params"]]):
            g = p.grad
            if g is None:
                zero.append(p._mm_view)
            elif g is not p._mm_view:
                dst.append(p._mm_view)
                src.append(g.detach().reshape(p.shape).to(p._mm_view.dtype))
            p.grad = p._mm_view
        if src:
            if src[0].is_cuda:             # gather may run on the side stream: keep the sources' memory until it has run
                cur = torch.cuda.current_stream(src[0].device)
                for t in src:
                    t.record_stream(cur)
            torch._foreach_copy_(dst, src)
        if zero:
            torch._foreach_zero_(zero)


_OWN_COMM_STREAM = os.environ.get("MM_DP_COMM_STREAM", "1") == "1"     # 0: async collectives on the process group's stream


class _EventWork:
    """Work handle of a collective issued on our own communication stream: wait() orders the current stream after it."""

    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)
        return True


class _Done:
    """Stand-in for a collective's work handle when the collective has already completed."""

    def wait(self):
        return True


class DataParallel:
    """Wraps a replicated module.  Usage per step:  dp.zero_grad(); dp.scale_loss(loss).backward(); dp.finish();
    optimizer.step()  (plain loss.backward() also works: finish() then divides the reduced gradients by the world size).
    Gradients are complete (all in their bucket slots, averaged over ranks) only after finish().
    host_staged: run the collectives on host copies of GPU tensors -- for backends without (reliable) device support;
    defaults to True when the process group's backend is gloo, False for nccl (= RCCL)."""

    def __init__(self, module: torch.nn.Module, process_group=None, bucket_bytes: int = 128 << 20, overlap: bool = True,
                 host_staged: Optional[bool] = None, force_collectives: Optional[bool] = None):
        """force_collectives (default: env MM_DP_FORCE_COLLECTIVES=1): issue the broadcast and every bucket all-reduce
        even when the process group has ONE rank -- the whole RCCL path (communicator, async all-reduce on RCCL's stream,
        hook-driven launches) then runs on a one-GPU box, where a second rank cannot share the device."""
        self.module = module
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if force_collectives is None:
            force_collectives = os.environ.get("MM_DP_FORCE_COLLECTIVES") == "1"
        self.collective = dist.is_initialized() and (self.world > 1 or bool(force_collectives))
        if host_staged is None:
            host_staged = dist.is_initialized() and dist.get_backend(process_group) == "gloo"
        self.host_staged = host_staged
        adjacent = [g for m in module.modules() if hasattr(m, "stacked_grad_groups") for g in m.stacked_grad_groups()]
        self.buckets = GradBuckets(list(module.parameters()), bucket_bytes, adjacent=adjacent)
        self.overlap = overlap and self.collective
        self._bucket_of = {}
        for b in self.buckets.buckets:
            for p in b["params"]:
                self._bucket_of[p] = b
        if self.collective:
            self.broadcast_parameters()
        self.optimizer = None
        self._prescaled = False
        self._hooked = False
        self._accumulating = False
        self._finished = False
        self._backward_done = False
        self._cb_queued = False
        self.n_allreduce = 0              # collectives issued so far (bench.py reports it: proof the exchange ran)
        if self.overlap:
            self._install_hooks()

    def _install_hooks(self):
        if not self._hooked:
            self._hooked = True
            for p in self.buckets.params:
                p.register_post_accumulate_grad_hook(self._hook)
                p._mm_ready = self._hook          # called by nodes that write p.grad themselves

    def attach_optimizer(self, optimizer):
        """Optimizer-in-backward: a bucket's parameters are updated (optimizer.step_subset, on the side stream) as soon
        as the bucket's gradients are final -- after its all-reduce when world > 1 -- while backward continues.  Valid
        for per-tensor optimizers without cross-parameter state (OrthoGradAdamW); optimizer.step() after finish() then
        only handles what is left.  Results are identical to stepping after backward."""
        self.optimizer = optimizer
        self._install_hooks()

    def broadcast_parameters(self, src: int = 0):
        for t in list(self.module.parameters()) + list(self.module.buffers()):
            if t.is_cuda and self.host_staged:
                h = t.data.cpu()
                dist.broadcast(h, src, group=self.pg)
                t.data.copy_(h)
            else:
                dist.broadcast(t.data, src, group=self.pg)

    def _all_reduce(self, flat):
        """Sum `flat` over the ranks, ordered after the current stream's work; returns a handle with wait()."""
        self.n_allreduce += 1
        if flat.is_cuda and self.host_staged:
            h = flat.cpu()                       # synchronises with the current stream
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.pg)
            flat.copy_(h)
            return _Done()
        if flat.is_cuda and _OWN_COMM_STREAM:
            # A synchronous collective runs on the CURRENT stream (torch >= 2.8; before that on the process group's internal
            # stream, with the current stream made to wait for it -- the event below marks completion either way).  Issued
            # from a stream of our own choosing, RCCL's kernels get a hardware queue that neither the main nor the side
            # stream uses (functional.comm_stream); the process group's internal stream is placed by creation order and was
            # seen to share a queue with compute, which serialises the two.
            cs = FN.comm_stream(flat.device)
            cs.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(cs):
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=False)
                ev = torch.cuda.Event()
                ev.record()
            return _EventWork(ev)
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def _launch(self, b):
        """Bucket b has all its gradients (or backward is over): complete it, reduce it, optionally update it.  On a GPU
        all of this is issued on the side stream, after everything the main stream has been given so far (gradients
        produced there, and the last uses of these weights in this backward)."""
        b["launched"] = True
        dev = b["flat"].device
        if dev.type != "cuda":
            self.buckets.gather(b)
            if self.collective:
                b["work"] = self._all_reduce(b["flat"])
            return
        main, side = FN.enter_side(dev)
        with torch.cuda.stream(side):
            self.buckets.gather(b)
            if self.collective:
                b["work"] = self._all_reduce(b["flat"])
            if self.optimizer is not None:
                if self.collective:
                    b["work"].wait()
                    if not self._prescaled and self.world > 1:
                        b["flat"].mul_(1.0 / self.world)
                    b["scaled"] = True
                self.optimizer.step_subset(b["params"], id(b))

    def _hook(self, p):
        if self._accumulating:        # inside accumulate(): gradients pile up in place, nothing is counted or reduced
            return
        if self.collective:
            if self._backward_done:
                raise RuntimeError("DataParallel: a second backward() since zero_grad() -- its buckets are already "
                                   "reduced.  For gradient accumulation run every backward but the last inside "
                                   "`with dp.accumulate():`")
            if not self._cb_queued:   # the engine calls this back when the backward pass that fired this hook is over
                self._cb_queued = True
                torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
        b = self._bucket_of[p]
        if id(p) in b["fired"]:       # a node that wrote p.grad itself has reported it; autograd's own hook may follow
            return
        b["fired"].add(id(p))
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)

    def _end_of_backward(self):
        self._cb_queued = False
        self._backward_done = True

    def scale_loss(self, loss):
        """loss / world for this step's backward: the summed all-reduce then IS the rank average and finish() skips its
        division pass over every gradient (one read+write of all parameters' worth of HBM per step)."""
        self._prescaled = True
        return loss if self.world == 1 else loss * (1.0 / self.world)

    @contextmanager
    def accumulate(self):
        """Gradient accumulation over several backward passes: run all but the LAST backward of a step inside this
        context.  Inside it no bucket is counted, reduced or updated (the gradients accumulate in place in their bucket
        slots); the last backward, outside it, fires the hooks and the ONE all-reduce per bucket."""
        prev, self._accumulating = self._accumulating, True
        try:
            yield self
        finally:
            self._accumulating = prev

    def zero_grad(self):
        self._prescaled = False
        self._finished = False
        self._backward_done = False
        self._cb_queued = False
        self.buckets.zero()
        for b in self.buckets.buckets:
            b["pending"] = len(b["params"])
            b["work"] = None
            b["launched"] = False
            b["scaled"] = False
            b["fired"] = set()
        if self.optimizer is not None:
            self.optimizer.begin_step()

    def finish(self):
        """Complete, reduce (and average) every bucket.  Parameters that received no gradient this step still take part
        (their bucket slots are zero), so every rank issues the same collectives in the same order."""
        if self._finished and self.collective:
            raise RuntimeError("DataParallel.finish() called twice without zero_grad(): the buckets are already reduced")
        todo = [b for b in self.buckets.buckets if not b.get("launched", False)]
        if todo and not self.collective and self.optimizer is None and todo[0]["flat"].is_cuda:
            # one process, nothing to exchange or update per bucket: complete every bucket at once on the main stream.  (Per
            # bucket on the side stream -- a wait, two small multi-tensor kernels and a join each -- the nine buckets of the
            # config model cost 1.05 ms of cross-queue dependencies for 0.2 ms of copying at the end of every backward.)
            FN.join_side_stream()
            self.buckets.gather(todo)
            for b in todo:
                b["launched"] = True
        else:
            for b in todo:
                self._launch(b)
            FN.join_side_stream()
        if self.collective:
            for b in self.buckets.buckets:
                if not b.get("scaled", False):
                    b["work"].wait()
                    if not self._prescaled and self.world > 1:
                        b["flat"].mul_(1.0 / self.world)
                    b["scaled"] = True
        self._finished = True

    def __call__(self, *a, **kw):
        return self.module(*a, **kw)
