"""Data-parallel gradient exchange for the Spectre-ViT training step.

The reference is single-process (spectre_vit/repl/train.py:41); the step shards naturally over images because no
op mixes samples, so the only exchange is one gradient all-reduce per step.  One process per GPU,
``torch.distributed`` backend "nccl" (= RCCL over xGMI on ROCm); gloo on CPU for the tests.

GradReducer lays a few flat fp32 buckets (8 MB, the last-completing one 2 MB) out in reverse registration order (~ the order backward produces the
gradients: head -> last layer -> ... -> embedding) and hands every parameter its slot as a ``GradSink``: the HIP backward
kernels write dW / db / dgamma ... directly into the bucket, autograd adopts that view as ``p.grad`` (no extra
``grad += new`` pass).  A post-accumulate hook (which also stages gradients that arrived from other ops) counts a
bucket's parameters; when the bucket is complete its all-reduce is launched
asynchronously on RCCL's own stream while backward continues, and ``finish()`` waits for all of them before the
optimizer runs.  xGMI is point-to-point (7 links x ~153 GB/s), so a few multi-megabyte buckets are used rather than many
small ones; ``reduce_dtype=torch.bfloat16`` halves the bytes on the links (sum still accumulated by RCCL in bf16,
so it is off by default).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


RESERVED_CUS = 16  # CUs left to RCCL while gradients are exchanged during the backward (bench.py caps RCCL's channels to match)


class GradReducer:
    def __init__(self, module: torch.nn.Module, bucket_mb: float = 8.0, process_group=None, reduce_dtype=None,
                 tail_mb: float = 2.0, always: bool = False, overlap: bool = True):
        """always=True: lay the buckets out even in a single process (no exchange then): every gradient gets a FIXED address, which
        a captured HIP graph and the one-launch optimizer's pointer table need (spectre_vit/graph.py, spectre_vit/optim.py).
        overlap=False: the hooks launch nothing; ``finish()`` exchanges ALL buckets with one all-reduce over the flat buffer they are
        views of (spectre_vit.graph.GraphedDPStep: the backward pass is a replayed HIP graph, there is no host in it to launch a
        collective from, and a 13 MB exchange is one call).  No CUs are set aside for RCCL then: nothing runs beside it."""
        self.group = process_group
        self.overlap = bool(overlap)
        self.flat = None
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.reduce_dtype = reduce_dtype
        # RCCL averages in the collective itself (ncclAvg); gloo only sums -> scale afterwards
        self._avg = self.world > 1 and dist.get_backend(process_group) == "nccl" and reduce_dtype in (None, torch.float32)
        self.adopted = 0   # gradients that arrived in the hook already living in their bucket slot (no staging copy) this step
        self.staged = 0    # gradients that had to be copied into their slot
        self.adopted_numel = self.staged_numel = 0
        self.buckets = []  # dict(flat, params, pending, handle, stage)
        self._bucket_of = {}
        if not self.params or (self.world == 1 and not always):
            return  # single process: no exchange, gradients stay ordinary per-parameter tensors
        if self.params[0].is_cuda and self.world > 1 and self.overlap:
            # the layer GEMMs run one workgroup per CU and cannot share a CU with a resident RCCL channel: leave 16 CUs to the
            # collectives that overlap the backward (free at the layer shapes, see spv.h: spv_set_reserved_cus)
            try:
                from spectre_vit import _native
                _native.call("spv_set_reserved_cus", RESERVED_CUS)
            except Exception:  # model-agnostic reducer on a stock model without the library
                pass
        try:
            from spectre_vit.hip_ops import GradSink
        except Exception:  # the reducer itself is model agnostic (CPU gloo tests use a stock model)
            GradSink = None
        cap = int(bucket_mb * 1024 * 1024 / 4)
        # The bucket that completes LAST (the first-registered parameters: embedding, first layer) has nothing left to hide
        # behind, so it is kept small: its all-reduce is the exposed tail of the step.
        tail_cap = int(min(tail_mb, bucket_mb) * 1024 * 1024 / 4)
        tail, tail_n = [], 0
        for p in self.params:
            if tail and tail_n + p.numel() > tail_cap:
                break
            tail.append(p)
            tail_n += p.numel()
        cur, cur_n = [], 0
        groups = []
        for p in reversed(self.params[len(tail):]):
            if cur and cur_n + p.numel() > cap:
                groups.append(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            groups.append(cur)
        groups.append(list(reversed(tail)))
        # ONE allocation behind all buckets (each bucket a 256-byte-aligned slice of it): the non-overlapped exchange is a single
        # all-reduce over it; the overlapped one still goes bucket by bucket
        starts, total = [], 0
        for ps in groups:
            starts.append(total)
            total += (sum(p.numel() for p in ps) + 63) // 64 * 64
        self.flat = torch.zeros(total, dtype=torch.float32, device=self.params[0].device)
        for bi, ps in enumerate(groups):
            n = sum(p.numel() for p in ps)
            flat = self.flat[starts[bi]:starts[bi] + n]
            off = 0
            views = []
            for p in ps:
                v = flat[off:off + p.numel()].view_as(p)
                views.append(v)
                if GradSink is not None and p.is_cuda:
                    p._spv_grad_sink = GradSink(v)  # backward kernels write the gradient straight into the bucket
                off += p.numel()
                self._bucket_of[p] = bi
            self.buckets.append(dict(flat=flat, params=ps, views=views, pending=len(ps), handle=None, stage=None))
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._hook)

    # -- per step ---------------------------------------------------------------------------------
    def zero_grad(self):
        """replaces optimizer.zero_grad(set_to_none=True); re-arms the buckets."""
        self.adopted = self.staged = self.adopted_numel = self.staged_numel = 0
        for p in self.params:
            p.grad = None
            s = getattr(p, "_spv_grad_sink", None)
            if s is not None:
                s.used = False
        for b in self.buckets:
            b["pending"] = len(b["params"])
            b["handle"] = None

    def _launch(self, b):
        if self.reduce_dtype is not None and self.reduce_dtype != torch.float32:
            b["stage"] = b["flat"].to(self.reduce_dtype)
            b["handle"] = dist.all_reduce(b["stage"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            b["handle"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, group=self.group,
                                          async_op=True)

    def _hook(self, p):
        b = self.buckets[self._bucket_of[p]]
        idx = b.setdefault("index", {id(q): i for i, q in enumerate(b["params"])})[id(p)]
        v = b["views"][idx]
        if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
            v.copy_(p.grad)  # gradient came from an op that did not write into the bucket (or autograd cloned it): stage it
            self.staged += 1
            self.staged_numel += p.numel()
        elif p.grad is not None:
            self.adopted += 1
            self.adopted_numel += p.numel()
        p.grad = v
        b["pending"] -= 1
        if b["pending"] == 0 and self.world > 1 and self.overlap:
            self._launch(b)

    def finish(self):
        """wait for every bucket's all-reduce and turn the sum into the mean over ranks."""
        if self.world == 1:
            return
        inv = 1.0 / self.world
        if not self.overlap:
            self.allreduce_flat()
            return
        for b in self.buckets:
            if b["handle"] is None:  # a parameter of this bucket received no gradient this step: its slot is stale -> zero
                for q, v in zip(b["params"], b["views"]):
                    if q.grad is None:
                        v.zero_()
                self._launch(b)
            b["handle"].wait()
            if b["stage"] is not None:
                b["flat"].copy_(b["stage"])
                b["stage"] = None
            if not self._avg:
                b["flat"].mul_(inv)

    def allreduce_flat(self, force: bool = False):
        """the non-overlapped exchange: every bucket in ONE all-reduce (mean over ranks) on the caller's stream order.  The caller
        guarantees that every slot holds this step's gradient (a replayed graph writes all of them; eager callers with parameters that
        may receive no gradient use the overlapped mode, whose finish() zeroes stale slots).  force=True issues the collective in a
        one-rank process group too (bench.py --dp-sequence: the launch sequence of a rank, measured on one GPU)."""
        if self.flat is None or (self.world == 1 and not (force and dist.is_available() and dist.is_initialized())):
            return
        buf = self.flat
        if self.reduce_dtype is not None and self.reduce_dtype != torch.float32:
            stage = buf.to(self.reduce_dtype)
            dist.all_reduce(stage, op=dist.ReduceOp.SUM, group=self.group)
            buf.copy_(stage)
            if self.world > 1:
                buf.mul_(1.0 / self.world)
            return
        dist.all_reduce(buf, op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, group=self.group)
        if not self._avg and self.world > 1:
            buf.mul_(1.0 / self.world)


def broadcast_module(module: torch.nn.Module, src: int = 0, process_group=None):
    """make parameters AND buffers (perms/signs are RNG-drawn at construction, reference layers.py:61-64) identical
    on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src, group=process_group)
