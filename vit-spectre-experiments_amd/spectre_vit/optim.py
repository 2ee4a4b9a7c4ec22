"""AdamW on one HIP launch per step -- drop-in for ``torch.optim.AdamW`` as the reference's script builds it
(spectre_vit/repl/train.py:199-201: ``optim.AdamW(model.parameters(), betas=..., lr=..., weight_decay=...)``).

Same update rule, same ``state`` layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter, so ``state_dict()`` is
interchangeable with torch's), one parameter group or many.  torch's own fused kernel spends 100+ us per step on the FFT
model's 63 small tensors; ``spv_adamw_multi`` walks a chunk table in a single launch.  ``capturable=True`` keeps the step
count on the device, so the whole training step can be captured in a HIP graph (spectre_vit/graph.py).
"""
from __future__ import annotations

import torch

from spectre_vit import _native
from spectre_vit.hip_ops import _stream

_CHUNK = 2048


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, capturable=False, static_grads=False):
        """static_grads=True: the caller expects fixed gradient addresses (spectre_vit.dp.GradReducer(model, always=True), or a
        captured graph): after two identical look-ups a step only COMPARES the gradient addresses with the table's (one tuple of
        ``p.grad.data_ptr()`` per group, every step: a caller that swapped a ``.grad`` gets a rebuilt table, never a write through a
        stale pointer) and skips the state walk and the table build -- 0.3 ms of host time per step otherwise."""
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"FusedAdamW: invalid hyper-parameters lr={lr} betas={betas} eps={eps} weight_decay={weight_decay}")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=capturable))
        self.static_grads = bool(static_grads)
        self._tables = {}  # group index -> dict(key, table, chunk_tensor, chunk_off, sizes, nchunks, step_dev)

    def _table(self, gi, ps):
        """device-side pointer / chunk tables of one parameter group; rebuilt only when a pointer moved (in the steady state the
        caching allocator hands the same gradient blocks back every step; with GradReducer the gradients live in fixed buckets)"""
        # (the moment tensors are part of the key: load_state_dict() on an optimizer that has already stepped replaces them)
        key = tuple((p.data_ptr(), p.grad.data_ptr(), p.numel(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr())
                    for p in ps)
        t = self._tables.get(gi)
        if t is not None and t["key"] == key:
            return t
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("FusedAdamW: a parameter or gradient moved while a HIP graph is being captured (its pointer table "
                               "cannot be uploaded inside a capture): give the gradients fixed addresses with "
                               "spectre_vit.dp.GradReducer(model, always=True) and run a warm-up step first")
        dev = ps[0].device
        rows, ct, co, sizes = [], [], [], []
        for i, p in enumerate(ps):
            st = self.state[p]
            if not p.grad.is_contiguous() or p.grad.dtype != torch.float32:
                raise RuntimeError("FusedAdamW needs contiguous fp32 gradients")
            rows += [p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()]
            n = p.numel()
            sizes.append(n)
            for off in range(0, n, _CHUNK):
                ct.append(i)
                co.append(off)
        t = dict(key=key,
                 table=torch.tensor(rows, dtype=torch.int64).to(dev), chunk_tensor=torch.tensor(ct, dtype=torch.int32).to(dev),
                 chunk_off=torch.tensor(co, dtype=torch.int32).to(dev), sizes=torch.tensor(sizes, dtype=torch.int32).to(dev),
                 nchunks=len(ct))
        prev = self._tables.get(gi)
        if prev is not None:  # the step counter outlives a table rebuild
            for k in ("step", "host_step"):
                if k in prev:
                    t[k] = prev[k]
        self._tables[gi] = t
        return t

    def load_state_dict(self, state_dict):
        """torch's loader replaces the per-parameter state (moments, step count): drop the device tables so that the next step() reads
        the loaded step count and points at the loaded moment tensors (in-place resume / rollback of an optimizer that has stepped)."""
        super().load_state_dict(state_dict)
        self._tables = {}
        for st in self.state.values():
            if isinstance(st.get("step"), torch.Tensor):
                st["step"] = st["step"].float()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            t = self._tables.get(gi)
            b1, b2 = group["betas"]
            if (t is not None and t.get("static")
                    and t["grad_ptrs"] == tuple(0 if p.grad is None else p.grad.data_ptr() for p in group["params"])):
                # fixed gradient addresses (GradReducer(always=True) / a captured graph), verified: nothing to look up
                t["calls"] += 1
            else:
                ps = [p for p in group["params"] if p.grad is not None]
                if not ps:
                    continue
                for p in ps:
                    if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                        raise RuntimeError("FusedAdamW: parameters must be contiguous fp32 tensors on the GPU (no CPU fallback)")
                    st = self.state[p]
                    if not st:
                        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        st["step"] = None
                had = t is not None
                old_key = t["key"] if had else None
                t = self._table(gi, ps)
                if "step" not in t:  # ONE step counter per group, shared by its parameters' state entries
                    prev = next((self.state[p]["step"] for p in ps if self.state[p].get("step") is not None), None)
                    dev = ps[0].device if group["capturable"] else "cpu"
                    t["step"] = prev.to(dev).clone() if prev is not None else torch.zeros((), dtype=torch.float32, device=dev)
                    t["host_step"] = int(float(t["step"]))
                for p in ps:
                    self.state[p]["step"] = t["step"]
                # static once the same table has served two consecutive look-ups
                t["static"] = self.static_grads and had and old_key == t["key"]
                # what the fast path compares every step: the gradient address of EVERY parameter of the group (0 = no gradient)
                t["grad_ptrs"] = tuple(0 if p.grad is None else p.grad.data_ptr() for p in group["params"])
                t["calls"] = 1
            if group["capturable"]:
                t["step"] += 1.0
                bc1 = bc2 = 0.0
                step_ptr = t["step"].data_ptr()
            else:
                t["host_step"] += 1
                t["step"].fill_(float(t["host_step"]))
                bc1, bc2 = 1.0 - b1 ** t["host_step"], 1.0 - b2 ** t["host_step"]
                step_ptr = 0
            _native.call("spv_adamw_multi", t["table"].data_ptr(), t["chunk_tensor"].data_ptr(), t["chunk_off"].data_ptr(),
                         t["sizes"].data_ptr(), t["nchunks"], float(group["lr"]), float(b1), float(b2), 1.0 - b1, 1.0 - b2,
                         float(group["eps"]), float(group["weight_decay"]), bc1, bc2, step_ptr, _stream())
        return loss
