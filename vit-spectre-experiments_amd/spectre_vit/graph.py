"""The training step replayed from HIP graphs (MI355X: ~70 kernel launches per 1.6 ms step leave an eager host behind the GPU;
a replay costs the host ~15 us).

    step = GraphedTrainStep(model, optimizer, criterion, example_img, example_labels, autocast_dtype=torch.bfloat16)
    loss = step(img, labels)        # copies the batch into the captured buffers, replays, returns the (device) loss

What is captured: zero_grad -> forward (autocast) -> loss -> backward -> optimizer.step(), i.e. reference
spectre_vit/repl/train.py:216-238 minus the host-side bookkeeping.  Requirements:
  * the optimizer must keep its step count on the device: ``spectre_vit.optim.FusedAdamW(capturable=True)`` or
    ``torch.optim.AdamW(fused=True, capturable=True)``;
  * dropout: seeds are by-value kernel arguments frozen at capture, so every dropout kernel adds a 64-bit device word that this
    class advances once per replay (``spv_seed_advance`` is the first node of the graph) -- fresh masks every step;
  * fixed shapes (one graph per batch shape).

``GraphedTrainStep`` is the single-process form: ONE graph.  ``GraphedDPStep`` is a data-parallel rank (the reference is single
device, train.py:41; SURVEY 8e makes the gradient exchange new work):

    graph A   seed word, zero_grad, forward, loss, backward -- with the layer weight gradients held back and computed by one batched
              launch, exactly as in the single-process graph -- every gradient written into the reducer's flat buffer
    host      ONE all-reduce over that buffer (torch.distributed: RCCL over xGMI; 13 MB for the FFT model)
    graph B   the optimizer step

so a rank's host issues two replays and one collective per step instead of ~70 launches, and the eager path's three handicaps
(host issue time above the GPU time, weight gradients one by one, no side stream) are gone.  The collective is NOT captured: RCCL
inside a graph has not run on a multi-GPU box in this project, and the exchange is one call either way.
"""
from __future__ import annotations

import contextlib
import weakref

import torch

from spectre_vit import _native, hip_ops
from spectre_vit.dp import GradReducer

_seed_owner = None   # weakref to the live step object whose seed word the library's dropout kernels read


def _claim_seed_word(obj):
    global _seed_owner
    cur = _seed_owner() if _seed_owner is not None else None
    if cur is not None and cur is not obj and not cur._closed:
        raise RuntimeError("a graph-replayed training step is already live in this process: the dropout seed word is one device "
                           "symbol of libspv_hip.so, so two live steps would share (or steal) each other's masks -- call .close() "
                           "on the old step first")
    _native.call("spv_set_seed_device_ptr", obj.seed_word.data_ptr())
    _seed_owner = weakref.ref(obj)


def _release_seed_word(obj):
    """eager kernels after this object's life must not read its seed word; only the CURRENT owner may clear the pointer (an old
    step's __del__ running after a new step registered its word must leave the new word in place)"""
    global _seed_owner
    cur = _seed_owner() if _seed_owner is not None else None
    if cur is obj or cur is None:
        if _seed_owner is not None:
            _native.call("spv_set_seed_device_ptr", 0)
        _seed_owner = None


class GraphedTrainStep:
    _dp = False

    def __init__(self, model, optimizer, criterion, example_img, example_labels, autocast_dtype=torch.bfloat16, warmup=3,
                 return_features=False, process_group=None, force_collective=False):
        if not example_img.is_cuda:
            raise RuntimeError(f"{type(self).__name__} needs the example batch on the GPU")
        for g in optimizer.param_groups:
            if not g.get("capturable", False):
                raise ValueError(f"{type(self).__name__}: build the optimizer with capturable=True (its step count must live on the device)")
        self._closed = True   # until the seed word is claimed
        self.model, self.optimizer, self.criterion = model, optimizer, criterion
        self.autocast_dtype = autocast_dtype
        self.img = example_img.clone()
        self.labels = example_labels.clone()
        dev = self.img.device
        self.force_collective = bool(force_collective)
        # fixed gradient addresses (the graphs replay into them; the optimizer's pointer table is built once, before the capture).
        # A rank of a data-parallel job exchanges them with one call between its two graphs: no hook launches anything.
        self.reducer = GradReducer(model, always=True, process_group=process_group, overlap=not self._dp)
        if not self._dp and self.reducer.world > 1:
            raise RuntimeError("GraphedTrainStep is the single-process step; a rank of a torch.distributed job uses GraphedDPStep")
        self.seed_word = torch.zeros(1, dtype=torch.int64, device=dev)
        _claim_seed_word(self)
        self._closed = False
        self._one = torch.ones((), dtype=torch.float32, device=dev)
        self.replays = 0
        try:
            self._build(max(1, warmup))
        except BaseException:
            self.close()
            raise

    # -- the two halves of a step -----------------------------------------------------------------------------------------------
    def _forward_backward(self):
        _native.call("spv_seed_advance", self.seed_word.data_ptr(), torch.cuda.current_stream().cuda_stream)
        self.reducer.zero_grad()
        with torch.autocast("cuda", dtype=self.autocast_dtype, enabled=self.autocast_dtype is not None):
            out = self.model(self.img)
        loss = self.criterion(out, self.labels)
        loss.backward(self._one)   # a kept 1.0 (backward() alone fills a fresh one: a launch per step)
        return loss, out

    def _eager_step(self):
        loss, out = self._forward_backward()
        self.reducer.finish()
        self.optimizer.step()
        return loss, out

    @contextlib.contextmanager
    def _hold(self):
        """while this object's backward passes run: layer weight gradients and folds may be held for the batched launch even in a
        multi-rank job (nothing is exchanged before the pass ends)"""
        prev = hip_ops.HOLD_UNDER_DP
        hip_ops.HOLD_UNDER_DP = prev or self._dp
        try:
            yield
        finally:
            hip_ops.HOLD_UNDER_DP = prev

    def _warm(self, warmup):
        # warm-up on a side stream (allocator pools, lazily built tables, optimizer state), then capture
        side = torch.cuda.Stream(device=self.img.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), self._hold():
            for _ in range(warmup):
                loss, out = self._eager_step()
        torch.cuda.current_stream().wait_stream(side)
        # the warm-up steps are REAL training steps on the example batch: a caller that feeds the example batch as its first batch
        # takes the last one's results from here instead of replaying that batch a second time
        self.warm_loss, self.warm_out = loss.detach(), out.detach()
        hip_ops._shadows = type(hip_ops._shadows)()  # the weight casts must be recorded in the graph, not served from a cache

    def _build(self, warmup):
        self._warm(warmup)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), self._hold():
            self.loss, self.out = self._eager_step()

    def _replay(self):
        self.graph.replay()

    def __call__(self, img=None, labels=None):
        if self._closed:
            raise RuntimeError("this graph-replayed step was closed")
        if img is not None:
            self.img.copy_(img, non_blocking=True)
        if labels is not None:
            self.labels.copy_(labels, non_blocking=True)
        self._replay()
        self.replays += 1
        # the replay updated the weights through raw pointers: neither the parameters' version counters nor the optimizer's post-step
        # hook saw it, so the inference-time cache of bf16 weight copies (hip_ops._ShadowCache) must be told
        hip_ops.invalidate_weight_shadows()
        return self.loss

    def close(self):
        if not self._closed:
            self._closed = True
            _release_seed_word(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GraphedDPStep(GraphedTrainStep):
    """One rank of the data-parallel step: graph A (forward + loss + backward, batched weight gradients) -> one all-reduce of the flat
    gradient buffer (mean over ranks) -> graph B (optimizer).  Build it AFTER ``torch.distributed.init_process_group`` and after the
    rank-0 weights / buffers were broadcast (``spectre_vit.dp.broadcast_module``); every rank must construct it (the warm-up steps
    contain the collective).  In a single process it runs the same launch sequence with the collective skipped
    (``force_collective=True``: issued in the one-rank group, when one is initialised)."""
    _dp = True

    def _eager_step(self):
        loss, out = self._forward_backward()
        self.reducer.allreduce_flat(force=self.force_collective)
        self.optimizer.step()
        return loss, out

    def _build(self, warmup):
        self._warm(warmup)
        torch.cuda.synchronize()   # the warm-up's collectives are complete before a capture begins
        pool = torch.cuda.graph_pool_handle()
        # capture_error_mode "thread_local": only THIS thread's calls are checked against the capture.  A process group keeps a
        # watchdog thread that queries its work events at any time; under the default ("global") mode such a query from another
        # thread can invalidate an ongoing capture.  Nothing of the collective is captured here, so thread-local checking is exact.
        mode = dict(capture_error_mode="thread_local")
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, pool=pool, **mode), self._hold():
            self.loss, self.out = self._forward_backward()
        self.graph_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_opt, pool=pool, **mode):
            self.optimizer.step()

    def _replay(self):
        self.graph.replay()
        self.reducer.allreduce_flat(force=self.force_collective)   # host-issued, on the current stream's order: RCCL over xGMI
        self.graph_opt.replay()
