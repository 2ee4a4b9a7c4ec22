"""The whole training step as ONE HIP graph (MI355X: ~100 kernel launches per 2.5 ms step leave the host one hiccup from being
the bottleneck; a replay costs the host ~15 us).

    step = GraphedTrainStep(model, optimizer, criterion, example_img, example_labels, autocast_dtype=torch.bfloat16)
    loss = step(img, labels)        # copies the batch into the captured buffers, replays, returns the (device) loss

What is captured: zero_grad -> forward (autocast) -> loss -> backward -> optimizer.step(), i.e. reference
spectre_vit/repl/train.py:216-238 minus the host-side bookkeeping.  Requirements:
  * the optimizer must keep its step count on the device: ``spectre_vit.optim.FusedAdamW(capturable=True)`` or
    ``torch.optim.AdamW(fused=True, capturable=True)``;
  * dropout: seeds are by-value kernel arguments frozen at capture, so every dropout kernel adds a 64-bit device word that this
    class advances once per replay (``spv_seed_advance`` is the first node of the graph) -- fresh masks every step;
  * fixed shapes (one graph per batch shape), single process (data parallel runs stay eager: their gradient all-reduce hides the
    host anyway).
"""
from __future__ import annotations

import torch

from spectre_vit import _native, hip_ops
from spectre_vit.dp import GradReducer


class GraphedTrainStep:
    def __init__(self, model, optimizer, criterion, example_img, example_labels, autocast_dtype=torch.bfloat16, warmup=3,
                 return_features=False):
        if not example_img.is_cuda:
            raise RuntimeError("GraphedTrainStep needs the example batch on the GPU")
        for g in optimizer.param_groups:
            if not g.get("capturable", False):
                raise ValueError("GraphedTrainStep: build the optimizer with capturable=True (its step count must live on the device)")
        self.model, self.optimizer, self.criterion = model, optimizer, criterion
        self.autocast_dtype = autocast_dtype
        self.img = example_img.clone()
        self.labels = example_labels.clone()
        dev = self.img.device
        # fixed gradient addresses (the graph replays into them; the optimizer's pointer table is built once, before the capture)
        self.reducer = GradReducer(model, always=True)
        self.seed_word = torch.zeros(1, dtype=torch.int64, device=dev)
        _native.call("spv_set_seed_device_ptr", self.seed_word.data_ptr())
        self._st = None
        one = torch.ones((), dtype=torch.float32, device=dev)

        def one_step():
            _native.call("spv_seed_advance", self.seed_word.data_ptr(), torch.cuda.current_stream().cuda_stream)
            self.reducer.zero_grad()
            with torch.autocast("cuda", dtype=self.autocast_dtype, enabled=self.autocast_dtype is not None):
                out = self.model(self.img)
            loss = self.criterion(out, self.labels)
            loss.backward(one)   # a kept 1.0 (backward() alone fills a fresh one: a launch per step)
            self.reducer.finish()
            self.optimizer.step()
            return loss, out

        # warm-up on a side stream (allocator pools, lazily built tables, optimizer state), then capture
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                one_step()
        torch.cuda.current_stream().wait_stream(side)
        hip_ops._shadows = type(hip_ops._shadows)()  # the weight casts must be recorded in the graph, not served from a cache
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.out = one_step()

    def __call__(self, img=None, labels=None):
        if img is not None:
            self.img.copy_(img, non_blocking=True)
        if labels is not None:
            self.labels.copy_(labels, non_blocking=True)
        self.graph.replay()
        return self.loss

    def close(self):
        """eager kernels after this object's life must not read its seed word"""
        _native.call("spv_set_seed_device_ptr", 0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
