#!/usr/bin/env python3
"""Weight-gradient (TN) GEMM at the layer shapes: LDS-DMA kernel vs the register-staged one vs fp64, and GPU time per launch
including the split-K reduce (development tool; run on the GPU box).   SPV_TN_DMA=0 selects the old kernel."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-spectre-experiments_amd"))
import torch  # noqa: E402

from spectre_vit import _native  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    bf = torch.bfloat16
    p = lambda t: t.data_ptr()  # noqa: E731
    st = torch.cuda.current_stream().cuda_stream
    iters = 100
    cases = ((768, 512, 33280, (8, 10, 13, 20)), (512, 768, 33280, (10,)), (512, 8192, 33280, (1,)), (256, 128, 640, (1, 2)))
    if len(sys.argv) >= 5:   # M N K s1,s2,...
        cases = ((int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), tuple(int(v) for v in sys.argv[4].split(","))),)
    for (M, N, K, splits_list) in cases:
        g = torch.Generator().manual_seed(M + N)
        A = (torch.randn(K, M, generator=g) * 0.5).to(bf).to(dev)
        Bm = (torch.randn(K, N, generator=g) * 0.5).to(bf).to(dev)
        ref = None
        if K * M * N < 3e10:
            ref = (A.double().T @ Bm.double()).cpu().numpy()
        for splits in splits_list:
            C = torch.zeros(M, N, device=dev)
            ws = torch.empty(max(splits, 1) * M * N, device=dev)
            fn = lambda: _native.call("spv_gemm_tn", p(A), p(Bm), p(C), M, N, K, M, N, N, 0, 0, splits, p(ws) if splits > 1 else 0, st)  # noqa: E731
            before = _native.call("spv_path_count", _native.PATH["gemm_tn_dma"])
            fn()
            torch.cuda.synchronize()
            dma = _native.call("spv_path_count", _native.PATH["gemm_tn_dma"]) - before
            err = ""
            if ref is not None:
                got = C.cpu().numpy().astype(np.float64)
                err = f"rel-L2 {np.linalg.norm(got - ref) / np.linalg.norm(ref):.2e} max {np.abs(got - ref).max() / np.abs(ref).max():.2e}"
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / iters
            print(f"tn {M}x{N}x{K} splits={splits} dma={dma}: {us:8.2f} us  {2.0 * M * N * K / us * 1e-6:7.1f} TFLOP/s ({2.0 * M * N * K / us * 1e-6 / 25:5.1f} % of 2.5 PF)  {err}", flush=True)


if __name__ == "__main__":
    main()
