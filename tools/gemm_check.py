#!/usr/bin/env python3
"""NT GEMM check + timing at the layer shapes (development tool).  Env SPV_GEMM_KB forces the direct-to-LDS kernel (tile-shape variants live in tools/gemm_lab.hip).

    python tools/gemm_check.py [M N K ...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-spectre-experiments_amd"))
import torch  # noqa: E402

from spectre_vit import hip_ops as H  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16


def run(M, N, K, acc=0, iters=30):
    torch.manual_seed(0)
    A = torch.randn(M, K, device=dev).to(bf)
    B = torch.randn(N, K, device=dev).to(bf)
    bias = torch.randn(N, device=dev)
    C = torch.zeros(M, N, device=dev, dtype=bf)
    H._gemm_launch(A, B, bias, C, M, N, K, K, K, N, 0, 1, None)
    torch.cuda.synchronize()
    rows = torch.cat([torch.arange(0, min(M, 300)), torch.arange(max(0, M - 300), M)]).to(dev)
    ref = A[rows].float() @ B.float().t() + bias
    err = (C[rows].float() - ref).abs().max().item() / ref.abs().max().item()
    fn = lambda: H._gemm_launch(A, B, None, C, M, N, K, K, K, N, acc, 1, None)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"gemm {M}x{N}x{K} acc={acc}: rel err {err:.2e}  {us:8.2f} us  {2.0 * M * N * K / us * 1e-6:7.1f} TFLOP/s "
          f"[KB={os.environ.get('SPV_GEMM_KB', '-')}]", flush=True)
    assert err < 2e-2, err


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    shapes = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)] or [(33280, 768, 512), (33280, 512, 768), (33280, 512, 8192),
                                                                   (33280, 8192, 512), (4096, 4096, 4096), (33000, 760, 544)]
    for s in shapes:
        run(*s)
