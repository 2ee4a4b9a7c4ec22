"""Isolated timing of the strip GEMM with and without the accumulate read (development aid)."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vit-spectre-experiments_amd"))
from spectre_vit import hip_ops as ops

dev = torch.device("cuda:0")
M = 33280
for N, K in ((768, 512), (512, 768)):
    A = torch.randn((M, K), device=dev).to(torch.bfloat16)
    B = (torch.randn((N, K), device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn((N,), device=dev)
    C = torch.zeros((M, N), device=dev, dtype=torch.bfloat16)
    for name, bi, acc in (("bias", bias, 0), ("accumulate", None, 1)):
        for r in range(3):
            for _ in range(5):
                ops._gemm(A, B, bi, C, M, N, K, K, K, N, acc, 1, None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops._gemm(A, B, bi, C, M, N, K, K, K, N, acc, 1, None)
            e1.record()
            torch.cuda.synchronize()
            print(f"N={N} K={K} {name:10s} round {r}: {e0.elapsed_time(e1) / 20 * 1e3:7.2f} us", flush=True)
