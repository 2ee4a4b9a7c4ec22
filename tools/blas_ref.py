"""Yardstick only (never on the product path): what the vendor GEMM (hipBLASLt through torch.matmul) reaches on this box at the
shapes of the training step, next to our kernels in tools/kbench.py.  Kernel names (tile shapes) come out of
`rocprofv3 --kernel-trace --stats -- python3 tools/blas_ref.py`."""
import sys
import torch

dev = torch.device("cuda:0")
bf = torch.bfloat16


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def main():
    shapes = [  # (M, N, K, form)  NT: C = A[M,K] . B[N,K]^T ; TN: C = A[K,M]^T . B[K,N]
        (33280, 768, 512, "NT"), (33280, 512, 768, "NT"), (33280, 512, 8192, "NT"), (33280, 8192, 512, "NT"), (4096, 4096, 4096, "NT"),
        (768, 512, 33280, "TN"), (512, 8192, 33280, "TN"),
    ]
    for M, N, K, form in shapes:
        if form == "NT":
            A = torch.randn(M, K, device=dev).to(bf)
            B = torch.randn(N, K, device=dev).to(bf)
            fn = lambda: torch.matmul(A, B.t())
        else:
            A = torch.randn(K, M, device=dev).to(bf)
            B = torch.randn(K, N, device=dev).to(bf)
            fn = lambda: torch.matmul(A.t(), B)
        us = timeit(fn)
        fl = 2.0 * M * N * K
        print(f"hipblaslt {form} {M}x{N}x{K:<6d} {us:9.2f} us  {fl / us * 1e-6:8.1f} TFLOP/s ({fl / us * 1e-6 / 2500 * 100:5.1f}% of 2.5 PF)")


if __name__ == "__main__":
    main()
