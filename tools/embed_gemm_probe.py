"""where the embedding's token GEMM (33280 x 512 x 48, 34 MB out) spends its 33 us: plain / grouped rows / + 2-D bias / + dropout"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vit-spectre-experiments_amd"), ROOT]
import torch
from spectre_vit import _native
from spectre_vit.hip_ops import _p, _stream

dev = torch.device("cuda:0")
bf = torch.bfloat16
M, N, K, T = 33280, 512, 48, 65
a = torch.randn(M, K, device=dev).to(bf)
w = torch.randn(N, K, device=dev).to(bf)
bias = torch.randn(N, device=dev)
b2 = torch.randn(T, N, device=dev)
c = torch.empty(M, N, device=dev, dtype=bf)


def t(fn, iters=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


st = _stream()
cases = {
    "plain spv_gemm_nt": lambda: _native.call("spv_gemm_nt", _p(a), _p(w), 0, _p(c), M, N, K, K, K, N, 1, 1, 0, 1, 0, st),
    "plain + bias": lambda: _native.call("spv_gemm_nt", _p(a), _p(w), _p(bias), _p(c), M, N, K, K, K, N, 1, 1, 0, 1, 0, st),
    "grouped rows (T, T, 0), no bias2d": lambda: _native.call("spv_gemm_nt_grouped_rows", _p(a), _p(w), 0, 0, _p(c), M, N, K, K, K, N, 1, 1, T, T, 0, st),
    "grouped rows + bias2d": lambda: _native.call("spv_gemm_nt_grouped_rows", _p(a), _p(w), 0, _p(b2), _p(c), M, N, K, K, K, N, 1, 1, T, T, 0, st),
    "grouped rows + bias2d + dropout 0.001": lambda: _native.call("spv_gemm_nt_grouped_rows_drop", _p(a), _p(w), 0, _p(b2), _p(c), M, N, K, K, K, N, 1, 1,
                                                                  T, T, 0, 0.001, 1234, st),
}
for name, fn in cases.items():
    print(f"{name:45s} {t(fn):7.2f} us", flush=True)
# for scale: a plain copy of the output's bytes
src = torch.empty_like(c)
print(f"{'copy of 34 MB (torch)':45s} {t(lambda: c.copy_(src)):7.2f} us")
