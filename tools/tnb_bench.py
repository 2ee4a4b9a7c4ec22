"""Batched weight gradients (spv_gemm_tn_batch) at the step's shapes: correctness vs fp32 matmul + time per launch by split count.
    SPV_LAB=1 SPV_LIB_PATH=.../libspv_hip_lab.so SPV_TNB_WIDE=0|1 python tools/tnb_bench.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vit-spectre-experiments_amd"), ROOT]
import torch
from spectre_vit import _native
from spectre_vit.hip_ops import _p, _stream

dev = torch.device("cuda:0")
rows = int(os.environ.get("ROWS", "33280"))
shapes = [(768, 512), (512, 768)] * 3
nshort = int(os.environ.get("SHORT", "0"))   # extra problems over the first SHORT_ROWS rows only (the CLS-only last layer's gradients)
short_rows = int(os.environ.get("SHORT_ROWS", "512"))
shapes = shapes + [(768, 512), (512, 768)][:nshort]
torch.manual_seed(0)
dh = [torch.randn(rows, n, device=dev).to(torch.bfloat16) for n, k in shapes]
x = [torch.randn(rows, k, device=dev).to(torch.bfloat16) for n, k in shapes]
out = [torch.empty(n, k, device=dev) for n, k in shapes]
ref = [(a.float().t() @ b.float()) for a, b in zip(dh[:2], x[:2])]
nlong = len(shapes) - nshort
floats = sum(n * k for n, k in shapes[:nlong])
probs = (_native.TnProblem * len(shapes))()
for i, (q, a, b, c, (n, k)) in enumerate(zip(probs, dh, x, out, shapes)):
    q.a, q.b, q.c, q.m, q.n, q.lda, q.ldb, q.ldc = _p(a), _p(b), _p(c), n, k, n, k, k
    q.k = short_rows if i >= nlong else 0
ref_short = [(a[:short_rows].float().t() @ b[:short_rows].float()) for a, b in zip(dh[nlong:], x[nlong:])]
for splits in [int(s) for s in os.environ.get("SPLITS", "3,4,5,6,7,8,10").split(",")]:
    ws = torch.empty(splits * floats, device=dev)
    def run():
        _native.call("spv_gemm_tn_batch_part", ctypes.addressof(probs), len(shapes), rows, splits, _p(ws), 0, 0, 1, _stream())
    def red():
        _native.call("spv_gemm_tn_batch_part", ctypes.addressof(probs), len(shapes), rows, splits, _p(ws), 0, 0, 2, _stream())
    run(); red()
    torch.cuda.synchronize()
    err = max(((o - r).abs().max() / r.abs().max()).item() for o, r in zip(out[:2] + out[nlong:], ref + ref_short))
    ts = []
    for fn in (run, red):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 30 * 1e3)
    tf = 2.0 * rows * floats / ts[0] / 1e6
    print(f"short={nshort} wide={os.environ.get('SPV_TNB_WIDE', '1')} splits {splits}: gemm {ts[0]:.1f} us ({tf:.0f} TFLOP/s, {tf / 2500:.3f} of peak)  reduce {ts[1]:.1f} us  max err {err:.2e}", flush=True)
