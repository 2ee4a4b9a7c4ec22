#!/usr/bin/env python3
"""FNet mixer kernels at the benchmark shape (512, 65, 512) bf16: correctness against numpy's fft2 and per-launch time
(development tool; run on the GPU box).

    python tools/fnet_check.py [--iters N] [--batch B]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-spectre-experiments_amd"))
import torch  # noqa: E402

from spectre_vit import hip_ops as H  # noqa: E402


def rel_l2(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def timeit(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 50
    B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 512
    dev = torch.device("cuda:0")
    ok = True
    for N in (65, 64, 50, 17, 2):
        g = torch.Generator().manual_seed(N)
        x = torch.randn(3, N, 512, generator=g).to(torch.bfloat16)
        ref = np.fft.fft2(x.float().numpy().astype(np.float64), axes=(-2, -1)).real
        y = H._fnet_raw(x.to(dev)).float().cpu().numpy().astype(np.float64)
        e = rel_l2(y, ref)
        m = float(np.abs(y - ref).max() / np.abs(ref).max())
        print(f"fnet_mix N={N}: rel-L2 {e:.3e}  max-norm {m:.3e}", flush=True)
        ok &= e < 6e-3
    # fused forward / backward at N = 65 against the composition of the plain ops in float64
    N = 65
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, N, 512, generator=g).to(torch.bfloat16)
    gam = (1.0 + 0.1 * torch.randn(512, generator=g)).float()
    bet = (0.1 * torch.randn(512, generator=g)).float()
    xin = x.to(dev).requires_grad_(True)
    gp, bp = gam.to(dev).requires_grad_(True), bet.to(dev).requires_grad_(True)
    out = H.FNetResidualFn.apply(xin, gp, bp)
    dout = torch.randn(4, N, 512, generator=g).to(torch.bfloat16)
    out.backward(dout.to(dev))
    x64 = x.double().requires_grad_(True)
    g64, b64 = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    m64 = torch.fft.fft2(x64, dim=(-2, -1)).real
    o64 = torch.nn.functional.layer_norm(m64, (512,), g64, b64, 1e-5) + x64
    o64.backward(dout.double())
    for name, got, want in (("ln_fwd out", out, o64), ("ln_bwd dx", xin.grad, x64.grad), ("dgamma", gp.grad, g64.grad), ("dbeta", bp.grad, b64.grad)):
        e = rel_l2(got.detach().float().cpu().numpy().astype(np.float64), want.detach().numpy())
        print(f"{name}: rel-L2 {e:.3e}", flush=True)
        ok &= e < 1.2e-2
    # timing
    xb = torch.randn(B, 65, 512, device=dev).to(torch.bfloat16)
    gw, gb = torch.ones(512, device=dev, requires_grad=True), torch.zeros(512, device=dev, requires_grad=True)
    byt = B * 65 * 512 * 2
    us = timeit(lambda: H._fnet_raw(xb), iters)
    print(f"fnet_mix      ({B},65,512): {us:7.2f} us  {2 * byt / us * 1e-3:7.1f} GB/s = {2 * byt / us * 1e-3 / 80:5.1f} % of 8 TB/s")
    xr = xb.clone().requires_grad_(True)
    outs = []

    def fwd():
        outs.clear()
        outs.append(H.FNetResidualFn.apply(xr, gw, gb))

    us = timeit(fwd, iters)
    print(f"fnet_ln_fwd   ({B},65,512): {us:7.2f} us  {3 * byt / us * 1e-3:7.1f} GB/s = {3 * byt / us * 1e-3 / 80:5.1f} % of 8 TB/s  (host-inclusive loop)")
    d = torch.randn_like(xb)
    fwd()

    def bwd():
        outs[0].backward(d, retain_graph=True)

    us = timeit(bwd, iters)
    print(f"fnet_ln_bwd   ({B},65,512): {us:7.2f} us  {3 * byt / us * 1e-3:7.1f} GB/s = {3 * byt / us * 1e-3 / 80:5.1f} % of 8 TB/s  (incl. fold + autograd glue)")
    print("OK" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
