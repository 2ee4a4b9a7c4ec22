#!/usr/bin/env python3
"""GPU-side time of the three FNet entry points at (B, 65, 512) bf16 through raw C-ABI calls on preallocated buffers
(host cost per call ~3 us, far below the kernels), HIP events around `iters` back-to-back launches."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-spectre-experiments_amd"))
import torch  # noqa: E402

from spectre_vit import _native, hip_ops as H  # noqa: E402


def main():
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 200
    B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 512
    N, D = 65, 512
    dev = torch.device("cuda:0")
    bf = torch.bfloat16
    p = lambda t: t.data_ptr()  # noqa: E731
    st = torch.cuda.current_stream().cuda_stream
    x = torch.randn(B, N, D, device=dev).to(bf)
    y, m, out, dx = (torch.empty_like(x) for _ in range(4))
    gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    mean, rstd = torch.empty(B * N, device=dev), torch.empty(B * N, device=dev)
    dg, db = torch.empty(D, device=dev), torch.empty(D, device=dev)
    part = torch.empty(B * 2 * D, device=dev)
    tw = H._fnet_twiddle(N, dev)
    byt = B * N * D * 2
    cases = [
        ("fnet_mix", 2, lambda: _native.call("spv_fnet_mix", p(x), p(y), 0, p(tw), B, N, D, 1, 0, st)),
        ("fnet_ln_fwd", 3, lambda: _native.call("spv_fnet_ln_fwd", p(x), p(m), p(out), p(gam), p(bet), p(mean), p(rstd), p(tw), B, N, D, 1, st)),
        ("fnet_ln_bwd(+fold)", 3, lambda: _native.call("spv_fnet_ln_bwd", p(x), p(m), p(mean), p(rstd), p(gam), p(dx), p(dg), p(db), p(part), p(tw), B, N, D, 1, st)),
    ]
    for name, passes, fn in cases:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        print(f"{name:20s} ({B},{N},{D}): {us:7.2f} us  {passes * byt / us * 1e-3:7.1f} GB/s = {passes * byt / us * 1e-3 / 80:5.1f} % of 8 TB/s", flush=True)


if __name__ == "__main__":
    main()
