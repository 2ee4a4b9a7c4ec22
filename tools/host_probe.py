#!/usr/bin/env python3
"""Host-side cost of one training step (development tool): issue time vs GPU time, and a cProfile of the issue path."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-spectre-experiments_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import SMALL  # noqa: E402
from spectre_vit.dp import GradReducer  # noqa: E402
from spectre_vit.models.spectre.spectre import SpectreViT  # noqa: E402


def main():
    mixer = sys.argv[1] if len(sys.argv) > 1 else "fft"
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = SpectreViT(**SMALL, mixer=mixer).to(dev).train()
    img = torch.randn(512, 3, 32, 32, device=dev)
    labels = torch.randint(0, 100, (512,), device=dev)
    from spectre_vit.optim import FusedAdamW
    reducer = GradReducer(model, always=True)
    opt = FusedAdamW(model.parameters(), lr=1e-3, static_grads=True)
    from spectre_vit.loss import CrossEntropyLoss
    crit = CrossEntropyLoss()

    def step():
        reducer.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(img)
        loss = crit(out, labels)
        loss.backward()
        reducer.finish()
        opt.step()

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{mixer}: host issue {1e3 * (t1 - t0) / n:.3f} ms/step, drained after {1e3 * (t2 - t1):.3f} ms more; total {1e3 * (t2 - t0) / n:.3f} ms/step")
    # forward / backward / optimizer split of the host time (GPU drained in between so that nothing blocks)
    parts = {"fwd": 0.0, "bwd": 0.0, "opt": 0.0}
    for _ in range(10):
        torch.cuda.synchronize(); a = time.perf_counter()
        reducer.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(img)
        loss = crit(out, labels)
        b = time.perf_counter(); torch.cuda.synchronize(); b2 = time.perf_counter()
        loss.backward(); reducer.finish()
        c = time.perf_counter(); torch.cuda.synchronize(); c2 = time.perf_counter()
        opt.step()
        d = time.perf_counter()
        parts["fwd"] += b - a; parts["bwd"] += c - b2; parts["opt"] += d - c2
    print({k: f"{1e2 * v:.3f} ms" for k, v in parts.items()})
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
