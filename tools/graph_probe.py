#!/usr/bin/env python3
"""Does a HIP graph of the whole training step beat eager launches? (development probe; dropout 0 so no seeds are baked)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-spectre-experiments_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import SMALL  # noqa: E402
from spectre_vit import hip_ops  # noqa: E402
from spectre_vit.models.spectre.spectre import SpectreViT  # noqa: E402


def main():
    mixer = sys.argv[1] if len(sys.argv) > 1 else "fft"
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cfg = dict(SMALL, dropout=0.0)
    model = SpectreViT(**cfg, mixer=mixer).to(dev).train()
    img = torch.randn(512, 3, 32, 32, device=dev)
    labels = torch.randint(0, 100, (512,), device=dev)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True, capturable=True)
    crit = torch.nn.CrossEntropyLoss()

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(img)
        loss = crit(out, labels)
        loss.backward()
        opt.step()
        return loss

    def timeit(fn, n=30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    for _ in range(5):
        step()
    print(f"eager: {timeit(step):.3f} ms/step", flush=True)
    # capture
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    hip_ops._shadows = type(hip_ops._shadows)()  # force the weight casts to be recorded
    g = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(img)
        loss = crit(out, labels)
        loss.backward()
        opt.step()
    l0 = None
    for i in range(5):
        g.replay()
        if i == 0:
            l0 = float(loss.item())
    print(f"graph: {timeit(g.replay):.3f} ms/step  loss {l0:.4f} -> {float(loss.item()):.4f}", flush=True)


if __name__ == "__main__":
    main()
