#!/usr/bin/env python3
"""Inference latency / throughput of the mirrored models -- counterpart of the reference's spectre_vit/repl/test.py:30-62
(which times forward passes without synchronising the device; here every sample is bracketed by HIP events).

    python tools/infer_bench.py [--mixer fft|permut|dwt_embed|dwt_token] [--model spectre|vit] [--batches 1,8,64,512]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-spectre-experiments_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import SMALL  # noqa: E402
from spectre_vit.models.spectre.spectre import SpectreViT  # noqa: E402
from spectre_vit.models.vit.vit import ViT  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mixer", default="fft")
    ap.add_argument("--model", default="spectre", choices=["spectre", "vit"])
    ap.add_argument("--batches", default="1,8,64,512")
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(42)
    model = (SpectreViT(**SMALL, mixer=args.mixer) if args.model == "spectre" else ViT(**SMALL)).to(dev).eval()
    out = []
    for bs in [int(b) for b in args.batches.split(",")]:
        x = torch.randn(bs, 3, 32, 32, device=dev)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            for _ in range(5):
                model(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                model(x)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        out.append({"batch": bs, "latency_ms": round(ms, 4), "images_per_s": round(bs / ms * 1e3, 1)})
    print(json.dumps({"model": args.model, "mixer": args.mixer if args.model == "spectre" else None, "dtype": "bf16", "results": out}))


if __name__ == "__main__":
    main()
