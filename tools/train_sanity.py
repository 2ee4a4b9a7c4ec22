#!/usr/bin/env python3
"""Longer training sanity run on the Small/CIFAR-100 config (synthetic class-conditional data): loss must fall and stay finite with
every fused kernel, dropout and the fused optimizer path in play.   python tools/train_sanity.py [mixer] [epochs]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-spectre-experiments_amd"))
from spectre_vit.harness import train  # noqa: E402

mixer = sys.argv[1] if len(sys.argv) > 1 else "fft"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = "spectre_vit/configs/spectre_vit_cifar100.py"  # module-style path, as the reference's parse_config takes it
_, hist = train(cfg, mixer=mixer, epochs=epochs, batch_size=256, n_train=4096, n_val=1024, use_amp=True, out_dir="/tmp/spv_sanity",
                log=lambda r: print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}, flush=True))
assert all(h["Loss/Train"] == h["Loss/Train"] for h in hist), "NaN loss"
assert hist[-1]["Loss/Train"] < 0.6 * hist[0]["Loss/Train"], (hist[0], hist[-1])
print("ok: train loss", round(hist[0]["Loss/Train"], 3), "->", round(hist[-1]["Loss/Train"], 3))
