#!/usr/bin/env python3
"""Host-side profile (cProfile) of the reference script's loop on the mirror package -- the eager, host-bound way to run a step
(bench.py's `as_script` leg).  python tools/script_profile.py [steps]"""
import cProfile
import os
import pstats
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vit-spectre-experiments_amd"), ROOT]
import torch  # noqa: E402

from bench import SMALL  # noqa: E402
from spectre_vit.models.spectre.spectre import SpectreViT  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
torch.manual_seed(42)
model = SpectreViT(**SMALL, mixer="fft").to(dev).train()
g = torch.Generator(device="cpu").manual_seed(1234)
img = torch.randn(512, 3, 32, 32, generator=g).to(dev)
label = torch.randint(0, 100, (512,), generator=g).type(torch.uint8).to(dev)
criterion = torch.nn.CrossEntropyLoss()
optimizer = torch.optim.AdamW(model.parameters(), betas=(0.9, 0.999), lr=1e-3, weight_decay=0.01)
scaler = torch.amp.GradScaler("cuda")


def one():
    with torch.autocast(device_type="cuda", dtype=torch.float16):
        y_pred = model(img)
    loss = criterion(y_pred, label)
    optimizer.zero_grad(set_to_none=True)
    scaler.scale(loss).backward()
    scaler.step(optimizer)
    scaler.update()
    return loss.item()


with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for _ in range(5):
        one()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
