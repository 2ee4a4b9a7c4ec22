"""diagnostic: gradient of a batch vs the mean of its two halves' gradients, through GraphedDPStep (lr 0), per parameter"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vit-spectre-experiments_amd"), ROOT]
import torch
from spectre_vit.graph import GraphedDPStep
from spectre_vit.loss import CrossEntropyLoss
from spectre_vit.models.spectre.spectre import SpectreViT
from spectre_vit.optim import FusedAdamW
from spectre_vit import hip_ops

CFG = dict(img_size=32, patch_size=4, in_channels=3, num_classes=100, embed_dim=512, num_encoders=2, num_heads=16, hidden_dim=768,
           dropout=0.0, activation="gelu")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(9)
x, y = torch.randn(128, 3, 32, 32, generator=g).to(dev), torch.randint(0, 100, (128,), generator=g).to(dev)


def grads(xs, ys, mode):
    torch.manual_seed(200)
    m = SpectreViT(**CFG, mixer="fft").to(dev).train()
    if mode == "graph":
        opt = FusedAdamW(m.parameters(), lr=LR, eps=1.0, weight_decay=0.01, capturable=True, static_grads=True)
        st = GraphedDPStep(m, opt, CrossEntropyLoss(), xs, ys, autocast_dtype=torch.bfloat16, warmup=1)
        st()
        torch.cuda.synchronize()
        out = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        st.close()
        return out
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = CrossEntropyLoss()(m(xs), ys)
    loss.backward()
    torch.cuda.synchronize()
    return {k: p.grad.detach().clone() for k, p in m.named_parameters()}


LR = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
for mode in ("eager", "graph"):
    full = grads(x, y, mode)
    h0, h1 = grads(x[:64], y[:64], mode), grads(x[64:], y[64:], mode)
    print("==", mode)
    for k in full:
        avg = 0.5 * (h0[k] + h1[k])
        err = (avg - full[k]).abs().max().item() / (full[k].abs().max().item() + 1e-30)
        if err > 2e-5:
            print(f"  {k}: {err:.2e}")
    if mode == "graph":   # what a 2-rank job does: the warm-up update uses the AVERAGED gradient; emulate with two steps
        pass
fe, fg = grads(x, y, "eager"), grads(x, y, "graph")
print("== eager vs graph, full batch")
for k in fe:
    err = (fe[k] - fg[k]).abs().max().item() / (fe[k].abs().max().item() + 1e-30)
    if err > 2e-5:
        print(f"  {k}: {err:.2e}")
