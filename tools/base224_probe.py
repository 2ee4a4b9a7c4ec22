"""BASELINE config 5, student side, for the record: Spectre-ViT-Base (E 768, 12 layers, 12 heads, F 3072, HEAD mixer) at 224 / 16 on one
GPU -- train step (fwd + CE + bwd + FusedAdamW) in bf16, eager and replayed.  Not a bench line (bench.py measures config 2)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-spectre-experiments_amd"))
import torch  # noqa: E402

from spectre_vit.graph import GraphedTrainStep  # noqa: E402
from spectre_vit.loss import CrossEntropyLoss  # noqa: E402
from spectre_vit.models.spectre.spectre import SpectreViT  # noqa: E402
from spectre_vit.optim import FusedAdamW  # noqa: E402


def main():
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    mixer = sys.argv[2] if len(sys.argv) > 2 else "permut"
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = SpectreViT(img_size=224, patch_size=16, in_channels=3, num_classes=100, embed_dim=768, num_encoders=12, num_heads=12,
                   hidden_dim=3072, dropout=0.1, mixer=mixer).to(dev).train()
    img = torch.randn(bs, 3, 224, 224, device=dev)
    lab = torch.randint(0, 100, (bs,), device=dev)
    opt = FusedAdamW(m.parameters(), lr=1e-4, weight_decay=0.01, capturable=True, static_grads=True)
    step = GraphedTrainStep(m, opt, CrossEntropyLoss(), img, lab, autocast_dtype=torch.bfloat16)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"Base/224 {mixer} bs {bs}: {dt * 1e3:.2f} ms/step, {bs / dt:.0f} img/s, loss {loss.item():.3f}, "
          f"max mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    step.close()


if __name__ == "__main__":
    main()
