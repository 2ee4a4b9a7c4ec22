import sys, time
sys.path.insert(0, "vit-spectre-experiments_amd")
import torch
from spectre_vit.models.spectre.spectre import SpectreViT
from spectre_vit.distillation import SyntheticTeacher, distillation_loss
dev = torch.device("cuda:0")
torch.manual_seed(0)
for mixer in ("permut", "fft"):
    m = SpectreViT(img_size=224, patch_size=16, in_channels=3, num_classes=100, mixer=mixer).to(dev)  # defaults: E768 L12 H12 F3072 dropout 0.1
    print(mixer, "params", sum(p.numel() for p in m.parameters()))
    t = SyntheticTeacher(100, 384, 3).to(dev)
    x = torch.randn(8, 3, 224, 224, device=dev)
    y = torch.randint(0, 100, (8,), device=dev)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, fused=True)
    for it in range(3):
        t0 = time.perf_counter()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            s_logits, s_feat = m(x, return_features=True)
        with torch.no_grad():
            t_logits, _ = t(x, return_features=True)
        loss, soft, ce = distillation_loss(s_logits, t_logits, y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        print(mixer, it, float(loss), f"{(time.perf_counter()-t0)*1e3:.1f} ms", all(torch.isfinite(p.grad).all().item() for p in m.parameters()))
    del m, opt
