"""Generate the golden fixtures in tests/golden/*.npz by EXECUTING the reference's own modules.

Run in the build container only (the reference is mounted read-only at /root/reference and never
travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference modules are imported (not copied), instantiated under a fixed seed, switched to
float64 (weights are drawn in fp32 first, so they stay exactly fp32-representable and are stored as
fp32), and run forward/backward on seeded inputs with dropout = 0.  Only arrays are written:
weights/buffers (state_dict), inputs, outputs, gradients, post-AdamW weights.
"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get("SPV_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

from spectre_vit.models.spectre.layers import MHPermutMix, SpectreLinear  # noqa: E402
from spectre_vit.models.spectre.spectre import (  # noqa: E402
    SpectralPatchEmbed, SpectreEncoderLayer, SpectreViT)
from spectre_vit.models.vit.vit import ViT  # noqa: E402
from spectre_vit.modules.patch_embeddings import PatchEmbedding  # noqa: E402
from spectre_vit.modules.spectre import FFT  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def npy(t):
    return t.detach().cpu().numpy()


def sd_np(mod, prefix="sd."):
    out = {}
    for k, v in mod.state_dict().items():
        a = npy(v)
        if a.dtype == np.float64:
            a = a.astype(np.float32)  # exact: values were drawn in fp32
        out[prefix + k] = a
    return out


def grads_np(mod, prefix="grad."):
    return {prefix + k: npy(p.grad) for k, p in mod.named_parameters()}


def gen(seed):
    return torch.Generator().manual_seed(seed)


def op_fixtures():
    d = {}
    # ---- SpectreLinear, every pooling regime (layers.py:76-101)
    for name, (cin, cout, shape) in {
        "sl_equal": (16, 16, (3, 5, 16)),
        "sl_down_exact": (32, 8, (2, 4, 32)),
        "sl_down_overlap": (12, 8, (2, 3, 12)),
        "sl_up": (8, 12, (2, 3, 8)),
        "sl_head2d": (16, 5, (4, 16)),
    }.items():
        torch.manual_seed(100 + cin + cout)
        m = SpectreLinear(cin, cout).double()
        # non-trivial LN affine so gamma/beta gradients are exercised
        with torch.no_grad():
            m.local_head[1].weight.copy_((torch.rand(cout, generator=gen(1)) + 0.5).float().double())
            m.local_head[1].bias.copy_((torch.randn(cout, generator=gen(2)) * 0.1).float().double())
        x = torch.randn(*shape, generator=gen(3)).double().requires_grad_(True)
        dy = torch.randn(*shape[:-1], cout, generator=gen(4)).double()
        y = m(x)
        y.backward(dy)
        d.update({f"{name}.x": npy(x), f"{name}.dy": npy(dy), f"{name}.y": npy(y), f"{name}.dx": npy(x.grad)})
        d.update(sd_np(m, f"{name}.sd."))
        d.update(grads_np(m, f"{name}.grad."))

    # ---- MHPermutMix (layers.py:53-73)
    torch.manual_seed(7)
    E, N, H = 8, 5, 3
    m = MHPermutMix(E, N, H, E).double()
    x = torch.randn(4, N, E, generator=gen(5)).double().requires_grad_(True)
    dy = torch.randn(4, N, E, generator=gen(6)).double()
    y = m(x)
    y.backward(dy)
    xg = x.detach().view(4, -1)[:, m.perms] * m.signs
    d.update({"permut.x": npy(x), "permut.dy": npy(dy), "permut.y": npy(y), "permut.dx": npy(x.grad),
              "permut.gathered": npy(xg.view(4, N, E * H))})
    d.update(sd_np(m, "permut.sd."))
    d.update(grads_np(m, "permut.grad."))

    # ---- SpectralPatchEmbed (spectre.py:106-156)
    torch.manual_seed(11)
    m = SpectralPatchEmbed(16, 4, 4, 0.0, 3).double()
    with torch.no_grad():
        m.freq_weight_h.copy_((torch.rand(4, generator=gen(7)) + 0.5).float().double())
        m.freq_weight_w.copy_((torch.rand(3, generator=gen(8)) + 0.5).float().double())
    x = torch.randn(3, 3, 8, 8, generator=gen(9)).double()
    dy = torch.randn(3, 5, 16, generator=gen(10)).double()
    y = m(x)
    y.backward(dy)
    d.update({"spe.x": npy(x), "spe.dy": npy(dy), "spe.y": npy(y)})
    d.update(sd_np(m, "spe.sd."))
    d.update(grads_np(m, "spe.grad."))

    # ---- PatchEmbedding conv (patch_embeddings.py:4-43)
    torch.manual_seed(12)
    m = PatchEmbedding(16, 4, 4, 0.0, 3).double()
    y = m(x)
    y.backward(dy)
    d.update({"pe.x": npy(x), "pe.dy": npy(dy), "pe.y": npy(y)})
    d.update(sd_np(m, "pe.sd."))
    d.update(grads_np(m, "pe.grad."))

    # ---- FFT module (modules/spectre.py:9-14) and FNet Re(fft2) (orthogonal_permut.py:23-28)
    x = torch.randn(2, 5, 16, generator=gen(13)).double().requires_grad_(True)
    y = FFT()(x)
    dy = torch.randn(*y.shape, generator=gen(14)).double()
    y.backward(dy)
    d.update({"fftmod.x": npy(x), "fftmod.y": npy(y), "fftmod.dy": npy(dy), "fftmod.dx": npy(x.grad)})
    x = torch.randn(2, 5, 16, generator=gen(15)).double().requires_grad_(True)
    y = torch.fft.fft2(x, dim=(-2, -1)).real
    dy = torch.randn(*y.shape, generator=gen(16)).double()
    y.backward(dy)
    d.update({"fnet.x": npy(x), "fnet.y": npy(y), "fnet.dy": npy(dy), "fnet.dx": npy(x.grad)})
    x = torch.randn(2, 65, 32, generator=gen(17)).double()  # the real token count (65 = 5*13)
    d.update({"fnet65.x": npy(x), "fnet65.y": npy(torch.fft.fft2(x, dim=(-2, -1)).real)})

    # ---- SpectreEncoderLayer (spectre.py:29-73)
    torch.manual_seed(21)
    m = SpectreEncoderLayer(seq_length=5, d_model=16, nhead=2, dim_feedforward=24, dropout=0.0,
                            activation="gelu").double()
    x = torch.randn(3, 5, 16, generator=gen(18)).double().requires_grad_(True)
    dy = torch.randn(3, 5, 16, generator=gen(19)).double()
    y = m(x)
    y.backward(dy)
    d.update({"layer.x": npy(x), "layer.dy": npy(dy), "layer.y": npy(y), "layer.dx": npy(x.grad)})
    d.update(sd_np(m, "layer.sd."))
    d.update(grads_np(m, "layer.grad."))

    # ---- stock TransformerEncoderLayer as the baseline ViT builds it (vit.py:30-36): batch_first=False
    torch.manual_seed(22)
    m = torch.nn.TransformerEncoderLayer(d_model=16, nhead=4, dim_feedforward=24, dropout=0.0,
                                         activation="gelu").double()
    x = torch.randn(3, 5, 16, generator=gen(20)).double().requires_grad_(True)
    dy = torch.randn(3, 5, 16, generator=gen(21)).double()
    y = m(x)
    y.backward(dy)
    d.update({"tel.x": npy(x), "tel.dy": npy(dy), "tel.y": npy(y), "tel.dx": npy(x.grad)})
    d.update(sd_np(m, "tel.sd."))
    d.update(grads_np(m, "tel.grad."))

    # ---- baseline ViT forward (vit.py:7-51), tiny
    torch.manual_seed(23)
    m = ViT(img_size=8, patch_size=4, in_channels=3, num_classes=7, embed_dim=16, num_encoders=2,
            num_heads=4, hidden_dim=24, dropout=0.0).double().eval()
    x = torch.randn(3, 3, 8, 8, generator=gen(22)).double()
    with torch.no_grad():
        logits, cls = m(x, return_features=True)
    d.update({"vit.x": npy(x), "vit.logits": npy(logits), "vit.cls": npy(cls)})
    d.update(sd_np(m, "vit.sd."))
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **d)
    print("ops.npz:", len(d), "arrays")


def model_fixture(name, cfg, batch, seed):
    """Whole-model: logits, cls, CE loss, all grads, weights after one AdamW step
    (train.py:196-201,216-238; AdamW lr 1e-3, betas (0.9,0.999), wd 0.01)."""
    torch.manual_seed(seed)
    m = SpectreViT(**cfg).double()
    g = gen(1234)
    img = torch.randn(batch, cfg["in_channels"], cfg["img_size"], cfg["img_size"], generator=g).double()
    labels = torch.randint(0, cfg["num_classes"], (batch,), generator=g)
    d = {"img": npy(img).astype(np.float32), "labels": npy(labels)}
    img = torch.from_numpy(d["img"]).double()  # inputs exactly fp32-representable
    d.update(sd_np(m))
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01)
    logits, cls = m(img, return_features=True)
    loss = torch.nn.CrossEntropyLoss()(logits, labels)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    d.update({"logits": npy(logits), "cls": npy(cls), "loss": npy(loss)})
    d.update(grads_np(m))
    opt.step()
    d.update({"after." + k: npy(p) for k, p in m.named_parameters()})
    with torch.no_grad():
        d["logits_after"] = npy(m(img))
    d["cfg"] = np.array(repr(cfg))
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **d)
    print(f"{name}.npz:", len(d), "arrays, loss", float(loss))


def config_fixtures():
    """values of every reference config module as parse_config would return them (spectre_vit/configs/*.py; the reference's
    parse_config itself needs Python 3.13 -- SimpleNamespace(mapping) -- so its two steps are applied by hand: module_to_dict,
    then `mod |= base_mod` when the literal key __base__ is present, parser.py:22-25)."""
    import importlib
    import json
    from spectre_vit.configs.parser import module_to_dict
    out = {}
    cfg_dir = os.path.join(REF, "spectre_vit", "configs")
    for f in sorted(os.listdir(cfg_dir)):
        if not f.endswith(".py") or f in ("parser.py", "__init__.py"):
            continue
        name = f[:-3]
        mod = module_to_dict(importlib.import_module(f"spectre_vit.configs.{name}"))
        if "__base__" in mod:
            mod |= module_to_dict(importlib.import_module("spectre_vit.configs." + mod["__base__"].replace(".py", "")))
        out[name] = {k: (list(v) if isinstance(v, tuple) else v) for k, v in mod.items()}
    json.dump(out, open(os.path.join(OUT, "configs.json"), "w"), indent=1, sort_keys=True)
    print("configs.json:", sorted(out))


def hadamard_fixtures():
    """fwht (hadamar.py:12-32), fwht_fast (:58-80), hadamard_transform (:83-112), LearnableHadamard (:115-141)."""
    from spectre_vit.models.spectre.hadamar import LearnableHadamard, fwht, fwht_fast, hadamard_transform
    d = {}
    for n in (2, 8, 64, 512):
        x = torch.randn(3, 5, n, generator=gen(100 + n)).double().requires_grad_(True)
        dy = torch.randn(3, 5, n, generator=gen(200 + n)).double()
        for name, fn in (("fwht", lambda t: fwht(t)), ("fwht_raw", lambda t: fwht(t, normalize=False)), ("fwht_fast", fwht_fast)):
            x.grad = None
            y = fn(x)
            y.backward(dy)
            d.update({f"{name}.{n}.x": npy(x), f"{name}.{n}.y": npy(y), f"{name}.{n}.dy": npy(dy), f"{name}.{n}.dx": npy(x.grad)})
        x2 = torch.randn(4, n, generator=gen(300 + n)).double()
        d.update({f"hadamard_transform.{n}.x": npy(x2), f"hadamard_transform.{n}.y": npy(hadamard_transform(x2))})
    # fwht along a non-last axis (token axis of a (B, N, D) tensor)
    x = torch.randn(2, 16, 6, generator=gen(41)).double()
    d.update({"fwht_dim1.x": npy(x), "fwht_dim1.y": npy(fwht(x, dim=1))})
    for dim, blocks in ((48, 2), (64, 1), (100, 3)):
        torch.manual_seed(7)
        m = LearnableHadamard(dim, num_blocks=blocks).double()
        x = torch.randn(2, 7, dim, generator=gen(400 + dim)).double().requires_grad_(True)
        dy = torch.randn(2, 7, dim, generator=gen(500 + dim)).double()
        y = m(x)
        y.backward(dy)
        key = f"lh.{dim}.{blocks}"
        d.update({key + ".x": npy(x), key + ".y": npy(y), key + ".dy": npy(dy), key + ".dx": npy(x.grad)})
        d[key + ".param_grads_none"] = np.array(all(p.grad is None for p in m.params))
        d[key + ".state_keys"] = np.array(",".join(m.state_dict().keys()))
    np.savez_compressed(os.path.join(OUT, "hadamard.npz"), **d)
    print("hadamard.npz:", len(d), "arrays")


if __name__ == "__main__":
    config_fixtures()
    hadamard_fixtures()
    if "--aux-only" in sys.argv:
        sys.exit(0)
    op_fixtures()
    # Tiny/MNIST: configs/spectre_vit_mnist.py:3-19 (img 28, P 4, C 3, E 48, H 8, F 256, L 4, 100 classes)
    model_fixture("model_tiny_mnist", dict(img_size=28, patch_size=4, in_channels=3, num_classes=100,
                                           embed_dim=48, num_encoders=4, num_heads=8, hidden_dim=256,
                                           dropout=0.0, activation="gelu"), batch=4, seed=42)
    # cut-down Small/CIFAR: same structure as configs/spectre_vit_cifar100.py:3-20 at E 64, H 4, F 96, L 2, img 16
    model_fixture("model_small_cut", dict(img_size=16, patch_size=4, in_channels=3, num_classes=100,
                                          embed_dim=64, num_encoders=2, num_heads=4, hidden_dim=96,
                                          dropout=0.0, activation="gelu"), batch=6, seed=42)
