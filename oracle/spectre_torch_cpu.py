"""CPU baseline port of the Spectre-ViT training step on stock ATen ops  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference's CPU path is PyTorch eager on the host (``device = "cuda" if ... else "cpu"``,
spectre_vit/repl/train.py:41): advanced-index gather, ``addmm``, ``native_layer_norm``, ``gelu``,
``adaptive_avg_pool1d``, ``torch.fft`` and autograd.  This file restates that path as ONE functional forward over a
``state_dict`` (no nn.Module mirror, no kernels of this repo), so that ``bench.py``'s ``cpu_baseline`` leg times what the
reference would execute on the same host cores instead of the single-threaded-by-construction numpy oracle
(``oracle/spectre_oracle.py``, which stays the fp64 parity checker).  Only ``tests/`` and ``bench.py``'s
``cpu_baseline`` leg may import it.

Pinning: ``tests/test_oracle_golden.py::test_torch_cpu_port_*`` hold it to the same reference-generated fixtures
(``tests/golden/model_*.npz``: logits, loss, every gradient, post-AdamW weights) as the numpy oracle.

Reference lines restated (paths relative to the reference root):
  SpectralPatchEmbed.forward   spectre_vit/models/spectre/spectre.py:124-156
  MHPermutMix.forward          spectre_vit/models/spectre/layers.py:68-73
  SpectreLinear.forward        spectre_vit/models/spectre/layers.py:95-101
  SpectreEncoderLayer.forward  spectre_vit/models/spectre/spectre.py:65-73
  SpectreEncoder / SpectreViT  spectre_vit/models/spectre/spectre.py:90-103, 194-202
  FNet mixer Re(fft2)          spectre_vit/repl/orthogonal_permut.py:23-28 (the build's "fft" mix_layer)
  train step                   spectre_vit/repl/train.py:216-238 (CrossEntropyLoss mean, AdamW)
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def spectre_linear(x, sd, pre):
    """GELU(LayerNorm(x W^T + b)) + skip(x); skip = identity when in == out else AdaptiveAvgPool1d(out) over channels
    (layers.py:85-101)."""
    w, b = sd[pre + "local_head.0.weight"], sd[pre + "local_head.0.bias"]
    g, be = sd[pre + "local_head.1.weight"], sd[pre + "local_head.1.bias"]
    n, k = w.shape
    h = F.gelu(F.layer_norm(F.linear(x, w, b), (n,), g, be, 1e-5))
    if n == k:
        return h + x
    lead = x.shape[:-1]
    skip = F.adaptive_avg_pool1d(x.reshape(-1, 1, k), n).reshape(*lead, n)
    return h + skip


def permut_mix(x, sd, pre):
    """x.view(B,-1)[:, perms] * signs -> raw reshape (B, N, E*H) -> SpectreLinear (layers.py:68-73)."""
    B, N, E = x.shape
    perms, signs = sd[pre + "perms"], sd[pre + "signs"]
    g = x.reshape(B, N * E)[:, perms] * signs
    return spectre_linear(g.reshape(B, N, -1), sd, pre + "linear.")


def patch_embed(img, sd, P):
    """per-patch Re(rfft2 ortho) * freq weights -> Linear -> cls + pos (spectre.py:124-156)."""
    pre = "embeddings_block."
    B, C, H, W = img.shape
    t = img.reshape(B, C, H // P, P, W // P, P).permute(0, 1, 2, 4, 3, 5).reshape(B, C, -1, P, P)
    f = torch.fft.rfft2(t, norm="ortho").real
    f = f * sd[pre + "freq_weight_h"].view(1, 1, 1, P, 1) * sd[pre + "freq_weight_w"].view(1, 1, 1, 1, P // 2 + 1)
    tok = F.linear(f.permute(0, 2, 1, 3, 4).flatten(2), sd[pre + "proj.weight"], sd[pre + "proj.bias"])
    tok = torch.cat([sd[pre + "cls_token"].expand(B, -1, -1), tok], dim=1)
    return tok + sd[pre + "position_embeddings"]


def forward(img, sd, num_layers, patch_size, mixer="permut"):
    x = patch_embed(img, sd, patch_size)
    src = x
    E = x.shape[-1]
    for i in range(num_layers):
        pre = f"encoder_blocks.layers.{i}."
        if mixer == "permut":
            m = permut_mix(x, sd, pre + "mix_layer.")
        elif mixer == "fft":
            m = torch.fft.fft2(x, dim=(-2, -1)).real
        else:
            raise ValueError(f"cpu baseline port: mixer {mixer!r} not restated (permut / fft only)")
        x = F.layer_norm(m, (E,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], 1e-5) + x            # spectre.py:66
        ff = spectre_linear(spectre_linear(x, sd, pre + "linear1."), sd, pre + "linear3.")                  # spectre.py:70-73
        x = F.layer_norm(x + ff, (E,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], 1e-5)            # spectre.py:67
    x = x + src                                                                                            # spectre.py:103
    cls = x[:, 0, :]
    return spectre_linear(cls, sd, "mlp_head.0."), cls


class TrainState:
    """leaf tensors (requires_grad for float parameters, buffers as they are) + a stock AdamW over them."""

    def __init__(self, state_dict, lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01, dtype=torch.float32):
        self.sd = {}
        params = []
        for k, v in state_dict.items():
            t = torch.as_tensor(v).detach().clone()
            is_param = t.is_floating_point() and not (k.endswith("signs") or k.endswith("local_idx"))
            if t.is_floating_point():
                t = t.to(dtype)
            if is_param:
                t.requires_grad_(True)
                params.append(t)
            self.sd[k] = t
        self.params = params
        self.opt = torch.optim.AdamW(params, lr=lr, betas=betas, weight_decay=weight_decay)

    def step(self, img, labels, num_layers, patch_size, mixer):
        """forward + CrossEntropy + backward + AdamW (train.py:216-238, no AMP on CPU)."""
        self.opt.zero_grad(set_to_none=True)
        logits, cls = forward(img, self.sd, num_layers, patch_size, mixer)
        loss = F.cross_entropy(logits, labels)
        loss.backward()
        self.opt.step()
        return loss.detach(), logits.detach(), cls.detach()
