"""Distillation helpers -- counterpart of reference spectre_vit/distillation.py:5-43 plus the loss of
spectre_vit/repl/train.py:334-348.

``DinoClassifier`` wraps a frozen backbone exposing ``forward_features(x)["x_norm_clstoken"]`` (the DINOv3 contract);
the real teacher weights are not available offline, so ``SyntheticTeacher`` provides the same output
contract with fixed random projections for benchmarks and tests.  The teacher runs under ``no_grad`` in stock PyTorch --
it is outside the accelerated path.
"""
import torch
import torch.nn as nn


class DinoClassifier(nn.Module):
    """backbone.forward_features -> CLS feature -> Linear decoder; ``forward(x, return_features)`` like the student."""

    def __init__(self, backbone, num_classes, embed_dim=384):
        super().__init__()
        self.backbone = backbone
        self.decoder = nn.Sequential(nn.Linear(embed_dim, num_classes))

    def forward(self, x, return_features=False):
        feats = self.backbone.forward_features(x)["x_norm_clstoken"]  # [B, C]
        logits = self.decoder(feats)
        return (logits, feats) if return_features else logits


class DistillationDatasetCls(torch.utils.data.Dataset):
    """Two views of one sample: {"img_teacher", "img_model", "label"} (reference distillation.py:25-43)."""

    def __init__(self, samples, teacher_tf, model_tf):
        self.samples = samples
        self.teacher_tf = teacher_tf
        self.model_tf = model_tf

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, idx):
        img, label = self.samples[idx]
        return {"img_teacher": self.teacher_tf(img), "img_model": self.model_tf(img), "label": label}


class _SyntheticBackbone(nn.Module):
    """fixed random features with the DINOv3 ``forward_features`` dictionary contract"""

    def __init__(self, in_channels, embed_dim, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.register_buffer("proj", torch.randn(in_channels * 64, embed_dim, generator=g) / 8.0)

    def forward_features(self, x):
        pooled = nn.functional.adaptive_avg_pool2d(x, 8).flatten(1)  # [B, C*64]
        return {"x_norm_clstoken": nn.functional.layer_norm(pooled @ self.proj, (self.proj.shape[1],))}


def SyntheticTeacher(num_classes=100, embed_dim=384, in_channels=3, seed=0):
    """stand-in for the frozen DINOv3-S teacher: (B, num_classes) logits and (B, embed_dim) features"""
    t = DinoClassifier(_SyntheticBackbone(in_channels, embed_dim, seed), num_classes, embed_dim)
    for p in t.parameters():
        p.requires_grad_(False)
    return t.eval()


def distillation_loss(student_logits, teacher_logits, labels, T=2.0, soft_target_loss_weight=0.25, ce_loss_weight=0.75):
    """0.25 * T^2 * sum(p_t (log p_t - log p_s)) / B + 0.75 * CE  (reference train.py:300-302, 334-348).
    Returns (loss, soft_targets_loss, ce_loss)."""
    soft_targets = nn.functional.softmax(teacher_logits / T, dim=-1)
    soft_prob = nn.functional.log_softmax(student_logits / T, dim=-1)
    soft = torch.sum(soft_targets * (soft_targets.log() - soft_prob)) / soft_prob.size(0) * (T ** 2)
    if student_logits.is_cuda and student_logits.dtype == torch.float32:
        from . import hip_ops
        ce = hip_ops.cross_entropy(student_logits, labels)  # one launch each way (csrc/spv_head.hip)
    else:
        ce = nn.functional.cross_entropy(student_logits, labels)
    return soft_target_loss_weight * soft + ce_loss_weight * ce, soft, ce
