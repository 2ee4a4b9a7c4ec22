"""FFT module -- mirror of reference spectre_vit/modules/spectre.py:5-14: rfft(x, dim=-1).real."""
import torch.nn as nn

from spectre_vit import hip_ops


class FFT(nn.Module):
    def __init__(self) -> None:
        super().__init__()

    def forward(self, x):
        x = hip_ops.cast(x, hip_ops.compute_dtype(x))
        return hip_ops.RfftRealFn.apply(x)
