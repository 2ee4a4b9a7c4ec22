"""Parameter-free spectral token mixers with the (B, N, D) -> (B, N, D) ``mix_layer`` contract.

The reference names them in SpectreEncoderLayer's docstring (spectre_vit/models/spectre/spectre.py:30-36:
fft_bare, dwt_embed, dwt_token) but wires none of them at HEAD; BASELINE.json's configs 2-4 ask for them.
"""
import torch.nn as nn

from spectre_vit import hip_ops


class FNetMixer(nn.Module):
    """y = Re(fft2(x)) over (tokens, dim), un-normalised (reference spectre_branch.py:79, orthogonal_permut.py:23-28)."""

    def forward(self, x):
        return hip_ops.FNetMixFn.apply(hip_ops.cast(x, hip_ops.compute_dtype(x)))


class HaarDWTMixer(nn.Module):
    """J-level orthonormal Haar DWT along dim ('dwt_embed') or tokens ('dwt_token'); output bands
    [a_J | d_J | ... | d_1] in place of the transformed axis.  PARITY UNPINNED (no reference model code)."""

    def __init__(self, axis: str = "embed", levels: int = 1):
        super().__init__()
        assert axis in ("embed", "token")
        self.axis = axis
        self.levels = levels

    def forward(self, x):
        x = hip_ops.cast(x, hip_ops.compute_dtype(x))
        return hip_ops.HaarDWTFn.apply(x, 2 if self.axis == "embed" else 1, self.levels)
