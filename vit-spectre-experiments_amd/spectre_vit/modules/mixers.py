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
    """J-level Haar DWT along dim ('dwt_embed') or tokens ('dwt_token'); output bands [a_J | d_J | ... | d_1] (pywt.wavedec's order)
    in place of the transformed axis, pairs a = (x0 + x1) / sqrt2, d = (x0 - x1) / sqrt2 (PyWavelets' documented 'haar').
    PARITY UNPINNED against the reference (no model code there, only repl/dwt_experiments.py:56).

    mode: what happens to the unpaired last element of an odd length (65 tokens).  "passthrough" (default): copied into the
    approximation band -- orthonormal.  "zero": pywt's mode="zero", the convention of the reference's call: paired with a zero, so
    a_last = d_last = x_last / sqrt2; pywt would return 33 + 33 = 66 coefficients for 65 tokens -- the mixer keeps the 33
    approximation and the first 32 detail coefficients (the dropped one is a copy of a_last).  Even lengths: the modes coincide."""

    def __init__(self, axis: str = "embed", levels: int = 1, mode: str = "passthrough"):
        super().__init__()
        assert axis in ("embed", "token")
        if mode not in ("passthrough", "zero"):
            raise ValueError(f"HaarDWTMixer mode must be 'passthrough' or 'zero', got {mode!r}")
        self.axis = axis
        self.levels = levels
        self.mode = mode

    def forward(self, x):
        x = hip_ops.cast(x, hip_ops.compute_dtype(x))
        return hip_ops.HaarDWTFn.apply(x, 2 if self.axis == "embed" else 1, self.levels, self.mode == "zero")
