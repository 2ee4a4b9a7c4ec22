"""Leaf layers of Spectre-ViT -- mirror of reference spectre_vit/models/spectre/layers.py.

Same classes, constructor arguments, parameter / buffer names (state_dict ABI); forward passes run on
libspv_hip.so through spectre_vit.hip_ops.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from spectre_vit import hip_ops
from spectre_vit.modules.spectre import FFT  # noqa: F401  (the reference imports it here too: layers.py:7)


class SpectreLinear(nn.Module):
    """out = GELU(LayerNorm(Linear(x))) + AdaptiveAvgPool1d(out)(x)   (reference layers.py:76-101).

    ``local_head`` keeps the reference's nn.Sequential(Linear, LayerNorm, GELU) so the state_dict keys
    ``local_head.0.weight/bias``, ``local_head.1.weight/bias`` and the ``local_idx`` buffer are identical.
    ``drop_p`` (not in the reference signature; set by the encoder layer) fuses the nn.Dropout that follows
    linear1/linear3 (reference spectre.py:70-73) into the same kernel.
    """

    def __init__(self, in_channels, out_channels, tokens=None):
        super().__init__()
        self.out_channels = out_channels
        self.in_channels = in_channels
        self.sparsity = 1
        local_idx = torch.arange(0, in_channels, self.sparsity)
        self.register_buffer("local_idx", local_idx)
        local_channels = int(math.ceil(in_channels / self.sparsity))
        self.local_head = nn.Sequential(
            nn.Linear(local_channels, out_channels),
            nn.LayerNorm(out_channels),
            nn.GELU(),
        )
        self.drop_p = 0.0
        self.out_fp32 = False

    def forward(self, x, dim=(-1)):
        lin, ln = self.local_head[0], self.local_head[1]
        dt = hip_ops.compute_dtype(x)
        mult = 8 if dt == torch.bfloat16 else 4
        if dt == torch.bfloat16 and (self.in_channels % mult or self.out_channels % mult):
            dt = torch.float32  # e.g. the 100-class head: rows are not 16-byte multiples in bf16
        x = hip_ops.cast(x, dt)
        p = self.drop_p if self.training else 0.0
        return hip_ops.spectre_linear(x, lin.weight, lin.bias, ln.weight, ln.bias, p, self.out_fp32)


class MHPermutMix(nn.Module):
    """Signed multi-head permutation token mixer (reference layers.py:53-73):
    g = x.view(B,-1)[:, perms] * signs -> raw view (B, tokens, embed*heads) -> SpectreLinear."""

    def __init__(self, embed_dim: int, token_dim: int, num_heads: int, out_channels: int):
        super().__init__()
        d = embed_dim * token_dim
        self.num_heads = num_heads
        self.token_dim = token_dim
        self.embed_dim = embed_dim
        self.concat_dim = self.embed_dim * self.num_heads
        signs = torch.randint(0, 2, (num_heads, d), dtype=torch.float32)
        signs = signs * 2 - 1
        self.register_buffer("signs", signs.unsqueeze(0))
        perms = torch.stack([torch.randperm(d) for _ in range(num_heads)])
        self.register_buffer("perms", perms)
        self.linear = SpectreLinear(embed_dim * num_heads, out_channels)
        self._packed = None  # (key, uint32 table) -- derived from the buffers, never part of the state_dict

    def __deepcopy__(self, memo):  # _get_clones deep-copies the layer (reference spectre.py:86)
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            setattr(new, k, None if k == "_packed" else copy.deepcopy(v, memo))
        return new

    def _table(self):
        key = (self.perms.data_ptr(), self.perms._version, self.signs.data_ptr(), self.signs._version)
        if self._packed is None or self._packed[0] != key:
            self._packed = (key, hip_ops.permut_pack(self.perms, self.signs.reshape(self.num_heads, -1)))
        return self._packed[1]

    def forward(self, x):
        B = x.shape[0]
        x = hip_ops.cast(x, hip_ops.compute_dtype(x))
        lin, ln = self.linear.local_head[0], self.linear.local_head[1]
        mult = 8 if x.dtype == torch.bfloat16 else 4
        if lin.in_features % mult == 0 and lin.out_features % mult == 0:
            # gather + Linear + LayerNorm/GELU/avg-pool skip as one autograd node
            return hip_ops.PermutMixFn.apply(x, self._table(), self.num_heads, lin.weight, lin.bias, ln.weight, ln.bias)
        g = hip_ops.PermutGatherFn.apply(x, self._table(), self.num_heads)
        return self.linear(g.view(B, self.token_dim, self.concat_dim))

    def forward_cls(self, x):
        """(row 0 of forward(x) as (B, out_channels), x[:, 0, :]).  Token row t of the gathered matrix is the chunk [t * concat_dim,
        (t + 1) * concat_dim) of the flattened (heads, d) gather (the raw view of reference layers.py:72), so row 0 needs the first
        concat_dim entries of the forward table: 8192 of the 532 480 gathered values per image, one GEMM row instead of 65.  For the
        last layer of a stack whose consumer reads the CLS row only (hip_ops.LAST_LAYER_CLS_ONLY)."""
        x = hip_ops.cast(x, hip_ops.compute_dtype(x))
        g0, x0 = hip_ops.PermutClsFn.apply(x, self._table(), self.concat_dim)
        return self.linear(g0), x0


def _padded_linear(x, weight):
    """x @ weight^T on the MFMA GEMM for an output width that is not a multiple of the 16-byte chunk (257 = D//2+1):
    the weight is zero-padded to the next multiple of 8 rows and the extra columns are sliced away again."""
    n = weight.shape[0]
    npad = (n + 7) // 8 * 8
    w = F.pad(weight, (0, 0, 0, npad - n)) if npad != n else weight
    dt = hip_ops.compute_dtype(x)
    y = hip_ops.linear(hip_ops.cast(x, dt), w, None)
    return y[..., :n] if npad != n else y


class FFTApproximator(nn.Module):
    """Learned stand-in for rfft: x @ W^T with W (D//2+1, D)  (reference layers.py:104-121)."""

    def __init__(self, dim) -> None:
        super().__init__()
        self.out_dim = dim // 2 + 1
        self.dim = dim
        self.weight = nn.Parameter(torch.randn(self.out_dim, self.dim))

    def forward(self, x):
        return _padded_linear(x, self.weight)


class BinaryLinear(nn.Module):
    """scale * (x @ sign(W)^T)  (reference layers.py:10-23; sign() passes no gradient to W, as in the reference)."""

    def __init__(self, in_features, out_features, requires_grad=True):
        super().__init__()
        if requires_grad:
            self.weight = nn.Parameter(torch.randn(out_features, in_features))
        else:
            self.weight = nn.Parameter(torch.ones(out_features, in_features), requires_grad=False)
        self.scale = nn.Parameter(torch.ones(1), requires_grad=requires_grad)

    def forward(self, x):
        y = _padded_linear(x, self.weight.sign())
        return self.scale.to(y.dtype) * y
