"""Walsh-Hadamard helpers on the HIP butterfly kernel (``spv_fwht``) -- the build's counterpart of reference
spectre_vit/models/spectre/hadamar.py (research side-branch, only imported by a benchmark there).

Same names and argument meaning: ``next_pow2``, ``fwht(x, dim=-1, normalize=True)`` (:12-32), ``fwht_fast(x)`` (:58-80:
interleaving stages, un-normalised, different output order), ``hadamard_transform(x)`` (:83-112: 1-D / 2-D input only,
normalised) and ``LearnableHadamard(dim, num_blocks=2)`` (:115-141) whose parameters exist (``params.0 ...`` in the
state_dict) but take no part in the forward, as in the reference (``# * p``, :136) -- they receive no gradient.
"""
import torch
from torch import nn

from spectre_vit import _native
from spectre_vit.hip_ops import _dt, _p, _require_gpu, _stream, compute_dtype


def next_pow2(n):
    return 1 << (n - 1).bit_length()


class _FwhtFn(torch.autograd.Function):
    """y = crop(M^repeat pad(x)) * scale (+ x); M = natural-order butterflies (mode 0, symmetric) or fwht_fast's network (mode 1,
    whose transpose is mode 2)."""

    @staticmethod
    def forward(ctx, x, n, n_out, mode, repeat, scale, add_residual):
        _require_gpu(x)
        n_in = x.shape[-1]
        dt = compute_dtype(x)
        xc = x.reshape(-1, n_in).to(dt).contiguous()
        rows = xc.shape[0]
        y = torch.empty((rows, n_out), dtype=dt, device=x.device)
        _native.call("spv_fwht", _p(xc), _p(y), _p(xc) if add_residual else 0, rows, n_in, n, n_out, mode, repeat, float(scale), _dt(xc),
                     _stream())
        ctx.meta = (x.shape, x.dtype, n_in, n, n_out, mode, repeat, float(scale), add_residual)
        return y.reshape(*x.shape[:-1], n_out)

    @staticmethod
    def backward(ctx, dy):
        shape, xdtype, n_in, n, n_out, mode, repeat, scale, add_residual = ctx.meta
        dyc = dy.reshape(-1, n_out).contiguous()
        rows = dyc.shape[0]
        dx = torch.empty((rows, n_in), dtype=dyc.dtype, device=dy.device)
        # transpose of crop . M^r . pad = crop' . (M^T)^r . pad'; the residual's gradient is dy itself (n_in == n_out there)
        _native.call("spv_fwht", _p(dyc), _p(dx), _p(dyc) if add_residual else 0, rows, n_out, n, n_in, {0: 0, 1: 2, 2: 1}[mode], repeat,
                     scale, _dt(dyc), _stream())
        return dx.reshape(shape).to(xdtype), None, None, None, None, None, None


def _pow2(n, who):
    if n < 1 or n & (n - 1):
        raise ValueError(f"{who}: length {n} is not a power of two")


def fwht(x, dim=-1, normalize=True):
    """Fast Walsh-Hadamard transform along ``dim`` in natural (Sylvester) order; ``normalize`` scales by n^-1/2."""
    n = x.size(dim)
    _pow2(n, "fwht")
    xt = x.transpose(dim, -1)
    y = _FwhtFn.apply(xt, n, n, 0, 1, n ** -0.5 if normalize else 1.0, False)
    return y.transpose(dim, -1)


def fwht_fast(x):
    """x: [..., N], N a power of two; the reference's interleaving butterfly network (un-normalised)."""
    n = x.shape[-1]
    _pow2(n, "fwht_fast")
    return _FwhtFn.apply(x, n, n, 1, 1, 1.0, False)


def hadamard_transform(x: torch.Tensor):
    """normalised Hadamard transform of a vector or of every row of a matrix."""
    assert 1 <= x.dim() <= 2, "input's dimension must be either 1 or 2"
    n = x.shape[-1]
    _pow2(n, "hadamard_transform")
    return _FwhtFn.apply(x, n, n, 0, 1, n ** -0.5, False)


class LearnableHadamard(nn.Module):
    def __init__(self, dim, num_blocks=2):
        super().__init__()
        self.orig_dim = dim
        self.dim = next_pow2(dim)  # internal power-of-2 dim
        self.pad = self.dim - dim
        self.params = nn.ParameterList([nn.Parameter(torch.ones(self.dim)) for _ in range(num_blocks)])

    def forward(self, x):
        # pad -> num_blocks x fwht_fast -> crop -> + residual, one kernel
        return _FwhtFn.apply(x, self.dim, self.orig_dim, 1, len(self.params), 1.0, True)
