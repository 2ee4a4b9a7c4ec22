"""CPU oracle for the Spectre-ViT training step  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-numpy restatement (forward + hand-derived backward) of the arithmetic on the
reference's hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this file; the product path
(``vit-spectre-experiments_amd/``) never does and fails loudly without its HIP library.

Pinning: every function here is checked by ``tests/test_oracle_golden.py`` against the
fixtures in ``tests/golden/*.npz``, which were produced by executing the reference's own
modules (imported read-only from /root/reference, CPU) with ``tests/golden/make_golden.py``.
Two rows have no reference implementation to execute and are therefore *parity unpinned*
(pinned to their mathematical definition only): the Haar-DWT mixers and the
FNet ``Re(fft2)`` mixer as an encoder ``mix_layer`` (the operator itself is pinned against
``torch.fft.fft2(...).real`` golden vectors).

Each function cites the reference file:line it restates (paths relative to the reference
root).  All functions take/return numpy arrays; ``dtype`` follows the inputs (float64 for
tight checks, float32 to mimic the reference's fp32 CPU path).
"""
from __future__ import annotations

import math

import numpy as np

try:  # scipy is available in the image; erf is the only thing taken from it
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf)

SQRT2 = math.sqrt(2.0)
INV_SQRT_2PI = 1.0 / math.sqrt(2.0 * math.pi)


# --------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------
def gelu(x):
    """nn.GELU() exact erf form -- spectre_vit/models/spectre/layers.py:88."""
    return 0.5 * x * (1.0 + _erf(x / SQRT2))


def gelu_grad(x):
    return 0.5 * (1.0 + _erf(x / SQRT2)) + x * np.exp(-0.5 * x * x) * INV_SQRT_2PI


def layernorm_fwd(x, gamma, beta, eps=1e-5):
    """nn.LayerNorm over the last axis (biased variance) -- layers.py:87, spectre.py:54-55."""
    mean = x.mean(axis=-1, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean) * rstd
    return xhat * gamma + beta, (xhat, rstd)


def layernorm_bwd(dy, gamma, cache):
    xhat, rstd = cache
    dxhat = dy * gamma
    m1 = dxhat.mean(axis=-1, keepdims=True)
    m2 = (dxhat * xhat).mean(axis=-1, keepdims=True)
    dx = rstd * (dxhat - m1 - xhat * m2)
    red = tuple(range(dy.ndim - 1))
    return dx, (dy * xhat).sum(axis=red), dy.sum(axis=red)


def adaptive_pool_windows(n_in, n_out):
    """Window i of nn.AdaptiveAvgPool1d(n_out) on length n_in: [floor(i*in/out), ceil((i+1)*in/out))
    -- layers.py:93."""
    starts = [(i * n_in) // n_out for i in range(n_out)]
    ends = [-((-(i + 1) * n_in) // n_out) for i in range(n_out)]
    return np.asarray(starts), np.asarray(ends)


def adaptive_pool_matrix(n_in, n_out, dtype=np.float64):
    """Dense (n_out, n_in) averaging matrix of the pooling; identity when n_in == n_out
    (layers.py:90-93)."""
    if n_in == n_out:
        return np.eye(n_in, dtype=dtype)
    s, e = adaptive_pool_windows(n_in, n_out)
    P = np.zeros((n_out, n_in), dtype=dtype)
    for i in range(n_out):
        P[i, s[i]:e[i]] = 1.0 / (e[i] - s[i])
    return P


# --------------------------------------------------------------------------------------
# SpectreLinear  (layers.py:76-101)
# --------------------------------------------------------------------------------------
def _mm(x, M):
    """x[..., k] @ M[k, n] through ONE 2-D BLAS call (numpy's batched matmul with a strided 2-D operand leaves BLAS and
    ran the MHPermutMix linear 100x slower)."""
    lead = x.shape[:-1]
    return (x.reshape(-1, x.shape[-1]) @ M).reshape(*lead, M.shape[1])


def spectre_linear_fwd(x, p):
    """out = GELU(LN(x W^T + b)) + avgpool(x)  -- layers.py:95-101.
    p: dict(weight (out,in), bias (out,), ln_weight, ln_bias)."""
    W = p["weight"]
    h = _mm(x, W.T) + p["bias"]
    ln, ln_cache = layernorm_fwd(h, p["ln_weight"], p["ln_bias"])
    P = adaptive_pool_matrix(W.shape[1], W.shape[0], x.dtype)
    out = gelu(ln) + _mm(x, P.T)
    return out, (x, ln, ln_cache, P)


def spectre_linear_bwd(dout, p, cache):
    x, ln, ln_cache, P = cache
    W = p["weight"]
    dln = dout * gelu_grad(ln)
    dh, dgamma, dbeta = layernorm_bwd(dln, p["ln_weight"], ln_cache)
    K = W.shape[1]
    dh2 = dh.reshape(-1, W.shape[0])
    dW = dh2.T @ x.reshape(-1, K)
    db = dh2.sum(axis=0)
    dx = _mm(dh, W) + _mm(dout, P)
    return dx, dict(weight=dW, bias=db, ln_weight=dgamma, ln_bias=dbeta)


# --------------------------------------------------------------------------------------
# MHPermutMix  (layers.py:53-73)
# --------------------------------------------------------------------------------------
def permut_gather_fwd(x, perms, signs):
    """g = x.view(B,-1)[:, perms] * signs ; raw reshape to (B, N, E*H) -- layers.py:68-72."""
    B, N, E = x.shape
    H = perms.shape[0]
    xf = x.reshape(B, N * E)
    g = xf[:, perms] * signs.reshape(1, H, N * E)
    return g.reshape(B, N, E * H)


def permut_gather_bwd(dg, perms, signs, N, E):
    B = dg.shape[0]
    H = perms.shape[0]
    d = N * E
    dgs = dg.reshape(B, H, d) * signs.reshape(1, H, d)
    dx = np.zeros((B, d), dtype=dg.dtype)
    for h in range(H):  # each perms[h] is a permutation -> plain scatter, no collisions
        dx[:, perms[h]] += dgs[:, h, :]
    return dx.reshape(B, N, E)


def mh_permut_mix_fwd(x, p):
    """MHPermutMix.forward -- layers.py:68-73.  p: perms, signs, linear{...}."""
    g = permut_gather_fwd(x, p["perms"], p["signs"])
    out, c = spectre_linear_fwd(g, p["linear"])
    return out, (c, x.shape)


def mh_permut_mix_bwd(dout, p, cache):
    c, xshape = cache
    dg, grads = spectre_linear_bwd(dout, p["linear"], c)
    dx = permut_gather_bwd(dg, p["perms"], p["signs"], xshape[1], xshape[2])
    return dx, dict(linear=grads)


# --------------------------------------------------------------------------------------
# spectral mixers
# --------------------------------------------------------------------------------------
def dft_cos_sin(n, dtype=np.float64):
    k = np.arange(n)
    ang = 2.0 * np.pi * ((k[:, None] * k[None, :]) % n) / n
    return np.cos(ang).astype(dtype), np.sin(ang).astype(dtype)


def fft_module_fwd(x):
    """FFT.forward: rfft(x, dim=-1).real -> (..., D//2+1) -- spectre_vit/modules/spectre.py:9-14."""
    D = x.shape[-1]
    k = np.arange(D // 2 + 1)
    n = np.arange(D)
    C = np.cos(2.0 * np.pi * ((k[:, None] * n[None, :]) % D) / D).astype(x.dtype)
    return x @ C.T


def fft_module_bwd(dy, D):
    k = np.arange(D // 2 + 1)
    n = np.arange(D)
    C = np.cos(2.0 * np.pi * ((k[:, None] * n[None, :]) % D) / D).astype(dy.dtype)
    return dy @ C


def fnet_mix_fwd(x):
    """FNet token mixer Re(fft2(x)) over the last two axes (N, D), un-normalised:
    y = C_N x C_D - S_N x S_D   -- spectre_branch.py:79 (comment), orthogonal_permut.py:23-28,
    'fft_bare' in spectre.py:31 (SURVEY 8a-6 ii)."""
    N, D = x.shape[-2:]
    CN, SN = dft_cos_sin(N, x.dtype)
    CD, SD = dft_cos_sin(D, x.dtype)
    return CN @ x @ CD - SN @ x @ SD


def fnet_mix_bwd(dy):
    """The operator is symmetric (C, S symmetric matrices) => adjoint == itself."""
    return fnet_mix_fwd(dy)


def haar_level_fwd(x, axis, mode="passthrough"):
    """One Haar level along `axis`: pairs (x0,x1)->((x0+x1)/sqrt2, (x0-x1)/sqrt2) -- the pair convention PyWavelets documents for
    'haar' = 'db1' (``pywt.dwt([1,2,3,4],'db1')`` -> cA [2.1213, 4.9497], cD [-0.7071, -0.7071]; tests/test_oracle_golden.py holds the
    documented vectors).  PARITY UNPINNED against the reference: it has no DWT model code, only dwt_experiments.py:56.

    An odd trailing element x_L has no partner.  mode="passthrough" (the default): it is copied into the approximation band, which
    keeps the map orthonormal and shape preserving.  mode="zero": the convention of the reference's only call
    (``DWTForward(J=3, wave="haar", mode="zero")``, dwt_experiments.py:56 -- pywt's 'zero' signal extension): the signal is extended
    by a zero, so the last pair is (x_L, 0) and gives cA_last = cD_last = x_L / sqrt2; pywt returns both (ceil(L/2) + ceil(L/2) =
    L + 1 coefficients).  A shape-preserving mixer has L slots: the band tensor keeps every cA and drops that LAST detail coefficient,
    which is a copy of the last approximation coefficient (no information is lost; the map is no longer orthonormal: x_L is scaled
    by 1/sqrt2).  haar_level_pywt_zero returns pywt's full (cA, cD) pair for the comparison."""
    if mode not in ("passthrough", "zero"):
        raise ValueError(mode)
    x = np.moveaxis(x, axis, -1)
    L = x.shape[-1]
    h = L // 2
    e, o = x[..., 0:2 * h:2], x[..., 1:2 * h:2]
    a = (e + o) / SQRT2
    d = (e - o) / SQRT2
    if L % 2:
        last = x[..., -1:] if mode == "passthrough" else x[..., -1:] / SQRT2
        a = np.concatenate([a, last], axis=-1)
    return np.moveaxis(a, -1, axis), np.moveaxis(d, -1, axis)


def haar_level_pywt_zero(x):
    """(cA, cD) of one level along the last axis as ``pywt.dwt(x, 'haar', mode='zero')`` returns them: ceil(L/2) coefficients each, an
    odd signal extended by one zero."""
    x = np.asarray(x)
    if x.shape[-1] % 2:
        x = np.concatenate([x, np.zeros(x.shape[:-1] + (1,), x.dtype)], axis=-1)
    return (x[..., 0::2] + x[..., 1::2]) / SQRT2, (x[..., 0::2] - x[..., 1::2]) / SQRT2


def haar_level_bwd(da, dd, axis, mode="passthrough"):
    """adjoint of haar_level_fwd"""
    da = np.moveaxis(da, axis, -1)
    dd = np.moveaxis(dd, axis, -1)
    h = dd.shape[-1]
    L = da.shape[-1] + h
    dx = np.empty(da.shape[:-1] + (L,), dtype=da.dtype)
    dx[..., 0:2 * h:2] = (da[..., :h] + dd) / SQRT2
    dx[..., 1:2 * h:2] = (da[..., :h] - dd) / SQRT2
    if L % 2:
        dx[..., -1] = da[..., -1] if mode == "passthrough" else da[..., -1] / SQRT2
    return np.moveaxis(dx, -1, axis)


def haar_dwt_fwd(x, axis=-1, levels=1, mode="passthrough"):
    """J-level Haar DWT mixer, output = concat[a_J, d_J, ..., d_1] along `axis` (same length) -- the coefficient order of
    ``pywt.wavedec`` ([cA_J, cD_J, ..., cD_1]).  axis=-1: 'dwt_embed', axis=-2: 'dwt_token' (spectre.py:33-34)."""
    bands = []
    a = x
    for _ in range(levels):
        a, d = haar_level_fwd(a, axis, mode)
        bands.append(d)
    return np.concatenate([a] + bands[::-1], axis=axis)


def haar_dwt_bwd(dy, axis=-1, levels=1, mode="passthrough"):
    """Adjoint (== inverse for mode "passthrough": that transform is orthonormal)."""
    L = dy.shape[axis]
    lens = []
    cur = L
    for _ in range(levels):
        lens.append(cur // 2)
        cur = cur - cur // 2
    dy = np.moveaxis(dy, axis, -1)
    da = dy[..., :cur]
    off = cur
    for h in lens[::-1]:
        dd = dy[..., off:off + h]
        off += h
        da = haar_level_bwd(da, dd, -1, mode)
    return np.moveaxis(da, -1, axis)


# --------------------------------------------------------------------------------------
# Walsh-Hadamard helpers  (hadamar.py:12-32, 58-80, 83-112, 115-141)
# --------------------------------------------------------------------------------------
def fwht(x, normalize=True):
    """natural-order fast Walsh-Hadamard transform along the last axis -- hadamar.py:12-32 (== hadamard_transform :83-112
    with normalize=True): stage h pairs (i, i+h) inside blocks of 2h."""
    n = x.shape[-1]
    y = x.reshape(-1, n).copy()
    h = 1
    while h < n:
        v = y.reshape(-1, n // (2 * h), 2, h)
        a, b = v[:, :, 0, :], v[:, :, 1, :]
        y = np.concatenate((a + b, a - b), axis=2).reshape(-1, n)
        h *= 2
    y = y.reshape(x.shape)
    return y * n ** -0.5 if normalize else y


def fwht_fast_fwd(x):
    """hadamar.py:58-80: per stage, block of 2h -> a = first h, b = last h, output interleaved (a+b)[i], (a-b)[i]."""
    n = x.shape[-1]
    y = x.reshape(-1, n).copy()
    h = 1
    while h < n:
        v = y.reshape(y.shape[0], -1, 2 * h)
        a, b = v[..., :h], v[..., h:]
        out = np.empty_like(v)
        out[..., 0::2] = a + b
        out[..., 1::2] = a - b
        y = out.reshape(-1, n)
        h *= 2
    return y.reshape(x.shape)


def fwht_fast_bwd(dy):
    """transpose of fwht_fast_fwd: stages in reverse order, each stage transposed."""
    n = dy.shape[-1]
    g = dy.reshape(-1, n).copy()
    h = n // 2
    while h >= 1:
        v = g.reshape(g.shape[0], -1, 2 * h)
        u, w = v[..., 0::2], v[..., 1::2]
        out = np.empty_like(v)
        out[..., :h] = u + w
        out[..., h:] = u - w
        g = out.reshape(-1, n)
        h //= 2
    return g.reshape(dy.shape)


def learnable_hadamard_fwd(x, num_blocks):
    """LearnableHadamard.forward -- hadamar.py:127-141 (parameters unused: `# * p` :136)."""
    d = x.shape[-1]
    n = 1 << (d - 1).bit_length()
    y = np.concatenate([x, np.zeros(x.shape[:-1] + (n - d,), x.dtype)], axis=-1)
    for _ in range(num_blocks):
        y = fwht_fast_fwd(y)
    return y[..., :d] + x


def learnable_hadamard_bwd(dy, num_blocks):
    d = dy.shape[-1]
    n = 1 << (d - 1).bit_length()
    g = np.concatenate([dy, np.zeros(dy.shape[:-1] + (n - d,), dy.dtype)], axis=-1)
    for _ in range(num_blocks):
        g = fwht_fast_bwd(g)
    return g[..., :d] + dy


# --------------------------------------------------------------------------------------
# multi-head self attention + stock TransformerEncoderLayer (vit.py:30-38)
# --------------------------------------------------------------------------------------
def _softmax(s):
    s = s - s.max(axis=-1, keepdims=True)
    e = np.exp(s)
    return e / e.sum(axis=-1, keepdims=True)


def mhsa_fwd(x, p, num_heads, batch_first=False):
    """nn.MultiheadAttention self-attention as used by nn.TransformerEncoderLayer.
    batch_first=False reproduces the reference quirk: a (B,N,E) tensor is read as (S=B, batch=N, E)
    -- vit.py:30-36,44-45.  p: in_proj_weight (3E,E), in_proj_bias, out_proj_weight, out_proj_bias."""
    xs = x if batch_first else np.swapaxes(x, 0, 1)  # -> (batch, S, E)
    Bt, S, E = xs.shape
    hd = E // num_heads
    qkv = xs @ p["in_proj_weight"].T + p["in_proj_bias"]
    q, k, v = np.split(qkv, 3, axis=-1)

    def heads(t):
        return t.reshape(Bt, S, num_heads, hd).transpose(0, 2, 1, 3)

    q, k, v = heads(q), heads(k), heads(v)
    att = _softmax((q @ np.swapaxes(k, -1, -2)) / math.sqrt(hd))
    ctx = (att @ v).transpose(0, 2, 1, 3).reshape(Bt, S, E)
    out = ctx @ p["out_proj_weight"].T + p["out_proj_bias"]
    cache = (xs, q, k, v, att, ctx)
    return (out if batch_first else np.swapaxes(out, 0, 1)), cache


def mhsa_bwd(dout, p, num_heads, cache, batch_first=False):
    xs, q, k, v, att, ctx = cache
    do = dout if batch_first else np.swapaxes(dout, 0, 1)
    Bt, S, E = xs.shape
    hd = E // num_heads
    g = {}
    g["out_proj_weight"] = do.reshape(-1, E).T @ ctx.reshape(-1, E)
    g["out_proj_bias"] = do.reshape(-1, E).sum(0)
    dctx = (do @ p["out_proj_weight"]).reshape(Bt, S, num_heads, hd).transpose(0, 2, 1, 3)
    datt = dctx @ np.swapaxes(v, -1, -2)
    dv = np.swapaxes(att, -1, -2) @ dctx
    ds = att * (datt - (datt * att).sum(-1, keepdims=True)) / math.sqrt(hd)
    dq = ds @ k
    dk = np.swapaxes(ds, -1, -2) @ q

    def unheads(t):
        return t.transpose(0, 2, 1, 3).reshape(Bt, S, E)

    dqkv = np.concatenate([unheads(dq), unheads(dk), unheads(dv)], axis=-1)
    g["in_proj_weight"] = dqkv.reshape(-1, 3 * E).T @ xs.reshape(-1, E)
    g["in_proj_bias"] = dqkv.reshape(-1, 3 * E).sum(0)
    dxs = dqkv @ p["in_proj_weight"]
    return (dxs if batch_first else np.swapaxes(dxs, 0, 1)), g


def transformer_layer_fwd(x, p, num_heads, batch_first=False):
    """Stock post-norm nn.TransformerEncoderLayer (norm_first=False, dropout off, exact GELU)
    -- vit.py:30-36.  p: attn{...}, linear1_weight/bias, linear2_weight/bias, norm1_*, norm2_*."""
    a, ca = mhsa_fwd(x, p["attn"], num_heads, batch_first)
    x1, c1 = layernorm_fwd(x + a, p["norm1_weight"], p["norm1_bias"])
    h = x1 @ p["linear1_weight"].T + p["linear1_bias"]
    f = gelu(h) @ p["linear2_weight"].T + p["linear2_bias"]
    x2, c2 = layernorm_fwd(x1 + f, p["norm2_weight"], p["norm2_bias"])
    return x2, (ca, c1, x1, h, c2)


def transformer_layer_bwd(dout, p, num_heads, cache, batch_first=False):
    ca, c1, x1, h, c2 = cache
    g = {}
    ds2, g["norm2_weight"], g["norm2_bias"] = layernorm_bwd(dout, p["norm2_weight"], c2)
    E = x1.shape[-1]
    F = h.shape[-1]
    gh = gelu(h)
    g["linear2_weight"] = ds2.reshape(-1, E).T @ gh.reshape(-1, F)
    g["linear2_bias"] = ds2.reshape(-1, E).sum(0)
    dh = (ds2 @ p["linear2_weight"]) * gelu_grad(h)
    g["linear1_weight"] = dh.reshape(-1, F).T @ x1.reshape(-1, E)
    g["linear1_bias"] = dh.reshape(-1, F).sum(0)
    dx1 = ds2 + dh @ p["linear1_weight"]
    ds1, g["norm1_weight"], g["norm1_bias"] = layernorm_bwd(dx1, p["norm1_weight"], c1)
    dxa, g["attn"] = mhsa_bwd(ds1, p["attn"], num_heads, ca, batch_first)
    return ds1 + dxa, g


# --------------------------------------------------------------------------------------
# patch embeddings
# --------------------------------------------------------------------------------------
def normalize_u8(img_u8_hwc, mean, std):
    """The loader's per-image transform restated for a whole batch: ToTensor (uint8 HWC -> float CHW / 255) followed by
    Normalize(mean, std) per channel -- spectre_vit/repl/train.py:102-112.  (B,H,W,C) uint8 -> (B,C,H,W) float64."""
    x = img_u8_hwc.astype(np.float64) / 255.0
    x = (x - np.asarray(mean, np.float64)) / np.asarray(std, np.float64)
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2))


def patchify(img, P):
    """(B,C,H,W) -> (B, N, C, P, P) with n = ih*nW + iw  -- spectre.py:130-133."""
    B, C, H, W = img.shape
    nH, nW = H // P, W // P
    x = img[:, :, :nH * P, :nW * P].reshape(B, C, nH, P, nW, P)
    return x.transpose(0, 2, 4, 1, 3, 5).reshape(B, nH * nW, C, P, P)


def rfft2_real_matrix(P, dtype=np.float64):
    """R[(u,v),(p,q)] = cos(2*pi*(u*p + v*q)/P) / P : Re(rfft2(norm='ortho')) of one PxP patch,
    u in [0,P), v in [0,P//2]  -- spectre.py:136."""
    Pv = P // 2 + 1
    u = np.arange(P)[:, None, None, None]
    v = np.arange(Pv)[None, :, None, None]
    pp = np.arange(P)[None, None, :, None]
    q = np.arange(P)[None, None, None, :]
    R = np.cos(2.0 * np.pi * (u * pp + v * q) / P) / P
    return R.reshape(P * Pv, P * P).astype(dtype)


def spectral_patch_embed_fwd(img, p, P):
    """SpectralPatchEmbed.forward (dropout off) -- spectre.py:124-156.
    p: freq_weight_h (P,), freq_weight_w (P//2+1,), proj_weight (E, C*P*Pv), proj_bias, cls_token (1,1,E),
    position_embeddings (1,N+1,E)."""
    B, C = img.shape[:2]
    Pv = P // 2 + 1
    pt = patchify(img, P)  # (B,N,C,P,P)
    N = pt.shape[1]
    R = rfft2_real_matrix(P, img.dtype)
    spec = pt.reshape(B, N, C, P * P) @ R.T  # (B,N,C,P*Pv)
    fw = (p["freq_weight_h"][:, None] * p["freq_weight_w"][None, :]).reshape(P * Pv)
    feat = (spec * fw).reshape(B, N, C * P * Pv)
    proj = feat @ p["proj_weight"].T + p["proj_bias"]
    cls = np.broadcast_to(p["cls_token"], (B, 1, proj.shape[-1]))
    out = np.concatenate([cls, proj], axis=1) + p["position_embeddings"]
    return out, (spec, feat, fw)


def spectral_patch_embed_bwd(dout, p, P, cache):
    spec, feat, fw = cache
    B, N, C, _ = spec.shape
    Pv = P // 2 + 1
    E = dout.shape[-1]
    g = {}
    g["position_embeddings"] = dout.sum(0, keepdims=True)
    g["cls_token"] = dout[:, 0:1, :].sum(0, keepdims=True)
    dproj = dout[:, 1:, :]
    g["proj_bias"] = dproj.reshape(-1, E).sum(0)
    g["proj_weight"] = dproj.reshape(-1, E).T @ feat.reshape(B * N, -1)
    dfeat = (dproj @ p["proj_weight"]).reshape(B, N, C, P, Pv)
    s = spec.reshape(B, N, C, P, Pv)
    g["freq_weight_h"] = (dfeat * s * p["freq_weight_w"][None, None, None, None, :]).sum((0, 1, 2, 4))
    g["freq_weight_w"] = (dfeat * s * p["freq_weight_h"][None, None, None, :, None]).sum((0, 1, 2, 3))
    return g


def conv_patch_embed_fwd(img, p, P):
    """PatchEmbedding.forward (dropout off): Conv2d(k=P,s=P) == per-patch GEMM -- patch_embeddings.py:28-43.
    p: conv_weight (E,C,P,P), conv_bias, cls_token, position_embeddings."""
    B = img.shape[0]
    pt = patchify(img, P)
    N = pt.shape[1]
    E = p["conv_weight"].shape[0]
    feat = pt.reshape(B, N, -1)
    proj = feat @ p["conv_weight"].reshape(E, -1).T + p["conv_bias"]
    cls = np.broadcast_to(p["cls_token"], (B, 1, E))
    return np.concatenate([cls, proj], axis=1) + p["position_embeddings"], feat


def conv_patch_embed_bwd(dout, p, P, feat):
    E = dout.shape[-1]
    g = {}
    g["position_embeddings"] = dout.sum(0, keepdims=True)
    g["cls_token"] = dout[:, 0:1, :].sum(0, keepdims=True)
    dproj = dout[:, 1:, :].reshape(-1, E)
    g["conv_bias"] = dproj.sum(0)
    g["conv_weight"] = (dproj.T @ feat.reshape(dproj.shape[0], -1)).reshape(p["conv_weight"].shape)
    return g


# --------------------------------------------------------------------------------------
# encoder layer / encoder / model  (spectre.py:29-103, 159-202)
# --------------------------------------------------------------------------------------
def mixer_fwd(x, lp, mixer):
    if mixer == "permut":
        return mh_permut_mix_fwd(x, lp["mix_layer"])
    if mixer == "fft":
        return fnet_mix_fwd(x), None
    if mixer == "dwt_embed":
        return haar_dwt_fwd(x, -1, lp.get("dwt_levels", 1)), None
    if mixer == "dwt_token":
        return haar_dwt_fwd(x, -2, lp.get("dwt_levels", 1), lp.get("dwt_mode", "passthrough")), None
    raise ValueError(mixer)


def mixer_bwd(dy, lp, mixer, cache):
    if mixer == "permut":
        return mh_permut_mix_bwd(dy, lp["mix_layer"], cache)
    if mixer == "fft":
        return fnet_mix_bwd(dy), {}
    if mixer == "dwt_embed":
        return haar_dwt_bwd(dy, -1, lp.get("dwt_levels", 1)), {}
    if mixer == "dwt_token":
        return haar_dwt_bwd(dy, -2, lp.get("dwt_levels", 1), lp.get("dwt_mode", "passthrough")), {}
    raise ValueError(mixer)


def encoder_layer_fwd(x, lp, mixer="permut"):
    """SpectreEncoderLayer.forward (dropout off) -- spectre.py:65-73:
    x1 = norm1(mix(x)) + x ; x2 = norm2(x1 + linear3(linear1(x1)))."""
    m, cm = mixer_fwd(x, lp, mixer)
    n1, c1 = layernorm_fwd(m, lp["norm1_weight"], lp["norm1_bias"])
    x1 = n1 + x
    f1, cf1 = spectre_linear_fwd(x1, lp["linear1"])
    f3, cf3 = spectre_linear_fwd(f1, lp["linear3"])
    x2, c2 = layernorm_fwd(x1 + f3, lp["norm2_weight"], lp["norm2_bias"])
    return x2, (cm, c1, cf1, cf3, c2)


def encoder_layer_bwd(dout, lp, cache, mixer="permut"):
    cm, c1, cf1, cf3, c2 = cache
    g = {}
    ds, g["norm2_weight"], g["norm2_bias"] = layernorm_bwd(dout, lp["norm2_weight"], c2)
    df1, g["linear3"] = spectre_linear_bwd(ds, lp["linear3"], cf3)
    dx1_ff, g["linear1"] = spectre_linear_bwd(df1, lp["linear1"], cf1)
    dx1 = ds + dx1_ff
    dm, g["norm1_weight"], g["norm1_bias"] = layernorm_bwd(dx1, lp["norm1_weight"], c1)
    dx_mix, gm = mixer_bwd(dm, lp, mixer, cm)
    if gm:
        g["mix_layer"] = gm
    return dx1 + dx_mix, g


def spectre_vit_fwd(img, params, patch_size, mixer="permut"):
    """SpectreViT.forward (dropout off) -- spectre.py:194-202 with SpectreEncoder.forward :90-103.
    params: embed{...}, layers[list of layer dicts], head{SpectreLinear}.  Returns logits, cls, cache."""
    x0, ce = spectral_patch_embed_fwd(img, params["embed"], patch_size)
    x = x0
    caches = []
    for lp in params["layers"]:
        x, c = encoder_layer_fwd(x, lp, mixer)
        caches.append(c)
    x = x + x0  # global residual, spectre.py:103
    cls = x[:, 0, :]
    logits, ch = spectre_linear_fwd(cls, params["head"])
    return logits, cls, (ce, caches, ch, x.shape)


def spectre_vit_bwd(dlogits, params, patch_size, cache, mixer="permut", dcls_extra=None):
    ce, caches, ch, xshape = cache
    g = {"layers": [None] * len(params["layers"])}
    dcls, g["head"] = spectre_linear_bwd(dlogits, params["head"], ch)
    if dcls_extra is not None:
        dcls = dcls + dcls_extra
    dx_top = np.zeros(xshape, dtype=dlogits.dtype)
    dx_top[:, 0, :] = dcls
    dx = dx_top
    for i in range(len(params["layers"]) - 1, -1, -1):
        dx, g["layers"][i] = encoder_layer_bwd(dx, params["layers"][i], caches[i], mixer)
    dx0 = dx + dx_top
    g["embed"] = spectral_patch_embed_bwd(dx0, params["embed"], patch_size, ce)
    return g


# --------------------------------------------------------------------------------------
# losses + optimizer as driven by the script (train.py:196-205,226,334-348)
# --------------------------------------------------------------------------------------
def log_softmax(z):
    z = z - z.max(axis=-1, keepdims=True)
    return z - np.log(np.exp(z).sum(axis=-1, keepdims=True))


def cross_entropy_fwd_bwd(logits, labels):
    """nn.CrossEntropyLoss() (mean) -- train.py:196,226."""
    B = logits.shape[0]
    ls = log_softmax(logits)
    loss = -ls[np.arange(B), labels].mean()
    d = np.exp(ls)
    d[np.arange(B), labels] -= 1.0
    return loss, d / B


def distill_loss_fwd_bwd(student_logits, teacher_logits, labels, T=2.0, w_soft=0.25, w_ce=0.75):
    """loss = w_soft * T^2 * sum(p_t (log p_t - log p_s)) / B + w_ce * CE -- train.py:300-302,334-348."""
    B = student_logits.shape[0]
    pt = np.exp(log_softmax(teacher_logits / T))
    lps = log_softmax(student_logits / T)
    soft = (pt * (np.log(pt) - lps)).sum() / B * T * T
    ce, dce = cross_entropy_fwd_bwd(student_logits, labels)
    dsoft = (np.exp(lps) - pt) * (T / B)  # d/dz of T^2/B * sum p_t(-log_softmax(z/T))
    return w_soft * soft + w_ce * ce, w_soft * dsoft + w_ce * dce, soft, ce


def adamw_step(param, grad, m, v, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, wd=0.01):
    """torch.optim.AdamW single-tensor update -- train.py:199-201 (configs/spectre_vit_cifar100.py:14-15)."""
    b1, b2 = betas
    param = param * (1.0 - lr * wd)
    m = b1 * m + (1 - b1) * grad
    v = b2 * v + (1 - b2) * grad * grad
    mhat = m / (1 - b1 ** step)
    vhat = v / (1 - b2 ** step)
    return param - lr * mhat / (np.sqrt(vhat) + eps), m, v


# --------------------------------------------------------------------------------------
# state_dict <-> oracle parameter trees  (key names: SURVEY 8b)
# --------------------------------------------------------------------------------------
def _sl_from_sd(sd, prefix, dtype):
    return dict(weight=np.asarray(sd[prefix + "local_head.0.weight"], dtype),
                bias=np.asarray(sd[prefix + "local_head.0.bias"], dtype),
                ln_weight=np.asarray(sd[prefix + "local_head.1.weight"], dtype),
                ln_bias=np.asarray(sd[prefix + "local_head.1.bias"], dtype))


def params_from_state_dict(sd, num_layers, mixer="permut", dtype=np.float64):
    """Build the oracle parameter tree from a SpectreViT state_dict (numpy values)."""
    e = "embeddings_block."
    embed = dict(freq_weight_h=np.asarray(sd[e + "freq_weight_h"], dtype),
                 freq_weight_w=np.asarray(sd[e + "freq_weight_w"], dtype),
                 proj_weight=np.asarray(sd[e + "proj.weight"], dtype),
                 proj_bias=np.asarray(sd[e + "proj.bias"], dtype),
                 cls_token=np.asarray(sd[e + "cls_token"], dtype),
                 position_embeddings=np.asarray(sd[e + "position_embeddings"], dtype))
    layers = []
    for i in range(num_layers):
        pre = f"encoder_blocks.layers.{i}."
        lp = dict(linear1=_sl_from_sd(sd, pre + "linear1.", dtype),
                  linear3=_sl_from_sd(sd, pre + "linear3.", dtype),
                  norm1_weight=np.asarray(sd[pre + "norm1.weight"], dtype),
                  norm1_bias=np.asarray(sd[pre + "norm1.bias"], dtype),
                  norm2_weight=np.asarray(sd[pre + "norm2.weight"], dtype),
                  norm2_bias=np.asarray(sd[pre + "norm2.bias"], dtype))
        if mixer == "permut":
            lp["mix_layer"] = dict(perms=np.asarray(sd[pre + "mix_layer.perms"], np.int64),
                                   signs=np.asarray(sd[pre + "mix_layer.signs"], dtype),
                                   linear=_sl_from_sd(sd, pre + "mix_layer.linear.", dtype))
        layers.append(lp)
    return dict(embed=embed, layers=layers, head=_sl_from_sd(sd, "mlp_head.0.", dtype))


def _sl_grads_to_sd(g, prefix, out):
    out[prefix + "local_head.0.weight"] = g["weight"]
    out[prefix + "local_head.0.bias"] = g["bias"]
    out[prefix + "local_head.1.weight"] = g["ln_weight"]
    out[prefix + "local_head.1.bias"] = g["ln_bias"]


def grads_to_state_dict(g):
    """Flatten an oracle gradient tree to state_dict-style names."""
    out = {}
    e = "embeddings_block."
    ge = g["embed"]
    out[e + "freq_weight_h"] = ge["freq_weight_h"]
    out[e + "freq_weight_w"] = ge["freq_weight_w"]
    out[e + "proj.weight"] = ge["proj_weight"]
    out[e + "proj.bias"] = ge["proj_bias"]
    out[e + "cls_token"] = ge["cls_token"]
    out[e + "position_embeddings"] = ge["position_embeddings"]
    for i, gl in enumerate(g["layers"]):
        pre = f"encoder_blocks.layers.{i}."
        _sl_grads_to_sd(gl["linear1"], pre + "linear1.", out)
        _sl_grads_to_sd(gl["linear3"], pre + "linear3.", out)
        out[pre + "norm1.weight"] = gl["norm1_weight"]
        out[pre + "norm1.bias"] = gl["norm1_bias"]
        out[pre + "norm2.weight"] = gl["norm2_weight"]
        out[pre + "norm2.bias"] = gl["norm2_bias"]
        if "mix_layer" in gl:
            _sl_grads_to_sd(gl["mix_layer"]["linear"], pre + "mix_layer.linear.", out)
    _sl_grads_to_sd(g["head"], "mlp_head.0.", out)
    return out


def train_step(img, labels, sd, num_layers, patch_size, mixer="permut", dtype=np.float64):
    """forward + CE + backward for a SpectreViT state_dict; returns loss, logits, cls, grads (sd names)."""
    params = params_from_state_dict(sd, num_layers, mixer, dtype)
    logits, cls, cache = spectre_vit_fwd(np.asarray(img, dtype), params, patch_size, mixer)
    loss, dlogits = cross_entropy_fwd_bwd(logits, np.asarray(labels, np.int64))
    g = spectre_vit_bwd(dlogits, params, patch_size, cache, mixer)
    return loss, logits, cls, grads_to_state_dict(g)
