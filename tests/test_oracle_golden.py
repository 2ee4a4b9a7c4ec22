"""Pins the numpy oracle (oracle/spectre_oracle.py) to golden vectors produced by executing the
reference's own modules (tests/golden/make_golden.py).  float64 on both sides -> tight tolerances."""
import numpy as np
import pytest

from conftest import load_model_fixture
from oracle import spectre_oracle as O

RT, AT = 1e-9, 1e-10


def close(a, b, rtol=RT, atol=AT):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol)


def sl_params(g, pre):
    return dict(weight=g[pre + "local_head.0.weight"].astype(np.float64),
                bias=g[pre + "local_head.0.bias"].astype(np.float64),
                ln_weight=g[pre + "local_head.1.weight"].astype(np.float64),
                ln_bias=g[pre + "local_head.1.bias"].astype(np.float64))


def check_sl_grads(grads, g, pre):
    close(grads["weight"], g[pre + "local_head.0.weight"])
    close(grads["bias"], g[pre + "local_head.0.bias"])
    close(grads["ln_weight"], g[pre + "local_head.1.weight"])
    close(grads["ln_bias"], g[pre + "local_head.1.bias"])


@pytest.mark.parametrize("name", ["sl_equal", "sl_down_exact", "sl_down_overlap", "sl_up", "sl_head2d"])
def test_spectre_linear(golden_ops, name):
    g = golden_ops
    p = sl_params(g, f"{name}.sd.")
    y, cache = O.spectre_linear_fwd(g[f"{name}.x"], p)
    close(y, g[f"{name}.y"])
    dx, grads = O.spectre_linear_bwd(g[f"{name}.dy"], p, cache)
    close(dx, g[f"{name}.dx"])
    check_sl_grads(grads, g, f"{name}.grad.")


def test_pool_windows_match_survey():
    # SURVEY 8a-3 probe: 8192->512 exact 16-means; 768->512 overlapping 2-windows; 512->768 sizes 1-2; 512->100 sizes 6-7
    s, e = O.adaptive_pool_windows(8192, 512)
    assert set(e - s) == {16}
    s, e = O.adaptive_pool_windows(768, 512)
    assert set(e - s) == {2}
    s, e = O.adaptive_pool_windows(512, 768)
    assert set(e - s) == {1, 2}
    s, e = O.adaptive_pool_windows(512, 100)
    assert set(e - s) == {6, 7}


def test_mh_permut_mix(golden_ops):
    g = golden_ops
    p = dict(perms=g["permut.sd.perms"], signs=g["permut.sd.signs"].astype(np.float64),
             linear=sl_params(g, "permut.sd.linear."))
    close(O.permut_gather_fwd(g["permut.x"], p["perms"], p["signs"]), g["permut.gathered"])
    y, cache = O.mh_permut_mix_fwd(g["permut.x"], p)
    close(y, g["permut.y"])
    dx, grads = O.mh_permut_mix_bwd(g["permut.dy"], p, cache)
    close(dx, g["permut.dx"])
    check_sl_grads(grads["linear"], g, "permut.grad.linear.")


def test_permut_raw_reshape_closed_form(golden_ops):
    """element (t,c) of the reshaped tensor is the gather at flat f=t*E*H+c, h=f//d, j=f%d."""
    g = golden_ops
    x = g["permut.x"]
    B, N, E = x.shape
    perms, signs = g["permut.sd.perms"], g["permut.sd.signs"]
    H, d = perms.shape
    got = g["permut.gathered"]
    xf = x.reshape(B, d)
    for t in range(N):
        for c in range(0, E * H, 5):
            f = t * E * H + c
            h, j = divmod(f, d)
            assert got[1, t, c] == xf[1, perms[h, j]] * signs[0, h, j]


def test_spectral_patch_embed(golden_ops):
    g = golden_ops
    f = lambda k: g["spe.sd." + k].astype(np.float64)
    p = dict(freq_weight_h=f("freq_weight_h"), freq_weight_w=f("freq_weight_w"), proj_weight=f("proj.weight"),
             proj_bias=f("proj.bias"), cls_token=f("cls_token"), position_embeddings=f("position_embeddings"))
    y, cache = O.spectral_patch_embed_fwd(g["spe.x"], p, 4)
    close(y, g["spe.y"])
    grads = O.spectral_patch_embed_bwd(g["spe.dy"], p, 4, cache)
    for k_o, k_g in [("freq_weight_h", "freq_weight_h"), ("freq_weight_w", "freq_weight_w"),
                     ("proj_weight", "proj.weight"), ("proj_bias", "proj.bias"), ("cls_token", "cls_token"),
                     ("position_embeddings", "position_embeddings")]:
        close(grads[k_o], g["spe.grad." + k_g])


def test_conv_patch_embed(golden_ops):
    g = golden_ops
    f = lambda k: g["pe.sd." + k].astype(np.float64)
    p = dict(conv_weight=f("patcher.0.weight"), conv_bias=f("patcher.0.bias"), cls_token=f("cls_token"),
             position_embeddings=f("position_embeddings"))
    y, feat = O.conv_patch_embed_fwd(g["pe.x"], p, 4)
    close(y, g["pe.y"])
    grads = O.conv_patch_embed_bwd(g["pe.dy"], p, 4, feat)
    close(grads["conv_weight"], g["pe.grad.patcher.0.weight"])
    close(grads["conv_bias"], g["pe.grad.patcher.0.bias"])
    close(grads["cls_token"], g["pe.grad.cls_token"])
    close(grads["position_embeddings"], g["pe.grad.position_embeddings"])


def test_fft_module_and_fnet(golden_ops):
    g = golden_ops
    close(O.fft_module_fwd(g["fftmod.x"]), g["fftmod.y"])
    close(O.fft_module_bwd(g["fftmod.dy"], g["fftmod.x"].shape[-1]), g["fftmod.dx"])
    close(O.fnet_mix_fwd(g["fnet.x"]), g["fnet.y"])
    close(O.fnet_mix_bwd(g["fnet.dy"]), g["fnet.dx"])
    close(O.fnet_mix_fwd(g["fnet65.x"]), g["fnet65.y"], rtol=1e-8, atol=1e-9)


def test_fnet_symmetry_property():
    """y[m,k] == y[(N-m)%N,(D-k)%D] for real input -- the property the HIP kernel exploits."""
    x = np.random.default_rng(0).standard_normal((2, 65, 32))
    y = O.fnet_mix_fwd(x)
    m = (-np.arange(65)) % 65
    k = (-np.arange(32)) % 32
    close(y, y[:, m][:, :, k], rtol=1e-8, atol=1e-9)


def test_encoder_layer(golden_ops):
    g = golden_ops
    sd = {k[len("layer.sd."):]: v for k, v in g.items() if k.startswith("layer.sd.")}
    f = lambda k: sd[k].astype(np.float64)
    lp = dict(mix_layer=dict(perms=sd["mix_layer.perms"], signs=f("mix_layer.signs"),
                             linear=sl_params(g, "layer.sd.mix_layer.linear.")),
              linear1=sl_params(g, "layer.sd.linear1."), linear3=sl_params(g, "layer.sd.linear3."),
              norm1_weight=f("norm1.weight"), norm1_bias=f("norm1.bias"),
              norm2_weight=f("norm2.weight"), norm2_bias=f("norm2.bias"))
    y, cache = O.encoder_layer_fwd(g["layer.x"], lp, "permut")
    close(y, g["layer.y"])
    dx, grads = O.encoder_layer_bwd(g["layer.dy"], lp, cache, "permut")
    close(dx, g["layer.dx"])
    check_sl_grads(grads["linear1"], g, "layer.grad.linear1.")
    check_sl_grads(grads["linear3"], g, "layer.grad.linear3.")
    check_sl_grads(grads["mix_layer"]["linear"], g, "layer.grad.mix_layer.linear.")
    close(grads["norm1_weight"], g["layer.grad.norm1.weight"])
    close(grads["norm2_bias"], g["layer.grad.norm2.bias"])


def tel_params(g, pre):
    f = lambda k: g[pre + k].astype(np.float64)
    return dict(attn=dict(in_proj_weight=f("self_attn.in_proj_weight"), in_proj_bias=f("self_attn.in_proj_bias"),
                          out_proj_weight=f("self_attn.out_proj.weight"), out_proj_bias=f("self_attn.out_proj.bias")),
                linear1_weight=f("linear1.weight"), linear1_bias=f("linear1.bias"),
                linear2_weight=f("linear2.weight"), linear2_bias=f("linear2.bias"),
                norm1_weight=f("norm1.weight"), norm1_bias=f("norm1.bias"),
                norm2_weight=f("norm2.weight"), norm2_bias=f("norm2.bias"))


def test_transformer_layer_batch_axis_quirk(golden_ops):
    """vit.py:30-36: batch_first=False fed (B,N,E) => attention across the batch axis."""
    g = golden_ops
    p = tel_params(g, "tel.sd.")
    y, cache = O.transformer_layer_fwd(g["tel.x"], p, 4, batch_first=False)
    close(y, g["tel.y"])
    dx, grads = O.transformer_layer_bwd(g["tel.dy"], p, 4, cache, batch_first=False)
    close(dx, g["tel.dx"])
    close(grads["attn"]["in_proj_weight"], g["tel.grad.self_attn.in_proj_weight"])
    close(grads["attn"]["out_proj_bias"], g["tel.grad.self_attn.out_proj.bias"])
    close(grads["linear1_weight"], g["tel.grad.linear1.weight"])
    close(grads["linear2_weight"], g["tel.grad.linear2.weight"])
    close(grads["norm1_weight"], g["tel.grad.norm1.weight"])
    # the sane batch_first=True mode must differ (it is the build's own extension)
    y2, _ = O.transformer_layer_fwd(g["tel.x"], p, 4, batch_first=True)
    assert np.abs(y2 - g["tel.y"]).max() > 1e-3


def test_baseline_vit_forward(golden_ops):
    g = golden_ops
    f = lambda k: g["vit.sd." + k].astype(np.float64)
    p = dict(conv_weight=f("embeddings_block.patcher.0.weight"), conv_bias=f("embeddings_block.patcher.0.bias"),
             cls_token=f("embeddings_block.cls_token"), position_embeddings=f("embeddings_block.position_embeddings"))
    x, _ = O.conv_patch_embed_fwd(g["vit.x"], p, 4)
    for i in range(2):
        x, _ = O.transformer_layer_fwd(x, tel_params(g, f"vit.sd.encoder_blocks.layers.{i}."), 4, batch_first=False)
    cls = x[:, 0, :]
    close(cls, g["vit.cls"])
    close(cls @ f("mlp_head.0.weight").T + f("mlp_head.0.bias"), g["vit.logits"])
    # logits identical for every image
    assert np.abs(g["vit.logits"] - g["vit.logits"][0]).max() < 1e-12


@pytest.mark.parametrize("name", ["model_tiny_mnist", "model_small_cut"])
def test_whole_model_train_step(name):
    d, cfg = load_model_fixture(name)
    sd = {k[3:]: v for k, v in d.items() if k.startswith("sd.")}
    loss, logits, cls, grads = O.train_step(d["img"], d["labels"], sd, cfg["num_encoders"], cfg["patch_size"],
                                            "permut", np.float64)
    close(logits, d["logits"], rtol=1e-8, atol=1e-9)
    close(cls, d["cls"], rtol=1e-8, atol=1e-9)
    close(loss, d["loss"], rtol=1e-9)
    n = 0
    for k, v in d.items():
        if k.startswith("grad."):
            close(grads[k[5:]], v, rtol=1e-7, atol=1e-9)
            n += 1
    assert n == len(grads)
    # one AdamW step (train.py:199-201)
    for k, gk in grads.items():
        p0 = sd[k].astype(np.float64)
        p1, _, _ = O.adamw_step(p0, gk, np.zeros_like(p0), np.zeros_like(p0), 1)
        close(p1, d["after." + k], rtol=1e-7, atol=1e-9)


def test_haar_dwt_orthonormal_roundtrip():
    """PARITY UNPINNED row: pinned to the definition -- orthonormal, adjoint == inverse."""
    rng = np.random.default_rng(1)
    for shape, axis, J in [((2, 65, 32), -2, 1), ((2, 65, 32), -2, 3), ((2, 5, 64), -1, 1), ((2, 5, 64), -1, 6)]:
        x = rng.standard_normal(shape)
        y = O.haar_dwt_fwd(x, axis, J)
        assert y.shape == x.shape
        close((y ** 2).sum(), (x ** 2).sum(), rtol=1e-12)
        close(O.haar_dwt_bwd(y, axis, J), x, rtol=1e-12, atol=1e-12)
    # first full pair follows pywt 'haar': a=(x0+x1)/sqrt2, d=(x0-x1)/sqrt2, bands = [a | d]
    x = np.array([[1.0, 3.0, 2.0, 6.0]])
    close(O.haar_dwt_fwd(x, -1, 1), np.array([[4.0, 8.0, -2.0, -4.0]]) / np.sqrt(2))


# Known-answer vectors of the Haar DWT as PRINTED in PyWavelets' documentation (the library behind pytorch_wavelets' DWTForward, the
# reference's only DWT call: repl/dwt_experiments.py:56; `pywavelets>=1.9.0` in its pyproject.toml:28; not installed here):
#   "Discrete Wavelet Transform (DWT)", pywt.dwt:      >>> (cA, cD) = pywt.dwt([1, 2, 3, 4, 5, 6], 'db1')  is the 6-sample form of
#                                                       >>> cA, cD = pywt.dwt([1, 2, 3, 4], 'db1')
#                                                       cA = [2.12132034 4.94974747]   cD = [-0.70710678 -0.70710678]
#   pywt.wavedec:                                      >>> coeffs = pywt.wavedec([1, 2, 3, 4, 5, 6, 7, 8], 'db1', level=2)
#                                                       cA2 = [ 5. 13.]   cD2 = [-2. -2.]   cD1 = [-0.70710678 x 4]
# ('db1' is the Haar wavelet.)  They pin the pair convention (a = (x0 + x1)/sqrt2, d = (x0 - x1)/sqrt2: the SIGN of the detail band)
# and the band order of a multi-level transform ([cA_J, cD_J, ..., cD_1]) -- the two choices a from-scratch Haar can get wrong.
PYWT_DWT_DB1 = (np.array([1.0, 2.0, 3.0, 4.0]), np.array([2.12132034, 4.94974747]), np.array([-0.70710678, -0.70710678]))
PYWT_WAVEDEC_DB1_L2 = (np.arange(1.0, 9.0), np.array([5.0, 13.0]), np.array([-2.0, -2.0]), np.full(4, -0.70710678))


def test_haar_matches_pywavelets_documented_vectors():
    x, cA, cD = PYWT_DWT_DB1
    a, d = O.haar_level_fwd(x[None], -1)
    close(a[0], cA, rtol=0, atol=5e-9)
    close(d[0], cD, rtol=0, atol=5e-9)
    close(O.haar_dwt_fwd(x[None], -1, 1)[0], np.concatenate([cA, cD]), rtol=0, atol=5e-9)   # the mixer's band layout: [cA | cD]
    x, cA2, cD2, cD1 = PYWT_WAVEDEC_DB1_L2
    close(O.haar_dwt_fwd(x[None], -1, 2)[0], np.concatenate([cA2, cD2, cD1]), rtol=0, atol=5e-9)   # wavedec's coefficient order
    for mode in ("passthrough", "zero"):   # even lengths: the two conventions for an unpaired element never apply
        close(O.haar_dwt_fwd(x[None], -1, 2, mode)[0], np.concatenate([cA2, cD2, cD1]), rtol=0, atol=5e-9)


def test_haar_zero_mode_odd_length():
    """mode="zero" (the reference's DWTForward(..., mode="zero"), dwt_experiments.py:56): an odd signal is extended by one zero, so
    pywt returns ceil(L/2) + ceil(L/2) coefficients with cA_last = cD_last = x_last / sqrt2 (by the pair convention above on the pair
    (x_last, 0): derived, not a printed vector).  The shape-preserving mixer keeps every cA and all but that duplicate last cD."""
    rng = np.random.default_rng(7)
    x = rng.standard_normal((3, 65))
    cA, cD = O.haar_level_pywt_zero(x)
    assert cA.shape == cD.shape == (3, 33)
    close(cA[:, -1], x[:, -1] / np.sqrt(2), rtol=1e-15)
    close(cD[:, -1], cA[:, -1], rtol=1e-15)   # the coefficient the mixer drops is a copy of one it keeps
    y = O.haar_dwt_fwd(x, -1, 1, "zero")
    assert y.shape == x.shape
    close(y, np.concatenate([cA, cD[:, :-1]], axis=-1), rtol=1e-15)
    close(O.haar_level_pywt_zero(np.array([1.0, 2.0, 3.0]))[0], np.array([3.0, 3.0]) / np.sqrt(2), rtol=1e-15)
    close(O.haar_level_pywt_zero(np.array([1.0, 2.0, 3.0]))[1], np.array([-1.0, 3.0]) / np.sqrt(2), rtol=1e-15)
    # pass-through differs from it in exactly one slot (x_last vs x_last / sqrt2) ...
    yp = O.haar_dwt_fwd(x, -1, 1, "passthrough")
    diff = np.abs(yp - y)
    assert (diff[:, :32] == 0).all() and (diff[:, 33:] == 0).all()
    close(yp[:, 32], x[:, -1], rtol=1e-15)
    # ... and the backward is the exact adjoint in both modes, over several levels and along the token axis too
    for mode in ("passthrough", "zero"):
        for shape, axis, J in [((2, 65, 8), -2, 1), ((2, 65, 8), -2, 3), ((2, 7, 33), -1, 2)]:
            u, v = rng.standard_normal(shape), rng.standard_normal(shape)
            close((O.haar_dwt_fwd(u, axis, J, mode) * v).sum(), (u * O.haar_dwt_bwd(v, axis, J, mode)).sum(), rtol=1e-12)


def test_distill_loss_gradient_numeric():
    rng = np.random.default_rng(2)
    s, t = rng.standard_normal((3, 7)), rng.standard_normal((3, 7))
    lab = np.array([1, 0, 6])
    loss, d, _, _ = O.distill_loss_fwd_bwd(s, t, lab)
    num = np.zeros_like(s)
    for i in np.ndindex(*s.shape):
        sp = s.copy(); sp[i] += 1e-6
        sm = s.copy(); sm[i] -= 1e-6
        num[i] = (O.distill_loss_fwd_bwd(sp, t, lab)[0] - O.distill_loss_fwd_bwd(sm, t, lab)[0]) / 2e-6
    close(d, num, rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("name", ["model_tiny_mnist", "model_small_cut"])
def test_torch_cpu_port_train_step(name):
    """bench.py's cpu_baseline leg (oracle/spectre_torch_cpu.py: stock ATen ops + autograd + torch AdamW) against the
    reference's own logits / loss / gradients / post-AdamW weights."""
    import torch
    from oracle import spectre_torch_cpu as T
    d, cfg = load_model_fixture(name)
    sd = {k[3:]: v for k, v in d.items() if k.startswith("sd.")}
    st = T.TrainState(sd, dtype=torch.float64)
    grads = {}
    img, labels = torch.as_tensor(d["img"], dtype=torch.float64), torch.as_tensor(d["labels"]).long()
    logits, cls = T.forward(img, st.sd, cfg["num_encoders"], cfg["patch_size"], "permut")
    loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    close(logits.detach().numpy(), d["logits"], rtol=1e-8, atol=1e-9)
    close(cls.detach().numpy(), d["cls"], rtol=1e-8, atol=1e-9)
    close(loss.item(), d["loss"], rtol=1e-9)
    n = 0
    for k, v in d.items():
        if k.startswith("grad."):
            close(st.sd[k[5:]].grad.numpy(), v, rtol=1e-7, atol=1e-9)
            n += 1
    assert n == len(st.params)
    st.step(img, labels, cfg["num_encoders"], cfg["patch_size"], "permut")
    for k, v in d.items():
        if k.startswith("after."):
            close(st.sd[k[6:]].detach().numpy(), v, rtol=1e-7, atol=1e-9)


def test_torch_cpu_port_fft_mixer_matches_numpy_oracle():
    import torch
    from oracle import spectre_torch_cpu as T
    d, cfg = load_model_fixture("model_small_cut")
    sd = {k[3:]: v for k, v in d.items() if k.startswith("sd.") and "mix_layer" not in k}
    loss, logits, cls, grads = O.train_step(d["img"], d["labels"], sd, cfg["num_encoders"], cfg["patch_size"], "fft", np.float64)
    st = T.TrainState(sd, dtype=torch.float64)
    lg, _ = T.forward(torch.as_tensor(d["img"], dtype=torch.float64), st.sd, cfg["num_encoders"], cfg["patch_size"], "fft")
    torch.nn.functional.cross_entropy(lg, torch.as_tensor(d["labels"]).long()).backward()
    close(lg.detach().numpy(), logits, rtol=1e-8, atol=1e-9)
    for k, gk in grads.items():
        close(st.sd[k].grad.numpy(), gk, rtol=1e-6, atol=1e-9)


@pytest.fixture(scope="module")
def golden_hadamard():
    import os
    from conftest import GOLDEN
    return dict(np.load(os.path.join(GOLDEN, "hadamard.npz")))


@pytest.mark.parametrize("n", [2, 8, 64, 512])
def test_hadamard_helpers(golden_hadamard, n):
    """fwht / fwht_fast / hadamard_transform of reference hadamar.py, forward and backward."""
    g = golden_hadamard
    close(O.fwht(g[f"fwht.{n}.x"]), g[f"fwht.{n}.y"])
    close(O.fwht(g[f"fwht.{n}.dy"]), g[f"fwht.{n}.dx"])  # symmetric matrix: backward == forward
    close(O.fwht(g[f"fwht_raw.{n}.x"], normalize=False), g[f"fwht_raw.{n}.y"])
    close(O.fwht(g[f"hadamard_transform.{n}.x"]), g[f"hadamard_transform.{n}.y"])
    close(O.fwht_fast_fwd(g[f"fwht_fast.{n}.x"]), g[f"fwht_fast.{n}.y"])
    close(O.fwht_fast_bwd(g[f"fwht_fast.{n}.dy"]), g[f"fwht_fast.{n}.dx"])
    if n >= 8:  # SURVEY section 2 row 6 probe: fwht_fast is NOT fwht * sqrt(n) (different output ordering)
        assert np.abs(O.fwht_fast_fwd(g[f"fwht.{n}.x"]) - g[f"fwht.{n}.y"] * np.sqrt(n)).max() > 1e-3


@pytest.mark.parametrize("dim,blocks", [(48, 2), (64, 1), (100, 3)])
def test_learnable_hadamard(golden_hadamard, dim, blocks):
    g = golden_hadamard
    key = f"lh.{dim}.{blocks}"
    close(O.learnable_hadamard_fwd(g[key + ".x"], blocks), g[key + ".y"])
    close(O.learnable_hadamard_bwd(g[key + ".dy"], blocks), g[key + ".dx"])
    assert bool(g[key + ".param_grads_none"])
