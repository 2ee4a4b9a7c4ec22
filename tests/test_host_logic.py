"""CPU-side checks (no GPU compute): the C-ABI library loads and exports every symbol include/spv.h declares, the
module mirror keeps the reference's state_dict ABI, the host-side FFT harness passes, CPU tensors fail loudly."""
import copy
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_model_fixture


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    return True


def header_functions():
    src = open(os.path.join(ROOT, "include", "spv.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(spv_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built):
    from spectre_vit import _native
    lib = _native.load()
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/spv.h but not exported by libspv_hip.so"
        assert n in _native.SIGNATURES, f"{n} has no ctypes signature in spectre_vit/_native.py"
    assert set(_native.SIGNATURES) == set(names), set(_native.SIGNATURES) ^ set(names)
    assert lib.spv_version() == 1


def test_fft_core_host_harness(built):
    exe = os.path.join(ROOT, "tests", "cpu_harness", "_build", "fft_core_test")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr


@pytest.mark.parametrize("name", ["model_tiny_mnist", "model_small_cut"])
def test_state_dict_abi_matches_reference(name):
    """keys, shapes and dtypes of the reference's own state_dict (tests/golden) == ours; strict load works on CPU."""
    from spectre_vit.models.spectre.spectre import SpectreViT
    d, cfg = load_model_fixture(name)
    ref = {k[3:]: v for k, v in d.items() if k.startswith("sd.")}
    m = SpectreViT(**cfg)
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref.keys())  # same names in the same registration order
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(ref[k].shape), k
        assert str(v.dtype).replace("torch.", "") == str(ref[k].dtype), k
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in ref.items()}, strict=True)
    assert sum(p.numel() for p in m.parameters()) == sum(v.size for k, v in ref.items() if "grad." + k in d)


def test_clones_share_initial_weights_and_deepcopy():
    """_get_clones deep-copies one layer: all layers start identical incl. perms/signs."""
    from spectre_vit.models.spectre.spectre import SpectreViT
    torch.manual_seed(0)
    m = SpectreViT(img_size=16, patch_size=4, in_channels=3, num_classes=10, embed_dim=32, num_encoders=3, num_heads=2,
                   hidden_dim=48, dropout=0.1)
    l0, l2 = m.encoder_blocks.layers[0], m.encoder_blocks.layers[2]
    assert torch.equal(l0.mix_layer.perms, l2.mix_layer.perms) and torch.equal(l0.mix_layer.signs, l2.mix_layer.signs)
    assert torch.equal(l0.linear1.local_head[0].weight, l2.linear1.local_head[0].weight)
    m2 = copy.deepcopy(m)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def test_mixer_variants_have_no_permut_buffers():
    from spectre_vit.models.spectre.spectre import SpectreViT
    for mixer in ("fft", "dwt_embed", "dwt_token"):
        m = SpectreViT(img_size=16, patch_size=4, in_channels=3, num_classes=10, embed_dim=32, num_encoders=2, num_heads=2,
                       hidden_dim=48, dropout=0.0, mixer=mixer)
        assert not any("perms" in k for k in m.state_dict())
    with pytest.raises(ValueError):
        SpectreViT(img_size=16, patch_size=4, embed_dim=32, num_encoders=1, num_heads=2, hidden_dim=48, mixer="nope")


def test_cpu_tensors_fail_loudly(built):
    from spectre_vit.models.spectre.spectre import SpectreViT
    m = SpectreViT(img_size=16, patch_size=4, in_channels=3, num_classes=10, embed_dim=32, num_encoders=1, num_heads=2,
                   hidden_dim=48, dropout=0.0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.randn(2, 3, 16, 16))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "vit-spectre-experiments_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                assert "oracle" not in open(os.path.join(dp, f)).read().replace("oracle/", "").lower() or f == "__init__.py", f


def test_oracle_normalize_u8_matches_totensor_normalize():
    """train.py:102-112: ToTensor (HWC uint8 -> CHW float / 255) then Normalize(mean, std), restated with torch ops."""
    import numpy as np
    import torch
    from oracle import spectre_oracle as O
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(3, 8, 8, 3), dtype=np.uint8)
    mean, std = (0.5071, 0.4867, 0.4408), (0.2675, 0.2565, 0.2761)
    x = torch.from_numpy(img).permute(0, 3, 1, 2).double() / 255.0
    ref = (x - torch.tensor(mean, dtype=torch.float64).view(1, 3, 1, 1)) / torch.tensor(std, dtype=torch.float64).view(1, 3, 1, 1)
    np.testing.assert_allclose(O.normalize_u8(img, mean, std), ref.numpy(), rtol=0, atol=1e-14)


def test_c_abi_rejects_bad_arguments_before_any_launch(built):
    """Error contract of the C-ABI: a bad call returns non-zero and leaves a message for spv_last_error(); the
    ctypes wrapper raises RuntimeError with it.  Every call below fails argument validation on the host, so nothing is
    launched (this test runs without a GPU)."""
    from spectre_vit import _native
    BF16, F32 = 1, 0
    cases = [
        ("spv_gemm_nt", (0, 0, 0, 0, 0, 128, 64, 0, 0, 128, BF16, BF16, 0, 1, 0, 0), "empty"),
        ("spv_gemm_nt", (16, 16, 0, 16, 128, 128, 12, 12, 12, 128, BF16, BF16, 0, 1, 0, 0), "multiples of 8"),
        ("spv_gemm_nt", (16, 16, 0, 16, 128, 128, 64, 64, 64, 128, 7, BF16, 0, 1, 0, 0), "in_dtype"),
        ("spv_gemm_nt", (16, 16, 0, 16, 128, 128, 64, 64, 64, 128, BF16, BF16, 0, 4, 0, 0), "workspace"),
        ("spv_gemm_tn", (16, 16, 16, 100, 128, 64, 100, 128, 128, F32, 0, 1, 0, 0), "multiples of 8"),
        ("spv_spectre_tail_fwd", (16, 16, 16, 16, 16, 16, 16, 4, 64, 64, BF16, BF16, 1.5, 0, 0), "p_drop"),
        ("spv_spectre_tail_fwd", (16, 16, 16, 16, 16, 16, 16, 4, 64, 64, 9, BF16, 0.0, 0, 0), "dtype"),
        ("spv_add_layernorm_fwd", (16, 16, 16, 16, 16, 16, 16, 4, 64, 3, BF16, 0), "mode"),
        ("spv_attention_fwd", (16, 16, 16, 2, 8, 2, 512, BF16, 0.0, 0, 0), "head_dim"),
        ("spv_patchify", (16, 16, 0, 3, 32, 32, 4, 48, 0, BF16, 0), "bad shape"),
        ("spv_patchify_u8", (16, 0, 0, 16, 2, 3, 32, 32, 4, 48, 0, BF16, 0), "mean"),
        ("spv_weight_shadows", (16, 0, 16, 100, 64, 96, BF16, 0), "bad shape"),
        ("spv_dropout", (16, 16, 10, 1.0, 0, BF16, 0), "p="),
    ]
    for name, args, needle in cases:
        with pytest.raises(RuntimeError) as e:
            _native.call(name, *args)
        assert name.replace("_fwd", "") in str(e.value) or name in str(e.value), (name, str(e.value))
        assert needle in str(e.value), (name, needle, str(e.value))


def test_pmc_groups_fit_counter_slots():
    """tools/pmc_run.py splits a requested pass that over-subscribes a block's counter slots (round 1 lost a pass to three TA events
    on the two TA slots) and keeps FETCH_SIZE / WRITE_SIZE apart (3 + 2 of the 4 TCC slots)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_run
    g = pmc_run.fit_groups([["TA_TA_BUSY_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum"],
                            ["FETCH_SIZE", "WRITE_SIZE"], ["SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"]])
    assert g == [["TA_TA_BUSY_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum"], ["TA_DATA_STALLED_BY_TC_CYCLES_sum"], ["FETCH_SIZE"],
                 ["WRITE_SIZE"], ["SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"]]
    for grp in g:
        used = {}
        for c in grp:
            blk, n = pmc_run.block_cost(c)
            used[blk] = used.get(blk, 0) + n
        assert all(v <= pmc_run.SLOTS.get(b, 2) for b, v in used.items())


def test_config_presets_match_reference_config_files():
    """every spectre_vit/configs/<name>.py parses to exactly the values of the reference's file of the same name
    (tests/golden/configs.json: dumped from the reference's modules by tests/golden/make_golden.py)."""
    import json
    from spectre_vit.configs.parser import parse_config
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "configs.json")))
    assert set(ref) >= {"spectre_vit_cifar100", "spectre_vit_mnist", "spectre_branch", "vit_cifar100", "fnet_cifar100", "vit_mnist", "fnet_mnist"}
    for name, want in ref.items():
        if name == "default":
            continue
        got = vars(parse_config(f"spectre_vit/configs/{name}.py"))
        got = {k: (list(v) if isinstance(v, tuple) else v) for k, v in got.items()}
        assert got == want, (name, {k: (got.get(k), want.get(k)) for k in set(got) | set(want) if got.get(k) != want.get(k)})


def test_validation_batches_cover_every_sample_for_any_world_size():
    """harness validation: with world >= 3 the per-rank shard is shorter than val_batch_size; every sample must
    still be seen exactly once (short tail batch yielded), and training keeps drop_last."""
    import types
    import torch
    from spectre_vit.harness import SyntheticCifar
    c = types.SimpleNamespace(num_classes=10, in_channels=3, img_size=8)
    ds = SyntheticCifar(1024, c, torch.device("cpu"), seed=3)
    for world in (1, 2, 3, 4, 8):
        seen = 0
        for rank in range(world):
            for img, label in ds.batches(512, False, None, rank, world, drop_last=False):
                assert 0 < label.numel() <= 512 and img.shape[0] == label.numel()
                seen += label.numel()
        assert seen == 1024, (world, seen)
    assert sum(lab.numel() for _, lab in ds.batches(300, True, torch.Generator().manual_seed(0))) == 900  # training: drop_last
