"""Config parser / distillation-loss checks on CPU; the training harness itself on the GPU (f-1, f-2 of SURVEY 8f)."""
import os

import numpy as np
import pytest
import torch

from oracle import spectre_oracle as O


def test_parse_config_semantics():
    """reference configs/parser.py:5-27: base overrides child, `_base_` typo dropped, SimpleNamespace result"""
    from spectre_vit.configs.parser import parse_config
    c = parse_config("spectre_vit/configs/spectre_vit_cifar100.py")
    assert (c.img_size, c.patch_size, c.embed_dim, c.num_heads, c.hidden_dim, c.num_encoders) == (32, 4, 512, 16, 768, 4)
    assert (c.num_classes, c.dropout, c.adam_weight_decay, c.adam_betas, c.num_patches) == (100, 0.001, 0.01, (0.9, 0.999), 64)
    assert c.random_seed == 42 and c.learning_rate == 1e-3 and c.__base__ == "default.py"
    m = parse_config("spectre_vit/configs/spectre_vit_mnist.py")
    assert (m.img_size, m.embed_dim, m.num_heads, m.hidden_dim) == (28, 48, 8, 256)
    assert not hasattr(m, "random_seed") and not hasattr(m, "_base_")


def test_distillation_loss_matches_oracle():
    from spectre_vit.distillation import distillation_loss
    g = torch.Generator().manual_seed(3)
    s = torch.randn(6, 10, generator=g, dtype=torch.float64, requires_grad=True)
    t = torch.randn(6, 10, generator=g, dtype=torch.float64)
    y = torch.randint(0, 10, (6,), generator=g)
    loss, soft, ce = distillation_loss(s, t, y)
    loss.backward()
    ref, dref, soft_ref, ce_ref = O.distill_loss_fwd_bwd(s.detach().numpy(), t.numpy(), y.numpy())
    np.testing.assert_allclose(loss.item(), ref, rtol=1e-12)
    np.testing.assert_allclose(soft.item(), soft_ref, rtol=1e-12)
    np.testing.assert_allclose(ce.item(), ce_ref, rtol=1e-12)
    np.testing.assert_allclose(s.grad.numpy(), dref, rtol=1e-10, atol=1e-14)


def test_synthetic_teacher_contract():
    from spectre_vit.distillation import SyntheticTeacher
    t = SyntheticTeacher(100, 384, 3)
    logits, feats = t(torch.randn(4, 3, 64, 64), return_features=True)
    assert logits.shape == (4, 100) and feats.shape == (4, 384) and not logits.requires_grad


@pytest.mark.gpu
@pytest.mark.parametrize("distill", [False, True])
def test_harness_trains_and_checkpoints(tmp_path, distill):
    from spectre_vit.harness import train
    from spectre_vit.models.spectre.spectre import SpectreViT
    from spectre_vit.configs.parser import parse_config
    cfg = "spectre_vit/configs/spectre_vit_mnist.py"
    model, hist = train(cfg, mixer="permut", epochs=3, steps_per_epoch=12, batch_size=64, n_train=1024, n_val=256,
                        distill=distill, out_dir=str(tmp_path), log=lambda r: None)
    assert len(hist) == 3 and all(np.isfinite(h["Loss/Train"]) for h in hist)
    assert hist[-1]["Loss/Train"] < hist[0]["Loss/Train"], hist
    ck = torch.load(os.path.join(tmp_path, "model_best.pt"), weights_only=True)
    c = parse_config(cfg)
    m2 = SpectreViT(img_size=c.img_size, patch_size=c.patch_size, in_channels=c.in_channels, num_classes=c.num_classes,
                    embed_dim=c.embed_dim, num_encoders=c.num_encoders, num_heads=c.num_heads, hidden_dim=c.hidden_dim,
                    dropout=c.dropout)
    m2.load_state_dict(ck, strict=True)  # export.py:59 loads with strict=True
    assert os.path.exists(os.path.join(tmp_path, "scalars.jsonl"))


@pytest.mark.gpu
def test_harness_uint8_input_matches_float_input(tmp_path):
    """the loader's uint8 batches fed straight to the model give the same training curve as host-normalised
    float batches (same seeds; the two paths differ by one fp32 rounding in the normalisation)."""
    from spectre_vit.harness import train
    cfg = "spectre_vit/configs/spectre_vit_mnist.py"
    kw = dict(mixer="fft", epochs=2, steps_per_epoch=8, batch_size=64, n_train=512, n_val=128, use_amp=False, log=lambda r: None)
    _, h_float = train(cfg, out_dir=str(tmp_path / "f"), **kw)
    _, h_u8 = train(cfg, out_dir=str(tmp_path / "u"), uint8_input=True, **kw)
    for a, b in zip(h_float, h_u8):
        assert abs(a["Loss/Train"] - b["Loss/Train"]) < 2e-3 * abs(a["Loss/Train"]), (a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_first_steps_loss_curve_vs_oracle(fused):
    """the first optimisation steps on a fixed synthetic batch follow the oracle's loss curve (fp32).
    fused=True: torch's fused AdamW updates parameters WITHOUT bumping their version counters -- the transposed / bf16
    weight copies the GEMMs read must follow anyway."""
    from conftest import load_model_fixture
    from spectre_vit.models.spectre.spectre import SpectreViT
    d, cfg = load_model_fixture("model_small_cut")
    sd = {k[3:]: v for k, v in d.items() if k.startswith("sd.")}
    dev = torch.device("cuda:0")
    m = SpectreViT(**cfg).to(dev)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01, fused=fused)
    img, labels = torch.from_numpy(d["img"]).to(dev), torch.from_numpy(d["labels"]).to(dev)
    sd64 = {k: np.asarray(v, np.float64) if v.dtype.kind == "f" else v for k, v in sd.items()}
    state = {}
    for step in range(1, 7):
        loss = torch.nn.functional.cross_entropy(m(img), labels)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        ref_loss, _, _, grads = O.train_step(d["img"], d["labels"], sd64, cfg["num_encoders"], cfg["patch_size"], "permut", np.float64)
        for k, g in grads.items():
            mv = state.setdefault(k, [np.zeros_like(g), np.zeros_like(g)])
            sd64[k], mv[0], mv[1] = O.adamw_step(sd64[k], g, mv[0], mv[1], step)
        assert abs(loss.item() - ref_loss) < 2e-3 * abs(ref_loss), (step, loss.item(), ref_loss)


@pytest.mark.gpu
@pytest.mark.parametrize("mixer", ["fft", "permut"])
def test_bf16_training_uses_updated_weights(mixer):
    """After optimizer steps (fused AdamW: no version bump) the bf16 forward must equal that of a fresh model loaded with
    the updated state_dict, in train and in eval mode; a frozen bf16 shadow of the weights would keep the old output."""
    from spectre_vit.models.spectre.spectre import SpectreViT
    cfg = dict(img_size=16, patch_size=4, in_channels=3, num_classes=16, embed_dim=64, num_encoders=2, num_heads=4, hidden_dim=96,
               dropout=0.0, mixer=mixer)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = SpectreViT(**cfg).to(dev).train()
    opt = torch.optim.AdamW(m.parameters(), lr=5e-2, fused=True)  # large steps: stale weights would be obvious
    x = torch.randn(8, 3, 16, 16, device=dev)
    y = torch.randint(0, 16, (8,), device=dev)

    def fwd(model):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return model(x).float()

    out0 = fwd(m).detach().clone()
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(fwd(m), y).backward()
        opt.step()
    fresh = SpectreViT(**cfg).to(dev)
    fresh.load_state_dict(m.state_dict())
    for mode in ("train", "eval"):
        getattr(m, mode)(); getattr(fresh, mode)()
        with torch.no_grad():
            a, b = fwd(m), fwd(fresh)
        assert torch.equal(a, b), (mode, (a - b).abs().max().item())
    assert (fwd(m).detach() - out0).abs().max().item() > 0.05  # and the steps did move the output
    # inference-time cache: an optimizer step between two no_grad forwards invalidates it
    m.eval()
    with torch.no_grad():
        before = fwd(m)
    m.train()
    opt.zero_grad(set_to_none=True)
    torch.nn.functional.cross_entropy(fwd(m), y).backward()
    opt.step()
    m.eval(); fresh.load_state_dict(m.state_dict()); fresh.eval()
    with torch.no_grad():
        after, ref = fwd(m), fwd(fresh)
    assert torch.equal(after, ref) and not torch.equal(after, before)


@pytest.mark.gpu
def test_config1_mnist_full_synthetic_epoch(tmp_path):
    """BASELINE config 1 as written, minus the CPU: one FULL epoch of configs/spectre_vit_mnist.py (Tiny: img 28, P 4, E 48, H 8,
    F 256, L 4; its own batch_size = 8 -> 7500 steps) over 60 000 synthetic MNIST-sized images with the FFT mixer, then the
    validation pass and the checkpoint.  (The reference falls back to the CPU, train.py:41; this package runs on the GPU only --
    DESIGN.md section 1.)"""
    from spectre_vit.configs.parser import parse_config
    from spectre_vit.harness import train
    cfg = "spectre_vit/configs/spectre_vit_mnist.py"
    c = parse_config(cfg)
    model, hist = train(cfg, mixer="fft", epochs=1, n_train=60000, n_val=2048, out_dir=str(tmp_path), log=lambda r: None)
    assert len(hist) == 1 and hist[0]["steps"] == 60000 // c.batch_size
    assert np.isfinite(hist[0]["Loss/Train"]) and np.isfinite(hist[0]["Loss/Validation"])
    assert hist[0]["val_samples"] == 2048
    # class-conditional synthetic images are learnable: one epoch must beat chance (1 %) by a wide margin
    assert hist[0]["Accuracy/Validation"] > 0.2, hist
    assert os.path.exists(os.path.join(tmp_path, "model_best.pt"))


@pytest.mark.gpu
@pytest.mark.parametrize("capturable", [False, True])
def test_fused_adamw_matches_torch_adamw_and_oracle(capturable):
    """spectre_vit.optim.FusedAdamW (one HIP launch per step) against torch.optim.AdamW on identical parameters / gradients over
    5 steps (tensors of awkward sizes: unaligned tails, a scalar), and its first step against the oracle's adamw_step
    (reference optimizer call: train.py:199-201)."""
    from oracle import spectre_oracle as O
    from spectre_vit.optim import FusedAdamW
    torch.manual_seed(0)
    shapes = [(512, 768), (768,), (3,), (1,), (100, 512), (65, 512), (2049,), (4095,)]
    a = [torch.randn(s, device="cuda").requires_grad_(True) for s in shapes]
    b = [t.detach().clone().requires_grad_(True) for t in a]
    oa = FusedAdamW(a, lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01, capturable=capturable)
    ob = torch.optim.AdamW(b, lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01)
    p0 = a[0].detach().cpu().numpy().astype(np.float64)
    for step in range(5):
        grads = [torch.randn(s, device="cuda") * (0.1 + step) for s in shapes]
        for t, g in zip(a, grads):
            t.grad = g.clone()
        for t, g in zip(b, grads):
            t.grad = g.clone()
        oa.step()
        ob.step()
        if step == 0:
            ref, _, _ = O.adamw_step(p0, grads[0].cpu().numpy().astype(np.float64), np.zeros_like(p0), np.zeros_like(p0), 1)
            assert np.abs(a[0].detach().cpu().numpy() - ref).max() <= 2e-6 * np.abs(ref).max()
        for x, y in zip(a, b):
            assert torch.allclose(x, y, rtol=2e-6, atol=2e-7), (step, x.shape, (x - y).abs().max().item())
    sa, sb = oa.state_dict()["state"], ob.state_dict()["state"]
    assert set(sa[0].keys()) == set(sb[0].keys()) == {"step", "exp_avg", "exp_avg_sq"} and float(sa[0]["step"]) == 5.0
    assert torch.allclose(sa[0]["exp_avg_sq"], sb[0]["exp_avg_sq"], rtol=1e-5, atol=1e-8)


@pytest.mark.gpu
def test_graphed_train_step_matches_eager_steps():
    """spectre_vit.graph.GraphedTrainStep: three replays == three eager steps (same kernels, same order: bit for bit with dropout
    off), and with dropout on every replay draws different masks (the device-side seed word advances inside the graph)."""
    from spectre_vit.graph import GraphedTrainStep
    from spectre_vit.models.spectre.spectre import SpectreViT
    from spectre_vit.optim import FusedAdamW
    cfg = dict(img_size=32, patch_size=4, in_channels=3, num_classes=100, embed_dim=512, num_encoders=2, num_heads=16, hidden_dim=768,
               activation="gelu")
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    img = torch.randn(64, 3, 32, 32, generator=g).to(dev)
    labels = torch.randint(0, 100, (64,), generator=g).to(dev)
    crit = torch.nn.CrossEntropyLoss()

    def make(dropout, capturable):
        torch.manual_seed(11)
        m = SpectreViT(**cfg, dropout=dropout, mixer="fft").to(dev).train()
        return m, FusedAdamW(m.parameters(), lr=1e-3, weight_decay=0.01, capturable=capturable)

    m1, o1 = make(0.0, True)  # (capturable on both sides: the bias corrections then come from the same device-side powf)
    eager_losses = []
    for _ in range(3 + 3):  # GraphedTrainStep warms up with 3 real steps before it captures
        o1.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = m1(img)
        loss = crit(out, labels)
        loss.backward()
        o1.step()
        eager_losses.append(loss.item())
    from spectre_vit import hip_ops
    m2, o2 = make(0.0, True)
    keep = hip_ops._WGRAD_HOLD
    hip_ops._WGRAD_HOLD = False   # the eager loop above has no gradient sinks, so its weight gradients run one launch each: same here
    try:
        step = GraphedTrainStep(m2, o2, crit, img, labels, warmup=3)   # 3 warm-up steps + the captured (not executed) one
        graph_losses = [step().item() for _ in range(3)]
        step.close()
    finally:
        hip_ops._WGRAD_HOLD = keep
    # the default: the layers' weight gradients held back and computed by one batched launch inside the graph (other K-slices: the
    # same sums to fp32 re-association)
    m4, o4 = make(0.0, True)
    step4 = GraphedTrainStep(m4, o4, crit, img, labels, warmup=3)
    batched_losses = [step4().item() for _ in range(3)]
    step4.close()
    assert all(abs(a - b) <= 5e-4 * abs(b) for a, b in zip(batched_losses, eager_losses[3:6])), (batched_losses, eager_losses)
    # the first replays reproduce the eager steps bit for bit; later ones may drift in the last bits (bf16 rounding of weights that
    # differ by one ulp after the device-side bias correction), so the bound is relative
    assert graph_losses[0] == eager_losses[3], (graph_losses, eager_losses)
    # (measured: 0, 6e-5, 1.4e-4 relative over the three replays -- a last-bit difference growing ~2.3x per step at lr 1e-3)
    assert all(abs(a - b) <= 5e-4 * abs(b) for a, b in zip(graph_losses, eager_losses[3:6])), (graph_losses, eager_losses)
    for (k, p), q in zip(m1.named_parameters(), m2.parameters()):
        # Adam's update m / (sqrt(v) + eps) is +-lr whatever the gradient's size, so one-ulp differences in tiny gradients move a
        # weight by a fraction of an lr step (1e-3): the bound is 0.2 lr steps, not machine epsilon
        assert torch.allclose(p, q, rtol=1e-3, atol=5e-4), (k, (p - q).abs().max().item())   # measured up to 2.4e-4 after six steps
    # dropout on: consecutive replays on the same batch and (frozen) weights differ only through the masks
    m3, o3 = make(0.3, True)
    for grp in o3.param_groups:
        grp["lr"] = 0.0
        grp["weight_decay"] = 0.0
    step3 = GraphedTrainStep(m3, o3, crit, img, labels, warmup=1)
    ls = [step3().item() for _ in range(4)]
    step3.close()
    assert len(set(ls)) == 4, ls


def _small_graph_parts(dropout=0.0, layers=2, batch=64):
    from spectre_vit.models.spectre.spectre import SpectreViT
    from spectre_vit.optim import FusedAdamW
    cfg = dict(img_size=32, patch_size=4, in_channels=3, num_classes=100, embed_dim=512, num_encoders=layers, num_heads=16,
               hidden_dim=768, activation="gelu")
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    img = torch.randn(batch, 3, 32, 32, generator=g).to(dev)
    labels = torch.randint(0, 100, (batch,), generator=g).to(dev)
    torch.manual_seed(11)
    m = SpectreViT(**cfg, dropout=dropout, mixer="fft").to(dev).train()
    opt = FusedAdamW(m.parameters(), lr=1e-3, weight_decay=0.01, capturable=True, static_grads=True)
    return m, opt, img, labels


@pytest.mark.gpu
def test_capture_refuses_gradients_that_moved():
    """The fault of round 2 (a replayed graph / the optimizer's device pointer table holding addresses of gradients that
    zero_grad(set_to_none=True) frees each step) must be a RuntimeError at capture time: FusedAdamW.step() inside a capture with a
    gradient that is not where its table says raises instead of uploading a table mid-capture (optim.py, _table)."""
    from spectre_vit.optim import FusedAdamW
    dev = torch.device("cuda:0")
    ps = [torch.randn(300, device=dev, requires_grad=True), torch.randn(7, 5, device=dev, requires_grad=True)]
    opt = FusedAdamW(ps, lr=1e-3, capturable=True)
    for p in ps:
        p.grad = torch.randn_like(p)
    opt.step()
    opt.step()   # same addresses twice: the table is settled
    keep = [p.grad for p in ps]
    opt.zero_grad(set_to_none=True)
    for p in ps:
        p.grad = torch.randn_like(p)   # fresh tensors while the old ones are still alive: new addresses
    assert all(p.grad.data_ptr() != k.data_ptr() for p, k in zip(ps, keep))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with pytest.raises(RuntimeError, match="moved while a HIP graph is being captured"):
        with torch.cuda.graph(graph):
            opt.step()
    torch.cuda.synchronize()
    opt.step()   # outside a capture the table is simply rebuilt
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_static_grads_notices_a_swapped_gradient():
    """static_grads=True skips the state walk once the table is settled, but compares every gradient address every step: a caller who
    swaps a .grad gets the update of THAT tensor (round 2 re-verified the pointers only every 64th step: 63 silent writes through
    stale pointers)."""
    from spectre_vit.optim import FusedAdamW
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    a = [torch.randn(1000, device=dev, requires_grad=True), torch.randn(33, device=dev, requires_grad=True)]
    b = [t.detach().clone().requires_grad_(True) for t in a]
    oa = FusedAdamW(a, lr=1e-2, static_grads=True)
    ob = torch.optim.AdamW(b, lr=1e-2)
    fixed = [torch.randn_like(t) for t in a]
    for step in range(6):
        if step < 4:   # fixed addresses: after two look-ups the fast path serves the step
            for t, g in zip(a, fixed):
                g.normal_()
                t.grad = g
        else:          # the caller swaps in fresh tensors
            for t in a:
                t.grad = torch.randn_like(t)
        for t, u in zip(a, b):
            u.grad = t.grad.clone()
        oa.step()
        ob.step()
        if step == 3:
            assert oa._tables[0]["static"]
        for x, y in zip(a, b):
            assert torch.allclose(x, y, rtol=2e-6, atol=2e-7), (step, (x - y).abs().max().item())


@pytest.mark.gpu
def test_fused_adamw_load_state_dict_after_stepping():
    """In-place resume / rollback: load_state_dict() on an optimizer that has already stepped must continue from the LOADED moments and
    step count (the cached device table pointed at the orphaned old moment tensors and kept its own count)."""
    from spectre_vit.optim import FusedAdamW
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    for capturable in (False, True):
        a = [torch.randn(513, device=dev, requires_grad=True)]
        b = [a[0].detach().clone().requires_grad_(True)]
        oa, ob = FusedAdamW(a, lr=1e-2, capturable=capturable), torch.optim.AdamW(b, lr=1e-2)
        grads = [torch.randn(513, device=dev) for _ in range(6)]
        snap = None
        for i, g in enumerate(grads):
            a[0].grad, b[0].grad = g.clone(), g.clone()
            oa.step(); ob.step()
            if i == 1:
                snap = (a[0].detach().clone(), oa.state_dict(), b[0].detach().clone(), ob.state_dict())
                import copy
                snap = copy.deepcopy(snap)
        # roll both back to the state after step 2 and replay steps 3..4
        with torch.no_grad():
            a[0].copy_(snap[0]); b[0].copy_(snap[2])
        oa.load_state_dict(snap[1]); ob.load_state_dict(snap[3])
        for g in grads[2:4]:
            a[0].grad, b[0].grad = g.clone(), g.clone()
            oa.step(); ob.step()
        assert float(oa.state_dict()["state"][0]["step"]) == 4.0
        # (lr 1e-2: an update computed from the orphaned moments or the old step count is off by ~1e-2; device-side powf in the
        # capturable bias correction accounts for the last-bit slack)
        assert torch.allclose(a[0], b[0], rtol=2e-6, atol=2e-6), (capturable, (a[0] - b[0]).abs().max().item())


@pytest.mark.gpu
def test_eval_after_graph_replays_sees_the_new_weights():
    """replays update the weights through raw pointers (no version bump, no optimizer hook), so the inference-time cache of
    bf16 weight copies must be invalidated by the step itself: eval logits after replays == the logits of a freshly built model."""
    from spectre_vit.graph import GraphedTrainStep
    from spectre_vit.models.spectre.spectre import SpectreViT
    m, opt, img, labels = _small_graph_parts()
    for grp in opt.param_groups:
        grp["lr"] = 5e-2   # large steps: stale shadows would be obvious
    step = GraphedTrainStep(m, opt, torch.nn.CrossEntropyLoss(), img, labels, warmup=1)

    def ev(model):
        model.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(img).float()
        model.train()
        return out

    before = ev(m)
    for _ in range(3):
        step()
    after = ev(m)
    step.close()
    fresh = SpectreViT(img_size=32, patch_size=4, in_channels=3, num_classes=100, embed_dim=512, num_encoders=2, num_heads=16,
                       hidden_dim=768, activation="gelu", dropout=0.0, mixer="fft").to(img.device)
    fresh.load_state_dict(m.state_dict())
    assert torch.equal(after, ev(fresh))
    assert (after - before).abs().max().item() > 0.05


@pytest.mark.gpu
def test_one_live_graphed_step_owns_the_seed_word():
    """the dropout seed word is one device symbol.  A second live step is refused; rebinding `step = GraphedTrainStep(...)`
    after close() works, and the OLD object's __del__ running later must not clear the new object's word (every replay would draw the
    same masks)."""
    import gc
    from spectre_vit.graph import GraphedTrainStep
    m, opt, img, labels = _small_graph_parts(dropout=0.3)
    for grp in opt.param_groups:
        grp["lr"] = 0.0
        grp["weight_decay"] = 0.0
    crit = torch.nn.CrossEntropyLoss()
    old = GraphedTrainStep(m, opt, crit, img, labels, warmup=1)
    with pytest.raises(RuntimeError, match="already live"):
        GraphedTrainStep(m, opt, crit, img, labels, warmup=1)
    old()
    old.close()
    new = GraphedTrainStep(m, opt, crit, img, labels, warmup=1)
    del old            # the closed object's destructor runs now -- after `new` registered its own word
    gc.collect()
    ls = [new().item() for _ in range(4)]
    new.close()
    assert len(set(ls)) == 4, ls   # frozen weights, same batch: the losses differ only through fresh masks
    with pytest.raises(RuntimeError, match="closed"):
        new()


@pytest.mark.gpu
@pytest.mark.parametrize("in_graph", [False, True])
def test_side_stream_batch_with_a_fresh_allocator(in_graph):
    """the batched weight gradients start on the side stream while the patch embedding's backward allocates on the
    main stream; the fold partials the batch's reduce reads had no owner once the flush returned.  With every tensor of the side
    launches kept until the join, the default path (sinks + batch + side stream) equals the one-by-one path (SPV_WGRAD_BATCH off) from a
    freshly emptied allocator, eagerly and inside a capture."""
    from spectre_vit import hip_ops
    from spectre_vit.dp import GradReducer
    from spectre_vit.graph import GraphedTrainStep
    crit = torch.nn.CrossEntropyLoss()

    def grads_of(hold):
        m, opt, img, labels = _small_graph_parts(layers=3, batch=96)   # (same seed: identical weights and batch every time)
        for grp in opt.param_groups:
            grp["lr"] = 0.0
            grp["weight_decay"] = 0.0
        keep = hip_ops._WGRAD_HOLD
        hip_ops._WGRAD_HOLD = hold
        try:
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            if in_graph:
                step = GraphedTrainStep(m, opt, crit, img, labels, warmup=1)
                step()
                torch.cuda.synchronize()
                out = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
                step.close()
                return out
            red = GradReducer(m, always=True)
            red.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = crit(m(img), labels)
            loss.backward()
            torch.cuda.synchronize()
            return {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        finally:
            hip_ops._WGRAD_HOLD = keep

    before = hip_ops.PATH_COUNTS["wgrad_side_start"]
    batched = grads_of(True)
    assert hip_ops.PATH_COUNTS["wgrad_side_start"] > before
    single = grads_of(False)
    for k in batched:
        ref = single[k]
        err = (batched[k] - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)
        assert err <= 2e-5, (k, err)   # other K-slices: the same sums to fp32 re-association; a clobbered partial would be O(1)


@pytest.mark.gpu
@pytest.mark.parametrize("uint8_input", [False, True])
def test_harness_graph_mode_follows_the_eager_loop(tmp_path, uint8_input):
    """harness.train(graph=True): the same run with the step replayed from a HIP graph and the one-launch AdamW -- loss curve,
    accuracies and checkpoint as the eager loop produces them (same seeds and batches; the first step of the graph run is the
    warm-up step, a real training step).  The two runs differ by the optimizer kernel (2e-6 parity) and the batched weight
    gradients (fp32 re-association): bf16 re-rounding turns that into ~1e-3 on the epoch loss."""
    from spectre_vit.harness import train
    cfg = "spectre_vit/configs/spectre_vit_mnist.py"
    kw = dict(mixer="fft", epochs=2, steps_per_epoch=10, batch_size=64, n_train=1024, n_val=256, log=lambda r: None, uint8_input=uint8_input)
    _, h_eager = train(cfg, out_dir=str(tmp_path / "e"), **kw)
    _, h_graph = train(cfg, out_dir=str(tmp_path / "g"), graph=True, **kw)
    for a, b in zip(h_eager, h_graph):
        assert a["steps"] == b["steps"] == 10
        assert abs(a["Loss/Train"] - b["Loss/Train"]) < 1e-2 * abs(a["Loss/Train"]), (a, b)
        assert abs(a["Loss/Validation"] - b["Loss/Validation"]) < 2e-2 * abs(a["Loss/Validation"]), (a, b)
    assert h_graph[-1]["Loss/Train"] < h_graph[0]["Loss/Train"]
    assert os.path.exists(os.path.join(tmp_path, "g", "model_best.pt"))
    with pytest.raises(ValueError, match="distillation"):
        train(cfg, graph=True, distill=True, out_dir=str(tmp_path / "x"), **kw)
