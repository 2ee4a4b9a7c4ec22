"""torch.autograd Functions over the C-ABI of libspv_hip.so.

Host-side mirror of the arithmetic the reference modules issue as stock ATen ops; every Function
borrows ``data_ptr()``s, launches on torch's current HIP stream and never synchronises.  Tensors must be
on a HIP device: a CPU tensor raises (there is deliberately no CPU / eager-PyTorch fallback).
"""
from __future__ import annotations

import ctypes
import os
import warnings
import weakref

import torch

from . import _native

F32, BF16 = 0, 1
_DT = {torch.float32: F32, torch.bfloat16: BF16}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """raw HIP handle of torch's current stream on the current device (the C getter: torch.cuda.current_stream() builds a Python
    Stream object per call -- 10 us, 38 times per training step)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"libspv_hip kernels take float32 or bfloat16 tensors, got {t.dtype}") from None


def _require_gpu(*tensors):
    """every tensor on a HIP device, and on the CURRENT one: kernels are launched on torch.cuda.current_stream(), which belongs
    to the current device -- raw pointers of another GPU on that stream would fault or run unordered."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("Spectre-ViT HIP kernels need tensors on an AMD GPU (cuda/HIP device); "
                               "there is no CPU fallback in this package")
        if t.device.index != torch.cuda.current_device():
            raise RuntimeError(f"tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()}: call "
                               "torch.cuda.set_device(local_rank) (or use `with torch.cuda.device(...)`) before the model runs")


_warned_fp16 = False


def compute_dtype(x: torch.Tensor) -> torch.dtype:
    """dtype the kernels run in for input x: the autocast dtype when autocast is on (bf16; an fp16 autocast
    region -- spectre_vit/repl/train.py:219 -- is served in bf16, same speed and no loss scaling issues),
    otherwise x's own dtype (fp32 parity runs, or bf16 activations)."""
    global _warned_fp16
    if torch.is_autocast_enabled("cuda"):
        dt = torch.get_autocast_dtype("cuda")
        if dt == torch.float16:
            if not _warned_fp16:
                warnings.warn("spectre_vit (MI355X): fp16 autocast is served by the bf16 kernels")
                _warned_fp16 = True
            return torch.bfloat16
        if dt == torch.bfloat16:
            return torch.bfloat16
    if x.dtype in (torch.float32, torch.bfloat16):
        return x.dtype
    return torch.float32


def cast(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """differentiable dtype cast through spv_cast (no-op when already `dtype`)."""
    if x.dtype == dtype:
        return x
    return _Cast.apply(x, dtype)


def _raw_cast(x, dtype):
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    _native.call("spv_cast", _p(x), _dt(x), _p(out), _DT[dtype], x.numel(), _stream())
    return out


class _Cast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        _require_gpu(x)
        ctx.src = x.dtype
        return _raw_cast(x, dtype)

    @staticmethod
    def backward(ctx, g):
        return _raw_cast(g, ctx.src), None


# ------------------------------------------------------------------------------------------------
# weight shadows: compute-dtype copy of W and of W^T, rebuilt only when the parameter changes
# ------------------------------------------------------------------------------------------------
class _ShadowCache:
    """(weight tensor, dtype) -> (W in dtype, W^T in dtype, padded to 8 columns).

    While a weight is being trained (grad mode on, requires_grad) the shadows are rebuilt at every forward: optimizers may
    update parameters without touching the tensor's version counter -- ``torch.optim.AdamW(fused=True)`` does exactly that --
    so no cheap test can prove a cached copy current, and a stale copy would silently freeze the layer.  Outside training
    (eval / no_grad inference loops) entries are reused, validated by a weak reference to the parameter (ids and addresses
    are recycled once a tensor dies), its version counter and the global optimizer-step epoch."""

    def __init__(self):
        self._d = {}
        self._fresh = {}
        self.epoch = 0

    def get(self, w: torch.Tensor, dtype: torch.dtype):
        key = (id(w), dtype)
        ent = self._d.get(key)
        ver = (w.data_ptr(), w._version, tuple(w.shape), self.epoch)
        training = torch.is_grad_enabled() and w.requires_grad
        if not training and ent is not None and ent[0]() is w and ent[1] == ver:
            return ent[2], ent[3]
        fresh = self._fresh.pop(key, None)
        # rebuilt for THIS forward by refresh_weight_shadows (one launch for all weights); void if an optimizer has stepped since
        if fresh is not None and fresh[0]() is w and fresh[3] == (w._version, self.epoch):
            return fresh[1], fresh[2]
        n, k = w.shape
        wd = w.detach()
        wc = wd if dtype == torch.float32 else torch.empty((n, k), dtype=dtype, device=w.device)
        ldt = (n + 7) // 8 * 8
        wt = torch.empty((k, ldt), dtype=dtype, device=w.device)
        _native.call("spv_weight_shadows", _p(wd), 0 if wc is wd else _p(wc), _p(wt), n, k, ldt, _DT[dtype], _stream())
        if len(self._d) > 1024:
            self._d = {kk: e for kk, e in self._d.items() if e[0]() is not None}
        self._d[key] = (weakref.ref(w), ver, wc, wt)
        return wc, wt


_shadows = _ShadowCache()


class ShadowSet:
    """Persistent bf16 (W, W^T) copies of a fixed list of fp32 nn.Linear weights, rebuilt by ONE spv_weight_shadows_multi launch
    per training forward (eight 4.9-us launches and sixteen allocations per step otherwise).  The copies are handed to the layers
    through _ShadowCache.get, which consumes them once per weight and forward."""

    def __init__(self, weights, dtype=torch.bfloat16):
        self.weights = [weakref.ref(w) for w in weights]
        self.dtype = dtype
        self.key = tuple((w.data_ptr(), tuple(w.shape)) for w in weights)
        dev = weights[0].device
        rows, tt, tx, ty = [], [], [], []
        self.bufs = []
        for i, w in enumerate(weights):
            n, k = w.shape
            ld = (n + 7) // 8 * 8
            wc = torch.empty((n, k), dtype=dtype, device=dev)
            wt = torch.empty((k, ld), dtype=dtype, device=dev)
            self.bufs.append((wc, wt))
            rows += [w.data_ptr(), wc.data_ptr(), wt.data_ptr(), n | (k << 32), ld]  # {src, plain, tr, (rows, cols), (ld, pad)}
            for by in range((ld + 31) // 32):
                for bx in range((k + 63) // 64):
                    tt.append(i)
                    tx.append(bx)
                    ty.append(by)
        self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
        self.tt = torch.tensor(tt, dtype=torch.int32).to(dev)
        self.tx = torch.tensor(tx, dtype=torch.int32).to(dev)
        self.ty = torch.tensor(ty, dtype=torch.int32).to(dev)
        self.ntiles = len(tt)

    def refresh(self):
        _native.call("spv_weight_shadows_multi", self.table.data_ptr(), self.tt.data_ptr(), self.tx.data_ptr(), self.ty.data_ptr(),
                     self.ntiles, _DT[self.dtype], _stream())
        for wr, (wc, wt) in zip(self.weights, self.bufs):
            w = wr()
            if w is not None:
                _shadows._fresh[(id(w), self.dtype)] = (wr, wc, wt, (w._version, _shadows.epoch))


def refresh_weight_shadows(module, weights_fn):
    """called at the top of a model's bf16 training forward: module._spv_shadow_set is (re)built when a weight moved or changed shape"""
    weights = weights_fn()
    if not weights:
        return
    ss = getattr(module, "_spv_shadow_set", None)
    key = tuple((w.data_ptr(), tuple(w.shape)) for w in weights)
    if ss is None or ss.key != key:
        if torch.cuda.is_current_stream_capturing():
            return  # (tables cannot be uploaded inside a capture: the per-weight path serves this forward)
        ss = ShadowSet(weights)
        object.__setattr__(module, "_spv_shadow_set", ss)
    ss.refresh()


def invalidate_weight_shadows(*_args, **_kwargs):
    """Drop every cached bf16 / transposed weight copy (needed only after updating weights in place, outside autograd's
    view, between two no_grad forwards).  Registered as a global optimizer post-step hook."""
    _shadows.epoch += 1


from torch.optim.optimizer import register_optimizer_step_post_hook as _register_post_step  # noqa: E402

_register_post_step(invalidate_weight_shadows)  # any torch optimizer's step() invalidates the inference-time cache


# ------------------------------------------------------------------------------------------------
# live kernel timing for bench.py's roofline block: HIP events recorded on the launch stream around EVERY C-ABI entry
# point that launches (hooked in _native.call; torch.cuda.Event records on torch's current stream == the stream we launch on)
# ------------------------------------------------------------------------------------------------
PEAK_TFLOPS = {BF16: 2500.0, F32: 157.3}  # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def _es(dt):
    return 2 if dt == BF16 else 4


# entry point -> f(ints) -> (label, shape, dtype code, bound, algorithmic work per call: flops (mfma) or compulsory bytes (hbm)).
# ints are the integer arguments of the C-ABI call in header order (include/spv.h).
_WORK_MODELS = {
    "spv_small_sl_fwd": lambda i: ("head_fwd", i[2:5], i[5], "mfma", 2.0 * i[2] * i[3] * i[4]),
    "spv_small_sl_bwd": lambda i: ("head_bwd", i[0:3], i[3], "mfma", 4.0 * i[0] * i[1] * i[2]),
    "spv_cross_entropy_fwd": lambda i: ("cross_entropy_fwd", i[0:2], F32, "hbm", i[0] * i[1] * 4.0),
    "spv_cross_entropy_bwd": lambda i: ("cross_entropy_bwd", i[0:2], F32, "hbm", i[0] * i[1] * 8.0),
    "spv_gemm_nt": lambda i: ("gemm_acc" if i[8] else "gemm", i[0:3], i[6], "mfma", 2.0 * i[0] * i[1] * i[2]),
    "spv_gemm_nt_grouped_rows": lambda i: ("gemm_grouped_rows", i[0:3], i[6], "mfma", 2.0 * i[0] * i[1] * i[2]),
    "spv_gemm_nt_grouped_rows_drop": lambda i: ("gemm_grouped_rows", i[0:3], i[6], "mfma", 2.0 * i[0] * i[1] * i[2]),
    "spv_gemm_nt_pool_bwd": lambda i: ("gemm_pool_bwd", i[1:4], i[7], "mfma", 2.0 * i[1] * i[2] * i[3]),
    "spv_gemm_tn": lambda i: ("gemm_tn", i[0:3], BF16, "mfma", 2.0 * i[0] * i[1] * i[2]),
    "spv_gemm_tn_fold": lambda i: ("gemm_tn", i[0:3], BF16, "mfma", 2.0 * i[0] * i[1] * i[2]),
    # (nprob, rows, splits, nfolds, part): the batched weight gradients; the work comes as the caller's hint (sum of 2 m n rows / the
    # slabs read + the sums written)
    "spv_gemm_tn_batch_part": lambda i: (("gemm_tn_batch", i[0:3], BF16, "mfma", float(i[-1])) if i[4] == 1 else
                                         ("splitk_reduce_batch", i[0:4], F32, "hbm", float(i[-1]))),
    # token-gradient pass of the embedding: read dtok, write the masked copy (when asked for: pointer 2)
    "spv_embed_bwd": lambda i: ("embed_bwd", i[0:3], i[3], "hbm", 1.0 * i[0] * i[1] * i[2] * _es(i[3]) * (1 if (i[-2] >> 2) & 1 else 2)),
    "spv_fnet_cls_fwd": lambda i: ("fnet_cls_fwd", i[0:3], i[3], "hbm", 1.0 * i[0] * (i[1] + 1) * i[2] * _es(i[3])),
    "spv_fnet_cls_bwd": lambda i: ("fnet_cls_bwd", i[0:3], i[3], "hbm", 1.0 * i[0] * (i[1] + 1) * i[2] * _es(i[3])),
    # one gathered row (n values) + row 0 (embed values) per image, read and written
    "spv_permut_row0_fwd": lambda i: ("permut_row0_fwd", i[0:4], i[4], "hbm", 2.0 * i[0] * (i[2] + i[3]) * _es(i[4])),
    # the dense input gradient [batch, d] written once (+ the n + embed values read)
    "spv_permut_row0_bwd": lambda i: ("permut_row0_bwd", i[0:4], i[4], "hbm", 1.0 * i[0] * (i[1] + i[2] + i[3]) * _es(i[4])),
    # (B, C, H, W, patch, K, mode, dtype): read the fp32 image, write the patch / token-row matrix
    "spv_patchify": lambda i: ("patchify", i[0:5], i[7], "hbm",
                               1.0 * i[0] * i[1] * i[2] * i[3] * 4 + 1.0 * i[0] * ((i[2] // i[4]) * (i[3] // i[4]) + (i[6] == 2)) * i[5] * _es(i[7])),
    # 32 x 64 tiles: fp32 read, two bf16 copies written
    "spv_weight_shadows_multi": lambda i: ("weight_shadows_multi", i[0:1], i[1], "hbm", 2048.0 * i[0] * (4 + 2 * _es(i[1]))),
    # read h [rows,n] + x [rows,k], write out [rows,n]
    "spv_spectre_tail_fwd": lambda i: ("tail_fwd", i[0:3], i[3], "hbm", i[0] * (2.0 * i[1] + i[2]) * _es(i[3])),
    # read dout, h; write dh [rows,n] and -- unless the caller passes no dx (MHPermutMix: the pooled skip gradient is added by the data
    # gradient GEMM's epilogue instead; pointer 7 of the call is NULL then) -- dx_pool [rows,k]
    "spv_spectre_tail_bwd": lambda i: ("tail_bwd" if not (i[-2] >> 7) & 1 else "tail_bwd_nodx", i[0:3], i[3], "hbm",
                                       i[0] * (3.0 * i[1] + (0 if (i[-2] >> 7) & 1 else i[2])) * _es(i[3])),
    # + read dx_add and up_src [rows,k]
    "spv_spectre_tail_bwd_up": lambda i: ("tail_bwd_up", i[0:3], i[3], "hbm", i[0] * (3.0 * i[1] + 3.0 * i[2]) * _es(i[3])),
    # read h3, res [rows,n], x [rows,k]; write f3, out2 [rows,n]
    "spv_spectre_tail_ln_fwd": lambda i: ("tail_ln_fwd", i[0:3], i[3], "hbm", i[0] * (4.0 * i[1] + i[2]) * _es(i[3])),
    # read dout2, f3, res, h3; write ds, dh3 [rows,n]
    "spv_spectre_tail_ln_bwd": lambda i: ("tail_ln_bwd", i[0:3], i[3], "hbm", i[0] * 6.0 * i[1] * _es(i[3])),
    "spv_add_layernorm_fwd": lambda i: ("addln_fwd", i[0:2], i[3], "hbm", 3.0 * i[0] * i[1] * _es(i[3])),
    "spv_add_layernorm_bwd": lambda i: ("addln_bwd", i[0:2], i[3], "hbm", (3.0 + (i[2] == 1)) * i[0] * i[1] * _es(i[3])),
    "spv_permut_gather_fwd": lambda i: ("gather_fwd", i[1:4], i[4], "hbm", i[1] * i[3] * (1.0 + i[2]) * _es(i[4])),
    "spv_permut_gather_bwd": lambda i: ("gather_bwd", i[0:3], i[3], "hbm", i[0] * i[2] * (1.0 + i[1]) * _es(i[3])),
    "spv_fnet_mix": lambda i: ("fnet_mix", i[0:3], i[3], "hbm", 2.0 * i[0] * i[1] * i[2] * _es(i[3])),
    # mixer + LayerNorm-1 + residual: read x, write the pre-norm tensor and x1
    "spv_fnet_ln_fwd": lambda i: ("fnet_ln_fwd", i[0:3], i[3], "hbm", 3.0 * i[0] * i[1] * i[2] * _es(i[3])),
    # read dout and the pre-norm tensor, write dx
    "spv_fnet_ln_bwd": lambda i: ("fnet_ln_bwd", i[0:3], i[3], "hbm", 3.0 * i[0] * i[1] * i[2] * _es(i[3])),
    "spv_haar_ln_fwd": lambda i: ("haar_ln_fwd", i[0:2], i[2], "hbm", 2.0 * i[0] * i[1] * _es(i[2])),
    "spv_haar_ln_bwd": lambda i: ("haar_ln_bwd", i[0:2], i[2], "hbm", 3.0 * i[0] * i[1] * _es(i[2])),
    "spv_haar_dwt": lambda i: ("haar_dwt", i[0:3], i[6], "hbm", 2.0 * i[0] * i[1] * i[2] * _es(i[6])),
    # p, g, m, v read + p, m, v written, 2048 elements per workgroup (the last chunk of a tensor is short: an upper bound)
    "spv_adamw_multi": lambda i: ("adamw_multi", i[0:1], F32, "hbm", 7.0 * 4 * 2048 * i[0]),
    "spv_weight_shadows": lambda i: ("weight_shadows", i[0:2], i[3], "hbm", i[0] * i[1] * (4.0 + 2 * _es(i[3]))),
    "spv_dropout": lambda i: ("dropout", i[0:1], i[1], "hbm", 2.0 * i[0] * _es(i[1])),
    "spv_axpby": lambda i: ("axpby", i[0:1], i[1], "hbm", 3.0 * i[0] * _es(i[1])),
    "spv_colsum": lambda i: ("colsum", i[0:2], i[2], "hbm", 1.0 * i[0] * i[1] * _es(i[2])),
    "spv_cast": lambda i: ("cast", i[2:3], i[1], "hbm", 1.0 * i[2] * (_es(i[0]) + _es(i[1]))),
}


class KernelTimer:
    def __init__(self):
        self.records = []  # (name, key, start, end)
        self.empties = []  # empty pairs recorded BETWEEN the kernel brackets, i.e. with the queue as busy as it is around them
        self.overhead_s = 0.0
        self.passes = 0    # steps recorded (bench.py counts them: per-step totals = totals / passes)

    def bracket(self, name, key, launch):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        self.records.append((name, key, e0, e1))
        if len(self.records) % 8 == 0:
            z0, z1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            z0.record()
            z1.record()
            self.empties.append((z0, z1))

    def _groups(self):
        torch.cuda.synchronize()
        if self.empties:
            # what a bracket adds to the kernel's own duration: the median of the empty pairs taken inside the pass (a pair on an
            # idle stream read 5-13 us depending on the box; the in-pass median agrees with rocprofv3's durations)
            self.overhead_s = sorted(a.elapsed_time(b) for a, b in self.empties)[len(self.empties) // 2] * 1e-3
            self.empties = []
        groups = {}
        for name, key, e0, e1 in self.records:
            g = groups.setdefault((name, key), [0, 0.0])
            g[0] += 1
            # the raw bracket: NOT reduced by the empty-pair overhead (round 2 subtracted 4.5 us per bracket and read the layer GEMM at
            # 30.85 us where rocprofv3 measured 34.35; the raw bracket is the conservative figure, within a few per cent of rocprofv3)
            g[1] += max(e0.elapsed_time(e1) * 1e-3, 1e-7)
        return groups

    @staticmethod
    def _work(name, key):
        """-> (label, shape, dtype name, bound or None, algorithmic work per launch, peak per second)"""
        model = _WORK_MODELS.get(name)
        if model is not None:
            label, shape, dt, bound, work = model(key)
            peak = PEAK_TFLOPS[dt] * 1e12 if bound == "mfma" else PEAK_HBM_GBS * 1e9
            return label, list(shape), "bf16" if dt == BF16 else "f32", bound, work, peak
        if name.startswith("torch:") and key and key[0] > 0:  # bench.py's own brackets around torch ops: key = (bytes,)
            return name, [], "f32", "hbm", float(key[0]), PEAK_HBM_GBS * 1e9
        return name.replace("spv_", ""), list(key[:4]), "-", None, 0.0, 1.0

    def summary(self):
        out = []
        for (name, key), (cnt, tot) in sorted(self._groups().items(), key=lambda kv: -kv[1][1]):
            label, shape, dt, bound, work, peak = self._work(name, key)
            d = dict(kernel=label, shape=shape, dtype=dt, launches=cnt, avg_us=round(tot / cnt * 1e6, 2), total_ms=round(tot * 1e3, 3))
            if bound is not None:
                ach = work * cnt / tot
                d.update(bound=bound, achieved=round(ach / (1e12 if bound == "mfma" else 1e9), 2),
                         unit="TFLOP/s" if bound == "mfma" else "GB/s", frac=round(ach / peak, 4), algorithmic=int(work))
            out.append(d)
        return out

    def roofline(self):
        """the single modelled (kernel, shape) with the largest total time in the recorded steps"""
        s = [d for d in self.summary() if "bound" in d]
        if not s:
            return None
        d = s[0]
        peak = (PEAK_TFLOPS[BF16 if d["dtype"] == "bf16" else F32]) if d["bound"] == "mfma" else PEAK_HBM_GBS
        return dict(bound=d["bound"], achieved=d["achieved"], peak=peak, unit=d["unit"], frac=d["frac"], traffic=None,
                    kernel=d["kernel"], shape=d["shape"], avg_us=d["avg_us"], launches=d["launches"],
                    event_overhead_us=round(self.overhead_s * 1e6, 2))


def set_kernel_timer(t):
    _native.timer = t


def _timing():
    return _native.timer is not None


def _gemm(a, b, bias, c, M, N, K, lda, ldb, ldc, accumulate=0, splits=1, workspace=None):
    if splits == 1 and K >= 256:
        # a handful of output tiles (the 512 x 100 classifier head: 4) would run on a handful of CUs for the whole K
        # loop -- 42 us for 52 MFLOP in fp32; split the reduction so that ~256 workgroups share it
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        # (bf16 problems of few rows take the library's 32 x 32-tile kernel instead: no workspace, no reduce launch)
        rows_kernel = (a.dtype == torch.bfloat16 and M <= 2048 and K <= 1536 and K % 16 == 0 and lda % 8 == 0 and ldb % 8 == 0
                       and 64 <= ((M + 31) // 32) * ((N + 31) // 32) <= 1024)
        if tiles <= 32 and not rows_kernel:
            splits = max(1, min(K // 64, 256 // tiles))
            if splits > 1:
                workspace = torch.empty((splits * M * N,), dtype=torch.float32, device=c.device)
    _native.call("spv_gemm_nt", _p(a), _p(b), _p(bias), _p(c), M, N, K, lda, ldb, ldc, _dt(a), _dt(c), accumulate, splits,
                 _p(workspace), _stream())


def _gemm_launch(a, b, bias, c, M, N, K, lda, ldb, ldc, accumulate=0, splits=1, workspace=None):
    """the raw C-ABI call (development tools time it directly)"""
    _native.call("spv_gemm_nt", _p(a), _p(b), _p(bias), _p(c), M, N, K, lda, ldb, ldc, _dt(a), _dt(c), accumulate, splits,
                 _p(workspace), _stream())


# Development switches: read from the environment ONLY in lab mode (SPV_LAB=1, together with the lab build of the library:
# `make -C csrc lab`, SPV_LIB_PATH=.../libspv_hip_lab.so).  In the product configuration every one of them is its default; tests
# that exercise an alternative path set the module attribute.
_LAB = os.environ.get("SPV_LAB") == "1"


def _lab(name, default):
    return os.environ.get(name, default) if _LAB else default


_SIDE_STREAM = _lab("SPV_SIDE_STREAM", "1") != "0"
_SIDE_MIN_FLOPS = float(_lab("SPV_SIDE_MIN_FLOPS", "1e11"))  # weight gradients at least this big fork to the side stream
_TN_DMA = _lab("SPV_TN_DMA", "0") == "1"  # must match the lab library's own switch (spv_gemm.hip)
_side_streams = {}
_side_keep = []  # tensors a side-stream kernel still reads/writes: kept alive until the join


def _side_stream(dev):
    s = _side_streams.get(dev)
    if s is None:
        s = _side_streams[dev] = torch.cuda.Stream(device=dev)
    return s


def join_side_stream():
    """Make the current stream wait for everything launched on the side stream (no-op when nothing is pending)."""
    if _side_keep:
        dev = _side_keep[0][0].device
        torch.cuda.current_stream().wait_stream(_side_streams[dev])
        _side_keep.clear()


# ------------------------------------------------------------------------------------------------
# folds travel with launches that happen anyway: a row kernel's fold of its partial column sums (dgamma / dbeta / dbias) either rides
# in its own layer's weight-gradient reduce (_sl_backward) or -- the FNet kernel has no GEMM beside it -- is held back for the next
# weight-gradient reduce of the same backward pass; whatever is still held when the autograd engine finishes runs as one launch.
# Held folds need gradient memory that outlives the node (a GradReducer sink): autograd would otherwise copy the unfinished tensor.
# With data parallelism the bucket hooks need every gradient as soon as its node has run: nothing is held.
# ------------------------------------------------------------------------------------------------
import collections  # noqa: E402

PATH_COUNTS = collections.Counter()   # host-side dispatch census (tests assert that the shapes they ran took the batched / side paths)
_held_folds = []   # (partials, outputs, parts, n)
_held_task = -2    # the autograd graph task (backward pass) the held folds belong to
FOLD_RIDERS = 6    # fold jobs a layer's own reduce launch carries
# Layer weight gradients travel together as well: alone each is 24 tiles of 128 x 128, i.e. ~21 K-slices to fill the chip (25 K-tiles
# per workgroup, 33 MB of partial sums written and re-read); eight of them in one launch fill it with 5 slices.  Same conditions as a
# held fold (sink-backed outputs, one process), and only gradients nothing reads before the backward pass ends.
_held_wgrads = []  # (dh, x, dw address, rows, n, k, fold or None)
WGRAD_BATCH = 8    # problems per launch (csrc/spv_gemm.hip TNB_MAX)
BATCH_FOLDS = 16   # fold jobs the batch's reduce launch carries (FJ_MAX)
_WGRAD_HOLD = _lab("SPV_WGRAD_BATCH", "1") != "0"
_WGRAD_SPLITS = int(_lab("SPV_WGRAD_BATCH_SPLITS", "0"))   # tuning aid: 0 = chosen per batch
_WGRAD_SIDE = _lab("SPV_WGRAD_SIDE", "1") != "0"            # the batch starts on the side stream beside the embedding's backward
NO_HOLD = bool(_lab("SPV_NO_HOLD", ""))         # nothing is held back (the launch sequence of the overlapped data-parallel path)
FOLD_RIDE = not _lab("SPV_NO_FOLD_RIDE", "")    # tail folds ride in their layer's weight-gradient reduce


# Under data parallelism the overlapped (eager) exchange needs every gradient as soon as its node has run, so nothing is held.  A
# step that exchanges its gradients in ONE call after the backward pass (spectre_vit.graph.GraphedDPStep) has no such need and
# sets this flag while its backward passes run.
HOLD_UNDER_DP = False
WGRAD_PER_LAYER = _lab("SPV_WGRAD_PER_LAYER", "0") == "1"   # lab: each layer's two weight gradients as their own side-stream batch, started when
# that layer's backward is done (operands still in the Infinity Cache) instead of one batch at the end of the pass
FOLDS_BESIDE_BATCH = False   # True: start_held_wgrads issues the held folds as their own launch on the main stream, beside the batch,
# instead of as extra workgroups of its reduce.  Measured (round 3): the reduce drops 29 -> 17 us, but the 68-us fold launch then
# stands in front of the embedding's backward on the main stream, which becomes the longer chain: window 251 -> 266 us.  Off.
TIME_HELD = False   # bench.py's roofline pass: bracket the launch sequence the headline times (held + batched weight gradients)


# Two private torch entry points carry the held launches: the id of the running backward pass and the autograd engine's end-of-pass
# callback.  Where a torch build lacks either, nothing is held (every launch runs at its own node: slower, same results).
_task_id_fn = getattr(torch._C, "_current_graph_task_id", None)
_engine = getattr(getattr(torch.autograd, "Variable", None), "_execution_engine", None)
_HOLD_API = _task_id_fn is not None and hasattr(_engine, "queue_callback")


def _graph_task_id():
    return _task_id_fn() if _task_id_fn is not None else -1


def _queue_end_of_backward(fn):
    _engine.queue_callback(fn)


def _hold_ok():
    if not _HOLD_API:
        return False
    if NO_HOLD or (_timing() and not TIME_HELD):
        return False
    if HOLD_UNDER_DP:
        return True
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)


def _fold_array(folds):
    arr = (_native.FoldJob * len(folds))()
    for j, (partials, outs, parts, n) in zip(arr, folds):
        j.partials = _p(partials)
        for i, o in enumerate(outs):
            j.out[i] = o if isinstance(o, int) else _p(o)   # held folds carry raw sink addresses (see _hold_fold)
        j.parts, j.nsum, j.n = parts, len(outs), n
    return arr


def _batch_splits(tiles):
    """K-slices of a batch of `tiles` 128 x 128 tiles: the count whose workgroups fill whole rounds of the chip's ~512 slots (two
    4-wave workgroups per CU) best; measured on 192 tiles: 5 slices (960 workgroups) 307 us, 2: 321, 4: 355, 3: 396"""
    if _WGRAD_SPLITS > 0:
        return _WGRAD_SPLITS
    if 160 <= tiles <= 224:
        # the six 33 280-row layer gradients of the Small model (192 tiles): re-measured in round 3 on both tile shapes of the batched
        # kernel (tools/tnb_bench.py: 3: 296 / 278 us, 4: 305 / 341, 5: 259 / 281, 6: 247 / 249, 7: 231 / 229, 8: 256 / 267, 10: 242 / 240)
        # and inside the replayed step (5 -> 7 slices: 1.734 / 1.719 -> 1.702 / 1.712 ms)
        return 7
    best, best_score = 1, -1.0
    for sp in range(1, 11):
        rounds = tiles * sp / 512.0
        score = rounds / max(1.0, float(-(-tiles * sp // 512))) - 0.01 * sp
        if rounds >= 0.7 and score > best_score:
            best, best_score = sp, score
    return best


def _flush_held_wgrads(folds_out=None):
    """the weight gradients held back during this backward pass: one launch (+ one reduce that carries their folds and the folds held
    so far) per group of up to eight.  Returns every tensor the launches read or write (operands, fold partials, split-K
    workspaces): a caller that runs this on a side stream keeps them alive until the streams are joined.  folds_out (a list): the
    folds are NOT given to the reduce but appended to it -- the caller launches them itself (start_held_wgrads: on the main stream)."""
    used = []
    while _held_wgrads:
        # up to eight per launch, the long reductions first; gradients over fewer rows (the CLS-only last layer's: 512) ride in the same
        # launch as short problems -- one workgroup per tile, no split-K, no launch + reduce of their own behind the big one
        rows = max(w[3] for w in _held_wgrads)
        group = sorted(_held_wgrads, key=lambda w: -w[3])[:WGRAD_BATCH]
        taken = {id(w) for w in group}
        _held_wgrads[:] = [w for w in _held_wgrads if id(w) not in taken]   # (by identity: the tuples hold tensors)
        probs = (_native.TnProblem * len(group))()
        tiles = floats = 0
        flops = 0.0
        folds = []
        for q, (dh, x, dwp, r, n, k, fold) in zip(probs, group):
            q.a, q.b, q.c, q.m, q.n, q.lda, q.ldb, q.ldc, q.k = _p(dh), _p(x), dwp, n, k, n, k, k, r
            if r == rows:
                tiles += ((n + 127) // 128) * ((k + 127) // 128)
                floats += n * k
            flops += 2.0 * r * n * k
            used += [dh, x]
            if fold is not None:
                folds.append(fold)
        while _held_folds and len(folds) < BATCH_FOLDS:
            folds.append(_held_folds.pop(0))
        if folds_out is not None:
            folds_out += folds
            folds = []
        splits = max(1, min(_batch_splits(tiles), rows // 256))   # (the CLS-only last layer: 512 rows)
        ws = torch.empty((splits * floats,), dtype=torch.float32, device=group[0][0].device)
        used.append(ws)
        used += [f[0] for f in folds]   # the folds' partial column sums: read by the reduce launch, long after this function returns
        arr = _fold_array(folds) if folds else None
        if _timing():   # a measuring pass brackets the GEMM launch and the reduce launch separately (same kernels, same order)
            _native.hint = int(flops)
            _native.call("spv_gemm_tn_batch_part", ctypes.addressof(probs), len(group), rows, splits, _p(ws), ctypes.addressof(arr) if folds else 0,
                         len(folds), 1, _stream())
            _native.hint = int(4.0 * floats * (splits + 1))
            _native.call("spv_gemm_tn_batch_part", ctypes.addressof(probs), len(group), rows, splits, _p(ws), ctypes.addressof(arr) if folds else 0,
                         len(folds), 2, _stream())
        else:
            _native.call("spv_gemm_tn_batch", ctypes.addressof(probs), len(group), rows, splits, _p(ws), ctypes.addressof(arr) if folds else 0,
                         len(folds), _stream())
        PATH_COUNTS["wgrad_batch"] += 1
        PATH_COUNTS["wgrad_batch_problems"] += len(group)
    return used


def start_held_wgrads():
    """Called by the LAST node of the backward pass that does real work (the patch embedding): every layer's weight gradient is held
    by now, so their batch starts here on the side stream and runs beside the embedding's backward -- a chain of small, latency-bound
    launches that leaves most of the chip idle.  The end-of-pass callback joins the streams.  True when something was started."""
    if not (_WGRAD_SIDE and _held_wgrads and _held_task == _graph_task_id()) or _timing():
        return False   # (a timed pass keeps the batch on the main stream: its event brackets must not overlap other kernels)
    side = _side_stream(_held_wgrads[0][0].device)
    side.wait_stream(torch.cuda.current_stream())
    folds = []
    with torch.cuda.stream(side):
        used = _flush_held_wgrads(folds if FOLDS_BESIDE_BATCH else None)
    if folds:
        # the folds (column sums of the tails' partial slabs: nothing reads them before the optimizer) as ONE launch of 256-thread
        # workgroups on the MAIN stream: they fit on the CUs beside the batch's workgroups and are done long before it ends -- as
        # extra workgroups of its reduce they were 16 of that launch's 29 us, behind the batch
        while _held_folds:
            folds.append(_held_folds.pop(0))
        arr = _fold_array(folds)
        _native.call("spv_fold_multi", ctypes.addressof(arr), len(folds), _stream())
        used += [f[0] for f in folds]
    # everything the side-stream launches touch stays referenced until join_side_stream(): the operands AND the fold partials (they
    # were allocated on the main stream and had no other owner once the flush returned -- the caching allocator could hand their
    # blocks to the embedding's backward, which runs on the main stream beside the batch, before the reduce has read them)
    if used:
        _side_keep.append(tuple(used))
    PATH_COUNTS["wgrad_side_start"] += 1
    return True


def _hold_wgrad(dh, x, dw, sink, rows, n, k, fold, fold_sunk):
    """hold a layer weight gradient for the batch launch at the end of this backward pass (see _held_wgrads).  False: not held."""
    global _held_task
    if not (_WGRAD_HOLD and _hold_ok() and sink is not None and dw.data_ptr() == sink.view.data_ptr() and (fold is None or fold_sunk)):
        return False
    task = _graph_task_id()
    if task < 0:
        return False
    if _held_task != task:
        _held_folds.clear()
        _held_wgrads.clear()
        _queue_end_of_backward(flush_held_folds)
        _held_task = task
    f = None if fold is None else (fold[0], tuple(o.data_ptr() for o in fold[1]), fold[2], fold[3])   # raw sink addresses, as _hold_fold
    _held_wgrads.append((dh, x, dw.data_ptr(), rows, n, k, f))
    return True


def flush_held_folds():
    """run the weight gradients and folds still held (called by the autograd engine when the backward pass is over)"""
    global _held_task
    _flush_held_wgrads()
    if _held_folds:
        arr = _fold_array(_held_folds)
        _native.call("spv_fold_multi", ctypes.addressof(arr), len(_held_folds), _stream())
        _held_folds.clear()
    _held_task = -2
    join_side_stream()   # a batch started early by start_held_wgrads


def _hold_fold(partials, outs, sinks, parts, n):
    """hold a fold for the next weight-gradient reduce of this backward pass.  Only when every output IS its parameter's sink slot
    (memory that outlives the node; autograd adopts the alias without copying), and never by keeping the output tensors themselves:
    a second reference makes AccumulateGrad clone the -- still unfolded -- gradient.  False (not held) otherwise."""
    global _held_task
    if not _hold_ok() or any(sk is None or o.data_ptr() != sk.view.data_ptr() for o, sk in zip(outs, sinks)):
        return False
    task = _graph_task_id()
    if task < 0:   # not inside a backward pass
        return False
    if _held_task != task:
        _held_folds.clear()   # leftovers of a backward pass that never finished (an exception): their launch must not ride along
        _held_wgrads.clear()
        _queue_end_of_backward(flush_held_folds)
        _held_task = task
    _held_folds.append((partials, tuple(o.data_ptr() for o in outs), parts, n))
    return True


def _fold_rides(dtype, rows, n, k):
    """the tail backward's fold can ride in this weight gradient's split-K reduce (bf16 TN path on the main stream)"""
    return (dtype == torch.bfloat16 and n % 8 == 0 and k % 8 == 0 and FOLD_RIDE
            and not (_SIDE_STREAM and not _timing() and 2.0 * rows * n * k >= _SIDE_MIN_FLOPS))


def _fold_job(partials, outs, rows, n):
    """(partials, outputs, partial slabs, row length): the fold of a tail backward's column sums, to ride in / be deferred with the
    weight gradient's reduction"""
    return (partials, tuple(outs), _native.call("spv_tail_bwd_parts", rows), n)


def _sunk(outs, sinks):
    """every gradient tensor IS its parameter's sink slot (memory that outlives the node: its content may be written later)"""
    return all(sk is not None and o.data_ptr() == sk.view.data_ptr() for o, sk in zip(outs, sinks))


def _weight_grad(dh, x, rows, n, k, sink=None, fold=None, fold_sunk=False):
    """dW[n,k] = dh[rows,n]^T . x[rows,k], split-K over rows.  bf16: TN kernel straight from the row-major activations
    (transposing LDS reads); fp32 (parity path): NT kernel over explicit transposes.  fold (a _fold_job, only when
    _fold_rides): the same layer's dgamma / dbeta / dbias fold, run as extra workgroups of the split-K reduce."""
    dev = dh.device
    dw = _grad_buf(sink, (n, k), dev)
    tiles = ((n + 127) // 128) * ((k + 127) // 128)
    # ~2 workgroups per CU: measured optimum on the 768 x 512 x 33280 weight gradient (21 splits: 51 us; 12: 69; 42: 56; 64: 64)
    splits = max(1, min(512 // tiles, (rows + 511) // 512 if tiles >= 8 else (rows + 63) // 64))
    if tiles < 8:
        splits = min(splits, 64)  # the patch-embedding gradient (512 x 48): 128 slices made the reduce (10 us) as long as the GEMM
    elif n == 512 and k % 128 == 0 and k >= 1024 and rows >= 8192:
        # the 512 x 128 tile (spv_gemm_tn takes it from 192 workgroups up): one workgroup per CU
        splits = max(1, min(256 // (k // 128), rows // 2048))
    # (The 256 x 128 tile at 4 slices is 8 % faster on the MHPermutMix gradient [512, 8192, 33280] in isolation -- 320 against 348 us --
    # and SLOWER where it runs, on the side stream beside the data-gradient GEMM and the inverse gather: 701 against ~500 us, step 5.92
    # -> 6.25 ms.  Its 112 KB of LDS per workgroup leave those kernels less of every CU than the 128 x 128 tile's 40 KB.)
    if _TN_DMA and n % 128 == 0 and k % 128 == 0 and rows % 64 == 0 and dh.dtype == torch.bfloat16:
        # the LDS-DMA kernel (spv_gemm.hip gemm_tn_dma_kernel; opt-in, SPV_TN_DMA=1) runs ONE 8-wave workgroup per CU: one dispatch
        # round of <= 256 workgroups (24 tiles x 10 K-slices of 52 K-tiles at the layer shapes)
        splits = max(1, min(256 // tiles, rows // 64))
    ws = None
    if dh.dtype == torch.bfloat16 and n % 8 == 0 and k % 8 == 0:
        if tiles >= 8 and 2.0 * rows * n * k < _SIDE_MIN_FLOPS and _hold_wgrad(dh, x, dw, sink, rows, n, k, fold, fold_sunk):
            return dw   # computed with the other layers' at the end of the backward pass
        def launch(riders=True):
            nonlocal ws
            if ws is None and splits > 1:
                ws = torch.empty((splits * n * k,), dtype=torch.float32, device=dev)
            folds = [fold] if fold is not None else []
            if riders and _held_task == _graph_task_id():
                while _held_folds and len(folds) < FOLD_RIDERS:
                    folds.append(_held_folds.pop(0))
            if folds:
                arr = _fold_array(folds)
                _native.call("spv_gemm_tn_fold", _p(dh), _p(x), _p(dw), n, k, rows, n, k, k, F32, 0, splits, _p(ws), ctypes.addressof(arr),
                             len(folds), _stream())
            else:
                _native.call("spv_gemm_tn", _p(dh), _p(x), _p(dw), n, k, rows, n, k, k, F32, 0, splits, _p(ws), _stream())
        if _SIDE_STREAM and not _timing() and 2.0 * rows * n * k >= _SIDE_MIN_FLOPS:
            # a big weight gradient (the MHPermutMix 8192 -> 512 linear: 279 GFLOP) has no consumer inside the backward
            # chain: run it on a second HIP stream so that it fills the ramp/tail gaps of the data-gradient GEMM and
            # overlaps the HBM-bound inverse gather on the main stream (10.57 -> 10.38 ms/step).  Not worth it for the
            # 26-GFLOP layer GEMMs (3.32 -> 3.42 ms/step: the fork/join costs more than the overlap gains).
            # join_side_stream() (end of the calling autograd node) orders everything after it again.
            side = _side_stream(dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                launch(False)  # allocates its split-K workspace from the side stream's pool; held folds stay with the main stream
            _side_keep.append((dh, x, ws, dw))
        else:
            launch()
        return dw
    ws = torch.empty((splits * n * k,), dtype=torch.float32, device=dev) if splits > 1 else None
    ld = (rows + 7) // 8 * 8
    dht = torch.empty((n, ld), dtype=dh.dtype, device=dev)
    xt = torch.empty((k, ld), dtype=x.dtype, device=dev)
    st = _stream()
    _native.call("spv_cast_transpose", _p(dh), _dt(dh), _p(dht), _dt(dht), rows, n, ld, 0, 0, 0, st)
    _native.call("spv_cast_transpose", _p(x), _dt(x), _p(xt), _dt(xt), rows, k, ld, 0, 0, 0, st)
    _gemm(dht, xt, None, dw, n, k, ld, ld, ld, k, 0, splits, ws)
    return dw


class GradSink:
    """A parameter's slot in a flat gradient bucket (installed by spectre_vit.dp.GradReducer as ``p._spv_grad_sink``).
    Backward kernels write a parameter's gradient straight into the slot, so autograd's AccumulateGrad adopts that
    tensor as ``p.grad`` without the extra ``grad += new`` pass over every weight.  ``used`` guards the (never taken
    here) case of a parameter that receives two gradient contributions in one step: the second one goes to fresh memory
    and autograd sums them."""
    __slots__ = ("view", "used")

    def __init__(self, view):
        self.view = view
        self.used = False


def _sink(p):
    return getattr(p, "_spv_grad_sink", None)


def _grad_buf(sink, shape, device):
    if sink is not None and not sink.used and tuple(sink.view.shape) == tuple(shape):
        sink.used = True
        # a FRESH alias of the slot: autograd's AccumulateGrad adopts an incoming gradient as p.grad only when nothing else holds
        # that tensor object (use_count check); returning sink.view itself made it clone the gradient and the reducer copy it back
        return sink.view.detach()
    return torch.empty(shape, dtype=torch.float32, device=device)


def _new_seed():
    return int(torch.randint(0, 2 ** 62, (1,)).item())


# ------------------------------------------------------------------------------------------------
# SpectreLinear: GELU(LN(x W^T + b)) + avgpool(x) [+ dropout]      (reference layers.py:76-101)
# ------------------------------------------------------------------------------------------------
def _sl_forward(x2, weight, bias, gamma, beta, p_drop, out_fp32):
    """raw SpectreLinear forward on a contiguous [rows, k] tensor -> (out [rows, n], saved-for-backward tuple)"""
    n, k = weight.shape
    rows = x2.shape[0]
    dt = x2.dtype
    mult = 8 if dt == torch.bfloat16 else 4
    if n % mult or k % mult:
        raise ValueError(f"SpectreLinear({k}->{n}) in {dt}: channel counts must be multiples of {mult}")
    wc, wt = _shadows.get(weight, dt)
    dev = x2.device
    h = torch.empty((rows, n), dtype=dt, device=dev)
    _gemm(x2, wc, bias, h, rows, n, k, k, k, n)
    out = torch.empty((rows, n), dtype=torch.float32 if out_fp32 else dt, device=dev)
    mean = torch.empty((rows,), dtype=torch.float32, device=dev)
    rstd = torch.empty((rows,), dtype=torch.float32, device=dev)
    seed = _new_seed() if p_drop > 0.0 else 0
    _native.call("spv_spectre_tail_fwd", _p(h), _p(x2), _p(gamma), _p(beta), _p(out), _p(mean), _p(rstd), rows, n, k,
                 _dt(h), _dt(out), float(p_drop), seed, _stream())
    sinks = (_sink(weight), _sink(bias), _sink(gamma), _sink(beta))
    return out, (x2, h, mean, rstd, gamma, beta, wt, sinks, rows, n, k, float(p_drop), seed)


def _sl_backward(dout2, saved, need_dx=True, dx_add=None, up=None):
    """raw SpectreLinear backward -> (dx or None, dW, dbias, dgamma, dbeta); dx_add: a gradient of the same input that is
    folded into dx by the tail kernel (saves a separate elementwise add); up = (src, p_drop, seed): the skip gradient of the
    layer above, formed here from its source instead of being written by that layer and accumulated by its GEMM"""
    x2, h, mean, rstd, gamma, beta, wt, sinks, rows, n, k, p_drop, seed = saved
    dev = x2.device
    if not dout2.is_contiguous():
        dout2 = dout2.contiguous()
    dh = torch.empty_like(h)
    dx = torch.empty_like(x2)
    s_w, s_b, s_g, s_be = sinks
    dgamma = _grad_buf(s_g, (n,), dev)
    dbeta = _grad_buf(s_be, (n,), dev)
    dbias = _grad_buf(s_b, (n,), dev)
    partials = torch.empty((_native.call("spv_rowop_partial_floats", n),), dtype=torch.float32, device=dev)
    ride = _fold_rides(dh.dtype, rows, n, k)  # the fold of the three column sums rides in the weight gradient's split-K reduce
    pg, pb, pbi = (0, 0, 0) if ride else (_p(dgamma), _p(dbeta), _p(dbias))
    if up is not None:
        _native.call("spv_spectre_tail_bwd_up", _p(dout2), _p(h), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dh), _p(dx),
                     pg, pb, pbi, _p(partials), rows, n, k, _dt(h), _dt(dout2), p_drop, seed,
                     _p(dx_add) if need_dx else 0, _p(up[0]), float(up[1]), int(up[2]), _stream())
    else:
        _native.call("spv_spectre_tail_bwd", _p(dout2), _p(h), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dh), _p(dx),
                     pg, pb, pbi, _p(partials), rows, n, k, _dt(h), _dt(dout2), p_drop, seed,
                     _p(dx_add) if need_dx else 0, _stream())
    dw = _weight_grad(dh, x2, rows, n, k, s_w, _fold_job(partials, (dgamma, dbeta, dbias), rows, n) if ride else None,
                      ride and _sunk((dgamma, dbeta, dbias), (s_g, s_be, s_b)))
    if need_dx:
        _gemm(dh, wt, None, dx, rows, k, n, n, wt.shape[1], k, accumulate=1)
    else:
        dx = None
    return dx, dw, dbias, dgamma, dbeta


class SpectreLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, p_drop, out_fp32):
        _require_gpu(x, weight)
        n, k = weight.shape
        shape = x.shape
        x2 = x.reshape(-1, k)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        out, saved = _sl_forward(x2, weight, bias, gamma, beta, p_drop, out_fp32)
        ctx.saved = saved
        ctx.shape = shape
        return out.reshape(*shape[:-1], n)

    @staticmethod
    def backward(ctx, dout):
        saved = ctx.saved
        rows, n = saved[8], saved[9]
        dx, dw, dbias, dgamma, dbeta = _sl_backward(dout.reshape(rows, n), saved, ctx.needs_input_grad[0])
        join_side_stream()
        return (dx.reshape(ctx.shape) if dx is not None else None), dw, dbias, dgamma, dbeta, None, None


def spectre_linear(x, weight, bias, gamma, beta, p_drop=0.0, out_fp32=False):
    return SpectreLinearFn.apply(x, weight, bias, gamma, beta, p_drop, out_fp32)


# ------------------------------------------------------------------------------------------------
# residual + LayerNorm   mode 0: LN(a) + b (spectre.py:66)    mode 1: LN(a + b) (spectre.py:67)
# ------------------------------------------------------------------------------------------------
def _addln_forward(a2, b2, gamma, beta, mode):
    rows, n = a2.shape
    out = torch.empty_like(a2)
    mean = torch.empty((rows,), dtype=torch.float32, device=a2.device)
    rstd = torch.empty_like(mean)
    _native.call("spv_add_layernorm_fwd", _p(a2), _p(b2), _p(gamma), _p(beta), _p(out), _p(mean), _p(rstd), rows, n, mode,
                 _dt(a2), _stream())
    return out, (a2, b2, mean, rstd, gamma, (_sink(gamma), _sink(beta)), rows, n, mode)


def _addln_backward(d2, saved):
    a2, b2, mean, rstd, gamma, sinks, rows, n, mode = saved
    if not d2.is_contiguous():
        d2 = d2.contiguous()
    din = torch.empty_like(a2)
    dgamma = _grad_buf(sinks[0], (n,), a2.device)
    dbeta = _grad_buf(sinks[1], (n,), a2.device)
    partials = torch.empty((_native.call("spv_rowop_partial_floats", n),), dtype=torch.float32, device=a2.device)
    _native.call("spv_add_layernorm_bwd", _p(d2), _p(a2), _p(b2), _p(mean), _p(rstd), _p(gamma), _p(din), _p(dgamma),
                 _p(dbeta), _p(partials), rows, n, mode, _dt(a2), _stream())
    return din, dgamma, dbeta


class AddLayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, gamma, beta, mode):
        _require_gpu(a, b)
        n = a.shape[-1]
        out, saved = _addln_forward(a.reshape(-1, n).contiguous(), b.reshape(-1, n).contiguous(), gamma, beta, mode)
        ctx.saved = saved
        ctx.shape = a.shape
        return out.reshape(a.shape)

    @staticmethod
    def backward(ctx, dout):
        rows, n, mode = ctx.saved[6], ctx.saved[7], ctx.saved[8]
        din, dgamma, dbeta = _addln_backward(dout.reshape(rows, n), ctx.saved)
        din = din.reshape(ctx.shape)
        return din, (dout if mode == 0 else din), dgamma, dbeta, None


def add_layernorm(a, b, gamma, beta, mode):
    return AddLayerNormFn.apply(a, b, gamma, beta, mode)


# ------------------------------------------------------------------------------------------------
# MHPermutMix gather (layers.py:68-72)
# ------------------------------------------------------------------------------------------------
def permut_pack(perms: torch.Tensor, signs: torch.Tensor) -> torch.Tensor:
    _require_gpu(perms, signs)
    heads, d = perms.shape
    # opaque to the caller: wide uint32 tables [2][heads][d] + (when d fits 16 bits) the compact 16-bit / sign-bit tables
    idx = torch.empty((_native.call("spv_permut_table_words", heads, d),), dtype=torch.int32, device=perms.device)
    _native.call("spv_permut_pack", _p(perms.contiguous()), _p(signs.contiguous().float()), _p(idx), heads, d, _stream())
    return idx


class PermutGatherFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx, heads):
        _require_gpu(x, idx)
        B = x.shape[0]
        xc = x.contiguous()
        d = xc.numel() // B
        g = torch.empty((B, heads * d), dtype=x.dtype, device=x.device)
        _native.call("spv_permut_gather_fwd", _p(xc), _p(idx), _p(g), 0, 0, B, heads, d, _dt(xc), _stream())
        ctx.idx = idx
        ctx.meta = (x.shape, B, heads, d)
        return g

    @staticmethod
    def backward(ctx, dg):
        shape, B, heads, d = ctx.meta
        dgc = dg.contiguous()
        dx = torch.empty((B, d), dtype=dg.dtype, device=dg.device)
        _native.call("spv_permut_gather_bwd", _p(dgc), _p(ctx.idx), _p(dx), B, heads, d, _dt(dgc), _stream())
        return dx.reshape(shape), None, None


# ------------------------------------------------------------------------------------------------
# spectral mixers
# ------------------------------------------------------------------------------------------------
_twiddles = {}


def _fnet_twiddle(tokens, device):
    key = (tokens, device)
    t = _twiddles.get(key)
    if t is None:
        t = torch.empty((_native.call("spv_fnet_twiddle_floats", tokens),), dtype=torch.float32, device=device)
        _native.call("spv_fnet_make_twiddle", _p(t), tokens, _stream())
        _twiddles[key] = t
    return t


def _fnet_raw(x, add_in=None):
    B, N, D = x.shape
    xc = x.contiguous()
    y = torch.empty_like(xc)
    wsn = _native.call("spv_fnet_workspace_floats", B, N, D)
    ws = torch.empty((wsn,), dtype=torch.float32, device=x.device) if wsn else None
    tw = _fnet_twiddle(N, x.device)

    _native.call("spv_fnet_mix", _p(xc), _p(y), _p(add_in), _p(tw), B, N, D, _dt(xc), _p(ws), _stream())
    return y


class FNetMixFn(torch.autograd.Function):
    """y = Re(fft2(x)) over the last two axes; symmetric operator => backward is the same kernel."""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        return _fnet_raw(x)

    @staticmethod
    def backward(ctx, dy):
        return _fnet_raw(dy)


class RfftRealFn(torch.autograd.Function):
    """rfft(x, dim=-1).real (reference spectre_vit/modules/spectre.py:9-14)."""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        D = x.shape[-1]
        xc = x.reshape(-1, D).contiguous()
        y = torch.empty((xc.shape[0], D // 2 + 1), dtype=x.dtype, device=x.device)
        _native.call("spv_rfft_real", _p(xc), _p(y), xc.shape[0], D, 0, _dt(xc), _stream())
        ctx.meta = (x.shape, D)
        return y.reshape(*x.shape[:-1], D // 2 + 1)

    @staticmethod
    def backward(ctx, dy):
        shape, D = ctx.meta
        dyc = dy.reshape(-1, D // 2 + 1).contiguous()
        dx = torch.empty((dyc.shape[0], D), dtype=dy.dtype, device=dy.device)
        _native.call("spv_rfft_real", _p(dyc), _p(dx), dyc.shape[0], D, 1, _dt(dyc), _stream())
        return dx.reshape(shape)


def _haar_raw(x, axis, levels, inverse, zero_mode=False):
    B, N, D = x.shape
    xc = x.contiguous()
    y = torch.empty_like(xc)
    scratch = torch.empty_like(xc) if levels > 1 else None
    _native.call("spv_haar_dwt", _p(xc), _p(y), B, N, D, axis, levels, inverse | (2 if zero_mode else 0), _dt(xc), _p(scratch), _stream())
    return y


class HaarDWTFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, axis, levels, zero_mode=False):
        _require_gpu(x)
        ctx.meta = (axis, levels, bool(zero_mode))
        return _haar_raw(x, axis, levels, 0, zero_mode)

    @staticmethod
    def backward(ctx, dy):
        axis, levels, zero_mode = ctx.meta
        return _haar_raw(dy, axis, levels, 1, zero_mode), None, None, None


# ------------------------------------------------------------------------------------------------
# patch embedding  (spectre.py:124-156, patch_embeddings.py:28-43)
# ------------------------------------------------------------------------------------------------
class SpectralFoldFn(torch.autograd.Function):
    """W_full[e,(c,p,q)] = sum_uv proj_w[e,(c,u,v)] fh[u] fw[v] Re(rfft2_ortho)[(u,v),(p,q)]."""

    @staticmethod
    def forward(ctx, proj_w, fh, fw, chans, patch):
        _require_gpu(proj_w)
        E = proj_w.shape[0]
        wf = torch.empty((E, chans * patch * patch), dtype=torch.float32, device=proj_w.device)
        if torch.is_autocast_enabled("cuda"):   # a bf16 step: the token GEMM's operand comes out of the same launch (PatchEmbedFn picks it up)
            wb = torch.empty(wf.shape, dtype=torch.bfloat16, device=proj_w.device)
            _native.call("spv_spectral_fold_bf16", _p(proj_w), _p(fh), _p(fw), _p(wf), _p(wb), E, chans, patch, _stream())
            wf._spv_bf16 = wb
        else:
            _native.call("spv_spectral_fold", _p(proj_w), _p(fh), _p(fw), _p(wf), E, chans, patch, _stream())
        ctx.save_for_backward(proj_w, fh, fw)
        ctx.sinks = (_sink(proj_w), _sink(fh), _sink(fw))
        ctx.meta = (E, chans, patch)
        return wf

    @staticmethod
    def backward(ctx, dwf):
        proj_w, fh, fw = ctx.saved_tensors
        E, chans, patch = ctx.meta
        dwf = dwf.contiguous()
        dw = _grad_buf(ctx.sinks[0], proj_w.shape, proj_w.device)
        dfh = _grad_buf(ctx.sinks[1], fh.shape, fh.device)
        dfw = _grad_buf(ctx.sinks[2], fw.shape, fw.device)
        scratch = torch.empty_like(proj_w)
        _native.call("spv_spectral_fold_bwd", _p(dwf), _p(proj_w), _p(fh), _p(fw), _p(dw), _p(dfh), _p(dfw), _p(scratch), E,
                     chans, patch, _stream())
        return dw, dfh, dfw, None, None


class _EmbedKey:
    """identity of one PatchEmbedFn node (its output tensor carries it; TapClsFn stashes the CLS-row gradient under it)"""
    __slots__ = ("__weakref__",)


_cls_grad_stash = weakref.WeakKeyDictionary()


class PatchEmbedFn(torch.autograd.Function):
    """tokens[b,0] = cls + pos[0]; tokens[b,1+n] = W_full . patch(b,n) + bias + pos[1+n]."""

    @staticmethod
    def forward(ctx, img, w_full, bias, cls, pos, patch, dtype, norm=None, p_drop=0.0):
        """img: float NCHW (the reference's input contract), or -- SURVEY 8f-3 -- the loader's uint8 NHWC batch with
        norm = (mean[C], inv_std[C]) device tensors: /255 + Normalize are then folded into the patch gather.
        p_drop: the nn.Dropout that follows the embedding (reference spectre.py:156), kept inside this node so that its backward,
        the CLS-row gradient of the global residual and the three column sums are one pass over the token gradient."""
        _require_gpu(img, w_full)
        u8 = img.dtype == torch.uint8
        if u8:
            if norm is None:
                raise ValueError("uint8 images need the (mean, inv_std) normalisation tensors")
            B, H, W, C = img.shape
        else:
            B, C, H, W = img.shape
        E, K = w_full.shape
        Np = (H // patch) * (W // patch)
        T = Np + 1
        dev = img.device
        img = img.contiguous() if u8 else img.contiguous().float()
        mult = 8 if dtype == torch.bfloat16 else 4
        if K % mult or E % mult:
            raise ValueError(f"patch embedding: C*P*P={K} and embed_dim={E} must be multiples of {mult}")
        st = _stream()
        # token rows [B][T][K], the CLS slot of every image zero: the GEMM below then writes cls + pos[0] there by itself (its row bias
        # holds that sum in row 0), and the backward's TN weight-gradient GEMM reads the same matrix against dtok as both lie in memory
        patches = torch.empty((B * T, K), dtype=dtype, device=dev)
        if u8:
            _native.call("spv_patchify_u8", _p(img), _p(norm[0]), _p(norm[1]), _p(patches), B, C, H, W, patch, K, 2, _DT[dtype], st)
        else:
            _native.call("spv_patchify", _p(img), _p(patches), B, C, H, W, patch, K, 2, _DT[dtype], st)
        wc = w_full if dtype == torch.float32 else (getattr(w_full, "_spv_bf16", None) if dtype == torch.bfloat16 else None)
        if wc is None:
            wc = _raw_cast(w_full, dtype)
        posbias = torch.empty((T, E), dtype=torch.float32, device=dev)
        _native.call("spv_embed_posbias", _p(pos), _p(bias), _p(cls), _p(posbias), Np, E, st)
        tokens = torch.empty((B, T, E), dtype=dtype, device=dev)
        seed = _new_seed() if p_drop > 0.0 else 0
        # the dropout rides in the GEMM's epilogue (the mask spv_dropout would draw from the same seed; the backward re-derives it)
        _native.call("spv_gemm_nt_grouped_rows_drop", _p(patches), _p(wc), 0, _p(posbias), _p(tokens), B * T, E, K, K, K, E,
                     _DT[dtype], _DT[dtype], T, T, 0, float(p_drop), seed, st)
        # bf16: the backward's TN weight-gradient GEMM reads the patch matrix as it lies here (3 MB), so keep it
        ctx.save_for_backward(None if u8 else img, patches if (dtype == torch.bfloat16 or u8) else None)
        ctx.meta = (B, C, H, W, patch, E, K, Np, T, dtype, cls.shape, pos.shape, float(p_drop), seed)
        ctx.sinks = (_sink(bias), _sink(cls), _sink(pos))
        ctx.key = _EmbedKey()   # TapClsFn hands the CLS-row gradient of the global residual to this node's backward under this key
        tokens._spv_embed_key = ctx.key
        return tokens

    @staticmethod
    def backward(ctx, dtok):
        img, patches = ctx.saved_tensors
        B, C, H, W, patch, E, K, Np, T, dtype, cls_shape, pos_shape, p_drop, seed = ctx.meta
        dev = dtok.device
        st = _stream()
        dtok = dtok.contiguous()
        early = start_held_wgrads()   # the layers' weight gradients, one batched launch on the side stream beside everything below
        gcls = _cls_grad_stash.pop(ctx.key, None)   # (B, E): the global residual's CLS-row gradient, not yet added (TapClsFn)
        if gcls is not None:
            gcls = gcls.to(dtok.dtype).contiguous()
        s_bias, s_cls, s_pos = ctx.sinks
        dpos_full = _grad_buf(s_pos, pos_shape, dev)  # straight into the data-parallel bucket / the optimizer's fixed gradient slot
        dbias = _grad_buf(s_bias, (E,), dev)
        dcls = _grad_buf(s_cls, cls_shape, dev)
        part = torch.empty((_native.call("spv_embed_bwd_groups", B) * T * E,), dtype=torch.float32, device=dev)
        # one pass: + CLS-row gradient, dropout mask, the batch sums of the three parameter gradients; then their fold
        masked = torch.empty_like(dtok) if (gcls is not None or p_drop > 0.0) else None
        _native.call("spv_embed_bwd", _p(dtok), _p(gcls), _p(masked), _p(part), _p(dpos_full), _p(dbias), _p(dcls), B, T, E, p_drop, seed,
                     _dt(dtok), st)
        if masked is not None:
            dtok = masked
        if patches is not None and dtok.dtype == torch.bfloat16:
            # dW = dtok^T . P over all B*T token rows, with P the patch matrix widened by a zero row per image (the CLS
            # row): the TN kernel then takes dtok as it lies in memory -- no transposed copies of a 34 MB tensor
            dwf = _weight_grad(dtok.view(B * T, E), patches, B * T, E, K)
            if not early:   # (with a batch in flight the end-of-pass callback joins: the spectral fold's backward overlaps it too)
                join_side_stream()
            return None, dwf, dbias, dcls, dpos_full, None, None, None, None
        rows = B * Np
        ld = (rows + 7) // 8 * 8
        dyt = torch.empty((E, ld), dtype=dtok.dtype, device=dev)
        _native.call("spv_cast_transpose", _p(dtok), _dt(dtok), _p(dyt), _dt(dyt), rows, E, ld, Np, T, 1, st)
        pt = torch.empty((K, ld), dtype=dtok.dtype, device=dev)
        if img is None:  # uint8 input: the forward kept the normalised patch matrix instead of a float image
            _native.call("spv_cast_transpose", _p(patches), _dt(patches), _p(pt), _dt(pt), rows, K, ld, Np, T, 1, st)  # token rows -> patch rows
        else:
            _native.call("spv_patchify", _p(img), _p(pt), B, C, H, W, patch, ld, 1, _dt(pt), st)
        dwf = torch.empty((E, K), dtype=torch.float32, device=dev)
        tiles = ((E + 127) // 128) * ((K + 127) // 128)
        splits = max(1, min(1024 // tiles, (ld + 511) // 512))
        ws = torch.empty((splits * E * K,), dtype=torch.float32, device=dev) if splits > 1 else None
        _gemm(dyt, pt, None, dwf, E, K, ld, ld, ld, K, 0, splits, ws)
        return None, dwf, dbias, dcls, dpos_full, None, None, None, None


# CIFAR-100 statistics of the reference loader (spectre_vit/repl/train.py:109-112)
CIFAR100_MEAN = (0.5071, 0.4867, 0.4408)
CIFAR100_STD = (0.2675, 0.2565, 0.2761)


class PixelNorm:
    """(mean, 1/std) per channel for uint8 NHWC input, kept as device tensors."""

    def __init__(self, mean=CIFAR100_MEAN, std=CIFAR100_STD):
        self.mean = tuple(float(m) for m in mean)
        self.std = tuple(float(v) for v in std)
        self._cache = {}

    def tensors(self, device, channels):
        if len(self.mean) != channels or len(self.std) != channels:
            raise ValueError(f"pixel normalisation has {len(self.mean)} channels, the image has {channels}")
        t = self._cache.get(device)
        if t is None:
            t = (torch.tensor(self.mean, dtype=torch.float32, device=device),
                 torch.tensor([1.0 / v for v in self.std], dtype=torch.float32, device=device))
            self._cache[device] = t
        return t

    def __deepcopy__(self, memo):
        return PixelNorm(self.mean, self.std)


def patch_embed(x, w_full, bias, cls, pos, patch, pixel_norm, p_drop=0.0):
    """float NCHW or uint8 NHWC images -> token tensor (B, 1 + patches, E) [-> dropout(p_drop)]."""
    if x.dtype == torch.uint8:
        dt = torch.bfloat16 if torch.is_autocast_enabled("cuda") else torch.float32
        norm = pixel_norm.tensors(x.device, x.shape[-1])
        return PatchEmbedFn.apply(x, w_full, bias, cls, pos, patch, dt, norm, float(p_drop))
    return PatchEmbedFn.apply(x, w_full, bias, cls, pos, patch, compute_dtype(x), None, float(p_drop))


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        _require_gpu(x)
        xc = x.contiguous()
        seed = _new_seed()
        y = torch.empty_like(xc)
        _native.call("spv_dropout", _p(xc), _p(y), xc.numel(), float(p), seed, _dt(xc), _stream())
        ctx.meta = (float(p), seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed = ctx.meta
        dyc = dy.contiguous()
        dx = torch.empty_like(dyc)
        _native.call("spv_dropout", _p(dyc), _p(dx), dyc.numel(), p, seed, _dt(dyc), _stream())
        return dx, None


def dropout(x, p, training):
    if not training or p <= 0.0:
        return x
    return DropoutFn.apply(x, p)


class AddFn(torch.autograd.Function):
    """out = a + b through spv_axpby (the encoder's global residual, spectre.py:103)."""

    @staticmethod
    def forward(ctx, a, b):
        _require_gpu(a, b)
        ac, bc = a.contiguous(), b.contiguous()
        out = torch.empty_like(ac)
        _native.call("spv_axpby", _p(ac), _p(bc), _p(out), 1.0, 1.0, ac.numel(), _dt(ac), _stream())
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


class TapClsFn(torch.autograd.Function):
    """x -> (x, x[:, 0, :].copy): the encoder's global residual `output + src` (spectre.py:103) is consumed at the CLS row only
    (spectre.py:198), so SpectreViT takes src's CLS row here, at the entrance of the layer stack.  src then has ONE consumer in the
    autograd graph: its gradient is not accumulated from two full (B, N, E) tensors (a 102 MB torch add per step); the CLS row's
    gradient is added in place to row 0 of the stack's input gradient instead."""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        ctx.key = getattr(x, "_spv_embed_key", None)   # x is a PatchEmbedFn output: its backward adds the CLS rows in its own pass
        return x.view_as(x), x[:, 0, :]   # a strided view: the class head reads the CLS rows where they lie

    @staticmethod
    def backward(ctx, gx, gcls):
        if gcls is None:
            return gx
        if gx is None:
            raise RuntimeError("TapClsFn: the layer stack produced no input gradient")
        if ctx.key is not None:
            _cls_grad_stash[ctx.key] = gcls
            return gx
        if not gx.is_contiguous():
            gx = gx.contiguous()
        gx[:, 0, :] += gcls.to(gx.dtype)  # in place: this edge owns the tensor (it was written for it by the first layer's backward)
        return gx


_cls_grad_bufs = {}
# SpectreViT reads the CLS row of the stack's output and nothing else (reference spectre.py:198), and everything behind the LAST
# layer's token mixer works row by row (LayerNorm, SpectreLinear, residual): that layer's feed-forward half only has to exist at the
# CLS rows -- the same logits, loss and gradients (the other rows' gradients are exactly zero in the reference too).
# SPV_FULL_LAST_LAYER=1 computes every row, as the reference does.
LAST_LAYER_CLS_ONLY = os.environ.get("SPV_FULL_LAST_LAYER", "0") == "0"


class TakeClsFn(torch.autograd.Function):
    """x (B, N, E) -> x[:, 0, :] as a contiguous (B, E) tensor.  Backward: the dense gradient that is zero off the CLS row, in the kept
    buffer of _cls_row_gradient (its consumer, the token mixer's backward, only reads it)."""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        ctx.meta = (x.shape, x.dtype)
        return x[:, 0, :].contiguous()

    @staticmethod
    def backward(ctx, g):
        shape, dtype = ctx.meta
        return _cls_row_gradient(shape, dtype, g.device, g)


class PermutClsFn(torch.autograd.Function):
    """x (B, N, E) -> (token row 0 of the MHPermutMix gather (B, n), x[:, 0, :] (B, E)): the two things the LAST layer of a stack needs
    of its input when the consumer reads the CLS row only (MHPermutMix.forward_cls).  Backward: ONE dense input gradient -- the CLS
    row's own gradient in row 0, zero elsewhere, + the n scattered values -- instead of two (B, N, E) tensors for autograd to add."""

    @staticmethod
    def forward(ctx, x, table, n):
        _require_gpu(x)
        B, N, E = x.shape
        xc = x.contiguous()
        g0 = torch.empty((B, n), dtype=xc.dtype, device=xc.device)
        x0 = torch.empty((B, E), dtype=xc.dtype, device=xc.device)
        _native.call("spv_permut_row0_fwd", _p(xc), _p(table), _p(g0), _p(x0), B, N * E, n, E, _dt(xc), _stream())
        ctx.table = table
        ctx.meta = (B, N, E, n, xc.dtype)
        return g0, x0

    @staticmethod
    def backward(ctx, dg0, dx0):
        B, N, E, n, dtype = ctx.meta
        dev = ctx.table.device
        dg0 = torch.zeros((B, n), dtype=dtype, device=dev) if dg0 is None else dg0.to(dtype).contiguous()
        dx0 = torch.zeros((B, E), dtype=dtype, device=dev) if dx0 is None else dx0.to(dtype).contiguous()
        dx = torch.empty((B, N, E), dtype=dtype, device=dev)
        _native.call("spv_permut_row0_bwd", _p(dg0), _p(dx0), _p(ctx.table), _p(dx), B, N * E, n, E, _DT[dtype], _stream())
        return dx, None, None


def _cls_row_gradient(shape, dtype, dev, rows):
    """The stack's output gradient when only the CLS rows carry one: a (B, N, E) tensor that is zero off row 0.  The buffer is kept
    across steps -- nothing ever writes its other rows (the consumers read it; TapClsFn adds in place to row 0 only) -- so a step
    costs the CLS-row copy, not a 34 MB fill."""
    if shape[1] == 1:   # the stack handed over its CLS rows only (LAST_LAYER_CLS_ONLY): the gradient is those rows
        return rows.to(dtype).reshape(shape)
    key = (tuple(shape), dtype, dev.index)   # not per stream: a graph capture runs on its own stream and must find the warm-up's buffer
    full = _cls_grad_bufs.get(key)
    if full is None:
        if len(_cls_grad_bufs) > 8:
            _cls_grad_bufs.clear()
        full = torch.zeros(shape, dtype=dtype, device=dev)
        _cls_grad_bufs[key] = full
    full[:, 0, :] = rows
    return full


class ClsAddFn(torch.autograd.Function):
    """(output, src_cls) -> output[:, 0, :] + src_cls: the only rows of `output + src` the model reads.  Backward hands the stack a
    dense gradient that is zero off the CLS row (what slicing the full sum gave it before)."""

    @staticmethod
    def forward(ctx, out, src_cls):
        _require_gpu(out, src_cls)
        ctx.meta = (out.shape, out.dtype)
        return out[:, 0, :] + src_cls

    @staticmethod
    def backward(ctx, g):
        shape, dtype = ctx.meta
        return _cls_row_gradient(shape, dtype, g.device, g), g


# ------------------------------------------------------------------------------------------------
# the classifier end of the step: class head over the CLS rows + mean cross-entropy (csrc/spv_head.hip)
# ------------------------------------------------------------------------------------------------
def small_head_ok(rows: int, n: int, k: int) -> bool:
    return bool(_native.call("spv_small_sl_supported", int(rows), int(n), int(k)))


class ClsHeadFn(torch.autograd.Function):
    """(output, src_cls, head parameters) -> (logits fp32, CLS features fp32): SpectreLinear((output + src)[:, 0]) of the reference
    (spectre.py:198-202, layers.py:95-101) as ONE launch over the fp32 master weights; backward two launches.  The generic path
    (ClsAddFn + cast + fp32 shadow + split-K GEMM + reduce + tail, and five launches more in the backward) costs the same
    arithmetic 13 launches at their 4-6 us floor."""

    @staticmethod
    def forward(ctx, out, src_cls, weight, bias, gamma, beta):
        _require_gpu(out, src_cls, weight)
        B, N, E = out.shape
        n = weight.shape[0]
        if not out.is_contiguous():
            out = out.contiguous()
        if src_cls.stride(-1) != 1:
            src_cls = src_cls.contiguous()
        if src_cls.dtype != out.dtype or weight.dtype != torch.float32 or not weight.is_contiguous():
            raise ValueError("ClsHeadFn: src_cls must have the stack's dtype and the head weight must be contiguous fp32")
        dev = out.device
        logits = torch.empty((B, n), dtype=torch.float32, device=dev)
        h = torch.empty((B, n), dtype=torch.float32, device=dev)
        xs = torch.empty((B, E), dtype=torch.float32, device=dev)
        mean = torch.empty((B,), dtype=torch.float32, device=dev)
        rstd = torch.empty((B,), dtype=torch.float32, device=dev)
        _native.call("spv_small_sl_fwd", _p(out), N * E, _p(src_cls), src_cls.stride(0), _p(weight), _p(bias), _p(gamma), _p(beta), _p(logits), _p(h),
                     _p(xs), _p(mean), _p(rstd), B, n, E, _dt(out), _stream())
        ctx.save_for_backward(h, xs, mean, rstd, weight, gamma, beta)
        ctx.meta = (out.shape, out.dtype)
        ctx.sinks = (_sink(weight), _sink(bias), _sink(gamma), _sink(beta))
        ctx.set_materialize_grads(False)
        return logits, xs

    @staticmethod
    def backward(ctx, dlogits, dfeats):
        h, xs, mean, rstd, weight, gamma, beta = ctx.saved_tensors
        shape, dtype = ctx.meta
        B, n = h.shape
        E = xs.shape[1]
        dev = h.device
        if dlogits is None:
            dlogits = torch.zeros_like(h)
        dlogits = dlogits.contiguous().float()
        s_w, s_b, s_g, s_be = ctx.sinks
        dh = torch.empty_like(h)
        dx = torch.empty((B, E), dtype=dtype, device=dev)
        dw = _grad_buf(s_w, (n, E), dev)
        dbias = _grad_buf(s_b, (n,), dev)
        dgamma = _grad_buf(s_g, (n,), dev)
        dbeta = _grad_buf(s_be, (n,), dev)
        partials = torch.empty((_native.call("spv_small_sl_partial_floats", B, n),), dtype=torch.float32, device=dev)
        _native.call("spv_small_sl_bwd", _p(dlogits), _p(h), _p(xs), _p(mean), _p(rstd), _p(weight), _p(gamma), _p(beta), _p(dh), _p(dx),
                     _p(dw), _p(dgamma), _p(dbeta), _p(dbias), _p(partials), B, n, E, _DT[dtype], _stream())
        if dfeats is not None:
            dx = dx + dfeats.to(dx.dtype)
        return _cls_row_gradient(shape, dtype, dev, dx), dx, dw, dbias, dgamma, dbeta  # the stack's gradient is dense: zero off the CLS row


_ce_workspaces = {}


def _ce_workspace(dev):
    key = dev.index   # one loss per step and device; not per stream, so that a graph capture reuses the warm-up's (zeroed) counter
    ws = _ce_workspaces.get(key)
    if ws is None:
        ws = torch.zeros((_native.call("spv_cross_entropy_workspace_floats"),), dtype=torch.float32, device=dev)
        _ce_workspaces[key] = ws
    return ws


class CrossEntropyFn(torch.autograd.Function):
    """nn.CrossEntropyLoss() with its defaults (mean over rows; reference repl/train.py:196,226): one launch forward (row-wise
    logsumexp, deterministic sum), one backward -- stock torch runs log_softmax, nll_loss, two fills and their two backwards."""

    @staticmethod
    def forward(ctx, logits, labels):
        _require_gpu(logits, labels)
        if logits.dim() != 2 or logits.dtype != torch.float32 or labels.dtype != torch.int64 or labels.shape != logits.shape[:1]:
            raise ValueError("cross_entropy: fp32 logits [rows, classes] and int64 labels [rows] expected")
        z = logits.contiguous()
        y = labels.contiguous()
        rows, C = z.shape
        lse = torch.empty((rows,), dtype=torch.float32, device=z.device)
        loss = torch.empty((), dtype=torch.float32, device=z.device)
        _native.call("spv_cross_entropy_fwd", _p(z), _p(y), _p(lse), _p(loss), _p(_ce_workspace(z.device)), rows, C, _stream())
        ctx.save_for_backward(z, y, lse)
        return loss

    @staticmethod
    def backward(ctx, go):
        z, y, lse = ctx.saved_tensors
        go = go.reshape(1).float().contiguous()
        dz = torch.empty_like(z)
        _native.call("spv_cross_entropy_bwd", _p(z), _p(y), _p(lse), _p(go), _p(dz), z.shape[0], z.shape[1], _stream())
        return dz, None


def cross_entropy(logits, labels):
    return CrossEntropyFn.apply(logits, labels)


# ------------------------------------------------------------------------------------------------
# baseline ViT pieces: plain Linear, GELU, softmax attention core  (reference vit.py:30-40)
# ------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y = x W^T + b on the MFMA GEMM (nn.Linear: in_proj / out_proj / linear1 / linear2 / the ViT head)."""

    @staticmethod
    def forward(ctx, x, weight, bias, out_fp32):
        _require_gpu(x, weight)
        n, k = weight.shape
        x2 = x.reshape(-1, k)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        rows = x2.shape[0]
        mult = 8 if x2.dtype == torch.bfloat16 else 4
        if n % mult or k % mult:
            raise ValueError(f"Linear({k}->{n}) in {x2.dtype}: feature counts must be multiples of {mult}")
        wc, wt = _shadows.get(weight, x2.dtype)
        y = torch.empty((rows, n), dtype=torch.float32 if out_fp32 else x2.dtype, device=x2.device)
        _gemm(x2, wc, bias, y, rows, n, k, k, k, n)
        ctx.save_for_backward(x2, weight)
        ctx.wt = wt
        ctx.sinks = (_sink(weight), _sink(bias))
        ctx.meta = (x.shape, rows, n, k, bias is not None)
        return y.reshape(*x.shape[:-1], n)

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        shape, rows, n, k, has_bias = ctx.meta
        dy2 = dy.reshape(rows, n)
        if dy2.dtype != x2.dtype:
            dy2 = _raw_cast(dy2, x2.dtype)
        elif not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dw = _weight_grad(dy2, x2, rows, n, k, ctx.sinks[0])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x2)
            _gemm(dy2, ctx.wt, None, dx, rows, k, n, n, ctx.wt.shape[1], k)
            dx = dx.reshape(shape)
        db = None
        if has_bias:
            db = _grad_buf(ctx.sinks[1], (n,), x2.device)
            part = torch.empty((min(rows, 512) * n,), dtype=torch.float32, device=x2.device)
            _native.call("spv_colsum", _p(dy2), _p(db), _p(part), rows, n, _dt(dy2), _stream())
        join_side_stream()
        return dx, dw, db, None


def linear(x, weight, bias=None, out_fp32=False):
    return LinearFn.apply(x, weight, bias, out_fp32)


class GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        xc = x.contiguous()
        y = torch.empty_like(xc)
        _native.call("spv_gelu_fwd", _p(xc), _p(y), xc.numel(), _dt(xc), _stream())
        ctx.save_for_backward(xc)
        return y

    @staticmethod
    def backward(ctx, dy):
        (xc,) = ctx.saved_tensors
        dyc = dy.contiguous()
        dx = torch.empty_like(xc)
        _native.call("spv_gelu_bwd", _p(dyc), _p(xc), _p(dx), xc.numel(), _dt(xc), _stream())
        return dx


class AttentionFn(torch.autograd.Function):
    """ctx = dropout(softmax(q k^T / sqrt(hd))) v per (sequence, head); qkv [seqs, len, 3E] -> [seqs, len, E]."""

    @staticmethod
    def forward(ctx, qkv, heads, p_drop):
        _require_gpu(qkv)
        seqs, length, e3 = qkv.shape
        E = e3 // 3
        hd = E // heads
        q = qkv.contiguous()
        out = torch.empty((seqs, length, E), dtype=q.dtype, device=q.device)
        probs = torch.empty((seqs, heads, length, length), dtype=q.dtype, device=q.device)
        seed = _new_seed() if p_drop > 0.0 else 0
        _native.call("spv_attention_fwd", _p(q), _p(out), _p(probs), seqs, length, heads, hd, _dt(q), float(p_drop), seed, _stream())
        ctx.save_for_backward(q, probs)
        ctx.meta = (seqs, length, heads, hd, float(p_drop), seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, probs = ctx.saved_tensors
        seqs, length, heads, hd, p_drop, seed = ctx.meta
        d = dout.contiguous()
        dqkv = torch.empty_like(q)
        ds = torch.empty_like(probs)
        _native.call("spv_attention_bwd", _p(d), _p(q), _p(probs), _p(ds), _p(dqkv), seqs, length, heads, hd, _dt(q), p_drop, seed,
                     _stream())
        return dqkv, None, None


# ------------------------------------------------------------------------------------------------
# fused halves of the encoder layer: same kernels, hand-written backward so that the residual-stream gradients are
# folded into the producing kernels instead of being summed by separate elementwise passes
# ------------------------------------------------------------------------------------------------
class FFResidualFn(torch.autograd.Function):
    """x2 = LayerNorm2(x1 + SpectreLinear3(SpectreLinear1(x1)))   (reference spectre.py:67,70-73)."""

    @staticmethod
    def forward(ctx, x1, w1, b1, g1, be1, w3, b3, g3, be3, n2w, n2b, p_drop):
        _require_gpu(x1, w1)
        shape = x1.shape
        x2d = x1.reshape(-1, shape[-1])
        if not x2d.is_contiguous():
            x2d = x2d.contiguous()
        f1, s1 = _sl_forward(x2d, w1, b1, g1, be1, p_drop, False)
        ctx.shape = shape
        n3, k3 = w3.shape
        if _native.call("spv_tail_ln_supported", n3, k3, _dt(f1)):
            # linear3's GEMM, then ONE row kernel: LayerNorm/GELU/pooled skip/dropout of the SpectreLinear tail, + x1, LayerNorm-2
            rows, dt, dev = f1.shape[0], f1.dtype, f1.device
            wc3, wt3 = _shadows.get(w3, dt)
            h3 = torch.empty((rows, n3), dtype=dt, device=dev)
            _gemm(f1, wc3, b3, h3, rows, n3, k3, k3, k3, n3)
            f3 = torch.empty_like(h3)
            out = torch.empty_like(h3)
            mean3, rstd3, mean2, rstd2 = (torch.empty((rows,), dtype=torch.float32, device=dev) for _ in range(4))
            seed = _new_seed() if p_drop > 0.0 else 0
            _native.call("spv_spectre_tail_ln_fwd", _p(h3), _p(f1), _p(g3), _p(be3), _p(f3), _p(mean3), _p(rstd3), _p(x2d), _p(n2w),
                         _p(n2b), _p(out), _p(mean2), _p(rstd2), rows, n3, k3, _dt(h3), float(p_drop), seed, _stream())
            s3 = (f1, h3, mean3, rstd3, g3, be3, wt3, (_sink(w3), _sink(b3), _sink(g3), _sink(be3)), rows, n3, k3, float(p_drop), seed)
            ctx.saved = (s1, s3, ("fused", f3, x2d, mean2, rstd2, n2w, (_sink(n2w), _sink(n2b))))
            return out.reshape(shape)
        f3, s3 = _sl_forward(f1, w3, b3, g3, be3, p_drop, False)
        out, sn = _addln_forward(f3, x2d, n2w, n2b, 1)
        ctx.saved = (s1, s3, sn)
        return out.reshape(shape)

    @staticmethod
    def backward(ctx, dout):
        s1, s3, sn = ctx.saved
        if isinstance(sn[0], str):  # ("fused", ...): LayerNorm-2 backward inside the linear3 tail backward
            _, f3, x1, mean2, rstd2, n2w, sinks2 = sn
            f1, h3, mean3, rstd3, g3, be3, wt3, sinks3, rows, n, k, p_drop, seed = s3
            dev = f1.device
            d2 = dout.reshape(rows, n)
            if not d2.is_contiguous():
                d2 = d2.contiguous()
            ds = torch.empty_like(f3)
            dh3 = torch.empty_like(h3)
            df1 = torch.empty_like(f1)
            s_w, s_b, s_g, s_be = sinks3
            dg3, dbe3, db3 = _grad_buf(s_g, (n,), dev), _grad_buf(s_be, (n,), dev), _grad_buf(s_b, (n,), dev)
            dn2w, dn2b = _grad_buf(sinks2[0], (n,), dev), _grad_buf(sinks2[1], (n,), dev)
            partials = torch.empty((_native.call("spv_tail_ln_partial_floats", n),), dtype=torch.float32, device=dev)
            # linear3's skip gradient (its transposed pooling) is taken by linear1's tail backward from `ds` itself when the
            # shapes allow: df1 is then a plain GEMM output (no [rows, 768] tensor written here and re-read by the GEMM)
            defer = _native.call("spv_tail_up_supported", s1[9], s1[10], _dt(h3)) and s1[9] == k
            ride = _fold_rides(dh3.dtype, rows, n, k)   # the five column sums' fold rides in the weight gradient's split-K reduce
            pp = (lambda t: 0) if ride else _p
            _native.call("spv_spectre_tail_ln_bwd", _p(d2), _p(f3), _p(x1), _p(mean2), _p(rstd2), _p(n2w), _p(ds), pp(dn2w), pp(dn2b),
                         _p(h3), _p(mean3), _p(rstd3), _p(g3), _p(be3), _p(dh3), 0 if defer else _p(df1), pp(dg3), pp(dbe3), pp(db3),
                         _p(partials), rows, n, k, _dt(h3), p_drop, seed, _stream())
            dw3 = _weight_grad(dh3, f1, rows, n, k, s_w, _fold_job(partials, (dg3, dbe3, db3, dn2w, dn2b), rows, n) if ride else None,
                               ride and _sunk((dg3, dbe3, db3, dn2w, dn2b), (s_g, s_be, s_b, sinks2[0], sinks2[1])))
            _gemm(dh3, wt3, None, df1, rows, k, n, n, wt3.shape[1], k, accumulate=0 if defer else 1)
            if defer:
                dx1, dw1, db1, dg1, dbe1 = _sl_backward(df1, s1, True, dx_add=ds, up=(ds, p_drop, seed))
                join_side_stream()
                return dx1.reshape(ctx.shape), dw1, db1, dg1, dbe1, dw3, db3, dg3, dbe3, dn2w, dn2b, None
        else:
            rows, n = sn[6], sn[7]
            ds, dn2w, dn2b = _addln_backward(dout.reshape(rows, n), sn)      # d(x1 + f3)
            df1, dw3, db3, dg3, dbe3 = _sl_backward(ds, s3, True)
        dx1, dw1, db1, dg1, dbe1 = _sl_backward(df1, s1, True, dx_add=ds)  # + the residual path, folded in
        join_side_stream()
        return dx1.reshape(ctx.shape), dw1, db1, dg1, dbe1, dw3, db3, dg3, dbe3, dn2w, dn2b, None


class FNetResidualFn(torch.autograd.Function):
    """x1 = LayerNorm1(Re(fft2(x))) + x   (reference spectre.py:66 with the 'fft_bare' mixer)."""

    @staticmethod
    def forward(ctx, x, n1w, n1b):
        _require_gpu(x)
        B, N, D = x.shape
        xc = x.contiguous()
        ctx.shape = (B, N, D)
        if _native.call("spv_fnet_ln_supported", N, D, _dt(xc)):
            # one kernel: mixer, LayerNorm statistics per finished row, residual
            dev = xc.device
            m = torch.empty_like(xc)
            out = torch.empty_like(xc)
            mean = torch.empty((B * N,), dtype=torch.float32, device=dev)
            rstd = torch.empty_like(mean)
            tw = _fnet_twiddle(N, dev)
            _native.call("spv_fnet_ln_fwd", _p(xc), _p(m), _p(out), _p(n1w), _p(n1b), _p(mean), _p(rstd), _p(tw), B, N, D, _dt(xc),
                         _stream())
            ctx.saved = ("fused", m, mean, rstd, n1w, (_sink(n1w), _sink(n1b)))
            return out
        m = _fnet_raw(xc)
        out, sn = _addln_forward(m.reshape(-1, D), xc.reshape(-1, D), n1w, n1b, 0)
        ctx.saved = sn
        return out.reshape(B, N, D)

    @staticmethod
    def backward(ctx, dout):
        sn = ctx.saved
        B, N, D = ctx.shape
        if WGRAD_PER_LAYER:
            start_held_wgrads()   # this layer's two weight gradients (held by the feed-forward half's backward, just done): side stream
        d2 = dout.reshape(-1, D)
        if not d2.is_contiguous():
            d2 = d2.contiguous()
        if isinstance(sn[0], str):  # ("fused", ...)
            _, m, mean, rstd, gamma, sinks = sn
            dev = m.device
            dx = torch.empty_like(m)
            dn1w = _grad_buf(sinks[0], (D,), dev)
            dn1b = _grad_buf(sinks[1], (D,), dev)
            partials = torch.empty((B * 2 * D,), dtype=torch.float32, device=dev)
            tw = _fnet_twiddle(N, dev)
            # the fold of the per-sample column sums travels with the next weight-gradient reduce (the layer below's); only into
            # sink memory -- a fresh tensor would be copied by autograd before the fold has run
            held = _hold_fold(partials, (dn1w, dn1b), sinks, B, D)
            _native.call("spv_fnet_ln_bwd", _p(d2), _p(m), _p(mean), _p(rstd), _p(gamma), _p(dx), 0 if held else _p(dn1w),
                         0 if held else _p(dn1b), _p(partials), _p(tw), B, N, D, _dt(m), _stream())
            return dx, dn1w, dn1b
        dm, dn1w, dn1b = _addln_backward(d2, sn)
        dx = _fnet_raw(dm.reshape(B, N, D), add_in=d2)  # symmetric operator; + the residual gradient, folded in
        return dx, dn1w, dn1b


def fnet_cls_ok(x):
    """row 0 of the FFT mixer + LayerNorm-1 + residual can come from the one-FFT kernels (spv_fnet_cls_fwd / _bwd)"""
    return x.is_cuda and x.dim() == 3 and x.dtype in _DT and bool(_native.call("spv_fnet_cls_supported", x.shape[1], x.shape[2], _dt(x)))


class FNetClsFn(torch.autograd.Function):
    """x (B, N, D) -> (LayerNorm1(Re(fft2(x))) + x)[:, 0, :] as (B, D): what the LAST layer of a stack needs of FNetResidualFn when the
    consumer reads the CLS row only.  Token frequency 0 is the sum over tokens, so the row is ONE D-point FFT of the token sum
    (spv_fnet_cls_fwd: one pass over x); the backward hands every token the same spectrum (+ the residual's gradient in row 0)."""

    @staticmethod
    def forward(ctx, x, n1w, n1b):
        _require_gpu(x)
        B, N, D = x.shape
        xc = x.contiguous()
        dev = xc.device
        out = torch.empty((B, D), dtype=xc.dtype, device=dev)
        m0 = torch.empty((B, D), dtype=torch.float32, device=dev)
        mean = torch.empty((B,), dtype=torch.float32, device=dev)
        rstd = torch.empty_like(mean)
        _native.call("spv_fnet_cls_fwd", _p(xc), _p(n1w), _p(n1b), _p(out), _p(m0), _p(mean), _p(rstd), B, N, D, _dt(xc), _stream())
        ctx.saved = (m0, mean, rstd, n1w, (_sink(n1w), _sink(n1b)))
        ctx.meta = (B, N, D, xc.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        m0, mean, rstd, n1w, sinks = ctx.saved
        B, N, D, dtype = ctx.meta
        dev = m0.device
        g = g.to(dtype).contiguous()
        dx = torch.empty((B, N, D), dtype=dtype, device=dev)
        dn1w = _grad_buf(sinks[0], (D,), dev)
        dn1b = _grad_buf(sinks[1], (D,), dev)
        partials = torch.empty((B * 2 * D,), dtype=torch.float32, device=dev)
        _native.call("spv_fnet_cls_bwd", _p(g), _p(m0), _p(mean), _p(rstd), _p(n1w), _p(dx), _p(partials), B, N, D, _DT[dtype], _stream())
        if not _hold_fold(partials, (dn1w, dn1b), sinks, B, D):   # the batch sums of dgamma / dbeta: with the next reduce, or now
            arr = _fold_array([(partials, (dn1w, dn1b), B, D)])
            _native.call("spv_fold_multi", ctypes.addressof(arr), 1, _stream())
        return dx, dn1w, dn1b


class HaarResidualFn(torch.autograd.Function):
    """x1 = LayerNorm1(haar(x)) + x, one-level Haar DWT along the embedding axis (reference spectre.py:66 with the 'dwt_embed' mixer of
    BASELINE config 3): one row kernel each way (spv_haar_ln_fwd / _bwd; the transform is lane-local, nothing of the mixer is stored)."""

    @staticmethod
    def forward(ctx, x, n1w, n1b):
        _require_gpu(x)
        xc = x.contiguous()
        D = xc.shape[-1]
        rows = xc.numel() // D
        dev = xc.device
        out = torch.empty_like(xc)
        mean = torch.empty((rows,), dtype=torch.float32, device=dev)
        rstd = torch.empty_like(mean)
        _native.call("spv_haar_ln_fwd", _p(xc), _p(n1w), _p(n1b), _p(out), _p(mean), _p(rstd), rows, D, _dt(xc), _stream())
        ctx.save_for_backward(xc, mean, rstd, n1w)
        ctx.sinks = (_sink(n1w), _sink(n1b))
        return out

    @staticmethod
    def backward(ctx, dout):
        xc, mean, rstd, n1w = ctx.saved_tensors
        D = xc.shape[-1]
        rows = xc.numel() // D
        dev = xc.device
        d2 = dout.contiguous()
        dx = torch.empty_like(xc)
        dn1w = _grad_buf(ctx.sinks[0], (D,), dev)
        dn1b = _grad_buf(ctx.sinks[1], (D,), dev)
        partials = torch.empty((_native.call("spv_rowop_partial_floats", D),), dtype=torch.float32, device=dev)
        # the fold of the column sums travels with the next weight-gradient reduce of the backward pass (sink memory only: _hold_fold)
        held = _hold_fold(partials, (dn1w, dn1b), ctx.sinks, _native.call("spv_tail_bwd_parts", rows), D)
        _native.call("spv_haar_ln_bwd", _p(d2), _p(xc), _p(mean), _p(rstd), _p(n1w), _p(dx), 0 if held else _p(dn1w), 0 if held else _p(dn1b),
                     _p(partials), rows, D, _dt(xc), _stream())
        return dx, dn1w, dn1b


def haar_ln_ok(x, axis, levels):
    return bool(axis == "embed" and levels == 1 and x.is_cuda and x.dtype == torch.bfloat16
                and _native.call("spv_haar_ln_supported", x.shape[-1], _dt(x)))


class PermutMixFn(torch.autograd.Function):
    """MHPermutMix as one autograd node: SpectreLinear(gather(x)) (reference layers.py:68-73).

    Forward: the gather kernel also emits the averages of every `heads` consecutive gathered elements, which is exactly
    the SpectreLinear skip (AdaptiveAvgPool1d(E) over E*heads channels), so the tail kernel does not re-read the 16x larger
    gathered tensor.  Backward: the transposed pooling is added in the data-gradient GEMM's epilogue instead of being
    written to and re-read from a (B*N, E*heads) buffer."""

    @staticmethod
    def forward(ctx, x, idx, heads, weight, bias, gamma, beta):
        _require_gpu(x, weight)
        B = x.shape[0]
        xc = x.contiguous()
        d = xc.numel() // B
        n, k = weight.shape
        total = heads * d
        rows = (B * total) // k
        dt = xc.dtype
        mult = 8 if dt == torch.bfloat16 else 4
        if n % mult or k % mult:
            raise ValueError(f"MHPermutMix linear ({k}->{n}) in {dt}: channel counts must be multiples of {mult}")
        pw = k // n if k % n == 0 else 0
        es = 2 if dt == torch.bfloat16 else 4
        can_pool = pw in (4, 8, 16, 32) and (d * es) % 16 == 0 and d * es <= 150 * 1024 and (total // 4) % 1024 == 0 and total % pw == 0
        if not can_pool and pw > 0 and total % pw == 0:   # rows longer than the LDS (Base / 224, window 12): the scatter gather pools too
            can_pool = bool(_native.call("spv_permut_pool_supported", heads, d, pw, _dt(xc)))
        dev = xc.device
        g = torch.empty((rows, k), dtype=dt, device=dev)
        pooled = torch.empty((rows, n), dtype=dt, device=dev) if can_pool else None
        st = _stream()
        _native.call("spv_permut_gather_fwd", _p(xc), _p(idx), _p(g), _p(pooled), pw, B, heads, d, _dt(xc), st)
        wc, wt = _shadows.get(weight, dt)
        h = torch.empty((rows, n), dtype=dt, device=dev)
        _gemm(g, wc, bias, h, rows, n, k, k, k, n)
        out = torch.empty((rows, n), dtype=dt, device=dev)
        mean = torch.empty((rows,), dtype=torch.float32, device=dev)
        rstd = torch.empty((rows,), dtype=torch.float32, device=dev)
        skip = pooled if can_pool else g
        _native.call("spv_spectre_tail_fwd", _p(h), _p(skip), _p(gamma), _p(beta), _p(out), _p(mean), _p(rstd), rows, n,
                     n if can_pool else k, _dt(h), _dt(out), 0.0, 0, st)
        ctx.save_for_backward(g, h, mean, rstd, gamma, beta)
        ctx.aux = (idx, wt, (_sink(weight), _sink(bias), _sink(gamma), _sink(beta)), x.shape, B, heads, d, rows, n, k, pw)
        return out.reshape(B, rows // B, n)

    @staticmethod
    def backward(ctx, dout):
        g, h, mean, rstd, gamma, beta = ctx.saved_tensors
        idx, wt, sinks, xshape, B, heads, d, rows, n, k, pw = ctx.aux
        dev = g.device
        st = _stream()
        d2 = dout.reshape(rows, n)
        if not d2.is_contiguous():
            d2 = d2.contiguous()
        s_w, s_b, s_g, s_be = sinks
        dh = torch.empty_like(h)
        dgamma = _grad_buf(s_g, (n,), dev)
        dbeta = _grad_buf(s_be, (n,), dev)
        dbias = _grad_buf(s_b, (n,), dev)
        partials = torch.empty((_native.call("spv_rowop_partial_floats", n),), dtype=torch.float32, device=dev)
        dg = torch.empty_like(g)
        fast = pw > 0
        _native.call("spv_spectre_tail_bwd", _p(d2), _p(h), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dh), 0 if fast else _p(dg),
                     _p(dgamma), _p(dbeta), _p(dbias), _p(partials), rows, n, k, _dt(h), _dt(d2), 0.0, 0, 0, st)
        dw = _weight_grad(dh, g, rows, n, k, s_w)  # side stream: overlaps the data gradient and the inverse gather
        if fast:
            _native.call("spv_gemm_nt_pool_bwd", _p(dh), _p(wt), _p(dg), _p(d2), pw, rows, k, n, n, wt.shape[1], k, _dt(dh),
                         _dt(dg), _dt(d2), st)
        else:
            _gemm(dh, wt, None, dg, rows, k, n, n, wt.shape[1], k, accumulate=1)
        dx = torch.empty((B, d), dtype=g.dtype, device=dev)
        _native.call("spv_permut_gather_bwd", _p(dg), _p(idx), _p(dx), B, heads, d, _dt(dg), st)
        join_side_stream()
        return dx.reshape(xshape), None, None, dw, dbias, dgamma, dbeta
