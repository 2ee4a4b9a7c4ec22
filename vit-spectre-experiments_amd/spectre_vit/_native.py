"""ctypes binding of libspv_hip.so (C-ABI declared in include/spv.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C vit-spectre-experiments_amd/csrc``.
There is NO fallback: if the shared object is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os

# torch must be imported BEFORE libspv_hip.so is dlopen'ed: the PyTorch-ROCm wheel bundles its own HIP runtime
# (torch/lib/libamdhip64.so); loading ours first would bind it to /opt/rocm's copy and leave two HIP runtimes in one
# process (the second one reports "no ROCm-capable device").  With torch first, both share torch's runtime, so
# torch's streams / allocations are valid handles in our kernels' launches.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPV_LIB_PATH: a second build of the same ABI, for A/B runs inside one GPU job (tools/); never a fallback
LIB_PATH = os.environ.get("SPV_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "lib", "libspv_hip.so")

c_vp, c_i, c_i64, c_u64, c_f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint64, ctypes.c_float

# indices of include/spv.h's SPV_PATH_* enum (dispatch census, test aid)
PATH = dict(gemm_strip=0, gemm_strip_acc=1, gemm_tn=2, tail_lc=3, tail_up=4, tail_ln=5, fnet_mfma=6, gather_lds=7, gemm_tn_dma=8, gemm_tn_wide=9,
            gemm_tn_batch=10, gemm_strip_pool=11, permut_row0=12, gemm_rows=13)

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/spv.h one to one
SIGNATURES = {
    "spv_version": [],
    "spv_last_error": [],
    "spv_path_count": [c_i],
    "spv_cast": [c_vp, c_i, c_vp, c_i, c_i64, c_vp],
    "spv_cast_transpose": [c_vp, c_i, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_vp],
    "spv_weight_shadows": [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp],
    "spv_weight_shadows_multi": [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_vp],
    "spv_gemm_nt": [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_vp, c_vp],
    "spv_gemm_nt_grouped_rows": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_vp],
    "spv_gemm_nt_grouped_rows_drop": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_u64, c_vp],
    "spv_set_reserved_cus": [c_i],
    "spv_gemm_tn": [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_vp, c_vp],
    "spv_spectre_tail_fwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_f, c_u64, c_vp],
    "spv_spectre_tail_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i,
                             c_i, c_f, c_u64, c_vp, c_vp],
    "spv_rowop_partial_floats": [c_i],
    "spv_tail_up_supported": [c_i, c_i, c_i],
    "spv_spectre_tail_bwd_up": [c_vp] * 12 + [c_i, c_i, c_i, c_i, c_i, c_f, c_u64, c_vp, c_vp, c_f, c_u64, c_vp],
    "spv_tail_ln_supported": [c_i, c_i, c_i],
    "spv_tail_ln_partial_floats": [c_i],
    "spv_spectre_tail_ln_fwd": [c_vp] * 13 + [c_i, c_i, c_i, c_i, c_f, c_u64, c_vp],
    "spv_spectre_tail_ln_bwd": [c_vp] * 20 + [c_i, c_i, c_i, c_i, c_f, c_u64, c_vp],
    "spv_add_layernorm_fwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp],
    "spv_add_layernorm_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp],
    "spv_permut_pack": [c_vp, c_vp, c_vp, c_i, c_i, c_vp],
    "spv_permut_row0_fwd": [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_vp],
    "spv_permut_row0_bwd": [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_vp],
    "spv_permut_table_words": [c_i, c_i],
    "spv_permut_pool_supported": [c_i, c_i, c_i, c_i],
    "spv_tail_bwd_parts": [c_i],
    "spv_haar_ln_supported": [c_i, c_i],
    "spv_haar_ln_fwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp],
    "spv_haar_ln_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp],
    "spv_embed_bwd_groups": [c_i],
    "spv_embed_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_f, c_u64, c_i, c_vp],
    "spv_fold_multi": [c_vp, c_i, c_vp],
    "spv_gemm_tn_fold": [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_vp, c_vp, c_i, c_vp],
    "spv_gemm_tn_batch": [c_vp, c_i, c_i, c_i, c_vp, c_vp, c_i, c_vp],
    "spv_gemm_tn_batch_part": [c_vp, c_i, c_i, c_i, c_vp, c_vp, c_i, c_i, c_vp],
    "spv_small_sl_supported": [c_i, c_i, c_i],
    "spv_small_sl_partial_floats": [c_i, c_i],
    "spv_small_sl_fwd": [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp],
    "spv_small_sl_bwd": [c_vp] * 15 + [c_i, c_i, c_i, c_i, c_vp],
    "spv_cross_entropy_workspace_floats": [],
    "spv_cross_entropy_fwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_vp],
    "spv_cross_entropy_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_vp],
    "spv_permut_gather_fwd": [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_vp],
    "spv_gemm_nt_pool_bwd": [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_vp],
    "spv_permut_gather_bwd": [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp],
    "spv_fnet_mix": [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp, c_vp],
    "spv_fnet_workspace_floats": [c_i, c_i, c_i],
    "spv_fnet_twiddle_floats": [c_i],
    "spv_fnet_make_twiddle": [c_vp, c_i, c_vp],
    "spv_fnet_ln_supported": [c_i, c_i, c_i],
    "spv_fnet_cls_supported": [c_i, c_i, c_i],
    "spv_fnet_cls_fwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp],
    "spv_fnet_cls_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp],
    "spv_fnet_ln_fwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp],
    "spv_fnet_ln_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp],
    "spv_rfft_real": [c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp],
    "spv_haar_dwt": [c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_vp, c_vp],
    "spv_patchify": [c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_vp],
    "spv_patchify_u8": [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_vp],
    "spv_embed_posbias": [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_vp],
    "spv_embed_cls_rows": [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_vp],
    "spv_spectral_fold": [c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp],
    "spv_spectral_fold_bf16": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp],
    "spv_spectral_fold_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp],
    "spv_dropout": [c_vp, c_vp, c_i64, c_f, c_u64, c_i, c_vp],
    "spv_attention_fwd": [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_f, c_u64, c_vp],
    "spv_attention_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_f, c_u64, c_vp],
    "spv_gelu_fwd": [c_vp, c_vp, c_i64, c_i, c_vp],
    "spv_gelu_bwd": [c_vp, c_vp, c_vp, c_i64, c_i, c_vp],
    "spv_colsum": [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_vp],
    "spv_set_seed_device_ptr": [c_vp],
    "spv_seed_advance": [c_vp, c_vp],
    "spv_adamw_multi": [c_vp, c_vp, c_vp, c_vp, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_vp, c_vp],
    "spv_fwht": [c_vp, c_vp, c_vp, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_vp],
    "spv_axpby": [c_vp, c_vp, c_vp, c_f, c_f, c_i64, c_i, c_vp],
}
_RESTYPES = {"spv_last_error": ctypes.c_char_p, "spv_path_count": ctypes.c_longlong, "spv_rowop_partial_floats": c_i64, "spv_fnet_workspace_floats": c_i64,
             "spv_fnet_twiddle_floats": c_i64, "spv_tail_ln_partial_floats": c_i64, "spv_permut_table_words": c_i64, "spv_small_sl_partial_floats": c_i64,
             "spv_cross_entropy_workspace_floats": c_i64}
_NO_STATUS = set(_RESTYPES) | {"spv_version", "spv_fnet_ln_supported", "spv_fnet_cls_supported", "spv_tail_ln_supported", "spv_tail_up_supported", "spv_small_sl_supported", "spv_tail_bwd_parts", "spv_embed_bwd_groups", "spv_haar_ln_supported", "spv_permut_pool_supported"}

class FoldJob(ctypes.Structure):
    """spv_fold_job (include/spv.h): the fold of a tail backward's partial column sums, handed to spv_gemm_tn_fold"""
    _fields_ = [("partials", c_vp), ("out", c_vp * 5), ("parts", c_i), ("nsum", c_i), ("n", c_i)]


class TnProblem(ctypes.Structure):
    """spv_tn_problem (include/spv.h): one weight gradient of a spv_gemm_tn_batch launch"""
    _fields_ = [("a", c_vp), ("b", c_vp), ("c", c_vp), ("m", c_i), ("n", c_i), ("lda", c_i), ("ldb", c_i), ("ldc", c_i), ("k", c_i)]


_lib = None
# live kernel timing (bench.py's roofline pass): when set, every entry point that launches on a stream is bracketed with HIP
# events by ``timer.bracket(name, ints, launch)``; ints = the integer arguments (shapes, dtype codes, flags) in header order
timer = None
hint = 0   # set by a caller right before a bracketed call whose algorithmic work is not a function of its integer arguments
_LAUNCHERS = {}


def load():
    """dlopen libspv_hip.so and attach the signatures.  Raises if the library is absent (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libspv_hip.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C vit-spectre-experiments_amd/csrc` (there is no CPU/PyTorch fallback for the Spectre-ViT kernels)")
        lib = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here == header/library mismatch
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, c_i)
            if argtypes and argtypes[-1] is c_vp and name not in _NO_STATUS and name != "spv_fnet_make_twiddle":
                _LAUNCHERS[name] = ([i for i, t in enumerate(argtypes) if t in (c_i, c_i64)],
                                    [i for i, t in enumerate(argtypes[:-1]) if t is c_vp])
        if lib.spv_version() != 1:
            raise RuntimeError(f"libspv_hip.so ABI version {lib.spv_version()} != 1")
        _lib = lib
    return _lib


def call(name, *args):
    """Invoke a status-returning entry point; raise RuntimeError(spv_last_error()) on failure."""
    lib = load()
    if timer is not None and name in _LAUNCHERS:
        global hint
        ii, pi = _LAUNCHERS[name]
        # key of a bracket: the integer arguments in header order, then a bit mask of the pointer arguments that are NULL (optional
        # outputs change a kernel's traffic), then the caller's work hint (algorithmic flops / bytes the integers do not determine)
        ints = tuple(int(args[i]) for i in ii) + (sum(1 << b for b, i in enumerate(pi) if not args[i]), hint)
        hint = 0
        box = []
        timer.bracket(name, ints, lambda: box.append(getattr(lib, name)(*args)))
        rc = box[0]
    else:
        rc = getattr(lib, name)(*args)
    if name in _NO_STATUS:
        return rc
    if rc != 0:
        raise RuntimeError(f"{name} failed: {lib.spv_last_error().decode()}")
    return 0
