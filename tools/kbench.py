#!/usr/bin/env python3
"""Per-kernel micro-benchmark at the Spectre-ViT-Small bs512 shapes (development tool; prints us + achieved GB/s / TFLOP/s).

    python tools/kbench.py [filter-substring] [--iters N]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-spectre-experiments_amd"))
import torch  # noqa: E402

from spectre_vit import _native, hip_ops as H  # noqa: E402

dev = torch.device("cuda:0")
ROWS, E, F, NTOK, B = 33280, 512, 768, 65, 512
bf = torch.bfloat16


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    flt = [a for a in sys.argv[1:] if not a.startswith("--")]
    iters = 20
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: t.data_ptr()
    cases = []

    def tail(n, k, pd):
        h = torch.randn(ROWS, n, device=dev).to(bf)
        x = torch.randn(ROWS, k, device=dev).to(bf)
        g, be = torch.ones(n, device=dev), torch.zeros(n, device=dev)
        out, dh, dx = torch.empty_like(h), torch.empty_like(h), torch.empty_like(x)
        mean, rstd = torch.empty(ROWS, device=dev), torch.empty(ROWS, device=dev)
        dg, db, dbi = (torch.empty(n, device=dev) for _ in range(3))
        part = torch.empty(_native.call("spv_rowop_partial_floats", n), device=dev)
        fwd = lambda: _native.call("spv_spectre_tail_fwd", p(h), p(x), p(g), p(be), p(out), p(mean), p(rstd), ROWS, n, k, 1, 1, pd, 7, st)
        bwd = lambda: _native.call("spv_spectre_tail_bwd", p(out), p(h), p(mean), p(rstd), p(g), p(be), p(dh), p(dx), p(dg), p(db),
                                   p(dbi), p(part), ROWS, n, k, 1, 1, pd, 7, 0, st)
        byt_f = ROWS * (2 * n + k) * 2
        byt_b = ROWS * (3 * n + k) * 2
        cases.append((f"tail_fwd n={n} k={k} p={pd}", fwd, byt_f, None))
        cases.append((f"tail_bwd n={n} k={k} p={pd}", bwd, byt_b, None))

    tail(768, 512, 0.001)
    tail(512, 768, 0.001)
    tail(768, 512, 0.0)
    tail(768, 768, 0.0)
    tail(512, 512, 0.0)
    tail(768, 1536, 0.0)
    tail(512, 8192, 0.0) if "mix" in "".join(flt) else None

    a = torch.randn(ROWS, E, device=dev).to(bf)
    b2 = torch.randn(ROWS, E, device=dev).to(bf)
    o = torch.empty_like(a)
    g, be = torch.ones(E, device=dev), torch.zeros(E, device=dev)
    mean, rstd = torch.empty(ROWS, device=dev), torch.empty(ROWS, device=dev)
    dg, db = torch.empty(E, device=dev), torch.empty(E, device=dev)
    part = torch.empty(_native.call("spv_rowop_partial_floats", E), device=dev)
    for mode in (0, 1):
        cases.append((f"addln_fwd mode{mode}", lambda mode=mode: _native.call("spv_add_layernorm_fwd", p(a), p(b2), p(g), p(be), p(o), p(mean), p(rstd), ROWS, E, mode, 1, st), ROWS * E * 2 * 3, None))
        cases.append((f"addln_bwd mode{mode}", lambda mode=mode: _native.call("spv_add_layernorm_bwd", p(o), p(a), p(b2), p(mean), p(rstd), p(g), p(o), p(dg), p(db), p(part), ROWS, E, mode, 1, st), ROWS * E * 2 * (3 + mode), None))

    x3 = torch.randn(B, NTOK, E, device=dev).to(bf)
    cases.append(("fnet_mix bf16", lambda: H._fnet_raw(x3), 2 * B * NTOK * E * 2, None))
    x3f = torch.randn(B, NTOK, E, device=dev)
    cases.append(("fnet_mix f32", lambda: H._fnet_raw(x3f), 2 * B * NTOK * E * 4, None))

    def gemm(M, N, K, out_f32=False, splits=1, acc=0):
        A = torch.randn(M, K, device=dev).to(bf)
        Bm = torch.randn(N, K, device=dev).to(bf)
        C = torch.empty(M, N, device=dev, dtype=torch.float32 if out_f32 else bf)
        ws = torch.empty(splits * M * N, device=dev) if splits > 1 else None
        cases.append((f"gemm {M}x{N}x{K} splits={splits} acc={acc}", lambda: H._gemm_launch(A, Bm, None, C, M, N, K, K, K, N, acc, splits, ws), None, 2.0 * M * N * K))

    gemm(ROWS, F, E)
    gemm(ROWS, E, F)
    gemm(ROWS, E, F, acc=1)
    gemm(F, E, ROWS, True, 12)
    gemm(E, F, ROWS, True, 12)
    gemm(4096, 4096, 4096)
    if "mix" in "".join(flt):
        gemm(ROWS, E, 8192)
        gemm(ROWS, 8192, E, acc=1)
        gemm(E, 8192, ROWS, True, 4)

    def gemm_tn(M, N, K, splits):
        A = torch.randn(K, M, device=dev).to(bf)
        Bm = torch.randn(K, N, device=dev).to(bf)
        C = torch.empty(M, N, device=dev)
        ws = torch.empty(splits * M * N, device=dev)
        cases.append((f"gemm_tn {M}x{N}x{K} splits={splits}", lambda: _native.call("spv_gemm_tn", p(A), p(Bm), p(C), M, N, K, M, N, N, 0, 0, splits, p(ws), st), None, 2.0 * M * N * K))

    gemm_tn(F, E, ROWS, 12)
    gemm_tn(E, F, ROWS, 12)
    gemm_tn(F, E, ROWS, 42)
    gemm_tn(F, E, ROWS, 21)
    gemm_tn(F, E, ROWS, 28)
    gemm_tn(F, E, ROWS, 64)
    t1 = torch.empty(F, ROWS, device=dev, dtype=bf)
    hh = torch.randn(ROWS, F, device=dev).to(bf)
    cases.append(("cast_transpose 33280x768 bf16", lambda: _native.call("spv_cast_transpose", p(hh), 1, p(t1), 1, ROWS, F, ROWS, 0, 0, 0, st), ROWS * F * 4, None))

    if "gather" in "".join(flt):
        heads, d = 16, NTOK * E
        perms = torch.stack([torch.randperm(d) for _ in range(heads)]).to(dev)
        signs = (torch.randint(0, 2, (heads, d)) * 2 - 1).float().to(dev)
        idx = H.permut_pack(perms, signs)
        xg = torch.randn(B, d, device=dev).to(bf)
        gg = torch.empty(B, heads * d, device=dev, dtype=bf)
        pooled = torch.empty(B * NTOK, E, device=dev, dtype=bf)
        dxg = torch.empty_like(xg)
        cases.append(("gather_fwd (+pooled skip)", lambda: _native.call("spv_permut_gather_fwd", p(xg), p(idx), p(gg), p(pooled), 16, B, heads, d, 1, st), B * d * (1 + heads) * 2, None))
        cases.append(("gather_bwd", lambda: _native.call("spv_permut_gather_bwd", p(gg), p(idx), p(dxg), B, heads, d, 1, st), B * d * (1 + heads) * 2, None))

    for name, fn, byt, flops in cases:
        if flt and not any(f in name for f in flt if f != "mix"):
            continue
        us = timeit(fn, iters)
        if byt:
            print(f"{name:44s} {us:9.2f} us  {byt / us * 1e-3:9.1f} GB/s ({byt / us * 1e-3 / 8000 * 100:5.1f}% of 8 TB/s)")
        else:
            print(f"{name:44s} {us:9.2f} us  {flops / us * 1e-6:9.1f} TFLOP/s ({flops / us * 1e-6 / 2500 * 100:5.1f}% of 2.5 PF)")


if __name__ == "__main__":
    main()
