"""GPU parity tests: every HIP kernel (through the C-ABI, via spectre_vit.hip_ops) against the numpy oracle
on the same seeded inputs.  Tolerances: fp32 kernels vs float64 oracle  max|err| <= 3e-5 * max|ref| (fp32
accumulation over K <= 8192); bf16 kernels (bf16 storage, fp32 accumulate) <= 3e-2 * max|ref|."""
import numpy as np
import pytest
import torch

from oracle import spectre_oracle as O

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 3e-5, torch.bfloat16: 3e-2}


def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def t(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev()).to(dtype)


def n64(x):
    return x.detach().float().cpu().numpy().astype(np.float64)


def relerr(got, ref):
    ref = np.asarray(ref, np.float64)
    return float(np.abs(n64(got) - ref).max() / (np.abs(ref).max() + 1e-30))


def check(got, ref, tol, what=""):
    e = relerr(got, ref)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"


def q(a, dtype):
    """round inputs to the kernel's storage dtype so the oracle sees the same values"""
    return n64(t(a, dtype))


@pytest.fixture(scope="module")
def ops():
    from spectre_vit import hip_ops
    return hip_ops


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,splits", [(300, 200, 64, 1), (128, 128, 128, 1), (1000, 100, 512, 1), (257, 384, 1032, 1),
                                          (96, 48, 4096, 7), (512, 768, 33280, 12), (5, 12, 8, 1)])
def test_gemm_nt(ops, dtype, M, N, K, splits):
    rng = np.random.default_rng(M + N + K)
    a = q(rng.standard_normal((M, K)), dtype)
    b = q(rng.standard_normal((N, K)), dtype)
    bias = rng.standard_normal(N)
    A, B, bi = t(a, dtype), t(b, dtype), t(bias)
    C = torch.empty((M, N), dtype=torch.float32, device=dev())
    ws = torch.empty((splits * M * N,), dtype=torch.float32, device=dev()) if splits > 1 else None
    ops._gemm(A, B, bi, C, M, N, K, K, K, N, 0, splits, ws)
    ref = a @ b.T + bias
    check(C, ref, 2e-5 if dtype == torch.float32 else 1e-4, "gemm")  # inputs already rounded: only accumulation differs
    # accumulate into an existing C, output in the input dtype
    C2 = t(rng.standard_normal((M, N)), dtype)
    c2 = n64(C2)
    ops._gemm(A, B, None, C2, M, N, K, K, K, N, 1, 1, None)
    check(C2, a @ b.T + c2, TOL[dtype], "gemm accumulate")


@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K", [(512, 768, 512), (512, 512, 768), (500, 776, 528), (2048, 264, 16), (33, 2048, 1040), (512, 512, 48), (96, 1024, 1536)])
def test_gemm_nt_rows_kernel(ops, M, N, K, out_dtype):
    """The few-rows kernel (bf16 operands, M <= 2048, one 32 x 32 tile per workgroup, K dealt over four waves; the CLS-only last layer's
    512-row GEMMs) against fp64: ragged M and N, a K that is not a multiple of the per-wave batch, bias, accumulate, both output types;
    the census proves the path; nothing is written outside the output."""
    from spectre_vit import _native
    rng = np.random.default_rng(M + N + K)
    a, b = q(rng.standard_normal((M, K)), torch.bfloat16), q(rng.standard_normal((N, K)) * 0.1, torch.bfloat16)
    bias = rng.standard_normal(N)
    A, B, bi = t(a, torch.bfloat16), t(b, torch.bfloat16), t(bias)
    C = torch.full((M + 1, N), 7.0, dtype=out_dtype, device=dev())
    before = _native.call("spv_path_count", _native.PATH["gemm_rows"])
    ops._gemm(A, B, bi, C, M, N, K, K, K, N)
    assert _native.call("spv_path_count", _native.PATH["gemm_rows"]) == before + 1
    tol = 1e-4 if out_dtype == torch.float32 else TOL[torch.bfloat16]
    check(C[:M], a @ b.T + bias, tol, "rows gemm")
    assert bool((C[M] == 7.0).all()), "wrote past the last row"
    C0 = t(rng.standard_normal((M, N)), out_dtype)
    c0 = n64(C0)
    ops._gemm(A, B, None, C0, M, N, K, K, K, N, 1)
    check(C0, a @ b.T + c0, tol, "rows gemm accumulate")


@pytest.mark.parametrize("M,N,K", [(33280, 768, 512), (33280, 512, 768), (33270, 256, 128), (20000, 512, 384), (33280, 1024, 256),
                                   (94203, 512, 128), (33280, 1536, 512), (16640, 2048, 128)])
def test_gemm_nt_strip_kernel(ops, M, N, K):
    """The one-workgroup-per-CU strip kernel (bf16 out, N % 256 == 0, K % 128 == 0, M >= 8192) against the 128 x 128 kernel of
    the same library (fp32 out never takes the strip path): same MFMA order along K, so the bf16 results must be the
    rounding of the fp32 ones bit for bit -- uneven row shares, the extra block, the masked last rows, bias, accumulate."""
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = (torch.randn((M, K), generator=g) * 0.5).to(dev()).to(torch.bfloat16)
    B = (torch.randn((N, K), generator=g) * 0.1).to(dev()).to(torch.bfloat16)
    bias = torch.randn((N,), generator=g).to(dev())
    C32 = torch.empty((M, N), dtype=torch.float32, device=dev())
    ops._gemm(A, B, bias, C32, M, N, K, K, K, N, 0, 1, None)
    # independent reference on a sample of rows (fp64)
    rows = torch.tensor([0, 1, 31, 32, 255, 256, 287, 288, 1000, M // 2, M - 33, M - 2, M - 1])
    ref = A[rows].double().cpu().numpy() @ B.double().cpu().numpy().T + bias.double().cpu().numpy()
    check(C32[rows], ref, 1e-4, "fp32-out reference kernel")
    C = torch.full((M + 1, N), 7.0, dtype=torch.bfloat16, device=dev())  # one guard row behind the output
    ops._gemm(A, B, bias, C, M, N, K, K, K, N, 0, 1, None)
    assert torch.equal(C[:M], C32.to(torch.bfloat16)), "strip kernel differs from the rounded fp32 result"
    assert bool((C[M] == 7.0).all()), "strip kernel wrote past the last row"
    # accumulate into an existing bf16 C (the data-gradient GEMMs of the layer): (acc + 0) + old, one rounding
    C0 = torch.randn((M, N), generator=g).to(dev()).to(torch.bfloat16)
    C2 = C0.clone()
    ops._gemm(A, B, None, C2, M, N, K, K, K, N, 1, 1, None)
    C32b = torch.empty((M, N), dtype=torch.float32, device=dev())
    ops._gemm(A, B, None, C32b, M, N, K, K, K, N, 0, 1, None)
    assert torch.equal(C2, (C32b + C0.float()).to(torch.bfloat16)), "strip kernel accumulate differs"


def test_gemm_nt_strip_kernel_random_shapes(ops):
    """Row-share edge cases of the strip kernel (no remainder, all-extra groups, short last sub-tiles, rows not a multiple of 32):
    a seeded sweep of shapes, bf16 output against the rounding of the 128 x 128 kernel's fp32 output, bit for bit."""
    rs = np.random.RandomState(20260)
    g = torch.Generator(device="cpu").manual_seed(77)
    cases = [(8192 * 2, 256, 128), (23 * 32 * 128, 512, 128), (256 * 9 * 32, 256, 256), (85 * 12 * 32, 768, 128)]
    for _ in range(10):
        cases.append((int(rs.randint(8192, 40000)), int(rs.choice([256, 512, 768, 1024, 1536, 2048])), int(rs.choice([128, 256, 384, 640]))))
    for M, N, K in cases:
        A = (torch.randn((M, K), generator=g) * 0.5).to(dev()).to(torch.bfloat16)
        B = (torch.randn((N, K), generator=g) * 0.1).to(dev()).to(torch.bfloat16)
        bias = torch.randn((N,), generator=g).to(dev())
        C32 = torch.empty((M, N), dtype=torch.float32, device=dev())
        ops._gemm(A, B, bias, C32, M, N, K, K, K, N, 0, 1, None)
        C = torch.full((M + 1, N), 7.0, dtype=torch.bfloat16, device=dev())
        ops._gemm(A, B, bias, C, M, N, K, K, K, N, 0, 1, None)
        assert torch.equal(C[:M], C32.to(torch.bfloat16)), f"M={M} N={N} K={K}"
        assert bool((C[M] == 7.0).all()), f"M={M} N={N} K={K}: wrote past the last row"
        C0 = torch.randn((M, N), generator=g).to(dev()).to(torch.bfloat16)
        C2 = C0.clone()
        ops._gemm(A, B, None, C2, M, N, K, K, K, N, 1, 1, None)
        C32b = torch.empty((M, N), dtype=torch.float32, device=dev())
        ops._gemm(A, B, None, C32b, M, N, K, K, K, N, 0, 1, None)
        assert torch.equal(C2, (C32b + C0.float()).to(torch.bfloat16)), f"M={M} N={N} K={K}: accumulate"
        del A, B, C32, C, C0, C2, C32b


def test_gemm_nt_strip_kernel_with_reserved_cus(ops):
    """spv_set_reserved_cus (multi-GPU runs leave CUs to RCCL) changes the row shares, not the numbers."""
    from spectre_vit import _native
    M, N, K = 33280, 768, 512
    g = torch.Generator(device="cpu").manual_seed(5)
    A = (torch.randn((M, K), generator=g) * 0.5).to(dev()).to(torch.bfloat16)
    B = (torch.randn((N, K), generator=g) * 0.1).to(dev()).to(torch.bfloat16)
    C0 = torch.empty((M, N), dtype=torch.bfloat16, device=dev())
    ops._gemm(A, B, None, C0, M, N, K, K, K, N, 0, 1, None)
    try:
        for reserve in (16, 40, 128):
            _native.call("spv_set_reserved_cus", reserve)
            C1 = torch.full((M, N), 3.0, dtype=torch.bfloat16, device=dev())
            ops._gemm(A, B, None, C1, M, N, K, K, K, N, 0, 1, None)
            assert torch.equal(C0, C1), f"reserve {reserve}"
    finally:
        _native.call("spv_set_reserved_cus", 0)
    with pytest.raises(RuntimeError):
        _native.call("spv_set_reserved_cus", 200)


@pytest.mark.parametrize("M,N,K,splits", [(128, 128, 64, 1), (768, 512, 33280, 12), (104, 48, 1000, 3), (512, 8192, 2600, 2),
                                          (8, 16, 40, 1), (264, 136, 4100, 5),
                                          (768, 512, 8010, 12),    # the 256 x 128 tile with a partial last K-tile in the last slice
                                          (512, 48, 3000, 16),     # the sliver shape of the patch-embedding gradient (one K-tile in flight)
                                          (256, 128, 100, 1),      # fewer rows than one pipelined group of K-tiles
                                          (512, 8192, 2600, 3),    # the 512 x 128 tile (all of M in one workgroup), partial last K-tile
                                          (512, 8192, 4000, 4)])
def test_gemm_tn(ops, M, N, K, splits):
    """C = A^T B from row-major [K,M], [K,N] bf16 (ds_read_b64_tr_b16 fragments); asymmetric random operands."""
    from spectre_vit import _native
    rng = np.random.default_rng(M * 3 + N * 5 + K)
    a = q(rng.standard_normal((K, M)), torch.bfloat16)
    b = q(rng.standard_normal((K, N)) + 0.25, torch.bfloat16)
    A, B = t(a, torch.bfloat16), t(b, torch.bfloat16)
    C = torch.empty((M, N), dtype=torch.float32, device=dev())
    ws = torch.empty((splits * M * N,), dtype=torch.float32, device=dev()) if splits > 1 else None
    _native.call("spv_gemm_tn", A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, M, N, N, 0, 0, splits,
                 0 if ws is None else ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
    check(C, a.T @ b, 1e-4, "gemm_tn")
    # identity check with asymmetric B: A = [I; 0] picks rows of B
    Kp = 64
    eye = np.zeros((Kp, 32)); eye[:32] = np.eye(32)
    bb = q(rng.standard_normal((Kp, 40)), torch.bfloat16)
    A2, B2 = t(eye, torch.bfloat16), t(bb, torch.bfloat16)
    C2 = torch.empty((32, 40), dtype=torch.float32, device=dev())
    _native.call("spv_gemm_tn", A2.data_ptr(), B2.data_ptr(), C2.data_ptr(), 32, 40, Kp, 32, 40, 40, 0, 0, 1, 0,
                 torch.cuda.current_stream().cuda_stream)
    assert np.array_equal(n64(C2), bb[:32]), "A = I must return B's rows exactly"


@pytest.mark.parametrize("shapes,rows,short_rows,splits", [([(768, 512), (512, 768)] * 2, 8320, 320, 7),   # the 256 x 128 tile
                                                           ([(264, 136), (128, 128), (136, 264)], 2100, 130, 3)])   # the 128 x 128 tile, edges
def test_gemm_tn_batch_long_and_short_problems(ops, shapes, rows, short_rows, splits):
    """spv_gemm_tn_batch: several weight gradients in one launch -- the LONG ones over the call's K rows (split-K + reduce), the last
    two as SHORT problems over their first `k` rows only (spv_tn_problem.k; the CLS-only last layer's gradients ride like this),
    each against A^T B in float64."""
    import ctypes
    from spectre_vit import _native
    rng = np.random.default_rng(rows + short_rows)
    nshort = 2
    A = [t(q(rng.standard_normal((rows, m)), torch.bfloat16), torch.bfloat16) for m, n in shapes]
    B = [t(q(rng.standard_normal((rows, n)) + 0.25, torch.bfloat16), torch.bfloat16) for m, n in shapes]
    C = [torch.full((m, n), 7.0, dtype=torch.float32, device=dev()) for m, n in shapes]
    probs = (_native.TnProblem * len(shapes))()
    for i, (pq, a, b, c, (m, n)) in enumerate(zip(probs, A, B, C, shapes)):
        pq.a, pq.b, pq.c, pq.m, pq.n, pq.lda, pq.ldb, pq.ldc = a.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, m, n, n
        pq.k = short_rows if i >= len(shapes) - nshort else 0
    floats = sum(m * n for m, n in shapes[:len(shapes) - nshort])
    ws = torch.empty((splits * floats,), dtype=torch.float32, device=dev())
    _native.call("spv_gemm_tn_batch", ctypes.addressof(probs), len(shapes), rows, splits, ws.data_ptr(), 0, 0,
                 torch.cuda.current_stream().cuda_stream)
    for i, (a, b, c) in enumerate(zip(A, B, C)):
        r = short_rows if i >= len(shapes) - nshort else rows
        check(c, n64(a)[:r].T @ n64(b)[:r], 1e-4, f"tn batch problem {i} ({r} rows)")
    with pytest.raises(RuntimeError):   # a problem may not reduce over MORE rows than the call
        probs[0].k = rows + 64
        _native.call("spv_gemm_tn_batch", ctypes.addressof(probs), len(shapes), rows, splits, ws.data_ptr(), 0, 0,
                     torch.cuda.current_stream().cuda_stream)


def test_gemm_grouped_rows(ops):
    from spectre_vit import _native
    rng = np.random.default_rng(3)
    Bn, Np, K, E = 5, 16, 48, 64
    a, w = rng.standard_normal((Bn * Np, K)), rng.standard_normal((E, K))
    pb = rng.standard_normal((Np, E))
    A, W, PB = t(a), t(w), t(pb)
    out = torch.zeros((Bn, Np + 1, E), device=dev())
    _native.call("spv_gemm_nt_grouped_rows", A.data_ptr(), W.data_ptr(), 0, PB.data_ptr(), out.data_ptr(), Bn * Np, E, K, K, K, E,
                 0, 0, Np, Np + 1, 1, torch.cuda.current_stream().cuda_stream)
    ref = np.zeros((Bn, Np + 1, E))
    ref[:, 1:, :] = (a @ w.T).reshape(Bn, Np, E) + pb
    check(out, ref, 2e-5, "grouped gemm")


# ------------------------------------------------------------------------------------------------ SpectreLinear
def sl_oracle(x, W, b, g, be, dy):
    p = dict(weight=W, bias=b, ln_weight=g, ln_bias=be)
    y, c = O.spectre_linear_fwd(x, p)
    dx, gr = O.spectre_linear_bwd(dy, p, c)
    return y, dx, gr


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,lead", [(64, 64, (3, 7)), (256, 16, (2, 9)), (96, 64, (5, 3)), (64, 96, (4, 5)),
                                           (512, 768, (2, 65)), (768, 512, (2, 65)), (512, 104, (33,)), (1024, 64, (130,)),
                                           (48, 256, (3, 50)), (64, 256, (9,)), (768, 3072, (5,)), (3072, 768, (5,)),
                                           (768, 3072, (3, 197)), (3072, 768, (3, 197))])   # Base widths: 4-waves-per-row / 48-inputs-per-lane tails
def test_spectre_linear(ops, dtype, cin, cout, lead):
    rng = np.random.default_rng(cin * 7 + cout)
    x = q(rng.standard_normal(lead + (cin,)), dtype)
    W = rng.standard_normal((cout, cin)) / np.sqrt(cin)
    b = rng.standard_normal(cout) * 0.1
    g = rng.random(cout) + 0.5
    be = rng.standard_normal(cout) * 0.1
    dy = q(rng.standard_normal(lead + (cout,)), dtype)
    Wq = q(W, dtype)  # the kernel multiplies with the dtype shadow of W
    y_ref, dx_ref, gr = sl_oracle(x, Wq, b, g, be, dy)
    X = t(x, dtype).requires_grad_(True)
    Wt, bt, gt, bet = (t(W).requires_grad_(True), t(b).requires_grad_(True), t(g).requires_grad_(True), t(be).requires_grad_(True))
    Y = ops.spectre_linear(X, Wt, bt, gt, bet, 0.0, False)
    Y.backward(t(dy, dtype))
    tol = TOL[dtype]
    check(Y, y_ref, tol, "y")
    check(X.grad, dx_ref, tol * 2, "dx")
    check(Wt.grad, gr["weight"], tol * 2, "dW")
    check(bt.grad, gr["bias"], tol * 2, "db")
    check(gt.grad, gr["ln_weight"], tol * 2, "dgamma")
    check(bet.grad, gr["ln_bias"], tol * 2, "dbeta")


def test_spectre_linear_golden(ops, golden_ops):
    """the reference's own SpectreLinear outputs (tests/golden/ops.npz), fp32 kernels"""
    g = golden_ops
    for name in ["sl_equal", "sl_down_exact", "sl_down_overlap", "sl_up"]:
        X = t(g[f"{name}.x"]).requires_grad_(True)
        P = [t(g[f"{name}.sd.local_head.{k}"]).requires_grad_(True) for k in ("0.weight", "0.bias", "1.weight", "1.bias")]
        Y = ops.spectre_linear(X, *P, 0.0, False)
        Y.backward(t(g[f"{name}.dy"]))
        check(Y, g[f"{name}.y"], 3e-5, name + ".y")
        check(X.grad, g[f"{name}.dx"], 6e-5, name + ".dx")
        for pt, k in zip(P, ("0.weight", "0.bias", "1.weight", "1.bias")):
            check(pt.grad, g[f"{name}.grad.local_head.{k}"], 6e-5, name + ".grad." + k)


@pytest.mark.parametrize("p_up", [0.0, 0.25])
def test_tail_bwd_takes_skip_gradient_of_layer_above(ops, p_up):
    """spv_spectre_tail_bwd_up (512 -> 768 layer): dout + pool^T(mask * up_src) formed inside the kernel must equal the old
    composition -- the 768 -> 512 layer's backward writes dx_pool, which is added to dout -- windows, lane mapping and the
    dropout mask of the layer above included.  bf16 storage: the composition rounds dx_pool and the sum once more."""
    from spectre_vit import _native
    rows, n, k = 300, 768, 512
    g = torch.Generator(device="cpu").manual_seed(11)
    bf = torch.bfloat16
    rnd = lambda *shape: torch.randn(shape, generator=g).to(dev())
    st = torch.cuda.current_stream().cuda_stream
    # the layer above (768 -> 512): only its dx_pool = pool^T(mask * dout_up) matters here
    dout_up, h_up = rnd(rows, k).to(bf), rnd(rows, k).to(bf)
    mean_up, rstd_up = rnd(rows) * 0.1, rnd(rows).abs() + 0.5
    ga_up, be_up = rnd(k), rnd(k)
    dh_up, dxp_up = torch.empty_like(h_up), torch.empty((rows, n), dtype=bf, device=dev())
    d3 = [torch.empty((k,), device=dev()) for _ in range(3)]
    part = torch.empty((_native.call("spv_rowop_partial_floats", k),), device=dev())
    seed_up = 1234567
    _native.call("spv_spectre_tail_bwd", dout_up.data_ptr(), h_up.data_ptr(), mean_up.data_ptr(), rstd_up.data_ptr(), ga_up.data_ptr(),
                 be_up.data_ptr(), dh_up.data_ptr(), dxp_up.data_ptr(), d3[0].data_ptr(), d3[1].data_ptr(), d3[2].data_ptr(),
                 part.data_ptr(), rows, k, n, 1, 1, p_up, seed_up, 0, st)
    # this layer (512 -> 768)
    dout, h = rnd(rows, n).to(bf), rnd(rows, n).to(bf)
    mean, rstd = rnd(rows) * 0.1, rnd(rows).abs() + 0.5
    ga, be = rnd(n), rnd(n)
    dx_add = rnd(rows, k).to(bf)

    def run(up):
        dh, dx = torch.empty_like(h), torch.empty((rows, k), dtype=bf, device=dev())
        dg, db, dbi = (torch.empty((n,), device=dev()) for _ in range(3))
        pt = torch.empty((_native.call("spv_rowop_partial_floats", n),), device=dev())
        if up:
            _native.call("spv_spectre_tail_bwd_up", dout.data_ptr(), h.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ga.data_ptr(),
                         be.data_ptr(), dh.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), dbi.data_ptr(), pt.data_ptr(), rows,
                         n, k, 1, 1, 0.0, 0, dx_add.data_ptr(), dout_up.data_ptr(), p_up, seed_up, st)
        else:
            dsum = (dout.float() + dxp_up.float()).to(bf)
            _native.call("spv_spectre_tail_bwd", dsum.data_ptr(), h.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ga.data_ptr(),
                         be.data_ptr(), dh.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), dbi.data_ptr(), pt.data_ptr(), rows,
                         n, k, 1, 1, 0.0, 0, dx_add.data_ptr(), st)
        torch.cuda.synchronize()
        return [n64(v) for v in (dh, dx, dg, db, dbi)]

    ref, got = run(False), run(True)
    for name, r, v in zip(("dh", "dx", "dgamma", "dbeta", "dbias"), ref, got):
        assert np.abs(v - r).max() <= 3e-2 * np.abs(r).max(), f"{name}: {np.abs(v - r).max()} vs max {np.abs(r).max()}"
    assert np.abs(n64(dxp_up)).max() > 0.1  # the skip term is not trivially zero


def test_spectre_linear_dropout(ops):
    torch.manual_seed(0)
    X = torch.randn(4096, 64, device=dev()).requires_grad_(True)
    W = (torch.randn(64, 64, device=dev()) / 8).requires_grad_(True)
    z = torch.zeros(64, device=dev(), requires_grad=True)
    o = torch.ones(64, device=dev(), requires_grad=True)
    y0 = ops.spectre_linear(X, W, z, o, z, 0.0, False)
    y1 = ops.spectre_linear(X, W, z, o, z, 0.25, False)
    kept = (y1 != 0)
    frac = kept.float().mean().item()
    assert abs(frac - 0.75) < 0.01, frac
    torch.testing.assert_close(y1[kept], (y0 / 0.75)[kept], rtol=1e-5, atol=1e-6)
    y1.sum().backward()  # backward regenerates the same mask: dropped outputs contribute no gradient
    gx = X.grad.clone()
    assert torch.isfinite(gx).all()


@pytest.mark.parametrize("n,k", [(768, 512), (512, 768)])
def test_spectre_linear_small_p_dropout_statistics(ops, n, k):
    """Small p on the 512 <-> 768 layer shapes (lane-contiguous row kernels): the counter-hash mask must be i.i.d.
    Bernoulli(p) -- rate, per-row count variance (a scheme with 'at most one drop per group' would show a deficit), column
    coverage, kept values -- and the backward must regenerate the same mask."""
    torch.manual_seed(3)
    rows, p = 4096, 0.004
    X = torch.randn(rows, k, device=dev()).requires_grad_(True)
    W = (torch.randn(n, k, device=dev()) / 16).requires_grad_(True)
    b = torch.zeros(n, device=dev(), requires_grad=True)
    g = torch.ones(n, device=dev(), requires_grad=True)
    be = torch.full((n,), 0.3, device=dev(), requires_grad=True)
    y0 = ops.spectre_linear(X, W, b, g, be, 0.0, False).detach()
    y1 = ops.spectre_linear(X, W, b, g, be, p, False)
    live = y0 != 0
    dropped = (y1 == 0) & live
    frac = dropped.float().sum().item() / live.float().sum().item()
    assert abs(frac - p) < 3e-4, frac
    cnt = dropped.float().sum(1)
    assert abs(cnt.mean().item() - n * p) < 0.15 and abs(cnt.var().item() / (n * p * (1 - p)) - 1.0) < 0.12, (cnt.mean().item(), cnt.var().item())
    col = dropped.float().sum(0)
    assert col.min().item() >= 1 and col.max().item() <= 50, (col.min().item(), col.max().item())
    kept = live & ~dropped
    torch.testing.assert_close(y1.detach()[kept], (y0 / (1 - p))[kept], rtol=1e-5, atol=1e-6)
    y1.sum().backward()
    gx1 = X.grad.clone()
    X.grad = None
    y2 = ops.spectre_linear(X, W, b, g, be, 0.0, False)
    y2.backward((~dropped).float() / (1 - p))  # the same mask applied through the output gradient
    torch.testing.assert_close(gx1, X.grad, rtol=2e-4, atol=2e-5)


# ------------------------------------------------------------------------------------------------ add + LayerNorm
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("rows,n", [(130, 512), (7, 48), (33, 768), (5, 100), (9, 3072), (6, 1000)])
def test_add_layernorm(ops, dtype, mode, rows, n):
    rng = np.random.default_rng(rows + n + mode)
    a, b = q(rng.standard_normal((rows, n)), dtype), q(rng.standard_normal((rows, n)), dtype)
    g, be = rng.random(n) + 0.5, rng.standard_normal(n) * 0.1
    dy = q(rng.standard_normal((rows, n)), dtype)
    if mode == 0:
        ln, c = O.layernorm_fwd(a, g, be)
        ref = ln + b
    else:
        ref, c = O.layernorm_fwd(a + b, g, be)
    din, dg, db = O.layernorm_bwd(dy, g, c)
    A, Bt = t(a, dtype).requires_grad_(True), t(b, dtype).requires_grad_(True)
    G, Be = t(g).requires_grad_(True), t(be).requires_grad_(True)
    Y = ops.add_layernorm(A, Bt, G, Be, mode)
    Y.backward(t(dy, dtype))
    tol = TOL[dtype]
    check(Y, ref, tol, "y")
    check(A.grad, din, tol * 2, "da")
    check(Bt.grad, dy if mode == 0 else din, tol * 2, "db")
    check(G.grad, dg, tol * 2, "dgamma")
    check(Be.grad, db, tol * 2, "dbeta")


# ------------------------------------------------------------------------------------------------ permutation gather
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,N,E,H", [(3, 5, 8, 3), (2, 65, 64, 4), (2, 65, 512, 16), (2, 100, 512, 2), (2, 7, 12, 5),
                                     (2, 197, 768, 2), (3, 160, 512, 3)])   # rows longer than the LDS (Base / 224): staged in parts (bf16)
def test_permut_gather(ops, dtype, B, N, E, H):
    rng = np.random.default_rng(B + N + E + H)
    d = N * E
    perms = np.stack([rng.permutation(d) for _ in range(H)]).astype(np.int64)
    signs = rng.integers(0, 2, (1, H, d)).astype(np.float64) * 2 - 1
    x = q(rng.standard_normal((B, N, E)), dtype)
    dg = q(rng.standard_normal((B, N, E * H)), dtype)
    idx = ops.permut_pack(torch.from_numpy(perms).to(dev()), t(signs).reshape(H, d))
    X = t(x, dtype).requires_grad_(True)
    G = ops.PermutGatherFn.apply(X, idx, H)
    ref = O.permut_gather_fwd(x, perms, signs)
    assert np.array_equal(n64(G).reshape(ref.shape), ref), "gather must be bit exact (pure data movement + sign flip)"
    G.backward(t(dg, dtype).reshape(G.shape))
    check(X.grad, O.permut_gather_bwd(dg, perms, signs, N, E), 1e-6 if dtype == torch.float32 else 8e-3, "dx")


# ------------------------------------------------------------------------------------------------ spectral mixers
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,N,D", [(3, 65, 512), (2, 5, 16), (2, 50, 48), (2, 17, 64), (4, 65, 128), (2, 6, 8), (2, 4, 1024),
                                   (2, 79, 256), (1, 197, 96)])
def test_fnet_mix(ops, dtype, B, N, D):
    rng = np.random.default_rng(N * 1000 + D)
    x = q(rng.standard_normal((B, N, D)), dtype)
    dy = q(rng.standard_normal((B, N, D)), dtype)
    X = t(x, dtype).requires_grad_(True)
    Y = ops.FNetMixFn.apply(X)
    Y.backward(t(dy, dtype))
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    check(Y, O.fnet_mix_fwd(x), tol, "y")
    check(X.grad, O.fnet_mix_bwd(dy), tol, "dx")


def test_fnet_golden(ops, golden_ops):
    check(ops.FNetMixFn.apply(t(golden_ops["fnet65.x"])), golden_ops["fnet65.y"], 2e-5, "fnet65 vs torch.fft.fft2.real")
    check(ops.FNetMixFn.apply(t(golden_ops["fnet.x"])), golden_ops["fnet.y"], 2e-5, "fnet vs torch.fft.fft2.real")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rfft_real(ops, golden_ops, dtype):
    g = golden_ops
    X = t(q(g["fftmod.x"], dtype), dtype).requires_grad_(True)
    Y = ops.RfftRealFn.apply(X)
    Y.backward(t(g["fftmod.dy"], dtype))
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    check(Y, O.fft_module_fwd(q(g["fftmod.x"], dtype)), tol, "y")
    check(X.grad, O.fft_module_bwd(q(g["fftmod.dy"], dtype), 16), tol, "dx")
    if dtype == torch.float32:
        check(Y, g["fftmod.y"], tol, "golden y")
        check(X.grad, g["fftmod.dx"], tol, "golden dx")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,axis,levels", [((2, 65, 32), 1, 1), ((2, 65, 32), 1, 3), ((3, 5, 64), 2, 1), ((3, 5, 64), 2, 6),
                                               ((2, 65, 512), 2, 2), ((2, 65, 512), 1, 2)])
def test_haar_dwt(ops, dtype, shape, axis, levels):
    rng = np.random.default_rng(levels + axis)
    x, dy = q(rng.standard_normal(shape), dtype), q(rng.standard_normal(shape), dtype)
    X = t(x, dtype).requires_grad_(True)
    Y = ops.HaarDWTFn.apply(X, axis, levels)
    Y.backward(t(dy, dtype))
    tol = 1e-6 if dtype == torch.float32 else 1.5e-2
    check(Y, O.haar_dwt_fwd(x, axis - 3, levels), tol, "y")
    check(X.grad, O.haar_dwt_bwd(dy, axis - 3, levels), tol, "dx")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,axis,levels", [((2, 65, 32), 1, 1), ((2, 65, 32), 1, 3), ((3, 5, 33), 2, 2), ((2, 65, 512), 1, 2)])
def test_haar_dwt_zero_mode(ops, dtype, shape, axis, levels):
    """mode="zero" (pywt's extension of odd lengths, the reference's dwt_experiments.py:56): forward and adjoint vs the oracle"""
    rng = np.random.default_rng(levels + axis)
    x, dy = q(rng.standard_normal(shape), dtype), q(rng.standard_normal(shape), dtype)
    X = t(x, dtype).requires_grad_(True)
    Y = ops.HaarDWTFn.apply(X, axis, levels, True)
    Y.backward(t(dy, dtype))
    tol = 1e-6 if dtype == torch.float32 else 1.5e-2
    check(Y, O.haar_dwt_fwd(x, axis - 3, levels, "zero"), tol, "y")
    check(X.grad, O.haar_dwt_bwd(dy, axis - 3, levels, "zero"), tol, "dx")


def test_haar_kernel_matches_pywavelets_documented_vectors(ops):
    """the HIP kernel on the vectors PyWavelets' documentation prints for 'db1' (tests/test_oracle_golden.py cites them): pair
    convention, sign of the detail band, wavedec's coefficient order"""
    from test_oracle_golden import PYWT_DWT_DB1, PYWT_WAVEDEC_DB1_L2
    x, cA, cD = PYWT_DWT_DB1
    for axis, shape in ((2, (1, 1, 4)), (1, (1, 4, 1))):
        y = ops.HaarDWTFn.apply(torch.tensor(x, dtype=torch.float32, device=dev()).reshape(shape), axis, 1)
        assert np.abs(y.cpu().numpy().reshape(-1) - np.concatenate([cA, cD])).max() < 1e-6
    x, cA2, cD2, cD1 = PYWT_WAVEDEC_DB1_L2
    for zero in (False, True):
        for axis, shape in ((2, (1, 1, 8)), (1, (1, 8, 1))):
            y = ops.HaarDWTFn.apply(torch.tensor(x, dtype=torch.float32, device=dev()).reshape(shape), axis, 2, zero)
            assert np.abs(y.cpu().numpy().reshape(-1) - np.concatenate([cA2, cD2, cD1])).max() < 1e-6
    # the mixer module itself, token axis, 65 tokens, mode "zero": 33 approximation + 32 detail coefficients of pywt's 33 + 33
    from spectre_vit.modules.mixers import HaarDWTMixer
    rng = np.random.default_rng(0)
    xs = rng.standard_normal((2, 65, 16))
    y = HaarDWTMixer("token", 1, "zero")(torch.tensor(xs, dtype=torch.float32, device=dev())).cpu().numpy()
    cA, cD = O.haar_level_pywt_zero(np.moveaxis(xs, 1, -1))
    ref = np.moveaxis(np.concatenate([cA, cD[..., :-1]], axis=-1), -1, 1)
    assert np.abs(y - ref).max() < 1e-6


@pytest.mark.parametrize("B,N", [(3, 65), (2, 17), (5, 64), (2, 2)])
def test_fnet_layernorm_residual_fused(ops, B, N):
    """x1 = LayerNorm1(Re(fft2(x))) + x as one kernel each way (bf16, D = 512): forward, dx, dgamma, dbeta against the oracle
    composition fnet_mix -> layernorm -> + x, and against the unfused kernels on the forward (the fused epilogue normalises
    the bf16-rounded mixer output, exactly what the separate kernel reads back: equal up to the summation order)."""
    D, dtype = 512, torch.bfloat16
    rng = np.random.default_rng(B * 100 + N)
    x = q(rng.standard_normal((B, N, D)) * 0.5, dtype)
    g, be = rng.random(D) + 0.5, rng.standard_normal(D) * 0.1
    dy = q(rng.standard_normal((B, N, D)), dtype)
    X = t(x, dtype).requires_grad_(True)
    G, Bt = t(g).requires_grad_(True), t(be).requires_grad_(True)
    assert ops._native.call("spv_fnet_ln_supported", N, D, 1) == 1
    Y = ops.FNetResidualFn.apply(X, G, Bt)
    Y.backward(t(dy, dtype))
    # oracle
    m = O.fnet_mix_fwd(x)
    mq = q(m, dtype)  # the pre-norm tensor is stored in bf16
    ln, cache = O.layernorm_fwd(mq, g, be)
    ref = ln + x
    dm, dg_ref, db_ref = O.layernorm_bwd(dy, g, cache)
    dx_ref = O.fnet_mix_bwd(q(dm, dtype)) + dy
    scale = np.abs(m).max()
    check(Y, ref, 3e-2, "x1")
    assert np.abs(n64(X.grad) - dx_ref).max() <= 3e-2 * np.abs(dx_ref).max(), "dx"
    check(G.grad, dg_ref, 3e-2, "dgamma")
    check(Bt.grad, db_ref, 3e-2, "dbeta")
    # unfused kernels on the same inputs: identical forward
    m_u = ops._fnet_raw(t(x, dtype))
    out_u, _ = ops._addln_forward(m_u.reshape(-1, D), t(x, dtype).reshape(-1, D), t(g), t(be), 0)
    a, b = out_u.reshape(B, N, D).float(), Y.detach().float()
    diff = (a - b).abs()
    # same arithmetic on the same bf16-rounded values; only the order of the two row sums differs -> at most a bf16 ulp, rarely
    assert float((diff > 0).float().mean()) < 1e-3 and float((diff / (b.abs() + 1e-3)).max()) <= 1.0 / 64, float(diff.max())
    assert scale > 0


# ------------------------------------------------------------------------------------------------ patch embeddings
def test_spectral_patch_embed_golden(ops, golden_ops):
    from spectre_vit.models.spectre.spectre import SpectralPatchEmbed
    g = golden_ops
    m = SpectralPatchEmbed(16, 4, 4, 0.0, 3).to(dev())
    m.load_state_dict({k[len("spe.sd."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("spe.sd.")})
    y = m(t(g["spe.x"]))
    y.backward(t(g["spe.dy"]))
    check(y, g["spe.y"], 3e-5, "tokens")
    for k, p in m.named_parameters():
        check(p.grad, g["spe.grad." + k], 1e-4, "grad " + k)


def test_conv_patch_embed_golden(ops, golden_ops):
    from spectre_vit.modules.patch_embeddings import PatchEmbedding
    g = golden_ops
    m = PatchEmbedding(16, 4, 4, 0.0, 3).to(dev())
    m.load_state_dict({k[len("pe.sd."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("pe.sd.")})
    y = m(t(g["pe.x"]))
    y.backward(t(g["pe.dy"]))
    check(y, g["pe.y"], 3e-5, "tokens")
    for k, p in m.named_parameters():
        check(p.grad, g["pe.grad." + k], 1e-4, "grad " + k)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_spectral_patch_embed_cifar_shape(ops, dtype):
    from spectre_vit.models.spectre.spectre import SpectralPatchEmbed
    torch.manual_seed(5)
    m = SpectralPatchEmbed(512, 4, 64, 0.0, 3).to(dev())
    with torch.no_grad():
        m.freq_weight_h.uniform_(0.5, 1.5)
        m.freq_weight_w.uniform_(0.5, 1.5)
    x = torch.randn(6, 3, 32, 32, device=dev())
    dy = torch.randn(6, 65, 512, device=dev())
    if dtype == torch.bfloat16:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(x)
    else:
        y = m(x)
    assert y.dtype == dtype and y.shape == (6, 65, 512)
    y.backward(dy.to(dtype))
    sd = {k: n64(v) for k, v in m.state_dict().items()}
    p = dict(freq_weight_h=sd["freq_weight_h"], freq_weight_w=sd["freq_weight_w"], proj_weight=sd["proj.weight"],
             proj_bias=sd["proj.bias"], cls_token=sd["cls_token"], position_embeddings=sd["position_embeddings"])
    ref, cache = O.spectral_patch_embed_fwd(n64(x), p, 4)
    gr = O.spectral_patch_embed_bwd(n64(dy.to(dtype)), p, 4, cache)
    tol = 3e-5 if dtype == torch.float32 else 2e-2
    check(y, ref, tol, "tokens")
    check(m.proj.weight.grad, gr["proj_weight"], tol * 3, "dproj")
    check(m.freq_weight_h.grad, gr["freq_weight_h"], tol * 3, "dfh")
    check(m.freq_weight_w.grad, gr["freq_weight_w"], tol * 3, "dfw")
    check(m.position_embeddings.grad, gr["position_embeddings"], tol * 3, "dpos")
    check(m.cls_token.grad, gr["cls_token"], tol * 3, "dcls")
    check(m.proj.bias.grad, gr["proj_bias"], tol * 3, "dbias")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind", ["spectral", "conv"])
def test_patch_embed_uint8_input(ops, dtype, kind):
    """uint8 NHWC batch, /255 + Normalize(mean, std) folded into the patch gather (train.py:102-112)."""
    from spectre_vit.models.spectre.spectre import SpectralPatchEmbed
    from spectre_vit.modules.patch_embeddings import PatchEmbedding
    torch.manual_seed(11)
    m = (SpectralPatchEmbed if kind == "spectral" else PatchEmbedding)(512, 4, 64, 0.0, 3).to(dev())
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(5, 32, 32, 3), dtype=np.uint8)
    x8 = torch.from_numpy(img).to(dev())
    dy = torch.randn(5, 65, 512, device=dev())
    ref_in = O.normalize_u8(img, ops.CIFAR100_MEAN, ops.CIFAR100_STD)

    def run(inp):
        m.zero_grad()
        if dtype == torch.bfloat16:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(inp)
        else:
            y = m(inp)
        y.backward(dy.to(y.dtype))
        return y.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}

    y8, g8 = run(x8)
    yf, gf = run(t(ref_in))  # the float NCHW contract on the oracle-normalised image
    assert y8.dtype == dtype and y8.shape == (5, 65, 512)
    tol = 2e-6 if dtype == torch.float32 else 1e-2  # fp32: one rounding apart (x * (1/255), * (1/std)); bf16: input rounding
    check(y8, n64(yf), tol, "tokens")
    for k in g8:
        check(g8[k], n64(gf[k]), tol * 4, "grad " + k)
    # and against the oracle end to end (fp32)
    if dtype == torch.float32 and kind == "spectral":
        sd = {k: n64(v) for k, v in m.state_dict().items()}
        p = dict(freq_weight_h=sd["freq_weight_h"], freq_weight_w=sd["freq_weight_w"], proj_weight=sd["proj.weight"],
                 proj_bias=sd["proj.bias"], cls_token=sd["cls_token"], position_embeddings=sd["position_embeddings"])
        ref, _ = O.spectral_patch_embed_fwd(ref_in, p, 4)
        check(y8, ref, 3e-5, "tokens vs oracle")


def test_patch_embed_uint8_needs_matching_channels(ops):
    from spectre_vit.models.spectre.spectre import SpectralPatchEmbed
    m = SpectralPatchEmbed(64, 4, 4, 0.0, 1).to(dev())
    with pytest.raises(ValueError):
        m(torch.zeros(2, 8, 8, 1, dtype=torch.uint8, device=dev()))  # default CIFAR statistics have 3 channels
    m.pixel_norm = ops.PixelNorm((0.5,), (0.25,))
    assert m(torch.zeros(2, 8, 8, 1, dtype=torch.uint8, device=dev())).shape == (2, 5, 64)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fft_approximator_and_binary_linear(ops, dtype):
    """SURVEY 8f-4 side branches (reference layers.py:10-23, 104-121): plain contractions, oracle = x @ W^T in float64."""
    from spectre_vit.models.spectre.layers import BinaryLinear, FFTApproximator
    torch.manual_seed(2)
    x = torch.randn(3, 65, 512, device=dev())
    fa = FFTApproximator(512).to(dev())
    bl = BinaryLinear(512, 96).to(dev())
    tol = TOL[dtype]
    for m in (fa, bl):
        xin = x.clone().requires_grad_(True)
        if dtype == torch.bfloat16:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(xin)
        else:
            y = m(xin)
        dy = torch.randn_like(y)
        y.backward(dy)
        W = n64(fa.weight) if m is fa else np.sign(n64(bl.weight)) * float(bl.scale)
        xq, dq = q(n64(x), dtype), n64(dy)
        name = type(m).__name__
        check(y, xq @ W.T, tol, name + " y")
        check(xin.grad, dq @ W, tol * 2, name + " dx")
        if m is fa:
            assert y.shape == (3, 65, 257)
            check(fa.weight.grad, np.einsum("bno,bnd->od", dq, xq), tol * 2, name + " dW")
        else:
            assert bl.weight.grad is None or float(bl.weight.grad.abs().max()) == 0.0  # sign() passes no gradient
            check(bl.scale.grad, np.array([(dq * (xq @ np.sign(n64(bl.weight)).T)).sum()]), tol * 4, name + " dscale")


def test_dropout_kernel(ops):
    x = torch.ones(1 << 20, device=dev(), requires_grad=True)
    y = ops.DropoutFn.apply(x, 0.1)
    frac = (y != 0).float().mean().item()
    assert abs(frac - 0.9) < 3e-3
    assert abs(y.mean().item() - 1.0) < 5e-3
    y.backward(torch.ones_like(y))
    assert torch.equal(x.grad, y.detach())  # same mask, same scale


def test_cpu_tensor_fails_loudly(ops):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.FNetMixFn.apply(torch.randn(2, 5, 16))


# ------------------------------------------------------------------------------------------------ Hadamard helpers
@pytest.fixture(scope="module")
def golden_hadamard():
    import os
    from conftest import GOLDEN
    return dict(np.load(os.path.join(GOLDEN, "hadamard.npz")))


@pytest.mark.parametrize("n", [2, 8, 64, 512])
def test_hadamard_helpers_golden(golden_hadamard, n):
    """fwht / fwht_fast / hadamard_transform (reference hadamar.py:12-32, 58-80, 83-112) on the HIP butterfly kernel against the
    reference's own outputs and gradients (fp32 kernels, float64 fixtures)."""
    from spectre_vit.models.spectre import hadamar as H
    g = golden_hadamard
    for name, fn in (("fwht", H.fwht), ("fwht_raw", lambda v: H.fwht(v, normalize=False)), ("fwht_fast", H.fwht_fast)):
        x = t(g[f"{name}.{n}.x"]).requires_grad_(True)
        y = fn(x)
        y.backward(t(g[f"{name}.{n}.dy"]))
        check(y, g[f"{name}.{n}.y"], 2e-6, f"{name} n={n} y")
        check(x.grad, g[f"{name}.{n}.dx"], 2e-6, f"{name} n={n} dx")
    check(H.hadamard_transform(t(g[f"hadamard_transform.{n}.x"])), g[f"hadamard_transform.{n}.y"], 2e-6, "hadamard_transform")
    check(H.hadamard_transform(t(g[f"hadamard_transform.{n}.x"][0])), g[f"hadamard_transform.{n}.y"][0], 2e-6, "hadamard_transform 1-D")


def test_fwht_other_axis_and_errors(golden_hadamard):
    from spectre_vit.models.spectre import hadamar as H
    g = golden_hadamard
    check(H.fwht(t(g["fwht_dim1.x"]), dim=1), g["fwht_dim1.y"], 2e-6, "fwht dim=1")
    with pytest.raises(ValueError):
        H.fwht_fast(torch.zeros(2, 12, device=dev()))
    with pytest.raises(AssertionError):
        H.hadamard_transform(torch.zeros(2, 2, 8, device=dev()))
    assert H.next_pow2(100) == 128 and H.next_pow2(64) == 64


@pytest.mark.parametrize("dim,blocks", [(48, 2), (64, 1), (100, 3)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_learnable_hadamard_golden(golden_hadamard, dim, blocks, dtype):
    """LearnableHadamard (hadamar.py:115-141): pad / num_blocks x fwht_fast / crop / + residual as one kernel; same state_dict keys;
    the parameters take no part in the forward and get no gradient, as in the reference."""
    from spectre_vit.models.spectre.hadamar import LearnableHadamard
    g = golden_hadamard
    key = f"lh.{dim}.{blocks}"
    m = LearnableHadamard(dim, num_blocks=blocks).to(dev())
    assert ",".join(m.state_dict().keys()) == str(g[key + ".state_keys"])
    xq, dyq = q(g[key + ".x"], dtype), q(g[key + ".dy"], dtype)
    x = t(xq, dtype).requires_grad_(True)
    y = m(x)
    y.backward(t(dyq, dtype))
    tol = 2e-6 if dtype == torch.float32 else 8e-3  # bf16: output rounding only (fp32 butterflies in LDS)
    check(y, O.learnable_hadamard_fwd(xq, blocks), tol, "y")
    check(x.grad, O.learnable_hadamard_bwd(dyq, blocks), tol, "dx")
    if dtype == torch.float32:
        check(y, g[key + ".y"], 2e-6, "y vs reference")
        check(x.grad, g[key + ".dx"], 2e-6, "dx vs reference")
    assert all(p.grad is None for p in m.params)


# ------------------------------------------------------------------------------------------------ class head + cross-entropy
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,N,E,n", [(512, 3, 512, 100), (6, 2, 64, 10), (9, 5, 768, 100), (5, 1, 64, 64), (130, 2, 1024, 128)])
def test_cls_head_vs_oracle(ops, dtype, B, N, E, n):
    """ClsHeadFn: SpectreLinear((out + src)[:, 0]) (reference spectre.py:198-202, layers.py:95-101) in one launch over the fp32
    master weights, vs the float64 oracle on the same (dtype-rounded) inputs.  The arithmetic is fp32 for both input dtypes."""
    rng = np.random.default_rng(B * 13 + E + n)
    out = q(rng.standard_normal((B, N, E)), dtype)
    src = q(rng.standard_normal((B, E)), dtype)
    W = q(rng.standard_normal((n, E)) / np.sqrt(E), torch.float32)
    b, g, be = (q(rng.standard_normal(n) * 0.1, torch.float32), q(rng.random(n) + 0.5, torch.float32),
                q(rng.standard_normal(n) * 0.1, torch.float32))
    dy = q(rng.standard_normal((B, n)), torch.float32)
    dfe = q(rng.standard_normal((B, E)), torch.float32)
    x = out[:, 0, :] + src
    y_ref, dx_ref, gr = sl_oracle(x, W, b, g, be, dy)
    Out, Src = t(out, dtype).requires_grad_(True), t(src, dtype).requires_grad_(True)
    Wt, bt, gt, bet = (t(v).requires_grad_(True) for v in (W, b, g, be))
    logits, feats = ops.ClsHeadFn.apply(Out, Src, Wt, bt, gt, bet)
    assert logits.dtype == torch.float32 and feats.dtype == torch.float32
    (logits * t(dy)).sum().backward(retain_graph=True)
    tol = 2e-5
    check(logits, y_ref, tol, "logits")
    check(feats, x, 1e-6, "features")
    dtol = tol * 2 if dtype == torch.float32 else 8e-3   # dx leaves in the stack's dtype
    check(Src.grad, dx_ref, dtol, "d src_cls")
    check(Out.grad[:, 0, :], dx_ref, dtol, "d out[:, 0]")
    assert N == 1 or float(Out.grad[:, 1:, :].abs().max()) == 0.0
    check(Wt.grad, gr["weight"], tol * 2, "dW")
    check(bt.grad, gr["bias"], tol * 2, "db")
    check(gt.grad, gr["ln_weight"], tol * 2, "dgamma")
    check(bet.grad, gr["ln_bias"], tol * 2, "dbeta")
    # a gradient arriving on the features (distillation's feature loss) is added to the input gradient
    Src.grad = None
    Out.grad = None
    (feats * t(dfe)).sum().backward()
    check(Src.grad, dfe, 8e-3 if dtype == torch.bfloat16 else 1e-6, "d src_cls via features")


@pytest.mark.parametrize("rows,C", [(512, 100), (7, 1000), (33, 3), (1, 10), (4096, 10)])
def test_cross_entropy_vs_torch_and_oracle(ops, rows, C):
    """spv_cross_entropy_fwd/bwd = nn.CrossEntropyLoss() (reference train.py:196,226): loss and dlogits vs torch's fp32 kernels and vs
    the float64 oracle, with an upstream factor on the loss (the distillation step scales it by 0.75)."""
    rng = np.random.default_rng(rows + C)
    z = rng.standard_normal((rows, C)) * 3.0
    y = rng.integers(0, C, rows)
    Z = t(z).requires_grad_(True)
    Y = torch.from_numpy(y).to(dev())
    loss = ops.cross_entropy(Z, Y)
    (loss * 0.75).backward()
    Zr = t(z).requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(Zr, Y)
    (ref * 0.75).backward()
    l64, d64 = O.cross_entropy_fwd_bwd(n64(t(z)), y)
    assert abs(loss.item() - l64) <= 2e-6 * abs(l64) + 1e-7, (loss.item(), l64)
    assert abs(loss.item() - ref.item()) <= 2e-6 * abs(l64) + 1e-7
    check(Z.grad, 0.75 * d64, 2e-6, "dlogits vs oracle")
    check(Z.grad, n64(Zr.grad), 2e-6, "dlogits vs torch")
    again = ops.cross_entropy(Z, Y)   # the arrival counter re-arms itself: same result on the next launch, bit for bit
    assert again.item() == loss.item()


def test_cross_entropy_module_contract():
    from spectre_vit.loss import CrossEntropyLoss
    with pytest.raises(NotImplementedError):
        CrossEntropyLoss(label_smoothing=0.1)
    crit = CrossEntropyLoss()
    with pytest.raises(RuntimeError):
        crit(torch.zeros(2, 3), torch.zeros(2, dtype=torch.long))
    z = torch.randn(8, 5, device=dev())
    bad = torch.full((8,), 7, device=dev())
    assert torch.isnan(crit(z, bad))  # a label outside [0, classes) poisons the loss; it never reads out of bounds


@pytest.mark.parametrize("rows,n,k", [(33280, 768, 512), (1300, 512, 768), (700, 64, 64)])
def test_fold_rides_in_the_split_k_reduce(ops, monkeypatch, rows, n, k):
    """spv_gemm_tn_fold: the tail backward's dgamma / dbeta / dbias fold as extra workgroups of the weight gradient's split-K reduce
    must give, bit for bit, what the tail backward's own fold launch and a plain spv_gemm_tn give (same sums, same order)."""
    torch.manual_seed(rows + n)
    bf = torch.bfloat16
    x = torch.randn(rows, k, device=dev()).to(bf)
    W = (torch.randn(n, k, device=dev()) / k ** 0.5)
    b, g, be = torch.randn(n, device=dev()) * 0.1, torch.rand(n, device=dev()) + 0.5, torch.randn(n, device=dev()) * 0.1
    dy = torch.randn(rows, n, device=dev()).to(bf)

    def run():
        ps = [t_.clone().requires_grad_(True) for t_ in (W, b, g, be)]
        xin = x.clone().requires_grad_(True)
        ops.spectre_linear(xin, *ps, 0.0, False).backward(dy)
        return [xin.grad] + [p_.grad for p_ in ps]

    assert ops._fold_rides(bf, rows, n, k)
    rode = run()
    monkeypatch.setattr(ops, "FOLD_RIDE", False)
    assert not ops._fold_rides(bf, rows, n, k)
    plain = run()
    for a, c, name in zip(rode, plain, ("dx", "dW", "dbias", "dgamma", "dbeta")):
        assert torch.equal(a, c), name


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_embedding_dropout_in_the_gemm_epilogue_matches_the_dropout_kernel(ops, dtype):
    """PatchEmbedFn with p > 0: the mask applied in the token GEMM's epilogue must be the one spv_dropout draws from the same seed on
    the flat token index (and the backward's spv_embed_bwd re-derives): compare with the unfused composition bit for bit."""
    from spectre_vit import _native
    B, C, H, P, E = 6, 3, 32, 4, 64
    g = torch.Generator().manual_seed(3)
    img = torch.randn(B, C, H, H, generator=g).to(dev())
    K = C * P * P
    w_full = (torch.randn(E, K, generator=g) / K ** 0.5).to(dev())
    bias, cls = torch.randn(E, generator=g).to(dev()), torch.randn(1, 1, E, generator=g).to(dev())
    T = (H // P) ** 2 + 1
    pos = torch.randn(1, T, E, generator=g).to(dev())
    base = ops.PatchEmbedFn.apply(img, w_full, bias, cls, pos, P, dtype, None, 0.0)
    calls = []
    orig = ops._new_seed
    ops._new_seed = lambda: (calls.append(1) or 12345)
    try:
        fused = ops.PatchEmbedFn.apply(img, w_full, bias, cls, pos, P, dtype, None, 0.25)
    finally:
        ops._new_seed = orig
    ref = torch.empty_like(base)
    _native.call("spv_dropout", base.data_ptr(), ref.data_ptr(), base.numel(), 0.25, 12345, 1 if dtype == torch.bfloat16 else 0,
                 torch.cuda.current_stream().cuda_stream)
    kept = (fused != 0).float().mean().item()
    assert abs(kept - 0.75) < 0.02, kept
    assert torch.equal(fused == 0, ref == 0)
    torch.testing.assert_close(fused.float(), ref.float(), rtol=2e-2 if dtype == torch.bfloat16 else 1e-6, atol=1e-6)


@pytest.mark.parametrize("B,N,D", [(4, 65, 512), (3, 7, 1024), (512, 65, 512)])
def test_haar_ln_residual_fused_vs_unfused_and_oracle(ops, B, N, D):
    """HaarResidualFn (one row kernel each way) vs the composition it replaces -- HaarDWTFn + add_layernorm(mode 0) -- and vs the
    float64 oracle: out = LN(haar(x)) * gamma + beta + x, dx, dgamma, dbeta.  bf16 storage on both sides."""
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(B + D)
    x = torch.randn(B, N, D, generator=g).to(dev()).to(bf)
    gam = (torch.rand(D, generator=g) + 0.5).to(dev())
    bet = (torch.randn(D, generator=g) * 0.1).to(dev())
    dy = torch.randn(B, N, D, generator=g).to(dev()).to(bf)

    def run(fused):
        xi = x.clone().requires_grad_(True)
        gi, bi = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
        if fused:
            y = ops.HaarResidualFn.apply(xi, gi, bi)
        else:
            y = ops.add_layernorm(ops.HaarDWTFn.apply(xi, 2, 1), xi, gi, bi, 0)
        y.backward(dy)
        return y, xi.grad, gi.grad, bi.grad

    assert ops.haar_ln_ok(x, "embed", 1)
    yf, dxf, dgf, dbf = run(True)
    yu, dxu, dgu, dbu = run(False)
    # against the unfused kernels: same arithmetic up to the summation order of the statistics and one rounding less in dx
    check(yf, n64(yu), 1e-2, "out vs unfused")
    check(dxf, n64(dxu), 1.5e-2, "dx vs unfused")
    check(dgf, n64(dgu), 2e-3, "dgamma vs unfused")
    check(dbf, n64(dbu), 2e-3, "dbeta vs unfused")
    if B * N <= 1000:   # oracle (float64) on the small cases
        xs = n64(x)
        m = O.haar_dwt_fwd(xs, axis=-1, levels=1)
        ln, cache = O.layernorm_fwd(m, n64(gam), n64(bet))
        check(yf, ln + xs, 2e-2, "out vs oracle")
        dm, dg64, db64 = O.layernorm_bwd(n64(dy), n64(gam), cache)
        check(dxf, O.haar_dwt_bwd(dm, axis=-1, levels=1) + n64(dy), 3e-2, "dx vs oracle")
        check(dgf, dg64, 2e-2, "dgamma vs oracle")
        check(dbf, db64, 2e-2, "dbeta vs oracle")


@pytest.mark.parametrize("cin,cout", [(768, 3072), (3072, 768)])
def test_spectre_linear_dropout_base_widths(ops, cin, cout):
    """The Base-width tail kernels (four waves per 3072-wide row; 48 inputs per lane for 3072 -> 768) with dropout on: the forward's mask
    (read off the zeros) must be the one the backward re-derives -- checked against torch autograd on the same composite with that mask."""
    torch.manual_seed(1)
    rows, p = 300, 0.25
    X = torch.randn(rows, cin, device=dev())
    W = torch.randn(cout, cin, device=dev()) / cin ** 0.5
    b, g, be = torch.randn(cout, device=dev()) * 0.1, torch.rand(cout, device=dev()) + 0.5, torch.randn(cout, device=dev()) * 0.1
    dy = torch.randn(rows, cout, device=dev())
    Xg = X.clone().requires_grad_(True)
    ps = [t_.clone().requires_grad_(True) for t_ in (W, b, g, be)]
    y = ops.spectre_linear(Xg, *ps, p, False)
    y.backward(dy)
    kept = (y != 0)
    assert abs(kept.float().mean().item() - (1 - p)) < 0.01
    Xr = X.clone().requires_grad_(True)
    pr = [t_.clone().requires_grad_(True) for t_ in (W, b, g, be)]
    h = torch.nn.functional.linear(Xr, pr[0], pr[1])
    f = torch.nn.functional.gelu(torch.nn.functional.layer_norm(h, (cout,), pr[2], pr[3]))
    f = f + torch.nn.functional.adaptive_avg_pool1d(Xr.unsqueeze(1), cout).squeeze(1)
    ref = f * kept / (1 - p)
    ref.backward(dy)
    check(y, n64(ref), 3e-5, "y")
    check(Xg.grad, n64(Xr.grad), 1e-4, "dx")
    for a_, r_, name in zip(ps, pr, ("dW", "db", "dgamma", "dbeta")):
        check(a_.grad, n64(r_.grad), 1e-4, name)


@pytest.mark.parametrize("M,N,K,pw", [(8320, 1024, 128, 16), (8320, 768, 128, 12), (8352, 1536, 256, 24), (300, 96, 64, 12)])
def test_gemm_pool_bwd_windows(ops, M, N, K, pw):
    """spv_gemm_nt_pool_bwd: C = A B^T + dout[:, col / pw] / pw (the transposed exact-window pooling of the MHPermutMix linear's skip,
    layers.py:66,93) -- on the strip kernel for M >= 8192 (windows of any width >= 8: a lane's 8 columns span at most two), on the
    128 x 128 kernel otherwise; vs float64."""
    from spectre_vit import _native
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(M + pw)
    A = torch.randn(M, K, generator=g).to(dev()).to(bf)
    B = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev()).to(bf)
    D = torch.randn(M, N // pw, generator=g).to(dev()).to(bf)
    C = torch.empty(M, N, device=dev(), dtype=bf)
    _native.call("spv_gemm_nt_pool_bwd", A.data_ptr(), B.data_ptr(), C.data_ptr(), D.data_ptr(), pw, M, N, K, K, K, N, 1, 1, 1,
                 torch.cuda.current_stream().cuda_stream)
    ref = n64(A) @ n64(B).T + np.repeat(n64(D), pw, axis=1) / pw
    check(C, ref, 1.5e-2, f"pool_bwd pw={pw}")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(6, 65, 512), (3, 17, 256), (2, 5, 1024)])
def test_fnet_cls_row_matches_the_full_node(dtype, shape):
    """FNetClsFn (row 0 of mixer + LayerNorm-1 + residual from ONE FFT of the token sum) against row 0 of the definition in float64:
    output, input gradient (every token the same spectrum, row 0 + the residual's) and the LayerNorm parameter gradients."""
    from spectre_vit import hip_ops
    d = torch.device("cuda:0")
    B, N, D = shape
    g = torch.Generator().manual_seed(B * 1000 + N)
    x = torch.randn(B, N, D, generator=g)
    w = 1.0 + 0.2 * torch.randn(D, generator=g)
    b = 0.2 * torch.randn(D, generator=g)
    go = torch.randn(B, D, generator=g)
    # float64 definition (reference spectre.py:66 with the FFT mixer), row 0
    x64 = x.double().requires_grad_(True)
    w64, b64 = w.double().requires_grad_(True), b.double().requires_grad_(True)
    m = torch.fft.fft(torch.fft.fft(x64, dim=-1), dim=-2).real
    ref = (torch.nn.functional.layer_norm(m, (D,), w64, b64, 1e-5) + x64)[:, 0, :]
    ref.backward(go.double())
    xd = x.to(d).to(dtype).requires_grad_(True)
    wd, bd = w.to(d).requires_grad_(True), b.to(d).requires_grad_(True)
    assert hip_ops.fnet_cls_ok(xd)
    out = hip_ops.FNetClsFn.apply(xd, wd, bd)
    out.backward(go.to(d).to(dtype))
    tol = 2e-5 if dtype == torch.float32 else 1.5e-2

    def rel(a, r):
        return ((a.double().cpu() - r).norm() / r.norm()).item()
    # the bf16 input is a rounded x: compare against the definition on the SAME rounded input
    if dtype == torch.bfloat16:
        x64b = x.to(dtype).double().requires_grad_(True)
        w64b, b64b = w.double().requires_grad_(True), b.double().requires_grad_(True)
        mb = torch.fft.fft(torch.fft.fft(x64b, dim=-1), dim=-2).real
        refb = (torch.nn.functional.layer_norm(mb, (D,), w64b, b64b, 1e-5) + x64b)[:, 0, :]
        refb.backward(go.to(dtype).double())
        ref, x64, w64, b64 = refb, x64b, w64b, b64b
    assert rel(out.detach(), ref.detach()) < tol
    assert rel(xd.grad, x64.grad) < tol
    assert rel(wd.grad, w64.grad) < tol and rel(bd.grad, b64.grad) < tol
