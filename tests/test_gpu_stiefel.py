"""GPU parity tests: sparse SpMM, Stiefel manifold kernels, the Rayleigh-quotient quadratic model
and the fused STPCG on its Riemannian Hessian -- against the CPU oracle at oracle-sized problems,
and through size-independent properties at BASELINE cfg2's full size (St(1e6,3), 100^3 grid)."""
import numpy as np
import pytest

from conftest import rel_err
from optimization_amd import workloads as wl

pytestmark = pytest.mark.gpu


def _random_csr(n, seed, max_row=9, empty_every=0):
    rng = np.random.default_rng(seed)
    rowptr = [0]
    col, val = [], []
    for i in range(n):
        k = 0 if (empty_every and i % empty_every == 0) else int(rng.integers(1, max_row + 1))
        c = np.unique(rng.integers(0, n, size=k))
        col.extend(c.tolist())
        val.extend(rng.normal(size=c.size).tolist())
        rowptr.append(len(col))
    return np.array(rowptr, np.int32), np.array(col, np.int32), np.array(val, np.float64)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 257, 5000])
@pytest.mark.parametrize("p", [1, 2, 3, 4, 5, 6, 7, 8])
def test_spmm_ragged_vs_oracle(ctx, oracle, n, p):
    import ctypes as C
    rowptr, col, val = _random_csr(n, seed=100 * n + p, empty_every=5 if n > 4 else 0)
    V = np.random.default_rng(p).normal(size=(n, p))
    A = ctx.csr(n, rowptr, col, val)
    W = A.spmm(p, ctx.upload(V)).numpy().reshape(n, p)
    Wo = np.zeros((n, p))
    ip = C.POINTER(C.c_int)
    dp = C.POINTER(C.c_double)
    Vc = np.ascontiguousarray(V)
    oracle.lib.orc_csr_spmm(n, p, rowptr.ctypes.data_as(ip), col.ctypes.data_as(ip), val.ctypes.data_as(dp),
                            Vc.ctypes.data_as(dp), Wo.ctypes.data_as(dp))
    assert np.allclose(W, Wo, rtol=1e-13, atol=1e-13)


def test_spmm_laplacian_eigvec(ctx):
    nx, ny, nz = 17, 13, 11
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    n = nx * ny * nz
    A = ctx.csr(n, rowptr, col, val)
    for mode in [(1, 1, 1), (2, 1, 3)]:
        v, lam = wl.laplacian_3d_eigvec(nx, ny, nz, *mode)
        V = np.stack([v, 2 * v, -v], axis=1)
        W = A.spmm(3, ctx.upload(V)).numpy().reshape(n, 3)
        assert np.allclose(W, (lam + 0.1) * V, atol=1e-12)


@pytest.fixture(scope="module")
def small_rq(ctx, oracle):
    nx, ny, nz, p = 12, 11, 10, 3
    n = nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    X0 = wl.random_stiefel(n, p, seed=3)
    A = ctx.csr(n, rowptr, col, val)
    prob = ctx.stiefel_rq(A, n, p)
    oprob = oracle.stiefel_rq(n, p, rowptr, col, val)
    yield dict(n=n, p=p, A=A, prob=prob, oprob=oprob, X0=X0, csr=(rowptr, col, val))
    oracle.free(oprob)


def test_stiefel_gram_project_retract(ctx, oracle, small_rq):
    n, p, X0 = small_rq["n"], small_rq["p"], small_rq["X0"]
    rng = np.random.default_rng(0)
    Z = rng.normal(size=(n, p))
    X, Zd = ctx.upload(X0), ctx.upload(Z)
    G = ctx.stiefel_gram(n, p, X, Zd)
    assert np.allclose(G, X0.T @ Z, rtol=1e-12, atol=1e-13)
    Pz = ctx.stiefel_project(n, p, X, Zd).numpy().reshape(n, p)
    M = X0.T @ Z
    assert np.allclose(Pz, Z - X0 @ (0.5 * (M + M.T)), atol=1e-13)
    # tangent: X' P_X(Z) is skew
    S = X0.T @ Pz
    assert np.abs(S + S.T).max() < 1e-13
    V = 0.3 * Pz
    Y = ctx.stiefel_retract(n, p, X, ctx.upload(V)).numpy().reshape(n, p)
    Yo = oracle.eval_retract(small_rq["oprob"], X0.ravel(), V.ravel()).reshape(n, p)
    assert rel_err(Y, Yo) < 1e-13
    assert np.abs(Y.T @ Y - np.eye(p)).max() < 1e-13
    # zero step retracts to the point itself
    Y0 = ctx.stiefel_retract(n, p, X, ctx.vec(n * p).fill(0.0)).numpy().reshape(n, p)
    assert np.abs(Y0 - X0).max() < 1e-14


def test_rq_objective_gradient_hessian_vs_oracle(ctx, oracle, small_rq):
    n, p, X0, prob, oprob = (small_rq[k] for k in ("n", "p", "X0", "prob", "oprob"))
    X = ctx.upload(X0)
    f = prob.objective(X)
    fo = oracle.eval_f(oprob, X0.ravel())
    assert abs(f - fo) <= 1e-13 * abs(fo)
    g, H = prob.model(X)
    go = oracle.eval_grad(oprob, X0.ravel())
    assert rel_err(g.numpy(), go) < 1e-13
    rng = np.random.default_rng(1)
    for _ in range(3):
        Z = rng.normal(size=(n, p))
        V = ctx.stiefel_project(n, p, X, ctx.upload(Z))
        Hv = H.apply(V).numpy()
        Hvo = oracle.eval_hess(oprob, X0.ravel(), V.numpy())
        assert rel_err(Hv, Hvo) < 1e-13
    # self-adjointness on the tangent space
    U = ctx.stiefel_project(n, p, X, ctx.upload(rng.normal(size=(n, p))))
    V = ctx.stiefel_project(n, p, X, ctx.upload(rng.normal(size=(n, p))))
    a, b = U.dot(H.apply(V)), V.dot(H.apply(U))
    assert abs(a - b) <= 1e-12 * max(abs(a), abs(b))


@pytest.mark.parametrize("Delta,maxit,kappa", [(1.0, 50, .1), (1e6, 50, 1e-6), (0.05, 50, .1)])
def test_fused_stpcg_on_rq_hessian_vs_oracle(ctx, oracle, small_rq, Delta, maxit, kappa):
    n, p, prob, oprob = (small_rq[k] for k in ("n", "p", "prob", "oprob"))
    # a point near the minimiser so that the Hessian is PSD and CG runs many iterations
    Xb, _ = wl.stiefel_bench_iterate(12, 11, 10, p, eps=1e-2, seed=5)
    X = ctx.upload(Xb)
    g, H = prob.model(X)
    go = oracle.eval_grad(oprob, Xb.ravel())
    r = ctx.stpcg(g, H, Delta=Delta, max_iterations=maxit, kappa_fgr=kappa, theta=.5, trace_cap=64)
    o = oracle.stpcg_problem(oprob, Xb.ravel(), go, Delta, max_iterations=maxit, kappa_fgr=kappa, theta=.5,
                             trace_cap=64)
    assert r["iterations"] == o["iterations"]
    assert r["exit_reason"] == o["exit_reason"]
    for k in ("alpha", "beta", "kappa", "rv"):
        assert np.allclose(r["trace"][k], o["trace"][k], rtol=1e-9), k
    assert rel_err(r["s"].numpy(), o["s"]) < 1e-10
    assert abs(r["M_norm"] - o["M_norm"]) <= 1e-10 * o["M_norm"]


def test_fused_stpcg_random_point_negative_curvature(ctx, oracle, small_rq):
    n, p, X0, prob, oprob = (small_rq[k] for k in ("n", "p", "X0", "prob", "oprob"))
    X = ctx.upload(X0)
    g, H = prob.model(X)
    go = oracle.eval_grad(oprob, X0.ravel())
    r = ctx.stpcg(g, H, Delta=1.0, max_iterations=50, trace_cap=64)
    o = oracle.stpcg_problem(oprob, X0.ravel(), go, 1.0, max_iterations=50, trace_cap=64)
    assert r["iterations"] == o["iterations"] and r["exit_reason"] == o["exit_reason"]
    assert rel_err(r["s"].numpy(), o["s"]) < 1e-10


def test_rq_projected_jacobi_precon_vs_oracle(ctx, oracle, small_rq):
    n, p = small_rq["n"], small_rq["p"]
    rowptr, col, val = small_rq["csr"]
    diag = np.array([val[rowptr[i]:rowptr[i + 1]][col[rowptr[i]:rowptr[i + 1]] == i][0] for i in range(n)])
    dinv = 1.0 / (diag * np.linspace(0.5, 2.0, n))
    oprob = oracle.stiefel_rq(n, p, rowptr, col, val, dinv=dinv)
    Xb, _ = wl.stiefel_bench_iterate(12, 11, 10, p, eps=1e-2, seed=5)
    X = ctx.upload(Xb)
    g, H = small_rq["prob"].model(X)
    P = small_rq["prob"].precon(X, ctx.upload(dinv))
    go = oracle.eval_grad(oprob, Xb.ravel())
    pv = P.apply(g).numpy()
    assert rel_err(pv, oracle.eval_precon(oprob, Xb.ravel(), go)) < 1e-12
    r = ctx.stpcg(g, H, P, Delta=1e3, max_iterations=40, kappa_fgr=1e-4, trace_cap=64)
    o = oracle.stpcg_problem(oprob, Xb.ravel(), go, 1e3, max_iterations=40, kappa_fgr=1e-4, trace_cap=64)
    assert r["iterations"] == o["iterations"] and r["exit_reason"] == o["exit_reason"]
    assert np.allclose(r["trace"]["alpha"], o["trace"]["alpha"], rtol=1e-9)
    assert rel_err(r["s"].numpy(), o["s"]) < 1e-9
    oracle.free(oprob)


# ----------------------------------------------------------------------------------------------
# BASELINE cfg2 full size: St(1e6, 3) on the 100^3 Laplacian
# ----------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def full_rq(ctx):
    nx = ny = nz = 100
    n, p = nx * ny * nz, 3
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    assert rowptr[-1] == 6_940_000
    A = ctx.csr(n, rowptr, col, val)
    prob = ctx.stiefel_rq(A, n, p)
    return dict(n=n, p=p, A=A, prob=prob, csr=(rowptr, col, val))


def test_full_size_properties(ctx, full_rq):
    n, p, prob = full_rq["n"], full_rq["p"], full_rq["prob"]
    nx = ny = nz = 100
    # exact eigen-subspace: f = .5 sum (lambda_i + .1), gradient = 0
    Xs = np.stack([wl.laplacian_3d_eigvec(nx, ny, nz, *m)[0] for m in [(1, 1, 1), (2, 1, 1), (1, 2, 1)]], axis=1)
    lams = [wl.laplacian_3d_eigvec(nx, ny, nz, *m)[1] + 0.1 for m in [(1, 1, 1), (2, 1, 1), (1, 2, 1)]]
    X = ctx.upload(Xs)
    assert abs(prob.objective(X) - 0.5 * sum(lams)) < 1e-12
    g, H = prob.model(X)
    assert np.sqrt(g.dot(g)) < 1e-12
    # random point: tangency, self-adjointness, retraction feasibility
    X0 = wl.random_stiefel(n, p)
    X = ctx.upload(X0)
    g, H = prob.model(X)
    S = ctx.stiefel_gram(n, p, X, g)
    assert np.abs(S + S.T).max() < 1e-11
    rng = np.random.default_rng(2)
    U = ctx.stiefel_project(n, p, X, ctx.upload(rng.normal(size=(n, p))))
    V = ctx.stiefel_project(n, p, X, ctx.upload(rng.normal(size=(n, p))))
    HU, HV = H.apply(U), H.apply(V)
    a, b = U.dot(HV), V.dot(HU)
    assert abs(a - b) <= 1e-11 * max(abs(a), abs(b))
    S = ctx.stiefel_gram(n, p, X, HV)
    assert np.abs(S + S.T).max() < 1e-10
    Y = ctx.stiefel_retract(n, p, X, V.copy().scale(1e-3))
    G = ctx.stiefel_gram(n, p, Y, Y)
    assert np.abs(G - np.eye(p)).max() < 1e-13
    # linearity of the fused Hessian operator
    W = ctx.vec(n * p).axpby(2.0, U, -3.0, V)
    HW = H.apply(W)
    lin = ctx.vec(n * p).axpby(2.0, HU, -3.0, HV)
    assert rel_err(HW.numpy(), lin.numpy()) < 1e-13


def test_full_size_fused_stpcg_vs_oracle(ctx, oracle, full_rq):
    """12 inner iterations at N = 3e6 against the oracle run on the same arrays."""
    n, p, prob = full_rq["n"], full_rq["p"], full_rq["prob"]
    rowptr, col, val = full_rq["csr"]
    Xb, _ = wl.stiefel_bench_iterate(100, 100, 100, p, eps=1e-3, seed=7)
    X = ctx.upload(Xb)
    g, H = prob.model(X)
    oprob = oracle.stiefel_rq(n, p, rowptr, col, val)
    go = oracle.eval_grad(oprob, Xb.ravel())
    assert rel_err(g.numpy(), go) < 1e-11
    r = ctx.stpcg(g, H, Delta=1e3, max_iterations=12, kappa_fgr=1e-12, theta=1.0, trace_cap=16)
    o = oracle.stpcg_problem(oprob, Xb.ravel(), go, 1e3, max_iterations=12, kappa_fgr=1e-12, theta=1.0,
                             trace_cap=16)
    assert r["iterations"] == o["iterations"] == 12
    assert np.allclose(r["trace"]["alpha"], o["trace"]["alpha"], rtol=1e-9)
    assert np.allclose(r["trace"]["beta"], o["trace"]["beta"], rtol=1e-8)
    assert rel_err(r["s"].numpy(), o["s"]) < 1e-10  # BASELINE.json: iterate match within 1e-10 relative
    oracle.free(oprob)


@pytest.mark.parametrize("p", [1, 2, 3, 4])
def test_one_pass_hessian_matches_two_pass_and_oracle(oracle, oracle_omp, monkeypatch, p):
    """STPCG's one-pass Stiefel Hessian (mi_op::dirgram) in both forms -- Gram of the direction carried by
    scalar recurrences (default without a preconditioner) and Gram rows formed by the direction kernel
    (MI355OPT_DIRGRAM_DIRECT=1) -- against the two-pass operator (MI355OPT_NO_DIRGRAM=1) and the oracle, for
    every supported p."""
    from optimization_amd import capi
    nx, ny, nz = 9, 8, 7
    n = nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    Xb, _ = wl.stiefel_bench_iterate(nx, ny, nz, p, eps=1e-2, seed=11 + p)
    oprob = oracle.stiefel_rq(n, p, rowptr, col, val)
    go = oracle.eval_grad(oprob, Xb.ravel())
    o = oracle.stpcg_problem(oprob, Xb.ravel(), go, 1e3, max_iterations=40, kappa_fgr=1e-8, theta=1.0,
                             trace_cap=64)
    oracle.free(oprob)
    # the conditioning floor of the late scalars: on this 504-row problem the Hessian is nearly singular at the iterate
    # (the residual RISES again after ~30 iterations), and the reference algorithm itself, sums re-associated, moves
    # beta_40 by 3e-8 at p = 4
    fl = None
    if oracle_omp is not None:
        mprob = oracle_omp.stiefel_rq(n, p, rowptr, col, val)
        oracle_omp.eval_grad(mprob, Xb.ravel())   # (binds the model; the floor solve takes the oracle's g as well)
        fl = oracle_omp.stpcg_problem(mprob, Xb.ravel(), go, 1e3, max_iterations=40, kappa_fgr=1e-8, theta=1.0,
                                      trace_cap=64)["trace"]
        oracle_omp.free(mprob)
    env = {"recurrence": ("0", "0"), "direct": ("0", "1"), "two-pass": ("1", "0")}
    res = {}
    for mode, (no_dirgram, direct) in env.items():
        monkeypatch.setenv("MI355OPT_NO_DIRGRAM", no_dirgram)
        monkeypatch.setenv("MI355OPT_DIRGRAM_DIRECT", direct)
        c = capi.Context(0)
        try:
            A = c.csr(n, rowptr, col, val)
            prob = c.stiefel_rq(A, n, p)
            g, H = prob.model(c.upload(Xb))
            assert rel_err(g.numpy(), go) < 1e-11
            names = ("stiefel_hess_fused", "stiefel_finish_dots")
            for k in names:
                c.ktime_enable(k, True)
            c.ktime_reset()
            # (r06) the solve's input is the oracle's gradient, bit for bit: parity of the solver on identical inputs
            r = c.stpcg(c.upload(go), H, Delta=1e3, max_iterations=40, kappa_fgr=1e-8, theta=1.0, trace_cap=64)
            launches = {k: c.ktime_read(k)[0] for k in names}
            res[mode] = dict(r, s=r["s"].numpy().copy(), launches=launches)
        finally:
            c.close()
    # the path under test really ran
    for mode in ("recurrence", "direct"):
        assert res[mode]["launches"]["stiefel_hess_fused"] > 0
        assert res[mode]["launches"]["stiefel_finish_dots"] == 0
    assert res["two-pass"]["launches"]["stiefel_hess_fused"] == 0
    assert res["two-pass"]["launches"]["stiefel_finish_dots"] > 0
    from conftest import trace_close
    for mode, r in res.items():
        assert r["iterations"] == o["iterations"] and r["exit_reason"] == o["exit_reason"], mode
        for key, tol in (("alpha", 1e-9), ("beta", 1e-8)):
            ok, msg = trace_close(r["trace"][key], o["trace"][key], fl[key] if fl else None, tol)
            assert ok, f"{mode} {key}: {msg}"
        assert rel_err(r["s"], o["s"]) < 1e-10, mode
    assert rel_err(res["recurrence"]["s"], res["two-pass"]["s"]) < 1e-11
    assert rel_err(res["direct"]["s"], res["two-pass"]["s"]) < 1e-11


def _graph_laplacian(n, seed, weights):
    """Connected random graph Laplacian + 0.1 I as CSR (symmetric, SPD).  weights: list to draw from, or None for
    all-distinct random weights (then the matrix has far more than 256 distinct values)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    i = np.concatenate([np.arange(n - 1), rng.integers(0, n, size=2 * n)])
    j = np.concatenate([np.arange(1, n), rng.integers(0, n, size=2 * n)])
    keep = i != j
    i, j = i[keep], j[keep]
    w = rng.choice(weights, size=i.size) if weights is not None else rng.uniform(.5, 2.0, size=i.size)
    W = sp.coo_matrix((w, (i, j)), shape=(n, n)).tocsr()
    W = W.maximum(W.T)  # symmetric, duplicate edges collapsed
    L = (sp.diags(np.asarray(W.sum(axis=1)).ravel() + .1) - W).tocsr()
    L.sort_indices()
    return L


@pytest.mark.parametrize("grid,p", [((24, 20, 16), 3), ((40, 40, 40), 3), ((64, 48, 5), 3), ((30, 30, 30), 2),
                                    ((33, 27, 21), 1), ((100, 10, 9), 3), ((16, 16, 130), 3)])
def test_window_form_of_the_hessian_has_the_bits_of_the_streaming_form(monkeypatch, grid, p):
    """The fused STPCG with the one-pass Hessian in its LDS-window form (sell_window: ring, far rows a tile ahead,
    planned runs cut to the far stride) against the same solve with MI355OPT_NO_WINDOW=1 (sell_stream) and with
    equal runs (MI355OPT_NO_WIN_BOUNDS=1): every output row is formed by the same products and sums in the same
    order; only the partition of the rows into per-workgroup partial sums differs, so the replicated scalars and
    with them the iterates agree to rounding (1e-11), the iteration counts and exit reasons exactly -- on cubes,
    slabs thinner than a run, grids whose plane is smaller than a tile, and p = 1, 2, 3."""
    from optimization_amd import capi
    nx, ny, nz = grid
    n = nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    Xb, _ = wl.stiefel_bench_iterate(nx, ny, nz, p, eps=1e-2, seed=3 + p)
    res = {}
    for mode, env in (("window", {}), ("stream", {"MI355OPT_NO_WINDOW": "1"}),
                      ("window-equal-runs", {"MI355OPT_NO_WIN_BOUNDS": "1"}),
                      ("window-loaded-far", {"MI355OPT_NO_FAR_COMPUTED": "1"}),
                      ("window-words16", {"MI355OPT_WORDS16": "1"})):
        for k in ("MI355OPT_NO_WINDOW", "MI355OPT_NO_WIN_BOUNDS", "MI355OPT_NO_FAR_COMPUTED", "MI355OPT_WORDS16"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = capi.Context(0)
        try:
            A = c.csr(n, rowptr, col, val)
            prob = c.stiefel_rq(A, n, p)
            g, H = prob.model(c.upload(Xb))
            r = c.stpcg(g, H, Delta=1e3, max_iterations=25, kappa_fgr=1e-10, theta=1.0)
            res[mode] = (r["s"].numpy().copy(), r["iterations"], r["exit_reason"], r["M_norm"])
        finally:
            c.close()
    w, st, eq = res["window"], res["stream"], res["window-equal-runs"]
    # far columns computed (row +- D, the default for a pure stencil) or loaded from wfar: the same kernel otherwise,
    # the same partition of the rows: bit-identical
    lf = res["window-loaded-far"]
    assert np.array_equal(w[0], lf[0]) and w[1:] == lf[1:]
    # 16-bit words (opt-in; value table of <= 32 entries) or 32-bit words: the same entries
    w16 = res["window-words16"]
    assert np.array_equal(w[0], w16[0]) and w[1:] == w16[1:]
    assert w[1:3] == st[1:3] == eq[1:3] and w[1] > 3
    # the partial rows are summed per workgroup: the forms partition the rows differently, so the replicated scalars
    # (and with them the iterates) agree to rounding, not to the bit, unless the partitions coincide
    scale = np.abs(st[0]).max()
    assert np.abs(w[0] - st[0]).max() <= 1e-11 * scale
    assert np.abs(eq[0] - st[0]).max() <= 1e-11 * scale
    assert abs(w[3] - st[3]) <= 1e-11 * abs(st[3])


@pytest.mark.parametrize("kind", ["few-values", "signed-zero", "many-values"])
def test_one_pass_hessian_random_graph_all_matrix_formats(oracle, monkeypatch, kind):
    """The one-pass Hessian on an unstructured matrix (ragged rows): value-indexed packed entries (few distinct
    values; one case stores an explicit -0.0 entry, which the table must keep apart from +0.0), the plain
    value/column arrays (too many distinct values, and MI355OPT_NO_PACKED=1), against the two-pass operator and
    the oracle."""
    from optimization_amd import capi
    n, p = 700, 3
    L = _graph_laplacian(n, seed=5, weights=None if kind == "many-values" else [1.0, 2.0, .5])
    rowptr, col, val = L.indptr.astype(np.int32), L.indices.astype(np.int32), L.data.astype(np.float64)
    if kind == "signed-zero":  # one stored off-diagonal pair becomes an explicit -0.0
        r = 3
        k = next(k for k in range(rowptr[r], rowptr[r + 1]) if col[k] != r)
        c = int(col[k])
        k2 = next(k2 for k2 in range(rowptr[c], rowptr[c + 1]) if col[k2] == r)
        val[k] = val[k2] = -0.0
        L = L.copy()
        L.data[:] = val
        assert np.signbit(val[k]) and (val == 0).sum() == 2
    distinct = np.unique(val.view(np.uint64)).size
    assert (distinct > 256) == (kind == "many-values")
    lam, U = np.linalg.eigh(L.toarray())
    rng = np.random.default_rng(9)
    Xb, _ = np.linalg.qr(U[:, :p] + 1e-2 * rng.normal(size=(n, p)))
    Xb = np.ascontiguousarray(Xb)
    oprob = oracle.stiefel_rq(n, p, rowptr, col, val)
    go = oracle.eval_grad(oprob, Xb.ravel())
    o = oracle.stpcg_problem(oprob, Xb.ravel(), go, 1e3, max_iterations=60, kappa_fgr=1e-9, theta=1.0,
                             trace_cap=64)
    oracle.free(oprob)
    assert o["iterations"] >= 10
    res = {}
    for mode, envs in {"packed-or-plain": {}, "plain": {"MI355OPT_NO_PACKED": "1"},
                       "two-pass": {"MI355OPT_NO_DIRGRAM": "1"}}.items():
        for k in ("MI355OPT_NO_PACKED", "MI355OPT_NO_DIRGRAM"):
            monkeypatch.setenv(k, envs.get(k, "0"))
        c = capi.Context(0)
        try:
            A = c.csr(n, rowptr, col, val)
            prob = c.stiefel_rq(A, n, p)
            g, H = prob.model(c.upload(Xb))
            r = c.stpcg(g, H, Delta=1e3, max_iterations=60, kappa_fgr=1e-9, theta=1.0, trace_cap=64)
            res[mode] = dict(r, s=r["s"].numpy().copy())
        finally:
            c.close()
    for mode, r in res.items():
        assert r["iterations"] == o["iterations"] and r["exit_reason"] == o["exit_reason"], mode
        assert np.allclose(r["trace"]["alpha"], o["trace"]["alpha"], rtol=1e-9), mode
        assert rel_err(r["s"], o["s"]) < 1e-10, mode
    # same arithmetic per entry in both matrix formats: identical bits
    assert np.array_equal(res["packed-or-plain"]["s"], res["plain"]["s"])


@pytest.mark.parametrize("n", [40, 64, 300, 1000])
def test_one_pass_hessian_empty_rows_and_ragged_tail(oracle, monkeypatch, n):
    """Rows without entries (whole 64-row slices of width 0 when n is large enough) and a last slice that is
    only partly filled: the one-pass kernel against the two-pass operator and the oracle over the first CG passes."""
    from optimization_amd import capi
    import scipy.sparse as sp
    p = 3
    L = _graph_laplacian(n, seed=n, weights=[1.0, 3.0]).tolil()
    lo, hi = n // 5, n // 5 + max(n // 3, 3)   # rows/columns lo..hi-1 become empty (130+ rows at n = 1000)
    L[lo:hi, :] = 0
    L[:, lo:hi] = 0
    L = sp.csr_matrix(L)
    L.eliminate_zeros()
    L.sort_indices()
    rowptr, col, val = L.indptr.astype(np.int32), L.indices.astype(np.int32), L.data.astype(np.float64)
    assert all(rowptr[r + 1] == rowptr[r] for r in range(lo, hi))
    X0 = wl.random_stiefel(n, p, seed=n + 1)
    oprob = oracle.stiefel_rq(n, p, rowptr, col, val)
    go = oracle.eval_grad(oprob, X0.ravel())
    res = {}
    for maxit in (1, 4):
        o = oracle.stpcg_problem(oprob, X0.ravel(), go, 0.5, max_iterations=maxit, trace_cap=8)
        for mode, nd in (("one-pass", "0"), ("two-pass", "1")):
            monkeypatch.setenv("MI355OPT_NO_DIRGRAM", nd)
            c = capi.Context(0)
            try:
                A = c.csr(n, rowptr, col, val)
                prob = c.stiefel_rq(A, n, p)
                g, H = prob.model(c.upload(X0))
                r = c.stpcg(g, H, Delta=0.5, max_iterations=maxit, trace_cap=8)
                res[mode] = r["s"].numpy().copy()
                assert r["iterations"] == o["iterations"] and r["exit_reason"] == o["exit_reason"], (mode, maxit)
                assert rel_err(res[mode], o["s"]) < 1e-10, (mode, maxit)
            finally:
                c.close()
        assert rel_err(res["one-pass"], res["two-pass"]) < 1e-11
    oracle.free(oprob)


@pytest.mark.parametrize("n", [1, 64, 65, 257, 5000])
@pytest.mark.parametrize("p", [1, 3, 4])
def test_spmm_packed_matrix_ragged_vs_plain_and_oracle(oracle, monkeypatch, n, p):
    """mi_csr_spmm on a ragged random matrix (empty rows, explicit -0.0) whose values come from a small set, so
    that the value-indexed packed copy is what the kernel reads: bit-identical to the plain arrays
    (MI355OPT_NO_PACKED=1) and to the straight core (MI355OPT_NO_SPMM_STREAM=1), and equal to the oracle."""
    import ctypes as C
    from optimization_amd import capi
    rowptr, col, val = _random_csr(n, seed=7 * n + p, empty_every=5 if n > 4 else 0)
    rng = np.random.default_rng(n + p)
    val = rng.choice(np.array([-1.0, .5, 2.0, -0.0]), size=val.size)
    V = rng.normal(size=(n, p))
    Wo = np.zeros((n, p))
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    Vc = np.ascontiguousarray(V)
    oracle.lib.orc_csr_spmm(n, p, rowptr.ctypes.data_as(ip), col.ctypes.data_as(ip), val.ctypes.data_as(dp),
                            Vc.ctypes.data_as(dp), Wo.ctypes.data_as(dp))
    out = {}
    for mode, envs in {"packed": {}, "plain": {"MI355OPT_NO_PACKED": "1"},
                       "straight": {"MI355OPT_NO_SPMM_STREAM": "1"}}.items():
        for k in ("MI355OPT_NO_PACKED", "MI355OPT_NO_SPMM_STREAM"):
            monkeypatch.setenv(k, envs.get(k, "0"))
        c = capi.Context(0)
        try:
            A = c.csr(n, rowptr, col, val)
            out[mode] = A.spmm(p, c.upload(V)).numpy().reshape(n, p).copy()
        finally:
            c.close()
    assert np.array_equal(out["packed"], out["plain"])
    assert np.array_equal(out["packed"], out["straight"])
    assert np.array_equal(out["packed"], Wo)  # same per-row order, products and sums rounded separately


@pytest.mark.parametrize("p", [1, 2, 3, 4, 5, 8])
def test_fused_trial_step_has_the_bits_of_the_separate_calls(ctx, p):
    """mi_stiefel_rq_trial (reference Riemannian/TNT.h:493-512,573-585 as one launch chain, one read-back) against
    the separate calls it replaces -- dot products, retraction, objective, model at the trial point: bit for bit."""
    nx, ny, nz = 20, 17, 13
    n = nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    A = ctx.csr(n, rowptr, col, val)
    prob = ctx.stiefel_rq(A, n, p)
    X = ctx.upload(wl.random_stiefel(n, p, seed=3))
    g, H = prob.model(X)
    h = ctx.stiefel_project(n, p, X, ctx.upload(np.random.default_rng(4).normal(size=(n, p)) * 1e-2))
    # separate calls
    Hh = H.apply(h)
    hh, gh, hHh = ctx.dot_batch([h, g, h], [h, h, Hh])
    Xt_ref = ctx.stiefel_retract(n, p, X, h)
    f_ref = prob.objective(Xt_ref)
    # fused
    Xt, t = prob.trial(X, h, g)
    assert np.array_equal(Xt.numpy(), Xt_ref.numpy())
    assert (t["f"], t["hh"], t["gh"], t["hHh"]) == (f_ref, hh, gh, hHh)
    g2, H2 = prob.model(Xt)          # takes A X+, S+ and the gradient from the trial call
    prob2 = ctx.stiefel_rq(A, n, p)
    g2_ref, H2_ref = prob2.model(Xt_ref)
    assert np.array_equal(g2.numpy(), g2_ref.numpy())
    assert t["grad_sqnorm"] == g2_ref.dot(g2_ref)
    v = ctx.upload(np.random.default_rng(5).normal(size=(n, p)))
    assert np.array_equal(H2.apply(v).numpy(), H2_ref.apply(v).numpy())
    r1 = ctx.stpcg(g2, H2, Delta=10.0, max_iterations=8, kappa_fgr=1e-10, theta=1.0)
    r2 = ctx.stpcg(g2_ref, H2_ref, Delta=10.0, max_iterations=8, kappa_fgr=1e-10, theta=1.0)
    assert np.array_equal(r1["s"].numpy(), r2["s"].numpy())


def test_speculative_trial_model_is_keyed_on_the_vector_contents(ctx):
    """mi_stiefel_rq_trial's cached A X+, S+, gradient must only serve a model() call for the SAME trial
    vector contents -- not an in-place overwritten one, not a new vector that recycled the handle / pooled pointer."""
    nx, ny, nz, p = 12, 10, 9, 3
    n = nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    A = ctx.csr(n, rowptr, col, val)
    prob = ctx.stiefel_rq(A, n, p)
    X = ctx.upload(wl.random_stiefel(n, p, seed=3))
    g, H = prob.model(X)
    h = ctx.stiefel_project(n, p, X, ctx.upload(np.random.default_rng(4).normal(size=(n, p)) * 1e-2))
    other = wl.random_stiefel(n, p, seed=99)
    prob2 = ctx.stiefel_rq(A, n, p)
    g_ref, _ = prob2.model(ctx.upload(other))
    # (i) in-place write to the trial vector
    Xt, _ = prob.trial(X, h, g)
    Xt.set(other)
    g2, _ = prob.model(Xt)
    assert np.array_equal(g2.numpy(), g_ref.numpy())
    # (ii) the trial vector dies; new vectors of the same size recycle its pooled storage (and maybe its handle)
    g, H = prob.model(X)
    Xt, _ = prob.trial(X, h, g)
    del Xt
    for _ in range(4):
        Y = ctx.upload(other)
        g3, _ = prob.model(Y)
        assert np.array_equal(g3.numpy(), g_ref.numpy())
        del Y, g3


def test_trial_cache_sees_writes_through_views_and_announced_raw_writes(ctx):
    """a view shares the generation stamp of the vector that owns the storage, so a write through a view (or
    through the base, for a cache keyed on a view) invalidates the speculative trial model; a write through the raw
    pointer of mi_vec_data is invisible to the library and must be announced with mi_vec_touch."""
    import ctypes
    nx, ny, nz, p = 12, 10, 9, 3
    n = nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    A = ctx.csr(n, rowptr, col, val)
    prob = ctx.stiefel_rq(A, n, p)
    X = ctx.upload(wl.random_stiefel(n, p, seed=3))
    h = ctx.stiefel_project(n, p, X, ctx.upload(np.random.default_rng(4).normal(size=(n, p)) * 1e-2))
    other = wl.random_stiefel(n, p, seed=99)
    g_ref = ctx.stiefel_rq(A, n, p).model(ctx.upload(other))[0].numpy()
    # (i) write through a view of the trial vector
    g, H = prob.model(X)
    Xt, _ = prob.trial(X, h, g)
    Xt.view(0, n * p).set(other)
    assert np.array_equal(prob.model(Xt)[0].numpy(), g_ref)
    # (ii) the cache keyed on a VIEW, the write through its base
    g, H = prob.model(X)
    base = ctx.vec(n * p + 8)
    Xv = base.view(8, n * p)
    ctx.L.mi_vec_copy(Xv.h, X.h)
    gv, Hv = prob.model(Xv)
    assert np.array_equal(gv.numpy(), g.numpy())
    # (iii) a raw-pointer write, announced
    g, H = prob.model(X)
    Xt, _ = prob.trial(X, h, g)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    ctx.sync()
    o = np.ascontiguousarray(other, dtype=np.float64)
    assert hip.hipMemcpy(Xt.data_ptr(), o.ctypes.data, o.nbytes, 1) == 0
    Xt.touch()
    assert np.array_equal(prob.model(Xt)[0].numpy(), g_ref)


def test_deferred_stpcg_result(ctx):
    """mi_stpcg with defer_result: no wait at the exit, s valid in stream order, scalars through mi_stpcg_collect --
    identical to the blocking call; collect after another read-back does not wait again."""
    nx, ny, nz, p = 16, 15, 14, 3
    n = nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    A = ctx.csr(n, rowptr, col, val)
    prob = ctx.stiefel_rq(A, n, p)
    Xb, _ = wl.stiefel_bench_iterate(nx, ny, nz, p, eps=1e-2, seed=5)
    X = ctx.upload(Xb)
    g, H = prob.model(X)
    kw = dict(Delta=1e3, max_iterations=25, kappa_fgr=1e-9, theta=1.0)
    a = ctx.stpcg(g, H, **kw)
    c0 = ctx.sync_count()
    b = ctx.stpcg(g, H, defer=True, **kw)
    assert ctx.sync_count() == c0            # the solve itself did not wait
    nrm = b["s"].dot(b["s"])                 # a read-back behind the solve ...
    c1 = ctx.sync_count()
    r = ctx.stpcg_collect()                  # ... so collecting costs no further wait
    assert ctx.sync_count() == c1
    assert (r["iterations"], r["exit_reason"], r["M_norm"]) == (a["iterations"], a["exit_reason"], a["M_norm"])
    assert np.array_equal(a["s"].numpy(), b["s"].numpy()) and nrm > 0
    # collect straight away: waits once
    b = ctx.stpcg(g, H, defer=True, **kw)
    r = ctx.stpcg_collect()
    assert ctx.sync_count() in (c1, c1 + 1)   # (no wait at all when the solve had finished before the call)
    assert r["iterations"] == a["iterations"] and r["M_norm"] == a["M_norm"]


def test_unsymmetric_matrix_keeps_the_two_pass_hessian(ctx, oracle):
    """the one-pass Hessian replaces X'(A p) by (A X)'p, i.e. presupposes A = A'.  Symmetry is checked
    when the matrix is created; an unsymmetric CSR keeps the two-pass operator P_X(A V - V S) with A as given --
    the operator the oracle (and the reference's user callable) would apply."""
    n, p = 700, 3
    rowptr, col, val = _random_csr(n, seed=42)            # unsymmetric pattern and values
    Xb = wl.random_stiefel(n, p, seed=8)
    A = ctx.csr(n, rowptr, col, val)
    prob = ctx.stiefel_rq(A, n, p)
    g, H = prob.model(ctx.upload(Xb))
    oprob = oracle.stiefel_rq(n, p, rowptr, col, val)
    go = oracle.eval_grad(oprob, Xb.ravel())
    assert rel_err(g.numpy(), go) < 1e-12
    ctx.ktime_enable("stiefel_hess_fused", True)
    ctx.ktime_enable("stiefel_finish_dots", True)
    ctx.ktime_reset()
    r = ctx.stpcg(g, H, Delta=0.5, max_iterations=6, kappa_fgr=1e-8, theta=1.0)
    assert ctx.ktime_read("stiefel_hess_fused")[0] == 0 and ctx.ktime_read("stiefel_finish_dots")[0] > 0
    ctx.ktime_enable("stiefel_hess_fused", False)
    ctx.ktime_enable("stiefel_finish_dots", False)
    o = oracle.stpcg_problem(oprob, Xb.ravel(), go, 0.5, max_iterations=6, kappa_fgr=1e-8, theta=1.0)
    assert (r["iterations"], r["exit_reason"]) == (o["iterations"], o["exit_reason"])
    assert rel_err(r["s"].numpy(), o["s"]) < 1e-10
    oracle.free(oprob)
    # the symmetrised matrix takes the one-pass kernel again
    import scipy.sparse as sps
    M = sps.csr_matrix((val, col, rowptr), shape=(n, n))
    S = sps.csr_matrix(M + M.T)
    S.sort_indices()
    A2 = ctx.csr(n, S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data)
    prob2 = ctx.stiefel_rq(A2, n, p)
    g2, H2 = prob2.model(ctx.upload(Xb))
    ctx.ktime_enable("stiefel_hess_fused", True)
    ctx.ktime_reset()
    ctx.stpcg(g2, H2, Delta=0.5, max_iterations=3, kappa_fgr=1e-8, theta=1.0)
    assert ctx.ktime_read("stiefel_hess_fused")[0] > 0
    ctx.ktime_enable("stiefel_hess_fused", False)


def test_trial_entry_points_refuse_a_point_the_model_is_not_bound_to(ctx):
    """mi_stiefel_rq_trial / _armijo_trial / mi_so3n_trial work at the point of the LAST model() call, identified by
    handle and serial number: another vector -- also one that recycled the old handle -- is an error, not a silent
    evaluation of the old model."""
    from optimization_amd import capi
    nx, ny, nz, p = 8, 7, 6, 2
    n = nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    A = ctx.csr(n, rowptr, col, val)
    prob = ctx.stiefel_rq(A, n, p)
    X = ctx.upload(wl.random_stiefel(n, p, seed=1))
    g, H = prob.model(X)
    h = ctx.stiefel_project(n, p, X, ctx.upload(np.random.default_rng(2).normal(size=(n, p)) * 1e-2))
    other = ctx.upload(wl.random_stiefel(n, p, seed=2))
    with pytest.raises(capi.MiError):
        prob.trial(other, h, g)
    # deferred result + a trace request: the trace needs the wait, the call stays synchronous and still correct
    r0 = ctx.stpcg(g, H, Delta=1.0, max_iterations=5, kappa_fgr=1e-6, theta=1.0, trace_cap=8)
    r1 = ctx.stpcg(g, H, Delta=1.0, max_iterations=5, kappa_fgr=1e-6, theta=1.0, trace_cap=8, defer=True)
    assert r1["iterations"] == r0["iterations"] and np.array_equal(r0["trace"]["alpha"], r1["trace"]["alpha"])
    with pytest.raises(capi.MiError):
        ctx.stpcg_collect()  # nothing pending


def test_symmetry_check_edge_cases(ctx, oracle):
    """csr_is_symmetric at matrix creation: empty rows, an explicit zero mirrored by a missing entry (NOT symmetric
    as stored: the entry lists differ), duplicates stored in different orders (symmetric), -0.0 vs 0.0 (different
    bits: not symmetric).  Observable through which Hessian form STPCG takes."""
    def one_pass(rowptr, col, val, n):
        A = ctx.csr(n, np.array(rowptr, np.int32), np.array(col, np.int32), np.array(val, np.float64))
        prob = ctx.stiefel_rq(A, n, 1)
        x = np.zeros((n, 1)); x[0] = 1.0
        g, H = prob.model(ctx.upload(x))
        ctx.ktime_enable("stiefel_hess_fused", True)
        ctx.ktime_reset()
        ctx.stpcg(g, H, Delta=1.0, max_iterations=2, kappa_fgr=1e-3, theta=1.0)
        k = ctx.ktime_read("stiefel_hess_fused")[0]
        ctx.ktime_enable("stiefel_hess_fused", False)
        return k > 0
    # 4 x 4: row 2 empty; symmetric
    assert one_pass([0, 2, 4, 4, 5], [0, 1, 0, 1, 3], [2.0, -1.0, -1.0, 2.0, 1.0], 4)
    # (0,1) = 0.0 stored, (1,0) not stored
    assert not one_pass([0, 2, 3, 4], [0, 1, 1, 2], [2.0, 0.0, 2.0, 1.0], 3)
    # -0.0 against +0.0
    assert not one_pass([0, 2, 4, 5], [0, 1, 0, 1, 2], [2.0, 0.0, -0.0, 2.0, 1.0], 3)
    # value asymmetry in the last bit
    assert not one_pass([0, 2, 4], [0, 1, 0, 1], [2.0, -1.0, np.nextafter(-1.0, 0), 2.0], 2)


# ----------------------------------------------------------------------------------------------
# Stiefel(n, p) for p = 5 ... 8 (r05; VERDICT r04: "a Stiefel object that stops at p = 4 is a benchmark object, not a
# manifold"): every kernel family of the problem object against the CPU oracle, as for p <= 4
# ----------------------------------------------------------------------------------------------
WIDE_P = [5, 6, 7, 8]


@pytest.mark.parametrize("p", WIDE_P)
def test_wide_rows_manifold_operations_vs_oracle(ctx, oracle, p):
    """Gram, tangent projection, polar retraction, objective, gradient and (two-pass) Hessian of the Rayleigh-quotient
    problem for rows of 5 ... 8 doubles against the oracle, plus the properties the reference's own TNT test checks on
    its sphere (tests/TNT_unit_test.cpp:73-117): tangency, orthonormality after the retraction, a self-adjoint Hessian,
    and a finite-difference check of gradient and Hessian."""
    nx, ny, nz = 13, 11, 9
    n = nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    X0 = wl.random_stiefel(n, p, seed=3 + p)
    A = ctx.csr(n, rowptr, col, val)
    prob = ctx.stiefel_rq(A, n, p)
    oprob = oracle.stiefel_rq(n, p, rowptr, col, val)
    rng = np.random.default_rng(p)
    Z = rng.normal(size=(n, p))
    X, Zd = ctx.upload(X0), ctx.upload(Z)
    assert np.allclose(ctx.stiefel_gram(n, p, X, Zd), X0.T @ Z, rtol=1e-12, atol=1e-13)
    Pz = ctx.stiefel_project(n, p, X, Zd).numpy().reshape(n, p)
    M = X0.T @ Z
    assert np.allclose(Pz, Z - X0 @ (0.5 * (M + M.T)), atol=1e-13)
    Sk = X0.T @ Pz
    assert np.abs(Sk + Sk.T).max() < 1e-13
    V = 0.3 * Pz
    Y = ctx.stiefel_retract(n, p, X, ctx.upload(V)).numpy().reshape(n, p)
    Yo = oracle.eval_retract(oprob, X0.ravel(), V.ravel()).reshape(n, p)
    assert rel_err(Y, Yo) < 1e-13 and np.abs(Y.T @ Y - np.eye(p)).max() < 1e-13
    f, fo = prob.objective(X), oracle.eval_f(oprob, X0.ravel())
    assert abs(f - fo) <= 1e-13 * abs(fo)
    g, H = prob.model(X)
    go = oracle.eval_grad(oprob, X0.ravel())
    assert rel_err(g.numpy(), go) < 1e-13
    Vt = ctx.stiefel_project(n, p, X, ctx.upload(rng.normal(size=(n, p))))
    Hv = H.apply(Vt)
    assert rel_err(Hv.numpy(), oracle.eval_hess(oprob, X0.ravel(), Vt.numpy())) < 1e-13
    U = ctx.stiefel_project(n, p, X, ctx.upload(rng.normal(size=(n, p))))
    a, b = U.dot(Hv), Vt.dot(H.apply(U))
    assert abs(a - b) <= 1e-12 * max(abs(a), abs(b))
    # finite differences along the retraction: f(R(tV)) = f + t <g,V> + t^2/2 <V,HV> + O(t^3)
    vn = np.linalg.norm(Vt.numpy())
    Vu = Vt.numpy() / vn                                             # unit tangent direction
    t = 1e-4
    fp = prob.objective(ctx.stiefel_retract(n, p, X, ctx.upload(t * Vu)))
    fm = prob.objective(ctx.stiefel_retract(n, p, X, ctx.upload(-t * Vu)))
    gV, VHV = g.dot(Vt) / vn, Vt.dot(Hv) / vn ** 2
    assert abs((fp - fm) / (2 * t) - gV) <= 1e-6 * max(1.0, abs(gV))
    assert abs((fp - 2 * f + fm) / (t * t) - VHV) <= 1e-4 * max(1.0, abs(VHV))
    oracle.free(oprob)


@pytest.mark.parametrize("fmt", ["packed", "plain"])
@pytest.mark.parametrize("p", WIDE_P)
def test_wide_rows_one_pass_hessian_matches_two_pass_and_oracle(oracle, monkeypatch, p, fmt):
    """STPCG on rows of 5 ... 8 doubles: the one-pass Hessian in its recurrence form (k_st_hess_wide: 256-thread
    workgroups, S and M in LDS, 3 + p (p + 1) / 2 = 18 ... 39 partial components re-reduced in k_cg_update's prologue)
    against the two-pass operator (MI355OPT_NO_DIRGRAM=1) and the oracle: counts, exit, alpha / beta traces, the step
    to 1e-10 -- on the value-indexed matrix and on plain 12-byte entries.  MI355OPT_DIRGRAM_DIRECT (the direction
    kernel forming the Gram rows) is defined for p <= 4 only: at these widths it keeps the two passes."""
    from optimization_amd import capi
    nx, ny, nz = 19, 14, 11          # 2926 rows: 46 slices, ragged last one
    n = nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    Xb, _ = wl.stiefel_bench_iterate(nx, ny, nz, p, eps=1e-2, seed=11 + p)
    oprob = oracle.stiefel_rq(n, p, rowptr, col, val)
    go = oracle.eval_grad(oprob, Xb.ravel())
    o = oracle.stpcg_problem(oprob, Xb.ravel(), go, 1e3, max_iterations=40, kappa_fgr=1e-8, theta=1.0, trace_cap=64)
    ob = oracle.stpcg_problem(oprob, Xb.ravel(), go, 1e-3, max_iterations=40, kappa_fgr=1e-8, theta=1.0)  # boundary exit
    oracle.free(oprob)
    monkeypatch.setenv("MI355OPT_NO_PACKED", "1" if fmt == "plain" else "0")
    res = {}
    # (the one-pass form in both of its layouts at every width: one lane per row, and a quad of lanes per row --
    # MI355OPT_WIDE_QUAD; the default picks by width)
    # r06: and in WINDOW form (k_st_hess_widewin: near rows from an LDS ring, the default for p <= 7 on matrices that have a
    # window -- this grid's does)
    modes = {"recurrence": ("0", "0", "-1", "-1"), "recurrence-quad": ("0", "0", "1", "0"),
             "recurrence-lanes": ("0", "0", "0", "0"), "recurrence-window": ("0", "0", "-1", "1"),
             "direct": ("0", "1", "-1", "-1"), "two-pass": ("1", "0", "-1", "-1")}
    for mode, (no_dirgram, direct, quad, window) in modes.items():
        monkeypatch.setenv("MI355OPT_NO_DIRGRAM", no_dirgram)
        monkeypatch.setenv("MI355OPT_DIRGRAM_DIRECT", direct)
        monkeypatch.setenv("MI355OPT_WIDE_QUAD", quad)
        monkeypatch.setenv("MI355OPT_WIDE_WINDOW", window)
        c = capi.Context(0)
        try:
            A = c.csr(n, rowptr, col, val)
            prob = c.stiefel_rq(A, n, p)
            g, H = prob.model(c.upload(Xb))
            names = ("stiefel_hess_fused", "stiefel_finish_dots")
            for k in names:
                c.ktime_enable(k, True)
            c.ktime_reset()
            r = c.stpcg(g, H, Delta=1e3, max_iterations=40, kappa_fgr=1e-8, theta=1.0, trace_cap=64)
            launches = {k: c.ktime_read(k)[0] for k in names}
            rb = c.stpcg(g, H, Delta=1e-3, max_iterations=40, kappa_fgr=1e-8, theta=1.0)
            res[mode] = dict(r, s=r["s"].numpy().copy(), launches=launches, b=dict(rb, s=rb["s"].numpy().copy()))
        finally:
            c.close()
    for mode in ("recurrence", "recurrence-quad", "recurrence-lanes", "recurrence-window"):
        assert res[mode]["launches"] == {"stiefel_hess_fused": res[mode]["hvp_calls"], "stiefel_finish_dots": 0}, mode
    for mode in ("direct", "two-pass"):
        assert res[mode]["launches"]["stiefel_hess_fused"] == 0 and res[mode]["launches"]["stiefel_finish_dots"] > 0
    for mode, r in res.items():
        assert r["iterations"] == o["iterations"] and r["exit_reason"] == o["exit_reason"], mode
        assert np.allclose(r["trace"]["alpha"], o["trace"]["alpha"], rtol=1e-9), mode
        assert np.allclose(r["trace"]["beta"], o["trace"]["beta"], rtol=1e-8), mode
        assert rel_err(r["s"], o["s"]) < 1e-10, mode
        assert (r["b"]["iterations"], r["b"]["exit_reason"]) == (ob["iterations"], ob["exit_reason"]), mode
        assert rel_err(r["b"]["s"], ob["s"]) < 1e-10 and abs(r["b"]["M_norm"] - ob["M_norm"]) <= 1e-12 * ob["M_norm"], mode
    assert rel_err(res["recurrence"]["s"], res["two-pass"]["s"]) < 1e-11
    assert rel_err(res["recurrence-quad"]["s"], res["recurrence-lanes"]["s"]) < 1e-11
    assert rel_err(res["recurrence-window"]["s"], res["recurrence-lanes"]["s"]) < 1e-11
    if fmt == "plain":   # (no packed copy: no window; the window switch then leaves the default choice of the other two)
        assert any(np.array_equal(res["recurrence-window"]["s"], res[m]["s"]) for m in ("recurrence-quad", "recurrence-lanes"))
    # (the default is one of the three, bit for bit)
    assert any(np.array_equal(res["recurrence"]["s"], res[m]["s"])
               for m in ("recurrence-quad", "recurrence-lanes", "recurrence-window"))


@pytest.mark.parametrize("p", [6, 8])
def test_wide_rows_preconditioned_and_sharded_slot_forms(oracle, monkeypatch, p):
    """The other routes a wide-row solve can take: the problem's preconditioner (two-pass operator, flat direction kernel)
    and the multi-GPU code path on one rank (MI355OPT_FORCE_SLOT_PATH: 39 components through the generic-width
    reduce-to-slots kernel and k_cg_update<., FROM_SLOTS>), each against the oracle / the plain path."""
    from optimization_amd import capi
    nx, ny, nz = 12, 11, 10
    n = nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    Xb, _ = wl.stiefel_bench_iterate(nx, ny, nz, p, eps=1e-2, seed=5)
    kw = dict(Delta=1e3, max_iterations=30, kappa_fgr=1e-8, theta=1.0, trace_cap=64)
    out = {}
    for mode in ("plain", "slots"):
        monkeypatch.setenv("MI355OPT_FORCE_SLOT_PATH", "1" if mode == "slots" else "0")
        c = capi.Context(0)
        try:
            A = c.csr(n, rowptr, col, val)
            prob = c.stiefel_rq(A, n, p)
            g, H = prob.model(c.upload(Xb))
            r = c.stpcg(g, H, **kw)
            out[mode] = dict(r, s=r["s"].numpy().copy())
            if mode == "plain":
                # the problem's own tangent-space preconditioner P_X(D^-1 r) (mi_stiefel_rq_precon: external
                # preconditioner, two-pass operator, flat direction kernel at these widths) against the oracle's
                # preconditioned solve of the same problem, as test_rq_projected_jacobi_precon_vs_oracle does for p = 3
                diag = np.array([val[rowptr[i]:rowptr[i + 1]][col[rowptr[i]:rowptr[i + 1]] == i][0] for i in range(n)])
                dinv = 1.0 / (diag * np.linspace(0.5, 2.0, n))
                oprob = oracle.stiefel_rq(n, p, rowptr, col, val, dinv=dinv)
                go = oracle.eval_grad(oprob, Xb.ravel())
                Xd = c.upload(Xb)
                g, H = prob.model(Xd)
                P = prob.precon(Xd, c.upload(dinv))
                assert rel_err(P.apply(g).numpy(), oracle.eval_precon(oprob, Xb.ravel(), go)) < 1e-12
                rp = c.stpcg(g, H, P, Delta=1e3, max_iterations=40, kappa_fgr=1e-4, trace_cap=64)
                op = oracle.stpcg_problem(oprob, Xb.ravel(), go, 1e3, max_iterations=40, kappa_fgr=1e-4, trace_cap=64)
                oracle.free(oprob)
                assert rp["iterations"] == op["iterations"] and rp["exit_reason"] == op["exit_reason"]
                assert np.allclose(rp["trace"]["alpha"], op["trace"]["alpha"], rtol=1e-9)
                assert rel_err(rp["s"].numpy(), op["s"]) < 1e-9
        finally:
            c.close()
    a, b = out["plain"], out["slots"]
    assert (a["iterations"], a["exit_reason"]) == (b["iterations"], b["exit_reason"])
    assert np.array_equal(a["trace"]["alpha"], b["trace"]["alpha"]) and np.array_equal(a["s"], b["s"])


@pytest.mark.parametrize("Delta,kappa,maxit", [(1e3, 1e-8, 40), (1e-3, 1e-8, 40), (1e3, 1e-12, 7), (0.05, .1, 50)])
def test_two_kernel_step_takes_every_exit_of_stpcg(oracle, Delta, kappa, maxit):
    """The opt-in two-kernel step (MI355OPT_TWO_KERNEL_STEP; stpcg.hip k_cg_step2) on a problem whose matrix takes the
    window form with computed far columns (a 3-D stencil): residual, boundary and iteration-limit exits inside the
    merged kernel -- same count and exit as the default three-kernel step and the oracle, |s|_M and the step to the
    accuracy a changed rounding of <r+,r+> leaves (asserted loosely; the measured distance is the experiment's result,
    tests/test_gpu_cfg2_full.py)."""
    from optimization_amd import capi
    import os
    if os.environ.get("MI355OPT_FORCE_SLOT_PATH") == "1" or os.environ.get("MI355OPT_FORCE_UNIFORM_GRID") == "1":
        pytest.skip("the two-kernel step exists on one rank only (the multi-rank forms keep the three kernels)")
    nx, ny, nz, p = 24, 22, 20, 3
    n = nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    Xb, _ = wl.stiefel_bench_iterate(nx, ny, nz, p, eps=1e-2, seed=4)
    oprob = oracle.stiefel_rq(n, p, rowptr, col, val)
    go = oracle.eval_grad(oprob, Xb.ravel())
    o = oracle.stpcg_problem(oprob, Xb.ravel(), go, Delta, max_iterations=maxit, kappa_fgr=kappa, theta=1.0)
    oracle.free(oprob)
    c = capi.Context(0)
    try:
        A = c.csr(n, rowptr, col, val)
        assert A.window_info()[0] > 0 and A.window_info()[2] == nx * ny      # window form, pure far structure
        prob = c.stiefel_rq(A, n, p)
        g, H = prob.model(c.upload(Xb))
        res = {}
        for mode in (0, 1):
            c.set_option("TWO_KERNEL_STEP", mode)
            c.ktime_enable("cg_pupdate", True)
            c.ktime_reset()
            r = c.stpcg(g, H, Delta=Delta, max_iterations=maxit, kappa_fgr=kappa, theta=1.0)
            res[mode] = dict(r, s=r["s"].numpy().copy(), pupdates=c.ktime_read("cg_pupdate")[0])
    finally:
        c.close()
    assert res[1]["pupdates"] == 0 and (res[0]["pupdates"] > 0 or res[0]["iterations"] == 0)
    for mode in (0, 1):
        assert (res[mode]["iterations"], res[mode]["exit_reason"]) == (o["iterations"], o["exit_reason"]), mode
        assert abs(res[mode]["M_norm"] - o["M_norm"]) <= 1e-9 * o["M_norm"]
    assert rel_err(res[0]["s"], o["s"]) < 1e-10 and rel_err(res[1]["s"], o["s"]) < 1e-7
