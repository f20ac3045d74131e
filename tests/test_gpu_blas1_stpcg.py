"""GPU parity tests through the C ABI: Vector-concept kernels and the fused device STPCG against
the CPU oracle (oracle/liboracle.so), the golden fixtures produced by the real reference, and the
property checks of the reference's own unit tests (tests/IterativeSolvers_unit_test.cpp)."""
import numpy as np
import pytest

from conftest import rel_err
from optimization_amd import capi

pytestmark = pytest.mark.gpu

DBL_MAX = float(np.finfo(np.float64).max)


# ----------------------------------------------------------------------------------------------
# Vector concept
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 1000, 1001, 262144 + 7, 3_000_001])
def test_axpby_scale_fill_dot(ctx, n):
    rng = np.random.default_rng(n)
    x, y = rng.normal(size=n), rng.normal(size=n)
    X, Y = ctx.upload(x), ctx.upload(y)
    Z = ctx.vec(n)
    Z.axpby(0.75, X, -1.25, Y)
    assert np.array_equal(Z.numpy(), 0.75 * x + (-1.25) * y) or rel_err(Z.numpy(), 0.75 * x - 1.25 * y) < 1e-15
    Z2 = Y.copy().axpy(2.5, X)
    assert rel_err(Z2.numpy(), y + 2.5 * x) < 1e-15
    assert np.array_equal(X.copy().scale(-1.0).numpy(), -x)
    assert np.array_equal(ctx.vec(n).fill(3.25).numpy(), np.full(n, 3.25))
    d = X.dot(Y)
    exact = float(np.dot(x.astype(np.longdouble), y.astype(np.longdouble)))
    scale = float(np.dot(np.abs(x), np.abs(y)))
    assert abs(d - exact) <= 1e-14 * scale
    # in-place aliasing forms used by the reference expressions s = s + a p ; p = -v + b p
    S = X.copy()
    S.axpby(1.0, S, 0.5, Y)
    assert rel_err(S.numpy(), x + 0.5 * y) < 1e-15
    Pv = X.copy()
    Pv.axpby(-1.0, Y, 0.3, Pv)
    assert rel_err(Pv.numpy(), -y + 0.3 * x) < 1e-15


def test_dot_batch_deterministic(ctx):
    n = 1_000_003
    rng = np.random.default_rng(1)
    a, b, c = (ctx.upload(rng.normal(size=n)) for _ in range(3))
    r1 = ctx.dot_batch([a, b, a, c], [b, b, a, a])
    r2 = ctx.dot_batch([a, b, a, c], [b, b, a, a])
    assert np.array_equal(r1, r2)  # fixed-shape reduction: bitwise reproducible
    an, bn, cn = a.numpy(), b.numpy(), c.numpy()
    ref = np.array([an @ bn, bn @ bn, an @ an, cn @ an])
    assert np.allclose(r1, ref, rtol=1e-12, atol=1e-9)


def test_zero_times_g_propagates_nan(ctx):
    # `Vector s_k = 0 * g;` IterativeSolvers.h:211 -- 0*Inf = NaN must propagate like on the host
    g = ctx.upload(np.array([1.0, np.inf, -2.0]))
    z = ctx.vec(3).axpby(0.0, g, 0.0, g)
    out = z.numpy()
    assert out[0] == 0 and np.isnan(out[1]) and out[2] == 0


def test_errors_are_loud(ctx):
    from optimization_amd import capi
    a, b = ctx.vec(4), ctx.vec(5)
    with pytest.raises(capi.MiError):
        a.dot(b)
    with pytest.raises(capi.MiError):
        a.axpy(1.0, b)


# ----------------------------------------------------------------------------------------------
# fused STPCG: golden small cases of the reference's own tests
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["ExactSTPCG", "ExactSTPCGwithNegativeCurvature",
                                  "ExactSTPCGwithPreconditioning",
                                  "ExactSTPCGwithNegativeCurvatureAndPreconditioning"])
def test_stpcg_small_golden(ctx, golden, case):
    c = golden("stpcg_small.json")[case]
    g = ctx.upload(c["g"])
    H = ctx.op_diag(ctx.upload(c["H_diag"]))
    P = ctx.precon_diag(ctx.upload(1.0 / np.array(c["M_diag"]))) if c["M_diag"] else None
    r = ctx.stpcg(g, H, P, Delta=c["Delta"], max_iterations=c["max_iterations"],
                  kappa_fgr=c["kappa_fgr"], theta=c["theta"], trace_cap=8)
    assert r["iterations"] == c["iterations"]
    assert abs(r["M_norm"] - c["M_norm"]) <= 1e-13 * abs(c["M_norm"])
    s = r["s"].numpy()
    assert np.allclose(s, c["s"], rtol=1e-12, atol=1e-15)
    assert np.allclose(r["trace"]["alpha"], c["alpha"][:len(r["trace"]["alpha"])], rtol=1e-12)
    # the assertions of tests/IterativeSolvers_unit_test.cpp:149-158,176-185,205-215,236-250
    if "NegativeCurvature" not in case:
        s_gt = -np.array(c["g"]) / np.array(c["H_diag"])
        assert np.linalg.norm(s - s_gt) < 1e-6
    Mdiag = np.array(c["M_diag"]) if c["M_diag"] else np.ones(3)
    s_M = np.sqrt(s @ (Mdiag * s))
    assert abs(r["M_norm"] - s_M) / s_M < 1e-6


def _diag_problem(n, seed, lo=1000.0, hi=3000.0):
    rng = np.random.default_rng(seed)
    g = rng.uniform(-1, 1, size=n)
    D = rng.uniform(lo, hi, size=n)
    M = rng.uniform(lo, hi, size=n)
    return g, D, M


@pytest.mark.parametrize("n", [1000, 4097, 100_000])
@pytest.mark.parametrize("precon", ["none", "diag", "callback"])
def test_stpcg_truncated_vs_oracle(ctx, oracle, n, precon):
    """tests/IterativeSolvers_unit_test.cpp:254-310 (STPCGwithTruncation / ...PreconditioningAndTruncation)
    with a seeded RNG, plus iteration-level parity with the oracle."""
    g, D, M = _diag_problem(n, seed=n)
    kappa, theta, Delta = .1, .7, 1000.0
    G, Dv = ctx.upload(g), ctx.upload(D)
    H = ctx.op_diag(Dv)
    P = None
    if precon == "diag":
        P = ctx.precon_diag(ctx.upload(1.0 / M))
    elif precon == "callback":
        Minv = ctx.upload(1.0 / M)
        dop = ctx.op_diag(Minv)
        P = ctx.precon_callback(n, lambda r, v: dop.apply(r, v))
    r = ctx.stpcg(G, H, P, Delta=Delta, max_iterations=n, kappa_fgr=kappa, theta=theta, trace_cap=64)
    o = oracle.stpcg(g, lambda v: D * v, P=(lambda v: v / M) if precon != "none" else None,
                     inner=lambda a, b: float(a @ b), Delta=Delta, max_iterations=n, kappa_fgr=kappa,
                     theta=theta, trace_cap=64)
    assert r["iterations"] == o["iterations"]
    assert r["exit_reason"] == o["exit_reason"]
    for k in ("alpha", "beta", "kappa", "rv"):
        assert np.allclose(r["trace"][k], o["trace"][k], rtol=1e-11), k
    s = r["s"].numpy()
    assert rel_err(s, o["s"]) < 1e-11
    assert abs(r["M_norm"] - o["M_norm"]) <= 1e-11 * o["M_norm"]
    # reference's property assertions
    Mi = 1.0 / M if precon != "none" else np.ones(n)
    res = g + D * s
    rel = np.sqrt(res @ (Mi * res)) / np.sqrt(g @ (Mi * g))
    assert rel < kappa
    Md = M if precon != "none" else np.ones(n)
    s_M = np.sqrt(s @ (Md * s))
    assert abs(r["M_norm"] - s_M) / s_M < 1e-6


def test_stpcg_block3_precon_vs_oracle(ctx, oracle):
    nb = 5000
    n = 3 * nb
    rng = np.random.default_rng(3)
    g = rng.normal(size=n)
    D = rng.uniform(1.0, 50.0, size=n)
    # SPD 3x3 blocks B_i; preconditioner = inverse blocks
    A = rng.normal(size=(nb, 3, 3))
    B = A @ np.transpose(A, (0, 2, 1)) + 3 * np.eye(3)[None]
    Binv = np.linalg.inv(B)
    P = ctx.precon_block3(ctx.upload(Binv.reshape(-1)))
    H = ctx.op_diag(ctx.upload(D))
    r = ctx.stpcg(ctx.upload(g), H, P, Delta=1e6, max_iterations=200, kappa_fgr=1e-6, theta=.5,
                  trace_cap=256)
    o = oracle.stpcg(g, lambda v: D * v, P=lambda v: np.einsum("bij,bj->bi", Binv, v.reshape(nb, 3)).ravel(),
                     inner=lambda a, b: float(a @ b), Delta=1e6, max_iterations=200, kappa_fgr=1e-6,
                     theta=.5, trace_cap=256)
    assert r["iterations"] == o["iterations"] and r["exit_reason"] == o["exit_reason"]
    assert np.allclose(r["trace"]["alpha"], o["trace"]["alpha"], rtol=1e-10)
    assert rel_err(r["s"].numpy(), o["s"]) < 1e-10


def test_stpcg_boundary_and_negative_curvature(ctx, oracle):
    n = 20_001
    rng = np.random.default_rng(11)
    g = rng.normal(size=n)
    D = rng.uniform(0.5, 4.0, size=n)
    G = ctx.upload(g)
    # (a) step leaves the trust region (:347, skplus1 > Delta^2)
    r = ctx.stpcg(G, ctx.op_diag(ctx.upload(D)), Delta=0.37 * np.linalg.norm(g / D), max_iterations=100,
                  kappa_fgr=1e-10, theta=1.0)
    o = oracle.stpcg(g, lambda v: D * v, inner=lambda a, b: float(a @ b),
                     Delta=0.37 * np.linalg.norm(g / D), max_iterations=100, kappa_fgr=1e-10, theta=1.0)
    assert o["exit_reason"] == 3 and r["exit_reason"] == 3
    assert r["iterations"] == o["iterations"]
    assert r["M_norm"] == o["M_norm"]
    assert rel_err(r["s"].numpy(), o["s"]) < 1e-11
    assert abs(np.linalg.norm(r["s"].numpy()) - r["M_norm"]) < 1e-9 * r["M_norm"]
    # (b) indefinite Hessian: negative curvature after a few iterations
    D2 = D.copy()
    D2[::7] = -D2[::7]
    r = ctx.stpcg(G, ctx.op_diag(ctx.upload(D2)), Delta=50.0, max_iterations=100, kappa_fgr=1e-10, theta=1.0)
    o = oracle.stpcg(g, lambda v: D2 * v, inner=lambda a, b: float(a @ b), Delta=50.0, max_iterations=100,
                     kappa_fgr=1e-10, theta=1.0)
    assert r["exit_reason"] == 3 and o["exit_reason"] == 3
    assert r["iterations"] == o["iterations"]
    assert rel_err(r["s"].numpy(), o["s"]) < 1e-10


def test_stpcg_kernel_direction(ctx, oracle):
    """p in ker H (:305-337) incl. the sign flip (:320-326)."""
    n = 1001
    rng = np.random.default_rng(5)
    g = rng.normal(size=n)
    for D in (np.zeros(n), np.full(n, 1e-12)):
        r = ctx.stpcg(ctx.upload(g), ctx.op_diag(ctx.upload(D)), Delta=7.0, max_iterations=10)
        o = oracle.stpcg(g, lambda v: D * v, inner=lambda a, b: float(a @ b), Delta=7.0, max_iterations=10)
        assert o["exit_reason"] == 2 and r["exit_reason"] == 2
        assert r["iterations"] == o["iterations"] == 0
        assert r["M_norm"] == 7.0
        assert rel_err(r["s"].numpy(), o["s"]) < 1e-12


def test_stpcg_edge_cases(ctx, oracle):
    from optimization_amd import capi
    n = 100
    g = np.linspace(-1, 1, n)
    G, H = ctx.upload(g), ctx.op_diag(ctx.upload(np.full(n, 2.0)))
    # max_iterations = 0 -> s = 0, M_norm = 0, zero iterations
    r = ctx.stpcg(G, H, Delta=1.0, max_iterations=0)
    assert r["iterations"] == 0 and r["M_norm"] == 0 and not r["s"].numpy().any()
    # g = 0 -> immediate residual exit
    r = ctx.stpcg(ctx.upload(np.zeros(n)), H, Delta=1.0)
    assert r["iterations"] == 0 and r["exit_reason"] == 0 and not r["s"].numpy().any()
    # exact one-step solve (H = 2 I): one iteration, s = -g/2
    r = ctx.stpcg(G, H, Delta=1e9, kappa_fgr=1e-12, theta=1.0)
    assert r["iterations"] == 1 and rel_err(r["s"].numpy(), -g / 2) < 1e-15
    # argument checks of IterativeSolvers.h:183-205
    for kw in (dict(Delta=0.0), dict(Delta=-1.0), dict(kappa_fgr=1.0), dict(kappa_fgr=-.1),
               dict(theta=1.5), dict(theta=-.1), dict(epsilon=0.0), dict(epsilon=1.0)):
        args = dict(Delta=1.0)
        args.update(kw)
        with pytest.raises(capi.MiError) as ei:
            ctx.stpcg(G, H, **args)
        assert ei.value.status == 1
        assert oracle.stpcg(g, lambda v: 2 * v, **args)["rc"] == -1


def test_stpcg_run_ahead_invariance(ctx):
    """The speculative-enqueue depth must not change any result bit."""
    n = 50_000
    g, D, M = _diag_problem(n, seed=9, lo=1.0, hi=400.0)
    G, H = ctx.upload(g), ctx.op_diag(ctx.upload(D))
    base = None
    for ra in (1, 2, 3, 8, 64):
        r = ctx.stpcg(G, H, Delta=1e9, max_iterations=80, kappa_fgr=1e-9, theta=1.0, run_ahead=ra)
        cur = (r["iterations"], r["M_norm"], r["s"].numpy())
        if base is None:
            base = cur
        else:
            assert cur[0] == base[0] and cur[1] == base[1] and np.array_equal(cur[2], base[2])


@pytest.mark.parametrize("n", [3_000_000, 2_999_999, 1_048_576 + 77, 40_000])
def test_early_s_direction_kernel_is_the_default_one_bit_for_bit(n):
    """MI355OPT_EARLY_S=1 (opt-in, stpcg.hip k_cg_pupdate_early): s += alpha p ahead of the direction kernel's own
    reduction, every operand requested at once -- the same walk and expressions as k_cg_pupdate, so every bit of the step,
    the traces, the count and the exit equal the default's; sizes: cfg2's (two grid-stride steps + the ragged one), an odd
    length, one just past a whole step, and one small enough for a single step."""
    from optimization_amd import capi
    g, D, M = _diag_problem(n, seed=5, lo=1.0, hi=900.0)
    out = []
    for early in (0, 1):
        c = capi.Context(0)
        try:
            c.set_option("EARLY_S", early)
            G, H = c.upload(g), c.op_diag(c.upload(D))
            c.ktime_enable("cg_pupdate", True)
            r = c.stpcg(G, H, Delta=30.0, max_iterations=60, kappa_fgr=1e-9, theta=1.0, trace_cap=64)
            assert c.ktime_read("cg_pupdate")[0] > 0
            out.append((r["iterations"], r["exit_reason"], r["M_norm"], r["s"].numpy().copy(),
                        np.array(r["trace"]["alpha"]), np.array(r["trace"]["beta"])))
        finally:
            c.close()
    a, b = out
    assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2]
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])


def test_polled_read_back_equals_stream_synchronize():
    """r05: a scalar read-back is a one-wave kernel that stores the slots and then a sequence number into coherent pinned
    words the host polls (`stream_wait` / `read_slots_sync`, context.hip / blas1.hip: 14.4 instead of 18.8 us per dot
    product and read, tools/sync_latency.py); MI355OPT_NO_POLLED_SYNC=1 is the copy + hipStreamSynchronize it replaces.
    Same values, same number of counted synchronisations, and a long queue behind the poll (its 2 ms bound, then the
    blocking wait) changes nothing."""
    from optimization_amd import capi
    n = 200_000
    rng = np.random.default_rng(5)
    a, b = rng.normal(size=n), rng.normal(size=n)
    g, D, M = _diag_problem(n, seed=3, lo=1.0, hi=900.0)
    long_queue = rng.normal(size=4_000_000)
    out = {}
    for tag, opt in (("polled", 0), ("sync", 1)):
        c = capi.Context(0)
        try:
            c.set_option("NO_POLLED_SYNC", opt)
            A, B = c.upload(a), c.upload(b)
            s0 = c.sync_count()
            dots = [A.dot(B), A.dot(A), B.dot(B)]
            syncs = c.sync_count() - s0
            big = c.upload(long_queue)   # ~100 queued passes: the read-back waits longer than the poll's bound
            for _ in range(100):
                big.scale(1.0000001)
            dots.append(big.dot(big))
            r = c.stpcg(c.upload(g), c.op_diag(c.upload(D)), Delta=1e9, max_iterations=60, kappa_fgr=1e-9, theta=1.0)
            out[tag] = (dots, syncs, r["iterations"], r["M_norm"], r["s"].numpy().copy())
        finally:
            c.close()
    assert out["polled"][0] == out["sync"][0] and out["polled"][1] == out["sync"][1] == 3
    assert out["polled"][2:4] == out["sync"][2:4] and np.array_equal(out["polled"][4], out["sync"][4])


def test_profiler_ranges_nest_around_a_solve(ctx):
    """mi_range_push/pop (roctx markers, SURVEY 8(b) group 9) are callable around and inside a solve; the solve
    itself opens a range named mi_stpcg (checked with rocprofv3 --marker-trace in profiles/)."""
    n = 4096
    D = np.linspace(1.0, 9.0, n)
    g = np.sin(np.arange(n, dtype=np.float64))
    ctx.range_push("outer iteration")
    r = ctx.stpcg(ctx.upload(g), ctx.op_diag(ctx.upload(D)), Delta=1e9, max_iterations=40, kappa_fgr=1e-10, theta=1.0)
    ctx.range_pop()
    assert np.abs(r["s"].numpy() + g / D).max() < 1e-8


def test_view_outlives_its_owner_across_the_c_abi(ctx):
    """a plain C client may destroy a vector before the views of it.  The owner then keeps storage and
    generation counter until its last view goes: writes through the view still land (and `touch` has a live counter),
    and the pool does not hand the same device memory to the next vector while a view still points into it."""
    import ctypes as C
    L = ctx.L
    n = 4096
    base = capi.vp()
    capi.check(L.mi_vec_create(ctx.h, n, C.byref(base)))
    v1, v2 = capi.vp(), capi.vp()
    capi.check(L.mi_vec_view(base, 1024, 1024, C.byref(v1)))
    capi.check(L.mi_vec_view(v1, 512, 256, C.byref(v2)))          # a view of a view counts on the owner
    p_base = capi.vp()
    capi.check(L.mi_vec_data(base, C.byref(p_base)))
    capi.check(L.mi_vec_destroy(base))                             # wrong order on purpose
    other = capi.Vec(ctx, n)                                       # same size class: must NOT recycle base's storage
    assert other.data_ptr() != p_base.value
    other.fill(7.0)
    a, b = capi.Vec(ctx, 0, handle=v1), capi.Vec(ctx, 0, handle=v2)
    a.owned = b.owned = False
    a.fill(1.0)
    b.fill(2.0)                                                    # writes + touch through both views
    x = a.numpy()
    assert (x[:512] == 1).all() and (x[512:768] == 2).all() and (x[768:] == 1).all()
    assert (other.numpy() == 7).all()
    capi.check(L.mi_vec_destroy(v1))
    capi.check(L.mi_vec_destroy(v2))                               # the last view releases the owner
    again = capi.Vec(ctx, n)                                       # now the storage is back in the pool
    del again, other
