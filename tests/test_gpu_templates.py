"""GPU tests of the drop-in C++ template API with Vector = MI355::DeviceVector (compiled harness
tests/cpp/harness_device.cpp), mirroring the reference's own unit tests and checked against the
golden fixtures produced by the real reference."""
import numpy as np
import pytest

from conftest import floor_or, rel_err
from optimization_amd import workloads as wl

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def harness():
    import harness_py
    return harness_py.DeviceHarness()


@pytest.mark.parametrize("mode", [0, 1], ids=["fused", "generic"])
@pytest.mark.parametrize("case", ["ExactSTPCG", "ExactSTPCGwithNegativeCurvature",
                                  "ExactSTPCGwithPreconditioning",
                                  "ExactSTPCGwithNegativeCurvatureAndPreconditioning"])
def test_stpcg_template_small_cases(harness, golden, case, mode):
    c = golden("stpcg_small.json")[case]
    Minv = 1.0 / np.array(c["M_diag"]) if c["M_diag"] else None
    r = harness.stpcg_diag(c["g"], c["H_diag"], Minv, c["Delta"], c["max_iterations"], c["kappa_fgr"],
                           c["theta"], mode)
    assert r["rc"] == 0, r["err"]
    assert r["iterations"] == c["iterations"]
    assert abs(r["M_norm"] - c["M_norm"]) <= 1e-13 * abs(c["M_norm"])
    assert np.allclose(r["s"], c["s"], rtol=1e-12, atol=1e-15)


def test_stpcg_template_fused_equals_generic_and_oracle(harness, oracle):
    n = 20_000
    rng = np.random.default_rng(4)
    g, D, M = rng.uniform(-1, 1, n), rng.uniform(1000, 3000, n), rng.uniform(1000, 3000, n)
    o = oracle.stpcg(g, lambda v: D * v, P=lambda v: v / M, inner=lambda a, b: float(a @ b), Delta=1000.0,
                     max_iterations=n, kappa_fgr=.1, theta=.7)
    for mode in (0, 1):
        r = harness.stpcg_diag(g, D, 1.0 / M, 1000.0, n, .1, .7, mode)
        assert r["rc"] == 0, r["err"]
        assert r["iterations"] == o["iterations"]
        assert rel_err(r["s"], o["s"]) < 1e-11
        assert abs(r["M_norm"] - o["M_norm"]) < 1e-11 * o["M_norm"]


def test_stpcg_template_throws_like_reference(harness):
    g, D = np.ones(8), np.ones(8)
    for kw in (dict(Delta=0.0), dict(kappa=1.0), dict(theta=1.5)):
        a = dict(Delta=1.0, kappa=.1, theta=.5)
        a.update(kw)
        for mode in (0, 1):
            r = harness.stpcg_diag(g, D, None, a["Delta"], 10, a["kappa"], a["theta"], mode)
            assert r["rc"] == -1  # std::invalid_argument


@pytest.mark.parametrize("key,pre", [("plain", False), ("precon", True)])
def test_tnt_sphere_device_golden(harness, oracle, golden, key, pre):
    """tests/TNT_unit_test.cpp:126-187 with DeviceVector and Args = {DeviceVector}."""
    g = golden("tnt_sphere.json")[key]
    p = oracle.default_params(gradient_tolerance=1e-8, relative_decrease_tolerance=0, stepsize_tolerance=0,
                              preconditioned_gradient_tolerance=0)
    r = harness.tnt_sphere(pre, g["x0"], p)
    assert r["rc"] == 0, r.get("err")
    assert r["status"] == 0  # TNTStatus::Gradient
    assert r["outer_iterations"] == g["outer_iterations"]
    assert list(r["inner_iterations"]) == g["inner_iterations"]
    assert np.allclose(r["trust_region_radius"], g["trust_region_radius"], rtol=1e-10)
    assert np.allclose(r["objective_values"][:6], g["objective_values"][:6], rtol=1e-8, atol=1e-15)
    assert np.allclose(r["gain_ratios"], g["gain_ratios"], rtol=1e-6)
    assert r["gradfx_norm"] < 1e-8
    assert r["f"] < r["objective_values"][0]
    assert np.linalg.norm(r["x"] - np.array([0, 0, 1.0])) < 1e-8


def test_stpcg_user_function_stop_on_device_vectors(harness, golden):
    """STPCG on DeviceVector with a user function that stops the solve at iteration k (reference
    IterativeSolvers.h:365-369): tagged device callables + a user function take the generic loop (the fused solver
    has no user hook); against the fixture of the REAL reference (tests/golden/stpcg_user_stop.json): same
    iteration and call counts, s WITHOUT the interrupted iteration's step, M-norm from the recurrence (:424)."""
    import oracle_py
    fx = golden("stpcg_user_stop.json")
    pr = oracle_py.stpcg_stop_problem(fx["n"], fx["seed"])
    for c in fx["cases"]:
        r = oracle_py.stpcg_diag_stop(harness.L, "hd", pr["g"], pr["D"], pr["Minv"] if c["precon"] else None,
                                      c["stop_at"])
        assert r["rc"] == 0, harness.err()
        assert (r["iterations"], r["calls"]) == (c["iterations"], c["calls"]), c["stop_at"]
        # (the never-stopped preconditioned case runs into the 100-iteration limit with the residual stagnating at
        # rounding level 30 iterations earlier: from there on CG amplifies last-bit differences of the inner products)
        tol = 1e-10 if c["iterations"] < 100 else 1e-5
        assert np.abs(r["s"] - np.array(c["s"])).max() <= tol * max(1e-300, np.abs(c["s"]).max())
        assert abs(r["M_norm"] - c["M_norm"]) <= tol * max(1e-300, c["M_norm"])


@pytest.mark.parametrize("mode", [0, 1, 2], ids=["generic_host_kkt", "fused_device_kkt", "generic_device_kkt"])
@pytest.mark.parametrize("case", ["exact", "truncated"])
def test_projected_stpcg_on_device_vectors(harness, golden, case, mode):
    """The `At` + constraint-preconditioner branch of STPCG (reference IterativeSolvers.h:229-253,381-405) with
    Vector = Multiplier = DeviceVector, on the reference's two equality-constrained cases
    (tests/IterativeSolvers_unit_test.cpp:316-496, n = 1000, 100 constraints), against the fixture the REAL
    reference produced (tests/golden/stpcg_projected.json): same iteration count, iterate to 1e-10, |A s| < 1e-6.
    fused_device_kkt: tagged constraint preconditioner + A' of one device KKT object (mi_precon_create_constraint)
    -> the whole projected solve is ONE mi_stpcg call, one host synchronisation."""
    import oracle_py
    pr = oracle_py.projected_stpcg_problem(case)
    fx = golden("stpcg_projected.json")[case]
    r = harness.stpcg_projected(pr, mode)
    assert r["rc"] == 0, r["err"]
    if mode == 1:
        assert r["syncs"] == 1, r["syncs"]
    assert r["iterations"] == fx["iterations"]
    assert rel_err(r["s"], np.array(fx["s"])) < 1e-10
    assert abs(r["M_norm"] - fx["M_norm"]) < 1e-10 * fx["M_norm"]
    assert np.linalg.norm(pr["A"] @ r["s"]) < 1e-6


def test_projected_stpcg_with_sparse_constraints_vs_the_reference(harness, golden):
    """The projected STPCG (reference IterativeSolvers.h:229-253,381-405) with SPARSE constraints (n = 2000, 150
    constraints of 9 non-zeros) through the sparse device KKT object (mi_precon_create_constraint_csr: S lambda = b by a
    CG iteration inside one workgroup) as ONE fused mi_stpcg call, against the fixture the REAL reference produced with a
    dense KKT factorisation on the same inputs (tests/golden/stpcg_projected_sparse.json, make_golden_sparse_kkt.py):
    same iteration count, iterate to 1e-10, |A s| ~ 0, one host synchronisation."""
    import hashlib
    import oracle_py
    pr = oracle_py.projected_stpcg_sparse_problem()
    fx = golden("stpcg_projected_sparse.json")
    sha = hashlib.sha256(pr["A"].tobytes() + pr["g"].tobytes() + pr["P"].tobytes() + pr["M"].tobytes()).hexdigest()
    assert sha == fx["inputs_sha256"]                      # the seeded inputs are the ones the fixture was made from
    r = harness.stpcg_projected(pr, 3)
    assert r["rc"] == 0, r["err"]
    print("sparse projected STPCG:", r["iterations"], "iterations; inner CG", r["kkt_inner"], "iterations in the last "
          "application, worst relative residual", r["kkt_worst_residual"])
    assert r["syncs"] == 1 and r["kkt_worst_residual"] <= 1e-13
    assert r["iterations"] == fx["iterations"]
    assert rel_err(r["s"], np.array(fx["s"])) < 1e-10
    assert abs(r["M_norm"] - fx["M_norm"]) < 1e-10 * fx["M_norm"]
    assert np.linalg.norm(pr["A"] @ r["s"]) < 1e-6


@pytest.mark.parametrize("key,pk", [("plain", 0), ("jacobi", 1)])
@pytest.mark.parametrize("mode", [0, 1], ids=["device-csr-hessian", "host-lambda-hessian"])
def test_tnt_rosenbrock100_device_golden(harness, oracle, golden, key, pk, mode):
    """BASELINE cfg1 on the HIP path: chained Rosenbrock n = 100 through EuclideanTNT<DeviceVector> (reference
    Riemannian/TNT.h:757-805) against the trace of the REAL reference (tests/golden/tnt_rosenbrock100.json).
    The outer loop, STPCG and every vector operation run on the GPU; the inner products are two-stage device
    reductions, i.e. re-associated with respect to the reference's sequential sums, and this problem takes
    ~370 outer iterations through a curved valley: the trace is compared while it agrees to rounding, the
    end result by its own optimality and against the reference's minimiser."""
    g = golden("tnt_rosenbrock100.json")[key]
    p = oracle.default_params(gradient_tolerance=1e-8, relative_decrease_tolerance=0, stepsize_tolerance=0,
                              preconditioned_gradient_tolerance=0, Delta_tolerance=0, max_iterations=500)
    r = harness.tnt_rosenbrock(100, pk, 0.1 * np.ones(100), p, mode)
    assert r["rc"] == 0, r.get("err")
    assert r["status"] == g["status"] == 0  # TNTStatus::Gradient
    assert r["gradfx_norm"] < 1e-8 and r["f"] < 1e-15
    assert np.linalg.norm(r["x"] - np.array(g["x"])) < 1e-8
    # the first iterations must follow the reference's trace step by step (same counts, same accept decisions,
    # f to rounding): 40 outer iterations are ~250 inner ones
    k = 40
    assert list(r["inner_iterations"][:k]) == g["inner_iterations"][:k]
    assert np.allclose(r["objective_values"][:k], g["objective_values"][:k], rtol=1e-9)
    assert np.allclose(r["trust_region_radius"][:k], g["trust_region_radius"][:k], rtol=1e-9)
    # ... and the whole run stays close to it in length
    assert abs(r["outer_iterations"] - g["outer_iterations"]) <= max(4, g["outer_iterations"] // 20), \
        (r["outer_iterations"], g["outer_iterations"])
    print("rosenbrock100", key, mode, "outer", r["outer_iterations"], "ref", g["outer_iterations"], "inner",
          int(np.sum(r["inner_iterations"])), "ref", int(np.sum(g["inner_iterations"])))


def test_gd_sphere_device(harness, golden):
    """tests/GradientDescent_unit_test.cpp:76-130 on the device: the same iteration and line-search counts as the
    real reference (tests/golden/gd_counts.json)."""
    g = golden("gd_counts.json")["sphere"]
    r = harness.gd_sphere(g["x0"])
    assert r["rc"] == 0, r["err"]
    assert r["status"] == 0
    assert abs(r["f"]) < 1e-4 and r["gradfx_norm"] < 1e-4
    assert np.linalg.norm(r["x"] - np.array([0, 0, 1.0])) < 1e-4
    assert r["iterations"] == g["iterations"]
    assert list(r["linesearch_iterations"]) == g["linesearch_iterations"]
    # f -> 0 at the optimum: absolute comparison
    assert np.allclose(r["objective_values"], g["objective_values"], rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("key", ["stiefel_p2", "stiefel_p3"])
def test_gd_stiefel_device_counts_vs_reference_fixture(harness, golden, key):
    """GradientDescent<DeviceVector> on the Stiefel Rayleigh quotient: iteration count, every line-search count and
    the objective trace of the REAL reference (tests/golden/gd_counts.json); the fused Armijo trial
    (mi_stiefel_rq_armijo_trial, one read-back per trial) gives the same bits as the statement-by-statement loop."""
    g = golden("gd_counts.json")[key]
    nx, ny, nz = g["grid"]
    p, n = g["p"], nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    X0 = wl.random_stiefel(n, p, seed=g["seed"])
    prm = g["params"]
    runs = {}
    for mode in (0, 1):
        r = harness.gd_stiefel(n, p, rowptr, col, val, X0, prm["max_iterations"], prm["gradient_tolerance"],
                               prm["alpha"], prm["beta"], prm["sigma"], prm["max_ls_iterations"], mode)
        assert r["rc"] == 0, r["err"]
        assert r["status"] == g["status"]
        assert r["iterations"] == g["iterations"]
        assert list(r["linesearch_iterations"]) == g["linesearch_iterations"], mode
        assert np.allclose(r["objective_values"], g["objective_values"], rtol=1e-12)
        assert abs(r["f"] - g["f"]) <= 1e-12 * abs(g["f"])
        assert abs(r["gradfx_norm"] - g["gradfx_norm"]) <= 1e-6 * g["gradfx_norm"]
        runs[mode] = r
    assert np.array_equal(runs[0]["x"], runs[1]["x"])
    assert np.array_equal(runs[0]["objective_values"], runs[1]["objective_values"])
    assert runs[0]["gradfx_norm"] == runs[1]["gradfx_norm"]
    trials = int(np.sum(runs[0]["linesearch_iterations"]))
    print(key, "trials", trials, "syncs fused", runs[0]["syncs"], "plain", runs[1]["syncs"])
    # one read-back per Armijo trial and nothing else; the statement-by-statement loop adds the gradient norm of
    # every iteration (and recomputes A X for the gradient, which the fused trial keeps from the objective)
    assert runs[0]["syncs"] <= trials + 4
    assert runs[1]["syncs"] >= runs[0]["syncs"] + g["iterations"] - 4


@pytest.mark.parametrize("mode", [0, 1], ids=["fused", "generic"])
def test_tnt_stiefel_device_vs_reference_fixture(harness, oracle, oracle_omp, golden, mode):
    """TNT<DeviceVector, DeviceVector> on the BASELINE cfg2 recipe (8x7x6 grid) against the trace the
    REAL reference produced on the same inputs (tests/golden/tnt_stiefel_8x7x6.json)."""
    g = golden("tnt_stiefel_8x7x6.json")
    nx, ny, nz = g["grid"]
    p, n = g["p"], nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    X0 = np.array(g["x0"]).reshape(n, p)
    assert np.array_equal(X0, wl.random_stiefel(n, p, seed=g["seed"]))
    prm = oracle.default_params(gradient_tolerance=1e-8, relative_decrease_tolerance=0, stepsize_tolerance=0,
                                preconditioned_gradient_tolerance=0, Delta_tolerance=0, max_iterations=200,
                                max_TPCG_iterations=50)
    r = harness.tnt_stiefel(n, p, rowptr, col, val, X0, prm, mode)
    assert r["rc"] == 0, r.get("err")
    assert r["status"] == g["status"]
    assert r["outer_iterations"] == g["outer_iterations"]
    assert r["accepted"] == g["accepted"]
    assert list(r["inner_iterations"]) == g["inner_iterations"]
    assert np.allclose(r["objective_values"], g["objective_values"], rtol=1e-11)
    assert np.allclose(r["trust_region_radius"], g["trust_region_radius"], rtol=1e-9)
    # gain ratio = df/dm: well conditioned while the step still changes f beyond roundoff
    fv = np.array(g["objective_values"])
    well = np.abs(fv[:-2] - fv[1:-1]) > 1e-9 * np.abs(fv[:-2])
    assert well.sum() >= len(well) - 2
    assert np.allclose(r["gain_ratios"][well], np.array(g["gain_ratios"])[well], rtol=1e-5)
    assert abs(r["f"] - g["f"]) < 1e-12
    # exact answer: f* = 1/2 (sum of the p smallest eigenvalues of A)
    lams = sorted(wl.laplacian_3d_eigvec(nx, ny, nz, a, b, c)[1] + 0.1
                  for a in (1, 2) for b in (1, 2) for c in (1, 2))[:p]
    assert abs(r["f"] - 0.5 * sum(lams)) < 1e-10
    X = r["x"].reshape(n, p)
    assert np.abs(X.T @ X - np.eye(p)).max() < 1e-12
    # the minimiser is a subspace: compare projectors.  BASELINE tolerance 1e-10 relative -- or the conditioning floor
    # of this run: the reference algorithm itself, sums re-associated (conftest.oracle_omp), ends this far from the
    # sequential-sum reference (the last iterations divide roundoff by an eigenvalue gap of 1e-2)
    Xr = np.array(g["x"]).reshape(n, p)
    dist = np.linalg.norm(X @ X.T - Xr @ Xr.T) / np.linalg.norm(Xr @ Xr.T)
    floor = None
    if oracle_omp is not None:
        op = oracle_omp.stiefel_rq(n, p, rowptr, col, val)
        Xm = oracle_omp.tnt(op, X0.ravel(), prm)["x"].reshape(n, p)
        oracle_omp.free(op)
        floor = np.linalg.norm(Xm @ Xm.T - Xr @ Xr.T) / np.linalg.norm(Xr @ Xr.T)
    print(f"tnt stiefel 8x7x6 mode {mode}: projector distance {dist:.2e}, re-associated reference {floor}")
    assert dist <= floor_or(1e-10, floor)


@pytest.mark.parametrize("key,pre", [("plain", False), ("block_jacobi", True)])
def test_tnt_so3n_device_vs_reference_fixture(harness, oracle, oracle_omp, golden, key, pre):
    """TNT<DeviceVector, DeviceVector> on rotation averaging (cfg3 recipe, N = 40) against the trace
    the REAL reference produced on the same inputs (tests/golden/tnt_so3n_40.json)."""
    g = golden("tnt_so3n_40.json")[key]
    N = g["N"]
    ei, ej, Rt, w, _, Rinit = wl.pose_graph(N, seed=g["seed"])
    prm = oracle.default_params(gradient_tolerance=1e-8, relative_decrease_tolerance=0, stepsize_tolerance=0,
                                preconditioned_gradient_tolerance=0, Delta_tolerance=0, max_iterations=100)
    r = harness.tnt_so3n(N, ei, ej, Rt, w, Rinit, prm, pre)
    assert r["rc"] == 0, r.get("err")
    assert r["status"] == g["status"]
    assert r["outer_iterations"] == g["outer_iterations"]
    assert r["accepted"] == g["accepted"]
    assert list(r["inner_iterations"]) == g["inner_iterations"]
    assert np.allclose(r["objective_values"], g["objective_values"], rtol=1e-10)
    assert np.allclose(r["trust_region_radius"], g["trust_region_radius"], rtol=1e-8)
    assert np.allclose(r["gradient_norms"][:-2], g["gradient_norms"][:-2], rtol=1e-6)
    ex, floor = rel_err(r["x"], g["x"]), None
    if oracle_omp is not None:  # the conditioning floor of this run (conftest.oracle_omp)
        op = oracle_omp.so3n(N, ei, ej, Rt, w, precon_kind=1 if pre else 0)
        floor = rel_err(oracle_omp.tnt(op, Rinit.ravel(), prm)["x"], g["x"])
        oracle_omp.free(op)
    print(f"tnt so3n {key}: iterate error {ex:.2e}, re-associated reference {floor}")
    assert ex <= floor_or(1e-10, floor)
    Rb = r["x"].reshape(N, 3, 3)
    assert np.abs(np.einsum("nij,nkj->nik", Rb, Rb) - np.eye(3)).max() < 1e-12


def test_tnt_so3n_full_size_cfg3_vs_oracle(harness, oracle, oracle_omp):
    """BASELINE cfg3 at its full size (N = 5e5 rotations, 1.5e6 measurements, 3x3 block-Jacobi preconditioner): a whole
    TNT<DeviceVector, DeviceVector> run through the header layer (fused trial step, deferred inner-solve results,
    streamed-measurement model assembly) against the oracle's TNT on the same arrays -- the oracle equals the reference
    headers bit for bit (tests/test_cpu_oracle_templates.py): same outer / inner counts and accept sequence, objective
    trace to 1e-12, final iterate to 1e-10 or the measured conditioning floor.  10 outer iterations: the objective has then
    converged to 1.5e-14 relative (7497.7839906120...), and the 11th iteration's accept / reject decision -- a ratio of
    differences of values that agree to 14 digits -- is decided by rounding (the device accepts a step the oracle
    rejects: seen; any re-association of the reference's own sums can flip it the same way)."""
    N = 500_000
    ei, ej, Rt, w, _, Rinit = wl.pose_graph(N, seed=7, init_sigma=0.02)
    prm = oracle.default_params(max_TPCG_iterations=50, gradient_tolerance=1e-12, relative_decrease_tolerance=0,
                                stepsize_tolerance=0, preconditioned_gradient_tolerance=0, Delta_tolerance=0,
                                max_iterations=10)
    r = harness.tnt_so3n(N, ei, ej, Rt, w, Rinit, prm, 1)
    assert r["rc"] == 0, r.get("err")
    oprob = oracle.so3n(N, ei, ej, Rt, w, precon_kind=1)
    o = oracle.tnt(oprob, Rinit.ravel(), prm)
    oracle.free(oprob)
    assert r["status"] == o["status"] and r["outer_iterations"] == o["outer_iterations"] == 10
    assert int(np.sum(r["inner_iterations"])) >= 20
    assert r["accepted"] == o["accepted"]
    assert list(r["inner_iterations"]) == list(o["inner_iterations"])
    assert np.allclose(r["objective_values"], o["objective_values"], rtol=1e-12)
    assert np.allclose(r["trust_region_radius"], o["trust_region_radius"], rtol=1e-10)
    ex, floor = rel_err(r["x"], o["x"]), None
    if oracle_omp is not None:
        op = oracle_omp.so3n(N, ei, ej, Rt, w, precon_kind=1)
        floor = rel_err(oracle_omp.tnt(op, Rinit.ravel(), prm)["x"], o["x"])
        oracle_omp.free(op)
    print(f"tnt so3n N=5e5: {r['outer_iterations']} outer / {int(np.sum(r['inner_iterations']))} inner, "
          f"iterate error {ex:.2e}, re-associated reference {floor}")
    assert ex <= floor_or(1e-10, floor)
    Rb = r["x"].reshape(N, 3, 3)
    assert np.abs(np.einsum("nij,nkj->nik", Rb, Rb) - np.eye(3)).max() < 1e-12


def test_tnt_stiefel_device_medium_vs_oracle(harness, oracle, oracle_omp):
    """A larger instance (40x36x32 = 46080 rows) against the CPU oracle run on the same arrays."""
    nx, ny, nz, p = 40, 36, 32, 3
    n = nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    X0 = wl.random_stiefel(n, p, seed=11)
    prm = oracle.default_params(gradient_tolerance=1e-6, relative_decrease_tolerance=0, stepsize_tolerance=0,
                                preconditioned_gradient_tolerance=0, Delta_tolerance=0, max_iterations=12,
                                max_TPCG_iterations=50)
    oprob = oracle.stiefel_rq(n, p, rowptr, col, val)
    o = oracle.tnt(oprob, X0.ravel(), prm)
    r = harness.tnt_stiefel(n, p, rowptr, col, val, X0, prm, 0)
    assert r["rc"] == 0, r.get("err")
    assert r["outer_iterations"] == o["outer_iterations"]
    assert list(r["inner_iterations"]) == list(o["inner_iterations"])
    assert r["accepted"] == o["accepted"]
    assert np.allclose(r["objective_values"], o["objective_values"], rtol=1e-11)
    assert np.allclose(r["gradient_norms"], o["gradient_norms"], rtol=1e-7, atol=1e-12)
    ex, floor = rel_err(r["x"], o["x"]), None
    if oracle_omp is not None:  # the conditioning floor of this run (conftest.oracle_omp)
        op = oracle_omp.stiefel_rq(n, p, rowptr, col, val)
        floor = rel_err(oracle_omp.tnt(op, X0.ravel(), prm)["x"], o["x"])
        oracle_omp.free(op)
    print(f"tnt stiefel 40x36x32: iterate error {ex:.2e}, re-associated reference {floor}")
    assert ex <= floor_or(1e-10, floor)
    oracle.free(oprob)


@pytest.mark.parametrize("p", [5, 8])
def test_tnt_stiefel_wide_rows_device_vs_oracle(harness, oracle, oracle_omp, p):
    """r05: the whole drop-in TNT (fused inner solves through the wide-row one-pass Hessian, fused trial steps, deferred
    results) on St(n, p) for p = 5 and 8 against the CPU oracle's run on the same arrays: counts, accept sequence,
    traces, final iterate at max(1e-10, 3 x the conditioning floor)."""
    nx, ny, nz = 30, 28, 26
    n = nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    X0 = wl.random_stiefel(n, p, seed=17 + p)
    prm = oracle.default_params(gradient_tolerance=1e-6, relative_decrease_tolerance=0, stepsize_tolerance=0,
                                preconditioned_gradient_tolerance=0, Delta_tolerance=0, max_iterations=10,
                                max_TPCG_iterations=40)
    oprob = oracle.stiefel_rq(n, p, rowptr, col, val)
    o = oracle.tnt(oprob, X0.ravel(), prm)
    r = harness.tnt_stiefel(n, p, rowptr, col, val, X0, prm, 0)
    assert r["rc"] == 0, r.get("err")
    assert r["outer_iterations"] == o["outer_iterations"]
    assert list(r["inner_iterations"]) == list(o["inner_iterations"])
    assert r["accepted"] == o["accepted"]
    assert np.allclose(r["objective_values"], o["objective_values"], rtol=1e-11)
    assert np.allclose(r["gradient_norms"], o["gradient_norms"], rtol=1e-7, atol=1e-12)
    ex, floor = rel_err(r["x"], o["x"]), None
    if oracle_omp is not None:
        op = oracle_omp.stiefel_rq(n, p, rowptr, col, val)
        floor = rel_err(oracle_omp.tnt(op, X0.ravel(), prm)["x"], o["x"])
        oracle_omp.free(op)
    print(f"tnt stiefel p = {p}: iterate error {ex:.2e}, re-associated reference {floor}")
    assert ex <= floor_or(1e-10, floor)
    X = r["x"].reshape(n, p)
    assert np.abs(X.T @ X - np.eye(p)).max() < 1e-12
    oracle.free(oprob)


# ----------------------------------------------------------------------------------------------
# LSQR and TNLS on DeviceVector (generic loops of the drop-in headers through the Vector concept)
# ----------------------------------------------------------------------------------------------
def _nonsym_sparse(n, seed):
    import scipy.sparse as sps
    rng = np.random.default_rng(seed)
    A = sps.diags([np.full(n - 1, -1.0), np.full(n, 4.0), np.full(n - 1, 2.0)], [-1, 0, 1]).tolil()
    for _ in range(3 * n):
        i, j = rng.integers(0, n, size=2)
        A[i, j] += rng.normal() * .3
    return sps.csr_matrix(A)


LSQR_TOL = 1e-10   # BASELINE.json's iterate tolerance (r03 asserted 1e-9 here; measured errors are printed)


@pytest.mark.parametrize("mode", [0, 1], ids=["fused", "generic"])
@pytest.mark.parametrize("kw", [dict(), dict(lam=0.3), dict(Delta=0.5), dict(max_iterations=7),
                                dict(btol=1e-12, Atol=1e-12, Acond_limit=50.0)])
def test_lsqr_device_matches_host_template(harness, golden, kw, mode):
    """IterativeSolvers.h:552-855 on DeviceVector against (i) what the REAL reference returned on the same inputs
    (tests/golden/lsqr_tnls.json, made by make_golden.py from oracle/_ref/libref.so) and (ii) the same template on a
    host vector: iterates to 1e-9, same iteration count."""
    import oracle_py
    hz = oracle_py.TemplateHarness()
    n = 300
    A = _nonsym_sparse(n, 2)
    b = np.random.default_rng(9).normal(size=n)
    d = harness.lsqr_csr(A, b, mode=mode, **kw)
    fx = [c for c in golden("lsqr_tnls.json")["lsqr"] if c["kw"] == kw][0]
    assert abs(float(A.sum()) - fx["A_checksum"]) < 1e-9
    assert d["rc"] == 0, d["err"]
    assert d["iterations"] == fx["iterations"]
    ex = np.abs(d["x"] - np.array(fx["x"])).max() / max(1.0, np.abs(fx["x"]).max())
    en = abs(d["xnorm"] - fx["xnorm"]) / max(1.0, fx["xnorm"])
    h = hz.lsqr_dense(A.toarray(), b, **kw)
    eh = np.abs(d["x"] - h["x"]).max() / max(1.0, np.abs(h["x"]).max())
    print(f"lsqr {kw} mode {mode}: x vs reference fixture {ex:.2e}, |x| {en:.2e}, x vs host template {eh:.2e}")
    assert ex <= LSQR_TOL and en <= LSQR_TOL
    assert d["iterations"] == h["iterations"]
    assert eh <= LSQR_TOL and abs(d["xnorm"] - h["xnorm"]) <= LSQR_TOL * max(1.0, h["xnorm"])


def test_lsqr_device_large(harness):
    """n = 1e6: residual reduction property of tests/IterativeSolvers_unit_test.cpp:254-310 style"""
    import scipy.sparse as sps
    n = 1_000_000
    A = sps.diags([np.full(n - 1, -1.0), np.full(n, 3.0), np.full(n - 1, 1.5)], [-1, 0, 1], format="csr")
    xs = np.sin(np.arange(n) * 1e-3)
    b = A @ xs
    d = harness.lsqr_csr(A, b, btol=1e-10, Atol=1e-10, max_iterations=200)     # fused mi_lsqr
    assert d["rc"] == 0, d["err"]
    g = harness.lsqr_csr(A, b, btol=1e-10, Atol=1e-10, max_iterations=200, mode=1)
    assert g["iterations"] == d["iterations"] and np.abs(g["x"] - d["x"]).max() < 1e-10
    assert np.linalg.norm(A @ d["x"] - b) <= 1e-8 * np.linalg.norm(b)
    assert np.abs(d["x"] - xs).max() < 1e-6


@pytest.mark.parametrize("mode", [0, 1], ids=["fused_lsqr", "generic"])
@pytest.mark.parametrize("kw", [dict(), dict(root_tolerance=0.0, gradient_tolerance=1e-6),
                                dict(max_LSQR_iterations=3, max_iterations=8)])
def test_tnls_device_matches_host_template(harness, golden, kw, mode):
    """TNLS.h:265-729 on DeviceVector for F(x) = A x - b against the REAL reference's result on the same inputs
    (tests/golden/lsqr_tnls.json) and the host-vector run of the same template."""
    import oracle_py
    hz = oracle_py.TemplateHarness()
    n = 200
    A = _nonsym_sparse(n, 5)
    rng = np.random.default_rng(6)
    b, x0 = rng.normal(size=n), rng.normal(size=n)
    d = harness.tnls_affine(A, b, x0, mode=mode, **kw)
    fx = [c for c in golden("lsqr_tnls.json")["tnls"] if c["kw"] == kw][0]
    assert abs(float(A.sum()) - fx["A_checksum"]) < 1e-9
    assert d["rc"] == 0, d["err"]
    assert d["status"] == fx["status"]
    if kw.get("root_tolerance", 1.0) > 0:
        assert (d["outer"], d["inner_total"]) == (fx["outer"], fx["inner_total"])
    ex = np.abs(d["x"] - np.array(fx["x"])).max() / max(1.0, np.abs(fx["x"]).max())
    ef = abs(d["f"] - fx["f"]) / max(1.0, abs(fx["f"]))
    print(f"tnls {kw} mode {mode}: x vs reference fixture {ex:.2e}, f {ef:.2e}")
    assert ex <= LSQR_TOL and ef <= LSQR_TOL
    h = hz.tnls_affine(A.toarray(), b, x0, **kw)
    assert d["status"] == h["status"]
    if kw.get("root_tolerance", 1.0) > 0:
        assert (d["outer"], d["inner_total"]) == (h["outer"], h["inner_total"])
    # else: |F| is driven to ~1e-15 and the run ends on the trust-region radius after a rounding-dependent
    # number of no-progress passes (15 vs 13 here); the iterates agree to 1e-16
    assert np.abs(d["x"] - h["x"]).max() <= 1e-9 * max(1.0, np.abs(h["x"]).max())
    assert abs(d["f"] - h["f"]) <= 1e-9 * max(1.0, abs(h["f"]))


def test_example_client_program_runs():
    """examples/tnt_stiefel_device.cpp: a stand-alone client of the drop-in headers (random start on St(64000,3) ->
    sum of the three smallest eigenvalues to 1e-6 relative; the program checks that itself)."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "examples", "bin", "tnt_stiefel_device")
    if not os.path.exists(exe):
        from optimization_amd import build
        build.build_harness()
    r = subprocess.run([exe, "40"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "smallest eigenvalues" in r.stdout


# ----------------------------------------------------------------------------------------------
# TNT outer loop on the device: fused trial step + deferred inner-solve result
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pre", [0, 1], ids=["plain", "block_jacobi"])
def test_tnt_so3n_fused_trial_equals_statement_sequence(harness, oracle, pre):
    """cfg3 recipe (N = 4000): TNT with the tagged retraction (mi_so3n_trial: one chain, one read-back per outer
    iteration, also WITH the block-Jacobi preconditioner -- the preconditioned TNT of BASELINE cfg3) against the
    same run with the plain retraction (the reference's statement sequence, Riemannian/TNT.h:493-587): identical
    bits in every trace, <= 1.2 host synchronisations per outer iteration instead of ~5."""
    N = 4000
    ei, ej, Rt, w, _, Rinit = wl.pose_graph(N, seed=13)
    prm = oracle.default_params(gradient_tolerance=1e-7, relative_decrease_tolerance=0, stepsize_tolerance=0,
                                preconditioned_gradient_tolerance=0, Delta_tolerance=0, max_iterations=60)
    runs, syncs = {}, {}
    for plain in (0, 1):
        r = harness.tnt_so3n(N, ei, ej, Rt, w, Rinit, prm, pre | (2 * plain))
        assert r["rc"] == 0, r.get("err")
        runs[plain], syncs[plain] = r, harness.L.hd_last_tnt_syncs()
    a, b = runs[0], runs[1]
    assert a["status"] == b["status"] == 0
    assert a["outer_iterations"] == b["outer_iterations"] and a["accepted"] == b["accepted"]
    for k in ("inner_iterations", "objective_values", "gradient_norms", "preconditioned_gradient_norms",
              "trust_region_radius", "update_step_norms", "update_step_M_norms", "gain_ratios", "x"):
        assert np.array_equal(a[k], b[k]), k
    outer = a["outer_iterations"]
    print("so3n pre", pre, "outer", outer, "syncs fused", syncs[0], "plain", syncs[1])
    assert syncs[0] <= 1.2 * outer + 3
    assert syncs[1] >= 3 * outer


def test_tnt_stiefel_syncs_per_outer_iteration(harness, oracle):
    """cfg2 recipe (40x36x32): one read-back per outer iteration (the inner solve's result travels with the trial
    step's, MI355::DeferScope / mi_stpcg_collect) -- and the same bits as the run without the fused trial step."""
    nx, ny, nz, p = 40, 36, 32, 3
    n = nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    X0 = wl.random_stiefel(n, p, seed=11)
    prm = oracle.default_params(gradient_tolerance=1e-6, relative_decrease_tolerance=0, stepsize_tolerance=0,
                                preconditioned_gradient_tolerance=0, Delta_tolerance=0, max_iterations=14,
                                max_TPCG_iterations=50)
    a = harness.tnt_stiefel(n, p, rowptr, col, val, X0, prm, 0)
    sa = harness.L.hd_last_tnt_syncs()
    b = harness.tnt_stiefel(n, p, rowptr, col, val, X0, prm, 2)
    sb = harness.L.hd_last_tnt_syncs()
    assert a["rc"] == 0 and b["rc"] == 0
    for k in ("inner_iterations", "objective_values", "gradient_norms", "trust_region_radius", "gain_ratios", "x"):
        assert np.array_equal(a[k], b[k]), k
    outer = a["outer_iterations"]
    print("stiefel outer", outer, "syncs fused", sa, "plain retraction", sb)
    assert sa <= 1.2 * outer + 3
    assert sb >= 2 * outer


def test_tnt_supplied_objective_is_the_one_called(harness, oracle):
    """a client objective that is NOT the problem's own (here f + 1 in a plain lambda) next to the
    problem's tagged model and retraction.  The reference always calls the supplied f; the fused trial step would
    evaluate the owner's f instead -- so TNT must keep the statement sequence: every recorded objective value is
    the owner's + 1, and the iterates are those of the plain run."""
    nx, ny, nz, p = 10, 9, 8, 3
    n = nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    X0 = wl.random_stiefel(n, p, seed=5)
    prm = oracle.default_params(gradient_tolerance=1e-7, relative_decrease_tolerance=0, stepsize_tolerance=0,
                                preconditioned_gradient_tolerance=0, Delta_tolerance=0, max_iterations=30,
                                max_TPCG_iterations=50)
    a = harness.tnt_stiefel(n, p, rowptr, col, val, X0, prm, 0)
    c = harness.tnt_stiefel(n, p, rowptr, col, val, X0, prm, 3)
    assert a["rc"] == 0 and c["rc"] == 0
    assert c["objective_values"][0] == a["objective_values"][0] + 1.0
    assert np.all(c["objective_values"] > 1.0)
    assert abs(c["f"] - (a["f"] + 1.0)) < 1e-12
    # same decisions while df is far above the rounding of (f + 1)
    k = 5
    assert list(c["inner_iterations"][:k]) == list(a["inner_iterations"][:k])


# ---- r06: TNLS with a Jacobian that CHANGES, on the device ------------------------------------
@pytest.fixture(scope="module")
def sinfit():
    import harness_py
    return harness_py.SinfitHarness()


@pytest.mark.parametrize("mode", [0, 1], ids=["fused_lsqr", "generic"])
@pytest.mark.parametrize("case", ["root", "least_squares", "least_squares_preconditioned"])
def test_tnls_sinfit_with_a_changing_jacobian_on_the_device(sinfit, golden, case, mode):
    """The reference's own TNLS problem (tests/TNLS_unit_test.cpp:151-260: y = sin(b0 t + b1), m = 100; root finding,
    least squares without and with the right preconditioner R^-1 of J'J = R'R) through Riemannian::TNLS on DeviceVector
    with HIP kernels of the client's own (tests/cpp/harness_sinfit.hip): J(x) hands back FRESH tagged operators at every
    linearisation, so TNLS must re-tag its LSQR operators every outer iteration (TNLS.h:414-462) -- and with the
    preconditioner compose them on the device (mi_op_create_compose).  Against the REAL reference's result on the same
    input bytes (tests/golden/tnls_sinfit.json): status, outer and inner counts, beta and |F| to 1e-10.  mode 0: every
    inner solve ran in the fused mi_lsqr (mi_ctx_fusion_counters); mode 1: the same kernels behind plain lambdas."""
    fx = golden("tnls_sinfit.json")
    c = [c_ for c_ in fx["cases"] if c_["name"] == case][0]
    r = sinfit.tnls_sinfit(fx["t"], c["y"], fx["beta0"], mode=mode, **c["kw"])
    assert r["rc"] == 0, r["err"]
    eb = float(np.abs(r["beta"] - np.array(c["beta"])).max() / np.abs(c["beta"]).max())
    ef = abs(r["f"] - c["f"]) / max(abs(c["f"]), 1e-3)   # (the root case ends at |F| ~ 4e-10: absolute there)
    print(f"tnls sinfit {case} mode {mode}: status {r['status']}, outer {r['outer']} (ref {c['outer']}), inner "
          f"{r['inner_total']} (ref {c['inner_total']}), beta {eb:.2e}, f {ef:.2e}, jacobians {r['jacobians']}, {r['counters']}")
    assert r["status"] == c["status"]
    assert (r["outer"], r["inner_total"]) == (c["outer"], c["inner_total"])
    assert eb <= 1e-10 and ef <= 1e-10
    assert r["jacobians"] >= 3          # the Jacobian really changed hands several times
    k = r["counters"]
    if mode == 0:
        assert k["fused_lsqr_solves"] == r["outer"] and k["generic_lsqr_solves"] == 0, k
    else:
        assert k["generic_lsqr_solves"] == r["outer"] and k["fused_lsqr_solves"] == 0, k


# ---- r06: the fusion boundary is observable -----------------------------------------------------
def test_wrapping_the_hessian_in_a_lambda_flips_the_fusion_counters_and_nothing_else(harness, oracle, golden, capfd,
                                                                                    monkeypatch):
    """The template layer picks the fused entry points by std::function::target<>() probes; a client that wraps the
    Hessian the quadratic model returns in a lambda of its own (mode 4 of hd_tnt_stiefel: what a logger does, and what
    the reference's own adapter lambdas TNT.h:400-426 look like) gets the generic loop.  That used to be silent.  Now:
    mi_ctx_fusion_counters says which side every solve and trial step ran on, MI355OPT_WARN_GENERIC=1 prints one line
    per kind naming the probe that failed -- and the run itself is unchanged (same counts, f to 1e-12)."""
    g = golden("tnt_stiefel_8x7x6.json")
    nx, ny, nz = g["grid"]
    p, n = g["p"], nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    X0 = np.array(g["x0"]).reshape(n, p)
    prm = oracle.default_params(gradient_tolerance=1e-8, relative_decrease_tolerance=0, stepsize_tolerance=0,
                                preconditioned_gradient_tolerance=0, Delta_tolerance=0, max_iterations=200,
                                max_TPCG_iterations=50)
    monkeypatch.setenv("MI355OPT_WARN_GENERIC", "1")
    capfd.readouterr()
    a = harness.tnt_stiefel(n, p, rowptr, col, val, X0, prm, 0)
    ka = harness.fusion_counters()
    err_a = capfd.readouterr().err
    b = harness.tnt_stiefel(n, p, rowptr, col, val, X0, prm, 4)
    kb = harness.fusion_counters()
    err_b = capfd.readouterr().err
    assert a["rc"] == 0 and b["rc"] == 0, (a.get("err"), b.get("err"))
    print("tagged :", ka, "\nwrapped:", kb, "\n", err_b.strip())
    outer = a["outer_iterations"]
    # nothing else changes
    assert (b["status"], b["outer_iterations"], b["accepted"]) == (a["status"], outer, a["accepted"])
    assert list(b["inner_iterations"]) == list(a["inner_iterations"])
    assert abs(a["f"] - b["f"]) < 1e-12 and abs(a["f"] - g["f"]) < 1e-12
    # the tagged run: every inner solve in mi_stpcg, every trial point by the fused chain, no generic work at all
    assert ka["fused_stpcg_solves"] == outer and ka["generic_stpcg_solves"] == 0
    assert ka["fused_trial_steps"] == outer and ka["generic_trial_steps"] == 0
    assert "runs the GENERIC loop" not in err_a
    # the wrapped run: the other side of the boundary, visibly
    assert kb["generic_stpcg_solves"] == outer and kb["fused_stpcg_solves"] == 0
    assert kb["generic_trial_steps"] == outer and kb["fused_trial_steps"] == 0
    assert kb["generic_inner_products"] > 10 * ka["generic_inner_products"]
    assert err_b.count("STPCG on MI355::DeviceVector runs the GENERIC loop") == 1      # once, not once per solve
    assert "MI355::DeviceOperator" in err_b and "trial step" in err_b
