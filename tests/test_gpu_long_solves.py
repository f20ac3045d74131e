"""Long inner solves at the reference's defaults: TNTParams::max_TPCG_iterations = 1000
(reference TNT.h:104), and the default Stiefel Hessian on the device carries the p x p projection matrix G(p) of the
direction by RECURRENCE (stpcg_kernels.inc: G(r') = G(r) + alpha G(Hp), G(p') = -G(r') + beta G(p)), whose absolute
error stays at the scale of the INITIAL residual.  Here: an ill-conditioned Riemannian Hessian (long thin grid 128 x 12 x
10, A = 7-point Laplacian + 1e-3 I: the p lowest modes lie along x with gaps ~ (2 p + 1) (pi / 129)^2, kappa_H ~ 1e3 ...
3e3), unpreconditioned, kappa_fgr = 1e-12, max_iterations 200 and 1000, p in {1, 3, 8} -- recurrence form against the
two-pass operator (MI355OPT_NO_DIRGRAM=1: the projection from the product itself, as the reference's callable forms it),
the direct form (p <= 4) and the CPU oracle, all on the SAME gradient bits (the oracle's g uploaded: parity of the
solver on identical inputs; the device's own gradient differs from the oracle's by ~5e-13 and the solve's conditioning
multiplies that, profiles/r06_parity_curve.md).

What the solves look like (oracle): 200 iterations reduce the residual by 1e-6 ... 1e-9 and every trace is still within
the conditioning floor; with the budget of 1000 the solves end after 310 ... 430 iterations in a boundary / kernel exit
(for p >= 2 the Rayleigh quotient is invariant under X -> X Q: the Hessian has p (p - 1) / 2 near-zero eigenvalues at a
minimiser) and the LAST ~100 iterations are chaotic for every implementation -- the re-associated reference itself
(oracle/liboracle_omp.so) is then O(1) away from the sequential reference in alpha, beta and the iteration count.  The
claim tested is therefore two-sided: (a) while the reference algorithm is well determined the recurrence form is the
two-pass form to rounding and both are the oracle to the floor; (b) where it is not, the recurrence form loses the
reference NO EARLIER than the two-pass form does.

MEASURED (r06, identical inputs): the never-anchored recurrences of r03-r05 DO depart from the two-pass form by more
than the re-association floor in long solves -- 200 iterations: step 1.9e-13 / 2.6e-13 from the reference against
3.5e-14 / 8e-15 with the two-pass operator (p = 1 / 3; all far inside 1e-10), and with the budget of 1000 they lose the
reference's alpha trace 30 ... 45 iterations earlier (270 vs 300, 354 vs 395, 256 vs 303).  So G(p) and G(r) are now
RE-ANCHORED by the direct form every 25 iterations (mi_ctx option REANCHOR; two passes over X, Y and the vector per 25
iterations = +2.1 % per iteration in long solves; none inside the driver's 20-iteration bench solve; every 50 would cost
0.9 % but follows the reference's trace only for 261 instead of 305 iterations at p = 8, tools/anchor_probe.py).  The never-anchored form stays
reachable (MI355OPT_REANCHOR=0) and is measured next to the default below.  Numbers: profiles/r06_deep_solves.md."""
import numpy as np
import pytest

from conftest import rel_err, trace_close
from optimization_amd import workloads as wl

pytestmark = pytest.mark.gpu

GRID = (128, 12, 10)
SHIFT = 1e-3
EPS = {1: 1e-6, 3: 1e-4, 8: 1e-6}   # the iterate must be closer to the minimiser than the spectral gap


def _tangency(X, s, n, p):
    S = s.reshape(n, p)
    G = X.T @ S
    return float(np.linalg.norm(G + G.T) / 2 / max(np.linalg.norm(S), 1e-300))


def _first_above(a, ref, bar):
    k = min(len(a), len(ref))
    e = np.abs(np.asarray(a[:k]) / np.asarray(ref[:k]) - 1)
    idx = np.nonzero(e > bar)[0]
    return int(idx[0]) if idx.size else k


@pytest.fixture(scope="module", params=[1, 3, 8])
def problem(request, oracle, oracle_omp):
    p = request.param
    nx, ny, nz = GRID
    n = nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz, shift=SHIFT)
    Xb, modes = wl.stiefel_bench_iterate(nx, ny, nz, p, eps=EPS[p], seed=7)
    assert modes[-1] == (p, 1, 1)   # the lowest modes of this grid all lie along x
    op = oracle.stiefel_rq(n, p, rowptr, col, val)
    g = oracle.eval_grad(op, Xb.ravel())
    mp = oracle_omp.stiefel_rq(n, p, rowptr, col, val) if oracle_omp is not None else None
    if mp is not None:
        oracle_omp.eval_grad(mp, Xb.ravel())   # (binds the model to Xb; the floor solves take the oracle's g too)
    yield dict(p=p, n=n, csr=(rowptr, col, val), Xb=Xb, g=g, op=op, mp=mp)
    oracle.free(op)
    if mp is not None:
        oracle_omp.free(mp)


def _device_solves(pr, maxit, monkeypatch):
    from optimization_amd import capi
    # "recurrence" = the default (G(p), G(r) re-anchored by the direct form every 25 iterations, r06);
    # "recurrence-never-anchored" = the r03-r05 behaviour, kept as the measurement of what the re-anchoring is for
    modes = {"recurrence": {}, "recurrence-never-anchored": {"MI355OPT_REANCHOR": "0"},
             "two-pass": {"MI355OPT_NO_DIRGRAM": "1"}}
    if pr["p"] <= 4:
        modes["direct"] = {"MI355OPT_DIRGRAM_DIRECT": "1"}
    out = {}
    for mode, env in modes.items():
        for k in ("MI355OPT_NO_DIRGRAM", "MI355OPT_DIRGRAM_DIRECT"):
            monkeypatch.setenv(k, env.get(k, "0"))
        monkeypatch.setenv("MI355OPT_REANCHOR", env.get("MI355OPT_REANCHOR", "25"))
        c = capi.Context(0)
        try:
            A = c.csr(pr["n"], *pr["csr"])
            prob = c.stiefel_rq(A, pr["n"], pr["p"])
            gd, H = prob.model(c.upload(pr["Xb"]))
            assert rel_err(gd.numpy(), pr["g"]) < 1e-9   # (|g| ~ eps |A X|: cancellation)
            for k in ("stiefel_hess_fused", "stiefel_finish_dots"):
                c.ktime_enable(k, True)
            r = c.stpcg(c.upload(pr["g"]), H, Delta=1e6, max_iterations=maxit, kappa_fgr=1e-12, theta=1.0,
                        trace_cap=maxit + 2)
            one, two = c.ktime_read("stiefel_hess_fused")[0], c.ktime_read("stiefel_finish_dots")[0]
            assert (one > 0, two > 0) == ((False, True) if mode == "two-pass" else (True, False)), (mode, one, two)
            out[mode] = dict(r, s=r["s"].numpy().copy())
        finally:
            c.close()
    return out


def test_200_iterations_recurrence_form_is_the_two_pass_form_and_the_oracle(problem, oracle, oracle_omp, monkeypatch):
    """(a): 200 iterations, residual reduced by 1e-6 ... 1e-9, well determined.  Counts and exits equal; alpha / beta over
    the WHOLE solve against the oracle at 1e-9 or 3 x the floor envelope; the step at 1e-10 or 3 x floor; recurrence vs
    two-pass: traces to 1e-9 of each other, steps to 1e-10, tangency of the step no worse."""
    pr, maxit = problem, 200
    kw = dict(max_iterations=maxit, kappa_fgr=1e-12, theta=1.0, trace_cap=maxit + 2)
    o = oracle.stpcg_problem(pr["op"], pr["Xb"].ravel(), pr["g"], 1e6, **kw)
    # the floor: the same algorithm with its sums re-associated THREE ways (2, 3, 4 threads) -- at the rounding end of a
    # solve one realisation of it moves by a factor of two from run to run (OpenMP combines the threads' partial sums in
    # arrival order), and with it a bar of "3 x floor" (r06: the two-pass form's beta at k = 197 ... 199 of the p = 8
    # solve, 1.1-1.2e-9, passed against one realisation and failed against the next)
    floors = []
    if pr["mp"] is not None:
        for t in (2, 3, 4):
            oracle_omp.set_threads(t)
            floors.append(oracle_omp.stpcg_problem(pr["mp"], pr["Xb"].ravel(), pr["g"], 1e6, **kw))
    m = floors[-1] if floors else None
    res = _device_solves(pr, maxit, monkeypatch)
    fl_s = max(rel_err(f["s"], o["s"]) for f in floors) if floors else 0.0
    red = float((o["trace"]["rv"][-1] / np.dot(pr["g"], pr["g"])) ** 0.5)
    print(f"p = {pr['p']}: oracle {o['iterations']} iterations, exit {o['exit_reason']}, residual reduction {red:.1e}; "
          f"floor of s {fl_s:.2e}")
    assert o["iterations"] == maxit
    errs = {}
    for mode, r in res.items():
        assert (r["iterations"], r["exit_reason"]) == (o["iterations"], o["exit_reason"]), mode
        errs[mode] = es = rel_err(r["s"], o["s"])
        ta = float(np.max(np.abs(r["trace"]["alpha"] / o["trace"]["alpha"] - 1)))
        print(f"   {mode:26s} s vs oracle {es:.2e}, alpha (max over the solve) {ta:.2e}, tangency of s "
              f"{_tangency(pr['Xb'], r['s'], pr['n'], pr['p']):.2e}")
        if mode == "recurrence-never-anchored":
            continue          # (measured, not held to the bars: it is what the default replaced)
        for key in ("alpha", "beta"):
            ok, msg = trace_close(r["trace"][key], o["trace"][key], [f["trace"][key] for f in floors] if floors else None, 1e-9)
            assert ok, f"{mode} {key}: {msg}"
        assert es <= max(1e-10, 3 * fl_s), (mode, es, fl_s)
    # the re-anchored recurrence is no further from the reference than the never-anchored one (up to rounding noise)
    assert errs["recurrence"] <= 2 * errs["recurrence-never-anchored"] + 1e-14, errs
    a, b = res["recurrence"], res["two-pass"]
    ea = float(np.max(np.abs(a["trace"]["alpha"] / b["trace"]["alpha"] - 1)))
    eb = float(np.max(np.abs(a["trace"]["beta"] / b["trace"]["beta"] - 1)))
    es = rel_err(a["s"], b["s"])
    print(f"   recurrence vs two-pass: alpha {ea:.2e}, beta {eb:.2e}, s {es:.2e}")
    fl_a = float(np.max(np.abs(m["trace"]["alpha"] / o["trace"]["alpha"] - 1))) if m else 0.0
    assert ea <= max(1e-9, 3 * fl_a), (ea, fl_a)
    assert es <= max(1e-10, 3 * fl_s)
    ta, tb = (_tangency(pr["Xb"], r["s"], pr["n"], pr["p"]) for r in (a, b))
    assert ta <= 2 * tb + 1e-12


def test_1000_iteration_budget_recurrence_form_holds_as_long_as_the_two_pass_form(problem, oracle, oracle_omp, monkeypatch):
    """(b): max_iterations = 1000 = the reference's default.  The solves end after 310 ... 430 iterations in a boundary /
    kernel exit, the last ~100 iterations chaotic for every implementation.  Asserted: the recurrence form -- re-anchored
    every 25 iterations, the default since r06 -- follows the oracle's alpha trace (to 1e-6) for as long as the two-pass
    form or the re-associated reference does, and no shorter than the never-anchored recurrences of r03-r05; the
    iteration counts of the device forms lie within 3 % of each other and 8 % of the oracle's."""
    pr, maxit = problem, 1000
    kw = dict(max_iterations=maxit, kappa_fgr=1e-12, theta=1.0, trace_cap=maxit + 2)
    o = oracle.stpcg_problem(pr["op"], pr["Xb"].ravel(), pr["g"], 1e6, **kw)
    m = oracle_omp.stpcg_problem(pr["mp"], pr["Xb"].ravel(), pr["g"], 1e6, **kw) if pr["mp"] is not None else None
    res = _device_solves(pr, maxit, monkeypatch)
    hold_floor = _first_above(m["trace"]["alpha"], o["trace"]["alpha"], 1e-6) if m else None
    hold = {mode: _first_above(r["trace"]["alpha"], o["trace"]["alpha"], 1e-6) for mode, r in res.items()}
    print(f"p = {pr['p']}: oracle {o['iterations']} iterations (exit {o['exit_reason']}); re-associated reference "
          f"{m['iterations'] if m else None}, follows the oracle's alpha to 1e-6 for {hold_floor} iterations; device: "
          + ", ".join(f"{mode} {r['iterations']} (exit {r['exit_reason']}), holds {hold[mode]}" for mode, r in res.items()))
    assert 250 < o["iterations"] < maxit
    bar = min(hold["two-pass"], hold_floor if hold_floor is not None else hold["two-pass"])
    # the (re-anchored) recurrence form follows the reference for as long as the two-pass form or the re-associated
    # reference does, whichever loses it first (5 % slack: once a trace is 1e-7 off, WHEN it crosses 1e-6 is noise) ...
    assert hold["recurrence"] >= 0.95 * bar - 3, (hold, hold_floor)
    # ... which the never-anchored recurrences of r03-r05 did not (p = 1: 270 vs 300, p = 3: 354 vs 377, p = 8: 256 vs 266)
    assert hold["recurrence"] >= hold["recurrence-never-anchored"] - 3, hold
    its = {mode: r["iterations"] for mode, r in res.items() if mode != "recurrence-never-anchored"}
    assert abs(its["recurrence"] - its["two-pass"]) <= max(3, 0.03 * o["iterations"]), its
    assert all(abs(v - o["iterations"]) <= 0.08 * o["iterations"] for v in its.values()), (its, o["iterations"])


def test_200_iterations_through_two_rank_sharding(oracle, oracle_omp):
    """The sharded path drops a collective per iteration by relying on the recurrence: St(15360, 3) on the same
    grid over 2 real processes on GPU 0 (peer-memory layer), 200 iterations on the oracle's gradient bits: counts, exit,
    alpha / beta over the whole solve and the step against the oracle as above; replicated scalars bit-identical."""
    from test_gpu_comm import _run_cfg4_workers
    p = 3
    nx, ny, nz = GRID
    n = nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz, shift=SHIFT)
    Xb, _ = wl.stiefel_bench_iterate(nx, ny, nz, p, eps=EPS[p], seed=7)
    op = oracle.stiefel_rq(n, p, rowptr, col, val)
    g = oracle.eval_grad(op, Xb.ravel())
    kw = dict(max_iterations=200, kappa_fgr=1e-12, theta=1.0, trace_cap=202)
    o = oracle.stpcg_problem(op, Xb.ravel(), g, 1e6, **kw)
    oracle.free(op)
    m = None
    if oracle_omp is not None:
        mp = oracle_omp.stiefel_rq(n, p, rowptr, col, val)
        oracle_omp.eval_grad(mp, Xb.ravel())
        m = oracle_omp.stpcg_problem(mp, Xb.ravel(), g, 1e6, **kw)
        oracle_omp.free(mp)
    env = {"CFG4_SHIFT": repr(SHIFT), "CFG4_MAXIT": "200", "CFG4_DELTA": "1e6"}
    outs, s_sh, g_sh = _run_cfg4_workers(2, GRID, Xb, extra_env=env, port=29640, g_input=g)
    assert all(o_["enabled"] and o_["ipc_error"] == 0 for o_ in outs), outs
    assert rel_err(g_sh, g) < 1e-9
    for k in ("iters", "exit", "M", "rv", "alpha", "beta"):
        assert outs[0][k] == outs[1][k], k
    assert (outs[0]["iters"], outs[0]["exit"]) == (o["iterations"], o["exit_reason"])
    assert all(o_["one_pass_launches"] >= 200 for o_ in outs)      # the recurrence-form one-pass Hessian on every rank
    al = np.array([float.fromhex(a) for a in outs[0]["alpha"]])
    be = np.array([float.fromhex(a) for a in outs[0]["beta"]])
    for key, tr in (("alpha", al), ("beta", be)):
        ok, msg = trace_close(tr, o["trace"][key], m["trace"][key] if m else None, 1e-9)
        assert ok, f"{key}: {msg}"
    es, fl = rel_err(s_sh, o["s"]), (rel_err(m["s"], o["s"]) if m else 0.0)
    print(f"2 ranks, 200 iterations: s vs oracle {es:.2e} (floor {fl:.2e})")
    assert es <= max(1e-10, 3 * fl)
