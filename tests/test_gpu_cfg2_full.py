"""Parity at bench.py's own operating point: BASELINE cfg2 at its full size, St(1e6,3) on the
100^3 grid.

  * the exact solve bench.py times (50 STPCG iterations, kappa_fgr 1e-12, theta 1, Delta 1e3, at the bench iterate)
    in every matrix format the library has, against the reference: iteration count, alpha / beta / kappa / <r,v>
    traces, |s|_M and the step s;
  * a whole TNT run from a random start (max_TPCG_iterations = 50) against the trace the REAL reference produced
    here (tests/golden/cfg2_full.json, written by tests/golden/make_golden_full.py from oracle/_ref/libref.so).

The fixture holds scalars and checksums only (24 MB vectors do not travel).  The vectors the device results are
compared with come from the plain-C oracle re-run on the GPU box's host on the same inputs.  That oracle IS the
reference bit for bit -- at small sizes on every problem (tests/test_cpu_oracle_templates.py) and on this very
full-size solve (::test_oracle_is_the_reference_on_the_full_size_bench_solve there, SHA-256 of the 24 MB step).
r06: the inputs come from optimization_amd/wlgen.c (mt19937_64, own sin, own QR) and are the same bytes on every
machine, so the oracle's run on the GPU box must reproduce the fixture made on the build machine exactly.

Tolerance: BASELINE.json asks for iterates within 1e-10 relative.  r06: the STPCG solves are compared ON IDENTICAL
INPUTS (the oracle's gradient uploaded bit for bit) at the plain 1e-10 -- measured 6e-15.  Where an input cannot be
shared (a whole TNT run: every outer iteration's gradient is the device's own; or the bench solve started from the
device's own gradient) the test says in numbers what the conditioning does: the SAME reference algorithm with its sums
re-associated (the oracle's OpenMP build: identical statements, per-thread partial sums) moves away from the
sequential-sum reference by `floor`; no implementation whose reduction order differs from the reference's can be asked
to do better than a small multiple of that.
"""
import os

import numpy as np
import pytest

from conftest import rel_err
from optimization_amd import workloads as wl

pytestmark = pytest.mark.gpu

NX = NY = NZ = 100
P = 3
N = NX * NY * NZ


@pytest.fixture(scope="module")
def cfg2(oracle, golden):
    import oracle_py
    fx = golden("cfg2_full.json")
    assert fx["grid"] == [NX, NY, NZ] and fx["p"] == P
    rowptr, col, val = wl.laplacian_3d(NX, NY, NZ)
    oprob = oracle.stiefel_rq(N, P, rowptr, col, val)
    omp = None
    try:
        omp = oracle_py.Oracle(omp=True)
        omp.set_threads(4)
        omp_prob = omp.stiefel_rq(N, P, rowptr, col, val)
    except OSError:
        omp_prob = None
    yield dict(fx=fx, csr=(rowptr, col, val), oprob=oprob, omp=omp, omp_prob=omp_prob)
    oracle.free(oprob)
    if omp_prob is not None:
        omp.free(omp_prob)


@pytest.fixture(scope="module")
def bench_solve(oracle, cfg2):
    """The oracle's bench solve on this host's arrays, and the re-associated run that gives the floor."""
    fx = cfg2["fx"]["bench_stpcg"]
    it = fx["iterate"]
    Xb, modes = wl.stiefel_bench_iterate(NX, NY, NZ, P, eps=it["eps"], seed=it["seed"])
    assert [list(m) for m in modes] == it["modes"]
    prm = fx["params"]
    g = oracle.eval_grad(cfg2["oprob"], Xb.ravel())
    o = oracle.stpcg_problem(cfg2["oprob"], Xb.ravel(), g, prm["Delta"], max_iterations=prm["max_iterations"],
                             kappa_fgr=prm["kappa_fgr"], theta=prm["theta"], trace_cap=64)
    # this host's oracle run against what the REAL reference did on the build container: the inputs are the same bytes
    # on every machine (wlgen.c) and the oracle library is the one built there, so the solve must be the fixture's
    # BIT FOR BIT -- the vectors the device is compared with below ARE the reference's
    import hashlib
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()  # noqa: E731
    assert sha(Xb) == it["sha256"]
    assert o["iterations"] == fx["iterations"] == 50
    assert o["M_norm"] == fx["M_norm"]
    for k in ("alpha", "beta", "kappa", "rv"):
        assert list(o["trace"][k]) == fx["trace"][k], k
    assert sha(g) == fx["g"]["sha256"] and sha(o["s"]) == fx["s"]["sha256"]
    floor = None
    if cfg2["omp_prob"] is not None:
        omp = cfg2["omp"]
        gm = omp.eval_grad(cfg2["omp_prob"], Xb.ravel())
        m = omp.stpcg_problem(cfg2["omp_prob"], Xb.ravel(), gm, prm["Delta"], max_iterations=prm["max_iterations"],
                              kappa_fgr=prm["kappa_fgr"], theta=prm["theta"], trace_cap=64)
        floor = dict(s=rel_err(m["s"], o["s"]), iterations=m["iterations"],
                     alpha=float(np.max(np.abs(m["trace"]["alpha"] / o["trace"]["alpha"] - 1))),
                     beta=float(np.max(np.abs(m["trace"]["beta"] / o["trace"]["beta"] - 1))))
    return dict(Xb=Xb, g=g, o=o, prm=prm, floor=floor)


FORMATS = {
    "default_window_packed": {},
    "generic_csr_12_byte_entries": {"MI355OPT_NO_PACKED": "1"},
    "words16": {"MI355OPT_WORDS16": "1"},
    "packed_streaming_no_window": {"MI355OPT_NO_WINDOW": "1"},
    "two_pass_operator": {"MI355OPT_NO_DIRGRAM": "1"},
}


@pytest.mark.parametrize("fmt", list(FORMATS))
def test_bench_solve_matches_the_reference(cfg2, bench_solve, monkeypatch, fmt):
    """bench.py's solve, N = 3e6, 50 iterations, in every matrix format, against the reference's -- on the SAME INPUTS
    (r06): STPCG's input g is the oracle's gradient uploaded bit for bit, and the bar is BASELINE.json's plain 1e-10 on
    the step, the alpha / beta / kappa traces and |s|_M, with no conditioning-floor allowance.  (Measured: 6e-15 on the
    step.  With the gradient the DEVICE computes as the input -- it differs from the oracle's by 5e-13 of |g|, the
    cancellation in A X - X sym(X'AX) at a near-optimal iterate -- the same solve ends 2e-10 from the reference's, and
    so does the reference itself with re-associated sums: that is the conditioning of the problem acting on a
    perturbed INPUT, profiles/r06_parity_curve.md; it is printed below, and asserted against the floor, for the default
    format.)"""
    from optimization_amd import capi
    for k, v in FORMATS[fmt].items():
        monkeypatch.setenv(k, v)
    c = capi.Context(0)  # (MI355OPT_NO_DIRGRAM is read at context creation, MI355OPT_NO_PACKED at matrix creation)
    try:
        rowptr, col, val = cfg2["csr"]
        A = c.csr(N, rowptr, col, val)
        prob = c.stiefel_rq(A, N, P)
        X = c.upload(bench_solve["Xb"])
        g, H = prob.model(X)
        eg = rel_err(g.numpy(), bench_solve["g"])
        assert eg < 1e-11  # (|g| is 1e-3 of |A X| at this near-optimal iterate)
        prm, o = bench_solve["prm"], bench_solve["o"]
        c.ktime_enable("stiefel_hess_fused", True)
        c.ktime_enable("stiefel_finish_dots", True)
        kw = dict(Delta=prm["Delta"], max_iterations=prm["max_iterations"], kappa_fgr=prm["kappa_fgr"],
                  theta=prm["theta"], trace_cap=64)
        r = c.stpcg(c.upload(bench_solve["g"]), H, **kw)
        one_pass = c.ktime_read("stiefel_hess_fused")[0]
        two_pass = c.ktime_read("stiefel_finish_dots")[0]
        assert (two_pass > 0) == (fmt == "two_pass_operator") and (one_pass > 0) == (fmt != "two_pass_operator")
        assert r["iterations"] == o["iterations"] == 50
        assert r["exit_reason"] == o["exit_reason"]
        ea = float(np.max(np.abs(r["trace"]["alpha"] / o["trace"]["alpha"] - 1)))
        eb = float(np.max(np.abs(r["trace"]["beta"] / o["trace"]["beta"] - 1)))
        ek = float(np.max(np.abs(r["trace"]["kappa"] / o["trace"]["kappa"] - 1)))
        es = rel_err(r["s"].numpy(), o["s"])
        em = abs(r["M_norm"] - o["M_norm"]) / o["M_norm"]
        print(f"{fmt}, same g bits: s {es:.2e}, alpha {ea:.2e}, beta {eb:.2e}, kappa {ek:.2e}, |s|_M {em:.2e}   (plain 1e-10)")
        assert es <= 1e-10 and ea <= 1e-10 and eb <= 1e-10 and ek <= 1e-10 and em <= 1e-11, (es, ea, eb, ek, em)
        if fmt == "default_window_packed":
            # the same solve from the gradient the device computed: a perturbed INPUT, held against the floor
            rd = c.stpcg(g, H, **kw)
            fl = bench_solve["floor"] or dict(s=0.0, alpha=0.0, beta=0.0)
            esd = rel_err(rd["s"].numpy(), o["s"])
            print(f"{fmt}, the device's own gradient as input (differs from the oracle's by {eg:.2e}): s {esd:.2e} "
                  f"(re-associated reference, its own gradient: {fl['s']:.2e})")
            assert rd["iterations"] == 50 and esd <= max(1e-10, 3 * fl["s"]), (esd, fl)
    finally:
        c.close()


def test_bench_solve_error_curve_plain_tolerance(cfg2, bench_solve, oracle):
    """the error of the step along the bench solve, k = 5, 10, 20 (= the driver's --steps), 30, 40,
    50 iterations -- plain 1e-10 at EVERY k on identical inputs (measured 1e-14 ... 6e-15; the whole curve for k = 1 ...
    50: profiles/r06_parity_curve.md).  With the device's own gradient as the input the error follows the conditioning
    curve of the solve (5e-13 at k = 1 -> 2e-10 at k = 50, crossing 1e-10 at k = 34): asserted at 1e-10 up to k = 30."""
    from optimization_amd import capi
    c = capi.Context(0)
    try:
        rowptr, col, val = cfg2["csr"]
        A = c.csr(N, rowptr, col, val)
        prob = c.stiefel_rq(A, N, P)
        g, H = prob.model(c.upload(bench_solve["Xb"]))
        gs = c.upload(bench_solve["g"])
        prm = bench_solve["prm"]
        for k in (5, 10, 20, 30, 40, 50):
            kw = dict(max_iterations=k, kappa_fgr=prm["kappa_fgr"], theta=prm["theta"])
            o = oracle.stpcg_problem(cfg2["oprob"], bench_solve["Xb"].ravel(), bench_solve["g"], prm["Delta"], **kw)
            r = c.stpcg(gs, H, Delta=prm["Delta"], **kw)
            rd = c.stpcg(g, H, Delta=prm["Delta"], **kw)
            es, esd = rel_err(r["s"].numpy(), o["s"]), rel_err(rd["s"].numpy(), o["s"])
            print(f"k = {k:2d}: s vs reference, same g bits {es:.2e}; the device's own g {esd:.2e}")
            assert r["iterations"] == o["iterations"] == k
            assert es <= 1e-10, (k, es)
            if k <= 30:
                assert esd <= 1e-10, (k, esd)
    finally:
        c.close()


def test_full_cfg2_tnt_run_matches_the_reference_trace(cfg2, oracle):
    """A whole TNT run on cfg2 at full size through the drop-in templates on DeviceVector (fused inner solves, fused
    trial steps), against what the REAL reference did on the same inputs: status, outer / inner iteration counts,
    the accept sequence, the f / |g| / radius traces -- and the final iterate against the oracle's run on the same
    arrays, also as a subspace (the minimiser of a Rayleigh quotient is one)."""
    import harness_py
    fx = cfg2["fx"]["tnt"]
    rowptr, col, val = cfg2["csr"]
    X0 = wl.random_stiefel(N, P, seed=fx["seed"])
    import hashlib
    assert hashlib.sha256(X0.tobytes()).hexdigest() == fx["x0"]["sha256"]   # (the same start, byte for byte: wlgen.c)
    prm = oracle.default_params(**fx["params"])
    hz = harness_py.DeviceHarness()
    r = hz.tnt_stiefel(N, P, rowptr, col, val, X0, prm, 0)
    assert r["rc"] == 0, r.get("err")
    syncs = hz.L.hd_last_tnt_syncs()
    assert r["status"] == fx["status"]
    assert r["outer_iterations"] == fx["outer_iterations"]
    assert list(r["inner_iterations"]) == fx["inner_iterations"]
    assert r["accepted"] == fx["accepted"]
    rho, rho_ref = np.array(r["gain_ratios"]), np.array(fx["gain_ratios"])
    assert list(rho > 0.05) == list(rho_ref > 0.05) and list(rho >= 0.9) == list(rho_ref >= 0.9)  # eta1, eta2
    assert np.allclose(r["objective_values"], fx["objective_values"], rtol=1e-11)
    assert np.allclose(r["trust_region_radius"], fx["trust_region_radius"], rtol=1e-10)
    assert np.allclose(r["update_step_M_norms"], fx["update_step_M_norms"], rtol=1e-9)
    assert np.allclose(r["gradient_norms"], fx["gradient_norms"], rtol=1e-6)
    assert abs(r["f"] - fx["f"]) <= 1e-12 * abs(fx["f"])
    print("cfg2 TNT: outer", r["outer_iterations"], "inner", int(np.sum(r["inner_iterations"])), "host syncs", syncs,
          "max |f - f_ref| / f", float(np.max(np.abs(np.array(r["objective_values"]) / np.array(fx["objective_values"]) - 1))))
    assert syncs <= 1.2 * r["outer_iterations"] + 3
    # final iterate: the oracle's run (bitwise the reference's) on this host
    o = oracle.tnt(cfg2["oprob"], X0.ravel(), prm)
    assert o["outer_iterations"] == fx["outer_iterations"] and list(o["inner_iterations"]) == fx["inner_iterations"]
    assert np.allclose(o["objective_values"], fx["objective_values"], rtol=1e-11)
    X, Xr = r["x"].reshape(N, P), o["x"].reshape(N, P)
    assert np.abs(X.T @ X - np.eye(P)).max() < 1e-12
    # distance of the subspaces: |(I - Xr Xr') X|_F = |sin Theta|_F  (= |X X' - Xr Xr'|_F / sqrt 2), no cancellation
    dist = np.linalg.norm(X - Xr @ (Xr.T @ X))
    # ... and of the iterates themselves
    ex = rel_err(X, Xr)
    floor = None
    if cfg2["omp_prob"] is not None:
        m = cfg2["omp"].tnt(cfg2["omp_prob"], X0.ravel(), prm)
        Xm = m["x"].reshape(N, P)
        floor = dict(x=rel_err(Xm, Xr), subspace=float(np.linalg.norm(Xm - Xr @ (Xr.T @ Xm))),
                     outer=int(m["outer_iterations"]))
    print(f"cfg2 TNT final iterate: rel err {ex:.2e}, subspace distance {dist:.2e}; re-associated reference: {floor}")
    fx_floor = floor or dict(x=0.0, subspace=0.0)
    assert dist <= max(1e-10, 3 * fx_floor["subspace"])
    assert ex <= max(1e-10, 3 * fx_floor["x"])


def test_two_kernel_step_experiment(cfg2, bench_solve):
    """r05, VERDICT r04 item 8 -- an OPT-IN experiment, never the default: with MI355OPT_TWO_KERNEL_STEP the Hessian pass
    also leaves <r,Hp> and <p,r>, <r+,r+> = <r,r> + 2 alpha <r,Hp> + alpha^2 <Hp,Hp> replaces the sum over the new residual
    (IterativeSolvers.h:408), and the two CG kernels of an iteration are one (k_cg_step2).  The deliverable is the DISTANCE
    FROM THE REFERENCE next to the conditioning floor at the bench point (N = 3e6, 50 iterations), printed here; the
    assertions only hold the experiment to being a CG solve of the same system: same count and exit, iterate to 1e-6."""
    from optimization_amd import capi
    prm, o, fl = bench_solve["prm"], bench_solve["o"], bench_solve["floor"] or dict(s=0.0, alpha=0.0, beta=0.0)
    out = {}
    c = capi.Context(0)
    try:
        rowptr, col, val = cfg2["csr"]
        A = c.csr(N, rowptr, col, val)
        prob = c.stiefel_rq(A, N, P)
        g, H = prob.model(c.upload(bench_solve["Xb"]))
        for mode in (0, 1):
            c.set_option("TWO_KERNEL_STEP", mode)
            c.ktime_enable("cg_pupdate", True)
            c.ktime_reset()
            r = c.stpcg(g, H, Delta=prm["Delta"], max_iterations=prm["max_iterations"], kappa_fgr=prm["kappa_fgr"],
                        theta=prm["theta"], trace_cap=64)
            out[mode] = dict(r, s=r["s"].numpy().copy(), pupdates=c.ktime_read("cg_pupdate")[0])
    finally:
        c.close()
    assert out[0]["pupdates"] >= 50 and out[1]["pupdates"] == 0          # the merged kernel really replaced both
    for mode, name in ((0, "three kernels"), (1, "two kernels (recurrence for <r+,r+>)")):
        r = out[mode]
        ea = float(np.max(np.abs(r["trace"]["alpha"] / o["trace"]["alpha"] - 1)))
        eb = float(np.max(np.abs(r["trace"]["beta"] / o["trace"]["beta"] - 1)))
        print(f"{name}: s {rel_err(r['s'], o['s']):.2e} (floor {fl['s']:.2e}), alpha {ea:.2e} ({fl['alpha']:.2e}), "
              f"beta {eb:.2e} ({fl['beta']:.2e}), |s|_M {abs(r['M_norm'] - o['M_norm']) / o['M_norm']:.2e}")
        assert r["iterations"] == o["iterations"] == 50 and r["exit_reason"] == o["exit_reason"]
    assert rel_err(out[1]["s"], o["s"]) < 1e-6
