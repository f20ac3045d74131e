"""Generate the golden fixtures under tests/golden/ by running the REAL reference
(oracle/_ref/libref.so = /root/reference templates compiled in this container) on the cases the
reference's own unit tests define, plus the harness problems at small sizes.

Run here (where /root/reference is mounted):  python tests/golden/make_golden.py
Only data (inputs + expected outputs) is written; no reference source text.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py  # noqa: E402
from optimization_amd import workloads as wl  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
DBL_MAX = float(np.finfo(np.float64).max)


def lst(a):
    return [float(x) for x in np.asarray(a).ravel()]


def tnt_record(r):
    return dict(status=int(r["status"]), status_name=r["status_name"],
                outer_iterations=int(r["outer_iterations"]), accepted=int(r["accepted"]),
                inner_iterations=[int(x) for x in r["inner_iterations"]],
                objective_values=lst(r["objective_values"]), gradient_norms=lst(r["gradient_norms"]),
                trust_region_radius=lst(r["trust_region_radius"]), gain_ratios=lst(r["gain_ratios"]),
                update_step_M_norms=lst(r["update_step_M_norms"]),
                update_step_norms=lst(r["update_step_norms"]), f=float(r["f"]),
                gradfx_norm=float(r["gradfx_norm"]), calls={k: int(v) for k, v in r["calls"].items()})


def main():
    O = oracle_py.Oracle()
    R = oracle_py.Reference()

    # --- STPCG small deterministic cases: tests/IterativeSolvers_unit_test.cpp:86-118,138-251 -------
    g = np.array([21, -.4, 19.])
    Pd = np.array([1000, 100, 1.])
    M = np.array([100, 10, 1.])
    cases = {}
    for name, D, Delta, pre in [("ExactSTPCG", Pd, DBL_MAX, None),
                                ("ExactSTPCGwithNegativeCurvature", -Pd, 1000.0, None),
                                ("ExactSTPCGwithPreconditioning", Pd, DBL_MAX, M),
                                ("ExactSTPCGwithNegativeCurvatureAndPreconditioning", -Pd, 1000.0, M)]:
        r = R.stpcg(g, lambda v, D=D: D * v, P=(lambda v, pre=pre: v / pre) if pre is not None else None,
                    Delta=Delta, max_iterations=3, kappa_fgr=1e-8, theta=.999, trace_cap=8)
        cases[name] = dict(g=lst(g), H_diag=lst(D), M_diag=lst(pre) if pre is not None else None,
                           Delta=Delta, max_iterations=3, kappa_fgr=1e-8, theta=.999,
                           iterations=int(r["iterations"]), M_norm=float(r["M_norm"]), s=lst(r["s"]),
                           alpha=lst(r["trace"]["alpha"]))
    json.dump(cases, open(os.path.join(OUT, "stpcg_small.json"), "w"), indent=1)

    # --- TNT on the sphere: tests/TNT_unit_test.cpp:63-187 -----------------------------------------
    out = {}
    x0 = [-0.5, -0.5, -0.707107]
    for pre in (False, True):
        pr = O.sphere(with_precon=pre)
        p = O.default_params(gradient_tolerance=1e-8, relative_decrease_tolerance=0,
                             stepsize_tolerance=0, preconditioned_gradient_tolerance=0)
        r = R.tnt(pr, x0, p)
        rec = tnt_record(r)
        rec["x"] = lst(r["x"])
        rec["x0"] = x0
        out["precon" if pre else "plain"] = rec
        O.free(pr)
    json.dump(out, open(os.path.join(OUT, "tnt_sphere.json"), "w"), indent=1)

    # --- chained Rosenbrock n = 100 (BASELINE cfg1) ------------------------------------------------
    out = {}
    for pk in (0, 1):
        pr = O.rosenbrock(100, pk)
        p = O.default_params(gradient_tolerance=1e-8, relative_decrease_tolerance=0, stepsize_tolerance=0,
                             preconditioned_gradient_tolerance=0, Delta_tolerance=0, max_iterations=500)
        r = R.tnt(pr, 0.1 * np.ones(100), p)
        rec = tnt_record(r)
        rec["x"] = lst(r["x"])
        out["jacobi" if pk else "plain"] = rec
        O.free(pr)
    json.dump(out, open(os.path.join(OUT, "tnt_rosenbrock100.json"), "w"), indent=1)

    # --- Stiefel Rayleigh quotient on a small grid (BASELINE cfg2 recipe at 8x7x6) ------------------
    nx, ny, nz, pcols = 8, 7, 6, 3
    n = nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    X0 = wl.random_stiefel(n, pcols, seed=20260928)
    pr = O.stiefel_rq(n, pcols, rowptr, col, val)
    p = O.default_params(gradient_tolerance=1e-8, relative_decrease_tolerance=0, stepsize_tolerance=0,
                         preconditioned_gradient_tolerance=0, Delta_tolerance=0, max_iterations=200,
                         max_TPCG_iterations=50)
    r = R.tnt(pr, X0.ravel(), p)
    rec = tnt_record(r)
    rec["grid"] = [nx, ny, nz]
    rec["p"] = pcols
    rec["seed"] = 20260928
    rec["x"] = lst(r["x"])
    rec["x0"] = lst(X0)
    json.dump(rec, open(os.path.join(OUT, "tnt_stiefel_8x7x6.json"), "w"), indent=1)
    O.free(pr)

    # --- SO(3)^N rotation averaging, N = 40, block-Jacobi (BASELINE cfg3 recipe, small) -------------
    N = 40
    ei, ej, Rt, w, Rtrue, Rinit = wl.pose_graph(N, seed=7)
    out = {}
    for pk in (0, 1):
        pr = O.so3n(N, ei, ej, Rt, w, precon_kind=pk)
        p = O.default_params(gradient_tolerance=1e-8, relative_decrease_tolerance=0, stepsize_tolerance=0,
                             preconditioned_gradient_tolerance=0, Delta_tolerance=0, max_iterations=100)
        r = R.tnt(pr, Rinit.ravel(), p)
        rec = tnt_record(r)
        rec["N"] = N
        rec["seed"] = 7
        rec["x"] = lst(r["x"])
        out["block_jacobi" if pk else "plain"] = rec
        O.free(pr)
    json.dump(out, open(os.path.join(OUT, "tnt_so3n_40.json"), "w"), indent=1)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
