"""Full-size golden fixture for BASELINE cfg2 (St(1e6,3), 100^3 grid): what the REAL reference
(oracle/_ref/libref.so = the templates of /root/reference compiled in this container) returns

  (a) for the exact solve bench.py times -- STPCG, 50 iterations, kappa_fgr 1e-12, theta 1, Delta 1e3 at the
      near-optimal bench iterate (reference IterativeSolvers.h:166-426);
  (b) for a whole TNT run from a random start (reference Riemannian/TNT.h:242-689), max_TPCG_iterations = 50.

Only scalars travel: per-iteration traces, counts, the accept sequence and checksums of the 24 MB vectors (sums,
projections on analytic eigenvectors, SHA-256 of the bytes).  On THIS machine the plain-C oracle reproduces the
hashes (tests/test_cpu_oracle_templates.py::test_oracle_is_the_reference_on_the_full_size_bench_solve: oracle ==
reference bit for bit at full size); on the GPU box, whose numpy generates the inputs with other last bits, the
tests re-run the oracle on the same arrays as the device, hold it against the scalars recorded here with a
tolerance and compare the device results with its vectors (tests/test_gpu_cfg2_full.py).

Run here (where /root/reference is mounted; ~4 minutes):  python tests/golden/make_golden_full.py
"""
import hashlib
import json
import os
import sys

# (the inputs come out of numpy's QR: its BLAS must run with the thread count the tests use -- tests/conftest.py --
# or the last bits of X differ)
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ[_v] = "1"

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py  # noqa: E402
from optimization_amd import workloads as wl  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

NX = NY = NZ = 100
P = 3
BENCH = dict(Delta=1e3, max_iterations=50, kappa_fgr=1e-12, theta=1.0)  # bench.py run_steps()
TNT = dict(gradient_tolerance=1e-5, relative_decrease_tolerance=0.0, stepsize_tolerance=0.0,
           preconditioned_gradient_tolerance=0.0, Delta_tolerance=0.0, max_iterations=40, max_TPCG_iterations=50)
TNT_SEED = 20260928
MODES = [(1, 1, 1), (1, 1, 2), (1, 2, 1), (2, 1, 1), (1, 2, 2), (2, 1, 2), (2, 2, 1), (1, 1, 3)]


def lst(a):
    return [float(x) for x in np.asarray(a).ravel()]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()


def vector_checksums(v, n, p):
    """sums + projections of the n x p field on the analytic eigenvectors of the grid operator + SHA-256"""
    V = np.asarray(v).reshape(n, p)
    E = np.stack([wl.laplacian_3d_eigvec(NX, NY, NZ, *m)[0] for m in MODES], axis=1)
    return dict(sum=float(V.sum()), abs_sum=float(np.abs(V).sum()), sq_sum=float((V * V).sum()),
                eig_projections=lst(E.T @ V), sha256=sha(V))


def main():
    O = oracle_py.Oracle()
    R = oracle_py.Reference()
    n = NX * NY * NZ
    rowptr, col, val = wl.laplacian_3d(NX, NY, NZ)
    prob = O.stiefel_rq(n, P, rowptr, col, val)
    out = dict(grid=[NX, NY, NZ], p=P, modes=MODES)

    # (a) the bench solve
    Xb, modes = wl.stiefel_bench_iterate(NX, NY, NZ, P, eps=1e-3, seed=7)
    g = O.eval_grad(prob, Xb.ravel())
    r = O.stpcg_problem(prob, Xb.ravel(), g, BENCH["Delta"], max_iterations=BENCH["max_iterations"],
                        kappa_fgr=BENCH["kappa_fgr"], theta=BENCH["theta"], trace_cap=64, lib=R)
    # the reference's observer (STPCGUserFunction, IterativeSolvers.h:50-59) sees alpha_k only; beta, kappa and <r,v>
    # are taken from the plain-C oracle's run, which must reproduce the reference's alphas and step bit for bit
    o = O.stpcg_problem(prob, Xb.ravel(), g, BENCH["Delta"], max_iterations=BENCH["max_iterations"],
                        kappa_fgr=BENCH["kappa_fgr"], theta=BENCH["theta"], trace_cap=64)
    assert list(o["trace"]["alpha"]) == list(r["trace"]["alpha"]) and np.array_equal(o["s"], r["s"])
    assert o["M_norm"] == r["M_norm"] and o["iterations"] == r["iterations"]
    out["bench_stpcg"] = dict(params=BENCH, iterate=dict(eps=1e-3, seed=7, modes=[list(m) for m in modes]),
                              iterations=int(r["iterations"]),  # (the reference reports no exit reason)
                              M_norm=float(r["M_norm"]),
                              trace=dict(alpha=lst(r["trace"]["alpha"]), beta=lst(o["trace"]["beta"]),
                                         kappa=lst(o["trace"]["kappa"]), rv=lst(o["trace"]["rv"])),
                              g=vector_checksums(g, n, P), s=vector_checksums(r["s"], n, P))
    print("bench solve:", r["iterations"], "iterations, |s|_M", r["M_norm"])

    # (b) a whole TNT run
    X0 = wl.random_stiefel(n, P, seed=TNT_SEED)
    prm = O.default_params(**TNT)
    t = R.tnt(prob, X0.ravel(), prm)
    rec = dict(status=int(t["status"]), status_name=t["status_name"], outer_iterations=int(t["outer_iterations"]),
               accepted=int(t["accepted"]), inner_iterations=[int(x) for x in t["inner_iterations"]],
               objective_values=lst(t["objective_values"]), gradient_norms=lst(t["gradient_norms"]),
               trust_region_radius=lst(t["trust_region_radius"]), gain_ratios=lst(t["gain_ratios"]),
               update_step_M_norms=lst(t["update_step_M_norms"]), update_step_norms=lst(t["update_step_norms"]),
               f=float(t["f"]), gradfx_norm=float(t["gradfx_norm"]),
               calls={k: int(v) for k, v in t["calls"].items()})
    rec["params"] = TNT
    rec["seed"] = TNT_SEED
    rec["x0"] = vector_checksums(X0, n, P)
    rec["x"] = vector_checksums(t["x"], n, P)
    out["tnt"] = rec
    print("TNT:", rec["status_name"], rec["outer_iterations"], "outer,", sum(rec["inner_iterations"]), "inner, f", rec["f"])
    O.free(prob)
    json.dump(out, open(os.path.join(OUT, "cfg2_full.json"), "w"), indent=1)
    print("written", os.path.join(OUT, "cfg2_full.json"))


if __name__ == "__main__":
    main()
