#!/usr/bin/env python3
"""Error-vs-iteration curve of the cfg2 bench solve: for k = 1 ... 50 the step s_k of a k-iteration
STPCG solve at bench.py's operating point (St(1e6,3), Delta 1e3, kappa_fgr 1e-12, theta 1)

    gpu      the fused device solve (default matrix format)                         vs the reference
    floor    the SAME reference algorithm with its sums re-associated (oracle/liboracle_omp.so, 4 threads) vs the reference

reference = the plain-C oracle (bit for bit the reference's templates: tests/test_cpu_oracle_templates.py).  Also the
relative error of alpha_k / beta_k (traces of the 50-iteration solve) and the residual reduction <r,v>_k / <r,v>_0, i.e. how
deep the solve is when the error crosses 1e-10.  CHECKER script (uses oracle/): prints one JSON record and a table.
Usage: python tools/parity_curve.py [kmax] [grid] [p]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
import numpy as np  # noqa: E402

import oracle_py  # noqa: E402
from optimization_amd import capi, workloads as wl  # noqa: E402

kmax = int(sys.argv[1]) if len(sys.argv) > 1 else 50
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 100
p = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n = nx ** 3
PRM = dict(Delta=1e3, kappa_fgr=1e-12, theta=1.0)


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


rowptr, col, val = wl.laplacian_3d(nx, nx, nx)
Xb, modes = wl.stiefel_bench_iterate(nx, nx, nx, p, eps=1e-3, seed=7)
O = oracle_py.Oracle()
M = oracle_py.Oracle(omp=True)
M.set_threads(4)
op, mp = O.stiefel_rq(n, p, rowptr, col, val), M.stiefel_rq(n, p, rowptr, col, val)
g = O.eval_grad(op, Xb.ravel())
gm = M.eval_grad(mp, Xb.ravel())
ctx = capi.Context(0)
A = ctx.csr(n, rowptr, col, val)
prob = ctx.stiefel_rq(A, n, p)
X = ctx.upload(Xb)
gd, H = prob.model(X)

gsame = ctx.upload(g)   # the reference's gradient, bit for bit, as the device solve's input
full_o = O.stpcg_problem(op, Xb.ravel(), g, max_iterations=kmax, trace_cap=kmax + 2, **PRM)
full_m = M.stpcg_problem(mp, Xb.ravel(), gm, max_iterations=kmax, trace_cap=kmax + 2, **PRM)
full_g = ctx.stpcg(gd, H, max_iterations=kmax, trace_cap=kmax + 2, **PRM)
rv0 = float(np.dot(g, g))
rows = []
for k in range(1, kmax + 1):
    o = O.stpcg_problem(op, Xb.ravel(), g, max_iterations=k, **PRM)
    m = M.stpcg_problem(mp, Xb.ravel(), gm, max_iterations=k, **PRM)
    r = ctx.stpcg(gd, H, max_iterations=k, **PRM)
    rs = ctx.stpcg(gsame, H, max_iterations=k, **PRM)
    ms = M.stpcg_problem(mp, Xb.ravel(), g, max_iterations=k, **PRM)   # (gm was evaluated last at Xb: model state valid)
    assert r["iterations"] == o["iterations"] == k, (k, r["iterations"], o["iterations"])
    rows.append(dict(k=k, s_rel_gpu=rel(r["s"].numpy(), o["s"]), s_rel_floor=rel(m["s"], o["s"]),
                     s_rel_gpu_same_g=rel(rs["s"].numpy(), o["s"]), s_rel_floor_same_g=rel(ms["s"], o["s"]),
                     alpha_rel_gpu=abs(full_g["trace"]["alpha"][k - 1] / full_o["trace"]["alpha"][k - 1] - 1),
                     alpha_rel_floor=abs(full_m["trace"]["alpha"][k - 1] / full_o["trace"]["alpha"][k - 1] - 1),
                     beta_rel_gpu=abs(full_g["trace"]["beta"][k - 1] / full_o["trace"]["beta"][k - 1] - 1),
                     beta_rel_floor=abs(full_m["trace"]["beta"][k - 1] / full_o["trace"]["beta"][k - 1] - 1),
                     residual_reduction=float(full_o["trace"]["rv"][k - 1] / rv0) ** 0.5))
out = dict(g_rel_device_vs_reference=rel(gd.numpy(), g), g_rel_reassociated_vs_reference=rel(gm, g), workload=f"cfg2 St({n},{p}) bench solve, modes {modes}", params=PRM, device=ctx.device_name(), rows=rows,
           first_k_with_gpu_above_1e10=next((r_["k"] for r_ in rows if r_["s_rel_gpu"] > 1e-10), None),
           first_k_with_floor_above_3e11=next((r_["k"] for r_ in rows if r_["s_rel_floor"] > 3e-11), None))
print(json.dumps(out))
print("| k | s: gpu vs ref | s: re-associated ref vs ref | s: gpu, SAME g bits | s: re-assoc. ref, SAME g bits | alpha gpu | alpha floor | beta gpu | beta floor | |r_k|/|r_0| |",
      file=sys.stderr)
print("|---|---|---|---|---|---|---|---|---|---|", file=sys.stderr)
for r_ in rows:
    print("| {k} | {s_rel_gpu:.2e} | {s_rel_floor:.2e} | {s_rel_gpu_same_g:.2e} | {s_rel_floor_same_g:.2e} | {alpha_rel_gpu:.1e} | {alpha_rel_floor:.1e} | {beta_rel_gpu:.1e} | "
          "{beta_rel_floor:.1e} | {residual_reduction:.2e} |".format(**r_), file=sys.stderr)
O.free(op)
M.free(mp)
ctx.close()
