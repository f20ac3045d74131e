#!/usr/bin/env python3
"""What re-anchoring the recurrence form costs: cfg2 at full size, solves of 200 iterations at the
bench iterate with MI355OPT_REANCHOR = 0 (never), 50 (default), 25, 10 -- microseconds per iteration, same process,
alternating.  One JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from optimization_amd import capi, workloads as wl
nx, p = 100, 3
n = nx ** 3
rowptr, col, val = wl.laplacian_3d(nx, nx, nx)
Xb, _ = wl.stiefel_bench_iterate(nx, nx, nx, p, eps=1e-3, seed=7)
ctx = capi.Context(0)
A = ctx.csr(n, rowptr, col, val)
prob = ctx.stiefel_rq(A, n, p)
g, H = prob.model(ctx.upload(Xb))
s_out = ctx.vec(n * p)
kw = dict(Delta=1e3, max_iterations=200, kappa_fgr=1e-12, theta=1.0, s_out=s_out)
out = {"workload": f"cfg2 St({n},{p}), solves of 200 iterations", "us_per_iteration": {}}
for _ in range(5):
    r0 = ctx.stpcg(g, H, **kw)
for rep in range(3):
    for K in (0, 50, 25, 10):
        ctx.set_option("REANCHOR", K)
        ctx.sync()
        t0 = time.perf_counter()
        it = 0
        for _ in range(10):
            it += ctx.stpcg(g, H, **kw)["iterations"]
        ctx.sync()
        out["us_per_iteration"].setdefault(str(K), []).append(1e6 * (time.perf_counter() - t0) / it)
out["iterations_per_solve"] = r0["iterations"]
print(json.dumps(out))
