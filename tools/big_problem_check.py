#!/usr/bin/env python3
"""One-pass vs two-pass Stiefel Hessian inside STPCG on an nx^3 grid (default 200^3 = 8e6 rows, beyond the Infinity Cache): agreement of the iterates and time per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from optimization_amd import capi, workloads as wl
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 200
p = 3; n = nx ** 3
rowptr, col, val = wl.laplacian_3d(nx, nx, nx)
Xb, _ = wl.stiefel_bench_iterate(nx, nx, nx, p, eps=1e-3, seed=7)
res = {}
for mode in ("one-pass", "two-pass"):
    os.environ["MI355OPT_NO_DIRGRAM"] = "1" if mode == "two-pass" else "0"
    c = capi.Context(0)
    A = c.csr(n, rowptr, col, val)
    prob = c.stiefel_rq(A, n, p)
    g, H = prob.model(c.upload(Xb))
    r = c.stpcg(g, H, Delta=1e3, max_iterations=12, kappa_fgr=1e-12, theta=1.0, trace_cap=16)
    c.sync(); t0 = time.perf_counter()
    r2 = c.stpcg(g, H, Delta=1e3, max_iterations=50, kappa_fgr=1e-12, theta=1.0)
    c.sync(); dt = time.perf_counter() - t0
    res[mode] = (r["s"].numpy().copy(), r["iterations"], r["trace"]["alpha"][:3])
    print(mode, "n", n, "iters", r["iterations"], r2["iterations"], "us/step", round(1e6 * dt / max(r2["iterations"], 1), 1))
    c.close()
a, b = res["one-pass"][0], res["two-pass"][0]
print("rel diff one-pass vs two-pass:", float(np.linalg.norm(a - b) / np.linalg.norm(b)))
