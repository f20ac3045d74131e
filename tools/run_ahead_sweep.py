#!/usr/bin/env python3
"""cfg2 step time as a function of the speculative-enqueue depth (host throttle diagnostics)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) if "tools" in os.path.abspath(__file__) else os.getcwd()
sys.path.insert(0, os.getcwd())
import numpy as np
from optimization_amd import capi, workloads as wl
ctx = capi.Context(0)
nx = ny = nz = 100; p = 3; n = nx * ny * nz
rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
A = ctx.csr(n, rowptr, col, val)
prob = ctx.stiefel_rq(A, n, p)
Xb, _ = wl.stiefel_bench_iterate(nx, ny, nz, p, eps=1e-3, seed=7)
X = ctx.upload(Xb); g, H = prob.model(X); s_out = ctx.vec(n * p)
for ra in (1, 2, 3, 5, 8, 16, 3):
    for rep in range(2):
        ctx.sync(); t0 = time.perf_counter(); done = 0
        while done < 500:
            r = ctx.stpcg(g, H, Delta=1e3, max_iterations=50, kappa_fgr=1e-12, theta=1.0, s_out=s_out, run_ahead=ra)
            done += r["iterations"]
        ctx.sync(); dt = time.perf_counter() - t0
    print(ra, round(1e6 * dt / done, 2), "us/step")
