#!/usr/bin/env python3
"""Euclidean STPCG on a plain CSR Hessian (7-point Laplacian + 0.1 I, n = 1e6 x p): time per inner iteration."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from optimization_amd import capi, workloads as wl
ctx = capi.Context(0)
n = 100 ** 3
rowptr, col, val = wl.laplacian_3d(100, 100, 100)
A = ctx.csr(n, rowptr, col, val)
for p in (1, 3):
    H = ctx.op_csr(A, p)
    g = ctx.upload(np.random.default_rng(p).normal(size=n * p))
    r = ctx.stpcg(g, H, Delta=1e9, max_iterations=60, kappa_fgr=1e-14, theta=1.0)
    best = 1e9
    for rep in range(3):
        ctx.sync(); t0 = time.perf_counter()
        r = ctx.stpcg(g, H, Delta=1e9, max_iterations=60, kappa_fgr=1e-14, theta=1.0)
        ctx.sync(); best = min(best, time.perf_counter() - t0)
    print("p", p, "iterations", r["iterations"], "us/iteration", round(1e6 * best / r["iterations"], 1),
          "|s|", float(np.linalg.norm(r["s"].numpy())))
