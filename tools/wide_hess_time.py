#!/usr/bin/env python3
"""Average time of the wide-row one-pass Hessian (k_st_hess_wide) inside 50-iteration solves on St(1e6,p), p = 5 and 8,
for the library named by MI355OPT_LIB -- the ablation builds (-DMI_WIDE_ABLATE_*, -DMI_ABLATE_GATHER_OWN: wrong
results, their launches never return early) are compared with the product build in one gpurun call.
Usage: python tools/wide_hess_time.py [p ...]"""
import sys, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from optimization_amd import capi, workloads as wl
nx = 100; n = nx ** 3
c = capi.Context(0)
A = c.csr(n, *wl.laplacian_3d(nx, nx, nx))
for p in ([int(a) for a in sys.argv[1:]] or [5, 6, 7, 8]):
    prob = c.stiefel_rq(A, n, p)
    X = c.upload(wl.stiefel_bench_iterate(nx, nx, nx, p, eps=1e-3, seed=7)[0])
    g, H = prob.model(X); s = c.vec(n * p)
    c.ktime_enable("stiefel_hess_fused", True)
    for _ in range(3): c.stpcg(g, H, Delta=1e3, max_iterations=50, kappa_fgr=1e-12, theta=1.0, s_out=s)
    c.ktime_reset()
    its = []
    for _ in range(4): its.append(c.stpcg(g, H, Delta=1e3, max_iterations=50, kappa_fgr=1e-12, theta=1.0, s_out=s)["iterations"])
    k, ms = c.ktime_read("stiefel_hess_fused")
    print(os.path.basename(os.environ.get("MI355OPT_LIB", "base")), "p", p, "us", round(1e3 * ms / k, 1), "launches", k, "iterations", its)
