import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, oracle_py
from optimization_amd import capi, workloads as wl
O = oracle_py.Oracle()
for p, eps in ((8, 1e-6), (3, 1e-4)):
    nx, ny, nz = 128, 12, 10; n = nx*ny*nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz, shift=1e-3)
    Xb, _ = wl.stiefel_bench_iterate(nx, ny, nz, p, eps=eps, seed=7)
    op = O.stiefel_rq(n, p, rowptr, col, val); g = O.eval_grad(op, Xb.ravel())
    o = O.stpcg_problem(op, Xb.ravel(), g, 1e6, max_iterations=1000, kappa_fgr=1e-12, theta=1.0, trace_cap=1002)
    for K in (0, 50, 25, 10, 5):
        c = capi.Context(0); c.set_option("REANCHOR", K)
        A = c.csr(n, rowptr, col, val); prob = c.stiefel_rq(A, n, p); gd, H = prob.model(c.upload(Xb))
        r = c.stpcg(c.upload(g), H, Delta=1e6, max_iterations=1000, kappa_fgr=1e-12, theta=1.0, trace_cap=1002)
        k = min(len(r["trace"]["alpha"]), len(o["trace"]["alpha"]))
        e = np.abs(r["trace"]["alpha"][:k] / o["trace"]["alpha"][:k] - 1)
        idx = np.nonzero(e > 1e-6)[0]
        print(p, "K", K, "iterations", r["iterations"], "exit", r["exit_reason"], "holds", int(idx[0]) if idx.size else k, "oracle", o["iterations"], flush=True)
        del gd, H, prob, A; c.close()
