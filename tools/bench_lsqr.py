#!/usr/bin/env python3
"""LSQR on the GPU: fused mi_lsqr vs the generic template loop on DeviceVector (n = 3e6 unknowns, nonsymmetric
tridiagonal operator, 400 passes).  Time of the LSQR call (measured inside the harness, device drained) at 401 passes minus 1 pass = per-pass cost."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, scipy.sparse as sps
import harness_py
n = 3_000_000
A = sps.diags([np.full(n - 1, -1.0), np.full(n, 2.02), np.full(n - 1, -0.99)], [-1, 0, 1], format="csr")  # slow to converge
b = A @ np.sin(np.arange(n) * 1e-3)
hd = harness_py.DeviceHarness()
out = {}
for mode, tag in ((0, "fused_mi_lsqr"), (1, "generic_template_loop")):
    import ctypes
    hd.L.hd_last_solve_seconds.restype = ctypes.c_double
    t = {}
    for iters in (1, 401):
        best = 1e9
        for rep in range(3):          # time of the LSQR call itself, measured inside the harness
            r = hd.lsqr_csr(A, b, btol=0.0, Atol=0.0, Acond_limit=1e300, max_iterations=iters, mode=mode)
            best = min(best, hd.L.hd_last_solve_seconds())
            assert r["iterations"] == iters, r["iterations"]
        t[iters] = best
    us = 1e6 * (t[401] - t[1]) / 400
    # Bytes a pass really MOVES (r06: r01-r05 priced the pass by 12-byte matrix entries it does not stream and printed
    # 8.3 TB/s, above the HBM peak).  The operator has three distinct values, so both SpMVs stream its value-indexed
    # packed copy: 4 B per entry + 4 B per row pointer.  Vector traffic of the fused pass (lsqr.hip): A v - alpha u folded
    # into the product reads v, u and writes u (3 n); u /= beta (2 n); A'u - beta v (3 n); v /= alpha with the partials
    # of <w,w>, <x,x>, <w,x> reads v, w, x and writes v (4 n); x += t1 w, w = v + t2 w (5 n) = 17 n doubles.  The generic
    # template loop makes one kernel per vector statement of the reference (IterativeSolvers.h:690-800): 2 products on
    # the plain 12-byte entries writing their results (2 x (12 nnz + 4 (n + 1) + 16 n)) and 31 n doubles of vector
    # statements and norms.
    if mode == 0:
        moved = 2 * (4 * A.nnz + 4 * (n + 1)) + 17 * 8 * n
    else:
        moved = 2 * (12 * A.nnz + 4 * (n + 1) + 16 * n) + 31 * 8 * n
    out[tag] = {"us_per_pass": us, "moved_bytes_per_pass": moved, "GBps_moved": moved / us / 1e3,
                "frac_of_8TBps": moved / us / 1e3 / 8000.0,
                # two matrices + u, v, w, x, b and a product's result
                "working_set_MB": (2 * ((4 if mode == 0 else 12) * A.nnz + 4 * (n + 1)) + 6 * 8 * n) / 1e6,
                "infinity_cache_MB": 268.4}
out["speedup"] = out["generic_template_loop"]["us_per_pass"] / out["fused_mi_lsqr"]["us_per_pass"]
print(json.dumps(out))
