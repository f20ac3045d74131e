#!/usr/bin/env python3
"""LSQR on the GPU: fused mi_lsqr vs the generic template loop on DeviceVector (n = 3e6 unknowns, nonsymmetric
tridiagonal operator, 100 passes).  Wall time of the whole harness call minus a 1-pass call = per-pass cost."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, scipy.sparse as sps
import harness_py
n = 3_000_000
A = sps.diags([np.full(n - 1, -1.0), np.full(n, 3.0), np.full(n - 1, 1.5)], [-1, 0, 1], format="csr")
b = A @ np.sin(np.arange(n) * 1e-3)
hd = harness_py.DeviceHarness()
out = {}
for mode, tag in ((0, "fused_mi_lsqr"), (1, "generic_template_loop")):
    t = {}
    for iters in (1, 101):
        hd.lsqr_csr(A, b, btol=0.0, Atol=0.0, Acond_limit=1e300, max_iterations=iters, mode=mode)  # warm
        t0 = time.perf_counter()
        r = hd.lsqr_csr(A, b, btol=0.0, Atol=0.0, Acond_limit=1e300, max_iterations=iters, mode=mode)
        t[iters] = time.perf_counter() - t0
        assert r["iterations"] == iters, r
    us = 1e6 * (t[101] - t[1]) / 100
    # per pass: 2 SpMV (12 nnz + 16 n bytes each) + 17 n * 8 bytes of vector traffic (lsqr.hip)
    bytes_pass = 2 * (12 * A.nnz + 16 * n) + 17 * 8 * n
    out[tag] = {"us_per_pass": us, "GBps_algorithmic": bytes_pass / us / 1e3}
out["speedup"] = out["generic_template_loop"]["us_per_pass"] / out["fused_mi_lsqr"]["us_per_pass"]
print(json.dumps(out))
