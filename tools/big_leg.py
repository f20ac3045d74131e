#!/usr/bin/env python3
"""The beyond-cache leg of bench.py on its own: St(nx^3, 3), nx = 200 by default (8e6 rows, 1.38 GB working set against
the 256 MiB Infinity Cache), `steps` fused STPCG iterations with per-kernel HIP-event timings.  One JSON line.
Usage: python tools/big_leg.py [nx] [steps]      (PMC: PMC_CMD="python tools/big_leg.py 200 60" bash tools/pmc_bytes.sh out)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from optimization_amd import capi, workloads as wl  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 200
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
p, n = 3, nx ** 3
c = capi.Context(0)
rowptr, col, val = wl.laplacian_3d(nx, nx, nx)
A = c.csr(n, rowptr, col, val)
nnz = int(rowptr[-1])
del rowptr, col, val
prob = c.stiefel_rq(A, n, p)
X = c.upload(wl.stiefel_bench_iterate(nx, nx, nx, p, eps=1e-3, seed=7)[0])
g, H = prob.model(X)
s = c.vec(n * p)


def run(k):
    done = 0
    while done < k:
        r = c.stpcg(g, H, Delta=1e3, max_iterations=min(50, k - done), kappa_fgr=1e-12, theta=1.0, s_out=s)
        done += r["iterations"]


run(50)   # (a whole solve of the length the timed ones have: the first one of a length pays one-time allocations)
c.sync()
t0 = time.perf_counter()
run(steps)
c.sync()
dt = time.perf_counter() - t0
names = ("stiefel_hess_fused", "cg_update", "cg_pupdate")
for k in names:
    c.ktime_enable(k, True)
c.ktime_reset()
run(min(steps, 50))
per = {k: c.ktime_read(k) for k in names}
N = n * p
bytes_ = {"stiefel_hess_fused": 4 * nnz + 4 * (n + 1) + 32 * N, "cg_update": 24 * N, "cg_pupdate": 40 * N}
print(json.dumps({"rows": n, "us_per_step": 1e6 * dt / steps,
                  "kernels": {k: {"avg_us_event_pairs": 1e3 * v[1] / max(v[0], 1), "bytes": bytes_[k],
                                  "GBps": bytes_[k] / (1e3 * v[1] / max(v[0], 1) * 1e-6) / 1e9} for k, v in per.items()}}))
