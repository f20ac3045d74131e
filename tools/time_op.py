#!/usr/bin/env python3
"""Stand-alone timing of the fused Stiefel Hessian pass (mi_debug_time_fused_apply) on an nx^3 grid, for A/B runs of
kernel variants inside ONE gpurun call:  python tools/time_op.py [nx] [reps]   -> one line."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from optimization_amd import capi, workloads as wl
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 100
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
p = 3; n = nx ** 3
rowptr, col, val = wl.laplacian_3d(nx, nx, nx)
Xb, _ = wl.stiefel_bench_iterate(nx, nx, nx, p, eps=1e-3, seed=7)
c = capi.Context(0)
A = c.csr(n, rowptr, col, val)
prob = c.stiefel_rq(A, n, p)
X = c.upload(Xb)
g, H = prob.model(X)
out = c.vec(n * p)
ts = [H.time_fused_apply(g, out, reps) for _ in range(3)]
nnz = int(rowptr[-1])
pk = os.environ.get("MI355OPT_NO_PACKED", "0") != "1"
byts = (4 if pk else 12) * nnz + 4 * (n + 1) + 8 * 4 * n * p
print("%-60s us %s  -> %.0f GB/s on %.1f MB" % (" ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("MI355OPT_")) or "-",
      " ".join("%.2f" % t for t in ts), byts / min(ts) / 1e3, byts / 1e6))
c.close()
