#!/usr/bin/env python3
"""cfg5 LOBPCG run (DeviceMatrix template path) for rocprofv3 --kernel-trace --stats."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from optimization_amd import workloads as wl
import harness_py
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rowptr, col, val = wl.laplacian_3d(126, 126, 126)
hz = harness_py.DeviceHarness()
t0 = time.perf_counter()
r = hz.lobpcg(126 ** 3, 24, 20, csr=(rowptr, col, val), X0=None, max_iters=iters, tau=1e-12)
print("wall", time.perf_counter() - t0, "iters", r["num_iters"], "nc", r["nc"])
