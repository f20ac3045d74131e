#!/usr/bin/env python3
"""The opt-in two-kernel cfg2 step (MI355OPT_TWO_KERNEL_STEP, DESIGN 3.1 / EXPERIMENTS r05) against the default three-kernel
step on the bench problem: us per inner iteration (wall, 500 steps, alternating in one process) and per-kernel event timings.
The distance from the reference it costs is measured by tests/test_gpu_cfg2_full.py::test_two_kernel_step_experiment."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from optimization_amd import capi, workloads as wl
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 100
p, n = 3, nx ** 3
c = capi.Context(0)
A = c.csr(n, *wl.laplacian_3d(nx, nx, nx))
prob = c.stiefel_rq(A, n, p)
X = c.upload(wl.stiefel_bench_iterate(nx, nx, nx, p, eps=1e-3, seed=7)[0])
g, H = prob.model(X)
s = c.vec(n * p)


def run(k):
    done = 0
    while done < k:
        r = c.stpcg(g, H, Delta=1e3, max_iterations=min(50, k - done), kappa_fgr=1e-12, theta=1.0, s_out=s)
        assert r["iterations"] > 0
        done += r["iterations"]


run(1500)
out = {"rows": n, "us_per_step": {"three_kernels": [], "two_kernels": []}, "kernels_us": {}}
for rep in range(4):
    for name, opt in (("three_kernels", 0), ("two_kernels", 1)):
        c.set_option("TWO_KERNEL_STEP", opt)
        run(100)
        c.sync()
        t0 = time.perf_counter()
        run(500)
        c.sync()
        out["us_per_step"][name].append(round(1e6 * (time.perf_counter() - t0) / 500, 2))
for name, opt in (("three_kernels", 0), ("two_kernels", 1)):
    c.set_option("TWO_KERNEL_STEP", opt)
    for k in ("stiefel_hess_fused", "cg_update", "cg_pupdate"):
        c.ktime_enable(k, True)
    c.ktime_reset()
    run(200)
    out["kernels_us"][name] = {k: round(1e3 * c.ktime_read(k)[1] / max(c.ktime_read(k)[0], 1), 2)
                               for k in ("stiefel_hess_fused", "cg_update", "cg_pupdate") if c.ktime_read(k)[0]}
print(json.dumps(out))
