#!/usr/bin/env python3
"""Whole-TNT timing through the drop-in template path on DeviceVector (the C++ header layer): cfg2 (Stiefel(1e6,3)) and
cfg3 (SO(3)^N, N = 5e5, 3x3 block-Jacobi preconditioner): how much of a run is the fused inner loop (bench.py's metric)
and how much the outer loop around it -- wall time of the TNT call on a microsecond clock divided by the inner
iterations it made.  One JSON line.  Usage: python tools/bench_tnt.py [eps] [cfg2|cfg3|both]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from optimization_amd import workloads as wl
import oracle_py as op
import harness_py

nx = ny = nz = 100
n, p = nx * ny * nz, 3
rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
X0, _ = wl.stiefel_bench_iterate(nx, ny, nz, p, eps=float(sys.argv[1]) if len(sys.argv) > 1 else 1e-2, seed=7)
hz = harness_py.DeviceHarness()
O = op.Oracle()   # only for the default parameter struct (TNTParams defaults, TNT.h:76-128)
hz.L.hd_last_tnt_seconds.restype = ctypes.c_double
hz.L.hd_last_tnt_wall_seconds.restype = ctypes.c_double
hz.L.hd_last_tnt_syncs.restype = ctypes.c_size_t
which = sys.argv[2] if len(sys.argv) > 2 else "both"
hz.L.hd_set_tnt_repeats(2)   # every run twice in its context, the second one timed (pool filled, kernels loaded)
out = {}


def record(tag, r):
    secs = hz.L.hd_last_tnt_wall_seconds()
    inner = int(np.sum(r["inner_iterations"]))
    out[tag] = {"seconds": secs, "outer": int(r["outer_iterations"]), "inner_total": inner,
                "ms_per_outer": 1e3 * secs / max(1, int(r["outer_iterations"])),
                "us_per_inner_if_all_time_were_inner": 1e6 * secs / max(1, inner), "f": float(r["f"]),
                "status": int(r["status"]), "host_syncs": int(hz.L.hd_last_tnt_syncs()),
                "host_syncs_per_outer": hz.L.hd_last_tnt_syncs() / max(1, int(r["outer_iterations"]))}


if which in ("cfg3", "both"):
    N = 500_000
    ei, ej, Rt, w, Rtrue, Rinit = wl.pose_graph(N, seed=7, init_sigma=0.02)  # near the optimum: PSD Hessians, full inner solves
    for tag, kw, flags in (("cfg3_warmup", dict(max_iterations=2), 1), ("cfg3_run", dict(max_iterations=12), 1),
                           ("cfg3_run_without_fused_trial_step", dict(max_iterations=12), 3)):
        prm = O.default_params(max_TPCG_iterations=50, gradient_tolerance=1e-12, relative_decrease_tolerance=0.0,
                               stepsize_tolerance=0.0, preconditioned_gradient_tolerance=0.0, **kw)
        record(tag, hz.tnt_so3n(N, ei, ej, Rt, w, Rinit, prm, flags))
for tag, kw, mode in (("warmup", dict(max_iterations=2), 0), ("run", dict(max_iterations=12), 0),
                      ("run_without_fused_trial_step", dict(max_iterations=12), 2)):
    prm = O.default_params(max_TPCG_iterations=50, gradient_tolerance=1e-12, relative_decrease_tolerance=0.0,
                           stepsize_tolerance=0.0, preconditioned_gradient_tolerance=0.0, **kw)
    if which == "cfg3":
        break
    r = hz.tnt_stiefel(n, p, rowptr, col, val, X0, prm, mode)
    record(tag, r)
print(json.dumps(out))
