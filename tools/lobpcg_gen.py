#!/usr/bin/env python3
"""The generalized eigenproblem through the drop-in LOBPCG (a B operator, LOBPCG.h:131-140): per-iteration time next to
the same run without B, both with plain callables (diagonal operators of the reference's own tests,
tests/LOBPCG_unit_test.cpp:40-74, at cfg5's size) -- the basis goes to the Gram / update kernels as column blocks in
both (r05).  Under `rocprofv3 --kernel-trace` the trace shows whether any copy kernel is left.
Usage: python tools/lobpcg_gen.py [m] [iters]"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np  # noqa: E402

import harness_py  # noqa: E402


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 126 ** 3
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    nx, nev = 24, 20
    a = np.linspace(1.0, 1000.0, m)
    b = 1.0 + 0.5 * np.sin(np.arange(m))
    hz = harness_py.DeviceHarness()
    hz.L.hd_lobpcg_seconds_per_iteration.restype = ctypes.c_double
    X0 = np.random.default_rng(3).uniform(-1, 1, size=(m, nx))
    out = {"m": m, "nx": nx, "iterations": iters}
    for name, B in (("standard (B absent)", None), ("generalized (B = diag)", b)):
        r = hz.lobpcg(m, nx, nev, Adiag=a, Bdiag=B, X0=X0, max_iters=iters, tau=1e-14)
        assert r["rc"] == 0, r["err"]
        out[name] = {"ms_per_iteration": 1e3 * hz.L.hd_lobpcg_seconds_per_iteration(), "nc": int(r["nc"]),
                     "theta0": float(r["Theta"][0])}
    out["ratio"] = out["generalized (B = diag)"]["ms_per_iteration"] / out["standard (B absent)"]["ms_per_iteration"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
