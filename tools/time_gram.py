#!/usr/bin/env python3
"""The fused Gram pair of a cfg5 Rayleigh-Ritz step (S'A(S) and S'S, m = 126^3, ns = 72) timed with HIP event pairs:
the shared last tile column (r05) against the r04 form with a column of its own per Gram (NO_GRAM_HALF), in one process.
Usage: python tools/time_gram.py [m] [k]      (MI355OPT_LIB selects an experiment build)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from optimization_amd import capi  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 126 ** 3
k = int(sys.argv[2]) if len(sys.argv) > 2 else 72
c = capi.Context(0)
rng = np.random.default_rng(0)
S = c.upload(rng.normal(size=m * k))
AX = c.upload(rng.normal(size=m * 24))
AR = c.upload(rng.normal(size=m * (k - 24)))
out = {"m": m, "k": k, "lib": os.environ.get("MI355OPT_LIB", "default")}
c.ktime_enable("lobpcg_gram", True)
for _ in range(60):   # (an idle device needs tens of milliseconds of work before its clocks settle)
    c.lobpcg_gram_pair_sym(m, S, k, AX, 24, AR)
acc = {"shared_last_column": [], "r04_padded": []}
for rep in range(4):
    for name, opt in (("shared_last_column", 0), ("r04_padded", 1)):
        c.set_option("NO_GRAM_HALF", opt)
        c.lobpcg_gram_pair_sym(m, S, k, AX, 24, AR)
        c.ktime_reset()
        for _ in range(10):
            c.lobpcg_gram_pair_sym(m, S, k, AX, 24, AR)
        n, ms = c.ktime_read("lobpcg_gram")
        acc[name].append(round(1e3 * ms / n, 1))
for name, v in acc.items():
    out[name] = {"us_per_call_event_pairs": v, "GBps_best": 16 * m * k / min(v) / 1e3}
print(json.dumps(out))
