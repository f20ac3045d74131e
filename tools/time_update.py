#!/usr/bin/env python3
"""The 72 x 48 Ritz update of a cfg5 iteration (X and P in one pass, mi_lobpcg_update2 on the matrix pipe) by HIP event
pairs: 32-row blocks with 16 bytes per lane (r05) against the r04 form (NO_UPDATE_PAIR), alternating in one process."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from optimization_amd import capi
m, ks, nx = 126 ** 3, int(sys.argv[1]) if len(sys.argv) > 1 else 72, 24
c = capi.Context(0)
rng = np.random.default_rng(0)
S = c.upload(rng.normal(size=m * ks))
Cm = rng.normal(size=(ks, 2 * nx))
c.ktime_enable("lobpcg_update", True)
for _ in range(40):
    c.lobpcg_update2(m, S, ks, Cm, nx)
acc = {"paired_rows_16B": [], "r04_8B": []}
for rep in range(4):
    for name, opt in (("paired_rows_16B", 0), ("r04_8B", 1)):
        c.set_option("NO_UPDATE_PAIR", opt)
        c.lobpcg_update2(m, S, ks, Cm, nx)
        c.ktime_reset()
        for _ in range(10):
            c.lobpcg_update2(m, S, ks, Cm, nx)
        n, ms = c.ktime_read("lobpcg_update")
        acc[name].append(round(1e3 * ms / n, 1))
print(json.dumps({"m": m, "ks": ks, "us_per_call_event_pairs": acc,
                  "GBps_best": {k: 8 * m * (ks + 2 * nx) / min(v) / 1e3 for k, v in acc.items()}}))
