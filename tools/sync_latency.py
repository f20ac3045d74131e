#!/usr/bin/env python3
"""What one host read-back costs: a 64-element dot product (one tiny kernel + the slot reduction) followed by the read of
its slot, 500 times -- us per round trip, with the polled pinned flag (stream_wait / read_slots_sync, default) and with
MI355OPT_NO_POLLED_SYNC=1 (copy + hipStreamSynchronize).  Usage: python tools/sync_latency.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from optimization_amd import capi

out = {}
for tag, opt in (("polled", 0), ("stream_synchronize", 1), ("polled_again", 0)):
    c = capi.Context(0)
    c.set_option("NO_POLLED_SYNC", opt)
    a = c.upload(np.ones(64))
    for _ in range(50):
        a.dot(a)
    t0 = time.perf_counter()
    for _ in range(500):
        a.dot(a)
    out[tag] = {"us_per_dot_and_read": 1e6 * (time.perf_counter() - t0) / 500}
    c.close()
print(json.dumps(out))
