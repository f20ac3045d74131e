#!/usr/bin/env python3
"""cfg3 outer-loop pieces through the C ABI: model assembly (mi_so3n_model), objective, retraction, fused trial step --
us per call at N = 5e5 (host wall time around 20 calls, device drained).  Usage: python tools/time_so3_model.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from optimization_amd import capi, workloads as wl
N = 500_000
ei, ej, Rt, w, Rtrue, Rinit = wl.pose_graph(N, seed=7, init_sigma=0.02)
c = capi.Context(0)
prob = c.so3n(N, ei, ej, Rt, w)
R = c.upload(Rinit)
g, H, P = prob.model(R)
h = c.upload(np.random.default_rng(0).normal(size=3 * N) * 1e-3)


def timed(fn, reps=20):
    fn(); c.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    c.sync()
    return 1e6 * (time.perf_counter() - t0) / reps


out = {"model_us": timed(lambda: prob.model(R)), "objective_us": timed(lambda: prob.objective(R)),
       "retract_us": timed(lambda: prob.retract(R, h)), "hvp_us": timed(lambda: H.apply(h)),
       "fused_trial_us": timed(lambda: prob.trial(R, h, g, with_precon=True))}
print(json.dumps(out))
c.close()
