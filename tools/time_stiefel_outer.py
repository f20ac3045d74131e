import sys, os, time, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
from optimization_amd import capi, workloads as wl
nx, p = 100, int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = nx ** 3
c = capi.Context(0)
A = c.csr(n, *wl.laplacian_3d(nx, nx, nx))
prob = c.stiefel_rq(A, n, p)
X = c.upload(wl.stiefel_bench_iterate(nx, nx, nx, p, eps=1e-2, seed=7)[0])
def timed(fn, reps=10):
    fn(); c.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    c.sync(); return 1e6 * (time.perf_counter() - t0) / reps
g, H = prob.model(X)
h = c.stiefel_project(n, p, X, c.upload(np.random.default_rng(4).normal(size=(n, p)) * 1e-3))
out = {"p": p, "model_us": timed(lambda: prob.model(X)), "objective_us": timed(lambda: prob.objective(X)),
       "retract_us": timed(lambda: c.stiefel_retract(n, p, X, h)), "two_pass_hvp_us": timed(lambda: H.apply(h)),
       "trial_us": timed(lambda: prob.trial(X, h, g))}
print(json.dumps(out))
