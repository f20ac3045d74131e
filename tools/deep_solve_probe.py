#!/usr/bin/env python3
"""Long and deep STPCG solves on an ill-conditioned Riemannian Hessian: the one-pass Stiefel Hessian
whose projection matrix G(p) is carried by scalar RECURRENCES (default without a preconditioner) against the two-pass
operator (MI355OPT_NO_DIRGRAM=1: projection from the product itself), the direct form (MI355OPT_DIRGRAM_DIRECT=1) and
the CPU oracle, with the re-associated oracle as the conditioning floor.  Per (grid, p, max_iterations): iteration count,
exit, alpha / beta traces over the WHOLE solve, the step, and the tangency sym(X' s) of the step (the non-tangent
component an inexact projection would leave behind).  CHECKER script.  One JSON line per case.
Usage: python tools/deep_solve_probe.py [quick]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np  # noqa: E402

import oracle_py  # noqa: E402
from optimization_amd import capi, workloads as wl  # noqa: E402

MODES = {"recurrence": {}, "two-pass": {"MI355OPT_NO_DIRGRAM": "1"}, "direct": {"MI355OPT_DIRGRAM_DIRECT": "1"}}
EXTRA = [m for m in sys.argv[1:] if m.startswith("MI355OPT_")]   # e.g. MI355OPT_REANCHOR=50 -> mode "recurrence+..."


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def tangency(X, s, n, p):
    S = s.reshape(n, p)
    G = X.T @ S
    return float(np.linalg.norm(G + G.T) / 2 / max(np.linalg.norm(S), 1e-300))


def trace_cmp(a, b):
    k = min(len(a), len(b))
    e = np.abs(np.asarray(a[:k]) / np.asarray(b[:k]) - 1)
    first = next((int(i) for i in range(k) if e[i] > 1e-6), None)
    return dict(max=float(e.max()) if k else 0.0, first_above_1e6=first,
                at_quarter=[float(e[min(k - 1, int(q * k))]) for q in (0.25, 0.5, 0.75, 1.0)] if k else [])


def case(grid, p, shift, maxits, eps, O, M):
    nx, ny, nz = grid
    n = nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz, shift=shift)
    Xb, modes = wl.stiefel_bench_iterate(nx, ny, nz, p, eps=eps, seed=7)
    op, mp = O.stiefel_rq(n, p, rowptr, col, val), M.stiefel_rq(n, p, rowptr, col, val)
    g, gm = O.eval_grad(op, Xb.ravel()), M.eval_grad(mp, Xb.ravel())
    for maxit in maxits:
        kw = dict(Delta=1e6, max_iterations=maxit, kappa_fgr=1e-12, theta=1.0)
        o = O.stpcg_problem(op, Xb.ravel(), g, trace_cap=maxit + 2, **kw)
        m = M.stpcg_problem(mp, Xb.ravel(), gm, trace_cap=maxit + 2, **kw)
        rec = dict(grid=grid, p=p, shift=shift, max_iterations=maxit, n=n,
                   oracle=dict(iterations=o["iterations"], exit=o["exit_reason"],
                               reduction=float((o["trace"]["rv"][-1] / np.dot(g, g)) ** 0.5), tangency_s=tangency(Xb, o["s"], n, p)),
                   floor=dict(iterations=m["iterations"], exit=m["exit_reason"], s_rel=rel(m["s"], o["s"]),
                              alpha=trace_cmp(m["trace"]["alpha"], o["trace"]["alpha"]),
                              beta=trace_cmp(m["trace"]["beta"], o["trace"]["beta"])))
        modes_ = dict(MODES)
        if EXTRA:
            modes_["recurrence+" + ",".join(EXTRA)] = dict(e.split("=") for e in EXTRA)
        for mode, env in modes_.items():
            if p > 4 and mode == "direct":
                continue
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                c = capi.Context(0)
                A = c.csr(n, rowptr, col, val)
                prob = c.stiefel_rq(A, n, p)
                gd, H = prob.model(c.upload(Xb))
                r = c.stpcg(gd, H, trace_cap=maxit + 2, **kw)
                s = r["s"].numpy()
                rec[mode] = dict(iterations=r["iterations"], exit=r["exit_reason"], s_rel=rel(s, o["s"]),
                                 M_norm_rel=abs(r["M_norm"] / o["M_norm"] - 1), tangency_s=tangency(Xb, s, n, p),
                                 alpha=trace_cmp(r["trace"]["alpha"], o["trace"]["alpha"]),
                                 beta=trace_cmp(r["trace"]["beta"], o["trace"]["beta"]))
                del gd, H, prob, A
                c.close()
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
        print(json.dumps(rec), flush=True)
    O.free(op)
    M.free(mp)


def main():
    O = oracle_py.Oracle()
    M = oracle_py.Oracle(omp=True)
    M.set_threads(4)
    quick = "quick" in sys.argv
    # long thin grids: the gap between the p-th and (p+1)-th mode is ~ (2 k + 1) pi^2 / (nx + 1)^2, the Hessian's
    # condition number (lambda_max - lambda_1) / gap ~ 1e4 ... 1e5
    # (r06 first probe: 400x16x12, kappa_H ~ 3e4 -- 500 iterations only reach a reduction of 7e-8, and beyond ~900 the
    # oracle's own re-association floor is 1e-3: long but not deep.  128x12x10: the lowest modes are x-modes, gap
    # (2 p + 1) (pi / 129)^2, kappa_H ~ 1e3 ... 3e3: a reduction of 1e-12 within 400 ... 800 iterations.  The iterate must
    # be closer to the minimiser than the gap, or the Hessian is indefinite there: eps 1e-6.)
    # p >= 2: the Rayleigh quotient is invariant under X -> X Q, so its Hessian has p (p - 1) / 2 (near-)zero eigenvalues
    # at a minimiser: every solve ends in a boundary / kernel exit once the residual is ~1e-9 ... 1e-10 of |g|.)
    cases = [((128, 12, 10), 1, 1e-3, (200, 1000), 1e-6), ((128, 12, 10), 3, 1e-3, (200, 1000), 1e-4),
             ((128, 12, 10), 8, 1e-3, (200, 1000), 1e-6)]
    if quick:
        cases = [((200, 8, 6), 3, 1e-3, (200, 500), 1e-3)]
    for c_ in cases:
        case(*c_, O, M)


if __name__ == "__main__":
    main()
