"""A USER-supplied HIP Hessian-vector product inside the fused STPCG (r04; reference: the callable `H` of
IterativeSolvers.h:166-179 applied at :294, bound from the caller's QuadraticModel at TNT.h:400-426).
examples/stpcg_user_stencil.hip registers a hand-written stencil kernel with mi_op_create_callback_fused -- the kernel
leaves the partial sums of the three curvature inner products itself -- and solves through the drop-in
LinearAlgebra::STPCG template on MI355::DeviceVector; this test runs that client and holds its answer against the CPU
oracle's STPCG on the same operator and right-hand side."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from conftest import ROOT, rel_err

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "examples", "bin", "stpcg_user_stencil")


def _run(n):
    with tempfile.TemporaryDirectory() as tmp:
        dump = os.path.join(tmp, "dump.bin")
        r = subprocess.run([EXE, str(n), dump], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr   # (nonzero also when fused and plain callbacks disagree on the count)
        raw = np.fromfile(dump, dtype=np.float64)
    assert int(raw[0]) == n
    return r.stdout, int(raw[1]), raw[2:2 + n].copy(), raw[2 + n:2 + 2 * n].copy()


def test_user_stencil_operator_fused_into_stpcg_matches_the_oracle(oracle):
    n, sigma = 30_000, 0.05
    out, iters, g, s = _run(n)
    print(out)

    def H(v):
        y = (2.0 + sigma) * v
        y[1:] -= v[:-1]
        y[:-1] -= v[1:]
        return y
    o = oracle.stpcg(g, H, Delta=1e9, max_iterations=200, kappa_fgr=1e-8, theta=1.0)
    assert o["rc"] == 0 and o["exit_reason"] == 0                      # residual exit
    assert iters == o["iterations"] and iters > 20
    assert rel_err(s, o["s"]) <= 1e-10
    assert np.linalg.norm(H(s) + g) <= 1e-7 * np.linalg.norm(g)


def test_user_stencil_operator_at_scale_three_launches_per_iteration():
    """n = 4 M: the fused callback and the plain one take the same number of iterations (the client's exit code), and the
    fused form is the faster one: it saves the separate inner-product pass and its launch."""
    out, iters, _, _ = _run(1 << 22)
    print(out)
    us = {ln.split()[0]: float(ln.split()[-4]) for ln in out.splitlines() if "us per iteration" in ln}
    assert iters > 20 and us["fused"] < us["plain"]


def test_fused_callback_contract_is_enforced(ctx):
    """mi_op_create_callback_fused: a callback that reports a row count outside [1, max_rows] -- or fails -- stops the
    solve with an error instead of letting k_cg_update re-reduce rows nobody wrote; the plain product of the same
    operator (mi_op_apply, TNT's `dm` product) does not touch the partial buffer."""
    from optimization_amd import capi
    n = 4096
    D = ctx.upload(np.full(n, 2.0))
    Hd = ctx.op_diag(D)
    seen = {}

    def plain(i, o):
        Hd.apply(i, o)

    def fused_bad_rows(i, o, a):
        seen["args"] = (a.partial_stride, a.max_rows, a.required_rows, bool(a.stream))
        Hd.apply(i, o)
        return a.max_rows + 1

    def fused_raises(i, o, a):
        raise RuntimeError("user kernel failed to launch")
    g = ctx.upload(np.ones(n))
    for fused in (fused_bad_rows, lambda i, o, a: 0, fused_raises):
        H = ctx.op_callback_fused(n, plain, fused)
        with pytest.raises(capi.MiError):
            ctx.stpcg(g, H, Delta=1e9, max_iterations=5)
        assert np.array_equal(H.apply(g).numpy(), np.full(n, 2.0))          # the plain product still works
    assert seen["args"] == (1024, 1024, 0, True)     # the contract a single-rank context hands out
    ctx.sync()
