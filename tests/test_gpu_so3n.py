"""GPU parity tests for BASELINE cfg3: chordal rotation averaging on SO(3)^N -- objective, quadratic
model (gradient + 3x3-block sparse Hessian), Rodrigues retraction, 3x3 block-Jacobi preconditioner and
the fused STPCG on top -- against the CPU oracle, at oracle-sized and at the full cfg3 size (N = 5e5)."""
import numpy as np
import pytest

from conftest import rel_err
from optimization_amd import workloads as wl

pytestmark = pytest.mark.gpu


def _problem(ctx, oracle, N, seed, precon=True):
    ei, ej, Rt, w, Rtrue, Rinit = wl.pose_graph(N, seed=seed)
    w = np.linspace(0.5, 1.5, w.size)  # non-uniform weights
    prob = ctx.so3n(N, ei, ej, Rt, w)
    oprob = oracle.so3n(N, ei, ej, Rt, w, precon_kind=1 if precon else 0)
    return prob, oprob, Rinit, Rtrue


@pytest.mark.parametrize("N", [3, 64, 65, 1000])
def test_so3n_pieces_vs_oracle(ctx, oracle, N):
    prob, oprob, Rinit, _ = _problem(ctx, oracle, N, seed=N)
    R = ctx.upload(Rinit)
    x = Rinit.ravel()
    assert abs(prob.objective(R) - oracle.eval_f(oprob, x)) <= 1e-13 * abs(oracle.eval_f(oprob, x))
    g, H, P = prob.model(R)
    go = oracle.eval_grad(oprob, x)
    assert rel_err(g.numpy(), go) < 1e-13
    rng = np.random.default_rng(N)
    for _ in range(3):
        xi = rng.normal(size=3 * N)
        hv = H.apply(ctx.upload(xi)).numpy()
        assert rel_err(hv, oracle.eval_hess(oprob, x, xi)) < 1e-12
    v = rng.normal(size=3 * N)
    assert rel_err(P.apply(ctx.upload(v)).numpy(), oracle.eval_precon(oprob, x, v)) < 1e-12
    xi = 0.3 * rng.normal(size=3 * N)
    xi[:3] = 1e-7  # exercises the small-angle series of the exponential
    Y = prob.retract(R, ctx.upload(xi)).numpy()
    assert rel_err(Y, oracle.eval_retract(oprob, x, xi)) < 1e-13
    Yb = Y.reshape(N, 3, 3)
    assert np.abs(np.einsum("nij,nkj->nik", Yb, Yb) - np.eye(3)).max() < 1e-13
    oracle.free(oprob)


@pytest.mark.parametrize("precon", [False, True])
def test_so3n_fused_stpcg_vs_oracle(ctx, oracle, precon):
    N = 2000
    prob, oprob, Rinit, _ = _problem(ctx, oracle, N, seed=5, precon=precon)
    R = ctx.upload(Rinit)
    g, H, P = prob.model(R)
    go = oracle.eval_grad(oprob, Rinit.ravel())
    for Delta, kappa in ((1e3, 1e-8), (0.5, .1)):
        r = ctx.stpcg(g, H, P if precon else None, Delta=Delta, max_iterations=60, kappa_fgr=kappa, theta=.5,
                      trace_cap=64)
        o = oracle.stpcg_problem(oprob, Rinit.ravel(), go, Delta, max_iterations=60, kappa_fgr=kappa, theta=.5,
                                 trace_cap=64)
        assert r["iterations"] == o["iterations"] and r["exit_reason"] == o["exit_reason"]
        assert np.allclose(r["trace"]["alpha"], o["trace"]["alpha"], rtol=1e-9)
        assert rel_err(r["s"].numpy(), o["s"]) < 1e-9
        assert abs(r["M_norm"] - o["M_norm"]) <= 1e-9 * o["M_norm"]
    oracle.free(oprob)


def test_so3n_full_size_cfg3(ctx, oracle):
    """N = 5e5 poses, ring + 2 chords per node (1.5e6 edges), block-Jacobi STPCG: parity with the oracle
    on the same arrays, plus size-independent properties."""
    N = 500_000
    ei, ej, Rt, w, Rtrue, Rinit = wl.pose_graph(N, seed=7)
    prob = ctx.so3n(N, ei, ej, Rt, w)
    oprob = oracle.so3n(N, ei, ej, Rt, w, precon_kind=1)
    R = ctx.upload(Rinit)
    x = Rinit.ravel()
    fo = oracle.eval_f(oprob, x)
    assert abs(prob.objective(R) - fo) <= 1e-12 * fo
    # noise-free ground truth would give f = 0; the noisy truth gives f ~ sigma^2 * E
    ftrue = prob.objective(ctx.upload(Rtrue))
    assert abs(ftrue - 0.5 * ei.size * 2 * 3 * 0.05 ** 2) < 0.05 * ftrue  # = 1/2 E * 2 * E|noise|^2
    assert ftrue < 0.05 * fo
    g, H, P = prob.model(R)
    go = oracle.eval_grad(oprob, x)
    assert rel_err(g.numpy(), go) < 1e-12
    rng = np.random.default_rng(0)
    u, v = ctx.upload(rng.normal(size=3 * N)), ctx.upload(rng.normal(size=3 * N))
    Hu, Hv = H.apply(u), H.apply(v)
    a, b = u.dot(Hv), v.dot(Hu)
    assert abs(a - b) <= 1e-11 * max(abs(a), abs(b))  # self-adjoint
    Pu, Pv = P.apply(u), P.apply(v)
    assert abs(u.dot(Pv) - v.dot(Pu)) <= 1e-11 * abs(u.dot(Pv))
    assert v.dot(Pv) > 0 and v.dot(Hv) > 0            # SPD near the solution
    r = ctx.stpcg(g, H, P, Delta=1e4, max_iterations=15, kappa_fgr=1e-10, theta=1.0, trace_cap=16)
    o = oracle.stpcg_problem(oprob, x, go, 1e4, max_iterations=15, kappa_fgr=1e-10, theta=1.0, trace_cap=16)
    assert r["iterations"] == o["iterations"] == 15
    assert np.allclose(r["trace"]["alpha"], o["trace"]["alpha"], rtol=1e-9)
    assert rel_err(r["s"].numpy(), o["s"]) < 1e-10
    # one Newton-CG step from the noisy initial guess must reduce the objective substantially
    Y = prob.retract(R, r["s"])
    assert prob.objective(Y) < 0.2 * fo
    oracle.free(oprob)


@pytest.mark.parametrize("with_precon", [False, True])
def test_so3n_fused_trial_step_has_the_bits_of_the_separate_calls(ctx, with_precon):
    """mi_so3n_trial (reference Riemannian/TNT.h:493-512,573-585 as one launch chain, one read-back) against the
    separate calls it replaces -- Hessian product + dot products, retraction, objective, model and both gradient
    norms at the trial point: bit for bit.  The speculative model is swapped in by the next model() call for the
    SAME vector contents only (handle + serial + generation stamp)."""
    N = 3000
    ei, ej, Rt, w, _, Rinit = wl.pose_graph(N, seed=21)
    prob = ctx.so3n(N, ei, ej, Rt, w)
    R = ctx.upload(Rinit)
    g, H, P = prob.model(R)
    h = ctx.upload(np.random.default_rng(4).normal(size=3 * N) * 1e-2)
    # separate calls
    Hh = H.apply(h)
    hh, gh, hHh = ctx.dot_batch([h, g, h], [h, h, Hh])
    Rt_ref = prob.retract(R, h)
    f_ref = prob.objective(Rt_ref)
    # fused
    Rtr, t = prob.trial(R, h, g, with_precon=with_precon)
    assert np.array_equal(Rtr.numpy(), Rt_ref.numpy())
    assert (t["f"], t["hh"], t["gh"], t["hHh"]) == (f_ref, hh, gh, hHh)
    g2, H2, P2 = prob.model(Rtr)      # swaps the speculative model in
    prob2 = ctx.so3n(N, ei, ej, Rt, w)
    g2_ref, H2_ref, P2_ref = prob2.model(Rt_ref)
    assert np.array_equal(g2.numpy(), g2_ref.numpy())
    assert t["grad_sqnorm"] == g2_ref.dot(g2_ref)
    if with_precon:
        Pg = P2_ref.apply(g2_ref)
        assert t["precon_grad_sqnorm"] == Pg.dot(Pg)
    else:
        assert t["precon_grad_sqnorm"] == -1.0
    v = ctx.upload(np.random.default_rng(5).normal(size=3 * N))
    assert np.array_equal(H2.apply(v).numpy(), H2_ref.apply(v).numpy())
    assert np.array_equal(P2.apply(v).numpy(), P2_ref.apply(v).numpy())
    r1 = ctx.stpcg(g2, H2, P2, Delta=10.0, max_iterations=8, kappa_fgr=1e-10, theta=1.0)
    r2 = ctx.stpcg(g2_ref, H2_ref, P2_ref, Delta=10.0, max_iterations=8, kappa_fgr=1e-10, theta=1.0)
    assert np.array_equal(r1["s"].numpy(), r2["s"].numpy())


def test_so3n_speculative_model_is_dropped_when_the_trial_vector_changes(ctx):
    """the speculation must not survive an in-place write to the trial vector, nor a recycled handle."""
    N = 500
    ei, ej, Rt, w, _, Rinit = wl.pose_graph(N, seed=2)
    prob = ctx.so3n(N, ei, ej, Rt, w)
    R = ctx.upload(Rinit)
    g, H, P = prob.model(R)
    h = ctx.upload(np.random.default_rng(1).normal(size=3 * N) * 1e-2)
    Rtr, _ = prob.trial(R, h, g)
    other = wl.pose_graph(N, seed=3)[5]
    Rtr.set(other)                     # same handle, same device pointer, other contents
    g2, _, _ = prob.model(Rtr)
    prob2 = ctx.so3n(N, ei, ej, Rt, w)
    g2_ref, _, _ = prob2.model(ctx.upload(other))
    assert np.array_equal(g2.numpy(), g2_ref.numpy())
