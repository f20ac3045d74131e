"""GPU tests of the device constraint preconditioner of the projected STPCG (csrc/kkt.hip; reference
LinearAlgebra/IterativeSolvers.h:83-85,229-253,381-405, cases of tests/IterativeSolvers_unit_test.cpp:316-496) against
dense numpy algebra, including the edges: one constraint, as many constraints as unknowns allow, the 512-constraint
limit, dependent constraint rows, and the projected solve through mi_stpcg against the CPU oracle."""
import numpy as np
import pytest

from conftest import rel_err
from optimization_amd import capi

pytestmark = pytest.mark.gpu


def _kkt_ref(A, Minv, r):
    S = (A * Minv[None, :]) @ A.T
    lam = np.linalg.solve(S, A @ (Minv * r))
    return Minv * (r - A.T @ lam), lam


@pytest.mark.parametrize("n,m", [(1, 1), (7, 1), (64, 64), (1000, 100), (5000, 512), (200_000, 37)])
def test_constraint_preconditioner_vs_dense_algebra(ctx, n, m):
    rng = np.random.default_rng(1000 * m + n % 997)
    A = rng.uniform(-1, 1, (m, n)) + (np.eye(m, n) * 3 if m == n else 0)
    Minv = 1.0 / rng.uniform(1000, 3000, n)
    r = rng.normal(size=n)
    P = ctx.precon_constraint(A, Minv)
    v, lam = P.solve(ctx.upload(r))
    vr, lr = _kkt_ref(A, Minv, r)
    assert rel_err(lam.numpy(), lr) < 1e-9
    assert np.abs(v.numpy() - vr).max() <= 1e-10 * max(np.abs(vr).max(), np.abs(Minv * r).max())
    assert np.abs(A @ v.numpy()).max() <= 1e-9 * np.abs(A).max() * np.abs(Minv * r).max() * n ** .5  # A v = 0
    assert rel_err(P.apply(ctx.upload(r)).numpy(), v.numpy()) == 0.0       # mi_precon_apply = the v of the pair
    w = P.At(lam).numpy()
    assert rel_err(w, A.T @ lam.numpy()) < 1e-13


def test_constraint_preconditioner_rejects_what_it_cannot_factor(ctx):
    rng = np.random.default_rng(3)
    A = rng.normal(size=(3, 50))
    A[2] = 2 * A[0] - A[1]                                   # dependent rows: A M^-1 A' singular
    with pytest.raises(capi.MiError):
        ctx.precon_constraint(A, np.ones(50))
    with pytest.raises(capi.MiError):
        ctx.precon_constraint(rng.normal(size=(513, 600)), np.ones(600))   # m > 512
    with pytest.raises(capi.MiError):
        ctx.precon_constraint(rng.normal(size=(5, 3)), np.ones(3))         # m > n


@pytest.mark.parametrize("use_At", [False, True])
def test_projected_stpcg_through_the_c_abi_vs_oracle(ctx, oracle, use_At):
    """mi_stpcg with the constraint preconditioner, without and with the `At` branch (r -= A' lambda after every
    application, :251,403), against the CPU oracle's STPCG driven with the same dense KKT algebra in numpy."""
    n, m = 600, 40
    rng = np.random.default_rng(11)
    g = rng.uniform(-1, 1, n)
    D = rng.uniform(1000, 3000, n)
    M = rng.uniform(1000, 3000, n)
    A = 1000 * rng.uniform(-1, 1, (m, n))
    Minv = 1.0 / M
    S = (A * Minv[None, :]) @ A.T

    # the oracle restates the unconstrained form; the projected branch is the same loop with v = P(r) and, with At,
    # the residual correction folded into the preconditioner callback's effect on r -- restate it here in numpy
    def stpcg_projected(kappa, theta, max_it):
        s, r = np.zeros(n), g.copy()

        def precond(r):
            lam = np.linalg.solve(S, A @ (Minv * r))
            v = Minv * (r - A.T @ lam)
            if use_At:
                r = r - A.T @ lam
            return v, r
        v, r = precond(r)
        p = -v
        r0 = np.sqrt(r @ v)
        target = r0 * min(kappa, r0 ** theta)
        k = 0
        while k < max_it:
            if np.sqrt(r @ v) <= target:
                break
            Hp = D * p
            alpha = (r @ v) / (p @ Hp)
            s = s + alpha * p
            r = r + alpha * Hp
            rv_old = None
            vn, r = precond(r)
            beta = (r @ vn) / (alpha * (p @ Hp))
            v = vn
            p = -v + beta * p
            k += 1
        return s, k
    so, ko = stpcg_projected(1e-8, .7, 5 * n)
    P = ctx.precon_constraint(A, Minv)
    H = ctx.op_diag(ctx.upload(D))
    r = ctx.stpcg(ctx.upload(g), H, P, Delta=1e300, max_iterations=5 * n, kappa_fgr=1e-8, theta=.7,
                  constraint_At=use_At)
    assert r["iterations"] == ko
    assert rel_err(r["s"].numpy(), so) < 1e-9
    assert np.abs(A @ r["s"].numpy()).max() < 1e-6
