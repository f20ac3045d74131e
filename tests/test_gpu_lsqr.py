"""mi_lsqr through the C ABI (ctypes): known-answer solves, the reference's argument checks, the trivial exit."""
import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from optimization_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def _ops(ctx, A):
    A = sps.csr_matrix(A)
    At = sps.csr_matrix(A.T)
    out = []
    for M in (A, At):
        M.sort_indices()
        out.append(ctx.op_csr(ctx.csr(M.shape[0], M.indptr, M.indices, M.data), 1))
    return out


def test_lsqr_solves_a_nonsymmetric_system(ctx):
    n = 20_000
    A = sps.diags([np.full(n - 1, -1.0), np.full(n, 3.0), np.full(n - 1, 1.5)], [-1, 0, 1], format="csr")
    xs = np.cos(np.arange(n) * 1e-2)
    b = A @ xs
    Aop, Atop = _ops(ctx, A)
    r = ctx.lsqr(Aop, Atop, ctx.upload(b), btol=1e-12, Atol=1e-12, max_iterations=500)
    x = r["x"].numpy()
    assert r["exit_reason"] in (1, 2) and r["iterations"] < 500
    assert np.abs(x - xs).max() < 1e-9
    assert abs(r["xnorm"] - np.linalg.norm(x)) <= 1e-9 * np.linalg.norm(x)      # the QR-based |x| estimate
    assert r["operator_applications"] >= 2 * r["iterations"] + 1


def test_lsqr_trust_region_boundary_and_damping(ctx):
    n = 5_000
    rng = np.random.default_rng(1)
    A = sps.diags([np.full(n - 1, -1.0), np.full(n, 2.5), np.full(n - 1, 0.7)], [-1, 0, 1], format="csr")
    b = rng.normal(size=n)
    Aop, Atop = _ops(ctx, A)
    r = ctx.lsqr(Aop, Atop, ctx.upload(b), Delta=1.0)
    assert r["exit_reason"] == 4 and abs(np.linalg.norm(r["x"].numpy()) - 1.0) < 1e-10    # S4: lands on |x| = Delta
    lam = 0.5
    r = ctx.lsqr(Aop, Atop, ctx.upload(b), lam=lam, btol=1e-13, Atol=1e-13)
    xs = sps.linalg.spsolve((A.T @ A + lam * sps.eye(n)).tocsc(), A.T @ b)              # normal equations of the damped problem
    assert np.abs(r["x"].numpy() - xs).max() < 1e-8


def test_lsqr_trivial_exit_and_argument_checks(ctx):
    from optimization_amd import capi
    n = 1000
    A = sps.eye(n, format="csr") * 2.0
    Aop, Atop = _ops(ctx, A)
    r = ctx.lsqr(Aop, Atop, ctx.upload(np.zeros(n)))
    assert r["exit_reason"] == 5 and r["iterations"] == 0 and not r["x"].numpy().any()   # A'b = 0 (:671-674)
    b = ctx.upload(np.ones(n))
    for bad in (dict(lam=-1.0), dict(btol=-1e-3), dict(Atol=-1.0), dict(Acond_limit=0.0), dict(Delta=0.0)):
        with pytest.raises(capi.MiError) as e:
            ctx.lsqr(Aop, Atop, b, **bad)
        assert e.value.status == 1                                                       # MI_ERR_INVALID_ARGUMENT
    r = ctx.lsqr(Aop, Atop, b, max_iterations=0)
    assert r["iterations"] == 0 and not r["x"].numpy().any()
