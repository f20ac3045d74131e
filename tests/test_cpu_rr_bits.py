"""The host Rayleigh-Ritz solve keeps its BITS (reference LinearAlgebra/LOBPCG.h:53-62; ours
LinearAlgebra/DenseSymmetricEigen.h through mi_rayleigh_ritz): LOBPCG's Ritz values are compared bit for bit between
rounds and between the host and device runs of the template, so a faster solver must be the same arithmetic.  The
fixture (tests/golden/rr_bits.json, tests/golden/make_rr_bits.py) holds SHA-256 of (Theta, C) for pencils made of small
integers -- exact inputs on any machine -- and was reproduced by the solver as it stood before the r03 restructuring
(pivot-outer triangular solves, QL rotations applied one step late)."""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_rr_bits  # noqa: E402


def test_rayleigh_ritz_bits_are_pinned():
    from optimization_amd import capi
    L = capi.load()
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rr_bits.json")) as f:
        golden = json.load(f)
    assert [(g["n"], g["seed"]) for g in golden] == make_rr_bits.CASES
    for g in golden:
        sha, t0 = make_rr_bits.solve_bits(L, g["n"], g["seed"])
        assert (sha, t0) == (g["sha256"], g["theta0"]), g["n"]


@pytest.mark.parametrize("n", [5, 72])
def test_rayleigh_ritz_identities_on_the_integer_pencils(n):
    """C'BC = I and C'AC = diag(Theta), ascending (the identities of the reference's tests/LOBPCG_unit_test.cpp)"""
    import ctypes as C
    from optimization_amd import capi
    L = capi.load()
    dp = C.POINTER(C.c_double)
    A, B = make_rr_bits.pencil(n, 11)
    th = np.zeros(n)
    Cm = np.zeros((n, n), order="F")
    assert L.mi_rayleigh_ritz(n, A.ctypes.data_as(dp), B.ctypes.data_as(dp), th.ctypes.data_as(dp), Cm.ctypes.data_as(dp)) == 0
    assert np.all(np.diff(th) >= 0)
    assert np.abs(Cm.T @ B @ Cm - np.eye(n)).max() < 1e-12
    assert np.abs(Cm.T @ A @ Cm - np.diag(th)).max() < 1e-10 * np.abs(th).max()


@pytest.mark.parametrize("n,k", [(1, 1), (2, 1), (5, 2), (24, 24), (48, 16), (72, 24), (96, 32), (72, 72), (60, 20)])
def test_lowest_pairs_solver_against_the_full_one(n, k):
    """mi_rayleigh_ritz_lowest (r04: the nx lowest Ritz pairs of the ns <= 3 nx an LOBPCG iteration reads; same
    reduction, same QL recurrence, the recorded plane rotations applied in reverse to k unit columns instead of forward
    to n rows): Ritz values BIT-identical to the full solver's first k, vectors equal to rounding -- also inside
    clusters (a triple and a sextuple that differs in the 12th digit: the converged Ritz values of cfg5 are such
    clusters), where an inverse-iteration scheme would need re-orthogonalisation -- and the reference's identities."""
    import ctypes as C
    from optimization_amd import capi
    L = capi.load()
    dp = C.POINTER(C.c_double)
    rng = np.random.default_rng(100 + n)
    G = rng.normal(size=(n + 5, n))
    B = G.T @ G + 0.1 * np.eye(n)
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    lam = np.sort(rng.uniform(0.1, 3.0, size=n))
    if n >= 24:
        lam[3:6] = lam[3]
        lam[10:16] = lam[10] * (1 + 1e-12 * np.arange(6))
    Lc = np.linalg.cholesky(B)
    A = Lc @ (Q * lam) @ Q.T @ Lc.T
    A = np.asfortranarray(.5 * (A + A.T))
    B = np.asfortranarray(B)
    th, Cm = np.zeros(n), np.zeros((n, n), order="F")
    assert L.mi_rayleigh_ritz(n, A.ctypes.data_as(dp), B.ctypes.data_as(dp), th.ctypes.data_as(dp), Cm.ctypes.data_as(dp)) == 0
    L.mi_rayleigh_ritz_lowest.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp]
    tl, Cl = np.zeros(k), np.zeros((n, k), order="F")
    assert L.mi_rayleigh_ritz_lowest(n, k, A.ctypes.data_as(dp), B.ctypes.data_as(dp), tl.ctypes.data_as(dp),
                                     Cl.ctypes.data_as(dp)) == 0
    assert np.array_equal(tl, th[:k])
    assert np.abs(Cl - Cm[:, :k]).max() <= 1e-13 * np.abs(Cm).max()
    assert np.abs(Cl.T @ B @ Cl - np.eye(k)).max() < 1e-12
    assert np.abs(Cl.T @ A @ Cl - np.diag(tl)).max() < 1e-11 * np.abs(th).max()
    # the reference's argument behaviour: a B that is not positive definite is an error, not a wrong answer
    Bbad = B.copy()
    Bbad[0, 0] = -1.0
    assert L.mi_rayleigh_ritz_lowest(n, k, A.ctypes.data_as(dp), Bbad.ctypes.data_as(dp), tl.ctypes.data_as(dp),
                                     Cl.ctypes.data_as(dp)) != 0
