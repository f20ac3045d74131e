#!/usr/bin/env python3
"""Golden BITS of the host Rayleigh-Ritz solve (mi_rayleigh_ritz = LinearAlgebra/DenseSymmetricEigen.h compiled without
FP contraction): SHA-256 of (Theta, C) for pencils built from small integers (so that A and B are exact in fp64 on any
machine and the bits depend on nothing but the solver's own operations).  Written when the triangular solves and the QL
sweep were restructured for speed (r03): the restructured solver was compared bitwise with the previous one on random
pencils before this file was generated, and the file pins the bits from then on.
Usage: python tests/golden/make_rr_bits.py > tests/golden/rr_bits.json"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def pencil(n, seed):
    """A = M'M and B = 8 I + K'K for integer M (n+5 x n, entries in -3..3) and K (3 x n, entries in -1..1): exact."""
    state = [seed * 2654435761 % (1 << 32) or 1]

    def nxt():
        state[0] = (1103515245 * state[0] + 12345) % (1 << 31)
        return state[0]
    M = np.array([[(nxt() >> 8) % 7 - 3 for _ in range(n)] for _ in range(n + 5)], dtype=np.int64)
    K = np.array([[(nxt() >> 8) % 3 - 1 for _ in range(n)] for _ in range(3)], dtype=np.int64)
    A = (M.T @ M).astype(np.float64)
    B = (8 * np.eye(n, dtype=np.int64) + K.T @ K).astype(np.float64)
    return np.asfortranarray(A), np.asfortranarray(B)


def solve_bits(L, n, seed):
    dp = C.POINTER(C.c_double)
    A, B = pencil(n, seed)
    th = np.zeros(n)
    Cm = np.zeros((n, n), order="F")
    rc = L.mi_rayleigh_ritz(n, A.ctypes.data_as(dp), B.ctypes.data_as(dp), th.ctypes.data_as(dp), Cm.ctypes.data_as(dp))
    assert rc == 0
    return hashlib.sha256(th.tobytes() + Cm.tobytes(order="F")).hexdigest(), float(th[0]).hex()


CASES = [(1, 1), (2, 2), (5, 3), (24, 4), (33, 5), (48, 6), (72, 7), (72, 8), (96, 9)]

if __name__ == "__main__":
    from optimization_amd import capi
    L = capi.load()
    out = []
    for n, seed in CASES:
        h, t0 = solve_bits(L, n, seed)
        out.append({"n": n, "seed": seed, "sha256": h, "theta0": t0})
    print(json.dumps(out, indent=1))
