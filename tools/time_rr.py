#!/usr/bin/env python3
"""Host Rayleigh-Ritz (mi_rayleigh_ritz, the header-only solver of DenseSymmetricEigen.h) timed on this host's CPU."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from optimization_amd import capi  # noqa: E402

L = capi.load()
dp = C.POINTER(C.c_double)
rng = np.random.default_rng(0)
for n in (24, 48, 72, 96):
    M = rng.normal(size=(200, n))
    A = np.asfortranarray(M.T @ M)
    B = np.asfortranarray(np.eye(n) + 0.01 * (M[:n].T @ M[:n]))
    th = np.zeros(n)
    Cm = np.zeros((n, n), order="F")
    L.mi_rayleigh_ritz(n, A.ctypes.data_as(dp), B.ctypes.data_as(dp), th.ctypes.data_as(dp), Cm.ctypes.data_as(dp))
    t0 = time.perf_counter()
    for _ in range(10):
        L.mi_rayleigh_ritz(n, A.ctypes.data_as(dp), B.ctypes.data_as(dp), th.ctypes.data_as(dp), Cm.ctypes.data_as(dp))
    full = (time.perf_counter() - t0) / 10 * 1e3
    k = max(1, n // 3)
    L.mi_rayleigh_ritz_lowest.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp]
    tl, Cl = np.zeros(k), np.zeros((n, k), order="F")
    t0 = time.perf_counter()
    for _ in range(10):
        L.mi_rayleigh_ritz_lowest(n, k, A.ctypes.data_as(dp), B.ctypes.data_as(dp), tl.ctypes.data_as(dp), Cl.ctypes.data_as(dp))
    low = (time.perf_counter() - t0) / 10 * 1e3
    print("n=%d  full %.3f ms   lowest %d pairs %.3f ms" % (n, full, k, low), "theta[0] %.17g" % th[0], tl[0] == th[0])
