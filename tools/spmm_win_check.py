#!/usr/bin/env python3
"""Panel SpMM (mi_csr_spmm_colmajor): result and time on the cfg5 Laplacian; with a path argument the result of the
first columns is saved / compared bitwise (A/B of MI355OPT_NO_SPMM_WIN across two processes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from optimization_amd import capi, workloads as wl
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 126
k = int(sys.argv[2]) if len(sys.argv) > 2 else 24
path = sys.argv[3] if len(sys.argv) > 3 else None
n = nx ** 3
rowptr, col, val = wl.laplacian_3d(nx, nx, nx)
rng = np.random.default_rng(5)
X = rng.normal(size=(n, k))
c = capi.Context(0)
A = c.csr(n, rowptr, col, val)
Xd = c.upload(np.asfortranarray(X).ravel(order="F"))
Y = A.spmm_colmajor(k, Xd).numpy().reshape(k, n).T
import scipy.sparse as sps
ref = sps.csr_matrix((val, col, rowptr), shape=(n, n)) @ X
err = float(np.abs(Y - ref).max() / np.abs(ref).max())
c.ktime_enable("csr_spmm", True)
A.spmm_colmajor(k, Xd)
c.ktime_reset()
for _ in range(10):
    A.spmm_colmajor(k, Xd)
cnt, ms = c.ktime_read("csr_spmm")
msg = "nx %d k %d  rel err vs scipy %.2e  %.1f us per product" % (nx, k, err, 1e3 * ms / cnt)
if path:
    if os.path.exists(path):
        msg += "  bitwise equal to saved: %s" % bool(np.array_equal(np.load(path), Y))
    else:
        np.save(path, Y)
print(msg)
c.close()
