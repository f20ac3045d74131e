#!/usr/bin/env python3
"""mi_csr_spmm (row-major n x p field) at cfg2 size: time per product.  MI355OPT_NO_SPMM_STREAM=1 / MI355OPT_NO_PACKED=1
select the older kernel / the plain 12-byte matrix entries."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from optimization_amd import capi, workloads as wl
ctx = capi.Context(0)
n = 100 ** 3
rowptr, col, val = wl.laplacian_3d(100, 100, 100)
A = ctx.csr(n, rowptr, col, val)
for p in (1, 3):
    V = ctx.upload(np.random.default_rng(p).normal(size=n * p))
    W = A.spmm(p, V); ctx.sync()
    chk = float(np.abs(W.numpy()).sum())
    best = 1e9
    for rep in range(3):
        ctx.timer_start()
        for _ in range(20): W = A.spmm(p, V)
        best = min(best, ctx.timer_stop() / 20 * 1e3)
    print("p", p, "us", round(best, 1), "checksum", chk)
