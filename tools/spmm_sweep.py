#!/usr/bin/env python3
"""Column-major SpMM (LOBPCG panel product) at cfg5 size: time per 24 columns; MI355OPT_SPMM_PK_CHUNK / MI355OPT_NO_PACKED select the variant."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from optimization_amd import capi, workloads as wl
ctx = capi.Context(0)
m = 126 ** 3; nx = 24
rowptr, col, val = wl.laplacian_3d(126, 126, 126)
A = ctx.csr(m, rowptr, col, val)
rng = np.random.default_rng(1)
S = ctx.upload(rng.normal(size=m * nx)); Y = ctx.vec(m * nx)
A.spmm_colmajor(nx, S, Y); ctx.sync()
ref = Y.numpy().copy()
for rep in range(3):
    ctx.timer_start()
    for _ in range(5): A.spmm_colmajor(nx, S, Y)
    ms = ctx.timer_stop()
print(os.environ.get("MI355OPT_SPMM_PK_CHUNK"), os.environ.get("MI355OPT_NO_PACKED"), "us per 24-col product:", round(ms / 5 * 1e3, 1), "checksum", float(np.abs(ref).sum()))
