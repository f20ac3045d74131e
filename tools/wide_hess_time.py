import sys, json, os
sys.path.insert(0,'/root/repo')
import numpy as np
from optimization_amd import capi, workloads as wl
nx=100; n=nx**3
c=capi.Context(0)
A=c.csr(n,*wl.laplacian_3d(nx,nx,nx))
for p in (5,8):
    prob=c.stiefel_rq(A,n,p)
    X=c.upload(wl.stiefel_bench_iterate(nx,nx,nx,p,eps=1e-3,seed=7)[0])
    g,H=prob.model(X); s=c.vec(n*p)
    c.ktime_enable("stiefel_hess_fused",True)
    for _ in range(3): c.stpcg(g,H,Delta=1e3,max_iterations=50,kappa_fgr=1e-12,theta=1.0,s_out=s)
    c.ktime_reset()
    for _ in range(4): c.stpcg(g,H,Delta=1e3,max_iterations=50,kappa_fgr=1e-12,theta=1.0,s_out=s)
    k,ms=c.ktime_read("stiefel_hess_fused"); print(os.environ.get("MI355OPT_LIB","base")[-12:], p, round(1e3*ms/k,1))
