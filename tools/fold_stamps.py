#!/usr/bin/env python3
"""Where a folded exchange spends its time (library built with -DMI_FOLD_STAMPS; comm_ipc.h FOLD_STAMP): one rank with
a communicator of size 1 through the peer-memory layer, the bench's cfg2 solve; the stamps of the LAST k_cg_update
launch, per workgroup, relative to the earliest kernel entry, in us of the 100 MHz clock.
Usage (GPU box): MI355OPT_LIB=$PWD/optimization_amd/libmi355opt_fstamp.so python tools/fold_stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29633")
import numpy as np
from optimization_amd import capi, workloads as wl
import torch.distributed as dist
single = len(sys.argv) > 1 and sys.argv[1] == "single"   # no communicator: entry / exit stamps only
nx, p = 100, 3
n = nx ** 3
c = capi.Context(0)
rowptr, col, val = wl.laplacian_3d(nx, nx, nx)
if single:
    if os.environ.get("FS_INIT_PG"):
        dist.init_process_group("gloo", rank=0, world_size=1)
    if os.environ.get("FS_PEER_OTHER_CTX"):   # the layer is brought up on ANOTHER context of this process
        c2 = capi.Context(0)
        assert c2.enable_peer_memory(1, 0, dist, force=True)
    A = c.csr(n, rowptr, col, val)
else:
    dist.init_process_group("gloo", rank=0, world_size=1)
    assert c.enable_peer_memory(1, 0, dist, force=True)
    A = c.csr(n, rowptr, col, val) if os.environ.get("FOLD_STAMPS_PLAIN_MATRIX") else c.csr_sharded(n, 0, n, rowptr, col, val, [0, n])
prob = c.stiefel_rq(A, n, p)
X = c.upload(wl.stiefel_bench_iterate(nx, nx, nx, p, eps=1e-3, seed=7)[0])
g, H = prob.model(X)
NWG = 512
buf = c.vec(NWG * 8)
ptr = C.c_void_p()
capi.check(c.L.mi_vec_data(buf.h, C.byref(ptr)))
c.L.mi_debug_fold_stamp_buffer.argtypes = [C.c_void_p]
c.stpcg(g, H, Delta=1e3, max_iterations=50, kappa_fgr=1e-12, theta=1.0)
buf.fill(0.0)
assert c.L.mi_debug_fold_stamp_buffer(ptr) == 0
c.stpcg(g, H, Delta=1e3, max_iterations=30, kappa_fgr=1e-12, theta=1.0)
assert c.L.mi_debug_fold_stamp_buffer(None) == 0
st = buf.numpy().view(np.uint64).reshape(NWG, 8).astype(np.int64)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
names = ["kernel entry", "local sums reduced (fold begins)", "push issued (workgroup 0 stores)", "flag seen",
         "barrier behind the wait", "sums read + broadcast (fold ends)", "kernel exit"]
print(f"{st.shape[0]} workgroups; us after the first workgroup's entry: median / min / max")
for i, nm in enumerate(names):
    v = (st[:, i] - t0) / 100.0
    print(f"  {i} {nm:40s} {np.median(v):7.2f} {v.min():7.2f} {v.max():7.2f}")
print("workgroup 0:", [(int(x) - int(t0)) / 100.0 for x in st[0, :7]])
for i in (0, 6):
    v = np.sort((st[:, i] - t0) / 100.0)
    print(names[i], "percentiles 0/10/25/50/75/90/100:", [round(float(v[int(q * (len(v) - 1))]), 2) for q in (0, .1, .25, .5, .75, .9, 1)])
print("workgroup duration (exit - entry): median %.2f min %.2f max %.2f" % tuple(f((st[:, 6] - st[:, 0]) / 100.0) for f in (np.median, np.min, np.max)))
print("entry by blockIdx (every 32nd):", [round(float((st[b, 0] - t0) / 100.0), 1) for b in range(0, st.shape[0], 32)])
if not single:
    c.comm_finalize()
c.close()
