#!/usr/bin/env python3
"""Timeline of the LDS-window Hessian kernel from its in-kernel stamps (library built with -DMI_WIN_STAMPS):
per stamp slot, the median / min / max over all waves of (stamp - that wave's kernel-entry stamp), in us of the
100 MHz-calibrated shader clock.  Usage: MI355OPT_LIB=.../libmi355opt_st.so python tools/stamps.py [nx]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from optimization_amd import capi, workloads as wl
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 100
p = 3; n = nx ** 3
rowptr, col, val = wl.laplacian_3d(nx, nx, nx)
Xb, _ = wl.stiefel_bench_iterate(nx, nx, nx, p, eps=1e-3, seed=7)
c = capi.Context(0)
L = c.L
A = c.csr(n, rowptr, col, val)
prob = c.stiefel_rq(A, n, p)
g, H = prob.model(c.upload(Xb))
out = c.vec(n * p)
NW, NS = 256 * 16, 64
buf = c.vec(NW * NS)  # 8-byte slots
ptr = C.c_void_p()
capi.check(L.mi_vec_data(buf.h, C.byref(ptr)))
L.mi_debug_stamp_buffer.argtypes = [C.c_void_p]
us = H.time_fused_apply(g, out, 20)
buf.fill(0.0)
capi.check(L.mi_debug_stamp_buffer(ptr))
us1 = H.time_fused_apply(g, out, 1)   # 3 warm-up calls + 1: the last launch's stamps remain
capi.check(L.mi_debug_stamp_buffer(None))
st = buf.numpy().view(np.uint64).reshape(NW, NS).astype(np.int64)
wg_of = np.arange(NW) // 4          # (4 waves per workgroup of the window kernels)
ran = st[:, 56] > 0
wg_of = wg_of[ran]
st = st[ran]  # waves that ran
t0 = st[:, 0:1]
rel = np.where(st > 0, st - t0, -1)
print(f"kernel {us:.2f} us/launch (stamped build)")
names = {0: "entry", 1: "ring filled + barrier"}
for t in range(6):
    b = 2 + 8 * t
    names.update({b: f"tile {t} start", b + 1: f"tile {t} chunk0 matrix words in", b + 2: f"tile {t} chunk0 consumed",
                  b + 3: f"tile {t} chunk1 matrix words in", b + 4: f"tile {t} chunk1 consumed", b + 5: f"tile {t} epilogue done",
                  b + 6: f"tile {t} chunk staged", b + 7: f"tile {t} barrier passed"})
e0, e1, eb = st[:, 56], st[:, 57], st[:, 58]
print("kernel entry/exit (100 MHz): launch %.2f us; first entry -> last exit %.2f us; wave spans median %.2f max %.2f us; "
      "entry skew median %.2f p90 %.2f max %.2f us; exit spread (last exit - median exit) %.2f us; body->exit median %.2f us"
      % (us, (e1.max() - e0.min()) / 100, np.median(e1 - e0) / 100, (e1 - e0).max() / 100, np.median(e0 - e0.min()) / 100,
         np.percentile(e0 - e0.min(), 90) / 100, (e0 - e0.min()).max() / 100, (e1.max() - np.median(e1)) / 100,
         np.median(e1 - eb) / 100))
# late r06: who is late?  exit time (relative to the first entry) by XCD (workgroup % 8) and by the number of tiles a wave walked
ntile = np.zeros(st.shape[0], dtype=int)
for t in range(6):
    ntile += (st[:, 2 + 8 * t] > 0).astype(int)
ex = (e1 - e0.min()) / 100.0
print("exit by XCD (us after the first entry): " + ", ".join("%d: median %.2f max %.2f" % (x, np.median(ex[wg_of % 8 == x]), ex[wg_of % 8 == x].max()) for x in range(8)))
span = (e1 - e0) / 100.0
print("wave span by XCD (us): " + ", ".join("%d: %.2f" % (x, np.median(span[wg_of % 8 == x])) for x in range(8)))
order = np.argsort(ex)
print("the 16 last waves: workgroups " + " ".join("%d(x%d,%.1f)" % (wg_of[i], wg_of[i] % 8, ex[i]) for i in order[-16:]))
nwg = wg_of.max() + 1
q = np.array([np.median(ex[(wg_of >= a) & (wg_of < a + nwg // 8)]) for a in range(0, nwg - nwg // 8 + 1, nwg // 8)])
print("exit by eighth of the grid (launch order): " + " ".join("%.2f" % v for v in q))
for sl in range(NS):
    v = rel[:, sl][rel[:, sl] >= 0]
    if sl == 0 or v.size == 0 or sl in (NS - 3, NS - 2, 56, 57, 58):
        continue
    print("%-36s n=%4d  median %7d  p10 %7d  p90 %7d  max %7d cycles" % (names.get(sl, f"slot {sl}"), v.size, np.median(v),
          np.percentile(v, 10), np.percentile(v, 90), v.max()))
c.close()
