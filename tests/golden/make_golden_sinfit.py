"""Golden fixture for the reference's own TNLS problem (tests/TNLS_unit_test.cpp:151-260 of the reference: fit
y = sin(b0 t + b1), m = 100 samples): what the REAL reference (oracle/_ref/libref.so) returns for the three cases of that
file -- root finding on exact data, least squares on noisy data without and with the right preconditioner -- together
with the INPUTS (t, y, the noise: 100 doubles each), so that the device test (tests/test_gpu_templates.py) runs on the
same bytes whatever the host's libm.  Run here (needs /root/reference):  python tests/golden/make_golden_sinfit.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py  # noqa: E402
from optimization_amd import workloads as wl  # noqa: E402

m = 100
t = np.linspace(-np.pi, np.pi, m)
y = np.sin(np.pi / 2 * t + np.pi / 4)
z = .1 * wl.uniform_pm1(20260929, m)      # mt19937_64 (wlgen.c)
R = oracle_py.Reference()
cases = []
for name, yy, kw in (("root", y, dict(root_tolerance=1e-6)),
                     ("least_squares", y + z, dict(with_precon=False, root_tolerance=1e-6, gradient_tolerance=1e-6,
                                                   Delta_tolerance=1e-10)),
                     ("least_squares_preconditioned", y + z, dict(with_precon=True, root_tolerance=1e-6,
                                                                   gradient_tolerance=1e-6, Delta_tolerance=1e-10))):
    r = R.tnls_sinfit(t, yy, [1., 1.], **kw)
    assert r["rc"] == 0
    cases.append(dict(name=name, kw=kw, y=[float(v) for v in yy], beta=[float(v) for v in r["beta"]], f=float(r["f"]),
                      gradfx_norm=float(r["gradfx_norm"]), status=int(r["status"]), outer=int(r["outer"]),
                      inner_total=int(r["inner_total"])))
    print(name, r["status"], r["outer"], r["inner_total"], r["beta"], r["f"])
json.dump(dict(m=m, t=[float(v) for v in t], beta0=[1.0, 1.0], cases=cases),
          open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tnls_sinfit.json"), "w"), indent=1)
