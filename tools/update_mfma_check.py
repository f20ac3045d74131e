import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from optimization_amd import capi
c = capi.Context(0)
rng = np.random.default_rng(3)
for m, ks in ((2000376, 72), (100008, 48), (4096, 72), (1000, 72)):
    S = rng.normal(size=(m, ks)); Cm = rng.normal(size=(ks, 48))
    Sd = c.upload(np.asfortranarray(S).ravel(order="F"))
    os.environ.pop("MI355OPT_NO_UPDATE_MFMA", None)
    Y1 = c.lobpcg_update(m, Sd, ks, Cm).numpy().reshape(48, m).T
    os.environ["MI355OPT_NO_UPDATE_MFMA"] = "1"
    Y0 = c.lobpcg_update(m, Sd, ks, Cm).numpy().reshape(48, m).T
    ref = S @ Cm
    print(m, ks, "bitwise equal:", bool(np.array_equal(Y0, Y1)), "max rel diff %.2e" % (np.abs(Y0 - Y1).max() / np.abs(ref).max()),
          "err vs numpy %.2e / %.2e" % (np.abs(Y1 - ref).max() / np.abs(ref).max(), np.abs(Y0 - ref).max() / np.abs(ref).max()))
c.close()
