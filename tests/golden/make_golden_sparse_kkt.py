#!/usr/bin/env python3
"""tests/golden/stpcg_projected_sparse.json: what the REAL reference (oracle/_ref/libref.so = the reference's
IterativeSolvers.h compiled from /root/reference, projected branch :229-253,381-405, KKT solves by the dense
factorisation of oracle/kkt_dense.h) returns on the sparse-constraint case of oracle_py.projected_stpcg_sparse_problem.
Data only (inputs are regenerated from their seed).  Run in the build container: python tests/golden/make_golden_sparse_kkt.py"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import oracle_py  # noqa: E402

ref = oracle_py.Reference()
pr = oracle_py.projected_stpcg_sparse_problem()
r = oracle_py.stpcg_projected(ref.lib, "ref", pr)
assert r["rc"] == 0
out = {"n": pr["n"], "m": pr["m"], "nnz": int(np.count_nonzero(pr["A"])), "iterations": r["iterations"],
       "M_norm": r["M_norm"], "s": [float(v) for v in r["s"]],
       "inputs_sha256": hashlib.sha256(pr["A"].tobytes() + pr["g"].tobytes() + pr["P"].tobytes() + pr["M"].tobytes()).hexdigest(),
       "As_norm": float(np.linalg.norm(pr["A"] @ r["s"])),
       "made_by": "oracle/_ref/libref.so (reference IterativeSolvers.h, projected STPCG) via make_golden_sparse_kkt.py"}
json.dump(out, open(os.path.join(HERE, "stpcg_projected_sparse.json"), "w"), indent=1)
print(out["iterations"], out["M_norm"], out["As_norm"], out["nnz"])
