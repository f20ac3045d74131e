"""GPU tests of the multi-GPU code path on ONE GPU: a size-1 RCCL communicator goes through exactly the
calls the 8-GPU node makes (ncclGetUniqueId / ncclCommInitRank / in-stream ncclAllReduce of the scalar
slots / halo planning), and the 'slot' variants of every consumer kernel (FROM_SLOTS) must reproduce the
single-GPU results bit for bit (the one-workgroup reduce uses the same fixed-order summation)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from optimization_amd import workloads as wl

pytestmark = pytest.mark.gpu


def _solve(c, nx, ny, nz, p, sharded_matrix):
    n = nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    if sharded_matrix:
        rp, colg, vl = wl.laplacian_3d(nx, ny, nz, z_range=(0, nz))
        A = c.csr_sharded(n, 0, n, rp, colg, vl, [0, n])
    else:
        A = c.csr(n, rowptr, col, val)
    prob = c.stiefel_rq(A, n, p)
    Xb, _ = wl.stiefel_bench_iterate(nx, ny, nz, p, eps=1e-2, seed=5)
    X = c.upload(Xb)
    f = prob.objective(X)
    g, H = prob.model(X)
    r = c.stpcg(g, H, Delta=1e3, max_iterations=40, kappa_fgr=1e-8, theta=.5, trace_cap=64)
    r2 = c.stpcg(g, H, Delta=0.01, max_iterations=40)
    Y = c.stiefel_retract(n, p, X, r2["s"]).numpy()
    return dict(f=f, g=g.numpy(), s=r["s"].numpy(), it=r["iterations"], M=r["M_norm"], alpha=r["trace"]["alpha"],
                s2=r2["s"].numpy(), it2=r2["iterations"], exit2=r2["exit_reason"], Y=Y, dot=g.dot(g))


def test_size1_communicator_matches_plain_context_bitwise():
    from optimization_amd import capi
    plain = capi.Context(0)
    a = _solve(plain, 24, 20, 16, 3, sharded_matrix=False)
    plain.close()
    comm = capi.Context(0)
    comm.comm_init(1, 0, comm.comm_unique_id())
    b = _solve(comm, 24, 20, 16, 3, sharded_matrix=True)
    comm.comm_finalize()
    comm.close()
    assert a["it"] == b["it"] and a["it2"] == b["it2"] and a["exit2"] == b["exit2"]
    for k in ("f", "M", "dot"):
        assert a[k] == b[k], k
    for k in ("g", "s", "alpha", "s2", "Y"):
        assert np.array_equal(a[k], b[k]), k


def test_forced_slot_path_env_runs_whole_suite_subset():
    """MI355OPT_FORCE_SLOT_PATH=1 switches every consumer to its FROM_SLOTS variant without RCCL."""
    env = dict(os.environ, MI355OPT_FORCE_SLOT_PATH="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu",
                        os.path.join(ROOT, "tests", "test_gpu_stiefel.py"),
                        os.path.join(ROOT, "tests", "test_gpu_blas1_stpcg.py"), "-k",
                        "not axpby and not full_size"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_bench_dress_rehearsal_of_the_multi_gpu_launch():
    """bench.py launched the way the driver launches N > 1 (torch.distributed.run, one rank per GPU) with
    one rank: gloo rendezvous, ncclUniqueId broadcast, communicator, sharded matrix, slot path."""
    env = dict(os.environ, MI355OPT_BENCH_FORCE_COMM="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "100",
           "--warmup", "10", "--no-cpu-baseline", "--no-roofline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 100 and d["value"] > 500
    assert "comm" in d["config"]["parallelism"]
