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

from conftest import ROOT, floor_or, rel_err
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


def test_lockstep_enqueue_count_is_a_function_of_the_device_state_only():
    """With several ranks every speculatively enqueued STPCG iteration carries collectives, so all ranks must
    enqueue the same number: (B-step launches at the exit) + run_ahead, whatever the host timing.  The rule is
    forced on one GPU (MI355OPT_FORCE_LOCKSTEP=1); hvp_calls counts the enqueued iterations."""
    code = r'''
import json, sys, time
import numpy as np
sys.path.insert(0, %r)
from optimization_amd import capi
c = capi.Context(0)
rng = np.random.default_rng(3)
n = 200_000
D = rng.uniform(1.0, 50.0, n); g = rng.normal(size=n)
G, H = c.upload(g), c.op_diag(c.upload(D))
out = []
for ra in (1, 3, 7):
    for rep in range(3):
        if rep == 2:
            time.sleep(0.05)           # perturb host timing
        for kw in (dict(Delta=1e9, max_iterations=200, kappa_fgr=1e-6, theta=1.0),     # residual exit
                   dict(Delta=0.05, max_iterations=200, kappa_fgr=1e-12, theta=1.0),   # boundary exit
                   dict(Delta=1e9, max_iterations=5, kappa_fgr=1e-12, theta=1.0)):     # iteration limit
            r = c.stpcg(G, H, run_ahead=ra, **kw)
            out.append((ra, r["iterations"], r["exit_reason"], r["hvp_calls"], float(r["M_norm"])))
print(json.dumps(out))
''' % ROOT
    res = {}
    for force in ("0", "1"):
        env = dict(os.environ, MI355OPT_FORCE_LOCKSTEP=force)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[force] = json.loads(r.stdout.strip().splitlines()[-1])
    # identical answers with and without the rule
    assert [(a, b, c_, e) for a, b, c_, _, e in res["0"]] == [(a, b, c_, e) for a, b, c_, _, e in res["1"]]
    # lockstep: the enqueue count depends only on (run_ahead, solve), never on the repetition / timing
    seen = {}
    for i, (ra, it, ex, hvp, _) in enumerate(res["1"]):
        key = (ra, i % 3)
        seen.setdefault(key, set()).add(hvp)
        launches = it + (1 if ex == 3 else 0)          # a boundary exit spends one more B-step launch... at most
        assert hvp <= min({0: 200, 1: 200, 2: 5}[i % 3], launches + 1 + ra), (ra, it, ex, hvp)
    assert all(len(v) == 1 for v in seen.values()), seen


@pytest.mark.parametrize("world", [2, 3])
def test_halo_addressing_of_the_sharded_spmm_against_the_global_product(world):
    """Each rank's slab of a row-sharded matrix, with its halo rows filled by hand (what the in-stream
    ncclSend/ncclRecv exchange delivers), must reproduce the corresponding rows of the global A V bit for bit:
    checks the shard plan, the local column remap and the kernels' halo addressing on one GPU."""
    from optimization_amd import capi
    nx, ny, nz, p = 12, 10, 9, 3
    n = nx * ny * nz
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    rng = np.random.default_rng(world)
    V = rng.normal(size=(n, p))
    c0 = capi.Context(0)
    ref = c0.csr(n, rowptr, col, val).spmm(p, c0.upload(V)).numpy().reshape(n, p)
    c0.close()
    slabs = wl.shard_rows(nz, world)
    starts = [nx * ny * a for a, _ in slabs] + [n]
    for rank, (z0, z1) in enumerate(slabs):
        c = capi.Context(0)
        c.debug_set_rank(world, rank)
        rp, colg, vl = wl.laplacian_3d(nx, ny, nz, z_range=(z0, z1))
        r0, r1 = starts[rank], starts[rank + 1]
        A = c.csr_sharded(n, r0, r1, rp, colg, vl, starts)
        _, need_lo, need_hi = capi.csr_shard_plan(n, world, rank, starts, colg)
        assert (need_lo > 0) == (rank > 0) and (need_hi > 0) == (rank + 1 < world)
        halo = np.concatenate([V[r0 - need_lo:r0], V[r1:r1 + need_hi]])      # rows the neighbours would send
        A.debug_set_halo(p, halo)
        Y = A.spmm(p, c.upload(V[r0:r1])).numpy().reshape(r1 - r0, p)
        assert np.array_equal(Y, ref[r0:r1]), rank
        # the fused Stiefel operator kernels read through the same view: objective/gradient pass on the slab
        c.close()


def test_uniform_grid_rows_mode_runs_the_suite_subset():
    """Several ranks all-reduce the partial ROWS (no one-workgroup reduce kernel); that needs every rank to
    leave the same number of rows, so all reduction-producing kernels run kMaxGrid workgroups.  Forced here on
    one GPU with a size-1 communicator: small problems then run mostly idle workgroups (zero partials) and must
    still agree with the CPU oracle."""
    code = r'''
import sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r + "/oracle")
from optimization_amd import capi, workloads as wl
import oracle_py
c = capi.Context(0)
c.comm_init(1, 0, c.comm_unique_id())
O = oracle_py.Oracle()
nx, ny, nz, p = 9, 8, 7, 3
n = nx * ny * nz
rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
A = c.csr_sharded(n, 0, n, *wl.laplacian_3d(nx, ny, nz, z_range=(0, nz)), [0, n])
prob = c.stiefel_rq(A, n, p)
Xb, _ = wl.stiefel_bench_iterate(nx, ny, nz, p, eps=1e-2, seed=5)
X = c.upload(Xb)
g, H = prob.model(X)
r = c.stpcg(g, H, Delta=1e3, max_iterations=30, kappa_fgr=1e-8, theta=.5)
op = O.stiefel_rq(n, p, rowptr, col, val)
go = O.eval_grad(op, Xb.ravel())
ro = O.stpcg_problem(op, Xb.ravel(), go, 1e3, max_iterations=30, kappa_fgr=1e-8, theta=.5)
assert np.abs(g.numpy() - go).max() <= 1e-12 * np.abs(go).max()
assert r["iterations"] == ro["iterations"], (r["iterations"], ro["iterations"])
assert np.abs(r["s"].numpy() - ro["s"]).max() <= 1e-9 * np.abs(ro["s"]).max()
# diagonal operator + block-free preconditioned solve through the generic kernels
D = np.linspace(1.0, 30.0, 5000); gg = np.cos(np.arange(5000.0))
r2 = c.stpcg(c.upload(gg), c.op_diag(c.upload(D)), c.precon_diag(c.upload(1.0 / D)), Delta=1e9, max_iterations=50,
             kappa_fgr=1e-10, theta=1.0)
assert np.abs(r2["s"].numpy() + gg / D).max() < 1e-9
# the fused LSQR through the same RCCL rows path (all-reduce of the partial rows in front of every consumer):
# the sharded symmetric operator is its own transpose; against the plain-context solve.  20 passes: the two SpMV
# variants (with / without a halo view) round differently in the last bit, and past convergence (~25 passes here) a
# Lanczos process amplifies that to 1e-7 by pass 60; with callback operators the two contexts agree bit for bit
opA = c.op_csr(A, 1)
bb = np.sin(np.arange(n) * 0.37)
l1 = c.lsqr(opA, opA, c.upload(bb), btol=1e-10, Atol=1e-10, max_iterations=20)
c2 = capi.Context(0)
A2 = c2.csr(n, rowptr, col, val)
opA2 = c2.op_csr(A2, 1)
l2 = c2.lsqr(opA2, opA2, c2.upload(bb), btol=1e-10, Atol=1e-10, max_iterations=20)
assert (l1["iterations"], l1["exit_reason"]) == (l2["iterations"], l2["exit_reason"]), (l1["iterations"], l2["iterations"])
assert np.abs(l1["x"].numpy() - l2["x"].numpy()).max() <= 1e-13 * np.abs(l2["x"].numpy()).max()
o1, o2 = opA, opA2
cb1 = c.op_callback(n, lambda i, o: o1.apply(i, o))
cb2 = c2.op_callback(n, lambda i, o: o2.apply(i, o))
l1 = c.lsqr(cb1, cb1, c.upload(bb), btol=1e-10, Atol=1e-10, max_iterations=60)
l2 = c2.lsqr(cb2, cb2, c2.upload(bb), btol=1e-10, Atol=1e-10, max_iterations=60)
assert l1["iterations"] == l2["iterations"] and np.array_equal(l1["x"].numpy(), l2["x"].numpy())
c2.close()
c.comm_finalize(); c.close()
print("ok")
''' % (ROOT, ROOT)
    env = dict(os.environ, MI355OPT_FORCE_UNIFORM_GRID="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("world", [2, 3])
def test_peer_memory_layer_with_real_peers_on_one_gpu(world):
    """W processes on GPU 0 talk through the peer-memory (hipIpc) layer: scalar all-reduces, halo exchange,
    sharded Stiefel operator, fused STPCG with the lockstep enqueue rule -- against a single-process solve."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29560 + world), os.path.join(ROOT, "tests", "ipc_worker.py")]
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run(cmd, env=dict(os.environ, IPC_WORKER_OUT=tmp), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        outs = [json.load(open(os.path.join(tmp, f))) for f in sorted(os.listdir(tmp))]
    assert len(outs) == world and all(o["enabled"] for o in outs), outs
    for o in outs:
        assert o["ipc_error"] == 0
        assert o["dot_err"] < 1e-13 and o["spmm_err"] == 0.0 and o["spmm2_err"] == 0.0, o
        assert abs(o["f"] - o["f_ref"]) <= 1e-12 * abs(o["f_ref"])
        assert o["g_err"] < 1e-12 and o["s_err"] < 1e-10 and o["retract_err"] < 1e-12, o
        assert (o["iters"], o["exit"]) == (o["iters_ref"], o["exit_ref"])
        assert (o["b_iters"], o["b_exit"]) == (o["b_iters_ref"], o["b_exit_ref"])
        assert abs(o["M"] - o["M_ref"]) <= 1e-10 * abs(o["M_ref"]) and o["same_s"]
        # row-sharded LOBPCG pieces and the whole loop
        assert o["gram_err"] < 1e-13 and o["resid_err"] < 1e-13 and o["xnorm_err"] < 1e-12, o
        assert o["spmm_colmajor_err"] < 1e-13, o
        assert o["lobpcg_rc"] == 0, o["lobpcg_err"]
        assert o["lobpcg_nconv"] >= 4 and o["lobpcg_nconv_ref"] >= 4
        assert abs(o["lobpcg_iters"] - o["lobpcg_iters_ref"]) <= 2, o
        assert np.allclose(o["lobpcg_theta"], o["lobpcg_theta_ref"], rtol=0, atol=1e-9), o
        assert np.allclose(o["lobpcg_theta"], o["lobpcg_exact"], rtol=0, atol=1e-6), o
        assert o["lobpcg_x_err"] < 1e-4, o
    # replicated scalars are bit-identical on all ranks, and so is the number of enqueued iterations
    for k in ("dot", "f", "M", "b_M", "iters", "hvp1", "hvp5", "gram_bits", "resid_bits", "lobpcg_iters"):
        assert len({o[k] for o in outs}) == 1, (k, [o[k] for o in outs])


def test_bench_two_ranks_on_one_gpu_functional_rehearsal():
    """bench.py's N = 2 flow end to end (weak-scaled 100x100x200 grid, z-slab sharding, uniform launches, barriers,
    max-over-ranks timing, one JSON line from rank 0) with both ranks on GPU 0 through the peer-memory layer."""
    env = dict(os.environ, MI355OPT_BENCH_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29571", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "100",
           "--warmup", "10"]  # exactly the driver's command line: the roofline leg runs too (on every rank)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    # stdout is the result line and nothing else (gloo / RCCL banners go to stderr)
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[:2000]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 100 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["rows_per_gpu"] == 1_000_000 and "peer-memory" in d["config"]["parallelism"]
    assert d["roofline"]["kernel"] in ("stiefel_hess_fused", "stiefel_spmm_gram", "cg_pupdate")  # by A/B switches
    # r05: the N > 1 line carries the CPU path too (rank 0, after the timed region, on the per-GPU problem, so labelled)
    cb = d["cpu_baseline"]
    assert cb and cb["cores"] == 1 and cb["per_gpu_problem"] and cb["value"] > 0 and "PER-GPU problem" in cb["sample"]
    assert d["cpu_baseline_all_cores"]["cores"] >= 1
    # ... and every exchange layer's solve was held against the CPU oracle, not only against the other ranks
    for leg in d["comm_ab_legs"]:
        oc = leg["oracle_check"]
        assert leg["verified"] and oc and max(oc["s_rel"], oc["alpha_rel"], oc["beta_rel"]) <= 1e-10, leg
    assert [leg["layer"] for leg in d["comm_ab_legs"]] == ["peer", "peer-separate", "peer-separate-rprime"]


def test_bench_falls_back_to_rccl_when_the_peer_memory_layer_fails_verification():
    # (the peer-memory layer is the default exchange layer; RCCL the fallback)
    env = dict(os.environ, MI355OPT_BENCH_FORCE_COMM="1", MI355OPT_BENCH_INJECT_VERIFY_FAILURE="1",
               MI355OPT_COMM="peer")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29574", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "100",
           "--warmup", "10", "--no-cpu-baseline", "--no-roofline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "falling back to RCCL" in r.stderr
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    # (one rank: no halo, `rccl2` is the same launches as `rccl`; whichever timed faster carries the headline)
    assert d["comm_layer"] in ("rccl", "rccl2") and "RCCL" in d["config"]["parallelism"] and d["value"] > 500
    assert [(leg["layer"], leg["verified"]) for leg in d["comm_ab_legs"]] == [
        ("peer", False), ("peer-separate", False), ("peer-separate-rprime", False), ("rccl", True), ("rccl2", True)]


def test_peer_memory_wait_is_bounded_and_fails_loudly():
    import tempfile
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29577", os.path.join(ROOT, "tests", "ipc_fault_worker.py")]
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, IPC_WORKER_OUT=tmp, MI355OPT_IPC_TIMEOUT_MS="300")
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        o = json.load(open(os.path.join(tmp, "rank0.json")))
    assert o["first"] == 2000.0
    assert o["err"] != 0 and 0.2 < o["waited_s"] < 5.0, o
    assert o["stpcg"].startswith("MiError"), o


def _run_cfg4_workers(world, grid, Xb, extra_env=None, port=29580, g_input=None):
    """W processes of tests/cfg4_worker.py on GPU 0 -> (per-rank records, concatenated step, concatenated gradient).
    g_input: the gradient the solves take as their input (its rows per slab) instead of the one the device computes."""
    import tempfile
    nx, ny, nz = grid
    with tempfile.TemporaryDirectory() as tmp:
        np.save(os.path.join(tmp, "Xb.npy"), Xb)
        if g_input is not None:
            np.save(os.path.join(tmp, "g.npy"), np.ascontiguousarray(g_input).ravel())
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(port + world),
               os.path.join(ROOT, "tests", "cfg4_worker.py")]
        env = dict(os.environ, CFG4_WORKER_OUT=tmp, CFG4_GRID=f"{nx},{ny},{nz}")
        env.update(extra_env or {})
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        outs = [json.load(open(os.path.join(tmp, f"rank{k}.json"))) for k in range(world)]
        s_sh = np.concatenate([np.load(os.path.join(tmp, f"s_rank{k}.npy")) for k in range(world)])
        g_sh = np.concatenate([np.load(os.path.join(tmp, f"g_rank{k}.npy")) for k in range(world)])
    return outs, s_sh, g_sh


def test_folded_exchanges_give_the_bits_of_the_exchange_kernels():
    """The scalar exchanges of a sharded STPCG iteration folded into the prologues of their consumer kernels
    (comm_ipc.h fold_exchange_sum) and the halo push folded into the direction kernel (HaloPush): the 3 launches
    per iteration of the single-GPU step instead of 6, against the separate one-workgroup
    exchange kernels (MI355OPT_NO_FOLD=1): the same local reduction, the same rank-order sum -- bit-identical
    scalars and steps, with 3 real peers on one GPU."""
    from optimization_amd import workloads as wl
    grid = (60, 50, 48)
    Xb, _ = wl.stiefel_bench_iterate(*grid, 3, eps=1e-3, seed=7)
    a, sa, _ = _run_cfg4_workers(3, grid, Xb, port=29590)
    b, sb, _ = _run_cfg4_workers(3, grid, Xb, extra_env={"MI355OPT_NO_FOLD": "1"}, port=29594)
    assert all(o["enabled"] and o["ipc_error"] == 0 for o in a + b)
    for k in ("f", "iters", "exit", "M", "rv", "hvp", "alpha", "beta"):
        assert a[0][k] == b[0][k], k
    assert np.array_equal(sa, sb)
    # the late form of the folded push (signal at the end of the direction kernel): the same bits again
    c_, sc, _ = _run_cfg4_workers(3, grid, Xb, extra_env={"MI355OPT_HALO_PUSH_LATE": "1"}, port=29598)
    assert all(o["enabled"] and o["ipc_error"] == 0 and o["comm_kernels"][3] == 0 for o in c_)
    assert np.array_equal(sa, sc) and all(a[0][k] == c_[0][k] for k in ("iters", "exit", "M", "rv", "alpha", "beta"))
    # launches of the exchange layer's own kernels during the solve: (scalar exchanges, halo pushes, pushes folded in)
    it = a[0]["iters"]
    print("comm kernels per solve, folded:", [o["comm_kernels"] for o in a], " separate:", [o["comm_kernels"] for o in b])
    for o in a:   # folded: the set-up's exchanges and the FIRST pass's halo push only; every later push rides along ...
        assert o["comm_kernels"][0] <= 3 and o["comm_kernels"][1] == 1 and o["comm_kernels"][2] >= it - 1, o
        # ... in the early form (the direction kernel starts at the neighbours' rows and signals after its first step)
        assert o["comm_kernels"][3] == o["comm_kernels"][2], o
    for o in b:   # separate kernels: two scalar exchanges and one halo push per iteration
        assert o["comm_kernels"][0] >= 2 * it and o["comm_kernels"][1] >= it and o["comm_kernels"][2] == 0, o


@pytest.mark.parametrize("p,world", [(6, 2), (8, 3)])
def test_wide_rows_sharded_on_one_gpu_vs_oracle(oracle, p, world):
    """r05: Stiefel rows of 6 and 8 doubles on the multi-rank path -- 8-column halo buffers, the wide one-pass Hessian
    with a halo, 24 / 39 partial components through the generic-width reduce-to-slots kernel and the exchange in padded
    chunks, `k_cg_update<., FROM_SLOTS, 24 / 39>`, never folded -- over 2 and 3 real processes on GPU 0 against the CPU
    oracle on the global problem: counts, exit, alpha / beta traces, step to 1e-10; replicated scalars bit-identical."""
    from optimization_amd import workloads as wl
    grid = (30, 26, 12 * world + 1)
    nx, ny, nz = grid
    n = nx * ny * nz
    Xb, _ = wl.stiefel_bench_iterate(nx, ny, nz, p, eps=1e-2, seed=9)
    outs, s_sh, g_sh = _run_cfg4_workers(world, grid, Xb, extra_env={"CFG4_P": str(p)}, port=29620 + p)
    assert all(o["enabled"] and o["ipc_error"] == 0 for o in outs), outs
    for k in ("f", "iters", "exit", "M", "rv", "hvp", "alpha", "beta"):
        assert all(o[k] == outs[0][k] for o in outs), k
    assert all(o["one_pass_launches"] >= outs[0]["iters"] for o in outs)      # the wide one-pass Hessian ran
    assert all(o["comm_kernels"][2] == 0 for o in outs)                       # nothing folded at these widths
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    oprob = oracle.stiefel_rq(n, p, rowptr, col, val)
    go = oracle.eval_grad(oprob, Xb.ravel())
    o = oracle.stpcg_problem(oprob, Xb.ravel(), go, 1e3, max_iterations=50, kappa_fgr=1e-12, theta=1.0, trace_cap=64)
    oracle.free(oprob)
    assert (outs[0]["iters"], outs[0]["exit"]) == (o["iterations"], o["exit_reason"])
    al = np.array([float.fromhex(a) for a in outs[0]["alpha"]])
    be = np.array([float.fromhex(a) for a in outs[0]["beta"]])
    assert rel_err(g_sh, go) < 1e-12
    assert float(np.max(np.abs(al / o["trace"]["alpha"] - 1))) < 1e-9 and float(np.max(np.abs(be / o["trace"]["beta"] - 1))) < 1e-8
    assert rel_err(s_sh, o["s"]) < 1e-10


@pytest.fixture(scope="module")
def cfg4_oracle(oracle, oracle_omp):
    """The CPU oracle (== the reference's templates bit for bit) on BASELINE cfg4 at FULL size -- St(8e6,3), 200^3 grid, the
    50-iteration solve bench.py times -- and the same statements with re-associated sums (OpenMP build, 4 threads): the
    conditioning floor of this comparison.  ~40 s of host time, once per module."""
    from optimization_amd import workloads as wl
    nx = ny = nz = 200
    p, n = 3, nx * ny * nz
    Xb, _ = wl.stiefel_bench_iterate(nx, ny, nz, p, eps=1e-3, seed=7)
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    kw = dict(max_iterations=50, kappa_fgr=1e-12, theta=1.0, trace_cap=64)
    prob = oracle.stiefel_rq(n, p, rowptr, col, val)
    g = oracle.eval_grad(prob, Xb.ravel())
    o = oracle.stpcg_problem(prob, Xb.ravel(), g, 1e3, **kw)
    oracle.free(prob)
    floor = None
    if oracle_omp is not None:
        pm = oracle_omp.stiefel_rq(n, p, rowptr, col, val)
        gm = oracle_omp.eval_grad(pm, Xb.ravel())
        m = oracle_omp.stpcg_problem(pm, Xb.ravel(), gm, 1e3, **kw)
        oracle_omp.free(pm)
        floor = dict(s=rel_err(m["s"], o["s"]),
                     alpha=float(np.max(np.abs(m["trace"]["alpha"] / o["trace"]["alpha"] - 1))),
                     beta=float(np.max(np.abs(m["trace"]["beta"] / o["trace"]["beta"] - 1))))
        assert m["iterations"] == o["iterations"]
    return dict(Xb=Xb, g=g, o=o, floor=floor)


@pytest.mark.parametrize("world,extra", [(2, None), (4, None), (2, "rprime")])
def test_cfg4_sharded_on_one_gpu_matches_the_single_context_solve(world, extra, cfg4_oracle):
    """BASELINE cfg4 (St(8e6,3), 200^3 grid) at its full size, row-sharded over `world` real processes on GPU 0
    through the (default) peer-memory layer: the solve bench.py times -- 50 fused STPCG iterations -- against (i) the
    CPU ORACLE on the same global problem (r04 verdict: not only HIP against HIP): counts, exit, alpha / beta traces and
    the step at max(1e-10, 3 x the measured conditioning floor), and (ii) the single-context device solve: |s|_M and the
    traces to rounding of the re-partitioned sums, the step to 1e-10 relative; every replicated scalar bit-identical on
    all ranks, the one-pass (recurrence-form) Hessian on every rank.  `rprime`: the r'-halo form (separate exchange
    kernels), the transport-independent part of `--comm rccl2`."""
    from optimization_amd import capi, workloads as wl
    nx = ny = nz = 200
    p, n = 3, nx * ny * nz
    Xb = cfg4_oracle["Xb"]
    env = {"MI355OPT_NO_FOLD": "1", "MI355OPT_HALO_RPRIME": "1"} if extra == "rprime" else None
    # (r06) every rank's solve takes its rows of the ORACLE's gradient as the input: parity on identical inputs, plain 1e-10
    outs, s_sh, g_sh = _run_cfg4_workers(world, (nx, ny, nz), Xb, extra_env=env, g_input=cfg4_oracle["g"])
    # (i) the oracle
    oc, fl = cfg4_oracle["o"], cfg4_oracle["floor"] or dict(s=0.0, alpha=0.0, beta=0.0)
    assert (outs[0]["iters"], outs[0]["exit"]) == (oc["iterations"], oc["exit_reason"]) and oc["iterations"] == 50
    al_o = np.array([float.fromhex(a) for a in outs[0]["alpha"]])
    be_o = np.array([float.fromhex(a) for a in outs[0]["beta"]])
    oa = float(np.max(np.abs(al_o / oc["trace"]["alpha"] - 1)))
    ob = float(np.max(np.abs(be_o / oc["trace"]["beta"] - 1)))
    os_ = rel_err(s_sh, oc["s"])
    om = abs(float.fromhex(outs[0]["M"]) - oc["M_norm"]) / oc["M_norm"]
    print(f"cfg4 on one GPU, {world} ranks{' (r-prime halo)' if extra else ''} vs the CPU oracle: s {os_:.2e} (floor "
          f"{fl['s']:.2e}), alpha {oa:.2e} ({fl['alpha']:.2e}), beta {ob:.2e} ({fl['beta']:.2e}), |s|_M {om:.2e}")
    assert rel_err(g_sh, cfg4_oracle["g"]) < 1e-11     # (the gradient the devices computed, which the solve did not use)
    assert os_ <= 1e-10 and om <= 1e-11, (os_, om)
    assert oa <= 1e-10 and ob <= 1e-10, (oa, ob)
    if extra == "rprime":   # one ordinary halo push (first pass), then the rows of r' once per iteration
        assert all(o["comm_kernels"][1] >= 50 and o["comm_kernels"][2] == 0 for o in outs), [o["comm_kernels"] for o in outs]
    assert all(o["enabled"] and o["ipc_error"] == 0 for o in outs), outs
    assert outs[0]["rows"][0] == 0 and outs[-1]["rows"][1] == n
    for k in ("f", "iters", "exit", "M", "rv", "hvp", "alpha", "beta"):  # replicated: the same bits on every rank
        assert all(o[k] == outs[0][k] for o in outs), k
    assert all(o["one_pass_launches"] >= 50 for o in outs)
    # every slab has the window form with COMPUTED far columns (stride = one z-plane), its halo columns included
    for o in outs:
        assert o["window_info"][0] > 0 and o["window_info"][2] == nx * ny and o["window_info"][3] > 0, o["window_info"]
    # the single-context solve
    c = capi.Context(0)
    try:
        rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
        A = c.csr(n, rowptr, col, val)
        del rowptr, col, val
        prob = c.stiefel_rq(A, n, p)
        g, H = prob.model(c.upload(Xb))
        one = c.stpcg(c.upload(cfg4_oracle["g"]), H, Delta=1e3, max_iterations=50, kappa_fgr=1e-12, theta=1.0, trace_cap=64)
        f1 = prob.objective(c.upload(Xb))
        s1, g1 = one["s"].numpy(), g.numpy()
    finally:
        c.close()
    o = outs[0]
    assert (o["iters"], o["exit"]) == (one["iterations"], one["exit_reason"]) and o["iters"] == 50
    assert abs(o["f"] - f1) <= 1e-12 * abs(f1)
    assert np.abs(g_sh - g1).max() <= 1e-12 * np.abs(g1).max()
    M = float.fromhex(o["M"])
    assert abs(M - one["M_norm"]) <= 1e-11 * one["M_norm"]
    al = np.array([float.fromhex(a) for a in o["alpha"]])
    be = np.array([float.fromhex(a) for a in o["beta"]])
    ea = float(np.max(np.abs(al / one["trace"]["alpha"] - 1)))
    eb = float(np.max(np.abs(be / one["trace"]["beta"] - 1)))
    es = float(np.linalg.norm(s_sh - s1) / np.linalg.norm(s1))
    print(f"cfg4 on one GPU, {world} ranks: s {es:.2e}, alpha {ea:.2e}, beta {eb:.2e}, |s|_M {abs(M - one['M_norm']) / M:.2e}")
    assert es <= 1e-10 and ea <= 1e-9 and eb <= 1e-8


@pytest.mark.parametrize("world", [2, 3])
def test_fused_lsqr_row_sharded_matches_the_single_context_solve(world):
    """mi_lsqr on a communicator (r03: it used to refuse one): a row-sharded symmetric sparse operator (built-in CSR
    operator with its halo exchange, and the same product behind a callback), x and b as row slabs, the five
    reductions of a pass completed across the ranks inside their consumers' prologues -- against the single-context
    solve: same iteration count and exit, x to 1e-10, replicated scalars bit-identical on all ranks.  Plain, damped
    and trust-region-bounded solves."""
    import tempfile
    from optimization_amd import capi, workloads as wl
    grid = (24, 20, 6 * world + 1)
    nx, ny, nz = grid
    n = nx * ny * nz
    b = np.random.default_rng(17).normal(size=n)
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    # A = Laplacian + 3.1 I: condition number 5, so that LSQR (CG on A'A) converges in a few dozen passes -- a Lanczos
    # process that runs for hundreds of passes loses orthogonality and two runs with differently grouped sums then
    # agree only to the accuracy of the solution, not to rounding
    shift = 3.0
    val = val + shift * (col == np.repeat(np.arange(n), np.diff(rowptr)))
    c = capi.Context(0)
    try:
        A1 = c.csr(n, rowptr, col, val)
        op1 = c.op_csr(A1, 1)
        for kw in (dict(btol=1e-11, Atol=1e-11, max_iterations=400), dict(lam=0.3, btol=1e-11, Atol=1e-11),
                   dict(Delta=2.0, btol=1e-12, Atol=1e-12), dict(max_iterations=7)):
            one = c.lsqr(op1, op1, c.upload(b), **kw)
            x1 = one["x"].numpy()
            with tempfile.TemporaryDirectory() as tmp:
                np.save(os.path.join(tmp, "b.npy"), b)
                cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                       "--master-addr", "127.0.0.1", "--master-port", str(29620 + world),
                       os.path.join(ROOT, "tests", "lsqr_worker.py")]
                env = dict(os.environ, LSQR_WORKER_OUT=tmp, LSQR_GRID=f"{nx},{ny},{nz}", LSQR_KW=json.dumps(kw),
                           LSQR_SHIFT=str(shift))
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
                assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
                outs = [json.load(open(os.path.join(tmp, f"rank{k}.json"))) for k in range(world)]
                xs = {m: np.concatenate([np.load(os.path.join(tmp, f"x_{m}_rank{k}.npy")) for k in range(world)])
                      for m in ("fused_sub_scaled", "plain_apply")}
            assert all(o["enabled"] and o["ipc_error"] == 0 for o in outs), outs
            for m in ("fused_sub_scaled", "plain_apply"):
                assert all(o[m] == outs[0][m] for o in outs), (kw, m)          # replicated scalars: same bits
                assert (outs[0][m]["iters"], outs[0][m]["exit"]) == (one["iterations"], one["exit_reason"]), (kw, m)
                err = np.abs(xs[m] - x1).max() / max(np.abs(x1).max(), 1e-300)
                print(f"sharded lsqr x{world} {kw} {m}: {one['iterations']} passes, x error {err:.2e}")
                assert err <= 1e-10, (kw, m, err)
                assert abs(float.fromhex(outs[0][m]["xnorm"]) - one["xnorm"]) <= 1e-11 * max(one["xnorm"], 1e-300)
    finally:
        c.close()


# ---- cross-device: these run the first time the suite meets a box with >= 2 GPUs (r04) -----------------------------
def _ngpus():
    try:
        from optimization_amd import capi
        return capi.device_count()
    except Exception:  # noqa: BLE001
        return 0


def _run_xdev(world, layer, one_gpu):
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(29700 + world), os.path.join(ROOT, "tests", "xdev_worker.py")]
        env = dict(os.environ, XDEV_WORKER_OUT=tmp, XDEV_LAYER=layer)
        if one_gpu:
            env["XDEV_ONE_GPU"] = "1"
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        return [json.load(open(os.path.join(tmp, f"rank{k}.json"))) for k in range(world)]


def _check_xdev(outs, world, layer, rccl):
    assert all(o["rccl_nranks"] == (world if rccl else 0) for o in outs)
    if layer not in ("rccl", "rccl2"):
        assert all(o["enabled"] for o in outs), "peer-memory layer did not come up (self-test incl. the folded forms)"
    for o in outs:
        assert o["ipc_error"] == 0
        assert o["dot_err"] < 1e-13 and o["spmm0_equal"] and o["spmm1_equal"] and o["spmm2_equal"], o
        assert o["f_err"] < 1e-12 and o["g_err"] < 1e-12 and o["s_err"] < 1e-10, o
        assert (o["iters"], o["exit"]) == (o["iters_ref"], o["exit_ref"])
        assert (o["b_iters"], o["b_exit"]) == (o["b_iters_ref"], o["b_exit_ref"])
        assert abs(float.fromhex(o["M"]) - o["M_ref"]) <= 1e-10 * abs(o["M_ref"]) and o["same_s"]
        # ... and against the CPU ORACLE (== the reference bit for bit), not only another HIP run (r04 verdict)
        q = o["oracle"]
        assert (o["iters"], o["exit"], o["b_iters"], o["b_exit"]) == (q["iters"], q["exit"], q["b_iters"], q["b_exit"]), q
        assert max(q["s_err"], q["b_s_err"], q["M_err"], q["b_M_err"], q["alpha_err"], q["beta_err"]) <= 1e-10, q
        if layer in ("rccl2", "peer-separate-rprime"):
            assert o["rprime_equal"], "the r'-halo form does not give the bits of the plain exchange"
            if layer == "peer-separate-rprime":   # one ordinary push (the first pass), then one r' push per iteration
                assert o["rprime_comm_kernels"][1] >= o["iters"] + 1, o["rprime_comm_kernels"]
    for k in ("dot", "f", "M", "b_M", "iters", "hvp1", "hvp5", "alpha"):   # replicated: the same bits on every rank
        assert all(o[k] == outs[0][k] for o in outs), k
    if layer == "peer":      # folded: the layer launches nothing of its own per iteration
        assert all(o["comm_kernels"][2] >= o["iters"] - 1 for o in outs), [o["comm_kernels"] for o in outs]


@pytest.mark.parametrize("layer", ["peer", "peer-separate", "peer-separate-rprime"])
def test_cross_device_worker_on_one_gpu(layer):
    """The worker of the cross-device tests below with both ranks on GPU 0 and without RCCL: every line of it except
    the RCCL bring-up runs in the 1-GPU suite, so that the first multi-GPU box meets a script that is known to work."""
    _check_xdev(_run_xdev(2, layer, one_gpu=True), 2, layer, rccl=False)


@pytest.mark.skipif(_ngpus() < 2, reason="needs two GPUs (cross-device RCCL / xGMI peer memory)")
@pytest.mark.parametrize("layer", ["rccl", "rccl2", "peer", "peer-separate", "peer-separate-rprime"])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_cross_device_exchange_layers(world, layer):
    """Rank r on GPU r: RCCL with more than one rank (ncclAllReduce of the partial rows, ncclSend / ncclRecv halo) and the
    peer-memory layer over real xGMI links (folded and with separate exchange kernels), each against the
    single-context solve: sharded products bit-identical to the global one over three exchanges in a row, the fused
    STPCG with the same exits and counts, the step to 1e-10, every replicated scalar bit-identical on all ranks."""
    if _ngpus() < world:
        pytest.skip(f"needs {world} GPUs")
    _check_xdev(_run_xdev(world, layer, one_gpu=False), world, layer, rccl=True)


@pytest.mark.parametrize("n", [2, 4, 8])
def test_bare_bench_command_self_launches_its_ranks(n):
    """`python bench.py --gpus N` with no launcher around it (r04): re-executes itself under torch.distributed.run, one
    rank per GPU -- on this box, with fewer GPUs than ranks, as the one-GPU functional rehearsal -- and prints exactly
    one JSON line with the exchange-layer A/B legs of the N-rank run."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "60", "--warmup", "5",
           "--wakeup-steps", "100", "--ab-steps", "60"] + (["--no-cpu-baseline"] if n != 4 else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[:2000]
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["steps"] == 60 and d["scaling"] == "weak" and d["value"] > 0
    assert d["rehearsal_one_gpu"] == (_ngpus() < n)
    assert d["peer_memory_probe"]["passed"]          # the throwaway-process probe of the peer-memory layer ran first
    legs = {leg["layer"]: leg for leg in d["comm_ab_legs"]}
    assert "peer" in legs and "peer-separate" in legs and "peer-separate-rprime" in legs
    assert d["rehearsal_one_gpu"] or ("rccl" in legs and "rccl2" in legs)
    assert all(leg["oracle_check"] and leg["oracle_check"]["s_rel"] <= 1e-10 for leg in legs.values()), legs
    assert (d["cpu_baseline"] is not None) == (n == 4)
    assert all(leg["verified"] and leg["ipc_error"] == 0 and leg["us_per_step"] > 0 for leg in legs.values()), legs
    assert d["comm_layer"] in legs and d["comm_layer_choice"].startswith("fastest")
    assert legs["peer"]["own_launches_per_step"]["scalar_exchange_kernels"] < 0.2            # folded: none per step
    assert legs["peer-separate"]["own_launches_per_step"]["scalar_exchange_kernels"] >= 2    # two per iteration
    if not d["rehearsal_one_gpu"]:
        assert d["rccl_nranks"] == n


def test_bench_watchdog_delivers_the_kept_line_when_a_later_layer_hangs():
    """N > 1: the headline is measured on the FIRST exchange layer that verifies and kept; a later layer that never
    returns (on a real node: an RCCL collective that hangs -- nothing of ours bounds that) must not cost the line: the
    watchdog prints the kept one with a note and every rank exits 0.  The hang is injected into the second layer of a
    2-rank one-GPU rehearsal."""
    env = dict(os.environ, MI355OPT_BENCH_INJECT_HANG="peer-separate")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "60", "--warmup", "5",
           "--wakeup-steps", "100", "--ab-steps", "60", "--leg-timeout", "15"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[:2000]
    d = json.loads(lines[0])
    assert "did not return" in d["watchdog"] and d["comm_layer"] == "peer" and d["value"] > 0 and d["n_gpus"] == 2
    assert [leg["layer"] for leg in d["comm_ab_legs"]] == ["peer"] and d["comm_ab_legs"][0]["verified"]


def test_bench_goes_on_without_rccl_when_its_probe_fails():
    """N > 1 (r04): RCCL with more than one rank has never run before the driver's scaling bench either, so its
    bring-up (ncclCommInitRank + two all-reduces of known values) is tried in throwaway processes like the peer-memory
    layer's; if that fails or never returns, the run goes on with the peer-memory layer alone and says so.  Here: two
    ranks on one GPU with the RCCL probe switched on -- RCCL refuses the duplicate device."""
    env = dict(os.environ, MI355OPT_BENCH_TRY_RCCL="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "60", "--warmup", "5",
           "--wakeup-steps", "100", "--ab-steps", "60"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[:2000]
    d = json.loads(lines[0])
    assert d["rccl_probe"] == {"passed": False, "seconds": d["rccl_probe"]["seconds"]} and d["rccl_nranks"] == 0
    assert d["peer_memory_probe"]["passed"] and d["comm_layer"] == "peer" and d["value"] > 0
    assert [leg["layer"] for leg in d["comm_ab_legs"]] == ["peer", "peer-separate", "peer-separate-rprime"]
    assert "RCCL probe failed" in r.stderr


def test_rccl_probe_process_with_one_rank():
    """the throwaway process of the RCCL probe itself, world = 1: communicator up, ncclCommCount == 1, exact sums"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--peer-probe", "rccl", "--gpus", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
