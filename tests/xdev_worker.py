"""One rank of the CROSS-DEVICE tests (tests/test_gpu_comm.py, skipped on a box with fewer GPUs than ranks): rank r on
GPU r, a real RCCL communicator (ncclCommInitRank with world > 1: ncclAllReduce of the partial rows, ncclSend/ncclRecv
halo) and the peer-memory layer over real xGMI links, one exchange layer per run (XDEV_LAYER = rccl | rccl2 | peer |
peer-separate | peer-separate-rprime; the two r'-halo forms are ALSO held bit for bit against the plain form of their
layer).  Every rank also solves the global problem on a plain context of its own GPU, and with the CPU oracle
(oracle/liboracle.so == the reference's templates bit for bit): the sharded solve answers to the reference, not only to
another HIP run."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np  # noqa: E402

from optimization_amd import capi, workloads as wl  # noqa: E402  (ROCm before torch)

import torch.distributed as dist  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    layer_asked = os.environ.get("XDEV_LAYER", "rccl")
    rprime = layer_asked in ("rccl2", "peer-separate-rprime")
    layer = {"rccl2": "rccl", "peer-separate-rprime": "peer-separate"}.get(layer_asked, layer_asked)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # XDEV_ONE_GPU=1: the same script with every rank on GPU 0 and without RCCL (it refuses duplicate devices) -- so that
    # the 1-GPU suite executes every line of this worker except the RCCL bring-up before a multi-GPU box ever does
    one_gpu = os.environ.get("XDEV_ONE_GPU") == "1"
    dev = 0 if one_gpu else int(os.environ.get("LOCAL_RANK", rank))
    if one_gpu:
        os.environ.setdefault("MI355OPT_MAX_GRID", str(max(16, 192 // world)))
        os.environ.setdefault("MI355OPT_IPC_TIMEOUT_MS", "5000")
    c = capi.Context(dev)
    if not one_gpu:
        uid = [c.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        c.comm_init(world, rank, uid[0])
    out = {"rank": rank, "layer": layer_asked, "rccl_nranks": c.comm_rccl_count(), "device": c.device_name()}
    if layer != "rccl":
        out["enabled"] = c.enable_peer_memory(world, rank, dist, force=True)   # incl. the folded-form self-test
        if out["enabled"]:
            c.comm_ipc_fold(layer == "peer")
    nx, ny, nz, p = 40, 36, 8 * world + 3, 3          # uneven slabs, halo = one 40 x 36 plane
    n = nx * ny * nz
    slabs = wl.shard_rows(nz, world)
    starts = [nx * ny * a for a, _ in slabs] + [n]
    r0, r1 = starts[rank], starts[rank + 1]
    rng = np.random.default_rng(23)
    V = rng.normal(size=(n, p))
    W = rng.normal(size=(n, p))
    # scalar all-reduce
    d = c.upload(V[r0:r1]).dot(c.upload(W[r0:r1]))
    out["dot"] = float(d).hex()
    out["dot_err"] = abs(d - float(np.sum(V * W))) / abs(float(np.sum(V * W)))
    # halo exchange: the sharded product has the bits of the global one (entry-by-entry rounding in storage order)
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    rp, colg, vl = wl.laplacian_3d(nx, ny, nz, z_range=slabs[rank])
    dist.barrier()
    A = c.csr_sharded(n, r0, r1, rp, colg, vl, starts)
    c1 = capi.Context(dev)
    A1 = c1.csr(n, rowptr, col, val)
    for k, F in enumerate((V, W, V + W)):   # three exchanges in a row: flags / double buffering must advance
        Y = A.spmm(p, c.upload(F[r0:r1])).numpy()
        Y1 = A1.spmm(p, c1.upload(F)).numpy().reshape(n, p)[r0:r1].ravel()
        out[f"spmm{k}_equal"] = bool(np.array_equal(Y, Y1))
    # sharded Stiefel model + fused STPCG (residual exit with run-ahead 1 and 5, boundary exit)
    Xb, _ = wl.stiefel_bench_iterate(nx, ny, nz, p, eps=1e-2, seed=5)
    prob = c.stiefel_rq(A, r1 - r0, p)
    X = c.upload(Xb[r0:r1])
    out["f"] = float(prob.objective(X)).hex()
    g_dev, H = prob.model(X)
    # (r06) the solves take the CPU oracle's gradient, bit for bit, as their input -- parity on identical inputs; the
    # gradient the devices computed is compared with it separately (g_err / oracle.g_err)
    import oracle_py
    O = oracle_py.Oracle()
    oprob = O.stiefel_rq(n, p, rowptr, col, val)
    go = O.eval_grad(oprob, Xb.ravel())
    g = c.upload(np.ascontiguousarray(go.reshape(n, p)[r0:r1]).ravel())
    k0 = c.comm_kernel_launches()
    res = {ra: c.stpcg(g, H, Delta=1e3, max_iterations=40, kappa_fgr=1e-9, theta=1.0, run_ahead=ra, trace_cap=64)
           for ra in (1, 5)}
    k1 = c.comm_kernel_launches()
    r = res[1]
    rb = c.stpcg(g, H, Delta=1e-3, max_iterations=25)
    if rprime:
        # the r'-halo form (2 dependent collectives per iteration on RCCL): halo rows of r' travel, halo(p') is formed
        # locally by the expression the owner evaluates -- the same bits as exchanging p', residual and boundary exits
        c.set_option("HALO_RPRIME", 1)
        k2 = c.comm_kernel_launches()
        r2 = c.stpcg(g, H, Delta=1e3, max_iterations=40, kappa_fgr=1e-9, theta=1.0, run_ahead=1, trace_cap=64)
        k3 = c.comm_kernel_launches()
        rb2 = c.stpcg(g, H, Delta=1e-3, max_iterations=25)
        c.set_option("HALO_RPRIME", 0)
        out["rprime_equal"] = bool(
            np.array_equal(r2["s"].numpy(), r["s"].numpy()) and np.array_equal(rb2["s"].numpy(), rb["s"].numpy()) and
            (r2["iterations"], r2["exit_reason"], r2["M_norm"]) == (r["iterations"], r["exit_reason"], r["M_norm"]) and
            np.array_equal(r2["trace"]["alpha"], r["trace"]["alpha"]) and np.array_equal(r2["trace"]["beta"], r["trace"]["beta"]))
        out["rprime_comm_kernels"] = [b - a for a, b in zip(k2, k3)]
    out.update(iters=r["iterations"], exit=r["exit_reason"], M=float(r["M_norm"]).hex(), hvp1=res[1]["hvp_calls"],
               hvp5=res[5]["hvp_calls"], same_s=bool(np.array_equal(res[1]["s"].numpy(), res[5]["s"].numpy())),
               alpha=[float(a).hex() for a in r["trace"]["alpha"]], b_iters=rb["iterations"], b_exit=rb["exit_reason"],
               b_M=float(rb["M_norm"]).hex(), comm_kernels=[b - a for a, b in zip(k0, k1)],
               ipc_error=c.comm_ipc_error())
    # reference on this rank's own GPU
    prob1 = c1.stiefel_rq(A1, n, p)
    X1 = c1.upload(Xb)
    f1 = prob1.objective(X1)
    g1, H1 = prob1.model(X1)
    g1_in = c1.upload(go)
    r1s = c1.stpcg(g1_in, H1, Delta=1e3, max_iterations=40, kappa_fgr=1e-9, theta=1.0)
    r1b = c1.stpcg(g1_in, H1, Delta=1e-3, max_iterations=25)
    sref = r1s["s"].numpy().reshape(n, p)
    out.update(f_err=abs(float.fromhex(out["f"]) - f1) / abs(f1),
               g_err=float(np.abs(g_dev.numpy().reshape(-1, p) - g1.numpy().reshape(n, p)[r0:r1]).max() / np.abs(g1.numpy()).max()),
               s_err=float(np.abs(r["s"].numpy().reshape(-1, p) - sref[r0:r1]).max() / np.abs(sref).max()),
               iters_ref=r1s["iterations"], exit_ref=r1s["exit_reason"], M_ref=r1s["M_norm"],
               b_iters_ref=r1b["iterations"], b_exit_ref=r1b["exit_reason"])
    c1.close()
    # the CPU oracle on the global problem (a 4e4-row grid: milliseconds)
    o = O.stpcg_problem(oprob, Xb.ravel(), go, 1e3, max_iterations=40, kappa_fgr=1e-9, theta=1.0, trace_cap=64)
    ob = O.stpcg_problem(oprob, Xb.ravel(), go, 1e-3, max_iterations=25)
    so, sbo = o["s"].reshape(n, p), ob["s"].reshape(n, p)
    out.update(oracle=dict(
        g_err=float(np.abs(g_dev.numpy().reshape(-1, p) - go.reshape(n, p)[r0:r1]).max() / np.abs(go).max()),
        iters=o["iterations"], exit=o["exit_reason"], b_iters=ob["iterations"], b_exit=ob["exit_reason"],
        M_err=abs(r["M_norm"] - o["M_norm"]) / o["M_norm"], b_M_err=abs(rb["M_norm"] - ob["M_norm"]) / ob["M_norm"],
        s_err=float(np.abs(r["s"].numpy().reshape(-1, p) - so[r0:r1]).max() / np.abs(so).max()),
        b_s_err=float(np.abs(rb["s"].numpy().reshape(-1, p) - sbo[r0:r1]).max() / np.abs(sbo).max()),
        alpha_err=float(np.max(np.abs(r["trace"]["alpha"] / o["trace"]["alpha"] - 1))),
        beta_err=float(np.max(np.abs(r["trace"]["beta"] / o["trace"]["beta"] - 1)))))
    O.free(oprob)
    dist.barrier()
    c.comm_finalize()
    c.close()
    with open(os.path.join(os.environ["XDEV_WORKER_OUT"], f"rank{rank}.json"), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
