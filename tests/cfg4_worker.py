"""One rank of the BASELINE cfg4 rehearsal on ONE GPU (tests/test_gpu_comm.py): Stiefel(8e6, 3) on the 200^3 grid,
row-sharded in z-slabs over W processes that all sit on GPU 0 and exchange through the peer-memory layer (RCCL
refuses duplicate devices).  Every rank builds its slab of the matrix and its rows of the bench iterate, runs the
exact solve bench.py times (50 fused STPCG iterations, kappa_fgr 1e-12, theta 1, Delta 1e3) and writes its rows of
the step plus the replicated scalars; the test compares them with the single-context solve of the whole problem."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from optimization_amd import capi, workloads as wl  # noqa: E402  (ROCm before torch)

import torch.distributed as dist  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out_dir = os.environ["CFG4_WORKER_OUT"]
    nx, ny, nz = (int(v) for v in os.environ.get("CFG4_GRID", "200,200,200").split(","))
    p = int(os.environ.get("CFG4_P", "3"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # All ranks share ONE GPU and the consumer kernels wait for their peers in their prologue; the push a waiter needs
    # comes from workgroup 0 of the SAME kernel of every other rank, so that workgroup must get a slot while the
    # kernels of all the other ranks sit resident and waiting: (world - 1) * grid + 1 <= 256 (one 1024-thread workgroup
    # of these kernels per CU).  On a real node every rank has a GPU to itself.
    os.environ.setdefault("MI355OPT_MAX_GRID", str(max(16, 192 // world)))
    os.environ.setdefault("MI355OPT_IPC_TIMEOUT_MS", "5000")
    c = capi.Context(0)
    enabled = c.enable_peer_memory(world, rank, dist, force=True)
    out = {"rank": rank, "enabled": enabled}
    if enabled:
        z0, z1 = wl.shard_rows(nz, world)[rank]
        n_glob, n = nx * ny * nz, nx * ny * (z1 - z0)
        r0 = nx * ny * z0
        Xb = np.load(os.path.join(out_dir, "Xb.npy"), mmap_mode="r")[r0:r0 + n]
        shift = float(os.environ.get("CFG4_SHIFT", "0.1"))
        rowptr, col, val = wl.laplacian_3d(nx, ny, nz, shift=shift, z_range=(z0, z1))
        starts = [nx * ny * a for a, _ in wl.shard_rows(nz, world)] + [n_glob]
        dist.barrier()
        A = c.csr_sharded(n_glob, r0, r0 + n, rowptr, col, val, starts)
        out["window_info"] = list(A.window_info())
        prob = c.stiefel_rq(A, n, p)
        X = c.upload(np.ascontiguousarray(Xb))
        out["f"] = prob.objective(X)
        g, H = prob.model(X)
        gsolve = g
        if os.path.exists(os.path.join(out_dir, "g.npy")):
            # the solve's INPUT is the caller's gradient, bit for bit (its rows of this slab): parity of the solver on
            # identical inputs, with the device's own gradient still written out below
            gsolve = c.upload(np.ascontiguousarray(np.load(os.path.join(out_dir, "g.npy"), mmap_mode="r")[p * r0:p * (r0 + n)]))
        c.ktime_enable("stiefel_hess_fused", True)
        k0 = c.comm_kernel_launches()
        maxit = int(os.environ.get("CFG4_MAXIT", "50"))
        r = c.stpcg(gsolve, H, Delta=float(os.environ.get("CFG4_DELTA", "1e3")), max_iterations=maxit, kappa_fgr=1e-12,
                    theta=1.0, trace_cap=maxit + 2)
        out["one_pass_launches"] = c.ktime_read("stiefel_hess_fused")[0]
        # kernels the exchange layer launched during the solve: (scalar exchanges, halo pushes, halo pushes folded in)
        out["comm_kernels"] = [b - a for a, b in zip(k0, c.comm_kernel_launches())]
        np.save(os.path.join(out_dir, f"s_rank{rank}.npy"), r["s"].numpy())
        np.save(os.path.join(out_dir, f"g_rank{rank}.npy"), g.numpy())
        out.update(rows=[r0, r0 + n], iters=r["iterations"], exit=r["exit_reason"], M=float(r["M_norm"]).hex(),
                   rv=float(r["rv_final"]).hex(), hvp=r["hvp_calls"],
                   alpha=[float(a).hex() for a in r["trace"]["alpha"]],
                   beta=[float(a).hex() for a in r["trace"]["beta"]], ipc_error=c.comm_ipc_error())
        dist.barrier()
        c.comm_finalize()
    c.close()
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
