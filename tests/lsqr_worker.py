"""One rank of the sharded fused-LSQR test (tests/test_gpu_comm.py): W processes on GPU 0 through the peer-memory layer,
a row-sharded SYMMETRIC sparse operator (so A' is the same operator and its halo exchange serves both products), the
right-hand side and the solution as row slabs.  Writes the rank's slab of x and the replicated scalars."""
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
    out_dir = os.environ["LSQR_WORKER_OUT"]
    nx, ny, nz = (int(v) for v in os.environ["LSQR_GRID"].split(","))
    kw = json.loads(os.environ.get("LSQR_KW", "{}"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    os.environ.setdefault("MI355OPT_MAX_GRID", str(max(16, 192 // world)))  # (see tests/cfg4_worker.py)
    os.environ.setdefault("MI355OPT_IPC_TIMEOUT_MS", "5000")
    c = capi.Context(0)
    enabled = c.enable_peer_memory(world, rank, dist, force=True)
    out = {"rank": rank, "enabled": enabled}
    if enabled:
        z0, z1 = wl.shard_rows(nz, world)[rank]
        n_glob, n = nx * ny * nz, nx * ny * (z1 - z0)
        r0 = nx * ny * z0
        b = np.load(os.path.join(out_dir, "b.npy"))[r0:r0 + n]
        rowptr, col, val = wl.laplacian_3d(nx, ny, nz, z_range=(z0, z1))
        rows_of = np.repeat(np.arange(n, dtype=np.int64) + r0, np.diff(rowptr))
        val = val + float(os.environ.get("LSQR_SHIFT", "0")) * (col == rows_of)   # A + shift I (better conditioned)
        starts = [nx * ny * a for a, _ in wl.shard_rows(nz, world)] + [n_glob]
        dist.barrier()
        A = c.csr_sharded(n_glob, r0, r0 + n, rowptr, col, val, starts)
        op = c.op_csr(A, 1)
        res = {}
        for mode in ("fused_sub_scaled", "plain_apply"):
            o = op if mode == "fused_sub_scaled" else c.op_callback(n, lambda vin, vout, op=op: op.apply(vin, vout))
            r = c.lsqr(o, o, c.upload(b), **kw)
            res[mode] = r
            np.save(os.path.join(out_dir, f"x_{mode}_rank{rank}.npy"), r["x"].numpy())
            out[mode] = dict(iters=r["iterations"], exit=r["exit_reason"], xnorm=float(r["xnorm"]).hex(),
                             Anorm=float(r["Anorm"]).hex(), rbar=float(r["rbar_norm"]).hex(),
                             applies=r["operator_applications"])
        out["ipc_error"] = c.comm_ipc_error()
        dist.barrier()
        c.comm_finalize()
    c.close()
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
