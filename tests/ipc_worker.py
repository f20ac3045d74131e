"""One rank of the multi-process peer-memory test (tests/test_gpu_comm.py): W processes, ALL ON GPU 0 (RCCL
refuses duplicate devices, the peer-memory layer does not care), gloo as control plane.  Exercises with real
peers what a one-rank box otherwise cannot: the scalar all-reduces, the halo exchange, the sharded Stiefel
operator, the lockstep enqueue rule of the fused STPCG.  Prints one JSON line per rank."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from optimization_amd import capi, workloads as wl  # noqa: E402  (ROCm before torch)

import torch.distributed as dist  # noqa: E402


def emit(out):
    """one file per rank (stdout of several ranks may interleave)"""
    d = os.environ.get("IPC_WORKER_OUT")
    if d:
        with open(os.path.join(d, f"rank{out['rank']}.json"), "w") as f:
            json.dump(out, f)
    else:
        print(json.dumps(out), flush=True)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    os.environ.setdefault("MI355OPT_MAX_GRID", str(max(16, 192 // world)))  # (see tests/cfg4_worker.py)
    c = capi.Context(0)
    enabled = c.enable_peer_memory(world, rank, dist, force=True)
    out = {"rank": rank, "enabled": enabled}
    if not enabled:
        emit(out)
        return
    nx, ny, nz, p = 16, 12, 4 * world + 1, 3          # uneven slabs
    n = nx * ny * nz
    slabs = wl.shard_rows(nz, world)
    starts = [nx * ny * a for a, _ in slabs] + [n]
    r0, r1 = starts[rank], starts[rank + 1]
    rng = np.random.default_rng(11)
    V = rng.normal(size=(n, p))
    W = rng.normal(size=(n, p))

    # A. dot products: local partial rows -> one exchange kernel -> identical sums everywhere
    d = c.upload(V[r0:r1]).dot(c.upload(W[r0:r1]))
    out["dot_err"] = abs(d - float(np.sum(V * W))) / abs(float(np.sum(V * W)))
    out["dot"] = d

    # B. sharded SpMM: halo rows arrive by peer stores
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    import scipy.sparse as sps
    Ag = sps.csr_matrix((val, col, rowptr), shape=(n, n))
    rp, colg, vl = wl.laplacian_3d(nx, ny, nz, z_range=slabs[rank])
    A = c.csr_sharded(n, r0, r1, rp, colg, vl, starts)
    Y = A.spmm(p, c.upload(V[r0:r1])).numpy().reshape(r1 - r0, p)
    out["spmm_err"] = float(np.abs(Y - (Ag @ V)[r0:r1]).max())
    Y2 = A.spmm(p, c.upload(W[r0:r1])).numpy().reshape(r1 - r0, p)   # second exchange: flags must advance
    out["spmm2_err"] = float(np.abs(Y2 - (Ag @ W)[r0:r1]).max())

    # C. sharded Stiefel model + fused STPCG against a single-process solve of the global problem
    Xb, _ = wl.stiefel_bench_iterate(nx, ny, nz, p, eps=1e-2, seed=5)
    prob = c.stiefel_rq(A, r1 - r0, p)
    X = c.upload(Xb[r0:r1])
    out["f"] = prob.objective(X)
    g, H = prob.model(X)
    res = {}
    for ra in (1, 5):
        r = c.stpcg(g, H, Delta=1e3, max_iterations=25, kappa_fgr=1e-9, theta=1.0, run_ahead=ra)
        res[ra] = r
    r = res[1]
    out.update(iters=r["iterations"], exit=r["exit_reason"], M=r["M_norm"], hvp1=res[1]["hvp_calls"],
               hvp5=res[5]["hvp_calls"], same_s=bool(np.array_equal(res[1]["s"].numpy(), res[5]["s"].numpy())))
    rb = c.stpcg(g, H, Delta=1e-3, max_iterations=25)      # boundary exit
    out.update(b_iters=rb["iterations"], b_exit=rb["exit_reason"], b_M=rb["M_norm"])
    Ynew = c.stiefel_retract(r1 - r0, p, X, rb["s"]).numpy().reshape(r1 - r0, p)
    s_loc = r["s"].numpy().reshape(r1 - r0, p)
    g_loc = g.numpy().reshape(r1 - r0, p)

    # D. row-sharded LOBPCG: Gram and residual norms are local products + an all-reduce, the operator
    #    is the sharded column-major SpMM, Rayleigh-Ritz is replicated
    k = 6
    m = r1 - r0
    Sg = rng.normal(size=(n, k))
    Tg = rng.normal(size=(n, k))
    Sd = c.upload(np.asfortranarray(Sg[r0:r1]).ravel(order="F"))
    Td = c.upload(np.asfortranarray(Tg[r0:r1]).ravel(order="F"))
    G = c.lobpcg_gram(m, Sd, k, Td, k)
    Gref = Sg.T @ Tg
    out["gram_err"] = float(np.abs(G - Gref).max() / np.abs(Gref).max())
    out["gram_bits"] = G.tobytes().hex()
    th = np.linspace(.5, 2.0, k)
    _, rn, xn = c.lobpcg_residual(m, k, Sd, Td, Sd, th)
    rn_ref = np.linalg.norm(Sg - Tg * th[None, :], axis=0)
    out["resid_err"] = float(np.abs(rn - rn_ref).max() / rn_ref.max())
    out["xnorm_err"] = float(np.abs(xn - np.linalg.norm(Sg, axis=0)).max())
    out["resid_bits"] = rn.tobytes().hex()
    Yc = A.spmm_colmajor(k, Sd).numpy().reshape(k, m).T
    out["spmm_colmajor_err"] = float(np.abs(Yc - (Ag @ Sg)[r0:r1]).max())
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import harness_py
    hd = harness_py.DeviceHarness()
    X0g = np.linalg.qr(rng.normal(size=(n, k)))[0]
    lob = hd.lobpcg_on(c, A, m, k, 4, X0g[r0:r1], max_iters=400, tau=1e-7)
    out.update(lobpcg_rc=lob["rc"], lobpcg_err=lob["err"], lobpcg_iters=lob["iterations"], lobpcg_nconv=lob["nconv"],
               lobpcg_theta=[float(v) for v in lob["theta"]])
    out["ipc_error"] = c.comm_ipc_error()

    # reference: the same problem on one plain context (every rank computes it: the GPU is shared anyway)
    c1 = capi.Context(0)
    A1 = c1.csr(n, rowptr, col, val)
    prob1 = c1.stiefel_rq(A1, n, p)
    X1 = c1.upload(Xb)
    out["f_ref"] = prob1.objective(X1)
    g1, H1 = prob1.model(X1)
    r1s = c1.stpcg(g1, H1, Delta=1e3, max_iterations=25, kappa_fgr=1e-9, theta=1.0)
    r1b = c1.stpcg(g1, H1, Delta=1e-3, max_iterations=25)
    Y1 = c1.stiefel_retract(n, p, X1, r1b["s"]).numpy().reshape(n, p)
    sref = r1s["s"].numpy().reshape(n, p)
    gref = g1.numpy().reshape(n, p)
    out.update(iters_ref=r1s["iterations"], exit_ref=r1s["exit_reason"], M_ref=r1s["M_norm"],
               b_iters_ref=r1b["iterations"], b_exit_ref=r1b["exit_reason"],
               g_err=float(np.abs(g_loc - gref[r0:r1]).max() / np.abs(gref).max()),
               s_err=float(np.abs(s_loc - sref[r0:r1]).max() / np.abs(sref).max()),
               retract_err=float(np.abs(Ynew - Y1[r0:r1]).max()))
    lob1 = hd.lobpcg_on(c1, A1, n, k, 4, X0g, max_iters=400, tau=1e-7)
    w = np.linalg.eigvalsh(Ag.toarray())[:4]
    out.update(lobpcg_iters_ref=lob1["iterations"], lobpcg_nconv_ref=lob1["nconv"],
               lobpcg_theta_ref=[float(v) for v in lob1["theta"]], lobpcg_exact=[float(v) for v in w],
               lobpcg_x_err=float(np.abs(np.abs(lob["X"]) - np.abs(lob1["X"][r0:r1])).max()))
    c1.close()
    dist.barrier()
    c.comm_finalize()
    c.close()
    emit(out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
