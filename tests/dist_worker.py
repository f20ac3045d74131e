"""Worker of the world_size-2 gloo tests (CPU).  Each rank owns a z-slab of the cfg2/cfg4 workload and
runs the SAME sharded algorithm the GPU build runs: local vector updates, every inner
product = local partial + all-reduce, a nearest-neighbour halo exchange in front of the sparse HVP,
scalar recurrences replicated on every rank.

Product code under test: the host-side shard planning of the C ABI (mi_csr_shard_plan: halo extents and
local column indices -- exactly what mi_csr_create_sharded uses) and optimization_amd.workloads' slab
partition.  The per-rank arithmetic is supplied by the CPU oracle's STPCG driven through distributed
callbacks, so the test checks the distributed ALGORITHM (same decisions on every rank, same iterates
as the unsharded run), not GPU kernels.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import oracle_py  # noqa: E402
from optimization_amd import capi, workloads as wl  # noqa: E402


def allreduce(x):
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64).copy())
    dist.all_reduce(t)
    return t.numpy()


class ShardedStiefel:
    """Row shard of f(X) = 1/2 tr(X'AX) on St(n,p): local slab of A with halo columns."""

    def __init__(self, grid, p, rank, world):
        nx, ny, nz = grid
        self.p, self.rank, self.world = p, rank, world
        slabs = wl.shard_rows(nz, world)
        z0, z1 = slabs[rank]
        self.row_starts = [nx * ny * a for a, _ in slabs] + [nx * ny * nz]
        self.n_glob = nx * ny * nz
        self.n = nx * ny * (z1 - z0)
        rowptr, colg, val = wl.laplacian_3d(nx, ny, nz, z_range=(z0, z1))
        col, self.need_lo, self.need_hi = capi.csr_shard_plan(self.n_glob, world, rank, self.row_starts, colg)
        import scipy.sparse as sps
        self.A = sps.csr_matrix((val, col, rowptr), shape=(self.n, self.n + self.need_lo + self.need_hi))
        # what we must send = what the neighbours need (all-gather of the halo requests)
        req = torch.zeros(2 * world, dtype=torch.int64)
        mine = torch.tensor([self.need_lo, self.need_hi], dtype=torch.int64)
        gathered = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(gathered, mine)
        self.send_lo = int(gathered[rank - 1][1]) if rank > 0 else 0          # rank-1 needs rows above its range
        self.send_hi = int(gathered[rank + 1][0]) if rank + 1 < world else 0  # rank+1 needs rows below its range
        del req

    def halo_exchange(self, V):
        """returns [V; halo_lo; halo_hi] (rows)."""
        p = self.p
        lo = torch.zeros((self.need_lo, p), dtype=torch.float64)
        hi = torch.zeros((self.need_hi, p), dtype=torch.float64)
        ops = []
        Vt = torch.from_numpy(np.ascontiguousarray(V))
        if self.rank > 0:
            if self.send_lo:
                ops.append(dist.P2POp(dist.isend, Vt[:self.send_lo].contiguous(), self.rank - 1))
            if self.need_lo:
                ops.append(dist.P2POp(dist.irecv, lo, self.rank - 1))
        if self.rank + 1 < self.world:
            if self.send_hi:
                ops.append(dist.P2POp(dist.isend, Vt[self.n - self.send_hi:].contiguous(), self.rank + 1))
            if self.need_hi:
                ops.append(dist.P2POp(dist.irecv, hi, self.rank + 1))
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        return np.vstack([V, lo.numpy(), hi.numpy()])

    def spmm(self, V):
        return self.A @ self.halo_exchange(V)

    def halo_of(self, V):
        """the halo rows alone (what the neighbours sent): [halo_lo; halo_hi]"""
        return self.halo_exchange(V)[self.n:]

    def hess_with_halo(self, V, halo):
        """hess(V) with the halo rows of V supplied by the caller instead of exchanged here"""
        Z = self.A @ np.vstack([V, halo]) - V @ self.S
        return Z - self.X @ self.sym_gram(self.X, Z)

    def sym_gram(self, X, Z):
        G = allreduce(X.T @ Z)
        return .5 * (G + G.T)

    def grad(self, X):
        W = self.spmm(X)
        self.S = self.sym_gram(X, W)
        self.X = X
        return W - X @ self.S

    def hess(self, V):
        Z = self.spmm(V) - V @ self.S
        return Z - self.X @ self.sym_gram(self.X, Z)


def main():
    out_path = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    grid, p = (6, 5, 8), 3
    nx, ny, nz = grid
    n_glob = nx * ny * nz
    prob = ShardedStiefel(grid, p, rank, world)
    Xb_glob, _ = wl.stiefel_bench_iterate(nx, ny, nz, p, eps=1e-2, seed=5)
    r0, r1 = prob.row_starts[rank], prob.row_starts[rank + 1]
    X = np.ascontiguousarray(Xb_glob[r0:r1])
    g_own = prob.grad(X)   # (binds S and X for hess; the sharded gradient itself is compared with the oracle's below)
    # The solves take the UNSHARDED oracle's gradient as their input, bit for bit (its rows of this slab): the test is
    # about the distributed ALGORITHM on identical inputs.  (The sharded gradient differs from it by rounding, ~1e-15
    # of |A X|, i.e. ~1e-13 of |g| at this near-optimal iterate, and the nearly singular Hessian multiplies that.)
    O0 = oracle_py.Oracle()
    rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
    gprob = O0.stiefel_rq(n_glob, p, rowptr, col, val)
    g_ref = O0.eval_grad(gprob, Xb_glob.ravel()).reshape(n_glob, p)
    O0.free(gprob)
    g = np.ascontiguousarray(g_ref[r0:r1])
    g_err = float(np.abs(g_own - g).max() / np.abs(g_ref).max())

    n_local = prob.n * p
    decisions = []

    def H(v):
        return prob.hess(v.reshape(prob.n, p)).ravel()

    def ip(a, b):
        val = float(allreduce(np.array([np.dot(a, b)]))[0])
        decisions.append(val)
        return val

    O = oracle_py.Oracle()
    res = O.stpcg(g.ravel(), H, inner=ip, Delta=0.05, max_iterations=40, kappa_fgr=1e-6, theta=.5, trace_cap=64)
    res2 = O.stpcg(g.ravel(), H, inner=ip, Delta=1e3, max_iterations=25, kappa_fgr=1e-10, theta=1.0, trace_cap=64)
    # r'-halo form of the sharded CG (`--comm rccl2`): the halo rows of the new RESIDUAL travel (they are final
    # one step earlier: on RCCL they share the collective of <r,r>), and every rank forms halo(p') = -halo(r') + beta halo(p)
    # from the halo rows of p it holds -- the owner's own expression for those rows, so the halo and every iterate must
    # have the BITS of the form that exchanges p'
    def plain_cg(form, iters=12):
        r = g.copy()
        pd = -r
        sacc = np.zeros_like(r)
        halo = prob.halo_of(pd)
        rr = float(allreduce(np.array([np.sum(r * r)]))[0])
        halos = []
        for _ in range(iters):
            Hp = prob.hess_with_halo(pd, halo)
            alpha = rr / float(allreduce(np.array([np.sum(pd * Hp)]))[0])
            sacc = sacc + alpha * pd
            r = r + alpha * Hp
            rr_new = float(allreduce(np.array([np.sum(r * r)]))[0])
            beta = rr_new / rr
            rr = rr_new
            p_new = -r + beta * pd
            halo = prob.halo_of(p_new) if form == "plain" else -prob.halo_of(r) + beta * halo
            pd = p_new
            halos.append(halo.copy())
        return sacc, halos
    s_plain, h_plain = plain_cg("plain")
    s_rp, h_rp = plain_cg("rprime")
    rprime_same = bool(np.array_equal(s_plain, s_rp) and all(np.array_equal(a, b) for a, b in zip(h_plain, h_rp))
                       and (prob.need_lo + prob.need_hi > 0))
    every_rp = [None] * world
    dist.all_gather_object(every_rp, rprime_same)
    # gather the sharded iterates on rank 0
    pieces = [None] * world
    dist.all_gather_object(pieces, (res["s"], res2["s"], prob.need_lo, prob.need_hi, prob.send_lo, prob.send_hi))
    dec = [None] * world
    dist.all_gather_object(dec, decisions)
    if rank == 0:
        json.dump(dict(
            s1=np.concatenate([pp[0] for pp in pieces]).tolist(), s2=np.concatenate([pp[1] for pp in pieces]).tolist(),
            it1=res["iterations"], exit1=res["exit_reason"], M1=res["M_norm"], it2=res2["iterations"],
            exit2=res2["exit_reason"], M2=res2["M_norm"], alpha2=res2["trace"]["alpha"].tolist(),
            halo=[list(pp[2:]) for pp in pieces], same_scalars=all(d == dec[0] for d in dec), n_local=n_local,
            rprime_same=all(every_rp), g_err=g_err,
            n_glob=n_glob), open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
