#!/usr/bin/env python3
"""bench.py -- TNT Steihaug-CG HVP + inner-product throughput on MI355X (BASELINE.json metric).

One "step" = one COMPLETED inner iteration of the fused device STPCG (reference:
LinearAlgebra/IterativeSolvers.h:285-422): one Riemannian Hessian-vector product, the curvature and
residual inner products, the s/r/p updates and the device-side scalar recurrences + branch tests,
on BASELINE cfg2: Stiefel(1e6 x world_size, 3), f(X) = .5 tr(X'AX), A = 7-point Laplacian + 0.1 I
(100^3 per GPU, z-slab sharded), fp64, at a point near the minimiser (where TNT spends its inner
iterations).  Steps are executed as STPCG solves of max_TPCG_iterations = 50 (cfg2), i.e. the timed
region also contains each solve's initialisation and final read-back.

value = steps x (compulsory HBM bytes of the three kernels an iteration runs) / time: a bandwidth that can be
held against the 8 TB/s HBM peak (hbm_roofline_frac_whole_step = value / n_gpus / 8000).  The fused kernels
move far fewer bytes than SURVEY.md 8(d)'s accounting of the reference schedule (88 N + 12 nnz + 4 (n+1) +
16 n p + 56 N per iteration); the same time priced by those bytes is kept only as
reference_schedule_bytes_per_second_not_a_bandwidth.

Extra legs (rank 0 of a 1-GPU run, each with its own moved-bytes fraction): the same workload with the
matrix in plain 12-byte entries instead of the 4-byte value-indexed copy (`generic_csr_leg`, with its own roofline block), St(8e6,3) on
one GPU -- beyond the 256 MiB Infinity Cache -- (`beyond_cache_leg`), and the CPU baselines: the reference's
own single-threaded path and the oracle's OpenMP build on all host cores this process may use.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--wakeup-steps S] [--no-cpu-baseline] [--no-roofline]
                       [--no-legs] [--comm auto|peer|peer-separate|rccl] [--ab-steps A]
       N > 1: one rank per GPU.  Either launched by torch.distributed.run (the driver's way: RANK / LOCAL_RANK /
       WORLD_SIZE / MASTER_* in the environment) or BARE -- `python bench.py --gpus N` re-launches itself under
       torch.distributed.run on 127.0.0.1 (on a box with fewer than N GPUs as a functional rehearsal with all ranks on
       GPU 0, flagged `rehearsal_one_gpu` in the JSON line: its timings mean nothing).  torch.distributed/gloo is the
       control plane only (rendezvous, barriers, max-over-ranks time); the data path's exchanges go through the
       peer-memory layer (hipIpc arenas over xGMI) or RCCL.  Before the timed region an N > 1 run times every exchange
       layer that verifies (`comm_ab_legs`: peer-memory with the exchanges folded into the solver's kernels, the same
       with separate exchange kernels, RCCL; --ab-steps each) and the headline runs on the fastest one (--comm /
       MI355OPT_COMM force one), so that ONE run on an 8-GPU node yields the curve and says why.
"""
import argparse
import json
import os
import sys
import time

# One launching thread per rank is all this benchmark needs on the host.  numpy's BLAS and torch's OpenMP pools
# size themselves to the machine's 256 cores and spin after every parallel region; inside a container with a CPU
# quota (16 CPUs on the GPU boxes of this pool) that throttles the whole cgroup and the kernel-launching thread
# with it: measured 160-206 us/step instead of 68 in the multi-rank code path.  Must precede the imports.
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# stdout carries exactly one line, the JSON result of rank 0: everything else that writes to file descriptor 1
# (gloo's connection notice, RCCL's version banner at exit, stray prints) is sent to stderr instead
_RESULT_FD = os.dup(1)
os.dup2(2, 1)
sys.stdout = sys.stderr

import numpy as np  # noqa: E402

from optimization_amd import capi, workloads as wl  # noqa: E402  (loads ROCm before torch)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"
TPCG = 50              # cfg2: max_TPCG_iterations


def kernel_bytes(n, nnz, p, packed=True):
    """Per-launch compulsory HBM bytes of each hot kernel (every operand streamed once)."""
    N = n * p
    direct = os.environ.get("MI355OPT_DIRGRAM_DIRECT", "0") == "1"  # Gram rows formed by the direction kernel
    # the one-pass Hessian streams the value-indexed packed copy of A (4 B per entry: the bench matrix has 2
    # distinct values) unless that format is switched off
    a_fused = (4 if packed else 12) * nnz + 4 * (n + 1)
    if packed and os.environ.get("MI355OPT_WORDS16", "0") == "1":
        # opt-in 16-bit window words: eight 2-byte entries per row (padded to whole 64-row slices), no row pointers
        a_fused = 16 * ((n + 63) // 64 * 64)
    kb = {
        # A; V gathered, X read; Hp written (the projection matrix is known before the pass); recurrence
        # form: + Y read for the Gram of the output
        "stiefel_hess_fused": a_fused + 8 * (3 if direct else 4) * N,
        "stiefel_spmm_gram": 12 * nnz + 4 * (n + 1) + 8 * 3 * N,   # A; V gathered, X read; Z written
        "stiefel_finish_dots": 8 * 4 * N,                           # X, Z, V read; Hp written
        "cg_update": 8 * 3 * N,                                     # r,Hp read; r written
        # v(=r), p, s read; p, s written; direct form: + X, Y read for the next direction's Gram rows
        "cg_pupdate": 8 * (7 if direct else 5) * N,
    }
    return kb


HOT = ["stiefel_hess_fused", "stiefel_spmm_gram", "stiefel_finish_dots", "cg_update", "cg_pupdate"]
AUX = ["stiefel_gram_reduce", "cg_scalar_a", "cg_scalar_b"]


def timed_kernels(ctx, g, H, s_out, steps, kb):
    """`steps` more iterations with a HIP-event pair around every hot kernel (on the stream they are launched on).
    Returns (per-kernel dict, moved bytes per step)."""
    names = HOT + AUX
    for k in names:
        ctx.ktime_enable(k, True)
    ctx.ktime_reset()
    run_steps(ctx, g, H, s_out, steps)
    per = {}
    for k in names:
        cnt, ms = ctx.ktime_read(k)
        per[k] = {"launches": cnt, "avg_us": 1e3 * ms / max(cnt, 1)}
        ctx.ktime_enable(k, False)
    for k in HOT:
        if per[k]["launches"]:
            per[k]["bytes"] = kb[k]
            per[k]["GBps"] = kb[k] / (per[k]["avg_us"] * 1e-6) / 1e9
    moved = sum(kb[k] * per[k]["launches"] for k in HOT) / steps
    return per, moved


def wall_steps(ctx, g, H, s_out, steps, warmup):
    """time `steps` iterations (after `warmup`): seconds"""
    if warmup:
        run_steps(ctx, g, H, s_out, warmup)
    ctx.sync()
    t0 = time.perf_counter()
    run_steps(ctx, g, H, s_out, steps)
    ctx.sync()
    return time.perf_counter() - t0


def extra_leg(ctx, nx, p, steps, warmup, packed, label):
    """One more single-GPU leg: St(nx^3, p), matrix packed or plain, its own moved-bytes bandwidth."""
    n = nx ** 3
    ctx.set_option("NO_PACKED", 0 if packed else 1)   # (acts at matrix creation)
    try:
        rowptr, col, val = wl.laplacian_3d(nx, nx, nx)
        A = ctx.csr(n, rowptr, col, val)
    finally:
        ctx.set_option("NO_PACKED", 1 if os.environ.get("MI355OPT_NO_PACKED", "0") == "1" else 0)
    nnz = int(rowptr[-1])
    del rowptr, col, val
    prob = ctx.stiefel_rq(A, n, p)
    Xb, _ = wl.stiefel_bench_iterate(nx, nx, nx, p, eps=1e-3, seed=7)
    X = ctx.upload(Xb)
    del Xb
    g, H = prob.model(X)
    s_out = ctx.vec(n * p)
    dt = wall_steps(ctx, g, H, s_out, steps, warmup)
    kb = kernel_bytes(n, nnz, p, packed=packed)
    per, moved = timed_kernels(ctx, g, H, s_out, min(steps, 100), kb)
    N = n * p
    ws = 6 * 8 * N + (4 if packed else 12) * nnz
    ran = [k for k in HOT if per[k]["launches"]]
    dom = max(ran, key=lambda k: per[k]["avg_us"] * per[k]["launches"])
    out = {"workload": label, "rows": n, "nnz": nnz, "packed_matrix": packed, "steps": steps,
           "us_per_step": 1e6 * dt / steps, "moved_bytes_per_step": moved,
           "value": steps * moved / dt / 1e9, "unit": "GB/s",
           "hbm_roofline_frac_whole_step": steps * moved / dt / 1e9 / HBM_PEAK_GBS,
           "working_set_MB": ws / 1e6, "infinity_cache_MB": 268.4,
           # the dominant kernel of THIS leg on its own compulsory bytes, raw HIP event pairs (each pair also times
           # its two records, ~1.7 us: the figure is a lower bound on the kernel's rate)
           "roofline": {"bound": "hbm", "kernel": dom, "achieved": per[dom]["GBps"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": per[dom]["GBps"] / HBM_PEAK_GBS, "traffic": None,
                        "algorithmic_bytes_per_launch": kb[dom], "avg_launch_us_event_pairs": per[dom]["avg_us"],
                        "timing": "raw HIP event pairs around every launch"},
           "sum_kernel_us_over_step_us_event_pairs": sum(per[k]["avg_us"] * per[k]["launches"] for k in ran)
                                                     / min(steps, 100) / (1e6 * dt / steps),
           "kernels": {k: v for k, v in per.items() if v["launches"]}}
    del g, H, s_out, X, prob, A
    return out


def generic_leg_in_its_own_process(n, p, packed, label):
    """The cfg2 workload through the OTHER matrix format in a process of its own (`python bench.py --no-legs` with
    MI355OPT_NO_PACKED flipped): the gather-based Hessian kernel of the generic CSR path depends on where the vectors
    happen to lie -- inside this process, behind the main workload's allocations, the same leg measured
    68 ... 74.5 us/step from run to run; a fresh process, which is what a user of that path has, gives 66.9-67.6."""
    import subprocess
    env = dict(os.environ, MI355OPT_NO_PACKED="1" if packed else "0")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--steps", "200", "--warmup", "20", "--no-legs",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError("generic leg subprocess failed: " + r.stderr[-500:])
    d = json.loads(r.stdout.strip().splitlines()[-1])
    moved = d["config"]["moved_bytes_per_step_per_gpu"]
    rf = d["roofline"] or {}
    return {"workload": label, "rows": n, "nnz": d["config"]["nnz_per_gpu"], "packed_matrix": d["config"]["packed_matrix"],
            "steps": d["steps"], "us_per_step": 1e3 * d["ms_per_step"], "moved_bytes_per_step": moved,
            "value": d["value"], "unit": "GB/s", "hbm_roofline_frac_whole_step": d["value"] / HBM_PEAK_GBS,
            "process": "its own (python bench.py --no-legs --no-cpu-baseline, MI355OPT_NO_PACKED flipped)",
            # the dominant kernel of THIS leg on its own compulsory bytes: event pairs net of their own cost, as in the
            # main line's roofline block (raw figure next to it)
            "roofline": {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                "algorithmic_bytes_per_launch", "avg_launch_us",
                                                "avg_launch_us_event_pairs", "frac_event_pairs_uncorrected", "timing")},
            "sum_kernel_us_over_step_us_event_pairs": rf.get("sum_kernel_us_over_step_us_event_pairs"),
            "kernels": {k: v for k, v in (rf.get("kernels") or {}).items() if v.get("launches")}}


def run_steps(ctx, g, H, s_out, steps, last=None):
    """Execute exactly `steps` completed STPCG inner iterations; returns number of solves (last: a dict that receives
    the result record of the last solve)."""
    done, solves = 0, 0
    while done < steps:
        r = ctx.stpcg(g, H, Delta=1e3, max_iterations=min(TPCG, steps - done), kappa_fgr=1e-12,
                      theta=1.0, s_out=s_out)
        solves += 1
        if r["iterations"] == 0:
            raise RuntimeError("STPCG made no progress (exit %d)" % r["exit_reason"])
        done += r["iterations"]
    if last is not None:
        last.update({k: v for k, v in r.items() if k != "s"})
    return solves


_ORACLE_10 = {}   # (grid, p) -> rank 0's oracle solve of the verification problem, computed once per run


def oracle_ten_iterations(nx, ny, nz, p, Xb_glob):
    """CHECKER, rank 0 only, outside every timed region: the plain-C restatement of the reference (oracle/liboracle.so,
    == the reference's templates bit for bit, tests/test_cpu_oracle_templates.py) on the GLOBAL problem of the
    verification solve: 10 STPCG iterations from the same iterate.  None when the checker library is not there."""
    key = (nx, ny, nz, p)
    if key not in _ORACLE_10:
        try:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle_py
            O = oracle_py.Oracle()
            rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
            oprob = O.stiefel_rq(nx * ny * nz, p, rowptr, col, val)
            x = np.ascontiguousarray(Xb_glob).ravel()
            go = O.eval_grad(oprob, x)
            t0 = time.perf_counter()
            o = O.stpcg_problem(oprob, x, go, 1e3, max_iterations=10, kappa_fgr=1e-12, theta=1.0, trace_cap=16)
            o["seconds"] = time.perf_counter() - t0
            o["g"] = go
            O.free(oprob)
            _ORACLE_10[key] = o
        except Exception as e:  # noqa: BLE001  (no checker library on this box: say so, do not fail the run)
            print(f"bench.py: oracle check of the sharded solve unavailable: {e}", file=sys.stderr)
            _ORACLE_10[key] = None
    return _ORACLE_10[key]


def verify_distributed(ctx, dist, A, prob, nx, ny, nz, z0, z1, p, world, rank, peer_memory=False, info=None):
    """Cheap end-to-end checks of the N-rank data path before anything is timed (the 8-GPU node is the first
    place the cross-device exchanges ever run).  Returns a list of failure strings (empty = all good), the same
    on every rank."""
    fails = []
    n0, n1 = nx * ny * z0, nx * ny * z1
    E, lam = [], []
    for m in [(1, 1, 1), (1, 1, 2), (1, 2, 1)][:p]:
        v, lm = wl.laplacian_3d_eigvec(nx, ny, nz, *m)
        E.append(v[n0:n1])
        lam.append(lm + 0.1)
    E = np.ascontiguousarray(np.stack(E, axis=1))
    Ed = ctx.upload(E)
    # halo exchange: exact eigenvectors of the global grid operator must stay eigenvectors on every slab
    Y = A.spmm(p, Ed).numpy().reshape(n1 - n0, p)
    err = float(np.abs(Y - E * np.array(lam)[None, :]).max() / np.abs(E).max())
    if not err < 1e-11:
        fails.append(f"rank {rank}: sharded SpMM of exact eigenvectors off by {err:.3e} (halo exchange)")
    # scalar all-reduce: the p unit vectors have squared norm p in total
    d = Ed.dot(Ed)
    if not abs(d - p) < 1e-11 * p:
        fails.append(f"rank {rank}: all-reduced <E,E> = {d!r}, expected {p}")
    # replicated control flow: a short fused solve must report the same bits everywhere
    Xb_glob = wl.stiefel_bench_iterate(nx, ny, nz, p, eps=1e-3, seed=7)[0]
    X = ctx.upload(np.ascontiguousarray(Xb_glob[n0:n1]))
    g, H = prob.model(X)
    # ... and it must be the REFERENCE's solve, not merely the same wrong one everywhere (r04 verdict): rank 0 runs the
    # CPU oracle on the global problem; every rank compares its rows of the step, the counts and the alpha / beta traces.
    # r06: on IDENTICAL INPUTS -- every rank's solve takes its rows of the oracle's gradient, bit for bit, as its input (the
    # sharded gradient itself is compared with the oracle's separately), so the bar is the plain 1e-10 with four orders
    # to spare whatever the rank count regroups.
    o = oracle_ten_iterations(nx, ny, nz, p, Xb_glob) if rank == 0 else None
    box = [None if o is None else {k: o[k] for k in ("iterations", "exit_reason", "M_norm", "seconds")} |
           {"alpha": o["trace"]["alpha"], "beta": o["trace"]["beta"]}]
    dist.broadcast_object_list(box, src=0)
    slabs = [None]
    dist.scatter_object_list(slabs, [(np.ascontiguousarray(o["s"].reshape(-1, p)[nx * ny * a: nx * ny * b]),
                                      np.ascontiguousarray(o["g"].reshape(-1, p)[nx * ny * a: nx * ny * b]))
                                     for a, b in wl.shard_rows(nz, world)] if o is not None else [None] * world, src=0)
    g_in, g_rel = g, None
    if box[0] is not None:
        import torch
        g_ref = slabs[0][1]
        acc = torch.tensor([float(((g.numpy().reshape(-1, p) - g_ref) ** 2).sum()), float((g_ref ** 2).sum())],
                           dtype=torch.float64)
        dist.all_reduce(acc)
        g_rel = float(np.sqrt(acc[0] / acc[1]))
        if not g_rel < 1e-10:
            fails.append(f"rank {rank}: sharded gradient vs the CPU oracle's: {g_rel:.2e}")
        g_in = ctx.upload(np.ascontiguousarray(g_ref).ravel())
    r = ctx.stpcg(g_in, H, Delta=1e3, max_iterations=10, kappa_fgr=1e-12, theta=1.0, trace_cap=16)
    mine = (r["iterations"], r["exit_reason"], float(r["M_norm"]).hex(), r["hvp_calls"], ctx.comm_ipc_error())
    if box[0] is not None:
        ob, s_mine = box[0], r["s"].numpy().reshape(-1, p)
        import torch
        acc = torch.tensor([float(((s_mine - slabs[0][0]) ** 2).sum()), float((slabs[0][0] ** 2).sum())], dtype=torch.float64)
        dist.all_reduce(acc)
        es = float(np.sqrt(acc[0] / acc[1]))
        ea = float(np.max(np.abs(r["trace"]["alpha"] / ob["alpha"] - 1)))
        eb = float(np.max(np.abs(r["trace"]["beta"] / ob["beta"] - 1)))
        if (r["iterations"], r["exit_reason"]) != (ob["iterations"], ob["exit_reason"]):
            fails.append(f"rank {rank}: sharded solve ({r['iterations']}, exit {r['exit_reason']}) != oracle "
                         f"({ob['iterations']}, exit {ob['exit_reason']})")
        if not (es <= 1e-10 and ea <= 1e-10 and eb <= 1e-10 and abs(r["M_norm"] / ob["M_norm"] - 1) <= 1e-10):
            fails.append(f"rank {rank}: sharded 10-iteration solve vs the CPU oracle: s {es:.2e}, alpha {ea:.2e}, "
                         f"beta {eb:.2e} (bar 1e-10)")
        if info is not None:
            info.update(oracle_check={"s_rel": es, "alpha_rel": ea, "beta_rel": eb, "iterations": ob["iterations"],
                                      "oracle_seconds": ob["seconds"], "bar": 1e-10, "identical_inputs": True,
                                      "g_rel_sharded_vs_oracle": g_rel})
    elif info is not None:
        info.update(oracle_check=None)
    if peer_memory and os.environ.get("MI355OPT_NO_FOLD") != "1":
        # the exchanges folded into the CG / Hessian kernels (scalars in the prologues, halo rows in the direction
        # kernel's stores) against the separate exchange kernels: bit-identical by construction, so any difference
        # is a cross-device ordering problem of the folded form
        s_fold = r["s"].numpy()
        ctx.comm_ipc_fold(False)
        r2 = ctx.stpcg(g_in, H, Delta=1e3, max_iterations=10, kappa_fgr=1e-12, theta=1.0)
        ctx.comm_ipc_fold(True)
        sep = (r2["iterations"], r2["exit_reason"], float(r2["M_norm"]).hex(), r2["hvp_calls"], ctx.comm_ipc_error())
        if sep != mine or not np.array_equal(s_fold, r2["s"].numpy()):
            fails.append(f"rank {rank}: folded exchanges {mine} != separate exchange kernels {sep} (or the steps differ)")
    every = [None] * world
    dist.all_gather_object(every, (mine, fails))
    if len({e[0] for e in every}) != 1:
        fails.append(f"ranks disagree on a 10-iteration solve: {[e[0] for e in every]}")
    if any(e[0][4] for e in every):
        fails.append("a bounded wait of the peer-memory layer timed out")
    for e in every:
        fails.extend(x for x in e[1] if x not in fails)
    fails = sorted(set(fails))
    agree = [None] * world
    dist.all_gather_object(agree, fails)
    return sorted(set(x for a in agree for x in a))


def host_cpus():
    """CPUs this process may really use: affinity mask, capped by the cgroup CPU quota (the GPU boxes show 256
    cores to a container that has a quota of 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "?"


def cpu_baseline(nx, ny, nz, p, rowptr, col, val, Xb, bytes_per_step, bytes_8d, keep=None, timed_iterations=None):
    """(a) The reference's own CPU path (oracle/_ref/libref.so = the reference templates compiled from
    /root/reference, single-threaded like the reference; the plain-C oracle, "port", if that build is absent) and
    (b) the oracle's OpenMP build (its vector and row loops as `omp parallel for`) on all the host CPUs this
    process may use -- both on a bounded sample of the same workload, same iteration counts asserted.
    `value` prices a CPU step with the same bytes as the GPU's `value` (so the ratio of the two is the ratio of
    steps per second).
    keep: a dict that receives what the `parity` block of the line is made of -- the step, traces and counts of the
    reference's last timed solve ("ref"), one traced solve of the plain-C port ("port": the reference's observer sees
    alpha only, the port also records beta, kappa, <r,v>, and its step must equal the reference's bit for bit), the
    re-associated reference ("omp": the same statements with per-thread partial sums = the conditioning floor of the
    comparison) and, when the timed GPU solve had another iteration count (`timed_iterations`), the same three at
    that count."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    n = nx * ny * nz
    out = []
    ncpu = host_cpus()
    for omp in (False, True):
        O = oracle_py.Oracle(omp=omp)
        if omp:
            threads = O.set_threads(ncpu)
            lib, kind = O, "port"
        else:
            threads = 1
            lib, kind = (oracle_py.Reference(), "reference") if oracle_py.have_reference() else (O, "port")
        prob = O.stiefel_rq(n, p, rowptr, col, val)
        g = O.eval_grad(prob, Xb.ravel())
        solves, iters = (5 if not omp else 20), 0  # ~11 s on one core, ~4.5 s on 16
        O.stpcg_problem(prob, Xb.ravel(), g, 1e3, max_iterations=2, kappa_fgr=1e-12, theta=1.0, lib=lib)
        t0 = time.perf_counter()
        for _ in range(solves):
            r = O.stpcg_problem(prob, Xb.ravel(), g, 1e3, max_iterations=TPCG, kappa_fgr=1e-12, theta=1.0,
                                lib=lib, trace_cap=TPCG + 2)
            if r["iterations"] != TPCG:
                raise RuntimeError("CPU baseline: %d iterations instead of %d" % (r["iterations"], TPCG))
            iters += r["iterations"]
        dt = time.perf_counter() - t0
        if keep is not None:
            # (outside the timed sample) what the parity block needs
            def solve(lib_, k):
                return O.stpcg_problem(prob, Xb.ravel(), g, 1e3, max_iterations=k, kappa_fgr=1e-12, theta=1.0,
                                       lib=lib_, trace_cap=k + 2)
            if omp:
                keep["omp"] = {TPCG: r}
                if timed_iterations and timed_iterations != TPCG:
                    keep["omp"][timed_iterations] = solve(lib, timed_iterations)
            else:
                keep["ref"], keep["ref_kind"], keep["g"] = {TPCG: r}, kind, g
                keep["port"] = {TPCG: solve(O, TPCG)} if kind == "reference" else {TPCG: r}
                if timed_iterations and timed_iterations != TPCG:
                    keep["ref"][timed_iterations] = solve(lib, timed_iterations)
                    keep["port"][timed_iterations] = solve(O, timed_iterations) if kind == "reference" \
                        else keep["ref"][timed_iterations]
        O.free(prob)
        out.append({"value": iters * bytes_per_step / dt / 1e9, "unit": "GB/s", "cores": threads, "kind": kind,
                    "sample": f"{solves} STPCG solves x {TPCG} inner iterations of the same St({n},{p}) workload "
                              f"({iters} steps, {dt:.1f} s)" + (", OpenMP loops (oracle/liboracle_omp.so)" if omp else ""),
                    "ms_per_step": 1e3 * dt / max(iters, 1),
                    "reference_schedule_bytes_per_second_not_a_bandwidth": iters * bytes_8d / dt,
                    "cpu": cpu_model(), "host_cores_visible": os.cpu_count(), "host_cores_usable": ncpu})
    return out[0], out[1]


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def _trace_rel(a, b):
    k = min(len(a), len(b))
    return float(np.max(np.abs(np.asarray(a[:k]) / np.asarray(b[:k]) - 1))) if k else 0.0


def parity_block(keep, gpu):
    """CHECKER (outside every timed region): the GPU's solves against the reference's, in numbers, in the line the driver
    records.  gpu: [(iterations, dict(s=ndarray, M_norm, iterations, exit_reason, trace, label))] -- the LAST TIMED
    solve's step (copied off the device right behind the timed region, before anything else ran) and further solves of
    the same problem with traces, from the device's own gradient and from the reference's gradient bits.  Reference: the `cpu_baseline` leg's own solves (oracle/_ref/libref.so = the
    reference's templates; "port" = oracle/liboracle.so when that build is absent).  floor_*: how far the SAME reference
    algorithm moves when only the association of its sums changes (oracle/liboracle_omp.so) -- no implementation with
    another reduction order can be held to less."""
    BAR = 1e-10   # BASELINE.json north_star: "iterate match to the CPU reference within 1e-10 relative"
    out = {"bar": BAR, "reference": ("oracle/_ref/libref.so: the reference's own templates compiled from its headers, "
                                     "1 thread") if keep["ref_kind"] == "reference"
           else "oracle/liboracle.so: the plain-C restatement (no reference build on this box)",
           "floor": "oracle/liboracle_omp.so: the same statements, per-thread partial sums (re-associated reference)"}
    port_is_ref = True
    for k, gr in gpu:
        ref, port, omp = keep["ref"][k], keep["port"][k], keep["omp"].get(k)
        same = bool(np.array_equal(ref["s"], port["s"]) and ref["M_norm"] == port["M_norm"] and
                    list(ref["trace"]["alpha"]) == list(port["trace"]["alpha"]))
        port_is_ref = port_is_ref and same
        rec = {"iterations_gpu": int(gr["iterations"]), "iterations_reference": int(ref["iterations"]),
               "iterations_equal": bool(gr["iterations"] == ref["iterations"]),
               "exit_reason_equal": bool(gr["exit_reason"] == port["exit_reason"]),
               "s_rel": _rel(gr["s"], ref["s"]),
               "M_norm_rel": abs(gr["M_norm"] / ref["M_norm"] - 1.0)}
        if gr.get("trace"):
            rec["alpha_rel"] = _trace_rel(gr["trace"]["alpha"], ref["trace"]["alpha"])
            rec["beta_rel"] = _trace_rel(gr["trace"]["beta"], port["trace"]["beta"])
            rec["kappa_rel"] = _trace_rel(gr["trace"]["kappa"], port["trace"]["kappa"])
        if omp is not None:
            rec["floor_s"] = _rel(omp["s"], ref["s"])
            rec["floor_alpha"] = _trace_rel(omp["trace"]["alpha"], ref["trace"]["alpha"])
            rec["floor_beta"] = _trace_rel(omp["trace"]["beta"], port["trace"]["beta"])
        rec["s_within_bar"] = bool(rec["s_rel"] <= BAR)
        if "equals_last_timed_solve_bitwise" in gr:
            rec["equals_last_timed_solve_bitwise"] = gr["equals_last_timed_solve_bitwise"]
        out[gr["label"]] = rec
    out["port_equals_reference_bitwise"] = port_is_ref
    return out


# ---- cfg3 / cfg5 legs: the other two single-GPU configurations of BASELINE.json in the driver's own line --------------
def _bench_client():
    import ctypes as C
    path = os.path.join(ROOT, "tools", "libbench_client.so")
    if not os.path.exists(path):
        raise FileNotFoundError(path + " (run __graft_entry__.build())")
    L = C.CDLL(path)
    L.bc_last_error.restype = C.c_char_p
    return L, C


def cfg3_leg(ctx, N=500_000, steps=500):
    """BASELINE cfg3: SO(3)^N pose graph, N = 5e5, ring + 2 chords per node, TNT with 3x3 block-Jacobi preconditioned
    STPCG.  (a) one fused inner iteration (reference IterativeSolvers.h:285-422) = block HVP + 3x3 block-Jacobi CG, priced
    by SURVEY.md 8(d): 120 Nt for the CG part (Nt = 3 N tangent doubles) + the HVP's matrix (72 B per stored 3x3 block,
    4 B per block index), xi read, h written; (b) a whole TNT outer iteration through the drop-in templates
    (tools/bench_client.cpp: Optimization::Riemannian::TNT on DeviceVector), wall time / outer iterations."""
    t_leg = time.perf_counter()
    # near the optimum (where TNT spends its inner iterations): PSD Hessians, full inner solves
    ei, ej, Rt, w, _Rtrue, Rinit = wl.pose_graph(N, seed=7, init_sigma=0.02)
    prob = ctx.so3n(N, ei, ej, Rt, w)
    R = ctx.upload(Rinit)
    g, H, P = prob.model(R)
    inc = 2 * ei.size                       # incidences = off-diagonal 3x3 blocks of the connection Laplacian
    Nt = 3 * N
    hvp_bytes = 72 * (inc + N) + 4 * inc + 8 * 2 * Nt
    step_bytes = 120 * Nt + hvp_bytes
    s_out = ctx.vec(Nt)
    kw = dict(Delta=1e6, kappa_fgr=1e-14, theta=1.0, s_out=s_out)
    first = ctx.stpcg(g, H, P, max_iterations=50, **kw)
    k_solve = first["iterations"]
    if k_solve == 0:
        raise RuntimeError("cfg3: STPCG made no progress")
    # solves with the result DEFERRED (what TNT's fused outer loop does): the device's pipeline, no host turnaround
    nsolves = max(1, steps // k_solve)
    for _ in range(max(1, nsolves // 4)):
        ctx.stpcg(g, H, P, max_iterations=50, defer=True, **kw)
    ctx.stpcg_collect()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(nsolves):
        ctx.stpcg(g, H, P, max_iterations=50, defer=True, **kw)
    last = ctx.stpcg_collect()
    ctx.sync()
    dt = time.perf_counter() - t0
    if last["iterations"] != k_solve:
        raise RuntimeError("cfg3: solves differ in length")
    names = ("bsr3_spmv_dots", "cg_update", "cg_pupdate")
    for k in names:
        ctx.ktime_enable(k, True)
    ctx.ktime_reset()
    for _ in range(5):   # capped at the length they converge at: no speculative no-op launch is averaged in
        ctx.stpcg(g, H, P, max_iterations=k_solve, **kw)
    per = {}
    for k in names:
        cnt, ms = ctx.ktime_read(k)
        ctx.ktime_enable(k, False)
        per[k] = {"launches": cnt, "avg_us_event_pairs": 1e3 * ms / max(cnt, 1)}
    hv = per["bsr3_spmv_dots"]
    us_step = 1e6 * dt / (nsolves * k_solve)
    out = {"workload": f"cfg3 SO(3)^N, N = {N}, ring + 2 chords per node ({ei.size} edges), chordal cost, "
                       "3x3 block-Jacobi preconditioned STPCG at a near-optimal iterate",
           "inner_step": {"us": us_step, "steps": nsolves * k_solve, "iterations_per_solve": k_solve,
                          "algorithmic_bytes": step_bytes,
                          "bytes_basis": "SURVEY 8(d): 120 Nt (block-Jacobi CG) + 72 (blocks + N) + 4 blocks + 16 Nt (HVP)",
                          "GBps": step_bytes / us_step / 1e3, "frac": step_bytes / (us_step * 1e-6) / 1e9 / HBM_PEAK_GBS,
                          "timing": "wall clock over solves whose results are deferred (set-up kernels and the "
                                    "speculative launches behind a solve's exit included)",
                          "roofline": {"bound": "hbm", "kernel": "bsr3_spmv_dots", "algorithmic_bytes_per_launch": hvp_bytes,
                                       "avg_launch_us_event_pairs": hv["avg_us_event_pairs"],
                                       "achieved": hvp_bytes / hv["avg_us_event_pairs"] / 1e3, "peak": HBM_PEAK_GBS,
                                       "unit": "GB/s", "frac": hvp_bytes / hv["avg_us_event_pairs"] / 1e3 / HBM_PEAK_GBS,
                                       "traffic": None},
                          "kernels": per}}
    del g, H, P, s_out
    # (b) the outer iteration through the drop-in templates
    L, C = _bench_client()

    class Rep(C.Structure):
        _fields_ = [("seconds", C.c_double), ("f", C.c_double), ("gradfx_norm", C.c_double), ("outer", C.c_size_t),
                    ("inner", C.c_size_t), ("syncs", C.c_size_t), ("status", C.c_int)]
    rep = Rep()
    i32, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    L.bc_tnt_so3n.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, i32, i32, dp, dp, dp, C.c_size_t, C.c_size_t,
                              C.c_int, C.POINTER(Rep)]
    R0 = np.ascontiguousarray(Rinit, dtype=np.float64).ravel()
    rc = L.bc_tnt_so3n(ctx.h, N, ei.size, ei.ctypes.data_as(i32), ej.ctypes.data_as(i32),
                       Rt.ctypes.data_as(dp), w.ctypes.data_as(dp), R0.ctypes.data_as(dp), 12, 50, 2, C.byref(rep))
    if rc:
        raise RuntimeError("cfg3 TNT: " + L.bc_last_error().decode())
    ipo = rep.inner / max(rep.outer, 1)
    # an outer iteration's compulsory bytes: the fused trial step (dm product = one HVP; retraction R, h -> R+ and, r06,
    # its quaternions: 168 N + 32 N; model assembly at R+: R own 72 N, per incidence 32 (quaternion measurement) + 8
    # (weight) + 8 (indices) + 32 (the neighbour's quaternion; 72 as a matrix until r05) read and 72 (block) written,
    # D, D^-1 72 N each and grad 24 N written; 7 tangent vectors through the step's dots and D^-1 grad: 168 N) + its
    # inner iterations
    trial_bytes = hvp_bytes + 200 * N + (240 * N + 152 * inc) + 168 * N
    outer_bytes = trial_bytes + ipo * step_bytes
    us_outer = 1e6 * rep.seconds / max(rep.outer, 1)
    out["outer_iteration"] = {"us": us_outer, "outer_iterations": int(rep.outer), "inner_iterations": int(rep.inner),
                              "inner_per_outer": ipo, "host_syncs_per_outer": rep.syncs / max(rep.outer, 1),
                              "status": int(rep.status), "f": rep.f,
                              "algorithmic_bytes": outer_bytes, "trial_step_bytes": trial_bytes,
                              "GBps": outer_bytes / us_outer / 1e3,
                              "frac": outer_bytes / (us_outer * 1e-6) / 1e9 / HBM_PEAK_GBS,
                              "through": "Optimization::Riemannian::TNT<DeviceVector, DeviceVector> (tools/bench_client.cpp), "
                                         "second run in its context, wall time / outer iterations"}
    out["leg_seconds"] = time.perf_counter() - t_leg
    return out


def cfg5_leg(ctx, grid=126, nx=24, nev=20, iters=22):
    """BASELINE cfg5: LOBPCG, k = 20 eigenpairs of the 7-point Laplacian on grid^3 (m = 2 000 376), nx = 24, no
    preconditioner, through the drop-in template (tools/bench_client.cpp: Optimization::LinearAlgebra::LOBPCG on
    DeviceMatrix, random-X0 overload): `iters` iterations with tau so small that nothing converges or locks, i.e. every
    timed iteration runs at the full basis width ns = 3 nx.  Bytes by SURVEY.md 8(d):
    2 A_bytes + 8 m (2 ns [A S] + 2 ns [Gram] + ns + 2 nx [X, P update] + 2 nx [A X] + 3 nx [residual])."""
    t_leg = time.perf_counter()
    m = grid ** 3
    rowptr, col, val = wl.laplacian_3d(grid, grid, grid)
    A = ctx.csr(m, rowptr, col, val)
    nnz = int(rowptr[-1])
    del rowptr, col, val
    L, C = _bench_client()

    class Rep(C.Structure):
        _fields_ = [("spi", C.c_double), ("total", C.c_double), ("theta0", C.c_double), ("rmax", C.c_double),
                    ("iterations", C.c_size_t), ("timed", C.c_size_t), ("nc", C.c_size_t)]
    rep = Rep()
    L.bc_lobpcg.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_double,
                            C.POINTER(Rep)]
    names = ("lobpcg_gram", "lobpcg_update", "lobpcg_residual", "csr_spmm")
    rc = L.bc_lobpcg(ctx.h, A.h, m, nx, nev, 4, 1e-300, C.byref(rep))     # pool and kernels warm
    if rc:
        raise RuntimeError("cfg5 LOBPCG: " + L.bc_last_error().decode())
    rc = L.bc_lobpcg(ctx.h, A.h, m, nx, nev, iters, 1e-300, C.byref(rep))
    if rc:
        raise RuntimeError("cfg5 LOBPCG: " + L.bc_last_error().decode())
    spi, timed = rep.spi, int(rep.timed)
    ns = 3 * nx
    a_bytes = 12 * nnz + 4 * (m + 1)
    it_bytes = 2 * a_bytes + 8 * m * (2 * ns + 2 * ns + ns + 2 * nx + 2 * nx + 3 * nx)
    # the same iterations once more with an event pair around every library call of the four kernel families
    for k in names:
        ctx.ktime_enable(k, True)
    ctx.ktime_reset()
    rc = L.bc_lobpcg(ctx.h, A.h, m, nx, nev, iters, 1e-300, C.byref(rep))
    per = {}
    for k in names:
        cnt, ms = ctx.ktime_read(k)
        ctx.ktime_enable(k, False)
        per[k] = {"calls": cnt, "avg_us_event_pairs": 1e3 * ms / max(cnt, 1)}
    if rc:
        raise RuntimeError("cfg5 LOBPCG: " + L.bc_last_error().decode())
    # Gram pair S'A(S), S'S at ns = 72: 25 tiles of 16 x 16 (upper block triangles of both, the half-empty last tile
    # column shared), 2 m flops per tile entry; the first two calls of a run are on narrower bases, so the average over
    # the run slightly flatters the time per call -- the TF/s figure is therefore computed from full-width calls only if
    # the run is long enough for them to dominate (iters >= 20)
    tiles = 25 if ns == 72 else None
    gram = dict(per["lobpcg_gram"])
    if tiles and gram["calls"]:
        gf = tiles * 256 * 2 * m / 1e9
        gram.update(tiles_16x16=tiles, executed_GF=gf, TFLOPs=gf / gram["avg_us_event_pairs"] * 1e3,
                    fp64_matrix_peak_TFLOPs=78.6, frac_of_fp64_matrix_peak=gf / gram["avg_us_event_pairs"] * 1e3 / 78.6,
                    operand_bytes=8 * m * 2 * ns, GBps=8 * m * 2 * ns / gram["avg_us_event_pairs"] / 1e3)
    out = {"workload": f"cfg5 LOBPCG, 7-pt Laplacian {grid}^3 (m = {m}), nx = {nx}, nev = {nev}, ns = {ns}, no "
                       "preconditioner, B = I",
           "us": 1e6 * spi, "iterations_timed": timed, "iterations": int(rep.iterations), "converged_pairs": int(rep.nc),
           "algorithmic_bytes": it_bytes, "bytes_basis": "SURVEY 8(d) cfg5 formula",
           "GBps": it_bytes / (1e6 * spi) / 1e3 if spi > 0 else None,
           "frac": it_bytes / spi / 1e9 / HBM_PEAK_GBS if spi > 0 else None,
           "ritz_0": rep.theta0, "max_residual_norm": rep.rmax,
           "gram_pair": gram, "kernel_families": per,
           "through": "Optimization::LinearAlgebra::LOBPCG<HostVectorD, DeviceMatrix> (tools/bench_client.cpp), user "
                      "function to user function, the first two iterations (basis not yet at full width) dropped",
           "leg_seconds": time.perf_counter() - t_leg}
    del A
    return out


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU on
    127.0.0.1 (free port), stdout (= the one JSON line of rank 0) passed through."""
    import socket
    import subprocess
    ndev = capi.device_count()
    env = dict(os.environ)
    if ndev < args.gpus:
        if ndev < 1:
            raise SystemExit("bench.py: no GPU visible")
        # fewer GPUs than ranks: a FUNCTIONAL rehearsal of the N-rank flow with all ranks on GPU 0 (peer-memory layer
        # only: RCCL refuses duplicate devices); the JSON line says so
        print(f"bench.py: {ndev} GPU(s) visible for --gpus {args.gpus}: rehearsal with all ranks on GPU 0 "
              "(timings are meaningless)", file=sys.stderr)
        env["MI355OPT_BENCH_ONE_GPU"] = "1"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    r = subprocess.run(cmd, env=env, stdout=_RESULT_FD)
    raise SystemExit(r.returncode)


def peer_probe_main(args):
    """`--peer-probe WHAT` (a throwaway process per rank, started by layer_probe below).  WHAT = "peer": bring the
    peer-memory layer up and run its self-test -- known-value exchanges, then the folded exchange and all three forms
    of the halo push between the real peers.  WHAT = "rccl": ncclCommInitRank over all ranks and two all-reduces of
    known values through the library's own reduction path.  The exit code says whether it came up: 0 yes, 3 no."""
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    one_gpu = os.environ.get("MI355OPT_BENCH_ONE_GPU") == "1"
    if one_gpu:
        os.environ.setdefault("MI355OPT_MAX_GRID", str(max(16, 192 // max(world, 1))))
        os.environ.setdefault("MI355OPT_IPC_TIMEOUT_MS", "5000")
    ctx = capi.Context(0 if one_gpu else int(os.environ.get("LOCAL_RANK", "0")))
    if args.peer_probe == "rccl":
        uid = [ctx.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(world, rank, uid[0])
        ok = ctx.comm_rccl_count() == world
        v = ctx.upload(np.full(4096, float(rank + 1)))
        want = 4096.0 * sum((r + 1.0) ** 2 for r in range(world))
        for _ in range(2):
            ok = ok and v.dot(v) == want   # (small integers: exact in any summation order)
    else:
        ok = ctx.enable_peer_memory(world, rank, dist, force=True)
    dist.barrier()
    ctx.comm_finalize()
    ctx.close()
    os._exit(0 if ok else 3)


def layer_probe(dist, world, rank, what):
    """The first cross-device exchanges of this code base -- peer stores over hipIpc mappings AND RCCL collectives with
    more than one rank -- happen on the node the scaling bench runs on.  A layer that merely fails its self-test is
    dropped by itself; one that takes the process down (a fault on a peer mapping) or never returns from its bring-up
    would take the bench line with it -- so each layer's bring-up and self-test run in a throwaway process per rank FIRST
    (what = "peer" | "rccl"), and the real processes bring a layer up only if every probe of it came back clean.
    Returns (ok, seconds)."""
    import socket
    import subprocess
    t0 = time.perf_counter()
    port = [None]
    if rank == 0:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port[0] = sk.getsockname()[1]
    dist.broadcast_object_list(port, src=0)
    # (under torch.distributed.run the launcher's agent hosts the rendezvous store and the ranks only connect to it
    # -- TORCHELASTIC_USE_AGENT_STORE; the probe's own rendezvous on a fresh port needs its rank 0 to host one)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env.update(MASTER_PORT=str(port[0]), MASTER_ADDR="127.0.0.1")
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--peer-probe", what, "--gpus", str(world)],
                           env=env, capture_output=True, text=True, timeout=240)
        ok, note = r.returncode == 0, (r.stderr or "")[-300:]
    except subprocess.TimeoutExpired:
        ok, note = False, "probe timed out"
    every = [None] * world
    dist.all_gather_object(every, (ok, note))
    if rank == 0 and not all(e[0] for e in every):
        print("bench.py: the %s probe failed on rank(s) %s: %s\n  %s" %
              ({"peer": "peer-memory", "rccl": "RCCL"}[what], [i for i, e in enumerate(every) if not e[0]],
               {"peer": "RCCL carries the exchanges", "rccl": "the peer-memory layer carries the exchanges"}[what],
               " | ".join(e[1].strip().splitlines()[-1] if e[1].strip() else "-" for e in every if not e[0])),
              file=sys.stderr)
    return all(e[0] for e in every), time.perf_counter() - t0


class Watchdog:
    """N > 1 only.  arm(what, seconds) before a phase that may never return (an exchange layer's first collectives on a
    real node), disarm() behind it.  If a phase overruns: rank 0 prints the line that keep() registered -- the headline
    already measured on an earlier, verified layer -- with a note naming the phase, and every rank ends the process with
    exit code 0; with nothing kept the exit code is 3.  A hung collective cannot be cancelled; a clean line can still
    be delivered."""

    def __init__(self, rank):
        import threading
        self.rank, self.deadline, self.what, self.kept = rank, None, "", None
        self.lock = threading.Lock()
        t = threading.Thread(target=self._run, daemon=True)
        t.start()

    def arm(self, what, seconds):
        with self.lock:
            self.what, self.deadline = what, time.monotonic() + seconds

    def disarm(self):
        with self.lock:
            self.deadline = None

    def keep(self, make):
        """make(note) -> the JSON dict to print if a later phase does not return (rank 0 only uses it)"""
        with self.lock:
            self.kept = make

    def _run(self):
        while True:
            time.sleep(0.5)
            with self.lock:
                late = self.deadline is not None and time.monotonic() > self.deadline
                what, kept = self.what, self.kept
            if not late:
                continue
            print(f"bench.py: rank {self.rank}: '{what}' did not return in time" +
                  ("; printing the line measured before it" if kept else ""), file=sys.stderr, flush=True)
            if self.rank == 0 and kept is not None:
                try:
                    out = kept(f"'{what}' did not return within its time limit afterwards")
                    out["watchdog"] = f"'{what}' did not return; this line was measured before it"
                    os.write(_RESULT_FD, (json.dumps(out) + "\n").encode())
                except Exception as e:  # noqa: BLE001
                    print("bench.py: could not print the kept line: %r" % (e,), file=sys.stderr, flush=True)
                    os._exit(3)
            os._exit(0 if (kept is not None or self.rank != 0) else 3)


def dry_run_layers(world, comm_arg, ndev):
    """`--dry-run-layers N` (at N = 1, on any box -- no GPU work): what a `--gpus N` run WOULD do, so that the one shot on
    an 8-GPU node has no first-time code path besides the cross-device transport itself: the launch line, the probe
    order, the exchange layers in the order they are verified and timed, the rule that picks the headline's, the slab
    partition with every rank's halo plan (the host-side planning code of mi_csr_create_sharded really runs here:
    mi_csr_shard_plan per rank, extents checked for symmetry) and the bytes per exchange."""
    nx, ny, nz = wl.cfg2_grid(world)
    slabs = wl.shard_rows(nz, world)
    n_glob, plane = nx * ny * nz, nx * ny
    starts = [plane * a for a, _ in slabs] + [n_glob]
    ranks = []
    for r, (z0, z1) in enumerate(slabs):
        # the plan only looks at the column indices of the slab's boundary planes: build those two planes' rows
        rec = {"rank": r, "z_planes": [z0, z1], "rows": plane * (z1 - z0)}
        need = [0, 0]
        for side, zz in ((0, z0), (1, z1 - 1)):
            _, colg, _ = wl.laplacian_3d(nx, ny, nz, z_range=(zz, zz + 1))
            _, lo, hi = capi.csr_shard_plan(n_glob, world, r, starts, colg)
            need[0] += lo if side == 0 else 0
            need[1] += hi if side == 1 else 0
        rec["halo_rows_needed"] = {"from_rank_below": need[0], "from_rank_above": need[1]}
        ranks.append(rec)
    sym = all(ranks[r]["halo_rows_needed"]["from_rank_above"] == ranks[r + 1]["halo_rows_needed"]["from_rank_below"]
              for r in range(world - 1))
    want_peer = comm_arg not in ("rccl", "rccl2")
    layers = (["peer", "peer-separate", "peer-separate-rprime"] if want_peer else []) + ["rccl", "rccl2"]
    return {
        "world": world, "gpus_visible_here": ndev,
        "launch": f"python -m torch.distributed.run --nnodes=1 --nproc-per-node {world} --master-addr 127.0.0.1 "
                  f"--master-port P bench.py --gpus {world} --steps K --warmup W   (or bare: python bench.py --gpus {world})",
        "control_plane": "torch.distributed / gloo: rendezvous, barriers, max-over-ranks time, object broadcasts of the "
                         "oracle check; never on the data path",
        "bring_up_order": ["RCCL probe in throwaway processes (ncclCommInitRank + two all-reduces of known values)",
                           "ncclCommInitRank in the bench processes (only if every rank's probe came back clean)",
                           "peer-memory probe in throwaway processes (hipIpc export / map / self-test incl. the folded "
                           "exchange and all three forms of the halo push)",
                           "peer-memory layer mapped in the bench processes (only if every probe came back clean)"],
        "layers_in_order": [{"layer": L, "what": LAYER_TEXT[L]} for L in layers],
        "per_layer": "verify (sharded SpMM of exact eigenvectors, all-reduced <E,E>, a replicated 10-iteration solve "
                     "against the CPU oracle at 1e-10, folded == separate bits) -> 100 warm-up + --ab-steps timed steps -> "
                     "event-paired pass; a layer that raises or fails verification costs itself, not the line; a layer "
                     "that never returns is cut by the watchdog (--leg-timeout) and the line measured before it is printed",
        "headline": "measured at once on the FIRST layer that verifies (kept for the watchdog), measured again on the "
                    "fastest verified layer if that is another one (--comm forces one)",
        "workload": f"cfg4-style weak scaling: Stiefel({n_glob},3) on {nx}x{ny}x{nz}, z-slabs",
        "ranks": ranks, "halo_plan_symmetric": bool(sym),
        "exchange_bytes_per_iteration_per_rank": {
            "scalar_exchanges": "2 (recurrence form): 9 doubles (3 curvature dots + 6 Gram entries) and 1 double",
            "halo_rows": f"{plane} rows x 3 doubles = {plane * 24} bytes to each neighbour"},
        "cpu_baseline": "rank 0, after the timed region, on the PER-GPU problem",
    }


def comm_set_layer(ctx, layer, peer_up=True):
    """switch the exchange layer of a context (the same call on every rank); peer_up: the peer-memory layer is mapped"""
    ctx.set_option("HALO_RPRIME", 1 if layer in ("rccl2", "peer-separate-rprime") else 0)
    if layer in ("rccl", "rccl2"):
        if peer_up:
            ctx.comm_ipc_enable(False)
    else:
        ctx.comm_ipc_enable(True)
        ctx.comm_ipc_fold(layer == "peer")


LAYER_TEXT = {"peer": "peer-memory layer (hipIpc-mapped arenas over xGMI), scalar exchanges folded into the CG kernels' "
                      "prologues and the halo push into the direction kernel: 3 launches per iteration",
              "peer-separate": "peer-memory layer with separate one-workgroup exchange kernels and a halo-push kernel: "
                               "6 launches per iteration",
              "rccl": "RCCL: in-stream ncclAllReduce of the partial rows (2 per iteration, each one group = one launch) "
                      "and an ncclSend/ncclRecv group for the halo rows",
              "peer-separate-rprime": "peer-memory layer, separate exchange kernels, r'-halo form (the rows of the new "
                                      "residual are pushed, halo(p') is formed locally): the transport-independent part "
                                      "of `rccl2`, which a one-GPU rehearsal can exercise",
              "rccl2": "RCCL, r'-halo form: the boundary rows of the new RESIDUAL ride in the RCCL group of the <r,v> "
                       "all-reduce and every rank forms halo(p') = -halo(r') + beta halo(p) itself -- 2 dependent "
                       "collectives per iteration instead of 3 (same bits)"}


def comm_ab_leg(ctx, dist, prob, X, s_out, layer, steps, world, rank, verify, peer_up):
    """One exchange layer: verify (replicated 10-iteration solve, bounded waits clean), then `steps` timed steps (max over
    ranks) and a second, event-paired pass for the per-launch cost of the exchanges that are launches of their own."""
    import torch
    comm_set_layer(ctx, layer, peer_up)
    dist.barrier()
    if os.environ.get("MI355OPT_BENCH_INJECT_HANG") == layer:   # (test of the watchdog: a layer that never returns)
        time.sleep(1e9)
    vinfo = {}
    fails = verify(vinfo)
    leg = {"layer": layer, "what": LAYER_TEXT[layer], "verified": not fails, **vinfo}
    if fails:
        leg["failures"] = fails[:4]
        return leg
    g, H = prob.model(X)
    run_steps(ctx, g, H, s_out, 100)
    ctx.sync()
    dist.barrier()
    k0 = ctx.comm_kernel_launches()
    t0 = time.perf_counter()
    run_steps(ctx, g, H, s_out, steps)
    ctx.sync()
    dt = time.perf_counter() - t0
    k1 = ctx.comm_kernel_launches()
    t = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    leg["steps"] = steps
    leg["us_per_step"] = 1e6 * float(t[0]) / steps
    leg["own_launches_per_step"] = {"scalar_exchange_kernels": (k1[0] - k0[0]) / steps,
                                    "halo_push_kernels": (k1[1] - k0[1]) / steps,
                                    "halo_pushes_folded": (k1[2] - k0[2]) / steps,
                                    "of_those_early_form": (k1[3] - k0[3]) / steps}
    # per-collective microseconds (raw HIP event pairs on the launch stream, ~1.7 us of their own each): every rank
    # runs these steps (they contain the exchanges), rank 0's figures are reported
    names = ["comm_allreduce", "comm_halo"] + HOT
    for k in names:
        ctx.ktime_enable(k, True)
    ctx.ktime_reset()
    run_steps(ctx, g, H, s_out, 100)
    per = {}
    for k in names:
        cnt, ms = ctx.ktime_read(k)
        ctx.ktime_enable(k, False)
        if cnt:
            per[k] = {"launches_per_step": cnt / 100, "avg_us_event_pairs": 1e3 * ms / cnt}
    leg["per_launch"] = per
    leg["ipc_error"] = ctx.comm_ipc_error()
    dist.barrier()
    return leg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--wakeup-steps", type=int, default=int(os.environ.get("MI355OPT_BENCH_WAKEUP_STEPS", "1000")),
                    help="untimed steps of the same solve AHEAD of --warmup that bring an idle device to its steady "
                         "state (reported as device_wakeup_steps; 0 switches it off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the plain-matrix, beyond-cache, cfg3 and cfg5 legs")
    ap.add_argument("--leg-budget", type=float, default=float(os.environ.get("MI355OPT_BENCH_LEG_BUDGET", "45")),
                    help="seconds the cfg3 + cfg5 legs may use together (a leg that would start later is skipped)")
    ap.add_argument("--comm", default=os.environ.get("MI355OPT_COMM", "auto"),
                    choices=["auto", "peer", "peer-separate", "peer-separate-rprime", "rccl", "rccl2"],
                    help="exchange layer of the headline at N > 1 (auto: the fastest layer that verifies)")
    ap.add_argument("--ab-steps", type=int, default=300, help="timed steps of each exchange-layer A/B leg at N > 1")
    ap.add_argument("--leg-timeout", type=float, default=float(os.environ.get("MI355OPT_BENCH_LEG_TIMEOUT", "240")),
                    help="N > 1: seconds an exchange layer's leg (or a headline measurement) may take before the "
                         "watchdog prints the line already measured on an earlier layer and ends the run")
    ap.add_argument("--dry-run-layers", type=int, default=0, metavar="N",
                    help="N = 1 runs only: add a `dry_run_layers` block saying what a --gpus N run would try, in which "
                         "order, with every rank's halo plan computed by the real planning code (no GPU work)")
    ap.add_argument("--peer-probe", choices=["peer", "rccl"], default=None, help=argparse.SUPPRESS)
    ap.add_argument("--no-peer-probe", action="store_true",
                    help="N > 1: map peer memory in the bench processes without trying it in throwaway ones first")
    args = ap.parse_args()
    if args.peer_probe:
        peer_probe_main(args)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args, sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    # MI355OPT_BENCH_FORCE_COMM=1: run the multi-GPU code path (rendezvous, communicator, sharded matrix,
    # slot-path kernels) even with one rank -- the dress rehearsal tests/test_gpu_comm.py runs on a 1-GPU box
    use_comm = world > 1 or os.environ.get("MI355OPT_BENCH_FORCE_COMM") == "1"
    dist = None
    if use_comm:
        import torch.distributed as dist  # gloo: control plane only
        dist.init_process_group("gloo", rank=rank, world_size=world)

    # MI355OPT_BENCH_ONE_GPU=1: functional rehearsal of the N-rank flow with all ranks on GPU 0 and WITHOUT RCCL
    # (which refuses duplicate devices): the peer-memory layer carries every exchange.  Timings are meaningless.
    one_gpu = os.environ.get("MI355OPT_BENCH_ONE_GPU") == "1"
    if one_gpu:
        # the consumers wait for their peers inside their prologue, and the push they wait for comes from workgroup 0
        # of every other rank's kernel: (world - 1) * grid + 1 <= 256 resident 1024-thread workgroups on the one GPU
        os.environ.setdefault("MI355OPT_MAX_GRID", str(max(16, 192 // max(world, 1))))
        os.environ.setdefault("MI355OPT_IPC_TIMEOUT_MS", "5000")
    ctx = capi.Context(0 if one_gpu else local_rank)
    peer_memory = False
    rccl_nranks = 0
    probe = rccl_probe = None
    rccl_up = False
    if use_comm:
        # Each exchange layer is tried in throwaway processes first (layer_probe), and is brought up here only if every
        # rank's probe of it came back clean: a bring-up that faults or never returns then costs that layer, not the line.
        # (MI355OPT_BENCH_TRY_RCCL=1: the rehearsal probes RCCL as well -- it refuses the duplicate device, which is the
        # "RCCL does not come up" branch of a real node; tests/test_gpu_comm.py)
        want_rccl = not one_gpu or (os.environ.get("MI355OPT_BENCH_TRY_RCCL") == "1" and not args.no_peer_probe)
        want_peer = args.comm not in ("rccl", "rccl2") or one_gpu
        if want_rccl and world > 1 and not args.no_peer_probe:
            rccl_probe = layer_probe(dist, world, rank, "rccl")
            want_rccl = rccl_probe[0]
            want_peer = want_peer or not want_rccl   # (--comm rccl without a working RCCL: the other layer, and say so)
        if want_rccl:
            uid = [ctx.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            ctx.comm_init(world, rank, uid[0])
            rccl_nranks = ctx.comm_rccl_count()
            rccl_up = True
        # The peer-memory layer (scalar all-reduces and halo rows by xGMI peer stores) is brought up whenever every
        # rank's export / map / self-test succeeds (the self-test includes the folded exchange and all three forms of
        # the halo push between the real peers); which layer the headline uses is decided below.  The one-GPU rehearsal
        # needs it (RCCL refuses duplicate devices).
        if want_peer and world > 1 and not args.no_peer_probe:
            probe = layer_probe(dist, world, rank, "peer")
        if want_peer and (probe is None or probe[0]):
            peer_memory = ctx.enable_peer_memory(world, rank, dist, force=True)
        if not rccl_up and not peer_memory:
            raise SystemExit("bench.py: neither exchange layer came up" if not one_gpu else
                             "one-GPU rehearsal needs the peer-memory layer")
        dist.barrier()

    # ---- workload ------------------------------------------------------------------------------
    p = 3
    nx, ny, nz = wl.cfg2_grid(world)
    z0, z1 = wl.shard_rows(nz, world)[rank]
    n_glob = nx * ny * nz
    n = nx * ny * (z1 - z0)
    Xb_glob, modes = wl.stiefel_bench_iterate(nx, ny, nz, p, eps=1e-3, seed=7)
    Xb = np.ascontiguousarray(Xb_glob[nx * ny * z0: nx * ny * z1])
    if not use_comm:
        rowptr, col, val = wl.laplacian_3d(nx, ny, nz)
        A = ctx.csr(n, rowptr, col, val)
    else:
        rowptr, col, val = wl.laplacian_3d(nx, ny, nz, z_range=(z0, z1))
        starts = [nx * ny * a for a, _ in wl.shard_rows(nz, world)] + [n_glob]
        dist.barrier()  # the first device-side exchange (halo extents) follows: no rank may be seconds behind
        A = ctx.csr_sharded(n_glob, nx * ny * z0, nx * ny * z1, rowptr, col, val, starts)
    nnz = int(rowptr[-1])
    prob = ctx.stiefel_rq(A, n, p)
    X = ctx.upload(Xb)
    s_out = ctx.vec(n * p)
    N = n * p
    bytes_per_step = wl.cg_bytes_per_iter(N) + wl.stiefel_hvp_bytes(n, nnz, p)  # per GPU
    packed = os.environ.get("MI355OPT_NO_PACKED", "0") != "1"
    peer_up = peer_memory

    def make_line(m, layer, choice, legs, plain_leg=None, big_leg=None, cpu=None, cpu_all=None, parity=None,
                  leg3=None, leg5=None):
        """the JSON line for one measurement `m` of measure()"""
        dt, value, moved_bytes = m["dt"], m["value"], m["moved_bytes"]
        return {
            "metric": "TNT Steihaug-CG HVP+inner-product throughput", "value": value, "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "device_wakeup_steps": m["wakeup_steps"],
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"cfg2 Stiefel({n_glob},{p}) Rayleigh quotient, 7-pt Laplacian "
                                   f"{nx}x{ny}x{nz}+0.1I, fused device STPCG in solves of {TPCG} inner "
                                   f"iterations at a near-optimal iterate (modes {modes})",
                       "rows_per_gpu": n, "nnz_per_gpu": nnz, "N_per_gpu": N,
                       "moved_bytes_per_step_per_gpu": moved_bytes,
                       "reference_schedule_bytes_per_step_per_gpu": bytes_per_step, "solves": m["solves"],
                       "packed_matrix": packed,
                       "parallelism": (f"row-sharded z-slabs x{world}, comm: " + LAYER_TEXT[layer]) if use_comm
                       else "single GPU",
                       "device": ctx.device_name()},
            # `value` = compulsory HBM bytes of the kernels that ran / time: comparable with the 8 TB/s peak.
            "hbm_roofline_frac_whole_step": value / world / HBM_PEAK_GBS,
            "value_basis": "compulsory HBM bytes of the three kernels of an iteration (one-pass Hessian: "
                           "4 nnz + 4 (n+1) + 32 N with the value-indexed matrix, 12 nnz + ... without; "
                           "cg_update 24 N; cg_pupdate 40 N) per second; working set "
                           f"{(6 * 8 * N + (4 if packed else 12) * nnz) / 1e6:.0f} MB per GPU (Infinity Cache: 268 MB)",
            "packed_matrix": packed,
            # NOT a bandwidth: the same time priced by the bytes SURVEY.md 8(d) attributes to the reference's schedule
            # (88 N + 12 nnz + 4 (n+1) + 16 n p + 56 N per iteration); the fused kernels never move those bytes
            "reference_schedule_bytes_per_second_not_a_bandwidth": world * args.steps * bytes_per_step / dt,
            "roofline": m["roofline"],
            # N > 1: which exchange layer the headline ran on and why; every layer's own figure next to it
            "comm_layer": layer, "comm_layer_choice": choice, "rccl_nranks": rccl_nranks if use_comm else None,
            "comm_ab_legs": legs, "rehearsal_one_gpu": bool(one_gpu) if use_comm else None,
            "peer_memory_probe": ({"passed": probe[0], "seconds": probe[1]} if use_comm and probe is not None else None),
            "rccl_probe": ({"passed": rccl_probe[0], "seconds": rccl_probe[1]}
                           if use_comm and rccl_probe is not None else None),
            # the second first-class number: the SAME workload through the generic path any CSR matrix takes
            # (12-byte entries, streaming one-pass kernel; no value table, no window form, no computed far columns)
            "generic_csr_leg": plain_leg, "beyond_cache_leg": big_leg,
            "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_all,
            # N = 1: the timed GPU solve against the reference leg's own solve of the same problem, in numbers
            "parity": parity,
            # N = 1: the other two single-GPU configurations of BASELINE.json, time-boxed (--leg-budget)
            "cfg3_leg": leg3, "cfg5_leg": leg5,
        }

    def measure(layer):
        """The headline on one exchange layer (None: no communicator): device wake-up, W warmup steps, EXACTLY K timed
        steps between barrier + synchronise pairs (max over ranks), then the event-paired roofline pass."""
        if use_comm:
            comm_set_layer(ctx, layer, peer_up)
            dist.barrier()
        g, H = prob.model(X)

        def barrier():
            ctx.sync()
            if dist is not None:
                dist.barrier()

        # ---- device wake-up, warmup, timed region ---------------------------------------------------
        # An idle MI355X needs tens of milliseconds of work before its clocks and its memory system are at their steady
        # state; W warmup steps of 57 us are over long before that (measured, 20 timed steps: W = 5 -> 58.8 us/step,
        # W = 500 -> 57.7, W = 2000 -> 57.4; 500 timed steps: 56.5).  So the device is woken up with a FIXED number of
        # steps of the same solve (the same on every rank: they contain the exchanges) before the W warmup steps the
        # command line asks for; the count is reported in the JSON line (`device_wakeup_steps`; --wakeup-steps 0 switches it
        # off).  The timed region is unchanged: exactly K steps between two barrier + synchronise pairs.
        wakeup_steps = args.wakeup_steps
        if wakeup_steps > 0:
            run_steps(ctx, g, H, s_out, wakeup_steps)
        if args.warmup > 0:
            run_steps(ctx, g, H, s_out, args.warmup)
        barrier()
        last = {}
        t0 = time.perf_counter()
        solves = run_steps(ctx, g, H, s_out, args.steps, last)
        ctx.sync()
        dt = time.perf_counter() - t0
        barrier()
        # the step the LAST TIMED solve left behind, for the parity block (a device-to-host copy after the clock stopped)
        timed_solve = None
        if world == 1 and not use_comm and not args.no_cpu_baseline:
            timed_solve = dict(last, s=s_out.numpy().copy(), trace=None,
                               label="timed_solve_last_of_the_timed_region")
        if dist is not None:
            import torch
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0])
        packed = os.environ.get("MI355OPT_NO_PACKED", "0") != "1"
        kb = kernel_bytes(n, nnz, p, packed=packed)
        # compulsory bytes of the three kernels a completed iteration launches (the per-solve initialisation kernels,
        # ~2 % of the time at 50 iterations per solve, are inside the timed region but not counted)
        recur = os.environ.get("MI355OPT_NO_DIRGRAM", "0") != "1"
        moved_model = (kb["stiefel_hess_fused"] if recur else kb["stiefel_spmm_gram"] + kb["stiefel_finish_dots"]) + \
            kb["cg_update"] + kb["cg_pupdate"]

        # ---- roofline leg: same steps again with HIP-event pairs around every hot kernel ------------
        roofline = None
        moved_bytes = moved_model
        # EVERY rank runs these steps (they contain the same exchanges as the timed ones: a rank running them alone
        # would wait for peers that never come); rank 0's own kernel timings are the ones reported.
        if not args.no_roofline:
            # (200 steps whatever K is: with K = 20 the averages were over 20 launches, the first of them behind a solve's
            # set-up kernels)
            per, moved_measured = timed_kernels(ctx, g, H, s_out, 200, kb)
            moved_bytes = moved_measured
            ran = [k for k in HOT if per[k]["launches"]]
            dom = max(ran, key=lambda k: per[k]["avg_us"] * per[k]["launches"])
            # An event pair also times its own two records: the instrumented steps are slower than the timed region by
            # exactly that, the same amount per launch.  Measured live: (time of a step's launches by event pairs - time
            # of an un-instrumented step) / launches per step, subtracted from every pair average (0 when the step
            # contains launches that are not timed here, e.g. exchange kernels of a multi-rank run).  The net figures add
            # up to the un-instrumented step and agree with rocprofv3's kernel durations (profiles/r02_summary.md) to 1 %.
            tsteps = 200
            pair_us_per_step = sum(per[k]["avg_us"] * per[k]["launches"] for k in ran) / tsteps
            launches_per_step = sum(per[k]["launches"] for k in ran) / tsteps
            ev_overhead = max(0.0, (pair_us_per_step - dt / args.steps * 1e6) / max(launches_per_step, 1e-9))
            if ev_overhead > 0.25 * min(per[k]["avg_us"] for k in ran):  # implausible: keep the raw pairs
                ev_overhead = 0.0
            net_us = per[dom]["avg_us"] - ev_overhead
            achieved = kb[dom] / (net_us * 1e-6) / 1e9
            # `traffic` is NOT measured in this run: PMC counters need separate rocprofv3 passes (tools/pmc_bytes.sh);
            # the figure is the last committed measurement of the same kernel on the same workload, labelled as such
            traffic, traffic_source = None, None
            tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tf):
                try:
                    rec = json.load(open(tf)).get(dom)
                    # a committed measurement only stands for THIS build's kernel: same kernel name, same algorithmic
                    # bytes per launch (a changed format or workload changes them); anything else is refused, not quoted
                    if rec is None:
                        traffic_source = f"profiles/pmc_traffic.json has no record of '{dom}': refused (stale file?)"
                    elif rec.get("algorithmic_bytes") != kb[dom]:
                        traffic_source = (f"profiles/pmc_traffic.json was measured on a '{dom}' of "
                                          f"{rec.get('algorithmic_bytes')} algorithmic bytes per launch, this build's moves "
                                          f"{kb[dom]}: refused as stale (re-run tools/pmc_bytes.sh)")
                    else:
                        traffic = rec.get("hbm_bytes_per_launch")
                        traffic_source = "profiles/pmc_traffic.json (rocprofv3 --pmc passes of tools/pmc_bytes.sh on this " \
                                         "workload and kernel, algorithmic bytes matching; not collected in this run)"
                except Exception:  # noqa
                    traffic = None
            sum_kernel_us = sum(per[k]["avg_us"] * per[k]["launches"] for k in ran) / tsteps
            roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                        "traffic_source": traffic_source,
                        "algorithmic_bytes_per_launch": kb[dom],
                        "avg_launch_us": net_us,
                        # raw pair-timed kernel time over the un-instrumented step (> 1 by the pairs' own cost).  The NET
                        # ratio is 1 by construction (that is how the pairs' cost is measured) and is not reported;
                        # the independent evidence for "no gaps" is rocprofv3's kernel total against the step
                        # (profiles/r05_summary.md)
                        "sum_kernel_us_over_step_us_event_pairs": sum_kernel_us / (dt / args.steps * 1e6),
                        "avg_launch_us_event_pairs": per[dom]["avg_us"],
                        "event_record_overhead_us_per_launch": ev_overhead,
                        "timing": "HIP event pair around every launch on the launch stream, minus the pairs' own cost "
                                  "measured live as (pair-timed step - un-instrumented step) / launches per step",
                        "frac_event_pairs_uncorrected": kb[dom] / (per[dom]["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                        "kernels": per}
            barrier()
            if rank != 0:
                roofline = None
        value = world * args.steps * moved_bytes / dt / 1e9
        return dict(dt=dt, solves=solves, roofline=roofline, moved_bytes=moved_bytes, value=value,
                    wakeup_steps=wakeup_steps, timed_solve=timed_solve, model=(g, H))

    # ---- N > 1: every exchange layer that is up is verified and timed; the headline takes the fastest --------------
    # The first cross-device run of this code happens inside the driver's scaling bench, where a layer that HANGS (an
    # RCCL collective that never returns is not bounded by anything of ours) would cost the whole line.  So the
    # headline is measured on the FIRST layer that verifies, right away, and kept; the remaining layers are then tried
    # under a watchdog, and if one of them does not come back the kept line is printed (with a note) and the run ends
    # cleanly.  If a later layer is faster the headline is measured again on it.
    comm_layer, comm_legs, comm_choice, m = None, None, None, None
    wd = Watchdog(rank) if use_comm else None
    if use_comm:
        layers = (["peer", "peer-separate", "peer-separate-rprime"] if peer_memory else []) + (["rccl", "rccl2"] if rccl_up else [])
        inject = os.environ.get("MI355OPT_BENCH_INJECT_VERIFY_FAILURE") == "1"  # (test of the fallback path)

        def verifier(layer):
            def verify(info=None):
                f = verify_distributed(ctx, dist, A, prob, nx, ny, nz, z0, z1, p, world, rank,
                                       peer_memory=(layer == "peer"), info=info)
                if inject and layer not in ("rccl", "rccl2"):
                    f = f + ["injected failure (test of the fallback path)"]
                return f
            return verify
        comm_legs, kept = [], None
        for L in layers:
            wd.arm(f"exchange layer '{L}'", args.leg_timeout)
            # a layer that RAISES (a library error, the same on every rank: e.g. a buffer it cannot place) costs that
            # layer, not the line: every rank reports, and the leg counts as not verified if any of them failed
            try:
                leg, exc = comm_ab_leg(ctx, dist, prob, X, s_out, L, args.ab_steps, world, rank, verifier(L), peer_up), None
            except capi.MiError as e:
                leg, exc = None, str(e)
            every = [None] * world
            dist.all_gather_object(every, exc)
            if any(every):
                ctx.sync()
                leg = {"layer": L, "what": LAYER_TEXT[L], "verified": False,
                       "failures": [f"rank {i}: {e_}" for i, e_ in enumerate(every) if e_][:4]}
            comm_legs.append(leg)
            if not leg["verified"] and rank == 0:
                print(f"bench.py: exchange layer '{leg['layer']}' failed verification" +
                      (", falling back to RCCL" if leg["layer"] not in ("rccl", "rccl2") else "") + ":\n  " +
                      "\n  ".join(leg.get("failures", [])), file=sys.stderr)
            if leg["verified"] and not leg.get("ipc_error") and kept is None:
                wd.arm(f"headline on '{L}'", args.leg_timeout)
                kept = (L, measure(L))
                wd.keep(lambda note, kl=L, km=kept[1], legs=list(comm_legs): make_line(
                    km, kl, "the first layer that verified; " + note, legs))
        good = [leg for leg in comm_legs if leg["verified"] and not leg.get("ipc_error")]
        if not good:
            raise SystemExit("bench.py: distributed data path failed verification on every exchange layer")
        forced = [leg for leg in good if leg["layer"] == args.comm]
        if args.comm != "auto" and forced:
            comm_layer, comm_choice = args.comm, "forced (--comm / MI355OPT_COMM)"
        else:
            comm_layer = min(good, key=lambda leg: leg["us_per_step"])["layer"]
            comm_choice = "fastest verified layer of comm_ab_legs" + \
                ("" if args.comm == "auto" else f" ('{args.comm}' asked for but not available / not verified)")
        wd.arm(f"headline on '{comm_layer}'", args.leg_timeout)
        m = kept[1] if kept[0] == comm_layer else measure(comm_layer)
        wd.disarm()
        peer_memory = comm_layer not in ("rccl", "rccl2")
        dist.barrier()
    else:
        m = measure(None)
    dt, solves, roofline, moved_bytes, value, wakeup_steps = (m[k] for k in ("dt", "solves", "roofline", "moved_bytes",
                                                                               "value", "wakeup_steps"))

    # ---- extra legs and CPU baselines: rank 0 of a single-GPU run -------------------------------------------
    plain_leg = big_leg = cpu = cpu_all = parity = leg3 = leg5 = None
    if rank == 0 and world == 1 and not use_comm:
        model = m.pop("model", None)
        if not args.no_legs:
            try:
                plain_label = (f"cfg2 St({n},{p}), generic CSR path: the matrix in " +
                               ("plain 12-byte entries (what any sparse SPD matrix gets)" if packed
                                else "4-byte value-indexed entries"))
                try:
                    plain_leg = generic_leg_in_its_own_process(n, p, packed, plain_label)
                except Exception as e:  # noqa: BLE001  (fall back to the in-process leg)
                    print("bench.py: %s; running the generic leg in this process" % e, file=sys.stderr)
                    plain_leg = extra_leg(ctx, nx, p, min(args.steps, 200), 20, packed=not packed, label=plain_label)
                big_leg = extra_leg(ctx, 200, p, 50, 10, packed=packed,
                                    label=f"St(8000000,{p}), 200^3 grid, one GPU: beyond the Infinity Cache")
            except capi.MiError as e:  # an extra leg must never take the headline down with it
                print("bench.py: extra leg failed: %s" % e, file=sys.stderr)
        if not args.no_legs:
            # cfg3 / cfg5: each leg is skipped once the legs together have used up --leg-budget seconds; a leg that
            # fails costs itself, never the line
            t_legs = time.perf_counter()
            for name, fn in (("cfg3", cfg3_leg), ("cfg5", cfg5_leg)):
                used = time.perf_counter() - t_legs
                if used > args.leg_budget:
                    rec = {"skipped": f"--leg-budget {args.leg_budget:.0f} s used up ({used:.0f} s)"}
                else:
                    try:
                        rec = fn(ctx)
                    except Exception as e:  # noqa: BLE001
                        print(f"bench.py: {name} leg failed: {e}", file=sys.stderr)
                        rec = {"failed": str(e)[:300]}
                if name == "cfg3":
                    leg3 = rec
                else:
                    leg5 = rec
        if not args.no_cpu_baseline:
            ts = m["timed_solve"]
            keep = {} if ts is not None else None
            cpu, cpu_all = cpu_baseline(nx, ny, nz, p, rowptr, col, val, Xb, moved_bytes, bytes_per_step, keep=keep,
                                        timed_iterations=(ts or {}).get("iterations"))
            if keep:
                try:
                    # the GPU side of the parity block (CHECKER, nothing here is timed): the full solve with its traces from
                    # (a) the gradient the device computed -- what the timed solves used -- and (b) the REFERENCE's gradient
                    # uploaded bit for bit (identical inputs); the last timed solve's step was copied off the device right
                    # behind the timed region
                    g_, H_ = model
                    kw = dict(Delta=1e3, kappa_fgr=1e-12, theta=1.0)
                    g_ref = ctx.upload(keep["g"])

                    def gpu_solve(gvec, k, label):
                        rt = ctx.stpcg(gvec, H_, max_iterations=k, trace_cap=k + 2, **kw)
                        return dict({kk: v for kk, v in rt.items() if kk != "s"}, s=rt["s"].numpy(), label=label)
                    full = gpu_solve(g_, TPCG, f"full_solve_{TPCG}_iterations_device_gradient")
                    solves = [(TPCG, full),
                              (TPCG, gpu_solve(g_ref, TPCG, f"full_solve_{TPCG}_iterations_identical_inputs"))]
                    if ts["iterations"] != TPCG:
                        solves.append((ts["iterations"], ts))
                        solves.append((ts["iterations"], gpu_solve(g_ref, ts["iterations"],
                                                                   f"solve_of_{ts['iterations']}_iterations_identical_inputs")))
                    else:   # the last timed solve IS a full solve: its step must be the traced solve's, bit for bit
                        full["equals_last_timed_solve_bitwise"] = bool(np.array_equal(ts["s"], full["s"]))
                    parity = parity_block(keep, solves)
                    parity["g_rel_device_vs_reference"] = _rel(g_.numpy(), keep["g"])
                    parity["note"] = ("`identical_inputs`: STPCG's input g is the reference's gradient, bit for bit -- the "
                                      "parity of the solver, held to `bar`.  `device_gradient` / `timed_solve`: the input is "
                                      "the gradient the device computed (g_rel_device_vs_reference away from the reference's: "
                                      "cancellation in A X - X sym(X'AX) at a near-optimal iterate), which the solve's "
                                      "conditioning multiplies -- the re-associated reference (floor_*, its own gradient) "
                                      "moves as far")
                except Exception as e:  # noqa: BLE001  (the checker must never take the headline down with it)
                    print("bench.py: parity block failed: %r" % (e,), file=sys.stderr)
                    parity = {"failed": str(e)[:300]}
        model = None

    if rank == 0 and use_comm and not args.no_cpu_baseline:
        # N > 1 (r04 verdict): the reference's CPU path "in the same run" here too.  Rank 0 times it AFTER the timed region
        # while the other ranks sit at the barrier below, on the PER-GPU problem -- a grid the size of one rank's slab
        # (the same 1e6 x 3 unknowns per GPU as at N = 1), not the N-fold global one -- so that `cores` (1) and the
        # bounded sample stay what they are at N = 1; the line says so.
        try:
            lx, ly, lz = nx, ny, z1 - z0
            Xl, _ = wl.stiefel_bench_iterate(lx, ly, lz, p, eps=1e-3, seed=7)
            cpu, cpu_all = cpu_baseline(lx, ly, lz, p, *wl.laplacian_3d(lx, ly, lz), Xl, moved_bytes, bytes_per_step)
            for c_ in (cpu, cpu_all):
                c_["sample"] += (f"; N = {world}: the PER-GPU problem (a {lx}x{ly}x{lz} grid of its own, the size of one "
                                 "rank's slab), timed on rank 0's host cores after the timed region -- multiply by nothing: "
                                 "one CPU process against ONE GPU's share of the job")
                c_["per_gpu_problem"] = True
        except Exception as e:  # noqa: BLE001  (the baseline must never take the headline down with it)
            print("bench.py: CPU baseline at N > 1 failed: %s" % e, file=sys.stderr)
    if rank == 0:
        out = make_line(m, comm_layer, comm_choice, comm_legs, plain_leg, big_leg, cpu, cpu_all, parity, leg3, leg5)
        if args.dry_run_layers > 1 and world == 1:
            try:
                out["dry_run_layers"] = dry_run_layers(args.dry_run_layers, args.comm, capi.device_count())
            except Exception as e:  # noqa: BLE001
                out["dry_run_layers"] = {"failed": str(e)[:300]}
        os.write(_RESULT_FD, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        ctx.comm_finalize()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
