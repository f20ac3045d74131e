#!/usr/bin/env python3
"""profiles/<tag>_bench_legs.md: the rocprofv3 kernel table of the driver's own command with its cfg3 / cfg5 legs
(tools/profile_run.sh: <tag>_bench_legs_kernel_stats.csv, <tag>_trace_legs.json) next to what the un-profiled bench line
of the same build says (<tag>_bench_default.json).  Usage: python tools/legs_summary.py r06"""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = os.path.join(ROOT, "profiles")
rows = list(csv.DictReader(open(os.path.join(P, f"{tag}_bench_legs_kernel_stats.csv"))))
d = json.loads(open(os.path.join(P, f"{tag}_trace_legs.json")).read().strip().splitlines()[-1])
dd = json.loads(open(os.path.join(P, f"{tag}_bench_default.json")).read().strip().splitlines()[0])


def kname(n):
    m = re.search(r"(k_\w+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:40]


leg3, leg5 = dd["cfg3_leg"], dd["cfg5_leg"]
notes = {
    "k_bsr3_spmv<true, true>": f"cfg3 HVP inside the fused solves: `cfg3_leg.inner_step.roofline.avg_launch_us_event_pairs` = {leg3['inner_step']['roofline']['avg_launch_us_event_pairs']:.1f} µs over REAL launches only (solves capped at their length); the profiler's average includes the ~4.5 µs no-op launches the host enqueues speculatively behind a solve's exit (11 % of the launches of these 22-iteration solves)",
    "k_bsr3_spmv<false, true>": "cfg3: the `dm` product of the TNT trial step (no fused dots)",
    "k_so3_model<true, true, true>": "cfg3 model assembly (trial step of the TNT outer iteration; r06: neighbours gathered as quaternions)",
    "k_spmm_colmajor_win<8, 7, 2, true, false>": f"cfg5 `A [W P]`, 48 columns (and 24-column products of the first iterations); `cfg5_leg.kernel_families.csr_spmm` averages both products of an iteration: {leg5['kernel_families']['csr_spmm']['avg_us_event_pairs']:.0f} µs per call",
    "k_spmm_colmajor_win<8, 7, 2, true, true>": "cfg5 `A X` of the new Ritz block with the residual and its norms fused (24 columns)",
    "k_gram_pair_sym<5, true, true>": f"cfg5 Gram pair at ns = 72 (25 tiles): `cfg5_leg.gram_pair.avg_us_event_pairs` = {leg5['gram_pair']['avg_us_event_pairs']:.0f} µs per CALL averaged over all {leg5['gram_pair']['calls']} calls of a run, three of which are on narrower bases",
    "k_panel_update_mfma2<18>": f"cfg5 Ritz update (X and P from one pass): `lobpcg_update` {leg5['kernel_families']['lobpcg_update']['avg_us_event_pairs']:.0f} µs",
    "k_st_hess_fused<3, false, false, true, true, 7, 1, false>": f"cfg2 one-pass Hessian, window form — headline AND beyond-cache leg (1/8 of the calls at 8x the rows): headline alone `roofline.avg_launch_us` = {dd['roofline']['avg_launch_us']:.1f} µs, beyond cache {dd['beyond_cache_leg']['kernels']['stiefel_hess_fused']['avg_us']:.0f} µs; the headline-only trace is `profiles/{tag}_kernel_stats.csv`",
}
md = f"""# {tag}: the driver's own command with its cfg3 / cfg5 legs under `rocprofv3 --kernel-trace --stats`

`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline`
(`tools/profile_run.sh`; full table: `profiles/{tag}_bench_legs_kernel_stats.csv`; the line that run printed:
`profiles/{tag}_trace_legs.json`; the un-profiled line of the same build: `profiles/{tag}_bench_default.json`).  One process runs
the cfg2 headline (1000 wake-up + 5 + 20 + 200 steps), the beyond-cache leg (St(8e6,3): the same kernel names at 8x the
rows), cfg3 and cfg5 (the generic-CSR leg runs in a process of its own and is not in here) — so a kernel name's average
mixes workloads where the same kernel serves several legs; the legs' dominant kernels have names of their own.

| kernel | calls | rocprofv3 avg µs | what the bench line says (event pairs, same build, un-profiled run) |
|---|---|---|---|
"""
for r in rows[:14]:
    n = kname(r["Name"])
    md += f"| `{n}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {notes.get(n, '')} |\n"
g = [r for r in rows if "k_gram_pair_sym<5, true, true>" in r["Name"]]
md += f"""
The line printed under the profiler: cfg3 inner step {d['cfg3_leg']['inner_step']['us']:.1f} µs, outer iteration {d['cfg3_leg']['outer_iteration']['us']:.0f} µs, cfg5 iteration
{d['cfg5_leg']['us']:.0f} µs (un-profiled: {leg3['inner_step']['us']:.1f} / {leg3['outer_iteration']['us']:.0f} / {leg5['us']:.0f}); the profiler's per-dispatch bookkeeping costs the
launch-bound cfg3 outer loop most.

Gram pair: the profiler's per-kernel duration ({float(g[0]['AverageNs']) / 1e3 if g else float('nan'):.0f} µs here; `profiles/{tag}_lobpcg_kernel_stats.csv` and the stand-alone
A/B `profiles/{tag}_gram_pair_ab.json` agree with it) is above the {leg5['gram_pair']['avg_us_event_pairs']:.0f} µs the leg's event pairs give per call inside the LOBPCG
run.  The TF/s figure of `cfg5_leg.gram_pair` is computed from the leg's own per-call time ({leg5['gram_pair']['TFLOPs']:.0f} TF/s on the 25 executed
tiles); priced by the profiler's duration it is {leg5['gram_pair']['executed_GF'] / (float(g[0]['AverageNs']) / 1e3) * 1e3 if g else float('nan'):.0f} TF/s.
"""
open(os.path.join(P, f"{tag}_bench_legs.md"), "w").write(md)
print(md[-900:])
