#!/usr/bin/env python3
"""Condense the rocprofv3 outputs that a gpurun call left under gpurun_out/ into the committed
summaries under profiles/:

  profiles/<tag>_kernel_stats.csv   the `rocprofv3 --kernel-trace --stats` per-kernel table
  profiles/pmc_traffic.json         per-kernel HBM traffic per launch from the FETCH_SIZE / WRITE_SIZE
                                    passes (separate --pmc runs), corrected as
                                    /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes:
                                    counter unit = KiB-ish kilobytes; on gfx950 FETCH_SIZE reports
                                    exactly half of a wide (16 B/lane) coalesced read stream -> doubled
  profiles/<tag>_summary.md         human-readable table (duration, algorithmic bytes, GB/s, traffic)

Usage: python tools/profile_summary.py r01
"""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_bytes  # noqa: E402

SHORT = {"k_st_hess_fused": "stiefel_hess_fused", "k_st_spmm_gram": "stiefel_spmm_gram", "k_st_finish": "stiefel_finish_dots", "k_cg_update": "cg_update",
         "k_cg_pupdate": "cg_pupdate", "k_cg_init": "cg_init", "k_cg_scalar_init": "cg_scalar_init",
         "k_cg_dot3": "cg_dot3", "k_reduce_rows_to_slots": "reduce_rows_to_slots"}


def short(name):
    m = re.search(r"(k_\w+?)(<|\()", name)
    base = m.group(1) if m else name
    if base == "k_st_finish" and "<3, false" in name:
        return "stiefel_finish_nodots"
    return SHORT.get(base, base)


def main(tag):
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    g = os.path.join(ROOT, "gpurun_out")
    import glob

    def find(sub, suffix):
        hits = sorted(glob.glob(os.path.join(g, f"{tag}_{sub}", "**", "*" + suffix), recursive=True))
        if not hits:
            raise SystemExit(f"no *{suffix} under gpurun_out/{tag}_{sub}")
        return max(hits, key=os.path.getmtime)  # gpurun merges calls: older runs' files stay around
    stats_src = find("trace", "kernel_stats.csv")
    shutil.copy(stats_src, os.path.join(out, f"{tag}_kernel_stats.csv"))
    stats = {}
    with open(stats_src) as f:
        for row in csv.DictReader(f):
            stats[short(row["Name"])] = dict(calls=int(row["Calls"]), avg_us=float(row["AverageNs"]) / 1e3,
                                             pct=float(row["Percentage"]))
    # exact bytes between L2 and the fabric per launch: tools/pmc_bytes.sh (read requests resolved by size, WRITE_SIZE),
    # calibrated on kernels of known byte counts (tools/fetch_calib.sh -> <tag>_fetch_calibration.json)
    kb = kernel_bytes(1_000_000, 6_940_000, 3, True)
    traffic = {}
    pj = os.path.join(g, f"{tag}_pmc", "bytes.json")
    if os.path.exists(pj):
        for name, v in json.load(open(pj)).items():
            k = short(name + "(")
            traffic[k] = {"read_bytes": v["read_bytes"], "write_bytes": v["write_bytes"],
                          "hbm_bytes_per_launch": v["total"], "l2_hit_rate": v["l2_hit"],
                          "algorithmic_bytes": kb.get(k), "launches_sampled": v["launches"],
                          "method": "128*TCC_EA0_RDREQ_128B + 64*TCC_EA0_RDREQ_64B + 32*TCC_EA0_RDREQ_32B + 1024*WRITE_SIZE"}
    json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    cj = os.path.join(g, f"{tag}_fetch_calibration.json")
    if os.path.exists(cj):
        shutil.copy(cj, os.path.join(out, f"{tag}_fetch_calibration.json"))
    bench = None
    bj = os.path.join(g, f"{tag}_bench.json")
    if os.path.exists(bj):
        try:
            bench = json.loads(open(bj).read().strip().splitlines()[-1])
            json.dump(bench, open(os.path.join(out, f"{tag}_bench.json"), "w"), indent=1)
        except Exception as e:  # noqa
            print("could not parse bench line:", e)
    with open(os.path.join(out, f"{tag}_summary.md"), "w") as f:
        f.write(f"# {tag}: rocprofv3 summary of `python bench.py` (cfg2, Stiefel(1e6,3), 1x MI355X)\n\n")
        f.write("Source: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 500 "
                "--warmup 50 --no-cpu-baseline --no-roofline --no-legs`; PMC: separate runs per counter group "
                "(`tools/pmc_bytes.sh`: read requests by size, WRITE_SIZE, L2 hit/miss).  bench.py's raw per-kernel figures (HIP event pairs on the launch "
                "stream, `roofline.kernels`) run 1.6-1.9 us above the profiler's kernel durations: an event pair also "
                "times its own two records; `roofline.avg_launch_us` / `frac` subtract that cost, measured live.\n\n")
        if bench:
            f.write(f"Un-profiled bench line of the same build: value = {bench['value']:.1f} GB/s "
                    f"({bench['ms_per_step'] * 1e3:.2f} us/step), roofline.frac = {bench['roofline']['frac']:.3f} "
                    f"on `{bench['roofline']['kernel']}`, cpu_baseline = {bench['cpu_baseline']['value']:.2f} GB/s "
                    f"({bench['cpu_baseline']['kind']}, {bench['cpu_baseline']['cores']} core, "
                    f"{bench['cpu_baseline']['cpu']}).\n\n")
        f.write("| kernel | calls | avg us | % time | algorithmic MB/launch | algorithmic GB/s | PMC read MB | "
                "PMC write MB | PMC total MB | fabric GB/s |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for k, s in sorted(stats.items(), key=lambda kv: -kv[1]["pct"]):
            ab = kb.get(k)
            t = traffic.get(k, {})
            f.write(f"| {k} | {s['calls']} | {s['avg_us']:.2f} | {s['pct']:.2f} | "
                    f"{(ab / 1e6 if ab else float('nan')):.1f} | "
                    f"{(ab / s['avg_us'] / 1e3 if ab else float('nan')):.0f} | "
                    f"{t.get('read_bytes', float('nan')) / 1e6:.1f} | "
                    f"{t.get('write_bytes', float('nan')) / 1e6:.1f} | "
                    f"{t.get('hbm_bytes_per_launch', float('nan')) / 1e6:.1f} | "
                    f"{t.get('hbm_bytes_per_launch', float('nan')) / s['avg_us'] / 1e3:.0f} |\n")
        f.write("\nPMC bytes are what crossed between the L2s and the fabric (Infinity Cache / HBM): every read request "
                "on gfx950 is a 128-byte line (`TCC_EA0_RDREQ_128B`; the 64- and 32-byte classes stay empty), "
                "`FETCH_SIZE` tallies it at 64 bytes; the sum over request sizes reproduces the known byte count of "
                "16/8/4-byte-per-lane streams and 24-byte row reads to 0.1 % (`<tag>_fetch_calibration.json`), and "
                "`WRITE_SIZE` is exact.  The cfg2 working set (6 fields x 24 MB + the matrix) is smaller than the "
                "256 MiB Infinity Cache, so these bytes are mostly served from it, not from HBM.\n")
    for name in ("extra", "lsqr", "tnt"):
        src = os.path.join(g, f"{tag}_{name}.json")
        if os.path.exists(src) and os.path.getsize(src) > 0:
            dst = {"extra": f"{tag}_extra_cfg3_cfg5.json"}.get(name, f"{tag}_{name}.json")
            shutil.copy(src, os.path.join(out, dst))
    print(open(os.path.join(out, f"{tag}_summary.md")).read())


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
