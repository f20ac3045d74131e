#!/usr/bin/env python3
"""The last solve of a rocprofv3 --kernel-trace of `bench.py --steps K`: every kernel with its start relative to the
solve's first kernel, its duration and the idle gap before it -- what a solve costs besides its iterations' three kernels.
Usage: python tools/solve_timeline.py <dir with *_kernel_trace.csv> [kernels to show, default 12 head + 8 tail]"""
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: (re.search(r"(k_\w+|__amd\w+)", r["Kernel_Name"]) or [None, r["Kernel_Name"][:30]])[1] if re.search(r"(k_\w+|__amd\w+)", r["Kernel_Name"]) else r["Kernel_Name"][:30]
# the last solve starts at the last init kernel
starts = [i for i, r in enumerate(rows) if "k_cg_init" in r["Kernel_Name"]]
i0 = starts[-1]
sol = rows[i0:]
t0 = int(sol[0]["Start_Timestamp"])
prev_end = int(rows[i0 - 1]["End_Timestamp"]) if i0 else t0
print("idle before the solve's first kernel: %.2f us" % ((t0 - prev_end) / 1e3))
tot_busy = 0
for j, r in enumerate(sol):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - (int(sol[j - 1]["End_Timestamp"]) if j else s)) / 1e3
    tot_busy += e - s
    if j < 12 or j >= len(sol) - 8:
        print("%4d %-28s start %9.2f us  dur %7.2f us  gap before %6.2f us" % (j, name(r), (s - t0) / 1e3, (e - s) / 1e3, gap))
span = (int(sol[-1]["End_Timestamp"]) - t0) / 1e3
print("kernels %d, span first start -> last end %.2f us, busy %.2f us" % (len(sol), span, tot_busy / 1e3))
