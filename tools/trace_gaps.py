#!/usr/bin/env python3
"""Per-kernel durations and the idle gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV.
Usage: python tools/trace_gaps.py <dir with *_kernel_trace.csv> [name filter]"""
import collections, csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = collections.defaultdict(list)
gap_after = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    m = re.search(r"(k_\w+)", a["Kernel_Name"])
    k = m.group(1) if m else a["Kernel_Name"][:30]
    dur[k].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    gap_after[k].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    d, g = dur[k], sorted(gap_after[k])
    # launches that found their solve finished return at once (~4.5 us): averaged in, they make a kernel look faster
    # than it is (the cfg3 figures of early r03 were diluted that way) -- reported apart from the real ones
    med = sorted(d)[len(d) // 2]
    noop = [x for x in d if med > 12000 and x < 6000]
    real = [x for x in d if not (med > 12000 and x < 6000)]
    print("%-26s n=%5d  avg %8.2f us (real launches: n=%5d avg %8.2f us; no-ops: n=%4d)   gap to the next kernel: median %6.2f us, mean %7.2f us" %
          (k, len(d), sum(d) / len(d) / 1e3, len(real), sum(real) / max(1, len(real)) / 1e3, len(noop),
           g[len(g) // 2] / 1e3, sum(g) / len(g) / 1e3))
