#!/usr/bin/env python3
"""Kernel totals of a whole run from a rocprofv3 database (rocpd .db, the default output of `rocprofv3 --kernel-trace`):
calls, total and average duration per kernel name, and the sum of the gaps between consecutive kernels.
    cd /tmp && rocprofv3 --kernel-trace -d $REPO/gpurun_out/conv_db -o t -- python $REPO/tools/cfg5_converge.py max_iters=8000
    python tools/run_trace_totals.py gpurun_out/conv_db/t_results.db          (runs anywhere: sqlite3 only)
This is what the "after" table of profiles/r04_cfg5_run_kernel_totals.md was made with."""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
short = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))[:48]
tot, cnt = collections.Counter(), collections.Counter()
for n, s, e in rows:
    tot[short(n)] += (e - s) / 1e3
    cnt[short(n)] += 1
T = sum(tot.values())
print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
for k, v in tot.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 16):
    print("| %s | %d | %.1f | %.1f | %.1f |" % (k, cnt[k], v / 1e3, v / cnt[k], 100 * v / T))
print("| all kernels | %d | **%.1f** | | |" % (len(rows), T / 1e3))
gaps = [(rows[i + 1][1] - rows[i][2]) / 1e3 for i in range(len(rows) - 1)]
print("\ngaps between consecutive kernels shorter than 5 ms: %.1f ms in total, %d longer than 100 us"
      % (sum(g for g in gaps if g < 5000) / 1e3, sum(1 for g in gaps if 100 < g < 5000)))
