#!/usr/bin/env python3
"""one line per run of a tools/bench_tnt.py JSON: ms per outer iteration, outer / inner counts, host syncs per outer"""
import json
import sys
for k, v in json.load(open(sys.argv[1])).items():
    print("%-38s %.4f ms/outer  outer %d inner %d  syncs/outer %.2f  f %.12g" % (k, v["ms_per_outer"], v["outer"], v["inner_total"], v["host_syncs_per_outer"], v["f"]))
