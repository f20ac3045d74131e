#!/bin/bash
# Same-call A/B of experiment builds of the library (MI355OPT_BUILD_TAG builds, loaded through MI355OPT_LIB): the bench
# line's step time and per-kernel event timings for each.  Usage (GPU box): tools/ab_libs.sh "" nt1 nt3 ... [-- bench args]
cd "$(dirname "$0")/.."
for tag in "$@"; do
  lib=optimization_amd/libmi355opt${tag:+_$tag}.so
  for rep in 1 2; do
    MI355OPT_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-legs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline']['kernels']
print('%-6s step %.2f us  value %.0f  ' % ('${tag:-base}', 1e3 * d['ms_per_step'], d['value']), {n: round(v['avg_us'], 2) for n, v in k.items() if v['launches']})"
  done
done
