#!/bin/bash
# Same-call A/B of experiment builds on the wide-row Stiefel step (tools/bench_extra.py wide): tools/wide_ab.sh "" w3 ...
cd "$(dirname "$0")/.."
for tag in "$@"; do
  lib=optimization_amd/libmi355opt${tag:+_$tag}.so
  echo "== ${tag:-base}"
  MI355OPT_LIB=$PWD/$lib python tools/bench_extra.py wide 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(d['p'], 'step %.1f us  frac %.3f ' % (d['us_per_step'], d['frac_of_8TBps']), {k: (round(v['avg_us_event_pairs'], 1), round(v['frac_of_8TBps_event_pairs'], 3)) for k, v in d['kernels'].items()})"
done
