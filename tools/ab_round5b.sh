#!/bin/bash
# r05 (second half): same-call A/B of the polled read-backs on the bench line at the driver's K = 20 / W = 5 and at K = 500
mkdir -p gpurun_out/r05b
for rep in 1 2 3; do
  for v in "A=1" "MI355OPT_NO_POLLED_SYNC=1"; do
    for kw in "20 5" "500 50"; do
      set -- $kw
      env $v python bench.py --no-cpu-baseline --no-legs --no-roofline --steps $1 --warmup $2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s K=%-4d us/step %.2f value %.0f' % ('$v', d['steps'], 1e3 * d['ms_per_step'], d['value']))"
    done
  done
done
