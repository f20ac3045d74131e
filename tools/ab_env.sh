#!/bin/bash
# Same-call A/B of bench.py under different environment settings (each GPU box differs by a few percent, so
# variants are only ever compared inside one gpurun call).  Usage: bash tools/ab_env.sh TAG "ENV1" "ENV2" ...
# ("-" = no extra environment).  Each variant runs twice, interleaved; one summary line per run.
TAG=$1; shift
OUT=gpurun_out
mkdir -p $OUT
for rep in 1 2; do
  i=0
  for v in "$@"; do
    i=$((i+1))
    e="$v"; [ "$v" = "-" ] && e=""
    env $e python bench.py --no-cpu-baseline --no-legs --steps 300 --warmup 50 > $OUT/${TAG}_v${i}_r${rep}.json 2> $OUT/${TAG}_v${i}_r${rep}.err
    python - "$OUT/${TAG}_v${i}_r${rep}.json" "$v" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["roofline"]["kernels"]
    print("%-40s us/step %.2f  " % (sys.argv[2], 1e3 * d["ms_per_step"]) +
          "  ".join("%s %.2f" % (n, v["avg_us"]) for n, v in k.items() if v["launches"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
done
