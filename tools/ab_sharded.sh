#!/bin/bash
# Same-call A/B of experiment builds (MI355OPT_BUILD_TAG builds, loaded through MI355OPT_LIB) on the SHARDED step at one
# rank (bench.py with MI355OPT_BENCH_FORCE_COMM=1: communicator of size 1, peer-memory layer, folded exchanges).
# Usage (GPU box): tools/ab_sharded.sh "" tag1 tag2 ""
cd "$(dirname "$0")/.."
for tag in "$@"; do
  lib=$PWD/optimization_amd/libmi355opt${tag:+_$tag}.so
  line=$(MI355OPT_BENCH_FORCE_COMM=1 MI355OPT_LIB=$lib python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
      --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 500 --warmup 50 --no-cpu-baseline --no-legs \
      2>/dev/null < /dev/null | tail -1)
  python - "${tag:-base}" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
k = d["roofline"]["kernels"]
print("%-8s step %.2f us " % (sys.argv[1], 1e3 * d["ms_per_step"]), {n: round(v["avg_us"], 2) for n, v in k.items() if v["launches"]})
PY
done
