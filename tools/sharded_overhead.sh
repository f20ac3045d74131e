#!/bin/bash
# Per-step cost of the SHARDED code path on one rank (no peer to wait for: launches, prologue work and the exchange
# layer's own cost only), against the plain single-GPU step.  Usage (GPU box): tools/sharded_overhead.sh > out.json
cd "$(dirname "$0")/.."
run() {  # label, env...
  label=$1; shift
  line=$(env "$@" python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 \
      bench.py --gpus 1 --steps 500 --warmup 50 --no-cpu-baseline --no-legs 2>/dev/null | tail -1)
  python - "$label" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
k = d["roofline"]["kernels"]
print(json.dumps({"leg": sys.argv[1], "us_per_step": 1e3 * d["ms_per_step"], "parallelism": d["config"]["parallelism"],
                  "kernels_us_event_pairs": {n: round(v["avg_us"], 2) for n, v in k.items() if v["launches"]}}))
PY
}
run "single GPU (no communicator)" MI355OPT_X=1
run "sharded path, peer-memory layer, exchanges folded into the consumers (default)" MI355OPT_BENCH_FORCE_COMM=1
run "sharded path, peer-memory layer, separate exchange kernels" MI355OPT_BENCH_FORCE_COMM=1 MI355OPT_NO_FOLD=1
run "sharded path, RCCL (all-reduce of partial rows, send/recv halo)" MI355OPT_BENCH_FORCE_COMM=1 MI355OPT_COMM=rccl
