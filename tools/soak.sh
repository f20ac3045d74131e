#!/bin/bash
# Repeats the bench (1 rank) and its one-GPU multi-rank rehearsal; prints one line per run.  Every command
# runs under `timeout`, so a hang shows up as rc=124 instead of stalling the box.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for i in $(seq 1 ${1:-8}); do
  timeout 200 python bench.py --no-cpu-baseline > /tmp/soak1.out 2> /tmp/soak1.err; rc=$?
  echo "N=1 run $i rc=$rc $(python -c "import json;d=json.load(open('/tmp/soak1.out'));print(round(d['ms_per_step']*1e3,2), d['config']['solves'])" 2>/dev/null)"
done
export MI355OPT_BENCH_ONE_GPU=1
for N in 2 2 2 2 4 4 8 8; do
  port=$((29000 + RANDOM % 900))
  timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $port bench.py --gpus $N --steps 60 --warmup 10 > /tmp/soakN.out 2> /tmp/soakN.err; rc=$?
  echo "N=$N rc=$rc lines=$(wc -l < /tmp/soakN.out) $(python -c "import json;d=json.load(open('/tmp/soakN.out'));print(round(d['ms_per_step']*1e3,1), d['config']['parallelism'][:40])" 2>/dev/null)"
  grep -i "falling back\|error\|timed out" /tmp/soakN.err | head -2
done
