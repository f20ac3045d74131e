#!/bin/bash
# Same-call A/B of the MULTI-RANK bench flow on ONE GPU (ranks share the device through the peer-memory layer; only the
# ratio between variants means anything: the ranks time-slice the GPU and a kernel that waits for its peers holds CUs the
# peers need).  Each variant is a list of VAR=value settings (or "-" for none).
# Usage (GPU box): tools/rehearsal_ab.sh "2 4" "-" "MI355OPT_HALO_PUSH_LATE=1" "MI355OPT_NO_FOLD=1"
cd "$(dirname "$0")/.."
export MI355OPT_BENCH_ONE_GPU=1
ranks=$1; shift
for N in $ranks; do
  for rep in 1 2; do
    for v in "$@"; do
      port=$((29000 + RANDOM % 900))
      envs=(); [ "$v" != "-" ] && envs=($v)
      env "${envs[@]}" timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
        --master-port $port bench.py --gpus $N --steps 60 --warmup 10 > /tmp/reh.out 2> /tmp/reh.err < /dev/null; rc=$?
      echo "N=$N [$v] rc=$rc $(python -c "import json;d=json.load(open('/tmp/reh.out'));print(round(d['ms_per_step']*1e3,1), 'us/step')" 2>/dev/null)"
      grep -i "falling back\|timed out" /tmp/reh.err | head -1
    done
  done
done
