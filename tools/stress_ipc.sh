#!/bin/bash
# Loops the 2- and 3-rank peer-memory worker (tests/ipc_worker.py) on one GPU and prints every run whose sharded
# products / solves disagree with the single-process references.  Usage (GPU box): bash tools/stress_ipc.sh
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
fails=0
for i in $(seq 1 16); do
  for w in 2 3; do
    d=$(mktemp -d)
    IPC_WORKER_OUT=$d timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29300 + i * 4 + w)) tests/ipc_worker.py > /dev/null 2>&1
    python - "$d" "$i" "$w" <<'PY'
import json, os, sys
d, i, w = sys.argv[1:]
outs = [json.load(open(os.path.join(d, f))) for f in sorted(os.listdir(d))]
bad = [(o.get("rank"), o["dot_err"], o["spmm_err"], o["spmm2_err"], o["g_err"], o["s_err"], o["ipc_error"]) for o in outs
       if not (o["dot_err"] < 1e-13 and o["spmm_err"] == 0.0 and o["spmm2_err"] == 0.0 and o["g_err"] < 1e-12 and o["s_err"] < 1e-9)]
print(f"run {i} world {w}: {len(outs)} ranks", "BAD " + str(bad) if bad or len(outs) != int(w) else "ok")
PY
  done
done
