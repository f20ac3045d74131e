mkdir -p gpurun_out/r05b
export MI355OPT_BENCH_ONE_GPU=1
for N in 2 4 8; do
  port=$((29000 + RANDOM % 900))
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $port bench.py --gpus $N --steps 20 --warmup 5 > gpurun_out/r05b/rehearsal_n$N.json 2> gpurun_out/r05b/rehearsal_n$N.err < /dev/null
  echo "N=$N rc=$?"; python - gpurun_out/r05b/rehearsal_n$N.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "n_gpus")}, "cpu_baseline" in d, [ (l.get("comm_layer"), l.get("oracle_check", {}).get("ok") if isinstance(l.get("oracle_check"), dict) else l.get("oracle_check")) for l in d.get("comm_legs", [])][:6])
PY
done
