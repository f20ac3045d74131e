(timeout 900 python -m pytest tests/test_gpu_lobpcg.py -x -q -m gpu 2>&1 | tail -5)
cp optimization_amd/libmi355opt.so /tmp/new.so
for rep in 1 2 3; do
 for lib in old new; do
  if [ $lib = old ]; then cp optimization_amd/libmi355opt_old.so optimization_amd/libmi355opt.so; else cp /tmp/new.so optimization_amd/libmi355opt.so; fi
  a=$(timeout 300 python tools/bench_extra.py cfg5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['gram_SAS']['us'], d['update_72x48(X and P fused)']['us'], d['lobpcg_ms_per_iteration'])")
  b=$(timeout 300 python tools/cfg5_converge.py max_iters=8000 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['iterations'], round(d['ms_per_iteration'],4), d['theta'][0])")
  echo "$lib: $a | $b"
 done
done
