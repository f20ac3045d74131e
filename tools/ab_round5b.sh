mkdir -p gpurun_out/r05b
python tools/sync_latency.py > gpurun_out/r05b/sync_latency.json 2>&1; cat gpurun_out/r05b/sync_latency.json
for rep in 1 2 3; do
 for v in head new nopoll; do
  case $v in
   head) E="MI355OPT_LIB=$PWD/optimization_amd/libmi355opt_head.so";;
   new) E="A=1";;
   nopoll) E="MI355OPT_NO_POLLED_SYNC=1";;
  esac
  echo "== $v rep $rep"
  env $E python tools/bench_tnt.py 1e-2 both 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print({k:(round(v['ms_per_outer'],4), v['host_syncs']) for k,v in d.items()})" 
 done
done > gpurun_out/r05b/ab_tnt.log 2>&1
cat gpurun_out/r05b/ab_tnt.log
