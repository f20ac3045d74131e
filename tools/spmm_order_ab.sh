for e in "$@"; do
  v="$e"; [ "$e" = "-" ] && v=""
  echo "$e: $(env $v python tools/bench_extra.py cfg5 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('spmm24 %.1f us' % d['spmm_colmajor_24']['us'], ' lobpcg %.3f ms/it' % d['lobpcg_ms_per_iteration'], d['ritz_0'])
")"
done
