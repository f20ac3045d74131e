for c in 24 16 8 4; do
  echo "chunk $c: $(MI355OPT_SPMM_PK_CHUNK=$c python tools/bench_extra.py cfg5 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('spmm24 %.1f us' % d['spmm_colmajor_24']['us'], ' lobpcg %.3f ms/it' % d['lobpcg_ms_per_iteration'], d['ritz_0'])
")"
done
