for e in "-" "MI355OPT_HESS_GRID=128"; do
  v="$e"; [ "$e" = "-" ] && v=""
  env $v python bench.py --no-cpu-baseline --steps 300 --warmup 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$e', 'step %.2f us' % (1e3*d['ms_per_step']), 'value', round(d['value']), 'plain', d['generic_csr_leg'], 'big', d['beyond_cache_leg'])
"
done
