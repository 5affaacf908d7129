timeout 300 python scripts/chain_check.py 2>&1 | tail -4
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['parity']['max_abs_dlogit'], ' | '.join(k['name'][:12]+' %.3f'%k['avg_ms'] for k in d['derived']['kernels']))
"; done
