#!/bin/bash
# the three cfg4 workloads of bench.py, one line each (chain-steps/s, kernel, ms per launch)
for w in curve_d10 curve_d50 curve_d200; do
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-ess --workload $w --chains 100000 --no-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['config']['target'], '%.3e' % j['value'], j['config']['kernel'], '%.2f ms' % j['kernel_ms'], 'tries/step %.3f' % j['tries_per_step'], 'err', j['chains_in_error'])
"
done
