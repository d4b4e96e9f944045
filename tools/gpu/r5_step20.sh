#!/bin/bash
for W in bingham_d50 bingham_d50_dense; do
  python bench.py --workload $W --chains 1000000 --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-ess 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W', j['config']['kernel'], 'kernel_ms', round(j['kernel_ms'],2), 'value %.4e' % j['value'], 'tries', round(j['tries_per_step'],3), 'valu', round(j['roofline_valu']['frac'],3))"
done
