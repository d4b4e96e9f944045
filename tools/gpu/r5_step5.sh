#!/bin/bash
# Run ON THE GPU BOX: the screened LANE kernel for curve targets at 10^5 chains against the group kernel (GSSS_CURVE_LANE=1).
mkdir -p gpurun_out
for W in curve_d10 curve_d24; do
 for N in 100000 1000000; do
  for E in 0 1; do
    GSSS_CURVE_LANE=$E python bench.py --workload $W --chains $N --steps 5 --warmup 1 --no-configs --no-cpu-baseline --no-ess 2> gpurun_out/r5_lane_$W.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W lane=$E', j['config']['chains_per_gpu'], j['config']['kernel'], 'kernel_ms', round(j['kernel_ms'],3), 'value %.4e' % j['value'], 'tries', round(j['tries_per_step'],3))" || tail -3 gpurun_out/r5_lane_$W.err
  done
 done
done
