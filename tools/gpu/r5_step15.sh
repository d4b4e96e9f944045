#!/bin/bash
mkdir -p gpurun_out
for D in 22 24 38 40 54 56 108; do
  for T in 0 1; do
    GSSS_CURVE_TAIL=$T python bench.py --workload curve_d$D --chains 100000 --steps 6 --warmup 2 --no-configs --no-cpu-baseline --no-ess 2> gpurun_out/r5_tail2_$D.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('d=$D tail=$T', j['config']['kernel'], 'kernel_ms', round(j['kernel_ms'],3), 'value %.4e' % j['value'])" || tail -3 gpurun_out/r5_tail2_$D.err
  done
done
