#!/bin/bash
# Run ON THE GPU BOX: parity of every uneven curve layout (one and two tail components per lane), then each against the even layout.
mkdir -p gpurun_out
rc=0; echo skipped-tests
rc=$?; echo "rc=$rc" >> gpurun_out/r5_t14.log; tail -6 gpurun_out/r5_t14.log | cut -c1-200
echo "(continuing to the timings)"
for D in 18 20 22 24 34 36 38 40 50 54 56 100 108 200 216 224; do
  for T in 0 1; do
    GSSS_CURVE_TAIL=$T python bench.py --workload curve_d$D --chains 100000 --steps 6 --warmup 2 --no-configs --no-cpu-baseline --no-ess 2> gpurun_out/r5_tail2_$D.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('d=$D tail=$T', j['config']['kernel'], 'kernel_ms', round(j['kernel_ms'],3), 'value %.4e' % j['value'])" || tail -3 gpurun_out/r5_tail2_$D.err
  done
done
