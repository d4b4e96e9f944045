#!/bin/bash
# Run ON THE GPU BOX: the whole GPU suite on the philox-v3 stream, then the default bench line and the d = 100 tail A/B.
mkdir -p gpurun_out
python -m pytest tests -m gpu --maxfail=12 -q > gpurun_out/r5_t3.log 2>&1
rc=$?; echo "rc=$rc" >> gpurun_out/r5_t3.log; tail -25 gpurun_out/r5_t3.log
[ $rc -eq 0 ] || exit $rc
python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench3.json 2> gpurun_out/r5_bench3.err || { tail -20 gpurun_out/r5_bench3.err; exit 1; }
cp bench_full.json gpurun_out/r5_bench3_full.json
tail -c 3500 gpurun_out/r5_bench3.json; echo; wc -c gpurun_out/r5_bench3.json
for T in 0 2; do
  GSSS_CURVE_TAIL=$T python bench.py --workload curve_d200 --chains 100000 --steps 10 --warmup 2 --no-configs --no-cpu-baseline --no-ess > gpurun_out/r5_ab_tail_curve_d200_$T.json 2> gpurun_out/r5_ab_tail_curve_d200_$T.err || { tail -5 gpurun_out/r5_ab_tail_curve_d200_$T.err; exit 1; }
  python - <<PY
import json
j=json.loads(open("gpurun_out/r5_ab_tail_curve_d200_$T.json").read().strip().splitlines()[-1])
print("curve_d200 tail=$T", j["config"]["kernel"], "kernel_ms", j["kernel_ms"], "value %.4e" % j["value"], "tries", j["tries_per_step"])
PY
done
for T in 0 1; do
  GSSS_CURVE_TAIL=$T python bench.py --workload curve_d100 --chains 100000 --steps 10 --warmup 2 --no-configs --no-cpu-baseline --no-ess > gpurun_out/r5_ab_tail_curve_d100_$T.json 2> gpurun_out/r5_ab_tail_curve_d100_$T.err || { tail -5 gpurun_out/r5_ab_tail_curve_d100_$T.err; exit 1; }
  python - <<PY
import json
j=json.loads(open("gpurun_out/r5_ab_tail_curve_d100_$T.json").read().strip().splitlines()[-1])
print("curve_d100 tail=$T", j["config"]["kernel"], "kernel_ms", j["kernel_ms"], "value %.4e" % j["value"], "tries", j["tries_per_step"])
PY
done
