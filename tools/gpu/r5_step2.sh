#!/bin/bash
# Run ON THE GPU BOX: parity of the uneven curve layouts, then A/B timings against the four-quad builds, then step 1's items.
mkdir -p gpurun_out
python -m pytest tests/test_hip_screen_bounds.py tests/test_hip_parity.py tests/test_hip_bench_shape.py -m gpu -x -q -k "screen or bounds or curve or sample_in_blocks or sample_plans or sample_api or sin_cos or exp2 or log2 or sqrt" > gpurun_out/r5_t2.log 2>&1
rc=$?; echo "rc=$rc" >> gpurun_out/r5_t2.log; tail -15 gpurun_out/r5_t2.log
[ $rc -eq 0 ] || exit $rc
for W in curve_d50 curve_d200; do
  for T in 0 1; do
    GSSS_CURVE_TAIL=$T python bench.py --workload $W --chains 100000 --steps 10 --warmup 2 --no-configs --no-cpu-baseline --no-ess > gpurun_out/r5_ab_tail_${W}_$T.json 2> gpurun_out/r5_ab_tail_${W}_$T.err || { tail -5 gpurun_out/r5_ab_tail_${W}_$T.err; exit 1; }
    python - <<PY
import json
j=json.loads(open("gpurun_out/r5_ab_tail_${W}_$T.json").read().strip().splitlines()[-1])
print("$W tail=$T", j["config"]["kernel"], "kernel_ms", j["kernel_ms"], "value %.4e" % j["value"], "tries", j["tries_per_step"])
PY
  done
done
tools/ab_libs.sh "libgsss_hip.so libgsss_try32.so libgsss_hip.so libgsss_try32.so" "vmfmix_readme:1000000 vmfmix_k10_kappa500:1000000" > gpurun_out/r5_ab_try32.log 2>&1; cat gpurun_out/r5_ab_try32.log
python tools/bench_host_api.py --json gpurun_out/r5_host_api.json > gpurun_out/r5_host_api.log 2>&1 || { tail -20 gpurun_out/r5_host_api.log; exit 1; }
cat gpurun_out/r5_host_api.log
