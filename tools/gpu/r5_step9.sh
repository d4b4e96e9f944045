#!/bin/bash
# Run ON THE GPU BOX: statistics on chip -- parity with the per-draw path and every statistics test, then timings.
mkdir -p gpurun_out
python -m pytest tests/test_hip_statistics.py tests/test_hip_fuzz.py -m gpu --maxfail=8 -q > gpurun_out/r5_t9.log 2>&1
rc=$?; echo "rc=$rc" >> gpurun_out/r5_t9.log; tail -30 gpurun_out/r5_t9.log | cut -c1-250
[ $rc -eq 0 ] || exit $rc
for E in 0 1; do
  GSSS_STATS_ONCHIP=$E python tools/bench_stats.py | cut -c1-260
  GSSS_STATS_ONCHIP=$E python tools/bench_stats.py --thin 4 --steps 800 --lags 64 | cut -c1-260
  GSSS_STATS_ONCHIP=$E python tools/bench_stats.py --workload bingham_d10 --thin 8 --steps 800 --lags 64 | cut -c1-260
done
