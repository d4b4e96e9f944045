#!/bin/bash
# Run ON THE GPU BOX: the whole GPU suite, the rocprofv3 evidence of every bench workload on the final sources, the default bench line.
mkdir -p gpurun_out
python -m pytest tests -m gpu --maxfail=12 -q > gpurun_out/r5_tfinal.log 2>&1
rc=$?; echo "rc=$rc" >> gpurun_out/r5_tfinal.log; tail -12 gpurun_out/r5_tfinal.log
[ $rc -eq 0 ] || exit $rc
tools/collect_all_profiles.sh r05 > gpurun_out/r5_collect_all.log 2>&1
grep "==\|FAILED" gpurun_out/r5_collect_all.log
tools/collect_stats_profile.sh r05 > gpurun_out/r5_collect_stats.log 2>&1
tools/collect_stats_profile.sh r05thin4 --thin 4 --steps 800 --lags 64 > gpurun_out/r5_collect_stats4.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench_final.json 2> gpurun_out/r5_bench_final.err || { tail -20 gpurun_out/r5_bench_final.err; exit 1; }
cp bench_full.json gpurun_out/r5_bench_final_full.json
tail -c 3200 gpurun_out/r5_bench_final.json
