#!/bin/bash
# Run ON THE GPU BOX: the round's rocprofv3 evidence for every bench workload (kernel trace + PMC passes, separate runs).
mkdir -p gpurun_out
tools/collect_all_profiles.sh r05 > gpurun_out/r5_collect_all.log 2>&1
tail -30 gpurun_out/r5_collect_all.log
tools/collect_stats_profile.sh r05 > gpurun_out/r5_collect_stats.log 2>&1
tail -25 gpurun_out/r5_collect_stats.log | cut -c1-200
tools/collect_stats_profile.sh r05thin4 --thin 4 --steps 800 --lags 64 > gpurun_out/r5_collect_stats4.log 2>&1
tail -12 gpurun_out/r5_collect_stats4.log | cut -c1-200
