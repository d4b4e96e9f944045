#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: rocprofv3 evidence for the dominant kernel of every bench
# workload.  Usage: tools/collect_all_profiles.sh <tag>   ->  gpurun_out/prof_<tag>_<workload>/summary.md etc.
set -u
TAG=${1:-r02}
for W in vmfmix_readme bingham_d10 curve_d10 curve_d50 curve_d200 curve_d10_kappa500 curve_d24 vmfmix_k10_kappa500 vmfmix_readme__numpy_stream; do
  case $W in curve_*) CH=100000;; *) CH=1000000;; esac
  case $W in *__numpy_stream) WL="--workload ${W%%__numpy_stream} --rng numpy";; *) WL="--workload $W";; esac
  echo "== $W"
  tools/collect_profiles.sh ${TAG}_$W $WL --chains $CH --no-configs > gpurun_out/collect_${TAG}_$W.log 2>&1 || { echo "FAILED $W"; tail -5 gpurun_out/collect_${TAG}_$W.log; }
  tail -1 gpurun_out/collect_${TAG}_$W.log | cut -c1-200
done
