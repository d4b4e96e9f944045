#!/bin/bash
# Run ON THE GPU BOX: like collect_all_profiles.sh for the named workloads only.  Usage: tools/collect_some_profiles.sh <tag> <workload> ...
set -u
TAG=$1; shift
for W in "$@"; do
  case $W in curve_*) CH=100000;; *) CH=1000000;; esac
  echo "== $W"
  tools/collect_profiles.sh ${TAG}_$W --workload $W --chains $CH --no-configs > gpurun_out/collect_${TAG}_$W.log 2>&1 || { echo "FAILED $W"; tail -5 gpurun_out/collect_${TAG}_$W.log; }
  tail -1 gpurun_out/collect_${TAG}_$W.log | cut -c1-200
done
