#!/bin/bash
# Run ON THE GPU BOX from the repo root: rocprofv3 evidence for one `Stats = true` launch shape (tools/bench_stats.py: 10^6 README
# chains, running statistics on every state, lag sums on, nothing stored).  Usage: tools/collect_stats_profile.sh <tag> [bench_stats args]
set -u
TAG=${1:-r05}; shift || true
OUT=gpurun_out/prof_${TAG}_stats
export TMPDIR=/tmp
mkdir -p $OUT
ARGS="$*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_stats.py $ARGS > $OUT/trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 tools/bench_stats.py $ARGS > $OUT/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 tools/bench_stats.py $ARGS > $OUT/write.log 2>&1 || exit 1
{
  echo "# rocprofv3 summary ($TAG, statistics launch): python3 tools/bench_stats.py $ARGS"
  echo; echo "## tool line (from the kernel-trace run)"; grep -h '^{"workload"' $OUT/trace.log
  echo; echo "## kernel-trace --stats (top kernels)"; head -5 $OUT/trace/*/*_kernel_stats.csv
  echo; echo "## PMC (mean per dispatch; HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB)"; python3 tools/pmc_summary.py $OUT/fetch $OUT/write
} > $OUT/summary.md
cat $OUT/summary.md | cut -c1-240
