#!/bin/bash
# Run ON THE GPU BOX: kernel time and WRITE_SIZE / FETCH_SIZE of one launch shape under the two layouts of the retained rows.
export TMPDIR=/tmp
for wl in ${WORKLOADS:-bingham_d10 vmfmix_readme curve_d10 curve_d50}; do
  case $wl in curve_*) CH=100000;; *) CH=1000000;; esac
  for lay in components chains; do
    ARGS="--workload $wl --chains $CH --steps 4 --warmup 1 --no-cpu-baseline --no-ess --no-configs --layout $lay"
    python bench.py $ARGS 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$wl', '$lay', '%.3f ms' % r['kernel_ms'])"
    for c in WRITE_SIZE FETCH_SIZE; do
      rm -rf gpurun_out/lay_$c; rocprofv3 --pmc $c --output-format csv -d gpurun_out/lay_$c -- python3 bench.py $ARGS > /dev/null 2>&1
      python3 tools/pmc_summary.py gpurun_out/lay_$c | grep -A1 "curvespec\|screened" | grep "$c" | head -1
    done
  done
done
