#!/bin/bash
# Run ON THE GPU BOX: A/B timing of side-by-side builds (python -m geosss_amd.build --out geosss_amd/libgsss_<tag>.so with
# GSSS_HIPCC_FLAGS=-D...): kernel_ms of the given workloads under each library.  Usage: tools/ab_libs.sh "<libs>" "<workload:chains> ..."
for lib in $1; do
  for wc in $2; do
    wl=${wc%%:*}; n=${wc##*:}
    GSSS_HIP_LIB=$PWD/geosss_amd/$lib python bench.py --workload $wl --chains $n --steps 5 --warmup 1 --no-cpu-baseline --no-ess --no-configs 2>/dev/null |
      python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$lib', r['config']['target'], r['config']['chains_per_gpu'], r['config']['kernel'], '%.3f ms' % r['kernel_ms'], '%.3e' % r['value'])"
  done
done
