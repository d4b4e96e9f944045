#!/bin/bash
# Quick throughput table of every bench workload (GPU box). Usage: tools/bench_all.sh [extra bench args]
for w in vmfmix_readme vmfmix_k10_kappa500 bingham_d10; do
  timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-ess "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-22s %-6s %-10s %.3e' % (d['config']['target'], d['config']['mode'], d['config']['kernel'], d['value']))"
done
for w in curve_d10 curve_d24 curve_d50 curve_d200; do
  timeout -k 10 300 python bench.py --workload $w --chains 100000 --inner 50 --steps 3 --warmup 1 --no-cpu-baseline --no-ess "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-22s %-6s %-10s %.3e' % (d['config']['target'], d['config']['mode'], d['config']['kernel'], d['value']))"
done
