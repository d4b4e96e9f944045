#!/bin/bash
# Run ON THE GPU BOX: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and kernel time of one workload's dominant kernel
# under several GSSS_SLICE_STEPS.  Usage: tools/traffic_experiment.sh <workload> <chains> <slice_steps> ...
set -u
W=$1; CH=$2; shift 2
export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-ess --workload $W --chains $CH --no-configs ${TX_EXTRA:-}"  # TX_EXTRA: more bench flags
for S in "$@"; do
  export GSSS_SLICE_STEPS=$S
  OUT=gpurun_out/tx_${W}_$S
  mkdir -p $OUT
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS > $OUT/fetch.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py $ARGS > $OUT/write.log 2>&1 || exit 1
  python3 bench.py $ARGS 2>/dev/null > $OUT/plain.log || exit 1
  python3 - $OUT $S <<'PY'
import sys, glob, csv, json, collections
out, S = sys.argv[1], sys.argv[2]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("fetch", "write"):
    for f in glob.glob(f"{out}/{sub}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
v = per[max(per, key=lambda k: sum(per[k].get("WRITE_SIZE", [0])))]   # the sampler kernel: the one that writes most
j = json.loads([l for l in open(f"{out}/plain.log") if l.startswith("{")][0])
f, w = (sum(v[k]) / max(len(v[k]), 1) for k in ("FETCH_SIZE", "WRITE_SIZE"))
alg = j["roofline"]["achieved"] * 1e9 * j["kernel_ms"] * 1e-3
print(f"slice {S:>4}: kernel {j['kernel_ms']:.2f} ms  sliced_fraction {j['config'].get('sliced_fraction')}  fetch {f/1024:.1f} MiB  write {w/1024:.1f} MiB  "
      f"2F+W {(2*f+w)*1024/1e6:.0f} MB  algorithmic {alg/1e6:.0f} MB  ratio {(2*f+w)*1024/alg:.2f}")
PY
done
