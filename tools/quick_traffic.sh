#!/bin/bash
# Run ON THE GPU BOX: HBM bytes per launch of one workload's dominant kernel (two PMC passes: FETCH_SIZE, WRITE_SIZE; FETCH doubled
# as MI355X_MICROARCH.md prescribes) and its kernel time, for A/B of byte-side changes.  Environment knobs (GSSS_ONE_PER_LANE,
# GSSS_STAGE_ROWS, GSSS_SLICE_STEPS ...) are exported by the caller.  Usage: tools/quick_traffic.sh <tag> <workload> <chains> [bench args]
set -u
TAG=$1; W=$2; CH=$3; shift 3
OUT=gpurun_out/qt_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-ess --workload $W --chains $CH --no-configs $*"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS > $OUT/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py $ARGS > $OUT/write.log 2>&1 || exit 1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
v = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("fetch", "write"):
    for f in glob.glob(f"{out}/{sub}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            v[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
name = max((k for k in v if "gsss" in k and "kernel<" in k), key=lambda k: sum(v[k].get("WRITE_SIZE", [0])))
fetch = sum(v[name]["FETCH_SIZE"]) / len(v[name]["FETCH_SIZE"]); write = sum(v[name]["WRITE_SIZE"]) / len(v[name]["WRITE_SIZE"])
line = [l for l in open(f"{out}/fetch.log") if l.startswith('{"metric"')][-1]
j = json.loads(line)
alg = j["roofline"]["algorithmic_bytes"]
print(f"{tag}: {name.split('(')[0][:70]}  kernel_ms(under pmc) {j['kernel_ms']:.2f}  fetch {fetch / 1024:.1f} MiB  write {write / 1024:.1f} MiB  "
      f"2F+W {(2 * fetch + write) * 1024 / 1e6:.0f} MB  algorithmic {alg / 1e6:.0f} MB  handover {j['roofline']['handover_bytes'] / 1e6:.0f} MB  "
      f"ratio {(2 * fetch + write) * 1024 / alg:.3f}  layout {j['config']['kept_rows_layout']} sliced {j['config']['sliced_fraction']}")
PY
