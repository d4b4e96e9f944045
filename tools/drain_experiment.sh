#!/bin/bash
# the last partial round of equal-length workgroups on a nearly empty chip.  kernel_ms per chain-step
# at the reference's 10^5 chains against ensembles that fill the resident slots exactly, whole-launch workgroups
# (GSSS_SLICE_STEPS=0) against sliced launches (SliceSched, gsss_device.h).
set -e
out=gpurun_out/drain_experiment.jsonl
: > $out
for ss in ${SLICES:-0 64 128}; do
  for wl in curve_d10 curve_d50 curve_d200; do
    for n in ${CHAINS:-98304 100000 200000}; do
      GSSS_SLICE_STEPS=$ss python bench.py --workload $wl --chains $n --steps 5 --warmup 1 --no-cpu-baseline --no-ess --no-configs | sed "s/^{/{\"slice_steps\": $ss, /" >> $out
    done
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/drain_experiment.jsonl"):
    r = json.loads(l)
    c = r["config"]
    print("slice", r["slice_steps"], c["target"], c["chains_per_gpu"], c["kernel"], "kernel_ms %.3f" % r["kernel_ms"],
          "ns/chain-step %.4f" % (r["kernel_ms"] * 1e6 / (c["chains_per_gpu"] * c["transitions_per_step"])))
PY
