#!/usr/bin/env python3
"""Build profiles/traffic.json (the HBM bytes per launch bench.py quotes as roofline.traffic) and copy the per-workload
rocprofv3 summaries from gpurun_out/prof_<tag>_<workload>/ into profiles/.

    python tools/pmc_traffic.py r02

HBM bytes per launch of the dominant kernel = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950 (it tallies 128-B requests at 64 B)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(pattern):
    """gpurun merges a call's files into gpurun_out/ beside those of earlier calls: of several runs of one pass, the last"""
    files = glob.glob(pattern, recursive=True)
    return [max(files, key=os.path.getmtime)] if files else []


tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
out = {}
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_*"))):
    wl = os.path.basename(d)[len(f"prof_{tag}_"):]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ("fetch", "write"):
        for f in newest(f"{d}/{sub}/**/*_counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not vals:
        continue
    # dominant kernel = the sampler kernel with the most traffic
    name = max((k for k in vals if "gsss" in k and ("kernel<" in k)), key=lambda k: sum(vals[k].get("WRITE_SIZE", [0])), default=None)
    if name is None:
        continue
    fetch = sum(vals[name]["FETCH_SIZE"]) / len(vals[name]["FETCH_SIZE"])
    write = sum(vals[name]["WRITE_SIZE"]) / len(vals[name]["WRITE_SIZE"])
    line = [ln for ln in open(os.path.join(d, "trace.log")) if ln.startswith('{"metric"')]
    cfg = json.loads(line[-1])["config"] if line else {}
    thin = int(re.search(r"thin=(\d+)", cfg.get("workload", "thin=0")).group(1))
    out[wl] = {"kernel": name.split("(")[0].replace("void ", ""), "fetch_size_kib": fetch, "write_size_kib": write,
               "bytes_per_launch": (2 * fetch + write) * 1024.0,
               "launch": {"chains": cfg.get("chains_per_gpu"), "steps": cfg.get("transitions_per_step"), "thin": thin,
                          "mode": cfg.get("mode")},
               "source": f"profiles/{tag}_{wl}_summary.md (2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc, separate passes)"}
    shutil.copy(os.path.join(d, "summary.md"), os.path.join(ROOT, "profiles", f"{tag}_{wl}_summary.md"))
    for f in newest(f"{d}/trace/**/*_kernel_stats.csv"):
        shutil.copy(f, os.path.join(ROOT, "profiles", f"{tag}_{wl}_kernel_stats.csv"))
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
