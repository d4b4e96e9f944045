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
               "csrc_sha256": cfg.get("csrc_sha256"),  # the kernel sources the counters were measured on (bench.py quotes them for no others)
               "launch": {"chains": cfg.get("chains_per_gpu"), "steps": cfg.get("transitions_per_step"), "thin": thin,
                          "mode": cfg.get("mode"), "layout": cfg.get("kept_rows_layout", "components")},
               "source": f"profiles/{tag}_{wl}_summary.md (2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc, separate passes)"}
    # issue side of the same kernel (passes sq1 / sq2 / sq3 of tools/collect_profiles.sh): what bench.py quotes next to
    # roofline_valu so that the delivered-work fraction is never read as a hardware utilisation
    sq = collections.defaultdict(list)
    for sub in ("sq1", "sq2", "sq3"):
        for f in newest(f"{d}/{sub}/**/*_counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if r["Kernel_Name"] == name:
                    sq[r["Counter_Name"]].append(float(r["Counter_Value"]))
    sq = {k: sum(v) / len(v) for k, v in sq.items()}
    chain_steps = (cfg.get("chains_per_gpu") or 0) * (cfg.get("transitions_per_step") or 0)
    if sq.get("GRBM_GUI_ACTIVE") and chain_steps:
        simd_cycles = 1024.0 * sq["GRBM_GUI_ACTIVE"] / 8.0          # 256 CUs x 4 SIMDs x cycles of the launch (GRBM counts per XCD)
        seconds = None                                               # the kernel's average duration in the kernel-trace run
        for f in newest(f"{d}/trace/**/*_kernel_stats.csv"):
            for r in csv.DictReader(open(f)):
                if r["Name"] == name:
                    seconds = float(r["AverageNs"]) * 1e-9
        seconds = seconds or sq["GRBM_GUI_ACTIVE"] / 8.0 / 2.4e9
        out[wl]["issue"] = {
            "valu_insts_per_chain_step": sq.get("SQ_INSTS_VALU", 0.0) / chain_steps,
            "valu_busy": 4.0 * sq.get("SQ_ACTIVE_INST_VALU", 0.0) / simd_cycles,
            "resident_waves_per_simd": 4.0 * sq.get("SQ_WAVE_CYCLES", 0.0) / simd_cycles,
            "lane_activity": sq.get("SQ_THREAD_CYCLES_VALU", 0.0) / 64.0 / max(1.0, sq.get("SQ_INSTS_VALU", 0.0)),
            "fp64_flops_issued_per_chain_step": 64.0 * sq.get("SQ_INSTS_VALU_FLOPS_FP64", 0.0) / chain_steps,
            "fp64_issued_frac": 64.0 * sq.get("SQ_INSTS_VALU_FLOPS_FP64", 0.0) / seconds / 78.6e12,
            "lds_bank_conflict_cycles_per_lds_inst": sq.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, sq.get("SQ_INSTS_LDS", 0.0)),
            "source": f"profiles/{tag}_{wl}_summary.md (SQ_* / GRBM_GUI_ACTIVE, rocprofv3 --pmc, separate passes)"}
    shutil.copy(os.path.join(d, "summary.md"), os.path.join(ROOT, "profiles", f"{tag}_{wl}_summary.md"))
    for f in newest(f"{d}/trace/**/*_kernel_stats.csv"):
        shutil.copy(f, os.path.join(ROOT, "profiles", f"{tag}_{wl}_kernel_stats.csv"))
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
