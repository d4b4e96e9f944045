#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            if "gsss" not in k:
                continue
            print(k)
            for c, v in sorted(cs.items()):
                print(f"   {c:28s} {sum(v)/len(v):.6g}  (n={len(v)})")
