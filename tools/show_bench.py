"""Prints the lines of a bench.py log in short form: python tools/show_bench.py gpurun_out/bench.log"""
import json
import sys

for l in open(sys.argv[1]):
    if not l.startswith("{"):
        continue
    j = json.loads(l)
    rv = j.get("roofline_valu", {})
    print(f"{j['config']['target']:22s} {j['value']:.4e} {j['unit']}  {j['kernel_ms']:.2f} ms  {j['config']['kernel']}  valu {rv.get('frac', 0):.3f}")
    for c in j.get("configs", []):
        rv = c.get("roofline_valu", {})
        print(f"{c['workload'].split(':')[0]:22s} {c['value']:.4e}  {c['kernel_ms']:.2f} ms  {c['kernel']}  valu {rv.get('frac', 0):.3f}")
