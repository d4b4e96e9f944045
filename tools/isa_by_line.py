"""Static instruction counts of one kernel by source line (llvm-objdump -d -l of a -gline-tables-only build).

    python tools/isa_by_line.py listing.lst 'curvespec_kernelILi4ELi1ELi10ELb0ELb0' [--valu] [--top N]

Attributes every instruction to the file:line objdump printed last before it (inlined frames: the innermost).  Static
counts only: weigh loops by hand.  A first look at where a kernel's instruction stream goes."""
import collections
import re
import sys


def main():
    lst, sym = sys.argv[1], sys.argv[2]
    valu_only = "--valu" in sys.argv
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 60
    inside = False
    cur = "?"
    by_line = collections.Counter()
    by_op = collections.Counter()
    total = 0
    for raw in open(lst):
        line = raw.rstrip("\n")
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            inside = sym in m.group(1)
            continue
        if not inside:
            continue
        if line.startswith("; ") and re.search(r":\d+$", line):
            cur = line[2:].split("/")[-1]
            continue
        ins = line.strip()
        if not ins or ins.startswith(";") or ins.startswith("<"):
            continue
        op = ins.split()[0]
        if not re.match(r"^[a-z_0-9]+$", op):
            continue
        if valu_only and not op.startswith("v_"):
            continue
        total += 1
        by_line[cur] += 1
        by_op[op] += 1
    print("total", total)
    for k, v in by_line.most_common(top):
        print(f"{v:6d}  {k}")
    print("--- opcodes")
    for k, v in by_op.most_common(25):
        print(f"{v:6d}  {k}")


if __name__ == "__main__":
    main()
