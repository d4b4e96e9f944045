"""Tabulates hipcc's -Rpass-analysis=kernel-resource-usage remarks (stdin or file): registers, spills, occupancy per kernel.

    hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip -o /dev/null 2>&1 | python tools/resource_usage.py [filter]
"""
import re
import subprocess
import sys


def main():
    args = [a for a in sys.argv[1:]]
    path = args[0] if args and args[0].endswith(".txt") else None
    flt = [a for a in args if a != path]
    txt = open(path).read() if path else sys.stdin.read()
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    names = [b.split("\n")[0].strip().split(" [")[0].rstrip("]") for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    for b, dn in zip(blocks, dem):
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return m.group(1) if m else "?"
        dn = dn.replace("gsss::", "").replace("void ", "")
        dn = re.sub(r"\(TargetBlock, RunBlock\)", "", dn)
        if flt and not all(f in dn for f in flt):
            continue
        print(f"{dn[:100]:100s} V{g('VGPRs'):>4} A{g('AGPRs'):>4} S{g('SGPRs'):>4} spV{g('VGPR Spill'):>3} spS{g('SGPR Spill'):>3} "
              f"occ{g('Occupancy .waves/SIMD.'):>2} lds{g('LDS Size .bytes/block.'):>6} scr{g('ScratchSize .bytes/lane.'):>4}")


if __name__ == "__main__":
    main()
