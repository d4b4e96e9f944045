"""Rough register-pressure census of a loop in AMDGPU assembly (hipcc -S --cuda-device-only).

    python tools/asm_liveness.py file.s first_line last_line

Lists the vector registers that are only READ between the two lines (loop invariants the compiler hoisted and keeps
live for the whole loop) and those written there, so that one sees what a kernel's register budget is spent on."""
import re
import sys


def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


def main():
    path, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    lines = open(path).read().split("\n")
    written, read = set(), set()
    first_read = {}
    for ln in range(lo - 1, hi):
        l = lines[ln].split(";")[0].strip()
        if not l or l.startswith(".") or l.endswith(":"):
            continue
        parts = l.split(None, 1)
        if len(parts) < 2:
            continue
        op, args = parts
        toks = [t.strip() for t in args.split(",")]
        # destination = first operand for VALU / loads; stores and compares have none
        n_dst = 1
        if op.startswith(("ds_write", "global_store", "buffer_store", "flat_store", "scratch_store", "v_cmp", "s_", "v_writelane")):
            n_dst = 0 if not op.startswith("v_writelane") else 1
        if op.startswith("v_cmp") and toks and toks[0].startswith(("s[", "vcc")):
            n_dst = 1
        if op.startswith("v_mad_u64_u32") or op.startswith("v_div_scale"):
            n_dst = 2
        for i, t in enumerate(toks):
            rs = regs(t)
            if i < n_dst:
                written.update(rs)
            else:
                for r in rs:
                    read.add(r)
                    first_read.setdefault(r, ln + 1)
    inv = sorted(read - written)
    print(f"{len(inv)} registers only read in [{lo}, {hi}] (loop invariants):")
    print(" ".join(f"v{r}@{first_read[r]}" for r in inv))
    print(f"{len(written)} registers written in the range")


if __name__ == "__main__":
    main()
