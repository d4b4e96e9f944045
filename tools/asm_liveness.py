"""Vector-register liveness of one kernel in AMDGPU assembly (hipcc -S --cuda-device-only): where a kernel's register
budget goes.

    python tools/asm_liveness.py file.s [kernel-name-substring] [--top N] [--at LINE]

Builds the control-flow graph from the labels and branches, solves backward liveness for VGPRs (AGPRs are ignored) and
prints the lines of highest pressure and, with --at, the registers live at a line together with where each was last written
before it (a hint to what it holds)."""
import re
import sys

REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(tok):
    out = []
    for m in REG.finditer(tok):
        if m.group(1):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


def parse(line):
    """(defs, uses, partial) of one instruction line"""
    l = line.split(";")[0].strip()
    if not l or l.startswith(".") or l.endswith(":"):
        return None
    parts = l.split(None, 1)
    op = parts[0]
    if len(parts) < 2:
        return (op, [], [], False)
    toks = [t.strip() for t in parts[1].split(",")]
    n_dst = 1
    if op.startswith(("ds_write", "global_store", "buffer_store", "flat_store", "scratch_store", "s_", "ds_bpermute_b32x")):
        n_dst = 0
    if op.startswith("v_cmp") or op.startswith(("v_readlane", "v_readfirstlane")):
        n_dst = 1  # destination is scalar / vcc: no vector def (regs() finds none in it)
        if op.startswith("v_cmpx"):
            n_dst = 0
    if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale")):
        n_dst = 2
    dst, src = [], []
    for i, t in enumerate(toks):
        (dst if i < n_dst else src).extend(regs(t))
    # partial writes keep the old value alive: lane writes, DPP/SDWA with bound_ctrl off, and (conservatively) nothing else
    partial = op.startswith("v_writelane") or "dpp" in l or "sdwa" in l or "row_" in l or "quad_perm" in l
    if op.startswith("v_accvgpr_write"):
        dst = []
    return (op, dst, src, partial)


def main():
    args = [a for a in sys.argv[1:]]
    path = args[0]
    top, at, name = 8, None, None
    i = 1
    while i < len(args):
        if args[i] == "--top":
            top = int(args[i + 1]); i += 2
        elif args[i] == "--at":
            at = int(args[i + 1]); i += 2
        else:
            name = args[i]; i += 1
    lines = open(path).read().split("\n")
    # kernel extent: from its label to s_endpgm-terminated end (.Lfunc_end)
    start, end = 0, len(lines)
    if name:
        for n, l in enumerate(lines):
            l = l.split(";")[0].strip()
            if l.endswith(":") and name in l and not l.startswith("."):
                start = n
                break
        for n in range(start, len(lines)):
            if lines[n].startswith(".Lfunc_end"):
                end = n
                break
    # basic blocks
    label_at = {}
    for n in range(start, end):
        l = lines[n].split(";")[0].strip()
        if l.endswith(":"):
            label_at[l[:-1]] = n
    ins = {n: parse(lines[n]) for n in range(start, end)}
    idx = [n for n in range(start, end) if ins[n] is not None]
    nxt = {}
    for k, n in enumerate(idx):
        op = ins[n][0]
        succ = []
        fall = idx[k + 1] if k + 1 < len(idx) else None
        if op == "s_endpgm":
            succ = []
        elif op == "s_branch":
            tgt = lines[n].split()[1]
            succ = [first_ins(label_at, idx, tgt)]
        elif op.startswith("s_cbranch"):
            tgt = lines[n].split()[1]
            succ = [first_ins(label_at, idx, tgt), fall]
        elif op.startswith("s_setpc") or op.startswith("s_swappc"):
            succ = [fall]
        else:
            succ = [fall]
        nxt[n] = [s for s in succ if s is not None]
    live_in = {n: frozenset() for n in idx}
    changed = True
    rounds = 0
    while changed and rounds < 200:
        changed = False
        rounds += 1
        for n in reversed(idx):
            op, dst, src, partial = ins[n]
            out = set()
            for s in nxt[n]:
                out |= live_in[s]
            if not partial:
                out -= set(dst)
            out |= set(src)
            if partial:
                out |= set(dst)
            f = frozenset(out)
            if f != live_in[n]:
                live_in[n] = f
                changed = True
    press = sorted(((len(live_in[n]), n + 1) for n in idx), reverse=True)
    print("highest pressure (live VGPRs, line):", press[:top])
    if at:
        n = at - 1
        live = sorted(live_in[n])
        print(f"{len(live)} live before line {at}:")
        # last textual definition before the line
        for r in live:
            where = None
            for m in range(n - 1, start, -1):
                p = ins.get(m)
                if p and r in p[1]:
                    where = m
                    break
            print(f"  v{r}: written at {where + 1 if where is not None else '?'}: {lines[where].strip()[:90] if where is not None else ''}")


def first_ins(label_at, idx, label):
    import bisect
    n = label_at.get(label)
    if n is None:
        return None
    k = bisect.bisect_left(idx, n)
    return idx[k] if k < len(idx) else None


if __name__ == "__main__":
    main()
