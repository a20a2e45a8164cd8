#!/usr/bin/env python
"""VGPR liveness over one kernel of a hipcc -save-temps .s file: where is the register-pressure peak?

    python tools/asm_liveness.py <file.s> <kernel-name-substring> [--top 12]

Builds basic blocks from labels / branches, runs backward liveness on v-registers (vN, v[a:b]) and prints the
instructions around the highest live counts with the nearest preceding source-line marker.  Operand roles follow the
AMDGPU convention "first operand(s) = destination" with the exceptions listed in DEST0 (stores, atomics without return).
"""
import re
import sys
from collections import defaultdict

NO_DEST = ("global_store", "scratch_store", "ds_write", "buffer_store", "global_atomic", "s_", "v_cmp", "v_cmpx", "ds_add",
           "flat_store", "v_writelane")  # (v_writelane reads+writes its dst; treated below)


def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def main():
    path, pat = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 12
    text = open(path).read()
    m = re.search(r"^(\S*%s\S*):" % re.escape(pat), text, re.M)
    if not m:
        sys.exit("kernel not found")
    start = m.end()
    end = text.index(".Lfunc_end", start)
    lines = text[start:end].split("\n")
    ins, labels, loc = [], {}, []
    cur_loc = ""
    for l in lines:
        t = l.strip()
        if not t:
            continue
        if t.startswith(".loc"):
            cur_loc = t
            continue
        if re.match(r"^\.?[A-Za-z_0-9$.]+:", t):
            labels[t.split(":")[0]] = len(ins)
            continue
        if t.startswith(".") or t.startswith(";"):
            if "hip:" in t or ".hip" in t:
                cur_loc = t
            continue
        ins.append(t.split(";")[0].strip())
        loc.append(cur_loc)
    n = len(ins)
    use, dfn, succ = [set() for _ in range(n)], [set() for _ in range(n)], [[] for _ in range(n)]
    for i, t in enumerate(ins):
        op = t.split()[0]
        ops = t[len(op):].split(",")
        nodest = op.startswith(NO_DEST) and "atomic" not in op or (op.startswith("global_atomic") and " sc0" not in t) \
            or op.startswith(("ds_write", "global_store", "scratch_store", "s_"))
        if op.startswith("v_cmp") and not op.endswith("_e64"):
            nodest = True
        first = True
        for o in ops:
            r = regs(o)
            if first and not nodest and r:
                dfn[i].update(r)
                if op.startswith(("v_fmac", "v_mac", "v_mfma", "v_writelane", "v_accvgpr")) or "dpp" in t or "mfma" in op:
                    pass
            else:
                use[i].update(r)
            first = False
        if op.startswith(("v_fmac", "v_mac", "v_writelane", "v_pk_fmac")) or "dpp" in t:
            use[i].update(dfn[i])  # read-modify-write destinations
        if "mfma" in op:  # D, A, B, C: C is the last operand and is read
            pass
        if op.startswith("s_branch"):
            succ[i] = [labels.get(ops[0].strip(), n)]
        elif op.startswith("s_cbranch"):
            succ[i] = [labels.get(ops[0].strip(), n), i + 1]
        elif op.startswith("s_endpgm"):
            succ[i] = []
        else:
            succ[i] = [i + 1]
    live_in = [set() for _ in range(n + 1)]
    changed = True
    while changed:
        changed = False
        for i in range(n - 1, -1, -1):
            out = set()
            for s_ in succ[i]:
                if s_ < n:
                    out |= live_in[s_]
            new = use[i] | (out - dfn[i])
            if new != live_in[i]:
                live_in[i] = new
                changed = True
    counts = [len(x) for x in live_in[:n]]
    order = sorted(range(n), key=lambda i: -counts[i])
    print("instructions %d, max live VGPRs %d" % (n, counts[order[0]] if n else 0))
    seen = []
    for i in order:
        if all(abs(i - j) > 40 for j in seen):
            seen.append(i)
            print("--- live %d at #%d  %s   [%s]" % (counts[i], i, ins[i][:70], loc[i][:60]))
        if len(seen) >= top:
            break
    # profile: live count every ~2% of the function
    step = max(1, n // 60)
    print("profile:", " ".join(str(counts[i]) for i in range(0, n, step)))


if __name__ == "__main__":
    main()
