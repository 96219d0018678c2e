#!/usr/bin/env python3
"""Static instruction mix of one kernel of a hipcc --save-temps .s file, per basic block:
   tools/isa_histogram.py file.s '<substring of the mangled kernel name>'
Columns: MFMA, other VALU, SALU, LDS, VMEM, waitcnt/barrier.  Backward branches are listed so loop bodies can be weighted."""
import re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and l.rstrip().endswith(l.split(":")[0].strip()) or (l.startswith("_Z") and key in l and ": " in l and "@" in l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur = [], {"label": "entry", "n": {}, "line": start}
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop"): return "wait"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    return "other"
labels = {}
for i in range(start + 1, end):
    l = lines[i].strip()
    if not l or l.startswith(";") or l.startswith("."):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur); cur = {"label": m.group(1), "n": {}, "line": i, "br": []}
            labels[m.group(1)] = len(blocks)
        continue
    op = l.split()[0]
    c = cls(op)
    cur["n"][c] = cur["n"].get(c, 0) + 1
    if op.startswith("s_cbranch") or op == "s_branch":
        cur.setdefault("br", []).append(l.split()[1])
blocks.append(cur)
tot = {}
print("%-12s %6s %6s %6s %6s %6s %6s   branches" % ("block", "mfma", "valu", "salu", "lds", "vmem", "wait"))
for bi, b in enumerate(blocks):
    n = b["n"]
    for k, v in n.items(): tot[k] = tot.get(k, 0) + v
    back = [t for t in b.get("br", []) if t in labels and labels[t] <= bi]
    if sum(n.values()) >= 8 or back:
        print("%-12s %6d %6d %6d %6d %6d %6d   %s" % (b["label"], n.get("mfma", 0), n.get("valu", 0), n.get("salu", 0), n.get("lds", 0), n.get("vmem", 0), n.get("wait", 0),
              " ".join(("<-" + t) if t in back else t for t in b.get("br", []))))
print("total", tot)
