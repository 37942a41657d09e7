#!/usr/bin/env python3
"""Static instruction census of one kernel in a gfx950 assembly listing (hipcc -S): per basic block the counts of vector,
scalar, matrix, LDS and memory instructions, with the block's label, line range and the labels it branches to -- to see where a
kernel's instruction stream goes before going to the GPU (tools/: python tools/isa_blocks.py file.s kernel-substring)."""
import re, sys
path, want = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and want in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")) - 1      # (a kernel may hold several s_endpgm)
cls = lambda op: ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
                  "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "smem" if op.startswith("s_load") or op.startswith("s_buffer_load") else
                  "wait" if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")) else "branch" if op.startswith(("s_cbranch", "s_branch")) else
                  "salu" if op.startswith("s_") else "other")
blocks, cur = [], dict(label="entry", first=start + 1, n={}, to=[], ops={})
for i in range(start + 1, end + 1):
    l = lines[i].split(";")[0].strip()
    if not l or l.startswith("."):
        if re.match(r"^\.LBB\d+_\d+:", l):
            cur["last"] = i; blocks.append(cur); cur = dict(label=l[:-1], first=i + 1, n={}, to=[], ops={})
        continue
    op = l.split()[0]
    c = cls(op)
    cur["n"][c] = cur["n"].get(c, 0) + 1
    cur["ops"][op] = cur["ops"].get(op, 0) + 1
    if c == "branch":
        cur["to"].append(l.split()[-1])
cur["last"] = end; blocks.append(cur)
tot = {}
order = {b["label"]: j for j, b in enumerate(blocks)}
print("%-12s %6s %5s %5s %5s %4s %4s %4s %4s  %s" % ("block", "line", "valu", "salu", "mfma", "lds", "vmem", "smem", "wait", "branches (* = backward)"))
for j, b in enumerate(blocks):
    n = b["n"]
    for k, v in n.items():
        tot[k] = tot.get(k, 0) + v
    to = ["%s%s" % (t, "*" if order.get(t, 1 << 30) <= j else "") for t in b["to"]]
    print("%-12s %6d %5d %5d %5d %4d %4d %4d %4d  %s" % (b["label"], b["first"] + 1, n.get("valu", 0), n.get("salu", 0) + n.get("branch", 0), n.get("mfma", 0),
                                                    n.get("lds", 0), n.get("vmem", 0), n.get("smem", 0), n.get("wait", 0), " ".join(to)))
print("static total:", tot)
if len(sys.argv) > 3:       # mnemonic histogram of the blocks whose labels are listed
    hist = {}
    for b in blocks:
        if b["label"] in sys.argv[3:]:
            for k, v in b["ops"].items():
                hist[k] = hist.get(k, 0) + v
    for k, v in sorted(hist.items(), key=lambda kv: -kv[1]):
        print("%5d %s" % (v, k))
