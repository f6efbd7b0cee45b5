#!/usr/bin/env python3
"""Instruction-class histogram of one kernel in a hipcc -S listing (development aid).
usage: isa_hist.py listing.s <mangled-name-substring> [first_line last_line]"""
import collections, re, sys
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and l.rstrip().split(":")[0].endswith(key.split()[-1]) or (l.startswith("_ZN") and key in l.split(":")[0]))
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (start, end)
h = collections.Counter()
for l in lines[lo:hi]:
    m = re.match(r"\s+([a-z_0-9]+)\s", l)
    if not m or l.strip().startswith((";", ".")):
        continue
    op = m.group(1)
    if op.startswith("v_mfma"): c = "mfma"
    elif op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): c = "v_lane"
    elif op.startswith("v_") and "f64" in op: c = "valu_f64"
    elif op.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")): c = "valu_trans"
    elif op.startswith("v_cmp"): c = "valu_cmp"
    elif op.startswith("v_"): c = "valu"
    elif op.startswith("s_waitcnt"): c = "s_waitcnt"
    elif op.startswith(("s_cbranch", "s_branch")): c = "s_branch"
    elif op.startswith("s_"): c = "salu"
    elif op.startswith("ds_"): c = "lds"
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c = "vmem"
    else: c = "other"
    h[c] += 1; h["_" + op] += 1
print("lines %d..%d" % (lo, hi))
for k, v in sorted(h.items(), key=lambda kv: -kv[1]):
    if not k.startswith("_"): print("%-12s %6d" % (k, v))
top = [(k[1:], v) for k, v in h.items() if k.startswith("_")]
print("top ops:", ", ".join("%s %d" % kv for kv in sorted(top, key=lambda kv: -kv[1])[:28]))
