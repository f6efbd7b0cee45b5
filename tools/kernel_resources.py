#!/usr/bin/env python3
"""Register / spill / scratch table of every kernel of a hipcc -Rpass-analysis=kernel-resource-usage log
(development aid and the source of profiles/rNN_kernel_resources.txt).
usage: hipcc ... -Rpass-analysis=kernel-resource-usage 2> log; python tools/kernel_resources.py log"""
import re
import subprocess
import sys

t = open(sys.argv[1]).read()
rows = []
for b in re.split(r"remark: Function Name: ", t)[1:]:
    name = b.split()[0]
    g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
    rows.append((name, g("VGPRs"), g(r"ScratchSize \[bytes/lane\]"), g("SGPRs Spill"), g("VGPRs Spill"), g(r"Occupancy \[waves/SIMD\]")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
print("%-100s %5s %8s %11s %11s %4s" % ("kernel", "VGPRs", "scratch", "SGPR spills", "VGPR spills", "occ"))
for r, d in zip(rows, names):
    d = d.replace("ptdev::", "").replace("(ptdev::KArgs)", "").replace("(ptdev::WArgs)", "").replace("void ", "")
    print("%-100s %5d %8d %11d %11d %4d" % (d[:100], r[1], r[2], r[3], r[4], r[5]))
