#!/bin/bash
# One PMC pass (default WRITE_SIZE) of a bench.py workload; prints the per-launch average of the frame kernel.
# (a counter set the hardware cannot collect aborts rocprofv3 and then hangs in its signal handler: hence the timeout)
# usage: bash tools/pmc_one.sh TAG "COUNTERS" [bench.py args]
TAG=$1; CNT=$2; shift 2
OUT=gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 5 240 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > $OUT.log 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, collections, re
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[0]
agg, disp = collections.defaultdict(float), collections.defaultdict(set)
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    m = re.search(r"pt_(?:trace|world)_kernel<([^>]*)>", n)
    if not m: continue
    fl = [x.strip() for x in m.group(1).split(",")]
    if "pt_trace_kernel" in n and len(fl) >= 5 and fl[4] == "true": continue
    rows.append(r)
first = min((int(r["Dispatch_Id"]) for r in rows), default=None)   # pt_scene_prepare's throw-away frame
if len({r["Dispatch_Id"] for r in rows}) > 1: rows = [r for r in rows if int(r["Dispatch_Id"]) != first]
for r in rows:
    agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp[r["Counter_Name"]].add(r["Dispatch_Id"])
print(sys.argv[2], {k: v / max(1, len(disp[k])) for k, v in agg.items()})
PY
