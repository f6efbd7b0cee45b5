#!/bin/bash
# Calibrates rocprof's VALUUtilization (bench.py `valu_lane_utilisation`) with a kernel whose active lane count is known, and measures what
# a lone wave pays per instruction. Run through gpurun from the repo root; writes gpurun_out/valu_lane_calibration.txt.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/valu_lanes.hip -o tools/valu_lanes || exit 1
out=gpurun_out/valu_lane_calibration.txt
mkdir -p gpurun_out; : > $out
for L in 64 32 16; do
  rm -rf gpurun_out/cal_$L
  timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d gpurun_out/cal_$L -o p -- ./tools/valu_lanes cal $L > gpurun_out/cal_$L.log 2>&1
  python3 - gpurun_out/cal_$L $L >> $out <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[0]
v = {}
for r in csv.DictReader(open(f)):
    if "cal" in r["Kernel_Name"]: v[r["Counter_Name"]] = v.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print("%s of 64 lanes active: SQ_INSTS_VALU %.4g  SQ_ACTIVE_INST_VALU %.4g  SQ_THREAD_CYCLES_VALU %.4g  ->  SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64) = %.4f  (expected %.4f)" % (
    sys.argv[2], v.get("SQ_INSTS_VALU", 0), v.get("SQ_ACTIVE_INST_VALU", 0), v.get("SQ_THREAD_CYCLES_VALU", 0),
    v.get("SQ_THREAD_CYCLES_VALU", 0) / max(v.get("SQ_ACTIVE_INST_VALU", 0) * 64.0, 1.0), int(sys.argv[2]) / 64.0))
PY
done
./tools/valu_lanes lone >> $out 2>&1
cat $out
