#!/bin/bash
# Development aid: the cooperative hand-over (pt_coop.h) on and off (tuning bit 65536) on the latency-bound workloads and the headline.
# Every command under its own timeout: a protocol bug would hang the kernel. Needs a -DPT_DEVKNOBS build for the PTGPU_COOP_* knobs.
out=gpurun_out/coop_check.log
: > $out
run() { echo "== $*" >> $out; timeout 300 "$@" >> $out 2>&1; echo "rc=$?" >> $out; }
run python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "exact_parity or progressive_frames_blend or random_sphere_worlds"
export PTGPU_VARIANT=65536
echo "---- coop off" >> $out
run python tools/bq.py --steps 10 --warmup 2 --no-extras
run python tools/bq.py --width 1200 --height 100 --samples 256 --steps 4 --warmup 1 --no-extras
unset PTGPU_VARIANT
for cfg in ${COOP_CFGS:-"4:3" "0:1000" "8:3" "4:2" "4:5" "2:3"}; do
  export PTGPU_COOP_LIVE=${cfg%%:*} PTGPU_COOP_STREAK=${cfg##*:}
  echo "---- coop live<=$PTGPU_COOP_LIVE streak>=$PTGPU_COOP_STREAK" >> $out
  run python tools/bq.py --steps 10 --warmup 2 --no-extras
  run python tools/bq.py --width 8 --height 8 --samples 4096 --steps 2 --warmup 1 --no-extras
  run python tools/bq.py --width 1200 --height 100 --samples 256 --steps 4 --warmup 1 --no-extras
done
export PTGPU_COOP_LIVE=4 PTGPU_COOP_STREAK=3
python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-pipeline 2>&1 | grep "ptgpu coop" | tail -1 >> $out
python bench.py --width 8 --height 8 --samples 4096 --steps 1 --warmup 1 --no-extras --no-cpu-baseline --no-pipeline 2>&1 | grep "ptgpu coop" | tail -1 >> $out
grep -v "^rc=0" $out | tail -80
