#!/bin/bash
export PTGPU_BUILD_DIR=_build_dev
out=gpurun_out/coop_timeline.log
: > $out
for a in "--shards 8" "--shards 4" "--shards 1 --samples 64"; do
  echo "== $a" >> $out
  PTGPU_TIMING=1 timeout 300 python tools/coop_timeline.py $a >> $out 2>&1
done
cat $out
