#!/bin/bash
# Development aid: what each part of the cooperative mode costs when NOTHING is handed over (PTGPU_COOP_DBG: 1 no workers, 2 no probes, 4 no counting)
export PTGPU_BUILD_DIR=_build_dev
out=gpurun_out/coop_dbg.log
: > $out
export PTGPU_COOP_LIVE=0 PTGPU_COOP_STREAK=100000
for args in "--samples 16 --steps 20 --warmup 3"; do
  echo "off: $(PTGPU_VARIANT=65536 timeout 300 python tools/bq.py $args --no-extras)" >> $out
  for dbg in 0 1 2 3 5; do
    echo "dbg=$dbg: $(PTGPU_COOP_DBG=$dbg timeout 300 python tools/bq.py $args --no-extras)" >> $out
  done
done
cat $out
