#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's N=1 workload on the GPU box (run through gpurun):
#   1. --kernel-trace --stats (CSV)                 -> per-kernel durations
#   2..4. separate --pmc passes (never mixed with other trace domains): SQ issue/wait, LDS, HBM traffic
# Raw output goes to gpurun_out/prof_$TAG/; tools/profile_summary.py condenses it into profiles/.
set -u
TAG=${1:-r01}
shift || true
EXTRA="$*"   # optional bench.py arguments (e.g. --preset perlin_spheres --bvh ...): profiles of the other kernels
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras $EXTRA"
# (the timing pass runs as many warm-up frames as the default bench: the first frames of a process run at ramping clocks, 6.9 ... 6.2 ms on config 3)
BENCH_T="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $BENCH_T > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  --kernel-trace --output-format csv -d $OUT/pmc_sq -o p -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA \
  --kernel-trace --output-format csv -d $OUT/pmc_lds -o p -- $BENCH > $OUT/pmc_lds.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- $BENCH > $OUT/pmc_write.log 2>&1
grep -h '^{' $OUT/stats.log | tail -1 > $OUT/bench_line.json
grep -h '^{' $OUT/pmc_sq.log | tail -1 > $OUT/bench_line_pmc.json
ls -R $OUT | head -40
