#!/bin/bash
# The general-world kernel's profiled workloads of a round (run through gpurun from the repo root): bash tools/profile_world.sh r04
R=${1:-r04}
bash tools/profile.sh ${R}_world --preset cornell_smoke > /dev/null 2>&1
bash tools/profile.sh ${R}_world_light --preset simple_light > /dev/null 2>&1
for t in ${R}_world ${R}_world_light; do python3 tools/profile_summary.py $t > gpurun_out/prof_$t/summary.txt 2>&1; tail -3 gpurun_out/prof_$t/summary.txt; done
mkdir -p gpurun_out/profiles_$R && cp profiles/${R}_world* gpurun_out/profiles_$R/
for p in simple_light cornell cornell_smoke; do python tools/bq.py --preset $p --steps 10 --warmup 2 --no-extras; python tools/bq.py --preset $p --bvh --steps 10 --warmup 2 --no-extras; done
