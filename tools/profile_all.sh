#!/bin/bash
# The six profiled workloads of a round (run through gpurun from the repo root): bash tools/profile_all.sh r03
R=${1:-r03}
bash tools/profile.sh $R > /dev/null 2>&1
bash tools/profile.sh ${R}_c2 --preset aras --width 1280 --height 720 --samples 16 > /dev/null 2>&1
bash tools/profile.sh ${R}_c4 --samples 256 > /dev/null 2>&1
bash tools/profile.sh ${R}_c5 --preset perlin_spheres --bvh --width 1920 --height 1080 --samples 128 > /dev/null 2>&1
bash tools/profile.sh ${R}_random --preset random > /dev/null 2>&1
bash tools/profile.sh ${R}_world --preset cornell_smoke > /dev/null 2>&1
for t in $R ${R}_c2 ${R}_c4 ${R}_c5 ${R}_random ${R}_world; do python3 tools/profile_summary.py $t > gpurun_out/prof_$t/summary.txt 2>&1; tail -3 gpurun_out/prof_$t/summary.txt; done
mkdir -p gpurun_out/profiles_$R && cp profiles/${R}* gpurun_out/profiles_$R/
