export PTGPU_BUILD_DIR=_build
export PTGPU_BUILD_DIR=_build_dev
echo "== config 3" > gpurun_out/sections_r05.log
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep -v "^{" | tail -4 >> gpurun_out/sections_r05.log
echo "== config 5" >> gpurun_out/sections_r05.log
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --preset perlin_spheres --bvh --width 1920 --height 1080 --samples 128 2>&1 | grep -v "^{" | tail -4 >> gpurun_out/sections_r05.log
export PTGPU_BUILD_DIR=_build
cat gpurun_out/sections_r05.log
