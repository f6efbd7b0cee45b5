export PTGPU_BUILD_DIR=_build_sec
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --preset perlin_spheres --bvh --width 1920 --height 1080 --samples 128 2>&1 | grep -v "^{" | tail -1
export PTGPU_BUILD_DIR=_build_gr
timeout 120 python tools/tree_stats.py perlin_spheres 960 540 8 1 2>&1 | grep -v binary | tail -5 | cut -c1-900
