export PTGPU_BUILD_DIR=_build
export PTGPU_BUILD_DIR=_build_dev
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --preset perlin_spheres --bvh --width 1920 --height 1080 --samples 128 2>&1 | grep -v "^{" | tail -3
export PTGPU_BUILD_DIR=_build
