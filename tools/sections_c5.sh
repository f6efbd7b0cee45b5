cp pathtrace-rs_amd/_build/libptgpu.so /tmp/cur.so
cp pathtrace-rs_amd/_build_dev/libptgpu.so pathtrace-rs_amd/_build/libptgpu.so
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --preset perlin_spheres --bvh --width 1920 --height 1080 --samples 128 2>&1 | grep -v "^{" | tail -3
cp /tmp/cur.so pathtrace-rs_amd/_build/libptgpu.so
