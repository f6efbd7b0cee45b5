run() { python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --no-pipeline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('   %.1f Mrays/s  %.3f ms/step  kernel %.3f ms rays %d' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['rays_per_step']))"; }
for rep in 1 2; do for b in _build _build_b5 _build_b6; do export PTGPU_BUILD_DIR=$b
echo "$b c5"; run --preset perlin_spheres --bvh --width 1920 --height 1080 --samples 128
done; done
for b in _build _build_b5 _build_b6; do export PTGPU_BUILD_DIR=$b
echo "$b two_perlin"; run --preset two_perlin_spheres --steps 10
done
