run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-pipeline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('   %.1f Mrays/s  %.3f ms/step  kernel %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for rep in 1 2 3; do for b in _build_prev _build; do export PTGPU_BUILD_DIR=$b
echo "$b c3"; run
echo "$b c5"; run --preset perlin_spheres --bvh --width 1920 --height 1080 --samples 128 --steps 4
done; done
for b in _build_prev _build; do export PTGPU_BUILD_DIR=$b
echo "$b aras"; run --preset aras --width 1280 --height 720 --samples 16
echo "$b c4"; run --samples 256 --steps 4
echo "$b -B"; run --bvh
echo "$b random"; run --preset random
echo "$b small"; run --preset small
echo "$b two_perlin"; run --preset two_perlin_spheres
echo "$b smallpt -B"; run --preset smallpt --bvh
done
unset PTGPU_BUILD_DIR
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
