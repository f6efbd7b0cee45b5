run() { python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-extras --no-pipeline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('   %.1f Mrays/s  %.3f ms/step  kernel %.3f ms rays %d' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['rays_per_step']))"; }
for rep in 1 2 3; do
for b in _build _build_cs4 _build_cs8 _build_cs16; do
export PTGPU_BUILD_DIR=$b
echo "$b c3"; run
done; done
export PTGPU_BUILD_DIR=_build_cs8
python -m pytest tests -m gpu -x -q -k "full_frames or exact_parity" 2>&1 | tail -2
