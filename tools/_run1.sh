run() { python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras --no-pipeline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('   %.1f Mrays/s  %.3f ms/step  kernel %.3f ms  lds %d block %d' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['lds_bytes'], d['config']['block']))"; }
for rep in 1 2; do
for b in _build_r5 _build_dev; do
export PTGPU_BUILD_DIR=$b
echo "$b c3"; run
echo "$b aras"; run --preset aras --width 1280 --height 720 --samples 16
done
done
for b in _build_r5 _build_dev; do
export PTGPU_BUILD_DIR=$b
echo $b; python tools/shard_times.py --counts 2,4,8 --reps 4 2>&1 | grep -v amdgpu.ids
done
