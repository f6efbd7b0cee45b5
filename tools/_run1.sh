export PTGPU_BUILD_DIR=_build_dev
for est in 24 200 400 600 800 1200; do
echo "== min_est $est"
PTGPU_COOP_EST=$est python tools/shard_times.py --counts 8 --reps 3 2>&1 | grep -v amdgpu.ids | head -1
done
