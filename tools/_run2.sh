bash tools/gpu_suite.sh 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_final.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_final.json'))
r=d['roofline']
print('value', d['value'], 'ms', d['ms_per_step'], 'build', d['build'])
print('frac', r.get('frac'), 'frac_useful', r.get('frac_useful'), 'stale', r.get('stale_counters'), 'kernel_ms', r['kernel_ms'], 'traffic', r['traffic'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
for k,v in d['baseline_configs'].items(): print(k, v['value'], v['roofline'].get('frac'), v['roofline'].get('frac_useful'))
for k,v in d['other_workloads_same_frame_size'].items(): print(k, v['value'])
PY
