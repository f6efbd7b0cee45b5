bash tools/gpu_suite.sh
timeout 600 python tools/fuzz_worlds.py 2100000 4000 all 2>&1 | tail -2
timeout 300 python tools/fuzz_worlds.py 2200000 800 noise 2>&1 | tail -1
timeout 300 python tools/fuzz_worlds.py 2300000 800 graphs 2>&1 | tail -1
timeout 500 python tools/grid_soak.py 50000 2000 2>&1 | tail -1
timeout 500 python tools/pool_soak.py 30000 1500 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py 2>/dev/null | tail -1
