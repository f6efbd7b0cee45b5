timeout 1500 python tools/fuzz_worlds.py 3000000 12000 all 2>&1 | tail -1
timeout 500 python tools/fuzz_worlds.py 3100000 2000 noise 2>&1 | tail -1
timeout 500 python tools/fuzz_worlds.py 3200000 2000 graphs 2>&1 | tail -1
timeout 900 python tools/grid_soak.py 110000 6000 2>&1 | tail -1
timeout 900 python tools/pool_soak.py 40000 3000 2>&1 | tail -1
timeout 600 python tools/order_soak.py 5000 150 2>&1 | tail -1
