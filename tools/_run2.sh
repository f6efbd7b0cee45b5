timeout 900 python tools/grid_soak.py 100000 4000 2>&1 | tail -4
timeout 300 python tools/grid_soak.py 1 600 2>&1 | tail -2
timeout 600 python -m pytest tests -m gpu -x -q -k "grid or dense or config5 or golden" 2>&1 | tail -2
