#!/usr/bin/env python3
"""Brief bench wrapper: runs bench.py with the given args and prints value / ms / launch geometry."""
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-pipeline"] + sys.argv[1:], capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    print("FAILED", out.stdout[-500:], out.stderr[-1500:]); sys.exit(1)
d = json.loads(line[-1])
print("%-40s %9.1f %s  %8.2f ms/step kernel %.2f ms grid %s lds %s rays %d" % (" ".join(sys.argv[1:]), d["value"], d["unit"], d["ms_per_step"],
      d["roofline"]["kernel_ms"], d["config"]["grid"], d["config"]["lds_bytes"], d["config"]["rays_per_step"]))
