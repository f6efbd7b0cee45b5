"""Traversal counters of the tree kernels (verify mode): nodes fetched and exact sphere tests per ray for the 4-wide
tree (default) and the binary tree (variant bit 2048). Usage: python tools/tree_stats.py [preset] [W H S] [bvh 0/1]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pthost = importlib.import_module("pathtrace-rs_amd.pthost")
ptgpu = pthost.ptgpu
preset = sys.argv[1] if len(sys.argv) > 1 else "perlin_spheres"
W, H, S = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (480, 270, 4)
bvh = int(sys.argv[5]) if len(sys.argv) > 5 else 1
hs = pthost.HostScene(preset, W, H, samples=S, use_bvh=bool(bvh), device=0)
dev = hs.device_scene()
for name, var in (("4-wide", 8 | 256), ("binary", 8 | 256 | 2048)):
    dev.set_tuning(0, var)
    dev.traversal_counters(reset=True)
    out = np.zeros((H, W, 3), np.float32)
    rays = dev.update(ptgpu.PtParams(W, H, S, 10, 0, bvh), hs.camera, 0, out)
    t = dev.traversal_counters(reset=True)
    print("%s %dx%d %dspp %s tree: %d rays, %.2f node fetches, %.2f exact sphere tests per ray" % (
        preset, W, H, S, name, rays, t["nodes"] / rays, t["sphere_tests"] / rays))
