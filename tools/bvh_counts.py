"""SURVEY 8d, BVH mode: node visits / sphere tests per ray of the REFERENCE traversal (oracle: both children of every
node whose box is hit, bvh.rs:37-62, over the caller's random-axis tree) next to the device's internal tree.
Usage: python tools/bvh_counts.py [preset] [W H S]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import oracle_binding as ob  # noqa: E402

pthost = importlib.import_module("pathtrace-rs_amd.pthost")
ptgpu = pthost.ptgpu
preset = sys.argv[1] if len(sys.argv) > 1 else "perlin_spheres"
W, H, S = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (480, 270, 4)
L = ob.lib(ob.build_native(os.path.join(ROOT, "gpurun_out", "ora_native")))
c = (C.c_uint64 * 2)()
L.ora_bvh_counters(c, 1)
sc = ob.OracleScene(preset, W, H, use_bvh=True, library=L)
px = np.arange(0, W * H, 7, dtype=np.uint32)
_, rays = sc.update(S, pixels=px)
L.ora_bvh_counters(c, 1)
print("%s %dx%d %d spp, BVH world" % (preset, W, H, S))
print("  reference traversal (oracle, %d rays): %.1f BVHNode::ray_hit calls, %.1f leaf tests per ray; tree: %d nodes" % (
    rays, c[0] / rays, c[1] / rays, L.ora_scene_num_bvh_nodes(sc.h)))
hs = pthost.HostScene(preset, W, H, samples=S, use_bvh=True, device=0)
dev = hs.device_scene()
dev.set_tuning(0, 8 | 256)   # verify mode on the tree kernel
dev.traversal_counters(reset=True)
out = np.zeros((H, W, 3), np.float32)
rays = dev.update(ptgpu.PtParams(W, H, S, 10, 0, 1), hs.camera, 0, out)
t = dev.traversal_counters(reset=True)
print("  device internal tree (%d rays): %.1f node fetches (each tests two child boxes), %.1f exact sphere tests per ray" % (
    rays, t["nodes"] / rays, t["sphere_tests"] / rays))
