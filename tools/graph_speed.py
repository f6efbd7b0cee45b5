#!/usr/bin/env python3
"""Development aid: what the interpreted walk of a scene graph (csrc/pt_graph.h) costs against the flattened list form, on seeded graphs of
the test generator (tests/test_gpu_parity.py _random_graph_world) at 600x400x16."""
import importlib.util, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import oracle_binding as ob
from conftest import load_ptgpu
spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
tgp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tgp)
ptgpu = load_ptgpu()
W, H, S = 600, 400, 16
for seed in (901, 904, 905, 907):
    for wild in (False, True):
        g = tgp._random_graph_world(ob, seed, W, H, n_top=6, max_depth=4, wild=wild)
        materials = [(int(r[0]), r[1:4], r[4], int(r[5])) for r in g["materials"]]
        textures = [(int(r[0]), r[1:4], int(r[4]), int(r[5]), r[6]) for r in g["textures"]]
        sc = ptgpu.Scene(ptgpu.WorldDesc(g["hitables"], g["transforms"], materials, textures, sky=g["sky"], nodes=g["nodes"], node_children=g["node_children"],
                                         root_node=g["root_node"], bvh_nodes=(g["bvh_minmax"], g["bvh_children"]) if len(g["bvh_minmax"]) else None), 0)
        out = np.zeros((H, W, 3), np.float32)
        p, cam = ptgpu.PtParams(W, H, S, 10, 0, 0), ptgpu.PtCamera.from_floats(g["camera"])
        rays = sc.update(p, cam, 0, out)
        t0 = time.time()
        for f in range(3):
            rays = sc.update(p, cam, 0, out)
        ms = sc.last_pass_ms()
        print("seed %d %-11s %3d nodes  %s  %.2f ms  %.0f Mrays/s" % (seed, "interpreted" if sc.last_kernel_choice()["world_graph"] else "flattened", len(g["nodes"]),
                                                                      sc.last_kernel_choice()["name"], ms, rays / ms / 1e3))
        sc.close()
