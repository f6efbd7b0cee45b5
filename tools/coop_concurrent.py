#!/usr/bin/env python3
"""Development aid: N scene handles render N frames concurrently on N streams of ONE device (the frame kernels are persistent grids
that want every CU: their workgroups are only partly resident at any time) and every frame is compared with the same frame rendered
alone. Exercises the cooperative hand-over (csrc/pt_coop.h) under partial residency."""
import argparse, importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel)); mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod; spec.loader.exec_module(mod); return mod
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=8); ap.add_argument("--width", type=int, default=96); ap.add_argument("--height", type=int, default=50)
ap.add_argument("--samples", type=int, default=4); ap.add_argument("--variant", type=int, default=0); ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--shards", action="store_true")
a = ap.parse_args()
import torch
ptgpu = _load("pathtrace_rs_amd_ptgpu", "pathtrace-rs_amd/ptgpu.py"); pthost = _load("pathtrace_rs_amd_pthost", "pathtrace-rs_amd/pthost.py")
W, H, S, N = a.width, a.height, a.samples, a.n
p = ptgpu.PtParams(W, H, S, 10, 0, 0)
scenes = [pthost.HostScene("random_spheres", W, H, samples=S, device=0) for _ in range(N)]
for s in scenes: s.device_scene().set_tuning(0, a.variant)
streams = [torch.cuda.Stream() for _ in range(N)]
rows = [ptgpu.shard_rows(H, r, N) if a.shards else H for r in range(N)]
bufs = [torch.zeros((rows[r], W, 3), dtype=torch.float32, device="cuda") for r in range(N)]
rcs = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in range(N)]
def render(r, frame, stream):
    sc = scenes[r].device_scene()
    if a.shards: sc.update_shard_device(p, scenes[r].camera, frame, r, N, bufs[r].data_ptr(), rcs[r].data_ptr(), stream)
    else: sc.update_device(p, scenes[r].camera, frame + r, bufs[r].data_ptr(), rcs[r].data_ptr(), stream)
bad = 0
for rep in range(a.reps):
    ref = []
    for r in range(N):   # alone, one after the other
        bufs[r].zero_(); torch.cuda.synchronize()
        render(r, rep, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
        ref.append((bufs[r].clone(), int(rcs[r].item())))
    for r in range(N): bufs[r].zero_()
    torch.cuda.synchronize()
    for r in range(N): render(r, rep, streams[r].cuda_stream)
    torch.cuda.synchronize()
    for r in range(N):
        d = int((bufs[r] != ref[r][0]).any(dim=-1).sum().item())
        if d or int(rcs[r].item()) != ref[r][1]:
            bad += 1
            print("rep %d rank %d: %d pixels differ, rays %d vs %d" % (rep, r, d, int(rcs[r].item()), ref[r][1]))
print("variant %d n %d shards %d: %s" % (a.variant, N, a.shards, "OK" if not bad else "%d bad frames" % bad))
