#!/usr/bin/env python3
"""Development aid (needs a -DPT_DEVKNOBS library and PTGPU_TIMING=1): renders one shard of config 4 and lets the library print the
timeline of the cooperative hand-over (when each pixel was received / finished, pt_launch.hip)."""
import argparse, importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel))
    mod = importlib.util.module_from_spec(spec); sys.modules[name] = mod; spec.loader.exec_module(mod); return mod
ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=256)
ap.add_argument("--shards", type=int, default=8)
ap.add_argument("--shard", type=int, default=0)
ap.add_argument("--reps", type=int, default=2)
args = ap.parse_args()
import torch
ptgpu = _load("pathtrace_rs_amd_ptgpu", "pathtrace-rs_amd/ptgpu.py")
pthost = _load("pathtrace_rs_amd_pthost", "pathtrace-rs_amd/pthost.py")
W, H, S = 1200, 800, args.samples
hs = pthost.HostScene("random_spheres", W, H, samples=S, device=0)
sc = hs.device_scene()
sc.set_tuning(0, 8192 | int(os.environ.get("PTGPU_VARIANT", "0")))
p = ptgpu.PtParams(W, H, S, 10, 0, 0)
stream = torch.cuda.current_stream().cuda_stream
rc = torch.zeros(1, dtype=torch.int64, device="cuda")
buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
for _ in range(args.reps):
    buf.zero_()
    sc.update_shard_device(p, hs.camera, 0, args.shard, args.shards, buf.data_ptr(), rc.data_ptr(), stream)
    torch.cuda.synchronize()
    print("pass %.3f ms, rays %d, coop %s" % (sc.last_pass_ms(), int(rc.item()), sc.coop_counters()), flush=True)
