#!/usr/bin/env python3
"""Development aid: a preset's frame time at 1200x800x64 for several workgroups-per-CU settings (pt_scene_set_tuning's first argument)."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel)); mod = importlib.util.module_from_spec(spec); sys.modules[name] = mod; spec.loader.exec_module(mod); return mod
import torch
ptgpu = _load("pathtrace_rs_amd_ptgpu", "pathtrace-rs_amd/ptgpu.py"); pthost = _load("pathtrace_rs_amd_pthost", "pathtrace-rs_amd/pthost.py")
W, H, S = 1200, 800, 64
for preset in sys.argv[1:]:
    for bpc in (0, 3, 4, 5):
        hs = pthost.HostScene(preset, W, H, samples=S, device=0); sc = hs.device_scene(); sc.set_tuning(bpc, 8192)
        p = ptgpu.PtParams(W, H, S, 10, 0, 0)
        buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"); rc = torch.zeros(1, dtype=torch.int64, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        best = 1e9
        for _ in range(5):
            sc.update_device(p, hs.camera, 0, buf.data_ptr(), rc.data_ptr(), st); torch.cuda.synchronize(); best = min(best, sc.last_pass_ms())
        g = sc.last_launch_info()
        print("%-20s bpc %d: %.2f ms  %.0f Mrays/s  grid %s  %s" % (preset, bpc, best, int(rc.item()) / best / 1e3, g, sc.last_kernel_choice()["name"]))
