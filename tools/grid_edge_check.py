"""Development aid (GPU box): the cell-grid kernel on config 5's scene at odd frame sizes, depths 0 / 1 / 50, list and BVH -- against the tree kernel, and
the three row shards of each frame against the whole."""
import importlib, os, sys
import numpy as np
import torch
torch.zeros(1, device="cuda:0")   # (torch's HIP context first: it does not find the device once another library has initialised the runtime)
sys.path.insert(0, os.getcwd())
pthost = importlib.import_module("pathtrace-rs_amd.pthost"); ptgpu = pthost.ptgpu
bad = 0
for (W, H, S, depth, bvh) in [(16, 8, 1, 0, True), (33, 17, 3, 1, False), (64, 36, 2, 50, True), (200, 3, 5, 10, False), (1, 1, 7, 10, True)]:
    hs = pthost.HostScene("perlin_spheres", W, H, samples=S, use_bvh=bvh, device=0)
    dev = hs.device_scene()
    res = {}
    for name, var in (("grid", 0), ("tree", 524288)):
        dev.set_tuning(0, var)
        out = np.zeros((H, W, 3), np.float32)
        rays = dev.update(ptgpu.PtParams(W, H, S, depth, 0, 1 if bvh else 0), hs.camera, 0, out)
        res[name] = (rays, out, dev.last_kernel_choice()["name"])
    ok = res["grid"][0] == res["tree"][0] and np.array_equal(res["grid"][1], res["tree"][1], equal_nan=True)
    # sharded: rows y % 3
    dev.set_tuning(0, 0)
    full = res["grid"][1]
    shard_ok = True
    if H >= 3:
        tot = 0
        for r in range(3):
            rows = len(range(r, H, 3))
            sh = torch.zeros((rows, W, 3), dtype=torch.float32, device="cuda:0")
            rc = torch.zeros(1, dtype=torch.int64, device="cuda:0")
            dev.update_shard_device(ptgpu.PtParams(W, H, S, depth, 0, 1 if bvh else 0), hs.camera, 0, r, 3, sh.data_ptr(), rc.data_ptr(), 0)
            torch.cuda.synchronize()
            tot += int(rc.item())
            shard_ok &= bool(np.array_equal(sh.cpu().numpy(), full[r::3], equal_nan=True))
        shard_ok &= tot == res["grid"][0]
    print(W, H, S, depth, bvh, res["grid"][2], res["tree"][2], "rays", res["grid"][0], "grid==tree", ok, "shards", shard_ok)
    bad += (not ok) or (not shard_ok)
print("edge check:", "OK" if not bad else "%d FAILED" % bad)
