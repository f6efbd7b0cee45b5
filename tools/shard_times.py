#!/usr/bin/env python3
"""What each GPU of an N-GPU run of BASELINE config 4 would spend rendering, measured on ONE GPU: shard r of N
(rows y % N == r) of random_spheres 1200x800 at 256 spp, pass time (pilot + sort + frame kernel) from HIP events.
The slowest shard bounds the N-GPU frame (the gather adds ~0.1 ms); prints the projected strong-scaling curve."""
import argparse
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="random_spheres")
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--samples", type=int, default=256)
    ap.add_argument("--counts", default="1,2,4,8")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    ptgpu = _load("pathtrace_rs_amd_ptgpu", "pathtrace-rs_amd/ptgpu.py")
    pthost = _load("pathtrace_rs_amd_pthost", "pathtrace-rs_amd/pthost.py")
    W, H, S = args.width, args.height, args.samples
    hs = pthost.HostScene(args.preset, W, H, samples=S, device=0)
    sc = hs.device_scene()
    sc.set_tuning(0, 8192)     # every repetition measures its own work order, as bench.py's timed steps do (no reuse of the last frame's costs)
    p = ptgpu.PtParams(W, H, S, 10, 0, 0)
    stream = torch.cuda.current_stream().cuda_stream
    rc = torch.zeros(1, dtype=torch.int64, device="cuda")
    buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    base = None
    for N in [int(x) for x in args.counts.split(",")]:
        worst, total_rays, times, shard_rays = 0.0, 0, [], []
        for r in range(N):
            best = 1e30
            for _ in range(args.reps):
                buf.zero_()
                sc.update_shard_device(p, hs.camera, 0, r, N, buf.data_ptr(), rc.data_ptr(), stream)
                torch.cuda.synchronize()
                best = min(best, sc.last_pass_ms())
            times.append(best)
            total_rays += int(rc.item())
            shard_rays.append(int(rc.item()))
            worst = max(worst, best)
        if base is None:
            base = worst * N
        print("N=%d  slowest shard %.2f ms (shards: %s)  -> %.0f Mrays/s, %.2fx of ideal vs N=%d"
              % (N, worst, " ".join("%.2f" % t for t in times), total_rays / 1e3 / worst, base / N / worst, int(args.counts.split(",")[0])))
        if N > 1:   # (are the slow shards the ones with more rays? rows dealt y % N give every shard the same rays to within a fraction of a percent)
            mean = sum(shard_rays) / N
            print("      rays per shard relative to their mean: %s; time max / min %.3f" % (" ".join("%+.2f%%" % (100.0 * (r / mean - 1.0)) for r in shard_rays), max(times) / min(times)))


if __name__ == "__main__":
    main()
