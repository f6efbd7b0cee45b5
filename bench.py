#!/usr/bin/env python3
"""bench.py -- Mrays/s of the path-tracing hot path (Scene::update) on N MI355X GPUs.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by torch.distributed.run, one rank per GPU over RCCL.
A "step" is one Scene::update pass (reference src/scene.rs:73-121) over one frame of
synthetic (preset-generated) input, pixel buffer resident in HBM:
  N = 1 : preset random_spheres 1200x800, 64 spp, depth 10, list world (BASELINE config 3,
          the configuration the metric is quoted on)
  N > 1 : the same frame at 256 spp (BASELINE config 4), rows interleaved across ranks
          (row y -> rank y % N), each rank renders its rows with no data-path collective, then
          ONE RCCL all_gather of the float3 shards (+ an 8-byte all_reduce of the ray count)
          inside the timed region. Total work is fixed as N grows -> "strong".
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def _load(name, rel):
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def cpu_baseline(preset, W, H, S, depth, use_bvh, target_secs=15.0):
    """Time the oracle (C restatement of the reference's rayon/AoS path; kind = "port") on all host
    cores, on a bounded strided pixel sample of the SAME workload. Checker only: never the product."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_binding as ob
    L = ob.lib(ob.build_native())
    cores = os.cpu_count() or 1
    sc = ob.OracleScene(preset, W, H, use_bvh=use_bvh, library=L)
    buf = np.zeros((H, W, 3), np.float32)
    total = W * H
    # calibration pass on ~0.05 % of the pixels, then size the sample for ~target_secs
    cal = np.arange(0, total, 2003, dtype=np.uint32)
    t0 = time.perf_counter()
    _, rays = sc.update(S, depth, 0, buffer=buf, nthreads=cores, pixels=cal)
    dt = max(time.perf_counter() - t0, 1e-6)
    per_pixel = dt / len(cal)
    n = int(min(total, max(len(cal), target_secs / per_pixel)))
    stride = max(1, total // n)
    px = np.arange(0, total, stride, dtype=np.uint32)
    t0 = time.perf_counter()
    _, rays = sc.update(S, depth, 0, buffer=buf, nthreads=cores, pixels=px)
    dt = time.perf_counter() - t0
    return {
        "value": rays / 1e6 / dt, "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": "%d of %d pixels (every %dth) of %s %dx%d %dspp depth %d, %d rays in %.1fs, "
                  "oracle/ptref.c -O3 -march=native -ffp-contract=off, %d pthreads"
                  % (len(px), total, stride, preset, W, H, S, depth, rays, dt, cores),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--preset", default="random_spheres")
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--samples", type=int, default=0, help="0 = 64 at N=1, 256 at N>1")
    ap.add_argument("--depth", type=int, default=10)
    ap.add_argument("--bvh", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-secs", type=float, default=15.0)
    args = ap.parse_args()

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    N = args.gpus
    if world != N and world > 1:
        N = world
    S = args.samples or (64 if N == 1 else 256)
    W, H, depth = args.width, args.height, args.depth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path is HIP-only (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    force_dist = os.environ.get("PT_BENCH_FORCE_DIST") == "1"   # exercise the sharded path with one rank (testing)
    if world > 1 or force_dist:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    ptgpu = _load("pathtrace_rs_amd_ptgpu", "pathtrace-rs_amd/ptgpu.py")
    pthost = _load("pathtrace_rs_amd_pthost", "pathtrace-rs_amd/pthost.py")
    sharding = _load("pathtrace_rs_amd_sharding", "pathtrace-rs_amd/sharding.py")

    # scene + camera built by the C++ host (presets.rs / camera.rs mirror), uploaded via the C ABI
    hs = pthost.HostScene(args.preset, W, H, samples=S, use_bvh=args.bvh, device=local_rank)
    scene = hs.device_scene()
    n_spheres = hs.desc.n_spheres
    params = ptgpu.PtParams(W, H, S, depth, 0, 1 if args.bvh else 0)
    cam = hs.camera

    stream = torch.cuda.current_stream()
    max_rows = sharding.padded_rows(H, N)
    shard = torch.zeros((max_rows, W, 3), dtype=torch.float32, device=dev)
    ray_count = torch.zeros(1, dtype=torch.int64, device=dev)
    gathered = torch.empty((N, max_rows, W, 3), dtype=torch.float32, device=dev) if (N > 1 or dist is not None) else None
    frame = None
    kernel_ms = []

    def step():
        nonlocal frame
        shard.zero_()  # frame 0 of a fresh accumulation (offline.rs:25 starts from zeros)
        if N == 1 and dist is None:
            scene.update_device(params, cam, 0, shard.data_ptr(), ray_count.data_ptr(), stream.cuda_stream)
            frame = shard
        else:
            scene.update_shard_device(params, cam, 0, rank, N, shard.data_ptr(), ray_count.data_ptr(),
                                      stream.cuda_stream)
            # RCCL over xGMI: ONE all_gather per frame (+ 8-byte all_reduce, scene.rs:118-120), de-interleave
            frame = sharding.gather_frame(dist, shard, gathered, ray_count, H)
        return frame

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # kernel duration of this step from the HIP events recorded on the launch stream
        # (pt_last_kernel_ms synchronises on the stop event only)
        kernel_ms.append(scene.last_kernel_ms())
    fence()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    rays_per_step = int(ray_count.item())  # already summed over ranks
    total_rays = rays_per_step * args.steps
    value = total_rays / 1e6 / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    kms = sum(kernel_ms) / max(1, len(kernel_ms))
    k_t = torch.tensor([kms], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(k_t, op=dist.ReduceOp.MAX)
    kms = float(k_t.item())

    if rank == 0:
        grid, block, lds = scene.last_launch_info()
        # algorithmic bytes/ray in list mode = 16 B x N_spheres (cx,cy,cz,r^2 scanned once per
        # ray); the scan is LDS-served, so "achieved" is an EFFECTIVE rate and may exceed the HBM peak.
        bytes_per_ray = 16.0 * n_spheres
        launch_bytes = bytes_per_ray * rays_per_step / N      # one launch = one rank's shard
        achieved = launch_bytes / (kms * 1e-3) / 1e9
        # The same algorithmic work expressed as the reference's arithmetic: 18 unfused f32 lane-ops per sphere
        # test. The kernel does NOT execute these for every pair: an f16 MFMA prefilter discards certain misses
        # and only survivors run the exact arithmetic, so this "equivalent" rate may exceed the VALU ceiling.
        valu_ops = (18.0 * n_spheres + 150.0) * rays_per_step / N
        valu_rate = valu_ops / (kms * 1e-3) / 1e12
        # matrix-core work actually issued by the prefilter: ceil(n/32) tiles x 4 v_mfma_f32_32x32x16_f16
        # (32768 flop each) per 64 rays (list mode only)
        mfma_tf = 0.0 if args.bvh else (-(-n_spheres // 32) * 4 * 32768.0 / 64.0) * rays_per_step / N / (kms * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/sec, random_spheres 1200x800 64spp" if (args.preset == "random_spheres" and N == 1)
                      else "Mrays/sec, %s %dx%d %dspp" % (args.preset, W, H, S),
            "value": value, "unit": "Mrays/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "preset %s (%d spheres) %dx%d %dspp depth %d %s, frame 0, seed 0"
                                   % (args.preset, n_spheres, W, H, S, depth, "BVH" if args.bvh else "list"),
                       "rays_per_step": rays_per_step, "wall_secs_per_frame": ms_per_step / 1e3,
                       "parallelism": "rows interleaved over %d GPU(s)%s" % (N, ", RCCL all_gather" if N > 1 else ""),
                       "grid": grid, "block": block, "lds_bytes": lds},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic,
                         "kernel": "pt_trace_kernel", "kernel_ms": kms,
                         "note": "effective scan bandwidth: 16 B x %d spheres per ray, served from LDS (never HBM), so it "
                                 "exceeds the HBM peak by construction; measured HBM traffic is in `traffic`. The reference's "
                                 "arithmetic for that scan equals %.1f T f32 lane-ops/s (VALU ceiling ~67 T measured, 78.6 T "
                                 "nominal); the kernel replaces most of it by an f16 MFMA prefilter running at %.0f TFLOP/s "
                                 "(dense f16 peak ~2500) and is bound by divergent shading + per-iteration latency, see "
                                 "DESIGN.md section 4" % (n_spheres, valu_rate, mfma_tf),
                         "valu_equiv_frac": valu_rate / 78.6, "mfma_tflops": mfma_tf, "mfma_frac": mfma_tf / 2500.0},
        }
        if N == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.preset, W, H, S, depth, args.bvh, args.cpu_secs)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
