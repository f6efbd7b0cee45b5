#!/usr/bin/env python3
"""bench.py -- Mrays/s of the path-tracing hot path (Scene::update) on N MI355X GPUs.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by torch.distributed.run, one rank per GPU over RCCL.
A "step" is one Scene::update pass (reference src/scene.rs:73-121) over one frame of
synthetic (preset-generated) input, pixel buffer resident in HBM:
  N = 1 : preset random_spheres 1200x800, 64 spp, depth 10, list world (BASELINE config 3,
          the configuration the metric is quoted on)
  N > 1 : two ways to use N GPUs, both measured, both inside the timed region end to end:
          --mode frames (default, "weak"): the unit of work is one frame of that SAME configuration. Rank r
              renders progressive frame frame_num = r (scene.rs:99-101 seeds depend on (x, y, frame) only) with no
              data-path collective, ONE RCCL all_gather collects the N frames and the reference's blend
              (scene.rs:113-116) is replayed in frame order: bit-identical to `-F N` on one GPU, N x 64 spp worth of
              samples in the image. Per-GPU work is fixed as N grows.
          --mode tiles ("strong"): ONE frame, rows interleaved across ranks (row y -> rank y % N), each rank
              renders its rows, ONE all_gather of the float3 shards. Samples of a pixel are serial (one RNG stream
              per pixel), so a 15 ms frame cannot strong-scale well; the default run reports this number too, as
              `strong_scaling_tiles`, measured right after the timed region.
          Both add an 8-byte all_reduce of the ray count (scene.rs:118-120).
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
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    sc = ob.OracleScene(preset, W, H, use_bvh=use_bvh, library=L)
    buf = np.zeros((H, W, 3), np.float32)
    total = W * H
    # one thread on ~0.02 % of the pixels: the per-core rate, so that the all-threads figure can be read as
    # "how many cores' worth of CPU this process really got" (containers often see more CPUs than they may use)
    one = np.arange(0, total, 4999, dtype=np.uint32)
    t0 = time.perf_counter()
    _, rays1 = sc.update(S, depth, 0, buffer=buf, nthreads=1, pixels=one)
    rate1 = rays1 / 1e6 / max(time.perf_counter() - t0, 1e-6)
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
    value = rays / 1e6 / dt
    return {
        "value": value, "unit": "Mrays/s", "cores": cores, "kind": "port",
        "one_thread": rate1, "speedup_over_one_thread": value / rate1 if rate1 > 0 else None,
        "sample": "%d of %d pixels (every %dth) of %s %dx%d %dspp depth %d, %d rays in %.1fs, "
                  "oracle/ptref.c -O3 -march=native -ffp-contract=off, %d pthreads (one thread alone: %.2f Mrays/s, "
                  "so the %d threads delivered %.1f cores' worth)"
                  % (len(px), total, stride, preset, W, H, S, depth, rays, dt, cores, rate1, cores, value / max(rate1, 1e-9)),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--preset", default="random_spheres")
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--samples", type=int, default=64, help="samples per pixel (BASELINE config 4 = 256 with --mode tiles)")
    ap.add_argument("--mode", choices=["frames", "tiles"], default="frames", help="how N > 1 GPUs are used (see above)")
    ap.add_argument("--depth", type=int, default=10)
    ap.add_argument("--bvh", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="N = 1: skip the extra measurement of overlapped independent frames")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: run the collective on the render stream (no double buffering)")
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
    S = args.samples
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

    if not os.path.exists(os.path.join(ROOT, "pathtrace-rs_amd", "_build", "libpthost.so")):
        if local_rank == 0:   # a checkout without the in-tree build: compile it once (never a fallback path)
            import subprocess
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "pathtrace-rs_amd"), "all"], stdout=subprocess.DEVNULL)
        if dist is not None:
            dist.barrier()
    ptgpu = _load("pathtrace_rs_amd_ptgpu", "pathtrace-rs_amd/ptgpu.py")
    pthost = _load("pathtrace_rs_amd_pthost", "pathtrace-rs_amd/pthost.py")
    sharding = _load("pathtrace_rs_amd_sharding", "pathtrace-rs_amd/sharding.py")

    # scene + camera built by the C++ host (presets.rs / camera.rs mirror), uploaded via the C ABI
    hs = pthost.HostScene(args.preset, W, H, samples=S, use_bvh=args.bvh, device=local_rank)
    scene = hs.device_scene()
    n_spheres = hs.world_desc.n_hitables
    params = ptgpu.PtParams(W, H, S, depth, 0, 1 if args.bvh else 0)
    cam = hs.camera

    stream = torch.cuda.current_stream()
    multi = N > 1 or dist is not None
    max_rows = sharding.padded_rows(H, N)
    # Three buffer sets: the collective + blend of step k run on their own HIP stream while the kernels of steps k + 1
    # and k + 2 render into the other sets. The persistent grid holds every CU, so the exchange of step k actually runs
    # when the workgroups of step k + 1 retire; with only two sets step k + 2 would have to wait for it.
    overlap = multi and not args.no_overlap
    comm = torch.cuda.Stream(device=dev) if overlap else stream
    sets = [dict(full=torch.zeros((H, W, 3), dtype=torch.float32, device=dev),
                 shard=torch.zeros((max_rows, W, 3), dtype=torch.float32, device=dev),
                 rays=torch.zeros(1, dtype=torch.int64, device=dev),
                 rows=torch.empty((N, max_rows, W, 3), dtype=torch.float32, device=dev) if multi else None,
                 frames=torch.empty((N, H, W, 3), dtype=torch.float32, device=dev) if multi else None,
                 done=None) for _ in range(3 if overlap else 1)]
    state = {"k": 0, "last": sets[0], "frame": None}

    def step(mode):
        b = sets[state["k"] % len(sets)]
        state["k"] += 1
        state["last"] = b
        if overlap and b["done"] is not None:
            stream.wait_event(b["done"])      # the collective that read this set two steps ago has finished
        if not multi:
            b["full"].zero_()  # frame 0 of a fresh accumulation (offline.rs:25 starts from zeros)
            scene.update_device(params, cam, 0, b["full"].data_ptr(), b["rays"].data_ptr(), stream.cuda_stream)
            state["frame"] = b["full"]
            return
        if mode == "frames":
            b["full"].zero_()
            scene.update_device(params, cam, rank, b["full"].data_ptr(), b["rays"].data_ptr(), stream.cuda_stream)
        else:
            b["shard"].zero_()
            scene.update_shard_device(params, cam, 0, rank, N, b["shard"].data_ptr(), b["rays"].data_ptr(), stream.cuda_stream)
        if overlap:
            ready = torch.cuda.Event()
            ready.record(stream)
            comm.wait_event(ready)
        with torch.cuda.stream(comm):
            if mode == "frames":
                # RCCL over xGMI: ONE all_gather of the N frames, blend replayed in frame order (scene.rs:113-116)
                state["frame"] = sharding.gather_progressive(dist, b["full"], b["frames"], b["rays"])
            else:
                # ONE all_gather of the row shards per frame (+ 8-byte all_reduce, scene.rs:118-120), de-interleave
                state["frame"] = sharding.gather_frame(dist, b["shard"], b["rows"], b["rays"], H)
            if overlap:
                b["done"] = torch.cuda.Event()
                b["done"].record(comm)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(mode, steps, warmup):
        """W untimed + K timed steps bracketed by barrier + synchronize; MAX over ranks."""
        kms_list = []
        for _ in range(warmup):
            step(mode)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(mode)
            # kernel duration of this step from the HIP events recorded on the launch stream (pt_last_kernel_ms
            # synchronises on the stop event only). In the overlapped multi-GPU pipeline the host must not wait per
            # step (the next kernel is enqueued behind the running one): there the last step's duration is read.
            if not overlap:
                kms_list.append(scene.last_kernel_ms())
        fence()
        el = time.perf_counter() - t0
        if overlap:
            kms_list.append(scene.last_kernel_ms())
        t = torch.tensor([el, sum(kms_list) / max(1, len(kms_list))], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0].item()), float(t[1].item()), int(state["last"]["rays"].item())  # already summed over ranks

    elapsed, kms, rays_per_step = timed(args.mode, args.steps, args.warmup)
    total_rays = rays_per_step * args.steps
    value = total_rays / 1e6 / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    strong = None
    if multi and args.mode == "frames":   # the other decomposition, reported beside the main value
        k2 = max(2, min(args.steps, 5))
        el2, kms2, rays2 = timed("tiles", k2, 1)
        strong = {"value": rays2 * k2 / 1e6 / el2, "unit": "Mrays/s", "ms_per_step": el2 / k2 * 1e3, "kernel_ms": kms2,
                  "steps": k2, "scaling": "strong",
                  "workload": "ONE %dx%d %dspp frame, rows interleaved over %d GPUs, all_gather of the shards" % (W, H, S, N)}

    pipelined = None
    if not multi and not args.no_pipeline:
        # Extra figure, never `value`: independent frames back to back on two scene handles / two HIP streams, so the
        # tail of frame k (its last, serial pixels) and the pilot pass of frame k + 1 overlap. A single frame cannot
        # use this; a renderer producing a sequence of independent frames (animation, tiles of a bigger image) can.
        hs2 = pthost.HostScene(args.preset, W, H, samples=S, use_bvh=args.bvh, device=local_rank)
        handles = [(scene, stream, torch.zeros((H, W, 3), dtype=torch.float32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)),
                   (hs2.device_scene(), torch.cuda.Stream(device=dev), torch.zeros((H, W, 3), dtype=torch.float32, device=dev),
                    torch.zeros(1, dtype=torch.int64, device=dev))]

        def pstep(k):
            sc, st, buf, rc = handles[k % 2]
            with torch.cuda.stream(st):
                buf.zero_()
                sc.update_device(params, cam, 0, buf.data_ptr(), rc.data_ptr(), st.cuda_stream)

        kp = max(4, args.steps)
        for k in range(2):
            pstep(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(kp):
            pstep(k)
        torch.cuda.synchronize()
        elp = time.perf_counter() - t0
        assert int(handles[0][3].item()) == rays_per_step and int(handles[1][3].item()) == rays_per_step
        assert torch.equal(handles[0][2], state["frame"]) and torch.equal(handles[1][2], state["frame"])
        pipelined = {"value": rays_per_step * kp / 1e6 / elp, "unit": "Mrays/s", "ms_per_frame": elp / kp * 1e3, "frames": kp,
                     "note": "independent frames alternating over two scene handles and two HIP streams (the tail of one frame "
                             "overlaps the start of the next); each frame is bit-identical to the single-frame result"}

    if multi and os.environ.get("PT_BENCH_CHECK") == "1":
        # self-check of the double-buffered pipeline: its last frame must equal a plain serial step, bit for bit
        for _ in range(3):
            step(args.mode)
        fence()
        got = state["frame"].clone()
        ref_full = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
        ref_shard = torch.zeros((max_rows, W, 3), dtype=torch.float32, device=dev)
        rc2 = torch.zeros(1, dtype=torch.int64, device=dev)
        if args.mode == "frames":
            scene.update_device(params, cam, rank, ref_full.data_ptr(), rc2.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            want = sharding.gather_progressive(dist, ref_full, torch.empty((N, H, W, 3), dtype=torch.float32, device=dev), rc2)
        else:
            scene.update_shard_device(params, cam, 0, rank, N, ref_shard.data_ptr(), rc2.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            want = sharding.gather_frame(dist, ref_shard, torch.empty((N, max_rows, W, 3), dtype=torch.float32, device=dev), rc2, H)
        torch.cuda.synchronize()
        assert torch.equal(got, want) and int(rc2.item()) == int(state["last"]["rays"].item()), "pipelined frame differs from the serial one"
        if rank == 0:
            print("[bench check] pipelined %s frame == serial frame, %d rays" % (args.mode, int(rc2.item())), file=sys.stderr)

    if rank == 0:
        grid, block, lds = scene.last_launch_info()
        # algorithmic bytes/ray in list mode = 16 B x N_spheres (cx,cy,cz,r^2 scanned once per
        # ray); the scan is LDS-served, so "achieved" is an EFFECTIVE rate and may exceed the HBM peak.
        bytes_per_ray = 16.0 * n_spheres
        launch_bytes = bytes_per_ray * rays_per_step / N      # one launch = one rank's frame (or shard)
        achieved = launch_bytes / (kms * 1e-3) / 1e9
        # The same algorithmic work expressed as the reference's arithmetic: 18 unfused f32 lane-ops per sphere
        # test. The kernel does NOT execute these for every pair: an f16 MFMA prefilter discards certain misses
        # and only survivors run the exact arithmetic, so this "equivalent" rate may exceed the VALU ceiling.
        valu_ops = (18.0 * n_spheres + 150.0) * rays_per_step / N
        valu_rate = valu_ops / (kms * 1e-3) / 1e12
        # matrix-core work of the prefilter if EVERY tile ran: ceil(n/32) tiles x 4 v_mfma_f32_32x32x16_f16 (32768 flop
        # each) per 64 rays (list mode only). Tile culling skips tiles wave by wave, so what was actually issued is
        # taken from the committed rocprofv3 counters (SQ_INSTS_MFMA per launch) when they are for this workload.
        mfma_tf_max = 0.0 if args.bvh else (-(-n_spheres // 32) * 4 * 32768.0 / 64.0) * rays_per_step / N / (kms * 1e-3) / 1e12
        mfma_tf = mfma_tf_max
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                prof = json.load(open(pmc))
                traffic = prof.get("hbm_bytes_per_launch")
                same = prof.get("bench_line_under_profiler", {}).get("config", {}).get("rays_per_step") == rays_per_step // N
                insts = prof.get("pmc_per_launch", {}).get("SQ_INSTS_MFMA")
                if same and insts and not args.bvh:
                    mfma_tf = insts * 32768.0 / (kms * 1e-3) / 1e12
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/sec, random_spheres 1200x800 64spp" if (args.preset == "random_spheres" and (W, H, S) == (1200, 800, 64))
                      else "Mrays/sec, %s %dx%d %dspp" % (args.preset, W, H, S),
            "value": value, "unit": "Mrays/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak" if (args.mode == "frames" or not multi) else "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "preset %s (%d hitables) %dx%d %dspp depth %d %s, seed 0, %s"
                                   % (args.preset, n_spheres, W, H, S, depth, "BVH" if args.bvh else "list",
                                      "frame 0" if not multi else ("progressive frames 0..%d, one per GPU" % (N - 1) if args.mode == "frames"
                                                                   else "frame 0 split by rows")),
                       "rays_per_step": rays_per_step, "wall_secs_per_step": ms_per_step / 1e3,
                       "parallelism": ("1 GPU" if not multi else
                                       ("frame_num = rank on each of %d GPUs, no data-path collective, RCCL all_gather of the frames + "
                                        "blend in frame order (scene.rs:113-116)" % N if args.mode == "frames"
                                        else "rows interleaved over %d GPUs, RCCL all_gather of the shards" % N)),
                       "overlap": ("collective + blend of step k on a second HIP stream under the kernel of step k + 1" if overlap
                                   else "none"),
                       "grid": grid, "block": block, "lds_bytes": lds},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic,
                         "kernel": "pt_trace_kernel", "kernel_ms": kms,
                         "note": "effective scan bandwidth: 16 B x %d spheres per ray, served from LDS (never HBM), so it "
                                 "exceeds the HBM peak by construction; measured HBM traffic is in `traffic`. The reference's "
                                 "arithmetic for that scan equals %.1f T f32 lane-ops/s (VALU ceiling ~67 T measured, 78.6 T "
                                 "nominal); the kernel replaces most of it by an f16 MFMA prefilter that issues %.0f TFLOP/s "
                                 "(%.0f if no tile were culled; dense f16 peak ~2500) and is bound by VALU issue in the tile "
                                 "loop, the exact phase 2 and divergent shading, see DESIGN.md section 4"
                                 % (n_spheres, valu_rate, mfma_tf, mfma_tf_max),
                         "valu_equiv_frac": valu_rate / 78.6, "mfma_tflops": mfma_tf, "mfma_frac": mfma_tf / 2500.0},
        }
        if strong is not None:
            out["strong_scaling_tiles"] = strong
        if pipelined is not None:
            out["pipelined_frames"] = pipelined
        if N == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.preset, W, H, S, depth, args.bvh, args.cpu_secs)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
