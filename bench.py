#!/usr/bin/env python3
"""bench.py -- Mrays/s of the path-tracing hot path (Scene::update) on N MI355X GPUs.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by torch.distributed.run, one rank per GPU over RCCL; started WITHOUT a launcher
  (`python bench.py --gpus 4`) it starts that launcher itself as a child process (never exec) and exits with its code:
      python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...
A "step" is one Scene::update pass (reference src/scene.rs:73-121) over one frame of synthetic (preset-generated)
input, pixel buffer resident in HBM:
  N = 1 : BASELINE config 3, the configuration the metric is quoted on: preset random_spheres 1200x800, 64 spp,
          depth 10, list world.
  N > 1 : BASELINE config 4: ONE random_spheres 1200x800 frame at 256 spp, split over the N GPUs by rows (row y ->
          rank y % N, disjoint pixels as scene.rs:90-93), no collective while rendering, then ONE RCCL gather of the
          float3 shards + an 8-byte all-reduce of the ray count (scene.rs:118-120) -- both issued by the C ABI
          (pt_comm_gather_frame / pt_render_sharded in include/ptgpu.h; ncclAllGather + ncclAllReduce on xGMI) and
          both inside the timed region. "scaling": "strong" (total work fixed as N grows).
          --mode frames is the other data-parallel axis (rank r renders progressive frame r of the 64-spp frame;
          bit-identical to `-F N`); the default run reports it as the extra `weak_scaling_frames`, never as `value`.
Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import importlib.util
import json
import os
import subprocess
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


def cpu_quota_cores():
    """CPU time this container may use, in cores, and where that was read: cgroup v2 cpu.max / v1 cfs quota, capped by the effective
    cpuset. (None, reason) when nothing limits it. sched_getaffinity alone says 256 on a box that pays for 8."""
    def read(path):
        try:
            return open(path).read().strip()
        except OSError:
            return None
    quota, src = None, "no cgroup CPU quota found"
    v2 = read("/sys/fs/cgroup/cpu.max")
    if v2:
        q, _, per = v2.partition(" ")
        if q != "max" and per:
            quota, src = float(q) / float(per), "/sys/fs/cgroup/cpu.max = " + v2
        else:
            src = "/sys/fs/cgroup/cpu.max = " + v2
    else:
        q, per = read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), read("/sys/fs/cgroup/cpu/cpu.cfs_period_us")
        if q and per and int(q) > 0:
            quota, src = int(q) / int(per), "cpu.cfs_quota_us / cpu.cfs_period_us = %s / %s" % (q, per)
    cs = read("/sys/fs/cgroup/cpuset.cpus.effective") or read("/sys/fs/cgroup/cpuset/cpuset.effective_cpus")
    if cs:
        n = 0
        for part in cs.split(","):
            a, _, b = part.partition("-")
            n += (int(b) - int(a) + 1) if b else 1
        if n and (quota is None or n < quota):
            quota, src = float(n), "cpuset.cpus.effective = " + cs
    return quota, src


def cpu_baseline(preset, W, H, S, depth, use_bvh, target_secs=15.0):
    """Time the oracle (C restatement of the reference's rayon/AoS path; kind = "port") on all host
    cores, on a bounded strided pixel sample of the SAME workload. Checker only: never the product."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_binding as ob
    L = ob.lib(ob.build_native())
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota, quota_src = cpu_quota_cores()
    quota_threads = max(1, min(cores, int(-(-quota // 1)))) if quota else None   # min(affinity, ceil(quota))
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
    # the same sample with as many threads as the container's CPU quota pays for: when that figure equals the all-threads one, "N cores'
    # worth" is the box's limit, not the oracle's scaling (offline.rs:27-41 is the timer both mirror)
    at_quota = None
    if quota_threads is not None and quota_threads != cores:
        t0 = time.perf_counter()
        _, rays_q = sc.update(S, depth, 0, buffer=buf, nthreads=quota_threads, pixels=px)
        at_quota = rays_q / 1e6 / (time.perf_counter() - t0)
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    # the figure to quote is the better of the two legs: with 256 threads on a 16-core quota the kernel throttles the process and the
    # all-threads leg delivers HALF of what 16 threads do (9.7 vs 20.2 Mrays/s on the pool's EPYC 9575F boxes)
    all_threads = value
    used = cores
    if at_quota is not None and at_quota > value:
        value, used = at_quota, quota_threads
    return {
        "value": value, "unit": "Mrays/s", "cores": used, "kind": "port",
        "cpu_model": model, "nproc": os.cpu_count(), "threads": used,
        "value_all_threads": all_threads, "all_threads": cores,
        "one_thread": rate1, "speedup_over_one_thread": value / rate1 if rate1 > 0 else None,
        "cpu_quota_cores": quota, "cpu_quota_source": quota_src, "quota_threads": quota_threads, "value_at_quota_threads": at_quota,
        "speedup_at_quota_threads": (at_quota / rate1 if (at_quota and rate1 > 0) else None),
        "sample": "%d of %d pixels (every %dth) of %s %dx%d %dspp depth %d, %d rays in %.1fs, "
                  "oracle/ptref.c -O3 -march=native -ffp-contract=off, %d pthreads: %.2f Mrays/s (one thread alone: %.2f Mrays/s, "
                  "so the %d threads delivered %.1f cores' worth)%s"
                  % (len(px), total, stride, preset, W, H, S, depth, rays, dt, cores, all_threads, rate1, cores, all_threads / max(rate1, 1e-9),
                     ("; the same sample with %d threads, what the container's CPU quota (%s) pays for: %.2f Mrays/s = %.1f cores' worth -- `value`"
                      % (quota_threads, quota_src, at_quota, at_quota / max(rate1, 1e-9))) if (at_quota is not None and at_quota > all_threads) else ""),
    }


# MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, 2.4 GHz, a wave64 VALU instruction issues over 2 cycles
VALU_PEAK_TLANEOPS = 256 * 4 * 2.4e9 / 2 * 64 / 1e12      # 78.6 T lane-ops/s nominal
VALU_PEAK_MEASURED = 67.0                                  # tools/valu_bench.hip on the GPU box: v_fma_f32 (2.3 cycles / wave-instruction)
VALU_SIMPLE_MEASURED = 105.0                               # same tool, plain v_mul_f32 / v_add_f32 at 4 waves per SIMD (1.5 cycles)
HBM_PEAK_GBS = 8000.0
N_SIMD = 1024


def committed_counters(preset, W, H, S, use_bvh):
    """rocprofv3 PMC counters of the frame kernel for this workload, per RAY, from the newest committed
    profiles/r*_pmc_traffic.json whose bench line matches (tools/profile.sh + tools/profile_summary.py write them;
    PMC passes cannot run inside bench.py). Returns (per-ray dict, per-launch dict, file name) or None."""
    want = "preset %s " % preset
    shape = " %dx%d %dspp " % (W, H, S)
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            prof = json.load(open(f))
            line = prof["bench_line_under_profiler"]
            wl = line["config"]["workload"]
            if line.get("n_gpus", 1) != 1 or want not in wl or shape not in wl or ((" BVH" in wl) != bool(use_bvh)):
                continue
            rays = float(line["config"]["rays_per_step"])
            per_launch = dict(prof["pmc_per_launch"])
            for k in ("hbm_bytes_per_launch", "hbm_read_bytes_corrected", "hbm_write_bytes", "hbm_bytes_measuring_launch"):
                if k in prof:
                    per_launch[k] = prof[k]
            per_launch["kernel_avg_ms_rocprof"] = prof.get("kernel_avg_ms")
            per_launch["build"] = line.get("build")   # pt_version() of the library the counters were taken on (None: a profile from before round 6)
            best = ({k: v / rays for k, v in per_launch.items() if isinstance(v, (int, float))}, per_launch, os.path.basename(f))
            break
        except Exception:
            continue
    return best


def roofline_block(kernel_name, kms, rays_launch, n_hitables, use_bvh, counters, build=None):
    """Counter-derived fractions of the resources the frame kernel uses. Nothing here is an "effective" figure:
    SURVEY 8(d)'s scan-equivalent rate is kept apart under `algorithmic_equiv`. `build`: pt_version() of the library being timed; counters
    taken on any other build are STALE: the block says so and carries no `frac` (the figures move to *_stale_counters keys)."""
    ksec = kms * 1e-3
    out = {"bound": "valu_issue", "achieved": None, "peak": VALU_PEAK_TLANEOPS, "peak_measured": VALU_PEAK_MEASURED,
           "unit": "T lane-ops/s", "frac": None, "traffic": None, "kernel": kernel_name, "kernel_ms": kms}
    if counters is not None:
        per_ray, per_launch, fname = counters
        valu = per_ray.get("SQ_INSTS_VALU", 0.0) * rays_launch               # wave-instructions this launch issued
        out["achieved"] = valu * 64.0 / ksec / 1e12
        out["frac"] = out["achieved"] / VALU_PEAK_TLANEOPS
        out["frac_of_measured_peak"] = out["achieved"] / VALU_PEAK_MEASURED
        # tools/valu_bench.hip at 4 waves per SIMD: v_fma_f32 62 T lane-ops/s, plain v_mul / v_add 105 T (121 T at 8 waves). The
        # path computes unfused (results must round like the reference), so most of its VALU work is of the second kind:
        out["frac_of_simple_op_rate"] = out["achieved"] / VALU_SIMPLE_MEASURED
        out["valu_wave_insts_per_64_rays"] = per_ray.get("SQ_INSTS_VALU", 0.0) * 64.0
        if "hbm_bytes_per_launch" in per_ray:
            out["traffic"] = per_ray["hbm_bytes_per_launch"] * rays_launch
            out["hbm_gbs"] = out["traffic"] / ksec / 1e9
            out["hbm_frac"] = out["hbm_gbs"] / HBM_PEAK_GBS
            out["traffic_note"] = "traffic = HBM bytes of the FRAME kernel alone (the dominant kernel, per launch)"
            if "hbm_bytes_measuring_launch" in per_ray:
                out["traffic_both_launches"] = (per_ray["hbm_bytes_per_launch"] + per_ray["hbm_bytes_measuring_launch"]) * rays_launch
                out["traffic_note"] += "; traffic_both_launches adds the measuring launch that precedes it (it parks 48 B per pixel)"
        if "SQ_BUSY_CYCLES" in per_launch and per_launch.get("kernel_avg_ms_rocprof"):
            # SQ_BUSY_CYCLES sums the 32 shader engines; / 32 = cycles the kernel ran = the clock it really had
            cyc = per_launch["SQ_BUSY_CYCLES"] / 32.0
            out["shader_clock_ghz"] = cyc / (per_launch["kernel_avg_ms_rocprof"] * 1e-3) / 1e9
            if "SQ_VALU_MFMA_BUSY_CYCLES" in per_launch:
                out["mfma_busy_frac"] = per_launch["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * N_SIMD)
            # the nominal peak assumes 2.4 GHz; the same fraction against the clock the kernel really ran at
            out["peak_at_measured_clock"] = VALU_PEAK_TLANEOPS * out["shader_clock_ghz"] / 2.4
            out["frac_at_measured_clock"] = out["achieved"] / out["peak_at_measured_clock"]
            if "SQ_WAIT_INST_ANY" in per_launch and "SQ_WAVE_CYCLES" in per_launch:
                out["issue_stall_frac_of_wave_cycles"] = per_launch["SQ_WAIT_INST_ANY"] / per_launch["SQ_WAVE_CYCLES"]
                out["parked_frac_of_wave_cycles"] = per_launch.get("SQ_WAIT_ANY", 0.0) / per_launch["SQ_WAVE_CYCLES"]
        if per_launch.get("SQ_ACTIVE_INST_VALU") and "SQ_THREAD_CYCLES_VALU" in per_launch:
            # `frac` counts every issued wave-instruction as 64 lanes: it is ISSUE-SLOT occupancy. How many of those lanes were
            # switched on is rocprof's VALUUtilization = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64) (calibrated with
            # tools/valu_bench.hip at 64 / 32 / 16 active lanes: profiles/r04_valu_lane_calibration.txt); their product is the
            # fraction of the peak spent on lanes that compute something
            out["valu_lane_utilisation"] = per_launch["SQ_THREAD_CYCLES_VALU"] / (per_launch["SQ_ACTIVE_INST_VALU"] * 64.0)
            out["frac_useful"] = out["frac"] * out["valu_lane_utilisation"]
            out["achieved_useful"] = out["achieved"] * out["valu_lane_utilisation"]
        if "SQ_INSTS_LDS" in per_ray:
            out["lds_wave_insts"] = per_ray["SQ_INSTS_LDS"] * rays_launch
            out["lds_bank_conflict_cycles_per_inst"] = per_launch.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(per_launch["SQ_INSTS_LDS"], 1.0)
        if "SQ_INSTS_SALU" in per_ray:
            out["salu_per_valu"] = per_ray["SQ_INSTS_SALU"] / max(per_ray.get("SQ_INSTS_VALU", 0.0), 1e-30)
        out["counters_from"] = "profiles/" + fname + " (rocprofv3 --pmc passes of this workload, scaled per ray to this launch)"
        out["counters_build"] = per_launch.get("build")
        out["stale_counters"] = bool(build) and per_launch.get("build") != build
        if out["stale_counters"]:
            for k in ("frac", "frac_of_measured_peak", "frac_of_simple_op_rate", "frac_at_measured_clock", "frac_useful", "achieved", "achieved_useful"):
                if k in out:
                    out[k + "_stale_counters"], out[k] = out[k], None
            out["note_stale"] = ("the committed counters were taken on build %r, this run times %r: the fractions above are withheld (kept under "
                                 "*_stale_counters for orientation); retake with tools/profile_all.sh" % (per_launch.get("build"), build))
    else:
        out["note_counters"] = "no committed rocprofv3 counters for this workload (profiles/r*_pmc_traffic.json): fractions unavailable"
    # the reference's scan reads 16 B x N spheres per ray (list) -- what the kernel would have to stream if
    # it performed that scan. It does not (MFMA prefilter + tile culling, or the internal tree), so this is NOT a
    # fraction of anything the hardware did; kept only so rounds can be compared in the survey's unit.
    if not use_bvh:
        eq = 16.0 * n_hitables * rays_launch / ksec / 1e9
        out["algorithmic_equiv"] = {"scan_bytes_per_ray": 16.0 * n_hitables, "scan_equiv_gbs": eq,
                                    "note": "reference-scan equivalent rate (SURVEY 8d), LDS/matrix-core served; not a roofline fraction"}
    out["note"] = ("the path is bound by VALU issue + dependency/divergence stalls, not by HBM or MFMA (SURVEY 8d): frac = "
                   "SQ_INSTS_VALU x 64 lanes / kernel time against 78.6 T lane-ops/s (256 CU x 4 SIMD x 2.4 GHz / 2 cycles) is issue-slot "
                   "occupancy; frac_useful = frac x valu_lane_utilisation counts only the lanes that were switched on; "
                   "hbm_frac = measured FETCH_SIZE(x2 gfx950 correction)+WRITE_SIZE bytes / kernel time against 8 TB/s")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--preset", default="random_spheres")
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--samples", type=int, default=0, help="samples per pixel (default: 64 on one GPU = BASELINE config 3, 256 in tiles mode = config 4)")
    ap.add_argument("--mode", choices=["tiles", "frames"], default="tiles", help="how N > 1 GPUs are used (see above)")
    ap.add_argument("--depth", type=int, default=10)
    ap.add_argument("--bvh", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="N = 1: skip the extra measurement of overlapped independent frames")
    ap.add_argument("--no-extras", action="store_true", help="skip every extra measurement (host-buffer contract, other multi-GPU mode)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: gather on the render stream through pt_render_sharded (no double buffering)")
    ap.add_argument("--cpu-secs", type=float, default=15.0)
    args = ap.parse_args()

    # ---- self-launch: `python bench.py --gpus N` without a launcher (nothing has touched the GPU yet)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("PT_BENCH_FORCE_DIST") != "1":
        port = os.environ.get("MASTER_PORT", str(29500 + os.getpid() % 2000))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        print("[bench] --gpus %d without a launcher: starting  %s" % (args.gpus, " ".join(cmd)), file=sys.stderr)
        raise SystemExit(subprocess.call(cmd))

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    N = args.gpus
    if world != N and world > 1:
        N = world
    W, H, depth = args.width, args.height, args.depth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path is HIP-only (no CPU fallback)")
    # Test hooks (tests/test_gpu_parity.py drives THIS script's N > 1 path with two processes on a one-GPU box): every rank on device 0, the
    # launcher's process group over gloo (real RCCL refuses two ranks on one device), the C ABI's RCCL pinned to the cross-process
    # test double through PTGPU_RCCL_LIBRARY. None of it is set in a real run.
    if os.environ.get("PT_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("PT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if backend == "nccl" else torch.device("cpu")   # where the few scalars the ranks reduce among themselves live
    dist = None
    force_dist = os.environ.get("PT_BENCH_FORCE_DIST") == "1"   # exercise the sharded path with one rank (testing)
    if world > 1 or force_dist:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    multi = N > 1 or dist is not None
    S = args.samples if args.samples > 0 else (256 if (multi and args.mode == "tiles") else 64)

    if not os.path.exists(os.path.join(ROOT, "pathtrace-rs_amd", "_build", "libpthost.so")):
        if local_rank == 0:   # a checkout without the in-tree build: compile it once (never a fallback path)
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
    cam = hs.camera
    # The headline is the offline.rs contract: ONE Scene::update of a view the library has not seen (offline.rs:27 "only ever
    # processing 1 frame"). The library orders the work of a repeated view by the rays its last frame measured per tile;
    # every timed step here must instead pay for its own measuring launch, so that reuse is switched off (variant bit
    # 8192). The default behaviour for repeated frames of one view is reported beside it as `progressive_view`.
    base_variant = int(os.environ.get("PTGPU_VARIANT", "0"))
    scene.set_tuning(0, base_variant | 8192)

    def params_for(spp):
        return ptgpu.PtParams(W, H, spp, depth, 0, 1 if args.bvh else 0)

    stream = torch.cuda.current_stream()
    comm = None
    if multi:
        # the RCCL communicator of the C ABI: rank 0's ncclUniqueId travels over the launcher's process group
        box = [ptgpu.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        comm = ptgpu.Comm.create(box[0], rank, N, local_rank)
    rccl_info = None
    if comm is not None:
        try:
            ver, path = ptgpu.comm_runtime()
            rccl_info = {"version_code": ver, "library": path, "note": "resolved at run time by the C ABI (pt_comm_runtime): the copy this process had already loaded"}
        except Exception as e:   # (cannot happen once a communicator exists)
            rccl_info = {"error": str(e)}
    launched_as = ("python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr %s --master-port %s bench.py %s"
                   % (world, os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ.get("MASTER_PORT", "?"), " ".join(sys.argv[1:]))) if world > 1 else "python bench.py " + " ".join(sys.argv[1:])
    max_rows = sharding.padded_rows(H, N)
    # Three buffer sets: the collective of step k runs on its own HIP stream while the kernels of steps k + 1 and
    # k + 2 render into the other sets. The persistent grid holds every CU, so the exchange of step k actually runs
    # when the workgroups of step k + 1 retire; with only two sets step k + 2 would have to wait for it.
    overlap = multi and not args.no_overlap
    comm_stream = torch.cuda.Stream(device=dev) if overlap else stream
    sets = [dict(full=torch.zeros((H, W, 3), dtype=torch.float32, device=dev),
                 shard=torch.zeros((max_rows, W, 3), dtype=torch.float32, device=dev),
                 rays=torch.zeros(1, dtype=torch.int64, device=dev),
                 frames=None, done=None) for _ in range(3 if overlap else 1)]
    state = {"k": 0, "last": sets[0], "frame": None}

    def step(mode, spp):
        p = params_for(spp)
        b = sets[state["k"] % len(sets)]
        state["k"] += 1
        state["last"] = b
        if overlap and b["done"] is not None:
            stream.wait_event(b["done"])      # the collective that read this set two steps ago has finished
        if not multi:
            b["full"].zero_()  # frame 0 of a fresh accumulation (offline.rs:25 starts from zeros)
            scene.update_device(p, cam, 0, b["full"].data_ptr(), b["rays"].data_ptr(), stream.cuda_stream)
            state["frame"] = b["full"]
            return
        if mode == "tiles" and not overlap:
            # the whole sharded Scene::update in one C-ABI call on one stream: pack, render, RCCL gather, unpack
            b["full"].zero_()
            scene.update_sharded(comm, p, cam, 0, b["full"].data_ptr(), b["rays"].data_ptr(), -1, stream.cuda_stream)
            state["frame"] = b["full"]
            return
        if mode == "frames":
            b["full"].zero_()
            scene.update_device(p, cam, rank, b["full"].data_ptr(), b["rays"].data_ptr(), stream.cuda_stream)
        else:
            b["shard"].zero_()
            scene.update_shard_device(p, cam, 0, rank, N, b["shard"].data_ptr(), b["rays"].data_ptr(), stream.cuda_stream)
        if overlap:
            ready = torch.cuda.Event()
            ready.record(stream)
            comm_stream.wait_event(ready)
        with torch.cuda.stream(comm_stream):
            if mode == "frames":
                # ONE all_gather of the N frames, blend replayed in frame order (scene.rs:113-116)
                if b["frames"] is None:
                    b["frames"] = torch.empty((N, H, W, 3), dtype=torch.float32, device=dev)
                state["frame"] = sharding.gather_progressive(dist, b["full"], b["frames"], b["rays"])
            else:
                # ONE ncclAllGather of the row shards + 8-byte ncclAllReduce (scene.rs:118-120) + de-interleave, C ABI
                comm.gather_frame(W, H, b["shard"].data_ptr(), b["full"].data_ptr(), b["rays"].data_ptr(), -1, comm_stream.cuda_stream)
                state["frame"] = b["full"]
            if overlap:
                b["done"] = torch.cuda.Event()
                b["done"].record(comm_stream)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(mode, spp, steps, warmup):
        """W untimed + K timed steps bracketed by barrier + synchronize; MAX over ranks."""
        kms_list, pms_list = [], []
        for _ in range(warmup):
            step(mode, spp)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(mode, spp)
            # kernel duration of this step from the HIP events recorded on the launch stream (pt_last_kernel_ms
            # synchronises on the stop event only). In the overlapped multi-GPU pipeline the host must not wait per
            # step (the next kernel is enqueued behind the running one): there the last step's duration is read.
            if not overlap:
                kms_list.append(scene.last_kernel_ms())
                pms_list.append(scene.last_pass_ms())
        fence()
        el = time.perf_counter() - t0
        if overlap:
            kms_list.append(scene.last_kernel_ms())
            pms_list.append(scene.last_pass_ms())
        t = torch.tensor([el, sum(kms_list) / max(1, len(kms_list)), sum(pms_list) / max(1, len(pms_list))], dtype=torch.float64, device=red_dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0].item()), float(t[1].item()), float(t[2].item()), int(state["last"]["rays"].item())  # rays already summed over ranks

    self_check = None
    if multi and os.environ.get("PT_BENCH_CHECK", "1") != "0":
        # Untimed, before the warm-up, on EVERY multi-rank run: the frame the sharded / pipelined path delivers on this rank must equal
        # this rank's own unsharded render bit for bit, and the all-reduced ray count the unsharded frame's (scene.rs:90-93, 118-120).
        # The first run on a multi-GPU node is thereby also the first evidence for pt_comm.hip on real RCCL with more than one rank;
        # a mismatch on any rank ends the run with a non-zero exit code and no bench line.
        for _ in range(len(sets) + 1):       # (every buffer set of the pipeline has carried a frame)
            step(args.mode, S)
        fence()
        got = state["frame"].clone()
        got_rays = int(state["last"]["rays"].item())
        ref_full = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
        rc2 = torch.zeros(1, dtype=torch.int64, device=dev)
        if args.mode == "tiles":
            scene.update_device(params_for(S), cam, 0, ref_full.data_ptr(), rc2.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            want = ref_full
        else:
            scene.update_device(params_for(S), cam, rank, ref_full.data_ptr(), rc2.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            want = sharding.gather_progressive(dist, ref_full, torch.empty((N, H, W, 3), dtype=torch.float32, device=dev), rc2)
            torch.cuda.synchronize()
        same = bool(torch.equal(got, want)) and int(rc2.item()) == got_rays
        flag = torch.tensor([1 if same else 0], dtype=torch.int32, device=red_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if not same:
            diff = (got != want).any(dim=-1)
            print("[bench check] rank %d: %s frame over %d rank(s) DIFFERS from this rank's own unsharded frame: %d pixels, rays %d vs %d"
                  % (rank, args.mode, N, int(diff.sum().item()), got_rays, int(rc2.item())), file=sys.stderr)
        if int(flag.item()) != 1:
            if comm is not None:
                comm.close()
            dist.barrier()
            dist.destroy_process_group()
            raise SystemExit(3)
        self_check = {"sharded_equals_single": True, "ranks_seen": comm.world if comm is not None else N, "rays": got_rays,
                      "rccl": rccl_info.get("version_code") if rccl_info else None,
                      "note": "untimed, before the warm-up: the gathered frame on every rank == that rank's own unsharded render, bit for bit, "
                              "and the all-reduced ray count == the unsharded frame's; a mismatch on any rank exits non-zero"}
        if rank == 0:
            print("[bench check] %s frame over %d rank(s) == single-GPU frame on every rank, %d rays" % (args.mode, N, got_rays), file=sys.stderr)
        state["k"] = 0

    elapsed, kms, pms, rays_per_step = timed(args.mode, S, args.steps, args.warmup)
    total_rays = rays_per_step * args.steps
    value = total_rays / 1e6 / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    rays_this_launch = int(sets[0]["rays"].item()) if not multi else rays_per_step / N   # tiles: ~1/N of the frame's rays per rank

    other = None
    if multi and not args.no_extras:   # the other decomposition, reported beside the main value
        k2 = max(2, min(args.steps, 5))
        m2, s2 = ("frames", 64) if args.mode == "tiles" else ("tiles", 256)
        el2, kms2, _, rays2 = timed(m2, s2, k2, 1)
        other = {"value": rays2 * k2 / 1e6 / el2, "unit": "Mrays/s", "ms_per_step": el2 / k2 * 1e3, "kernel_ms": kms2, "steps": k2,
                 "scaling": "weak" if m2 == "frames" else "strong",
                 "workload": ("progressive frames 0..%d of the %dx%d %dspp frame, one per GPU, all_gather + blend in frame order "
                              "(bit-identical to -F %d)" % (N - 1, W, H, s2, N)) if m2 == "frames" else
                             ("ONE %dx%d %dspp frame, rows interleaved over %d GPUs, RCCL gather of the shards" % (W, H, s2, N))}

    pipelined = None
    host_buffer = None
    progressive = None
    if not multi and not args.no_extras and not hs.is_world:
        # Extra, never `value`: the preview-window pattern (glium_window.rs: Scene::update(frame_num = 0, 1, 2, ...) blending
        # into ONE buffer). From the second frame of a view on, the work order comes from the rays each tile took in the frame
        # before (measured by the frame kernel itself), not from a measuring launch of its own. Same pixels either way.
        kf = max(4, args.steps)
        pbuf = {v: torch.zeros((H, W, 3), dtype=torch.float32, device=dev) for v in ("reuse", "pilot")}
        prc = torch.zeros(1, dtype=torch.int64, device=dev)
        res = {}
        for name, var in (("pilot", base_variant | 8192), ("reuse", base_variant)):
            scene.set_tuning(0, var)
            pbuf[name].zero_()
            scene.update_device(params_for(S), cam, 0, pbuf[name].data_ptr(), prc.data_ptr(), stream.cuda_stream)   # frame 0: untimed
            torch.cuda.synchronize()
            acc = torch.zeros(1, dtype=torch.int64, device=dev)
            acc += prc               # (loads torch's add kernel outside the timer)
            acc.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for f in range(1, kf + 1):
                scene.update_device(params_for(S), cam, f, pbuf[name].data_ptr(), prc.data_ptr(), stream.cuda_stream)
                acc += prc           # (enqueued behind the frame on the same stream: no host wait between frames)
            torch.cuda.synchronize()
            res[name] = (int(acc.item()), time.perf_counter() - t0)
        scene.set_tuning(0, base_variant | 8192)
        assert torch.equal(pbuf["reuse"], pbuf["pilot"]) and res["reuse"][0] == res["pilot"][0], "work order changed a pixel"
        progressive = {"value": res["reuse"][0] / 1e6 / res["reuse"][1], "unit": "Mrays/s", "ms_per_frame": res["reuse"][1] / kf * 1e3,
                       "frames": kf, "measuring_every_frame_anew": res["pilot"][0] / 1e6 / res["pilot"][1],
                       "note": "Scene::update(frame_num = 1..%d) of one view accumulating into one buffer (the preview-window loop); "
                               "work ordered by the previous frame's measured rays per tile instead of a measuring launch of its own; "
                               "accumulated image bit-identical to the run that measures every frame anew" % kf}
    if not multi and not args.no_extras:
        # The reference's own contract -- Scene::update on a HOST buffer (offline.rs:27-34 times exactly this call), PCIe
        # inclusive. Reported at top level as `host_contract`; `value` stays the device-resident rate (the harness contract:
        # inputs resident in HBM when the timed region starts).
        hb = np.zeros((H, W, 3), np.float32)
        kh = max(3, min(args.steps, 10))

        def host_steps(zero_first):
            scene.update(params_for(S), cam, 0, hb)
            th, rays_h, parts = 0.0, 0, np.zeros(4)
            for _ in range(kh):
                if zero_first:
                    hb[:] = 0.0          # offline.rs:25 hands Scene::update a freshly zeroed Vec (untimed here, as its allocation is there)
                t0 = time.perf_counter()
                rays_h = scene.update(params_for(S), cam, 0, hb)
                th += time.perf_counter() - t0
                parts += np.array(scene.last_host_ms())
            assert rays_h == rays_per_step and np.array_equal(hb, state["frame"].cpu().numpy())
            return {"value": rays_h * kh / 1e6 / th, "unit": "Mrays/s", "ms_per_step": th / kh * 1e3, "steps": kh}, parts / kh

        host_buffer, parts = host_steps(True)
        host_buffer["host_ms"] = {"scan_under_measuring_launch": float(parts[0]), "gpu_wait": float(parts[1]), "copy_out": float(parts[2]), "whole_call": float(parts[3])}
        host_buffer["note"] = ("pt_render(frame 0) on a zeroed PAGEABLE host buffer, read + written, PCIe inclusive: the call offline.rs:27-34 times. "
                               "The kernels render into a pinned + mapped copy over PCIe (no D2H phase); the scan that finds the buffer all +0.0f "
                               "(so nothing is uploaded) runs under the measuring launch; helper threads copy the frame back; frame identical to the device-resident one")
        reused, parts_r = host_steps(False)
        reused["host_ms"] = {"copy_in_under_measuring_launch": float(parts_r[0]), "gpu_wait": float(parts_r[1]), "copy_out": float(parts_r[2]), "whole_call": float(parts_r[3])}
        reused["note"] = "same call on a buffer that still holds a frame (the preview window's loop): the previous frame is copied into the pinned copy under the measuring launch"
        host_buffer["reused_buffer"] = reused
        ptgpu.buffer_register(hb)      # a host that keeps its Vec alive registers it once: pt_render then renders in place over PCIe
        try:
            host_buffer["registered"] = host_steps(True)[0]
            host_buffer["registered"]["note"] = ("same call on a buffer pinned + mapped by pt_buffer_register: previous frame read and "
                                                 "new frame written pixel by pixel under the kernel, no copies at all")
        finally:
            ptgpu.buffer_unregister(hb)
    if not multi and not args.no_pipeline and not args.no_extras:
        # Extra figure, never `value`: independent frames back to back on two scene handles / two HIP streams, so the
        # tail of frame k (its last, serial pixels) and the measuring launch of frame k + 1 overlap. A single frame cannot
        # use this; a renderer producing a sequence of independent frames (animation, tiles of a bigger image) can.
        hs2 = pthost.HostScene(args.preset, W, H, samples=S, use_bvh=args.bvh, device=local_rank)
        hs2.device_scene().set_tuning(0, base_variant | 8192)
        handles = [(scene, stream, torch.zeros((H, W, 3), dtype=torch.float32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)),
                   (hs2.device_scene(), torch.cuda.Stream(device=dev), torch.zeros((H, W, 3), dtype=torch.float32, device=dev),
                    torch.zeros(1, dtype=torch.int64, device=dev))]

        def pstep(k):
            sc, st, buf, rc = handles[k % 2]
            with torch.cuda.stream(st):
                buf.zero_()
                sc.update_device(params_for(S), cam, 0, buf.data_ptr(), rc.data_ptr(), st.cuda_stream)

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

    other_workloads = None
    if not multi and not args.no_extras and args.preset == "random_spheres" and (W, H, S) == (1200, 800, 64) and not args.bvh:
        # Extra, never `value`: the other kernels of the path at the headline's frame size (a BVH world on the list kernel, the
        # general-world kernel on the reference's box / light presets), three frames of a never-seen view each after one warm-up.
        other_workloads = {}
        for name, preset, bvh in (("random_spheres -B", "random_spheres", True), ("random", "random", False), ("simple_light", "simple_light", False),
                                  ("cornell", "cornell", False), ("cornell_smoke", "cornell_smoke", False), ("cornell_smoke -B", "cornell_smoke", True)):
            ho = pthost.HostScene(preset, W, H, samples=S, use_bvh=bvh, device=local_rank)
            so = ho.device_scene()
            so.set_tuning(0, base_variant | 8192)
            po = ptgpu.PtParams(W, H, S, depth, 0, 1 if bvh else 0)
            bo = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
            ro = torch.zeros(1, dtype=torch.int64, device=dev)
            so.update_device(po, ho.camera, 0, bo.data_ptr(), ro.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                bo.zero_()
                so.update_device(po, ho.camera, 0, bo.data_ptr(), ro.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            elo = (time.perf_counter() - t0) / 3.0
            other_workloads[name] = {"value": int(ro.item()) / 1e6 / elo, "unit": "Mrays/s", "ms_per_frame": elo * 1e3, "kernel": so.last_kernel_choice()["name"]}
            del so, ho
    baseline_configs = None
    if not multi and not args.no_extras and args.preset == "random_spheres" and (W, H, S) == (1200, 800, 64) and not args.bvh:
        # Extra, never `value`: the other BASELINE.json configurations on this one GPU, each at ITS frame size -- config 2 (aras
        # 1280x720x16), config 4's frame (random_spheres 1200x800x256; the multi-GPU run shards it) and config 5 (perlin_spheres,
        # 10 002 spheres, BVH, 1920x1080x128): one warm-up + three timed frames of a never-seen view each (bit 8192), the frame kernel's
        # own time from HIP events, and the same counter-derived sub-block as the headline where a committed profile matches.
        baseline_configs = {}
        for name, preset, bw, bh, bs, bvh in (("config 2: aras 1280x720 16spp", "aras", 1280, 720, 16, False),
                                              ("config 4 on one GPU: random_spheres 1200x800 256spp", "random_spheres", 1200, 800, 256, False),
                                              ("config 5: perlin_spheres 1920x1080 128spp BVH", "perlin_spheres", 1920, 1080, 128, True)):
            ho = pthost.HostScene(preset, bw, bh, samples=bs, use_bvh=bvh, device=local_rank)
            so = ho.device_scene()
            so.set_tuning(0, base_variant | 8192)
            po = ptgpu.PtParams(bw, bh, bs, depth, 0, 1 if bvh else 0)
            bo = torch.zeros((bh, bw, 3), dtype=torch.float32, device=dev)
            ro = torch.zeros(1, dtype=torch.int64, device=dev)
            so.update_device(po, ho.camera, 0, bo.data_ptr(), ro.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            nf, kms_o = 3, []
            t0 = time.perf_counter()
            for _ in range(nf):
                bo.zero_()
                so.update_device(po, ho.camera, 0, bo.data_ptr(), ro.data_ptr(), stream.cuda_stream)
                kms_o.append(so.last_kernel_ms())     # (waits for the frame: these frames are timed one by one)
            torch.cuda.synchronize()
            elo = (time.perf_counter() - t0) / nf
            rays_o = int(ro.item())
            ent = {"value": rays_o / 1e6 / elo, "unit": "Mrays/s", "ms_per_frame": elo * 1e3, "frames": nf, "rays_per_frame": rays_o,
                   "kernel": so.last_kernel_choice()["name"], "hitables": ho.world_desc.n_hitables}
            cnt = committed_counters(preset, bw, bh, bs, bvh)
            if cnt is not None:
                ent["roofline"] = roofline_block("pt_trace_kernel", sum(kms_o) / nf, float(rays_o), ho.world_desc.n_hitables, bvh, cnt, build=ptgpu.lib().pt_version().decode())
            baseline_configs[name] = ent
            del so, ho, bo
    if rank == 0:
        grid, block, lds = scene.last_launch_info()
        tiles = multi and args.mode == "tiles"
        counters = committed_counters(args.preset, W, H, S, args.bvh)
        build = ptgpu.lib().pt_version().decode()
        roof = roofline_block("pt_trace_kernel" if not hs.is_world else "pt_world_kernel", kms, float(rays_this_launch), n_spheres, args.bvh, counters, build=build)
        roof["pass_ms"] = pms
        roof["note_pass"] = "kernel_ms = the frame kernel alone (what rocprofv3 reports; samples 2..S); pass_ms adds the measuring launch (first sample of every other 8x8 tile) and the tile sort that precede it; traffic = the frame kernel's HBM bytes, which include reading back the 48 B per pixel (RNG stream + colour sum) the measuring launch parked"
        is_headline = args.preset == "random_spheres" and (W, H) == (1200, 800) and not args.bvh and ((not multi and S == 64) or (tiles and S == 256))
        out = {
            "metric": ("Mrays/sec, random_spheres 1200x800 %dspp" % S) if is_headline else "Mrays/sec, %s %dx%d %dspp" % (args.preset, W, H, S),
            "value": value, "unit": "Mrays/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if tiles else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "build": build,   # pt_version(): the source hash of the library that was timed (profiles record the same string)
            "config": {"workload": "preset %s (%d hitables) %dx%d %dspp depth %d %s, seed 0, %s"
                                   % (args.preset, n_spheres, W, H, S, depth, "BVH" if args.bvh else "list",
                                      "frame 0" if not multi else ("frame 0 split by rows over %d GPUs (BASELINE config 4)" % N if tiles
                                                                   else "progressive frames 0..%d, one per GPU" % (N - 1))),
                       "rays_per_step": rays_per_step, "wall_secs_per_step": ms_per_step / 1e3,
                       "parallelism": ("1 GPU" if not multi else
                                       ("rows y %% %d == rank on each GPU, no collective while rendering, then ONE ncclAllGather of the "
                                        "float3 shards + 8-byte ncclAllReduce of the ray count through the C ABI (pt_comm_gather_frame)" % N if tiles
                                        else "frame_num = rank on each of %d GPUs, no data-path collective, all_gather of the frames + "
                                             "blend in frame order (scene.rs:113-116)" % N)),
                       "overlap": ("collective of step k on a second HIP stream under the kernel of step k + 1" if overlap else "none"),
                       "ranks_seen": (comm.world if comm is not None else 1),
                       "rccl": rccl_info,
                       "launched_as": launched_as,
                       "grid": grid, "block": block, "lds_bytes": lds,
                       "work_order": "every timed step measures its own tile costs (launch 1: first sample of every other 8x8 tile; launch 2: the "
                                     "rest, expensive tiles first); the library's reuse of the previous frame's measured costs for a "
                                     "repeated view is switched off for `value` (pt_scene_set_tuning bit 8192) and reported as "
                                     "`progressive_view`"},
            "roofline": roof,
        }
        if other is not None:
            out["weak_scaling_frames" if args.mode == "tiles" else "strong_scaling_tiles"] = other
        if host_buffer is not None:
            out["host_contract"] = host_buffer
            out["value_note"] = ("`value` = frames rendered into an HBM-resident buffer (bench contract: inputs resident when the timed region starts); "
                                 "`host_contract.value` = the same frame through pt_render on a pageable host buffer, the call the reference times "
                                 "(offline.rs:27-34, SURVEY 8d), PCIe and host copies included")
        if pipelined is not None:
            out["pipelined_frames"] = pipelined
        if progressive is not None:
            out["progressive_view"] = progressive
        if other_workloads is not None:
            out["other_workloads_same_frame_size"] = other_workloads
        if baseline_configs is not None:
            out["baseline_configs"] = baseline_configs
        if self_check is not None:
            out["self_check"] = self_check
            out["config"]["sharded_equals_single"] = True
        if N == 1 and not multi and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.preset, W, H, S, depth, args.bvh, args.cpu_secs)
        print(json.dumps(out))
    if comm is not None:
        torch.cuda.synchronize()
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
