#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (tools/profile.sh) into profiles/<tag>_*.{txt,json}."""
import collections
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def is_frame_kernel(name):
    """The frame kernel (5th template flag PILOT = false); the 1-spp pilot pass is a separate symbol."""
    m = re.search(r"pt_(?:trace|world)_kernel<([^>]*)>", name)
    if not m:
        return "pt_trace_kernel" in name or "pt_world_kernel" in name
    flags = [f.strip() for f in m.group(1).split(",")]
    return len(flags) < 5 or flags[4] == "false"

def is_measuring_kernel(name):
    """The measuring launch of a sphere-kernel frame (5th template flag PILOT = true): the first sample of every pixel (MFMA list kernels: of every other tile)."""
    m = re.search(r"pt_trace_kernel<([^>]*)>", name)
    if not m:
        return False
    flags = [f.strip() for f in m.group(1).split(",")]
    return len(flags) >= 5 and flags[4] == "true"


src = os.path.join("gpurun_out", "prof_" + tag)
os.makedirs("profiles", exist_ok=True)
lines = []


def find(pattern):
    hits = sorted(glob.glob(os.path.join(src, pattern), recursive=True))
    return hits[0] if hits else None


# bench frames under the profiler (steps + warm-up) -- the frame kernel is dispatched once per frame, or TWICE for general
# worlds (their two launches of a new view share one symbol), plus pt_scene_prepare's throw-away frame before them
def frames_of(name):
    f = os.path.join(src, name)
    if os.path.exists(f) and os.path.getsize(f):
        l = json.load(open(f))
        return int(l.get("steps", 0)) + int(l.get("warmup", 0)), int(l.get("steps", 0))
    return None, None


n_frames_t, n_steps_t = frames_of("bench_line.json")          # the timing pass (--kernel-trace --stats)
n_frames_p, _ = frames_of("bench_line_pmc.json")              # the counter passes (fewer frames: counters per launch do not depend on clocks)
if n_frames_p is None:
    n_frames_p = n_frames_t
n_frames = n_frames_p


def bench_groups(items, n_frames=None):
    """items: the frame kernel's dispatches in order. Returns them grouped per bench frame (prepare's dispatches dropped)."""
    n_frames = n_frames or n_frames_p
    if not n_frames or len(items) < n_frames:
        return [[x] for x in items[1:]] if len(items) > 1 else [[x] for x in items]
    k = max(1, len(items) // (n_frames + 1)) if len(items) > n_frames else 1
    tail = items[len(items) - n_frames * k:]
    return [tail[i * k:(i + 1) * k] for i in range(n_frames)]


# 1. kernel stats
ks = find("stats/**/*kernel_stats.csv")
kt = find("stats/**/*kernel_trace.csv")
lines.append("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps %s --warmup %s --no-cpu-baseline" % (n_steps_t, (n_frames_t or 0) - (n_steps_t or 0)))
avg_ms = None
if ks:
    for r in csv.DictReader(open(ks)):
        lines.append("%-70s calls %s total_ns %s avg_ns %s pct %s" % (r["Name"][:70], r["Calls"], r["TotalDurationNs"],
                                                                    r["AverageNs"], r["Percentage"]))
        if is_frame_kernel(r["Name"]):
            avg_ms = float(r["AverageNs"]) / 1e6
if kt:
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(kt))
            if is_frame_kernel(r["Kernel_Name"])]
    lines.append("pt_trace_kernel dispatch durations (ms): " + ", ".join("%.3f" % d for d in durs))
    if len(durs) > 1:
        # pt_scene_prepare's throw-away frame (part of Scene::new) comes first and is not a bench step; a bench frame of a
        # general world is two dispatches of this symbol (first sample of every pixel, then the rest): summed per frame
        groups = bench_groups(durs, n_frames_t)
        timed = groups[-n_steps_t:] if n_steps_t and len(groups) >= n_steps_t else groups   # the frames bench.py times (its warm-up frames dropped)
        avg_ms = sum(sum(g) for g in timed) / len(timed)
        lines.append("frame kernel, average over the %d TIMED bench frames (%d dispatch(es) each; pt_scene_prepare's frame and the %d warm-up frames excluded): %.3f ms" % (
            len(timed), len(timed[0]), len(groups) - len(timed), avg_ms))
    rows = [r for r in csv.DictReader(open(kt)) if is_frame_kernel(r["Kernel_Name"])]
    if rows:
        r = rows[-1]
        lines.append("launch (as rocprofv3 reports it; the LDS is dynamic, see lds_bytes in the bench line): grid %s threads, "
                     "workgroup %s, VGPR %s accVGPR %s SGPR %s scratch %s" % (
            r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", r.get("Workgroup_Size")), r.get("VGPR_Count"),
            r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("Scratch_Size")))

# 2. PMC passes (sum over dispatches of the trace kernel, then per launch)
pmc = {}
for d in ("pmc_sq", "pmc_lds", "pmc_fetch", "pmc_write"):
    f = find(d + "/**/*counter_collection.csv")
    if not f:
        continue
    rows = [r for r in csv.DictReader(open(f)) if is_frame_kernel(r["Kernel_Name"])]
    # per bench FRAME: pt_scene_prepare's throw-away frame dropped, the dispatches of one frame summed
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    groups = bench_groups(ids)
    keep = {i for g in groups for i in g}
    agg = collections.defaultdict(float)
    for r in rows:
        if int(r["Dispatch_Id"]) in keep:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
    n = max(1, len(groups))
    for k, v in agg.items():
        pmc[k] = v / n
# HBM traffic of the measuring launch (its own symbol for the sphere kernels), per bench frame
pmc_m = {}
for d in ("pmc_fetch", "pmc_write"):
    f = find(d + "/**/*counter_collection.csv")
    if not f:
        continue
    rows = [r for r in csv.DictReader(open(f)) if is_measuring_kernel(r["Kernel_Name"])]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    keep = set(bench_groups(ids)[g][0] for g in range(len(bench_groups(ids)))) if ids else set()
    keep = {i for g in bench_groups(ids) for i in g} if ids else set()
    agg = collections.defaultdict(float)
    for r in rows:
        if int(r["Dispatch_Id"]) in keep:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        pmc_m[k] = v / max(1, len(bench_groups(ids)))
lines.append("")
lines.append("# PMC counters of the frame kernel per bench frame (separate --pmc passes; pt_scene_prepare's frame excluded)")
for k in sorted(pmc):
    lines.append("%-32s %.6g" % (k, pmc[k]))
out = {"tag": tag, "kernel_avg_ms": avg_ms, "pmc_per_launch": pmc}
if "FETCH_SIZE" in pmc or "WRITE_SIZE" in pmc:
    # MI355X_MICROARCH.md "HBM": counters are in KiB; FETCH_SIZE under-reports wide coalesced reads by 2x on
    # gfx950, so the read side is doubled before it is compared with byte counts. WRITE_SIZE is uncalibrated.
    fetch = pmc.get("FETCH_SIZE", 0.0) * 1024 * 2
    write = pmc.get("WRITE_SIZE", 0.0) * 1024
    out["hbm_bytes_per_launch"] = fetch + write
    out["hbm_read_bytes_corrected"] = fetch
    out["hbm_write_bytes"] = write
    lines.append("HBM traffic per launch: read %.3f MB (FETCH_SIZE KiB x 1024 x 2 gfx950 correction), write %.3f MB"
                 % (fetch / 1e6, write / 1e6))
if pmc_m:
    m_bytes = pmc_m.get("FETCH_SIZE", 0.0) * 1024 * 2 + pmc_m.get("WRITE_SIZE", 0.0) * 1024
    out["hbm_bytes_measuring_launch"] = m_bytes
    lines.append("HBM traffic of the measuring launch (first sample of the measured pixels; parks 48 B per pixel): %.3f MB; both launches of a frame: %.3f MB"
                 % (m_bytes / 1e6, (m_bytes + out.get("hbm_bytes_per_launch", 0.0)) / 1e6))
bl = os.path.join(src, "bench_line.json")
if os.path.exists(bl) and os.path.getsize(bl):
    out["bench_line_under_profiler"] = json.load(open(bl))
    out["build"] = out["bench_line_under_profiler"].get("build")   # pt_version() of the library these counters were taken on: bench.py marks them stale for any other build
    lines.append("")
    lines.append("# bench.py line under the profiler: value %.1f %s, kernel_ms %.3f" % (
        out["bench_line_under_profiler"]["value"], out["bench_line_under_profiler"]["unit"],
        out["bench_line_under_profiler"]["roofline"]["kernel_ms"]))
open(os.path.join("profiles", tag + "_rocprof_summary.txt"), "w").write("\n".join(lines) + "\n")
json.dump(out, open(os.path.join("profiles", tag + "_pmc_traffic.json"), "w"), indent=1)
print("\n".join(lines))
