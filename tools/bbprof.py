#!/usr/bin/env python3
"""Development aid: dynamic instruction profile of the MFMA list kernels by assembly instrumentation.

No PC sampling or thread trace is available on the pool, and the kernels' time tracks their VALU instruction count
(DESIGN.md section 4), so the question "where do the instructions go" is answered by counting basic-block executions:

  python3 tools/bbprof.py build      # here (no GPU): pt_kernels_list.hip -> assembly -> one scalar atomic per basic
                                     # block -> pathtrace-rs_amd/_build/bbprof/libptgpu.so + bbprof_map.json
  gpurun -- python3 tools/bbprof.py run [--preset P ...]    # GPU box: renders frames, dumps gpurun_out/bbprof_counts.txt
  python3 tools/bbprof.py report     # here: counts x static per-block instruction mix, by kernel and by source line

Every block also adds popcount(exec) to a second counter, so the report knows how many of the 64 lanes were switched on when the
block ran: blocks are ranked by MASKED lane-instructions (VALU instructions x lanes that were off) as well, and the kernel's lane
utilisation comes out as rocprof's VALUUtilization does (sum of active lanes over sum of issued lanes).

The instrumented kernels reserve s[92:99] (the build caps the compiler at 92 SGPRs), so their register allocation differs a
little from the shipped ones: use the result for proportions, not absolute counts.
"""
import collections, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pathtrace-rs_amd")
OUT = os.path.join(PKG, "_build", "bbprof")
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-I../include", "-Icsrc",
         "-I/opt/rocm/include", "-DPT_BBPROF"]
UNIT = os.environ.get("BBPROF_UNIT", "pt_kernels_list")   # pt_kernels_list | pt_kernels_gate | pt_kernels_tree | pt_kernels_world
if UNIT in ("pt_kernels_list", "pt_kernels_tree"):   # (as pathtrace-rs_amd/Makefile builds these two units)
    FLAGS += ["-mllvm", "-amdgpu-use-amdgpu-trackers"]
# BBPROF_DEFS="-DPT_SECTIONS": the kernel then reads the cycle counter (s_memtime) at its section boundaries, in source order; the
# instrumenter numbers every instruction by how many of those reads precede it in the assembly, and `report` adds up instructions
# per SECTION -- to be set beside the cycle shares a -DPT_SECTIONS library prints (cold blocks the compiler moved to the end of the
# function land in the last section: they are cold)
FLAGS += os.environ.get("BBPROF_DEFS", "").split()


def sh(cmd, **kw):
    r = subprocess.run(cmd, cwd=PKG, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, **kw)
    if r.returncode:
        sys.exit("FAILED: " + " ".join(cmd) + "\n" + r.stdout[-3000:])
    return r.stdout


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")): return "vmem"
    return "other"


def instrument(src_lines):
    """One counter per basic block of every pt_trace_kernel instance; returns (patched lines, map)."""
    files, out, blocks = {}, [], []
    kernel, cur, loc = None, None, None
    marks = 0   # cycle-counter reads seen so far in this kernel (section boundaries of a -DPT_SECTIONS build)

    def open_block(name):
        nonlocal cur
        cur = {"kernel": kernel, "label": name, "valu": 0, "salu": 0, "lds": 0, "mfma": 0, "vmem": 0, "other": 0, "trans": 0, "lanes": 0, "movs": 0, "lines": collections.Counter(),
               "sec_valu": collections.Counter(), "sec_all": collections.Counter()}
        blocks.append(cur)
        out.append("\ts_atomic_add_x2 s[96:97], s[98:99], 0x%x" % (8 * (len(blocks) - 1)))
        # active lanes of this execution into the second half of the counter array. s_bcnt1 writes SCC, which may be live across
        # the block's entry (a compare in the block above): saved and restored. The wait keeps the data pair stable under the atomic.
        out.extend(["\ts_cselect_b32 s94, 1, 0", "\ts_bcnt1_i32_b64 s92, exec", "\ts_mov_b32 s93, 0",
                    "\ts_atomic_add_x2 s[92:93], s[98:99], 0x%x" % (8 * (len(blocks) - 1) + 8 * 32768), "\ts_waitcnt lgkmcnt(0)", "\ts_cmp_lg_u32 s94, 0"])

    for l in src_lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
        if m: files[int(m.group(1))] = m.group(2)
        m = re.match(r"^(_ZN5ptdev\d+pt_(?:trace|world)_kernel\w+):", l)
        if m:
            kernel = m.group(1)
            marks = 0
            out.append(l)
            out += ["\ts_mov_b64 s[96:97], 1", "\ts_getpc_b64 s[98:99]", "\ts_add_u32 s98, s98, pt_bbprof@gotpcrel32@lo+4", "\ts_addc_u32 s99, s99, pt_bbprof@gotpcrel32@hi+12",
                    "\ts_load_dwordx2 s[98:99], s[98:99], 0x0", "\ts_waitcnt lgkmcnt(0)"]
            open_block("entry")
            continue
        if kernel and re.match(r"^\.Lfunc_end", l):
            kernel, cur = None, None
        if kernel:
            m = re.match(r"^(\.LBB[0-9_]+):", l) or re.match(r"^; (%bb\.\d+):", l)   # (fall-through blocks carry no label)
            if m:
                out.append(l)
                open_block(m.group(1))
                continue
            if ".amdhsa_next_free_sgpr" in l:
                l = "\t\t.amdhsa_next_free_sgpr 100"
            m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
            if m: loc = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            # an instruction that writes exec (the restore at a join, a saveexec whose region was too short for a skip branch) changes
            # how many lanes the REST of the block runs with: a new counted segment starts behind it
            if cur is not None and re.match(r"^\s+s_\w+\s+exec\b", l) or (cur is not None and re.match(r"^\s+s_\w*saveexec_b64\s", l)):
                cur["salu"] += 1
                out.append(l)
                open_block(cur["label"].rstrip("+") + "+")
                continue
            m = re.match(r"^\s+([a-z_0-9]+)(\s|$)", l)
            if m and cur is not None and not m.group(1).startswith("."):
                op = m.group(1)
                k = classify(op)
                cur[k] += 1
                if op == "s_memtime": marks += 1
                cur["sec_all"][str(marks)] += 1
                if k == "valu": cur["sec_valu"][str(marks)] += 1
                if k == "valu":
                    if re.match(r"v_(rcp|sqrt|rsq|div_|exp|log|sin|cos)", op): cur["trans"] += 1
                    if re.match(r"v_(readlane|writelane|readfirstlane)", op): cur["lanes"] += 1   # SGPR spill traffic and wave-uniform reads
                    if re.match(r"v_(mov_b|accvgpr)", op): cur["movs"] += 1
                    if loc: cur["lines"]["%s:%d" % loc] += 1
        elif ".amdhsa_next_free_sgpr" in l and blocks and ("pt_trace_kernel" in (blocks[-1]["kernel"] or "") or "pt_world_kernel" in (blocks[-1]["kernel"] or "")):
            l = "\t\t.amdhsa_next_free_sgpr 100"   # the descriptor follows the function body
        out.append(l)
    for b in blocks: b["lines"], b["sec_valu"], b["sec_all"] = dict(b["lines"]), dict(b["sec_valu"]), dict(b["sec_all"])
    return out, blocks


def build():
    os.makedirs(OUT, exist_ok=True)
    s_file = os.path.join(OUT, UNIT + ".s")
    sh(["/opt/rocm/bin/hipcc"] + FLAGS + ["--cuda-device-only", "-S", "-gline-tables-only", "csrc/%s.hip" % UNIT, "-o", s_file])
    patched, blocks = instrument(open(s_file).read().split("\n"))
    p_file = os.path.join(OUT, UNIT + "_bb.s")
    open(p_file, "w").write("\n".join(patched))
    json.dump(blocks, open(os.path.join(OUT, "bbprof_map.json"), "w"))
    if len(blocks) > 32768: sys.exit("more blocks than counters")
    sh([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", p_file, "-o", os.path.join(OUT, "dev.o")])
    sh([LLVM + "/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", os.path.join(OUT, "dev.out"), os.path.join(OUT, "dev.o")])
    sh([LLVM + "/clang-offload-bundler", "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
        "-input=/dev/null", "-input=" + os.path.join(OUT, "dev.out"), "-output=" + os.path.join(OUT, "dev.hipfb")])
    sh(["/opt/rocm/bin/hipcc"] + FLAGS + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", os.path.join(OUT, "dev.hipfb"),
                                         "-c", "csrc/%s.hip" % UNIT, "-o", os.path.join(OUT, UNIT + ".o")])
    objs = []
    for f in sorted(os.listdir(os.path.join(PKG, "_build"))):
        if f.endswith(".o") and f != UNIT + ".o" and f.startswith("pt_"):
            objs.append(os.path.join(PKG, "_build", f))
    objs.append(os.path.join(OUT, UNIT + ".o"))
    sh(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", os.path.join(OUT, "libptgpu.so"), "-Wl,-soname,libptgpu.so", "-ldl", "-lpthread"])
    print("built", os.path.join(OUT, "libptgpu.so"), "with", len(blocks), "counted blocks")


def run(argv):
    """Renders frames of one workload (default: BASELINE config 3) on the instrumented library and dumps the counters."""
    import argparse, ctypes, importlib.util
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="random_spheres")
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--bvh", action="store_true")
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "bbprof_counts.txt"))
    a = ap.parse_args(argv)

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m
    import torch
    ptgpu = load("pathtrace_rs_amd_ptgpu", "pathtrace-rs_amd/ptgpu.py")
    ptgpu.LIB_PATH = os.path.join(OUT, "libptgpu.so")   # its soname is libptgpu.so: libpthost's dependency resolves to this copy
    ctypes.CDLL(ptgpu.LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    pthost = load("pathtrace_rs_amd_pthost", "pathtrace-rs_amd/pthost.py")
    hs = pthost.HostScene(a.preset, a.width, a.height, samples=a.samples, use_bvh=a.bvh, device=0)
    scene = hs.device_scene()
    scene.set_tuning(0, 8192)   # every frame pays its own measuring launch, as the bench's headline does
    frame = torch.zeros((a.height, a.width, 3), dtype=torch.float32, device="cuda:0")
    rays = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    p = ptgpu.PtParams(a.width, a.height, a.samples, 10, 0, 1 if a.bvh else 0)
    st = torch.cuda.current_stream()
    total = 0
    for _ in range(a.frames):
        frame.zero_()
        scene.update_device(p, hs.camera, 0, frame.data_ptr(), rays.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        total += int(rays.item())
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    rc = ptgpu.lib().pt_bbprof_dump(a.out.encode())
    open(a.out, "a").write("rays %d\n" % total)
    print("dump rc", rc, "rays", total, "kernel", scene.last_kernel_choice() if hasattr(scene, "last_kernel_choice") else "")


def remap():
    """Re-derives bbprof_map.json from the kept assembly (after a change to the classification above)."""
    _, blocks = instrument(open(os.path.join(OUT, UNIT + ".s")).read().split("\n"))
    json.dump(blocks, open(os.path.join(OUT, "bbprof_map.json"), "w"))


def section_of_line_table():
    """Which section of the MFMA list kernel's main loop a source line belongs to, where that is unambiguous: the kernel body's own lines
    (between its PT_SEC markers) and the functions only one section calls. Lines of shared helpers (pt_device.h: RNG, square roots,
    normalisation ...) are attributed per BASIC BLOCK, by the majority of the block's unambiguous lines."""
    srcs = {f: open(os.path.join(PKG, "csrc", f)).read().split("\n") for f in ("pt_kernel.h", "pt_prefilter.h")}
    def find(f, text, start=0):
        for i in range(start, len(srcs[f])):
            if text in srcs[f][i]: return i + 1
        raise KeyError(text)
    K, P = "pt_kernel.h", "pt_prefilter.h"
    feat0 = find(P, "__device__ __forceinline__ RayFeat make_ray_features")
    feat1 = find(P, "// Candidate queue without atomics")
    clip0 = find(P, "struct TileClip {")
    mf0 = find(P, "__device__ __forceinline__ int intersect_list_mfma")
    drain0, drain1 = find(P, "auto drain = [&]() {", mf0), find(P, "    // the always-tested spheres first", mf0)
    sub6 = find(P, "PT_SUB(6);", mf0)
    refill0 = find(K, "// ---- refill:")
    sec0, sec1, sec2, sec3 = find(K, "PT_SEC(0);", refill0), find(K, "PT_SEC(1);", refill0), find(K, "PT_SEC(2);", refill0), find(K, "PT_SEC(3);", refill0)
    fold0 = find(K, "if (terminal) {", sec2)
    end = find(K, "#undef PT_DEPTH", sec3)
    tiles, phase2 = "tiles: always-tested spheres, masks, MFMA loop", "phase 2: balanced exact tests"
    table = [(K, refill0, sec0, "refill"), (K, sec0 + 1, sec1, "camera + rejection loop"), (K, sec1 + 1, sec2, tiles), (K, sec2 + 1, fold0 - 1, "shade"),
             (K, fold0, sec3, "fold + sample end"), (K, sec3 + 1, end, "hand-over + worker (pt_coop.h)"),
             (P, feat0, feat1 - 1, "ray features"), (P, clip0, mf0 - 1, tiles), (P, mf0, drain0 - 1, tiles), (P, drain0, drain1 - 1, phase2), (P, drain1, sub6, tiles),
             (P, sub6 + 1, len(srcs[P]), phase2), ("pt_coop.h", 1, 100000, "hand-over + worker (pt_coop.h)")]
    def lookup(loc):
        f, _, ln = loc.rpartition(":")
        f, ln = os.path.basename(f), int(ln)
        for tf, a, b, sec in table:
            if f == tf and a <= ln <= b: return sec
        return None
    return lookup


def report(argv):
    counts_file = argv[0] if argv else os.path.join(ROOT, "gpurun_out", "bbprof_counts.txt")
    blocks = json.load(open(os.path.join(OUT, "bbprof_map.json")))
    counts, lanes_on, rays = {}, {}, 0
    for l in open(counts_file):
        a, b = l.split()
        if a == "rays": rays = int(b)
        elif int(a) >= 32768: lanes_on[int(a) - 32768] = int(b)
        else: counts[int(a)] = int(b)
    per_kernel = collections.defaultdict(lambda: collections.Counter())
    lines = collections.defaultdict(lambda: collections.Counter())
    rows = collections.defaultdict(list)
    for i, b in enumerate(blocks):
        n = counts.get(i, 0)
        if not n: continue
        k = b["kernel"]
        for c in ("valu", "salu", "lds", "mfma", "vmem", "trans", "lanes", "movs"): per_kernel[k][c] += n * b.get(c, 0)
        per_kernel[k]["blocks"] += n
        for sidx, c in b.get("sec_valu", {}).items():
            per_kernel[k]["sec_valu_" + sidx] += n * c
            per_kernel[k]["sec_lane_on_" + sidx] += lanes_on.get(i, 64 * n) * c
        for sidx, c in b.get("sec_all", {}).items(): per_kernel[k]["sec_all_" + sidx] += n * c
        if b["label"] == "entry": per_kernel[k]["waves"] += n
        for ln, c in b["lines"].items(): lines[k][ln] += n * c
        on = lanes_on.get(i, 64 * n)
        per_kernel[k]["lane_issued"] += 64 * n * b["valu"]
        per_kernel[k]["lane_on"] += on * b["valu"]
        rows[k].append((n * b["valu"], n, b, on))
    for k, tot in sorted(per_kernel.items(), key=lambda kv: -kv[1]["valu"]):
        print("==", k)
        if rays: print("   VALU wave-instructions per 64 rays (all kernels' rays): %.1f" % (tot["valu"] / (rays / 64.0)))
        print("   waves %d; dynamic wave-instructions: VALU %.4g (of them v_rcp/sqrt/div_* %.3g)  SALU %.4g  LDS %.4g  MFMA %.4g  VMEM %.4g" %
              (tot["waves"], tot["valu"], tot["trans"], tot["salu"], tot["lds"], tot["mfma"], tot["vmem"]))
        print("   of the VALU instructions: v_readlane/v_writelane %.3g (%.1f%%), v_mov %.3g (%.1f%%)" % (tot["lanes"], 100.0 * tot["lanes"] / tot["valu"], tot["movs"], 100.0 * tot["movs"] / tot["valu"]))
        if tot["lane_issued"]:
            print("   VALU lane utilisation (active lanes / issued lanes, as rocprof's VALUUtilization): %.1f%%" % (100.0 * tot["lane_on"] / tot["lane_issued"]))
            print("   -- blocks by MASKED lane-instructions (VALU instructions x lanes switched off): share of all masked, lanes on of 64, share of all VALU")
            masked_total = tot["lane_issued"] - tot["lane_on"]
            for dv, n, b, on in sorted(rows[k], key=lambda r: -(64 * r[1] - r[3]) * r[2]["valu"])[:40]:
                top = sorted(b["lines"].items(), key=lambda kv: -kv[1])[:4]
                print("   %5.2f%%  %-11s lanes on %4.1f  valu share %5.2f%%  runs %.3g x valu %d  %s" % (
                    100.0 * (64 * n - on) * b["valu"] / max(masked_total, 1), b["label"], on / max(n, 1), 100.0 * dv / tot["valu"], n, b["valu"],
                    " ".join("%s(%d)" % (a.replace("pt_kernel.h", "k").replace("pt_device.h", "d").replace("pt_coop.h", "c").replace("__clang_hip_math.h", "m"), c) for a, c in top)))
        if UNIT == "pt_kernels_list" and "pt_trace_kernel" in k:
            # sections by SOURCE: a block belongs to the section most of its unambiguous lines belong to
            lookup = section_of_line_table()
            secs_all = [key for key in tot if key.startswith("sec_valu_")]
            by_sec, on_sec, all_sec = collections.Counter(), collections.Counter(), collections.Counter()
            for dv, n, b, on in rows[k]:
                votes = collections.Counter()
                for loc, c in b["lines"].items():
                    sec = lookup(loc)
                    if sec: votes[sec] += c
                if votes:
                    sec = votes.most_common(1)[0][0]
                else:
                    # only shared helpers (pt_device.h: RNG steps, square roots, sines ...): where the block LIES in the assembly, by the
                    # cycle-counter reads of a -DPT_SECTIONS build that precede it (BBPROF_DEFS; hot helper blocks stay in line)
                    marks = b.get("sec_valu", {})
                    idx = int(max(marks.items(), key=lambda kv: kv[1])[0]) if marks else -1
                    sec = {1: "refill", 2: "camera + rejection loop", 3: "ray features", 4: "ray features", 5: "tiles: always-tested spheres, masks, MFMA loop",
                           6: "phase 2: balanced exact tests", 7: "shade", 8: "shade", 9: "hand-over + worker (pt_coop.h)", 10: "hand-over + worker (pt_coop.h)"}.get(idx, "(unattributed)") if len(secs_all) > 1 else "(shared helpers only: unattributed)"
                by_sec[sec] += dv
                on_sec[sec] += on * b["valu"]
                all_sec[sec] += n * (b["valu"] + b["salu"] + b["lds"] + b["mfma"] + b["vmem"])
            alli = sum(all_sec.values())
            print("   -- instructions by section of the main loop (a basic block counts for the section most of its own source lines lie in)")
            for sec, v in by_sec.most_common():
                print("   %-50s valu %5.2f%%  lanes on %4.1f  all instructions %5.2f%%" % (sec, 100.0 * v / tot["valu"], on_sec[sec] / max(v, 1), 100.0 * all_sec[sec] / max(alli, 1)))
        secs = sorted((int(key[len("sec_valu_"):]) for key in tot if key.startswith("sec_valu_")))
        if len(secs) > 1:
            print("   -- instructions by SECTION (index = cycle-counter reads that precede the instruction in the assembly): share of VALU, lanes on, share of all instructions")
            alli = sum(v for key, v in tot.items() if key.startswith("sec_all_"))
            for sidx in secs:
                v = tot["sec_valu_%d" % sidx]
                print("   section %2d  valu %5.2f%%  lanes on %4.1f  all instructions %5.2f%%" % (sidx, 100.0 * v / tot["valu"], tot["sec_lane_on_%d" % sidx] / max(v, 1), 100.0 * tot["sec_all_%d" % sidx] / max(alli, 1)))
        print("   -- blocks by dynamic VALU")
        for dv, n, b, _on in sorted(rows[k], key=lambda r: -r[0])[:45]:
            top = sorted(b["lines"].items(), key=lambda kv: -kv[1])[:5]
            print("   %5.2f%%  %-11s runs %.3g x (valu %d salu %d lds %d mfma %d)  %s" % (100.0 * dv / tot["valu"], b["label"], n, b["valu"], b["salu"], b["lds"], b["mfma"],
                                                                                    " ".join("%s(%d)" % (a.replace("pt_kernel.h", "k").replace("pt_device.h", "d").replace("__clang_hip_math.h", "m"), c) for a, c in top)))
        print("   -- blocks by dynamic v_readlane / v_writelane / v_readfirstlane (SGPR spill traffic and wave-uniform reads)")
        for dl, n, b in sorted(((r[1] * r[2].get("lanes", 0), r[1], r[2]) for r in rows[k]), key=lambda r: -r[0])[:25]:
            if not dl: break
            top = sorted(b["lines"].items(), key=lambda kv: -kv[1])[:4]
            print("   %5.2f%%  %-11s runs %.3g x %d of valu %d  %s" % (100.0 * dl / max(tot["lanes"], 1), b["label"], n, b["lanes"], b["valu"],
                                                                  " ".join("%s(%d)" % (a.replace("pt_kernel.h", "k").replace("pt_device.h", "d").replace("pt_coop.h", "c"), c) for a, c in top)))
        print("   -- blocks by dynamic v_mov_b32 / v_accvgpr moves (register shuffling the allocator added)")
        for dm, n, b in sorted(((r[1] * r[2].get("movs", 0), r[1], r[2]) for r in rows[k]), key=lambda r: -r[0])[:25]:
            if not dm: break
            top = sorted(b["lines"].items(), key=lambda kv: -kv[1])[:4]
            print("   %5.2f%%  %-11s runs %.3g x %d of valu %d  %s" % (100.0 * dm / max(tot["movs"], 1), b["label"], n, b["movs"], b["valu"],
                                                                  " ".join("%s(%d)" % (a.replace("pt_kernel.h", "k").replace("pt_device.h", "d").replace("pt_coop.h", "c"), c) for a, c in top)))
        print("   -- source lines by dynamic VALU (innermost inlined location)")
        for ln, c in lines[k].most_common(60):
            print("   %5.2f%%  %s" % (100.0 * c / tot["valu"], ln))


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else ""
    if cmd == "build": build()
    elif cmd == "run": run(sys.argv[2:])
    elif cmd == "report": report(sys.argv[2:])
    elif cmd == "remap": remap()
    else: sys.exit(__doc__)
