#!/bin/bash
# ASan + UBSan over the CPU code (GPU sanitizers are not available on the pool): the oracle renders every preset
# (list and BVH, plus the flat-description round trip), the C++ host builds every preset's description.
# Usage (repo root, needs pathtrace-rs_amd/_build/libptgpu.so for linking): bash tools/sanitize_cpu.sh
set -e
OUT=${TMPDIR:-/tmp}/pt_sanitize
mkdir -p "$OUT"
gcc -O1 -g -std=c11 -D_GNU_SOURCE -fPIC -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -shared \
    -o "$OUT/libptref.so" oracle/ptref.c -lm -lpthread
g++ -O1 -g -std=c++17 -ffp-contract=off -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude -Ipathtrace-rs_amd/host \
    -shared pathtrace-rs_amd/host/scene.cpp pathtrace-rs_amd/host/presets.cpp pathtrace-rs_amd/host/offline.cpp \
    pathtrace-rs_amd/host/capi.cpp -o "$OUT/libpthost.so" -Lpathtrace-rs_amd/_build -lptgpu -Wl,-rpath,"$PWD/pathtrace-rs_amd/_build"
cat > "$OUT/run.py" <<PY
import importlib, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_binding as ob
L = ob.lib("$OUT/libptref.so")
pthost = importlib.import_module("pathtrace-rs_amd.pthost")
pthost.LIB_PATH = "$OUT/libpthost.so"
for name in ["small", "random_spheres", "two_perlin_spheres", "aras", "random", "simple_light", "cornell", "cornell_smoke", "smallpt", "final"]:
    for bvh in (False, True):
        if name == "final" and bvh:
            continue          # throws by design (params.rs:37 unwraps None)
        sc = ob.OracleScene(name, 40, 30, use_bvh=bvh, library=L)
        ex = sc.export()
        sc.update(2, nthreads=2)
        b = ob.OracleScene.from_world(ex["hitables"], ex["transforms"], ex["materials"], ex["textures"], ex["camera"], 40, 30,
                                      sky=ex["sky"], use_bvh=bvh, library=L)
        b.update(1, nthreads=1)
        h = pthost.HostScene(name, 64, 48, use_bvh=bvh)
        h.export()
print("sanitize_cpu: clean")
PY
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so)" \
    ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 python "$OUT/run.py"

# ---- the LIBRARY's host side: every translation unit of libptgpu.so compiled host-only (kernels become launch stubs that are never
# called) with ASan + UBSan, linked with tools/fuzz_desc.cpp, and fed seeded malformed descriptions through pt_debug_select -- the
# validators, the scene-graph flattener, the tree restatements and the kernel selection, none of which touch a device.
#   FUZZ_CASES (default 100000), FUZZ_SEED (default 1)
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SANFLAGS="--offload-host-only -O1 -g -std=c++17 -ffp-contract=off -fPIC -Iinclude -Ipathtrace-rs_amd/csrc -I/opt/rocm/include -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
objs=""
for u in pt_api pt_prep pt_scene pt_launch pt_render pt_comm pt_query pt_build pt_kernels_list pt_kernels_gate pt_kernels_tree pt_kernels_world; do
    if [ ! -f "$OUT/$u.o" ] || [ -n "$(find pathtrace-rs_amd/csrc include -newer "$OUT/$u.o" -type f | head -1)" ]; then
        $HIPCC $SANFLAGS -c pathtrace-rs_amd/csrc/$u.hip -o "$OUT/$u.o" &
    fi
    objs="$objs $OUT/$u.o"
done
wait
# (a host-only object still refers to its device image, __hip_fatbin_<hash>: every such symbol gets one small valid image, so that the
#  HIP runtime's registration at load time sees what it expects; no kernel of it is ever launched)
printf '#include <hip/hip_runtime.h>\n__global__ void pt_san_dummy() {}\n' > "$OUT/tiny.hip"
$HIPCC --offload-arch=gfx950 -c "$OUT/tiny.hip" -o "$OUT/tiny.o"
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin "$OUT/tiny.o" "$OUT/tiny.fatbin"
: > "$OUT/fatbins.s"
for s in $(nm -u $objs | grep -o "__hip_fatbin_[0-9a-f]*" | sort -u); do
    printf '.globl %s\n.section .hip_fatbin,"a",@progbits\n.p2align 12\n%s:\n.incbin "%s/tiny.fatbin"\n' $s $s "$OUT" >> "$OUT/fatbins.s"
done
/opt/rocm/lib/llvm/bin/clang -c "$OUT/fatbins.s" -o "$OUT/fatbins.o"
$HIPCC $SANFLAGS -x hip -c tools/fuzz_desc.cpp -o "$OUT/fuzz_desc.o"
$HIPCC --offload-host-only -fsanitize=address,undefined $objs "$OUT/fuzz_desc.o" "$OUT/fatbins.o" -o "$OUT/fuzz_desc" -ldl -lpthread
ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 "$OUT/fuzz_desc" ${FUZZ_CASES:-100000} ${FUZZ_SEED:-1}
