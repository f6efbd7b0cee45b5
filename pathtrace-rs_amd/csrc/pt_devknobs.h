// pt_devknobs.h -- the DEVELOPMENT switches of the library: the bits of pt_scene_set_tuning's variant word beyond the two an integrator
// may set (include/ptgpu.h: PT_TUNE_MEASURE_EVERY_FRAME, PT_TUNE_NO_HANDOVER), and the PTGPU_* environment knobs of -DPT_DEVKNOBS builds.
// They exist for A/B measurements and for the parity tests (every variant renders the SAME frame: tests/test_gpu_parity.py
// test_scan_variants_agree and friends); their meaning may change between versions, which is why they are documented here and not in the
// public header.
#pragma once
#include <stdint.h>

namespace ptsel {

// Tuning word (pt_scene_set_tuning): the bits that select code paths. All variants render identical frames.
enum : uint32_t {
    kVarScanFromHbm = 1u,        // scan table / binary-tree nodes from HBM/L2 instead of LDS
    kVarStackInHbm = 2u,         // attenuation stack in HBM (MFMA kernels: 3 x 256 threads per CU instead of one wide workgroup)
    kVarExactScan = 4u,          // exact VALU scan instead of the MFMA prefilter
    kVarVerify = 8u,             // verify mode: audits prefilter + culling (pt_scene_verify_counters) / counts tree work (pt_scene_bvh_counters)
    kVarNoStack = 16u,           // (timing experiment)
    kVarNaturalOrder = 32u,      // no heavy-first work order
    kVarNoListTree = 64u,        // list worlds never walk the internal tree (forces the scan)
    kVarGeneralMoving = 128u,    // Sphere + MovingSphere worlds on the general kernel instead of the MOVING sphere kernels
    kVarBvhOnTree = 256u,        // use_bvh worlds always walk the internal tree (default: the MFMA list kernel + ancestor gate when it fits)
    kVarNoCulling = 1024u,       // MFMA kernels run every sphere tile for every wave
    kVarBinaryTree = 2048u,      // tree kernels walk the binary internal tree (host-built) instead of the 4-wide one
    kVarMeasureEveryFrame = 8192u,   // = PT_TUNE_MEASURE_EVERY_FRAME: no reuse of the previous frame's measured tile costs
    kVarNoCoop = 65536u,         // = PT_TUNE_NO_HANDOVER: wide list kernels hand no pixels over to idle waves (pt_coop.h)
    kVarWorldEager = 131072u,    // general-world kernel: Noise colours (texture.rs:86-88) where the surface is hit, every lane its own. Default for
                                 // worlds with Noise textures: a Lambertian / Isotropic scatter keeps the hit POINT and the colour is formed only
                                 // when the path ends on something that is not black, by the whole wave (pt_world.h LAZY); a path that ends in black
                                 // multiplies each of its finite attenuations by zero (scene.rs:62-64): 0 + a * 0 = 0, the same bits
    kVarMeasureAllTiles = 262144u,   // MFMA list kernels: the measuring launch of a new view traces EVERY tile. Default: one colour of a checkerboard of
                                     // 8x8 tiles; a tile of the other colour takes the mean of its measured neighbours as its cost and starts at its
                                     // first sample in the second launch (measuring launch + order of config 3: 0.30 -> 0.23 ms)
    kVarNoGrid = 524288u,        // tree kernels always walk the 4-wide tree (default: the uniform cell grid of pt_grid.h when the scene has one)
    kVarNoPool = 1048576u,       // wide list frame kernels: no per-wave pool of ready pixels in LDS; freed lanes wait for a batched refill (rounds 2-5)
    kVarNoPark = 2097152u,       // cell-grid kernels: every call walks all its rays to their end (default: the last few lanes still walking park their walk
                                 // in LDS and finish it in the wave's next call, pt_grid.h)
    // (4096, 16384 and 32768 were A/B switches of questions settled in rounds 2-3 and are ignored)
};

}  // namespace ptsel

// Environment knobs, read ONLY by builds made with `make DEFS=-DPT_DEVKNOBS` (pt_api.hip dev_knobs(); the shipped library reads one
// variable, PTGPU_HOST_BUILD, the hook with which the parity tests compare the two tree builders):
//   PTGPU_REFILL / PTGPU_PHASE1_REFILL   lanes that must want a pixel before the wave refills (frame / measuring launch)
//   PTGPU_POOL / PTGPU_POOL_TAIL         wide list kernels: entries of a wave's pixel pool (0: off) / fair share of the list (items left per wave) below which claims stop filling it
//   PTGPU_PARK_MAX / PTGPU_PARK_AFTER    cell-grid kernels: lanes still walking that park their walk (0: never) / rounds of a call before they may
//   PTGPU_READY                          4-wide tree: lanes without traversal work before subtrees change hands (kShareMin)
//   PTGPU_DRAIN                          4-wide tree: queued leaf candidates of one lane that trigger the wave's drain
//   PTGPU_COOP_LIVE / _STREAK / _PERIOD / _EST / PTGPU_COOP_DBG     hand-over policy of the wide list kernels (pt_coop.h)
//   PTGPU_WORLD_OCC3 / PTGPU_WORLD_OCC4  general-world kernel: force three / four waves per SIMD
//   PTGPU_CLAMP_GRID, PTGPU_TIMING, PTGPU_DEBUG                     launch geometry / per-wave timing output / selection trace
