// pt_kernels.h -- the kernel instantiations live in four translation units (they dominate the build time and compile in
// parallel); each exposes a lookup by the selection's coordinates. pt_launch.hip only ever sees function pointers.
#pragma once
#include "pt_host.h"

namespace pthostside {

// MFMA list kernels: pt_trace_kernel<false, true, true, VERIFY, PILOT, MOVING, GATE, BLK>; blk in {256, 768, 1024}; pool: the 1024-thread frame kernel with per-wave pixel pools
void mfma_list_kernels(bool moving, uint32_t blk, bool verify, bool pool, SphereKernel *frame, SphereKernel *measure);   // GATE = false (pt_kernels_list.hip)
void mfma_gate_kernels(bool moving, uint32_t blk, bool verify, bool pool, SphereKernel *frame, SphereKernel *measure);   // GATE = true  (pt_kernels_gate.hip)
// tree kernels pt_trace_kernel<true, TREE4, false, VERIFY, PILOT, MOVING> and the exact-scan kernels (pt_kernels_tree.hip)
void tree_kernels(bool tree4, bool moving, bool verify, bool grid, SphereKernel *frame, SphereKernel *measure);
void scan_kernels(bool sph_lds, SphereKernel *frame, SphereKernel *measure);
// pt_world_kernel<BVH, HIT_LDS, OCC, MEDIA, CHAINS> (pt_kernels_world.hip)
WorldKernel world_kernel(bool bvh, bool hit_lds, uint32_t occ, bool media, bool chains, bool lazy, bool graph);

}  // namespace pthostside
