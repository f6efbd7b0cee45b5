// pt_build.h -- interface between the C ABI (ptgpu.hip) and the device tree builder (pt_build.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "pt_tree4.h"

namespace ptdev {

// Builds the 4-wide internal tree over `n` items (host array, n >= 2) on the current device: *d_nodes_out receives a
// hipMalloc'ed array of *n_nodes_out nodes (root = node 0, level order), *depth_out the number of node levels and *ms_out
// the device time of the build (HIP events around everything after the upload of the items). Returns 0 or a hipError_t.
int tree4_build_device(const TreeItem *h_items, uint32_t n, hipStream_t stream, DNode4 **d_nodes_out, uint32_t *n_nodes_out, uint32_t *depth_out,
                       float *ms_out);

// Packs the finished tree into the 64-byte nodes the kernels read (pt_tree4.h DNode4Q) and builds the leaves' slot records
// from the per-sphere arrays already on the device (`d_leafrec` of a BVH world, else `d_spheres`). *ok_out = false when
// some node cannot be packed (the caller then walks the binary tree). Returns 0 or a hipError_t.
int tree4_pack_device(const DNode4 *d_nodes, uint32_t n_nodes, const float4 *d_spheres, const float4 *d_leafrec, hipStream_t stream,
                      DNode4Q **d_packed_out, float4 **d_slotrec_out, bool *ok_out);

}  // namespace ptdev
