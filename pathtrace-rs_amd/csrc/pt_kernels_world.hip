// pt_kernels_world.hip -- the general-world kernel's instantiations: <BVH, HIT_LDS, OCC, MEDIA, CHAINS, LAZY>.
#include "pt_kernels.h"
#include "pt_world.h"

namespace pthostside {

WorldKernel world_kernel(bool bvh, bool hit_lds, uint32_t occ, bool media, bool chains, bool lazy, bool graph) {
    if (graph) return pt_world_kernel<false, false, 3, true, true, false, true>;   // an interpreted scene graph (pt_graph.h)
    // worlds with Noise textures (pt_select.h: records in LDS, three waves per SIMD, no chains)
    if (lazy)
        return bvh ? (media ? pt_world_kernel<true, true, 3, true, false, true> : pt_world_kernel<true, true, 3, false, false, true>)
                   : (media ? pt_world_kernel<false, true, 3, true, false, true> : pt_world_kernel<false, true, 3, false, false, true>);
    // worlds with Instance chains (scene graphs) take the most general build: three waves per SIMD, MEDIA code present
    if (chains)
        return bvh ? (hit_lds ? pt_world_kernel<true, true, 3, true, true> : pt_world_kernel<true, false, 3, true, true>)
                   : (hit_lds ? pt_world_kernel<false, true, 3, true, true> : pt_world_kernel<false, false, 3, true, true>);
    if (occ == 5u)   // five waves per SIMD (96 VGPRs): pt_select.h picks it for worlds whose records sit in LDS
        return bvh ? (media ? pt_world_kernel<true, true, 5, true> : pt_world_kernel<true, true, 5, false>) : (media ? pt_world_kernel<false, true, 5, true> : pt_world_kernel<false, true, 5, false>);
    // worlds whose records do not fit LDS share the MEDIA = true code
    if (occ == 4u)
        return bvh ? (hit_lds ? (media ? pt_world_kernel<true, true, 4, true> : pt_world_kernel<true, true, 4, false>) : pt_world_kernel<true, false, 4, true>)
                   : (hit_lds ? (media ? pt_world_kernel<false, true, 4, true> : pt_world_kernel<false, true, 4, false>) : pt_world_kernel<false, false, 4, true>);
    return bvh ? (hit_lds ? (media ? pt_world_kernel<true, true, 3, true> : pt_world_kernel<true, true, 3, false>) : pt_world_kernel<true, false, 3, true>)
               : (hit_lds ? (media ? pt_world_kernel<false, true, 3, true> : pt_world_kernel<false, true, 3, false>) : pt_world_kernel<false, false, 3, true>);
}

}  // namespace pthostside
