// pt_kernels_tree.hip -- tree kernels (4-wide packed tree / binary tree) and the exact-scan kernels.
#include "pt_kernel.h"
#include "pt_kernels.h"

namespace pthostside {

// the 4-wide tree kernels' flavour that walks the uniform cell grid (pt_grid.h): [moving][frame, measure, verify]
static void grid_kernels(bool moving, bool verify, SphereKernel *frame, SphereKernel *measure) {
    static const SphereKernel table[2][3] = {
        {pt_trace_kernel<true, true, false, false, false, false, false, kBlock, true>, pt_trace_kernel<true, true, false, false, true, false, false, kBlock, true>,
         pt_trace_kernel<true, true, false, true, false, false, false, kBlock, true>},
        {pt_trace_kernel<true, true, false, false, false, true, false, kBlock, true>, pt_trace_kernel<true, true, false, false, true, true, false, kBlock, true>,
         pt_trace_kernel<true, true, false, true, false, true, false, kBlock, true>}};
    const SphereKernel *t = table[moving ? 1 : 0];
    *frame = verify ? t[2] : t[0];
    *measure = verify ? nullptr : t[1];
}

void tree_kernels(bool tree4, bool moving, bool verify, bool grid, SphereKernel *frame, SphereKernel *measure) {
    if (grid) return grid_kernels(moving, verify, frame, measure);
    // [tree4][moving][frame, measure, verify] (SPH_LDS = true selects the 4-wide tree); verify counts node fetches / sphere tests
    static const SphereKernel table[2][2][3] = {
        {{pt_trace_kernel<true, false, false, false, false, false>, pt_trace_kernel<true, false, false, false, true, false>, pt_trace_kernel<true, false, false, true, false, false>},
         {pt_trace_kernel<true, false, false, false, false, true>, pt_trace_kernel<true, false, false, false, true, true>, pt_trace_kernel<true, false, false, true, false, true>}},
        {{pt_trace_kernel<true, true, false, false, false, false>, pt_trace_kernel<true, true, false, false, true, false>, pt_trace_kernel<true, true, false, true, false, false>},
         {pt_trace_kernel<true, true, false, false, false, true>, pt_trace_kernel<true, true, false, false, true, true>, pt_trace_kernel<true, true, false, true, false, true>}}};
    const SphereKernel *t = table[tree4 ? 1 : 0][moving ? 1 : 0];
    *frame = verify ? t[2] : t[0];
    *measure = verify ? nullptr : t[1];
}

void scan_kernels(bool sph_lds, SphereKernel *frame, SphereKernel *measure) {
    if (sph_lds) {
        *frame = pt_trace_kernel<false, true, false, false, false>;
        *measure = pt_trace_kernel<false, true, false, false, true>;
    } else {
        *frame = pt_trace_kernel<false, false, false, false, false>;
        *measure = nullptr;
    }
}

SphereKernel tree4_kernel_for_registers(bool moving) {
    return moving ? pt_trace_kernel<true, true, false, false, false, true> : pt_trace_kernel<true, true, false, false, false, false>;
}

}  // namespace pthostside
