// pt_kernels_gate.hip -- MFMA list kernels of BVH worlds (GATE = true: ancestor-AABB gate + DFS-rank ties at hit acceptance).
#include "pt_kernel.h"
#include "pt_kernels.h"

namespace pthostside {

void mfma_gate_kernels(bool moving, uint32_t blk, bool verify, bool pool, SphereKernel *frame, SphereKernel *measure) {
    static const SphereKernel table[2][8] = {
        {pt_trace_kernel<false, true, true, false, false, false, true>, pt_trace_kernel<false, true, true, false, true, false, true>,
         pt_trace_kernel<false, true, true, true, false, false, true>,
         pt_trace_kernel<false, true, true, false, false, false, true, 768>, pt_trace_kernel<false, true, true, false, true, false, true, 768>,
         pt_trace_kernel<false, true, true, false, false, false, true, 1024>, pt_trace_kernel<false, true, true, false, true, false, true, 1024>,
         pt_trace_kernel<false, true, true, false, false, false, true, 1024, false, true>},
        {pt_trace_kernel<false, true, true, false, false, true, true>, pt_trace_kernel<false, true, true, false, true, true, true>,
         pt_trace_kernel<false, true, true, true, false, true, true>,
         pt_trace_kernel<false, true, true, false, false, true, true, 768>, pt_trace_kernel<false, true, true, false, true, true, true, 768>,
         pt_trace_kernel<false, true, true, false, false, true, true, 1024>, pt_trace_kernel<false, true, true, false, true, true, true, 1024>,
         pt_trace_kernel<false, true, true, false, false, true, true, 1024, false, true>}};
    const SphereKernel *t = table[moving ? 1 : 0];
    const int w = blk == 1024u ? 5 : (blk == 768u ? 3 : 0);
    *frame = verify ? t[2] : ((blk == 1024u && !pool) ? t[7] : t[w]);   // ([7]: the 1024-thread frame kernel without pixel pools)
    *measure = verify ? nullptr : t[w + 1];
}

}  // namespace pthostside
