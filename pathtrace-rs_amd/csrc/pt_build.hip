// pt_build.hip -- DEVICE build of the tree kernels' 4-wide internal tree ("GPU-side BVH build").
//
// Replaces, for traversal, what the reference does on the host in BVHNode::new (bvh.rs:64-94,268-333: per split a
// random axis, sort_unstable_by bbox.min[axis], pivot len/2). Build rule: pt_tree4.h. The build is level-synchronous:
// every level orders ALL of its segments at once with one global stable radix sort per phase (key = start of the range
// an item belongs to << 32 | orderable coordinate on that range's own axis; items outside any range keep their place
// because their key is their position), so the host never sorts anything -- it only reads one counter per level.
//
//   per level:  bounds (atomic min/max of the centroids per range)  ->  keys  ->  rocprim::radix_sort_pairs     (phase A: whole segments)
//               bounds  ->  keys  ->  radix_sort_pairs                                                          (phase B: the halves cut again)
//               emit: child slots of every node, node numbers of the next level (block-wide scan), next segment list
//   then, bottom-up per level: child boxes -> node records (tree_finish_node, the same code the host reference runs)
#include <hip/hip_runtime.h>

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "pt_build.h"

namespace ptdev {
namespace {

struct Seg {
    uint32_t lo, hi, node;
};

// the range position p is ordered in during `phase` (0: its whole segment, 1: the half of it that is cut again);
// false when p belongs to no range of this phase
__device__ __forceinline__ bool range_of(const Seg *segs, uint32_t n_segs, uint32_t p, int phase, uint32_t &slot, uint32_t &lo, uint32_t &hi) {
    uint32_t a = 0, b = n_segs;   // last segment with lo <= p
    while (b - a > 1) {
        const uint32_t m = (a + b) >> 1;
        if (segs[m].lo <= p) a = m; else b = m;
    }
    const Seg s = segs[a];
    if (p < s.lo || p >= s.hi) return false;
    if (phase == 0) {
        slot = 2u * a, lo = s.lo, hi = s.hi;
        return true;
    }
    const TreePlan pl = tree_plan(s.hi - s.lo);
    const uint32_t mid = s.lo + pl.cut[pl.half];
    if (p < mid) {
        slot = 2u * a, lo = s.lo, hi = mid;
        return pl.half == 2u;
    }
    slot = 2u * a + 1u, lo = mid, hi = s.hi;
    return pl.c - pl.half == 2u;
}

__global__ void bounds_kernel(const TreeItem *items, const uint32_t *perm, uint32_t n, const Seg *segs, uint32_t n_segs, int phase, uint32_t *bounds) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint32_t slot, lo, hi;
    if (!range_of(segs, n_segs, p, phase, slot, lo, hi)) return;
    const TreeItem &it = items[perm[p]];
    for (int k = 0; k < 3; ++k) {
        const uint32_t u = tree_orderable(it.c[k]);
        atomicMin(&bounds[slot * 6u + k], u);
        atomicMax(&bounds[slot * 6u + 3u + k], u);
    }
}

__device__ __forceinline__ float from_orderable(uint32_t u) {
    // (arithmetic form: the select form of this inverse makes hipcc 7.2 crash inside keys_kernel)
    return __uint_as_float(u ^ (~(uint32_t)((int32_t)u >> 31) | 0x80000000u));
}

__global__ void keys_kernel(const TreeItem *items, const uint32_t *perm, uint32_t n, const Seg *segs, uint32_t n_segs, int phase, const uint32_t *bounds,
                            unsigned long long *keys) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint32_t slot, lo, hi;
    if (!range_of(segs, n_segs, p, phase, slot, lo, hi)) {
        keys[p] = (unsigned long long)p << 32;
        return;
    }
    const float ex = from_orderable(bounds[slot * 6u + 3u]) - from_orderable(bounds[slot * 6u + 0u]);
    const float ey = from_orderable(bounds[slot * 6u + 4u]) - from_orderable(bounds[slot * 6u + 1u]);
    const float ez = from_orderable(bounds[slot * 6u + 5u]) - from_orderable(bounds[slot * 6u + 2u]);
    // coordinate on the range's axis (tree_axis_of_extents: the longest extent, the lower axis on ties)
    const TreeItem &it = items[perm[p]];
    float cv = it.c[0], best = ex;
    if (ey > best) cv = it.c[1], best = ey;
    if (ez > best) cv = it.c[2];
    keys[p] = ((unsigned long long)lo << 32) | (unsigned long long)tree_orderable(cv);
}

// One block: children of every segment of this level. Inner children (more than one sphere) take the node's first slots
// and the next free node numbers in segment order; single spheres follow as leaf slots.
__global__ void emit_kernel(const TreeItem *items, const uint32_t *perm, const Seg *segs, uint32_t n_segs, uint32_t next_base, Seg *next_segs,
                            DNode4 *nodes, uint32_t *leaf_item, uint32_t *n_next) {
    __shared__ uint32_t scan[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_segs; base += blockDim.x) {
        const uint32_t s = base + threadIdx.x;
        Seg sg{0, 0, 0};
        TreePlan pl{};
        uint32_t n_inner = 0;
        if (s < n_segs) {
            sg = segs[s];
            pl = tree_plan(sg.hi - sg.lo);
            for (uint32_t j = 0; j < pl.c; ++j) n_inner += (pl.cut[j + 1] - pl.cut[j] > 1u) ? 1u : 0u;
        }
        scan[threadIdx.x] = n_inner;
        __syncthreads();
        for (uint32_t off = 1; off < blockDim.x; off <<= 1) {   // inclusive Hillis-Steele scan
            const uint32_t v = threadIdx.x >= off ? scan[threadIdx.x - off] : 0u;
            __syncthreads();
            scan[threadIdx.x] += v;
            __syncthreads();
        }
        const uint32_t first = next_base + carry + scan[threadIdx.x] - n_inner;
        if (s < n_segs) {
            DNode4 &w = nodes[sg.node];
            uint32_t slot = 0;
            for (int pass = 0; pass < 2; ++pass)           // inner children first, then leaves, each in cut order
                for (uint32_t j = 0; j < pl.c; ++j) {
                    const uint32_t a = sg.lo + pl.cut[j], b = sg.lo + pl.cut[j + 1];
                    if ((b - a > 1u) != (pass == 0)) continue;
                    if (pass == 0) {
                        w.child[slot] = (int32_t)(first + slot);
                        leaf_item[sg.node * 4u + slot] = 0xffffffffu;
                        next_segs[first + slot - next_base] = Seg{a, b, first + slot};
                    } else {
                        const uint32_t item = perm[a];
                        w.child[slot] = ~(int32_t)items[item].sphere;
                        leaf_item[sg.node * 4u + slot] = item;
                    }
                    ++slot;
                }
            for (; slot < 4u; ++slot) w.child[slot] = kNoChild4, leaf_item[sg.node * 4u + slot] = 0xffffffffu;
        }
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) carry += scan[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_next = carry;
}

// bottom-up: the nodes [first, first + count) of one level, whose children are finished
__global__ void boxes_kernel(const TreeItem *items, const uint32_t *leaf_item, DNode4 *nodes, TreeBox *node_box, uint32_t first, uint32_t count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t node = first + i;
    DNode4 w = nodes[node];
    TreeBox ch[4];
    uint32_t c = 0;
    for (uint32_t j = 0; j < 4u; ++j) {
        if (w.child[j] == kNoChild4) break;
        if (w.child[j] >= 0) {
            ch[j] = node_box[w.child[j]];
        } else {
            const TreeItem &it = items[leaf_item[node * 4u + j]];
            for (int k = 0; k < 3; ++k) ch[j].mn[k] = it.mn[k], ch[j].mx[k] = it.mx[k];
            ch[j].rmin = it.r;
        }
        ++c;
    }
    node_box[node] = tree_finish_node(w, ch, c);
    nodes[node] = w;
}

__global__ void bounds_init_kernel(uint32_t *bounds, uint32_t n_slots) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_slots) return;
    for (int k = 0; k < 3; ++k) bounds[i * 6u + k] = 0xffffffffu, bounds[i * 6u + 3u + k] = 0u;
}

__global__ void iota_kernel(uint32_t *perm, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) perm[i] = i;
}

// packed nodes (tree_pack_node) + the slot records of the leaves: record (node * 4 + slot) = the sphere's 64-byte leaf record
// (BVH worlds: sphere | gate min | gate max | rank) or its (cx, cy, cz, r) alone (list worlds), with the sphere index in [3].y
__global__ void pack_kernel(const DNode4 *nodes, uint32_t n_nodes, const float4 *spheres, const float4 *leafrec, DNode4Q *packed, float4 *slotrec,
                            uint32_t *failed) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const DNode4 w = nodes[i];
    DNode4Q q;
    if (!tree_pack_node(w, q)) atomicOr(failed, 1u);
    packed[i] = q;
    for (uint32_t j = 0; j < 4u; ++j) {
        // A slot without a leaf must never produce a hit: its planes cannot be met while the node-level pad stays below half
        // the f16 range, but a ray starting ~10^4 units away pads every box by more than that (pt_tree.h bvh4_trace), and the
        // exact test then runs on this record. A NaN centre makes the reference discriminant NaN, which is not > 0.
        const float qnan = __uint_as_float(0x7fc00000u);
        float4 r0 = make_float4(qnan, qnan, qnan, 0.f), r1 = make_float4(0.f, 0.f, 0.f, 0.f), r2 = r1, r3 = r1;
        if (w.child[j] != kNoChild4 && w.child[j] < 0) {
            const uint32_t k = (uint32_t)~w.child[j];
            if (leafrec) r0 = leafrec[4 * k], r1 = leafrec[4 * k + 1], r2 = leafrec[4 * k + 2], r3 = leafrec[4 * k + 3];
            else r0 = spheres[k];
            r3.y = __uint_as_float(k);
        }
        float4 *out = slotrec + 4 * ((size_t)i * 4u + j);
        out[0] = r0, out[1] = r1, out[2] = r2, out[3] = r3;
    }
}

#define BUILD_TRY(expr)                     \
    do {                                    \
        const hipError_t e_ = (expr);       \
        if (e_ != hipSuccess) {             \
            rc = (int)e_;                   \
            goto done;                      \
        }                                   \
    } while (0)

}  // namespace

int tree4_build_device(const TreeItem *h_items, uint32_t n, hipStream_t stream, DNode4 **d_nodes_out, uint32_t *n_nodes_out, uint32_t *depth_out,
                       float *ms_out) {
    *d_nodes_out = nullptr, *n_nodes_out = 0, *depth_out = 0;
    if (ms_out) *ms_out = 0.f;
    if (n < 2) return 0;
    int rc = 0;
    const uint32_t max_nodes = n;            // every node has >= 2 children: at most n - 1 nodes
    const uint32_t max_segs = n / 2 + 1;
    TreeItem *d_items = nullptr;
    uint32_t *d_perm[2] = {nullptr, nullptr}, *d_bounds = nullptr, *d_leaf_item = nullptr, *d_n_next = nullptr;
    unsigned long long *d_keys[2] = {nullptr, nullptr};
    Seg *d_segs[2] = {nullptr, nullptr};
    DNode4 *d_nodes = nullptr;
    TreeBox *d_box = nullptr;
    void *d_tmp = nullptr;
    size_t tmp_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint32_t level_first[64], level_count[64], levels = 0, node_count = 1, n_segs = 1;
    const uint32_t grid_n = (n + 255u) / 256u;
    int cur = 0;
    BUILD_TRY(hipMalloc((void **)&d_items, (size_t)n * sizeof(TreeItem)));
    BUILD_TRY(hipMalloc((void **)&d_perm[0], (size_t)n * 4));
    BUILD_TRY(hipMalloc((void **)&d_perm[1], (size_t)n * 4));
    BUILD_TRY(hipMalloc((void **)&d_keys[0], (size_t)n * 8));
    BUILD_TRY(hipMalloc((void **)&d_keys[1], (size_t)n * 8));
    BUILD_TRY(hipMalloc((void **)&d_bounds, (size_t)max_segs * 2 * 6 * 4));
    BUILD_TRY(hipMalloc((void **)&d_segs[0], (size_t)max_segs * sizeof(Seg)));
    BUILD_TRY(hipMalloc((void **)&d_segs[1], (size_t)max_segs * sizeof(Seg)));
    BUILD_TRY(hipMalloc((void **)&d_nodes, (size_t)max_nodes * sizeof(DNode4)));
    BUILD_TRY(hipMalloc((void **)&d_leaf_item, (size_t)max_nodes * 4 * 4));
    BUILD_TRY(hipMalloc((void **)&d_box, (size_t)max_nodes * sizeof(TreeBox)));
    BUILD_TRY(hipMalloc((void **)&d_n_next, 64));
    BUILD_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys[0], d_keys[1], d_perm[0], d_perm[1], (size_t)n, 0u, 64u, stream));
    BUILD_TRY(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16));
    BUILD_TRY(hipEventCreate(&ev0));
    BUILD_TRY(hipEventCreate(&ev1));
    BUILD_TRY(hipMemcpyAsync(d_items, h_items, (size_t)n * sizeof(TreeItem), hipMemcpyHostToDevice, stream));
    BUILD_TRY(hipEventRecord(ev0, stream));
    hipLaunchKernelGGL(iota_kernel, dim3(grid_n), dim3(256), 0, stream, d_perm[0], n);
    {
        const Seg root{0u, n, 0u};
        BUILD_TRY(hipMemcpyAsync(d_segs[0], &root, sizeof root, hipMemcpyHostToDevice, stream));
    }
    while (n_segs > 0) {
        if (levels >= 64u) {
            rc = (int)hipErrorInvalidValue;
            goto done;
        }
        for (int phase = 0; phase < 2; ++phase) {
            hipLaunchKernelGGL(bounds_init_kernel, dim3((n_segs * 2u + 255u) / 256u), dim3(256), 0, stream, d_bounds, n_segs * 2u);
            hipLaunchKernelGGL(bounds_kernel, dim3(grid_n), dim3(256), 0, stream, d_items, d_perm[cur], n, d_segs[levels & 1u], n_segs, phase, d_bounds);
            hipLaunchKernelGGL(keys_kernel, dim3(grid_n), dim3(256), 0, stream, d_items, d_perm[cur], n, d_segs[levels & 1u], n_segs, phase, d_bounds, d_keys[0]);
            BUILD_TRY(hipGetLastError());
            BUILD_TRY(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_keys[0], d_keys[1], d_perm[cur], d_perm[cur ^ 1], (size_t)n, 0u, 64u, stream));
            cur ^= 1;
        }
        level_first[levels] = node_count - n_segs;   // this level's nodes are the last n_segs numbered so far
        level_count[levels] = n_segs;
        hipLaunchKernelGGL(emit_kernel, dim3(1), dim3(1024), 0, stream, d_items, d_perm[cur], d_segs[levels & 1u], n_segs, node_count, d_segs[(levels & 1u) ^ 1u],
                           d_nodes, d_leaf_item, d_n_next);
        BUILD_TRY(hipGetLastError());
        uint32_t n_next = 0;
        BUILD_TRY(hipMemcpyAsync(&n_next, d_n_next, 4, hipMemcpyDeviceToHost, stream));
        BUILD_TRY(hipStreamSynchronize(stream));   // the only thing the host reads per level: how many segments the next one has
        if (n_next > max_segs || node_count + n_next > max_nodes) {
            rc = (int)hipErrorInvalidValue;
            goto done;
        }
        node_count += n_next;
        n_segs = n_next;
        ++levels;
    }
    for (uint32_t l = levels; l-- > 0;) {
        hipLaunchKernelGGL(boxes_kernel, dim3((level_count[l] + 127u) / 128u), dim3(128), 0, stream, d_items, d_leaf_item, d_nodes, d_box, level_first[l], level_count[l]);
    }
    BUILD_TRY(hipGetLastError());
    BUILD_TRY(hipEventRecord(ev1, stream));
    BUILD_TRY(hipStreamSynchronize(stream));
    if (ms_out) BUILD_TRY(hipEventElapsedTime(ms_out, ev0, ev1));
    *d_nodes_out = d_nodes, d_nodes = nullptr;
    *n_nodes_out = node_count;
    *depth_out = levels;
done:
    (void)hipFree(d_items);
    (void)hipFree(d_perm[0]);
    (void)hipFree(d_perm[1]);
    (void)hipFree(d_keys[0]);
    (void)hipFree(d_keys[1]);
    (void)hipFree(d_bounds);
    (void)hipFree(d_segs[0]);
    (void)hipFree(d_segs[1]);
    (void)hipFree(d_nodes);
    (void)hipFree(d_leaf_item);
    (void)hipFree(d_box);
    (void)hipFree(d_n_next);
    (void)hipFree(d_tmp);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    return rc;
}

int tree4_pack_device(const DNode4 *d_nodes, uint32_t n_nodes, const float4 *d_spheres, const float4 *d_leafrec, hipStream_t stream,
                      DNode4Q **d_packed_out, float4 **d_slotrec_out, bool *ok_out) {
    *d_packed_out = nullptr, *d_slotrec_out = nullptr, *ok_out = false;
    if (n_nodes == 0) return 0;
    int rc = 0;
    DNode4Q *d_packed = nullptr;
    float4 *d_slotrec = nullptr;
    uint32_t *d_failed = nullptr, failed = 1;
    BUILD_TRY(hipMalloc((void **)&d_packed, (size_t)n_nodes * sizeof(DNode4Q)));
    BUILD_TRY(hipMalloc((void **)&d_slotrec, (size_t)n_nodes * 4 * 4 * sizeof(float4)));
    BUILD_TRY(hipMalloc((void **)&d_failed, 64));
    BUILD_TRY(hipMemsetAsync(d_failed, 0, 4, stream));
    hipLaunchKernelGGL(pack_kernel, dim3((n_nodes + 127u) / 128u), dim3(128), 0, stream, d_nodes, n_nodes, d_spheres, d_leafrec, d_packed, d_slotrec, d_failed);
    BUILD_TRY(hipGetLastError());
    BUILD_TRY(hipMemcpyAsync(&failed, d_failed, 4, hipMemcpyDeviceToHost, stream));
    BUILD_TRY(hipStreamSynchronize(stream));
    *ok_out = failed == 0;
    *d_packed_out = d_packed, d_packed = nullptr;
    *d_slotrec_out = d_slotrec, d_slotrec = nullptr;
done:
    (void)hipFree(d_packed);
    (void)hipFree(d_slotrec);
    (void)hipFree(d_failed);
    return rc;
}

}  // namespace ptdev
