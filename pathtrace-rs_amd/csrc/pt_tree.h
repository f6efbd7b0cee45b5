// pt_tree.h -- closest hit through the internal trees: the binary tree (variant) and the 4-wide packed tree with whole-wave, work-sharing traversal.
#pragma once
#include "pt_prefilter.h"
#include "pt_tree4.h"

namespace ptdev {

// bvh.rs:37-62 over the CALLER's tree, restructured for the GPU without changing its result.
//
// Reference semantics: a leaf sphere is tested (with t_max = f32::MAX) iff every ancestor node's
// AABB passes aabb.rs:46-58 with (t_min, f32::MAX); among the hits the smallest t wins and equal t
// resolves to the leaf that comes LAST in lhs-before-rhs DFS order (`lhs.t < rhs.t ? lhs : rhs`).
//
// Here: each device node carries the AABBs of its two children (one 64-byte fetch tests both), the
// children are visited near-first, and a subtree is skipped when its slab entry distance exceeds the
// best hit so far by a safety slack. Skipping such a subtree cannot change the winner: every sphere
// inside has t >= entry distance (up to rounding, covered by the slack), and the
// subtree's AABB test itself is the reference's, evaluated with the reference's arithmetic. Equal-t
// ties are resolved by the precomputed DFS rank of the leaf instead of by visiting order.
// Relative / absolute slack of the distance cull. The entry distance of a (padded) box is a GEOMETRIC lower bound of every hit inside it;
// what the slack has to cover is how far the reference's f32 ROOT (sphere.rs:40: (-b - sqrt(disc)) / a) can fall below the true parameter:
// ~1e-6 of t for origins far from the sphere (cancellation in -b - sqrt), nothing near it. 5e-4 relative + 5e-4 absolute is two orders
// above that. (Rounds 1-4 used 2 % + 0.02: at t = 20 every box entered within 0.4 units behind the nearest hit was still visited.)
#ifndef PT_CULL_REL
#define PT_CULL_REL 1.0005f
#define PT_CULL_ABS 5.0e-4f
#endif
constexpr float kCullRel = PT_CULL_REL;
constexpr float kCullAbs = PT_CULL_ABS;

// One leaf of the reference tree: hitable.rs:47 passes the ORIGINAL t_max to the sphere, and the sphere only
// counts if every ancestor AABB passed aabb.rs:46-58. Ancestor boxes nest (each is the union of its
// children, aabb.rs:61-66, and the slab arithmetic is monotone in the box), so testing the sphere's PARENT
// box with the reference's exact arithmetic decides all of them.
__device__ __forceinline__ void bvh_leaf(const KArgs &A, int k, const float4 c, f3 o, f3 d, f3 rcp, const DivA &av, float &best,
                                         int &idx, uint32_t &best_rank) {
    const float a = av.a;
    const float ocx = o.x - c.x, ocy = o.y - c.y, ocz = o.z - c.z;
    const float b = (ocx * d.x + ocy * d.y) + ocz * d.z;
    const float cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - c.w * c.w;
    const float disc = b * b - a * cc;
    if (disc > 0.0f) {
        float t = kMaxT;
        if (sphere_roots(av, b, disc, t)) {
            // BVH world: DFS-last leaf wins equal t (bvh.rs:47-53); list world: the lower list index (hitable_list.rs:48)
            const uint32_t rank = A.gate ? A.leaf_rank[k] : ~(uint32_t)k;
            if (idx < 0 || t < best || (t == best && rank > best_rank)) {
                if (!A.gate || gate_pass(A, k, o, rcp)) {
                    best = t;
                    idx = k;
                    best_rank = rank;
                }
            }
        }
    }
}

// Conservative slab test of an INTERNAL-tree box: never rejects a box that contains a sphere whose
// reference discriminant can be positive. The reference's f32 discriminant differs from the exact one by
// <= ~1.3e-6 * a * (|o-c|^2 + r^2), i.e. a sphere behaves as if its radius were larger by at most
// ~0.65e-6 * (|o-c|^2 + r^2) / r; the box is padded by >= 4x that bound (r_min = smallest radius below the
// node) plus an absolute epsilon. NaNs (0 * inf) count as a hit.
__device__ __forceinline__ bool accel_box_hit(const float c[3], const float h[3], float inv_rmin, f3 o, f3 rcp, float limit,
                                              float &t_enter) {
    const float cx = c[0] - o.x, cy = c[1] - o.y, cz = c[2] - o.z;
    const float reach2 = 2.0f * ((cx * cx + cy * cy + cz * cz) + (h[0] * h[0] + h[1] * h[1] + h[2] * h[2]));
    const float pad = 3.0e-6f * reach2 * inv_rmin + 1.0e-4f;
    const float hx = h[0] + pad, hy = h[1] + pad, hz = h[2] + pad;
    const float ax = (cx - hx) * rcp.x, bx = (cx + hx) * rcp.x;
    const float ay = (cy - hy) * rcp.y, by = (cy + hy) * rcp.y;
    const float az = (cz - hz) * rcp.z, bz = (cz + hz) * rcp.z;
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    t_enter = tn;
    return !(tf < tn) && !(tn > limit);
}

// Resumable per-lane traversal state. A ray-iteration in BVH mode is NOT lockstep: lanes whose traversal
// has finished are shaded (and given their next ray) as soon as enough of them are waiting, while the
// long-tail lanes simply keep their stack and continue in the next round -- otherwise every wave would
// run as long as its slowest ray (measured: 19 % lane utilisation with lockstep iterations).
struct BvhTrav {
    uint32_t visits, leaves;   // VERIFY kernels: internal-tree nodes fetched / spheres tested (SURVEY 8d counters)
    int sp;
    float best;
    int idx;
    uint32_t rank;
    bool active;
};

template <bool MOVING>
__device__ __forceinline__ void bvh_start(const KArgs &A, uint32_t *s_stack, f3 o, f3 d, float a, float time, BvhTrav &st) {
    const f3 rcp = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);  // ray.rs:14
    st.sp = 0;
    st.best = kMaxT;
    st.idx = -1;
    st.rank = 0;
    st.active = true;
    for (uint32_t j = 0; j < A.n_bvh_large; ++j) {
        const int k = (int)A.bvh_large[j];
        bvh_leaf(A, k, sphere_at<MOVING>(A, k, A.spheres[k], time), o, d, rcp, DivA{a, 0.0f, false}, st.best, st.idx, st.rank);
    }
    if (A.bvh_root >= 0) s_stack[(st.sp++) * kBlock + threadIdx.x] = (uint32_t)A.bvh_root;
}

template <bool NODES_LDS, bool MOVING, bool COUNT>
__device__ __forceinline__ void bvh_run(const KArgs &A, uint32_t *s_stack, const DWideNode *nodes, f3 o, f3 d, float a,
                                        float time, bool have, BvhTrav &st) {
    const f3 rcp = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const int tid = threadIdx.x;
    for (;;) {
        if (st.active) {
            if (st.sp == 0) {
                st.active = false;
            } else {
                const int32_t ref = (int32_t)s_stack[(--st.sp) * kBlock + tid];
                const DWideNode n = nodes[ref];
                if (COUNT) st.visits += 1u, st.leaves += (uint32_t)(n.lhs < 0) + (uint32_t)(n.rhs < 0);
                // leaves first: they can only shrink `best` before the inner children are considered
                // a leaf child's box slot holds the sphere itself (centre, radius): no second fetch
                if (n.lhs < 0) bvh_leaf(A, ~n.lhs, sphere_at<MOVING>(A, ~n.lhs, make_float4(n.lmin[0], n.lmin[1], n.lmin[2], n.lmax[0]), time), o, d, rcp, DivA{a, 0.0f, false}, st.best, st.idx, st.rank);
                if (n.rhs < 0) bvh_leaf(A, ~n.rhs, sphere_at<MOVING>(A, ~n.rhs, make_float4(n.rmin[0], n.rmin[1], n.rmin[2], n.rmax[0]), time), o, d, rcp, DivA{a, 0.0f, false}, st.best, st.idx, st.rank);
                const float limit = (st.idx >= 0) ? (st.best * kCullRel + kCullAbs) : kMaxT;
                float tl = 0.f, tr = 0.f;
                bool hl = false, hr = false;
                if (n.lhs >= 0) hl = accel_box_hit(n.lmin, n.lmax, __uint_as_float(n.pad0), o, rcp, limit, tl);
                if (n.rhs >= 0) hr = accel_box_hit(n.rmin, n.rmax, __uint_as_float(n.pad1), o, rcp, limit, tr);
                if (hl && hr) {
                    const bool l_near = tl <= tr;
                    s_stack[(st.sp++) * kBlock + tid] = (uint32_t)(l_near ? n.rhs : n.lhs);
                    s_stack[(st.sp++) * kBlock + tid] = (uint32_t)(l_near ? n.lhs : n.rhs);
                } else if (hl) {
                    s_stack[(st.sp++) * kBlock + tid] = (uint32_t)n.lhs;
                } else if (hr) {
                    s_stack[(st.sp++) * kBlock + tid] = (uint32_t)n.rhs;
                }
            }
        }
        if (wave_ballot(st.active) == 0ull) break;
        if (__popcll(wave_ballot(have && !st.active)) >= kReadyMin) break;
    }
}

// ---- 4-wide internal tree ------------------------------------------------------------------------------------
// The tree kernels' default traversal structure (DESIGN.md "tree kernel"). One 64-byte node (pt_tree4.h DNode4Q) holds the
// boxes of up to four children as plane arrays (SoA) of f16 offsets from the node's min corner, so a visit is FOUR 16-byte
// loads (the vector L1 pays one tag lookup per lane and load: the visit is bound by their number) and four box tests of
// identical, branch-free code; a child is an inner node or ONE sphere (leaf). Per visit the lane
//   * pads all four boxes by ONE node-level bound of the reference's f32 discriminant error (same bound as
//     accel_box_hit, taken over the node: every sphere below lies within |c_node - o| + |h_node| of the origin),
//   * evaluates each plane with one mixed-precision FMA, t = offset(f16) * rcp_d + ((origin - o) * rcp_d -+ pad * |rcp_d|) --
//     the near / far plane arrays are picked by the ray's direction signs with two selects per axis and side,
//   * pushes the inner children it hit far-to-near (4 sort keys = entry distance bits | slot, a 5-exchange network of
//     v_min_u32 / v_max_u32), keeps the nearest in a register as the next node, and
//   * appends the leaf children it hit to its queue of (sphere) candidates.
// Candidates are NOT tested by the lane that found them: like phase 2 of the MFMA list kernel they are expanded into one
// (owner ray, sphere) pair list per wave and every lane takes one pair per round (exact reference arithmetic, bvh_leaf's
// accept rule, ds_min_u64 on the owner's (t, tie-break) key), so the exact tests run on full waves whatever the spread
// of the lanes' traversals. A lane's nearest hit so far (`best`, the culling limit) is refreshed from its key after
// every drain.

__device__ __forceinline__ float trav4_limit(float best) { return best < kMaxT ? (best * kCullRel + kCullAbs) : kMaxT; }

// key of an accepted hit: smaller t wins; equal t goes to the lower list index (hitable_list.rs:48) or, in a BVH world,
// to the DFS-later leaf (bvh.rs:47-53) -- the order-independent form of both scans (accept_hit / bvh_leaf)
__device__ __forceinline__ unsigned long long key4_of(const KArgs &A, float t, int k) {
    const uint32_t low = A.gate ? (0xffffffffu - A.leaf_rank[k]) : (uint32_t)k;
    return ((unsigned long long)__float_as_uint(t) << 32) | low;
}

// exact reference test of one (ray, leaf slot) pair reduced into the owner's key (sphere.rs:29-66 with t_max = f32::MAX,
// then the ancestor-AABB gate of a BVH world). The slot record holds the sphere together with its gate box, rank and
// index: the accept rule needs no dependent loads.
template <bool MOVING>
__device__ __forceinline__ void pair_test4(const KArgs &A, const float4 *recs, uint32_t e, float time, f3 o, f3 d, f3 rcp, const DivA &av, unsigned long long *key) {
    const float a = av.a;
    const bool gated = A.gate != nullptr;
    const float4 *R = recs + 4 * (size_t)e;
    float4 c = R[0], g0 = make_float4(0, 0, 0, 0), g1 = g0;
    const float4 g2 = R[3];
    if (gated) g0 = R[1], g1 = R[2];
    const int k = (int)__float_as_uint(g2.y);
    c = sphere_at<MOVING>(A, k, c, time);
    const float ocx = o.x - c.x, ocy = o.y - c.y, ocz = o.z - c.z;
    const float b = (ocx * d.x + ocy * d.y) + ocz * d.z;
    const float cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - c.w * c.w;
    const float disc = b * b - a * cc;
    const float t = sphere_hit_t(av, b, disc, true);
    if (t < kMaxT) {
        const uint32_t low = gated ? (0xffffffffu - __float_as_uint(g2.x)) : (uint32_t)k;
        const unsigned long long kk = ((unsigned long long)__float_as_uint(t) << 32) | low;
        if (kk < *key && (!gated || gate_pass_loaded(A, g0, g1, o, rcp))) atomicMin(key, kk);   // rcp = ray.rs:14 rcp_direction of the OWNER's ray
    }
}

// ---- the traversal loop: whole-wave iterations with work sharing (round 5) ------------------------------------------------------------
// One call traces the rays of ALL lanes of the wave to their end. A lane whose own walk is finished does not wait: it takes a pending
// subtree off the stack of a lane that has one to spare -- the entry at the BOTTOM of that stack: the shallowest, i.e. largest,
// pending subtree -- and walks it for the owner's ray. The ray (origin, 1 / d, pad, culling limit) is fetched across lanes once per
// hand-over, leaf candidates are queued under the OWNER's lane number and the exact tests reduce into the owner's key. The call
// returns when NO lane has work left, so every ray of the wave is finished at the same point and nothing about "whose ray is
// complete" has to be communicated; no traversal state survives a trip of the kernel's main loop. (Until round 4 a traversal was
// resumable per lane and a wave left the loop as soon as 56 of its lanes were done, because a finished lane could only wait: 19 visit
// rounds per wave-iteration for the 9.8 visits a ray needs, 41 of 64 lanes switched on in the block that is 45 % of the kernel --
// profiles/r04_c5_bbprof_lanes.txt. Now: 13.7 rounds for 10.5 visits -- a helper's subtree is sometimes one the owner would have culled.)
// Results cannot change: the winner is the (t, tie-break) minimum over every leaf whose boxes the ray enters, whatever the order and
// whoever visits them; a helper culls with the limit it fetched (refreshed from the OWNER's key at each
// drain), which is never tighter than what the owner's own walk would use at that moment... and never looser than "no limit".
struct Steal4 {
    uint32_t visits, leaves;
};

template <bool MOVING>
__device__ __forceinline__ void pair_test4_owner(const KArgs &A, const float4 *recs, uint32_t slot, float time, f3 o, f3 d, const DivA &av, unsigned long long *key) {
    // (the gate of a BVH world needs ray.rs:14's 1 / d of the OWNER's ray; the owner's lane may be walking somebody else's subtree with
    //  another ray's reciprocal in its registers, so it is formed here, from the fetched direction, in its short exact form)
    const f3 rcp = A.gate ? mk3(recip_exact(d.x), recip_exact(d.y), recip_exact(d.z)) : mk3(0.f, 0.f, 0.f);
    pair_test4<MOVING>(A, recs, slot, time, o, d, rcp, av, key);
}

// Exact tests of the leaf candidates the lanes of a wave have queued (`leafq`, `qn` per lane; an entry = owner lane << kPairLaneShift | slot
// record): expanded into ONE list of (owner ray, slot) pairs per wave, every lane takes one pair per round whoever's ray it belongs to,
// and the test reduces into the owner's (t, tie-break) key with ds_min_u64. The owners' rays are fetched across lanes from (o, d, av,
// time): the registers every lane holds for ITS OWN ray. `recs`: the 64-byte records the entries index (the tree's leaf slots, or the cell
// grid's per-sphere records). Returns false when nothing was queued; empties the queues.
template <bool MOVING, int BLK>
__device__ __forceinline__ bool drain_pairs4(const KArgs &A, const float4 *recs, uint32_t *leafq, uint32_t *w_pairs, unsigned long long *w_keys, uint32_t &qn, f3 o, f3 d, const DivA &av, float time,
                                             uint32_t owner_tag) {
    const int tid = threadIdx.x;
    const uint32_t lane = (uint32_t)tid & 63u;
    const uint32_t incl = wave_inclusive_sum(qn);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (total == 0u) return false;
    const float a = av.a;
    if (total > (uint32_t)kPairCap) {
        // more pairs than the wave's list holds (rare): every lane walks its own queue, the owners' rays still come across lanes
        for (uint32_t j = 0; wave_any(j < qn); ++j) {
            const bool valid = j < qn;
            const uint32_t e = valid ? leafq[j * BLK + tid] : owner_tag;
            const uint32_t ow = e >> kPairLaneShift;
            const f3 po = mk3(lane_fetch(ow, o.x), lane_fetch(ow, o.y), lane_fetch(ow, o.z));
            const f3 pd = mk3(lane_fetch(ow, d.x), lane_fetch(ow, d.y), lane_fetch(ow, d.z));
            const DivA pav{lane_fetch(ow, a), lane_fetch(ow, av.y), av.fast};
            const float ptime = MOVING ? lane_fetch(ow, time) : 0.0f;
            if (valid) pair_test4_owner<MOVING>(A, recs, e & ((1u << kPairLaneShift) - 1u), ptime, po, pd, pav, &w_keys[ow]);
        }
    } else {
        uint32_t pos = incl - qn;
        for (uint32_t j = 0; j < qn; ++j) w_pairs[pos++] = leafq[j * BLK + tid];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t b0 = 0; b0 < total; b0 += 64u) {
            const bool valid = b0 + lane < total;
            const uint32_t e = valid ? w_pairs[b0 + lane] : owner_tag;
            const uint32_t ow = e >> kPairLaneShift;
            const f3 po = mk3(lane_fetch(ow, o.x), lane_fetch(ow, o.y), lane_fetch(ow, o.z));
            const f3 pd = mk3(lane_fetch(ow, d.x), lane_fetch(ow, d.y), lane_fetch(ow, d.z));
            const DivA pav{lane_fetch(ow, a), lane_fetch(ow, av.y), av.fast};
            const float ptime = MOVING ? lane_fetch(ow, time) : 0.0f;
            if (valid) pair_test4_owner<MOVING>(A, recs, e & ((1u << kPairLaneShift) - 1u), ptime, po, pd, pav, &w_keys[ow]);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    qn = 0;
    return true;
}

template <bool MOVING, bool COUNT, int BLK>
__device__ __forceinline__ void bvh4_trace(const KArgs &A, uint16_t *s_stack, uint32_t *leafq, uint32_t *w_pairs, unsigned long long *w_keys,
                                           f3 o, f3 d, const DivA &av, float time, bool start, Steal4 &cnt, unsigned long long *sec = nullptr) {
    const int tid = threadIdx.x;
    const uint32_t lane = (uint32_t)tid & 63u;
#ifdef PT_SECTIONS
    unsigned long long sub_last = __builtin_readcyclecounter();   // sec[5] node visits, sec[6] drains + hand-overs, sec[7] rounds (count)
#define PT_SUBT(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); sec[i] += now_ - sub_last; sub_last = now_; } while (0)
#else
    (void)sec;
#define PT_SUBT(i) do { } while (0)
#endif
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    const uint4 *base = reinterpret_cast<const uint4 *>(A.nodes4);
    const auto slot_of_entry = [&](int e, uint32_t column) -> uint32_t { return (uint32_t)e * (uint32_t)BLK + column; };
    // the ray this lane TRAVERSES with (its own until it takes over part of another lane's walk)
    f3 to = o, trcp = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);   // ray.rs:14
    float tpad = 1.0e-6f * (__builtin_fabsf(o.x) + __builtin_fabsf(o.y) + __builtin_fabsf(o.z));
    float limit = kMaxT;
    uint32_t owner_tag = lane << kPairLaneShift;   // whose ray that is, as the pair list wants it
    int sp = 0, sb = 0;                            // live stack entries of this lane: [sb, sp)
    int32_t cur = kNoChild4;
    uint32_t qn = 0;
    if (start) {
        float best = kMaxT;
        int idx = -1;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < A.n_bvh_large; ++j) {   // spheres kept out of the tree: tested for every ray
            const int k = (int)A.bvh_large[j];
            bvh_leaf(A, k, sphere_at<MOVING>(A, k, A.spheres[k], time), o, d, trcp, av, best, idx, rank);
        }
        w_keys[lane] = idx < 0 ? ~0ull : key4_of(A, best, idx);
        limit = trav4_limit(idx < 0 ? kMaxT : best);
        cur = A.bvh_root >= 0 ? 0 : kNoChild4;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    for (;;) {
        if (cur == kNoChild4 && sp > sb) cur = (int32_t)s_stack[slot_of_entry(--sp, (uint32_t)tid)];
        if (cur != kNoChild4) {
            const uint4 *np = base + (size_t)(uint32_t)cur * 4u;
            const uint4 qx = np[0], qy = np[1], qz = np[2], qm = np[3];   // (lo[4], hi[4]) f16 offsets per axis | origin, meta
            if (COUNT) cnt.visits += 1u;
            const bool neg_x = trcp.x < 0.0f, neg_y = trcp.y < 0.0f, neg_z = trcp.z < 0.0f;   // near plane of an axis = the upper one when the ray runs down it
            const uint32_t meta = qm.w;
            const uint32_t cbase = meta & 0xffffu, n_inner = (meta >> 16) & 7u;
            const float pk = __uint_as_float((__builtin_amdgcn_ubfe(meta, 22, 5) << 23) + (96u << 23));
            const float p0 = __uint_as_float(((meta >> 27) << 23) + (113u << 23));
            const float ex = __uint_as_float(qm.x) - to.x, ey = __uint_as_float(qm.y) - to.y, ez = __uint_as_float(qm.z) - to.z;
            const float pad = __builtin_fmaf(pk, __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex)), p0) + tpad;
            const float px = pad * __builtin_fabsf(trcp.x), py = pad * __builtin_fabsf(trcp.y), pz = pad * __builtin_fabsf(trcp.z);
            const float Bnx = __builtin_fmaf(ex, trcp.x, -px), Bny = __builtin_fmaf(ey, trcp.y, -py), Bnz = __builtin_fmaf(ez, trcp.z, -pz);
            const float Bfx = __builtin_fmaf(ex, trcp.x, px), Bfy = __builtin_fmaf(ey, trcp.y, py), Bfz = __builtin_fmaf(ez, trcp.z, pz);
            const uint32_t nxw[2] = {neg_x ? qx.z : qx.x, neg_x ? qx.w : qx.y}, fxw[2] = {neg_x ? qx.x : qx.z, neg_x ? qx.y : qx.w};
            const uint32_t nyw[2] = {neg_y ? qy.z : qy.x, neg_y ? qy.w : qy.y}, fyw[2] = {neg_y ? qy.x : qy.z, neg_y ? qy.y : qy.w};
            const uint32_t nzw[2] = {neg_z ? qz.z : qz.x, neg_z ? qz.w : qz.y}, fzw[2] = {neg_z ? qz.x : qz.z, neg_z ? qz.y : qz.w};
            uint32_t key[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const half2v hnx = __builtin_bit_cast(half2v, nxw[j >> 1]), hny = __builtin_bit_cast(half2v, nyw[j >> 1]), hnz = __builtin_bit_cast(half2v, nzw[j >> 1]);
                const half2v hfx = __builtin_bit_cast(half2v, fxw[j >> 1]), hfy = __builtin_bit_cast(half2v, fyw[j >> 1]), hfz = __builtin_bit_cast(half2v, fzw[j >> 1]);
                const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf((float)hnx[j & 1], trcp.x, Bnx), __builtin_fmaf((float)hny[j & 1], trcp.y, Bny)),
                                                                 __builtin_fmaf((float)hnz[j & 1], trcp.z, Bnz)), 0.0f);
                const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaf((float)hfx[j & 1], trcp.x, Bfx), __builtin_fmaf((float)hfy[j & 1], trcp.y, Bfy)),
                                                 __builtin_fmaf((float)hfz[j & 1], trcp.z, Bfz));
                const bool miss = __builtin_fminf(tf, limit) < tn;
                const bool leaf = (uint32_t)j >= n_inner;
                leafq[qn * BLK + tid] = owner_tag | ((uint32_t)cur << 2) | (uint32_t)j;
                qn += (miss || !leaf) ? 0u : 1u;
                if (COUNT) cnt.leaves += (miss || !leaf) ? 0u : 1u;
                key[j] = (miss || leaf) ? 0xffffffffu : ((__float_as_uint(tn) & ~3u) | (uint32_t)j);
            }
#define PT_CE(a, b) { const uint32_t lo_ = min(key[a], key[b]), hi_ = max(key[a], key[b]); key[a] = lo_; key[b] = hi_; }
            PT_CE(0, 1) PT_CE(2, 3) PT_CE(0, 2) PT_CE(1, 3) PT_CE(1, 2)
#undef PT_CE
            s_stack[slot_of_entry(sp, (uint32_t)tid)] = (uint16_t)(cbase + (key[3] & 3u));
            sp += key[3] != 0xffffffffu ? 1 : 0;
            s_stack[slot_of_entry(sp, (uint32_t)tid)] = (uint16_t)(cbase + (key[2] & 3u));
            sp += key[2] != 0xffffffffu ? 1 : 0;
            s_stack[slot_of_entry(sp, (uint32_t)tid)] = (uint16_t)(cbase + (key[1] & 3u));
            sp += key[1] != 0xffffffffu ? 1 : 0;
            cur = key[0] != 0xffffffffu ? (int32_t)(cbase + (key[0] & 3u)) : kNoChild4;
        }
        const bool work = cur != kNoChild4 || sp > sb;
        const unsigned long long wm = wave_ballot(work);
        const bool stop = wm == 0ull;
        PT_SUBT(5);
#ifdef PT_SECTIONS
        sec[7] += 1ull;
#endif
        if (stop || wave_any(qn > A.drain_at)) {
            // exact tests of the queued leaf candidates (drain_pairs4); afterwards every lane refreshes its culling limit from ITS owner's key
            if (drain_pairs4<MOVING, BLK>(A, A.slotrec, leafq, w_pairs, w_keys, qn, o, d, av, time, owner_tag)) {
                limit = trav4_limit(__uint_as_float((uint32_t)(w_keys[owner_tag >> kPairLaneShift] >> 32)));   // (an empty key's t field is a NaN pattern: not < kMaxT)
            }
        }
        if (stop) break;
        // ---- hand-over: lanes without work take the bottom entry of the stacks of lanes that can spare one
        const unsigned long long im = ~wm;
        if ((uint32_t)__popcll(im) >= A.ready_min) {
#ifndef PT_SHARE_DEPTH
#define PT_SHARE_DEPTH 1
#endif
            const bool offer = sp - sb >= PT_SHARE_DEPTH && (cur != kNoChild4 || sp - sb >= 2);
            const unsigned long long om = wave_ballot(offer);
            if (om != 0ull) {
                const uint32_t n_pairs = min((uint32_t)__popcll(om), (uint32_t)__popcll(im));
                const uint32_t ro = __builtin_amdgcn_mbcnt_hi((uint32_t)(om >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)om, 0u));
                const uint32_t ri = __builtin_amdgcn_mbcnt_hi((uint32_t)(im >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)im, 0u));
                const bool give = offer && ro < n_pairs, take = !work && ri < n_pairs;
                if (give) {
                    w_pairs[ro] = lane | ((uint32_t)sb << 8);
                    sb += 1;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t g = take ? w_pairs[ri] : lane;
                const uint32_t from = g & 63u;
                // (every lane fetches -- a lane that takes nothing fetches its own values)
                const f3 fo = mk3(lane_fetch(from, to.x), lane_fetch(from, to.y), lane_fetch(from, to.z));
                const f3 fr = mk3(lane_fetch(from, trcp.x), lane_fetch(from, trcp.y), lane_fetch(from, trcp.z));
                const float fpad = lane_fetch(from, tpad), flim = lane_fetch(from, limit);
                const uint32_t ftag = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from << 2), (int)owner_tag);
                if (take) {
                    cur = (int32_t)s_stack[slot_of_entry((int)(g >> 8), ((uint32_t)tid & ~63u) | from)];
                    to = fo, trcp = fr, tpad = fpad, limit = flim, owner_tag = ftag;
                    sp = 0, sb = 0;
                }
                __builtin_amdgcn_wave_barrier();   // (the scratch words are the pair list again from here on)
            }
        }
        PT_SUBT(6);
    }
#undef PT_SUBT
}

}  // namespace ptdev
