// pt_graph.h -- scene graphs that the list form cannot express (include/ptgpu.h pt_node), INTERPRETED: collision/hitable.rs:12-21
// lets Hitables nest freely, and two nestings do not flatten --
//   * a ConstantMedium whose boundary is a HitableList or another ConstantMedium (constant_medium.rs:32-43 asks the boundary twice; a
//     medium in there draws from the pixel's RNG both times), and
//   * a BVHNode anywhere below the root (bvh.rs:37-62: the box first, then lhs AND rhs with the original t_max).
// For such a graph the general-world kernel replaces its list scan by the walk below: Hitable::ray_hit (hitable.rs:39-65) as the
// reference recurses, with an explicit stack of frames per lane in global memory -- every lane its own walk, no attempt at speed (the
// flattened form is the fast one; this one exists so that every graph the reference accepts renders, bit for bit).
//
// A frame = one ray_hit call in progress: the node, where it is (`state`), its (t_min, t_max), the ray it was called with (an Instance
// replaces the lane's ray for its child and puts it back), a medium's first boundary parameter, and the best / lhs hit so far.
#pragma once

namespace ptdev {

// (kGraphDepth nested ray_hit calls -- the host refuses deeper graphs by name -- of kGraphFrame words each: pt_args.h)
enum : uint32_t { kGfNode = 0, kGfState = 1, kGfTmin = 2, kGfTmax = 3, kGfRay = 4 /* o, d, time: 7 */, kGfFirst = 11, kGfBest = 12 /* found, mat, t, point, normal, u, v: 11 */ };

struct GHit {   // what a ray_hit call returns (ray.rs:43-50 + the material)
    bool found;
    uint32_t mat;
    WHit h;
};

struct GraphSrc {
    const uint4 *nodes;          // pt_node rows
    const uint32_t *children;    // HitableList children
    const pt_bvh_node *boxes;    // BVHNode rows: box + the two child NODES
    uint32_t root;
};

// Hitable::ray_hit of the graph's root for `ray` in (t_min, t_max). `fr`: this lane's frames (word w of level l at fr[(l * kGraphFrame + w) * kBlock]).
static __device__ __noinline__ GHit graph_ray_hit(const GraphSrc G, const pt_hitable *hit, const pt_affine *xf, WRay ray, float t_min, float t_max, Rng &rng, float *fr,
                                           bool want_uv) {
    auto W = [&](uint32_t level, uint32_t word) -> float & { return fr[((size_t)level * kGraphFrame + word) * (size_t)kBlock]; };
    auto Wu = [&](uint32_t level, uint32_t word) -> uint32_t & { return reinterpret_cast<uint32_t *>(fr)[((size_t)level * kGraphFrame + word) * (size_t)kBlock]; };
    auto store_hit = [&](uint32_t level, const GHit &g) {
        Wu(level, kGfBest) = g.found ? 1u : 0u, Wu(level, kGfBest + 1) = g.mat, W(level, kGfBest + 2) = g.h.t;
        W(level, kGfBest + 3) = g.h.point.x, W(level, kGfBest + 4) = g.h.point.y, W(level, kGfBest + 5) = g.h.point.z;
        W(level, kGfBest + 6) = g.h.normal.x, W(level, kGfBest + 7) = g.h.normal.y, W(level, kGfBest + 8) = g.h.normal.z;
        W(level, kGfBest + 9) = g.h.u, W(level, kGfBest + 10) = g.h.v;
    };
    auto load_hit = [&](uint32_t level) -> GHit {
        GHit g;
        g.found = Wu(level, kGfBest) != 0u, g.mat = Wu(level, kGfBest + 1), g.h.t = W(level, kGfBest + 2);
        g.h.point = mk3(W(level, kGfBest + 3), W(level, kGfBest + 4), W(level, kGfBest + 5));
        g.h.normal = mk3(W(level, kGfBest + 6), W(level, kGfBest + 7), W(level, kGfBest + 8));
        g.h.u = W(level, kGfBest + 9), g.h.v = W(level, kGfBest + 10);
        return g;
    };
    uint32_t sp = 0;
    auto call = [&](uint32_t node, float lo, float hi) {   // (the host checked the depth)
        Wu(sp, kGfNode) = node, Wu(sp, kGfState) = 0u, W(sp, kGfTmin) = lo, W(sp, kGfTmax) = hi;
        sp += 1u;
    };
    GHit ret;
    ret.found = false, ret.mat = 0u, ret.h = WHit{};
    bool returned = false;   // `ret` is the answer of the call the top frame made last
    call(G.root, t_min, t_max);
    while (sp != 0u) {
        const uint32_t lv = sp - 1u;
        const uint4 N = G.nodes[Wu(lv, kGfNode)];
        const uint32_t state = Wu(lv, kGfState);
        const float lo = W(lv, kGfTmin), hi = W(lv, kGfTmax);
        switch (N.x) {
        case PT_NODE_HITABLE: {   // hitable.rs:47-57, a leaf shape with the ray as its callers made it
            const pt_hitable &H = hit[N.y];
            float t;
            uint32_t face = 0u;
            ret.found = w_shape_t(H, ray, lo, hi, t, face);
            if (ret.found) {
                w_shape_rec(H, ray, t, face, ret.h, want_uv);
                ret.h.t = t, ret.mat = H.material;
            }
            returned = true, sp -= 1u;
            break;
        }
        case PT_NODE_LIST: {   // hitable_list.rs:40-56: state = children asked so far; frame t_max = closest_so_far
            if (state == 0u) Wu(lv, kGfBest) = 0u;
            if (returned) {
                if (ret.found) store_hit(lv, ret), W(lv, kGfTmax) = ret.h.t;
                returned = false;
            }
            if (state == N.z) {
                ret = load_hit(lv);
                returned = true, sp -= 1u;
            } else {
                Wu(lv, kGfState) = state + 1u;
                call(G.children[N.y + state], lo, W(lv, kGfTmax));
            }
            break;
        }
        case PT_NODE_INSTANCE: {   // instance.rs:32-47
            const pt_affine &T = xf[N.y];
            if (state == 0u) {
                W(lv, kGfRay + 0) = ray.o.x, W(lv, kGfRay + 1) = ray.o.y, W(lv, kGfRay + 2) = ray.o.z;
                W(lv, kGfRay + 3) = ray.d.x, W(lv, kGfRay + 4) = ray.d.y, W(lv, kGfRay + 5) = ray.d.z, W(lv, kGfRay + 6) = ray.time;
                ray = w_ray_new(w_xf_point(T.inv, ray.o), w_xf_vector(T.inv, ray.d), ray.time);
                Wu(lv, kGfState) = 1u;
                call(N.z, lo, hi);
            } else {
                ray = w_ray_new(mk3(W(lv, kGfRay + 0), W(lv, kGfRay + 1), W(lv, kGfRay + 2)), mk3(W(lv, kGfRay + 3), W(lv, kGfRay + 4), W(lv, kGfRay + 5)), W(lv, kGfRay + 6));
                if (ret.found) ret.h.point = w_xf_point(T.m, ret.h.point), ret.h.normal = w_xf_vector(T.m, ret.h.normal);
                sp -= 1u;   // (returned stays true: the child's answer, carried out)
            }
            break;
        }
        case PT_NODE_MEDIUM: {   // constant_medium.rs:32-77
            if (state == 0u) {
                Wu(lv, kGfState) = 1u;
                call(N.z, -kMaxT, kMaxT);
            } else if (state == 1u) {
                if (!ret.found) {
                    sp -= 1u;
                } else {
                    W(lv, kGfFirst) = ret.h.t;
                    Wu(lv, kGfState) = 2u;
                    returned = false;
                    call(N.z, ret.h.t + 0.0001f, kMaxT);
                }
            } else {
                if (ret.found) {
                    float t1 = W(lv, kGfFirst), t2 = ret.h.t;
                    ret.found = false;
                    if (t1 < lo) t1 = lo;
                    if (t2 > hi) t2 = hi;
                    if (!(t1 >= t2)) {
                        if (t1 < 0.0f) t1 = 0.0f;
                        const float ray_length = length3(ray.d);
                        const float distance_inside_boundary = (t2 - t1) * ray_length;
                        const float hit_distance = -(1.0f / __uint_as_float(N.w)) * logf_ref(rng_f32(rng));
                        if (hit_distance < distance_inside_boundary) {
                            const float t = t1 + hit_distance / ray_length;
                            ret.found = true, ret.mat = N.y;
                            ret.h.t = t, ret.h.point = add3(ray.o, scale3(ray.d, t)), ret.h.normal = mk3(1.0f, 0.0f, 0.0f), ret.h.u = 0.0f, ret.h.v = 0.0f;
                        }
                    }
                }
                sp -= 1u;
            }
            break;
        }
        default: {   // PT_NODE_BVH, bvh.rs:37-62
            const pt_bvh_node B = G.boxes[N.y];
            if (state == 0u) {
                if (!w_aabb_hit(mk3(B.min[0], B.min[1], B.min[2]), mk3(B.max[0], B.max[1], B.max[2]), ray, lo, hi)) {
                    ret.found = false;
                    returned = true, sp -= 1u;
                } else {
                    Wu(lv, kGfState) = 1u;
                    call((uint32_t)B.lhs, lo, hi);
                }
            } else if (state == 1u) {
                store_hit(lv, ret);
                Wu(lv, kGfState) = 2u;
                returned = false;
                call((uint32_t)B.rhs, lo, hi);   // (the original t_max)
            } else {
                const GHit l = load_hit(lv);
                if (l.found && (!ret.found || l.h.t < ret.h.t)) ret = l;   // bvh.rs:48-53: lhs only when lhs.t < rhs.t
                sp -= 1u;
            }
            break;
        }
        }
    }
    (void)returned;
    return ret;
}

}  // namespace ptdev
