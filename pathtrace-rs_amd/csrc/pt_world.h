// pt_world.h -- the GENERAL-WORLD kernel: every Hitable arm of collision/hitable.rs:12-21 that the
// reference's presets use besides plain spheres (MovingSphere, Rect, Cuboid, Instance, ConstantMedium),
// traced with the reference's own visiting order.
//
// Why a second kernel instead of more cases in pt_trace_kernel: the sphere kernel is allowed to reorder
// the scan (closest hit is order-independent for spheres). Here it is not:
//   * ConstantMedium::ray_hit (constant_medium.rs:32-77) draws from the PIXEL's RNG inside the
//     intersection and clamps against the running t_max, so HitableList's narrowing scan
//     (hitable_list.rs:40-56) has to run in list order with `closest_so_far` exactly as written;
//   * BVHNode::ray_hit (bvh.rs:37-62) visits lhs then rhs with the ORIGINAL t_max; media below it draw in
//     that DFS order.
// One pixel per lane, persistent workgroups and the regeneration loop are the same as in pt_kernel.h. In
// list mode the hitable index is wave-uniform, so records come in through scalar loads and the `kind`
// switch does not diverge; BVH mode walks the caller's tree per lane (LDS stack).
#pragma once
#include "pt_kernel.h"
#include "ptgpu.h"

namespace ptdev {


struct WRay {  // ray.rs:4-9
    f3 o, d, rcp;
    float time;
};
struct WHit {  // ray.rs:43-50; (u, v) are only computed when the world has an Image texture (nothing else reads them)
    f3 point, normal;
    float t, u, v;
};

// ray.rs:13-21
__device__ __forceinline__ WRay w_ray_new(f3 o, f3 d, float time) {
    return WRay{o, d, mk3(recip_exact(d.x), recip_exact(d.y), recip_exact(d.z)), time};
}

// f32::ln as glibc's logf computes it (constant_medium.rs:60; sysdeps/ieee754/flt-32/e_logf.c, the
// table-driven binary64 evaluation). The value decides whether a ray scatters inside a medium, so it has
// to agree with the CPU bit for bit; checked exhaustively against glibc 2.35 on every positive float
// (FMA and non-FMA builds of that routine round to the same float everywhere).
__device__ __forceinline__ float logf_ref(float x) {
    constexpr double kInvC[16] = {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0,  0x1.3c995b0b80385p+0,
                                  0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,  0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0,
                                  0x1.0953f419900a7p+0, 0x1p+0,               0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
                                  0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
    constexpr double kLogC[16] = {-0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3,
                                  -0x1.6574f0ac07758p-3, -0x1.1aa2bc79c81p-3,   -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4,
                                  -0x1.252f438e10c1ep-5, 0x0p+0,                0x1.aa5aa5df25984p-5,  0x1.c5e53aa362eb4p-4,
                                  0x1.526e57720db08p-3,  0x1.bc2860d22477p-3,   0x1.1058bc8a07ee1p-2,  0x1.4043057b6ee09p-2};
    constexpr double kLn2 = 0x1.62e42fefa39efp-1;
    constexpr double kA0 = -0x1.00ea348b88334p-2, kA1 = 0x1.5575b0be00b6ap-2, kA2 = -0x1.ffffef20a4123p-2;
    uint32_t ix = __float_as_uint(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2u == 0u) return -__builtin_huge_valf();  // ln(+-0) = -inf
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return __builtin_nanf("");
        ix = __float_as_uint(x * 0x1p23f);  // subnormal: normalise
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) & 15u;
    const int32_t k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double z = (double)__uint_as_float(iz);
    const double r = z * kInvC[i] - 1.0;
    const double y0 = kLogC[i] + (double)k * kLn2;
    const double r2 = r * r;
    double y = kA1 * r + kA2;
    y = kA0 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}

// The hit tests answer in two steps. While the world is scanned they only return the accepted PARAMETER t (and, for a cuboid, the
// face): `*_t`. The record of the one hit that wins (point, normal, u, v: ray.rs:43-50) is built afterwards from (entry, t, face) by
// `*_rec`, with the expressions the reference uses at the hit -- same inputs, same operations, same bits. (Filling a nine-register
// record inside every test and merging it across the tests' exits cost the general-world kernel 22 % of its VALU instructions in
// register copies, and an LDS record of the running best hit.)

// sphere.rs:29-66 with the caller's t_max (the list scan narrows it)
__device__ __forceinline__ bool w_sphere_t(f3 centre, float radius, const WRay &r, float t_min, float t_max, float &t_out) {
    const f3 oc = sub3(r.o, centre);
    const float a = dot3(r.d, r.d);
    const float b = dot3(oc, r.d);
    const float c = dot3(oc, oc) - radius * radius;
    const float discriminant = b * b - a * c;
    if (discriminant > 0.0f) {
        const float ds = sqrt_exact(discriminant);
        float t = (-b - ds) / a;
        if (!(t < t_max && t > t_min)) {
            t = (-b + ds) / a;
            if (!(t < t_max && t > t_min)) return false;
        }
        t_out = t;
        return true;
    }
    return false;
}
__device__ __forceinline__ void w_sphere_rec(f3 centre, float radius, const WRay &r, float t, WHit &h) {
    h.point = add3(r.o, scale3(r.d, t));
    h.normal = divs3(sub3(h.point, centre), radius);
    h.u = 0.0f, h.v = 0.0f;   // sphere.rs:47-48
}

// rect.rs:73-190. `axis` 0/1/2 = XY/XZ/YZ. The comparisons keep the reference's form: a NaN t or
// coordinate falls through every test exactly as it does there.
__device__ __forceinline__ void w_rect_axes(uint32_t axis, const WRay &r, float &ok, float &rk, float &oa, float &da, float &ob, float &db) {
    if (axis == 0u) {
        ok = r.o.z, rk = r.rcp.z, oa = r.o.x, da = r.d.x, ob = r.o.y, db = r.d.y;
    } else if (axis == 1u) {
        ok = r.o.y, rk = r.rcp.y, oa = r.o.x, da = r.d.x, ob = r.o.z, db = r.d.z;
    } else {
        ok = r.o.x, rk = r.rcp.x, oa = r.o.y, da = r.d.y, ob = r.o.z, db = r.d.z;
    }
}
__device__ __forceinline__ bool w_rect_t(uint32_t axis, float a0, float a1, float b0, float b1, float k, const WRay &r, float t_min, float t_max,
                                         float &t_out) {
    float ok, rk, oa, da, ob, db;
    w_rect_axes(axis, r, ok, rk, oa, da, ob, db);
    const float t = (k - ok) * rk;
    const float a = oa + t * da;
    const float b = ob + t * db;
    // (one block, no branch between the range test and the rectangle test: the same comparisons, combined without short-circuit)
    const bool miss = (t < t_min) | (t > t_max) | (a < a0) | (a > a1) | (b < b0) | (b > b1);
    if (miss) return false;
    t_out = t;
    return true;
}
__device__ __forceinline__ void w_rect_rec(uint32_t axis, float a0, float a1, float b0, float b1, bool flip, const WRay &r, float t, WHit &h, bool want_uv) {
    float ok, rk, oa, da, ob, db;
    w_rect_axes(axis, r, ok, rk, oa, da, ob, db);
    const float a = oa + t * da;
    const float b = ob + t * db;
    const float sgn = flip ? -1.0f : 1.0f;  // rect.rs:33 FLIP_SIGN
    h.normal = axis == 0u ? mk3(0.0f, 0.0f, sgn) : (axis == 1u ? mk3(0.0f, sgn, 0.0f) : mk3(sgn, 0.0f, 0.0f));
    h.point = add3(r.o, scale3(r.d, t));
    h.u = want_uv ? (a - a0) / (a1 - a0) : 0.0f;   // rect.rs:97-98
    h.v = want_uv ? (b - b0) / (b1 - b0) : 0.0f;
}

// aabb.rs:46-58 (Vec3A min/max = _mm_min_ps/_mm_max_ps: the SECOND operand wins on NaN), one axis
__device__ __forceinline__ bool w_slab(float mn, float mx, float o, float rc, float tmin, float tmax) {
    const float lo = (mn - o) * rc, hi = (mx - o) * rc;
    const float t0 = sse_min(lo, hi), t1 = sse_max(lo, hi);
    return sse_min(t1, tmax) > sse_max(t0, tmin);
}
__device__ __forceinline__ bool w_aabb_hit(f3 mn, f3 mx, const WRay &r, float tmin, float tmax) {
    const bool x = w_slab(mn.x, mx.x, r.o.x, r.rcp.x, tmin, tmax);
    const bool y = w_slab(mn.y, mx.y, r.o.y, r.rcp.y, tmin, tmax);
    const bool z = w_slab(mn.z, mx.z, r.o.z, r.rcp.z, tmin, tmax);
    return x && y && z;
}

// cuboid.rs:11-37: AABB test, then the six faces in construction order with narrowing. The faces are written out (axis
// and side are compile-time constants of each call): a loop over a face index made the compiler keep p0 / p1 in scratch
// memory and index them dynamically. Face id = 2 * axis + (1 for the face at p0, which looks the other way).
template <int AXIS, bool FLIP>
__device__ __forceinline__ void w_cuboid_face_bounds(f3 p0, f3 p1, float &a0, float &a1, float &b0, float &b1, float &k) {
    if (AXIS == 0) a0 = p0.x, a1 = p1.x, b0 = p0.y, b1 = p1.y, k = FLIP ? p0.z : p1.z;        // XY
    else if (AXIS == 1) a0 = p0.x, a1 = p1.x, b0 = p0.z, b1 = p1.z, k = FLIP ? p0.y : p1.y;   // XZ
    else a0 = p0.y, a1 = p1.y, b0 = p0.z, b1 = p1.z, k = FLIP ? p0.x : p1.x;                   // YZ
}
template <int AXIS, bool FLIP>
__device__ __forceinline__ void w_cuboid_face_t(f3 p0, f3 p1, const WRay &r, float t_min, float &closest, uint32_t &face, bool &found) {
    float a0, a1, b0, b1, k, t;
    w_cuboid_face_bounds<AXIS, FLIP>(p0, p1, a0, a1, b0, b1, k);
    if (w_rect_t((uint32_t)AXIS, a0, a1, b0, b1, k, r, t_min, closest, t)) closest = t, face = 2u * (uint32_t)AXIS + (FLIP ? 1u : 0u), found = true;
}
__device__ __forceinline__ bool w_cuboid_t(f3 p0, f3 p1, const WRay &r, float t_min, float t_max, float &t_out, uint32_t &face) {
    if (!w_aabb_hit(p0, p1, r, t_min, t_max)) return false;
    bool found = false;
    float closest = t_max;
    w_cuboid_face_t<0, false>(p0, p1, r, t_min, closest, face, found);   // odd faces sit at p0 and face the other way
    w_cuboid_face_t<0, true>(p0, p1, r, t_min, closest, face, found);
    w_cuboid_face_t<1, false>(p0, p1, r, t_min, closest, face, found);
    w_cuboid_face_t<1, true>(p0, p1, r, t_min, closest, face, found);
    w_cuboid_face_t<2, false>(p0, p1, r, t_min, closest, face, found);
    w_cuboid_face_t<2, true>(p0, p1, r, t_min, closest, face, found);
    t_out = closest;
    return found;
}
// A ConstantMedium asks its boundary twice (constant_medium.rs:39-43): (-MAX, MAX), then (t_first + 0.0001, MAX). For a Cuboid both
// questions meet the same six planes: the plane parameters, the in-rectangle tests and the slabs of the AABB test (cuboid.rs:11-37,
// rect.rs:73-190, aabb.rs:46-58) do not depend on the range -- they are formed ONCE here, and a question is then the range tests over
// them, with the same comparisons in the same order as w_cuboid_t (ties between faces go to the later one there and here).
struct WCuboidPlanes {
    float t[6];      // plane parameter of face 0..5 (XY, XY', XZ, XZ', YZ, YZ'; ' = the face at p0)
    uint32_t inb;    // bit f: the point at t[f] lies inside face f's rectangle
    float s0[3], s1[3];   // per axis: sse_min / sse_max of the two slab parameters
};
template <int AXIS, bool FLIP>
__device__ __forceinline__ void w_cuboid_plane(f3 p0, f3 p1, const WRay &r, WCuboidPlanes &P) {
    float a0, a1, b0, b1, k, ok, rk, oa, da, ob, db;
    w_cuboid_face_bounds<AXIS, FLIP>(p0, p1, a0, a1, b0, b1, k);
    w_rect_axes((uint32_t)AXIS, r, ok, rk, oa, da, ob, db);
    const float t = (k - ok) * rk;
    const float a = oa + t * da, b = ob + t * db;
    constexpr int f = 2 * AXIS + (FLIP ? 1 : 0);
    P.t[f] = t;
    P.inb |= (a < a0 || a > a1 || b < b0 || b > b1) ? 0u : (1u << f);
}
__device__ __forceinline__ void w_cuboid_slabs(f3 p0, f3 p1, const WRay &r, WCuboidPlanes &P) {
    const float lx = (p0.x - r.o.x) * r.rcp.x, hx = (p1.x - r.o.x) * r.rcp.x, ly = (p0.y - r.o.y) * r.rcp.y, hy = (p1.y - r.o.y) * r.rcp.y;
    const float lz = (p0.z - r.o.z) * r.rcp.z, hz = (p1.z - r.o.z) * r.rcp.z;
    P.s0[0] = sse_min(lx, hx), P.s1[0] = sse_max(lx, hx), P.s0[1] = sse_min(ly, hy), P.s1[1] = sse_max(ly, hy), P.s0[2] = sse_min(lz, hz), P.s1[2] = sse_max(lz, hz);
}
__device__ __forceinline__ bool w_cuboid_box(const WCuboidPlanes &P, float t_min, float t_max) {   // w_aabb_hit over the slabs
    const bool x = sse_min(P.s1[0], t_max) > sse_max(P.s0[0], t_min), y = sse_min(P.s1[1], t_max) > sse_max(P.s0[1], t_min), z = sse_min(P.s1[2], t_max) > sse_max(P.s0[2], t_min);
    return x && y && z;
}
__device__ __forceinline__ void w_cuboid_faces(f3 p0, f3 p1, const WRay &r, WCuboidPlanes &P) {
    P.inb = 0u;
    w_cuboid_plane<0, false>(p0, p1, r, P), w_cuboid_plane<0, true>(p0, p1, r, P);
    w_cuboid_plane<1, false>(p0, p1, r, P), w_cuboid_plane<1, true>(p0, p1, r, P);
    w_cuboid_plane<2, false>(p0, p1, r, P), w_cuboid_plane<2, true>(p0, p1, r, P);
}
// the six faces of w_cuboid_t(p0, p1, r, t_min, t_max, ...) over the planes formed above (the caller has asked w_cuboid_box)
__device__ __forceinline__ bool w_cuboid_ask(const WCuboidPlanes &P, float t_min, float t_max, float &t_out, uint32_t &face) {
    bool found = false;
    float closest = t_max;
#pragma unroll
    for (int f = 0; f < 6; ++f) {
        const float t = P.t[f];
        if (!(t < t_min || t > closest) && ((P.inb >> f) & 1u) != 0u) closest = t, face = (uint32_t)f, found = true;
    }
    t_out = closest;
    return found;
}
__device__ __forceinline__ void w_cuboid_rec(f3 p0, f3 p1, uint32_t face, const WRay &r, float t, WHit &h, bool want_uv) {
    const uint32_t axis = face >> 1;
    const bool flip = (face & 1u) != 0u;
    float a0, a1, b0, b1;
    if (axis == 0u) a0 = p0.x, a1 = p1.x, b0 = p0.y, b1 = p1.y;
    else if (axis == 1u) a0 = p0.x, a1 = p1.x, b0 = p0.z, b1 = p1.z;
    else a0 = p0.y, a1 = p1.y, b0 = p0.z, b1 = p1.z;
    w_rect_rec(axis, a0, a1, b0, b1, flip, r, t, h, want_uv);
}

// The innermost shape (hitable.rs:50-56)
__device__ __forceinline__ f3 w_moving_centre(const pt_hitable &H, const WRay &r) {   // moving_sphere.rs:29-31
    const float s = (r.time - H.p[7]) * H.p[8];
    return add3(mk3(H.p[0], H.p[1], H.p[2]), scale3(mk3(H.p[3], H.p[4], H.p[5]), s));
}
__device__ __forceinline__ bool w_shape_t(const pt_hitable &H, const WRay &r, float t_min, float t_max, float &t, uint32_t &face) {
    switch (H.kind) {
    case PT_HIT_SPHERE: return w_sphere_t(mk3(H.p[0], H.p[1], H.p[2]), H.p[3], r, t_min, t_max, t);
    case PT_HIT_MOVING_SPHERE: return w_sphere_t(w_moving_centre(H, r), H.p[6], r, t_min, t_max, t);   // moving_sphere.rs:38-73
    case PT_HIT_CUBOID: return w_cuboid_t(mk3(H.p[0], H.p[1], H.p[2]), mk3(H.p[3], H.p[4], H.p[5]), r, t_min, t_max, t, face);
    // (one call per plane orientation, each with a constant axis: a run-time axis made the compiler index the ray's
    //  components through scratch memory)
    case PT_HIT_RECT_XY: return w_rect_t(0u, H.p[0], H.p[1], H.p[2], H.p[3], H.p[4], r, t_min, t_max, t);
    case PT_HIT_RECT_XZ: return w_rect_t(1u, H.p[0], H.p[1], H.p[2], H.p[3], H.p[4], r, t_min, t_max, t);
    default: return w_rect_t(2u, H.p[0], H.p[1], H.p[2], H.p[3], H.p[4], r, t_min, t_max, t);
    }
}
__device__ __forceinline__ void w_shape_rec(const pt_hitable &H, const WRay &r, float t, uint32_t face, WHit &h, bool want_uv) {
    switch (H.kind) {
    case PT_HIT_SPHERE: w_sphere_rec(mk3(H.p[0], H.p[1], H.p[2]), H.p[3], r, t, h); break;
    case PT_HIT_MOVING_SPHERE: w_sphere_rec(w_moving_centre(H, r), H.p[6], r, t, h); break;
    case PT_HIT_CUBOID: w_cuboid_rec(mk3(H.p[0], H.p[1], H.p[2]), mk3(H.p[3], H.p[4], H.p[5]), face, r, t, h, want_uv); break;
    case PT_HIT_RECT_XY: w_rect_rec(0u, H.p[0], H.p[1], H.p[2], H.p[3], H.flip_normals != 0u, r, t, h, want_uv); break;
    case PT_HIT_RECT_XZ: w_rect_rec(1u, H.p[0], H.p[1], H.p[2], H.p[3], H.flip_normals != 0u, r, t, h, want_uv); break;
    default: w_rect_rec(2u, H.p[0], H.p[1], H.p[2], H.p[3], H.flip_normals != 0u, r, t, h, want_uv); break;
    }
}

// glam Affine3A::transform_vector3 / transform_point3: ((x_axis*v.x + y_axis*v.y) + z_axis*v.z) [+ translation]
__device__ __forceinline__ f3 w_xf_vector(const float m[12], f3 v) {
    return add3(add3(scale3(mk3(m[0], m[1], m[2]), v.x), scale3(mk3(m[3], m[4], m[5]), v.y)), scale3(mk3(m[6], m[7], m[8]), v.z));
}
__device__ __forceinline__ f3 w_xf_point(const float m[12], f3 p) { return add3(w_xf_vector(m, p), mk3(m[9], m[10], m[11])); }

// pt_hitable.transform (include/ptgpu.h): -1 none; a plain index = ONE Instance around the shape; or, as the scene-graph
// flattener writes it, first index | inner levels << 20 | outer levels << 24 -- `outer` Instances around the ConstantMedium
// (if any), then `inner` ones between it and the shape, outermost first, consecutive in the transform table.
struct WChain {
    uint32_t first, n_in, n_out;
};
__device__ __forceinline__ WChain w_chain(int32_t tf) {
    if (tf < 0) return WChain{0u, 0u, 0u};
    const uint32_t u = (uint32_t)tf;
    if ((u >> 20) == 0u) return WChain{u, 1u, 0u};
    return WChain{u & 0xfffffu, (u >> 20) & 15u, (u >> 24) & 15u};
}

// instance.rs:32-47 (ray.rs:28-40, 52-64), `n` levels deep: each level builds a new Ray from the transformed origin and
// direction (Ray::new recomputes the reciprocal), the innermost hit is carried back out level by level
__device__ __forceinline__ WRay w_ray_into(const pt_affine *xf, uint32_t first, uint32_t n, const WRay &r) {
    WRay local = r;
    for (uint32_t j = 0; j < n; ++j) {
        const pt_affine &T = xf[first + j];
        local = w_ray_new(w_xf_point(T.inv, local.o), w_xf_vector(T.inv, local.d), r.time);
    }
    return local;
}
__device__ __forceinline__ void w_hit_out_of(const pt_affine *xf, uint32_t first, uint32_t n, WHit &h) {
    for (uint32_t j = n; j-- > 0u;) {
        const pt_affine &T = xf[first + j];
        h.point = w_xf_point(T.m, h.point);
        h.normal = w_xf_vector(T.m, h.normal);
    }
}
// CHAINS = false: at most ONE Instance around the shape and none around a medium (every preset; the kernels selected for
// such worlds do not carry the chain loops' registers). An Instance keeps the ray's parameter: the local ray is the transformed
// origin and the transformed, NOT re-normalised direction (instance.rs:32-47).
template <bool CHAINS>
__device__ __forceinline__ WRay w_local_ray(const pt_affine *xf, const WChain &c, const WRay &r) {
    if (c.n_in == 0u) return r;
    if (!CHAINS) {
        const pt_affine &T = xf[c.first];
        return w_ray_new(w_xf_point(T.inv, r.o), w_xf_vector(T.inv, r.d), r.time);
    }
    return w_ray_into(xf, c.first + c.n_out, c.n_in, r);
}
template <bool CHAINS>
__device__ __forceinline__ void w_instanced_rec(const pt_hitable &H, const pt_affine *xf, const WChain &c, const WRay &r, float t, uint32_t face, WHit &h,
                                                bool want_uv) {
    w_shape_rec(H, w_local_ray<CHAINS>(xf, c, r), t, face, h, want_uv);
    if (c.n_in == 0u) return;
    if (!CHAINS) {
        const pt_affine &T = xf[c.first];
        h.point = w_xf_point(T.m, h.point);
        h.normal = w_xf_vector(T.m, h.normal);
        return;
    }
    w_hit_out_of(xf, c.first + c.n_out, c.n_in, h);
}

constexpr uint32_t kFaceMedium = 0xffu;   // `face` of a hit INSIDE a ConstantMedium (its record is the scatter point, not the boundary's)

// One HitableList entry, first step: the accepted parameter. Returns the material index to shade with, or -1 for no hit.
// A ConstantMedium asks its boundary twice (constant_medium.rs:39-43) -- two call sites of the shape code: a two-trip loop around
// one call site carried its state around the back edge for EVERY entry of the list.
// SHARE: a Cuboid that bounds a medium answers both questions from one set of plane parameters (list scans: cornell_smoke +3.5 %; the
// per-lane walk of a BVH world keeps the two calls -- there the shared form measured 5 % slower).
template <bool MEDIA, bool CHAINS, bool SHARE = false>
__device__ __forceinline__ int w_hitable_t(const pt_hitable &H, const pt_affine *xf, const WRay &r_in, float t_min, float t_max, Rng &rng, float &t,
                                           uint32_t &face) {
    const bool medium = MEDIA && H.medium_material >= 0;   // MEDIA = false: the world has no ConstantMedium (its code, and the RNG's liveness across the scan, drop out)
    const WChain chain = w_chain(H.transform);
    // Instances AROUND the medium (scene graphs only; MEDIA kernels): the medium then sees the transformed ray -- its
    // length enters the distance it samples (constant_medium.rs:51) -- and the hit is carried back out at the end
    const bool outer = MEDIA && CHAINS && chain.n_out != 0u;
    const WRay r = outer ? w_ray_into(xf, chain.first, chain.n_out, r_in) : r_in;
    face = 0u;
    const WRay local = w_local_ray<CHAINS>(xf, chain, r);   // (once for both questions of a medium)
    float t_first, t_second;
    if (SHARE && medium && H.kind == PT_HIT_CUBOID) {   // both questions over one set of plane parameters (WCuboidPlanes)
        const f3 p0 = mk3(H.p[0], H.p[1], H.p[2]), p1 = mk3(H.p[3], H.p[4], H.p[5]);
        WCuboidPlanes P;
        w_cuboid_slabs(p0, p1, local, P);
        if (!w_cuboid_box(P, -kMaxT, kMaxT)) return -1;   // (most rays of a frame: the line misses the box, no plane is formed)
        w_cuboid_faces(p0, p1, local, P);
        uint32_t face2 = 0u;
        if (!w_cuboid_ask(P, -kMaxT, kMaxT, t_first, face2)) return -1;
        if (!w_cuboid_box(P, t_first + 0.0001f, kMaxT) || !w_cuboid_ask(P, t_first + 0.0001f, kMaxT, t_second, face2)) return -1;   // constant_medium.rs:41
    } else {
        const bool ok = w_shape_t(H, local, medium ? -kMaxT : t_min, medium ? kMaxT : t_max, t, face);
        if (!medium) return ok ? (int)H.material : -1;
        if (!ok) return -1;
        t_first = t;
        uint32_t face2 = 0u;
        if (!w_shape_t(H, local, t_first + 0.0001f, kMaxT, t_second, face2)) return -1;   // constant_medium.rs:41
    }
    // constant_medium.rs:44-76
    float t1 = t_first, t2 = t_second;
    if (t1 < t_min) t1 = t_min;
    if (t2 > t_max) t2 = t_max;
    if (t1 >= t2) return -1;
    if (t1 < 0.0f) t1 = 0.0f;
    const float ray_length = length3(r.d);
    const float distance_inside_boundary = (t2 - t1) * ray_length;
    const float hit_distance = -(1.0f / H.density) * logf_ref(rng_f32(rng));
    if (hit_distance < distance_inside_boundary) {
        t = t1 + hit_distance / ray_length;
        face = kFaceMedium;
        return H.medium_material;
    }
    return -1;
}
// A ConstantMedium whose BOUNDARY is a HitableList (include/ptgpu.h PT_HIT_MEDIUM_GROUP): the header entry `G[0]` carries the medium
// (phase-function material, density, the Instances AROUND it), the `n` entries behind it are the list's children -- shapes under their own
// Instance levels, no list entries of their own. constant_medium.rs:39-43 asks the boundary twice, and the boundary answers as
// hitable_list.rs:40-56 does: children in order, each with t_max = the closest parameter so far. Then constant_medium.rs:44-76 as above.
// (CHAINS kernels only: worlds with such an entry are selected onto them.)
__device__ __forceinline__ bool w_group_ask(const pt_hitable *G, uint32_t n, const pt_affine *xf, const WRay &r, float t_min, float t_max, float &t_out) {
    bool any = false;
    float closest = t_max;
    for (uint32_t j = 1; j <= n; ++j) {
        const pt_hitable &C = G[j];
        const WChain c = w_chain(C.transform);
        const WRay local = c.n_in != 0u ? w_ray_into(xf, c.first + c.n_out, c.n_in, r) : r;
        float t;
        uint32_t face;
        if (w_shape_t(C, local, t_min, closest, t, face)) closest = t, any = true;
    }
    t_out = closest;
    return any;
}
__device__ __forceinline__ int w_group_t(const pt_hitable *G, uint32_t n, const pt_affine *xf, const WRay &r_in, float t_min, float t_max, Rng &rng, float &t, uint32_t &face) {
    const pt_hitable &H = G[0];
    const WChain chain = w_chain(H.transform);
    const WRay r = chain.n_out != 0u ? w_ray_into(xf, chain.first, chain.n_out, r_in) : r_in;
    face = 0u;
    float t_first, t_second;
    if (!w_group_ask(G, n, xf, r, -kMaxT, kMaxT, t_first)) return -1;
    if (!w_group_ask(G, n, xf, r, t_first + 0.0001f, kMaxT, t_second)) return -1;   // constant_medium.rs:41
    float t1 = t_first, t2 = t_second;   // constant_medium.rs:44-76
    if (t1 < t_min) t1 = t_min;
    if (t2 > t_max) t2 = t_max;
    if (t1 >= t2) return -1;
    if (t1 < 0.0f) t1 = 0.0f;
    const float ray_length = length3(r.d);
    const float distance_inside_boundary = (t2 - t1) * ray_length;
    const float hit_distance = -(1.0f / H.density) * logf_ref(rng_f32(rng));
    if (hit_distance < distance_inside_boundary) {
        t = t1 + hit_distance / ray_length;
        face = kFaceMedium;
        return H.medium_material;
    }
    return -1;
}
__device__ __forceinline__ uint32_t w_group_size(const pt_hitable &H) { return __float_as_uint(H.p[0]); }

// ... second step, for the entry whose hit won: its record at parameter t
template <bool MEDIA, bool CHAINS>
__device__ __forceinline__ void w_hitable_rec(const pt_hitable &H, const pt_affine *xf, const WRay &r_in, float t, uint32_t face, WHit &h, bool want_uv) {
    const WChain chain = w_chain(H.transform);
    const bool outer = MEDIA && CHAINS && chain.n_out != 0u;
    const WRay r = outer ? w_ray_into(xf, chain.first, chain.n_out, r_in) : r_in;
    if (MEDIA && face == kFaceMedium) {
        h.point = add3(r.o, scale3(r.d, t));
        h.normal = mk3(1.0f, 0.0f, 0.0f);  // Vec3::X, arbitrary (constant_medium.rs:67)
        h.u = 0.0f, h.v = 0.0f;
    } else {
        w_instanced_rec<CHAINS>(H, xf, chain, r, t, face, h, want_uv);
    }
    if (outer) w_hit_out_of(xf, chain.first, chain.n_out, h);
    h.t = t;
}

}  // namespace ptdev
#include "pt_graph.h"
namespace ptdev {

// HIT_LDS: the hitable records and transforms are staged in LDS (worlds up to 16 KB: every preset); the list scan
// then reads them at LDS latency instead of waiting on the scalar cache for each entry (58 % of the wave-cycles of
// cornell_smoke were such waits), and BVH mode gathers them per lane from LDS instead of L2.
// OCC: waves per SIMD the kernel is compiled for. 4 (128 VGPRs, a few spills) pays for worlds without noise textures
// whose LDS lets four workgroups share a CU (cornell +7 %, cornell_smoke +11 %), 5 (96 VGPRs, 5-11 spilled) once more where it
// lets five (list worlds at depth 10: cornell +7.5 %, cornell_smoke +9 %); with Perlin noise inlined the spills
// cost more than the fourth wave brings (simple_light -6 %), so those keep 3 (the compiler then uses ~130 VGPRs).
// MEDIA: some hitable is a ConstantMedium (own instantiations: worlds without media do not carry that path's registers).
// CHAINS: some entry sits below more than one Instance level, or below Instances around its medium (scene graphs only).
#ifdef PT_BBPROF   // tools/bbprof.py (BBPROF_UNIT=pt_kernels_world): the instrumented assembly keeps its counter registers above the compiler's
#include "pt_bbprof.h"
#define PT_WBBPROF_ATTR __attribute__((amdgpu_num_sgpr(100)))
#else
#define PT_WBBPROF_ATTR
#endif
// LAZY (worlds with Noise textures): what a Lambertian / Isotropic scatter off a Noise texture pushes on the attenuation stack is the
// hit POINT, not the colour -- the colour (seven octaves of perlin.rs:54-111, 58 % of simple_light's VALU instructions when every hit
// evaluates it) is only formed when the path ends on something that is not black, and then for all such lanes of the wave together,
// spread over its 64 lanes (wave_balanced_turb). A path that ends in black (a dark sky, the depth limit) multiplies every one of its
// attenuations by zero: scene.rs:62-64 gives 0 + a * (+-0) = +0 per level for any finite a, so those colours are never needed.
#ifndef PT_WORLD_TURB_ROUNDS
#define PT_WORLD_TURB_ROUNDS 6
#endif
// GRAPH: the world is a scene graph that does not flatten (pt_graph.h): the scan is the interpreted Hitable::ray_hit of its root.
template <bool BVH, bool HIT_LDS, int OCC = 3, bool MEDIA = true, bool CHAINS = false, bool LAZY = false, bool GRAPH = false>
__global__ __launch_bounds__(kBlock, OCC) PT_WBBPROF_ATTR void pt_world_kernel(const WArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *p = smem;
    float4 *s_pvec = reinterpret_cast<float4 *>(p);
    uint8_t *s_perm = p + (A.has_noise ? 4096 : 0);
    uint32_t *s_turb = reinterpret_cast<uint32_t *>(p + 4096 + 768) + 192u * (threadIdx.x >> 6);   // wave_balanced_turb's words of this wave (LAZY)
    p += A.has_noise ? kWorldNoiseLds : 0;
    int32_t *s_stack = reinterpret_cast<int32_t *>(p);
    p += BVH ? (A.bvh_stack_entries * kBlock * 4) : 0;
    const pt_hitable *s_hit = reinterpret_cast<const pt_hitable *>(p);
    p += HIT_LDS ? A.n_hit * 64u : 0u;
    const pt_affine *s_xf = reinterpret_cast<const pt_affine *>(p);
    p += HIT_LDS ? A.n_xf * 96u : 0u;
    float *s_path = reinterpret_cast<float *>(p);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    if (HIT_LDS) {
        uint4 *dh = reinterpret_cast<uint4 *>(const_cast<pt_hitable *>(s_hit));
        const uint4 *sh = reinterpret_cast<const uint4 *>(A.hit);
        for (uint32_t k = tid; k < A.n_hit * 4u; k += kBlock) dh[k] = sh[k];
        uint4 *dx = reinterpret_cast<uint4 *>(const_cast<pt_affine *>(s_xf));
        const uint4 *sx = reinterpret_cast<const uint4 *>(A.xf);
        for (uint32_t k = tid; k < A.n_xf * 6u; k += kBlock) dx[k] = sx[k];
    }
    if (A.has_noise) {
        for (int k = tid; k < 256; k += kBlock) s_pvec[k] = A.perlin_vec[k];
        for (int k = tid; k < 768; k += kBlock) s_perm[k] = (uint8_t)A.perlin_perm[k];
    }
    __syncthreads();
    PerlinLds pn{s_pvec, s_perm, OCC < 4};
    // (u, v) of a hit only feed Image textures; the 4-waves-per-SIMD instantiations are launched for worlds without them and
    // do not carry the two registers (nor their divisions) at all
    const bool want_uv = OCC < 4 && A.has_image != 0u;
    const pt_hitable *hit = HIT_LDS ? s_hit : A.hit;
    const pt_affine *xf = HIT_LDS ? s_xf : A.xf;
    constexpr uint32_t PS = LAZY ? 4u : 3u;   // words per level of the attenuation stack: colour -- or hit point + the Noise texture's scale
    float *path = A.stack_in_lds ? (s_path + tid) : (A.gstack + (size_t)blockIdx.x * A.max_depth * PS * kBlock + tid);

    bool have = false, exhausted = false, need_cam = true;
    uint32_t pxy = 0, pix_start = 0, sample = 0, depth = 0, nrays = 0;   // pxy = x | local row << 16; pix_start = nrays when the pixel began
    Rng rng{0, 0, 0, 0};
    f3 col = mk3(0.f, 0.f, 0.f);
    WRay ray = w_ray_new(mk3(0.f, 0.f, 0.f), mk3(0.f, 0.f, 1.f), 0.f);
    unsigned long long lazy_mask = 0ull;   // LAZY: bit k = level k of this path holds a point whose Noise colour has not been formed
    bool path_odd = false;                 // some attenuation of this path is not finite (inf / NaN colours): its product with zero is not zero

    for (;;) {
        // ---- refill (same scheme as pt_trace_kernel: one wave-aggregated atomic, 8x8 pixel tiles, batched until
        // A.refill_min lanes are waiting)
        const bool refill_now = __popcll(wave_ballot(!have && !exhausted)) >= (int)A.refill_min || wave_ballot(have) == 0ull;
        if (!have && !exhausted && refill_now) {
            const unsigned long long m = wave_ballot(1);
            const int leader = __ffsll((long long)m) - 1;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(A.work_counter, (uint32_t)__popcll(m));
            base = __shfl(base, leader);
            const uint32_t item = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (item >= A.n_items) {
                exhausted = true;
            } else {
                const uint32_t in = item & (kTilePix - 1u);
                const uint32_t tile = A.tile_order ? A.tile_order[item >> (2u * kTileLog2)] : (item >> (2u * kTileLog2));
                const uint32_t x = (tile % A.tiles_x) * kTileSide + (in & (kTileSide - 1u));
                const uint32_t ly = (tile / A.tiles_x) * kTileSide + (in >> kTileLog2);
                if (x < A.width && ly < A.local_rows) {
                    have = true;
                    pxy = x | (ly << 16);
                    pix_start = nrays;
                    sample = 0;
                    need_cam = true;
                    if (A.phase == 2u) {   // continue the stream and the sum phase 1 parked
                        const uint4 *st = A.px_state + 3u * (size_t)(ly * A.width + x);
                        const uint4 a = st[0], b = st[1], c = st[2];
                        rng.s0 = (uint64_t)a.x | ((uint64_t)a.y << 32), rng.s1 = (uint64_t)a.z | ((uint64_t)a.w << 32);
                        rng.s2 = (uint64_t)b.x | ((uint64_t)b.y << 32), rng.s3 = (uint64_t)b.z | ((uint64_t)b.w << 32);
                        col = mk3(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z));
                    } else {
                        col = mk3(0.f, 0.f, 0.f);
                        const uint32_t px = x, py = ly * A.shard_count + A.shard_index;
                        uint64_t seed = ((uint64_t)px * 1973ull + (uint64_t)py * 9277ull + (uint64_t)A.frame_num * 26699ull) | 1ull;  // scene.rs:99-101
                        if (A.random_seed) {
                            uint64_t hsh = A.seed_base ^ (seed * 0x9e3779b97f4a7c15ULL);
                            seed = splitmix64_next(hsh);
                        }
                        rng_seed_from_u64(rng, seed);
                    }
                }
            }
        }
        // (no `continue` / `break` up here for a wave whose refill brought nothing: the loop's only exit is at its end -- see pt_kernel.h)
        bool terminal = false;
        f3 V = mk3(0.f, 0.f, 0.f);
        if (have) {
            // ---- camera.rs:56-68 + scene.rs:107-108
            if (need_cam) {
                const uint32_t px = pxy & 0xffffu, py = (pxy >> 16) * A.shard_count + A.shard_index;
                const float u = rng_plus(rng, (float)px) * A.inv_nx;
                const float v = rng_plus(rng, (float)py) * A.inv_ny;
                float dx, dy;
                random_in_unit_disk(rng, dx, dy);
                const float rdx = A.cam.lens_radius * dx, rdy = A.cam.lens_radius * dy;
                const f3 offset = add3(scale3(A.cam.u, rdx), scale3(A.cam.v, rdy));
                const float time = A.cam.time0 + rng_f32(rng) * (A.cam.time1 - A.cam.time0);  // camera.rs:59
                const f3 dir = sub3(sub3(add3(add3(A.cam.lower_left_corner, scale3(A.cam.horizontal, u)), scale3(A.cam.vertical, v)),
                                         A.cam.origin),
                                    offset);
                ray = w_ray_new(add3(A.cam.origin, offset), normalize3(dir), time);
                depth = 0;
                need_cam = false;
                lazy_mask = 0ull, path_odd = false;
            }

            // ---- Hitable::ray_hit(ray, MIN_T, MAX_T) on the world (scene.rs:58)
            // The scan keeps (entry, t, face) of the running closest hit -- three registers; the winner's record is built once, after
            // the scan (w_hitable_rec).
            bool found = false;
            uint32_t best_mat = 0, best_k = 0, best_face = 0;
            float best_t = kMaxT;
            GHit gh;
            gh.found = false;
            if (GRAPH) {   // scene.rs:58 on the graph's root (pt_graph.h)
                gh = graph_ray_hit(GraphSrc{A.gnodes, A.gchildren, A.nodes, A.groot}, hit, xf, ray, kMinT, kMaxT, rng,
                                   A.gframes + (size_t)blockIdx.x * kGraphDepth * kGraphFrame * kBlock + tid, want_uv);
                found = gh.found, best_mat = gh.mat;
            } else if (!BVH) {  // hitable_list.rs:40-56
                float closest = kMaxT;
                for (uint32_t k = 0; k < A.n_hit; ++k) {
                    float t;
                    uint32_t face;
                    int m;
                    uint32_t members = 0u;
                    if (MEDIA && CHAINS && hit[k].kind == (uint32_t)PT_HIT_MEDIUM_GROUP) {   // (wave-uniform: a medium around a list, its children behind it)
                        members = w_group_size(hit[k]);
                        m = w_group_t(hit + k, members, xf, ray, kMinT, closest, rng, t, face);
                    } else {
                        m = w_hitable_t<MEDIA, CHAINS, true>(hit[k], xf, ray, kMinT, closest, rng, t, face);
                    }
                    if (m >= 0) {
                        best_k = k, best_face = face, best_t = t, best_mat = (uint32_t)m, found = true;
                        closest = t;
                    }
                    k += members;
                }
            } else {  // bvh.rs:37-62, iterative: lhs subtree, then rhs, both with the original t_max
                int sp = 0;
                s_stack[sp++ * kBlock + tid] = A.bvh_root;
                while (sp > 0) {
                    const int32_t ref = s_stack[--sp * kBlock + tid];
                    if (ref < 0) {
                        float t;
                        uint32_t face;
                        const bool group = MEDIA && CHAINS && hit[~ref].kind == (uint32_t)PT_HIT_MEDIUM_GROUP;
                        const int m = group ? w_group_t(hit + ~ref, w_group_size(hit[~ref]), xf, ray, kMinT, kMaxT, rng, t, face)
                                            : w_hitable_t<MEDIA, CHAINS>(hit[~ref], xf, ray, kMinT, kMaxT, rng, t, face);
                        if (m >= 0) {
                            // bvh.rs:48-53: lhs only when lhs.t < rhs.t -> an equal t goes to the later leaf
                            if (!found || !(best_t < t)) best_k = (uint32_t)~ref, best_face = face, best_t = t, best_mat = (uint32_t)m;
                            found = true;
                        }
                    } else {
                        const pt_bvh_node nd = A.nodes[ref];
                        if (w_aabb_hit(mk3(nd.min[0], nd.min[1], nd.min[2]), mk3(nd.max[0], nd.max[1], nd.max[2]), ray, kMinT, kMaxT)) {
                            s_stack[sp++ * kBlock + tid] = nd.rhs;
                            s_stack[sp++ * kBlock + tid] = nd.lhs;
                        }
                    }
                }
            }

            // ---- scene.rs:49-71 one level of ray_trace
            nrays += 1;
            terminal = true;
            if (!found) {
                if (A.has_sky) {
                    V = A.sky;
                } else {  // scene.rs:40-47
                    const float t = 0.5f * (ray.d.y + 1.0f);
                    const float w1 = 1.0f - t;
                    V = mk3(w1 + (t * 0.5f) * 0.3f, w1 + (t * 0.7f) * 0.3f, w1 + (t * 1.0f) * 0.3f);
                }
            } else {
                const DMat m = A.mats[best_mat];
                WHit bh;
                if (GRAPH) bh = gh.h;
                else w_hitable_rec<MEDIA, CHAINS>(hit[best_k], xf, ray, best_t, best_face, bh, want_uv);
                const f3 point = bh.point, normal = bh.normal, d = ray.d;
                const float best_u = want_uv ? bh.u : 0.0f, best_v = want_uv ? bh.v : 0.0f;
                // Texture::value (texture.rs:74-91); Constant textures were folded into the material record
                auto colour = [&]() __attribute__((always_inline)) -> f3 {   // (always_inline: a closure that is CALLED holds the kernel's argument block by address, which puts all of it into scratch)
                    if (m.pad0 != 0.0f) return mk3(m.a0, m.a1, m.a2);
                    const f3 c = texture_value(A.texs, pn, m.tex, point, best_u, best_v, DImages{A.image_table, A.image_bytes});
                    path_odd = path_odd || !(__builtin_isfinite(c.x) && __builtin_isfinite(c.y) && __builtin_isfinite(c.z));
                    return c;
                };
                // the attenuation of a Lambertian / Isotropic scatter: the colour -- or, LAZY, the point of a Noise texture. Only where the
                // colour is certainly finite (pt_args.h kLazyNoiseReach); a point further out is coloured here and now, and path_odd
                // notices what comes out
                bool lazy = false;
                float lazy_scale = 0.0f;
                auto surface = [&]() __attribute__((always_inline)) -> f3 {
                    if (LAZY && m.pad0 == 0.0f) {
                        const DTex leaf = texture_leaf(A.texs, m.tex, point);
                        if (leaf.kind == PT_TEX_NOISE && __builtin_fabsf(point.x) < kLazyNoiseReach && __builtin_fabsf(point.y) < kLazyNoiseReach && __builtin_fabsf(point.z) < kLazyNoiseReach) {
                            lazy = true, lazy_scale = leaf.scale;
                            return point;
                        }
                    }
                    return colour();
                };
                f3 emitted = mk3(0.f, 0.f, 0.f);  // material.rs:161-167
                if (m.kind == PT_MAT_DIFFUSE_LIGHT) emitted = colour();
                bool scattered = false;
                f3 att = mk3(1.f, 1.f, 1.f), nd = d;
                if (depth < A.max_depth) {
                    if (m.kind == PT_MAT_LAMBERTIAN) {  // material.rs:52-67
                        const f3 target = add3(add3(point, normal), random_unit_vector(rng));
                        att = surface();
                        nd = normalize3(sub3(target, point));
                        scattered = true;
                    } else if (m.kind == PT_MAT_METAL) {  // material.rs:69-89
                        const f3 reflected = reflect3(d, normal);
                        if (dot3(reflected, normal) > 0.0f) {
                            att = mk3(m.a0, m.a1, m.a2);
                            const f3 rs = random_in_unit_sphere(rng);
                            nd = normalize3(add3(reflected, scale3(rs, m.param)));
                            scattered = true;
                        }
                    } else if (m.kind == PT_MAT_DIELECTRIC) {  // material.rs:91-124
                        const float ref_idx = m.param;
                        const float rdotn = dot3(d, normal);
                        f3 outward_normal;
                        float ni_over_nt, cosine;
                        if (rdotn > 0.0f) {
                            cosine = rdotn / length3(d);
                            cosine = sqrt_exact(1.0f - ref_idx * ref_idx * (1.0f - cosine * cosine));
                            outward_normal = neg3(normal);
                            ni_over_nt = ref_idx;
                        } else {
                            cosine = -rdotn / length3(d);
                            outward_normal = normal;
                            ni_over_nt = 1.0f / ref_idx;
                        }
                        f3 refracted;
                        bool use_refract = false;
                        if (refract3(d, outward_normal, ni_over_nt, refracted)) {
                            const float reflect_prob = schlick_ref(cosine, ref_idx);
                            if (rng_f32(rng) > reflect_prob) use_refract = true;
                        }
                        nd = use_refract ? normalize3(refracted) : normalize3(reflect3(d, normal));
                        scattered = true;
                    } else if (m.kind == PT_MAT_ISOTROPIC) {  // material.rs:126-136: direction NOT normalised
                        att = surface();
                        nd = random_in_unit_sphere(rng);
                        scattered = true;
                    }
                }
                if (scattered) {
                    path[(depth * PS + 0) * kBlock] = att.x;
                    path[(depth * PS + 1) * kBlock] = att.y;
                    path[(depth * PS + 2) * kBlock] = att.z;
                    if (LAZY && lazy) path[(depth * PS + 3) * kBlock] = lazy_scale, lazy_mask |= 1ull << depth;   // (launch(): LAZY only with max_depth <= 64)
                    // scene.rs:62-64: emitted + attenuation * deeper. Only DiffuseLight emits and it never scatters
                    // (material.rs:157), so `emitted` is zero on this branch and the fold below adds 0.0f for it.
                    depth += 1;
                    ray = w_ray_new(point, nd, ray.time);
                    terminal = false;
                } else {
                    V = emitted;
                }
            }
        }
        // ---- scene.rs:62-64 unwound: emitted (= 0) + attenuation * deeper, innermost level first. A path that ended in black keeps
        // nothing of its attenuations: 0 + a * (+-0) = +0 at every level for finite a -- no loop, and (LAZY) no Noise colours.
        const bool dark = have && terminal && depth != 0u && A.atts_finite != 0u && !path_odd && V.x == 0.0f && V.y == 0.0f && V.z == 0.0f;
        if (LAZY) {
            // the Noise colours the fold below will read, for all lanes of the wave that end a lit path in this iteration, one level per trip
            bool pend = have && terminal && !dark && lazy_mask != 0ull;
            while (wave_any(pend)) {
                const uint32_t k = pend ? 63u - (uint32_t)__builtin_clzll(lazy_mask) : 0u;
                const f3 q = pend ? mk3(path[(k * PS + 0) * kBlock], path[(k * PS + 1) * kBlock], path[(k * PS + 2) * kBlock]) : mk3(0.f, 0.f, 0.f);
                const float turb = wave_balanced_turb<PT_WORLD_TURB_ROUNDS, false>(pn, s_turb, pend, q);
                if (pend) {
                    DTex leaf{};
                    leaf.kind = PT_TEX_NOISE, leaf.scale = path[(k * PS + 3) * kBlock];
                    const f3 c = texture_leaf_value(leaf, turb, q, 0.0f, 0.0f, DImages{nullptr, nullptr});   // texture.rs:86-88
                    path[(k * PS + 0) * kBlock] = c.x, path[(k * PS + 1) * kBlock] = c.y, path[(k * PS + 2) * kBlock] = c.z;
                    lazy_mask &= ~(1ull << k);
                    pend = lazy_mask != 0ull;
                }
            }
        }
        if (have) {
            if (terminal) {
                if (dark) {
                    V = mk3(0.f, 0.f, 0.f);
                } else {
                    for (int k = (int)depth - 1; k >= 0; --k) {
                        V.x = 0.0f + path[(k * PS + 0) * kBlock] * V.x;
                        V.y = 0.0f + path[(k * PS + 1) * kBlock] * V.y;
                        V.z = 0.0f + path[(k * PS + 2) * kBlock] * V.z;
                    }
                }
                col = add3(col, V);  // scene.rs:110
                sample += 1;
                need_cam = true;
                if (sample == A.samples) {  // scene.rs:113-116
                    const uint32_t x = pxy & 0xffffu, ly = pxy >> 16;
                    if (A.phase == 1u) {   // to be continued by the second launch
                        uint4 *st = A.px_state + 3u * (size_t)(ly * A.width + x);
                        st[0] = make_uint4((uint32_t)rng.s0, (uint32_t)(rng.s0 >> 32), (uint32_t)rng.s1, (uint32_t)(rng.s1 >> 32));
                        st[1] = make_uint4((uint32_t)rng.s2, (uint32_t)(rng.s2 >> 32), (uint32_t)rng.s3, (uint32_t)(rng.s3 >> 32));
                        st[2] = make_uint4(__float_as_uint(col.x), __float_as_uint(col.y), __float_as_uint(col.z), 0u);
                    } else {
                        col = scale3(col, A.inv_ns);
                        float *out = A.rgb + (ly * A.width + x) * 3u;
                        const float p0 = A.prev_zero ? 0.0f : out[0], p1 = A.prev_zero ? 0.0f : out[1], p2 = A.prev_zero ? 0.0f : out[2];
                        out[0] = p0 * A.mix_prev + col.x * A.mix_new;
                        out[1] = p1 * A.mix_prev + col.y * A.mix_new;
                        out[2] = p2 * A.mix_prev + col.z * A.mix_new;
                    }
                    if (A.tile_cost) atomicAdd(&A.tile_cost[(ly >> kTileLog2) * A.tiles_x + (x >> kTileLog2)], nrays - pix_start);
                    have = false;
                }
            }
        }
        if (wave_ballot(have) == 0ull && wave_ballot(!exhausted) == 0ull) break;
    }

    unsigned long long total = nrays;  // scene.rs:118
    for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off);
    if (lane == 0) atomicAdd(A.ray_count, total);
}

}  // namespace ptdev
