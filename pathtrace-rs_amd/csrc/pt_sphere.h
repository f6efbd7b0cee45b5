// pt_sphere.h -- Sphere::ray_hit (sphere.rs:29-66) in its exact forms, and the exact VALU scan of a HitableList (hitable_list.rs:40-56) from LDS / HBM.
#pragma once
#include "pt_args.h"
#include "pt_device.h"

namespace ptdev {

// ---- sphere.rs:29-66 exact slow path for one sphere ---------------------------
// Returns true and narrows `closest` when the sphere is hit in (kMinT, closest).
__device__ __forceinline__ bool sphere_roots(const DivA &av, float b, float disc, float &closest) {
    const float sq = sqrt_exact(disc);
    float t;
    if (av.fast) t = div_by_unit_range(-b - sq, av.a, av.y); else t = (-b - sq) / av.a;   // (wave-uniform branch)
    if (t < closest && t > kMinT) {
        closest = t;
        return true;
    }
    if (av.fast) t = div_by_unit_range(-b + sq, av.a, av.y); else t = (-b + sq) / av.a;
    if (t < closest && t > kMinT) {
        closest = t;
        return true;
    }
    return false;
}
__device__ __forceinline__ bool sphere_roots(float a, float b, float disc, float &closest) { return sphere_roots(DivA{a, 0.0f, false}, b, disc, closest); }

// sphere.rs:38-64 for t_max = f32::MAX WITHOUT a branch: the root the reference accepts, or kMaxT when there is none (`tested`
// false, discriminant <= 0 or NaN, both roots outside (t_min, f32::MAX)). Both quotients are always formed -- the second one is
// needed whenever a ray starts on the sphere it tests, i.e. in nearly every wave -- so that the square root's refinement and the two
// divisions are ONE basic block of independent chains; the inputs the short forms do not cover (pt_device.h: a discriminant below
// 2^-96, a divisor outside [0.5, 2]) are recomputed in full behind one wave-uniform test. Same arithmetic, same result as
// sphere_roots with closest = kMaxT.
__device__ __forceinline__ float sphere_hit_t(const DivA &av, float b, float disc, bool tested) {
    float sq = __builtin_amdgcn_sqrtf(disc);   // sqrt_exact's common path
    {
        const float sm = __uint_as_float(__float_as_uint(sq) - 1u), sp = __uint_as_float(__float_as_uint(sq) + 1u);
        const float rm = __builtin_fmaf(-sm, sq, disc), rp = __builtin_fmaf(-sp, sq, disc);
        sq = (0.0f >= rm) ? sm : sq;
        sq = (0.0f < rp) ? sp : sq;
    }
    float t1 = div_by_unit_range(-b - sq, av.a, av.y), t2 = div_by_unit_range(-b + sq, av.a, av.y);
    const bool ok = tested && disc > 0.0f;
    if (__builtin_expect(!av.fast || wave_any(ok && disc < 0x1p-96f), 0)) {
        const float s2 = __builtin_sqrtf(disc);
        t1 = (-b - s2) / av.a, t2 = (-b + s2) / av.a;
    }
    const bool h1 = ok && t1 < kMaxT && t1 > kMinT;          // sphere.rs:40-49
    const bool h2 = ok && !h1 && t2 < kMaxT && t2 > kMinT;   // sphere.rs:51-60
    return h1 ? t1 : (h2 ? t2 : kMaxT);
}

// hitable_list.rs:40-56 over sphere.rs:29-66, restructured for the GPU in two phases that
// together perform exactly the reference's sequence of accepted hits:
//
//  phase 1 (wave-uniform, branch-free): for every sphere k compute the reference's
//     discriminant with the reference's operation order (sphere.rs:33-37). Lanes whose
//     discriminant is > 0 append k to a per-lane candidate queue in LDS (unconditional
//     ds_write to slot `cnt`, then cnt += pass). Spheres with discriminant <= 0 do
//     nothing in the reference either (sphere.rs:38), so skipping them is exact.
//  phase 2 (per-lane, short): replay the queued spheres IN INDEX ORDER through the exact
//     root / t_min / closest_so_far logic (sphere.rs:38-64, hitable_list.rs:48-54).
//
// The sphere table is read 8 entries at a time (kScanUnroll) so the loads of a group are
// in flight before its arithmetic starts; the table is padded to a multiple of 8 with
// (3e38, 3e38, 3e38, 0) entries whose discriminant is NaN or -inf.

__device__ __forceinline__ void drain_candidates(const float4 *sph, const uint16_t *q, uint32_t &cnt, f3 o, f3 d,
                                                 float a, float &closest, int &idx) {
    for (uint32_t j = 0; wave_any(j < cnt); ++j) {
        if (j < cnt) {
            const int k = q[j * kBlock];
            const float4 c = sph[k];
            const float ocx = o.x - c.x, ocy = o.y - c.y, ocz = o.z - c.z;
            const float b = (ocx * d.x + ocy * d.y) + ocz * d.z;
            const float cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - c.w;
            const float disc = b * b - a * cc;
            if (sphere_roots(a, b, disc, closest)) idx = k;
        }
    }
    cnt = 0;
}

// sph: (cx, cy, cz, r*r) table with n_pad (multiple of kScanUnroll) entries, in LDS or HBM;
// q: this lane's column of the [kQueueCap+1][kBlock] u16 queue in LDS.
// The per-sphere pass/fail is kept as a wave lane mask (v_cmp -> SGPR pair); lanes touch their
// queue only inside the (rare) groups where some lane passed.
__device__ __forceinline__ int intersect_list(const float4 *sph, int n_pad, uint16_t *q, f3 o, f3 d, float a,
                                              float &t_out) {
    float closest = kMaxT;
    int idx = -1;
    uint32_t cnt = 0;
    for (int k0 = 0; k0 < n_pad; k0 += kScanUnroll) {
        float4 c[kScanUnroll];
#pragma unroll
        for (int u = 0; u < kScanUnroll; ++u) c[u] = sph[k0 + u];
        float disc[kScanUnroll];
#pragma unroll
        for (int u = 0; u < kScanUnroll; ++u) {
            const float ocx = o.x - c[u].x, ocy = o.y - c[u].y, ocz = o.z - c[u].z;
            const float b = (ocx * d.x + ocy * d.y) + ocz * d.z;
            const float cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - c[u].w;
            disc[u] = b * b - a * cc;
        }
        // one compare per group: max over the group's discriminants (v_max3; NaNs are ignored by
        // maxNum exactly as `NaN > 0` is false)
        float m = __builtin_fmaxf(__builtin_fmaxf(disc[0], disc[1]), disc[2]);
#pragma unroll
        for (int u = 3; u + 1 < kScanUnroll; u += 2) m = __builtin_fmaxf(__builtin_fmaxf(m, disc[u]), disc[u + 1]);
        if ((kScanUnroll & 1) == 0) m = __builtin_fmaxf(m, disc[kScanUnroll - 1]);
        if (wave_any(m > 0.0f)) {
#pragma unroll
            for (int u = 0; u < kScanUnroll; ++u) {
                if (disc[u] > 0.0f) {
                    q[cnt * kBlock] = (uint16_t)(k0 + u);
                    cnt += 1;
                }
            }
            if (wave_any(cnt > (uint32_t)(kQueueCap - kScanUnroll))) drain_candidates(sph, q, cnt, o, d, a, closest, idx);
        }
    }
    drain_candidates(sph, q, cnt, o, d, a, closest, idx);
    t_out = closest;
    return idx;
}

}  // namespace ptdev
