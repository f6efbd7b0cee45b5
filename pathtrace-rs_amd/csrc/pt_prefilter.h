// pt_prefilter.h -- closest hit of a list world on the wide kernels: MFMA prefilter over lifted ray / sphere features, tile culling, balanced exact phase 2, and the accept rules of BVH worlds (4.3).
#pragma once
#ifndef PT_PAIR_SLOTS
#define PT_PAIR_SLOTS 0   // 1: the wave's pair list holds (owner, tile slot); the sphere index is looked up when the pair is tested (NOTES.md)
#endif
#include "pt_sphere.h"

namespace ptdev {

// ---- MFMA prefilter ------------------------------------------------------------------------
// The line-sphere discriminant of sphere.rs:33-37 is invariant under moving the ray origin along
// the ray, and it is a bilinear form in lifted features:
//     disc = (o'.d - c.d)^2 - a (|o'|^2 - 2 c.o' + |c|^2 - r^2)  =  S(c, r) . R(o', d) + (o'.d)^2 - a |o'|^2
//     S = [cx^2 cy^2 cz^2 cx*cy cx*cz cy*cz cx cy cz |c|^2-r^2]
//     R = [dx^2 dy^2 dz^2 2dxdy 2dxdz 2dydz  2a*o'x-2(o'.d)dx  2a*o'y-2(o'.d)dy  2a*o'z-2(o'.d)dz  -a]
// with c, o' relative to a fixed centre c0 and o' = the point of the ray's line closest to c0 (so all
// magnitudes stay ~ scene radius). S.R for 32 spheres x 32 rays is ONE pair of
// v_mfma_f32_32x32x16_f16 (K = 32 slots: Sh*Rh, Sh*Rl, Sl*Rh with hi/lo-split f16 operands, ~22-bit
// inputs, f32 accumulation). A pair is a CANDIDATE when S.R > a|o'|^2 - (o'.d)^2 - margin; the margin
// bounds every rounding difference between this evaluation and the reference's f32 discriminant
// (DESIGN.md "MFMA prefilter: error budget"), so every sphere whose reference discriminant is > 0
// is a candidate. Candidates are then run through the exact reference arithmetic (phase 2); the
// prefilter never decides a hit, it only discards certain misses.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

struct RayFeat {
    half8 b0[2], b1[2];  // B fragments for ray-half 0 / 1, chunk 0 / 1 (slots 30/31 carry the threshold)
};

__device__ __forceinline__ half8 shfl_xor32(half8 v) {
    union { half8 h; int i[4]; } u, r;
    u.h = v;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.i[k] = __shfl_xor(u.i[k], 32);
    return r.h;
}

__device__ __forceinline__ RayFeat make_ray_features(const float4 *P, f3 o, f3 d, float a, bool active, int lane) {
    const float4 pc = P[2], pm = P[3];   // c0.xyz, rs2 | m0, gamma
    // origin relative to c0, moved along the ray to the point closest to c0 (any point of the line
    // is valid; rounding here only needs to be covered by the margin)
    const f3 ot = mk3(o.x - pc.x, o.y - pc.y, o.z - pc.z);
    const float od0 = __builtin_fmaf(ot.z, d.z, __builtin_fmaf(ot.y, d.y, ot.x * d.x));
    const float s = active ? (-od0 * __builtin_amdgcn_rcpf(a)) : 0.0f;   // (1 ulp is plenty: s only picks the point on the line)
    f3 op = mk3(__builtin_fmaf(s, d.x, ot.x), __builtin_fmaf(s, d.y, ot.y), __builtin_fmaf(s, d.z, ot.z));
    float od = __builtin_fmaf(op.z, d.z, __builtin_fmaf(op.y, d.y, op.x * d.x));
    const float oo = __builtin_fmaf(op.z, op.z, __builtin_fmaf(op.y, op.y, op.x * op.x));
    const float ot2 = __builtin_fmaf(ot.z, ot.z, __builtin_fmaf(ot.y, ot.y, ot.x * ot.x));
    const float margin = a * __builtin_fmaf(pm.y, ot2 + pc.w, pm.x);
    // candidate <=> S.R > thr. The tile GEMM evaluates thr - S.R directly (sphere fragments hold -S, and
    // slots 30/31 hold 1 x thr_hi, 1 x thr_lo), so a candidate is simply a NEGATIVE accumulator.
    float thr = __builtin_fmaf(a, oo, -(od * od)) - margin;
    thr -= 1.0e-6f * __builtin_fabsf(thr);                         // covers the hi/lo f16 representation of thr
    // Rays the f16 features cannot describe: an origin so far away that the margin alone exceeds the feature range
    // (|o - c0| > ~86 000: the reference's own discriminant error is then of that size, and S.R of a reference-positive pair
    // may lie below any threshold f16 can hold), a line passing farther from c0 than f16 can hold (|o'| > 30 000), or a NaN.
    // Such a lane presents the null line through c0 (o' = 0: |S.R| <= 2 a Rs^2 <= 4608 a) with a threshold below that:
    // EVERY prefiltered sphere becomes its candidate and the exact phase 2 decides, as for any other ray. (The floor stays
    // above the -60000 a that a fragment's padding rows evaluate to; phase 2 skips padding rows anyway.)
    const bool far = !(thr >= -50000.0f && oo < 9.0e8f);
    if (far) op = mk3(0.f, 0.f, 0.f), od = 0.0f, thr = -50000.0f;
    thr = __builtin_fminf(thr, 60000.0f);                          // (LOWERING a threshold only adds candidates)
    if (!active) thr = 60000.0f;
    float R[10];
    R[0] = d.x * d.x; R[1] = d.y * d.y; R[2] = d.z * d.z;
    R[3] = 2.0f * d.x * d.y; R[4] = 2.0f * d.x * d.z; R[5] = 2.0f * d.y * d.z;
    const float a2 = 2.0f * a, od2 = 2.0f * od;
    R[6] = __builtin_fmaf(a2, op.x, -od2 * d.x);
    R[7] = __builtin_fmaf(a2, op.y, -od2 * d.y);
    R[8] = __builtin_fmaf(a2, op.z, -od2 * d.z);
    R[9] = -a;
    // hi/lo split of the ten features and the threshold into the 32 f16 slots of a ray: slots 0..9 = hi (x -Sh), 10..19 = lo (x -Sh),
    // 20..29 = hi again (x -Sl), 30 / 31 = thr hi / lo (x 1). The residual MUST be taken against the very f16 value that is stored.
    // (hipcc was observed to round two uses of (_Float16)v differently at exact ties -- RNE for the stored half, RTZ inside a folded
    // residual -- which loses one f16 ulp: each pair of features is therefore converted ONCE, by one v_cvt_pk_f16_f32 whose result is
    // pinned behind an opaque register copy, and both the stored halves and the residuals come from that register.) Slots are
    // consumed in pairs, so a packed pair is a finished dword of a fragment: no packing instructions.
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    typedef float float2v __attribute__((ext_vector_type(2)));
    uint32_t hi[5], lo[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const float2v v = {R[2 * q], R[2 * q + 1]};
        uint32_t hb = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, half2v));
        asm volatile("" : "+v"(hb));
        const half2v h = __builtin_bit_cast(half2v, hb);
        const float2v res = {v.x - (float)h.x, v.y - (float)h.y};
        hi[q] = hb;
        lo[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(res, half2v));
    }
    uint32_t thr_pair;
    {
        unsigned int tb = (unsigned int)__builtin_bit_cast(unsigned short, (_Float16)thr);
        asm volatile("" : "+v"(tb));
        const _Float16 th = __builtin_bit_cast(_Float16, (unsigned short)tb);
        const _Float16 tl = (_Float16)(thr - (float)th);
        thr_pair = tb | ((uint32_t)__builtin_bit_cast(unsigned short, tl) << 16);
    }
    // dwords of the four k-groups: own[chunk][k-half] = slots chunk * 16 + k-half * 8 + 0..7
    union H8 { half8 h; uint32_t u[4]; };
    H8 own00, own01, own10, own11;
    own00.u[0] = hi[0], own00.u[1] = hi[1], own00.u[2] = hi[2], own00.u[3] = hi[3];   // slots 0..7
    own01.u[0] = hi[4], own01.u[1] = lo[0], own01.u[2] = lo[1], own01.u[3] = lo[2];   // slots 8..15
    own10.u[0] = lo[3], own10.u[1] = lo[4], own10.u[2] = hi[0], own10.u[3] = hi[1];   // slots 16..23
    own11.u[0] = hi[2], own11.u[1] = hi[3], own11.u[2] = hi[4], own11.u[3] = thr_pair;   // slots 24..31
    half8 own[2][2];
    own[0][0] = own00.h, own[0][1] = own01.h, own[1][0] = own10.h, own[1][1] = own11.h;
    // B operand of v_mfma_f32_32x32x16_f16: lane l supplies column (ray) l & 31, k-half l >> 5. For the MFMA over rays
    // 0..31 the low lanes supply their own k-half 0 and the high lanes k-half 1 of ray l - 32; for rays 32..63 the
    // low lanes supply k-half 0 of ray l + 32 and the high lanes their own k-half 1. v_permlane32_swap(X = k-half 0,
    // Y = k-half 1) produces exactly that pair: X' = {lo: X[l], hi: Y[l-32]}, Y' = {lo: X[l+32], hi: Y[l]}.
    (void)lane;
    RayFeat f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        union { half8 h; uint32_t u[4]; } x, y, b0, b1;
        x.h = own[c][0];
        y.h = own[c][1];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const auto sw = __builtin_amdgcn_permlane32_swap(x.u[w], y.u[w], false, false);
            b0.u[w] = sw[0];
            b1.u[w] = sw[1];
        }
        f.b0[c] = b0.h;
        f.b1[c] = b1.h;
    }
    return f;
}

// Candidate queue without atomics and without a per-candidate loop: per tile a lane packs the sign bits of its
// accumulators into masks, swaps the partner ray's half with lane ^ 32, and appends the 32-bit mask of ITS OWN
// ray when it is non-zero (~1 candidate per ray per bounce, so most tiles append nothing). Phase 2 walks the
// set bits; bit -> fragment slot (tile*32 + row) -> sphere.
// inclusive prefix sum over the 64 lanes of a wave (row_shr 1/2/4/8 inside each row of 16, then row_bcast 15 and 31)
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}
__device__ __forceinline__ float lane_fetch(uint32_t src_lane, float v) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), __float_as_int(v)));
}

// aabb.rs:46-58 with the SSE min/max NaN rule (second operand on NaN)
__device__ __forceinline__ float sse_min(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float sse_max(float a, float b) { return a > b ? a : b; }

// aabb.rs:46-58 (exact), also returning the entry distance max(t0x, t0y, t0z, t_min) for ordering
__device__ __forceinline__ bool aabb_hit_enter(const float mn[3], const float mx[3], f3 o, f3 rcp, float &t_enter) {
    const float mnx = (mn[0] - o.x) * rcp.x, mny = (mn[1] - o.y) * rcp.y, mnz = (mn[2] - o.z) * rcp.z;
    const float mxx = (mx[0] - o.x) * rcp.x, mxy = (mx[1] - o.y) * rcp.y, mxz = (mx[2] - o.z) * rcp.z;
    const float t0x = sse_min(mnx, mxx), t0y = sse_min(mny, mxy), t0z = sse_min(mnz, mxz);
    const float t1x = sse_max(mnx, mxx), t1y = sse_max(mny, mxy), t1z = sse_max(mnz, mxz);
    const float lox = sse_max(t0x, kMinT), loy = sse_max(t0y, kMinT), loz = sse_max(t0z, kMinT);
    const float hix = sse_min(t1x, kMaxT), hiy = sse_min(t1y, kMaxT), hiz = sse_min(t1z, kMaxT);
    t_enter = fmaxf(fmaxf(lox, loy), loz);
    return (hix > lox) && (hiy > loy) && (hiz > loz);
}

// BVH-world acceptance of a sphere hit (bvh.rs:37-62): the sphere only counts if every ancestor AABB of its leaf in
// the CALLER's tree passes aabb.rs:46-58. Ancestor boxes nest, so the parent's box decides (plus the few ancestors
// recorded in gate_chain above inverted boxes). A.gate == nullptr: list world, every hit counts.
// where the MFMA list kernels read a BVH world's gate boxes and ranks: global memory, or the LDS copy of the wide kernels
struct GateSrc {
    const float4 *gate;
    const uint32_t *rank;
};
__device__ __forceinline__ bool gate_pass_loaded(const KArgs &A, const float4 gmn, const float4 gmx, f3 o, f3 rcp);
__device__ __forceinline__ bool gate_pass(const KArgs &A, int k, f3 o, f3 rcp) {
    return gate_pass_loaded(A, A.gate[2 * k], A.gate[2 * k + 1], o, rcp);
}
__device__ __forceinline__ bool gate_pass_from(const KArgs &A, const GateSrc &G, int k, f3 o, f3 rcp) {
    return gate_pass_loaded(A, G.gate[2 * k], G.gate[2 * k + 1], o, rcp);
}
__device__ __forceinline__ bool gate_pass_loaded(const KArgs &A, const float4 gmn, const float4 gmx, f3 o, f3 rcp) {
    const float mn[3] = {gmn.x, gmn.y, gmn.z}, mx[3] = {gmx.x, gmx.y, gmx.z};
    float te;
    const uint32_t extra = __float_as_uint(gmn.w);
    bool pass = extra != 0xffffffffu && aabb_hit_enter(mn, mx, o, rcp, te);
    if (pass && extra != 0u) {   // rare: ancestors above an inverted (negative-radius) box
        const float4 *ch = A.gate_chain + 2u * __float_as_uint(gmx.w);
        for (uint32_t j = 0; j < extra && pass; ++j) {
            const float4 cmn = ch[2 * j], cmx = ch[2 * j + 1];
            const float bmn[3] = {cmn.x, cmn.y, cmn.z}, bmx[3] = {cmx.x, cmx.y, cmx.z};
            pass = aabb_hit_enter(bmn, bmx, o, rcp, te);
        }
    }
    return pass;
}

// One accepted-hit rule for both worlds: smaller t wins; equal t goes to the higher RANK, which is the DFS position of
// the leaf in a BVH world (bvh.rs:47-53: `lhs.t < rhs.t ? lhs : rhs`) and ~index in a list world (hitable_list.rs:48:
// the earlier entry keeps an equal t).
// GATED = false compiles the list-world rule alone (no rank register, no gate code in the hot list kernel).
template <bool GATED>
__device__ __forceinline__ void accept_hit(const KArgs &A, const GateSrc &G, int k, float t, f3 o, f3 d, float &best, int &idx, uint32_t &best_rank) {
    if (!GATED) {
        if (idx < 0 || t < best || (t == best && k < idx)) {
            best = t;
            idx = k;
        }
        return;
    }
    const uint32_t rank = G.rank[k];
    if (idx < 0 || t < best || (t == best && rank > best_rank)) {
        if (gate_pass_from(A, G, k, o, mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z))) {   // ray.rs:14 rcp_direction
            best = t;
            idx = k;
            best_rank = rank;
        }
    }
}

// MovingSphere::centre (moving_sphere.rs:29-31): centre_start + ((time - time_start) * inv_time_delta) * centre_delta.
// `c` carries the sphere as stored (centre_start in xyz; w untouched). Plain spheres are returned as they are.
template <bool MOVING>
__device__ __forceinline__ float4 sphere_at_m(const float4 *motion, int k, float4 c, float time) {
    if (MOVING) {
        const float4 m0 = motion[2 * k], m1 = motion[2 * k + 1];
        if (m1.y != 0.0f) {
            const float s = (time - m1.x) * m0.w;
            c.x = c.x + s * m0.x;
            c.y = c.y + s * m0.y;
            c.z = c.z + s * m0.z;
        }
    }
    return c;
}
template <bool MOVING>
__device__ __forceinline__ float4 sphere_at(const KArgs &A, int k, float4 c, float time) {
    return sphere_at_m<MOVING>(A.motion, k, c, time);
}

// exact reference test of one sphere, order independent: candidate t as sphere.rs:38-64 would
// return it for t_max = f32::MAX, winner = lexicographic (t, index) minimum == the sequential
// closest_so_far scan of hitable_list.rs:40-56 (DESIGN.md "order-independent closest hit")
template <bool GATED>
__device__ __forceinline__ void exact_candidate(const KArgs &A, const GateSrc &G, const float4 c, int k, f3 o, f3 d, const DivA &av, float &best, int &idx,
                                                uint32_t &best_rank) {
    const float a = av.a;
    const float ocx = o.x - c.x, ocy = o.y - c.y, ocz = o.z - c.z;
    const float b = (ocx * d.x + ocy * d.y) + ocz * d.z;
    const float cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - c.w;
    const float disc = b * b - a * cc;
    if (disc > 0.0f) {
        float t = kMaxT;
        if (sphere_roots(av, b, disc, t)) accept_hit<GATED>(A, G, k, t, o, d, best, idx, best_rank);
    }
}

// OR over the 64 lanes of a wave, returned wave-uniform (four DPP steps inside each row of 16, then one lane per row)
__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
    v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true);  // row_half_mirror
    v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, true);  // row_mirror
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) | (uint32_t)__builtin_amdgcn_readlane((int)v, 16) |
           (uint32_t)__builtin_amdgcn_readlane((int)v, 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

// Tiles this lane's ray can still find a WINNING hit in. A hit on a sorted sphere lies inside the (padded) box of
// the sorted spheres and at t in (t_min, t_end], t_end = the nearest exact hit known so far (the always-tested large
// spheres, e.g. the ground) -- a farther hit cannot be the closest one. The ray is clipped to that box and range;
// the extent of the clipped segment along the sort axis selects the tiles whose own extent overlaps it (two table
// lookups on a grid of kCullCells cells). Approximate reciprocals are fine: every bound is padded far beyond their error, and a
// NaN anywhere yields "no tile", which is what the reference's `discriminant > 0` does with such a ray as well.
//
// How far the reference's f32 arithmetic can place a hit OUTSIDE a sorted sphere depends on where the RAY starts: its
// discriminant (sphere.rs:33-37) carries an error of <= ~1.3e-6 a (|o - c|^2 + r^2), so from |o - c| = 2000
// it accepts lines passing ~2 units outside a sphere of radius 0.2, and the accepted point o + t d then lies within
// sqrt(r^2 + E) of the centre. The clip box and the segment's extent along the sort axis are therefore padded PER RAY by
//     reach = sqrt(r_min^2 + 4 * 1.3e-6 * (D^2 + r_max^2)) - r_min,   D^2 = 2 |o - c0|^2 + 2 Rs^2 >= (|o - c0| + Rs)^2 >= |o - c|^2
// (P[13] = 2 kappa, kappa (2 Rs^2 + r_max^2) + r_min^2, r_min; kappa = 5.2e-6, the same 4x safety as the tree kernels' box
// pad). A bounce off a huge enclosing or ground sphere far from the cloud thus widens its own mask -- up to every tile --
// whatever the camera's position.
// (in two parts: the ray against the padded box depends on nothing but the ray, and is computed next to the ray's features and the
//  first always-tested sphere -- three independent chains in one basic block; only the few instructions of the second part wait
//  for t_end, the nearest hit on the always-tested spheres)
struct TileClip {
    float t0, t1, reach;
    bool inside;
};
__device__ __forceinline__ TileClip lane_tile_clip(const float4 *P, f3 o, f3 d, bool active) {
    const float4 bmin = P[0], bmax = P[1];   // clip_min.xyz, cull_u0 | clip_max.xyz, cull_inv_cell
    const float4 pc = P[2], pr = P[13];      // c0.xyz | reach constants
    const float otx = o.x - pc.x, oty = o.y - pc.y, otz = o.z - pc.z;
    const float ot2 = __builtin_fmaf(otz, otz, __builtin_fmaf(oty, oty, otx * otx));
    const float reach = __builtin_amdgcn_sqrtf(__builtin_fmaf(pr.x, ot2, pr.y)) * 1.000001f - pr.z;
    float t0 = 0.0f, t1 = kMaxT;
    bool inside = active;
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    const float mn[3] = {bmin.x - reach, bmin.y - reach, bmin.z - reach}, mx[3] = {bmax.x + reach, bmax.y + reach, bmax.z + reach};
#pragma unroll
    for (int k = 0; k < 3; ++k) {   // branch-free slabs: an axis the ray (nearly) does not move along only asks "inside?"
        const bool flat = !(__builtin_fabsf(dd[k]) > 1.0e-12f);
        const float inv = __builtin_amdgcn_rcpf(flat ? 1.0f : dd[k]);
        const float ta = (mn[k] - oo[k]) * inv, tb = (mx[k] - oo[k]) * inv;
        t0 = __builtin_fmaxf(t0, flat ? 0.0f : __builtin_fminf(ta, tb));
        t1 = __builtin_fminf(t1, flat ? kMaxT : __builtin_fmaxf(ta, tb));
        inside = inside && (!flat || (oo[k] >= mn[k] && oo[k] <= mx[k]));
    }
    return TileClip{t0, t1, reach, inside};
}
__device__ __forceinline__ uint32_t lane_tile_mask_of(const float4 *P, const uint32_t *s_cull, const TileClip &c, f3 o, f3 d, float t_end, uint32_t cull_axis,
                                                      uint32_t cull_always) {
    const float4 bmin = P[0], bmax = P[1];
    float t0 = c.t0, t1 = __builtin_fminf(c.t1, t_end * 1.00001f + 1.0e-5f);
    const float slack = 1.0e-3f * (1.0f + t1);     // relative to the distance travelled: covers rcp and f32 rounding
    t0 = t0 - slack, t1 = t1 + slack;
    const float ou = cull_axis == 0u ? o.x : (cull_axis == 1u ? o.y : o.z), du = cull_axis == 0u ? d.x : (cull_axis == 1u ? d.y : d.z);
    const float ua = ou + t0 * du, ub = ou + t1 * du;
    const float pad = 1.0e-3f + c.reach;
    const float lo = __builtin_fminf(ua, ub) - pad, hi = __builtin_fmaxf(ua, ub) + pad;
    const float cl = __builtin_fminf(__builtin_fmaxf((lo - bmin.w) * bmax.w, 0.0f), (float)(kCullCells - 1));
    const float ch = __builtin_fminf(__builtin_fmaxf((hi - bmin.w) * bmax.w, 0.0f), (float)(kCullCells - 1));
    // the same along the second axis of the tiles' boxes (tables of all ones when the scene has a single strip)
    const float4 p2 = P[14];   // cull_u0_2, cull_inv_cell_2, cull_axis2
    const uint32_t axis2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(p2.z));
    const float ov = axis2 == 0u ? o.x : (axis2 == 1u ? o.y : o.z), dv = axis2 == 0u ? d.x : (axis2 == 1u ? d.y : d.z);
    const float va = ov + t0 * dv, vb = ov + t1 * dv;
    const float lo2 = __builtin_fminf(va, vb) - pad, hi2 = __builtin_fmaxf(va, vb) + pad;
    const float cl2 = __builtin_fminf(__builtin_fmaxf((lo2 - p2.x) * p2.y, 0.0f), (float)(kCullCells - 1));
    const float ch2 = __builtin_fminf(__builtin_fmaxf((hi2 - p2.x) * p2.y, 0.0f), (float)(kCullCells - 1));
    const uint32_t tiles = (s_cull[(int)cl] & s_cull[kCullCells + (int)ch]) & (s_cull[2 * kCullCells + (int)cl2] & s_cull[3 * kCullCells + (int)ch2]);
    return ((c.inside && t0 <= t1 && lo <= hi && lo2 <= hi2) ? tiles : 0u) | cull_always;
}
__device__ __forceinline__ uint32_t lane_tile_mask(const float4 *P, const uint32_t *s_cull, f3 o, f3 d, bool active, float t_end,
                                                   uint32_t cull_axis, uint32_t cull_always) {
    return lane_tile_mask_of(P, s_cull, lane_tile_clip(P, o, d, active), o, d, t_end, cull_axis, cull_always);
}

template <bool VERIFY, bool MOVING, bool GATED, int BLK>
__device__ __forceinline__ int intersect_list_mfma(const KArgs &A, const GateSrc &G, const float4 *mot, const float4 *P, const float4 *sph, const uint4 *s_afrag,
                                                   const uint16_t *s_tile_sphere, const uint32_t *s_cull, uint16_t *queue,
                                                   uint32_t *w_pairs, unsigned long long *w_keys,
                                                   f3 o, f3 d, const DivA &av, bool active, float time, float &t_out,
                                                   unsigned long long *sec = nullptr) {
    const float a = av.a;
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef PT_SECTIONS
    unsigned long long sub_last = __builtin_readcyclecounter();
#define PT_SUB(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); sec[i] += now_ - sub_last; sub_last = now_; } while (0)
#else
    (void)sec;
#define PT_SUB(i) do { } while (0)
#endif
    const f3 rcp_own = GATED ? mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z) : mk3(0.f, 0.f, 0.f);   // ray.rs:14 (only the gate of a BVH world reads it)
    const RayFeat rf = make_ray_features(P, o, d, a, active, lane);
    PT_SUB(5);
    const float16v zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // Candidates of MY ray, one 32-bit mask per tile that has any: bits 0..15 come from my own accumulators (my
    // rows of the tile), bits 16..31 from lane ^ 32's (the other 16 rows), exchanged with one cross-half swap per
    // tile. Non-empty masks are appended to this lane's queue; `tbits` remembers which tiles they belong to.
    uint32_t tbits = 0, cnt = 0, ncand = 0;
    uint32_t queued = 0;   // candidates behind the masks this lane has queued since the last drain (counted as they are queued: the
                           // drain's prefix sum needs no pass over the queue)
    uint32_t *queue32 = reinterpret_cast<uint32_t *>(queue);
    // slot of bit b of a tile mask. (Bit b comes from accumulator register r = 15 - (b & 15), i.e. fragment row (r & 3) + 8 (r >> 2)
    // + 4 * (half of the wave that computed it), bits 0..15 from the low half, 16..31 from the high half; the host stores
    // tile_sphere in BIT order -- pt_args.h tile_bit_of_row -- so the lookup in the per-bit loops below needs no arithmetic.)
    auto slot_of = [&](uint32_t T, uint32_t b) -> uint32_t { return T * 32u + b; };
    float best = kMaxT;
    int idx = -1;
    uint32_t best_rank = 0;
    // phase 2 on the queued masks: exact arithmetic for every set bit, then the queue is empty again
    // one exact test: the candidate t as sphere.rs:38-64 returns it for t_max = f32::MAX, reduced into the owner's key.
    // Key = (bits of t, tie-break): smaller t wins; on equal t the lower list index (hitable_list.rs:48) or, in a BVH
    // world, the higher DFS rank (bvh.rs:47-53) -- the same order-independent rule as accept_hit.
    auto key_of = [&](float t, int k, uint32_t rank) -> unsigned long long {
        const uint32_t low = GATED ? (((0xffffu - rank) << 16) | (uint32_t)k) : (uint32_t)k;
        return ((unsigned long long)__float_as_uint(t) << 32) | low;
    };
    auto drain = [&]() {
        // 1. how many candidates does the wave hold, and where do mine go in its list
        const uint32_t mine_n = queued;
        const uint32_t incl = wave_inclusive_sum(mine_n);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
#ifdef PT_WAVEDBG
        if (sec) sec[1] += total, sec[3] += 1, sec[2] += total > (uint32_t)kPairCap ? 1 : 0;
#endif
        if (total > (uint32_t)kPairCap) {
            // more pairs than the list holds (rays far outside the prefilter's accuracy range): every lane walks its own
            uint32_t tb = tbits, j = 0, cur = 0, curT = 0;
            while (wave_any((cur | tb) != 0u)) {
                if (cur == 0u && tb != 0u) {  // next non-empty tile of my ray
                    curT = (uint32_t)__builtin_ctz(tb);
                    tb &= tb - 1u;
                    cur = queue32[j * BLK + tid];
                    j += 1;
                }
                if (cur != 0u) {
                    const uint32_t b = (uint32_t)__builtin_ctz(cur);
                    cur &= cur - 1u;
                    const int k = s_tile_sphere[slot_of(curT, b)];
                    if (k != 0xffff)   // (a padding row of the fragment: flagged only by rays with a = d.d well below 1)
                        exact_candidate<GATED>(A, G, sphere_at_m<MOVING>(mot, k, sph[k], time), k, o, d, av, best, idx, best_rank);
                }
            }
        } else if (total != 0u) {
            // 2. expand my masks into (owner lane, sphere) pairs at my offset of the wave's list
            {
                uint32_t tb = tbits, j = 0, pos = incl - mine_n;
                while (tb != 0u) {
                    const uint32_t T = (uint32_t)__builtin_ctz(tb);
                    tb &= tb - 1u;
                    uint32_t mk = queue32[j * BLK + tid];
                    j += 1;
                    while (mk != 0u) {
                        const uint32_t b = (uint32_t)__builtin_ctz(mk);
                        mk &= mk - 1u;
#if PT_PAIR_SLOTS
                        w_pairs[pos++] = ((uint32_t)lane << 16) | slot_of(T, b);   // (the sphere behind the slot is looked up by the lane that tests the pair: no dependent LDS read in this serial loop)
#else
                        w_pairs[pos++] = ((uint32_t)lane << 16) | (uint32_t)s_tile_sphere[slot_of(T, b)];
#endif
                    }
                }
            }
            const unsigned long long key0 = idx < 0 ? ~0ull : key_of(best, idx, best_rank);
            w_keys[lane] = key0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // 3. one pair per lane and round, with the owner's ray fetched across lanes
            for (uint32_t base = 0; base < total; base += 64u) {
#if PT_PAIR_SLOTS
                const bool listed = base + (uint32_t)lane < total;
                const uint32_t e = listed ? w_pairs[base + lane] : 0u;
                const uint32_t owner = e >> 16;
                const int k = listed ? (int)s_tile_sphere[e & 0xffffu] : 0xffff;
#else
                const uint32_t e = base + (uint32_t)lane < total ? w_pairs[base + lane] : 0xffffu;
                const uint32_t owner = e >> 16;
                const int k = (int)(e & 0xffffu);
#endif
                const bool valid = k != 0xffff;   // (beyond the list, or a padding row of a fragment)
                const f3 po = mk3(lane_fetch(owner, o.x), lane_fetch(owner, o.y), lane_fetch(owner, o.z));
                const f3 pd = mk3(lane_fetch(owner, d.x), lane_fetch(owner, d.y), lane_fetch(owner, d.z));
                const float pa = lane_fetch(owner, a);
                const DivA pav{pa, lane_fetch(owner, av.y), av.fast};
                const float ptime = MOVING ? lane_fetch(owner, time) : 0.0f;
                // (BVH worlds: 1 / d of the owner's ray for the gate test, fetched instead of three IEEE divisions per round)
                const f3 prcp = GATED ? mk3(lane_fetch(owner, rcp_own.x), lane_fetch(owner, rcp_own.y), lane_fetch(owner, rcp_own.z)) : mk3(0.f, 0.f, 0.f);
                if (valid) {
                    const float4 c = sphere_at_m<MOVING>(mot, k, sph[k], ptime);
                    const float ocx = po.x - c.x, ocy = po.y - c.y, ocz = po.z - c.z;
                    const float b = (ocx * pd.x + ocy * pd.y) + ocz * pd.z;
                    const float cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - c.w;
                    const float disc = b * b - pa * cc;
                    const float t = sphere_hit_t(pav, b, disc, true);
                    if (t < kMaxT) {
                        const uint32_t rank = GATED ? G.rank[k] : 0u;
                        if (!GATED || gate_pass_from(A, G, k, po, prcp))   // ray.rs:14 rcp_direction
                            atomicMin(&w_keys[owner], key_of(t, k, rank));
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // 4. my ray's winner
            const unsigned long long key = w_keys[lane];
            if (key != key0) {
                best = __uint_as_float((uint32_t)(key >> 32));
                idx = (int)((uint32_t)key & 0xffffu);
                if (GATED) best_rank = 0xffffu - (((uint32_t)key >> 16) & 0xffffu);
            }
        }
        tbits = 0;
        cnt = 0;
        queued = 0;
    };
    // the always-tested spheres first (wave-uniform): their nearest hit bounds the segment the tiles are culled against
    const float4 pcull = P[12];
    const uint32_t cull_axis = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(pcull.x)), cull_always = __float_as_uint(pcull.y);
    const bool culling = !VERIFY && cull_axis < 3u;
    uint32_t j_first = 0;
    TileClip clip{0.0f, 0.0f, 0.0f, false};
    {
        // The FIRST always-tested sphere (the ground of most scenes) without a branch, so that its chain -- load,
        // discriminant, square root, two quotients -- shares one basic block with the ray's features above and the box clip of the
        // tile culling: three independent chains for the scheduler instead of one after the other (this stretch was 16 % of the
        // wave-cycles for 10 % of the instructions). The arithmetic is sphere.rs:33-64 as everywhere else; the rare inputs the
        // short square root / quotients do not cover are recomputed in full behind ONE wave-uniform test at the end.
        if (culling) clip = lane_tile_clip(P, o, d, active);
        const bool has0 = A.large0 != 0xffffffffu;
        const int k0 = has0 ? (int)A.large0 : 0;
        const float4 c = sphere_at_m<MOVING>(mot, k0, sph[k0], time);
        const float ocx = o.x - c.x, ocy = o.y - c.y, ocz = o.z - c.z;
        const float b = (ocx * d.x + ocy * d.y) + ocz * d.z;
        const float cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - c.w;
        const float disc = b * b - a * cc;
        const float t = sphere_hit_t(av, b, disc, has0 && active);
        if (!GATED) {
            best = t;
            idx = t < kMaxT ? k0 : -1;
        } else if (t < kMaxT) {
            accept_hit<GATED>(A, G, k0, t, o, d, best, idx, best_rank);   // BVH world: the ancestor-AABB gate decides (bvh.rs:37-62)
        }
        j_first = 1;
    }
    for (uint32_t j = j_first; j < A.n_large; ++j) {
        const int k = (int)A.large[j];
        if (active) exact_candidate<GATED>(A, G, sphere_at_m<MOVING>(mot, k, sph[k], time), k, o, d, av, best, idx, best_rank);
    }
    // wave-uniform set of tiles to run: the union of the lanes' tile masks (verify mode audits every tile)
    uint32_t rem = A.n_tiles >= 32u ? 0xffffffffu : ((1u << A.n_tiles) - 1u);
    uint32_t mine = rem;   // tiles THIS lane's ray can find its winner in; the wave runs the union
    if (culling) {
        mine = lane_tile_mask_of(P, s_cull, clip, o, d, best, cull_axis, cull_always);
        rem = wave_or(mine);
#ifdef PT_CULLSTATS
        // development aid: debug[24] wave-iterations, [25] tiles run, [26] active lanes, [27] tiles the lanes asked for,
        // [28 + min(n, 17)] histogram of tiles run per wave-iteration, [48 + min(n, 17)] of tiles asked for per lane
        {   // the union per quarter / half of the wave (what v_mfma_f32_16x16x32 tiles with per-16-ray masks could skip): debug[100] / [101]
            uint32_t v = mine;
            v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);
            v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);
            v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true);
            v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, true);
            const uint32_t q0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), q1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16),
                           q2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), q3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
            if (lane == 0) {
                atomicAdd(&A.debug[100], (unsigned long long)(__popc(q0) + __popc(q1) + __popc(q2) + __popc(q3)));
                atomicAdd(&A.debug[101], (unsigned long long)(__popc(q0 | q1) + __popc(q2 | q3)));
            }
        }
        {   // how many lanes ask for each tile the wave runs: debug[113 + bucket], buckets 1, 2, 3, 4, 5-8, 9-16, 17-32, 33-64 lanes
            uint32_t tb = rem;
            while (tb != 0u) {
                const uint32_t T = (uint32_t)__builtin_ctz(tb);
                tb &= tb - 1u;
                const uint32_t n = (uint32_t)__popcll(wave_ballot(active && ((mine >> T) & 1u) != 0u));
                const uint32_t bucket = n <= 4u ? (n ? n - 1u : 0u) : (n <= 8u ? 4u : (n <= 16u ? 5u : (n <= 32u ? 6u : 7u)));
                if (lane == 0) atomicAdd(&A.debug[113u + bucket], 1ull);
            }
        }
        if (lane == 0) {
            atomicAdd(&A.debug[24], 1ull);
            atomicAdd(&A.debug[25], (unsigned long long)__popc(rem));
            atomicAdd(&A.debug[28 + (__popc(rem) < 17 ? __popc(rem) : 17)], 1ull);
        }
        if (active) {
            atomicAdd(&A.debug[26], 1ull);
            atomicAdd(&A.debug[27], (unsigned long long)__popc(mine));
            atomicAdd(&A.debug[48 + (__popc(mine) < 17 ? __popc(mine) : 17)], 1ull);
        }
#endif
    }
    union Frag { uint4 u; half8 h; };
    Frag a0, a1;
    {
        const uint32_t T0 = rem ? (uint32_t)__builtin_ctz(rem) : 0u;
        a0.u = s_afrag[(T0 * 2 + 0) * 64 + lane];
        a1.u = s_afrag[(T0 * 2 + 1) * 64 + lane];
    }
    while (rem != 0u) {
        const uint32_t T = (uint32_t)__builtin_ctz(rem);
        rem &= rem - 1u;
#ifdef PT_WAVEDBG
        if (sec) sec[0] += 1;
#endif
        const uint32_t Tn = rem ? (uint32_t)__builtin_ctz(rem) : T;   // the next tile's fragments are fetched under this one
        float16v acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0.h, rf.b0[0], zero, 0, 0, 0);
        float16v acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0.h, rf.b1[0], zero, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);   // each fragment is reloaded in place right after its last use (no register copies)
        a0.u = s_afrag[(Tn * 2 + 0) * 64 + lane];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1.h, rf.b0[1], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1.h, rf.b1[1], acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a1.u = s_afrag[(Tn * 2 + 1) * 64 + lane];
        // sign bits of the 2 x 16 accumulators -> 16-bit masks (v_alignbit shifts a sign in): register r ends up
        // at bit 15 - r
        uint32_t m0 = 0, m1 = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            m0 = __builtin_amdgcn_alignbit(m0, __float_as_uint(acc0[r]), 31);
            m1 = __builtin_amdgcn_alignbit(m1, __float_as_uint(acc1[r]), 31);
        }
        // acc0 serves ray (lane & 31), acc1 ray 32 + (lane & 31). v_permlane32_swap exchanges lanes 32..63 of m0
        // with lanes 0..31 of m1: afterwards m0 holds, in EVERY lane, the bits of that lane's own ray computed by the
        // low half of the wave (rows +0) and m1 those computed by the high half (rows +4).
        const auto sw = __builtin_amdgcn_permlane32_swap(m0, m1, false, false);
        const uint32_t full = sw[0] | (sw[1] << 16);
        // candidates in a tile the lane did not ask for (run for another lane's sake) are behind the ray's origin or
        // beyond its nearest hit so far: dropped here instead of going through phase 2
        if (full != 0u && ((mine >> T) & 1u)) {
            if (cnt < (uint32_t)kEntCap) {   // (only verify mode can get past the capacity: everyone else drains when full)
                queue32[cnt * BLK + tid] = full;
                tbits |= 1u << T;
                queued += (uint32_t)__popc(full);
            }
            cnt += 1;
            if (VERIFY) ncand += (uint32_t)__popc(full);
        }
        // a full queue is drained on the spot (exact phase 2 on what is queued so far); verify mode keeps
        // everything for its end-of-scan audit and treats an overflow as "every sphere is a candidate"
        if (!VERIFY && wave_any(cnt >= (uint32_t)kEntCap)) drain();
    }
    PT_SUB(6);
    // ---- phase 2: exact arithmetic on the candidates of this lane's own ray ----
    const bool overflow = VERIFY && cnt > (uint32_t)kEntCap;
    float vbest = kMaxT;   // verify mode: the brute-force winner
    int vidx = -1;
    uint32_t vrank = 0;
    if (wave_any(overflow || (VERIFY && active))) {
        if (overflow || VERIFY) {
            // verify mode (and its queue overflows, where the masks of a ray were not all kept): brute force
            for (int k = 0; k < (int)A.n_spheres; ++k) {
                const float4 c = sphere_at_m<MOVING>(mot, k, sph[k], time);
                exact_candidate<GATED>(A, G, c, k, o, d, av, vbest, vidx, vrank);
                if (VERIFY && active) {
                    const float ocx = o.x - c.x, ocy = o.y - c.y, ocz = o.z - c.z;
                    const float b = (ocx * d.x + ocy * d.y) + ocz * d.z;
                    const float cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - c.w;
                    if (b * b - a * cc > 0.0f) {
                        bool found = overflow;
                        for (uint32_t j = 0; j < A.n_large && !found; ++j) found = ((int)A.large[j] == k);
                        uint32_t tb = tbits;
                        for (uint32_t j = 0; tb != 0u && j < (uint32_t)kEntCap && !found; ++j) {
                            const uint32_t T = (uint32_t)__builtin_ctz(tb);
                            tb &= tb - 1u;
                            for (uint32_t mk = queue32[j * BLK + tid]; mk != 0u && !found; mk &= mk - 1u)
                                found = (s_tile_sphere[slot_of(T, (uint32_t)__builtin_ctz(mk))] == k);
                        }
                        atomicAdd(&A.debug[3], 1ull);
                        if (!found) {
                            if (atomicAdd(&A.debug[0], 1ull) == 0ull) {  // record the first miss for offline analysis
                                float *dbg = reinterpret_cast<float *>(A.debug + 4);
                                dbg[0] = o.x, dbg[1] = o.y, dbg[2] = o.z, dbg[3] = d.x, dbg[4] = d.y, dbg[5] = d.z;
                                dbg[6] = (float)k, dbg[7] = b * b - a * cc, dbg[8] = 0.f;
                                dbg[9] = a, dbg[10] = (float)ncand;
                            }
                        }
                    }
                }
            }
            if (VERIFY && active) {
                atomicAdd(&A.debug[1], (unsigned long long)ncand);
                if (overflow) atomicAdd(&A.debug[2], 1ull);
                // audit of the tile culling (which verify mode itself does not apply): the tile holding the brute-force
                // WINNER must be among the tiles this lane would have asked for; a culled winner counts as a miss
                if (cull_axis < 3u && vidx >= 0) {
                    const uint32_t mine = lane_tile_mask(P, s_cull, o, d, active, best, cull_axis, cull_always);   // `best`: the large spheres only so far
                    bool is_large = false;
                    for (uint32_t j = 0; j < A.n_large; ++j) is_large = is_large || ((int)A.large[j] == vidx);
                    uint32_t slot = 0;
                    while (slot < A.n_tiles * 32u && (int)s_tile_sphere[slot] != vidx) ++slot;
                    if (!is_large && !((mine >> (slot >> 5)) & 1u)) atomicAdd(&A.debug[0], 1ull);
                }
            }
        }
    }
    drain();   // wave-wide (prefix sums, cross-lane fetches): every lane takes part
    if (overflow) best = vbest, idx = vidx, best_rank = vrank;
    PT_SUB(7);
    t_out = best;
    return idx;
}

}  // namespace ptdev
