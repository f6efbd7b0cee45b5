// pt_kernel.h -- persistent-threads path-tracing kernel for gfx950 (MI355X).
//
// One pixel per lane; every lane owns its pixel's xoshiro256+ stream
// (scene.rs:96-102) and walks samples and bounces iteratively. When a path
// terminates the lane regenerates the next camera ray in place; when the
// pixel's samples are exhausted the lane pulls the next pixel from a global
// work counter (one wave-aggregated atomic per refill).
//
// Closest hit, list world (hitable_list.rs:40-56): phase 1 finds a superset of the spheres whose reference
// discriminant is positive -- by default with an f16 MFMA GEMM over lifted ray/sphere features (64 rays x
// 32 spheres x K=32 per tile, "MFMA prefilter" below), alternatively with a wave-uniform exact VALU scan --
// and phase 2 runs the survivors through the reference's exact arithmetic. The MFMA kernels only run the sphere
// tiles some lane's clipped ray segment can reach ("tile culling") and balance phase 2 over the wave (one
// (ray, sphere) pair per lane and round, reduced per ray with a 64-bit LDS atomic min).
// Closest hit, BVH world (bvh.rs:37-62): per-lane resumable traversal of an internal tree, with the reference's
// accept/reject decision reproduced by a slab test on each sphere's parent AABB in the caller's tree.
// DESIGN.md section 4 has the derivations and the error budget.
#pragma once
#include "pt_args.h"
#include "pt_device.h"
#include "pt_tree4.h"
#include "ptgpu.h"


namespace ptdev {

// value of `v` in lane `src_lane` (any lane may ask for any lane's)
__device__ __forceinline__ float lane_fetch_any(uint32_t src_lane, float v) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), __float_as_int(v)));
}

// ---- perlin.rs:54-111 -------------------------------------------------------
struct PerlinLds {
    const float4 *vec;       // 256 x float4
    const uint8_t *perm;     // 768 BYTES: perm_x | perm_y | perm_z (a 256-byte table spans each LDS bank exactly once, so two
                             // lanes on one bank read the same word: the gathers have no bank conflicts)
    bool prefetch;           // fetch the eight gradients of an octave before its arithmetic (32 more live registers: kernels
                             // compiled for 128 VGPRs spill with it, the general-world kernel gains 10 % from it)
};

__device__ __forceinline__ float perlin_noise(const PerlinLds &pn, f3 p) {
    const float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    const float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    const uint32_t i = floor_as_usize_low8(fx), j = floor_as_usize_low8(fy), k = floor_as_usize_low8(fz);
    const float uu = u * u * (3.0f - 2.0f * u);
    const float vv = v * v * (3.0f - 2.0f * v);
    const float ww = w * w * (3.0f - 2.0f * w);
    // perlin.rs:66-69: the trilinear weights (ii*uu + (1-ii)*(1-uu)) with ii in {0, 1} are EXACTLY (1-uu) and uu
    // (0*x = +0 and x + 0 = x for the non-negative finite uu; NaN propagates either way), so they are folded
    // here; products and the accumulation keep the reference's order. The 24 permutation lookups of
    // perlin.rs:101-107 reduce to 6 distinct ones.
    const float wu[2] = {1.0f - uu, uu}, wv[2] = {1.0f - vv, vv}, ww2[2] = {1.0f - ww, ww};
    const uint32_t px[2] = {pn.perm[i], pn.perm[(i + 1) & 255]};
    const uint32_t py[2] = {pn.perm[256 + j], pn.perm[256 + ((j + 1) & 255)]};
    const uint32_t pz[2] = {pn.perm[512 + k], pn.perm[512 + ((k + 1) & 255)]};
    float accum = 0.0f;
    if (!pn.prefetch) {
#pragma unroll
        for (int di = 0; di < 2; ++di) {
#pragma unroll
            for (int dj = 0; dj < 2; ++dj) {
#pragma unroll
                for (int dk = 0; dk < 2; ++dk) {
                    const float4 gc = pn.vec[px[di] ^ py[dj] ^ pz[dk]];
                    const f3 weight = mk3(u - (float)di, v - (float)dj, w - (float)dk);
                    accum += wu[di] * wv[dj] * ww2[dk] * dot3(mk3(gc.x, gc.y, gc.z), weight);
                }
            }
        }
        return accum;
    }
    // all eight gradient fetches are issued before the arithmetic starts (one LDS round trip per octave instead of eight)
    float4 g[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) g[c] = pn.vec[px[c >> 2] ^ py[(c >> 1) & 1] ^ pz[c & 1]];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int di = 0; di < 2; ++di) {
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
#pragma unroll
            for (int dk = 0; dk < 2; ++dk) {
                const float4 gc = g[di * 4 + dj * 2 + dk];
                const f3 weight = mk3(u - (float)di, v - (float)dj, w - (float)dk);
                accum += wu[di] * wv[dj] * ww2[dk] * dot3(mk3(gc.x, gc.y, gc.z), weight);
            }
        }
    }
    return accum;
}

// perlin.rs:76-87
__device__ __forceinline__ float perlin_turb(const PerlinLds &pn, f3 p) {
    float accum = 0.0f;
    f3 temp_p = p;
    float weight = 1.0f;
    for (int d = 0; d < 7; ++d) {
        accum += weight * perlin_noise(pn, temp_p);
        weight *= 0.5f;
        temp_p = scale3(temp_p, 2.0f);
    }
    return fabsf(accum);
}

// perlin.rs:76-87 for the lanes of a wave that need it, BALANCED over the wave: the seven octaves of a point are independent
// evaluations of perlin_noise (at p, 2p, 4p, ... -- doubling is exact), so the wave's 7 n (point, octave) tasks are spread
// over all 64 lanes, ceil(7 n / 64) rounds instead of seven when only n of the 64 lanes hit a noise-textured surface (the
// rest are sky misses or lanes still traversing). Each owner then adds its octaves up in the reference's order,
// accum += weight * noise with weight = 1, 1/2, 1/4 ..., fetching them across lanes: bit-identical to perlin_turb.
// `scratch`: 192 words of this wave's LDS (the pair list, idle between drains). Returns 0 for lanes that do not `need`.
#ifndef PT_BALANCE_MAX
#define PT_BALANCE_MAX 4
#endif
template <int MAX_ROUNDS = PT_BALANCE_MAX>
__device__ __forceinline__ float wave_balanced_turb(const PerlinLds &pn, uint32_t *scratch, bool need, f3 p) {
    const unsigned long long mask = wave_ballot(need);
    const uint32_t n = (uint32_t)__popcll(mask);
    if (n == 0u) return 0.0f;
    if (7u * n > (uint32_t)MAX_ROUNDS * 64u) return need ? perlin_turb(pn, p) : 0.0f;   // (measured on config 5: balancing pays up to four rounds -- 8.03 Grays/s against 7.73 without, 7.97 when always on)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    float *sp = reinterpret_cast<float *>(scratch);
    if (need) sp[3u * rank] = p.x, sp[3u * rank + 1u] = p.y, sp[3u * rank + 2u] = p.z;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float accum = 0.0f;
    const uint32_t tasks = 7u * n;
    for (uint32_t base = 0; base < tasks; base += 64u) {
        const uint32_t t = base + lane;
        float val = 0.0f;
        if (t < tasks) {
            const uint32_t k = t / 7u, oct = t - 7u * k;
            const float sc = (float)(1u << oct);          // temp_p after `oct` doublings (perlin.rs:83)
            val = perlin_noise(pn, mk3(sp[3u * k] * sc, sp[3u * k + 1u] * sc, sp[3u * k + 2u] * sc));
        }
        // octave j of the owner with rank r is task 7 r + j: computed in round (7 r + j) / 64 by lane (7 r + j) % 64
        float weight = 1.0f;
#pragma unroll
        for (uint32_t j = 0; j < 7u; ++j) {
            const uint32_t tj = 7u * rank + j;
            const float v = lane_fetch_any(tj & 63u, val);
            if (need && (tj & ~63u) == base) accum += weight * v;   // perlin.rs:82
            weight *= 0.5f;
        }
    }
    __builtin_amdgcn_wave_barrier();   // (the scratch words are the pair list again from here on)
    return fabsf(accum);
}

// texture.rs:78-85: `sin(s.x) * sin(s.y) * sin(s.z) < 0.0`. Only the SIGN of the product is used, and
// sign(sin x) = sign(x) * (-1)^floor(|x| / pi) for every finite x != 0 (libm's sinf is accurate to
// < 1 ulp and |sin x| of an f32 x is never small enough to round to zero, so its sign is the exact
// sign; the f64 quotient has ~1000x more resolution than the closest an f32 gets to a multiple of pi
// at scene scale). A zero factor makes the product +-0, which is not < 0. Three f64 multiplies
// replace three full-range sinf evaluations; huge or non-finite arguments take the sinf path.
__device__ __forceinline__ bool checker_is_odd(float sx, float sy, float sz) {
    const float ax = __builtin_fabsf(sx), ay = __builtin_fabsf(sy), az = __builtin_fabsf(sz);
    if (!(ax < 1.0e6f && ay < 1.0e6f && az < 1.0e6f)) return sinf(sx) * sinf(sy) * sinf(sz) < 0.0f;
    if (sx == 0.0f || sy == 0.0f || sz == 0.0f) return false;
    constexpr double kInvPi = 0.31830988618379067154;
    const int kx = (int)((double)ax * kInvPi), ky = (int)((double)ay * kInvPi), kz = (int)((double)az * kInvPi);
    const int neg = (kx ^ ky ^ kz) & 1;
    const int sgn = (int)((__float_as_uint(sx) ^ __float_as_uint(sy) ^ __float_as_uint(sz)) >> 31);
    return (neg ^ sgn) != 0;
}

// texture.rs:5-37 RgbImage sources of a general world: (byte offset, width, height) per image + one byte blob
struct DImages {
    const uint4 *table;
    const uint8_t *bytes;
};

// texture.rs:27-37 (Rust `as i32` saturates and maps NaN to 0, like v_cvt_i32_f32)
__device__ __forceinline__ f3 image_value(const DImages &im, int32_t index, float u, float v) {
    const uint4 e = im.table[index];
    const float fi = u * (float)e.y, fj = (1.0f - v) * (float)e.z - 0.001f;
    int32_t i = (fi == fi) ? (int32_t)fminf(fmaxf(fi, -2147483648.0f), 2147483520.0f) : 0;
    int32_t j = (fj == fj) ? (int32_t)fminf(fmaxf(fj, -2147483648.0f), 2147483520.0f) : 0;
    i = max(i, 0), i = min(i, (int32_t)e.y - 1);
    j = max(j, 0), j = min(j, (int32_t)e.z - 1);
    const uint8_t *px = im.bytes + e.x + 3u * (uint32_t)i + 3u * e.y * (uint32_t)j;
    return mk3((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
}

// texture.rs:74-91 in two steps: which leaf texture colours the point (Checker may nest, texture.rs:78-85) ...
__device__ __forceinline__ DTex texture_leaf(const DTex *texs, int32_t tex, f3 p) {
    DTex t = texs[tex];
    while (t.kind == PT_TEX_CHECKER) {
        const f3 s = mk3(10.0f * p.x, 10.0f * p.y, 10.0f * p.z);
        t = texs[checker_is_odd(s.x, s.y, s.z) ? t.odd : t.even];
    }
    return t;
}
// ... and its value (Constant / Noise / Image); `turb` = perlin.rs:76-87 at p, wherever it was evaluated. (u, v, images) only matter for Image.
__device__ __forceinline__ f3 texture_leaf_value(const DTex &t, float turb, f3 p, float u, float v, DImages images) {
    if (t.kind == PT_TEX_NOISE) {
        const float v1 = 1.0f + sin_colour(t.scale * p.z + 10.0f * turb);
        return mk3(0.5f * v1, 0.5f * v1, 0.5f * v1);  // vec3(1,1,1) * 0.5 * (1 + sin(..))
    }
    if (t.kind == PT_TEX_IMAGE) return image_value(images, t.odd, u, v);
    return mk3(t.c0, t.c1, t.c2);
}
__device__ __forceinline__ f3 texture_value(const DTex *texs, const PerlinLds &pn, int32_t tex, f3 p, float u = 0.0f, float v = 0.0f,
                                         DImages images = DImages{nullptr, nullptr}) {
    const DTex t = texture_leaf(texs, tex, p);
    return texture_leaf_value(t, t.kind == PT_TEX_NOISE ? perlin_turb(pn, p) : 0.0f, p, u, v, images);
}

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
                        w_pairs[pos++] = ((uint32_t)lane << 16) | (uint32_t)s_tile_sphere[slot_of(T, b)];
                    }
                }
            }
            const unsigned long long key0 = idx < 0 ? ~0ull : key_of(best, idx, best_rank);
            w_keys[lane] = key0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // 3. one pair per lane and round, with the owner's ray fetched across lanes
            for (uint32_t base = 0; base < total; base += 64u) {
                const uint32_t e = base + (uint32_t)lane < total ? w_pairs[base + lane] : 0xffffu;
                const uint32_t owner = e >> 16;
                const int k = (int)(e & 0xffffu);
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
// relative / absolute slack of the distance cull (DESIGN.md "BVH culling slack")
constexpr float kCullRel = 1.02f;
constexpr float kCullAbs = 0.02f;

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
__device__ __forceinline__ void pair_test4(const KArgs &A, uint32_t e, float time, f3 o, f3 d, f3 rcp, const DivA &av, unsigned long long *key) {
    const float a = av.a;
    const bool gated = A.gate != nullptr;
    const float4 *R = A.slotrec + 4 * (size_t)e;
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
__device__ __forceinline__ void pair_test4_owner(const KArgs &A, uint32_t slot, float time, f3 o, f3 d, const DivA &av, unsigned long long *key) {
    // (the gate of a BVH world needs ray.rs:14's 1 / d of the OWNER's ray; the owner's lane may be walking somebody else's subtree with
    //  another ray's reciprocal in its registers, so it is formed here, from the fetched direction, in its short exact form)
    const f3 rcp = A.gate ? mk3(recip_exact(d.x), recip_exact(d.y), recip_exact(d.z)) : mk3(0.f, 0.f, 0.f);
    pair_test4<MOVING>(A, slot, time, o, d, rcp, av, key);
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
            // exact tests of the queued leaf candidates, one (owner ray, leaf slot) pair per lane and round, reduced into the owner's key
            // with ds_min_u64 (the entries carry their owner); afterwards every lane refreshes its culling limit from ITS owner's key
            const uint32_t incl = wave_inclusive_sum(qn);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (total != 0u) {
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
                        if (valid) pair_test4_owner<MOVING>(A, e & ((1u << kPairLaneShift) - 1u), ptime, po, pd, pav, &w_keys[ow]);
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
                        if (valid) pair_test4_owner<MOVING>(A, e & ((1u << kPairLaneShift) - 1u), ptime, po, pd, pav, &w_keys[ow]);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                qn = 0;
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
#include "pt_coop.h"   // the wave-cooperative mode of the wide list kernels (one pixel per wave), built from the pieces above
namespace ptdev {

// SPH_LDS: list-mode sphere scan reads the (cx,cy,cz,r^2) table from LDS
// (staged once per workgroup); otherwise from HBM/L2 through wave-uniform loads.
// PILOT: the measuring launch that precedes the frame kernel of a new view (own symbol so profiles keep the two apart; A.phase == 1):
// the FIRST sample of every pixel, for real -- it parks each pixel's RNG stream and colour sum for the frame kernel (A.phase == 2)
// and counts the rays per tile.
// MOVING: the world also holds MovingSphere entries (moving_sphere.rs): rays keep their time (camera.rs:59) and
// every exact sphere test / normal uses the centre at that time; prefilter fragments and internal-tree boxes
// were built over the motion's whole sweep.
// GATE: a BVH world on the MFMA list kernel (ancestor-AABB gate + DFS-rank ties at hit acceptance).
// BLK: threads per workgroup. 256 (three workgroups per CU) everywhere except the MFMA list kernels, which run ONE
// 768-thread workgroup per CU when the scene allows: the sphere fragments are then staged once per CU instead of three
// times, and the LDS that frees holds the per-lane attenuation stacks (no HBM traffic for them).
//
// MAP of the kernel body (search for the quoted banner):
//   prologue      LDS carve, staging of tables / fragments / frame parameters, per-lane state
//   main loop, one trip = one ray per live lane ("for (;;)"), its ONLY exit at the very end:
//     "---- refill"                 finished pixels are written, free lanes claim pixels (batched), parked streams are reloaded
//     "TAIL: once the list is dry"  a look into one other wave's mailbox (pt_coop.h), read at the end of the trip
//     "---- camera.rs:56-68"        next sample's camera ray + the sphere draws a Metal scatter still owes, one shared rejection loop
//     "---- hitable.rs:39-65"       closest hit: 4-wide tree (bvh4_trace) | binary tree | MFMA prefilter + balanced exact tests | exact scan
//     "---- scene.rs:49-71"         one level of ray_trace: material, scatter, attenuation push; on termination fold + sample count
//     "Hand-over (pt_coop.h)"       a pixel between two samples goes to an idle worker
//   epilogue      the wave retires or becomes a worker (coop_worker), ray count reduction, development counters
// The stages share ~40 loop-carried registers per lane and are kept in one function body on purpose: every attempt to carry that state
// through a struct or across call boundaries has cost registers in a kernel that lives at its 128-VGPR limit (NOTES.md).
#ifdef PT_BBPROF   // tools/bbprof.py: the instrumented assembly keeps its counter registers above the compiler's
#include "pt_bbprof.h"
#define PT_BBPROF_ATTR __attribute__((amdgpu_num_sgpr(100)))
#else
#define PT_BBPROF_ATTR
#endif
template <bool BVH, bool SPH_LDS, bool MFMA, bool VERIFY, bool PILOT, bool MOVING = false, bool GATE = false, int BLK = kBlock>
__global__ __launch_bounds__(BLK, (BLK == kBlock) ? ((BVH && SPH_LDS) ? PT_TREE4_WAVES : PT_MINWAVES) : 1) PT_BBPROF_ATTR void pt_trace_kernel(const KArgs A) {
    static_assert(BLK == kBlock || (MFMA && !BVH), "only the MFMA list kernels take another workgroup size");
    constexpr bool TREE4 = BVH && SPH_LDS;   // tree kernels: SPH_LDS selects the 4-wide tree (false: the binary one, variant bit 2048)
    // Wide (one workgroup per CU) MFMA kernels: every attenuation a path can pick up is one of a finite PALETTE -- a sphere's
    // constant / metal albedo, one of its two checker colours, or white (dielectric) -- so the per-lane attenuation stack
    // holds 16-bit codes (sphere index | even-checker bit; 0x7fff = white) instead of three floats, and the fold reads the
    // colours back from the per-sphere shading records, which live in LDS here. 18 instead of 108 bytes of LDS per lane.
    // (launch() only picks a wide kernel for scenes whose textures are all Constant or Checker-of-two-Constants.)
    constexpr bool PAL = (BLK != kBlock);
    // The wide frame kernels hand pixels over to waves that have run out of work (pt_coop.h); A.tail_cap == 0 switches it off.
    constexpr bool TAIL = PAL && !PILOT && !VERIFY;
    // 4-wide tree kernels: ONE 32-bit word per attenuation-stack level -- the grey value of a Noise texture as its float bits
    // (texture.rs:86-89 yields (v, v, v)), or a palette code like PAL's for everything else: 0xFFE00000 | even-checker bit << 20 |
    // record index (0xFFFFF = white). No arithmetic produces such a NaN pattern (canonical NaNs are 0x7FC00000 / 0xFFC00000).
    // A quarter of the LDS (and of the HBM traffic of the levels that do not fit) of three floats, and one register instead
    // of three for the first bounce. (launch() only walks this tree for scenes whose textures are Noise, Constant or
    // Checker-of-two-Constants; others use the binary tree kernel with the float stack.)
    constexpr bool WST = TREE4;
    constexpr uint32_t kWstCode = 0xFFE00000u, kWstWhite = kWstCode | 0xFFFFFu;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS carve (all offsets multiples of 16)
    float4 *s_sph = reinterpret_cast<float4 *>(smem);  // list mode: n_spheres x (cx,cy,cz,r^2)
    unsigned char *p = smem + A.lds_sphere_bytes;
    float4 *s_pvec = reinterpret_cast<float4 *>(p);     // perlin gradients (4 KB) when has_noise
    uint8_t *s_perm = p + (A.has_noise ? 4096 : 0);
    p += A.has_noise ? (4096 + 768) : 0;
    uint32_t *s_bvh = reinterpret_cast<uint32_t *>(p);
    p += BVH ? (A.bvh_stack_entries * BLK * (TREE4 ? 2 : 4)) : 0;   // (4-wide tree: 16-bit entries, an even number of them)
    DWideNode *s_nodes = reinterpret_cast<DWideNode *>(p);
    p += (BVH && A.nodes_in_lds) ? A.n_nodes * 64u : 0u;
    uint16_t *s_queue = reinterpret_cast<uint16_t *>(p);  // exact scan: [kQueueCap+1][BLK] u16; MFMA: [kEntCap][BLK] u32 tile masks + per-wave pair lists
    //                                                   4-wide tree: [kLeafQ][BLK] u32 leaf candidates + per-wave pair lists
    uint32_t *w_pairs = reinterpret_cast<uint32_t *>(p + (TREE4 ? kLeafQ : kEntCap) * BLK * 4 + (threadIdx.x >> 6) * kWavePairBytes);
    unsigned long long *w_keys = reinterpret_cast<unsigned long long *>(w_pairs + kPairCap);
    p += BVH ? (TREE4 ? tree4_queue_bytes(BLK) : 0u) : (MFMA ? mfma_queue_bytes(BLK) : scan_queue_bytes(BLK));
    uint4 *s_afrag = reinterpret_cast<uint4 *>(p);        // MFMA: [n_tiles][2][64] x 16 B
    p += MFMA ? A.n_tiles * 2048u : 0u;
    uint16_t *s_tile_sphere = reinterpret_cast<uint16_t *>(p);
    p += MFMA ? ((A.n_tiles * 64u + 15u) & ~15u) : 0u;
    uint32_t *s_cull = reinterpret_cast<uint32_t *>(p);   // MFMA: tile-culling tables, 2 axes x 2 x kCullCells words
    p += MFMA ? 16u * kCullCells : 0u;

    const float4 *s_par = reinterpret_cast<const float4 *>(p);   // frame parameters (kLdsParamBytes)
    p += kLdsParamBytes;
    float4 *s_shade = reinterpret_cast<float4 *>(p);    // PAL: the per-sphere shading records (64 B each)
    p += PAL ? (A.n_spheres + 1u) * 64u : 0u;
    float4 *s_gate = reinterpret_cast<float4 *>(p);     // PAL && GATE: gate boxes (2 float4 per sphere) and DFS ranks of a BVH world
    p += (PAL && GATE) ? A.n_spheres * 32u : 0u;
    uint32_t *s_rank = reinterpret_cast<uint32_t *>(p);
    p += (PAL && GATE) ? ((A.n_spheres * 4u + 15u) & ~15u) : 0u;
    float4 *s_motion = reinterpret_cast<float4 *>(p);   // PAL && MOVING: the MovingSphere records (2 float4 per sphere)
    p += (PAL && MOVING) ? A.n_spheres * 32u : 0u;
    const float4 *const mot = (PAL && MOVING) ? (const float4 *)s_motion : A.motion;
    float *s_path = reinterpret_cast<float *>(p);       // [max_depth][3][BLK] attenuation stack (PAL: u16 [max_depth][BLK] palette codes)
    uint16_t *s_pal = reinterpret_cast<uint16_t *>(p);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t wave_id = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: lives in a scalar register

    if (!BVH && SPH_LDS) {
        for (uint32_t k = tid; k < A.n_spheres_pad; k += BLK) s_sph[k] = A.spheres_r2[k];
    }
    if (MFMA) {
        for (uint32_t k = tid; k < A.n_tiles * 128u; k += BLK) s_afrag[k] = A.afrag[k];
        for (uint32_t k = tid; k < A.n_tiles * 32u; k += BLK) s_tile_sphere[k] = A.tile_sphere[k];
        if (A.cull_axis < 3u)
            for (uint32_t k = tid; k < 4u * kCullCells; k += BLK) s_cull[k] = A.cull_tab[k];
    }
    if (PAL) {
        for (uint32_t k = tid; k < A.n_spheres * 4u; k += BLK) s_shade[k] = A.shade[k];
        if (tid < 4) s_shade[A.n_spheres * 4u + tid] = make_float4(1.f, 1.f, 1.f, 0.f);   // the white entry (Dielectric, material.rs:117)
        if (GATE) {
            for (uint32_t k = tid; k < A.n_spheres * 2u; k += BLK) s_gate[k] = A.gate[k];
            for (uint32_t k = tid; k < A.n_spheres; k += BLK) s_rank[k] = A.leaf_rank[k];
        }
        if (MOVING)
            for (uint32_t k = tid; k < A.n_spheres * 2u; k += BLK) s_motion[k] = A.motion[k];
    }
    if (BVH && A.nodes_in_lds) {
        const uint4 *src = reinterpret_cast<const uint4 *>(A.wnodes);
        uint4 *dst = reinterpret_cast<uint4 *>(s_nodes);
        for (uint32_t k = tid; k < A.n_nodes * 4u; k += BLK) dst[k] = src[k];
    }
    if (tid == 0) {
        float4 *w = const_cast<float4 *>(s_par);
        w[0] = make_float4(A.clip_min[0], A.clip_min[1], A.clip_min[2], A.cull_u0);
        w[1] = make_float4(A.clip_max[0], A.clip_max[1], A.clip_max[2], A.cull_inv_cell);
        w[2] = make_float4(A.c0[0], A.c0[1], A.c0[2], A.rs2);
        w[3] = make_float4(A.m0, A.gamma, A.inv_nx, A.inv_ny);
        w[4] = make_float4(A.cam.origin.x, A.cam.origin.y, A.cam.origin.z, A.cam.lower_left_corner.x);
        w[5] = make_float4(A.cam.lower_left_corner.y, A.cam.lower_left_corner.z, A.cam.horizontal.x, A.cam.horizontal.y);
        w[6] = make_float4(A.cam.horizontal.z, A.cam.vertical.x, A.cam.vertical.y, A.cam.vertical.z);
        w[7] = make_float4(A.cam.u.x, A.cam.u.y, A.cam.u.z, A.cam.v.x);
        w[8] = make_float4(A.cam.v.y, A.cam.v.z, A.cam.w.x, A.cam.w.y);
        w[9] = make_float4(A.cam.w.z, A.cam.time0, A.cam.time1, A.cam.lens_radius);
        w[10] = make_float4(A.inv_ns, A.mix_prev, A.mix_new, A.prev_zero ? 1.0f : 0.0f);
        w[11] = make_float4(A.sky.x, A.sky.y, A.sky.z, A.has_sky ? 1.0f : 0.0f);
        w[12] = make_float4(__uint_as_float(A.cull_axis), __uint_as_float(A.cull_always), __uint_as_float(A.max_depth), __uint_as_float(A.samples));
        w[13] = make_float4(A.cull_reach[0], A.cull_reach[1], A.cull_reach[2], 0.0f);
        w[14] = make_float4(A.cull_u0_2, A.cull_inv_cell_2, __uint_as_float(A.cull_axis2), 0.0f);
    }
    if (A.has_noise) {
        for (int k = tid; k < 256; k += BLK) s_pvec[k] = A.perlin_vec[k];
        for (int k = tid; k < 768; k += BLK) s_perm[k] = (uint8_t)A.perlin_perm[k];
    }
    __syncthreads();
    if (TAIL && A.tail_cap != 0u && tid == 0 && PT_COOP_DBG(4u)) atomicAdd(&A.work_counter[kCtlStarted], 1u);   // (counted per workgroup: pt_coop.h "end")

    PerlinLds pn{s_pvec, s_perm, false};
    // attenuation stack of this lane: the 768-thread kernels always have it in LDS; the 256-thread ones keep the first
    // A.stack_in_lds slots (3 per level) in LDS and deeper, rarely reached levels in HBM/L2 (what fits next to four
    // resident workgroups of a tree kernel)
    float *gpath = A.gstack + (size_t)blockIdx.x * A.max_depth * 3 * BLK + tid;
    auto path_ld = [&](uint32_t slot) -> float {
        if (BLK != kBlock || slot < A.stack_in_lds) return s_path[slot * BLK + tid];
        return gpath[slot * BLK];
    };
    auto path_st = [&](uint32_t slot, float v) {
        if (BLK != kBlock || slot < A.stack_in_lds)
            s_path[slot * BLK + tid] = v;
        else
            gpath[slot * BLK] = v;
    };

#ifdef PT_SECTIONS
    // development aid (-DPT_SECTIONS): per-wave cycle counts of the main loop's sections (s_memtime deltas at
    // wave-uniform points), summed into debug[16..23]; ptgpu.hip prints the shares after each launch
    unsigned long long sec_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sec_last = __builtin_readcyclecounter();
#define PT_SEC(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); sec_t[i] += now_ - sec_last; sec_last = now_; } while (0)
#else
#define PT_SEC(i) do { } while (0)
#endif
    bool have = false, exhausted = false, need_cam = true, trav_new = false, finished = false;
    bool pend_metal = false;    // the lane's Metal scatter waits for its sphere sample (drawn with the next camera rays' lens samples)
    float metal_fuzz = 0.0f;
    uint32_t pix_rays = 0;
    BvhTrav trav{0u, 0u, 0, kMaxT, -1, 0u, false};
    Steal4 steal4{0u, 0u};
    // per-lane bookkeeping, packed (every register counts: the 4-waves-per-SIMD kernels are compiled for 128 VGPRs):
    //   pxy = pixel column | local row << 16 (launch() keeps width and height below 65536)
    //   sd  = bounce depth (12 bits) | sample number << 12 (launch(): max_depth <= 4095, samples < 2^20)
    uint32_t pxy = 0, sd = 0;
    unsigned long long wave_rays = 0;   // scene.rs:57 ray_count of this WAVE (wave-uniform: lives in scalar registers)
#define PT_DEPTH (sd & 0xfffu)
    Rng rng{0, 0, 0, 0};
    f3 col = mk3(0.f, 0.f, 0.f), o = mk3(0.f, 0.f, 0.f), d = mk3(0.f, 0.f, 0.f);
    f3 att0 = mk3(1.f, 1.f, 1.f);   // attenuation of the first bounce (deeper ones: the stack)
    uint32_t att0c = 0u;            // PAL: its palette code; WST: its word
    const float4 *shade = PAL ? (const float4 *)s_shade : ((TREE4 && !MOVING && A.gate) ? A.shade_rank : A.shade);
    const uint32_t kWhite = A.n_spheres;   // PAL: code of (1, 1, 1): one extra record behind the spheres'
    auto word_colour = [&](uint32_t w) -> f3 {   // WST: the attenuation behind a stack word
        if ((w & kWstCode) != kWstCode) {
            const float v = __uint_as_float(w);
            return mk3(v, v, v);
        }
        if (w == kWstWhite) return mk3(1.f, 1.f, 1.f);
        const float4 q = shade[4u * (w & 0xFFFFFu) + 2u + ((w >> 20) & 1u)];
        return mk3(q.x, q.y, q.z);
    };
    auto palette_colour = [&](uint32_t code) -> f3 {   // PAL: the colour behind a stack entry
        const float4 q = s_shade[4u * (code & 0x7fffu) + 2u + (code >> 15)];
        return mk3(q.x, q.y, q.z);
    };
    float rtime = 0.f;  // ray.time (only MOVING kernels read it)
    bool first_claim = true;   // (wave-uniform: every lane of a wave takes part in its first fetch)
    bool tail_dry = TAIL && A.tail_dry0 != 0u;     // TAIL, wave-uniform: the work list has run dry (from then on pixels may be handed over)
    uint32_t tail_it = 0, tail_streak = 0, tail_rand = (blockIdx.x * (BLK / 64) + wave_id) * 2654435761u + 12345u;   // (wave-uniform)
#ifdef PT_DEVKNOBS
    if (A.wave_end && lane == 0) atomicMin(&A.wave_end[65535], (unsigned long long)wall_clock64());   // (the launch's first wave: origin of the hand-over log's timeline)
#endif
#if defined(PT_SECTIONS) || defined(PT_WAVEDBG)
#define PT_WAVE_DETAIL 1   // development builds: per-wave iteration counts, first / last pixel, moment the work list ran dry
    uint32_t dbg_first_pxy = 0xffffffffu;
#ifdef PT_WAVEDBG
    unsigned long long dbgc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    unsigned long long dbg_exh_iter = 0, dbg_iters = 0, dbg_last_refill = 0, dbg_start = A.wave_end ? wall_clock64() : 0ull;   // (PTGPU_TIMING)
#endif

    for (;;) {
        // ---- refill: one wave-aggregated atomic for all lanes that need a pixel. The refill code (a global atomic
        // round trip and four SplitMix64 steps of 64-bit multiplies) runs for the whole wave whenever ANY lane needs
        // it, so lanes wait until A.refill_min of them do (or nobody has work left): fewer, fuller refills.
        const unsigned long long want = wave_ballot(!have && !exhausted);
        const bool refill_now = __popcll(want) >= (int)A.refill_min || wave_ballot(have) == 0ull;
        if (!have && !exhausted && refill_now) {
            if (finished) {
                // scene.rs:113-116
                finished = false;
                const float4 pf = s_par[10];   // inv_ns, mix_prev, mix_new
                if (PILOT && A.phase == 1u) {   // to be continued: park the stream and the sum
                    uint4 *st = A.px_state + 3u * (size_t)((pxy >> 16) * A.width + (pxy & 0xffffu));
                    st[0] = make_uint4((uint32_t)rng.s0, (uint32_t)(rng.s0 >> 32), (uint32_t)rng.s1, (uint32_t)(rng.s1 >> 32));
                    st[1] = make_uint4((uint32_t)rng.s2, (uint32_t)(rng.s2 >> 32), (uint32_t)rng.s3, (uint32_t)(rng.s3 >> 32));
                    st[2] = make_uint4(__float_as_uint(col.x), __float_as_uint(col.y), __float_as_uint(col.z), 0u);
                }
                col = scale3(col, pf.x);
                if (!PILOT) {
                    float *out = A.rgb + ((pxy >> 16) * A.width + (pxy & 0xffffu)) * 3u;
                    // (prev_zero: pt_render found the host buffer all +0.0f and did not upload it -- same products, same sums)
                    const bool pz = pf.w != 0.0f;   // (KArgs::prev_zero, through the LDS parameter block like its neighbours)
                    const float p0 = pz ? 0.0f : out[0], p1 = pz ? 0.0f : out[1], p2 = pz ? 0.0f : out[2];
                    out[0] = p0 * pf.y + col.x * pf.z;
                    out[1] = p1 * pf.y + col.y * pf.z;
                    out[2] = p2 * pf.y + col.z * pf.z;
                }
                // (frame kernels: the NEXT frame's work order; the pixel's work tile is recomputed from its coordinates)
                if (PILOT || A.tile_cost) atomicAdd(&A.tile_cost[((pxy >> 16) >> kTileLog2) * A.tiles_x + ((pxy & 0xffffu) >> kTileLog2)], pix_rays);
            }
            const unsigned long long m = wave_ballot(1);
            const int leader = __ffsll((long long)m) - 1;
            uint32_t base = 0;
            if (A.first_static != 0u && first_claim) {
                // A SIMD's arbiter serves its OLDEST wave first: the first waves of a 16-wave workgroup advance up to twice as fast
                // as the last ones ("The end of a frame"). The head of the heavy-first list -- the pixels whose
                // serial sample chains decide when the frame ends -- therefore goes to them: a wave's first 64 items are fixed by
                // its age class (wave >> 2) instead of by the race for the counter, which starts behind these items.
                const uint32_t wv = wave_id, cls = wv >> 2, idx = blockIdx.x * 4u + (wv & 3u);
                base = (cls * gridDim.x * 4u + idx) * 64u;
            } else {
                if (lane == leader) base = atomicAdd(A.work_counter, (uint32_t)__popcll(m));
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader) + A.first_static;
            }
            first_claim = false;
            const uint32_t item = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));   // lanes of m below this one
            if (item >= A.n_items) {
                exhausted = true;
#ifdef PT_WAVE_DETAIL
                if (A.wave_end && dbg_exh_iter == 0) dbg_exh_iter = dbg_iters, dbg_last_refill = wall_clock64();
#endif
            } else {
                const uint32_t in = item & (kTilePix - 1u);
                const uint32_t tile = A.tile_order ? A.tile_order[item >> (2u * kTileLog2)] : (item >> (2u * kTileLog2));
                pix_rays = 0;
                // tile / tiles_x by the host's magic (a u32 division costs ~25 instructions and a hoisted reciprocal register):
                // umulhi underestimates the quotient by at most one for any tile < 2^32
                uint32_t trow = __umulhi(tile, A.tiles_x_magic), tcol = tile - trow * A.tiles_x;
                if (tcol >= A.tiles_x) trow += 1u, tcol -= A.tiles_x;
                const uint32_t x = tcol * kTileSide + (in & (kTileSide - 1u));
                const uint32_t ly = trow * kTileSide + (in >> kTileLog2);
                if (x < A.width && ly < A.local_rows) {
                    have = true;
                    pxy = x | (ly << 16);
#ifdef PT_WAVE_DETAIL
                    if (dbg_first_pxy == 0xffffffffu) dbg_first_pxy = pxy;
#endif
                    const uint32_t px = x, py = ly * A.shard_count + A.shard_index;
                    need_cam = true;
                    // phase 2 of a frame whose measuring launch traced every OTHER tile (KArgs::checker): a pixel of an unmeasured tile
                    // starts here, one sample behind the others -- its sample number starts at -1 (20 bits), so that it too is done
                    // when the number reaches s_par[12].w = samples - 1
                    const bool parked = !PILOT && A.phase == 2u && (A.checker == 0u || ((tcol + trow) & 1u) == 0u);
                    sd = (!PILOT && A.phase == 2u && !parked) ? 0xfffff000u : 0u;
                    if (parked) {   // continue the stream and the sum phase 1 parked
                        const uint4 *st = A.px_state + 3u * (size_t)(ly * A.width + x);
                        const uint4 a = st[0], b = st[1], c = st[2];
                        rng.s0 = (uint64_t)a.x | ((uint64_t)a.y << 32), rng.s1 = (uint64_t)a.z | ((uint64_t)a.w << 32);
                        rng.s2 = (uint64_t)b.x | ((uint64_t)b.y << 32), rng.s3 = (uint64_t)b.z | ((uint64_t)b.w << 32);
                        col = mk3(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z));
                    } else {
                        col = mk3(0.f, 0.f, 0.f);
                        // scene.rs:96-102
                        uint64_t seed = ((uint64_t)px * 1973ull + (uint64_t)py * 9277ull + (uint64_t)A.frame_num * 26699ull) | 1ull;
                        if (A.random_seed) {
                            uint64_t h = A.seed_base ^ (seed * 0x9e3779b97f4a7c15ULL);
                            seed = splitmix64_next(h);
                        }
                        rng_seed_from_u64(rng, seed);
                    }
                }
            }
        }
        // TAIL: once the list is dry, every few iterations a look into ONE other wave's mailbox (pt_coop.h): a load of a line nobody
        // else polls, issued here and read at the end of the iteration, so its latency hides behind the iteration's work
        uint64_t tail_probe = 0;
        uint32_t tail_target = 0;
        bool tail_polled = false;   // (wave-uniform)
        if (TAIL && A.tail_cap != 0u) {
            tail_dry = tail_dry || wave_any(exhausted);
            tail_polled = tail_dry && PT_COOP_DBG(2u) && ((tail_it & A.tail_period_mask) == 0u || (uint32_t)__popcll(wave_ballot(have)) <= A.tail_live_max);
            if (tail_polled) {
                tail_rand = tail_rand * 1664525u + 1013904223u;
                tail_target = __umulhi(tail_rand, A.tail_cap);
                tail_probe = wt_load(A.tail_box + 16u * (size_t)tail_target + 8);
            }
            tail_it += 1u;
        }
#ifdef PT_WAVE_DETAIL
        if (A.wave_end) dbg_iters += 1;
#endif
        PT_SEC(0);

        // ---- camera.rs:56-68 + scene.rs:107-108: start the next sample -- and, in the same rejection loop, the
        // random_in_unit_sphere a Metal scatter of the LAST iteration still owes (material.rs:77).
        // Both are "draw until the point lies inside the unit ball" loops (math.rs:6-13 in the plane, math.rs:15-26 in space), and
        // a wave runs such a loop as often as its unluckiest lane needs: 2.8 trips for the lens, 3.1 for the metal lanes, one
        // after the other. A lane is in at most one of the two roles here -- a path that scattered off metal continues, so it
        // needs no camera ray -- and its draws keep their order (the sphere's three draws were the lane's next ones anyway), so
        // both loops become ONE whose trips cost the maximum instead of the sum. Same arithmetic per role: x = 2a - 1 etc.,
        // (x x + y y) + z z with z = 0 in the plane, which is the reference's (x x + y y) + 0.
        const bool cam_role = have && need_cam, met_role = have && pend_metal;
        if (cam_role || met_role) {
            const float4 c0 = s_par[4], c1 = s_par[5], c2 = s_par[6], c3 = s_par[7], c4 = s_par[8], c5 = s_par[9], pn2 = s_par[3];
            const f3 cam_origin = mk3(c0.x, c0.y, c0.z), cam_llc = mk3(c0.w, c1.x, c1.y), cam_horizontal = mk3(c1.z, c1.w, c2.x),
                     cam_vertical = mk3(c2.y, c2.z, c2.w), cam_u = mk3(c3.x, c3.y, c3.z), cam_v = mk3(c3.w, c4.x, c4.y);
            const float cam_time0 = c5.y, cam_time1 = c5.z, cam_lens_radius = c5.w;
            const uint32_t px = pxy & 0xffffu, py = (pxy >> 16) * A.shard_count + A.shard_index;
            float u = 0.f, v = 0.f;
            if (cam_role) {   // scene.rs:107-108: the jitter draws come before the lens draws
                u = rng_plus(rng, (float)px) * pn2.z;
                v = rng_plus(rng, (float)py) * pn2.w;
            }
            float sx, sy, sz;
            for (;;) {   // (a plain divergent loop: a lane leaves when its point is inside, the wave when its last lane has)
                sx = rng_pm1(rng), sy = rng_pm1(rng);   // math.rs:8 / math.rs:17-21: 2 * draw - 1
                sz = 0.0f;
                if (met_role) sz = rng_pm1(rng);
                if (((sx * sx + sy * sy) + sz * sz) < 1.0f) break;
            }
            f3 vec;
            if (cam_role) {
                const float rdx = cam_lens_radius * sx, rdy = cam_lens_radius * sy;
                const f3 offset = add3(scale3(cam_u, rdx), scale3(cam_v, rdy));
                const float tdraw = rng_f32(rng);  // camera.rs:59 time draw (plain spheres ignore ray.time)
                if (MOVING) rtime = cam_time0 + tdraw * (cam_time1 - cam_time0);
                vec = sub3(sub3(add3(add3(cam_llc, scale3(cam_horizontal, u)), scale3(cam_vertical, v)), cam_origin), offset);
                o = add3(cam_origin, offset);
                sd &= ~0xfffu;   // depth = 0
                need_cam = false;
                trav_new = true;
            } else {
                // material.rs:82: reflected + fuzz * random_in_unit_sphere (the fuzz waited in the lane's idle candidate-queue slot)
                const float fuzz = MFMA ? __uint_as_float(reinterpret_cast<const uint32_t *>(s_queue)[tid]) : metal_fuzz;
                vec = add3(d, scale3(mk3(sx, sy, sz), fuzz));
            }
            d = normalize3(vec);   // camera.rs:66 / material.rs:84, once for both roles
            pend_metal = false;
        }

        PT_SEC(1);
        // ---- hitable.rs:39-65: closest hit (inactive lanes carry a null ray)
        const f3 ro = have ? o : mk3(0.f, 0.f, 0.f);
        const f3 rd = have ? d : mk3(0.f, 0.f, 0.f);
        const float a = dot3(rd, rd);  // sphere.rs:34
        // its reciprocal, once per ray, for the quotients of the exact sphere tests (pt_device.h DivA; the exact-scan and binary-tree kernels divide in full)
        const bool short_div = (MFMA && !BVH) || TREE4;
        const DivA av{a, short_div ? recip_unit_range(a) : 0.0f, short_div && wave_ballot(have && !in_unit_range(a)) == 0ull};
        float t_hit;
        int idx;
        if (TREE4) {
            // (every ray of the wave is finished when this returns: no traversal state is carried into the next trip)
            bvh4_trace<MOVING, VERIFY, BLK>(A, reinterpret_cast<uint16_t *>(s_bvh), reinterpret_cast<uint32_t *>(s_queue), w_pairs, w_keys, ro, rd, av, rtime, have, steal4
#ifdef PT_SECTIONS
                                            , sec_t
#endif
                                            );
            trav_new = false;
            idx = -1, t_hit = kMaxT;
            if (have) {   // (lanes without a ray hold a stale or never-written key)
                const unsigned long long key = w_keys[lane];
                const uint32_t low = (uint32_t)key;
                // BVH world: the key carries the leaf's DFS rank; shading reads the rank-ordered copy of the records, so
                // the sphere index itself (one more dependent load) is only needed for a moving sphere's motion record
                if (key != ~0ull) idx = (int)(A.gate ? (MOVING ? A.rank_sphere[0xffffffffu - low] : (0xffffffffu - low)) : low);
                t_hit = __uint_as_float((uint32_t)(key >> 32));
            }
        } else if (BVH) {
            if (have && trav_new) {
                bvh_start<MOVING>(A, s_bvh, o, d, dot3(d, d), rtime, trav);
                trav_new = false;
            }
            if (A.nodes_in_lds)
                bvh_run<true, MOVING, VERIFY>(A, s_bvh, s_nodes, ro, rd, a, rtime, have, trav);
            else
                bvh_run<false, MOVING, VERIFY>(A, s_bvh, A.wnodes, ro, rd, a, rtime, have, trav);
            idx = trav.idx;
            t_hit = trav.best;
        } else if (MFMA)
            idx = intersect_list_mfma<VERIFY, MOVING, GATE, BLK>(A, (PAL && GATE) ? GateSrc{s_gate, s_rank} : GateSrc{A.gate, A.leaf_rank}, mot, s_par, SPH_LDS ? (const float4 *)s_sph : A.spheres_r2, s_afrag, s_tile_sphere, s_cull,
                                                      s_queue, w_pairs, w_keys, ro, rd, av, have, rtime, t_hit
#ifdef PT_SECTIONS
                                                      , sec_t
#elif defined(PT_WAVEDBG)
                                                      , dbgc
#endif
                                                      );
        else
            idx = intersect_list(SPH_LDS ? (const float4 *)s_sph : A.spheres_r2, (int)A.n_spheres_pad, s_queue + tid, ro, rd,
                                 a, t_hit);

        PT_SEC(2);
        // ---- scene.rs:49-71 one level of ray_trace (BVH mode: only lanes whose traversal has finished)
        const bool shading = have && !(BVH && !TREE4 && trav.active);
        wave_rays += (unsigned long long)__popcll(wave_ballot(shading));   // scene.rs:57 `ray_count += 1` for every lane shaded below
        // 4-wide tree kernels: Texture::Noise of the lanes that will scatter off a noise-textured Lambertian, evaluated for the
        // whole wave at once (wave_balanced_turb): the shading below is divergent, and a third of its lanes (sky misses) idle
        float turb_pre = 0.0f;
        if (WST && A.has_noise) {
            bool need = false;
            f3 np = mk3(0.f, 0.f, 0.f);
            if (shading && idx >= 0) {
                const float4 q1n = shade[4 * idx + 1];
                need = __float_as_uint(q1n.x) == (uint32_t)PT_MAT_LAMBERTIAN && (__float_as_uint(q1n.y) & kShadeNoise) != 0u &&
                       PT_DEPTH < __float_as_uint(s_par[12].z);
                np = add3(o, scale3(d, t_hit));   // ray.rs:24-26, the same point the shading computes
            }
            // (the eight gradients of an octave are fetched before its arithmetic here: outside the divergent shading block the
            //  32 registers that takes are free -- 128 VGPRs, no spill; +0.8 %)
            turb_pre = wave_balanced_turb(PerlinLds{s_pvec, s_perm, true}, w_pairs, need, np);
        }
        if (shading) {
            pix_rays += 1;
            bool terminal = true;
            f3 V;
            if (idx < 0) {
                // scene.rs:40-47
                const float4 psky = s_par[11];
                if (psky.w != 0.0f) {
                    V = mk3(psky.x, psky.y, psky.z);
                } else {
                    const float t = 0.5f * (d.y + 1.0f);
                    const float w1 = 1.0f - t;
                    V = mk3(w1 + (t * 0.5f) * 0.3f, w1 + (t * 0.7f) * 0.3f, w1 + (t * 1.0f) * 0.3f);
                }
            } else {
                const float4 sp = sphere_at_m<MOVING>(mot, idx, shade[4 * idx], rtime), q1 = shade[4 * idx + 1],
                             qa = shade[4 * idx + 2], qb = shade[4 * idx + 3];
                const f3 centre = mk3(sp.x, sp.y, sp.z);
                // ray.rs:24-26. The hit point IS the next ray's origin when the path scatters, and a path that ends here gets a new
                // origin from the camera: the lane's origin is advanced in place (no second copy of the point kept alive)
                o = add3(o, scale3(d, t_hit));
                const f3 point = o;
                const f3 normal = divs3_known(sub3(point, centre), sp.w, qa.w);    // sphere.rs:42 (qa.w: 1 / radius from the host)
                struct { uint32_t kind, flags; int32_t tex; float param; } m = {
                    __float_as_uint(q1.x), __float_as_uint(q1.y), (int32_t)__float_as_uint(q1.z), q1.w};
                // Texture::value for this sphere's texture (texture.rs:74-91), inlined for the resolved cases
                auto surface_colour = [&]() -> f3 {
                    if (m.flags & kShadeConst) return mk3(qa.x, qa.y, qa.z);
                    if (m.flags & kShadeChecker2) {
                        const bool odd = checker_is_odd(10.0f * point.x, 10.0f * point.y, 10.0f * point.z);
                        return odd ? mk3(qa.x, qa.y, qa.z) : mk3(qb.x, qb.y, qb.z);
                    }
                    if (m.flags & kShadeNoise) {   // texture.rs:86-89, resolved here: no dependent fetch of the texture record
                        const float v1 = 1.0f + sin_colour(qa.x * point.z + 10.0f * perlin_turb(pn, point));
                        return mk3(0.5f * v1, 0.5f * v1, 0.5f * v1);
                    }
                    return texture_value(A.texs, pn, m.tex, point);
                };
                f3 emitted = mk3(0.f, 0.f, 0.f);                        // material.rs:161-167
                if (m.kind == PT_MAT_DIFFUSE_LIGHT) emitted = surface_colour();
                bool scattered = false;
                f3 att = mk3(1.f, 1.f, 1.f);
                uint32_t attc = WST ? kWstWhite : kWhite;    // PAL / WST: code of `att` (white unless a branch says otherwise)
                if (PT_DEPTH < __float_as_uint(s_par[12].z)) {   // max_depth
                    // every scatter ends in `.normalize()` of some vector (material.rs:63,84,112,119): the branches
                    // only produce that vector, the normalisation is issued once for the whole wave
                    f3 raw = d;
                    if (m.kind == PT_MAT_LAMBERTIAN) {  // material.rs:52-67
                        const f3 target = add3(add3(point, normal), random_unit_vector(rng));
                        if (PAL) {
                            const bool even = (m.flags & kShadeChecker2) && !checker_is_odd(10.0f * point.x, 10.0f * point.y, 10.0f * point.z);
                            attc = (uint32_t)idx | (even ? 0x8000u : 0u);
                        } else if (WST) {
                            if (m.flags & kShadeNoise) {   // texture.rs:86-89 with the turbulence evaluated above
                                const float v1 = 1.0f + sin_colour(qa.x * point.z + 10.0f * turb_pre);
                                attc = __float_as_uint(0.5f * v1);
                            } else {
                                const bool even = (m.flags & kShadeChecker2) && !checker_is_odd(10.0f * point.x, 10.0f * point.y, 10.0f * point.z);
                                attc = kWstCode | (uint32_t)idx | (even ? (1u << 20) : 0u);
                            }
                        } else {
                            att = surface_colour();
                        }
                        raw = sub3(target, point);
                        scattered = true;
                    } else if (m.kind == PT_MAT_METAL) {  // material.rs:69-89
                        const f3 reflected = reflect3(d, normal);
                        if (dot3(reflected, normal) > 0.0f) {
                            att = mk3(qa.x, qa.y, qa.z);
                            attc = (WST ? kWstCode : 0u) | (uint32_t)idx;
                            raw = reflected, pend_metal = true;   // sampled at the top of the next iteration
                            if (MFMA) reinterpret_cast<uint32_t *>(s_queue)[tid] = __float_as_uint(m.param);
                            else metal_fuzz = m.param;
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
                            ni_over_nt = qb.y;   // 1.0 / ref_idx (f32, from the host: pt_prep.hip)
                        }
                        f3 refracted;
                        bool use_refract = false;
                        if (refract3(d, outward_normal, ni_over_nt, refracted)) {
                            const float reflect_prob = qb.x + (1.0f - qb.x) * pow5_ref(1.0f - cosine);   // math.rs:76-80, r0 from the host
                            if (rng_f32(rng) > reflect_prob) use_refract = true;
                        }
                        raw = use_refract ? refracted : reflect3(d, normal);
                        scattered = true;
                    }
                    // (the direction is replaced in place as well: a path that does not scatter ends, and its lane's next ray is a camera ray)
                    if (scattered) d = pend_metal ? raw : normalize3(raw);
                }
                if (scattered) {
                    // the first scatter's attenuation stays in registers (measured best: 0 levels -1.5 %, 2 levels -2.3 %;
                    // it also removes 40 % of the stack's HBM writes); deeper levels go to the per-lane stack, level d
                    // at slot d - 1
                    if (PAL) {
                        if (PT_DEPTH == 0u) att0c = attc;
                        else (s_pal + tid)[(PT_DEPTH - 1u) * BLK] = (uint16_t)attc;
                    } else if (WST) {
                        if (PT_DEPTH == 0u) att0c = attc;
                        else path_st(PT_DEPTH - 1u, __uint_as_float(attc));
                    } else if (PT_DEPTH == 0u) {
                        att0 = att;
                    } else {
                        path_st((PT_DEPTH - 1u) * 3u + 0u, att.x);
                        path_st((PT_DEPTH - 1u) * 3u + 1u, att.y);
                        path_st((PT_DEPTH - 1u) * 3u + 2u, att.z);
                    }
                    sd += 1u;   // depth += 1
                    terminal = false;
                    trav_new = true;
                } else {
                    V = emitted;
                }
            }
            if (terminal) {
                // scene.rs:62-64 unwound: emitted(=0) + attenuation * deeper, innermost first
                if (PAL) {
                    // three levels per trip: the codes, then the colours, are fetched together (two LDS round trips per
                    // trip instead of two per level); the products keep the innermost-first order
                    // (the lane's column is re-derived here, behind an opaque copy: hoisted out of the main loop this address is one
                    //  register too many for the 128 the kernel may use, and it was the last value the compiler spilled)
                    uint32_t tid_here = (uint32_t)tid;
                    asm volatile("" : "+v"(tid_here));
                    const uint16_t *my_pal = s_pal + tid_here;
                    for (int k = (int)PT_DEPTH - 1; k >= 1; k -= 3) {
                        const uint32_t ca = my_pal[(uint32_t)(k - 1) * BLK];
                        const uint32_t cb = my_pal[(uint32_t)(k >= 2 ? k - 2 : 0) * BLK];
                        const uint32_t cc = my_pal[(uint32_t)(k >= 3 ? k - 3 : 0) * BLK];
                        const f3 qa3 = palette_colour(ca), qb3 = palette_colour(cb), qc3 = palette_colour(cc);
                        V = mk3(0.0f + qa3.x * V.x, 0.0f + qa3.y * V.y, 0.0f + qa3.z * V.z);
                        if (k >= 2) V = mk3(0.0f + qb3.x * V.x, 0.0f + qb3.y * V.y, 0.0f + qb3.z * V.z);
                        if (k >= 3) V = mk3(0.0f + qc3.x * V.x, 0.0f + qc3.y * V.y, 0.0f + qc3.z * V.z);
                    }
                }
                if (WST) {
                    // three levels per trip, as above: the words, then the colours behind them, are fetched together (a level per trip ran
                    // 8 trips per wave-iteration on config 5, four lanes switched on, each trip one dependent LDS round trip)
                    for (int k = (int)PT_DEPTH - 1; k >= 1; k -= 3) {
                        const uint32_t wa = __float_as_uint(path_ld((uint32_t)(k - 1)));
                        const uint32_t wb = __float_as_uint(path_ld((uint32_t)(k >= 2 ? k - 2 : 0)));
                        const uint32_t wc = __float_as_uint(path_ld((uint32_t)(k >= 3 ? k - 3 : 0)));
                        const f3 ca = word_colour(wa), cb = word_colour(wb), cc = word_colour(wc);
                        V = mk3(0.0f + ca.x * V.x, 0.0f + ca.y * V.y, 0.0f + ca.z * V.z);
                        if (k >= 2) V = mk3(0.0f + cb.x * V.x, 0.0f + cb.y * V.y, 0.0f + cb.z * V.z);
                        if (k >= 3) V = mk3(0.0f + cc.x * V.x, 0.0f + cc.y * V.y, 0.0f + cc.z * V.z);
                    }
                }
                for (int k = (PAL || WST) ? 0 : (int)PT_DEPTH - 1; k >= 1; --k) {
                    if (PAL || WST) {
                    } else {
                        V.x = 0.0f + path_ld((uint32_t)(k - 1) * 3u + 0u) * V.x;
                        V.y = 0.0f + path_ld((uint32_t)(k - 1) * 3u + 1u) * V.y;
                        V.z = 0.0f + path_ld((uint32_t)(k - 1) * 3u + 2u) * V.z;
                    }
                }
                if (PT_DEPTH > 0u) {
                    const f3 c0 = PAL ? palette_colour(att0c) : (WST ? word_colour(att0c) : att0);
                    V = mk3(0.0f + c0.x * V.x, 0.0f + c0.y * V.y, 0.0f + c0.z * V.z);
                }
                col = add3(col, V);  // scene.rs:110
                sd += 0x1000u;   // sample += 1
                need_cam = true;
                if ((sd >> 12) == __float_as_uint(s_par[12].w)) {   // samples
                    // the pixel is written when the lane fetches its next one (the refill below is batched over
                    // several lanes, and so is this read-modify-write of the frame buffer)
                    have = false;
                    finished = true;
                }
            }
        }
        PT_SEC(3);
        if (TAIL && tail_polled) {
            // Hand-over (pt_coop.h): the list is dry and the probed wave is an idle worker -- the lane at a sample boundary with the
            // most estimated work left parks its pixel (RNG stream, colour sum, counters: the pixel's whole state between two samples)
            // in that worker's mailbox, to be finished with all 64 lanes on each of its rays. At most one pixel per wave and iteration.
            const bool idle = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tail_probe) == ((A.tail_gen << 2) | kBoxIdle) &&
                              (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(tail_probe >> 32)) == (A.tail_gen >> 30);
            tail_streak = idle ? tail_streak + 1u : 0u;
            const uint32_t live = (uint32_t)__popcll(wave_ballot(have));
            if (idle && (live <= A.tail_live_max || tail_streak >= A.tail_streak)) {
                const bool cand = have && need_cam;   // between two samples
                const uint32_t done_s = sd >> 12;
                const float est = (float)pix_rays * (float)(__float_as_uint(s_par[12].w) - done_s) * __builtin_amdgcn_rcpf((float)done_s);   // rays per sample so far x samples left
                const uint32_t eb = (cand && est >= A.tail_min_est) ? __float_as_uint(est) : 0u;   // (NaN before the first sample: not >=)
                const uint32_t mx = wave_max_u32(eb);
                if (mx != 0u && eb == mx && lane == __builtin_ctzll(wave_ballot(eb == mx))) {
                    if (coop_hand_over(A, tail_target, rng, col, pxy, done_s, pix_rays)) have = false;   // (not `finished`: nothing is written, the lane simply holds no pixel any more)
                }
            }
        }
        // The loop's only exit, at its very end. (A wave whose refill brought no pixel -- beyond the frame's edge, or the list ran dry --
        // used to skip the body with `continue` / leave with `break` from here up there. The compiler's structurizer turns such an edge
        // into a flag tested after the body, which keeps every loop-carried register's start-of-iteration value alive THROUGH the body:
        // a second home for ~28 registers and ~45 copies per wave-iteration. An idle trip through the body is harmless: no lane has a ray.)
        if (wave_ballot(have) == 0ull && wave_ballot(!exhausted) == 0ull) break;
    }

    if (TAIL && A.tail_cap != 0u) {
        // this wave hands nothing over any more (its mailbox stores were drained before their flags): count it, and become a worker --
        // unless it is the last one out of a main loop: then nobody can hand anything over, and every worker is told so
        uint32_t last = 0u;
        if (lane == 0 && PT_COOP_DBG(4u)) {
            // (the waves of a workgroup count in LDS -- the one spare word of the parameter block, zero since it was staged -- and only
            // the last of them touches the global counters: 4 096 waves counting on one word cost a 2 ms frame 0.2 ms)
            uint32_t *wg_done = reinterpret_cast<uint32_t *>(const_cast<float4 *>(s_par) + 13) + 3;
            if (atomicAdd(wg_done, 1u) + 1u == (uint32_t)(BLK / 64))
                last = (atomicAdd(&A.work_counter[kCtlDone], 1u) + 1u == ctl_load(A.work_counter + kCtlStarted)) ? 1u : 0u;
        }
        if (__builtin_amdgcn_readfirstlane((int)last) != 0) coop_broadcast_exit(A, A.tail_cap);
        else if (PT_COOP_DBG(1u))
            coop_worker<MOVING, GATE>(A, GATE ? GateSrc{s_gate, s_rank} : GateSrc{nullptr, nullptr}, mot, s_par, s_sph, s_shade, pn, blockIdx.x * (BLK / 64) + wave_id, wave_rays);
    }
#ifdef PT_SECTIONS
    PT_SEC(4);
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&A.debug[16 + i], sec_t[i]);
    if (A.wave_end && lane == 0)
        for (int q = 0; q < 8; ++q) A.wave_end[(5 + q) * (gridDim.x * (BLK / 64)) + blockIdx.x * (BLK / 64) + wave_id] = sec_t[q];
#endif
    if (A.wave_end && lane == 0) {
        const uint32_t w = blockIdx.x * (BLK / 64) + wave_id, nw = gridDim.x * (BLK / 64);
        A.wave_end[w] = wall_clock64();
        // (per-lane values of lane 0 would miss other lanes' exhaustion: take the wave's earliest)
#ifdef PT_WAVE_DETAIL
        A.wave_end[nw + w] = dbg_iters | (dbg_exh_iter << 32), A.wave_end[2 * nw + w] = dbg_last_refill, A.wave_end[3 * nw + w] = dbg_start;
        A.wave_end[4 * nw + w] = dbg_first_pxy | ((unsigned long long)pxy << 32);
#else
        (void)nw;
#endif
#ifdef PT_WAVEDBG
        for (int q = 0; q < 4; ++q) A.wave_end[(5 + q) * nw + w] = dbgc[q];
#endif
    }
    if (BVH && VERIFY) {   // traversal counters (accumulate over the lane's whole life: never reset per ray)
        atomicAdd(&A.debug[8], (unsigned long long)(TREE4 ? steal4.visits : trav.visits));
        atomicAdd(&A.debug[9], (unsigned long long)(TREE4 ? steal4.leaves : trav.leaves) + (lane == 0 ? wave_rays * A.n_bvh_large : 0ull));
    }
    // scene.rs:118 ray_count: wave reduce, one atomic per wave
    if (lane == 0) atomicAdd(A.ray_count, wave_rays);
#undef PT_DEPTH
}

}  // namespace ptdev
