// pt_texture.h -- Texture::value (texture.rs:74-91) and Perlin noise (perlin.rs:54-111) on the device, incl. the wave-balanced turbulence; part of the sphere / world kernels (pt_kernel.h, pt_world.h).
#pragma once
#include "pt_args.h"
#include "pt_device.h"
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

// `small`: every lane's coordinates are known to lie below 2^31 in magnitude (wave-uniform; perlin_small_range on the base point covers all seven
// octaves). Rust's saturating `f32 as usize` (perlin.rs:96-98) is then just "negative -> 0, else the exact integer": the two comparisons against
// 2^31 and 2^64 of floor_as_usize_low8 fall away -- 3 instead of 8 instructions per axis, 15 of an octave's ~150.
__device__ __forceinline__ bool perlin_small_range(f3 p) {
    return __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(p.x), __builtin_fabsf(p.y)), __builtin_fabsf(p.z)) < 16777216.0f;   // 2^24: x 2^6 for the last octave (a NaN fails)
}
template <bool SMALL = false>
__device__ __forceinline__ float perlin_noise(const PerlinLds &pn, f3 p) {
    const float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    const float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    const uint32_t i = SMALL ? ((uint32_t)__builtin_fmaxf(fx, 0.0f) & 255u) : floor_as_usize_low8(fx);
    const uint32_t j = SMALL ? ((uint32_t)__builtin_fmaxf(fy, 0.0f) & 255u) : floor_as_usize_low8(fy);
    const uint32_t k = SMALL ? ((uint32_t)__builtin_fmaxf(fz, 0.0f) & 255u) : floor_as_usize_low8(fz);
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
template <bool SMALL>
__device__ __forceinline__ float perlin_turb_octaves(const PerlinLds &pn, f3 p) {
    float accum = 0.0f;
    f3 temp_p = p;
    float weight = 1.0f;
    for (int d = 0; d < 7; ++d) {
        accum += weight * perlin_noise<SMALL>(pn, temp_p);
        weight *= 0.5f;
        temp_p = scale3(temp_p, 2.0f);
    }
    return fabsf(accum);
}
// FAST = false: the caller keeps one copy of the octave (the general-world kernel: two copies cost it 1.7 % on simple_light; the sphere kernels gain 2-4 %)
template <bool FAST = true>
__device__ __forceinline__ float perlin_turb(const PerlinLds &pn, f3 p) {
    // (one wave-uniform test for all lanes that are here; the lanes of a divergent caller vote among themselves)
    if (!FAST || __builtin_expect(wave_any(!perlin_small_range(p)), 0)) return perlin_turb_octaves<false>(pn, p);
    return perlin_turb_octaves<true>(pn, p);
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
template <int MAX_ROUNDS = PT_BALANCE_MAX, bool FAST = true>
__device__ __forceinline__ float wave_balanced_turb(const PerlinLds &pn, uint32_t *scratch, bool need, f3 p) {
    const unsigned long long mask = wave_ballot(need);
    const uint32_t n = (uint32_t)__popcll(mask);
    if (n == 0u) return 0.0f;
    if (7u * n > (uint32_t)MAX_ROUNDS * 64u) return need ? perlin_turb<FAST>(pn, p) : 0.0f;   // (measured on config 5: balancing pays up to four rounds -- 8.03 Grays/s against 7.73 without, 7.97 when always on)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    float *sp = reinterpret_cast<float *>(scratch);
    if (need) sp[3u * rank] = p.x, sp[3u * rank + 1u] = p.y, sp[3u * rank + 2u] = p.z;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float accum = 0.0f;
    const uint32_t tasks = 7u * n;
    const bool small = FAST && !wave_any(need && !perlin_small_range(p));   // (wave-uniform: every point of this call, hence every task)
    for (uint32_t base = 0; base < tasks; base += 64u) {
        const uint32_t t = base + lane;
        float val = 0.0f;
        if (t < tasks) {
            const uint32_t k = t / 7u, oct = t - 7u * k;
            const float sc = (float)(1u << oct);          // temp_p after `oct` doublings (perlin.rs:83)
            const f3 q = mk3(sp[3u * k] * sc, sp[3u * k + 1u] * sc, sp[3u * k + 2u] * sc);
            val = small ? perlin_noise<true>(pn, q) : perlin_noise<false>(pn, q);
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
    return texture_leaf_value(t, t.kind == PT_TEX_NOISE ? perlin_turb<false>(pn, p) : 0.0f, p, u, v, images);
}

}  // namespace ptdev
