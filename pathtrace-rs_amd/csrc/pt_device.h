// pt_device.h -- device-side arithmetic of the path-tracing hot path (gfx950).
//
// Every function states the reference file:line whose arithmetic it must
// reproduce. The rule for this file: anything that feeds CONTROL FLOW (hit /
// miss, t, normal, scatter direction, Schlick probability, RNG draws) is written
// with the reference's exact operation order, separate mul and add (the TU is
// compiled with -ffp-contract=off; Rust never fuses), IEEE division and sqrt
// (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt), no fast-math.
// Colour-only values (texture colours, attenuation folding) follow the same
// order as well, so list/BVH scenes without libm calls are bit-exact.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_args.h"

namespace ptdev {

constexpr float kMaxT = 3.40282346638528859812e+38f;  // scene.rs:15 f32::MAX
constexpr float kMinT = 0.001f;                       // scene.rs:16
constexpr float kPi = 3.14159274101257324f;           // f32::consts::PI


// Wave votes straight on the condition's lane mask. (hip's __any / __ballot first materialise the condition as 0 / 1 in a VGPR and
// compare it again: two VALU instructions per vote, ~20 per wave-iteration of the list kernels.)
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 add3(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 sub3(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 mul3(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ f3 scale3(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 divs3(f3 a, float s) { return f3{a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ f3 neg3(f3 a) { return f3{-a.x, -a.y, -a.z}; }
// v / r per component (sphere.rs:42 `(point - center) / radius`), correctly rounded, for a divisor whose f32 reciprocal `y` the
// host already holds (pt_prep.hip: 1.0f / radius, NaN when |radius| is outside [2^-20, 2^20]). With no operand or quotient near
// the ends of the exponent range the scaling steps of an IEEE division never apply and it reduces to what is left of clang's
// lowering: q = n * y refined twice through the exact fused residual n - q r; v_div_fixup_f32 supplies signed zeros, infinities
// and NaNs. A wave in which some lane falls outside that range (a component below 2^-90 -- zero included --, one above 2^100, no
// reciprocal) takes the three full divisions. 23 VALU instructions instead of 34 per hit; checked against `/` on 2^32 seeded pairs
// per divisor class (PT_PROBE_SWEEP_DIV).
__device__ __forceinline__ float div_by_known(float n, float r, float y) {
    float q = n * y;
    q = __builtin_fmaf(__builtin_fmaf(-q, r, n), y, q);
    q = __builtin_fmaf(__builtin_fmaf(-q, r, n), y, q);
    return __builtin_amdgcn_div_fixupf(q, r, n);
}
__device__ __forceinline__ bool div_by_known_ok(f3 v, float y) {
    const float lo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), __builtin_fabsf(v.z));
    const float hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), __builtin_fabsf(v.z));
    return lo >= 0x1p-90f && hi < 0x1p100f && y == y;   // (a NaN component fails the first two)
}
__device__ __forceinline__ f3 divs3_known(f3 v, float r, float y) {
    // (the short form first, the rare full divisions behind it: the common path then has no branch in front of its arithmetic)
    f3 q = f3{div_by_known(v.x, r, y), div_by_known(v.y, r, y), div_by_known(v.z, r, y)};
    if (__builtin_expect(wave_any(!div_by_known_ok(v, y)), 0)) q = f3{v.x / r, v.y / r, v.z / r};
    return q;
}
// glam Vec3::dot: (x*x' + y*y') + z*z'
__device__ __forceinline__ float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// f32::sqrt, correctly rounded like sqrtf -- with the parts of clang's lowering that the callers here never need taken out of
// the common path (16 -> 11 VALU instructions, five to seven square roots per wave-iteration). The lowering scales inputs below
// 2^-96 up, takes v_sqrt_f32 (1 ulp), picks among {s - 1 ulp, s, s + 1 ulp} by the sign of two fused residuals, scales back and
// patches +-0 / +inf. The residual selection alone already returns +-0, +inf and NaN unchanged, and without the scaling it is the
// same arithmetic for every input of magnitude >= 2^-96; a wave in which some lane holds a smaller non-zero value takes the library path.
// Checked against __builtin_sqrtf on ALL 2^32 bit patterns (pt_selftest_probe PT_PROBE_SWEEP_SQRT, tests/test_gpu_parity.py).
__device__ __forceinline__ float sqrt_exact(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    s = (0.0f >= rm) ? sm : s;
    s = (0.0f < rp) ? sp : s;
    // (the library path for the rare wave BEHIND the common arithmetic: v_sqrt_f32 issues without waiting for the vote)
    if (__builtin_expect(wave_any(__builtin_fabsf(x) < 0x1p-96f && x != 0.0f), 0)) s = __builtin_sqrtf(x);
    return s;
}
__device__ __forceinline__ float length3(f3 a) { return sqrt_exact(dot3(a, a)); }
// 1.0 / sqrt(t), both correctly rounded (glam 0.20's scalar `1.0 / length`). The divisor is a square root, so it is never
// denormal, never above 2^64 and its reciprocal is a normal number: the scaling steps of a general IEEE division never apply, and
// v_rcp_f32 refined by two fused Newton steps is already the correctly rounded quotient; v_div_fixup_f32 supplies the results for
// 0, inf and NaN. 7 VALU instructions after the root instead of 11 -- and a function of ONE f32, so it is checked against the
// compiler's `1.0f / sqrtf(t)` on all 2^32 inputs (PT_PROBE_SWEEP_INVLEN).
__device__ __forceinline__ float inv_sqrt_exact(float t) {
    const float s = sqrt_exact(t);
    float y = __builtin_amdgcn_rcpf(s);
    y = __builtin_fmaf(__builtin_fmaf(-s, y, 1.0f), y, y);
    y = __builtin_fmaf(__builtin_fmaf(-s, y, 1.0f), y, y);
    return __builtin_amdgcn_div_fixupf(y, s, 1.0f);
}
// n / a for the divisor of the sphere test, a = d.d of the ray (sphere.rs:34,40,51: `(-b -+ sqrt(disc)) / a`). Every ray direction
// of the sphere kernels is a normalised vector, so a lies within a few ulps of 1; for ANY a in [0.5, 2] the scaling steps of an IEEE
// division never apply to the divisor and its reciprocal is a normal number: y = v_rcp_f32(a) refined by two fused Newton steps is
// the correctly rounded 1 / a (computed ONCE per ray), and q = n * y refined twice through the exact residual n - q a is the
// correctly rounded n / a -- what is left of clang's lowering of `/` when nothing needs scaling -- with v_div_fixup_f32 supplying
// zeros, infinities and NaNs. 6 VALU instructions per division instead of 11, on the longest dependent chain of the exact phase.
// Numerators below 2^-99 (the lowering would scale; here the last bits may differ) give quotients far below t_min = 0.001, rejected
// by sphere.rs:41,52 either way; the numerators -b -+ sqrt(disc) of a finite discriminant stay below 2^66, so nothing overflows.
// A wave in which some ray's a is outside [0.5, 2] (never seen; a NaN direction) takes the full divisions: `fast` is wave-uniform.
// Checked against `/` by pt_selftest_probe PT_PROBE_SWEEP_DIVA: every a in [0.5, 2] for the reciprocal, 2^32 seeded (n, a) pairs
// (a within 64 ulps of 1, or anywhere in [0.5, 2]; n any bit pattern) for the quotient.
struct DivA {
    float a, y;   // d.d and its reciprocal
    bool fast;    // wave-uniform: every ray the wave tests this iteration has a in [0.5, 2]
};
__device__ __forceinline__ float recip_unit_range(float a) {
    float y = __builtin_amdgcn_rcpf(a);
    y = __builtin_fmaf(__builtin_fmaf(-a, y, 1.0f), y, y);
    y = __builtin_fmaf(__builtin_fmaf(-a, y, 1.0f), y, y);
    return y;
}
__device__ __forceinline__ bool in_unit_range(float a) { return a >= 0.5f && a <= 2.0f; }
__device__ __forceinline__ float div_by_unit_range(float n, float a, float y) {
    float q = n * y;
    q = __builtin_fmaf(__builtin_fmaf(-q, a, n), y, q);
    q = __builtin_fmaf(__builtin_fmaf(-q, a, n), y, q);
    return __builtin_amdgcn_div_fixupf(q, a, n);
}
// 1.0f / x, correctly rounded, for the ray's reciprocal direction (ray.rs:14). The short form of recip_unit_range holds for every x
// whose magnitude lies in [2^-60, 2^60]: all of its operations scale by exact powers of two there, so it rounds like the same
// mantissa in [1, 2) does (checked exhaustively); a wave in which some lane's x lies outside that range -- an axis-parallel ray's
// 0, a denormal, an infinity, a NaN -- divides in full. Checked against `1.0f / x` on ALL 2^32 inputs (PT_PROBE_SWEEP_RECIP).
__device__ __forceinline__ float recip_exact(float x) {
    float y = recip_unit_range(x);
    const float ax = __builtin_fabsf(x);
    if (__builtin_expect(wave_any(!(ax >= 0x1p-60f && ax <= 0x1p60f)), 0)) y = 1.0f / x;
    return y;
}
__device__ __forceinline__ float div_a(float n, const DivA &v) { return v.fast ? div_by_unit_range(n, v.a, v.y) : n / v.a; }

// glam 0.20 scalar Vec3::normalize: v * (1.0 / length)
__device__ __forceinline__ f3 normalize3(f3 a) { return scale3(a, inv_sqrt_exact(dot3(a, a))); }

// ---- RNG: rand_xoshiro 0.6 Xoshiro256Plus seeded through SplitMix64 -------
// call sites: scene.rs:96-102 (per-pixel seed), every rng.gen::<f32>()
struct Rng {
    uint64_t s0, s1, s2, s3;
};

__device__ __forceinline__ uint64_t splitmix64_next(uint64_t &x) {
    x += 0x9e3779b97f4a7c15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

__device__ __forceinline__ void rng_seed_from_u64(Rng &r, uint64_t seed) {
    uint64_t x = seed;
    r.s0 = splitmix64_next(x);
    r.s1 = splitmix64_next(x);
    r.s2 = splitmix64_next(x);
    r.s3 = splitmix64_next(x);
}

// The state update is the reference's (rand_xoshiro 0.6: s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t; s3 = rotl(s3, 45)) written
// on 32-bit halves with gfx950's three-input bit operation, so that no intermediate is materialised twice: s1' = s1 ^ s2 ^ s0 and
// s2' = s2 ^ s0 ^ t are one v_bitop3_b32 (truth table 0x96) per half, the rotation two v_alignbit_b32 -- 13 VALU instructions per
// step where the 64-bit form compiles to 15 (ten two-input xors, and a 64-bit shift + shift + or for the rotation).
__device__ __forceinline__ uint64_t rng_next_u64(Rng &r) {
    const uint64_t result = r.s0 + r.s3;
    const uint64_t t = r.s1 << 17;
    const uint32_t s0l = (uint32_t)r.s0, s0h = (uint32_t)(r.s0 >> 32), s1l = (uint32_t)r.s1, s1h = (uint32_t)(r.s1 >> 32);
    const uint32_t s2l = (uint32_t)r.s2, s2h = (uint32_t)(r.s2 >> 32), s3l = (uint32_t)r.s3, s3h = (uint32_t)(r.s3 >> 32);
    const uint32_t bl = s3l ^ s1l, bh = s3h ^ s1h;                                      // s3 ^= s1
    const uint32_t n1l = __builtin_amdgcn_bitop3_b32(s1l, s2l, s0l, 0x96), n1h = __builtin_amdgcn_bitop3_b32(s1h, s2h, s0h, 0x96);   // s1 ^= (s2 ^ s0)
    const uint32_t n2l = __builtin_amdgcn_bitop3_b32(s2l, s0l, (uint32_t)t, 0x96), n2h = __builtin_amdgcn_bitop3_b32(s2h, s0h, (uint32_t)(t >> 32), 0x96);   // s2 = (s2 ^ s0) ^ t
    const uint32_t n0l = s0l ^ bl, n0h = s0h ^ bh;                                      // s0 ^= s3
    // rotl(b, 45) = rotr(b, 19): low word = (bh : bl) >> 19, high word = (bl : bh) >> 19
    const uint32_t n3l = __builtin_amdgcn_alignbit(bh, bl, 19), n3h = __builtin_amdgcn_alignbit(bl, bh, 19);
    r.s0 = (uint64_t)n0l | ((uint64_t)n0h << 32);
    r.s1 = (uint64_t)n1l | ((uint64_t)n1h << 32);
    r.s2 = (uint64_t)n2l | ((uint64_t)n2h << 32);
    r.s3 = (uint64_t)n3l | ((uint64_t)n3h << 32);
    return result;
}

// rand 0.8 Standard f32: (next_u32 >> 8) * 2^-24, next_u32 = next_u64 >> 32
// (The high word is pinned in a 32-bit register before the shift: left to itself the compiler converts the 64-BIT value
// `next >> 40` to f32, six VALU instructions where v_lshrrev + v_cvt_f32_u32 do.)
__device__ __forceinline__ uint32_t rng_next_u24(Rng &r) {
    uint32_t hi = (uint32_t)(rng_next_u64(r) >> 32);
    asm volatile("" : "+v"(hi));
    return hi >> 8;
}
__device__ __forceinline__ float rng_f32(Rng &r) { return (1.0f / 16777216.0f) * (float)rng_next_u24(r); }

// rand's f32 draw is k * 2^-24 with an integer k < 2^24. The expressions the reference builds on a fresh draw are exact in their
// first operations, so ONE correctly rounded operation on the integer yields the same bits:
//   2 * draw - 1  (math.rs:8,17-21,29): 2k * 2^-24 - 1 is representable            -> fma(k, 2^-23, -1), no rounding at all
//   draw * 2 * c  (math.rs:30):         draw * 2 is exact, c * 2^-23 is representable -> k * (c * 2^-23), the one rounding of the product
//   n + draw      (scene.rs:107-108):   k * 2^-24 is exact                           -> fma(k, 2^-24, n), the one rounding of the sum
__device__ __forceinline__ float rng_u24(Rng &r) { return (float)rng_next_u24(r); }
__device__ __forceinline__ float rng_pm1(Rng &r) { return __builtin_fmaf(rng_u24(r), 1.0f / 8388608.0f, -1.0f); }
__device__ __forceinline__ float rng_times_2c(Rng &r, float c) { return rng_u24(r) * (c * (1.0f / 8388608.0f)); }
__device__ __forceinline__ float rng_plus(Rng &r, float n) { return __builtin_fmaf(rng_u24(r), 1.0f / 16777216.0f, n); }

// ---- simd.rs:107-208 sinf_cosf (Cephes polynomial, lane 0 of the SSE2 code)
__device__ __forceinline__ void sinf_cosf_ref(float xin, float &sin_out, float &cos_out) {
    uint32_t sign_bit_sin = __float_as_uint(xin) & 0x80000000u;
    float x = __uint_as_float(__float_as_uint(xin) & 0x7fffffffu);
    float y = x * 1.27323954473516f;  // simd.rs:128
    int emm2 = (int)y;                // cvttps: truncate
    emm2 = (emm2 + 1) & ~1;           // simd.rs:134-135
    y = (float)emm2;
    int emm4 = emm2;
    const uint32_t swap_sign_bit_sin = ((uint32_t)(emm2 & 4)) << 29;
    const uint32_t poly_mask = ((emm2 & 2) == 0) ? 0xffffffffu : 0u;
    float xmm1 = y * -0.78515625f;  // simd.rs:152-160
    float xmm2 = y * -2.4187564849853515625e-4f;
    float xmm3 = y * -3.77489497744594108e-8f;
    x = x + xmm1;
    x = x + xmm2;
    x = x + xmm3;
    emm4 = emm4 - 2;
    const uint32_t sign_bit_cos = ((~(uint32_t)emm4) & 4u) << 29;
    sign_bit_sin ^= swap_sign_bit_sin;
    const float z = x * x;
    y = 2.443315711809948E-005f;  // simd.rs:170-181
    y = y * z;
    y = y + -1.388731625493765E-003f;
    y = y * z;
    y = y + 4.166664568298827E-002f;
    y = y * z;
    y = y * z;
    const float tmp = z * 0.5f;
    y = y - tmp;
    y = y + 1.0f;
    float y2 = -1.9515295891E-4f;  // simd.rs:184-191
    y2 = y2 * z;
    y2 = y2 + 8.3321608736E-3f;
    y2 = y2 * z;
    y2 = y2 + -1.6666654611E-1f;
    y2 = y2 * z;
    y2 = y2 * x;
    y2 = y2 + x;
    const float ysin2 = __uint_as_float(poly_mask & __float_as_uint(y2));  // simd.rs:194-201
    const float ysin1 = __uint_as_float(~poly_mask & __float_as_uint(y));
    y2 = y2 - ysin2;
    y = y - ysin1;
    const float s = ysin1 + ysin2;
    const float c = y + y2;
    sin_out = __uint_as_float(__float_as_uint(s) ^ sign_bit_sin);
    cos_out = __uint_as_float(__float_as_uint(c) ^ sign_bit_cos);
}

// f32::sin of a Noise texture's argument (texture.rs:86-89). The value only colours (it never feeds control flow), and the tests
// hold it to 2e-6 against the CPU's libm -- which differs from ANY other libm in the last ulp anyway. The device library's sinf
// carries its large-argument (Payne-Hanek) reduction branch-free in some kernels: 380 VALU instructions per call in the
// general-world kernel, 16 % of a `simple_light` frame. Arguments below 8192 -- `scale * z + 10 * turbulence` of any sensible
// scene -- take the Cephes evaluation the path already has for its own sin / cos (sinf_cosf_ref: three-part pi/4 reduction, < 2
// ulp there); a lane whose argument is larger, infinite or NaN takes the library's value.
__device__ __forceinline__ float sin_colour(float x) {
    float s, c;
    sinf_cosf_ref(x, s, c);
    // (the library call is wave-uniform for its cost, its RESULT is taken per lane: a lane's colour must not depend on which other
    // pixels happen to share its wave -- the work order, and with it a wave's composition, varies from run to run)
    const bool big = !(__builtin_fabsf(x) < 8192.0f);
    if (__builtin_expect(wave_any(big), 0)) {
        const float l = sinf(x);
        s = big ? l : s;
    }
    return s;
}

// ---- math.rs:6-34 sampling -------------------------------------------------
// math.rs:6-13 (runs even when lens_radius == 0; 2 draws per iteration)
__device__ __forceinline__ void random_in_unit_disk(Rng &rng, float &px, float &py) {
    for (;;) {
        const float x = rng_pm1(rng);   // a * 2 - 1
        const float y = rng_pm1(rng);
        // p.dot(p) with z = 0*2 - 0 = 0: (x*x + y*y) + 0*0
        if (((x * x + y * y) + 0.0f) < 1.0f) {
            px = x;
            py = y;
            return;
        }
    }
}

// math.rs:15-26
__device__ __forceinline__ f3 random_in_unit_sphere(Rng &rng) {
    for (;;) {
        const float a = rng_pm1(rng);   // 2 * draw - 1
        const float b = rng_pm1(rng);
        const float c = rng_pm1(rng);
        const f3 p = mk3(a, b, c);
        if (dot3(p, p) < 1.0f) return p;
    }
}

// math.rs:28-34
__device__ __forceinline__ f3 random_unit_vector(Rng &rng) {
    const float z = rng_pm1(rng);             // draw * 2 - 1
    const float a = rng_times_2c(rng, kPi);   // draw * 2 * PI
    const float r = sqrt_exact(1.0f - z * z);
    float sina, cosa;
    sinf_cosf_ref(a, sina, cosa);
    return mk3(r * cosa, r * sina, z);
}

// math.rs:61-63
__device__ __forceinline__ f3 reflect3(f3 v, f3 n) { return sub3(v, scale3(n, 2.0f * dot3(v, n))); }

// math.rs:65-73
__device__ __forceinline__ bool refract3(f3 v, f3 n, float ni_over_nt, f3 &out) {
    const float dt = dot3(v, n);
    const float discriminant = 1.0f - (ni_over_nt * ni_over_nt) * (1.0f - (dt * dt));
    if (discriminant > 0.0f) {
        out = sub3(scale3(sub3(v, scale3(n, dt)), ni_over_nt), scale3(n, sqrt_exact(discriminant)));
        return true;
    }
    return false;
}

// math.rs:76-80. powf(x, 5.0) is the one control-affecting libm call
// (material.rs:108-109). glibc's powf is not correctly rounded (<= 0.82 ULP)
// and even differs between its own FMA / non-FMA ifunc variants, so it cannot
// be matched bit-for-bit; we evaluate x^5 in binary64 (relative error < 2^-51)
// and round once, i.e. the correctly rounded result except for astronomically
// rare double-rounding ties. tests/test_gpu_parity.py measures the mismatch
// rate against the host powf.
__device__ __forceinline__ float pow5_ref(float x) {
    const double xd = (double)x;
    const double x2 = xd * xd;
    return (float)(x2 * x2 * xd);
}

__device__ __forceinline__ float schlick_ref(float cosine, float ref_idx) {
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * pow5_ref(1.0f - cosine);
}

// Rust `f32 as usize` (saturating, NaN -> 0) followed by `& 255`, branch-free. A float >= 2^31 is a multiple of 256
// (its ulp is), so its low byte is 0 -- except at saturation (f >= 2^64 -> usize::MAX, low byte 255); below 2^31 the
// 32-bit conversion is exact. v_cvt_u32_f32 itself maps NaN and negatives to 0.
__device__ __forceinline__ uint32_t floor_as_usize_low8(float f) {
    const uint32_t low = (uint32_t)__builtin_fminf(__builtin_fmaxf(f, 0.0f), 2147483520.0f) & 255u;   // NaN -> 0 (fmaxf returns the number)
    return f >= 18446744073709551616.0f ? 255u : (f >= 2147483648.0f ? 0u : low);
}

}  // namespace ptdev
