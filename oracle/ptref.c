/*
 * ptref.c -- CPU ORACLE: plain-C restatement of pathtrace-rs 0.1.2's hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see ptref.h). Never linked into the product.
 *
 * PARITY STATUS: "parity unpinned" against the Rust binary (the reference has
 * no tests / golden vectors and cannot be built here). Pinned to the public
 * xoshiro256+ / SplitMix64 vectors and to tests/golden/ fixtures.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (Rust never fuses a*b+c and
 * never reassociates), SSE2 scalar float math (no x87 excess precision).
 *
 * All citations are path:line under the reference repo root.
 *
 * Assumptions for arithmetic that lives in un-vendored crates:
 *  A1 rand_xoshiro 0.6.0 SplitMix64::next_u64:
 *       x += 0x9e3779b97f4a7c15; z = x;
 *       z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9;
 *       z = (z ^ (z >> 27)) * 0x94d049bb133111eb; return z ^ (z >> 31)
 *     Xoshiro256Plus::seed_from_u64(s) = four successive SplitMix64(s) outputs.
 *  A2 Xoshiro256Plus::next_u64: r = s0 + s3; t = s1 << 17; s2 ^= s0; s3 ^= s1;
 *       s1 ^= s2; s0 ^= s3; s2 ^= t; s3 = rotl(s3, 45); return r
 *     next_u32 = next_u64 >> 32.
 *  A3 rand 0.8.5 Standard f32: (next_u32 >> 8) as f32 * 2^-24.
 *  A4 rand 0.8.5 gen_range(low..high) for i32 (UniformInt::sample_single):
 *       range = high-low; zone = (range << lzcnt(range)) - 1;
 *       loop { v = next_u32; (hi, lo) = widening v*range; if lo <= zone return low+hi }
 *  A5 glam 0.20.5 scalar Vec3: dot = (x*x' + y*y') + z*z';
 *       length = sqrt(dot(v,v)); normalize = v * (1.0 / length)   [LEAST CERTAIN,
 *       control-affecting]; Vec3 / f32 = per-component true division;
 *       recip = 1.0/x per component; f32*Vec3 and Vec3*f32 per component;
 *       cross = (y*z' - y'*z, z*x' - z'*x, x*y' - x'*y).
 *  A6 glam 0.20.5 SSE2 Vec3A: min/max = _mm_min_ps/_mm_max_ps (return the
 *       SECOND operand when either is NaN); dot rounds like the scalar form.
 *  A7 Rust std f32::sin/powf/tan/floor/sqrt = glibc sinf/powf/tanf/floorf and
 *       IEEE sqrt; float->usize `as` casts saturate (NaN -> 0).
 *  A8 slice::sort_unstable_by tie order (pdqsort internals) is NOT reproduced:
 *       the BVH build below uses a stable merge sort. BVH topology affects
 *       speed and exact-t tie breaks only, never which spheres can be hit.
 */
#include "ptref.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

/* ======================================================================== */
/* Vec3 (glam 0.20.5 scalar Vec3; assumption A5)                            */
/* ======================================================================== */
typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_splat(float s) { return V3(s, s, s); }
static inline v3 v3_add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); } /* Vec3 * f32 and f32 * Vec3 */
static inline v3 v3_divs(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 v3_neg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float v3_length(v3 a) { return sqrtf(v3_dot(a, a)); }
static inline v3 v3_normalize(v3 a) { return v3_scale(a, 1.0f / v3_length(a)); }
static inline v3 v3_recip(v3 a) { return V3(1.0f / a.x, 1.0f / a.y, 1.0f / a.z); }
static inline v3 v3_cross(v3 a, v3 b) {
    return V3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline v3 v3_min(v3 a, v3 b) { /* Vec3::min: f32::min per component (no NaNs on these paths) */
    return V3(a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z);
}
static inline v3 v3_max(v3 a, v3 b) {
    return V3(a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z);
}

/* ======================================================================== */
/* RNG (assumptions A1-A4; call sites scene.rs:96-102, params.rs:21-27)     */
/* ======================================================================== */
typedef struct { uint64_t s[4]; } xoshiro;

static inline uint64_t splitmix64_next(uint64_t *x) {
    *x += 0x9e3779b97f4a7c15ULL;
    uint64_t z = *x;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

static inline void xoshiro_seed_from_u64(xoshiro *r, uint64_t seed) {
    uint64_t x = seed;
    for (int i = 0; i < 4; ++i) r->s[i] = splitmix64_next(&x);
}

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

static inline uint64_t xoshiro_next_u64(xoshiro *r) {
    uint64_t result = r->s[0] + r->s[3];
    uint64_t t = r->s[1] << 17;
    r->s[2] ^= r->s[0];
    r->s[3] ^= r->s[1];
    r->s[1] ^= r->s[2];
    r->s[0] ^= r->s[3];
    r->s[2] ^= t;
    r->s[3] = rotl64(r->s[3], 45);
    return result;
}

static inline uint32_t xoshiro_next_u32(xoshiro *r) { return (uint32_t)(xoshiro_next_u64(r) >> 32); }

/* rng.gen::<f32>() */
static inline float gen_f32(xoshiro *r) {
    uint32_t v = xoshiro_next_u32(r) >> 8;
    return (1.0f / 16777216.0f) * (float)v;
}

/* rng.gen_range(low..high) for i32 -- bvh.rs:269 */
static int32_t gen_range_i32(xoshiro *r, int32_t low, int32_t high) {
    uint32_t range = (uint32_t)(high - low);
    uint32_t zone = (range << __builtin_clz(range)) - 1u;
    for (;;) {
        uint32_t v = xoshiro_next_u32(r);
        uint64_t m = (uint64_t)v * (uint64_t)range;
        uint32_t hi = (uint32_t)(m >> 32), lo = (uint32_t)m;
        if (lo <= zone) return low + (int32_t)hi;
    }
}

/* ======================================================================== */
/* simd.rs:85-208  sinf_cosf (Cephes / sse_mathfun, lane 0 of the SSE2 code) */
/* ======================================================================== */
static inline float f32_from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f32_to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static void sinf_cosf(float xin, float *sin_out, float *cos_out) {
    /* simd.rs:121-125 */
    uint32_t sign_bit_sin = f32_to_bits(xin) & 0x80000000u;
    float x = f32_from_bits(f32_to_bits(xin) & 0x7fffffffu);
    /* simd.rs:128 scale by 4/Pi */
    float y = x * 1.27323954473516f;
    /* simd.rs:131-136 */
    int32_t emm2 = (int32_t)y; /* _mm_cvttps_epi32: truncate */
    emm2 = emm2 + 1;
    emm2 = emm2 & ~1;
    y = (float)emm2;
    int32_t emm4 = emm2;
    /* simd.rs:141-143 swap sign flag for sine */
    uint32_t swap_sign_bit_sin = ((uint32_t)(emm2 & 4)) << 29;
    /* simd.rs:146-148 polynom selection mask */
    uint32_t poly_mask = ((emm2 & 2) == 0) ? 0xffffffffu : 0u;
    /* simd.rs:152-160 extended precision modular arithmetic */
    float xmm1 = y * -0.78515625f;
    float xmm2 = y * -2.4187564849853515625e-4f;
    float xmm3 = y * -3.77489497744594108e-8f;
    x = x + xmm1;
    x = x + xmm2;
    x = x + xmm3;
    /* simd.rs:162-165 */
    emm4 = emm4 - 2;
    uint32_t sign_bit_cos = ((~(uint32_t)emm4) & 4u) << 29; /* andnot(emm4, 4) */
    /* simd.rs:167 */
    sign_bit_sin ^= swap_sign_bit_sin;
    /* simd.rs:170-181 first polynom (0 <= x <= Pi/4) */
    float z = x * x;
    y = 2.443315711809948E-005f;
    y = y * z;
    y = y + -1.388731625493765E-003f;
    y = y * z;
    y = y + 4.166664568298827E-002f;
    y = y * z;
    y = y * z;
    float tmp = z * 0.5f;
    y = y - tmp;
    y = y + 1.0f;
    /* simd.rs:184-191 second polynom */
    float y2 = -1.9515295891E-4f;
    y2 = y2 * z;
    y2 = y2 + 8.3321608736E-3f;
    y2 = y2 * z;
    y2 = y2 + -1.6666654611E-1f;
    y2 = y2 * z;
    y2 = y2 * x;
    y2 = y2 + x;
    /* simd.rs:194-201 select */
    float ysin2 = f32_from_bits(poly_mask & f32_to_bits(y2));
    float ysin1 = f32_from_bits(~poly_mask & f32_to_bits(y));
    y2 = y2 - ysin2;
    y = y - ysin1;
    float s = ysin1 + ysin2;
    float c = y + y2;
    /* simd.rs:204-207 update the sign */
    *sin_out = f32_from_bits(f32_to_bits(s) ^ sign_bit_sin);
    *cos_out = f32_from_bits(f32_to_bits(c) ^ sign_bit_cos);
}

/* ======================================================================== */
/* math.rs:6-34,61-80 sampling + reflect/refract/schlick                    */
/* ======================================================================== */
#define PT_PI 3.14159274101257324f /* f32::consts::PI */

/* math.rs:6-13 */
static v3 random_in_unit_disk(xoshiro *rng) {
    for (;;) {
        float a = gen_f32(rng);
        float b = gen_f32(rng);
        v3 p = v3_sub(v3_scale(V3(a, b, 0.0f), 2.0f), V3(1.0f, 1.0f, 0.0f));
        if (v3_dot(p, p) < 1.0f) return p;
    }
}

/* math.rs:15-26 */
static v3 random_in_unit_sphere(xoshiro *rng) {
    for (;;) {
        float a = 2.0f * gen_f32(rng) - 1.0f;
        float b = 2.0f * gen_f32(rng) - 1.0f;
        float c = 2.0f * gen_f32(rng) - 1.0f;
        v3 p = V3(a, b, c);
        if (v3_dot(p, p) < 1.0f) return p;
    }
}

/* math.rs:28-34 */
static v3 random_unit_vector(xoshiro *rng) {
    float z = gen_f32(rng) * 2.0f - 1.0f;
    float a = gen_f32(rng) * 2.0f * PT_PI;
    float r = sqrtf(1.0f - z * z);
    float sina, cosa;
    sinf_cosf(a, &sina, &cosa);
    return V3(r * cosa, r * sina, z);
}

/* math.rs:61-63 */
static inline v3 reflect(v3 v, v3 n) { return v3_sub(v, v3_scale(n, 2.0f * v3_dot(v, n))); }

/* math.rs:65-73 */
static int refract(v3 v, v3 n, float ni_over_nt, v3 *out) {
    float dt = v3_dot(v, n);
    float discriminant = 1.0f - (ni_over_nt * ni_over_nt) * (1.0f - (dt * dt));
    if (discriminant > 0.0f) {
        *out = v3_sub(v3_scale(v3_sub(v, v3_scale(n, dt)), ni_over_nt),
                      v3_scale(n, sqrtf(discriminant)));
        return 1;
    }
    return 0;
}

/* math.rs:76-80 */
static inline float schlick(float cosine, float ref_idx) {
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * powf(1.0f - cosine, 5.0f);
}

/* ======================================================================== */
/* perlin.rs:7-111                                                          */
/* ======================================================================== */
typedef struct {
    v3 randvec[256];
    uint32_t perm_x[256], perm_y[256], perm_z[256];
} perlin;

/* perlin.rs:27-41 */
static void perlin_generate_perm(xoshiro *rng, uint32_t *perm) {
    for (int i = 0; i < 256; ++i) perm[i] = (uint32_t)i;
    for (int i = 255; i >= 0; --i) {
        size_t target = (size_t)floorf(gen_f32(rng) * (float)(i + 1));
        uint32_t t = perm[i]; perm[i] = perm[target]; perm[target] = t;
    }
}

/* perlin.rs:15-25,43-51 */
static void perlin_new(perlin *p, xoshiro *rng) {
    for (int i = 0; i < 256; ++i) {
        float a = -1.0f + 2.0f * gen_f32(rng);
        float b = -1.0f + 2.0f * gen_f32(rng);
        float c = -1.0f + 2.0f * gen_f32(rng);
        p->randvec[i] = v3_normalize(V3(a, b, c));
    }
    perlin_generate_perm(rng, p->perm_x);
    perlin_generate_perm(rng, p->perm_y);
    perlin_generate_perm(rng, p->perm_z);
}

/* Rust `f32 as usize`: saturating, NaN -> 0 (assumption A7) */
static inline uint64_t f32_as_usize(float f) {
    if (!(f > 0.0f)) return 0;
    if (f >= 18446744073709551616.0f) return UINT64_MAX;
    return (uint64_t)f;
}

/* perlin.rs:54-74 */
static float perlin_interpolate(v3 c[2][2][2], float u, float v, float w) {
    float uu = u * u * (3.0f - 2.0f * u);
    float vv = v * v * (3.0f - 2.0f * v);
    float ww = w * w * (3.0f - 2.0f * w);
    float accum = 0.0f;
    for (int i = 0; i < 2; ++i) {
        float ii = (float)i;
        for (int j = 0; j < 2; ++j) {
            float jj = (float)j;
            for (int k = 0; k < 2; ++k) {
                float kk = (float)k;
                v3 weight = V3(u - ii, v - jj, w - kk);
                accum += (ii * uu + (1.0f - ii) * (1.0f - uu)) *
                         (jj * vv + (1.0f - jj) * (1.0f - vv)) *
                         (kk * ww + (1.0f - kk) * (1.0f - ww)) *
                         v3_dot(c[i][j][k], weight);
            }
        }
    }
    return accum;
}

/* perlin.rs:89-111 */
static float perlin_noise(const perlin *pn, v3 p) {
    float x = p.x, y = p.y, z = p.z;
    float u = x - floorf(x);
    float v = y - floorf(y);
    float w = z - floorf(z);
    uint64_t i = f32_as_usize(floorf(x));
    uint64_t j = f32_as_usize(floorf(y));
    uint64_t k = f32_as_usize(floorf(z));
    v3 c[2][2][2];
    for (uint64_t di = 0; di < 2; ++di)
        for (uint64_t dj = 0; dj < 2; ++dj)
            for (uint64_t dk = 0; dk < 2; ++dk)
                c[di][dj][dk] = pn->randvec[pn->perm_x[(i + di) & 255] ^
                                           pn->perm_y[(j + dj) & 255] ^
                                           pn->perm_z[(k + dk) & 255]];
    return perlin_interpolate(c, u, v, w);
}

/* perlin.rs:76-87 */
static float perlin_turb(const perlin *pn, v3 p) {
    float accum = 0.0f;
    v3 temp_p = p;
    float weight = 1.0f;
    for (int d = 0; d < 7; ++d) {
        accum += weight * perlin_noise(pn, temp_p);
        weight *= 0.5f;
        temp_p = v3_scale(temp_p, 2.0f);
    }
    return fabsf(accum);
}

/* ======================================================================== */
/* texture.rs:40-91                                                         */
/* ======================================================================== */
enum { TEX_CONSTANT = 0, TEX_CHECKER = 1, TEX_NOISE = 2 };
typedef struct texture {
    int kind;
    v3 color;
    const struct texture *odd, *even;
    const perlin *noise;
    float scale;
} texture;

/* texture.rs:74-91 */
static v3 texture_value(const texture *t, float u, float v, v3 p) {
    switch (t->kind) {
    case TEX_CONSTANT:
        return t->color;
    case TEX_CHECKER: {
        v3 s = v3_mul(V3(10.0f, 10.0f, 10.0f), p);
        float sines = sinf(s.x) * sinf(s.y) * sinf(s.z);
        if (sines < 0.0f) return texture_value(t->odd, u, v, p);
        return texture_value(t->even, u, v, p);
    }
    default: /* TEX_NOISE */
        return v3_scale(v3_scale(V3(1.0f, 1.0f, 1.0f), 0.5f),
                        1.0f + sinf(t->scale * p.z + 10.0f * perlin_turb(t->noise, p)));
    }
}

/* ======================================================================== */
/* collision/ray.rs:4-26,43-50                                              */
/* ======================================================================== */
typedef struct { v3 origin, direction, rcp_direction; float time; } ray;
typedef struct { v3 point, normal; float t, u, v; } ray_hit;

/* ray.rs:13-21 */
static inline ray ray_new(v3 origin, v3 direction, float time) {
    ray r; r.origin = origin; r.direction = direction;
    r.rcp_direction = v3_recip(direction); r.time = time; return r;
}
/* ray.rs:24-26 */
static inline v3 point_at_parameter(const ray *r, float t) { return v3_add(r->origin, v3_scale(r->direction, t)); }

/* ======================================================================== */
/* material.rs:13-167                                                       */
/* ======================================================================== */
enum { MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_DIELECTRIC = 2, MAT_DIFFUSE_LIGHT = 3 };
typedef struct {
    int kind;
    const texture *tex; /* Lambertian albedo / DiffuseLight emit */
    v3 albedo;          /* Metal */
    float fuzz;         /* Metal */
    float ref_idx;      /* Dielectric */
} material;

/* material.rs:52-67 */
static int scatter_lambertian(const texture *albedo, const ray *ray_in, const ray_hit *hit,
                              xoshiro *rng, v3 *attenuation, ray *scattered) {
    v3 target = v3_add(v3_add(hit->point, hit->normal), random_unit_vector(rng));
    *attenuation = texture_value(albedo, hit->u, hit->v, hit->point);
    *scattered = ray_new(hit->point, v3_normalize(v3_sub(target, hit->point)), ray_in->time);
    return 1;
}

/* material.rs:69-89 */
static int scatter_metal(v3 albedo, float fuzz, const ray *ray_in, const ray_hit *hit,
                         xoshiro *rng, v3 *attenuation, ray *scattered) {
    v3 reflected = reflect(ray_in->direction, hit->normal);
    if (v3_dot(reflected, hit->normal) > 0.0f) {
        *attenuation = albedo;
        v3 rs = random_in_unit_sphere(rng);
        *scattered = ray_new(hit->point, v3_normalize(v3_add(reflected, v3_scale(rs, fuzz))), ray_in->time);
        return 1;
    }
    return 0;
}

/* material.rs:91-124 */
static int scatter_dielectric(float ref_idx, const ray *ray_in, const ray_hit *hit,
                              xoshiro *rng, v3 *attenuation, ray *scattered) {
    *attenuation = V3(1.0f, 1.0f, 1.0f);
    float rdotn = v3_dot(ray_in->direction, hit->normal);
    v3 outward_normal; float ni_over_nt, cosine;
    if (rdotn > 0.0f) {
        cosine = rdotn / v3_length(ray_in->direction);
        cosine = sqrtf(1.0f - ref_idx * ref_idx * (1.0f - cosine * cosine));
        outward_normal = v3_neg(hit->normal); ni_over_nt = ref_idx;
    } else {
        cosine = -rdotn / v3_length(ray_in->direction);
        outward_normal = hit->normal; ni_over_nt = 1.0f / ref_idx;
    }
    v3 refracted;
    if (refract(ray_in->direction, outward_normal, ni_over_nt, &refracted)) {
        float reflect_prob = schlick(cosine, ref_idx);
        if (gen_f32(rng) > reflect_prob) {
            *scattered = ray_new(hit->point, v3_normalize(refracted), ray_in->time);
            return 1;
        }
    }
    *scattered = ray_new(hit->point, v3_normalize(reflect(ray_in->direction, hit->normal)), ray_in->time);
    return 1;
}

/* material.rs:138-159 */
static int material_scatter(const material *m, const ray *ray_in, const ray_hit *hit,
                            xoshiro *rng, v3 *attenuation, ray *scattered) {
    switch (m->kind) {
    case MAT_LAMBERTIAN: return scatter_lambertian(m->tex, ray_in, hit, rng, attenuation, scattered);
    case MAT_METAL: return scatter_metal(m->albedo, m->fuzz, ray_in, hit, rng, attenuation, scattered);
    case MAT_DIELECTRIC: return scatter_dielectric(m->ref_idx, ray_in, hit, rng, attenuation, scattered);
    default: return 0; /* DiffuseLight */
    }
}

/* material.rs:161-167 */
static v3 material_emitted(const material *m, float u, float v, v3 point) {
    if (m->kind == MAT_DIFFUSE_LIGHT) return texture_value(m->tex, u, v, point);
    return V3(0.0f, 0.0f, 0.0f);
}

/* ======================================================================== */
/* collision/aabb.rs:8-72 (assumption A6 for the Vec3A min/max NaN rule)    */
/* ======================================================================== */
typedef struct { v3 min, max; } aabb;

static inline float sse_min(float a, float b) { return a < b ? a : b; } /* _mm_min_ps */
static inline float sse_max(float a, float b) { return a > b ? a : b; } /* _mm_max_ps */

/* aabb.rs:46-58 */
static int aabb_ray_hit(const aabb *bb, const ray *r, float tmin, float tmax) {
    v3 min_delta = v3_mul(v3_sub(bb->min, r->origin), r->rcp_direction);
    v3 max_delta = v3_mul(v3_sub(bb->max, r->origin), r->rcp_direction);
    v3 t0 = V3(sse_min(min_delta.x, max_delta.x), sse_min(min_delta.y, max_delta.y), sse_min(min_delta.z, max_delta.z));
    v3 t1 = V3(sse_max(min_delta.x, max_delta.x), sse_max(min_delta.y, max_delta.y), sse_max(min_delta.z, max_delta.z));
    v3 vmin = V3(sse_max(t0.x, tmin), sse_max(t0.y, tmin), sse_max(t0.z, tmin));
    v3 vmax = V3(sse_min(t1.x, tmax), sse_min(t1.y, tmax), sse_min(t1.z, tmax));
    return (vmax.x > vmin.x) && (vmax.y > vmin.y) && (vmax.z > vmin.z);
}

/* aabb.rs:61-66 */
static inline aabb aabb_add(aabb a, aabb b) { aabb r = { v3_min(a.min, b.min), v3_max(a.max, b.max) }; return r; }

/* ======================================================================== */
/* collision/sphere.rs:8-75                                                 */
/* ======================================================================== */
typedef struct { v3 centre; float radius; } sphere;

/* sphere.rs:29-66 */
static int sphere_ray_hit(const sphere *s, const ray *r, float t_min, float t_max, ray_hit *out) {
    v3 oc = v3_sub(r->origin, s->centre);
    float a = v3_dot(r->direction, r->direction);
    float b = v3_dot(oc, r->direction);
    float c = v3_dot(oc, oc) - s->radius * s->radius;
    float discriminant = b * b - a * c;
    if (discriminant > 0.0f) {
        float discriminant_sqrt = sqrtf(discriminant);
        float t = (-b - discriminant_sqrt) / a;
        if (t < t_max && t > t_min) {
            out->point = point_at_parameter(r, t);
            out->normal = v3_divs(v3_sub(out->point, s->centre), s->radius);
            out->t = t; out->u = 0.0f; out->v = 0.0f;
            return 1;
        }
        t = (-b + discriminant_sqrt) / a;
        if (t < t_max && t > t_min) {
            out->point = point_at_parameter(r, t);
            out->normal = v3_divs(v3_sub(out->point, s->centre), s->radius);
            out->t = t; out->u = 0.0f; out->v = 0.0f;
            return 1;
        }
    }
    return 0;
}

/* sphere.rs:69-75 */
static inline aabb sphere_bounding_box(const sphere *s) {
    v3 rad = v3_splat(s->radius);
    aabb r = { v3_sub(s->centre, rad), v3_add(s->centre, rad) };
    return r;
}

/* ======================================================================== */
/* collision/hitable.rs:12-65, hitable_list.rs:40-56, bvh.rs:24-62          */
/* ======================================================================== */
enum { HIT_SPHERE = 0, HIT_BVHNODE = 1, HIT_LIST = 2 };
struct bvhnode; struct hitable_list;
typedef struct hitable {
    int kind;
    const sphere *sph; const material *mat;  /* Sphere(&Sphere,&Material) */
    const struct bvhnode *node;              /* BVHNode(&BVHNode) */
    const struct hitable_list *list;         /* List(&HitableList) */
} hitable;
typedef struct bvhnode { aabb bb; hitable lhs, rhs; } bvhnode;
typedef struct hitable_list { hitable *hitables; size_t len; } hitable_list;

static int hitable_ray_hit(const hitable *h, const ray *r, float t_min, float t_max,
                           ray_hit *out, const material **mat);

/* hitable_list.rs:40-56 */
static int list_ray_hit(const hitable_list *l, const ray *r, float t_min, float t_max,
                        ray_hit *out, const material **mat) {
    int found = 0;
    float closest_so_far = t_max;
    for (size_t i = 0; i < l->len; ++i) {
        ray_hit h; const material *m;
        if (hitable_ray_hit(&l->hitables[i], r, t_min, closest_so_far, &h, &m)) {
            *out = h; *mat = m; found = 1;
            closest_so_far = h.t;
        }
    }
    return found;
}

/* bvh.rs:37-62 */
static int bvh_ray_hit(const bvhnode *n, const ray *r, float t_min, float t_max,
                       ray_hit *out, const material **mat) {
    if (aabb_ray_hit(&n->bb, r, t_min, t_max)) {
        ray_hit hl, hr; const material *ml, *mr;
        int has_l = hitable_ray_hit(&n->lhs, r, t_min, t_max, &hl, &ml);
        int has_r = hitable_ray_hit(&n->rhs, r, t_min, t_max, &hr, &mr);
        if (has_l && has_r) {
            if (hl.t < hr.t) { *out = hl; *mat = ml; } else { *out = hr; *mat = mr; }
            return 1;
        }
        if (has_l) { *out = hl; *mat = ml; return 1; }
        if (has_r) { *out = hr; *mat = mr; return 1; }
        return 0;
    }
    return 0;
}

/* hitable.rs:39-65 (Sphere / BVHNode / List arms) */
static int hitable_ray_hit(const hitable *h, const ray *r, float t_min, float t_max,
                           ray_hit *out, const material **mat) {
    switch (h->kind) {
    case HIT_BVHNODE: return bvh_ray_hit(h->node, r, t_min, t_max, out, mat);
    case HIT_LIST: return list_ray_hit(h->list, r, t_min, t_max, out, mat);
    default:
        if (sphere_ray_hit(h->sph, r, t_min, t_max, out)) { *mat = h->mat; return 1; }
        return 0;
    }
}

/* hitable.rs:25-36 (t0 = t1 = 0) */
static aabb hitable_bounding_box(const hitable *h) {
    if (h->kind == HIT_BVHNODE) return h->node->bb;
    return sphere_bounding_box(h->sph);
}

/* ======================================================================== */
/* camera.rs:8-68                                                           */
/* ======================================================================== */
typedef struct {
    v3 origin, lower_left_corner, horizontal, vertical, u, v, w;
    float time0, time1, lens_radius;
} camera;

/* camera.rs:22-54 */
static camera camera_new(v3 lookfrom, v3 lookat, v3 vup, float vfov, float aspect,
                         float aperture, float focus_dist, float time0, float time1) {
    float theta = vfov * PT_PI / 180.0f;
    float half_height = tanf(theta * 0.5f);
    float half_width = aspect * half_height;
    v3 w = v3_normalize(v3_sub(lookfrom, lookat));
    v3 u = v3_normalize(v3_cross(vup, w));
    v3 v = v3_cross(w, u);
    camera c;
    c.origin = lookfrom;
    c.lower_left_corner = v3_sub(v3_sub(v3_sub(lookfrom, v3_scale(u, half_width * focus_dist)),
                                        v3_scale(v, half_height * focus_dist)),
                                 v3_scale(w, focus_dist));
    c.horizontal = v3_scale(u, 2.0f * half_width * focus_dist);
    c.vertical = v3_scale(v, 2.0f * half_height * focus_dist);
    c.u = u; c.v = v; c.w = w;
    c.time0 = time0; c.time1 = time1;
    c.lens_radius = aperture * 0.5f;
    return c;
}

/* camera.rs:56-68 */
static ray camera_get_ray(const camera *c, float s, float t, xoshiro *rng) {
    v3 rd = v3_scale(random_in_unit_disk(rng), c->lens_radius);
    v3 offset = v3_add(v3_scale(c->u, rd.x), v3_scale(c->v, rd.y));
    float time = c->time0 + gen_f32(rng) * (c->time1 - c->time0);
    v3 dir = v3_sub(v3_sub(v3_add(v3_add(c->lower_left_corner, v3_scale(c->horizontal, s)),
                                  v3_scale(c->vertical, t)),
                           c->origin),
                    offset);
    return ray_new(v3_add(c->origin, offset), v3_normalize(dir), time);
}

/* ======================================================================== */
/* storage.rs:12-43 -- arenas (append-only pools, allocation order kept)    */
/* ======================================================================== */
typedef struct {
    texture *textures; size_t n_textures, cap_textures;
    material *materials; size_t n_materials, cap_materials;
    sphere *spheres; size_t n_spheres, cap_spheres;
    bvhnode *nodes; size_t n_nodes, cap_nodes;
    perlin perlin_noise;
} storage;

struct ora_scene {
    storage st;
    hitable world;          /* scene.rs:19 */
    hitable_list list;      /* storage.rs:86 alloc_hitables */
    int has_sky; v3 sky;    /* scene.rs:20 */
    camera cam;
    int use_bvh;
    uint64_t build_draws;
};

/* arenas are sized up-front so pointers stay stable (typed_arena semantics) */
static void storage_init(storage *st, xoshiro *rng, size_t max_items) {
    memset(st, 0, sizeof(*st));
    st->cap_textures = max_items + 16; st->textures = calloc(st->cap_textures, sizeof(texture));
    st->cap_materials = max_items + 16; st->materials = calloc(st->cap_materials, sizeof(material));
    st->cap_spheres = max_items + 16; st->spheres = calloc(st->cap_spheres, sizeof(sphere));
    st->cap_nodes = max_items + 16; st->nodes = calloc(st->cap_nodes, sizeof(bvhnode));
    perlin_new(&st->perlin_noise, rng); /* storage.rs:41 */
}
static void storage_free(storage *st) { free(st->textures); free(st->materials); free(st->spheres); free(st->nodes); }

static const texture *alloc_texture(storage *st, texture t) { st->textures[st->n_textures] = t; return &st->textures[st->n_textures++]; }
static const material *alloc_material(storage *st, material m) { st->materials[st->n_materials] = m; return &st->materials[st->n_materials++]; }
static const sphere *alloc_sphere(storage *st, sphere s) { st->spheres[st->n_spheres] = s; return &st->spheres[st->n_spheres++]; }
static bvhnode *alloc_bvhnode(storage *st, hitable lhs, hitable rhs, aabb bb) {
    bvhnode *n = &st->nodes[st->n_nodes++]; n->bb = bb; n->lhs = lhs; n->rhs = rhs; return n;
}

/* texture.rs:57-67 */
static texture tex_constant(v3 color) { texture t; memset(&t, 0, sizeof t); t.kind = TEX_CONSTANT; t.color = color; return t; }
static texture tex_checker(const texture *odd, const texture *even) { texture t; memset(&t, 0, sizeof t); t.kind = TEX_CHECKER; t.odd = odd; t.even = even; return t; }
static texture tex_noise(const perlin *n, float scale) { texture t; memset(&t, 0, sizeof t); t.kind = TEX_NOISE; t.noise = n; t.scale = scale; return t; }
/* material.rs:21-35 */
static material mat_lambertian(const texture *albedo) { material m; memset(&m, 0, sizeof m); m.kind = MAT_LAMBERTIAN; m.tex = albedo; return m; }
static material mat_metal(v3 albedo, float fuzz) { material m; memset(&m, 0, sizeof m); m.kind = MAT_METAL; m.albedo = albedo; m.fuzz = fuzz; return m; }
static material mat_dielectric(float ref_idx) { material m; memset(&m, 0, sizeof m); m.kind = MAT_DIELECTRIC; m.ref_idx = ref_idx; return m; }
static material mat_diffuse_light(const texture *emit) { material m; memset(&m, 0, sizeof m); m.kind = MAT_DIFFUSE_LIGHT; m.tex = emit; return m; }

typedef struct { hitable *v; size_t len, cap; } hitvec;
static void hv_push(hitvec *hv, hitable h) {
    if (hv->len == hv->cap) { hv->cap = hv->cap ? hv->cap * 2 : 64; hv->v = realloc(hv->v, hv->cap * sizeof(hitable)); }
    hv->v[hv->len++] = h;
}
/* the `sphere` closure of presets.rs:115-120: sphere arena first, then material arena */
static hitable mk_sphere(storage *st, v3 centre, float radius, material m) {
    hitable h; memset(&h, 0, sizeof h); h.kind = HIT_SPHERE;
    sphere s = { centre, radius };
    h.sph = alloc_sphere(st, s);
    h.mat = alloc_material(st, m);
    return h;
}

/* ======================================================================== */
/* presets.rs                                                               */
/* ======================================================================== */

/* presets.rs:89-215 random_impl(only_spheres = true) */
static void preset_random_spheres(ora_scene *sc, uint32_t width, uint32_t height, xoshiro *rng, hitvec *hv) {
    storage *st = &sc->st;
    sc->cam = camera_new(V3(13.0f, 2.0f, 3.0f), V3(0.0f, 0.0f, 0.0f), V3(0.0f, 1.0f, 0.0f), 20.0f,
                         (float)width / (float)height, 0.1f, 10.0f, 0.0f, 1.0f);
    /* presets.rs:132-139: constant(odd), constant(even), checker, then sphere+material */
    const texture *odd = alloc_texture(st, tex_constant(V3(0.2f, 0.3f, 0.1f)));
    const texture *even = alloc_texture(st, tex_constant(V3(0.9f, 0.9f, 0.9f)));
    const texture *chk = alloc_texture(st, tex_checker(odd, even));
    hv_push(hv, mk_sphere(st, V3(0.0f, -1000.0f, 0.0f), 1000.0f, mat_lambertian(chk)));
    for (int a = -11; a < 11; ++a) {
        for (int b = -11; b < 11; ++b) {
            float choose_material = gen_f32(rng);
            float cx = (float)a + 0.9f * gen_f32(rng);
            float cz = (float)b + 0.9f * gen_f32(rng);
            v3 centre = V3(cx, 0.2f, cz);
            if (choose_material < 0.8f) {
                (void)gen_f32(rng); /* presets.rs:150 centre1: consumed even for only_spheres */
                float r0 = gen_f32(rng), r1 = gen_f32(rng), r2 = gen_f32(rng),
                      r3 = gen_f32(rng), r4 = gen_f32(rng), r5 = gen_f32(rng);
                const texture *t = alloc_texture(st, tex_constant(V3(r0 * r1, r2 * r3, r4 * r5)));
                hv_push(hv, mk_sphere(st, centre, 0.2f, mat_lambertian(t)));
            } else if (choose_material < 0.95f) {
                float ax = 0.5f * (1.0f + gen_f32(rng));
                float ay = 0.5f * (1.0f + gen_f32(rng));
                float az = 0.5f * (1.0f + gen_f32(rng));
                float fuzz = 0.5f * gen_f32(rng);
                hv_push(hv, mk_sphere(st, centre, 0.2f, mat_metal(V3(ax, ay, az), fuzz)));
            } else {
                hv_push(hv, mk_sphere(st, centre, 0.2f, mat_dielectric(1.5f)));
            }
        }
    }
    hv_push(hv, mk_sphere(st, V3(0.0f, 1.0f, 0.0f), 1.0f, mat_dielectric(1.5f)));
    const texture *t = alloc_texture(st, tex_constant(V3(0.4f, 0.2f, 0.1f)));
    hv_push(hv, mk_sphere(st, V3(-4.0f, 1.0f, 0.0f), 1.0f, mat_lambertian(t)));
    hv_push(hv, mk_sphere(st, V3(4.0f, 1.0f, 0.0f), 1.0f, mat_metal(V3(0.7f, 0.6f, 0.5f), 0.0f)));
    sc->has_sky = 0;
}

/* presets.rs:217-269 */
static void preset_small(ora_scene *sc, uint32_t width, uint32_t height, hitvec *hv) {
    storage *st = &sc->st;
    v3 lookfrom = V3(3.0f, 3.0f, 2.0f), lookat = V3(0.0f, 0.0f, -1.0f);
    float dist_to_focus = v3_length(v3_sub(lookfrom, lookat));
    sc->cam = camera_new(lookfrom, lookat, V3(0.0f, 1.0f, 0.0f), 20.0f, (float)width / (float)height,
                         0.1f, dist_to_focus, 0.0f, 1.0f);
    /* argument evaluation order: centre, radius, then material (texture alloc), then the closure body */
    const texture *t0 = alloc_texture(st, tex_constant(V3(0.1f, 0.2f, 0.5f)));
    hv_push(hv, mk_sphere(st, V3(0.0f, 0.0f, -1.0f), 0.5f, mat_lambertian(t0)));
    const texture *t1 = alloc_texture(st, tex_constant(V3(0.8f, 0.8f, 0.0f)));
    hv_push(hv, mk_sphere(st, V3(0.0f, -100.5f, -1.0f), 100.0f, mat_lambertian(t1)));
    hv_push(hv, mk_sphere(st, V3(1.0f, 0.0f, -1.0f), 0.5f, mat_metal(V3(0.8f, 0.6f, 0.2f), 0.0f)));
    hv_push(hv, mk_sphere(st, V3(-1.0f, 0.0f, -1.0f), 0.5f, mat_dielectric(1.5f)));
    hv_push(hv, mk_sphere(st, V3(-1.0f, 0.0f, -1.0f), -0.45f, mat_dielectric(1.5f)));
    sc->has_sky = 0;
}

/* presets.rs:271-315 */
static void preset_two_perlin_spheres(ora_scene *sc, uint32_t width, uint32_t height, hitvec *hv) {
    storage *st = &sc->st;
    sc->cam = camera_new(V3(13.0f, 2.0f, 3.0f), V3(0.0f, 0.0f, 0.0f), V3(0.0f, 1.0f, 0.0f), 20.0f,
                         (float)width / (float)height, 0.0f, 10.0f, 0.0f, 0.0f);
    const texture *noise_texture = alloc_texture(st, tex_noise(&st->perlin_noise, 4.0f));
    hv_push(hv, mk_sphere(st, V3(0.0f, -1000.0f, 0.0f), 1000.0f, mat_lambertian(noise_texture)));
    hv_push(hv, mk_sphere(st, V3(0.0f, 2.0f, 0.0f), 2.0f, mat_lambertian(noise_texture)));
    sc->has_sky = 0;
}

/* presets.rs:595-851 (commented-out `aras_p`, written against an older
 * Camera::new without time0/time1 -> time0 = time1 = 0 here; both
 * diffuse_light spheres are kept; sky = None as for every list preset) */
static void preset_aras(ora_scene *sc, uint32_t width, uint32_t height, hitvec *hv) {
    storage *st = &sc->st;
    sc->cam = camera_new(V3(0.0f, 2.0f, 3.0f), V3(0.0f, 0.0f, 0.0f), V3(0.0f, 1.0f, 0.0f), 60.0f,
                         (float)width / (float)height, 0.02f, 3.0f, 0.0f, 0.0f);
#define LAMB(cx, cy, cz, r, ax, ay, az) do { const texture *t_ = alloc_texture(st, tex_constant(V3(ax, ay, az))); \
        hv_push(hv, mk_sphere(st, V3(cx, cy, cz), r, mat_lambertian(t_))); } while (0)
#define METAL(cx, cy, cz, r, ax, ay, az, fz) hv_push(hv, mk_sphere(st, V3(cx, cy, cz), r, mat_metal(V3(ax, ay, az), fz)))
#define LIGHT(cx, cy, cz, r, ax, ay, az) do { const texture *t_ = alloc_texture(st, tex_constant(V3(ax, ay, az))); \
        hv_push(hv, mk_sphere(st, V3(cx, cy, cz), r, mat_diffuse_light(t_))); } while (0)
    LAMB(0.0f, -100.5f, -1.0f, 100.0f, 0.8f, 0.8f, 0.8f);   /* presets.rs:622-626 */
    LAMB(2.0f, 0.0f, -1.0f, 0.5f, 0.8f, 0.4f, 0.4f);
    LAMB(0.0f, 0.0f, -1.0f, 0.5f, 0.4f, 0.8f, 0.4f);
    METAL(-2.0f, 0.0f, -1.0f, 0.5f, 0.4f, 0.4f, 0.8f, 0.0f);
    METAL(2.0f, 0.0f, 1.0f, 0.5f, 0.4f, 0.8f, 0.4f, 0.0f);
    METAL(0.0f, 0.0f, 1.0f, 0.5f, 0.4f, 0.8f, 0.4f, 0.2f);
    METAL(-2.0f, 0.0f, 1.0f, 0.5f, 0.4f, 0.8f, 0.4f, 0.6f);
    hv_push(hv, mk_sphere(st, V3(0.5f, 1.0f, 0.5f), 0.5f, mat_dielectric(1.5f))); /* presets.rs:657 */
    LIGHT(-1.5f, 1.5f, 0.0f, 0.3f, 30.0f, 25.0f, 15.0f);    /* presets.rs:658-662 */
    { /* presets.rs:663-707: lambertian greys at z = -3 */
        const float xs[9] = {4.0f, 3.0f, 2.0f, 1.0f, 0.0f, -1.0f, -2.0f, -3.0f, -4.0f};
        const float g[9] = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f, 0.9f};
        for (int i = 0; i < 9; ++i) LAMB(xs[i], 0.0f, -3.0f, 0.5f, g[i], g[i], g[i]);
        /* presets.rs:708-752: metal greys at z = -4 */
        for (int i = 0; i < 9; ++i) METAL(xs[i], 0.0f, -4.0f, 0.5f, g[i], g[i], g[i], 0.0f);
        /* presets.rs:753-797: coloured metals at z = -5 */
        const float cm[9][3] = {{0.8f, 0.1f, 0.1f}, {0.8f, 0.5f, 0.1f}, {0.8f, 0.8f, 0.1f}, {0.4f, 0.8f, 0.1f},
                                {0.1f, 0.8f, 0.1f}, {0.1f, 0.8f, 0.5f}, {0.1f, 0.8f, 0.8f}, {0.1f, 0.1f, 0.8f},
                                {0.5f, 0.1f, 0.8f}};
        for (int i = 0; i < 9; ++i) METAL(xs[i], 0.0f, -5.0f, 0.5f, cm[i][0], cm[i][1], cm[i][2], 0.0f);
        /* presets.rs:798-842: coloured lambertians at z = -6, last one metal */
        for (int i = 0; i < 8; ++i) LAMB(xs[i], 0.0f, -6.0f, 0.5f, cm[i][0], cm[i][1], cm[i][2]);
        METAL(-4.0f, 0.0f, -6.0f, 0.5f, 0.5f, 0.1f, 0.8f, 0.0f);
    }
    LIGHT(1.5f, 1.5f, -2.0f, 0.3f, 3.0f, 10.0f, 20.0f);     /* presets.rs:843-847 */
#undef LAMB
#undef METAL
#undef LIGHT
    sc->has_sky = 0;
}

/* Builder-defined (no reference preset; BASELINE.json config 5): the two
 * spheres of two_perlin_spheres (presets.rs:299-312) plus a 100x100 grid of
 * r=0.2 spheres jittered with the scene rng; 80 % lambertian(noise) over four
 * noise scales, 15 % metal, 5 % dielectric. Camera pulled back to frame it. */
static void preset_perlin_spheres(ora_scene *sc, uint32_t width, uint32_t height, xoshiro *rng, hitvec *hv) {
    storage *st = &sc->st;
    sc->cam = camera_new(V3(26.0f, 6.0f, 6.0f), V3(0.0f, 0.0f, 0.0f), V3(0.0f, 1.0f, 0.0f), 30.0f,
                         (float)width / (float)height, 0.0f, 10.0f, 0.0f, 0.0f);
    const texture *noise4 = alloc_texture(st, tex_noise(&st->perlin_noise, 4.0f));
    hv_push(hv, mk_sphere(st, V3(0.0f, -1000.0f, 0.0f), 1000.0f, mat_lambertian(noise4)));
    hv_push(hv, mk_sphere(st, V3(0.0f, 2.0f, 0.0f), 2.0f, mat_lambertian(noise4)));
    const texture *nz[4];
    nz[0] = alloc_texture(st, tex_noise(&st->perlin_noise, 1.0f));
    nz[1] = alloc_texture(st, tex_noise(&st->perlin_noise, 2.0f));
    nz[2] = noise4;
    nz[3] = alloc_texture(st, tex_noise(&st->perlin_noise, 8.0f));
    for (int a = -50; a < 50; ++a) {
        for (int b = -50; b < 50; ++b) {
            float choose_material = gen_f32(rng);
            float cx = 0.5f * (float)a + 0.3f * gen_f32(rng);
            float cz = 0.5f * (float)b + 0.3f * gen_f32(rng);
            v3 centre = V3(cx, 0.2f, cz);
            if (choose_material < 0.8f) {
                int k = (int)(gen_f32(rng) * 4.0f);
                hv_push(hv, mk_sphere(st, centre, 0.2f, mat_lambertian(nz[k & 3])));
            } else if (choose_material < 0.95f) {
                float ax = 0.5f * (1.0f + gen_f32(rng));
                float ay = 0.5f * (1.0f + gen_f32(rng));
                float az = 0.5f * (1.0f + gen_f32(rng));
                float fuzz = 0.5f * gen_f32(rng);
                hv_push(hv, mk_sphere(st, centre, 0.2f, mat_metal(V3(ax, ay, az), fuzz)));
            } else {
                hv_push(hv, mk_sphere(st, centre, 0.2f, mat_dielectric(1.5f)));
            }
        }
    }
    sc->has_sky = 0;
}

/* ======================================================================== */
/* bvh.rs:64-94,268-347 BVH build (assumption A8: stable sort)              */
/* ======================================================================== */
static float hit_min_axis(const hitable *h, int axis) {
    aabb bb = hitable_bounding_box(h);
    return axis == 0 ? bb.min.x : (axis == 1 ? bb.min.y : bb.min.z);
}

static void merge_sort_hitables(hitable *v, hitable *tmp, size_t n, int axis) {
    if (n < 2) return;
    size_t h = n / 2;
    merge_sort_hitables(v, tmp, h, axis);
    merge_sort_hitables(v + h, tmp, n - h, axis);
    size_t i = 0, j = h, k = 0;
    while (i < h && j < n) {
        if (hit_min_axis(&v[j], axis) < hit_min_axis(&v[i], axis)) tmp[k++] = v[j++];
        else tmp[k++] = v[i++];
    }
    while (i < h) tmp[k++] = v[i++];
    while (j < n) tmp[k++] = v[j++];
    memcpy(v, tmp, n * sizeof(hitable));
}

/* bvh.rs:268-283 */
static void sort_by_axis(xoshiro *rng, hitable *v, hitable *tmp, size_t n) {
    int axis = gen_range_i32(rng, 0, 3);
    merge_sort_hitables(v, tmp, n, axis);
}

static hitable mk_node_hitable(bvhnode *n) { hitable h; memset(&h, 0, sizeof h); h.kind = HIT_BVHNODE; h.node = n; return h; }
static bvhnode *bvh_new_node(storage *st, xoshiro *rng, hitable *v, hitable *tmp, size_t n);

/* bvh.rs:315-333 */
static hitable bvh_new_split(storage *st, xoshiro *rng, hitable *v, hitable *tmp, size_t n) {
    sort_by_axis(rng, v, tmp, n);
    if (n == 1) return v[0];
    if (n == 2) {
        aabb bb = aabb_add(hitable_bounding_box(&v[0]), hitable_bounding_box(&v[1]));
        return mk_node_hitable(alloc_bvhnode(st, v[0], v[1], bb));
    }
    return mk_node_hitable(bvh_new_node(st, rng, v, tmp, n));
}

/* bvh.rs:298-313 */
static bvhnode *bvh_new_node(storage *st, xoshiro *rng, hitable *v, hitable *tmp, size_t n) {
    size_t pivot = n / 2;
    hitable lhs = bvh_new_split(st, rng, v, tmp, pivot);
    hitable rhs = bvh_new_split(st, rng, v + pivot, tmp, n - pivot);
    aabb bb = aabb_add(hitable_bounding_box(&lhs), hitable_bounding_box(&rhs));
    return alloc_bvhnode(st, lhs, rhs, bb);
}

/* bvh.rs:64-94 */
static bvhnode *bvh_new(storage *st, xoshiro *rng, hitable *v, size_t n) {
    if (n == 0) return NULL;
    if (n == 1) return alloc_bvhnode(st, v[0], v[0], hitable_bounding_box(&v[0]));
    if (n == 2) return alloc_bvhnode(st, v[0], v[1], aabb_add(hitable_bounding_box(&v[0]), hitable_bounding_box(&v[1])));
    hitable *tmp = malloc(n * sizeof(hitable));
    sort_by_axis(rng, v, tmp, n); /* bvh.rs:285-296 new_root */
    bvhnode *root = bvh_new_node(st, rng, v, tmp, n);
    free(tmp);
    return root;
}

/* ======================================================================== */
/* offline.rs:16-24 + params.rs:21-46                                       */
/* ======================================================================== */
static uint64_t g_draw_probe; /* unused placeholder to keep the ledger explicit */

ora_scene *ora_scene_from_preset(const char *name, uint32_t width, uint32_t height, int use_bvh) {
    int which;
    if (!strcmp(name, "random_spheres")) which = 0;
    else if (!strcmp(name, "small")) which = 1;
    else if (!strcmp(name, "two_perlin_spheres")) which = 2;
    else if (!strcmp(name, "aras")) which = 3;
    else if (!strcmp(name, "perlin_spheres")) which = 4;
    else return NULL; /* presets.rs:36 */
    (void)g_draw_probe;

    ora_scene *sc = calloc(1, sizeof(*sc));
    xoshiro rng, rng0;
    xoshiro_seed_from_u64(&rng, 0); /* params.rs:21-27 (random_seed = false) */
    rng0 = rng;
    storage_init(&sc->st, &rng, which == 4 ? 2 * 10100 : 2 * 600); /* storage.rs:28-43: 1536 draws */
    hitvec hv = {0};
    switch (which) {
    case 0: preset_random_spheres(sc, width, height, &rng, &hv); break;
    case 1: preset_small(sc, width, height, &hv); break;
    case 2: preset_two_perlin_spheres(sc, width, height, &hv); break;
    case 3: preset_aras(sc, width, height, &hv); break;
    default: preset_perlin_spheres(sc, width, height, &rng, &hv); break;
    }
    /* params.rs:29-46 new_scene */
    sc->use_bvh = use_bvh;
    sc->list.hitables = malloc(hv.len * sizeof(hitable));
    memcpy(sc->list.hitables, hv.v, hv.len * sizeof(hitable)); /* list order survives the BVH's in-place sort */
    sc->list.len = hv.len;
    if (use_bvh) {
        bvhnode *root = bvh_new(&sc->st, &rng, hv.v, hv.len);
        memset(&sc->world, 0, sizeof sc->world);
        sc->world.kind = HIT_BVHNODE; sc->world.node = root;
    } else {
        memset(&sc->world, 0, sizeof sc->world);
        sc->world.kind = HIT_LIST; sc->world.list = &sc->list;
    }
    free(hv.v);
    /* ledger: count draws by replaying the generator from the seed */
    {
        uint64_t n = 0; xoshiro probe = rng0;
        while (memcmp(probe.s, rng.s, sizeof probe.s) != 0 && n < 100000000ULL) { xoshiro_next_u64(&probe); ++n; }
        sc->build_draws = n;
    }
    return sc;
}

void ora_scene_free(ora_scene *s) {
    if (!s) return;
    storage_free(&s->st);
    free(s->list.hitables);
    free(s);
}

/* ======================================================================== */
/* scene.rs:15-16,40-121                                                    */
/* ======================================================================== */
#define MAX_T 3.40282346638528859812e+38f /* f32::MAX */
#define MIN_T 0.001f

/* scene.rs:40-47 */
static v3 scene_sky(const ora_scene *s, const ray *r) {
    if (s->has_sky) return s->sky;
    float t = 0.5f * (r->direction.y + 1.0f);
    return v3_add(v3_splat(1.0f - t), v3_scale(v3_scale(V3(0.5f, 0.7f, 1.0f), t), 0.3f));
}

/* scene.rs:49-71 */
static v3 scene_ray_trace(const ora_scene *s, const ray *ray_in, uint32_t depth, uint32_t max_depth,
                          xoshiro *rng, uint64_t *ray_count) {
    *ray_count += 1;
    ray_hit hit; const material *mat;
    if (hitable_ray_hit(&s->world, ray_in, MIN_T, MAX_T, &hit, &mat)) {
        v3 emitted = material_emitted(mat, hit.u, hit.v, hit.point);
        if (depth < max_depth) {
            v3 attenuation; ray scattered;
            if (material_scatter(mat, ray_in, &hit, rng, &attenuation, &scattered)) {
                v3 rec = scene_ray_trace(s, &scattered, depth + 1, max_depth, rng, ray_count);
                return v3_add(emitted, v3_mul(attenuation, rec));
            }
        }
        return emitted;
    }
    return scene_sky(s, ray_in);
}

uint64_t ora_pixel_seed(uint32_t x, uint32_t y, uint32_t frame_num) { /* scene.rs:99-101 */
    return ((uint64_t)x * 1973u + (uint64_t)y * 9277u + (uint64_t)frame_num * 26699u) | 1u;
}

/* scene.rs:94-118 body of the per-pixel closure */
static uint64_t render_pixel(const ora_scene *s, uint32_t width, uint32_t height, uint32_t samples,
                             uint32_t max_depth, uint32_t frame_num, uint64_t i, float *color_out) {
    float inv_nx = 1.0f / (float)width;
    float inv_ny = 1.0f / (float)height;
    float inv_ns = 1.0f / (float)samples;
    float mix_prev = (float)frame_num / (float)(frame_num + 1);
    float mix_new = 1.0f - mix_prev;
    (void)height;
    uint32_t y = (uint32_t)i / width;
    uint32_t x = (uint32_t)i - (y * width);
    xoshiro rng;
    xoshiro_seed_from_u64(&rng, ora_pixel_seed(x, y, frame_num));
    uint64_t ray_count = 0;
    v3 col = V3(0.0f, 0.0f, 0.0f);
    for (uint32_t sidx = 0; sidx < samples; ++sidx) {
        float u = ((float)x + gen_f32(&rng)) * inv_nx;
        float v = ((float)y + gen_f32(&rng)) * inv_ny;
        ray r = camera_get_ray(&s->cam, u, v, &rng);
        col = v3_add(col, scene_ray_trace(s, &r, 0, max_depth, &rng, &ray_count));
    }
    col = v3_scale(col, inv_ns);
    color_out[0] = color_out[0] * mix_prev + col.x * mix_new;
    color_out[1] = color_out[1] * mix_prev + col.y * mix_new;
    color_out[2] = color_out[2] * mix_prev + col.z * mix_new;
    return ray_count;
}

typedef struct {
    const ora_scene *s; uint32_t width, height, samples, max_depth, frame_num;
    float *buffer; const uint32_t *pixels; uint64_t begin, end;
    atomic_ullong next; atomic_ullong ray_count;
} job;

#define JOB_CHUNK 64
static void *worker(void *arg) {
    job *j = (job *)arg;
    uint64_t local = 0;
    for (;;) {
        uint64_t b = atomic_fetch_add(&j->next, JOB_CHUNK);
        if (b >= j->end) break;
        uint64_t e = b + JOB_CHUNK; if (e > j->end) e = j->end;
        for (uint64_t k = b; k < e; ++k) {
            uint64_t i = j->pixels ? j->pixels[k] : k;
            local += render_pixel(j->s, j->width, j->height, j->samples, j->max_depth, j->frame_num, i,
                                  j->buffer + 3 * i);
        }
    }
    atomic_fetch_add(&j->ray_count, local); /* scene.rs:118 */
    return NULL;
}

static uint64_t run_job(job *j, int nthreads) {
    if (nthreads <= 0) { long n = sysconf(_SC_NPROCESSORS_ONLN); nthreads = n > 0 ? (int)n : 1; }
    if (nthreads > 256) nthreads = 256;
    atomic_store(&j->ray_count, 0);
    if (nthreads == 1) { worker(j); return atomic_load(&j->ray_count); }
    pthread_t th[256];
    for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, worker, j);
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    return atomic_load(&j->ray_count);
}

uint64_t ora_scene_update_range(const ora_scene *s, uint32_t width, uint32_t height, uint32_t samples,
                                uint32_t max_depth, uint32_t frame_num, float *buffer,
                                uint64_t pix_begin, uint64_t pix_end, int nthreads) {
    job j; memset(&j, 0, sizeof j);
    j.s = s; j.width = width; j.height = height; j.samples = samples; j.max_depth = max_depth;
    j.frame_num = frame_num; j.buffer = buffer; j.pixels = NULL; j.begin = pix_begin; j.end = pix_end;
    atomic_store(&j.next, pix_begin);
    return run_job(&j, nthreads);
}

uint64_t ora_scene_update(const ora_scene *s, uint32_t width, uint32_t height, uint32_t samples,
                          uint32_t max_depth, uint32_t frame_num, float *buffer, int nthreads) {
    return ora_scene_update_range(s, width, height, samples, max_depth, frame_num, buffer, 0,
                                  (uint64_t)width * height, nthreads);
}

uint64_t ora_scene_update_pixels(const ora_scene *s, uint32_t width, uint32_t height, uint32_t samples,
                                 uint32_t max_depth, uint32_t frame_num, float *buffer,
                                 const uint32_t *pixels, uint64_t n_pixels, int nthreads) {
    job j; memset(&j, 0, sizeof j);
    j.s = s; j.width = width; j.height = height; j.samples = samples; j.max_depth = max_depth;
    j.frame_num = frame_num; j.buffer = buffer; j.pixels = pixels; j.begin = 0; j.end = n_pixels;
    atomic_store(&j.next, 0);
    return run_job(&j, nthreads);
}

/* ======================================================================== */
/* flat export                                                              */
/* ======================================================================== */
uint32_t ora_scene_num_spheres(const ora_scene *s) { return (uint32_t)s->list.len; }
uint32_t ora_scene_num_materials(const ora_scene *s) { return (uint32_t)s->st.n_materials; }
uint32_t ora_scene_num_textures(const ora_scene *s) { return (uint32_t)s->st.n_textures; }
uint32_t ora_scene_num_bvh_nodes(const ora_scene *s) { return (uint32_t)s->st.n_nodes; }
uint64_t ora_scene_build_draws(const ora_scene *s) { return s->build_draws; }
int32_t ora_scene_bvh_root(const ora_scene *s) {
    if (s->world.kind != HIT_BVHNODE) return -1;
    return (int32_t)(s->world.node - s->st.nodes);
}
int ora_scene_has_perlin_texture(const ora_scene *s) {
    for (size_t i = 0; i < s->st.n_textures; ++i) if (s->st.textures[i].kind == TEX_NOISE) return 1;
    return 0;
}

void ora_scene_export_spheres(const ora_scene *s, float *xyzr, uint32_t *material_id) {
    for (size_t i = 0; i < s->list.len; ++i) {
        const hitable *h = &s->list.hitables[i];
        xyzr[4 * i + 0] = h->sph->centre.x; xyzr[4 * i + 1] = h->sph->centre.y;
        xyzr[4 * i + 2] = h->sph->centre.z; xyzr[4 * i + 3] = h->sph->radius;
        material_id[i] = (uint32_t)(h->mat - s->st.materials);
    }
}

void ora_scene_export_materials(const ora_scene *s, float *rows6) {
    for (size_t i = 0; i < s->st.n_materials; ++i) {
        const material *m = &s->st.materials[i];
        float *r = rows6 + 6 * i;
        r[0] = (float)m->kind; r[1] = m->albedo.x; r[2] = m->albedo.y; r[3] = m->albedo.z;
        r[4] = m->kind == MAT_METAL ? m->fuzz : (m->kind == MAT_DIELECTRIC ? m->ref_idx : 0.0f);
        r[5] = m->tex ? (float)(m->tex - s->st.textures) : -1.0f;
    }
}

void ora_scene_export_textures(const ora_scene *s, float *rows7) {
    for (size_t i = 0; i < s->st.n_textures; ++i) {
        const texture *t = &s->st.textures[i];
        float *r = rows7 + 7 * i;
        r[0] = (float)t->kind; r[1] = t->color.x; r[2] = t->color.y; r[3] = t->color.z;
        r[4] = t->odd ? (float)(t->odd - s->st.textures) : -1.0f;
        r[5] = t->even ? (float)(t->even - s->st.textures) : -1.0f;
        r[6] = t->scale;
    }
}

void ora_scene_export_perlin(const ora_scene *s, float *randvec, uint32_t *perm_x, uint32_t *perm_y, uint32_t *perm_z) {
    const perlin *p = &s->st.perlin_noise;
    for (int i = 0; i < 256; ++i) {
        randvec[3 * i] = p->randvec[i].x; randvec[3 * i + 1] = p->randvec[i].y; randvec[3 * i + 2] = p->randvec[i].z;
        perm_x[i] = p->perm_x[i]; perm_y[i] = p->perm_y[i]; perm_z[i] = p->perm_z[i];
    }
}

static int32_t export_child(const ora_scene *s, const hitable *h) {
    if (h->kind == HIT_BVHNODE) return (int32_t)(h->node - s->st.nodes);
    /* sphere: index in LIST order (spheres arena order == list order) */
    return ~(int32_t)(h->sph - s->st.spheres);
}

void ora_scene_export_bvh(const ora_scene *s, float *minmax6, int32_t *lhs_rhs2) {
    for (size_t i = 0; i < s->st.n_nodes; ++i) {
        const bvhnode *n = &s->st.nodes[i];
        minmax6[6 * i + 0] = n->bb.min.x; minmax6[6 * i + 1] = n->bb.min.y; minmax6[6 * i + 2] = n->bb.min.z;
        minmax6[6 * i + 3] = n->bb.max.x; minmax6[6 * i + 4] = n->bb.max.y; minmax6[6 * i + 5] = n->bb.max.z;
        lhs_rhs2[2 * i] = export_child(s, &n->lhs);
        lhs_rhs2[2 * i + 1] = export_child(s, &n->rhs);
    }
}

static void cam_to_floats(const camera *c, float *f) {
    const v3 *vs[7] = {&c->origin, &c->lower_left_corner, &c->horizontal, &c->vertical, &c->u, &c->v, &c->w};
    for (int i = 0; i < 7; ++i) { f[3 * i] = vs[i]->x; f[3 * i + 1] = vs[i]->y; f[3 * i + 2] = vs[i]->z; }
    f[21] = c->time0; f[22] = c->time1; f[23] = c->lens_radius;
}
static camera cam_from_floats(const float *f) {
    camera c;
    v3 *vs[7] = {&c.origin, &c.lower_left_corner, &c.horizontal, &c.vertical, &c.u, &c.v, &c.w};
    for (int i = 0; i < 7; ++i) *vs[i] = V3(f[3 * i], f[3 * i + 1], f[3 * i + 2]);
    c.time0 = f[21]; c.time1 = f[22]; c.lens_radius = f[23];
    return c;
}
void ora_scene_export_camera(const ora_scene *s, float *cam24) { cam_to_floats(&s->cam, cam24); }
int ora_scene_export_sky(const ora_scene *s, float *rgb3) {
    if (s->has_sky) { rgb3[0] = s->sky.x; rgb3[1] = s->sky.y; rgb3[2] = s->sky.z; }
    return s->has_sky;
}

/* ======================================================================== */
/* unit-level probes                                                        */
/* ======================================================================== */
void ora_splitmix64(uint64_t seed, uint64_t *out, int n) { uint64_t x = seed; for (int i = 0; i < n; ++i) out[i] = splitmix64_next(&x); }
void ora_xoshiro_seed_from_u64(uint64_t seed, uint64_t state[4]) { xoshiro r; xoshiro_seed_from_u64(&r, seed); memcpy(state, r.s, 32); }
uint64_t ora_xoshiro_next_u64(uint64_t state[4]) { xoshiro r; memcpy(r.s, state, 32); uint64_t v = xoshiro_next_u64(&r); memcpy(state, r.s, 32); return v; }
float ora_xoshiro_gen_f32(uint64_t state[4]) { xoshiro r; memcpy(r.s, state, 32); float v = gen_f32(&r); memcpy(state, r.s, 32); return v; }
int32_t ora_xoshiro_gen_range_i32(uint64_t state[4], int32_t low, int32_t high) {
    xoshiro r; memcpy(r.s, state, 32); int32_t v = gen_range_i32(&r, low, high); memcpy(state, r.s, 32); return v;
}
void ora_sinf_cosf(float x, float *s, float *c) { sinf_cosf(x, s, c); }

int ora_sphere_ray_hit(const float cr[4], const float o[3], const float d[3], float t_min, float t_max, float out9[9]) {
    sphere s = { V3(cr[0], cr[1], cr[2]), cr[3] };
    ray r = ray_new(V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), 0.0f);
    ray_hit h;
    if (!sphere_ray_hit(&s, &r, t_min, t_max, &h)) return 0;
    out9[0] = h.point.x; out9[1] = h.point.y; out9[2] = h.point.z;
    out9[3] = h.normal.x; out9[4] = h.normal.y; out9[5] = h.normal.z;
    out9[6] = h.t; out9[7] = h.u; out9[8] = h.v;
    return 1;
}

int ora_aabb_ray_hit(const float mn[3], const float mx[3], const float o[3], const float d[3], float t_min, float t_max) {
    aabb bb = { V3(mn[0], mn[1], mn[2]), V3(mx[0], mx[1], mx[2]) };
    ray r = ray_new(V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), 0.0f);
    return aabb_ray_hit(&bb, &r, t_min, t_max);
}

float ora_schlick(float cosine, float ref_idx) { return schlick(cosine, ref_idx); }

#define RNG_WRAP(fn) xoshiro r; memcpy(r.s, state, 32); v3 p = fn(&r); memcpy(state, r.s, 32); out3[0] = p.x; out3[1] = p.y; out3[2] = p.z
void ora_random_unit_vector(uint64_t state[4], float out3[3]) { RNG_WRAP(random_unit_vector); }
void ora_random_in_unit_sphere(uint64_t state[4], float out3[3]) { RNG_WRAP(random_in_unit_sphere); }
void ora_random_in_unit_disk(uint64_t state[4], float out3[3]) { RNG_WRAP(random_in_unit_disk); }

void ora_camera_get_ray(const float cam24[24], float s, float t, uint64_t state[4], float out7[7]) {
    camera c = cam_from_floats(cam24);
    xoshiro r; memcpy(r.s, state, 32);
    ray ry = camera_get_ray(&c, s, t, &r);
    memcpy(state, r.s, 32);
    out7[0] = ry.origin.x; out7[1] = ry.origin.y; out7[2] = ry.origin.z;
    out7[3] = ry.direction.x; out7[4] = ry.direction.y; out7[5] = ry.direction.z; out7[6] = ry.time;
}

void ora_camera_new(const float lookfrom[3], const float lookat[3], const float vup[3], float vfov, float aspect,
                    float aperture, float focus_dist, float time0, float time1, float cam24[24]) {
    camera c = camera_new(V3(lookfrom[0], lookfrom[1], lookfrom[2]), V3(lookat[0], lookat[1], lookat[2]),
                          V3(vup[0], vup[1], vup[2]), vfov, aspect, aperture, focus_dist, time0, time1);
    cam_to_floats(&c, cam24);
}

float ora_perlin_noise(const ora_scene *s, const float p[3]) { return perlin_noise(&s->st.perlin_noise, V3(p[0], p[1], p[2])); }
float ora_perlin_turb(const ora_scene *s, const float p[3]) { return perlin_turb(&s->st.perlin_noise, V3(p[0], p[1], p[2])); }

void ora_texture_value(const ora_scene *s, uint32_t texture_id, const float p[3], float rgb[3]) {
    v3 c = texture_value(&s->st.textures[texture_id], 0.0f, 0.0f, V3(p[0], p[1], p[2]));
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

void ora_ray_trace(const ora_scene *s, const float o[3], const float d[3], float time, uint32_t max_depth,
                   uint64_t state[4], float rgb[3], uint64_t *ray_count) {
    xoshiro r; memcpy(r.s, state, 32);
    ray ry = ray_new(V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), time);
    v3 c = scene_ray_trace(s, &ry, 0, max_depth, &r, ray_count);
    memcpy(state, r.s, 32);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

/* math.rs:36-48 */
static inline float fmax0(float a) { return a > 0.0f ? a : 0.0f; } /* f32::max(0.0): NaN -> 0.0 */
static inline uint8_t f32_as_u8(float f) { if (!(f > 0.0f)) return 0; if (f >= 255.0f) return 255; return (uint8_t)f; }
void ora_linear_to_srgb(const float rgb[3], uint8_t out[3]) {
    for (int i = 0; i < 3; ++i) {
        float c = fmax0(rgb[i]);
        float s = 1.055f * powf(c, 0.41666666f) - 0.055f;
        s = s > 0.0f ? s : 0.0f;   /* .max(0.0) */
        s = s < 1.0f ? s : 1.0f;   /* .min(1.0) */
        out[i] = f32_as_u8(s * 255.99f);
    }
}

/* offline.rs:43-51 */
void ora_frame_to_srgb8(const float *buffer, uint32_t width, uint32_t height, uint8_t *out) {
    size_t k = 0;
    for (uint32_t row = height; row-- > 0;) {
        for (uint32_t x = 0; x < width; ++x) {
            ora_linear_to_srgb(buffer + 3 * ((size_t)row * width + x), out + k);
            k += 3;
        }
    }
}
