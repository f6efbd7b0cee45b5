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
 *  A9 glam 0.20.5 Affine3A (presets.rs:383-390, instance.rs, ray.rs:28-40,52-64):
 *       transform_vector3(v) = ((x_axis*v.x) + (y_axis*v.y)) + (z_axis*v.z);
 *       transform_point3 = that + translation; from_rotation_translation =
 *       { Mat3A::from_quat(q), t }; Quat::from_rotation_y(a) = (0, sin(a/2), 0, cos(a/2));
 *       from_quat: x2=x+x.. xx=x*x2.. wx=w*x2.. cols (1-(yy+zz), xy+wz, xz-wy),
 *       (xy-wz, 1-(xx+zz), yz+wx), (xz+wy, yz-wx, 1-(xx+yy)); inverse = Mat3A::inverse
 *       (columns y^z, z^x, x^y scaled by 1/det, det = z.(x^y), transposed) and
 *       translation = -(inv * t). f32::to_radians(d) = d * (PI / 180).
 */
#include "ptref.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define MAX_T_F 3.40282346638528859812e+38f /* f32::MAX */

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
enum { TEX_CONSTANT = 0, TEX_CHECKER = 1, TEX_NOISE = 2, TEX_IMAGE = 3 };
/* texture.rs:5-10 */
typedef struct { uint32_t width, height; const uint8_t *data; } rgb_image;
typedef struct texture {
    int kind;
    v3 color;
    const struct texture *odd, *even;
    const perlin *noise;
    float scale;
    const rgb_image *image;
} texture;

/* Rust `f32 as i32`: saturating, NaN -> 0 */
static inline int32_t f32_as_i32(float f) {
    if (!(f == f)) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int32_t)f;
}
/* texture.rs:27-37 */
static v3 rgb_image_value(const rgb_image *im, float u, float v) {
    int32_t i = f32_as_i32(u * (float)im->width);
    int32_t j = f32_as_i32((1.0f - v) * (float)im->height - 0.001f);
    int32_t wi = (int32_t)im->width - 1, hi = (int32_t)im->height - 1;
    i = i > 0 ? i : 0; i = i < wi ? i : wi;   /* i.max(0).min(width - 1) */
    j = j > 0 ? j : 0; j = j < hi ? j : hi;
    size_t at = 3 * (size_t)i + 3 * (size_t)im->width * (size_t)j;
    return V3((float)im->data[at] / 255.0f, (float)im->data[at + 1] / 255.0f, (float)im->data[at + 2] / 255.0f);
}

/* texture.rs:74-91 */
static v3 texture_value(const texture *t, float u, float v, v3 p) {
    switch (t->kind) {
    case TEX_IMAGE:
        return rgb_image_value(t->image, u, v);
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
enum { MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_DIELECTRIC = 2, MAT_DIFFUSE_LIGHT = 3, MAT_ISOTROPIC = 4 };
typedef struct {
    int kind;
    const texture *tex; /* Lambertian albedo / DiffuseLight emit / Isotropic albedo */
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

/* material.rs:126-136: the scattered direction is NOT normalised */
static int scatter_isotropic(const texture *albedo, const ray *ray_in, const ray_hit *hit,
                             xoshiro *rng, v3 *attenuation, ray *scattered) {
    *attenuation = texture_value(albedo, hit->u, hit->v, hit->point);
    *scattered = ray_new(hit->point, random_in_unit_sphere(rng), ray_in->time);
    return 1;
}

/* material.rs:138-159 */
static int material_scatter(const material *m, const ray *ray_in, const ray_hit *hit,
                            xoshiro *rng, v3 *attenuation, ray *scattered) {
    switch (m->kind) {
    case MAT_LAMBERTIAN: return scatter_lambertian(m->tex, ray_in, hit, rng, attenuation, scattered);
    case MAT_METAL: return scatter_metal(m->albedo, m->fuzz, ray_in, hit, rng, attenuation, scattered);
    case MAT_DIELECTRIC: return scatter_dielectric(m->ref_idx, ray_in, hit, rng, attenuation, scattered);
    case MAT_ISOTROPIC: return scatter_isotropic(m->tex, ray_in, hit, rng, attenuation, scattered);
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
static inline __attribute__((always_inline)) int sphere_ray_hit(const sphere *s, const ray *r, float t_min, float t_max, ray_hit *out) {
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
/* collision/moving_sphere.rs:8-90                                          */
/* ======================================================================== */
typedef struct { v3 centre_start, centre_delta; float radius, time_start, inv_time_delta; } moving_sphere;

/* moving_sphere.rs:18-26 */
static moving_sphere moving_sphere_new(v3 centre0, v3 centre1, float time0, float time1, float radius) {
    moving_sphere m;
    m.centre_start = centre0; m.centre_delta = v3_sub(centre1, centre0); m.radius = radius;
    m.time_start = time0; m.inv_time_delta = 1.0f / (time1 - time0);
    return m;
}
/* moving_sphere.rs:29-31 */
static inline v3 moving_sphere_centre(const moving_sphere *m, float time) {
    return v3_add(m->centre_start, v3_scale(m->centre_delta, (time - m->time_start) * m->inv_time_delta));
}
/* moving_sphere.rs:38-73 */
static int moving_sphere_ray_hit(const moving_sphere *s, const ray *r, float t_min, float t_max, ray_hit *out) {
    v3 centre = moving_sphere_centre(s, r->time);
    v3 oc = v3_sub(r->origin, centre);
    float a = v3_dot(r->direction, r->direction);
    float b = v3_dot(oc, r->direction);
    float c = v3_dot(oc, oc) - s->radius * s->radius;
    float discriminant = b * b - a * c;
    if (discriminant > 0.0f) {
        float discriminant_sqrt = sqrtf(discriminant);
        float t = (-b - discriminant_sqrt) / a;
        if (t < t_max && t > t_min) {
            out->point = point_at_parameter(r, t);
            out->normal = v3_divs(v3_sub(out->point, centre), s->radius);
            out->t = t; out->u = 0.0f; out->v = 0.0f;
            return 1;
        }
        t = (-b + discriminant_sqrt) / a;
        if (t < t_max && t > t_min) {
            out->point = point_at_parameter(r, t);
            out->normal = v3_divs(v3_sub(out->point, centre), s->radius);
            out->t = t; out->u = 0.0f; out->v = 0.0f;
            return 1;
        }
    }
    return 0;
}
/* moving_sphere.rs:76-89 */
static aabb moving_sphere_bounding_box(const moving_sphere *s, float t0, float t1) {
    v3 c0 = moving_sphere_centre(s, t0), c1 = moving_sphere_centre(s, t1), rad = v3_splat(s->radius);
    aabb b0 = { v3_sub(c0, rad), v3_add(c0, rad) }, b1 = { v3_sub(c1, rad), v3_add(c1, rad) };
    return aabb_add(b0, b1);
}

/* ======================================================================== */
/* collision/rect.rs:5-230 -- one struct, `axis` selects XY / XZ / YZ; (a, b)
 * are the two in-plane coordinates in the variant's order, k the plane       */
/* ======================================================================== */
enum { RECT_XY = 0, RECT_XZ = 1, RECT_YZ = 2 };
typedef struct { int axis; float a0, a1, b0, b1, k; int flip_normals; } rect;
static const float FLIP_SIGN[2] = { 1.0f, -1.0f }; /* rect.rs:33 */

static rect rect_new(int axis, float a0, float a1, float b0, float b1, float k, int flip) {
    rect r = { axis, a0, a1, b0, b1, k, flip }; return r;
}

/* rect.rs:73-190: the comparisons are kept in the reference's form (a NaN t or
 * coordinate falls through every `<`/`>` test exactly as it does there) */
static int rect_ray_hit(const rect *q, const ray *r, float t_min, float t_max, ray_hit *out) {
    float t, a, b;
    switch (q->axis) {
    case RECT_XY: /* rect.rs:73-100 */
        t = (q->k - r->origin.z) * r->rcp_direction.z;
        if (t < t_min || t > t_max) return 0;
        a = r->origin.x + t * r->direction.x;
        b = r->origin.y + t * r->direction.y;
        if (a < q->a0 || a > q->a1 || b < q->b0 || b > q->b1) return 0;
        out->normal = V3(0.0f, 0.0f, FLIP_SIGN[q->flip_normals]);
        break;
    case RECT_XZ: /* rect.rs:102-130 */
        t = (q->k - r->origin.y) * r->rcp_direction.y;
        if (t < t_min || t > t_max) return 0;
        a = r->origin.x + t * r->direction.x;
        b = r->origin.z + t * r->direction.z;
        if (a < q->a0 || a > q->a1 || b < q->b0 || b > q->b1) return 0;
        out->normal = V3(0.0f, FLIP_SIGN[q->flip_normals], 0.0f);
        break;
    default: /* rect.rs:132-160 */
        t = (q->k - r->origin.x) * r->rcp_direction.x;
        if (t < t_min || t > t_max) return 0;
        a = r->origin.y + t * r->direction.y;
        b = r->origin.z + t * r->direction.z;
        if (a < q->a0 || a > q->a1 || b < q->b0 || b > q->b1) return 0;
        out->normal = V3(FLIP_SIGN[q->flip_normals], 0.0f, 0.0f);
        break;
    }
    out->point = point_at_parameter(r, t);
    out->t = t;
    out->u = (a - q->a0) / (q->a1 - q->a0);
    out->v = (b - q->b0) / (q->b1 - q->b0);
    return 1;
}

/* rect.rs:193-229 (the YZ arm really has `k - 0.0001` in BOTH corners, rect.rs:225-226) */
static aabb rect_bounding_box(const rect *q) {
    aabb r;
    switch (q->axis) {
    case RECT_XY: r.min = V3(q->a0, q->b0, q->k - 0.0001f); r.max = V3(q->a1, q->b1, q->k + 0.0001f); break;
    case RECT_XZ: r.min = V3(q->a0, q->k - 0.0001f, q->b0); r.max = V3(q->a1, q->k + 0.0001f, q->b1); break;
    default:      r.min = V3(q->k - 0.0001f, q->a0, q->b0); r.max = V3(q->k - 0.0001f, q->a1, q->b1); break;
    }
    return r;
}

/* ======================================================================== */
/* collision/cuboid.rs:4-42                                                 */
/* ======================================================================== */
typedef struct { rect faces[6]; aabb bb; } cuboid;

/* cuboid.rs:11-23 */
static cuboid cuboid_new(v3 p0, v3 p1) {
    cuboid c;
    c.faces[0] = rect_new(RECT_XY, p0.x, p1.x, p0.y, p1.y, p1.z, 0);
    c.faces[1] = rect_new(RECT_XY, p0.x, p1.x, p0.y, p1.y, p0.z, 1);
    c.faces[2] = rect_new(RECT_XZ, p0.x, p1.x, p0.z, p1.z, p1.y, 0);
    c.faces[3] = rect_new(RECT_XZ, p0.x, p1.x, p0.z, p1.z, p0.y, 1);
    c.faces[4] = rect_new(RECT_YZ, p0.y, p1.y, p0.z, p1.z, p1.x, 0);
    c.faces[5] = rect_new(RECT_YZ, p0.y, p1.y, p0.z, p1.z, p0.x, 1);
    c.bb.min = p0; c.bb.max = p1;
    return c;
}
/* cuboid.rs:25-37 */
static int cuboid_ray_hit(const cuboid *c, const ray *r, float t_min, float t_max, ray_hit *out) {
    int found = 0;
    if (aabb_ray_hit(&c->bb, r, t_min, t_max)) {
        float closest_so_far = t_max;
        for (int i = 0; i < 6; ++i) {
            ray_hit h;
            if (rect_ray_hit(&c->faces[i], r, t_min, closest_so_far, &h)) { *out = h; found = 1; closest_so_far = h.t; }
        }
    }
    return found;
}

/* ======================================================================== */
/* glam 0.20.5 Affine3A / Quat / Mat3A as used by presets.rs:383-390 and
 * instance.rs / ray.rs:28-40,52-64 / aabb.rs:75-100 (assumption A9)        */
/* ======================================================================== */
typedef struct { v3 x_axis, y_axis, z_axis, translation; } affine3;

/* Affine3A::transform_vector3: ((x_axis * v.x) + (y_axis * v.y)) + (z_axis * v.z) */
static inline v3 affine_transform_vector3(const affine3 *m, v3 v) {
    return v3_add(v3_add(v3_scale(m->x_axis, v.x), v3_scale(m->y_axis, v.y)), v3_scale(m->z_axis, v.z));
}
/* Affine3A::transform_point3: the same sum, then + translation */
static inline v3 affine_transform_point3(const affine3 *m, v3 p) {
    return v3_add(affine_transform_vector3(m, p), m->translation);
}
/* Affine3A::from_rotation_translation(Quat::from_rotation_y(angle), t):
 * Quat::from_rotation_y = (0, sin(a/2), 0, cos(a/2)); Mat3A::from_quat */
static affine3 affine_from_rotation_y_translation(float angle, v3 translation) {
    float qs = sinf(angle * 0.5f), qc = cosf(angle * 0.5f);
    float qx = 0.0f, qy = qs, qz = 0.0f, qw = qc;
    float x2 = qx + qx, y2 = qy + qy, z2 = qz + qz;
    float xx = qx * x2, xy = qx * y2, xz = qx * z2, yy = qy * y2, yz = qy * z2, zz = qz * z2;
    float wx = qw * x2, wy = qw * y2, wz = qw * z2;
    affine3 m;
    m.x_axis = V3(1.0f - (yy + zz), xy + wz, xz - wy);
    m.y_axis = V3(xy - wz, 1.0f - (xx + zz), yz + wx);
    m.z_axis = V3(xz + wy, yz - wx, 1.0f - (xx + yy));
    m.translation = translation;
    return m;
}
/* Affine3A::inverse: Mat3A::inverse (cross products / determinant, transposed), translation = -(inv * t) */
static affine3 affine_inverse(const affine3 *m) {
    v3 tmp0 = v3_cross(m->y_axis, m->z_axis);
    v3 tmp1 = v3_cross(m->z_axis, m->x_axis);
    v3 tmp2 = v3_cross(m->x_axis, m->y_axis);
    float det = v3_dot(m->z_axis, tmp2);
    float inv_det = 1.0f / det;
    v3 c0 = v3_scale(tmp0, inv_det), c1 = v3_scale(tmp1, inv_det), c2 = v3_scale(tmp2, inv_det);
    affine3 r;
    r.x_axis = V3(c0.x, c1.x, c2.x); /* transpose */
    r.y_axis = V3(c0.y, c1.y, c2.y);
    r.z_axis = V3(c0.z, c1.z, c2.z);
    r.translation = V3(0.0f, 0.0f, 0.0f);
    r.translation = v3_neg(affine_transform_vector3(&r, m->translation));
    return r;
}
/* ray.rs:28-40 */
static inline ray ray_transform(const ray *r, const affine3 *m) {
    ray o;
    o.origin = affine_transform_point3(m, r->origin);
    o.direction = affine_transform_vector3(m, r->direction);
    o.rcp_direction = v3_recip(o.direction);
    o.time = r->time;
    return o;
}
/* ray.rs:52-64 */
static inline ray_hit ray_hit_transform(const ray_hit *h, const affine3 *m) {
    ray_hit o = *h;
    o.point = affine_transform_point3(m, h->point);
    o.normal = affine_transform_vector3(m, h->normal);
    return o;
}
/* aabb.rs:75-100: starts from `min = max = m.w_axis` and never reads `self`, so the
 * result is the single point t + x_axis*t + y_axis*t + z_axis*t (component-wise
 * products). Reproduced as written: it decides what a BVH over instances can hit. */
static aabb aabb_transform_as_written(const affine3 *m) {
    v3 t = m->translation;
    v3 out = t;
    out = v3_add(out, v3_mul(m->x_axis, t));
    out = v3_add(out, v3_mul(m->y_axis, t));
    out = v3_add(out, v3_mul(m->z_axis, t));
    aabb r = { out, out };
    return r;
}

/* ======================================================================== */
/* collision/hitable.rs:12-65, hitable_list.rs:40-56, bvh.rs:24-62,
 * instance.rs:9-47, constant_medium.rs:11-77                               */
/* ======================================================================== */
enum { HIT_SPHERE = 0, HIT_BVHNODE = 1, HIT_LIST = 2, HIT_MOVING_SPHERE = 3, HIT_RECT = 4, HIT_CUBOID = 5,
       HIT_INSTANCE = 6, HIT_CONSTANT_MEDIUM = 7 };
struct bvhnode; struct hitable_list; struct instance; struct constant_medium;
typedef struct hitable { /* 24 bytes like the Rust enum: tag + two references */
    int kind;
    union {
        const sphere *sph;                  /* Sphere(&Sphere, &Material) */
        const moving_sphere *msph;          /* MovingSphere(&MovingSphere, &Material) */
        const rect *rct;                    /* Rect(&Rect, &Material) */
        const cuboid *cub;                  /* Cuboid(&Cuboid, &Material) */
        const struct bvhnode *node;         /* BVHNode(&BVHNode) */
        const struct hitable_list *list;    /* List(&HitableList) */
        const struct instance *inst;        /* Instance(&Instance) */
        const struct constant_medium *med;  /* ConstantMedium(&ConstantMedium) */
    };
    const material *mat;
} hitable;
typedef struct bvhnode { aabb bb; hitable lhs, rhs; } bvhnode;
typedef struct hitable_list { hitable *hitables; size_t len; } hitable_list;
typedef struct instance { hitable child; affine3 transform, inv_transform; } instance;
typedef struct constant_medium { hitable child; material phase_function; float density; } constant_medium;

static int hitable_ray_hit(const hitable *h, const ray *r, float t_min, float t_max, xoshiro *rng,
                           ray_hit *out, const material **mat);

/* hitable_list.rs:40-56 */
static int list_ray_hit(const hitable_list *l, const ray *r, float t_min, float t_max, xoshiro *rng,
                        ray_hit *out, const material **mat) {
    int found = 0;
    float closest_so_far = t_max;
    for (size_t i = 0; i < l->len; ++i) {
        ray_hit h; const material *m;
        const hitable *e = &l->hitables[i];
        int hit;
        if (e->kind == HIT_SPHERE) { hit = sphere_ray_hit(e->sph, r, t_min, closest_so_far, &h); m = e->mat; } /* Sphere arm, inlined */
        else hit = hitable_ray_hit(e, r, t_min, closest_so_far, rng, &h, &m);
        if (hit) {
            *out = h; *mat = m; found = 1;
            closest_so_far = h.t;
        }
    }
    return found;
}

/* optional instrumentation: BVHNode::ray_hit calls and leaf (non-node) hitables tested below them */
static __thread uint64_t tl_bvh_nodes, tl_bvh_leaves;
static atomic_ullong g_bvh_nodes, g_bvh_leaves;

/* bvh.rs:37-62 */
static int bvh_ray_hit(const bvhnode *n, const ray *r, float t_min, float t_max, xoshiro *rng,
                       ray_hit *out, const material **mat) {
    tl_bvh_nodes += 1;
    if (aabb_ray_hit(&n->bb, r, t_min, t_max)) {
        tl_bvh_leaves += (n->lhs.kind != HIT_BVHNODE) + (n->rhs.kind != HIT_BVHNODE);
        ray_hit hl, hr; const material *ml, *mr;
        int has_l = hitable_ray_hit(&n->lhs, r, t_min, t_max, rng, &hl, &ml);
        int has_r = hitable_ray_hit(&n->rhs, r, t_min, t_max, rng, &hr, &mr);
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

/* instance.rs:32-47 */
static int instance_ray_hit(const instance *in, const ray *r, float t_min, float t_max, xoshiro *rng,
                            ray_hit *out, const material **mat) {
    ray local = ray_transform(r, &in->inv_transform);
    ray_hit h;
    if (hitable_ray_hit(&in->child, &local, t_min, t_max, rng, &h, mat)) {
        *out = ray_hit_transform(&h, &in->transform);
        return 1;
    }
    return 0;
}

/* constant_medium.rs:32-77 (f32::ln = glibc logf, assumption A7) */
static int constant_medium_ray_hit(const constant_medium *cm, const ray *r, float t_min, float t_max,
                                   xoshiro *rng, ray_hit *out, const material **mat) {
    ray_hit h1, h2; const material *unused;
    if (hitable_ray_hit(&cm->child, r, -MAX_T_F, MAX_T_F, rng, &h1, &unused)) {
        if (hitable_ray_hit(&cm->child, r, h1.t + 0.0001f, MAX_T_F, rng, &h2, &unused)) {
            float t1 = h1.t, t2 = h2.t;
            if (t1 < t_min) t1 = t_min;
            if (t2 > t_max) t2 = t_max;
            if (t1 >= t2) return 0;
            if (t1 < 0.0f) t1 = 0.0f;
            float ray_length = v3_length(r->direction);
            float distance_inside_boundary = (t2 - t1) * ray_length;
            float hit_distance = -(1.0f / cm->density) * logf(gen_f32(rng));
            if (hit_distance < distance_inside_boundary) {
                float t = t1 + hit_distance / ray_length;
                out->point = point_at_parameter(r, t);
                out->normal = V3(1.0f, 0.0f, 0.0f); /* Vec3::X, arbitrary */
                out->t = t; out->u = 0.0f; out->v = 0.0f;
                *mat = &cm->phase_function;
                return 1;
            }
        }
    }
    return 0;
}

/* hitable.rs:39-65 */
static int hitable_ray_hit(const hitable *h, const ray *r, float t_min, float t_max, xoshiro *rng,
                           ray_hit *out, const material **mat) {
    int hit;
    if (h->kind == HIT_SPHERE) { /* the common arm first; same result as the match in hitable.rs:47-57 */
        if (sphere_ray_hit(h->sph, r, t_min, t_max, out)) { *mat = h->mat; return 1; }
        return 0;
    }
    switch (h->kind) {
    case HIT_BVHNODE: return bvh_ray_hit(h->node, r, t_min, t_max, rng, out, mat);
    case HIT_LIST: return list_ray_hit(h->list, r, t_min, t_max, rng, out, mat);
    case HIT_INSTANCE: return instance_ray_hit(h->inst, r, t_min, t_max, rng, out, mat);
    case HIT_CONSTANT_MEDIUM: return constant_medium_ray_hit(h->med, r, t_min, t_max, rng, out, mat);
    case HIT_RECT: hit = rect_ray_hit(h->rct, r, t_min, t_max, out); break;
    case HIT_CUBOID: hit = cuboid_ray_hit(h->cub, r, t_min, t_max, out); break;
    case HIT_MOVING_SPHERE: hit = moving_sphere_ray_hit(h->msph, r, t_min, t_max, out); break;
    default: hit = sphere_ray_hit(h->sph, r, t_min, t_max, out); break;
    }
    if (hit) { *mat = h->mat; return 1; }
    return 0;
}

/* hitable.rs:25-36 with t0 = t1 = 0 (bvh.rs:69-70); every arm is Some for the kinds built here */
static aabb hitable_bounding_box(const hitable *h) {
    switch (h->kind) {
    case HIT_BVHNODE: return h->node->bb;
    case HIT_INSTANCE: return aabb_transform_as_written(&h->inst->transform); /* instance.rs:24-30 */
    case HIT_CONSTANT_MEDIUM: return hitable_bounding_box(&h->med->child);      /* constant_medium.rs:28-30 */
    case HIT_RECT: return rect_bounding_box(h->rct);
    case HIT_CUBOID: return h->cub->bb;
    case HIT_MOVING_SPHERE: return moving_sphere_bounding_box(h->msph, 0.0f, 0.0f);
    default: return sphere_bounding_box(h->sph);
    }
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
    moving_sphere *moving; size_t n_moving;
    rect *rects; size_t n_rects;
    cuboid *cuboids; size_t n_cuboids;
    instance *instances; size_t n_instances;
    constant_medium *media; size_t n_media;
    rgb_image *images; size_t n_images; uint8_t *image_bytes;
    perlin perlin_noise;
} storage;

struct ora_scene {
    storage st;
    hitable world;          /* scene.rs:19 */
    hitable_list list;      /* storage.rs:86 alloc_hitables */
    /* scene graphs (ora_scene_from_graph): the nested Hitables built over the leaves in `list` */
    instance *g_instances; constant_medium *g_media; hitable_list *g_lists; hitable *g_children; bvhnode *g_bvh;
    const float *gb_minmax6; const int32_t *gb_lr2; uint32_t gb_n; size_t gb_used;   /* the BVHNode rows a graph is being built from (build time only) */
    int has_sky; v3 sky;    /* scene.rs:20 */
    camera cam;
    int use_bvh;
    uint64_t build_draws;
};

/* arenas are sized up-front so pointers stay stable (typed_arena semantics) */
static void storage_init(storage *st, xoshiro *rng, size_t max_items, size_t max_shapes) {
    memset(st, 0, sizeof(*st));
    st->cap_textures = max_items + 16; st->textures = calloc(st->cap_textures, sizeof(texture));
    st->cap_materials = max_items + 16; st->materials = calloc(st->cap_materials, sizeof(material));
    st->cap_spheres = max_items + 16; st->spheres = calloc(st->cap_spheres, sizeof(sphere));
    st->cap_nodes = max_items + 16; st->nodes = calloc(st->cap_nodes, sizeof(bvhnode));
    st->moving = calloc(max_items + 16, sizeof(moving_sphere));
    st->rects = calloc(max_shapes + 64, sizeof(rect)); st->cuboids = calloc(max_shapes + 16, sizeof(cuboid));
    st->instances = calloc(max_shapes + 16, sizeof(instance)); st->media = calloc(max_shapes + 16, sizeof(constant_medium));
    perlin_new(&st->perlin_noise, rng); /* storage.rs:41 */
}
static void storage_free(storage *st) {
    free(st->textures); free(st->materials); free(st->spheres); free(st->nodes);
    free(st->moving); free(st->rects); free(st->cuboids); free(st->instances); free(st->media);
    free(st->images); free(st->image_bytes);
}

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

/* the `moving_sphere` closure of presets.rs:122-127 (times 0.0 .. 1.0) */
static hitable mk_moving_sphere(storage *st, v3 centre0, v3 centre1, float radius, material m) {
    hitable h; memset(&h, 0, sizeof h); h.kind = HIT_MOVING_SPHERE;
    st->moving[st->n_moving] = moving_sphere_new(centre0, centre1, 0.0f, 1.0f, radius);
    h.msph = &st->moving[st->n_moving++];
    h.mat = alloc_material(st, m);
    return h;
}
/* Hitable::Rect(storage.alloc_rect(..), material) */
static hitable mk_rect(storage *st, rect q, const material *m) {
    hitable h; memset(&h, 0, sizeof h); h.kind = HIT_RECT;
    st->rects[st->n_rects] = q; h.rct = &st->rects[st->n_rects++]; h.mat = m;
    return h;
}
static hitable mk_cuboid(storage *st, v3 p0, v3 p1, const material *m) {
    hitable h; memset(&h, 0, sizeof h); h.kind = HIT_CUBOID;
    st->cuboids[st->n_cuboids] = cuboid_new(p0, p1); h.cub = &st->cuboids[st->n_cuboids++]; h.mat = m;
    return h;
}
/* instance.rs:16-22 */
static hitable mk_instance(storage *st, hitable child, affine3 transform) {
    hitable h; memset(&h, 0, sizeof h); h.kind = HIT_INSTANCE;
    instance *in = &st->instances[st->n_instances++];
    in->child = child; in->transform = transform; in->inv_transform = affine_inverse(&transform);
    h.inst = in;
    return h;
}
/* constant_medium.rs:18-26; material.rs:37-39 isotropic(albedo) */
static hitable mk_constant_medium(storage *st, hitable child, float density, const texture *albedo) {
    hitable h; memset(&h, 0, sizeof h); h.kind = HIT_CONSTANT_MEDIUM;
    constant_medium *cm = &st->media[st->n_media++];
    cm->child = child; cm->density = density;
    memset(&cm->phase_function, 0, sizeof cm->phase_function);
    cm->phase_function.kind = MAT_ISOTROPIC; cm->phase_function.tex = albedo;
    h.med = cm;
    return h;
}

/* ======================================================================== */
/* presets.rs                                                               */
/* ======================================================================== */

/* presets.rs:89-215 random_impl(only_spheres = false): the `random` preset */
static void preset_random(ora_scene *sc, uint32_t width, uint32_t height, xoshiro *rng, hitvec *hv) {
    storage *st = &sc->st;
    sc->cam = camera_new(V3(13.0f, 2.0f, 3.0f), V3(0.0f, 0.0f, 0.0f), V3(0.0f, 1.0f, 0.0f), 20.0f,
                         (float)width / (float)height, 0.1f, 10.0f, 0.0f, 1.0f);
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
                v3 centre1 = v3_add(centre, V3(0.0f, 0.5f * gen_f32(rng), 0.0f)); /* presets.rs:150 */
                float r0 = gen_f32(rng), r1 = gen_f32(rng), r2 = gen_f32(rng),
                      r3 = gen_f32(rng), r4 = gen_f32(rng), r5 = gen_f32(rng);
                const texture *t = alloc_texture(st, tex_constant(V3(r0 * r1, r2 * r3, r4 * r5)));
                hv_push(hv, mk_moving_sphere(st, centre, centre1, 0.2f, mat_lambertian(t))); /* presets.rs:162-171 */
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

/* presets.rs:317-370 */
static void preset_simple_light(ora_scene *sc, uint32_t width, uint32_t height, hitvec *hv) {
    storage *st = &sc->st;
    sc->cam = camera_new(V3(50.0f, 2.0f, 3.0f), V3(0.0f, 0.0f, 0.0f), V3(0.0f, 1.0f, 0.0f), 20.0f,
                         (float)width / (float)height, 0.0f, 10.0f, 0.0f, 0.0f);
    const texture *noise_texture = alloc_texture(st, tex_noise(&st->perlin_noise, 4.0f));
    const texture *constant_texture = alloc_texture(st, tex_constant(V3(4.0f, 4.0f, 4.0f)));
    hv_push(hv, mk_sphere(st, V3(0.0f, -1000.0f, 0.0f), 1000.0f, mat_lambertian(noise_texture)));
    hv_push(hv, mk_sphere(st, V3(0.0f, 2.0f, 0.0f), 2.0f, mat_lambertian(noise_texture)));
    hv_push(hv, mk_sphere(st, V3(0.0f, 7.0f, 0.0f), 2.0f, mat_diffuse_light(constant_texture)));
    rect q = rect_new(RECT_XY, 3.0f, 5.0f, 1.0f, 3.0f, -2.0f, 0);
    hitable hr = mk_rect(st, q, NULL);
    hr.mat = alloc_material(st, mat_diffuse_light(constant_texture)); /* rect arena first, then material (presets.rs:363-366) */
    hv_push(hv, hr);
    sc->has_sky = 1; sc->sky = V3(0.0f, 0.0f, 0.0f);
}

/* presets.rs:372-456 (cornell_box) and 458-552 (cornell_smoke) */
static void preset_cornell(ora_scene *sc, uint32_t width, uint32_t height, hitvec *hv, int smoke) {
    storage *st = &sc->st;
    sc->cam = camera_new(V3(278.0f, 278.0f, -800.0f), V3(278.0f, 278.0f, 0.0f), V3(0.0f, 1.0f, 0.0f), 40.0f,
                         (float)width / (float)height, 0.0f, 10.0f, 0.0f, 1.0f);
    const texture *tr = alloc_texture(st, tex_constant(V3(0.65f, 0.05f, 0.05f)));
    const material *red = alloc_material(st, mat_lambertian(tr));
    const texture *tw = alloc_texture(st, tex_constant(V3(0.73f, 0.73f, 0.73f)));
    const material *white = alloc_material(st, mat_lambertian(tw));
    const texture *tg = alloc_texture(st, tex_constant(V3(0.12f, 0.45f, 0.15f)));
    const material *green = alloc_material(st, mat_lambertian(tg));
    const float e = smoke ? 7.0f : 15.0f;
    const texture *tl = alloc_texture(st, tex_constant(V3(e, e, e)));
    const material *light = alloc_material(st, mat_diffuse_light(tl));
    const float to_rad = PT_PI / 180.0f; /* f32::to_radians */
    affine3 box1 = affine_from_rotation_y_translation(-18.0f * to_rad, V3(130.0f, 0.0f, 65.0f));
    affine3 box2 = affine_from_rotation_y_translation(15.0f * to_rad, V3(265.0f, 0.0f, 295.0f));
    hv_push(hv, mk_rect(st, rect_new(RECT_YZ, 0.0f, 555.0f, 0.0f, 555.0f, 555.0f, 1), green));
    hv_push(hv, mk_rect(st, rect_new(RECT_YZ, 0.0f, 555.0f, 0.0f, 555.0f, 0.0f, 0), red));
    if (smoke) hv_push(hv, mk_rect(st, rect_new(RECT_XZ, 113.0f, 443.0f, 127.0f, 432.0f, 554.0f, 0), light));
    else hv_push(hv, mk_rect(st, rect_new(RECT_XZ, 213.0f, 343.0f, 227.0f, 332.0f, 554.0f, 0), light));
    hv_push(hv, mk_rect(st, rect_new(RECT_XZ, 0.0f, 555.0f, 0.0f, 555.0f, 555.0f, 1), white));
    hv_push(hv, mk_rect(st, rect_new(RECT_XZ, 0.0f, 555.0f, 0.0f, 555.0f, 0.0f, 0), white));
    hv_push(hv, mk_rect(st, rect_new(RECT_XY, 0.0f, 555.0f, 0.0f, 555.0f, 555.0f, 1), white));
    hitable b1 = mk_instance(st, mk_cuboid(st, V3(0.0f, 0.0f, 0.0f), V3(165.0f, 165.0f, 165.0f), white), box1);
    if (smoke) {
        const texture *one = alloc_texture(st, tex_constant(V3(1.0f, 1.0f, 1.0f)));
        hv_push(hv, mk_constant_medium(st, b1, 0.01f, one));
    } else hv_push(hv, b1);
    hitable b2 = mk_instance(st, mk_cuboid(st, V3(0.0f, 0.0f, 0.0f), V3(165.0f, 330.0f, 165.0f), white), box2);
    if (smoke) {
        const texture *zero = alloc_texture(st, tex_constant(V3(0.0f, 0.0f, 0.0f)));
        hv_push(hv, mk_constant_medium(st, b2, 0.01f, zero));
    } else hv_push(hv, b2);
    sc->has_sky = 1; sc->sky = V3(0.0f, 0.0f, 0.0f);
}

/* presets.rs:40-71 final_scene: camera and two textures, but the hitable list is returned EMPTY */
static void preset_final(ora_scene *sc, uint32_t width, uint32_t height, hitvec *hv) {
    storage *st = &sc->st;
    (void)hv;
    sc->cam = camera_new(V3(13.0f, 2.0f, 3.0f), V3(0.0f, 0.0f, 0.0f), V3(0.0f, 1.0f, 0.0f), 20.0f,
                         (float)width / (float)height, 0.1f, 10.0f, 0.0f, 1.0f);
    alloc_texture(st, tex_constant(V3(0.73f, 0.73f, 0.73f))); /* `white` / `ground`: Material VALUES, never put in the arena */
    alloc_texture(st, tex_constant(V3(0.48f, 0.83f, 0.53f)));
    sc->has_sky = 0;
}

/* presets.rs:853-930 */
static void preset_smallpt(ora_scene *sc, uint32_t width, uint32_t height, hitvec *hv) {
    storage *st = &sc->st;
    sc->cam = camera_new(V3(50.0f, 52.0f, 295.6f), V3(50.0f, 33.0f, 0.0f), V3(0.0f, 1.0f, 0.0f), 30.0f,
                         (float)width / (float)height, 0.05f, 100.0f, 0.0f, 1.0f);
#define LAMB(cx, cy, cz, r, ax, ay, az) do { const texture *t_ = alloc_texture(st, tex_constant(V3(ax, ay, az))); \
        hv_push(hv, mk_sphere(st, V3(cx, cy, cz), r, mat_lambertian(t_))); } while (0)
    LAMB(1e3f + 1.0f, 40.8f, 81.6f, 1e3f, 0.75f, 0.25f, 0.25f);   /* Left */
    LAMB(-1e3f + 99.0f, 40.8f, 81.6f, 1e3f, 0.25f, 0.25f, 0.75f); /* Rght */
    LAMB(50.0f, 40.8f, 1e3f, 1e3f, 0.75f, 0.75f, 0.75f);          /* Back */
    LAMB(50.0f, 1e3f, 81.6f, 1e3f, 0.75f, 0.75f, 0.75f);          /* Botm */
    LAMB(50.0f, -1e3f + 81.6f, 81.6f, 1e3f, 0.75f, 0.75f, 0.75f); /* Top */
#undef LAMB
    hv_push(hv, mk_sphere(st, V3(27.0f, 16.5f, 47.0f), 16.5f,
                          mat_metal(v3_scale(V3(1.0f, 1.0f, 1.0f), 0.999f), 0.0f)));      /* Mirr */
    hv_push(hv, mk_sphere(st, V3(73.0f, 16.5f, 78.0f), 16.5f, mat_dielectric(1.5f)));       /* Glas */
    const texture *tl = alloc_texture(st, tex_constant(v3_scale(V3(4.0f, 4.0f, 4.0f), 100.0f)));
    hv_push(hv, mk_sphere(st, V3(50.0f, 81.6f - 16.5f, 81.6f), 1.5f, mat_diffuse_light(tl))); /* Lite */
    sc->has_sky = 1; sc->sky = V3(0.0f, 0.0f, 0.0f);
}

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

ora_scene *ora_scene_from_preset(const char *name, uint32_t width, uint32_t height, int use_bvh) {
    int which;
    if (!strcmp(name, "random_spheres")) which = 0;
    else if (!strcmp(name, "small")) which = 1;
    else if (!strcmp(name, "two_perlin_spheres")) which = 2;
    else if (!strcmp(name, "aras")) which = 3;
    else if (!strcmp(name, "perlin_spheres")) which = 4;
    else if (!strcmp(name, "random")) which = 5;
    else if (!strcmp(name, "simple_light")) which = 6;
    else if (!strcmp(name, "cornell")) which = 7;
    else if (!strcmp(name, "cornell_smoke")) which = 8;
    else if (!strcmp(name, "smallpt")) which = 9;
    else if (!strcmp(name, "final")) which = 10;
    else return NULL; /* presets.rs:36 (`earth` needs media/earthmap.jpg, absent upstream) */
    if (which == 10 && use_bvh) return NULL; /* BVHNode::new(&[]) is None and params.rs:37 unwraps it: the reference panics */

    ora_scene *sc = calloc(1, sizeof(*sc));
    xoshiro rng, rng0;
    xoshiro_seed_from_u64(&rng, 0); /* params.rs:21-27 (random_seed = false) */
    rng0 = rng;
    storage_init(&sc->st, &rng, which == 4 ? 2 * 10100 : 2 * 600, 0); /* storage.rs:28-43: 1536 draws */
    hitvec hv = {0};
    switch (which) {
    case 0: preset_random_spheres(sc, width, height, &rng, &hv); break;
    case 1: preset_small(sc, width, height, &hv); break;
    case 2: preset_two_perlin_spheres(sc, width, height, &hv); break;
    case 3: preset_aras(sc, width, height, &hv); break;
    case 4: preset_perlin_spheres(sc, width, height, &rng, &hv); break;
    case 5: preset_random(sc, width, height, &rng, &hv); break;
    case 6: preset_simple_light(sc, width, height, &hv); break;
    case 7: preset_cornell(sc, width, height, &hv, 0); break;
    case 8: preset_cornell(sc, width, height, &hv, 1); break;
    case 9: preset_smallpt(sc, width, height, &hv); break;
    default: preset_final(sc, width, height, &hv); break;
    }
    /* params.rs:29-46 new_scene */
    sc->use_bvh = use_bvh;
    sc->list.hitables = malloc((hv.len ? hv.len : 1) * sizeof(hitable));
    if (hv.len) memcpy(sc->list.hitables, hv.v, hv.len * sizeof(hitable)); /* list order survives the BVH's in-place sort */
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

static camera cam_from_floats(const float *f);
/* Test helper: a scene from the flat description ora_scene_export_* produces (ptref.h), so tests can feed
 * arbitrary worlds to the oracle and to the product alike. The Perlin tables are those of Storage::new with the
 * seed-0 rng (storage.rs:28-43); the BVH, when asked for, is built by BVHNode::new (bvh.rs:64-94) continuing that
 * rng. Isotropic rows must come after all other materials (they are the media's phase functions). Returns NULL on
 * a malformed description. */
ora_scene *ora_scene_from_world(const uint32_t *records16, uint32_t n_hitables, const float *transforms24,
                                uint32_t n_transforms, const float *materials6, uint32_t n_materials,
                                const float *textures7, uint32_t n_textures, const float *cam24, int has_sky,
                                const float *sky3, int use_bvh, const uint32_t *image_wh, const uint8_t *image_bytes,
                                uint32_t n_images) {
    if (!cam24 || (n_hitables && (!records16 || !materials6 || !n_materials)) || (use_bvh && !n_hitables)) return NULL;
    ora_scene *sc = calloc(1, sizeof(*sc));
    xoshiro rng, rng0;
    xoshiro_seed_from_u64(&rng, 0);
    rng0 = rng;
    storage_init(&sc->st, &rng, 2 * (size_t)n_hitables + n_materials + n_textures + 16, n_hitables);
    storage *st = &sc->st;
    if (n_images) {   /* RgbImage sources: (width, height) pairs, pixel data concatenated */
        size_t total = 0;
        for (uint32_t i = 0; i < n_images; ++i) total += 3 * (size_t)image_wh[2 * i] * image_wh[2 * i + 1];
        st->images = calloc(n_images, sizeof(rgb_image));
        st->image_bytes = malloc(total ? total : 1);
        memcpy(st->image_bytes, image_bytes, total);
        size_t off = 0;
        for (uint32_t i = 0; i < n_images; ++i) {
            st->images[i].width = image_wh[2 * i]; st->images[i].height = image_wh[2 * i + 1];
            st->images[i].data = st->image_bytes + off;
            off += 3 * (size_t)image_wh[2 * i] * image_wh[2 * i + 1];
            if (!st->images[i].width || !st->images[i].height) goto bad;
        }
        st->n_images = n_images;
    }
    for (uint32_t i = 0; i < n_textures; ++i) {
        const float *r = textures7 + 7 * i;
        texture t; memset(&t, 0, sizeof t);
        t.kind = (int)r[0]; t.color = V3(r[1], r[2], r[3]); t.scale = r[6];
        if (t.kind == TEX_CHECKER) {
            if (r[4] < 0 || r[5] < 0 || (uint32_t)r[4] >= i || (uint32_t)r[5] >= i) goto bad;
            t.odd = &st->textures[(uint32_t)r[4]]; t.even = &st->textures[(uint32_t)r[5]];
        } else if (t.kind == TEX_NOISE) t.noise = &st->perlin_noise;
        else if (t.kind == TEX_IMAGE) {
            if (r[4] < 0 || (uint32_t)r[4] >= n_images) goto bad;
            t.image = &st->images[(uint32_t)r[4]];
        } else if (t.kind != TEX_CONSTANT) goto bad;
        alloc_texture(st, t);
    }
    uint32_t n_arena = 0;
    for (uint32_t i = 0; i < n_materials; ++i) {
        const float *r = materials6 + 6 * i;
        const int kind = (int)r[0];
        if (kind == MAT_ISOTROPIC) continue;
        if (n_arena != i || kind < 0 || kind > MAT_DIFFUSE_LIGHT) goto bad; /* isotropic rows last */
        material m; memset(&m, 0, sizeof m);
        m.kind = kind; m.albedo = V3(r[1], r[2], r[3]);
        if (kind == MAT_METAL) m.fuzz = r[4];
        if (kind == MAT_DIELECTRIC) m.ref_idx = r[4];
        if (kind == MAT_LAMBERTIAN || kind == MAT_DIFFUSE_LIGHT) {
            if (r[5] < 0 || (uint32_t)r[5] >= n_textures) goto bad;
            m.tex = &st->textures[(uint32_t)r[5]];
        }
        alloc_material(st, m);
        ++n_arena;
    }
    {
        hitvec hv = {0};
        for (uint32_t i = 0; i < n_hitables; ++i) {
            const uint32_t *w = records16 + 16 * i;
            float p[10], density; memcpy(p, w + 6, 40); memcpy(&density, w + 5, 4);
            const int32_t tr = (int32_t)w[3], med = (int32_t)w[4];
            if (w[1] >= n_arena || (tr >= 0 && (uint32_t)tr >= n_transforms) || w[0] > 5) { free(hv.v); goto bad; }
            const material *m = &st->materials[w[1]];
            hitable h; memset(&h, 0, sizeof h);
            switch (w[0]) {
            case 0: { sphere sp = { V3(p[0], p[1], p[2]), p[3] }; h.kind = HIT_SPHERE; h.sph = alloc_sphere(st, sp); h.mat = m; break; }
            case 1: {
                moving_sphere ms; ms.centre_start = V3(p[0], p[1], p[2]); ms.centre_delta = V3(p[3], p[4], p[5]);
                ms.radius = p[6]; ms.time_start = p[7]; ms.inv_time_delta = p[8];
                st->moving[st->n_moving] = ms; h.kind = HIT_MOVING_SPHERE; h.msph = &st->moving[st->n_moving++]; h.mat = m; break; }
            case 5: h = mk_cuboid(st, V3(p[0], p[1], p[2]), V3(p[3], p[4], p[5]), m); break;
            default: h = mk_rect(st, rect_new((int)w[0] - 2, p[0], p[1], p[2], p[3], p[4], w[2] != 0), m); break;
            }
            if (tr >= 0) { /* Instance keeps the transform AND the inverse it was given (instance.rs:16-22 computes it once) */
                const float *a = transforms24 + 24 * tr;
                hitable hi; memset(&hi, 0, sizeof hi); hi.kind = HIT_INSTANCE;
                instance *in = &st->instances[st->n_instances++];
                in->child = h;
                in->transform.x_axis = V3(a[0], a[1], a[2]); in->transform.y_axis = V3(a[3], a[4], a[5]);
                in->transform.z_axis = V3(a[6], a[7], a[8]); in->transform.translation = V3(a[9], a[10], a[11]);
                a += 12;
                in->inv_transform.x_axis = V3(a[0], a[1], a[2]); in->inv_transform.y_axis = V3(a[3], a[4], a[5]);
                in->inv_transform.z_axis = V3(a[6], a[7], a[8]); in->inv_transform.translation = V3(a[9], a[10], a[11]);
                hi.inst = in; h = hi;
            }
            if (med >= 0) {
                if ((uint32_t)med >= n_materials || (int)materials6[6 * med] != MAT_ISOTROPIC) { free(hv.v); goto bad; }
                const float ti = materials6[6 * med + 5];
                if (ti < 0 || (uint32_t)ti >= n_textures) { free(hv.v); goto bad; }
                h = mk_constant_medium(st, h, density, &st->textures[(uint32_t)ti]);
            }
            hv_push(&hv, h);
        }
        sc->cam = cam_from_floats(cam24);
        sc->has_sky = has_sky;
        if (has_sky) sc->sky = V3(sky3[0], sky3[1], sky3[2]);
        sc->use_bvh = use_bvh;
        sc->list.hitables = malloc((hv.len ? hv.len : 1) * sizeof(hitable));
        if (hv.len) memcpy(sc->list.hitables, hv.v, hv.len * sizeof(hitable));
        sc->list.len = hv.len;
        memset(&sc->world, 0, sizeof sc->world);
        if (use_bvh) { sc->world.kind = HIT_BVHNODE; sc->world.node = bvh_new(st, &rng, hv.v, hv.len); }
        else { sc->world.kind = HIT_LIST; sc->world.list = &sc->list; }
        free(hv.v);
    }
    (void)rng0;
    return sc;
bad:
    storage_free(&sc->st);
    free(sc);
    return NULL;
}

/* A world given as a scene graph (collision/hitable.rs:12-21 lets Hitables nest freely): built LITERALLY -- HitableList inside
 * HitableList, Instance of Instance, Instance around a ConstantMedium ... -- over the leaf shapes of `records16` (which carry no
 * wrappers of their own). nodes4: n_nodes rows of (kind, a, b, density bits): kind 0 shape a = leaf index | 1 HitableList of
 * children[a .. a+b) | 2 Instance transforms[a] around node b | 3 ConstantMedium, Isotropic material a, boundary node b |
 * 4 BVHNode (bvh.rs:19-35 as a Hitable anywhere in the graph, hitable.rs:12-21): box bvh_minmax6[a], children = the NODES bvh_lr2[a].
 * The product flattens the same graph where the list form can express it and interprets it otherwise (include/ptgpu.h pt_node);
 * this is what it must agree with. List worlds only. */
#define GRAPH_POOL ((size_t)1 << 16)   /* entries per pool of ora_scene_from_graph */
static int graph_build(ora_scene *sc, const uint32_t *nodes4, uint32_t n_nodes, const uint32_t *children, uint32_t n_children,
                       const float *transforms24, uint32_t n_transforms, const float *materials6, uint32_t n_materials, uint32_t n_textures,
                       uint32_t node, uint32_t depth, size_t *n_inst, size_t *n_med, size_t *n_lists, size_t *n_child, hitable *out) {
    if (node >= n_nodes || depth > 64) return 0;
    const uint32_t *w = nodes4 + 4 * node;
    const size_t pool = GRAPH_POOL;   /* every pool is checked BEFORE it is written: a DAG expands per path */
    switch (w[0]) {
    case 0:
        if (w[1] >= sc->list.len) return 0;
        *out = sc->list.hitables[w[1]];
        return 1;
    case 1: {
        if ((uint64_t)w[1] + w[2] > n_children) return 0;
        if (*n_lists >= pool || *n_child + (size_t)w[2] > pool) return 0;
        hitable_list *l = &sc->g_lists[(*n_lists)++];
        l->hitables = sc->g_children + *n_child; l->len = w[2];
        *n_child += w[2];
        for (uint32_t j = 0; j < w[2]; ++j)
            if (!graph_build(sc, nodes4, n_nodes, children, n_children, transforms24, n_transforms, materials6, n_materials, n_textures, children[w[1] + j],
                             depth + 1, n_inst, n_med, n_lists, n_child, &l->hitables[j])) return 0;
        memset(out, 0, sizeof *out); out->kind = HIT_LIST; out->list = l;
        return 1; }
    case 2: {
        if (w[1] >= n_transforms) return 0;
        hitable child;
        if (!graph_build(sc, nodes4, n_nodes, children, n_children, transforms24, n_transforms, materials6, n_materials, n_textures, w[2], depth + 1, n_inst,
                         n_med, n_lists, n_child, &child)) return 0;
        const float *a = transforms24 + 24 * w[1];
        if (*n_inst >= pool) return 0;
        instance *in = &sc->g_instances[(*n_inst)++];
        in->child = child;   /* keeps the transform AND the inverse it was given (instance.rs:16-22 computes it once) */
        in->transform.x_axis = V3(a[0], a[1], a[2]); in->transform.y_axis = V3(a[3], a[4], a[5]);
        in->transform.z_axis = V3(a[6], a[7], a[8]); in->transform.translation = V3(a[9], a[10], a[11]);
        a += 12;
        in->inv_transform.x_axis = V3(a[0], a[1], a[2]); in->inv_transform.y_axis = V3(a[3], a[4], a[5]);
        in->inv_transform.z_axis = V3(a[6], a[7], a[8]); in->inv_transform.translation = V3(a[9], a[10], a[11]);
        memset(out, 0, sizeof *out); out->kind = HIT_INSTANCE; out->inst = in;
        return 1; }
    case 3: {
        if (w[1] >= n_materials || (int)materials6[6 * w[1]] != MAT_ISOTROPIC) return 0;
        const float ti = materials6[6 * w[1] + 5];
        if (ti < 0 || (uint32_t)ti >= n_textures) return 0;
        hitable child;
        if (!graph_build(sc, nodes4, n_nodes, children, n_children, transforms24, n_transforms, materials6, n_materials, n_textures, w[2], depth + 1, n_inst,
                         n_med, n_lists, n_child, &child)) return 0;
        if (*n_med >= pool) return 0;
        constant_medium *cm = &sc->g_media[(*n_med)++];
        cm->child = child; memcpy(&cm->density, &w[3], 4);
        memset(&cm->phase_function, 0, sizeof cm->phase_function);
        cm->phase_function.kind = MAT_ISOTROPIC; cm->phase_function.tex = &sc->st.textures[(uint32_t)ti];
        memset(out, 0, sizeof *out); out->kind = HIT_CONSTANT_MEDIUM; out->med = cm;
        return 1; }
    case 4: {
        if (w[1] >= sc->gb_n || !sc->gb_minmax6 || !sc->gb_lr2) return 0;
        const int32_t l = sc->gb_lr2[2 * w[1]], r = sc->gb_lr2[2 * w[1] + 1];
        if (l < 0 || r < 0) return 0;
        hitable lhs, rhs;
        if (!graph_build(sc, nodes4, n_nodes, children, n_children, transforms24, n_transforms, materials6, n_materials, n_textures, (uint32_t)l, depth + 1, n_inst,
                         n_med, n_lists, n_child, &lhs) ||
            !graph_build(sc, nodes4, n_nodes, children, n_children, transforms24, n_transforms, materials6, n_materials, n_textures, (uint32_t)r, depth + 1, n_inst,
                         n_med, n_lists, n_child, &rhs)) return 0;
        if (sc->gb_used >= pool) return 0;
        bvhnode *bn = &sc->g_bvh[sc->gb_used++];
        const float *m = sc->gb_minmax6 + 6 * w[1];
        bn->bb.min = V3(m[0], m[1], m[2]); bn->bb.max = V3(m[3], m[4], m[5]);
        bn->lhs = lhs; bn->rhs = rhs;
        *out = mk_node_hitable(bn);
        return 1; }
    default: return 0;
    }
}

ora_scene *ora_scene_from_graph_bvh(const uint32_t *records16, uint32_t n_hitables, const float *transforms24, uint32_t n_transforms,
                                    const float *materials6, uint32_t n_materials, const float *textures7, uint32_t n_textures,
                                    const float *cam24, int has_sky, const float *sky3, const uint32_t *nodes4, uint32_t n_nodes,
                                    const uint32_t *children, uint32_t n_children, uint32_t root, const float *bvh_minmax6, const int32_t *bvh_lr2, uint32_t n_bvh) {
    if (!nodes4 || !n_nodes || root >= n_nodes) return NULL;
    for (uint32_t i = 0; i < n_hitables; ++i)   /* leaves carry no wrappers: those are nodes */
        if ((int32_t)records16[16 * i + 3] >= 0 || (int32_t)records16[16 * i + 4] >= 0) return NULL;
    ora_scene *sc = ora_scene_from_world(records16, n_hitables, transforms24, n_transforms, materials6, n_materials, textures7, n_textures, cam24,
                                         has_sky, sky3, 0, NULL, NULL, 0);
    if (!sc) return NULL;
    sc->gb_minmax6 = bvh_minmax6, sc->gb_lr2 = bvh_lr2, sc->gb_n = n_bvh, sc->gb_used = 0;
    /* a node can be reached along several paths (a DAG): size the pools for the expanded tree, bounded */
    const size_t cap = GRAPH_POOL;
    sc->g_instances = calloc(cap, sizeof(instance)); sc->g_media = calloc(cap, sizeof(constant_medium));
    sc->g_lists = calloc(cap, sizeof(hitable_list)); sc->g_children = calloc(cap, sizeof(hitable)); sc->g_bvh = calloc(cap, sizeof(bvhnode));
    size_t ni = 0, nm = 0, nl = 0, nc = 0;
    hitable rooth;
    /* (graph_build checks its indices and the pools' capacity before every write) */
    if (!sc->g_instances || !sc->g_media || !sc->g_lists || !sc->g_children || !sc->g_bvh ||
        !graph_build(sc, nodes4, n_nodes, children, n_children, transforms24, n_transforms, materials6, n_materials, n_textures, root, 0, &ni, &nm, &nl, &nc,
                     &rooth)) {
        ora_scene_free(sc);
        return NULL;
    }
    sc->world = rooth;   /* scene.rs:19: the world is whatever Hitable the graph's root is */
    sc->gb_minmax6 = NULL, sc->gb_lr2 = NULL, sc->gb_n = 0;   /* (the caller's arrays are not kept) */
    return sc;
}

ora_scene *ora_scene_from_graph(const uint32_t *records16, uint32_t n_hitables, const float *transforms24, uint32_t n_transforms,
                                const float *materials6, uint32_t n_materials, const float *textures7, uint32_t n_textures,
                                const float *cam24, int has_sky, const float *sky3, const uint32_t *nodes4, uint32_t n_nodes,
                                const uint32_t *children, uint32_t n_children, uint32_t root) {
    return ora_scene_from_graph_bvh(records16, n_hitables, transforms24, n_transforms, materials6, n_materials, textures7, n_textures, cam24, has_sky, sky3,
                                    nodes4, n_nodes, children, n_children, root, NULL, NULL, 0);
}

void ora_scene_free(ora_scene *s) {
    if (!s) return;
    storage_free(&s->st);
    free(s->list.hitables);
    free(s->g_instances); free(s->g_media); free(s->g_lists); free(s->g_children); free(s->g_bvh);
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
    if (hitable_ray_hit(&s->world, ray_in, MIN_T, MAX_T, rng, &hit, &mat)) {
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
    uint32_t *pixel_rays; /* optional: rays of job pixel k (the per-pixel summand of scene.rs:118) */
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
            uint64_t rays = render_pixel(j->s, j->width, j->height, j->samples, j->max_depth, j->frame_num, i,
                                         j->buffer + 3 * i);
            if (j->pixel_rays) j->pixel_rays[k - j->begin] = (uint32_t)rays;
            local += rays;
        }
    }
    atomic_fetch_add(&j->ray_count, local); /* scene.rs:118 */
    atomic_fetch_add(&g_bvh_nodes, tl_bvh_nodes); atomic_fetch_add(&g_bvh_leaves, tl_bvh_leaves);
    tl_bvh_nodes = tl_bvh_leaves = 0;
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

uint64_t ora_scene_update_pixels_counted(const ora_scene *s, uint32_t width, uint32_t height, uint32_t samples,
                                         uint32_t max_depth, uint32_t frame_num, float *buffer,
                                         const uint32_t *pixels, uint64_t n_pixels, uint32_t *pixel_rays, int nthreads) {
    job j; memset(&j, 0, sizeof j);
    j.s = s; j.width = width; j.height = height; j.samples = samples; j.max_depth = max_depth;
    j.frame_num = frame_num; j.buffer = buffer; j.pixels = pixels; j.begin = 0; j.end = n_pixels;
    j.pixel_rays = pixel_rays;
    atomic_store(&j.next, 0);
    return run_job(&j, nthreads);
}

/* ======================================================================== */
/* flat export                                                              */
/* ======================================================================== */
uint32_t ora_scene_num_spheres(const ora_scene *s) { return (uint32_t)s->list.len; }
/* arena materials, then one Isotropic phase function per ConstantMedium (constant_medium.rs:13) */
uint32_t ora_scene_num_materials(const ora_scene *s) { return (uint32_t)(s->st.n_materials + s->st.n_media); }
uint32_t ora_scene_num_hitables(const ora_scene *s) { return (uint32_t)s->list.len; }
uint32_t ora_scene_num_transforms(const ora_scene *s) { return (uint32_t)s->st.n_instances; }
int ora_scene_is_sphere_world(const ora_scene *s) {
    for (size_t i = 0; i < s->list.len; ++i) if (s->list.hitables[i].kind != HIT_SPHERE) return 0;
    return 1;
}
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
    for (size_t i = 0; i < s->st.n_media; ++i) {
        const material *m = &s->st.media[i].phase_function;
        float *r = rows6 + 6 * (s->st.n_materials + i);
        r[0] = (float)m->kind; r[1] = r[2] = r[3] = r[4] = 0.0f;
        r[5] = (float)(m->tex - s->st.textures);
    }
}

/* world export (ptref.h): one 16-word record per list entry */
static void put_f(uint32_t *w, float f) { memcpy(w, &f, 4); }
static void export_base(const ora_scene *s, const hitable *h, uint32_t *w) {
    float p[10] = {0};
    switch (h->kind) {
    case HIT_SPHERE:
        w[0] = 0; p[0] = h->sph->centre.x; p[1] = h->sph->centre.y; p[2] = h->sph->centre.z; p[3] = h->sph->radius; break;
    case HIT_MOVING_SPHERE:
        w[0] = 1;
        p[0] = h->msph->centre_start.x; p[1] = h->msph->centre_start.y; p[2] = h->msph->centre_start.z;
        p[3] = h->msph->centre_delta.x; p[4] = h->msph->centre_delta.y; p[5] = h->msph->centre_delta.z;
        p[6] = h->msph->radius; p[7] = h->msph->time_start; p[8] = h->msph->inv_time_delta; break;
    case HIT_RECT:
        w[0] = 2 + (uint32_t)h->rct->axis; w[2] = (uint32_t)h->rct->flip_normals;
        p[0] = h->rct->a0; p[1] = h->rct->a1; p[2] = h->rct->b0; p[3] = h->rct->b1; p[4] = h->rct->k; break;
    default: /* HIT_CUBOID: p0, p1 (cuboid.rs:11-23 derives the faces) */
        w[0] = 5;
        p[0] = h->cub->bb.min.x; p[1] = h->cub->bb.min.y; p[2] = h->cub->bb.min.z;
        p[3] = h->cub->bb.max.x; p[4] = h->cub->bb.max.y; p[5] = h->cub->bb.max.z; break;
    }
    w[1] = (uint32_t)(h->mat - s->st.materials);
    for (int i = 0; i < 10; ++i) put_f(&w[6 + i], p[i]);
}
void ora_scene_export_world(const ora_scene *s, uint32_t *records16, float *transforms24) {
    for (size_t i = 0; i < s->list.len; ++i) {
        const hitable *h = &s->list.hitables[i];
        uint32_t *w = records16 + 16 * i;
        memset(w, 0, 64);
        w[3] = (uint32_t)-1; w[4] = (uint32_t)-1;
        if (h->kind == HIT_CONSTANT_MEDIUM) {
            w[4] = (uint32_t)(s->st.n_materials + (size_t)(h->med - s->st.media));
            put_f(&w[5], h->med->density);
            h = &h->med->child;
        }
        if (h->kind == HIT_INSTANCE) {
            w[3] = (uint32_t)(h->inst - s->st.instances);
            h = &h->inst->child;
        }
        export_base(s, h, w);
    }
    for (size_t i = 0; i < s->st.n_instances; ++i) {
        const affine3 *m[2] = { &s->st.instances[i].transform, &s->st.instances[i].inv_transform };
        for (int k = 0; k < 2; ++k) {
            float *o = transforms24 + 24 * i + 12 * k;
            const v3 *c[4] = { &m[k]->x_axis, &m[k]->y_axis, &m[k]->z_axis, &m[k]->translation };
            for (int j = 0; j < 4; ++j) { o[3 * j] = c[j]->x; o[3 * j + 1] = c[j]->y; o[3 * j + 2] = c[j]->z; }
        }
    }
}

void ora_scene_export_textures(const ora_scene *s, float *rows7) {
    for (size_t i = 0; i < s->st.n_textures; ++i) {
        const texture *t = &s->st.textures[i];
        float *r = rows7 + 7 * i;
        r[0] = (float)t->kind; r[1] = t->color.x; r[2] = t->color.y; r[3] = t->color.z;
        r[4] = t->odd ? (float)(t->odd - s->st.textures) : (t->image ? (float)(t->image - s->st.images) : -1.0f);
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
    /* leaf: position in LIST order (the BVH build sorts copies of the list entries) */
    for (size_t i = 0; i < s->list.len; ++i)
        if (!memcmp(&s->list.hitables[i], h, sizeof(hitable))) return ~(int32_t)i;
    return ~(int32_t)0x3fffffff; /* unreachable */
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
/* f32::ln as constant_medium.rs:60 evaluates it (glibc logf, assumption A7) */
/* BVHNode::ray_hit invocations and leaf hitables tested since the last reset (all update calls, all threads) */
void ora_bvh_counters(uint64_t out2[2], int reset) {
    out2[0] = atomic_load(&g_bvh_nodes); out2[1] = atomic_load(&g_bvh_leaves);
    if (reset) { atomic_store(&g_bvh_nodes, 0); atomic_store(&g_bvh_leaves, 0); }
}
void ora_ln_array(const float *in, float *out, uint64_t n) { for (uint64_t i = 0; i < n; ++i) out[i] = logf(in[i]); }
/* One Hitable::ray_hit on list entry `index` of the built scene (any kind; media draw from `state`).
 * out7 = point3, normal3, t; returns 1 on hit and the material index (export numbering) in *material. */
int ora_hitable_ray_hit(const ora_scene *s, uint32_t index, const float origin[3], const float direction[3], float time,
                        float t_min, float t_max, uint64_t state[4], float out7[7], uint32_t *material_out) {
    xoshiro r; memcpy(r.s, state, 32);
    ray ry = ray_new(V3(origin[0], origin[1], origin[2]), V3(direction[0], direction[1], direction[2]), time);
    ray_hit h; const material *m = NULL;
    int hit = hitable_ray_hit(&s->list.hitables[index], &ry, t_min, t_max, &r, &h, &m);
    memcpy(state, r.s, 32);
    if (hit) {
        out7[0] = h.point.x; out7[1] = h.point.y; out7[2] = h.point.z;
        out7[3] = h.normal.x; out7[4] = h.normal.y; out7[5] = h.normal.z; out7[6] = h.t;
        if (m >= s->st.materials && m < s->st.materials + s->st.n_materials) *material_out = (uint32_t)(m - s->st.materials);
        else *material_out = (uint32_t)s->st.n_materials + (uint32_t)(((const constant_medium *)((const char *)m - offsetof(constant_medium, phase_function))) - s->st.media);
    }
    return hit;
}

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

/* ======================================================================== */
/* collision/spheres_soa.rs:12-392 -- SpheresSoA (bench-only in the reference:  */
/* nothing but its own #[bench] functions calls it).           */
/* The arithmetic DIFFERS from Sphere::ray_hit (sphere.rs:29-66): co = centre -  */
/* origin, discriminant = nb*nb - c WITHOUT `a` (it assumes |direction| = 1),    */
/* t = nb -+ sqrt WITHOUT the division, normal = (p - c) * (1/r) instead of / r. */
/* ======================================================================== */
typedef struct { size_t len, num; float *cx, *cy, *cz, *rsq, *rinv; const material **mat; } spheres_soa;

/* spheres_soa.rs:26-74 (chunk = TargetFeature::get_bits() / 32: 8 AVX2, 4 SSE4.1, 1 FallBack; simd.rs:35-41) */
static int soa_new(spheres_soa *o, const hitable_list *l, size_t chunk) {
    memset(o, 0, sizeof *o);
    o->num = l->len;
    o->len = (l->len + chunk - 1) / chunk * chunk;   /* math.rs align_to */
    o->cx = malloc(5 * (o->len + 1) * sizeof(float));
    o->cy = o->cx + o->len; o->cz = o->cy + o->len; o->rsq = o->cz + o->len; o->rinv = o->rsq + o->len;
    o->mat = calloc(o->len + 1, sizeof *o->mat);
    if (!o->cx || !o->mat) return 0;
    for (size_t i = 0; i < l->len; ++i) {
        const hitable *h = &l->hitables[i];
        if (h->kind != HIT_SPHERE) return 0;   /* spheres_soa.rs:52: panics on anything else */
        o->cx[i] = h->sph->centre.x; o->cy[i] = h->sph->centre.y; o->cz[i] = h->sph->centre.z;
        o->rsq[i] = h->sph->radius * h->sph->radius;
        o->rinv[i] = 1.0f / h->sph->radius;
        o->mat[i] = h->mat;
    }
    for (size_t i = l->len; i < o->len; ++i) {   /* spheres_soa.rs:55-62 padding */
        o->cx[i] = o->cy[i] = o->cz[i] = 3.40282346638528859812e+38f;
        o->rsq[i] = 0.0f; o->rinv[i] = 0.0f; o->mat[i] = NULL;
    }
    return 1;
}
static void soa_free(spheres_soa *o) { free(o->cx); free(o->mat); }

/* material.rs:41-49, 169-180: (u, v) only for Image-textured Lambertian / DiffuseLight */
static void soa_sphere_uv(const material *m, v3 normal, float *u, float *v) {
    *u = 0.0f; *v = 0.0f;
    if ((m->kind == MAT_LAMBERTIAN || m->kind == MAT_DIFFUSE_LIGHT) && m->tex && m->tex->kind == TEX_IMAGE) {
        const float pi = 3.14159274101257324f, frac_1_2pi = 1.0f / (2.0f * pi);
        float phi = atan2f(normal.x, normal.y), theta = asinf(normal.y);
        *u = 1.0f - (phi + pi) * frac_1_2pi;
        *v = (theta + 1.57079637050628662f) * 0.318309873342514038f;
    }
}
/* the common tail of all three variants (spheres_soa.rs:135-155, 235-265, 358-388) */
static int soa_finish(const spheres_soa *o, const ray *r, size_t hit_index, float hit_t, ray_hit *out, size_t *index_out) {
    out->point = point_at_parameter(r, hit_t);
    v3 centre = V3(o->cx[hit_index], o->cy[hit_index], o->cz[hit_index]);
    out->normal = v3_scale(v3_sub(out->point, centre), o->rinv[hit_index]);
    out->t = hit_t;
    soa_sphere_uv(o->mat[hit_index], out->normal, &out->u, &out->v);
    *index_out = hit_index;
    return 1;
}
/* spheres_soa.rs:105-155 hit_scalar */
static int soa_hit_scalar(const spheres_soa *o, const ray *r, float t_min, float t_max, ray_hit *out, size_t *index_out) {
    float hit_t = t_max;
    size_t hit_index = o->len;
    for (size_t index = 0; index < o->len; ++index) {
        v3 co = v3_sub(V3(o->cx[index], o->cy[index], o->cz[index]), r->origin);
        float nb = v3_dot(co, r->direction);
        float c = v3_dot(co, co) - o->rsq[index];
        float discriminant = nb * nb - c;
        if (discriminant > 0.0f) {
            float discriminant_sqrt = sqrtf(discriminant);
            float t = nb - discriminant_sqrt;
            if (t < t_min) t = nb + discriminant_sqrt;
            if (t > t_min && t < hit_t) { hit_t = t; hit_index = index; }
        }
    }
    if (hit_index < o->len) return soa_finish(o, r, hit_index, hit_t, out, index_out);
    return 0;
}
/* spheres_soa.rs:161-268 hit_sse4_1 (lanes = 4) / :274-391 hit_avx2 (lanes = 8): `lanes` running minima, one per SIMD lane,
 * then the horizontal minimum and the LOWEST LANE holding it (:232-236, :355-359) -- not the lowest sphere index. */
static int soa_hit_lanes(const spheres_soa *o, size_t lanes, const ray *r, float t_min, float t_max, ray_hit *out, size_t *index_out) {
    float hit_t[8]; int32_t hit_index[8];
    for (size_t l = 0; l < lanes; ++l) { hit_t[l] = t_max; hit_index[l] = -1; }
    for (size_t chunk = 0; chunk + lanes <= o->len; chunk += lanes) {
        for (size_t l = 0; l < lanes; ++l) {   /* (the lanes are independent: the movemask test around :209 / :323 only skips work) */
            size_t i = chunk + l;
            float co_x = o->cx[i] - r->origin.x, co_y = o->cy[i] - r->origin.y, co_z = o->cz[i] - r->origin.z;
            float nb = (co_x * r->direction.x + co_y * r->direction.y) + co_z * r->direction.z;   /* simd.rs:253-265 dot3 */
            float c = ((co_x * co_x + co_y * co_y) + co_z * co_z) - o->rsq[i];
            float discr = nb * nb - c;
            if (discr > 0.0f) {
                float discr_sqrt = sqrtf(discr);
                float t0 = nb - discr_sqrt, t1 = nb + discr_sqrt;
                float t = t0 > t_min ? t0 : t1;             /* blendv(t1, t0, t0 > t_min) */
                if (t > t_min && t < hit_t[l]) { hit_index[l] = (int32_t)i; hit_t[l] = t; }
            }
        }
    }
    float min_hit_t = hit_t[0];
    for (size_t l = 1; l < lanes; ++l) min_hit_t = hit_t[l] < min_hit_t ? hit_t[l] : min_hit_t;   /* hmin (no NaNs can be stored) */
    if (min_hit_t < t_max)
        for (size_t l = 0; l < lanes; ++l)
            if (hit_t[l] == min_hit_t) return soa_finish(o, r, (size_t)hit_index[l], hit_t[l], out, index_out);
    return 0;
}

static void put_hit9(const ray_hit *h, float out9[9]) {
    out9[0] = h->point.x; out9[1] = h->point.y; out9[2] = h->point.z;
    out9[3] = h->normal.x; out9[4] = h->normal.y; out9[5] = h->normal.z; out9[6] = h->t; out9[7] = h->u; out9[8] = h->v;
}
/* SpheresSoA::new over the scene's list + one ray_hit; lanes: 1 hit_scalar, 4 hit_sse4_1, 8 hit_avx2. -1: the list holds a non-sphere. */
int ora_soa_ray_hit(const ora_scene *s, int lanes, const float o[3], const float d[3], float t_min, float t_max, float out9[9], uint32_t *index_out) {
    if (lanes != 1 && lanes != 4 && lanes != 8) return -1;
    spheres_soa soa;
    if (!soa_new(&soa, &s->list, (size_t)lanes)) { soa_free(&soa); return -1; }
    ray ry = ray_new(V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), 0.0f);
    ray_hit h; size_t idx = 0;
    int hit = lanes == 1 ? soa_hit_scalar(&soa, &ry, t_min, t_max, &h, &idx) : soa_hit_lanes(&soa, (size_t)lanes, &ry, t_min, t_max, &h, &idx);
    if (hit) { put_hit9(&h, out9); *index_out = (uint32_t)idx; }
    soa_free(&soa);
    return hit;
}
/* Hitable::ray_hit on the scene's WORLD (scene.rs:58: the HitableList, or the BVH root when the scene was built with one) for one
 * explicit ray; out9 as above, *index_out = the list entry that was hit (sphere worlds: the entry owning the returned material). */
int ora_world_ray_hit(const ora_scene *s, const float o[3], const float d[3], float time, float t_min, float t_max, uint64_t state[4],
                      float out9[9], uint32_t *index_out) {
    xoshiro r; memcpy(r.s, state, 32);
    ray ry = ray_new(V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), time);
    ray_hit h; const material *m = NULL;
    int hit = hitable_ray_hit(&s->world, &ry, t_min, t_max, &r, &h, &m);
    memcpy(state, r.s, 32);
    if (hit) {
        put_hit9(&h, out9);
        *index_out = 0xffffffffu;
        for (size_t i = 0; i < s->list.len; ++i)
            if (s->list.hitables[i].mat == m) { *index_out = (uint32_t)i; break; }
    }
    return hit;
}
/* The reference's six #[bench] units (bench.rs:8-26; hitable_list.rs:68-75, spheres_soa.rs:464-485, bvh.rs:361-379): `reps` calls of
 * one closest-hit query on one fixed ray, nanoseconds per call (what `b.iter` reports). which: 0 world (list or BVH as built),
 * 1 / 4 / 8 SpheresSoA scalar / 4 lanes / 8 lanes. */
double ora_bench_ray_hit(const ora_scene *s, int which, const float o[3], const float d[3], float time, uint64_t reps) {
    ray ry = ray_new(V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), time);
    xoshiro rng; xoshiro_seed_from_u64(&rng, 0);
    spheres_soa soa; memset(&soa, 0, sizeof soa);
    if (which != 0 && !soa_new(&soa, &s->list, (size_t)which)) { soa_free(&soa); return -1.0; }
    volatile float sink = 0.0f;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (uint64_t i = 0; i < reps; ++i) {
        ray_hit h; const material *m; size_t idx;
        __asm__ volatile("" : "+m"(ry) : : "memory");   /* (the query is re-evaluated every time, as under test::black_box) */
        int hit = which == 0 ? hitable_ray_hit(&s->world, &ry, MIN_T, MAX_T, &rng, &h, &m)
                : which == 1 ? soa_hit_scalar(&soa, &ry, MIN_T, MAX_T, &h, &idx) : soa_hit_lanes(&soa, (size_t)which, &ry, MIN_T, MAX_T, &h, &idx);
        if (hit) sink = sink + h.t;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (which != 0) soa_free(&soa);
    (void)sink;
    return ((double)(t1.tv_sec - t0.tv_sec) * 1e9 + (double)(t1.tv_nsec - t0.tv_nsec)) / (double)(reps ? reps : 1);
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
