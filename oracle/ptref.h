/*
 * ptref.h -- public interface of the CPU ORACLE ("ptref").
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library. The product path (pathtrace-rs_amd/) never includes,
 * links or calls anything in oracle/.
 *
 * ptref is a plain-C restatement of the reference's (bitshifter/pathtrace-rs
 * 0.1.2) hot path: Scene::update -> ray_trace -> Hitable::ray_hit ->
 * Material::scatter, plus the host-side scene construction needed to feed it
 * (Storage::new / presets / Params::new_scene / Camera::new). Every function
 * in ptref.c cites the reference file:line it follows.
 *
 * PARITY STATUS: "parity unpinned" against the Rust binary. The reference
 * ships no tests, no golden vectors and cannot be built here (no Rust
 * toolchain, no vendored crates). The oracle is pinned only to
 *   (1) the public xoshiro256+ / SplitMix64 known-answer vectors, and
 *   (2) the survey-side derived fixtures in tests/golden/ (SURVEY.md 8c).
 * Un-vendored third-party arithmetic (rand 0.8.5, rand_xoshiro 0.6.0,
 * glam 0.20.5) is restated from the crates' published algorithms; the
 * assumptions are listed at the top of ptref.c.
 */
#ifndef PTREF_H
#define PTREF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_scene ora_scene;

/* ---- scene construction (offline.rs:16-24) ------------------------------
 * rng = Params::new_rng (seed 0) -> Storage::new(rng) -> presets::from_name
 * -> Params::new_scene (List, or BVH when use_bvh). Returns NULL when the
 * preset name is unknown (presets.rs:36). */
ora_scene *ora_scene_from_preset(const char *name, uint32_t width,
                                 uint32_t height, int use_bvh);
void ora_scene_free(ora_scene *s);
/* Test helper: build the scene from the flat description the export functions below produce
 * (records16 / transforms24 / materials rows6 / textures rows7 / cam24), so arbitrary worlds can be
 * fed to the oracle and to the product alike. Perlin tables = Storage::new with the seed-0 rng;
 * use_bvh builds the tree with BVHNode::new (bvh.rs:64-94). Isotropic material rows (the media's
 * phase functions) must come last. NULL on a malformed description. */
ora_scene *ora_scene_from_world(const uint32_t *records16, uint32_t n_hitables,
                                const float *transforms24, uint32_t n_transforms,
                                const float *materials6, uint32_t n_materials,
                                const float *textures7, uint32_t n_textures,
                                const float *cam24, int has_sky, const float *sky3, int use_bvh,
                                const uint32_t *image_wh, const uint8_t *image_bytes, uint32_t n_images);
/* The same with the world given as a scene graph (nested Hitables built literally, list worlds): nodes4 = n_nodes rows of
 * (kind, a, b, density bits) as include/ptgpu.h pt_node; records16 are the leaf shapes (no wrappers of their own). */
ora_scene *ora_scene_from_graph(const uint32_t *records16, uint32_t n_hitables, const float *transforms24, uint32_t n_transforms,
                                const float *materials6, uint32_t n_materials, const float *textures7, uint32_t n_textures,
                                const float *cam24, int has_sky, const float *sky3, const uint32_t *nodes4, uint32_t n_nodes,
                                const uint32_t *children, uint32_t n_children, uint32_t root);
/* ... and with BVHNodes inside the graph (node kind 4: a = row of bvh_minmax6 / bvh_lr2, whose two children are NODE indices) */
ora_scene *ora_scene_from_graph_bvh(const uint32_t *records16, uint32_t n_hitables, const float *transforms24, uint32_t n_transforms,
                                    const float *materials6, uint32_t n_materials, const float *textures7, uint32_t n_textures,
                                    const float *cam24, int has_sky, const float *sky3, const uint32_t *nodes4, uint32_t n_nodes,
                                    const uint32_t *children, uint32_t n_children, uint32_t root, const float *bvh_minmax6, const int32_t *bvh_lr2, uint32_t n_bvh);
/* (Texture::Image rows: kind 3, odd_id = image index; images = (width, height) pairs + concatenated RGB8 rows) */

/* ---- Scene::update (scene.rs:73-121) ------------------------------------
 * buffer: width*height*3 floats, row 0 = bottom row, read AND written
 * (frame blend). Returns the ray count. nthreads<=0 -> all online cores. */
uint64_t ora_scene_update(const ora_scene *s, uint32_t width, uint32_t height,
                          uint32_t samples, uint32_t max_depth,
                          uint32_t frame_num, float *buffer, int nthreads);

/* Same, restricted to pixel indices [pix_begin, pix_end) of the full frame
 * (buffer is still the full-frame buffer; other pixels are untouched). Lets
 * tests and the bounded cpu_baseline sample check full-size configs. */
uint64_t ora_scene_update_range(const ora_scene *s, uint32_t width,
                                uint32_t height, uint32_t samples,
                                uint32_t max_depth, uint32_t frame_num,
                                float *buffer, uint64_t pix_begin,
                                uint64_t pix_end, int nthreads);

/* Same, for an explicit list of pixel indices. */
uint64_t ora_scene_update_pixels(const ora_scene *s, uint32_t width,
                                 uint32_t height, uint32_t samples,
                                 uint32_t max_depth, uint32_t frame_num,
                                 float *buffer, const uint32_t *pixels,
                                 uint64_t n_pixels, int nthreads);

/* Same, and pixel_rays[k] receives the rays traced for pixels[k] (the
 * per-pixel summand of scene.rs:118's ray_count; test fixtures keep them
 * per 8x8 tile). pixel_rays may be NULL. */
uint64_t ora_scene_update_pixels_counted(const ora_scene *s, uint32_t width,
                                         uint32_t height, uint32_t samples,
                                         uint32_t max_depth, uint32_t frame_num,
                                         float *buffer, const uint32_t *pixels,
                                         uint64_t n_pixels, uint32_t *pixel_rays,
                                         int nthreads);

/* ---- flat export of the built scene (for cross-checks / feeding the
 *      product through its C ABI in tests) -------------------------------- */
uint32_t ora_scene_num_spheres(const ora_scene *s);
uint32_t ora_scene_num_materials(const ora_scene *s);
uint32_t ora_scene_num_textures(const ora_scene *s);
uint32_t ora_scene_num_bvh_nodes(const ora_scene *s);
int32_t ora_scene_bvh_root(const ora_scene *s);   /* -1 when list mode */
int ora_scene_has_perlin_texture(const ora_scene *s);
uint64_t ora_scene_build_draws(const ora_scene *s); /* scene-build RNG ledger */
/* spheres: n*4 floats (cx,cy,cz,radius) in list order; material: n ids */
void ora_scene_export_spheres(const ora_scene *s, float *xyzr,
                              uint32_t *material_id);
/* ---- general worlds (moving spheres, rects, cuboids,
 * instances, constant media). ora_scene_export_spheres is only valid when
 * ora_scene_is_sphere_world(); ora_scene_export_world works for every preset:
 * one 16-word (64-byte) record per HitableList entry,
 *   w0 kind: 0 Sphere, 1 MovingSphere, 2/3/4 Rect XY/XZ/YZ, 5 Cuboid (the innermost shape)
 *   w1 material index   w2 flip_normals (Rect)
 *   w3 Instance wrapper: transform index or -1
 *   w4 ConstantMedium wrapper (outermost): index of its Isotropic phase-function material or -1
 *   w5 density (f32)    w6..w15 ten f32 shape parameters:
 *      Sphere cx cy cz r | MovingSphere c0(3) delta(3) r time_start inv_time_delta
 *      Rect a0 a1 b0 b1 k (in-plane coordinates in the variant's own order) | Cuboid p0(3) p1(3)
 * transforms: 24 floats each = Affine3A {x_axis,y_axis,z_axis,translation} then its inverse.
 * Materials are the arena materials followed by one Isotropic per ConstantMedium. */
uint32_t ora_scene_num_hitables(const ora_scene *s);
uint32_t ora_scene_num_transforms(const ora_scene *s);
int ora_scene_is_sphere_world(const ora_scene *s);
void ora_scene_export_world(const ora_scene *s, uint32_t *records16, float *transforms24);

/* materials: n rows of 6 floats: kind, a0,a1,a2, param(fuzz|ref_idx), texture_id(-1 none)
 * kinds: 0 lambertian 1 metal 2 dielectric 3 diffuse_light 4 isotropic */
void ora_scene_export_materials(const ora_scene *s, float *rows6);
/* textures: n rows of 7 floats: kind, c0,c1,c2, odd_id, even_id, scale
 * kinds: 0 constant 1 checker 2 noise 3 image (odd_id = image index) */
void ora_scene_export_textures(const ora_scene *s, float *rows7);
/* perlin: randvec 256*3 floats, perm_x/y/z 256 u32 each */
void ora_scene_export_perlin(const ora_scene *s, float *randvec,
                             uint32_t *perm_x, uint32_t *perm_y,
                             uint32_t *perm_z);
/* bvh nodes: n rows of 8: min3,max3 as floats then lhs,rhs as int32 bit
 * patterns (child >= 0: node index; child < 0: ~sphere_index) */
void ora_scene_export_bvh(const ora_scene *s, float *minmax6, int32_t *lhs_rhs2);
/* camera: 24 floats in camera.rs:8-19 field order */
void ora_scene_export_camera(const ora_scene *s, float *cam24);
/* sky: returns has_sky; rgb filled when set */
int ora_scene_export_sky(const ora_scene *s, float *rgb3);

/* ---- unit-level probes for known-answer tests --------------------------- */
void ora_splitmix64(uint64_t seed, uint64_t *out, int n);
void ora_xoshiro_seed_from_u64(uint64_t seed, uint64_t state[4]);
uint64_t ora_xoshiro_next_u64(uint64_t state[4]);
float ora_xoshiro_gen_f32(uint64_t state[4]);
int32_t ora_xoshiro_gen_range_i32(uint64_t state[4], int32_t low, int32_t high);
uint64_t ora_pixel_seed(uint32_t x, uint32_t y, uint32_t frame_num);
void ora_sinf_cosf(float x, float *s, float *c);
/* SURVEY 8d instrumentation: out2 = { BVHNode::ray_hit calls, leaf hitables tested } over every
 * ora_scene_update* call since the last reset */
void ora_bvh_counters(uint64_t out2[2], int reset);
void ora_ln_array(const float *in, float *out, uint64_t n); /* f32::ln (constant_medium.rs:60) */
/* Hitable::ray_hit on list entry `index` (hitable.rs:39-65, any arm); out7 = point3, normal3, t */
int ora_hitable_ray_hit(const ora_scene *s, uint32_t index, const float origin[3],
                        const float direction[3], float time, float t_min, float t_max,
                        uint64_t state[4], float out7[7], uint32_t *material);
/* Sphere::ray_hit: returns 1 on hit; out9 = point3, normal3, t, u, v */
int ora_sphere_ray_hit(const float centre_radius[4], const float origin[3],
                       const float direction[3], float t_min, float t_max,
                       float out9[9]);
int ora_aabb_ray_hit(const float min3[3], const float max3[3],
                     const float origin[3], const float direction[3],
                     float t_min, float t_max);
float ora_schlick(float cosine, float ref_idx);
void ora_random_unit_vector(uint64_t state[4], float out3[3]);
void ora_random_in_unit_sphere(uint64_t state[4], float out3[3]);
void ora_random_in_unit_disk(uint64_t state[4], float out3[3]);
/* Camera::get_ray: out7 = origin3, direction3, time */
void ora_camera_get_ray(const float cam24[24], float s, float t,
                        uint64_t state[4], float out7[7]);
void ora_camera_new(const float lookfrom[3], const float lookat[3],
                    const float vup[3], float vfov, float aspect,
                    float aperture, float focus_dist, float time0, float time1,
                    float cam24[24]);
float ora_perlin_noise(const ora_scene *s, const float p[3]);
float ora_perlin_turb(const ora_scene *s, const float p[3]);
/* Texture::value for texture id */
void ora_texture_value(const ora_scene *s, uint32_t texture_id,
                       const float p[3], float rgb[3]);
/* Scene::ray_trace on one explicit ray; returns colour, adds to *ray_count */
void ora_ray_trace(const ora_scene *s, const float origin[3],
                   const float direction[3], float time, uint32_t max_depth,
                   uint64_t state[4], float rgb[3], uint64_t *ray_count);
/* ---- collision/spheres_soa.rs (bench-only in the reference) and whole-world closest-hit queries ----
 * ora_soa_ray_hit: SpheresSoA::new over the scene's list, then hit_scalar (lanes = 1, spheres_soa.rs:105-155), hit_sse4_1
 * (4, :161-268) or hit_avx2 (8, :274-391). out9 = point3, normal3, t, u, v; *index_out = sphere index. Returns 1 hit, 0 miss,
 * -1 when the list holds anything but spheres (the reference panics). Note the arithmetic differs from Sphere::ray_hit.
 * ora_world_ray_hit: Hitable::ray_hit on the scene's world (list or BVH as built) for one explicit ray.
 * ora_bench_ray_hit: ns per call of one such query repeated `reps` times (which: 0 world, 1 / 4 / 8 SpheresSoA variants) -- the
 * unit of the reference's own #[bench] functions (bench.rs:8-26). */
int ora_soa_ray_hit(const ora_scene *s, int lanes, const float origin[3], const float direction[3], float t_min, float t_max,
                    float out9[9], uint32_t *index_out);
int ora_world_ray_hit(const ora_scene *s, const float origin[3], const float direction[3], float time, float t_min, float t_max,
                      uint64_t state[4], float out9[9], uint32_t *index_out);
double ora_bench_ray_hit(const ora_scene *s, int which, const float origin[3], const float direction[3], float time, uint64_t reps);
/* math.rs:36-48 */
void ora_linear_to_srgb(const float rgb[3], uint8_t out[3]);
/* offline.rs:43-51: full frame -> top-down RGB8 */
void ora_frame_to_srgb8(const float *buffer, uint32_t width, uint32_t height,
                        uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
