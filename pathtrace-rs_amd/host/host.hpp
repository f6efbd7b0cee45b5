// host.hpp -- C++ host mirroring the reference's Scene / Camera / Params /
// presets / offline API for the one path that calls into libptgpu.so.
//
// The reference host is Rust (no toolchain in this image), so the host above
// the C ABI is C++ with the same names, argument meaning and error behaviour:
//   Params            params.rs:11-46
//   Camera            camera.rs:8-54   (get_ray runs on the GPU)
//   Storage / Perlin  storage.rs:12-43, perlin.rs:7-51
//   presets::from_name presets.rs:13-38
//   Scene             scene.rs:18-31, Scene::update scene.rs:73-121 -> pt_render
//   render_offline    offline.rs:16-60
#pragma once
#include <cstdint>
#include <memory>
#include <optional>
#include <string>
#include <tuple>
#include <vector>

#include "ptgpu.h"
#include "vecmath.hpp"

namespace pt {

// params.rs:11-18
struct Params {
    uint32_t width = 1280;
    uint32_t height = 720;
    uint32_t samples = 4;
    uint32_t max_depth = 10;
    bool random_seed = false;
    bool use_bvh = false;

    Xoshiro256Plus new_rng() const;  // params.rs:21-27
    pt_params c_params() const;
};

// camera.rs:8-19; the POD handed to the kernels
struct Camera {
    pt_camera pod;
    // camera.rs:22-54
    static Camera create(Vec3 lookfrom, Vec3 lookat, Vec3 vup, float vfov, float aspect, float aperture,
                         float focus_dist, float time0, float time1);
};

// perlin.rs:7-12
struct Perlin {
    Vec3 randvec[256];
    uint32_t perm_x[256], perm_y[256], perm_z[256];
    explicit Perlin(Xoshiro256Plus &rng);  // perlin.rs:43-51
};

enum class TextureKind : uint32_t { Constant = PT_TEX_CONSTANT, Checker = PT_TEX_CHECKER, Noise = PT_TEX_NOISE, Image = PT_TEX_IMAGE };
enum class MaterialKind : uint32_t {
    Lambertian = PT_MAT_LAMBERTIAN,
    Metal = PT_MAT_METAL,
    Dielectric = PT_MAT_DIELECTRIC,
    DiffuseLight = PT_MAT_DIFFUSE_LIGHT,
    Isotropic = PT_MAT_ISOTROPIC
};

using TextureId = int32_t;
using MaterialId = uint32_t;

// One entry of the world list (hitable.rs:12-21), flattened to the nesting the reference's presets build:
// the innermost shape (Sphere / MovingSphere / Rect / Cuboid with its &Material), optionally inside an
// Instance, optionally inside a ConstantMedium. The POD is the C-ABI record; `medium_material` holds the
// index into Storage::phase_functions until Scene::new_scene appends those after the arena materials.
using Hitable = pt_hitable;

// storage.rs:12-43 -- typed arenas become index-addressed vectors
class Storage {
public:
    explicit Storage(Xoshiro256Plus &rng) : perlin_noise(rng) {}  // storage.rs:28-43 (Perlin::new draws 1536 f32)

    TextureId alloc_constant(Vec3 color);                   // texture.rs:57-59
    TextureId alloc_checker(TextureId odd, TextureId even);  // texture.rs:61-63
    TextureId alloc_noise(float scale);                      // texture.rs:65-67 (&storage.perlin_noise)
    // texture.rs:5-25,69-71: an RgbImage as image::open(..).to_rgb8().into_raw() yields it (no decoder here: the
    // caller supplies the RGB8 rows), and the Texture::Image over it
    uint32_t alloc_image(uint32_t width, uint32_t height, const uint8_t *rgb);
    TextureId alloc_rgb_image(uint32_t image);
    MaterialId alloc_lambertian(TextureId albedo);           // material.rs:21-23
    MaterialId alloc_metal(Vec3 albedo, float fuzz);         // material.rs:25-27
    MaterialId alloc_dielectric(float ref_idx);              // material.rs:29-31
    MaterialId alloc_diffuse_light(TextureId emit);          // material.rs:33-35
    uint32_t alloc_sphere(Vec3 centre, float radius);        // sphere.rs:15-17

    // Hitable constructors (the closures of presets.rs:115-127 and the Hitable::X(...) expressions)
    Hitable sphere(Vec3 centre, float radius, MaterialId material);                           // Hitable::Sphere
    Hitable moving_sphere(Vec3 centre0, Vec3 centre1, float time0, float time1, float radius,
                          MaterialId material);                                               // moving_sphere.rs:18-26
    Hitable rect_xy(float x0, float x1, float y0, float y1, float k, bool flip, MaterialId material);  // rect.rs:37-46
    Hitable rect_xz(float x0, float x1, float z0, float z1, float k, bool flip, MaterialId material);  // rect.rs:49-58
    Hitable rect_yz(float y0, float y1, float z0, float z1, float k, bool flip, MaterialId material);  // rect.rs:61-70
    Hitable cuboid(Vec3 p0, Vec3 p1, MaterialId material);                                    // cuboid.rs:11-23
    Hitable instance(const Hitable &child, const Affine3A &transform);                        // instance.rs:16-22
    Hitable constant_medium(const Hitable &child, float density, TextureId albedo);           // constant_medium.rs:18-26

    std::vector<pt_texture> textures;
    std::vector<pt_material> materials;
    std::vector<pt_material> phase_functions;  // one Isotropic per ConstantMedium (constant_medium.rs:13)
    std::vector<pt_affine> transforms;         // Instance { transform, inv_transform }
    struct Image {
        uint32_t width, height;
        std::vector<uint8_t> rgb;
    };
    std::vector<Image> images;                 // RgbImage arena
    std::vector<pt_sphere> spheres;
    Perlin perlin_noise;
    bool uses_noise = false;
};

struct PresetResult {
    std::vector<Hitable> hitables;
    Camera camera;
    std::optional<Vec3> sky;
};

namespace presets {
// presets.rs:13-38; prints the banner of presets.rs:19-22 unless quiet
std::optional<PresetResult> from_name(const std::string &name, const Params &params, Xoshiro256Plus &rng,
                                      Storage &storage, bool quiet = false);
std::vector<std::string> names();
}  // namespace presets

// bvh.rs:64-94,268-347 host-side build, flattened for the device
struct BvhBuild {
    std::vector<pt_bvh_node> nodes;
    int32_t root = -1;
    uint32_t max_depth = 0;
};
BvhBuild build_bvh(Xoshiro256Plus &rng, const Storage &storage, const std::vector<Hitable> &hitables);
// Hitable::bounding_box(0, 0) (hitable.rs:25-36) including the reference's quirks: AABB::transform ignores
// `self` (aabb.rs:75-100), the YZ rect box is flat at k - 0.0001 (rect.rs:225-226), moving spheres are
// boxed at t = 0 only (bvh.rs:69-70).
void bounding_box(const Storage &storage, const Hitable &h, Vec3 &min_out, Vec3 &max_out);

// scene.rs:18-31 + Params::new_scene (params.rs:29-46)
class Scene {
public:
    ~Scene();
    Scene(const Scene &) = delete;
    Scene &operator=(const Scene &) = delete;

    // Params::new_scene: List, or BVH when params.use_bvh. Throws std::runtime_error
    // with pt_last_error() when the device library refuses (the reference unwrap()s).
    static std::unique_ptr<Scene> new_scene(const Params &params, Xoshiro256Plus &rng, const Storage &storage,
                                            const std::vector<Hitable> &hitables, std::optional<Vec3> sky,
                                            int device = 0);

    // Scene::update (scene.rs:73-121): buffer is width*height (r,g,b) float triples, read and written.
    size_t update(const Params &params, const Camera &camera, uint32_t frame_num, float *buffer);

    pt_scene *handle() const { return handle_; }
    const pt_scene_desc &desc() const { return desc_; }   // valid when !is_world()
    const pt_world_desc &world_desc() const { return world_; }  // always filled
    bool is_world() const { return is_world_; }            // any non-sphere hitable: traced by the general kernel
    float last_kernel_ms() const;

private:
    Scene() = default;
    pt_scene *handle_ = nullptr;
    // flattened description kept for inspection (C API / tests)
    std::vector<pt_sphere> spheres_;
    std::vector<uint32_t> sphere_material_;
    std::vector<pt_material> materials_;
    std::vector<pt_texture> textures_;
    std::unique_ptr<pt_perlin> perlin_;
    std::vector<pt_bvh_node> bvh_nodes_;
    std::vector<pt_hitable> hitables_;
    std::vector<pt_affine> transforms_;
    std::vector<Storage::Image> image_store_;
    std::vector<pt_image> images_;
    pt_scene_desc desc_{};
    pt_world_desc world_{};
    bool is_world_ = false;
    friend struct SceneAccess;
};

// math.rs:36-48
void linear_to_srgb(const float rgb[3], uint8_t out[3]);
// offline.rs:43-59: sRGB + vertical flip + RGB8 PNG
bool save_png(const std::string &path, const float *buffer, uint32_t width, uint32_t height);
// offline.rs:16-60; returns 0, or non-zero after printing the error (the reference panics)
int render_offline(const std::string &preset, const Params &params, int device = 0,
                   const std::string &output = "output.png", uint32_t frames = 1);

}  // namespace pt
