// scene.cpp -- Params, Camera, Perlin, Storage, BVH build and the Scene handle.
#include <algorithm>
#include <cstring>
#include <stdexcept>

#include "host.hpp"

namespace pt {

// ---- params.rs:21-27 -------------------------------------------------------
Xoshiro256Plus Params::new_rng() const {
    if (random_seed) {
        // rand::random(): OS entropy in the reference; not reproducible by design
        uint64_t seed = 0x9e3779b97f4a7c15ULL;
        if (FILE *f = fopen("/dev/urandom", "rb")) {
            if (fread(&seed, sizeof seed, 1, f) != 1) seed ^= reinterpret_cast<uintptr_t>(&seed);
            fclose(f);
        }
        return Xoshiro256Plus::seed_from_u64(seed);
    }
    return Xoshiro256Plus::seed_from_u64(0);
}

pt_params Params::c_params() const {
    return pt_params{width, height, samples, max_depth, random_seed ? 1u : 0u, use_bvh ? 1u : 0u};
}

// ---- camera.rs:22-54 -------------------------------------------------------
static void put(float dst[3], Vec3 v) {
    dst[0] = v.x;
    dst[1] = v.y;
    dst[2] = v.z;
}

Camera Camera::create(Vec3 lookfrom, Vec3 lookat, Vec3 vup, float vfov, float aspect, float aperture,
                      float focus_dist, float time0, float time1) {
    constexpr float kPi = 3.14159274101257324f;
    const float theta = vfov * kPi / 180.0f;
    const float half_height = std::tan(theta * 0.5f);
    const float half_width = aspect * half_height;
    const Vec3 w = normalize(lookfrom - lookat);
    const Vec3 u = normalize(cross(vup, w));
    const Vec3 v = cross(w, u);
    Camera c{};
    put(c.pod.origin, lookfrom);
    put(c.pod.lower_left_corner,
        lookfrom - half_width * focus_dist * u - half_height * focus_dist * v - focus_dist * w);
    put(c.pod.horizontal, 2.0f * half_width * focus_dist * u);
    put(c.pod.vertical, 2.0f * half_height * focus_dist * v);
    put(c.pod.u, u);
    put(c.pod.v, v);
    put(c.pod.w, w);
    c.pod.time0 = time0;
    c.pod.time1 = time1;
    c.pod.lens_radius = aperture * 0.5f;
    return c;
}

// ---- perlin.rs:15-51 -------------------------------------------------------
static void generate_perm(Xoshiro256Plus &rng, uint32_t *perm) {
    for (uint32_t i = 0; i < 256; ++i) perm[i] = i;
    for (int i = 255; i >= 0; --i) {  // perlin.rs:27-32
        const size_t target = static_cast<size_t>(std::floor(rng.gen_f32() * static_cast<float>(i + 1)));
        std::swap(perm[i], perm[target]);
    }
}

Perlin::Perlin(Xoshiro256Plus &rng) {
    for (auto &v : randvec) {  // perlin.rs:15-25
        const float a = -1.0f + 2.0f * rng.gen_f32();
        const float b = -1.0f + 2.0f * rng.gen_f32();
        const float c = -1.0f + 2.0f * rng.gen_f32();
        v = normalize(Vec3(a, b, c));
    }
    generate_perm(rng, perm_x);
    generate_perm(rng, perm_y);
    generate_perm(rng, perm_z);
}

// ---- storage.rs:45-96 ------------------------------------------------------
TextureId Storage::alloc_constant(Vec3 color) {
    textures.push_back(pt_texture{PT_TEX_CONSTANT, {color.x, color.y, color.z}, -1, -1, 0.0f});
    return static_cast<TextureId>(textures.size() - 1);
}
TextureId Storage::alloc_checker(TextureId odd, TextureId even) {
    textures.push_back(pt_texture{PT_TEX_CHECKER, {0.f, 0.f, 0.f}, odd, even, 0.0f});
    return static_cast<TextureId>(textures.size() - 1);
}
TextureId Storage::alloc_noise(float scale) {
    uses_noise = true;
    textures.push_back(pt_texture{PT_TEX_NOISE, {0.f, 0.f, 0.f}, -1, -1, scale});
    return static_cast<TextureId>(textures.size() - 1);
}
uint32_t Storage::alloc_image(uint32_t width, uint32_t height, const uint8_t *rgb) {
    if (!width || !height || !rgb) throw std::runtime_error("RgbImage: empty image");
    images.push_back(Image{width, height, std::vector<uint8_t>(rgb, rgb + 3ull * width * height)});
    return static_cast<uint32_t>(images.size() - 1);
}
TextureId Storage::alloc_rgb_image(uint32_t image) {
    if (image >= images.size()) throw std::runtime_error("rgb_image: no such image");
    textures.push_back(pt_texture{PT_TEX_IMAGE, {0.f, 0.f, 0.f}, static_cast<int32_t>(image), -1, 0.0f});
    return static_cast<TextureId>(textures.size() - 1);
}
MaterialId Storage::alloc_lambertian(TextureId albedo) {
    materials.push_back(pt_material{PT_MAT_LAMBERTIAN, {0.f, 0.f, 0.f}, 0.0f, albedo});
    return static_cast<MaterialId>(materials.size() - 1);
}
MaterialId Storage::alloc_metal(Vec3 albedo, float fuzz) {
    materials.push_back(pt_material{PT_MAT_METAL, {albedo.x, albedo.y, albedo.z}, fuzz, -1});
    return static_cast<MaterialId>(materials.size() - 1);
}
MaterialId Storage::alloc_dielectric(float ref_idx) {
    materials.push_back(pt_material{PT_MAT_DIELECTRIC, {0.f, 0.f, 0.f}, ref_idx, -1});
    return static_cast<MaterialId>(materials.size() - 1);
}
MaterialId Storage::alloc_diffuse_light(TextureId emit) {
    materials.push_back(pt_material{PT_MAT_DIFFUSE_LIGHT, {0.f, 0.f, 0.f}, 0.0f, emit});
    return static_cast<MaterialId>(materials.size() - 1);
}
uint32_t Storage::alloc_sphere(Vec3 centre, float radius) {
    spheres.push_back(pt_sphere{centre.x, centre.y, centre.z, radius});
    return static_cast<uint32_t>(spheres.size() - 1);
}

// ---- Hitable constructors ---------------------------------------------------
namespace {
Hitable blank(uint32_t kind, MaterialId material) {
    Hitable h{};
    h.kind = kind;
    h.material = material;
    h.transform = -1;
    h.medium_material = -1;
    return h;
}
}  // namespace

Hitable Storage::sphere(Vec3 centre, float radius, MaterialId material) {
    alloc_sphere(centre, radius);  // Hitable::Sphere(storage.alloc_sphere(..), storage.alloc_material(..))
    Hitable h = blank(PT_HIT_SPHERE, material);
    h.p[0] = centre.x, h.p[1] = centre.y, h.p[2] = centre.z, h.p[3] = radius;
    return h;
}
Hitable Storage::moving_sphere(Vec3 centre0, Vec3 centre1, float time0, float time1, float radius, MaterialId material) {
    Hitable h = blank(PT_HIT_MOVING_SPHERE, material);
    const Vec3 delta = centre1 - centre0;  // moving_sphere.rs:21
    h.p[0] = centre0.x, h.p[1] = centre0.y, h.p[2] = centre0.z;
    h.p[3] = delta.x, h.p[4] = delta.y, h.p[5] = delta.z;
    h.p[6] = radius, h.p[7] = time0, h.p[8] = 1.0f / (time1 - time0);  // moving_sphere.rs:24
    return h;
}
namespace {
Hitable rect(uint32_t kind, float a0, float a1, float b0, float b1, float k, bool flip, MaterialId material) {
    Hitable h = blank(kind, material);
    h.flip_normals = flip ? 1u : 0u;
    h.p[0] = a0, h.p[1] = a1, h.p[2] = b0, h.p[3] = b1, h.p[4] = k;
    return h;
}
}  // namespace
Hitable Storage::rect_xy(float x0, float x1, float y0, float y1, float k, bool flip, MaterialId m) { return rect(PT_HIT_RECT_XY, x0, x1, y0, y1, k, flip, m); }
Hitable Storage::rect_xz(float x0, float x1, float z0, float z1, float k, bool flip, MaterialId m) { return rect(PT_HIT_RECT_XZ, x0, x1, z0, z1, k, flip, m); }
Hitable Storage::rect_yz(float y0, float y1, float z0, float z1, float k, bool flip, MaterialId m) { return rect(PT_HIT_RECT_YZ, y0, y1, z0, z1, k, flip, m); }
Hitable Storage::cuboid(Vec3 p0, Vec3 p1, MaterialId material) {
    Hitable h = blank(PT_HIT_CUBOID, material);
    h.p[0] = p0.x, h.p[1] = p0.y, h.p[2] = p0.z, h.p[3] = p1.x, h.p[4] = p1.y, h.p[5] = p1.z;
    return h;
}
Hitable Storage::instance(const Hitable &child, const Affine3A &transform) {
    if (child.transform >= 0 || child.medium_material >= 0)
        throw std::runtime_error("Instance: only Instance(shape) nesting is supported");
    const Affine3A inv = transform.inverse();  // instance.rs:20
    pt_affine a{};
    const Affine3A *src[2] = {&transform, &inv};
    float *dst[2] = {a.m, a.inv};
    for (int k = 0; k < 2; ++k) {
        const Vec3 cols[4] = {src[k]->x_axis, src[k]->y_axis, src[k]->z_axis, src[k]->translation};
        for (int j = 0; j < 4; ++j) dst[k][3 * j] = cols[j].x, dst[k][3 * j + 1] = cols[j].y, dst[k][3 * j + 2] = cols[j].z;
    }
    transforms.push_back(a);
    Hitable h = child;
    h.transform = static_cast<int32_t>(transforms.size() - 1);
    return h;
}
Hitable Storage::constant_medium(const Hitable &child, float density, TextureId albedo) {
    if (child.medium_material >= 0) throw std::runtime_error("ConstantMedium: nested media are not supported");
    phase_functions.push_back(pt_material{PT_MAT_ISOTROPIC, {0.f, 0.f, 0.f}, 0.0f, albedo});  // material.rs:37-39
    Hitable h = child;
    h.medium_material = static_cast<int32_t>(phase_functions.size() - 1);  // rebased by Scene::new_scene
    h.density = density;
    return h;
}

// ---- hitable.rs:25-36 bounding boxes (t0 = t1 = 0, bvh.rs:69-70) ---------------
void bounding_box(const Storage &storage, const Hitable &h, Vec3 &mn, Vec3 &mx) {
    if (h.transform >= 0) {
        // instance.rs:24-30 -> AABB::transform (aabb.rs:75-100) starts from min = max = w_axis and never
        // reads the child's box: the result is one point. ConstantMedium forwards it (constant_medium.rs:28-30).
        const float *m = storage.transforms[h.transform].m;
        const Vec3 x(m[0], m[1], m[2]), y(m[3], m[4], m[5]), z(m[6], m[7], m[8]), t(m[9], m[10], m[11]);
        Vec3 out = t;
        out = out + x * t;
        out = out + y * t;
        out = out + z * t;
        mn = mx = out;
        return;
    }
    const float *p = h.p;
    switch (h.kind) {
    case PT_HIT_SPHERE: {  // sphere.rs:69-75
        const Vec3 c(p[0], p[1], p[2]), r = Vec3::splat(p[3]);
        mn = c - r, mx = c + r;
        return;
    }
    case PT_HIT_MOVING_SPHERE: {  // moving_sphere.rs:76-89 at t0 = t1 = 0
        const float s = (0.0f - p[7]) * p[8];
        const Vec3 c = Vec3(p[0], p[1], p[2]) + s * Vec3(p[3], p[4], p[5]), r = Vec3::splat(p[6]);
        mn = vmin(c - r, c - r), mx = vmax(c + r, c + r);
        return;
    }
    case PT_HIT_RECT_XY: mn = Vec3(p[0], p[2], p[4] - 0.0001f), mx = Vec3(p[1], p[3], p[4] + 0.0001f); return;  // rect.rs:195-206
    case PT_HIT_RECT_XZ: mn = Vec3(p[0], p[4] - 0.0001f, p[2]), mx = Vec3(p[1], p[4] + 0.0001f, p[3]); return;  // rect.rs:207-218
    case PT_HIT_RECT_YZ: mn = Vec3(p[4] - 0.0001f, p[0], p[2]), mx = Vec3(p[4] - 0.0001f, p[1], p[3]); return;  // rect.rs:219-228 (sic)
    default: mn = Vec3(p[0], p[1], p[2]), mx = Vec3(p[3], p[4], p[5]); return;                                   // cuboid.rs:39-41
    }
}

// ---- bvh.rs:64-94,268-347 --------------------------------------------------
namespace {

struct Box {
    Vec3 min, max;
};
Box unite(const Box &a, const Box &b) { return Box{vmin(a.min, b.min), vmax(a.max, b.max)}; }  // aabb.rs:61-66

// A build-time Hitable: either a leaf sphere (ref < 0 => ~sphere index) or a BVHNode index.
struct BuildRef {
    int32_t ref;
    Box box;
};

class BvhBuilder {
public:
    BvhBuilder(Xoshiro256Plus &rng, BvhBuild &out) : rng_(rng), out_(out) {}

    // bvh.rs:268-283; sort_unstable_by's tie order is unspecified, a stable sort is used here
    void sort_by_axis(BuildRef *v, size_t n) {
        const int axis = rng_.gen_range(0, 3);
        std::stable_sort(v, v + n, [axis](const BuildRef &l, const BuildRef &r) { return l.box.min[axis] < r.box.min[axis]; });
    }
    int32_t alloc(const BuildRef &lhs, const BuildRef &rhs, const Box &box) {  // bvh.rs:335-347
        pt_bvh_node n{};
        n.min[0] = box.min.x, n.min[1] = box.min.y, n.min[2] = box.min.z;
        n.max[0] = box.max.x, n.max[1] = box.max.y, n.max[2] = box.max.z;
        n.lhs = lhs.ref;
        n.rhs = rhs.ref;
        out_.nodes.push_back(n);
        return static_cast<int32_t>(out_.nodes.size() - 1);
    }
    BuildRef new_node(BuildRef *v, size_t n) {  // bvh.rs:298-313
        const size_t pivot = n / 2;
        const BuildRef lhs = new_split(v, pivot);
        const BuildRef rhs = new_split(v + pivot, n - pivot);
        const Box box = unite(lhs.box, rhs.box);
        return BuildRef{alloc(lhs, rhs, box), box};
    }
    BuildRef new_split(BuildRef *v, size_t n) {  // bvh.rs:315-333
        sort_by_axis(v, n);
        if (n == 1) return v[0];
        if (n == 2) {
            const Box box = unite(v[0].box, v[1].box);
            return BuildRef{alloc(v[0], v[1], box), box};
        }
        return new_node(v, n);
    }

private:
    Xoshiro256Plus &rng_;
    BvhBuild &out_;
};

uint32_t depth_of(const std::vector<pt_bvh_node> &nodes, int32_t ref) {
    if (ref < 0) return 0;
    return 1 + std::max(depth_of(nodes, nodes[ref].lhs), depth_of(nodes, nodes[ref].rhs));
}

}  // namespace

BvhBuild build_bvh(Xoshiro256Plus &rng, const Storage &storage, const std::vector<Hitable> &hitables) {
    BvhBuild out;
    std::vector<BuildRef> refs;
    refs.reserve(hitables.size());
    for (size_t i = 0; i < hitables.size(); ++i) {  // leaves are numbered by LIST position
        Box b;
        bounding_box(storage, hitables[i], b.min, b.max);
        refs.push_back(BuildRef{~static_cast<int32_t>(i), b});
    }
    BvhBuilder b(rng, out);
    const size_t n = refs.size();
    if (n == 0) return out;  // bvh.rs:72
    if (n == 1) {            // bvh.rs:73-79 lhs == rhs
        out.root = b.alloc(refs[0], refs[0], refs[0].box);
    } else if (n == 2) {     // bvh.rs:80-88
        out.root = b.alloc(refs[0], refs[1], unite(refs[0].box, refs[1].box));
    } else {                 // bvh.rs:89-92 new_root
        b.sort_by_axis(refs.data(), n);
        out.root = b.new_node(refs.data(), n).ref;
    }
    out.max_depth = depth_of(out.nodes, out.root);
    return out;
}

// ---- scene.rs:18-31, params.rs:29-46 ---------------------------------------
Scene::~Scene() {
    if (handle_) pt_scene_destroy(handle_);
}

std::unique_ptr<Scene> Scene::new_scene(const Params &params, Xoshiro256Plus &rng, const Storage &storage,
                                        const std::vector<Hitable> &hitables, std::optional<Vec3> sky,
                                        int device) {
    std::unique_ptr<Scene> s(new Scene());
    // the world list in HitableList order (hitable_list.rs:13-16)
    s->hitables_ = hitables;
    s->materials_ = storage.materials;
    const int32_t phase_base = static_cast<int32_t>(s->materials_.size());
    s->materials_.insert(s->materials_.end(), storage.phase_functions.begin(), storage.phase_functions.end());
    bool all_spheres = true;
    for (Hitable &h : s->hitables_) {
        if (h.medium_material >= 0) h.medium_material += phase_base;
        if (h.kind != PT_HIT_SPHERE || h.transform >= 0 || h.medium_material >= 0) all_spheres = false;
    }
    if (s->hitables_.empty()) all_spheres = false;  // an empty world is traced (to the sky) by the general kernel
    s->image_store_ = storage.images;
    for (const Storage::Image &im : s->image_store_) s->images_.push_back(pt_image{im.width, im.height, im.rgb.data()});
    if (!s->images_.empty()) all_spheres = false;   // Image textures travel in pt_world_desc (the library folds them for sphere worlds)
    s->is_world_ = !all_spheres;
    s->transforms_ = storage.transforms;
    s->textures_ = storage.textures;
    if (storage.uses_noise) {
        s->perlin_.reset(new pt_perlin());
        for (int i = 0; i < 256; ++i) {
            s->perlin_->randvec[i][0] = storage.perlin_noise.randvec[i].x;
            s->perlin_->randvec[i][1] = storage.perlin_noise.randvec[i].y;
            s->perlin_->randvec[i][2] = storage.perlin_noise.randvec[i].z;
            s->perlin_->perm_x[i] = storage.perlin_noise.perm_x[i];
            s->perlin_->perm_y[i] = storage.perlin_noise.perm_y[i];
            s->perlin_->perm_z[i] = storage.perlin_noise.perm_z[i];
        }
    }
    int32_t root = -1;
    if (params.use_bvh) {  // params.rs:36-40
        if (hitables.empty()) throw std::runtime_error("BVHNode::new on an empty list (params.rs:37 unwraps None)");
        BvhBuild bvh = build_bvh(rng, storage, hitables);
        s->bvh_nodes_ = std::move(bvh.nodes);
        root = bvh.root;
    }
    pt_world_desc &w = s->world_;
    w.n_hitables = static_cast<uint32_t>(s->hitables_.size());
    w.hitables = s->hitables_.empty() ? nullptr : s->hitables_.data();
    w.n_transforms = static_cast<uint32_t>(s->transforms_.size());
    w.transforms = s->transforms_.empty() ? nullptr : s->transforms_.data();
    w.n_materials = static_cast<uint32_t>(s->materials_.size());
    w.materials = s->materials_.empty() ? nullptr : s->materials_.data();
    w.n_textures = static_cast<uint32_t>(s->textures_.size());
    w.textures = s->textures_.data();
    w.perlin = s->perlin_.get();
    w.n_bvh_nodes = static_cast<uint32_t>(s->bvh_nodes_.size());
    w.bvh_nodes = s->bvh_nodes_.empty() ? nullptr : s->bvh_nodes_.data();
    w.bvh_root = root;
    w.has_sky = sky.has_value() ? 1u : 0u;
    if (sky) w.sky[0] = sky->x, w.sky[1] = sky->y, w.sky[2] = sky->z;
    w.n_images = static_cast<uint32_t>(s->images_.size());
    w.images = s->images_.empty() ? nullptr : s->images_.data();
    if (all_spheres) {  // the sphere-only description (the specialised kernels' entry point)
        s->spheres_.reserve(hitables.size());
        s->sphere_material_.reserve(hitables.size());
        for (const Hitable &h : s->hitables_) {
            s->spheres_.push_back(pt_sphere{h.p[0], h.p[1], h.p[2], h.p[3]});
            s->sphere_material_.push_back(h.material);
        }
        pt_scene_desc &d = s->desc_;
        d.n_spheres = static_cast<uint32_t>(s->spheres_.size());
        d.spheres = s->spheres_.data();
        d.sphere_material = s->sphere_material_.data();
        d.n_materials = w.n_materials, d.materials = w.materials;
        d.n_textures = w.n_textures, d.textures = w.textures;
        d.perlin = w.perlin;
        d.n_bvh_nodes = w.n_bvh_nodes, d.bvh_nodes = w.bvh_nodes, d.bvh_root = root;
        d.has_sky = w.has_sky;
        memcpy(d.sky, w.sky, sizeof d.sky);
    }
    if (device >= 0) {
        const int rc = all_spheres ? pt_scene_create(&s->desc_, device, &s->handle_) : pt_scene_create_world(&w, device, &s->handle_);
        if (rc != PT_OK) throw std::runtime_error(std::string(all_spheres ? "pt_scene_create: " : "pt_scene_create_world: ") + pt_last_error());
        // still Scene::new: size the per-frame buffers now, so Scene::update's timer sees rendering only
        const pt_params p = params.c_params();
        if (pt_scene_prepare(s->handle_, &p) != PT_OK) throw std::runtime_error(std::string("pt_scene_prepare: ") + pt_last_error());
    }
    return s;
}

size_t Scene::update(const Params &params, const Camera &camera, uint32_t frame_num, float *buffer) {
    if (!handle_) throw std::runtime_error("Scene::update: scene was built without a device (inspection only)");
    const pt_params p = params.c_params();
    uint64_t rays = 0;
    const int rc = pt_render(handle_, &p, &camera.pod, frame_num, buffer, &rays);
    if (rc != PT_OK) throw std::runtime_error(std::string("pt_render: ") + pt_last_error());
    return static_cast<size_t>(rays);
}

float Scene::last_kernel_ms() const {
    float ms = 0.f;
    if (!handle_ || pt_last_kernel_ms(handle_, &ms) != PT_OK) return -1.f;
    return ms;
}

}  // namespace pt
