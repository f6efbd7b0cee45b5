// presets.cpp -- scene presets of the hot-path scope:
//   small               presets.rs:217-269
//   random_spheres      presets.rs:81-215 (random_impl, only_spheres = true)
//   two_perlin_spheres  presets.rs:271-315
//   aras                presets.rs:595-851 (commented out upstream; reconstructed)
//   perlin_spheres      builder-defined 10k-sphere scene for BASELINE config 5
//   random              presets.rs:81-215 (only_spheres = false: MovingSphere for the lambertians)
//   simple_light        presets.rs:317-370 (Rect + DiffuseLight)
//   cornell             presets.rs:372-456 (Rects, Instance(Cuboid))
//   cornell_smoke       presets.rs:458-552 (ConstantMedium(Instance(Cuboid)))
//   smallpt             presets.rs:853-930
//   final               presets.rs:40-71 (returns an EMPTY list: sky only; with -B the reference panics)
// `earth` needs media/earthmap.jpg (absent upstream).
#include <cstdio>

#include "host.hpp"

namespace pt {
namespace presets {
namespace {

// the `sphere` closure (presets.rs:115-120): sphere arena first, then material arena
struct SceneBuilder {
    Storage &st;
    std::vector<Hitable> hitables;
    void sphere(Vec3 centre, float radius, MaterialId material) { hitables.push_back(st.sphere(centre, radius, material)); }
    void push(const Hitable &h) { hitables.push_back(h); }
};

float aspect_of(const Params &p) { return static_cast<float>(p.width) / static_cast<float>(p.height); }

// presets.rs:89-215 random_impl
PresetResult random_impl(const Params &params, bool only_spheres, Xoshiro256Plus &rng, Storage &st) {
    const Camera camera = Camera::create(Vec3(13.0f, 2.0f, 3.0f), Vec3(0.0f, 0.0f, 0.0f), Vec3(0.0f, 1.0f, 0.0f), 20.0f,
                                         aspect_of(params), 0.1f, 10.0f, 0.0f, 1.0f);
    SceneBuilder b{st, {}};
    b.hitables.reserve(501);
    {  // presets.rs:132-139 ground: checker(constant, constant)
        const TextureId odd = st.alloc_constant(Vec3(0.2f, 0.3f, 0.1f));
        const TextureId even = st.alloc_constant(Vec3(0.9f, 0.9f, 0.9f));
        const MaterialId ground = st.alloc_lambertian(st.alloc_checker(odd, even));
        // NOTE arena order in the reference closure: Sphere is allocated before the Material.
        // Indices are only labels here, so the material may be created first.
        b.sphere(Vec3(0.0f, -1000.0f, 0.0f), 1000.0f, ground);
    }
    for (int a = -11; a < 11; ++a) {
        for (int bb = -11; bb < 11; ++bb) {
            const float choose_material = rng.gen_f32();
            const float cx = static_cast<float>(a) + 0.9f * rng.gen_f32();
            const float cz = static_cast<float>(bb) + 0.9f * rng.gen_f32();
            const Vec3 centre(cx, 0.2f, cz);
            if (choose_material < 0.8f) {
                const Vec3 centre1 = centre + Vec3(0.0f, 0.5f * rng.gen_f32(), 0.0f);  // presets.rs:150, drawn either way
                const float r0 = rng.gen_f32(), r1 = rng.gen_f32();
                const float r2 = rng.gen_f32(), r3 = rng.gen_f32();
                const float r4 = rng.gen_f32(), r5 = rng.gen_f32();
                const MaterialId m = st.alloc_lambertian(st.alloc_constant(Vec3(r0 * r1, r2 * r3, r4 * r5)));
                if (only_spheres) b.sphere(centre, 0.2f, m);
                else b.push(st.moving_sphere(centre, centre1, 0.0f, 1.0f, 0.2f, m));  // presets.rs:122-127,162-171
            } else if (choose_material < 0.95f) {
                const float ax = 0.5f * (1.0f + rng.gen_f32());
                const float ay = 0.5f * (1.0f + rng.gen_f32());
                const float az = 0.5f * (1.0f + rng.gen_f32());
                const float fuzz = 0.5f * rng.gen_f32();
                b.sphere(centre, 0.2f, st.alloc_metal(Vec3(ax, ay, az), fuzz));
            } else {
                b.sphere(centre, 0.2f, st.alloc_dielectric(1.5f));
            }
        }
    }
    b.sphere(Vec3(0.0f, 1.0f, 0.0f), 1.0f, st.alloc_dielectric(1.5f));
    b.sphere(Vec3(-4.0f, 1.0f, 0.0f), 1.0f, st.alloc_lambertian(st.alloc_constant(Vec3(0.4f, 0.2f, 0.1f))));
    b.sphere(Vec3(4.0f, 1.0f, 0.0f), 1.0f, st.alloc_metal(Vec3(0.7f, 0.6f, 0.5f), 0.0f));
    return PresetResult{std::move(b.hitables), camera, std::nullopt};
}

PresetResult simple_light(const Params &params, Storage &st) {  // presets.rs:317-370
    const Camera camera = Camera::create(Vec3(50.0f, 2.0f, 3.0f), Vec3(0.0f, 0.0f, 0.0f), Vec3(0.0f, 1.0f, 0.0f), 20.0f,
                                         aspect_of(params), 0.0f, 10.0f, 0.0f, 0.0f);
    SceneBuilder b{st, {}};
    const TextureId noise_texture = st.alloc_noise(4.0f);
    const TextureId constant_texture = st.alloc_constant(Vec3(4.0f, 4.0f, 4.0f));
    b.sphere(Vec3(0.0f, -1000.0f, 0.0f), 1000.0f, st.alloc_lambertian(noise_texture));
    b.sphere(Vec3(0.0f, 2.0f, 0.0f), 2.0f, st.alloc_lambertian(noise_texture));
    b.sphere(Vec3(0.0f, 7.0f, 0.0f), 2.0f, st.alloc_diffuse_light(constant_texture));
    b.push(st.rect_xy(3.0f, 5.0f, 1.0f, 3.0f, -2.0f, false, st.alloc_diffuse_light(constant_texture)));
    return PresetResult{std::move(b.hitables), camera, Vec3(0.0f, 0.0f, 0.0f)};
}

// presets.rs:372-456 cornell_box / 458-552 cornell_smoke
PresetResult cornell(const Params &params, Storage &st, bool smoke) {
    const Camera camera = Camera::create(Vec3(278.0f, 278.0f, -800.0f), Vec3(278.0f, 278.0f, 0.0f), Vec3(0.0f, 1.0f, 0.0f),
                                         40.0f, aspect_of(params), 0.0f, 10.0f, 0.0f, 1.0f);
    SceneBuilder b{st, {}};
    const MaterialId red = st.alloc_lambertian(st.alloc_constant(Vec3(0.65f, 0.05f, 0.05f)));
    const MaterialId white = st.alloc_lambertian(st.alloc_constant(Vec3(0.73f, 0.73f, 0.73f)));
    const MaterialId green = st.alloc_lambertian(st.alloc_constant(Vec3(0.12f, 0.45f, 0.15f)));
    const float e = smoke ? 7.0f : 15.0f;
    const MaterialId light = st.alloc_diffuse_light(st.alloc_constant(Vec3(e, e, e)));
    const Affine3A box1_transform = Affine3A::from_rotation_translation(Quat::from_rotation_y(to_radians(-18.0f)), Vec3(130.0f, 0.0f, 65.0f));
    const Affine3A box2_transform = Affine3A::from_rotation_translation(Quat::from_rotation_y(to_radians(15.0f)), Vec3(265.0f, 0.0f, 295.0f));
    b.push(st.rect_yz(0.0f, 555.0f, 0.0f, 555.0f, 555.0f, true, green));
    b.push(st.rect_yz(0.0f, 555.0f, 0.0f, 555.0f, 0.0f, false, red));
    if (smoke) b.push(st.rect_xz(113.0f, 443.0f, 127.0f, 432.0f, 554.0f, false, light));
    else b.push(st.rect_xz(213.0f, 343.0f, 227.0f, 332.0f, 554.0f, false, light));
    b.push(st.rect_xz(0.0f, 555.0f, 0.0f, 555.0f, 555.0f, true, white));
    b.push(st.rect_xz(0.0f, 555.0f, 0.0f, 555.0f, 0.0f, false, white));
    b.push(st.rect_xy(0.0f, 555.0f, 0.0f, 555.0f, 555.0f, true, white));
    const Hitable box1 = st.instance(st.cuboid(Vec3(0.0f, 0.0f, 0.0f), Vec3(165.0f, 165.0f, 165.0f), white), box1_transform);
    if (smoke) b.push(st.constant_medium(box1, 0.01f, st.alloc_constant(Vec3(1.0f, 1.0f, 1.0f))));
    else b.push(box1);
    const Hitable box2 = st.instance(st.cuboid(Vec3(0.0f, 0.0f, 0.0f), Vec3(165.0f, 330.0f, 165.0f), white), box2_transform);
    if (smoke) b.push(st.constant_medium(box2, 0.01f, st.alloc_constant(Vec3(0.0f, 0.0f, 0.0f))));
    else b.push(box2);
    return PresetResult{std::move(b.hitables), camera, Vec3(0.0f, 0.0f, 0.0f)};
}

// presets.rs:40-71: camera and two textures, but the list is returned EMPTY (every ray sees the sky)
PresetResult final_scene(const Params &params, Storage &st) {
    const Camera camera = Camera::create(Vec3(13.0f, 2.0f, 3.0f), Vec3(0.0f, 0.0f, 0.0f), Vec3(0.0f, 1.0f, 0.0f), 20.0f,
                                         aspect_of(params), 0.1f, 10.0f, 0.0f, 1.0f);
    (void)st.alloc_constant(Vec3(0.73f, 0.73f, 0.73f));  // `white` and `ground` are Material values, never arena entries
    (void)st.alloc_constant(Vec3(0.48f, 0.83f, 0.53f));
    return PresetResult{{}, camera, std::nullopt};
}

PresetResult smallpt(const Params &params, Storage &st) {  // presets.rs:853-930
    const Camera camera = Camera::create(Vec3(50.0f, 52.0f, 295.6f), Vec3(50.0f, 33.0f, 0.0f), Vec3(0.0f, 1.0f, 0.0f), 30.0f,
                                         aspect_of(params), 0.05f, 100.0f, 0.0f, 1.0f);
    SceneBuilder b{st, {}};
    auto lamb = [&](Vec3 c, float r, Vec3 albedo) { b.sphere(c, r, st.alloc_lambertian(st.alloc_constant(albedo))); };
    lamb(Vec3(1e3f + 1.0f, 40.8f, 81.6f), 1e3f, Vec3(0.75f, 0.25f, 0.25f));    // Left
    lamb(Vec3(-1e3f + 99.0f, 40.8f, 81.6f), 1e3f, Vec3(0.25f, 0.25f, 0.75f));  // Rght
    lamb(Vec3(50.0f, 40.8f, 1e3f), 1e3f, Vec3(0.75f, 0.75f, 0.75f));           // Back
    lamb(Vec3(50.0f, 1e3f, 81.6f), 1e3f, Vec3(0.75f, 0.75f, 0.75f));           // Botm
    lamb(Vec3(50.0f, -1e3f + 81.6f, 81.6f), 1e3f, Vec3(0.75f, 0.75f, 0.75f));  // Top
    b.sphere(Vec3(27.0f, 16.5f, 47.0f), 16.5f, st.alloc_metal(Vec3(1.0f, 1.0f, 1.0f) * 0.999f, 0.0f));  // Mirr
    b.sphere(Vec3(73.0f, 16.5f, 78.0f), 16.5f, st.alloc_dielectric(1.5f));                              // Glas
    b.sphere(Vec3(50.0f, 81.6f - 16.5f, 81.6f), 1.5f, st.alloc_diffuse_light(st.alloc_constant(Vec3(4.0f, 4.0f, 4.0f) * 100.0f)));  // Lite
    return PresetResult{std::move(b.hitables), camera, Vec3(0.0f, 0.0f, 0.0f)};
}

PresetResult small(const Params &params, Storage &st) {
    const Vec3 lookfrom(3.0f, 3.0f, 2.0f), lookat(0.0f, 0.0f, -1.0f);
    const float dist_to_focus = length(lookfrom - lookat);
    const Camera camera = Camera::create(lookfrom, lookat, Vec3(0.0f, 1.0f, 0.0f), 20.0f, aspect_of(params), 0.1f,
                                         dist_to_focus, 0.0f, 1.0f);
    SceneBuilder b{st, {}};
    b.sphere(Vec3(0.0f, 0.0f, -1.0f), 0.5f, st.alloc_lambertian(st.alloc_constant(Vec3(0.1f, 0.2f, 0.5f))));
    b.sphere(Vec3(0.0f, -100.5f, -1.0f), 100.0f, st.alloc_lambertian(st.alloc_constant(Vec3(0.8f, 0.8f, 0.0f))));
    b.sphere(Vec3(1.0f, 0.0f, -1.0f), 0.5f, st.alloc_metal(Vec3(0.8f, 0.6f, 0.2f), 0.0f));
    b.sphere(Vec3(-1.0f, 0.0f, -1.0f), 0.5f, st.alloc_dielectric(1.5f));
    b.sphere(Vec3(-1.0f, 0.0f, -1.0f), -0.45f, st.alloc_dielectric(1.5f));  // hollow glass: negative radius
    return PresetResult{std::move(b.hitables), camera, std::nullopt};
}

PresetResult two_perlin_spheres(const Params &params, Storage &st) {
    const Camera camera = Camera::create(Vec3(13.0f, 2.0f, 3.0f), Vec3(0.0f, 0.0f, 0.0f), Vec3(0.0f, 1.0f, 0.0f), 20.0f,
                                         aspect_of(params), 0.0f, 10.0f, 0.0f, 0.0f);
    SceneBuilder b{st, {}};
    const TextureId noise_texture = st.alloc_noise(4.0f);
    b.sphere(Vec3(0.0f, -1000.0f, 0.0f), 1000.0f, st.alloc_lambertian(noise_texture));
    b.sphere(Vec3(0.0f, 2.0f, 0.0f), 2.0f, st.alloc_lambertian(noise_texture));
    return PresetResult{std::move(b.hitables), camera, std::nullopt};
}

// Reconstructed from the commented block presets.rs:595-851. That block targets an older
// Camera::new without time0/time1: both are 0 here. The two diffuse_light spheres are kept.
PresetResult aras(const Params &params, Storage &st) {
    const Camera camera = Camera::create(Vec3(0.0f, 2.0f, 3.0f), Vec3(0.0f, 0.0f, 0.0f), Vec3(0.0f, 1.0f, 0.0f), 60.0f,
                                         aspect_of(params), 0.02f, 3.0f, 0.0f, 0.0f);
    SceneBuilder b{st, {}};
    auto lamb = [&](Vec3 c, float r, Vec3 albedo) { b.sphere(c, r, st.alloc_lambertian(st.alloc_constant(albedo))); };
    auto metal = [&](Vec3 c, float r, Vec3 albedo, float fuzz) { b.sphere(c, r, st.alloc_metal(albedo, fuzz)); };
    auto light = [&](Vec3 c, float r, Vec3 emit) { b.sphere(c, r, st.alloc_diffuse_light(st.alloc_constant(emit))); };
    lamb(Vec3(0.0f, -100.5f, -1.0f), 100.0f, Vec3(0.8f, 0.8f, 0.8f));
    lamb(Vec3(2.0f, 0.0f, -1.0f), 0.5f, Vec3(0.8f, 0.4f, 0.4f));
    lamb(Vec3(0.0f, 0.0f, -1.0f), 0.5f, Vec3(0.4f, 0.8f, 0.4f));
    metal(Vec3(-2.0f, 0.0f, -1.0f), 0.5f, Vec3(0.4f, 0.4f, 0.8f), 0.0f);
    metal(Vec3(2.0f, 0.0f, 1.0f), 0.5f, Vec3(0.4f, 0.8f, 0.4f), 0.0f);
    metal(Vec3(0.0f, 0.0f, 1.0f), 0.5f, Vec3(0.4f, 0.8f, 0.4f), 0.2f);
    metal(Vec3(-2.0f, 0.0f, 1.0f), 0.5f, Vec3(0.4f, 0.8f, 0.4f), 0.6f);
    b.sphere(Vec3(0.5f, 1.0f, 0.5f), 0.5f, st.alloc_dielectric(1.5f));
    light(Vec3(-1.5f, 1.5f, 0.0f), 0.3f, Vec3(30.0f, 25.0f, 15.0f));
    const float xs[9] = {4.0f, 3.0f, 2.0f, 1.0f, 0.0f, -1.0f, -2.0f, -3.0f, -4.0f};
    const float grey[9] = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f, 0.9f};
    const Vec3 hue[9] = {Vec3(0.8f, 0.1f, 0.1f), Vec3(0.8f, 0.5f, 0.1f), Vec3(0.8f, 0.8f, 0.1f),
                         Vec3(0.4f, 0.8f, 0.1f), Vec3(0.1f, 0.8f, 0.1f), Vec3(0.1f, 0.8f, 0.5f),
                         Vec3(0.1f, 0.8f, 0.8f), Vec3(0.1f, 0.1f, 0.8f), Vec3(0.5f, 0.1f, 0.8f)};
    for (int i = 0; i < 9; ++i) lamb(Vec3(xs[i], 0.0f, -3.0f), 0.5f, Vec3::splat(grey[i]));
    for (int i = 0; i < 9; ++i) metal(Vec3(xs[i], 0.0f, -4.0f), 0.5f, Vec3::splat(grey[i]), 0.0f);
    for (int i = 0; i < 9; ++i) metal(Vec3(xs[i], 0.0f, -5.0f), 0.5f, hue[i], 0.0f);
    for (int i = 0; i < 8; ++i) lamb(Vec3(xs[i], 0.0f, -6.0f), 0.5f, hue[i]);
    metal(Vec3(-4.0f, 0.0f, -6.0f), 0.5f, hue[8], 0.0f);
    light(Vec3(1.5f, 1.5f, -2.0f), 0.3f, Vec3(3.0f, 10.0f, 20.0f));
    return PresetResult{std::move(b.hitables), camera, std::nullopt};
}

// BASELINE config 5 ("perlin_spheres" + BVH, 10k spheres). No such preset exists upstream; this
// generator is the published definition: the two spheres of two_perlin_spheres plus a 100x100
// grid of r = 0.2 spheres jittered with the scene rng, 80 % lambertian(noise) over four noise
// scales, 15 % metal, 5 % dielectric; camera pulled back to frame the grid.
PresetResult perlin_spheres(const Params &params, Xoshiro256Plus &rng, Storage &st) {
    const Camera camera = Camera::create(Vec3(26.0f, 6.0f, 6.0f), Vec3(0.0f, 0.0f, 0.0f), Vec3(0.0f, 1.0f, 0.0f), 30.0f,
                                         aspect_of(params), 0.0f, 10.0f, 0.0f, 0.0f);
    SceneBuilder b{st, {}};
    b.hitables.reserve(10002);
    const TextureId noise4 = st.alloc_noise(4.0f);
    b.sphere(Vec3(0.0f, -1000.0f, 0.0f), 1000.0f, st.alloc_lambertian(noise4));
    b.sphere(Vec3(0.0f, 2.0f, 0.0f), 2.0f, st.alloc_lambertian(noise4));
    TextureId scales[4];
    scales[0] = st.alloc_noise(1.0f);
    scales[1] = st.alloc_noise(2.0f);
    scales[2] = noise4;
    scales[3] = st.alloc_noise(8.0f);
    for (int a = -50; a < 50; ++a) {
        for (int bb = -50; bb < 50; ++bb) {
            const float choose_material = rng.gen_f32();
            const float cx = 0.5f * static_cast<float>(a) + 0.3f * rng.gen_f32();
            const float cz = 0.5f * static_cast<float>(bb) + 0.3f * rng.gen_f32();
            const Vec3 centre(cx, 0.2f, cz);
            if (choose_material < 0.8f) {
                const int k = static_cast<int>(rng.gen_f32() * 4.0f) & 3;
                b.sphere(centre, 0.2f, st.alloc_lambertian(scales[k]));
            } else if (choose_material < 0.95f) {
                const float ax = 0.5f * (1.0f + rng.gen_f32());
                const float ay = 0.5f * (1.0f + rng.gen_f32());
                const float az = 0.5f * (1.0f + rng.gen_f32());
                const float fuzz = 0.5f * rng.gen_f32();
                b.sphere(centre, 0.2f, st.alloc_metal(Vec3(ax, ay, az), fuzz));
            } else {
                b.sphere(centre, 0.2f, st.alloc_dielectric(1.5f));
            }
        }
    }
    return PresetResult{std::move(b.hitables), camera, std::nullopt};
}

}  // namespace

std::vector<std::string> names() {
    return {"small", "random_spheres", "two_perlin_spheres", "aras", "perlin_spheres", "random", "simple_light", "cornell", "cornell_smoke", "smallpt", "final"};
}

std::optional<PresetResult> from_name(const std::string &name, const Params &params, Xoshiro256Plus &rng,
                                      Storage &storage, bool quiet) {
    if (!quiet)  // presets.rs:19-22
        printf("generating '%s' preset at %ux%u with %u samples per pixel\n", name.c_str(), params.width,
               params.height, params.samples);
    if (name == "random") return random_impl(params, false, rng, storage);
    if (name == "random_spheres") return random_impl(params, true, rng, storage);
    if (name == "simple_light") return simple_light(params, storage);
    if (name == "cornell") return cornell(params, storage, false);
    if (name == "cornell_smoke") return cornell(params, storage, true);
    if (name == "smallpt") return smallpt(params, storage);
    if (name == "final") return final_scene(params, storage);
    if (name == "small") return small(params, storage);
    if (name == "two_perlin_spheres") return two_perlin_spheres(params, storage);
    if (name == "aras") return aras(params, storage);
    if (name == "perlin_spheres") return perlin_spheres(params, rng, storage);
    return std::nullopt;  // presets.rs:36
}

}  // namespace presets
}  // namespace pt
