// capi.cpp -- extern "C" view of the C++ host for ctypes callers.
#include "pthost_c.h"

#include <cstdio>
#include <cstring>
#include <stdexcept>

#include "host.hpp"

struct pth_scene {
    std::unique_ptr<pt::Scene> scene;
    pt::Camera camera;
    uint64_t build_draws = 0;
    uint32_t bvh_depth = 0;
};

namespace {
thread_local char g_err[512] = "";
void set_err(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg); }
uint32_t depth_of(const pt_world_desc &d, int32_t ref) {
    if (ref < 0) return 0;
    const uint32_t l = depth_of(d, d.bvh_nodes[ref].lhs), r = depth_of(d, d.bvh_nodes[ref].rhs);
    return 1 + (l > r ? l : r);
}
}  // namespace

extern "C" const char *pth_last_error(void) { return g_err; }

extern "C" int pth_scene_build(const char *preset, uint32_t width, uint32_t height, uint32_t samples, int use_bvh,
                               int device, int quiet, pth_scene **out) {
    if (!preset || !out) {
        set_err("NULL argument");
        return 1;
    }
    *out = nullptr;
    try {
        pt::Params params;
        params.width = width;
        params.height = height;
        params.samples = samples;
        params.use_bvh = use_bvh != 0;
        pt::Xoshiro256Plus rng = params.new_rng();
        pt::Storage storage(rng);
        auto built = pt::presets::from_name(preset, params, rng, storage, quiet != 0);
        if (!built) {
            set_err("unrecognised preset");
            return 2;
        }
        std::unique_ptr<pth_scene> s(new pth_scene());
        s->scene = pt::Scene::new_scene(params, rng, storage, built->hitables, built->sky, device);
        s->camera = built->camera;
        s->build_draws = rng.draws();
        if (s->scene->world_desc().n_bvh_nodes) s->bvh_depth = depth_of(s->scene->world_desc(), s->scene->world_desc().bvh_root);
        *out = s.release();
        return 0;
    } catch (const std::exception &e) {
        set_err(e.what());
        return 1;
    }
}

extern "C" void pth_scene_free(pth_scene *s) { delete s; }
extern "C" const pt_scene_desc *pth_scene_desc(const pth_scene *s) { return s ? &s->scene->desc() : nullptr; }
extern "C" const pt_world_desc *pth_scene_world_desc(const pth_scene *s) { return s ? &s->scene->world_desc() : nullptr; }
extern "C" int pth_scene_is_world(const pth_scene *s) { return s && s->scene->is_world() ? 1 : 0; }
extern "C" const pt_camera *pth_scene_camera(const pth_scene *s) { return s ? &s->camera.pod : nullptr; }
extern "C" pt_scene *pth_scene_handle(const pth_scene *s) { return s ? s->scene->handle() : nullptr; }
extern "C" uint64_t pth_scene_build_draws(const pth_scene *s) { return s ? s->build_draws : 0; }
extern "C" uint32_t pth_scene_bvh_depth(const pth_scene *s) { return s ? s->bvh_depth : 0; }

extern "C" int pth_render_offline(const char *preset, uint32_t width, uint32_t height, uint32_t samples,
                                  uint32_t max_depth, int use_bvh, int random_seed, int device, const char *output,
                                  uint32_t frames) {
    pt::Params params;
    params.width = width;
    params.height = height;
    params.samples = samples;
    params.max_depth = max_depth;
    params.use_bvh = use_bvh != 0;
    params.random_seed = random_seed != 0;
    return pt::render_offline(preset ? preset : "", params, device, output ? output : "output.png", frames);
}

extern "C" void pth_linear_to_srgb(const float rgb[3], uint8_t out[3]) { pt::linear_to_srgb(rgb, out); }
extern "C" int pth_save_png(const char *path, const float *buffer, uint32_t width, uint32_t height) {
    return pt::save_png(path, buffer, width, height) ? 0 : 1;
}
