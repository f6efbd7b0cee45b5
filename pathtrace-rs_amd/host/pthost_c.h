/* pthost_c.h -- C view of the C++ host (libpthost.so) for ctypes callers
 * (bench.py, tests). The host mirrors the reference's scene-construction API
 * (presets.rs / camera.rs / params.rs / storage.rs / offline.rs); see host.hpp. */
#ifndef PTHOST_C_H
#define PTHOST_C_H
#include <stdint.h>

#include "ptgpu.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pth_scene pth_scene;

/* offline.rs:16-24: Params::new_rng -> Storage::new -> presets::from_name -> Params::new_scene.
 * device >= 0 uploads through pt_scene_create; device < 0 builds the description only.
 * Returns 0, 2 for an unrecognised preset, 1 for any other failure (pth_last_error()). */
int pth_scene_build(const char *preset, uint32_t width, uint32_t height, uint32_t samples, int use_bvh,
                    int device, int quiet, pth_scene **out);
void pth_scene_free(pth_scene *s);
const pt_scene_desc *pth_scene_desc(const pth_scene *s);            /* sphere-only worlds (!pth_scene_is_world) */
const pt_world_desc *pth_scene_world_desc(const pth_scene *s);      /* every world */
int pth_scene_is_world(const pth_scene *s);                         /* 1: has non-sphere hitables (general kernel) */
const pt_camera *pth_scene_camera(const pth_scene *s);
pt_scene *pth_scene_handle(const pth_scene *s);
uint64_t pth_scene_build_draws(const pth_scene *s);
uint32_t pth_scene_bvh_depth(const pth_scene *s);

/* offline.rs:16-60 */
int pth_render_offline(const char *preset, uint32_t width, uint32_t height, uint32_t samples, uint32_t max_depth,
                       int use_bvh, int random_seed, int device, const char *output, uint32_t frames);
/* math.rs:36-48 / offline.rs:43-59 */
void pth_linear_to_srgb(const float rgb[3], uint8_t out[3]);
int pth_save_png(const char *path, const float *buffer, uint32_t width, uint32_t height);
const char *pth_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
