// pt_host.h -- internal header of the host side of libptgpu.so (the C ABI in include/ptgpu.h): the scene handle, error
// plumbing and the functions the translation units share. Nothing here is part of the ABI.
//
//   pt_api.hip      errors, version, device count, tuning, timing / debug getters, self-test probes
//   pt_prep.hip     host-side scene analysis: MFMA prefilter tiles + culling tables, binary tree, 4-wide tree restatement
//   pt_scene.hip    pt_scene_create / pt_scene_create_world / pt_scene_destroy (validation, flattening, upload)
//   pt_select.h     which kernel, which geometry (pure); pt_kernels_*.hip hold the instantiations
//   pt_launch.hip   one Scene::update: argument blocks, work order, enqueue
//   pt_render.hip   the entry points over launch(): device buffer, shard, host buffer (pipelined copies), prepare
//   pt_comm.hip     multi-GPU frames: RCCL communicator (resolved at run time), shard pack / unpack kernels
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#include "pt_args.h"
#include "pt_build.h"
#include "pt_select.h"
#include "ptgpu.h"

namespace pthostside {

using namespace ptdev;

// error plumbing: every ABI function returns fail(code, ...) on error; pt_last_error() hands out the message
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
const char *last_error_message();

#define HIP_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) return ::pthostside::fail(PT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));  \
    } while (0)

// Development knobs (environment), read ONCE per process and only in builds made with -DPT_DEVKNOBS (the shipped library
// has none: every knob below then keeps its default).
struct DevKnobs {
    int refill = -1, ready = -1, drain = -1, phase1_refill = -1, cull_axis = -1, cull_strips = -1;
    bool world_occ4 = false, world_occ3 = false, debug = false, clamp_grid = false, timing = false;
    int host_threads = -1;
    int coop_live = -1, coop_streak = -1, coop_period = -1, coop_est = -1;   // cooperative hand-over policy (pt_coop.h)
    int park_max = -1, park_after = -1;   // cell-grid kernels: lanes that may park their walk / rounds before they may (pt_grid.h)
    int pool = -1, pool_tail = -1;   // wide list kernels: pixel pool entries per wave / its end-of-list threshold (pt_kernel.h POOL)
    uint32_t variant = 0, blocks_per_cu = 0;
};
const DevKnobs &dev_knobs();

// Motion of a MovingSphere entry (moving_sphere.rs:8-14) next to the pt_sphere holding centre_start / radius.
struct MotionIn {
    float delta[3];
    float time_start, inv_time_delta;
    uint32_t moving;
};

template <typename T>
int upload(T **dst, const void *src, size_t count) {
    HIP_TRY(hipMalloc((void **)dst, count * sizeof(T) > 0 ? count * sizeof(T) : sizeof(T)));
    if (count) HIP_TRY(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    return PT_OK;
}

}  // namespace pthostside

struct pthostside_grid_geom {   // the scalars of a GridPlan the launch copies into KArgs
    uint32_t n[3], n_records, n_large;
    float gmin[3], h, ha[3], centre[3], half_diag, d_build;
};

// ---- the scene handle ------------------------------------------------------------------------------------------------
struct pt_scene {
    int device = 0;
    int num_cus = 0;
    ptsel::SceneTraits tr;     // everything kernel selection may look at (pt_select.h)
    uint32_t n_materials = 0, n_textures = 0;
    int32_t bvh_root = -1;     // root of the CALLER's tree (-1: created without BVH nodes)
    uint32_t has_sky = 0;
    float sky[3] = {0, 0, 0};
    // device memory: sphere scenes
    float4 *d_spheres = nullptr, *d_spheres_r2 = nullptr, *d_shade = nullptr;
    uint32_t *d_sphere_mat = nullptr;
    ptdev::DMat *d_mats = nullptr;
    ptdev::DTex *d_texs = nullptr;
    float4 *d_perlin_vec = nullptr;
    uint32_t *d_perlin_perm = nullptr;
    float4 *d_gate = nullptr, *d_gate_chain = nullptr;
    uint32_t *d_bvh_large = nullptr;
    uint32_t n_bvh_large = 0;
    ptdev::DWideNode *d_wnodes = nullptr;            // binary internal tree, built on the host when first needed
    ptdev::DNode4 *d_nodes4 = nullptr;               // 4-wide internal tree (default of the tree kernels), built on the device
    ptdev::DNode4Q *d_nodes4q = nullptr;             // ... as the packed 64-byte nodes the kernels read (pt_tree4.h), and the leaves' slot records
    float4 *d_slotrec = nullptr;
    uint4 *d_grid_cells = nullptr;                   // uniform cell grid (GridPlan below; tr.grid_ok)
    uint32_t *d_grid_large = nullptr;
    float4 *d_grid_rec = nullptr;
    pthostside_grid_geom grid_geom{};
    float tree_build_ms = 0.f;                       // device time of that build (HIP events)
    bool tree_on_device = false;
    std::vector<pt_sphere> h_spheres;                // kept for the lazily built binary tree
    std::vector<pthostside::MotionIn> h_motion;
    int32_t bin_root = -1;
    bool has_tree_items = false;                     // some sphere is inside the internal tree (else every one is in d_bvh_large)
    double h_t_lo = 0.0, h_t_hi = 0.0;
    bool binary_built = false;
    uint32_t *d_leaf_rank = nullptr, *d_rank_sphere = nullptr;
    float4 *d_leafrec = nullptr, *d_shade_rank = nullptr;   // BVH worlds: sphere + gate + rank per sphere (KArgs::slotrec source), shading records in rank order
    // MFMA prefilter data (tr.n_tiles == 0: prefilter not applicable to this scene)
    uint4 *d_afrag = nullptr;
    uint16_t *d_tile_sphere = nullptr;
    uint32_t *d_large = nullptr;
    uint32_t n_large = 0, large0 = 0xffffffffu;
    float c0[3] = {0, 0, 0};
    float rs2 = 0.f, m0 = 0.f, gamma = 0.f;
    uint32_t *d_cull_tab = nullptr;                  // tile-culling tables (cull_axis == 3: off)
    uint32_t cull_axis = 3, cull_always = 0, cull_axis2 = 0;
    float cull_u0_2 = 0.f, cull_inv_cell_2 = 0.f;
    float cull_u0 = 0.f, cull_inv_cell = 0.f, cull_rmin = 0.f, cull_rmax = 0.f, rs_small = 0.f, clip_min[3] = {0, 0, 0}, clip_max[3] = {0, 0, 0};
    float4 *d_motion = nullptr;                      // MovingSphere records of a Sphere + MovingSphere world
    // device memory: general worlds (also the fallback data of a Sphere + MovingSphere world)
    pt_hitable *d_hitables = nullptr;
    pt_affine *d_transforms = nullptr;
    pt_bvh_node *d_ref_nodes = nullptr;              // the caller's tree as given (BVHNode::ray_hit is followed literally)
    uint4 *d_image_table = nullptr;                  // Texture::Image sources: (byte offset, width, height, 0)
    uint8_t *d_image_bytes = nullptr;
    // per-frame state. ONE frame in flight per handle (include/ptgpu.h "Threading"): d_px_state, d_tile_buf, d_work_counter
    // and the work-order hint below belong to the frame being rendered.
    unsigned long long *d_debug = nullptr;           // verify-mode / traversal counters
    uint32_t *d_tile_buf = nullptr;                  // [8 scratch words | n tile costs | n tile order | n tile costs measured by the last frame]
    size_t d_tile_cap = 0;
    uint4 *d_px_state = nullptr;                     // two-launch frames: parked (xoshiro state, colour sum) per pixel, 48 B each
    size_t d_px_state_pixels = 0;
    uint32_t *d_work_counter = nullptr;              // 64 bytes: [0] next work item, the rest the hand-over protocol of the cooperative mode (pt_coop.h)
    uint64_t *d_tail_box = nullptr;                  // cooperative mode: one mailbox (128 B) per wave of the largest grid so far, stamped with the launch's generation
    uint32_t tail_cap = 0, tail_gen = 0;
    unsigned long long *d_ray_count = nullptr;       // internal counter of the host-buffer entry point
    float *d_frame = nullptr;                        // internal frame (pt_scene_prepare's throw-away frame, unpinned fallback of pt_render)
    float *h_stage = nullptr;                        // pinned + mapped copy of the caller's pageable buffer: the kernels render into it over PCIe
    float *h_stage_dev = nullptr;                    // ... as the device addresses it
    size_t frame_floats = 0;
    float *d_gstack = nullptr;                       // attenuation-stack levels that do not fit the LDS
    uint4 *d_gnodes = nullptr;                       // interpreted scene graphs (pt_graph.h): pt_node rows, HitableList children, frames of the walk
    uint32_t *d_gchildren = nullptr;
    uint32_t groot = 0;
    float *d_gframes = nullptr;
    size_t d_gframes_floats = 0;
    size_t d_gstack_floats = 0;
    // Work order of the NEXT frame of the same view: the rays each tile really took in the last frame (same scene, camera,
    // size, samples, depth, shard). Only the order of the work depends on it, never a pixel.
    struct ViewKey {
        pt_params params;
        pt_camera cam;
        uint32_t shard_index, shard_count, variant, n_tiles;
    } hint_key{};
    bool hint_valid = false;
    // what the last frame's launches left in d_tile_buf, for pt_scene_debug_tile_rays: tiles, tiles per row, whether a measuring launch counted the
    // first sample (of every tile / of the even tiles of a checkerboard) into the cost words, whether the frame kernel counted into the measured words
    struct { uint32_t n_tiles = 0, tiles_x = 0; bool first_sample_in_cost = false, checker = false, frame_counted = false; } tile_rays_info;
    uint32_t hint_scale = 1;                         // bucket width of the measured costs
    uint64_t seed_base = 0x243f6a8885a308d3ull;
    uint32_t blocks_per_cu = 0, variant = 0;         // pt_scene_set_tuning
    // last launch
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;   // around the frame kernel alone (what rocprofv3 reports for it)
    hipEvent_t ev_pass = nullptr;                       // before the measuring launch: ev_pass..ev_stop = the whole Scene::update
    bool ev_valid = false;
    uint32_t last_grid = 0, last_block = 0, last_lds = 0;
    ptsel::KernelChoice last_choice;
    float host_ms[4] = {0, 0, 0, 0};                 // last pt_render on a pageable buffer: scan / copy-in, wait for the GPU, copy-out, whole call
    unsigned long long *d_wave_end = nullptr;        // -DPT_DEVKNOBS builds with PTGPU_TIMING=1: per-wave finish times of the last launch
};

namespace pthostside {

// ---- pt_prep.hip: host-side analysis of a sphere scene ---------------------------------------------------------------
// Conservative bound of sphere i over every ray time in [t_lo, t_hi]: centre of the swept segment and the half-length to
// add to |radius| (zero for plain spheres).
struct Sweep {
    double c[3];
    double half;
};
Sweep sweep_of(const pt_sphere &p, const MotionIn *m, double t_lo, double t_hi);

// MFMA prefilter preparation (DESIGN.md "MFMA prefilter"): sphere-feature fragments, the always-tested "large" set, and the
// tile-culling tables.
struct MfmaPrep {
    std::vector<uint16_t> afrag;  // f16 bit patterns, [tile][chunk][lane][8]
    std::vector<uint16_t> tile_sphere;
    std::vector<uint32_t> large;
    float c0[3] = {0, 0, 0};
    double rs = 0.0;
    double sweep_ratio = 0.0;  // max over prefiltered spheres of (swept half-length / |radius|)
    uint32_t n_tiles = 0;
    // tile culling: sort axis (3 = off), tiles that are always run, lookup tables (kCullCells cells), padded box of the sorted spheres
    uint32_t cull_axis = 3, cull_always = 0, cull_axis2 = 0, cull_strips = 1;
    std::vector<uint32_t> cull_tab;
    float cull_u0_2 = 0.f, cull_inv_cell_2 = 0.f;
    float cull_u0 = 0.f, cull_inv_cell = 0.f, cull_rmin = 0.f, cull_rmax = 0.f, clip_min[3] = {0, 0, 0}, clip_max[3] = {0, 0, 0};
};
bool prepare_mfma(const pt_scene_desc *desc, const MotionIn *motion, double t_lo, double t_hi, MfmaPrep &out);

// Internal traversal trees (results never depend on them).
struct AccelItem {
    uint32_t sphere;
    float c[3], mn[3], mx[3], r, signed_r;
    float c_start[3];  // the sphere as stored (centre_start for a MovingSphere): what the leaf slot carries
};
struct AccelBuild {
    std::vector<DWideNode> nodes;
    std::vector<uint32_t> large;
    int32_t root = -1;
    uint32_t depth = 0;
};
std::vector<AccelItem> accel_items(const pt_scene_desc *desc, const MotionIn *motion, double t_lo, double t_hi, std::vector<uint32_t> &large);
AccelBuild build_accel(const pt_scene_desc *desc, const MotionIn *motion, double t_lo, double t_hi);
struct Tree4Host {
    std::vector<DNode4> nodes;
    uint32_t depth = 0;
};
Tree4Host tree4_build_host(std::vector<TreeItem> items);

// Uniform cell grid over the small spheres of a big scene (csrc/pt_grid.h): what the tree kernels walk instead of the 4-wide tree when the
// spheres are many, of similar size and spread evenly enough. Results never depend on it. One RECORD is five 16-byte words: up to four
// spheres (cx, cy, cz, radius as KArgs::spheres holds them; an empty slot holds a sphere nothing hits; spheres 0 | 1 and 2 | 3 are stored
// side by side, component by component: (x0 x1 y0 y1) (z0 z1 r0 r1) (x2 x3 y2 y3) (z2 z3 r2 r3)) and their list indices; the first
// n[0] n[1] n[2] records are the cells (x fastest), a cell with more than four spheres continues in a record behind them (three spheres
// + link 0x80000000 | record). A sphere is registered in every cell its box -- padded by how far the reference's f32 discriminant can
// inflate it for a ray whose origin lies within `d_build` of it (pt_tree4.h: 0.65e-6 (|o - c|^2 + r^2) / r; 1e-6 here) plus h / 1000 for the
// walk's own rounding -- overlaps. Rays from farther away walk the 4-wide tree instead (pt_grid.h), so `d_build` is a cost knob, not a limit.
struct GridPlan {
    bool ok = false;
    uint32_t n[3] = {1, 1, 1}, n_records = 0;
    float gmin[3] = {0, 0, 0}, h = 0.f, ha[3] = {0, 0, 0};   // h: the cubic cell; ha: per axis (an axis with ONE cell spans the whole extent, up to 1.5 h)
    float centre[3] = {0, 0, 0}, half_diag = 0.f, d_build = 0.f;
    std::vector<uint4> cells;          // 5 per record
    std::vector<float4> rec;           // per SPHERE, the slot record of an exact test (KArgs::slotrec layout: sphere | gate min | gate max | rank bits, index bits): filled by plan_sphere_scene
    std::vector<uint32_t> large;       // spheres outside the grid: tested for every ray
    double items_per_cell = 0.0, records_per_cell = 0.0, occupied = 0.0;   // (statistics of the plan; occupied: share of the cells that hold a sphere)
};
bool plan_cell_grid(const pt_scene_desc *desc, const MotionIn *motion, double t_lo, double t_hi, const std::vector<float4> &sph, GridPlan &out);

// Everything pt_scene_create derives from a sphere scene's description BEFORE anything touches the device: the flattened
// device layouts and the traits kernel selection looks at. pt_debug_select runs exactly this, so the selection table can be
// enumerated on a machine without a GPU.
struct SpherePlan {
    ptsel::SceneTraits tr;
    bool has_motion = false;
    double t_lo = 0.0, t_hi = 0.0;
    std::vector<float4> sph, sph_r2, shade, gate, gate_chain, leafrec, shade_rank, pvec, mot;
    std::vector<uint32_t> leaf_rank, bvh_large, rank_sphere, pperm;
    std::vector<DMat> mats;
    std::vector<DTex> texs;
    std::vector<TreeItem> titems;
    bool has_prep = false;
    MfmaPrep prep;
    GridPlan grid;
};
// validation + analysis of a sphere scene (no HIP call); `motion` optional (n_spheres entries)
int plan_sphere_scene(const pt_scene_desc *desc, const MotionIn *motion, SpherePlan &plan);
int validate_tables(uint32_t n_materials, const pt_material *materials, uint32_t n_textures, const pt_texture *textures, const pt_perlin *perlin,
                    bool allow_isotropic, bool *has_noise_out, uint32_t n_images = 0, const pt_image *images = nullptr);
uint32_t bvh_depth_checked(const pt_bvh_node *nodes, uint32_t n_nodes, uint32_t n_leaves, int32_t root);
// world description -> (is it a Sphere / MovingSphere world?) + the sphere-scene view of it
struct WorldAsSpheres {
    bool sphere_like = false, all_spheres = false, has_media = false, has_image = false, has_noise = false, has_chains = false;
    bool is_graph = false;   // an interpreted scene graph (pt_graph.h): `hitables` are its leaves
    uint32_t ref_depth = 0;
    std::vector<pt_sphere> sph;
    std::vector<uint32_t> mat;
    std::vector<MotionIn> motion;
    std::vector<pt_texture> folded;
    pt_scene_desc desc{};
};
int analyze_world(const pt_world_desc *desc, WorldAsSpheres &out);
// A world given as a scene graph (pt_world_desc::nodes), flattened into the list form every kernel reads (include/ptgpu.h
// states which nestings that is exact for). `flat` is a copy of the description whose hitables / transforms / BVH leaves
// point into this object; descriptions without a graph are passed through untouched.
struct FlatWorld {
    std::vector<pt_hitable> hit;
    std::vector<pt_affine> xf;
    std::vector<pt_bvh_node> bvh;   // the caller's BVH with its leaves re-indexed (root-list child -> first list entry of that child)
    pt_world_desc flat{};
    bool interpreted = false;   // the graph does not flatten (pt_graph.h): `flat` is the caller's description, nodes and all
};
int flatten_world_graph(const pt_world_desc *desc, FlatWorld &out, const pt_world_desc **use);
void world_traits(const pt_world_desc *desc, const WorldAsSpheres &w, ptsel::SceneTraits &tr);

// ---- pt_kernels_*.hip: the instantiations behind a KernelChoice --------------------------------------------------------
typedef void (*SphereKernel)(const KArgs);
typedef void (*WorldKernel)(const WArgs);
// frame kernel and its measuring twin (PILOT instantiation; nullptr when the family has none)
void sphere_kernels_for(const ptsel::KernelChoice &c, SphereKernel *frame, SphereKernel *measure);
WorldKernel world_kernel_for(const ptsel::KernelChoice &c);
SphereKernel tree4_kernel_for_registers(bool moving);   // the symbol whose register count decides how many tree workgroups fit a CU
const char *kernel_name(const ptsel::KernelChoice &c, char *buf, size_t cap);
void launch_tile_order(uint32_t n_work_tiles, uint32_t *tile_cost, uint32_t cost_scale, uint32_t *tile_order, uint32_t checker_tiles_x, uint32_t checker_tiles_y,
                       uint32_t *work_counter, hipStream_t stream);
void launch_frame_reset(uint32_t *work_counter, unsigned long long *ray_count, uint32_t *cost, uint32_t n_cost, uint32_t *list, uint32_t tiles_x, uint32_t tiles_y,
                        hipStream_t stream);

// ---- pt_launch.hip -------------------------------------------------------------------------------------------------
// One Scene::update (scene.rs:73-121) enqueued on `stream`. `before_frame`, when given, runs on the host after the measuring
// launch has been enqueued and before the frame kernel is: the host-buffer entry point scans / copies the caller's buffer
// there, under the measuring launch, and reports whether the buffer is all +0.0f (the kernel then does not read it).
typedef std::function<int(bool *prev_zero)> BeforeFrame;
int launch(pt_scene *s, const pt_params *params, const pt_camera *cam, uint32_t frame_num, uint32_t shard_index, uint32_t shard_count, float *d_rgb,
           uint64_t *d_ray_count, hipStream_t stream, const BeforeFrame *before_frame = nullptr);
int ensure_frame_buffers(pt_scene *s, size_t floats);

}  // namespace pthostside
