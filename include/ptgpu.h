/*
 * ptgpu.h -- C ABI of the MI355X-native path-tracing hot path (libptgpu.so).
 *
 * This is the drop-in boundary for ONE path of bitshifter/pathtrace-rs 0.1.2:
 *     Scene::update(&self, &Params, &Camera, frame_num, &mut [(f32,f32,f32)]) -> usize
 *                                                     (reference src/scene.rs:73-121)
 * and everything it calls (ray_trace scene.rs:49-71, Hitable::ray_hit
 * collision/hitable.rs:39-65, Material::scatter material.rs:138-159,
 * Texture::value texture.rs:74-91, Camera::get_ray camera.rs:56-68).
 *
 * The reference has no FFI / plugin layer; a Rust host would bind these entry
 * points with `extern "C"` inside Scene::new / Scene::update (binding shown in
 * INTEGRATION.md). Plain pointers and sizes only; no C++ or torch types.
 * All functions return PT_OK (0) or a PT_ERR_* code and never throw or abort;
 * pt_last_error() returns a thread-local message for the last failure.
 *
 * Threading: thread-compatible. ONE frame in flight per pt_scene handle, on ONE
 * stream at a time (same contract as Scene::update(&self) being called from one
 * thread, offline.rs:29 / glium_window.rs:102): a handle owns per-frame device
 * state (parked pixel streams of a two-launch frame, tile costs, the work
 * counter), so frames of one handle enqueued on different streams must be
 * ordered by the caller. Distinct handles may be used from distinct threads
 * and streams (bench.py's pipelined frames use two).
 */
#ifndef PTGPU_H
#define PTGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_OK 0
#define PT_ERR_INVALID_ARG 1  /* NULL pointer, zero size, bad index in the scene description */
#define PT_ERR_HIP 2          /* a HIP runtime call failed (message carries hipGetErrorString) */
#define PT_ERR_NO_DEVICE 3    /* no gfx950 device / HIP runtime unavailable */
#define PT_ERR_UNSUPPORTED 4  /* e.g. use_bvh requested on a scene created without BVH nodes */

/* params.rs:11-18 `Params` (bools widened to u32) */
typedef struct pt_params {
    uint32_t width;
    uint32_t height;
    uint32_t samples;
    uint32_t max_depth;
    uint32_t random_seed; /* != 0: per-pixel seeds derive from pt_scene_set_seed_base (no parity claim) */
    uint32_t use_bvh;     /* != 0: BVHNode::ray_hit traversal (bvh.rs:37-62); 0: HitableList scan (hitable_list.rs:40-56) */
} pt_params;

/* camera.rs:8-19 `Camera`: 24 floats in declaration order */
typedef struct pt_camera {
    float origin[3];
    float lower_left_corner[3];
    float horizontal[3];
    float vertical[3];
    float u[3];
    float v[3];
    float w[3];
    float time0;
    float time1;
    float lens_radius;
} pt_camera;

/* collision/sphere.rs:8-11 `Sphere` (radius is signed: presets.rs:265 uses -0.45) */
typedef struct pt_sphere {
    float cx, cy, cz, radius;
} pt_sphere;

/* material.rs:13-19 `Material` (Isotropic: phase function of a ConstantMedium, general worlds only) */
enum { PT_MAT_LAMBERTIAN = 0, PT_MAT_METAL = 1, PT_MAT_DIELECTRIC = 2, PT_MAT_DIFFUSE_LIGHT = 3, PT_MAT_ISOTROPIC = 4 };
typedef struct pt_material {
    uint32_t kind;
    float albedo[3]; /* Metal albedo */
    float param;     /* Metal fuzz | Dielectric ref_idx */
    int32_t texture; /* Lambertian / Isotropic albedo, DiffuseLight emit texture index, else -1 */
} pt_material;

/* texture.rs:40-55 `Texture`. Image (texture.rs:5-37 RgbImage) is sampled at the hit's (u, v): Rect hits carry real
 * coordinates (rect.rs:97-98), every sphere hit of the live path has u = v = 0 (sphere.rs:47-48), i.e. reads the
 * first texel of the LAST row. Images belong to general worlds (pt_world_desc.images). */
enum { PT_TEX_CONSTANT = 0, PT_TEX_CHECKER = 1, PT_TEX_NOISE = 2, PT_TEX_IMAGE = 3 };
typedef struct pt_texture {
    uint32_t kind;
    float color[3]; /* Constant */
    int32_t odd;    /* Checker: texture indices (may nest) | Image: index into pt_world_desc.images */
    int32_t even;
    float scale;    /* Noise */
} pt_texture;

/* perlin.rs:7-12 `Perlin` */
typedef struct pt_perlin {
    float randvec[256][3];
    uint32_t perm_x[256];
    uint32_t perm_y[256];
    uint32_t perm_z[256];
} pt_perlin;

/* collision/bvh.rs:24-28 `BVHNode`, flattened: child >= 0 is a node index,
 * child < 0 is ~sphere_index (a Hitable::Sphere leaf). */
typedef struct pt_bvh_node {
    float min[3];
    float max[3];
    int32_t lhs;
    int32_t rhs;
} pt_bvh_node;

/* The Scene (scene.rs:18-22) as flat PODs: `world` (List order = sphere
 * order; optional BVH over the same spheres) + `sky`. */
typedef struct pt_scene_desc {
    uint32_t n_spheres;
    const pt_sphere *spheres;        /* n_spheres, in HitableList order */
    const uint32_t *sphere_material; /* n_spheres material indices */
    uint32_t n_materials;
    const pt_material *materials;
    uint32_t n_textures;
    const pt_texture *textures;
    const pt_perlin *perlin;         /* NULL unless a Noise texture exists */
    uint32_t n_bvh_nodes;            /* 0: list only */
    const pt_bvh_node *bvh_nodes;
    int32_t bvh_root;                /* node index of the root, -1 when no BVH */
    uint32_t has_sky;                /* scene.rs:20 Option<Vec3> */
    float sky[3];
} pt_scene_desc;

/* ---- general worlds: the other Hitable arms of collision/hitable.rs:12-21 ------
 * One pt_hitable per HitableList entry: the innermost shape plus, optionally, the Instance
 * (instance.rs:9-13) and ConstantMedium (constant_medium.rs:11-15) wrapped around it in the order the
 * reference's presets nest them: ConstantMedium(Instance(shape)). Deeper nestings go through the scene graph
 * of pt_world_desc (pt_node), which is flattened into this form. */
enum {
    PT_HIT_SPHERE = 0,        /* sphere.rs:8-11          p = cx cy cz radius */
    PT_HIT_MOVING_SPHERE = 1, /* moving_sphere.rs:8-14   p = centre_start(3) centre_delta(3) radius time_start inv_time_delta */
    PT_HIT_RECT_XY = 2,       /* rect.rs:6-31            p = a0 a1 b0 b1 k: the two in-plane ranges in the variant's */
    PT_HIT_RECT_XZ = 3,       /*                          own order (XY: x,y  XZ: x,z  YZ: y,z) and the plane offset */
    PT_HIT_RECT_YZ = 4,
    PT_HIT_CUBOID = 5,        /* cuboid.rs:4-23          p = p0(3) p1(3); faces derived as Cuboid::new does */
    PT_HIT_MEDIUM_GROUP = 6   /* constant_medium.rs:11-15 whose BOUNDARY is a HitableList: this entry is the medium (medium_material, density, and in
                               * `transform` the Instances AROUND it: outer levels only), and the next N entries -- N = the bits of p[0] read as a u32,
                               * >= 1 -- are the list's children in order: shapes (kinds 0..5) under their own Instance levels (`transform`: inner
                               * levels only), medium_material = -1. The children are NOT list entries of their own: the scan asks them twice, as the
                               * medium's boundary (constant_medium.rs:39-43 over hitable_list.rs:40-56), and skips them; BVH leaves may index the
                               * group's first entry only. `material` of the group entry is ignored. Written by the scene-graph flattener for a
                               * ConstantMedium node around a List node; accepted in a hand-written list as well. */
};
typedef struct pt_hitable {
    uint32_t kind;
    uint32_t material;       /* the &Material paired with the shape */
    uint32_t flip_normals;   /* Rect */
    int32_t transform;       /* Instance: index into transforms, -1 = none. (Several nested Instances, as the scene-graph
                              * flattener below writes them: first index | inner levels << 20 | outer levels << 24 -- `outer`
                              * Instances around the ConstantMedium, then `inner` ones between it and the shape, outermost
                              * first, consecutive in `transforms`; at most 15 each, n_transforms < 2^20.) */
    int32_t medium_material; /* ConstantMedium: index of its Isotropic phase-function material, -1 = none */
    float density;           /* ConstantMedium */
    float p[10];
} pt_hitable;

/* glam Affine3A (x_axis, y_axis, z_axis, translation) and its inverse, as Instance::new stores them
 * (instance.rs:16-22); the inverse is taken from the caller so the device never inverts. */
typedef struct pt_affine {
    float m[12];
    float inv[12];
} pt_affine;

/* texture.rs:5-10 `RgbImage`: tightly packed RGB8 rows as image::open(..).to_rgb8().into_raw() yields them */
typedef struct pt_image {
    uint32_t width, height;
    const uint8_t *rgb; /* width * height * 3 bytes */
} pt_image;

/* Optional scene graph: collision/hitable.rs:12-21 lets Hitables nest freely (List in List, Instance of Instance, an
 * Instance around a ConstantMedium, ...). With n_nodes > 0 the world is the graph rooted at nodes[root_node] and
 * `hitables` are its leaf shapes (their own transform / medium_material must be -1: wrappers are nodes here).
 * pt_scene_create_world FLATTENS the graph into the list form above, which is exact for
 *   - HitableList inside HitableList: the narrowing scan of hitable_list.rs:40-56 is the scan of the concatenation;
 *   - Instance around a HitableList: Instance::ray_hit (instance.rs:32-47) builds the same local ray for every child
 *     and t is shared between the two spaces, so Instance(List(a, b)) = List(Instance(a), Instance(b));
 *   - any depth of Instance around a shape, and around or inside a ConstantMedium (transform chains, see pt_hitable).
 * A ConstantMedium whose boundary is a HitableList of shapes (each under any number of Instances) becomes a PT_HIT_MEDIUM_GROUP entry
 * followed by its children. Not expressible in the list form: a ConstantMedium whose boundary holds another ConstantMedium or a List
 * inside a List (constant_medium.rs:32-43 asks its boundary twice, and a medium in there draws from the pixel's RNG both times), and a
 * BVHNode below the root (PT_NODE_BVH: bvh.rs:37-62 as a Hitable anywhere, hitable.rs:12-21). A graph with either is not
 * flattened but INTERPRETED on the device (csrc/pt_graph.h: Hitable::ray_hit as the reference recurses, one stack of frames
 * per lane; correct for every nesting, several times slower than the list form; at most 24 nested ray_hit calls;
 * pt_params.use_bvh is refused for it -- its BVHNodes are part of the graph). Rejected with PT_ERR_UNSUPPORTED and a message
 * that names the node: a flattened graph deeper than 15 Instance levels on one side of a medium, an interpreted one deeper
 * than 24 levels, a cycle. With bvh_nodes and a graph that flattens, leaves index the ROOT list's children, each of which
 * must flatten to exactly one list entry. */
enum { PT_NODE_HITABLE = 0, PT_NODE_LIST = 1, PT_NODE_INSTANCE = 2, PT_NODE_MEDIUM = 3, PT_NODE_BVH = 4 };
typedef struct pt_node {
    uint32_t kind;
    uint32_t a;    /* HITABLE: index into hitables | LIST: first index into node_children | INSTANCE: index into transforms |
                    * MEDIUM: index of its Isotropic phase-function material | BVH: index into bvh_nodes -- that row's box, and its
                    * lhs / rhs read as NODE indices (>= 0) */
    uint32_t b;    /* LIST: number of children | INSTANCE, MEDIUM: the child node | BVH: 0 */
    float density; /* MEDIUM */
} pt_node;

/* bvh_nodes children: >= 0 node index, < 0 ~hitable_index. */
typedef struct pt_world_desc {
    uint32_t n_hitables;
    const pt_hitable *hitables; /* HitableList order */
    uint32_t n_transforms;
    const pt_affine *transforms;
    uint32_t n_materials;
    const pt_material *materials;
    uint32_t n_textures;
    const pt_texture *textures;
    const pt_perlin *perlin;
    uint32_t n_bvh_nodes;
    const pt_bvh_node *bvh_nodes;
    int32_t bvh_root;
    uint32_t has_sky;
    float sky[3];
    uint32_t n_images;          /* Texture::Image sources (0 / NULL when unused) */
    const pt_image *images;
    uint32_t n_nodes;           /* scene graph (0: the world is the flat list `hitables`) */
    const pt_node *nodes;
    uint32_t n_node_children;
    const uint32_t *node_children;
    uint32_t root_node;
} pt_world_desc;

typedef struct pt_scene pt_scene;

/* Number of visible HIP devices. */
int pt_device_count(int *count_out);

/* Scene::new (scene.rs:25-31): validates and uploads the scene to `device`
 * (HBM-resident SoA). The description is copied; the caller
 * may free it afterwards. */
int pt_scene_create(const pt_scene_desc *desc, int device, pt_scene **scene_out);
void pt_scene_destroy(pt_scene *scene);

/* Scene::new for a world with any of the Hitable arms above. Every render entry point below accepts the
 * handle. Such worlds are traced by the general kernel: HitableList::ray_hit's sequential narrowing scan
 * (hitable_list.rs:40-56) or BVHNode::ray_hit's both-children recursion (bvh.rs:37-62) exactly as written,
 * because ConstantMedium::ray_hit draws from the pixel's RNG inside the intersection (constant_medium.rs:60)
 * and clamps against the running t_max, so visiting order is part of the result. */
int pt_scene_create_world(const pt_world_desc *desc, int device, pt_scene **scene_out);

/* Optional, part of Scene::new: allocates the per-frame device buffers (frame + pinned staging copy, path
 * stacks, tile-order scratch) for `params` and runs the kernels once on a throw-away 1-spp frame, so that the
 * first pt_render measures rendering and not hipMalloc / code-object loading (the reference's timer spans
 * Scene::update only, offline.rs:27-34). Rendering works without it. */
int pt_scene_prepare(pt_scene *scene, const pt_params *params);

/* Scene::update (scene.rs:73-121) with a HOST pixel buffer, exactly the
 * reference's contract: rgb_inout is width*height*3 floats (row 0 = bottom
 * row, offline.rs:44), READ (frame blend scene.rs:114-116) and written;
 * *ray_count_out receives the number of ray_trace invocations. Synchronous.
 * Limits (PT_ERR_UNSUPPORTED beyond them): width, height < 65536 (a lane keeps its pixel in one register); sphere scenes also
 * max_depth < 4096 and samples < 2^20 (their kernels pack the (depth, sample) counters into one register). */
int pt_render(pt_scene *scene, const pt_params *params, const pt_camera *camera,
              uint32_t frame_num, float *rgb_inout, uint64_t *ray_count_out);

/* Optional, for a host that keeps ONE pixel buffer alive across calls (offline.rs:25 allocates it once; the preview
 * window reuses it every frame, glium_window.rs:94-133): pins and maps `bytes` of host memory so that pt_render on any
 * sub-range of it renders IN PLACE over PCIe -- the previous frame is read and the new one written pixel by pixel while
 * the kernel runs, with no staging copy before or after. The caller must unregister before freeing the memory. */
int pt_buffer_register(void *host_ptr, size_t bytes);
int pt_buffer_unregister(void *host_ptr);

/* Same with a DEVICE-resident buffer on `hip_stream` (hipStream_t, NULL =
 * default stream); asynchronous. d_ray_count (device, 8 bytes) is overwritten
 * with this call's ray count. The accumulation buffer stays in HBM across
 * frames (progressive mode, glium_window.rs:94-133). */
int pt_render_device(pt_scene *scene, const pt_params *params, const pt_camera *camera,
                     uint32_t frame_num, float *d_rgb_inout, uint64_t *d_ray_count,
                     void *hip_stream);

/* Multi-GPU shard of Scene::update: renders only the rows y of the frame with
 * y % shard_count == shard_index into a COMPACT device buffer of
 * pt_shard_rows() * width * 3 floats (local row j <-> frame row
 * j*shard_count + shard_index). Seeds depend only on (x, y, frame)
 * (scene.rs:99-101), so the union of shards is bit-identical to the full
 * frame. No data-path collective happens here; the caller gathers. */
int pt_render_shard_device(pt_scene *scene, const pt_params *params, const pt_camera *camera,
                           uint32_t frame_num, uint32_t shard_index, uint32_t shard_count,
                           float *d_rgb_shard_inout, uint64_t *d_ray_count, void *hip_stream);
uint32_t pt_shard_rows(uint32_t height, uint32_t shard_index, uint32_t shard_count);

/* ---- multi-GPU frames (SURVEY 8b "multi-GPU variant takes a device list / communicator", 8e) ----------------
 * One frame of Scene::update split over the ranks of an RCCL communicator: rank r renders the rows y with
 * y % world == r (disjoint pixels, scene.rs:90-93), the float3 shards are collected with ONE ncclAllGather (or
 * ncclGather to `root`) over xGMI and the per-rank ray counts are summed with an 8-byte ncclAllReduce
 * (scene.rs:118-120). No collective happens while rendering. One process per GPU (pt_comm_create, the unique id
 * travels over any host channel) or one process driving several GPUs (pt_comm_create_all with a device list).
 * Threading of the second form: RCCL requires that collectives which ONE thread issues for several communicators sit inside
 * one group, otherwise the first rank's call waits for peers the thread has not reached yet. pt_comm_gather_frame and
 * pt_render_sharded issue ONE rank's collectives (their own group): call them from one thread PER communicator -- or use the
 * *_all forms below, which take every rank of the clique and put all their collectives into a single group. */
typedef struct pt_comm pt_comm;
#define PT_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */

/* ncclGetUniqueId: called once (by rank 0), then handed to every rank's pt_comm_create. */
int pt_comm_unique_id(uint8_t id_out[PT_COMM_ID_BYTES]);
/* ncclCommInitRank on `device`; collective over the `world` ranks sharing `id`. */
int pt_comm_create(const uint8_t id[PT_COMM_ID_BYTES], uint32_t rank, uint32_t world, int device, pt_comm **comm_out);
/* ncclCommInitAll: `n` communicators of one process, comms_out[i] on devices[i] (rank i). */
int pt_comm_create_all(const int *devices, uint32_t n, pt_comm **comms_out);
void pt_comm_destroy(pt_comm *comm);
/* The RCCL the pt_comm_* functions run on: it is resolved at run time, the first time one of them is called -- the copy the
 * process has already loaded (a PyTorch process carries its own) or else the system's librccl.so.1; the render entry points
 * never need it. version_code = ncclGetVersion (e.g. 22606), path = the shared object. PT_ERR_UNSUPPORTED when there is none. */
int pt_comm_runtime(int *version_code_out, char *path_out, size_t path_capacity);
int pt_comm_rank(const pt_comm *comm, uint32_t *rank_out, uint32_t *world_out);

/* The exchange step alone, asynchronous on `hip_stream`: d_rgb_shard is this rank's compact shard as
 * pt_render_shard_device wrote it (pt_shard_rows(height, rank, world) rows); d_rgb_full (width*height*3 floats)
 * receives the whole frame on every rank (root < 0: ncclAllGather) or on rank `root` only (ncclGather; other ranks
 * may pass NULL); d_ray_count (8 bytes) is replaced by the sum over ranks on every rank. */
int pt_comm_gather_frame(pt_comm *comm, uint32_t width, uint32_t height, const float *d_rgb_shard,
                         float *d_rgb_full, uint64_t *d_ray_count, int root, void *hip_stream);

/* Scene::update on `world` GPUs: pack this rank's rows out of d_rgb_full_inout (the blend reads the previous frame,
 * scene.rs:114-116), pt_render_shard_device, pt_comm_gather_frame -- all enqueued on `hip_stream`. Every rank
 * passes the same params / camera / frame_num and a scene created from the same description on its own device;
 * d_rgb_full_inout must hold the same previous frame on every rank (it does after an all-gather call). */
int pt_render_sharded(pt_scene *scene, pt_comm *comm, const pt_params *params, const pt_camera *camera,
                      uint32_t frame_num, float *d_rgb_full_inout, uint64_t *d_ray_count, int root,
                      void *hip_stream);

/* The same for ALL ranks of a pt_comm_create_all clique from one thread: comms[i] must be rank i of n. Per rank i: scenes[i]
 * (a scene on comms[i]'s device), d_rgb_full_inout[i] / d_rgb_shards[i] / d_rgb_fulls[i], d_ray_counts[i] (8 bytes on that
 * device) and hip_streams[i] (NULL array: the default streams). Every rank's pack and render is enqueued first, then all ranks'
 * collectives inside ONE ncclGroupStart / ncclGroupEnd, then the unpack kernels. Buffers a rank does not need (the full frame
 * on ranks that do not receive it, root >= 0) may be NULL exactly as in the one-rank forms. */
int pt_render_sharded_all(pt_scene *const *scenes, pt_comm *const *comms, uint32_t n, const pt_params *params,
                          const pt_camera *camera, uint32_t frame_num, float *const *d_rgb_full_inout,
                          uint64_t *const *d_ray_counts, int root, void *const *hip_streams);
int pt_comm_gather_frame_all(pt_comm *const *comms, uint32_t n, uint32_t width, uint32_t height,
                             const float *const *d_rgb_shards, float *const *d_rgb_fulls, uint64_t *const *d_ray_counts,
                             int root, void *const *hip_streams);

/* The two layout kernels of the sharded path, exposed so that a single GPU can check them (and so that a caller
 * with its own transport can reuse them): full frame -> one rank's compact shard, and the gathered
 * [shard_count][ceil(height/shard_count)][width][3] buffer -> full frame (row y = gathered[y % N][y / N]). */
int pt_shard_pack(const float *d_rgb_full, float *d_rgb_shard, uint32_t width, uint32_t height,
                  uint32_t shard_index, uint32_t shard_count, void *hip_stream);
int pt_shard_unpack_all(const float *d_gathered, float *d_rgb_full, uint32_t width, uint32_t height,
                        uint32_t shard_count, void *hip_stream);

/* Base seed used when params->random_seed != 0 (scene.rs:96-97 uses
 * rand::random(), i.e. non-reproducible by design). */
int pt_scene_set_seed_base(pt_scene *scene, uint64_t seed_base);

/* Duration in ms of the trace kernel of the most recent render on this
 * handle, measured with HIP events on the launch stream (synchronises on the
 * stop event). Also the launch geometry, for roofline bookkeeping. */
int pt_last_kernel_ms(pt_scene *scene, float *ms_out);
/* Same for the whole pass of that render: the 1-spp pilot pass and the tile sort that order the work (when the
 * frame is large enough to use them) plus the frame kernel. */
int pt_last_pass_ms(pt_scene *scene, float *ms_out);
int pt_last_launch_info(pt_scene *scene, uint32_t *grid_out, uint32_t *block_out,
                        uint32_t *lds_bytes_out);
/* Host-side breakdown (ms, wall clock) of the most recent pt_render on a PAGEABLE buffer: out4 = { all-zero scan or copy-in
 * (runs under the measuring launch), wait for the GPU after the last enqueue, copy-out to the caller's pages, the whole call }. */
int pt_last_host_ms(pt_scene *scene, float out4[4]);

/* The traversal tree the library builds for itself in pt_scene_create (it never affects results): a
 * 4-wide tree over the sphere centres / sweeps, built ON THE DEVICE (level-synchronous: one global stable radix sort per
 * phase, csrc/pt_build.hip) -- it replaces, for traversal only, BVHNode::new (bvh.rs:64-94,268-333). build_ms = device time
 * of that build (HIP events), n_nodes / depth = its size, on_device = 0 when the host restatement ran instead
 * (PTGPU_HOST_BUILD=1, or no device scratch). pt_scene_debug_tree downloads the 128-byte nodes (tests compare the two
 * builders byte for byte). */
int pt_scene_build_info(pt_scene *scene, float *build_ms_out, uint32_t *n_nodes_out, uint32_t *depth_out, uint32_t *on_device_out);
int pt_scene_debug_tree(pt_scene *scene, void *nodes_out, size_t capacity_bytes);
/* The same tree as the kernels READ it: 64-byte nodes (csrc/pt_tree4.h DNode4Q: per axis the lower / upper planes of the four
 * children as f16 offsets from the node's min corner -- lower rounded down, upper rounded up --, the corner as three f32, one
 * word of child counts and pad constants). *usable_out = 0 when some node could not be packed (the kernels then walk the
 * binary tree). Tests check that every packed box contains the 128-byte node's box. */
int pt_scene_debug_tree_packed(pt_scene *scene, void *nodes_out, size_t capacity_bytes, uint32_t *usable_out);

/* Tuning (0 = library default for both arguments): workgroups resident per CU for the persistent grid, and a word of flags.
 * Two flags are meant for integrators:
 *   PT_TUNE_MEASURE_EVERY_FRAME  every frame measures its own work order. Default: a frame of the SAME view as the scene's last one (equal
 *        pt_params, pt_camera and shard) is ordered by the rays each 8x8 tile took in that last frame, measured by the frame kernel itself;
 *        a frame of a new view (from 12 samples on) runs as two launches, the first tracing the first sample of its pixels for real while it
 *        counts the rays per tile. Set it when consecutive frames are unrelated although their parameters are equal (a benchmark of
 *        "one frame of an unseen view"); the order of the work never changes a pixel or the ray count.
 *   PT_TUNE_NO_HANDOVER  no cooperative hand-over. Default on the wide (one workgroup per CU) list kernels: once the work list is dry, a wave
 *        that has run out of pixels finishes pixels handed over by waves that still have some, all 64 lanes on each ray (scene.rs:96-111
 *        makes a pixel one serial chain, and this shortens it). A pixel's RNG stream travels with it: who traces a pixel never changes it.
 * Every other bit is a development switch (A/B measurements, parity tests of the kernel variants -- all variants render identical frames);
 * they are listed in pathtrace-rs_amd/csrc/pt_devknobs.h and may change between versions. */
#define PT_TUNE_MEASURE_EVERY_FRAME 8192u
#define PT_TUNE_NO_HANDOVER 65536u
int pt_scene_set_tuning(pt_scene *scene, uint32_t blocks_per_cu, uint32_t variant);

/* Which kernel a frame runs on, and with what geometry (csrc/pt_select.h; DESIGN.md "kernel selection"). family: 0 general-world
 * kernel, 1 binary-tree kernel, 2 4-wide tree kernel (`name` starts with "grid<" when it walks the scene's uniform cell grid instead of the tree:
 * csrc/pt_grid.h), 3 MFMA list kernel, 4 exact scan from LDS, 5 exact scan from HBM/L2. */
typedef struct pt_kernel_choice {
    uint32_t family, block, lds_bytes, blocks_per_cu; /* threads per workgroup, dynamic LDS per workgroup, resident workgroups per CU (before the register clamp) */
    uint32_t moving, gate, verify, ref_bvh;           /* MOVING / GATE / VERIFY instantiation; BVHNode::ray_hit semantics */
    uint32_t ordered;                                 /* heavy-first work order (two launches for a new view) */
    uint32_t stack_in_lds, global_stack, n_tiles;     /* attenuation-stack slots in LDS, some levels in HBM, MFMA tiles */
    uint32_t world_hit_lds, world_occ, world_media;   /* general-world kernel: <BVH = ref_bvh, HIT_LDS, OCC, MEDIA> */
    uint32_t refill_min;
    uint32_t coop;                                    /* wide MFMA list kernels: waves that run out of work finish pixels handed over by busy ones, 64 lanes per ray (PT_TUNE_NO_HANDOVER switches it off) */
    uint32_t world_graph;                             /* general-world kernel: the world is a scene graph that does not flatten and is interpreted (csrc/pt_graph.h) */
    uint32_t world_lazy;                              /* general-world kernel, worlds with Noise textures: a scatter's Noise colour is formed when its path ends lit, by the whole wave (a development switch turns it off: csrc/pt_devknobs.h kVarWorldEager) */
    uint32_t pool_slots;                              /* 1024-thread MFMA list kernels: entries of each wave's pool of ready-to-start pixels in LDS (0: lanes wait for batched refills; `name` carries ",pool" otherwise) */
    char name[96];
} pt_kernel_choice;
/* The choice the scene's most recent render made. */
int pt_last_kernel_choice(pt_scene *scene, pt_kernel_choice *out);
/* The choice pt_render* WOULD make for a description (exactly one of sphere_desc / world_desc), computed on the host alone:
 * no device is touched, so the selection table can be checked on a machine without a GPU (tests/test_host_cpu.py).
 * shard_count > 1 selects for one shard of a multi-GPU frame. */
int pt_debug_select(const pt_scene_desc *sphere_desc, const pt_world_desc *world_desc, const pt_params *params,
                    const pt_camera *camera, uint32_t shard_count, uint32_t blocks_per_cu, uint32_t variant,
                    pt_kernel_choice *out);

/* The uniform cell grid pt_scene_create would plan for a sphere scene (csrc/pt_grid.h; host only, no device): the structure the tree kernels'
 * "grid<...>" flavour walks. info16 = { cells x, y, z, records, large spheres, as f32 bits: spheres registered per cell, fraction of the cells occupied, 0 | as f32 bits: box min x, y, z, cell size x, y, z, centre-to-origin
 * distance the registration is padded for (d_build), half diagonal of the field }. records5x4 (may be NULL): up to `capacity_records` records of
 * five 16-byte words -- spheres 0 | 1 and 2 | 3 interleaved component by component, then four list indices (0x7fffffff: empty; in the last word
 * 0x80000000 | record: the cell continues there). large (may be NULL): up to 16 list indices. Returns PT_ERR_UNSUPPORTED (with the plan's
 * reason in pt_last_error) when the scene gets no grid. tests/test_host_cpu.py checks the registration against the spheres themselves. */
int pt_debug_cell_grid(const pt_scene_desc *sphere_desc, uint32_t info16[16], uint32_t *records5x4, size_t capacity_records, uint32_t *large16);

/* The kernel instantiations that choice launches, as the symbols of their host-side launch stubs (frame kernel; measuring kernel, or ""
 * when the work is not ordered by one): for the thread's last successful pt_debug_select. Tests hold the union over many descriptions
 * against the stubs the shared object defines, so that no instantiation is carried that nothing selects. */
int pt_debug_last_kernel_symbols(char *frame_out, char *measure_out, size_t capacity);

/* Verify-mode counters of the MFMA prefilter (verify mode: csrc/pt_devknobs.h kVarVerify = 8): out4 = { exact-positive pairs the
 * prefilter failed to flag (must be 0), queued candidates, queue-overflow fallbacks, exact-positive
 * pairs }. Synchronises the device. */
int pt_scene_debug_counters(pt_scene *scene, uint64_t out4[4], int reset);

/* Traversal counters of the tree kernels in verify mode (verify mode, kVarVerify = 8, with use_bvh, or a list world large enough
 * to walk the internal tree): out2 = { nodes of the internal tree fetched, spheres tested exactly } summed over
 * all rays since the last reset (reported next to the oracle's counts for the caller's tree). */
int pt_scene_traversal_counters(pt_scene *scene, uint64_t out2[2], int reset);

/* Cooperative hand-over (csrc/pt_coop.h): out2 = { pixels handed over to idle waves, rays those waves traced } since the last
 * reset. Synchronises the device. */
int pt_scene_coop_counters(pt_scene *scene, uint64_t out2[2], int reset);

/* Rays per 8x8 WORK TILE of the handle's last frame -- the per-pixel summands of scene.rs:118's ray_count added up per tile, which the
 * kernels count anyway to order the next frame's work (tile t = (t / tiles_x, t % tiles_x), tile row 0 = the frame's bottom rows; a shard's
 * tiles cover its compact rows). rays_out may be NULL to ask for the sizes. PT_ERR_UNSUPPORTED when the last frame ran as one launch
 * (fewer than 12 samples, small frames) or under PT_TUNE_MEASURE_EVERY_FRAME: nothing was counted then. Synchronises the device.
 * Tests compare these counts tile by tile between kernels and with the oracle's full-frame fixture of BASELINE config 5. */
int pt_scene_debug_tile_rays(pt_scene *scene, uint32_t *rays_out, uint32_t capacity, uint32_t *n_tiles_out, uint32_t *tiles_x_out);

/* Closest-hit QUERIES on explicit rays, asynchronous on `hip_stream` -- the unit of the reference's own #[bench] functions (one
 * `ray_hit` on the centre ray of a preset: bench.rs:8-26, hitable_list.rs:68-75, spheres_soa.rs:464-485, bvh.rs:361-379) and the home of
 * SpheresSoA (collision/spheres_soa.rs:12-392), which only those benches call. Sphere / MovingSphere worlds. Not on the render
 * path; every mode is the reference's algorithm as written, one ray per lane (csrc/pt_query.hip):
 *   PT_QUERY_LIST        HitableList::ray_hit (hitable_list.rs:40-56; sphere.rs:29-66, moving_sphere.rs:38-73)
 *   PT_QUERY_BVH         BVHNode::ray_hit over the tree the scene was created with (bvh.rs:37-62, aabb.rs:46-58)
 *   PT_QUERY_SOA_SCALAR / _SSE4_1 / _AVX2   SpheresSoA::hit_scalar / hit_sse4_1 / hit_avx2 (:105-155 / :161-268 / :274-391). Their
 *                        arithmetic differs from Sphere::ray_hit's (no `a`, no division, normal * (1 / r), ties to the lowest LANE).
 * d_rays7: n_rays x (origin3, direction3, time) floats on the device. d_hits8: n_rays x (t, entry, point3, normal3); entry = the list
 * index as u32 bits, 0xffffffff for a miss (t = t_max then), 0xfffffffe when a BVH deeper than the walk's 48-entry stack was met. */
enum { PT_QUERY_LIST = 0, PT_QUERY_BVH = 1, PT_QUERY_SOA_SCALAR = 2, PT_QUERY_SOA_SSE4_1 = 3, PT_QUERY_SOA_AVX2 = 4 };
int pt_closest_hit(pt_scene *scene, uint32_t mode, uint32_t n_rays, const float *d_rays7, float t_min, float t_max,
                   float *d_hits8, void *hip_stream);

/* Device self-test probes (diagnostics for the parity tests; not part of the reference's
 * interface): evaluate one device primitive on n host inputs.
 *   PT_PROBE_POW5    out[i] = device x^5 used by schlick (math.rs:79 powf(x, 5.0))
 *   PT_PROBE_SIN/COS out[i] = sinf_cosf(in[i]) (simd.rs:107-208)
 *   PT_PROBE_RNG     in[i] reinterpreted as a u32 seed s; out[i] = the (i%16+1)-th gen::<f32>() of
 *                    Xoshiro256Plus::seed_from_u64(s)
 *   PT_PROBE_LN      out[i] = device f32::ln used by ConstantMedium (constant_medium.rs:60; glibc logf)
 *   PT_PROBE_SWEEP_SQRT  (n >= 2, `in` is not read) the kernels' shortened f32::sqrt against the compiler's correctly rounded
 *                    sqrtf on ALL 2^32 bit patterns: out[0] = number of mismatches, out[1] = bits of one mismatching input
 *   PT_PROBE_SWEEP_DRAWS the same report for the single-rounding forms of `2 * draw - 1`, `draw * 2 * PI` and `n + draw` against
 *                    the reference's expressions, for every draw k * 2^-24 beside 256 pixel coordinates n
 *   PT_PROBE_SWEEP_INVLEN the same report for the kernels' 1 / sqrt(t) of Vec3::normalize against `1.0f / sqrtf(t)`, all 2^32 t
 *   PT_PROBE_SWEEP_DIV   the same report for the normal's division by a radius whose reciprocal the host supplies, against `/`, on
 *                    2^32 seeded (numerator, radius) pairs including zeros, denormals, infinities and NaNs
 *   PT_PROBE_SWEEP_DIVA  the same report for the sphere test's quotients n / (d.d): the once-per-ray reciprocal of every a in [0.5, 2]
 *                    against `1.0f / a`, and the short division against `/` on 2^32 seeded (n, a) pairs (a within 64 ulps of 1 or
 *                    anywhere in [0.5, 2], n any bit pattern)
 *   PT_PROBE_SWEEP_RECIP the same report for the ray's reciprocal direction (ray.rs:14) against `1.0f / x`, all 2^32 x */
enum { PT_PROBE_POW5 = 0, PT_PROBE_SIN = 1, PT_PROBE_COS = 2, PT_PROBE_RNG = 3, PT_PROBE_LN = 4, PT_PROBE_SWEEP_SQRT = 5, PT_PROBE_SWEEP_DRAWS = 6, PT_PROBE_SWEEP_INVLEN = 7, PT_PROBE_SWEEP_DIV = 8,
       PT_PROBE_SWEEP_DIVA = 9, PT_PROBE_SWEEP_RECIP = 10 };
int pt_selftest_probe(int device, uint32_t probe, const float *in, float *out, size_t n);

/* Thread-local message describing the last error returned on this thread. */
const char *pt_last_error(void);

/* Library / kernel identification string: "ptgpu <version> gfx950 src <12 hex digits>[ defs <build defines>]". The hex digits are a hash
 * of the library's device and host sources as built (csrc, this header, the Makefile and its DEFS): bench.py writes the string into its line
 * and every committed profile records it, so a number can be tied to the build it was measured on. */
const char *pt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PTGPU_H */
