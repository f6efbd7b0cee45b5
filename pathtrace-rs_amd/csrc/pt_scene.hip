// pt_scene.hip -- Scene::new (scene.rs:25-31) behind the C ABI: pt_scene_create / pt_scene_create_world validate and analyse
// the description on the host (pt_prep.hip), upload the flattened layouts and build the internal traversal tree on the
// device (pt_build.hip). pt_scene_destroy releases everything.
#include "pt_host.h"

#include <new>

using namespace pthostside;

namespace {

int device_for(int device, hipDeviceProp_t *prop) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PT_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(PT_ERR_INVALID_ARG, "device %d out of range (%d devices)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipGetDeviceProperties(prop, device));
    return PT_OK;
}

// counters, events and the tuning defaults every scene has
int finish_scene(pt_scene *s) {
    if (hipMalloc((void **)&s->d_debug, 1024) != hipSuccess || hipMemset(s->d_debug, 0, 1024) != hipSuccess ||
        hipMalloc((void **)&s->d_work_counter, 64) != hipSuccess || hipMemset(s->d_work_counter, 0, 64) != hipSuccess || hipMalloc((void **)&s->d_ray_count, 64) != hipSuccess ||
        hipEventCreate(&s->ev_start) != hipSuccess || hipEventCreate(&s->ev_stop) != hipSuccess || hipEventCreate(&s->ev_pass) != hipSuccess)
        return fail(PT_ERR_HIP, "allocating counters / events failed");
    // (the uploads and clears above ran on the NULL stream, which a caller's non-blocking stream does not wait for)
    if (hipDeviceSynchronize() != hipSuccess) return fail(PT_ERR_HIP, "hipDeviceSynchronize failed");
    s->blocks_per_cu = dev_knobs().blocks_per_cu;
    s->variant = dev_knobs().variant;
    if (dev_knobs().timing && hipMalloc((void **)&s->d_wave_end, 65536 * 8 + 130000 * 32) != hipSuccess) s->d_wave_end = nullptr;   // (+ the hand-over log, pt_coop.h)
    return PT_OK;
}

// `motion` (optional, n_spheres entries): MovingSphere parameters of the entries that move; desc->spheres then holds
// centre_start / radius for them.
int create_sphere_scene(const pt_scene_desc *desc, const MotionIn *motion, int device, pt_scene **scene_out) {
    if (!desc || !scene_out) return fail(PT_ERR_INVALID_ARG, "desc/scene_out is NULL");
    *scene_out = nullptr;
    SpherePlan P;
    if (int rc = plan_sphere_scene(desc, motion, P)) return rc;
    hipDeviceProp_t prop;
    if (int rc = device_for(device, &prop)) return rc;

    pt_scene *s = new (std::nothrow) pt_scene();
    if (!s) return fail(PT_ERR_INVALID_ARG, "out of host memory");
    s->device = device;
    s->num_cus = prop.multiProcessorCount;
    s->tr = P.tr;
    s->n_materials = desc->n_materials;
    s->n_textures = desc->n_textures;
    s->bvh_root = desc->n_bvh_nodes ? desc->bvh_root : -1;
    s->has_sky = desc->has_sky ? 1u : 0u;
    memcpy(s->sky, desc->sky, sizeof s->sky);
    s->n_bvh_large = (uint32_t)P.bvh_large.size();
    s->has_tree_items = !P.titems.empty();
    s->h_spheres.assign(desc->spheres, desc->spheres + desc->n_spheres);
    if (P.has_motion) s->h_motion.assign(motion, motion + desc->n_spheres);
    s->h_t_lo = P.t_lo, s->h_t_hi = P.t_hi;
    const auto bail = [&](int rc) {
        pt_scene_destroy(s);
        return rc;
    };
    if (!P.titems.empty()) {
        // 4-wide tree: built ON THE DEVICE (pt_build.hip); the host restatement runs instead when the device build cannot
        // (no sort scratch) or PTGPU_HOST_BUILD is set (the one environment hook of the shipped library: the parity tests
        // compare the two builders byte for byte)
        const int brc = getenv("PTGPU_HOST_BUILD") ? -1
                                               : tree4_build_device(P.titems.data(), (uint32_t)P.titems.size(), nullptr, &s->d_nodes4, &s->tr.n_nodes4, &s->tr.depth4, &s->tree_build_ms);
        if (brc == 0) {
            s->tree_on_device = true;
        } else {
            const Tree4Host t4 = tree4_build_host(P.titems);
            s->tr.n_nodes4 = (uint32_t)t4.nodes.size(), s->tr.depth4 = t4.depth;
            if (upload(&s->d_nodes4, t4.nodes.data(), t4.nodes.size()) != PT_OK) return bail(fail(PT_ERR_HIP, "uploading the internal tree failed"));
        }
    }
    int rc = PT_OK;
    if ((rc = upload(&s->d_shade_rank, P.shade_rank.data(), P.shade_rank.size())) || (rc = upload(&s->d_leafrec, P.leafrec.data(), P.leafrec.size())) ||
        (rc = upload(&s->d_spheres, P.sph.data(), P.sph.size())) || (rc = upload(&s->d_spheres_r2, P.sph_r2.data(), P.sph_r2.size())) ||
        (rc = upload(&s->d_shade, P.shade.data(), P.shade.size())) || (rc = upload(&s->d_sphere_mat, desc->sphere_material, desc->n_spheres)) ||
        (rc = upload(&s->d_mats, P.mats.data(), P.mats.size())) || (rc = upload(&s->d_texs, P.texs.data(), P.texs.size())) ||
        (rc = upload(&s->d_perlin_vec, P.pvec.data(), P.pvec.size())) || (rc = upload(&s->d_perlin_perm, P.pperm.data(), P.pperm.size())) ||
        (rc = upload(&s->d_leaf_rank, P.leaf_rank.data(), P.leaf_rank.size())) || (rc = upload(&s->d_gate, P.gate.data(), P.gate.size())) ||
        (rc = upload(&s->d_gate_chain, P.gate_chain.data(), P.gate_chain.size())) || (rc = upload(&s->d_bvh_large, P.bvh_large.data(), P.bvh_large.size())) ||
        (rc = upload(&s->d_rank_sphere, P.rank_sphere.data(), P.rank_sphere.size())))
        return bail(rc);
    if (s->d_nodes4) {   // the nodes the kernels read + the leaves' slot records, from what is on the device now
        const int prc = tree4_pack_device(s->d_nodes4, s->tr.n_nodes4, s->d_spheres, desc->n_bvh_nodes ? s->d_leafrec : nullptr, nullptr, &s->d_nodes4q, &s->d_slotrec,
                                          &s->tr.tree4_packed);
        if (prc != 0) return bail(fail(PT_ERR_HIP, "packing the internal tree failed (hipError %d)", prc));
    }
    if (P.has_motion && (rc = upload(&s->d_motion, P.mot.data(), P.mot.size()))) return bail(rc);
    if (P.grid.ok) {   // uniform cell grid (csrc/pt_grid.h): tr.grid_ok was set by the plan
        const GridPlan &g = P.grid;
        if ((rc = upload(&s->d_grid_cells, g.cells.data(), g.cells.size())) || (rc = upload(&s->d_grid_large, g.large.data(), g.large.size())) ||
            (rc = upload(&s->d_grid_rec, g.rec.data(), g.rec.size())))
            return bail(rc);
        pthostside_grid_geom &q = s->grid_geom;
        for (int k = 0; k < 3; ++k) q.n[k] = g.n[k], q.gmin[k] = g.gmin[k], q.centre[k] = g.centre[k];
        q.n_records = g.n_records, q.n_large = (uint32_t)g.large.size();
        q.h = g.h, q.half_diag = g.half_diag, q.d_build = g.d_build;
        for (int k = 0; k < 3; ++k) q.ha[k] = g.ha[k];
    }
    // the caller's tree as given: only pt_closest_hit(PT_QUERY_BVH) walks it (the frame kernels use the gates and ranks derived from it)
    if (desc->n_bvh_nodes && (rc = upload(&s->d_ref_nodes, desc->bvh_nodes, desc->n_bvh_nodes))) return bail(rc);
    if (P.has_prep) {
        const MfmaPrep &prep = P.prep;
        if ((rc = upload(&s->d_afrag, prep.afrag.data(), prep.afrag.size() / 8)) || (rc = upload(&s->d_tile_sphere, prep.tile_sphere.data(), prep.tile_sphere.size())) ||
            (rc = upload(&s->d_large, prep.large.data(), prep.large.size())))
            return bail(rc);
        if (prep.cull_axis < 3u) {
            if ((rc = upload(&s->d_cull_tab, prep.cull_tab.data(), prep.cull_tab.size()))) return bail(rc);
            s->cull_axis = prep.cull_axis, s->cull_always = prep.cull_always;
            s->cull_u0 = prep.cull_u0, s->cull_inv_cell = prep.cull_inv_cell;
            s->cull_axis2 = prep.cull_axis2, s->cull_u0_2 = prep.cull_u0_2, s->cull_inv_cell_2 = prep.cull_inv_cell_2;
            s->cull_rmin = prep.cull_rmin, s->cull_rmax = prep.cull_rmax, s->rs_small = (float)prep.rs;
            memcpy(s->clip_min, prep.clip_min, 12), memcpy(s->clip_max, prep.clip_max, 12);
        }
        s->n_large = (uint32_t)prep.large.size();
        s->large0 = prep.large.empty() ? 0xffffffffu : prep.large[0];
        memcpy(s->c0, prep.c0, sizeof s->c0);
        s->rs2 = (float)(prep.rs * prep.rs * 1.0001);
        // margin = a * (m0 + gamma * (|o - c0|^2 + Rs^2)); see DESIGN.md for the derivation. A swept bound of half-length h moves
        // the reference's rounding slack from radius r to r + h: scale by (1 + h/r)
        const double widen = 1.0 + 1.5 * prep.sweep_ratio;
        s->m0 = (float)((1.0e-5 * prep.rs * prep.rs + 1.0e-4) * widen);
        s->gamma = (float)(8.0e-6 * widen);
    }
    if ((rc = finish_scene(s))) return bail(rc);
    *scene_out = s;
    return PT_OK;
}

}  // namespace

extern "C" int pt_scene_create(const pt_scene_desc *desc, int device, pt_scene **scene_out) {
    if (desc && scene_out && desc->n_spheres == 0) {  // an empty HitableList is a valid world: every ray sees the sky
        pt_world_desc w{};
        w.n_materials = desc->n_materials, w.materials = desc->materials;
        w.n_textures = desc->n_textures, w.textures = desc->textures, w.perlin = desc->perlin;
        w.n_bvh_nodes = desc->n_bvh_nodes, w.bvh_nodes = desc->bvh_nodes, w.bvh_root = desc->bvh_root;
        w.has_sky = desc->has_sky;
        memcpy(w.sky, desc->sky, sizeof w.sky);
        return pt_scene_create_world(&w, device, scene_out);
    }
    return create_sphere_scene(desc, nullptr, device, scene_out);
}

// ---- general worlds --------------------------------------------------------------------------------
extern "C" int pt_scene_create_world(const pt_world_desc *given, int device, pt_scene **scene_out) {
    if (!given || !scene_out) return fail(PT_ERR_INVALID_ARG, "desc/scene_out is NULL");
    *scene_out = nullptr;
    FlatWorld flat;   // a scene graph is flattened into the list form first (pt_prep.hip); plain descriptions pass through
    const pt_world_desc *desc = given;
    if (int rc = flatten_world_graph(given, flat, &desc)) return rc;
    WorldAsSpheres W;
    if (int rc = analyze_world(desc, W)) return rc;
    if (W.sphere_like) {
        // Sphere / MovingSphere entries only: the specialised kernels apply (MFMA prefilter, internal tree); with moving
        // entries their MOVING instantiations, and the general-world data rides along as the fallback
        if (W.all_spheres) return create_sphere_scene(&W.desc, nullptr, device, scene_out);
        int rc = create_sphere_scene(&W.desc, W.motion.data(), device, scene_out);
        if (rc == PT_OK) {
            pt_scene *s = *scene_out;
            s->tr.n_hitables = desc->n_hitables;
            s->tr.n_world_xf = desc->n_transforms;
            s->tr.ref_bvh_depth = W.ref_depth;
            if (W.ref_depth + 2 > 64u || (rc = upload(&s->d_hitables, desc->hitables, desc->n_hitables)) ||
                (rc = upload(&s->d_transforms, desc->transforms, desc->n_transforms)) || (!s->d_ref_nodes && (rc = upload(&s->d_ref_nodes, desc->bvh_nodes, desc->n_bvh_nodes)))) {
                pt_scene_destroy(s);
                *scene_out = nullptr;
                return rc ? rc : fail(PT_ERR_UNSUPPORTED, "BVH depth %u exceeds the traversal stack", W.ref_depth);
            }
            return PT_OK;
        }
        if (rc != PT_ERR_UNSUPPORTED) return rc;
        // (unsupported by the specialised path, e.g. a degenerate time interval: trace it as a general world)
    }
    if (W.ref_depth + 2 > 64u) return fail(PT_ERR_UNSUPPORTED, "BVH depth %u exceeds the traversal stack", W.ref_depth);

    hipDeviceProp_t prop;
    if (int rc = device_for(device, &prop)) return rc;
    pt_scene *s = new (std::nothrow) pt_scene();
    if (!s) return fail(PT_ERR_INVALID_ARG, "out of host memory");
    s->device = device;
    s->num_cus = prop.multiProcessorCount;
    world_traits(desc, W, s->tr);
    s->n_materials = desc->n_materials;
    s->n_textures = desc->n_textures;
    s->bvh_root = desc->n_bvh_nodes ? desc->bvh_root : -1;
    s->has_sky = desc->has_sky ? 1u : 0u;
    memcpy(s->sky, desc->sky, sizeof s->sky);
    std::vector<DMat> mats(desc->n_materials);
    for (uint32_t i = 0; i < desc->n_materials; ++i) {
        const pt_material &m = desc->materials[i];
        mats[i] = DMat{m.kind, m.albedo[0], m.albedo[1], m.albedo[2], m.param, m.texture, 0.f, 0.f};
        // a Constant texture is resolved here: the general kernel then needs no texture call for it (pad0 = 1)
        if ((m.kind == PT_MAT_LAMBERTIAN || m.kind == PT_MAT_DIFFUSE_LIGHT || m.kind == PT_MAT_ISOTROPIC) && desc->textures[m.texture].kind == PT_TEX_CONSTANT) {
            const pt_texture &t = desc->textures[m.texture];
            mats[i].a0 = t.color[0], mats[i].a1 = t.color[1], mats[i].a2 = t.color[2], mats[i].pad0 = 1.0f;
        }
    }
    std::vector<DTex> texs(desc->n_textures ? desc->n_textures : 1);
    for (uint32_t i = 0; i < desc->n_textures; ++i) {
        const pt_texture &t = desc->textures[i];
        texs[i] = DTex{t.kind, t.color[0], t.color[1], t.color[2], t.odd, t.even, t.scale, 0.f};
    }
    std::vector<float4> pvec(256, make_float4(0, 0, 0, 0));
    std::vector<uint32_t> pperm(768, 0);
    if (desc->perlin) {
        for (int i = 0; i < 256; ++i) {
            pvec[i] = make_float4(desc->perlin->randvec[i][0], desc->perlin->randvec[i][1], desc->perlin->randvec[i][2], 0.f);
            pperm[i] = desc->perlin->perm_x[i];
            pperm[256 + i] = desc->perlin->perm_y[i];
            pperm[512 + i] = desc->perlin->perm_z[i];
        }
    }
    int rc = PT_OK;
    const auto bail = [&](int code) {
        pt_scene_destroy(s);
        return code;
    };
    if (W.has_image) {
        std::vector<uint4> table(desc->n_images);
        std::vector<uint8_t> blob;
        for (uint32_t i = 0; i < desc->n_images; ++i) {
            const pt_image &im = desc->images[i];
            table[i] = make_uint4((uint32_t)blob.size(), im.width, im.height, 0u);
            blob.insert(blob.end(), im.rgb, im.rgb + 3ull * im.width * im.height);
        }
        if (blob.size() > 0xf0000000ull) return bail(fail(PT_ERR_UNSUPPORTED, "image textures exceed 3.75 GB"));
        if ((rc = upload(&s->d_image_table, table.data(), table.size())) || (rc = upload(&s->d_image_bytes, blob.data(), blob.size()))) return bail(rc);
    }
    if ((rc = upload(&s->d_hitables, desc->hitables, desc->n_hitables)) || (rc = upload(&s->d_transforms, desc->transforms, desc->n_transforms)) ||
        (rc = upload(&s->d_ref_nodes, desc->bvh_nodes, desc->n_bvh_nodes)) || (rc = upload(&s->d_mats, mats.data(), mats.size())) ||
        (rc = upload(&s->d_texs, texs.data(), texs.size())) || (rc = upload(&s->d_perlin_vec, pvec.data(), pvec.size())) ||
        (rc = upload(&s->d_perlin_perm, pperm.data(), pperm.size())))
        return bail(rc);
    if (W.is_graph) {   // interpreted: the graph itself goes to the device (pt_graph.h); pt_node is four 32-bit words
        static_assert(sizeof(pt_node) == sizeof(uint4), "pt_node rows are read as uint4");
        s->groot = desc->root_node;
        if ((rc = upload(&s->d_gnodes, reinterpret_cast<const uint4 *>(desc->nodes), desc->n_nodes)) ||
            (rc = upload(&s->d_gchildren, desc->node_children, desc->n_node_children)))
            return bail(rc);
    }
    if ((rc = finish_scene(s))) return bail(rc);
    *scene_out = s;
    return PT_OK;
}

extern "C" void pt_scene_destroy(pt_scene *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    void *const dev[] = {s->d_hitables, s->d_motion, s->d_transforms, s->d_ref_nodes, s->d_image_table, s->d_image_bytes, s->d_spheres, s->d_spheres_r2,
                         s->d_shade, s->d_sphere_mat, s->d_mats, s->d_texs, s->d_perlin_vec, s->d_perlin_perm, s->d_gate, s->d_gate_chain, s->d_bvh_large,
                         s->d_wnodes, s->d_nodes4, s->d_nodes4q, s->d_slotrec, s->d_rank_sphere, s->d_leafrec, s->d_shade_rank, s->d_leaf_rank, s->d_afrag,
                         s->d_tile_sphere, s->d_cull_tab, s->d_large, s->d_debug, s->d_tile_buf, s->d_px_state, s->d_work_counter, s->d_ray_count,
                         s->d_frame, s->d_gstack, s->d_wave_end, s->d_tail_box, s->d_gnodes, s->d_gchildren, s->d_gframes, s->d_grid_cells, s->d_grid_large, s->d_grid_rec};
    for (void *p : dev) (void)hipFree(p);
    (void)hipHostFree(s->h_stage);
    if (s->ev_start) (void)hipEventDestroy(s->ev_start);
    if (s->ev_stop) (void)hipEventDestroy(s->ev_stop);
    if (s->ev_pass) (void)hipEventDestroy(s->ev_pass);
    delete s;
}
