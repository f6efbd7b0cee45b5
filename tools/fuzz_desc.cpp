// fuzz_desc.cpp -- seeded fuzzer of the library's HOST-SIDE validators: malformed pt_scene_desc / pt_world_desc / pt_node / pt_params
// inputs through pt_debug_select (analysis + flattening + kernel selection, no device touched), on a build of the whole library made
// host-only with AddressSanitizer + UndefinedBehaviorSanitizer (tools/sanitize_cpu.sh). Every description must come back either
// accepted (PT_OK, a kernel chosen) or refused with PT_ERR_INVALID_ARG / PT_ERR_UNSUPPORTED and a message; anything else -- a crash, a
// sanitizer report, a run-away (alarm), another code -- is a defect. Arrays are allocated at exactly the size the description states,
// so an index the validators let through shows up as a heap overflow.
//   usage: fuzz_desc [cases = 100000] [first seed = 1]
#include <signal.h>
#include <unistd.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "ptgpu.h"

namespace {

struct Rng {
    uint64_t s;
    uint32_t next() {
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        return (uint32_t)(s >> 32);
    }
    uint32_t below(uint32_t n) { return n ? next() % n : 0u; }
    bool chance(uint32_t percent) { return below(100) < percent; }
    float uni(float a, float b) { return a + (b - a) * (float)(next() >> 8) * (1.0f / 16777216.0f); }
};

// exactly-sized heap copies (so that ASan sees every out-of-range read); an empty vector becomes a 1-byte block, never NULL
template <typename T>
T *exact(const std::vector<T> &v, std::vector<void *> &owned) {
    T *p = (T *)malloc(v.size() ? v.size() * sizeof(T) : 1);
    if (v.size()) memcpy(p, v.data(), v.size() * sizeof(T));
    owned.push_back(p);
    return p;
}

const uint32_t kBadIndex[] = {0x7fffffffu, 0x80000000u, 0xffffffffu, 0xfffffffeu, 0x40000000u, 0x00100000u, 0x000fffffu, 0x01000000u, 0x0f0fffffu};
const float kBadFloat[] = {NAN, INFINITY, -INFINITY, 0.0f, -0.0f, -1.0f, 1.0e-30f, 3.0e38f, -3.0e38f, 1.0e-45f};

struct Case {
    std::vector<pt_sphere> spheres;
    std::vector<uint32_t> sphere_mat;
    std::vector<pt_hitable> hitables;
    std::vector<pt_affine> transforms;
    std::vector<pt_material> materials;
    std::vector<pt_texture> textures;
    std::vector<pt_bvh_node> bvh;
    std::vector<pt_image> images;
    std::vector<std::vector<uint8_t>> image_bytes;
    std::vector<pt_node> nodes;
    std::vector<uint32_t> children;
    pt_perlin perlin;
    bool has_perlin = false, null_perlin = false;
    int32_t bvh_root = -1;
    uint32_t root_node = 0;
    uint32_t has_sky = 0;
    float sky[3] = {0.5f, 0.6f, 0.8f};
    pt_params params{64, 48, 8, 10, 0, 0};
    pt_camera cam{};
    uint32_t shard_count = 1, variant = 0, blocks = 0;
    uint32_t null_mask = 0;     // bit i: pass NULL for array i although its count is not zero
};

void tables(Rng &r, Case &c, bool world) {
    const uint32_t nt = 1 + r.below(6);
    for (uint32_t i = 0; i < nt; ++i) {
        pt_texture t{};
        const uint32_t k = r.below(world ? 10 : 8);
        t.kind = k < 5 ? PT_TEX_CONSTANT : (k < 7 ? PT_TEX_CHECKER : (k < 9 ? PT_TEX_NOISE : PT_TEX_IMAGE));
        for (int q = 0; q < 3; ++q) t.color[q] = r.uni(0.0f, 1.0f);
        t.odd = t.even = -1;
        if (t.kind == PT_TEX_CHECKER) t.odd = (int32_t)r.below(i ? i : 1), t.even = (int32_t)r.below(i ? i : 1), t.kind = i ? t.kind : PT_TEX_CONSTANT;   // (children come earlier: no cycle)
        if (t.kind == PT_TEX_NOISE) t.scale = r.uni(0.5f, 8.0f), c.has_perlin = true;
        if (t.kind == PT_TEX_IMAGE) {
            pt_image im{1 + r.below(4), 1 + r.below(4), nullptr};
            c.image_bytes.emplace_back((size_t)im.width * im.height * 3u, (uint8_t)r.below(256));
            c.images.push_back(im);
            t.odd = (int32_t)c.images.size() - 1;
        }
        c.textures.push_back(t);
    }
    const uint32_t nm = 1 + r.below(6);
    for (uint32_t i = 0; i < nm; ++i) {
        pt_material m{};
        m.kind = r.below(world ? 5 : 4);
        for (int q = 0; q < 3; ++q) m.albedo[q] = r.uni(0.0f, 1.0f);
        m.param = m.kind == PT_MAT_DIELECTRIC ? r.uni(1.1f, 2.0f) : r.uni(0.0f, 0.5f);
        m.texture = (m.kind == PT_MAT_LAMBERTIAN || m.kind == PT_MAT_DIFFUSE_LIGHT || m.kind == PT_MAT_ISOTROPIC) ? (int32_t)r.below(nt) : -1;
        c.materials.push_back(m);
    }
    if (c.has_perlin) {
        for (int i = 0; i < 256; ++i) {
            for (int q = 0; q < 3; ++q) c.perlin.randvec[i][q] = r.uni(-1.0f, 1.0f);
            c.perlin.perm_x[i] = (uint32_t)i, c.perlin.perm_y[i] = (uint32_t)(255 - i), c.perlin.perm_z[i] = (uint32_t)((i * 7) & 255);
        }
    }
}

// a median-split tree over n leaves (leaf j = ~j), boxes arbitrary but finite: what a caller's BVHNode::new would hand over
int32_t build_bvh(Rng &r, Case &c, uint32_t lo, uint32_t hi) {
    if (hi - lo == 1) return ~(int32_t)lo;
    const int32_t me = (int32_t)c.bvh.size();
    c.bvh.emplace_back();
    const uint32_t mid = lo + (hi - lo) / 2;
    const int32_t l = build_bvh(r, c, lo, mid), rr = build_bvh(r, c, mid, hi);
    pt_bvh_node n{};
    for (int q = 0; q < 3; ++q) n.min[q] = r.uni(-20.0f, -1.0f), n.max[q] = r.uni(1.0f, 20.0f);
    n.lhs = l, n.rhs = rr;
    c.bvh[(size_t)me] = n;
    return me;
}

void sphere_case(Rng &r, Case &c) {
    tables(r, c, false);
    const bool field = r.chance(4);   // an even, dense field of equal spheres (a jittered lattice like BASELINE config 5's): the cell grid's territory
    const uint32_t side = 33u + r.below(10);
    const uint32_t n = field ? side * side : (r.chance(5) ? 0u : (r.chance(10) ? 700u + r.below(400) : 1u + r.below(60)));
    for (uint32_t i = 0; i < n; ++i) {
        if (field) c.spheres.push_back(pt_sphere{0.5f * (float)(i % side) + r.uni(0.0f, 0.3f), 0.2f, 0.5f * (float)(i / side) + r.uni(0.0f, 0.3f), 0.2f});
        else
        c.spheres.push_back(pt_sphere{r.uni(-8, 8), r.uni(0, 2), r.uni(-8, 8), r.chance(5) ? -r.uni(0.1f, 0.6f) : r.uni(0.1f, 0.6f)});
        uint32_t m = r.below((uint32_t)c.materials.size());
        c.sphere_mat.push_back(m);
    }
    if (n && r.chance(50)) {
        const int32_t root = build_bvh(r, c, 0, n);
        if (root < 0) {   // bvh.rs:73-79: one hitable, lhs == rhs
            pt_bvh_node b{{-1, -1, -1}, {1, 1, 1}, ~0, ~0};
            c.bvh.push_back(b);
            c.bvh_root = 0;
        } else {
            c.bvh_root = root;
        }
        c.params.use_bvh = r.chance(70);
    }
}

void world_case(Rng &r, Case &c, bool graph) {
    tables(r, c, true);
    const bool spheres_only = !graph && r.chance(25);          // Sphere + MovingSphere worlds run on the sphere kernels' MOVING instantiations
    const bool field = spheres_only && r.chance(15);   // (as in sphere_case: a dense lattice, here with a few MovingSphere entries)
    const uint32_t side = 33u + r.below(10);
    const uint32_t n = field ? side * side : r.chance(5) ? 0u : (r.chance(spheres_only ? 25 : 8) ? 500u + r.below(1500) : (spheres_only && r.chance(60) ? 33u + r.below(600) : 1u + r.below(40)));
    const uint32_t nx = spheres_only ? 0u : r.below(5);
    for (uint32_t i = 0; i < nx; ++i) {
        pt_affine a{};
        for (int q = 0; q < 12; ++q) a.m[q] = (q % 4 == q / 3 || q == 0 || q == 4 || q == 8) ? 1.0f : r.uni(-0.3f, 0.3f), a.inv[q] = a.m[q];
        c.transforms.push_back(a);
    }
    for (uint32_t i = 0; i < n; ++i) {
        pt_hitable h{};
        h.kind = spheres_only ? r.below(2) : r.below(6);
        h.material = r.below((uint32_t)c.materials.size());
        for (uint32_t tries = 0; tries < 8 && c.materials[h.material].kind == PT_MAT_ISOTROPIC; ++tries) h.material = r.below((uint32_t)c.materials.size());   // (a phase function is no surface)
        h.flip_normals = r.below(2);
        h.transform = (!graph && nx && r.chance(30)) ? (int32_t)r.below(nx) : -1;
        h.medium_material = -1;
        if (!graph && !spheres_only && r.chance(15)) {
            for (uint32_t m = 0; m < c.materials.size(); ++m)
                if (c.materials[m].kind == PT_MAT_ISOTROPIC) h.medium_material = (int32_t)m;
            h.density = r.uni(0.01f, 0.5f);
        }
        for (int q = 0; q < 10; ++q) h.p[q] = r.uni(-3.0f, 3.0f);
        if (spheres_only) h.p[0] = r.uni(-8, 8), h.p[1] = r.uni(0, 2), h.p[2] = r.uni(-8, 8);
        if (h.kind == PT_HIT_SPHERE) h.p[3] = r.uni(0.1f, 1.0f);
        if (h.kind == PT_HIT_MOVING_SPHERE) h.p[3] = 0.0f, h.p[4] = r.uni(0.0f, 0.5f), h.p[5] = 0.0f, h.p[6] = r.uni(0.1f, 1.0f), h.p[7] = 0.0f, h.p[8] = 1.0f;
        if (field) {
            h.kind = r.chance(3) ? PT_HIT_MOVING_SPHERE : PT_HIT_SPHERE;
            h.p[0] = 0.5f * (float)(i % side) + r.uni(0.0f, 0.3f), h.p[1] = 0.2f, h.p[2] = 0.5f * (float)(i / side) + r.uni(0.0f, 0.3f);
            if (h.kind == PT_HIT_SPHERE) h.p[3] = 0.2f;
            else h.p[3] = 0.0f, h.p[4] = r.uni(0.0f, 0.1f), h.p[5] = 0.0f, h.p[6] = 0.2f, h.p[7] = 0.0f, h.p[8] = 1.0f;
        }
        c.hitables.push_back(h);
    }
    if (graph && n) {
        // leaves first, then wrappers over random earlier nodes (a DAG: hitable.rs holds references), the root last
        for (uint32_t i = 0; i < n; ++i) c.nodes.push_back(pt_node{PT_NODE_HITABLE, i, 0, 0.0f});
        const uint32_t extra = 1 + r.below(12);
        for (uint32_t e = 0; e < extra; ++e) {
            const uint32_t have = (uint32_t)c.nodes.size();
            const uint32_t k = r.below(10);
            if (k < 5) {
                const uint32_t cnt = 1 + r.below(4);
                pt_node nd{PT_NODE_LIST, (uint32_t)c.children.size(), cnt, 0.0f};
                for (uint32_t j = 0; j < cnt; ++j) c.children.push_back(r.below(have));
                c.nodes.push_back(nd);
            } else if (k < 7 && nx) {
                c.nodes.push_back(pt_node{PT_NODE_INSTANCE, r.below(nx), r.below(have), 0.0f});
            } else if (k < 9) {
                uint32_t iso = 0xffffffffu;
                for (uint32_t m = 0; m < c.materials.size(); ++m)
                    if (c.materials[m].kind == PT_MAT_ISOTROPIC) iso = m;
                if (iso != 0xffffffffu) c.nodes.push_back(pt_node{PT_NODE_MEDIUM, iso, r.below(have), r.uni(0.05f, 0.5f)});
            } else if (have >= 2) {
                pt_bvh_node b{{-5, -5, -5}, {5, 5, 5}, (int32_t)r.below(have), (int32_t)r.below(have)};
                c.bvh.push_back(b);
                c.nodes.push_back(pt_node{PT_NODE_BVH, (uint32_t)c.bvh.size() - 1u, 0, 0.0f});
            }
        }
        // the root: a list over a few of the nodes
        const uint32_t cnt = r.chance(10) ? 400u + r.below(1200) : 1 + r.below(5);   // (now and then more list entries than the LDS holds records)
        pt_node root{PT_NODE_LIST, (uint32_t)c.children.size(), cnt, 0.0f};
        const bool with_bvh = r.chance(30);   // a BVH over the root list: its leaves index the list's children, each of which must flatten to ONE entry
        std::vector<uint32_t> kids;
        for (uint32_t j = 0; j < cnt; ++j) {
            if (!with_bvh) {
                kids.push_back(r.below((uint32_t)c.nodes.size()));
                continue;
            }
            uint32_t node = r.below(n);                                     // a leaf shape ...
            for (uint32_t lv = r.below(3); lv > 0 && nx; --lv) {            // ... below up to two Instance levels
                c.nodes.push_back(pt_node{PT_NODE_INSTANCE, r.below(nx), node, 0.0f});
                node = (uint32_t)c.nodes.size() - 1u;
            }
            kids.push_back(node);
        }
        root.a = (uint32_t)c.children.size();
        for (uint32_t k : kids) c.children.push_back(k);
        c.nodes.push_back(root);
        c.root_node = (uint32_t)c.nodes.size() - 1u;
        if (with_bvh) {
            const int32_t top = build_bvh(r, c, 0, cnt);
            if (top < 0) {
                pt_bvh_node b{{-1, -1, -1}, {1, 1, 1}, ~0, ~0};
                c.bvh.push_back(b);
                c.bvh_root = 0;
            } else {
                c.bvh_root = top;
            }
            c.params.use_bvh = r.chance(70);
        }
    } else if (n && r.chance(50)) {
        const int32_t root = build_bvh(r, c, 0, n);
        if (root < 0) {
            pt_bvh_node b{{-1, -1, -1}, {1, 1, 1}, ~0, ~0};
            c.bvh.push_back(b);
            c.bvh_root = 0;
        } else {
            c.bvh_root = root;
        }
        c.params.use_bvh = r.chance(70);
    }
}

template <typename T>
T &pick(Rng &r, std::vector<T> &v) { return v[r.below((uint32_t)v.size())]; }

void mutate(Rng &r, Case &c, bool world) {
    const uint32_t bad = kBadIndex[r.below(sizeof kBadIndex / sizeof *kBadIndex)];
    const float bf = kBadFloat[r.below(sizeof kBadFloat / sizeof *kBadFloat)];
    switch (r.below(34)) {
    case 0: if (!c.sphere_mat.empty()) pick(r, c.sphere_mat) = r.chance(50) ? (uint32_t)c.materials.size() : bad; break;
    case 1: if (!c.materials.empty()) pick(r, c.materials).texture = r.chance(50) ? (int32_t)c.textures.size() : (int32_t)bad; break;
    case 2: if (!c.materials.empty()) pick(r, c.materials).kind = r.chance(50) ? 5u : bad; break;
    case 3: if (!c.textures.empty()) pick(r, c.textures).kind = r.chance(50) ? 4u : bad; break;
    case 4: if (!c.textures.empty()) { pt_texture &t = pick(r, c.textures); t.kind = PT_TEX_CHECKER; t.odd = (int32_t)(&t - c.textures.data()); t.even = r.chance(50) ? t.odd : (int32_t)bad; } break;   // a Checker of itself
    case 5: if (c.textures.size() >= 2) { c.textures[0].kind = c.textures[1].kind = PT_TEX_CHECKER; c.textures[0].odd = c.textures[0].even = 1; c.textures[1].odd = c.textures[1].even = 0; } break;   // a two-texture cycle
    case 6: if (!c.textures.empty()) { pt_texture &t = pick(r, c.textures); t.kind = PT_TEX_IMAGE; t.odd = r.chance(50) ? (int32_t)c.images.size() : (int32_t)bad; } break;
    case 7: if (!c.textures.empty()) { pt_texture &t = pick(r, c.textures); t.kind = PT_TEX_NOISE; t.scale = bf; c.null_perlin = r.chance(50); } break;
    case 8: if (!c.spheres.empty()) { pt_sphere &s = pick(r, c.spheres); (r.chance(50) ? s.radius : s.cx) = bf; } break;
    case 9: if (!c.bvh.empty()) { pt_bvh_node &b = pick(r, c.bvh); (r.chance(50) ? b.lhs : b.rhs) = r.chance(50) ? (int32_t)c.bvh.size() : (int32_t)bad; } break;
    case 10: if (!c.bvh.empty()) { pt_bvh_node &b = pick(r, c.bvh); b.lhs = b.rhs = (int32_t)(&b - c.bvh.data()); } break;   // a node that is its own child
    case 11: if (!c.bvh.empty()) { pt_bvh_node &b = pick(r, c.bvh); b.lhs = ~(int32_t)(world ? c.hitables.size() : c.spheres.size()); } break;   // leaf index == count
    case 12: c.bvh_root = r.chance(50) ? (int32_t)c.bvh.size() : (int32_t)bad; break;
    case 13: if (!c.bvh.empty()) { pt_bvh_node &b = pick(r, c.bvh); b.min[r.below(3)] = bf; b.max[r.below(3)] = bf; } break;
    case 14: c.params.width = r.chance(50) ? 0u : (r.chance(50) ? 65536u : bad); break;
    case 15: c.params.height = r.chance(50) ? 0u : (r.chance(50) ? 65536u : bad); break;
    case 16: c.params.samples = r.chance(50) ? 0u : (r.chance(50) ? (1u << 20) : bad); break;
    case 17: c.params.max_depth = r.chance(30) ? 0u : (r.chance(50) ? 4096u : bad); break;
    case 18: c.params.use_bvh = 1; break;
    case 19: c.shard_count = r.chance(50) ? 0u : (r.chance(50) ? 65537u : bad); break;
    case 20: c.null_mask |= 1u << r.below(12); break;
    case 21: c.variant = r.next(); c.blocks = r.chance(50) ? r.below(64) : bad; break;
    case 22: if (!c.hitables.empty()) {   // an unknown kind -- or a medium group (PT_HIT_MEDIUM_GROUP) with a plausible / wild member count
        pt_hitable &h = pick(r, c.hitables);
        if (r.chance(40)) { h.kind = r.chance(50) ? 7u : bad; break; }
        h.kind = PT_HIT_MEDIUM_GROUP;
        const uint32_t left = (uint32_t)(c.hitables.data() + c.hitables.size() - &h) - 1u;
        const uint32_t n = r.chance(60) ? (left ? 1u + r.below(left) : 0u) : (r.chance(50) ? left + 1u + r.below(3) : bad);
        memcpy(&h.p[0], &n, 4);
        h.density = r.uni(0.05f, 0.5f);
        h.transform = -1;
        for (uint32_t m = 0; m < c.materials.size(); ++m)
            if (c.materials[m].kind == PT_MAT_ISOTROPIC && r.chance(80)) h.medium_material = (int32_t)m;
        break;
    }
    case 23: if (!c.hitables.empty()) pick(r, c.hitables).material = r.chance(50) ? (uint32_t)c.materials.size() : bad; break;
    case 24: if (!c.hitables.empty()) pick(r, c.hitables).transform = r.chance(30) ? (int32_t)c.transforms.size() : (int32_t)(r.below(4) | (r.below(16) << 20) | (r.below(16) << 24) | (r.chance(20) ? 0x80000000u : 0u)); break;
    case 25: if (!c.hitables.empty()) { pt_hitable &h = pick(r, c.hitables); h.medium_material = r.chance(30) ? (int32_t)c.materials.size() : (r.chance(50) ? (int32_t)r.below((uint32_t)c.materials.size()) : (int32_t)bad); h.density = bf; } break;
    case 26: if (!c.hitables.empty()) pick(r, c.hitables).p[r.below(10)] = bf; break;
    case 27: if (!c.nodes.empty()) { pt_node &n = pick(r, c.nodes); n.kind = r.chance(50) ? 5u : bad; } break;
    case 28: if (!c.nodes.empty()) { pt_node &n = pick(r, c.nodes); (r.chance(50) ? n.a : n.b) = r.chance(30) ? (uint32_t)c.nodes.size() : bad; } break;
    case 29: if (!c.nodes.empty()) { pt_node &n = pick(r, c.nodes); n.kind = r.chance(50) ? PT_NODE_INSTANCE : PT_NODE_MEDIUM; n.b = (uint32_t)(&n - c.nodes.data()); n.a = 0; } break;   // a wrapper around itself
    case 30: if (!c.children.empty()) pick(r, c.children) = r.chance(30) ? c.root_node : (r.chance(50) ? (uint32_t)c.nodes.size() : bad); break;   // a list that holds the root: a cycle
    case 31: if (!c.nodes.empty()) { pt_node &n = pick(r, c.nodes); n.kind = PT_NODE_LIST; n.a = 0xfffffff0u; n.b = 0x20u; } break;   // a + b wraps
    case 32: {   // a chain of lists that each hold the one below TWICE: 2^depth list entries from a few dozen nodes
        if (c.nodes.empty()) break;
        uint32_t below = r.below((uint32_t)c.nodes.size());
        const uint32_t depth = r.chance(4) ? 20 + r.below(18) : 4 + r.below(12);   // (mostly a few thousand entries; now and then beyond every limit)
        for (uint32_t d = 0; d < depth; ++d) {
            pt_node nd{PT_NODE_LIST, (uint32_t)c.children.size(), 2, 0.0f};
            c.children.push_back(below), c.children.push_back(below);
            c.nodes.push_back(nd);
            below = (uint32_t)c.nodes.size() - 1u;
        }
        c.root_node = below;
        break;
    }
    case 33: {   // a chain of Instances / media deeper than any limit
        if (c.nodes.empty() || c.transforms.empty()) break;
        uint32_t below = r.below((uint32_t)c.nodes.size());
        const uint32_t depth = 10 + r.below(60);
        for (uint32_t d = 0; d < depth; ++d) {
            c.nodes.push_back(pt_node{PT_NODE_INSTANCE, r.below((uint32_t)c.transforms.size()), below, 0.0f});
            below = (uint32_t)c.nodes.size() - 1u;
        }
        c.root_node = below;
        break;
    }
    }
}

volatile uint64_t g_case = 0;
void on_alarm(int) {
    fprintf(stderr, "fuzz_desc: case %llu did not return within the time limit\n", (unsigned long long)g_case);
    _exit(3);
}

}  // namespace

int main(int argc, char **argv) {
    const uint64_t cases = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000ull;
    const uint64_t seed0 = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1ull;
    signal(SIGALRM, on_alarm);
    std::map<int, uint64_t> by_code;
    std::map<std::string, uint64_t> by_message, by_kernel, symbols;
    uint64_t mutated = 0;
    for (uint64_t i = 0; i < cases; ++i) {
        g_case = seed0 + i;
        Rng r{(seed0 + i) * 0x9e3779b97f4a7c15ull | 1ull};
        r.next(), r.next();
        Case c;
        const uint32_t kind = r.below(10);
        const bool world = kind >= 4, graph = kind >= 7;
        if (!world) sphere_case(r, c);
        else world_case(r, c, graph);
        for (int q = 0; q < 3; ++q) c.cam.origin[q] = r.uni(-5, 5), c.cam.horizontal[q] = r.uni(-1, 1), c.cam.vertical[q] = r.uni(-1, 1), c.cam.lower_left_corner[q] = r.uni(-1, 1);
        c.cam.time1 = 1.0f;
        // the coordinates kernel selection looks at, spread over their thresholds (valid values: these are not mutations)
        if (r.chance(60)) {
            static const uint32_t depths[] = {1, 3, 10, 26, 27, 40, 41, 64, 65, 200}, spps[] = {1, 4, 11, 12, 16, 64, 256};
            static const uint32_t sizes[][2] = {{8, 8}, {64, 48}, {200, 100}, {1200, 100}, {1200, 800}, {2400, 1600}};
            static const uint32_t bits[] = {1, 2, 4, 8, 32, 64, 128, 256, 1024, 2048, 8192, 65536, 131072, 262144, 524288, 1048576};
            c.params.max_depth = depths[r.below(10)], c.params.samples = spps[r.below(7)];
            const uint32_t sz = r.below(6);
            c.params.width = sizes[sz][0], c.params.height = sizes[sz][1];
            if (r.chance(40)) c.variant = bits[r.below(16)] | (r.chance(30) ? bits[r.below(16)] : 0u);
            if (r.chance(15)) c.shard_count = 2u + r.below(7);
            if (r.chance(10)) c.blocks = 1u + r.below(5);
        }
        c.has_sky = r.below(2);
        const uint32_t n_mut = r.chance(25) ? 0u : 1u + r.below(3);
        for (uint32_t m = 0; m < n_mut; ++m) mutate(r, c, world);
        mutated += n_mut != 0;
        std::vector<void *> owned;
        for (size_t k = 0; k < c.images.size(); ++k) c.images[k].rgb = exact(c.image_bytes[k], owned);
        const auto null_if = [&](int bit, const void *p, size_t count) -> const void * { return (c.null_mask >> bit & 1u) && count ? nullptr : p; };
        pt_scene_desc sd{};
        pt_world_desc wd{};
        pt_perlin *perlin = nullptr;
        if (c.has_perlin && !c.null_perlin) perlin = (pt_perlin *)malloc(sizeof(pt_perlin)), memcpy(perlin, &c.perlin, sizeof c.perlin), owned.push_back(perlin);
        if (!world) {
            sd.n_spheres = (uint32_t)c.spheres.size();
            sd.spheres = (const pt_sphere *)null_if(0, exact(c.spheres, owned), c.spheres.size());
            sd.sphere_material = (const uint32_t *)null_if(1, exact(c.sphere_mat, owned), c.sphere_mat.size());
            sd.n_materials = (uint32_t)c.materials.size(), sd.materials = (const pt_material *)null_if(2, exact(c.materials, owned), c.materials.size());
            sd.n_textures = (uint32_t)c.textures.size(), sd.textures = (const pt_texture *)null_if(3, exact(c.textures, owned), c.textures.size());
            sd.perlin = perlin;
            sd.n_bvh_nodes = (uint32_t)c.bvh.size(), sd.bvh_nodes = (const pt_bvh_node *)null_if(4, exact(c.bvh, owned), c.bvh.size());
            sd.bvh_root = c.bvh_root, sd.has_sky = c.has_sky;
            memcpy(sd.sky, c.sky, sizeof sd.sky);
        } else {
            wd.n_hitables = (uint32_t)c.hitables.size(), wd.hitables = (const pt_hitable *)null_if(0, exact(c.hitables, owned), c.hitables.size());
            wd.n_transforms = (uint32_t)c.transforms.size(), wd.transforms = (const pt_affine *)null_if(1, exact(c.transforms, owned), c.transforms.size());
            wd.n_materials = (uint32_t)c.materials.size(), wd.materials = (const pt_material *)null_if(2, exact(c.materials, owned), c.materials.size());
            wd.n_textures = (uint32_t)c.textures.size(), wd.textures = (const pt_texture *)null_if(3, exact(c.textures, owned), c.textures.size());
            wd.perlin = perlin;
            wd.n_bvh_nodes = (uint32_t)c.bvh.size(), wd.bvh_nodes = (const pt_bvh_node *)null_if(4, exact(c.bvh, owned), c.bvh.size());
            wd.bvh_root = c.bvh_root, wd.has_sky = c.has_sky;
            memcpy(wd.sky, c.sky, sizeof wd.sky);
            wd.n_images = (uint32_t)c.images.size(), wd.images = (const pt_image *)null_if(5, exact(c.images, owned), c.images.size());
            wd.n_nodes = (uint32_t)c.nodes.size(), wd.nodes = (const pt_node *)null_if(6, exact(c.nodes, owned), c.nodes.size());
            wd.n_node_children = (uint32_t)c.children.size(), wd.node_children = (const uint32_t *)null_if(7, exact(c.children, owned), c.children.size());
            wd.root_node = c.root_node;
        }
        if (getenv("FUZZ_VERBOSE"))
            fprintf(stderr, "case %llu: %s spheres %zu hitables %zu transforms %zu materials %zu textures %zu bvh %zu (root %d) nodes %zu children %zu root_node %u params %ux%u s%u d%u bvh%u shards %u variant %u blocks %u null %x\n",
                    (unsigned long long)g_case, world ? (graph ? "graph" : "world") : "spheres", c.spheres.size(), c.hitables.size(), c.transforms.size(), c.materials.size(), c.textures.size(),
                    c.bvh.size(), c.bvh_root, c.nodes.size(), c.children.size(), c.root_node, c.params.width, c.params.height, c.params.samples, c.params.max_depth, c.params.use_bvh, c.shard_count,
                    c.variant, c.blocks, c.null_mask);
        pt_kernel_choice out{};
        const pt_params *pp = (c.null_mask >> 8 & 1u) ? nullptr : &c.params;
        const pt_camera *pc = (c.null_mask >> 9 & 1u) ? nullptr : &c.cam;
        pt_kernel_choice *po = (c.null_mask >> 10 & 1u) ? nullptr : &out;
        alarm(getenv("FUZZ_ALARM") ? (unsigned)atoi(getenv("FUZZ_ALARM")) : 20u);
        const bool both_or_neither = (c.null_mask >> 11 & 1u) != 0u;   // (the function wants exactly one of the two descriptions)
        const int rc = both_or_neither ? ((c.null_mask & 1u) ? pt_debug_select(&sd, &wd, pp, pc, c.shard_count, c.blocks, c.variant, po)
                                                             : pt_debug_select(nullptr, nullptr, pp, pc, c.shard_count, c.blocks, c.variant, po))
                                       : pt_debug_select(world ? nullptr : &sd, world ? &wd : nullptr, pp, pc, c.shard_count, c.blocks, c.variant, po);
        alarm(0);
        by_code[rc] += 1;
        const char *msg = pt_last_error();
        if (rc == PT_OK) {
            if (po) by_kernel[std::string(out.name)] += 1;
            char fs[256], ms[256];
            if (pt_debug_last_kernel_symbols(fs, ms, sizeof fs) == PT_OK) {
                if (fs[0]) symbols[fs] += 1;
                if (ms[0]) symbols[ms] += 1;
            }
        } else if (rc == PT_ERR_INVALID_ARG || rc == PT_ERR_UNSUPPORTED) {
            if (!msg || !msg[0]) {
                fprintf(stderr, "fuzz_desc: case %llu refused with code %d and NO message\n", (unsigned long long)g_case, rc);
                return 2;
            }
            std::string key(msg);
            for (char &ch : key)
                if (ch >= '0' && ch <= '9') ch = '#';
            by_message[(rc == PT_ERR_INVALID_ARG ? "INVALID_ARG  " : "UNSUPPORTED  ") + key.substr(0, 70)] += 1;
        } else {
            fprintf(stderr, "fuzz_desc: case %llu returned the unexpected code %d (%s)\n", (unsigned long long)g_case, rc, msg ? msg : "");
            return 2;
        }
        for (void *p : owned) free(p);
    }
    printf("fuzz_desc: %llu descriptions (seeds %llu ..), %llu of them mutated: every one accepted or refused by name\n", (unsigned long long)cases, (unsigned long long)seed0,
           (unsigned long long)mutated);
    for (const auto &kv : by_code) printf("  code %d (%s): %llu\n", kv.first, kv.first == 0 ? "PT_OK" : (kv.first == 1 ? "PT_ERR_INVALID_ARG" : "PT_ERR_UNSUPPORTED"), (unsigned long long)kv.second);
    printf("  kernels chosen for the accepted ones:\n");
    for (const auto &kv : by_kernel) printf("    %-40s %llu\n", kv.first.c_str(), (unsigned long long)kv.second);
    printf("  kernel instantiations the accepted ones launch (launch-stub symbols): %zu\n", symbols.size());
    for (const auto &kv : symbols) printf("    symbol %s %llu\n", kv.first.c_str(), (unsigned long long)kv.second);
    printf("  refusals by message (digits masked):\n");
    for (const auto &kv : by_message) printf("    %8llu  %s\n", (unsigned long long)kv.second, kv.first.c_str());
    return 0;
}
