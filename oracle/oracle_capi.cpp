// oracle/oracle_capi.cpp — TEST INFRASTRUCTURE ONLY (the checker, never the product).
//
// C entry points over the CPU restatement (restate_*.hpp) so that tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg can drive it through ctypes.  Nothing under pathtracer_amd/ links or
// loads this library.  Ray/hit/node layouts are those of include/ptx.h.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <map>
#include <sstream>

#include "restate_wavefront.hpp"

using namespace orc;

namespace {

struct RayIn {
    float ox, oy, oz, tmin, dx, dy, dz, tmax;
};
struct HitOut {
    uint32_t index;
    float t, u, v;
};
struct FlatNode {
    uint32_t w0, w1;
};

struct Tree {
    std::vector<AnyTriangle> triangles;
    std::unique_ptr<KdNode> root;
};

struct SceneH {
    Scene scene;
    Obj obj;
    Mtl mtl;
    MtlCamera camera;
};

Triangle tri_from(const float* v) {
    return Triangle{{v[0], v[1], v[2]}, {v[3], v[4], v[5]}, {v[6], v[7], v[8]}};
}
Ray ray_from6(const float* r) { return Ray{{r[0], r[1], r[2]}, {r[3], r[4], r[5]}}; }

std::unique_ptr<KdNode> from_flat(const FlatNode* nodes, uint32_t n_nodes, const uint32_t* leaf_idx,
                                  uint32_t i) {
    if (i >= n_nodes) throw std::runtime_error("flat tree: node index out of range");
    const FlatNode& nd = nodes[i];
    uint32_t kind = nd.w1 & 3u;
    if (kind == 3u) {
        uint32_t count = nd.w1 >> 2;
        return KdNode::new_leaf(std::vector<uint32_t>(leaf_idx + nd.w0, leaf_idx + nd.w0 + count));
    }
    float dist;
    std::memcpy(&dist, &nd.w0, 4);
    auto l = from_flat(nodes, n_nodes, leaf_idx, i + 1);
    auto r = from_flat(nodes, n_nodes, leaf_idx, nd.w1 >> 2);
    return KdNode::new_node(Aap{(int)kind, dist}, std::move(l), std::move(r));
}

void to_flat(const KdNode* n, std::vector<FlatNode>& nodes, std::vector<uint32_t>& leaf_idx) {
    size_t me = nodes.size();
    nodes.push_back({0, 0});
    if (n->is_leaf) {
        nodes[me].w0 = (uint32_t)leaf_idx.size();
        nodes[me].w1 = ((uint32_t)n->indices.size() << 2) | 3u;
        leaf_idx.insert(leaf_idx.end(), n->indices.begin(), n->indices.end());
    } else {
        std::memcpy(&nodes[me].w0, &n->plane.distance, 4);
        to_flat(n->left.get(), nodes, leaf_idx);
        uint32_t right = (uint32_t)nodes.size();
        nodes[me].w1 = (right << 2) | (uint32_t)n->plane.axis;
        to_flat(n->right.get(), nodes, leaf_idx);
    }
}

void write_hit(HitOut& o, const std::optional<IndexedIntersection>& h) {
    if (h) {
        o = {h->index, h->inner.t, h->inner.u, h->inner.v};
    } else {
        o = {0xFFFFFFFFu, 0.0f, 0.0f, 0.0f};
    }
}

template <class F>
void parallel_for(size_t n, int threads, F f) {
    if (threads <= 1) {
        f(0, 0, n);
        return;
    }
    std::vector<std::thread> pool;
    size_t chunk = (n + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        size_t a = (size_t)t * chunk, b = std::min(n, a + chunk);
        if (a >= b) break;
        pool.emplace_back([=] { f(t, a, b); });
    }
    for (auto& th : pool) th.join();
}

int copy_out(const std::string& s, char* buf, size_t cap) {
    if (s.size() + 1 > cap) return -(int)(s.size() + 1);
    std::memcpy(buf, s.c_str(), s.size() + 1);
    return (int)s.size();
}

thread_local std::string g_err;

}  // namespace

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

// ---- geometry known-answer entry points -------------------------------------------------------
// as_any != 0 applies AnyTriangle::from first (any_triangle.rs:9-16).  Returns 1 on hit.
int orc_triangle_intersect(const float* tri9, int as_any, const float* ray6, float* tuv) {
    Triangle t = tri_from(tri9);
    Ray r = ray_from6(ray6);
    auto h = as_any ? AnyTriangle::from(t).intersect_ray(r) : t.intersect_ray(r);
    if (!h) return 0;
    tuv[0] = h->t;
    tuv[1] = h->u;
    tuv[2] = h->v;
    return 1;
}
int orc_axial_axis(const float* tri9) {
    AnyTriangle a = AnyTriangle::from(tri_from(tri9));
    return a.axial ? a.ax.plane.axis : -1;
}
int orc_axial_intersect_point(int axis, float dist, const float* v6, const float* p2, float* uv) {
    AxiallyAlignedTriangle t{Aap{axis, dist}, {v6[0], v6[1]}, {v6[2], v6[3]}, {v6[4], v6[5]}};
    return t.intersect_point({p2[0], p2[1]}, uv[0], uv[1]) ? 1 : 0;
}
int orc_aap_intersect(int axis, float dist, const float* ray6, float* t) {
    auto r = Aap{axis, dist}.intersect_ray(ray_from6(ray6));
    if (!r) return 0;
    *t = *r;
    return 1;
}
int orc_clip_points(const float* tri9, const float* aabb6, float* out, int cap_points) {
    Triangle t = tri_from(tri9);
    Aabb b{{aabb6[0], aabb6[1], aabb6[2]}, {aabb6[3], aabb6[4], aabb6[5]}};
    auto pts = clip_triangle_aabb_points(t.v0, t.v1, t.v2, b);
    for (size_t i = 0; i < pts.size() && (int)i < cap_points; ++i) {
        out[3 * i] = pts[i].x;
        out[3 * i + 1] = pts[i].y;
        out[3 * i + 2] = pts[i].z;
    }
    return (int)pts.size();
}
int orc_clip_aabb(const float* tri9, const float* aabb6, float* out6) {
    Triangle t = tri_from(tri9);
    Aabb b{{aabb6[0], aabb6[1], aabb6[2]}, {aabb6[3], aabb6[4], aabb6[5]}};
    auto r = clip_triangle_aabb(t.v0, t.v1, t.v2, b);
    if (!r) return 0;
    out6[0] = r->min.x; out6[1] = r->min.y; out6[2] = r->min.z;
    out6[3] = r->max.x; out6[4] = r->max.y; out6[5] = r->max.z;
    return 1;
}
void orc_bounding_box(const float* tris, uint32_t n, float* out6) {
    std::vector<AnyTriangle> g;
    for (uint32_t i = 0; i < n; ++i) g.push_back(AnyTriangle::from(tri_from(tris + 9 * i)));
    Aabb b = geometries_bounding_box(g);
    out6[0] = b.min.x; out6[1] = b.min.y; out6[2] = b.min.z;
    out6[3] = b.max.x; out6[4] = b.max.y; out6[5] = b.max.z;
}
// Event lists of KdCell::generate_initial for one axis: out = (index, distance bits, kind) triples.
int orc_event_list(const float* tris, uint32_t n, int axis, uint32_t* out, int cap) {
    std::vector<AnyTriangle> g;
    for (uint32_t i = 0; i < n; ++i) g.push_back(AnyTriangle::from(tri_from(tris + 9 * i)));
    KdCell c = generate_initial_cell(g);
    int k = 0;
    for (const Event& e : c.events[axis]) {
        if (k < cap) {
            out[3 * k] = e.index;
            std::memcpy(&out[3 * k + 1], &e.distance, 4);
            out[3 * k + 2] = e.kind;
        }
        ++k;
    }
    return k;
}
// sah.rs tests: update_geometry_side for triangle set, chosen plane/side → per-triangle side (0 both,1 L,2 R)
void orc_update_geometry_side(const float* tris, uint32_t n, int axis, float dist, int side_right,
                              uint8_t* out) {
    std::vector<AnyTriangle> g;
    for (uint32_t i = 0; i < n; ++i) g.push_back(AnyTriangle::from(tri_from(tris + 9 * i)));
    KdCell c = generate_initial_cell(g);
    std::vector<EventSide> sides(n, EventSide::Both);
    SahSplit best{Aap{axis, dist}, side_right ? Side::Right : Side::Left, 1.0f};
    update_geometry_side(c.indices, c.events, best, sides);
    for (uint32_t i = 0; i < n; ++i) out[i] = (uint8_t)sides[i];
}

// ---- trees ------------------------------------------------------------------------------------
void* orc_tree_build(const float* tris, uint32_t n, const float* sah3) {
    auto* t = new Tree();
    for (uint32_t i = 0; i < n; ++i) t->triangles.push_back(AnyTriangle::from(tri_from(tris + 9 * i)));
    SahCost s;
    if (sah3) s = {sah3[0], sah3[1], sah3[2]};
    t->root = build_kdtree(t->triangles, s);
    return t;
}
// raw_triangles != 0: keep plain `Triangle`s (no AnyTriangle conversion), as the reference's
// kdtree/src/lib.rs tests do with `[Triangle; N]` geometries.
void* orc_tree_from_flat(const float* tris, uint32_t n, int raw_triangles, const void* nodes,
                         uint32_t n_nodes, const uint32_t* leaf_idx) {
    try {
        auto t = std::make_unique<Tree>();
        for (uint32_t i = 0; i < n; ++i) {
            AnyTriangle a = AnyTriangle::from(tri_from(tris + 9 * i));
            if (raw_triangles) a.axial = false;
            t->triangles.push_back(a);
        }
        t->root = from_flat((const FlatNode*)nodes, n_nodes, leaf_idx, 0);
        return t.release();
    } catch (std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}
void orc_tree_free(void* h) { delete (Tree*)h; }
int orc_tree_string(void* h, char* buf, size_t cap) {
    std::string s;
    tree_to_string(((Tree*)h)->root.get(), s);
    return copy_out(s, buf, cap);
}
// out: nodes, leaves, empty_leaves, leaf_refs, max_depth, max_leaf
void orc_tree_stats(void* h, uint64_t* out6) {
    TreeStats s;
    tree_stats(((Tree*)h)->root.get(), 1, s);
    out6[0] = s.nodes; out6[1] = s.leaves; out6[2] = s.empty_leaves;
    out6[3] = s.leaf_refs; out6[4] = s.max_depth; out6[5] = s.max_leaf;
}
// Flatten to the 8-byte preorder layout of include/ptx.h.  Pass NULL buffers to query sizes.
void orc_tree_to_flat(void* h, void* nodes_out, uint32_t* n_nodes, uint32_t* leaf_out,
                      uint32_t* n_leaf) {
    std::vector<FlatNode> nodes;
    std::vector<uint32_t> leaf;
    to_flat(((Tree*)h)->root.get(), nodes, leaf);
    if (nodes_out) std::memcpy(nodes_out, nodes.data(), nodes.size() * sizeof(FlatNode));
    if (leaf_out) std::memcpy(leaf_out, leaf.data(), leaf.size() * 4);
    *n_nodes = (uint32_t)nodes.size();
    *n_leaf = (uint32_t)leaf.size();
}
// KdNode::intersect over a batch (pointer tree).  counters3 (optional): inner nodes, leaves, tri tests.
void orc_tree_intersect(void* h, const void* rays, size_t n, void* hits, uint64_t* counters3,
                        int threads) {
    Tree* t = (Tree*)h;
    const RayIn* r = (const RayIn*)rays;
    HitOut* o = (HitOut*)hits;
    std::vector<TraversalCounters> ctr(std::max(threads, 1));
    parallel_for(n, threads, [&](int tid, size_t a, size_t b) {
        for (size_t i = a; i < b; ++i) {
            Ray ray{{r[i].ox, r[i].oy, r[i].oz}, {r[i].dx, r[i].dy, r[i].dz}};
            write_hit(o[i], kd_intersect(t->root.get(), t->triangles, ray, r[i].tmin, r[i].tmax,
                                         counters3 ? &ctr[tid] : nullptr));
        }
    });
    if (counters3) {
        counters3[0] = counters3[1] = counters3[2] = 0;
        for (auto& c : ctr) {
            counters3[0] += c.inner_nodes;
            counters3[1] += c.leaves;
            counters3[2] += c.tri_tests;
        }
    }
}
// Brute force over all triangles in index order (ray_bouncer.rs:32-39).
void orc_brute_intersect(void* h, const void* rays, size_t n, void* hits, int threads) {
    Tree* t = (Tree*)h;
    const RayIn* r = (const RayIn*)rays;
    HitOut* o = (HitOut*)hits;
    std::vector<uint32_t> all(t->triangles.size());
    for (uint32_t i = 0; i < all.size(); ++i) all[i] = i;
    parallel_for(n, threads, [&](int, size_t a, size_t b) {
        for (size_t i = a; i < b; ++i) {
            Ray ray{{r[i].ox, r[i].oy, r[i].oz}, {r[i].dx, r[i].dy, r[i].dz}};
            write_hit(o[i], intersect_closest_geometry(t->triangles, all.begin(), all.end(), ray,
                                                       r[i].tmin, r[i].tmax));
        }
    });
}

// Visits per tree depth (root = depth 0) over a ray batch: sizing aid for caching the top of the tree.
void orc_depth_histogram(void* h, const void* rays, size_t n, uint64_t* hist32) {
    Tree* t = (Tree*)h;
    const RayIn* r = (const RayIn*)rays;
    std::vector<std::pair<const KdNode*, int>> depth_of;
    std::map<const KdNode*, int> depth;
    std::vector<std::pair<const KdNode*, int>> st{{t->root.get(), 0}};
    while (!st.empty()) {
        auto [nd, d] = st.back();
        st.pop_back();
        depth[nd] = d;
        if (!nd->is_leaf) {
            st.push_back({nd->left.get(), d + 1});
            st.push_back({nd->right.get(), d + 1});
        }
    }
    for (int i = 0; i < 32; ++i) hist32[i] = 0;
    for (size_t i = 0; i < n; ++i) {
        Ray ray{{r[i].ox, r[i].oy, r[i].oz}, {r[i].dx, r[i].dy, r[i].dz}};
        // same walk as kd_intersect, recording node depths
        struct E { const KdNode* node; float t1, t2; };
        const KdNode* node = t->root.get();
        float t1 = r[i].tmin, t2 = r[i].tmax, t_end = r[i].tmax;
        E stack[MAX_DEPTH];
        size_t sp = 0;
        for (;;) {
            ++hist32[std::min(depth[node], 31)];
            if (node->is_leaf) {
                auto res = intersect_closest_geometry(t->triangles, node->indices.begin(), node->indices.end(), ray, t1, t2);
                if (res || t2 == t_end || sp == 0) break;
                --sp;
                node = stack[sp].node; t1 = stack[sp].t1; t2 = stack[sp].t2;
            } else {
                auto tt = node->plane.intersect_ray(ray);
                int axis = node->plane.axis;
                if (tt) {
                    const KdNode* nn = ray.direction[axis] >= 0.0f ? node->left.get() : node->right.get();
                    const KdNode* fn = ray.direction[axis] >= 0.0f ? node->right.get() : node->left.get();
                    if (*tt > t2) node = nn;
                    else if (*tt < t1) node = fn;
                    else { stack[sp++] = {fn, *tt, t2}; node = nn; t2 = *tt; }
                } else node = ray.origin[axis] <= node->plane.distance ? node->left.get() : node->right.get();
            }
        }
    }
}

// ---- materials / sampling ---------------------------------------------------------------------
// out10: is_delta, pdf, bsdf[3], wo[3]
static void write_sample(const BsdfSample& s, float* out) {
    out[0] = s.is_delta ? 1.0f : 0.0f;
    out[1] = s.pdf;
    out[2] = s.bsdf.x; out[3] = s.bsdf.y; out[4] = s.bsdf.z;
    out[5] = s.wo.x; out[6] = s.wo.y; out[7] = s.wo.z;
}
void orc_sample_refraction(const float* wi, const float* n, float ior, const float* diffuse,
                           const float* td, float probability, float* out8) {
    Surface s{{wi[0], wi[1], wi[2]}, {n[0], n[1], n[2]}, {0, 0}};
    write_sample(sample_refraction(s, ior, {diffuse[0], diffuse[1], diffuse[2]},
                                   {td[0], td[1], td[2]}, probability),
                 out8);
}
// material: color[3], f0[3], transmission, ior.  mode 0 = SEQ(seed), 1 = COUNTER(key = seed), dim base.
void orc_material_sample(const float* mat8, const float* wi, const float* n, const float* uv, int mode,
                         uint64_t seed, uint32_t dim, float* out8) {
    Material m;
    m.color = {mat8[0], mat8[1], mat8[2]};
    m.schlick_f0 = {mat8[3], mat8[4], mat8[5]};
    m.transmission = mat8[6];
    m.ior = mat8[7];
    Xoshiro256pp seq(seed);
    Sampler rng;
    rng.mode = (Sampler::Mode)mode;
    rng.seq = &seq;
    rng.key = seed;
    std::vector<Texture> none;
    Surface s{{wi[0], wi[1], wi[2]}, {n[0], n[1], n[2]}, {uv[0], uv[1]}};
    write_sample(material_sample(m, none, s, rng, dim), out8);
}
void orc_material_from_mtl(const float* kd, const float* ks, float ni, float pm, float tr, float* out8) {
    Material m = material_from_mtl({kd[0], kd[1], kd[2]}, {ks[0], ks[1], ks[2]}, ni, pm, tr, -1);
    out8[0] = m.color.x; out8[1] = m.color.y; out8[2] = m.color.z;
    out8[3] = m.schlick_f0.x; out8[4] = m.schlick_f0.y; out8[5] = m.schlick_f0.z;
    out8[6] = m.transmission; out8[7] = m.ior;
}
// which: 0 unit square (2), 1 unit sphere (3), 2 concentric disk (2), 3 cosine hemisphere (3).
// COUNTER mode: draw i uses key = ptx_path_key(seed, i, 0), dimension 0.
void orc_sample_many(int which, int mode, uint64_t seed, uint32_t count, float* out) {
    Xoshiro256pp seq(seed);
    Sampler rng;
    rng.mode = (Sampler::Mode)mode;
    rng.seq = &seq;
    for (uint32_t i = 0; i < count; ++i) {
        rng.key = ptx_path_key(seed, i, 0);
        switch (which) {
            case 0: out[2 * i] = rng.u01(0); out[2 * i + 1] = rng.u01(1); break;
            case 1: { Vec3 v = uniform_sample_unit_sphere(rng, 0); out[3*i]=v.x; out[3*i+1]=v.y; out[3*i+2]=v.z; break; }
            case 2: { Vec2 v = concentric_sample_unit_disk(rng, 0); out[2*i]=v.x; out[2*i+1]=v.y; break; }
            default: { Vec3 v = cosine_sample_hemisphere(rng, 0); out[3*i]=v.x; out[3*i+1]=v.y; out[3*i+2]=v.z; break; }
        }
    }
}
float orc_wrap01(float x) { return wrap01(x); }
void orc_texture_get(const float* rgb, uint32_t w, uint32_t h, float u, float v, float* out3) {
    Material m;
    m.texture = 0;
    std::vector<Texture> tex(1);
    tex[0].width = w;
    tex[0].height = h;
    tex[0].rgb.assign(rgb, rgb + (size_t)w * h * 3);
    Vec3 c = albedo_get(m, tex, {u, v});
    out3[0] = c.x; out3[1] = c.y; out3[2] = c.z;
}
void orc_sincos(const float* x, uint32_t n, float* s, float* c) {
    for (uint32_t i = 0; i < n; ++i) ptx_sincos(x[i], &s[i], &c[i]);
}
void orc_to_rgb8(const float* accum, size_t n_pixels, uint16_t iterations, uint8_t* out) {
    to_rgb8(accum, n_pixels, iterations, out);
}

// ---- parsers (JSON dump for the parser known-answer tests) --------------------------------------
static void jf(std::ostringstream& o, float f) {
    char b[40];
    snprintf(b, sizeof b, "%.9g", f);
    o << b;
}
static void jv3(std::ostringstream& o, Vec3 v) {
    o << '['; jf(o, v.x); o << ','; jf(o, v.y); o << ','; jf(o, v.z); o << ']';
}
static std::string jstr(const std::string& s) {
    std::string r = "\"";
    for (char c : s) {
        if (c == '"' || c == '\\') r += '\\';
        r += c;
    }
    return r + "\"";
}
int orc_parse_obj_json(const char* text, char* buf, size_t cap) {
    try {
        std::istringstream in(text);
        Obj obj = parse_obj(in);
        std::ostringstream o;
        o << "{\"mtl_lib\":" << jstr(obj.mtl_lib) << ",\"vertices\":[";
        for (size_t i = 0; i < obj.vertices.size(); ++i) { if (i) o << ','; jv3(o, obj.vertices[i]); }
        o << "],\"normals\":[";
        for (size_t i = 0; i < obj.normals.size(); ++i) { if (i) o << ','; jv3(o, obj.normals[i]); }
        o << "],\"texcoords\":[";
        for (size_t i = 0; i < obj.texcoords.size(); ++i) {
            if (i) o << ',';
            o << '['; jf(o, obj.texcoords[i].x); o << ','; jf(o, obj.texcoords[i].y); o << ']';
        }
        o << "],\"chunks\":[";
        for (size_t c = 0; c < obj.chunks.size(); ++c) {
            if (c) o << ',';
            o << "{\"material\":" << jstr(obj.chunks[c].material) << ",\"faces\":[";
            for (size_t f = 0; f < obj.chunks[c].faces.size(); ++f) {
                if (f) o << ',';
                o << '[';
                for (size_t p = 0; p < obj.chunks[c].faces[f].size(); ++p) {
                    if (p) o << ',';
                    const ObjPoint& pt = obj.chunks[c].faces[f][p];
                    o << '[' << pt.v << ',' << pt.t << ',' << pt.n << ']';
                }
                o << ']';
            }
            o << "]}";
        }
        o << "]}";
        return copy_out(o.str(), buf, cap);
    } catch (std::exception& e) {
        g_err = e.what();
        return -1;
    }
}
int orc_parse_mtl_json(const char* text, char* buf, size_t cap) {
    try {
        std::istringstream in(text);
        Mtl mtl = parse_mtl(in);
        std::ostringstream o;
        o << "{\"materials\":[";
        for (size_t i = 0; i < mtl.materials.size(); ++i) {
            const MtlMaterial& m = mtl.materials[i];
            if (i) o << ',';
            o << "{\"name\":" << jstr(m.name) << ",\"diffuse_map\":" << jstr(m.diffuse_map)
              << ",\"diffuse_reflection\":";
            jv3(o, m.diffuse_reflection);
            o << ",\"specular_reflection\":";
            jv3(o, m.specular_reflection);
            o << ",\"transparency\":"; jf(o, m.transparency);
            o << ",\"reflection_0_degrees\":"; jf(o, m.reflection_0_degrees);
            o << ",\"reflection_90_degrees\":"; jf(o, m.reflection_90_degrees);
            o << ",\"index_of_refraction\":"; jf(o, m.index_of_refraction);
            o << ",\"metalness\":"; jf(o, m.metalness);
            o << '}';
        }
        o << "],\"lights\":[";
        for (size_t i = 0; i < mtl.lights.size(); ++i) {
            if (i) o << ',';
            o << "{\"position\":"; jv3(o, mtl.lights[i].position);
            o << ",\"color\":"; jv3(o, mtl.lights[i].color);
            o << ",\"radius\":"; jf(o, mtl.lights[i].radius);
            o << ",\"intensity\":"; jf(o, mtl.lights[i].intensity);
            o << '}';
        }
        o << "],\"cameras\":[";
        for (size_t i = 0; i < mtl.cameras.size(); ++i) {
            if (i) o << ',';
            o << "{\"position\":"; jv3(o, mtl.cameras[i].position);
            o << ",\"target\":"; jv3(o, mtl.cameras[i].target);
            o << ",\"up\":"; jv3(o, mtl.cameras[i].up);
            o << ",\"fov\":"; jf(o, mtl.cameras[i].fov);
            o << '}';
        }
        o << "]}";
        return copy_out(o.str(), buf, cap);
    } catch (std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

// ---- scenes -----------------------------------------------------------------------------------
// pathtracer-cli/src/main.rs:123-159 (setup_scene) without textures; attach them with
// orc_scene_set_texture before rendering.
void* orc_scene_load(const char* obj_path, const float* sah3, uint32_t max_bounces) {
    try {
        auto h = std::make_unique<SceneH>();
        std::ifstream fo(obj_path);
        if (!fo) throw std::runtime_error(std::string("cannot open ") + obj_path);
        h->obj = parse_obj(fo);
        std::string p(obj_path);
        size_t slash = p.find_last_of('/');
        std::string mtl_path = (slash == std::string::npos ? std::string() : p.substr(0, slash + 1)) + h->obj.mtl_lib;
        std::ifstream fm(mtl_path);
        if (!fm) throw std::runtime_error("cannot open " + mtl_path);
        h->mtl = parse_mtl(fm);
        from_wavefront(h->obj, h->mtl, h->scene.triangles, h->scene.properties);
        SahCost s;
        if (sah3) s = {sah3[0], sah3[1], sah3[2]};
        h->scene.kdtree = build_kdtree(h->scene.triangles, s);
        if (h->mtl.cameras.empty()) throw std::runtime_error("scene has no camera");
        h->camera = h->mtl.cameras[0];  // always the first camera (main.rs:137)
        for (const MtlMaterial& m : h->mtl.materials)
            h->scene.materials.push_back(material_from_mtl(m.diffuse_reflection, m.specular_reflection,
                                                           m.index_of_refraction, m.metalness,
                                                           m.transparency, -1));
        for (const MtlLight& l : h->mtl.lights)  // light.rs:79-89
            h->scene.lights.push_back({l.position, l.color * l.intensity, l.radius});
        h->scene.max_bounces = max_bounces;
        return h.release();
    } catch (std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}
void orc_scene_free(void* h) { delete (SceneH*)h; }
// out: triangles, materials, lights, cameras, vertices, normals, texcoords, chunks
void orc_scene_counts(void* hh, uint32_t* out8) {
    SceneH* h = (SceneH*)hh;
    out8[0] = (uint32_t)h->scene.triangles.size();
    out8[1] = (uint32_t)h->scene.materials.size();
    out8[2] = (uint32_t)h->scene.lights.size();
    out8[3] = (uint32_t)h->mtl.cameras.size();
    out8[4] = (uint32_t)h->obj.vertices.size();
    out8[5] = (uint32_t)h->obj.normals.size();
    out8[6] = (uint32_t)h->obj.texcoords.size();
    out8[7] = (uint32_t)h->obj.chunks.size();
}
int orc_scene_texture_name(void* hh, uint32_t material, char* buf, size_t cap) {
    return copy_out(((SceneH*)hh)->mtl.materials.at(material).diffuse_map, buf, cap);
}
void orc_scene_set_texture(void* hh, uint32_t material, uint32_t w, uint32_t h, const float* rgb) {
    SceneH* s = (SceneH*)hh;
    Texture t;
    t.width = w;
    t.height = h;
    t.rgb.assign(rgb, rgb + (size_t)w * h * 3);
    s->scene.textures.push_back(std::move(t));
    s->scene.materials.at(material).texture = (int)s->scene.textures.size() - 1;
}
// Geometry/shading arrays as the product's scene description takes them (include/ptx.h).
void orc_scene_export(void* hh, float* tris9, float* normals9, float* uvs6, uint32_t* material_ids) {
    SceneH* h = (SceneH*)hh;
    for (size_t i = 0; i < h->scene.triangles.size(); ++i) {
        const Triangle& t = h->scene.triangles[i].tri;
        const TriangleProperties& p = h->scene.properties[i];
        if (tris9) {
            float* o = tris9 + 9 * i;
            o[0] = t.v0.x; o[1] = t.v0.y; o[2] = t.v0.z;
            o[3] = t.v1.x; o[4] = t.v1.y; o[5] = t.v1.z;
            o[6] = t.v2.x; o[7] = t.v2.y; o[8] = t.v2.z;
        }
        if (normals9) {
            float* o = normals9 + 9 * i;
            o[0] = p.n0.x; o[1] = p.n0.y; o[2] = p.n0.z;
            o[3] = p.n1.x; o[4] = p.n1.y; o[5] = p.n1.z;
            o[6] = p.n2.x; o[7] = p.n2.y; o[8] = p.n2.z;
        }
        if (uvs6) {
            float* o = uvs6 + 6 * i;
            o[0] = p.uv0.x; o[1] = p.uv0.y; o[2] = p.uv1.x; o[3] = p.uv1.y; o[4] = p.uv2.x; o[5] = p.uv2.y;
        }
        if (material_ids) material_ids[i] = p.material;
    }
}
// materials: 8 floats each (color, f0, transmission, ior); lights: 7 floats (center, intensity, radius)
void orc_scene_export_shading(void* hh, float* materials8, float* lights7, float* camera10) {
    SceneH* h = (SceneH*)hh;
    for (size_t i = 0; i < h->scene.materials.size(); ++i) {
        const Material& m = h->scene.materials[i];
        float* o = materials8 + 8 * i;
        o[0] = m.color.x; o[1] = m.color.y; o[2] = m.color.z;
        o[3] = m.schlick_f0.x; o[4] = m.schlick_f0.y; o[5] = m.schlick_f0.z;
        o[6] = m.transmission; o[7] = m.ior;
    }
    for (size_t i = 0; i < h->scene.lights.size(); ++i) {
        const Light& l = h->scene.lights[i];
        float* o = lights7 + 7 * i;
        o[0] = l.center.x; o[1] = l.center.y; o[2] = l.center.z;
        o[3] = l.intensity.x; o[4] = l.intensity.y; o[5] = l.intensity.z;
        o[6] = l.radius;
    }
    const MtlCamera& c = h->camera;
    float cam[10] = {c.position.x, c.position.y, c.position.z, c.target.x, c.target.y, c.target.z,
                     c.up.x, c.up.y, c.up.z, c.fov};
    std::memcpy(camera10, cam, sizeof cam);
}
void* orc_scene_tree(void* hh) {  // borrowed view for the orc_tree_* calls (do not free separately)
    SceneH* h = (SceneH*)hh;
    auto* t = new Tree();
    t->triangles = h->scene.triangles;
    std::vector<FlatNode> nodes;
    std::vector<uint32_t> leaf;
    to_flat(h->scene.kdtree.get(), nodes, leaf);
    t->root = from_flat(nodes.data(), (uint32_t)nodes.size(), leaf.data(), 0);
    return t;
}
// Pinhole::new(camera, size) → position, plane, dx, dy (12 floats)
void orc_scene_pinhole(void* hh, uint32_t w, uint32_t h, float* out12) {
    const MtlCamera& c = ((SceneH*)hh)->camera;
    Pinhole p = Pinhole::make(c.position, c.target, c.up, c.fov, w, h);
    float o[12] = {p.position.x, p.position.y, p.position.z, p.plane.x, p.plane.y, p.plane.z,
                   p.dx.x, p.dx.y, p.dx.z, p.dy.x, p.dy.y, p.dy.z};
    std::memcpy(out12, o, sizeof o);
}
// render_parallel_iterations: accumulates INTO accum (W*H*3).  stats8 (optional): extend rays, shadow
// rays, extend inner nodes, extend tri tests, shadow inner nodes, shadow tri tests, extend leaves,
// shadow leaves.  Returns wall seconds.
double orc_render(void* hh, uint32_t w, uint32_t h, uint32_t threads, uint32_t iterations_per_thread,
                  int mode, uint64_t seed, uint32_t first_sample, float* accum, uint64_t* stats8) {
    SceneH* s = (SceneH*)hh;
    Pinhole p = Pinhole::make(s->camera.position, s->camera.target, s->camera.up, s->camera.fov, w, h);
    RayStats st;
    auto t0 = std::chrono::steady_clock::now();
    render_parallel_iterations(s->scene, p, threads, iterations_per_thread, mode, seed, first_sample,
                               accum, stats8 ? &st : nullptr);
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (stats8) {
        stats8[0] = st.extend_rays; stats8[1] = st.shadow_rays;
        stats8[2] = st.extend.inner_nodes; stats8[3] = st.extend.tri_tests;
        stats8[4] = st.shadow.inner_nodes; stats8[5] = st.shadow.tri_tests;
        stats8[6] = st.extend.leaves; stats8[7] = st.shadow.leaves;
    }
    return dt;
}

// kdtree-tester-cli ray set (ray_bouncer.rs:57-120, ray_tester.rs:44-66): one jittered camera ray per
// pixel, lights sampled before the material, shadow rays on every bounce, the path continues from the
// brute-force hit, no throughput cut-off.  COUNTER sampler keyed by (seed, y*W+x, 0).  Every extend
// ray (t in [0,MAX]) and shadow ray (t in [0,1]) is appended to rays_out (32 B each) up to cap.
// Returns the number of rays produced; *fails = kd-tree results rejected by
// CheckedIntersection::is_valid (checked_intersection.rs:10-25) against brute force.
// check == 0 skips the brute-force side (35 G triangle tests at 512 x 512 on cornell_teapot: 97 s on 8 cores) and
// continues every path from the kd-tree hit: the same ray set whenever the checked run finds no fail, which
// tests/golden/make_config2_golden.py establishes once (it stores the checked run's digest).
size_t orc_bounce_rays2(void* hh, uint32_t w, uint32_t h, uint32_t bounces, uint64_t seed, void* rays_out,
                        size_t cap, uint64_t* fails, int threads, int check) {
    SceneH* s = (SceneH*)hh;
    const Scene& sc = s->scene;
    Pinhole cam = Pinhole::make(s->camera.position, s->camera.target, s->camera.up, s->camera.fov, w, h);
    std::vector<uint32_t> all(sc.triangles.size());
    for (uint32_t i = 0; i < all.size(); ++i) all[i] = i;
    const uint32_t n_lights = (uint32_t)sc.lights.size();
    size_t n_pix = (size_t)w * h;
    int nt = std::max(threads, 1);
    std::vector<std::vector<RayIn>> per_thread(nt);
    std::atomic<uint64_t> n_fail{0};
    auto is_valid = [](const std::optional<IndexedIntersection>& a, const std::optional<IndexedIntersection>& b) {
        if (!a && !b) return true;
        if (a && b)
            return a->index == b->index && std::fabs(a->inner.t - b->inner.t) < 0.000001f &&
                   std::fabs(a->inner.u - b->inner.u) < 0.00001f &&
                   std::fabs(a->inner.v - b->inner.v) < 0.00001f;
        return false;
    };
    parallel_for(n_pix, nt, [&](int tid, size_t a, size_t b) {
        std::vector<RayIn>& out = per_thread[tid];
        for (size_t pix = a; pix < b; ++pix) {
            uint32_t x = (uint32_t)(pix % w), y = (uint32_t)(pix / w);
            Sampler rng;
            rng.mode = Sampler::COUNTER;
            rng.key = ptx_path_key(seed, (uint32_t)pix, 0);
            Ray ray = sample_ray_for_pixel(cam, rng, x, y);
            for (uint32_t acc = 0; acc < bounces; ++acc) {
                const uint32_t base = ptx_dim_base(acc + 1, n_lights);
                const float fmax = std::numeric_limits<float>::max();
                out.push_back({ray.origin.x, ray.origin.y, ray.origin.z, 0.0f, ray.direction.x,
                               ray.direction.y, ray.direction.z, fmax});
                auto kd = kd_intersect(sc.kdtree.get(), sc.triangles, ray, 0.0f, fmax);
                auto ref = check ? intersect_closest_geometry(sc.triangles, all.begin(), all.end(), ray, 0.0f, fmax) : kd;
                if (!is_valid(ref, kd)) { ++n_fail; break; }
                if (!ref) break;
                const TriangleProperties& prop = sc.properties[ref->index];
                Vec3 wi = -ray.direction;
                Vec3 n = prop.compute_normal(ref->inner.u, ref->inner.v);
                Vec2 uv = prop.compute_texcoord(ref->inner.u, ref->inner.v);
                const Material& material = sc.materials[prop.material];
                Vec3 offset = 0.00001f * n;
                Vec3 point = ray.param(ref->inner.t);
                Vec3 point_above = point + offset;
                Vec3 point_below = point - offset;
                bool shadow_fail = false;
                for (uint32_t l = 0; l < n_lights; ++l) {
                    Ray sr = sc.lights[l].sample_shadow_ray(point_above, rng, base + 3 + 2 * l);
                    out.push_back({sr.origin.x, sr.origin.y, sr.origin.z, 0.0f, sr.direction.x,
                                   sr.direction.y, sr.direction.z, 1.0f});
                    if (!check) continue;
                    auto skd = kd_intersect(sc.kdtree.get(), sc.triangles, sr, 0.0f, 1.0f);
                    auto sref = intersect_closest_geometry(sc.triangles, all.begin(), all.end(), sr, 0.0f, 1.0f);
                    if (!is_valid(sref, skd)) shadow_fail = true;
                }
                if (shadow_fail) { ++n_fail; break; }
                BsdfSample sample = material_sample(material, sc.textures, Surface{wi, n, uv}, rng, base);
                ray = Ray{dot(sample.wo, n) >= 0.0f ? point_above : point_below, sample.wo};
            }
        }
    });
    size_t total = 0;
    RayIn* dst = (RayIn*)rays_out;
    for (auto& v : per_thread) {
        for (const RayIn& r : v) {
            if (dst && total < cap) dst[total] = r;
            ++total;
        }
    }
    if (fails) *fails = n_fail.load();
    return total;
}

size_t orc_bounce_rays(void* hh, uint32_t w, uint32_t h, uint32_t bounces, uint64_t seed, void* rays_out,
                       size_t cap, uint64_t* fails, int threads) {
    return orc_bounce_rays2(hh, w, h, bounces, seed, rays_out, cap, fails, threads, 1);
}


// The reference's ray log (raylogger.rs:39-66; pathtracer.rs:34-43, 69-71; worker.rs:36-47): what ONE thread
// rendering iterations first_sample .. first_sample+count-1 with the COUNTER sampler writes to its
// raylog file, 33 bytes per ray.  Returns the number of records; only the first cap are stored.
size_t orc_render_raylog(void* hh, uint32_t w, uint32_t h, uint64_t seed, uint32_t first_sample, uint32_t count,
                         uint8_t* out, size_t cap) {
    SceneH* s = (SceneH*)hh;
    const Scene& sc = s->scene;
    Pinhole cam = Pinhole::make(s->camera.position, s->camera.target, s->camera.up, s->camera.fov, w, h);
    size_t total = 0;
    std::vector<RayRecord> rec;
    for (uint32_t it = first_sample; it < first_sample + count; ++it) {
        for (uint32_t y = 0; y < h; ++y) {
            for (uint32_t x = 0; x < w; ++x) {
                Sampler rng;
                rng.mode = Sampler::COUNTER;
                rng.key = ptx_path_key(seed, y * w + x, it);
                Ray ray = sample_ray_for_pixel(cam, rng, x, y);
                rec.clear();
                trace_ray(sc, rng, ray, nullptr, &rec);
                uint8_t bounce = 0;
                for (const RayRecord& r : rec) {
                    const bool shadow = r.t_hi == 1.0f;
                    if (!shadow) ++bounce;
                    Ray logged = r.ray;
                    if (!shadow && r.hit) logged = Ray{r.ray.origin, r.hit->inner.t * r.ray.direction};  // Ray::extended
                    if (total < cap) {
                        uint8_t* b = out + 33 * total;
                        const uint16_t i16 = (uint16_t)it, px = (uint16_t)x, py = (uint16_t)y;
                        std::memcpy(b, &i16, 2);
                        std::memcpy(b + 2, &px, 2);
                        std::memcpy(b + 4, &py, 2);
                        b[6] = bounce;
                        b[7] = shadow ? 1 : 0;
                        b[8] = r.hit ? 1 : 0;
                        const float f[6] = {logged.origin.x, logged.origin.y, logged.origin.z,
                                            logged.direction.x, logged.direction.y, logged.direction.z};
                        std::memcpy(b + 9, f, 24);
                    }
                    ++total;
                }
            }
        }
    }
    return total;
}

}  // extern "C"
