// host_scene.cpp — CPU-resident parts of the product library: Wavefront OBJ + the reference's MTL
// dialect, triangle/property extraction, material and camera set-up, tone mapping, flat-tree
// validation, triangle packing and the scalar host query.  No dependency on oracle/.
//
// Reference behaviour mirrored (file:line under the reference root):
//   wavefront/src/obj.rs:117-166, mtl.rs:80-157, lib.rs:10-30      parsing + "Loading ..." lines
//   tracing/src/properties.rs:49-90                                 from_wavefront
//   pathtracer-cli/src/main.rs:123-159                              setup_scene
//   tracing/src/material.rs:153-164, light.rs:79-89, camera.rs:15-24,71-85
//   tracing/src/image_buffer.rs:11-13,29-38                         gamma + u8 conversion
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <iterator>
#include <limits>
#include <map>
#include <memory>
#include <sstream>
#include <thread>
#include <stdexcept>

#include "host_geom.hpp"
#include "host_internal.hpp"

namespace ptx {

static thread_local std::string g_error;
int fail(int code, const std::string& what) {
    g_error = what;
    return code;
}
void clear_error() { g_error.clear(); }

// ------------------------------------------------------------------------------------ text scanning
namespace {

struct Cursor {
    const char* p;
    void blanks() {
        while (*p == ' ' || *p == '\t') ++p;
    }
    bool keyword(const char* kw, bool fold_case) {
        const char* q = p;
        for (; *kw; ++kw, ++q) {
            int a = (unsigned char)*q, b = (unsigned char)*kw;
            if (fold_case) {
                a = tolower(a);
                b = tolower(b);
            }
            if (a != b) return false;
        }
        p = q;
        blanks();
        return true;
    }
    bool real(float& out) {
        if (*p == 0 || isspace((unsigned char)*p)) return false;
        char* end;
        out = strtof(p, &end);
        if (end == p) return false;
        p = end;
        return true;
    }
    bool reals(float* out, int n) {
        for (int i = 0; i < n; ++i) {
            if (i) blanks();
            if (!real(out[i])) return false;
        }
        return true;
    }
    bool whole(int& out) {
        const char* q = p;
        if (*q == '+' || *q == '-') ++q;
        if (!isdigit((unsigned char)*q)) return false;
        char* end;
        out = (int)strtol(p, &end, 10);
        p = end;
        return true;
    }
};

std::string strip(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) ++a;
    while (b > a && isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

struct Corner {
    int v, t, n;
};
struct Group {
    std::string material;
    std::vector<std::vector<Corner>> faces;
};
struct ObjFile {
    std::string mtllib;
    std::vector<float> v, vn, vt;  // 3,3,2 floats per entry
    std::vector<Group> groups;
};
struct MtlEntry {
    std::string name, map_kd;
    float kd[3] = {0.7f, 0.7f, 0.7f}, ks[3] = {0, 0, 0};
    float transparency = 0, ni = 1.0f, pm = 0;
};
struct LightEntry {
    float position[3] = {0, 0, 0}, color[3] = {0, 0, 0}, radius = 0, intensity = 0;
};
struct CameraEntry {
    float position[3] = {0, 0, 0}, target[3] = {0, 0, 0}, up[3] = {0, 0, 0}, fov = 0;
};
struct MtlFile {
    std::vector<MtlEntry> materials;
    std::vector<LightEntry> lights;
    std::vector<CameraEntry> cameras;
};

bool corner(Cursor& c, Corner& out) {  // obj.rs:102-109: v/[t]/[n]
    Cursor q = c;
    if (!q.whole(out.v) || *q.p != '/') return false;
    ++q.p;
    if (!q.whole(out.t)) out.t = 0;
    if (*q.p != '/') return false;
    ++q.p;
    if (!q.whole(out.n)) out.n = 0;
    c = q;
    return true;
}

// One part of an OBJ file: what its lines add, in file order.  `head` holds the faces that come before the
// part's first usemtl: they belong to the group that is open when the part starts.
struct ObjPart {
    std::string mtllib;
    bool has_mtllib = false;
    std::vector<float> v, vn, vt;
    std::vector<std::vector<Corner>> head;
    std::string head_line;  // first face line of `head` (for the "face before any usemtl" message)
    std::vector<Group> groups;
    std::string error;  // first malformed line of the part
};

void parse_obj_part(const char* p, const char* end, ObjPart& part) {
    std::string raw;
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* stop = nl ? nl : end;
        raw.assign(p, stop);
        p = nl ? nl + 1 : end;
        std::string line = strip(raw);
        if (line.empty() || line[0] == '#') continue;
        Cursor c{line.c_str()};
        float f[3];
        // tag order of obj.rs:135-155; the tags are matched case-insensitively (tag_no_case)
        if (c = Cursor{line.c_str()}, c.keyword("mtllib", true)) {
            part.mtllib = c.p;
            part.has_mtllib = true;
        } else if (c = Cursor{line.c_str()}, c.keyword("usemtl", true)) {
            part.groups.push_back(Group{c.p, {}});
        } else if (c = Cursor{line.c_str()}, c.keyword("v", true) && c.reals(f, 3)) {
            part.v.insert(part.v.end(), f, f + 3);
        } else if (c = Cursor{line.c_str()}, c.keyword("vn", true) && c.reals(f, 3)) {
            part.vn.insert(part.vn.end(), f, f + 3);
        } else if (c = Cursor{line.c_str()}, c.keyword("vt", true) && c.reals(f, 2)) {
            part.vt.insert(part.vt.end(), f, f + 2);
        } else if (c = Cursor{line.c_str()}, c.keyword("f", true)) {
            std::vector<Corner> face;
            Corner k;
            if (corner(c, k)) {
                face.push_back(k);
                for (;;) {
                    Cursor q = c;
                    if (*q.p != ' ' && *q.p != '\t') break;
                    q.blanks();
                    if (!corner(q, k)) break;
                    face.push_back(k);
                    c = q;
                }
            }
            if (part.groups.empty()) {
                if (part.head.empty()) part.head_line = raw;
                part.head.push_back(face);
            } else {
                part.groups.back().faces.push_back(face);
            }
        } else if ((c = Cursor{line.c_str()}, c.keyword("g", true)) ||
                   (c = Cursor{line.c_str()}, c.keyword("o", true)) ||
                   (c = Cursor{line.c_str()}, c.keyword("s", true))) {
            // ignored, as in the reference
        } else {
            part.error = "Unexpected line: \"" + raw + "\"";
            return;  // the reference panics here (obj.rs:154): nothing after the line matters
        }
    }
}

// `obj::obj` (wavefront/src/obj.rs:117-166).  Lines are independent except for "the group that is open", so the
// file is cut at line ends into one part per thread (PTX_BUILD_THREADS, at most 16; small files stay on the calling
// thread), the parts are parsed concurrently and spliced in file order: the result, and the first error, are those
// of a sequential pass.
void read_obj(std::istream& in, ObjFile& obj) {
    std::string text((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    unsigned want = std::thread::hardware_concurrency();
    if (want > 16) want = 16;
    if (const char* e = getenv("PTX_BUILD_THREADS")) want = (unsigned)std::max(1, atoi(e));
    if (want < 1 || text.size() < (1u << 18)) want = 1;
    std::vector<size_t> cut{0};
    for (unsigned k = 1; k < want; ++k) {
        size_t at = text.size() * k / want;
        const size_t nl = text.find('\n', at);
        at = nl == std::string::npos ? text.size() : nl + 1;
        if (at > cut.back() && at < text.size()) cut.push_back(at);
    }
    cut.push_back(text.size());
    const size_t n_parts = cut.size() - 1;
    std::vector<ObjPart> parts(n_parts);
    std::vector<std::exception_ptr> thrown(n_parts);
    auto work = [&](size_t k) {
        try {
            parse_obj_part(text.data() + cut[k], text.data() + cut[k + 1], parts[k]);
        } catch (...) {
            thrown[k] = std::current_exception();
        }
    };
    {
        std::vector<std::thread> pool;
        for (size_t k = 1; k < n_parts; ++k) pool.emplace_back(work, k);
        work(0);
        for (auto& t : pool) t.join();
    }
    for (size_t k = 0; k < n_parts; ++k) {  // splice in file order; the first problem in file order wins
        if (thrown[k]) std::rethrow_exception(thrown[k]);
        ObjPart& part = parts[k];
        if (part.has_mtllib) obj.mtllib = part.mtllib;
        obj.v.insert(obj.v.end(), part.v.begin(), part.v.end());
        obj.vn.insert(obj.vn.end(), part.vn.begin(), part.vn.end());
        obj.vt.insert(obj.vt.end(), part.vt.begin(), part.vt.end());
        if (!part.head.empty()) {
            if (obj.groups.empty()) throw std::runtime_error("face before any usemtl: \"" + part.head_line + "\"");
            auto& faces = obj.groups.back().faces;
            faces.insert(faces.end(), std::make_move_iterator(part.head.begin()), std::make_move_iterator(part.head.end()));
        }
        for (auto& g : part.groups) obj.groups.push_back(std::move(g));
        if (!part.error.empty()) throw std::runtime_error(part.error);
    }
}

void read_mtl(std::istream& in, MtlFile& mtl) {
    std::string raw;
    while (std::getline(in, raw)) {
        std::string line = strip(raw);
        if (line.empty() || line[0] == '#') continue;
        Cursor c{line.c_str()};
        float f[3];
        auto key = [&](const char* k) {
            c = Cursor{line.c_str()};
            return c.keyword(k, false);
        };
        auto key1 = [&](const char* k) { return key(k) && c.real(f[0]); };
        auto key3 = [&](const char* k) { return key(k) && c.reals(f, 3); };
        auto light = [&]() -> LightEntry& {
            if (mtl.lights.empty()) throw std::runtime_error("light property before newlight");
            return mtl.lights.back();
        };
        auto camera = [&]() -> CameraEntry& {
            if (mtl.cameras.empty()) throw std::runtime_error("camera property before newcamera");
            return mtl.cameras.back();
        };
        auto material = [&]() -> MtlEntry& {
            if (mtl.materials.empty()) throw std::runtime_error("material property before newmtl");
            return mtl.materials.back();
        };
        auto set3 = [&](float* dst) { dst[0] = f[0], dst[1] = f[1], dst[2] = f[2]; };
        // tag order of mtl.rs:96-148 (case-sensitive `tag`)
        if (key("newlight")) mtl.lights.emplace_back();
        else if (key3("lightposition")) set3(light().position);
        else if (key3("lightcolor")) set3(light().color);
        else if (key1("lightradius")) light().radius = f[0];
        else if (key1("lightintensity")) light().intensity = f[0];
        else if (key("newcamera")) mtl.cameras.emplace_back();
        else if (key3("cameraposition")) set3(camera().position);
        else if (key3("cameratarget")) set3(camera().target);
        else if (key3("cameraup")) set3(camera().up);
        else if (key1("camerafov")) camera().fov = f[0];
        else if (key("newmtl")) {
            mtl.materials.emplace_back();
            mtl.materials.back().name = c.p;
        } else if (key1("illum")) {
        } else if (key1("Ka")) {
        } else if (key3("Kd")) set3(material().kd);
        else if (key("map_Kd")) material().map_kd = c.p;
        else if (key3("Ks")) set3(material().ks);
        else if (key1("Ns")) {
        } else if (key1("Ke")) {
        } else if (key1("reflat0deg")) material();   // parsed and unused (mtl.rs:132-135)
        else if (key1("reflat90deg")) material();
        else if (key1("Ni")) material().ni = f[0];
        else if (key1("d")) material().transparency = 1.0f - f[0];
        else if (key1("Tr")) material().transparency = f[0];
        else if (key1("Pm")) material().pm = f[0];
        else if (key1("specularroughness")) {
        } else throw std::runtime_error("Unexpected line: \"" + raw + "\"");
    }
}

// obj.rs:62-68: 1-based, negative counts from the end, 0 = default (zeros)
void fetch(const std::vector<float>& pool, int width, int i, float* out) {
    int count = (int)(pool.size() / width);
    int at;
    if (i == 0) {
        for (int k = 0; k < width; ++k) out[k] = 0.0f;
        return;
    }
    at = i < 0 ? count + i : i - 1;
    if (at < 0 || at >= count) throw std::runtime_error("OBJ index out of range");
    for (int k = 0; k < width; ++k) out[k] = pool[(size_t)at * width + k];
}

}  // namespace

}  // namespace ptx

// ------------------------------------------------------------------------------------ loaded scene
struct ptx_loaded {
    std::vector<float> triangles, normals, texcoords;
    std::vector<uint32_t> material_ids;
    std::vector<ptx_kdnode> nodes;
    std::vector<uint32_t> leaf_indices;
    std::vector<ptx_material> materials;
    std::vector<std::vector<float>> texture_data;
    std::vector<ptx_texture> textures;
    std::vector<ptx_light> lights;
    float camera[10];
    uint32_t max_bounces;
};

struct ptx_kdtree {
    std::vector<ptx_kdnode> nodes;
    std::vector<uint32_t> leaves;
};

namespace ptx {
ptx_kdtree* kdtree_from_arrays(std::vector<ptx_kdnode>&& nodes, std::vector<uint32_t>&& leaves) {
    auto t = std::make_unique<ptx_kdtree>();
    t->nodes = std::move(nodes);
    t->leaves = std::move(leaves);
    return t.release();
}
}  // namespace ptx

using namespace ptx;

extern "C" {

const char* ptx_last_error(void) { return g_error.c_str(); }
const char* ptx_version(void) { return "ptx-mi355x 0.1 (gfx950)"; }
void ptx_free(void* p) { free(p); }

int ptx_material_make(const float kd[3], const float ks[3], float ni, float pm, float transparency,
                      int32_t texture, ptx_material* out) {
    if (!kd || !ks || !out) return fail(PTX_E_ARG, "ptx_material_make: null argument");
    float q = (ni - 1.0f) / (ni + 1.0f);
    float f0 = q * q;  // powi(2)
    for (int c = 0; c < 3; ++c) {
        out->albedo[c] = kd[c];
        out->schlick_f0[c] = f0 * (1.0f - pm) + ks[c] * pm;  // glam lerp: a*(1-s) + b*s
    }
    out->texture = texture;
    out->transmission = transparency;
    out->ior = ni;
    return PTX_OK;
}

int ptx_pinhole_make(const float position[3], const float target[3], const float up[3],
                     float fov_degrees, uint32_t width, uint32_t height, ptx_pinhole* out) {
    if (!position || !target || !up || !out || width == 0 || height == 0)
        return fail(PTX_E_ARG, "ptx_pinhole_make: bad argument");
    F3 pos = f3(position[0], position[1], position[2]);
    F3 dir = unit(sub(f3(target[0], target[1], target[2]), pos));
    F3 upv = f3(up[0], up[1], up[2]);
    F3 upn = unit(upv);
    F3 right = unit(cross3(dir, upv));  // uses the un-normalised `up` argument (camera.rs:21)
    float aspect = (float)width / (float)height;
    float half = fov_degrees * 3.14159265358979323846f / 360.0f;
    F3 x = scale(right, std::sin(half) * aspect);
    F3 y = scale(upn, std::sin(half));
    F3 z = scale(dir, std::cos(half));
    F3 plane = sub(add(z, y), x);
    for (int c = 0; c < 3; ++c) {
        out->position[c] = pos[c];
        out->plane[c] = plane[c];
        out->dx[c] = 2.0f * x[c];
        out->dy[c] = -2.0f * y[c];
    }
    out->width = width;
    out->height = height;
    return PTX_OK;
}

int ptx_tonemap_rgb8(const float* accum, uint32_t width, uint32_t height, uint32_t iterations,
                     uint8_t* out) {
    if (!accum || !out) return fail(PTX_E_ARG, "ptx_tonemap_rgb8: null argument");
    const uint16_t iters = (uint16_t)iterations;  // `total_iterations as u16` (worker.rs:113)
    const float inv = 1.0f / (float)iters;
    const size_t n = (size_t)width * height * 3;
    auto span = [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            float g = std::pow(accum[i] * inv, 1.0f / 2.2f);
            g = g < 1.0f ? g : 1.0f;
            float c = std::round(g * 255.0f);
            out[i] = !(c == c) ? 0 : (c <= 0.0f ? 0 : (c >= 255.0f ? 255 : (uint8_t)c));
        }
    };
    // every value is independent: slices on a few threads (the powf dominates; 3.5 M of them at 1080^2)
    unsigned workers = std::thread::hardware_concurrency();
    workers = workers > 16 ? 16 : (workers < 1 ? 1 : workers);
    if (n < (1u << 16) || workers == 1) {
        span(0, n);
    } else {
        std::vector<std::thread> pool;
        const size_t step = (n + workers - 1) / workers;
        for (unsigned w = 0; w < workers; ++w) {
            const size_t lo = (size_t)w * step, hi = lo + step < n ? lo + step : n;
            if (lo < hi) pool.emplace_back(span, lo, hi);
        }
        for (auto& t : pool) t.join();
    }
    return PTX_OK;
}

int ptx_kdtree_build(const float* triangles, uint32_t n_triangles, const ptx_sah_cost* sah,
                     ptx_kdtree** out) {
    if ((!triangles && n_triangles) || !out) return fail(PTX_E_ARG, "ptx_kdtree_build: null argument");
    ptx_sah_cost cost = sah ? *sah : ptx_sah_cost{1.0f, 1.5f, 0.8f};
    try {
        auto t = std::make_unique<ptx_kdtree>();
        build_flat_kdtree(triangles, n_triangles, cost, t->nodes, t->leaves);
        *out = t.release();
        return PTX_OK;
    } catch (std::bad_alloc&) {
        return fail(PTX_E_OOM, "ptx_kdtree_build: out of memory");
    }
}

int ptx_kdtree_view(const ptx_kdtree* tree, const ptx_kdnode** nodes, uint32_t* n_nodes,
                    const uint32_t** leaf_indices, uint32_t* n_leaf_indices) {
    if (!tree) return fail(PTX_E_ARG, "ptx_kdtree_view: null tree");
    if (nodes) *nodes = tree->nodes.data();
    if (n_nodes) *n_nodes = (uint32_t)tree->nodes.size();
    if (leaf_indices) *leaf_indices = tree->leaves.data();
    if (n_leaf_indices) *n_leaf_indices = (uint32_t)tree->leaves.size();
    return PTX_OK;
}

void ptx_kdtree_free(ptx_kdtree* tree) { delete tree; }

int ptx_kdtree_intersect_host(const float* triangles, uint32_t n_triangles, const ptx_kdnode* nodes,
                              uint32_t n_nodes, const uint32_t* leaf_indices, const ptx_ray* ray,
                              ptx_hit* hit) {
    if (!nodes || !ray || !hit || n_nodes == 0) return fail(PTX_E_ARG, "ptx_kdtree_intersect_host: null argument");
    if (n_triangles && !triangles) return fail(PTX_E_ARG, "ptx_kdtree_intersect_host: triangles missing");
    // The walk indexes with what the tree says: prove it sound first.  The ABI carries no length for the leaf
    // index list, so the list must cover every leaf range the tree names (documented in ptx.h); what CAN be
    // checked is - children, ranges below 2^30, depth, and every listed triangle index below n_triangles.
    uint64_t need = 0;
    for (uint32_t i = 0; i < n_nodes; ++i)
        if ((nodes[i].w1 & 3u) == 3u) need = std::max<uint64_t>(need, (uint64_t)nodes[i].w0 + (nodes[i].w1 >> 2));
    if (need >= (1ull << 30) || (need && !leaf_indices)) return fail(PTX_E_ARG, "ptx_kdtree_intersect_host: leaf ranges out of bounds");
    const std::string why = validate_flat_tree(nodes, n_nodes, leaf_indices, (uint32_t)need, n_triangles);
    if (!why.empty()) return fail(PTX_E_ARG, "ptx_kdtree_intersect_host: invalid flattened tree: " + why);
    struct OnDemand {  // packs only the triangles of the leaves the ray visits
        const float* t;
        TriRecord operator[](uint32_t i) const {
            TriRecord r;
            pack_triangle(t + 9 * (size_t)i, r);
            return r;
        }
    } tris{triangles};
    WorkCount wc{0, 0};
    kd_traverse<false, false>(nodes, leaf_indices, tris, ray->ox, ray->oy, ray->oz, ray->dx,
                              ray->dy, ray->dz, ray->tmin, ray->tmax, *hit, wc);
    return PTX_OK;
}

int ptx_load_obj(const char* obj_path, const ptx_sah_cost* sah, uint32_t max_bounces, int verbose,
                 ptx_loaded** out) {
    if (!obj_path || !out) return fail(PTX_E_ARG, "ptx_load_obj: null argument");
    try {
        auto L = std::make_unique<ptx_loaded>();
        ObjFile obj;
        MtlFile mtl;
        if (verbose) printf("Loading %s...\n", obj_path);
        {
            std::ifstream in(obj_path);
            if (!in) return fail(PTX_E_IO, std::string("cannot open ") + obj_path);
            read_obj(in, obj);
        }
        if (verbose) {
            printf("  Chunks: %zu\n  Vertices: %zu\n  Normals: %zu\n  Texcoords: %zu\n", obj.groups.size(),
                   obj.v.size() / 3, obj.vn.size() / 3, obj.vt.size() / 2);
        }
        std::string base(obj_path);
        size_t slash = base.find_last_of('/');
        std::string dir = slash == std::string::npos ? std::string() : base.substr(0, slash + 1);
        std::string mtl_path = dir + obj.mtllib;
        if (verbose) printf("Loading %s...\n", mtl_path.c_str());
        {
            std::ifstream in(mtl_path);
            if (!in) return fail(PTX_E_IO, "cannot open " + mtl_path);
            read_mtl(in, mtl);
        }
        if (verbose)
            printf("  Materials: %zu\n  Lights: %zu\n  Cameras: %zu\n", mtl.materials.size(), mtl.lights.size(),
                   mtl.cameras.size());
        if (mtl.cameras.empty()) return fail(PTX_E_IO, "scene has no camera");
        if (mtl.lights.size() > PTX_MAX_LIGHTS) return fail(PTX_E_LIMIT, "more than PTX_MAX_LIGHTS lights");

        std::map<std::string, uint32_t> by_name;
        for (size_t i = mtl.materials.size(); i-- > 0;) by_name[mtl.materials[i].name] = (uint32_t)i;  // first wins
        for (const Group& g : obj.groups) {
            for (const auto& face : g.faces) {
                if (face.size() != 3)
                    return fail(PTX_E_IO, "Only triangular faces supported but found " + std::to_string(face.size()) + " vertices.");
                auto m = by_name.find(g.material);
                if (m == by_name.end()) return fail(PTX_E_IO, "unknown material " + g.material);
                float tmp[3];
                for (int k = 0; k < 3; ++k) {
                    fetch(obj.v, 3, face[k].v, tmp);
                    L->triangles.insert(L->triangles.end(), tmp, tmp + 3);
                }
                for (int k = 0; k < 3; ++k) {
                    fetch(obj.vn, 3, face[k].n, tmp);
                    L->normals.insert(L->normals.end(), tmp, tmp + 3);
                }
                for (int k = 0; k < 3; ++k) {
                    fetch(obj.vt, 2, face[k].t, tmp);
                    L->texcoords.insert(L->texcoords.end(), tmp, tmp + 2);
                }
                L->material_ids.push_back(m->second);
            }
        }
        for (const MtlEntry& e : mtl.materials) {
            ptx_material pm;
            int32_t tex = -1;
            if (!e.map_kd.empty()) {
                float* rgb = nullptr;
                uint32_t w = 0, h = 0;
                int rc = ptx_png_read_linear((dir + e.map_kd).c_str(), &rgb, &w, &h);
                if (rc != PTX_OK) return rc;
                L->texture_data.emplace_back(rgb, rgb + (size_t)w * h * 3);
                free(rgb);
                L->textures.push_back(ptx_texture{w, h, nullptr});
                tex = (int32_t)L->textures.size() - 1;
            }
            ptx_material_make(e.kd, e.ks, e.ni, e.pm, e.transparency, tex, &pm);
            L->materials.push_back(pm);
        }
        for (size_t i = 0; i < L->textures.size(); ++i) L->textures[i].rgb = L->texture_data[i].data();
        for (const LightEntry& e : mtl.lights) {
            ptx_light pl;
            for (int c = 0; c < 3; ++c) {
                pl.center[c] = e.position[c];
                pl.intensity[c] = e.color[c] * e.intensity;
            }
            pl.radius = e.radius;
            L->lights.push_back(pl);
        }
        const CameraEntry& cam = mtl.cameras[0];  // always the first camera (main.rs:137)
        for (int c = 0; c < 3; ++c) {
            L->camera[c] = cam.position[c];
            L->camera[3 + c] = cam.target[c];
            L->camera[6 + c] = cam.up[c];
        }
        L->camera[9] = cam.fov;
        L->max_bounces = max_bounces;
        if (verbose) printf("Building kdtree...\n");
        ptx_sah_cost cost = sah ? *sah : ptx_sah_cost{1.0f, 1.5f, 0.8f};
        build_flat_kdtree(L->triangles.data(), (uint32_t)L->material_ids.size(), cost, L->nodes, L->leaf_indices);
        *out = L.release();
        return PTX_OK;
    } catch (std::bad_alloc&) {
        return fail(PTX_E_OOM, "ptx_load_obj: out of memory");
    } catch (std::exception& e) {
        return fail(PTX_E_IO, e.what());
    }
}

int ptx_loaded_desc(const ptx_loaded* L, ptx_scene_desc* d) {
    if (!L || !d) return fail(PTX_E_ARG, "ptx_loaded_desc: null argument");
    memset(d, 0, sizeof *d);
    d->triangles = L->triangles.data();
    d->n_triangles = (uint32_t)L->material_ids.size();
    d->nodes = L->nodes.data();
    d->n_nodes = (uint32_t)L->nodes.size();
    d->leaf_indices = L->leaf_indices.data();
    d->n_leaf_indices = (uint32_t)L->leaf_indices.size();
    d->normals = L->normals.data();
    d->texcoords = L->texcoords.data();
    d->material_ids = L->material_ids.data();
    d->materials = L->materials.data();
    d->n_materials = (uint32_t)L->materials.size();
    d->textures = L->textures.data();
    d->n_textures = (uint32_t)L->textures.size();
    d->lights = L->lights.data();
    d->n_lights = (uint32_t)L->lights.size();
    d->environment[0] = d->environment[1] = d->environment[2] = 0.8f;  // main.rs:155
    d->max_bounces = L->max_bounces;
    return PTX_OK;
}

int ptx_loaded_camera(const ptx_loaded* L, float out10[10]) {
    if (!L || !out10) return fail(PTX_E_ARG, "ptx_loaded_camera: null argument");
    memcpy(out10, L->camera, sizeof L->camera);
    return PTX_OK;
}

void ptx_loaded_free(ptx_loaded* L) { delete L; }

}  // extern "C"

namespace ptx {

void pack_triangles(const float* t, uint32_t n, std::vector<TriRecord>& out) {
    out.resize(n);
    for (uint32_t i = 0; i < n; ++i) pack_triangle(t + 9 * (size_t)i, out[i]);
}

void pack_triangle(const float* v, TriRecord& r) {
    memset(&r, 0, sizeof r);
    int axis = aligned_axis(v);
    r.kind = (uint32_t)axis;
    if (axis == 3) {
        for (int c = 0; c < 3; ++c) {
            r.f[c] = v[c];
            r.f[3 + c] = v[3 + c] - v[c];
            r.f[6 + c] = v[6 + c] - v[c];
        }
    } else {
        const int a0 = axis == 0 ? 1 : 0;  // Axis::remove_from (axis.rs:27-33)
        const int a1 = axis == 2 ? 1 : 2;
        r.f[0] = v[axis];
        r.f[1] = v[a0];
        r.f[2] = v[a1];
        r.f[3] = v[3 + a0] - v[a0];
        r.f[4] = v[3 + a1] - v[a1];
        r.f[5] = v[6 + a0] - v[a0];
        r.f[6] = v[6 + a1] - v[a1];
        const float det = r.f[3] * r.f[6] - r.f[5] * r.f[4];  // axial_triangle.rs:70
        r.f[7] = det == 0.0f ? std::numeric_limits<float>::quiet_NaN() : 1.0f / det;
    }
}

std::string validate_flat_tree(const ptx_kdnode* nodes, uint32_t n_nodes, const uint32_t* leaf,
                               uint32_t n_leaf, uint32_t n_tris) {
    if (n_nodes == 0) return "tree has no nodes";
    if (n_nodes >= (1u << 30) || n_leaf >= (1u << 30)) return "tree too large for the 30-bit fields";
    // iterative preorder walk with explicit depth; every node must be visited exactly once
    struct Item {
        uint32_t node, depth;
    };
    std::vector<Item> stack{{0, 1}};
    std::vector<uint8_t> seen(n_nodes, 0);
    uint32_t visited = 0;
    while (!stack.empty()) {
        Item it = stack.back();
        stack.pop_back();
        if (it.node >= n_nodes) return "child index out of range";
        if (seen[it.node]) return "node reachable twice";
        seen[it.node] = 1;
        ++visited;
        const ptx_kdnode& nd = nodes[it.node];
        if ((nd.w1 & 3u) == 3u) {
            uint32_t count = nd.w1 >> 2;
            if ((uint64_t)nd.w0 + count > n_leaf) return "leaf range outside the index list";
            for (uint32_t i = 0; i < count; ++i)
                if (leaf[nd.w0 + i] >= n_tris) return "triangle index out of range";
        } else {
            if (it.depth >= PTX_MAX_DEPTH) return "tree deeper than PTX_MAX_DEPTH";
            uint32_t right = nd.w1 >> 2;
            if (right <= it.node + 1 || right >= n_nodes) return "right child must follow the left subtree";
            stack.push_back({right, it.depth + 1});
            stack.push_back({it.node + 1, it.depth + 1});
        }
    }
    if (visited != n_nodes) return "unreachable nodes in the array";
    return std::string();
}

}  // namespace ptx
