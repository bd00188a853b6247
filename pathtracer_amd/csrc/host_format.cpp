// host_format.cpp — tree interchange and statistics: the reference's JSON tree dump, its
// pretty dump, and the kdtree-cli statistics, over the flattened tree of include/ptx.h.
//   write_tree_json / write_node_json     kdtree/src/format.rs:94-130
//   write_node_pretty                     kdtree/src/format.rs:22-47
//   statistics, tree_cost                 kdtree-cli/src/main.rs:45-159
// Numbers are printed as Rust prints them: `{}` (Display) = shortest digits that round-trip, never an
// exponent; `{:?}` (Debug) = the same digits with at least ".0", or `d.ddde-7` outside [1e-4, 1e16).
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "host_geom.hpp"
#include "host_internal.hpp"

namespace ptx {
namespace {

void put_display(std::string& out, float v) {
    if (std::isnan(v)) {
        out += "NaN";
        return;
    }
    if (std::isinf(v)) {
        out += v < 0 ? "-inf" : "inf";
        return;
    }
    // shortest round-trip digits (scientific form), laid out positionally with zero padding, as
    // core::fmt::float::float_to_decimal_display does (to_chars' own fixed form prints exact digits instead)
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);
    std::string sci(buf, r.ptr);
    size_t at = 0;
    if (sci[0] == '-') {
        out += '-';
        at = 1;
    }
    const size_t e = sci.find('e');
    std::string digits;
    for (size_t k = at; k < e; ++k)
        if (sci[k] != '.') digits += sci[k];
    const int exp10 = atoi(sci.c_str() + e + 1);
    if (exp10 >= 0) {
        const size_t int_len = (size_t)exp10 + 1;
        if (digits.size() <= int_len) {
            out += digits;
            out.append(int_len - digits.size(), '0');
        } else {
            out.append(digits, 0, int_len);
            out += '.';
            out.append(digits, int_len, std::string::npos);
        }
    } else {
        out += "0.";
        out.append((size_t)(-exp10 - 1), '0');
        out += digits;
    }
}

void put_debug(std::string& out, float v) {
    if (std::isnan(v) || std::isinf(v)) {
        put_display(out, v);
        return;
    }
    const float a = std::fabs(v);
    if (a != 0.0f && (a < 1e-4f || a >= 1e16f)) {  // core::fmt::float::float_to_general_debug
        char buf[64];
        auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);
        std::string s(buf, r.ptr);  // d.ddde+XX / d.ddde-XX
        const size_t e = s.find('e');
        out.append(s, 0, e + 1);
        size_t k = e + 1;
        if (s[k] == '-') out += '-';
        if (s[k] == '-' || s[k] == '+') ++k;
        while (k + 1 < s.size() && s[k] == '0') ++k;
        out.append(s, k, std::string::npos);
        return;
    }
    const size_t at = out.size();
    put_display(out, v);
    if (out.find('.', at) == std::string::npos) out += ".0";
}

void put_indices(std::string& out, const uint32_t* idx, uint32_t n) {  // `{:?}` of Vec<u32>
    out += '[';
    for (uint32_t i = 0; i < n; ++i) {
        if (i) out += ", ";
        out += std::to_string(idx[i]);
    }
    out += ']';
}

const char* AXIS_NAME[3] = {"X", "Y", "Z"};

// A flattened tree that has passed validate_flat_tree is a proper preorder binary tree of depth <= 25.
struct Flat {
    const ptx_kdnode* nodes;
    const uint32_t* leaves;
    bool leaf(uint32_t i) const { return (nodes[i].w1 & 3u) == 3u; }
};

void node_json(std::string& out, const Flat& t, uint32_t i) {
    if (t.leaf(i)) {
        put_indices(out, t.leaves + t.nodes[i].w0, t.nodes[i].w1 >> 2);
        return;
    }
    float dist;
    memcpy(&dist, &t.nodes[i].w0, 4);
    out += "{\"axis\": \"";
    out += AXIS_NAME[t.nodes[i].w1 & 3u];
    out += "\", \"distance\": ";
    put_display(out, dist);
    out += ", \"left\": ";
    node_json(out, t, i + 1u);
    out += ", \"right\": ";
    node_json(out, t, t.nodes[i].w1 >> 2);
    out += '}';
}

// ---------------------------------------------------------------------------------------------- reader
struct Reader {
    const char* p;
    const char* end;
    std::string err;
    void ws() {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p;
    }
    bool eat(char c) {
        ws();
        if (p < end && *p == c) {
            ++p;
            return true;
        }
        return false;
    }
    bool expect(char c) {
        if (eat(c)) return true;
        if (err.empty()) err = std::string("expected '") + c + "' (" + std::to_string(end - p) + " bytes before the end)";
        return false;
    }
    bool key(const char* k) {  // "k":
        ws();
        const size_t n = strlen(k);
        if ((size_t)(end - p) < n + 2 || *p != '"' || memcmp(p + 1, k, n) != 0 || p[n + 1] != '"') {
            if (err.empty()) err = std::string("expected key \"") + k + "\"";
            return false;
        }
        p += n + 2;
        return expect(':');
    }
    bool number(float& v) {
        ws();
        char buf[80];
        size_t n = 0;
        while (p + n < end && n + 1 < sizeof buf && (isalnum((unsigned char)p[n]) || p[n] == '-' || p[n] == '+' || p[n] == '.')) ++n;
        if (!n) {
            if (err.empty()) err = "expected a number";
            return false;
        }
        memcpy(buf, p, n);
        buf[n] = 0;
        char* stop = nullptr;
        v = strtof(buf, &stop);
        if (stop != buf + n) {
            if (err.empty()) err = std::string("malformed number '") + buf + "'";
            return false;
        }
        p += n;
        return true;
    }
    bool index(uint32_t& v) {
        ws();
        if (p >= end || *p < '0' || *p > '9') {
            if (err.empty()) err = "expected a triangle index";
            return false;
        }
        unsigned long long x = 0;
        while (p < end && *p >= '0' && *p <= '9') {
            x = x * 10 + (unsigned)(*p - '0');
            if (x > 0xFFFFFFFFull) {
                err = "triangle index does not fit 32 bits";
                return false;
            }
            ++p;
        }
        v = (uint32_t)x;
        return true;
    }
};

bool read_node(Reader& r, std::vector<ptx_kdnode>& nodes, std::vector<uint32_t>& leaves, int depth) {
    if (depth > PTX_MAX_DEPTH) {
        r.err = "tree deeper than the traversal stack allows (25)";
        return false;
    }
    const size_t me = nodes.size();
    nodes.push_back(ptx_kdnode{0u, 3u});
    if (r.eat('[')) {
        const size_t off = leaves.size();
        uint32_t n = 0;
        if (!r.eat(']')) {
            do {
                uint32_t v;
                if (!r.index(v)) return false;
                leaves.push_back(v);
                ++n;
            } while (r.eat(','));
            if (!r.expect(']')) return false;
        }
        nodes[me] = ptx_kdnode{(uint32_t)off, (n << 2) | 3u};
        return true;
    }
    if (!r.expect('{') || !r.key("axis") || !r.expect('"')) return false;
    uint32_t axis = 3;
    if (r.p < r.end) axis = *r.p == 'X' ? 0u : (*r.p == 'Y' ? 1u : (*r.p == 'Z' ? 2u : 3u));
    if (axis == 3u) {
        r.err = "axis must be \"X\", \"Y\" or \"Z\"";
        return false;
    }
    ++r.p;
    float dist;
    if (!r.expect('"') || !r.expect(',') || !r.key("distance") || !r.number(dist) || !r.expect(',') || !r.key("left")) return false;
    if (!read_node(r, nodes, leaves, depth + 1)) return false;
    if (!r.expect(',') || !r.key("right")) return false;
    const size_t right = nodes.size();
    if (right >= (1u << 30)) {
        r.err = "more than 2^30 nodes";
        return false;
    }
    if (!read_node(r, nodes, leaves, depth + 1)) return false;
    uint32_t bits;
    memcpy(&bits, &dist, 4);
    nodes[me] = ptx_kdnode{bits, ((uint32_t)right << 2) | axis};
    return r.expect('}');
}

// -------------------------------------------------------------------------------------------- statistics
struct Stat {
    uint64_t min = 0, max = 0, total = 0;
    float median = 0, mean = 0;
};

Stat compute(std::vector<uint64_t> v) {  // Statistics::compute, kdtree-cli/src/main.rs:113-133, quirks kept
    Stat s;
    if (v.empty()) return s;
    std::sort(v.begin(), v.end());
    const size_t n = v.size();
    if (n == 1) s.median = (float)v[0];
    else if (n % 2 == 0) s.median = (float)v[n / 2];
    else s.median = (float)(v[n / 2] + v[n / 2 + 1]) / 2.0f;
    float sum = 0.0f;
    for (uint64_t x : v) sum += (float)x;  // f32 sum over the sorted values
    s.mean = sum / (float)n;
    s.min = v.front();
    s.max = v.back();
    for (uint64_t x : v) s.total += x;
    return s;
}

bool leaf_empty(const Flat& t, uint32_t i) { return t.leaf(i) && (t.nodes[i].w1 >> 2) == 0u; }

float node_cost(const Flat& t, uint32_t i, const ptx_sah_cost& c, float scene_area, const Box& box) {  // main.rs:45-85
    if (t.leaf(i)) return c.intersect_cost * (float)(t.nodes[i].w1 >> 2) * box.area() / scene_area;
    const float split_cost = box.area() / scene_area;
    float dist;
    memcpy(&dist, &t.nodes[i].w0, 4);
    const uint32_t axis = t.nodes[i].w1 & 3u, l = i + 1u, r = t.nodes[i].w1 >> 2;
    Box lb = box, rb = box;  // Aabb::split, geometry/src/aabb.rs:85-98
    lb.hi[axis] = dist;
    rb.lo[axis] = dist;
    const float lc = node_cost(t, l, c, scene_area, lb);
    const float rc = node_cost(t, r, c, scene_area, rb);
    const float cost = c.traverse_cost + split_cost + lc + rc;
    const float factor = (leaf_empty(t, l) || leaf_empty(t, r)) ? c.empty_factor : 1.0f;
    return factor * cost;
}

int check_tree(const float* triangles, uint32_t n_triangles, const ptx_kdnode* nodes, uint32_t n_nodes,
               const uint32_t* leaves, uint32_t n_leaves, const char* who, bool need_triangles = true) {
    if (!nodes || n_nodes == 0 || (need_triangles && n_triangles && !triangles) || (n_leaves && !leaves))
        return fail(PTX_E_ARG, std::string(who) + ": null argument");
    const std::string why = validate_flat_tree(nodes, n_nodes, leaves, n_leaves, n_triangles);
    if (!why.empty()) return fail(PTX_E_ARG, std::string(who) + ": invalid flattened tree: " + why);
    return PTX_OK;
}

int give(const std::string& s, char** text_out, size_t* len_out, const char* who) {
    char* mem = (char*)malloc(s.size() + 1);
    if (!mem) return fail(PTX_E_OOM, std::string(who) + ": out of memory");
    memcpy(mem, s.data(), s.size());
    mem[s.size()] = 0;
    *text_out = mem;
    if (len_out) *len_out = s.size();
    return PTX_OK;
}

}  // namespace
}  // namespace ptx

using namespace ptx;

extern "C" {

int ptx_kdtree_write_json(const float* triangles, uint32_t n_triangles, const ptx_kdnode* nodes, uint32_t n_nodes,
                          const uint32_t* leaf_indices, uint32_t n_leaf_indices, char** text_out, size_t* len_out) {
    if (!text_out) return fail(PTX_E_ARG, "ptx_kdtree_write_json: null argument");
    int rc = check_tree(triangles, n_triangles, nodes, n_nodes, leaf_indices, n_leaf_indices, "ptx_kdtree_write_json");
    if (rc) return rc;
    try {
        std::string out = "{\"triangles\": [";
        for (uint32_t i = 0; i < n_triangles; ++i) {
            float v[9];
            memcpy(v, triangles + 9 * (size_t)i, sizeof v);
            // AxiallyAlignedTriangle::as_arrays (axial_triangle.rs:43-49): one plane distance, v0's
            const int flat = aligned_axis(v);
            if (flat < 3) v[3 + flat] = v[6 + flat] = v[flat];
            if (i) out += ", ";
            out += '[';
            for (int k = 0; k < 3; ++k) {
                if (k) out += ", ";
                out += '[';
                for (int c = 0; c < 3; ++c) {
                    if (c) out += ", ";
                    put_debug(out, v[3 * k + c]);
                }
                out += ']';
            }
            out += ']';
        }
        out += "], \"root\": ";
        node_json(out, Flat{nodes, leaf_indices}, 0);
        out += "}\n";
        return give(out, text_out, len_out, "ptx_kdtree_write_json");
    } catch (std::bad_alloc&) {
        return fail(PTX_E_OOM, "ptx_kdtree_write_json: out of memory");
    }
}

int ptx_kdtree_write_pretty(const ptx_kdnode* nodes, uint32_t n_nodes, const uint32_t* leaf_indices,
                            uint32_t n_leaf_indices, char** text_out, size_t* len_out) {
    if (!text_out) return fail(PTX_E_ARG, "ptx_kdtree_write_pretty: null argument");
    int rc = check_tree(nullptr, 0xFFFFFFFFu, nodes, n_nodes, leaf_indices, n_leaf_indices, "ptx_kdtree_write_pretty", false);
    if (rc) return rc;
    try {
        const Flat t{nodes, leaf_indices};
        std::string out;
        std::vector<std::pair<uint32_t, uint32_t>> stack{{0u, 0u}};  // (depth, node): right subtree first, as the reference
        while (!stack.empty()) {
            const auto e = stack.back();
            stack.pop_back();
            out.append(2 * (size_t)e.first, ' ');
            if (t.leaf(e.second)) {
                out += "Leaf ";
                put_indices(out, leaf_indices + nodes[e.second].w0, nodes[e.second].w1 >> 2);
            } else {
                stack.push_back({e.first + 1u, e.second + 1u});
                stack.push_back({e.first + 1u, nodes[e.second].w1 >> 2});
                float dist;
                memcpy(&dist, &nodes[e.second].w0, 4);
                out += "Split ";
                out += AXIS_NAME[nodes[e.second].w1 & 3u];
                out += ' ';
                put_display(out, dist);
            }
            out += '\n';
        }
        return give(out, text_out, len_out, "ptx_kdtree_write_pretty");
    } catch (std::bad_alloc&) {
        return fail(PTX_E_OOM, "ptx_kdtree_write_pretty: out of memory");
    }
}

int ptx_kdtree_read_json(const char* text, size_t len, ptx_kdtree** tree_out, float** triangles_out,
                         uint32_t* n_triangles_out) {
    if (!text || !tree_out || !triangles_out || !n_triangles_out) return fail(PTX_E_ARG, "ptx_kdtree_read_json: null argument");
    try {
        Reader r{text, text + len, {}};
        std::vector<float> tris;
        std::vector<ptx_kdnode> nodes;
        std::vector<uint32_t> leaves;
        bool ok = r.expect('{') && r.key("triangles") && r.expect('[');
        if (ok && !r.eat(']')) {
            do {
                ok = r.expect('[');
                for (int k = 0; ok && k < 3; ++k) {
                    ok = (k == 0 || r.expect(',')) && r.expect('[');
                    for (int c = 0; ok && c < 3; ++c) {
                        float v = 0;
                        ok = (c == 0 || r.expect(',')) && r.number(v);
                        tris.push_back(v);
                    }
                    ok = ok && r.expect(']');
                }
                ok = ok && r.expect(']');
            } while (ok && r.eat(','));
            ok = ok && r.expect(']');
        }
        ok = ok && r.expect(',') && r.key("root") && read_node(r, nodes, leaves, 1) && r.expect('}');
        if (ok) {
            r.ws();
            if (r.p != r.end) {
                ok = false;
                r.err = "trailing characters after the tree";
            }
        }
        if (!ok) return fail(PTX_E_IO, "ptx_kdtree_read_json: " + (r.err.empty() ? std::string("malformed input") : r.err));
        const uint32_t n_tris = (uint32_t)(tris.size() / 9);
        const std::string why = validate_flat_tree(nodes.data(), (uint32_t)nodes.size(), leaves.data(), (uint32_t)leaves.size(), n_tris);
        if (!why.empty()) return fail(PTX_E_ARG, "ptx_kdtree_read_json: invalid tree: " + why);
        float* mem = (float*)malloc(std::max<size_t>(tris.size(), 1) * sizeof(float));
        if (!mem) return fail(PTX_E_OOM, "ptx_kdtree_read_json: out of memory");
        memcpy(mem, tris.data(), tris.size() * sizeof(float));
        ptx_kdtree* t = kdtree_from_arrays(std::move(nodes), std::move(leaves));
        *tree_out = t;
        *triangles_out = mem;
        *n_triangles_out = n_tris;
        return PTX_OK;
    } catch (std::bad_alloc&) {
        return fail(PTX_E_OOM, "ptx_kdtree_read_json: out of memory");
    }
}

int ptx_kdtree_stats(const float* triangles, uint32_t n_triangles, const ptx_kdnode* nodes, uint32_t n_nodes,
                     const uint32_t* leaf_indices, uint32_t n_leaf_indices, const ptx_sah_cost* sah, ptx_tree_stats* out) {
    if (!out) return fail(PTX_E_ARG, "ptx_kdtree_stats: null argument");
    int rc = check_tree(triangles, n_triangles, nodes, n_nodes, leaf_indices, n_leaf_indices, "ptx_kdtree_stats");
    if (rc) return rc;
    try {
        const ptx_sah_cost cost = sah ? *sah : ptx_sah_cost{1.0f, 1.5f, 0.8f};
        const Flat t{nodes, leaf_indices};
        memset(out, 0, sizeof *out);
        // geometries_bounding_box over AnyTriangle min/max (bound.rs:7-19, triangle.rs:49-56, axial_triangle.rs:15-24)
        Box box{f3(0, 0, 0), f3(0, 0, 0)};
        for (uint32_t i = 0; i < n_triangles; ++i) {
            const float* v = triangles + 9 * (size_t)i;
            F3 a = f3(v[0], v[1], v[2]), b = f3(v[3], v[4], v[5]), c = f3(v[6], v[7], v[8]);
            F3 lo = lo3(a, lo3(b, c)), hi = hi3(a, hi3(b, c));
            const int flat = aligned_axis(v);
            if (flat < 3) lo[flat] = hi[flat] = v[flat];
            if (i == 0) box = {lo, hi};
            else box = {lo3(box.lo, lo), hi3(box.hi, hi)};
        }
        out->sah_cost = node_cost(t, 0, cost, box.area(), box);
        std::vector<uint64_t> depths, sizes;
        std::vector<std::pair<uint32_t, uint32_t>> stack{{1u, 0u}};  // KdNodeIter: the root has depth 1 (lib.rs:156-161)
        while (!stack.empty()) {
            const auto e = stack.back();
            stack.pop_back();
            ++out->node_count;
            if (t.leaf(e.second)) {
                depths.push_back(e.first);
                sizes.push_back(nodes[e.second].w1 >> 2);
            } else {
                stack.push_back({e.first + 1u, e.second + 1u});
                stack.push_back({e.first + 1u, nodes[e.second].w1 >> 2});
            }
        }
        out->geometries = n_triangles;
        out->leaf_count = depths.size();
        const Stat d = compute(depths), g = compute(sizes);
        out->leaf_depth_min = d.min, out->leaf_depth_max = d.max, out->leaf_depth_total = d.total;
        out->leaf_depth_mean = d.mean, out->leaf_depth_median = d.median;
        out->leaf_geometries_min = g.min, out->leaf_geometries_max = g.max, out->leaf_geometries_total = g.total;
        out->leaf_geometries_mean = g.mean, out->leaf_geometries_median = g.median;
        return PTX_OK;
    } catch (std::bad_alloc&) {
        return fail(PTX_E_OOM, "ptx_kdtree_stats: out of memory");
    }
}

}  // extern "C"
