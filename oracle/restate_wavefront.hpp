// oracle/restate_wavefront.hpp — TEST INFRASTRUCTURE ONLY (the checker, never the product).
//
// CPU restatement of the reference's `wavefront` crate (OBJ + the reference's custom MTL dialect) and
// of tracing/src/properties.rs::from_wavefront.  The reference parses with `nom 8` (not vendored):
// `float` ≈ strtof on a decimal literal, `i32` = optional sign + digits, `space0/1` = spaces/tabs.
// Pinned by wavefront/src/obj.rs:172-303 and mtl.rs:163-250 through tests/test_oracle_kat.py.
#pragma once
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>

#include "restate_tracing.hpp"

namespace orc {

struct ObjPoint {
    int v, t, n;
};
struct ObjChunk {
    std::string material;
    std::vector<std::vector<ObjPoint>> faces;
};
struct Obj {
    std::string mtl_lib;
    std::vector<Vec3> vertices, normals;
    std::vector<Vec2> texcoords;
    std::vector<ObjChunk> chunks;
};
struct MtlMaterial {  // mtl.rs:8-36
    std::string name, diffuse_map;
    Vec3 diffuse_reflection{0.7f, 0.7f, 0.7f};
    Vec3 specular_reflection{0, 0, 0};
    float transparency = 0.0f, reflection_0_degrees = 0.0f, reflection_90_degrees = 0.0f;
    float index_of_refraction = 1.0f, metalness = 0.0f;
};
struct MtlLight {
    Vec3 position{0, 0, 0}, color{0, 0, 0};
    float radius = 0, intensity = 0;
};
struct MtlCamera {
    Vec3 position{0, 0, 0}, target{0, 0, 0}, up{0, 0, 0};
    float fov = 0;
};
struct Mtl {
    std::vector<MtlMaterial> materials;
    std::vector<MtlLight> lights;
    std::vector<MtlCamera> cameras;
};

namespace parse {
inline std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) ++a;
    while (b > a && isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}
inline void space0(const char*& p) {
    while (*p == ' ' || *p == '\t') ++p;
}
inline bool tag(const char*& p, const char* name, bool no_case) {
    const char* q = p;
    for (const char* n = name; *n; ++n, ++q) {
        char a = *q, b = *n;
        if (no_case) {
            a = (char)tolower((unsigned char)a);
            b = (char)tolower((unsigned char)b);
        }
        if (a != b) return false;
    }
    p = q;
    return true;
}
inline bool number(const char*& p, float& out) {
    if (*p == ' ' || *p == '\t' || *p == 0) return false;
    char* end = nullptr;
    out = strtof(p, &end);
    if (end == p) return false;
    p = end;
    return true;
}
inline bool vec3(const char*& p, Vec3& o) {
    if (!number(p, o.x)) return false;
    space0(p);
    if (!number(p, o.y)) return false;
    space0(p);
    return number(p, o.z);
}
inline bool vec2(const char*& p, Vec2& o) {
    if (!number(p, o.x)) return false;
    space0(p);
    return number(p, o.y);
}
inline bool integer(const char*& p, int& out) {
    const char* q = p;
    if (*q == '+' || *q == '-') ++q;
    if (!isdigit((unsigned char)*q)) return false;
    char* end = nullptr;
    out = (int)strtol(p, &end, 10);
    p = end;
    return true;
}
// obj.rs:102-109
inline bool point(const char*& p, ObjPoint& o) {
    const char* q = p;
    if (!integer(q, o.v)) return false;
    if (*q != '/') return false;
    ++q;
    if (!integer(q, o.t)) o.t = 0;
    if (*q != '/') return false;
    ++q;
    if (!integer(q, o.n)) o.n = 0;
    p = q;
    return true;
}
}  // namespace parse

// obj.rs:117-166
inline Obj parse_obj(std::istream& in) {
    using namespace parse;
    Obj obj;
    std::string line;
    while (std::getline(in, line)) {
        std::string t = trim(line);
        if (t.empty() || t[0] == '#') continue;
        const char* p = t.c_str();
        const char* q;
        Vec3 v3;
        Vec2 v2;
        if (q = p, tag(q, "mtllib", true)) {
            space0(q);
            obj.mtl_lib = q;
        } else if (q = p, tag(q, "usemtl", true)) {
            space0(q);
            obj.chunks.push_back({q, {}});
        } else if (q = p, tag(q, "v", true) && (space0(q), vec3(q, v3))) {
            obj.vertices.push_back(v3);
        } else if (q = p, tag(q, "vn", true) && (space0(q), vec3(q, v3))) {
            obj.normals.push_back(v3);
        } else if (q = p, tag(q, "vt", true) && (space0(q), vec2(q, v2))) {
            obj.texcoords.push_back(v2);
        } else if (q = p, tag(q, "f", true)) {
            space0(q);
            std::vector<ObjPoint> pts;  // separated_list0(space1, point), obj.rs:111-115
            ObjPoint pt;
            if (point(q, pt)) {
                pts.push_back(pt);
                for (;;) {
                    const char* r = q;
                    if (!(*r == ' ' || *r == '\t')) break;
                    space0(r);
                    if (!point(r, pt)) break;
                    pts.push_back(pt);
                    q = r;
                }
            }
            if (obj.chunks.empty()) throw std::runtime_error("face before usemtl");
            obj.chunks.back().faces.push_back(pts);
        } else if ((q = p, tag(q, "g", true)) || (q = p, tag(q, "o", true)) ||
                   (q = p, tag(q, "s", true))) {
            // not supported (obj.rs:147-152)
        } else {
            throw std::runtime_error("Unexpected line: \"" + line + "\"");
        }
    }
    return obj;
}

// mtl.rs:80-157
inline Mtl parse_mtl(std::istream& in) {
    using namespace parse;
    Mtl mtl;
    std::string line;
    auto need = [&](bool ok) {
        if (!ok) throw std::runtime_error("property before its newlight/newcamera/newmtl");
    };
    while (std::getline(in, line)) {
        std::string t = trim(line);
        if (t.empty() || t[0] == '#') continue;
        const char* p = t.c_str();
        const char* q;
        Vec3 v3;
        float f;
        auto T3 = [&](const char* name) { return q = p, tag(q, name, false) && (space0(q), vec3(q, v3)); };
        auto T1 = [&](const char* name) { return q = p, tag(q, name, false) && (space0(q), number(q, f)); };
        auto TR = [&](const char* name) { return q = p, tag(q, name, false) && (space0(q), true); };
        if (TR("newlight")) {
            mtl.lights.emplace_back();
        } else if (T3("lightposition")) {
            need(!mtl.lights.empty());
            mtl.lights.back().position = v3;
        } else if (T3("lightcolor")) {
            need(!mtl.lights.empty());
            mtl.lights.back().color = v3;
        } else if (T1("lightradius")) {
            need(!mtl.lights.empty());
            mtl.lights.back().radius = f;
        } else if (T1("lightintensity")) {
            need(!mtl.lights.empty());
            mtl.lights.back().intensity = f;
        } else if (TR("newcamera")) {
            mtl.cameras.emplace_back();
        } else if (T3("cameraposition")) {
            need(!mtl.cameras.empty());
            mtl.cameras.back().position = v3;
        } else if (T3("cameratarget")) {
            need(!mtl.cameras.empty());
            mtl.cameras.back().target = v3;
        } else if (T3("cameraup")) {
            need(!mtl.cameras.empty());
            mtl.cameras.back().up = v3;
        } else if (T1("camerafov")) {
            need(!mtl.cameras.empty());
            mtl.cameras.back().fov = f;
        } else if (TR("newmtl")) {
            MtlMaterial m;
            m.name = q;
            mtl.materials.push_back(m);
        } else if (T1("illum")) {
        } else if (T1("Ka")) {
        } else if (T3("Kd")) {
            need(!mtl.materials.empty());
            mtl.materials.back().diffuse_reflection = v3;
        } else if (TR("map_Kd")) {
            need(!mtl.materials.empty());
            mtl.materials.back().diffuse_map = q;
        } else if (T3("Ks")) {
            need(!mtl.materials.empty());
            mtl.materials.back().specular_reflection = v3;
        } else if (T1("Ns")) {
        } else if (T1("Ke")) {
        } else if (T1("reflat0deg")) {
            need(!mtl.materials.empty());
            mtl.materials.back().reflection_0_degrees = f;
        } else if (T1("reflat90deg")) {
            need(!mtl.materials.empty());
            mtl.materials.back().reflection_90_degrees = f;
        } else if (T1("Ni")) {
            need(!mtl.materials.empty());
            mtl.materials.back().index_of_refraction = f;
        } else if (T1("d")) {
            need(!mtl.materials.empty());
            mtl.materials.back().transparency = 1.0f - f;
        } else if (T1("Tr")) {
            need(!mtl.materials.empty());
            mtl.materials.back().transparency = f;
        } else if (T1("Pm")) {
            need(!mtl.materials.empty());
            mtl.materials.back().metalness = f;
        } else if (T1("specularroughness")) {
        } else {
            throw std::runtime_error("Unexpected line: \"" + line + "\"");
        }
    }
    return mtl;
}

// obj.rs:62-68
template <class T>
inline T index_wavefront_vec(const std::vector<T>& v, int i) {
    if (i == 0) return T{};
    if (i < 0) return v.at((size_t)((int)v.size() + i));
    return v.at((size_t)(i - 1));
}

// tracing/src/properties.rs:49-90
inline void from_wavefront(const Obj& obj, const Mtl& mtl, std::vector<AnyTriangle>& triangles,
                           std::vector<TriangleProperties>& properties) {
    for (const ObjChunk& chunk : obj.chunks) {
        for (const auto& face : chunk.faces) {
            if (face.size() != 3) throw std::runtime_error("Only triangular faces supported");
            Triangle t{index_wavefront_vec(obj.vertices, face[0].v),
                       index_wavefront_vec(obj.vertices, face[1].v),
                       index_wavefront_vec(obj.vertices, face[2].v)};
            triangles.push_back(AnyTriangle::from(t));
            TriangleProperties p;
            p.n0 = index_wavefront_vec(obj.normals, face[0].n);
            p.n1 = index_wavefront_vec(obj.normals, face[1].n);
            p.n2 = index_wavefront_vec(obj.normals, face[2].n);
            p.uv0 = index_wavefront_vec(obj.texcoords, face[0].t);
            p.uv1 = index_wavefront_vec(obj.texcoords, face[1].t);
            p.uv2 = index_wavefront_vec(obj.texcoords, face[2].t);
            size_t mi = 0;
            for (; mi < mtl.materials.size(); ++mi)
                if (mtl.materials[mi].name == chunk.material) break;
            if (mi == mtl.materials.size()) throw std::runtime_error("unknown material " + chunk.material);
            p.material = (uint32_t)mi;
            properties.push_back(p);
        }
    }
}

}  // namespace orc
