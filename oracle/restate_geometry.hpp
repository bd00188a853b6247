// oracle/restate_geometry.hpp — TEST INFRASTRUCTURE ONLY (the checker, never the product).
//
// CPU restatement of the reference's `geometry` crate for the render hot path.  Every function cites
// the reference file:line it follows.  Third-party arithmetic: glam 0.32.1 scalar Vec3/Vec2
// (Cargo.lock:261-267, not vendored) — dot = (x*x' + y*y') + z*z', cross as written below,
// normalize = v * (1/sqrt(dot(v,v))); Rust never contracts to FMA, so this file must be built with
// -ffp-contract=off.  Pinned by the reference's exact-f32 known-answer tests (tests/test_oracle_kat.py).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <optional>
#include <vector>

namespace orc {

struct Vec2 {
    float x, y;
};
inline Vec2 operator+(Vec2 a, Vec2 b) { return {a.x + b.x, a.y + b.y}; }
inline Vec2 operator-(Vec2 a, Vec2 b) { return {a.x - b.x, a.y - b.y}; }
inline Vec2 operator*(float s, Vec2 a) { return {s * a.x, s * a.y}; }
inline Vec2 operator*(Vec2 a, float s) { return {a.x * s, a.y * s}; }
inline Vec2 operator/(Vec2 a, Vec2 b) { return {a.x / b.x, a.y / b.y}; }

struct Vec3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
inline Vec3 operator*(Vec3 a, Vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline Vec3 operator*(float s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator/(Vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline Vec3 operator-(float s, Vec3 a) { return {s - a.x, s - a.y, s - a.z}; }
inline bool operator==(Vec3 a, Vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline bool operator!=(Vec3 a, Vec3 b) { return !(a == b); }
// glam scalar Vec3::dot / cross / length_squared / normalize / min / max
inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
inline Vec3 cross(Vec3 a, Vec3 b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
inline float length_squared(Vec3 a) { return dot(a, a); }
inline float length(Vec3 a) { return std::sqrt(dot(a, a)); }
inline Vec3 normalize(Vec3 a) { return a * (1.0f / length(a)); }
inline Vec3 vmin(Vec3 a, Vec3 b) {
    return {a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z};
}
inline Vec3 vmax(Vec3 a, Vec3 b) {
    return {a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z};
}

enum Axis : int { AX = 0, AY = 1, AZ = 2 };

// geometry/src/axis.rs:19-33
inline Vec3 axis_add_to(int axis, Vec2 v, float s) {
    switch (axis) {
        case AX: return {s, v.x, v.y};
        case AY: return {v.x, s, v.y};
        default: return {v.x, v.y, s};
    }
}
inline Vec2 axis_remove_from(int axis, Vec3 v) {
    switch (axis) {
        case AX: return {v.y, v.z};
        case AY: return {v.x, v.z};
        default: return {v.x, v.y};
    }
}

// geometry/src/ray.rs:4-38
struct Ray {
    Vec3 origin, direction;
    static Ray between(Vec3 a, Vec3 b) { return {a, b - a}; }
    Vec3 param(float t) const { return origin + t * direction; }
    Ray reverse() const { return {origin + direction, -direction}; }
    Ray extended(float t) const { return {origin, t * direction}; }
};

// geometry/src/aap.rs:8-72
struct Aap {
    int axis;
    float distance;
    // aap.rs:44-51
    std::optional<float> intersect_ray(const Ray& ray) const {
        float denom = ray.direction[axis];
        if (denom == 0.0f) return std::nullopt;
        return (distance - ray.origin[axis]) / denom;
    }
    // aap.rs:53-71
    std::optional<Vec3> intersect_ray_point(const Ray& ray) const {
        auto t = intersect_ray(ray);
        if (!t) return std::nullopt;
        float tt = *t;
        switch (axis) {
            case AX:
                return Vec3{distance, ray.origin.y + tt * ray.direction.y,
                            ray.origin.z + tt * ray.direction.z};
            case AY:
                return Vec3{ray.origin.x + tt * ray.direction.x, distance,
                            ray.origin.z + tt * ray.direction.z};
            default:
                return Vec3{ray.origin.x + tt * ray.direction.x,
                            ray.origin.y + tt * ray.direction.y, distance};
        }
    }
};

// geometry/src/aabb.rs:6-120
struct Aabb {
    Vec3 min, max;
    Vec3 size() const { return max - min; }
    float surface_area() const {
        Vec3 s = size();
        return 2.0f * (s.x * s.y + s.x * s.z + s.y * s.z);
    }
    float volume() const {
        Vec3 s = size();
        return s.x * s.y * s.z;
    }
    // aabb.rs:85-98
    void split(const Aap& plane, Aabb& fst, Aabb& snd) const {
        Vec3 new_max = max;
        new_max.at(plane.axis) = plane.distance;
        Vec3 new_min = min;
        new_min.at(plane.axis) = plane.distance;
        fst = {min, new_max};
        snd = {new_min, max};
    }
    // aabb.rs:100-115
    Vec3 clamp(Vec3 p) const {
        auto c = [](float x, float a, float b) { return x < a ? a : (x > b ? b : x); };
        return {c(p.x, min.x, max.x), c(p.y, min.y, max.y), c(p.z, min.z, max.z)};
    }
};

struct TriangleIntersection {
    float t, u, v;
};

// geometry/src/triangle.rs:6-105
struct Triangle {
    Vec3 v0, v1, v2;
    Vec3 tmin() const { return vmin(v0, vmin(v1, v2)); }  // triangle.rs:49-51
    Vec3 tmax() const { return vmax(v0, vmax(v1, v2)); }  // triangle.rs:54-56
    // triangle.rs:80-105 (Möller–Trumbore)
    std::optional<TriangleIntersection> intersect_ray(const Ray& ray) const {
        Vec3 base1 = v1 - v0;
        Vec3 base2 = v2 - v0;
        Vec3 ray_cross_base2 = cross(ray.direction, base2);
        float det = dot(base1, ray_cross_base2);
        if (det == 0.0f) return std::nullopt;
        float inv_det = 1.0f / det;
        Vec3 s = ray.origin - v0;
        float u = inv_det * dot(s, ray_cross_base2);
        if (!(0.0f <= u && u <= 1.0f)) return std::nullopt;
        Vec3 s_cross_base1 = cross(s, base1);
        float v = inv_det * dot(ray.direction, s_cross_base1);
        if (v < 0.0f || (u + v) > 1.0f) return std::nullopt;
        float t = inv_det * dot(base2, s_cross_base1);
        return TriangleIntersection{t, u, v};
    }
};

// geometry/src/axial_triangle.rs:6-85
struct AxiallyAlignedTriangle {
    Aap plane;
    Vec2 v0, v1, v2;
    Vec3 tmin() const {  // axial_triangle.rs:15-18
        auto m2 = [](Vec2 a, Vec2 b) { return Vec2{a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y}; };
        Vec2 p = m2(v0, m2(v1, v2));
        return axis_add_to(plane.axis, p, plane.distance);
    }
    Vec3 tmax() const {  // axial_triangle.rs:21-24
        auto m2 = [](Vec2 a, Vec2 b) { return Vec2{a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y}; };
        Vec2 p = m2(v0, m2(v1, v2));
        return axis_add_to(plane.axis, p, plane.distance);
    }
    // axial_triangle.rs:51-73
    bool intersect_point(Vec2 point, float& u, float& v) const {
        Vec2 base1 = v1 - v0;
        Vec2 base2 = v2 - v0;
        Vec2 s = point - v0;
        float det = base1.x * base2.y - base2.x * base1.y;
        if (det == 0.0f) return false;
        float inv_det = 1.0f / det;
        u = inv_det * (s.x * base2.y - base2.x * s.y);
        if (!(0.0f <= u && u <= 1.0f)) return false;
        v = inv_det * (base1.x * s.y - s.x * base1.y);
        if (v < 0.0f || v + u > 1.0f) return false;
        return true;
    }
    // axial_triangle.rs:75-84
    std::optional<TriangleIntersection> intersect_ray(const Ray& ray) const {
        int axis = plane.axis;
        if (ray.direction[axis] == 0.0f) return std::nullopt;
        float t = (plane.distance - ray.origin[axis]) / ray.direction[axis];
        Vec3 point = ray.param(t);
        float u, v;
        if (!intersect_point(axis_remove_from(axis, point), u, v)) return std::nullopt;
        return TriangleIntersection{t, u, v};
    }
    Vec3 vertex3(Vec2 p) const { return axis_add_to(plane.axis, p, plane.distance); }
};

std::optional<Aabb> clip_triangle_aabb(const Vec3& v0, const Vec3& v1, const Vec3& v2,
                                       const Aabb& aabb);

// geometry/src/any_triangle.rs:4-16 + triangle.rs:59-77 + geometry.rs:144-174
struct AnyTriangle {
    bool axial;
    Triangle tri;
    AxiallyAlignedTriangle ax;

    static AnyTriangle from(const Triangle& t) {
        AnyTriangle r{};
        r.tri = t;
        r.axial = false;
        for (int axis = 0; axis < 3; ++axis) {  // X, then Y, then Z (triangle.rs:74-76)
            if (t.v0[axis] == t.v1[axis] && t.v0[axis] == t.v2[axis]) {
                r.axial = true;
                r.ax = {Aap{axis, t.v0[axis]}, axis_remove_from(axis, t.v0),
                        axis_remove_from(axis, t.v1), axis_remove_from(axis, t.v2)};
                break;
            }
        }
        return r;
    }
    Vec3 gmin() const { return axial ? ax.tmin() : tri.tmin(); }
    Vec3 gmax() const { return axial ? ax.tmax() : tri.tmax(); }
    std::optional<TriangleIntersection> intersect_ray(const Ray& ray) const {
        return axial ? ax.intersect_ray(ray) : tri.intersect_ray(ray);
    }
    // geometry.rs:86-88, 110-117
    std::optional<Aabb> clip_aabb(const Aabb& aabb) const {
        if (axial)
            return clip_triangle_aabb(ax.vertex3(ax.v0), ax.vertex3(ax.v1), ax.vertex3(ax.v2), aabb);
        return clip_triangle_aabb(tri.v0, tri.v1, tri.v2, aabb);
    }
};

// geometry/src/clip.rs:25-91 (Sutherland–Hodgman; the reference's ArrayVec<_,9> is a std::vector here)
inline std::vector<Vec3> clip_triangle_aabb_points(const Vec3& v0, const Vec3& v1, const Vec3& v2,
                                                   const Aabb& aabb) {
    struct ClipPlane {
        bool is_max;
        Aap plane;
    };
    const ClipPlane clip_planes[6] = {
        {false, {AX, aabb.min.x}}, {false, {AY, aabb.min.y}}, {false, {AZ, aabb.min.z}},
        {true, {AX, aabb.max.x}},  {true, {AY, aabb.max.y}},  {true, {AZ, aabb.max.z}},
    };
    auto is_inside = [](const ClipPlane& cp, Vec3 p) {
        return cp.is_max ? p[cp.plane.axis] <= cp.plane.distance : p[cp.plane.axis] >= cp.plane.distance;
    };
    std::vector<Vec3> output{v1, v2, v0};
    auto push_consecutive = [](std::vector<Vec3>& o, Vec3 p) {
        if (o.empty() || o.back() != p) o.push_back(p);
    };
    for (const ClipPlane& cp : clip_planes) {
        if (output.empty()) return output;
        std::vector<Vec3> input;
        input.swap(output);
        for (size_t i = 0; i < input.size(); ++i) {
            Vec3 a = i == 0 ? input[input.size() - 1] : input[i - 1];
            Vec3 b = input[i];
            bool a_inside = is_inside(cp, a);
            bool b_inside = is_inside(cp, b);
            if (a_inside && b_inside) {
                push_consecutive(output, b);
            } else if (a_inside != b_inside) {
                Ray ray = Ray::between(a, b);
                auto p = cp.plane.intersect_ray_point(ray);
                // reference: unwrap_unchecked (clip.rs:82); a parallel edge cannot straddle the plane
                push_consecutive(output, p ? *p : b);
                if (b_inside) push_consecutive(output, b);
            }
        }
    }
    return output;
}

// geometry/src/clip.rs:6-20
inline std::optional<Aabb> clip_triangle_aabb(const Vec3& v0, const Vec3& v1, const Vec3& v2,
                                              const Aabb& aabb) {
    std::vector<Vec3> clipped = clip_triangle_aabb_points(v0, v1, v2, aabb);
    if (clipped.size() <= 2) return std::nullopt;
    Vec3 first = aabb.clamp(clipped[0]);
    Vec3 mn = first, mx = first;
    for (size_t i = 1; i < clipped.size(); ++i) {
        Vec3 p = aabb.clamp(clipped[i]);
        mn = vmin(mn, p);
        mx = vmax(mx, p);
    }
    return Aabb{mn, mx};
}

// geometry/src/bound.rs:7-19
inline Aabb geometries_bounding_box(const std::vector<AnyTriangle>& shape) {
    if (shape.empty()) return Aabb{{0, 0, 0}, {0, 0, 0}};
    Vec3 a = shape[0].gmin(), b = shape[0].gmax();
    for (size_t i = 1; i < shape.size(); ++i) {
        a = vmin(a, shape[i].gmin());
        b = vmax(b, shape[i].gmax());
    }
    return Aabb{a, b};
}

struct IndexedIntersection {
    uint32_t index;
    TriangleIntersection inner;
};

// geometry/src/geometry.rs:176-196 with IndexedIntersection::min (:47-53): first in list wins on t <=.
template <class It>
inline std::optional<IndexedIntersection> intersect_closest_geometry(
    const std::vector<AnyTriangle>& geometries, It begin, It end, const Ray& ray, float t_lo,
    float t_hi, uint64_t* tri_tests = nullptr) {
    std::optional<IndexedIntersection> best;
    for (It it = begin; it != end; ++it) {
        uint32_t index = *it;
        if (tri_tests) ++*tri_tests;
        auto isect = geometries[index].intersect_ray(ray);
        if (!isect) continue;
        if (!(t_lo <= isect->t && isect->t <= t_hi)) continue;
        if (!best) {
            best = IndexedIntersection{index, *isect};
        } else if (!(best->inner.t <= isect->t)) {
            best = IndexedIntersection{index, *isect};
        }
    }
    return best;
}

}  // namespace orc
