// host_geom.hpp — host-side geometry for the product library (no dependency on oracle/).
//
// The arithmetic follows the reference bit for bit because the KD-tree it produces decides which
// triangle wins at ties: geometry/src/{aabb,aap,clip,triangle,axial_triangle}.rs with glam's scalar
// Vec3 evaluation order; this translation unit is compiled with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace ptx {

struct F3 {
    float v[3];
    float operator[](int i) const { return v[i]; }
    float& operator[](int i) { return v[i]; }
};
inline F3 f3(float x, float y, float z) { return F3{{x, y, z}}; }
inline F3 sub(F3 a, F3 b) { return f3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline F3 add(F3 a, F3 b) { return f3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline F3 scale(F3 a, float s) { return f3(a[0] * s, a[1] * s, a[2] * s); }
inline float dot3(F3 a, F3 b) { return (a[0] * b[0]) + (a[1] * b[1]) + (a[2] * b[2]); }
inline F3 cross3(F3 a, F3 b) {
    return f3(a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]);
}
inline F3 unit(F3 a) { return scale(a, 1.0f / std::sqrt(dot3(a, a))); }
inline float lesser(float a, float b) { return a < b ? a : b; }    // glam min: `if a < b {a} else {b}`
inline float greater(float a, float b) { return a > b ? a : b; }
inline F3 lo3(F3 a, F3 b) { return f3(lesser(a[0], b[0]), lesser(a[1], b[1]), lesser(a[2], b[2])); }
inline F3 hi3(F3 a, F3 b) { return f3(greater(a[0], b[0]), greater(a[1], b[1]), greater(a[2], b[2])); }
inline bool same(F3 a, F3 b) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2]; }

struct Box {
    F3 lo, hi;
    float area() const {  // geometry/src/aabb.rs:55-58
        F3 s = sub(hi, lo);
        return 2.0f * (s[0] * s[1] + s[0] * s[2] + s[1] * s[2]);
    }
    float volume() const {  // aabb.rs:60-63
        F3 s = sub(hi, lo);
        return s[0] * s[1] * s[2];
    }
    F3 clamp(F3 p) const {  // aabb.rs:100-115
        F3 r;
        for (int a = 0; a < 3; ++a) r[a] = p[a] < lo[a] ? lo[a] : (p[a] > hi[a] ? hi[a] : p[a]);
        return r;
    }
};

// Axis on which all three vertices coincide, X before Y before Z (geometry/src/triangle.rs:59-77), or 3.
inline int aligned_axis(const float* t9) {
    for (int a = 0; a < 3; ++a)
        if (t9[a] == t9[3 + a] && t9[a] == t9[6 + a]) return a;
    return 3;
}

// Sutherland–Hodgman clip of a triangle against a box, then the clamped bounds of the polygon
// (geometry/src/clip.rs:6-91).  Returns false when fewer than 3 points survive.
inline bool clip_bounds(F3 v0, F3 v1, F3 v2, const Box& box, Box& out) {
    F3 poly_a[16], poly_b[16];
    F3* cur = poly_a;
    F3* nxt = poly_b;
    int n = 3;
    cur[0] = v1;
    cur[1] = v2;
    cur[2] = v0;
    for (int pl = 0; pl < 6; ++pl) {  // min x,y,z then max x,y,z (clip.rs:28-35)
        if (n == 0) return false;
        const int axis = pl % 3;
        const bool upper = pl >= 3;
        const float d = upper ? box.hi[axis] : box.lo[axis];
        int m = 0;
        auto emit = [&](F3 p) {
            if (m == 0 || !same(nxt[m - 1], p)) {
                if (m < 16) nxt[m++] = p;
            }
        };
        for (int i = 0; i < n; ++i) {
            F3 a = cur[i == 0 ? n - 1 : i - 1];
            F3 b = cur[i];
            bool ain = upper ? a[axis] <= d : a[axis] >= d;
            bool bin = upper ? b[axis] <= d : b[axis] >= d;
            if (ain && bin) {
                emit(b);
            } else if (ain != bin) {
                F3 dir = sub(b, a);  // Ray::between(a, b)
                float t = (d - a[axis]) / dir[axis];  // Aap::intersect_ray (aap.rs:44-51)
                F3 p;  // Aap::intersect_ray_point (aap.rs:53-71): the plane coordinate is exact
                for (int c = 0; c < 3; ++c) p[c] = c == axis ? d : a[c] + t * dir[c];
                emit(p);
                if (bin) emit(b);
            }
        }
        F3* tmp = cur;
        cur = nxt;
        nxt = tmp;
        n = m;
    }
    if (n <= 2) return false;
    F3 first = box.clamp(cur[0]);
    out.lo = first;
    out.hi = first;
    for (int i = 1; i < n; ++i) {
        F3 p = box.clamp(cur[i]);
        out.lo = lo3(out.lo, p);
        out.hi = hi3(out.hi, p);
    }
    return true;
}

}  // namespace ptx
