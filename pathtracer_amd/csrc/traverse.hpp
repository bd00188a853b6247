// traverse.hpp — nearest-hit / any-hit traversal of the flattened KD-tree and the two triangle tests.
//
// tri_test() is shared by the gfx950 kernels (ptx_device.hip: k_traverse, k_check) and the host; kd_traverse()
// is the scalar host query (ptx_kdtree_intersect_host) and the statement of the decisions k_traverse takes.  Semantics follow the reference:
//   KdNode::intersect               kdtree/src/lib.rs:73-132
//   intersect_closest_geometry      geometry/src/geometry.rs:176-196 (+ IndexedIntersection::min :47-53)
//   Triangle::intersect_ray         geometry/src/triangle.rs:80-105   (Möller–Trumbore, two-sided)
//   AxiallyAlignedTriangle          geometry/src/axial_triangle.rs:51-84
//   Aap::intersect_ray              geometry/src/aap.rs:44-51
// Plain IEEE f32, no contraction (-ffp-contract=off), correctly rounded division.
//
// The traversal stack stores (far child, t_split) only: the reference's third field, the segment end
// t2 saved at push time, always equals the t_split of the entry below it (or the ray's own t_end when
// the stack is empty), so it is re-read from there on pop instead of stored (8 bytes per level, not 12).
#pragma once
#include <stdint.h>

#include "../../include/ptx.h"

#if defined(__HIPCC__)
#define PTX_FN __host__ __device__ inline
#else
#define PTX_FN inline
#endif

namespace ptx {

struct alignas(16) TriRec {  // 48 bytes; see host_internal.hpp TriRecord for the field layout
    float f[11];
    uint32_t kind;
};

PTX_FN float pick(float x, float y, float z, uint32_t axis) { return axis == 0 ? x : (axis == 1 ? y : z); }

PTX_FN uint32_t f2u(float f) {
    union { float f; uint32_t u; } c;
    c.f = f;
    return c.u;
}
PTX_FN float u2f(uint32_t u) {
    union { float f; uint32_t u; } c;
    c.u = u;
    return c.f;
}

// Returns true and (t,u,v) when the ray meets the triangle (no range test here).
PTX_FN bool tri_test(const TriRec& r, float ox, float oy, float oz, float dx, float dy, float dz,
                     float& t, float& u, float& v) {
    if (r.kind == 3u) {
        const float v0x = r.f[0], v0y = r.f[1], v0z = r.f[2];
        const float e1x = r.f[3], e1y = r.f[4], e1z = r.f[5];
        const float e2x = r.f[6], e2y = r.f[7], e2z = r.f[8];
        // p = d x e2
        const float px = dy * e2z - e2y * dz;
        const float py = dz * e2x - e2z * dx;
        const float pz = dx * e2y - e2x * dy;
        const float det = (e1x * px + e1y * py) + e1z * pz;
        if (det == 0.0f) return false;
        const float inv = 1.0f / det;
        const float sx = ox - v0x, sy = oy - v0y, sz = oz - v0z;
        u = inv * ((sx * px + sy * py) + sz * pz);
        if (!(0.0f <= u && u <= 1.0f)) return false;
        // q = s x e1
        const float qx = sy * e1z - e1y * sz;
        const float qy = sz * e1x - e1z * sx;
        const float qz = sx * e1y - e1x * sy;
        v = inv * ((dx * qx + dy * qy) + dz * qz);
        if (v < 0.0f || (u + v) > 1.0f) return false;
        t = inv * ((e2x * qx + e2y * qy) + e2z * qz);
        return true;
    }
    const uint32_t axis = r.kind;
    const float da = pick(dx, dy, dz, axis);
    if (da == 0.0f) return false;
    t = (r.f[0] - pick(ox, oy, oz, axis)) / da;
    const float hx = ox + t * dx, hy = oy + t * dy, hz = oz + t * dz;
    const float p0 = axis == 0 ? hy : hx;  // Axis::remove_from (axis.rs:27-33)
    const float p1 = axis == 2 ? hy : hz;
    const float b1x = r.f[3], b1y = r.f[4], b2x = r.f[5], b2y = r.f[6];
    const float sx = p0 - r.f[1], sy = p1 - r.f[2];
    // det = b1x*b2y - b2x*b1y and inv = 1/det depend on the triangle alone: pack_triangles() evaluates
    // them once with the same IEEE operations and stores inv, or NaN for det == 0 (then u is NaN and the
    // range test below rejects, as the reference's `det == 0` early return does).
    const float inv = r.f[7];
    u = inv * (sx * b2y - b2x * sy);
    if (!(0.0f <= u && u <= 1.0f)) return false;
    v = inv * (b1x * sy - sx * b1y);
    if (v < 0.0f || (v + u) > 1.0f) return false;
    return true;
}

struct WorkCount {
    uint32_t inner, tris;
};

// ANY = true: stop at the first accepted triangle (occlusion query; same yes/no as `.is_some()` on the
// nearest-hit search, because both accept in the first leaf that holds a triangle inside its segment).
// `tris` is anything indexable by triangle number that yields a TriRec (an array, or a packer on demand).
template <bool ANY, bool COUNT, class Tris>
PTX_FN bool kd_traverse(const ptx_kdnode* __restrict__ nodes, const uint32_t* __restrict__ leaf_idx,
                        const Tris& tris, float ox, float oy, float oz, float dx,
                        float dy, float dz, float t_start, float t_end, ptx_hit& hit, WorkCount& wc) {
    uint32_t st_node[PTX_MAX_DEPTH];
    float st_t[PTX_MAX_DEPTH];
    int sp = 0;
    uint32_t node = 0;
    float t1 = t_start, t2 = t_end;
    for (;;) {
        const ptx_kdnode nd = nodes[node];
        const uint32_t kind = nd.w1 & 3u;
        if (kind != 3u) {
            if (COUNT) ++wc.inner;
            const float dist = u2f(nd.w0);
            const float da = pick(dx, dy, dz, kind);
            const float oa = pick(ox, oy, oz, kind);
            const uint32_t left = node + 1u, right = nd.w1 >> 2;
            if (da == 0.0f) {
                node = oa <= dist ? left : right;
            } else {
                const float t = (dist - oa) / da;
                const uint32_t near_n = da >= 0.0f ? left : right;
                const uint32_t far_n = da >= 0.0f ? right : left;
                if (t > t2) {
                    node = near_n;
                } else if (t < t1) {
                    node = far_n;
                } else {
                    st_node[sp] = far_n;
                    st_t[sp] = t;
                    ++sp;
                    node = near_n;
                    t2 = t;
                }
            }
        } else {
            const uint32_t count = nd.w1 >> 2, off = nd.w0;
            bool found = false;
            float bt = 0.0f, bu = 0.0f, bv = 0.0f;
            uint32_t bi = PTX_MISS;
            for (uint32_t i = 0; i < count; ++i) {
                const uint32_t idx = leaf_idx[off + i];
                if (COUNT) ++wc.tris;
                float t, u, v;
                if (!tri_test(tris[idx], ox, oy, oz, dx, dy, dz, t, u, v)) continue;
                if (!(t1 <= t && t <= t2)) continue;
                if (!found || !(bt <= t)) {  // keep the earlier triangle on t ties
                    found = true;
                    bt = t;
                    bu = u;
                    bv = v;
                    bi = idx;
                    if (ANY) break;
                }
            }
            if (found) {
                hit.index = bi;
                hit.t = bt;
                hit.u = bu;
                hit.v = bv;
                return true;
            }
            if (t2 == t_end || sp == 0) break;
            --sp;
            node = st_node[sp];
            t1 = st_t[sp];
            t2 = sp > 0 ? st_t[sp - 1] : t_end;
        }
    }
    hit.index = PTX_MISS;
    hit.t = 0.0f;
    hit.u = 0.0f;
    hit.v = 0.0f;
    return false;
}

}  // namespace ptx
