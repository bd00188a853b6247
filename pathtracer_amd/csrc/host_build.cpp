// host_build.cpp — SAH KD-tree builder that emits the flattened 8-byte-node layout directly.
//
// Stays on the CPU (north star: "the kdtree SAH builder ... run on the CPU").  Same decisions as the
// reference builder so that the tree — and through it the winning triangle at ties — is identical:
//   kdtree/src/build.rs:11-41   recursion, leaf at depth >= 25 or empty cell
//   kdtree/src/sah.rs:59-119    cost model (empty bonus only when the empty side has volume > 0.01)
//   kdtree/src/sah.rs:139-180   plane sweep, End → Planar → Start groups per distinct distance
//   kdtree/src/sah.rs:189-312   side classification, clip-and-re-event of straddlers, merge
//   kdtree/src/event.rs:46-96   event order = (f32 total order of distance, kind)
// Output layout: include/ptx.h `ptx_kdnode`.
#include <algorithm>
#include <exception>
#include <thread>
#include <cstdlib>
#include <vector>

#include "../../include/ptx.h"
#include "host_geom.hpp"
#include "host_internal.hpp"

namespace ptx {
namespace {

enum : uint8_t { EV_END = 0, EV_PLANAR = 1, EV_START = 2 };
enum : uint8_t { SIDE_BOTH = 0, SIDE_LEFT = 1, SIDE_RIGHT = 2 };

struct Ev {
    float d;
    uint32_t id;
    uint32_t kind;
};

inline int32_t order_bits(float f) {  // f32::total_cmp key
    int32_t b;
    std::memcpy(&b, &f, 4);
    return b ^ (int32_t)((uint32_t)(b >> 31) >> 1);
}
inline bool ev_before(const Ev& a, const Ev& b) {
    int32_t ka = order_bits(a.d), kb = order_bits(b.d);
    return ka != kb ? ka < kb : a.kind < b.kind;
}

struct Cell {
    Box box;
    std::vector<uint32_t> ids;
    std::vector<Ev> ev[3];
};

struct Plane {
    bool valid = false;
    int axis = 0;
    float d = 0.0f;
    bool planar_right = false;
    float cost = 0.0f;
};

struct Builder {
    const float* tris;
    uint32_t n;
    ptx_sah_cost sah;
    std::vector<uint8_t> side;
    std::vector<ptx_kdnode> nodes;
    std::vector<uint32_t> leaves;

    F3 vert(uint32_t t, int k) const { return f3(tris[9 * t + 3 * k], tris[9 * t + 3 * k + 1], tris[9 * t + 3 * k + 2]); }

    static void push_events(std::vector<Ev>* ev, uint32_t id, F3 lo, F3 hi) {
        for (int a = 0; a < 3; ++a) {
            if (lo[a] == hi[a]) {
                ev[a].push_back({lo[a], id, EV_PLANAR});
            } else {
                ev[a].push_back({lo[a], id, EV_START});
                ev[a].push_back({hi[a], id, EV_END});
            }
        }
    }

    float cost_of(float v0, float v1, float p0, float p1, size_t c0, size_t c1) const {
        float bonus = ((c0 == 0 && v0 > 0.01f) || (c1 == 0 && v1 > 0.01f)) ? sah.empty_factor : 1.0f;
        float isect = sah.intersect_cost * (p0 * (float)c0 + p1 * (float)c1);
        return bonus * (sah.traverse_cost + isect);
    }

    // Candidate plane with `np` triangles lying in it, nl strictly left, nr strictly right.
    void consider(Plane& best, const Box& box, int axis, float d, size_t nl, size_t np, size_t nr) const {
        size_t total = nl + np + nr;
        if (box.volume() == 0.0f || total == 0) return;
        Box l = box, r = box;
        l.hi[axis] = d;
        r.lo[axis] = d;
        float sa = box.area();
        float v0 = l.volume(), v1 = r.volume();
        float p0 = l.area() / sa, p1 = r.area() / sa;
        float leaf = sah.intersect_cost * (float)total;
        Plane cand;
        cand.axis = axis;
        cand.d = d;
        if (v0 > 0.0f && v1 > 0.0f) {
            float cl = cost_of(v0, v1, p0, p1, nl + np, nr);
            float cr = cost_of(v0, v1, p0, p1, nl, nr + np);
            if (!(cl < leaf || cr < leaf)) return;
            cand.planar_right = !(cl <= cr);
            cand.cost = cand.planar_right ? cr : cl;
        } else if (v0 == 0.0f && nl + np > 0) {
            float c = cost_of(v0, v1, p0, p1, nl + np, nr);
            if (!(c < leaf)) return;
            cand.planar_right = false;
            cand.cost = c;
        } else if (v1 == 0.0f && np + nr > 0) {
            float c = cost_of(v0, v1, p0, p1, nl, np + nr);
            if (!(c < leaf)) return;
            cand.planar_right = true;
            cand.cost = c;
        } else {
            return;
        }
        cand.valid = true;
        // earlier candidate wins ties (sah.rs:38-49 `<=`)
        if (!best.valid || !(best.cost <= cand.cost)) best = cand;
    }

    Plane sweep_axis(const Cell& c, int axis) const {
        const std::vector<Ev>& ev = c.ev[axis];
        Plane best;
        size_t nl = 0, nr = c.ids.size(), i = 0;
        while (i < ev.size()) {
            const float d = ev[i].d;
            size_t cnt[3] = {0, 0, 0};
            for (uint32_t kind = EV_END; kind <= EV_START; ++kind)
                while (i < ev.size() && ev[i].d == d && ev[i].kind == kind) {
                    ++cnt[kind];
                    ++i;
                }
            if (cnt[0] + cnt[1] + cnt[2] == 0) {  // NaN coordinate: never equal to itself
                ++i;
                continue;
            }
            nr -= cnt[EV_PLANAR];
            nr -= cnt[EV_END];
            consider(best, c.box, axis, d, nl, cnt[EV_PLANAR], nr);
            nl += cnt[EV_START];
            nl += cnt[EV_PLANAR];
        }
        return best;
    }

    Plane choose(const Cell& c) const {
        Plane x = sweep_axis(c, 0), y = sweep_axis(c, 1), z = sweep_axis(c, 2);
        // zip_min(x, zip_min(y, z)): ties go to X, then Y, then Z
        Plane yz = !y.valid ? z : (!z.valid ? y : (y.cost <= z.cost ? y : z));
        return !x.valid ? yz : (!yz.valid ? x : (x.cost <= yz.cost ? x : yz));
    }

    void split(const Cell& c, const Plane& p, Cell& L, Cell& R) {
        for (uint32_t id : c.ids) side[id] = SIDE_BOTH;
        for (const Ev& e : c.ev[p.axis]) {
            uint8_t& s = side[e.id];
            if (e.kind == EV_END && e.d <= p.d) {
                s = SIDE_LEFT;
            } else if (e.kind == EV_START && e.d >= p.d) {
                s = SIDE_RIGHT;
            } else if (e.kind == EV_PLANAR) {
                if (e.d < p.d || (e.d == p.d && !p.planar_right)) {
                    s = SIDE_LEFT;
                } else if (e.d > p.d || (e.d == p.d && p.planar_right)) {
                    s = SIDE_RIGHT;
                }
            }
        }
        L.box = c.box;
        R.box = c.box;
        L.box.hi[p.axis] = p.d;
        R.box.lo[p.axis] = p.d;

        std::vector<Ev> keepL[3], keepR[3], freshL[3], freshR[3];
        for (int a = 0; a < 3; ++a)
            for (const Ev& e : c.ev[a]) {
                if (side[e.id] == SIDE_LEFT) keepL[a].push_back(e);
                else if (side[e.id] == SIDE_RIGHT) keepR[a].push_back(e);
            }
        for (uint32_t id : c.ids) {
            if (side[id] == SIDE_LEFT) {
                L.ids.push_back(id);
            } else if (side[id] == SIDE_RIGHT) {
                R.ids.push_back(id);
            } else {
                Box b;
                if (clip_bounds(vert(id, 0), vert(id, 1), vert(id, 2), L.box, b)) {
                    push_events(freshL, id, b.lo, b.hi);
                    L.ids.push_back(id);
                }
                if (clip_bounds(vert(id, 0), vert(id, 1), vert(id, 2), R.box, b)) {
                    push_events(freshR, id, b.lo, b.hi);
                    R.ids.push_back(id);
                }
            }
        }
        for (int a = 0; a < 3; ++a) {
            std::sort(freshL[a].begin(), freshL[a].end(), ev_before);
            std::sort(freshR[a].begin(), freshR[a].end(), ev_before);
            L.ev[a].resize(keepL[a].size() + freshL[a].size());
            std::merge(keepL[a].begin(), keepL[a].end(), freshL[a].begin(), freshL[a].end(), L.ev[a].begin(), ev_before);
            R.ev[a].resize(keepR[a].size() + freshR[a].size());
            std::merge(keepR[a].begin(), keepR[a].end(), freshR[a].begin(), freshR[a].end(), R.ev[a].begin(), ev_before);
        }
    }

    void emit_leaf(const std::vector<uint32_t>& ids) {
        nodes.push_back({(uint32_t)leaves.size(), ((uint32_t)ids.size() << 2) | 3u});
        leaves.insert(leaves.end(), ids.begin(), ids.end());
    }

    void grow(Cell& c, uint32_t depth) {
        if (depth >= PTX_MAX_DEPTH || c.ids.empty()) {
            emit_leaf(c.ids);
            return;
        }
        Plane p = choose(c);
        if (!p.valid) {
            emit_leaf(c.ids);
            return;
        }
        Cell L, R;
        split(c, p, L, R);
        c = Cell();  // release the parent's lists before recursing
        size_t me = nodes.size();
        uint32_t bits;
        std::memcpy(&bits, &p.d, 4);
        nodes.push_back({bits, (uint32_t)p.axis});
        grow(L, depth + 1);
        nodes[me].w1 = ((uint32_t)nodes.size() << 2) | (uint32_t)p.axis;
        grow(R, depth + 1);
    }

    // The same tree, built by several threads: the two subtrees of a node share nothing but the (read-only)
    // triangles, so above `par_levels` the left one is grown by another thread into arrays of its own and
    // spliced in afterwards (child and leaf offsets shifted).  Decisions, node order and leaf order are those
    // of grow(): the result is identical bit for bit.
    void append(const Builder& sub) {
        const uint32_t node_at = (uint32_t)nodes.size(), leaf_at = (uint32_t)leaves.size();
        for (const ptx_kdnode& nd : sub.nodes) {
            if ((nd.w1 & 3u) == 3u) nodes.push_back({nd.w0 + leaf_at, nd.w1});
            else nodes.push_back({nd.w0, (((nd.w1 >> 2) + node_at) << 2) | (nd.w1 & 3u)});
        }
        leaves.insert(leaves.end(), sub.leaves.begin(), sub.leaves.end());
    }

    Builder child() const {
        Builder b;
        b.tris = tris;
        b.n = n;
        b.sah = sah;
        b.side.assign(n, SIDE_BOTH);
        return b;
    }

    void grow_parallel(Cell& c, uint32_t depth, int par_levels) {
        if (par_levels <= 0 || c.ids.size() < 512 || depth >= PTX_MAX_DEPTH) {
            grow(c, depth);
            return;
        }
        Plane p = choose(c);
        if (!p.valid) {
            emit_leaf(c.ids);
            return;
        }
        Cell L, R;
        split(c, p, L, R);
        c = Cell();
        Builder left = child(), right = child();
        // an exception on either side (bad_alloc on a huge scene) must reach the C ABI as an error code: the
        // worker hands its own over, and the caller's side never leaves a joinable thread behind
        std::exception_ptr worker_error;
        std::thread worker([&] {
            try {
                left.grow_parallel(L, depth + 1, par_levels - 1);
            } catch (...) {
                worker_error = std::current_exception();
            }
        });
        try {
            right.grow_parallel(R, depth + 1, par_levels - 1);
        } catch (...) {
            worker.join();
            throw;
        }
        worker.join();
        if (worker_error) std::rethrow_exception(worker_error);
        const size_t me = nodes.size();
        uint32_t bits;
        std::memcpy(&bits, &p.d, 4);
        nodes.push_back({bits, (uint32_t)p.axis});
        append(left);
        nodes[me].w1 = ((uint32_t)nodes.size() << 2) | (uint32_t)p.axis;
        append(right);
    }
};

}  // namespace

void build_flat_kdtree(const float* tris, uint32_t n, const ptx_sah_cost& sah,
                       std::vector<ptx_kdnode>& nodes, std::vector<uint32_t>& leaves) {
    // AnyTriangle::from (any_triangle.rs:9-16): an axis-aligned triangle keeps ONE plane distance,
    // v0[axis]; rebuild its vertices from it as AxiallyAlignedTriangle::as_arrays does (only the sign
    // of a zero can differ from the input).
    std::vector<float> canon(tris, tris + (size_t)9 * n);
    for (uint32_t i = 0; i < n; ++i) {
        int flat = aligned_axis(&canon[9 * (size_t)i]);
        if (flat < 3) canon[9 * (size_t)i + 3 + flat] = canon[9 * (size_t)i + 6 + flat] = canon[9 * (size_t)i + flat];
    }
    tris = canon.data();
    Builder b;
    b.tris = tris;
    b.n = n;
    b.sah = sah;
    b.side.assign(n, SIDE_BOTH);
    Cell root;
    root.ids.resize(n);
    bool first = true;
    for (uint32_t i = 0; i < n; ++i) {
        root.ids[i] = i;
        // bounds through the same min/max nesting as Triangle::min/max (triangle.rs:49-56); for an
        // axis-aligned triangle the 2-D form (axial_triangle.rs:15-24) gives the same three numbers.
        F3 lo = lo3(b.vert(i, 0), lo3(b.vert(i, 1), b.vert(i, 2)));
        F3 hi = hi3(b.vert(i, 0), hi3(b.vert(i, 1), b.vert(i, 2)));
        int flat = aligned_axis(tris + 9 * i);
        if (flat < 3) lo[flat] = hi[flat] = tris[9 * i + flat];  // plane.distance = v0[axis], sign of zero kept
        Builder::push_events(root.ev, i, lo, hi);
        if (first) {
            root.box = {lo, hi};
            first = false;
        } else {
            root.box.lo = lo3(root.box.lo, lo);
            root.box.hi = hi3(root.box.hi, hi);
        }
    }
    if (first) root.box = {f3(0, 0, 0), f3(0, 0, 0)};
    for (int a = 0; a < 3; ++a) std::sort(root.ev[a].begin(), root.ev[a].end(), ev_before);
    // 2^4 = 16 subtrees in flight at most (PTX_BUILD_THREADS=1 builds on the calling thread alone)
    int par_levels = 4;
    if (const char* e = getenv("PTX_BUILD_THREADS")) {
        const int t = atoi(e);
        par_levels = t <= 1 ? 0 : (t <= 2 ? 1 : (t <= 4 ? 2 : (t <= 8 ? 3 : 4)));
    }
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw && hw < 4) par_levels = hw < 2 ? 0 : 1;
    b.grow_parallel(root, 1, par_levels);
    nodes.swap(b.nodes);
    leaves.swap(b.leaves);
}

}  // namespace ptx
