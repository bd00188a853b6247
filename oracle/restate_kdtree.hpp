// oracle/restate_kdtree.hpp — TEST INFRASTRUCTURE ONLY (the checker, never the product).
//
// CPU restatement of the reference's `kdtree` crate: pointer tree, SAH builder with perfect splits
// and the nearest-hit stack traversal.  Each function cites the reference file:line it follows.
// Pinned by the reference's own tests (kdtree/src/lib.rs:202-467, build.rs:52-223, sah.rs:359-442,
// event.rs:105-190) through tests/test_oracle_kat.py.
#pragma once
#include <algorithm>
#include <memory>
#include <string>

#include "restate_geometry.hpp"

namespace orc {

constexpr size_t MAX_DEPTH = 25;  // kdtree/src/lib.rs:16

// kdtree/src/lib.rs:18-26
struct KdNode {
    bool is_leaf = true;
    std::vector<uint32_t> indices;  // Leaf
    Aap plane{0, 0.0f};             // Node
    std::unique_ptr<KdNode> left, right;

    static std::unique_ptr<KdNode> new_leaf(std::vector<uint32_t> idx) {
        auto n = std::make_unique<KdNode>();
        n->is_leaf = true;
        n->indices = std::move(idx);
        return n;
    }
    static std::unique_ptr<KdNode> new_node(Aap plane, std::unique_ptr<KdNode> l,
                                            std::unique_ptr<KdNode> r) {
        auto n = std::make_unique<KdNode>();
        n->is_leaf = false;
        n->plane = plane;
        n->left = std::move(l);
        n->right = std::move(r);
        return n;
    }
};

struct TraversalCounters {
    uint64_t inner_nodes = 0;  // inner-node steps (8 B each in the flat layout)
    uint64_t leaves = 0;
    uint64_t tri_tests = 0;  // triangle tests (4 B index + record)
};

// kdtree/src/lib.rs:73-132
inline std::optional<IndexedIntersection> kd_intersect(const KdNode* root,
                                                       const std::vector<AnyTriangle>& geometries,
                                                       const Ray& ray, float t_start, float t_end,
                                                       TraversalCounters* ctr = nullptr) {
    struct Entry {
        const KdNode* node;
        float t1, t2;
    };
    const KdNode* node = root;
    float t1 = t_start;
    float t2 = t_end;
    Entry stack[MAX_DEPTH];
    size_t sp = 0;
    for (;;) {
        if (node->is_leaf) {
            if (ctr) ++ctr->leaves;
            auto result = intersect_closest_geometry(geometries, node->indices.begin(),
                                                     node->indices.end(), ray, t1, t2,
                                                     ctr ? &ctr->tri_tests : nullptr);
            if (result) return result;
            if (t2 == t_end) return std::nullopt;
            if (sp == 0) return std::nullopt;
            --sp;
            node = stack[sp].node;
            t1 = stack[sp].t1;
            t2 = stack[sp].t2;
        } else {
            if (ctr) ++ctr->inner_nodes;
            int axis = node->plane.axis;
            auto t = node->plane.intersect_ray(ray);
            if (t) {
                const KdNode* near_n;
                const KdNode* far_n;
                if (ray.direction[axis] >= 0.0f) {
                    near_n = node->left.get();
                    far_n = node->right.get();
                } else {
                    near_n = node->right.get();
                    far_n = node->left.get();
                }
                if (*t > t2) {
                    node = near_n;
                } else if (*t < t1) {
                    node = far_n;
                } else {
                    stack[sp++] = {far_n, *t, t2};
                    node = near_n;
                    t2 = *t;
                }
            } else if (ray.origin[axis] <= node->plane.distance) {
                node = node->left.get();
            } else {
                node = node->right.get();
            }
        }
    }
}

// ---------------------------------------------------------------- events (kdtree/src/event.rs)
enum EventKind : uint8_t { END = 0, PLANAR = 1, START = 2 };  // event.rs:8-12 (derive Ord order)

struct Event {
    uint32_t index;
    float distance;
    EventKind kind;
};

// f32::total_cmp as a signed-integer key
inline int32_t total_order_key(float f) {
    int32_t b;
    std::memcpy(&b, &f, 4);
    b ^= (int32_t)((uint32_t)(b >> 31) >> 1);
    return b;
}
// event.rs:46-50
inline int event_total_cmp(const Event& a, const Event& b) {
    int32_t ka = total_order_key(a.distance), kb = total_order_key(b.distance);
    if (ka != kb) return ka < kb ? -1 : 1;
    if (a.kind != b.kind) return a.kind < b.kind ? -1 : 1;
    return 0;
}
inline bool event_lt(const Event& a, const Event& b) { return event_total_cmp(a, b) < 0; }

using EventLists = std::vector<Event>[3];

// event.rs:57-81
inline void extend_vec_with_events(std::vector<Event>* events, uint32_t index, Vec3 mn, Vec3 mx) {
    for (int axis = 0; axis < 3; ++axis) {
        if (mn[axis] == mx[axis]) {
            events[axis].push_back({index, mn[axis], PLANAR});
        } else {
            events[axis].push_back({index, mn[axis], START});
            events[axis].push_back({index, mx[axis], END});
        }
    }
}

// kdtree/src/cell.rs:5-9
struct KdCell {
    Aabb boundary;
    std::vector<uint32_t> indices;
    std::vector<Event> events[3];
};

// event.rs:83-96 + cell.rs:32-38
inline KdCell generate_initial_cell(const std::vector<AnyTriangle>& geometries) {
    KdCell cell;
    cell.boundary = geometries_bounding_box(geometries);
    cell.indices.resize(geometries.size());
    for (uint32_t i = 0; i < geometries.size(); ++i) cell.indices[i] = i;
    for (uint32_t i = 0; i < geometries.size(); ++i)
        extend_vec_with_events(cell.events, i, geometries[i].gmin(), geometries[i].gmax());
    // sort_unstable_by(total_cmp): relative order of equal (distance, kind) keys is irrelevant to
    // every consumer (only counts per key and per-index side flags are derived from the lists).
    for (int a = 0; a < 3; ++a) std::sort(cell.events[a].begin(), cell.events[a].end(), event_lt);
    return cell;
}

// ---------------------------------------------------------------- SAH (kdtree/src/sah.rs)
struct SahCost {
    float traverse_cost = 1.0f;   // sah.rs:122-130
    float intersect_cost = 1.5f;
    float empty_factor = 0.8f;
};

enum class Side { Left, Right };
struct SahSplit {
    Aap plane;
    Side side;
    float cost;
};

// sah.rs:38-49
inline std::optional<SahSplit> zip_min(std::optional<SahSplit> a, std::optional<SahSplit> b) {
    if (!a && !b) return std::nullopt;
    if (a && !b) return a;
    if (!a && b) return b;
    return a->cost <= b->cost ? a : b;
}

// sah.rs:59-80
inline float sah_leaf_cost(const SahCost& s, size_t count) { return s.intersect_cost * (float)count; }
inline float sah_split_cost(const SahCost& s, float vol0, float vol1, float p0, float p1, size_t c0,
                            size_t c1) {
    float empty_factor =
        ((c0 == 0 && vol0 > 0.01f) || (c1 == 0 && vol1 > 0.01f)) ? s.empty_factor : 1.0f;
    float intersect_cost = s.intersect_cost * (p0 * (float)c0 + p1 * (float)c1);
    return empty_factor * (s.traverse_cost + intersect_cost);
}

// sah.rs:82-119
inline std::optional<SahSplit> split_cost_with_planar(const SahCost& s, const Aabb& boundary,
                                                      Aap plane, size_t c0, size_t c1, size_t c2) {
    size_t count = c0 + c1 + c2;
    if (boundary.volume() == 0.0f || count == 0) return std::nullopt;
    Aabb left, right;
    boundary.split(plane, left, right);
    float surface_area = boundary.surface_area();
    float vol0 = left.volume(), vol1 = right.volume();
    float p0 = left.surface_area() / surface_area;
    float p1 = right.surface_area() / surface_area;
    float intersect_cost = sah_leaf_cost(s, count);
    if (vol0 > 0.0f && vol1 > 0.0f) {
        float l = sah_split_cost(s, vol0, vol1, p0, p1, c0 + c1, c2);
        float r = sah_split_cost(s, vol0, vol1, p0, p1, c0, c2 + c1);
        if (l < intersect_cost || r < intersect_cost) {
            if (l <= r) return SahSplit{plane, Side::Left, l};
            return SahSplit{plane, Side::Right, r};
        }
        return std::nullopt;
    } else if (vol0 == 0.0f && c0 + c1 > 0) {
        float c = sah_split_cost(s, vol0, vol1, p0, p1, c0 + c1, c2);
        if (c < intersect_cost) return SahSplit{plane, Side::Left, c};
        return std::nullopt;
    } else if (vol1 == 0.0f && c1 + c2 > 0) {
        float c = sah_split_cost(s, vol0, vol1, p0, p1, c0, c1 + c2);
        if (c < intersect_cost) return SahSplit{plane, Side::Right, c};
        return std::nullopt;
    }
    return std::nullopt;
}

// sah.rs:139-180
inline std::optional<SahSplit> sweep_plane(const SahCost& sah, const Aabb& boundary, size_t count,
                                           int axis, const std::vector<Event>& events) {
    auto advance = [&](size_t i, float distance, EventKind kind) {
        size_t c = 0;
        while (i < events.size() && events[i].distance == distance && events[i].kind == kind) {
            ++c;
            ++i;
        }
        return c;
    };
    std::optional<SahSplit> best;
    size_t n_left = 0, n_right = count, i = 0;
    while (i < events.size()) {
        Aap p{axis, events[i].distance};
        size_t p_end = advance(i, p.distance, END);
        i += p_end;
        size_t p_planar = advance(i, p.distance, PLANAR);
        i += p_planar;
        size_t p_start = advance(i, p.distance, START);
        i += p_start;
        n_right -= p_planar;
        n_right -= p_end;
        auto cost = split_cost_with_planar(sah, boundary, p, n_left, p_planar, n_right);
        best = zip_min(best, cost);
        n_left += p_start;
        n_left += p_planar;
        if (p_end + p_planar + p_start == 0) {
            // Unreachable for finite distances; a NaN distance would stall the reference loop
            // forever (sah.rs:159-178).  Skip the event instead of hanging the checker.
            ++i;
        }
    }
    return best;
}

enum class EventSide : uint8_t { Both, LeftOnly, RightOnly };  // sah.rs:182-187

// sah.rs:189-216
inline void update_geometry_side(const std::vector<uint32_t>& indices, const std::vector<Event>* events,
                                 const SahSplit& best, std::vector<EventSide>& sides) {
    for (uint32_t i : indices) sides[i] = EventSide::Both;
    for (const Event& e : events[best.plane.axis]) {
        EventSide& side = sides[e.index];
        if (e.kind == END && e.distance <= best.plane.distance) {
            side = EventSide::LeftOnly;
        } else if (e.kind == START && e.distance >= best.plane.distance) {
            side = EventSide::RightOnly;
        } else if (e.kind == PLANAR) {
            if (e.distance < best.plane.distance ||
                (e.distance == best.plane.distance && best.side == Side::Left)) {
                side = EventSide::LeftOnly;
            } else if (e.distance > best.plane.distance ||
                       (e.distance == best.plane.distance && best.side == Side::Right)) {
                side = EventSide::RightOnly;
            }
        }
    }
}

struct KdSplit {
    Aap plane;
    KdCell left, right;
};

// sah.rs:218-312
inline KdSplit repartition(const std::vector<AnyTriangle>& geometries, const KdCell& cell,
                           const SahSplit& best, std::vector<EventSide>& sides) {
    update_geometry_side(cell.indices, cell.events, best, sides);
    KdSplit out;
    out.plane = best.plane;
    cell.boundary.split(best.plane, out.left.boundary, out.right.boundary);

    std::vector<Event> part_l[3], part_r[3];
    for (int a = 0; a < 3; ++a) {
        for (const Event& e : cell.events[a]) {
            switch (sides[e.index]) {
                case EventSide::Both: break;
                case EventSide::LeftOnly: part_l[a].push_back(e); break;
                case EventSide::RightOnly: part_r[a].push_back(e); break;
            }
        }
    }
    std::vector<Event> both_l[3], both_r[3];
    for (uint32_t index : cell.indices) {
        switch (sides[index]) {
            case EventSide::Both: {
                const AnyTriangle& g = geometries[index];
                if (auto clipped = g.clip_aabb(out.left.boundary)) {
                    extend_vec_with_events(both_l, index, clipped->min, clipped->max);
                    out.left.indices.push_back(index);
                }
                if (auto clipped = g.clip_aabb(out.right.boundary)) {
                    extend_vec_with_events(both_r, index, clipped->min, clipped->max);
                    out.right.indices.push_back(index);
                }
                break;
            }
            case EventSide::LeftOnly: out.left.indices.push_back(index); break;
            case EventSide::RightOnly: out.right.indices.push_back(index); break;
        }
    }
    for (int a = 0; a < 3; ++a) {
        std::sort(both_l[a].begin(), both_l[a].end(), event_lt);
        std::sort(both_r[a].begin(), both_r[a].end(), event_lt);
        // itertools merge_by(b, Event::le): take from `a` while a <= b  == std::merge (stable, a first)
        out.left.events[a].resize(part_l[a].size() + both_l[a].size());
        std::merge(part_l[a].begin(), part_l[a].end(), both_l[a].begin(), both_l[a].end(),
                   out.left.events[a].begin(), event_lt);
        out.right.events[a].resize(part_r[a].size() + both_r[a].size());
        std::merge(part_r[a].begin(), part_r[a].end(), both_r[a].begin(), both_r[a].end(),
                   out.right.events[a].begin(), event_lt);
    }
    return out;
}

// sah.rs:314-348
inline std::optional<SahSplit> find_best_plane(const SahCost& sah, const KdCell& cell) {
    auto x = sweep_plane(sah, cell.boundary, cell.indices.size(), AX, cell.events[0]);
    auto y = sweep_plane(sah, cell.boundary, cell.indices.size(), AY, cell.events[1]);
    auto z = sweep_plane(sah, cell.boundary, cell.indices.size(), AZ, cell.events[2]);
    return zip_min(x, zip_min(y, z));
}

// kdtree/src/build.rs:11-31
inline std::unique_ptr<KdNode> build_helper(const std::vector<AnyTriangle>& geometries,
                                            const SahCost& sah, uint32_t depth, KdCell cell,
                                            std::vector<EventSide>& sides) {
    if (depth >= MAX_DEPTH || cell.indices.empty()) return KdNode::new_leaf(std::move(cell.indices));
    auto best = find_best_plane(sah, cell);
    if (!best) return KdNode::new_leaf(std::move(cell.indices));
    KdSplit split = repartition(geometries, cell, *best, sides);
    cell = KdCell{};  // the parent cell is dead once repartitioned
    auto left = build_helper(geometries, sah, depth + 1, std::move(split.left), sides);
    auto right = build_helper(geometries, sah, depth + 1, std::move(split.right), sides);
    return KdNode::new_node(split.plane, std::move(left), std::move(right));
}

// kdtree/src/build.rs:33-41
inline std::unique_ptr<KdNode> build_kdtree(const std::vector<AnyTriangle>& geometries,
                                            const SahCost& sah) {
    std::vector<EventSide> sides(geometries.size(), EventSide::Both);
    return build_helper(geometries, sah, 1, generate_initial_cell(geometries), sides);
}

// Canonical text form used by the tests to compare trees: "N(axis,<hex bits of distance>,L,R)" for
// inner nodes and "[i,j,...]" for leaves (layout of `impl Display for KdNode`, kdtree/src/lib.rs:135-148,
// with the distance as raw bits so the comparison is exact).
inline void tree_to_string(const KdNode* n, std::string& out) {
    if (n->is_leaf) {
        out += '[';
        for (size_t i = 0; i < n->indices.size(); ++i) {
            if (i) out += ',';
            out += std::to_string(n->indices[i]);
        }
        out += ']';
    } else {
        uint32_t bits;
        std::memcpy(&bits, &n->plane.distance, 4);
        char buf[32];
        snprintf(buf, sizeof buf, "N(%c,%08x,", "XYZ"[n->plane.axis], bits);
        out += buf;
        tree_to_string(n->left.get(), out);
        out += ',';
        tree_to_string(n->right.get(), out);
        out += ')';
    }
}

// Statistics in the spirit of kdtree-cli/src/main.rs:142-224.
struct TreeStats {
    uint64_t nodes = 0, leaves = 0, empty_leaves = 0, leaf_refs = 0;
    uint32_t max_depth = 0, max_leaf = 0;
};
inline void tree_stats(const KdNode* n, uint32_t depth, TreeStats& s) {
    if (n->is_leaf) {
        ++s.leaves;
        if (n->indices.empty()) ++s.empty_leaves;
        s.leaf_refs += n->indices.size();
        s.max_depth = std::max(s.max_depth, depth);
        s.max_leaf = std::max<uint32_t>(s.max_leaf, (uint32_t)n->indices.size());
    } else {
        ++s.nodes;
        tree_stats(n->left.get(), depth + 1, s);
        tree_stats(n->right.get(), depth + 1, s);
    }
}

}  // namespace orc
