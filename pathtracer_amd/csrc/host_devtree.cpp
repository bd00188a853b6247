// host_devtree.cpp - derives the device layout of the kd-tree (device_tree.hpp) from the ABI's preorder tree.
#include <algorithm>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>
#include "device_tree.hpp"
#include "host_internal.hpp"

namespace ptx {

std::string build_device_tree(const ptx_kdnode* in, uint32_t n_nodes, const uint32_t* leaf_idx, uint32_t top_nodes,
                              std::vector<DevNode>& nodes, std::vector<uint32_t>& list, int32_t& root, uint32_t& top_n) {
    top_n = 0;
    list.clear();
    auto is_leaf = [&](uint32_t i) { return (in[i].w1 & 3u) == 3u; };
    bool too_big = false;
    // Leaves with the same ordered triangle list share ONE run of `list` (and so of leaf_tris), and a list that is
    // the tail of a longer one points into that run - the "last of the leaf" flag sits on the run's last entry
    // either way.  The reference's trees repeat themselves a lot (cornell_teapot: 239 012 references in 139 958
    // non-empty leaves, 48 890 distinct lists), so the records the traversal gathers from shrink by more than half
    // and more of them stay in the L2.  Which triangles a leaf tests, and in which order, is unchanged.
    // Only tails of at most TAIL_MAX triangles are offered for sharing (a key is the bytes of the list it names: every
    // tail of a 20 000-triangle leaf - a depth-capped or heavily overlapping scene - would cost 2 x 800 MB of keys);
    // longer lists are still stored once each.  The shipped scenes' longest leaf holds 56 triangles.
    constexpr uint32_t TAIL_MAX = 64;
    std::unordered_map<std::string, uint32_t> run_of;  // bytes of a list (or of a short tail of a stored run) -> offset
    {
        auto key_of = [&](const uint32_t* p, uint32_t n) { return std::string(reinterpret_cast<const char*>(p), (size_t)n * 4u); };
        // which lists need a run of their own: taken longest first, a list that is already the tail of a longer one
        // does not.  The runs are then laid down in the tree's own (depth-first) order, so that the records of
        // neighbouring leaves stay near each other.
        std::vector<uint32_t> leaves;
        for (uint32_t i = 0; i < n_nodes; ++i)
            if (is_leaf(i) && (in[i].w1 >> 2) != 0) leaves.push_back(i);
        std::vector<uint32_t> by_length(leaves);
        std::stable_sort(by_length.begin(), by_length.end(), [&](uint32_t a, uint32_t b) { return (in[a].w1 >> 2) > (in[b].w1 >> 2); });
        std::unordered_map<std::string, bool> seen;  // list or tail of a list -> it is a list with a run of its own
        for (uint32_t src : by_length) {
            const uint32_t cnt = in[src].w1 >> 2, off = in[src].w0;
            if (seen.count(key_of(leaf_idx + off, cnt))) continue;
            seen[key_of(leaf_idx + off, cnt)] = true;
            for (uint32_t k = cnt > TAIL_MAX ? cnt - TAIL_MAX : 1u; k < cnt; ++k) seen.emplace(key_of(leaf_idx + off + k, cnt - k), false);
        }
        for (uint32_t src : leaves) {
            const uint32_t cnt = in[src].w1 >> 2, off = in[src].w0;
            const auto own = seen.find(key_of(leaf_idx + off, cnt));
            if (own == seen.end() || !own->second || run_of.count(own->first)) continue;
            const size_t at = list.size();
            if (at + cnt > DESC_MAX_PAYLOAD) {
                too_big = true;
                break;
            }
            for (uint32_t k = 0; k < cnt; ++k) {
                if (leaf_idx[off + k] > DESC_MAX_PAYLOAD) too_big = true;
                list.push_back(leaf_idx[off + k] & DESC_PAYLOAD);
            }
            list.back() |= 0x80000000u;
            run_of.emplace(key_of(leaf_idx + off, cnt), (uint32_t)at);
            for (uint32_t k = cnt > TAIL_MAX ? cnt - TAIL_MAX : 1u; k < cnt; ++k) run_of.emplace(key_of(leaf_idx + off + k, cnt - k), (uint32_t)(at + k));
        }
    }
    auto leaf_desc = [&](uint32_t src) -> uint32_t {
        const uint32_t cnt = in[src].w1 >> 2, off = in[src].w0;
        if (cnt == 0) return (uint32_t)DESC_EXIT;
        const auto it = run_of.find(std::string(reinterpret_cast<const char*>(leaf_idx + off), (size_t)cnt * 4u));
        if (it == run_of.end()) return (uint32_t)DESC_EXIT;  // only after too_big
        return DESC_LEAF | it->second;
    };
    nodes.clear();
    if (is_leaf(0)) {
        root = (int32_t)leaf_desc(0);
        return too_big ? "triangle index or leaf list beyond 2^30 - 3" : "";
    }
    uint32_t n_inner = 0;
    for (uint32_t i = 0; i < n_nodes; ++i) n_inner += is_leaf(i) ? 0u : 1u;
    nodes.assign(n_inner, DevNode{0u, 0u, (uint32_t)DESC_EXIT, (uint32_t)DESC_EXIT});
    root = 0;
    uint32_t used = 1;
    auto emit = [&](uint32_t src, uint32_t dst, std::vector<std::pair<uint32_t, uint32_t>>& out, bool left_last) {
        const uint32_t l = src + 1u, r = in[src].w1 >> 2;
        const bool ll = is_leaf(l), rl = is_leaf(r);
        const uint32_t dl = ll ? 0u : used++, dr = rl ? 0u : used++;  // inner siblings adjacent
        const uint32_t desc_l = ll ? leaf_desc(l) : dl, desc_r = rl ? leaf_desc(r) : dr;
        nodes[dst] = DevNode{in[src].w0, in[src].w1 & 3u, desc_l, desc_r};
        if (left_last) {  // for a LIFO: right pushed first so that the left subtree is numbered first
            if (!rl) out.push_back({r, dr});
            if (!ll) out.push_back({l, dl});
        } else {
            if (!ll) out.push_back({l, dl});
            if (!rl) out.push_back({r, dr});
        }
    };
    // breadth-first while the node being emitted is inside the LDS prefix ...
    std::vector<std::pair<uint32_t, uint32_t>> queue{{0u, 0u}};
    size_t head = 0;
    while (head < queue.size() && queue[head].second < top_nodes) {
        const auto e = queue[head++];
        emit(e.first, e.second, queue, false);
    }
    top_n = used < top_nodes ? used : top_nodes;
    if (top_n > n_inner) top_n = n_inner;
    // ... depth-first below it (every pending node roots a contiguous subtree)
    for (; head < queue.size(); ++head) {
        std::vector<std::pair<uint32_t, uint32_t>> stack{queue[head]};
        while (!stack.empty()) {
            const auto e = stack.back();
            stack.pop_back();
            emit(e.first, e.second, stack, true);
        }
    }
    return too_big ? "triangle index or leaf list beyond 2^30 - 3" : "";
}

}  // namespace ptx

using namespace ptx;

extern "C" int ptx_debug_device_tree(const ptx_kdnode* nodes, uint32_t n_nodes, const uint32_t* leaf_indices, uint32_t n_leaf_indices,
                                     uint32_t n_triangles, uint32_t top_nodes, uint32_t* node_words, uint32_t* records, uint32_t counts[4]) {
    if (!nodes || !counts || (n_leaf_indices && !leaf_indices)) return fail(PTX_E_ARG, "ptx_debug_device_tree: null argument");
    return guarded("ptx_debug_device_tree", [&]() -> int {
    const std::string why = validate_flat_tree(nodes, n_nodes, leaf_indices, n_leaf_indices, n_triangles);
    if (!why.empty()) return fail(PTX_E_ARG, "ptx_debug_device_tree: " + why);
    std::vector<DevNode> dn;
    std::vector<uint32_t> dl;
    int32_t root = 0;
    uint32_t top_n = 0;
    const std::string big = build_device_tree(nodes, n_nodes, leaf_indices, top_nodes, dn, dl, root, top_n);
    if (!big.empty()) return fail(PTX_E_LIMIT, "ptx_debug_device_tree: " + big);
    counts[0] = (uint32_t)dn.size(), counts[1] = (uint32_t)dl.size(), counts[2] = (uint32_t)root, counts[3] = top_n;
    if (node_words)
        for (size_t i = 0; i < dn.size(); ++i)
            node_words[4 * i] = dn[i].x, node_words[4 * i + 1] = dn[i].y, node_words[4 * i + 2] = dn[i].z, node_words[4 * i + 3] = dn[i].w;
    if (records)
        for (size_t i = 0; i < dl.size(); ++i) records[i] = dl[i];
    return PTX_OK;
    });
}
