// device_tree.hpp - the device layout of the kd-tree (child descriptors, shared leaf-record runs) and the host
// function that derives it from the ABI's preorder tree.  Host code; the kernels' view is described above
// k_traverse in ptx_device.hip.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/ptx.h"

namespace ptx {

// child descriptors (int32): >= 0 inner-node index, DESC_LEAF | offset of the leaf's first record, DESC_EXIT = empty
// leaf, DESC_IDLE = the lane holds no ray
constexpr int32_t DESC_EXIT = -1, DESC_IDLE = -2;
constexpr uint32_t DESC_LEAF = 0x80000000u, DESC_PAYLOAD = 0x3FFFFFFFu;
constexpr uint32_t DESC_MAX_PAYLOAD = 0x3FFFFFFDu;  // offsets and triangle indices stay below 2^30 - 2

struct DevNode {  // 16 bytes: one dwordx4 per visit
    uint32_t x;   // bits of the plane distance
    uint32_t y;   // axis
    uint32_t z;   // descriptor of the left child
    uint32_t w;   // descriptor of the right child
};

// Re-indexes a VALIDATED preorder tree (validate_flat_tree) into the device layout: inner nodes only, the first
// `top_nodes` of them breadth-first (the prefix the kernels keep in LDS), the rest depth-first with siblings adjacent;
// `list` = the leaf records to lay down, as triangle index | last-of-run << 31, equal lists and tails of longer lists
// sharing one run.  Returns an empty string, or why the tree does not fit the descriptor fields.
std::string build_device_tree(const ptx_kdnode* in, uint32_t n_nodes, const uint32_t* leaf_idx, uint32_t top_nodes,
                              std::vector<DevNode>& nodes, std::vector<uint32_t>& list, int32_t& root, uint32_t& top_n);

}  // namespace ptx
