// host_internal.hpp — declarations shared by the translation units of libptx_mi355x.so.
#pragma once
#include <cstdint>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "../../include/ptx.h"
#include "traverse.hpp"

namespace ptx {

// error plumbing (thread-local text behind ptx_last_error)
int fail(int code, const std::string& what);
void clear_error();

// No exception crosses the C ABI (include/ptx.h): the body of an entry point that allocates with C++ containers
// or starts threads runs inside this; std::bad_alloc becomes PTX_E_OOM, anything else PTX_E_HIP with its text.
template <class F>
int guarded(const char* who, F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        try {
            return fail(PTX_E_OOM, std::string(who) + ": out of host memory");
        } catch (...) {
            return PTX_E_OOM;
        }
    } catch (const std::exception& e) {
        try {
            return fail(PTX_E_HIP, std::string(who) + ": " + e.what());
        } catch (...) {
            return PTX_E_HIP;
        }
    } catch (...) {
        return PTX_E_HIP;
    }
}

void build_flat_kdtree(const float* tris, uint32_t n, const ptx_sah_cost& sah,
                       std::vector<ptx_kdnode>& nodes, std::vector<uint32_t>& leaves);

// Checks a flattened tree against what the traversal kernels assume (indices in range, preorder
// children, depth <= PTX_MAX_DEPTH).  Returns an empty string when valid, else the reason.
std::string validate_flat_tree(const ptx_kdnode* nodes, uint32_t n_nodes, const uint32_t* leaf_indices,
                               uint32_t n_leaf_indices, uint32_t n_triangles);

// Wraps arrays into the handle of ptx_kdtree_build / ptx_kdtree_view (host_scene.cpp owns the type).
ptx_kdtree* kdtree_from_arrays(std::vector<ptx_kdnode>&& nodes, std::vector<uint32_t>&& leaves);

// Device triangle record, 48 bytes = three 16-byte loads.
//   kind 3 (general, Möller–Trumbore):  f[0..2]=v0  f[3..5]=e1=v1-v0  f[6..8]=e2=v2-v0
//   kind 0/1/2 (axis-aligned on X/Y/Z): f[0]=plane distance  f[1..2]=v0'  f[3..4]=b1=v1'-v0'
//                                        f[5..6]=b2=v2'-v0'  (' = the two remaining coordinates)
//                                        f[7]=1/det(b1,b2), NaN when det == 0
using TriRecord = TriRec;
void pack_triangles(const float* tris, uint32_t n, std::vector<TriRecord>& out);
void pack_triangle(const float* v0v1v2, TriRecord& out);

}  // namespace ptx
