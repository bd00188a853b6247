// ptx_device.hip — gfx950 kernels and the device half of the C ABI (include/ptx.h).
//
// Wavefront path tracer: start → [extend → shade → shadow → resolve] × rounds → accumulate.
//   start       the first paths of every queue segment: jittered pinhole ray     pathtracer.rs:113-117, camera.rs:88-93
//   extend      nearest hit through the KD-tree (k_traverse<false>)              kdtree/src/lib.rs:73-132
//   shade       everything between the two intersect calls of trace_ray          pathtracer.rs:44-108, material.rs:167-191,
//               + next-ray and shadow-ray emission, compacted per queue segment    light.rs:61-76, properties.rs:24-31
//               (wave ballot + LDS prefix, no atomics), + path regeneration
//   shadow      occlusion of the light samples (k_traverse<true>, any hit)       pathtracer.rs:72-78
//   resolve     radiance += T * sum of the unoccluded contributions              pathtracer.rs:68-86
//   accumulate  image += per-path radiance, samples added in index order         pathtracer.rs:138-140
// A path's radiance is owned by its path id (no float atomics anywhere) and the queue order is deterministic,
// so an image is bit-reproducible.  All arithmetic is IEEE f32 without contraction; random numbers and
// sin/cos come from include/ptx_sampling.h.
// Layout of this file: helpers, k_traverse, shading kernels, host-side state (ptx_scene), the entry points.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include <unordered_map>
#include "../../include/ptx.h"
#include "../../include/ptx_sampling.h"
#include "host_internal.hpp"
#include "device_internal.hpp"
#include "device_tree.hpp"
#include "traverse.hpp"

using namespace ptx;

namespace {

#ifndef PTX_BLOCK
#define PTX_BLOCK 256
#endif
constexpr int BLOCK = PTX_BLOCK;
constexpr int WAVES = BLOCK / 64;

struct DevMaterial {
    float albedo[3];
    int32_t texture;
    float f0[3];
    float transmission;
    float ior;
    uint32_t tex_w, tex_h;
    const float* tex_rgb;
};

struct DevScene {
    const ptx_kdnode* nodes;
    const uint32_t* leaf_idx;
    const TriRec* tris;
    const float4* props;  // 4 x float4 per triangle: n0 n1 n2 (9) uv0 uv1 uv2 (6) material id (1)
    const DevMaterial* materials;
    ptx_light lights[PTX_MAX_LIGHTS];
    uint32_t n_lights;
    float env[3];
    uint32_t max_bounces;
};

struct Rad {  // packed rgb: one dwordx3 per access
    float x, y, z;
};

struct RenderState {
    ptx_ray* rays[2];
    ptx_hit* hits;
    Rad* transq[2];     // accumulated transport of the entries of extend queue A / B (travels with the ray; 12 bytes)
    Rad* radiance;       // [path id] accumulated radiance of the sample (12 bytes)
    // Shadow entries of several lights are laid out LIGHT-MAJOR inside a queue segment: the entry of light l of the
    // segment's k-th shadow record sits at sh_index(seg, l, k) = (seg * n_lights + l) * seg_cap + k, so the rays of a
    // segment towards ONE light are contiguous and the 64 lanes of a traversal wave aim at one light from neighbouring
    // surface points (as coherent as camera rays) instead of holding 16 origins x 4 targets.  The traversal launch
    // sees n_seg * n_lights segments of seg_sh[seg] rays each.  k_resolve folds the lights in MTL order either way.
    ptx_ray* sh_rays;    // [sh_index] shadow ray: origin point_above, direction target - origin, t in [0,1]
    // (24-byte shadow entries - the range [0, 1] is the same for every shadow ray and need not travel - take 3 % off k_shade
    // and add as much to the shadow launch, whose refill then reads two arrays or unaligned entries: round 3, not kept.)
    float4* sh_T;        // [i] {transport before this bounce's update, bits(pid)}
    Rad* sh_contrib;     // [sh_index] bsdf * emitted * |cos| (12 bytes)
    uint8_t* sh_occ;     // [sh_index] 1 = occluded
    uint8_t* sh_bounce;  // [i] bounce of the path that shot the shadow rays; only while the ray log is on, else null
    uint32_t* perm;      // [slot] PTX_FLAG_MATERIAL_QUEUES: the order in which k_shade takes the entries of a segment (by material of the hit), else null
    // Queues are SEGMENTED: segment s owns slots [s * seg_cap, (s + 1) * seg_cap) of every queue for the
    // whole batch; k_start fills the segments in path order, and workgroup s of k_shade / k_resolve reads
    // segment s and compacts its survivors to the front of segment s of the output queue with a counter of
    // its own.  No global atomic sits on the path of a queue write (a single hot counter line serialises at
    // ~12 ns per workgroup: it was what bound k_shade), and the queue order is deterministic.
    uint32_t* seg_ext[2];  // [n_seg] entries of each segment of the extend queues A/B
    uint32_t* seg_sh;      // [n_seg] entries of each segment of the shadow queue (paths; x n_lights rays)
    uint32_t n_seg, seg_cap;
    // Path regeneration: segment s is responsible for the paths [s * seg_range, (s + 1) * seg_range) of the
    // batch (path id = sample * n_pixels + pixel) but holds only seg_cap of them at a time; seg_next[s] paths
    // of its range have been started so far, and k_shade tops the segment up with fresh camera rays as paths
    // end.  Every launch of a long batch therefore stays wide, whatever the depth of the individual paths.
    // Queue entries carry their own path id and bounce number ({o, bits(path id)}, {d, bits(bounce)}).
    uint32_t* seg_next;
    uint32_t seg_range, n_paths;
    // Totals per round, as pairs: [2k] = extend queue of round k+1 (k = 0..rounds), [2r+1] = shadow queue of round r.
    // One 64-bit atomic add per workgroup and kernel.
    uint32_t* qcounts;
    uint32_t* cursors;     // [2 * CURSOR_WORDS * rounds] work cursors of the persistent traversal launches
    unsigned long long* totals;  // extend rays, shadow rays, ext inner, ext tris, sh inner, sh tris
    ptx_pinhole cam;
    uint64_t seed;
    uint32_t n_pixels, n_samples, first_sample;  // n_pixels = pixels of the rendered region
    uint32_t reg_x, reg_y, reg_w;                // the region's corner and width inside the cam.width x cam.height image
    uint32_t tiles_per_row;                      // reg_w / 8 when local pixel numbers run over 8 x 8 tiles, else 0 (see image_pixel)
    // Camera rays that cannot hit anything - they miss the (widened) bounding box of the triangles, box_clip() - are
    // finished where they are generated: radiance = 0 + 1 * environment (pathtracer.rs:44-46 at the first bounce) and
    // no queue entry.  They still count as extend rays (the reference traces them); totals[6] counts them separately.
    uint32_t cull;
    float box_lo[3], box_hi[3], env[3];
};

// ------------------------------------------------------------------------------------------ helpers
struct V3 {
    float x, y, z;
};
__device__ inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
__device__ inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ inline V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
__device__ inline V3 operator-(float s, V3 a) { return v3(s - a.x, s - a.y, s - a.z); }
__device__ inline float dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
__device__ inline V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
__device__ inline V3 normalize(V3 a) { return a * (1.0f / sqrtf(dot(a, a))); }
__device__ inline float luminance(V3 c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; }

// Stable compaction slot for the lanes of a workgroup that raise `flag`: wave ballot + prefix over the
// waves in LDS + ONE atomic on the queue counter per workgroup.  All threads of the block must call.
__device__ inline uint32_t block_slot(bool flag, uint32_t* counter, uint32_t* lds /* WAVES + 1 */) {
    const unsigned long long ballot = __ballot(flag);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t before = __popcll(ballot & ((1ull << lane) - 1ull));
    if (lane == 0) lds[wave] = (uint32_t)__popcll(ballot);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int w = 0; w < WAVES; ++w) {
            uint32_t c = lds[w];
            lds[w] = total;
            total += c;
        }
        lds[WAVES] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    const uint32_t slot = lds[WAVES] + lds[wave] + before;
    __syncthreads();  // lds is reused by the next call
    return slot;
}

// Workgroup-wide exclusive prefix of two flags at once (wave ballots + wave totals through LDS).  One barrier:
// the LDS scratch alternates between two halves, chosen by the caller's iteration parity.
template <int NW>
__device__ inline void block_prefix2(bool f0, bool f1, uint32_t* lds /* 2 x 2 x NW */, uint32_t parity, uint32_t& pre0,
                                     uint32_t& pre1, uint32_t& tot0, uint32_t& tot1) {
    const unsigned long long b0 = __ballot(f0), b1 = __ballot(f1);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t* l = lds + parity * 2u * NW;
    if (lane == 0) {
        l[wave] = (uint32_t)__popcll(b0);
        l[NW + wave] = (uint32_t)__popcll(b1);
    }
    __syncthreads();
    uint32_t p0 = 0, p1 = 0, t0 = 0, t1 = 0;
    for (uint32_t w = 0; w < (uint32_t)NW; ++w) {
        const uint32_t c0 = l[w], c1 = l[NW + w];
        p0 += w < wave ? c0 : 0u;
        p1 += w < wave ? c1 : 0u;
        t0 += c0;
        t1 += c1;
    }
    pre0 = p0 + (uint32_t)__popcll(b0 & lt);
    pre1 = p1 + (uint32_t)__popcll(b1 & lt);
    tot0 = t0;
    tot1 = t1;
}

__device__ inline void wave_add(unsigned long long* dst, uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63u) == 0 && v) atomicAdd(dst, (unsigned long long)v);
}

// ------------------------------------------------------------------------------------------ kernels
// Persistent traversal with lane refill over a device-private tree of child descriptors.
//
// Device tree (built by build_device_tree() from the ABI's preorder array; same tree, same decisions):
//   only INNER nodes are stored, 16 bytes each (one dwordx4 load per visit):
//       x = bits of the plane distance, y = axis, z = descriptor of the LEFT child, w = of the RIGHT child
//   a descriptor (int32) says what a child is without visiting it:
//       >= 0              index of an inner node
//       0x80000000 | off  leaf: leaf_tris[off ..] holds COPIES of its triangle records in leaf order, each
//                         with the triangle's index and a "last of the leaf" flag in its spare word, so a
//                         leaf visit is one dependent fetch (record) instead of two (index, then record)
//                         and the records of a leaf share cache lines
//       DESC_EXIT  (-1)   empty leaf, which is the same thing as "this lane must leave its leaf"
//       DESC_IDLE  (-2)   the lane holds no ray
//   so a lane's whole control state is ONE register (`cur`), every predicate of the loop is one compare,
//   and arriving at a leaf costs no node visit (19 % of the visits of the 8-byte layout on cornell_teapot).
//   Inner nodes are numbered depth-first with siblings adjacent.
//
// Scheduling
//   * every wave owns a cursor window [lo,hi) of the ray array, taken `chunk` rays at a time with ONE
//     atomic on the launch's cursor; lanes whose ray has finished are refilled from the window as soon
//     as REFILL_IDLE lanes of the wave are idle, so a long ray never holds 63 idle lanes hostage;
//   * each loop iteration steps all node-walking lanes by one node; the triangle step (three
//     16-byte loads of one record) only runs when at least TRI_BATCH lanes are waiting for it or no lane can
//     walk any further; lanes that have left their leaf pop or finish in the same iteration;
//   * the traversal stack (far child descriptor, t_split): its LDS_DEPTH newest levels live in an LDS ring
//     ([level][thread]: ds_write_b64 / ds_read_b64 without bank conflicts), older levels in a per-thread global
//     spill area (the reference allows 25 levels, kdtree/src/lib.rs:16).
// Decisions per ray are exactly those of kd_traverse() in traverse.hpp (= KdNode::intersect).
#ifndef PTX_TBLOCK
#define PTX_TBLOCK 1024  // 16 waves: two workgroups fill a CU, so the LDS copy of the tree top is held twice, not four times
#endif
#ifndef PTX_LDS_DEPTH
#define PTX_LDS_DEPTH 8
#endif
#ifndef PTX_TOP_NODES
#define PTX_TOP_NODES 1023  // 16 KB beside the 64 KB ring of a 1024-thread workgroup: two workgroups fill the 160 KB of a CU
#endif
#ifndef PTX_REFILL_IDLE
#define PTX_REFILL_IDLE 16
#endif
#ifndef PTX_TRI_BATCH
#define PTX_TRI_BATCH 24
#endif
#ifndef PTX_TRI_BATCH_ANY
#define PTX_TRI_BATCH_ANY 16
#endif
#ifndef PTX_BLOCKS_PER_CU
#define PTX_BLOCKS_PER_CU 2
#endif
constexpr int TBLOCK = PTX_TBLOCK;        // threads per traversal workgroup
constexpr int LDS_DEPTH = PTX_LDS_DEPTH;  // stack levels kept in LDS
constexpr int TOP_NODES = PTX_TOP_NODES;  // inner nodes [0, TOP_NODES) = the top of the tree, served from LDS
constexpr uint32_t REFILL_IDLE = PTX_REFILL_IDLE;
constexpr uint32_t TRI_BATCH_NEAREST = PTX_TRI_BATCH;  // lanes that must wait in a leaf before the triangle step runs
constexpr uint32_t TRI_BATCH_ANYHIT = PTX_TRI_BATCH_ANY;

#ifndef PTX_NODE_UNROLL
#define PTX_NODE_UNROLL 2  // node steps per loop iteration (1 / 2 / 3 / 4: 1.64 / 1.70 / 1.68 / 1.64 G samples/s)
#endif
#ifndef PTX_TRI_UNROLL
#define PTX_TRI_UNROLL 2  // triangles per triangle step (1 / 2 / 3: 1.705 / 1.725 / 1.67 G samples/s)
#endif
#ifndef PTX_TAIL_ROUNDS
#define PTX_TAIL_ROUNDS 2u  // 64-ray windows per wave handed out after the large ones
#endif
#ifndef PTX_WINDOWS_PER_WAVE
#define PTX_WINDOWS_PER_WAVE 4u  // a launch's windows are sized so that a wave gets about this many (up to PTX_MAX_CHUNK rays each)
#endif
#ifndef PTX_MAX_CHUNK
#define PTX_MAX_CHUNK 512u  // largest window of a nearest-hit launch (1024 while the cursor was one counter: see win_shape)
#endif
#ifndef PTX_MAX_CHUNK_ANY
#define PTX_MAX_CHUNK_ANY 256u  // ... of an any-hit launch
#endif
#ifndef PTX_SHADE_WAVES
#define PTX_SHADE_WAVES 1  // minimum waves per SIMD the register allocation of k_shade must allow
#endif
#ifndef PTX_ROUNDS
#define PTX_ROUNDS 3  // (node steps, exit step) rounds per loop iteration before the triangle step is considered
#endif
#ifndef PTX_STRIPES
#define PTX_STRIPES 4u  // counters of a launch's work cursor (a power of two, at most 8: CURSOR_WORDS); 4 / 8: a 4 M-ray launch on cornell 119 / 153 us, the scenes equal
#endif
#ifndef PTX_GRAB
#define PTX_GRAB 2u  // windows per atomic in launches of up to PTX_STATIC_UPTO windows per wave
#endif
#ifndef PTX_STATIC_UPTO
#define PTX_STATIC_UPTO 8u  // launches with up to this many windows per wave hand half of them out without the cursor
#endif
#ifndef PTX_FORCE_LOBE
#define PTX_FORCE_LOBE 0
#endif
#ifndef PTX_PIXEL_TILES
#define PTX_PIXEL_TILES 1
#endif
#ifndef PTX_SEG_MIN_PATHS
#define PTX_SEG_MIN_PATHS 16384ull
#endif
#ifndef PTX_SEG_SMALL_PATHS
#define PTX_SEG_SMALL_PATHS 1024ull
#endif
#ifndef PTX_SEG_ROUNDS
#define PTX_SEG_ROUNDS 8
#endif
#ifndef PTX_AUTO_MEMORY_FRACTION
#define PTX_AUTO_MEMORY_FRACTION 0.5
#endif
constexpr int CURSOR_STRIDE_WORDS = 32;                  // the counters of a cursor sit on separate 128-byte lines
constexpr int CURSOR_WORDS = 8 * CURSOR_STRIDE_WORDS;    // a launch's work cursor: 1 KB, zero before the launch
static_assert(PTX_STRIPES >= 1u && PTX_STRIPES <= 8u && (PTX_STRIPES & (PTX_STRIPES - 1u)) == 0u, "PTX_STRIPES: 1, 2, 4 or 8");

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Queue entries (rays, hits, transport, shadow records) are written once by one kernel and read once by the next:
// marked non-temporal so that they stream through the L2 instead of evicting the tree and the triangle records the
// traversal lives on (extend -2.5 %, shadow -1.7 % time; profiles/r02_loop_variants.txt).  Full 16-byte entries
// only: the same hint on the 12-byte read-modify-writes of the per-path radiance and on k_start's stores costs
// more than it saves (resolve +20 %, start x2).
#ifndef PTX_STREAM_HINT
#define PTX_STREAM_HINT 1
#endif
__device__ __forceinline__ float4 ld_stream(const float4* p) {
#if PTX_STREAM_HINT
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ uint4 ld_stream(const uint4* p) {
#if PTX_STREAM_HINT
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ void st_stream(float4* p, float4 v) {
#if PTX_STREAM_HINT
    f32x4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<f32x4*>(p));
#else
    *p = v;
#endif
}
__device__ __forceinline__ void st_stream(uint4* p, uint4 v) {
#if PTX_STREAM_HINT
    u32x4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(p));
#else
    *p = v;
#endif
}
// 12-byte entries of a stream that is read and written front to back (the transport queues): three dwords, one instruction
__device__ __forceinline__ void ld_stream3(const float* p, float& x, float& y, float& z) {
#if PTX_STREAM_HINT
    x = __builtin_nontemporal_load(p), y = __builtin_nontemporal_load(p + 1), z = __builtin_nontemporal_load(p + 2);
#else
    x = p[0], y = p[1], z = p[2];
#endif
}
__device__ __forceinline__ void st_stream3(float* p, float x, float y, float z) {
#if PTX_STREAM_HINT
    __builtin_nontemporal_store(x, p), __builtin_nontemporal_store(y, p + 1), __builtin_nontemporal_store(z, p + 2);
#else
    p[0] = x, p[1] = y, p[2] = z;
#endif
}
__device__ __forceinline__ void st_stream(uint8_t* p, uint8_t v) {
#if PTX_STREAM_HINT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// Scratch of ptx_intersect_batch_device_noalloc (include/ptx.h): the launch's work cursor (first 1024 bytes) and
// the spill levels of the traversal stacks of NOALLOC_GRID workgroups.  File-scope device memory: every GPU the
// library runs on holds its own copy at its own address until the process ends.  Host code never uses the
// array's name as an address: ptx_scene_create asks hipGetSymbolAddress once per device and keeps the pointer.
constexpr int NOALLOC_GRID = 256;  // workgroups of the allocation-free launch (one per CU)
constexpr size_t NOALLOC_SPILL_BYTES = (size_t)(PTX_MAX_DEPTH - LDS_DEPTH) * NOALLOC_GRID * TBLOCK * sizeof(uint2);
constexpr size_t NOALLOC_SCRATCH_BYTES = CURSOR_WORDS * sizeof(uint32_t) + NOALLOC_SPILL_BYTES;  // 34.0 MiB
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

// The reference walks a ray's whole range [t_min, t_max] - [0, f32::MAX] for the renderer's rays - and so visits, and
// tests the triangles of, every cell whose split planes the LINE crosses, also beyond the point where the ray has left
// the scene.  A triangle can only be hit inside the bounding box of the triangles, so the range is narrowed to the ray's
// passage through that box: the traversal then makes the same decisions on the part of the range that is left (the
// split distances do not change), reaches every leaf whose segment of the ray meets the box, and accepts there what the
// reference accepts.  The box is widened by a thousandth of the scene's extent, the slabs by 1e-5 of the ray's distance
// to them and the range by 1e-4 of itself - four orders of magnitude more than the rounding of the reciprocal and of the
// Moeller-Trumbore t, including for a ray that grazes a face of the box, where the widening of the box is divided by the
// same small direction component that amplifies the rounding.  PTX_FLAG_NO_BOX_CLIP walks the full range (the counting
// pass that prices the reference algorithm; tests compare the two hit for hit).
// Narrows [t1, t_end] (t1 <= t_end on entry); returns false when nothing is left: the ray cannot hit anything.
__device__ __forceinline__ bool box_clip(float ox, float oy, float oz, float dx, float dy, float dz, const float* box_lo, const float* box_hi,
                                         float& t1, float& t_end) {
    float te = t1, tx = t_end;
#define PTX_SLAB(o, d, k)                                                         \
    {                                                                             \
        const float inv = __builtin_amdgcn_rcpf(d);                               \
        const float wl = box_lo[k] - o, wh = box_hi[k] - o;                       \
        const float pad = 1.0e-5f * fmaxf(fabsf(wl), fabsf(wh));                  \
        const float ta = (wl - pad) * inv, tb = (wh + pad) * inv;                 \
        te = fmaxf(te, fminf(ta, tb)); /* a NaN (0 * inf) leaves the bound alone */ \
        tx = fminf(tx, fmaxf(ta, tb));                                            \
    }
    PTX_SLAB(ox, dx, 0)
    PTX_SLAB(oy, dy, 1)
    PTX_SLAB(oz, dz, 2)
#undef PTX_SLAB
    te = te - fabsf(te) * 1.0e-4f;
    tx = tx + fabsf(tx) * 1.0e-4f;
    if (te > t1) t1 = te;
    if (tx < t_end) t_end = tx;
    return t1 <= t_end;
}

// Windows.  A traversal launch hands its rays out in WINDOWS of one queue segment: `chunk` rays (a power of two in
// [64, PTX_MAX_CHUNK], about four windows per wave over the launch), all of them first, segment-major, so that
// neighbouring waves work on neighbouring rays; then the last rays of every segment (together about PTX_TAIL_ROUNDS x 64
// rays per wave, shared out in proportion to the segments' sizes) in windows of 64, so that the launch does not end
// with a few waves working through a large window each.  A window costs one atomic on the launch's cursor, and atomics
// on one address complete ~16 ns apart, so only windows that HOLD rays may exist: the segments of a launch are far
// from equally full on the open 4K scenes (nine camera rays in ten leave sr2 without a hit; a queue segment is a
// compact part of the image), and a window grid laid out for the fullest segment - what this kernel used until round 3
// - made the launch take one atomic per EMPTY window too (sr2 3840x2160: extend -20 %, shadow -15 % time just from
// 1024-ray instead of 512-ray windows).  k_plan therefore lays the windows of a segmented launch down as two prefix
// arrays over the segments (large windows, 64-ray windows) and k_traverse finds the segment of window w with a
// wave-wide 64-ary search (two or three loads of 64 entries).
struct WinShape {
    uint32_t chunk, tail_num;  // tail_num / 65536 = share of every segment that goes out in 64-ray windows
};
// The largest window differs by kind.  While the cursor was ONE counter (one window per 11-16 ns) large windows were what kept
// it off the critical path: 1024 rays for nearest-hit launches (sr2 3840x2160: -19 % against 512), 512 for any-hit ones.
// With the striped cursor (v31) smaller windows even the launches out again: 1024 / 512 -> 512 / 256: sr2 +3.1 %,
// cubesplosion +1.3 %, cornell_teapot +1.6 %, cornell_textured +0.4 %; 256 / 256: sr2 +4.2 %, cubesplosion +2.2 %,
// cornell_teapot -0.6 %, cornell_textured -1.1 % (profiles/r03_loop_variants.txt 18).
__device__ inline WinShape win_shape(unsigned long long total_rays, uint32_t n_waves, bool any_hit) {
    const unsigned long long est = total_rays / (n_waves * (unsigned long long)PTX_WINDOWS_PER_WAVE);
    const uint32_t max_chunk = any_hit ? PTX_MAX_CHUNK_ANY : PTX_MAX_CHUNK;
    WinShape ws{64u, 0u};
    while (ws.chunk < max_chunk && 2ull * ws.chunk <= est) ws.chunk *= 2u;
    if (ws.chunk > 64u) {
        const unsigned long long tail_total = (unsigned long long)PTX_TAIL_ROUNDS * n_waves * 64ull;
        const unsigned long long num = tail_total * 65536ull / total_rays;  // (chunk > 64 implies total_rays > 0)
        ws.tail_num = num < 65536ull ? (uint32_t)num : 65536u;
    }
    return ws;
}
// Rays of a segment of `cnt` rays that are handed out in 64-ray windows (the segment's last ones).
__device__ inline uint32_t win_tail(uint32_t cnt, uint32_t tail_num) {
    const uint32_t t = ((uint32_t)(((unsigned long long)cnt * tail_num) >> 16) + 63u) & ~63u;
    return t < cnt ? t : cnt;
}

// plan: chunk, large windows, 64-ray windows, tail_num, 4 spare words; prefix of the large windows [n_seg + 1]; prefix
// of the 64-ray windows [n_seg + 1]; rays per segment [n_seg]
constexpr uint32_t PLAN_HEADER = 8;
struct PlanArgs {
    const uint32_t* counts;  // segment s holds counts[s / div] * scale rays
    uint32_t n_seg, div, scale;
    uint32_t n_waves;        // waves of the traversal launch the plan is for
    uint32_t any_hit;        // ... and its kind
    uint32_t* plan;          // [PLAN_HEADER + 3 * n_seg + 2], null = nothing to plan
};

// One workgroup per plan (a launch plans the shadow queue of round b and the extend queue of round b + 1 together:
// k_shade(b) wrote both).  Pass 1 sums the rays (-> window size), pass 2 lays the exclusive prefix sums of the
// segments' window counts down, 1024 segments at a time.
__global__ __launch_bounds__(1024) void k_plan(PlanArgs p0, PlanArgs p1) {
    const PlanArgs p = blockIdx.x == 0 ? p0 : p1;
    if (!p.plan) return;
    __shared__ unsigned long long s_sum[16];
    __shared__ uint32_t s_tot[2][16];
    __shared__ WinShape s_shape;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    unsigned long long sum = 0;
    for (uint32_t s = tid; s < p.n_seg; s += 1024u) sum += (unsigned long long)p.counts[s / p.div] * p.scale;
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if (lane == 0) s_sum[wave] = sum;
    __syncthreads();
    if (tid == 0) {
        unsigned long long total = 0;
        for (int w = 0; w < 16; ++w) total += s_sum[w];
        s_shape = win_shape(total, p.n_waves, p.any_hit != 0u);
        p.plan[0] = s_shape.chunk;
        p.plan[3] = s_shape.tail_num;
    }
    __syncthreads();
    const WinShape ws = s_shape;
    uint32_t* const pref_a = p.plan + PLAN_HEADER;
    uint32_t* const pref_b = pref_a + p.n_seg + 1u;
    uint32_t carry_a = 0, carry_b = 0;
    for (uint32_t base = 0; base < p.n_seg; base += 1024u) {
        const uint32_t s = base + tid;
        uint32_t wa = 0, wb = 0;
        if (s < p.n_seg) {
            const uint32_t cnt = p.counts[s / p.div] * p.scale;
            const uint32_t tail = win_tail(cnt, ws.tail_num);
            wa = (cnt - tail + ws.chunk - 1u) / ws.chunk;
            wb = (tail + 63u) / 64u;
        }
        uint32_t ia = wa, ib = wb;  // inclusive scan inside the wave
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t ua = __shfl_up(ia, off), ub = __shfl_up(ib, off);
            if (lane >= (uint32_t)off) ia += ua, ib += ub;
        }
        if (lane == 63u) s_tot[0][wave] = ia, s_tot[1][wave] = ib;
        __syncthreads();
        uint32_t before_a = 0, before_b = 0, tile_a = 0, tile_b = 0;
        for (uint32_t w = 0; w < 16u; ++w) {
            const uint32_t ta = s_tot[0][w], tb = s_tot[1][w];
            before_a += w < wave ? ta : 0u, before_b += w < wave ? tb : 0u;
            tile_a += ta, tile_b += tb;
        }
        if (s < p.n_seg) {
            pref_a[s] = carry_a + before_a + ia - wa;
            pref_b[s] = carry_b + before_b + ib - wb;
            pref_b[p.n_seg + 1u + s] = p.counts[s / p.div] * p.scale;  // the segment's rays, for the traversal kernel
        }
        carry_a += tile_a, carry_b += tile_b;
        __syncthreads();
    }
    if (tid == 0) {
        pref_a[p.n_seg] = carry_a;
        pref_b[p.n_seg] = carry_b;
        p.plan[1] = carry_a;
        p.plan[2] = carry_b;
    }
}

struct TravArgs {
    const uint4* nodes;        // inner nodes, device layout
    uint32_t top_n;            // nodes [0, top_n) are copied into LDS by every workgroup (<= TOP_NODES)
    const TriRec* tris;        // leaf_tris: records in leaf order, f[10] = triangle index | last << 31
    const ptx_ray* rays;
    // rays [0, fixed_count), or (plan != NULL) a SEGMENTED queue: the first rays of each of n_seg segments, seg_cap
    // rays apart; how many each segment holds, and the windows over them, are in the plan (k_plan)
    const uint32_t* plan;
    uint32_t n_seg, seg_cap;
    uint32_t fixed_count;
    uint32_t internal;  // rays of the renderer's own extend queue: t in [0, f32::MAX], w fields carry path id / bounce
    uint32_t clip;      // narrow every ray's t range to the bounding box of the triangles (see the refill)
    uint32_t long_div;  // a split distance lies outside split_quotient()'s window: every ray is walked with the long division
    float box_lo[3], box_hi[3];  // that box, widened
    int32_t root;    // descriptor of the root
    ptx_hit* hits;   // nearest-hit output (ANY == false)
    uint8_t* occ;    // occlusion output    (ANY == true)
    uint32_t* cursor;  // zero before the launch
    uint2* spill;      // [(PTX_MAX_DEPTH - LDS_DEPTH) * spill_stride]
    uint32_t spill_stride;
    unsigned long long* work;  // [2] inner nodes, triangle tests (COUNT)
    unsigned long long* prof;  // [12] loop profile (COUNT): see ptx_debug_loop_profile
};

// Leaving a leaf (kdtree/src/lib.rs:95-103).  The two newest entries are read unconditionally from the ring (for an
// empty stack the values go unused) and selected afterwards; entries that were evicted to global memory are a rare
// branch on top.  A finished lane keeps its answer in registers until the next refill writes it out.
#ifdef PTX_SPILL_PROBE  // probe build: the "refills" / "live_lanes" slots of the loop profile count evictions / reloads
#define PTX_COUNT_SPILL_POP if (COUNT) ++pf_live;
#define PTX_COUNT_SPILL_PUSH if (COUNT) ++pf_refill;
#else
#define PTX_COUNT_SPILL_POP
#define PTX_COUNT_SPILL_PUSH
#endif
// The split distance t = (dist - o) / d of a node visit (kdtree/src/lib.rs:113, aap.rs:44-51) must be the IEEE quotient:
// it is compared with t1 / t2, becomes the t2 of the near segment and the t1 of the far one, and those decide which
// triangle hits a leaf accepts.  The compiler's division (v_div_scale x 2, v_rcp, five fma, v_div_fmas, v_div_fixup) is
// 12 of the node step's ~40 vector instructions.  The short form
//     r = RN(1/d)  (v_rcp_f32, one Newton step),  q0 = RN(n r),  t = fma(fma(-d, q0, n), r, q0)
// gives the same bits for EVERY pair of significands (tools/div_probe.hip: all 2^46 pairs, and 2^32 random pairs with
// random signs and exponents, on the GPU: profiles/r03_div_probe.txt) as long as nothing overflows or underflows on the
// way: with |d| in [2^-40, 2^10] and n = 0 or |n| in [2^-100, 2^61], r, q0 and t are normal and the remainder - a
// multiple of 2^(exponent(n) - 47) - is exact.  short_div_ok() is that guard, taken once per ray: when the ray's origin
// components and the tree's split distances are all 0 or in [2^-77, 2^60] in magnitude, n = dist - o is 0 or at least an
// ulp of the smaller of the two; the direction components must be 0 (no division then) or in [2^-40, 2^10].  A ray that
// fails it - NaN, infinities, denormals, 1e-13 in a direction - never enters the stepping loop: it is parked at the refill
// and walked to its end right after it, in a plain loop with the compiler's division (PTX_WALK_LONG).  (The sign of a
// zero quotient may differ between the two forms; t is only ever compared.)  What it buys (round 3, v28): 4.7 % fewer
// vector instructions in the extend kernel, 5.1 % in the shadow kernel, and 1-2 % of their time - cornell_teapot +0.9 %,
// sr2 +1.7 % of the frame (profiles/r03_loop_variants.txt 12): the kernel waits on its node and triangle loads for 60 %
// of a wave's life, and instructions taken out of the node step shorten that chain only a little.
__device__ __forceinline__ float split_quotient(float n, float d) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r0, 1.0f);
    const float r = __builtin_fmaf(e, r0, r0);
    const float q0 = n * r;
    const float rem = __builtin_fmaf(-d, q0, n);
    return __builtin_fmaf(rem, r, q0);
}
constexpr uint32_t DIV_POS_LO = (127u - 77u) << 23, DIV_POS_HI = (127u + 60u) << 23;  // origins, split distances
constexpr uint32_t DIV_DIR_LO = (127u - 40u) << 23, DIV_DIR_HI = (127u + 10u) << 23;  // directions
__host__ __device__ inline bool div_pos_ok(float x) {  // x is 0 or 2^-77 <= |x| <= 2^60
    const uint32_t b = f2u(x) & 0x7FFFFFFFu;
    return b - 1u >= DIV_POS_LO - 1u && b <= DIV_POS_HI;  // (0 - 1 wraps to the largest value)
}
__device__ __forceinline__ bool short_div_ok(float ox, float oy, float oz, float dx, float dy, float dz) {
    const uint32_t o0 = f2u(ox) & 0x7FFFFFFFu, o1 = f2u(oy) & 0x7FFFFFFFu, o2 = f2u(oz) & 0x7FFFFFFFu;
    const uint32_t d0 = f2u(dx) & 0x7FFFFFFFu, d1 = f2u(dy) & 0x7FFFFFFFu, d2 = f2u(dz) & 0x7FFFFFFFu;
    const bool pos = min(min(o0 - 1u, o1 - 1u), o2 - 1u) >= DIV_POS_LO - 1u && max(max(o0, o1), o2) <= DIV_POS_HI;
    const bool dir = min(min(d0 - 1u, d1 - 1u), d2 - 1u) >= DIV_DIR_LO - 1u && max(max(d0, d1), d2) <= DIV_DIR_HI;
    return pos && dir;
}
#define PTX_EXIT_STEP \
            if (COUNT && __any(cur == DESC_EXIT)) pf_end_iter += lane == 0 ? 1u : 0u; \
            if (cur == DESC_EXIT) { \
                if (COUNT) ++pf_end; \
                const uint32_t st = so - LVL; \
                const uint32_t sb = st - LVL; \
                uint2 top = PTX_SLOT(st); \
                uint32_t tb = PTX_SLOT(sb).y; \
                tb = sb == so0 ? f2u(t_end) : tb; \
                if (sb <= ev && ev != so0) { \
                    if (st <= ev) { \
                        top = PTX_SPILL(st); \
                        ev = sb; \
                        PTX_COUNT_SPILL_POP \
                    } \
                    if (sb != so0) tb = PTX_SPILL(sb).y; \
                } \
                const bool done = bi != PTX_MISS || t2 == t_end || st == so0; \
                cur = done ? DESC_IDLE : (int32_t)top.x; \
                t1 = done ? t1 : u2f(top.y); \
                t2 = done ? t2 : u2f(tb); \
                so = done ? so : st; \
            }
// One node visit (kdtree/src/lib.rs:105-129) of the lanes that stand at an inner node; QUOT(n, d) is the division.
// The top of the tree comes from LDS: no L1 tag lookup, a fifth of the latency.  The LDS read is written as an
// instruction: as C++ the two loads are merged into one flat_load of a selected pointer, which goes through the
// vector memory path for LDS addresses too.
#define PTX_NODE_FROM_LDS(nd) \
                { \
                    u32x4 v; \
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(top_base + ((uint32_t)cur << 4)) : "memory"); \
                    nd = make_uint4(v.x, v.y, v.z, v.w); \
                }
#define PTX_NODE_STEP(QUOT) \
            if (cur >= 0) { \
                uint4 nd; \
                if (cur < top_n) PTX_NODE_FROM_LDS(nd) else nd = a.nodes[cur]; \
                PTX_NODE_BODY(QUOT) \
            }
// (Requesting BOTH children of a node below the LDS prefix as soon as it has arrived, so that the next visit finds its
// node under way while this one is decided, was slower - extend +2.2 %, shadow +4 % on cornell_teapot, +2-3 % on sr2: half
// again as many node loads through a vector memory path that is two-thirds busy cost more than the shorter wait saved;
// profiles/r03_loop_variants.txt 13.)
#define PTX_NODE_BODY(QUOT) \
            { \
                if (COUNT) ++n_inner, ++pf_walk; \
                const uint32_t axis = nd.y; \
                const float dist = u2f(nd.x); \
                const float da = pick(dx, dy, dz, axis); \
                const float oa = pick(ox, oy, oz, axis); \
                const float t = QUOT(dist - oa, da); /* unused (selected away) when da == 0 */ \
                const bool fwd = da >= 0.0f; \
                const uint32_t near_d = fwd ? nd.z : nd.w; \
                const uint32_t far_d = fwd ? nd.w : nd.z; \
                const bool parallel = da == 0.0f; \
                const bool only_near = t > t2, only_far = t < t1; \
                const bool push = !parallel && !only_near && !only_far; \
                uint32_t next = only_far ? far_d : near_d; \
                next = parallel ? (oa <= dist ? nd.z : nd.w) : next; \
                if (push) { \
                    if (so - ev > RING) { /* the ring is full: its oldest level sits in the slot this push takes */ \
                        ev += LVL; \
                        PTX_SPILL(ev) = PTX_SLOT(so); \
                        PTX_COUNT_SPILL_PUSH \
                    } \
                    PTX_SLOT(so) = make_uint2(far_d, f2u(t)); \
                    so += LVL; \
                    t2 = t; \
                } \
                cur = (int32_t)next; \
            }
#define PTX_QUOT_LONG(n, d) ((n) / (d))
#define PTX_QUOT_SHORT(n, d) split_quotient((n), (d))
// The kernel's argument block as the refill reads it (see there).
typedef __attribute__((address_space(4))) const TravArgs KernArgs;
__device__ __forceinline__ KernArgs* kernargs() {
    KernArgs* p = (KernArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
// One triangle of the leaf a lane waits in: geometry/src/geometry.rs:176-196.  The whole 48-byte record comes in one
// go; the kind is its last word.
#define PTX_TRI_STEP \
                if (cur < -2) { \
                    if (COUNT) ++pf_tri, ++n_tris; \
                    TriRec rec; \
                    { \
                        const float4* rp = reinterpret_cast<const float4*>(a.tris + ((uint32_t)cur & DESC_PAYLOAD)); \
                        const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2]; \
                        __builtin_memcpy(&rec.f[0], &r0, 16); \
                        __builtin_memcpy(&rec.f[4], &r1, 16); \
                        __builtin_memcpy(&rec.f[8], &r2, 16); \
                    } \
                    const uint32_t tag = f2u(rec.f[10]); \
                    const uint32_t idx = tag & DESC_PAYLOAD; \
                    const bool last = (int32_t)tag < 0; \
                    float t, u, v; \
                    bool accept = false; \
                    if (tri_test(rec, ox, oy, oz, dx, dy, dz, t, u, v) && t1 <= t && t <= t2) { \
                        if (bi == PTX_MISS || !(bt <= t)) { /* the earlier triangle wins t ties */ \
                            bt = t, bu = u, bv = v, bi = idx; \
                            accept = true; \
                        } \
                    } \
                    cur = (last || (ANY && accept)) ? DESC_EXIT : cur + 1; \
                }
// The answer of a lane that has finished its ray.  (Resolving here - radiance[path] += contribution for an unoccluded
// shadow ray - was measured slower than the streaming resolve: the scattered read-modify-write stalls the whole wave.)
#define PTX_WRITE_OUT \
            { \
                const auto* const kw = kernargs(); \
                if (ANY) { \
                    st_stream(kw->occ + ridx, (uint8_t)(bi != PTX_MISS ? 1 : 0)); \
                } else { \
                    st_stream(reinterpret_cast<uint4*>(kw->hits) + ridx, make_uint4(bi, f2u(bt), f2u(bu), f2u(bv))); \
                } \
                ridx = NO_RAY; \
            }
// Scalar registers count: two 1024-thread workgroups share a CU only while a wave's SGPR allocation stays at 80 or
// below (measured in round 3: at 82 - the compiler still reports "occupancy 8" - half of the waves of a launch are
// resident on average, SQ_WAVE_CYCLES x 4 / kernel cycles = 3840 instead of 7450, and every launch takes 1.45x as
// long; capping the allocation with amdgpu_num_sgpr spills scalars into a 65th VGPR, which costs the same workgroup).
// That is why the launch's uniform state is kept small: a segmented launch reads everything about its segments from
// the plan (one pointer); what only the refill needs - the queue, the plan, the window arithmetic, the box, the outputs -
// is read from the kernel's argument block there, through kernargs(), a pointer the compiler cannot see through, and so
// does not stay in scalar registers across the stepping loop (76 -> 54 SGPRs, v28); and the Makefile checks the kernel's
// register counts after every build.  The launch bounds say what the allocator must fit: 8 waves per SIMD, 64 VGPRs.
template <bool ANY, bool COUNT>
__global__ __launch_bounds__(TBLOCK, 8) void k_traverse(TravArgs a) {
    // The traversal stack of a lane is a ring of LDS_DEPTH entries in LDS that always holds the NEWEST levels: when
    // a push finds the ring full, the oldest level it holds moves to the lane's column of a.spill (a store, off the
    // critical path), and a pop only reads global memory once the ring has run empty while older levels remain -
    // rare, because a ray that finds its hit never comes back for them.  (Spilling the newest levels instead put a
    // dependent global load into 0.27 of the wave iterations of the extend kernel on cornell_teapot and 0.87 of the
    // shadow kernel's on sr2: tools/spill_probe.py.)  Stack positions are LOGICAL byte offsets, off(level) = so0 +
    // LVL * level, kept so that the ring slot of a level is off & (RING - 1): one AND per access.  Level 0 is "below
    // the first entry": its t is the ray's t_end, taken from the register.
    static_assert((LDS_DEPTH & (LDS_DEPTH - 1)) == 0 && (TBLOCK & (TBLOCK - 1)) == 0, "the ring needs powers of two");
    __shared__ uint2 s_stack[LDS_DEPTH][TBLOCK];
    __shared__ uint4 s_top[TOP_NODES > 0 ? TOP_NODES : 1];
    __shared__ uint32_t s_dry;  // bit c: a wave of this workgroup has found counter c of the cursor dry - the others need not ask
    char* const s_base = reinterpret_cast<char*>(&s_stack[0][0]);
    constexpr uint32_t LVL = TBLOCK * 8u;        // bytes per stack level
    constexpr uint32_t RING = LVL * LDS_DEPTH;   // bytes of the ring
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t gtid = blockIdx.x * TBLOCK + tid;
    const uint32_t so0 = RING - LVL + tid * 8u;  // off(0): level 1 lands in ring slot 0 of this lane's column
    const int32_t top_n = (int32_t)a.top_n;
    const uint32_t top_base = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint4*)s_top);
    for (uint32_t i = tid; i < a.top_n; i += TBLOCK) s_top[i] = a.nodes[i];
    if (tid == 0) s_dry = 0u;
    __syncthreads();
    uint32_t lo = 0, hi = 0;
    uint32_t windows = 0;  // 2: no window is left for this wave
    uint32_t taken = 0;    // windows this wave has taken by its own number
    uint32_t pend_next = 0, pend_end = 0;  // windows taken from the cursor several at a time and not yet used (PTX_STRIPES apart)
    uint32_t stripe = __builtin_amdgcn_readfirstlane((blockIdx.x * (TBLOCK / 64) + (tid >> 6)) & (PTX_STRIPES - 1u));  // the counter this wave asks first
    constexpr int32_t DESC_PARKED = -3;  // a ray that waits for the long walk (no leaf descriptor: those end at 0xBFFFFFFD)
    int32_t cur = DESC_IDLE;
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, t_end = 0, t1 = 0, t2 = 0;
    constexpr uint32_t NO_RAY = 0xFFFFFFFFu;
    uint32_t ridx = NO_RAY;  // ray whose answer this lane holds or works on
    uint32_t so = so0 + LVL;  // off(level of the next push); so0 + LVL == empty
    uint32_t ev = so0;        // off(newest level that lives in a.spill); so0 == none
    uint32_t bi = PTX_MISS;
    float bt = 0.0f, bu = 0.0f, bv = 0.0f;  // stay 0 while bi == PTX_MISS
    uint32_t n_inner = 0, n_tris = 0;
    uint32_t pf_iter = 0, pf_walk = 0, pf_tri_iter = 0, pf_tri = 0, pf_end_iter = 0, pf_end = 0, pf_refill = 0, pf_live = 0;
    uint32_t ck_refill = 0, ck_node = 0, ck_tri = 0, ck_exit = 0;  // wave clock spent in each part (COUNT)
    long long ck = COUNT ? clock64() : 0;
#define PTX_CLOCK(acc)                               \
    if (COUNT) {                                     \
        const long long now = clock64();             \
        acc += lane == 0 ? (uint32_t)(now - ck) : 0u; \
        ck = now;                                    \
    }
#define PTX_SLOT(off) (*reinterpret_cast<uint2*>(s_base + ((off) & (RING - 1u))))
#define PTX_SPILL(off) a.spill[(size_t)(((off) - so0) / LVL - 1u) * a.spill_stride + gtid]

    for (;;) {
#ifndef PTX_SPILL_PROBE
        if (COUNT) pf_refill += lane == 0 ? 1u : 0u;
#endif
        // ---- write out the answers of the lanes that have finished since the last refill: once per refill
        // with many lanes, instead of inside the stepping loop whenever a single lane finishes
        if (cur == DESC_IDLE && ridx != NO_RAY) PTX_WRITE_OUT
        // ---- refill idle lanes from the wave's window
        for (;;) {
            const unsigned long long idle = __ballot(cur == DESC_IDLE);
            if (!idle) break;
            // Everything the refill needs of the launch - the queue, the plan, the box - is read from the kernel's argument
            // block when it is needed (scalar loads through a pointer the compiler cannot see through) instead of living
            // in scalar registers across the stepping loop: see "Scalar registers count" above.
            const auto* const ka = kernargs();
            if (lo >= hi) {
                if (windows == 2u) break;
                // Windows (see k_plan): planned for a segmented launch, one segment's arithmetic otherwise.
                const uint32_t* const plan = ka->plan;
                const bool segmented = plan != nullptr;
                const uint32_t fixed_count = ka->fixed_count;
                const uint32_t n_waves = gridDim.x * (TBLOCK / 64);
                const uint32_t wave_id = __builtin_amdgcn_readfirstlane(blockIdx.x * (TBLOCK / 64) + (tid >> 6));
                uint32_t chunk, n_a, n_windows, tail_num;
                if (segmented) {
                    chunk = plan[0], n_a = plan[1], n_windows = n_a + plan[2], tail_num = plan[3];
                } else {
                    const WinShape ws = win_shape(fixed_count, n_waves, ANY);
                    const uint32_t tail = win_tail(fixed_count, ws.tail_num);
                    chunk = ws.chunk, tail_num = ws.tail_num;
                    n_a = (fixed_count - tail + chunk - 1u) / chunk;
                    n_windows = n_a + (tail + 63u) / 64u;
                }
                const uint32_t n_pref = ka->n_seg + 1u;  // entries of each prefix array of the plan
                // Atomics on ONE address complete ~16 ns apart however many waves issue them (they are resolved at the memory
                // side), and a launch has about four windows per wave whatever its size, so handing every window out through
                // the cursor cost every launch 32 k x 16 ns = 0.5 ms - most of the time of the small late-bounce launches of a
                // batch and all of config 1's (round 3, tools/launch_floor.py: a million camera rays took 200 us on cornell AND
                // on cornell_teapot).  So a wave takes its first windows by its own number - window wave + k * n_waves for
                // k < n_own, no atomic: all of them when there are at most two per wave (the cursor is then never touched),
                // half of them up to PTX_STATIC_UPTO per wave, and just the first one in the large launches, where the cursor
                // is what keeps the waves even (three quarters by number cost cornell_teapot's camera launch 11 %: 29.0 ->
                // 25.8 Grays/s).
                const uint32_t per_wave = n_windows / n_waves;
                const uint32_t n_own = per_wave <= 2u ? (n_windows + n_waves - 1u) / n_waves : (per_wave <= PTX_STATIC_UPTO ? per_wave / 2u : 1u);
                const uint32_t n_static = n_own * n_waves;  // windows [0, n_static) are taken by number, [n_static, n_windows) from the cursor
                uint32_t w = 0;
                if (taken < n_own) {
                    w = wave_id + taken * n_waves;
                    ++taken;
                    if (w >= n_windows) {  // (only the last of a wave's own windows can lie past the end)
                        windows = 2u;
                        break;
                    }
                } else {
                    if (n_static >= n_windows) {  // nothing was left for the cursor: it is never touched
                        windows = 2u;
                        break;
                    }
                    // The cursor is PTX_STRIPES counters, one 128-byte line each: counter c hands out the windows c, c + STRIPES,
                    // c + 2 STRIPES ... of the pool.  Atomics on one line complete 11 ns apart however many waves issue them, and
                    // counters on separate lines do so independently (tools/atomic_probe.hip: 8 counters, 1.5 ns per atomic) -
                    // while the windows under way at any time stay neighbours in the queue, as with one counter.  A wave starts at
                    // counter wave % STRIPES and, when that one has run dry, goes on to the next; a workgroup keeps the set of dry
                    // counters in LDS, so that the last, failing atomic of a counter is paid once per workgroup and not once per
                    // wave.  A launch of a few windows per wave (all of them under way at once; a 4 M-ray launch on cornell took
                    // 405 us, 393 of them the single cursor's) takes PTX_GRAB windows of a counter per atomic.  (Large launches keep
                    // single windows: grabs of remaining / (4 x waves) windows cost cornell_teapot 15 % - the waves then work on
                    // distant parts of the queue at the same time and share less of the tree in the L2.)
                    if (pend_next < pend_end) {
                        w = pend_next;
                        pend_next += PTX_STRIPES;
                    } else {
                        const uint32_t pool = n_windows - n_static;
                        const uint32_t g = per_wave <= PTX_STATIC_UPTO ? PTX_GRAB : 1u;
                        uint32_t dry = *(volatile uint32_t*)&s_dry;
                        bool found = false;
                        for (uint32_t tries = 0; tries < PTX_STRIPES && !found; ++tries) {
                            const uint32_t c = (stripe + tries) & (PTX_STRIPES - 1u);
                            if ((dry >> c) & 1u) continue;
                            uint32_t unit = 0;
                            if (lane == 0) unit = atomicAdd(ka->cursor + c * CURSOR_STRIDE_WORDS, g);
                            unit = __builtin_amdgcn_readfirstlane(unit);
                            const uint32_t first = unit * PTX_STRIPES + c;  // its place in the pool
                            if (first >= pool) {
                                dry |= 1u << c;
                                if (lane == 0) atomicOr(&s_dry, 1u << c);
                                continue;
                            }
                            found = true;
                            stripe = c;
                            w = first + n_static;
                            const uint32_t stop = (unit + g) * PTX_STRIPES < pool ? (unit + g) * PTX_STRIPES : pool;
                            pend_next = w + PTX_STRIPES;
                            pend_end = stop + n_static;
                        }
                        if (!found) {
                            windows = 2u;
                            break;
                        }
                    }
                }
                const bool large = w < n_a;
                const uint32_t wl = large ? w : w - n_a;
                uint32_t seg = 0, k = wl;
                if (segmented) {
                    // the segment of window wl: the largest s with pref[s] <= wl (pref[0] = 0, pref[n_seg] = all windows
                    // > wl; empty segments repeat their successor's value), by a 64-ary search with the whole wave
                    const uint32_t* pref = plan + PLAN_HEADER + (large ? 0u : n_pref);
                    uint32_t span = n_pref, at = 0, at_val = 0;
                    while (span > 1u) {
                        const uint32_t step = (span + 63u) / 64u;
                        const bool in = lane * step < span;
                        const uint32_t v = in ? pref[at + lane * step] : 0u;
                        const uint32_t c = (uint32_t)__popcll(__ballot(in && v <= wl)) - 1u;  // lane 0 always qualifies
                        at_val = __builtin_amdgcn_readlane(v, c);
                        at += c * step;
                        span = span - c * step < step ? span - c * step : step;
                    }
                    seg = __builtin_amdgcn_readfirstlane(at);
                    k = wl - at_val;
                }
                const uint32_t cnt = segmented ? plan[PLAN_HEADER + 2u * n_pref + seg] : fixed_count;
                const uint32_t head = cnt - win_tail(cnt, tail_num);  // rays of this segment handed out in large windows
                const uint32_t first = large ? k * chunk : head + k * 64u;
                const uint32_t end = large ? head : cnt;
                const uint32_t len = large ? chunk : 64u;
                const uint32_t seg_cap = ka->seg_cap;
                lo = seg * seg_cap + first;
                hi = seg * seg_cap + (first + len < end ? first + len : end);
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));  // idle lanes below this one
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            const uint32_t avail = hi - lo;
            if (cur == DESC_IDLE && rank < avail) {
                ridx = lo + rank;
                const ptx_ray* const rays = ka->rays;
                const bool internal = ka->internal != 0u;
                const float4 ra = ld_stream(reinterpret_cast<const float4*>(rays) + 2 * (size_t)ridx);
                const float4 rb = ld_stream(reinterpret_cast<const float4*>(rays) + 2 * (size_t)ridx + 1);
                ox = ra.x, oy = ra.y, oz = ra.z, t1 = internal ? 0.0f : ra.w;
                dx = rb.x, dy = rb.y, dz = rb.z, t2 = t_end = internal ? 3.402823466e+38f : rb.w;
                cur = ka->root;
                const bool odd = ka->long_div != 0u || !short_div_ok(ox, oy, oz, dx, dy, dz);  // outside split_quotient()'s window
                if (ka->clip && t1 <= t_end) {  // (a window that is empty or NaN is walked as the reference walks it)
                    // narrow the range to the ray's passage through the box of the triangles (box_clip above)
                    float lo_t = t1, hi_t = t_end;
                    const float box_lo[3] = {ka->box_lo[0], ka->box_lo[1], ka->box_lo[2]};
                    const float box_hi[3] = {ka->box_hi[0], ka->box_hi[1], ka->box_hi[2]};
                    const bool meets = box_clip(ox, oy, oz, dx, dy, dz, box_lo, box_hi, lo_t, hi_t);
                    t1 = lo_t;
                    t2 = t_end = hi_t;
                    if (!meets) cur = DESC_EXIT;  // the ray misses the box: done at the next exit step (empty stack)
                }
                so = so0 + LVL;
                ev = so0;
                bi = PTX_MISS;
                bt = bu = bv = 0.0f;
                if (odd && cur >= 0) cur = DESC_PARKED;  // walked below
            }
            lo += n_idle < avail ? n_idle : avail;
        }
        {
            // PTX_WALK_LONG: while a parked ray is under way, EVERY live lane steps with the compiler's division (one node
            // visit, one triangle, one leaf exit per pass - the long division is right for all of them); then back to the
            // top, where the finished lanes write their answers and take new rays
            const unsigned long long parked = __ballot(cur == DESC_PARKED);
            if (parked) {
                if (cur == DESC_PARKED) cur = kernargs()->root;
                while (__ballot(cur != DESC_IDLE) & parked) {
                    PTX_NODE_STEP(PTX_QUOT_LONG)
                    PTX_TRI_STEP
                    PTX_EXIT_STEP
                }
                continue;
            }
        }
        if (__ballot(cur != DESC_IDLE) == 0ull) break;
        PTX_CLOCK(ck_refill)

        // ---- step the live lanes until enough of them have finished to make a refill worthwhile
        for (;;) {
#ifdef PTX_SPILL_PROBE
            if (COUNT) pf_iter += lane == 0 ? 1u : 0u;
#else
            if (COUNT) pf_iter += lane == 0 ? 1u : 0u, pf_live += cur != DESC_IDLE ? 1u : 0u;
#endif
#pragma unroll
            for (int round = 0; round < PTX_ROUNDS; ++round) {
            if (round > 0) { PTX_EXIT_STEP }  // (the last round's exit step comes after the triangle step)
#pragma unroll
            for (int node_sub = 0; node_sub < PTX_NODE_UNROLL; ++node_sub)
            PTX_NODE_STEP(PTX_QUOT_SHORT)
            }
            PTX_CLOCK(ck_node)
            const unsigned long long in_leaf = __ballot(cur < -2);
            const unsigned long long can_walk = __ballot(cur >= 0);
            if (in_leaf && ((uint32_t)__popcll(in_leaf) >= (ANY ? TRI_BATCH_ANYHIT : TRI_BATCH_NEAREST) || !can_walk)) {
                if (COUNT) pf_tri_iter += lane == 0 ? 1u : 0u;
#pragma unroll
                for (int tri_sub = 0; tri_sub < PTX_TRI_UNROLL; ++tri_sub)
                PTX_TRI_STEP
            }
            PTX_CLOCK(ck_tri)
            PTX_EXIT_STEP
            PTX_CLOCK(ck_exit)
            const uint32_t live = (uint32_t)__popcll(__ballot(cur != DESC_IDLE));
            if (live == 0) break;
            if (!(windows == 2u && lo >= hi) && live <= 64u - REFILL_IDLE) break;
        }
    }
#undef PTX_CLOCK
#undef PTX_SLOT
#undef PTX_SPILL
#undef PTX_EXIT_STEP
#undef PTX_NODE_STEP
#undef PTX_NODE_BODY
#undef PTX_NODE_FROM_LDS
#undef PTX_TRI_STEP
#undef PTX_WRITE_OUT
    if (COUNT) {
        wave_add(&a.work[0], n_inner);
        wave_add(&a.work[1], n_tris);
        wave_add(&a.prof[0], pf_iter);
        wave_add(&a.prof[1], pf_walk);
        wave_add(&a.prof[2], pf_tri_iter);
        wave_add(&a.prof[3], pf_tri);
        wave_add(&a.prof[4], pf_end_iter);
        wave_add(&a.prof[5], pf_end);
        wave_add(&a.prof[6], pf_refill);
        wave_add(&a.prof[7], pf_live);
        wave_add(&a.prof[8], ck_refill);
        wave_add(&a.prof[9], ck_node);
        wave_add(&a.prof[10], ck_tri);
        wave_add(&a.prof[11], ck_exit);
    }
}

// Image pixel (row-major index into the full image) of the region's local pixel number.  Local numbers run over
// 8 x 8 pixel tiles when the region is a whole number of tiles (tiles_per_row > 0), else along the rows: 64
// consecutive path ids - one wave of camera rays, and what survives of it in the compacted queues of the later
// bounces - then cover a square patch of the image instead of a 64 x 1 strip, so their rays share more of the
// tree.  A pixel's samples do not depend on its number (the sampler is keyed by the image pixel).
__device__ inline uint32_t image_pixel(const RenderState& st, uint32_t local) {
    if (st.tiles_per_row) {
        const uint32_t tile = local >> 6, in = local & 63u;
        const uint32_t ty = tile / st.tiles_per_row, tx = tile - ty * st.tiles_per_row;
        return (st.reg_y + ty * 8u + (in >> 3)) * st.cam.width + st.reg_x + tx * 8u + (in & 7u);
    }
    return (st.reg_y + local / st.reg_w) * st.cam.width + st.reg_x + local % st.reg_w;
}

// sample_ray_for_pixel (pathtracer.rs:113-117) + Pinhole::ray (camera.rs:88-93): the camera ray of path `pid`.
__device__ inline void camera_ray(const RenderState& st, uint32_t pid, float& dx, float& dy, float& dz) {
    const uint32_t pixel = image_pixel(st, pid % st.n_pixels), s = pid / st.n_pixels;
    const uint32_t px = pixel % st.cam.width, py = pixel / st.cam.width;
    const uint64_t key = ptx_path_key(st.seed, pixel, st.first_sample + s);
    const float cx = (float)px + ptx_u01(key, 0), cy = (float)py + ptx_u01(key, 1);
    const float u = cx / (float)st.cam.width, v = cy / (float)st.cam.height;
    dx = st.cam.plane[0] + u * st.cam.dx[0] + v * st.cam.dy[0];
    dy = st.cam.plane[1] + u * st.cam.dx[1] + v * st.cam.dy[1];
    dz = st.cam.plane[2] + u * st.cam.dx[2] + v * st.cam.dy[2];
}

// Starts path `pid` in queue slot `slot` and resets its state.
__device__ inline void start_path(const RenderState& st, ptx_ray* __restrict__ rays, uint32_t slot, uint32_t pid, float dx, float dy, float dz) {
    reinterpret_cast<float4*>(rays)[2 * (size_t)slot] = make_float4(st.cam.position[0], st.cam.position[1], st.cam.position[2], u2f(pid));
    reinterpret_cast<float4*>(rays)[2 * (size_t)slot + 1] = make_float4(dx, dy, dz, u2f(1u));
    // (a path starts with transport (1, 1, 1): k_shade knows that from the bounce number, nothing is stored)
    st.radiance[pid] = Rad{0.0f, 0.0f, 0.0f};
}

// Tops segment `seg` of `rays` (holding `have` entries) up with the next unstarted paths of its range.
// All threads of the workgroup call; returns the new entry count and, in `begun`, the number of paths started (with
// st.cull some of them end at once and take no entry).  lds / parity: the scratch of block_prefix2.
__device__ inline uint32_t regenerate(const RenderState& st, ptx_ray* __restrict__ rays, uint32_t seg, uint32_t have, uint32_t started,
                                      uint32_t* lds, uint32_t& parity, uint32_t& begun) {
    const uint32_t range_lo = seg * st.seg_range;
    const uint32_t range_n = range_lo >= st.n_paths ? 0u : (st.n_paths - range_lo < st.seg_range ? st.n_paths - range_lo : st.seg_range);
    const uint32_t room = st.seg_cap - have, left = range_n - started;
    const uint32_t k = room < left ? room : left;
    begun = k;
    if (threadIdx.x == 0) st.seg_next[seg] = started + k;
    if (!st.cull) {
        for (uint32_t j = threadIdx.x; j < k; j += BLOCK) {
            float dx, dy, dz;
            camera_ray(st, range_lo + started + j, dx, dy, dz);
            start_path(st, rays, seg * st.seg_cap + have + j, range_lo + started + j, dx, dy, dz);
        }
        return have + k;
    }
    uint32_t filled = have;
    for (uint32_t j0 = 0; j0 < k; j0 += BLOCK) {
        const uint32_t j = j0 + threadIdx.x, pid = range_lo + started + j;
        bool enter = false;
        float dx = 0, dy = 0, dz = 0;
        if (j < k) {
            camera_ray(st, pid, dx, dy, dz);
            float t1 = 0.0f, t_end = 3.402823466e+38f;  // the renderer's range (pathtracer.rs:34)
            enter = box_clip(st.cam.position[0], st.cam.position[1], st.cam.position[2], dx, dy, dz, st.box_lo, st.box_hi, t1, t_end);
            if (!enter)  // accumulated_radiance (0) + accumulated_transport (1) * environment
                st.radiance[pid] = Rad{0.0f + 1.0f * st.env[0], 0.0f + 1.0f * st.env[1], 0.0f + 1.0f * st.env[2]};
        }
        uint32_t pre, pre_unused, tot, tot_unused;
        block_prefix2<WAVES>(enter, false, lds, parity, pre, pre_unused, tot, tot_unused);
        parity ^= 1u;
        if (enter) start_path(st, rays, seg * st.seg_cap + filled + pre, pid, dx, dy, dz);
        filled += tot;
    }
    if (threadIdx.x == 0 && have + k != filled) atomicAdd(st.totals + 6, (unsigned long long)(have + k - filled));
    return filled;
}

__global__ __launch_bounds__(BLOCK) void k_start(RenderState st) {
    __shared__ uint32_t lds[4 * WAVES];
    uint32_t parity = 0;
    for (uint32_t seg = blockIdx.x; seg < st.n_seg; seg += gridDim.x) {
        uint32_t begun;
        const uint32_t n = regenerate(st, st.rays[0], seg, 0u, 0u, lds, parity, begun);
        if (threadIdx.x == 0) {
            st.seg_ext[0][seg] = n;
            if (begun) atomicAdd(st.qcounts, begun);  // every camera ray counts, also the ones that ended here
        }
    }
}

struct Bsdf {
    bool is_delta;
    float pdf;
    V3 bsdf, wo;
};

// material.rs:74-82
__device__ inline Bsdf sample_specular(V3 wi, V3 n, V3 color, float probability) {
    const V3 mi = -wi;
    const V3 wo = normalize(mi - 2.0f * dot(mi, n) * n);  // reflect(-wi, n), material.rs:17-19
    return Bsdf{true, probability, color, wo};
}

// material.rs:167-191 with its helpers (:34-38, :84-136) and AlbedoSource::get (albedo.rs:10-28)
__device__ inline Bsdf material_sample(const DevMaterial& m, V3 wi, V3 n, float uvx, float uvy, uint64_t key,
                                       uint32_t dim) {
    const float cos_theta0 = fmaxf(dot(wi, n), 0.0f);
    const float b = 1.0f - cos_theta0;
    const float b2 = b * b;
    const float t5 = b * (b2 * b2);  // powi(5)
    const V3 r0 = v3(m.f0[0], m.f0[1], m.f0[2]);
    const V3 f = r0 + (1.0f - r0) * t5;
    V3 diffuse;
    if (m.texture < 0) {
        diffuse = v3(m.albedo[0], m.albedo[1], m.albedo[2]);
    } else {
        float wx = uvx - floorf(uvx), wy = uvy - floorf(uvy);
        wx = wx == 1.0f ? 0.0f : wx;
        wy = wy == 1.0f ? 0.0f : wy;
        const uint32_t px = (uint32_t)floorf((float)m.tex_w * wx), py = (uint32_t)floorf((float)m.tex_h * wy);
        const float* t = m.tex_rgb + ((size_t)py * m.tex_w + px) * 3;
        diffuse = v3(t[0], t[1], t[2]);
    }
    const V3 td = (1.0f - f) * diffuse;
    const float specular_strength = luminance(f);
    const float diffuse_strength = luminance(td) * (1.0f - m.transmission);
    const float refraction_strength = luminance(td) * m.transmission;
    const float total = specular_strength + diffuse_strength + refraction_strength;
    if (total <= 0.0f) return Bsdf{true, 0.0f, v3(0, 0, 0), n};
    const float p_specular = specular_strength / total;
    const float p_diffuse = diffuse_strength / total;
    const float p_refraction = refraction_strength / total;
#if PTX_FORCE_LOBE  // probe builds only (tools/lobe_probe.sh): every path takes one lobe, so k_shade runs without lobe divergence
    const float r = PTX_FORCE_LOBE == 1 ? 0.0f : (PTX_FORCE_LOBE == 2 ? p_specular : 2.0f);
#else
    const float r = ptx_u01(key, dim);
#endif
    if (p_specular > 0.0f && r < p_specular) return sample_specular(wi, n, f, p_specular);
    if (r < p_specular + p_refraction) {
        // sample_refraction(surface, ior, f, td, p_refraction), material.rs:84-112
        const bool entering = dot(wi, n) < 0.0f;
        const float n1 = entering ? 1.0f : m.ior, n2 = entering ? m.ior : 1.0f;
        const float eta = n1 / n2;
        const V3 normal = entering ? n : -n;
        const V3 incoming = -wi;
        const float ci = dot(incoming, normal);
        const float s2 = eta * eta * (1.0f - ci * ci);
        if (s2 >= 1.0f) return sample_specular(wi, n, f, p_refraction);
        const float ct = sqrtf(1.0f - s2);
        const V3 wo = normalize(eta * incoming + (eta * ci - ct) * normal);
        const float eta_scale = (n2 * n2) / (n1 * n1);
        return Bsdf{true, p_refraction, td * eta_scale, wo};
    }
    if (p_diffuse > 0.0f) {
        // sample_diffuse, material.rs:114-136; perpendicular :21-32; cosine hemisphere sampling.rs:27-48
        const float ax = fabsf(n.x), ay = fabsf(n.y), az = fabsf(n.z);
        V3 perp;
        if (ax < ay && ax < az) perp = v3(0.0f, -n.z, n.y);
        else if (ay < az) perp = v3(-n.z, 0.0f, n.x);
        else perp = v3(-n.y, n.x, 0.0f);
        const V3 tangent = normalize(perp);
        const V3 bitangent = cross(n, tangent);
        const float x = ptx_sym_closed(key, dim + 1), y = ptx_sym_closed(key, dim + 2);
        float hx = 0.0f, hy = 0.0f;
        if (!(x == 0.0f && y == 0.0f)) {
            float rr, theta;
            if (x >= -y && x > y) {
                rr = x;
                theta = y / x;
            } else if (x >= -y) {
                rr = y;
                theta = 2.0f - x / y;
            } else if (x <= y) {
                rr = -x;
                theta = 4.0f + y / x;
            } else {
                rr = -y;
                theta = 6.0f - x / y;
            }
            float sn, cs;
            ptx_sincos(theta * 0.785398163397448309616f, &sn, &cs);
            hx = rr * sn;
            hy = rr * cs;
        }
        const float hz = sqrtf(fmaxf(0.0f, 1.0f - hx * hx - hy * hy));
        const V3 wo = normalize(hx * tangent + hy * bitangent + hz * n);
        const float cos_theta = fmaxf(dot(wo, n), 0.0f);
        const float inv_pi = 0.318309886183790671538f;
        return Bsdf{false, p_diffuse * cos_theta * inv_pi, td * inv_pi, wo};
    }
    return Bsdf{true, 0.0f, v3(0, 0, 0), n};
}

// Material-keyed shade queues (second half; PTX_FLAG_MATERIAL_QUEUES): after the extend launch the
// entries of every queue segment are bucketed by the material of the triangle they hit (bucket 0 = miss, then
// material id + 1, ids >= 6 share the last bucket), stably, and k_shade takes them bucket by bucket: the 64 paths of a
// wave then share a material, hence a lobe set (`Material::sample`, material.rs:167-191) and the texture branch.
// Only the ORDER of shading changes; every path is shaded with its own numbers, so images stay bit-identical.
// Two passes over the segment per workgroup (bucket totals, then positions): ballots per bucket, no atomics.
constexpr int MQ_BUCKETS = 8;
__global__ __launch_bounds__(BLOCK) void k_bucket(DevScene sc, RenderState st, uint32_t q) {
    __shared__ uint32_t s_wave[WAVES][MQ_BUCKETS];
    __shared__ uint32_t s_run[MQ_BUCKETS];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (uint32_t seg = blockIdx.x; seg < st.n_seg; seg += gridDim.x) {
        const uint32_t count = st.seg_ext[q][seg], seg_base = seg * st.seg_cap;
        for (int pass = 0; pass < 2; ++pass) {
            __syncthreads();
            uint32_t first_pos = 0;  // pass 1: totals -> first position of every bucket (read, barrier, then write)
            if (pass == 1 && threadIdx.x < MQ_BUCKETS)
                for (uint32_t b = 0; b < (uint32_t)MQ_BUCKETS; ++b) first_pos += b < threadIdx.x ? s_run[b] : 0u;
            __syncthreads();
            if (threadIdx.x < MQ_BUCKETS) s_run[threadIdx.x] = first_pos;
            __syncthreads();
            for (uint32_t start = 0; start < count; start += BLOCK) {
                const uint32_t k = start + threadIdx.x;
                uint32_t key = MQ_BUCKETS;  // inactive lane
                if (k < count) {
                    const uint32_t idx = reinterpret_cast<const uint4*>(st.hits)[seg_base + k].x;
                    key = 0;
                    if (idx != PTX_MISS) {
                        const uint32_t m = f2u(sc.props[4 * (size_t)idx + 3].w) + 1u;
                        key = m < (uint32_t)MQ_BUCKETS ? m : (uint32_t)MQ_BUCKETS - 1u;
                    }
                }
                uint32_t rank = 0;
                for (uint32_t b = 0; b < (uint32_t)MQ_BUCKETS; ++b) {
                    const unsigned long long m = __ballot(key == b);
                    if (key == b) rank = (uint32_t)__popcll(m & lt);
                    if (lane == b) s_wave[wave][b] = (uint32_t)__popcll(m);
                }
                __syncthreads();
                if (pass == 1 && key < (uint32_t)MQ_BUCKETS) {
                    uint32_t at = s_run[key] + rank;
                    for (uint32_t w = 0; w < wave; ++w) at += s_wave[w][key];
                    st.perm[seg_base + at] = k;
                }
                uint32_t chunk_total = 0;  // every read of s_wave and s_run happens before the barrier, the update after it
                if (threadIdx.x < MQ_BUCKETS)
                    for (int w = 0; w < WAVES; ++w) chunk_total += s_wave[w][threadIdx.x];
                __syncthreads();
                if (threadIdx.x < MQ_BUCKETS) s_run[threadIdx.x] += chunk_total;
                // (the next chunk reads s_run only after its own barrier)
            }
        }
    }
}

// Next-event estimation resolve of one queue segment: accumulated_radiance += accumulated_transport * sum over lights
// of the unoccluded contributions, in MTL light order (pathtracer.rs:68-86).  One path per lane; the occlusion flags
// come from the any-hit traversal launch over the shadow-ray queue.
__device__ inline void resolve_segment(const RenderState& st, uint32_t seg, uint32_t n_lights) {
    for (uint32_t k = threadIdx.x, count = st.seg_sh[seg]; k < count; k += BLOCK) {
        const uint32_t i = seg * st.seg_cap + k;
        const float4 T = ld_stream(st.sh_T + i);
        const uint32_t pid = f2u(T.w);
        if (n_lights == 1u) {  // k_shade<1> stored transport * contribution (sh_index == i with one light)
            if (!st.sh_occ[i]) {
                Rad R = st.radiance[pid];
                R.x = R.x + T.x;
                R.y = R.y + T.y;
                R.z = R.z + T.z;
                st.radiance[pid] = R;
            }
            continue;
        }
        V3 incoming = v3(0.0f, 0.0f, 0.0f);
        for (uint32_t l = 0; l < n_lights; ++l) {
            const size_t j = ((size_t)seg * n_lights + l) * st.seg_cap + k;
            V3 c = v3(0.0f, 0.0f, 0.0f);
            if (!st.sh_occ[j]) {
                const Rad cc = st.sh_contrib[j];
                c = v3(cc.x, cc.y, cc.z);
            }
            incoming = incoming + c;
        }
        Rad R = st.radiance[pid];
        R.x = R.x + T.x * incoming.x;
        R.y = R.y + T.y * incoming.y;
        R.z = R.z + T.z * incoming.z;
        st.radiance[pid] = R;
    }
}

// The resolve of round b runs as the prologue of k_shade(b + 1), segment by segment (a path's radiance is only ever
// touched by its own queue segment, so the order of the additions is the reference's); only the last round's resolve is
// a launch of its own.  Same bytes, but the scattered read-modify-writes of a segment overlap with the streaming loads
// of the workgroups next to it instead of being a kernel of nothing else: shade + resolve -4 ... -8 %, the frame +2.3 %
// on cornell_teapot, +4.5 % on cornell_textured, +1.8 % on sr2 (round 3, profiles/r03_loop_variants.txt 10; VERDICT round
// 2 proposed it, DESIGN round 2 had priced it at < 0.5 %).
#ifndef PTX_RESOLVE_IN_SHADE
#define PTX_RESOLVE_IN_SHADE 1
#endif

// One bounce of trace_ray (pathtracer.rs:33-108) for every entry of the current extend queue.
template <int NL /* compile-time light count, 0 = read it from the scene */>
__global__ __launch_bounds__(BLOCK, PTX_SHADE_WAVES) void k_shade(DevScene sc, RenderState st, uint32_t round /* 1-based */) {
    __shared__ uint32_t lds[4 * WAVES];
    const uint32_t q = (round - 1u) & 1u;
    const ptx_ray* __restrict__ rays_in = st.rays[q];
    ptx_ray* __restrict__ rays_out = st.rays[q ^ 1u];
    const uint32_t n_lights = NL ? (uint32_t)NL : sc.n_lights;
    uint32_t parity = 0;

    // one workgroup per queue segment (see RenderState): read segment `seg`, write the survivors to the front
    // of segment `seg` of the output queues
    for (uint32_t seg = blockIdx.x; seg < st.n_seg; seg += gridDim.x) {
    const uint32_t count = st.seg_ext[q][seg];
    const uint32_t seg_base = seg * st.seg_cap;
    uint32_t n_next = 0, n_shadow = 0;
#if PTX_RESOLVE_IN_SHADE
    if (round > 1u) {  // the light samples of the round before: a path's radiance is only ever touched by its own segment
        resolve_segment(st, seg, n_lights);
        __syncthreads();
    }
#endif
    for (uint32_t start = 0; start < count; start += BLOCK) {
        const bool active = start + threadIdx.x < count;
        // entry of this lane: in queue order, or (material-keyed queues) in the order k_bucket laid down
        const uint32_t i = seg_base + (st.perm && active ? st.perm[seg_base + start + threadIdx.x] : start + threadIdx.x);
        bool emit_next = false, emit_shadow = false;
        uint32_t pid = 0, bounce = 1;
        V3 next_o = v3(0, 0, 0), next_d = v3(0, 0, 0), point_above = v3(0, 0, 0), T_old = v3(0, 0, 0), T_next = v3(0, 0, 0);
        // The lights are sampled AFTER the compaction barrier, one at a time, straight into the shadow queue: only what a
        // sample needs - the point, the normal, the bsdf: 9 registers - crosses the barrier, not 6 registers per light
        // (round 3: k_shade<4> 111 -> 76 VGPRs, 4 -> 6 waves per SIMD; sr2 3840x2160 shade -36 %, frame +18 %;
        // cubesplosion shade -10 %; one light -1.5 %; profiles/r03_loop_variants.txt).
        V3 sh_point = v3(0, 0, 0), sh_n = v3(0, 0, 0), sh_bsdf = v3(0, 0, 0);
        // Light::sample_shadow_ray (light.rs:34-37,69-76), uniform_sample_unit_sphere (sampling.rs:9-17);
        // emitted(&point) (light.rs:17-19), |wo . n| with wo = normalize(shadow dir) (pathtracer.rs:79-81)
        auto light_sample = [&](uint32_t l, uint64_t key, uint32_t base, V3 point, V3 above, V3 n, V3 bsdf, V3& dir, V3& c) {
            const ptx_light& L = sc.lights[l];
            const float z = ptx_sym_open(key, base + 3u + 2u * l);
            const float a = ptx_angle(key, base + 4u + 2u * l);
            const float r = sqrtf(1.0f - z * z);
            float sn, cs;
            ptx_sincos(a, &sn, &cs);
            const V3 center = v3(L.center[0], L.center[1], L.center[2]);
            const V3 target = center + v3(r * cs, r * sn, z) * L.radius;
            dir = target - above;
            const V3 cp = center - point;
            const V3 emitted = v3(L.intensity[0], L.intensity[1], L.intensity[2]) / dot(cp, cp);
            const V3 wo = normalize(dir);
            c = bsdf * emitted * fabsf(dot(wo, n));
        };
        if (active) {
            const float4 ra = ld_stream(reinterpret_cast<const float4*>(rays_in) + 2 * (size_t)i);
            const float4 rb = ld_stream(reinterpret_cast<const float4*>(rays_in) + 2 * (size_t)i + 1);
            pid = f2u(ra.w);
            bounce = f2u(rb.w);  // 1-based: the bounce this ray has just traced
#ifdef PTX_SHADE_DUMMY  // probe: what do vector instructions cost this kernel?  (tools/r3_variants.sh)
            {
                float x = ra.x;
#pragma unroll
                for (int k = 0; k < PTX_SHADE_DUMMY; ++k) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
                if (x == 123.456f) bounce += 1u;
            }
#endif
            const uint32_t base = ptx_dim_base(bounce, n_lights);
            const uint4 hraw = ld_stream(reinterpret_cast<const uint4*>(st.hits) + i);
            // accumulated_transport = Vec3::ONE at the first bounce (pathtracer.rs:31): not stored by start_path
            V3 T = v3(1.0f, 1.0f, 1.0f);
            if (bounce != 1u) {
                ld_stream3(&st.transq[q][i].x, T.x, T.y, T.z);
            }
            if (hraw.x == PTX_MISS) {
                // return accumulated_radiance + accumulated_transport * environment (pathtracer.rs:44-46)
                Rad R = st.radiance[pid];
                R.x = R.x + T.x * sc.env[0];
                R.y = R.y + T.y * sc.env[1];
                R.z = R.z + T.z * sc.env[2];
                st.radiance[pid] = R;
            } else {
                const float ht = u2f(hraw.y), hu = u2f(hraw.z), hv = u2f(hraw.w);
                const V3 ro = v3(ra.x, ra.y, ra.z), rd = v3(rb.x, rb.y, rb.z);
                const float4 p0 = sc.props[4 * (size_t)hraw.x], p1 = sc.props[4 * (size_t)hraw.x + 1];
                const float4 p2 = sc.props[4 * (size_t)hraw.x + 2], p3 = sc.props[4 * (size_t)hraw.x + 3];
                const V3 n0 = v3(p0.x, p0.y, p0.z), n1 = v3(p0.w, p1.x, p1.y), n2 = v3(p1.z, p1.w, p2.x);
                const float w = 1.0f - (hu + hv);
                // TriangleNormals::lerp / TriangleTexcoords::lerp (geometry/src/triangle.rs:144-160)
                const V3 n = normalize(w * n0 + hu * n1 + hv * n2);
                const float uvx = w * p2.y + hu * p2.w + hv * p3.y;
                const float uvy = w * p2.z + hu * p3.x + hv * p3.z;
                const DevMaterial& mat = sc.materials[f2u(p3.w)];
                const V3 wi = -rd;
                const V3 offset = 0.00001f * n;
                const V3 point = ro + ht * rd;
                point_above = point + offset;
                const V3 point_below = point - offset;
                const uint32_t pixel = image_pixel(st, pid % st.n_pixels), s = pid / st.n_pixels;
                const uint64_t key = ptx_path_key(st.seed, pixel, st.first_sample + s);
                const Bsdf smp = material_sample(mat, wi, n, uvx, uvy, key, base);

                if (!smp.is_delta && n_lights > 0) {
                    emit_shadow = true;
                    T_old = T;
                    sh_point = point, sh_n = n, sh_bsdf = smp.bsdf;
                }
                if (smp.pdf != 0.0f) {
                    const float cosine = dot(smp.wo, n);
                    V3 Tn;
                    if (smp.is_delta) Tn = T * smp.bsdf;
                    else Tn = T * (smp.bsdf * (fabsf(cosine) / smp.pdf));
                    if (!(dot(Tn, Tn) <= 1.0e-4f) && bounce < sc.max_bounces) {
                        emit_next = true;
                        T_next = Tn;
                        next_o = cosine >= 0.0f ? point_above : point_below;
                        next_d = smp.wo;
                    }
                }
            }
        }
        // one compaction pass for both queues: wave ballots + LDS prefix over the waves, no atomics
        uint32_t pre_n, pre_s, tot_n, tot_s;
        block_prefix2<WAVES>(emit_next, emit_shadow, lds, parity, pre_n, pre_s, tot_n, tot_s);
        parity ^= 1u;
        const uint32_t slot_n = seg_base + n_next + pre_n, slot_s = seg_base + n_shadow + pre_s;
        n_next += tot_n;
        n_shadow += tot_s;
        if (emit_next) {
            st_stream(reinterpret_cast<float4*>(rays_out) + 2 * (size_t)slot_n, make_float4(next_o.x, next_o.y, next_o.z, u2f(pid)));
            st_stream(reinterpret_cast<float4*>(rays_out) + 2 * (size_t)slot_n + 1, make_float4(next_d.x, next_d.y, next_d.z, u2f(bounce + 1u)));
            st_stream3(&st.transq[q ^ 1u][slot_n].x, T_next.x, T_next.y, T_next.z);
        }
        if (emit_shadow) {
            // one light: the record is the product k_resolve would form (0 + c == c, so T * (0 + c) == T * c
            // bit for bit) and no separate contribution array is written
            if (NL != 1) st_stream(st.sh_T + slot_s, make_float4(T_old.x, T_old.y, T_old.z, u2f(pid)));
            if (st.sh_bounce) st.sh_bounce[slot_s] = (uint8_t)bounce;
            // (the sampler key of the path again: cheaper than carrying it across the barrier)
            const uint64_t key = ptx_path_key(st.seed, image_pixel(st, pid % st.n_pixels), st.first_sample + pid / st.n_pixels);
            const uint32_t base = ptx_dim_base(bounce, n_lights);
            V3 c0 = v3(0, 0, 0);  // one light: the contribution goes into the record below
            for (uint32_t l = 0; l < (NL ? (uint32_t)NL : n_lights); ++l) {
                const size_t j = ((size_t)seg * n_lights + l) * st.seg_cap + (slot_s - seg_base);  // light-major (RenderState)
                V3 dir, c;
                light_sample(l, key, base, sh_point, point_above, sh_n, sh_bsdf, dir, c);
                if (NL == 1) c0 = c;
                st_stream(reinterpret_cast<float4*>(st.sh_rays) + 2 * j, make_float4(point_above.x, point_above.y, point_above.z, 0.0f));
                st_stream(reinterpret_cast<float4*>(st.sh_rays) + 2 * j + 1, make_float4(dir.x, dir.y, dir.z, 1.0f));
                if (NL != 1) st_stream3(&st.sh_contrib[j].x, c.x, c.y, c.z);
            }
            if (NL == 1) st_stream(st.sh_T + slot_s, make_float4(T_old.x * c0.x, T_old.y * c0.y, T_old.z * c0.z, u2f(pid)));
        }
    }
    // (a batch with more paths than queue slots tops the segment up with fresh camera rays next: k_regenerate)
    if (threadIdx.x == 0) {
        st.seg_ext[q ^ 1u][seg] = n_next;
        st.seg_sh[seg] = n_shadow;
        if (n_next | n_shadow)
            atomicAdd(reinterpret_cast<unsigned long long*>(st.qcounts + 2u * round),
                      (unsigned long long)n_next | ((unsigned long long)n_shadow << 32));
    }
    }
}

// Path regeneration (a batch with more paths than queue slots, see RenderState): after k_shade(round) every segment
// starts the next paths of its range in the slots that have become free.  A kernel of its own, launched only for such
// batches: inside k_shade the camera-ray code cost that kernel ten registers and a wave per SIMD.
__global__ __launch_bounds__(BLOCK) void k_regenerate(RenderState st, uint32_t round /* 1-based: the round k_shade has just run */) {
    __shared__ uint32_t lds[4 * WAVES];
    uint32_t parity = 0;
    const uint32_t q = round & 1u;  // the queue k_shade(round) wrote
    for (uint32_t seg = blockIdx.x; seg < st.n_seg; seg += gridDim.x) {
        const uint32_t have = st.seg_ext[q][seg], started = st.seg_next[seg];
        __syncthreads();  // everybody has read the two before thread 0 rewrites them
        uint32_t begun;
        const uint32_t n = regenerate(st, st.rays[q], seg, have, started, lds, parity, begun);
        if (threadIdx.x == 0) {
            st.seg_ext[q][seg] = n;
            if (begun) atomicAdd(st.qcounts + 2u * round, begun);  // extend rays of the next round (also the ones that ended at once)
        }
    }
}

__global__ __launch_bounds__(BLOCK) void k_resolve(RenderState st, uint32_t bounce, uint32_t n_lights) {
    (void)bounce;
    for (uint32_t seg = blockIdx.x; seg < st.n_seg; seg += gridDim.x) resolve_segment(st, seg, n_lights);
}

// Checker (PTX_FLAG_CHECK_BRUTE): every ray of a queue again, against ALL triangles in index order
// (`intersect_closest_geometry`, geometry/src/geometry.rs:176-196 — what kdtree-tester-cli calls the reference
// intersection, ray_bouncer.rs:35-42), compared with what the KD-tree traversal answered under the tolerances
// of `CheckedIntersection::is_valid` (kdtree-tester-cli/src/checked_intersection.rs:10-25).  Disagreements are
// appended to a fail list.  One workgroup per queue segment; the triangles pass through LDS in tiles.
struct FailRec {
    float o[3], d[3];
    float t_ref, t_kd;
    uint32_t flags;  // bit 0: the reference hit something, bit 1: the kd-tree did, bit 2: shadow ray
    uint32_t pad;
};
struct CheckArgs {
    const TriRec* tris;
    uint32_t n_tris;
    const ptx_ray* rays;
    const uint32_t* seg_counts;
    uint32_t n_seg, seg_cap, scale, div;  // segment s holds seg_counts[s / div] * scale rays
    uint32_t internal;      // extend queue: t in [0, f32::MAX]
    const ptx_hit* hits;    // nearest-hit answers, or
    const uint8_t* occ;     // occlusion answers
    FailRec* fails;
    uint32_t fail_cap;
    unsigned long long* counters;  // [0] rays checked, [1] fails found
};
constexpr int CHECK_TILE = 256;

__global__ __launch_bounds__(BLOCK) void k_check(CheckArgs c) {
    __shared__ TriRec s_tris[CHECK_TILE];
    for (uint32_t seg = blockIdx.x; seg < c.n_seg; seg += gridDim.x) {
        const uint32_t count = c.seg_counts[seg / c.div] * c.scale;
        const size_t base = (size_t)seg * c.seg_cap * c.scale;
        for (uint32_t start = 0; start < count; start += BLOCK) {
            const bool active = start + threadIdx.x < count;
            const size_t i = base + start + threadIdx.x;
            float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, t1 = 0, t2 = 0;
            if (active) {
                const float4 ra = reinterpret_cast<const float4*>(c.rays)[2 * i];
                const float4 rb = reinterpret_cast<const float4*>(c.rays)[2 * i + 1];
                ox = ra.x, oy = ra.y, oz = ra.z, t1 = c.internal ? 0.0f : ra.w;
                dx = rb.x, dy = rb.y, dz = rb.z, t2 = c.internal ? 3.402823466e+38f : rb.w;
            }
            uint32_t bi = PTX_MISS;
            float bt = 0.0f, bu = 0.0f, bv = 0.0f;
            for (uint32_t tile = 0; tile < c.n_tris; tile += CHECK_TILE) {
                __syncthreads();
                for (uint32_t k = threadIdx.x; k < CHECK_TILE && tile + k < c.n_tris; k += BLOCK) s_tris[k] = c.tris[tile + k];
                __syncthreads();
                const uint32_t n = c.n_tris - tile < (uint32_t)CHECK_TILE ? c.n_tris - tile : (uint32_t)CHECK_TILE;
                if (active) {
                    for (uint32_t k = 0; k < n; ++k) {
                        float t, u, v;
                        if (tri_test(s_tris[k], ox, oy, oz, dx, dy, dz, t, u, v) && t1 <= t && t <= t2) {
                            if (bi == PTX_MISS || !(bt <= t)) bt = t, bu = u, bv = v, bi = tile + k;  // IndexedIntersection::min
                        }
                    }
                }
            }
            if (active) {
                bool kd_some, valid;
                float t_kd = 0.0f;
                if (c.hits) {
                    const uint4 h = reinterpret_cast<const uint4*>(c.hits)[i];
                    kd_some = h.x != PTX_MISS;
                    t_kd = u2f(h.y);
                    valid = (!kd_some && bi == PTX_MISS) ||
                            (kd_some && bi != PTX_MISS && h.x == bi && fabsf(bt - t_kd) < 0.000001f &&
                             fabsf(bu - u2f(h.z)) < 0.00001f && fabsf(bv - u2f(h.w)) < 0.00001f);
                } else {
                    kd_some = c.occ[i] != 0;  // the shadow query only knows yes / no
                    valid = kd_some == (bi != PTX_MISS);
                }
                if (!valid) {
                    const unsigned long long at = atomicAdd(&c.counters[1], 1ull);
                    if (at < c.fail_cap) {
                        FailRec r;
                        r.o[0] = ox, r.o[1] = oy, r.o[2] = oz, r.d[0] = dx, r.d[1] = dy, r.d[2] = dz;
                        r.t_ref = bt, r.t_kd = t_kd;
                        r.flags = (bi != PTX_MISS ? 1u : 0u) | (kd_some ? 2u : 0u) | (c.hits ? 0u : 4u);
                        r.pad = 0;
                        c.fails[at] = r;
                    }
                }
            }
        }
        if (threadIdx.x == 0 && count) atomicAdd(&c.counters[0], (unsigned long long)count);
    }
}

// Ray log (tracing/src/raylogger.rs:39-66, filled from pathtracer.rs:34-43 and 69-71 when the reference is built
// with its `ray_logging` feature): one record per extend ray - stretched to its hit point, `ray.extended(t)`,
// when it hit - and per shadow ray.  Records are appended in queue order behind one counter (a wave reserves
// its records with one atomic); `key` holds every integer field of the reference's record in the order the
// reference writes them, so sorting by it on the host gives the reference's file order.
struct LogRec {
    unsigned long long key;  // iteration:16 | pixel_y:16 | pixel_x:16 | bounce:8 | shadow:1 | light:4 | intersect:1
    float o[3], d[3];
};
struct LogArgs {
    LogRec* recs;
    unsigned long long* count;  // records produced so far (may pass cap: the excess is dropped)
    unsigned long long cap;
};

template <bool SHADOW>
__global__ __launch_bounds__(BLOCK) void k_log(RenderState st, LogArgs lg, uint32_t q, uint32_t n_lights) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t per = SHADOW ? n_lights : 1u;
    for (uint32_t seg = blockIdx.x; seg < st.n_seg; seg += gridDim.x) {
        const uint32_t count = (SHADOW ? st.seg_sh[seg] : st.seg_ext[q][seg]) * per;
        for (uint32_t start = 0; start < count; start += BLOCK) {
            const uint32_t k = start + threadIdx.x;
            const bool active = k < count;
            LogRec r{};
            if (active) {
                const uint32_t slot = seg * st.seg_cap + k / per, light = k % per;
                uint32_t pid, bounce, intersect;
                float4 ra, rb;
                if (SHADOW) {
                    const size_t j = ((size_t)seg * n_lights + light) * st.seg_cap + k / per;
                    ra = reinterpret_cast<const float4*>(st.sh_rays)[2 * j];
                    rb = reinterpret_cast<const float4*>(st.sh_rays)[2 * j + 1];
                    pid = f2u(st.sh_T[slot].w);
                    bounce = st.sh_bounce[slot];
                    intersect = st.sh_occ[j];
                } else {
                    ra = reinterpret_cast<const float4*>(st.rays[q])[2 * (size_t)slot];
                    rb = reinterpret_cast<const float4*>(st.rays[q])[2 * (size_t)slot + 1];
                    pid = f2u(ra.w);
                    bounce = f2u(rb.w);
                    const uint4 h = reinterpret_cast<const uint4*>(st.hits)[slot];
                    intersect = h.x != PTX_MISS ? 1u : 0u;
                    if (intersect) {  // Ray::extended (geometry/src/ray.rs:21-26): t * direction
                        const float t = u2f(h.y);
                        rb.x = t * rb.x, rb.y = t * rb.y, rb.z = t * rb.z;
                    }
                }
                const uint32_t pixel = image_pixel(st, pid % st.n_pixels), s = pid / st.n_pixels;
                const unsigned long long it = (st.first_sample + s) & 0xFFFFu;
                const unsigned long long px = (pixel % st.cam.width) & 0xFFFFu, py = (pixel / st.cam.width) & 0xFFFFu;
                r.key = it << 46 | py << 30 | px << 14 | (unsigned long long)(bounce & 0xFFu) << 6 |
                        (SHADOW ? 1ull : 0ull) << 5 | (unsigned long long)(light & 0xFu) << 1 | intersect;
                r.o[0] = ra.x, r.o[1] = ra.y, r.o[2] = ra.z;
                r.d[0] = rb.x, r.d[1] = rb.y, r.d[2] = rb.z;
            }
            const unsigned long long mask = __ballot(active);
            if (mask) {
                unsigned long long base = 0;
                if (lane == (uint32_t)__ffsll((long long)mask) - 1u) base = atomicAdd(lg.count, (unsigned long long)__popcll(mask));
                base = __shfl(base, __ffsll((long long)mask) - 1);
                const unsigned long long at = base + (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull));
                if (active && at < lg.cap) lg.recs[at] = r;
            }
        }
    }
}

// `*value += render_pixel(..)` (pathtracer.rs:138-140), samples added in index order.
__global__ __launch_bounds__(BLOCK) void k_accumulate(RenderState st, float* __restrict__ accum) {
    for (uint32_t local = blockIdx.x * BLOCK + threadIdx.x; local < st.n_pixels; local += gridDim.x * BLOCK) {
        const uint32_t pixel = image_pixel(st, local);
        float r = accum[3 * (size_t)pixel], g = accum[3 * (size_t)pixel + 1], b = accum[3 * (size_t)pixel + 2];
        for (uint32_t s = 0; s < st.n_samples; ++s) {
            const Rad v = st.radiance[(size_t)s * st.n_pixels + local];
            r += v.x;
            g += v.y;
            b += v.z;
        }
        accum[3 * (size_t)pixel] = r;
        accum[3 * (size_t)pixel + 1] = g;
        accum[3 * (size_t)pixel + 2] = b;
    }
}

// `ImageBuffer::to_rgb8` (image_buffer.rs:29-38) on the device: gamma 1/2.2, clamp, round, saturate; NaN -> 255 (Rust's f32::min(NaN, 1.0) is 1.0).  The
// power is taken in f64 and rounded once to f32, which is what a correctly rounded powf gives (the host entry point
// uses libm's powf): 4 channel values per lane, one dword store.
__global__ __launch_bounds__(256) void k_tonemap(const float* __restrict__ accum, size_t n, float inv, uint8_t* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        uint32_t packed = 0;
        const uint32_t m = n - i < 4 ? (uint32_t)(n - i) : 4u;
        for (uint32_t k = 0; k < m; ++k) {
            float g = (float)pow((double)(accum[i + k] * inv), (double)(1.0f / 2.2f));
            g = g < 1.0f ? g : 1.0f;
            const float c = roundf(g * 255.0f);
            const uint32_t b = !(c == c) ? 0u : (c <= 0.0f ? 0u : (c >= 255.0f ? 255u : (uint32_t)c));
            packed |= b << (8u * k);
        }
        if (m == 4 && (reinterpret_cast<uintptr_t>(out + i) & 3u) == 0) *reinterpret_cast<uint32_t*>(out + i) = packed;
        else for (uint32_t k = 0; k < m; ++k) out[i + k] = (uint8_t)(packed >> (8u * k));
    }
}

__global__ void k_tally(RenderState st, uint32_t rounds, uint32_t n_lights) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned long long e = 0, s = 0;
        for (uint32_t b = 0; b < rounds; ++b) {
            e += st.qcounts[2u * b];
            s += (unsigned long long)st.qcounts[2u * b + 3u] * n_lights;
        }
        st.totals[0] += e;
        st.totals[1] += s;
    }
}

// ------------------------------------------------------------------------------------------ host side
template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    hipError_t ensure(size_t count) {
        if (count <= n) return hipSuccess;
        release();
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    hipError_t upload(const T* src, size_t count) {
        hipError_t e = ensure(count ? count : 1);
        if (e != hipSuccess || !count) return e;
        return hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice);
    }
};

int hip_fail(hipError_t e, const char* where) {
    int code = e == hipErrorOutOfMemory ? PTX_E_OOM : PTX_E_HIP;
    return fail(code, std::string(where) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                        \
    do {                                                     \
        hipError_t e__ = (expr);                             \
        if (e__ != hipSuccess) return hip_fail(e__, #expr);  \
    } while (0)

enum KernelClass { KC_GENERATE = 0, KC_EXTEND, KC_SHADE, KC_SHADOW, KC_OTHER, KC_COUNT };

}  // namespace

struct ptx_scene {
    int device = 0;
    uint32_t flags = 0;
    uint32_t n_triangles = 0, n_nodes = 0;
    bool shading = false;
    DevBuf<ptx_kdnode> nodes;
    DevBuf<uint4> dev_nodes;     // traversal layout (see k_traverse)
    DevBuf<TriRec> leaf_tris;   // triangle records in leaf order, index + last flag in f[10] (see k_traverse)
    int32_t root_desc = DESC_EXIT;
    uint32_t top_n = 0;  // inner nodes served from LDS
    bool clip_box = false;
    float box_lo[3] = {0, 0, 0}, box_hi[3] = {0, 0, 0};  // bounding box of the triangles, widened
    bool long_div = false;  // a split distance outside split_quotient()'s window (or PTX_LONG_DIV=1)
    int tgrid = 1024;  // traversal workgroups (TBLOCK threads each)
    DevBuf<uint32_t> leaf_idx;
    DevBuf<TriRec> tris;
    DevBuf<float4> props;
    DevBuf<DevMaterial> materials;
    std::vector<std::unique_ptr<DevBuf<float>>> textures;
    DevScene dev{};
    int grid = 2048;
    int shade_slots = 2048;  // k_shade workgroups resident at once: sized by the occupancy that kernel really gets
    int seg_rounds = 1;
    int shade_grid = 2048;   // most queue segments a batch may have (shade_slots x seg_rounds)

    // render state (allocated on first render, grown on demand)
    uint32_t batch_samples = 0;  // 0 = choose from the image size
    uint64_t slot_budget = 0;    // queue slots over all pipelines (ptx_set_queue_slots, PTX_SLOTS_MI = Mi slots); 0 = from auto_bytes
    uint64_t auto_bytes = 0;     // fallback for auto_queue_bytes(): a share of the memory that was free at create time
    size_t cap_slots = 0, cap_paths = 0;
    uint32_t cap_rounds = 0, last_rounds = 0;
    // One pipeline = the queues and state of one sample batch in flight on its own stream.  Batches are
    // independent until their accumulate step, so several run concurrently: the long-ray tails of the
    // late bounces of one batch (a few waves busy, the rest of the chip idle) overlap with the wide early
    // bounces of another.
    struct Pipe {
        hipStream_t stream = nullptr;
        // the shadow rays of bounce b and the extend rays of bounce b+1 only depend on k_shade(b): they are
        // traversed concurrently, the shadow launch (and its resolve) on this second stream, so that the
        // long-ray tail of one launch (~0.3 ms with the GPU mostly idle) is covered by the other
        hipStream_t stream2 = nullptr;
        hipEvent_t ev_shade = nullptr, ev_resolve = nullptr;
        DevBuf<uint2> spill2;
        hipEvent_t ev_accum = nullptr, ev_join = nullptr;
        DevBuf<ptx_ray> rays[2];
        DevBuf<ptx_hit> hits;
        DevBuf<Rad> transq[2];
        DevBuf<float4> sh_T;
        DevBuf<Rad> sh_contrib;
        DevBuf<Rad> radiance;
        DevBuf<ptx_ray> sh_rays;
        DevBuf<uint8_t> sh_occ, sh_bounce;
        DevBuf<uint32_t> perm;  // PTX_FLAG_MATERIAL_QUEUES
        DevBuf<uint2> spill;
        DevBuf<uint32_t> counts, segs;
        DevBuf<uint32_t> plans;  // [2][plan_words]: window plans of the next extend and the next shadow launch (k_plan)
        ~Pipe() {
            if (ev_accum) (void)hipEventDestroy(ev_accum);
            if (ev_join) (void)hipEventDestroy(ev_join);
            if (ev_shade) (void)hipEventDestroy(ev_shade);
            if (ev_resolve) (void)hipEventDestroy(ev_resolve);
            if (stream2) (void)hipStreamDestroy(stream2);
            if (stream) (void)hipStreamDestroy(stream);
        }
    };
    std::vector<std::unique_ptr<Pipe>> pipes;
    uint32_t n_pipes = 1;
    bool overlap_shadow = false;  // PTX_OVERLAP=1: shadow(b) and extend(b+1) on two streams (measured: +0.9 %, and it blurs the per-kernel timings)
    int last_pipe = 0;
    hipEvent_t ev_fork = nullptr;
    DevBuf<uint2> spill;  // of the *_batch entry points
    DevBuf<uint32_t> batch_cursor;
    // The *_batch_device entry points share that one cursor and spill area, so their launches are serialised:
    // every call waits for the event its predecessor recorded, whatever streams the two were given.
    hipEvent_t ev_batch = nullptr;
    bool batch_pending = false;
    unsigned char* noalloc_scratch = nullptr;  // this device's copy of `scratch` (hipGetSymbolAddress at create time)
    // ptx_render_multi (ptx_multi.hip): this shard's accumulation buffer, a landing buffer for a peer's, a stream
    DevBuf<float> multi_accum, multi_tmp;
    hipStream_t multi_stream = nullptr;
    DevBuf<unsigned long long> totals;
    DevBuf<FailRec> fails;                   // PTX_FLAG_CHECK_BRUTE
    DevBuf<unsigned long long> check_count;  // [2] rays checked, fails found
    static constexpr uint32_t FAIL_CAP = 1u << 16;
    DevBuf<LogRec> raylog;                    // ptx_raylog_enable
    DevBuf<unsigned long long> raylog_count;  // [1] records produced
    uint64_t raylog_cap = 0;                  // 0 = logging off
    uint64_t samples_done = 0;
    uint64_t batch_extend_rays = 0, batch_shadow_rays = 0;  // rays of the *_batch entry points
    uint64_t extend_launches = 0, shadow_launches = 0;

    // kernel timing (PTX_FLAG_TIME_KERNELS)
    struct Stamp {
        hipEvent_t a, b;
        int cls;
        uint32_t round;  // 1-based round of the batch for the traversal launches of a render, else 0
    };
    std::vector<double> round_ms[2];  // [0] extend, [1] shadow: kernel time per round since the last reset
    std::vector<Stamp> stamps;
    std::vector<hipEvent_t> free_events;
    double class_ms[KC_COUNT] = {0, 0, 0, 0, 0};
    double total_ms = 0;

    // staging for ptx_intersect_batch
    DevBuf<ptx_ray> stage_rays;
    DevBuf<ptx_hit> stage_hits;
    DevBuf<float> stage_accum;

    ~ptx_scene() {
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_batch) (void)hipEventDestroy(ev_batch);
        if (multi_stream) (void)hipStreamDestroy(multi_stream);
        for (auto& s : stamps) {
            (void)hipEventDestroy(s.a);
            (void)hipEventDestroy(s.b);
        }
        for (auto e : free_events) (void)hipEventDestroy(e);
    }
};

namespace {

hipEvent_t take_event(ptx_scene* sc) {
    if (!sc->free_events.empty()) {
        hipEvent_t e = sc->free_events.back();
        sc->free_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;  // the launch this would have timed simply goes untimed (see Timed)
    }
    return e;
}

struct Timed {  // RAII bracket of one launch with two events when timing is on
    ptx_scene* sc;
    hipStream_t stream;
    ptx_scene::Stamp st{};
    bool on;
    Timed(ptx_scene* s, hipStream_t str, int cls, uint32_t round = 0) : sc(s), stream(str), on((s->flags & PTX_FLAG_TIME_KERNELS) != 0) {
        if (on) {
            st.a = take_event(sc);
            st.b = take_event(sc);
            st.cls = cls;
            st.round = round;
            if (!st.a || !st.b) {  // no event to be had: this launch is not timed
                if (st.a) sc->free_events.push_back(st.a);
                if (st.b) sc->free_events.push_back(st.b);
                on = false;
            } else {
                (void)hipEventRecord(st.a, stream);
            }
        }
    }
    ~Timed() {
        if (on) {
            (void)hipEventRecord(st.b, stream);
            sc->stamps.push_back(st);
        }
    }
};

void book_stamp(ptx_scene* sc, const ptx_scene::Stamp& s, float ms) {
    sc->class_ms[s.cls] += ms;
    if (s.round && (s.cls == KC_EXTEND || s.cls == KC_SHADOW)) {
        std::vector<double>& v = sc->round_ms[s.cls == KC_EXTEND ? 0 : 1];
        if (v.size() < s.round) v.resize(s.round, 0.0);
        v[s.round - 1] += ms;
    }
}

// Without anybody asking for ptx_stats a long-running render loop would pile up stamps (two events per launch):
// past a threshold the finished ones are folded into the class totals and their events recycled.
void clear_stale_stamps(ptx_scene* sc) {
    if (sc->stamps.size() < 4096) return;
    size_t kept = 0;
    for (auto& s : sc->stamps) {
        float ms = 0.0f;
        if (hipEventQuery(s.b) == hipSuccess) {
            if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) book_stamp(sc, s, ms);
            sc->free_events.push_back(s.a);
            sc->free_events.push_back(s.b);
        } else {
            sc->stamps[kept++] = s;
        }
    }
    sc->stamps.resize(kept);
}

void resolve_stamps(ptx_scene* sc) {
    for (auto& s : sc->stamps) {
        float ms = 0.0f;
        if (hipEventSynchronize(s.b) == hipSuccess && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) book_stamp(sc, s, ms);
        sc->free_events.push_back(s.a);
        sc->free_events.push_back(s.b);
    }
    sc->stamps.clear();
}

// Words of one window plan: the most segments a traversal launch of this scene can have is one per k_shade
// workgroup and light.
size_t plan_words(const ptx_scene* sc) {
    return PLAN_HEADER + 3 * ((size_t)sc->shade_grid * (sc->dev.n_lights ? sc->dev.n_lights : 1u) + 1);
}

int grow_render_state(ptx_scene* sc, size_t slots, size_t paths, uint32_t rounds) {
    const uint32_t L = sc->dev.n_lights ? sc->dev.n_lights : 1u;
    if (!sc->ev_fork) HIP_TRY(hipEventCreateWithFlags(&sc->ev_fork, hipEventDisableTiming));
    sc->pipes.resize(sc->n_pipes);
    sc->last_pipe = 0;
    for (auto& up : sc->pipes) {
        if (!up) {
            up = std::make_unique<ptx_scene::Pipe>();
            HIP_TRY(hipStreamCreateWithFlags(&up->stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&up->ev_accum, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&up->ev_join, hipEventDisableTiming));
            HIP_TRY(up->spill.ensure((size_t)(PTX_MAX_DEPTH - LDS_DEPTH) * sc->tgrid * TBLOCK));
            HIP_TRY(hipStreamCreateWithFlags(&up->stream2, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&up->ev_shade, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&up->ev_resolve, hipEventDisableTiming));
            HIP_TRY(up->spill2.ensure((size_t)(PTX_MAX_DEPTH - LDS_DEPTH) * sc->tgrid * TBLOCK));
        }
        ptx_scene::Pipe& p = *up;
        // per slot: 2 x 32 rays + 2 x 12 transport + 16 hit + 16 shadow record + 1 bounce, and per light
        // 32 shadow ray + 1 occlusion flag (+ 12 contribution when there are several lights)
        for (int q = 0; q < 2; ++q) {
            HIP_TRY(p.rays[q].ensure(slots));
            HIP_TRY(p.transq[q].ensure(slots));
        }
        HIP_TRY(p.hits.ensure(slots));
        HIP_TRY(p.radiance.ensure(paths));
        HIP_TRY(p.sh_T.ensure(slots));
        HIP_TRY(p.sh_rays.ensure(slots * L));
        if (L > 1) HIP_TRY(p.sh_contrib.ensure(slots * L));
        HIP_TRY(p.sh_occ.ensure(slots * L));
        HIP_TRY(p.sh_bounce.ensure(slots));
        if (sc->flags & PTX_FLAG_MATERIAL_QUEUES) HIP_TRY(p.perm.ensure(slots));
        HIP_TRY(p.counts.ensure((2 + 2 * CURSOR_WORDS) * (size_t)rounds + 2));
        HIP_TRY(p.segs.ensure(4 * (size_t)sc->shade_grid));
        HIP_TRY(p.plans.ensure(2 * plan_words(sc)));
    }
    if (!sc->totals.p) {
        HIP_TRY(sc->totals.ensure(32));
        HIP_TRY(hipMemset(sc->totals.p, 0, 32 * sizeof(unsigned long long)));
    }
    return PTX_OK;
}

// slots = queue entries per pipeline (n_seg x seg_cap), paths = path ids per batch (radiance),
// rounds = extend/shade/shadow rounds per batch (bookkeeping words).  On failure every pipeline is released
// (the caller may try again with less).
int ensure_render_state(ptx_scene* sc, size_t slots, size_t paths, uint32_t rounds) {
    if (sc->pipes.size() == sc->n_pipes && slots <= sc->cap_slots && paths <= sc->cap_paths && rounds <= sc->cap_rounds)
        return PTX_OK;
    HIP_TRY(hipDeviceSynchronize());
    // every pipeline, old or new, holds the high-water marks
    if (slots < sc->cap_slots) slots = sc->cap_slots;
    if (paths < sc->cap_paths) paths = sc->cap_paths;
    if (rounds < sc->cap_rounds) rounds = sc->cap_rounds;
    const int rc = grow_render_state(sc, slots, paths, rounds);
    if (rc != PTX_OK) {
        sc->pipes.clear();
        sc->cap_slots = sc->cap_paths = 0;
        sc->cap_rounds = 0;
        return rc;
    }
    sc->cap_slots = slots;
    sc->cap_paths = paths;
    sc->cap_rounds = rounds;
    return PTX_OK;
}

int check_scene(ptx_scene* sc, const char* who) {
    if (!sc) return fail(PTX_E_ARG, std::string(who) + ": null scene");
    hipError_t e = hipSetDevice(sc->device);
    if (e != hipSuccess) return hip_fail(e, who);
    return PTX_OK;
}

struct SegRange {  // a segmented queue as the traversal kernel sees it (see RenderState)
    const uint32_t* counts = nullptr;
    const uint32_t* plan = nullptr;  // window plan of the launch (k_plan)
    uint32_t n_seg = 1, cap = 0, scale = 1;
    uint32_t div = 1;  // segment s takes its count from counts[s / div]
    bool internal = false;
};

template <bool ANY>
int launch_traverse(ptx_scene* sc, const ptx_ray* d_rays, const SegRange& sr, uint32_t n, ptx_hit* d_hits, uint8_t* d_occ,
                    uint32_t* d_cursor, uint2* d_spill, unsigned long long* work, hipStream_t stream, int max_grid = 0) {
    const bool count = (sc->flags & PTX_FLAG_COUNT_WORK) != 0 && work;
    TravArgs a{};
    a.nodes = sc->dev_nodes.p;
    a.top_n = sc->top_n;
    a.root = sc->root_desc;
    a.tris = sc->leaf_tris.p;
    a.rays = d_rays;
    a.plan = sr.plan;
    a.n_seg = sr.n_seg;
    a.seg_cap = sr.cap;
    a.internal = sr.internal ? 1u : 0u;
    a.clip = sc->clip_box ? 1u : 0u;
    a.long_div = sc->long_div ? 1u : 0u;
    for (int k = 0; k < 3; ++k) a.box_lo[k] = sc->box_lo[k], a.box_hi[k] = sc->box_hi[k];
    a.fixed_count = n;
    a.hits = d_hits;
    a.occ = d_occ;
    a.cursor = d_cursor;
    a.spill = d_spill;
    // the spill area is sized for this many workgroups: never launch more
    int grid = max_grid > 0 && max_grid < sc->tgrid ? max_grid : sc->tgrid;
    a.spill_stride = (uint32_t)grid * TBLOCK;
    a.work = work;
    a.prof = sc->totals.p + (ANY ? 20 : 8);
    if (!sr.plan) {
        int need = (int)((n + 63) / 64 + (TBLOCK / 64) - 1) / (TBLOCK / 64);
        if (need < grid) grid = need > 0 ? need : 1;
    }
    if (count) k_traverse<ANY, true><<<grid, TBLOCK, 0, stream>>>(a);
    else k_traverse<ANY, false><<<grid, TBLOCK, 0, stream>>>(a);
    HIP_TRY(hipGetLastError());
    return PTX_OK;
}

void launch_check(ptx_scene* sc, const ptx_ray* rays, const SegRange& sr, const ptx_hit* hits, const uint8_t* occ,
                  hipStream_t stream) {
    CheckArgs c{};
    c.tris = sc->dev.tris;
    c.n_tris = sc->n_triangles;
    c.rays = rays;
    c.seg_counts = sr.counts;
    c.n_seg = sr.n_seg;
    c.seg_cap = sr.cap;
    c.scale = sr.scale;
    c.div = sr.div ? sr.div : 1u;
    c.internal = sr.internal ? 1u : 0u;
    c.hits = hits;
    c.occ = occ;
    c.fails = sc->fails.p;
    c.fail_cap = ptx_scene::FAIL_CAP;
    c.counters = sc->check_count.p;
    k_check<<<(int)sr.n_seg, BLOCK, 0, stream>>>(c);
}

// Bytes of render state (queues, per-path radiance, stacks) the scene holds right now.
uint64_t held_queue_bytes(const ptx_scene* sc) {
    uint64_t b = 0;
    for (const auto& up : sc->pipes) {
        if (!up) continue;
        const ptx_scene::Pipe& p = *up;
        b += (p.rays[0].n + p.rays[1].n + p.sh_rays.n) * sizeof(ptx_ray) + p.hits.n * sizeof(ptx_hit);
        b += p.sh_T.n * sizeof(float4) + (p.sh_contrib.n + p.transq[0].n + p.transq[1].n + p.radiance.n) * sizeof(Rad);
        b += p.perm.n * sizeof(uint32_t) + p.sh_occ.n + p.sh_bounce.n + (p.spill.n + p.spill2.n) * sizeof(uint2) + (p.counts.n + p.segs.n + p.plans.n) * sizeof(uint32_t);
    }
    return b;
}

// Default queue budget (no ptx_set_queue_slots): HALF of the device memory this scene could have at the time of
// the call - what is free now plus what the scene already holds - so that a second handle on the same GPU, or
// buffers a caller allocates later (RCCL, its own images), still find room; the rest of the budget logic
// (halving on an allocation failure) stays as the safety net.
uint64_t auto_queue_bytes(const ptx_scene* sc) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return sc->auto_bytes;
    return (uint64_t)((free_b + held_queue_bytes(sc)) * PTX_AUTO_MEMORY_FRACTION);
}

}  // namespace

extern "C" {

static int scene_create_impl(const ptx_scene_desc* d, uint32_t flags, ptx_scene** out);

int ptx_scene_create(const ptx_scene_desc* d, uint32_t flags, ptx_scene** out) {
    // (vectors, strings and maps below: std::bad_alloc must not cross the boundary)
    return guarded("ptx_scene_create", [&]() -> int { return scene_create_impl(d, flags, out); });
}

static int scene_create_impl(const ptx_scene_desc* d, uint32_t flags, ptx_scene** out) {
    if (!d || !out) return fail(PTX_E_ARG, "ptx_scene_create: null argument");
    if (!d->nodes || d->n_nodes == 0) return fail(PTX_E_ARG, "ptx_scene_create: tree has no nodes");
    if (d->n_triangles && !d->triangles) return fail(PTX_E_ARG, "ptx_scene_create: triangles missing");
    if (d->n_leaf_indices && !d->leaf_indices) return fail(PTX_E_ARG, "ptx_scene_create: leaf index list missing");
    if (d->n_lights > PTX_MAX_LIGHTS) return fail(PTX_E_LIMIT, "ptx_scene_create: more than PTX_MAX_LIGHTS lights");
    if (d->n_lights && !d->lights) return fail(PTX_E_ARG, "ptx_scene_create: lights missing");
    if (d->max_bounces > 255) return fail(PTX_E_LIMIT, "ptx_scene_create: max_bounces is a u8 in the reference (<= 255)");
    // the kernels index with what the tree says: prove every index in range before anything is launched
    std::string why = validate_flat_tree(d->nodes, d->n_nodes, d->leaf_indices, d->n_leaf_indices, d->n_triangles);
    if (!why.empty()) return fail(PTX_E_ARG, "ptx_scene_create: invalid flattened tree: " + why);
    const bool shading = d->normals && d->texcoords && d->material_ids && d->materials && d->n_materials;
    if (shading) {
        for (uint32_t i = 0; i < d->n_triangles; ++i)
            if (d->material_ids[i] >= d->n_materials) return fail(PTX_E_ARG, "ptx_scene_create: material id out of range");
        for (uint32_t i = 0; i < d->n_materials; ++i) {
            int32_t t = d->materials[i].texture;
            if (t >= (int32_t)d->n_textures) return fail(PTX_E_ARG, "ptx_scene_create: texture index out of range");
            if (t >= 0 && (!d->textures || !d->textures[t].rgb || !d->textures[t].width || !d->textures[t].height))
                return fail(PTX_E_ARG, "ptx_scene_create: empty texture");
        }
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return fail(PTX_E_HIP, "ptx_scene_create: no HIP device available");
    auto sc = std::make_unique<ptx_scene>();
    HIP_TRY(hipGetDevice(&sc->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, sc->device));
    sc->grid = prop.multiProcessorCount * 8;
    {
        // grid-stride kernels want exactly one resident round of workgroups: a ninth block per CU that only
        // starts when another has finished doubles the kernel's time
        int per_cu = 0;
        const uint32_t L = d->n_lights;
        hipError_t oe = L == 1   ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_shade<1>, BLOCK, 0)
                        : L == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_shade<2>, BLOCK, 0)
                        : L == 4 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_shade<4>, BLOCK, 0)
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_shade<0>, BLOCK, 0);
        if (oe != hipSuccess || per_cu < 1) per_cu = 4;
        // ... and as many as fit: eight since v30 (62 VGPRs without packed math; with the 69-71 VGPRs before, seven fitted and
        // six were faster).  6 / 7 / 8, three alternating repetitions (tools/r3_env_ab.sh, profiles/r03_loop_variants.txt 16):
        // cornell_teapot 1946 / 1953 / 1967 M samples/s, cornell_textured 2784 / 2823 / 2831, cubesplosion 8343 / 8471 / 8441,
        // sr2 12184 / 12320 / 12171.
        if (per_cu > 8) per_cu = 8;
        if (const char* ev = getenv("PTX_SHADE_PER_CU")) per_cu = atoi(ev) > 0 ? atoi(ev) : per_cu;
        if (getenv("PTX_VERBOSE")) fprintf(stderr, "[ptx] k_shade blocks per CU: %d\n", per_cu);
        // Queue segments = workgroups of k_start / k_shade / k_resolve.  One per resident workgroup slot leaves a third of
        // the slots idle on average (PMC, cornell_teapot: 4100 of 6144 waves resident - a workgroup that has finished its
        // segment has nothing else to do until the slowest one ends), so a large batch is cut into PTX_SEG_ROUNDS times
        // as many segments and the hardware dispatcher hands the later ones out as slots free up.  With the resolve folded into
        // k_shade (v26), 2 / 4 / 6 / 8 rounds: cornell_teapot 1864 / 1888 / 1910 / 1934 M samples/s, cornell_textured 2598 /
        // 2654 / 2747 / 2762, cubesplosion 8036 / 8143 / 8278 / 8224, sr2 11700 / 11530 / 11698 / 11321 (its segments are
        // mostly sky: smaller ones only add per-workgroup overhead) - tools/r3_envsweep.sh.  A small batch keeps one per slot.
        sc->seg_rounds = PTX_SEG_ROUNDS;
        if (const char* ev = getenv("PTX_SEG_ROUNDS")) sc->seg_rounds = atoi(ev) > 0 && atoi(ev) <= 16 ? atoi(ev) : sc->seg_rounds;
        sc->shade_slots = prop.multiProcessorCount * per_cu;
        sc->shade_grid = sc->shade_slots * sc->seg_rounds;
    }
    sc->tgrid = prop.multiProcessorCount * PTX_BLOCKS_PER_CU;  // sized so that the LDS of a CU is filled
    sc->flags = flags;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)64 << 30;
        sc->auto_bytes = (uint64_t)(free_b * PTX_AUTO_MEMORY_FRACTION);  // only used if hipMemGetInfo fails later
    }
    if (const char* e = getenv("PTX_OVERLAP")) sc->overlap_shadow = atoi(e) != 0;
    if (const char* e = getenv("PTX_SLOTS_MI")) {
        const long v = atol(e);
        if (v >= 1 && v <= 4000) sc->slot_budget = (uint64_t)v << 20;  // clamped by the index limit at render time
    }
    if (const char* e = getenv("PTX_PIPES")) {
        int k = atoi(e);
        if (k >= 1 && k <= 8) sc->n_pipes = (uint32_t)k;
    }
    sc->n_triangles = d->n_triangles;
    sc->n_nodes = d->n_nodes;
    sc->shading = shading;

    std::vector<TriRec> recs;
    pack_triangles(d->triangles, d->n_triangles, recs);
    if (!(flags & PTX_FLAG_NO_BOX_CLIP) && d->n_triangles) {  // bounding box of the triangles for k_traverse's range clip
        float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
        for (size_t v = 0; v < 3 * (size_t)d->n_triangles; ++v)
            for (int k = 0; k < 3; ++k) {
                const float c = d->triangles[3 * v + k];
                lo[k] = c < lo[k] ? c : lo[k], hi[k] = c > hi[k] ? c : hi[k];
            }
        float ext = 0;
        for (int k = 0; k < 3; ++k) ext = hi[k] - lo[k] > ext ? hi[k] - lo[k] : ext;
        for (int k = 0; k < 3; ++k) sc->box_lo[k] = lo[k] - 1e-3f * ext - 1e-6f, sc->box_hi[k] = hi[k] + 1e-3f * ext + 1e-6f;
        sc->clip_box = true;
    }
    HIP_TRY(sc->nodes.upload(d->nodes, d->n_nodes));
    for (uint32_t i = 0; i < d->n_nodes && !sc->long_div; ++i)
        if ((d->nodes[i].w1 & 3u) != 3u && !div_pos_ok(u2f(d->nodes[i].w0))) sc->long_div = true;
    if (const char* e = getenv("PTX_LONG_DIV")) sc->long_div = sc->long_div || atoi(e) != 0;  // A/B and tests
    {
        std::vector<DevNode> dn;
        std::vector<uint32_t> dl;
        const std::string big = build_device_tree(d->nodes, d->n_nodes, d->leaf_indices, (uint32_t)TOP_NODES, dn, dl, sc->root_desc, sc->top_n);
        if (!big.empty()) return fail(PTX_E_LIMIT, "ptx_scene_create: " + big);
        static_assert(sizeof(DevNode) == sizeof(uint4), "a device node is one dwordx4");
        HIP_TRY(sc->dev_nodes.upload(reinterpret_cast<const uint4*>(dn.data()), dn.size()));
        if (getenv("PTX_VERBOSE"))
            fprintf(stderr, "[ptx] device tree: %zu inner nodes, %zu leaf records for %u leaf references\n", dn.size(), dl.size(), d->n_leaf_indices);
        std::vector<TriRec> lt(dl.size());  // the records in leaf order, tagged with index and last flag
        for (size_t k = 0; k < dl.size(); ++k) {
            lt[k] = recs[dl[k] & DESC_PAYLOAD];
            lt[k].f[10] = u2f(dl[k]);
        }
        HIP_TRY(sc->leaf_tris.upload(lt.data(), lt.size()));
    }
    HIP_TRY(sc->leaf_idx.upload(d->leaf_indices, d->n_leaf_indices));
    HIP_TRY(sc->tris.upload(recs.data(), recs.size()));
    sc->dev.nodes = sc->nodes.p;
    sc->dev.leaf_idx = sc->leaf_idx.p;
    sc->dev.tris = sc->tris.p;
    if (shading) {
        std::vector<float4> props(4 * (size_t)d->n_triangles);
        for (uint32_t i = 0; i < d->n_triangles; ++i) {
            float f[16];
            memcpy(f, d->normals + 9 * (size_t)i, 9 * sizeof(float));
            memcpy(f + 9, d->texcoords + 6 * (size_t)i, 6 * sizeof(float));
            memcpy(f + 15, &d->material_ids[i], 4);
            memcpy(&props[4 * (size_t)i], f, sizeof f);
        }
        HIP_TRY(sc->props.upload(props.data(), props.size()));
        std::vector<DevMaterial> mats(d->n_materials);
        for (uint32_t t = 0; t < d->n_textures; ++t) {
            auto buf = std::make_unique<DevBuf<float>>();
            HIP_TRY(buf->upload(d->textures[t].rgb, (size_t)d->textures[t].width * d->textures[t].height * 3));
            sc->textures.push_back(std::move(buf));
        }
        for (uint32_t i = 0; i < d->n_materials; ++i) {
            const ptx_material& m = d->materials[i];
            DevMaterial& o = mats[i];
            memcpy(o.albedo, m.albedo, sizeof o.albedo);
            memcpy(o.f0, m.schlick_f0, sizeof o.f0);
            o.texture = m.texture;
            o.transmission = m.transmission;
            o.ior = m.ior;
            o.tex_w = o.tex_h = 0;
            o.tex_rgb = nullptr;
            if (m.texture >= 0) {
                o.tex_w = d->textures[m.texture].width;
                o.tex_h = d->textures[m.texture].height;
                o.tex_rgb = sc->textures[m.texture]->p;
            }
        }
        HIP_TRY(sc->materials.upload(mats.data(), mats.size()));
        sc->dev.props = sc->props.p;
        sc->dev.materials = sc->materials.p;
    }
    sc->dev.n_lights = d->n_lights;
    for (uint32_t l = 0; l < d->n_lights; ++l) sc->dev.lights[l] = d->lights[l];
    memcpy(sc->dev.env, d->environment, sizeof sc->dev.env);
    sc->dev.max_bounces = d->max_bounces;
    HIP_TRY(sc->totals.ensure(32));
    HIP_TRY(hipMemset(sc->totals.p, 0, 32 * sizeof(unsigned long long)));
    HIP_TRY(sc->spill.ensure((size_t)(PTX_MAX_DEPTH - LDS_DEPTH) * sc->tgrid * TBLOCK));
    HIP_TRY(sc->batch_cursor.ensure(CURSOR_WORDS));
    HIP_TRY(hipEventCreateWithFlags(&sc->ev_batch, hipEventDisableTiming));
    {
        // the scratch array of the allocation-free entry point: its address on THIS device (every GPU holds
        // its own copy), looked up once per device with that device current
        static std::mutex mu;
        static unsigned char* per_device[64] = {nullptr};
        std::lock_guard<std::mutex> lock(mu);
        unsigned char* ptr = sc->device < 64 ? per_device[sc->device] : nullptr;
        if (!ptr) {
            HIP_TRY(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
            if (sc->device < 64) per_device[sc->device] = ptr;
        }
        sc->noalloc_scratch = ptr;
    }
    if (flags & PTX_FLAG_CHECK_BRUTE) {
        HIP_TRY(sc->fails.ensure(ptx_scene::FAIL_CAP));
        HIP_TRY(sc->check_count.ensure(2));
        HIP_TRY(hipMemset(sc->check_count.p, 0, 2 * sizeof(unsigned long long)));
    }
    *out = sc.release();
    return PTX_OK;
}

int ptx_device_count(int* count) {
    if (!count) return fail(PTX_E_ARG, "ptx_device_count: null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return PTX_OK;
}

int ptx_scene_create_on(const ptx_scene_desc* d, uint32_t flags, int device, ptx_scene** out) {
    int n = 0, before = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return fail(PTX_E_HIP, "ptx_scene_create_on: no HIP device available");
    if (device < 0 || device >= n) return fail(PTX_E_ARG, "ptx_scene_create_on: no such device");
    HIP_TRY(hipGetDevice(&before));
    HIP_TRY(hipSetDevice(device));
    const int rc = ptx_scene_create(d, flags, out);
    (void)hipSetDevice(before);
    return rc;
}

int ptx_scene_device(const ptx_scene* scene, int* device) {
    if (!scene || !device) return fail(PTX_E_ARG, "ptx_scene_device: null argument");
    *device = scene->device;
    return PTX_OK;
}

void ptx_scene_destroy(ptx_scene* scene) {
    if (!scene) return;
    (void)hipSetDevice(scene->device);
    (void)hipDeviceSynchronize();
    delete scene;
}

int ptx_intersect_batch_device(ptx_scene* sc, const ptx_ray* d_rays, size_t n, ptx_hit* d_hits, void* stream) {
    return guarded("ptx_intersect_batch_device", [&]() -> int {
    int rc = check_scene(sc, "ptx_intersect_batch_device");
    if (rc) return rc;
    if (n == 0) return PTX_OK;
    if (!d_rays || !d_hits) return fail(PTX_E_ARG, "ptx_intersect_batch_device: null buffer");
    if (n > 0xFFFFFFFFull) return fail(PTX_E_LIMIT, "ptx_intersect_batch_device: more than 2^32-1 rays in one call");
    if (sc->batch_pending) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, sc->ev_batch, 0));  // see ptx_scene::ev_batch
    {
        Timed tm(sc, (hipStream_t)stream, KC_EXTEND);
        sc->batch_extend_rays += n;
        ++sc->extend_launches;
        HIP_TRY(hipMemsetAsync(sc->batch_cursor.p, 0, CURSOR_WORDS * sizeof(uint32_t), (hipStream_t)stream));
        rc = launch_traverse<false>(sc, d_rays, SegRange{}, (uint32_t)n, d_hits, nullptr, sc->batch_cursor.p, sc->spill.p,
                                    sc->totals.p + 2, (hipStream_t)stream);
    }
    HIP_TRY(hipEventRecord(sc->ev_batch, (hipStream_t)stream));
    sc->batch_pending = true;
    return rc;
    });
}

int ptx_intersect_batch_device_noalloc(ptx_scene* sc, const ptx_ray* d_rays, size_t n, ptx_hit* d_hits, void* stream) {
    int rc = check_scene(sc, "ptx_intersect_batch_device_noalloc");
    if (rc) return rc;
    if (n == 0) return PTX_OK;
    if (!d_rays || !d_hits) return fail(PTX_E_ARG, "ptx_intersect_batch_device_noalloc: null buffer");
    if (n > PTX_NOALLOC_MAX_RAYS)
        return fail(PTX_E_LIMIT, "ptx_intersect_batch_device_noalloc: more than PTX_NOALLOC_MAX_RAYS (2^24) rays in one call");
    unsigned char* const ptr = sc->noalloc_scratch;  // never the array's name: that is a host-side stand-in
    if (!ptr) return fail(PTX_E_HIP, "ptx_intersect_batch_device_noalloc: the scratch array has no address on this device");
    // No allocation, no event, no timing bracket: a memset node and one kernel launch on the caller's stream.
    sc->batch_extend_rays += n;
    ++sc->extend_launches;
    HIP_TRY(hipMemsetAsync(ptr, 0, CURSOR_WORDS * sizeof(uint32_t), (hipStream_t)stream));  // the cursors must start at zero
    return launch_traverse<false>(sc, d_rays, SegRange{}, (uint32_t)n, d_hits, nullptr, reinterpret_cast<uint32_t*>(ptr),
                                  reinterpret_cast<uint2*>(ptr + CURSOR_WORDS * sizeof(uint32_t)), sc->totals.p + 2, (hipStream_t)stream, NOALLOC_GRID);
}

int ptx_occluded_batch_device(ptx_scene* sc, const ptx_ray* d_rays, size_t n, uint8_t* d_out, void* stream) {
    return guarded("ptx_occluded_batch_device", [&]() -> int {
    int rc = check_scene(sc, "ptx_occluded_batch_device");
    if (rc) return rc;
    if (n == 0) return PTX_OK;
    if (!d_rays || !d_out) return fail(PTX_E_ARG, "ptx_occluded_batch_device: null buffer");
    if (n > 0xFFFFFFFFull) return fail(PTX_E_LIMIT, "ptx_occluded_batch_device: more than 2^32-1 rays in one call");
    if (sc->batch_pending) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, sc->ev_batch, 0));
    {
        Timed tm(sc, (hipStream_t)stream, KC_SHADOW);
        sc->batch_shadow_rays += n;
        ++sc->shadow_launches;
        HIP_TRY(hipMemsetAsync(sc->batch_cursor.p, 0, CURSOR_WORDS * sizeof(uint32_t), (hipStream_t)stream));
        rc = launch_traverse<true>(sc, d_rays, SegRange{}, (uint32_t)n, nullptr, d_out, sc->batch_cursor.p, sc->spill.p,
                                   sc->totals.p + 4, (hipStream_t)stream);
    }
    HIP_TRY(hipEventRecord(sc->ev_batch, (hipStream_t)stream));
    sc->batch_pending = true;
    return rc;
    });
}

int ptx_intersect_batch(ptx_scene* sc, const ptx_ray* rays, size_t n, ptx_hit* hits) {
    return guarded("ptx_intersect_batch", [&]() -> int {
    int rc = check_scene(sc, "ptx_intersect_batch");
    if (rc) return rc;
    if (n == 0) return PTX_OK;
    if (!rays || !hits) return fail(PTX_E_ARG, "ptx_intersect_batch: null buffer");
    const size_t chunk = (size_t)1 << 24;
    HIP_TRY(sc->stage_rays.ensure(n < chunk ? n : chunk));
    HIP_TRY(sc->stage_hits.ensure(n < chunk ? n : chunk));
    for (size_t at = 0; at < n; at += chunk) {
        size_t m = n - at < chunk ? n - at : chunk;
        HIP_TRY(hipMemcpy(sc->stage_rays.p, rays + at, m * sizeof(ptx_ray), hipMemcpyHostToDevice));
        rc = ptx_intersect_batch_device(sc, sc->stage_rays.p, m, sc->stage_hits.p, nullptr);
        if (rc) return rc;
        HIP_TRY(hipMemcpy(hits + at, sc->stage_hits.p, m * sizeof(ptx_hit), hipMemcpyDeviceToHost));
    }
    return PTX_OK;
    });
}

int ptx_set_batch_samples(ptx_scene* sc, uint32_t samples_per_batch) {
    if (!sc) return fail(PTX_E_ARG, "ptx_set_batch_samples: null scene");
    sc->batch_samples = samples_per_batch;
    return PTX_OK;
}

int ptx_set_queue_slots(ptx_scene* sc, uint64_t slots) {
    if (!sc) return fail(PTX_E_ARG, "ptx_set_queue_slots: null scene");
    if (slots >= (1ull << 32)) return fail(PTX_E_LIMIT, "ptx_set_queue_slots: at most 2^32 - 1 slots");
    sc->slot_budget = slots;
    // a smaller budget than what is held takes effect now: the queues are released and rebuilt by the next render
    if (slots && (uint64_t)sc->cap_slots * sc->pipes.size() > slots) return ptx_scene_trim(sc);
    return PTX_OK;
}

int ptx_scene_trim(ptx_scene* sc) {
    int rc = check_scene(sc, "ptx_scene_trim");
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    sc->pipes.clear();
    sc->cap_slots = sc->cap_paths = 0;
    sc->cap_rounds = 0;
    sc->stage_rays.release();
    sc->stage_hits.release();
    sc->stage_accum.release();
    sc->multi_accum.release();
    sc->multi_tmp.release();
    return PTX_OK;
}

int ptx_scene_memory(ptx_scene* sc, uint64_t* queue_bytes) {
    if (!sc || !queue_bytes) return fail(PTX_E_ARG, "ptx_scene_memory: null argument");
    *queue_bytes = held_queue_bytes(sc);
    return PTX_OK;
}

int ptx_set_pipelines(ptx_scene* sc, uint32_t n) {
    int rc = check_scene(sc, "ptx_set_pipelines");
    if (rc) return rc;
    if (n < 1 || n > 8) return fail(PTX_E_ARG, "ptx_set_pipelines: 1..8 pipelines");
    HIP_TRY(hipDeviceSynchronize());
    sc->n_pipes = n;  // the pipelines are (re)built by the next render call
    return PTX_OK;
}

struct Region {
    uint32_t x, y, w, h;
};

static int render_impl(ptx_scene* sc, const ptx_pinhole* cam, uint64_t seed, uint32_t first_sample,
                       uint32_t sample_count, float* d_accum, hipStream_t user_stream, ptx_progress_cb progress,
                       void* user, const Region* region = nullptr) {
    clear_stale_stamps(sc);
    if (!cam || !d_accum) return fail(PTX_E_ARG, "ptx_render: null argument");
    if (!sc->shading) return fail(PTX_E_ARG, "ptx_render: scene was created without shading data");
    if (cam->width == 0 || cam->height == 0) return fail(PTX_E_ARG, "ptx_render: empty image");
    if ((uint64_t)cam->width * cam->height > (1ull << 27)) return fail(PTX_E_LIMIT, "ptx_render: image larger than 2^27 pixels");
    const Region reg = region ? *region : Region{0u, 0u, cam->width, cam->height};
    if (reg.w == 0 || reg.h == 0 || (uint64_t)reg.x + reg.w > cam->width || (uint64_t)reg.y + reg.h > cam->height)
        return fail(PTX_E_ARG, "ptx_render_region: the region is empty or reaches outside the image");
    const uint64_t n_pixels64 = (uint64_t)reg.w * reg.h;
    if ((uint64_t)first_sample + sample_count > 0xFFFFFFFFull) return fail(PTX_E_LIMIT, "ptx_render: sample index overflow");
    const uint32_t n_pixels = (uint32_t)n_pixels64;
    // Batches.  The last rays of a launch finish ~0.3 ms after its cursor drains, and the late bounces of a
    // batch are small launches that cannot fill the chip (measured at 1080^2: 64 / 128 / 256 / 448 samples per
    // batch = 1.0 / 1.26 / 1.35 / 1.45 G samples/s), so a batch should be as wide as possible.  Two budgets
    // bound it: queue SLOTS (162 B each with one light; default: half of the memory the scene could have at the time of
    // the call, auto_queue_bytes(); slot x light indices are 32-bit)
    // and path ids (12 B of radiance each, < 2^31).  A batch with more paths than slots runs with path regeneration:
    // every queue segment starts the next paths of its range as earlier ones end (see RenderState), so all
    // launches but the last few stay slot-wide.
    const uint32_t n_pipes = sc->n_pipes;
    const uint32_t B = sc->dev.max_bounces, L = sc->dev.n_lights;
    // bytes per queue slot (see ensure_render_state) + the radiance of up to two paths per slot
    const uint64_t slot_bytes = 121 + (uint64_t)(L ? L : 1u) * (33 + (L > 1 ? 12 : 0)) + 24;
    uint64_t slot_budget = (sc->slot_budget ? sc->slot_budget : auto_queue_bytes(sc) / slot_bytes) / n_pipes;
    // The automatic budget follows the free memory, which moves a little from call to call; queues that already
    // cover three quarters of it are kept as they are rather than rebuilt 1 % wider (a rebuild stalls the call:
    // device synchronisation plus ~150 GB of hipFree / hipMalloc)
    if (!sc->slot_budget && sc->pipes.size() == n_pipes && sc->cap_slots >= slot_budget - slot_budget / 4) slot_budget = sc->cap_slots;
    const uint64_t max_batch_paths = (1ull << 31) - (1ull << 20);
    uint32_t S = 1, n_seg = 1, seg_range = 1, seg_cap = 64, rounds = 0;
    uint32_t want_S = sc->batch_samples;  // samples per batch, 0 = choose
    for (int attempt = 0;; ++attempt) {
        const uint64_t index_limit = ((1ull << 32) - (1ull << 22)) / (L ? L : 1u);  // shadow-ray indices = slot * L + l
        if (slot_budget > index_limit) slot_budget = index_limit;
        S = want_S;
        if (S == 0) {
            // up to 2 slot-loads of paths per batch, spread evenly over the pipelines
            const uint64_t by_rounds = 2 * slot_budget / n_pixels, by_ids = max_batch_paths / n_pixels;
            uint64_t most = by_rounds < by_ids ? by_rounds : by_ids;
            if (most < 1) most = 1;
            const uint32_t per_pipe = (sample_count + n_pipes - 1) / n_pipes;
            S = (uint32_t)(most < per_pipe ? most : (per_pipe ? per_pipe : 1));
        }
        while ((uint64_t)S * n_pixels > max_batch_paths && S > 1) S = (S + 1) / 2;
        if (S < 1) S = 1;
        if (sample_count > 0) {  // equal-sized batches, their number a multiple of the pipeline count when possible
            uint32_t n_batches = (sample_count + S - 1) / S;
            if (want_S == 0 && n_batches > 1) n_batches = (n_batches + n_pipes - 1) / n_pipes * n_pipes;
            if (n_batches > sample_count) n_batches = sample_count;
            S = (sample_count + n_batches - 1) / n_batches;
        }
        // queue segments (see RenderState): one per k_shade workgroup; seg_range paths each, seg_cap of them live
        const uint64_t batch_paths = (uint64_t)S * n_pixels;
        // (several rounds of segments only while a segment keeps PTX_SEG_MIN_PATHS paths; one per slot below that)
        n_seg = (uint32_t)sc->shade_slots;
        if (batch_paths / PTX_SEG_MIN_PATHS > n_seg) n_seg = (uint32_t)(batch_paths / PTX_SEG_MIN_PATHS < (uint64_t)sc->shade_grid ? batch_paths / PTX_SEG_MIN_PATHS : (uint64_t)sc->shade_grid);
        // (a small batch: segments of about a thousand paths, one per CU at least - cornell 256 x 256, 16 spp with 2 / 3 / 4 /
        // 5 / 6 / 8 segments per CU: 720 / 726 / 725 / 701 / 694 / 657 M samples/s)
        const uint64_t cus = (uint64_t)(sc->tgrid / PTX_BLOCKS_PER_CU);
        if (batch_paths / PTX_SEG_SMALL_PATHS < n_seg) n_seg = (uint32_t)(batch_paths / PTX_SEG_SMALL_PATHS > cus ? batch_paths / PTX_SEG_SMALL_PATHS : cus);
        if ((batch_paths + 255) / 256 < n_seg) n_seg = (uint32_t)((batch_paths + 255) / 256);
        if (n_seg < 1) n_seg = 1;
        seg_range = (uint32_t)(((batch_paths + n_seg - 1) / n_seg + 63) / 64 * 64);  // whole waves / pixel tiles per segment
        seg_cap = seg_range;
        if (slot_budget < (uint64_t)n_seg * seg_cap) {
            // the budget bounds the slots: fewer segments when it is under 64 slots per segment, then
            // narrower segments (path regeneration covers the rest of each segment's range)
            if (slot_budget / n_seg < 64) {
                n_seg = (uint32_t)(slot_budget / 64 > 0 ? slot_budget / 64 : 1);
                seg_range = (uint32_t)(((batch_paths + n_seg - 1) / n_seg + 63) / 64 * 64);
                seg_cap = seg_range;
            }
            const uint64_t by_budget = slot_budget / n_seg / 64 * 64;
            if (by_budget < seg_cap) seg_cap = (uint32_t)(by_budget >= 64 ? by_budget : 64);
        }
        if ((uint64_t)n_seg * seg_cap > index_limit)
            return fail(PTX_E_LIMIT, "ptx_render: queue slots x lights beyond the 32-bit index range (lower ptx_set_batch_samples or ptx_set_queue_slots)");
        // a slot turns over after at most B rounds: B x ceil(range / cap) rounds start and finish every path
        rounds = B * ((seg_range + seg_cap - 1) / seg_cap);
        const int rc = ensure_render_state(sc, (size_t)n_seg * seg_cap, (size_t)batch_paths, rounds);
        if (rc == PTX_OK) break;
        // out of device memory (somebody else holds it): everything was released; try again half as wide
        if (rc != PTX_E_OOM || attempt >= 6 || (uint64_t)n_seg * seg_cap <= (1u << 16)) return rc;
        (void)hipGetLastError();
        slot_budget = (uint64_t)n_seg * seg_cap / 2;
        if (want_S > 1) want_S = (S + 1) / 2;
    }
    sc->last_rounds = rounds;
    int rc = PTX_OK;

    const int grid = sc->grid;
    auto t0 = std::chrono::steady_clock::now();
    // fork: the pipelines start after everything already queued on the caller's stream
    HIP_TRY(hipEventRecord(sc->ev_fork, user_stream));
    for (auto& p : sc->pipes) HIP_TRY(hipStreamWaitEvent(p->stream, sc->ev_fork, 0));
    hipEvent_t prev_accum = nullptr;
    uint32_t batch_no = 0;
    // the batches run inside a lambda so that EVERY way out of them, errors included, passes the join below:
    // the caller's stream must never be left without its dependency on the forked pipelines
    auto run_batches = [&]() -> int {
    for (uint32_t done = 0; done < sample_count; done += S, ++batch_no) {
        ptx_scene::Pipe& p = *sc->pipes[batch_no % sc->pipes.size()];
        sc->last_pipe = (int)(batch_no % sc->pipes.size());
        hipStream_t stream = p.stream;
        const uint32_t ns = sample_count - done < S ? sample_count - done : S;
        RenderState st{};
        st.rays[0] = p.rays[0].p;
        st.rays[1] = p.rays[1].p;
        st.hits = p.hits.p;
        st.transq[0] = p.transq[0].p;
        st.transq[1] = p.transq[1].p;
        st.radiance = p.radiance.p;
        st.sh_rays = p.sh_rays.p;
        st.sh_T = p.sh_T.p;
        st.sh_contrib = p.sh_contrib.p;
        st.sh_occ = p.sh_occ.p;
        st.sh_bounce = sc->raylog_cap ? p.sh_bounce.p : nullptr;
        st.perm = (sc->flags & PTX_FLAG_MATERIAL_QUEUES) ? p.perm.p : nullptr;
        st.qcounts = p.counts.p;
        st.cursors = p.counts.p + (2 * rounds + 2);
        st.seg_ext[0] = p.segs.p;
        st.seg_ext[1] = p.segs.p + n_seg;
        st.seg_sh = p.segs.p + 2 * (size_t)n_seg;
        st.seg_next = p.segs.p + 3 * (size_t)n_seg;
        st.n_seg = n_seg;
        st.seg_cap = seg_cap;
        st.seg_range = seg_range;
        st.totals = sc->totals.p;
        // (the ray log and the brute-force checker want every ray in the queue)
        st.cull = sc->clip_box && !sc->raylog_cap && !(sc->flags & PTX_FLAG_CHECK_BRUTE) ? 1u : 0u;
        for (int k = 0; k < 3; ++k) st.box_lo[k] = sc->box_lo[k], st.box_hi[k] = sc->box_hi[k], st.env[k] = sc->dev.env[k];
        st.cam = *cam;
        st.seed = seed;
        st.n_pixels = n_pixels;
        st.reg_x = reg.x, st.reg_y = reg.y, st.reg_w = reg.w;
        st.tiles_per_row = (PTX_PIXEL_TILES && reg.w % 8u == 0 && reg.h % 8u == 0) ? reg.w / 8u : 0u;
        st.n_samples = ns;
        st.n_paths = ns * n_pixels;
        st.first_sample = first_sample + done;
        HIP_TRY(hipMemsetAsync(p.counts.p, 0, ((2 + 2 * CURSOR_WORDS) * (size_t)rounds + 2) * sizeof(uint32_t), stream));
        // window plans (k_plan): [0] of the next extend launch, [1] of the next shadow launch
        uint32_t* const plan_ext = p.plans.p;
        uint32_t* const plan_sh = p.plans.p + plan_words(sc);
        const uint32_t n_waves = (uint32_t)sc->tgrid * (TBLOCK / 64);
        if (rounds == 0) {  // max_bounces == 0: nothing is traced (the reference's loop body never runs)
            HIP_TRY(hipMemsetAsync(p.radiance.p, 0, (size_t)st.n_paths * sizeof(Rad), stream));
        } else {
            Timed tm(sc, stream, KC_GENERATE);
            k_start<<<(int)n_seg, BLOCK, 0, stream>>>(st);
            k_plan<<<1, 1024, 0, stream>>>(PlanArgs{st.seg_ext[0], n_seg, 1u, 1u, n_waves, 0u, plan_ext}, PlanArgs{});
        }
        const bool overlap = sc->overlap_shadow && L > 0;
        hipStream_t sstream = overlap ? p.stream2 : stream;  // stream of the shadow launches
        uint2* sspill = overlap ? p.spill2.p : p.spill.p;
        for (uint32_t b = 1; b <= rounds; ++b) {
            {
                Timed tm(sc, stream, KC_EXTEND, b);
                ++sc->extend_launches;
                SegRange sr;
                sr.counts = st.seg_ext[(b - 1) & 1];
                sr.plan = plan_ext;
                sr.n_seg = n_seg;
                sr.cap = seg_cap;
                sr.internal = true;
                rc = launch_traverse<false>(sc, st.rays[(b - 1) & 1], sr, 0, st.hits, nullptr, st.cursors + 2 * CURSOR_WORDS * (b - 1),
                                            p.spill.p, sc->totals.p + 2, stream);
                if (rc) return rc;
                if (sc->flags & PTX_FLAG_CHECK_BRUTE) launch_check(sc, st.rays[(b - 1) & 1], sr, st.hits, nullptr, stream);
                if (sc->raylog_cap) k_log<false><<<(int)n_seg, BLOCK, 0, stream>>>(st, LogArgs{sc->raylog.p, sc->raylog_count.p, sc->raylog_cap}, (b - 1) & 1, 1u);
            }
            // k_shade(b) reuses the shadow queue and adds to the radiance: after resolve(b-1)
            if (overlap && b > 1) HIP_TRY(hipStreamWaitEvent(stream, p.ev_resolve, 0));
            {
                Timed tm(sc, stream, KC_SHADE);
                const int sg = (int)n_seg;
                if (st.perm) k_bucket<<<sg, BLOCK, 0, stream>>>(sc->dev, st, (b - 1) & 1);  // (its time counts as shading)
                if (L == 1) k_shade<1><<<sg, BLOCK, 0, stream>>>(sc->dev, st, b);
                else if (L == 2) k_shade<2><<<sg, BLOCK, 0, stream>>>(sc->dev, st, b);
                else if (L == 4) k_shade<4><<<sg, BLOCK, 0, stream>>>(sc->dev, st, b);
                else k_shade<0><<<sg, BLOCK, 0, stream>>>(sc->dev, st, b);
                if (seg_range > seg_cap && b < rounds) k_regenerate<<<sg, BLOCK, 0, stream>>>(st, b);
                // both queues k_shade has just written: the shadow launch of this round, the extend launch of the next
                k_plan<<<2, 1024, 0, stream>>>(PlanArgs{st.seg_ext[b & 1], n_seg, 1u, 1u, n_waves, 0u, b < rounds ? plan_ext : nullptr},
                                               PlanArgs{st.seg_sh, n_seg * L, L ? L : 1u, 1u, n_waves, 1u, L ? plan_sh : nullptr});
            }
            if (L > 0) {
                if (overlap) {
                    HIP_TRY(hipEventRecord(p.ev_shade, stream));
                    HIP_TRY(hipStreamWaitEvent(sstream, p.ev_shade, 0));
                }
                {
                    Timed tm(sc, sstream, KC_SHADOW, b);
                    ++sc->shadow_launches;
                    SegRange sr;
                    sr.counts = st.seg_sh;
                    sr.plan = plan_sh;
                    sr.n_seg = n_seg * L;  // light-major: every (segment, light) pair is a segment of the launch
                    sr.cap = seg_cap;
                    sr.div = L;
                    rc = launch_traverse<true>(sc, st.sh_rays, sr, 0, nullptr, st.sh_occ, st.cursors + 2 * CURSOR_WORDS * (b - 1) + CURSOR_WORDS, sspill,
                                               sc->totals.p + 4, sstream);
                    if (rc) return rc;
                    static const bool debug_plan = getenv("PTX_DEBUG_PLAN") != nullptr;  // prints the window plans of every round (stalls the stream)
                    if (debug_plan) {
                        (void)hipStreamSynchronize(sstream);
                        uint32_t h[2][8];
                        (void)hipMemcpy(h[0], plan_ext, 32, hipMemcpyDeviceToHost);
                        (void)hipMemcpy(h[1], plan_sh, 32, hipMemcpyDeviceToHost);
                        uint32_t cur[2];  // (the first counter of each cursor)
                        (void)hipMemcpy(&cur[0], st.cursors + 2 * CURSOR_WORDS * (b - 1), 4, hipMemcpyDeviceToHost);
                        (void)hipMemcpy(&cur[1], st.cursors + 2 * CURSOR_WORDS * (b - 1) + CURSOR_WORDS, 4, hipMemcpyDeviceToHost);
                        fprintf(stderr, "[plan] round %u: next-extend chunk %u large %u small %u tail_num %u | shadow chunk %u large %u small %u tail_num %u | cursors ext %u sh %u (n_waves %u)\n",
                                b, h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3], cur[0], cur[1], n_waves);
                    }
                    if (sc->flags & PTX_FLAG_CHECK_BRUTE) launch_check(sc, st.sh_rays, sr, nullptr, st.sh_occ, sstream);
                    if (sc->raylog_cap) k_log<true><<<(int)n_seg, BLOCK, 0, sstream>>>(st, LogArgs{sc->raylog.p, sc->raylog_count.p, sc->raylog_cap}, 0u, L);
                }
                if (!PTX_RESOLVE_IN_SHADE || b == rounds) {
                    Timed tm(sc, sstream, KC_OTHER);
                    k_resolve<<<(int)n_seg, BLOCK, 0, sstream>>>(st, b, L);
                }
                if (overlap) HIP_TRY(hipEventRecord(p.ev_resolve, sstream));
            }
        }
        if (overlap) HIP_TRY(hipStreamWaitEvent(stream, p.ev_resolve, 0));
        // samples reach a pixel in index order: this batch's accumulate waits for the previous batch's
        if (prev_accum) HIP_TRY(hipStreamWaitEvent(stream, prev_accum, 0));
        {
            Timed tm(sc, stream, KC_OTHER);
            k_accumulate<<<grid, BLOCK, 0, stream>>>(st, d_accum);
            k_tally<<<1, 64, 0, stream>>>(st, rounds, L);
        }
        HIP_TRY(hipEventRecord(p.ev_accum, stream));
        prev_accum = p.ev_accum;
        HIP_TRY(hipGetLastError());
        sc->samples_done += (uint64_t)ns * n_pixels;
        if (progress) {
            HIP_TRY(hipStreamSynchronize(stream));
            double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            progress(done + ns, secs, user);
        }
    }
    return PTX_OK;
    };
    rc = run_batches();
    // join: the caller's stream continues after every pipeline has drained (after an error too: whatever was
    // launched before it still runs, and the second stream of a pipeline may not have been joined to the first)
    for (auto& p : sc->pipes) {
        if (rc != PTX_OK && sc->overlap_shadow) {
            (void)hipEventRecord(p->ev_resolve, p->stream2);
            (void)hipStreamWaitEvent(p->stream, p->ev_resolve, 0);
        }
        const hipError_t e1 = hipEventRecord(p->ev_join, p->stream);
        const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(user_stream, p->ev_join, 0) : e1;
        if (e2 != hipSuccess && rc == PTX_OK) rc = hip_fail(e2, "ptx_render: joining the pipelines");
    }
    return rc;
}

}  // extern "C"

namespace ptx {
int internal_render(ptx_scene* sc, const ptx_pinhole* cam, uint64_t seed, uint32_t first_sample, uint32_t sample_count,
                    float* d_accum, void* stream, ptx_progress_cb progress, void* user) {
    int rc = check_scene(sc, "ptx_render_multi");
    if (rc) return rc;
    return guarded("ptx_render_multi", [&]() -> int { return render_impl(sc, cam, seed, first_sample, sample_count, d_accum, (hipStream_t)stream, progress, user); });
}
int internal_device(const ptx_scene* sc) { return sc->device; }
int internal_multi_slot(ptx_scene* sc, size_t n_floats, float** accum, float** tmp, void** stream) {
    int rc = check_scene(sc, "ptx_render_multi");
    if (rc) return rc;
    if (!sc->multi_stream) HIP_TRY(hipStreamCreateWithFlags(&sc->multi_stream, hipStreamNonBlocking));
    HIP_TRY(sc->multi_accum.ensure(n_floats));
    if (tmp) HIP_TRY(sc->multi_tmp.ensure(n_floats));
    *accum = sc->multi_accum.p;
    if (tmp) *tmp = sc->multi_tmp.p;
    *stream = sc->multi_stream;
    return PTX_OK;
}
}  // namespace ptx

extern "C" {

int ptx_render_device(ptx_scene* sc, const ptx_pinhole* cam, uint64_t seed, uint32_t first_sample,
                      uint32_t sample_count, float* d_accum, void* stream) {
    int rc = check_scene(sc, "ptx_render_device");
    if (rc) return rc;
    return guarded("ptx_render_device", [&]() -> int { return render_impl(sc, cam, seed, first_sample, sample_count, d_accum, (hipStream_t)stream, nullptr, nullptr); });
}

int ptx_render_region_device(ptx_scene* sc, const ptx_pinhole* cam, uint32_t x, uint32_t y, uint32_t width, uint32_t height,
                             uint64_t seed, uint32_t first_sample, uint32_t sample_count, float* d_accum, void* stream) {
    int rc = check_scene(sc, "ptx_render_region_device");
    if (rc) return rc;
    const Region reg{x, y, width, height};
    return guarded("ptx_render_region_device", [&]() -> int { return render_impl(sc, cam, seed, first_sample, sample_count, d_accum, (hipStream_t)stream, nullptr, nullptr, &reg); });
}

int ptx_render(ptx_scene* sc, const ptx_pinhole* cam, uint64_t seed, uint32_t first_sample, uint32_t sample_count,
               float* accum_host, ptx_progress_cb progress, void* user) {
    int rc = check_scene(sc, "ptx_render");
    if (rc) return rc;
    if (!cam || !accum_host) return fail(PTX_E_ARG, "ptx_render: null argument");
    const size_t n = (size_t)cam->width * cam->height * 3;
    if (n == 0) return fail(PTX_E_ARG, "ptx_render: empty image");
    HIP_TRY(sc->stage_accum.ensure(n));
    HIP_TRY(hipMemcpy(sc->stage_accum.p, accum_host, n * sizeof(float), hipMemcpyHostToDevice));
    auto t0 = std::chrono::steady_clock::now();
    rc = guarded("ptx_render", [&]() -> int { return render_impl(sc, cam, seed, first_sample, sample_count, sc->stage_accum.p, nullptr, progress, user); });
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    sc->total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    HIP_TRY(hipMemcpy(accum_host, sc->stage_accum.p, n * sizeof(float), hipMemcpyDeviceToHost));
    return PTX_OK;
}

int ptx_tonemap_rgb8_device(const float* d_accum, uint32_t width, uint32_t height, uint32_t iterations, uint8_t* d_out,
                            void* stream) {
    if (!d_accum || !d_out) return fail(PTX_E_ARG, "ptx_tonemap_rgb8_device: null argument");
    const size_t n = (size_t)width * height * 3;
    if (n == 0) return PTX_OK;
    const uint16_t iters = (uint16_t)iterations;  // `total_iterations as u16` (worker.rs:113)
    const float inv = 1.0f / (float)iters;
    const size_t groups = (n + 3) / 4;
    const int grid = (int)(groups / 256 + 1 < 4096 ? groups / 256 + 1 : 4096);
    k_tonemap<<<grid, 256, 0, (hipStream_t)stream>>>(d_accum, n, inv, d_out);
    HIP_TRY(hipGetLastError());
    return PTX_OK;
}

int ptx_last_batch_counts(ptx_scene* sc, uint32_t* ext, uint32_t* sh, uint32_t n) {
    return guarded("ptx_last_batch_counts", [&]() -> int {
    int rc = check_scene(sc, "ptx_last_batch_counts");
    if (rc) return rc;
    if (!ext || !sh) return fail(PTX_E_ARG, "ptx_last_batch_counts: null argument");
    const uint32_t B = sc->last_rounds;
    if (sc->pipes.empty() || B == 0) return fail(PTX_E_ARG, "ptx_last_batch_counts: nothing rendered yet");
    HIP_TRY(hipDeviceSynchronize());
    std::vector<uint32_t> host(2 * (size_t)B + 4);
    HIP_TRY(hipMemcpy(host.data(), sc->pipes[sc->last_pipe]->counts.p, host.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint32_t b = 0; b < n; ++b) {
        ext[b] = b < B ? host[2 * b] : 0;
        sh[b] = b < B ? host[2 * b + 3] : 0;
    }
    return PTX_OK;
    });
}

int ptx_stats(ptx_scene* sc, ptx_counters* out) {
    return guarded("ptx_stats", [&]() -> int {
    int rc = check_scene(sc, "ptx_stats");
    if (rc) return rc;
    if (!out) return fail(PTX_E_ARG, "ptx_stats: null argument");
    HIP_TRY(hipDeviceSynchronize());
    resolve_stamps(sc);
    unsigned long long t[8] = {0};
    HIP_TRY(hipMemcpy(t, sc->totals.p, sizeof t, hipMemcpyDeviceToHost));
    memset(out, 0, sizeof *out);
    out->extend_rays = t[0] + sc->batch_extend_rays;
    out->shadow_rays = t[1] + sc->batch_shadow_rays;
    out->extend_inner_nodes = t[2];
    out->extend_tri_tests = t[3];
    out->shadow_inner_nodes = t[4];
    out->shadow_tri_tests = t[5];
    out->culled_camera_rays = t[6];
    out->samples = sc->samples_done;
    out->extend_launches = sc->extend_launches;
    out->shadow_launches = sc->shadow_launches;
    out->generate_ms = sc->class_ms[KC_GENERATE];
    out->other_ms = sc->class_ms[KC_OTHER];
    out->extend_ms = sc->class_ms[KC_EXTEND];
    out->shade_ms = sc->class_ms[KC_SHADE];
    out->shadow_ms = sc->class_ms[KC_SHADOW];
    out->total_ms = sc->total_ms;
    return PTX_OK;
    });
}

int ptx_round_times(ptx_scene* sc, double* extend_ms, double* shadow_ms, uint32_t n) {
    return guarded("ptx_round_times", [&]() -> int {
    int rc = check_scene(sc, "ptx_round_times");
    if (rc) return rc;
    if (!extend_ms || !shadow_ms) return fail(PTX_E_ARG, "ptx_round_times: null argument");
    HIP_TRY(hipDeviceSynchronize());
    resolve_stamps(sc);
    for (uint32_t r = 0; r < n; ++r) {
        extend_ms[r] = r < sc->round_ms[0].size() ? sc->round_ms[0][r] : 0.0;
        shadow_ms[r] = r < sc->round_ms[1].size() ? sc->round_ms[1][r] : 0.0;
    }
    return PTX_OK;
    });
}

int ptx_debug_loop_profile(ptx_scene* sc, uint64_t* out24) {
    int rc = check_scene(sc, "ptx_debug_loop_profile");
    if (rc) return rc;
    if (!out24) return fail(PTX_E_ARG, "ptx_debug_loop_profile: null argument");
    HIP_TRY(hipDeviceSynchronize());
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "");
    HIP_TRY(hipMemcpy(out24, sc->totals.p + 8, 24 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return PTX_OK;
}

int ptx_check_report(ptx_scene* sc, uint64_t* n_checked, uint64_t* n_fails) {
    int rc = check_scene(sc, "ptx_check_report");
    if (rc) return rc;
    if (!(sc->flags & PTX_FLAG_CHECK_BRUTE)) return fail(PTX_E_ARG, "ptx_check_report: scene was created without PTX_FLAG_CHECK_BRUTE");
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long c[2] = {0, 0};
    HIP_TRY(hipMemcpy(c, sc->check_count.p, sizeof c, hipMemcpyDeviceToHost));
    if (n_checked) *n_checked = c[0];
    if (n_fails) *n_fails = c[1];
    return PTX_OK;
}

int ptx_check_fail_log(ptx_scene* sc, uint8_t* out, uint32_t cap_records, uint32_t* n_written) {
    return guarded("ptx_check_fail_log", [&]() -> int {
    uint64_t n_fails = 0;
    int rc = ptx_check_report(sc, nullptr, &n_fails);
    if (rc) return rc;
    if ((!out && cap_records) || !n_written) return fail(PTX_E_ARG, "ptx_check_fail_log: null argument");
    uint32_t n = (uint32_t)(n_fails < ptx_scene::FAIL_CAP ? n_fails : ptx_scene::FAIL_CAP);
    if (n > cap_records) n = cap_records;
    std::vector<FailRec> recs(n);
    if (n) HIP_TRY(hipMemcpy(recs.data(), sc->fails.p, (size_t)n * sizeof(FailRec), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; ++i) {  // CheckedIntersection::as_bytes, checked_intersection.rs:27-56
        const FailRec& r = recs[i];
        const bool ref_some = (r.flags & 1u) != 0, kd_some = (r.flags & 2u) != 0 && !(r.flags & 4u);
        const float t_ext = kd_some ? r.t_kd : (ref_some ? r.t_ref : 1.0f);  // Ray::extended(t)
        const bool extend = kd_some || ref_some;
        float f[12];
        for (int k = 0; k < 3; ++k) {
            f[k] = r.o[k];
            f[3 + k] = extend ? t_ext * r.d[k] : r.d[k];
            f[6 + k] = ref_some ? r.o[k] + r.t_ref * r.d[k] : 0.0f;  // (the reference unwraps here)
            f[9 + k] = kd_some ? r.o[k] + r.t_kd * r.d[k] : 0.0f;
        }
        uint8_t* b = out + 50 * (size_t)i;
        const uint16_t it = (uint16_t)i;
        memcpy(b, &it, 2);
        memcpy(b + 2, f, 48);
    }
    *n_written = n;
    return PTX_OK;
    });
}

int ptx_raylog_enable(ptx_scene* sc, uint64_t max_records) {
    int rc = check_scene(sc, "ptx_raylog_enable");
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    if (max_records) {
        HIP_TRY(sc->raylog.ensure((size_t)max_records));
        HIP_TRY(sc->raylog_count.ensure(1));
        HIP_TRY(hipMemset(sc->raylog_count.p, 0, sizeof(unsigned long long)));
    }
    sc->raylog_cap = max_records;
    return PTX_OK;
}

int ptx_raylog_read(ptx_scene* sc, uint8_t* out, uint64_t cap_records, uint64_t* n_written, uint64_t* n_dropped) {
    return guarded("ptx_raylog_read", [&]() -> int {
    int rc = check_scene(sc, "ptx_raylog_read");
    if (rc) return rc;
    if (!sc->raylog_cap) return fail(PTX_E_ARG, "ptx_raylog_read: the ray log is off (ptx_raylog_enable)");
    if ((!out && cap_records) || !n_written) return fail(PTX_E_ARG, "ptx_raylog_read: null argument");
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long produced = 0;
    HIP_TRY(hipMemcpy(&produced, sc->raylog_count.p, sizeof produced, hipMemcpyDeviceToHost));
    const uint64_t kept = produced < sc->raylog_cap ? produced : sc->raylog_cap;
    std::vector<LogRec> recs((size_t)kept);
    if (kept) HIP_TRY(hipMemcpy(recs.data(), sc->raylog.p, (size_t)kept * sizeof(LogRec), hipMemcpyDeviceToHost));
    // iteration, pixel row, pixel column, bounce, extend before shadow, light: the order the reference writes in
    std::sort(recs.begin(), recs.end(), [](const LogRec& a, const LogRec& b) { return a.key < b.key; });
    const uint64_t n = kept < cap_records ? kept : cap_records;
    for (uint64_t i = 0; i < n; ++i) {  // RayLoggerWriter::write, raylogger.rs:54-66
        const LogRec& r = recs[(size_t)i];
        uint8_t* b = out + 33 * (size_t)i;
        const uint16_t it = (uint16_t)(r.key >> 46), py = (uint16_t)(r.key >> 30), px = (uint16_t)(r.key >> 14);
        memcpy(b, &it, 2);
        memcpy(b + 2, &px, 2);
        memcpy(b + 4, &py, 2);
        b[6] = (uint8_t)(r.key >> 6);
        b[7] = (uint8_t)((r.key >> 5) & 1u);
        b[8] = (uint8_t)(r.key & 1u);
        memcpy(b + 9, r.o, 12);
        memcpy(b + 21, r.d, 12);
    }
    *n_written = n;
    if (n_dropped) *n_dropped = produced - n;
    HIP_TRY(hipMemset(sc->raylog_count.p, 0, sizeof(unsigned long long)));  // the log starts over
    return PTX_OK;
    });
}

int ptx_stats_reset(ptx_scene* sc) {
    return guarded("ptx_stats_reset", [&]() -> int {
    int rc = check_scene(sc, "ptx_stats_reset");
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    resolve_stamps(sc);
    HIP_TRY(hipMemset(sc->totals.p, 0, 32 * sizeof(unsigned long long)));
    for (double& v : sc->class_ms) v = 0;
    sc->round_ms[0].clear();
    sc->round_ms[1].clear();
    sc->total_ms = 0;
    sc->samples_done = 0;
    sc->batch_extend_rays = sc->batch_shadow_rays = 0;
    if (sc->check_count.p) HIP_TRY(hipMemset(sc->check_count.p, 0, 2 * sizeof(unsigned long long)));
    sc->extend_launches = sc->shadow_launches = 0;
    return PTX_OK;
    });
}

}  // extern "C"
