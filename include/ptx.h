/* ptx.h — C ABI of libptx_mi355x.so: the MI355X (gfx950) implementation of daoo/pathtracer's render hot path.
 *
 * The reference is a Rust workspace without any FFI; its seams are ordinary Rust APIs.
 * Every entry point below names the reference interface it replaces (file:line under the reference
 * root).  A Rust `-sys` crate would bind exactly these symbols (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - every call returns int: 0 = ok, <0 = error class (PTX_E_*); ptx_last_error() gives the text
 *    (thread-local).  No exception crosses the boundary.
 *  - handles are opaque; the library copies what it needs and never keeps a caller pointer after return.
 *  - one handle = one GPU (the current HIP device at create time, or the one named to
 *    ptx_scene_create_on).  Calls on one handle are not re-entrant; different handles are independent.
 *    Multi-GPU inside one process = one handle per device and ptx_render_multi (sample ranges sharded over the
 *    handles, per-GPU buffers summed with RCCL); across processes = one handle per process, ptx_shard_samples
 *    + ptx_render_device + the caller's own RCCL reduce (bench.py, pathtracer_amd/parallel.py).
 *  - "host" pointers are ordinary memory; "device" pointers are HIP device memory on the handle's GPU.
 *  - there is no CPU fallback: without a usable HIP device every device call fails with PTX_E_HIP.
 */
#ifndef PTX_H
#define PTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTX_OK 0
#define PTX_E_ARG (-1)   /* bad argument / malformed scene description */
#define PTX_E_HIP (-2)   /* HIP runtime error or no device */
#define PTX_E_OOM (-3)   /* host or device allocation failed */
#define PTX_E_IO (-4)    /* file could not be read / parsed / written */
#define PTX_E_LIMIT (-5) /* size beyond a documented maximum */

#define PTX_MISS 0xFFFFFFFFu
#define PTX_MAX_LIGHTS 8
#define PTX_MAX_DEPTH 25 /* kdtree/src/lib.rs:16 MAX_DEPTH */

/* Ray + parametric range: `Ray{origin,direction}` (geometry/src/ray.rs:4-8) and the
 * `RangeInclusive<f32>` argument of `KdNode::intersect` (kdtree/src/lib.rs:73-78).  32 bytes. */
typedef struct ptx_ray {
    float ox, oy, oz, tmin;
    float dx, dy, dz, tmax;
} ptx_ray;

/* `Option<IndexedIntersection<TriangleIntersection>>` (geometry/src/geometry.rs:31-35,
 * geometry/src/triangle.rs:108-113); index == PTX_MISS encodes None (then t=u=v=0).  16 bytes. */
typedef struct ptx_hit {
    uint32_t index;
    float t, u, v;
} ptx_hit;

/* Flattened `KdNode` (kdtree/src/lib.rs:18-26), preorder, left child at i+1.  8 bytes.
 *   inner: w0 = bits of plane.distance (f32), w1 = (index of right child << 2) | axis (0=X,1=Y,2=Z)
 *   leaf : w0 = offset into the leaf index list,  w1 = (number of indices << 2) | 3            */
typedef struct ptx_kdnode {
    uint32_t w0, w1;
} ptx_kdnode;

/* `SahCost` (kdtree/src/sah.rs:52-56; defaults :122-130 = 1.0 / 1.5 / 0.8). */
typedef struct ptx_sah_cost {
    float traverse_cost, intersect_cost, empty_factor;
} ptx_sah_cost;

/* `Material` (tracing/src/material.rs:66-72): albedo colour or texture, Schlick f0, transmission, ior. */
typedef struct ptx_material {
    float albedo[3];
    int32_t texture; /* index into textures, or -1 for AlbedoSource::Color */
    float schlick_f0[3];
    float transmission;
    float ior;
} ptx_material;

/* `AlbedoSource::Texture(Rgb32FImage)` (tracing/src/material/albedo.rs:5-8): linear-light f32 RGB,
 * row 0 = top row of the image file. */
typedef struct ptx_texture {
    uint32_t width, height;
    const float* rgb; /* width*height*3 */
} ptx_texture;

/* `SphericalLight` (tracing/src/light.rs:26-30) — what every MTL light becomes (light.rs:79-89). */
typedef struct ptx_light {
    float center[3];
    float intensity[3]; /* color * lightintensity */
    float radius;
} ptx_light;

/* `Pinhole` (tracing/src/camera.rs:61-68): position, plane, dx, dy and the image size. */
typedef struct ptx_pinhole {
    float position[3], plane[3], dx[3], dy[3];
    uint32_t width, height;
} ptx_pinhole;

/* Everything `Pathtracer<TriangleCollection>` owns (tracing/src/pathtracer.rs:14-19,
 * tracing/src/collections.rs:32-37).  Arrays are caller-owned host memory, copied by ptx_scene_create.
 * normals/texcoords/material_ids/materials may be NULL for an intersect-only scene. */
typedef struct ptx_scene_desc {
    const float* triangles; /* n_triangles * 9: v0 v1 v2 (AnyTriangle::from is applied inside) */
    uint32_t n_triangles;
    const ptx_kdnode* nodes;
    uint32_t n_nodes;
    const uint32_t* leaf_indices;
    uint32_t n_leaf_indices;
    const float* normals;         /* n_triangles * 9: n0 n1 n2 (tracing/src/properties.rs:10-14) */
    const float* texcoords;       /* n_triangles * 6: uv0 uv1 uv2 */
    const uint32_t* material_ids; /* n_triangles */
    const ptx_material* materials;
    uint32_t n_materials;
    const ptx_texture* textures;
    uint32_t n_textures;
    const ptx_light* lights;
    uint32_t n_lights; /* <= PTX_MAX_LIGHTS */
    float environment[3]; /* pathtracer-cli/src/main.rs:155 uses 0.8,0.8,0.8 */
    uint32_t max_bounces; /* `Pathtracer::max_bounces` (u8 in the reference) */
} ptx_scene_desc;

/* Counters accumulated since ptx_stats_reset.  Node / triangle counts are only collected when the
 * scene was created with PTX_FLAG_COUNT_WORK (they cost atomics). */
typedef struct ptx_counters {
    uint64_t extend_rays, shadow_rays;
    uint64_t extend_inner_nodes, extend_tri_tests;
    uint64_t shadow_inner_nodes, shadow_tri_tests;
    uint64_t samples; /* pixel samples completed */
    uint64_t extend_launches, shadow_launches; /* kernel launches of each traversal class */
    double extend_ms, shadow_ms, shade_ms, generate_ms, total_ms; /* only with PTX_FLAG_TIME_KERNELS */
    double other_ms; /* resolve + accumulate kernels */
    uint64_t culled_camera_rays; /* of extend_rays: camera rays that miss the bounding box of the triangles and were finished
                                    where they were generated (radiance = environment) instead of being queued and traversed */
} ptx_counters;

typedef struct ptx_scene ptx_scene; /* opaque: device-resident scene (one GPU) */

#define PTX_FLAG_COUNT_WORK 1u   /* count node visits / triangle tests in the traversal kernels */
#define PTX_FLAG_TIME_KERNELS 2u /* bracket each kernel class with HIP events inside ptx_render */
#define PTX_FLAG_CHECK_BRUTE 4u  /* re-intersect every ray of ptx_render against all triangles (see ptx_check_report) */
#define PTX_FLAG_MATERIAL_QUEUES 8u /* shade the entries of every queue segment grouped by the material they hit (SURVEY 8f-1;
                                       same images bit for bit; measured slower than queue order, off by default: DESIGN.md 8) */

#define PTX_FLAG_NO_BOX_CLIP 16u /* traverse every ray over its whole [t_min, t_max] as the reference does, instead of only over its
                                    passage through the (widened) bounding box of the triangles: same hits, more node visits -
                                    the counting pass that prices the reference algorithm, and the A/B of the tests */

const char* ptx_last_error(void);
const char* ptx_version(void);

/* ---- host side: the parts that stay on the CPU ------------------------------------------------ */

/* `build_kdtree(&[impl Geometry], &SahCost) -> KdNode` (kdtree/src/build.rs:33-41), emitted directly in
 * the flattened layout.  Triangles are 9 floats each (v0 v1 v2); sah == NULL uses the defaults.  The
 * returned tree owns its arrays; ptx_kdtree_view exposes them (valid until ptx_kdtree_free). */
typedef struct ptx_kdtree ptx_kdtree;
int ptx_kdtree_build(const float* triangles, uint32_t n_triangles, const ptx_sah_cost* sah,
                     ptx_kdtree** out);
int ptx_kdtree_view(const ptx_kdtree* tree, const ptx_kdnode** nodes, uint32_t* n_nodes,
                    const uint32_t** leaf_indices, uint32_t* n_leaf_indices);
void ptx_kdtree_free(ptx_kdtree* tree);

/* Tree interchange and statistics.  Text is malloc'ed: release with ptx_free.
 * ptx_kdtree_write_json = `write_tree_json` (kdtree/src/format.rs:116-130): one line
 *   {"triangles": [[[x, y, z], [..], [..]], ..], "root": NODE}  with NODE = [i, j, ..] for a leaf or
 *   {"axis": "X", "distance": d, "left": NODE, "right": NODE}; numbers exactly as Rust prints them.
 * ptx_kdtree_read_json parses that format (e.g. a tree dumped by the reference's `kdtree-cli -j`) into the
 *   flattened layout; the triangles come back as 9 floats each (release with ptx_free).
 * ptx_kdtree_write_pretty = `write_node_pretty` (format.rs:22-47), kdtree-cli's default output.
 * ptx_kdtree_stats = `statistics` + `tree_cost` of kdtree-cli (kdtree-cli/src/main.rs:45-159). */
typedef struct ptx_tree_stats {
    float sah_cost;
    uint64_t geometries, node_count /* inner nodes + leaves */, leaf_count;
    uint64_t leaf_depth_min, leaf_depth_max, leaf_depth_total; /* the root has depth 1 */
    float leaf_depth_mean, leaf_depth_median;
    uint64_t leaf_geometries_min, leaf_geometries_max, leaf_geometries_total;
    float leaf_geometries_mean, leaf_geometries_median;
} ptx_tree_stats;
int ptx_kdtree_write_json(const float* triangles, uint32_t n_triangles, const ptx_kdnode* nodes, uint32_t n_nodes,
                          const uint32_t* leaf_indices, uint32_t n_leaf_indices, char** text_out, size_t* len_out);
int ptx_kdtree_read_json(const char* text, size_t len, ptx_kdtree** tree_out, float** triangles_out,
                         uint32_t* n_triangles_out);
int ptx_kdtree_write_pretty(const ptx_kdnode* nodes, uint32_t n_nodes, const uint32_t* leaf_indices,
                            uint32_t n_leaf_indices, char** text_out, size_t* len_out);
int ptx_kdtree_stats(const float* triangles, uint32_t n_triangles, const ptx_kdnode* nodes, uint32_t n_nodes,
                     const uint32_t* leaf_indices, uint32_t n_leaf_indices, const ptx_sah_cost* sah, ptx_tree_stats* out);

/* `KdNode::intersect` for ONE ray on the host, over the flattened tree — the scalar form of the accel
 * seam (kdtree/src/lib.rs:73-132), for callers that need an occasional query without a GPU round trip.  The
 * tree is validated first (PTX_E_ARG when malformed); leaf_indices must cover every leaf range the tree names
 * (the signature carries no length for it).  Only the triangles of the visited leaves are touched. */
int ptx_kdtree_intersect_host(const float* triangles, uint32_t n_triangles, const ptx_kdnode* nodes,
                              uint32_t n_nodes, const uint32_t* leaf_indices, const ptx_ray* ray,
                              ptx_hit* hit);

/* `Pinhole::new(Camera::new(position,target,up,fov), size)` (tracing/src/camera.rs:15-24, 71-85). */
int ptx_pinhole_make(const float position[3], const float target[3], const float up[3],
                     float fov_degrees, uint32_t width, uint32_t height, ptx_pinhole* out);

/* `Material::load_from_mtl` arithmetic (tracing/src/material.rs:153-164) without the texture load. */
int ptx_material_make(const float kd[3], const float ks[3], float ni, float pm, float transparency,
                      int32_t texture, ptx_material* out);

/* `ImageBuffer::to_rgb8(iterations as u16)` (tracing/src/image_buffer.rs:29-38; tracing/src/worker.rs:113
 * truncates the iteration count to u16, kept). */
int ptx_tonemap_rgb8(const float* accum_rgb, uint32_t width, uint32_t height, uint32_t iterations,
                     uint8_t* out_rgb8);

/* Loader: `read_obj_and_mtl_with_print_logging` + `from_wavefront` + `setup_scene`
 * (wavefront/src/lib.rs:10-30, tracing/src/properties.rs:49-90, pathtracer-cli/src/main.rs:123-159)
 * without the device upload.  The loaded scene owns its arrays; ptx_loaded_desc fills a ptx_scene_desc
 * pointing into it (valid until ptx_loaded_free).  `verbose` prints the reference's "Loading ..." lines. */
typedef struct ptx_loaded ptx_loaded;
int ptx_load_obj(const char* obj_path, const ptx_sah_cost* sah, uint32_t max_bounces, int verbose,
                 ptx_loaded** out);
int ptx_loaded_desc(const ptx_loaded* loaded, ptx_scene_desc* desc);
/* camera 0 of the MTL as 10 floats: position, target, up, fov (wavefront/src/mtl.rs:46-52) */
int ptx_loaded_camera(const ptx_loaded* loaded, float out10[10]);
void ptx_loaded_free(ptx_loaded* loaded);

/* PNG in/out for the CLI (`image::open` in material.rs:143, `save_with_format` in
 * pathtracer-cli/src/main.rs:186-188).  8-bit RGB/RGBA/grey non-interlaced only. */
int ptx_png_write_rgb8(const char* path, const uint8_t* rgb, uint32_t width, uint32_t height);
int ptx_png_read_linear(const char* path, float** rgb_out, uint32_t* width, uint32_t* height);
void ptx_free(void* p);

/* Accumulation-buffer checkpoint / resume (no counterpart in the reference, where an interrupted run loses its
 * samples).  The file holds the image size, the seed, the number of samples already added and the
 * f32 sums with a checksum.  Samples are keyed by (seed, pixel, sample index) and reach a pixel in index order,
 * so ptx_render(first_sample = samples) on top of a loaded buffer ends with the bits of one uninterrupted run.
 * ptx_accum_load with accum_rgb == NULL only reads the header into width/height/seed/samples; with a buffer,
 * width and height are inputs that the file must match. */
int ptx_accum_save(const char* path, const float* accum_rgb, uint32_t width, uint32_t height, uint64_t seed,
                   uint32_t samples);
int ptx_accum_load(const char* path, float* accum_rgb, uint32_t* width, uint32_t* height, uint64_t* seed,
                   uint32_t* samples);

/* ---- device side: the hot path ----------------------------------------------------------------- */

/* Upload a scene to the current HIP device. */
int ptx_scene_create(const ptx_scene_desc* desc, uint32_t flags, ptx_scene** out);
/* Same on HIP device `device` (the caller's current device is restored); ptx_device_count = visible GPUs.
 * The reference's analogue of a device is a render thread (tracing/src/worker.rs:95-104). */
int ptx_scene_create_on(const ptx_scene_desc* desc, uint32_t flags, int device, ptx_scene** out);
int ptx_device_count(int* count);
int ptx_scene_device(const ptx_scene* scene, int* device);
void ptx_scene_destroy(ptx_scene* scene);
/* Device memory behaviour (no counterpart in the reference).  Queues are allocated by the first render that
 * needs them, sized for that call, and kept for the next; without ptx_set_queue_slots their budget is half of
 * the device memory the scene could have when the call starts (free + already held by it).  ptx_scene_trim
 * hands everything but the scene itself back to the device (the next render allocates again);
 * ptx_scene_memory reports the bytes of render state held right now. */
int ptx_scene_trim(ptx_scene* scene);
int ptx_scene_memory(ptx_scene* scene, uint64_t* queue_bytes);

/* `KdNode::intersect(&self, geometries, ray, t_range)` in bulk (kdtree/src/lib.rs:73-78) =
 * `GeometryCollection::intersect` (tracing/src/collections.rs:43-49).  Host buffers; synchronous. */
int ptx_intersect_batch(ptx_scene* scene, const ptx_ray* rays, size_t n, ptx_hit* hits);
/* Same with device-resident rays/hits, asynchronous on `stream` (a hipStream_t, NULL = default).
 * The *_batch_device calls of one handle share one work cursor and one stack-spill area, so the library
 * serialises them: each call makes its stream wait for the event the previous call recorded, whatever the two
 * streams are.  Launches of one handle therefore never overlap; use one handle per stream to overlap them. */
int ptx_intersect_batch_device(ptx_scene* scene, const ptx_ray* d_rays, size_t n, ptx_hit* d_hits,
                               void* stream);
/* ptx_intersect_batch_device for callers that cannot allocate (HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE.  Same kernel and same results; the work
 * cursor and the stack-spill area live in a file-scope device array of the library (34.0 MiB per GPU, held
 * until the process ends) instead of scene-owned allocations, and the call creates no event and takes no
 * timing stamp: it enqueues one 1-KB memset (the launch's cursor words) and one kernel on `stream`.  At most PTX_NOALLOC_MAX_RAYS rays
 * per call (PTX_E_LIMIT beyond).  All handles on one GPU share that array: calls that overlap on one GPU
 * (two streams, two handles, two threads) race - the caller orders them. */
#define PTX_NOALLOC_MAX_RAYS (1u << 24)
int ptx_intersect_batch_device_noalloc(ptx_scene* scene, const ptx_ray* d_rays, size_t n, ptx_hit* d_hits,
                                       void* stream);
/* Occlusion only: out[i] = 1 iff `intersect(ray, t_range).is_some()` — the only thing the integrator
 * asks of a shadow ray (tracing/src/pathtracer.rs:72-78).  Device buffers. */
int ptx_occluded_batch_device(ptx_scene* scene, const ptx_ray* d_rays, size_t n, uint8_t* d_out,
                              void* stream);

/* `render_parallel_iterations` minus the tonemap (tracing/src/worker.rs:84-119 over
 * `Pathtracer::render_mut`, tracing/src/pathtracer.rs:131-141): renders global samples
 * [first_sample, first_sample + sample_count) of every pixel with the counter-based sampler of
 * ptx_sampling.h and ADDS the radiance sum into accum (width*height*3 floats, row-major, never
 * cleared by the callee — image_buffer semantics).  progress (optional) is called after each
 * completed batch with (samples completed so far, seconds since the call started, user) — the
 * analogue of `tx.send(duration)` (worker.rs:47). */
typedef void (*ptx_progress_cb)(uint32_t samples_done, double seconds, void* user);
int ptx_render(ptx_scene* scene, const ptx_pinhole* camera, uint64_t seed, uint32_t first_sample,
               uint32_t sample_count, float* accum_host, ptx_progress_cb progress, void* user);
/* Same with a device-resident accumulation buffer, asynchronous on `stream`; this is what a
 * multi-GPU driver calls before its RCCL reduce of the per-GPU buffers. */
int ptx_render_device(ptx_scene* scene, const ptx_pinhole* camera, uint64_t seed, uint32_t first_sample,
                      uint32_t sample_count, float* d_accum, void* stream);
/* `render_parallel_iterations` over SEVERAL GPUs in one process (tracing/src/worker.rs:84-119: T threads with
 * private buffers, then `buffers.reduce(Add::add)`, :106-107): scenes[g] is the same scene uploaded to GPU g
 * (ptx_scene_create_on).  Samples [first_sample, first_sample + sample_count) are cut into n_scenes contiguous
 * ranges (ptx_shard_samples), one host thread per GPU renders its range into a device buffer of its own, the
 * buffers are summed onto scenes[0]'s GPU and the result replaces accum_host (which goes in as the starting
 * value of shard 0, so the image-buffer "add into" semantics hold and one scene gives the bits of ptx_render).
 * reduce: PTX_REDUCE_AUTO = RCCL (`ncclReduce`, sum, root 0; librccl is loaded at first use) when the scenes
 * sit on distinct GPUs - falling back to PTX_REDUCE_PEER when librccl cannot be loaded or its communicators do
 * not come up - else PTX_REDUCE_PEER; PTX_REDUCE_RCCL forces RCCL (also with one scene: a one-rank
 * communicator) and fails hard without it; PTX_REDUCE_PEER = peer copies to GPU 0 + an add kernel per shard, in shard order (the
 * reference's left fold; also the only way when several scenes share a GPU).  progress is called from the
 * render threads, one call at a time, with the samples completed over all shards.
 * ptx_multi_last_reduce_ms: duration of the most recent reduce of this thread's ptx_render_multi call.
 * ptx_multi_shutdown: destroys the RCCL communicators the library cached (one set per device list, kept across calls
 * because ncclCommInitAll costs ~0.5 s); call it when no ptx_render_multi is running, e.g. before unloading the library.
 * The distinct-GPU paths (ncclCommInitAll over several devices, the grouped ncclReduce, hipMemcpyPeerAsync between
 * devices) have only run on one-GPU hosts so far - one-rank communicator, handles sharing a GPU: parity unpinned for
 * N > 1 distinct GPUs until an N-GPU run of tests/test_gpu_multi.py. */
#define PTX_REDUCE_AUTO 0u
#define PTX_REDUCE_RCCL 1u
#define PTX_REDUCE_PEER 2u
int ptx_render_multi(ptx_scene* const* scenes, uint32_t n_scenes, const ptx_pinhole* camera, uint64_t seed,
                     uint32_t first_sample, uint32_t sample_count, float* accum_host, uint32_t reduce,
                     ptx_progress_cb progress, void* user);
int ptx_multi_last_reduce_ms(double* ms);
int ptx_multi_shutdown(void);
/* The sample range of shard `shard` of `n_shards`: GPU g renders [first + g*S/N, first + (g+1)*S/N) — the
 * contiguous form of the reference's "every thread renders iterations_per_thread full frames"
 * (worker.rs:92-104).  Host arithmetic only. */
int ptx_shard_samples(uint32_t first_sample, uint32_t sample_count, uint32_t n_shards, uint32_t shard,
                      uint32_t* shard_first, uint32_t* shard_count);
/* Samples per batch (0 = as many as the path-id and queue-slot budgets allow: one batch if possible).  A batch
 * is rendered as one sequence of rounds over its own queues; results do not depend on the batching. */
int ptx_set_batch_samples(ptx_scene* scene, uint32_t samples_per_batch);
/* `Pathtracer::render_subdivided_mut` (tracing/src/pathtracer.rs:143-161, driven by
 * worker.rs:52-82 `render_parallel_subdivided`): adds the samples of the pixels of one rectangle only, into the
 * full-size width*height*3 buffer.  A pixel's samples do not depend on the tiling (the sampler is keyed by the
 * image pixel), so tiles rendered in any order add up to exactly what ptx_render_device gives. */
int ptx_render_region_device(ptx_scene* scene, const ptx_pinhole* camera, uint32_t x, uint32_t y, uint32_t width,
                             uint32_t height, uint64_t seed, uint32_t first_sample, uint32_t sample_count,
                             float* d_accum, void* stream);
/* Queue slots (paths in flight) over all pipelines; 0 = default: what fits in half of the device memory the
 * scene could have when a render call starts (162 bytes per slot with one light).  Setting fewer slots than
 * the scene holds releases its queues at once (ptx_scene_trim).  A batch with more paths
 * than slots is rendered with path regeneration: finished paths are replaced by the next camera rays of the
 * batch, so the memory stays bounded while every launch stays wide.  Results do not depend on it. */
int ptx_set_queue_slots(ptx_scene* scene, uint64_t slots);
/* Sample batches rendered concurrently, each on a HIP stream of its own (1..8, default 1 or the
 * PTX_PIPES environment variable).  With 1 every kernel has the GPU to itself (one wide batch was measured
 * faster than two narrower overlapped ones); results do not depend on it.  No counterpart in the reference. */
int ptx_set_pipelines(ptx_scene* scene, uint32_t pipelines);

/* `ImageBuffer::to_rgb8` on the device (tracing/src/image_buffer.rs:29-38): d_accum (width*height*3
 * floats) -> d_out (width*height*3 bytes), both on the current HIP device, asynchronous on `stream`.  Same rule as
 * ptx_tonemap_rgb8 including the u16 truncation of `iterations`; the power is taken in f64 and rounded once, so a
 * byte can differ from the host entry point's only where libm's powf is not correctly rounded. */
int ptx_tonemap_rgb8_device(const float* d_accum, uint32_t width, uint32_t height, uint32_t iterations,
                            uint8_t* d_out_rgb8, void* stream);
/* Queue lengths of the most recent render batch: extend_counts[r] / shadow_counts[r] = entries of round r+1
 * (one round = extend + shade + shadow over everything in flight; without path regeneration round r is
 * bounce r; shadow counts are paths: multiply by the number of lights).  n = capacity of each array. */
int ptx_last_batch_counts(ptx_scene* scene, uint32_t* extend_counts, uint32_t* shadow_counts, uint32_t n);
int ptx_stats(ptx_scene* scene, ptx_counters* out);
/* Kernel time of the extend / shadow launches per round since ptx_stats_reset (scenes created with
 * PTX_FLAG_TIME_KERNELS): extend_ms[r], shadow_ms[r] = round r + 1.  Without path regeneration round r is bounce r, so
 * round 1 is the camera rays: SURVEY 8d asks for camera, bounced and shadow rays separately.  n = array capacity. */
int ptx_round_times(ptx_scene* scene, double* extend_ms, double* shadow_ms, uint32_t n);
int ptx_stats_reset(ptx_scene* scene);

/* `kdtree-tester-cli test` on the GPU (kdtree-tester-cli/src/ray_tester.rs, ray_bouncer.rs:35-56): a scene
 * created with PTX_FLAG_CHECK_BRUTE intersects every extend and shadow ray of ptx_render a second time
 * against all triangles in index order (`intersect_closest_geometry`) and compares the two answers with
 * `CheckedIntersection::is_valid` (checked_intersection.rs:10-25: same index, |dt| < 1e-6, |du|,|dv| < 1e-5;
 * shadow rays: same yes/no).  ptx_check_report: rays checked / disagreements since the last ptx_stats_reset.
 * ptx_check_fail_log: the first disagreements (at most 65536) as the 50-byte records of
 * `CheckedIntersection::as_bytes` (checked_intersection.rs:27-56), readable by pybits/visualize_fails.py. */
int ptx_check_report(ptx_scene* scene, uint64_t* n_checked, uint64_t* n_fails);
int ptx_check_fail_log(ptx_scene* scene, uint8_t* out, uint32_t cap_records, uint32_t* n_written);
/* Ray log (tracing/src/raylogger.rs:9-72, written from pathtracer.rs:34-43 and 69-71 when the reference is
 * built with its `ray_logging` feature; worker.rs:18-25 names the files ./tmp/raylog{thread}.bin).
 * ptx_raylog_enable(max_records > 0) makes every following render append one record per extend ray - stretched
 * to its hit point (`ray.extended(t)`) when it hit - and per shadow ray, up to max_records (32 bytes of device
 * memory each); 0 switches the log off.  ptx_raylog_read hands out what was logged since the last read as the
 * reference's 33-byte records {iteration u16, pixel_x u16, pixel_y u16, bounces u8, shadow u8, intersect u8,
 * origin 3 x f32, direction 3 x f32} (raylogger.rs:54-66, little endian), in the order one reference thread
 * writes them (iteration = sample index, then pixel rows, columns, the path's bounces, extend ray before its
 * shadow rays), and empties the log.  n_dropped = records that did not fit max_records or cap_records. */
int ptx_raylog_enable(ptx_scene* scene, uint64_t max_records);
int ptx_raylog_read(ptx_scene* scene, uint8_t* out, uint64_t cap_records, uint64_t* n_written, uint64_t* n_dropped);
/* Diagnostic (scenes created with PTX_FLAG_COUNT_WORK; no counterpart in the reference): how the lanes of
 * the traversal kernels spent their loop iterations since the last reset.  out24[0..11] = extend,
 * out24[12..23] = shadow: wave iterations (one iteration = 3 rounds of 2 node steps + a leaf-exit step, then
 * up to 2 triangles per waiting lane), lane-visits of the node steps, wave iterations that ran the
 * triangle step, lane-visits of it, wave iterations that ran the leaf-exit step, lane-visits of it,
 * refill rounds, live lane-iterations, then the wave clock (s_memtime ticks, summed over waves) spent in the
 * refill, node, triangle and leaf-exit parts of the loop. */
int ptx_debug_loop_profile(ptx_scene* scene, uint64_t* out24);
/* Test hook (host only, no GPU call): the device layout ptx_scene_create derives from a flattened tree - 16-byte inner
 * nodes {bits of the plane distance, axis, descriptor of the left child, of the right child}, a descriptor being an
 * inner-node index (>= 0), 0x80000000 | offset of the leaf's first record, or -1 for an empty leaf; and the leaf
 * records to lay down, as triangle index | last-of-run << 31, where leaves with equal lists, and lists that are
 * the tail of a longer one, share a run.  `node_words` (4 per inner node, capacity 4 * n_nodes) and `records`
 * (capacity n_leaf_indices) may be NULL; counts = {inner nodes, records, root descriptor, nodes of the
 * breadth-first prefix that the kernels keep in LDS (at most top_nodes)}.  No counterpart in the reference. */
int ptx_debug_device_tree(const ptx_kdnode* nodes, uint32_t n_nodes, const uint32_t* leaf_indices, uint32_t n_leaf_indices,
                          uint32_t n_triangles, uint32_t top_nodes, uint32_t* node_words, uint32_t* records, uint32_t counts[4]);

#ifdef __cplusplus
}
#endif
#endif /* PTX_H */
