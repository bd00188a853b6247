// ptx_multi.hip — sample-sharded rendering over several GPUs of one node, behind the C ABI.
//
// The reference's driver (`render_parallel_iterations`, tracing/src/worker.rs:84-119) spawns T threads, gives
// each a private RNG and a private ImageBuffer, lets each render its iterations, joins them and folds the
// buffers with `buffers.reduce(Add::add)` (:106-107).  Here a GPU plays a group of threads: one host thread
// per GPU renders a contiguous range of the global sample indices into a device buffer of its own (the sampler
// is keyed by (seed, pixel, global sample index), so the image does not depend on the sharding except for the
// order of the final sum), and the buffers are summed onto GPU 0:
//   RCCL   one ncclReduce(sum, f32, root 0) per render, all ranks driven from this process inside one group
//          call (communicators from ncclCommInitAll, cached per device list).  librccl is loaded at first use
//          with dlopen: a process that never renders on several GPUs never maps it, and inside a PyTorch
//          process the RCCL that torch already mapped is the one that is found (same soname).
//   PEER   hipMemcpyPeerAsync of every shard's buffer to GPU 0 + one add kernel per shard, in shard order: the
//          reference's left fold, bit-reproducible; also what runs when several scenes share one GPU (RCCL
//          refuses two ranks on one device), which is how a one-GPU box exercises the whole path.
// The path shards by independent samples: there is no data-path collective, only this final sum (14 MB at
// 1080 x 1080, 99.5 MB at 3840 x 2160).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and enums only: every function is resolved with dlsym

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>

#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ptx.h"
#include "device_internal.hpp"
#include "host_internal.hpp"

using namespace ptx;

namespace {

// dst[i] += src[i]: `ImageBuffer + ImageBuffer` (tracing/src/image_buffer.rs:79-95).  HBM-bound streaming:
// 12 bytes per float moved (two reads, one write), float4 per lane.
__global__ __launch_bounds__(256) void k_add(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
    const size_t n4 = n / 4, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 a = reinterpret_cast<float4*>(dst)[i];
        const float4 b = reinterpret_cast<const float4*>(src)[i];
        a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
        reinterpret_cast<float4*>(dst)[i] = a;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] += src[i];
}

int hip_fail(hipError_t e, const char* where) {
    return fail(e == hipErrorOutOfMemory ? PTX_E_OOM : PTX_E_HIP, std::string(where) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                       \
    do {                                                    \
        hipError_t e__ = (expr);                            \
        if (e__ != hipSuccess) return hip_fail(e__, #expr); \
    } while (0)

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;  // why it could not be loaded
};

std::mutex g_mu;  // guards the RCCL table and the communicator cache
Rccl g_rccl;
std::map<std::vector<int>, std::vector<ncclComm_t>> g_comms;

bool load_rccl() {  // g_mu held
    if (g_rccl.handle) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) {
        const char* e = dlerror();
        g_rccl.why = std::string("librccl could not be loaded: ") + (e ? e : "?");
        return false;
    }
    auto sym = [&](const char* name) -> void* {
        void* p = dlsym(g_rccl.handle, name);
        if (!p) g_rccl.why = std::string("librccl lacks ") + name;
        return p;
    };
    g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))sym("ncclCommInitAll");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.Reduce = (decltype(g_rccl.Reduce))sym("ncclReduce");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    if (!g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.Reduce || !g_rccl.GroupStart || !g_rccl.GroupEnd ||
        !g_rccl.GetErrorString) {
        dlclose(g_rccl.handle);
        g_rccl.handle = nullptr;
        return false;
    }
    return true;
}

int rccl_fail(ncclResult_t r, const char* where) {
    return fail(PTX_E_HIP, std::string(where) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error"));
}
#define RCCL_TRY(expr)                                        \
    do {                                                      \
        ncclResult_t r__ = (expr);                            \
        if (r__ != ncclSuccess) return rccl_fail(r__, #expr); \
    } while (0)

struct Shard {
    ptx_scene* scene = nullptr;
    int device = 0;
    float* accum = nullptr;
    float* tmp = nullptr;
    hipStream_t stream = nullptr;
    uint32_t first = 0, count = 0;
    int rc = PTX_OK;
    std::string error;
};

struct ProgressHub {  // folds the per-shard progress calls into the caller's one callback
    ptx_progress_cb cb = nullptr;
    void* user = nullptr;
    std::mutex mu;
    std::vector<uint32_t> done;
    std::chrono::steady_clock::time_point t0;
};
struct ProgressSlot {
    ProgressHub* hub;
    uint32_t shard;
};

void shard_progress(uint32_t samples_done, double, void* user) {
    ProgressSlot* s = (ProgressSlot*)user;
    std::lock_guard<std::mutex> lock(s->hub->mu);
    s->hub->done[s->shard] = samples_done;
    uint32_t total = 0;
    for (uint32_t d : s->hub->done) total += d;
    s->hub->cb(total, std::chrono::duration<double>(std::chrono::steady_clock::now() - s->hub->t0).count(), s->hub->user);
}

thread_local double t_last_reduce_ms = 0.0;

int reduce_peer(std::vector<Shard>& sh, size_t n_floats) {
    Shard& root = sh[0];
    HIP_TRY(hipSetDevice(root.device));
    const int grid = 2048;
    for (size_t g = 1; g < sh.size(); ++g) {  // left fold in shard order
        if (sh[g].count == 0) continue;       // that buffer is all zeros
        const float* src = sh[g].accum;
        if (sh[g].device != root.device) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, root.device, sh[g].device) == hipSuccess && can) {
                const hipError_t e = hipDeviceEnablePeerAccess(sh[g].device, 0);  // direct xGMI copy; else staged by the runtime
                if (e != hipSuccess) (void)hipGetLastError();                     // "already enabled" is fine
            }
            HIP_TRY(hipMemcpyPeerAsync(root.tmp, root.device, sh[g].accum, sh[g].device, n_floats * sizeof(float), root.stream));
            src = root.tmp;
        }
        k_add<<<grid, 256, 0, root.stream>>>(root.accum, src, n_floats);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(root.stream));
    return PTX_OK;
}

constexpr int PTX_E_NO_RCCL = -1000;  // internal: RCCL could not be set up (nothing was enqueued yet)

int reduce_rccl(std::vector<Shard>& sh, size_t n_floats, bool soft_setup) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (!load_rccl()) return fail(soft_setup ? PTX_E_NO_RCCL : PTX_E_HIP, "ptx_render_multi: " + g_rccl.why);
    std::vector<int> devs;
    for (auto& s : sh) devs.push_back(s.device);
    auto it = g_comms.find(devs);
    if (it == g_comms.end()) {
        std::vector<ncclComm_t> comms(devs.size());
        const ncclResult_t r = g_rccl.CommInitAll(comms.data(), (int)devs.size(), devs.data());
        if (r != ncclSuccess) {
            const int rc = rccl_fail(r, "ncclCommInitAll");
            return soft_setup ? PTX_E_NO_RCCL : rc;
        }
        it = g_comms.emplace(devs, std::move(comms)).first;
    }
    std::vector<ncclComm_t>& comms = it->second;
    RCCL_TRY(g_rccl.GroupStart());
    for (size_t g = 0; g < sh.size(); ++g) {
        HIP_TRY(hipSetDevice(sh[g].device));
        // in place on the root; the receive buffer of the other ranks is not written
        const ncclResult_t r = g_rccl.Reduce(sh[g].accum, sh[g].accum, n_floats, ncclFloat32, ncclSum, 0, comms[g], sh[g].stream);
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            return rccl_fail(r, "ncclReduce");
        }
    }
    RCCL_TRY(g_rccl.GroupEnd());
    for (auto& s : sh) {
        HIP_TRY(hipSetDevice(s.device));
        HIP_TRY(hipStreamSynchronize(s.stream));
    }
    return PTX_OK;
}

}  // namespace

extern "C" {

int ptx_shard_samples(uint32_t first_sample, uint32_t sample_count, uint32_t n_shards, uint32_t shard, uint32_t* shard_first,
                      uint32_t* shard_count) {
    if (!shard_first || !shard_count) return fail(PTX_E_ARG, "ptx_shard_samples: null argument");
    if (n_shards == 0 || shard >= n_shards) return fail(PTX_E_ARG, "ptx_shard_samples: shard outside [0, n_shards)");
    if ((uint64_t)first_sample + sample_count > 0xFFFFFFFFull) return fail(PTX_E_LIMIT, "ptx_shard_samples: sample index overflow");
    const uint64_t lo = (uint64_t)sample_count * shard / n_shards, hi = (uint64_t)sample_count * (shard + 1) / n_shards;
    *shard_first = first_sample + (uint32_t)lo;
    *shard_count = (uint32_t)(hi - lo);
    return PTX_OK;
}

int ptx_multi_last_reduce_ms(double* ms) {
    if (!ms) return fail(PTX_E_ARG, "ptx_multi_last_reduce_ms: null argument");
    *ms = t_last_reduce_ms;
    return PTX_OK;
}

int ptx_multi_shutdown(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    int rc = PTX_OK;
    int before = 0;
    (void)hipGetDevice(&before);
    for (auto& kv : g_comms)
        for (size_t g = 0; g < kv.second.size(); ++g) {
            (void)hipSetDevice(kv.first[g]);
            if (g_rccl.CommDestroy && g_rccl.CommDestroy(kv.second[g]) != ncclSuccess) rc = fail(PTX_E_HIP, "ptx_multi_shutdown: ncclCommDestroy failed");
        }
    g_comms.clear();
    (void)hipSetDevice(before);
    return rc;
}

static int render_multi_impl(ptx_scene* const* scenes, uint32_t n_scenes, const ptx_pinhole* cam, uint64_t seed, uint32_t first_sample,
                             uint32_t sample_count, float* accum_host, uint32_t reduce, ptx_progress_cb progress, void* user);

int ptx_render_multi(ptx_scene* const* scenes, uint32_t n_scenes, const ptx_pinhole* cam, uint64_t seed, uint32_t first_sample,
                     uint32_t sample_count, float* accum_host, uint32_t reduce, ptx_progress_cb progress, void* user) {
    return guarded("ptx_render_multi", [&]() -> int {
        return render_multi_impl(scenes, n_scenes, cam, seed, first_sample, sample_count, accum_host, reduce, progress, user);
    });
}

static int render_multi_impl(ptx_scene* const* scenes, uint32_t n_scenes, const ptx_pinhole* cam, uint64_t seed, uint32_t first_sample,
                             uint32_t sample_count, float* accum_host, uint32_t reduce, ptx_progress_cb progress, void* user) {
    if (!scenes || n_scenes == 0) return fail(PTX_E_ARG, "ptx_render_multi: no scenes");
    if (n_scenes > 64) return fail(PTX_E_LIMIT, "ptx_render_multi: more than 64 scenes");
    if (!cam || !accum_host) return fail(PTX_E_ARG, "ptx_render_multi: null argument");
    if (reduce > PTX_REDUCE_PEER) return fail(PTX_E_ARG, "ptx_render_multi: unknown reduce method");
    const size_t n_floats = (size_t)cam->width * cam->height * 3;
    if (n_floats == 0) return fail(PTX_E_ARG, "ptx_render_multi: empty image");
    if ((uint64_t)first_sample + sample_count > 0xFFFFFFFFull) return fail(PTX_E_LIMIT, "ptx_render_multi: sample index overflow");
    std::vector<Shard> sh(n_scenes);
    bool distinct = true;
    for (uint32_t g = 0; g < n_scenes; ++g) {
        if (!scenes[g]) return fail(PTX_E_ARG, "ptx_render_multi: null scene");
        for (uint32_t k = 0; k < g; ++k)
            if (scenes[k] == scenes[g]) return fail(PTX_E_ARG, "ptx_render_multi: the same handle twice (a handle is not re-entrant)");
        sh[g].scene = scenes[g];
        sh[g].device = internal_device(scenes[g]);
        for (uint32_t k = 0; k < g; ++k) distinct = distinct && sh[k].device != sh[g].device;
        int rc = ptx_shard_samples(first_sample, sample_count, n_scenes, g, &sh[g].first, &sh[g].count);
        if (rc) return rc;
    }
    uint32_t how = reduce;
    if (how == PTX_REDUCE_AUTO) how = (distinct && n_scenes > 1) ? PTX_REDUCE_RCCL : PTX_REDUCE_PEER;
    if (how == PTX_REDUCE_RCCL && !distinct)
        return fail(PTX_E_ARG, "ptx_render_multi: RCCL needs every scene on a GPU of its own (use PTX_REDUCE_PEER)");
    struct DeviceGuard {  // whatever happens below, the caller's current device is what it was
        int before = 0;
        ~DeviceGuard() { (void)hipSetDevice(before); }
    } guard;
    HIP_TRY(hipGetDevice(&guard.before));
    for (uint32_t g = 0; g < n_scenes; ++g) {
        void* stream = nullptr;
        int rc = internal_multi_slot(sh[g].scene, n_floats, &sh[g].accum, g == 0 && how == PTX_REDUCE_PEER ? &sh[g].tmp : nullptr, &stream);
        if (rc) return rc;
        sh[g].stream = (hipStream_t)stream;
        // shard 0 starts from the caller's buffer (`ImageBuffer` is added into, never cleared), the others from zero
        if (g == 0) HIP_TRY(hipMemcpyAsync(sh[g].accum, accum_host, n_floats * sizeof(float), hipMemcpyHostToDevice, sh[g].stream));
        else HIP_TRY(hipMemsetAsync(sh[g].accum, 0, n_floats * sizeof(float), sh[g].stream));
    }
    ProgressHub hub;
    hub.cb = progress;
    hub.user = user;
    hub.done.assign(n_scenes, 0u);
    hub.t0 = std::chrono::steady_clock::now();
    std::vector<ProgressSlot> slots(n_scenes);
    auto work = [&](uint32_t g) {  // one host thread per GPU: the reference's render thread (worker.rs:27-50)
        Shard& s = sh[g];
        slots[g] = ProgressSlot{&hub, g};
        if (hipSetDevice(s.device) != hipSuccess) {
            s.rc = PTX_E_HIP;
            s.error = "ptx_render_multi: hipSetDevice failed in the render thread";
            return;
        }
        if (s.count) s.rc = internal_render(s.scene, cam, seed, s.first, s.count, s.accum, s.stream, progress ? shard_progress : nullptr, &slots[g]);
        if (s.rc == PTX_OK && hipStreamSynchronize(s.stream) != hipSuccess) s.rc = PTX_E_HIP, s.error = "ptx_render_multi: stream synchronisation failed";
        else if (s.rc != PTX_OK) s.error = ptx_last_error();  // thread-local: carried over to the caller's thread below
    };
    {
        // whatever happens while the threads are being started (std::system_error), the ones already running are
        // joined before this frame is left: a joinable std::thread must never be destroyed
        struct Joiner {
            std::vector<std::thread> threads;
            ~Joiner() {
                for (auto& t : threads)
                    if (t.joinable()) t.join();
            }
        } joiner;
        joiner.threads.reserve(n_scenes);
        for (uint32_t g = 1; g < n_scenes; ++g) joiner.threads.emplace_back(work, g);
        work(0);  // the calling thread renders shard 0
    }
    int rc = PTX_OK;
    for (auto& s : sh)
        if (s.rc != PTX_OK && rc == PTX_OK) rc = fail(s.rc, s.error.empty() ? "ptx_render_multi: a shard failed" : s.error);
    if (rc == PTX_OK) {
        const auto r0 = std::chrono::steady_clock::now();
        if (how == PTX_REDUCE_RCCL) {
            rc = reduce_rccl(sh, n_floats, /*setup_only=*/reduce == PTX_REDUCE_AUTO);
            if (rc == PTX_E_NO_RCCL) {
                // PTX_REDUCE_AUTO and librccl cannot be loaded or its communicators do not come up: the peer fold needs
                // neither (an explicit PTX_REDUCE_RCCL keeps the hard error)
                if (getenv("PTX_VERBOSE")) fprintf(stderr, "[ptx] RCCL unavailable (%s): summing with peer copies\n", ptx_last_error());
                rc = internal_multi_slot(sh[0].scene, n_floats, &sh[0].accum, &sh[0].tmp, (void**)&sh[0].stream);
                if (rc == PTX_OK) rc = reduce_peer(sh, n_floats);
            }
        } else if (n_scenes > 1) {
            rc = reduce_peer(sh, n_floats);
        }
        t_last_reduce_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r0).count();
    }
    if (rc == PTX_OK) {
        const hipError_t e1 = hipSetDevice(sh[0].device);
        const hipError_t e2 = e1 == hipSuccess ? hipMemcpy(accum_host, sh[0].accum, n_floats * sizeof(float), hipMemcpyDeviceToHost) : e1;
        if (e2 != hipSuccess) rc = hip_fail(e2, "ptx_render_multi: reading the summed buffer back");
    }
    return rc;
}

}  // extern "C"
