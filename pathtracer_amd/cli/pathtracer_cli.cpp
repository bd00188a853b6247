// pathtracer-cli for MI355X — same flags, defaults and stdout lines as the reference binary
// (pathtracer-cli/src/main.rs:53-85 flags, :87-121 progress line, :161-190 main), rendering through
// libptx_mi355x.so.  `-t threads` x `-n iterations-per-thread` keeps its meaning as the total sample
// count (there is no global spp flag in the reference, tracing/src/worker.rs:92); the GPU plays all threads.
// Extra long flags (not in the reference): --seed N, --device D, --batch-samples S, --stats, --checkpoint FILE / --resume FILE
// (accumulation buffer save / continue), --gpus N [--reduce rccl|peer] [--share-device] (the iterations are
// sharded over N GPUs, devices D .. D+N-1, and the per-GPU buffers summed with RCCL: ptx_render_multi; with
// --share-device all N shards run on device D, the rehearsal a one-GPU machine can do), and
// --ray-log FILE [--ray-log-max N]: what the reference writes to ./tmp/raylog0.bin when built with its
// `ray_logging` feature (raylogger.rs), for all iterations, as one thread would.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ptx.h"

namespace {

struct Args {
    std::string input, output;
    uint32_t width = 512, height = 512;
    uint32_t max_bounces = 10, iterations_per_thread = 4, threads = 1;
    ptx_sah_cost sah{1.0f, 1.5f, 0.8f};
    uint64_t seed = 0;
    int device = 0;
    uint32_t batch_samples = 0;
    bool stats = false;
    uint32_t gpus = 1;
    uint32_t reduce = PTX_REDUCE_AUTO;
    bool share_device = false;
    bool multi = false;  // go through ptx_render_multi (implied by --gpus > 1; --reduce asks for it with one GPU too)
    std::string ray_log, checkpoint, resume;
    uint64_t ray_log_max = 1ull << 26;  // records held on the device per iteration
};

[[noreturn]] void usage(const char* why) {
    fprintf(stderr,
            "error: %s\n\nUsage: pathtracer-cli --input <INPUT> --output <OUTPUT> [OPTIONS]\n\n"
            "  -i, --input <INPUT>                    Wavefront OBJ input path\n"
            "  -o, --output <OUTPUT>                  PNG output path\n"
            "  -s, --size <SIZE>                      Image size in pixels [default: 512x512]\n"
            "  -m, --max-bounces <N>                  Max number of bounces [default: 10]\n"
            "  -n, --iterations-per-thread <N>        Iterations to execute per thread [default: 4]\n"
            "  -t, --threads <N>                      Number of threads [default: 1]\n"
            "      --traverse-cost <F>                SAH kd-tree traverse cost [default: 1]\n"
            "      --intersect-cost <F>               SAH kd-tree intersect cost [default: 1.5]\n"
            "      --empty-factor <F>                 SAH kd-tree empty factor [default: 0.8]\n"
            "      --seed <N> --device <D> --batch-samples <S> --stats   (MI355X extras)\n"
            "      --gpus <N> [--reduce rccl|peer] [--share-device]      shard the iterations over N GPUs, sum with RCCL\n"
            "      --ray-log <FILE> [--ray-log-max <RECORDS>]           33-byte ray records (raylogger.rs)\n"
            "      --checkpoint <FILE>                save the accumulation buffer after the render\n"
            "      --resume <FILE>                    continue a saved buffer (its seed; iterations are added)\n",
            why);
    exit(2);
}

struct Progress {
    uint32_t threads, total, last_done = 0;
    double last_secs = 0, sum = 0, sum_sq = 0;
    uint32_t completed = 0;
};

// printer_thread (main.rs:87-121): one line, rewritten in place.  A batch of k samples that took d
// seconds is reported as k iterations of d/k seconds each.
void on_progress(uint32_t done, double secs, void* user) {
    Progress* p = (Progress*)user;
    uint32_t k = done - p->last_done;
    if (k == 0) return;
    double each = (secs - p->last_secs) / k;
    p->last_done = done;
    p->last_secs = secs;
    for (uint32_t i = 0; i < k; ++i) {
        p->sum += each;
        p->sum_sq += each * each;
        ++p->completed;
    }
    double mean = p->sum / p->completed;
    double var = p->sum_sq / p->completed - mean * mean;
    double sdev = std::sqrt(var > 0 ? var : 0);
    double eta = (double)(p->total - p->completed) * mean;  // one GPU renders the iterations back to back
    printf("\x1B[1K\x1B[1G[%u/%u] mean: %.2fs, sdev: %.2fs, eta: %.2fs", p->completed, p->total, mean, sdev, eta);
    fflush(stdout);
}

}  // namespace

int main(int argc, char** argv) {
    Args a;
    for (int i = 1; i < argc; ++i) {
        std::string k = argv[i];
        auto val = [&]() -> const char* {
            if (i + 1 >= argc) usage(("missing value for " + k).c_str());
            return argv[++i];
        };
        if (k == "-i" || k == "--input") a.input = val();
        else if (k == "-o" || k == "--output") a.output = val();
        else if (k == "-s" || k == "--size") {
            const char* s = val();
            const char* x = strchr(s, 'x');
            if (!x) usage("Could not parse size (expected WxH)");
            a.width = (uint32_t)strtoul(s, nullptr, 10);
            a.height = (uint32_t)strtoul(x + 1, nullptr, 10);
        } else if (k == "-m" || k == "--max-bounces") a.max_bounces = (uint32_t)strtoul(val(), nullptr, 10);
        else if (k == "-n" || k == "--iterations-per-thread") a.iterations_per_thread = (uint32_t)strtoul(val(), nullptr, 10);
        else if (k == "-t" || k == "--threads") a.threads = (uint32_t)strtoul(val(), nullptr, 10);
        else if (k == "--traverse-cost") a.sah.traverse_cost = strtof(val(), nullptr);
        else if (k == "--intersect-cost") a.sah.intersect_cost = strtof(val(), nullptr);
        else if (k == "--empty-factor") a.sah.empty_factor = strtof(val(), nullptr);
        else if (k == "--seed") a.seed = strtoull(val(), nullptr, 10);
        else if (k == "--device") a.device = atoi(val());
        else if (k == "--batch-samples") a.batch_samples = (uint32_t)strtoul(val(), nullptr, 10);
        else if (k == "--stats") a.stats = true;
        else if (k == "--gpus") a.gpus = (uint32_t)strtoul(val(), nullptr, 10);
        else if (k == "--share-device") a.share_device = true;
        else if (k == "--reduce") {
            a.multi = true;
            const std::string r = val();
            if (r == "rccl") a.reduce = PTX_REDUCE_RCCL;
            else if (r == "peer") a.reduce = PTX_REDUCE_PEER;
            else if (r == "auto") a.reduce = PTX_REDUCE_AUTO;
            else usage("--reduce takes rccl, peer or auto");
        }
        else if (k == "--ray-log") a.ray_log = val();
        else if (k == "--checkpoint") a.checkpoint = val();
        else if (k == "--resume") a.resume = val();
        else if (k == "--ray-log-max") a.ray_log_max = strtoull(val(), nullptr, 10);
        else usage(("unexpected argument '" + k + "'").c_str());
    }
    if (a.input.empty()) usage("the following required arguments were not provided: --input <INPUT>");
    if (a.output.empty()) usage("the following required arguments were not provided: --output <OUTPUT>");
    if (a.max_bounces > 255) usage("max-bounces is a u8");
    if (a.gpus < 1 || a.gpus > 64) usage("--gpus takes 1..64");
    if (a.gpus > 1) a.multi = true;
    if (a.multi && !a.ray_log.empty()) usage("--ray-log writes one thread's file: use one GPU and no --reduce");

    auto die = [](const char* what) {
        fprintf(stderr, "%s: %s\n", what, ptx_last_error());
        exit(1);
    };
    ptx_loaded* loaded = nullptr;
    if (ptx_load_obj(a.input.c_str(), &a.sah, a.max_bounces, 1, &loaded)) die("load");
    ptx_scene_desc desc;
    ptx_loaded_desc(loaded, &desc);
    float cam[10];
    ptx_loaded_camera(loaded, cam);
    ptx_pinhole pinhole;
    if (ptx_pinhole_make(cam, cam + 3, cam + 6, cam[9], a.width, a.height, &pinhole)) die("camera");
    int n_devices = 0;
    ptx_device_count(&n_devices);
    const uint32_t need = a.share_device ? 1u : a.gpus;
    if (a.device < 0 || (uint32_t)a.device + need > (uint32_t)n_devices) {
        fprintf(stderr, "no usable HIP device(s) %d..%u: %d visible (this build has no CPU fallback)\n", a.device,
                (uint32_t)a.device + need - 1, n_devices);
        return 1;
    }
    std::vector<ptx_scene*> scenes(a.gpus, nullptr);  // the scene replicated on every GPU
    for (uint32_t g = 0; g < a.gpus; ++g) {
        if (ptx_scene_create_on(&desc, a.stats ? PTX_FLAG_TIME_KERNELS : 0, a.device + (a.share_device ? 0 : (int)g), &scenes[g])) die("upload");
        if (a.batch_samples) ptx_set_batch_samples(scenes[g], a.batch_samples);
    }
    ptx_scene* scene = scenes[0];

    const uint32_t total = a.threads * a.iterations_per_thread;
    printf("Rendering %ux%u px image with %u thread(s) and %u total iteration(s)...\n", a.width, a.height, a.threads, total);
    if (a.multi) printf("Sharding the iterations over %u GPU(s)%s\n", a.gpus, a.share_device ? " (all on one device)" : "");
    std::vector<float> accum((size_t)a.width * a.height * 3, 0.0f);
    uint32_t first = 0;  // samples already in the buffer
    if (!a.resume.empty()) {
        uint32_t w = a.width, h = a.height;
        if (ptx_accum_load(a.resume.c_str(), accum.data(), &w, &h, &a.seed, &first)) die("resume");
        printf("Resuming %s: %u iteration(s) done, seed %llu\n", a.resume.c_str(), first, (unsigned long long)a.seed);
    }
    Progress prog{a.threads, total};
    auto t0 = std::chrono::steady_clock::now();
    if (a.multi) {
        if (ptx_render_multi(scenes.data(), a.gpus, &pinhole, a.seed, first, total, accum.data(), a.reduce, on_progress, &prog)) die("render");
    } else if (a.ray_log.empty()) {
        if (ptx_render(scene, &pinhole, a.seed, first, total, accum.data(), on_progress, &prog)) die("render");
    } else {  // one iteration at a time, so that the log of one iteration is all the device has to hold
        FILE* f = fopen(a.ray_log.c_str(), "wb");
        if (!f) {
            perror(a.ray_log.c_str());
            return 1;
        }
        if (a.ray_log_max == 0 || ptx_raylog_enable(scene, a.ray_log_max)) die("ray log");
        std::vector<uint8_t> bytes((size_t)a.ray_log_max * 33);
        uint64_t dropped_total = 0;
        for (uint32_t it = 0; it < total; ++it) {
            if (ptx_render(scene, &pinhole, a.seed, first + it, 1, accum.data(), nullptr, nullptr)) die("render");
            uint64_t n = 0, dropped = 0;
            if (ptx_raylog_read(scene, bytes.data(), a.ray_log_max, &n, &dropped)) die("ray log");
            fwrite(bytes.data(), 33, (size_t)n, f);
            dropped_total += dropped;
            on_progress(it + 1, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), &prog);
        }
        fclose(f);
        if (dropped_total) fprintf(stderr, "\nray log: %llu records dropped (raise --ray-log-max)\n", (unsigned long long)dropped_total);
    }
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("\nTotal time: %.2fs\n", secs);
    if (a.multi) {
        double ms = 0;
        ptx_multi_last_reduce_ms(&ms);
        printf("Reduce over %u GPU(s): %.2f ms\n", a.gpus, ms);
    }
    if (a.stats) {
        ptx_counters c{};
        for (ptx_scene* s : scenes) {  // sums over the GPUs; the kernel times add up across devices
            ptx_counters k;
            ptx_stats(s, &k);
            c.samples += k.samples, c.extend_rays += k.extend_rays, c.shadow_rays += k.shadow_rays;
            c.extend_ms += k.extend_ms, c.shadow_ms += k.shadow_ms, c.shade_ms += k.shade_ms, c.generate_ms += k.generate_ms;
        }
        printf("  samples: %llu (%.3f Msamples/s)\n  extend rays: %llu in %.1f ms (%.1f Mrays/s)\n  shadow rays: %llu in %.1f ms (%.1f Mrays/s)\n"
               "  shade %.1f ms, generate %.1f ms\n",
               (unsigned long long)c.samples, c.samples / secs / 1e6, (unsigned long long)c.extend_rays, c.extend_ms,
               c.extend_ms > 0 ? c.extend_rays / c.extend_ms / 1e3 : 0.0, (unsigned long long)c.shadow_rays, c.shadow_ms,
               c.shadow_ms > 0 ? c.shadow_rays / c.shadow_ms / 1e3 : 0.0, c.shade_ms, c.generate_ms);
    }
    if (!a.checkpoint.empty() && ptx_accum_save(a.checkpoint.c_str(), accum.data(), a.width, a.height, a.seed, first + total))
        die("checkpoint");
    printf("Writing %s...\n", a.output.c_str());
    std::vector<uint8_t> rgb(accum.size());
    ptx_tonemap_rgb8(accum.data(), a.width, a.height, first + total, rgb.data());
    if (ptx_png_write_rgb8(a.output.c_str(), rgb.data(), a.width, a.height)) die("write");
    for (ptx_scene* s : scenes) ptx_scene_destroy(s);
    ptx_loaded_free(loaded);
    return 0;
}
