// kdtree-tester-cli for MI355X — the `test` subcommand of the reference tool (kdtree-tester-cli/src/main.rs:19-49,
// ray_tester.rs:13-82): compare the KD-tree intersection with the naive one on bounced rays and write the
// disagreements as 50-byte records (checked_intersection.rs:27-56).  Here the rays are those of one path-traced
// sample per pixel (extend and shadow rays of every bounce, ptx_render with PTX_FLAG_CHECK_BRUTE), both
// intersections run on the GPU, and every failing ray is reported (the reference stops a pixel at its first).
// The `reduce` subcommand (a delta debugger) is not offered.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ptx.h"

namespace {
[[noreturn]] void usage(const char* why) {
    fprintf(stderr,
            "error: %s\n\nUsage: kdtree-tester-cli test --input <INPUT> [OPTIONS]\n\n"
            "  -i, --input <INPUT>            Wavefront OBJ input path\n"
            "  -o, --output <OUTPUT>          Output ray fail binary data path\n"
            "  -s, --size <SIZE>              Image size in pixels [default: 512x512]\n"
            "  -b, --bounces <BOUNCES>        Max number of bounces to test [default: 10]\n"
            "      --traverse-cost <F>        SAH kd-tree traverse cost [default: 1]\n"
            "      --intersect-cost <F>       SAH kd-tree intersect cost [default: 1.5]\n"
            "      --empty-factor <F>         SAH kd-tree empty factor [default: 0.8]\n",
            why);
    exit(2);
}
void die(const char* what) {
    fprintf(stderr, "%s: %s\n", what, ptx_last_error());
    exit(1);
}
}  // namespace

int main(int argc, char** argv) {
    if (argc < 2 || std::string(argv[1]) != "test") usage("expected the subcommand `test` (`reduce` is not offered by this build)");
    std::string input, output;
    uint32_t width = 512, height = 512, bounces = 10;
    ptx_sah_cost sah{1.0f, 1.5f, 0.8f};
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i];
        auto val = [&]() -> const char* {
            if (i + 1 >= argc) usage(("missing value for " + a).c_str());
            return argv[++i];
        };
        if (a == "-i" || a == "--input") input = val();
        else if (a == "-o" || a == "--output") output = val();
        else if (a == "-s" || a == "--size") {
            if (sscanf(val(), "%ux%u", &width, &height) != 2 || !width || !height) usage("size must be WIDTHxHEIGHT");
        } else if (a == "-b" || a == "--bounces") bounces = (uint32_t)strtoul(val(), nullptr, 10);
        else if (a == "--traverse-cost") sah.traverse_cost = strtof(val(), nullptr);
        else if (a == "--intersect-cost") sah.intersect_cost = strtof(val(), nullptr);
        else if (a == "--empty-factor") sah.empty_factor = strtof(val(), nullptr);
        else usage(("unexpected argument " + a).c_str());
    }
    if (input.empty()) usage("the following required arguments were not provided: --input <INPUT>");
    if (bounces > 255) usage("at most 255 bounces");

    ptx_loaded* loaded = nullptr;
    if (ptx_load_obj(input.c_str(), &sah, bounces, 1, &loaded)) die("loading the scene");
    ptx_scene_desc d;
    if (ptx_loaded_desc(loaded, &d)) die("describing the scene");
    float cam[10];
    if (ptx_loaded_camera(loaded, cam)) die("camera");
    ptx_pinhole ph;
    if (ptx_pinhole_make(cam, cam + 3, cam + 6, cam[9], width, height, &ph)) die("camera");
    ptx_scene* scene = nullptr;
    if (ptx_scene_create(&d, PTX_FLAG_CHECK_BRUTE, &scene)) die("uploading the scene");
    printf("Testing up to %llu rays...\n", (unsigned long long)width * height * bounces);
    std::vector<float> accum((size_t)width * height * 3, 0.0f);
    if (ptx_render(scene, &ph, 0, 0, 1, accum.data(), nullptr, nullptr)) die("rendering");
    uint64_t checked = 0, fails = 0;
    if (ptx_check_report(scene, &checked, &fails)) die("report");
    std::vector<uint8_t> log(50 * (size_t)65536);
    uint32_t n = 0;
    if (ptx_check_fail_log(scene, log.data(), 65536, &n)) die("fail log");
    for (uint32_t i = 0; i < n && i < 16; ++i) {
        float f[12];
        memcpy(f, log.data() + 50 * (size_t)i + 2, 48);
        fprintf(stderr, "Fail %u\n  Ray { origin: (%g, %g, %g), direction x t: (%g, %g, %g) }\n  Expected: (%g, %g, %g)\n    Actual: (%g, %g, %g)\n", i,
                f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7], f[8], f[9], f[10], f[11]);
    }
    printf("Found %llu fails\n", (unsigned long long)fails);
    fprintf(stderr, "(%llu rays tested: extend and shadow rays of one sample per pixel)\n", (unsigned long long)checked);
    if (!output.empty()) {
        printf("Writing failed rays to %s...\n", output.c_str());
        FILE* f = fopen(output.c_str(), "wb");
        if (!f) {
            fprintf(stderr, "cannot write %s\n", output.c_str());
            return 1;
        }
        fwrite(log.data(), 50, n, f);
        fclose(f);
    }
    ptx_scene_destroy(scene);
    ptx_loaded_free(loaded);
    return 0;
}
