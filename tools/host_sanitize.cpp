#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../include/ptx.h"
int main(int argc, char** argv) {
    const char* all[] = {"cornell", "cornell_textured", "cube", "cornell_teapot", "sr2", "cubesplosion", "teapot"};
    const char* few[] = {"cornell", "cornell_textured", "cube", "teapot"};  // the ThreadSanitizer leg: instrumented builds of the big scenes take many minutes
    const bool quick = argc > 1 && !strcmp(argv[1], "--quick");
    const char** scenes = quick ? few : all;
    const int n_scenes = quick ? 4 : 7;
    for (int si = 0; si < n_scenes; ++si) {
        const char* s = scenes[si];
        std::string path = std::string("scenes/") + s + ".obj";
        ptx_loaded* L = nullptr;
        if (ptx_load_obj(path.c_str(), nullptr, 10, 0, &L)) { printf("%s: %s\n", s, ptx_last_error()); return 1; }
        ptx_scene_desc d; ptx_loaded_desc(L, &d);
        char* text = nullptr; size_t len = 0;
        if (ptx_kdtree_write_json(d.triangles, d.n_triangles, d.nodes, d.n_nodes, d.leaf_indices, d.n_leaf_indices, &text, &len)) return 2;
        ptx_kdtree* t = nullptr; float* tris = nullptr; uint32_t nt = 0;
        if (ptx_kdtree_read_json(text, len, &t, &tris, &nt)) { printf("read_json: %s\n", ptx_last_error()); return 3; }
        const ptx_kdnode* nodes; uint32_t nn; const uint32_t* li; uint32_t nl;
        ptx_kdtree_view(t, &nodes, &nn, &li, &nl);
        if (nn != d.n_nodes || memcmp(nodes, d.nodes, nn * 8) || nt != d.n_triangles) return 4;
        ptx_tree_stats st; ptx_kdtree_stats(d.triangles, d.n_triangles, d.nodes, d.n_nodes, d.leaf_indices, d.n_leaf_indices, nullptr, &st);
        char* pretty = nullptr; size_t pl = 0; ptx_kdtree_write_pretty(d.nodes, d.n_nodes, d.leaf_indices, d.n_leaf_indices, &pretty, &pl);
        // the device layout of the tree (host_devtree.cpp): shared record runs, breadth-first prefix
        {
            std::vector<uint32_t> words(4 * (size_t)d.n_nodes), recs(d.n_leaf_indices + 1);
            uint32_t counts[4];
            if (ptx_debug_device_tree(d.nodes, d.n_nodes, d.leaf_indices, d.n_leaf_indices, d.n_triangles, 1023, words.data(), recs.data(), counts)) return 7;
            if (counts[1] > d.n_leaf_indices) return 8;
        }
        // rays through the host traversal
        unsigned seed = 1; int hits = 0;
        for (int i = 0; i < 2000; ++i) {
            auto r = [&]() { seed = seed * 1664525u + 1013904223u; return (float)(seed >> 8) / 16777216.0f * 2 - 1; };
            ptx_ray ray{r(), r(), r(), 0.0f, r(), r(), r(), 3.4e38f}; ptx_hit h;
            if (ptx_kdtree_intersect_host(d.triangles, d.n_triangles, d.nodes, d.n_nodes, d.leaf_indices, &ray, &h)) return 5;
            hits += h.index != PTX_MISS;
        }
        printf("%s: %u tris %u nodes json %zu B, sah %.1f, %d/2000 hits\n", s, d.n_triangles, d.n_nodes, len, st.sah_cost, hits);
        ptx_free(text); ptx_free(pretty); ptx_free(tris); ptx_kdtree_free(t); ptx_loaded_free(L);
    }
    std::vector<uint8_t> img(64 * 48 * 3); for (size_t i = 0; i < img.size(); ++i) img[i] = (uint8_t)(i * 7);
    if (ptx_png_write_rgb8("/tmp/x.png", img.data(), 64, 48)) return 6;
    float* lin = nullptr; uint32_t w, h; if (ptx_png_read_linear("/tmp/x.png", &lin, &w, &h)) return 7; ptx_free(lin);
    std::vector<float> acc(64 * 48 * 3, 0.5f);
    ptx_accum_save("/tmp/a.acc", acc.data(), 64, 48, 3, 9); uint32_t ww = 64, hh = 48, n; uint64_t sd;
    if (ptx_accum_load("/tmp/a.acc", acc.data(), &ww, &hh, &sd, &n)) return 8;
    ptx_tonemap_rgb8(acc.data(), 64, 48, 4, img.data());
    uint32_t f, c; ptx_shard_samples(0, 100, 8, 3, &f, &c);
    puts("host paths clean");
    return 0;
}
