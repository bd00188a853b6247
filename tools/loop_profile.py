#!/usr/bin/env python3
"""Lane occupancy of the traversal loop on the bench workload (counting variant of the kernels)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pathtracer_amd as pt

NODE_STEPS = 6  # PTX_ROUNDS x PTX_NODE_UNROLL of the library build (3 x 2); the triangle step tests PTX_TRI_UNROLL = 2 triangles, the exit step runs PTX_ROUNDS times
scene_name = sys.argv[1] if len(sys.argv) > 1 else "cornell_teapot"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
W = H = 1080
scene = pt.LoadedScene(os.path.join(os.path.dirname(__file__), "..", "scenes", scene_name + ".obj"), max_bounces=10)
dev = scene.upload(flags=pt.FLAG_COUNT_WORK)
acc = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
dev.render_device(scene.pinhole(W, H), spp, acc.data_ptr(), seed=0, first_sample=0, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
st = dev.stats()
pf = dev.loop_profile()
for k, rays, inner, tris in (("extend", st["extend_rays"], st["extend_inner_nodes"], st["extend_tri_tests"]),
                            ("shadow", st["shadow_rays"], st["shadow_inner_nodes"], st["shadow_tri_tests"])):
    p = pf[k]
    it = max(p["iters"], 1)
    print("%s: rays %d  inner/ray %.2f  tris/ray %.2f  leaf-exits/ray %.2f" % (k, rays, inner / rays, tris / rays, p["end_lanes"] / rays))
    print("   wave iterations/ray*64 = %.1f  (node-step visits/ray %.2f)" % (it * 64 / rays, p["walk_lanes"] / rays))
    print("   node steps: %.1f lane-visits per iteration (%d node steps: %.1f/64 lanes each)   live lanes at iteration start %.1f/64" % (p["walk_lanes"] / it, NODE_STEPS, p["walk_lanes"] / it / NODE_STEPS, p["live_lanes"] / it))
    print("   tri step : ran in %.1f%% of iterations, lanes active %.1f/64" % (100 * p["tri_iters"] / it, p["tri_lanes"] / max(p["tri_iters"], 1)))
    print("   leaf exit: ran in %.1f%% of iterations, lanes active %.1f/64" % (100 * p["end_iters"] / it, p["end_lanes"] / max(p["end_iters"], 1)))
    print("   refill rounds per 100 iterations %.1f" % (100 * p["refills"] / it))
    clk = sum(p[k] for k in ("clk_refill", "clk_node", "clk_tri", "clk_exit")) or 1
    print("   wave clock: refill %.1f%%  node step %.1f%%  triangle step %.1f%%  leaf exit %.1f%%   (%.0f ticks per iteration)" % (
        100 * p["clk_refill"] / clk, 100 * p["clk_node"] / clk, 100 * p["clk_tri"] / clk, 100 * p["clk_exit"] / clk, clk / it))
