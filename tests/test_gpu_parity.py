"""GPU parity (-m gpu): the HIP path through the C ABI against the CPU oracle on the same inputs.

Bar: hit records (index, t, u, v) bit-exact; images bit-exact against the oracle's counter-sampler mode
with one thread (same paths, same per-pixel summation order)."""
import os

import numpy as np
import pytest

import pathtracer_amd as pt
import pyoracle as orc

from test_oracle_kat import L, N, ray_between, reverse

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "scenes")
f32 = np.float32


def synthetic_rays(lo, hi, n, seed):
    """SURVEY §8d ray set B: origins uniform in the box enlarged by 10 %, directions uniform on the sphere,
    plus segment rays between two points (t in [0,1]) and axis-parallel rays with +-0 components."""
    rng = np.random.default_rng(seed)
    c, h = (lo + hi) / 2, (hi - lo) / 2 * 1.1
    o = (c + rng.uniform(-1, 1, (n, 3)) * h).astype(f32)
    z = rng.uniform(-1, 1, n)
    phi = rng.uniform(0, 2 * np.pi, n)
    r = np.sqrt(1 - z * z)
    d = np.stack([r * np.cos(phi), r * np.sin(phi), z], 1).astype(f32)
    rays = orc.make_rays(o, d)
    m = n // 4
    p = (c + rng.uniform(-1, 1, (m, 3)) * h).astype(f32)
    seg = orc.make_rays(o[:m], p - o[:m], 0.0, 1.0)
    k = n // 8
    ax = orc.make_rays(o[:k], d[:k])
    axis = rng.integers(0, 3, k)
    zero = np.where(rng.random(k) < 0.5, f32(0.0), f32(-0.0))
    ax["d"][np.arange(k), axis] = zero
    two = rng.random(k) < 0.3
    ax["d"][np.arange(k)[two], (axis[two] + 1) % 3] = 0.0
    return np.concatenate([rays, seg, ax])


def assert_hits_equal(got, want):
    same = got.tobytes() == want.tobytes()
    if not same:
        bad = np.where((got["index"] != want["index"]) | (got["t"].view(np.uint32) != want["t"].view(np.uint32))
                       | (got["u"].view(np.uint32) != want["u"].view(np.uint32)) | (got["v"].view(np.uint32) != want["v"].view(np.uint32)))[0]
        raise AssertionError("%d of %d hits differ, first %d: got %s want %s" % (len(bad), len(got), bad[0], got[bad[0]], want[bad[0]]))


def test_reference_kdtree_vectors_on_gpu():  # kdtree/src/lib.rs:202-467
    cases = []
    tris = [[[0, 0, -1], [2, 0, -1], [2, 2, -1]], [[0, 0, 1], [2, 0, 1], [2, 2, 1]]]
    ray = ray_between([1, 1, -2], [1, 1, 2])
    cases.append((tris, N(0, 1.0, L(0, 1), L(0, 1)), [ray, reverse(ray)], 1.0, [(0, 0.25, 0.0, 0.5), (1, 0.25, 0.0, 0.5)]))
    tris = [[[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[1, 0, 0], [2, 0, 0], [2, 1, 0]]]
    cases.append((tris, N(0, 1.0, L(0), L(1)), [ray_between([0, 0, -1], [0, 0, 1]), ray_between([2, 0, -1], [2, 0, 1])], 1.0,
                  [(0, 0.5, 0.0, 0.0), (1, 0.5, 1.0, 0.0)]))
    z1 = [[[0, 0, 1], [1, 0, 1], [0, 1, 1]]]
    ray = ray_between([0, 0, 0], [0, 0, 2])
    for spec in (N(2, 1.0, L(0), L()), N(2, 1.0, L(), L(0)), N(2, 1.0, N(2, 1.0, L(0), L()), L()), N(2, 1.0, L(), N(2, 1.0, L(0), L()))):
        cases.append((z1, spec, [ray, reverse(ray)], 1.0, [(0, 0.5, 0.0, 0.0)] * 2))
    cases.append(([[[-1, -1, 1], [-1, -1, -1], [-1, 1, 1]]], N(0, -1.0, L(), L(0)),
                  [(np.array([-0.5170438, -0.4394186, -0.045965273], f32), np.array([-0.8491798, -0.1408107, -0.5089852], f32))],
                  pt.FMAX, [(0, 0.5687325, 0.66772085, 0.24024889)]))
    for tris, spec, rays, tmax, expect in cases:
        nodes, leaf = orc.flat_tree(spec)
        dev = pt.DeviceScene.from_arrays(tris, nodes, leaf)
        r = pt.make_rays([x[0] for x in rays], [x[1] for x in rays], 0.0, tmax)
        hits = dev.intersect(r)
        for h, e in zip(hits, expect):
            assert (int(h["index"]), f32(h["t"]), f32(h["u"]), f32(h["v"])) == (e[0], f32(e[1]), f32(e[2]), f32(e[3]))
        dev.close()
    nodes, leaf = orc.flat_tree(L())
    dev = pt.DeviceScene.from_arrays(np.zeros((0, 9), f32), nodes, leaf)  # empty tree (:203)
    assert dev.intersect(pt.make_rays([[0, 0, 0]], [[1, 1, 1]], 0.0, 1.0))[0]["index"] == pt.MISS
    assert len(dev.intersect(np.zeros(0, pt.RAY_DTYPE))) == 0  # empty batch
    dev.close()


def test_reference_embedded_sr2_ray_on_gpu():  # kdtree-tester-cli/src/reducer.rs:104-117
    """k_traverse (nearest and any-hit) on the one real-scene vector the reference holds: sr2, index 7589."""
    import torch

    from test_oracle_kat import SR2_RAY, SR2_REFERENCE

    dev = pt.LoadedScene(os.path.join(SCENES, "sr2.obj")).upload()
    rays = np.concatenate([pt.make_rays([SR2_RAY[0]], [SR2_RAY[1]], 0.0, tmax) for tmax in (1.0, pt.FMAX)] * 40)  # > one wave
    hits = dev.intersect(rays)
    want = (SR2_REFERENCE[0],) + tuple(map(f32, SR2_REFERENCE[1:]))
    for h in hits:
        assert (int(h["index"]), f32(h["t"]), f32(h["u"]), f32(h["v"])) == want
    d_rays = torch.from_numpy(rays.view(np.uint8).copy()).cuda()
    d_out = torch.zeros(len(rays), dtype=torch.uint8, device="cuda")
    dev.occluded_device(d_rays.data_ptr(), len(rays), d_out.data_ptr())
    torch.cuda.synchronize()
    assert d_out.cpu().numpy().all()
    dev.close()


@pytest.mark.parametrize("name,res", [("cornell", 96), ("cornell_teapot", 96), ("cornell_textured", 64), ("sr2", 64), ("cubesplosion", 64)])
def test_extend_bit_exact_vs_oracle(name, res):
    """Config 2: kdtree-tester-cli bounced ray set + synthetic rays; GPU == CPU pointer-tree traversal."""
    path = os.path.join(SCENES, name + ".obj")
    ref = orc.Scene(path)
    bounced, fails = ref.bounce_rays(res, res, 10, seed=3, threads=8)
    assert fails == 0  # kd-tree vs brute force inside the oracle (checked_intersection.rs tolerances)
    tris = ref.export()["triangles"].reshape(-1, 3, 3)
    rays = np.concatenate([bounced, synthetic_rays(tris.min((0, 1)), tris.max((0, 1)), 1 << 16, 0x5EED0001)])
    tree = ref.tree()
    want = tree.intersect(rays, threads=8)
    scene = pt.LoadedScene(path)
    dev = scene.upload()
    got = dev.intersect(rays)
    assert_hits_equal(got, want)
    assert (got["index"] != pt.MISS).mean() > 0.2
    dev.close()


def test_occlusion_query_equals_is_some():
    import torch

    path = os.path.join(SCENES, "cornell_teapot.obj")
    ref = orc.Scene(path)
    bounced, _ = ref.bounce_rays(64, 64, 6, seed=9, threads=8)
    want = ref.tree().intersect(bounced, threads=8)["index"] != orc.MISS
    dev = pt.LoadedScene(path).upload()
    d_rays = torch.from_numpy(bounced.view(np.uint8).copy()).cuda()
    d_out = torch.zeros(len(bounced), dtype=torch.uint8, device="cuda")
    dev.occluded_device(d_rays.data_ptr(), len(bounced), d_out.data_ptr())
    torch.cuda.synchronize()
    assert (d_out.cpu().numpy().astype(bool) == want).all()
    dev.close()


@pytest.mark.parametrize("name,w,h,spp,bounces", [("cornell", 48, 40, 3, 10), ("cornell_textured", 40, 40, 2, 10), ("cornell_teapot", 32, 32, 2, 6), ("sr2", 32, 18, 2, 4)])
def test_render_bit_exact_vs_oracle_counter_mode(name, w, h, spp, bounces):
    """Same counter-based sampler, one CPU thread: identical paths, identical per-pixel sum order."""
    path = os.path.join(SCENES, name + ".obj")
    ref = orc.Scene(path, max_bounces=bounces)
    want, _ = ref.render(w, h, threads=1, iterations_per_thread=spp, mode=1, seed=77, first_sample=5)
    scene = pt.LoadedScene(path, max_bounces=bounces)
    dev = scene.upload()
    got = dev.render(scene.pinhole(w, h), spp, seed=77, first_sample=5)
    diff = np.abs(got.astype(np.float64) - want.astype(np.float64))
    assert np.isfinite(got).all()
    assert got.tobytes() == want.tobytes(), "max abs diff %g at %s" % (diff.max(), np.unravel_index(diff.argmax(), diff.shape))
    # batch split must not matter (counter-based sampler): one sample per batch gives the same image
    dev.set_batch_samples(1)
    again = dev.render(scene.pinhole(w, h), spp, seed=77, first_sample=5)
    assert again.tobytes() == got.tobytes()
    # nor must the number of batches in flight: two pipelines (streams) add the samples in the same order
    dev.set_pipelines(2)
    again = dev.render(scene.pinhole(w, h), spp, seed=77, first_sample=5)
    assert again.tobytes() == got.tobytes()
    dev.set_pipelines(1)
    dev.set_batch_samples(0)
    again = dev.render(scene.pinhole(w, h), spp, seed=77, first_sample=5)
    assert again.tobytes() == got.tobytes()
    # nor must path regeneration: with a quarter of the queue slots the batch needs, finished paths are
    # replaced by the next camera rays round after round (64 slots per segment, the minimum)
    dev.set_queue_slots(64 * ((w * h * spp + 255) // 256))
    again = dev.render(scene.pinhole(w, h), spp, seed=77, first_sample=5)
    assert again.tobytes() == got.tobytes()
    ext, _ = dev.last_batch_counts()
    assert len(ext) > bounces  # it really took more rounds than bounces
    dev.close()


def test_render_accumulates_into_buffer_and_counts_rays():
    path = os.path.join(SCENES, "cornell.obj")
    scene = pt.LoadedScene(path)
    # FLAG_NO_BOX_CLIP: the reference's own traversal range, so that the node and triangle counts are the oracle's
    dev = scene.upload(flags=pt.FLAG_COUNT_WORK | pt.FLAG_TIME_KERNELS | pt.FLAG_NO_BOX_CLIP)
    ph = scene.pinhole(32, 32)
    a = dev.render(ph, 2, seed=1)
    b = dev.render(ph, 2, seed=1, first_sample=2, accum=a.copy())
    c = dev.render(ph, 4, seed=1)
    assert b.tobytes() == c.tobytes()  # buffer is accumulated into, never cleared (image_buffer semantics)
    ref = orc.Scene(path)
    _, _, st = ref.render(32, 32, threads=1, iterations_per_thread=2, mode=1, seed=1, stats=True)
    dev.stats_reset()
    dev.render(ph, 2, seed=1)
    s = dev.stats()
    assert s["extend_rays"] == st["extend_rays"] and s["shadow_rays"] == st["shadow_rays"]
    assert s["extend_inner_nodes"] == st["extend_inner_nodes"] and s["extend_tri_tests"] == st["extend_tri_tests"]
    assert s["samples"] == 2 * 32 * 32 and s["extend_ms"] > 0
    dev.close()
    # the default (ranges narrowed to the bounding box of the triangles): same image, same rays, fewer node visits
    clip = scene.upload(flags=pt.FLAG_COUNT_WORK)
    assert clip.render(ph, 2, seed=1).tobytes() == a.tobytes()
    sc = clip.stats()
    assert sc["extend_rays"] == s["extend_rays"] and sc["shadow_rays"] == s["shadow_rays"]
    assert sc["extend_inner_nodes"] < s["extend_inner_nodes"] and sc["extend_tri_tests"] <= s["extend_tri_tests"]
    clip.close()


@pytest.mark.parametrize("name", ["sr2", "cubesplosion"])
def test_camera_rays_that_miss_the_box_end_at_generation_and_change_nothing(name):
    """Open scenes: most camera rays miss the bounding box of the triangles and are finished by k_start / k_regenerate
    (radiance = environment, no queue entry).  Same image and same ray counts as the oracle and as the full-range walk
    (FLAG_NO_BOX_CLIP, which also switches the shortcut off), also with path regeneration over a small queue and in a
    region; the ray log and the brute-force checker still see every ray."""
    path = os.path.join(SCENES, name + ".obj")
    w, h, spp = 160, 90, 3
    ref = orc.Scene(path, max_bounces=6)
    want, _, st = ref.render(w, h, threads=spp, iterations_per_thread=1, mode=1, seed=41, stats=True)
    scene = pt.LoadedScene(path, max_bounces=6)
    ph = scene.pinhole(w, h)
    dev = scene.upload()
    assert dev.render(ph, spp, seed=41).tobytes() == want.tobytes()
    s = dev.stats()
    assert s["extend_rays"] == st["extend_rays"] and s["shadow_rays"] == st["shadow_rays"]
    assert 0.3 * w * h * spp < s["culled_camera_rays"] < w * h * spp  # most of the frame is sky
    dev.set_queue_slots(64 * 40)  # far fewer slots than paths: k_regenerate starts (and culls) paths round after round
    dev.stats_reset()
    assert dev.render(ph, spp, seed=41).tobytes() == want.tobytes()
    s2 = dev.stats()
    assert s2["extend_rays"] == st["extend_rays"] and s2["culled_camera_rays"] == s["culled_camera_rays"]
    dev.close()
    full = scene.upload(flags=pt.FLAG_NO_BOX_CLIP)
    assert full.render(ph, spp, seed=41).tobytes() == want.tobytes()
    assert full.stats()["culled_camera_rays"] == 0 and full.stats()["extend_rays"] == st["extend_rays"]
    full.close()
    chk = scene.upload(flags=pt.FLAG_CHECK_BRUTE)  # the checker wants every ray in the queue
    assert chk.render(ph, 1, seed=41).tobytes() == dev_render_one(scene, ph).tobytes()
    checked, fails = chk.check_report()
    assert chk.stats()["culled_camera_rays"] == 0 and checked == chk.stats()["extend_rays"] + chk.stats()["shadow_rays"]
    chk.close()


def dev_render_one(scene, ph):
    dev = scene.upload()
    img = dev.render(ph, 1, seed=41)
    dev.close()
    return img


def test_sample_sharding_is_exact_on_gpu():
    """Two disjoint sample ranges rendered into separate device buffers sum to the same partial sums the
    oracle produces — what the RCCL reduce of parallel.render_sharded adds up."""
    import torch

    from pathtracer_amd.parallel import shard_samples

    path = os.path.join(SCENES, "cornell.obj")
    scene = pt.LoadedScene(path)
    dev = scene.upload()
    ph = scene.pinhole(40, 30)
    ref = orc.Scene(path)
    total = np.zeros((30, 40, 3), f32)
    for rank in range(2):
        first, count = shard_samples(5, 2, rank)
        buf = torch.zeros(30 * 40 * 3, dtype=torch.float32, device="cuda")
        dev.render_device(ph, count, buf.data_ptr(), seed=9, first_sample=first, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        want, _ = ref.render(40, 30, threads=1, iterations_per_thread=count, mode=1, seed=9, first_sample=first)
        got = buf.cpu().numpy().reshape(30, 40, 3)
        assert got.tobytes() == want.tobytes()
        total += got
    whole = dev.render(ph, 5, seed=9)
    assert np.allclose(total, whole, rtol=1e-5, atol=1e-6)
    dev.close()


def test_pathtracer_cli_renders_a_png(tmp_path):
    import subprocess

    exe = os.path.join(ROOT, "pathtracer_amd", "bin", "pathtracer-cli")
    out = str(tmp_path / "cornell.png")
    r = subprocess.run([exe, "-i", os.path.join(SCENES, "cornell.obj"), "-o", out, "-s", "64x48", "-t", "2", "-n", "2", "--seed", "5"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    for line in ("Loading", "Chunks: 7", "Materials: 7", "Building kdtree...", "Rendering 64x48 px image with 2 thread(s) and 4 total iteration(s)...", "Total time:", "Writing"):
        assert line in r.stdout, r.stdout
    from PIL import Image

    img = np.asarray(Image.open(out))
    assert img.shape == (48, 64, 3)
    # same image as the library path + tonemap (and therefore as the oracle)
    scene = pt.LoadedScene(os.path.join(SCENES, "cornell.obj"))
    dev = scene.upload()
    want = pt.to_rgb8(dev.render(scene.pinhole(64, 48), 4, seed=5), 4)
    assert (img == want).all()
    dev.close()


@pytest.mark.parametrize("name", ["cornell_teapot", "sr2", "cubesplosion"])
def test_box_clip_changes_no_hit(name):
    """k_traverse narrows every ray's range to its passage through the (widened) bounding box of the triangles; with
    FLAG_NO_BOX_CLIP it walks the reference's whole range.  Both must return the same hit records, bit for bit, for rays
    of every kind: from inside and from far outside, towards and past the scene, along and on the faces of the box,
    axis-parallel, with t windows that start or end inside the scene, empty and NaN windows."""
    import torch

    scene = pt.LoadedScene(os.path.join(SCENES, name + ".obj"))
    tris = scene.arrays()["triangles"].reshape(-1, 3)
    lo, hi = tris.min(0), tris.max(0)
    ext = float((hi - lo).max())
    rng = np.random.default_rng(12)
    n = 1 << 19
    parts_o, parts_d = [], []
    inside = lo + rng.random((n, 3), dtype=f32) * (hi - lo)
    sphere = rng.normal(size=(n, 3)).astype(f32)
    sphere /= np.linalg.norm(sphere, axis=1, keepdims=True)
    parts_o.append(inside), parts_d.append(sphere)  # from inside, all directions
    far = ((lo + hi) / 2 + sphere * ext * rng.uniform(1.0, 50.0, (n, 1))).astype(f32)
    parts_o.append(far), parts_d.append((inside[::-1] - far).astype(f32))  # from outside towards the scene (not normalised)
    parts_o.append(far), parts_d.append(rng.normal(size=(n, 3)).astype(f32))  # from outside, mostly past it
    on_face = inside.copy()  # on a face of the box, grazing along it and leaving / entering at small angles
    axis = rng.integers(0, 3, n)
    side = rng.integers(0, 2, n)
    on_face[np.arange(n), axis] = np.where(side == 0, lo[axis], hi[axis])
    graze = sphere.copy()
    graze[np.arange(n), axis] *= rng.choice(np.array([0.0, 1e-7, -1e-7, 1e-4, -1e-4, 1.0], f32), n)
    parts_o.append(on_face), parts_d.append(graze)
    axial = np.zeros((n, 3), f32)  # axis-parallel, +-0 in the other components
    axial[np.arange(n), axis] = np.where(side == 0, 1.0, -1.0)
    axial[np.arange(n), (axis + 1) % 3] = np.where(rng.integers(0, 2, n) == 0, 0.0, -0.0)
    parts_o.append(np.where(rng.integers(0, 2, (n, 1)) == 0, inside, far).astype(f32)), parts_d.append(axial)
    o = np.concatenate(parts_o)
    d = np.concatenate(parts_d)
    m = len(o)
    tmin = np.zeros(m, f32)
    tmax = np.full(m, np.finfo(f32).max, f32)
    k = rng.integers(0, 8, m)  # windows: [0, 1], [a, b] inside the scene, empty, NaN
    tmax[k == 1] = 1.0
    tmin[k == 2] = rng.random(m, dtype=f32)[k == 2] * 0.5
    tmax[k == 2] = tmin[k == 2] + rng.random(m, dtype=f32)[k == 2] * ext
    tmin[k == 3], tmax[k == 3] = 2.0, 1.0
    tmax[k == 4] = np.nan
    rays = np.zeros(m, pt.RAY_DTYPE)
    rays["o"], rays["d"], rays["tmin"], rays["tmax"] = o, d, tmin, tmax
    d_rays = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
    out = []
    for flags in (0, pt.FLAG_NO_BOX_CLIP):
        dev = scene.kdtree().upload(flags=flags)
        hits = torch.zeros(m * 16, dtype=torch.uint8, device="cuda")
        dev.intersect_device(d_rays.data_ptr(), m, hits.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        occ = torch.zeros(m, dtype=torch.uint8, device="cuda")
        dev.occluded_device(d_rays.data_ptr(), m, occ.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        out.append((hits.cpu().numpy().tobytes(), occ.cpu().numpy().tobytes()))
        dev.close()
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
    h = np.frombuffer(out[0][0], pt.HIT_DTYPE)
    assert 0.05 < (h["index"] != 0xFFFFFFFF).mean() < 0.95  # the set exercises hits and misses


def _odd_rays(lo, hi, n, seed):
    """Synthetic rays of which every third has a component pushed outside the window in which k_traverse divides the
    short way (ptx_device.hip: split_quotient / short_div_ok): origins 1e-30, denormal, 1e20, 3e38, infinite, NaN;
    direction components 1e-15, denormal, 1e6 (unnormalised), infinite, NaN; and values exactly on the window's edges."""
    rng = np.random.default_rng(seed)
    rays = synthetic_rays(lo, hi, n, seed)
    m = len(rays)
    odd_o = np.array([1e-30, -1e-30, 1e-41, -3e-45, 1e20, -3e38, np.inf, -np.inf, np.nan, 2.0 ** -77, 2.0 ** -78, 2.0 ** 60, 2.0 ** 61], f32)
    odd_d = np.array([1e-15, -1e-15, 1e-41, -1.4e-45, 1e6, -1e6, np.inf, np.nan, 2.0 ** -40, 2.0 ** -41, 2.0 ** 10, 2.0 ** 11, -0.0], f32)
    pick = np.arange(0, m, 3)
    which = rng.integers(0, 3, len(pick))
    comp = rng.integers(0, 3, len(pick))
    for k, (i, w, c) in enumerate(zip(pick, which, comp)):
        if w == 0:
            rays["o"][i, c] = odd_o[k % len(odd_o)]
        elif w == 1:
            rays["d"][i, c] = odd_d[k % len(odd_d)]
        else:
            rays["o"][i, c] = odd_o[(k // 3) % len(odd_o)]
            rays["d"][i, (c + 1) % 3] = odd_d[(k // 5) % len(odd_d)]
    return rays


def _same_hits_nan_aware(got, want):
    """Bit-exact, except that two NaNs in the same field count as equal whatever their payload."""
    assert (got["index"] == want["index"]).all()
    for f in ("t", "u", "v"):
        g, w = got[f], want[f]
        same = (g.view(np.uint32) == w.view(np.uint32)) | (np.isnan(g) & np.isnan(w))
        assert same.all(), (f, int((~same).sum()), got[~same][:3], want[~same][:3])


@pytest.mark.parametrize("name", ["cornell_teapot", "sr2"])
def test_rays_outside_the_short_division_window_take_the_long_one(name, monkeypatch):
    """The node step divides by multiplying with a refined reciprocal where that is proven to give the IEEE quotient
    (tools/div_probe.hip) and with the compiler's division everywhere else.  Rays on either side of the guard, mixed
    inside the same waves: hits equal the oracle's pointer-tree traversal (which always divides); and a library that
    is told to divide the long way throughout (PTX_LONG_DIV=1) returns the same bytes as the default one."""
    path = os.path.join(SCENES, name + ".obj")
    ref = orc.Scene(path)
    tris = ref.export()["triangles"].reshape(-1, 3, 3)
    rays = _odd_rays(tris.min((0, 1)), tris.max((0, 1)), 1 << 17, 0x5EED0007)
    want = ref.tree().intersect(rays, threads=8)
    scene = pt.LoadedScene(path)
    dev = scene.upload()
    got = dev.intersect(rays)
    _same_hits_nan_aware(got, want)
    assert (got["index"] != pt.MISS).mean() > 0.1
    plain = synthetic_rays(tris.min((0, 1)), tris.max((0, 1)), 1 << 18, 0x5EED0008)
    short = dev.intersect(plain)
    dev.close()
    monkeypatch.setenv("PTX_LONG_DIV", "1")
    dev = scene.upload()
    assert dev.intersect(plain).tobytes() == short.tobytes()
    assert dev.intersect(rays).tobytes() == got.tobytes()
    dev.close()


def test_scene_with_split_planes_outside_the_short_division_window():
    """A tree with split planes outside the window (1e-30, a denormal, 2^61): the library sees it when the scene is
    created and walks every ray with the long division.  Every leaf of the hand-made tree holds every triangle, so any
    plane is a valid split.  (Whole scenes outside the window - coordinates around 1e-24 or 1e19 - are beyond what the
    triangle tests can do in f32: their determinants underflow or overflow.)"""
    rng = np.random.default_rng(11)
    tris = rng.uniform(-1, 1, (12, 3, 3)).astype(f32)
    tris[:4, :, 2] = tris[:4, :1, 2]  # four axis-aligned ones
    every = L(*range(12))
    spec = N(0, 1e-30, N(1, -1e-41, every, N(2, 0.25, every, every)), N(2, 2.0 ** 61, N(0, 0.5, every, every), every))
    nodes, leaf = orc.flat_tree(spec)
    lo, hi = np.array([-1, -1, -1], f32), np.array([1, 1, 1], f32)
    rays = synthetic_rays(lo, hi, 1 << 16, 0x5EED0009)
    want = orc.Tree.from_flat(tris.reshape(-1, 9), nodes, leaf).intersect(rays, threads=8)  # (axis-aligned ones as such)
    dev = pt.DeviceScene.from_arrays(tris, nodes, leaf)
    got = dev.intersect(rays)
    dev.close()
    assert_hits_equal(got, want)
    assert (got["index"] != pt.MISS).mean() > 0.2


def test_every_ray_is_answered_exactly_once_at_any_launch_size():
    """The windows of a traversal launch are handed out by number, from four striped counters, one or two at a time
    (k_traverse's refill): whatever the ray count, every ray gets its answer, no slot beyond the last ray is written,
    and the occlusion query agrees - from one ray to a few million, on sizes that leave partial windows, partial stripes
    and partial pairs."""
    import torch

    path = os.path.join(SCENES, "cornell_teapot.obj")
    ref = orc.Scene(path)
    tris = ref.export()["triangles"].reshape(-1, 3, 3)
    rays = synthetic_rays(tris.min((0, 1)), tris.max((0, 1)), 1 << 22, 0x5EED000A)[: (1 << 22) + 13]
    want = ref.tree().intersect(rays, threads=8)
    dev = pt.LoadedScene(path).upload()
    d_rays = torch.from_numpy(rays.view(np.uint8).copy()).cuda()
    sentinel = np.uint8(0xA5)
    for n in (1, 63, 64, 65, 1000, 65537, 300001, (1 << 20) + 1, (1 << 21) + 7, (1 << 22) + 13):
        d_hits = torch.full(((len(rays) + 1) * 16,), int(sentinel), dtype=torch.uint8, device="cuda")
        d_occ = torch.full((len(rays) + 1,), int(sentinel), dtype=torch.uint8, device="cuda")
        dev.intersect_device(d_rays.data_ptr(), n, d_hits.data_ptr())
        dev.occluded_device(d_rays.data_ptr(), n, d_occ.data_ptr())
        torch.cuda.synchronize()
        got = d_hits.cpu().numpy()
        assert got[: n * 16].tobytes() == want[:n].tobytes(), n
        assert (got[n * 16:] == sentinel).all(), n
        occ = d_occ.cpu().numpy()
        assert (occ[:n].astype(bool) == (want[:n]["index"] != pt.MISS)).all(), n
        assert (occ[n:] == sentinel).all(), n
    dev.close()
