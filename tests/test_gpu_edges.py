"""GPU edge cases (-m gpu): light counts that take the generic shade kernel (0, 3, 8 lights), bounce limits,
a scene without geometry, ragged and empty ray batches, rays that can never hit, non-finite rays (must
terminate), and the argument errors of the C ABI."""
import os
import shutil

import numpy as np
import pytest

import pathtracer_amd as pt
import pyoracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "scenes")
f32 = np.float32


def cornell_with_lights(tmp_path, n_lights):
    """cornell.obj with its MTL rewritten to carry `n_lights` lights (custom MTL syntax, wavefront/src/mtl.rs)."""
    src_obj, src_mtl = os.path.join(SCENES, "cornell.obj"), os.path.join(SCENES, "cornell.mtl")
    obj = tmp_path / "cornell.obj"
    shutil.copy(src_obj, obj)
    lines, skipping = [], False
    for line in open(src_mtl):
        if line.startswith("newlight"):
            skipping = True
            continue
        if skipping and line.strip().startswith("light"):
            continue
        skipping = False
        lines.append(line)
    rng = np.random.default_rng(n_lights)
    lights = []
    for i in range(n_lights):
        p, c = rng.uniform(-0.7, 0.7, 3), rng.uniform(0.3, 1.0, 3)
        lights.append("newlight L%d\n\tlightposition %.3f %.3f %.3f\n\tlightcolor %.3f %.3f %.3f\n\tlightradius %.2f\n\tlightintensity %.2f\n\n"
                      % (i, p[0], abs(p[1]), p[2], c[0], c[1], c[2], 0.05 + 0.05 * i, 1.0 + 0.5 * i))
    (tmp_path / "cornell.mtl").write_text("".join(lights) + "".join(lines))
    return str(obj)


@pytest.mark.parametrize("n_lights", [0, 3, 8])
def test_light_counts_of_the_generic_shade_kernel(tmp_path, n_lights):
    path = cornell_with_lights(tmp_path, n_lights)
    ref = orc.Scene(path, max_bounces=5)
    assert ref.counts["lights"] == n_lights
    want, _, st = ref.render(40, 32, threads=1, iterations_per_thread=2, mode=1, seed=4, stats=True)
    scene = pt.LoadedScene(path, max_bounces=5)
    dev = scene.upload()
    got = dev.render(scene.pinhole(40, 32), 2, seed=4)
    assert got.tobytes() == want.tobytes()
    s = dev.stats()
    assert s["extend_rays"] == st["extend_rays"] and s["shadow_rays"] == st["shadow_rays"]
    assert (s["shadow_rays"] == 0) == (n_lights == 0)
    dev.close()


@pytest.mark.parametrize("bounces", [0, 1, 2, 40])
def test_bounce_limits(bounces):
    path = os.path.join(SCENES, "cornell_textured.obj")
    ref = orc.Scene(path, max_bounces=bounces)
    want, _ = ref.render(32, 32, threads=1, iterations_per_thread=2, mode=1, seed=8)
    scene = pt.LoadedScene(path, max_bounces=bounces)
    dev = scene.upload()
    got = dev.render(scene.pinhole(32, 32), 2, seed=8)
    assert got.tobytes() == want.tobytes()
    if bounces == 0:
        assert not got.any()  # the reference's loop body never runs: black
    dev.close()


def one_leaf(count):
    n = np.zeros(1, pt.NODE_DTYPE)
    n["w0"], n["w1"] = 0, (count << 2) | 3
    return n


def test_scene_without_geometry_and_ragged_batches():
    import torch

    empty = pt.DeviceScene.from_arrays(np.zeros((0, 9), f32), one_leaf(0), np.zeros(0, np.uint32))
    for n in (0, 1, 63, 64, 65, 1000):
        rays = pt.make_rays(np.zeros((n, 3), f32), np.tile(f32([0, 0, -1]), (n, 1)))
        hits = empty.intersect(rays)
        assert len(hits) == n and (hits["index"] == pt.MISS).all() and not hits["t"].any()
    empty.close()
    # one triangle, the tree a single one-triangle leaf: hit / miss / out of range / parallel / behind
    tri = f32([[-1, -1, -2, 1, -1, -2, 0, 1, -2]])
    dev = pt.DeviceScene.from_arrays(tri, one_leaf(1), np.array([0], np.uint32))
    o = f32([[0, 0, 0], [5, 5, 0], [0, 0, 0], [0, 0, 0], [0, 0, -3]])
    d = f32([[0, 0, -1], [0, 0, -1], [0, 0, -1], [1, 0, 0], [0, 0, -1]])
    rays = pt.make_rays(o, d)
    rays["tmax"][2] = 1.5  # the triangle is at t = 2
    hits = dev.intersect(rays)
    assert hits["index"].tolist() == [0, pt.MISS, pt.MISS, pt.MISS, pt.MISS] and hits["t"][0] == 2.0
    # device-buffer entry point with n == 0 is a no-op
    buf = torch.zeros(64, dtype=torch.uint8, device="cuda")
    dev.intersect_device(buf.data_ptr(), 0, buf.data_ptr())
    dev.occluded_device(buf.data_ptr(), 0, buf.data_ptr())
    dev.close()


def test_non_finite_rays_terminate_and_match_the_host_traversal():
    path = os.path.join(SCENES, "cornell_teapot.obj")
    scene = pt.LoadedScene(path)
    kd = scene.kdtree()
    rng = np.random.default_rng(3)
    n = 4096
    o = rng.uniform(-1, 1, (n, 3)).astype(f32)
    d = rng.normal(size=(n, 3)).astype(f32)
    bad = rng.integers(0, 6, n)
    d[bad == 0, 0] = np.nan
    d[bad == 1] = 0.0
    d[bad == 2, 1] = np.inf
    o[bad == 3, 2] = np.nan
    o[bad == 4, 0] = -np.inf
    rays = pt.make_rays(o, d)
    rays["tmax"][bad == 5] = np.inf
    dev = scene.upload()
    got = dev.intersect(rays)  # must come back (every lane's loop is bounded by the tree depth)
    ref = orc.Scene(path).tree().intersect(rays, threads=8)
    same = got.tobytes() == ref.tobytes()
    if not same:  # NaN payloads may differ in sign/bits; indices and finite fields must not
        assert (got["index"] == ref["index"]).all()
        for k in ("t", "u", "v"):
            a, b = got[k], ref[k]
            assert ((a == b) | (np.isnan(a) & np.isnan(b))).all()
    for i in np.flatnonzero(bad < 5)[:32]:
        h = kd.intersect_host(o[i], d[i], (0.0, float(rays["tmax"][i])))
        assert h["index"] == got["index"][i]
    dev.close()


def test_argument_errors():
    scene = pt.LoadedScene(os.path.join(SCENES, "cornell.obj"))
    dev = scene.upload()
    with pytest.raises(pt.PtxError):
        dev.set_pipelines(0)
    with pytest.raises(pt.PtxError):
        dev.set_pipelines(9)
    with pytest.raises(pt.PtxError):
        dev.set_queue_slots(1 << 32)
    ph = scene.pinhole(16, 16)
    assert not dev.render(ph, 0).any()  # zero samples: nothing added
    with pytest.raises(pt.PtxError):
        dev.render_device(ph, 1, 0)  # null accumulation buffer
    bare = scene.kdtree().upload()  # no shading data
    with pytest.raises(pt.PtxError):
        bare.render(ph, 1)
    bare.close()
    dev.close()


def test_tiles_add_up_to_the_full_frame():
    """`render_subdivided_mut` (pathtracer.rs:143-161): ragged tiles in shuffled order == one full-frame call."""
    import torch

    scene = pt.LoadedScene(os.path.join(SCENES, "cornell_textured.obj"), max_bounces=6)
    dev = scene.upload()
    w, h, spp = 70, 45, 3
    ph = scene.pinhole(w, h)
    stream = torch.cuda.current_stream().cuda_stream
    full = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
    dev.render_device(ph, spp, full.data_ptr(), seed=21, first_sample=4, stream=stream)
    tiled = torch.zeros_like(full)
    tiles = [(x, y, min(32, w - x), min(16, h - y)) for y in range(0, h, 16) for x in range(0, w, 32)]
    for k in np.random.default_rng(0).permutation(len(tiles)):
        x, y, tw, th = tiles[k]
        dev.render_region_device(ph, x, y, tw, th, spp, tiled.data_ptr(), seed=21, first_sample=4, stream=stream)
    torch.cuda.synchronize()
    assert torch.equal(full, tiled)
    # a one-pixel region touches one pixel
    one = torch.zeros_like(full)
    dev.render_region_device(ph, 69, 44, 1, 1, spp, one.data_ptr(), seed=21, first_sample=4, stream=stream)
    torch.cuda.synchronize()
    img = one.cpu().numpy().reshape(h, w, 3)
    assert img[44, 69].tobytes() == full.cpu().numpy().reshape(h, w, 3)[44, 69].tobytes() and not img[:44].any() and not img[44, :69].any()
    for bad in ((0, 0, 0, 5), (60, 0, 11, 5), (0, 45, 1, 1)):
        with pytest.raises(pt.PtxError):
            dev.render_region_device(ph, *bad, 1, one.data_ptr(), stream=stream)
    dev.close()


def test_checkpoint_resume_equals_uninterrupted_run(tmp_path):
    """[0, 3) saved, loaded, continued with [3, 7) == [0, 7) in one go, bit for bit; same through the CLI."""
    import subprocess

    scene = pt.LoadedScene(os.path.join(SCENES, "cornell.obj"), max_bounces=6)
    dev = scene.upload()
    ph = scene.pinhole(48, 40)
    whole = dev.render(ph, 7, seed=21)
    part = dev.render(ph, 3, seed=21)
    ck = str(tmp_path / "run.ptxacc")
    pt.save_accum(ck, part, seed=21, samples=3)
    acc, seed, done = pt.load_accum(ck)
    assert (seed, done) == (21, 3)
    dev.render(ph, 7 - done, seed=seed, first_sample=done, accum=acc)
    assert acc.tobytes() == whole.tobytes()
    dev.close()

    exe = os.path.join(ROOT, "pathtracer_amd", "bin", "pathtracer-cli")
    common = [exe, "-i", os.path.join(SCENES, "cornell.obj"), "-s", "48x40", "-m", "6", "-t", "1"]
    a, b, c = (str(tmp_path / n) for n in ("a.png", "b.png", "c.png"))
    ck2 = str(tmp_path / "cli.ptxacc")
    for args in (["-o", a, "-n", "7", "--seed", "21"], ["-o", b, "-n", "3", "--seed", "21", "--checkpoint", ck2],
                 ["-o", c, "-n", "4", "--resume", ck2, "--checkpoint", ck2]):
        r = subprocess.run(common + args, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
    assert "Resuming" in r.stdout and "3 iteration(s) done" in r.stdout
    assert open(a, "rb").read() == open(c, "rb").read()
    acc2, seed2, done2 = pt.load_accum(ck2)
    assert (seed2, done2) == (21, 7) and acc2.tobytes() == whole.tobytes()


def test_render_shrinks_its_queues_when_device_memory_is_short():
    """The queue budget comes from the memory that is free at scene creation; when an allocation fails later
    (somebody else took the memory) the batch is retried half as wide until it fits - same image, bit for bit."""
    import torch

    scene = pt.LoadedScene(os.path.join(SCENES, "cornell.obj"), max_bounces=5)
    dev = scene.upload()
    ph = scene.pinhole(1024, 1024)
    want = dev.render(ph, 48, seed=3)  # 50 M paths in one batch: ~9 GB of queues
    dev.close()
    dev = scene.upload()
    free_b, _ = torch.cuda.mem_get_info()
    hog = torch.empty(free_b - (4 << 30), dtype=torch.uint8, device="cuda")  # leaves ~4 GB
    try:
        got = dev.render(ph, 48, seed=3)
        ext, _ = dev.last_batch_counts()
    finally:
        del hog
        torch.cuda.empty_cache()
    assert got.tobytes() == want.tobytes()
    assert len(ext) > 5  # narrower queues: path regeneration ran more rounds than bounces
    dev.close()


def test_device_tonemap_equals_the_host_tonemap():
    """`to_rgb8` as a kernel.  Bytes equal the host entry point's (== the oracle's) on a rendered frame and
    on the special values; a byte may differ by one code only where libm's powf is not correctly rounded."""
    import torch

    scene = pt.LoadedScene(os.path.join(SCENES, "cornell_textured.obj"), max_bounces=6)
    dev = scene.upload()
    w, h, spp = 250, 131, 24  # 3 * w * h is not a multiple of 4: the tail lanes store bytes
    acc = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
    dev.render_device(scene.pinhole(w, h), spp, acc.data_ptr(), seed=4)
    acc[:12] = torch.tensor([0.0, -1.0, float("nan"), float("inf"), -float("inf"), 1e-30, spp * 1.0, spp * 0.999, spp * 1.001, 1e30, -0.0, spp * 0.5])
    out = torch.full((h * w * 3,), 7, dtype=torch.uint8, device="cuda")
    for iters in (spp, 65536 + spp):  # `as u16`: 65560 iterations tone-map like 24
        pt.to_rgb8_device(acc.data_ptr(), w, h, iters, out.data_ptr())
        torch.cuda.synchronize()
        want = pt.to_rgb8(acc.cpu().numpy().reshape(h, w, 3), iters).reshape(-1).astype(np.int32)
        got = out.cpu().numpy().astype(np.int32)
        diff = np.abs(got - want)
        assert diff.max() <= 1 and (diff != 0).mean() < 1e-5
        assert list(got[:12]) == list(want[:12]) and got[0] == 0 and got[3] == 255  # zero, negative, NaN (`min` keeps the 1.0), inf, ...
    dev.close()


@pytest.mark.parametrize("name,w,h,spp", [("cornell_textured", 160, 120, 6), ("sr2", 128, 72, 4), ("cornell_teapot", 96, 96, 3)])
def test_material_keyed_shade_queues_change_no_bit(name, w, h, spp):
    """SURVEY 8f-1, second half: with PTX_FLAG_MATERIAL_QUEUES the entries of every queue segment are shaded grouped by
    the material they hit (k_bucket + a permuted k_shade).  Only the order of shading changes: same image, bit for bit,
    with plain batches, with path regeneration, and in a tile."""
    import torch

    scene = pt.LoadedScene(os.path.join(SCENES, name + ".obj"), max_bounces=8)
    plain, keyed = scene.upload(), scene.upload(flags=pt.FLAG_MATERIAL_QUEUES)
    ph = scene.pinhole(w, h)
    want = plain.render(ph, spp, seed=6)
    got = keyed.render(ph, spp, seed=6)
    assert got.tobytes() == want.tobytes()
    # ... and against the oracle itself (Material::sample per path, tracing/src/material.rs:167-191), not only the
    # unflagged HIP path: spp threads x 1 iteration = the same left fold of one-sample buffers
    expect, _ = orc.Scene(os.path.join(SCENES, name + ".obj"), max_bounces=8).render(w, h, threads=spp, iterations_per_thread=1, mode=1, seed=6)
    assert got.tobytes() == expect.tobytes()
    keyed.set_queue_slots(4096)  # far fewer slots than paths: regeneration rounds mix bounces inside a segment
    assert keyed.render(ph, spp, seed=6).tobytes() == want.tobytes()
    keyed.set_queue_slots(0)
    stream = torch.cuda.current_stream().cuda_stream
    a = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
    b = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
    plain.render_region_device(ph, 8, 16, 40, 24, spp, a.data_ptr(), seed=6, stream=stream)
    keyed.render_region_device(ph, 8, 16, 40, 24, spp, b.data_ptr(), seed=6, stream=stream)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    plain.close()
    keyed.close()
