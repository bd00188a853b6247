"""CPU-side (no GPU) checks of the product library: the C ABI loads and exports what include/ptx.h
declares, the host builder/loader/tonemap/PNG code agrees with the oracle, and device calls fail loudly
without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import pathtracer_amd as pt
import pyoracle as orc

from test_oracle_kat import CUBE, L, N, ray_between, reverse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "scenes")
f32 = np.float32


def test_library_exports_every_declared_symbol():
    text = open(os.path.join(ROOT, "include", "ptx.h")).read()
    names = sorted(set(re.findall(r"\b(ptx_[a-z0-9_]+)\s*\(", text)) - {"ptx_progress_cb"})
    assert len(names) >= 20
    lib = pt.lib()
    for n in names:
        assert hasattr(lib, n), n
    assert "gfx950" in pt.version()


def test_struct_sizes_match_header_contract():
    assert pt.RAY_DTYPE.itemsize == 32 and pt.HIT_DTYPE.itemsize == 16 and pt.NODE_DTYPE.itemsize == 8


@pytest.mark.parametrize(
    "tris,sah",
    [
        ([[[0, 0, 0], [1, 0, 0], [1, 1, 1]], [[1, 0, 0], [2, 0, 0], [2, 1, 1]]], (0.1, 1.0, 0.8)),  # build.rs:53
        ([[[0, 0, 0], [1, 0, 0], [1, 1, 0]], [[0, 0, 1], [1, 0, 1], [1, 1, 1]]], (0.0, 1.0, 1.0)),  # build.rs:85
        (CUBE, (0.0, 1.0, 1.0)),  # build.rs:117
        (CUBE, (1.0, 1.5, 0.8)),
    ],
)
def test_builder_matches_oracle_on_reference_cases(tris, sah):
    mine = pt.build_kdtree(tris, pt.SahCost(*sah))
    ref_nodes, ref_leaf = orc.Tree.build(tris, sah).to_flat()
    assert mine.nodes.tobytes() == ref_nodes.tobytes()
    assert mine.leaf_indices.tobytes() == ref_leaf.tobytes()


def test_builder_reference_cube_tree_exact():  # kdtree/src/build.rs:193-217 through the product builder
    mine = pt.build_kdtree(CUBE, pt.SahCost(0.0, 1.0, 1.0))
    expected = N(0, 0.0, L(8, 9), N(0, 1.0, N(1, 0.0, L(4, 5), N(1, 1.0, N(2, 0.0, L(0, 1), N(2, 1.0, L(), L(2, 3))), L(6, 7))), L(10, 11)))
    nodes, leaf = orc.flat_tree(expected)
    assert mine.nodes.tobytes() == nodes.tobytes() and mine.leaf_indices.tobytes() == leaf.tobytes()


def test_builder_empty_and_degenerate():
    t = pt.build_kdtree(np.zeros((0, 9), f32))
    assert len(t.nodes) == 1 and t.nodes[0]["w1"] == 3
    same = pt.build_kdtree([[[1, 1, 1]] * 3] * 4)  # four identical degenerate triangles
    ref_nodes, ref_leaf = orc.Tree.build([[[1, 1, 1]] * 3] * 4).to_flat()
    assert same.nodes.tobytes() == ref_nodes.tobytes() and same.leaf_indices.tobytes() == ref_leaf.tobytes()


@pytest.mark.parametrize("name", ["cornell", "cornell_textured", "cube", "cornell_teapot", "cubesplosion", "sr2"])
def test_loader_and_builder_match_oracle_on_scenes(name):
    path = os.path.join(SCENES, name + ".obj")
    mine = pt.LoadedScene(path)
    ref = orc.Scene(path, load_textures=False)
    a, e = mine.arrays(), ref.export()
    assert a["triangles"].tobytes() == e["triangles"].tobytes()
    assert a["normals"].tobytes() == e["normals"].tobytes()
    assert a["texcoords"].tobytes() == e["texcoords"].tobytes()
    assert a["material_ids"].tobytes() == e["material_ids"].tobytes()
    assert a["materials"].tobytes() == e["materials"].tobytes()
    assert a["lights"].tobytes() == e["lights"].tobytes()
    assert mine.camera.tobytes() == e["camera"].tobytes()
    ref_nodes, ref_leaf = ref.tree().to_flat()
    assert a["nodes"].tobytes() == ref_nodes.tobytes()
    assert a["leaf_indices"].tobytes() == ref_leaf.tobytes()
    ph = mine.pinhole(640, 360)
    got = np.array([ph.position[:], ph.plane[:], ph.dx[:], ph.dy[:]], f32)
    assert got.tobytes() == ref.pinhole(640, 360).tobytes()


def test_loader_errors_are_reported_not_fatal():
    with pytest.raises(pt.PtxError) as e:
        pt.LoadedScene(os.path.join(SCENES, "does_not_exist.obj"))
    assert e.value.code == -4


def _host_isect(tris, spec, ray, t0, t1):
    nodes, leaf = orc.flat_tree(spec)
    tree = pt.KdTree(np.array(tris, f32).reshape(-1, 9) if len(tris) else np.zeros((0, 9), f32), nodes, leaf)
    h = tree.intersect_host(ray[0], ray[1], (t0, t1))
    return None if h["index"] == pt.MISS else (int(h["index"]), f32(h["t"]), f32(h["u"]), f32(h["v"]))


def test_host_traversal_reference_vectors():  # kdtree/src/lib.rs:202-467 through traverse.hpp (host build)
    assert _host_isect([], L(), ray_between([0, 0, 0], [1, 1, 1]), 0.0, 1.0) is None
    tris = [[[0, 0, -1], [2, 0, -1], [2, 2, -1]], [[0, 0, 1], [2, 0, 1], [2, 2, 1]]]
    ray = ray_between([1, 1, -2], [1, 1, 2])
    spec = N(0, 1.0, L(0, 1), L(0, 1))
    assert _host_isect(tris, spec, ray, 0.0, 1.0) == (0, 0.25, 0.0, 0.5)
    assert _host_isect(tris, spec, reverse(ray), 0.0, 1.0) == (1, 0.25, 0.0, 0.5)
    z1 = [[[0, 0, 1], [1, 0, 1], [0, 1, 1]]]
    ray = ray_between([0, 0, 0], [0, 0, 2])
    for spec in (N(2, 1.0, L(0), L()), N(2, 1.0, L(), L(0)), N(2, 1.0, N(2, 1.0, L(0), L()), L()), N(2, 1.0, L(), N(2, 1.0, L(0), L()))):
        assert _host_isect(z1, spec, ray, 0.0, 1.0) == (0, 0.5, 0.0, 0.0)
        assert _host_isect(z1, spec, reverse(ray), 0.0, 1.0) == (0, 0.5, 0.0, 0.0)
    # full-precision vectors: Möller–Trumbore path (:416-442) ...
    got = _host_isect([[[1, 1, -1], [-1, 1, -1], [1, -1, -1]]], N(2, -1.0, L(), N(2, -1.0, L(0), L())),
                      (np.array([0, 0, 3], f32), np.array([0.06646079, 0.08247295, -0.9238795], f32)), 0.0, pt.FMAX)
    # NB: the product always applies AnyTriangle::from, so this triangle takes the axis-aligned path;
    # the reference test uses a raw `Triangle` — compare against the oracle in AnyTriangle mode instead.
    ref = orc.triangle_intersect([[1, 1, -1], [-1, 1, -1], [1, -1, -1]],
                                 (np.array([0, 0, 3], f32), np.array([0.06646079, 0.08247295, -0.9238795], f32)), as_any=True)
    assert got == (0,) + tuple(ref)
    # ... and the axis-aligned path (:445-467)
    got = _host_isect([[[-1, -1, 1], [-1, -1, -1], [-1, 1, 1]]], N(0, -1.0, L(), L(0)),
                      (np.array([-0.5170438, -0.4394186, -0.045965273], f32), np.array([-0.8491798, -0.1408107, -0.5089852], f32)), 0.0, pt.FMAX)
    assert got == (0, f32(0.5687325), f32(0.66772085), f32(0.24024889))


def test_reference_embedded_sr2_ray_through_the_product_host_path():  # kdtree-tester-cli/src/reducer.rs:104-117
    """Product loader (OBJ face order), product SAH builder and the product's scalar traversal give the
    reference's brute-force answer for the one real-scene ray the reference embeds."""
    from test_oracle_kat import SR2_RAY, SR2_REFERENCE

    tree = pt.LoadedScene(os.path.join(SCENES, "sr2.obj")).kdtree()
    for tmax in (1.0, pt.FMAX):
        h = tree.intersect_host(SR2_RAY[0], SR2_RAY[1], (0.0, tmax))
        assert (int(h["index"]), f32(h["t"]), f32(h["u"]), f32(h["v"])) == (SR2_REFERENCE[0],) + tuple(map(f32, SR2_REFERENCE[1:]))


def test_host_traversal_matches_oracle_on_random_rays():
    scene = pt.LoadedScene(os.path.join(SCENES, "cornell.obj"))
    tree = scene.kdtree()
    ref = orc.Scene(os.path.join(SCENES, "cornell.obj"), load_textures=False).tree()
    rng = np.random.default_rng(5)
    o = rng.uniform(-1.2, 1.2, (2000, 3)).astype(f32)
    d = rng.normal(size=(2000, 3)).astype(f32)
    d[::7, 0] = 0.0  # rays parallel to split planes
    d[::11, 1] = -0.0
    rays = orc.make_rays(o, d)
    want = ref.intersect(rays)
    for i in range(len(rays)):
        h = tree.intersect_host(o[i], d[i])
        assert h.tobytes() == want[i].tobytes(), i


def test_tonemap_matches_oracle():
    rng = np.random.default_rng(1)
    acc = (rng.random((17, 13, 3)) * 40).astype(f32)
    acc[0, 0] = [0, np.nan, -1]
    assert (pt.to_rgb8(acc, 16) == orc.to_rgb8(acc, 16)).all()
    assert (pt.to_rgb8(acc, 65536 + 16) == orc.to_rgb8(acc, 16)).all()  # `as u16` wrap (worker.rs:113)


def test_png_roundtrip_and_texture_decode(tmp_path):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (9, 14, 3), dtype=np.uint8)
    p = str(tmp_path / "x.png")
    pt.write_png(p, img)
    from PIL import Image

    assert (np.asarray(Image.open(p)) == img).all()
    lin = pt.read_png_linear(p)
    assert np.abs(lin - orc.srgb_to_linear_u8()[img]).max() <= 1e-7
    tex = pt.read_png_linear(os.path.join(SCENES, "uvtest.png"))
    ref = orc.load_texture_linear(os.path.join(SCENES, "uvtest.png"))
    assert tex.shape == ref.shape == (1000, 1000, 3)
    assert np.abs(tex - ref).max() <= 1e-7


def test_textured_scene_loads_its_texture():
    s = pt.LoadedScene(os.path.join(SCENES, "cornell_textured.obj"))
    tex_ids = s.arrays()["material_textures"]
    assert sorted(tex_ids)[-1] == 0 and tex_ids.count(0) == 1
    assert s.texture(0).shape == (1000, 1000, 3)


def test_scene_create_rejects_bad_trees_before_touching_the_gpu():
    tris = np.array(CUBE, f32).reshape(-1, 9)
    t = pt.build_kdtree(tris)
    bad = t.nodes.copy()
    inner = np.where((bad["w1"] & 3) != 3)[0][0]
    bad["w1"][inner] = (len(bad) + 5 << 2) | (bad["w1"][inner] & 3)
    with pytest.raises(pt.PtxError) as e:
        pt.DeviceScene.from_arrays(tris, bad, t.leaf_indices)
    assert e.value.code == -1
    bad_leaf = t.leaf_indices.copy()
    bad_leaf[0] = 999
    with pytest.raises(pt.PtxError) as e:
        pt.DeviceScene.from_arrays(tris, t.nodes, bad_leaf)
    assert e.value.code == -1


def test_device_calls_fail_loudly_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    t = pt.build_kdtree(CUBE)
    with pytest.raises(pt.PtxError) as e:
        t.upload()
    assert e.value.code == -2  # PTX_E_HIP: no fallback


def test_serial_and_parallel_builds_are_the_same_tree():
    """the builder grows the top subtrees on several threads; PTX_BUILD_THREADS=1 keeps it on one."""
    import subprocess
    import sys

    code = ("import sys, hashlib; sys.path.insert(0, %r); import pathtracer_amd as pt; "
            "kd = pt.LoadedScene(%r).kdtree(); print(hashlib.sha256(kd.nodes.tobytes() + kd.leaf_indices.tobytes()).hexdigest())"
            % (ROOT, os.path.join(ROOT, "scenes", "cornell_teapot.obj")))
    digests = {subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PTX_BUILD_THREADS=t), capture_output=True, text=True,
                              check=True).stdout.strip() for t in ("1", "2", "16")}
    assert len(digests) == 1 and len(digests.pop()) == 64


def test_accum_checkpoint_round_trip_and_damage(tmp_path):
    """ptx_accum_save / ptx_accum_load: bits, header, and the ways a file can be wrong."""
    rng = np.random.default_rng(3)
    acc = rng.random((13, 17, 3), dtype=np.float32) * 40
    acc[0, 0] = [0.0, np.float32(1e-40), np.float32(3e38)]
    path = str(tmp_path / "a.ptxacc")
    pt.save_accum(path, acc, seed=2**63 + 5, samples=77)
    got, seed, n = pt.load_accum(path)
    assert got.tobytes() == acc.tobytes() and seed == 2**63 + 5 and n == 77
    assert os.path.getsize(path) == 32 + acc.nbytes and not os.path.exists(path + ".tmp")
    raw = bytearray(open(path, "rb").read())
    for damage in ("flip", "short", "long", "magic"):
        bad = bytearray(raw)
        if damage == "flip":
            bad[100] ^= 1
        elif damage == "short":
            bad = bad[:-4]
        elif damage == "long":
            bad += b"\0"
        else:
            bad[0] = ord("X")
        open(path, "wb").write(bad)
        with pytest.raises(pt.PtxError):
            pt.load_accum(path)
    with pytest.raises(pt.PtxError):
        pt.load_accum(str(tmp_path / "missing"))
    # a buffer of another size is refused
    open(path, "wb").write(raw)
    import ctypes as C

    w, h = C.c_uint32(16), C.c_uint32(13)
    buf = np.zeros((13, 16, 3), np.float32)
    assert pt.lib().ptx_accum_load(os.fsencode(path), buf.ctypes.data_as(C.c_void_p), C.byref(w), C.byref(h), None, None) == -1  # PTX_E_ARG


def test_shard_samples_tiles_the_range_exactly():
    """ptx_shard_samples (worker.rs:92-104 as contiguous ranges): GPU g renders [first + g*S/N, first + (g+1)*S/N)."""
    for first, count, n in [(0, 1024, 8), (5, 7, 3), (0, 2, 4), (100, 0, 2), (0, 4096, 7), (2**32 - 11, 10, 4)]:
        at = first
        for g in range(n):
            f, c = pt.shard_samples(first, count, n, g)
            assert f == at and c in (count // n, count // n + 1)
            at = f + c
        assert at == first + count
    for bad in [(0, 10, 0, 0), (0, 10, 2, 2), (2**32 - 5, 10, 2, 0)]:
        with pytest.raises(pt.PtxError):
            pt.shard_samples(*bad)
    from pathtracer_amd.parallel import shard_samples  # the per-process path uses the same definition

    assert [shard_samples(10, 4, r) for r in range(4)] == [(0, 2), (2, 3), (5, 2), (7, 3)]


def test_render_multi_rejects_bad_arguments_before_touching_a_gpu():
    L = pt.lib()
    ph = pt.Pinhole()
    ph.width = ph.height = 4
    acc = np.zeros(48, f32)
    none = (C.c_void_p * 1)(None)
    assert L.ptx_render_multi(None, 0, C.byref(ph), C.c_uint64(0), 0, 1, acc.ctypes.data_as(C.c_void_p), 0, None, None) == -1
    assert L.ptx_render_multi(none, 1, C.byref(ph), C.c_uint64(0), 0, 1, acc.ctypes.data_as(C.c_void_p), 0, None, None) == -1
    assert L.ptx_render_multi(none, 1, C.byref(ph), C.c_uint64(0), 0, 1, acc.ctypes.data_as(C.c_void_p), 7, None, None) == -1
    assert b"reduce" in L.ptx_last_error()
    assert L.ptx_render_multi(none, 65, C.byref(ph), C.c_uint64(0), 0, 1, acc.ctypes.data_as(C.c_void_p), 0, None, None) == -5
    n = C.c_int(-1)
    assert L.ptx_device_count(C.byref(n)) == 0 and n.value >= 0


def test_host_traversal_rejects_a_malformed_tree():  # validated before the walk, like ptx_scene_create
    tris = np.array([[[0, 0, 1], [1, 0, 1], [0, 1, 1]]], f32).reshape(-1, 9)
    nodes, leaf = orc.flat_tree(N(2, 1.0, L(0), L()))
    ok = pt.KdTree(tris, nodes, leaf)
    assert ok.intersect_host([0, 0, 0], [0, 0, 2], (0.0, 1.0))["index"] == 0
    bad = nodes.copy()
    bad["w1"][0] = (50 << 2) | 2  # right child far outside the array
    with pytest.raises(pt.PtxError):
        pt.KdTree(tris, bad, leaf).intersect_host([0, 0, 0], [0, 0, 2], (0.0, 1.0))
    with pytest.raises(pt.PtxError):  # a leaf that names a triangle the caller did not pass
        pt.KdTree(tris, nodes, np.array([7], np.uint32)).intersect_host([0, 0, 0], [0, 0, 2], (0.0, 1.0))


def test_parallel_obj_parse_equals_the_serial_pass(tmp_path):
    """the OBJ file is cut at line ends and its parts parsed on several threads; arrays, group
    boundaries and the first error are those of one sequential pass (PTX_BUILD_THREADS=1)."""
    import hashlib
    import subprocess
    import sys

    # a file large enough to be cut (> 256 KiB): 30 000 triangles in 3 groups, the usemtl lines far from the cuts' ends
    rng = np.random.default_rng(3)
    lines = ["mtllib big.mtl"]
    n = 30000
    for i in range(n):
        if i % 10000 == 0:
            lines.append("usemtl m%d" % (i // 10000))
        p = rng.random((3, 3)).astype(f32) + i % 7
        for v in p:
            lines.append("v %.6f %.6f %.6f" % tuple(v))
        lines.append("vn 0 0 1")
        lines.append("f %d//%d %d//%d %d//%d" % (3 * i + 1, i + 1, 3 * i + 2, i + 1, 3 * i + 3, i + 1))
    (tmp_path / "big.mtl").write_text("newmtl m0\nKd 0.5 0.5 0.5\nnewmtl m1\nKd 0.1 0.9 0.1\nnewmtl m2\nKd 0.9 0.1 0.1\n"
                                      "newcamera c\ncameraposition 0 0 5\ncameratarget 0 0 0\ncameraup 0 1 0\ncamerafov 45\n")
    obj = tmp_path / "big.obj"
    obj.write_text("\n".join(lines) + "\n")
    code = ("import sys, hashlib; sys.path.insert(0, %r); import pathtracer_amd as pt; a = pt.LoadedScene(%r).arrays(); "
            "print(hashlib.sha256(a['triangles'].tobytes() + a['normals'].tobytes() + a['material_ids'].tobytes()).hexdigest(), len(a['triangles']), a['material_ids'][[0, 9999, 10000, 29999]].tolist())"
            % (ROOT, str(obj)))
    outs = {subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PTX_BUILD_THREADS=t), capture_output=True, text=True,
                           check=True).stdout.strip() for t in ("1", "3", "16")}
    assert len(outs) == 1
    out = outs.pop()
    assert out.endswith("30000 [0, 0, 1, 2]")
    # errors: the FIRST bad line of the file is the one reported, wherever the cuts fall
    bad = lines[:]
    bad[len(bad) // 3] = "bogus first"
    bad[2 * len(bad) // 3] = "bogus second"
    obj.write_text("\n".join(bad) + "\n")
    for t in ("1", "8"):
        os.environ["PTX_BUILD_THREADS"] = t
        try:
            with pytest.raises(pt.PtxError) as e:
                pt.LoadedScene(str(obj))
        finally:
            del os.environ["PTX_BUILD_THREADS"]
        assert "bogus first" in str(e.value) and "second" not in str(e.value)
    obj.write_text("\n".join(l for l in lines if not l.startswith("usemtl")) + "\n")  # faces, but no group ever opens
    with pytest.raises(pt.PtxError) as e:
        pt.LoadedScene(str(obj))
    assert "usemtl" in str(e.value)


def _walk_device_tree(kd, words, records, root):
    """Walks the ABI tree and the device tree side by side; returns the number of leaves compared."""
    nodes = np.asarray(kd.nodes).view(np.uint32).reshape(-1, 2)
    leaf_idx = np.asarray(kd.leaf_indices)
    stack, leaves = [(0, root)], 0
    while stack:
        src, desc = stack.pop()
        w0, w1 = int(nodes[src, 0]), int(nodes[src, 1])
        if w1 & 3 == 3:  # leaf: the descriptor names a run that reproduces the list, in order, flagged on its last record
            want = [int(x) for x in leaf_idx[w0:w0 + (w1 >> 2)]]
            if not want:
                assert desc == -1
            else:
                assert desc < -2
                off = desc & 0x3FFFFFFF
                got = [int(r) for r in records[off:off + len(want)]]
                assert [g & 0x3FFFFFFF for g in got] == want
                assert [g >> 31 for g in got] == [0] * (len(want) - 1) + [1]
            leaves += 1
            continue
        assert desc >= 0
        x, y, z, w = (int(v) for v in words[desc])
        assert x == w0 and y == (w1 & 3)
        to_i32 = lambda v: v - (1 << 32) if v & 0x80000000 else v
        stack.append((w1 >> 2, to_i32(w)))
        stack.append((src + 1, to_i32(z)))
    return leaves


@pytest.mark.parametrize("name", ["cornell", "cornell_teapot", "sr2"])
def test_device_tree_layout(name):
    """The layout ptx_scene_create hands to the kernels (host_devtree.cpp) is the same tree: every inner node keeps its
    plane and axis, every leaf its triangles in order; equal lists and tails share record runs; the LDS prefix is
    breadth-first and closed under 'parent of'."""
    scene = pt.LoadedScene(os.path.join(SCENES, name + ".obj"))
    kd = scene.kdtree()
    words, records, root, top_n = kd.device_layout(top_nodes=1023)
    nodes = np.asarray(kd.nodes).view(np.uint32).reshape(-1, 2)
    n_leaves = int(((nodes[:, 1] & 3) == 3).sum())
    assert len(words) == len(nodes) - n_leaves
    assert _walk_device_tree(kd, words, records, root) == n_leaves
    # prefix: nodes [0, top_n) are reached from the root without leaving the prefix, and top_n is as large as allowed
    assert top_n == min(1023, len(words))
    inner_children = [int(c) for c in words[:top_n, 2:4].ravel() if int(c) < 0x80000000]
    assert set(range(1, top_n)) <= set(inner_children)
    # sharing: never more records than references, and far fewer on the big trees
    assert len(records) <= len(kd.leaf_indices)
    if name != "cornell":
        assert len(records) < 0.55 * len(kd.leaf_indices)
    # a different prefix size gives the same tree
    w2, r2, root2, top2 = kd.device_layout(top_nodes=255)
    assert top2 == min(255, len(w2)) and _walk_device_tree(kd, w2, r2, root2) == n_leaves
    assert np.array_equal(r2, records)


def test_device_tree_huge_leaves_stay_cheap():
    """A depth-capped or heavily overlapping scene can hold leaves of tens of thousands of triangles.  Tail sharing is
    keyed by the bytes of a list, so it is only offered for tails of at most 64 triangles: a 20 000-triangle leaf costs
    its own length in keys, not its length squared (2 x 800 MB before), and the layout stays correct: the long list is
    stored once, a short tail of it points into its run, a long tail gets a run of its own."""
    import resource
    import time

    n = 20_000
    tris = np.zeros((n, 9), np.float32)
    tris[:, 0], tris[:, 4], tris[:, 8] = np.arange(n), 1.0, 1.0  # (never intersected here)
    nodes = np.zeros(5, pt.NODE_DTYPE)  # root(x) -> [leaf A | node(y) -> [leaf B | leaf C]]
    leaf = np.concatenate([np.arange(n), np.arange(n - 40, n), np.arange(100, n)]).astype(np.uint32)
    nodes["w0"][0], nodes["w1"][0] = np.float32(0.5).view(np.uint32), (2 << 2) | 0
    nodes["w0"][1], nodes["w1"][1] = 0, (n << 2) | 3                     # A: all n triangles
    nodes["w0"][2], nodes["w1"][2] = np.float32(0.5).view(np.uint32), (4 << 2) | 1
    nodes["w0"][3], nodes["w1"][3] = n, (40 << 2) | 3                    # B: the last 40 of A (a short tail: shared)
    nodes["w0"][4], nodes["w1"][4] = n + 40, ((n - 100) << 2) | 3        # C: A without its first 100 (a long tail: own run)
    kd = pt.KdTree(tris, nodes, leaf)
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    t0 = time.perf_counter()
    words, records, root, top_n = kd.device_layout()
    assert time.perf_counter() - t0 < 5.0
    assert resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - before < 200_000  # KiB; the quadratic version needed GBs
    assert len(words) == 2 and len(records) == n + (n - 100)
    assert _walk_device_tree(kd, words, records, root) == 3
    a_desc, b_desc, c_desc = int(words[0, 2]), int(words[1, 2]), int(words[1, 3])
    assert a_desc == 0x80000000 and b_desc == 0x80000000 | (n - 40) and c_desc == 0x80000000 | n


def test_device_tree_rejects_malformed():
    scene = pt.LoadedScene(os.path.join(SCENES, "cornell.obj"))
    kd = scene.kdtree()
    bad = pt.KdTree(kd.triangles, kd.nodes.copy(), kd.leaf_indices[:-1])
    with pytest.raises(pt.PtxError):
        bad.device_layout()
