"""tree interchange and statistics — kdtree/src/format.rs:22-47, 94-130 and
kdtree-cli/src/main.rs:45-159.  The product's C++ (std::to_chars) and the oracle's Python (numpy Dragon4)
implement Rust's float printing independently and must agree byte for byte; format.rs carries no tests of
its own, so the absolute anchor is the hand-written table below (Rust's documented `{}` / `{:?}` output)."""
import os
import subprocess

import numpy as np
import pytest

import pathtracer_amd as pt
import pyoracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "scenes")
f32 = np.float32


def one_node_tree(distance, tri_value):
    """X split at `distance` over one triangle whose first coordinate is `tri_value`."""
    tris = np.array([[tri_value, 0, 0, 1, 1, 0, 2, 0, 1]], f32)  # on no axis plane whatever the value
    nodes = np.zeros(3, pt.NODE_DTYPE)
    nodes["w0"][0] = np.array([distance], f32).view(np.uint32)[0]
    nodes["w1"][0] = (2 << 2) | 0
    nodes["w0"][1], nodes["w1"][1] = 0, (1 << 2) | 3
    nodes["w0"][2], nodes["w1"][2] = 1, (0 << 2) | 3
    return pt.KdTree(tris, nodes, np.array([0], np.uint32))


# value, Rust `{}`, Rust `{:?}`
RUST_FLOATS = [
    (1.0, "1", "1.0"), (-1.0, "-1", "-1.0"), (0.5, "0.5", "0.5"), (0.1, "0.1", "0.1"), (-0.0, "-0", "-0.0"), (0.0, "0", "0.0"),
    (1e-7, "0.0000001", "1e-7"), (1.5e-5, "0.000015", "1.5e-5"), (1e-4, "0.0001", "0.0001"),
    (1e16, "10000000000000000", "1e16"), (1.5e16, "15000000000000000", "1.5e16"), (9.999e15, "9999000000000000", "9999000000000000.0"),
    (123456.789, "123456.79", "123456.79"), (16777216.0, "16777216", "16777216.0"), (-0.0028, "-0.0028", "-0.0028"),
    (3.4028235e38, "340282350000000000000000000000000000000", "3.4028235e38"), (1.1754944e-38, "0." + "0" * 37 + "11754944", "1.1754944e-38"),
]


@pytest.mark.parametrize("value,display,debug", RUST_FLOATS)
def test_numbers_print_as_rust_prints_them(value, display, debug):
    tree = one_node_tree(value, value)
    text = tree.to_json()
    assert text == '{"triangles": [[[%s, 0.0, 0.0], [1.0, 1.0, 0.0], [2.0, 0.0, 1.0]]], "root": {"axis": "X", "distance": %s, "left": [0], "right": []}}\n' % (debug, display)
    assert tree.pretty() == "Split X %s\n  Leaf []\n  Leaf [0]\n" % display
    assert orc.rust_display_f32(value) == display and orc.rust_debug_f32(value) == debug
    back = pt.KdTree.from_json(text)
    assert back.nodes.tobytes() == tree.nodes.tobytes() and back.triangles.tobytes() == tree.triangles.tobytes()


@pytest.mark.parametrize("name", ["cornell", "cornell_textured", "cubesplosion", "cornell_teapot"])
def test_json_pretty_and_statistics_equal_the_oracle(name):
    path = os.path.join(SCENES, name + ".obj")
    kd = pt.LoadedScene(path).kdtree()
    ref = orc.Scene(path)
    nodes, leaves = ref.tree().to_flat()
    tris = ref.export()["triangles"]
    text = kd.to_json()
    if name != "cornell_teapot":  # 14 MB of text through Python loops: the small scenes are enough for the oracle side
        assert text == orc.tree_json(tris, nodes, leaves)
        assert kd.pretty() == orc.tree_pretty(nodes, leaves)
        want = orc.tree_statistics(tris, nodes, leaves)
        got = kd.stats()
        assert f32(got["sah_cost"]).tobytes() == f32(want["sah_cost"]).tobytes()
        assert (got["geometries"], got["node_count"], got["leaf_count"]) == (want["geometries"], want["node_count"], want["leaf_count"])
        for key in ("leaf_depth", "leaf_geometries"):
            for field in ("min", "max", "total"):
                assert got[key + "_" + field] == want[key][field]
            for field in ("mean", "median"):
                assert f32(got[key + "_" + field]).tobytes() == f32(want[key][field]).tobytes()
    # write -> read is the identity on the flattened tree, and the axis-aligned canonical form of the triangles
    back = pt.KdTree.from_json(text)
    assert back.nodes.tobytes() == kd.nodes.tobytes() and back.leaf_indices.tobytes() == kd.leaf_indices.tobytes()
    assert back.to_json() == text
    canon = np.stack([orc.any_triangle_arrays(t).reshape(9) for t in kd.triangles])
    assert back.triangles.tobytes() == canon.tobytes()
    # a tree that came in as JSON answers queries like the one it was dumped from
    rng = np.random.default_rng(5)
    for _ in range(50):
        o, d = rng.uniform(-1, 1, 3).astype(f32), rng.normal(size=3).astype(f32)
        a, b = kd.intersect_host(o, d), back.intersect_host(o, d)
        assert a.tobytes() == b.tobytes()


def test_statistics_median_quirk_and_single_leaf():
    """Statistics::compute (main.rs:113-133): even length takes the upper middle, odd length averages the
    middle with its successor; one leaf is its own median."""
    tris = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], f32)
    leaf = np.zeros(1, pt.NODE_DTYPE)
    leaf["w0"], leaf["w1"] = 0, (1 << 2) | 3
    st = pt.KdTree(tris, leaf, np.array([0], np.uint32)).stats()
    assert (st["node_count"], st["leaf_count"], st["leaf_depth_min"], st["leaf_depth_max"]) == (1, 1, 1, 1)
    assert st["leaf_depth_median"] == 1.0 and st["leaf_geometries_median"] == 1.0 and st["leaf_geometries_mean"] == 1.0
    st = one_node_tree(0.25, 0.0).stats()  # two leaves, sizes [0, 1] -> upper middle = 1
    assert st["leaf_geometries_median"] == 1.0 and st["leaf_geometries_mean"] == 0.5 and st["leaf_depth_median"] == 2.0


@pytest.mark.parametrize("text,what", [
    ("", "expected"), ("{}", "triangles"), ('{"triangles": [], "root": [0]}', "invalid tree"),
    ('{"triangles": [], "root": {"axis": "W", "distance": 1, "left": [], "right": []}}', "axis"),
    ('{"triangles": [[[0,0,0],[1,0,0]]], "root": []}', "expected"),
    ('{"triangles": [], "root": [] } trailing', "trailing"),
    ('{"triangles": [], "root": ' + '{"axis": "X", "distance": 0, "left": ' * 30 + "[]" + ', "right": []}' * 30 + "}", "deeper"),
])
def test_malformed_json_is_rejected(text, what):
    with pytest.raises(pt.PtxError) as e:
        pt.KdTree.from_json(text)
    assert what in str(e.value)


def test_kdtree_cli_reports_and_round_trips(tmp_path):
    exe = os.path.join(ROOT, "pathtracer_amd", "bin", "kdtree-cli")
    obj = os.path.join(SCENES, "cornell.obj")
    a = subprocess.run([exe, "-i", obj, "-j"], capture_output=True, text=True, check=True)
    kd = pt.LoadedScene(obj).kdtree()
    assert a.stdout == kd.to_json()
    for line in ("  Geometries: 34", "Building kdtree...", "  Traverse cost: 1.0", "  Intersect cost: 1.5", "  Empty factor: 0.8",
                 "Tree statistics:", "  Node count: 119", "  Leaf count: 60", "    Mean: 8.2", "    Median: 8", "    Total: 75"):
        assert line in a.stderr.splitlines()
    dump = tmp_path / "tree.json"
    dump.write_text(a.stdout)
    b = subprocess.run([exe, "--from-json", str(dump)], capture_output=True, text=True, check=True)
    assert b.stdout == kd.pretty()
    assert b.stderr.split("Tree statistics:")[1] == a.stderr.split("Tree statistics:")[1]
