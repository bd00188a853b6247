"""The CPU restatement (oracle/) against every known-answer test the reference holds for the hot path.

Vectors are transcribed as data from the reference's inline tests (file:line cited per test); all
comparisons are exact f32 equality unless the reference itself uses ulps.
"""
import numpy as np
import pytest

import pyoracle as orc

f32 = np.float32
FMAX = orc.FMAX


def ray_between(a, b):
    a = np.array(a, f32)
    b = np.array(b, f32)
    return a, (b - a).astype(f32)


def reverse(ray):
    o, d = ray
    return (o + d).astype(f32), (-d).astype(f32)


UNIT = [[0, 0, 0], [1, 0, 0], [0, 1, 0]]


# ---------------------------------------------------------------- geometry/src/triangle.rs:188-427
@pytest.mark.parametrize(
    "xy,expect",
    [
        ((0.5, 0.5), (0.5, 0.5, 0.5)),  # base_center :189  (v0 + .5 b0 + .5 b1)
        ((0.0, 0.0), (0.5, 0.0, 0.0)),  # v0 :211
        ((1.0, 0.0), (0.5, 1.0, 0.0)),  # v1 :233
        ((0.0, 1.0), (0.5, 0.0, 1.0)),  # v2 :255
        ((0.5, 0.0), (0.5, 0.5, 0.0)),  # edge0 :277
        ((0.5, 0.5), (0.5, 0.5, 0.5)),  # edge1 :300
        ((0.0, 0.5), (0.5, 0.0, 0.5)),  # edge2 :323
    ],
)
def test_triangle_intersect_points(xy, expect):
    ray = ray_between([xy[0], xy[1], -1.0], [xy[0], xy[1], 1.0])
    assert orc.triangle_intersect(UNIT, ray) == tuple(map(f32, expect))


def test_triangle_parallel():  # triangle.rs:346-373
    assert orc.triangle_intersect(UNIT, ray_between([0, 0, 0], [1, 0, 0])) is None
    assert orc.triangle_intersect(UNIT, ray_between([0, 0, 1], [1, 0, 1])) is None


def test_triangle_almost_parallel_touching():  # triangle.rs:376-395
    got = orc.triangle_intersect(UNIT, ray_between([0, 0, -0.000001], [1, 0, 0.000001]))
    assert got == (f32(0.5), f32(0.5), f32(0.0))


def test_triangle_orientation():  # triangle.rs:398-427
    ray = ray_between([0.5, 0.0, -1.0], [0.5, 0.0, 1.0])
    assert orc.triangle_intersect([[0, 0, 0], [1, 0, 0], [0, 1, 0]], ray) == (f32(0.5), f32(0.5), f32(0.0))
    assert orc.triangle_intersect([[0, 0, 0], [0, 1, 0], [1, 0, 0]], ray) == (f32(0.5), f32(0.0), f32(0.5))


# ---------------------------------------------------------------- geometry/src/axial_triangle.rs:103-156
AX_V = [[0, 0], [1, 0], [0, 1]]


def test_axial_intersect_point():
    assert orc.axial_intersect_point(0, 0.0, AX_V, [2.0, 2.0]) is None  # :104
    assert orc.axial_intersect_point(0, 0.0, AX_V, [1.0, 0.0]) == (1.0, 0.0)  # :109
    assert orc.axial_intersect_point(0, 0.0, AX_V, [0.0, 1.0]) == (0.0, 1.0)  # :117
    assert orc.axial_intersect_point(0, 0.0, AX_V, [0.5, 0.5]) == (0.5, 0.5)  # :125
    assert orc.axial_intersect_point(0, 0.0, AX_V, [0.5, 0.0]) == (0.5, 0.0)  # :133 positive
    assert orc.axial_intersect_point(0, 0.0, [[0, 0], [0, 1], [1, 0]], [0.5, 0.0]) == (0.0, 0.5)  # negative


def test_as_axially_aligned_axis_order():  # triangle.rs:59-77: X first, then Y, then Z
    assert orc.axial_axis([[1, 2, 3], [1, 5, 6], [1, 8, 9]]) == 0
    assert orc.axial_axis([[0, 2, 3], [1, 2, 6], [4, 2, 9]]) == 1
    assert orc.axial_axis([[0, 2, 3], [1, 5, 3], [4, 8, 3]]) == 2
    assert orc.axial_axis([[1, 2, 3], [1, 2, 6], [1, 2, 9]]) == 0  # degenerate: X wins
    assert orc.axial_axis([[0, 0, 0], [1, 0, 0], [1, 1, 1]]) == -1


# ---------------------------------------------------------------- geometry/src/aap.rs:80-191
@pytest.mark.parametrize(
    "axis,dist,a,b,expect",
    [
        (0, 5.0, [0, 0, 0], [5, 0, 0], 1.0),
        (0, 5.0, [0, 0, 0], [10, 0, 0], 0.5),
        (0, 5.0, [0, 0, 0], [4, 0, 0], 1.25),
        (0, 5.0, [4, 0, 0], [6, 0, 0], 0.5),
        (0, 5.0, [6, 0, 0], [4, 0, 0], 0.5),
        (0, 2.0, [1, 1, 0], [3, 3, 0], 0.5),
        (1, 2.0, [1, 1, 0], [3, 3, 0], 0.5),
        (1, 2.0, [3, 1, 0], [1, 3, 0], 0.5),
        (0, 2.0, [0, 0, 0], [0, 1, 0], None),
    ],
)
def test_aap_intersect(axis, dist, a, b, expect):
    got = orc.aap_intersect(axis, dist, ray_between(a, b))
    assert got == (None if expect is None else f32(expect))


def test_aap_negative_zero_direction_is_parallel():  # aap.rs:46 `denom == 0.0` is true for -0.0
    assert orc.aap_intersect(0, 2.0, (np.zeros(3, f32), np.array([-0.0, 1.0, 0.0], f32))) is None


# ---------------------------------------------------------------- geometry/src/clip.rs:97-287
BOX3 = [[0, 0, 0], [3, 3, 3]]


def _clip(v0, v1, v2, aabb):
    return orc.clip_points([v0, v1, v2], aabb)


def assert_points(actual, expected):
    expected = np.array(expected, f32).reshape(-1, 3)
    assert actual.shape == expected.shape, (actual, expected)
    assert (actual.view(np.uint32) == expected.view(np.uint32)).all() or (actual == expected).all(), (actual, expected)


def test_clip_enclosed():  # :98
    v0, v1, v2 = [1, 1, 1], [2, 1, 1], [2, 2, 1]
    assert_points(_clip(v0, v1, v2, BOX3), [v1, v2, v0])


def test_clip_in_box_side():  # :111
    v0, v1, v2 = [1, 1, 0], [2, 1, 0], [2, 2, 0]
    assert_points(_clip(v0, v1, v2, BOX3), [v1, v2, v0])


def test_clip_flat_box_fully():  # :124
    v0, v1, v2 = [1, 1, 0], [2, 1, 0], [2, 2, 0]
    assert_points(_clip(v0, v1, v2, [[0, 0, 0], [3, 3, 0]]), [v1, v2, v0])


def test_clip_flat_box_line():  # :137
    got = _clip([1, 1, -1], [2, 1, 0], [2, 2, 1], [[0, 0, 0], [3, 3, 0]])
    assert_points(got, [[2.0, 1.0, 0.0], [1.5, 1.5, 0.0]])


def test_clip_above_below():  # :150, :163
    assert len(_clip([0, 2, 0], [1, 2, 0], [1, 2, 1], [[0, 0, 0], [1, 1, 1]])) == 0
    assert len(_clip([0, -1, 0], [1, -1, 0], [1, -1, 1], [[0, 0, 0], [1, 1, 1]])) == 0


def test_clip_all_edges():  # :176
    got = _clip([0, 0, 0], [12, 0, 0], [6, 6, 0], [[2, -1, 0], [10, 4, 0]])
    assert_points(got, [[2, 0, 0], [10, 0, 0], [10, 2, 0], [8, 4, 0], [4, 4, 0], [2, 2, 0]])


def test_clip_rounding_example_1():  # :196
    tri = ([-1, -1, -1], [-1, -1, 1], [1, -1, -1])
    aabb = [[-1.5, -1.5012, -1.5], [-0.076, 1.5, 1.0]]
    assert_points(_clip(*tri, aabb), [[-1, -1, 1], [-0.076, -1.0, 0.075999975], [-0.076, -1.0, -1.0], [-1, -1, -1]])
    box = orc.clip_aabb(list(tri), aabb)
    assert_points(box, [[-1, -1, -1], [-0.076, -1.0, 1.0]])


def test_clip_rounding_example_2():  # :218
    tri = ([-1, -1, -1], [-1, -1, 1], [1, -1, -1])
    aabb = [[-1.5, -1.5012, -1.5], [-0.076, 0.075999975, 0.075999975]]
    assert_points(
        _clip(*tri, aabb),
        [[-1.0, -1.0, 0.075999975], [-0.075999975, -1.0, 0.075999975], [-0.076, -1.0, 0.075999975], [-0.076, -1.0, -1.0], [-1, -1, -1]],
    )
    assert_points(orc.clip_aabb(list(tri), aabb), [[-1, -1, -1], [-0.076, -1.0, 0.075999975]])


def test_clip_outside_two_planes():  # :246 (note v2 declared before v1 in the reference)
    v0 = [3.835834, 0.136162, -3.724971]
    v2 = [3.836198, 0.135679, -4.556344]
    v1 = [3.952836, 0.369915, -4.555017]
    aabb = [[3.8359935, 0.241052, -4.272935], [3.901177, 0.274277, -4.089322]]
    assert_points(
        _clip(v0, v1, v2, aabb),
        [[3.901177, 0.2665847, -4.272935], [3.8884628, 0.241052, -4.272935], [3.8883352, 0.241052, -4.0974307], [3.901177, 0.2667079, -4.1885333]],
    )


def test_clip_max_capacity():  # :267
    got = _clip([1.8820591, 1.8356464, -0.024532795], [0.6663188, 0.14820933, 0.0025525093], [-0.74280226, -0.56288826, 1.474189], [[0, 0, 0], [1, 1, 1]])
    assert_points(
        got,
        [
            [1.0, 1.0, 0.44480032],
            [1.0, 0.61557496, 0.0],
            [0.78088975, 0.30723286, 0.0],
            [0.6663188, 0.14820933, 0.0025525093],
            [0.37262508, 0.0, 0.3092759],
            [0.0, 0.0, 0.9160062],
            [0.0, 0.07259283, 1.0],
            [0.08769246, 0.19599774, 1.0],
            [0.9675606, 1.0, 0.49762005],
        ],
    )


def test_bounding_box():  # geometry/src/bound.rs:29-48
    tris = [[[1, 1, 0], [1, 1, 1], [0, 0, 0]], [[-1, -1, 0], [-1, -1, -1], [0, 0, 0]]]
    assert_points(orc.bounding_box(tris), [[-1, -1, -1], [1, 1, 1]])


# ---------------------------------------------------------------- kdtree/src/lib.rs:202-467
def L(*idx):
    return ("leaf", list(idx))


def N(axis, dist, l, r):
    return ("node", axis, dist, l, r)


def tree_of(tris, spec, raw=True):
    nodes, leaf = orc.flat_tree(spec)
    return orc.Tree.from_flat(np.array(tris, f32).reshape(-1, 9) if len(tris) else np.zeros((0, 9), f32), nodes, leaf, raw_triangles=raw)


def isect(tree, ray, t0, t1):
    h = tree.intersect(orc.make_rays([ray[0]], [ray[1]], t0, t1))[0]
    if h["index"] == orc.MISS:
        return None
    return int(h["index"]), f32(h["t"]), f32(h["u"]), f32(h["v"])


def test_kd_intersect_empty_node():  # :203
    t = tree_of([], L())
    assert isect(t, ray_between([0, 0, 0], [1, 1, 1]), 0.0, 1.0) is None


def test_kd_ray_through_split_plane_both_triangles():  # :212
    tris = [[[0, 0, -1], [2, 0, -1], [2, 2, -1]], [[0, 0, 1], [2, 0, 1], [2, 2, 1]]]
    t = tree_of(tris, N(0, 1.0, L(0, 1), L(0, 1)))
    ray = ray_between([1, 1, -2], [1, 1, 2])
    assert isect(t, ray, 0.0, 1.0) == (0, 0.25, 0.0, 0.5)
    assert isect(t, reverse(ray), 0.0, 1.0) == (1, 0.25, 0.0, 0.5)


def test_kd_ray_parallel_to_split_plane():  # :251
    tris = [[[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[1, 0, 0], [2, 0, 0], [2, 1, 0]]]
    t = tree_of(tris, N(0, 1.0, L(0), L(1)))
    assert isect(t, ray_between([0, 0, -1], [0, 0, 1]), 0.0, 1.0) == (0, 0.5, 0.0, 0.0)
    assert isect(t, ray_between([2, 0, -1], [2, 0, 1]), 0.0, 1.0) == (1, 0.5, 1.0, 0.0)


def test_kd_ray_orthogonal_to_split_plane():  # :288
    tris = [[[0, -1, -1], [0, 1, -1], [0, 1, 1]], [[2, -1, -1], [2, 1, -1], [2, 1, 1]]]
    t = tree_of(tris, N(0, 1.0, L(0), L(1)))
    ray = ray_between([-1, 0, 0], [3, 0, 0])
    assert isect(t, ray, 0.0, 1.0) == (0, 0.25, 0.0, 0.5)
    assert isect(t, reverse(ray), 0.0, 1.0) == (1, 0.25, 0.0, 0.5)


Z1 = [[[0, 0, 1], [1, 0, 1], [0, 1, 1]]]


def test_kd_split_at_axially_aligned_triangle():  # :324
    ray = ray_between([0, 0, 0], [0, 0, 2])
    assert isect(tree_of(Z1, N(2, 1.0, L(0), L())), ray, 0.0, 1.0) == (0, 0.5, 0.0, 0.0)
    assert isect(tree_of(Z1, N(2, 1.0, L(), L(0))), ray, 0.0, 1.0) == (0, 0.5, 0.0, 0.0)


def test_kd_flat_cell_left_left():  # :354
    t = tree_of(Z1, N(2, 1.0, N(2, 1.0, L(0), L()), L()))
    ray = ray_between([0, 0, 0], [0, 0, 2])
    assert isect(t, ray, 0.0, 1.0) == (0, 0.5, 0.0, 0.0)
    assert isect(t, reverse(ray), 0.0, 1.0) == (0, 0.5, 0.0, 0.0)


def test_kd_flat_cell_right_left():  # :385
    t = tree_of(Z1, N(2, 1.0, L(), N(2, 1.0, L(0), L())))
    ray = ray_between([0, 0, 0], [0, 0, 2])
    assert isect(t, ray, 0.0, 1.0) == (0, 0.5, 0.0, 0.0)
    assert isect(t, reverse(ray), 0.0, 1.0) == (0, 0.5, 0.0, 0.0)


def test_kd_flat_cell_minimized_example():  # :416-442 — pins glam dot/cross order on the Möller–Trumbore path
    tris = [[[1, 1, -1], [-1, 1, -1], [1, -1, -1]]]
    t = tree_of(tris, N(2, -1.0, L(), N(2, -1.0, L(0), L())), raw=True)
    ray = (np.array([0, 0, 3], f32), np.array([0.06646079, 0.08247295, -0.9238795], f32))
    assert isect(t, ray, 0.0, FMAX) == (0, f32(4.329569), f32(0.35612673), f32(0.32146382))


def test_kd_rounding_error_example():  # :445-467 — AnyTriangle::from → axis-aligned path
    tris = [[[-1, -1, 1], [-1, -1, -1], [-1, 1, 1]]]
    t = tree_of(tris, N(0, -1.0, L(), L(0)), raw=False)
    ray = (np.array([-0.5170438, -0.4394186, -0.045965273], f32), np.array([-0.8491798, -0.1408107, -0.5089852], f32))
    assert isect(t, ray, 0.0, FMAX) == (0, f32(0.5687325), f32(0.66772085), f32(0.24024889))


# ---------------------------------------------------------------- kdtree/src/event.rs:105-190
def test_event_list_single_triangle():
    tris = [[[0, 0, 0], [1, 1, 0], [1, 1, 1]]]
    for axis in range(3):
        assert orc.event_list(tris, axis) == [("start", 0, 0.0), ("end", 0, 1.0)]


def test_event_list_planar_triangle():
    tris = [[[0, 0, 0], [0, 1, 0], [0, 1, 1]]]
    assert orc.event_list(tris, 0) == [("planar", 0, 0.0)]
    assert orc.event_list(tris, 1) == [("start", 0, 0.0), ("end", 0, 1.0)]
    assert orc.event_list(tris, 2) == [("start", 0, 0.0), ("end", 0, 1.0)]


def test_event_list_overlapping():
    tris = [[[0, 0, 0], [1, 1, 0], [1, 1, 1]], [[1, 1, 1], [1, 2, 1], [1, 2, 2]], [[1, 1, 1], [2, 2, 1], [2, 2, 2]]]
    assert orc.event_list(tris, 0) == [("start", 0, 0.0), ("end", 0, 1.0), ("planar", 1, 1.0), ("start", 2, 1.0), ("end", 2, 2.0)]
    yz = [("start", 0, 0.0), ("end", 0, 1.0), ("start", 1, 1.0), ("start", 2, 1.0), ("end", 1, 2.0), ("end", 2, 2.0)]
    for axis in (1, 2):
        got = orc.event_list(tris, axis)
        # sort_unstable: order among equal (distance, kind) is unspecified; compare as key-sorted
        assert sorted(got, key=lambda e: (e[2], ["end", "planar", "start"].index(e[0]), e[1])) == yz


# ---------------------------------------------------------------- kdtree/src/sah.rs:359-442
def test_update_geometry_side():
    tri = [[[0, 0, 0], [1, 0, 0], [1, 1, 0]]]
    assert orc.update_geometry_side(tri, 0, 1.0, False) == ["left"]
    assert orc.update_geometry_side(tri, 0, 0.0, False) == ["right"]
    assert orc.update_geometry_side(tri, 0, 0.5, False) == ["both"]
    assert orc.update_geometry_side(tri, 2, 0.0, False) == ["left"]
    assert orc.update_geometry_side(tri, 2, 0.0, True) == ["right"]


# ---------------------------------------------------------------- kdtree/src/build.rs:52-223
def test_build_two_oriented_triangles():
    tris = [[[0, 0, 0], [1, 0, 0], [1, 1, 1]], [[1, 0, 0], [2, 0, 0], [2, 1, 1]]]
    t = orc.Tree.build(tris, (0.1, 1.0, 0.8))
    assert t.string() == orc.tree_spec_string(N(0, 1.0, L(0), L(1)))


def test_build_two_axially_aligned_triangles():
    tris = [[[0, 0, 0], [1, 0, 0], [1, 1, 0]], [[0, 0, 1], [1, 0, 1], [1, 1, 1]]]
    t = orc.Tree.build(tris, (0.0, 1.0, 1.0))
    assert t.string() == orc.tree_spec_string(N(2, 0.0, L(0), N(2, 1.0, L(), L(1))))


CUBE = [
    [[0, 0, 0], [1, 0, 0], [1, 1, 0]], [[0, 0, 0], [0, 1, 0], [1, 1, 0]],  # front
    [[0, 0, 1], [1, 0, 1], [1, 1, 1]], [[0, 0, 1], [0, 1, 1], [1, 1, 1]],  # back
    [[0, 0, 0], [1, 0, 0], [1, 0, 1]], [[0, 0, 0], [0, 0, 1], [1, 0, 1]],  # bottom
    [[0, 1, 0], [1, 1, 0], [1, 1, 1]], [[0, 1, 0], [0, 1, 1], [1, 1, 1]],  # top
    [[0, 0, 0], [0, 0, 1], [0, 1, 1]], [[0, 0, 0], [0, 1, 0], [0, 1, 1]],  # right
    [[1, 0, 0], [1, 1, 0], [1, 1, 1]], [[1, 0, 0], [1, 0, 1], [1, 1, 1]],  # left
]


def test_build_one_cube():
    t = orc.Tree.build(CUBE, (0.0, 1.0, 1.0))
    expected = N(0, 0.0, L(8, 9), N(0, 1.0, N(1, 0.0, L(4, 5), N(1, 1.0, N(2, 0.0, L(0, 1), N(2, 1.0, L(), L(2, 3))), L(6, 7))), L(10, 11)))
    assert t.string() == orc.tree_spec_string(expected)


def test_flat_roundtrip_cube():
    t = orc.Tree.build(CUBE, (0.0, 1.0, 1.0))
    nodes, leaf = t.to_flat()
    t2 = orc.Tree.from_flat(CUBE, nodes, leaf)
    assert t2.string() == t.string()


# ---------------------------------------------------------------- tracing/src/material.rs:201-370
def test_refraction_entering():  # :202
    s = orc.sample_refraction([-1, 0, 0], [1, 0, 0], 1.5, [1, 1, 1], [1, 1, 1], 1.0)
    assert (s["wo"] == np.array([1, 0, 0], f32)).all()
    assert (s["bsdf"] == f32(2.25)).all()


def test_refraction_exiting():  # :220
    s = orc.sample_refraction([1, 0, 0], [1, 0, 0], 1.5, [1, 1, 1], [1, 1, 1], 1.0)
    assert (s["wo"] == np.array([-1, 0, 0], f32)).all()
    assert (s["bsdf"] == f32(0.44444445)).all()


@pytest.mark.parametrize("mode", [0, 1])
def test_sample_zero_when_no_lobes(mode):  # :238
    s = orc.material_sample([0, 0, 0], [0, 0, 0], 0.0, 1.0, [0, 1, 0], [0, 1, 0], mode=mode, seed=1234)
    assert s["is_delta"] and s["pdf"] == 0.0 and (s["bsdf"] == 0).all() and (s["wo"] == np.array([0, 1, 0], f32)).all()


def ulps(a, b):
    a = np.asarray(a, f32).view(np.int32).astype(np.int64)
    b = np.asarray(b, f32).view(np.int32).astype(np.int64)
    return np.abs(a - b).max()


@pytest.mark.parametrize("mode", [0, 1])
def test_sample_specular_only(mode):  # :266 (assert_ulps_eq, default 4 ulps)
    s = orc.material_sample([0, 0, 0], [0.2, 0.4, 0.6], 0.0, 1.0, [0.8, 0.6, 0.0], [0, 1, 0], mode=mode, seed=1234)
    assert s["is_delta"]
    assert ulps(s["pdf"], 1.0) <= 4
    assert ulps(s["bsdf"], [0.208192, 0.406144, 0.604096]) <= 4
    assert ulps(s["wo"][:2], [-0.8, 0.6]) <= 4 and s["wo"][2] == 0.0


@pytest.mark.parametrize("mode", [0, 1])
def test_sample_diffuse_only(mode):  # :289
    albedo = np.array([0.2, 0.4, 0.6], f32)
    s = orc.material_sample(albedo, [0, 0, 0], 0.0, 1.0, [0, 1, 0], [0, 1, 0], mode=mode, seed=1)
    assert not s["is_delta"]
    cos_theta = max(s["wo"][1], 0.0)
    inv_pi = f32(0.31830987)
    assert ulps(s["pdf"], f32(cos_theta) * inv_pi) <= 4
    assert ulps(s["bsdf"], albedo * inv_pi) <= 4
    assert cos_theta > 0.0
    assert abs(np.linalg.norm(s["wo"].astype(np.float64)) - 1.0) < 1e-6


@pytest.mark.parametrize("mode", [0, 1])
def test_sample_mixed_lobes(mode):  # :314-370: p_specular == 0.5, both branches reachable by seed search
    albedo = [0.2, 0.34604772, 0.6]
    seen = set()
    for seed in range(64):
        s = orc.material_sample(albedo, [0.25] * 3, 0.0, 1.0, [0, 1, 0], [0, 1, 0], uv=(0.3, 0.7), mode=mode, seed=seed)
        if s["is_delta"]:
            seen.add("spec")
            assert 0.0 < s["pdf"] < 1.0
            assert ulps(s["bsdf"], [0.25] * 3) <= 4
            assert ulps(s["wo"], [0, 1, 0]) <= 4
        else:
            seen.add("diff")
            cos_theta = max(s["wo"][1], 0.0)
            assert 0.0 < s["pdf"] < cos_theta * 0.31830987
            assert ulps(s["bsdf"], (1 - f32(0.25)) * np.array(albedo, f32) * f32(0.31830987)) <= 4
    assert seen == {"spec", "diff"}


# ---------------------------------------------------------------- tracing/src/sampling.rs:55-104 (properties only)
@pytest.mark.parametrize("mode", [0, 1])
def test_sampling_properties(mode):
    sq = orc.sample_many(0, 1000, mode, 1)
    assert (sq >= 0).all() and (sq < 1).all()
    sp = orc.sample_many(1, 1000, mode, 1).astype(np.float64)
    n = np.linalg.norm(sp, axis=1)
    assert (n >= 0.9999998).all() and (n <= 1.0000002).all()
    dk = orc.sample_many(2, 1000, mode, 1).astype(np.float64)
    assert ((dk**2).sum(1) <= 1.0 + 1e-6).all()
    hs = orc.sample_many(3, 1000, mode, 1).astype(np.float64)
    assert (hs[:, 2] >= 0).all() and (hs[:, 2] <= 1).all()
    assert (np.abs((hs**2).sum(1) - 1.0) <= 2e-6).all()


def test_counter_sampler_is_uniform():
    u = orc.sample_many(0, 200000, 1, 7).astype(np.float64)
    assert abs(u.mean() - 0.5) < 3e-3
    assert abs(np.corrcoef(u[:, 0], u[:, 1])[0, 1]) < 1e-2
    hist = np.histogram(u[:, 0], bins=64, range=(0, 1))[0]
    assert hist.min() > 0.9 * 200000 / 64 and hist.max() < 1.1 * 200000 / 64


def test_deterministic_sincos_accuracy():
    x = np.linspace(-0.8, 6.3, 200001).astype(f32)
    s, c = orc.sincos(x)
    xs = x.astype(np.float64)
    assert np.abs(s - np.sin(xs)).max() < 2.5e-7
    assert np.abs(c - np.cos(xs)).max() < 2.5e-7


# ---------------------------------------------------------------- tracing/src/material/albedo.rs:36-46
def test_texture_repeat():
    tex = np.zeros((2, 2, 3), f32)
    for y in range(2):
        for x in range(2):
            tex[y, x] = [x / 2.0, y / 2.0, 0.0]
    a = orc.texture_get(tex, 0.2, 0.3)
    b = orc.texture_get(tex, 1.2, 1.3)
    assert ulps(a, b) <= 4
    assert orc.lib().orc_wrap01(1.0) == 0.0 and orc.lib().orc_wrap01(-0.25) == 0.75


# ---------------------------------------------------------------- wavefront/src/obj.rs:172-303
def test_obj_parser():
    assert orc.parse_obj("v 1 2 3")["vertices"] == [[1, 2, 3]]
    assert orc.parse_obj("vt 1 2")["texcoords"] == [[1, 2]]
    assert orc.parse_obj("vn 1 2 3")["normals"] == [[1, 2, 3]]
    assert orc.parse_obj("v 1. 2. 3.")["vertices"] == [[1, 2, 3]]
    assert orc.parse_obj("v -1. -2. -3.")["vertices"] == [[-1, -2, -3]]
    assert orc.parse_obj("usemtl m1\nf 1/2/3")["chunks"] == [{"material": "m1", "faces": [[[1, 2, 3]]]}]
    assert orc.parse_obj("usemtl m1\nf 1//3")["chunks"] == [{"material": "m1", "faces": [[[1, 0, 3]]]}]
    assert orc.parse_obj("usemtl m1\nf 1//")["chunks"] == [{"material": "m1", "faces": [[[1, 0, 0]]]}]
    assert orc.parse_obj("usemtl m1\nf 1/2/3 4/5/6 7/8/9")["chunks"][0]["faces"] == [[[1, 2, 3], [4, 5, 6], [7, 8, 9]]]
    assert orc.parse_obj("usemtl m1\nf 1/2/3 4/5/6 7/8/9 10/11/12")["chunks"][0]["faces"][0][3] == [10, 11, 12]
    assert orc.parse_obj("usemtl m1")["chunks"] == [{"material": "m1", "faces": []}]
    assert orc.parse_obj("mtllib test.mtl")["mtl_lib"] == "test.mtl"
    assert len(orc.parse_obj("# comment\nusemtl m1")["chunks"]) == 1
    for unsupported in ("g mesh1 model", "o todo", "s todo"):
        assert len(orc.parse_obj(unsupported)["chunks"]) == 0
    with pytest.raises(ValueError):
        orc.parse_obj("bogus line")


# ---------------------------------------------------------------- wavefront/src/mtl.rs:163-250
def test_mtl_parser():
    m = orc.parse_mtl
    assert len(m("newlight l1")["lights"]) == 1
    assert m("newlight l1\nlightposition 1. 2. 3.")["lights"][0]["position"] == [1, 2, 3]
    assert m("newlight l1\nlightcolor 1. 2. 3.")["lights"][0]["color"] == [1, 2, 3]
    assert m("newlight l1\nlightradius 1.")["lights"][0]["radius"] == 1
    assert m("newlight l1\nlightintensity 1.")["lights"][0]["intensity"] == 1
    assert len(m("newcamera c1")["cameras"]) == 1
    assert m("newcamera c1\ncameraposition 1. 2. 3.")["cameras"][0]["position"] == [1, 2, 3]
    assert m("newcamera c1\ncameratarget 1. 2. 3.")["cameras"][0]["target"] == [1, 2, 3]
    assert m("newcamera c1\ncameraup 1. 2. 3.")["cameras"][0]["up"] == [1, 2, 3]
    assert m("newcamera c1\ncamerafov 1.")["cameras"][0]["fov"] == 1
    assert m("newmtl m1")["materials"][0]["name"] == "m1"
    assert m("newmtl m1\nKd 1. 2. 3.")["materials"][0]["diffuse_reflection"] == [1, 2, 3]
    assert m("newmtl m1\nmap_Kd file.png")["materials"][0]["diffuse_map"] == "file.png"
    assert m("newmtl m1\nKs 1. 2. 3.")["materials"][0]["specular_reflection"] == [1, 2, 3]
    assert m("newmtl m1\nreflat0deg 0.5")["materials"][0]["reflection_0_degrees"] == 0.5
    assert m("newmtl m1\nreflat90deg 0.5")["materials"][0]["reflection_90_degrees"] == 0.5
    assert m("newmtl m1\nNi 0.5")["materials"][0]["index_of_refraction"] == 0.5
    assert m("newmtl m1\nd 1.0")["materials"][0]["transparency"] == 0.0
    assert m("newmtl m1\nTr 0.5")["materials"][0]["transparency"] == 0.5
    assert len(m("newmtl m1\nspecularroughness 1.")["materials"]) == 1
    assert len(m("# comment\nnewmtl m1")["materials"]) == 1
    d = m("newmtl m1")["materials"][0]  # defaults, mtl.rs:22-35
    assert d["diffuse_reflection"] == [pytest.approx(0.7)] * 3 and d["index_of_refraction"] == 1.0 and d["metalness"] == 0.0


# ---------------------------------------------------------------- tracing/src/image_buffer.rs:29-38
def test_to_rgb8():
    acc = np.array([[0.0, 1.0, 4.0], [0.5, 2.0, -1.0]], f32)
    out = orc.to_rgb8(acc, 2)
    expect = lambda p: int(np.round(min((p / 2.0) ** (1 / 2.2), 1.0) * 255))
    assert list(out[0]) == [0, expect(1.0), 255]
    assert list(out[1][:2]) == [expect(0.5), 255]


# ---------------------------------------------------------------- kdtree-tester-cli/src/reducer.rs:104-117
# The one real-scene vector the reference holds: a shadow ray on sr2.obj (towards the light at (-10, 10, 10),
# t in [0, 1] as ray_bouncer.rs:84-95 shoots it) with the brute-force ("reference") answer 7589 and the answer
# 5556 of a kd-tree bug of the past.  Both records are Möller–Trumbore results on real triangles, so they pin
# the OBJ face order -> triangle index mapping and the (t, u, v) arithmetic against the reference itself.
SR2_RAY = ([3.897963, 0.24242611, -4.203691], [-13.897963, 9.757574, 14.2036915])
SR2_REFERENCE = (7589, 0.0004729527, 0.09395919, 0.47453666)
SR2_PAST_BUG = (5556, 0.05429069, 0.2189177, 0.74337834)


def sr2_path():
    import os

    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes", "sr2.obj")


def test_reference_embedded_sr2_ray():
    scene = orc.Scene(sr2_path())
    tris = scene.export()["triangles"].reshape(-1, 3, 3)
    ray = (np.array(SR2_RAY[0], f32), np.array(SR2_RAY[1], f32))
    for index, t, u, v in (SR2_REFERENCE, SR2_PAST_BUG):  # each triangle on its own, both intersection routes
        assert orc.triangle_intersect(tris[index], ray) == (f32(t), f32(u), f32(v))
        assert orc.triangle_intersect(tris[index], ray, as_any=True) == (f32(t), f32(u), f32(v))
    tree = scene.tree()
    for tmax in (1.0, FMAX):  # the tester's shadow range and the extend range
        rays = orc.make_rays([SR2_RAY[0]], [SR2_RAY[1]], 0.0, tmax)
        for hit in (tree.brute(rays)[0], tree.intersect(rays)[0]):  # brute force, then the kd-tree: the bug is gone
            assert (int(hit["index"]), f32(hit["t"]), f32(hit["u"]), f32(hit["v"])) == (SR2_REFERENCE[0],) + tuple(map(f32, SR2_REFERENCE[1:]))
