"""ctypes view of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
package (pathtracer_amd/) never does.  Layouts (ray 32 B, hit 16 B, node 8 B) are those of include/ptx.h.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")

RAY_DTYPE = np.dtype([("o", "<f4", 3), ("tmin", "<f4"), ("d", "<f4", 3), ("tmax", "<f4")])
HIT_DTYPE = np.dtype([("index", "<u4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])
NODE_DTYPE = np.dtype([("w0", "<u4"), ("w1", "<u4")])
FMAX = np.finfo(np.float32).max
MISS = 0xFFFFFFFF


def build(force=False):
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", HERE, "-s"] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.orc_last_error.restype = C.c_char_p
        for name in ("orc_tree_build", "orc_tree_from_flat", "orc_scene_load", "orc_scene_tree"):
            getattr(L, name).restype = C.c_void_p
        L.orc_render.restype = C.c_double
        L.orc_bounce_rays.restype = C.c_size_t
        L.orc_render_raylog.restype = C.c_size_t
        L.orc_wrap01.restype = C.c_float
        L.orc_wrap01.argtypes = [C.c_float]
        _lib = L
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def triangle_intersect(tri, ray, as_any=False):
    """tri: 3x3, ray: (origin, direction).  Returns (t,u,v) float32 or None."""
    out = np.zeros(3, np.float32)
    r = _f(np.concatenate([ray[0], ray[1]]))
    hit = lib().orc_triangle_intersect(_p(_f(tri).reshape(-1)), int(as_any), _p(r), _p(out))
    return tuple(out) if hit else None


def axial_axis(tri):
    return lib().orc_axial_axis(_p(_f(tri).reshape(-1)))


def axial_intersect_point(axis, dist, v, p):
    out = np.zeros(2, np.float32)
    hit = lib().orc_axial_intersect_point(axis, C.c_float(dist), _p(_f(v).reshape(-1)), _p(_f(p)), _p(out))
    return tuple(out) if hit else None


def aap_intersect(axis, dist, ray):
    t = C.c_float()
    r = _f(np.concatenate([ray[0], ray[1]]))
    hit = lib().orc_aap_intersect(axis, C.c_float(dist), _p(r), C.byref(t))
    return np.float32(t.value) if hit else None


def clip_points(tri, aabb):
    out = np.zeros((16, 3), np.float32)
    n = lib().orc_clip_points(_p(_f(tri).reshape(-1)), _p(_f(aabb).reshape(-1)), _p(out), 16)
    return out[:n]


def clip_aabb(tri, aabb):
    out = np.zeros(6, np.float32)
    ok = lib().orc_clip_aabb(_p(_f(tri).reshape(-1)), _p(_f(aabb).reshape(-1)), _p(out))
    return out.reshape(2, 3) if ok else None


def bounding_box(tris):
    t = _f(tris).reshape(-1, 9)
    out = np.zeros(6, np.float32)
    lib().orc_bounding_box(_p(t), len(t), _p(out))
    return out.reshape(2, 3)


def event_list(tris, axis):
    t = _f(tris).reshape(-1, 9)
    out = np.zeros((4 * len(t) + 4, 3), np.uint32)
    n = lib().orc_event_list(_p(t), len(t), axis, _p(out), len(out))
    kinds = {0: "end", 1: "planar", 2: "start"}
    return [(kinds[int(k)], int(i), float(np.uint32(d).view(np.float32))) for i, d, k in out[:n]]


def update_geometry_side(tris, axis, dist, side_right):
    t = _f(tris).reshape(-1, 9)
    out = np.zeros(len(t), np.uint8)
    lib().orc_update_geometry_side(_p(t), len(t), axis, C.c_float(dist), int(side_right), _p(out))
    return [["both", "left", "right"][i] for i in out]


def make_rays(origins, directions, tmin=0.0, tmax=FMAX):
    o = _f(origins).reshape(-1, 3)
    r = np.zeros(len(o), RAY_DTYPE)
    r["o"] = o
    r["d"] = _f(directions).reshape(-1, 3)
    r["tmin"] = tmin
    r["tmax"] = tmax
    return r


class Tree:
    """A pointer KdNode tree plus its triangles."""

    def __init__(self, handle, owner=True):
        if not handle:
            raise RuntimeError(lib().orc_last_error().decode())
        self.h = C.c_void_p(handle)

    @classmethod
    def build(cls, tris, sah=(1.0, 1.5, 0.8)):
        t = _f(tris).reshape(-1, 9)
        return cls(lib().orc_tree_build(_p(t), len(t), _p(_f(sah))))

    @classmethod
    def from_flat(cls, tris, nodes, leaf_idx, raw_triangles=False):
        t = _f(tris).reshape(-1, 9)
        nodes = np.ascontiguousarray(nodes, NODE_DTYPE)
        leaf_idx = np.ascontiguousarray(leaf_idx, np.uint32)
        return cls(lib().orc_tree_from_flat(_p(t), len(t), int(raw_triangles), _p(nodes), len(nodes), _p(leaf_idx)))

    def __del__(self):
        try:
            lib().orc_tree_free(self.h)
        except Exception:
            pass

    def string(self):
        n = 1 << 16
        while True:
            buf = C.create_string_buffer(n)
            r = lib().orc_tree_string(self.h, buf, C.c_size_t(n))
            if r >= 0:
                return buf.value.decode()
            n = -r + 16

    def stats(self):
        out = np.zeros(6, np.uint64)
        lib().orc_tree_stats(self.h, _p(out))
        return dict(zip(["nodes", "leaves", "empty_leaves", "leaf_refs", "max_depth", "max_leaf"], map(int, out)))

    def to_flat(self):
        nn, nl = C.c_uint32(), C.c_uint32()
        lib().orc_tree_to_flat(self.h, None, C.byref(nn), None, C.byref(nl))
        nodes = np.zeros(nn.value, NODE_DTYPE)
        leaf = np.zeros(max(nl.value, 1), np.uint32)
        lib().orc_tree_to_flat(self.h, _p(nodes), C.byref(nn), _p(leaf), C.byref(nl))
        return nodes, leaf[: nl.value]

    def intersect(self, rays, threads=1, counters=False):
        rays = np.ascontiguousarray(rays, RAY_DTYPE)
        hits = np.zeros(len(rays), HIT_DTYPE)
        ctr = np.zeros(3, np.uint64)
        lib().orc_tree_intersect(self.h, _p(rays), C.c_size_t(len(rays)), _p(hits), _p(ctr) if counters else None, threads)
        if counters:
            return hits, dict(inner_nodes=int(ctr[0]), leaves=int(ctr[1]), tri_tests=int(ctr[2]))
        return hits

    def brute(self, rays, threads=1):
        rays = np.ascontiguousarray(rays, RAY_DTYPE)
        hits = np.zeros(len(rays), HIT_DTYPE)
        lib().orc_brute_intersect(self.h, _p(rays), C.c_size_t(len(rays)), _p(hits), threads)
        return hits


def flat_leaf(indices, leaf_pool):
    """helper for hand-written trees: returns a node spec consumed by flat_tree()."""
    return ("leaf", list(indices))


def flat_tree(spec):
    """spec: ("leaf",[idx..]) | ("node", axis, distance, left_spec, right_spec) -> (nodes, leaf_idx) preorder."""
    nodes, leaf = [], []

    def rec(s):
        me = len(nodes)
        nodes.append([0, 0])
        if s[0] == "leaf":
            nodes[me] = [len(leaf), (len(s[1]) << 2) | 3]
            leaf.extend(s[1])
        else:
            _, axis, dist, l, r = s
            rec(l)
            right = len(nodes)
            nodes[me] = [int(np.float32(dist).view(np.uint32)), (right << 2) | axis]
            rec(r)

    rec(spec)
    n = np.zeros(len(nodes), NODE_DTYPE)
    for i, (a, b) in enumerate(nodes):
        n[i] = (a, b)
    return n, np.array(leaf, np.uint32)


def tree_spec_string(spec):
    """Canonical string of a hand-written tree spec, same format as Tree.string()."""
    if spec[0] == "leaf":
        return "[" + ",".join(str(i) for i in spec[1]) + "]"
    _, axis, dist, l, r = spec
    return "N(%s,%08x,%s,%s)" % ("XYZ"[axis], int(np.float32(dist).view(np.uint32)), tree_spec_string(l), tree_spec_string(r))


def sample_refraction(wi, n, ior, diffuse, td, probability):
    out = np.zeros(8, np.float32)
    lib().orc_sample_refraction(_p(_f(wi)), _p(_f(n)), C.c_float(ior), _p(_f(diffuse)), _p(_f(td)), C.c_float(probability), _p(out))
    return dict(is_delta=bool(out[0]), pdf=out[1], bsdf=out[2:5].copy(), wo=out[5:8].copy())


def material_sample(color, f0, transmission, ior, wi, n, uv=(0, 0), mode=0, seed=0, dim=2):
    out = np.zeros(8, np.float32)
    mat = _f(list(color) + list(f0) + [transmission, ior])
    lib().orc_material_sample(_p(mat), _p(_f(wi)), _p(_f(n)), _p(_f(uv)), mode, C.c_uint64(seed), dim, _p(out))
    return dict(is_delta=bool(out[0]), pdf=out[1], bsdf=out[2:5].copy(), wo=out[5:8].copy())


def material_from_mtl(kd, ks, ni, pm, tr):
    out = np.zeros(8, np.float32)
    lib().orc_material_from_mtl(_p(_f(kd)), _p(_f(ks)), C.c_float(ni), C.c_float(pm), C.c_float(tr), _p(out))
    return out


def sample_many(which, count, mode=0, seed=1):
    width = {0: 2, 1: 3, 2: 2, 3: 3}[which]
    out = np.zeros((count, width), np.float32)
    lib().orc_sample_many(which, mode, C.c_uint64(seed), count, _p(out))
    return out


def texture_get(rgb, u, v):
    rgb = _f(rgb)
    out = np.zeros(3, np.float32)
    lib().orc_texture_get(_p(rgb), rgb.shape[1], rgb.shape[0], C.c_float(u), C.c_float(v), _p(out))
    return out


def sincos(x):
    x = _f(x)
    s, c = np.zeros_like(x), np.zeros_like(x)
    lib().orc_sincos(_p(x), len(x), _p(s), _p(c))
    return s, c


def to_rgb8(accum, iterations):
    a = _f(accum)
    out = np.zeros(a.shape, np.uint8)
    lib().orc_to_rgb8(_p(a), C.c_size_t(a.size // 3), C.c_uint16(iterations & 0xFFFF), _p(out))
    return out


def _json_call(fn, text):
    n = 1 << 20
    while True:
        buf = C.create_string_buffer(n)
        r = fn(text.encode(), buf, C.c_size_t(n))
        if r == -1:
            raise ValueError(lib().orc_last_error().decode())
        if r >= 0:
            return json.loads(buf.value.decode())
        n = -r + 16


def parse_obj(text):
    return _json_call(lib().orc_parse_obj_json, text)


def parse_mtl(text):
    return _json_call(lib().orc_parse_mtl_json, text)


def srgb_to_linear_u8():
    """sRGB EOTF for 8-bit texels (image crate convert_color_space to SRGB_LINEAR; parity unpinned)."""
    c = np.arange(256, dtype=np.float64) / 255.0
    lin = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)
    return lin.astype(np.float32)


def load_texture_linear(path):
    from PIL import Image

    im = np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)
    return srgb_to_linear_u8()[im]


class Scene:
    """setup_scene of pathtracer-cli (main.rs:123-159) on the CPU restatement."""

    def __init__(self, obj_path, sah=(1.0, 1.5, 0.8), max_bounces=10, load_textures=True):
        h = lib().orc_scene_load(obj_path.encode(), _p(_f(sah)), max_bounces)
        if not h:
            raise RuntimeError(lib().orc_last_error().decode())
        self.h = C.c_void_p(h)
        self.path = obj_path
        c = np.zeros(8, np.uint32)
        lib().orc_scene_counts(self.h, _p(c))
        self.counts = dict(zip(["triangles", "materials", "lights", "cameras", "vertices", "normals", "texcoords", "chunks"], map(int, c)))
        self.textures = {}
        if load_textures:
            for m in range(self.counts["materials"]):
                buf = C.create_string_buffer(1024)
                lib().orc_scene_texture_name(self.h, m, buf, C.c_size_t(1024))
                name = buf.value.decode()
                if name:
                    tex = load_texture_linear(os.path.join(os.path.dirname(obj_path), name))
                    self.set_texture(m, tex)

    def __del__(self):
        try:
            lib().orc_scene_free(self.h)
        except Exception:
            pass

    def set_texture(self, material, rgb):
        rgb = _f(rgb)
        self.textures[material] = rgb
        lib().orc_scene_set_texture(self.h, material, rgb.shape[1], rgb.shape[0], _p(rgb))

    def export(self):
        n = self.counts["triangles"]
        tris = np.zeros((n, 9), np.float32)
        normals = np.zeros((n, 9), np.float32)
        uvs = np.zeros((n, 6), np.float32)
        mats = np.zeros(n, np.uint32)
        lib().orc_scene_export(self.h, _p(tris), _p(normals), _p(uvs), _p(mats))
        materials = np.zeros((self.counts["materials"], 8), np.float32)
        lights = np.zeros((max(self.counts["lights"], 1), 7), np.float32)
        camera = np.zeros(10, np.float32)
        lib().orc_scene_export_shading(self.h, _p(materials), _p(lights), _p(camera))
        return dict(triangles=tris, normals=normals, texcoords=uvs, material_ids=mats, materials=materials,
                    lights=lights[: self.counts["lights"]], camera=camera)

    def tree(self):
        return Tree(lib().orc_scene_tree(self.h))

    def pinhole(self, w, h):
        out = np.zeros(12, np.float32)
        lib().orc_scene_pinhole(self.h, w, h, _p(out))
        return out.reshape(4, 3)

    def render(self, w, h, threads=1, iterations_per_thread=1, mode=1, seed=0, first_sample=0, accum=None, stats=False):
        """render_parallel_iterations; returns (accum, seconds[, stats])."""
        if accum is None:
            accum = np.zeros((h, w, 3), np.float32)
        st = np.zeros(8, np.uint64)
        secs = lib().orc_render(self.h, w, h, threads, iterations_per_thread, mode, C.c_uint64(seed), first_sample, _p(accum), _p(st) if stats else None)
        if stats:
            keys = ["extend_rays", "shadow_rays", "extend_inner_nodes", "extend_tri_tests", "shadow_inner_nodes", "shadow_tri_tests", "extend_leaves", "shadow_leaves"]
            return accum, secs, dict(zip(keys, map(int, st)))
        return accum, secs

    def render_raylog(self, w, h, seed=0, first_sample=0, count=1):
        """Bytes of the raylog file one reference thread writes for these iterations (raylogger.rs:54-66)."""
        n = lib().orc_render_raylog(self.h, w, h, C.c_uint64(seed), first_sample, count, None, C.c_size_t(0))
        raw = np.zeros(n * 33, np.uint8)
        lib().orc_render_raylog(self.h, w, h, C.c_uint64(seed), first_sample, count, _p(raw), C.c_size_t(n))
        return raw

    def bounce_rays(self, w, h, bounces=10, seed=0, threads=8, check=True):
        """kdtree-tester-cli's ray set; check=False skips the brute-force side (same rays whenever the checked run
        finds no fail: see orc_bounce_rays2)."""
        fails = C.c_uint64()
        cap = w * h * bounces * (1 + max(self.counts["lights"], 1))
        rays = np.zeros(cap, RAY_DTYPE)
        lib().orc_bounce_rays2.restype = C.c_size_t
        n = lib().orc_bounce_rays2(self.h, w, h, bounces, C.c_uint64(seed), _p(rays), C.c_size_t(cap), C.byref(fails), threads, int(bool(check)))
        return rays[:n].copy(), int(fails.value)


# ---- kdtree/src/format.rs and kdtree-cli statistics, restated over the oracle's own tree -------
# Formatting follows Rust's float printing: `{}` = shortest round-trip digits, positional; `{:?}` = the same
# with at least ".0", scientific outside [1e-4, 1e16) (core::fmt::float).  numpy's Dragon4 "unique" mode gives
# the same shortest digits independently of the product's std::to_chars.  Parity unpinned: format.rs has no
# tests of its own; the two implementations pin each other and the hand-written cases in tests/.
def rust_display_f32(x):
    x = np.float32(x)
    if np.isnan(x):
        return "NaN"
    if np.isinf(x):
        return "-inf" if x < 0 else "inf"
    return np.format_float_positional(x, unique=True, trim="-")


def rust_debug_f32(x):
    x = np.float32(x)
    if np.isnan(x) or np.isinf(x):
        return rust_display_f32(x)
    a = abs(float(x))
    if a != 0.0 and (a < float(np.float32(1e-4)) or a >= 1e16):
        return np.format_float_scientific(x, unique=True, trim="-", exp_digits=1).replace("e+", "e")
    return np.format_float_positional(x, unique=True, trim="0")


def any_triangle_arrays(tri9):
    """`AnyTriangle::from` then `as_arrays()` (any_triangle.rs:9-16, triangle.rs:59-77, axial_triangle.rs:43-49)."""
    t = np.array(tri9, np.float32).reshape(3, 3)
    for axis in range(3):
        if t[0, axis] == t[1, axis] and t[1, axis] == t[2, axis]:
            t[1, axis] = t[2, axis] = t[0, axis]
            break
    return t


def tree_json(triangles, nodes, leaves):
    """`write_tree_json` (format.rs:116-130) over the flattened arrays of Tree.to_flat()."""
    tri_txt = ", ".join("[" + ", ".join("[" + ", ".join(rust_debug_f32(c) for c in v) + "]" for v in any_triangle_arrays(t)) + "]"
                        for t in np.asarray(triangles, np.float32).reshape(-1, 9))
    w0, w1 = nodes["w0"], nodes["w1"]
    out = []
    todo = [("node", 0)]  # explicit stack: emit text pieces in preorder
    while todo:
        kind, i = todo.pop()
        if kind == "text":
            out.append(i)
            continue
        if w1[i] & 3 == 3:
            cnt, off = int(w1[i] >> 2), int(w0[i])
            out.append("[" + ", ".join(str(int(k)) for k in leaves[off:off + cnt]) + "]")
            continue
        dist = np.array([w0[i]], np.uint32).view(np.float32)[0]
        out.append('{"axis": "%s", "distance": %s, "left": ' % ("XYZ"[int(w1[i] & 3)], rust_display_f32(dist)))
        todo += [("text", "}"), ("node", int(w1[i] >> 2)), ("text", ', "right": '), ("node", i + 1)]
    return '{"triangles": [' + tri_txt + '], "root": ' + "".join(out) + "}\n"


def tree_pretty(nodes, leaves):
    """`write_node_pretty` (format.rs:22-47): right subtree is printed before the left one (LIFO)."""
    w0, w1 = nodes["w0"], nodes["w1"]
    out, stack = [], [(0, 0)]
    while stack:
        depth, i = stack.pop()
        if w1[i] & 3 == 3:
            cnt, off = int(w1[i] >> 2), int(w0[i])
            out.append("  " * depth + "Leaf [" + ", ".join(str(int(k)) for k in leaves[off:off + cnt]) + "]\n")
        else:
            stack.append((depth + 1, i + 1))
            stack.append((depth + 1, int(w1[i] >> 2)))
            dist = np.array([w0[i]], np.uint32).view(np.float32)[0]
            out.append("  " * depth + "Split %s %s\n" % ("XYZ"[int(w1[i] & 3)], rust_display_f32(dist)))
    return "".join(out)


def tree_statistics(triangles, nodes, leaves, sah=(1.0, 1.5, 0.8)):
    """`statistics` (kdtree-cli/src/main.rs:142-159), `Statistics::compute` (:113-133), `tree_cost` (:45-103)."""
    f = np.float32
    w0, w1 = nodes["w0"], nodes["w1"]
    tris = np.stack([any_triangle_arrays(t) for t in np.asarray(triangles, np.float32).reshape(-1, 9)]) if len(triangles) else np.zeros((0, 3, 3), np.float32)
    lo, hi = (tris.min((0, 1)), tris.max((0, 1))) if len(tris) else (np.zeros(3, f), np.zeros(3, f))

    def area(a, b):
        s = (b - a).astype(f)
        return f(2.0) * f(f(f(s[0] * s[1]) + f(s[0] * s[2])) + f(s[1] * s[2]))

    ct, ci, ef = (f(x) for x in sah)
    scene_area = area(lo, hi)
    is_empty = lambda i: (w1[i] & 3 == 3) and (w1[i] >> 2) == 0

    def cost(i, a, b):
        if w1[i] & 3 == 3:
            return f(f(f(ci * f(int(w1[i] >> 2))) * area(a, b)) / scene_area)
        split_cost = f(area(a, b) / scene_area)
        axis, dist = int(w1[i] & 3), np.array([w0[i]], np.uint32).view(f)[0]
        l, r = i + 1, int(w1[i] >> 2)
        lb, ra = b.copy(), a.copy()
        lb[axis] = dist
        ra[axis] = dist
        lc, rc = cost(l, a, lb), cost(r, ra, b)
        node = f(f(f(ct + split_cost) + lc) + rc)
        return f((ef if (is_empty(l) or is_empty(r)) else f(1.0)) * node)

    import sys
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 10000))
    depths, sizes, n_nodes, stack = [], [], 0, [(1, 0)]
    while stack:
        d, i = stack.pop()
        n_nodes += 1
        if w1[i] & 3 == 3:
            depths.append(d)
            sizes.append(int(w1[i] >> 2))
        else:
            stack += [(d + 1, i + 1), (d + 1, int(w1[i] >> 2))]

    def compute(v):
        v = sorted(v)
        n = len(v)
        median = f(v[0]) if n == 1 else (f(v[n // 2]) if n % 2 == 0 else f(f(v[n // 2] + v[n // 2 + 1]) / f(2.0)))
        total = f(0.0)
        for x in v:
            total = f(total + f(x))
        return dict(min=v[0], max=v[-1], total=sum(v), median=median, mean=f(total / f(n)))

    return dict(sah_cost=cost(0, lo.astype(f), hi.astype(f)), geometries=len(tris), node_count=n_nodes, leaf_count=len(depths),
                leaf_depth=compute(depths), leaf_geometries=compute(sizes))
