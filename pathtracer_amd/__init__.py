"""pathtracer_amd — host-side mirror of daoo/pathtracer's render seams over libptx_mi355x.so (gfx950).

The reference is Rust; no Rust toolchain exists in the build image, so the host layer above the C ABI
(include/ptx.h) is this thin ctypes binding plus the C++ CLIs under pathtracer_amd/cli.  Names follow the
reference: `SahCost`, `build_kdtree`, `KdTree.intersect` (kdtree/src/lib.rs:73-78, build.rs:33),
`Pinhole` (tracing/src/camera.rs:61-93), `Pathtracer.render` / `render_parallel_iterations`
(tracing/src/pathtracer.rs:131-141, worker.rs:84-119), `ImageBuffer.to_rgb8` (image_buffer.rs:29-38).

There is NO CPU fallback: every device call fails loudly when the HIP library or a GPU is missing.
This package never imports anything from oracle/.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PTX_LIB") or os.path.join(_HERE, "lib", "libptx_mi355x.so")

RAY_DTYPE = np.dtype([("o", "<f4", 3), ("tmin", "<f4"), ("d", "<f4", 3), ("tmax", "<f4")])
HIT_DTYPE = np.dtype([("index", "<u4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])
NODE_DTYPE = np.dtype([("w0", "<u4"), ("w1", "<u4")])
MISS = 0xFFFFFFFF
FMAX = float(np.finfo(np.float32).max)
MAX_LIGHTS = 8
FLAG_COUNT_WORK = 1
FLAG_TIME_KERNELS = 2
FLAG_CHECK_BRUTE = 4
FLAG_MATERIAL_QUEUES = 8
FLAG_NO_BOX_CLIP = 16


class PtxError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("ptx error %d: %s" % (code, text))
        self.code = code


class SahCost(C.Structure):
    """kdtree/src/sah.rs:52-56; defaults :122-130."""

    _fields_ = [("traverse_cost", C.c_float), ("intersect_cost", C.c_float), ("empty_factor", C.c_float)]

    def __init__(self, traverse_cost=1.0, intersect_cost=1.5, empty_factor=0.8):
        super().__init__(traverse_cost, intersect_cost, empty_factor)


class _Material(C.Structure):
    _fields_ = [("albedo", C.c_float * 3), ("texture", C.c_int32), ("schlick_f0", C.c_float * 3),
                ("transmission", C.c_float), ("ior", C.c_float)]


class _Texture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgb", C.POINTER(C.c_float))]


class _Light(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("intensity", C.c_float * 3), ("radius", C.c_float)]


class Pinhole(C.Structure):
    """tracing/src/camera.rs:61-68."""

    _fields_ = [("position", C.c_float * 3), ("plane", C.c_float * 3), ("dx", C.c_float * 3),
                ("dy", C.c_float * 3), ("width", C.c_uint32), ("height", C.c_uint32)]

    @classmethod
    def new(cls, position, target, up, fov_degrees, width, height):
        """Pinhole::new(Camera::new(position, target, up, fov), size)."""
        out = cls()
        f3 = C.c_float * 3
        _check(lib().ptx_pinhole_make(f3(*position), f3(*target), f3(*up), C.c_float(fov_degrees), width, height, C.byref(out)))
        return out


class _SceneDesc(C.Structure):
    _fields_ = [
        ("triangles", C.c_void_p), ("n_triangles", C.c_uint32),
        ("nodes", C.c_void_p), ("n_nodes", C.c_uint32),
        ("leaf_indices", C.c_void_p), ("n_leaf_indices", C.c_uint32),
        ("normals", C.c_void_p), ("texcoords", C.c_void_p), ("material_ids", C.c_void_p),
        ("materials", C.POINTER(_Material)), ("n_materials", C.c_uint32),
        ("textures", C.POINTER(_Texture)), ("n_textures", C.c_uint32),
        ("lights", C.POINTER(_Light)), ("n_lights", C.c_uint32),
        ("environment", C.c_float * 3), ("max_bounces", C.c_uint32),
    ]


class Counters(C.Structure):
    _fields_ = [("extend_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("extend_inner_nodes", C.c_uint64), ("extend_tri_tests", C.c_uint64),
                ("shadow_inner_nodes", C.c_uint64), ("shadow_tri_tests", C.c_uint64),
                ("samples", C.c_uint64), ("extend_launches", C.c_uint64), ("shadow_launches", C.c_uint64),
                ("extend_ms", C.c_double), ("shadow_ms", C.c_double), ("shade_ms", C.c_double),
                ("generate_ms", C.c_double), ("total_ms", C.c_double), ("other_ms", C.c_double), ("culled_camera_rays", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def lib():
    """The HIP library; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PtxError(-2, "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(or make -C pathtracer_amd/csrc); there is no CPU fallback" % LIB_PATH)
        # PyTorch-ROCm ships its own libamdhip64.so.7; whichever HIP runtime is mapped first serves the
        # whole process, and torch cannot find a GPU once the system runtime is in.  Torch (used for
        # device tensors, streams and RCCL) therefore goes first when it is installed.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.ptx_last_error.restype = C.c_char_p
        L.ptx_version.restype = C.c_char_p
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise PtxError(rc, lib().ptx_last_error().decode(errors="replace"))


def _ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None and a.size else C.c_void_p(None)


def version():
    return lib().ptx_version().decode()


def make_rays(origins, directions, tmin=0.0, tmax=FMAX):
    o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
    r = np.zeros(len(o), RAY_DTYPE)
    r["o"] = o
    r["d"] = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
    r["tmin"] = tmin
    r["tmax"] = tmax
    return r


# tracing/src/raylogger.rs:54-66 (33 bytes, packed)
RAYLOG_DTYPE = np.dtype([("iteration", "<u2"), ("px", "<u2"), ("py", "<u2"), ("bounces", "u1"), ("shadow", "u1"), ("intersect", "u1"),
                         ("o", "<f4", 3), ("d", "<f4", 3)])
FAIL_DTYPE = np.dtype([("i", "<u2"), ("o", "<f4", 3), ("d", "<f4", 3), ("correct", "<f4", 3), ("actual", "<f4", 3)])  # pybits/visualize_fails.py


class KdTree:
    """Flattened `KdNode` (8-byte preorder nodes + leaf index list) with its triangles."""

    def __init__(self, triangles, nodes, leaf_indices):
        self.triangles = np.ascontiguousarray(triangles, np.float32).reshape(-1, 9)
        self.nodes = np.ascontiguousarray(nodes, NODE_DTYPE)
        self.leaf_indices = np.ascontiguousarray(leaf_indices, np.uint32)

    def intersect_host(self, origin, direction, t_range=(0.0, FMAX)):
        """`KdNode::intersect` for one ray on the CPU (scalar form of the same traversal source)."""
        ray = make_rays([origin], [direction], t_range[0], t_range[1])
        hit = np.zeros(1, HIT_DTYPE)
        _check(lib().ptx_kdtree_intersect_host(_ptr(self.triangles), len(self.triangles), _ptr(self.nodes), len(self.nodes),
                                              _ptr(self.leaf_indices), _ptr(ray), _ptr(hit)))
        return hit[0]

    def upload(self, flags=0):
        """Intersect-only device scene (the accel seam without shading data)."""
        return DeviceScene.from_arrays(self.triangles, self.nodes, self.leaf_indices, flags=flags)

    def device_layout(self, top_nodes=1023):
        """What `upload()` lays down for the kernels (host computation, no GPU): (inner nodes as an (n, 4) uint32 array
        {plane bits, axis, left descriptor, right descriptor}, leaf records as triangle index | last-of-run << 31, root
        descriptor, size of the breadth-first prefix kept in LDS)."""
        words = np.zeros((max(len(self.nodes), 1), 4), np.uint32)
        records = np.zeros(max(len(self.leaf_indices), 1), np.uint32)
        counts = (C.c_uint32 * 4)()
        _check(lib().ptx_debug_device_tree(_ptr(self.nodes), len(self.nodes), _ptr(self.leaf_indices), len(self.leaf_indices),
                                          len(self.triangles), int(top_nodes), _ptr(words), _ptr(records), counts))
        return words[:counts[0]].copy(), records[:counts[1]].copy(), int(np.int32(np.uint32(counts[2]))), int(counts[3])

    def _text(self, fn, *head):
        out, n = C.c_void_p(), C.c_size_t()
        _check(fn(*head, _ptr(self.nodes), len(self.nodes), _ptr(self.leaf_indices), len(self.leaf_indices), C.byref(out), C.byref(n)))
        try:
            return C.string_at(out, n.value).decode()
        finally:
            lib().ptx_free(out)

    def to_json(self):
        """`write_tree_json` (kdtree/src/format.rs:116-130): what `kdtree-cli -j` prints."""
        return self._text(lib().ptx_kdtree_write_json, _ptr(self.triangles), len(self.triangles))

    def pretty(self):
        """`write_node_pretty` (kdtree/src/format.rs:22-47): kdtree-cli's default dump."""
        return self._text(lib().ptx_kdtree_write_pretty)

    def stats(self, sah=None):
        """`statistics` + `tree_cost` of kdtree-cli (kdtree-cli/src/main.rs:45-159)."""
        sah = sah or SahCost()
        st = TreeStats()
        _check(lib().ptx_kdtree_stats(_ptr(self.triangles), len(self.triangles), _ptr(self.nodes), len(self.nodes),
                                      _ptr(self.leaf_indices), len(self.leaf_indices), C.byref(sah), C.byref(st)))
        return {k: getattr(st, k) for k, _ in TreeStats._fields_}

    @classmethod
    def from_json(cls, text):
        """Reads the reference's JSON tree dump into the flattened layout."""
        raw = text.encode() if isinstance(text, str) else bytes(text)
        h, tp, nt = C.c_void_p(), C.c_void_p(), C.c_uint32()
        _check(lib().ptx_kdtree_read_json(raw, C.c_size_t(len(raw)), C.byref(h), C.byref(tp), C.byref(nt)))
        try:
            tris = np.frombuffer(C.string_at(tp, nt.value * 36), np.float32).copy().reshape(-1, 9) if nt.value else np.zeros((0, 9), np.float32)
            nodes, leaf = _view_tree(h)
        finally:
            lib().ptx_kdtree_free(h)
            lib().ptx_free(tp)
        return cls(tris, nodes, leaf)


class TreeStats(C.Structure):
    _fields_ = [("sah_cost", C.c_float), ("geometries", C.c_uint64), ("node_count", C.c_uint64), ("leaf_count", C.c_uint64),
                ("leaf_depth_min", C.c_uint64), ("leaf_depth_max", C.c_uint64), ("leaf_depth_total", C.c_uint64),
                ("leaf_depth_mean", C.c_float), ("leaf_depth_median", C.c_float),
                ("leaf_geometries_min", C.c_uint64), ("leaf_geometries_max", C.c_uint64), ("leaf_geometries_total", C.c_uint64),
                ("leaf_geometries_mean", C.c_float), ("leaf_geometries_median", C.c_float)]


def _view_tree(h):
    pn, pl = C.c_void_p(), C.c_void_p()
    nn, nl = C.c_uint32(), C.c_uint32()
    _check(lib().ptx_kdtree_view(h, C.byref(pn), C.byref(nn), C.byref(pl), C.byref(nl)))
    nodes = np.frombuffer(C.string_at(pn, nn.value * 8), NODE_DTYPE).copy()
    leaf = np.frombuffer(C.string_at(pl, nl.value * 4), np.uint32).copy() if nl.value else np.zeros(0, np.uint32)
    return nodes, leaf


def build_kdtree(triangles, sah=None):
    """`build_kdtree(&[impl Geometry], &SahCost) -> KdNode` (kdtree/src/build.rs:33-41) on the host."""
    t = np.ascontiguousarray(triangles, np.float32).reshape(-1, 9)
    sah = sah or SahCost()
    h = C.c_void_p()
    _check(lib().ptx_kdtree_build(_ptr(t), len(t), C.byref(sah), C.byref(h)))
    try:
        nodes, leaf = _view_tree(h)
    finally:
        lib().ptx_kdtree_free(h)
    return KdTree(t, nodes, leaf)


class LoadedScene:
    """`setup_scene` (pathtracer-cli/src/main.rs:123-159): OBJ + MTL + KD-tree, still on the host."""

    def __init__(self, obj_path, sah=None, max_bounces=10, verbose=False):
        sah = sah or SahCost()
        self._h = C.c_void_p()
        _check(lib().ptx_load_obj(os.fsencode(obj_path), C.byref(sah), max_bounces, int(verbose), C.byref(self._h)))
        self.desc = _SceneDesc()
        _check(lib().ptx_loaded_desc(self._h, C.byref(self.desc)))
        cam = (C.c_float * 10)()
        _check(lib().ptx_loaded_camera(self._h, cam))
        self.camera = np.array(cam[:], np.float32)
        self.path = obj_path

    def __del__(self):
        try:
            if self._h:
                lib().ptx_loaded_free(self._h)
        except Exception:
            pass

    def _view(self, ptr, count, dtype):
        if not ptr or not count:
            return np.zeros(0, dtype)
        return np.frombuffer(C.string_at(ptr, count * np.dtype(dtype).itemsize), dtype).copy()

    @property
    def n_triangles(self):
        return self.desc.n_triangles

    def arrays(self):
        d = self.desc
        return dict(
            triangles=self._view(d.triangles, d.n_triangles * 9, np.float32).reshape(-1, 9),
            normals=self._view(d.normals, d.n_triangles * 9, np.float32).reshape(-1, 9),
            texcoords=self._view(d.texcoords, d.n_triangles * 6, np.float32).reshape(-1, 6),
            material_ids=self._view(d.material_ids, d.n_triangles, np.uint32),
            nodes=self._view(d.nodes, d.n_nodes, NODE_DTYPE),
            leaf_indices=self._view(d.leaf_indices, d.n_leaf_indices, np.uint32),
            materials=np.array([[*m.albedo, *m.schlick_f0, m.transmission, m.ior] for m in (d.materials[i] for i in range(d.n_materials))], np.float32).reshape(-1, 8),
            material_textures=[d.materials[i].texture for i in range(d.n_materials)],
            lights=np.array([[*l.center, *l.intensity, l.radius] for l in (d.lights[i] for i in range(d.n_lights))], np.float32).reshape(-1, 7),
        )

    def texture(self, index):
        t = self.desc.textures[index]
        n = t.width * t.height * 3
        return np.frombuffer(C.string_at(t.rgb, n * 4), np.float32).reshape(t.height, t.width, 3).copy()

    def kdtree(self):
        a = self.arrays()
        return KdTree(a["triangles"], a["nodes"], a["leaf_indices"])

    def pinhole(self, width, height):
        """`Pinhole::new(mtl.cameras[0].clone().into(), size)` (main.rs:137)."""
        c = self.camera
        return Pinhole.new(c[0:3], c[3:6], c[6:9], float(c[9]), width, height)

    def upload(self, flags=0, device=None):
        """Device-resident copy on the current HIP device, or on HIP device `device`."""
        return DeviceScene(self.desc, flags=flags, keepalive=self, device=device)


class DeviceScene:
    """A scene resident on the current HIP device: the `TriangleCollection` + `Pathtracer` state."""

    def __init__(self, desc, flags=0, keepalive=None, device=None):
        self._h = C.c_void_p()
        self._keep = keepalive
        if device is None:
            _check(lib().ptx_scene_create(C.byref(desc), flags, C.byref(self._h)))
        else:
            _check(lib().ptx_scene_create_on(C.byref(desc), flags, int(device), C.byref(self._h)))
        self.max_bounces = desc.max_bounces
        self.n_lights = desc.n_lights
        self._raylog_cap = 0

    @classmethod
    def from_arrays(cls, triangles, nodes, leaf_indices, normals=None, texcoords=None, material_ids=None,
                    materials=None, material_textures=None, textures=None, lights=None,
                    environment=(0.8, 0.8, 0.8), max_bounces=10, flags=0):
        keep = {}
        d = _SceneDesc()
        keep["tri"] = t = np.ascontiguousarray(triangles, np.float32).reshape(-1, 9)
        keep["nodes"] = n = np.ascontiguousarray(nodes, NODE_DTYPE)
        keep["leaf"] = l = np.ascontiguousarray(leaf_indices, np.uint32)
        d.triangles, d.n_triangles = _ptr(t), len(t)
        d.nodes, d.n_nodes = _ptr(n), len(n)
        d.leaf_indices, d.n_leaf_indices = _ptr(l), len(l)
        if normals is not None:
            keep["nrm"] = a = np.ascontiguousarray(normals, np.float32).reshape(-1, 9)
            keep["uv"] = b = np.ascontiguousarray(texcoords, np.float32).reshape(-1, 6)
            keep["mid"] = c = np.ascontiguousarray(material_ids, np.uint32)
            if not (len(a) == len(b) == len(c) == len(t)):
                raise PtxError(-1, "per-triangle arrays must all have n_triangles rows")
            d.normals, d.texcoords, d.material_ids = _ptr(a), _ptr(b), _ptr(c)
            textures = textures or []
            keep["texdata"] = td = [np.ascontiguousarray(x, np.float32) for x in textures]
            keep["tex"] = tx = (_Texture * max(len(td), 1))()
            for i, x in enumerate(td):
                tx[i].width, tx[i].height = x.shape[1], x.shape[0]
                tx[i].rgb = x.ctypes.data_as(C.POINTER(C.c_float))
            d.textures, d.n_textures = tx, len(td)
            mats = np.asarray(materials, np.float32).reshape(-1, 8)
            keep["mat"] = ma = (_Material * max(len(mats), 1))()
            for i, m in enumerate(mats):
                ma[i].albedo = (C.c_float * 3)(*m[0:3])
                ma[i].schlick_f0 = (C.c_float * 3)(*m[3:6])
                ma[i].transmission, ma[i].ior = float(m[6]), float(m[7])
                ma[i].texture = -1 if material_textures is None else int(material_textures[i])
            d.materials, d.n_materials = ma, len(mats)
        lights = np.zeros((0, 7), np.float32) if lights is None else np.asarray(lights, np.float32).reshape(-1, 7)
        keep["lights"] = li = (_Light * max(len(lights), 1))()
        for i, x in enumerate(lights[:MAX_LIGHTS]):
            li[i].center = (C.c_float * 3)(*x[0:3])
            li[i].intensity = (C.c_float * 3)(*x[3:6])
            li[i].radius = float(x[6])
        d.lights, d.n_lights = li, len(lights)
        d.environment = (C.c_float * 3)(*environment)
        d.max_bounces = max_bounces
        return cls(d, flags=flags, keepalive=keep)

    def close(self):
        if self._h:
            lib().ptx_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- accel seam: KdNode::intersect in bulk ------------------------------------------------------
    def intersect(self, rays):
        """Host arrays in, host arrays out (copies over PCIe)."""
        rays = np.ascontiguousarray(rays, RAY_DTYPE)
        hits = np.zeros(len(rays), HIT_DTYPE)
        _check(lib().ptx_intersect_batch(self._h, _ptr(rays), C.c_size_t(len(rays)), _ptr(hits)))
        return hits

    def intersect_device(self, d_rays_ptr, n, d_hits_ptr, stream=None):
        _check(lib().ptx_intersect_batch_device(self._h, C.c_void_p(d_rays_ptr), C.c_size_t(n), C.c_void_p(d_hits_ptr), C.c_void_p(stream)))

    def intersect_device_noalloc(self, d_rays_ptr, n, d_hits_ptr, stream=None):
        """ptx_intersect_batch_device for callers that cannot allocate (library-static scratch, <= NOALLOC_MAX_RAYS)."""
        _check(lib().ptx_intersect_batch_device_noalloc(self._h, C.c_void_p(d_rays_ptr), C.c_size_t(n), C.c_void_p(d_hits_ptr), C.c_void_p(stream)))

    def occluded_device(self, d_rays_ptr, n, d_out_ptr, stream=None):
        _check(lib().ptx_occluded_batch_device(self._h, C.c_void_p(d_rays_ptr), C.c_size_t(n), C.c_void_p(d_out_ptr), C.c_void_p(stream)))

    # ---- integrator seam ----------------------------------------------------------------------------
    def set_batch_samples(self, n):
        _check(lib().ptx_set_batch_samples(self._h, n))

    def set_queue_slots(self, slots):
        _check(lib().ptx_set_queue_slots(self._h, C.c_uint64(slots)))

    def set_pipelines(self, n):
        _check(lib().ptx_set_pipelines(self._h, n))

    def trim(self):
        """Hands the queues and staging buffers back to the device (the next render allocates again)."""
        _check(lib().ptx_scene_trim(self._h))

    def memory(self):
        """Bytes of render state (queues, radiance, stacks) held right now."""
        b = C.c_uint64()
        _check(lib().ptx_scene_memory(self._h, C.byref(b)))
        return b.value

    @property
    def device(self):
        d = C.c_int()
        _check(lib().ptx_scene_device(self._h, C.byref(d)))
        return d.value

    def render(self, pinhole, sample_count, seed=0, first_sample=0, accum=None, progress=None):
        """Adds samples [first_sample, first_sample+sample_count) into accum (H x W x 3 float32)."""
        if accum is None:
            accum = np.zeros((pinhole.height, pinhole.width, 3), np.float32)
        assert accum.dtype == np.float32 and accum.flags.c_contiguous and accum.size == pinhole.width * pinhole.height * 3
        cb_t = C.CFUNCTYPE(None, C.c_uint32, C.c_double, C.c_void_p)
        cb = cb_t(lambda done, secs, _u: progress(done, secs)) if progress else C.cast(None, cb_t)
        _check(lib().ptx_render(self._h, C.byref(pinhole), C.c_uint64(seed), first_sample, sample_count, _ptr(accum), cb, None))
        return accum

    def render_device(self, pinhole, sample_count, d_accum_ptr, seed=0, first_sample=0, stream=None):
        _check(lib().ptx_render_device(self._h, C.byref(pinhole), C.c_uint64(seed), first_sample, sample_count,
                                       C.c_void_p(d_accum_ptr), C.c_void_p(stream)))

    def render_region_device(self, pinhole, x, y, width, height, sample_count, d_accum_ptr, seed=0, first_sample=0, stream=None):
        """`render_subdivided_mut`: the samples of one rectangle of the image only."""
        _check(lib().ptx_render_region_device(self._h, C.byref(pinhole), x, y, width, height, C.c_uint64(seed), first_sample,
                                              sample_count, C.c_void_p(d_accum_ptr), C.c_void_p(stream)))

    def stats(self):
        c = Counters()
        _check(lib().ptx_stats(self._h, C.byref(c)))
        return c.as_dict()

    def round_times(self, n=None):
        """(extend ms, shadow ms) per round since stats_reset (FLAG_TIME_KERNELS); round 1 = camera rays."""
        n = n or max(self.max_bounces, 1)
        e, s = np.zeros(n, np.float64), np.zeros(n, np.float64)
        _check(lib().ptx_round_times(self._h, _ptr(e), _ptr(s), n))
        return e, s

    def last_batch_counts(self):
        n = 8 * max(self.max_bounces, 1)  # rounds: up to 8 slot-loads of paths per batch
        e, s = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        _check(lib().ptx_last_batch_counts(self._h, _ptr(e), _ptr(s), n))
        used = max(int(np.nonzero(e)[0].max()) + 1 if e.any() else 1, 1)
        return e[:used], s[:used]

    def stats_reset(self):
        _check(lib().ptx_stats_reset(self._h))

    def check_report(self):
        """(rays checked, disagreements) of a FLAG_CHECK_BRUTE scene: `kdtree-tester-cli test` on the GPU."""
        a, b = C.c_uint64(), C.c_uint64()
        _check(lib().ptx_check_report(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def check_fail_log(self, cap=65536):
        """The 50-byte fail records of checked_intersection.rs:27-56 as a structured array."""
        raw = np.zeros(cap * 50, np.uint8)
        n = C.c_uint32()
        _check(lib().ptx_check_fail_log(self._h, _ptr(raw), cap, C.byref(n)))
        return raw[: n.value * 50].view(FAIL_DTYPE)

    def raylog_enable(self, max_records):
        """The reference's `ray_logging` feature: log every ray of the following renders (0 = off)."""
        _check(lib().ptx_raylog_enable(self._h, C.c_uint64(max_records)))
        self._raylog_cap = int(max_records)

    def raylog_read(self):
        """(records in the reference's file order as RAYLOG_DTYPE, dropped count); empties the log."""
        cap = self._raylog_cap
        raw = np.zeros(cap * 33, np.uint8)
        n, dropped = C.c_uint64(), C.c_uint64()
        _check(lib().ptx_raylog_read(self._h, _ptr(raw), C.c_uint64(cap), C.byref(n), C.byref(dropped)))
        return raw[: n.value * 33].view(RAYLOG_DTYPE), dropped.value

    def loop_profile(self):
        """Diagnostic (FLAG_COUNT_WORK scenes): lane occupancy of the traversal loop, per kernel."""
        raw = np.zeros(24, np.uint64)
        _check(lib().ptx_debug_loop_profile(self._h, _ptr(raw)))
        keys = ("iters", "walk_lanes", "tri_iters", "tri_lanes", "end_iters", "end_lanes", "refills", "live_lanes",
                "clk_refill", "clk_node", "clk_tri", "clk_exit")
        return {"extend": dict(zip(keys, (int(x) for x in raw[:12]))), "shadow": dict(zip(keys, (int(x) for x in raw[12:])))}


NOALLOC_MAX_RAYS = 1 << 24
REDUCE_AUTO, REDUCE_RCCL, REDUCE_PEER = 0, 1, 2


def device_count():
    n = C.c_int()
    _check(lib().ptx_device_count(C.byref(n)))
    return n.value


def shard_samples(first_sample, sample_count, n_shards, shard):
    """(first, count) of shard `shard`: GPU g renders [first + g*S/N, first + (g+1)*S/N) (ptx_shard_samples)."""
    a, b = C.c_uint32(), C.c_uint32()
    _check(lib().ptx_shard_samples(first_sample, sample_count, n_shards, shard, C.byref(a), C.byref(b)))
    return a.value, b.value


def render_multi(scenes, pinhole, sample_count, seed=0, first_sample=0, accum=None, reduce=REDUCE_AUTO, progress=None):
    """`render_parallel_iterations` over several GPUs of this process (ptx_render_multi): scenes[g] lives on
    GPU g; adds samples [first_sample, first_sample+sample_count) into accum (H x W x 3 float32)."""
    if accum is None:
        accum = np.zeros((pinhole.height, pinhole.width, 3), np.float32)
    assert accum.dtype == np.float32 and accum.flags.c_contiguous and accum.size == pinhole.width * pinhole.height * 3
    handles = (C.c_void_p * len(scenes))(*[s._h for s in scenes])
    cb_t = C.CFUNCTYPE(None, C.c_uint32, C.c_double, C.c_void_p)
    cb = cb_t(lambda done, secs, _u: progress(done, secs)) if progress else C.cast(None, cb_t)
    _check(lib().ptx_render_multi(handles, len(scenes), C.byref(pinhole), C.c_uint64(seed), first_sample, sample_count,
                                  _ptr(accum), reduce, cb, None))
    return accum


def multi_shutdown():
    """Destroys the RCCL communicators ptx_render_multi cached (ptx_multi_shutdown)."""
    _check(lib().ptx_multi_shutdown())


def multi_last_reduce_ms():
    ms = C.c_double()
    _check(lib().ptx_multi_last_reduce_ms(C.byref(ms)))
    return ms.value


def to_rgb8(accum, iterations):
    """`ImageBuffer::to_rgb8(iterations as u16)`."""
    a = np.ascontiguousarray(accum, np.float32)
    h, w = a.shape[0], a.shape[1]
    out = np.zeros((h, w, 3), np.uint8)
    _check(lib().ptx_tonemap_rgb8(_ptr(a), w, h, iterations, _ptr(out)))
    return out


def to_rgb8_device(d_accum_ptr, width, height, iterations, d_out_ptr, stream=None):
    """`ImageBuffer::to_rgb8` on the device: width*height*3 floats -> bytes, asynchronous on `stream`."""
    _check(lib().ptx_tonemap_rgb8_device(C.c_void_p(d_accum_ptr), width, height, iterations, C.c_void_p(d_out_ptr), C.c_void_p(stream)))


def write_png(path, rgb8):
    a = np.ascontiguousarray(rgb8, np.uint8)
    _check(lib().ptx_png_write_rgb8(os.fsencode(path), _ptr(a), a.shape[1], a.shape[0]))


def save_accum(path, accum, seed, samples):
    """Accumulation-buffer checkpoint: H x W x 3 float32 sums + seed + samples done (ptx_accum_save)."""
    a = np.ascontiguousarray(accum, np.float32)
    assert a.ndim == 3 and a.shape[2] == 3
    _check(lib().ptx_accum_save(os.fsencode(path), _ptr(a), a.shape[1], a.shape[0], C.c_uint64(seed), samples))


def load_accum(path):
    """(accum H x W x 3, seed, samples done) of a checkpoint; continue with render(first_sample=samples)."""
    w, h, seed, n = C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_uint32()
    _check(lib().ptx_accum_load(os.fsencode(path), None, C.byref(w), C.byref(h), C.byref(seed), C.byref(n)))
    a = np.zeros((h.value, w.value, 3), np.float32)
    _check(lib().ptx_accum_load(os.fsencode(path), _ptr(a), C.byref(w), C.byref(h), C.byref(seed), C.byref(n)))
    return a, seed.value, n.value


def read_png_linear(path):
    p = C.POINTER(C.c_float)()
    w, h = C.c_uint32(), C.c_uint32()
    _check(lib().ptx_png_read_linear(os.fsencode(path), C.byref(p), C.byref(w), C.byref(h)))
    try:
        return np.frombuffer(C.string_at(p, w.value * h.value * 12), np.float32).reshape(h.value, w.value, 3).copy()
    finally:
        lib().ptx_free(p)


def render_parallel_iterations(scene, pinhole, threads, iterations_per_thread, seed=0, progress=None):
    """`render_parallel_iterations` (tracing/src/worker.rs:84-119): threads x iterations_per_thread
    full-frame samples; returns (seconds, rgb8 image).  `scene` is one DeviceScene (one GPU plays every
    thread) or a list of them, one per GPU (the samples are sharded over the GPUs, ptx_render_multi)."""
    import time

    total = threads * iterations_per_thread
    t0 = time.perf_counter()
    if isinstance(scene, (list, tuple)):
        accum = render_multi(scene, pinhole, total, seed=seed, progress=progress)
    else:
        accum = scene.render(pinhole, total, seed=seed, progress=progress)
    secs = time.perf_counter() - t0
    return secs, to_rgb8(accum, total)
