"""Python binding (ctypes) to libraytracer_amd.so, the MI355X path-tracing
backend.  The C ABI is declared in include/raytracer_amd.h; this module is a
thin convenience layer for tests, the benchmark and scripts.  All compute runs
in the HIP library: there is no Python or CPU fallback, and a missing library
or GPU raises.

A process that also uses torch on the GPU must `import torch` BEFORE the first
call into this module: the torch wheel bundles its own HIP runtime, and the
runtime that is loaded second finds no devices.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libraytracer_amd.so")
PROBE_LIB_PATH = os.path.join(_HERE, "libraytracer_amd_probe.so")
CLI_PATH = os.path.join(_HERE, "raytracer")

MEM_HOST, MEM_DEVICE = 0, 1
PRIM_TRIANGLE, PRIM_SPHERE = 0, 1
RENDER_TIME_KERNELS, RENDER_COUNT_WORK, RENDER_SMALL_PASSES, RENDER_EXACT_ONLY = 1, 2, 4, 8
STATIC_MAX_RAYS = 1 << 20
TRACE_COUNT_WORK, TRACE_EXACT_ONLY = 0x100, 0x200

OK = 0
E_INVALID, E_PARSE, E_IO, E_NO_DEVICE, E_DEVICE, E_NO_SCENE, E_LIMIT = -1, -2, -3, -4, -5, -6, -7


class RtaError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"[{code}] {text}")
        self.code = code
        self.text = text


class SceneSettings(C.Structure):
    _fields_ = [("camera", C.c_float * 3), ("look_at", C.c_float * 3), ("up", C.c_float * 3),
                ("fov", C.c_float), ("defocus_angle", C.c_float), ("exposure", C.c_float),
                ("sky", C.c_float * 3), ("width", C.c_int32), ("height", C.c_int32), ("samples", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [("n_prims", C.c_int32), ("n_nodes", C.c_int32), ("n_materials", C.c_int32), ("n_lights", C.c_int32),
                ("prim_type", C.POINTER(C.c_int32)), ("prim_material", C.POINTER(C.c_int32)),
                ("prim_light", C.POINTER(C.c_int32)), ("prim_geom", C.POINTER(C.c_float)),
                ("prim_normals", C.POINTER(C.c_float)), ("node_bounds", C.POINTER(C.c_float)),
                ("node_index", C.POINTER(C.c_int32)), ("node_count", C.POINTER(C.c_int32)),
                ("materials", C.POINTER(C.c_float)), ("light_color", C.POINTER(C.c_float)),
                ("light_prim", C.POINTER(C.c_int32))]


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("upper_left", C.c_float * 3), ("du", C.c_float * 3),
                ("dv", C.c_float * 3), ("lens_u", C.c_float * 3), ("lens_v", C.c_float * 3),
                ("defocus_angle", C.c_float)]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("samples", C.c_int32), ("exposure", C.c_float),
                ("sky", C.c_float * 3), ("seed", C.c_uint32), ("samples_per_stream", C.c_int32),
                ("max_bounces", C.c_int32), ("shard_index", C.c_int32), ("shard_count", C.c_int32),
                ("slots", C.c_int32), ("out_mem", C.c_int32), ("flags", C.c_int32), ("run_blocks", C.c_int32),
                ("streams", C.c_int32), ("block_begin", C.c_int32), ("block_end", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("steps", C.c_uint64),
                ("inner_visits", C.c_uint64), ("tri_tests", C.c_uint64), ("sph_tests", C.c_uint64),
                ("gpu_ms", C.c_double), ("intersect_ms", C.c_double), ("shadow_ms", C.c_double),
                ("intersect_launches", C.c_uint64), ("shadow_launches", C.c_uint64),
                ("early_rays", C.c_uint64), ("advance_ms", C.c_double), ("advance_launches", C.c_uint64),
                ("retraced_rays", C.c_uint64), ("walked_rays", C.c_uint64), ("listed_rays", C.c_uint64),
                ("cell_tests", C.c_uint64), ("camera_ms", C.c_double), ("leftover_ms", C.c_double), ("camera_launches", C.c_uint64),
                ("scanned_rays", C.c_uint64)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class BatchStats(C.Structure):
    _fields_ = [("rays", C.c_int64), ("sorted", C.c_int32), ("walked", C.c_int32), ("packets", C.c_int32),
                ("scanned", C.c_int32), ("retraced_rays", C.c_uint64),
                ("order_ms", C.c_double), ("walk_ms", C.c_double), ("replay_ms", C.c_double),
                ("walk_node_visits", C.c_uint64), ("walk_tri_tests", C.c_uint64),
                ("replay_node_visits", C.c_uint64), ("replay_prim_tests", C.c_uint64)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


HIT_DTYPE = np.dtype([("prim", "<i4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])

# name -> (restype, argtypes); also the list of symbols the header declares
_SIGNATURES = {
    "rta_scene_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]),
    "rta_scene_parse": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]),
    "rta_scene_from_triangles": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]),
    "rta_scene_free": (None, [C.c_void_p]),
    "rta_scene_build_bvh": (C.c_int, [C.c_void_p]),
    "rta_scene_build_bvh_device": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "rta_scene_get_settings": (C.c_int, [C.c_void_p, C.POINTER(SceneSettings)]),
    "rta_scene_set_settings": (C.c_int, [C.c_void_p, C.POINTER(SceneSettings)]),
    "rta_scene_describe": (C.c_int, [C.c_void_p, C.POINTER(SceneDesc)]),
    "rta_validate_scene": (C.c_int, [C.POINTER(SceneDesc), C.c_char_p, C.c_size_t]),
    "rta_scene_permutation": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_int32)]),
    "rta_camera_from_settings": (C.c_int, [C.POINTER(SceneSettings), C.POINTER(Camera)]),
    "rta_write_bmp": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.c_void_p]),
    "rta_global_error": (C.c_char_p, []),
    "rta_device_count": (C.c_int, []),
    "rta_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "rta_destroy": (None, [C.c_void_p]),
    "rta_last_error": (C.c_char_p, [C.c_void_p]),
    "rta_upload_scene": (C.c_int, [C.c_void_p, C.POINTER(SceneDesc)]),
    "rta_trace_closest": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]),
    "rta_trace_closest_static": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "rta_last_batch_stats": (C.c_int, [C.c_void_p, C.POINTER(BatchStats)]),
    "rta_trace_occluded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]),
    "rta_render": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.c_void_p, C.c_void_p,
                             C.POINTER(Stats)]),
    "rta_resolve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p,
                              C.c_int]),
    "rta_shard_pixels": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "rta_stream": (C.c_void_p, [C.c_void_p]),
}

# the probe library (include/raytracer_amd_probe.h): measurement and test hooks
_PROBE_SIGNATURES = {
    "rta_debug_arith": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "rta_debug_trace_variant": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]),
    "rta_debug_scene_buffer": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
}

_lib = None


def lib():
    """The loaded shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `make` (hipcc, gfx950)")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


_probe = None


def probe_lib():
    """The probe library, loaded after the product library it works on."""
    global _probe
    if _probe is None:
        lib()
        if not os.path.exists(PROBE_LIB_PATH):
            raise RuntimeError(f"{PROBE_LIB_PATH} not found: build it with `make` (hipcc, gfx950)")
        handle = C.CDLL(PROBE_LIB_PATH)
        for name, (res, args) in _PROBE_SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _probe = handle
    return _probe


def exported_symbols():
    return sorted(_SIGNATURES)


def probe_symbols():
    return sorted(_PROBE_SIGNATURES)


def device_count():
    return int(lib().rta_device_count())


def _global_error():
    return (lib().rta_global_error() or b"").decode(errors="replace")


def _np(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


class Scene:
    """A host-side scene: parse, build the BVH, expose the flat arrays."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)
        self.built = False

    @classmethod
    def load(cls, path):
        h = C.c_void_p()
        err = C.create_string_buffer(1024)
        rc = lib().rta_scene_load(os.fsencode(path), C.byref(h), err, len(err))
        if rc != OK:
            raise RtaError(rc, err.value.decode(errors="replace"))
        return cls(h.value)

    @classmethod
    def parse(cls, text, name="scene.txt"):
        h = C.c_void_p()
        err = C.create_string_buffer(1024)
        rc = lib().rta_scene_parse(name.encode(), text.encode(), C.byref(h), err, len(err))
        if rc != OK:
            raise RtaError(rc, err.value.decode(errors="replace"))
        return cls(h.value)

    @classmethod
    def from_triangles(cls, verts):
        verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 9)
        h = C.c_void_p()
        rc = lib().rta_scene_from_triangles(verts.ctypes.data, verts.shape[0], C.byref(h))
        if rc != OK:
            raise RtaError(rc, _global_error())
        return cls(h.value)

    def __del__(self):
        try:
            if self._h:
                lib().rta_scene_free(self._h)
                self._h = None
        except Exception:
            pass

    def build_bvh(self):
        rc = lib().rta_scene_build_bvh(self._h)
        if rc != OK:
            raise RtaError(rc, _global_error())
        self.built = True
        return self

    def build_bvh_on(self, ctx):
        """The same tree, built on ctx's GPU; returns the build kernels' time in ms."""
        ms = C.c_float(0)
        ctx._check(lib().rta_scene_build_bvh_device(ctx._h, self._h, C.byref(ms)))
        self.built = True
        return ms.value

    @property
    def settings(self):
        s = SceneSettings()
        lib().rta_scene_get_settings(self._h, C.byref(s))
        return s

    @settings.setter
    def settings(self, s):
        lib().rta_scene_set_settings(self._h, C.byref(s))

    def set_render(self, width=None, height=None, samples=None):
        s = self.settings
        if width is not None:
            s.width = width
        if height is not None:
            s.height = height
        if samples is not None:
            s.samples = samples
        self.settings = s
        return self

    def camera(self):
        cam = Camera()
        s = self.settings
        rc = lib().rta_camera_from_settings(C.byref(s), C.byref(cam))
        if rc != OK:
            raise RtaError(rc, "bad settings")
        return cam

    def desc(self):
        d = SceneDesc()
        rc = lib().rta_scene_describe(self._h, C.byref(d))
        if rc != OK:
            raise RtaError(rc, _global_error())
        return d

    def arrays(self):
        """Copies of the flat arrays as numpy, keyed like rta_scene_desc."""
        d = self.desc()
        n, m, k, l = d.n_prims, d.n_nodes, d.n_materials, d.n_lights
        return {
            "prim_type": _np(d.prim_type, n, np.int32),
            "prim_material": _np(d.prim_material, n, np.int32),
            "prim_light": _np(d.prim_light, n, np.int32),
            "prim_geom": _np(d.prim_geom, n * 9, np.float32).reshape(n, 9),
            "prim_normals": _np(d.prim_normals, n * 9, np.float32).reshape(n, 9),
            "node_bounds": _np(d.node_bounds, m * 6, np.float32).reshape(m, 6),
            "node_index": _np(d.node_index, m, np.int32),
            "node_count": _np(d.node_count, m, np.int32),
            "materials": _np(d.materials, k * 6, np.float32).reshape(k, 6),
            "light_color": _np(d.light_color, l * 3, np.float32).reshape(l, 3),
            "light_prim": _np(d.light_prim, l, np.int32),
        }

    def permutation(self):
        p = C.POINTER(C.c_int32)()
        n = C.c_int32()
        lib().rta_scene_permutation(self._h, C.byref(p), C.byref(n))
        return _np(p, n.value, np.int32)


def desc_from_arrays(a):
    """Build an rta_scene_desc over numpy arrays (kept alive on the result)."""
    keep = {}

    def ptr(key, dtype, ctype):
        arr = np.ascontiguousarray(a[key], dtype)
        keep[key] = arr
        return arr.ctypes.data_as(C.POINTER(ctype))

    d = SceneDesc()
    d.n_prims = len(a["prim_type"])
    d.n_nodes = len(a["node_index"])
    d.n_materials = len(np.asarray(a["materials"]).reshape(-1, 6))
    d.n_lights = len(a["light_prim"])
    for key in ("prim_type", "prim_material", "prim_light", "node_index", "node_count", "light_prim"):
        setattr(d, key, ptr(key, np.int32, C.c_int32))
    for key in ("prim_geom", "prim_normals", "node_bounds", "materials", "light_color"):
        setattr(d, key, ptr(key, np.float32, C.c_float))
    d._keep = keep
    return d


def validate(desc):
    """Raises RtaError with the library's explanation if the description is malformed."""
    err = C.create_string_buffer(512)
    rc = lib().rta_validate_scene(C.byref(desc), err, len(err))
    if rc != OK:
        raise RtaError(rc, err.value.decode(errors="replace"))


def write_bmp(path, bgr):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w, _ = bgr.shape
    rc = lib().rta_write_bmp(os.fsencode(path), w, h, bgr.ctypes.data)
    if rc != OK:
        raise RtaError(rc, _global_error())


def shard_pixels(width, height, shard_index=0, shard_count=1):
    n = lib().rta_shard_pixels(width, height, shard_index, shard_count, None)
    if n < 0:
        raise RtaError(int(n), "bad shard arguments")
    out = np.empty(n, np.int32)
    lib().rta_shard_pixels(width, height, shard_index, shard_count, out.ctypes.data)
    return out


class Context:
    """One GPU.  Raises RtaError(E_NO_DEVICE) when there is none."""

    def __init__(self, device=0):
        h = C.c_void_p()
        rc = lib().rta_create(device, C.byref(h))
        if rc != OK:
            raise RtaError(rc, _global_error())
        self._h = h
        self.device = device

    def close(self):
        if self._h:
            lib().rta_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != OK:
            raise RtaError(rc, (lib().rta_last_error(self._h) or b"").decode(errors="replace"))

    @property
    def stream(self):
        return lib().rta_stream(self._h)

    def upload(self, scene_or_desc):
        d = scene_or_desc.desc() if isinstance(scene_or_desc, Scene) else scene_or_desc
        self._check(lib().rta_upload_scene(self._h, C.byref(d)))
        return self

    def trace_closest(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        hits = np.empty(rays.shape[0], HIT_DTYPE)
        self._check(lib().rta_trace_closest(self._h, rays.ctypes.data, rays.shape[0], hits.ctypes.data, MEM_HOST))
        return hits

    def trace_closest_device(self, rays_ptr, n, hits_ptr, count_work=False, exact_only=False):
        self._check(lib().rta_trace_closest(self._h, rays_ptr, n, hits_ptr,
                                            MEM_DEVICE | (TRACE_COUNT_WORK if count_work else 0) |
                                            (TRACE_EXACT_ONLY if exact_only else 0)))

    def last_batch_stats(self):
        """What the last trace_closest call did, stage by stage (rta_last_batch_stats)."""
        st = BatchStats()
        self._check(lib().rta_last_batch_stats(self._h, C.byref(st)))
        return st.as_dict()

    def trace_closest_static_device(self, rays_ptr, n, hits_ptr):
        """rta_trace_closest_static: device pointers, at most STATIC_MAX_RAYS rays, allocates nothing."""
        self._check(lib().rta_trace_closest_static(self._h, rays_ptr, n, hits_ptr))

    def trace_variant_device(self, variant, rays_ptr, n, hits_ptr):
        self._check(probe_lib().rta_debug_trace_variant(self._h, variant, rays_ptr, n, hits_ptr, MEM_DEVICE))

    def trace_occluded_device(self, rays_ptr, tmax_ptr, n, out_ptr):
        self._check(lib().rta_trace_occluded(self._h, rays_ptr, tmax_ptr, n, out_ptr, MEM_DEVICE))

    def trace_occluded(self, rays, tmax):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        tmax = np.ascontiguousarray(tmax, np.float32)
        out = np.empty(rays.shape[0], np.uint8)
        self._check(lib().rta_trace_occluded(self._h, rays.ctypes.data, tmax.ctypes.data, rays.shape[0],
                                             out.ctypes.data, MEM_HOST))
        return out

    def render(self, camera, width, height, samples, exposure=1.0, sky=(0, 0, 0), seed=69420, samples_per_stream=0,
               max_bounces=0, shard_index=0, shard_count=1, slots=0, flags=0, run_blocks=0, streams=0,
               block_begin=0, block_end=0, want_accum=True, want_bgr=True):
        """Host-buffer render.  Returns (bgr[h,w,3] uint8, accum[h,w,3] float32, stats dict)."""
        p = RenderParams(width=width, height=height, samples=samples, exposure=exposure, seed=seed,
                         samples_per_stream=samples_per_stream, max_bounces=max_bounces, shard_index=shard_index,
                         shard_count=shard_count, slots=slots, out_mem=MEM_HOST, flags=flags, run_blocks=run_blocks,
                         streams=streams, block_begin=block_begin, block_end=block_end)
        p.sky[:] = [float(x) for x in sky]
        bgr = np.zeros((height, width, 3), np.uint8) if want_bgr else None
        acc = np.zeros((height, width, 3), np.float32) if want_accum else None
        st = Stats()
        self._check(lib().rta_render(self._h, C.byref(camera), C.byref(p), acc.ctypes.data if want_accum else None,
                                     bgr.ctypes.data if want_bgr else None, C.byref(st)))
        return bgr, acc, st.as_dict()

    def render_device(self, camera, params, accum_ptr, bgr_ptr):
        """Device-buffer render: params.out_mem must be MEM_DEVICE."""
        st = Stats()
        self._check(lib().rta_render(self._h, C.byref(camera), C.byref(params), accum_ptr, bgr_ptr, C.byref(st)))
        return st.as_dict()

    def render_scene(self, scene, **kw):
        s = scene.settings
        kw.setdefault("exposure", s.exposure)
        kw.setdefault("sky", tuple(s.sky))
        return self.render(scene.camera(), s.width, s.height, s.samples, **kw)

    def resolve(self, accum, samples, exposure=1.0):
        accum = np.ascontiguousarray(accum, np.float32)
        h, w, _ = accum.shape
        bgr = np.empty((h, w, 3), np.uint8)
        self._check(lib().rta_resolve(self._h, accum.ctypes.data, w, h, samples, exposure, bgr.ctypes.data, MEM_HOST))
        return bgr

    def resolve_device(self, accum_ptr, width, height, samples, exposure, bgr_ptr):
        self._check(lib().rta_resolve(self._h, accum_ptr, width, height, samples, exposure, bgr_ptr, MEM_DEVICE))

    def scene_buffer(self, which):
        """A device layout of the uploaded scene as bytes (rta_debug_scene_buffer): 0 pairs, 1 primitives,
        2 shading records, 3 reference nodes, 4 wide nodes, 5 leaf boxes, 6 scalars; 7, 8, 9: the cell lists of the view the
        last render prepared (starts, triangle numbers, keys)."""
        n = C.c_size_t(0)
        self._check(probe_lib().rta_debug_scene_buffer(self._h, which, None, 0, C.byref(n)))
        out = np.empty(n.value, np.uint8)
        if n.value:
            self._check(probe_lib().rta_debug_scene_buffer(self._h, which, out.ctypes.data, n.value, C.byref(n)))
        return out

    def arith(self, op, x, y=None):
        ops = {"div": 0, "sqrt": 1, "sin": 2, "cos": 3, "log": 4, "pow": 5, "fma": 6, "muladd": 7, "quot": 8}
        x = np.ascontiguousarray(x, np.float32)
        y = np.ascontiguousarray(y if y is not None else np.ones_like(x), np.float32)
        out = np.empty_like(x)
        self._check(probe_lib().rta_debug_arith(self._h, ops[op], x.ctypes.data, y.ctypes.data, out.ctypes.data, x.size))
        return out
