"""ctypes bindings to the CPU checker (oracle/liboracle.so) and, where it has
been built, to the real reference (oracle/_ref/libref.so).

Test infrastructure: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() only.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref.so")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "raytracer")
SCENES = os.path.join(ROOT, "scenes")

RNG_ROW, RNG_STREAM = 0, 1
MATH_LIBM, MATH_PM = 0, 1

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")

OBJ_DTYPE = np.dtype([("type", "<i4"), ("material", "<i4"), ("light", "<i4"), ("geom", "<f4", (18,))])
NODE_DTYPE = np.dtype([("lo", "<f4", (3,)), ("hi", "<f4", (3,)), ("index", "<i4"), ("count", "<i4")])
MAT_DTYPE = np.dtype([("color", "<f4", (3,)), ("roughness", "<f4"), ("ior", "<f4"), ("metallic", "<f4")])
LIGHT_DTYPE = np.dtype([("color", "<f4", (3,)), ("object", "<i4")])
assert OBJ_DTYPE.itemsize == 84 and NODE_DTYPE.itemsize == 32


def _load_oracle():
    lib = C.CDLL(ORACLE_SO)
    lib.orc_scene_load.restype = C.c_void_p
    lib.orc_scene_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    lib.orc_scene_parse_text.restype = C.c_void_p
    lib.orc_scene_parse_text.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
    lib.orc_scene_from_triangles.restype = C.c_void_p
    lib.orc_scene_from_triangles.argtypes = [_f32p, C.c_int32]
    lib.orc_scene_free.argtypes = [C.c_void_p]
    lib.orc_build_bvh.argtypes = [C.c_void_p]
    lib.orc_bvh_depth.argtypes = [C.c_void_p]
    lib.orc_scene_info.argtypes = [C.c_void_p, _i32p]
    lib.orc_scene_params.argtypes = [C.c_void_p, _f32p]
    lib.orc_scene_set_render.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    lib.orc_scene_set_camera.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, C.c_float, C.c_float]
    lib.orc_camera.argtypes = [C.c_void_p, _f32p]
    for name in ("orc_scene_objects", "orc_scene_nodes", "orc_scene_materials", "orc_scene_lights"):
        getattr(lib, name).restype = C.c_void_p
        getattr(lib, name).argtypes = [C.c_void_p]
    lib.orc_trace_closest.argtypes = [C.c_void_p, _f32p, C.c_int64, _i32p, _f32p, _f32p, _f32p, C.c_void_p]
    lib.orc_trace_occluded.argtypes = [C.c_void_p, _f32p, _f32p, C.c_int64, _u8p]
    lib.orc_hit_surface.argtypes = [C.c_void_p, _f32p, C.c_int64, _i32p, _f32p, _f32p, _f32p, _f32p, _f32p]
    lib.orc_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_int32, C.c_int, C.c_void_p,
                               C.c_void_p, C.c_void_p]
    lib.orc_rng_next.restype = C.c_uint32
    lib.orc_rng_next.argtypes = [C.POINTER(C.c_uint32)]
    lib.orc_rng_float.restype = C.c_float
    lib.orc_rng_float.argtypes = [C.POINTER(C.c_uint32)]
    lib.orc_stream_seed.restype = C.c_uint32
    lib.orc_stream_seed.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    lib.orc_sample_bsdf.argtypes = [C.c_int, _f32p, _f32p, C.POINTER(C.c_uint32), _f32p]
    lib.orc_eval_bsdf.argtypes = [C.c_int, _f32p, _f32p, _f32p, _f32p]
    lib.orc_trace_path.argtypes = [C.c_void_p, C.c_int, _f32p, C.POINTER(C.c_uint32), _f32p]
    lib.orc_tonemap.restype = C.c_float
    lib.orc_tonemap.argtypes = [C.c_int, C.c_float, C.c_float]
    lib.orc_pmath.argtypes = [C.c_int, _f32p, _f32p, _f32p, C.c_int64]
    lib.orc_write_bmp.argtypes = [C.c_char_p, C.c_int32, C.c_int32, _u8p]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            raise RuntimeError(f"{ORACLE_SO} missing: run `make -C oracle`")
        _lib = _load_oracle()
    return _lib


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Scene:
    """A scene held by the checker library."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("null scene")
        self.h = C.c_void_p(handle)

    @classmethod
    def load(cls, path):
        err = C.create_string_buffer(512)
        h = lib().orc_scene_load(os.fsencode(path), err, 512)
        if not h:
            raise ValueError(err.value.decode())
        return cls(h)

    @classmethod
    def parse(cls, text, name="scene.txt"):
        err = C.create_string_buffer(512)
        h = lib().orc_scene_parse_text(name.encode(), text.encode(), err, 512)
        if not h:
            raise ValueError(err.value.decode())
        return cls(h)

    @classmethod
    def from_triangles(cls, verts):
        verts = f32(verts).reshape(-1, 9)
        return cls(lib().orc_scene_from_triangles(verts, verts.shape[0]))

    def __del__(self):
        try:
            if self.h:
                lib().orc_scene_free(self.h)
                self.h = None
        except Exception:
            pass

    def build_bvh(self):
        if lib().orc_build_bvh(self.h) != 0:
            raise RuntimeError("orc_build_bvh failed (empty scene?)")
        return self

    def info(self):
        out = np.zeros(8, np.int32)
        lib().orc_scene_info(self.h, out)
        keys = ("objects", "nodes", "materials", "lights", "width", "height", "samples")
        return dict(zip(keys, (int(x) for x in out)))

    def params(self):
        out = np.zeros(16, np.float32)
        lib().orc_scene_params(self.h, out)
        return {"camera": out[0:3].copy(), "look_at": out[3:6].copy(), "up": out[6:9].copy(), "fov": float(out[9]),
                "defocus_angle": float(out[10]), "exposure": float(out[11]), "sky": out[12:15].copy()}

    def set_render(self, width=-1, height=-1, samples=-1):
        lib().orc_scene_set_render(self.h, width, height, samples)
        return self

    def set_camera(self, camera, look_at, up=(0, 0, 1), fov=90.0, defocus_angle=-1.0):
        lib().orc_scene_set_camera(self.h, f32(camera), f32(look_at), f32(up), fov, defocus_angle)
        return self

    def camera(self):
        out = np.zeros(19, np.float32)
        lib().orc_camera(self.h, out)
        return out

    def depth(self):
        return int(lib().orc_bvh_depth(self.h))

    def _array(self, getter, dtype, n):
        ptr = getattr(lib(), getter)(self.h)
        if not ptr or n == 0:
            return np.zeros(0, dtype)
        buf = (C.c_char * (dtype.itemsize * n)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=n).copy()

    def objects(self):
        return self._array("orc_scene_objects", OBJ_DTYPE, self.info()["objects"])

    def nodes(self):
        return self._array("orc_scene_nodes", NODE_DTYPE, self.info()["nodes"])

    def materials(self):
        return self._array("orc_scene_materials", MAT_DTYPE, self.info()["materials"])

    def lights(self):
        return self._array("orc_scene_lights", LIGHT_DTYPE, self.info()["lights"])

    def trace_closest(self, rays, counters=False):
        rays = f32(rays).reshape(-1, 6)
        n = rays.shape[0]
        idx = np.empty(n, np.int32)
        t = np.empty(n, np.float32)
        u = np.empty(n, np.float32)
        v = np.empty(n, np.float32)
        cn = np.zeros(6, np.uint64)
        rc = lib().orc_trace_closest(self.h, rays, n, idx, t, u, v, cn.ctypes.data)
        assert rc == 0
        if counters:
            return idx, t, u, v, cn
        return idx, t, u, v

    def trace_occluded(self, rays, dist):
        rays = f32(rays).reshape(-1, 6)
        dist = f32(dist)
        out = np.empty(rays.shape[0], np.uint8)
        rc = lib().orc_trace_occluded(self.h, rays, dist, rays.shape[0], out)
        assert rc == 0
        return out

    def hit_surface(self, rays, idx, t, u, v):
        rays = f32(rays).reshape(-1, 6)
        n = rays.shape[0]
        point = np.empty((n, 3), np.float32)
        normal = np.empty((n, 3), np.float32)
        lib().orc_hit_surface(self.h, rays, n, np.ascontiguousarray(idx, np.int32), f32(t), f32(u), f32(v), point,
                              normal)
        return point, normal

    def render(self, rng_mode=RNG_ROW, math_mode=MATH_LIBM, seed=69420, samples_per_stream=0, threads=0):
        """Returns (bgr[h,w,3] uint8, accum[h,w,3] float32, counters dict)."""
        i = self.info()
        w, h = i["width"], i["height"]
        bgr = np.zeros((h, w, 3), np.uint8)
        acc = np.zeros((h, w, 3), np.float32)
        cn = np.zeros(6, np.uint64)
        rc = lib().orc_render(self.h, rng_mode, math_mode, seed, samples_per_stream, threads, bgr.ctypes.data,
                              acc.ctypes.data, cn.ctypes.data)
        assert rc == 0
        keys = ("rays", "shadow_rays", "samples", "inner_visits", "tri_tests", "sph_tests")
        return bgr, acc, dict(zip(keys, (int(x) for x in cn)))

    def trace_path(self, ray, rng_state, math_mode=MATH_LIBM):
        st = C.c_uint32(rng_state)
        out = np.zeros(3, np.float32)
        lib().orc_trace_path(self.h, math_mode, f32(ray), C.byref(st), out)
        return out, st.value


def bmp_bytes(bgr):
    """The BMP file image the checker's writer produces for a bgr[h,w,3] array."""
    import tempfile
    h, w, _ = bgr.shape
    with tempfile.NamedTemporaryFile(suffix=".bmp", delete=False) as f:
        path = f.name
    try:
        assert lib().orc_write_bmp(path.encode(), w, h, np.ascontiguousarray(bgr)) == 0
        with open(path, "rb") as f:
            return f.read()
    finally:
        os.unlink(path)


def read_bmp(path_or_bytes):
    """24-bit bottom-up BMP -> bgr[h,w,3] uint8 (top row first)."""
    data = path_or_bytes
    if not isinstance(data, (bytes, bytearray)):
        with open(path_or_bytes, "rb") as f:
            data = f.read()
    w = int.from_bytes(data[18:22], "little")
    h = int.from_bytes(data[22:26], "little")
    stride = (w * 3 + 3) // 4 * 4
    rows = np.frombuffer(data, np.uint8, count=stride * h, offset=54).reshape(h, stride)
    return rows[::-1, : w * 3].reshape(h, w, 3).copy()


def sample_bsdf(v, mat, rng_state, math_mode=MATH_LIBM):
    st = C.c_uint32(rng_state)
    out = np.zeros(7, np.float32)
    lib().orc_sample_bsdf(math_mode, f32(v), f32(mat), C.byref(st), out)
    return out, st.value


def eval_bsdf(v, l, mat, math_mode=MATH_LIBM):
    out = np.zeros(4, np.float32)
    lib().orc_eval_bsdf(math_mode, f32(v), f32(l), f32(mat), out)
    return out


def pmath(which, x, y=None):
    x = f32(x)
    y = f32(y) if y is not None else np.zeros_like(x)
    out = np.empty_like(x)
    lib().orc_pmath({"sin": 0, "cos": 1, "log": 2, "pow": 3, "sincos_s": 4, "sincos_c": 5}[which], x, y, out, x.size)
    return out


# ---------------------------------------------------------------- reference

_ref = None


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    """The real reference, compiled from its own sources (build container only)."""
    global _ref
    if _ref is None:
        r = C.CDLL(REF_SO)
        r.ref_run.argtypes = [C.c_char_p]
        r.ref_load.argtypes = [C.c_char_p]
        r.ref_info.argtypes = [_i32p]
        for name in ("ref_objects", "ref_nodes", "ref_materials", "ref_lights"):
            getattr(r, name).restype = C.c_void_p
        r.ref_nearest.argtypes = [_f32p, C.c_int64, _i32p, _f32p, _f32p, _f32p, _f32p, _f32p]
        r.ref_occluded.argtypes = [_f32p, _f32p, C.c_int64, _u8p]
        r.ref_trace_path.argtypes = [_f32p, C.POINTER(C.c_uint32), _f32p]
        r.ref_sample_bsdf.argtypes = [_f32p, _f32p, C.POINTER(C.c_uint32), _f32p]
        r.ref_eval_bsdf.argtypes = [_f32p, _f32p, _f32p, _f32p]
        r.ref_tonemap.restype = C.c_float
        r.ref_tonemap.argtypes = [C.c_float, C.c_float]
        r.ref_rng_next.restype = C.c_uint32
        r.ref_rng_next.argtypes = [C.POINTER(C.c_uint32)]
        r.ref_rng_float.restype = C.c_float
        r.ref_rng_float.argtypes = [C.POINTER(C.c_uint32)]
        r.ref_random_triangle.argtypes = [C.POINTER(C.c_uint32), _f32p]
        _ref = r
    return _ref


def ref_array(getter, dtype, n):
    ptr = getattr(ref(), getter)()
    buf = (C.c_char * (dtype.itemsize * n)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


def ref_info():
    out = np.zeros(8, np.int32)
    ref().ref_info(out)
    keys = ("objects", "nodes", "materials", "lights", "width", "height", "samples", "sizeof_object")
    return dict(zip(keys, (int(x) for x in out)))


def scene_variant(name, width=None, height=None, samples=None):
    """Text of scenes/<name> with only the width/height/samples tokens of its
    render block replaced."""
    import re
    with open(os.path.join(SCENES, name)) as f:
        text = f.read()
    head, sep, tail = text.partition("}")
    for key, val in (("width", width), ("height", height), ("samples", samples)):
        if val is not None:
            head, n = re.subn(rf"(\b{key}\s+)\d+", rf"\g<1>{val}", head, count=1)
            assert n == 1, key
    return head + sep + tail
