"""ctypes access to the CPU oracle (oracle/liboracle.so) — test infrastructure only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "liboracle.so")
MATH_LIBM, MATH_DET = 0, 1
OPT_SCATTER_RIGHT_TO_LEFT = 100      # oracle-only, see oracle/pt_oracle.h
FN = dict(sin=0, cos=1, acos=2, atan2=3, exp=4, log=5, pow5=6, div=7, sqrt=8)

_lib = C.CDLL(LIB)
vp, ip = C.c_void_p, C.c_int
_lib.oracle_create.restype = vp
_lib.oracle_create.argtypes = [vp]
_lib.oracle_destroy.argtypes = [vp]
_lib.oracle_set_math.argtypes = [vp, ip]
_lib.oracle_set_option.argtypes = [vp, ip, ip]
_lib.oracle_set_threads.argtypes = [ip]
_lib.oracle_threads.restype = ip
_lib.oracle_trace.argtypes = [vp, ip]
_lib.oracle_clear.argtypes = [vp]
_lib.oracle_image.argtypes = [vp, vp]
_lib.oracle_display.argtypes = [vp, ip, vp]
_lib.oracle_stats.argtypes = [vp, vp]
_lib.oracle_camera_rays.argtypes = [vp, ip, vp, vp]
_lib.oracle_intersect_rays.argtypes = [vp, vp, vp, ip, vp]
_lib.oracle_intersect_rays_linear.argtypes = [vp, vp, vp, ip, vp]
_lib.oracle_rng_u01.argtypes = [ip, ip, ip, ip, vp]
_lib.oracle_hemisphere.argtypes = [vp, ip, ip, ip, ip, vp, vp]
_lib.oracle_math_eval.restype = C.c_float
_lib.oracle_math_eval.argtypes = [ip, ip, C.c_float, C.c_float]
_lib.oracle_math_eval_array.argtypes = [ip, ip, vp, vp, ip, vp]
_lib.oracle_bvh_build.argtypes = [vp, ip, ip, C.POINTER(vp), C.POINTER(ip), C.POINTER(vp), C.POINTER(ip),
                                  C.POINTER(ip), C.POINTER(ip)]
_lib.oracle_free.argtypes = [vp]


def set_threads(n):
    _lib.oracle_set_threads(int(n))


def threads():
    return _lib.oracle_threads()


def rng_u01(it, index, depth, n):
    out = np.zeros(n, np.float32)
    _lib.oracle_rng_u01(it, index, depth, n, out.ctypes.data)
    return out


def hemisphere(it, index, depth, skip, normal, oracle=None):
    n = np.asarray(normal, np.float32)
    out = np.zeros(3, np.float32)
    _lib.oracle_hemisphere(oracle._h if oracle else None, it, index, depth, skip, n.ctypes.data, out.ctypes.data)
    return out


def math_eval(mode, fn, a, b=None):
    a = np.ascontiguousarray(a, np.float32)
    out = np.empty_like(a)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(b, np.float32)
        bp = b.ctypes.data
    _lib.oracle_math_eval_array(mode, FN[fn] if isinstance(fn, str) else fn, a.ctypes.data, bp, a.size, out.ctypes.data)
    return out


def bvh_build(pt, tris, max_depth=24):
    tris = np.ascontiguousarray(tris, dtype=pt.TRI_DTYPE)
    nodes, idx = vp(), vp()
    nn, ni, leaves, depth = ip(), ip(), ip(), ip()
    _lib.oracle_bvh_build(tris.ctypes.data, len(tris), max_depth, C.byref(nodes), C.byref(nn), C.byref(idx), C.byref(ni),
                          C.byref(leaves), C.byref(depth))
    try:
        n = np.frombuffer(C.string_at(nodes, nn.value * 40), dtype=pt.NODE_DTYPE).copy()
        i = np.frombuffer(C.string_at(idx, ni.value * 4), dtype=np.int32).copy()
    finally:
        _lib.oracle_free(nodes)
        _lib.oracle_free(idx)
    return n, i, {"nodes": nn.value, "leaves": leaves.value, "max_depth": depth.value}


class Oracle:
    def __init__(self, pt, scene, math=MATH_LIBM):
        self.pt = pt
        self.scene = scene
        desc = scene.desc if hasattr(scene, "desc") else scene
        self._h = _lib.oracle_create(C.cast(desc, vp))
        d = desc.contents
        self.width, self.height = int(d.camera.resolution[0]), int(d.camera.resolution[1])
        self.set_math(math)

    def set_math(self, mode):
        _lib.oracle_set_math(self._h, mode)

    def set_option(self, opt, value):
        _lib.oracle_set_option(self._h, opt, value)

    def trace(self, it):
        _lib.oracle_trace(self._h, int(it))

    def clear(self):
        _lib.oracle_clear(self._h)

    def image(self):
        out = np.empty((self.height, self.width, 3), np.float32)
        _lib.oracle_image(self._h, out.ctypes.data)
        return out

    def display(self, it):
        out = np.empty((self.height, self.width, 4), np.uint8)
        _lib.oracle_display(self._h, int(it), out.ctypes.data)
        return out

    def stats(self):
        s = self.pt.PtStats()
        _lib.oracle_stats(self._h, C.byref(s))
        return s

    def camera_rays(self, it):
        n = self.width * self.height
        o = np.empty((n, 3), np.float32)
        d = np.empty((n, 3), np.float32)
        _lib.oracle_camera_rays(self._h, int(it), o.ctypes.data, d.ctypes.data)
        return o, d

    def intersect(self, origins, directions, linear=False):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        out = np.zeros(len(o), dtype=self.pt.HIT_DTYPE)
        fn = _lib.oracle_intersect_rays_linear if linear else _lib.oracle_intersect_rays
        fn(self._h, o.ctypes.data, d.ctypes.data, len(o), out.ctypes.data)
        return out

    def close(self):
        if self._h:
            _lib.oracle_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
