"""ctypes binding of libmi_pathtrace.so (the C ABI in include/mi_pathtrace.h).

The host-side mirror of the reference's pathtrace surface (src/pathtrace.h:8-11):
``Scene`` wraps the scene loader / assembler, ``Tracer`` wraps
pathtraceInit / pathtrace / pathtraceFree.  Nothing here computes: every call
goes through the shared library, and importing fails loudly when it is missing.
"""
import ctypes as C
import os

import numpy as np

try:
    # One HIP runtime per process: torch ships its own libamdhip64.so.7; loading it first makes the
    # dynamic linker bind libmi_pathtrace.so to that same copy (equal SONAME) instead of opening a second
    # runtime from /opt/rocm, which cannot see the device once another runtime holds it.
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI_PATHTRACE_LIB") or os.path.join(_HERE, "csrc", "libmi_pathtrace.so")

PT_SPHERE, PT_CUBE, PT_GLTF_MESH = 0, 1, 2
PT_DIFFUSE, PT_SPECULAR, PT_REFRACTIVE, PT_PBR, PT_EMITTING = 0, 1, 2, 3, 4
PT_OPT_MATERIAL_SORT, PT_OPT_USE_BVH, PT_OPT_STAGE_TIMING, PT_OPT_COUNTERS, PT_OPT_USE_GRAPH, PT_OPT_BATCH, PT_OPT_LAUNCH_STREAMS = 1, 2, 3, 4, 5, 6, 7
PT_MAX_DEPTH = 64
PT_NOALLOC_MAX_RAYS = 262144


class PtGeom(C.Structure):
    _fields_ = [("type", C.c_int32), ("materialid", C.c_int32),
                ("translation", C.c_float * 3), ("rotation", C.c_float * 3), ("scale", C.c_float * 3),
                ("transform", C.c_float * 16), ("inverseTransform", C.c_float * 16), ("invTranspose", C.c_float * 16),
                ("meshIndex", C.c_int32), ("triangleStart", C.c_int32), ("triangleCount", C.c_int32),
                ("bvhIndex", C.c_int32)]


class PtMaterial(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("specularExponent", C.c_float), ("specularColor", C.c_float * 3),
                ("transparency", C.c_float), ("roughness", C.c_float), ("metallic", C.c_float),
                ("indexOfRefraction", C.c_float), ("emittance", C.c_float), ("type", C.c_int32),
                ("baseColorTextureIdx", C.c_int32), ("metallicRoughnessTextureIdx", C.c_int32),
                ("normalTextureIdx", C.c_int32), ("emissiveTextureIdx", C.c_int32), ("occlusionTextureIdx", C.c_int32),
                ("emissiveFactor", C.c_float * 3), ("subsurfaceColor", C.c_float * 3), ("subsurfaceRadius", C.c_float),
                ("subsurfaceRadiusRGB", C.c_float * 3), ("subsurfaceScale", C.c_float),
                ("subsurfaceAnisotropy", C.c_float), ("subsurfaceEnabled", C.c_int32)]


TRI_DTYPE = np.dtype([("v0", "<f4", 3), ("v1", "<f4", 3), ("v2", "<f4", 3),
                      ("n0", "<f4", 3), ("n1", "<f4", 3), ("n2", "<f4", 3),
                      ("uv0", "<f4", 2), ("uv1", "<f4", 2), ("uv2", "<f4", 2),
                      ("materialId", "<i4")])
NODE_DTYPE = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("leftChild", "<i4"), ("rightChild", "<i4"),
                       ("triangleStart", "<i4"), ("triangleCount", "<i4")])
HIT_DTYPE = np.dtype([("t", "<f4"), ("geom", "<i4"), ("triangle", "<i4"), ("materialId", "<i4"),
                      ("normal", "<f4", 3), ("uv", "<f4", 2)])
assert TRI_DTYPE.itemsize == 100 and NODE_DTYPE.itemsize == 40 and HIT_DTYPE.itemsize == 36


class PtMesh(C.Structure):
    _fields_ = [("triangles", C.c_void_p), ("triangleCount", C.c_int32),
                ("nodes", C.c_void_p), ("nodeCount", C.c_int32),
                ("triangleIndices", C.c_void_p), ("triangleIndexCount", C.c_int32)]


class PtTexture(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("components", C.c_int32)]


class PtCamera(C.Structure):
    _fields_ = [("resolution", C.c_int32 * 2), ("position", C.c_float * 3), ("lookAt", C.c_float * 3),
                ("view", C.c_float * 3), ("up", C.c_float * 3), ("right", C.c_float * 3),
                ("fov", C.c_float * 2), ("pixelLength", C.c_float * 2)]


class PtEnvMap(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32),
                ("intensity", C.c_float), ("enabled", C.c_int32)]


class PtSceneDesc(C.Structure):
    _fields_ = [("geoms", C.POINTER(PtGeom)), ("geomCount", C.c_int32),
                ("materials", C.POINTER(PtMaterial)), ("materialCount", C.c_int32),
                ("meshes", C.POINTER(PtMesh)), ("meshCount", C.c_int32),
                ("textures", C.POINTER(PtTexture)), ("textureCount", C.c_int32),
                ("env", PtEnvMap), ("camera", PtCamera),
                ("traceDepth", C.c_int32), ("iterations", C.c_int32)]


class PtStats(C.Structure):
    _fields_ = [("pathBounces", C.c_uint64), ("iterations", C.c_uint64),
                ("lastLive", C.c_uint32 * PT_MAX_DEPTH), ("lastDepth", C.c_uint32),
                ("nodesVisited", C.c_uint64), ("trisTested", C.c_uint64), ("meshHits", C.c_uint64),
                ("msIntersect", C.c_float), ("msSort", C.c_float), ("msShade", C.c_float),
                ("msCompact", C.c_float), ("msCamera", C.c_float),
                ("launchesIntersect", C.c_uint32), ("launchesShade", C.c_uint32),
                ("shadeNodesVisited", C.c_uint64), ("shadeTrisTested", C.c_uint64), ("shadeMeshHits", C.c_uint64),
                ("msTrace", C.c_float), ("msShadowTrace", C.c_float),
                ("launchesTrace", C.c_uint32), ("launchesShadowTrace", C.c_uint32),
                ("tracedRays", C.c_uint64), ("shadowRays", C.c_uint64),
                ("boxesTested", C.c_uint64), ("shadeBoxesTested", C.c_uint64),
                ("maxStackDepth", C.c_uint32), ("ldsStackEntries", C.c_uint32), ("referenceOrderWalks", C.c_uint64)]


class PtError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libmi_pathtrace.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C cuda-path-tracer_amd/csrc`" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, ip, fp = C.c_void_p, C.c_int, C.c_float
    sig = {
        "pt_create": (ip, [C.POINTER(PtSceneDesc), ip, C.POINTER(vp)]),
        "pt_destroy": (ip, [vp]),
        "pt_trace": (ip, [vp, ip]),
        "pt_trace_range": (ip, [vp, ip, ip, ip]),
        "pt_trace_lookahead": (ip, [vp, ip, ip, ip]),
        "pt_bind_host_image": (ip, [vp, vp]),
        "pt_reduce_sum": (ip, [C.POINTER(vp), ip, ip]),
        "pt_device_count": (ip, []),
        "pt_host_image_wait": (ip, [vp]),
        "pt_display": (ip, [vp, ip, vp]),
        "pt_download": (ip, [vp, vp]),
        "pt_bind_image": (ip, [vp, vp]),
        "pt_upload_image": (ip, [vp, vp]),
        "pt_clear": (ip, [vp]),
        "pt_sync": (ip, [vp]),
        "pt_set_option": (ip, [vp, ip, ip]),
        "pt_capture_gbuffer": (ip, [vp, ip, vp, vp]),
        "pt_denoise": (ip, [vp, fp, ip, fp, fp, fp, vp]),
        "pt_stats": (ip, [vp, C.POINTER(PtStats)]),
        "pt_reset_stats": (ip, [vp]),
        "pt_stream": (vp, [vp]),
        "pt_intersect_rays": (ip, [vp, vp, vp, ip, vp]),
        "pt_intersect_rays_noalloc": (ip, [vp, vp, vp, ip, vp]),
        "pt_camera_rays": (ip, [vp, ip, vp, vp]),
        "pt_math_eval": (ip, [ip, ip, vp, vp, ip, vp]),
        "pt_last_error": (C.c_char_p, []),
        "pt_version": (C.c_char_p, []),
        "pt_resource_calls": (C.c_ulonglong, []),
        "pt_set_tree_passes": (C.c_int, [C.c_int]),
        "pt_set_camera": (C.c_int, [C.c_void_p, C.c_void_p]),
        "pt_scene_fingerprint": (C.c_ulonglong, [C.c_void_p]),
        "pt_bvh_build": (ip, [vp, ip, ip, C.POINTER(vp)]),
        "pt_bvh_nodes": (vp, [vp, C.POINTER(ip)]),
        "pt_bvh_indices": (vp, [vp, C.POINTER(ip)]),
        "pt_bvh_stats": (ip, [vp, C.POINTER(ip), C.POINTER(ip), C.POINTER(ip)]),
        "pt_bvh_free": (None, [vp]),
        "pt_scene_load": (ip, [C.c_char_p, C.POINTER(vp)]),
        "pt_scene_desc": (C.POINTER(PtSceneDesc), [vp]),
        "pt_scene_set_resolution": (ip, [vp, ip, ip]),
        "pt_scene_set_depth": (ip, [vp, ip]),
        "pt_scene_set_iterations": (ip, [vp, ip]),
        "pt_scene_add_mesh": (ip, [vp, vp, ip, C.POINTER(fp), C.POINTER(fp), C.POINTER(fp)]),
        "pt_scene_import_gltf": (ip, [vp, C.c_char_p, C.POINTER(fp), C.POINTER(fp), C.POINTER(fp)]),
        "pt_scene_log": (C.c_char_p, [vp]),
        "pt_decode_png": (ip, [vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "pt_decode_jpeg": (ip, [vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "pt_image_to_rgb8": (ip, [vp, ip, ip, fp, vp]),
        "pt_image_average": (ip, [vp, ip, ip, fp, ip, vp]),
        "pt_write_png": (ip, [C.c_char_p, vp, ip, ip, ip]),
        "pt_write_hdr": (ip, [C.c_char_p, vp, ip, ip]),
        "pt_scene_add_material": (ip, [vp, C.POINTER(PtMaterial)]),
        "pt_scene_add_texture": (ip, [vp, vp, ip, ip, ip]),
        "pt_scene_set_envmap": (ip, [vp, vp, ip, ip, fp]),
        "pt_scene_free": (None, [vp]),
        "pt_scene_image": (C.POINTER(C.c_float), [vp]),
        "pt_scene_iterations": (C.c_int, [vp]),
        "pt_scene_image_name": (C.c_char_p, [vp]),
        "pt_build_transforms": (None, [C.POINTER(fp)] * 6),
        "pt_bvh_device_layout": (C.c_int, [vp, C.c_int, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)   # AttributeError here = the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib, sorted(sig)


lib, EXPORTED = _load()


def _check(rc, what):
    if rc < 0:
        raise PtError("%s failed (%d): %s" % (what, rc, lib.pt_last_error().decode()))
    return rc


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def build_transforms(t, r, s):
    out = [(C.c_float * 16)() for _ in range(3)]
    lib.pt_build_transforms(_f3(t), _f3(r), _f3(s), out[0], out[1], out[2])
    return [np.array(o, dtype=np.float32).reshape(4, 4) for o in out]   # [col][row]


def bvh_build(tris, max_depth=24):
    """CPU BVH build (role of BVH::build, src/bvh.cpp:30).  Returns (nodes, indices, stats)."""
    tris = np.ascontiguousarray(tris, dtype=TRI_DTYPE)
    h = C.c_void_p()
    _check(lib.pt_bvh_build(tris.ctypes.data, len(tris), max_depth, C.byref(h)), "pt_bvh_build")
    try:
        n = C.c_int()
        p = lib.pt_bvh_nodes(h, C.byref(n))
        nodes = np.frombuffer(C.string_at(p, n.value * 40), dtype=NODE_DTYPE).copy() if n.value else np.zeros(0, NODE_DTYPE)
        p = lib.pt_bvh_indices(h, C.byref(n))
        idx = np.frombuffer(C.string_at(p, n.value * 4), dtype=np.int32).copy() if n.value else np.zeros(0, np.int32)
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        lib.pt_bvh_stats(h, C.byref(a), C.byref(b), C.byref(c))
        return nodes, idx, {"nodes": a.value, "leaves": b.value, "max_depth": c.value}
    finally:
        lib.pt_bvh_free(h)


class PtLayoutStats(C.Structure):
    _fields_ = [("wideNodes", C.c_int32), ("wideDepth", C.c_int32), ("leaves", C.c_int32), ("trianglesReferenced", C.c_int32),
                ("sahCostHost", C.c_float), ("sahCostDevice", C.c_float)]


def bvh_device_layout(tris, max_depth=24):
    """Build the host BVH, derive the device layout from it on the CPU and validate it (pt_bvh_device_layout)."""
    tris = np.ascontiguousarray(tris, dtype=TRI_DTYPE)
    h = C.c_void_p()
    _check(lib.pt_bvh_build(tris.ctypes.data, len(tris), max_depth, C.byref(h)), "pt_bvh_build")
    try:
        st = PtLayoutStats()
        _check(lib.pt_bvh_device_layout(tris.ctypes.data, len(tris), h, C.byref(st)), "pt_bvh_device_layout")
        return st
    finally:
        lib.pt_bvh_free(h)


def scene_fingerprint(desc):
    """pt_scene_fingerprint of a PtSceneDesc pointer (Scene.desc): equal for descriptions that differ in the camera's pose only."""
    return int(lib.pt_scene_fingerprint(desc))


def set_tree_passes(passes):
    """Rounds of the walked tree's reinsertion optimiser for trees built from now on (0: none; default 3)."""
    _check(lib.pt_set_tree_passes(int(passes)), "pt_set_tree_passes")


def decode_png(data):
    """PNG bytes -> (H, W, 4) uint8, through the library's own decoder (the texture path of the glTF import)."""
    buf = np.frombuffer(bytes(data), np.uint8)
    w, h = C.c_int(), C.c_int()
    _check(lib.pt_decode_png(buf.ctypes.data, len(buf), None, 0, C.byref(w), C.byref(h)), "pt_decode_png")
    out = np.empty((h.value, w.value, 4), np.uint8)
    _check(lib.pt_decode_png(buf.ctypes.data, len(buf), out.ctypes.data, out.nbytes, C.byref(w), C.byref(h)), "pt_decode_png")
    return out


def decode_jpeg(data):
    """Sequential JPEG bytes -> (H, W, 4) uint8 (the texture path of the glTF import)."""
    buf = np.frombuffer(bytes(data), np.uint8)
    w, h = C.c_int(), C.c_int()
    _check(lib.pt_decode_jpeg(buf.ctypes.data, len(buf), None, 0, C.byref(w), C.byref(h)), "pt_decode_jpeg")
    out = np.empty((h.value, w.value, 4), np.uint8)
    _check(lib.pt_decode_jpeg(buf.ctypes.data, len(buf), out.ctypes.data, out.nbytes, C.byref(w), C.byref(h)), "pt_decode_jpeg")
    return out


def image_to_rgb8(image_sum, samples):
    """Running sum (H, W, 3) -> the 8-bit image the reference saves: x mirrored, averaged, clamped, truncated."""
    a = np.ascontiguousarray(image_sum, np.float32)
    h, w = a.shape[:2]
    out = np.empty((h, w, 3), np.uint8)
    _check(lib.pt_image_to_rgb8(a.ctypes.data, w, h, float(samples), out.ctypes.data), "pt_image_to_rgb8")
    return out


def image_average(image_sum, samples, mirror_x=True):
    a = np.ascontiguousarray(image_sum, np.float32)
    h, w = a.shape[:2]
    out = np.empty((h, w, 3), np.float32)
    _check(lib.pt_image_average(a.ctypes.data, w, h, float(samples), int(mirror_x), out.ctypes.data), "pt_image_average")
    return out


def write_png(path, pixels):
    a = np.ascontiguousarray(pixels, np.uint8)
    h, w = a.shape[:2]
    _check(lib.pt_write_png(os.fspath(path).encode(), a.ctypes.data, w, h, 1 if a.ndim == 2 else a.shape[2]), "pt_write_png")


def write_hdr(path, rgb):
    a = np.ascontiguousarray(rgb, np.float32)
    h, w = a.shape[:2]
    _check(lib.pt_write_hdr(os.fspath(path).encode(), a.ctypes.data, w, h), "pt_write_hdr")


class Scene:
    """Owning scene (PtScene): JSON loader + programmatic assembly; no GPU needed."""

    def __init__(self, json_path):
        self._h = C.c_void_p()
        _check(lib.pt_scene_load(os.fspath(json_path).encode(), C.byref(self._h)), "pt_scene_load")
        self._keep = []

    @property
    def desc(self):
        return lib.pt_scene_desc(self._h)

    @property
    def resolution(self):
        d = self.desc.contents
        return int(d.camera.resolution[0]), int(d.camera.resolution[1])

    @property
    def depth(self):
        return int(self.desc.contents.traceDepth)

    def set_resolution(self, w, h):
        _check(lib.pt_scene_set_resolution(self._h, int(w), int(h)), "pt_scene_set_resolution")
        return self

    def set_depth(self, d):
        _check(lib.pt_scene_set_depth(self._h, int(d)), "pt_scene_set_depth")
        return self

    def add_material(self, **kw):
        m = PtMaterial()
        for k in ("baseColorTextureIdx", "metallicRoughnessTextureIdx", "normalTextureIdx", "emissiveTextureIdx",
                  "occlusionTextureIdx"):
            setattr(m, k, -1)
        m.indexOfRefraction = 1.0
        for k, v in kw.items():
            if isinstance(v, (list, tuple, np.ndarray)):
                setattr(m, k, _f3(v))
            else:
                setattr(m, k, v)
        return _check(lib.pt_scene_add_material(self._h, C.byref(m)), "pt_scene_add_material")

    def add_texture(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape[:2]
        comps = 1 if img.ndim == 2 else img.shape[2]
        return _check(lib.pt_scene_add_texture(self._h, img.ctypes.data, w, h, comps), "pt_scene_add_texture")

    def add_mesh(self, tris, translation=(0, 0, 0), rotation=(0, 0, 0), scale=(1, 1, 1)):
        tris = np.ascontiguousarray(tris, dtype=TRI_DTYPE)
        return _check(lib.pt_scene_add_mesh(self._h, tris.ctypes.data, len(tris), _f3(translation), _f3(rotation),
                                            _f3(scale)), "pt_scene_add_mesh")

    def import_gltf(self, path, translation=(0, 0, 0), rotation=(0, 0, 0), scale=(1, 1, 1)):
        """Import a .glb / .gltf file as one mesh object (textures, materials, triangles); returns the geom index."""
        return _check(lib.pt_scene_import_gltf(self._h, os.fspath(path).encode(), _f3(translation), _f3(rotation),
                                               _f3(scale)), "pt_scene_import_gltf")

    @property
    def log(self):
        return lib.pt_scene_log(self._h).decode()

    def set_envmap(self, rgb, intensity=1.0):
        rgb = np.ascontiguousarray(rgb, dtype=np.float32)
        h, w = rgb.shape[:2]
        _check(lib.pt_scene_set_envmap(self._h, rgb.ctypes.data, w, h, float(intensity)), "pt_scene_set_envmap")
        return self

    def materials(self):
        d = self.desc.contents
        return [d.materials[i] for i in range(d.materialCount)]

    def geoms(self):
        d = self.desc.contents
        return [d.geoms[i] for i in range(d.geomCount)]

    def mesh_triangles(self, mesh_index):
        """Copy of one mesh's triangle records (TRI_DTYPE)."""
        d = self.desc.contents
        if not 0 <= mesh_index < d.meshCount:
            raise IndexError(mesh_index)
        m = d.meshes[mesh_index]
        n = int(m.triangleCount)
        if n == 0:
            return np.zeros(0, TRI_DTYPE)
        return np.frombuffer(C.string_at(m.triangles, n * TRI_DTYPE.itemsize), TRI_DTYPE).copy()

    def textures(self):
        """Copies of the texture table as (H, W, components) uint8 arrays."""
        d = self.desc.contents
        out = []
        for i in range(d.textureCount):
            t = d.textures[i]
            n = int(t.width) * int(t.height) * int(t.components)
            out.append(np.frombuffer(C.string_at(t.data, n), np.uint8).reshape(t.height, t.width, t.components).copy())
        return out

    def close(self):
        if self._h:
            lib.pt_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Tracer:
    """pathtraceInit / pathtrace / pathtraceFree over the C ABI (one context per GPU)."""

    def __init__(self, scene, device=0):
        self._h = C.c_void_p()
        self.scene = scene
        desc = scene.desc if isinstance(scene, Scene) else scene
        _check(lib.pt_create(desc, int(device), C.byref(self._h)), "pt_create")
        d = desc.contents
        self.width, self.height = int(d.camera.resolution[0]), int(d.camera.resolution[1])

    def trace(self, it):
        _check(lib.pt_trace(self._h, int(it)), "pt_trace")

    def trace_range(self, begin, end, stride=1):
        _check(lib.pt_trace_range(self._h, int(begin), int(end), int(stride)), "pt_trace_range")

    def trace_lookahead(self, it, ahead, last=0):
        """pathtrace() once per frame at the rate of launch groups: see pt_trace_lookahead."""
        _check(lib.pt_trace_lookahead(self._h, int(it), int(ahead), int(last)), "pt_trace_lookahead")

    def bind_host_image(self, array):
        """array: C-contiguous float32 of W*H*3 elements that receives the running sum after every trace_lookahead call
        (None unbinds); host_image_wait() blocks until the latest copy has landed."""
        if array is None:
            self._host_image = None
            _check(lib.pt_bind_host_image(self._h, None), "pt_bind_host_image")
            return
        assert array.dtype == np.float32 and array.flags["C_CONTIGUOUS"] and array.size == self.width * self.height * 3
        self._host_image = array            # kept alive while bound
        _check(lib.pt_bind_host_image(self._h, array.ctypes.data), "pt_bind_host_image")

    def host_image_wait(self):
        _check(lib.pt_host_image_wait(self._h), "pt_host_image_wait")

    def sync(self):
        _check(lib.pt_sync(self._h), "pt_sync")

    def clear(self):
        _check(lib.pt_clear(self._h), "pt_clear")

    def set_camera(self, camera):
        """A new camera (PtCamera, same resolution) for the iterations traced from now on; the sum is not cleared."""
        _check(lib.pt_set_camera(self._h, C.byref(camera)), "pt_set_camera")

    def set_option(self, opt, value):
        _check(lib.pt_set_option(self._h, int(opt), int(value)), "pt_set_option")

    def bind_image(self, dev_ptr):
        _check(lib.pt_bind_image(self._h, C.c_void_p(dev_ptr)), "pt_bind_image")

    def upload_image(self, img):
        img = np.ascontiguousarray(img, dtype=np.float32)
        assert img.size == self.width * self.height * 3
        _check(lib.pt_upload_image(self._h, img.ctypes.data), "pt_upload_image")

    def image(self):
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        _check(lib.pt_download(self._h, out.ctypes.data), "pt_download")
        return out

    def display(self, it, dev_rgba_ptr):
        _check(lib.pt_display(self._h, int(it), C.c_void_p(dev_rgba_ptr)), "pt_display")

    def stats(self):
        s = PtStats()
        _check(lib.pt_stats(self._h, C.byref(s)), "pt_stats")
        return s

    def capture_gbuffer(self, iteration):
        """First-hit normals and base colours of the camera rays of one iteration: two (H, W, 3) float32 arrays."""
        w, h = self.width, self.height
        n = np.empty((h, w, 3), np.float32)
        a = np.empty((h, w, 3), np.float32)
        _check(lib.pt_capture_gbuffer(self._h, int(iteration), n.ctypes.data, a.ctypes.data), "pt_capture_gbuffer")
        return n, a

    def denoise(self, samples, levels=5, sigma_color=0.6, sigma_normal=0.3, sigma_albedo=0.3):
        """A-trous filtered (sum / samples) guided by the captured buffers: (H, W, 3) float32."""
        w, h = self.width, self.height
        out = np.empty((h, w, 3), np.float32)
        _check(lib.pt_denoise(self._h, float(samples), int(levels), float(sigma_color), float(sigma_normal), float(sigma_albedo),
                              out.ctypes.data), "pt_denoise")
        return out

    def reset_stats(self):
        _check(lib.pt_reset_stats(self._h), "pt_reset_stats")

    def stream(self):
        return lib.pt_stream(self._h)

    def intersect(self, origins, directions):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        assert o.shape == d.shape
        out = np.zeros(len(o), dtype=HIT_DTYPE)
        _check(lib.pt_intersect_rays(self._h, o.ctypes.data, d.ctypes.data, len(o), out.ctypes.data), "pt_intersect_rays")
        return out

    def intersect_noalloc(self, origins, directions, out=None):
        """pt_intersect_rays_noalloc: at most PT_NOALLOC_MAX_RAYS rays, staging in the library's static device array."""
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        assert o.shape == d.shape
        if out is None:
            out = np.zeros(len(o), dtype=HIT_DTYPE)
        _check(lib.pt_intersect_rays_noalloc(self._h, o.ctypes.data, d.ctypes.data, len(o), out.ctypes.data),
               "pt_intersect_rays_noalloc")
        return out

    def camera_rays(self, it):
        n = self.width * self.height
        o = np.empty((n, 3), np.float32)
        d = np.empty((n, 3), np.float32)
        _check(lib.pt_camera_rays(self._h, int(it), o.ctypes.data, d.ctypes.data), "pt_camera_rays")
        return o, d

    def close(self):
        if self._h:
            lib.pt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def reduce_sum(tracers, root=0):
    """pt_reduce_sum: fold the running sums of several Tracers (sample split in one process) into tracers[root]."""
    arr = (C.c_void_p * len(tracers))(*[t._h.value for t in tracers])
    _check(lib.pt_reduce_sum(arr, len(tracers), int(root)), "pt_reduce_sum")


def math_eval(fn, a, b=None, device=0):
    a = np.ascontiguousarray(a, dtype=np.float32)
    out = np.empty_like(a)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(b, dtype=np.float32)
        bp = b.ctypes.data
    _check(lib.pt_math_eval(int(device), int(fn), a.ctypes.data, bp, a.size, out.ctypes.data), "pt_math_eval")
    return out
