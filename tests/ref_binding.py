"""Access to oracle/_ref/ref_tool — the reference's own host-buildable sources (scene loader, BVH
builder, intersection tests, stb image codecs, Image writer) compiled where they lie under
/root/reference by oracle/Makefile.  Test infrastructure only: the checker of the checker.

The tool is a separate process (the reference's loader prints to stdout and calls exit() on bad
input); requests and answers travel through files in a temporary directory."""
import os
import struct
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "oracle", "_ref", "ref_tool")
REFERENCE = "/root/reference"

REF_HIT_DTYPE = np.dtype([("t", "<f4"), ("geom", "<i4"), ("materialId", "<i4"), ("normal", "<f4", 3), ("uv", "<f4", 2)])
GEOM_DTYPE = np.dtype([("type", "<i4"), ("materialid", "<i4"), ("translation", "<f4", 3), ("rotation", "<f4", 3),
                       ("scale", "<f4", 3), ("transform", "<f4", 16), ("inverseTransform", "<f4", 16),
                       ("invTranspose", "<f4", 16), ("meshIndex", "<i4"), ("triangleCount", "<i4")])
MATERIAL_DTYPE = np.dtype([("color", "<f4", 3), ("specularExponent", "<f4"), ("specularColor", "<f4", 3),
                           ("transparency", "<f4"), ("roughness", "<f4"), ("metallic", "<f4"), ("indexOfRefraction", "<f4"),
                           ("emittance", "<f4"), ("type", "<i4"), ("baseColorTextureIdx", "<i4"),
                           ("metallicRoughnessTextureIdx", "<i4"), ("normalTextureIdx", "<i4"), ("emissiveTextureIdx", "<i4"),
                           ("occlusionTextureIdx", "<i4"), ("emissiveFactor", "<f4", 3), ("subsurfaceColor", "<f4", 3),
                           ("subsurfaceRadius", "<f4"), ("subsurfaceRadiusRGB", "<f4", 3), ("subsurfaceScale", "<f4"),
                           ("subsurfaceAnisotropy", "<f4"), ("subsurfaceEnabled", "<i4")])
CAMERA_DTYPE = np.dtype([("resolution", "<i4", 2), ("position", "<f4", 3), ("lookAt", "<f4", 3), ("view", "<f4", 3),
                         ("up", "<f4", 3), ("right", "<f4", 3), ("fov", "<f4", 2), ("pixelLength", "<f4", 2)])
assert GEOM_DTYPE.itemsize == 244 and MATERIAL_DTYPE.itemsize == 124 and CAMERA_DTYPE.itemsize == 84
# the reference enumerates SPHERE, CUBE, GLTF_MESH and DIFFUSE, SPECULAR, REFRACTIVE, PBR, EMITTING in the
# order of include/mi_pathtrace.h (src/sceneStructs.h:16-30)


def available():
    return os.path.exists(TOOL)


def _run(*args):
    env = dict(os.environ)
    env.pop("LD_BIND_NOW", None)          # the three CUDA runtime symbols of BVH::uploadToGPU stay unresolved
    p = subprocess.run([TOOL] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    if p.returncode != 0:
        raise RuntimeError("ref_tool %s failed (%d): %s" % (args[0], p.returncode, p.stderr.decode()[-400:]))
    return p.stdout.decode(errors="replace")


class _Reader:
    def __init__(self, data):
        self.b, self.at = data, 0

    def i32(self):
        v = struct.unpack_from("<i", self.b, self.at)[0]
        self.at += 4
        return v

    def f32(self):
        v = struct.unpack_from("<f", self.b, self.at)[0]
        self.at += 4
        return v

    def arr(self, dtype, n):
        dtype = np.dtype(dtype)
        a = np.frombuffer(self.b, dtype, n, self.at).copy()
        self.at += n * dtype.itemsize
        return a


def load_scene(json_path, tri_dtype):
    """Scene::Scene(json) -> dict of tables (geoms, materials, meshes, textures, env, camera, ...)."""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "scene.bin")
        log = _run("scene", json_path, out)
        r = _Reader(open(out, "rb").read())
    assert r.i32() == 0x52454631
    ng, nm, nmesh, ntex = r.i32(), r.i32(), r.i32(), r.i32()
    s = dict(log=log)
    s["geoms"] = r.arr(GEOM_DTYPE, ng)
    s["materials"] = r.arr(MATERIAL_DTYPE, nm)
    s["meshes"] = [r.arr(tri_dtype, r.i32()) for _ in range(nmesh)]
    s["textures"] = []
    for _ in range(ntex):
        w, h, c = r.i32(), r.i32(), r.i32()
        s["textures"].append(r.arr(np.uint8, w * h * c).reshape(h, w, c))
    s["env"] = None
    if r.i32():
        w, h, inten = r.i32(), r.i32(), r.f32()
        s["env"] = dict(width=w, height=h, intensity=inten, data=r.arr(np.float32, w * h * 3).reshape(h, w, 3))
    s["camera"] = r.arr(CAMERA_DTYPE, 1)[0]
    s["iterations"], s["traceDepth"] = r.i32(), r.i32()
    n = r.i32()
    s["imageName"] = bytes(r.b[r.at:r.at + n]).decode()
    return s


def bvh_build(tris, node_dtype, max_depth=24):
    """BVH::build(triangles, max_depth) -> (nodes, indices, stats)."""
    with tempfile.TemporaryDirectory() as d:
        tp, out = os.path.join(d, "tris.bin"), os.path.join(d, "bvh.bin")
        np.ascontiguousarray(tris).tofile(tp)
        _run("bvh", tp, max_depth, out)
        r = _Reader(open(out, "rb").read())
    nn, ni, nodes, leaves, depth = r.i32(), r.i32(), r.i32(), r.i32(), r.i32()
    return r.arr(node_dtype, nn), r.arr(np.int32, ni), {"nodes": nodes, "leaves": leaves, "max_depth": depth}


def _desc_bytes(desc, tri_itemsize):
    """PtSceneDesc -> the geometry part ref_tool's isect mode reads."""
    import ctypes as C
    d = desc.contents
    parts = [struct.pack("<ii", d.geomCount, d.meshCount)]
    for i in range(d.geomCount):
        g = d.geoms[i]
        parts.append(struct.pack("<ii", g.type, g.materialid))
        for f in ("translation", "rotation", "scale", "transform", "inverseTransform", "invTranspose"):
            parts.append(bytes(getattr(g, f)))
        parts.append(struct.pack("<i", g.meshIndex))
    for m in range(d.meshCount):
        mesh = d.meshes[m]
        n = int(mesh.triangleCount)
        parts.append(struct.pack("<i", n))
        parts.append(C.string_at(mesh.triangles, n * tri_itemsize) if n else b"")
    return b"".join(parts)


def _rays_file(d, origins, directions):
    o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
    dd = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
    path = os.path.join(d, "rays.bin")
    np.concatenate([o, dd], 1).astype(np.float32).tofile(path)
    return path, len(o)


def intersect(scene, origins, directions, tri_itemsize=100, use_bvh=True, timing=None):
    """Closest hits through the reference's boxIntersectionTest / sphereIntersectionTest /
    meshIntersectionTestBVH (BVHs from the reference's BVH::build) for a scene of this repository.
    timing: a dict that receives rays / seconds / threads of the closest-hit loop alone."""
    desc = scene.desc if hasattr(scene, "desc") else scene
    with tempfile.TemporaryDirectory() as d:
        dp, out = os.path.join(d, "desc.bin"), os.path.join(d, "hits.bin")
        open(dp, "wb").write(_desc_bytes(desc, tri_itemsize))
        rp, n = _rays_file(d, origins, directions)
        log = _run(*(["isect", dp, rp, out] + ([] if use_bvh else ["nobvh"])))
        if timing is not None:
            for line in log.splitlines():
                if line.startswith("ref_tool isect:"):
                    w = line.split()
                    timing.update(rays=int(w[2]), seconds=float(w[5]), threads=int(w[8]))
        return np.fromfile(out, REF_HIT_DTYPE, n)


def intersect_json(json_path, origins, directions):
    """The same with the scene loaded by the reference's own loader."""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "hits.bin")
        rp, n = _rays_file(d, origins, directions)
        _run("isect-json", json_path, rp, out)
        return np.fromfile(out, REF_HIT_DTYPE, n)


def stbi(path):
    """stb_image on a file: ('ldr', native_components, rgba8 (H,W,4), float rgb (H,W,3)) or ('hdr', comps, float rgb)."""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "img.bin")
        _run("stbi", path, out)
        r = _Reader(open(out, "rb").read())
    kind = r.i32()
    if kind == 0:
        return None
    w, h, c = r.i32(), r.i32(), r.i32()
    if kind == 2:
        return "hdr", c, r.arr(np.float32, w * h * 3).reshape(h, w, 3)
    rgba = r.arr(np.uint8, w * h * 4).reshape(h, w, 4)
    return "ldr", c, rgba, r.arr(np.float32, w * h * 3).reshape(h, w, 3)


def save_image(kind, pixels, base):
    """Image::setPixel for every pixel + savePNG / saveHDR -> base + '.png' / '.hdr'."""
    a = np.ascontiguousarray(pixels, np.float32)
    h, w = a.shape[:2]
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "pix.bin")
        a.tofile(p)
        _run(kind, w, h, p, base)
    return base + "." + kind
