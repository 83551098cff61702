"""The CPU oracle against the known-answer values recorded from the reference (tests/golden)."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))
REFERENCE_DUCK = "/root/reference/gltf/Duck.glb"


def test_rng_stream(orc):
    k = KAT["rng"]
    u = orc.rng_u01(k["iter"], k["index"], k["depth"], 4)
    assert sorted(np.float32(x) for x in k["first_two_u01_as_set"]) == sorted(u[:2].tolist())
    assert np.all((u >= 0) & (u <= 1))


def test_hemisphere_sample(orc):
    k = KAT["hemisphere"]
    d = orc.hemisphere(k["iter"], k["index"], k["depth"], k["draws_consumed_before"], k["normal"])
    assert np.allclose(d, k["direction"], atol=2e-6)
    assert abs(np.linalg.norm(d) - 1) < 1e-6


def test_unit_sphere_hit(pt, scenes, orc, tmp_path):
    k = KAT["unit_sphere"]
    d = dict(Materials={"m": dict(TYPE="Diffuse", RGB=[1, 1, 1])}, Camera=scenes._camera("x", 8, 1, res=(4, 4)),
             Objects=[scenes._obj("sphere", "m", (0, 0, 0), (0, 0, 0), (1, 1, 1))])
    p = tmp_path / "s.json"
    p.write_text(json.dumps(d))
    sc = pt.Scene(p)
    o = orc.Oracle(pt, sc)
    h = o.intersect([k["origin"]], [k["direction"]])
    assert h["t"][0] == np.float32(k["t"])
    assert np.allclose(h["normal"][0], k["normal"], atol=1e-7)
    assert h["geom"][0] == 0 and h["triangle"][0] == -1


def test_camera_ray(pt, scenes, orc):
    k = KAT["camera"]
    sc = scenes.make_scene("cornell", res=tuple(k["res"]))
    cam = sc.desc.contents.camera
    for name in ("view", "up", "right", "position"):
        for i in range(3):
            getattr(cam, name)[i] = k[name][i]
    cam.pixelLength[0], cam.pixelLength[1] = k["pixelLength"]
    o = orc.Oracle(pt, sc)
    _, d = o.camera_rays(k["iter"])
    px = k["pixel"][0] + k["pixel"][1] * k["res"][0]
    assert np.array_equal(d[px], np.asarray(k["direction"], np.float32))


def test_cornell_800_live_curve_and_radiance(pt, scenes, orc):
    """The strongest pin: every stage (camera, primitives, sort, MIS diffuse, mirror, glass, RR,
    compaction, gather) must reproduce the reference's path counts exactly."""
    k = KAT["cornell_800x800_depth8_no_env"]
    sc = scenes.make_scene("cornell", res=(800, 800), depth=8)
    o = orc.Oracle(pt, sc, orc.MATH_LIBM)
    o.trace(1)
    s = o.stats()
    assert list(s.lastLive[:9]) == k["live_iter1"]
    assert s.pathBounces == sum(k["live_iter1"])
    o.trace(2)
    assert o.stats().pathBounces == k["path_bounces_two_iterations"] == 2 * k["path_bounces_per_spp_mean_of_two_iterations"]
    mean = o.image().reshape(-1, 3).astype(np.float64).mean(0) / 2
    assert np.allclose(mean, k["mean_rgb_per_sample_two_iterations"], atol=6e-7)


def test_struct_sizes(pt):
    k = KAT["struct_sizes"]
    assert pt.TRI_DTYPE.itemsize == k["Triangle"]
    assert pt.NODE_DTYPE.itemsize == k["BVHNode"]
    assert C.sizeof(pt.PtMaterial) == k["Material"]
    assert C.sizeof(pt.PtCamera) == k["Camera"]


def _load_glb_triangles(path, pt):
    """Minimal GLB reader for the KAT only: POSITION / NORMAL / TEXCOORD_0 + indices of every primitive,
    node matrices ignored except the scene's TRS chain (Duck.glb has a single mesh node)."""
    import struct
    raw = open(path, "rb").read()
    magic, ver, total = struct.unpack_from("<III", raw, 0)
    assert magic == 0x46546C67
    off = 12
    chunks = []
    while off < total:
        ln, typ = struct.unpack_from("<II", raw, off)
        chunks.append((typ, raw[off + 8: off + 8 + ln]))
        off += 8 + ln
    doc = json.loads(chunks[0][1].decode())
    blob = chunks[1][1]

    def accessor(i):
        a = doc["accessors"][i]
        bv = doc["bufferViews"][a["bufferView"]]
        comp = {5126: np.float32, 5123: np.uint16, 5125: np.uint32}[a["componentType"]]
        n = {"SCALAR": 1, "VEC2": 2, "VEC3": 3}[a["type"]]
        start = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        return np.frombuffer(blob, comp, a["count"] * n, start).reshape(a["count"], n)

    return doc, accessor


@pytest.mark.skipif(not os.path.exists(REFERENCE_DUCK), reason="reference assets are not present on this machine")
def test_duck_bvh_statistics(pt, scenes, orc):
    """BVH::build on gltf/Duck.glb as scenes/duck.json places it: 4 212 triangles -> 2 669 nodes / 1 335 leaves / depth 15."""
    k = KAT["duck_bvh"]
    doc, accessor = _load_glb_triangles(REFERENCE_DUCK, pt)
    node_chain = []

    def walk(i, parents):
        n = doc["nodes"][i]
        if "mesh" in n:
            node_chain.append((parents + [n], n["mesh"]))
        for c in n.get("children", []):
            walk(c, parents + [n])
    for r in doc["scenes"][doc.get("scene", 0)]["nodes"]:
        walk(r, [])
    trs = ((-0.5, 0.0, 0.0), (0.0, -22.5, 0.0), (2.5, 2.5, 2.5))
    M = pt.build_transforms(*trs)[0].T.astype(np.float32)       # row-major 4x4
    tris = []
    for chain, mesh in node_chain:
        T = M.copy()
        for n in chain:
            if "matrix" in n:
                L = np.asarray(n["matrix"], np.float32).reshape(4, 4).T
            else:
                L = np.eye(4, dtype=np.float32)
                if "scale" in n:
                    L = L @ np.diag(list(n["scale"]) + [1]).astype(np.float32)
                if "translation" in n:
                    Tm = np.eye(4, dtype=np.float32)
                    Tm[:3, 3] = n["translation"]
                    L = Tm @ L
            T = (T @ L).astype(np.float32)
        for prim in doc["meshes"][mesh]["primitives"]:
            pos = accessor(prim["attributes"]["POSITION"]).astype(np.float32)
            idx = accessor(prim["indices"]).ravel().astype(np.int64)
            wp = (np.concatenate([pos, np.ones((len(pos), 1), np.float32)], 1) @ T.T)[:, :3].astype(np.float32)
            t = np.zeros(len(idx) // 3, dtype=pt.TRI_DTYPE)
            t["v0"], t["v1"], t["v2"] = wp[idx[0::3]], wp[idx[1::3]], wp[idx[2::3]]
            tris.append(t)
    tris = np.concatenate(tris)
    assert len(tris) == k["triangles"]
    for build in (lambda: pt.bvh_build(tris, 24), lambda: orc.bvh_build(pt, tris, 24)):
        nodes, idx, st = build()
        # vertex positions above come from a float32 numpy product, not GLM's operation order, so a handful of
        # centroids may fall on the other side of a split plane: allow a small slack on the counts
        assert abs(st["nodes"] - k["nodes"]) <= 8 and abs(st["leaves"] - k["leaves"]) <= 4
        assert abs(st["max_depth"] - k["max_depth"]) <= 1
        assert sorted(idx.tolist()) == list(range(len(tris)))


def _golden():
    return np.load(os.path.join(HERE, "golden", "oracle_vectors.npz"))


def test_oracle_reproduces_committed_vectors(pt, scenes, orc):
    """tests/golden/oracle_vectors.npz (written by tests/golden/make_golden.py): the oracle's deterministic mode is
    bit-stable; its libm mode depends on the C library's sinf/cosf/..., so that image is compared with a tolerance."""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden
    g = _golden()
    ro, rd = g["ray_o"], g["ray_d"]
    sc = scenes.make_scene("cornell", res=(48, 48), depth=8)
    o = orc.Oracle(pt, sc, orc.MATH_DET)
    h = o.intersect(ro, rd)
    assert h.tobytes() == g["cornell_hits"].tobytes()
    for it in (1, 2, 3):
        o.trace(it)
    assert o.image().tobytes() == g["cornell_det_image"].tobytes()
    assert int(o.stats().pathBounces) == int(g["cornell_det_path_bounces"])
    assert list(o.stats().lastLive[:9]) == list(g["cornell_det_live"][2])
    o.close()
    o = orc.Oracle(pt, sc, orc.MATH_LIBM)
    for it in (1, 2, 3):
        o.trace(it)
    ref = g["cornell_libm_image"]
    rmse = float(np.sqrt(np.mean((o.image() - ref) ** 2))) / float(ref.mean())
    assert rmse < 0.02, rmse
    o.close()
    sc = make_golden.build_mesh_scene(pt, scenes, g["blob"], g["cube"], (40, 40))
    o = orc.Oracle(pt, sc, orc.MATH_DET)
    assert o.intersect(ro, rd).tobytes() == g["mesh_hits"].tobytes()
    for it in (1, 2):
        o.trace(it)
    assert o.image().tobytes() == g["mesh_det_image"].tobytes()
    o.close()
