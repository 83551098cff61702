"""glTF import and PNG decoding of the host-side scene loader (role of Scene::loadGLTFModel, src/scene.cpp:44-568).

The importer is checked against (i) a float32 restatement of the reference's transform chain written here with
numpy, in GLM's order of operations, on generated GLB files; (ii) the reference's own asset gltf/Duck.glb loaded
through its scenes/duck.json, whose triangle count and BVH statistics were recorded from a host compilation of
the reference (tests/golden/reference_kats.json) — only where the reference tree is present.
PNG decoding is lossless, so expected pixels are the written samples expanded the way stb_image expands them.
"""
import json
import os
import zlib

import numpy as np
import pytest

import glb_util as G

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))
REFERENCE_DUCK_SCENE = "/root/reference/scenes/duck.json"
f32 = np.float32


# ------------------------------------------------------------------ PNG

def _rand(shape, hi, seed):
    return np.random.default_rng(seed).integers(0, hi, size=shape, dtype=np.uint8)


PNG_CASES = [
    dict(ctype=6, shape=(9, 13, 4)), dict(ctype=2, shape=(7, 5, 3)), dict(ctype=0, shape=(6, 11, 1)), dict(ctype=4, shape=(5, 9, 2)),
    dict(ctype=0, depth=1, shape=(9, 19, 1)), dict(ctype=0, depth=2, shape=(4, 7, 1)), dict(ctype=0, depth=4, shape=(5, 5, 1)),
    dict(ctype=3, depth=8, shape=(8, 8, 1), pal=200), dict(ctype=3, depth=4, shape=(6, 9, 1), pal=16), dict(ctype=3, depth=2, shape=(3, 6, 1), pal=4),
    dict(ctype=3, depth=1, shape=(5, 17, 1), pal=2),
]


@pytest.mark.parametrize("case", PNG_CASES, ids=lambda c: "ct%d_d%d" % (c["ctype"], c.get("depth", 8)))
@pytest.mark.parametrize("interlace", [False, True])
def test_png_colour_types_filters_interlace(pt, case, interlace):
    depth = case.get("depth", 8)
    hi = case.get("pal", 1 << depth) if case["ctype"] == 3 else (1 << depth)
    s = _rand(case["shape"], hi, seed=case["ctype"] * 10 + depth)
    pal = _rand((case["pal"], 3), 256, 5) if "pal" in case else None
    for filt in ("cycle", 0, 1, 2, 3, 4):
        png = G.write_png(s, case["ctype"], depth, palette=pal, filt=filt, interlace=interlace)
        got = pt.decode_png(png)
        assert np.array_equal(got, G.expected_rgba(s, case["ctype"], depth, palette=pal)), (case, filt)


def test_png_transparency_chunks(pt):
    s = _rand((6, 6, 1), 4, 1)
    trns = bytes([0, 2])
    assert np.array_equal(pt.decode_png(G.write_png(s, 0, 2, trns=trns)), G.expected_rgba(s, 0, 2, trns=trns))
    s = _rand((6, 6, 3), 3, 2)
    trns = bytes([0, 1, 0, 2, 0, 0])
    s[2, 3] = (1, 2, 0)
    exp = G.expected_rgba(s, 2, trns=trns)
    assert exp[2, 3, 3] == 0
    assert np.array_equal(pt.decode_png(G.write_png(s, 2, trns=trns)), exp)
    s = _rand((5, 5, 1), 6, 3)
    pal = _rand((6, 3), 256, 4)
    trns = bytes([0, 128, 255, 7])
    assert np.array_equal(pt.decode_png(G.write_png(s, 3, 8, palette=pal, trns=trns)), G.expected_rgba(s, 3, 8, palette=pal, trns=trns))


def test_png_deflate_block_kinds_and_split_idat(pt):
    """Stored, fixed-Huffman and dynamic-Huffman blocks; long matches; IDAT cut into small chunks."""
    s = np.zeros((64, 64, 4), np.uint8)
    s[:, :, 0] = np.arange(64)[None, :] * 3
    s[16:48, 16:48, 1] = 200
    s[:, :, 3] = _rand((64, 64), 256, 9)
    exp = G.expected_rgba(s, 6)
    for level, strategy in ((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED),
                            (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)):
        assert np.array_equal(pt.decode_png(G.write_png(s, 6, level=level, strategy=strategy, idat_split=97)), exp), (level, strategy)


def test_png_rejects_damage(pt):
    png = bytearray(G.write_png(_rand((8, 8, 3), 256, 1), 2))
    for bad in (b"", bytes(png[:20]), b"\x89PNG\r\n\x1a\n" + b"\0" * 32):
        with pytest.raises(pt.PtError):
            pt.decode_png(bad)
    broken = bytearray(png)
    broken[45] ^= 0x5A                      # inside the deflate stream: bad code or Adler-32 mismatch
    with pytest.raises(pt.PtError):
        pt.decode_png(bytes(broken))
    s16 = bytes(png[:24]) + bytes([16]) + bytes(png[25:])
    with pytest.raises(pt.PtError, match="16-bit"):
        pt.decode_png(s16)


# ------------------------------------------------------------------ float32 restatement of the transform chain

def _matmul(A, B):
    R = np.zeros((4, 4), f32)
    for j in range(4):
        R[j] = ((A[0] * B[j][0] + A[1] * B[j][1]) + A[2] * B[j][2]) + A[3] * B[j][3]
    return R


def _translate(v):
    I = np.eye(4, dtype=f32)
    R = I.copy()
    R[3] = ((I[0] * f32(v[0]) + I[1] * f32(v[1])) + I[2] * f32(v[2])) + I[3]
    return R


def _scale(v):
    I = np.eye(4, dtype=f32)
    R = I.copy()
    for k in range(3):
        R[k] = I[k] * f32(v[k])
    return R


def _quat(x, y, z, w):
    x, y, z, w = f32(x), f32(y), f32(z), f32(w)
    R = np.eye(4, dtype=f32)
    one, two = f32(1), f32(2)
    qxx, qyy, qzz, qxz, qxy, qyz, qwx, qwy, qwz = x * x, y * y, z * z, x * z, x * y, y * z, w * x, w * y, w * z
    R[0][0] = one - two * (qyy + qzz); R[0][1] = two * (qxy + qwz); R[0][2] = two * (qxz - qwy)
    R[1][0] = two * (qxy - qwz); R[1][1] = one - two * (qxx + qzz); R[1][2] = two * (qyz + qwx)
    R[2][0] = two * (qxz + qwy); R[2][1] = two * (qyz - qwx); R[2][2] = one - two * (qxx + qyy)
    return R


def _local(node):
    if "matrix" in node:
        return np.array(node["matrix"], np.float64).astype(f32).reshape(4, 4).T.copy()     # read transposed, as the reference does
    T = _translate(node["translation"]) if "translation" in node else np.eye(4, dtype=f32)
    R = _quat(*node["rotation"]) if "rotation" in node else np.eye(4, dtype=f32)
    S = _scale(node["scale"]) if "scale" in node else np.eye(4, dtype=f32)
    return _matmul(_matmul(T, R), S)


def _point(M, p):
    p = np.asarray(p, f32)
    return ((M[0] * p[0] + M[1] * p[1]) + (M[2] * p[2] + M[3] * f32(1)))[:3]


def _build_test_glb(tmp_path, name="t.glb", text=None, interleaved=False):
    rng = np.random.default_rng(7)
    b = G.GlbBuilder()
    tex = b.image_png(G.write_png(_rand((8, 8, 3), 256, 3), 2))
    b.material(pbrMetallicRoughness=dict(baseColorFactor=[0.8, 0.2, 0.1, 0.75], metallicFactor=0.3, roughnessFactor=0.6,
                                         baseColorTexture=dict(index=tex), metallicRoughnessTexture=dict(index=5)),
               normalTexture=dict(index=tex))
    b.material(emissiveFactor=[1.0, 2.0, 2.0])
    b.material(alphaMode="MASK", alphaCutoff=0.7, pbrMetallicRoughness=dict(baseColorFactor=[0.5, 0.5, 0.5, 1.0]))
    b.material()
    # mesh 0: an indexed u16 primitive with three attributes, an indexed u32 primitive with positions only.  The reference
    # reads attributes tightly packed whatever the buffer view's byteStride says (src/scene.cpp:363-420), so the
    # interleaved form of the file only makes sense next to the reference's own loader (tests/test_reference_build.py)
    pos = rng.normal(size=(12, 3)).astype(f32)
    nrm = rng.normal(size=(12, 3)).astype(f32)
    uv = rng.random(size=(12, 2)).astype(f32)
    if interleaved:
        a_pos, a_nrm, a_uv = b.interleaved(pos, nrm, uv)
    else:
        a_pos, a_nrm, a_uv = b.accessor(pos, "VEC3", 5126), b.accessor(nrm, "VEC3", 5126), b.accessor(uv, "VEC2", 5126)
    idx16 = rng.integers(0, 12, size=18).astype(np.uint16)
    pos2 = rng.normal(size=(9, 3)).astype(f32)
    idx32 = np.concatenate([rng.permutation(9), rng.permutation(9)[:3]]).astype(np.uint32)
    idx8 = np.arange(6, dtype=np.uint8)
    m0 = b.mesh([
        dict(attributes=dict(POSITION=a_pos, NORMAL=a_nrm, TEXCOORD_0=a_uv), indices=b.accessor(idx16, "SCALAR", 5123), material=0),
        dict(attributes=dict(POSITION=b.accessor(pos2, "VEC3", 5126)), indices=b.accessor(idx32, "SCALAR", 5125), material=2),
        dict(attributes=dict(POSITION=a_pos), indices=b.accessor(idx16, "SCALAR", 5123), mode=1),                  # LINES: skipped
        dict(attributes=dict(POSITION=a_pos), indices=b.accessor(idx8, "SCALAR", 5121), material=1),               # u8 indices: skipped
    ])
    # mesh 1: no indices, normals only, material index out of range -> first imported material
    pos3 = rng.normal(size=(6, 3)).astype(f32)
    nrm3 = rng.normal(size=(6, 3)).astype(f32)
    m1 = b.mesh([dict(attributes=dict(POSITION=b.accessor(pos3, "VEC3", 5126), NORMAL=b.accessor(nrm3, "VEC3", 5126)), material=9),
                 dict(attributes=dict(POSITION=b.accessor(pos3, "VEC3", 5126)), material=1)])
    q = np.array([0.2, -0.4, 0.1, 0.8])
    q = (q / np.linalg.norm(q)).tolist()
    leaf = b.node(mesh=m1, translation=[0.5, -0.25, 2.0], scale=[1.5, 0.5, 2.0])
    mid = b.node(mesh=m0, translation=[1.0, 2.0, -3.0], rotation=q, scale=[0.5, 2.0, 1.25], children=[leaf])
    b.node(root=True, children=[mid], matrix=[0.0, 2.0, 0.0, 0.0, -1.5, 0.0, 0.0, 0.0, 0.0, 0.0, 1.25, 0.0, 0.25, -0.5, 0.75, 1.0])
    b.node(root=True, mesh=m1)                                                                                     # a second root, no transform
    path = tmp_path / name
    if text is None:
        b.write_glb(path)
    else:
        b.write_gltf(path, embed=(text == "embed"))
    data = dict(pos=pos, nrm=nrm, uv=uv, idx16=idx16, pos2=pos2, idx32=idx32, pos3=pos3, nrm3=nrm3)
    return path, b.doc, data


def _expected_triangles(pt, doc, data, root, mat_base):
    tris = []

    def emit(M, P, N, UV, order, mat):
        Nm = np.linalg.inv(M.astype(np.float64).T)[:3, :3] if N is not None else None     # [col][row] of transpose(inverse(M))
        for i in range(0, len(order) - 2, 3):
            v = [int(order[i + k]) for k in range(3)]
            p = [_point(M, P[j]) for j in v]
            if N is not None:
                n = []
                for j in v:
                    x = N[j].astype(np.float64)
                    w = Nm[0] * x[0] + Nm[1] * x[1] + Nm[2] * x[2]
                    n.append(w / np.linalg.norm(w))
            else:
                e1, e2 = p[1] - p[0], p[2] - p[0]
                c = np.array([e1[1] * e2[2] - e2[1] * e1[2], e1[2] * e2[0] - e2[2] * e1[0], e1[0] * e2[1] - e2[0] * e1[1]], f32)
                d = c * c
                fn = c * (f32(1) / np.sqrt(f32(d[0] + d[1]) + d[2], dtype=f32))
                n = [fn, fn, fn]
            uvs = [UV[j] for j in v] if UV is not None else [np.zeros(2, f32)] * 3
            tris.append((p, n, uvs, mat, N is not None))

    def walk(ni, parent):
        node = doc["nodes"][ni]
        M = _matmul(parent, _local(node))
        if "mesh" in node:
            if node["mesh"] == 0:
                emit(M, data["pos"], data["nrm"], data["uv"], data["idx16"], mat_base + 0)
                emit(M, data["pos2"], None, None, data["idx32"], mat_base + 2)
            else:
                emit(M, data["pos3"], data["nrm3"], None, np.arange(6), mat_base + 0)
                emit(M, data["pos3"], None, None, np.arange(6), mat_base + 1)
        for c in node.get("children", []):
            walk(c, M)

    for r in doc["scenes"][0]["nodes"]:
        walk(r, root)
    return tris


def _base_scene(pt, scenes, tmp_path):
    d = dict(Materials={"a": dict(TYPE="Diffuse", RGB=[1, 1, 1]), "l": dict(TYPE="Emitting", RGB=[1, 1, 1], EMITTANCE=4.0)},
             Camera=scenes._camera("g", 4, 1, res=(8, 8)), Objects=[scenes._obj("cube", "l", (0, 9, 0), (0, 0, 0), (3, 0.3, 3))])
    p = tmp_path / "base.json"
    p.write_text(json.dumps(d))
    return pt.Scene(p)


@pytest.mark.parametrize("form", [None, "embed", "external"])
def test_import_matches_float32_restatement(pt, scenes, tmp_path, form):
    path, doc, data = _build_test_glb(tmp_path, "t.glb" if form is None else "t.gltf", form)
    sc = _base_scene(pt, scenes, tmp_path)
    trs = ((0.5, -1.0, 2.0), (10.0, 20.0, 30.0), (1.5, 0.75, 2.0))
    slot = sc.import_gltf(path, *trs)
    g = sc.geoms()[slot]
    assert g.type == 2 and g.meshIndex == 0 and g.materialid == -1
    got = sc.mesh_triangles(0)
    root = pt.build_transforms(*trs)[0]
    exp = _expected_triangles(pt, doc, data, root, mat_base=2)
    assert len(got) == len(exp) == g.triangleCount == 6 + 4 + 2 + 2 + 2 + 2
    for t, (p, n, uvs, mat, smooth) in zip(got, exp):
        for k, name in enumerate(("v0", "v1", "v2")):
            assert np.array_equal(t[name], p[k]), name
        for k, name in enumerate(("n0", "n1", "n2")):
            if smooth:
                assert np.allclose(t[name], n[k], atol=2e-6)
                assert abs(np.linalg.norm(t[name].astype(np.float64)) - 1) < 1e-6
            else:
                assert np.array_equal(t[name], n[k])
        for k, name in enumerate(("uv0", "uv1", "uv2")):
            assert np.array_equal(t[name], uvs[k])
        assert t["materialId"] == mat
    assert "non-triangle primitive skipped" in sc.log and "unsupported index type" in sc.log and "texture 5" not in sc.log

    mats = sc.materials()
    assert len(mats) == 2 + 4
    m = mats[2]
    assert m.type == pt.PT_PBR and list(m.color) == [f32(0.8), f32(0.2), f32(0.1)]
    assert m.transparency == f32(1.0 - 0.75) and m.metallic == f32(0.3) and m.roughness == f32(0.6)
    assert (m.baseColorTextureIdx, m.metallicRoughnessTextureIdx, m.normalTextureIdx, m.emissiveTextureIdx, m.occlusionTextureIdx) == (0, -1, 0, -1, -1)
    m = mats[3]
    assert m.type == pt.PT_EMITTING and m.emittance == f32(3.0) and list(m.emissiveFactor) == [1.0, 2.0, 2.0]
    assert list(m.color) == [1.0, 1.0, 1.0] and m.metallic == 1.0 and m.roughness == 1.0 and m.transparency == 0.0
    m = mats[4]
    assert m.type == pt.PT_PBR and m.transparency == 1.0 and list(m.color) == [0.5, 0.5, 0.5]
    m = mats[5]
    assert m.type == pt.PT_PBR and m.transparency == 0.0 and m.emittance == 0.0 and m.subsurfaceEnabled == 0
    tex = sc.textures()
    assert len(tex) == 1 and tex[0].shape == (8, 8, 4)
    assert np.array_equal(tex[0], G.expected_rgba(_rand((8, 8, 3), 256, 3), 2))


def test_interleaved_file_is_imported_the_reference_way_and_says_so(pt, scenes, tmp_path):
    """A buffer view's byteStride is not followed (the reference reads attributes tightly packed, src/scene.cpp:363-420):
    the import still succeeds, and pt_scene_log names every accessor whose stride was ignored; a packed file leaves no
    such line."""
    for inter in (False, True):
        glb, _doc, _data = _build_test_glb(tmp_path, name="s%d.glb" % inter, interleaved=inter)
        sc = pt.Scene(scenes.write_scene_json("cornell_orig", str(tmp_path)))
        sc.import_gltf(glb)
        notes = [l for l in sc.log.splitlines() if "byteStride" in l]
        if inter:
            assert len(notes) >= 3 and any("POSITION" in l for l in notes) and any("TEXCOORD_0" in l for l in notes), sc.log
        else:
            assert not notes, notes


def test_import_errors(pt, scenes, tmp_path):
    sc = _base_scene(pt, scenes, tmp_path)
    with pytest.raises(pt.PtError, match="cannot open"):
        sc.import_gltf(tmp_path / "missing.glb")
    bad = tmp_path / "bad.glb"
    bad.write_bytes(b"glTF" + b"\x02\0\0\0" + b"\x10\0\0\0" + b"\0\0\0\0")
    with pytest.raises(pt.PtError):
        sc.import_gltf(bad)
    # an accessor that reaches past its buffer
    b = G.GlbBuilder()
    a = b.accessor(np.zeros((3, 3), f32), "VEC3", 5126)
    b.doc["accessors"][a]["count"] = 300
    b.node(root=True, mesh=b.mesh([dict(attributes=dict(POSITION=a))]))
    b.write_glb(tmp_path / "oob.glb")
    with pytest.raises(pt.PtError, match="past its buffer"):
        sc.import_gltf(tmp_path / "oob.glb")
    # an index beyond the vertex count
    b = G.GlbBuilder()
    a = b.accessor(np.zeros((3, 3), f32), "VEC3", 5126)
    i = b.accessor(np.array([0, 1, 7], np.uint16), "SCALAR", 5123)
    b.node(root=True, mesh=b.mesh([dict(attributes=dict(POSITION=a), indices=i)]))
    b.write_glb(tmp_path / "idx.glb")
    with pytest.raises(pt.PtError, match="index out of range"):
        sc.import_gltf(tmp_path / "idx.glb")
    # a node cycle ends with an error instead of recursing for ever
    b = G.GlbBuilder()
    b.node(root=True, children=[0])
    b.write_glb(tmp_path / "cycle.glb")
    with pytest.raises(pt.PtError, match="too deep"):
        sc.import_gltf(tmp_path / "cycle.glb")
    assert len(sc.geoms()) == 1 and len(sc.materials()) == 2          # failed imports add nothing


def test_scene_json_gltf_object(pt, scenes, tmp_path):
    """TYPE "gltf" objects of a scene file: FILE is relative to the scene file; a missing file leaves an empty mesh object."""
    sub = tmp_path / "models"
    sub.mkdir()
    path, doc, data = _build_test_glb(sub)
    d = dict(Materials={"a": dict(TYPE="Diffuse", RGB=[1, 1, 1])}, Camera=scenes._camera("g", 4, 1, res=(8, 8)),
             Objects=[dict(TYPE="gltf", FILE="models/t.glb", TRANS=[0, 1, 0], ROTAT=[0, 45, 0], SCALE=[2, 2, 2]),
                      dict(TYPE="gltf", FILE="models/none.glb", TRANS=[0, 0, 0], ROTAT=[0, 0, 0], SCALE=[1, 1, 1])])
    p = tmp_path / "s.json"
    p.write_text(json.dumps(d))
    sc = pt.Scene(p)
    g = sc.geoms()
    assert len(g) == 2 and g[0].type == 2 and g[0].meshIndex == 0 and g[0].triangleCount == 18
    assert g[1].type == 2 and g[1].meshIndex == -1 and g[1].triangleCount == 0
    assert "glTF object left empty" in sc.log and "none.glb" in sc.log
    assert len(sc.materials()) == 1 + 4
    exp = _expected_triangles(pt, doc, data, pt.build_transforms((0, 1, 0), (0, 45, 0), (2, 2, 2))[0], mat_base=1)
    got = sc.mesh_triangles(0)
    assert all(np.array_equal(t["v1"], e[0][1]) for t, e in zip(got, exp))


@pytest.mark.skipif(not os.path.exists(REFERENCE_DUCK_SCENE), reason="reference assets are not present on this machine")
def test_reference_duck_scene_known_answers(pt, orc):
    """scenes/duck.json of the reference, loaded as is: Duck.glb -> 4 212 triangles, one textured PBR material after the
    three of the scene file, and BVH::build statistics equal to the values recorded from the reference."""
    k = KAT["duck_bvh"]
    sc = pt.Scene(REFERENCE_DUCK_SCENE)
    g = sc.geoms()
    assert len(g) == 1 and g[0].type == 2 and g[0].triangleCount == k["triangles"]
    tris = sc.mesh_triangles(0)
    mats = sc.materials()
    assert len(mats) == 4 and mats[3].type == pt.PT_PBR and mats[3].baseColorTextureIdx == 0 and mats[3].metallic == 0.0 and mats[3].roughness == 1.0
    assert set(tris["materialId"].tolist()) == {3}
    tex = sc.textures()
    assert len(tex) == 1 and tex[0].shape[2] == 4 and tex[0].shape[0] == tex[0].shape[1] == 512
    assert tex[0][:, :, 3].min() == 255 and tex[0][:, :, :3].std() > 10
    for n in ("n0", "n1", "n2"):
        assert np.allclose(np.linalg.norm(tris[n].astype(np.float64), axis=1), 1, atol=1e-5)
    assert 0 <= tris["uv0"].min() and tris["uv0"].max() <= 1
    for build in (lambda: pt.bvh_build(tris, 24), lambda: orc.bvh_build(pt, tris, 24)):
        nodes, idx, st = build()
        assert (st["nodes"], st["leaves"], st["max_depth"]) == (k["nodes"], k["leaves"], k["max_depth"])
        assert sorted(idx.tolist()) == list(range(len(tris)))


# ------------------------------------------------------------------ JPEG

def _smooth_image(h, w, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(x / 7.0 + seed) * np.cos(y / 9.0), 128 + 90 * np.cos(x / 5.0) * np.sin(y / 11.0 + 1), 128 + 110 * np.sin((x + y) / 13.0)], -1)
    img += rng.normal(0, 6, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("sampling", [(1, 1), (2, 1), (1, 2), (2, 2)])
@pytest.mark.parametrize("size", [(16, 16), (37, 53), (8, 9), (1, 1), (64, 33)])
def test_jpeg_decoder_against_independent_float_decode(pt, sampling, size):
    """Baseline JPEG written by tests/jpeg_util.py (every chroma sampling the upsampler treats specially, sizes
    off the MCU grid, with and without restart intervals): the library's integer decode stays within 4 levels of
    the float decode of the very coefficients the file stores."""
    import jpeg_util as J
    img = _smooth_image(size[0], size[1], sampling[0] * 3 + sampling[1])
    for restart in (0, 3):
        data, info = J.encode(img, sampling=sampling, restart=restart)
        got = pt.decode_jpeg(data).astype(np.float64)
        exp = J.expected(info)
        assert got.shape == exp.shape
        assert np.abs(got - exp).max() <= 4.0, (sampling, size, restart, float(np.abs(got - exp).max()))
        assert (got[..., 3] == 255).all()
        # and the file really is a lossy version of the picture, not something else
        assert np.abs(got[..., :3] - img).mean() < 14


def test_jpeg_grey_rgb_and_marker_rules(pt):
    import jpeg_util as J
    grey = _smooth_image(21, 30, 5)[..., 0]
    data, info = J.encode(grey)
    got = pt.decode_jpeg(data)
    assert np.abs(got.astype(np.float64) - J.expected(info)).max() <= 2 and (got[..., 0] == got[..., 1]).all() and (got[..., 1] == got[..., 2]).all()
    img = _smooth_image(24, 24, 6)
    # component ids 'R','G','B': planes are taken as they are
    data, info = J.encode(img, rgb_ids=True)
    assert np.abs(pt.decode_jpeg(data).astype(np.float64) - J.expected(info)).max() <= 2
    # Adobe marker with transform 0 and no JFIF marker: RGB as well; with a JFIF marker the file counts as YCbCr
    data, info = J.encode(img, jfif=False, adobe=0)
    raw = dict(info, rgb=True)
    assert np.abs(pt.decode_jpeg(data).astype(np.float64) - J.expected(raw)).max() <= 2
    data, info = J.encode(img, jfif=True, adobe=0)
    assert np.abs(pt.decode_jpeg(data).astype(np.float64) - J.expected(info)).max() <= 3


def test_jpeg_rejects_what_it_cannot_read(pt):
    import jpeg_util as J
    data, _ = J.encode(_smooth_image(16, 16, 1))
    with pytest.raises(pt.PtError, match="arithmetic"):
        pt.decode_jpeg(data.replace(b"\xff\xc0", b"\xff\xc9", 1))
    with pytest.raises(pt.PtError):      # a sequential scan (full band) under a progressive frame header
        pt.decode_jpeg(data.replace(b"\xff\xc0", b"\xff\xc2", 1))
    for bad in (b"", b"\xff\xd8", data[:40], data[:len(data) // 2].replace(b"\xff\xda", b"\xff\xdb", 1)):
        with pytest.raises(pt.PtError):
            pt.decode_jpeg(bad)
    # a truncated scan still gives an image of the right size (missing blocks decode from zero bits)
    cut = pt.decode_jpeg(data[:len(data) - 40] + b"\xff\xd9")
    assert cut.shape == (16, 16, 4)


def test_progressive_jpeg_fixtures(pt):
    """Progressive files (tests/golden/progressive, made by libjpeg through Pillow): the pixels are the ones the
    reference's vendored stb_image makes of them (hashes stored by tests/golden/make_progressive_jpegs.py)."""
    import hashlib
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "progressive")
    expected = json.load(open(os.path.join(gold, "expected.json")))
    assert len(expected) >= 7
    for name, e in expected.items():
        data = open(os.path.join(gold, name + ".jpg"), "rb").read()
        assert len(data) == e["bytes"] and b"\xff\xc2" in data
        got = pt.decode_jpeg(data)
        assert got.shape == (e["height"], e["width"], 4)
        assert hashlib.sha256(got.tobytes()).hexdigest() == e["rgba_sha256"], name


def test_glb_with_jpeg_texture(pt, scenes, tmp_path):
    import jpeg_util as J
    img = _smooth_image(32, 32, 9)
    data, info = J.encode(img, sampling=(2, 2))
    b = G.GlbBuilder()
    bv = b._view(data)
    b.doc["images"].append(dict(bufferView=bv, mimeType="image/jpeg"))
    b.doc["textures"].append(dict(source=0))
    b.material(pbrMetallicRoughness=dict(baseColorTexture=dict(index=0)))
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], f32)
    b.node(root=True, mesh=b.mesh([dict(attributes=dict(POSITION=b.accessor(pos, "VEC3", 5126)), material=0)]))
    b.write_glb(tmp_path / "j.glb")
    sc = _base_scene(pt, scenes, tmp_path)
    sc.import_gltf(tmp_path / "j.glb")
    tex = sc.textures()
    assert sc.log == "" and len(tex) == 1 and tex[0].shape == (32, 32, 4)
    assert np.array_equal(tex[0], pt.decode_jpeg(data))
    assert sc.materials()[-1].baseColorTextureIdx == 0
