"""The CPU oracle and this repository's host code against the REFERENCE ITSELF: oracle/_ref/ref_tool links the
reference's own scene.cpp, bvh.cpp, utilities.cpp, stb.cpp, image.cpp and intersections.cu, compiled where they lie
(oracle/Makefile).  These are the pins of
  * the closest-hit oracle (boxes, spheres, BVH and linear mesh traversal, hit epilogue)      bit-exact
  * the BVH builder (node array, index array, statistics)                                     byte-exact
  * the scene loader and glTF import (geoms, matrices, materials, triangles, textures, camera) byte-exact
  * PNG / JPEG / Radiance decoding and the PNG / .hdr writers                                  byte-exact pixels
The shading half of the path (src/pathtrace.cu) cannot be compiled by a host compiler and stays pinned by the
recorded known answers in tests/golden/reference_kats.json only.

tests/golden/Duck.glb is the reference's own asset gltf/Duck.glb (data), so these run where /root/reference is absent
as long as the prebuilt tool travelled."""
import ctypes as C
import json
import os
import zlib

import numpy as np
import pytest

import ref_binding as ref

HERE = os.path.dirname(os.path.abspath(__file__))
DUCK = os.path.join(HERE, "golden", "Duck.glb")

if not ref.available():
    if os.path.isdir(ref.REFERENCE):
        raise RuntimeError("oracle/_ref/ref_tool is not built although %s is present: run `make -C oracle`" % ref.REFERENCE)
    pytest.skip("oracle/_ref/ref_tool is not built and the reference tree is absent", allow_module_level=True)


def _rays(seed, n, center=(0.0, 5.0, 0.0), radius=9.0, spread=4.0):
    rng = np.random.RandomState(seed)
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    o = np.asarray(center) + radius * v
    d = np.asarray(center) + rng.uniform(-spread, spread, size=(n, 3)) - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o.astype(np.float32), d.astype(np.float32)


def _same_hits(mine, theirs):
    """mine: HIT_DTYPE records of the oracle / the library; theirs: ref_tool records (no triangle index)."""
    assert np.array_equal(mine["geom"], theirs["geom"])
    assert mine["t"].tobytes() == theirs["t"].tobytes()
    hit = theirs["t"] > 0
    assert np.array_equal(mine["materialId"][hit], theirs["materialId"][hit])
    assert mine["normal"][hit].tobytes() == theirs["normal"][hit].tobytes()
    assert mine["uv"][hit].tobytes() == theirs["uv"][hit].tobytes()
    return int(hit.sum())


def _duck_json(tmp_path, scenes, name="duck_real"):
    d = scenes.scene_dict("duck")
    d["Objects"][0]["FILE"] = DUCK
    d["Objects"].append(dict(TYPE="cube", MATERIAL="light", TRANS=[0, 10, 0], ROTAT=[0, 0, 0], SCALE=[3, 0.3, 3]))
    p = tmp_path / (name + ".json")
    p.write_text(json.dumps(d))
    return str(p)


# ------------------------------------------------------------------ closest hit
@pytest.mark.parametrize("name", ["cornell", "sss", "pbr", "ssaa", "cornell_orig", "spheres"])
def test_oracle_closest_hit_equals_reference_primitives(pt, scenes, orc, name):
    sc = scenes.make_scene(name, res=(16, 16))
    ro, rd = _rays(11, 20000)
    ro[:6] = [[0, 5, 9], [0, 5, 9], [3, 5, 0], [0, 9.9, 0], [0, 5, 0], [0, 5, 0]]       # axis-parallel: x / 0 in the slab test
    rd[:6] = [[0, 0, -1], [0, 1, 0], [-1, 0, 0], [0, -1, 0], [1, 0, 0], [0, 0, 1]]
    n = _same_hits(orc.Oracle(pt, sc).intersect(ro, rd), ref.intersect(sc, ro, rd))
    assert n > 3000


def test_oracle_closest_hit_equals_reference_meshes(pt, scenes, orc):
    sc = scenes.make_scene("dragon", res=(16, 16), grid=(96, 97), env_size=(8, 4), tex_size=4)
    ro, rd = _rays(3, 40000, center=(0.0, 4.0, 0.0), radius=12.0)
    o = orc.Oracle(pt, sc)
    mine = o.intersect(ro, rd)
    n = _same_hits(mine, ref.intersect(sc, ro, rd))
    assert n > 10000 and (mine["triangle"] >= 0).sum() > 4000
    # USE_BVH off: meshIntersectionTest (linear scan) on both sides
    _same_hits(o.intersect(ro[:3000], rd[:3000], linear=True), ref.intersect(sc, ro[:3000], rd[:3000], use_bvh=False))


def test_oracle_closest_hit_equals_reference_ties_and_several_meshes(pt, scenes, orc):
    """axis-aligned 12-triangle cube (rays through shared edges and corners: exact-distance ties inside one mesh) next to
    two blobs that intersect it (ties and near-ties across geoms)."""
    sc = scenes.make_scene("cornell_orig", res=(16, 16))
    ident = ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    mid = sc.add_material(type=pt.PT_PBR, color=[0.8, 0.8, 0.8], roughness=0.5, metallic=0.0, indexOfRefraction=1.5)
    sc.add_mesh(scenes.cube_mesh(((0.0, 5.0, 0.0), (0.0, 0.0, 0.0), (2.0, 2.0, 2.0)), mid))
    sc.add_mesh(scenes.blob_mesh(40, 41, 5, ident, [mid], center=(0.8, 5.0, 0.2), radius=1.4))
    sc.add_mesh(scenes.blob_mesh(30, 31, 6, ((0.5, 0.0, 0.0), (0.0, 30.0, 0.0), (1.0, 1.2, 1.0)), [mid], center=(-0.6, 4.2, 0.0), radius=1.2),
                (0.5, 0.0, 0.0), (0.0, 30.0, 0.0), (1.0, 1.2, 1.0))
    xs = np.linspace(-1.0, 1.0, 41, dtype=np.float32)
    gx, gy = np.meshgrid(xs, xs)
    ro = np.stack([gx.ravel(), 5 + gy.ravel(), np.full(gx.size, 9.0, np.float32)], 1).astype(np.float32)
    rd = np.tile(np.array([[0, 0, -1]], np.float32), (gx.size, 1))
    r2o, r2d = _rays(8, 30000, center=(0.0, 5.0, 0.0), radius=7.0, spread=2.0)
    ro, rd = np.concatenate([ro, r2o]), np.concatenate([rd, r2d])
    mine = orc.Oracle(pt, sc).intersect(ro, rd)
    _same_hits(mine, ref.intersect(sc, ro, rd))
    assert len(set(mine["geom"].tolist())) >= 8


def test_reference_loader_and_traversal_on_its_own_duck(pt, scenes, orc, tmp_path):
    """End to end through the reference alone (its loader, its glTF import, its BVH build, its traversal) against this
    repository's loader + oracle on the same scene file."""
    path = _duck_json(tmp_path, scenes)
    sc = pt.Scene(path)
    ro, rd = _rays(5, 30000, center=(0.0, 2.0, 0.0), radius=10.0, spread=3.0)
    mine = orc.Oracle(pt, sc).intersect(ro, rd)
    n = _same_hits(mine, ref.intersect_json(path, ro, rd))
    assert (mine["triangle"] >= 0).sum() > 1500 and n > 3000


# ------------------------------------------------------------------ BVH builder
def _mesh_cases(pt, scenes):
    ident = ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    rng = np.random.RandomState(4)
    soup = np.zeros(3000, pt.TRI_DTYPE)
    c = rng.uniform(-5, 5, (3000, 3))
    for k in ("v0", "v1", "v2"):
        soup[k] = (c + rng.normal(0, 0.2, (3000, 3))).astype(np.float32)
    flat = scenes.blob_mesh(20, 21, 2, ident, [0])
    flat["v0"][:, 1] = flat["v1"][:, 1] = flat["v2"][:, 1] = 1.0              # zero extent on one axis
    dup = np.repeat(scenes.blob_mesh(8, 9, 3, ident, [0])[:40], 3)            # coincident centroids
    sc = pt.Scene(scenes.write_scene_json("cornell_orig"))
    sc.import_gltf(DUCK, (-0.5, 0.0, 0.0), (0.0, -22.5, 0.0), (2.5, 2.5, 2.5))
    return dict(blob=scenes.blob_mesh(96, 97, 1, ident, [0, 1, 2]), cube=scenes.cube_mesh(ident, 0), soup=soup, flat=flat, dup=dup,
                one=soup[:1], five=soup[:5], thirtythree=soup[:33], duck=sc.mesh_triangles(0))


def test_bvh_builder_equals_reference_builder(pt, scenes, orc):
    for name, tris in _mesh_cases(pt, scenes).items():
        for depth in (24, 6):
            rn, ri, rs = ref.bvh_build(tris, pt.NODE_DTYPE, depth)
            for who, (n, i, s) in (("library", pt.bvh_build(tris, depth)), ("oracle", orc.bvh_build(pt, tris, depth))):
                assert s == rs, (name, depth, who, s, rs)
                assert np.array_equal(i, ri), (name, depth, who)
                assert n.tobytes() == rn.tobytes(), (name, depth, who)
    rn, ri, rs = ref.bvh_build(_mesh_cases(pt, scenes)["duck"], pt.NODE_DTYPE)
    assert (len(ri), rs["nodes"], rs["leaves"], rs["max_depth"]) == (4212, 2669, 1335, 15)


# ------------------------------------------------------------------ scene loader
_GLTF_MATERIAL_FIELDS = ["type", "color", "transparency", "metallic", "roughness", "emittance", "baseColorTextureIdx",
                         "metallicRoughnessTextureIdx", "normalTextureIdx", "emissiveTextureIdx", "occlusionTextureIdx",
                         "emissiveFactor"]      # what loadGLTFMaterial assigns (src/scene.cpp:96-190); the rest is left uninitialised there


def _compare_loaded(pt, sc, r, json_material_count):
    d = sc.desc.contents
    assert d.geomCount == len(r["geoms"]) and d.materialCount == len(r["materials"])
    assert d.meshCount == len(r["meshes"]) and d.textureCount == len(r["textures"])
    for i, g in enumerate(sc.geoms()):
        q = r["geoms"][i]
        assert (g.type, g.materialid if g.type != pt.PT_GLTF_MESH else None) == (q["type"], q["materialid"] if q["type"] != pt.PT_GLTF_MESH else None)
        for f in ("translation", "rotation", "scale", "transform", "inverseTransform", "invTranspose"):
            assert bytes(getattr(g, f)) == q[f].tobytes(), (i, f)
        assert g.meshIndex == q["meshIndex"] and g.triangleCount == q["triangleCount"], i
    mats = np.frombuffer(C.string_at(d.materials, d.materialCount * 124), ref.MATERIAL_DTYPE)
    assert mats[:json_material_count].tobytes() == r["materials"][:json_material_count].tobytes()
    for f in _GLTF_MATERIAL_FIELDS:
        assert mats[f][json_material_count:].tobytes() == r["materials"][f][json_material_count:].tobytes(), f
    for m in range(d.meshCount):
        assert sc.mesh_triangles(m).tobytes() == r["meshes"][m].tobytes(), m
    for mine, theirs in zip(sc.textures(), r["textures"]):
        assert mine.shape == theirs.shape and mine.tobytes() == theirs.tobytes()
    cam = np.frombuffer(bytes(d.camera), ref.CAMERA_DTYPE)[0]
    # position / view / right / up: what the kernels see is main.cpp's recomputation before the first frame
    # (src/main.cpp:750-769, :788-807), which this repository's loader replays; the loader's own values are not kept
    for f in ("resolution", "lookAt", "fov", "pixelLength"):
        assert cam[f].tobytes() == r["camera"][f].tobytes(), f
    assert (d.traceDepth, d.iterations) == (r["traceDepth"], r["iterations"])
    assert pt.lib.pt_scene_image_name(sc._h).decode() == r["imageName"]
    if r["env"] is None:
        assert not d.env.enabled
    else:
        e = r["env"]
        assert (d.env.enabled, d.env.width, d.env.height) == (1, e["width"], e["height"])
        assert C.string_at(d.env.data, e["width"] * e["height"] * 12) == e["data"].tobytes()


@pytest.mark.parametrize("name", ["cornell", "sss", "cornell_orig", "pbr", "ssaa", "spheres", "halo"])
def test_scene_loader_equals_reference_loader(pt, scenes, tmp_path, name):
    path = scenes.write_scene_json(name, str(tmp_path))
    _compare_loaded(pt, pt.Scene(path), ref.load_scene(path, pt.TRI_DTYPE), len(scenes.scene_dict(name)["Materials"]))


def test_gltf_import_equals_reference_import_on_duck(pt, scenes, tmp_path):
    path = _duck_json(tmp_path, scenes)
    r = ref.load_scene(path, pt.TRI_DTYPE)
    sc = pt.Scene(path)
    assert len(r["meshes"][0]) == 4212 and len(r["textures"]) == 1 and len(r["materials"]) == 4
    _compare_loaded(pt, sc, r, 3)


def test_gltf_import_equals_reference_import_on_generated_files(pt, scenes, tmp_path):
    """node hierarchy with matrix and TRS nodes, indexed and non-indexed primitives, PNG and JPEG textures"""
    import test_gltf_import as T
    for k, inter in enumerate((False, True)):       # interleaved attributes: both loaders ignore byteStride the same way
        glb, _doc, _data = T._build_test_glb(tmp_path, name="g%d.glb" % k, interleaved=inter)
        d = scenes.scene_dict("cornell_orig")
        d["Objects"].append(dict(TYPE="gltf", FILE=str(glb), TRANS=[0.5, 1.0, -0.25], ROTAT=[10.0, 20.0, 30.0], SCALE=[1.5, 0.75, 1.25]))
        p = tmp_path / ("s%d.json" % k)
        p.write_text(json.dumps(d))
        r = ref.load_scene(str(p), pt.TRI_DTYPE)
        assert len(r["meshes"][0]) == 18
        _compare_loaded(pt, pt.Scene(p), r, len(d["Materials"]))


def test_gltf_import_equals_reference_import_with_a_progressive_jpeg_texture(pt, scenes, tmp_path):
    """tinygltf + stb_image read progressive JPEG textures; so does the import, to the same bytes"""
    import glb_util as G
    data = open(os.path.join(HERE, "golden", "progressive", "rgb420_130x67_q95.jpg"), "rb").read()
    b = G.GlbBuilder()
    bv = b._view(data)
    b.doc["images"].append(dict(bufferView=bv, mimeType="image/jpeg"))
    b.doc["textures"].append(dict(source=0))
    b.material(pbrMetallicRoughness=dict(baseColorTexture=dict(index=0)))
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    b.node(root=True, mesh=b.mesh([dict(attributes=dict(POSITION=b.accessor(pos, "VEC3", 5126)), material=0)]))
    b.write_glb(tmp_path / "pj.glb")
    d = scenes.scene_dict("cornell_orig")
    d["Objects"].append(dict(TYPE="gltf", FILE=str(tmp_path / "pj.glb"), TRANS=[0.0, 1.0, 0.0], ROTAT=[0.0, 0.0, 0.0], SCALE=[1.0, 1.0, 1.0]))
    p = tmp_path / "pj.json"
    p.write_text(json.dumps(d))
    r = ref.load_scene(str(p), pt.TRI_DTYPE)
    sc = pt.Scene(p)
    assert sc.log == "" and len(r["textures"]) == 1 and r["textures"][0].shape == (67, 130, 4)
    _compare_loaded(pt, sc, r, len(d["Materials"]))


@pytest.mark.skipif(not os.path.isdir(os.path.join(ref.REFERENCE, "scenes")), reason="reference scene files are not present on this machine")
def test_every_shipped_scene_file_loads_like_the_reference(pt):
    import glob
    for f in sorted(glob.glob(os.path.join(ref.REFERENCE, "scenes", "*.json"))):
        _compare_loaded(pt, pt.Scene(f), ref.load_scene(f, pt.TRI_DTYPE), len(json.load(open(f))["Materials"]))


# ------------------------------------------------------------------ image codecs
def _png_bytes(img):
    import glb_util as G
    ctype = {1: 0, 3: 2, 4: 6}[1 if img.ndim == 2 else img.shape[2]]
    return G.write_png(img, ctype)


def test_png_jpeg_hdr_decoders_equal_stb_image(pt, scenes, tmp_path):
    import jpeg_util as J
    import test_gltf_import as T
    rng = np.random.RandomState(9)
    # PNG: grey, RGB, RGBA
    for k, shape in enumerate([(13, 17), (9, 11, 3), (16, 5, 4)]):
        img = rng.randint(0, 256, shape).astype(np.uint8)
        p = tmp_path / ("p%d.png" % k)
        p.write_bytes(_png_bytes(img))
        kind, comps, rgba, _f = ref.stbi(str(p))
        assert kind == "ldr" and pt.decode_png(p.read_bytes()).tobytes() == rgba.tobytes()
    # JPEG: every chroma sampling, sizes off the MCU grid, restart intervals, grey
    for k, (sampling, size, restart) in enumerate([((1, 1), (16, 16), 0), ((2, 1), (37, 53), 3), ((1, 2), (8, 9), 0), ((2, 2), (64, 33), 3),
                                                   ((2, 2), (1, 1), 0), ((2, 2), (130, 70), 0)]):
        data, _info = J.encode(T._smooth_image(size[0], size[1], k), sampling=sampling, restart=restart)
        p = tmp_path / ("j%d.jpg" % k)
        p.write_bytes(data)
        kind, comps, rgba, _f = ref.stbi(str(p))
        assert kind == "ldr" and comps == 3
        assert pt.decode_jpeg(data).tobytes() == rgba.tobytes(), (sampling, size, restart)
    data, _info = J.encode(T._smooth_image(21, 30, 5)[..., 0])
    (tmp_path / "grey.jpg").write_bytes(data)
    assert pt.decode_jpeg(data).tobytes() == ref.stbi(str(tmp_path / "grey.jpg"))[2].tobytes()
    # progressive files (libjpeg-made fixtures): DC / AC first and refinement scans, end-of-band runs, restart markers
    gold = os.path.join(HERE, "golden", "progressive")
    names = sorted(f for f in os.listdir(gold) if f.endswith(".jpg"))
    assert len(names) >= 7
    for f in names:
        data = open(os.path.join(gold, f), "rb").read()
        assert pt.decode_jpeg(data).tobytes() == ref.stbi(os.path.join(gold, f))[2].tobytes(), f
    # Radiance .hdr (run-length and flat scanlines) and 8-bit files as environment maps: stbi_loadf on both sides
    img = (rng.rand(6, 40, 3) ** 4 * 50).astype(np.float32)
    img[0, :20] = (0.25, 0.5, 1.0)
    pt.write_hdr(tmp_path / "e.hdr", img)
    pt.write_hdr(tmp_path / "f.hdr", img[:, :5])
    for name in ("e.hdr", "f.hdr", "p1.png", "j1.jpg"):
        d = dict(Materials={"a": dict(TYPE="Diffuse", RGB=[1, 1, 1])}, Camera=scenes._camera("x", 2, 1, res=(4, 4)), Objects=[],
                 EnvironmentMap=dict(FILE=name, INTENSITY=1.0))
        (tmp_path / "s.json").write_text(json.dumps(d))
        sc = pt.Scene(tmp_path / "s.json")
        env = sc.desc.contents.env
        got = ref.stbi(str(tmp_path / name))
        theirs = got[2] if got[0] == "hdr" else got[3]
        assert (env.height, env.width) == theirs.shape[:2]
        assert C.string_at(env.data, theirs.size * 4) == theirs.tobytes(), name


def _png_pixels(data):
    """independent PNG read (zlib + filters) of an 8-bit RGB file"""
    import test_image_io as TI
    return TI._read_png_independent(data)


def test_image_writers_equal_the_reference_writers(pt, tmp_path):
    rng = np.random.RandomState(12)
    h, w, samples = 7, 19, 5
    s = (rng.rand(h, w, 3) ** 3 * 9).astype(np.float32)
    s[0, 0] = (-1.0, 0.0, 1e9)
    avg = pt.image_average(s, samples)                                  # sum / samples, x mirrored (saveImage, src/main.cpp:351-375)
    # PNG: the reference clamps and truncates inside savePNG; same bytes as pt_image_to_rgb8 + pt_write_png
    theirs = ref.save_image("png", avg, str(tmp_path / "ref"))
    pt.write_png(tmp_path / "mine.png", pt.image_to_rgb8(s, samples))
    a, b = _png_pixels(open(theirs, "rb").read()), _png_pixels((tmp_path / "mine.png").read_bytes())
    assert a.shape == (h, w, 3) and a.tobytes() == b.tobytes()
    # .hdr: identical files (header fields, run-length coding) but for the writer's name, for narrow and wide images
    for k, img in enumerate((avg, np.tile(avg, (1, 3, 1))[:, :40], avg[:, :5])):
        theirs = ref.save_image("hdr", img, str(tmp_path / ("ref%d" % k)))
        pt.write_hdr(tmp_path / ("mine%d.hdr" % k), img)
        a, b = open(theirs, "rb").read(), (tmp_path / ("mine%d.hdr" % k)).read_bytes()
        assert b"# Written by stb_image_write.h\n" in a and b"# Written by libmi_pathtrace\n" in b      # the one line that differs
        assert a.replace(b"stb_image_write.h", b"X") == b.replace(b"libmi_pathtrace", b"X"), k


def test_maintainer_shim_compiles_against_the_reference_headers():
    """integration/pathtrace_mi355x.cpp (the file a maintainer of the reference adds to its build) is compiled by oracle/Makefile
    against the reference's own pathtrace.h / scene.h / sceneStructs.h and linked with the reference's loader and
    libmi_pathtrace.so; what it renders is checked on the GPU (tests/test_gpu_full_size.py)."""
    render = os.path.join(os.path.dirname(ref.TOOL), "ref_render")
    if os.path.isdir(ref.REFERENCE):
        assert os.path.exists(render), "run `make -C oracle`"
    text = open(os.path.join(os.path.dirname(HERE), "integration", "pathtrace_mi355x.cpp")).read()
    for name in ("void InitDataContainer(GuiDataContainer*", "void pathtraceInit(Scene*", "void pathtraceFree()", "void pathtrace(uchar4*"):
        assert name in text
