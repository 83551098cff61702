"""CPU-only checks of the host side: the C ABI library loads and exports every declared symbol, the scene
loader reproduces the reference's numbering and camera set-up, the BVH builder matches the oracle's
restatement of BVH::build, and the two arithmetic modes of the oracle stay close."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
KAT = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))


def test_library_exports_every_declared_symbol(pt):
    header = open(os.path.join(ROOT, "include", "mi_pathtrace.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 30
    lib = C.CDLL(pt.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libmi_pathtrace.so does not export " + name
    assert declared == set(pt.EXPORTED)
    assert b"gfx950" in pt.lib.pt_version()


def test_library_exports_the_pathtrace_surface(pt):
    """pathtraceInit / pathtrace / pathtraceFree / InitDataContainer (src/pathtrace.h:8-11) as C++ symbols."""
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", "-C", pt.LIB_PATH], text=True)
    for sig in ("pathtraceInit(PtScene*)", "pathtraceFree()", "pathtrace(uchar4*, int, int)", "InitDataContainer(GuiDataContainer*)"):
        assert sig in out, sig
    headless = os.path.join(os.path.dirname(pt.LIB_PATH), "pt_headless")
    assert os.access(headless, os.X_OK)


def test_pod_sizes_match_the_reference_structs(pt):
    """Material 124 B, Triangle 100 B, BVHNode 40 B, Camera 84 B (src/sceneStructs.h:40-194, src/bvh.h:16-65,
    sizes measured by compiling the reference headers): a binding can hand the reference's arrays over as they are."""
    assert C.sizeof(pt.PtMaterial) == 124
    assert pt.TRI_DTYPE.itemsize == 100
    assert pt.NODE_DTYPE.itemsize == 40
    assert C.sizeof(pt.PtCamera) == 84
    assert pt.HIT_DTYPE.itemsize == 36


def test_library_holds_gfx950_code_object(pt):
    blob = open(pt.LIB_PATH, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob
    for kernel in (b"k_trace", b"k_prims", b"k_cull", b"k_shade", b"k_sort_scatter", b"k_compact_scatter", b"k_camera"):
        assert kernel in blob


def test_material_numbering_is_alphabetical(pt, scenes):
    """nlohmann::json objects iterate in key order, so ids are positions in the sorted key list."""
    sc = scenes.make_scene("duck", res=(8, 8), grid=(8, 9), env_size=(8, 4), tex_size=4)
    k = KAT["material_order_duck_json"]
    names = sorted(scenes.scene_dict("duck")["Materials"])
    assert {n: i for i, n in enumerate(names)} == {n: k[n] for n in names}
    mats = sc.materials()
    assert mats[k["light"]].emittance == 5.0 and mats[k["light"]].type == pt.PT_EMITTING
    assert mats[k["polished_gold"]].type == pt.PT_PBR and mats[k["polished_gold"]].metallic == 1.0
    assert len(mats) == 4 and mats[k["first_gltf_material"]].baseColorTextureIdx == 0     # mesh materials are appended


def test_material_defaults(pt, scenes):
    sc = scenes.make_scene("cornell", res=(8, 8))
    names = sorted(scenes.scene_dict("cornell")["Materials"])
    mats = dict(zip(names, sc.materials()))
    assert len(mats) == 13
    d = mats["diffuse_white"]
    assert (d.type, d.roughness, d.metallic, d.transparency, d.indexOfRefraction) == (pt.PT_DIFFUSE, 1.0, 0.0, 0.0, 1.0)
    s = mats["specular_white"]
    assert (s.type, s.roughness, np.float32(s.metallic)) == (pt.PT_SPECULAR, 0.0, np.float32(0.2))
    r = mats["refractive_white"]
    assert (r.type, r.transparency, r.indexOfRefraction) == (pt.PT_REFRACTIVE, 1.0, 1.5)
    m = mats["milk"]
    assert m.type == pt.PT_PBR and m.subsurfaceEnabled == 1 and np.float32(m.transparency) == np.float32(0.2)
    assert np.float32(m.indexOfRefraction) == np.float32(1.35) and np.float32(m.subsurfaceScale) == np.float32(3.0)
    assert np.allclose(m.subsurfaceRadius, np.linalg.norm([0.10, 0.08, 0.05]) / 1.732, rtol=1e-6)
    g = mats["pbr_glass"]
    assert g.subsurfaceEnabled == 0 and list(g.subsurfaceColor) == [1.0, 1.0, 1.0] and np.float32(g.subsurfaceRadius) == np.float32(0.01)
    for mm in mats.values():
        assert mm.baseColorTextureIdx == -1 and mm.emissiveTextureIdx == -1


def test_camera_setup_matches_recorded_values(pt, scenes):
    k = KAT["camera_setup_eye_0_5_10p5"]
    sc = scenes.make_scene("cornell", res=(800, 800))
    cam = sc.desc.contents.camera
    assert np.allclose(list(cam.position), k["position"], atol=1e-6)
    assert np.allclose(list(cam.view), k["view"], atol=1e-9) and cam.view[1] > 0
    assert np.allclose(list(cam.right), k["right"], atol=1e-9)
    assert np.allclose(list(cam.up), k["up"], atol=1e-9)
    # FOVY 45 -> yscaled = tan(45 deg) = 1: pixelLength = 2 / resolution
    assert np.allclose(list(cam.pixelLength), [2 / 800, 2 / 800], rtol=1e-6)
    sc.set_resolution(1920, 1080)
    cam = sc.desc.contents.camera
    assert list(cam.resolution) == [1920, 1080]
    assert np.allclose(list(cam.pixelLength), [2 * (1920 / 1080) / 1920, 2 / 1080], rtol=1e-6)


def test_transforms(pt):
    M, I, IT = pt.build_transforms((1, 2, 3), (30, 45, 60), (2, 3, 4))
    A = M.T.astype(np.float64)                  # row-major
    assert np.allclose(A @ I.T.astype(np.float64), np.eye(4), atol=1e-5)
    assert np.allclose(IT.T.astype(np.float64), np.linalg.inv(A).T, atol=1e-5)
    # T * Rx * Ry * Rz * S
    rx, ry, rz = np.radians([30, 45, 60])
    Rx = np.array([[1, 0, 0], [0, np.cos(rx), -np.sin(rx)], [0, np.sin(rx), np.cos(rx)]])
    Ry = np.array([[np.cos(ry), 0, np.sin(ry)], [0, 1, 0], [-np.sin(ry), 0, np.cos(ry)]])
    Rz = np.array([[np.cos(rz), -np.sin(rz), 0], [np.sin(rz), np.cos(rz), 0], [0, 0, 1]])
    assert np.allclose(A[:3, :3], Rx @ Ry @ Rz @ np.diag([2, 3, 4]), atol=1e-5)
    assert np.allclose(A[:3, 3], [1, 2, 3])
    Mi, Ii, _ = pt.build_transforms((0, 0, 0), (0, 0, 0), (1, 1, 1))
    assert np.array_equal(Mi, np.eye(4, dtype=np.float32)) and np.array_equal(Ii, np.eye(4, dtype=np.float32))


def test_loader_errors(pt, tmp_path):
    with pytest.raises(pt.PtError):
        pt.Scene(tmp_path / "missing.json")
    bad = tmp_path / "bad.json"
    bad.write_text("{ \"Materials\": { ")
    with pytest.raises(pt.PtError):
        pt.Scene(bad)
    nocam = tmp_path / "nocam.json"
    nocam.write_text(json.dumps({"Materials": {"a": {"TYPE": "Diffuse", "RGB": [1, 1, 1]}}, "Objects": []}))
    with pytest.raises(pt.PtError, match="Camera"):
        pt.Scene(nocam)


def test_bvh_builder_equals_oracle_restatement(pt, scenes, orc):
    sc = scenes.make_scene("dragon", res=(8, 8), grid=(96, 97), env_size=(8, 4), tex_size=4)
    m = sc.desc.contents.meshes[0]
    tris = np.frombuffer(C.string_at(m.triangles, m.triangleCount * 100), dtype=pt.TRI_DTYPE)
    n1, i1, s1 = pt.bvh_build(tris, 24)
    n2, i2, s2 = orc.bvh_build(pt, tris, 24)
    assert s1 == s2 and np.array_equal(i1, i2) and n1.tobytes() == n2.tobytes()
    leaf = n1["leftChild"] == -1
    assert n1["triangleCount"][leaf].max() <= 4 and sorted(i1.tolist()) == list(range(len(tris)))
    inner = ~leaf
    assert np.all(n1["leftChild"][inner] == np.nonzero(inner)[0] + 1)        # depth-first pre-order
    # depth cap: every triangle at the same centroid cannot be split
    same = np.repeat(tris[:1], 50)
    n3, i3, s3 = pt.bvh_build(same, 24)
    assert s3["leaves"] >= 1 and sorted(i3.tolist()) == list(range(50))
    n4, i4, s4 = pt.bvh_build(tris[:0], 24)
    assert s4["nodes"] == 1 and len(i4) == 0


def test_bvh_closest_hit_equals_brute_force(pt, scenes, orc):
    sc = scenes.make_scene("dragon", res=(8, 8), grid=(40, 41), env_size=(8, 4), tex_size=4)
    o = orc.Oracle(pt, sc)
    rng = np.random.RandomState(2)
    v = rng.normal(size=(3000, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    ro = (np.array([0, 5, 0]) + 14 * v).astype(np.float32)
    d = (np.array([0, 5, 0]) + rng.uniform(-5, 5, (3000, 3)) - ro); d /= np.linalg.norm(d, axis=1, keepdims=True)
    a, b = o.intersect(ro, d.astype(np.float32)), o.intersect(ro, d.astype(np.float32), linear=True)
    assert np.array_equal(a["triangle"], b["triangle"]) and a["t"].tobytes() == b["t"].tobytes()
    assert (a["triangle"] >= 0).sum() > 300


def test_device_bvh_layout_is_valid_and_tighter_than_the_host_topology(pt, scenes):
    """The 4-wide hierarchy the device walks, built over boxes around single triangles, sibling pairs of which share
    a leaf (the host BVH's leaves only decide which hits count): every triangle referenced once, one or two triangles
    per leaf, child boxes hold what is below them, ranks are a permutation; its summed relative node area (the SAH
    measure of expected box tests) is not worse than the host topology's although it holds 4 children per node and
    goes far deeper in leaves."""
    for tris in (scenes.blob_mesh(64, 65, 3, ((0, 0, 0), (0, 0, 0), (1, 1, 1)), [0, 1], bumps=8, amplitude=0.4),
                 scenes.cube_mesh(((0, 0, 0), (0, 0, 0), (1, 1, 1)), 0),
                 scenes.blob_mesh(9, 7, 4, ((1, 2, 3), (10, 20, 30), (2, 1, 0.5)), [0])):
        st = pt.bvh_device_layout(tris)
        _, _, host = pt.bvh_build(tris)
        assert st.trianglesReferenced == len(tris)
        assert (len(tris) + 1) // 2 <= st.leaves <= len(tris) and st.leaves >= host["leaves"]
        assert 1 <= st.wideDepth <= host["max_depth"] + 1
        assert st.wideNodes >= 1
        assert st.sahCostDevice <= st.sahCostHost * 1.05 + 1.0


def test_scene_fingerprint_ignores_the_camera_pose_only(pt, scenes):
    """pt_scene_fingerprint: equal for two descriptions that differ in where the camera is, different as soon as a
    triangle, a material, a texture byte, the resolution or the trace depth differs (what decides whether the shims take a
    parked context back on pathtraceInit)."""
    def fp(edit=None, **kw):
        sc = scenes.make_scene("dragon", res=kw.pop("res", (64, 48)), depth=kw.pop("depth", 8), grid=(20, 21), env_size=(32, 16), tex_size=8)
        if edit:
            edit(sc)
        return pt.scene_fingerprint(sc.desc)

    base = fp()
    assert base != 0 and base == fp() and pt.scene_fingerprint(None) == 0

    def move(sc):
        cam = sc.desc.contents.camera
        cam.position[0] += 1.5
        cam.view[1] = 0.25

    assert fp(move) == base
    assert fp(res=(64, 40)) != base and fp(depth=7) != base

    def recolour(sc):
        sc.desc.contents.materials[1].color[0] = 0.123

    def nudge(sc):
        m = sc.desc.contents.meshes[0]
        f = C.cast(m.triangles, C.POINTER(C.c_float))       # 25 floats per 100-byte triangle
        f[5 * 25 + 4] += 1e-3

    assert fp(recolour) != base and fp(nudge) != base
    assert fp(lambda sc: sc.add_texture(np.zeros((4, 4, 3), np.uint8))) != base


def test_tree_passes_setting_trades_set_up_for_area_only(pt, scenes):
    """pt_set_tree_passes: the reinsertion rounds of the walked tree lower its summed node area and change nothing else
    about a valid layout (every triangle referenced once); out-of-range values are refused."""
    tris = scenes.blob_mesh(64, 65, 3, ((0, 0, 0), (0, 0, 0), (1, 1, 1)), [0, 1], bumps=8, amplitude=0.4)
    try:
        pt.set_tree_passes(0)
        plain = pt.bvh_device_layout(tris)
        pt.set_tree_passes(3)
        opt = pt.bvh_device_layout(tris)
    finally:
        pt.set_tree_passes(3)
    assert plain.trianglesReferenced == opt.trianglesReferenced == len(tris)
    assert opt.sahCostDevice < plain.sahCostDevice
    for bad in (-1, 17):
        with pytest.raises(pt.PtError):
            pt.set_tree_passes(bad)


def test_det_math_close_to_libm(orc):
    rng = np.random.RandomState(0)
    cases = [("sin", rng.uniform(-7, 7, 20000)), ("cos", rng.uniform(-7, 7, 20000)), ("acos", rng.uniform(-1, 1, 20000)),
             ("exp", rng.uniform(-40, 5, 20000)), ("log", rng.uniform(0.05, 4, 20000)), ("pow5", rng.uniform(0, 1, 20000))]
    for fn, x in cases:
        x = x.astype(np.float32)
        a, b = orc.math_eval(orc.MATH_DET, fn, x), orc.math_eval(orc.MATH_LIBM, fn, x)
        scale = np.maximum(np.abs(b), 1e-3 if fn in ("sin", "cos", "acos") else 1e-30)
        assert np.max(np.abs(a - b) / scale) < 2e-6, fn
    y, x = rng.uniform(-3, 3, 20000).astype(np.float32), rng.uniform(-3, 3, 20000).astype(np.float32)
    a, b = orc.math_eval(orc.MATH_DET, "atan2", y, x), orc.math_eval(orc.MATH_LIBM, "atan2", y, x)
    assert np.max(np.abs(a - b)) < 1e-6


def test_det_mode_radiance_close_to_libm_mode(pt, scenes, orc):
    """The oracle in the GPU's arithmetic (det) vs the reference's (libm): individual paths may flip a branch,
    the image stays within a small RMSE of the mean radiance."""
    sc = scenes.make_scene("cornell", res=(64, 64), depth=8)
    a, b = orc.Oracle(pt, sc, orc.MATH_DET), orc.Oracle(pt, sc, orc.MATH_LIBM)
    spp = 12
    for it in range(1, spp + 1):
        a.trace(it)
        b.trace(it)
    ia, ib = a.image() / spp, b.image() / spp
    mean = ib.reshape(-1, 3).mean(0)
    rmse = np.sqrt(((ia - ib) ** 2).reshape(-1, 3).mean(0))
    assert np.all(rmse / mean < 0.03), rmse / mean
    assert abs(a.stats().pathBounces / b.stats().pathBounces - 1) < 2e-3


def test_oracle_options_and_edge_scenes(pt, scenes, orc):
    sc = scenes.make_scene("sss", res=(48, 48), depth=8)
    o = orc.Oracle(pt, sc)
    o.trace(1)
    live = list(o.stats().lastLive[:9])
    assert live[0] == 48 * 48 and live == sorted(live, reverse=True) and o.stats().lastDepth <= 8
    assert np.isfinite(o.image()).all() and o.image().sum() > 0
    # sort off changes slot numbering (and so the RNG streams) but not the estimator
    o2 = orc.Oracle(pt, sc)
    o2.set_option(pt.PT_OPT_MATERIAL_SORT, 0)
    for it in range(1, 9):
        o2.trace(it)
    for it in range(2, 9):
        o.trace(it)
    a, b = o.image().reshape(-1, 3).mean(0), o2.image().reshape(-1, 3).mean(0)
    assert np.allclose(a, b, rtol=0.08)
    disp = o.display(8)
    assert disp.shape == (48, 48, 4) and disp[..., 3].max() == 0 and disp[..., :3].max() > 0
    # depth 1: every non-emissive path is zeroed on its first bounce
    sc1 = scenes.make_scene("cornell", res=(32, 32), depth=1)
    o3 = orc.Oracle(pt, sc1)
    o3.trace(1)
    assert list(o3.stats().lastLive[:2]) == [1024, 0]


REFERENCE_SCENES = "/root/reference/scenes"


@pytest.mark.skipif(not os.path.isdir(REFERENCE_SCENES), reason="reference scene files are not present on this machine")
def test_every_reference_scene_file_loads(pt):
    """All scene files the reference ships go through the loader as they are: every object and material arrives, the
    camera block is read, and assets the reference does not distribute (its .glb models, its environment maps) leave
    a line in the log instead of an error, as the reference carries on without them (src/scene.cpp:224, :827)."""
    import glob
    files = sorted(glob.glob(os.path.join(REFERENCE_SCENES, "*.json")))
    assert len(files) >= 13
    for f in files:
        j = json.load(open(f))
        sc = pt.Scene(f)
        assert len(sc.geoms()) == len(j["Objects"]), f
        assert sc.resolution == tuple(j["Camera"]["RES"]) and sc.depth == j["Camera"]["DEPTH"], f
        extra = len(sc.materials()) - len(j["Materials"])
        gltf = [o for o in j["Objects"] if o["TYPE"] == "gltf"]
        missing = [o for o in gltf if not os.path.exists(os.path.join(REFERENCE_SCENES, o["FILE"]))]
        assert sc.log.count("glTF object left empty") == len(missing), (f, sc.log)
        assert extra >= 0 and (extra == 0 or len(gltf) > len(missing)), f          # only an imported model adds materials
        names = sorted(j["Materials"])                                              # std::map order, src/scene.cpp:660
        types = dict(Diffuse=pt.PT_DIFFUSE, Emitting=pt.PT_EMITTING, Specular=pt.PT_SPECULAR, Refractive=pt.PT_REFRACTIVE, PBR=pt.PT_PBR)
        for m, name in zip(sc.materials(), names):
            assert m.type == types[j["Materials"][name]["TYPE"]], (f, name)
        if "EnvironmentMap" in j:
            present = os.path.exists(os.path.join(REFERENCE_SCENES, j["EnvironmentMap"]["FILE"]))
            assert bool(sc.desc.contents.env.enabled) == present
            assert ("environment map not loaded" in sc.log) == (not present)
