"""The CPU oracle against the known-answer values recorded from the reference (tests/golden)."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))


def _utilhash(a):
    """src/intersections.h:13-22 on 32-bit unsigned integers"""
    m = 0xffffffff
    a = ((a + 0x7ed55d16) + (a << 12)) & m
    a = ((a ^ 0xc761c23c) ^ (a >> 19)) & m
    a = ((a + 0x165667b1) + (a << 5)) & m
    a = ((a + 0xd3a2646c) ^ (a << 9)) & m
    a = ((a + 0xfd7046c5) + (a << 3)) & m
    a = ((a ^ 0xb55a4f09) ^ (a >> 16)) & m
    return a


def _stream(it, index, depth, n):
    """makeSeededRandomEngine (src/pathtrace.cu:138-143) + thrust::minstd_rand + uniform_real_distribution<float>,
    restated here in Python integers so that the order of the draws does not rest on the oracle alone."""
    h = _utilhash(((1 << 31) | (depth << 22) | it) & 0xffffffff) ^ _utilhash(index)
    x = h % 2147483647 or 1
    out = []
    for _ in range(n):
        x = (48271 * x) % 2147483647
        out.append(np.float32(x - 1) / (np.float32(1.0) + np.float32(2147483645)))
    return np.asarray(out, np.float32)


def test_rng_stream(orc):
    k = KAT["rng"]
    u = orc.rng_u01(k["iter"], k["index"], k["depth"], 4)
    printed = [np.float32(x) for x in k["first_two_u01_as_printed"]]
    assert u[:2].tolist() == printed[::-1]          # stream order: see "_order" in the golden file
    assert np.array_equal(u, _stream(k["iter"], k["index"], k["depth"], 4))
    for it, index, depth in ((7, 123456, 3), (5000, 1439999, 8), (1, 0, 64)):
        assert np.array_equal(orc.rng_u01(it, index, depth, 16), _stream(it, index, depth, 16))
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


def test_sss_200_path_bounces_both_evaluation_orders(pt, scenes, orc):
    """The second whole-loop pin: subsurface scattering, the PBR transmission branch and Russian roulette on
    scenes/sss.json.  The recorded host builds of the reference differ in one thing, the order in which a constructor's
    three arguments draw from the stream (left to right under clang / nvcc, right to left under g++): the oracle
    reproduces both totals."""
    k = KAT["sss_200x200_depth8_iterations_1_to_4"]
    sc = scenes.make_scene("sss", res=(200, 200), depth=8)
    for opt, want in ((0, k["path_bounces_left_to_right"]), (1, k["path_bounces_right_to_left"])):
        o = orc.Oracle(pt, sc, orc.MATH_LIBM)
        o.set_option(orc.OPT_SCATTER_RIGHT_TO_LEFT, opt)
        for it in (1, 2, 3, 4):
            o.trace(it)
        assert o.stats().pathBounces == want, (opt, o.stats().pathBounces)
        o.close()


def test_struct_sizes(pt):
    k = KAT["struct_sizes"]
    assert pt.TRI_DTYPE.itemsize == k["Triangle"]
    assert pt.NODE_DTYPE.itemsize == k["BVHNode"]
    assert C.sizeof(pt.PtMaterial) == k["Material"]
    assert C.sizeof(pt.PtCamera) == k["Camera"]


# The Duck.glb known answers (triangle count, BVH node / leaf / depth counts) are checked in test_gltf_import.py,
# through the library's own glTF import of the reference's scenes/duck.json.


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
