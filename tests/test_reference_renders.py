"""The reference's own finished renders as a pin on the shading half of the loop.

The reference holds no vectors for its shaders, and `src/pathtrace.cu` cannot be built here; but its write-up shows
1200 x 1200 renders of scene files that ship with it and need no missing asset (tests/golden/make_reference_renders.py
lists them).  A converged render is a statistic of everything on the path — camera rays, intersection, the diffuse /
specular / PBR / subsurface shaders, light sampling and its weights, Russian roulette, gather, the 8-bit conversion — so
these tests render the same scenes and compare 8 x 8 block means and whole-image means with the stored images.

Bars (8-bit levels, stated per test): the reference's images are finite-sample renders themselves, so the block RMSE
floors at their own noise; the whole-image mean does not.
"""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_renders.npz")
REF = "/root/reference"
BLOCK = 8

# render key -> (scene name in scenes.py, reference scene file, materials edit)
RENDERS = {
    "sss_1": ("sss", "sss.json", None),
    "sss_0": ("sss", "sss.json", "subsurface_off"),
    "diffuse_orig": ("cornell_orig", "cornell-orig.json", None),
    "ssaa_1024": ("ssaa", "ssaa.json", None),
}


def _gold():
    return np.load(GOLD)


def _scene_path(scenes, key, directory):
    name, _, edit = RENDERS[key]
    d = scenes.scene_dict(name)
    if edit == "subsurface_off":
        for m in d["Materials"].values():
            if "SUBSURFACE_ENABLED" in m:
                m["SUBSURFACE_ENABLED"] = 0
    path = os.path.join(str(directory), key + ".json")
    with open(path, "w") as f:
        json.dump(d, f)
    return path


def _block_means(rgb8):
    h, w, c = rgb8.shape
    return rgb8.astype(np.float64).reshape(h // BLOCK, BLOCK, w // BLOCK, BLOCK, c).mean((1, 3))


def _compare(pt, image_sum, samples, gold, key):
    rgb8 = np.asarray(pt.image_to_rgb8(image_sum, samples))
    ref_blocks = gold[key + "_sum8"].astype(np.float64) / (BLOCK * BLOCK)
    d = _block_means(rgb8) - ref_blocks
    rmse = np.sqrt((d ** 2).mean((0, 1)))
    mean = rgb8.reshape(-1, 3).mean(0)
    ref_mean = gold[key + "_mean"]
    return rmse, mean, ref_mean


def test_fixture_matches_the_reference_files_and_scenes(pt, scenes, tmp_path):
    """Where the reference tree is present: the stored block sums are those of its files, and the scene descriptions
    rendered here are the ones its own scene files give (geoms, materials, camera, byte for byte)."""
    if not os.path.isdir(REF):
        pytest.skip("reference tree not present")
    import hashlib
    import ctypes as C
    gold = _gold()
    names = {"sss_1": "sss-1.png", "sss_0": "sss-0.png", "diffuse_orig": "diffuse-orig.png", "ssaa_1024": "ssaa-1024.png"}
    for key, fn in names.items():
        data = open(os.path.join(REF, "img", fn), "rb").read()
        assert hashlib.sha256(data).hexdigest() == str(gold[key + "_sha256"])
        rgb = np.asarray(pt.decode_png(data))[..., :3]
        sums = rgb.astype(np.uint32).reshape(150, BLOCK, 150, BLOCK, 3).sum((1, 3))
        assert (sums == gold[key + "_sum8"]).all()

    def blob(sc):
        d = sc.desc.contents
        return (C.string_at(d.geoms, d.geomCount * C.sizeof(pt.PtGeom)), C.string_at(d.materials, d.materialCount * C.sizeof(pt.PtMaterial)),
                bytes(d.camera), int(d.traceDepth))

    for key, (name, ref_file, edit) in RENDERS.items():
        if edit:
            continue
        mine = pt.Scene(_scene_path(scenes, key, tmp_path))
        theirs = pt.Scene(os.path.join(REF, "scenes", ref_file))
        assert blob(mine) == blob(theirs), key


def test_oracle_render_matches_reference_image_sss(pt, scenes, orc, tmp_path):
    """oracle (libm arithmetic) vs img/sss-1.png: 64 iterations of scenes/sss.json at 1200 x 1200.  Bars: whole-image
    mean within 1.5 % per channel (measured 0.04 / 0.3 / 0.35 %), 8 x 8 block RMSE <= 9 levels (measured 6: the noise
    of 64 samples; it falls as 1/sqrt(samples), see the GPU test)."""
    gold = _gold()
    sc = pt.Scene(_scene_path(scenes, "sss_1", tmp_path))
    o = orc.Oracle(pt, sc, orc.MATH_LIBM)
    spp = 64
    for it in range(1, spp + 1):
        o.trace(it)
    rmse, mean, ref_mean = _compare(pt, o.image(), spp, gold, "sss_1")
    o.close()
    print("sss_1 oracle", spp, "spp: block rmse", rmse, "mean", mean, "reference", ref_mean)
    assert (np.abs(mean / ref_mean - 1.0) <= 0.015).all(), (mean, ref_mean)
    assert (rmse <= 9.0).all(), rmse


# spp, trace depth (None: the scene file's), relative bar on the whole-image mean, bar on the 8 x 8 block RMSE (8-bit levels)
#
# sss-1 / sss-0 were made by the loop as it stands today (subsurface scattering is its last feature).  Their pixel noise
# (0.3 levels on the walls; 2048 samples here leave 3.6) says they went through the reference's denoiser, and 46 % of
# this scene's unclamped energy sits in pixels that the 8-bit conversion clamps, so the clamped mean still creeps up
# with the sample count (64 / 256 / 2048 samples: 1.000 / 1.012 / 1.016 of the reference's): measured 1.4-1.8 % above,
# block RMSE 1.7-2.1; bars 2.5 % and 3 levels.
#
# diffuse-orig and ssaa-1024 illustrate the write-up's first sections (README.md:68-77, :140-158) and carry the noise of
# an undenoised 5000-sample render.  They predate next-event estimation: in the plain diffuse shader described there
# light is gathered only when a path runs into the emitter, so DEPTH 8 ends a light path after 8 segments, where
# today's loop adds the light sample of the last vertex as a ninth.  Rendered at the scene files' depth 8 today's loop is
# uniformly 4.6 % / 3.7 % brighter than those two images (most where the light is indirect); at depth 7 — the same
# longest light path — it is within 1.4 % (2.5 % in blue) / 0.2 % (1 % in blue).  They are kept as a second, weaker
# pin at depth 7: bars 3 % and 4 levels.
GPU_CASES = {
    "sss_1": (2048, None, 0.025, 3.0),
    "sss_0": (2048, None, 0.025, 3.0),
    "diffuse_orig": (2048, 7, 0.03, 4.0),
    "ssaa_1024": (2048, 7, 0.03, 4.0),
}


@pytest.mark.gpu
@pytest.mark.parametrize("key", list(GPU_CASES))
def test_gpu_render_matches_reference_image(pt, scenes, gpu_ok, tmp_path, key):
    """GPU render through the C ABI vs the reference's own image of the same scene."""
    gold = _gold()
    spp, depth, mean_bar, rmse_bar = GPU_CASES[key]
    sc = pt.Scene(_scene_path(scenes, key, tmp_path))
    assert sc.resolution == (1200, 1200)
    if depth is not None:
        sc.set_depth(depth)
    tr = pt.Tracer(sc, device=0)
    tr.set_option(pt.PT_OPT_BATCH, 64)
    tr.trace_range(1, spp + 1)
    tr.sync()
    rmse, mean, ref_mean = _compare(pt, tr.image(), spp, gold, key)
    tr.close()
    print(key, "gpu", spp, "spp: block rmse", rmse, "mean", mean, "reference", ref_mean, "ratio", mean / ref_mean)
    assert (np.abs(mean / ref_mean - 1.0) <= mean_bar).all(), (mean, ref_mean)
    assert (rmse <= rmse_bar).all(), rmse
