"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bars:
  * arithmetic primitives (div, sqrt, sin/cos/acos/atan2/exp/log)  bit-exact vs the oracle's det mode
  * camera rays                                                     bit-exact
  * closest hit on fixed ray sets (geom, triangle, t, normal, uv)   bit-exact
  * whole iterations (image sum, live-path curve, path-bounce total) bit-exact vs the oracle's det mode
  * accumulated radiance vs the oracle's libm mode (the arithmetic a host build of the
    reference performs): per-channel RMSE <= RMSE_TOL of the mean radiance after SPP samples
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RMSE_TOL = 0.02     # per-channel RMSE of (gpu - oracle_libm) / mean radiance over the image, 16 spp at 96x96


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _same_bits(a, b):
    a, b = _bits(a), _bits(b)
    nan = (np.isnan(a.view(np.float32)) & np.isnan(b.view(np.float32)))
    return bool(np.all((a == b) | nan))


def _ray_set(seed, n, center=(0.0, 5.0, 0.0), radius=9.0):
    """Seeded rays: origins on a sphere around the scene, directions towards a jittered interior point."""
    rng = np.random.RandomState(seed)
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    o = np.asarray(center) + radius * v
    tgt = np.asarray(center) + rng.uniform(-4, 4, size=(n, 3))
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o.astype(np.float32), d.astype(np.float32)


def _hits_equal(g, o):
    for f in ("geom", "triangle", "materialId"):
        assert np.array_equal(g[f], o[f]), f
    assert _same_bits(g["t"], o["t"])
    hit = o["t"] > 0
    assert _same_bits(g["normal"][hit], o["normal"][hit])
    assert _same_bits(g["uv"][hit], o["uv"][hit])


@pytest.mark.parametrize("fn,lo,hi", [("sin", -7.0, 7.0), ("cos", -7.0, 7.0), ("acos", -1.0, 1.0), ("exp", -90.0, 10.0),
                                      ("log", 1e-3, 10.0), ("pow5", 0.0, 1.0), ("sqrt", 0.0, 100.0)])
def test_math_unary_bit_exact(pt, orc, gpu_ok, fn, lo, hi):
    rng = np.random.RandomState(5)
    x = rng.uniform(lo, hi, 1 << 16).astype(np.float32)
    x[:8] = [lo, hi, 0.0, 1.0, -1.0 if lo < 0 else 0.5, 0.5, 0.25, (lo + hi) / 2]
    x = np.clip(x, lo, hi).astype(np.float32)
    g = pt.math_eval(orc.FN[fn], x)
    o = orc.math_eval(orc.MATH_DET, fn, x)
    assert _same_bits(g, o)


@pytest.mark.parametrize("fn", ["atan2", "div"])
def test_math_binary_bit_exact(pt, orc, gpu_ok, fn):
    rng = np.random.RandomState(6)
    a = rng.uniform(-3, 3, 1 << 16).astype(np.float32)
    b = rng.uniform(-3, 3, 1 << 16).astype(np.float32)
    a[:4] = [0.0, 0.0, 1.0, -1.0]
    b[:4] = [1.0, -1.0, 0.0, 0.0]
    b[b == 0] = np.float32(0.5) if fn == "div" else b[b == 0]
    g = pt.math_eval(orc.FN[fn], a, b)
    o = orc.math_eval(orc.MATH_DET, fn, a, b)
    assert _same_bits(g, o)


def test_camera_rays_bit_exact(pt, scenes, orc, gpu_ok):
    sc = scenes.make_scene("cornell", res=(200, 120), depth=8)
    tr = pt.Tracer(sc)
    o = orc.Oracle(pt, sc)
    for it in (1, 2, 3, 4, 7):
        go, gd = tr.camera_rays(it)
        oo, od = o.camera_rays(it)
        assert _same_bits(go, oo) and _same_bits(gd, od)


@pytest.mark.parametrize("name", ["cornell", "sss", "pbr", "ssaa"])
def test_closest_hit_primitives_bit_exact(pt, scenes, orc, gpu_ok, name):
    sc = scenes.make_scene(name, res=(64, 64))
    tr = pt.Tracer(sc)
    o = orc.Oracle(pt, sc)
    ro, rd = _ray_set(11, 20000)
    # axis-parallel rays exercise the unguarded division by zero in the slab test
    ro[:6] = [[0, 5, 9], [0, 5, 9], [3, 5, 0], [0, 9.9, 0], [0, 5, 0], [0, 5, 0]]
    rd[:6] = [[0, 0, -1], [0, 1, 0], [-1, 0, 0], [0, -1, 0], [1, 0, 0], [0, 0, 1]]
    _hits_equal(tr.intersect(ro, rd), o.intersect(ro, rd))


def test_closest_hit_mesh_bit_exact(pt, scenes, orc, gpu_ok):
    sc = scenes.make_scene("dragon", res=(64, 64), grid=(64, 65), env_size=(64, 32), tex_size=16)
    tr = pt.Tracer(sc)
    o = orc.Oracle(pt, sc)
    ro, rd = _ray_set(3, 50000, center=(0.0, 4.0, 0.0), radius=12.0)
    g = tr.intersect(ro, rd)
    ob = o.intersect(ro, rd)
    _hits_equal(g, ob)
    assert (g["triangle"] >= 0).sum() > 5000
    # the BVH path agrees with brute force over every triangle (topology-independent check)
    lin = o.intersect(ro[:4000], rd[:4000], linear=True)
    assert np.array_equal(lin["triangle"], ob["triangle"][:4000]) and _same_bits(lin["t"], ob["t"][:4000])
    # USE_BVH = 0: linear fallback on the device
    tr.set_option(pt.PT_OPT_USE_BVH, 0)
    _hits_equal(tr.intersect(ro[:4000], rd[:4000]), lin)


def test_closest_hit_cube_mesh_ties(pt, scenes, orc, gpu_ok):
    """12-triangle axis-aligned cube: rays through shared edges / vertices give exact-distance ties."""
    sc = scenes.make_scene("cornell_orig", res=(32, 32))
    trs = ((0.0, 5.0, 0.0), (0.0, 0.0, 0.0), (2.0, 2.0, 2.0))
    mid = sc.add_material(type=pt.PT_PBR, color=[0.8, 0.8, 0.8], roughness=0.5, metallic=0.0, indexOfRefraction=1.5)
    sc.add_mesh(scenes.cube_mesh(trs, mid))      # vertices pre-placed, identity object transform
    tr = pt.Tracer(sc)
    o = orc.Oracle(pt, sc)
    xs = np.linspace(-1.0, 1.0, 41, dtype=np.float32)
    gx, gy = np.meshgrid(xs, xs)
    n = gx.size
    ro = np.stack([gx.ravel(), 5 + gy.ravel(), np.full(n, 9.0, np.float32)], 1).astype(np.float32)
    rd = np.tile(np.array([[0, 0, -1]], np.float32), (n, 1))
    # diagonal rays through the cube's corners and edge midpoints
    ro2 = np.array([[0, 5, 9], [4, 9, 4], [-4, 5, 4], [0, 9, 0]], np.float32)
    rd2 = np.array([[0, 0, -1], [-1, -1, -1], [1, 0, -1], [0, -1, 0]], np.float32)
    rd2 /= np.linalg.norm(rd2, axis=1, keepdims=True)
    ro, rd = np.concatenate([ro, ro2]), np.concatenate([rd, rd2])
    g, ob = tr.intersect(ro, rd), o.intersect(ro, rd)
    _hits_equal(g, ob)
    assert (ob["triangle"] >= 0).sum() > 100


def _render_pair(pt, orc, sc, iters, math, sort=1):
    tr = pt.Tracer(sc)
    tr.set_option(pt.PT_OPT_MATERIAL_SORT, sort)
    o = orc.Oracle(pt, sc, math)
    o.set_option(pt.PT_OPT_MATERIAL_SORT, sort)
    curves = []
    for it in iters:
        tr.trace(it)
        o.trace(it)
        gs, os_ = tr.stats(), o.stats()
        curves.append((list(gs.lastLive[:sc.depth + 1]), list(os_.lastLive[:sc.depth + 1]), gs.lastDepth, os_.lastDepth))
    return tr, o, curves


def _assert_full_parity(pt, orc, sc, iters, sort=1, lit=True):
    tr, o, curves = _render_pair(pt, orc, sc, iters, orc.MATH_DET, sort)
    for g, ob, gd, od in curves:
        assert g == ob, (g, ob)
        assert gd == od
    gs, os_ = tr.stats(), o.stats()
    assert gs.pathBounces == os_.pathBounces
    gi, oi = tr.image(), o.image()
    assert _same_bits(gi, oi), "max abs diff %g" % np.nanmax(np.abs(gi - oi))
    if lit is not None:
        assert (float(oi.sum()) > 0) == lit
    return tr, o


@pytest.mark.parametrize("name,res,depth", [("cornell", (128, 128), 8), ("cornell", (97, 61), 12), ("sss", (96, 96), 8),
                                           ("pbr", (96, 96), 8), ("ssaa", (96, 96), 8), ("cornell_orig", (64, 64), 8),
                                           ("spheres", (64, 64), 8), ("sss", (64, 64), 32), ("cornell", (40, 40), 64)])
def test_iterations_bit_exact_primitives(pt, scenes, orc, gpu_ok, name, res, depth):
    sc = scenes.make_scene(name, res=res, depth=depth)
    _assert_full_parity(pt, orc, sc, [1, 2, 3, 4, 5], lit=name != "spheres")    # no light and no env map: black


def test_iterations_bit_exact_env_map(pt, scenes, orc, gpu_ok):
    """All three MIS strategies (light / BRDF / environment CDF sampling) and miss shading."""
    for name in ("pbr", "cornell", "spheres"):
        sc = scenes.make_scene(name, res=(96, 96), depth=8, env=True, env_size=(256, 128))
        _assert_full_parity(pt, orc, sc, [1, 2, 3])


def test_iterations_bit_exact_mesh_textures_env(pt, scenes, orc, gpu_ok):
    """Mesh + BVH shadow rays + RGBA / metallic-roughness textures + transmissive material + env map."""
    sc = scenes.make_scene("chess", res=(96, 96), depth=10, grid=(48, 49), env_size=(128, 64), tex_size=32)
    _assert_full_parity(pt, orc, sc, [1, 2, 3])
    sc = scenes.make_scene("dragon", res=(128, 72), depth=8, grid=(64, 65), env_size=(128, 64), tex_size=32)
    _assert_full_parity(pt, orc, sc, [1, 2])
    sc = scenes.make_scene("halo", res=(64, 64), depth=8, grid=(40, 41), env_size=(64, 32), tex_size=16)
    _assert_full_parity(pt, orc, sc, [1, 2])
    sc = scenes.make_scene("duck", res=(64, 64), depth=8, env_size=(64, 32), tex_size=16)   # no light geom at all
    _assert_full_parity(pt, orc, sc, [1, 2])


def test_ragged_texture_and_env_sizes_all_channel_counts(pt, scenes, orc, gpu_ok):
    """Textures whose sides are no multiple of the device tile (8 x 4 texels), with one, two, three and four channels, and
    environment maps whose width is no multiple of the column-search block (32) or smaller than one block: the device
    layouts differ from the image files', the values fetched must not.  Materials whose two textures have one size are
    sampled through the paired layout, the others one texture at a time."""
    rng = np.random.RandomState(5)
    for env_size, tex_sides in (((50, 23), ((13, 7), (9, 5), (1, 1), (33, 2))), ((97, 40), ((8, 4), (17, 31), (5, 64), (64, 3))),
                                ((31, 16), ((2, 2), (7, 9), (16, 16), (3, 40)))):
        sc = scenes.make_scene("dragon", res=(96, 64), depth=8, grid=(40, 41), env_size=env_size, tex_size=16)
        ids = []
        for k, (comp, (w, h)) in enumerate(zip((1, 2, 3, 4), tex_sides)):
            img = rng.randint(0, 256, (h, w, comp)).astype(np.uint8)
            base = sc.add_texture(img[..., 0] if comp == 1 else img)
            # metallic-roughness of the base colour's size for two of the four (the device then keeps the pair side by side), of another size for the rest
            mr = sc.add_texture(rng.randint(0, 256, (h, w, 3) if k % 2 == 0 else (h + 1, w + 2, 3)).astype(np.uint8))
            ids.append(sc.add_material(type=pt.PT_PBR, color=[0.9, 0.8, 0.7], roughness=0.6, metallic=0.5 * (k % 2), indexOfRefraction=1.5,
                                       baseColorTextureIdx=base, metallicRoughnessTextureIdx=mr))
        ident = ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
        sc.add_mesh(scenes.blob_mesh(20, 21, 3, ident, ids, radius=2.5, center=(2.5, 3.0, 1.0)))
        _assert_full_parity(pt, orc, sc, [1, 2, 3])


def _three_mesh_scene(pt, scenes, res):
    """cornell_orig + three overlapping BVH meshes (two blobs that intersect each other and a cube sunk into both):
    closest hits and shadow rays have to be folded across meshes."""
    sc = scenes.make_scene("cornell_orig", res=res, depth=8)
    ident = ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    m0 = sc.add_material(type=pt.PT_PBR, color=[0.9, 0.6, 0.3], roughness=0.4, metallic=0.0, indexOfRefraction=1.5)
    m1 = sc.add_material(type=pt.PT_PBR, color=[0.95, 0.8, 0.6], roughness=0.05, metallic=0.0, transparency=0.9, indexOfRefraction=1.5)
    m2 = sc.add_material(type=pt.PT_PBR, color=[0.7, 0.7, 0.9], roughness=0.3, metallic=1.0, indexOfRefraction=1.5)
    sc.add_mesh(scenes.blob_mesh(24, 25, 11, ident, [m0, m1], radius=2.2, center=(-1.0, 4.0, 0.0)))
    sc.add_mesh(scenes.cube_mesh(((0.2, 6.4, 0.3), (0.0, 0.0, 0.0), (1.5, 1.5, 1.5)), m2))
    sc.add_mesh(scenes.blob_mesh(20, 21, 12, ident, [m2, m0], radius=2.0, center=(1.2, 4.5, 0.4)))
    return sc


def test_several_bvh_meshes_bit_exact(pt, scenes, orc, gpu_ok):
    sc = _three_mesh_scene(pt, scenes, (72, 72))
    tr = pt.Tracer(sc)
    o = orc.Oracle(pt, sc)
    ro, rd = _ray_set(5, 30000, center=(0.0, 4.0, 0.0), radius=9.0)
    g, ob = tr.intersect(ro, rd), o.intersect(ro, rd)
    _hits_equal(g, ob)
    geoms_hit = set(np.unique(g["geom"][g["triangle"] >= 0]).tolist())
    assert len(geoms_hit) == 3, geoms_hit
    _assert_full_parity(pt, orc, sc, [1, 2, 3])


def test_use_bvh_off_iterations_bit_exact(pt, scenes, orc, gpu_ok):
    """USE_BVH 0 (src/pathtrace.cu:193): meshes are scanned linearly, in the closest-hit stage and in shadow rays."""
    sc = _three_mesh_scene(pt, scenes, (40, 40))
    tr = pt.Tracer(sc)
    tr.set_option(pt.PT_OPT_USE_BVH, 0)
    o = orc.Oracle(pt, sc, orc.MATH_DET)
    o.set_option(pt.PT_OPT_USE_BVH, 0)
    for it in (1, 2):
        tr.trace(it)
        o.trace(it)
    assert tr.stats().pathBounces == o.stats().pathBounces
    assert _same_bits(tr.image(), o.image())
    # and it is the same picture as with the BVH on
    tr2 = pt.Tracer(sc)
    for it in (1, 2):
        tr2.trace(it)
    assert _same_bits(tr.image(), tr2.image())


def test_material_sort_off_bit_exact(pt, scenes, orc, gpu_ok):
    """MATERIAL_SORTING 0 (src/pathtrace.h:6): slots keep compacted order, so RNG streams differ from sort-on."""
    sc = scenes.make_scene("cornell", res=(96, 96), depth=8)
    tr, o = _assert_full_parity(pt, orc, sc, [1, 2, 3], sort=0)
    tr2, _ = _assert_full_parity(pt, orc, sc, [1, 2, 3], sort=1)
    assert not np.array_equal(tr.image(), tr2.image())


def test_batched_iterations_bit_exact(pt, scenes, orc, gpu_ok):
    """pt_trace_range runs several iterations through the same launches (one queue each); the image must
    equal the one-iteration-at-a-time result bit for bit, for any batch size and any stride."""
    sc = scenes.make_scene("cornell", res=(96, 64), depth=8)
    o = orc.Oracle(pt, sc, orc.MATH_DET)
    for it in range(2, 25, 2):
        o.trace(it)
    ref = o.image()
    for batch in (1, 5, 8, 12):
        tr = pt.Tracer(sc)
        tr.set_option(pt.PT_OPT_BATCH, batch)
        tr.trace_range(2, 25, 2)
        assert _same_bits(tr.image(), ref), batch
        s = tr.stats()
        assert s.pathBounces == o.stats().pathBounces and s.iterations == 12
        assert list(s.lastLive[:9]) == list(o.stats().lastLive[:9])
    sc = scenes.make_scene("dragon", res=(80, 48), depth=8, grid=(40, 41), env_size=(64, 32), tex_size=16)
    o = orc.Oracle(pt, sc, orc.MATH_DET)
    for it in range(1, 8):
        o.trace(it)
    tr = pt.Tracer(sc)
    tr.trace_range(1, 8)
    assert _same_bits(tr.image(), o.image())


def test_one_iteration_calls_in_flight_bit_exact(pt, scenes, orc, gpu_ok):
    """pt_trace calls issued without waiting rotate over rows of the path state with streams of their own and overlap on
    the device; their sums must equal the one-at-a-time result, also with a wait after every call, with stage-level calls
    (which use row 0 on the main stream) in between, and with pt_clear / pt_display between calls."""
    sc = scenes.make_scene("dragon", res=(96, 64), depth=8, grid=(48, 49), env_size=(64, 32), tex_size=16)
    o = orc.Oracle(pt, sc, orc.MATH_DET)
    sums = {}
    for it in range(1, 30):
        o.trace(it)
        sums[it] = o.image().copy()
    tr = pt.Tracer(sc)
    for it in range(1, 30):
        tr.trace(it)                            # no wait: up to four iterations in flight
    assert _same_bits(tr.image(), sums[29])
    assert tr.stats().pathBounces == o.stats().pathBounces
    assert list(tr.stats().lastLive[:9]) == list(o.stats().lastLive[:9])
    tr = pt.Tracer(sc)
    for it in range(1, 12):
        tr.trace(it)
        tr.sync()                               # a caller that waits every time
    assert _same_bits(tr.image(), sums[11])
    tr = pt.Tracer(sc)
    ro, rd = _ray_set(9, 5000, center=(0.0, 4.0, 0.0), radius=9.0)
    want_hits = o.intersect(ro, rd)
    for it in range(1, 12):
        tr.trace(it)
        if it in (3, 4, 9):
            _hits_equal(tr.intersect(ro, rd), want_hits)
        if it == 6:
            tr.capture_gbuffer(1)
        if it == 7:
            tr.camera_rays(2)
    assert _same_bits(tr.image(), sums[11])
    # launch groups and single calls mixed
    tr = pt.Tracer(sc)
    tr.trace(1); tr.trace(2); tr.trace(3)
    tr.set_option(pt.PT_OPT_BATCH, 4)
    tr.trace_range(4, 10)
    tr.trace(10); tr.trace(11)
    assert _same_bits(tr.image(), sums[11])


def test_lookahead_calls_bit_exact(pt, scenes, orc, gpu_ok):
    """pt_trace_lookahead: one call per iteration in the order 1, 2, 3, ... (the reference's runCuda loop), iterations
    rendered ahead in launch groups on two halves of the path-state rows that take turns.  After call k the running sum —
    on the device and in the bound host image — is the one-iteration-at-a-time sum of iterations 1..k, bit for bit: for a
    mesh scene and a primitives-only one, with a wait after every call and with none until the end, across group
    boundaries, with the last iteration named, after calls out of order, after pt_clear, mixed with pt_trace and
    pt_trace_range, and after an option change."""
    for name, kw in (("dragon", dict(res=(96, 64), depth=8, grid=(48, 49), env_size=(64, 32), tex_size=16)), ("cornell", dict(res=(80, 56), depth=8))):
        sc = scenes.make_scene(name, **kw)
        o = orc.Oracle(pt, sc, orc.MATH_DET)
        sums = {0: np.zeros((sc.resolution[1], sc.resolution[0], 3), np.float32)}
        for it in range(1, 41):
            o.trace(it)
            sums[it] = o.image().copy()
        host = np.full((sc.resolution[1], sc.resolution[0], 3), -7.0, np.float32)
        # strict: the host image is waited for after every call
        tr = pt.Tracer(sc)
        tr.set_option(pt.PT_OPT_BATCH, 12)                  # two halves of six rows
        tr.bind_host_image(host)
        for it in range(1, 41):
            tr.trace_lookahead(it, 6)
            tr.host_image_wait()
            assert _same_bits(host, sums[it]), it
        assert _same_bits(tr.image(), sums[40])
        st = tr.stats()
        assert st.iterations >= 40 and st.iterations % 6 == 0       # whole groups were rendered, some of them ahead
        # no wait until the end; the last iteration is named, so nothing beyond it is rendered
        tr = pt.Tracer(sc)
        tr.set_option(pt.PT_OPT_BATCH, 16)
        host[:] = -7.0
        tr.bind_host_image(host)
        for it in range(1, 31):
            tr.trace_lookahead(it, 8, 30)
        tr.host_image_wait()
        assert _same_bits(host, sums[30]) and _same_bits(tr.image(), sums[30])
        assert tr.stats().iterations == 30 and tr.stats().pathBounces == o_path_bounces(pt, orc, sc, 30)
        # more look-ahead than rows: it shrinks to half the rows; with two rows or fewer it is pt_trace
        for batch in (5, 2, 1):
            tr = pt.Tracer(sc)
            tr.set_option(pt.PT_OPT_BATCH, batch)
            for it in range(1, 12):
                tr.trace_lookahead(it, 64)
            assert _same_bits(tr.image(), sums[11]), batch
        # out of order (what was rendered ahead is dropped), a clear in between, other entry points in between
        tr = pt.Tracer(sc)
        tr.set_option(pt.PT_OPT_BATCH, 8)
        for it in (1, 2, 3):
            tr.trace_lookahead(it, 4)
        tr.clear()
        for it in (1, 2, 3, 4, 5, 6):
            tr.trace_lookahead(it, 4)
        assert _same_bits(tr.image(), sums[6])
        tr.trace(7)                                         # rows 0..3 are overwritten: the look-ahead is dropped
        tr.trace_lookahead(8, 4)
        tr.trace_lookahead(9, 4)
        tr.trace_range(10, 12)
        tr.trace_lookahead(12, 4)
        tr.set_option(pt.PT_OPT_MATERIAL_SORT, 1)           # any option change drops it as well
        tr.trace_lookahead(13, 4)
        assert _same_bits(tr.image(), sums[13])
        tr.clear()
        tr.trace_lookahead(20, 4)                           # a jump
        tr.trace_lookahead(21, 4)
        tr.trace_lookahead(5, 4)                            # and back
        want = (sums[20] - sums[19]) + (sums[21] - sums[20])
        got = tr.image()
        ref = pt.Tracer(sc)
        for it in (20, 21, 5):
            ref.trace(it)
        assert _same_bits(got, ref.image()) and np.allclose(got, want + (sums[5] - sums[4]), rtol=1e-4, atol=1e-5)
        tr.bind_host_image(None)


def test_contexts_side_by_side_serve_lookahead_calls(pt, scenes, gpu_ok):
    """Several contexts alive in one process, created at different moments (the second while the first still has
    pipelined calls queued, a third after look-ahead work), serving look-ahead calls in turn with bound host images: each
    image is the one-at-a-time render of its scene, bit for bit, and nothing waits for ever (every context owns a
    high-priority stream; the runtime shares a process's few hardware queues among all streams)."""
    kw = dict(res=(160, 96), depth=8, grid=(32, 33), env_size=(64, 32), tex_size=8)
    sa, sb = scenes.make_scene("dragon", **kw), scenes.make_scene("chess", **kw)

    def reference(sc, n):
        t = pt.Tracer(sc)
        for it in range(1, n + 1):
            t.trace(it)
        return t.image().copy()

    ra, rb = reference(sa, 24), reference(sb, 24)
    a = pt.Tracer(sa)
    for it in range(1, 9):
        a.trace(it)
    b = pt.Tracer(sb)
    ha, hb = np.zeros_like(ra), np.zeros_like(rb)
    a.clear()
    for t, h in ((a, ha), (b, hb)):
        t.set_option(pt.PT_OPT_BATCH, 12)
        t.set_option(pt.PT_OPT_LAUNCH_STREAMS, 1)
        t.bind_host_image(h)
    for it in range(1, 25):
        a.trace_lookahead(it, 6, 24)
        b.trace_lookahead(it, 6, 24)
        if it % 3 == 0:
            a.host_image_wait()
            b.host_image_wait()
    a.host_image_wait()
    b.host_image_wait()
    assert _same_bits(ha, ra) and _same_bits(hb, rb)
    c = pt.Tracer(sa)
    hc = np.zeros_like(ra)
    c.set_option(pt.PT_OPT_BATCH, 12)
    c.bind_host_image(hc)
    for it in range(1, 25):
        c.trace_lookahead(it, 6, 24)
    c.host_image_wait()
    assert _same_bits(hc, ra)
    for t in (a, b, c):
        t.bind_host_image(None)


def o_path_bounces(pt, orc, sc, n):
    o = orc.Oracle(pt, sc, orc.MATH_DET)
    for it in range(1, n + 1):
        o.trace(it)
    return o.stats().pathBounces


def _desc_with_bvh(pt, sc, mesh, nodes, indices, keep):
    """A copy of the scene description in which mesh `mesh` carries the given host BVH."""
    d = pt.PtSceneDesc.from_buffer_copy(sc.desc.contents)
    meshes = (pt.PtMesh * d.meshCount)()
    for m in range(d.meshCount):
        meshes[m] = sc.desc.contents.meshes[m]
    nodes = np.ascontiguousarray(nodes)
    indices = np.ascontiguousarray(indices, np.int32)
    meshes[mesh].nodes = nodes.ctypes.data
    meshes[mesh].nodeCount = len(nodes)
    meshes[mesh].triangleIndices = indices.ctypes.data
    meshes[mesh].triangleIndexCount = len(indices)
    d.meshes = C.cast(meshes, C.POINTER(pt.PtMesh))
    keep.extend([d, meshes, nodes, indices, sc])
    return C.pointer(d)


def test_host_leaf_boxes_decide_which_triangles_count(pt, scenes, orc, gpu_ok):
    """The library walks boxes of its own, down to single triangles; which triangle hits count is decided by the leaf
    boxes of the host BVH, as in the reference (a leaf's triangles are tested only when the ray passes the leaf's box).
    With the host BVH the library builds itself a hit fails its leaf box only by a rounding accident, so this test hands
    in a BVH whose leaf boxes have their far part (seen from the camera) cut away: rays that enter a leaf box only
    there still hit its triangles geometrically, but the reference does not look at them, and the closest hit falls
    through to a triangle behind.  The cut keeps every box's near faces, so a ray that passes a cut box enters it where
    it entered the full one and the reference's distance prune sees the same entry distance (its result then does not
    depend on its visiting order, which is all this library reproduces).  The second, checking walk of a ray runs
    for every ray whose closest geometric hit does not count."""
    sc = scenes.make_scene("dragon", res=(160, 96), depth=8, grid=(40, 41), env_size=(64, 32), tex_size=16)
    d0 = sc.desc.contents
    mesh = 0
    gi = [i for i in range(d0.geomCount) if d0.geoms[i].type == pt.PT_GLTF_MESH and d0.geoms[i].meshIndex == mesh][0]
    inv = np.array(list(d0.geoms[gi].inverseTransform), np.float64).reshape(4, 4).T        # column-major in memory
    eye = (inv @ np.array(list(d0.camera.position) + [1.0]))[:3]
    n = int(d0.meshes[mesh].triangleCount)
    tris = np.frombuffer(C.string_at(d0.meshes[mesh].triangles, n * pt.TRI_DTYPE.itemsize), dtype=pt.TRI_DTYPE).copy()
    nodes, idx, _ = pt.bvh_build(tris)
    nd = nodes.copy()
    cut = 0
    for i in np.flatnonzero(nd["leftChild"] == -1):
        lo, hi = nd["bmin"][i].astype(np.float64), nd["bmax"][i].astype(np.float64)
        gap = np.where(eye < lo, lo - eye, np.where(eye > hi, eye - hi, np.inf))
        gap = np.where(hi - lo > 0, gap, np.inf)
        k = int(np.argmin(gap))                 # the axis along which the camera looks at this box most obliquely
        if not np.isfinite(gap[k]):
            continue
        if eye[k] < lo[k]:
            nd["bmax"][i][k] = np.float32(lo[k] + 0.3 * (hi[k] - lo[k]))
        else:
            nd["bmin"][i][k] = np.float32(hi[k] - 0.3 * (hi[k] - lo[k]))
        cut += 1
    assert cut > 100
    keep = []
    desc = _desc_with_bvh(pt, sc, mesh, nd, idx, keep)
    tr = pt.Tracer(desc)
    o = orc.Oracle(pt, desc, orc.MATH_DET)
    ro, rd = tr.camera_rays(1)
    g, ob = tr.intersect(ro, rd), o.intersect(ro, rd)
    _hits_equal(g, ob)
    # the cut boxes do change the picture: mesh hits are lost or replaced by triangles behind
    full = pt.Tracer(sc).intersect(ro, rd)
    on_mesh = full["triangle"] >= 0
    changed = on_mesh & ((g["triangle"] != full["triangle"]) | (g["t"] != full["t"]))
    assert changed.sum() > 0.02 * on_mesh.sum(), (int(changed.sum()), int(on_mesh.sum()))
    assert (changed & (g["triangle"] >= 0)).sum() > 50        # fell through to another triangle, not just to a miss


def test_launch_shapes_and_largest_groups_bit_exact(pt, scenes, orc, gpu_ok):
    """Groups of 12 iterations or more go through the library's launch-shape trial (one, two, three streams, one stream
    with the shadow traversal in line, then the kept shape), and a group may hold up to 512 iterations: the sum is
    the one-iteration-at-a-time result whatever the shape and the group size."""
    sc = scenes.make_scene("dragon", res=(32, 24), depth=6, grid=(24, 25), env_size=(64, 32), tex_size=16)
    o = orc.Oracle(pt, sc, orc.MATH_DET)
    sums = {}
    for it in range(1, 521):
        o.trace(it)
        if it in (112, 520):
            sums[it] = o.image().copy()
    tr = pt.Tracer(sc)
    tr.set_option(pt.PT_OPT_BATCH, 16)
    tr.trace_range(1, 113)                      # seven groups of 16: unmeasured, four shapes, two with the kept one
    assert _same_bits(tr.image(), sums[112])
    tr = pt.Tracer(sc)
    tr.set_option(pt.PT_OPT_BATCH, 512)
    tr.trace_range(1, 521)                      # one group of 512 and one of 8
    assert _same_bits(tr.image(), sums[520])
    assert tr.stats().pathBounces == o.stats().pathBounces
    with pytest.raises(pt.PtError):
        tr.set_option(pt.PT_OPT_BATCH, 513)
    with pytest.raises(pt.PtError, match="not supported"):
        tr.set_option(pt.PT_OPT_USE_GRAPH, 1)
    tr.set_option(pt.PT_OPT_USE_GRAPH, 0)


def test_depth_one_and_tiny_images(pt, scenes, orc, gpu_ok):
    for res, depth in (((1, 1), 8), ((3, 2), 1), ((257, 1), 2), ((16, 16), 1)):
        sc = scenes.make_scene("cornell", res=res, depth=depth)
        tr, o = _assert_full_parity(pt, orc, sc, [1, 2], lit=None)
        # look-ahead calls with a bound host image on images of a few bytes (the staged copy travels in pieces of whole pages)
        host = np.full((res[1], res[0], 3), -1.0, np.float32)
        tr.clear()
        tr.set_option(pt.PT_OPT_BATCH, 4)
        tr.bind_host_image(host)
        for it in (1, 2):
            tr.trace_lookahead(it, 2)
            tr.host_image_wait()
        assert _same_bits(host, o.image())
        tr.bind_host_image(None)


def test_radiance_rmse_vs_reference_arithmetic(pt, scenes, orc, gpu_ok):
    """GPU (fixed-sequence math) vs the oracle in libm mode (glibc sinf/cosf/...): paths whose branch
    decisions flip diverge, so the comparison is statistical."""
    sc = scenes.make_scene("cornell", res=(96, 96), depth=8)
    spp = 16
    tr, o, _ = _render_pair(pt, orc, sc, range(1, spp + 1), orc.MATH_LIBM)
    gi, oi = tr.image() / spp, o.image() / spp
    mean = oi.reshape(-1, 3).mean(0)
    rmse = np.sqrt(((gi - oi) ** 2).reshape(-1, 3).mean(0))
    assert np.all(rmse / mean <= RMSE_TOL), (rmse / mean)
    assert np.allclose(gi.reshape(-1, 3).mean(0), mean, rtol=5e-3)


def test_display_clear_upload_bind(pt, scenes, orc, gpu_ok):
    import torch
    sc = scenes.make_scene("cornell", res=(64, 48), depth=8)
    tr = pt.Tracer(sc)
    o = orc.Oracle(pt, sc, orc.MATH_DET)
    for it in (1, 2):
        tr.trace(it)
        o.trace(it)
    rgba = torch.zeros(48 * 64 * 4, dtype=torch.uint8, device="cuda")
    tr.display(2, rgba.data_ptr())
    tr.sync()
    assert np.array_equal(rgba.cpu().numpy().reshape(48, 64, 4), o.display(2))
    # the running sum can live in caller memory (a torch tensor) ...
    img2 = torch.zeros(48 * 64 * 3, dtype=torch.float32, device="cuda")
    snapshot = tr.image()
    tr.bind_image(img2.data_ptr())
    tr.trace(1)
    tr.trace(2)
    tr.sync()
    assert _same_bits(img2.cpu().numpy().reshape(48, 64, 3), snapshot)
    tr.bind_image(0)
    # ... be cleared, and be reloaded (resume = reload the sum)
    tr.clear()
    assert float(tr.image().sum()) == 0.0
    tr.upload_image(snapshot)
    tr.trace(3)
    o.trace(3)
    assert _same_bits(tr.image(), o.image())


def test_sample_split_equals_single_render(pt, scenes, orc, gpu_ok):
    """Iterations interleaved over R contexts and summed == one context rendering them all (up to fp32 order)."""
    sc = scenes.make_scene("cornell", res=(64, 64), depth=8)
    single = pt.Tracer(sc)
    single.trace_range(1, 9)
    parts = []
    for r in range(2):
        t = pt.Tracer(sc)
        t.trace_range(1 + r, 9, 2)
        parts.append(t.image())
    a, b = single.image(), parts[0] + parts[1]
    assert np.allclose(a, b, rtol=1e-5, atol=1e-6)
    assert single.stats().pathBounces > 0


def test_reduce_sum_of_split_contexts(pt, scenes, gpu_ok, tmp_path):
    """Sample split inside one process (pt_reduce_sum, pt_headless --gpus): R contexts — here on the one GPU of the
    box — render iterations 1+r, 1+r+R, ... and their running sums are folded into the first.  The union is the
    one-context render: equal up to the order of the fp32 additions, and bit for bit equal to adding the parts on the
    host in the same order.  The other contexts' sums are left alone; bad arguments are refused."""
    import subprocess
    sc = scenes.make_scene("dragon", res=(96, 64), depth=8, grid=(48, 49), env_size=(64, 32), tex_size=16)
    n = 12
    single = pt.Tracer(sc)
    single.set_option(pt.PT_OPT_BATCH, 4)
    single.trace_range(1, n + 1)
    whole = single.image()
    for R in (2, 3):
        parts = [pt.Tracer(sc) for _ in range(R)]
        for r, t in enumerate(parts):
            t.set_option(pt.PT_OPT_BATCH, 3)
            t.trace_range(1 + r, n + 1, R)
        before = [t.image().copy() for t in parts]
        pt.reduce_sum(parts, root=0)
        got = parts[0].image()
        want = before[0].copy()
        for b in before[1:]:
            want += b
        assert _same_bits(got, want)
        assert np.allclose(got, whole, rtol=1e-5, atol=1e-6)
        for t, b in zip(parts[1:], before[1:]):
            assert _same_bits(t.image(), b)
        assert sum(t.stats().pathBounces for t in parts) == single.stats().pathBounces
    with pytest.raises(pt.PtError):
        pt.reduce_sum([parts[0], parts[0]])
    other = pt.Tracer(scenes.make_scene("cornell", res=(32, 32)))
    with pytest.raises(pt.PtError, match="pixels"):
        pt.reduce_sum([parts[0], other])
    assert pt.lib.pt_device_count() >= 1
    # the same split through the headless program: two host threads, two contexts, one sum
    exe = os.path.join(os.path.dirname(pt.LIB_PATH), "pt_headless")
    path = scenes.write_scene_json("cornell", str(tmp_path))
    outs = []
    for extra in (["--batch"], ["--gpus", "2"], ["--gpus", "4"]):
        out = str(tmp_path / ("split%d.pfm" % len(outs)))
        subprocess.check_call([exe, path, "-n", "8", "-r", "64", "48", "-d", "6", "-o", out] + extra, timeout=180)
        with open(out, "rb") as f:
            f.readline(); f.readline(); f.readline()
            outs.append(np.frombuffer(f.read(), dtype="<f4").reshape(48, 64, 3))
    assert outs[0].sum() > 0
    assert np.allclose(outs[0], outs[1], rtol=1e-5, atol=1e-6) and np.allclose(outs[0], outs[2], rtol=1e-5, atol=1e-6)


def test_counters(pt, scenes, orc, gpu_ok):
    """Traversal counters: the same closest hits as the oracle (mesh-hit counts are traversal-independent);
    the ordered 4-wide traversal must test fewer triangles than the reference's unordered binary walk."""
    sc = scenes.make_scene("dragon", res=(64, 64), depth=4, grid=(48, 49), env_size=(64, 32), tex_size=16)
    tr = pt.Tracer(sc)
    tr.set_option(pt.PT_OPT_COUNTERS, 1)
    o = orc.Oracle(pt, sc, orc.MATH_DET)
    tr.trace(1)
    o.trace(1)
    gs, os_ = tr.stats(), o.stats()
    assert gs.meshHits == os_.meshHits and gs.shadeMeshHits == os_.shadeMeshHits
    assert 0 < gs.trisTested <= os_.trisTested
    assert 0 < gs.nodesVisited < os_.nodesVisited


def test_errors_are_reported(pt, scenes, gpu_ok):
    sc = scenes.make_scene("cornell", res=(16, 16), depth=8)
    with pytest.raises(pt.PtError):
        pt.Tracer(sc, device=99)
    d = sc.desc.contents
    d.traceDepth = 0
    with pytest.raises(pt.PtError):
        pt.Tracer(sc)
    assert pt.lib.pt_destroy(None) == 0          # pathtraceFree before the first init is a no-op


@pytest.mark.gpu
def test_pathtrace_surface_headless_driver(pt, scenes, orc, gpu_ok, tmp_path):
    """pathtraceInit / pathtrace(nullptr, 0, iter) / pathtraceFree driven the way runCuda drives them
    (src/main.cpp:812-837), through the headless executable: the averaged image equals the oracle's, bit for bit,
    one iteration per call as well as batched."""
    import subprocess
    exe = os.path.join(os.path.dirname(pt.LIB_PATH), "pt_headless")
    path = scenes.write_scene_json("cornell", str(tmp_path))
    W, H, n = 48, 40, 3
    sc = pt.Scene(path)
    sc.set_resolution(W, H)
    sc.set_depth(6)
    o = orc.Oracle(pt, sc, orc.MATH_DET)
    for it in range(1, n + 1):
        o.trace(it)
    total = o.image().reshape(H, W, 3)
    expect = (total / np.float32(n)).astype(np.float32)
    o.close()
    for extra in ([], ["--batch"]):
        out = str(tmp_path / ("img%d.pfm" % len(extra)))
        subprocess.check_call([exe, path, "-n", str(n), "-r", str(W), str(H), "-d", "6", "-o", out] + extra, timeout=120)
        with open(out, "rb") as f:
            assert f.readline().strip() == b"PF"
            w, h = map(int, f.readline().split())
            assert (w, h) == (W, H) and float(f.readline()) < 0
            got = np.frombuffer(f.read(), dtype="<f4").reshape(H, W, 3)[::-1]
        assert np.array_equal(got.view(np.uint32), expect.view(np.uint32))
        assert got.sum() > 0
    # pathtraceFree() + pathtraceInit() in the middle of a render, as the reference's preview does when the camera moves:
    # the parked context is taken back (a handful of resource calls instead of the hundreds of a pt_create), the render
    # that follows starts from nothing; with PT_KEEP_CONTEXT=0 everything is freed and built again, same image
    for keep in ("1", "0"):
        out = str(tmp_path / ("restart%s.pfm" % keep))
        r = subprocess.run([exe, path, "-n", str(n), "-r", str(W), str(H), "-d", "6", "-o", out, "--restart-after", "2"], timeout=120,
                           stderr=subprocess.PIPE, env=dict(os.environ, PT_KEEP_CONTEXT=keep))
        assert r.returncode == 0, r.stderr.decode()[-400:]
        calls = int([l for l in r.stderr.decode().splitlines() if l.startswith("restart after")][-1].split(": ")[1].split()[0])
        assert (calls < 16) if keep == "1" else (calls > 40), calls
        with open(out, "rb") as f:
            f.readline(); f.readline(); f.readline()
            got = np.frombuffer(f.read(), dtype="<f4").reshape(H, W, 3)[::-1]
        assert np.array_equal(got.view(np.uint32), expect.view(np.uint32))
    # the reference's own output forms: 8-bit PNG (averaged, x mirrored, clamped, truncated) and Radiance .hdr
    png, hdr = str(tmp_path / "img.png"), str(tmp_path / "img.hdr")
    subprocess.check_call([exe, path, "-n", str(n), "-r", str(W), str(H), "-d", "6", "-o", png], timeout=120)
    subprocess.check_call([exe, path, "-n", str(n), "-r", str(W), str(H), "-d", "6", "-o", hdr, "--batch"], timeout=120)
    dn = str(tmp_path / "dn.pfm")
    subprocess.check_call([exe, path, "-n", str(n), "-r", str(W), str(H), "-d", "6", "-o", dn, "--denoise"], timeout=120)
    with open(dn, "rb") as f:
        f.readline(); f.readline(); f.readline()
        smooth = np.frombuffer(f.read(), dtype="<f4").reshape(H, W, 3)[::-1]
    assert np.isfinite(smooth).all() and not np.array_equal(smooth, expect) and abs(float(smooth.mean()) / float(expect.mean()) - 1) < 0.2
    want8 = (np.clip(expect[:, ::-1], 0, 1) * np.float32(255)).astype(np.uint8)
    got8 = pt.decode_png(open(png, "rb").read())
    assert np.array_equal(got8[:, :, :3], want8) and (got8[:, :, 3] == 255).all()
    env = json.load(open(path))
    env["EnvironmentMap"] = dict(FILE="img.hdr", INTENSITY=1.0)
    (tmp_path / "env.json").write_text(json.dumps(env))
    sc2 = pt.Scene(tmp_path / "env.json")
    back = sc2.desc.contents.env
    assert (back.width, back.height, back.enabled) == (W, H, 1)
    rgbe = np.frombuffer(C.string_at(back.data, W * H * 12), np.float32).reshape(H, W, 3)
    m = expect[:, ::-1].max(axis=2, keepdims=True)
    assert np.all(np.abs(rgbe - expect[:, ::-1]) <= m / 128 + 1e-30)       # 8-bit mantissa shared by the three channels


def test_full_size_properties_dragon_1080p(pt, scenes, gpu_ok):
    """At the benchmark's own size (1920x1080, depth 8, 135 200 triangles) the oracle is too slow to be the checker,
    so size-independent properties are: launch groups of any size and stream split give the same bits as one
    iteration per call; an interleaved sample split over two tracers sums to the single render up to fp32
    summation order; the path-bounce total equals the sum of the live-path curve; every pixel is finite and >= 0."""
    sc = scenes.make_scene("dragon", res=(1920, 1080), depth=8)
    n = 6
    a = pt.Tracer(sc)
    for it in range(1, n + 1):
        a.trace(it)
    img_a = a.image()
    sa = a.stats()
    assert np.isfinite(img_a).all() and (img_a >= 0).all() and img_a.sum() > 0
    # the last iteration's live curve sums to that iteration's share of the path-bounce total
    b = pt.Tracer(sc)
    b.set_option(pt.PT_OPT_BATCH, n)
    b.trace_range(1, n + 1, 1)
    sb = b.stats()
    assert sb.pathBounces == sa.pathBounces
    assert list(sb.lastLive[:9]) == list(sa.lastLive[:9])
    assert _same_bits(b.image(), img_a)
    c = pt.Tracer(sc)
    c.set_option(pt.PT_OPT_BATCH, 4)
    c.trace_range(1, n + 1, 1)
    assert _same_bits(c.image(), img_a)
    one = pt.Tracer(sc)
    one.trace(n)
    so = one.stats()
    assert so.pathBounces == sum(so.lastLive[:8])
    # two "ranks": iterations 1,3,5 and 2,4,6
    r0, r1 = pt.Tracer(sc), pt.Tracer(sc)
    r0.trace_range(1, n + 1, 2)
    r1.trace_range(2, n + 1, 2)
    assert r0.stats().pathBounces + r1.stats().pathBounces == sa.pathBounces
    s = r0.image() + r1.image()
    assert np.allclose(s, img_a, rtol=1e-5, atol=1e-5)


def test_gpu_reproduces_committed_vectors(pt, scenes, gpu_ok):
    """The committed oracle vectors (tests/golden/oracle_vectors.npz), without the oracle in the loop: closest hits on the
    stored ray set and whole iterations, bit for bit, on the analytic scene and on the stored two-mesh scene."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden
    g = np.load(os.path.join(here, "golden", "oracle_vectors.npz"))
    ro, rd = g["ray_o"], g["ray_d"]
    sc = scenes.make_scene("cornell", res=(48, 48), depth=8)
    tr = pt.Tracer(sc)
    _hits_equal(tr.intersect(ro, rd), g["cornell_hits"])
    for it in (1, 2, 3):
        tr.trace(it)
    assert _same_bits(tr.image(), g["cornell_det_image"])
    st = tr.stats()
    assert int(st.pathBounces) == int(g["cornell_det_path_bounces"])
    assert list(st.lastLive[:9]) == list(g["cornell_det_live"][2])
    sc = make_golden.build_mesh_scene(pt, scenes, g["blob"], g["cube"], (40, 40))
    tr = pt.Tracer(sc)
    _hits_equal(tr.intersect(ro, rd), g["mesh_hits"])
    tr.trace_range(1, 3, 1)
    assert _same_bits(tr.image(), g["mesh_det_image"])
    assert int(tr.stats().pathBounces) == int(g["mesh_det_path_bounces"])


def test_empty_scene_and_odd_rays(pt, scenes, orc, gpu_ok, tmp_path):
    """No objects at all (every ray misses: black without an env map, env radiance with one), and rays with zero
    direction components, origins exactly on box planes and zero-length directions on the mesh scene."""
    import json
    d = scenes.scene_dict("cornell")
    d["Objects"] = []
    path = str(tmp_path / "empty.json")
    json.dump(d, open(path, "w"))
    for env in (False, True):
        sc = pt.Scene(path)
        sc.set_resolution(40, 24)
        if env:
            sc.set_envmap(scenes.sky_hdr(64, 32), 0.7)
        _assert_full_parity(pt, orc, sc, [1, 2], lit=env)
    sc = _three_mesh_scene(pt, scenes, (16, 16))
    tr, o = pt.Tracer(sc), orc.Oracle(pt, sc)
    ro = np.array([[0, 4, 9], [0, 4, 9], [-5, 4, 0], [0.2, 12, 0.3], [0, 4, 0], [0, 4, 9], [1.2, 4.5, 9]], np.float32)
    rd = np.array([[0, 0, -1], [0, 1, 0], [1, 0, 0], [0, -1, 0], [0, 0, 1], [0, 0, 0], [0, 0, -1]], np.float32)
    _hits_equal(tr.intersect(ro, rd), o.intersect(ro, rd))


def test_scene_file_with_gltf_object_bit_exact(pt, scenes, orc, gpu_ok, tmp_path):
    """A scene file whose mesh comes in through the glTF import (GLB with a node hierarchy, an indexed and a
    non-indexed primitive, a PNG base-colour texture, an emissive material): closest hits and whole iterations
    equal the oracle's on the same loaded scene, and the imported parts are really in play."""
    import glb_util as G
    ident = ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    tris = scenes.blob_mesh(28, 29, 21, ident, [0, 1], radius=1.0)
    b = G.GlbBuilder()
    tex = b.image_png(G.write_png(scenes.checker_texture(32, 4, comps=3), 2))
    b.material(pbrMetallicRoughness=dict(baseColorFactor=[1.0, 0.9, 0.8, 1.0], metallicFactor=0.1, roughnessFactor=0.45, baseColorTexture=dict(index=tex)))
    b.material(pbrMetallicRoughness=dict(baseColorFactor=[0.9, 0.95, 1.0, 0.15], metallicFactor=0.0, roughnessFactor=0.05))
    b.material(emissiveFactor=[2.0, 1.0, 0.5])
    prims = []
    for mid in (0, 1):
        t = tris[tris["materialId"] == mid]
        pos = np.stack([t["v0"], t["v1"], t["v2"]], 1).reshape(-1, 3)
        nrm = np.stack([t["n0"], t["n1"], t["n2"]], 1).reshape(-1, 3)
        uv = np.stack([t["uv0"], t["uv1"], t["uv2"]], 1).reshape(-1, 2)
        if mid == 0:
            rec = np.concatenate([pos, nrm, uv], 1)
            uniq, inv = np.unique(rec, axis=0, return_inverse=True)
            a_pos, a_nrm, a_uv = b.interleaved(uniq[:, 0:3], uniq[:, 3:6], uniq[:, 6:8])
            prims.append(dict(attributes=dict(POSITION=a_pos, NORMAL=a_nrm, TEXCOORD_0=a_uv), indices=b.accessor(inv.astype(np.uint16), "SCALAR", 5123), material=0))
        else:
            prims.append(dict(attributes=dict(POSITION=b.accessor(pos, "VEC3", 5126), NORMAL=b.accessor(nrm, "VEC3", 5126)), material=1))
    body = b.node(mesh=b.mesh(prims), rotation=[0.0, 0.38268343, 0.0, 0.92387953], scale=[1.2, 1.0, 1.2])
    cube = scenes.cube_mesh(ident, 0)
    cpos = np.stack([cube["v0"], cube["v1"], cube["v2"]], 1).reshape(-1, 3)
    lamp = b.node(mesh=b.mesh([dict(attributes=dict(POSITION=b.accessor(cpos, "VEC3", 5126)), material=2)]), translation=[0.0, 2.0, 0.0], scale=[0.6, 0.1, 0.6])
    b.node(root=True, children=[body, lamp], translation=[0.0, 1.2, 0.0])
    (tmp_path / "assets").mkdir()
    b.write_glb(tmp_path / "assets" / "blob.glb")
    d = scenes.scene_dict("cornell_orig")
    d["Camera"] = scenes._camera("gltf", 8, 1, res=(88, 80))
    d["Objects"].append(dict(TYPE="gltf", FILE="assets/blob.glb", TRANS=[0.15, 1.0, 0.25], ROTAT=[0.0, 30.0, 10.0], SCALE=[1.4, 1.5, 1.4]))
    p = tmp_path / "gltf_scene.json"
    p.write_text(json.dumps(d))
    sc = pt.Scene(p)
    # the first primitive is interleaved: its buffer view's byteStride is ignored, as the reference ignores it (the geometry
    # that comes out is the reference's, scrambled but the same on both sides), and the import says so — nothing else
    notes = [l for l in sc.log.splitlines() if l.strip()]
    assert len(notes) == 3 and all("byteStride 32 is ignored" in l for l in notes), sc.log
    assert sc.geoms()[-1].triangleCount == len(tris) + 12 and len(sc.textures()) == 1
    nm = len(sc.materials())
    assert sc.materials()[nm - 1].type == pt.PT_EMITTING and sc.materials()[nm - 2].transparency == np.float32(1.0 - 0.15)
    tr = pt.Tracer(sc)
    o = orc.Oracle(pt, sc)
    # the object transform acts on the baked triangles once more when rays are taken to object space: aim a second
    # ray set at where the small emissive part ends up
    M = pt.build_transforms((0.15, 1.0, 0.25), (0.0, 30.0, 10.0), (1.4, 1.5, 1.4))[0].T.astype(np.float64)
    lamp_at = (M @ np.append(sc.mesh_triangles(0)["v0"][-12:].astype(np.float64).mean(0), 1.0))[:3]
    sets = [_ray_set(9, 20000, center=(0.3, 5.0, 0.5), radius=4.4), _ray_set(10, 4000, center=tuple(lamp_at), radius=1.3)]
    ro, rd = np.concatenate([s_[0] for s_ in sets]), np.concatenate([s_[1] for s_ in sets])
    g, ob = tr.intersect(ro, rd), o.intersect(ro, rd)
    _hits_equal(g, ob)
    on_mesh = g["triangle"] >= 0
    assert on_mesh.sum() > 2000 and set(np.unique(g["materialId"][on_mesh]).tolist()) == {nm - 3, nm - 2, nm - 1}
    _assert_full_parity(pt, orc, sc, [1, 2, 3])


def test_sparse_launch_form_is_bit_identical(pt, scenes, orc, gpu_ok, monkeypatch):
    """The heavy per-slot kernels have a bounded-grid form for nearly empty bounces, picked from the previous launch
    group's live counts; forced on for every bounce (PT_SPARSE=1), forced off, or picked by the hint over several
    groups, the image is the same and equals the oracle's."""
    sc = scenes.make_scene("dragon", res=(160, 90), depth=8, grid=(48, 49), env_size=(64, 32), tex_size=16)
    images = []
    for mode in ("1", "0", None):
        if mode is None:
            monkeypatch.delenv("PT_SPARSE", raising=False)
        else:
            monkeypatch.setenv("PT_SPARSE", mode)
        tr = pt.Tracer(sc)
        tr.set_option(pt.PT_OPT_BATCH, 4)
        for first in (1, 5, 9):                 # three launch groups: the second and third may use the hint of the one before
            tr.trace_range(first, first + 4, 1)
            tr.sync()
        images.append(tr.image())
        del tr
    assert _same_bits(images[0], images[1]) and _same_bits(images[0], images[2])
    o = orc.Oracle(pt, sc, orc.MATH_DET)
    for it in range(1, 13):
        o.trace(it)
    assert _same_bits(images[0], o.image())


def _atrous_reference(img, normals, albedo, levels, sc, sn, sa):
    """CPU statement of the library's a-trous filter (float64 accumulators; the device works in float32)."""
    kern = np.array([1 / 16, 1 / 4, 3 / 8, 1 / 4, 1 / 16])
    cur = img.astype(np.float64)
    H, W = cur.shape[:2]
    for l in range(levels):
        step = 1 << l
        acc = np.zeros_like(cur)
        wsum = np.zeros((H, W))
        for dy in range(-2, 3):
            for dx in range(-2, 3):
                oy, ox = dy * step, dx * step
                ys = slice(max(0, -oy), min(H, H - oy))
                xs = slice(max(0, -ox), min(W, W - ox))
                yt = slice(ys.start + oy, ys.stop + oy)
                xt = slice(xs.start + ox, xs.stop + ox)
                e = ((cur[yt, xt] - cur[ys, xs]) ** 2).sum(-1) / (sc * sc)
                if sn > 0:
                    e = e + ((normals[yt, xt].astype(np.float64) - normals[ys, xs]) ** 2).sum(-1) / (sn * sn)
                if sa > 0:
                    e = e + ((albedo[yt, xt].astype(np.float64) - albedo[ys, xs]) ** 2).sum(-1) / (sa * sa)
                w = kern[dy + 2] * kern[dx + 2] * np.exp(-e)
                acc[ys, xs] += cur[yt, xt] * w[..., None]
                wsum[ys, xs] += w
        cur = acc / wsum[..., None]
        sc *= 0.5
    return cur


def test_gbuffer_capture_and_denoiser(pt, scenes, orc, gpu_ok):
    """captureNormalsAndAlbedo (src/pathtrace.cu:94-135): first-hit normal (normalised) and material colour per pixel,
    bit-exact against the oracle's camera rays + intersect; the a-trous stand-in for the closed denoiser against a
    CPU statement of the same filter, and it does reduce the error of a noisy render."""
    sc = scenes.make_scene("dragon", res=(96, 64), depth=6, grid=(40, 41), env_size=(64, 32), tex_size=16)
    tr = pt.Tracer(sc)
    o = orc.Oracle(pt, sc)
    normals, albedo = tr.capture_gbuffer(3)
    ro, rd = o.camera_rays(3)
    h = o.intersect(ro, rd)
    hit = h["t"] > 0
    n = h["normal"].astype(np.float32)
    d = (n[:, 0] * n[:, 0] + n[:, 1] * n[:, 1]) + n[:, 2] * n[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        nn = n * (np.float32(1.0) / np.sqrt(d, dtype=np.float32))[:, None]
    mats = sc.materials()
    col = np.array([list(m.color) for m in mats], np.float32)
    want_n = np.where(hit[:, None], nn, np.float32(0)).reshape(64, 96, 3)
    want_a = np.where(hit[:, None], col[np.clip(h["materialId"], 0, len(mats) - 1)], np.float32(0)).reshape(64, 96, 3)
    assert _same_bits(normals, want_n) and _same_bits(albedo, want_a)
    assert 0.2 < hit.mean() < 1.0 and len(np.unique(h["materialId"][hit])) >= 3
    # denoise 4 spp against a 256 spp render of the same scene
    tr.set_option(pt.PT_OPT_BATCH, 64)
    tr.trace_range(1, 5)
    noisy = tr.image() / np.float32(4)
    for guides in ((0.0, 0.0), (0.35, 0.25)):
        got = tr.denoise(4, levels=4, sigma_color=0.8, sigma_normal=guides[0], sigma_albedo=guides[1])
        ref = _atrous_reference(noisy, normals, albedo, 4, 0.8, guides[0], guides[1])
        assert np.allclose(got, ref, rtol=2e-4, atol=2e-5), float(np.abs(got - ref).max())
    tr.clear()
    tr.trace_range(1, 257)
    clean = tr.image() / np.float32(256)
    tr.upload_image(noisy * np.float32(4))
    strong = tr.denoise(4, levels=5, sigma_color=4.0, sigma_normal=0.35, sigma_albedo=0.25)
    err = lambda a: float(np.mean(np.abs(np.clip(a, 0, 2) - np.clip(clean, 0, 2))))
    print("mean abs error: noisy %.4f, filtered %.4f" % (err(noisy), err(strong)))
    assert err(strong) < 0.85 * err(noisy), (err(strong), err(noisy))
    with pytest.raises(pt.PtError):
        pt.Tracer(sc).denoise(4)            # guides wanted, nothing captured


def test_cube_sphere_edge_rays_bit_exact(pt, scenes, orc, gpu_ok, tmp_path):
    """Boxes and spheres at the edges of their tests' arithmetic: rays aimed exactly at, and a hair beside, every corner,
    edge midpoint and face centre of every geom, rays lying in face planes, rays from far away, unnormalised, zero,
    infinite and NaN rays, flat (scale 0: non-finite inverse matrices) and tiny geoms.  Same hits as the oracle, and
    whole iterations of that scene stay bit-identical.  (A shortcut that skipped the exact test of geoms whose padded
    world box a ray misses failed this test and bought 3-8 % of the primitive stage; it was dropped.)"""
    import json
    d = scenes.scene_dict("ssaa")
    d["Objects"] += [
        dict(TYPE="cube", MATERIAL="diffuse_red", TRANS=[2.0, 3.0, 1.0], ROTAT=[10.0, 20.0, 30.0], SCALE=[2.0, 0.0, 2.0]),
        dict(TYPE="sphere", MATERIAL="diffuse_green", TRANS=[-2.0, 2.0, 1.0], ROTAT=[0.0, 0.0, 0.0], SCALE=[1e-7, 1e-7, 1e-7]),
        dict(TYPE="sphere", MATERIAL="diffuse_green", TRANS=[-3.0, 7.0, -2.0], ROTAT=[30.0, 0.0, 60.0], SCALE=[3.0, 0.5, 1.0]),
        dict(TYPE="cube", MATERIAL="diffuse_red", TRANS=[3.0, 7.5, -2.0], ROTAT=[45.0, 45.0, 0.0], SCALE=[1.0, 2.0, 0.02]),
    ]
    path = tmp_path / "edge.json"
    path.write_text(json.dumps(d))
    sc = pt.Scene(path)
    sc.set_resolution(32, 32)
    tr = pt.Tracer(sc)
    o = orc.Oracle(pt, sc)
    rng = np.random.RandomState(77)
    targets = []
    for g in sc.geoms():
        T = np.array(list(g.transform), np.float64).reshape(4, 4).T      # column-major [col][row] -> row-major
        for p in np.array(np.meshgrid([-0.5, 0.0, 0.5], [-0.5, 0.0, 0.5], [-0.5, 0.0, 0.5])).reshape(3, -1).T:
            if np.count_nonzero(p) == 0:
                continue
            targets.append((T @ np.array([p[0], p[1], p[2], 1.0]))[:3])
    targets = np.array(targets)
    origins = np.array([[0, 5, 6.6], [0, 5, 0.3], [4.9, 9.9, 4.9], [-4.9, 0.1, -4.9], [0.3, 9.0, -3.0], [1e4, 2e4, 3e4], [0, 5, 2e6]], np.float64)
    ro, rd = [], []
    for org in origins:
        dd = targets - org
        dd /= np.maximum(np.linalg.norm(dd, axis=1, keepdims=True), 1e-30)
        for eps in (0.0, 1e-6, -1e-6, 3e-4):
            ro.append(np.tile(org, (len(dd), 1)))
            rd.append(dd + eps * rng.normal(size=dd.shape))
    ro, rd = np.concatenate(ro), np.concatenate(rd)
    # rays inside the planes of the room's walls and of the lights, axis-parallel
    plane_o = np.array([[-5, 5, 0], [5, 5, 0], [0, 0, 0], [0, 10, 0], [0, 5, -5], [0, 9.85, 0], [0, 0.15, 0], [-4.995, 5, 3], [0, 0.005, 3]], np.float64)
    for ax in np.eye(3):
        for sgn in (1.0, -1.0):
            ro = np.concatenate([ro, plane_o - 3.0 * sgn * ax]); rd = np.concatenate([rd, np.tile(sgn * ax, (len(plane_o), 1))])
    # unnormalised and odd rays
    so, sd = _ray_set(5, 3000)
    scale = np.float32(10.0) ** rng.randint(-25, 25, size=(3000, 1)).astype(np.float32)
    ro = np.concatenate([ro, so]); rd = np.concatenate([rd, sd * scale])
    odd_o = np.array([[0, 5, 6], [0, 5, 6], [0, 5, 6], [np.nan, 5, 6], [0, np.inf, 6], [0, 5, 6], [0, 5, 6], [0, 5, 6]], np.float64)
    odd_d = np.array([[0, 0, 0], [np.nan, 0, -1], [0, np.inf, -1], [0, 0, -1], [0, 0, -1], [1e-30, 0, -1e-30], [1e-25, 1e-25, -1e-25], [0, -0.0, -1]], np.float64)
    ro = np.concatenate([ro, odd_o]).astype(np.float32); rd = np.concatenate([rd, odd_d]).astype(np.float32)
    g, ob = tr.intersect(ro, rd), o.intersect(ro, rd)
    bad = np.nonzero((g["geom"] != ob["geom"]) | (g["t"].view(np.uint32) != ob["t"].view(np.uint32)))[0]
    for i in bad[:20]:
        print("ray", i, "of", len(ro), ro[i], rd[i], "gpu", g["geom"][i], g["t"][i], "oracle", ob["geom"][i], ob["t"][i])
    _hits_equal(g, ob)
    assert (ob["t"] > 0).sum() > 2000
    # whole iterations of the same scene (shadow tests against the flat and the tiny geom included)
    o2 = orc.Oracle(pt, sc, orc.MATH_DET)
    for it in (1, 2):
        tr.trace(it); o2.trace(it)
    assert _same_bits(tr.image(), o2.image())
    assert tr.stats().pathBounces == o2.stats().pathBounces


def test_many_geoms_and_a_mesh_bit_exact(pt, scenes, orc, gpu_ok, tmp_path):
    """More than 32 geoms: the traversal hands the geom of a mesh hit to the primitive stage through the array entry
    instead of the spare bits of the hit record.  Closest hits and whole iterations against the oracle."""
    d = scenes.scene_dict("cornell_orig")
    rng = np.random.RandomState(12)
    for k in range(36):
        c = rng.uniform([-4, 1, -4], [4, 9, 4])
        d["Objects"].append(dict(TYPE="sphere" if k % 2 else "cube", MATERIAL="diffuse_white", TRANS=[float(v) for v in c],
                                 ROTAT=[float(v) for v in rng.uniform(0, 90, 3)], SCALE=[0.5, 0.4, 0.6]))
    path = tmp_path / "many.json"
    path.write_text(json.dumps(d))
    sc = pt.Scene(path)
    sc.set_resolution(64, 48)
    ident = ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    m = sc.add_material(type=pt.PT_PBR, color=[0.8, 0.5, 0.3], roughness=0.4, metallic=0.0, indexOfRefraction=1.5)
    blob = scenes.blob_mesh(24, 25, 5, ident, [0], radius=2.0, center=(0.0, 4.0, 0.0))
    blob["materialId"] = m
    sc.add_mesh(blob)
    assert sc.desc.contents.geomCount > 32
    tr = pt.Tracer(sc)
    o = orc.Oracle(pt, sc, orc.MATH_DET)
    ro, rd = _ray_set(21, 20000)
    g, ob = tr.intersect(ro, rd), o.intersect(ro, rd)
    _hits_equal(g, ob)
    assert (ob["triangle"] >= 0).sum() > 500
    for it in (1, 2, 3):
        tr.trace(it); o.trace(it)
    assert _same_bits(tr.image(), o.image())
    assert tr.stats().pathBounces == o.stats().pathBounces
