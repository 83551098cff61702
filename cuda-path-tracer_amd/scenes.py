"""Scene configurations for tests and benchmarks.

The benchmark configurations name the reference's scene files (scenes/cornell.json,
sss.json, dragon.json, chess.json, 911.json, halo.json, ...).  Primitive-only scenes are
emitted here as JSON in the reference's schema (src/scene.cpp:570-919) and go through the
C++ loader.  The glTF / HDR assets behind dragon, chess, 911 and halo are not distributed
with the reference (.gitignore:7-8), so those configurations use seeded procedural
stand-ins of the same class: a closed displaced mesh with the published triangle count,
the material mix the write-up quotes, procedural textures and a procedural HDR sky.
"""
import json
import os
import tempfile

import numpy as np

from . import PT_EMITTING, PT_PBR, TRI_DTYPE, Scene, build_transforms

# ---------------------------------------------------------------- material library
_SSS = dict(TYPE="PBR", METALLIC=0.0, TRANSPARENCY=0.0, SUBSURFACE_ENABLED=1)
MATERIALS = {
    "light": dict(TYPE="Emitting", RGB=[1.0, 1.0, 1.0], EMITTANCE=5.0),
    "diffuse_white": dict(TYPE="Diffuse", RGB=[0.98, 0.98, 0.98]),
    "diffuse_blue": dict(TYPE="Diffuse", RGB=[0.0, 0.0, 0.98]),
    "diffuse_red": dict(TYPE="Diffuse", RGB=[0.85, 0.35, 0.35]),
    "diffuse_green": dict(TYPE="Diffuse", RGB=[0.35, 0.85, 0.35]),
    "specular_white": dict(TYPE="Specular", RGB=[0.98, 0.98, 0.98], ROUGHNESS=0.0),
    "refractive_white": dict(TYPE="Refractive", RGB=[0.98, 0.98, 0.98], IOR=1.5),
    "pbr_glass": dict(TYPE="PBR", RGB=[0.95, 0.95, 0.95], IOR=1.5, ROUGHNESS=0.0, METALLIC=0.0, TRANSPARENCY=0.85),
    "human_skin": dict(_SSS, RGB=[0.95, 0.71, 0.67], ROUGHNESS=0.35, SUBSURFACE_COLOR=[0.85, 0.57, 0.50],
                       SUBSURFACE_RADIUS=[0.032, 0.017, 0.005], SUBSURFACE_SCALE=1.2, SUBSURFACE_ANISOTROPY=0.0),
    "candle_wax": dict(_SSS, RGB=[0.95, 0.92, 0.85], ROUGHNESS=0.45, SUBSURFACE_COLOR=[0.9, 0.85, 0.7],
                       SUBSURFACE_RADIUS=[0.08, 0.06, 0.03], SUBSURFACE_SCALE=2.5, SUBSURFACE_ANISOTROPY=-0.2),
    "white_marble": dict(_SSS, RGB=[0.98, 0.97, 0.95], ROUGHNESS=0.08, SUBSURFACE_COLOR=[0.95, 0.88, 0.82],
                         SUBSURFACE_RADIUS=[0.02, 0.016, 0.011], SUBSURFACE_SCALE=0.7, SUBSURFACE_ANISOTROPY=0.0),
    "jade": dict(_SSS, RGB=[0.54, 0.77, 0.63], ROUGHNESS=0.15, SUBSURFACE_COLOR=[0.45, 0.70, 0.55],
                 SUBSURFACE_RADIUS=[0.04, 0.055, 0.045], SUBSURFACE_SCALE=1.5, SUBSURFACE_ANISOTROPY=0.1),
    "milk": dict(_SSS, RGB=[1.0, 0.98, 0.95], ROUGHNESS=0.6, TRANSPARENCY=0.2, IOR=1.35,
                 SUBSURFACE_COLOR=[0.95, 0.93, 0.88], SUBSURFACE_RADIUS=[0.10, 0.08, 0.05], SUBSURFACE_SCALE=3.0,
                 SUBSURFACE_ANISOTROPY=-0.8),
    "polished_gold": dict(TYPE="PBR", RGB=[1.0, 0.765, 0.336], ROUGHNESS=0.05, METALLIC=1.0, TRANSPARENCY=0.0),
    "brushed_metal": dict(TYPE="PBR", RGB=[0.7, 0.7, 0.8], ROUGHNESS=0.3, METALLIC=1.0, TRANSPARENCY=0.0),
    "frosted_glass": dict(TYPE="PBR", RGB=[0.95, 0.95, 0.95], IOR=1.5, ROUGHNESS=0.2, METALLIC=0.0, TRANSPARENCY=0.9),
    "plastic_red": dict(TYPE="PBR", RGB=[0.8, 0.2, 0.2], ROUGHNESS=0.4, METALLIC=0.0, TRANSPARENCY=0.0),
    "rough_aluminum": dict(TYPE="PBR", RGB=[0.913, 0.921, 0.925], ROUGHNESS=0.5, METALLIC=1.0, TRANSPARENCY=0.0),
    "semi_transparent_plastic": dict(TYPE="PBR", RGB=[0.3, 0.6, 0.9], ROUGHNESS=0.6, METALLIC=0.0, TRANSPARENCY=0.3),
}
_CORNELL_SET = ["human_skin", "candle_wax", "white_marble", "light", "diffuse_white", "diffuse_blue", "diffuse_red",
                "diffuse_green", "specular_white", "refractive_white", "pbr_glass", "jade", "milk"]


def _obj(kind, material, t, r, s):
    return dict(TYPE=kind, MATERIAL=material, TRANS=list(map(float, t)), ROTAT=list(map(float, r)), SCALE=list(map(float, s)))


def _mesh_obj(name, t, r, s):
    return dict(TYPE="gltf", FILE=name, TRANS=list(map(float, t)), ROTAT=list(map(float, r)), SCALE=list(map(float, s)))


def _room(back="diffuse_white", light_scale=(3.0, 0.1, 3.0)):
    """light + floor + ceiling + back wall + red / green side walls (the 10-unit box all room scenes share)."""
    return [
        _obj("cube", "light", (0, 10, 0), (0, 0, 0), light_scale),
        _obj("cube", "diffuse_white", (0, 0, 0), (0, 0, 0), (10, 0.01, 10)),
        _obj("cube", "diffuse_white", (0, 10, 0), (0, 0, 90), (0.01, 10, 10)),
        _obj("cube", back, (0, 5, -5), (0, 90, 0), (0.01, 10, 10)),
        _obj("cube", "diffuse_red", (-5, 5, 0), (0, 0, 0), (0.01, 10, 10)),
        _obj("cube", "diffuse_green", (5, 5, 0), (0, 0, 0), (0.01, 10, 10)),
    ]


def _camera(name, depth, iters, eye=(0.0, 5.0, 10.5), look=(0.0, 5.0, 0.0), res=(1200, 1200)):
    return dict(RES=list(res), FOVY=45.0, ITERATIONS=iters, DEPTH=depth, FILE=name, EYE=list(eye), LOOKAT=list(look),
                UP=[0.0, 1.0, 0.0])


def _mats(names, **override):
    out = {n: dict(MATERIALS[n]) for n in names}
    for k, v in override.items():
        out[k] = v
    return out


def scene_dict(name):
    """JSON content of a named configuration (reference schema)."""
    if name == "cornell":       # scenes/cornell.json: 9 geoms, 13 materials, depth 12
        objs = _room(back="specular_white") + [
            _obj("sphere", "specular_white", (-2, 3, -1), (0, 0, 0), (3, 3, 3)),
            _obj("sphere", "diffuse_white", (2, 3, -1), (0, 0, 0), (3, 3, 3)),
            _obj("sphere", "refractive_white", (0, 6, -1), (0, 0, 0), (3, 3, 3))]
        return dict(Materials=_mats(_CORNELL_SET), Camera=_camera("cornell", 12, 5000), Objects=objs)
    if name == "sss":           # scenes/sss.json: room + one `milk` sphere, depth 8
        objs = _room() + [_obj("sphere", "milk", (0, 8, -1), (0, 0, 0), (3, 3, 3))]
        return dict(Materials=_mats(_CORNELL_SET), Camera=_camera("sss", 8, 1000, eye=(0.0, 6.5, 5.9), look=(0.0, 6.5, 0.0)),
                    Objects=objs)
    if name == "cornell_orig":  # scenes/cornell-orig.json
        objs = _room(light_scale=(3.0, 0.3, 3.0)) + [_obj("sphere", "diffuse_white", (-1, 4, -1), (0, 0, 0), (3, 3, 3))]
        return dict(Materials=_mats(["light", "diffuse_white", "diffuse_red", "diffuse_green", "specular_white"]),
                    Camera=_camera("cornell-orig", 8, 5000), Objects=objs)
    if name == "pbr":           # scenes/pbr.json: six PBR spheres + a cube light behind the camera
        names = ["brushed_metal", "polished_gold", "frosted_glass", "plastic_red", "rough_aluminum", "semi_transparent_plastic"]
        objs = [_obj("sphere", n, (-3 + 3 * (i % 3), 3 + 4 * (i // 3), 0), (45 * (i // 3), 45 * (i // 3), 0), (2, 2, 2))
                for i, n in enumerate(names)]
        objs.append(_obj("cube", "light", (0, 4, 12), (90, 0, 0), (10, 0.1, 10)))
        return dict(Materials=_mats(names + ["light"]), Camera=_camera("pbr", 8, 5000), Objects=objs)
    if name == "ssaa":          # scenes/ssaa.json: two lights, rotated mirror cube
        room = _room(light_scale=(3.0, 0.3, 3.0))
        objs = [room[0], _obj("cube", "light", (0, 0, 0), (0, 0, 0), (3.0, 0.3, 3.0))] + room[1:] + [
            _obj("cube", "specular_white", (0, 5, 0), (45, 0, 45), (4, 4, 4))]
        mats = _mats(["light", "diffuse_white", "diffuse_red", "diffuse_green", "specular_white"],
                     diffuse_blue=dict(TYPE="Diffuse", RGB=[0.0, 0.0, 1.0]))
        return dict(Materials=mats, Camera=_camera("ssaa", 8, 1000, eye=(0.0, 5.0, 6.62)), Objects=objs)
    if name == "spheres":       # scenes/sphere.json: three spheres, no room (black without an env map)
        objs = [_obj("sphere", "specular_white", (-2, 3, -1), (0, 0, 0), (3, 3, 3)),
                _obj("sphere", "diffuse_white", (2, 3, -1), (0, 0, 0), (3, 3, 3)),
                _obj("sphere", "refractive_white", (0, 6, -1), (0, 0, 0), (3, 3, 3))]
        return dict(Materials=_mats(["light", "diffuse_white", "diffuse_blue", "diffuse_red", "diffuse_green",
                                     "specular_white", "refractive_white"]), Camera=_camera("sphere", 8, 5000), Objects=objs)
    base = _mats(["light", "diffuse_white", "polished_gold"])
    if name == "dragon":        # scenes/dragon.json: cube light + one mesh, depth 8
        objs = [_obj("cube", "light", (11.71, 10.0, 13.22), (90, 0, -45), (10, 0.1, 10)),
                _mesh_obj("dragon", (0, 1, 0), (0, 0, 0), (2, 2, 2))]
        return dict(Materials=base, Camera=_camera("gltf", 8, 1000), Objects=objs)
    if name == "chess":         # scenes/chess.json: depth 10
        objs = [_obj("cube", "light", (10.0, 6.0, 13.22), (90, 0, -45), (10, 0.1, 10)),
                _mesh_obj("chess", (-0.1, 0.3, 0.4), (0, 25, 0), (5, 5, 5))]
        return dict(Materials=base, Camera=_camera("chess", 10, 2500), Objects=objs)
    if name == "p911":          # scenes/911.json: light emittance 1
        objs = [_obj("cube", "light", (-9.5, 2.0, 4.0), (90, 0, 22.5), (10, 0.1, 10)),
                _mesh_obj("911", (-0.3, -0.55, -1.2), (0, 22.5, 0), (1.85, 1.85, 1.85))]
        mats = dict(base, light=dict(TYPE="Emitting", RGB=[1.0, 1.0, 1.0], EMITTANCE=1.0))
        return dict(Materials=mats, Camera=_camera("911", 8, 5000), Objects=objs)
    if name == "halo":          # scenes/halo.json: two lights
        objs = [_obj("cube", "light1", (-9.5, 2.0, 4.0), (90, 0, 22.5), (10, 0.1, 10)),
                _obj("cube", "light", (3.39, 4.02, 13.38), (90, 0, -22.5), (10, 0.1, 10)),
                _mesh_obj("halo", (-0.2, -9.0, -1.0), (0, -10, 0), (4.3, 4.3, 4.3))]
        mats = dict(base, light1=dict(TYPE="Emitting", RGB=[1.0, 1.0, 1.0], EMITTANCE=2.0))
        return dict(Materials=mats, Camera=_camera("halo", 8, 5000), Objects=objs)
    if name == "duck":          # scenes/duck.json: one mesh, no light geom
        objs = [_mesh_obj("duck", (-0.5, 0, 0), (0, -22.5, 0), (2.5, 2.5, 2.5))]
        return dict(Materials=base, Camera=_camera("duck", 8, 5000), Objects=objs)
    raise KeyError(name)


SCENE_NAMES = ["cornell", "sss", "cornell_orig", "pbr", "ssaa", "spheres", "dragon", "chess", "p911", "halo", "duck"]


def write_scene_json(name, directory=None):
    directory = directory or os.path.join(tempfile.gettempdir(), "mi_pathtrace_scenes_%d" % os.getuid())
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, name + ".json")
    # written under a name of its own and renamed into place: several processes may prepare the same scene at once
    # (ranks of a multi-GPU run that share a directory), and none of them may ever read a half-written file
    tmp = "%s.%d.tmp" % (path, os.getpid())
    with open(tmp, "w") as f:
        json.dump(scene_dict(name), f, indent=1)
    os.replace(tmp, path)
    return path


# ---------------------------------------------------------------- procedural assets
def _bake(local_pos, local_nrm, trs):
    """Apply the object TRS to vertices / normals the way the glTF import bakes it into
    the triangles (src/scene.cpp:265, :423-436)."""
    M, _, IT = build_transforms(*trs)          # [col][row]
    M = M.T.astype(np.float64)
    IT = IT.T.astype(np.float64)
    p = local_pos.astype(np.float64) @ M[:3, :3].T + M[:3, 3]
    n = local_nrm.astype(np.float64) @ IT[:3, :3].T
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)
    return p.astype(np.float32), n.astype(np.float32)


def blob_mesh(n_lon, n_lat, seed, trs, material_ids, bumps=6, amplitude=0.18, radius=1.0, center=(0.0, 0.0, 0.0),
              squash=(1.0, 1.0, 1.0)):
    """Closed lat/long mesh with seeded sinusoidal displacement: 2*n_lon*(n_lat-1) triangles.
    material_ids are assigned in latitude/longitude patches."""
    rng = np.random.RandomState(seed)
    th = np.linspace(0.0, np.pi, n_lat + 1)
    ph = np.linspace(0.0, 2.0 * np.pi, n_lon + 1)
    TH, PH = np.meshgrid(th, ph, indexing="ij")
    disp = np.zeros_like(TH)
    for _ in range(bumps):
        a, b = rng.randint(1, 9), rng.randint(1, 7)
        disp += rng.uniform(0.3, 1.0) * np.sin(a * PH + rng.uniform(0, 6.28)) * np.sin(b * TH + rng.uniform(0, 6.28)) * np.sin(TH)
    r = radius * (1.0 + amplitude * disp / bumps * 2.0)
    d = np.stack([np.sin(TH) * np.cos(PH), np.cos(TH), np.sin(TH) * np.sin(PH)], -1)
    pos = d * r[..., None] * np.asarray(squash) + np.asarray(center)
    pos[:, -1] = pos[:, 0]                       # close the seam exactly
    # normals from finite differences of the grid
    dth = np.gradient(pos, axis=0)
    dph = np.gradient(pos, axis=1)
    nrm = np.cross(dph, dth)
    ln = np.linalg.norm(nrm, axis=-1, keepdims=True)
    nrm = np.where(ln > 1e-12, nrm / np.maximum(ln, 1e-12), d)
    uv = np.stack([PH / (2 * np.pi), TH / np.pi], -1)
    P, N = _bake(pos.reshape(-1, 3), nrm.reshape(-1, 3), trs)
    P = P.reshape(pos.shape); N = N.reshape(pos.shape)
    quads_i, quads_j = np.meshgrid(np.arange(n_lat), np.arange(n_lon), indexing="ij")
    i, j = quads_i.ravel(), quads_j.ravel()
    tri_a = (np.stack([i, j], 1), np.stack([i + 1, j], 1), np.stack([i + 1, j + 1], 1))
    tri_b = (np.stack([i, j], 1), np.stack([i + 1, j + 1], 1), np.stack([i, j + 1], 1))
    keep_a = i < n_lat - 0            # degenerate rows at the poles are dropped below
    out = []
    mids = np.asarray(material_ids, dtype=np.int32)
    npatch = len(mids)
    patch = ((i * 4 // n_lat) * 5 + (j * 5 // n_lon)) % npatch
    for (a, b, c), rows in ((tri_a, i != 0), (tri_b, i != n_lat - 1)):
        sel = rows & keep_a
        t = np.zeros(int(sel.sum()), dtype=TRI_DTYPE)
        for name_v, name_n, name_uv, idx in (("v0", "n0", "uv0", a), ("v1", "n1", "uv1", b), ("v2", "n2", "uv2", c)):
            t[name_v] = P[idx[sel, 0], idx[sel, 1]]
            t[name_n] = N[idx[sel, 0], idx[sel, 1]]
            t[name_uv] = uv[idx[sel, 0], idx[sel, 1]].astype(np.float32)
        t["materialId"] = mids[patch[sel]]
        out.append(t)
    return np.concatenate(out)


def _grid_mesh(pos, nrm, uv, mat_of_quad, flip=False):
    """(nu+1, nv+1) grids of positions / normals / uvs -> 2*nu*nv triangles (TRI_DTYPE); mat_of_quad: (nu, nv) ints."""
    nu, nv = pos.shape[0] - 1, pos.shape[1] - 1
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    i, j = i.ravel(), j.ravel()
    corners = ((i, j), (i + 1, j), (i + 1, j + 1), (i, j + 1))
    order = ((0, 1, 2), (0, 2, 3)) if not flip else ((0, 2, 1), (0, 3, 2))
    out = np.zeros(2 * nu * nv, dtype=TRI_DTYPE)
    for k, tri in enumerate(order):
        t = out[k::2]
        for (name_v, name_n, name_uv), c in zip((("v0", "n0", "uv0"), ("v1", "n1", "uv1"), ("v2", "n2", "uv2")), tri):
            a, b = corners[c]
            t[name_v] = pos[a, b]
            t[name_n] = nrm[a, b] if not flip else -nrm[a, b]
            t[name_uv] = uv[a, b]
        t["materialId"] = mat_of_quad[i, j]
    return out


def _grid_normals(pos, wrap_u, wrap_v):
    """unit normals of a parametric grid by central differences (periodic where the surface closes on itself)"""
    def diff(a, axis, wrap):
        if wrap:          # first and last sample coincide
            core = np.take(a, range(a.shape[axis] - 1), axis=axis)
            d = np.roll(core, -1, axis=axis) - np.roll(core, 1, axis=axis)
            return np.concatenate([d, np.take(d, [0], axis=axis)], axis=axis)
        return np.gradient(a, axis=axis)
    n = np.cross(diff(pos, 0, wrap_u), diff(pos, 1, wrap_v))
    ln = np.linalg.norm(n, axis=-1, keepdims=True)
    return np.where(ln > 1e-20, n / np.maximum(ln, 1e-20), np.array([0.0, 1.0, 0.0]))


def _patch_materials(nu, nv, mids, pu, pv, shift=0):
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    mids = np.asarray(mids, np.int32)
    return mids[((i * pu // max(nu, 1)) * (pv + 1) + (j * pv // max(nv, 1)) + shift) % len(mids)]


def _split(budget, aspect):
    """(nu, nv) with 2 * nu * nv <= budget, nu / nv about aspect"""
    nv = max(3, int(np.sqrt(budget / 2.0 / aspect)))
    nu = max(3, int(budget // (2 * nv)))
    return nu, nv


def _knot_tube(budget, rng, mids, p=2, q=3, r0=0.13):
    """closed tube around a (p, q) torus knot with a rippled radius: strongly concave, parts of it pass close to
    other parts"""
    nu, nv = _split(budget, 14.0)
    u = np.linspace(0.0, 2.0 * np.pi, nu + 1)
    v = np.linspace(0.0, 2.0 * np.pi, nv + 1)

    def curve(t):
        r = 0.62 + 0.26 * np.cos(q * t)
        return np.stack([r * np.cos(p * t), 0.30 * np.sin(q * t), r * np.sin(p * t)], -1)
    c = curve(u)
    tan = curve(u + 1e-4) - curve(u - 1e-4)
    tan /= np.linalg.norm(tan, axis=-1, keepdims=True)
    side = np.cross(tan, np.array([0.0, 1.0, 0.0]))
    side /= np.maximum(np.linalg.norm(side, axis=-1, keepdims=True), 1e-9)
    up = np.cross(side, tan)
    U, V = np.meshgrid(u, v, indexing="ij")
    a, b, ph = rng.randint(5, 12), rng.randint(3, 7), rng.uniform(0, 6.28)
    rad = r0 * (1.0 + 0.35 * np.sin(a * U + ph) * np.cos(b * V) + 0.18 * np.sin(23 * U + 2 * V))
    pos = c[:, None, :] + rad[..., None] * (np.cos(V)[..., None] * side[:, None, :] + np.sin(V)[..., None] * up[:, None, :])
    pos[-1] = pos[0]
    pos[:, -1] = pos[:, 0]
    nrm = _grid_normals(pos, True, True)
    uv = np.stack([U / (2 * np.pi) * 6.0, V / (2 * np.pi)], -1)
    return _grid_mesh(pos, nrm, uv, _patch_materials(nu, nv, mids, 9, 2))


def _thin_sheets(budget, rng, mids, count=2, thickness=0.004):
    """curved sheets with a front and a back layer a few thousandths apart (thin shells with open rims)"""
    out = []
    per = budget // (2 * count)
    for k in range(count):
        nu, nv = _split(per, 1.6)
        U, V = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="ij")
        sgn = 1.0 if k % 2 == 0 else -1.0
        f1, f2, ph = rng.uniform(2, 5), rng.uniform(3, 7), rng.uniform(0, 6.28)
        x = sgn * (0.25 + 0.75 * U)
        y = 0.15 + 0.55 * V * (1.0 - 0.5 * U) + 0.06 * np.sin(f1 * np.pi * U + ph) * np.sin(f2 * np.pi * V)
        z = -0.15 - 0.35 * U * U + 0.08 * np.cos(f2 * np.pi * U) * V + 0.25 * (k // 2)
        pos = np.stack([x, y, z], -1)
        nrm = _grid_normals(pos, False, False)
        uv = np.stack([U * 3.0, V * 3.0], -1)
        m = _patch_materials(nu, nv, mids, 3, 3, shift=k)
        out.append(_grid_mesh(pos, nrm, uv, m))
        out.append(_grid_mesh(pos - thickness * nrm, nrm, uv, m, flip=True))
    return np.concatenate(out)


def _shell(budget, rng, mids, radius, center, bumps, amp, squash=(1.0, 1.0, 1.0)):
    """closed displaced lat/long shell without the degenerate pole rows"""
    nu, nv = _split(budget, 0.5)                       # nu: latitude, nv: longitude
    th = np.linspace(0.02, np.pi - 0.02, nu + 1)
    ph = np.linspace(0.0, 2.0 * np.pi, nv + 1)
    TH, PH = np.meshgrid(th, ph, indexing="ij")
    disp = np.zeros_like(TH)
    for _ in range(bumps):
        disp += rng.uniform(0.3, 1.0) * np.sin(rng.randint(1, 9) * PH + rng.uniform(0, 6.28)) * np.sin(rng.randint(1, 7) * TH + rng.uniform(0, 6.28))
    r = radius * (1.0 + amp * disp / max(bumps, 1))
    d = np.stack([np.sin(TH) * np.cos(PH), np.cos(TH), np.sin(TH) * np.sin(PH)], -1)
    pos = d * r[..., None] * np.asarray(squash) + np.asarray(center)
    pos[:, -1] = pos[:, 0]
    nrm = -_grid_normals(pos, False, True)
    uv = np.stack([PH / (2 * np.pi) * 2.0, TH / np.pi * 2.0], -1)
    return _grid_mesh(pos, nrm, uv, _patch_materials(nu, nv, mids, 4, 5), flip=True)


def _beads(budget, rng, mids, count, radius, path):
    """many small, finely tessellated shells strung along a path (the small end of the triangle-size range);
    neighbours overlap, and all of them dig into whatever the path runs through"""
    out = []
    per = budget // count
    for k in range(count):
        c = path(k / float(count))
        out.append(_shell(per, rng, [mids[k % len(mids)]], radius * rng.uniform(0.7, 1.3), c, 3, 0.25))
    return np.concatenate(out)


def _coarse_slab(mids, y=-0.42, half=1.25, n=2):
    """a few very large triangles (the big end of the size range): an n x n grid plate under the model"""
    g = np.linspace(-half, half, n + 1)
    X, Z = np.meshgrid(g, g, indexing="ij")
    pos = np.stack([X, np.full_like(X, y), Z], -1)
    nrm = np.tile(np.array([0.0, 1.0, 0.0]), X.shape + (1,))
    uv = np.stack([(X + half) / (2 * half) * 4, (Z + half) / (2 * half) * 4], -1)
    return _grid_mesh(pos, nrm, uv, _patch_materials(n, n, mids, n, n), flip=True)


def _ribbon(triangles, rng, mids, width=0.012, turns=9.0):
    """a narrow strip of exactly `triangles` (even) small triangles spiralling around the model"""
    nq = triangles // 2
    if nq <= 0:
        return np.zeros(0, dtype=TRI_DTYPE)
    t = np.linspace(0.0, 1.0, nq + 1)
    ang = 2 * np.pi * turns * t
    r = 0.95 - 0.35 * t
    c = np.stack([r * np.cos(ang), -0.35 + 0.9 * t, r * np.sin(ang)], -1)
    up = np.array([0.0, 1.0, 0.0])
    pos = np.stack([c, c + width * up], 1)                       # (nq+1, 2, 3)
    nrm = np.stack([np.cos(ang), np.zeros_like(ang), np.sin(ang)], -1)
    nrm = np.stack([nrm, nrm], 1)
    uv = np.stack([np.stack([t * 40, np.zeros_like(t)], -1), np.stack([t * 40, np.ones_like(t)], -1)], 1)
    return _grid_mesh(pos, nrm, uv, _patch_materials(nq, 1, mids, 16, 1))


def composite_mesh(style, total, seed, trs, material_ids, center=(0.0, 0.0, 0.0), radius=1.0, squash=(1.0, 1.0, 1.0), tilt=(0.0, 0.0)):
    """Seeded stand-in model of exactly `total` (even) triangles for a scene class whose asset is not distributed.
    Unlike a single displaced sphere it has what scanned and modelled assets have and what makes a BVH work for its
    living: deep concavities and parts that pass close to each other (knot tube), thin two-layer shells with open rims,
    nested closed shells (an outer one around an inner one, so rays enter, cross a gap and hit again), parts that
    intersect each other (beads dug into the tube), and triangle edges from about 1 down to about 0.005 of the model
    radius (a coarse plate, beads of a few hundred tiny triangles each).  style picks the mix:
      dragon  one knot, two wings, nested core, beads along the knot
      chess   a coarse board and 32 pieces (stacked nested shells + beads) on an 8 x 4 + 8 x 4 layout
      vehicle (911 / halo class) a big thin two-layer hull, four knot "wheels", nested interior, many beads
    material_ids are spread over the parts in patches; id [1] (the transmissive material of the stand-in material set)
    goes to the outer nested shell."""
    rng = np.random.RandomState(seed)
    mids = list(material_ids)
    glass = [mids[1 % len(mids)]]
    parts = []
    slab = _coarse_slab(mids, n=2 if style != "chess" else 8)
    budget = total - len(slab)
    if style == "dragon":
        parts.append(_knot_tube(int(budget * 0.42), rng, mids))
        parts.append(_thin_sheets(int(budget * 0.16), rng, mids, count=2))
        parts.append(_shell(int(budget * 0.10), rng, glass, 0.30, (0.0, 0.05, 0.0), 6, 0.35))
        parts.append(_shell(int(budget * 0.06), rng, mids[2:] or mids, 0.17, (0.02, 0.05, 0.0), 5, 0.5))
        knot = lambda t: np.array([(0.62 + 0.26 * np.cos(6 * np.pi * t)) * np.cos(4 * np.pi * t), 0.30 * np.sin(6 * np.pi * t) + 0.12,
                                   (0.62 + 0.26 * np.cos(6 * np.pi * t)) * np.sin(4 * np.pi * t)])
        parts.append(_beads(int(budget * 0.22), rng, mids, 60, 0.035, knot))
    elif style == "chess":
        per_piece = int(budget * 0.97) // 32
        for k in range(32):
            col, row = k % 8, k // 8
            cx = (col - 3.5) * 0.27
            cz = (row - 1.5) * 0.27 * (1.0 if row < 2 else 1.0) + (0.55 if row >= 2 else -0.55) * 0.0
            cz = (-0.95 + 0.27 * row) if row < 2 else (0.41 + 0.27 * (row - 2))
            h = 0.16 + 0.05 * ((k * 7) % 5)
            m = [mids[(k + j) % len(mids)] for j in range(3)]
            piece = [_shell(int(per_piece * 0.30), rng, m[:1], 0.105, (cx, -0.42 + 0.07, cz), 2, 0.15, squash=(1.0, 0.55, 1.0)),
                     _shell(int(per_piece * 0.25), rng, m[1:2], 0.07, (cx, -0.42 + 0.07 + h * 0.5, cz), 3, 0.3, squash=(1.0, 1.8, 1.0)),
                     _shell(int(per_piece * 0.15), rng, glass if k % 5 == 0 else m[2:3], 0.085, (cx, -0.42 + 0.09 + h, cz), 4, 0.35),
                     _shell(int(per_piece * 0.10), rng, m[:1], 0.05, (cx, -0.42 + 0.09 + h, cz), 3, 0.4)]      # nested in the head
            ring = lambda t, cx=cx, cz=cz, h=h: np.array([cx + 0.09 * np.cos(2 * np.pi * t), -0.42 + 0.05 + h * 0.8, cz + 0.09 * np.sin(2 * np.pi * t)])
            piece.append(_beads(int(per_piece * 0.20), rng, m, 10, 0.018, ring))
            parts.append(np.concatenate(piece))
    else:
        parts.append(_thin_sheets(int(budget * 0.10), rng, mids, count=4, thickness=0.003))
        hull = _shell(int(budget * 0.22), rng, glass + mids[2:5], 0.78, (0.0, 0.0, 0.0), 8, 0.22, squash=(1.0, 0.42, 0.55))
        inner = _shell(int(budget * 0.12), rng, mids, 0.76, (0.0, 0.0, 0.0), 8, 0.22, squash=(1.0, 0.40, 0.53))     # second layer close inside the first
        parts += [hull, inner]
        for w in range(4):
            wheel = _knot_tube(int(budget * 0.07), rng, mids, p=1, q=int(5 + w), r0=0.10)
            for f in ("v0", "v1", "v2"):
                wheel[f] = (wheel[f] * np.array([0.28, 0.28, 0.28]) + np.array([0.5 * (1 if w % 2 else -1), -0.3, 0.36 * (1 if w // 2 else -1)])).astype(np.float32)
            parts.append(wheel)
        parts.append(_shell(int(budget * 0.08), rng, mids, 0.3, (0.1, 0.0, 0.0), 6, 0.5))
        parts.append(_knot_tube(int(budget * 0.08), rng, mids, p=3, q=4, r0=0.05))
        line = lambda t: np.array([-0.75 + 1.5 * t, 0.18 + 0.1 * np.sin(9 * t), 0.30 * np.sin(25 * t)])
        parts.append(_beads(int(budget * 0.11), rng, mids, 160, 0.02, line))
    used = len(slab) + sum(len(q) for q in parts)
    rest = total - used
    assert rest >= 0 and rest % 2 == 0, (total, used)
    parts.append(_ribbon(rest, rng, mids))
    tris = np.concatenate([slab] + parts)
    assert len(tris) == total
    # model frame -> turned towards the camera (tilt: degrees about x, then about y) -> the place the scene wants
    # it, then the object's TRS baked in as the glTF import bakes it
    sq = np.asarray(squash, np.float64)
    ctr = np.asarray(center, np.float64)
    ax, ay = np.radians(tilt[0]), np.radians(tilt[1])
    Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    R = Ry @ Rx
    for fv, fn in (("v0", "n0"), ("v1", "n1"), ("v2", "n2")):
        local = (tris[fv].astype(np.float64) @ R.T) * radius * sq + ctr
        nl = (tris[fn].astype(np.float64) @ R.T) / sq
        nl /= np.maximum(np.linalg.norm(nl, axis=1, keepdims=True), 1e-30)
        P, N = _bake(local, nl, trs)
        tris[fv], tris[fn] = P, N
    return tris


def cube_mesh(trs, material_id):
    """Axis-aligned unit cube as 12 triangles (the edge-case mesh INSTRUCTION.md:215 suggests)."""
    c = np.array([[x, y, z] for x in (-.5, .5) for y in (-.5, .5) for z in (-.5, .5)], np.float32)
    faces = [(0, 1, 3, 2, (-1, 0, 0)), (4, 6, 7, 5, (1, 0, 0)), (0, 4, 5, 1, (0, -1, 0)), (2, 3, 7, 6, (0, 1, 0)),
             (0, 2, 6, 4, (0, 0, -1)), (1, 5, 7, 3, (0, 0, 1))]
    tris = np.zeros(12, dtype=TRI_DTYPE)
    k = 0
    for a, b, cc, d, n in faces:
        for tri in ((a, b, cc), (a, cc, d)):
            pos = c[list(tri)]
            nn = np.tile(np.asarray(n, np.float32), (3, 1))
            P, N = _bake(pos, nn, trs)
            tris[k]["v0"], tris[k]["v1"], tris[k]["v2"] = P
            tris[k]["n0"], tris[k]["n1"], tris[k]["n2"] = N
            tris[k]["uv0"], tris[k]["uv1"], tris[k]["uv2"] = (0, 0), (1, 0), (1, 1)
            tris[k]["materialId"] = material_id
            k += 1
    return tris


def sky_hdr(width=2048, height=1024, seed=1, sun=(0.35, 0.25), sun_power=60.0):
    """Procedural equirect HDR: horizon gradient + warm sun disc + seeded soft clouds."""
    rng = np.random.RandomState(seed)
    v = (np.arange(height) + 0.5) / height
    u = (np.arange(width) + 0.5) / width
    U, V = np.meshgrid(u, v)
    up = np.clip(1.0 - 2.0 * V, -1.0, 1.0)
    sky = np.stack([0.35 + 0.25 * (1 - up), 0.5 + 0.2 * (1 - up), 0.85 + 0.05 * up], -1) * np.clip(up + 0.3, 0.05, 1.0)[..., None]
    ground = np.array([0.18, 0.16, 0.14])
    img = np.where((up > 0)[..., None], sky, ground * np.clip(1 + up, 0.2, 1)[..., None])
    clouds = np.zeros_like(U)
    for _ in range(5):
        clouds += np.sin(rng.randint(2, 12) * 2 * np.pi * U + rng.uniform(0, 6.28)) * np.sin(rng.randint(2, 9) * np.pi * V + rng.uniform(0, 6.28))
    img = img * (1.0 + 0.15 * clouds[..., None] * (up > 0)[..., None])
    du = np.minimum(np.abs(U - sun[0]), 1 - np.abs(U - sun[0])) * 2 * np.pi * np.sin(np.pi * sun[1])
    dv = (V - sun[1]) * np.pi
    ang2 = du * du + dv * dv
    img = img + sun_power * np.exp(-ang2 / (2 * 0.03 ** 2))[..., None] * np.array([1.0, 0.9, 0.7])
    return np.clip(img, 0.0, None).astype(np.float32)


def checker_texture(size, seed, comps=4, cells=16):
    rng = np.random.RandomState(seed)
    a, b = rng.randint(40, 255, 3), rng.randint(40, 255, 3)
    y, x = np.mgrid[0:size, 0:size]
    m = (((x * cells // size) + (y * cells // size)) & 1).astype(bool)
    img = np.where(m[..., None], a, b).astype(np.uint8)
    img = (img.astype(np.int32) + rng.randint(-12, 12, (size, size, 1))).clip(0, 255).astype(np.uint8)
    if comps == 4:
        alpha = np.full((size, size, 1), 255, np.uint8)
        img = np.concatenate([img, alpha], -1)
    return img


def metal_rough_texture(size, seed):
    rng = np.random.RandomState(seed)
    y, x = np.mgrid[0:size, 0:size]
    g = (128 + 100 * np.sin(x * (2 * np.pi * rng.randint(1, 6) / size)) * np.cos(y * (2 * np.pi * rng.randint(1, 6) / size)))
    b = (128 + 120 * np.sin((x + y) * (2 * np.pi * rng.randint(1, 4) / size)))
    img = np.stack([np.zeros_like(g), g, b], -1).clip(0, 255).astype(np.uint8)
    return img


# name -> (triangles target, #mesh materials, textured?, env?, grid)
_ASSET_CLASS = {
    "dragon": dict(grid=(260, 261), total=135200, style="dragon", nmat=6, textured=False, env=True, env_intensity=0.6),     # README: 134 995
    "chess": dict(grid=(866, 867), total=1499912, style="chess", nmat=18, textured=True, env=True, env_intensity=0.7),      # README: 1.5 M
    "p911": dict(grid=(1225, 1226), total=3001250, style="vehicle", nmat=16, textured=True, env=True, env_intensity=3.0),   # multi-million class
    "halo": dict(grid=(1225, 1226), total=3001250, style="vehicle", nmat=13, textured=True, env=True, env_intensity=1.0),
    "duck": dict(grid=(46, 47), nmat=1, textured=True, env=True, env_intensity=1.0),          # 4 232 tris blob (the real Duck.glb: make_duck_scene)
}


def _mesh_materials(scene, name, cfg, tex_size):
    ids = []
    rng = np.random.RandomState(sum(map(ord, name)) % 1000 + 7)
    for k in range(cfg["nmat"]):
        kw = dict(type=PT_PBR, color=[float(c) for c in rng.uniform(0.35, 1.0, 3)],
                  roughness=float(rng.uniform(0.05, 0.9)), metallic=float(rng.choice([0.0, 0.0, 1.0, 0.5])),
                  transparency=0.0, indexOfRefraction=1.5, subsurfaceColor=[1.0, 1.0, 1.0], subsurfaceRadius=0.01,
                  subsurfaceRadiusRGB=[0.01, 0.01, 0.01], subsurfaceScale=1.0, subsurfaceEnabled=0)
        if k == 1:                      # one transmissive material (attenuating glass body)
            kw.update(transparency=0.9, roughness=0.05, metallic=0.0, color=[0.95, 0.75, 0.55])
        if cfg["textured"]:
            kw["baseColorTextureIdx"] = scene.add_texture(checker_texture(tex_size, 100 + k))
            kw["metallicRoughnessTextureIdx"] = scene.add_texture(metal_rough_texture(tex_size, 200 + k))
            kw["metallic"] = 1.0 if k % 3 == 0 else kw["metallic"]
        ids.append(scene.add_material(**kw))
    return ids


def make_scene(name, res=None, depth=None, directory=None, tex_size=1024, env_size=(2048, 1024), grid=None, env=None):
    """Load a named configuration through the C++ loader and attach procedural assets."""
    sc = Scene(write_scene_json(name, directory))
    if res is not None:
        sc.set_resolution(*res)
    if depth is not None:
        sc.set_depth(depth)
    cfg = _ASSET_CLASS.get(name)
    if cfg is not None:
        mesh_obj = [o for o in scene_dict(name)["Objects"] if o["TYPE"] == "gltf"][0]
        trs = (mesh_obj["TRANS"], mesh_obj["ROTAT"], mesh_obj["SCALE"])
        ids = _mesh_materials(sc, name, cfg, tex_size)
        g = grid or cfg["grid"]
        # The object's TRS reaches the rays twice (baked into the triangles and again as the geom transform,
        # src/scene.cpp:265 + src/intersections.cu:325-326), so place the local blob where M*M maps it onto
        # the part of the room the camera looks at.
        M = build_transforms(*trs)[0].T.astype(np.float64)
        M2 = M @ M
        s2 = abs(np.linalg.det(M2[:3, :3])) ** (1.0 / 3.0)
        centre = np.linalg.solve(M2, np.array([0.0, 5.0, 0.0, 1.0]))[:3]
        if grid is None and "style" in cfg:
            # full size: the composite model (knot tube, thin shells, nested shells, beads, coarse plate)
            tris = composite_mesh(cfg["style"], cfg["total"], seed=1, trs=trs, material_ids=ids, center=tuple(centre),
                                  radius=7.5 / s2, squash=(1.25, 1.0, 1.0),
                                  tilt={"dragon": (65.0, 15.0), "chess": (38.0, 0.0), "vehicle": (25.0, 35.0)}[cfg["style"]])
        else:
            # a forced grid (small test scenes): one displaced closed blob of 2 * n_lon * (n_lat - 1) triangles
            tris = blob_mesh(g[0], g[1], seed=1, trs=trs, material_ids=ids, center=tuple(centre), squash=(1.25, 1.0, 0.8),
                             radius=5.5 / s2, bumps=10, amplitude=0.45)
        sc.add_mesh(tris, *trs)
        want_env = cfg["env"] if env is None else env
        if want_env:
            sc.set_envmap(sky_hdr(env_size[0], env_size[1]), cfg["env_intensity"])
    elif env:
        sc.set_envmap(sky_hdr(env_size[0], env_size[1]), 1.0)
    return sc


GOLDEN_DUCK = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "Duck.glb")


def make_duck_scene(res=None, depth=None, directory=None, glb=None, env_size=(2048, 1024)):
    """scenes/duck.json with the reference's own gltf/Duck.glb (kept as a data fixture under tests/golden) through
    the glTF import.  The scene file names an environment map the reference does not distribute and has no light
    geometry, so a procedural sky stands in for the map."""
    d = scene_dict("duck")
    d["Objects"][0]["FILE"] = glb or GOLDEN_DUCK
    directory = directory or os.path.join(tempfile.gettempdir(), "mi_pathtrace_scenes_%d" % os.getuid())
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, "duck_real.json")
    tmp = "%s.%d.tmp" % (path, os.getpid())
    with open(tmp, "w") as f:
        json.dump(d, f, indent=1)
    os.replace(tmp, path)
    sc = Scene(path)
    if res is not None:
        sc.set_resolution(*res)
    if depth is not None:
        sc.set_depth(depth)
    sc.set_envmap(sky_hdr(env_size[0], env_size[1]), 1.0)
    return sc
