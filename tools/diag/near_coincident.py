"""Diagnostic: rays of tests/test_gpu_full_size.py::test_near_coincident_surfaces_bit_exact on which two closest-hit
implementations differ (oracle vs reference tool on the CPU; GPU vs oracle when a GPU is present)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry

pt = entry._load_package()
scenes = importlib.import_module("cuda_path_tracer_amd.scenes")
import oracle_binding as orc
import ref_binding as ref
import test_gpu_full_size as T


def build():
    sc = scenes.make_scene("cornell_orig", res=(64, 64), depth=8)
    ident = ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    m0 = sc.add_material(type=pt.PT_PBR, color=[0.9, 0.6, 0.3], roughness=0.4, metallic=0.0, indexOfRefraction=1.5)
    m1 = sc.add_material(type=pt.PT_PBR, color=[0.3, 0.6, 0.9], roughness=0.2, metallic=1.0, indexOfRefraction=1.5)
    rng = np.random.RandomState(17)
    a = scenes.blob_mesh(48, 49, 21, ident, [m0], radius=1.8, center=(-1.5, 4.0, 0.0))
    b = a.copy()
    for f in ("v0", "v1", "v2"):
        b[f] = T._ulp_shift(a[f], rng)
    b["materialId"] = m1
    sc.add_mesh(np.concatenate([a, b]))
    c = scenes.blob_mesh(40, 41, 22, ident, [m0], radius=1.5, center=(2.0, 4.0, 0.5))
    c2 = c.copy()
    for f in ("v0", "v1", "v2"):
        c2[f] = T._ulp_shift(c[f], rng, k=1)
    c2["materialId"] = m1
    sc.add_mesh(c)
    sc.add_mesh(c2)
    sc.add_mesh(np.concatenate([scenes.cube_mesh(((0.0, 7.5, 0.0), (0.0, 0.0, 0.0), (2.0, 1.0, 2.0)), m0),
                                scenes.cube_mesh(((1.0, 7.5, 0.0), (0.0, 0.0, 0.0), (2.0, 1.0, 2.0)), m1),
                                scenes.cube_mesh(((0.0, 8.5, 0.0), (0.0, 0.0, 0.0), (2.0, 1.0, 2.0)), m1)]))
    ro, rd = T._ray_set(41, 120000, center=(0.0, 5.0, 0.0), radius=9.0)
    xs = np.linspace(-1.4, 2.4, 60, dtype=np.float32)
    gx, gz = np.meshgrid(xs, xs)
    down_o = np.stack([gx.ravel(), np.full(gx.size, 9.5, np.float32), gz.ravel()], 1).astype(np.float32)
    down_d = np.tile(np.array([[0, -1, 0]], np.float32), (gx.size, 1))
    side_o = np.stack([np.full(gx.size, -8.0, np.float32), 7.0 + (gx.ravel() + 1.4) / 3.8 * 2.0, gz.ravel()], 1).astype(np.float32)
    side_d = np.tile(np.array([[1, 0, 0]], np.float32), (gx.size, 1))
    return sc, np.concatenate([ro, down_o, side_o]), np.concatenate([rd, down_d, side_d])


def report(name_a, a, name_b, b, ro, rd, has_tri=True):
    bad = np.flatnonzero((a["geom"] != b["geom"]) | (a["t"].view(np.uint32) != b["t"].view(np.uint32)))
    print("%s vs %s: %d of %d rays differ" % (name_a, name_b, len(bad), len(a)))
    for i in bad[:12]:
        print("  ray %d o=%s d=%s | %s geom %d t %.9g %s| %s geom %d t %.9g" % (
            i, ro[i], rd[i], name_a, a["geom"][i], a["t"][i], ("tri %d " % a["triangle"][i]) if has_tri else "",
            name_b, b["geom"][i], b["t"][i]))
    return bad


if __name__ == "__main__":
    sc, ro, rd = build()
    o = orc.Oracle(pt, sc, orc.MATH_DET).intersect(ro, rd)
    if ref.available():
        report("oracle", o, "reference", ref.intersect(sc, ro, rd), ro, rd)
    import torch
    if torch.cuda.is_available():
        g = pt.Tracer(sc).intersect(ro, rd)
        bad = report("gpu", g, "oracle", o, ro, rd)
        for i in bad[:12]:
            print("    oracle tri %d mat %d | gpu mat %d" % (o["triangle"][i], o["materialId"][i], g["materialId"][i]))
