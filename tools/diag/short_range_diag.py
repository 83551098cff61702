#!/usr/bin/env python3
"""Diagnosis of the large-triangles-at-short-range case (tests/test_gpu_full_size.py): runs the ray set on the GPU and
on the oracle, and writes the rays on which they differ, with both hit records and the brute-force answer, to
gpurun_out/short_range_diff.npz."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import conftest  # noqa: E402

pt = conftest.load_package()
scenes = importlib.import_module("cuda_path_tracer_amd.scenes")
import oracle_binding as orc  # noqa: E402
import test_gpu_full_size as T  # noqa: E402

sc = scenes.make_scene("cornell_orig", res=(48, 48), depth=8)
m = [sc.add_material(type=pt.PT_PBR, color=[0.2 + 0.1 * k, 0.5, 0.9 - 0.1 * k], roughness=0.4, metallic=0.0, indexOfRefraction=1.5) for k in range(4)]
rng = np.random.RandomState(29)
sc.add_mesh(T._big_quads(pt, rng, 9, 8.0, m))
ro, rd = T._short_range_rays(rng, 60000)
tr = pt.Tracer(sc)
o = orc.Oracle(pt, sc, orc.MATH_DET)
g = tr.intersect(ro, rd)
h = o.intersect(ro, rd)
lin = o.intersect(ro, rd, linear=True)
bad = np.flatnonzero((g["triangle"] != h["triangle"]) | (g["t"].view(np.uint32) != h["t"].view(np.uint32)) | (g["geom"] != h["geom"]))
print("differing rays:", len(bad), "of", len(ro), "| oracle bvh != brute force on", int((lin["triangle"] != h["triangle"]).sum()))
for i in bad[:40]:
    print(i, "o", ro[i], "d", rd[i], "| gpu", g["geom"][i], g["triangle"][i], repr(float(g["t"][i])), "| oracle", h["geom"][i], h["triangle"][i], repr(float(h["t"][i])),
          "| brute", lin["triangle"][i], repr(float(lin["t"][i])))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "short_range_diff.npz"), idx=bad, ro=ro[bad], rd=rd[bad], gpu=g[bad], oracle=h[bad], brute=lin[bad])
