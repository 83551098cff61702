"""Find the most expensive rays of a third-bounce ray set (bisection on counted node visits) and describe them."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import __graft_entry__ as entry

pt = entry._load_package()
scenes = importlib.import_module("cuda_path_tracer_amd.scenes")
from test_gpu_full_size import _bounce3_rays

name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
scene_name, res, depth, _ = bench.WORKLOADS[name]
sc = bench.make_workload_scene(scenes, pt, scene_name, (960, 540), depth, 0)
tr = pt.Tracer(sc)
o, d = _bounce3_rays(tr, 200000, 3)
tr.set_option(pt.PT_OPT_COUNTERS, 1)


def cost(idx):
    tr.reset_stats()
    tr.intersect(o[idx], d[idx])
    st = tr.stats()
    return st.nodesVisited


allc = cost(np.arange(len(o)))
print("mean nodes per ray %.1f" % (allc / len(o)))
chunks = np.array_split(np.arange(len(o)), 400)
cc = np.array([cost(c) for c in chunks])
print("chunk totals: median %.0f max %.0f (of %d rays each)" % (np.median(cc), cc.max(), len(chunks[0])))
top = chunks[int(np.argmax(cc))]
per = np.array([cost(np.array([i])) for i in top])
order = np.argsort(per)[::-1]
tris = sc.mesh_triangles(0)
for k in order[:6]:
    i = top[k]
    h = tr.intersect(o[i:i + 1], d[i:i + 1])[0]
    print("ray %d: %d nodes; origin %s dir %s -> t %.4f tri %d" % (i, per[k], o[i], d[i], h["t"], h["triangle"]))
print("histogram of per-ray nodes in that chunk:", np.percentile(per, [50, 90, 99, 100]))
