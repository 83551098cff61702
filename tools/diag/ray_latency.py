"""Latency of small ray sets through the closest-hit kernel (how long the longest ray of a launch keeps it alive):
python tools/diag/ray_latency.py [workload]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
import __graft_entry__ as entry

pt = entry._load_package()
scenes = importlib.import_module("cuda_path_tracer_amd.scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
scene_name, res, depth, _ = bench.WORKLOADS[name]
sc = bench.make_workload_scene(scenes, pt, scene_name, res, depth, 0)
tr = pt.Tracer(sc)
o, d = tr.camera_rays(1)
rng = np.random.default_rng(5)
# secondary-like rays: from first hits into random directions
h = tr.intersect(o, d)
ok = np.nonzero((h["t"] > 0) & (h["triangle"] >= 0))[0]
pick = rng.choice(ok, min(200000, len(ok)), replace=False)
p = o[pick] + d[pick] * h["t"][pick, None] * 0.999
v = rng.normal(size=(len(pick), 3)).astype(np.float32)
v /= np.linalg.norm(v, axis=1, keepdims=True)
tr.set_option(pt.PT_OPT_COUNTERS, 1)
for n in (1, 5, 64, 1300, 10000, 100000):
    oo, dd = np.ascontiguousarray(p[:n]), np.ascontiguousarray(v[:n])
    tr.reset_stats()
    tr.intersect(oo, dd)
    st = tr.stats()
    print("n=%6d nodes/ray %.1f" % (n, st.nodesVisited / n), flush=True)
tr.set_option(pt.PT_OPT_COUNTERS, 0)
for n in (1, 5, 64, 1300, 10000, 100000):
    oo, dd = np.ascontiguousarray(p[:n]), np.ascontiguousarray(v[:n])
    out = None
    for _ in range(3):
        tr.intersect_noalloc(oo, dd)
    t0 = time.perf_counter()
    for _ in range(50):
        tr.intersect_noalloc(oo, dd)
    t1 = time.perf_counter()
    print("n=%6d %.1f us per call (with host copies)" % (n, (t1 - t0) / 50 * 1e6), flush=True)
