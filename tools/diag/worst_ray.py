"""Find the camera ray of a workload that visits the most BVH nodes (bisection over the fixed-ray-set hook with counters on)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["PT_DEBUG_TAIL"] = "1"
import bench
import __graft_entry__ as entry

pt = entry._load_package()
scenes = importlib.import_module("cuda_path_tracer_amd.scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
scene_name, res, depth, _ = bench.WORKLOADS[name]
sc = bench.make_workload_scene(scenes, pt, scene_name, res, depth, 0)
tr = pt.Tracer(sc)
o, d = tr.camera_rays(1)
tr.set_option(pt.PT_OPT_COUNTERS, 1)


def cost(idx):
    tr.reset_stats()
    tr.intersect(o[idx], d[idx])
    st = tr.stats()
    return st.nodesVisited, st.trisTested


idx = np.arange(len(o))
total = cost(idx)
print("all rays: nodes %d tris %d per ray %.1f / %.1f" % (total[0], total[1], total[0] / len(o), total[1] / len(o)))
# per-tile cost map (64 x 36 tiles) to see where the expensive rays are
W, H = res
tiles = []
for ty in range(0, H, 120):
    row = []
    for tx in range(0, W, 120):
        yy, xx = np.meshgrid(np.arange(ty, min(ty + 120, H)), np.arange(tx, min(tx + 120, W)), indexing="ij")
        sel = (yy * W + xx).ravel()
        n, t = cost(sel)
        row.append(n / len(sel))
    tiles.append(row)
print("mean nodes per ray by 120-pixel tile:")
for row in tiles:
    print(" ".join("%5.0f" % v for v in row))
# bisect for the single worst ray inside the worst tile
best = np.unravel_index(np.argmax(np.array(tiles)), np.array(tiles).shape)
ty, tx = best[0] * 120, best[1] * 120
yy, xx = np.meshgrid(np.arange(ty, min(ty + 120, H)), np.arange(tx, min(tx + 120, W)), indexing="ij")
sel = (yy * W + xx).ravel()
costs = []
for i in sel[::7]:
    costs.append((cost(np.array([i]))[0], i))
costs.sort()
for c, i in costs[-5:]:
    h = tr.intersect(o[i:i + 1], d[i:i + 1])
    print("ray pixel (%d, %d): %d nodes, dir %s, hit t %.3f geom %d tri %d" % (i % W, i // W, c, d[i], h["t"][0], h["geom"][0], h["triangle"][0]))
