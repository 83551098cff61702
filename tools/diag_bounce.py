#!/usr/bin/env python3
"""Per-bounce cost of the closest-hit traversal: renders the workload at trace depth 1..D and differences the
traversal time, ray count and box / triangle tests of consecutive depths (paths differ between depths, the
statistics do not)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pt = entry._load_package()
scenes = importlib.import_module("cuda_path_tracer_amd.scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
D = 8
prev = None
print("%5s %12s %10s %10s %10s %12s %12s" % ("bounce", "rays/spp", "ms/spp", "ns/ray", "boxes/ray", "tris/ray", "ns/box"))
for d in range(1, D + 1):
    sc = scenes.make_scene(name, res=(1920, 1080), depth=d)
    tr = pt.Tracer(sc, device=0)
    tr.set_option(pt.PT_OPT_BATCH, B)
    tr.trace_range(1, 1 + B, 1)            # warm-up
    tr.sync()
    tr.reset_stats()
    tr.set_option(pt.PT_OPT_STAGE_TIMING, 1)
    tr.trace_range(1 + B, 1 + 2 * B, 1)
    tr.sync()
    s = tr.stats()
    ms, rays = s.msTrace, s.tracedRays
    tr.set_option(pt.PT_OPT_STAGE_TIMING, 0)
    tr.reset_stats()
    tr.set_option(pt.PT_OPT_COUNTERS, 1)
    tr.trace_range(1 + B, 1 + 2 * B, 1)
    tr.sync()
    s = tr.stats()
    boxes, tris = s.boxesTested - s.shadeBoxesTested, s.trisTested - s.shadeTrisTested
    cur = (ms, rays, boxes, tris)
    dm, dr, db, dt = (cur if prev is None else tuple(c - p for c, p in zip(cur, prev)))
    prev = cur
    print("%5d %12.0f %10.4f %10.2f %10.2f %12.2f %12.3f" % (d - 1, dr / B, dm / B, 1e6 * dm / max(dr, 1), db / max(dr, 1), dt / max(dr, 1), 1e6 * dm / max(db, 1)))
    del tr
