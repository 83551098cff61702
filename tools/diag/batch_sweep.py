"""ms per iteration against the number of iterations that share a launch group: python tools/diag/batch_sweep.py [workload]"""
import importlib
import os
import sys
import time

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
for b in (1, 2, 3, 4, 6, 8, 16, 32, 64):
    tr.set_option(pt.PT_OPT_BATCH, b)
    n = max(64, 4 * b)
    tr.trace_range(1, 1 + b)
    tr.sync()
    t0 = time.perf_counter()
    tr.trace_range(1 + b, 1 + b + n)
    tr.sync()
    t1 = time.perf_counter()
    print("batch %3d: %.3f ms per iteration" % (b, (t1 - t0) / n * 1e3), flush=True)
