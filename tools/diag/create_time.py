"""Seconds to build the device scene of a workload (host BVH, walked tree, uploads): python tools/diag/create_time.py [workload]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["PT_DEBUG_TAIL"] = "1"
import bench
import __graft_entry__ as entry

pt = entry._load_package()
scenes = importlib.import_module("cuda_path_tracer_amd.scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
scene_name, res, depth, _ = bench.WORKLOADS[name]
t0 = time.perf_counter()
sc = bench.make_workload_scene(scenes, pt, scene_name, res, depth, 0)
t1 = time.perf_counter()
tr = pt.Tracer(sc)
t2 = time.perf_counter()
print("%s: scene description %.1f s, pt_create %.1f s" % (name, t1 - t0, t2 - t1))
