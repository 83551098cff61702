"""Traversal rate of one launch over a fixed ray set (camera rays, rays after three diffuse bounces):
python tools/diag/lone_launch.py [workload]   (environment: PT_PERSIST_BLOCKS etc.)"""
import importlib
import os
import sys

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
print(name, bench._ray_sets(pt, tr, res[0] * res[1]), flush=True)
