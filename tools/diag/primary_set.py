"""The camera rays of a workload as one closest-hit launch, a few times over (for counter passes): python tools/diag/primary_set.py [workload] [times]"""
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
times = int(sys.argv[2]) if len(sys.argv) > 2 else 6
scene_name, res, depth, _ = bench.WORKLOADS[name]
sc = bench.make_workload_scene(scenes, pt, scene_name, res, depth, 0)
tr = pt.Tracer(sc)
o, d = tr.camera_rays(1)
for _ in range(times):
    tr.intersect(o, d)
