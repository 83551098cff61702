"""When the waves of one closest-hit launch over all camera rays see the queue dry and when they end.
python tools/diag/tail_hist.py [workload]"""
import importlib
import os
import sys

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
tr.intersect(o, d)
tr.set_option(pt.PT_OPT_COUNTERS, 1)
tr.reset_stats()
tr.intersect(o, d)
tr.stats()
