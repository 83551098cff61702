"""Critical path of one iteration launched alone: longest wave (loop iterations) and shared subtrees per stage.
python tools/diag/tail_diag.py [workload]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["PT_DEBUG_TAIL"] = "1"
os.environ.setdefault("PT_BATCH", "1")
import bench
import __graft_entry__ as entry

pt = entry._load_package()
scenes = importlib.import_module("cuda_path_tracer_amd.scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
scene_name, res, depth, _ = bench.WORKLOADS[name]
for d in range(1, depth + 1):
    sc = bench.make_workload_scene(scenes, pt, scene_name, res, d, 0)
    tr = pt.Tracer(sc)
    tr.set_option(pt.PT_OPT_COUNTERS, 1)
    tr.trace(6)
    tr.sync()
    tr.reset_stats()
    tr.trace(7)
    tr.sync()
    print("depth", d, file=sys.stderr, flush=True)
    tr.stats()
    tr.close()
