"""Lane bookkeeping of the traversal kernel on a workload (counted kernels): python tools/diag/wave_diag.py [workload] [spp]"""
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
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
scene_name, res, depth, _ = bench.WORKLOADS[name]
sc = bench.make_workload_scene(scenes, pt, scene_name, res, depth, 0)
tr = pt.Tracer(sc)
tr.set_option(pt.PT_OPT_BATCH, spp)
tr.set_option(pt.PT_OPT_COUNTERS, 1)
tr.trace_range(1, 1 + spp)
st = tr.stats()
print("rays %d shadow %d boxes %d tris %d nodes %d deepest stack %d" % (st.tracedRays, st.shadowRays, st.boxesTested, st.trisTested, st.nodesVisited, st.maxStackDepth))
