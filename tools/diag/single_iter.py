"""One iteration per call (the reference-shaped use of the loop): python tools/diag/single_iter.py [workload] [iterations]
SYNC_EACH=1 waits for the device after every call."""
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
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 32
scene_name, res, depth, _ = bench.WORKLOADS[name]
sc = bench.make_workload_scene(scenes, pt, scene_name, res, depth, 0)
tr = pt.Tracer(sc)
for it in range(1, 4):
    tr.trace(it)
tr.sync()
tr.reset_stats()
t0 = time.perf_counter()
for it in range(4, 4 + iters):
    tr.trace(it)
    if os.environ.get("SYNC_EACH"):
        tr.sync()
tr.sync()
t1 = time.perf_counter()
st = tr.stats()
print("%s: %.3f ms per iteration, %.1f Mpaths*bounce/s, live %s" % (name, (t1 - t0) / iters * 1e3, st.pathBounces / (t1 - t0) / 1e6, list(st.lastLive[:depth + 1])))
