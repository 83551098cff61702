#!/usr/bin/env python3
"""Look-ahead calls at full size against one launch-group render of the same iterations (GPU).
    python tools/diag/lookahead_full.py [calls] [lookahead]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import conftest
pt = conftest.load_package()
scenes = importlib.import_module("cuda_path_tracer_amd.scenes")
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 257
ahead = int(sys.argv[2]) if len(sys.argv) > 2 else 64
sc = scenes.make_scene("dragon", res=(1920, 1080), depth=8)
tr = pt.Tracer(sc)
tr.set_option(pt.PT_OPT_MATERIAL_SORT, 1)
tr.set_option(pt.PT_OPT_BATCH, 2 * ahead)
tr.set_option(pt.PT_OPT_LAUNCH_STREAMS, 1)
tr.trace_range(1, calls + 1)
want = tr.image().copy()
print("reference sum: mean / calls = %.5f" % (want.mean() / calls), flush=True)
del tr
os.environ["PT_DEBUG_TAIL"] = "1"
for rows in (ahead,):
    tr = pt.Tracer(sc)
    tr.set_option(pt.PT_OPT_MATERIAL_SORT, 1)
    tr.set_option(pt.PT_OPT_BATCH, rows)
    tr.trace_range(1, 6 * rows + 1); tr.sync()       # the library tries its launch shapes on the first five full groups
    t0 = time.time()
    tr.trace_range(1, 2 * rows + 1); tr.sync()
    print("plain range, groups of %d rows, launch shape chosen by the library: %.3f ms per iteration" % (rows, (time.time() - t0) * 1e3 / (2 * rows)), flush=True)
    del tr
del os.environ["PT_DEBUG_TAIL"]
streams_list = [int(x) for x in os.environ.get("DIAG_STREAMS", "1").split(",")]
for label, env, wait, streams in [(l + ", %d stream(s)" % n, e, w, n) for n in streams_list for l, e, w in (("strict", {}, True), ("queued", {}, False), ("no host image", {}, None))]:
    for k, v in env.items():
        os.environ[k] = v
    tr = pt.Tracer(sc)
    tr.set_option(pt.PT_OPT_MATERIAL_SORT, 1)
    for k in env:
        del os.environ[k]
    tr.set_option(pt.PT_OPT_BATCH, 2 * ahead)
    tr.set_option(pt.PT_OPT_LAUNCH_STREAMS, streams)
    host = np.zeros_like(want)
    if wait is not None:
        tr.bind_host_image(host)
    t0 = time.time()
    for it in range(1, calls + 1):
        tr.trace_lookahead(it, ahead, calls)
        if wait:
            tr.host_image_wait()
    tr.host_image_wait()
    tr.sync()
    ms = (time.time() - t0) * 1e3 / calls
    dev = tr.image()
    if wait is None:
        host = dev
    bad = (host.view(np.uint32) != want.view(np.uint32))
    print("%-32s %.3f ms per call; host image == reference: %s (%d values differ, mean / calls %.5f); device sum == reference: %s" % (
        label, ms, not bad.any(), int(bad.sum()), host.mean() / calls, bool((dev.view(np.uint32) == want.view(np.uint32)).all())), flush=True)
    if bad.any():
        ys, xs = np.nonzero(bad.any(-1))
        print("   differing pixels: rows %d..%d, cols %d..%d; first few (y, x, host, want): %s" % (ys.min(), ys.max(), xs.min(), xs.max(),
              [(int(y), int(x), host[y, x].tolist(), want[y, x].tolist()) for y, x in list(zip(ys, xs))[:3]]), flush=True)
    tr.bind_host_image(None)
    del tr
