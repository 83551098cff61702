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
sc = scenes.make_scene("dragon")
tr = pt.Tracer(sc)
tr.set_option(pt.PT_OPT_BATCH, 2 * ahead)
tr.set_option(pt.PT_OPT_LAUNCH_STREAMS, 1)
tr.trace_range(1, calls + 1)
want = tr.image().copy()
print("reference sum: mean / calls = %.5f" % (want.mean() / calls), flush=True)
del tr
for label, env, wait in (("strict", {}, True), ("queued", {}, False), ("queued, PT_SPARSE=0", {"PT_SPARSE": "0"}, False), ("queued, PT_SPARSE=1", {"PT_SPARSE": "1"}, False)):
    for k, v in env.items():
        os.environ[k] = v
    tr = pt.Tracer(sc)
    for k in env:
        del os.environ[k]
    tr.set_option(pt.PT_OPT_BATCH, 2 * ahead)
    tr.set_option(pt.PT_OPT_LAUNCH_STREAMS, 1)
    host = np.zeros_like(want)
    tr.bind_host_image(host)
    t0 = time.time()
    for it in range(1, calls + 1):
        tr.trace_lookahead(it, ahead, calls)
        if wait:
            tr.host_image_wait()
    tr.host_image_wait()
    ms = (time.time() - t0) * 1e3 / calls
    dev = tr.image()
    bad = (host.view(np.uint32) != want.view(np.uint32))
    print("%-22s %.3f ms per call; host image == reference: %s (%d values differ, mean / calls %.5f); device sum == reference: %s" % (
        label, ms, not bad.any(), int(bad.sum()), host.mean() / calls, bool((dev.view(np.uint32) == want.view(np.uint32)).all())), flush=True)
    if bad.any():
        ys, xs = np.nonzero(bad.any(-1))
        print("   differing pixels: rows %d..%d, cols %d..%d; first few (y, x, host, want): %s" % (ys.min(), ys.max(), xs.min(), xs.max(),
              [(int(y), int(x), host[y, x].tolist(), want[y, x].tolist()) for y, x in list(zip(ys, xs))[:3]]), flush=True)
    tr.bind_host_image(None)
    del tr
