#!/usr/bin/env python3
"""How much of the shade stage's time is environment-map traffic: the same workload with environment maps of
different sizes (the picture differs, the path structure barely does).

    python tools/diag/env_size_probe.py [--workload dragon]
"""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="dragon")
    ap.add_argument("--batch", type=int, default=128)
    a = ap.parse_args()
    import bench
    import __graft_entry__ as entry
    pt = entry._load_package()
    scenes = importlib.import_module("cuda_path_tracer_amd.scenes")
    name, res, depth, _ = bench.WORKLOADS[a.workload]
    for env in ((2048, 1024), (512, 256), (64, 32)):
        sc = scenes.make_scene(name, res=res, depth=depth, env_size=env)
        tr = pt.Tracer(sc, device=0)
        tr.set_option(pt.PT_OPT_BATCH, a.batch)
        tr.trace_range(1, 1 + 6 * a.batch)
        tr.sync()
        tr.clear()
        tr.set_option(pt.PT_OPT_STAGE_TIMING, 1)
        tr.reset_stats()
        tr.trace_range(1, 1 + 2 * a.batch)
        tr.sync()
        st = tr.stats()
        n = 2 * a.batch
        print("env %4dx%-4d  path-bounces/spp %.0f  ms/spp: shade %.4f  prims %.4f  trace %.4f  shadow %.4f  camera %.4f" % (
            env[0], env[1], st.pathBounces / n, (st.msShade - st.msShadowTrace) / n, (st.msIntersect - st.msTrace) / n,
            st.msTrace / n, st.msShadowTrace / n, st.msCamera / n), flush=True)
        tr.close()
        sc.close()


if __name__ == "__main__":
    main()
