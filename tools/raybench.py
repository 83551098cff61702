#!/usr/bin/env python3
"""Ray-set microbenchmark of the closest-hit stage: fixed sets of rays against a workload's scene through the
fixed-ray-set hook (pt_intersect_rays), which runs the production cull / traversal / primitive kernels.

    python tools/raybench.py [--workload dragon|chess|halo] [--rays 16777216]

Three sets per scene:
  primary   camera rays of consecutive iterations (coherent);
  sphere    origins uniform on a sphere around the geometry the primary rays hit, directions cosine-distributed
            about the inward normal (incoherent, every ray aimed at the scene; numpy generator, seed 12345);
  bounce3   rays leaving the third hit point of a chain of cosine-weighted diffuse bounces that starts with the
            primary set: each stage draws its rays (with replacement, so the set keeps its size) from the hits of
            the stage before, origin at the hit point offset along the normal.
The hook moves one image's worth of rays per launch (H2D, kernels, D2H), so the scene is set up with a 4096 x 4096
image: one launch per set of 2^24 rays.  The rates below use the kernels' own device time (HIP events around the stages, PT_OPT_STAGE_TIMING), a counted pass gives boxes / triangles per ray.
Prints one JSON line per set.
"""
import argparse
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cosine_about(normals, rng):
    n = normals / np.maximum(np.linalg.norm(normals, axis=1, keepdims=True), 1e-20)
    u1, u2 = rng.random(len(n), dtype=np.float32), rng.random(len(n), dtype=np.float32)
    r, phi = np.sqrt(u1), 2 * np.pi * u2
    a = np.where(np.abs(n[:, :1]) < 0.577, np.array([[1.0, 0, 0]], np.float32), np.array([[0, 1.0, 0]], np.float32))
    t = np.cross(n, a)
    t /= np.maximum(np.linalg.norm(t, axis=1, keepdims=True), 1e-20)
    b = np.cross(n, t)
    d = t * (r * np.cos(phi))[:, None] + b * (r * np.sin(phi))[:, None] + n * np.sqrt(np.maximum(0.0, 1.0 - u1))[:, None]
    return d.astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="dragon")
    ap.add_argument("--rays", type=int, default=1 << 24)
    args = ap.parse_args()
    import bench
    import __graft_entry__ as entry
    pt = entry._load_package()
    scenes = importlib.import_module("cuda_path_tracer_amd.scenes")
    scene_name, res, depth, desc = bench.WORKLOADS[args.workload]
    res = (4096, 4096)
    sc = scenes.make_scene(scene_name, res=res, depth=depth)
    tr = pt.Tracer(sc, device=0)
    npix = res[0] * res[1]
    iters = max(1, -(-args.rays // npix))
    rng = np.random.default_rng(12345)

    po, pd = zip(*[tr.camera_rays(1 + i) for i in range(iters)])
    po, pd = np.concatenate(po)[:args.rays], np.concatenate(pd)[:args.rays]
    n = len(po)

    def run(name, o, d):
        tr.intersect(o[:npix], d[:npix])                                   # warm the caches and the allocator
        tr.set_option(pt.PT_OPT_STAGE_TIMING, 1)
        tr.reset_stats()
        hits = tr.intersect(o, d)
        st = tr.stats()
        tr.set_option(pt.PT_OPT_STAGE_TIMING, 0)
        tr.set_option(pt.PT_OPT_COUNTERS, 1)
        tr.reset_stats()
        m = min(len(o), 4 * npix)
        tr.intersect(o[:m], d[:m])
        cs = tr.stats()
        tr.set_option(pt.PT_OPT_COUNTERS, 0)
        ms_trace, ms_rest = st.msTrace, st.msIntersect
        print(json.dumps(dict(workload=args.workload, set=name, rays=len(o), launches=st.launchesTrace,
                              hit_rate=round(float((hits["t"] > 0).mean()), 4),
                              mesh_hit_rate=round(float(((hits["t"] > 0) & (hits["triangle"] >= 0)).mean()), 4),
                              ms_traversal=round(ms_trace, 3), ms_cull_and_primitives=round(ms_rest, 3),
                              mrays_per_s_traversal=round(len(o) / ms_trace / 1e3, 1) if ms_trace > 0 else None,
                              mrays_per_s_stage=round(len(o) / (ms_trace + ms_rest) / 1e3, 1),
                              boxes_per_ray=round(cs.boxesTested / m, 2), tris_per_ray=round(cs.trisTested / m, 2))), flush=True)
        return hits

    h = run("primary", po, pd)

    hit = h["t"] > 0
    pts = po[hit] + pd[hit] * h["t"][hit, None]
    centre = pts.mean(axis=0)
    radius = 1.2 * float(np.sqrt(((pts - centre) ** 2).sum(axis=1)).max())
    v = rng.normal(size=(n, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    so = (centre + radius * v).astype(np.float32)
    sd = cosine_about(-v, rng)
    run("sphere", so, sd)

    o, d, hh = po, pd, h
    for bounce in range(3):
        idx = np.flatnonzero(hh["t"] > 0)
        sel = idx[rng.integers(0, len(idx), n)]             # with replacement: the set keeps its size
        nrm = hh["normal"][sel]
        o, d = (o[sel] + d[sel] * hh["t"][sel, None] + nrm * 1e-3).astype(np.float32), cosine_about(nrm, rng)
        if bounce < 2:
            hh = tr.intersect(o, d)
    run("bounce3", o, d)


if __name__ == "__main__":
    main()
