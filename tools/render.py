#!/usr/bin/env python3
"""Headless render of a named configuration to PNG (sum / spp, clamp, 8 bit, x-mirrored like saveImage)."""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("--res", default="480x270")
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--spp", type=int, default=32)
    ap.add_argument("--out", default=None)
    ap.add_argument("--grid", default=None)
    args = ap.parse_args()
    import __graft_entry__ as entry
    pt = entry._load_package()
    scenes = importlib.import_module("cuda_path_tracer_amd.scenes")
    w, h = map(int, args.res.split("x"))
    kw = {}
    if args.grid:
        kw["grid"] = tuple(map(int, args.grid.split("x")))
    sc = scenes.make_scene(args.scene, res=(w, h), depth=args.depth, **kw)
    tr = pt.Tracer(sc)
    tr.trace_range(1, args.spp + 1)
    st = tr.stats()
    print("live curve of the last iteration:", list(st.lastLive[:sc.depth + 1]), "path-bounces/spp:", st.pathBounces / args.spp)
    img = tr.image() / args.spp
    out = args.out or os.path.join(ROOT, "gpurun_out", args.scene + ".png")
    from PIL import Image
    rgb = np.clip(img * 255.0, 0, 255).astype(np.uint8)[:, ::-1]      # saveImage mirrors x (src/main.cpp:351-375)
    Image.fromarray(rgb).save(out)
    print("wrote", out)


if __name__ == "__main__":
    main()
