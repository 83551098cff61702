#!/usr/bin/env python3
"""Writes tests/golden/reference_renders.npz from the rendered images the reference itself holds.

    python tests/golden/make_reference_renders.py        (needs /root/reference)

The reference ships no test vectors for its shading code, but its write-up shows finished renders of scene files it
also ships, and three of those scenes need no asset that is missing from its tree:

    img/sss-1.png        scenes/sss.json                  README.md:248-256 "Subsurface Scattering On"
    img/sss-0.png        scenes/sss.json with every SUBSURFACE_ENABLED set to 0   "Subsurface Scattering Off"
    img/diffuse-orig.png scenes/cornell-orig.json         README.md:77
    img/ssaa-1024.png    scenes/ssaa.json                 README.md:150-158

Each is a 1200 x 1200 8-bit PNG as Image::savePNG wrote it (src/image.cpp:23-37): the outputs of the whole loop —
camera rays, intersection, the diffuse / specular / PBR / subsurface shaders with their light sampling, gather — after
hundreds to thousands of iterations.  They are data.  Stored here are the sums of their 8 x 8 pixel blocks per channel
(uint16, exact), the largest 8-bit value of every block (over its pixels and channels: a block whose maximum is below
250 holds no pixel the 8-bit conversion clamped), the per-channel means and the SHA-256 of each file, which is all the
render tests compare with.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
REF_IMG = "/root/reference/img"
RENDERS = {"sss_1": "sss-1.png", "sss_0": "sss-0.png", "diffuse_orig": "diffuse-orig.png", "ssaa_1024": "ssaa-1024.png"}
BLOCK = 8


def block_sums(rgb8, block=BLOCK):
    h, w, c = rgb8.shape
    return rgb8.astype(np.uint32).reshape(h // block, block, w // block, block, c).sum((1, 3)).astype(np.uint16)


def main():
    import conftest
    pt = conftest.load_package()
    out = {}
    for key, name in RENDERS.items():
        data = open(os.path.join(REF_IMG, name), "rb").read()
        rgba = np.asarray(pt.decode_png(data))
        assert rgba.shape == (1200, 1200, 4) and (rgba[..., 3] == 255).all(), (name, rgba.shape)
        rgb = np.ascontiguousarray(rgba[..., :3])
        out[key + "_sum8"] = block_sums(rgb)
        out[key + "_max8"] = rgb.reshape(1200 // BLOCK, BLOCK, 1200 // BLOCK, BLOCK, 3).max((1, 3, 4)).astype(np.uint8)
        out[key + "_mean"] = rgb.reshape(-1, 3).mean(0)
        out[key + "_sha256"] = np.array(hashlib.sha256(data).hexdigest())
        print(name, rgb.shape, out[key + "_mean"])
    path = os.path.join(HERE, "reference_renders.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
