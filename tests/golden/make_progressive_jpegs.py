#!/usr/bin/env python3
"""Writes tests/golden/progressive/*.jpg and expected.json: small progressive (SOF2) JPEG files and the SHA-256 of
the RGBA pixels stb_image makes of them.

    python tests/golden/make_progressive_jpegs.py      (needs Pillow for the files and oracle/_ref for stb_image)

The files come from Pillow's libjpeg encoder (spectral selection + successive approximation scans, optimised or
default Huffman tables, restart markers); the expected pixels come from the reference's own vendored stb_image,
built from its tree as oracle/_ref/libref.so (tests/ref_binding.py) — the decoder tinygltf hands every texture to.
"""
import hashlib
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
OUT = os.path.join(HERE, "progressive")

CASES = [  # name, mode, (w, h), subsampling, quality, extra
    ("rgb444_64x48_q75", "RGB", (64, 48), 0, 75, {}),
    ("rgb422_37x29_q30", "RGB", (37, 29), 1, 30, {"optimize": True}),
    ("rgb420_130x67_q95", "RGB", (130, 67), 2, 95, {}),
    ("rgb420_8x9_q75_restart", "RGB", (8, 9), 2, 75, {"restart_marker_blocks": 1}),
    ("rgb420_1x1_q75", "RGB", (1, 1), 2, 75, {}),
    ("grey_37x29_q75_restart", "L", (37, 29), 0, 75, {"restart_marker_blocks": 3}),
    ("cmyk_64x48_q75", "CMYK", (64, 48), 0, 75, {}),
]


def image(w, h, mode, rng):
    from PIL import Image
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([(x * 255 // max(w - 1, 1)), (y * 255 // max(h - 1, 1)), ((x + y) * 5) % 256], -1).astype(np.float64)
    base += rng.normal(0, 25, (h, w, 3))
    im = Image.fromarray(np.clip(base, 0, 255).astype(np.uint8), "RGB")
    return im if mode == "RGB" else im.convert(mode)


def main():
    import ref_binding as ref
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.RandomState(3)
    expected = {}
    for name, mode, (w, h), sub, q, extra in CASES:
        kw = dict(quality=q, progressive=True, **extra)
        if mode == "RGB":
            kw["subsampling"] = sub
        buf = io.BytesIO()
        image(w, h, mode, rng).save(buf, "JPEG", **kw)
        data = buf.getvalue()
        assert b"\xff\xc2" in data
        path = os.path.join(OUT, name + ".jpg")
        open(path, "wb").write(data)
        kind, comps, rgba, _f = ref.stbi(path)
        assert kind == "ldr"
        expected[name] = dict(width=w, height=h, bytes=len(data), rgba_sha256=hashlib.sha256(np.asarray(rgba).tobytes()).hexdigest())
        print(name, len(data), "bytes")
    json.dump(expected, open(os.path.join(OUT, "expected.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
