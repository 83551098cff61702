#!/usr/bin/env python3
"""RMSE / PSNR / largest difference between two renders (.pfm, .ppm, .png).

    python tools/imgdiff.py a.png b.png [--tolerance RMSE]

8-bit files are compared on the 0..255 scale, float files on their own scale (relative RMSE is printed as
well).  With --tolerance the exit status is 1 when the RMSE exceeds it.  PNG files are read with the
library's own decoder (no GPU needed)."""
import argparse
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _package():
    path = os.path.join(ROOT, "cuda-path-tracer_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("cuda_path_tracer_amd", path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["cuda_path_tracer_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def read_image(path):
    data = open(path, "rb").read()
    if data[:8] == b"\x89PNG\r\n\x1a\n":
        return _package().decode_png(data)[:, :, :3].astype(np.float64), 255.0
    if data[:2] in (b"PF", b"P6"):
        parts, pos = [], 0
        while len(parts) < (4 if data[:2] == b"P6" else 4):
            end = pos
            while data[end:end + 1] not in (b" ", b"\n", b"\t", b"\r"):
                end += 1
            parts.append(data[pos:end])
            pos = end + 1
        w, h = int(parts[1]), int(parts[2])
        if data[:2] == b"P6":
            return np.frombuffer(data[pos:pos + w * h * 3], np.uint8).reshape(h, w, 3).astype(np.float64), 255.0
        order = "<f4" if float(parts[3]) < 0 else ">f4"
        return np.frombuffer(data[pos:pos + w * h * 12], order).reshape(h, w, 3)[::-1].astype(np.float64), None
    raise SystemExit("%s: unknown image format" % path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("a")
    ap.add_argument("b")
    ap.add_argument("--tolerance", type=float)
    a = ap.parse_args()
    x, peak_x = read_image(a.a)
    y, peak_y = read_image(a.b)
    if x.shape != y.shape:
        raise SystemExit("sizes differ: %s vs %s" % (x.shape, y.shape))
    d = x - y
    rmse = float(np.sqrt(np.mean(d * d)))
    peak = peak_x or peak_y or float(max(x.max(), y.max(), 1e-30))
    psnr = float("inf") if rmse == 0 else 20 * np.log10(peak / rmse)
    mean = float(0.5 * (x.mean() + y.mean()))
    print("size %dx%d  rmse %.6g  relative rmse %.6g  psnr %.2f dB  max |diff| %.6g  differing samples %d"
          % (x.shape[1], x.shape[0], rmse, rmse / mean if mean else 0.0, psnr, float(np.abs(d).max()), int((d != 0).sum())))
    if a.tolerance is not None and rmse > a.tolerance:
        sys.exit(1)


if __name__ == "__main__":
    main()
