"""Output side of a render (role of saveImage / Image::savePNG / Image::saveHDR, src/main.cpp:351-375,
src/image.cpp:23-50): 8-bit conversion, PNG and Radiance writers.  Files are read back with independent code
(zlib / struct here) as well as with the library's own readers."""
import json
import struct
import zlib

import ctypes as C
import numpy as np
import pytest


def _read_png_independent(data):
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xFFFFFFFF
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    w, h, depth, ctype, _, _, interlace = hdr
    comps = {0: 1, 2: 3, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * comps)
    assert depth == 8 and interlace == 0 and (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, comps)


@pytest.mark.parametrize("shape", [(1, 1, 3), (7, 5, 3), (64, 48, 3), (33, 17, 4), (20, 31, 1), (300, 200, 3)])
def test_png_writer_round_trip(pt, tmp_path, shape):
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256, size=shape, dtype=np.uint8)
    img[: shape[0] // 2] = img[0, 0]                       # long runs: matches at the maximum length and distance 1
    if shape[0] > 40:
        img[40:] = img[8:8 + shape[0] - 40]               # repeats far back
    p = tmp_path / "a.png"
    pt.write_png(p, img[:, :, 0] if shape[2] == 1 else img)
    data = p.read_bytes()
    assert np.array_equal(_read_png_independent(data), img)
    own = pt.decode_png(data)
    if shape[2] == 4:
        assert np.array_equal(own, img)
    else:
        assert np.array_equal(own[:, :, :3], np.broadcast_to(img[:, :, :3] if shape[2] == 3 else img, (shape[0], shape[1], 3)))
    if shape[0] >= 64:
        assert len(data) < img.nbytes                       # the matcher does compress


def test_rgb8_conversion_is_the_reference_tone_path(pt):
    rng = np.random.default_rng(3)
    s = (rng.random((9, 13, 3), dtype=np.float32) * 40 - 2).astype(np.float32)
    s[0, 0] = (0.0, 7.0, 7.0000005)
    n = np.float32(7.0)
    avg = s / n
    want = (np.minimum(np.maximum(avg, np.float32(0)), np.float32(1)) * np.float32(255)).astype(np.uint8)[:, ::-1]
    assert np.array_equal(pt.image_to_rgb8(s, 7), want)
    assert np.array_equal(pt.image_average(s, 7), avg[:, ::-1])
    assert np.array_equal(pt.image_average(s, 7, mirror_x=False), avg)


def _rgbe_expected(rgb):
    rgb = rgb.astype(np.float32)
    m = rgb.max(axis=2)
    mant, exp = np.frexp(m.astype(np.float64))
    norm = (mant.astype(np.float32) * np.float32(256.0) / np.where(m > 0, m, 1)).astype(np.float32)
    q = (rgb * norm[..., None]).astype(np.float32)
    q8 = np.where(m[..., None] < 1e-32, 0, q.astype(np.uint8))
    e8 = np.where(m < 1e-32, 0, (exp + 128).astype(np.uint8))
    f = np.ldexp(np.float32(1.0), e8.astype(np.int32) - 136).astype(np.float32)
    return np.where(e8[..., None] != 0, q8.astype(np.float32) * f[..., None], 0).astype(np.float32)


@pytest.mark.parametrize("shape", [(5, 4), (6, 40), (3, 300)])
def test_hdr_writer_round_trip_through_the_env_map_loader(pt, scenes, tmp_path, shape):
    """flat scanlines below 8 pixels, run-length ones above; pixels come back as stb's float -> RGBE -> float gives them."""
    rng = np.random.default_rng(5)
    h, w = shape
    img = (rng.random((h, w, 3)) ** 4 * 50).astype(np.float32)
    img[0, : w // 2] = (0.25, 0.5, 1.0)        # runs
    img[1] = 0.0                               # black row: all four planes one run
    img[2, 0] = (1e-35, 0, 0)
    pt.write_hdr(tmp_path / "e.hdr", img)
    d = dict(Materials={"a": dict(TYPE="Diffuse", RGB=[1, 1, 1])}, Camera=scenes._camera("x", 2, 1, res=(4, 4)), Objects=[],
             EnvironmentMap=dict(FILE="e.hdr", INTENSITY=1.0))
    (tmp_path / "s.json").write_text(json.dumps(d))
    sc = pt.Scene(tmp_path / "s.json")
    env = sc.desc.contents.env
    assert (env.width, env.height, env.enabled) == (w, h, 1)
    back = np.frombuffer(C.string_at(env.data, w * h * 12), np.float32).reshape(h, w, 3)
    assert np.array_equal(back, _rgbe_expected(img))
    text = (tmp_path / "e.hdr").read_bytes()
    assert text.startswith(b"#?RADIANCE\n") and b"FORMAT=32-bit_rle_rgbe\n" in text and (b"-Y %d +X %d\n" % (h, w)) in text
    if w >= 8:
        assert len(text) < 200 + w * h * 4


def test_writers_report_errors(pt, tmp_path):
    with pytest.raises(pt.PtError, match="cannot write"):
        pt.write_png(tmp_path / "no" / "dir.png", np.zeros((2, 2, 3), np.uint8))
    with pytest.raises(pt.PtError):
        pt.write_png(tmp_path / "x.png", np.zeros((2, 2, 2), np.uint8))
    with pytest.raises(pt.PtError, match="cannot write"):
        pt.write_hdr(tmp_path / "no" / "dir.hdr", np.zeros((2, 2, 3), np.float32))


def test_eight_bit_environment_maps_and_missing_files(pt, scenes, tmp_path):
    """A PNG or JPEG environment map becomes floats the way stbi_loadf makes them (src/scene.cpp:790): three colour
    components, pow(v / 255, 2.2) evaluated in double, then INTENSITY; a file that cannot be read leaves the scene
    without an environment map and a line in the log."""
    import glb_util, jpeg_util
    rng = np.random.default_rng(9)
    rgba = rng.integers(0, 256, (5, 7, 4), dtype=np.uint8)
    (tmp_path / "e.png").write_bytes(glb_util.write_png(rgba, 6))
    grey = rng.integers(0, 256, (16, 24), dtype=np.uint8)
    jpg, _ = jpeg_util.encode(grey)
    (tmp_path / "e.jpg").write_bytes(jpg)
    cam = scenes._camera("x", 2, 1, res=(4, 4))
    mats = {"a": dict(TYPE="Diffuse", RGB=[1, 1, 1])}

    def load(name, intensity):
        d = dict(Materials=mats, Camera=cam, Objects=[], EnvironmentMap=dict(FILE=name, INTENSITY=intensity))
        (tmp_path / "s.json").write_text(json.dumps(d))
        return pt.Scene(tmp_path / "s.json")

    def floats(v8, intensity):
        f = np.power((v8.astype(np.float32) / np.float32(255)).astype(np.float64), np.float64(np.float32(2.2))).astype(np.float32)
        return f * np.float32(intensity)

    sc = load("e.png", 0.5)
    env = sc.desc.contents.env
    assert (env.width, env.height, env.enabled) == (7, 5, 1)
    got = np.frombuffer(C.string_at(env.data, 7 * 5 * 12), np.float32).reshape(5, 7, 3)
    assert np.array_equal(got, floats(rgba[..., :3], 0.5))
    assert sc.log == ""

    sc = load("e.jpg", 1.0)
    env = sc.desc.contents.env
    assert (env.width, env.height, env.enabled) == (24, 16, 1)
    got = np.frombuffer(C.string_at(env.data, 24 * 16 * 12), np.float32).reshape(16, 24, 3)
    px = pt.decode_jpeg(jpg)
    assert np.array_equal(got, floats(px[..., :3], 1.0))
    assert np.array_equal(got[..., 0], got[..., 1]) and np.array_equal(got[..., 0], got[..., 2])

    for name in ("nothing.hdr", "s.json"):
        sc = load(name, 1.0)
        assert sc.desc.contents.env.enabled == 0
        assert "environment map not loaded" in sc.log and name in sc.log
