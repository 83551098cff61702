"""Writers for test inputs of the glTF import: PNG streams (every colour type, filter, interlace, deflate block
kind) and GLB containers.  Independent of the library under test: zlib / struct / numpy only."""
import json
import struct
import zlib

import numpy as np

CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


def _chunk(tag, body):
    return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def _filter_rows(rows, bpp, filt):
    """rows: list of bytes (packed scanlines).  filt: 0..4 or 'cycle'."""
    out = bytearray()
    prev = bytes(len(rows[0])) if rows else b""
    for y, row in enumerate(rows):
        f = (y % 5) if filt == "cycle" else filt
        cur = bytearray(len(row))
        for i, v in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = (0, a, b, (a + b) >> 1, _paeth(a, b, c))[f]
            cur[i] = (v - pred) & 255
        out.append(f)
        out += cur
        prev = row
    return bytes(out)


def _pack(samples, depth):
    """samples (H, W, C) uint8 values < 2**depth -> list of packed scanlines."""
    h, w, c = samples.shape
    if depth == 8:
        return [samples[y].tobytes() for y in range(h)]
    rows = []
    per = 8 // depth
    for y in range(h):
        flat = samples[y].reshape(-1)
        pad = (-len(flat)) % per
        flat = np.concatenate([flat, np.zeros(pad, np.uint8)]).reshape(-1, per)
        acc = np.zeros(len(flat), np.uint32)
        for k in range(per):
            acc = (acc << depth) | flat[:, k]
        rows.append(acc.astype(np.uint8).tobytes())
    return rows


def write_png(samples, ctype, depth=8, palette=None, trns=None, filt="cycle", interlace=False, level=6, strategy=zlib.Z_DEFAULT_STRATEGY,
              idat_split=0):
    """samples: (H, W, C) uint8 raw sample values (indices for ctype 3)."""
    samples = np.ascontiguousarray(samples, np.uint8)
    if samples.ndim == 2:
        samples = samples[:, :, None]
    h, w, c = samples.shape
    assert c == CHANNELS[ctype]
    bpp = max(1, c * depth // 8)
    if not interlace:
        raw = _filter_rows(_pack(samples, depth), bpp, filt)
    else:
        raw = b""
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = samples[y0::dy, x0::dx]
            if sub.shape[0] == 0 or sub.shape[1] == 0:
                continue
            raw += _filter_rows(_pack(np.ascontiguousarray(sub), depth), bpp, filt)
    co = zlib.compressobj(level, zlib.DEFLATED, 15, 8, strategy)
    z = co.compress(raw) + co.flush()
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if palette is not None:
        out += _chunk(b"PLTE", np.ascontiguousarray(palette, np.uint8).tobytes())
    if trns is not None:
        out += _chunk(b"tRNS", bytes(trns))
    if idat_split:
        for i in range(0, len(z), idat_split):
            out += _chunk(b"IDAT", z[i:i + idat_split])
    else:
        out += _chunk(b"IDAT", z)
    return out + _chunk(b"IEND", b"")


def expected_rgba(samples, ctype, depth=8, palette=None, trns=None):
    """What stb_image returns for 4 requested components."""
    samples = np.ascontiguousarray(samples, np.uint8)
    if samples.ndim == 2:
        samples = samples[:, :, None]
    h, w, _ = samples.shape
    out = np.zeros((h, w, 4), np.uint8)
    if ctype == 0:
        scale = {1: 255, 2: 85, 4: 17, 8: 1}[depth]
        g = (samples[:, :, 0].astype(np.uint32) * scale).astype(np.uint8)
        out[:, :, 0] = out[:, :, 1] = out[:, :, 2] = g
        out[:, :, 3] = 255
        if trns is not None:
            key = (trns[1] * scale) & 255
            out[:, :, 3][g == key] = 0
    elif ctype == 2:
        out[:, :, :3] = samples
        out[:, :, 3] = 255
        if trns is not None:
            key = np.array([trns[1], trns[3], trns[5]], np.uint8)
            out[:, :, 3][(samples == key).all(axis=2)] = 0
    elif ctype == 3:
        pal = np.zeros((256, 4), np.uint8)
        pal[:, 3] = 255
        p = np.asarray(palette, np.uint8).reshape(-1, 3)
        pal[:len(p), :3] = p
        if trns is not None:
            pal[:len(trns), 3] = np.frombuffer(bytes(trns), np.uint8)
        out[:] = pal[samples[:, :, 0]]
    elif ctype == 4:
        out[:, :, 0] = out[:, :, 1] = out[:, :, 2] = samples[:, :, 0]
        out[:, :, 3] = samples[:, :, 1]
    else:
        out[:] = samples
    return out


class GlbBuilder:
    """Assembles a glTF 2.0 document with one binary buffer; write_glb() / write_gltf() serialise it."""

    def __init__(self):
        self.doc = dict(asset=dict(version="2.0"), scenes=[dict(nodes=[])], scene=0, nodes=[], meshes=[], accessors=[], bufferViews=[],
                        buffers=[], materials=[], textures=[], images=[])
        self.bin = bytearray()

    def _view(self, data, stride=None):
        while len(self.bin) % 4:
            self.bin.append(0)
        bv = dict(buffer=0, byteOffset=len(self.bin), byteLength=len(data))
        if stride:
            bv["byteStride"] = stride
        self.bin += data
        self.doc["bufferViews"].append(bv)
        return len(self.doc["bufferViews"]) - 1

    def accessor(self, array, kind, ctype):
        array = np.ascontiguousarray(array)
        bv = self._view(array.tobytes())
        self.doc["accessors"].append(dict(bufferView=bv, byteOffset=0, componentType=ctype, count=int(array.shape[0]), type=kind))
        return len(self.doc["accessors"]) - 1

    def interleaved(self, pos, nrm, uv):
        """POSITION / NORMAL / TEXCOORD_0 in one strided buffer view (stride 32)."""
        n = len(pos)
        rec = np.zeros((n, 8), np.float32)
        rec[:, 0:3], rec[:, 3:6], rec[:, 6:8] = pos, nrm, uv
        bv = self._view(rec.tobytes(), stride=32)
        ids = []
        for off, kind in ((0, "VEC3"), (12, "VEC3"), (24, "VEC2")):
            self.doc["accessors"].append(dict(bufferView=bv, byteOffset=off, componentType=5126, count=n, type=kind))
            ids.append(len(self.doc["accessors"]) - 1)
        return ids

    def image_png(self, png_bytes):
        bv = self._view(png_bytes)
        self.doc["images"].append(dict(bufferView=bv, mimeType="image/png"))
        self.doc["textures"].append(dict(source=len(self.doc["images"]) - 1))
        return len(self.doc["textures"]) - 1

    def material(self, **kw):
        self.doc["materials"].append(kw)
        return len(self.doc["materials"]) - 1

    def mesh(self, primitives):
        self.doc["meshes"].append(dict(primitives=primitives))
        return len(self.doc["meshes"]) - 1

    def node(self, root=False, **kw):
        self.doc["nodes"].append(kw)
        i = len(self.doc["nodes"]) - 1
        if root:
            self.doc["scenes"][0]["nodes"].append(i)
        return i

    def _finish(self):
        doc = {k: v for k, v in self.doc.items() if v != []}
        doc["buffers"] = [dict(byteLength=len(self.bin))]
        return doc

    def write_glb(self, path):
        doc = self._finish()
        js = json.dumps(doc).encode()
        js += b" " * ((-len(js)) % 4)
        bn = bytes(self.bin) + b"\0" * ((-len(self.bin)) % 4)
        total = 12 + 8 + len(js) + 8 + len(bn)
        with open(path, "wb") as f:
            f.write(struct.pack("<III", 0x46546C67, 2, total))
            f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
            f.write(struct.pack("<II", len(bn), 0x004E4942) + bn)

    def write_gltf(self, path, embed=True):
        """Text form: the buffer as a base64 data URI (embed) or as a .bin file next to it."""
        import base64
        import os
        doc = self._finish()
        if embed:
            doc["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(bytes(self.bin)).decode()
        else:
            name = os.path.basename(str(path)) + ".bin"
            with open(os.path.join(os.path.dirname(str(path)), name), "wb") as f:
                f.write(bytes(self.bin))
            doc["buffers"][0]["uri"] = name
        with open(path, "w") as f:
            json.dump(doc, f)
