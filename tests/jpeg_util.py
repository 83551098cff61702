"""A small baseline JPEG writer and an independent float decoder, for testing the library's JPEG texture path.

The writer knows the quantised coefficients it stores, so the expected image does not depend on any decoder: it is
the ideal (float) inverse DCT of those coefficients, rounded to samples, chroma upsampled with the triangle filter
stb_image uses, converted with the JFIF equations in float.  The library works in integers (as stb_image does), so
the comparison carries a tolerance of a few levels.  numpy only."""
import struct

import numpy as np

ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]

_k = np.arange(8)
DCT = np.sqrt(2.0 / 8) * np.cos((2 * _k[None, :] + 1) * _k[:, None] * np.pi / 16)
DCT[0, :] = np.sqrt(1.0 / 8)

# code tables of the writer's own: every DC category 0..11 gets a 4-bit code, every AC symbol an 8-bit code
DC_SYMBOLS = list(range(12))
AC_SYMBOLS = [0x00, 0xF0] + [(r << 4) | s for r in range(16) for s in range(1, 11)]


def _category(v):
    v = abs(int(v))
    n = 0
    while v:
        n += 1
        v >>= 1
    return n


class _Bits:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value, nbits):
        for i in range(nbits - 1, -1, -1):
            self.acc = (self.acc << 1) | ((value >> i) & 1)
            self.n += 1
            if self.n == 8:
                self.out.append(self.acc)
                if self.acc == 0xFF:
                    self.out.append(0)
                self.acc, self.n = 0, 0

    def flush(self):
        while self.n:
            self.put(1, 1)


def _planes(img, hs, vs):
    """(H, W, 3) or (H, W) uint8 -> list of float planes padded to whole MCUs, with their sampling factors."""
    img = np.asarray(img, np.float64)
    h, w = img.shape[:2]
    mh, mw = 8 * vs, 8 * hs
    H, W = -(-h // mh) * mh, -(-w // mw) * mw
    pad = [(0, H - h), (0, W - w)] + ([(0, 0)] if img.ndim == 3 else [])
    img = np.pad(img, pad, mode="edge")
    if img.ndim == 2:
        return [(img, 1, 1)], 1, 1
    r, g, b = img[..., 0], img[..., 1], img[..., 2]
    y = 0.299 * r + 0.587 * g + 0.114 * b
    cb = -0.168736 * r - 0.331264 * g + 0.5 * b + 128
    cr = 0.5 * r - 0.418688 * g - 0.081312 * b + 128
    sub = lambda p: p.reshape(H // vs, vs, W // hs, hs).mean(axis=(1, 3))
    return [(y, hs, vs), (sub(cb), 1, 1), (sub(cr), 1, 1)], hs, vs


def encode(img, sampling=(1, 1), quant=(8, 12), restart=0, jfif=True, rgb_ids=False, adobe=None):
    """Baseline JPEG bytes + what the writer stored: {'coef': [per component (rows, cols, 64) int], 'q': [...], ...}.
    rgb_ids: write the three planes of img as they are, with component ids 'R','G','B' (no colour transform)."""
    img = np.asarray(img, np.uint8)
    h, w = img.shape[:2]
    hs, vs = sampling
    if rgb_ids:
        f = np.asarray(img, np.float64)
        H, W = -(-h // 8) * 8, -(-w // 8) * 8
        f = np.pad(f, [(0, H - h), (0, W - w), (0, 0)], mode="edge")
        planes, hs, vs = [(f[..., k], 1, 1) for k in range(3)], 1, 1
    else:
        planes, hs, vs = _planes(img, hs, vs)
    ncomp = len(planes)
    qt = []
    for t, base in enumerate(quant):
        q = np.clip(base + (np.add.outer(np.arange(8), np.arange(8)) * (1 + t)), 1, 255).astype(np.int64)
        qt.append(q)
    coefs = []
    for ci, (p, ch, cv) in enumerate(planes):
        q = qt[0 if ci == 0 else 1]
        rows, cols = p.shape[0] // 8, p.shape[1] // 8
        blocks = p.reshape(rows, 8, cols, 8).transpose(0, 2, 1, 3) - 128.0
        F = np.einsum("ij,rcjk,lk->rcil", DCT, blocks, DCT)
        coefs.append(np.rint(F / q).astype(np.int64))
    out = bytearray(b"\xff\xd8")
    if jfif:
        out += b"\xff\xe0" + struct.pack(">H", 16) + b"JFIF\0\x01\x01\0\0\x01\0\x01\0\0"
    if adobe is not None:
        out += b"\xff\xee" + struct.pack(">H", 14) + b"Adobe\0\x64\0\0\0\0" + bytes([adobe])
    for t, q in enumerate(qt):
        out += b"\xff\xdb" + struct.pack(">HB", 67, t) + bytes(int(q.reshape(-1)[z]) for z in ZIGZAG)
    ids = [ord(c) for c in "RGB"] if rgb_ids else [1, 2, 3]
    sof = struct.pack(">BHHB", 8, h, w, ncomp)
    for ci, (p, ch, cv) in enumerate(planes):
        sof += bytes([ids[ci], (ch << 4) | cv, 0 if ci == 0 else 1])
    out += b"\xff\xc0" + struct.pack(">H", 2 + len(sof)) + sof
    for tc, th, syms, length in ((0, 0, DC_SYMBOLS, 4), (1, 0, AC_SYMBOLS, 8)):
        counts = [0] * 16
        counts[length - 1] = len(syms)
        out += b"\xff\xc4" + struct.pack(">HB", 3 + 16 + len(syms), (tc << 4) | th) + bytes(counts) + bytes(syms)
    if restart:
        out += b"\xff\xdd" + struct.pack(">HH", 4, restart)
    sos = bytes([ncomp]) + b"".join(bytes([ids[ci], 0x00]) for ci in range(ncomp)) + b"\0\x3f\0"
    out += b"\xff\xda" + struct.pack(">H", 2 + len(sos)) + sos
    bits = _Bits()
    pred = [0] * ncomp
    mcus_y, mcus_x = planes[0][0].shape[0] // (8 * vs), planes[0][0].shape[1] // (8 * hs)
    count = 0
    rst = 0
    for my in range(mcus_y):
        for mx in range(mcus_x):
            if restart and count and count % restart == 0:
                bits.flush()
                bits.out += bytes([0xFF, 0xD0 + (rst & 7)])
                rst += 1
                pred = [0] * ncomp
            count += 1
            for ci, (p, ch, cv) in enumerate(planes):
                for by in range(cv):
                    for bx in range(ch):
                        blk = coefs[ci][my * cv + by, mx * ch + bx].reshape(-1)
                        zz = [int(blk[z]) for z in ZIGZAG]
                        d = zz[0] - pred[ci]
                        pred[ci] = zz[0]
                        cat = _category(d)
                        assert cat <= 11
                        bits.put(DC_SYMBOLS.index(cat), 4)
                        if cat:
                            bits.put(d if d > 0 else d + (1 << cat) - 1, cat)
                        run = 0
                        last = max([k for k in range(1, 64) if zz[k]], default=0)
                        for k in range(1, last + 1):
                            if zz[k] == 0:
                                run += 1
                                continue
                            while run > 15:
                                bits.put(AC_SYMBOLS.index(0xF0), 8)
                                run -= 16
                            cat = _category(zz[k])
                            assert cat <= 10
                            bits.put(AC_SYMBOLS.index((run << 4) | cat), 8)
                            bits.put(zz[k] if zz[k] > 0 else zz[k] + (1 << cat) - 1, cat)
                            run = 0
                        if last < 63:
                            bits.put(AC_SYMBOLS.index(0x00), 8)
    bits.flush()
    out += bits.out + b"\xff\xd9"
    info = dict(coef=coefs, q=[qt[0]] + [qt[1]] * (ncomp - 1), factors=[(ch, cv) for _, ch, cv in planes], hs=hs, vs=vs, size=(h, w), rgb=rgb_ids)
    return bytes(out), info


def _upsample(plane, hs, vs, width, height):
    """stb_image's chroma upsampling of an integer sample plane to width x height."""
    p = plane.astype(np.int64)
    rows_lo = -(-height // vs)
    w_lo = -(-width // hs)
    out = np.zeros((height, width), np.int64)
    for j in range(height):
        near = min(j // vs, rows_lo - 1) if vs == 2 else j
        if vs == 2:
            far = near - 1 if j % 2 == 0 else near + 1
            far = min(max(far, 0), rows_lo - 1)
        else:
            far = near
        n, f = p[near, :w_lo], p[far, :w_lo]
        if hs == 1 and vs == 1:
            line = n
        elif hs == 1 and vs == 2:
            line = (3 * n + f + 2) >> 2
        elif hs == 2 and vs == 1:
            line = np.zeros(2 * w_lo, np.int64)
            if w_lo == 1:
                line[:] = n[0]
            else:
                line[0] = n[0]
                line[1] = (3 * n[0] + n[1] + 2) >> 2
                for i in range(1, w_lo - 1):
                    line[2 * i] = (3 * n[i] + 2 + n[i - 1]) >> 2
                    line[2 * i + 1] = (3 * n[i] + 2 + n[i + 1]) >> 2
                line[2 * (w_lo - 1)] = (3 * n[w_lo - 2] + n[w_lo - 1] + 2) >> 2
                line[2 * (w_lo - 1) + 1] = n[w_lo - 1]
        else:
            t = 3 * n + f
            line = np.zeros(2 * w_lo, np.int64)
            if w_lo == 1:
                line[:] = (t[0] + 2) >> 2
            else:
                line[0] = (t[0] + 2) >> 2
                for i in range(1, w_lo):
                    line[2 * i - 1] = (3 * t[i - 1] + t[i] + 8) >> 4
                    line[2 * i] = (3 * t[i] + t[i - 1] + 8) >> 4
                line[2 * w_lo - 1] = (t[w_lo - 1] + 2) >> 2
        out[j] = line[:width]
    return out


def expected(info):
    """(H, W, 4) float image the stored coefficients stand for (alpha 255)."""
    h, w = info["size"]
    planes = []
    for coef, q, (ch, cv) in zip(info["coef"], info["q"], info["factors"]):
        F = coef * q
        blocks = np.einsum("ji,rcjk,kl->rcil", DCT, F.astype(np.float64), DCT) + 128.0
        rows, cols = blocks.shape[:2]
        plane = np.clip(np.rint(blocks.transpose(0, 2, 1, 3).reshape(rows * 8, cols * 8)), 0, 255)
        planes.append(_upsample(plane, info["hs"] // ch, info["vs"] // cv, w, h).astype(np.float64))
    out = np.full((h, w, 4), 255.0)
    if len(planes) == 1:
        out[..., 0] = out[..., 1] = out[..., 2] = planes[0]
    elif info["rgb"]:
        for k in range(3):
            out[..., k] = planes[k]
    else:
        y, cb, cr = planes[0], planes[1] - 128, planes[2] - 128
        out[..., 0] = y + 1.402 * cr
        out[..., 1] = y - 0.71414 * cr - 0.34414 * cb
        out[..., 2] = y + 1.772 * cb
    return np.clip(out, 0, 255)
