/*
 * jpeg_decode.h — baseline / extended-sequential / progressive JPEG (8 bits per sample, Huffman) to interleaved RGBA8.
 *
 * The reference gets texture pixels from tinygltf -> stb_image with four requested components
 * (external/include/tiny_gltf.h LoadImageData; src/scene.cpp:69-77).  JPEG decoding is not bit-defined by
 * the standard, so the steps that decide the bytes follow stb_image's choices
 * (external/include/stb_image.h): dequantised coefficients kept as 16-bit values, the integer "islow"
 * inverse DCT with 12-bit constants, two extra bits after the column pass and +128 folded into the row pass
 * (:2426-2524), triangle-filter chroma upsampling for 2x1 / 1x2 / 2x2 and sample repetition otherwise
 * (:3465-3527, :3646-3655, row stepping :3926-3941), fixed-point YCbCr -> RGB (:3659-3683), Adobe / JFIF
 * marker rules for RGB, CMYK and YCCK data (:3879, :3942-3983).  Progressive files (SOF2) keep their coefficients,
 * undequantised, until the last scan: DC first / refinement scans and AC first / refinement scans with end-of-band
 * runs as stb_image reads them (:2265-2412, scan loop :3011-3068), then every block is dequantised and transformed
 * once (:3073-3097).  tests/test_reference_build.py compares the pixels with stb_image's, built from the reference's
 * tree, on sequential and progressive files.  Arithmetic-coded and differential files are refused, as stb_image
 * refuses them.
 */
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace ptjpg {

struct Huff {
    uint8_t size[257];
    uint16_t code[256];
    uint8_t value[256];
    int maxcode[18];     /* left-aligned to 16 bits, per length */
    int delta[17];
    int count = 0;
    bool built = false;
    bool build(const int* counts)
    {
        int k = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < counts[i]; ++j) { if (k >= 256) return false; size[k++] = (uint8_t)(i + 1); }
        size[k] = 0;
        count = k;
        int c = 0;
        k = 0;
        for (int len = 1; len <= 16; ++len) {
            delta[len] = k - c;
            if (size[k] == len) {
                while (size[k] == len) code[k++] = (uint16_t)c++;
                if (c - 1 >= (1 << len)) return false;
            }
            maxcode[len] = c << (16 - len);
            c <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        built = true;
        return true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0, dcPred = 0;
    int x = 0, y = 0, w2 = 0, h2 = 0;
    std::vector<uint8_t> data;
    std::vector<short> coeff;   /* progressive files: 64 coefficients per block, natural order, blocks in rows of coeffW */
    int coeffW = 0;
};

class Decoder {
public:
    Decoder(const uint8_t* bytes, size_t n) : p(bytes), end(bytes + n) {}

    bool run(std::vector<uint8_t>& rgba, int& W, int& H, std::string& err)
    {
        if (!parse(err)) return false;
        W = width; H = height;
        output(rgba);
        return true;
    }

private:
    const uint8_t* p;
    const uint8_t* end;
    int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1, mcuW = 0, mcuH = 0, mcusX = 0, mcusY = 0;
    int restartInterval = 0, jfif = 0, adobeTransform = -1, rgbIds = 0;
    uint16_t dequant[4][64];
    Huff hdc[4], hac[4];
    Component comp[4];
    /* entropy-coded segment reader */
    uint32_t codeBuffer = 0;
    int codeBits = 0;
    bool noMore = false;
    uint8_t marker = 0xff;
    int scanN = 0, order[4] = { 0, 0, 0, 0 };
    /* progressive files: the current scan's band and bit positions, the blocks still covered by an end-of-band run */
    bool progressive = false;
    int specStart = 0, specEnd = 63, succHigh = 0, succLow = 0, eobRun = 0;

    static bool fail(std::string& err, const char* m) { err = m; return false; }
    int get8() { return p < end ? *p++ : 0; }
    int get16() { int a = get8(); return (a << 8) | get8(); }

    void growBuffer()
    {
        do {
            unsigned b = noMore ? 0 : (unsigned)get8();
            if (b == 0xff) {
                int c = get8();
                while (c == 0xff) c = get8();
                if (c != 0) { marker = (uint8_t)c; noMore = true; return; }
            }
            codeBuffer |= b << (24 - codeBits);
            codeBits += 8;
        } while (codeBits <= 24);
    }

    int huffDecode(const Huff& h)
    {
        if (codeBits < 16) growBuffer();
        const int top = (int)(codeBuffer >> 16);
        int len = 1;
        while (len <= 16 && top >= h.maxcode[len]) ++len;
        if (len > 16 || len > codeBits) return -1;
        const int c = (int)((codeBuffer >> (32 - len)) & ((1u << len) - 1)) + h.delta[len];
        if (c < 0 || c >= h.count) return -1;
        codeBits -= len;
        codeBuffer <<= len;
        return h.value[c];
    }

    /* n-bit magnitude with the JPEG sign rule */
    int extendReceive(int n)
    {
        if (n == 0) return 0;
        if (codeBits < n) growBuffer();
        if (codeBits < n) return 0;
        const int sgn = (int)(codeBuffer >> 31);
        const uint32_t k = (codeBuffer << n) | (codeBuffer >> (32 - n));      /* rotate left */
        const uint32_t mask = (1u << n) - 1u;
        codeBuffer = k & ~mask;
        codeBits -= n;
        const int bias = sgn ? 0 : (int)((~0u) << n) + 1;
        return (int)(k & mask) + bias;
    }

    int getBit()
    {
        if (codeBits < 1) growBuffer();
        if (codeBits < 1) return 0;
        const uint32_t k = codeBuffer;
        codeBuffer <<= 1;
        --codeBits;
        return (int)(k >> 31);
    }

    int getBits(int n)
    {
        if (n == 0) return 0;
        if (codeBits < n) growBuffer();
        if (codeBits < n) return 0;
        const uint32_t k = (codeBuffer << n) | (codeBuffer >> (32 - n));
        const uint32_t mask = (1u << n) - 1u;
        codeBuffer = k & ~mask;
        codeBits -= n;
        return (int)(k & mask);
    }

    static const uint8_t* zigzagTable()
    {
        static const uint8_t zigzag[64 + 15] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                                 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                                 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63 };
        return zigzag;
    }

    /* DC coefficient of one block in a progressive scan: the first scan stores the prediction sum shifted up by the
     * scan's low bit position, a refinement scan adds one more bit (stb_image.h:2265-2292) */
    bool decodeBlockProgDc(short data[64], Component& c)
    {
        if (specEnd != 0) return false;
        if (codeBits < 16) growBuffer();
        if (succHigh == 0) {
            memset(data, 0, 64 * sizeof(short));
            const int t = huffDecode(hdc[c.hd]);
            if (t < 0 || t > 15) return false;
            const int diff = t ? extendReceive(t) : 0;
            const long long dc = (long long)c.dcPred + diff;
            if (dc > 2147483647LL || dc < -2147483647LL - 1) return false;
            c.dcPred = (int)dc;
            const long long v = dc * (1LL << succLow);
            if (v > 32767 || v < -32768) return false;
            data[0] = (short)v;
        } else if (getBit()) {
            data[0] = (short)(data[0] + (short)(1 << succLow));
        }
        return true;
    }

    /* AC band of one block in a progressive scan (stb_image.h:2295-2412): first scans place new coefficients and
     * count end-of-band runs over whole blocks; refinement scans add one bit to every coefficient that is already
     * non-zero and place the new +-1 coefficients between them */
    bool decodeBlockProgAc(short data[64], Component& c)
    {
        if (specStart == 0) return false;
        const uint8_t* zigzag = zigzagTable();
        const Huff& ac = hac[c.ha];
        if (succHigh == 0) {
            const int shift = succLow;
            if (eobRun) { --eobRun; return true; }
            int k = specStart;
            do {
                const int rs = huffDecode(ac);
                if (rs < 0) return false;
                const int s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) {
                        eobRun = 1 << r;
                        if (r) eobRun += getBits(r);
                        --eobRun;
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    const int z = zigzag[k++];
                    data[z] = (short)(extendReceive(s) * (1 << shift));
                }
            } while (k <= specEnd);
            return true;
        }
        const short bit = (short)(1 << succLow);
        auto refine = [&](short* q) {
            if (getBit())
                if ((*q & bit) == 0) *q = (short)(*q > 0 ? *q + bit : *q - bit);
        };
        if (eobRun) {
            --eobRun;
            for (int k = specStart; k <= specEnd; ++k) {
                short* q = &data[zigzag[k]];
                if (*q != 0) refine(q);
            }
            return true;
        }
        int k = specStart;
        do {
            const int rs = huffDecode(ac);
            if (rs < 0) return false;
            int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (r < 15) {
                    eobRun = (1 << r) - 1;
                    if (r) eobRun += getBits(r);
                    r = 64;                         /* to the end of the band */
                }
            } else {
                if (s != 1) return false;
                s = getBit() ? bit : -bit;
            }
            while (k <= specEnd) {
                short* q = &data[zigzag[k++]];
                if (*q != 0) refine(q);
                else {
                    if (r == 0) { *q = (short)s; break; }
                    --r;
                }
            }
        } while (k <= specEnd);
        return true;
    }

    bool decodeBlock(short data[64], Component& c)
    {
        const uint8_t* zigzag = zigzagTable();
        const Huff& dc = hdc[c.hd];
        const Huff& ac = hac[c.ha];
        const uint16_t* dq = dequant[c.tq];
        if (codeBits < 16) growBuffer();
        const int t = huffDecode(dc);
        if (t < 0 || t > 15) return false;
        memset(data, 0, 64 * sizeof(short));
        const int diff = t ? extendReceive(t) : 0;
        c.dcPred = (int)((unsigned)c.dcPred + (unsigned)diff);          /* wraps on damaged streams instead of overflowing */
        data[0] = (short)((unsigned)c.dcPred * (unsigned)dq[0]);
        int k = 1;
        do {
            const int rs = huffDecode(ac);
            if (rs < 0) return false;
            const int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xf0) break;
                k += 16;
            } else {
                k += r;
                const int z = zigzag[k++];
                data[z] = (short)(extendReceive(s) * dq[z]);
            }
        } while (k < 64);
        return true;
    }

    typedef long long I64;      /* the transform's intermediates: 32 bits hold them for any valid stream, 64 keep damaged ones defined */
    static uint8_t clamp8(I64 x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

    /* one pass of the 8-point inverse DCT on s[0..7] (stride st): 12-bit constants, results scaled by 4096 */
    static void idct1d(const int* s, int st, I64& x0, I64& x1, I64& x2, I64& x3, I64& t0, I64& t1, I64& t2, I64& t3)
    {
        auto fx = [](double v) { return (int)(v * 4096 + 0.5); };
        static const int c0541 = fx(0.5411961f), c1847 = fx(-1.847759065f), c0765 = fx(0.765366865f), c1175 = fx(1.175875602f),
                         c0298 = fx(0.298631336f), c2053 = fx(2.053119869f), c3072 = fx(3.072711026f), c1501 = fx(1.501321110f),
                         c0899 = fx(-0.899976223f), c2562 = fx(-2.562915447f), c1961 = fx(-1.961570560f), c0390 = fx(-0.390180644f);
        I64 p2 = s[2 * st], p3 = s[6 * st];
        I64 p1 = (p2 + p3) * c0541;
        const I64 e2 = p1 + p3 * c1847, e3 = p1 + p2 * c0765;
        p2 = s[0]; p3 = s[4 * st];
        const I64 e0 = (p2 + p3) * 4096, e1 = (p2 - p3) * 4096;
        x0 = e0 + e3; x3 = e0 - e3; x1 = e1 + e2; x2 = e1 - e2;
        t0 = s[7 * st]; t1 = s[5 * st]; t2 = s[3 * st]; t3 = s[1 * st];
        p3 = t0 + t2;
        I64 p4 = t1 + t3;
        p1 = t0 + t3;
        p2 = t1 + t2;
        const I64 p5 = (p3 + p4) * c1175;
        t0 = t0 * c0298; t1 = t1 * c2053; t2 = t2 * c3072; t3 = t3 * c1501;
        p1 = p5 + p1 * c0899;
        p2 = p5 + p2 * c2562;
        p3 = p3 * c1961;
        p4 = p4 * c0390;
        t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;
    }

    static void idctBlock(uint8_t* out, int stride, const short data[64])
    {
        int val[64];
        for (int i = 0; i < 8; ++i) {
            const short* d = data + i;
            int* v = val + i;
            if (d[8] == 0 && d[16] == 0 && d[24] == 0 && d[32] == 0 && d[40] == 0 && d[48] == 0 && d[56] == 0) {
                const int dc = d[0] * 4;
                for (int r = 0; r < 8; ++r) v[8 * r] = dc;
                continue;
            }
            int col[8];
            for (int r = 0; r < 8; ++r) col[r] = d[8 * r];
            I64 x0, x1, x2, x3, t0, t1, t2, t3;
            idct1d(col, 1, x0, x1, x2, x3, t0, t1, t2, t3);
            x0 += 512; x1 += 512; x2 += 512; x3 += 512;       /* keep two extra bits */
            v[0] = (int)((x0 + t3) >> 10); v[56] = (int)((x0 - t3) >> 10);
            v[8] = (int)((x1 + t2) >> 10); v[48] = (int)((x1 - t2) >> 10);
            v[16] = (int)((x2 + t1) >> 10); v[40] = (int)((x2 - t1) >> 10);
            v[24] = (int)((x3 + t0) >> 10); v[32] = (int)((x3 - t0) >> 10);
        }
        for (int i = 0; i < 8; ++i) {
            I64 x0, x1, x2, x3, t0, t1, t2, t3;
            idct1d(val + 8 * i, 1, x0, x1, x2, x3, t0, t1, t2, t3);
            const I64 bias = 65536 + (128 << 17);             /* rounding and the level shift in one */
            x0 += bias; x1 += bias; x2 += bias; x3 += bias;
            uint8_t* o = out + (size_t)i * stride;
            o[0] = clamp8((x0 + t3) >> 17); o[7] = clamp8((x0 - t3) >> 17);
            o[1] = clamp8((x1 + t2) >> 17); o[6] = clamp8((x1 - t2) >> 17);
            o[2] = clamp8((x2 + t1) >> 17); o[5] = clamp8((x2 - t1) >> 17);
            o[3] = clamp8((x3 + t0) >> 17); o[4] = clamp8((x3 - t0) >> 17);
        }
    }

    void resetEntropy()
    {
        codeBits = 0; codeBuffer = 0; noMore = false; marker = 0xff;
        eobRun = 0;
        for (int i = 0; i < 4; ++i) comp[i].dcPred = 0;
    }

    bool restartIfDue(int& todo)
    {
        if (--todo > 0) return true;
        if (codeBits < 24) growBuffer();
        if (!(marker >= 0xd0 && marker <= 0xd7)) return false;     /* the scan ends here */
        resetEntropy();
        todo = restartInterval ? restartInterval : 0x7fffffff;
        return true;
    }

    bool decodeScan(std::string& err)
    {
        resetEntropy();
        int todo = restartInterval ? restartInterval : 0x7fffffff;
        short block[64];
        if (progressive) {
            if (scanN == 1) {
                Component& c = comp[order[0]];
                const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
                for (int j = 0; j < h; ++j)
                    for (int i = 0; i < w; ++i) {
                        short* data = c.coeff.data() + 64 * ((size_t)i + (size_t)j * c.coeffW);
                        const bool ok = specStart == 0 ? decodeBlockProgDc(data, c) : decodeBlockProgAc(data, c);
                        if (!ok) return fail(err, "bad progressive JPEG scan");
                        if (!restartIfDue(todo)) return true;
                    }
                return true;
            }
            for (int j = 0; j < mcusY; ++j)
                for (int i = 0; i < mcusX; ++i) {
                    for (int k = 0; k < scanN; ++k) {
                        Component& c = comp[order[k]];
                        for (int y = 0; y < c.v; ++y)
                            for (int x = 0; x < c.h; ++x) {
                                const int x2 = i * c.h + x, y2 = j * c.v + y;
                                short* data = c.coeff.data() + 64 * ((size_t)x2 + (size_t)y2 * c.coeffW);
                                if (!decodeBlockProgDc(data, c)) return fail(err, "bad progressive JPEG scan");     /* interleaved scans carry DC only */
                            }
                    }
                    if (!restartIfDue(todo)) return true;
                }
            return true;
        }
        if (scanN == 1) {
            Component& c = comp[order[0]];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int j = 0; j < h; ++j)
                for (int i = 0; i < w; ++i) {
                    if (!decodeBlock(block, c)) return fail(err, "bad JPEG Huffman code");
                    idctBlock(c.data.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, block);
                    if (!restartIfDue(todo)) return true;
                }
            return true;
        }
        for (int j = 0; j < mcusY; ++j)
            for (int i = 0; i < mcusX; ++i) {
                for (int k = 0; k < scanN; ++k) {
                    Component& c = comp[order[k]];
                    for (int y = 0; y < c.v; ++y)
                        for (int x = 0; x < c.h; ++x) {
                            const int x2 = (i * c.h + x) * 8, y2 = (j * c.v + y) * 8;
                            if (!decodeBlock(block, c)) return fail(err, "bad JPEG Huffman code");
                            idctBlock(c.data.data() + (size_t)c.w2 * y2 + x2, c.w2, block);
                        }
                }
                if (!restartIfDue(todo)) return true;
            }
        return true;
    }

    /* progressive files: dequantise and transform every block that holds pixels, once all scans are in (stb_image.h:3080-3097) */
    void finishProgressive()
    {
        for (int n = 0; n < ncomp; ++n) {
            Component& c = comp[n];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            const uint16_t* dq = dequant[c.tq];
            for (int j = 0; j < h; ++j)
                for (int i = 0; i < w; ++i) {
                    short* data = c.coeff.data() + 64 * ((size_t)i + (size_t)j * c.coeffW);
                    for (int k = 0; k < 64; ++k) data[k] = (short)(data[k] * dq[k]);
                    idctBlock(c.data.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, data);
                }
        }
    }

    bool parse(std::string& err)
    {
        if (end - p < 4 || p[0] != 0xff || p[1] != 0xd8) return fail(err, "not a JPEG stream");
        p += 2;
        bool haveFrame = false, decoded = false;
        for (;;) {
            int m;
            if (marker != 0xff) { m = marker; marker = 0xff; }
            else {
                if (p >= end) break;
                int x = get8();
                if (x != 0xff) continue;                  /* skip to the next marker */
                while (x == 0xff && p < end) x = get8();
                m = x;
                if (m == 0) continue;
            }
            if (m == 0xd9) break;
            if (m >= 0xd0 && m <= 0xd7) continue;
            if (m == 0xc3 || m == 0xc5 || m == 0xc6 || m == 0xc7) return fail(err, "lossless / differential JPEG is not supported");
            if (m == 0xc9 || m == 0xca || m == 0xcb || m == 0xcd || m == 0xce || m == 0xcf) return fail(err, "arithmetic-coded JPEG is not supported");
            if (end - p < 2) return fail(err, "truncated JPEG");
            int len = get16() - 2;
            if (len < 0 || len > end - p) return fail(err, "bad JPEG segment length");
            const uint8_t* seg = p;
            const uint8_t* segEnd = p + len;
            switch (m) {
            case 0xdb:
                while (seg < segEnd) {
                    const int q = *seg++, prec = q >> 4, t = q & 15;
                    if ((prec != 0 && prec != 1) || t > 3) return fail(err, "bad DQT");
                    static const uint8_t zz[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                                    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };
                    if (segEnd - seg < (prec ? 128 : 64)) return fail(err, "bad DQT");
                    for (int i = 0; i < 64; ++i) {
                        int v = *seg++;
                        if (prec) v = (v << 8) | *seg++;
                        dequant[t][zz[i]] = (uint16_t)v;
                    }
                }
                break;
            case 0xc4:
                while (seg < segEnd) {
                    if (segEnd - seg < 17) return fail(err, "bad DHT");
                    const int q = *seg++, tc = q >> 4, th = q & 15;
                    if (tc > 1 || th > 3) return fail(err, "bad DHT");
                    int counts[16], total = 0;
                    for (int i = 0; i < 16; ++i) { counts[i] = *seg++; total += counts[i]; }
                    if (total > 256 || segEnd - seg < total) return fail(err, "bad DHT");
                    Huff& h = tc ? hac[th] : hdc[th];
                    if (!h.build(counts)) return fail(err, "bad JPEG code lengths");
                    memcpy(h.value, seg, (size_t)total);
                    seg += total;
                }
                break;
            case 0xdd:
                if (len != 2) return fail(err, "bad DRI");
                restartInterval = (seg[0] << 8) | seg[1];
                break;
            case 0xe0:
                if (len >= 5 && !memcmp(seg, "JFIF\0", 5)) jfif = 1;
                break;
            case 0xee:
                if (len >= 12 && !memcmp(seg, "Adobe\0", 6)) adobeTransform = seg[11];
                break;
            case 0xc0: case 0xc1: case 0xc2: {
                if (haveFrame || len < 6) return fail(err, "bad SOF");
                progressive = m == 0xc2;
                if (seg[0] != 8) return fail(err, "only 8-bit JPEG is supported");
                height = (seg[1] << 8) | seg[2]; width = (seg[3] << 8) | seg[4]; ncomp = seg[5];
                if (width <= 0 || height <= 0) return fail(err, "bad JPEG size");
                if (ncomp != 1 && ncomp != 3 && ncomp != 4) return fail(err, "bad JPEG component count");
                if (len != 6 + 3 * ncomp) return fail(err, "bad SOF length");
                static const uint8_t rgb[3] = { 'R', 'G', 'B' };
                for (int i = 0; i < ncomp; ++i) {
                    Component& c = comp[i];
                    c.id = seg[6 + 3 * i];
                    if (ncomp == 3 && c.id == rgb[i]) ++rgbIds;
                    c.h = seg[7 + 3 * i] >> 4; c.v = seg[7 + 3 * i] & 15; c.tq = seg[8 + 3 * i];
                    if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return fail(err, "bad JPEG sampling factors");
                    if (c.h > hmax) hmax = c.h;
                    if (c.v > vmax) vmax = c.v;
                }
                for (int i = 0; i < ncomp; ++i)
                    if (hmax % comp[i].h != 0 || vmax % comp[i].v != 0) return fail(err, "bad JPEG sampling factors");
                mcuW = hmax * 8; mcuH = vmax * 8;
                mcusX = (width + mcuW - 1) / mcuW; mcusY = (height + mcuH - 1) / mcuH;
                if ((size_t)mcusX * mcuW * (size_t)mcusY * mcuH > ((size_t)1 << 30)) return fail(err, "JPEG too large");
                for (int i = 0; i < ncomp; ++i) {
                    Component& c = comp[i];
                    c.x = (width * c.h + hmax - 1) / hmax;
                    c.y = (height * c.v + vmax - 1) / vmax;
                    c.w2 = mcusX * c.h * 8;
                    c.h2 = mcusY * c.v * 8;
                    c.data.assign((size_t)c.w2 * c.h2, 0);
                    if (progressive) { c.coeffW = c.w2 / 8; c.coeff.assign((size_t)c.w2 * c.h2, 0); }
                }
                haveFrame = true;
                break;
            }
            case 0xda: {
                if (!haveFrame || len < 1) return fail(err, "SOS before SOF");
                scanN = seg[0];
                if (scanN < 1 || scanN > ncomp || len != 4 + 2 * scanN) return fail(err, "bad SOS");
                for (int i = 0; i < scanN; ++i) {
                    const int id = seg[1 + 2 * i], q = seg[2 + 2 * i];
                    int which = 0;
                    while (which < ncomp && comp[which].id != id) ++which;
                    if (which == ncomp) return fail(err, "bad SOS component");
                    comp[which].hd = q >> 4; comp[which].ha = q & 15;
                    if (comp[which].hd > 3 || comp[which].ha > 3) return fail(err, "bad SOS table index");
                    order[i] = which;
                }
                specStart = seg[1 + 2 * scanN]; specEnd = seg[2 + 2 * scanN];
                succHigh = seg[3 + 2 * scanN] >> 4; succLow = seg[3 + 2 * scanN] & 15;
                if (progressive) {
                    if (specStart > 63 || specEnd > 63 || specStart > specEnd || succHigh > 13 || succLow > 13) return fail(err, "bad SOS");
                } else {
                    if (specStart != 0 || succHigh != 0 || succLow != 0) return fail(err, "bad SOS (spectral selection in a sequential file)");
                    specEnd = 63;
                }
                for (int i = 0; i < scanN; ++i) {
                    /* the tables this scan reads: both in a sequential file, DC only or AC only in a progressive one */
                    const Component& c = comp[order[i]];
                    const bool needDc = !progressive || specStart == 0, needAc = !progressive || specStart != 0;
                    if ((needDc && !hdc[c.hd].built) || (needAc && !hac[c.ha].built)) return fail(err, "JPEG scan without its Huffman table");
                }
                p = segEnd;
                if (!decodeScan(err)) return false;
                decoded = true;
                continue;                                   /* p already stands behind the entropy-coded data */
            }
            default: break;
            }
            p = segEnd;
        }
        if (!decoded) return fail(err, "JPEG without image data");
        if (progressive) finishProgressive();
        return true;
    }

    /* upsample one component row into line[] (at least width + 3 bytes) */
    static const uint8_t* resample(uint8_t* line, const uint8_t* nearRow, const uint8_t* farRow, int w, int hs, int vs)
    {
        if (hs == 1 && vs == 1) return nearRow;
        if (hs == 1 && vs == 2) {
            for (int i = 0; i < w; ++i) line[i] = (uint8_t)((3 * nearRow[i] + farRow[i] + 2) >> 2);
            return line;
        }
        if (hs == 2 && vs == 1) {
            const uint8_t* in = nearRow;
            if (w == 1) { line[0] = line[1] = in[0]; return line; }
            line[0] = in[0];
            line[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
            int i;
            for (i = 1; i < w - 1; ++i) {
                const int n = 3 * in[i] + 2;
                line[i * 2] = (uint8_t)((n + in[i - 1]) >> 2);
                line[i * 2 + 1] = (uint8_t)((n + in[i + 1]) >> 2);
            }
            line[i * 2] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
            line[i * 2 + 1] = in[w - 1];
            return line;
        }
        if (hs == 2 && vs == 2) {
            if (w == 1) { line[0] = line[1] = (uint8_t)((3 * nearRow[0] + farRow[0] + 2) >> 2); return line; }
            int t1 = 3 * nearRow[0] + farRow[0];
            line[0] = (uint8_t)((t1 + 2) >> 2);
            for (int i = 1; i < w; ++i) {
                const int t0 = t1;
                t1 = 3 * nearRow[i] + farRow[i];
                line[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
                line[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
            }
            line[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
            return line;
        }
        for (int i = 0; i < w; ++i)
            for (int j = 0; j < hs; ++j) line[i * hs + j] = nearRow[i];
        return line;
    }

    static uint8_t mul8(uint8_t x, uint8_t y) { unsigned t = (unsigned)x * y + 128; return (uint8_t)((t + (t >> 8)) >> 8); }

    static void ycc(uint8_t* out, const uint8_t* y, const uint8_t* pcb, const uint8_t* pcr, int count)
    {
        auto fx = [](float v) { return ((int)(v * 4096.0f + 0.5f)) << 8; };
        static const int kr = fx(1.40200f), kgr = fx(0.71414f), kgb = fx(0.34414f), kb = fx(1.77200f);
        for (int i = 0; i < count; ++i) {
            const int yf = (y[i] << 20) + (1 << 19);
            const int cr = pcr[i] - 128, cb = pcb[i] - 128;
            int r = yf + cr * kr;
            int g = yf + (cr * -kgr) + (int)((unsigned)(cb * -kgb) & 0xffff0000u);
            int b = yf + cb * kb;
            r >>= 20; g >>= 20; b >>= 20;
            out[0] = clamp8(r); out[1] = clamp8(g); out[2] = clamp8(b); out[3] = 255;
            out += 4;
        }
    }

    void output(std::vector<uint8_t>& rgba)
    {
        rgba.assign((size_t)width * height * 4, 255);
        const bool isRgb = ncomp == 3 && (rgbIds == 3 || (adobeTransform == 0 && !jfif));
        struct Res { int hs, vs, ystep, wLo, ypos; const uint8_t* line0; const uint8_t* line1; std::vector<uint8_t> buf; } res[4];
        for (int k = 0; k < ncomp; ++k) {
            Res& r = res[k];
            r.hs = hmax / comp[k].h; r.vs = vmax / comp[k].v;
            r.ystep = r.vs >> 1;
            r.wLo = (width + r.hs - 1) / r.hs;
            r.ypos = 0;
            r.line0 = r.line1 = comp[k].data.data();
            r.buf.assign((size_t)width + 8 + (size_t)r.hs * 2, 0);
        }
        const uint8_t* rows[4] = { nullptr, nullptr, nullptr, nullptr };
        for (int j = 0; j < height; ++j) {
            uint8_t* out = &rgba[(size_t)j * width * 4];
            for (int k = 0; k < ncomp; ++k) {
                Res& r = res[k];
                const bool bot = r.ystep >= (r.vs >> 1);
                rows[k] = resample(r.buf.data(), bot ? r.line1 : r.line0, bot ? r.line0 : r.line1, r.wLo, r.hs, r.vs);
                if (++r.ystep >= r.vs) {
                    r.ystep = 0;
                    r.line0 = r.line1;
                    if (++r.ypos < comp[k].y) r.line1 += comp[k].w2;
                }
            }
            if (ncomp == 3) {
                if (isRgb) for (int i = 0; i < width; ++i) { out[4 * i] = rows[0][i]; out[4 * i + 1] = rows[1][i]; out[4 * i + 2] = rows[2][i]; }
                else ycc(out, rows[0], rows[1], rows[2], width);
            } else if (ncomp == 4) {
                if (adobeTransform == 0) {                   /* CMYK */
                    for (int i = 0; i < width; ++i) {
                        const uint8_t m = rows[3][i];
                        out[4 * i] = mul8(rows[0][i], m); out[4 * i + 1] = mul8(rows[1][i], m); out[4 * i + 2] = mul8(rows[2][i], m);
                    }
                } else {
                    ycc(out, rows[0], rows[1], rows[2], width);
                    if (adobeTransform == 2)                /* YCCK */
                        for (int i = 0; i < width; ++i) {
                            const uint8_t m = rows[3][i];
                            out[4 * i] = mul8((uint8_t)(255 - out[4 * i]), m); out[4 * i + 1] = mul8((uint8_t)(255 - out[4 * i + 1]), m);
                            out[4 * i + 2] = mul8((uint8_t)(255 - out[4 * i + 2]), m);
                        }
                }
            } else {
                for (int i = 0; i < width; ++i) out[4 * i] = out[4 * i + 1] = out[4 * i + 2] = rows[0][i];
            }
        }
    }
};

inline bool decode(const uint8_t* bytes, size_t n, std::vector<uint8_t>& rgba, int& W, int& H, std::string& err)
{
    Decoder d(bytes, n);
    return d.run(rgba, W, H, err);
}

} // namespace ptjpg
