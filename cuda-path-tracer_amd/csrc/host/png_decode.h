/*
 * png_decode.h — PNG (8 bits per sample or fewer) to interleaved RGBA8, with its own inflate.
 *
 * The reference gets texture pixels from tinygltf, which hands every image to stb_image with four
 * requested components (external/include/tiny_gltf.h LoadImageData; src/scene.cpp:69-77 copies
 * width*height*component bytes).  PNG is lossless, so any conforming decoder yields the same samples;
 * what is restated here is stb_image's expansion to RGBA: grey -> (g,g,g,255), sub-byte grey scaled by
 * 255/(2^depth-1), palette entries with tRNS alpha, a tRNS colour key -> alpha 0.
 * 16-bit images are refused (the reference would copy only half of their data).
 */
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace ptpng {

/* ---- inflate (RFC 1951) behind a zlib wrapper (RFC 1950) ---- */
class Inflater {
public:
    Inflater(const uint8_t* data, size_t n) : in(data), len(n) {}
    bool run(std::vector<uint8_t>& out, std::string& err)
    {
        if (len < 2) return fail(err, "zlib stream too short");
        unsigned cmf = in[0], flg = in[1];
        if ((cmf & 15) != 8 || ((cmf << 8) | flg) % 31 != 0 || (flg & 32)) return fail(err, "bad zlib header");
        pos = 2;
        bool last = false;
        while (!last) {
            uint32_t hdr;
            if (!bits(3, hdr)) return fail(err, "truncated deflate stream");
            last = hdr & 1;
            switch (hdr >> 1) {
            case 0: if (!stored(out)) return fail(err, "bad stored block"); break;
            case 1: fixedTables(); if (!codes(out)) return fail(err, "bad fixed-Huffman block"); break;
            case 2: if (!dynamicTables() || !codes(out)) return fail(err, "bad dynamic-Huffman block"); break;
            default: return fail(err, "reserved deflate block type");
            }
        }
        /* Adler-32 of the output follows, big endian, on a byte boundary */
        bitbuf = 0; bitcnt = 0;
        if (pos + 4 <= len) {
            uint32_t want = ((uint32_t)in[pos] << 24) | ((uint32_t)in[pos + 1] << 16) | ((uint32_t)in[pos + 2] << 8) | in[pos + 3];
            uint32_t a = 1, b = 0;
            for (size_t i = 0; i < out.size(); ++i) { a = (a + out[i]) % 65521u; b = (b + a) % 65521u; }
            if (((b << 16) | a) != want) return fail(err, "Adler-32 mismatch");
        }
        return true;
    }

private:
    struct Huff { uint16_t count[16]; uint16_t symbol[320]; };
    const uint8_t* in;
    size_t len, pos = 0;
    uint32_t bitbuf = 0;
    int bitcnt = 0;
    Huff lit, dist;

    static bool fail(std::string& err, const char* m) { err = m; return false; }

    bool bits(int need, uint32_t& v)
    {
        while (bitcnt < need) {
            if (pos >= len) return false;
            bitbuf |= (uint32_t)in[pos++] << bitcnt;
            bitcnt += 8;
        }
        v = bitbuf & ((1u << need) - 1u);
        bitbuf >>= need;
        bitcnt -= need;
        return true;
    }

    bool stored(std::vector<uint8_t>& out)
    {
        bitbuf = 0; bitcnt = 0;
        if (pos + 4 > len) return false;
        unsigned n = in[pos] | (in[pos + 1] << 8), nn = in[pos + 2] | (in[pos + 3] << 8);
        pos += 4;
        if ((n ^ 0xFFFFu) != nn || pos + n > len) return false;
        out.insert(out.end(), in + pos, in + pos + n);
        pos += n;
        return true;
    }

    /* canonical code from code lengths; returns false for an over-subscribed set */
    static bool build(Huff& h, const uint8_t* lengths, int n)
    {
        memset(h.count, 0, sizeof(h.count));
        for (int i = 0; i < n; ++i) h.count[lengths[i]]++;
        int left = 1;
        for (int l = 1; l < 16; ++l) {
            left <<= 1;
            left -= h.count[l];
            if (left < 0) return false;
        }
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + h.count[l];
        for (int i = 0; i < n; ++i)
            if (lengths[i]) h.symbol[offs[lengths[i]]++] = (uint16_t)i;
        return true;
    }

    int decode(const Huff& h)
    {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; ++l) {
            uint32_t b;
            if (!bits(1, b)) return -1;
            code |= (int)b;
            int count = h.count[l];
            if (code - count < first) return h.symbol[index + (code - first)];
            index += count;
            first += count;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }

    void fixedTables()
    {
        uint8_t l[288];
        int i = 0;
        for (; i < 144; ++i) l[i] = 8;
        for (; i < 256; ++i) l[i] = 9;
        for (; i < 280; ++i) l[i] = 7;
        for (; i < 288; ++i) l[i] = 8;
        build(lit, l, 288);
        for (i = 0; i < 30; ++i) l[i] = 5;
        build(dist, l, 30);
    }

    bool dynamicTables()
    {
        static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
        uint32_t nlen, ndist, ncode;
        if (!bits(5, nlen) || !bits(5, ndist) || !bits(4, ncode)) return false;
        nlen += 257; ndist += 1; ncode += 4;
        if (nlen > 286 || ndist > 30) return false;
        uint8_t l[320];
        memset(l, 0, sizeof(l));
        for (uint32_t i = 0; i < ncode; ++i) { uint32_t v; if (!bits(3, v)) return false; l[order[i]] = (uint8_t)v; }
        Huff lencode;
        if (!build(lencode, l, 19)) return false;
        uint8_t ll[320];
        uint32_t idx = 0;
        while (idx < nlen + ndist) {
            int sym = decode(lencode);
            if (sym < 0) return false;
            if (sym < 16) { ll[idx++] = (uint8_t)sym; continue; }
            uint8_t prev = 0;
            uint32_t rep;
            if (sym == 16) { if (idx == 0) return false; prev = ll[idx - 1]; if (!bits(2, rep)) return false; rep += 3; }
            else if (sym == 17) { if (!bits(3, rep)) return false; rep += 3; }
            else { if (!bits(7, rep)) return false; rep += 11; }
            if (idx + rep > nlen + ndist) return false;
            while (rep--) ll[idx++] = prev;
        }
        if (ll[256] == 0) return false;
        if (!build(lit, ll, (int)nlen)) return false;
        if (!build(dist, ll + nlen, (int)ndist)) return false;
        return true;
    }

    bool codes(std::vector<uint8_t>& out)
    {
        static const uint16_t lbase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
        static const uint8_t lext[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
        static const uint16_t dbase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
        static const uint8_t dext[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
        for (;;) {
            int sym = decode(lit);
            if (sym < 0) return false;
            if (sym < 256) { out.push_back((uint8_t)sym); continue; }
            if (sym == 256) return true;
            sym -= 257;
            if (sym >= 29) return false;
            uint32_t e;
            if (!bits(lext[sym], e)) return false;
            size_t length = lbase[sym] + e;
            int ds = decode(dist);
            if (ds < 0 || ds >= 30) return false;
            if (!bits(dext[ds], e)) return false;
            size_t d = dbase[ds] + e;
            if (d > out.size()) return false;
            size_t from = out.size() - d;
            for (size_t i = 0; i < length; ++i) out.push_back(out[from + i]);
        }
    }
};

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline int paeth(int a, int b, int c)
{
    int p = a + b - c;
    int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    if (pa <= pb && pa <= pc) return a;
    if (pb <= pc) return b;
    return c;
}

/* Undo the scanline filters of one (sub-)image in place; rows are 1 + stride bytes. Returns false on a bad filter byte. */
inline bool unfilter(uint8_t* d, int rows, size_t stride, int bpp)
{
    std::vector<uint8_t> zero(stride, 0);
    const uint8_t* prev = zero.data();
    for (int y = 0; y < rows; ++y) {
        uint8_t* row = d + (size_t)y * (stride + 1);
        uint8_t* cur = row + 1;
        switch (row[0]) {
        case 0: break;
        case 1: for (size_t i = bpp; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]); break;
        case 2: for (size_t i = 0; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + prev[i]); break;
        case 3:
            for (size_t i = 0; i < stride; ++i) {
                int left = i >= (size_t)bpp ? cur[i - bpp] : 0;
                cur[i] = (uint8_t)(cur[i] + ((left + prev[i]) >> 1));
            }
            break;
        case 4:
            for (size_t i = 0; i < stride; ++i) {
                int left = i >= (size_t)bpp ? cur[i - bpp] : 0, ul = i >= (size_t)bpp ? prev[i - bpp] : 0;
                cur[i] = (uint8_t)(cur[i] + paeth(left, prev[i], ul));
            }
            break;
        default: return false;
        }
        prev = cur;
    }
    return true;
}

/* PNG bytes -> width*height*4 RGBA8. */
inline bool decode(const uint8_t* p, size_t n, std::vector<uint8_t>& rgba, int& W, int& H, std::string& err)
{
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    if (n < 8 || memcmp(p, sig, 8) != 0) { err = "not a PNG stream"; return false; }
    size_t pos = 8;
    int depth = 0, ctype = 0, interlace = 0;
    bool haveHdr = false, haveTrns = false;
    uint8_t pal[256][4];
    int palN = 0;
    uint8_t key[3] = { 0, 0, 0 };
    for (int i = 0; i < 256; ++i) { pal[i][0] = pal[i][1] = pal[i][2] = 0; pal[i][3] = 255; }
    std::vector<uint8_t> z;
    for (;;) {
        if (pos + 8 > n) { err = "truncated PNG"; return false; }
        uint32_t clen = be32(p + pos);
        const uint8_t* tag = p + pos + 4;
        const uint8_t* body = p + pos + 8;
        if ((size_t)clen > n - pos - 8 || pos + 12 + clen > n) { err = "truncated PNG chunk"; return false; }
        if (!memcmp(tag, "IHDR", 4)) {
            if (clen != 13) { err = "bad IHDR"; return false; }
            W = (int)be32(body); H = (int)be32(body + 4);
            depth = body[8]; ctype = body[9]; interlace = body[12];
            if (W <= 0 || H <= 0 || W > (1 << 24) || H > (1 << 24)) { err = "bad PNG size"; return false; }
            if (body[10] != 0 || body[11] != 0 || interlace > 1) { err = "bad PNG method"; return false; }
            if (depth == 16) { err = "16-bit PNG is not supported"; return false; }
            bool ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) || (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                      ((ctype == 2 || ctype == 4 || ctype == 6) && depth == 8);
            if (!ok) { err = "bad PNG colour type / depth"; return false; }
            haveHdr = true;
        } else if (!haveHdr) { err = "PNG: first chunk is not IHDR"; return false; }
        else if (!memcmp(tag, "PLTE", 4)) {
            if (clen > 768 || clen % 3) { err = "bad PLTE"; return false; }
            palN = (int)clen / 3;
            for (int i = 0; i < palN; ++i) { pal[i][0] = body[3 * i]; pal[i][1] = body[3 * i + 1]; pal[i][2] = body[3 * i + 2]; }
        } else if (!memcmp(tag, "tRNS", 4)) {
            if (ctype == 3) { for (uint32_t i = 0; i < clen && i < 256; ++i) pal[i][3] = body[i]; }
            else if (ctype == 0 && clen >= 2) { key[0] = body[1]; haveTrns = true; }
            else if (ctype == 2 && clen >= 6) { key[0] = body[1]; key[1] = body[3]; key[2] = body[5]; haveTrns = true; }
        } else if (!memcmp(tag, "IDAT", 4)) z.insert(z.end(), body, body + clen);
        else if (!memcmp(tag, "IEND", 4)) break;
        pos += 12 + (size_t)clen;
    }
    if (!haveHdr || z.empty()) { err = "PNG without image data"; return false; }
    if (ctype == 3 && palN == 0) { err = "paletted PNG without PLTE"; return false; }
    std::vector<uint8_t> raw;
    {
        Inflater inf(z.data(), z.size());
        if (!inf.run(raw, err)) return false;
    }
    const int chan = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4;
    const int bpp = (chan * depth + 7) / 8;
    rgba.assign((size_t)W * H * 4, 0);
    const int scale = (ctype == 0) ? (depth == 1 ? 255 : depth == 2 ? 85 : depth == 4 ? 17 : 1) : 1;

    auto emit = [&](const uint8_t* row, int xcount, int y, int x0, int dx) {
        for (int i = 0; i < xcount; ++i) {
            uint8_t s[4];
            if (depth == 8) memcpy(s, row + (size_t)i * chan, chan);
            else {
                int bit = i * depth;
                s[0] = (uint8_t)((row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1));
            }
            uint8_t* o = &rgba[((size_t)y * W + (x0 + i * dx)) * 4];
            switch (ctype) {
            case 0: {
                uint8_t g = (uint8_t)(s[0] * scale);
                o[0] = o[1] = o[2] = g;
                o[3] = (haveTrns && g == (uint8_t)(key[0] * scale)) ? 0 : 255;
                break;
            }
            case 2:
                o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
                o[3] = (haveTrns && s[0] == key[0] && s[1] == key[1] && s[2] == key[2]) ? 0 : 255;
                break;
            case 3: memcpy(o, pal[s[0]], 4); break;
            case 4: o[0] = o[1] = o[2] = s[0]; o[3] = s[1]; break;
            default: memcpy(o, s, 4); break;
            }
        }
    };

    if (!interlace) {
        size_t stride = ((size_t)W * chan * depth + 7) / 8;
        if (raw.size() < (stride + 1) * (size_t)H) { err = "PNG image data too short"; return false; }
        if (!unfilter(raw.data(), H, stride, bpp)) { err = "bad PNG filter"; return false; }
        for (int y = 0; y < H; ++y) emit(raw.data() + (size_t)y * (stride + 1) + 1, W, y, 0, 1);
    } else {
        static const int xo[7] = { 0, 4, 0, 2, 0, 1, 0 }, yo[7] = { 0, 0, 4, 0, 2, 0, 1 };
        static const int xs[7] = { 8, 8, 4, 4, 2, 2, 1 }, ys[7] = { 8, 8, 8, 4, 4, 2, 2 };
        size_t off = 0;
        for (int ps = 0; ps < 7; ++ps) {
            int pw = (W - xo[ps] + xs[ps] - 1) / xs[ps], ph = (H - yo[ps] + ys[ps] - 1) / ys[ps];
            if (pw <= 0 || ph <= 0) continue;
            size_t stride = ((size_t)pw * chan * depth + 7) / 8;
            if (raw.size() < off + (stride + 1) * (size_t)ph) { err = "PNG image data too short"; return false; }
            if (!unfilter(raw.data() + off, ph, stride, bpp)) { err = "bad PNG filter"; return false; }
            for (int y = 0; y < ph; ++y) emit(raw.data() + off + (size_t)y * (stride + 1) + 1, pw, yo[ps] + y * ys[ps], xo[ps], xs[ps]);
            off += (stride + 1) * (size_t)ph;
        }
    }
    return true;
}

} // namespace ptpng
