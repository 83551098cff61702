/*
 * image_io.cpp — the output side of a render: running sum -> 8-bit image, PNG and Radiance .hdr writers.
 * No GPU calls.
 *
 * saveImage() (src/main.cpp:351-375) divides the running sum by the sample count, mirrors x and hands the
 * image to Image::savePNG (src/image.cpp:23-43), which clamps to [0,1], scales by 255 and truncates; the
 * optional Image::saveHDR (src/image.cpp:45-50) writes the averaged floats through stb_image_write, whose
 * float -> RGBE step (external/include/stb_image_write.h:639-657) is restated in toRgbe().  File encodings
 * are this library's own (PNG: filter 0, one zlib stream of fixed-Huffman blocks with a small LZ77 matcher;
 * .hdr: run-length scanlines) — any PNG / Radiance reader yields the same pixels the reference's files hold.
 */
#include "../../../include/mi_pathtrace.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

void pt_set_error(const char* fmt, ...);

namespace {

/* ---- zlib stream: fixed-Huffman deflate with a hash-table LZ77 matcher ---- */
class Deflater {
public:
    explicit Deflater(std::vector<uint8_t>& sink) : out(sink) {}
    void run(const uint8_t* d, size_t n)
    {
        out.push_back(0x78); out.push_back(0x01);
        put(1, 1);              /* final block */
        put(1, 2);              /* fixed Huffman */
        std::vector<int32_t> head(1 << 15, -1);
        size_t i = 0;
        while (i < n) {
            size_t bestLen = 0, bestDist = 0;
            if (i + 3 <= n) {
                uint32_t h = ((uint32_t)d[i] * 506832829u ^ (uint32_t)d[i + 1] * 2654435761u ^ (uint32_t)d[i + 2] * 40503u) >> 7 & 0x7FFF;
                int32_t cand = head[h];
                head[h] = (int32_t)i;
                if (cand >= 0 && i - (size_t)cand <= 32768) {
                    size_t l = 0, lim = n - i < 258 ? n - i : 258;
                    while (l < lim && d[cand + l] == d[i + l]) ++l;
                    if (l >= 3) { bestLen = l; bestDist = i - (size_t)cand; }
                }
            }
            if (bestLen) { match(bestLen, bestDist); i += bestLen; }
            else literal(d[i++]);
        }
        symbol(256);
        if (nbits) { out.push_back((uint8_t)acc); acc = 0; nbits = 0; }
        uint32_t a = 1, b = 0;
        for (size_t k = 0; k < n; ++k) { a = (a + d[k]) % 65521u; b = (b + a) % 65521u; }
        uint32_t ad = (b << 16) | a;
        for (int s = 24; s >= 0; s -= 8) out.push_back((uint8_t)(ad >> s));
    }

private:
    std::vector<uint8_t>& out;
    uint32_t acc = 0;
    int nbits = 0;

    void put(uint32_t v, int n)      /* LSB first */
    {
        acc |= v << nbits;
        nbits += n;
        while (nbits >= 8) { out.push_back((uint8_t)acc); acc >>= 8; nbits -= 8; }
    }
    void putCode(uint32_t code, int n)  /* Huffman codes go MSB first */
    {
        uint32_t r = 0;
        for (int i = 0; i < n; ++i) r |= ((code >> i) & 1u) << (n - 1 - i);
        put(r, n);
    }
    void symbol(int s)
    {
        if (s < 144) putCode(0x30 + s, 8);
        else if (s < 256) putCode(0x190 + (s - 144), 9);
        else if (s < 280) putCode(s - 256, 7);
        else putCode(0xC0 + (s - 280), 8);
    }
    void literal(uint8_t v) { symbol(v); }
    void match(size_t len, size_t dist)
    {
        static const uint16_t lbase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
        static const uint8_t lext[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
        static const uint16_t dbase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
        static const uint8_t dext[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
        int li = 28;
        while (lbase[li] > len) --li;
        symbol(257 + li);
        if (lext[li]) put((uint32_t)(len - lbase[li]), lext[li]);
        int di = 29;
        while (dbase[di] > dist) --di;
        putCode((uint32_t)di, 5);
        if (dext[di]) put((uint32_t)(dist - dbase[di]), dext[di]);
    }
};

uint32_t crc32(const uint8_t* p, size_t n, uint32_t crc = 0)
{
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}

void chunk(std::vector<uint8_t>& f, const char tag[4], const uint8_t* body, size_t n)
{
    for (int s = 24; s >= 0; s -= 8) f.push_back((uint8_t)((uint32_t)n >> s));
    size_t at = f.size();
    f.insert(f.end(), tag, tag + 4);
    if (n) f.insert(f.end(), body, body + n);
    uint32_t c = crc32(f.data() + at, 4 + n);
    for (int s = 24; s >= 0; s -= 8) f.push_back((uint8_t)(c >> s));
}

bool writeAll(const char* path, const uint8_t* p, size_t n)
{
    FILE* f = fopen(path, "wb");
    if (!f) return false;
    bool ok = fwrite(p, 1, n, f) == n;
    return (fclose(f) == 0) && ok;
}

/* external/include/stb_image_write.h:639-657 */
void toRgbe(uint8_t rgbe[4], const float* linear)
{
    float m = linear[1] > linear[2] ? linear[1] : linear[2];
    float maxcomp = linear[0] > m ? linear[0] : m;
    if (maxcomp < 1e-32f) { rgbe[0] = rgbe[1] = rgbe[2] = rgbe[3] = 0; return; }
    int exponent;
    float normalize = (float)frexp(maxcomp, &exponent) * 256.0f / maxcomp;
    rgbe[0] = (uint8_t)(linear[0] * normalize);
    rgbe[1] = (uint8_t)(linear[1] * normalize);
    rgbe[2] = (uint8_t)(linear[2] * normalize);
    rgbe[3] = (uint8_t)(exponent + 128);
}

} // namespace

extern "C" {

int pt_image_average(const float* sum, int width, int height, float samples, int mirror_x, float* out)
{
    if (!sum || !out || width <= 0 || height <= 0) { pt_set_error("pt_image_average: bad argument"); return PT_EINVAL; }
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            const float* p = sum + 3 * ((size_t)y * width + x);
            float* o = out + 3 * ((size_t)y * width + (mirror_x ? width - 1 - x : x));
            o[0] = p[0] / samples; o[1] = p[1] / samples; o[2] = p[2] / samples;
        }
    return PT_OK;
}

int pt_image_to_rgb8(const float* sum, int width, int height, float samples, uint8_t* rgb)
{
    if (!sum || !rgb || width <= 0 || height <= 0) { pt_set_error("pt_image_to_rgb8: bad argument"); return PT_EINVAL; }
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            const float* p = sum + 3 * ((size_t)y * width + x);
            uint8_t* o = rgb + 3 * ((size_t)y * width + (width - 1 - x));
            for (int k = 0; k < 3; ++k) {
                float v = p[k] / samples;
                /* glm::clamp(v, 0, 1) = min(max(v, 0), 1); a NaN sample comes out as 0 here (its cast is undefined in the reference) */
                float lo = v > 0.0f ? v : 0.0f;
                float c = lo < 1.0f ? lo : 1.0f;
                o[k] = (uint8_t)(c * 255.f);
            }
        }
    return PT_OK;
}

int pt_write_png(const char* path, const uint8_t* pixels, int width, int height, int components)
{
    if (!path || !pixels || width <= 0 || height <= 0 || (components != 1 && components != 3 && components != 4)) { pt_set_error("pt_write_png: bad argument"); return PT_EINVAL; }
    const size_t stride = (size_t)width * components;
    std::vector<uint8_t> raw((stride + 1) * (size_t)height);
    for (int y = 0; y < height; ++y) {
        raw[(size_t)y * (stride + 1)] = 0;
        memcpy(&raw[(size_t)y * (stride + 1) + 1], pixels + (size_t)y * stride, stride);
    }
    std::vector<uint8_t> z;
    Deflater(z).run(raw.data(), raw.size());
    std::vector<uint8_t> f = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    uint8_t hdr[13];
    for (int s = 0; s < 4; ++s) { hdr[s] = (uint8_t)((uint32_t)width >> (24 - 8 * s)); hdr[4 + s] = (uint8_t)((uint32_t)height >> (24 - 8 * s)); }
    hdr[8] = 8; hdr[9] = components == 1 ? 0 : components == 3 ? 2 : 6; hdr[10] = hdr[11] = hdr[12] = 0;
    chunk(f, "IHDR", hdr, 13);
    chunk(f, "IDAT", z.data(), z.size());
    chunk(f, "IEND", nullptr, 0);
    if (!writeAll(path, f.data(), f.size())) { pt_set_error("pt_write_png: cannot write %s", path); return PT_EIO; }
    return PT_OK;
}

int pt_write_hdr(const char* path, const float* rgb, int width, int height)
{
    if (!path || !rgb || width <= 0 || height <= 0) { pt_set_error("pt_write_hdr: bad argument"); return PT_EINVAL; }
    char head[160];
    int n = snprintf(head, sizeof(head), "#?RADIANCE\n# Written by libmi_pathtrace\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=          1.0000000000000\n\n-Y %d +X %d\n", height, width);
    std::vector<uint8_t> f(head, head + n);
    std::vector<uint8_t> line((size_t)width * 4);
    for (int y = 0; y < height; ++y) {
        for (int x = 0; x < width; ++x) toRgbe(&line[(size_t)x * 4], rgb + 3 * ((size_t)y * width + x));
        if (width < 8 || width >= 32768) { f.insert(f.end(), line.begin(), line.end()); continue; }   /* flat scanline */
        /* run-length scanline: marker, then the four byte planes, each as runs (128+n, value) and literal stretches (n, bytes) */
        f.push_back(2); f.push_back(2); f.push_back((uint8_t)(width >> 8)); f.push_back((uint8_t)(width & 0xFF));
        for (int k = 0; k < 4; ++k) {
            int x = 0;
            while (x < width) {
                int r = x;              /* start of the next run of at least three equal bytes */
                while (r < width) {
                    int e = r + 1;
                    while (e < width && e - r < 127 && line[(size_t)e * 4 + k] == line[(size_t)r * 4 + k]) ++e;
                    if (e - r >= 3) break;
                    r = e;
                }
                while (x < r) {         /* literals up to the run */
                    int len = r - x < 128 ? r - x : 128;
                    f.push_back((uint8_t)len);
                    for (int i = 0; i < len; ++i) f.push_back(line[(size_t)(x + i) * 4 + k]);
                    x += len;
                }
                if (r < width) {
                    int e = r + 1;
                    while (e < width && e - r < 127 && line[(size_t)e * 4 + k] == line[(size_t)r * 4 + k]) ++e;
                    f.push_back((uint8_t)(128 + (e - r)));
                    f.push_back(line[(size_t)r * 4 + k]);
                    x = e;
                }
            }
        }
    }
    if (!writeAll(path, f.data(), f.size())) { pt_set_error("pt_write_hdr: cannot write %s", path); return PT_EIO; }
    return PT_OK;
}

} // extern "C"
