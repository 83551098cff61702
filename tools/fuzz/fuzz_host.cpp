// Host-only robustness run: random corruptions of valid GLB / PNG / HDR / JSON inputs through the loaders, under ASan+UBSan.
#include "../../include/mi_pathtrace.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>
static std::vector<uint8_t> slurp(const char* p) { std::ifstream f(p, std::ios::binary); return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), {}); }
static void spit(const char* p, const std::vector<uint8_t>& d) { std::ofstream f(p, std::ios::binary); f.write((const char*)d.data(), d.size()); }
int main(int argc, char** argv)
{
    if (argc < 5) return 2;
    std::vector<uint8_t> glb = slurp(argv[1]), png = slurp(argv[2]), scene = slurp(argv[3]);
    int rounds = atoi(argv[4]);
    std::mt19937 rng(argc > 6 ? atoi(argv[6]) : 12345);
    const float t[3] = {0, 0, 0}, s[3] = {1, 1, 1};
    int okGlb = 0, okPng = 0, okScene = 0, okJpg = 0;
    std::vector<uint8_t> jpg = argc > 7 ? slurp(argv[7]) : std::vector<uint8_t>();
    std::string dir = argv[5];
    for (int r = 0; r < rounds; ++r) {
        auto mutate = [&](std::vector<uint8_t> d) {
            int n = 1 + rng() % 4;
            for (int k = 0; k < n; ++k) {
                size_t at = rng() % d.size();
                switch (rng() % 4) {
                case 0: d[at] ^= (uint8_t)(1u << (rng() % 8)); break;
                case 1: d[at] = (uint8_t)rng(); break;
                case 2: d.resize(at + 1); break;
                default: { uint32_t v = (rng() % 2) ? 0xFFFFFFFFu : (uint32_t)rng(); if (at + 4 <= d.size()) memcpy(&d[at], &v, 4); }
                }
            }
            return d;
        };
        {   // PNG
            std::vector<uint8_t> d = mutate(png);
            int w, h;
            std::vector<uint8_t> out(1 << 22);
            if (pt_decode_png(d.data(), d.size(), out.data(), out.size(), &w, &h) == PT_OK) ++okPng;
        }
        if (!jpg.empty()) {   // JPEG
            std::vector<uint8_t> d = mutate(jpg);
            int w, h;
            std::vector<uint8_t> out(1 << 22);
            if (pt_decode_jpeg(d.data(), d.size(), nullptr, 0, &w, &h) == PT_OK && (size_t)w * h * 4 <= out.size() &&
                pt_decode_jpeg(d.data(), d.size(), out.data(), out.size(), &w, &h) == PT_OK) ++okJpg;
        }
        {   // GLB (mutations mostly in the first 3000 bytes: header + JSON)
            std::vector<uint8_t> d = glb;
            if (rng() % 2) { size_t keep = d.size(); d.resize(3000 < keep ? 3000 : keep); d = mutate(d); d.insert(d.end(), glb.begin() + (3000 < keep ? 3000 : keep), glb.end()); }
            else d = mutate(d);
            std::string p = dir + "/m.glb";
            spit(p.c_str(), d);
            PtScene* sc = nullptr;
            std::string sp = dir + "/base.json";
            if (pt_scene_load(sp.c_str(), &sc) != PT_OK) { fprintf(stderr, "base scene failed: %s\n", pt_last_error()); return 1; }
            if (pt_scene_import_gltf(sc, p.c_str(), t, t, s) >= 0) ++okGlb;
            pt_scene_free(sc);
        }
        {   // scene JSON
            std::vector<uint8_t> d = mutate(scene);
            std::string p = dir + "/m.json";
            spit(p.c_str(), d);
            PtScene* sc = nullptr;
            if (pt_scene_load(p.c_str(), &sc) == PT_OK) { ++okScene; pt_scene_free(sc); }
        }
    }
    printf("rounds %d: png ok %d, jpeg ok %d, glb ok %d, scene ok %d\n", rounds, okPng, okJpg, okGlb, okScene);
    return 0;
}
