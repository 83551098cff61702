/*
 * main_headless.cpp — the role of the reference's main.cpp without a window: load the scene, call
 * pathtraceInit once, pathtrace(nullptr, 0, iteration) for iteration = 1..N exactly as runCuda does
 * (src/main.cpp:812-837), write the image.
 *
 *   pt_headless scene.json [-n iterations] [-r W H] [-d depth] [-o out.pfm|out.ppm] [--batch]
 *
 * .pfm holds the running sum divided by the iteration count (float RGB, bottom row first as the
 * format demands); .ppm the 8-bit image saveImage() produces: samples averaged, clamped to [0,1],
 * x mirrored (src/main.cpp:351-375, src/image.cpp:23-50).
 */
#include "pathtrace.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static bool endsWith(const std::string& s, const char* suf)
{
    size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s scene.json [-n iterations] [-r W H] [-d depth] [-o out.pfm|out.ppm] [--batch]\n", argv[0]); return 2; }
    int iterations = -1, W = 0, H = 0, depth = 0;
    bool batch = false;
    std::string out;
    for (int i = 2; i < argc; ++i) {
        if (!strcmp(argv[i], "-n") && i + 1 < argc) iterations = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-r") && i + 2 < argc) { W = atoi(argv[++i]); H = atoi(argv[++i]); }
        else if (!strcmp(argv[i], "-d") && i + 1 < argc) depth = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-o") && i + 1 < argc) out = argv[++i];
        else if (!strcmp(argv[i], "--batch")) batch = true;
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    Scene* scene = nullptr;
    if (pt_scene_load(argv[1], &scene) != PT_OK) { fprintf(stderr, "%s\n", pt_last_error()); return 1; }
    if (W > 0 && H > 0 && pt_scene_set_resolution(scene, W, H) != PT_OK) { fprintf(stderr, "%s\n", pt_last_error()); return 1; }
    if (depth > 0 && pt_scene_set_depth(scene, depth) != PT_OK) { fprintf(stderr, "%s\n", pt_last_error()); return 1; }
    if (iterations < 0) iterations = pt_scene_iterations(scene);
    if (out.empty()) out = std::string(pt_scene_image_name(scene)) + ".pfm";

    GuiDataContainer gui;
    InitDataContainer(&gui);
    pathtraceFree();                 /* the reference frees before the first init as well (src/main.cpp:814) */
    pathtraceInit(scene);
    if (batch) {
        pathtraceRange(1, iterations);
    } else {
        pathtraceSetHostCopy(0);
        for (int it = 1; it <= iterations; ++it) {
            if (it == iterations) pathtraceSetHostCopy(1);
            pathtrace(nullptr, 0, it);
        }
    }
    const PtSceneDesc* d = pt_scene_desc(scene);
    const int w = d->camera.resolution[0], h = d->camera.resolution[1];
    const float* img = pt_scene_image(scene);
    FILE* f = fopen(out.c_str(), "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
    const float inv = iterations > 0 ? 1.0f / (float)iterations : 1.0f;
    if (endsWith(out, ".ppm")) {
        fprintf(f, "P6\n%d %d\n255\n", w, h);
        std::vector<unsigned char> row(3 * (size_t)w);
        for (int y = 0; y < h; ++y) {
            for (int x = 0; x < w; ++x) {
                const float* p = img + 3 * ((size_t)y * w + x);
                for (int k = 0; k < 3; ++k) {
                    float v = p[k] * inv;
                    v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
                    row[3 * (size_t)(w - 1 - x) + k] = (unsigned char)(v * 255.0f);
                }
            }
            fwrite(row.data(), 1, row.size(), f);
        }
    } else {
        fprintf(f, "PF\n%d %d\n-1.0\n", w, h);
        std::vector<float> row(3 * (size_t)w);
        for (int y = h - 1; y >= 0; --y) {
            for (size_t i = 0; i < row.size(); ++i) row[i] = img[3 * (size_t)y * w + i] * inv;
            fwrite(row.data(), sizeof(float), row.size(), f);
        }
    }
    fclose(f);
    printf("%s: %d iterations, %dx%d, traced depth %d -> %s\n", argv[1], iterations, w, h, gui.TracedDepth, out.c_str());
    pathtraceFree();
    pt_scene_free(scene);
    return 0;
}
