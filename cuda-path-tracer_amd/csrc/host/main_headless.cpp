/*
 * main_headless.cpp — the role of the reference's main.cpp without a window: load the scene, call
 * pathtraceInit once, pathtrace(nullptr, 0, iteration) for iteration = 1..N exactly as runCuda does
 * (src/main.cpp:812-837), write the image.
 *
 *   pt_headless scene.json [-n iterations] [-r W H] [-d depth] [-o out.png|.hdr|.ppm|.pfm] [--batch] [--denoise] [--gpus G]
 *
 * --gpus G splits the iterations over G contexts, one host thread each: context r renders iterations 1+r, 1+r+G, ...
 * of the same scene on device r (modulo the devices present, so the split can be rehearsed on one GPU), and the G
 * running sums are folded into the first (pt_reduce_sum).  The iterations are independent given their index, so the
 * union is the one-GPU render up to the order of the fp32 additions.
 *
 * .png / .ppm hold the 8-bit image saveImage() + Image::savePNG produce (samples averaged, x mirrored,
 * clamped to [0,1]; src/main.cpp:351-375, src/image.cpp:23-43), .hdr the Radiance file of Image::saveHDR
 * (src/image.cpp:45-50), .pfm the unmirrored float average (bottom row first as the format demands).
 * --denoise filters the result the way the reference's denoiser step does after the last iteration
 * (src/pathtrace.cu:2885-2910: guide buffers from the first hits, filtered image copied to the host image),
 * with the library's open a-trous filter in place of the closed OptiX one.
 * Without -o the image goes to <FILE of the camera block>.<iterations>samp.png, the reference's naming
 * without its start-time part.
 */
#include "pathtrace.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static bool endsWith(const std::string& s, const char* suf)
{
    size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s scene.json [-n iterations] [-r W H] [-d depth] [-o out.png|.hdr|.ppm|.pfm] [--batch] [--denoise] [--gpus G] [--restart-after K]\n", argv[0]); return 2; }
    int iterations = -1, W = 0, H = 0, depth = 0, gpus = 1, restartAfter = 0;
    bool batch = false, denoise = false;
    std::string out;
    for (int i = 2; i < argc; ++i) {
        if (!strcmp(argv[i], "-n") && i + 1 < argc) iterations = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-r") && i + 2 < argc) { W = atoi(argv[++i]); H = atoi(argv[++i]); }
        else if (!strcmp(argv[i], "-d") && i + 1 < argc) depth = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-o") && i + 1 < argc) out = argv[++i];
        else if (!strcmp(argv[i], "--batch")) batch = true;
        else if (!strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--denoise")) denoise = true;
        else if (!strcmp(argv[i], "--restart-after") && i + 1 < argc) restartAfter = atoi(argv[++i]);
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    Scene* scene = nullptr;
    if (pt_scene_load(argv[1], &scene) != PT_OK) { fprintf(stderr, "%s\n", pt_last_error()); return 1; }
    if (*pt_scene_log(scene)) fputs(pt_scene_log(scene), stderr);
    if (W > 0 && H > 0 && pt_scene_set_resolution(scene, W, H) != PT_OK) { fprintf(stderr, "%s\n", pt_last_error()); return 1; }
    if (depth > 0 && pt_scene_set_depth(scene, depth) != PT_OK) { fprintf(stderr, "%s\n", pt_last_error()); return 1; }
    if (iterations < 0) iterations = pt_scene_iterations(scene);
    /* the loop below asks for exactly these iterations: nothing beyond them is rendered ahead */
    if (iterations > 0 && pt_scene_set_iterations(scene, iterations) != PT_OK) { fprintf(stderr, "%s\n", pt_last_error()); return 1; }
    if (out.empty()) out = std::string(pt_scene_image_name(scene)) + "." + std::to_string(iterations) + "samp.png";

    GuiDataContainer gui;
    InitDataContainer(&gui);
    std::vector<PtContext*> split;       /* --gpus: one context per share of the iterations */
    if (gpus > 1) {
        const PtSceneDesc* desc = pt_scene_desc(scene);
        const int devices = pt_device_count();
        if (devices < 1) { fprintf(stderr, "no HIP device\n"); return 1; }
        if (devices < gpus) fprintf(stderr, "%d contexts share %d device(s)\n", gpus, devices);
        split.assign(gpus, nullptr);
        std::vector<int> status(gpus, PT_OK);
        std::vector<std::string> message(gpus);
        std::vector<std::thread> workers;
        for (int r = 0; r < gpus; ++r)
            workers.emplace_back([&, r] {
                /* the error text is kept per thread by the library */
                int rc = pt_create(desc, r % devices, &split[r]);
                const int share = (iterations - r + gpus - 1) / gpus;
                if (rc == PT_OK && share > 1) rc = pt_set_option(split[r], PT_OPT_BATCH, share < 64 ? share : 64);
                if (rc == PT_OK) rc = pt_trace_range(split[r], 1 + r, iterations + 1, gpus);
                if (rc == PT_OK) rc = pt_sync(split[r]);
                status[r] = rc;
                if (rc != PT_OK) message[r] = pt_last_error();
            });
        for (auto& w : workers) w.join();
        for (int r = 0; r < gpus; ++r)
            if (status[r] != PT_OK) { fprintf(stderr, "context %d: %s\n", r, message[r].c_str()); return 1; }
        if (pt_reduce_sum(split.data(), gpus, 0) != PT_OK || pt_download(split[0], pt_scene_image(scene)) != PT_OK) { fprintf(stderr, "%s\n", pt_last_error()); return 1; }
        gui.TracedDepth = desc->traceDepth;
    } else {
        pathtraceFree();                 /* the reference frees before the first init as well (src/main.cpp:814) */
        pathtraceInit(scene);
        if (restartAfter > 0) {
            /* what the reference's preview does when the camera has moved (src/main.cpp:812-822), without the move: K calls,
             * then iteration back to zero, pathtraceFree(), pathtraceInit() — the render that follows starts from nothing */
            for (int it = 1; it <= restartAfter; ++it) pathtrace(nullptr, 0, it);
            const unsigned long long before = pt_resource_calls();
            pathtraceFree();
            pathtraceInit(scene);
            fprintf(stderr, "restart after %d calls: %llu device resources created or released\n", restartAfter, pt_resource_calls() - before);
        }
        if (batch) {
            pathtraceRange(1, iterations);
        } else {
            /* the reference's loop: one call per frame; the host image is read once, after the last */
            pathtraceSetHostCopy(2);
            for (int it = 1; it <= iterations; ++it) pathtrace(nullptr, 0, it);
            pathtraceHostImageWait();
        }
    }
    PtContext* ctx0 = gpus > 1 ? split[0] : pathtraceContext();
    const PtSceneDesc* d = pt_scene_desc(scene);
    const int w = d->camera.resolution[0], h = d->camera.resolution[1];
    const float* img = pt_scene_image(scene);
    const float samples = (float)(iterations > 0 ? iterations : 1);
    if (denoise) {
        std::vector<float> filtered(3 * (size_t)w * h);
        if (pt_capture_gbuffer(ctx0, 1, nullptr, nullptr) != PT_OK ||
            pt_denoise(ctx0, samples, 5, 2.0f, 0.35f, 0.25f, filtered.data()) != PT_OK) { fprintf(stderr, "%s\n", pt_last_error()); return 1; }
        float* host = pt_scene_image(scene);
        for (size_t i = 0; i < filtered.size(); ++i) host[i] = filtered[i] * samples;      /* the host image holds sums */
    }
    int rc = PT_OK;
    if (endsWith(out, ".png") || endsWith(out, ".ppm")) {
        /* what saveImage() + Image::savePNG produce: averaged, x mirrored, clamped, 8 bit */
        std::vector<unsigned char> rgb(3 * (size_t)w * h);
        rc = pt_image_to_rgb8(img, w, h, samples, rgb.data());
        if (rc == PT_OK && endsWith(out, ".png")) rc = pt_write_png(out.c_str(), rgb.data(), w, h, 3);
        else if (rc == PT_OK) {
            FILE* f = fopen(out.c_str(), "wb");
            if (!f) { fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
            fprintf(f, "P6\n%d %d\n255\n", w, h);
            fwrite(rgb.data(), 1, rgb.size(), f);
            fclose(f);
        }
    } else if (endsWith(out, ".hdr")) {
        /* what saveImage() + Image::saveHDR produce */
        std::vector<float> avg(3 * (size_t)w * h);
        rc = pt_image_average(img, w, h, samples, 1, avg.data());
        if (rc == PT_OK) rc = pt_write_hdr(out.c_str(), avg.data(), w, h);
    } else {
        /* .pfm: the unmirrored average, for numeric comparison */
        std::vector<float> avg(3 * (size_t)w * h);
        rc = pt_image_average(img, w, h, samples, 0, avg.data());
        FILE* f = fopen(out.c_str(), "wb");
        if (!f) { fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
        fprintf(f, "PF\n%d %d\n-1.0\n", w, h);
        for (int y = h - 1; y >= 0; --y) fwrite(avg.data() + 3 * (size_t)y * w, sizeof(float), 3 * (size_t)w, f);
        fclose(f);
    }
    if (rc != PT_OK) { fprintf(stderr, "%s\n", pt_last_error()); return 1; }
    printf("%s: %d iterations, %dx%d, traced depth %d, %d context(s) -> %s\n", argv[1], iterations, w, h, gui.TracedDepth, gpus, out.c_str());
    for (PtContext* c : split) pt_destroy(c);
    pathtraceRelease();
    pt_scene_free(scene);
    return 0;
}
