/*
 * ref_render — headless stand-in for the reference's main.cpp around pathtrace_mi355x.cpp: the REFERENCE's scene
 * loader (Scene::Scene, linked from the reference's own scene.cpp) feeds the reference-shaped pathtraceInit /
 * pathtrace / pathtraceFree, which run on libmi_pathtrace.so.  Test infrastructure (built by oracle/Makefile into
 * oracle/_ref/, only where the reference tree is present).
 *
 *   ref_render <scene.json> <iterations> <out.bin> [width height]
 * writes width, height (int32) and the W*H*3 float running sum.
 */
#include "pathtrace.h"
#include "scene.h"

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

struct uchar4;
void pathtraceHostImageWait();      /* integration/pathtrace_mi355x.cpp */
extern "C" unsigned long long pt_resource_calls(void);      /* include/mi_pathtrace.h */

/* FNV-1a over a block of loader output: two runs that print different values were not given the same scene */
static uint64_t fnv(const void* p, size_t n, uint64_t h = 1469598103934665603ull)
{
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char** argv)
{
    if (argc < 4) { fprintf(stderr, "usage: ref_render scene.json iterations out.bin [width height]\n"); return 2; }
    Scene* scene = new Scene(std::string(argv[1]));
    const int iterations = atoi(argv[2]);
    Camera& cam = scene->state.camera;
    if (argc >= 6) {
        /* what a changed RES entry does in the loader (src/scene.cpp:886-917) */
        cam.resolution.x = atoi(argv[4]);
        cam.resolution.y = atoi(argv[5]);
        float fovy = cam.fov.y;
        float yscaled = tan(fovy * (PI / 180));
        float xscaled = (yscaled * cam.resolution.x) / cam.resolution.y;
        float fovx = (atan(xscaled) * 180) / PI;
        cam.fov = glm::vec2(fovx, fovy);
        cam.pixelLength = glm::vec2(2 * xscaled / (float)cam.resolution.x, 2 * yscaled / (float)cam.resolution.y);
        scene->state.image.assign((size_t)cam.resolution.x * cam.resolution.y, glm::vec3());
    }
    /* main() and the first runCuda() of the reference (src/main.cpp:750-769, :788-807): the camera the kernels see */
    glm::vec3 view = cam.view;
    glm::vec3 viewXZ = glm::vec3(view.x, 0.0f, view.z);
    glm::vec3 viewZY = glm::vec3(0.0f, view.y, view.z);
    float phi = glm::acos(glm::dot(glm::normalize(viewXZ), glm::vec3(0, 0, -1)));
    float theta = glm::acos(glm::dot(glm::normalize(viewZY), glm::vec3(0, 1, 0)));
    float zoom = glm::length(cam.position - cam.lookAt);
    glm::vec3 cameraPosition;
    cameraPosition.x = zoom * sin(phi) * sin(theta);
    cameraPosition.y = zoom * cos(theta);
    cameraPosition.z = zoom * cos(phi) * sin(theta);
    cam.view = -glm::normalize(cameraPosition);
    glm::vec3 v = cam.view;
    glm::vec3 u = glm::vec3(0, 1, 0);
    glm::vec3 r = glm::cross(v, u);
    cam.up = glm::cross(r, v);
    cam.right = r;
    cam.position = cameraPosition;
    cameraPosition += cam.lookAt;
    cam.position = cameraPosition;

    if (getenv("REF_RENDER_PIN_MATERIALS")) {
        /* The loader's loadGLTFMaterial fills a `Material mat;` it never zeroes (src/scene.cpp:96): index of refraction, the
         * specular block and the subsurface fields of a glTF material are whatever the stack held.  For timing runs that
         * have to be comparable, a material whose subsurface switch is neither 0 nor 1, or whose index of refraction is no
         * plausible number, gets those fields set (no subsurface, 1.5), as this repository's own loader sets them. */
        int pinned = 0;
        for (Material& m : scene->materials) {
            const bool wild = (m.subsurfaceEnabled != 0 && m.subsurfaceEnabled != 1) || !(m.indexOfRefraction >= 1.0f && m.indexOfRefraction <= 4.0f);
            if (!wild) continue;
            m.specular.exponent = 0.0f; m.specular.color = glm::vec3(0.0f);
            m.subsurfaceColor = glm::vec3(0.0f); m.subsurfaceRadius = 0.0f; m.subsurfaceRadiusRGB = glm::vec3(0.0f);
            m.subsurfaceScale = 0.0f; m.subsurfaceAnisotropy = 0.0f; m.subsurfaceEnabled = 0;
            m.indexOfRefraction = 1.5f;
            ++pinned;
        }
        fprintf(stderr, "ref_render: %d materials had uninitialised fields pinned\n", pinned);
    }
    {
        uint64_t hm = fnv(scene->materials.data(), scene->materials.size() * sizeof(Material));
        uint64_t hg = fnv(scene->geoms.data(), scene->geoms.size() * sizeof(Geom));
        uint64_t ht = 1469598103934665603ull;
        for (const MeshData& md : scene->meshes) ht = fnv(md.triangles.data(), md.triangles.size() * sizeof(Triangle), ht);
        uint64_t he = fnv(scene->environmentMap.data.data(), scene->environmentMap.data.size() * sizeof(glm::vec3));
        uint64_t hc = fnv(&cam, sizeof(Camera));
        fprintf(stderr, "ref_render: loader output hashes: materials %016llx, geoms %016llx (these hold a pointer), triangles %016llx, environment %016llx, camera %016llx\n",
                (unsigned long long)hm, (unsigned long long)hg, (unsigned long long)ht, (unsigned long long)he, (unsigned long long)hc);
    }
    GuiDataContainer gui;
    InitDataContainer(&gui);
    pathtraceFree();                                    /* src/main.cpp:814 */
    pathtraceInit(scene);
    /* the loop of runCuda (src/main.cpp:812-837): one call per frame; wall time per call, the first call (which also
     * carries the one-off set-up of the first launch groups) apart */
    double firstMs = 0.0;
    auto t0 = std::chrono::steady_clock::now();
    auto tHalf = t0;
    for (int it = 1; it <= iterations; ++it) {
        pathtrace(nullptr, 0, it);
        if (it == 1) {
            auto t1 = std::chrono::steady_clock::now();
            firstMs = std::chrono::duration<double, std::milli>(t1 - t0).count();
            t0 = t1;
        }
        if (it == iterations / 2) tHalf = std::chrono::steady_clock::now();
    }
    const double lastHalfMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tHalf).count();
    pathtraceHostImageWait();
    double restMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (const char* at = getenv("REF_RENDER_REINIT_AT")) {
        /* what runCuda does when the camera has moved (src/main.cpp:812-822): iteration back to 0, pathtraceFree(),
         * pathtraceInit() — here after the calls above, with the camera moved by REF_RENDER_MOVE="dx,dy,dz"; then
         * atoi(REF_RENDER_REINIT_AT) calls from iteration 1 again, whose sum is what gets written */
        float mv[3] = { 0.5f, 0.25f, 0.0f };
        if (const char* m = getenv("REF_RENDER_MOVE")) sscanf(m, "%f,%f,%f", &mv[0], &mv[1], &mv[2]);
        cam.position += glm::vec3(mv[0], mv[1], mv[2]);
        const unsigned long long before = pt_resource_calls();
        auto r0 = std::chrono::steady_clock::now();
        pathtraceFree();
        pathtraceInit(scene);
        pathtrace(nullptr, 0, 1);
        const double reinitMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r0).count();
        fprintf(stderr, "ref_render: camera moved; pathtraceFree + pathtraceInit + first pathtrace() call %.3f ms, %llu device resources created or released\n",
                reinitMs, pt_resource_calls() - before);
        const int again = atoi(at);
        for (int it = 2; it <= again; ++it) pathtrace(nullptr, 0, it);
        pathtraceHostImageWait();
    }
    const char* copyEnv = getenv("PT_HOST_COPY");
    const bool queued = copyEnv && std::string(copyEnv) == "queued";
    FILE* f = fopen(argv[3], "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", argv[3]); return 2; }
    int32_t wh[2] = { cam.resolution.x, cam.resolution.y };
    fwrite(wh, 4, 2, f);
    fwrite(scene->state.image.data(), 12, scene->state.image.size(), f);
    fclose(f);
    fprintf(stderr, "ref_render: %d iterations at %dx%d, traced depth %d, first pathtrace() call %.3f ms, the other %d calls %.4f ms each, the last %d calls %.4f ms each (%s)\n",
            iterations, cam.resolution.x, cam.resolution.y, gui.TracedDepth, firstMs, iterations - 1, iterations > 1 ? restMs / (iterations - 1) : 0.0,
            iterations - iterations / 2, iterations > 1 ? lastHalfMs / (iterations - iterations / 2) : 0.0,
            queued ? "host copies queued in call order, the last one waited for" : "host image up to date after every call");
    pathtraceFree();
    return 0;
}
