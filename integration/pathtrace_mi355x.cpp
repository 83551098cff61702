/*
 * pathtrace_mi355x.cpp — the file a maintainer of the reference adds in place of src/pathtrace.cu
 * (+ intersections.cu, interactions.cu): the four functions of the reference's own src/pathtrace.h:8-11
 * on top of libmi_pathtrace.so.  It is written against the REFERENCE's headers (scene.h, sceneStructs.h,
 * utilities.h), not against this repository's host scene, and is compiled and run as such by
 * oracle/Makefile (target _ref/ref_render) wherever the reference tree is present, so it cannot rot.
 *
 * Layout facts it relies on (checked by the static_asserts below and by tests/test_host_side.py):
 * Material == PtMaterial (124 B), Triangle == PtTriangle (100 B), Camera == PtCamera (84 B).
 */
#include "pathtrace.h"              /* the reference's header, unchanged */
#include "scene.h"
#include <mi_pathtrace.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static_assert(sizeof(Material) == sizeof(PtMaterial) && sizeof(Triangle) == sizeof(PtTriangle) && sizeof(Camera) == sizeof(PtCamera),
              "the reference's PODs and the C ABI's must have one layout");

static PtContext* ctx = nullptr;
/* runCuda frees and initialises again whenever the camera moves (src/main.cpp:812-822).  Building the device scene takes
 * seconds for a large mesh, so pathtraceFree() parks the context, and a pathtraceInit() whose scene differs from the
 * parked one in the camera's pose only takes it back (pt_set_camera + pt_clear).  Another scene destroys the parked
 * context first; PT_KEEP_CONTEXT=0 makes pathtraceFree() free at once, as the reference does. */
static PtContext* parked = nullptr;
static unsigned long long parkedFingerprint = 0;
static Scene* hst_scene = nullptr;
static GuiDataContainer* guiData = nullptr;
static int lookahead = 0;               /* iterations rendered ahead of the calls (0: one per call) */
static bool strictHostCopy = true;      /* pathtrace() returns with scene->state.image up to date (PT_HOST_COPY=queued: see pathtrace) */

static void die(const char* what)
{
    /* checkCUDAError (src/pathtrace.cu:27-48): print and leave */
    fprintf(stderr, "%s failed: %s\n", what, pt_last_error());
    exit(EXIT_FAILURE);
}

void InitDataContainer(GuiDataContainer* g) { guiData = g; }

void pathtraceFree()
{
    static const bool keep = [] { const char* e = getenv("PT_KEEP_CONTEXT"); return !(e && atoi(e) == 0); }();
    if (ctx && keep) {
        if (parked) pt_destroy(parked);
        if (pt_bind_host_image(ctx, nullptr) != PT_OK) die("pathtraceFree");      /* waits for the device; the scene's image memory is its own again */
        parked = ctx;
    } else if (ctx) pt_destroy(ctx);
    ctx = nullptr;
}

/* iterations to render ahead when call `iter` needs a new launch group: 8, 8, 8, 16, 16, 32, 32, ... up to `most`, so that
 * the first frames after an initialisation (a camera move) arrive in milliseconds */
static int aheadFor(int iter, int most)
{
    int a = 8;
    while (2 * a <= iter && 2 * a <= most) a *= 2;
    return a < most ? a : most;
}

void pathtraceInit(Scene* scene)
{
    pathtraceFree();
    hst_scene = scene;
    std::vector<PtGeom> geoms;
    std::vector<PtMesh> meshes;
    size_t loadedModels = 0;            /* models the loader pushed onto scene->meshes so far, empty ones included */
    for (const Geom& g : scene->geoms) {
        PtGeom o;
        memset(&o, 0, sizeof(o));
        o.type = (int32_t)g.type;                       /* SPHERE, CUBE, GLTF_MESH in one order on both sides */
        o.materialid = g.type == GLTF_MESH ? -1 : g.materialid;      /* never assigned for glTF objects (src/scene.cpp:863-884) */
        memcpy(o.translation, &g.translation, 12);
        memcpy(o.rotation, &g.rotation, 12);
        memcpy(o.scale, &g.scale, 12);
        memcpy(o.transform, &g.transform, 64);
        memcpy(o.inverseTransform, &g.inverseTransform, 64);
        memcpy(o.invTranspose, &g.invTranspose, 64);
        o.meshIndex = -1;
        o.triangleStart = -1;
        o.bvhIndex = -1;
        /* Geom::meshData points at scene->meshes.back() of the moment it was loaded and dangles once a second model is
         * pushed (src/scene.cpp:535-537), so it is only looked at for being set: the loader sets it, and pushes a
         * MeshData, for every model file it could open — one without triangles included — and leaves it null when the
         * file would not load.  The mesh itself is taken by position, the n-th loaded model being scene->meshes[n]. */
        const bool loaded = g.type == GLTF_MESH && g.meshData != nullptr;
        const size_t modelIndex = loadedModels;
        if (loaded) ++loadedModels;
        if (loaded && modelIndex >= scene->meshes.size()) { fprintf(stderr, "pathtraceInit: scene holds %zu models, geom wants model %zu\n", scene->meshes.size(), modelIndex); exit(EXIT_FAILURE); }
        if (loaded && g.triangleCount > 0) {
            const MeshData& md = scene->meshes[modelIndex];
            if ((int)md.triangles.size() != g.triangleCount) { fprintf(stderr, "pathtraceInit: model %zu has %zu triangles, its geom says %d\n", modelIndex, md.triangles.size(), g.triangleCount); exit(EXIT_FAILURE); }
            PtMesh m;
            memset(&m, 0, sizeof(m));
            m.triangles = reinterpret_cast<const PtTriangle*>(md.triangles.data());
            m.triangleCount = (int32_t)md.triangles.size();
            /* nodes / triangleIndices stay null: pt_create runs BVH::build's algorithm itself.  A maintainer who keeps
             * the reference's BVH objects hands their node and index arrays over here instead (BVHNode == PtBvhNode). */
            o.meshIndex = (int32_t)meshes.size();
            o.triangleCount = m.triangleCount;
            meshes.push_back(m);
        }
        geoms.push_back(o);
    }
    std::vector<PtTexture> textures;
    for (const Texture& t : scene->textures) {          /* host pixels as loadGLTFTexture left them (src/scene.cpp:44-90) */
        PtTexture o;
        o.data = t.data; o.width = t.width; o.height = t.height; o.components = t.components;
        textures.push_back(o);
    }
    PtSceneDesc d;
    memset(&d, 0, sizeof(d));
    d.geoms = geoms.data(); d.geomCount = (int32_t)geoms.size();
    d.materials = reinterpret_cast<const PtMaterial*>(scene->materials.data()); d.materialCount = (int32_t)scene->materials.size();
    d.meshes = meshes.data(); d.meshCount = (int32_t)meshes.size();
    d.textures = textures.data(); d.textureCount = (int32_t)textures.size();
    /* HostEnvironmentMap::enabled / width / height are only written when the file loaded (src/scene.cpp:792-795) and
     * are indeterminate otherwise: the pixel vector says whether there is a map */
    const HostEnvironmentMap& e = scene->environmentMap;
    if (!e.data.empty()) {
        d.env.data = reinterpret_cast<const float*>(e.data.data());      /* already multiplied by the intensity (:810-817) */
        d.env.width = e.width; d.env.height = e.height; d.env.intensity = e.intensity; d.env.enabled = 1;
    }
    memcpy(&d.camera, &scene->state.camera, sizeof(PtCamera));
    d.traceDepth = scene->state.traceDepth;
    d.iterations = (int32_t)scene->state.iterations;
    const unsigned long long fingerprint = pt_scene_fingerprint(&d);
    if (parked && fingerprint == parkedFingerprint && pt_set_camera(parked, &d.camera) == PT_OK) {
        /* the same scene from another place: the context is kept, the image starts from zero (src/pathtrace.cu:447) */
        ctx = parked;
        parked = nullptr;
        if (pt_clear(ctx) != PT_OK) die("pathtraceInit");
    } else {
        if (parked) { pt_destroy(parked); parked = nullptr; }
        if (pt_create(&d, /* device */ 0, &ctx) != PT_OK) die("pathtraceInit");
        /* runCuda calls pathtrace(pbo, 0, ++iteration) once per frame (src/main.cpp:812-837).  The iterations behind the one
         * asked for are rendered ahead in the same launch group, up to 128 at a time (PT_LOOKAHEAD), and handed out one per
         * call: two halves of the path-state rows take turns, so rows for twice as many iterations are allocated, fewer
         * when the device cannot hold them. */
        const char* aheadEnv = getenv("PT_LOOKAHEAD");
        lookahead = 0;
        for (int a = aheadEnv ? atoi(aheadEnv) : 128; a >= 2; a /= 2)
            if (pt_set_option(ctx, PT_OPT_BATCH, 2 * (a > 256 ? 256 : a)) == PT_OK) { lookahead = a > 256 ? 256 : a; break; }
        /* one stream per launch group, fixed, so that no call waits inside the library for a measurement of launch shapes
         * (three streams measured slower behind the serve stream and the host copies: 1.32 against 1.10 ms per call;
         * PT_LOOKAHEAD_STREAMS sets another number) */
        const char* streamsEnv = getenv("PT_LOOKAHEAD_STREAMS");
        if (pt_set_option(ctx, PT_OPT_LAUNCH_STREAMS, streamsEnv ? atoi(streamsEnv) : 1) != PT_OK) die("pathtraceInit");
    }
    parkedFingerprint = fingerprint;
    if (pt_set_option(ctx, PT_OPT_MATERIAL_SORT, MATERIAL_SORTING) != PT_OK) die("pathtraceInit");
    /* hst_scene->state.image receives the running sum after every iteration (src/pathtrace.cu:2913-2917): the library
     * page-locks its memory where it lies if the runtime allows that, and otherwise copies through a pinned buffer of its
     * own in pieces; either way the copy is queued behind the call's addition to the sum */
    if (pt_bind_host_image(ctx, reinterpret_cast<float*>(scene->state.image.data())) != PT_OK) die("pathtraceInit");
    const char* copyEnv = getenv("PT_HOST_COPY");
    strictHostCopy = !(copyEnv && strcmp(copyEnv, "queued") == 0);
}

/* PT_HOST_COPY=queued: the call that must find scene->state.image complete (saveImage, src/main.cpp:292-318) makes this
 * one first; declared by the maintainer next to the reference's four functions */
void pathtraceHostImageWait()
{
    if (ctx && pt_host_image_wait(ctx) != PT_OK) die("pathtraceHostImageWait");
}

void pathtrace(uchar4* pbo, int /* frame */, int iter)
{
    if (!ctx) { fprintf(stderr, "pathtrace: pathtraceInit has not been called\n"); exit(EXIT_FAILURE); }
    if (guiData) guiData->TracedDepth = hst_scene->state.traceDepth;        /* src/pathtrace.cu:2712-2715 */
    const int last = (int)hst_scene->state.iterations;                      /* the loop saves and exits there (src/main.cpp:831-837) */
    if (pt_trace_lookahead(ctx, iter, lookahead > 0 ? aheadFor(iter, lookahead) : 1, iter <= last ? last : 0) != PT_OK) die("pathtrace");
    if (pbo && pt_display(ctx, iter, reinterpret_cast<uint8_t*>(pbo)) != PT_OK) die("pathtrace");      /* pbo: a device pointer, as the GL interop map yields */
    /* The running sum is back on the host when the call returns, as after the blocking copy at src/pathtrace.cu:2913-2917.
     * With PT_HOST_COPY=queued the copy of this call is only queued — it runs while the next call is being made, each copy
     * in call order — and the call that ends the render (the loop saves the image right after it, src/main.cpp:831-837)
     * waits; any other reader of scene->state.image calls pathtraceHostImageWait() first. */
    if ((strictHostCopy || iter >= last) && pt_host_image_wait(ctx) != PT_OK) die("pathtrace");
}
