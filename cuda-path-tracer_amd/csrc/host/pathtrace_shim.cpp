/*
 * pathtrace_shim.cpp — pathtraceInit / pathtrace / pathtraceFree on top of the C ABI.
 * One static context, like the reference's file-static device pointers (src/pathtrace.cu:168-201).
 */
#include "pathtrace.h"

#include <cstdio>
#include <cstdlib>

namespace {
PtContext* g_ctx = nullptr;
/* pathtraceFree() parks the context instead of destroying it: the reference's preview frees and initialises again
 * whenever the camera moves (src/main.cpp:812-822), and a pathtraceInit() with a scene that differs from the parked
 * one in the camera's pose only takes the context back with pt_set_camera + pt_clear instead of building everything
 * anew.  A different scene destroys the parked context first; pathtraceRelease() (or PT_KEEP_CONTEXT=0) frees at once. */
PtContext* g_parked = nullptr;
unsigned long long g_fingerprint = 0;
GuiDataContainer* g_gui = nullptr;
float* g_hostImage = nullptr;
int g_sort = MATERIAL_SORTING;
int g_hostCopy = 1;
int g_lookahead = -1;           /* iterations rendered ahead of the calls; -1: PT_LOOKAHEAD or the default */
int g_ahead = 0;                /* what the context in use was set up for (0: one iteration per call) */
int g_traceDepth = 0, g_lastIteration = 0;
constexpr int kDefaultLookahead = 128;
constexpr int kRampStart = 8;   /* the first launch groups after an initialisation are small, so that the first frames arrive quickly */

bool keepContext()
{
    static const bool keep = [] { const char* e = getenv("PT_KEEP_CONTEXT"); return !(e && atoi(e) == 0); }();
    return keep;
}

/* iterations to render ahead when call `iteration` needs a new launch group: 8, 8, 8, 16, 16, 32, 32, ... up to the
 * depth the context was set up for — the largest power of two not above the iteration, at least kRampStart */
int aheadFor(int iteration, int most)
{
    int a = kRampStart;
    while (2 * a <= iteration && 2 * a <= most) a *= 2;
    return a < most ? a : most;
}

void check(int rc, const char* what)
{
    if (rc == PT_OK) return;
    /* checkCUDAError: print and leave (src/pathtrace.cu:27-48) */
    fprintf(stderr, "%s failed: %s\n", what, pt_last_error());
    exit(EXIT_FAILURE);
}
} // namespace

void InitDataContainer(GuiDataContainer* guiData) { g_gui = guiData; }

/* pathtrace() is called once per frame with iteration = 1, 2, 3, ... (src/main.cpp:812-837): the iterations after the one
 * asked for are rendered in the same launch group, `ahead` at a time, and handed out one per call (pt_trace_lookahead).
 * Two halves of the path-state rows take turns, so 2 * ahead rows are allocated; when the device cannot hold them the
 * look-ahead is halved until it can. */
static void setUpLookahead()
{
    int want = g_lookahead;
    if (want < 0) { const char* e = getenv("PT_LOOKAHEAD"); want = e ? atoi(e) : kDefaultLookahead; }
    g_ahead = 0;
    for (int a = want > 256 ? 256 : want; a >= 2; a /= 2)
        if (pt_set_option(g_ctx, PT_OPT_BATCH, 2 * a) == PT_OK) { g_ahead = a; break; }
    /* one stream per launch group: the rows ahead are rendered beside the calls that consume them, a host wait
     * inside a call (the library's trial of launch shapes) would stall the caller */
    check(pt_set_option(g_ctx, PT_OPT_LAUNCH_STREAMS, 1), "pathtraceInit");
}

void pathtraceInitDesc(const PtSceneDesc* desc, float* hostImage)
{
    pathtraceFree();
    const unsigned long long fp = pt_scene_fingerprint(desc);
    if (g_parked && fp == g_fingerprint && pt_set_camera(g_parked, &desc->camera) == PT_OK) {
        /* the same scene seen from another place: what pathtraceInit does to the image (cudaMemset, src/pathtrace.cu:447) */
        g_ctx = g_parked;
        g_parked = nullptr;
        check(pt_clear(g_ctx), "pathtraceInit");
        check(pt_reset_stats(g_ctx), "pathtraceInit");
    } else {
        if (g_parked) { pt_destroy(g_parked); g_parked = nullptr; }
        check(pt_create(desc, 0, &g_ctx), "pathtraceInit");
        setUpLookahead();
    }
    g_fingerprint = fp;
    check(pt_set_option(g_ctx, PT_OPT_MATERIAL_SORT, g_sort), "pathtraceInit");
    g_hostImage = hostImage;
    g_traceDepth = desc->traceDepth;
    g_lastIteration = desc->iterations;
    if (g_hostImage && g_hostCopy) check(pt_bind_host_image(g_ctx, g_hostImage), "pathtraceInit");
}

void pathtraceInit(Scene* scene)
{
    if (!scene) { fprintf(stderr, "pathtraceInit: null scene\n"); exit(EXIT_FAILURE); }
    pathtraceInitDesc(pt_scene_desc(scene), pt_scene_image(scene));
}

void pathtraceFree()
{
    if (g_ctx && keepContext()) {
        if (g_parked) pt_destroy(g_parked);
        check(pt_bind_host_image(g_ctx, nullptr), "pathtraceFree");     /* waits for the device; the caller's image memory is its own again */
        g_parked = g_ctx;
    } else if (g_ctx) pt_destroy(g_ctx);    /* unbinds the host image */
    g_ctx = nullptr;
    g_hostImage = nullptr;
}

void pathtraceRelease()
{
    pathtraceFree();
    if (g_parked) pt_destroy(g_parked);
    g_parked = nullptr;
    g_fingerprint = 0;
}

void pathtrace(uchar4* pbo, int /*frame*/, int iteration)
{
    if (!g_ctx) { fprintf(stderr, "pathtrace: pathtraceInit has not been called\n"); exit(EXIT_FAILURE); }
    if (g_gui) g_gui->TracedDepth = g_traceDepth;               /* src/pathtrace.cu:2712-2715 */
    /* iterations beyond the scene's ITERATIONS are never asked for by the reference's loop (src/main.cpp:831-837): none are rendered ahead */
    check(pt_trace_lookahead(g_ctx, iteration, g_ahead > 0 ? aheadFor(iteration, g_ahead) : 1, iteration <= g_lastIteration ? g_lastIteration : 0), "pathtrace");
    if (pbo) check(pt_display(g_ctx, iteration, reinterpret_cast<uint8_t*>(pbo)), "pathtrace");
    /* the running sum reaches the host image after every call (src/pathtrace.cu:2913-2917); mode 1 waits for it here, as the
     * reference's blocking copy does, mode 2 leaves the copy in flight until pathtraceHostImageWait() */
    if (g_hostCopy == 1 && g_hostImage) check(pt_host_image_wait(g_ctx), "pathtrace");
}

void pathtraceHostImageWait()
{
    if (g_ctx && g_hostImage) check(pt_host_image_wait(g_ctx), "pathtraceHostImageWait");
}

void pathtraceRange(int first, int lastInclusive)
{
    if (!g_ctx) { fprintf(stderr, "pathtraceRange: pathtraceInit has not been called\n"); exit(EXIT_FAILURE); }
    if (g_gui) g_gui->TracedDepth = g_traceDepth;
    check(pt_trace_range(g_ctx, first, lastInclusive + 1, 1), "pathtraceRange");
    if (g_hostCopy && g_hostImage) check(pt_download(g_ctx, g_hostImage), "pathtraceRange");
}

void pathtraceSetMaterialSorting(int on)
{
    g_sort = on ? 1 : 0;
    if (g_ctx) check(pt_set_option(g_ctx, PT_OPT_MATERIAL_SORT, g_sort), "pathtraceSetMaterialSorting");
}
void pathtraceSetHostCopy(int mode)
{
    g_hostCopy = mode < 0 ? 0 : mode > 2 ? 2 : mode;
    if (g_ctx) check(pt_bind_host_image(g_ctx, (g_hostCopy && g_hostImage) ? g_hostImage : nullptr), "pathtraceSetHostCopy");
}
void pathtraceSetLookahead(int iterations)
{
    g_lookahead = iterations;
    if (g_ctx) setUpLookahead();
}
PtContext* pathtraceContext() { return g_ctx; }
