/*
 * pathtrace_shim.cpp — pathtraceInit / pathtrace / pathtraceFree on top of the C ABI.
 * One static context, like the reference's file-static device pointers (src/pathtrace.cu:168-201).
 */
#include "pathtrace.h"

#include <cstdio>
#include <cstdlib>

namespace {
PtContext* g_ctx = nullptr;
GuiDataContainer* g_gui = nullptr;
float* g_hostImage = nullptr;
int g_sort = MATERIAL_SORTING;
int g_hostCopy = 1;

void check(int rc, const char* what)
{
    if (rc == PT_OK) return;
    /* checkCUDAError: print and leave (src/pathtrace.cu:27-48) */
    fprintf(stderr, "%s failed: %s\n", what, pt_last_error());
    exit(EXIT_FAILURE);
}
} // namespace

void InitDataContainer(GuiDataContainer* guiData) { g_gui = guiData; }

void pathtraceInitDesc(const PtSceneDesc* desc, float* hostImage)
{
    pathtraceFree();
    check(pt_create(desc, 0, &g_ctx), "pathtraceInit");
    check(pt_set_option(g_ctx, PT_OPT_MATERIAL_SORT, g_sort), "pathtraceInit");
    g_hostImage = hostImage;
}

void pathtraceInit(Scene* scene)
{
    if (!scene) { fprintf(stderr, "pathtraceInit: null scene\n"); exit(EXIT_FAILURE); }
    pathtraceInitDesc(pt_scene_desc(scene), pt_scene_image(scene));
}

void pathtraceFree()
{
    if (g_ctx) pt_destroy(g_ctx);
    g_ctx = nullptr;
    g_hostImage = nullptr;
}

static void afterTrace(uchar4* pbo, int iteration)
{
    if (pbo) check(pt_display(g_ctx, iteration, reinterpret_cast<uint8_t*>(pbo)), "pathtrace");
    if (g_gui) {
        PtStats st;
        check(pt_stats(g_ctx, &st), "pathtrace");
        g_gui->TracedDepth = (int)st.lastDepth;                 /* src/pathtrace.cu:2712-2715 */
    }
    if (g_hostCopy && g_hostImage) check(pt_download(g_ctx, g_hostImage), "pathtrace");   /* src/pathtrace.cu:2913-2917 */
}

void pathtrace(uchar4* pbo, int /*frame*/, int iteration)
{
    if (!g_ctx) { fprintf(stderr, "pathtrace: pathtraceInit has not been called\n"); exit(EXIT_FAILURE); }
    check(pt_trace(g_ctx, iteration), "pathtrace");
    afterTrace(pbo, iteration);
}

void pathtraceRange(int first, int lastInclusive)
{
    if (!g_ctx) { fprintf(stderr, "pathtraceRange: pathtraceInit has not been called\n"); exit(EXIT_FAILURE); }
    check(pt_trace_range(g_ctx, first, lastInclusive + 1, 1), "pathtraceRange");
    afterTrace(nullptr, lastInclusive);
}

void pathtraceSetMaterialSorting(int on)
{
    g_sort = on ? 1 : 0;
    if (g_ctx) check(pt_set_option(g_ctx, PT_OPT_MATERIAL_SORT, g_sort), "pathtraceSetMaterialSorting");
}
void pathtraceSetHostCopy(int on) { g_hostCopy = on ? 1 : 0; }
PtContext* pathtraceContext() { return g_ctx; }
