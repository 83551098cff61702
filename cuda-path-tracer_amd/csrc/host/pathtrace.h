/*
 * pathtrace.h — the renderer-facing surface of the wavefront loop, as C++ callers of the
 * reference see it (src/pathtrace.h:8-11):
 *
 *     void InitDataContainer(GuiDataContainer* guiData);
 *     void pathtraceInit(Scene* scene);
 *     void pathtraceFree();
 *     void pathtrace(uchar4* pbo, int frame, int iteration);
 *
 * Same names, same argument meaning, same error behaviour (a device error prints and exits,
 * src/pathtrace.cu:27-48), same ownership (the scene is borrowed and must outlive the calls,
 * src/pathtrace.cu:430; pathtraceFree() before any pathtraceInit() is a no-op, src/main.cpp:814).
 * Everything below these four functions goes through the C ABI of include/mi_pathtrace.h.
 *
 * Scene is this library's host scene (pt_scene_load); a renderer that keeps the reference's own
 * Scene class fills a PtSceneDesc from it instead and calls pathtraceInitDesc().
 */
#ifndef MI_PATHTRACE_SHIM_H
#define MI_PATHTRACE_SHIM_H

#include "../../../include/mi_pathtrace.h"

#define MATERIAL_SORTING 1          /* src/pathtrace.h:6; run-time switch: pathtraceSetMaterialSorting() */

typedef PtScene Scene;
struct uchar4 { unsigned char x, y, z, w; };
struct GuiDataContainer { int TracedDepth = 0; };   /* src/utilities.h / src/main.h: the one field the loop writes */

void InitDataContainer(GuiDataContainer* guiData);
void pathtraceInit(Scene* scene);
void pathtraceFree();
/* pbo: DEVICE pointer to W*H uchar4 or nullptr (headless); frame is unused as in the reference
 * (src/main.cpp:825-826).  After the call the scene's host image holds the running sum. */
void pathtrace(uchar4* pbo, int frame, int iteration);

/* additions for headless use */
void pathtraceInitDesc(const PtSceneDesc* desc, float* hostImage);     /* any scene source; hostImage may be null */
void pathtraceSetMaterialSorting(int on);
void pathtraceSetHostCopy(int on);          /* 0: skip the per-iteration device-to-host copy (default 1, as the reference) */
void pathtraceRange(int firstIteration, int lastIterationInclusive);    /* batched iterations, then one host copy */
PtContext* pathtraceContext();

#endif
