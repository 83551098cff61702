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
/* Releases the caller's side (the host image is unbound, the device is idle) but keeps the context parked: the next
 * pathtraceInit() with a scene that differs in the camera's pose only — what the reference's preview does on every camera
 * move, src/main.cpp:812-822 — takes it back without rebuilding anything.  pathtraceRelease() frees the device memory too. */
void pathtraceFree();
void pathtraceRelease();
/* pbo: DEVICE pointer to W*H uchar4 or nullptr (headless); frame is unused as in the reference
 * (src/main.cpp:825-826).  After the call the scene's host image holds the running sum.
 * The reference's loop calls this once per frame with iteration = 1, 2, 3, ...: the iterations behind the one asked
 * for are rendered ahead in the same launch group (pathtraceSetLookahead) and every later call only adds its own, so
 * the image after call k is bit for bit the sum of iterations 1..k while the device works at the rate of launch groups.
 * The first groups after an initialisation are small (8, 8, 8, 16, 16, 32, ... iterations) so that the first frames after a
 * camera move arrive in milliseconds; the full depth is reached after about 250 calls. */
void pathtrace(uchar4* pbo, int frame, int iteration);

/* additions for headless use */
void pathtraceInitDesc(const PtSceneDesc* desc, float* hostImage);     /* any scene source; hostImage may be null */
void pathtraceSetMaterialSorting(int on);
/* the per-iteration copy of the running sum into the scene's host image (src/pathtrace.cu:2913-2917):
 * 1 (default, as the reference) every pathtrace() returns with the host image up to date; 2 the copy is queued behind the
 * call and pathtraceHostImageWait() waits for the latest one (before saveImage reads the image); 0 no copy */
void pathtraceSetHostCopy(int mode);
void pathtraceHostImageWait();
/* iterations rendered ahead of the calls (default 128, environment PT_LOOKAHEAD; 0 or 1: one iteration per call).  Path
 * state for twice as many iterations is allocated, 203 bytes per pixel each. */
void pathtraceSetLookahead(int iterations);
void pathtraceRange(int firstIteration, int lastIterationInclusive);    /* batched iterations, then one host copy */
PtContext* pathtraceContext();

#endif
