/*
 * pt_oracle.h — TEST INFRASTRUCTURE.  C interface of the CPU oracle: a plain
 * C++ restatement of the reference's wavefront loop (src/pathtrace.cu,
 * src/intersections.cu, src/interactions.cu, src/bvh.cpp).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * shipped library never does.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include "../include/mi_pathtrace.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct Oracle Oracle;

enum OracleMath {
    ORACLE_MATH_LIBM = 0,   /* glibc sinf/cosf/...: what a host build of the reference computes */
    ORACLE_MATH_DET = 1     /* det_math.h: bit-identical to the GPU kernels' arithmetic         */
};

/* oracle-only option, beside the PtOption values: draw the three scatter offsets of the PBR transmission branch
 * (src/pathtrace.cu:1786-1790, a constructor's argument list) in the order z, y, x, as a g++ host build evaluates it */
enum { ORACLE_OPT_SCATTER_RIGHT_TO_LEFT = 100 };

Oracle* oracle_create(const PtSceneDesc* scene);
void oracle_destroy(Oracle* o);
void oracle_set_math(Oracle* o, int mode);
void oracle_set_option(Oracle* o, int option, int value);   /* PtOption: sort / use_bvh / counters */
void oracle_set_threads(int n);
int  oracle_threads(void);

/* pathtrace(pbo=NULL, 0, iter): one iteration accumulated into the running sum. */
void oracle_trace(Oracle* o, int iter);
void oracle_clear(Oracle* o);
void oracle_image(Oracle* o, float* rgb);                    /* W*H*3 running sum */
void oracle_display(Oracle* o, int iter, uint8_t* rgba);     /* sendImageToPBO    */
void oracle_stats(Oracle* o, PtStats* out);

/* stage-level entry points */
void oracle_camera_rays(Oracle* o, int iter, float* origins, float* directions);
void oracle_intersect_rays(Oracle* o, const float* origins, const float* directions, int n, PtHit* out);
/* brute force: every triangle of every mesh tested linearly (meshIntersectionTest) */
void oracle_intersect_rays_linear(Oracle* o, const float* origins, const float* directions, int n, PtHit* out);

/* known-answer helpers */
void oracle_rng_u01(int iter, int index, int depth, int n, float* out);
void oracle_hemisphere(Oracle* o, int iter, int index, int depth, int skip, const float normal[3], float out[3]);
float oracle_math_eval(int mode, int fn, float a, float b);
void oracle_math_eval_array(int mode, int fn, const float* a, const float* b, int n, float* out);

/* the oracle's own BVH build (BVH::build restated); returns node/leaf/depth counts */
int oracle_bvh_build(const PtTriangle* tris, int n, int maxDepth,
                     PtBvhNode** nodes, int* nodeCount, int32_t** indices, int* indexCount,
                     int* leafCount, int* maxDepthOut);
void oracle_free(void* p);
int oracle_mesh_bvh(Oracle* o, int mesh, const PtBvhNode** nodes, int* nodeCount,
                    const int32_t** indices, int* indexCount);

#ifdef __cplusplus
}
#endif
#endif
