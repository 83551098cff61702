/*
 * pt_scene_dev.h — device-resident scene and path-state layouts (HBM) and the
 * closest-hit / shading device functions shared by the kernels.
 *
 * Layout decisions (all arrays live in HBM for the lifetime of the context):
 *   path state   SoA indexed by PIXEL, three 16-byte records per path
 *                  rayO = (origin.xyz, prevIsSpecular as int bits)
 *                  rayD = (direction.xyz, unused)
 *                  col  = (throughput.rgb, remainingBounces as int bits)
 *                and two 16-byte hit records
 *                  hitA = (t, normal.xyz)     hitB = (materialId bits, u, v, geom bits)
 *                (between the closest-hit traversal and the primitive stage a mesh hit is the lean record
 *                  hitA = (t, bary u, bary v, leaf-order triangle bits), hitTri = geom)
 *                The wavefront queue holds 4-byte pixel indices only; sorting by
 *                material and compaction permute that queue, never the 80 bytes of
 *                state.  A queue slot index plays the role of the array index the
 *                reference seeds its RNG with (src/pathtrace.cu:2483-2484).
 *   BVH nodes    4-wide, 128 bytes = one cache line = eight float4, child boxes SoA:
 *                  minx[4] miny[4] minz[4] maxx[4] maxy[4] maxz[4] ref[4] (pad)
 *                  ref >= 0: child node;  ref < 0: leaf, ~ref = first triangle << 5 | count-1;
 *                  ref == kEmptyRef: unused slot.  The node set is the host builder's binary
 *                  tree with inner levels collapsed; every leaf keeps the host tree's box.
 *   triangles    stored in LEAF ORDER so a leaf's triangles are contiguous:
 *                  hot  48 bytes: (v0.xyz, e1.x) (e1.yz, e2.xy) (e2.z, orig index, visit rank, -)
 *                  cold 64 bytes: n0 n1 n2 uv0 uv1 uv2 materialId (fetched once per hit)
 */
#pragma once
#include "../../include/mi_pathtrace.h"
#include "pt_device.h"

namespace ptd {

constexpr int kBlock = 256;          /* threads per workgroup of every 1-D kernel (4 waves) */
#ifndef PT_LDS_STACK
#define PT_LDS_STACK 12
#endif
#ifndef PT_LDS_TOP
#define PT_LDS_TOP 0
#endif
constexpr int kLdsStack = PT_LDS_STACK; /* traversal stack entries (ref + entry distance) per ray held in LDS */
/* The first kTopNodes 4-wide nodes of the node array — the top levels of the first BVH mesh, which the host numbers
 * breadth first — can be kept in LDS by every traversal workgroup (PT_LDS_TOP = number of nodes, 0: none): every ray
 * starts there, and a step on an LDS-resident node costs the vector L1 no tag look-ups. */
constexpr int kTopNodes = PT_LDS_TOP;
constexpr int kStackDepth = 2 * kLdsStack; /* LDS dwords per lane                                    */
constexpr int kOverflowStack = 64;   /* further entries per ray in private memory, touched only past kLdsStack */
constexpr int kEmptyRef = 0x7fffffff;
constexpr int kLeafMaxTris = 32;     /* triangles addressable by one leaf reference            */
constexpr int kCtrStride = 32;        /* dwords between work counters: one 128-byte line each, so that atomics on
                                         different counters do not queue up on one line */
constexpr int kCtrRows = 512;         /* rows (iterations of a launch group) the counter arrays are laid out for */
constexpr int kFetchSegments = 8;     /* fetch counters per work queue, see k_trace */
/* Fetch counters of the traversal kernel: [bounce][segment][row], a 128-byte line each.  The eight counters of one
 * queue sit on lines of their own, kSegPitch dwords apart: atomics on one line are served one after the other,
 * whatever the address inside it, and every wave waits for its fetch.  A launch of its own over two million rays
 * makes 66 000 fetches; with all eight counters on one line and 4 096 waves contending they came back at about 60 M
 * per second, and that, not the traversal, set the launch's duration (1.1 ms; 0.78 ms with a line per counter). */
constexpr int kSegPitch = kCtrRows * kCtrStride;
constexpr int kFetchBouncePitch = kFetchSegments * kSegPitch;
constexpr int kMaxBuckets = 256;     /* material ids + the miss bucket handled by the sort   */

struct DGeom {
    int type, materialid, nodeStart, triStart;
    int triCount, hasBvh, meshIndex, hostNodeStart;     /* hostNodeStart / hostIndexStart: this mesh's part of SceneView::hostNodes / hostIndices */
    float T[16], I[16], IT[16];
    float translation[3]; int hostIndexStart;
    float scale[3], pad2;
    /* BVH meshes: world-space box around the transformed root box, padded by a percent of its size.
     * A ray that misses it by the plain slab test cannot pass any leaf box of the mesh. */
    float cullLo[3], pad3;
    float cullHi[3], pad4;
};
static_assert(sizeof(DGeom) == 288, "DGeom layout");

struct DTexture { const uint8_t* data; int width, height, components, tilesPerRow; };   /* data: 4-byte texels in 8 x 4 tiles, see sampleTexture */
/* base-colour and metallic-roughness texels of one material side by side (x: base colour, y: metallic-roughness, 4 bytes each) in
 * tiles of 4 x 4 texels = one 128-byte line, for materials whose two textures have one size: both bilinear footprints are
 * then the same four 8-byte loads; data null: the material's textures are sampled one by one */
struct DTexPair { const uint2* data; int width, height, tilesPerRow, pad; };
struct DLight { int geomIdx; float area; float pdf; };

struct SceneView {
    const DGeom* geoms;
    const PtMaterial* materials;
    const float4* nodes;
    const float4* triHot;
    const float4* triCold;
    const float4* triBox;       /* 2 per leaf-order triangle: min, max of its leaf in the host BVH; a hit counts only if the ray passes that box */
    const int* leafOfOrig;
    const PtBvhNode* hostNodes; /* the host BVHs as built (40-byte binary nodes, src/bvh.h:55-65) and their triangle index arrays: walked only by */
    const int* hostIndices;     /* travReference, for the rare ray whose result depends on the reference's own visiting order */
    const DTexture* textures;
    const DTexPair* texPairs;   /* one per material, or null */
    const DLight* lights;
    const float4* envData;      /* one 16-byte texel per pixel of the map (rgb, unused) */
    const float* envMarginal;
    const float* envConditional;
    const uint16_t* envGuide;   /* guide tables of the CDF searches (lowerBoundGuided): kCdfGuide entries for the marginal CDF, then as many per row of the conditional one; null: none */
    int numGeoms, numMaterials, numTextures, numLights;
    int envWidth, envHeight, envEnabled, useBvh;
    float envTotalPower;
    int numBuckets;                 /* numMaterials + 1 */
    int numBvhMeshes;               /* mesh geoms traversed through their BVH (0 when BVH use is off) */
    int topCount;                   /* leading nodes of the node array that traversal workgroups keep in LDS (<= kTopNodes) */
    /* Primary rays share their origin, throughput (1,1,1) and bounce budget: when cameraOrigin is set the camera
     * kernel does not store them and the stages of bounce 0 take them from here (a third of the bytes a primary
     * path moves).  Off for the stage hooks, whose rays come from the caller. */
    int cameraOrigin;
    /* A miss is fully described by its distance (-1): with leanHits set the traversal writes only the first 16 bytes
     * of a miss record and the later stages do not read the rest (set while tracing; the stage hooks export whole
     * records and leave it off). */
    int leanHits;
    /* With at most 32 geoms the traversal hands the geom of a mesh hit to the primitive stage in the five bits above the
     * 26-bit leaf-order triangle index of the hit record instead of a 4-byte array entry of its own (a scattered
     * write and a gathered read per mesh hit). */
    int packGeom;
    int traceDepth;
    float camPos[3];
};

/* Four of the 16-byte records of a path — ray origin, ray direction, colour, first half of the hit record — are
 * addressed through this accessor: record i of a field lives at p[i * kRecStride].  Stride 1 is one array per field;
 * stride 4 with the base pointers one record apart puts the four into one aligned 64-byte record, so a stage that
 * reaches a path through a queue (traversal, primitives of later bounces, shading) touches one record per path instead
 * of one line per field.  Bounce 0 walks the image in pixel order and needs only the direction: the camera kernel
 * writes that into an array of its own (PathState::rayD0) which the stages of bounce 0 read, and they exchange the
 * first half of the hit record through another (hitA0).  The second half of the hit record (material, uv, geom: hits
 * only) stays an array of its own. */
#ifndef PT_REC_STRIDE
#define PT_REC_STRIDE 4
#endif
constexpr int kRecStride = PT_REC_STRIDE;
struct PathField {
    float4* p;
    PT_DEV float4& operator[](size_t i) const { return p[i * kRecStride]; }
    __host__ PT_DEV PathField& operator+=(size_t records) { p += records * kRecStride; return *this; }
};

/* origin record (xyz, previous-bounce-was-specular flag) and colour record (throughput, remaining bounces) of a path */
PT_DEV float4 loadRayO(const SceneView& S, const PathField& rayO, int pix, int bounce)
{
    if (bounce == 0 && S.cameraOrigin) return make_float4(S.camPos[0], S.camPos[1], S.camPos[2], __int_as_float(0));
    return rayO[pix];
}
PT_DEV float4 loadCol(const SceneView& S, const PathField& col, int pix, int bounce)
{
    if (bounce == 0 && S.cameraOrigin) return make_float4(1.0f, 1.0f, 1.0f, __int_as_float(S.traceDepth));
    return col[pix];
}

/* Several iterations ("a batch") run through the same launches, one grid row (blockIdx.y) each.
 * Every iteration owns a full copy of the path state and of the queues, so slot numbering, sort
 * and compaction stay per iteration; at(k) selects iteration k's copy. */
struct PathState {
    PathField rayO;
    PathField rayD;
    PathField col;
    PathField hitA;
    float4* hitB;
    float4* rayD0;                  /* directions of the primary rays, one array in pixel order (the rayD array itself at stride 1) */
    float4* hitA0;                  /* first half of the hit records of bounce 0, likewise */
    float* image;                   /* W*H*3: this iteration's contribution inside the kernels */
    int* hitTri;                    /* mesh-local triangle of the hit (parity hook) */
    /* shadow rays whose mesh part is still to be traced, in arrival order, three records each: (origin, max distance)
       (direction, light geom or -1 for the environment) (path colour if the ray turns out unoccluded, pixel).  The
       shade stage appends them — consecutive lanes write consecutive records — and the shadow traversal reads them
       by queue position, so neither side gathers or scatters by pixel. */
    float4* shRec;
    int* traceQueue;                /* pixels whose ray may hit a BVH mesh, in arrival order (filled by the stage that
                                       creates the ray: camera, shade) */
    int* refList;                   /* rays whose mesh result depends on the reference's own leaf order and is settled by k_refwalk:
                                       pixels of closest-hit rays in [0, numPixels), shadow-queue positions in [numPixels, 2 numPixels) */
    size_t numPixels;
    __host__ PT_DEV PathState at(int k) const
    {
        PathState r = *this;
        const size_t off = (size_t)k * numPixels;
        r.rayO += off; r.rayD += off; r.col += off; r.hitA += off; r.hitB += off; r.rayD0 += off; r.hitA0 += off; r.image += 3 * off;
        r.hitTri += off; r.shRec += 3 * off; r.traceQueue += off; r.refList += 2 * off;
        return r;
    }
};

/* direction record (xyz, "may hit a BVH mesh" flag) of a path entering `bounce` */
PT_DEV float4 loadRayD(const SceneView& S, const PathState& st, int pix, int bounce)
{
    if (bounce == 0 && S.cameraOrigin) return st.rayD0[pix];
    return st.rayD[pix];
}

/* first half of the hit record (distance; barycentrics + triangle after the traversal, normal after the primitive stage) */
PT_DEV float4& hitARef(const SceneView& S, const PathState& st, int pix, int bounce)
{
    if (bounce == 0 && S.cameraOrigin) return st.hitA0[pix];
    return st.hitA[pix];
}

struct Queues {
    int* q0;            /* compacted queue (input of a bounce)                          */
    int* q1;            /* material-sorted queue (input of shade)                        */
    int* keys;          /* bucket per slot of q0                                         */
    int* blockHist;     /* [numBuckets][numBlocks] histogram, scanned in place           */
    int* blockAlive;    /* [numBlocks] survivors per workgroup, scanned in place         */
    uint32_t* live;     /* [PT_MAX_DEPTH + 2] live paths entering bounce b               */
    int numBlocks;      /* ceil(N / kBlock)                                              */
    int numPixels;
    int numBuckets;
    __host__ PT_DEV Queues at(int k) const
    {
        Queues r = *this;
        const size_t off = (size_t)k * numPixels;
        r.q0 += off; r.q1 += off; r.keys += off;
        r.blockHist += (size_t)k * numBuckets * numBlocks;
        r.blockAlive += (size_t)k * numBlocks;
        r.live += (size_t)k * (PT_MAX_DEPTH + 2);
        return r;
    }
};

struct Batch { int count; int iterBase; int iterStride; };

struct DevCounters {
    unsigned long long pathBounces;
    unsigned long long trav[2][4];      /* [0] intersect stage, [1] shade stage: nodes, triangles, mesh hits, child boxes */
    unsigned long long workItems[2];    /* rays handed to the traversal kernel: closest-hit, shadow */
    unsigned long long refWalks;        /* mesh walks repeated in the reference's own visiting order (travReference) */
    unsigned maxNodes[2], maxTris[2];   /* per-ray maxima of the counters (tail diagnosis) */
    unsigned maxStack[2];               /* deepest traversal stack any ray reached, per stage (entries; beyond kLdsStack they live in private memory) */
    unsigned maxWaveIters[2];           /* most loop iterations any wave of a traversal launch made (the launch's critical path) */
    unsigned long long launchT0;        /* diagnostics: wall clock of the first wave of the traversal launch being looked at */
    unsigned finishHist[64], dryHist[64];   /* waves by the 25 us slot they ended in / first saw the queue dry in */
    unsigned long long waveDiag[2][6];  /* per traversal mode: wave iterations, node-step lanes, leaf-step lanes, held-leaf lanes, refills, refilled lanes */
    uint32_t lastLive[PT_MAX_DEPTH + 2];
    uint32_t liveMax[PT_MAX_DEPTH + 2]; /* per bounce, the largest live count over the rows of the last launch group */
};

struct TravCounters { uint32_t nodes, tris, meshHits, boxes, maxSp; };

constexpr float kPI = 3.1415926535897932384626422832795028841971f;
constexpr float kTWO_PI = 6.2831853071795864769252867665590057683943f;
constexpr float kSQRT_OF_ONE_THIRD = 0.5773502691896257645091487805019574556476f;

struct Ray { F3 o, d; };
struct Hit { float t; F3 n; int mat; F2 uv; int geom; int tri; };

/* ------------------------------------------------------------- primitives */

PT_DEV F3 pointOnRay(F3 o, F3 d, float t) { return o + (t - .0001f) * normalize(d); }

/* unit cube, slab test in object space; semantics of boxIntersectionTest
 * (src/intersections.cu:4-58): no parallel-axis guard, entry or exit face,
 * hit point pulled back 1e-4, world distance returned */
PT_DEV float hitBox(const DGeom& g, const Ray& r, F3& N)
{
    F3 qo = mulMV(g.I, r.o, 1.0f);
    F3 qd = normalize(mulMV(g.I, r.d, 0.0f));
    float tmin = -1e38f, tmax = 1e38f;
    int nminAxis = -1, nmaxAxis = -1;
    float nminSign = 0.0f, nmaxSign = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float d = comp(qd, a), o = comp(qo, a);
        float t1 = (-0.5f - o) / d;
        float t2 = (+0.5f - o) / d;
        float ta = selMin(t1, t2);
        float tb = selMax(t1, t2);
        float s = t2 < t1 ? +1.0f : -1.0f;
        if (ta > 0 && ta > tmin) { tmin = ta; nminAxis = a; nminSign = s; }
        if (tb < tmax) { tmax = tb; nmaxAxis = a; nmaxSign = s; }
    }
    if (tmax >= tmin && tmax > 0) {
        if (tmin <= 0) { tmin = tmax; nminAxis = nmaxAxis; nminSign = nmaxSign; }
        F3 n = f3(nminAxis == 0 ? nminSign : 0.0f, nminAxis == 1 ? nminSign : 0.0f, nminAxis == 2 ? nminSign : 0.0f);
        F3 P = mulMV(g.T, pointOnRay(qo, qd, tmin), 1.0f);
        N = normalize(mulMV(g.IT, n, 0.0f));
        return length(r.o - P);
    }
    return -1.0f;
}

/* radius-0.5 sphere in object space (src/intersections.cu:60-114) */
PT_DEV float hitSphere(const DGeom& g, const Ray& r, F3& N)
{
    F3 ro = mulMV(g.I, r.o, 1.0f);
    F3 rd = normalize(mulMV(g.I, r.d, 0.0f));
    float vd = dot(ro, rd);
    float radicand = vd * vd - (dot(ro, ro) - 0.25f);
    if (radicand < 0) return -1.0f;
    float sq = __builtin_sqrtf(radicand);
    float first = -vd;
    float t1 = first + sq;
    float t2 = first - sq;
    float t;
    bool outside;
    if (t1 < 0 && t2 < 0) return -1.0f;
    else if (t1 > 0 && t2 > 0) { t = nanMin(t1, t2); outside = true; }
    else { t = nanMax(t1, t2); outside = false; }
    F3 objP = pointOnRay(ro, rd, t);
    F3 P = mulMV(g.T, objP, 1.0f);
    N = normalize(mulMV(g.IT, objP, 0.0f));
    if (!outside) N = -N;
    return length(r.o - P);
}

/* Moller-Trumbore on a hot triangle record (src/intersections.cu:268-308) */
PT_DEV float hitTriangleRec(float4 a, float4 b, float4 c, F3 lo, F3 ld, F3& bary);
PT_DEV float hitTriangle(const float4* hot, F3 lo, F3 ld, F3& bary)
{
    return hitTriangleRec(hot[0], hot[1], hot[2], lo, ld, bary);
}
PT_DEV float hitTriangleRec(float4 a, float4 b, float4 c, F3 lo, F3 ld, F3& bary)
{
    const float EPS = 0.0000001f;
    F3 v0 = f3(a.x, a.y, a.z), e1 = f3(a.w, b.x, b.y), e2 = f3(b.z, b.w, c.x);
    F3 h = cross(ld, e2);
    float det = dot(e1, h);
    if (det > -EPS && det < EPS) return -1.0f;
    float f = 1.0f / det;
    F3 s = lo - v0;
    float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return -1.0f;
    F3 q = cross(s, e1);
    float v = f * dot(ld, q);
    if (v < 0.0f || u + v > 1.0f) return -1.0f;
    float t = f * dot(e2, q);
    if (t > EPS) {
        float w = 1.0f - u - v;
        bary = f3(w, u, v);
        return t;
    }
    return -1.0f;
}

/* Closest hit of one mesh.
 *
 * The reference walks a binary tree with an explicit stack, right child first, no near/far
 * order, one 40-byte node per dependent fetch (src/intersections.cu:344-386).  What that loop
 * computes is order-independent except for exact-distance ties: the closest of all triangles
 * whose LEAF box passes the slab test, the first-visited one among equals (strict '<').
 * Ancestor boxes only accelerate: slab intervals of nested boxes nest in IEEE arithmetic
 * (subtraction and multiplication by the same 1/d are monotone), so a ray that passes a leaf
 * box passes every ancestor.  The leaf boxes therefore are a FILTER on triangle hits, and the
 * hierarchy that is walked to find those hits is free.  This traversal walks its own: 4-wide
 * nodes down to boxes around one or two triangles, nearest child first, entry distance kept on
 * the stack for pruning at pop time, boxes that only have to be conservative (kBoxSlack), and the
 * reference's choice among equals reproduced with a per-triangle visit rank computed on the
 * host from the reference's order.  The reference also skips a node whose entry distance
 * exceeds the closest hit so far.  For boxes that contain their triangles that can only skip
 * triangles that would lose anyway — up to rounding at a box face, where its result depends on
 * its visiting order.  The walk therefore also keeps the runner-up distance, the finished walk's
 * winner is checked against the box of its leaf in the reference's BVH with the reference's
 * arithmetic (travCertain), and the rare ray for which that check does not settle the matter is
 * walked again in the reference's own order over the reference's own nodes (travReference).
 *
 * stack: this lane's column of a [kStackDepth][kBlock] LDS array (refs in rows 0..kLdsStack-1,
 * entry distances in the rows above); deeper entries spill to private memory.
 * Returns world distance or -1; attributes are fetched only when wantAttr. */
struct MeshTrav {
    F3 lo, ld, invDir;          /* ray in the mesh's object space */
    float closestT;
    int best, bestRank;
    F3 bary;                    /* .y, .z: barycentrics of the closest hit; .x: smallest distance of any OTHER triangle hit the walk has seen (see travCertain) */
    int sp, ref;
    int nodeStart, triStart;    /* of the mesh being walked, kept in registers: a per-step reload from the geom
                                   record would put one more dependent memory access in front of every node fetch */
    uint32_t selX, selY, selZ;  /* byte offset (0 or 48) of the plane array a slab is entered through, per axis */
};

PT_DEV void travInit(const DGeom& g, const Ray& r, MeshTrav& tv)
{
    tv.lo = mulMV(g.I, r.o, 1.0f);
    tv.ld = normalize(mulMV(g.I, r.d, 0.0f));
    /* reciprocal for the walked boxes only, which carry a slack of 2^-15: the hardware's one-ulp approximation (one
     * instruction per component instead of an IEEE division) is far inside it; the check of a hit against its host
     * leaf box takes the true quotient (refSlab) */
    tv.invDir = f3(__builtin_amdgcn_rcpf(tv.ld.x), __builtin_amdgcn_rcpf(tv.ld.y), __builtin_amdgcn_rcpf(tv.ld.z));
    tv.closestT = 3.402823466e+38f;
    tv.best = -1;
    tv.bestRank = 0x7fffffff;
    tv.bary = f3(3.402823466e+38f, 0.0f, 0.0f);
    tv.sp = 0;
    tv.ref = 0;
    tv.nodeStart = g.nodeStart;
    tv.triStart = g.triStart;
    const float INF = __builtin_inff();
    /* A ray with a NaN or an infinity in its origin or direction hits no triangle in the reference either (every
     * barycentric test or the distance test fails, or the distance comes out infinite), whatever its box tests say.
     * Its origin is moved to infinity: every walked box then fails (slab distances of -inf or +inf on all planes),
     * and a triangle test against it fails as well. */
    const bool valid = __builtin_fabsf(tv.lo.x) < INF && __builtin_fabsf(tv.lo.y) < INF && __builtin_fabsf(tv.lo.z) < INF &&
                       __builtin_fabsf(tv.ld.x) < INF && __builtin_fabsf(tv.ld.y) < INF && __builtin_fabsf(tv.ld.z) < INF;
    tv.lo.x = valid ? tv.lo.x : INF;
    tv.lo.y = valid ? tv.lo.y : INF;
    tv.lo.z = valid ? tv.lo.z : INF;
    /* The walked boxes are tested with a finite reciprocal: a zero direction component gets 1e25 instead of infinity.
     * With the origin between the two planes of that axis the slab interval then covers every distance that matters
     * (or ends at 0 when the origin lies in a plane, which only the ray's own starting point needs); outside them it
     * lies beyond any hit or behind the origin, as it should — and 0 x infinity, a NaN that the plain min / max
     * would take for a miss, cannot arise.  The host leaf box test of a hit uses the true reciprocal. */
    if (!(__builtin_fabsf(tv.invDir.x) < INF)) tv.invDir.x = 1e25f;
    if (!(__builtin_fabsf(tv.invDir.y) < INF)) tv.invDir.y = 1e25f;
    if (!(__builtin_fabsf(tv.invDir.z) < INF)) tv.invDir.z = 1e25f;
    /* a node stores its four child boxes as six plane arrays: min x, y, z at bytes 0, 16, 32 and max x, y, z at 48, 64, 80 */
    tv.selX = tv.invDir.x < 0.0f ? 48u : 0u;
    tv.selY = tv.invDir.y < 0.0f ? 48u : 0u;
    tv.selZ = tv.invDir.z < 0.0f ? 48u : 0u;
}

/* The boxes of the hierarchy that is walked only have to be conservative: a triangle the reference would hit must
 * never be cut off by them.  Their slab distances are computed with rounding (each product is within two units in
 * the last place of its exact value), and the triangle test has an error of its own of a few parts in ten million
 * of the distance; both scale with the distances themselves.  The interval of a walked box is therefore widened by
 * 2^-15 of its far end before it is compared with anything (and the closest distance likewise before an entry distance
 * is compared with it) — about a hundred times those errors, at any scale. */
constexpr float kBoxSlack = 3.0517578125e-05f;
PT_DEV float slackFar(float t) { return __builtin_fmaf(__builtin_fabsf(t), kBoxSlack, t); }

/* intersectAABB (src/intersections.cu:248-266) in the reference's arithmetic: component-wise min / max as compare-and-select
 * (NaN behaviour included), the true reciprocal of the direction, infinities included. */
PT_DEV bool refSlab(F3 bmin, F3 bmax, F3 lo, F3 invDir, float& tMin, float& tMax)
{
    float ax0 = (bmin.x - lo.x) * invDir.x, ax1 = (bmax.x - lo.x) * invDir.x;
    float ay0 = (bmin.y - lo.y) * invDir.y, ay1 = (bmax.y - lo.y) * invDir.y;
    float az0 = (bmin.z - lo.z) * invDir.z, az1 = (bmax.z - lo.z) * invDir.z;
    tMin = selMax(selMax(selMin(ax0, ax1), selMin(ay0, ay1)), selMin(az0, az1));
    tMax = selMin(selMin(selMax(ax0, ax1), selMax(ay0, ay1)), selMax(az0, az1));
    return tMax >= tMin && tMax >= 0.0f;
}

/* Is the closest hit G of a finished walk certainly the reference's answer?
 *
 * The walk has tested every triangle whose (padded) box the ray passes up to slackFar(distance of G), so G is the smallest
 * distance over ALL triangles, the first-visited among equals.  The reference scans its leaves in its own order, tests a
 * leaf's triangles only if the ray passes the leaf's box, and skips a node whose box is entered beyond the closest hit
 * so far (src/intersections.cu:351-358).  G is the reference's answer whenever the reference cannot have skipped it:
 *   - the ray passes the box of G's leaf in the reference's BVH (its ancestors' boxes then too: nested slab intervals
 *     nest in IEEE arithmetic), and
 *   - that box is entered no later than G's own distance — every "closest so far" the reference ever holds is the
 *     distance of some triangle hit, hence >= G's, hence >= the entry distance of every box around G — or, when rounding
 *     puts the entry a little beyond G's distance (a triangle that lies in a face of its leaf's box; a hit a hair outside
 *     the triangle's rim), no OTHER triangle was hit before that entry distance: then "closest so far" was never
 *     below it while G's leaf waited.  The walk knows the smallest other hit distance (bary.x) for everything it
 *     tested, which covers all distances up to slackFar(G's).
 * Otherwise the outcome depends on the order in which the reference visits its leaves, and travReference repeats
 * that order. */
PT_DEV bool travCertain(const SceneView& S, const MeshTrav& tv)
{
    if (tv.best < 0) return true;
    const float4* b = S.triBox + 2 * ((size_t)tv.triStart + (size_t)tv.best);
    const float4 bmin = b[0], bmax = b[1];
    float tMin, tMax;
    if (!refSlab(f3(bmin.x, bmin.y, bmin.z), f3(bmax.x, bmax.y, bmax.z), tv.lo, 1.0f / tv.ld, tMin, tMax)) return false;
    if (!(tMin > tv.closestT)) return true;
    return !(tMin > slackFar(tv.closestT)) && !(tv.bary.x < tMin);
}

/* meshIntersectionTestBVH's own loop (src/intersections.cu:331-386) over the host BVH as built: explicit stack of 64
 * node indices (a push beyond it is dropped, as there), left child pushed first and so the right one visited first, a
 * node skipped when the ray misses its box or enters it beyond the closest hit so far, strict '<' between triangles.
 * Same slab and triangle arithmetic as everywhere else; the triangle records are reached through the leaf-order map.
 * stack: 64 ints of private memory. */
PT_DEV void travReference(const SceneView& S, const DGeom& g, MeshTrav& tv, int* stack)
{
    const PtBvhNode* nodes = S.hostNodes + g.hostNodeStart;
    const int* indices = S.hostIndices + g.hostIndexStart;
    const float4* hot = S.triHot + 3 * (size_t)g.triStart;
    const F3 lo = tv.lo, ld = tv.ld, invDir = 1.0f / tv.ld;
    float closestT = 3.402823466e+38f;
    int best = -1;
    float bu = 0.0f, bv = 0.0f;
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const PtBvhNode nd = nodes[stack[--sp]];
        float tMin, tMax;
        if (!refSlab(f3(nd.bmin[0], nd.bmin[1], nd.bmin[2]), f3(nd.bmax[0], nd.bmax[1], nd.bmax[2]), lo, invDir, tMin, tMax)) continue;
        if (tMin > closestT) continue;
        if (nd.leftChild == -1) {
            for (int i = 0; i < nd.triangleCount; ++i) {
                const int slot = S.leafOfOrig[g.triStart + indices[nd.triangleStart + i]];
                F3 bary;
                const float tt = hitTriangle(hot + 3 * (size_t)slot, lo, ld, bary);
                if (tt > 0.0f && tt < closestT) { closestT = tt; best = slot; bu = bary.y; bv = bary.z; }
            }
        } else {
            if (nd.leftChild >= 0 && sp < 64) stack[sp++] = nd.leftChild;
            if (nd.rightChild >= 0 && sp < 64) stack[sp++] = nd.rightChild;
        }
    }
    tv.closestT = closestT;
    tv.best = best;
    tv.bary.y = bu;
    tv.bary.z = bv;
}

/* One step: visit tv.ref (a 4-wide node or a leaf), then pop the next entry that can still hold a
 * closer-or-equal hit.  Returns false when the stack has run dry. */
template <bool COUNT>
PT_DEV bool travStep(const SceneView& S, MeshTrav& tv, int* stack, int* ovRef, float* ovT, TravCounters& cnt, bool leafTurn = true, const float4* top = nullptr)
{
    const float INF = __builtin_inff();
    const int ref = tv.ref;
    if (ref >= 0) {
        float tk[4];
        int rk[4];
        const F3 lo = tv.lo, invDir = tv.invDir;
        /* the slack of the box tests (kBoxSlack) sits on two of the three comparisons' far sides: the exit distance and
         * the closest hit so far; entry distances go to the stack as computed and meet the same widened bound at pop time */
        const float closestSlack = __builtin_fminf(slackFar(tv.closestT), 3.402823466e+38f);       /* finite: an empty slot's +inf must still fail */
        {
            /* With a finite origin and a finite reciprocal direction (travInit) no NaN can arise, and for a box with min <= max the
             * smaller of (min - o) * inv and (max - o) * inv is the one whose plane the ray's direction sign names
             * (subtraction and the multiplication by inv are monotone).  The entry planes are therefore picked by
             * address — three offsets per ray — and the interval is max3 of the entry distances, min3 of the exit
             * distances: the values the compare-and-select forms of the reference's vector library give (up to the
             * sign of a zero, which no comparison below sees), at two thirds of the instructions.
             * Empty slots fail: their boxes sit at +inf. */
            const uint32_t nodeIdx = (uint32_t)tv.nodeStart + (uint32_t)ref;
            const uint32_t off = nodeIdx * 128u;
            const uint32_t exitOff = off + 48u;
            float4 nX, nY, nZ, fX, fY, fZ, rf;
            if (kTopNodes > 0 && nodeIdx < (uint32_t)S.topCount) {
                const char* lb = (const char*)top;
                nX = *(const float4*)(lb + (off + tv.selX));
                nY = *(const float4*)(lb + (off + tv.selY) + 16);
                nZ = *(const float4*)(lb + (off + tv.selZ) + 32);
                fX = *(const float4*)(lb + (exitOff - tv.selX));
                fY = *(const float4*)(lb + (exitOff - tv.selY) + 16);
                fZ = *(const float4*)(lb + (exitOff - tv.selZ) + 32);
                rf = *(const float4*)(lb + off + 96);
            } else {
                const char* base = (const char*)S.nodes;
                nX = *(const float4*)(base + (off + tv.selX));
                nY = *(const float4*)(base + (off + tv.selY) + 16);
                nZ = *(const float4*)(base + (off + tv.selZ) + 32);
                fX = *(const float4*)(base + (exitOff - tv.selX));
                fY = *(const float4*)(base + (exitOff - tv.selY) + 16);
                fZ = *(const float4*)(base + (exitOff - tv.selZ) + 32);
                rf = *(const float4*)(base + off + 96);
            }
            if (COUNT) {
                cnt.nodes++;
                cnt.boxes += (__float_as_int(rf.x) != kEmptyRef) + (__float_as_int(rf.y) != kEmptyRef) +
                             (__float_as_int(rf.z) != kEmptyRef) + (__float_as_int(rf.w) != kEmptyRef);
            }
            /* plane distances two children at a time: gfx950 has packed single-precision add and multiply (two IEEE
             * operations per instruction, same results as the scalar forms), which halves the 48 instructions of the
             * 24 slab distances */
            typedef float pk2 __attribute__((ext_vector_type(2)));
            const pk2 ox = { lo.x, lo.x }, oy = { lo.y, lo.y }, oz = { lo.z, lo.z };
            const pk2 ix = { invDir.x, invDir.x }, iy = { invDir.y, invDir.y }, iz = { invDir.z, invDir.z };
#define PT_PAIR(P, a, b) (pk2){ P.a, P.b }
            const pk2 nx0 = (PT_PAIR(nX, x, y) - ox) * ix, nx1 = (PT_PAIR(nX, z, w) - ox) * ix;
            const pk2 ny0 = (PT_PAIR(nY, x, y) - oy) * iy, ny1 = (PT_PAIR(nY, z, w) - oy) * iy;
            const pk2 nz0 = (PT_PAIR(nZ, x, y) - oz) * iz, nz1 = (PT_PAIR(nZ, z, w) - oz) * iz;
            const pk2 fx0 = (PT_PAIR(fX, x, y) - ox) * ix, fx1 = (PT_PAIR(fX, z, w) - ox) * ix;
            const pk2 fy0 = (PT_PAIR(fY, x, y) - oy) * iy, fy1 = (PT_PAIR(fY, z, w) - oy) * iy;
            const pk2 fz0 = (PT_PAIR(fZ, x, y) - oz) * iz, fz1 = (PT_PAIR(fZ, z, w) - oz) * iz;
#undef PT_PAIR
#define PT_CHILD_FAST(k, H, E, C)                                                                   \
            {                                                                                        \
                float tMin = __builtin_fmaxf(__builtin_fmaxf(nx##H.E, ny##H.E), nz##H.E);             \
                float tMax = slackFar(__builtin_fminf(__builtin_fminf(fx##H.E, fy##H.E), fz##H.E));   \
                bool in = (tMax >= tMin && tMax >= 0.0f) && !(tMin > closestSlack);                  \
                tk[k] = in ? tMin : INF;                                                             \
                rk[k] = __float_as_int(rf.C);                                                        \
            }
            PT_CHILD_FAST(0, 0, x, x) PT_CHILD_FAST(1, 0, y, y) PT_CHILD_FAST(2, 1, x, z) PT_CHILD_FAST(3, 1, y, w)
#undef PT_CHILD_FAST
        }
        /* nearest first: 5-exchange network on (distance, ref) */
#define PT_CSWAP(i, j) { bool sw = tk[j] < tk[i]; float tt_ = sw ? tk[j] : tk[i]; float tu_ = sw ? tk[i] : tk[j]; \
                         int ra_ = sw ? rk[j] : rk[i]; int rb_ = sw ? rk[i] : rk[j]; tk[i] = tt_; tk[j] = tu_; rk[i] = ra_; rk[j] = rb_; }
        PT_CSWAP(0, 1) PT_CSWAP(2, 3) PT_CSWAP(0, 2) PT_CSWAP(1, 3) PT_CSWAP(1, 2)
#undef PT_CSWAP
        int sp = tv.sp;
        if (sp + 3 <= kLdsStack) {
            /* the common case: all three pushes fit the LDS part of the stack, no bounds to look at */
#define PT_PUSH_LDS(k)                                                                               \
            if (tk[k] < INF) { stack[sp * kBlock] = rk[k]; stack[(kLdsStack + sp) * kBlock] = __float_as_int(tk[k]); sp++; }
            PT_PUSH_LDS(3) PT_PUSH_LDS(2) PT_PUSH_LDS(1)
#undef PT_PUSH_LDS
        } else {
#define PT_PUSH(k)                                                                                   \
            if (tk[k] < INF) {                                                                       \
                if (sp < kLdsStack) { stack[sp * kBlock] = rk[k]; stack[(kLdsStack + sp) * kBlock] = __float_as_int(tk[k]); sp++; } \
                else if (sp < kLdsStack + kOverflowStack) { ovRef[sp - kLdsStack] = rk[k]; ovT[sp - kLdsStack] = tk[k]; sp++; }     \
            }
            PT_PUSH(3) PT_PUSH(2) PT_PUSH(1)
#undef PT_PUSH
        }
        tv.sp = sp;
        if (COUNT && (uint32_t)sp > cnt.maxSp) cnt.maxSp = (uint32_t)sp;
        if (tk[0] < INF) { tv.ref = rk[0]; return true; }
    } else {
        if (!leafTurn) return true;     /* hold the leaf until enough lanes of the wave have one */
        const unsigned x = ~(unsigned)ref;
        const int first = (int)(x >> 5), count = (int)(x & 31u) + 1;
        const float4* tp = S.triHot + 3 * ((size_t)tv.triStart + (size_t)first);
        /* one triangle against the ray: a new closest hit, a tie that the reference would have met first, or a candidate
         * for the runner-up distance (see travCertain) */
#define PT_TEST_TRIANGLE(ra, rb, rc, index)                                                                     \
        {                                                                                                        \
            F3 bary;                                                                                             \
            const float tt = hitTriangleRec(ra, rb, rc, tv.lo, tv.ld, bary);                                     \
            if (COUNT) cnt.tris++;                                                                               \
            if (tt > 0.0f) {                                                                                     \
                if (!(tt > tv.closestT)) {                                                                       \
                    const int rank = __float_as_int(rc.z);                                                       \
                    const bool closer = tt < tv.closestT;                                                        \
                    /* the runner-up: the hit this one displaces, or this one when it only ties */              \
                    tv.bary.x = closer ? tv.closestT : tt;                                                       \
                    if (closer || rank < tv.bestRank) { tv.closestT = tt; tv.bary.y = bary.y; tv.bary.z = bary.z; tv.best = (index); tv.bestRank = rank; } \
                } else if (tt < tv.bary.x) tv.bary.x = tt;                                                       \
            }                                                                                                    \
        }
        /* the next triangle's record is requested before the current one is tested: one memory round trip
         * per leaf instead of one per triangle */
        float4 ra = tp[0], rb = tp[1], rc = tp[2];
        for (int i = 0; i < count; ++i) {
            float4 na = ra, nb = rb, nc = rc;
            if (i + 1 < count) { na = tp[3 * (i + 1)]; nb = tp[3 * (i + 1) + 1]; nc = tp[3 * (i + 1) + 2]; }
            PT_TEST_TRIANGLE(ra, rb, rc, first + i)
            ra = na; rb = nb; rc = nc;
        }
#undef PT_TEST_TRIANGLE
    }
    /* pop the next entry that can still hold a closer (or equal) hit */
    int sp = tv.sp;
    const float popBound = slackFar(tv.closestT);
    if (sp <= kLdsStack) {
        /* everything left is in LDS */
        while (sp > 0) {
            sp--;
            const int r2 = stack[sp * kBlock];
            const float t = __int_as_float(stack[(kLdsStack + sp) * kBlock]);
            if (!(t > popBound)) { tv.sp = sp; tv.ref = r2; return true; }
        }
    } else {
        while (sp > 0) {
            sp--;
            float t;
            int r2;
            /* the empty asm statements keep the two address spaces apart: merged into one generic-pointer
             * load, every pop would be a flat access that waits for all outstanding vector memory as well */
            if (sp < kLdsStack) {
                r2 = stack[sp * kBlock]; t = __int_as_float(stack[(kLdsStack + sp) * kBlock]);
                asm volatile("" : "+v"(r2), "+v"(t));
            } else {
                r2 = ovRef[sp - kLdsStack]; t = ovT[sp - kLdsStack];
                asm volatile("" : "+v"(r2), "+v"(t));
            }
            if (!(t > popBound)) { tv.sp = sp; tv.ref = r2; return true; }
        }
    }
    tv.sp = 0;
    return false;
}

/* Attributes of a mesh hit (second half of the hit epilogue, src/intersections.cu:388-419): interpolated and
 * transformed normal facing the ray, uv, material and original triangle index of leaf-order triangle `best` hit at
 * barycentrics (1 - u - v, u, v).  The closest-hit traversal only records (distance, u, v, best, geom); the
 * one-thread-per-slot stage that follows calls this, with every lane busy, for the hits that survive. */
PT_DEV void meshHitAttributes(const SceneView& S, const DGeom& g, F3 rayD, int best, float bu, float bv,
                              F3& N, F2& uv, int& mat, int& triOrig, bool wantOrig = true)
{
    const float bw = 1.0f - bu - bv;            /* as the triangle test forms it */
    const float4* cold = S.triCold + 4 * (size_t)(g.triStart + best);
    float4 c0 = cold[0], c1 = cold[1], c2 = cold[2], c3 = cold[3];
    F3 n0 = f3(c0.x, c0.y, c0.z), n1 = f3(c0.w, c1.x, c1.y), n2 = f3(c1.z, c1.w, c2.x);
    F3 ln = normalize(bw * n0 + bu * n1 + bv * n2);
    uv.x = (bw * c2.y + bu * c2.w) + bv * c3.y;
    uv.y = (bw * c2.z + bu * c3.x) + bv * c3.z;
    mat = __float_as_int(c3.w);
    /* the original index is wanted by the stage hooks only; inside the loop nobody reads it, and fetching it would
     * touch one more memory line per mesh hit */
    triOrig = wantOrig ? __float_as_int(S.triHot[3 * ((size_t)g.triStart + (size_t)best) + 2].y) : -1;
    N = normalize(mulMV(g.IT, ln, 0.0f));
    bool outside = dot(rayD, N) < 0.0f;
    if (!outside) N = -N;
}

/* hit epilogue (src/intersections.cu:388-419): world distance, and when wantAttr the interpolated
 * normal / uv / material of the best triangle */
PT_DEV float travFinish(const SceneView& S, const DGeom& g, const Ray& r, const MeshTrav& tv, bool wantAttr,
                        F3& N, F2& uv, int& mat, int& triOrig)
{
    if (tv.best < 0) return -1.0f;
    F3 localP = tv.lo + tv.ld * tv.closestT;
    F3 P = mulMV(g.T, localP, 1.0f);
    if (wantAttr) meshHitAttributes(S, g, r.d, tv.best, tv.bary.y, tv.bary.z, N, uv, mat, triOrig);
    return length(r.o - P);
}

/* linear scan in ORIGINAL triangle order (meshIntersectionTest, src/intersections.cu:168-245), for
 * meshes that are not traced through their BVH */
template <bool COUNT>
PT_DEV float hitMeshLinear(const SceneView& S, const DGeom& g, const Ray& r, bool wantAttr,
                           F3& N, F2& uv, int& mat, int& triOrig, TravCounters& cnt)
{
    MeshTrav tv;
    travInit(g, r, tv);
    const float4* hot = S.triHot + 3 * (size_t)g.triStart;
    for (int i = 0; i < g.triCount; ++i) {
        int leafPos = S.leafOfOrig[g.triStart + i];
        F3 bary;
        float tt = hitTriangle(hot + 3 * (size_t)leafPos, tv.lo, tv.ld, bary);
        if (COUNT) cnt.tris++;
        if (tt > 0.0f && tt < tv.closestT) { tv.closestT = tt; tv.bary = bary; tv.best = leafPos; }
    }
    if (COUNT && tv.best >= 0) cnt.meshHits++;
    return travFinish(S, g, r, tv, wantAttr, N, uv, mat, triOrig);
}

/* true when geom i is traced by the persistent BVH kernel rather than inline */
PT_DEV bool isBvhMesh(const SceneView& S, const DGeom& g) { return g.type == PT_GLTF_MESH && S.useBvh && g.hasBvh; }

/* The closest hit over all geoms is the reference's in-order loop with a strict '<'
 * (src/pathtrace.cu:868-919): the smallest distance wins and equal distances go to the lower geom
 * index.  Written as that rule, candidates can be folded in any order. */
PT_DEV bool closer(float t, int geom, const Hit& best)
{
    return t > 0.0f && (best.geom < 0 || t < best.t || (t == best.t && geom < best.geom));
}

/* fold every geom that is not a BVH mesh (boxes, spheres, meshes in linear-scan mode) into best */
template <bool COUNT>
PT_DEV void foldInlineGeoms(const SceneView& S, const Ray& r, Hit& best, TravCounters& cnt)
{
    for (int i = 0; i < S.numGeoms; ++i) {
        const DGeom& g = S.geoms[i];
        if (isBvhMesh(S, g)) continue;
        float t = -1.0f;
        F3 n = f3(0.0f);
        F2 uv; uv.x = 0.5f; uv.y = 0.5f;
        int mat = g.materialid, tri = -1;
        if (g.type == PT_CUBE) t = hitBox(g, r, n);
        else if (g.type == PT_SPHERE) t = hitSphere(g, r, n);
        else if (g.type == PT_GLTF_MESH) t = hitMeshLinear<COUNT>(S, g, r, true, n, uv, mat, tri, cnt);
        if (closer(t, i, best)) { best.t = t; best.n = n; best.mat = mat; best.uv = uv; best.geom = i; best.tri = tri; }
    }
}

/* distance of the closest hit on one inline geom, for shadow rays */
template <bool COUNT>
PT_DEV float inlineOccluderDistance(const SceneView& S, const DGeom& g, const Ray& r, TravCounters& cnt)
{
    F3 n; F2 uv; int m, tri;
    if (g.type == PT_SPHERE) return hitSphere(g, r, n);
    if (g.type == PT_CUBE) return hitBox(g, r, n);
    if (g.type == PT_GLTF_MESH) return hitMeshLinear<COUNT>(S, g, r, false, n, uv, m, tri, cnt);
    return -1.0f;
}

/* The traversal's first step, taken ahead of time: true when the ray fails all four child boxes of the mesh's
 * root node, with the arithmetic travStep uses for them (same loads, same operations, closest distance still at
 * its initial value).  Such a ray leaves the traversal empty-handed after that one step, so it need not be queued.
 * (A ray with a non-finite origin or direction fails them all, see travInit: it hits nothing.) */
PT_DEV bool failsRootNode(const SceneView& S, const DGeom& g, const Ray& r)
{
    MeshTrav tv;
    travInit(g, r, tv);
    const F3 lo = tv.lo, invDir = tv.invDir;
    const char* base = (const char*)S.nodes;
    const uint32_t off = (uint32_t)tv.nodeStart * 128u;
    const uint32_t exitOff = off + 48u;
    const float4 nX = *(const float4*)(base + (off + tv.selX));
    const float4 nY = *(const float4*)(base + (off + tv.selY) + 16);
    const float4 nZ = *(const float4*)(base + (off + tv.selZ) + 32);
    const float4 fX = *(const float4*)(base + (exitOff - tv.selX));
    const float4 fY = *(const float4*)(base + (exitOff - tv.selY) + 16);
    const float4 fZ = *(const float4*)(base + (exitOff - tv.selZ) + 32);
    bool any = false;
#define PT_CHILD_ROOT(C)                                                                            \
    {                                                                                                \
        float tMin = __builtin_fmaxf(__builtin_fmaxf((nX.C - lo.x) * invDir.x, (nY.C - lo.y) * invDir.y), (nZ.C - lo.z) * invDir.z); \
        float tMax = slackFar(__builtin_fminf(__builtin_fminf((fX.C - lo.x) * invDir.x, (fY.C - lo.y) * invDir.y), (fZ.C - lo.z) * invDir.z)); \
        any = any || ((tMax >= tMin && tMax >= 0.0f) && !(tMin > tv.closestT));                      \
    }
    PT_CHILD_ROOT(x) PT_CHILD_ROOT(y) PT_CHILD_ROOT(z) PT_CHILD_ROOT(w)
#undef PT_CHILD_ROOT
    return !any;
}

/* Pre-test of a ray against the BVH meshes: false only when the ray (up to distance maxDist along it, in units
 * of |d|) certainly misses all of them.  First the padded world box of a mesh — every comparison is false on NaN
 * and the min/max ignore a NaN operand, so odd rays are kept — then, for rays that pass it, the mesh's root node. */
PT_DEV bool mayHitBvhMeshes(const SceneView& S, const Ray& r, float maxDist, bool rootTest = false)
{
    /* the box is padded by a percent of its size: the hardware's one-ulp reciprocal is exact enough */
    const F3 inv = f3(__builtin_amdgcn_rcpf(r.d.x), __builtin_amdgcn_rcpf(r.d.y), __builtin_amdgcn_rcpf(r.d.z));
    for (int i = 0; i < S.numGeoms; ++i) {
        const DGeom& g = S.geoms[i];
        if (!isBvhMesh(S, g)) continue;
        float x0 = (g.cullLo[0] - r.o.x) * inv.x, x1 = (g.cullHi[0] - r.o.x) * inv.x;
        float y0 = (g.cullLo[1] - r.o.y) * inv.y, y1 = (g.cullHi[1] - r.o.y) * inv.y;
        float z0 = (g.cullLo[2] - r.o.z) * inv.z, z1 = (g.cullHi[2] - r.o.z) * inv.z;
        float tMin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x0, x1), __builtin_fminf(y0, y1)), __builtin_fminf(z0, z1));
        float tMax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x0, x1), __builtin_fmaxf(y0, y1)), __builtin_fmaxf(z0, z1));
        const bool miss = (tMax < tMin) || (tMax < 0.0f) || (tMin > maxDist);
        if (!miss && !(rootTest && failsRootNode(S, g, r))) return true;
    }
    return false;
}

/* occlusion rules of the two shadow tests, per geom (src/pathtrace.cu:2222 and :2293) */
PT_DEV bool blocksLight(float t, float dist) { return t > 0.001f && t < dist - 0.001f; }
PT_DEV bool blocksEnv(float t) { return !(t < 0.001f); }

/* ---------------------------------------------------------------- sampling */

/* cosine-weighted hemisphere (src/interactions.cu:7-45) */
PT_DEV F3 sampleHemisphere(F3 normal, Lcg& rng)
{
    float up = __builtin_sqrtf(rng.u01());
    float over = __builtin_sqrtf(1 - up * up);
    float around = rng.u01() * kTWO_PI;
    F3 other;
    if (__builtin_fabsf(normal.x) < kSQRT_OF_ONE_THIRD) other = f3(1, 0, 0);
    else if (__builtin_fabsf(normal.y) < kSQRT_OF_ONE_THIRD) other = f3(0, 1, 0);
    else other = f3(0, 0, 1);
    F3 p1 = normalize(cross(normal, other));
    F3 p2 = normalize(cross(normal, p1));
    return up * normal + dm::cos(around) * over * p1 + dm::sin(around) * over * p2;
}

/* hand-written bilinear fetch on 8-bit texels, repeat wrap, no sRGB decode (src/pathtrace.cu:937-1019).
 * The device copy of a texture holds four bytes per texel whatever the file had (grey replicated, alpha 255 where there
 * was none: 255 / 255.0f is the 1.0f the reference sets), in tiles of 8 x 4 texels = one 128-byte line each, so that a
 * 2 x 2 footprint is four 4-byte loads which mostly share a line instead of two lines of a row-major image.  The values
 * fetched, and all arithmetic on them, are the reference's. */
constexpr int kTexTileW = 8, kTexTileH = 4;
PT_DEV int texRowPart(const DTexture& tex, int y) { return (y >> 2) * tex.tilesPerRow * 32 + ((y & 3) << 3); }
PT_DEV int texColPart(int x) { return ((x >> 3) << 5) + (x & 7); }
PT_DEV F4 decodeTexel(uint32_t p)
{
    F4 r;
    r.x = (p & 255u) / 255.0f; r.y = ((p >> 8) & 255u) / 255.0f; r.z = ((p >> 16) & 255u) / 255.0f; r.w = (p >> 24) / 255.0f;
    return r;
}
PT_DEV F4 sampleTexture(const DTexture& tex, F2 uv)
{
    if (tex.components == 2) { F4 m; m.x = 1.0f; m.y = 0.0f; m.z = 1.0f; m.w = 1.0f; return m; }   /* two-channel images: the reference's magenta */
#ifdef PT_AUDIT_NOTEX           /* byte audit (tools/audit_bytes.sh): every lookup lands on one texel; results are not the reference's */
    uv.x = 0.25f; uv.y = 0.25f;
#endif
    uv.x = uv.x - __builtin_floorf(uv.x);
    uv.y = uv.y - __builtin_floorf(uv.y);
    float fx = uv.x * (tex.width - 1);
    float fy = uv.y * (tex.height - 1);
    int x0 = (int)__builtin_floorf(fx);
    int y0 = (int)__builtin_floorf(fy);
    int x1 = min(x0 + 1, tex.width - 1);
    int y1 = min(y0 + 1, tex.height - 1);
    float wx = fx - x0;
    float wy = fy - y0;
    const uint32_t* d = reinterpret_cast<const uint32_t*>(tex.data);
    const int r0 = texRowPart(tex, y0), r1 = texRowPart(tex, y1), c0 = texColPart(x0), c1 = texColPart(x1);
    const uint32_t q00 = d[r0 + c0], q10 = d[r0 + c1], q01 = d[r1 + c0], q11 = d[r1 + c1];
    F4 p00 = decodeTexel(q00), p10 = decodeTexel(q10), p01 = decodeTexel(q01), p11 = decodeTexel(q11);
    F4 p0 = p00 * (1.0f - wx) + p10 * wx;
    F4 p1 = p01 * (1.0f - wx) + p11 * wx;
    return p0 * (1.0f - wy) + p1 * wy;
}

/* sampleTexture of a material's base-colour and metallic-roughness textures at once (DTexPair): the same texel values and
 * the same arithmetic per texture, from four 8-byte loads that share a line more often than not */
PT_DEV void sampleTexturePair(const DTexPair& tp, F2 uv, F4& base, F4& mr)
{
#ifdef PT_AUDIT_NOTEX
    uv.x = 0.25f; uv.y = 0.25f;
#endif
    uv.x = uv.x - __builtin_floorf(uv.x);
    uv.y = uv.y - __builtin_floorf(uv.y);
    float fx = uv.x * (tp.width - 1);
    float fy = uv.y * (tp.height - 1);
    int x0 = (int)__builtin_floorf(fx);
    int y0 = (int)__builtin_floorf(fy);
    int x1 = min(x0 + 1, tp.width - 1);
    int y1 = min(y0 + 1, tp.height - 1);
    float wx = fx - x0;
    float wy = fy - y0;
    const int r0 = (y0 >> 2) * tp.tilesPerRow * 16 + ((y0 & 3) << 2), r1 = (y1 >> 2) * tp.tilesPerRow * 16 + ((y1 & 3) << 2);
    const int c0 = ((x0 >> 2) << 4) + (x0 & 3), c1 = ((x1 >> 2) << 4) + (x1 & 3);
    const uint2 q00 = tp.data[r0 + c0], q10 = tp.data[r0 + c1], q01 = tp.data[r1 + c0], q11 = tp.data[r1 + c1];
    {
        F4 p00 = decodeTexel(q00.x), p10 = decodeTexel(q10.x), p01 = decodeTexel(q01.x), p11 = decodeTexel(q11.x);
        F4 p0 = p00 * (1.0f - wx) + p10 * wx;
        F4 p1 = p01 * (1.0f - wx) + p11 * wx;
        base = p0 * (1.0f - wy) + p1 * wy;
    }
    {
        F4 p00 = decodeTexel(q00.y), p10 = decodeTexel(q10.y), p01 = decodeTexel(q01.y), p11 = decodeTexel(q11.y);
        F4 p0 = p00 * (1.0f - wx) + p10 * wx;
        F4 p1 = p01 * (1.0f - wx) + p11 * wx;
        mr = p0 * (1.0f - wy) + p1 * wy;
    }
}

PT_DEV F3 envTexel(const SceneView& S, int x, int y)
{
#ifdef PT_AUDIT_NOENV           /* byte audit: every lookup lands in one corner of the map */
    x &= 7; y &= 7;
#endif
    const float4 p = S.envData[(size_t)y * S.envWidth + x];
    return f3(p.x, p.y, p.z);
}
/* equirect bilinear lookup with clamp, no seam wrap (src/pathtrace.cu:1030-1075) */
PT_DEV F3 sampleEnv(const SceneView& S, F3 dir)
{
    if (!S.envEnabled) return f3(0.0f);
    float theta = dm::acos(nanMax(-1.0f, nanMin(1.0f, dir.y)));
    float phi = dm::atan2(dir.z, dir.x);
    float u = (phi + kPI) / (2.0f * kPI);
    float v = theta / kPI;
    u = nanMax(0.0f, nanMin(1.0f, u));
    v = nanMax(0.0f, nanMin(1.0f, v));
    float fx = u * (S.envWidth - 1);
    float fy = v * (S.envHeight - 1);
    int x0 = (int)__builtin_floorf(fx);
    int y0 = (int)__builtin_floorf(fy);
    int x1 = min(x0 + 1, S.envWidth - 1);
    int y1 = min(y0 + 1, S.envHeight - 1);
    float wx = fx - x0;
    float wy = fy - y0;
    F3 p00 = envTexel(S, x0, y0), p10 = envTexel(S, x1, y0), p01 = envTexel(S, x0, y1), p11 = envTexel(S, x1, y1);
    F3 p0 = p00 * (1.0f - wx) + p10 * wx;
    F3 p1 = p01 * (1.0f - wx) + p11 * wx;
    return p0 * (1.0f - wy) + p1 * wy;
}

/* point on an area light (src/pathtrace.cu:1091-1144) */
PT_DEV F3 sampleLightPoint(const DGeom& g, Lcg& rng)
{
    if (g.type == PT_SPHERE) {
        float u = rng.u01();
        float v = rng.u01();
        float theta = 2.0f * kPI * u;
        float phi = dm::acos(1.0f - 2.0f * v);
        float radius = g.scale[0] * 0.5f;
        F3 local = f3(radius * dm::sin(phi) * dm::cos(theta), radius * dm::sin(phi) * dm::sin(theta), radius * dm::cos(phi));
        return mulMV(g.T, local, 1.0f);
    }
    if (g.type == PT_CUBE) {
        float faceChoice = rng.u01() * 6.0f;
        int face = (int)faceChoice;
        float u = rng.u01() - 0.5f;
        float v = rng.u01() - 0.5f;
        F3 local;
        switch (face) {
        case 0: local = f3(0.5f, u, v); break;
        case 1: local = f3(-0.5f, u, v); break;
        case 2: local = f3(u, 0.5f, v); break;
        case 3: local = f3(u, -0.5f, v); break;
        case 4: local = f3(u, v, 0.5f); break;
        default: local = f3(u, v, -0.5f); break;
        }
        return mulMV(g.T, local, 1.0f);
    }
    return f3(g.translation);
}

/* first index with cdf[i] >= value (src/pathtrace.cu:1339-1354) */
PT_DEV int lowerBound(const float* cdf, int size, float value)
{
    int left = 0, right = size - 1;
    while (left < right) {
        int mid = (left + right) / 2;
        if (cdf[mid] < value) left = mid + 1; else right = mid;
    }
    return left;
}

/* lowerBound with a guide table.  For a CDF that never decreases (the host checks, and passes no guide otherwise) the
 * first entry >= value lies between the answers for the multiples of 1/kCdfGuide on either side of value; guide[j] is
 * the answer for j/kCdfGuide (16-bit entries), and value * kCdfGuide and j / kCdfGuide are exact in binary floating
 * point, so the bisection starts on a window of a handful of entries — one line of the CDF and one to three dependent
 * loads instead of the eleven, over six lines, that halving a 2 048-entry row takes.  (64 / 256 / 1 024 guide entries:
 * the shade stage takes 0.170 / 0.166 / 0.164 ms per sample on the dragon-class workload.)  Same index as
 * lowerBound(cdf, size, value), including size - 1 when no entry qualifies. */
#ifndef PT_CDF_GUIDE
#define PT_CDF_GUIDE 1024
#endif
constexpr int kCdfGuide = PT_CDF_GUIDE;       /* a power of two */
PT_DEV int lowerBoundGuided(const float* cdf, int size, const uint16_t* guide, float value)
{
    if (!guide) return lowerBound(cdf, size, value);
#ifdef PT_AUDIT_NOCDF           /* byte audit: no CDF is read */
    return (int)(value * (size - 1));
#endif
    int j = (int)(value * (float)kCdfGuide);
    j = max(0, min(j, kCdfGuide - 1));
    int left = guide[j], right = j + 1 < kCdfGuide ? (int)guide[j + 1] : size - 1;
    while (left < right) {
        int mid = (left + right) / 2;
        if (cdf[mid] < value) left = mid + 1; else right = mid;
    }
    return left;
}

/* env-map importance sample: row by marginal CDF, column by conditional CDF
 * (src/pathtrace.cu:1412-1472) */
PT_DEV F3 sampleEnvImportance(const SceneView& S, Lcg& rng, F3& outDir, float& outPdf)
{
    if (!S.envMarginal || !S.envConditional) {
        outDir = sampleHemisphere(f3(0, 1, 0), rng);
        outPdf = 1.0f / (2.0f * kPI);
        return sampleEnv(S, outDir);
    }
    float u = rng.u01();
    float v = rng.u01();
    int y = lowerBoundGuided(S.envMarginal, S.envHeight, S.envGuide, v);
    y = min(y, S.envHeight - 1);
    int x = lowerBoundGuided(S.envConditional + (size_t)y * S.envWidth, S.envWidth, S.envGuide ? S.envGuide + (size_t)(y + 1) * kCdfGuide : nullptr, u);
    x = min(x, S.envWidth - 1);
    float phi = (x + 0.5f) * 2.0f * kPI / S.envWidth - kPI;
    float theta = (y + 0.5f) * kPI / S.envHeight;
    float sinTheta = dm::sin(theta);
    float cosTheta = dm::cos(theta);
    outDir = normalize(f3(sinTheta * dm::cos(phi), cosTheta, sinTheta * dm::sin(phi)));
    F3 color = envTexel(S, x, y);
    float luminance = 0.299f * color.x + 0.587f * color.y + 0.114f * color.z;
    float pdfImage = luminance * sinTheta;
    if (S.envTotalPower > 0) pdfImage /= S.envTotalPower;
    float pixelArea = (2.0f * kPI / S.envWidth) * (kPI / S.envHeight);
    outPdf = pdfImage / (pixelArea * sinTheta);
    outPdf = nanMax(outPdf, 1e-6f);
    return color;
}

/* pdf of a direction under the env-map sampler (src/pathtrace.cu:1474-1511) */
PT_DEV float envPdf(const SceneView& S, F3 dir)
{
    if (!S.envMarginal || !S.envConditional) return 1.0f / (2.0f * kPI);
    float theta = dm::acos(nanMax(-1.0f, nanMin(1.0f, dir.y)));
    float phi = dm::atan2(dir.z, dir.x);
    float u = (phi + kPI) / (2.0f * kPI);
    float v = theta / kPI;
    int x = min((int)(u * S.envWidth), S.envWidth - 1);
    int y = min((int)(v * S.envHeight), S.envHeight - 1);
    F3 color = envTexel(S, x, y);
    float luminance = 0.299f * color.x + 0.587f * color.y + 0.114f * color.z;
    float sinTheta = dm::sin(theta);
    float pdfImage = luminance * sinTheta;
    if (S.envTotalPower > 0) pdfImage /= S.envTotalPower;
    float pixelArea = (2.0f * kPI / S.envWidth) * (kPI / S.envHeight);
    float pdf = pdfImage / (pixelArea * sinTheta);
    return nanMax(pdf, 1e-6f);
}

/* A shadow ray is blocked when ANY geom blocks it (the reference's loops stop at the first blocker,
 * src/pathtrace.cu:2174-2226 and :2271-2294), so the geoms can be tested in any order and in two places:
 * boxes, spheres and linear-scan meshes right here in the shader, BVH meshes later in the traversal
 * kernel.  These return true when no inline geom blocks the ray. */
template <bool COUNT>
PT_DEV bool lightVisibleInline(const SceneView& S, const Ray& sh, int lightGeom, float dist, TravCounters& cnt)
{
    for (int gi = 0; gi < S.numGeoms; ++gi) {
        if (gi == lightGeom) continue;
        const DGeom& g = S.geoms[gi];
        if (isBvhMesh(S, g)) continue;
        if (blocksLight(inlineOccluderDistance<COUNT>(S, g, sh, cnt), dist)) return false;
    }
    return true;
}
template <bool COUNT>
PT_DEV bool envVisibleInline(const SceneView& S, const Ray& sh, TravCounters& cnt)
{
    for (int i = 0; i < S.numGeoms; ++i) {
        const DGeom& g = S.geoms[i];
        if (isBvhMesh(S, g)) continue;
        if (blocksEnv(inlineOccluderDistance<COUNT>(S, g, sh, cnt))) return false;
    }
    return true;
}

/* a shadow ray whose BVH-mesh part is still open when the shader returns */
struct ShadowReq {
    bool pending;
    F3 o, d;
    float dist;
    int lightGeom;      /* -1: environment ray */
    F3 colorVisible;    /* path colour if nothing blocks; the path itself carries the blocked outcome */
};

/* ----------------------------------------------------------------- shaders */

struct Path {
    F3 o, d, color;
    int remaining;
    bool prevSpecular;
};

/* Lambert + one-sample direct light chosen among {light, BRDF, environment}
 * (src/pathtrace.cu:2101-2327) */
template <bool COUNT>
PT_DEV void shadeDiffuse(const SceneView& S, Path& p, float t, F3 normal, F3 materialColor, Lcg& rng, ShadowReq& sq, TravCounters& cnt)
{
    const int nl = S.numLights;
    const bool env = S.envEnabled != 0;
    F3 P = p.o + p.d * t;
    const float MIN_PDF = 1e-6f, MAX_CONTRIBUTION = 20.0f;
    F3 total = f3(0.0f);

    float choice = rng.u01();
    int strategy;
    if (nl > 0 && env) strategy = choice < 0.33f ? 0 : (choice < 0.66f ? 1 : 2);
    else if (nl > 0) strategy = choice < 0.5f ? 0 : 1;
    else if (env) strategy = choice < 0.5f ? 1 : 2;
    else strategy = 1;

    if (strategy == 0 && nl > 0) {
        int li = min((int)(rng.u01() * nl), nl - 1);
        DLight L = S.lights[li];
        const DGeom& lg = S.geoms[L.geomIdx];
        const PtMaterial& lm = S.materials[lg.materialid];
        F3 lp = sampleLightPoint(lg, rng);
        F3 wi = lp - P;
        float dist = length(wi);
        if (dist > 0.01f) {
            wi = wi / dist;
            float NdotL = dot(normal, wi);
            if (NdotL > 0.0f) {
                Ray sh; sh.o = P + normal * 0.001f; sh.d = wi;
                if (lightVisibleInline<COUNT>(S, sh, L.geomIdx, dist, cnt)) {
                    if (S.numBvhMeshes > 0 && mayHitBvhMeshes(S, sh, dist * 1.001f)) { sq.pending = true; sq.o = sh.o; sq.d = sh.d; sq.dist = dist; sq.lightGeom = L.geomIdx; }
                    F3 ln = normalize(lp - f3(lg.translation));
                    float NdotLl = nanMax(0.0f, dot(-wi, ln));
                    float pdfLight = 1.0f / (nanMax(L.area, 0.01f) * nl);
                    float pdfBRDF = NdotL / kPI;
                    float pdfE = env ? envPdf(S, wi) : 0.0f;
                    float sum = pdfLight + pdfBRDF + pdfE;
                    float weight = pdfLight / nanMax(sum, MIN_PDF);
                    F3 Le = f3(lm.color) * lm.emittance;
                    F3 f = materialColor / kPI;
                    float G = NdotL * NdotLl / (dist * dist);
                    float sp = (nl > 0 && env) ? 0.33f : ((nl > 0 || env) ? 0.5f : 1.0f);
                    total = total + weight * Le * f * G * (float)nl / (sp * nanMax(pdfLight, MIN_PDF));
                }
            }
        }
    } else if (strategy == 2 && env) {
        F3 ed; float ep;
        F3 ec = sampleEnvImportance(S, rng, ed, ep);
        float NdotL = dot(normal, ed);
        if (NdotL > 0.0f) {
            Ray sh; sh.o = P + normal * 0.001f; sh.d = ed;
            if (envVisibleInline<COUNT>(S, sh, cnt)) {
                if (S.numBvhMeshes > 0 && mayHitBvhMeshes(S, sh, __builtin_inff())) { sq.pending = true; sq.o = sh.o; sq.d = sh.d; sq.dist = 0.0f; sq.lightGeom = -1; }
                float pdfBRDF = NdotL / kPI;
                float sum = ep + pdfBRDF;
                float weight = ep / nanMax(sum, MIN_PDF);
                F3 f = materialColor / kPI;
                float sp = (nl > 0 && env) ? 0.33f : 0.5f;
                total = total + weight * ec * f * NdotL / (sp * nanMax(ep, MIN_PDF));
            }
        }
    }
    F3 dir = sampleHemisphere(normal, rng);
    total = vclamp(total, f3(0.0f), f3(MAX_CONTRIBUTION));
    if (sq.pending) {       /* both outcomes of the open shadow ray; the blocked one adds the same clamped zero */
        sq.colorVisible = p.color * (materialColor + total);
        total = vclamp(f3(0.0f), f3(0.0f), f3(MAX_CONTRIBUTION));
    }
    p.color = p.color * (materialColor + total);
    p.o = P + normal * 0.001f;
    p.d = dir;
}

/* ideal mirror (src/pathtrace.cu:2329-2348) */
PT_DEV void shadeMirror(Path& p, float t, F3 normal, F3 materialColor)
{
    F3 in = p.d;
    F3 refl = normalize(in - 2.0f * dot(in, normal) * normal);
    p.color = p.color * materialColor;
    F3 P = p.o + p.d * t;
    p.o = P + normal * 0.001f;
    p.d = refl;
}

/* dielectric with Schlick reflectance, TIR handled first (src/pathtrace.cu:2350-2415) */
PT_DEV void shadeDielectric(Path& p, float t, F3 normal, F3 materialColor, float ior, Lcg& rng)
{
    F3 in = normalize(p.d);
    float cosTheta = dot(in, normal);
    bool entering = cosTheta < 0.0f;
    float etaI, etaT;
    if (entering) { etaI = 1.0f; etaT = ior; cosTheta = -cosTheta; }
    else { etaI = ior; etaT = 1.0f; normal = -normal; }
    float eta = etaI / etaT;
    float sin2T = eta * eta * (1.0f - cosTheta * cosTheta);
    F3 nd;
    if (sin2T > 1.0f) {
        nd = normalize(in - 2.0f * dot(in, normal) * normal);
    } else {
        float cosT = __builtin_sqrtf(1.0f - sin2T);
        float c = entering ? cosTheta : cosT;
        float r0 = (1.0f - eta) / (1.0f + eta);
        r0 = r0 * r0;
        float fres = r0 + (1.0f - r0) * dm::pow5(1.0f - c);
        if (rng.u01() < fres) nd = reflect(in, normal);
        else nd = normalize(eta * in + (eta * cosTheta - cosT) * normal);
    }
    p.color = p.color * materialColor;
    F3 P = p.o + p.d * t;
    p.o = P + nd * 0.001f;
    p.d = nd;
}

/* GGX terms (src/pathtrace.cu:1278-1337) */
PT_DEV float ggxD(F3 n, F3 h, float rough)
{
    float a = rough * rough;
    float a2 = a * a;
    float nh = nanMax(0.0f, dot(n, h));
    float nh2 = nh * nh;
    float den = nh2 * (a2 - 1.0f) + 1.0f;
    den = kPI * den * den;
    return a2 / nanMax(0.0001f, den);
}
PT_DEV float ggxG1(F3 n, F3 v, float rough)
{
    float a = rough * rough;
    float k = a / 2.0f;
    float nv = nanMax(0.0f, dot(n, v));
    float den = nv * (1.0f - k) + k;
    return nv / nanMax(0.0001f, den);
}
PT_DEV F3 schlick(float c, F3 F0) { return F0 + (f3(1.0f) - F0) * dm::pow5(1.0f - c); }
PT_DEV F3 sampleGgxHalf(F3 normal, float rough, Lcg& rng)
{
    float u = rng.u01();
    float v = rng.u01();
    float a = rough * rough;
    float phi = 2.0f * kPI * u;
    float ct = __builtin_sqrtf((1.0f - v) / (1.0f + (a * a - 1.0f) * v));
    float st = __builtin_sqrtf(1.0f - ct * ct);
    F3 ht = f3(st * dm::cos(phi), st * dm::sin(phi), ct);
    F3 up = __builtin_fabsf(normal.z) < 0.999f ? f3(0.0f, 0.0f, 1.0f) : f3(1.0f, 0.0f, 0.0f);
    F3 tx = normalize(cross(up, normal));
    F3 ty = cross(normal, tx);
    return tx * ht.x + ty * ht.y + normal * ht.z;
}
/* Cook-Torrance specular + Lambert diffuse for one light direction (src/pathtrace.cu:1898-1918) */
PT_DEV F3 pbrBrdf(F3 n, F3 wo, F3 wi, float NdotL, float rough, float metallic, F3 albedo, F3 F0, float& D, float& NdotH, float& VdotH)
{
    F3 h = normalize(wi + wo);
    float NdotV = nanMax(0.0f, dot(n, wo));
    NdotH = nanMax(0.0f, dot(n, h));
    VdotH = nanMax(0.0f, dot(wo, h));
    D = ggxD(n, h, rough);
    float G = ggxG1(n, wo, rough) * ggxG1(n, wi, rough);
    F3 F = schlick(VdotH, F0);
    F3 num = D * G * F;
    float den = 4.0f * NdotV * NdotL;
    F3 spec = num / nanMax(0.001f, den);
    F3 kD = f3(1.0f) - F;
    kD = kD * (1.0f - metallic);
    F3 diff = kD * albedo / kPI;
    return diff + spec;
}

/* dipole diffusion profile (src/pathtrace.cu:1518-1543) */
PT_DEV F3 dipoleProfile(float r, F3 sa, F3 ssp)
{
    F3 stp = sa + ssp;
    F3 str = vsqrt(3.0f * sa * stp);
    F3 zr = f3(1.0f) / stp;
    F3 zv = zr * (1.0f + 4.0f / 3.0f * f3(1.44f));
    F3 dr = vsqrt(zr * zr + r * r);
    F3 dv = vsqrt(zv * zv + r * r);
    F3 C = f3(0.25f) / kPI;
    F3 er = -str * dr, ev = -str * dv;
    F3 xr = f3(dm::exp(er.x), dm::exp(er.y), dm::exp(er.z));
    F3 xv = f3(dm::exp(ev.x), dm::exp(ev.y), dm::exp(ev.z));
    F3 res = C * (zr * (str + f3(1.0f) / dr) * xr / (dr * dr) + zv * (str + f3(1.0f) / dv) * xv / (dv * dv));
    return vclamp(res, f3(0.0f), f3(1.0f));
}

/* subsurface random-walk step of the PBR shader (src/pathtrace.cu:1546-1702); true when taken */
PT_DEV bool shadeSubsurface(Path& p, F3 P, F3 normal, const PtMaterial& m, F3 materialColor, Lcg& rng)
{
    if (m.subsurfaceEnabled == 0 || m.metallic > 0.5f) return false;
    float prob = 0.3f * (1.0f - m.metallic);
    prob = selClamp(prob, 0.1f, 0.3f);
    if (rng.u01() > prob) return false;

    /* exit point */
    float avg = (m.subsurfaceRadiusRGB[0] + m.subsurfaceRadiusRGB[1] + m.subsurfaceRadiusRGB[2]) / 3.0f;
    avg *= m.subsurfaceScale;
    avg = selClamp(avg, 0.001f, 1.0f);
    float u = rng.u01();
    float dist = -dm::log(1.0f - u * 0.9f) * avg;
    dist = selClamp(dist, 0.001f, avg * 3.0f);
    float theta = 2.0f * kPI * rng.u01();
    float phi = dm::acos(1.0f - rng.u01());
    F3 ld = f3(dm::sin(phi) * dm::cos(theta), dm::sin(phi) * dm::sin(theta), dm::cos(phi));
    F3 tg = (__builtin_fabsf(normal.x) > __builtin_fabsf(normal.y)) ? normalize(f3(-normal.z, 0, normal.x)) : normalize(f3(0, -normal.z, normal.y));
    F3 bt = cross(normal, tg);
    F3 wd = tg * ld.x + bt * ld.y - normal * ld.z;
    F3 exitP = P + normalize(wd) * dist;
    float distance = length(exitP - P);

    /* coefficients */
    F3 A = f3(m.subsurfaceColor);
    F3 safeR = vmax(f3(m.subsurfaceRadiusRGB) * m.subsurfaceScale, f3(0.001f));
    F3 ssp = f3(1.0f) / safeR;
    F3 sa = ssp * (f3(1.0f) - A) * 0.01f;
    sa = vclamp(sa, f3(0.001f), f3(10.0f));
    ssp = vclamp(ssp, f3(0.1f), f3(100.0f));

    F3 bssrdf = dipoleProfile(distance, sa, ssp);
    F3 e = -sa * distance;
    F3 tr = f3(dm::exp(e.x), dm::exp(e.y), dm::exp(e.z));
    tr = vclamp(tr, f3(0.0f), f3(1.0f));
    F3 thr = bssrdf * tr * A;
    thr = thr * mix(f3(1.0f), materialColor, 0.5f);
    thr = thr * (1.0f / prob);
    float mx = nanMax(nanMax(thr.x, thr.y), thr.z);
    if (mx > 2.0f) thr = thr * (2.0f / mx);
    p.color = p.color * thr;

    float r1 = rng.u01();
    float r2 = rng.u01();
    float st = __builtin_sqrtf(r1);
    float ct = __builtin_sqrtf(1.0f - r1);
    float ph = 2.0f * kPI * r2;
    F3 l2 = f3(st * dm::cos(ph), st * dm::sin(ph), ct);
    F3 w2 = tg * l2.x + bt * l2.y + normal * l2.z;     /* same tangent frame: exit normal == entry normal */
    p.o = exitP + normal * 0.001f;
    p.d = normalize(w2);
    p.remaining--;
    return true;
}

/* metallic-roughness shader: optional subsurface step, stochastic transmission, one-sample
 * direct light with the reference's weights, GGX / cosine lobe pick (src/pathtrace.cu:1708-2099) */
template <bool COUNT>
PT_DEV void shadePbr(const SceneView& S, Path& p, float t, F3 normal, const PtMaterial& m, float roughIn, float metallic,
                     F3 albedo, float texAlpha, Lcg& rng, ShadowReq& sq, TravCounters& cnt)
{
    const int nl = S.numLights;
    const bool env = S.envEnabled != 0;
    F3 P = p.o + p.d * t;

    if (m.subsurfaceEnabled > 0 && metallic < 0.5f) {
        PtMaterial mm = m;
        mm.metallic = metallic;
        if (shadeSubsurface(p, P, normal, mm, albedo, rng)) return;
    }
    F3 wo = -p.d;
    float rough = selClamp(roughIn, 0.02f, 1.0f);

    float transp = m.transparency;
    if (texAlpha < 1.0f) {
        float tt = 1.0f - texAlpha;
        transp = 1.0f - ((1.0f - transp) * (1.0f - tt));
    }
    if (transp > 0.0f && rng.u01() < transp) {
        float ior = m.indexOfRefraction > 0 ? m.indexOfRefraction : 1.5f;
        bool entering = dot(normal, wo) > 0;
        F3 n = entering ? normal : -normal;
        float eta = entering ? (1.0f / ior) : ior;
        float c = dot(n, wo);
        float k = 1.0f - eta * eta * (1.0f - c * c);
        F3 nd;
        if (k < 0.0f || rough > 0.8f) nd = p.d;
        else nd = normalize(eta * (-wo) + (eta * c - __builtin_sqrtf(k)) * n);
        p.o = P - n * 0.001f;
        if (rough > 0.1f && rough < 0.8f) {
            float sx = rng.u01() - 0.5f;            /* drawn in x, y, z order */
            float sy = rng.u01() - 0.5f;
            float sz = rng.u01() - 0.5f;
            F3 sc = f3(sx, sy, sz) * rough * 0.2f;
            nd = normalize(nd + sc);
        }
        p.d = nd;
        p.color = p.color * mix(f3(1.0f), albedo, 1.0f - transp);
        return;
    }

    F3 F0 = mix(f3(0.04f), albedo, metallic);
    const float MIN_PDF = 1e-6f, MAX_CONTRIBUTION = 10.0f;
    F3 direct = f3(0.0f);

    float choice = rng.u01();
    int strategy = 1;
    if (nl > 0 && env) strategy = choice < 0.33f ? 0 : (choice < 0.66f ? 1 : 2);
    else if (nl > 0 && !env) strategy = choice < 0.5f ? 0 : 1;
    else if (nl == 0 && env) strategy = choice < 0.5f ? 1 : 2;

    if (strategy == 0 && nl > 0) {
        int li = min((int)(rng.u01() * nl), nl - 1);
        DLight L = S.lights[li];
        const DGeom& lg = S.geoms[L.geomIdx];
        const PtMaterial& lm = S.materials[lg.materialid];
        F3 lp = sampleLightPoint(lg, rng);
        F3 wi = lp - P;
        float dist = length(wi);
        if (dist > 0.01f) {
            wi = wi / dist;
            float NdotL = dot(normal, wi);
            if (NdotL > 0.0f) {
                Ray sh; sh.o = P + normal * 0.001f; sh.d = wi;
                if (lightVisibleInline<COUNT>(S, sh, L.geomIdx, dist, cnt)) {
                    if (S.numBvhMeshes > 0 && mayHitBvhMeshes(S, sh, dist * 1.001f)) { sq.pending = true; sq.o = sh.o; sq.d = sh.d; sq.dist = dist; sq.lightGeom = L.geomIdx; }
                    float D, NdotH, VdotH;
                    F3 brdf = pbrBrdf(normal, wo, wi, NdotL, rough, metallic, albedo, F0, D, NdotH, VdotH);
                    F3 ln = normalize(lp - f3(lg.translation));
                    float NdotLl = nanMax(0.0f, dot(-wi, ln));
                    float pdfLight = 1.0f / (nanMax(L.area, 0.01f) * nl);
                    float pdfBRDF = (NdotL / kPI) * (1.0f - metallic) + (D * NdotH / (4.0f * VdotH)) * metallic;
                    float pdfE = env ? envPdf(S, wi) : 0.0f;
                    float sum = pdfLight + pdfBRDF + pdfE;
                    float weight = pdfLight / nanMax(sum, MIN_PDF);
                    F3 Le = f3(lm.color) * lm.emittance;
                    float G = NdotL * NdotLl / (dist * dist);
                    float sp = (nl > 0 && env) ? 0.33f : ((nl > 0 || env) ? 0.5f : 1.0f);
                    direct = direct + weight * brdf * Le * G * (float)nl / (sp * nanMax(pdfLight, MIN_PDF));
                }
            }
        }
    } else if (strategy == 2 && env) {
        F3 ed; float ep;
        F3 ec = sampleEnvImportance(S, rng, ed, ep);
        float NdotL = dot(normal, ed);
        if (NdotL > 0.0f) {
            Ray sh; sh.o = P + normal * 0.001f; sh.d = ed;
            if (envVisibleInline<COUNT>(S, sh, cnt)) {
                if (S.numBvhMeshes > 0 && mayHitBvhMeshes(S, sh, __builtin_inff())) { sq.pending = true; sq.o = sh.o; sq.d = sh.d; sq.dist = 0.0f; sq.lightGeom = -1; }
                float D, NdotH, VdotH;
                F3 brdf = pbrBrdf(normal, wo, ed, NdotL, rough, metallic, albedo, F0, D, NdotH, VdotH);
                float pdfBRDF = (NdotL / kPI) * (1.0f - metallic) + (D * NdotH / (4.0f * VdotH)) * metallic;
                float sum = ep + pdfBRDF;
                float weight = ep / nanMax(sum, MIN_PDF);
                float sp = (nl > 0 && env) ? 0.33f : (env ? 0.5f : 1.0f);
                direct = direct + weight * brdf * ec * NdotL / (sp * nanMax(ep, MIN_PDF));
            }
        }
    }
    direct = vclamp(direct, f3(0.0f), f3(MAX_CONTRIBUTION));
    F3 directVisible = direct;
    if (sq.pending) direct = vclamp(f3(0.0f), f3(0.0f), f3(MAX_CONTRIBUTION));
    sq.colorVisible = f3(0.0f);

    float specProb = 0.5f + 0.5f * metallic;
    if (rng.u01() < specProb) {
        F3 h = sampleGgxHalf(normal, rough, rng);
        F3 wi = reflect(-wo, h);
        if (dot(wi, normal) > 0.0f) {
            float VdotH = nanMax(0.0f, dot(wo, h));
            F3 F = schlick(VdotH, F0);
            F3 sc = mix(f3(1.0f), albedo, metallic);
            if (sq.pending) sq.colorVisible = p.color * ((sc * F / specProb) + directVisible);
            p.color = p.color * ((sc * F / specProb) + direct);
            p.o = P + normal * 0.001f;
            p.d = wi;
            p.prevSpecular = true;
        } else {
            p.remaining = 0;
            p.color = f3(0.0f);
        }
    } else {
        if (metallic < 1.0f) {
            F3 wi = sampleHemisphere(normal, rng);
            F3 dc = albedo * (1.0f - metallic);
            if (sq.pending) sq.colorVisible = p.color * ((dc / (1.0f - specProb)) + directVisible);
            p.color = p.color * ((dc / (1.0f - specProb)) + direct);
            p.o = P + normal * 0.001f;
            p.d = wi;
            p.prevSpecular = false;
        } else {
            p.remaining = 0;
            p.color = f3(0.0f);
        }
    }
    p.remaining--;
}

/* one queue slot of the shade stage: Russian roulette, textures, emission, material
 * dispatch, miss / environment (shadeMaterialMIS, src/pathtrace.cu:2455-2626) */
template <bool COUNT>
PT_DEV void shadeSlot(const SceneView& S, int iter, int bounceOneBased, int slot, Path& p, float t, F3 normal, int matId, F2 uv,
                      bool firstBounce, ShadowReq& sq, TravCounters& cnt)
{
    if (p.remaining <= 0) return;
    Lcg rng;
    rng.seed(iter, slot, p.remaining);

    if (bounceOneBased >= 3) {          /* RR_START_BOUNCE (src/sceneStructs.h:12) */
        float lum = 0.2126f * p.color.x + 0.7152f * p.color.y + 0.0722f * p.color.z;
        float survive = selClamp(lum, 0.05f, 0.95f);
        if (rng.u01() >= survive) { p.remaining = 0; p.color = f3(0.0f); return; }
        p.color = p.color / survive;
    }

    if (t > 0.0f) {
        const PtMaterial& m = S.materials[matId];
        F3 mc = f3(m.color);
        float alpha = 1.0f;
        float metallic = m.metallic, rough = m.roughness;
        const DTexPair* pair = S.texPairs ? &S.texPairs[matId] : nullptr;
        if (pair && pair->data) {
            /* both textures present and of one size: one footprint serves both */
            F4 c, mr;
            sampleTexturePair(*pair, uv, c, mr);
            mc = f3(c.x, c.y, c.z);
            alpha = c.w;
            mc = mc * f3(m.color);
            metallic *= mr.z;
            rough *= mr.y;
        } else {
            if (m.baseColorTextureIdx >= 0 && m.baseColorTextureIdx < S.numTextures) {
                F4 c = sampleTexture(S.textures[m.baseColorTextureIdx], uv);
                mc = f3(c.x, c.y, c.z);
                alpha = c.w;
                mc = mc * f3(m.color);
            }
            if (m.metallicRoughnessTextureIdx >= 0 && m.metallicRoughnessTextureIdx < S.numTextures) {
                F4 mr = sampleTexture(S.textures[m.metallicRoughnessTextureIdx], uv);
                metallic *= mr.z;
                rough *= mr.y;
            }
        }
        if (m.emissiveTextureIdx >= 0 && m.emissiveTextureIdx < S.numTextures) {
            F4 e4 = sampleTexture(S.textures[m.emissiveTextureIdx], uv);
            F3 em = f3(e4.x, e4.y, e4.z);
            if (length(em) > 0.0f) {
                p.color = p.color * (em * f3(m.emissiveFactor));
                p.remaining = 0;
                return;
            }
        }
        if (m.emittance > 0.0f) {
            p.color = p.color * (mc * m.emittance);
            p.remaining = 0;
            return;
        }
        p.remaining--;
        if (p.remaining <= 0) { p.color = f3(0.0f); return; }

        switch (m.type) {
        case PT_DIFFUSE:
            p.prevSpecular = false;
            shadeDiffuse<COUNT>(S, p, t, normal, mc, rng, sq, cnt);
            break;
        case PT_SPECULAR:
            p.prevSpecular = true;
            shadeMirror(p, t, normal, mc);
            break;
        case PT_REFRACTIVE:
            p.prevSpecular = true;
            shadeDielectric(p, t, normal, mc, m.indexOfRefraction, rng);
            break;
        case PT_PBR:
            p.prevSpecular = (rough < 0.1f && metallic > 0.5f);
            shadePbr<COUNT>(S, p, t, normal, m, rough, metallic, mc, alpha, rng, sq, cnt);
            break;
        default:
            p.prevSpecular = false;
            break;
        }
    } else {
        if (S.envEnabled) {
            F3 ec = sampleEnv(S, p.d);
            if (firstBounce) p.color = p.color * ec;
            else if (p.prevSpecular) p.color = p.color * ec;
            else p.color = p.color * (ec * 0.5f);
        } else {
            p.color = p.color * f3(0.0f);
        }
        p.remaining = 0;
    }
}

} // namespace ptd
