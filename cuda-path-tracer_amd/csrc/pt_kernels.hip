/*
 * pt_kernels.hip — gfx950 kernels of the wavefront loop and their launchers.
 *
 * A launch group holds up to 512 iterations (= samples per pixel), one grid row each; per bounce b
 * every row runs:
 *   k_trace<closest>  persistent BVH traversal over the rays filed for it (by k_camera for b = 0, by
 *                     k_shade afterwards): closest mesh hit by pixel
 *   k_prims           boxes, spheres, linear-scan meshes folded in; final hit record by pixel, sort
 *                     bucket by slot, this workgroup's bucket histogram
 *   k_sort_scan       exclusive scan of the (bucket-major, workgroup-minor) histogram
 *   k_sort_scatter    stable counting sort of the 4-byte pixel queue by material bucket
 *   k_shade           shades slot i = sorted position (the index the RNG is seeded with), adds
 *                     terminated paths to the image, flags survivors in the queue, files the new
 *                     ray for the next traversal and open shadow rays for the shadow traversal
 *   k_trace<shadow>   BVH part of the shadow rays; settles the path colour
 *   k_compact_scan / k_compact_scatter   order-preserving compaction of the queue
 * All launches use a fixed grid and read their item counts from device memory, so the host never
 * waits for a count (the reference synchronises >= 5 times per bounce, src/pathtrace.cu:2729-2879).
 *
 * Wave64 throughout: ranks come from 64-bit ballots + popcount, block-level carries go through
 * LDS, the per-ray traversal stack is a [32][256] LDS array (one conflict-free column per lane).
 */
#include "pt_scene_dev.h"

#include <cstdlib>

namespace ptd {

PT_DEV int laneId() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
PT_DEV unsigned long long laneMaskLt() { return (1ull << laneId()) - 1ull; }

/* Workgroup-wide append: every thread with flag set gets a distinct index of the list whose length
 * *counter holds (one atomic per workgroup); -1 for the others.  s_tmp: kBlock / 64 + 1 ints of LDS.
 * All threads of the workgroup must call it. */
PT_DEV int blockAppendIndex(bool flag, uint32_t* counter, int* s_tmp)
{
    const unsigned long long m = __ballot(flag);
    const int wave = threadIdx.x >> 6;
    if (laneId() == 0) s_tmp[wave] = (int)__builtin_popcountll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) { int c = s_tmp[w]; s_tmp[w] = total; total += c; }
        s_tmp[kBlock / 64] = total ? (int)atomicAdd(counter, (uint32_t)total) : 0;
    }
    __syncthreads();
    const int idx = s_tmp[kBlock / 64] + s_tmp[wave] + (int)__builtin_popcountll(m & laneMaskLt());
    __syncthreads();
    return flag ? idx : -1;
}

/* The light one-thread-per-slot kernels (sort scatter, compaction, queue copies) are launched with a grid that
 * is bounded by the machine, not by the image: a workgroup walks "virtual" workgroups vb = blockIdx.x,
 * blockIdx.x + gridDim.x, ... while they hold live slots.  Late bounces have few live paths; a grid sized for
 * the whole image would spend its time launching workgroups that exit at once (a quarter of a million per
 * kernel for 32 iterations at 1080p, about 55 us).  The heavy kernels (k_prims, k_shade) come in two forms: one
 * workgroup per 256 slots (on a bounded grid a crowded bounce loses more than the launches cost), and a SPARSE
 * form on a bounded grid that the host picks for bounces it knows to be nearly empty. */
#define PT_FOR_VIRTUAL_BLOCKS(vb, n) for (int vb = (int)blockIdx.x; (uint32_t)(vb * kBlock) < (n); vb += (int)gridDim.x)

PT_DEV void flushCounters(const TravCounters& c, DevCounters* dc, int stage)
{
    /* one atomic per wave and counter */
    unsigned n = c.nodes, t = c.tris, h = c.meshHits, b = c.boxes;
    atomicMax(&dc->maxNodes[stage], n);
    atomicMax(&dc->maxTris[stage], t);
    atomicMax(&dc->maxStack[stage], c.maxSp);
    for (int off = 32; off > 0; off >>= 1) {
        n += __shfl_down(n, off);
        t += __shfl_down(t, off);
        h += __shfl_down(h, off);
        b += __shfl_down(b, off);
    }
    if (laneId() == 0) {
        if (n) atomicAdd(&dc->trav[stage][0], (unsigned long long)n);
        if (t) atomicAdd(&dc->trav[stage][1], (unsigned long long)t);
        if (h) atomicAdd(&dc->trav[stage][2], (unsigned long long)h);
        if (b) atomicAdd(&dc->trav[stage][3], (unsigned long long)b);
    }
}

/* ------------------------------------------------------------------ camera
 * One thread per pixel: 2x2 stratified jitter cell = iter % 4, two draws from the
 * (iter, pixel, 0) stream, pinhole direction (generateRayFromCamera,
 * src/pathtrace.cu:694-745).  Threads walk the image in 8x8 tiles (a wave = one tile, a workgroup
 * = a strip of four tiles, the grid = strips across x strips down x iterations: no division
 * to find a pixel), so the work list of the first traversal is filed tile by tile and the 64 rays
 * a traversal wave picks up together cross the same part of the scene; everything a pixel owns
 * (RNG stream, state slot) is addressed by its row-major index as before. */
__global__ void __launch_bounds__(kBlock) k_camera(SceneView S, PtCamera cam, Batch bt, int traceDepth, PathState st, Queues q, uint32_t* traceCount)
{
    __shared__ int s_tmp[kBlock / 64 + 1];
    const int row = (int)blockIdx.z;
    st = st.at(row);
    uint32_t* live = q.at(row).live;
    const int iter = bt.iterBase + row * bt.iterStride;
    const int W = cam.resolution[0], H = cam.resolution[1];
    const int t = threadIdx.x;
    const int x = ((int)blockIdx.x << 5) + ((t >> 6) << 3) + (t & 7), y = ((int)blockIdx.y << 3) + ((t >> 3) & 7);
    const int index = y * W + x;
    if (t == 0 && blockIdx.x == 0 && blockIdx.y == 0) live[0] = (uint32_t)(W * H);
    bool cand = false;
    if (x < W && y < H) {
    Lcg rng;
    rng.seed(iter, index, 0);
    const int cell = iter % 4;
    const int cellX = cell % 2, cellY = cell / 2;
    const float cellSize = 1.0f / 2.0f;
    float jx = (cellX + rng.u01()) * cellSize;
    float jy = (cellY + rng.u01()) * cellSize;
    float px = (float)x + jx - 0.5f;
    float py = (float)y + jy - 0.5f;
    F3 dir = normalize(f3(cam.view) - f3(cam.right) * cam.pixelLength[0] * (px - (float)W * 0.5f)
                       - f3(cam.up) * cam.pixelLength[1] * (py - (float)H * 0.5f));
    /* work list of the first closest-hit traversal (see k_cull); the verdict travels with the ray (rayD.w) */
    if (S.numBvhMeshes > 0) {
        Ray r; r.o = f3(cam.position); r.d = dir;
        cand = mayHitBvhMeshes(S, r, __builtin_inff(), true);
    }
    const float4 dRec = make_float4(dir.x, dir.y, dir.z, __int_as_float(cand ? 1 : 0));
    if (S.cameraOrigin) st.rayD0[index] = dRec;
    else st.rayD[index] = dRec;
    if (!S.cameraOrigin) {      /* otherwise bounce 0 takes these from the scene view (loadRayO / loadCol) */
        st.rayO[index] = make_float4(cam.position[0], cam.position[1], cam.position[2], __int_as_float(0));
        st.col[index] = make_float4(1.0f, 1.0f, 1.0f, __int_as_float(traceDepth));
    }
    }
    if (S.numBvhMeshes > 0) {
        const int at = blockAppendIndex(cand, traceCount + row * kCtrStride, s_tmp);
        if (at >= 0) st.traceQueue[at] = index;
    }
}

/* --------------------------------------------------------------- intersect
 * The closest hit of a ray is found in two kernels.
 *
 * k_trace (persistent) walks the BVH meshes.  Per-ray cost there is heavy-tailed (a ray that
 * misses the mesh box takes one step, a grazing ray hundreds), so a wave that walks 64 fixed rays
 * in lockstep idles most lanes.  Here every lane owns its traversal state and pulls the next
 * queue slot from a device counter as soon as its ray is finished; finished lanes are refilled
 * together once enough of the wave is idle, so set-up and write-back code runs with many lanes.
 * The grid is sized to the machine (not to the ray count) and every wave leaves its loop when the
 * counter has passed the item count.  The same kernel, in shadow mode, finishes the shadow rays the
 * shade stage left open.
 *
 * k_prims (one thread per slot, coherent) folds boxes, spheres and linear-scan meshes into the
 * mesh result, writes the final hit record and the sort bucket, and builds this workgroup's bucket
 * histogram.
 *
 * Per geom the test is the reference's (computeIntersectionsBVH, src/pathtrace.cu:834-934); the
 * closest of all geoms is order-independent once equal distances go to the lower geom index, which
 * is what the reference's in-order strict '<' loop yields. */
#ifndef PT_REFILL
#define PT_REFILL 24
#endif
/* Registers: the traversal kernel is held to 96 (five waves per SIMD by the register file) although its persistent grid
 * only fills four — the room that leaves on every SIMD is what lets the kernels of other stages and streams (the shadow
 * traversal of the previous bounce, the primitive stage, the sort) run beside it.  The shade kernel fits six. */
#ifndef PT_TRACE_MINWAVES
#define PT_TRACE_MINWAVES 5
#endif
#ifndef PT_SHADE_MINWAVES
#define PT_SHADE_MINWAVES 6
#endif
#ifndef PT_LEAF_WAIT
#define PT_LEAF_WAIT 20
#endif
constexpr int kRefillThreshold = PT_REFILL;
constexpr int kSegments = kFetchSegments;      /* pieces of a work queue = XCDs (private L2s) of the device */
constexpr int kLeafWait = PT_LEAF_WAIT;

PT_DEV int nextBvhMesh(const SceneView& S, int from)
{
    for (int i = from; i < S.numGeoms; ++i)
        if (isBvhMesh(S, S.geoms[i])) return i;
    return -1;
}

enum { TRACE_CLOSEST = 0, TRACE_SHADOW = 1 };

/* Work list of the closest-hit traversal: rays that may hit a BVH mesh go to the trace queue (one
 * atomic per workgroup; the order is free, results are stored by pixel) and are marked in rayD.w; k_prims
 * does not look for a mesh result for the others.  Most rays of a typical bounce never come near a
 * mesh; keeping them out of the traversal kernel keeps its lanes busy with rays that actually walk a
 * tree.  Inside the loop the stage that creates a ray (k_camera, k_shade) files it; this kernel does it
 * for rays that arrive from outside (the fixed-ray-set hook). */
__global__ void __launch_bounds__(kBlock) k_cull(SceneView S, PathState st, Queues q, int bounce, uint32_t* workCount)
{
    __shared__ int s_tmp[kBlock / 64 + 1];
    st = st.at(blockIdx.y);
    q = q.at(blockIdx.y);
    const uint32_t n = q.live[bounce];
    if ((uint32_t)(blockIdx.x * kBlock) >= n) return;
    const int slot = blockIdx.x * kBlock + threadIdx.x;
    bool cand = false;
    int pix = 0;
    if ((uint32_t)slot < n) {
        pix = bounce == 0 ? slot : q.q0[slot];
        float4 o = st.rayO[pix], d = st.rayD[pix];
        Ray r; r.o = f3(o.x, o.y, o.z); r.d = f3(d.x, d.y, d.z);
        cand = mayHitBvhMeshes(S, r, __builtin_inff(), true);
        st.rayD[pix] = make_float4(d.x, d.y, d.z, __int_as_float(cand ? 1 : 0));
    }
    const int idx = blockAppendIndex(cand, workCount + blockIdx.y * kCtrStride, s_tmp);
    if (idx >= 0) st.traceQueue[idx] = pix;
}

/* items: work queue entries [0, workCount[row]) */
/* Settling a finished shadow ray (both traversal kernels): the path takes the colour of the unoccluded outcome if nothing
 * blocked, and a path that has ended makes its one contribution to the iteration's image, which the shade stage held back. */
PT_DEV void settleShadow(const PathState& st, int pix, int shSlot, bool blocked)
{
    float4 c = st.col[pix];
    if (!blocked) {
        float4 v = st.shRec[3 * (size_t)shSlot + 2];
        c.x = v.x; c.y = v.y; c.z = v.z;
        st.col[pix] = c;
    }
    if (__float_as_int(c.w) <= 0) {
        /* the path's one contribution, see k_shade (nothing for a path that ended below zero) */
        const bool counts = __float_as_int(c.w) == 0;
        float* im = st.image + 3 * (size_t)pix;
        im[0] = counts ? c.x : 0.0f; im[1] = counts ? c.y : 0.0f; im[2] = counts ? c.z : 0.0f;
    }
}

/* what the closest-hit traversal hands to the primitive stage for a ray: (distance, barycentric u, v, leaf-order triangle [+ geom]) */
PT_DEV void writeMeshHit(const SceneView& S, const PathState& st, int pix, int bounce, const Hit& best)
{
    const int triWord = S.packGeom ? (best.tri | (best.geom << 26)) : best.tri;
    hitARef(S, st, pix, bounce) = make_float4(best.t, best.uv.x, best.uv.y, __int_as_float(triWord));
    if (!S.packGeom && best.t > 0.0f) st.hitTri[pix] = best.geom;
}

template <int MODE, bool COUNT>
__global__ void __launch_bounds__(kBlock, PT_TRACE_MINWAVES) k_trace(SceneView S, PathState st, Queues q, int bounce,
                                                                     uint32_t* fetchCtr, const uint32_t* workCount, uint32_t* refCount, DevCounters* dc)
{
    __shared__ int s_stack[kStackDepth * kBlock];
    __shared__ float4 s_top[kTopNodes > 0 ? kTopNodes * 8 : 1];
    const int tid = threadIdx.x;
    const int lane = laneId();
    st = st.at(blockIdx.y);
    q = q.at(blockIdx.y);
    const uint32_t n = workCount[blockIdx.y * kCtrStride];
    /* a short queue does not need every workgroup of the row: those beyond one ray per lane leave at once, without
     * touching the counters (late bounces hold a few thousand rays per row) */
    if ((uint32_t)blockIdx.x * kBlock >= n) return;
    if (kTopNodes > 0) {
        /* the top of the first mesh's tree, a copy per workgroup (the branch above is uniform over the workgroup) */
        for (int i = tid; i < S.topCount * 8; i += kBlock) s_top[i] = S.nodes[i];
        __syncthreads();
    }
    /* The work queue is cut into kSegments contiguous pieces with a fetch counter each (on a line of its own,
     * kSegPitch dwords apart).  Workgroups are dealt round-robin over the 8 XCDs, so (linear workgroup id) % 8 names the
     * groups that share an L2: each starts on "its" piece and moves on to the others when that runs dry.  Queues
     * are filed in pixel / slot order, so a piece is a coherent part of the image and an XCD's L2 mostly sees one
     * part of the BVH. */
    uint32_t* ctr = fetchCtr + blockIdx.y * kCtrStride;
    int seg = (int)((blockIdx.y * gridDim.x + blockIdx.x) % kSegments);
    int segsLeft = kSegments;       /* 0 once every piece has been seen dry */
    bool needScan = false;
    int* stack = s_stack + tid;
    int ovRef[kOverflowStack];
    float ovT[kOverflowStack];
    const unsigned long long lt = laneMaskLt();
    const int firstMesh = nextBvhMesh(S, 0);          /* >= 0: the host launches this kernel only then */

    /* lane states: trav = walking mesh gi; done = that walk has ended and its epilogue is due;
     * neither = wants a new item; exhausted = the counter has run past the last item */
    bool trav = false, done = false, exhausted = false;
    bool uncertain = false;     /* a mesh walk of the lane's ray gave an answer that depends on the reference's leaf order */
    int pix = -1, gi = 0;
    Ray r; r.o = f3(0.0f); r.d = f3(0.0f);
    Hit best; best.t = -1.0f; best.n = f3(0.0f); best.mat = 0; best.uv.x = 0; best.uv.y = 0; best.geom = -1; best.tri = -1;
    float shDist = 0.0f;        /* shadow mode: distance to the light sample */
    int shLight = -1;           /* shadow mode: geom of the light, -1 for an environment ray */
    int shSlot = 0;             /* shadow mode: queue position of the lane's ray (its colour record is read when the verdict is in) */
    MeshTrav tv;
    travInit(S.geoms[firstMesh], r, tv);
    TravCounters cnt; cnt.nodes = 0; cnt.tris = 0; cnt.meshHits = 0; cnt.boxes = 0; cnt.maxSp = 0;
    unsigned long long diag[6] = { 0, 0, 0, 0, 0, 0 };
    unsigned long long tStart = 0, tDry = 0;
    if (COUNT && MODE == TRACE_CLOSEST) {
        tStart = wall_clock64();
        if (lane == 0) atomicCAS(&dc->launchT0, 0ull, tStart);
    }

    /* every wave leaves this loop: a lane's walk ends when its stack is empty, a wave stops fetching once all
     * eight pieces have been seen dry, and then breaks out as soon as no lane is walking */
    while (true) {
        const unsigned long long idle = __ballot(!trav && !exhausted);
        if (__ballot(trav) == 0 || __builtin_popcountll(idle) >= kRefillThreshold) {
            /* epilogues and refills run together for all idle lanes */
            bool fetch = !trav && !exhausted;
            if (done) {
                const DGeom& g = S.geoms[gi];
                /* the rare ray whose closest hit depends on the order the reference visits its leaves in: k_refwalk redoes it */
                if (!travCertain(S, tv)) uncertain = true;
                F3 nn; F2 uv; int mat = 0, tri = -1;
                float t = travFinish(S, g, r, tv, false, nn, uv, mat, tri);
                if (COUNT && tv.best >= 0) cnt.meshHits++;
                bool blocked = false;
                if (MODE == TRACE_CLOSEST) {
                    /* only what decides the closest hit is kept here: distance, barycentrics, triangle, geom.  Normal,
                     * uv and material are interpolated by k_prims, one thread per slot with every lane busy, instead of
                     * here with the few lanes of a refill pass */
                    if (closer(t, gi, best)) { best.t = t; best.geom = gi; best.tri = tv.best; best.uv.x = tv.bary.y; best.uv.y = tv.bary.z; }
                } else {
                    blocked = shLight >= 0 ? blocksLight(t, shDist) : blocksEnv(t);
                }
                const int next = blocked ? -1 : nextBvhMesh(S, gi + 1);
                done = false;
                if (next >= 0) {
                    gi = next;
                    travInit(S.geoms[gi], r, tv);
                    trav = true;
                    fetch = false;
                } else {
                    /* the ray is finished.  One whose result hangs on the reference's leaf order goes on k_refwalk's list and is
                     * settled there (its shadow verdict must not be acted on here; its hit record is simply overwritten) */
                    const unsigned long long um = __ballot(uncertain);
                    if (um) {
                        const int leader = __builtin_ctzll(um);
                        uint32_t at = 0;
                        if (lane == leader) at = atomicAdd(refCount + blockIdx.y * kCtrStride, (uint32_t)__builtin_popcountll(um));
                        at = (uint32_t)__shfl((int)at, leader);
                        if (uncertain) st.refList[(MODE == TRACE_CLOSEST ? 0 : st.numPixels) + at + (uint32_t)__builtin_popcountll(um & lt)] = MODE == TRACE_CLOSEST ? pix : shSlot;
                    }
                    if (MODE == TRACE_CLOSEST) writeMeshHit(S, st, pix, bounce, best);
                    else if (!uncertain) settleShadow(st, pix, shSlot, blocked);
                    uncertain = false;
                }
            }
            if (COUNT) { diag[4]++; diag[5] += __builtin_popcountll(__ballot(fetch)); }
            while (true) {
                const unsigned long long fm = __ballot(fetch);
                if (fm == 0) break;
                if (segsLeft == 0) { if (fetch) exhausted = true; break; }
                if (needScan) {
                    needScan = false;
                    /* the piece this wave was on is used up: read all eight counters and go to the next
                     * piece that still has entries, instead of finding every dry piece with an atomic of its own —
                     * at the end of a launch thousands of waves do that at once, and their atomics on one line took
                     * longer than the last rays.  The counters only grow, so a stale value costs at worst the atomic
                     * it was meant to save. */
                    int foundDist = kSegments;
#pragma unroll
                    for (int k = 0; k < kSegments; ++k) {
                        const uint32_t len = (uint32_t)(((unsigned long long)n * (unsigned)(k + 1)) / kSegments) - (uint32_t)(((unsigned long long)n * (unsigned)k) / kSegments);
                        const uint32_t cur = __hip_atomic_load(ctr + k * kSegPitch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const int dist = (k - seg) & (kSegments - 1);
                        if (cur < len && dist < foundDist) foundDist = dist;
                    }
                    if (foundDist == kSegments) { segsLeft = 0; if (COUNT) tDry = wall_clock64(); if (fetch) exhausted = true; break; }
                    seg = (seg + foundDist) & (kSegments - 1);
                }
                const uint32_t segBegin = (uint32_t)(((unsigned long long)n * (unsigned)seg) / kSegments);
                const uint32_t segLen = (uint32_t)(((unsigned long long)n * (unsigned)(seg + 1)) / kSegments) - segBegin;
                const uint32_t want = (uint32_t)__builtin_popcountll(fm);
                const int leader = __builtin_ctzll(fm);
                uint32_t base = 0;
                if (lane == leader) base = atomicAdd(ctr + seg * kSegPitch, want);
                base = (uint32_t)__shfl((int)base, leader);
                if (base + want >= segLen) { seg = (seg + 1) % kSegments; needScan = true; }     /* this piece is used up */
                if (fetch) {
                    const uint32_t sloc = base + (uint32_t)__builtin_popcountll(fm & lt);
                    if (sloc < segLen) {
                        const uint32_t sidx = segBegin + sloc;
                        fetch = false;
                        if (MODE == TRACE_CLOSEST) {
                            pix = st.traceQueue[sidx];
                            float4 o = loadRayO(S, st.rayO, pix, bounce), d = loadRayD(S, st, pix, bounce);
                            r.o = f3(o.x, o.y, o.z); r.d = f3(d.x, d.y, d.z);
                            best.t = -1.0f; best.n = f3(0.0f); best.mat = 0; best.uv.x = 0.0f; best.uv.y = 0.0f; best.geom = -1; best.tri = -1;
                        } else {
                            const float4* rec = st.shRec + 3 * (size_t)sidx;
                            float4 o = rec[0], d = rec[1];
                            pix = __float_as_int(rec[2].w);
                            shSlot = (int)sidx;
                            r.o = f3(o.x, o.y, o.z); r.d = f3(d.x, d.y, d.z);
                            shDist = o.w;
                            shLight = __float_as_int(d.w);
                        }
                        gi = firstMesh;
                        travInit(S.geoms[gi], r, tv);
                        trav = true;
                    }
                }
            }
            if (__ballot(trav) == 0) {
                if (__ballot(!exhausted) == 0) break;
                continue;
            }
        }
        /* leaves wait until kLeafWait lanes hold one (or no lane has a node left): triangle tests then
         * run with fuller waves */
        const unsigned long long nodeLanes = __ballot(trav && tv.ref >= 0);
        const unsigned long long leafLanes = __ballot(trav && tv.ref < 0);
        const bool leafTurn = nodeLanes == 0 || __builtin_popcountll(leafLanes) >= kLeafWait;
        if (COUNT) {
            diag[0]++; diag[1] += __builtin_popcountll(nodeLanes);
            diag[leafTurn ? 2 : 3] += __builtin_popcountll(leafLanes);
        }
        if (trav) {
            bool more = travStep<COUNT>(S, tv, stack, ovRef, ovT, cnt, leafTurn, s_top);
            if (!more) { trav = false; done = true; }
        }
    }
    if (COUNT) {
        flushCounters(cnt, dc, MODE);
        if (lane == 0) {
            for (int i = 0; i < 6; ++i) atomicAdd(&dc->waveDiag[MODE][i], diag[i]);
            atomicMax(&dc->maxWaveIters[MODE], (unsigned)diag[0]);
            if (MODE == TRACE_CLOSEST) {
                const unsigned long long t0 = __hip_atomic_load(&dc->launchT0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long te = wall_clock64();
                unsigned b = (unsigned)((te - t0) / 2500ull); if (b > 63u) b = 63u;
                atomicAdd(&dc->finishHist[b], 1u);
                if (tDry) { unsigned d = (unsigned)((tDry - t0) / 2500ull); if (d > 63u) d = 63u; atomicAdd(&dc->dryHist[d], 1u); }
            }
        }
    }
}

/* The rays k_trace could not vouch for (travCertain), once more: every BVH mesh of the ray is walked the way the reference
 * walks it (travReference), which is the definition of the result.  One thread per list entry; the list is normally empty
 * and at most a handful of entries long (a hit a rounding error outside a triangle's rim with a second hit right behind it). */
template <int MODE>
__global__ void __launch_bounds__(64) k_refwalk(SceneView S, PathState st, int bounce, const uint32_t* refCount, DevCounters* dc)
{
    st = st.at(blockIdx.y);
    const uint32_t n = refCount[blockIdx.y * kCtrStride];
    int stack[64];
    for (uint32_t i = blockIdx.x * 64 + threadIdx.x; i < n; i += gridDim.x * 64) {
        const int id = st.refList[(MODE == TRACE_CLOSEST ? 0 : st.numPixels) + i];
        Ray r;
        int pix = id, shLight = -1;
        float shDist = 0.0f;
        if (MODE == TRACE_CLOSEST) {
            float4 o = loadRayO(S, st.rayO, pix, bounce), d = loadRayD(S, st, pix, bounce);
            r.o = f3(o.x, o.y, o.z); r.d = f3(d.x, d.y, d.z);
        } else {
            const float4* rec = st.shRec + 3 * (size_t)id;
            float4 o = rec[0], d = rec[1];
            pix = __float_as_int(rec[2].w);
            r.o = f3(o.x, o.y, o.z); r.d = f3(d.x, d.y, d.z);
            shDist = o.w;
            shLight = __float_as_int(d.w);
        }
        Hit best; best.t = -1.0f; best.n = f3(0.0f); best.mat = 0; best.uv.x = 0.0f; best.uv.y = 0.0f; best.geom = -1; best.tri = -1;
        bool blocked = false;
        for (int gi = nextBvhMesh(S, 0); gi >= 0 && !blocked; gi = nextBvhMesh(S, gi + 1)) {
            const DGeom& g = S.geoms[gi];
            MeshTrav tv;
            travInit(g, r, tv);
            /* a ray with a non-finite origin or direction hits nothing (travInit) */
            if (__builtin_fabsf(tv.lo.x) < __builtin_inff()) travReference(S, g, tv, stack);
            F3 nn; F2 uv; int mat = 0, tri = -1;
            const float t = travFinish(S, g, r, tv, false, nn, uv, mat, tri);
            if (MODE == TRACE_CLOSEST) {
                if (closer(t, gi, best)) { best.t = t; best.geom = gi; best.tri = tv.best; best.uv.x = tv.bary.y; best.uv.y = tv.bary.z; }
            } else blocked = shLight >= 0 ? blocksLight(t, shDist) : blocksEnv(t);
        }
        if (MODE == TRACE_CLOSEST) writeMeshHit(S, st, pix, bounce, best);
        else settleShadow(st, pix, id, blocked);
    }
    if (dc && n && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&dc->refWalks, (unsigned long long)n);
}

/* fold the inline geoms into the mesh result; final hit record, sort bucket, bucket histogram.
 * SPARSE: the grid is bounded and a workgroup walks the 256-slot pieces vb, vb + gridDim.x, ... of its row — for
 * bounces whose rows are known to be nearly empty, where one workgroup per 256 slots of the full image is mostly
 * launch overhead.  Both forms give every piece the same treatment. */
template <bool COUNT, bool SPARSE>
__global__ void __launch_bounds__(kBlock) k_prims(SceneView S, PathState st, Queues q, int bounce, int sortOn, DevCounters* dc)
{
    __shared__ int s_hist[kMaxBuckets];
    st = st.at(blockIdx.y);
    q = q.at(blockIdx.y);
    const int tid = threadIdx.x;
    const uint32_t n = q.live[bounce];
    int vb = (int)blockIdx.x;
    if ((uint32_t)(vb * kBlock) >= n) return;
    const int numBuckets = S.numBuckets;
    TravCounters cnt; cnt.nodes = 0; cnt.tris = 0; cnt.meshHits = 0; cnt.boxes = 0; cnt.maxSp = 0;
    do {
        const int slot = vb * kBlock + tid;
        if (sortOn) {
            for (int i = tid; i < numBuckets; i += kBlock) s_hist[i] = 0;
            __syncthreads();
        }
        const bool active = (uint32_t)slot < n;
        int key = 0;
        if (active) {
            const int pix = bounce == 0 ? slot : q.q0[slot];
            float4 o = loadRayO(S, st.rayO, pix, bounce), d = loadRayD(S, st, pix, bounce);
            Ray r; r.o = f3(o.x, o.y, o.z); r.d = f3(d.x, d.y, d.z);
            Hit best; best.t = -1.0f; best.n = f3(0.0f); best.mat = 0; best.uv.x = 0.0f; best.uv.y = 0.0f; best.geom = -1; best.tri = -1;
            const bool traced = S.numBvhMeshes > 0 && __float_as_int(d.w) != 0;      /* the ray was filed for traversal */
            if (traced) {
                /* the traversal left (distance, barycentric u, v, leaf-order triangle) and the geom: finish the hit */
                float4 ha = hitARef(S, st, pix, bounce);
                if (ha.x > 0.0f) {
                    const int triWord = __float_as_int(ha.w);
                    const int geom = S.packGeom ? ((triWord >> 26) & 31) : st.hitTri[pix];
                    const int leafTri = S.packGeom ? (triWord & 0x3ffffff) : triWord;
                    best.t = ha.x; best.geom = geom;
                    meshHitAttributes(S, S.geoms[geom], r.d, leafTri, ha.y, ha.z, best.n, best.uv, best.mat, best.tri, !S.leanHits);
                }
            }
            foldInlineGeoms<COUNT>(S, r, best, cnt);
            /* a hit always gets its full record here; a miss is fully described by its distance (-1), which the
             * traversal has already stored for a ray it was handed; with the sort on, a miss needs no record at all:
             * the shade stage knows the misses by their place in the sorted order */
            const bool hit = best.t > 0.0f;
            if (hit || !S.leanHits || (!traced && !sortOn)) {
                hitARef(S, st, pix, bounce) = make_float4(best.t, best.n.x, best.n.y, best.n.z);
                if (hit || !S.leanHits) {
                    st.hitB[pix] = make_float4(__int_as_float(best.mat), best.uv.x, best.uv.y, __int_as_float(best.geom));
                    if (!S.leanHits) st.hitTri[pix] = best.tri;        /* read by the stage hooks only */
                }
            }
            /* sort key: material id, misses last (INT_MAX in the reference, src/pathtrace.cu:2427-2429) */
            key = (best.t > 0.0f) ? best.mat : (numBuckets - 1);
            if (sortOn) q.keys[slot] = key;
        }
        if (sortOn) {
            /* wave-aggregated: one LDS atomic per distinct bucket per wave */
            unsigned long long todo = __ballot(active);
            while (todo) {
                int leader = __builtin_ctzll(todo);
                int k = __shfl(key, leader);
                unsigned long long m = __ballot(active && key == k);
                if (laneId() == leader) atomicAdd(&s_hist[k], (int)__builtin_popcountll(m));
                todo &= ~m;
            }
            __syncthreads();
            for (int i = tid; i < numBuckets; i += kBlock) q.blockHist[(size_t)i * q.numBlocks + vb] = s_hist[i];
        }
        if (!SPARSE) break;
        vb += (int)gridDim.x;
        __syncthreads();
    } while ((uint32_t)(vb * kBlock) < n);
    if (COUNT) flushCounters(cnt, dc, 0);
}

/* block-wide exclusive scan helper for 1024 threads; returns the exclusive prefix of v and the
 * block total through *total */
PT_DEV int blockExclusiveScan1024(int v, int* s_wave, int* total)
{
    const int lane = laneId(), wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        int w = lane < 16 ? s_wave[lane] : 0;
        int winc = w;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            int t = __shfl_up(winc, off);
            if (lane >= off) winc += t;
        }
        if (lane < 16) s_wave[lane] = winc - w;          /* exclusive wave offsets */
        if (lane == 15) s_wave[16] = winc;
    }
    __syncthreads();
    int res = s_wave[wave] + inc - v;
    *total = s_wave[16];
    __syncthreads();
    return res;
}

/* exclusive scan over the histogram in (bucket, workgroup) order, restricted to the
 * workgroups that held live slots; single workgroup of 1024 threads */
__global__ void __launch_bounds__(1024) k_sort_scan(Queues q, int bounce, int numBuckets)
{
    __shared__ int s_wave[17];
    q = q.at(blockIdx.y);
    const uint32_t n = q.live[bounce];
    if (n == 0) return;
    const int nbAct = (int)((n + kBlock - 1) / kBlock);
    const int total = numBuckets * nbAct;
    int carry = 0;
    for (int base = 0; base < total; base += 1024 * 4) {
        int idx[4], v[4];
        int sum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int i = base + threadIdx.x * 4 + k;
            int val = 0, pos = -1;
            if (i < total) {
                int bucket = i / nbAct, blk = i - bucket * nbAct;
                pos = bucket * q.numBlocks + blk;
                val = q.blockHist[pos];
            }
            idx[k] = pos; v[k] = val; sum += val;
        }
        int blockTotal;
        int excl = blockExclusiveScan1024(sum, s_wave, &blockTotal) + carry;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (idx[k] >= 0) q.blockHist[idx[k]] = excl;
            excl += v[k];
        }
        carry += blockTotal;
    }
}

/* stable scatter: rank inside the wave by ballot, wave offsets through LDS, workgroup
 * offsets from the scanned histogram */
__global__ void __launch_bounds__(kBlock) k_sort_scatter(Queues q, int bounce, int numBuckets)
{
    __shared__ int s_waveCnt[(kBlock / 64) * kMaxBuckets];
    q = q.at(blockIdx.y);
    const int tid = threadIdx.x;
    const uint32_t n = q.live[bounce];
    const int wave = tid >> 6, lane = laneId();
    PT_FOR_VIRTUAL_BLOCKS(vb, n) {
    const int slot = vb * kBlock + tid;
    for (int i = tid; i < (kBlock / 64) * numBuckets; i += kBlock) s_waveCnt[i] = 0;
    __syncthreads();
    const bool active = (uint32_t)slot < n;
    int key = active ? q.keys[slot] : 0;
    int pix = active ? (bounce == 0 ? slot : q.q0[slot]) : 0;
    int rank = 0;
    unsigned long long todo = __ballot(active);
    const unsigned long long lt = laneMaskLt();
    while (todo) {
        int leader = __builtin_ctzll(todo);
        int k = __shfl(key, leader);
        unsigned long long m = __ballot(active && key == k);
        if (active && key == k) rank = (int)__builtin_popcountll(m & lt);
        if (lane == leader) s_waveCnt[wave * numBuckets + k] = (int)__builtin_popcountll(m);
        todo &= ~m;
    }
    __syncthreads();
    if (active) {
        int base = q.blockHist[(size_t)key * q.numBlocks + vb];
        for (int w = 0; w < wave; ++w) base += s_waveCnt[w * numBuckets + key];
        q.q1[base + rank] = pix;
    }
    __syncthreads();
    }
}

/* ------------------------------------------------------------------- shade
 * One thread per sorted slot.  Shadow rays are tested against the inline geoms here; when BVH
 * meshes exist and nothing inline blocks, the ray goes to the shadow queue with the path colour of
 * the unblocked outcome, and the path itself continues with the blocked outcome until k_trace in
 * shadow mode settles it (before the next stage reads the colour). */
template <bool COUNT, bool SPARSE>
__global__ void __launch_bounds__(kBlock, PT_SHADE_MINWAVES) k_shade(SceneView S, PathState st, Queues q, Batch bt, int bounce, int sortOn,
                                                                     uint32_t* shadowCount, uint32_t* nextTraceCount, DevCounters* dc)
{
    __shared__ int s_alive[kBlock / 64];
    __shared__ int s_tmp[kBlock / 64 + 1];
    st = st.at(blockIdx.y);
    q = q.at(blockIdx.y);
    const int iter = bt.iterBase + (int)blockIdx.y * bt.iterStride;
    const int tid = threadIdx.x;
    const uint32_t n = q.live[bounce];
    int vb = (int)blockIdx.x;                   /* SPARSE: bounded grid, see k_prims */
    if ((uint32_t)(vb * kBlock) >= n) return;
    int* queue = sortOn ? q.q1 : q.q0;
    /* sorted order: the misses are the last bucket, and its first slot is the scanned histogram entry of that bucket's
     * first workgroup — their hit records (distance -1, nothing else read) are not fetched */
    const uint32_t missStart = sortOn ? (uint32_t)q.blockHist[(size_t)(S.numBuckets - 1) * q.numBlocks] : 0xffffffffu;
    TravCounters cnt; cnt.nodes = 0; cnt.tris = 0; cnt.meshHits = 0; cnt.boxes = 0; cnt.maxSp = 0;
    do {
        const int slot = vb * kBlock + tid;
        const bool active = (uint32_t)slot < n;
        bool alive = false;
        ShadowReq sq;
        sq.pending = false; sq.o = f3(0.0f); sq.d = f3(0.0f); sq.dist = 0.0f; sq.lightGeom = -1; sq.colorVisible = f3(0.0f);
        int pix = 0;
        bool cand = false;
        if (active) {
            pix = (!sortOn && bounce == 0) ? slot : queue[slot];
            float4 o = loadRayO(S, st.rayO, pix, bounce), d = loadRayD(S, st, pix, bounce), c = loadCol(S, st.col, pix, bounce);
            float4 ha = make_float4(-1.0f, 0.0f, 0.0f, 0.0f);
            if ((uint32_t)slot < missStart) ha = hitARef(S, st, pix, bounce);
            float4 hb = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (ha.x > 0.0f) hb = st.hitB[pix];        /* material and uv only matter for a hit */
            Path p;
            p.o = f3(o.x, o.y, o.z); p.d = f3(d.x, d.y, d.z); p.color = f3(c.x, c.y, c.z);
            p.remaining = __float_as_int(c.w);
            p.prevSpecular = __float_as_int(o.w) != 0;
            F2 uv; uv.x = hb.y; uv.y = hb.z;
            shadeSlot<COUNT>(S, iter, bounce + 1, slot, p, ha.x, f3(ha.y, ha.z, ha.w), __float_as_int(hb.x), uv, bounce == 0, sq, cnt);
            alive = p.remaining > 0;
            if (alive) {
                /* the surviving path's new ray goes on the work list of the next closest-hit traversal if it can hit a mesh */
                if (nextTraceCount) { Ray nr; nr.o = p.o; nr.d = p.d; cand = mayHitBvhMeshes(S, nr, __builtin_inff(), true); }
                st.rayO[pix] = make_float4(p.o.x, p.o.y, p.o.z, __int_as_float(p.prevSpecular ? 1 : 0));
                st.rayD[pix] = make_float4(p.d.x, p.d.y, p.d.z, __int_as_float(cand ? 1 : 0));
            }
            if (alive || sq.pending) {
                st.col[pix] = make_float4(p.color.x, p.color.y, p.color.z, __int_as_float(p.remaining));
            } else {
                /* gatherTerminatedPaths (src/pathtrace.cu:2641-2653).  A pixel has one path per iteration and every path
                 * ends exactly once — here, in the shadow traversal that settles its last colour, or in the final
                 * gather — so its contribution is a plain store into this iteration's contribution image, and that
                 * image is rewritten completely by every iteration (nothing to clear, no read-modify-write). */
                float* im = st.image + 3 * (size_t)pix;
                const bool counts = p.remaining == 0;       /* a path that ended below zero adds nothing (the reference gathers == 0 only) */
                im[0] = counts ? p.color.x : 0.0f; im[1] = counts ? p.color.y : 0.0f; im[2] = counts ? p.color.z : 0.0f;
            }
            queue[slot] = alive ? pix : (-1 - pix);
        }
        if (nextTraceCount) {
            const int at = blockAppendIndex(cand, nextTraceCount + blockIdx.y * kCtrStride, s_tmp);
            if (at >= 0) st.traceQueue[at] = pix;
        }
        /* shadow queue append */
        const int shIdx = blockAppendIndex(sq.pending, shadowCount + blockIdx.y * kCtrStride, s_tmp);
        if (shIdx >= 0) {
            float4* rec = st.shRec + 3 * (size_t)shIdx;
            rec[0] = make_float4(sq.o.x, sq.o.y, sq.o.z, sq.dist);
            rec[1] = make_float4(sq.d.x, sq.d.y, sq.d.z, __int_as_float(sq.lightGeom));
            rec[2] = make_float4(sq.colorVisible.x, sq.colorVisible.y, sq.colorVisible.z, __int_as_float(pix));
        }
        unsigned long long m = __ballot(alive);
        if (laneId() == 0) s_alive[tid >> 6] = (int)__builtin_popcountll(m);
        __syncthreads();
        if (tid == 0) {
            int s = 0;
#pragma unroll
            for (int w = 0; w < kBlock / 64; ++w) s += s_alive[w];
            q.blockAlive[vb] = s;
        }
        if (!SPARSE) break;
        vb += (int)gridDim.x;
        __syncthreads();
    } while ((uint32_t)(vb * kBlock) < n);
    if (COUNT) flushCounters(cnt, dc, 1);
}

/* exclusive scan of survivors per workgroup; publishes live[bounce + 1] */
__global__ void __launch_bounds__(1024) k_compact_scan(Queues q, int bounce)
{
    __shared__ int s_wave[17];
    q = q.at(blockIdx.y);
    const uint32_t n = q.live[bounce];
    const int nbAct = (int)((n + kBlock - 1) / kBlock);
    int carry = 0;
    for (int base = 0; base < nbAct; base += 1024) {
        int i = base + threadIdx.x;
        int v = i < nbAct ? q.blockAlive[i] : 0;
        int blockTotal;
        int excl = blockExclusiveScan1024(v, s_wave, &blockTotal) + carry;
        if (i < nbAct) q.blockAlive[i] = excl;
        carry += blockTotal;
    }
    if (threadIdx.x == 0) q.live[bounce + 1] = (uint32_t)carry;
}

/* order-preserving compaction of the flagged queue into q0 (the role thrust::remove_if plays at
 * src/pathtrace.cu:2865-2868; a 64-lane ballot scan instead of a multi-launch Blelloch scan) */
__global__ void __launch_bounds__(kBlock) k_compact_scatter(Queues q, int bounce, int sortOn)
{
    __shared__ int s_cnt[kBlock / 64];
    q = q.at(blockIdx.y);
    const int tid = threadIdx.x;
    const uint32_t n = q.live[bounce];
    const int* src = sortOn ? q.q1 : q.q0;
    int* dst = sortOn ? q.q0 : q.q1;
    PT_FOR_VIRTUAL_BLOCKS(vb, n) {
    const int slot = vb * kBlock + tid;
    int e = ((uint32_t)slot < n) ? src[slot] : -1;
    bool alive = e >= 0;
    unsigned long long m = __ballot(alive);
    int rank = (int)__builtin_popcountll(m & laneMaskLt());
    if (laneId() == 0) s_cnt[tid >> 6] = (int)__builtin_popcountll(m);
    __syncthreads();
    int base = q.blockAlive[vb];
    for (int w = 0; w < (tid >> 6); ++w) base += s_cnt[w];
    if (alive) dst[base + rank] = e;
    __syncthreads();
    }
}

/* unsorted mode compacts q0 -> q1; copy back so q0 is always the input of the next bounce */
__global__ void __launch_bounds__(kBlock) k_copy_queue(Queues q, int bounce)
{
    q = q.at(blockIdx.y);
    const uint32_t n = q.live[bounce + 1];
    PT_FOR_VIRTUAL_BLOCKS(vb, n) {
        const int slot = vb * kBlock + threadIdx.x;
        if ((uint32_t)slot < n) q.q0[slot] = q.q1[slot];
    }
}

/* finalGather (src/pathtrace.cu:2630-2639): survivors of the last bounce, normally none */
__global__ void __launch_bounds__(kBlock) k_final_gather(PathState st, Queues q, int depth)
{
    st = st.at(blockIdx.y);
    q = q.at(blockIdx.y);
    const uint32_t n = q.live[depth];
    PT_FOR_VIRTUAL_BLOCKS(vb, n) {
        const int slot = vb * kBlock + threadIdx.x;
        if ((uint32_t)slot >= n) continue;
        const int pix = q.q0[slot];
        float4 c = st.col[pix];
        float* im = st.image + 3 * (size_t)pix;
        im[0] = c.x; im[1] = c.y; im[2] = c.z;
    }
}

/* per-batch bookkeeping: path-bounce total over the batch, live curve of its last iteration */
__global__ void __launch_bounds__(64) k_finish(Queues q0, int depth, int batch, DevCounters* dc, const uint32_t* shadowCounts, const uint32_t* closestCounts, int rowStride)
{
    if (blockIdx.x != 0) return;
    /* one wave: lane t takes rows t, t + 64, ...; sums and maxima are folded across the lanes */
    const int lane = threadIdx.x;
    unsigned long long s = 0, wc = 0, ws = 0;
    uint32_t mx[PT_MAX_DEPTH + 2];
    for (int b = 0; b <= depth; ++b) mx[b] = 0;
    for (int k = lane; k < batch; k += 64) {
        for (int b = 0; b < depth; ++b) {
            wc += closestCounts[((size_t)b * rowStride + k) * kCtrStride];
            ws += shadowCounts[((size_t)b * rowStride + k) * kCtrStride];
        }
        Queues q = q0.at(k);
        bool open = true;
        for (int b = 0; b <= depth; ++b) {
            const uint32_t v = q.live[b];
            if (open && b < depth) { s += v; if (v == 0) open = false; }
            if (v > mx[b]) mx[b] = v;
            if (k == batch - 1) dc->lastLive[b] = v;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off); wc += __shfl_down(wc, off); ws += __shfl_down(ws, off);
    }
    for (int b = 0; b <= depth; ++b) {
        uint32_t m = mx[b];
        for (int off = 32; off > 0; off >>= 1) { uint32_t o = (uint32_t)__shfl_down((int)m, off); if (o > m) m = o; }
        if (lane == 0) dc->liveMax[b] = m;
    }
    if (lane == 0) {
        dc->workItems[0] += wc;
        dc->workItems[1] += ws;
        dc->pathBounces += s;
    }
}

/* image += contribution of iteration 0, then 1, ... : the order one-iteration-at-a-time accumulation
 * would add them in (exactly one contribution per pixel and iteration, stored by the stage that ends the path) */
__global__ void __launch_bounds__(kBlock) k_accumulate(float* image, const float* partial, size_t n3, int batch)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n3) return;
    float acc = image[i];
    for (int k = 0; k < batch; ++k) {
        acc += partial[(size_t)k * n3 + i];
    }
    image[i] = acc;
}

/* the same, four floats per lane (16-byte accesses); n4 = n3 / 4, for 16-byte aligned buffers with n3 % 4 == 0 */
__global__ void __launch_bounds__(kBlock) k_accumulate4(float4* image, const float4* partial, size_t n4, int batch)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n4) return;
    float4 acc = image[i];
    for (int k = 0; k < batch; ++k) {
        float4 v = partial[(size_t)k * n4 + i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    image[i] = acc;
}

/* dst += src over n floats (the fold of another context's running sum, pt_reduce_sum) */
__global__ void __launch_bounds__(kBlock) k_add_image(float* dst, const float* src, size_t n)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

/* sendImageToPBO (src/pathtrace.cu:145-166): (int)(sum / iter * 255.0) with a double 255.0 */
__global__ void __launch_bounds__(kBlock) k_display(const float* image, int n, int iter, uint8_t* rgba)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    int c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int v = (int)(image[3 * (size_t)i + k] / iter * 255.0);
        c[k] = v < 0 ? 0 : (v > 255 ? 255 : v);
    }
    uchar4 px; px.x = (unsigned char)c[0]; px.y = (unsigned char)c[1]; px.z = (unsigned char)c[2]; px.w = 0;
    reinterpret_cast<uchar4*>(rgba)[i] = px;
}

__global__ void __launch_bounds__(kBlock) k_export_rays(PathState st, int n, float* o, float* d)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float4 a = st.rayO[i], b = st.rayD[i];
    o[3 * (size_t)i] = a.x; o[3 * (size_t)i + 1] = a.y; o[3 * (size_t)i + 2] = a.z;
    d[3 * (size_t)i] = b.x; d[3 * (size_t)i + 1] = b.y; d[3 * (size_t)i + 2] = b.z;
}

/* stage-level hook: caller rays into row 0 of the path state, final hit records out */
__global__ void __launch_bounds__(kBlock) k_import_rays(PathState st, Queues q, int n, const float* o, const float* d)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i == 0) q.live[0] = (uint32_t)n;
    if (i >= n) return;
    st.rayO[i] = make_float4(o[3 * (size_t)i], o[3 * (size_t)i + 1], o[3 * (size_t)i + 2], 0.0f);
    st.rayD[i] = make_float4(d[3 * (size_t)i], d[3 * (size_t)i + 1], d[3 * (size_t)i + 2], 0.0f);
}
__global__ void __launch_bounds__(kBlock) k_export_hits(PathState st, int n, PtHit* out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float4 ha = st.hitA[i], hb = st.hitB[i];
    PtHit ph;
    ph.t = ha.x; ph.geom = __float_as_int(hb.w); ph.triangle = st.hitTri[i];
    ph.materialId = ha.x > 0.0f ? __float_as_int(hb.x) : -1;
    ph.normal[0] = ha.y; ph.normal[1] = ha.z; ph.normal[2] = ha.w;
    ph.uv[0] = hb.y; ph.uv[1] = hb.z;
    out[i] = ph;
}

__global__ void __launch_bounds__(kBlock) k_math(int fn, const float* a, const float* b, int n, float* out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b ? b[i] : 0.0f, r = 0.0f;
    switch (fn) {
    case 0: r = dm::sin(x); break;
    case 1: r = dm::cos(x); break;
    case 2: r = dm::acos(x); break;
    case 3: r = dm::atan2(x, y); break;
    case 4: r = dm::exp(x); break;
    case 5: r = dm::log(x); break;
    case 6: r = dm::pow5(x); break;
    case 7: r = x / y; break;
    case 8: r = __builtin_sqrtf(x); break;
    default: break;
    }
    out[i] = r;
}

} // namespace ptd

/* ---------------------------------------------------------------- launchers
 * Plain functions so that the API translation unit stays free of kernel syntax. */
using namespace ptd;

static inline int gridFor(long long n) { return (int)((n + kBlock - 1) / kBlock); }

extern "C++" {

void ptl_camera(hipStream_t s, const SceneView& S, const PtCamera& cam, Batch bt, int depth, PathState st, Queues q, uint32_t* traceCount)
{
    static_assert(kBlock == 256, "a camera workgroup is a strip of four 8x8 tiles");
    hipLaunchKernelGGL(k_camera, dim3((cam.resolution[0] + 31) / 32, (cam.resolution[1] + 7) / 8, bt.count), dim3(kBlock), 0, s, S, cam, bt, depth, st, q, traceCount);
}
/* grid of the light one-thread-per-slot kernels: all rows together about 16k workgroups (see PT_FOR_VIRTUAL_BLOCKS) */
static inline dim3 slotGrid(const Queues& q, const Batch& bt)
{
    int x = (16384 + bt.count - 1) / bt.count;
    if (x > q.numBlocks) x = q.numBlocks;
    if (x < 1) x = 1;
    return dim3(x, bt.count);
}
static inline int persistentGridX(int batch)
{
    static const int total = [] { const char* e = getenv("PT_PERSIST_BLOCKS"); int v = e ? atoi(e) : 0; return v > 0 ? v : 1024; }();
    /* 256 CUs x 4 workgroups of 4 waves = 16 waves per CU in flight; split over the batch rows.  Rounded DOWN: a
     * workgroup only serves its own row, so workgroups beyond what the machine holds at once would start when the
     * others are done and walk the last rows alone (1071 workgroups for 63 rows cost 30 % on a large scene). */
    int x = total / batch;
    return x < 1 ? 1 : x;
}
/* intersect stage = cull (work list of the traversal) + trace (BVH meshes) + prims (everything else, hit record, sort bucket) */
void ptl_cull(hipStream_t s, const SceneView& S, PathState st, Queues q, Batch bt, int bounce, uint32_t* workCount)
{
    hipLaunchKernelGGL(k_cull, dim3(q.numBlocks, bt.count), dim3(kBlock), 0, s, S, st, q, bounce, workCount);
}
/* refCount: length of the list of rays k_refwalk settles behind this launch (zero when launched) */
void ptl_trace(hipStream_t s, const SceneView& S, PathState st, Queues q, Batch bt, int bounce, bool shadow, bool count,
               uint32_t* fetchCtr, const uint32_t* workCount, uint32_t* refCount, void* dc, int share)
{
    /* share: launches expected to run side by side with this one (they split the machine's workgroup slots) */
    dim3 grid(persistentGridX(bt.count * (share > 1 ? share : 1)), bt.count);
    DevCounters* d = (DevCounters*)dc;
    /* the follow-up kernel finds its list empty almost always: a few small workgroups per row */
    const dim3 rgrid(bt.count >= 64 ? 1 : bt.count >= 8 ? 4 : 16, bt.count);
    if (shadow) {
        if (count) hipLaunchKernelGGL((k_trace<TRACE_SHADOW, true>), grid, dim3(kBlock), 0, s, S, st, q, bounce, fetchCtr, workCount, refCount, d);
        else hipLaunchKernelGGL((k_trace<TRACE_SHADOW, false>), grid, dim3(kBlock), 0, s, S, st, q, bounce, fetchCtr, workCount, refCount, d);
        hipLaunchKernelGGL((k_refwalk<TRACE_SHADOW>), rgrid, dim3(64), 0, s, S, st, bounce, refCount, count ? d : nullptr);
    } else {
        if (count) hipLaunchKernelGGL((k_trace<TRACE_CLOSEST, true>), grid, dim3(kBlock), 0, s, S, st, q, bounce, fetchCtr, workCount, refCount, d);
        else hipLaunchKernelGGL((k_trace<TRACE_CLOSEST, false>), grid, dim3(kBlock), 0, s, S, st, q, bounce, fetchCtr, workCount, refCount, d);
        hipLaunchKernelGGL((k_refwalk<TRACE_CLOSEST>), rgrid, dim3(64), 0, s, S, st, bounce, refCount, count ? d : nullptr);
    }
}
/* grid of the SPARSE form of the heavy per-slot kernels: about 4096 workgroups over all rows */
static inline dim3 sparseGrid(const Queues& q, const Batch& bt)
{
    int x = (4096 + bt.count - 1) / bt.count;
    if (x > q.numBlocks) x = q.numBlocks;
    if (x < 1) x = 1;
    return dim3(x, bt.count);
}
void ptl_prims(hipStream_t s, const SceneView& S, PathState st, Queues q, Batch bt, int bounce, int sortOn, bool count, void* dc, bool sparse)
{
    const dim3 full(q.numBlocks, bt.count);
    if (count) hipLaunchKernelGGL((k_prims<true, false>), full, dim3(kBlock), 0, s, S, st, q, bounce, sortOn, (DevCounters*)dc);
    else if (sparse) hipLaunchKernelGGL((k_prims<false, true>), sparseGrid(q, bt), dim3(kBlock), 0, s, S, st, q, bounce, sortOn, (DevCounters*)dc);
    else hipLaunchKernelGGL((k_prims<false, false>), full, dim3(kBlock), 0, s, S, st, q, bounce, sortOn, (DevCounters*)dc);
}
void ptl_sort(hipStream_t s, Queues q, Batch bt, int bounce, int numBuckets)
{
    hipLaunchKernelGGL(k_sort_scan, dim3(1, bt.count), dim3(1024), 0, s, q, bounce, numBuckets);
    hipLaunchKernelGGL(k_sort_scatter, slotGrid(q, bt), dim3(kBlock), 0, s, q, bounce, numBuckets);
}
void ptl_shade(hipStream_t s, const SceneView& S, PathState st, Queues q, Batch bt, int bounce, int sortOn, bool count,
               uint32_t* shadowCount, uint32_t* nextTraceCount, void* dc, bool sparse)
{
    const dim3 full(q.numBlocks, bt.count);
    if (count) hipLaunchKernelGGL((k_shade<true, false>), full, dim3(kBlock), 0, s, S, st, q, bt, bounce, sortOn, shadowCount, nextTraceCount, (DevCounters*)dc);
    else if (sparse) hipLaunchKernelGGL((k_shade<false, true>), sparseGrid(q, bt), dim3(kBlock), 0, s, S, st, q, bt, bounce, sortOn, shadowCount, nextTraceCount, (DevCounters*)dc);
    else hipLaunchKernelGGL((k_shade<false, false>), full, dim3(kBlock), 0, s, S, st, q, bt, bounce, sortOn, shadowCount, nextTraceCount, (DevCounters*)dc);
}
void ptl_compact(hipStream_t s, Queues q, Batch bt, int bounce, int sortOn)
{
    hipLaunchKernelGGL(k_compact_scan, dim3(1, bt.count), dim3(1024), 0, s, q, bounce);
    hipLaunchKernelGGL(k_compact_scatter, slotGrid(q, bt), dim3(kBlock), 0, s, q, bounce, sortOn);
    if (!sortOn) hipLaunchKernelGGL(k_copy_queue, slotGrid(q, bt), dim3(kBlock), 0, s, q, bounce);
}
/* image += the contribution images of `rows` consecutive rows starting at st's, in row order */
void ptl_accumulate(hipStream_t s, PathState st, int rows, float* image)
{
    const size_t n3 = 3 * st.numPixels;
    if (n3 % 4 == 0 && (((uintptr_t)image | (uintptr_t)st.image) & 15u) == 0)
        hipLaunchKernelGGL(k_accumulate4, dim3(gridFor((long long)(n3 / 4))), dim3(kBlock), 0, s, (float4*)image, (float4*)st.image, n3 / 4, rows);
    else
        hipLaunchKernelGGL(k_accumulate, dim3(gridFor((long long)n3)), dim3(kBlock), 0, s, image, st.image, n3, rows);
}
/* image == nullptr: the rows' contribution images are left for a later ptl_accumulate */
void ptl_finish(hipStream_t s, PathState st, Queues q, Batch bt, int depth, float* image, void* dc,
                const uint32_t* shadowCounts, const uint32_t* closestCounts, int rowStride)
{
    hipLaunchKernelGGL(k_final_gather, slotGrid(q, bt), dim3(kBlock), 0, s, st, q, depth);
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(64), 0, s, q, depth, bt.count, (DevCounters*)dc, shadowCounts, closestCounts, rowStride);
    if (!image) return;
    const size_t n3 = 3 * st.numPixels;
    if (n3 % 4 == 0 && (((uintptr_t)image | (uintptr_t)st.image) & 15u) == 0)
        hipLaunchKernelGGL(k_accumulate4, dim3(gridFor((long long)(n3 / 4))), dim3(kBlock), 0, s, (float4*)image, (float4*)st.image, n3 / 4, bt.count);
    else
        hipLaunchKernelGGL(k_accumulate, dim3(gridFor((long long)n3)), dim3(kBlock), 0, s, image, st.image, n3, bt.count);
}
void ptl_add_image(hipStream_t s, float* dst, const float* src, size_t n)
{
    hipLaunchKernelGGL(k_add_image, dim3(gridFor((long long)n)), dim3(kBlock), 0, s, dst, src, n);
}
void ptl_display(hipStream_t s, const float* image, int n, int iter, uint8_t* rgba)
{
    hipLaunchKernelGGL(k_display, dim3(gridFor(n)), dim3(kBlock), 0, s, image, n, iter, rgba);
}
/* stage-level hook: n <= st.numPixels caller rays into row 0 of the path state / hit records out */
void ptl_import_rays(hipStream_t s, PathState st, Queues q, const float* o, const float* d, int n)
{
    hipLaunchKernelGGL(k_import_rays, dim3(gridFor(n)), dim3(kBlock), 0, s, st, q, n, o, d);
}
void ptl_export_hits(hipStream_t s, PathState st, int n, PtHit* out)
{
    hipLaunchKernelGGL(k_export_hits, dim3(gridFor(n)), dim3(kBlock), 0, s, st, n, out);
}
void ptl_export_rays(hipStream_t s, PathState st, int n, float* o, float* d)
{
    hipLaunchKernelGGL(k_export_rays, dim3(gridFor(n)), dim3(kBlock), 0, s, st, n, o, d);
}
void ptl_math(hipStream_t s, int fn, const float* a, const float* b, int n, float* out)
{
    hipLaunchKernelGGL(k_math, dim3(gridFor(n)), dim3(kBlock), 0, s, fn, a, b, n, out);
}

} // extern "C++"
