/*
 * pt_api.hip — the C ABI of include/mi_pathtrace.h: context lifetime
 * (pathtraceInit / pathtraceFree, src/pathtrace.cu:428-684), the per-iteration
 * host loop (pathtrace, src/pathtrace.cu:2659-2930) and the parity hooks.
 *
 * The host loop enqueues a fixed sequence of launches per bounce on one HIP
 * stream and never reads a count back: live-path counts stay in device memory.
 */
#include "pt_scene_dev.h"

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <utility>
#include <vector>

#include <atomic>

using namespace ptd;

void pt_set_error(const char* fmt, ...);

/* Every call of this file that creates or releases a device-side resource — memory, pinned host memory, events, streams —
 * is counted (pt_resource_calls), so that a caller, or a test, can see that an entry point which promises to allocate
 * nothing made none.  The macros name the runtime functions they wrap; a macro is not expanded inside its own
 * expansion, so the inner name is the runtime's. */
static std::atomic<unsigned long long> g_resourceCalls{0};
#define hipMalloc(...) (++g_resourceCalls, hipMalloc(__VA_ARGS__))
#define hipFree(...) (++g_resourceCalls, hipFree(__VA_ARGS__))
#define hipHostMalloc(...) (++g_resourceCalls, hipHostMalloc(__VA_ARGS__))
#define hipHostFree(...) (++g_resourceCalls, hipHostFree(__VA_ARGS__))
#define hipHostRegister(...) (++g_resourceCalls, hipHostRegister(__VA_ARGS__))
#define hipHostUnregister(...) (++g_resourceCalls, hipHostUnregister(__VA_ARGS__))
#define hipEventCreate(...) (++g_resourceCalls, hipEventCreate(__VA_ARGS__))
#define hipEventCreateWithFlags(...) (++g_resourceCalls, hipEventCreateWithFlags(__VA_ARGS__))
#define hipEventDestroy(...) (++g_resourceCalls, hipEventDestroy(__VA_ARGS__))
#define hipStreamCreateWithFlags(...) (++g_resourceCalls, hipStreamCreateWithFlags(__VA_ARGS__))
#define hipStreamCreateWithPriority(...) (++g_resourceCalls, hipStreamCreateWithPriority(__VA_ARGS__))
#define hipStreamDestroy(...) (++g_resourceCalls, hipStreamDestroy(__VA_ARGS__))

void ptl_camera(hipStream_t s, const SceneView& S, const PtCamera& cam, Batch bt, int depth, PathState st, Queues q, uint32_t* traceCount);
void ptl_cull(hipStream_t s, const SceneView& S, PathState st, Queues q, Batch bt, int bounce, uint32_t* workCount);
void ptl_trace(hipStream_t s, const SceneView& S, PathState st, Queues q, Batch bt, int bounce, bool shadow, bool count,
               uint32_t* fetchCtr, const uint32_t* workCount, uint32_t* refCount, void* dc, int share = 1);
void ptl_prims(hipStream_t s, const SceneView& S, PathState st, Queues q, Batch bt, int bounce, int sortOn, bool count, void* dc, bool sparse);
void ptl_sort(hipStream_t s, Queues q, Batch bt, int bounce, int numBuckets);
void ptl_shade(hipStream_t s, const SceneView& S, PathState st, Queues q, Batch bt, int bounce, int sortOn, bool count,
               uint32_t* shadowCount, uint32_t* nextTraceCount, void* dc, bool sparse);
void ptl_compact(hipStream_t s, Queues q, Batch bt, int bounce, int sortOn);
void ptl_finish(hipStream_t s, PathState st, Queues q, Batch bt, int depth, float* image, void* dc,
                const uint32_t* shadowCounts, const uint32_t* closestCounts, int rowStride);
void ptl_accumulate(hipStream_t s, PathState st, int rows, float* image);
void ptl_add_image(hipStream_t s, float* dst, const float* src, size_t n);
void ptl_display(hipStream_t s, const float* image, int n, int iter, uint8_t* rgba);
void ptl_gbuffer(hipStream_t s, PathState st, const PtMaterial* materials, int n, float* normals, float* albedo);
void ptl_average(hipStream_t s, const float* sum, float samples, size_t n3, float* out);
void ptl_atrous(hipStream_t s, const float* in, const float* normals, const float* albedo, int W, int H, int step,
                float invSigmaC2, float invSigmaN2, float invSigmaA2, float* out);
void ptl_import_rays(hipStream_t s, PathState st, Queues q, const float* o, const float* d, int n);
void ptl_export_hits(hipStream_t s, PathState st, int n, PtHit* out);
void ptl_export_rays(hipStream_t s, PathState st, int n, float* o, float* d);
void ptl_math(hipStream_t s, int fn, const float* a, const float* b, int n, float* out);

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            pt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return PT_EHIP;                                                                        \
        }                                                                                          \
    } while (0)

constexpr int kDefaultStreamGroups = 3;
constexpr int kDefaultBatch = 16;   /* iterations per launch group in pt_trace_range */
constexpr int kMaxBatch = 512;
constexpr int kMaxStreamGroups = 8;
constexpr int kTuneShapes = 4;          /* launch shapes tried: 1, 2, 3 streams (1 with the shadow traversal on a side stream), then one stream and nothing beside it */
constexpr int kTuneMinRows = 12;        /* smaller launch groups stay on one stream */
/* device work counters, zeroed before every launch group: [section][bounce][batch row] with sections
 * 0 = closest-hit rays left for k_refwalk, 1 = shadow rays left for k_refwalk, 2 = shadow queue length,
 * 3 = closest-hit work queue length */
constexpr int kWorkCtrSection = (PT_MAX_DEPTH + 1) * kMaxBatch * kCtrStride;      /* one 128-byte line per counter */
constexpr int kWorkCtrCount = 4 * kWorkCtrSection;
/* the fetch counters of the traversal are [kind: closest, shadow][bounce][segment][row] in PtContext::fetchCtr (see kSegPitch) */
static_assert(kMaxBatch == kCtrRows, "counter layout");
constexpr size_t kFetchKind = (size_t)(PT_MAX_DEPTH + 1) * kFetchBouncePitch;
constexpr size_t kFetchCtrCount = 2 * kFetchKind;

enum Stage { ST_CAMERA = 0, ST_INTERSECT, ST_SORT, ST_SHADE, ST_COMPACT, ST_TRACE, ST_SHADOW_TRACE, ST_COUNT };   /* the last two nest inside intersect / shade */

struct TimedSpan { hipEvent_t a, b; int stage; };

/* Staging memory of pt_intersect_rays_noalloc (ray origins, directions, hit records): part of the code object, so
 * every GPU the library runs on holds its own copy from load until the process ends.  Host code reaches it only
 * through the per-device address hipGetSymbolAddress returns (PtContext::staging, resolved in pt_create). */
static_assert(sizeof(PtHit) == 36, "PtHit layout");
constexpr size_t kStagingBytes = (size_t)PT_NOALLOC_MAX_RAYS * (12 + 12 + sizeof(PtHit));
static __device__ __align__(256) unsigned char scratch[kStagingBytes];

struct PtContext {
    int device = 0;
    hipStream_t stream = nullptr;
    SceneView view{};
    PathState state{};
    Queues queues{};
    DevCounters* dc = nullptr;
    uint32_t* workCtr = nullptr;
    uint32_t* fetchCtr = nullptr;
    void* staging = nullptr;            /* this device's copy of `scratch` */
    int bvhMeshGeoms = 0;               /* mesh geoms that own a BVH */
    int streamGroups = 1;               /* streams the rows of a launch group are spread over */
    hipStream_t groupStreams[kMaxStreamGroups] = {};
    hipEvent_t forkEvent = nullptr, joinEvents[kMaxStreamGroups] = {};
    /* shadow traversal of bounce b next to the closest-hit stage of bounce b + 1 */
    hipStream_t shadowStreams[kMaxStreamGroups] = {};
    hipEvent_t shadedEvents[kMaxStreamGroups] = {}, shadowDoneEvents[kMaxStreamGroups] = {};
    int shadowOverlap = 1;
    /* One-iteration calls (pt_trace) rotate over `pipeDepth` rows of the path state, each with streams of its own:
     * the traversal kernels of one iteration spend most of their time draining a few long rays, and the next calls'
     * kernels fill the machine meanwhile.  Every call's accumulation into the image runs on the main stream in call
     * order, so the image is the same as with one call at a time. */
    int pipeDepth = 4;
    unsigned pipeNext = 0;
    int pipeLast = -1;
    bool pipeBarrier = false;           /* work on the main stream has used the path state since the last pipelined call */
    bool slotBusy[kMaxStreamGroups] = {};
    hipEvent_t slotFreeEvents[kMaxStreamGroups] = {};
    /* Live-path maxima per bounce of the last finished launch group, copied to pinned memory behind every group.
     * They only choose the launch form of the heavy per-slot kernels for the next group (nearly empty bounces take
     * the bounded grid); either form is correct for any count. */
    uint32_t* liveHintHost = nullptr;   /* pinned */
    hipEvent_t hintEvent = nullptr;
    bool hintPending = false, hintValid = false;
    uint32_t liveHint[PT_MAX_DEPTH + 2] = {};
    int sparseMode = -1;                /* PT_SPARSE: -1 by hint, 0 never, 1 always (testing) */
    int rowsPerGroup = 32;              /* a stream group is worth it from this many iterations on (fixed stream count only) */
    /* Launch-shape choice.  Whether the rows of a launch group run best on one stream (shadow traversal beside the
     * next closest-hit stage) or spread over two or three depends on the scene: with a cache-resident BVH one stream
     * reaches the sum of the stage times and more streams only disturb each other, with a multi-million-triangle
     * mesh the overlap is worth 20 %.  So the first launch groups of a given size try the shapes in turn (GPU time
     * between two events on the main stream, one blocking wait each) and the rest use the fastest.  PT_STREAMS fixes
     * the shape instead.  Results do not depend on the shape. */
    bool streamsFixed = false;
    int tuneStage = 0, tuneChoice = 1;  /* for full groups of `batch` iterations; reset when the batch size is set */
    float tuneMs[kMaxStreamGroups + 1] = {};
    hipEvent_t tuneStart = nullptr, tuneEnd = nullptr;
    /* pt_trace_lookahead: iterations rendered ahead of the calls that ask for them.  `cur` is the launch group whose rows
     * are being handed out (one per call, in order), `next` the group launched behind it on the other half of the path
     * state rows, so that the device renders while the caller consumes.  A row's contribution joins the running sum on the
     * serve stream when its iteration is asked for; the main stream waits for every such addition, so everything
     * issued on it later (the next group that reuses the rows, pt_sync, pt_download) is ordered behind. */
    struct AheadGroup { int base = 0, count = 0, rowBase = 0; bool valid = false; hipEvent_t done = nullptr; };
    AheadGroup aheadCur, aheadNext;
    int aheadServed = 0;
    hipStream_t serveStream = nullptr;
    hipEvent_t servedEvent = nullptr, sumReadyEvent = nullptr;
    bool sumTouchedOnMain = true;       /* the main stream has written the running sum since the serve stream last waited for it */
    /* host image (pt_bind_host_image): the caller's W*H*3 floats, page-locked in place when the runtime allows, else
     * reached through a pinned staging buffer of the library's own */
    float* hostImage = nullptr;
    bool hostRegistered = false;
    float* hostStaging = nullptr;
    hipEvent_t hostCopyDone = nullptr;
    /* through the staging buffer the image travels in kHostCopyParts pieces with an event each, so that the host's own
     * copy of a piece runs while the next piece is still on the bus */
    static constexpr int kHostCopyParts = 4;
    hipEvent_t hostPartDone[kHostCopyParts] = {};
    bool hostCopyPending = false;
    PtCamera cam{};
    int traceDepth = 8;
    int numPixels = 0;
    bool ownImage = true;
    float* imageOwned = nullptr;
    float* imageSum = nullptr;          /* the running sum (owned or caller-bound) */
    float *gNormals = nullptr, *gAlbedo = nullptr;      /* guide buffers of pt_capture_gbuffer (allocated on first use) */
    float *dnA = nullptr, *dnB = nullptr;               /* ping-pong images of pt_denoise */
    bool gbufferValid = false;
    /* options */
    int sortOn = 1, useBvh = 1, timing = 0, counters = 0;
    int batch = 0;                      /* iterations the allocated path state can hold */
    /* bookkeeping */
    std::vector<void*> allocs;
    std::vector<void*> batchAllocs;
    uint64_t iterations = 0;
    std::vector<TimedSpan> spans;
    std::vector<hipEvent_t> freeEvents;
    float ms[ST_COUNT] = {};
    uint32_t launches[ST_COUNT] = {};
};

namespace {

template <typename T>
int devAlloc(PtContext* c, T** out, size_t count)
{
    *out = nullptr;
    if (count == 0) count = 1;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, count * sizeof(T));
    if (e != hipSuccess) {
        pt_set_error("hipMalloc of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e));
        return PT_ENOMEM;
    }
    c->allocs.push_back(p);
    *out = (T*)p;
    return PT_OK;
}

template <typename T>
int batchAlloc(std::vector<void*>& owned, T** out, size_t count, bool zero)
{
    *out = nullptr;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) {
        pt_set_error("hipMalloc of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e));
        return PT_ENOMEM;
    }
    owned.push_back(p);
    if (zero && hipMemset(p, 0, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) { pt_set_error("hipMemset failed"); return PT_EHIP; }
    *out = (T*)p;
    return PT_OK;
}

/* path state, queues and per-iteration contribution buffers for `batch` iterations in flight
 * (about 192 bytes per pixel and iteration).  The new buffers replace the old ones only when every allocation
 * has succeeded; after a failure the context still holds, and keeps using, the state it had.  Path state at the
 * largest group sizes takes most of the device, so the old buffers are released first when the new ones would not
 * fit beside them — in that case a failure leaves the context without path state (batch 0), every entry point that
 * needs it then returns PT_EINVAL, and a later pt_set_option(PT_OPT_BATCH) can allocate again. */
int allocBatchState(PtContext* c, int batch)
{
    const size_t N = (size_t)c->numPixels, B = (size_t)batch;
    size_t freeB = 0, totalB = 0;
    const size_t need = N * B * 203;
    const bool known = hipMemGetInfo(&freeB, &totalB) == hipSuccess;
    if (known && need > totalB) {       /* cannot fit on this device at all: fail before anything is released */
        pt_set_error("path state for %d iterations at %d pixels needs about %zu MB, the device has %zu MB", batch, c->numPixels, need >> 20, totalB >> 20);
        return PT_ENOMEM;
    }
    if (!c->batchAllocs.empty() && (!known || freeB < need + (size_t(1) << 30))) {
        for (void* p : c->batchAllocs) (void)hipFree(p);
        c->batchAllocs.clear();
        c->batch = 0;
        c->state = PathState{};
        c->queues = Queues{};
    }
    PathState P{};
    Queues Q{};
    std::vector<void*> owned;
    P.numPixels = N;
    Q.numPixels = c->numPixels;
    Q.numBuckets = c->view.numBuckets;
    Q.numBlocks = (int)((N + kBlock - 1) / kBlock);
    int rc = PT_OK;
#define BA(ptr, count, zero) if (rc == PT_OK) rc = batchAlloc(owned, &(ptr), (count), (zero))
    if (kRecStride == 1) {
        BA(P.rayO.p, N * B, false); BA(P.rayD.p, N * B, false); BA(P.col.p, N * B, false); BA(P.hitA.p, N * B, false);
        P.rayD0 = P.rayD.p; P.hitA0 = P.hitA.p;
    } else {
        /* one aligned 64-byte record per path: origin, direction, colour, first half of the hit record */
        static_assert(kRecStride == 1 || kRecStride == 4, "path records are separate arrays or one 64-byte record per path");
        float4* block = nullptr;
        BA(block, N * B * 4, false);
        P.rayO.p = block; P.rayD.p = block + 1; P.col.p = block + 2; P.hitA.p = block + 3;
        BA(P.rayD0, N * B, false); BA(P.hitA0, N * B, false);
    }
    BA(P.hitB, N * B, false);
    BA(P.hitTri, N * B, false);
    BA(P.shRec, N * B * 3, false); BA(P.traceQueue, N * B, false); BA(P.refList, N * B * 2, false);
    BA(P.image, N * B * 3, true);
    BA(Q.q0, N * B, false); BA(Q.q1, N * B, false); BA(Q.keys, N * B, false);
    BA(Q.blockHist, (size_t)Q.numBuckets * Q.numBlocks * B, false);
    BA(Q.blockAlive, (size_t)Q.numBlocks * B, false);
    BA(Q.live, (size_t)(PT_MAX_DEPTH + 2) * B, true);
#undef BA
    if (rc != PT_OK) {
        for (void* p : owned) (void)hipFree(p);
        return rc;
    }
    for (void* p : c->batchAllocs) (void)hipFree(p);
    c->batchAllocs = std::move(owned);
    c->state = P;
    c->queues = Q;
    c->batch = batch;
    c->tuneStage = 0;           /* the launch shape is chosen for full groups of this size */
    c->tuneChoice = 1;
    return PT_OK;
}

template <typename T>
int devUpload(PtContext* c, T** out, const T* host, size_t count)
{
    int rc = devAlloc(c, out, count);
    if (rc != PT_OK) return rc;
    if (count) HIP_TRY(hipMemcpy(*out, host, count * sizeof(T), hipMemcpyHostToDevice));
    return PT_OK;
}

/* getGeomArea (src/pathtrace.cu:217-230) */
float geomArea(const PtGeom& g)
{
    const float PI = kPI;
    if (g.type == PT_SPHERE) { float radius = g.scale[0] * 0.5f; return 4.0f * PI * radius * radius; }
    if (g.type == PT_CUBE) { float sx = g.scale[0], sy = g.scale[1], sz = g.scale[2]; return 2.0f * (sx * sy + sy * sz + sz * sx); }
    return 1.0f;
}

/* passes of the walked tree's reinsertion optimiser (pt_set_tree_passes); PT_UPPER_OPT sets the start value */
static std::atomic<int> g_treePasses{[] { const char* e = getenv("PT_UPPER_OPT"); const int v = e ? atoi(e) : 3; return v < 0 ? 0 : v > 16 ? 16 : v; }()};

/* luminance * sin(theta) tables: conditional CDF per row, marginal CDF over rows
 * (buildEnvMapDistribution, src/pathtrace.cu:336-425; host arithmetic, libm sinf) */
void buildEnvCdfs(const PtEnvMap& env, std::vector<float>& marginal, std::vector<float>& conditional, float& totalPower)
{
    const int W = env.width, H = env.height;
    const float PI = kPI;
    conditional.assign((size_t)W * H, 0.0f);
    marginal.assign(H, 0.0f);
    std::vector<float> rowPower(H);
    for (int y = 0; y < H; ++y) {
        float theta = (y + 0.5f) * PI / H;
        float st = sinf(theta);
        float rowSum = 0.0f;
        for (int x = 0; x < W; ++x) {
            const float* c = env.data + 3 * ((size_t)y * W + x);
            float lum = 0.299f * c[0] + 0.587f * c[1] + 0.114f * c[2];
            lum *= st;
            rowSum += lum;
            conditional[(size_t)y * W + x] = rowSum;
        }
        rowPower[y] = rowSum;
        if (rowSum > 0) {
            float inv = 1.0f / rowSum;
            for (int x = 0; x < W; ++x) conditional[(size_t)y * W + x] *= inv;
        }
    }
    float total = 0.0f;
    for (int y = 0; y < H; ++y) { total += rowPower[y]; marginal[y] = total; }
    totalPower = total;
    if (total > 0) { float inv = 1.0f / total; for (int y = 0; y < H; ++y) marginal[y] *= inv; }
}

struct FlatMesh {
    std::vector<float4> nodes;     /* 8 per 4-wide node */
    std::vector<float4> hot;       /* 3 per leaf-order triangle */
    std::vector<float4> cold;      /* 4 per leaf-order triangle */
    std::vector<int> leafOfOrig;
    std::vector<float4> box;       /* 2 per leaf-order triangle: min, max of its leaf in the host BVH (decides whether it may be hit) */
    int depth4 = 0;
    bool hasBvh = false;
};

inline float asF(int v) { float f; memcpy(&f, &v, 4); return f; }

void pushTriangle(FlatMesh& fm, const PtTriangle& t, int orig)
{
    /* edges precomputed with the same single subtraction the test would do */
    float e1[3], e2[3];
    for (int k = 0; k < 3; ++k) { e1[k] = t.v1[k] - t.v0[k]; e2[k] = t.v2[k] - t.v0[k]; }
    fm.hot.push_back(make_float4(t.v0[0], t.v0[1], t.v0[2], e1[0]));
    fm.hot.push_back(make_float4(e1[1], e1[2], e2[0], e2[1]));
    fm.hot.push_back(make_float4(e2[2], asF(orig), asF(0x7fffffff), 0.0f));        /* .z: visit rank, set below */
    const float inf = std::numeric_limits<float>::infinity();
    fm.box.push_back(make_float4(-inf, -inf, -inf, 0.0f));                         /* box of its host leaf, set below */
    fm.box.push_back(make_float4(inf, inf, inf, 0.0f));
    fm.cold.push_back(make_float4(t.n0[0], t.n0[1], t.n0[2], t.n1[0]));
    fm.cold.push_back(make_float4(t.n1[1], t.n1[2], t.n2[0], t.n2[1]));
    fm.cold.push_back(make_float4(t.n2[2], t.uv0[0], t.uv0[1], t.uv1[0]));
    fm.cold.push_back(make_float4(t.uv1[1], t.uv2[0], t.uv2[1], asF(t.materialId)));
}

/* The hierarchy ABOVE the leaves is free: a traversal result depends on the reference's leaves and their
 * boxes only (which triangles get tested, and their visit ranks).  UpperTree rebuilds it over the leaves of
 * the host BVH, top-down with a binned surface-area heuristic on the leaf boxes (32 bins per axis, all three
 * axes, cost = area x triangles), median split on the widest axis when binning cannot separate or the tree
 * gets deep, and emits 4-wide nodes by opening, per node, the child with the largest box. */
struct UpperTree {
    struct Leaf { float lo[3], hi[3], cen[3]; int first, count; };
    struct Node { float lo[3], hi[3]; int left, right, leaf; };   /* leaf >= 0: index into leaves */
    std::vector<Leaf> leaves;
    std::vector<int> order;
    std::vector<Node> nodes;
    static constexpr int kBins = 32;

    static float halfArea(const float* lo, const float* hi)
    {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
    static void grow(float* lo, float* hi, const float* l2, const float* h2)
    {
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], l2[k]); hi[k] = std::max(hi[k], h2[k]); }
    }
    int build(int first, int last, int depth)
    {
        const int id = (int)nodes.size();
        nodes.push_back(Node{});
        float lo[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, hi[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
        float clo[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, chi[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
        for (int i = first; i < last; ++i) {
            const Leaf& L = leaves[order[i]];
            grow(lo, hi, L.lo, L.hi);
            grow(clo, chi, L.cen, L.cen);
        }
        Node nd{};
        memcpy(nd.lo, lo, 12); memcpy(nd.hi, hi, 12);
        nd.left = nd.right = -1; nd.leaf = -1;
        if (last - first == 1) { nd.leaf = order[first]; nodes[id] = nd; return id; }
        int mid = -1;
        if (depth < 28) {
            float bestCost = 3.4e38f; int bestAxis = -1, bestBin = -1;
            for (int axis = 0; axis < 3; ++axis) {
                const float ext = chi[axis] - clo[axis];
                if (!(ext > 0.0f)) continue;
                const float scale = kBins / ext;
                float blo[kBins][3], bhi[kBins][3]; int bn[kBins];
                for (int b = 0; b < kBins; ++b) { for (int k = 0; k < 3; ++k) { blo[b][k] = 3.4e38f; bhi[b][k] = -3.4e38f; } bn[b] = 0; }
                for (int i = first; i < last; ++i) {
                    const Leaf& L = leaves[order[i]];
                    int b = std::min(kBins - 1, std::max(0, (int)((L.cen[axis] - clo[axis]) * scale)));
                    grow(blo[b], bhi[b], L.lo, L.hi);
                    bn[b] += L.count;
                }
                float rlo[kBins][3], rhi[kBins][3]; int rn[kBins];
                float alo[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, ahi[3] = { -3.4e38f, -3.4e38f, -3.4e38f }; int an = 0;
                for (int b = kBins - 1; b > 0; --b) {
                    if (bn[b]) grow(alo, ahi, blo[b], bhi[b]);
                    an += bn[b];
                    memcpy(rlo[b], alo, 12); memcpy(rhi[b], ahi, 12); rn[b] = an;
                }
                float llo[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, lhi[3] = { -3.4e38f, -3.4e38f, -3.4e38f }; int ln = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    if (bn[b]) grow(llo, lhi, blo[b], bhi[b]);
                    ln += bn[b];
                    if (ln == 0 || rn[b + 1] == 0) continue;
                    float cost = halfArea(llo, lhi) * ln + halfArea(rlo[b + 1], rhi[b + 1]) * rn[b + 1];
                    if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestBin = b; }
                }
            }
            if (bestAxis >= 0) {
                const float scale = kBins / (chi[bestAxis] - clo[bestAxis]);
                auto it = std::partition(order.begin() + first, order.begin() + last, [&](int li) {
                    int b = std::min(kBins - 1, std::max(0, (int)((leaves[li].cen[bestAxis] - clo[bestAxis]) * scale)));
                    return b <= bestBin;
                });
                mid = (int)(it - order.begin());
            }
        }
        if (mid <= first || mid >= last) {
            int axis = 0;
            for (int k = 1; k < 3; ++k) if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
            mid = (first + last) / 2;
            std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + last,
                             [&](int a, int b) { return leaves[a].cen[axis] < leaves[b].cen[axis]; });
        }
        int l = build(first, mid, depth + 1);
        int r = build(mid, last, depth + 1);
        nd.left = l; nd.right = r;
        nodes[id] = nd;
        return id;
    }
};

/* Insertion-based optimisation of the rebuilt hierarchy (after Bittner, Hapala, Havran: "Fast insertion-based
 * optimization of bounding volume hierarchies", 2013).  The top-down binned build decides every split once, with the
 * coarse view it has at that level; here subtrees are taken out and put back where they add the least surface area,
 * largest boxes first, for a few passes.  Only the tree ABOVE the reference's leaves moves — the leaves, their boxes
 * and so the traversal's results stay what they are; what changes is how many boxes a ray tests on its way down. */
struct UpperOptimizer {
    UpperTree& U;
    std::vector<int> parent;
    int root;
    explicit UpperOptimizer(UpperTree& u, int r) : U(u), parent(u.nodes.size(), -1), root(r)
    {
        for (size_t i = 0; i < U.nodes.size(); ++i)
            if (U.nodes[i].leaf < 0) { parent[U.nodes[i].left] = (int)i; parent[U.nodes[i].right] = (int)i; }
    }
    static float area(const UpperTree::Node& n) { return UpperTree::halfArea(n.lo, n.hi); }
    static float unionArea(const UpperTree::Node& a, const UpperTree::Node& b)
    {
        float lo[3], hi[3];
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(a.lo[k], b.lo[k]); hi[k] = std::max(a.hi[k], b.hi[k]); }
        return UpperTree::halfArea(lo, hi);
    }
    void refitFrom(int i)
    {
        for (; i >= 0; i = parent[i]) {
            UpperTree::Node& n = U.nodes[i];
            const UpperTree::Node &a = U.nodes[n.left], &b = U.nodes[n.right];
            float lo[3], hi[3];
            bool same = true;
            for (int k = 0; k < 3; ++k) {
                lo[k] = std::min(a.lo[k], b.lo[k]); hi[k] = std::max(a.hi[k], b.hi[k]);
                same = same && lo[k] == n.lo[k] && hi[k] == n.hi[k];
            }
            if (same) break;                       /* nothing changes further up either */
            memcpy(n.lo, lo, 12); memcpy(n.hi, hi, 12);
        }
    }
    /* take subtree n out (its parent p disappears, the sibling moves up); returns the freed node p */
    int detach(int n)
    {
        const int p = parent[n], g = parent[p];
        const int sib = U.nodes[p].left == n ? U.nodes[p].right : U.nodes[p].left;
        if (U.nodes[g].left == p) U.nodes[g].left = sib; else U.nodes[g].right = sib;
        parent[sib] = g;
        refitFrom(g);
        return p;
    }
    /* the node x beside which n costs least: area(x + n) plus what the boxes above x grow by; branch and bound over
     * the growth accumulated on the way down */
    int bestSibling(int n)
    {
        const UpperTree::Node& N = U.nodes[n];
        const float areaN = area(N);
        struct Item { float inherited; int node; };
        auto cmp = [](const Item& a, const Item& b) { return a.inherited > b.inherited; };
        std::vector<Item> heap;
        heap.push_back({ 0.0f, root });
        float best = 3.4e38f;
        int bestNode = root;
        int visited = 0;
        while (!heap.empty() && visited < 4096) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            Item it = heap.back();
            heap.pop_back();
            if (it.inherited + areaN >= best) break;            /* no place below can beat the best found */
            ++visited;
            const UpperTree::Node& X = U.nodes[it.node];
            const float direct = unionArea(X, N);
            const float total = it.inherited + direct;
            if (total < best) { best = total; bestNode = it.node; }
            if (X.leaf < 0) {
                const float down = it.inherited + (direct - area(X));   /* growth of X if n goes somewhere below it */
                if (down + areaN < best) {
                    heap.push_back({ down, X.left }); std::push_heap(heap.begin(), heap.end(), cmp);
                    heap.push_back({ down, X.right }); std::push_heap(heap.begin(), heap.end(), cmp);
                }
            }
        }
        return bestNode;
    }
    void attach(int n, int x, int freeNode)
    {
        const int g = parent[x];
        UpperTree::Node& P = U.nodes[freeNode];
        P.leaf = -1; P.left = x; P.right = n;
        for (int k = 0; k < 3; ++k) { P.lo[k] = std::min(U.nodes[x].lo[k], U.nodes[n].lo[k]); P.hi[k] = std::max(U.nodes[x].hi[k], U.nodes[n].hi[k]); }
        parent[freeNode] = g;
        if (g < 0) root = freeNode;
        else if (U.nodes[g].left == x) U.nodes[g].left = freeNode; else U.nodes[g].right = freeNode;
        parent[x] = freeNode; parent[n] = freeNode;
        refitFrom(g);
    }
    float cost() const
    {
        double c = 0.0;
        std::vector<int> st{ root };
        while (!st.empty()) { int i = st.back(); st.pop_back(); const UpperTree::Node& n = U.nodes[i]; if (n.leaf < 0) { c += area(n); st.push_back(n.left); st.push_back(n.right); } }
        return (float)(c / area(U.nodes[root]));
    }
    /* passes over the nodes in order of decreasing box area; `budget` bounds the number of reinsertions per pass */
    int run(int passes, size_t budget)
    {
        for (int pass = 0; pass < passes; ++pass) {
            std::vector<std::pair<float, int>> cand;
            cand.reserve(U.nodes.size());
            for (size_t i = 0; i < U.nodes.size(); ++i) {
                const int p = parent[i];
                if ((int)i == root || p < 0 || p == root) continue;       /* the root's children stay: their parent cannot be freed */
                cand.push_back({ area(U.nodes[i]), (int)i });
            }
            const size_t take = std::min(budget, cand.size());
            std::partial_sort(cand.begin(), cand.begin() + take, cand.end(), [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first > b.first; });
            for (size_t k = 0; k < take; ++k) {
                const int n = cand[k].second;
                const int p = parent[n];
                if (p < 0 || p == root || n == root) continue;          /* the tree moved since the list was made */
                const int freeNode = detach(n);
                const int x = bestSibling(n);
                attach(n, x, freeNode);
            }
        }
        return root;
    }
};

/* Turns the host BVH (40-byte binary nodes + index array, src/bvh.h:55-65) into the device
 * layout: triangles permuted into leaf order, each stamped with its rank in the reference's
 * visit order (right subtree first, src/intersections.cu:379-384), and the inner levels
 * collapsed into 4-wide nodes. */
struct Flattener {
    const PtTriangle* tris; int ntri;
    const PtBvhNode* src; int nsrc;
    const int32_t* indices; int nidx;
    FlatMesh& fm;
    std::vector<int> leafFirst;         /* per source node: first leaf-order triangle */
    std::vector<char> seen;
    int error = PT_OK;

    bool check(int i)
    {
        if (i < 0 || i >= nsrc) { pt_set_error("BVH child index %d out of range", i); error = PT_EINVAL; return false; }
        return true;
    }
    /* pass 1 (left-first DFS): leaf order; detects cycles and bad ranges */
    void emitLeaves(int root)
    {
        std::vector<std::pair<int, int>> work;
        work.push_back({ root, 0 });
        int leafPos = 0;
        while (!work.empty() && error == PT_OK) {
            auto [i, depth] = work.back();
            work.pop_back();
            if (!check(i)) return;
            if (seen[i]) { pt_set_error("BVH is not a tree (node %d reached twice)", i); error = PT_EINVAL; return; }
            if (depth > 62) { pt_set_error("BVH deeper than 62 levels is not supported"); error = PT_ELIMIT; return; }
            seen[i] = 1;
            const PtBvhNode& n = src[i];
            if (n.leftChild == -1) {
                if (n.triangleCount <= 0 || n.triangleStart < 0 || n.triangleStart + n.triangleCount > nidx) {
                    pt_set_error("BVH leaf %d has an invalid triangle range", i); error = PT_EINVAL; return;
                }
                leafFirst[i] = leafPos;
                for (int k = 0; k < n.triangleCount; ++k) {
                    int orig = indices[n.triangleStart + k];
                    if (orig < 0 || orig >= ntri) { pt_set_error("BVH triangle index out of range"); error = PT_EINVAL; return; }
                    pushTriangle(fm, tris[orig], orig);
                    if (fm.leafOfOrig[orig] < 0) fm.leafOfOrig[orig] = leafPos;
                    ++leafPos;
                }
            } else {
                work.push_back({ n.rightChild, depth + 1 });
                work.push_back({ n.leftChild, depth + 1 });
            }
        }
    }
    /* pass 2 (right-first DFS): rank = position in the reference's visit order */
    void stampRanks(int root)
    {
        std::vector<int> work;
        work.push_back(root);
        int rank = 0;
        while (!work.empty()) {
            int i = work.back();
            work.pop_back();
            const PtBvhNode& n = src[i];
            if (n.leftChild == -1) {
                for (int k = 0; k < n.triangleCount; ++k) fm.hot[3 * (size_t)(leafFirst[i] + k) + 2].z = asF(rank++);
            } else {
                work.push_back(n.leftChild);     /* popped second */
                work.push_back(n.rightChild);    /* popped first  */
            }
        }
    }
    static float area(const PtBvhNode& n)
    {
        float dx = n.bmax[0] - n.bmin[0], dy = n.bmax[1] - n.bmin[1], dz = n.bmax[2] - n.bmin[2];
        return dx * dy + dy * dz + dz * dx;
    }
    int newNode4()
    {
        int id = (int)(fm.nodes.size() / 8);
        /* an empty slot is a box at +infinity: its slab distances are infinite, so the plain test rejects it
         * (entry beyond any hit, or exit behind the origin) without a look at the reference word */
        const float inf = std::numeric_limits<float>::infinity();
        for (int k = 0; k < 6; ++k) fm.nodes.push_back(make_float4(inf, inf, inf, inf));
        fm.nodes.push_back(make_float4(asF(kEmptyRef), asF(kEmptyRef), asF(kEmptyRef), asF(kEmptyRef)));
        fm.nodes.push_back(make_float4(0, 0, 0, 0));
        return id;
    }
    void setChild(int node4, int slot, const float* bmin, const float* bmax, int ref)
    {
        float4* q = &fm.nodes[8 * (size_t)node4];
        for (int k = 0; k < 3; ++k) { (&q[k].x)[slot] = bmin[k]; (&q[3 + k].x)[slot] = bmax[k]; }
        (&q[6].x)[slot] = asF(ref);
    }
    /* reference to `count` triangles starting at `first`, all under one box; more than 32 triangles
       (possible only at the builder's depth cap) fan out over extra nodes that repeat the leaf box */
    int leafRef(int first, int count, const float* bmin, const float* bmax, int depth)
    {
        if (count <= kLeafMaxTris) return ~((first << 5) | (count - 1));
        int id = newNode4();
        fm.depth4 = std::max(fm.depth4, depth + 1);
        int per = (count + 3) / 4;
        per = std::max(per, 1);
        for (int s = 0, at = 0; s < 4 && at < count; ++s, at += per) {
            int c = std::min(per, count - at);
            int r = leafRef(first + at, c, bmin, bmax, depth + 1);
            setChild(id, s, bmin, bmax, r);
        }
        return id;
    }
    /* same collapse over a rebuilt upper tree */
    int collapseUpper(const UpperTree& U, int i, int depth)
    {
        int id = newNode4();
        fm.depth4 = std::max(fm.depth4, depth + 1);
        int kids[4] = { U.nodes[i].left, U.nodes[i].right, -1, -1 };
        int nk = 2;
        while (nk < 4) {
            int pick = -1;
            float bestA = -1.0f;
            for (int k = 0; k < nk; ++k) {
                const UpperTree::Node& c = U.nodes[kids[k]];
                float a = UpperTree::halfArea(c.lo, c.hi);
                if (c.leaf < 0 && a > bestA) { bestA = a; pick = k; }
            }
            if (pick < 0) break;
            int c = kids[pick];
            for (int k = nk; k > pick + 1; --k) kids[k] = kids[k - 1];
            kids[pick] = U.nodes[c].left;
            kids[pick + 1] = U.nodes[c].right;
            ++nk;
        }
        for (int k = 0; k < nk; ++k) {
            const UpperTree::Node& c = U.nodes[kids[k]];
            int ref;
            if (c.leaf >= 0) {
                const UpperTree::Leaf& L = U.leaves[c.leaf];
                ref = leafRef(L.first, L.count, L.lo, L.hi, depth + 1);
                setChild(id, k, L.lo, L.hi, ref);
            } else {
                ref = collapseUpper(U, kids[k], depth + 1);
                setChild(id, k, c.lo, c.hi, ref);
            }
        }
        return id;
    }
    /* inner source node -> 4-wide node: open the child with the largest box until four slots are used */
    int collapse(int i, int depth)
    {
        int id = newNode4();
        fm.depth4 = std::max(fm.depth4, depth + 1);
        int kids[4] = { src[i].leftChild, src[i].rightChild, -1, -1 };
        int nk = 2;
        while (nk < 4) {
            int pick = -1;
            float bestA = -1.0f;
            for (int k = 0; k < nk; ++k)
                if (src[kids[k]].leftChild != -1 && area(src[kids[k]]) > bestA) { bestA = area(src[kids[k]]); pick = k; }
            if (pick < 0) break;
            int c = kids[pick];
            for (int k = nk; k > pick + 1; --k) kids[k] = kids[k - 1];
            kids[pick] = src[c].leftChild;
            kids[pick + 1] = src[c].rightChild;
            ++nk;
        }
        for (int k = 0; k < nk; ++k) {
            const PtBvhNode& c = src[kids[k]];
            int ref = (c.leftChild == -1) ? leafRef(leafFirst[kids[k]], c.triangleCount, c.bmin, c.bmax, depth + 1)
                                          : collapse(kids[k], depth + 1);
            setChild(id, k, c.bmin, c.bmax, ref);
        }
        return id;
    }
};

/* The first `top` 4-wide nodes in breadth-first order get the numbers 0 .. top-1 (the root stays 0); the others keep their
 * depth-first order behind them, which keeps a subtree's nodes close together in memory.  A traversal workgroup can then
 * hold "nodes 0 .. k-1" in LDS for any k <= top and has the top levels of the tree. */
void numberTopBreadthFirst(FlatMesh& fm, int top)
{
    const int n = (int)(fm.nodes.size() / 8);
    if (n < 2) return;
    auto childRef = [&](int node, int slot) { int r; float f = (&fm.nodes[8 * (size_t)node + 6].x)[slot]; memcpy(&r, &f, 4); return r; };
    std::vector<int> order;                 /* breadth-first prefix */
    order.push_back(0);
    for (size_t at = 0; at < order.size() && (int)order.size() < top; ++at)
        for (int s = 0; s < 4 && (int)order.size() < top; ++s) {
            const int r = childRef(order[at], s);
            if (r >= 0 && r != kEmptyRef && r < n) order.push_back(r);
        }
    std::vector<int> newId(n, -1);
    for (size_t i = 0; i < order.size(); ++i) newId[order[i]] = (int)i;
    int next = (int)order.size();
    for (int i = 0; i < n; ++i) if (newId[i] < 0) newId[i] = next++;
    std::vector<float4> out(fm.nodes.size());
    for (int i = 0; i < n; ++i) {
        float4* dst = &out[8 * (size_t)newId[i]];
        for (int k = 0; k < 8; ++k) dst[k] = fm.nodes[8 * (size_t)i + k];
        for (int s = 0; s < 4; ++s) {
            const int r = childRef(i, s);
            if (r >= 0 && r != kEmptyRef && r < n) (&dst[6].x)[s] = asF(newId[r]);
        }
    }
    fm.nodes.swap(out);
}

int flattenMesh(const PtTriangle* tris, int ntri, const PtBvhNode* nodes, int nnodes, const int32_t* indices, int nidx,
                FlatMesh& fm)
{
    fm.leafOfOrig.assign(std::max(ntri, 1), -1);
    if (ntri == 0 || nnodes == 0) { fm.hasBvh = false; return PT_OK; }
    if (ntri >= (1 << 26)) { pt_set_error("meshes are limited to %d triangles", (1 << 26) - 1); return PT_ELIMIT; }
    Flattener F{ tris, ntri, nodes, nnodes, indices, nidx, fm, std::vector<int>(nnodes, -1), std::vector<char>(nnodes, 0) };
    F.emitLeaves(0);
    if (F.error != PT_OK) return F.error;
    F.stampRanks(0);
    /* triangles no leaf references (possible only with a caller-supplied BVH) stay reachable
       for the linear fallback */
    int leafPos = (int)(fm.hot.size() / 3);
    for (int i = 0; i < ntri; ++i)
        if (fm.leafOfOrig[i] < 0) { pushTriangle(fm, tris[i], i); fm.leafOfOrig[i] = leafPos++; }
    /* The leaves of the host BVH (the reference's) decide WHICH triangles a ray may hit: the reference tests a leaf's
     * triangles only when the ray passes the leaf's box (and every ancestor's, which for nested boxes and monotone slab
     * arithmetic follows from the leaf's).  They do not have to be the leaves that are walked.  Every triangle carries
     * the box of its host leaf, and the traversal checks a triangle hit against that box with the reference's own
     * arithmetic before it may become the closest hit.  The hierarchy that is walked is then free, down to boxes
     * around single triangles — median-split leaves of four triangles are loose (a ray tests 12 to 69 triangles in
     * them on the test scenes, 3 to 4 with a box each). */
    for (int i = 0; i < nnodes; ++i) {
        if (!F.seen[i] || nodes[i].leftChild != -1) continue;
        for (int k = 0; k < nodes[i].triangleCount; ++k) {
            float4* b = &fm.box[2 * (size_t)(F.leafFirst[i] + k)];
            b[0] = make_float4(nodes[i].bmin[0], nodes[i].bmin[1], nodes[i].bmin[2], 0.0f);
            b[1] = make_float4(nodes[i].bmax[0], nodes[i].bmax[1], nodes[i].bmax[2], 0.0f);
        }
    }
    {
        static const bool rebuildUpper = [] { const char* e = getenv("PT_UPPER_TREE"); return !e || atoi(e) != 0; }();
        bool done = false;
        if (rebuildUpper) {
            UpperTree U;
            float ext = 0.0f;
            for (int k = 0; k < 3; ++k) ext = std::max(ext, nodes[0].bmax[k] - nodes[0].bmin[k]);
            /* The walked boxes have to contain every point at which the triangle test can report a hit: the test accepts
             * points a rounding error outside the triangle's rim, and that error is a few units in the last place of the
             * largest coordinate difference it forms (origin - vertex, an edge), whatever the size of the hit distance.
             * For origins within a few mesh extents that is a few parts in 10^7 of the extent; the pad is 2^-17 of it
             * (64 such units).  For origins far outside, the distance itself is as large as those differences and the
             * relative slack of the box test (kBoxSlack) covers the error.  The pad also gives a flat box some thickness. */
            const float pad = ext * (1.0f / 131072.0f);
            for (int i = 0; i < nnodes; ++i) {
                if (!F.seen[i] || nodes[i].leftChild != -1) continue;
                for (int j = 0; j < nodes[i].triangleCount; ++j) {
                    UpperTree::Leaf L;
                    L.first = F.leafFirst[i] + j; L.count = 1;
                    int orig; float f = fm.hot[3 * (size_t)L.first + 2].y; memcpy(&orig, &f, 4);
                    const PtTriangle& t = tris[orig];
                    for (int k = 0; k < 3; ++k) {
                        L.lo[k] = std::min(t.v0[k], std::min(t.v1[k], t.v2[k])) - pad;
                        L.hi[k] = std::max(t.v0[k], std::max(t.v1[k], t.v2[k])) + pad;
                        L.cen[k] = 0.5f * (L.lo[k] + L.hi[k]);
                    }
                    U.leaves.push_back(L);
                }
            }
            U.order.resize(U.leaves.size());
            for (size_t i = 0; i < U.order.size(); ++i) U.order[i] = (int)i;
            U.nodes.reserve(2 * U.leaves.size());
            if (U.leaves.size() == 1) {
                const UpperTree::Leaf& L = U.leaves[0];
                int id = F.newNode4();
                fm.depth4 = 1;
                F.setChild(id, 0, L.lo, L.hi, F.leafRef(L.first, 1, L.lo, L.hi, 1));
                done = true;
            } else if (U.leaves.size() >= 2) {
                int root = U.build(0, (int)U.leaves.size(), 0);
                const int optPasses = g_treePasses.load();
                if (optPasses > 0 && U.leaves.size() >= 8) {
                    UpperOptimizer O(U, root);
                    const float before = getenv("PT_DEBUG_TAIL") ? O.cost() : 0.0f;
                    root = O.run(optPasses, std::min<size_t>(U.nodes.size(), 600000));
                    if (getenv("PT_DEBUG_TAIL")) fprintf(stderr, "[pt] walked tree over %zu triangles: surface-area cost %.2f -> %.2f after %d reinsertion passes\n", U.leaves.size(), before, O.cost(), optPasses);
                }
                /* Leaves of two: a node whose two children are single triangles becomes one leaf of both — one slot of a
                 * 4-wide node then leads to two triangles, the tree above loses a seventh of its nodes, and a ray tests
                 * 44 instead of 47 boxes and 3.9 instead of 3.0 triangles on the dragon-class model (+6 % throughput;
                 * leaves of three or four, or pairs of triangles that are not siblings in this tree, lose).  The triangle
                 * records are put in the order of the walked tree, so that the two lie side by side.
                 * PT_UPPER_PAIR = largest box(pair) / (box(a) + box(b)), in percent, that is merged; 0: none. */
                static const int pairPct = [] { const char* e = getenv("PT_UPPER_PAIR"); return e ? atoi(e) : 100; }();
                std::vector<int> newPos;
                if (pairPct > 0) {
                    std::vector<std::pair<int, int>> members(U.leaves.size(), { -1, -1 });
                    const size_t nn0 = U.nodes.size();
                    for (size_t i = 0; i < nn0; ++i) {
                        UpperTree::Node& n = U.nodes[i];
                        if (n.leaf >= 0 || n.left < 0) continue;
                        const UpperTree::Node &a = U.nodes[n.left], &b = U.nodes[n.right];
                        if (a.leaf < 0 || b.leaf < 0 || U.leaves[a.leaf].count != 1 || U.leaves[b.leaf].count != 1) continue;
                        if (UpperTree::halfArea(n.lo, n.hi) * 100.0f > (float)pairPct * (UpperTree::halfArea(a.lo, a.hi) + UpperTree::halfArea(b.lo, b.hi))) continue;
                        UpperTree::Leaf L{};
                        memcpy(L.lo, n.lo, 12); memcpy(L.hi, n.hi, 12);
                        for (int k = 0; k < 3; ++k) L.cen[k] = 0.5f * (L.lo[k] + L.hi[k]);
                        L.first = -1; L.count = 2;
                        members.push_back({ a.leaf, b.leaf });
                        U.leaves.push_back(L);
                        n.leaf = (int)U.leaves.size() - 1; n.left = n.right = -1;
                    }
                    /* new positions in left-first order of the walked tree */
                    const int nref = (int)(fm.hot.size() / 3);
                    newPos.assign(nref, -1);
                    int at = 0;
                    std::vector<int> work{ root };
                    while (!work.empty()) {
                        const int i = work.back(); work.pop_back();
                        UpperTree::Node& n = U.nodes[i];
                        if (n.leaf < 0) { work.push_back(n.right); work.push_back(n.left); continue; }
                        UpperTree::Leaf& L = U.leaves[n.leaf];
                        if (members[n.leaf].first >= 0) {
                            newPos[U.leaves[members[n.leaf].first].first] = at;
                            newPos[U.leaves[members[n.leaf].second].first] = at + 1;
                        } else newPos[L.first] = at;
                        L.first = at;
                        at += L.count;
                    }
                    for (int i = 0; i < nref; ++i) if (newPos[i] < 0) newPos[i] = at++;      /* triangles no leaf references */
                }
                F.collapseUpper(U, root, 0);
                done = 3 * fm.depth4 + 1 <= kLdsStack + kOverflowStack;
                if (done && !newPos.empty()) {
                    const int nref = (int)newPos.size();
                    std::vector<float4> hot(fm.hot.size()), cold(fm.cold.size()), box(fm.box.size());
                    for (int i = 0; i < nref; ++i) {
                        const int j = newPos[i];
                        for (int k = 0; k < 3; ++k) hot[3 * (size_t)j + k] = fm.hot[3 * (size_t)i + k];
                        for (int k = 0; k < 4; ++k) cold[4 * (size_t)j + k] = fm.cold[4 * (size_t)i + k];
                        for (int k = 0; k < 2; ++k) box[2 * (size_t)j + k] = fm.box[2 * (size_t)i + k];
                    }
                    fm.hot.swap(hot); fm.cold.swap(cold); fm.box.swap(box);
                    for (int& v : fm.leafOfOrig) if (v >= 0) v = newPos[v];
                }
                if (!done) { fm.nodes.clear(); fm.depth4 = 0; }      /* too deep: fall back to the host topology */
            }
        }
        if (!done) {
            if (nodes[0].leftChild == -1) {
                int id = F.newNode4();
                fm.depth4 = 1;
                F.setChild(id, 0, nodes[0].bmin, nodes[0].bmax, F.leafRef(F.leafFirst[0], nodes[0].triangleCount, nodes[0].bmin, nodes[0].bmax, 1));
            } else F.collapse(0, 0);
        }
    }
    /* every visit pushes at most three entries */
    if (3 * fm.depth4 + 1 > kLdsStack + kOverflowStack) {
        pt_set_error("BVH too deep for the traversal stack (%d wide levels)", fm.depth4);
        return PT_ELIMIT;
    }
    numberTopBreadthFirst(fm, 85);          /* root + three full levels */
    fm.hasBvh = true;
    return PT_OK;
}

hipEvent_t takeEvent(PtContext* c)
{
    if (!c->freeEvents.empty()) { hipEvent_t e = c->freeEvents.back(); c->freeEvents.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct SpanGuard {
    PtContext* c; int stage; hipStream_t stream; hipEvent_t a = nullptr;
    SpanGuard(PtContext* ctx, int st, hipStream_t s = nullptr) : c(ctx), stage(st), stream(s ? s : ctx->stream)
    {
        c->launches[stage]++;
        if (c->timing) { a = takeEvent(c); (void)hipEventRecord(a, stream); }
    }
    ~SpanGuard()
    {
        if (c->timing) { hipEvent_t b = takeEvent(c); (void)hipEventRecord(b, stream); c->spans.push_back({ a, b, stage }); }
    }
};

void resolveSpans(PtContext* c)
{
    for (auto& s : c->spans) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) c->ms[s.stage] += ms;
        c->freeEvents.push_back(s.a);
        c->freeEvents.push_back(s.b);
    }
    c->spans.clear();
}

} // namespace

extern "C" {

int pt_bvh_device_layout(const PtTriangle* triangles, int triangleCount, const PtBvh* bvh, PtLayoutStats* out)
{
    if (!triangles || triangleCount <= 0 || !bvh || !out) { pt_set_error("pt_bvh_device_layout: bad argument"); return PT_EINVAL; }
    int nn = 0, ni = 0;
    const PtBvhNode* nodes = pt_bvh_nodes(bvh, &nn);
    const int32_t* idx = pt_bvh_indices(bvh, &ni);
    FlatMesh fm;
    int rc = flattenMesh(triangles, triangleCount, nodes, nn, idx, ni, fm);
    if (rc != PT_OK) return rc;
    memset(out, 0, sizeof(*out));
    const int ntri = (int)(fm.hot.size() / 3);
    std::vector<int> refs(ntri, 0);
    auto halfArea = [](const float* lo, const float* hi) { float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2]; return dx * dy + dy * dz + dz * dx; };
    /* walk the 4-wide hierarchy; returns the box of everything below `ref` measured on the triangles themselves */
    struct Box { float lo[3], hi[3]; };
    const float rootArea = halfArea(nodes[0].bmin, nodes[0].bmax);
    double cost = 0.0;
    int maxDepth = 0;
    bool ok = true;
    std::function<Box(int, int)> walk = [&](int ref, int depth) -> Box {
        Box b; for (int k = 0; k < 3; ++k) { b.lo[k] = 3.4e38f; b.hi[k] = -3.4e38f; }
        if (ref < 0) {
            unsigned x = ~(unsigned)ref;
            int first = (int)(x >> 5), count = (int)(x & 31u) + 1;
            out->leaves++;
            for (int i = first; i < first + count && i < ntri; ++i) {
                refs[i]++;
                const float4* h = &fm.hot[3 * (size_t)i];
                float v0[3] = { h[0].x, h[0].y, h[0].z }, e1[3] = { h[0].w, h[1].x, h[1].y }, e2[3] = { h[1].z, h[1].w, h[2].x };
                for (int k = 0; k < 3; ++k) {
                    /* the vertices as the host BVH saw them: v0 and v0 + edge can differ from v1, v2 by one rounding */
                    float a = v0[k], c1 = v0[k] + e1[k], c2 = v0[k] + e2[k];
                    b.lo[k] = std::min(b.lo[k], std::min(a, std::min(c1, c2)));
                    b.hi[k] = std::max(b.hi[k], std::max(a, std::max(c1, c2)));
                }
            }
            return b;
        }
        maxDepth = std::max(maxDepth, depth + 1);
        out->wideNodes++;
        const float4* q = &fm.nodes[8 * (size_t)ref];
        for (int s = 0; s < 4; ++s) {
            int cr; float f = (&q[6].x)[s]; memcpy(&cr, &f, 4);
            if (cr == kEmptyRef) continue;
            float lo[3] = { (&q[0].x)[s], (&q[1].x)[s], (&q[2].x)[s] }, hi[3] = { (&q[3].x)[s], (&q[4].x)[s], (&q[5].x)[s] };
            if (cr >= 0) cost += halfArea(lo, hi) / rootArea;
            Box c = walk(cr, depth + 1);
            for (int k = 0; k < 3; ++k) {
                const float tol = 1e-5f * (std::fabs(c.lo[k]) + std::fabs(c.hi[k]) + 1.0f);
                if (c.lo[k] < lo[k] - tol || c.hi[k] > hi[k] + tol) ok = false;
                b.lo[k] = std::min(b.lo[k], lo[k]); b.hi[k] = std::max(b.hi[k], hi[k]);
            }
        }
        return b;
    };
    walk(0, 0);
    if (!ok) { pt_set_error("pt_bvh_device_layout: a child box does not hold the triangles below it"); return PT_EINVAL; }
    std::vector<char> rankSeen(ntri, 0);
    for (int i = 0; i < ntri; ++i) {
        if (fm.leafOfOrig.size() && refs[i] != 1) { pt_set_error("pt_bvh_device_layout: triangle slot %d is referenced %d times", i, refs[i]); return PT_EINVAL; }
        int rank; float f = fm.hot[3 * (size_t)i + 2].z; memcpy(&rank, &f, 4);
        if (rank < 0 || rank >= ntri || rankSeen[rank]) { pt_set_error("pt_bvh_device_layout: visit ranks are not a permutation"); return PT_EINVAL; }
        rankSeen[rank] = 1;
        out->trianglesReferenced += refs[i];
    }
    double hostCost = 0.0;
    for (int i = 0; i < nn; ++i)
        if (nodes[i].leftChild != -1 && i != 0) hostCost += halfArea(nodes[i].bmin, nodes[i].bmax) / rootArea;
    out->wideDepth = maxDepth;
    out->sahCostHost = (float)hostCost;
    out->sahCostDevice = (float)cost;
    return PT_OK;
}

int pt_destroy(PtContext* c)
{
    if (!c) return PT_OK;
    (void)hipSetDevice(c->device);
    /* kernels of a call that failed half-way may still run on a row's streams without having been joined to the main one */
    for (int g = 0; g < kMaxStreamGroups; ++g) {
        if (c->groupStreams[g]) (void)hipStreamSynchronize(c->groupStreams[g]);
        if (c->shadowStreams[g]) (void)hipStreamSynchronize(c->shadowStreams[g]);
    }
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->serveStream) (void)hipStreamSynchronize(c->serveStream);
    if (c->hostImage && c->hostRegistered) (void)hipHostUnregister(c->hostImage);
    if (c->hostStaging) (void)hipHostFree(c->hostStaging);
    if (c->serveStream) (void)hipStreamDestroy(c->serveStream);
    for (hipEvent_t e : c->hostPartDone) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : { c->servedEvent, c->sumReadyEvent, c->hostCopyDone, c->aheadCur.done, c->aheadNext.done })
        if (e) (void)hipEventDestroy(e);
    resolveSpans(c);
    for (hipEvent_t e : c->freeEvents) (void)hipEventDestroy(e);
    for (void* p : c->allocs) (void)hipFree(p);
    for (void* p : c->batchAllocs) (void)hipFree(p);
    for (int g = 0; g < kMaxStreamGroups; ++g) {
        if (c->groupStreams[g]) (void)hipStreamDestroy(c->groupStreams[g]);
        if (c->joinEvents[g]) (void)hipEventDestroy(c->joinEvents[g]);
        if (c->shadowStreams[g]) (void)hipStreamDestroy(c->shadowStreams[g]);
        if (c->shadedEvents[g]) (void)hipEventDestroy(c->shadedEvents[g]);
        if (c->shadowDoneEvents[g]) (void)hipEventDestroy(c->shadowDoneEvents[g]);
        if (c->slotFreeEvents[g]) (void)hipEventDestroy(c->slotFreeEvents[g]);
    }
    if (c->forkEvent) (void)hipEventDestroy(c->forkEvent);
    if (c->hintEvent) (void)hipEventDestroy(c->hintEvent);
    if (c->tuneStart) (void)hipEventDestroy(c->tuneStart);
    if (c->tuneEnd) (void)hipEventDestroy(c->tuneEnd);
    if (c->liveHintHost) (void)hipHostFree(c->liveHintHost);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return PT_OK;
}

int pt_create(const PtSceneDesc* d, int device, PtContext** out)
{
    if (!d || !out) { pt_set_error("pt_create: null argument"); return PT_EINVAL; }
    *out = nullptr;
    const int W = d->camera.resolution[0], H = d->camera.resolution[1];
    if (W <= 0 || H <= 0 || (long long)W * H > (1ll << 30)) { pt_set_error("pt_create: bad resolution %dx%d", W, H); return PT_EINVAL; }
    if (d->traceDepth < 1 || d->traceDepth > PT_MAX_DEPTH) { pt_set_error("pt_create: traceDepth must be in [1,%d]", PT_MAX_DEPTH); return PT_EINVAL; }
    if (d->geomCount < 0 || d->materialCount < 1) { pt_set_error("pt_create: scene needs at least one material"); return PT_EINVAL; }
    if (d->materialCount + 1 > kMaxBuckets) { pt_set_error("pt_create: at most %d materials are supported", kMaxBuckets - 1); return PT_ELIMIT; }
    for (int i = 0; i < d->geomCount; ++i) {
        const PtGeom& g = d->geoms[i];
        if (g.type == PT_CUBE || g.type == PT_SPHERE) {
            if (g.materialid < 0 || g.materialid >= d->materialCount) { pt_set_error("pt_create: geom %d has material id %d out of range", i, g.materialid); return PT_EINVAL; }
        } else if (g.type == PT_GLTF_MESH) {
            if (g.meshIndex >= d->meshCount) { pt_set_error("pt_create: geom %d references mesh %d of %d", i, g.meshIndex, d->meshCount); return PT_EINVAL; }
        } else { pt_set_error("pt_create: geom %d has unknown type %d", i, g.type); return PT_EINVAL; }
    }
    for (int m = 0; m < d->meshCount; ++m)
        for (int t = 0; t < d->meshes[m].triangleCount; ++t) {
            int id = d->meshes[m].triangles[t].materialId;
            if (id < 0 || id >= d->materialCount) { pt_set_error("pt_create: mesh %d triangle %d has material id %d out of range", m, t, id); return PT_EINVAL; }
        }

    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) { pt_set_error("pt_create: device %d of %d", device, ndev); return PT_EINVAL; }
    HIP_TRY(hipSetDevice(device));

    PtContext* c = new PtContext();
    c->device = device;
    c->cam = d->camera;
    c->traceDepth = d->traceDepth;
    c->numPixels = W * H;
    int rc = PT_OK;
#define TRY_RC(x) do { rc = (x); if (rc != PT_OK) { pt_destroy(c); return rc; } } while (0)
#define TRY_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { pt_set_error("%s failed: %s", #x, hipGetErrorString(e_)); pt_destroy(c); return PT_EHIP; } } while (0)
    TRY_HIP(hipGetSymbolAddress(&c->staging, HIP_SYMBOL(scratch)));       /* device `device` is current */
    TRY_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    TRY_HIP(hipEventCreateWithFlags(&c->forkEvent, hipEventDisableTiming));
    TRY_HIP(hipEventCreateWithFlags(&c->hintEvent, hipEventDisableTiming));
    TRY_HIP(hipEventCreate(&c->tuneStart));
    TRY_HIP(hipEventCreate(&c->tuneEnd));
    {
        /* Additions to the running sum and copies to the host run beside the rendering of the rows ahead, on a stream of
         * their own.  A high-priority one (PT_SERVE_PRIORITY=0: plain): the runtime gives it a hardware queue of its own,
         * where a plain stream may land on the queue of the rendering stream and then serves a call only after
         * everything queued for rendering (2.0 instead of 1.05 ms per call, depending on what else the process has
         * created).  Nothing is ever queued on it that has to wait on the device for the other streams — such a wait was
         * seen to stop the lower-priority queues it waited for, for good; the host waits instead (pt_trace_lookahead). */
        const char* pe = getenv("PT_SERVE_PRIORITY");
        if (!pe || atoi(pe) != 0) {
            int lo = 0, hi = 0;
            TRY_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
            TRY_HIP(hipStreamCreateWithPriority(&c->serveStream, hipStreamNonBlocking, hi));
        } else TRY_HIP(hipStreamCreateWithFlags(&c->serveStream, hipStreamNonBlocking));
    }
    TRY_HIP(hipEventCreateWithFlags(&c->servedEvent, hipEventDisableTiming));
    TRY_HIP(hipEventCreateWithFlags(&c->sumReadyEvent, hipEventDisableTiming));
    TRY_HIP(hipEventCreateWithFlags(&c->hostCopyDone, hipEventDisableTiming));
    for (hipEvent_t& e : c->hostPartDone) TRY_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    TRY_HIP(hipEventCreateWithFlags(&c->aheadCur.done, hipEventDisableTiming));
    TRY_HIP(hipEventCreateWithFlags(&c->aheadNext.done, hipEventDisableTiming));
    TRY_HIP(hipHostMalloc((void**)&c->liveHintHost, sizeof(uint32_t) * (PT_MAX_DEPTH + 2), hipHostMallocDefault));
    { const char* e = getenv("PT_SPARSE"); if (e) c->sparseMode = atoi(e); }
    for (int g = 0; g < kMaxStreamGroups; ++g) {
        TRY_HIP(hipStreamCreateWithFlags(&c->groupStreams[g], hipStreamNonBlocking));
        TRY_HIP(hipEventCreateWithFlags(&c->joinEvents[g], hipEventDisableTiming));
        TRY_HIP(hipStreamCreateWithFlags(&c->shadowStreams[g], hipStreamNonBlocking));
        TRY_HIP(hipEventCreateWithFlags(&c->shadedEvents[g], hipEventDisableTiming));
        TRY_HIP(hipEventCreateWithFlags(&c->shadowDoneEvents[g], hipEventDisableTiming));
        TRY_HIP(hipEventCreateWithFlags(&c->slotFreeEvents[g], hipEventDisableTiming));
    }
    {
        /* calls in flight follow the hardware queues the process has: HIP gives it four unless GPU_MAX_HW_QUEUES says
         * otherwise (read by the runtime when it starts), and streams beyond the queues serialise.  With 16 queues
         * eight calls overlap (dragon workload: 2.17 -> 1.77 ms per call); PT_PIPELINE fixes the depth. */
        const char* q = getenv("GPU_MAX_HW_QUEUES");
        if (q && atoi(q) >= 8) c->pipeDepth = atoi(q) >= 16 ? 8 : 6;
        const char* e = getenv("PT_PIPELINE");
        if (e) c->pipeDepth = std::max(1, std::min(atoi(e), kMaxStreamGroups));
    }
    { const char* e = getenv("PT_SHADOW_OVERLAP"); if (e) c->shadowOverlap = atoi(e) != 0; }
    { const char* e = getenv("PT_ROWS_PER_GROUP"); if (e && atoi(e) > 0) c->rowsPerGroup = atoi(e); }
    { const char* e = getenv("PT_STREAMS"); int v = e ? atoi(e) : kDefaultStreamGroups; c->streamGroups = std::max(1, std::min(v, kMaxStreamGroups)); c->streamsFixed = e != nullptr; }

    /* ---- meshes: concatenated in geom order (src/pathtrace.cu:483-505), one BVH per mesh (:269-334) */
    std::vector<DGeom> geoms(d->geomCount);
    std::vector<float4> allNodes, allHot, allCold, allBox;
    std::vector<int> allLeafOfOrig;
    std::vector<PtBvhNode> allHostNodes;        /* the host BVHs as built, for travReference */
    std::vector<int32_t> allHostIndices;
    for (int i = 0; i < d->geomCount; ++i) {
        const PtGeom& g = d->geoms[i];
        DGeom& G = geoms[i];
        memset(&G, 0, sizeof(G));
        G.type = g.type; G.materialid = g.materialid; G.meshIndex = g.meshIndex;
        memcpy(G.T, g.transform, 64); memcpy(G.I, g.inverseTransform, 64); memcpy(G.IT, g.invTranspose, 64);
        memcpy(G.translation, g.translation, 12); memcpy(G.scale, g.scale, 12);
        if (g.type == PT_GLTF_MESH && g.meshIndex >= 0) {
            const PtMesh& pm = d->meshes[g.meshIndex];
            FlatMesh fm;
            PtBvh* built = nullptr;
            const PtBvhNode* nodes = pm.nodes; int nn = pm.nodeCount;
            const int32_t* idx = pm.triangleIndices; int ni = pm.triangleIndexCount;
            if ((!nodes || !idx) && pm.triangleCount > 0) {
                TRY_RC(pt_bvh_build(pm.triangles, pm.triangleCount, 24 /* BVH_MAX_TREE_DEPTH, src/pathtrace.cu:194 */, &built));
                nodes = pt_bvh_nodes(built, &nn);
                idx = pt_bvh_indices(built, &ni);
            }
            PtBvhNode root{};
            if (nodes && nn > 0) root = nodes[0];          /* copied: a BVH built here is freed right after flattening */
            rc = flattenMesh(pm.triangles, pm.triangleCount, nodes, nn, idx, ni, fm);
            if (rc == PT_OK && fm.hasBvh) {
                G.hostNodeStart = (int)allHostNodes.size();
                G.hostIndexStart = (int)allHostIndices.size();
                allHostNodes.insert(allHostNodes.end(), nodes, nodes + nn);
                allHostIndices.insert(allHostIndices.end(), idx, idx + ni);
            }
            if (built) pt_bvh_free(built);
            if (rc != PT_OK) { pt_destroy(c); return rc; }
            G.nodeStart = (int)(allNodes.size() / 8);
            G.triStart = (int)(allHot.size() / 3);
            G.triCount = pm.triangleCount;
            G.hasBvh = fm.hasBvh ? 1 : 0;
            if (fm.hasBvh) {
                c->bvhMeshGeoms++;
                /* padded world box of the transformed root box */
                float lo[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, hi[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
                for (int corner = 0; corner < 8; ++corner) {
                    float p[3] = { (corner & 1) ? root.bmax[0] : root.bmin[0], (corner & 2) ? root.bmax[1] : root.bmin[1],
                                   (corner & 4) ? root.bmax[2] : root.bmin[2] };
                    for (int k = 0; k < 3; ++k) {        /* column-major 4x4, w = 1 */
                        float w = g.transform[k] * p[0] + g.transform[4 + k] * p[1] + g.transform[8 + k] * p[2] + g.transform[12 + k];
                        lo[k] = std::min(lo[k], w); hi[k] = std::max(hi[k], w);
                    }
                }
                float ext = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
                float mag = 0.0f;
                for (int k = 0; k < 3; ++k) mag = std::max(mag, std::max(std::fabs(lo[k]), std::fabs(hi[k])));
                const float pad = 0.01f * ext + 1e-5f * mag + 1e-6f;
                for (int k = 0; k < 3; ++k) { G.cullLo[k] = lo[k] - pad; G.cullHi[k] = hi[k] + pad; }
            }
            /* leafOfOrig is indexed by triStart + original index; keep both arrays aligned */
            size_t ntriFlat = fm.hot.size() / 3;
            allBox.insert(allBox.end(), fm.box.begin(), fm.box.end());
            allNodes.insert(allNodes.end(), fm.nodes.begin(), fm.nodes.end());
            allHot.insert(allHot.end(), fm.hot.begin(), fm.hot.end());
            allCold.insert(allCold.end(), fm.cold.begin(), fm.cold.end());
            fm.leafOfOrig.resize(ntriFlat, 0);
            allLeafOfOrig.insert(allLeafOfOrig.end(), fm.leafOfOrig.begin(), fm.leafOfOrig.end());
        }
    }
    float4 *dNodes, *dHot, *dCold, *dBox; int* dLeaf; DGeom* dGeoms; PtMaterial* dMats; PtBvhNode* dHostNodes; int32_t* dHostIdx;
    TRY_RC(devUpload(c, &dHostNodes, allHostNodes.data(), allHostNodes.size()));
    TRY_RC(devUpload(c, &dHostIdx, allHostIndices.data(), allHostIndices.size()));
    TRY_RC(devUpload(c, &dNodes, allNodes.data(), allNodes.size()));
    TRY_RC(devUpload(c, &dHot, allHot.data(), allHot.size()));
    TRY_RC(devUpload(c, &dCold, allCold.data(), allCold.size()));
    TRY_RC(devUpload(c, &dBox, allBox.data(), allBox.size()));
    TRY_RC(devUpload(c, &dLeaf, allLeafOfOrig.data(), allLeafOfOrig.size()));
    TRY_RC(devUpload(c, &dGeoms, geoms.data(), geoms.size()));
    TRY_RC(devUpload(c, &dMats, d->materials, (size_t)d->materialCount));

    /* ---- textures (src/pathtrace.cu:507-537) */
    std::vector<DTexture> tex(d->textureCount);
    for (int t = 0; t < d->textureCount; ++t) {
        const PtTexture& pt = d->textures[t];
        if (!pt.data || pt.width <= 0 || pt.height <= 0 || pt.components < 1 || pt.components > 4) { pt_set_error("pt_create: texture %d is invalid", t); pt_destroy(c); return PT_EINVAL; }
        /* device copy: four bytes per texel (grey replicated, alpha 255 where the image has none) in 8 x 4-texel tiles
         * of 128 bytes, edges padded to whole tiles; see sampleTexture */
        const int tilesX = (pt.width + kTexTileW - 1) / kTexTileW, tilesY = (pt.height + kTexTileH - 1) / kTexTileH;
        std::vector<uint32_t> tiled((size_t)tilesX * tilesY * kTexTileW * kTexTileH, 0u);
        if (pt.components != 2) {
            for (int y = 0; y < pt.height; ++y) {
                const uint8_t* row = pt.data + (size_t)y * pt.width * pt.components;
                uint32_t* out = tiled.data() + ((size_t)(y / kTexTileH) * tilesX) * 32 + (size_t)(y % kTexTileH) * kTexTileW;
                for (int x = 0; x < pt.width; ++x) {
                    const uint8_t* px = row + (size_t)x * pt.components;
                    uint32_t v;
                    if (pt.components == 1) v = px[0] * 0x010101u | 0xff000000u;
                    else if (pt.components == 3) v = px[0] | (uint32_t)px[1] << 8 | (uint32_t)px[2] << 16 | 0xff000000u;
                    else v = px[0] | (uint32_t)px[1] << 8 | (uint32_t)px[2] << 16 | (uint32_t)px[3] << 24;
                    out[(size_t)(x / kTexTileW) * 32 + (x % kTexTileW)] = v;
                }
            }
        }
        uint32_t* dd;
        TRY_RC(devUpload(c, &dd, tiled.data(), tiled.size()));
        tex[t].data = reinterpret_cast<const uint8_t*>(dd); tex[t].width = pt.width; tex[t].height = pt.height; tex[t].components = pt.components; tex[t].tilesPerRow = tilesX;
    }
    DTexture* dTex;
    TRY_RC(devUpload(c, &dTex, tex.data(), tex.size()));
    /* materials whose base-colour and metallic-roughness textures have one size get the two side by side, 8 bytes per texel in
     * 4 x 4-texel tiles of 128 bytes (sampleTexturePair); pairs that several materials share are made once */
    std::vector<DTexPair> pairs((size_t)d->materialCount);
    DTexPair* dPairs = nullptr;
    {
        auto texel = [&](const PtTexture& pt, int x, int y) -> uint32_t {
            const uint8_t* px = pt.data + ((size_t)y * pt.width + x) * pt.components;
            if (pt.components == 1) return px[0] * 0x010101u | 0xff000000u;
            if (pt.components == 3) return px[0] | (uint32_t)px[1] << 8 | (uint32_t)px[2] << 16 | 0xff000000u;
            return px[0] | (uint32_t)px[1] << 8 | (uint32_t)px[2] << 16 | (uint32_t)px[3] << 24;
        };
        std::vector<std::pair<std::pair<int, int>, DTexPair>> made;
        bool any = false;
        for (int m = 0; m < d->materialCount; ++m) {
            DTexPair& P = pairs[m];
            P.data = nullptr; P.width = P.height = P.tilesPerRow = P.pad = 0;
            const int b = d->materials[m].baseColorTextureIdx, r = d->materials[m].metallicRoughnessTextureIdx;
            if (b < 0 || r < 0 || b >= d->textureCount || r >= d->textureCount) continue;
            const PtTexture& tb = d->textures[b];
            const PtTexture& tr = d->textures[r];
            if (tb.width != tr.width || tb.height != tr.height || tb.components == 2 || tr.components == 2) continue;
            bool found = false;
            for (const auto& e : made) if (e.first == std::make_pair(b, r)) { P = e.second; found = true; break; }
            if (!found) {
                const int tilesX = (tb.width + 3) / 4, tilesY = (tb.height + 3) / 4;
                std::vector<uint2> tiled((size_t)tilesX * tilesY * 16, make_uint2(0u, 0u));
                for (int y = 0; y < tb.height; ++y)
                    for (int x = 0; x < tb.width; ++x)
                        tiled[((size_t)(y / 4) * tilesX + x / 4) * 16 + (size_t)(y % 4) * 4 + x % 4] = make_uint2(texel(tb, x, y), texel(tr, x, y));
                uint2* dd;
                TRY_RC(devUpload(c, &dd, tiled.data(), tiled.size()));
                P.data = dd; P.width = tb.width; P.height = tb.height; P.tilesPerRow = tilesX;
                made.push_back({ { b, r }, P });
            }
            any = true;
        }
        if (any) TRY_RC(devUpload(c, &dPairs, pairs.data(), pairs.size()));
    }

    /* ---- lights (initializeLights, src/pathtrace.cu:232-267) */
    std::vector<DLight> lights;
    float totalArea = 0.0f;
    for (int i = 0; i < d->geomCount; ++i) {
        int matId = d->geoms[i].materialid;
        if (matId >= 0 && matId < d->materialCount && d->materials[matId].emittance > 0.0f) {
            DLight L; L.geomIdx = i; L.area = geomArea(d->geoms[i]); L.pdf = 0.0f;
            totalArea += L.area;
            lights.push_back(L);
        }
    }
    for (auto& L : lights) L.pdf = L.area / totalArea;
    DLight* dLights;
    TRY_RC(devUpload(c, &dLights, lights.data(), lights.size()));

    /* ---- environment map + CDFs (src/pathtrace.cu:453-476) */
    const bool envOn = d->env.enabled && d->env.data && d->env.width > 0 && d->env.height > 0;
    float4* dEnv = nullptr; float* dMarg = nullptr; float* dCond = nullptr; uint16_t* dGuide = nullptr; float totalPower = 0.0f;
    if (envOn) {
        std::vector<float> marg, cond;
        buildEnvCdfs(d->env, marg, cond, totalPower);
        {   /* one 16-byte load per texel instead of three 4-byte ones */
            std::vector<float4> texels((size_t)d->env.width * d->env.height);
            for (size_t i = 0; i < texels.size(); ++i) texels[i] = make_float4(d->env.data[3 * i], d->env.data[3 * i + 1], d->env.data[3 * i + 2], 0.0f);
            TRY_RC(devUpload(c, &dEnv, texels.data(), texels.size()));
        }
        TRY_RC(devUpload(c, &dMarg, marg.data(), marg.size()));
        TRY_RC(devUpload(c, &dCond, cond.data(), cond.size()));
        /* guide tables of the CDF searches (lowerBoundGuided): the answers for the multiples of 1/kCdfGuide, for the marginal CDF and
         * for every row of the conditional one; only when no CDF starts below zero, ever decreases or holds a NaN, and indices fit 16 bits */
        const int W = d->env.width, H = d->env.height;
        auto monotone = [](const float* f, int n) { if (!(f[0] >= 0.0f)) return false; for (int i = 1; i < n; ++i) if (!(f[i] >= f[i - 1])) return false; return true; };
        bool guided = W <= 65536 && H <= 65536 && monotone(marg.data(), H);
        for (int y = 0; y < H && guided; ++y) guided = monotone(cond.data() + (size_t)y * W, W);
        if (guided) {
            std::vector<uint16_t> guide((size_t)(H + 1) * kCdfGuide);
            auto fill = [](const float* f, int n, uint16_t* g) {
                int at = 0;
                for (int j = 0; j < kCdfGuide; ++j) {
                    const float v = (float)j / (float)kCdfGuide;
                    while (at < n - 1 && f[at] < v) ++at;        /* first entry >= v, or n - 1 */
                    g[j] = (uint16_t)at;
                }
            };
            fill(marg.data(), H, guide.data());
            for (int y = 0; y < H; ++y) fill(cond.data() + (size_t)y * W, W, guide.data() + (size_t)(y + 1) * kCdfGuide);
            TRY_RC(devUpload(c, &dGuide, guide.data(), guide.size()));
        }
    }

    SceneView& V = c->view;
    V.cameraOrigin = 0; V.leanHits = 0; V.traceDepth = c->traceDepth; memcpy(V.camPos, c->cam.position, sizeof(V.camPos));     /* set per call in traceBatch */
    V.geoms = dGeoms; V.materials = dMats; V.nodes = dNodes; V.triHot = dHot; V.triCold = dCold; V.triBox = dBox; V.leafOfOrig = dLeaf; V.hostNodes = dHostNodes; V.hostIndices = dHostIdx;
    V.textures = dTex; V.texPairs = dPairs; V.lights = dLights; V.envData = dEnv; V.envMarginal = dMarg; V.envConditional = dCond; V.envGuide = dGuide;
    V.numGeoms = d->geomCount; V.numMaterials = d->materialCount; V.numTextures = d->textureCount; V.numLights = (int)lights.size();
    V.envWidth = envOn ? d->env.width : 0; V.envHeight = envOn ? d->env.height : 0; V.envEnabled = envOn ? 1 : 0; V.useBvh = 1;
    V.envTotalPower = totalPower;
    V.numBuckets = d->materialCount + 1;
    V.numBvhMeshes = c->bvhMeshGeoms;
    V.packGeom = d->geomCount <= 32 ? 1 : 0;
    /* nodes the traversal workgroups copy to LDS: the head of the node array, which is the breadth-first numbered top of
     * the first mesh's tree (any node below that index is read from the copy, whichever mesh it belongs to) */
    V.topCount = std::min<int>(kTopNodes, (int)(allNodes.size() / 8));

    /* ---- running sum, path state and queues */
    const size_t N = (size_t)c->numPixels;
    TRY_RC(devAlloc(c, &c->imageOwned, N * 3));
    c->imageSum = c->imageOwned;
    TRY_HIP(hipMemset(c->imageSum, 0, N * 3 * sizeof(float)));
    {
        const char* e = getenv("PT_BATCH");
        int batch = e ? atoi(e) : kDefaultBatch;
        batch = std::max(1, std::min(batch, kMaxBatch));
        TRY_RC(allocBatchState(c, batch));
    }
    TRY_RC(devAlloc(c, &c->workCtr, (size_t)kWorkCtrCount));
    TRY_RC(devAlloc(c, &c->fetchCtr, kFetchCtrCount));
    TRY_HIP(hipMemset(c->fetchCtr, 0, kFetchCtrCount * sizeof(uint32_t)));
    TRY_RC(devAlloc(c, &c->dc, 1));
    TRY_HIP(hipMemset(c->dc, 0, sizeof(DevCounters)));
    TRY_HIP(hipDeviceSynchronize());
#undef TRY_RC
#undef TRY_HIP
    *out = c;
    return PT_OK;
}

static int traceBatchLaunch(PtContext* c, Batch bt, int rowBase, bool deferSum)
{
    if (c->batch < 1 || bt.count < 1 || rowBase < 0 || rowBase + bt.count > c->batch) { pt_set_error("pt_trace: no path state for a group of %d iterations (allocated: %d)", bt.count, c->batch); return PT_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    SceneView V = c->view;
    V.useBvh = c->useBvh;
    V.numBvhMeshes = c->useBvh ? c->bvhMeshGeoms : 0;
    V.cameraOrigin = 1;
    V.leanHits = 1;
    V.traceDepth = c->traceDepth;
    memcpy(V.camPos, c->cam.position, sizeof(V.camPos));
    const bool count = c->counters != 0;
    uint32_t* ctr = c->workCtr;
    const bool meshes = V.numBvhMeshes > 0;
    /* a one-iteration call takes the next of `pipe` path-state rows and that row's streams; see PtContext::pipeDepth */
    const int pipe = std::min(std::min(c->pipeDepth, c->batch), kMaxStreamGroups);
    const bool piped = bt.count == 1 && pipe > 1 && !c->timing && meshes && !deferSum && rowBase == 0;
    const int slot = piped ? (int)(c->pipeNext++ % (unsigned)pipe) : 0;
    /* Is the caller issuing calls ahead of the device?  Then up to `pipe` iterations are in flight: each traversal launch
     * takes 1/pipe of the machine's workgroup slots so that they run side by side (as the rows of a launch group do), and
     * the shadow traversal stays on the row's one stream (more streams than hardware queues get in each other's way).
     * A caller that waits after every call finds the previous row idle and gets the whole machine per launch. */
    bool ahead = false;
    if (piped && c->pipeLast >= 0 && c->slotBusy[c->pipeLast]) ahead = hipEventQuery(c->slotFreeEvents[c->pipeLast]) == hipErrorNotReady;
    static const int shareEnv = [] { const char* e = getenv("PT_PIPE_SHARE"); return e ? atoi(e) : 0; }();
    const int share = !piped ? 1 : shareEnv > 0 ? shareEnv : ahead ? pipe : 1;
    if (piped) c->pipeLast = slot;
    if (piped) {
        if (c->pipeBarrier) {
            /* other work on the main stream has used the path state or the counters since: every row waits for it once */
            HIP_TRY(hipEventRecord(c->forkEvent, s));
            for (int g = 0; g < pipe; ++g) HIP_TRY(hipStreamWaitEvent(c->groupStreams[g], c->forkEvent, 0));
            c->pipeBarrier = false;
        }
        /* the row is free again once the accumulation of the call that used it last has read it */
        if (c->slotBusy[slot]) HIP_TRY(hipStreamWaitEvent(c->groupStreams[slot], c->slotFreeEvents[slot], 0));
        /* this row's counter lines only (one line per section and bounce, kMaxBatch lines apart) */
        HIP_TRY(hipMemset2DAsync(ctr + (size_t)slot * kCtrStride, (size_t)kMaxBatch * kCtrStride * sizeof(uint32_t), 0, kCtrStride * sizeof(uint32_t),
                                 (size_t)kWorkCtrCount / ((size_t)kMaxBatch * kCtrStride), c->groupStreams[slot]));
        for (int kind = 0; kind < 2; ++kind)
            HIP_TRY(hipMemset2DAsync(c->fetchCtr + kind * kFetchKind + (size_t)slot * kCtrStride, (size_t)kSegPitch * sizeof(uint32_t), 0, kCtrStride * sizeof(uint32_t),
                                     (size_t)c->traceDepth * kFetchSegments, c->groupStreams[slot]));
    } else {
        /* the main stream has been made to wait for every pipelined call when that call was issued */
        HIP_TRY(hipMemsetAsync(c->workCtr, 0, kWorkCtrCount * sizeof(uint32_t), s));
        for (int kind = 0; kind < 2; ++kind)
            HIP_TRY(hipMemsetAsync(c->fetchCtr + kind * kFetchKind, 0, (size_t)c->traceDepth * kFetchBouncePitch * sizeof(uint32_t), s));
        c->pipeBarrier = true;
    }
    uint32_t* closestFetch = c->fetchCtr;
    uint32_t* shadowFetch = c->fetchCtr + kFetchKind;
    uint32_t* closestRef = ctr;
    uint32_t* shadowRef = ctr + kWorkCtrSection;
    uint32_t* shadowCount = ctr + 2 * kWorkCtrSection;
    uint32_t* closestCount = ctr + 3 * kWorkCtrSection;
    if (c->hintPending && hipEventQuery(c->hintEvent) == hipSuccess) {
        memcpy(c->liveHint, c->liveHintHost, sizeof(c->liveHint));
        c->hintPending = false;
        c->hintValid = true;
    }
    const uint32_t sparseBelow = (uint32_t)c->numPixels / 16u;
    auto sparseBounce = [&](int b) { return c->sparseMode >= 0 ? c->sparseMode != 0 : (c->hintValid && c->liveHint[b] <= sparseBelow); };
    /* The rows of a launch group are independent until the final accumulation, so they can be spread over
     * several streams: while one group's traversal kernel drains its last long rays, the kernels of the
     * other groups fill the machine.  With stage timing on everything stays on one stream so that the
     * event-bracketed durations are those of kernels running alone. */
    int groups = 1;                     /* stage timing, or no traversal and so no tails to hide: one stream */
    int shape = 1;                      /* 1..3: that many streams; kTuneShapes: one stream, shadow traversal in line */
    bool tuning = false;
    if (!c->timing && meshes) {
        if (c->streamsFixed) groups = std::max(1, std::min(c->streamGroups, bt.count / c->rowsPerGroup));
        else if (bt.count >= kTuneMinRows && (count || bt.count != c->batch)) {
            /* counted kernels are no measure, and only full groups (PT_OPT_BATCH iterations) take part in the
             * choice: the remainder group of a range keeps the shape chosen so far and never restarts the trial */
            if (c->tuneStage > kTuneShapes) shape = c->tuneChoice;
        } else if (bt.count >= kTuneMinRows) {
            /* stages: 0 = a first group on one stream, unmeasured (first launches, no live-count hint yet);
             * 1..kTuneShapes = measure that shape; then the choice stands until the group size is set anew */
            if (c->tuneStage == 0) shape = 1;
            else if (c->tuneStage <= kTuneShapes) { shape = c->tuneStage; tuning = true; }
            else shape = c->tuneChoice;
        }
        if (!c->streamsFixed) groups = shape == kTuneShapes ? 1 : shape;
    }
    const bool inlineShadow = shape == kTuneShapes;
    if (tuning) HIP_TRY(hipEventRecord(c->tuneStart, s));
    if (groups > 1) HIP_TRY(hipEventRecord(c->forkEvent, s));
    for (int g = 0; g < groups; ++g) {
        const int r0 = piped ? slot : (int)((long long)bt.count * g / groups), r1 = piped ? slot + 1 : (int)((long long)bt.count * (g + 1) / groups);
        const int sx = piped ? slot : g;        /* which streams and events */
        const int rowAt = rowBase + r0;         /* first path-state row (and counter line) of this part */
        hipStream_t gs = (piped || groups > 1) ? c->groupStreams[sx] : s;
        if (groups > 1) HIP_TRY(hipStreamWaitEvent(gs, c->forkEvent, 0));
        Batch gb; gb.count = r1 - r0; gb.iterBase = bt.iterBase + (piped ? 0 : r0 * bt.iterStride); gb.iterStride = bt.iterStride;
        PathState st = c->state.at(rowAt);
        Queues q = c->queues.at(rowAt);
        const size_t rowOff = (size_t)rowAt * kCtrStride;
        { SpanGuard sg(c, ST_CAMERA, gs); ptl_camera(gs, V, c->cam, gb, c->traceDepth, st, q, closestCount + rowOff); }
        /* The shadow traversal of a bounce only settles colours and the image; compaction, the next closest-hit
         * stage and the sort do not look at either.  It therefore runs on a side stream next to them and is
         * joined before the next shade stage (or the final gather) reads colours again: one more kernel whose
         * tail of long rays is covered by other work. */
        const bool side = meshes && !c->timing && c->shadowOverlap && groups == 1 && !inlineShadow && share == 1;     /* more streams than three get in each other's way */
        hipStream_t ss = side ? c->shadowStreams[sx] : gs;
        bool shadowOpen = false;
        for (int b = 0; b < c->traceDepth; ++b) {
            const size_t row = (size_t)b * kMaxBatch * kCtrStride + rowOff;
            {
                SpanGuard sg(c, ST_INTERSECT, gs);
                if (meshes) {
                    SpanGuard t(c, ST_TRACE, gs);
                    ptl_trace(gs, V, st, q, gb, b, false, count, closestFetch + (size_t)b * kFetchBouncePitch + rowOff, closestCount + row, closestRef + row, c->dc, share);
                }
                ptl_prims(gs, V, st, q, gb, b, c->sortOn, count, c->dc, sparseBounce(b));
            }
            if (c->sortOn) { SpanGuard sg(c, ST_SORT, gs); ptl_sort(gs, q, gb, b, V.numBuckets); }
            {
                if (shadowOpen) { HIP_TRY(hipStreamWaitEvent(gs, c->shadowDoneEvents[sx], 0)); shadowOpen = false; }
                SpanGuard sg(c, ST_SHADE, gs);
                ptl_shade(gs, V, st, q, gb, b, c->sortOn, count, shadowCount + row,
                          (meshes && b + 1 < c->traceDepth) ? closestCount + row + (size_t)kMaxBatch * kCtrStride : nullptr, c->dc, sparseBounce(b));
                if (meshes) {
                    if (side) { HIP_TRY(hipEventRecord(c->shadedEvents[sx], gs)); HIP_TRY(hipStreamWaitEvent(ss, c->shadedEvents[sx], 0)); }
                    SpanGuard t(c, ST_SHADOW_TRACE, ss);
                    ptl_trace(ss, V, st, q, gb, b, true, count, shadowFetch + (size_t)b * kFetchBouncePitch + rowOff, shadowCount + row, shadowRef + row, c->dc, share);
                    if (side) { HIP_TRY(hipEventRecord(c->shadowDoneEvents[sx], ss)); shadowOpen = true; }
                }
            }
            { SpanGuard sg(c, ST_COMPACT, gs); ptl_compact(gs, q, gb, b, c->sortOn); }
        }
        if (shadowOpen) HIP_TRY(hipStreamWaitEvent(gs, c->shadowDoneEvents[sx], 0));
        if (piped || groups > 1) {
            HIP_TRY(hipEventRecord(c->joinEvents[sx], gs));
            HIP_TRY(hipStreamWaitEvent(s, c->joinEvents[sx], 0));
        }
    }
    const int finishRow = rowBase + slot;
    ptl_finish(s, c->state.at(finishRow), c->queues.at(finishRow), bt, c->traceDepth, deferSum ? nullptr : c->imageSum, c->dc,
               shadowCount + (size_t)finishRow * kCtrStride, closestCount + (size_t)finishRow * kCtrStride, kMaxBatch);
    if (!deferSum) c->sumTouchedOnMain = true;
    if (piped) { HIP_TRY(hipEventRecord(c->slotFreeEvents[slot], s)); c->slotBusy[slot] = true; }
    if (tuning) {
        HIP_TRY(hipEventRecord(c->tuneEnd, s));
        HIP_TRY(hipEventSynchronize(c->tuneEnd));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, c->tuneStart, c->tuneEnd));
        c->tuneMs[shape] = ms;
        if (shape == kTuneShapes) {
            /* the faster of the two one-stream shapes, and more streams only when they pay: the fewest streams within
             * 3 % of the fastest shape (a shape's time varies by 1-2 % from group to group, more so with several streams) */
            float best = c->tuneMs[1];
            for (int g = 2; g <= kTuneShapes; ++g) best = std::min(best, c->tuneMs[g]);
            const int order[3] = { c->tuneMs[kTuneShapes] < c->tuneMs[1] ? kTuneShapes : 1, 2, 3 };
            c->tuneChoice = order[2];
            for (int k = 0; k < 3; ++k) if (c->tuneMs[order[k]] <= 1.03f * best) { c->tuneChoice = order[k]; break; }
            if (getenv("PT_DEBUG_TAIL")) fprintf(stderr, "[pt] launch shape for groups of %d: %d (1-3 streams: %.2f / %.2f / %.2f ms, one stream with the shadow traversal in line: %.2f ms)\n", bt.count, c->tuneChoice, c->tuneMs[1], c->tuneMs[2], c->tuneMs[3], c->tuneMs[4]);
        }
    }
    const bool tunePhase = !c->timing && meshes && !c->streamsFixed && !count && bt.count >= kTuneMinRows && bt.count == c->batch && c->tuneStage <= kTuneShapes;
    if (tunePhase) c->tuneStage++;
    if (!c->hintPending) {
        HIP_TRY(hipMemcpyAsync(c->liveHintHost, (const char*)c->dc + offsetof(DevCounters, liveMax), sizeof(c->liveHint), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(c->hintEvent, s));
        c->hintPending = true;
    }
    HIP_TRY(hipGetLastError());
    /* the unmeasured first group ends with a wait as well, so that the measured ones start with the live-count hint in place */
    if (tunePhase && !tuning) HIP_TRY(hipStreamSynchronize(s));
    c->iterations += (uint64_t)bt.count;
    return PT_OK;
}

/* A launch sequence that failed half-way leaves kernels on a row's streams that were never joined to the main stream,
 * and the bookkeeping of the pipelined rows half advanced: wait for everything the context has in flight and start the
 * rows afresh, so that a later pt_sync / pt_destroy / allocation cannot overtake work that still uses the path state. */
static int traceBatch(PtContext* c, Batch bt, int rowBase = 0, bool deferSum = false)
{
    if (!deferSum) { c->aheadCur.valid = false; c->aheadNext.valid = false; }      /* rows rendered ahead are about to be overwritten */
    const int rc = traceBatchLaunch(c, bt, rowBase, deferSum);
    if (rc != PT_OK && c->stream) {
        for (int g = 0; g < kMaxStreamGroups; ++g) {
            if (c->groupStreams[g]) (void)hipStreamSynchronize(c->groupStreams[g]);
            if (c->shadowStreams[g]) (void)hipStreamSynchronize(c->shadowStreams[g]);
            c->slotBusy[g] = false;
        }
        (void)hipStreamSynchronize(c->stream);
        (void)hipGetLastError();
        c->pipeLast = -1;
        c->pipeBarrier = true;
        c->aheadCur.valid = false; c->aheadNext.valid = false;
    }
    return rc;
}

int pt_trace(PtContext* c, int iter)
{
    if (!c) { pt_set_error("pt_trace: null context"); return PT_EINVAL; }
    Batch bt; bt.count = 1; bt.iterBase = iter; bt.iterStride = 1;
    return traceBatch(c, bt);
}

int pt_trace_range(PtContext* c, int iter_begin, int iter_end, int iter_stride)
{
    if (!c || iter_stride <= 0) { pt_set_error("pt_trace_range: bad argument"); return PT_EINVAL; }
    /* iterations are independent given their index: up to `batch` of them share every launch */
    if (c->batch < 1) { pt_set_error("pt_trace_range: no path state is allocated"); return PT_EINVAL; }
    for (int it = iter_begin; it < iter_end;) {
        int remaining = (iter_end - it + iter_stride - 1) / iter_stride;
        Batch bt; bt.count = std::min(remaining, c->batch); bt.iterBase = it; bt.iterStride = iter_stride;
        int rc = traceBatch(c, bt);
        if (rc != PT_OK) return rc;
        it += bt.count * iter_stride;
    }
    return PT_OK;
}

int pt_sync(PtContext* c)
{
    if (!c) { pt_set_error("pt_sync: null context"); return PT_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->serveStream));       /* the last addition to the sum and the host copy behind it */
    c->hostCopyPending = false;
    resolveSpans(c);
    return PT_OK;
}

int pt_display(PtContext* c, int iter, uint8_t* dev_rgba)
{
    if (!c || !dev_rgba || iter <= 0) { pt_set_error("pt_display: bad argument"); return PT_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    /* while iterations are rendered ahead the sum is kept on the serve stream; the main stream is busy with the rows ahead */
    ptl_display(c->aheadCur.valid ? c->serveStream : c->stream, c->imageSum, c->numPixels, iter, dev_rgba);
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

/* the serve stream's next addition to the sum comes after everything the main stream has done to it so far */
static int serveAfterMain(PtContext* c)
{
    if (!c->sumTouchedOnMain) return PT_OK;
    HIP_TRY(hipEventRecord(c->sumReadyEvent, c->stream));
    HIP_TRY(hipEventSynchronize(c->sumReadyEvent));         /* on the host: see the note at the serve stream's creation */
    c->sumTouchedOnMain = false;
    return PT_OK;
}

/* byte offset at which piece k of a staged host copy starts (pieces of whole 4 KB pages) */
static size_t hostPartStart(size_t bytes, int k)
{
    const size_t pages = (bytes + 4095) / 4096;
    return std::min(bytes, pages * (size_t)k / PtContext::kHostCopyParts * 4096);
}

/* copy of the running sum into the bound host image, behind whatever the stream holds */
static int queueHostCopy(PtContext* c, hipStream_t st)
{
    if (!c->hostImage) return PT_OK;
    const size_t bytes = (size_t)c->numPixels * 3 * sizeof(float);
    if (c->hostRegistered) HIP_TRY(hipMemcpyAsync(c->hostImage, c->imageSum, bytes, hipMemcpyDeviceToHost, st));
    else {
        for (int k = 0; k < PtContext::kHostCopyParts; ++k) {
            const size_t a = hostPartStart(bytes, k), b = hostPartStart(bytes, k + 1);
            if (b > a) HIP_TRY(hipMemcpyAsync((char*)c->hostStaging + a, (const char*)c->imageSum + a, b - a, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipEventRecord(c->hostPartDone[k], st));
        }
    }
    HIP_TRY(hipEventRecord(c->hostCopyDone, st));
    c->hostCopyPending = true;
    return PT_OK;
}

int pt_bind_host_image(PtContext* c, float* host_rgb)
{
    if (!c) { pt_set_error("pt_bind_host_image: null context"); return PT_EINVAL; }
    int rc = pt_sync(c);
    if (rc != PT_OK) return rc;
    if (c->hostImage && c->hostRegistered) (void)hipHostUnregister(c->hostImage);
    c->hostImage = nullptr;
    c->hostRegistered = false;
    if (!host_rgb) return PT_OK;
    const size_t bytes = (size_t)c->numPixels * 3 * sizeof(float);
    if (hipHostRegister(host_rgb, bytes, hipHostRegisterDefault) == hipSuccess) c->hostRegistered = true;
    else {
        (void)hipGetLastError();
        if (!c->hostStaging) HIP_TRY(hipHostMalloc((void**)&c->hostStaging, bytes, hipHostMallocDefault));
    }
    c->hostImage = host_rgb;
    return PT_OK;
}

int pt_host_image_wait(PtContext* c)
{
    if (!c) { pt_set_error("pt_host_image_wait: null context"); return PT_EINVAL; }
    if (!c->hostImage || !c->hostCopyPending) return PT_OK;
    HIP_TRY(hipSetDevice(c->device));
    if (c->hostRegistered) HIP_TRY(hipEventSynchronize(c->hostCopyDone));
    else {
        const size_t bytes = (size_t)c->numPixels * 3 * sizeof(float);
        for (int k = 0; k < PtContext::kHostCopyParts; ++k) {
            const size_t a = hostPartStart(bytes, k), b = hostPartStart(bytes, k + 1);
            HIP_TRY(hipEventSynchronize(c->hostPartDone[k]));
            memcpy((char*)c->hostImage + a, (const char*)c->hostStaging + a, b - a);
        }
    }
    c->hostCopyPending = false;
    return PT_OK;
}

int pt_trace_lookahead(PtContext* c, int iter, int ahead, int iter_last)
{
    if (!c || ahead < 1) { pt_set_error("pt_trace_lookahead: bad argument"); return PT_EINVAL; }
    if (c->batch < 1) { pt_set_error("pt_trace_lookahead: no path state is allocated"); return PT_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    const int B = std::min(ahead, c->batch / 2);
    if (B < 2) {
        /* no room to render ahead: one iteration, as pt_trace does it, then the host copy */
        Batch bt; bt.count = 1; bt.iterBase = iter; bt.iterStride = 1;
        int rc = traceBatch(c, bt);
        if (rc != PT_OK) return rc;
        return queueHostCopy(c, c->stream);
    }
    auto launch = [&](PtContext::AheadGroup& g, int base, int rowBase) -> int {
        int count = B;
        if (iter_last > 0) count = std::min(count, iter_last - base + 1);
        g.valid = false;
        if (count < 1) return PT_OK;
        Batch bt; bt.count = count; bt.iterBase = base; bt.iterStride = 1;
        int rc = traceBatch(c, bt, rowBase, true);
        if (rc != PT_OK) return rc;
        HIP_TRY(hipEventRecord(g.done, c->stream));
        g.base = base; g.count = count; g.rowBase = rowBase; g.valid = true;
        return PT_OK;
    };
    /* before anything of this call reaches the main stream: the serve stream's additions come after what the main
     * stream has done to the sum up to here (a clear, an upload, iterations added by pt_trace) */
    {
        int rc0 = serveAfterMain(c);
        if (rc0 != PT_OK) return rc0;
    }
    PtContext::AheadGroup& cur = c->aheadCur;
    PtContext::AheadGroup& next = c->aheadNext;
    if (!(cur.valid && iter == cur.base + c->aheadServed && c->aheadServed < cur.count)) {
        if (next.valid && iter == next.base) {
            std::swap(cur, next);                       /* the group rendered meanwhile; its events travel with it */
            next.valid = false;
        } else {
            /* not the iteration that was expected: whatever was rendered ahead is dropped */
            next.valid = false;
            int rc = launch(cur, iter, 0);
            if (rc != PT_OK) return rc;
            if (!cur.valid) { pt_set_error("pt_trace_lookahead: iteration %d lies beyond the last one (%d)", iter, iter_last); return PT_EINVAL; }
        }
        c->aheadServed = 0;
    }
    /* keep the device rendering: the group after this one goes onto the other half of the rows, whose contributions
     * have all joined the sum (the main stream waited for each of them) */
    if (!next.valid) {
        int rc = launch(next, cur.base + cur.count, cur.rowBase == 0 ? c->batch / 2 : 0);       /* the halves are fixed, so `ahead` may differ from call to call */
        if (rc != PT_OK) return rc;
    }
    /* the first call served from a group waits for the group on the host (a call of the strict form would wait for it
     * anyway, behind its host copy; the queued form runs ahead of the device by at most the group being rendered) */
    if (c->aheadServed == 0) HIP_TRY(hipEventSynchronize(cur.done));
    ptl_accumulate(c->serveStream, c->state.at(cur.rowBase + c->aheadServed), 1, c->imageSum);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->servedEvent, c->serveStream));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->servedEvent, 0));
    c->aheadServed++;
    return queueHostCopy(c, c->serveStream);
}

int pt_download(PtContext* c, float* host_rgb)
{
    if (!c || !host_rgb) { pt_set_error("pt_download: null argument"); return PT_EINVAL; }
    int rc = pt_sync(c);
    if (rc != PT_OK) return rc;
    HIP_TRY(hipMemcpy(host_rgb, c->imageSum, (size_t)c->numPixels * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return PT_OK;
}

/* Sample split inside one process: contexts (one per GPU, or several on one GPU) that rendered disjoint iterations of the
 * same scene; their running sums are folded into contexts[root]'s, in index order.  A sum on another device crosses as
 * one peer copy (device to device over xGMI where the devices can reach each other, through the host otherwise) into a
 * buffer on the root's device; the addition runs there.  The other contexts' sums are left as they are. */
int pt_reduce_sum(PtContext** contexts, int count, int root)
{
    if (!contexts || count < 1 || root < 0 || root >= count || !contexts[root]) { pt_set_error("pt_reduce_sum: bad argument"); return PT_EINVAL; }
    PtContext* r = contexts[root];
    const size_t n3 = (size_t)r->numPixels * 3;
    for (int i = 0; i < count; ++i) {
        if (!contexts[i]) { pt_set_error("pt_reduce_sum: context %d is null", i); return PT_EINVAL; }
        if (contexts[i]->numPixels != r->numPixels) { pt_set_error("pt_reduce_sum: context %d renders %d pixels, the root %d", i, contexts[i]->numPixels, r->numPixels); return PT_EINVAL; }
        for (int j = 0; j < i; ++j) if (contexts[j] == contexts[i]) { pt_set_error("pt_reduce_sum: context %d is listed twice", i); return PT_EINVAL; }
    }
    /* every contribution is complete before it is read */
    for (int i = 0; i < count; ++i) { int rc = pt_sync(contexts[i]); if (rc != PT_OK) return rc; }
    HIP_TRY(hipSetDevice(r->device));
    float* staging = nullptr;           /* one image on the root's device, for sums that live elsewhere */
    int rc = PT_OK;
    for (int i = 0; i < count && rc == PT_OK; ++i) {
        if (i == root) continue;
        PtContext* c = contexts[i];
        const float* src = c->imageSum;
        if (c->device != r->device) {
            if (!staging && hipMalloc((void**)&staging, n3 * sizeof(float)) != hipSuccess) { pt_set_error("pt_reduce_sum: out of device memory"); rc = PT_ENOMEM; break; }
            hipError_t e = hipMemcpyPeerAsync(staging, r->device, c->imageSum, c->device, n3 * sizeof(float), r->stream);
            if (e != hipSuccess) { pt_set_error("pt_reduce_sum: copy from device %d to device %d failed: %s", c->device, r->device, hipGetErrorString(e)); rc = PT_EHIP; break; }
            src = staging;
        }
        ptl_add_image(r->stream, r->imageSum, src, n3);
        if (hipGetLastError() != hipSuccess) { pt_set_error("pt_reduce_sum: launch failed"); rc = PT_EHIP; }
    }
    r->sumTouchedOnMain = true;
    if (hipStreamSynchronize(r->stream) != hipSuccess && rc == PT_OK) { pt_set_error("pt_reduce_sum: %s", hipGetErrorString(hipGetLastError())); rc = PT_EHIP; }
    if (staging) (void)hipFree(staging);
    return rc;
}

int pt_bind_image(PtContext* c, float* dev_rgb)
{
    if (!c) { pt_set_error("pt_bind_image: null context"); return PT_EINVAL; }
    int rc = pt_sync(c);
    if (rc != PT_OK) return rc;
    c->imageSum = dev_rgb ? dev_rgb : c->imageOwned;
    c->ownImage = dev_rgb == nullptr;
    c->sumTouchedOnMain = true;
    return PT_OK;
}

int pt_upload_image(PtContext* c, const float* host_rgb)
{
    if (!c || !host_rgb) { pt_set_error("pt_upload_image: null argument"); return PT_EINVAL; }
    int rc = pt_sync(c);
    if (rc != PT_OK) return rc;
    HIP_TRY(hipMemcpy(c->imageSum, host_rgb, (size_t)c->numPixels * 3 * sizeof(float), hipMemcpyHostToDevice));
    c->sumTouchedOnMain = true;
    return PT_OK;
}

int pt_clear(PtContext* c)
{
    if (!c) { pt_set_error("pt_clear: null context"); return PT_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(c->imageSum, 0, (size_t)c->numPixels * 3 * sizeof(float), c->stream));
    c->sumTouchedOnMain = true;
    return PT_OK;
}

int pt_set_camera(PtContext* c, const PtCamera* cam)
{
    if (!c || !cam) { pt_set_error("pt_set_camera: null argument"); return PT_EINVAL; }
    if (cam->resolution[0] != c->cam.resolution[0] || cam->resolution[1] != c->cam.resolution[1]) {
        pt_set_error("pt_set_camera: resolution %d x %d, the context was created for %d x %d", cam->resolution[0], cam->resolution[1], c->cam.resolution[0], c->cam.resolution[1]);
        return PT_EINVAL;
    }
    /* launches take the camera by value, so work already queued keeps the old one; rows rendered ahead were made with it */
    c->aheadCur.valid = false; c->aheadNext.valid = false;
    c->cam = *cam;
    return PT_OK;
}

/* 64-bit FNV-1a over 8-byte words (the tail byte-wise), continued from h */
static uint64_t hashBytes(const void* p, size_t n, uint64_t h)
{
    const unsigned char* b = static_cast<const unsigned char*>(p);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, b + i, 8); h = (h ^ w) * 1099511628211ull; }
    for (; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}
/* an array of `count` records: all of it up to 32 MB, beyond that 65 536 records spread evenly plus the first and last */
static uint64_t hashArray(const void* p, size_t count, size_t stride, uint64_t h)
{
    h = hashBytes(&count, sizeof(count), h);
    if (!p || count == 0) return h;
    const unsigned char* b = static_cast<const unsigned char*>(p);
    if (count * stride <= (32u << 20)) return hashBytes(b, count * stride, h);
    const size_t samples = 65536;
    for (size_t k = 0; k < samples; ++k) h = hashBytes(b + (count - 1) * k / (samples - 1) * stride, stride, h);
    return h;
}

unsigned long long pt_scene_fingerprint(const PtSceneDesc* d)
{
    if (!d) return 0;
    uint64_t h = 1469598103934665603ull;
    h = hashArray(d->geoms, (size_t)d->geomCount, sizeof(PtGeom), h);
    h = hashArray(d->materials, (size_t)d->materialCount, sizeof(PtMaterial), h);
    h = hashBytes(&d->meshCount, sizeof(d->meshCount), h);
    for (int m = 0; m < d->meshCount; ++m) {
        const PtMesh& mesh = d->meshes[m];
        h = hashArray(mesh.triangles, (size_t)mesh.triangleCount, sizeof(PtTriangle), h);
        h = hashArray(mesh.nodes, (size_t)(mesh.nodes ? mesh.nodeCount : 0), sizeof(PtBvhNode), h);
        h = hashArray(mesh.triangleIndices, (size_t)(mesh.triangleIndices ? mesh.triangleIndexCount : 0), sizeof(int32_t), h);
    }
    h = hashBytes(&d->textureCount, sizeof(d->textureCount), h);
    for (int t = 0; t < d->textureCount; ++t) {
        const PtTexture& tx = d->textures[t];
        const int32_t dims[3] = { tx.width, tx.height, tx.components };
        h = hashBytes(dims, sizeof(dims), h);
        h = hashArray(tx.data, tx.data && tx.width > 0 && tx.height > 0 ? (size_t)tx.width * tx.height : 0, (size_t)(tx.components > 0 ? tx.components : 1), h);
    }
    const int32_t env[3] = { d->env.enabled && d->env.data ? 1 : 0, d->env.width, d->env.height };
    h = hashBytes(env, sizeof(env), h);
    if (env[0]) h = hashArray(d->env.data, (size_t)d->env.width * d->env.height, 3 * sizeof(float), h);
    h = hashBytes(d->camera.resolution, sizeof(d->camera.resolution), h);
    h = hashBytes(&d->traceDepth, sizeof(d->traceDepth), h);
    return h ? h : 1;
}

int pt_set_option(PtContext* c, int option, int value)
{
    if (!c) { pt_set_error("pt_set_option: null context"); return PT_EINVAL; }
    c->aheadCur.valid = false; c->aheadNext.valid = false;      /* rows rendered ahead were made under the old settings */
    switch (option) {
    case PT_OPT_MATERIAL_SORT: c->sortOn = value ? 1 : 0; break;
    case PT_OPT_USE_BVH: c->useBvh = value ? 1 : 0; break;
    case PT_OPT_STAGE_TIMING: { int rc = pt_sync(c); if (rc != PT_OK) return rc; c->timing = value ? 1 : 0; break; }
    case PT_OPT_COUNTERS: c->counters = value ? 1 : 0; break;
    case PT_OPT_USE_GRAPH:
        /* launches are direct: one iteration is about 67 launches that run back to back on the device (gaps of about
         * 5 us in the kernel trace, under 0.3 ms of a 6 ms iteration at 1920x1080), so a replayed graph has nothing to
         * win; asking for one is refused rather than silently ignored */
        if (value) { pt_set_error("pt_set_option: PT_OPT_USE_GRAPH is not supported (launches are direct)"); return PT_EINVAL; }
        break;
    case PT_OPT_BATCH: {
        if (value < 1 || value > kMaxBatch) { pt_set_error("pt_set_option: batch must be in [1,%d]", kMaxBatch); return PT_ELIMIT; }
        int rc = pt_sync(c); if (rc != PT_OK) return rc;
        if (value != c->batch) { rc = allocBatchState(c, value); if (rc != PT_OK) return rc; }
        break;
    }
    case PT_OPT_LAUNCH_STREAMS:
        if (value < 0 || value > kMaxStreamGroups) { pt_set_error("pt_set_option: launch streams must be in [0,%d]", kMaxStreamGroups); return PT_ELIMIT; }
        c->streamsFixed = value > 0;
        c->streamGroups = value > 0 ? value : kDefaultStreamGroups;
        if (value == 0) { c->tuneStage = 0; c->tuneChoice = 1; }
        break;
    default: pt_set_error("pt_set_option: unknown option %d", option); return PT_EINVAL;
    }
    return PT_OK;
}

int pt_stats(PtContext* c, PtStats* out)
{
    if (!c || !out) { pt_set_error("pt_stats: null argument"); return PT_EINVAL; }
    int rc = pt_sync(c);
    if (rc != PT_OK) return rc;
    DevCounters h;
    HIP_TRY(hipMemcpy(&h, c->dc, sizeof(h), hipMemcpyDeviceToHost));
    memset(out, 0, sizeof(*out));
    out->pathBounces = h.pathBounces;
    out->iterations = c->iterations;
    /* bounces executed: the reference leaves its loop when no path is left or depth is reached */
    uint32_t depth = (uint32_t)c->traceDepth;
    for (int b = 0; b <= c->traceDepth && b < PT_MAX_DEPTH; ++b) out->lastLive[b] = h.lastLive[b];
    for (int b = 1; b <= c->traceDepth; ++b) if (h.lastLive[b] == 0) { depth = (uint32_t)b; break; }
    out->lastDepth = c->iterations ? depth : 0;
    out->nodesVisited = h.trav[0][0] + h.trav[1][0];
    out->trisTested = h.trav[0][1] + h.trav[1][1];
    out->meshHits = h.trav[0][2] + h.trav[1][2];
    if (getenv("PT_DEBUG_TAIL")) fprintf(stderr, "[pt] per-ray max: intersect nodes %u tris %u | shade nodes %u tris %u\n", h.maxNodes[0], h.maxTris[0], h.maxNodes[1], h.maxTris[1]);
    if (getenv("PT_DEBUG_TAIL"))
        for (int m = 0; m < 2; ++m)
            fprintf(stderr, "[pt] traversal mode %d: wave iterations %llu, node lanes %llu, leaf lanes %llu, held leaf lanes %llu, refills %llu, refilled lanes %llu\n", m,
                    h.waveDiag[m][0], h.waveDiag[m][1], h.waveDiag[m][2], h.waveDiag[m][3], h.waveDiag[m][4], h.waveDiag[m][5]),
            fprintf(stderr, "[pt] traversal mode %d: longest wave %u iterations\n", m, h.maxWaveIters[m]);
    if (getenv("PT_DEBUG_TAIL")) {
        fprintf(stderr, "[pt] waves ending per 25 us slot:");
        for (int i = 0; i < 64; ++i) fprintf(stderr, " %u", h.finishHist[i]);
        fprintf(stderr, "\n[pt] waves seeing the queue dry per 25 us slot:");
        for (int i = 0; i < 64; ++i) fprintf(stderr, " %u", h.dryHist[i]);
        fprintf(stderr, "\n");
    }
    out->shadeNodesVisited = h.trav[1][0]; out->shadeTrisTested = h.trav[1][1]; out->shadeMeshHits = h.trav[1][2];
    out->msCamera = c->ms[ST_CAMERA]; out->msIntersect = c->ms[ST_INTERSECT]; out->msSort = c->ms[ST_SORT];
    out->msShade = c->ms[ST_SHADE]; out->msCompact = c->ms[ST_COMPACT];
    out->launchesIntersect = c->launches[ST_INTERSECT]; out->launchesShade = c->launches[ST_SHADE];
    out->msTrace = c->ms[ST_TRACE]; out->msShadowTrace = c->ms[ST_SHADOW_TRACE];
    out->launchesTrace = c->launches[ST_TRACE]; out->launchesShadowTrace = c->launches[ST_SHADOW_TRACE];
    out->tracedRays = h.workItems[0]; out->shadowRays = h.workItems[1];
    out->boxesTested = h.trav[0][3] + h.trav[1][3]; out->shadeBoxesTested = h.trav[1][3];
    out->maxStackDepth = std::max(h.maxStack[0], h.maxStack[1]); out->ldsStackEntries = kLdsStack;
    out->referenceOrderWalks = h.refWalks;
    return PT_OK;
}

int pt_reset_stats(PtContext* c)
{
    if (!c) { pt_set_error("pt_reset_stats: null context"); return PT_EINVAL; }
    int rc = pt_sync(c);
    if (rc != PT_OK) return rc;
    HIP_TRY(hipMemset(c->dc, 0, sizeof(DevCounters)));
    c->iterations = 0;
    for (int i = 0; i < ST_COUNT; ++i) { c->ms[i] = 0.0f; c->launches[i] = 0; }
    return PT_OK;
}

void* pt_stream(PtContext* c) { return c ? (void*)c->stream : nullptr; }

unsigned long long pt_resource_calls(void) { return g_resourceCalls.load(); }

int pt_set_tree_passes(int passes)
{
    if (passes < 0 || passes > 16) { pt_set_error("pt_set_tree_passes: %d passes (0 to 16)", passes); return PT_EINVAL; }
    g_treePasses.store(passes);
    return PT_OK;
}

int pt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

/* the fixed-ray-set hook over staging buffers the caller of this helper owns: rays go through the production
 * intersect stage in chunks of at most `chunk` (<= one image's worth of paths) */
static int intersectStaged(PtContext* c, const float* origins, const float* directions, int n, PtHit* out,
                           float* dO, float* dD, PtHit* dH, int chunk, bool spans)
{
    SceneView V = c->view;
    V.useBvh = c->useBvh;
    V.numBvhMeshes = c->useBvh ? c->bvhMeshGeoms : 0;
    uint32_t* ctr = c->workCtr + (size_t)PT_MAX_DEPTH * kMaxBatch * kCtrStride;     /* a slot no bounce of a trace uses */
    uint32_t* fetch = c->fetchCtr + (size_t)PT_MAX_DEPTH * kFetchBouncePitch;
    const bool count = c->counters != 0;
    /* spans (stage timing, launch counts) take HIP events from a pool that grows on demand: only for callers that may allocate */
    struct MaybeSpan { PtContext* c; int stage; bool on; hipEvent_t a = nullptr;
        MaybeSpan(PtContext* ctx, int st, bool o) : c(ctx), stage(st), on(o) { if (on) { c->launches[stage]++; if (c->timing) { a = takeEvent(c); (void)hipEventRecord(a, c->stream); } } }
        ~MaybeSpan() { if (on && c->timing) { hipEvent_t b = takeEvent(c); (void)hipEventRecord(b, c->stream); c->spans.push_back({ a, b, stage }); } } };
    for (int first = 0; first < n; first += chunk) {
        const int m = std::min(chunk, n - first);
        hipError_t e = hipMemcpyAsync(dO, origins + 3 * (size_t)first, (size_t)m * 12, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dD, directions + 3 * (size_t)first, (size_t)m * 12, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(fetch, 0, (size_t)kFetchBouncePitch * sizeof(uint32_t), c->stream);     /* the eight counter lines of every row */
        if (e == hipSuccess) e = hipMemsetAsync(ctr + 3 * kWorkCtrSection, 0, sizeof(uint32_t), c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(ctr, 0, sizeof(uint32_t), c->stream);          /* k_refwalk's list length */
        if (e == hipSuccess) {
            Batch bt; bt.count = 1; bt.iterBase = 0; bt.iterStride = 0;
            c->pipeBarrier = true;      /* row 0 of the path state is used on the main stream */
            ptl_import_rays(c->stream, c->state, c->queues, dO, dD, m);
            if (V.numBvhMeshes > 0) {
                { MaybeSpan g(c, ST_INTERSECT, spans); ptl_cull(c->stream, V, c->state, c->queues, bt, 0, ctr + 3 * kWorkCtrSection); }
                { MaybeSpan g(c, ST_TRACE, spans); ptl_trace(c->stream, V, c->state, c->queues, bt, 0, false, count, fetch, ctr + 3 * kWorkCtrSection, ctr, c->dc); }
            }
            { MaybeSpan g(c, ST_INTERSECT, spans); ptl_prims(c->stream, V, c->state, c->queues, bt, 0, 0, count, c->dc, false); }
            ptl_export_hits(c->stream, c->state, m, dH);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(out + first, dH, (size_t)m * sizeof(PtHit), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { pt_set_error("pt_intersect_rays: %s", hipGetErrorString(e)); return PT_EHIP; }
        if (spans) resolveSpans(c);
    }
    return PT_OK;
}

int pt_intersect_rays(PtContext* c, const float* origins, const float* directions, int n, PtHit* out)
{
    if (!c || n < 0 || (n > 0 && (!origins || !directions || !out))) { pt_set_error("pt_intersect_rays: bad argument"); return PT_EINVAL; }
    if (n == 0) return PT_OK;
    if (c->batch < 1) { pt_set_error("pt_intersect_rays: no path state is allocated"); return PT_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    /* the rays go through the production intersect stage in chunks of one image's worth of paths */
    const int chunk = std::min(n, c->numPixels);
    float *dO = nullptr, *dD = nullptr; PtHit* dH = nullptr;
    HIP_TRY(hipMalloc((void**)&dO, (size_t)chunk * 12));
    hipError_t e1 = hipMalloc((void**)&dD, (size_t)chunk * 12);
    hipError_t e2 = hipMalloc((void**)&dH, (size_t)chunk * sizeof(PtHit));
    int rc = PT_OK;
    if (e1 != hipSuccess || e2 != hipSuccess) { pt_set_error("pt_intersect_rays: out of device memory"); rc = PT_ENOMEM; }
    if (rc == PT_OK) rc = intersectStaged(c, origins, directions, n, out, dO, dD, dH, chunk, true);
    (void)hipFree(dO); if (dD) (void)hipFree(dD); if (dH) (void)hipFree(dH);
    return rc;
}

int pt_intersect_rays_noalloc(PtContext* c, const float* origins, const float* directions, int n, PtHit* out)
{
    if (!c || n < 0 || (n > 0 && (!origins || !directions || !out))) { pt_set_error("pt_intersect_rays_noalloc: bad argument"); return PT_EINVAL; }
    if (n > PT_NOALLOC_MAX_RAYS) { pt_set_error("pt_intersect_rays_noalloc: %d rays exceed the staging array's %d", n, PT_NOALLOC_MAX_RAYS); return PT_ELIMIT; }
    if (n == 0) return PT_OK;
    if (c->batch < 1) { pt_set_error("pt_intersect_rays_noalloc: no path state is allocated"); return PT_EINVAL; }
    if (!c->staging) { pt_set_error("pt_intersect_rays_noalloc: the staging array was not resolved on device %d", c->device); return PT_EHIP; }
    HIP_TRY(hipSetDevice(c->device));
    /* origins | directions | hit records, each region a multiple of 256 bytes long */
    unsigned char* base = (unsigned char*)c->staging;
    float* dO = (float*)base;
    float* dD = (float*)(base + (size_t)PT_NOALLOC_MAX_RAYS * 12);
    PtHit* dH = (PtHit*)(base + (size_t)PT_NOALLOC_MAX_RAYS * 24);
    return intersectStaged(c, origins, directions, n, out, dO, dD, dH, std::min(n, c->numPixels), false);
}

int pt_capture_gbuffer(PtContext* c, int iter, float* normals, float* albedo)
{
    if (!c) { pt_set_error("pt_capture_gbuffer: null context"); return PT_EINVAL; }
    if (c->batch < 1) { pt_set_error("pt_capture_gbuffer: no path state is allocated"); return PT_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = (size_t)c->numPixels;
    if (!c->gNormals) { int rc = devAlloc(c, &c->gNormals, n * 3); if (rc != PT_OK) return rc; }
    if (!c->gAlbedo) { int rc = devAlloc(c, &c->gAlbedo, n * 3); if (rc != PT_OK) return rc; }
    /* camera rays of this iteration through the production intersect stage, on row 0 of the path state */
    SceneView V = c->view;
    V.useBvh = c->useBvh;
    V.numBvhMeshes = c->useBvh ? c->bvhMeshGeoms : 0;
    uint32_t* ctr = c->workCtr + (size_t)PT_MAX_DEPTH * kMaxBatch * kCtrStride;     /* a slot no bounce of a trace uses */
    uint32_t* fetch = c->fetchCtr + (size_t)PT_MAX_DEPTH * kFetchBouncePitch;
    Batch bt; bt.count = 1; bt.iterBase = iter; bt.iterStride = 1;
    HIP_TRY(hipMemsetAsync(fetch, 0, (size_t)kFetchBouncePitch * sizeof(uint32_t), c->stream));
    HIP_TRY(hipMemsetAsync(ctr + 3 * kWorkCtrSection, 0, sizeof(uint32_t), c->stream));
    HIP_TRY(hipMemsetAsync(ctr, 0, sizeof(uint32_t), c->stream));
    c->pipeBarrier = true;      /* row 0 of the path state is used on the main stream */
    ptl_camera(c->stream, V, c->cam, bt, c->traceDepth, c->state, c->queues, V.numBvhMeshes > 0 ? ctr + 3 * kWorkCtrSection : nullptr);
    if (V.numBvhMeshes > 0) ptl_trace(c->stream, V, c->state, c->queues, bt, 0, false, false, fetch, ctr + 3 * kWorkCtrSection, ctr, nullptr);
    ptl_prims(c->stream, V, c->state, c->queues, bt, 0, 0, false, nullptr, false);
    ptl_gbuffer(c->stream, c->state, V.materials, (int)n, c->gNormals, c->gAlbedo);
    HIP_TRY(hipGetLastError());
    if (normals) HIP_TRY(hipMemcpyAsync(normals, c->gNormals, n * 12, hipMemcpyDeviceToHost, c->stream));
    if (albedo) HIP_TRY(hipMemcpyAsync(albedo, c->gAlbedo, n * 12, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->gbufferValid = true;
    return PT_OK;
}

int pt_denoise(PtContext* c, float samples, int levels, float sigma_color, float sigma_normal, float sigma_albedo, float* out_rgb)
{
    if (!c || !out_rgb || !(samples > 0.0f) || levels < 1 || levels > 8 || !(sigma_color > 0.0f)) { pt_set_error("pt_denoise: bad argument"); return PT_EINVAL; }
    if ((sigma_normal > 0.0f || sigma_albedo > 0.0f) && !c->gbufferValid) { pt_set_error("pt_denoise: guide buffers wanted but pt_capture_gbuffer has not run"); return PT_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    const size_t n3 = (size_t)c->numPixels * 3;
    if (!c->dnA) { int rc = devAlloc(c, &c->dnA, n3); if (rc != PT_OK) return rc; }
    if (!c->dnB) { int rc = devAlloc(c, &c->dnB, n3); if (rc != PT_OK) return rc; }
    const int W = c->cam.resolution[0], H = c->cam.resolution[1];
    ptl_average(c->stream, c->imageSum, samples, n3, c->dnA);
    float* a = c->dnA;
    float* b = c->dnB;
    float sc = sigma_color;
    for (int l = 0; l < levels; ++l) {
        ptl_atrous(c->stream, a, sigma_normal > 0.0f ? c->gNormals : nullptr, sigma_albedo > 0.0f ? c->gAlbedo : nullptr, W, H, 1 << l,
                   1.0f / (sc * sc), sigma_normal > 0.0f ? 1.0f / (sigma_normal * sigma_normal) : 0.0f,
                   sigma_albedo > 0.0f ? 1.0f / (sigma_albedo * sigma_albedo) : 0.0f, b);
        std::swap(a, b);
        sc *= 0.5f;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_rgb, a, n3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_camera_rays(PtContext* c, int iter, float* origins, float* directions)
{
    if (!c || !origins || !directions) { pt_set_error("pt_camera_rays: null argument"); return PT_EINVAL; }
    if (c->batch < 1) { pt_set_error("pt_camera_rays: no path state is allocated"); return PT_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    const int n = c->numPixels;
    float *dO = nullptr, *dD = nullptr;
    HIP_TRY(hipMalloc((void**)&dO, (size_t)n * 12));
    if (hipMalloc((void**)&dD, (size_t)n * 12) != hipSuccess) { (void)hipFree(dO); pt_set_error("pt_camera_rays: out of device memory"); return PT_ENOMEM; }
    Batch bt; bt.count = 1; bt.iterBase = iter; bt.iterStride = 1;
    SceneView V = c->view;
    V.numBvhMeshes = 0;          /* no work list wanted here */
    c->pipeBarrier = true;      /* row 0 of the path state is used on the main stream */
    ptl_camera(c->stream, V, c->cam, bt, c->traceDepth, c->state, c->queues, nullptr);
    ptl_export_rays(c->stream, c->state, n, dO, dD);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(origins, dO, (size_t)n * 12, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(directions, dD, (size_t)n * 12, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dO); (void)hipFree(dD);
    if (e != hipSuccess) { pt_set_error("pt_camera_rays: %s", hipGetErrorString(e)); return PT_EHIP; }
    return PT_OK;
}

int pt_math_eval(int device, int fn, const float* a, const float* b, int n, float* out)
{
    if (!a || !out || n < 0) { pt_set_error("pt_math_eval: bad argument"); return PT_EINVAL; }
    if (n == 0) return PT_OK;
    HIP_TRY(hipSetDevice(device));
    float *dA = nullptr, *dB = nullptr, *dR = nullptr;
    /* every exit frees what was allocated */
    hipError_t e = hipMalloc((void**)&dA, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&dR, (size_t)n * 4);
    if (e == hipSuccess && b) e = hipMalloc((void**)&dB, (size_t)n * 4);
    if (e == hipSuccess) e = hipMemcpy(dA, a, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && b) e = hipMemcpy(dB, b, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        ptl_math(nullptr, fn, dA, dB, n, dR);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, dR, (size_t)n * 4, hipMemcpyDeviceToHost);
    if (dA) (void)hipFree(dA);
    if (dR) (void)hipFree(dR);
    if (dB) (void)hipFree(dB);
    if (e != hipSuccess) { pt_set_error("pt_math_eval: %s", hipGetErrorString(e)); return PT_EHIP; }
    return PT_OK;
}

} // extern "C"
