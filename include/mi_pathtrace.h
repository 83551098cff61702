/*
 * mi_pathtrace.h — C ABI of the MI355X wavefront path tracer (libmi_pathtrace.so).
 *
 * Drop-in boundary for the wavefront loop of tonytgrt/CUDA-Path-Tracer.  The
 * reference exposes four C++ functions (src/pathtrace.h:8-11):
 *     void InitDataContainer(GuiDataContainer*);
 *     void pathtraceInit(Scene*);
 *     void pathtraceFree();
 *     void pathtrace(uchar4* pbo, int frame, int iteration);
 * operating on file-static device pointers (src/pathtrace.cu:168-201).  This
 * header is the plain-C surface those four functions bind to: the scene is
 * handed over as POD arrays (PtSceneDesc), state lives in an opaque context,
 * every call returns 0 on success and a negative PT_E* code otherwise
 * (the reference calls exit(1) in checkCUDAError, src/pathtrace.cu:27-48).
 *
 * All pointers inside PtSceneDesc are HOST pointers, borrowed for the
 * duration of the call only.  Matrices are column-major (m[col*4+row]), the
 * layout glm::mat4 uses in the reference (src/sceneStructs.h:68-70).
 */
#ifndef MI_PATHTRACE_H
#define MI_PATHTRACE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ enums */

/* src/sceneStructs.h:16-21 */
enum PtGeomType { PT_SPHERE = 0, PT_CUBE = 1, PT_GLTF_MESH = 2 };
/* src/sceneStructs.h:23-30 */
enum PtMaterialType { PT_DIFFUSE = 0, PT_SPECULAR = 1, PT_REFRACTIVE = 2, PT_PBR = 3, PT_EMITTING = 4 };

enum PtError {
    PT_OK = 0,
    PT_EINVAL = -1,   /* bad argument / inconsistent scene description */
    PT_EHIP = -2,     /* a HIP runtime call failed; see pt_last_error()  */
    PT_ENOMEM = -3,
    PT_EIO = -4,      /* scene file missing / unparsable                 */
    PT_ELIMIT = -5    /* size beyond a documented maximum                */
};

/* ------------------------------------------------------------ scene PODs */

/* src/sceneStructs.h:61-77 (Geom).  meshIndex replaces the CPU MeshData*.
 * triangleStart / bvhIndex are filled by pt_create exactly where
 * pathtraceInit fills them (src/pathtrace.cu:491, :308). */
typedef struct PtGeom {
    int32_t type;               /* PtGeomType */
    int32_t materialid;         /* -1 for meshes (per-triangle ids)     */
    float translation[3];
    float rotation[3];
    float scale[3];
    float transform[16];
    float inverseTransform[16];
    float invTranspose[16];
    int32_t meshIndex;          /* index into PtSceneDesc.meshes or -1   */
    int32_t triangleStart;
    int32_t triangleCount;
    int32_t bvhIndex;
} PtGeom;

/* src/sceneStructs.h:79-111 (Material) */
typedef struct PtMaterial {
    float color[3];
    float specularExponent;
    float specularColor[3];
    float transparency;
    float roughness;
    float metallic;
    float indexOfRefraction;
    float emittance;
    int32_t type;               /* PtMaterialType */
    int32_t baseColorTextureIdx;
    int32_t metallicRoughnessTextureIdx;
    int32_t normalTextureIdx;
    int32_t emissiveTextureIdx;
    int32_t occlusionTextureIdx;
    float emissiveFactor[3];
    float subsurfaceColor[3];
    float subsurfaceRadius;
    float subsurfaceRadiusRGB[3];
    float subsurfaceScale;
    float subsurfaceAnisotropy;
    int32_t subsurfaceEnabled;
} PtMaterial;

/* src/sceneStructs.h:40-46 (Triangle), 100 bytes */
typedef struct PtTriangle {
    float v0[3], v1[3], v2[3];
    float n0[3], n1[3], n2[3];
    float uv0[2], uv1[2], uv2[2];
    int32_t materialId;
} PtTriangle;

/* src/bvh.h:55-65 (BVHNode), 40 bytes; leaf <=> leftChild == -1 */
typedef struct PtBvhNode {
    float bmin[3];
    float bmax[3];
    int32_t leftChild;
    int32_t rightChild;
    int32_t triangleStart;
    int32_t triangleCount;
} PtBvhNode;

/* One mesh = the triangles of one glTF object (src/sceneStructs.h:48-53),
 * optionally with a prebuilt BVH (output of pt_bvh_build).  When nodes is
 * NULL pt_create builds the BVH itself, as initializeBVHs does
 * (src/pathtrace.cu:269-334).  A prebuilt BVH decides, as in the reference,
 * which triangle hits count (those whose leaf box the ray passes) and which of
 * two hits at the same distance wins (the one the reference visits first); its
 * boxes must contain the triangles below them, as any builder's do — for
 * boxes that do not, the reference's own result depends on its visiting order
 * and is not reproduced. */
typedef struct PtMesh {
    const PtTriangle* triangles;
    int32_t triangleCount;
    const PtBvhNode* nodes;         /* optional */
    int32_t nodeCount;
    const int32_t* triangleIndices; /* optional, leaf order -> triangle */
    int32_t triangleIndexCount;
} PtMesh;

/* src/sceneStructs.h:32-38 (Texture): 8-bit, 1/3/4 interleaved components */
typedef struct PtTexture {
    const uint8_t* data;
    int32_t width;
    int32_t height;
    int32_t components;
} PtTexture;

/* src/sceneStructs.h:119-129 (Camera) */
typedef struct PtCamera {
    int32_t resolution[2];
    float position[3];
    float lookAt[3];
    float view[3];
    float up[3];
    float right[3];
    float fov[2];
    float pixelLength[2];
} PtCamera;

/* src/sceneStructs.h:150-157 (HostEnvironmentMap).  data = width*height RGB
 * floats already multiplied by intensity (src/scene.cpp:810-817). */
typedef struct PtEnvMap {
    const float* data;
    int32_t width;
    int32_t height;
    float intensity;
    int32_t enabled;
} PtEnvMap;

typedef struct PtSceneDesc {
    const PtGeom* geoms;         int32_t geomCount;
    const PtMaterial* materials; int32_t materialCount;
    const PtMesh* meshes;        int32_t meshCount;
    const PtTexture* textures;   int32_t textureCount;
    PtEnvMap env;
    PtCamera camera;
    int32_t traceDepth;          /* RenderState::traceDepth */
    int32_t iterations;          /* RenderState::iterations (informative) */
} PtSceneDesc;

/* ------------------------------------------------------------- run-time */

typedef struct PtContext PtContext;

#define PT_MAX_DEPTH 64

typedef struct PtStats {
    uint64_t pathBounces;        /* sum over iterations and bounces of live paths
                                    entering the intersect stage
                                    (num_paths at src/pathtrace.cu:2735)          */
    uint64_t iterations;         /* pt_trace calls since create/clear              */
    uint32_t lastLive[PT_MAX_DEPTH]; /* live paths per bounce, last iteration      */
    uint32_t lastDepth;          /* bounces executed in the last iteration         */
    uint64_t nodesVisited;       /* BVH nodes box-tested  (only when counters on)  */
    uint64_t trisTested;         /* triangle tests        (only when counters on)  */
    uint64_t meshHits;           /* closest-hit queries that ended on a mesh       */
    float msIntersect;           /* HIP-event time of the stages, accumulated,     */
    float msSort;                /* only when timing is on (pt_set_option)         */
    float msShade;
    float msCompact;
    float msCamera;
    uint32_t launchesIntersect;
    uint32_t launchesShade;
    uint64_t shadeNodesVisited;  /* the shade stage's share of the three traversal     */
    uint64_t shadeTrisTested;    /* counters above (shadow rays); the rest belongs to  */
    uint64_t shadeMeshHits;      /* the intersect stage                                */
    /* the BVH traversal kernel on its own, in its two modes (the dominant kernels) */
    float msTrace;               /* closest-hit traversal, HIP-event time (timing on)  */
    float msShadowTrace;         /* shadow-ray traversal                               */
    uint32_t launchesTrace;
    uint32_t launchesShadowTrace;
    uint64_t tracedRays;         /* rays handed to the closest-hit traversal (always)  */
    uint64_t shadowRays;         /* shadow rays handed to the traversal (always)       */
    uint64_t boxesTested;        /* child boxes slab-tested, both modes (counters on)  */
    uint64_t shadeBoxesTested;   /* the shadow mode's share                            */
    uint32_t maxStackDepth;      /* deepest traversal stack any ray reached (counters on) */
    uint32_t ldsStackEntries;    /* entries per ray held in LDS; deeper ones spill to private memory */
    uint64_t referenceOrderWalks; /* mesh walks repeated in the reference's own leaf order because the closest hit
                                     depended on it (counters on) */
} PtStats;

/* Closest-hit record for the fixed-ray-set parity hook. */
typedef struct PtHit {
    float t;                     /* world distance, -1 on miss                      */
    int32_t geom;                /* index of the hit geom, -1 on miss               */
    int32_t triangle;            /* mesh-local triangle index, -1 for primitives    */
    int32_t materialId;
    float normal[3];
    float uv[2];
} PtHit;

enum PtOption {
    PT_OPT_MATERIAL_SORT = 1,    /* MATERIAL_SORTING, src/pathtrace.h:6 (default 1) */
    PT_OPT_USE_BVH = 2,          /* USE_BVH, src/pathtrace.cu:193 (default 1)       */
    PT_OPT_STAGE_TIMING = 3,     /* bracket every stage with HIP events (default 0);
                                    while on, a launch group stays on one stream    */
    PT_OPT_COUNTERS = 4,         /* count nodes/triangles in traversal (default 0)  */
    PT_OPT_USE_GRAPH = 5,        /* hipGraph replay of an iteration: not supported, a non-zero value returns PT_EINVAL */
    PT_OPT_BATCH = 6,            /* iterations that share one launch group in
                                    pt_trace_range (default 16, at most 512); results
                                    do not depend on it                              */
    PT_OPT_LAUNCH_STREAMS = 7    /* streams a launch group is spread over: 1..8 fixes the
                                    count, and pt_trace_range then never waits on the host;
                                    0 (default) lets the first five groups of a size try
                                    the shapes, each of those five calls ending with a host
                                    wait.  Results do not depend on it                  */
};

/* pathtraceInit (src/pathtrace.cu:428-576): uploads the scene, builds lights,
 * env-map CDFs and per-mesh BVHs, allocates path state for W*H paths. */
int pt_create(const PtSceneDesc* scene, int device, PtContext** out);

/* pathtraceFree (src/pathtrace.cu:578-684); NULL is a no-op. */
int pt_destroy(PtContext* ctx);

/* pathtrace(pbo = NULL, frame, iter) (src/pathtrace.cu:2659-2930): one sample
 * per pixel, accumulated into the running-sum image.  Asynchronous: the call
 * returns before the device is done, and calls issued ahead of the device
 * overlap there (up to four iterations in flight, each on a row of the path
 * state with streams of its own; six or eight when the process was started
 * with GPU_MAX_HW_QUEUES of 8 or 16, since streams beyond the hardware queues
 * serialise; environment PT_PIPELINE=n fixes the depth, 1 turns it off).
 * Their sums reach the image in call order, so it holds the same bits as with
 * one call at a time; every other entry point that reads the image or uses the
 * path state is ordered after all calls made before it. */
int pt_trace(PtContext* ctx, int iter);

/* Sample split: iterations iter_begin, iter_begin+stride, ... < iter_end.  Up to
 * PT_OPT_BATCH iterations form a launch group that runs through the same kernel
 * launches (each iteration with its own path state and queues, so results equal one
 * pt_trace call per iteration, bit for bit); a group runs on one to three streams
 * internally and is joined on the context's stream: the first five groups of a size
 * (12 iterations or more) try the shapes (one, two, three streams, and one stream with
 * the shadow traversal in line instead of beside the next bounce), each of those calls
 * ends with a HOST wait, and the fewest streams within 3 % of the fastest shape are kept.
 * A caller that needs every call to return without waiting fixes the count beforehand:
 * pt_set_option(ctx, PT_OPT_LAUNCH_STREAMS, n) with n in 1..8, or PT_STREAMS=n in the
 * environment when the context is created; no trial runs then, and no waits. */
int pt_trace_range(PtContext* ctx, int iter_begin, int iter_end, int iter_stride);

/* pathtrace(pbo, frame, iteration) called once per frame with iteration = 1, 2, 3, ... (src/main.cpp:812-837), at the
 * rate of launch groups.  The call has pt_trace's meaning — when it returns, iteration `iter` has been issued and the
 * running sum will hold every iteration asked for so far, added in call order, bit for bit what pt_trace gives — but the
 * first call renders iterations iter .. iter+ahead-1 in one launch group and starts the group after it, and the calls
 * that follow only add the row that was rendered for them while the device renders the rows further ahead.  Needs
 * PT_OPT_BATCH >= 2 * ahead (two halves of the path-state rows take turns); with less, ahead shrinks to half the rows,
 * down to plain pt_trace.  iter_last > 0 names the last iteration that will ever be asked for: nothing beyond it is
 * rendered.  A call that is not the expected next iteration, any pt_set_option, pt_trace or pt_trace_range drops what
 * was rendered ahead (correct, only slower).  pt_sync / pt_download / pt_stats wait for the rows ahead as well. */
int pt_trace_lookahead(PtContext* ctx, int iter, int ahead, int iter_last);

/* cudaMemcpy(hst_scene->state.image, dev_image) after every iteration (src/pathtrace.cu:2913-2917), without a wait in
 * it: host_rgb (W*H*3 floats that the caller keeps alive) is page-locked in place — or, if the runtime refuses,
 * reached through a pinned buffer of the library's — and every pt_trace_lookahead call from then on ends by queueing a
 * copy of the running sum into it, ordered behind that call's addition.  pt_host_image_wait blocks until the copy
 * of the most recent call has landed; a caller that reads the host image after every call (the reference's saveImage
 * only does after the last) waits after every call.  NULL unbinds. */
int pt_bind_host_image(PtContext* ctx, float* host_rgb);
int pt_host_image_wait(PtContext* ctx);

/* sendImageToPBO (src/pathtrace.cu:145-166): sum/iter -> RGBA8 into a DEVICE
 * buffer of W*H*4 bytes. */
int pt_display(PtContext* ctx, int iter, uint8_t* dev_rgba);

/* cudaMemcpy(hst_scene->state.image, dev_image) (src/pathtrace.cu:2916):
 * copies the W*H*3 float running sum to host memory (synchronises). */
int pt_download(PtContext* ctx, float* host_rgb);

/* Sample split inside one process (no counterpart in the reference, which drives one device, src/main.cpp:202): `count`
 * contexts — one per GPU, or several on one — have rendered disjoint iterations of the same scene, e.g. context r of R
 * the iterations 1+r, 1+r+R, ... (pt_trace_range with stride R).  Adds the running sums of the others to
 * contexts[root]'s, in index order, and waits for the result; a sum on another device crosses as one device-to-device
 * copy.  The union of the iterations is what one context would have rendered, so the result equals the one-context
 * image up to the order of the fp32 additions.  All contexts must have the same resolution. */
int pt_reduce_sum(PtContext** contexts, int count, int root);
/* devices a context can be created on (0 when there is none or the runtime is not usable) */
int pt_device_count(void);

/* The running sum lives in caller-provided DEVICE memory (W*H*3 floats), e.g.
 * a torch tensor that a collective reduces afterwards.  NULL = library-owned. */
int pt_bind_image(PtContext* ctx, float* dev_rgb);
/* Load a running sum (checkpoint/resume: the image is a plain sum). */
int pt_upload_image(PtContext* ctx, const float* host_rgb);
int pt_clear(PtContext* ctx);
/* A new camera for the iterations traced from now on, everything else of the scene kept: the reference's preview calls
 * pathtraceFree() + pathtraceInit() whenever the camera moves (src/main.cpp:812-822), which there is a few allocations
 * and copies, and here would rebuild the hierarchy the kernels walk (seconds for a large mesh).  Same resolution
 * only (PT_EINVAL otherwise); does not clear the running sum (pt_clear does); iterations rendered ahead are dropped. */
int pt_set_camera(PtContext* ctx, const PtCamera* camera);
/* A hash of everything in a scene description that pt_create builds device data from — geoms, materials, triangles and
 * host BVHs (arrays beyond 32 MB by 65 536 evenly spread records), textures, environment map, resolution, trace depth —
 * and not of the camera's pose: two descriptions with one fingerprint can share a context through pt_set_camera.
 * Never 0 for a description; 0 for NULL. */
unsigned long long pt_scene_fingerprint(const PtSceneDesc* scene);
int pt_sync(PtContext* ctx);
int pt_set_option(PtContext* ctx, int option, int value);
int pt_stats(PtContext* ctx, PtStats* out);
int pt_reset_stats(PtContext* ctx);
/* The HIP stream the context launches on (hipStream_t as void*). */
void* pt_stream(PtContext* ctx);

/* Parity hook: closest hit of n world-space rays (host arrays of xyz triples)
 * against the uploaded scene, through the same device code the intersect
 * stage runs (computeIntersectionsBVH, src/pathtrace.cu:834-934), one image's
 * worth of rays per launch.  With PT_OPT_STAGE_TIMING on, the kernels' device
 * time is added to msTrace (traversal) and msIntersect (work list, primitives)
 * of pt_stats; with PT_OPT_COUNTERS on, their box / triangle counts as well. */
int pt_intersect_rays(PtContext* ctx, const float* origins, const float* directions,
                      int n, PtHit* out);

/* pt_intersect_rays for callers that cannot allocate (code under HIP graph capture set-up, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Same kernels and results as
 * pt_intersect_rays; the three staging buffers (origins, directions, hit records) live in a static device array
 * of PT_NOALLOC_MAX_RAYS * 60 bytes (15 MiB) that every GPU running the library holds until the process ends, so
 * the call performs no hipMalloc / hipFree and takes no HIP event from the context's pool (stage timing and launch
 * counts of pt_stats do not see it).  n > PT_NOALLOC_MAX_RAYS returns PT_ELIMIT.  All contexts on one GPU share
 * the array: calls that overlap on one GPU race.  Pass page-locked host arrays if the HIP runtime's own staging
 * of pageable copies matters to the caller. */
#define PT_NOALLOC_MAX_RAYS 262144
int pt_intersect_rays_noalloc(PtContext* ctx, const float* origins, const float* directions,
                              int n, PtHit* out);

/* Parity hook: the camera stage alone (generateRayFromCamera,
 * src/pathtrace.cu:694-745): W*H origins and directions to host arrays. */
int pt_camera_rays(PtContext* ctx, int iter, float* origins, float* directions);

/* Device-side self check of the arithmetic the kernels rely on (IEEE div/sqrt,
 * the library's sin/cos/acos/atan2/exp/log): evaluates fn on n host inputs
 * (b may be NULL for unary functions) and returns the raw float results. */
int pt_math_eval(int device, int fn, const float* a, const float* b, int n, float* out);

/* Number of calls the library has made, in this process, that create or release a device-side resource (hipMalloc,
 * hipFree, pinned host memory, events, streams).  No counterpart in the reference; it lets a caller that may not
 * allocate — and the test of pt_intersect_rays_noalloc — see that a call made none. */
unsigned long long pt_resource_calls(void);

const char* pt_last_error(void);
const char* pt_version(void);

/* ------------------------------------------------- host-side scene helpers */

/* CPU BVH builder (role of BVH::build, src/bvh.cpp:30-238): SAH with 11
 * sampled planes per axis above 32 triangles, median split below, leaves of
 * <= 4 triangles, depth cap maxDepth.  No GPU needed. */
typedef struct PtBvh PtBvh;
int pt_bvh_build(const PtTriangle* tris, int n, int maxDepth, PtBvh** out);
const PtBvhNode* pt_bvh_nodes(const PtBvh* b, int* count);
const int32_t* pt_bvh_indices(const PtBvh* b, int* count);
int pt_bvh_stats(const PtBvh* b, int* nodeCount, int* leafCount, int* maxDepth);
void pt_bvh_free(PtBvh* b);

/* The device layout pt_create derives from a host BVH (src/bvh.h:55-65 nodes + index array): every
 * triangle keeps the box of its leaf in the host BVH (which decides whether a hit on it counts) and its
 * rank in the reference's visiting order; the hierarchy that is walked is built over boxes around
 * single triangles (two that end up as siblings share a leaf) and stored as 4-wide nodes.  This runs the same code on the host only (no GPU) and
 * checks the result: every triangle is referenced exactly once, every child box holds what is below
 * it, visit ranks are a permutation.  Returns PT_OK or PT_EINVAL with pt_last_error() naming the violation. */
typedef struct PtLayoutStats {
    int32_t wideNodes;           /* 4-wide nodes                                      */
    int32_t wideDepth;           /* levels of 4-wide nodes                            */
    int32_t leaves;              /* leaf references of the walked hierarchy (one or two triangles each) */
    int32_t trianglesReferenced;
    float sahCostHost;           /* sum over inner nodes of area(node)/area(root): host topology */
    float sahCostDevice;         /* the same measure for the 4-wide hierarchy         */
} PtLayoutStats;
int pt_bvh_device_layout(const PtTriangle* triangles, int triangleCount, const PtBvh* bvh, PtLayoutStats* out);

/* Set-up time against traversal speed.  The hierarchy the kernels walk is built by pt_create (and by
 * pt_bvh_device_layout) on one host thread: a binned-SAH build over single triangles, then `passes` rounds in which
 * subtrees are taken out and put back where they add the least surface area.  The default of 3 rounds is worth about
 * 4 % of traversal time and most of the set-up time of a large mesh (10 s of 10.2 s for 3 M triangles); 0 skips them.
 * Process-wide, read when a tree is built; results do not depend on it (the reference's own BVH, which decides the
 * hits, is built separately: BVH::build, src/bvh.cpp:74-238).  No counterpart in the reference. */
int pt_set_tree_passes(int passes);

/* Guide buffers and denoising (role of captureNormalsAndAlbedo, src/pathtrace.cu:94-135, and of the denoiser
 * step of pathtrace(), :2885-2910).
 * pt_capture_gbuffer: the camera rays of iteration iter through the intersect stage; per pixel the normalised
 * first-hit normal and the hit material's base colour, zeros where the ray leaves the scene.  Kept on the device
 * for pt_denoise; copied to the host arrays (W*H*3 floats each) that are not NULL.  Uses the path state, so call
 * it between traces.
 * pt_denoise: the reference's OptiX denoiser is closed; this is an open stand-in with the same inputs — an
 * edge-avoiding a-trous wavelet filter over (running sum / samples): `levels` passes (1..8) of a 5x5 B3-spline
 * kernel with holes 1, 2, 4, ...; tap weight exp(-(|dc|^2/sc^2 + |dn|^2/sn^2 + |da|^2/sa^2)), sc halving every
 * pass.  sigma_normal / sigma_albedo <= 0 leave that guide out.  out_rgb: W*H*3 floats on the host. */
int pt_capture_gbuffer(PtContext* ctx, int iter, float* normals, float* albedo);
int pt_denoise(PtContext* ctx, float samples, int levels, float sigma_color, float sigma_normal, float sigma_albedo,
               float* out_rgb);

/* Scene loader (role of Scene::loadFromJSON, src/scene.cpp:570-919, and of
 * the camera set-up in main.cpp:750-769, :788-807).  Returns an owning scene
 * whose PtSceneDesc stays valid until pt_scene_free.  No GPU needed. */
typedef struct PtScene PtScene;
int pt_scene_load(const char* json_path, PtScene** out);
const PtSceneDesc* pt_scene_desc(const PtScene* s);
/* Override RES / DEPTH (all shipped scenes are 1200x1200). */
int pt_scene_set_resolution(PtScene* s, int width, int height);
int pt_scene_set_depth(PtScene* s, int depth);
int pt_scene_set_iterations(PtScene* s, int iterations);     /* the ITERATIONS entry of the camera block (src/scene.cpp:868-905) */
/* Append a triangle mesh object (triangles already carry the object's TRS
 * baked in, as loadGLTFModel does, src/scene.cpp:265,423-425) together with
 * its materials and textures; TRS is kept as the geom transform as well
 * (src/scene.cpp:848-851). */
int pt_scene_add_mesh(PtScene* s, const PtTriangle* tris, int n,
                      const float translation[3], const float rotation[3], const float scale[3]);
/* Scene::loadGLTFModel + loadGLTFMaterial + loadGLTFTexture (src/scene.cpp:44-568): import a .glb / .gltf
 * file as one mesh object under the TRS triple — its textures (8-bit PNG or sequential JPEG, expanded to RGBA8 as
 * tinygltf / stb_image do for the reference) and materials are appended to the scene's tables, the triangles of every
 * scene node are baked into world space (node "matrix" read transposed, as src/scene.cpp:286-292 does).
 * pt_scene_load calls this for every object of TYPE "gltf" (src/scene.cpp:863-884); a file that cannot be
 * read leaves that object empty there, as in the reference.  Returns the geom index, or PT_EIO. */
int pt_scene_import_gltf(PtScene* s, const char* path,
                         const float translation[3], const float rotation[3], const float scale[3]);
/* Messages of the loader (skipped textures / primitives, unreadable glTF files), newline separated; the
 * reference prints the like to stdout (src/scene.cpp:214-226, :337-344). */
const char* pt_scene_log(const PtScene* s);
/* The PNG step of the texture import on its own (role of stbi_load_from_memory with 4 requested components
 * inside tinygltf, external/include/tiny_gltf.h): writes width*height*4 bytes when rgba is not NULL and
 * capacity suffices; call with rgba == NULL to learn the size. */
int pt_decode_png(const uint8_t* bytes, size_t n, uint8_t* rgba, size_t capacity, int* width, int* height);
/* The same for sequential and progressive 8-bit JPEG streams; follows stb_image's integer inverse DCT, chroma
 * upsampling, YCbCr -> RGB arithmetic and progressive scan handling (external/include/stb_image.h:2265-2412,
 * :2426-2524, :3465-3527, :3659-3683).  Arithmetic-coded files are refused, as stb_image refuses them. */
int pt_decode_jpeg(const uint8_t* bytes, size_t n, uint8_t* rgba, size_t capacity, int* width, int* height);
int pt_scene_add_material(PtScene* s, const PtMaterial* m);
int pt_scene_add_texture(PtScene* s, const uint8_t* data, int w, int h, int comps);
int pt_scene_set_envmap(PtScene* s, const float* rgb, int w, int h, float intensity);
void pt_scene_free(PtScene* s);
/* Host image of the scene, W*H*3 floats (role of RenderState::image, src/sceneStructs.h:160-166, which
 * pathtrace() refreshes with the running sum after every iteration, src/pathtrace.cu:2913-2917),
 * plus the ITERATIONS and FILE entries of the camera block (src/scene.cpp:868-905). */
float* pt_scene_image(PtScene* s);
int pt_scene_iterations(const PtScene* s);
const char* pt_scene_image_name(const PtScene* s);

/* Output side (role of saveImage, src/main.cpp:351-375, and Image::savePNG / saveHDR, src/image.cpp:23-50).
 * sum is the W*H*3 running sum (pt_download / pt_scene_image), samples the iteration count.
 *   pt_image_to_rgb8   x mirrored, sum / samples, clamp to [0,1], * 255, truncated: the bytes savePNG encodes
 *   pt_image_average   sum / samples as floats (mirror_x != 0: mirrored as saveImage does), what saveHDR encodes
 *   pt_write_png       8-bit grey / RGB / RGBA file (components 1, 3, 4)
 *   pt_write_hdr       Radiance RGBE file; float -> RGBE as stb_image_write does (stb_image_write.h:639-657) */
int pt_image_to_rgb8(const float* sum, int width, int height, float samples, uint8_t* rgb);
int pt_image_average(const float* sum, int width, int height, float samples, int mirror_x, float* out);
int pt_write_png(const char* path, const uint8_t* pixels, int width, int height, int components);
int pt_write_hdr(const char* path, const float* rgb, int width, int height);

/* utilityCore::buildTransformationMatrix + glm::inverse + glm::inverseTranspose
 * (src/utilities.cpp:85-93, src/scene.cpp:848-851) for one TRS triple. */
void pt_build_transforms(const float translation[3], const float rotation[3], const float scale[3],
                         float transform[16], float inverse[16], float invTranspose[16]);

#ifdef __cplusplus
}
#endif
#endif /* MI_PATHTRACE_H */
