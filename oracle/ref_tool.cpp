/*
 * ref_tool — command-line driver around the REFERENCE's own host-buildable sources, compiled
 * where they lie under /root/reference (see oracle/Makefile, target _ref/ref_tool).  Test
 * infrastructure: it exists so that the CPU oracle and the product's host code can be checked
 * against outputs of the reference itself; nothing in the product links or runs it.
 *
 * Linked in, unmodified:  src/scene.cpp (Scene::Scene -> loadFromJSON, loadGLTFModel, ...),
 * src/bvh.cpp (BVH::build), src/utilities.cpp, src/stb.cpp (stb_image, stb_image_write, tinygltf),
 * src/image.cpp (Image::savePNG / saveHDR), and src/intersections.cu through ref_isect_tu.cpp.
 * This file holds no algorithm of the reference except the closest-hit loop over geoms, which
 * lives inside a kernel of src/pathtrace.cu (:868-919) that a host compiler cannot parse and
 * is restated in closestHit() below around calls of the reference's own intersection functions.
 *
 * Modes (all files little-endian, packed):
 *   scene <scene.json> <out>            load through the reference's loader, dump every table
 *   bvh <tris.bin> <maxDepth> <out>     BVH::build on n 100-byte Triangle records
 *   isect <desc.bin> <rays.bin> <out> [nobvh]
 *                                       closest hits of rays against a dumped scene description
 *   isect-json <scene.json> <rays.bin> <out>
 *                                       the same, scene from the reference's loader
 *   stbi <file> <out>                   stbi_load (native + 4 components) or stbi_loadf (3) of a file
 *   png|hdr <w> <h> <pixels.bin> <base> Image::setPixel + savePNG / saveHDR -> <base>.png|.hdr
 */
#include "scene.h"
#include "bvh.h"
#include "image.h"
#include "intersections.h"
#include "stb_image.h"

#include <omp.h>

#include <cfloat>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static_assert(sizeof(Triangle) == 100 && sizeof(BVHNode) == 40 && sizeof(Material) == 124 && sizeof(Camera) == 84, "reference struct sizes");

/* read access to BVH's private vectors (explicit instantiation may name private members) */
template <typename Tag, typename Tag::type M> struct Expose { friend typename Tag::type get(Tag) { return M; } };
struct NodesTag { typedef std::vector<BVHNode> BVH::*type; friend type get(NodesTag); };
struct IndexTag { typedef std::vector<int> BVH::*type; friend type get(IndexTag); };
template struct Expose<NodesTag, &BVH::nodes>;
template struct Expose<IndexTag, &BVH::triangleIndices>;

static std::vector<unsigned char> readFile(const char* path)
{
    std::vector<unsigned char> b;
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "ref_tool: cannot read %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    b.resize((size_t)n);
    if (n && fread(b.data(), 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "ref_tool: short read %s\n", path); exit(2); }
    fclose(f);
    return b;
}

struct Out {
    FILE* f;
    explicit Out(const char* p) : f(fopen(p, "wb")) { if (!f) { fprintf(stderr, "ref_tool: cannot write %s\n", p); exit(2); } }
    ~Out() { fclose(f); }
    void raw(const void* p, size_t n) { if (n && fwrite(p, 1, n, f) != n) { fprintf(stderr, "ref_tool: short write\n"); exit(2); } }
    void i32(int32_t v) { raw(&v, 4); }
    void f32(float v) { raw(&v, 4); }
};

/* ---- scene tables as the wavefront loop would upload them (src/pathtrace.cu:483-505, :269-334) */
struct Loaded {
    std::vector<Geom> geoms;
    std::vector<Triangle> triangles;            /* concatenated in geom order */
    std::vector<BVH*> bvhs;
    std::vector<const BVHNode*> nodePtr;
    std::vector<const int*> indexPtr;
};

static void prepare(Loaded& L, const std::vector<std::vector<Triangle>>& meshes, const std::vector<int>& meshOfGeom, bool useBvh)
{
    for (size_t g = 0; g < L.geoms.size(); ++g) {
        Geom& G = L.geoms[g];
        G.bvhIndex = -1;
        if (G.type != GLTF_MESH || meshOfGeom[g] < 0) continue;
        const std::vector<Triangle>& t = meshes[(size_t)meshOfGeom[g]];
        G.triangleStart = (int)L.triangles.size();                       /* :491 */
        G.triangleCount = (int)t.size();
        L.triangles.insert(L.triangles.end(), t.begin(), t.end());
        if (useBvh && !t.empty()) {
            BVH* b = new BVH();
            b->build(t, 24);                                             /* BVH_MAX_TREE_DEPTH, :194 */
            G.bvhIndex = (int)L.bvhs.size();                             /* :308 */
            L.bvhs.push_back(b);
            L.nodePtr.push_back((b->*get(NodesTag())).data());
            L.indexPtr.push_back((b->*get(IndexTag())).data());
        }
    }
}

struct HitRec { float t; int32_t geom; int32_t materialId; float n[3]; float uv[2]; };

/* the loop of computeIntersectionsBVH (src/pathtrace.cu:868-933) for one ray */
static HitRec closestHit(const Loaded& L, const Ray& ray, bool useBvh)
{
    float t_min = FLT_MAX;
    int hit = -1, hitMat = -1;
    glm::vec3 normal(0.0f); glm::vec2 hitUv(0.0f);
    for (int i = 0; i < (int)L.geoms.size(); ++i) {
        const Geom& geom = L.geoms[(size_t)i];
        float t = -1.0f;
        glm::vec3 tmpP, tmpN; bool tmpOut; glm::vec2 tmpUv(0.5f); int tmpMat = geom.materialid;
        if (geom.type == CUBE) t = boxIntersectionTest(geom, ray, tmpP, tmpN, tmpOut);
        else if (geom.type == SPHERE) t = sphereIntersectionTest(geom, ray, tmpP, tmpN, tmpOut);
        else if (geom.type == GLTF_MESH) {
            if (useBvh && geom.bvhIndex >= 0)
                t = meshIntersectionTestBVH(geom, L.triangles.data(), L.nodePtr[(size_t)geom.bvhIndex], L.indexPtr[(size_t)geom.bvhIndex],
                                            ray, tmpP, tmpN, tmpOut, tmpUv, tmpMat);
            else
                t = meshIntersectionTest(geom, L.triangles.data(), ray, tmpP, tmpN, tmpOut, tmpUv, tmpMat);
        }
        if (t > 0.0f && t_min > t) { t_min = t; hit = i; normal = tmpN; hitMat = tmpMat; hitUv = tmpUv; }
    }
    HitRec r;
    memset(&r, 0, sizeof(r));
    r.t = hit < 0 ? -1.0f : t_min; r.geom = hit; r.materialId = hit < 0 ? -1 : hitMat;
    if (hit >= 0) { r.n[0] = normal.x; r.n[1] = normal.y; r.n[2] = normal.z; r.uv[0] = hitUv.x; r.uv[1] = hitUv.y; }
    return r;
}

static void runRays(const Loaded& L, const char* raysPath, const char* outPath, bool useBvh)
{
    std::vector<unsigned char> rb = readFile(raysPath);
    const size_t n = rb.size() / 24;
    const float* r = (const float*)rb.data();
    std::vector<HitRec> hits(n);
    const double t0 = omp_get_wtime();
#pragma omp parallel for schedule(dynamic, 256)
    for (long long i = 0; i < (long long)n; ++i) {
        Ray ray;
        ray.origin = glm::vec3(r[6 * i], r[6 * i + 1], r[6 * i + 2]);
        ray.direction = glm::vec3(r[6 * i + 3], r[6 * i + 4], r[6 * i + 5]);
        hits[(size_t)i] = closestHit(L, ray, useBvh);
    }
    const double t1 = omp_get_wtime();
    /* the closest-hit loop alone (no loading, no BVH build), for the CPU baseline of the benchmark */
    printf("ref_tool isect: %zu rays in %.6f s on %d threads\n", n, t1 - t0, omp_get_max_threads());
    Out o(outPath);
    o.raw(hits.data(), n * sizeof(HitRec));
}

static void dumpGeom(Out& o, const Geom& g, int meshIndex)
{
    o.i32((int)g.type); o.i32(g.materialid);
    o.raw(&g.translation, 12); o.raw(&g.rotation, 12); o.raw(&g.scale, 12);
    o.raw(&g.transform, 64); o.raw(&g.inverseTransform, 64); o.raw(&g.invTranspose, 64);
    o.i32(meshIndex); o.i32(g.triangleCount);
}

static std::vector<int> meshIndices(const Scene& s)
{
    std::vector<int> m(s.geoms.size(), -1);
    int next = 0;
    for (size_t g = 0; g < s.geoms.size(); ++g)
        if (s.geoms[g].type == GLTF_MESH && s.geoms[g].meshData != nullptr) m[g] = next++;     /* one mesh pushed per loaded model, src/scene.cpp:535-537 */
    return m;
}

static int modeScene(const char* json, const char* out)
{
    Scene s{ std::string(json) };
    std::vector<int> mi = meshIndices(s);
    Out o(out);
    o.i32(0x52454631);
    o.i32((int)s.geoms.size()); o.i32((int)s.materials.size()); o.i32((int)s.meshes.size()); o.i32((int)s.textures.size());
    for (size_t g = 0; g < s.geoms.size(); ++g) dumpGeom(o, s.geoms[g], mi[g]);
    o.raw(s.materials.data(), s.materials.size() * sizeof(Material));
    for (const MeshData& m : s.meshes) { o.i32((int)m.triangles.size()); o.raw(m.triangles.data(), m.triangles.size() * sizeof(Triangle)); }
    for (const Texture& t : s.textures) { o.i32(t.width); o.i32(t.height); o.i32(t.components); o.raw(t.data, (size_t)t.width * t.height * t.components); }
    /* enabled / width / height are only written when the file loads (src/scene.cpp:792-795): report "has data" */
    const bool env = !s.environmentMap.data.empty();
    o.i32(env ? 1 : 0);
    if (env) { o.i32(s.environmentMap.width); o.i32(s.environmentMap.height); o.f32(s.environmentMap.intensity);
               o.raw(s.environmentMap.data.data(), s.environmentMap.data.size() * 12); }
    o.raw(&s.state.camera, sizeof(Camera));
    o.i32((int)s.state.iterations); o.i32(s.state.traceDepth);
    o.i32((int)s.state.imageName.size()); o.raw(s.state.imageName.data(), s.state.imageName.size());
    return 0;
}

static int modeBvh(const char* trisPath, int maxDepth, const char* out)
{
    std::vector<unsigned char> b = readFile(trisPath);
    std::vector<Triangle> tris(b.size() / sizeof(Triangle));
    memcpy(tris.data(), b.data(), tris.size() * sizeof(Triangle));
    BVH bvh;
    bvh.build(tris, maxDepth);
    const std::vector<BVHNode>& nodes = bvh.*get(NodesTag());
    const std::vector<int>& idx = bvh.*get(IndexTag());
    const BVHStats& st = bvh.getStats();
    Out o(out);
    o.i32((int)nodes.size()); o.i32((int)idx.size()); o.i32(st.nodeCount); o.i32(st.leafCount); o.i32(st.maxDepth);
    o.raw(nodes.data(), nodes.size() * sizeof(BVHNode));
    o.raw(idx.data(), idx.size() * 4);
    return 0;
}

/* desc.bin: int32 geomCount, meshCount; per geom { int32 type, materialid; float TRS[9]; float transform[16], inverse[16],
 * invTranspose[16]; int32 meshIndex }; per mesh { int32 n; n Triangle records } */
static int modeIsect(const char* descPath, const char* rays, const char* out, bool useBvh)
{
    std::vector<unsigned char> b = readFile(descPath);
    const unsigned char* p = b.data();
    auto rdI = [&]() { int32_t v; memcpy(&v, p, 4); p += 4; return v; };
    const int ng = rdI(), nm = rdI();
    Loaded L;
    std::vector<int> meshOfGeom;
    for (int g = 0; g < ng; ++g) {
        Geom G;
        memset((void*)&G, 0, sizeof(G));
        G.type = (GeomType)rdI(); G.materialid = rdI();
        memcpy(&G.translation, p, 12); p += 12; memcpy(&G.rotation, p, 12); p += 12; memcpy(&G.scale, p, 12); p += 12;
        memcpy(&G.transform, p, 64); p += 64; memcpy(&G.inverseTransform, p, 64); p += 64; memcpy(&G.invTranspose, p, 64); p += 64;
        meshOfGeom.push_back(rdI());
        G.meshData = nullptr; G.triangleStart = -1; G.triangleCount = 0; G.bvhIndex = -1;
        L.geoms.push_back(G);
    }
    std::vector<std::vector<Triangle>> meshes((size_t)nm);
    for (int m = 0; m < nm; ++m) {
        const int n = rdI();
        meshes[(size_t)m].resize((size_t)n);
        memcpy(meshes[(size_t)m].data(), p, (size_t)n * sizeof(Triangle)); p += (size_t)n * sizeof(Triangle);
    }
    prepare(L, meshes, meshOfGeom, useBvh);
    runRays(L, rays, out, useBvh);
    return 0;
}

static int modeIsectJson(const char* json, const char* rays, const char* out)
{
    Scene s{ std::string(json) };
    Loaded L;
    L.geoms = s.geoms;
    std::vector<std::vector<Triangle>> meshes;
    for (const MeshData& m : s.meshes) meshes.push_back(m.triangles);
    prepare(L, meshes, meshIndices(s), true);
    runRays(L, rays, out, true);
    return 0;
}

static int modeStbi(const char* file, const char* out)
{
    Out o(out);
    int w = 0, h = 0, c = 0;
    if (stbi_is_hdr(file)) {
        float* d = stbi_loadf(file, &w, &h, &c, 3);          /* the call of src/scene.cpp:790 */
        if (!d) { o.i32(0); return 0; }
        o.i32(2); o.i32(w); o.i32(h); o.i32(c);
        o.raw(d, (size_t)w * h * 12);
        stbi_image_free(d);
        return 0;
    }
    unsigned char* d4 = stbi_load(file, &w, &h, &c, 4);      /* tinygltf asks for 4 components, external/include/tiny_gltf.h LoadImageData */
    if (!d4) { o.i32(0); return 0; }
    o.i32(1); o.i32(w); o.i32(h); o.i32(c);
    o.raw(d4, (size_t)w * h * 4);
    stbi_image_free(d4);
    float* df = stbi_loadf(file, &w, &h, &c, 3);              /* an 8-bit environment map goes through stbi_loadf as well */
    o.raw(df, (size_t)w * h * 12);
    stbi_image_free(df);
    return 0;
}

static int modeImage(bool png, int w, int h, const char* pixels, const char* base)
{
    std::vector<unsigned char> b = readFile(pixels);
    if (b.size() != (size_t)w * h * 12) { fprintf(stderr, "ref_tool: pixel file size\n"); return 2; }
    const float* f = (const float*)b.data();
    Image img(w, h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) img.setPixel(x, y, glm::vec3(f[3 * (y * w + x)], f[3 * (y * w + x) + 1], f[3 * (y * w + x) + 2]));
    if (png) img.savePNG(base); else img.saveHDR(base);
    return 0;
}

int main(int argc, char** argv)
{
    const std::string m = argc > 1 ? argv[1] : "";
    if (m == "scene" && argc == 4) return modeScene(argv[2], argv[3]);
    if (m == "bvh" && argc == 5) return modeBvh(argv[2], atoi(argv[3]), argv[4]);
    if (m == "isect" && (argc == 5 || argc == 6)) return modeIsect(argv[2], argv[3], argv[4], !(argc == 6 && std::string(argv[5]) == "nobvh"));
    if (m == "isect-json" && argc == 5) return modeIsectJson(argv[2], argv[3], argv[4]);
    if (m == "stbi" && argc == 4) return modeStbi(argv[2], argv[3]);
    if ((m == "png" || m == "hdr") && argc == 6) return modeImage(m == "png", atoi(argv[2]), atoi(argv[3]), argv[4], argv[5]);
    fprintf(stderr, "usage: ref_tool scene|bvh|isect|isect-json|stbi|png|hdr ... (see oracle/ref_tool.cpp)\n");
    return 2;
}
