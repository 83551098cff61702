/*
 * scene_host.cpp — host-side scene description: JSON scene loader, camera
 * set-up and programmatic scene assembly.  No GPU calls.
 *
 * Role of Scene::loadFromJSON (src/scene.cpp:570-919) for materials, primitive
 * objects, the camera block and the environment-map block, plus the camera
 * recomputation the reference's main loop applies before the first frame
 * (src/main.cpp:750-769 and :788-807): the kernels never see the JSON camera,
 * they see view/right/up/position rebuilt from (zoom, phi, theta), with an
 * unnormalised `right`.
 *
 * Differences that are deliberate:
 *  - HostEnvironmentMap.enabled is left uninitialised by the reference when the
 *    .hdr is missing (src/sceneStructs.h:150-157); here it is false.
 *  - a mesh object's Geom.materialid is uninitialised in the reference
 *    (src/scene.cpp:837-885); here it is -1.
 *  - glTF-imported materials leave indexOfRefraction / subsurface* unset
 *    (src/scene.cpp:96); pt_scene_add_material takes fully specified materials
 *    and the glTF importer (gltf_load.cpp) fills them with fixed values.
 */
#include "../../../include/mi_pathtrace.h"
#include "gltf_load.h"
#include "json_min.h"
#include "jpeg_decode.h"
#include "png_decode.h"
#include "xform.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <fstream>
#include <iterator>
#include <sstream>
#include <string>
#include <vector>

void pt_set_error(const char* fmt, ...);

namespace {
const float PI = 3.1415926535897932384626422832795028841971f;

struct V3 { float x, y, z; };
inline V3 v3(const float* p) { return V3{ p[0], p[1], p[2] }; }
inline V3 sub(V3 a, V3 b) { return V3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
inline V3 add(V3 a, V3 b) { return V3{ a.x + b.x, a.y + b.y, a.z + b.z }; }
inline V3 neg(V3 a) { return V3{ -a.x, -a.y, -a.z }; }
inline float dot(V3 a, V3 b) { float x = a.x * b.x, y = a.y * b.y, z = a.z * b.z; return x + y + z; }
inline V3 cross(V3 x, V3 y) { return V3{ x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y }; }
inline V3 normalize(V3 v) { float s = 1.0f / sqrtf(dot(v, v)); return V3{ v.x * s, v.y * s, v.z * s }; }
inline float length(V3 v) { return sqrtf(dot(v, v)); }
inline void put(float* d, V3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
inline float clamp01(float x) { float m = x > 0.0f ? x : 0.0f; return m < 1.0f ? m : 1.0f; } /* glm::clamp */
} // namespace

struct PtScene {
    std::vector<PtGeom> geoms;
    std::vector<PtMaterial> materials;
    std::vector<std::string> materialNames;
    std::deque<std::vector<PtTriangle>> meshTris;
    std::vector<PtMesh> meshes;
    std::deque<std::vector<uint8_t>> texData;
    std::vector<PtTexture> textures;
    std::vector<float> envData;
    PtEnvMap env{};
    PtCamera cam{};
    float fovy = 45.0f;
    int traceDepth = 8, iterations = 1;
    std::string imageName;
    std::string log;                    /* loader messages (the reference prints them to stdout) */
    std::vector<float> image;           /* W*H*3 running sum on the host, filled by the pathtrace() shim */
    PtSceneDesc desc{};

    void refresh()
    {
        for (size_t i = 0; i < meshes.size(); ++i) {
            meshes[i].triangles = meshTris[i].data();
            meshes[i].triangleCount = (int)meshTris[i].size();
        }
        for (size_t i = 0; i < textures.size(); ++i) textures[i].data = texData[i].data();
        env.data = envData.empty() ? nullptr : envData.data();
        desc.geoms = geoms.data(); desc.geomCount = (int)geoms.size();
        desc.materials = materials.data(); desc.materialCount = (int)materials.size();
        desc.meshes = meshes.data(); desc.meshCount = (int)meshes.size();
        desc.textures = textures.data(); desc.textureCount = (int)textures.size();
        desc.env = env;
        desc.camera = cam;
        desc.traceDepth = traceDepth;
        desc.iterations = iterations;
    }

    /* src/scene.cpp:903-913 followed by src/main.cpp:750-769 and :788-807 */
    void setupCamera()
    {
        float yscaled = tanf(fovy * (PI / 180));
        float xscaled = (yscaled * cam.resolution[0]) / cam.resolution[1];
        float fovx = (atanf(xscaled) * 180) / PI;
        cam.fov[0] = fovx; cam.fov[1] = fovy;
        cam.pixelLength[0] = 2 * xscaled / (float)cam.resolution[0];
        cam.pixelLength[1] = 2 * yscaled / (float)cam.resolution[1];
        V3 pos = v3(cam.position), look = v3(cam.lookAt);
        V3 view = normalize(sub(look, pos));

        /* main(): orbit angles from the loaded view */
        V3 viewXZ{ view.x, 0.0f, view.z };
        V3 viewZY{ 0.0f, view.y, view.z };
        float phi = acosf(dot(normalize(viewXZ), V3{ 0, 0, -1 }));
        float theta = acosf(dot(normalize(viewZY), V3{ 0, 1, 0 }));
        float zoom = length(sub(pos, look));

        /* runCuda(), first frame (camchanged) */
        V3 cp;
        cp.x = zoom * sinf(phi) * sinf(theta);
        cp.y = zoom * cosf(theta);
        cp.z = zoom * cosf(phi) * sinf(theta);
        V3 v = neg(normalize(cp));
        V3 u{ 0, 1, 0 };
        V3 r = cross(v, u);
        put(cam.view, v);
        put(cam.up, cross(r, v));
        put(cam.right, r);
        put(cam.position, add(cp, look));
    }
};

namespace {

PtMaterial defaultMaterial()
{
    PtMaterial m;
    memset(&m, 0, sizeof(m));           /* Material newMaterial{}; src/scene.cpp:580 */
    m.baseColorTextureIdx = -1;
    m.metallicRoughnessTextureIdx = -1;
    m.normalTextureIdx = -1;
    m.emissiveTextureIdx = -1;
    m.occlusionTextureIdx = -1;
    m.emittance = 0.0f;
    m.indexOfRefraction = 1.0f;
    return m;
}

/* src/scene.cpp:576-762 */
PtMaterial parseMaterial(const ptjson::Value& p)
{
    PtMaterial m = defaultMaterial();
    const auto& col = p.at("RGB");
    m.color[0] = col.at(0).f(); m.color[1] = col.at(1).f(); m.color[2] = col.at(2).f();
    const std::string& type = p.at("TYPE").s();
    if (type == "Diffuse") {
        m.transparency = 0.0f; m.roughness = 1.0f; m.metallic = 0.0f;
        m.type = PT_DIFFUSE;
    } else if (type == "Emitting") {
        m.emittance = p.at("EMITTANCE").f();
        m.type = PT_EMITTING;
    } else if (type == "Specular") {
        m.transparency = 0.0f; m.roughness = 0.0f; m.metallic = 0.2f;
        m.type = PT_SPECULAR;
    } else if (type == "Refractive") {
        m.transparency = 1.0f; m.roughness = 0.0f; m.metallic = 0.0f;
        m.indexOfRefraction = p.contains("IOR") ? p.at("IOR").f() : 1.5f;
        m.type = PT_REFRACTIVE;
    } else if (type == "PBR") {
        m.transparency = 0.0f; m.roughness = 0.5f; m.metallic = 0.0f;
        if (p.contains("TRANSPARENCY")) {
            m.transparency = clamp01(p.at("TRANSPARENCY").f());
            m.indexOfRefraction = p.contains("IOR") ? p.at("IOR").f() : 1.5f;
        }
        if (p.contains("ROUGHNESS")) m.roughness = clamp01(p.at("ROUGHNESS").f());
        if (p.contains("METALLIC")) m.metallic = clamp01(p.at("METALLIC").f());
        if (p.contains("SUBSURFACE_ENABLED")) {
            m.subsurfaceEnabled = p.at("SUBSURFACE_ENABLED").i();
            if (m.subsurfaceEnabled > 0) {
                if (p.contains("SUBSURFACE_COLOR")) {
                    const auto& c = p.at("SUBSURFACE_COLOR");
                    for (int k = 0; k < 3; ++k) m.subsurfaceColor[k] = c.at(k).f();
                } else {
                    for (int k = 0; k < 3; ++k) m.subsurfaceColor[k] = m.color[k] * 0.8f;
                }
                if (p.contains("SUBSURFACE_RADIUS")) {
                    const auto& r = p.at("SUBSURFACE_RADIUS");
                    if (r.isArray() && r.arr.size() == 3) { for (int k = 0; k < 3; ++k) m.subsurfaceRadiusRGB[k] = r.at(k).f(); }
                    else if (r.isNumber()) { for (int k = 0; k < 3; ++k) m.subsurfaceRadiusRGB[k] = r.f(); }
                } else {
                    for (int k = 0; k < 3; ++k) m.subsurfaceRadiusRGB[k] = 0.01f;
                }
                m.subsurfaceScale = p.contains("SUBSURFACE_SCALE") ? p.at("SUBSURFACE_SCALE").f() : 1.0f;
                if (p.contains("SUBSURFACE_ANISOTROPY")) {
                    float a = p.at("SUBSURFACE_ANISOTROPY").f();
                    float lo = a > -1.0f ? a : -1.0f;
                    m.subsurfaceAnisotropy = lo < 1.0f ? lo : 1.0f;
                } else m.subsurfaceAnisotropy = 0.0f;
                if (p.contains("SUBSURFACE_RADIUS_SINGLE")) {
                    float radius = p.at("SUBSURFACE_RADIUS_SINGLE").f();
                    m.subsurfaceRadius = radius;
                    if (!p.contains("SUBSURFACE_RADIUS")) for (int k = 0; k < 3; ++k) m.subsurfaceRadiusRGB[k] = radius;
                } else {
                    m.subsurfaceRadius = length(v3(m.subsurfaceRadiusRGB)) / 1.732f;
                }
            }
        } else {
            m.subsurfaceEnabled = 0;
            for (int k = 0; k < 3; ++k) { m.subsurfaceColor[k] = 1.0f; m.subsurfaceRadiusRGB[k] = 0.01f; }
            m.subsurfaceRadius = 0.01f;
            m.subsurfaceScale = 1.0f;
            m.subsurfaceAnisotropy = 0.0f;
        }
        m.type = PT_PBR;
    }
    return m;
}

/* An environment map from a Radiance .hdr file, or from a PNG or JPEG file the way stbi_loadf (src/scene.cpp:790)
 * turns 8-bit images into floats: three colour components, each pow(v / 255, 2.2) evaluated in double. */
bool loadRadianceHdr(const std::string& path, std::vector<float>& rgb, int& W, int& H);
bool loadEnvironmentImage(const std::string& path, std::vector<float>& rgb, int& W, int& H)
{
    if (loadRadianceHdr(path, rgb, W, H)) return true;
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::vector<uint8_t> px;
    std::string why;
    const bool png = bytes.size() >= 8 && memcmp(bytes.data(), "\x89PNG\r\n\x1a\n", 8) == 0;
    const bool jpg = bytes.size() >= 2 && bytes[0] == 0xFF && bytes[1] == 0xD8;
    if (png ? !ptpng::decode(bytes.data(), bytes.size(), px, W, H, why) : jpg ? !ptjpg::decode(bytes.data(), bytes.size(), px, W, H, why) : true) return false;
    float table[256];
    for (int v = 0; v < 256; ++v) table[v] = (float)(pow((double)(v / 255.0f), (double)2.2f) * 1.0f);
    rgb.resize((size_t)W * H * 3);
    for (size_t i = 0; i < (size_t)W * H; ++i)
        for (int k = 0; k < 3; ++k) rgb[3 * i + k] = table[px[4 * i + k]];
    return true;
}

PtGeom makeGeom(int type, int materialid, const float t[3], const float r[3], const float s[3])
{
    PtGeom g;
    memset(&g, 0, sizeof(g));
    g.type = type;
    g.materialid = materialid;
    memcpy(g.translation, t, 12); memcpy(g.rotation, r, 12); memcpy(g.scale, s, 12);
    pt_build_transforms(t, r, s, g.transform, g.inverseTransform, g.invTranspose);
    g.meshIndex = -1;
    g.triangleStart = -1;
    g.triangleCount = 0;
    g.bvhIndex = -1;
    return g;
}

/* Radiance .hdr (RGBE, optionally new-style RLE) -> float RGB, the subset of
 * stbi_loadf (src/scene.cpp:790) that equirect environment maps use. */
bool loadRadianceHdr(const std::string& path, std::vector<float>& rgb, int& W, int& H)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::string line;
    if (!std::getline(f, line) || (line.rfind("#?RADIANCE", 0) != 0 && line.rfind("#?RGBE", 0) != 0)) return false;
    bool fmtOk = false;
    while (std::getline(f, line)) {
        if (line.empty() || line == "\r") break;
        if (line.rfind("FORMAT=32-bit_rle_rgbe", 0) == 0) fmtOk = true;
    }
    if (!fmtOk) return false;
    if (!std::getline(f, line)) return false;
    if (sscanf(line.c_str(), "-Y %d +X %d", &H, &W) != 2 || W <= 0 || H <= 0) return false;
    rgb.assign((size_t)W * H * 3, 0.0f);
    std::vector<uint8_t> scan((size_t)W * 4);
    auto decode = [&](const uint8_t* px, float* out) {
        if (px[3] != 0) {
            float f1 = ldexpf(1.0f, (int)px[3] - (128 + 8));
            out[0] = px[0] * f1; out[1] = px[1] * f1; out[2] = px[2] * f1;
        } else out[0] = out[1] = out[2] = 0.0f;
    };
    for (int y = 0; y < H; ++y) {
        uint8_t hd[4];
        if (!f.read((char*)hd, 4)) return false;
        if (W < 8 || W >= 32768 || hd[0] != 2 || hd[1] != 2 || (hd[2] & 0x80)) {
            /* flat scanline */
            memcpy(scan.data(), hd, 4);
            if (!f.read((char*)scan.data() + 4, (std::streamsize)((size_t)W * 4 - 4))) return false;
            for (int x = 0; x < W; ++x) decode(&scan[(size_t)x * 4], &rgb[((size_t)y * W + x) * 3]);
            continue;
        }
        if (((int)hd[2] << 8 | hd[3]) != W) return false;
        std::vector<uint8_t> planes((size_t)W * 4);
        for (int k = 0; k < 4; ++k) {
            int x = 0;
            while (x < W) {
                int c = f.get();
                if (c == EOF) return false;
                if (c > 128) {
                    int n = c - 128, v = f.get();
                    if (v == EOF || x + n > W) return false;
                    for (int i = 0; i < n; ++i) planes[(size_t)k * W + x++] = (uint8_t)v;
                } else {
                    if (c == 0 || x + c > W) return false;
                    if (!f.read((char*)&planes[(size_t)k * W + x], c)) return false;
                    x += c;
                }
            }
        }
        for (int x = 0; x < W; ++x) {
            uint8_t px[4] = { planes[x], planes[(size_t)W + x], planes[(size_t)2 * W + x], planes[(size_t)3 * W + x] };
            decode(px, &rgb[((size_t)y * W + x) * 3]);
        }
    }
    return true;
}

std::string dirOf(const std::string& path)
{
    size_t s = path.find_last_of("/\\");
    return s == std::string::npos ? std::string() : path.substr(0, s + 1);
}

/* Put a triangle list into the first still-empty glTF object with the same TRS, else into a new object. */
int attachMesh(PtScene* s, std::vector<PtTriangle>&& tris, const float t[3], const float r[3], const float sc[3])
{
    int slot = -1;
    for (size_t i = 0; i < s->geoms.size(); ++i) {
        PtGeom& g = s->geoms[i];
        if (g.type == PT_GLTF_MESH && g.meshIndex < 0 && !memcmp(g.translation, t, 12) && !memcmp(g.rotation, r, 12) && !memcmp(g.scale, sc, 12)) { slot = (int)i; break; }
    }
    if (slot < 0) { s->geoms.push_back(makeGeom(PT_GLTF_MESH, -1, t, r, sc)); slot = (int)s->geoms.size() - 1; }
    const int n = (int)tris.size();
    s->meshTris.emplace_back(std::move(tris));
    PtMesh m; memset(&m, 0, sizeof(m));
    s->meshes.push_back(m);
    s->geoms[slot].meshIndex = (int)s->meshes.size() - 1;
    s->geoms[slot].triangleCount = n;
    return slot;
}

/* Scene::loadGLTFModel (src/scene.cpp:195-568): textures, then materials, then the triangles of every
 * scene node under the object's TRS.  Nothing is added when the file cannot be read or parsed. */
int importGltf(PtScene* s, const std::string& path, const float t[3], const float r[3], const float sc[3], std::string& err)
{
    ptgltf::Import imp;
    const ptx::Mat4 root = ptx::buildTransformationMatrix(t, r, sc);
    if (!ptgltf::load(path, root, (int)s->materials.size(), (int)s->textures.size(), imp, err)) return PT_EIO;
    for (const auto& w : imp.warnings) s->log += path + ": " + w + "\n";
    for (auto& img : imp.textures) {
        PtTexture tx; tx.data = nullptr; tx.width = img.width; tx.height = img.height; tx.components = img.components;
        s->texData.emplace_back(std::move(img.pixels));
        s->textures.push_back(tx);
    }
    for (const auto& m : imp.materials) { s->materials.push_back(m); s->materialNames.push_back(std::string()); }
    return attachMesh(s, std::move(imp.triangles), t, r, sc);
}

} // namespace

extern "C" {

void pt_build_transforms(const float t[3], const float r[3], const float s[3], float transform[16], float inverse[16],
                         float invTranspose[16])
{
    ptx::Mat4 M = ptx::buildTransformationMatrix(t, r, s);
    ptx::Mat4 I = ptx::inverse(M);
    ptx::Mat4 IT = ptx::inverseTranspose(M);
    memcpy(transform, M.c, 64);
    memcpy(inverse, I.c, 64);
    memcpy(invTranspose, IT.c, 64);
}

int pt_scene_load(const char* json_path, PtScene** out)
{
    if (!json_path || !out) { pt_set_error("pt_scene_load: null argument"); return PT_EINVAL; }
    std::ifstream f(json_path);
    if (!f) { pt_set_error("pt_scene_load: cannot open %s", json_path); return PT_EIO; }
    std::stringstream ss;
    ss << f.rdbuf();
    std::string text = ss.str();
    PtScene* S = new PtScene();
    try {
        ptjson::Value data = ptjson::Parser(text).parse();
        /* materials in key order; ids are positions in that order */
        for (const auto& kv : data.at("Materials").obj) {
            S->materialNames.push_back(kv.first);
            S->materials.push_back(parseMaterial(kv.second));
        }
        auto matId = [&](const std::string& name) -> int {
            for (size_t i = 0; i < S->materialNames.size(); ++i) if (S->materialNames[i] == name) return (int)i;
            /* unordered_map::operator[] default-inserts 0 (src/scene.cpp:856) */
            return 0;
        };
        if (data.contains("EnvironmentMap")) {
            const auto& e = data.at("EnvironmentMap");
            if (e.contains("FILE")) {
                std::string file = e.at("FILE").s();
                std::string full = file;
                if (file.size() > 1 && file[0] != '/' && file[1] != ':') full = dirOf(json_path) + file;
                int W = 0, H = 0;
                std::vector<float> rgb;
                if (!loadEnvironmentImage(full, rgb, W, H)) S->log += "environment map not loaded: " + full + "\n";
                else {
                    float intensity = e.contains("INTENSITY") ? e.at("INTENSITY").f() : 1.0f;
                    for (auto& v : rgb) v = v * intensity;
                    S->envData.swap(rgb);
                    S->env.width = W; S->env.height = H; S->env.intensity = intensity; S->env.enabled = 1;
                }
            }
        }
        for (const auto& p : data.at("Objects").arr) {
            const std::string& type = p.at("TYPE").s();
            float t[3], r[3], s[3];
            for (int k = 0; k < 3; ++k) { t[k] = p.at("TRANS").at(k).f(); r[k] = p.at("ROTAT").at(k).f(); s[k] = p.at("SCALE").at(k).f(); }
            if (type == "cube") S->geoms.push_back(makeGeom(PT_CUBE, matId(p.at("MATERIAL").s()), t, r, s));
            else if (type == "sphere") S->geoms.push_back(makeGeom(PT_SPHERE, matId(p.at("MATERIAL").s()), t, r, s));
            else if (type == "gltf") {
                /* src/scene.cpp:863-884.  When the file is missing or unreadable the object stays an empty
                   mesh, which is what the reference is left with (src/scene.cpp:218-226, :885); callers may
                   still attach geometry to it with pt_scene_add_mesh. */
                S->geoms.push_back(makeGeom(PT_GLTF_MESH, -1, t, r, s));
                if (p.contains("FILE")) {
                    std::string file = p.at("FILE").s();
                    std::string full = file;
                    if (file.size() > 1 && file[0] != '/' && file[1] != ':') full = dirOf(json_path) + file;
                    std::string why;
                    if (importGltf(S, full, t, r, s, why) < 0) S->log += "glTF object left empty: " + why + "\n";
                }
            } else {
                /* unknown TYPE: the reference pushes the geom with an unset type (src/scene.cpp:885) */
                throw std::runtime_error("unknown object TYPE '" + type + "'");
            }
        }
        const auto& c = data.at("Camera");
        S->cam.resolution[0] = c.at("RES").at(0).i();
        S->cam.resolution[1] = c.at("RES").at(1).i();
        S->fovy = c.at("FOVY").f();
        S->iterations = c.at("ITERATIONS").i();
        S->traceDepth = c.at("DEPTH").i();
        S->imageName = c.at("FILE").s();
        for (int k = 0; k < 3; ++k) {
            S->cam.position[k] = c.at("EYE").at(k).f();
            S->cam.lookAt[k] = c.at("LOOKAT").at(k).f();
            S->cam.up[k] = c.at("UP").at(k).f();
        }
        S->setupCamera();
    } catch (const std::exception& e) {
        pt_set_error("pt_scene_load: %s: %s", json_path, e.what());
        delete S;
        return PT_EIO;
    }
    S->refresh();
    *out = S;
    return PT_OK;
}

const PtSceneDesc* pt_scene_desc(const PtScene* s) { return s ? &s->desc : nullptr; }

int pt_scene_set_resolution(PtScene* s, int width, int height)
{
    if (!s || width <= 0 || height <= 0) { pt_set_error("pt_scene_set_resolution: bad argument"); return PT_EINVAL; }
    /* as if RES in the file had been (width, height): only pixelLength / fov depend on it */
    s->cam.resolution[0] = width; s->cam.resolution[1] = height;
    float yscaled = tanf(s->fovy * (PI / 180));
    float xscaled = (yscaled * width) / height;
    s->cam.fov[0] = (atanf(xscaled) * 180) / PI;
    s->cam.pixelLength[0] = 2 * xscaled / (float)width;
    s->cam.pixelLength[1] = 2 * yscaled / (float)height;
    s->refresh();
    return PT_OK;
}

int pt_scene_set_depth(PtScene* s, int depth)
{
    if (!s || depth < 1 || depth > PT_MAX_DEPTH) { pt_set_error("pt_scene_set_depth: depth must be in [1,%d]", PT_MAX_DEPTH); return PT_EINVAL; }
    s->traceDepth = depth;
    s->refresh();
    return PT_OK;
}

int pt_scene_set_iterations(PtScene* s, int iterations)
{
    if (!s || iterations < 1) { pt_set_error("pt_scene_set_iterations: iterations must be positive"); return PT_EINVAL; }
    s->iterations = iterations;
    s->refresh();
    return PT_OK;
}

int pt_scene_add_material(PtScene* s, const PtMaterial* m)
{
    if (!s || !m) { pt_set_error("pt_scene_add_material: null argument"); return PT_EINVAL; }
    s->materials.push_back(*m);
    s->materialNames.push_back(std::string());
    s->refresh();
    return (int)s->materials.size() - 1;
}

int pt_scene_add_texture(PtScene* s, const uint8_t* data, int w, int h, int comps)
{
    if (!s || !data || w <= 0 || h <= 0 || comps < 1 || comps > 4) { pt_set_error("pt_scene_add_texture: bad argument"); return PT_EINVAL; }
    s->texData.emplace_back(data, data + (size_t)w * h * comps);
    PtTexture t; t.data = nullptr; t.width = w; t.height = h; t.components = comps;
    s->textures.push_back(t);
    s->refresh();
    return (int)s->textures.size() - 1;
}

int pt_scene_add_mesh(PtScene* s, const PtTriangle* tris, int n, const float t[3], const float r[3], const float sc[3])
{
    if (!s || (!tris && n > 0) || n < 0) { pt_set_error("pt_scene_add_mesh: bad argument"); return PT_EINVAL; }
    int slot = attachMesh(s, std::vector<PtTriangle>(tris, tris + n), t, r, sc);
    s->refresh();
    return slot;
}

int pt_scene_import_gltf(PtScene* s, const char* path, const float t[3], const float r[3], const float sc[3])
{
    if (!s || !path || !t || !r || !sc) { pt_set_error("pt_scene_import_gltf: null argument"); return PT_EINVAL; }
    std::string why;
    int slot = importGltf(s, path, t, r, sc, why);
    if (slot < 0) { pt_set_error("pt_scene_import_gltf: %s: %s", path, why.c_str()); return slot; }
    s->refresh();
    return slot;
}

int pt_decode_png(const uint8_t* bytes, size_t n, uint8_t* rgba, size_t capacity, int* width, int* height)
{
    if (!bytes || !width || !height) { pt_set_error("pt_decode_png: null argument"); return PT_EINVAL; }
    std::vector<uint8_t> px;
    std::string why;
    if (!ptpng::decode(bytes, n, px, *width, *height, why)) { pt_set_error("pt_decode_png: %s", why.c_str()); return PT_EIO; }
    if (rgba) {
        if (capacity < px.size()) { pt_set_error("pt_decode_png: output buffer too small"); return PT_EINVAL; }
        memcpy(rgba, px.data(), px.size());
    }
    return PT_OK;
}

int pt_decode_jpeg(const uint8_t* bytes, size_t n, uint8_t* rgba, size_t capacity, int* width, int* height)
{
    if (!bytes || !width || !height) { pt_set_error("pt_decode_jpeg: null argument"); return PT_EINVAL; }
    std::vector<uint8_t> px;
    std::string why;
    if (!ptjpg::decode(bytes, n, px, *width, *height, why)) { pt_set_error("pt_decode_jpeg: %s", why.c_str()); return PT_EIO; }
    if (rgba) {
        if (capacity < px.size()) { pt_set_error("pt_decode_jpeg: output buffer too small"); return PT_EINVAL; }
        memcpy(rgba, px.data(), px.size());
    }
    return PT_OK;
}

const char* pt_scene_log(const PtScene* s) { return s ? s->log.c_str() : ""; }

int pt_scene_set_envmap(PtScene* s, const float* rgb, int w, int h, float intensity)
{
    if (!s) { pt_set_error("pt_scene_set_envmap: null scene"); return PT_EINVAL; }
    if (!rgb || w <= 0 || h <= 0) { s->envData.clear(); memset(&s->env, 0, sizeof(s->env)); s->refresh(); return PT_OK; }
    /* data arrives as loaded from file; the loader scales by INTENSITY once (src/scene.cpp:810-817) */
    s->envData.assign(rgb, rgb + (size_t)w * h * 3);
    for (auto& v : s->envData) v = v * intensity;
    s->env.width = w; s->env.height = h; s->env.intensity = intensity; s->env.enabled = 1;
    s->refresh();
    return PT_OK;
}

void pt_scene_free(PtScene* s) { delete s; }

float* pt_scene_image(PtScene* s)
{
    if (!s) return nullptr;
    const size_t n = (size_t)s->cam.resolution[0] * s->cam.resolution[1] * 3;
    if (s->image.size() != n) s->image.assign(n, 0.0f);
    return s->image.data();
}
int pt_scene_iterations(const PtScene* s) { return s ? s->iterations : 0; }
const char* pt_scene_image_name(const PtScene* s) { return s ? s->imageName.c_str() : ""; }

} // extern "C"
