/*
 * gltf_load.cpp — glTF 2.0 import: container (.glb chunks, .gltf with external or base64 buffers),
 * node hierarchy, triangle primitives, metallic-roughness materials, PNG / JPEG textures.  No GPU calls.
 *
 * Behaviour follows Scene::loadGLTFModel / loadGLTFMaterial / loadGLTFTexture (src/scene.cpp:44-568),
 * including the parts that decide the numbers the tracer sees:
 *  - every scene's root nodes are walked, primitives of a node before its children (:275-333, :525-527);
 *  - a node's "matrix" is read transposed (localTransform[j][i] = matrix[i*4+j], :286-292); without a
 *    matrix the local transform is T * R * S from translation / rotation (x,y,z,w) / scale (:296-326);
 *  - vertices are vec3(nodeTransform * vec4(p, 1)) with no perspective divide, normals
 *    normalize(mat3(transpose(inverse(nodeTransform))) * n), missing normals become the face normal of the
 *    transformed triangle, missing TEXCOORD_0 becomes (0,0) (:418-452);
 *  - only TRIANGLES primitives with 16- or 32-bit indices, or no indices, give triangles (:336-416, :467-520);
 *  - a primitive without a valid material takes the first imported material, a file without materials
 *    refers to scene material 0 (:254-258, :347-354);
 *  - materials: PBR; colour = baseColorFactor.rgb, transparency = 1 - alpha, metallic / roughness factors
 *    (glTF defaults 1 / 1), texture slots through the texture table, a non-zero emissiveFactor turns the
 *    material into an emitter of emittance |emissiveFactor|, alphaMode MASK sets transparency from
 *    alphaCutoff > 0.5 (:93-193);
 *  - every glTF texture becomes its own RGBA8 scene texture (:44-90; tinygltf asks stb_image for 4 components).
 * Arithmetic is in GLM 0.9.6.3's order of operations (xform.h), doubles of the file are narrowed to float
 * where the reference narrows them.
 *
 *  - a buffer view's byteStride is ignored: attributes and indices are read tightly packed (:363-420), so an
 *    interleaved file imports the way the reference imports it;
 * Deliberate differences: the scalars the reference leaves uninitialised in an imported Material (specular exponent, index of
 * refraction, subsurface radius / scale / anisotropy / enabled, src/scene.cpp:96) are 0, 1.5, 0, 0, 0, 0;
 * images that are neither 8-bit PNG nor 8-bit Huffman-coded JPEG (16-bit PNG, arithmetic-coded JPEG, other formats) are
 * skipped with a warning (texture index -1) instead of failing the load.
 */
#include "gltf_load.h"
#include "jpeg_decode.h"
#include "json_min.h"
#include "png_decode.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>

namespace ptgltf {
namespace {

using ptjson::Value;

struct F3 { float x, y, z; };
inline float dot3(F3 a, F3 b) { float x = a.x * b.x, y = a.y * b.y, z = a.z * b.z; return x + y + z; }
inline F3 sub3(F3 a, F3 b) { return F3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
inline F3 cross3(F3 x, F3 y) { return F3{ x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y }; }
inline F3 normalize3(F3 v) { float s = 1.0f / sqrtf(dot3(v, v)); return F3{ v.x * s, v.y * s, v.z * s }; }
inline void put3(float* d, F3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

/* vec3(M * vec4(p, 1)): (c0 x + c1 y) + (c2 z + c3 w), detail/type_mat4x4.inl */
inline F3 xformPoint(const ptx::Mat4& M, F3 p)
{
    float r[3];
    for (int i = 0; i < 3; ++i) r[i] = (M.c[0][i] * p.x + M.c[1][i] * p.y) + (M.c[2][i] * p.z + M.c[3][i] * 1.0f);
    return F3{ r[0], r[1], r[2] };
}

struct Mat3 { float c[3][3]; };
/* mat3(transpose(inverse(M))) */
inline Mat3 normalMatrix(const ptx::Mat4& M)
{
    ptx::Mat4 inv = ptx::inverse(M);
    Mat3 n;
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) n.c[c][r] = inv.c[r][c];
    return n;
}
inline F3 mul3(const Mat3& m, F3 v)
{
    return F3{ m.c[0][0] * v.x + m.c[1][0] * v.y + m.c[2][0] * v.z, m.c[0][1] * v.x + m.c[1][1] * v.y + m.c[2][1] * v.z,
               m.c[0][2] * v.x + m.c[1][2] * v.y + m.c[2][2] * v.z };
}

/* glm::mat4_cast(quat(w, x, y, z)), gtc/quaternion.inl:597-628 */
inline ptx::Mat4 quatToMat(float w, float x, float y, float z)
{
    ptx::Mat4 R = ptx::identity();
    float qxx = x * x, qyy = y * y, qzz = z * z, qxz = x * z, qxy = x * y, qyz = y * z, qwx = w * x, qwy = w * y, qwz = w * z;
    R.c[0][0] = 1 - 2 * (qyy + qzz);
    R.c[0][1] = 2 * (qxy + qwz);
    R.c[0][2] = 2 * (qxz - qwy);
    R.c[1][0] = 2 * (qxy - qwz);
    R.c[1][1] = 1 - 2 * (qxx + qzz);
    R.c[1][2] = 2 * (qyz + qwx);
    R.c[2][0] = 2 * (qxz + qwy);
    R.c[2][1] = 2 * (qyz - qwx);
    R.c[2][2] = 1 - 2 * (qxx + qyy);
    return R;
}

bool readFile(const std::string& path, std::vector<uint8_t>& out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end);
    std::streamoff n = f.tellg();
    if (n < 0) return false;
    f.seekg(0, std::ios::beg);
    out.resize((size_t)n);
    if (n > 0 && !f.read((char*)out.data(), n)) return false;
    return true;
}

bool decodeBase64(const std::string& s, size_t from, std::vector<uint8_t>& out)
{
    auto val = [](char c) -> int {
        if (c >= 'A' && c <= 'Z') return c - 'A';
        if (c >= 'a' && c <= 'z') return c - 'a' + 26;
        if (c >= '0' && c <= '9') return c - '0' + 52;
        if (c == '+' || c == '-') return 62;
        if (c == '/' || c == '_') return 63;
        return -1;
    };
    uint32_t acc = 0;
    int nbits = 0;
    for (size_t i = from; i < s.size(); ++i) {
        char c = s[i];
        if (c == '=' || c == '\n' || c == '\r') continue;
        int v = val(c);
        if (v < 0) return false;
        acc = (acc << 6) | (uint32_t)v;
        nbits += 6;
        if (nbits >= 8) { nbits -= 8; out.push_back((uint8_t)((acc >> nbits) & 0xFF)); }
    }
    return true;
}

/* uri -> bytes: data:...;base64,<payload> or a file next to the glTF */
bool loadUri(const std::string& uri, const std::string& baseDir, std::vector<uint8_t>& out)
{
    if (uri.compare(0, 5, "data:") == 0) {
        size_t comma = uri.find(',');
        if (comma == std::string::npos || uri.find(";base64") == std::string::npos) return false;
        return decodeBase64(uri, comma + 1, out);
    }
    return readFile(baseDir + uri, out);
}

struct Doc {
    Value json;
    std::vector<std::vector<uint8_t>> buffers;
    std::string baseDir;
};

const Value* member(const Value& v, const char* k) { return v.contains(k) ? &v.obj.find(k)->second : nullptr; }
int intOr(const Value& v, const char* k, int d) { const Value* m = member(v, k); return (m && m->isNumber()) ? (int)m->num : d; }
double numOr(const Value& v, const char* k, double d) { const Value* m = member(v, k); return (m && m->isNumber()) ? m->num : d; }
const std::vector<Value>& arrayOf(const Value& v, const char* k)
{
    static const std::vector<Value> none;
    const Value* m = member(v, k);
    return (m && m->isArray()) ? m->arr : none;
}

/* a bounds-checked view of one accessor */
struct View {
    const uint8_t* base = nullptr;
    size_t stride = 0, count = 0;
    int componentType = 0, comps = 0;
    size_t ignoredByteStride = 0;      /* the view's byteStride when it differs from the element size (it is not followed) */
};

bool accessorView(const Doc& d, int index, View& v, std::string& err)
{
    const auto& accs = arrayOf(d.json, "accessors");
    if (index < 0 || (size_t)index >= accs.size()) { err = "accessor index out of range"; return false; }
    const Value& a = accs[index];
    const auto& bvs = arrayOf(d.json, "bufferViews");
    int bvi = intOr(a, "bufferView", -1);
    if (bvi < 0 || (size_t)bvi >= bvs.size()) { err = "accessor without a buffer view"; return false; }
    const Value& bv = bvs[bvi];
    int bi = intOr(bv, "buffer", -1);
    if (bi < 0 || (size_t)bi >= d.buffers.size()) { err = "buffer index out of range"; return false; }
    v.componentType = intOr(a, "componentType", 0);
    const Value* type = member(a, "type");
    std::string t = (type && type->kind == Value::String) ? type->str : "";
    v.comps = t == "SCALAR" ? 1 : t == "VEC2" ? 2 : t == "VEC3" ? 3 : t == "VEC4" ? 4 : 0;
    size_t csize = v.componentType == 5126 || v.componentType == 5125 ? 4 : v.componentType == 5123 || v.componentType == 5122 ? 2
                 : v.componentType == 5121 || v.componentType == 5120 ? 1 : 0;
    if (!v.comps || !csize) { err = "unsupported accessor type"; return false; }
    v.count = (size_t)numOr(a, "count", 0);
    size_t elem = csize * v.comps;
    /* elements are read tightly packed from the accessor's start whatever the view's byteStride says: the reference
     * indexes positions[i * 3], normals[i * 3], texcoords[i * 2] and indices[i] directly (src/scene.cpp:363-420) */
    v.stride = elem;
    const size_t declared = (size_t)numOr(bv, "byteStride", 0);
    v.ignoredByteStride = (declared != 0 && declared != elem) ? declared : 0;
    size_t off = (size_t)numOr(bv, "byteOffset", 0) + (size_t)numOr(a, "byteOffset", 0);
    const auto& buf = d.buffers[bi];
    if (v.count && (v.stride < elem || off > buf.size() || (v.count - 1) * v.stride + elem > buf.size() - off)) { err = "accessor reaches past its buffer"; return false; }
    v.base = buf.data() + off;
    return true;
}

inline F3 f3At(const View& v, size_t i)
{
    float t[3];
    memcpy(t, v.base + i * v.stride, 12);
    return F3{ t[0], t[1], t[2] };
}
inline void f2At(const View& v, size_t i, float* out) { memcpy(out, v.base + i * v.stride, 8); }

bool parseContainer(const std::string& path, Doc& d, std::string& err)
{
    std::vector<uint8_t> raw;
    if (!readFile(path, raw)) { err = "cannot open " + path; return false; }
    size_t slash = path.find_last_of("/\\");
    d.baseDir = slash == std::string::npos ? std::string() : path.substr(0, slash + 1);
    std::string text;
    std::vector<uint8_t> bin;
    bool haveBin = false;
    if (raw.size() >= 12 && !memcmp(raw.data(), "glTF", 4)) {
        uint32_t version, total;
        memcpy(&version, raw.data() + 4, 4);
        memcpy(&total, raw.data() + 8, 4);
        if (version != 2 || total > raw.size()) { err = "unsupported GLB header"; return false; }
        size_t pos = 12;
        int chunk = 0;
        while (pos + 8 <= total) {
            uint32_t clen, ctype;
            memcpy(&clen, raw.data() + pos, 4);
            memcpy(&ctype, raw.data() + pos + 4, 4);
            if ((size_t)clen > total - pos - 8) { err = "GLB chunk reaches past the file"; return false; }
            const uint8_t* body = raw.data() + pos + 8;
            if (chunk == 0) {
                if (ctype != 0x4E4F534Au) { err = "GLB: first chunk is not JSON"; return false; }
                text.assign((const char*)body, clen);
            } else if (ctype == 0x004E4942u && !haveBin) { bin.assign(body, body + clen); haveBin = true; }
            pos += 8 + (size_t)clen;
            pos = (pos + 3) & ~(size_t)3;
            ++chunk;
        }
        if (text.empty()) { err = "GLB without a JSON chunk"; return false; }
    } else text.assign((const char*)raw.data(), raw.size());
    try {
        d.json = ptjson::Parser(text).parse();
    } catch (const std::exception& e) { err = e.what(); return false; }
    if (!d.json.isObject()) { err = "glTF: top level is not an object"; return false; }
    for (const Value& b : arrayOf(d.json, "buffers")) {
        std::vector<uint8_t> bytes;
        const Value* uri = member(b, "uri");
        if (uri && uri->kind == Value::String) {
            if (!loadUri(uri->str, d.baseDir, bytes)) { err = "cannot load buffer " + uri->str.substr(0, 64); return false; }
        } else if (haveBin && d.buffers.empty()) bytes.swap(bin);
        else { err = "buffer without data"; return false; }
        d.buffers.push_back(std::move(bytes));
    }
    return true;
}

/* src/scene.cpp:44-90 + tinygltf's image step: returns false (texture index -1) when the image cannot be used */
bool loadTextureImage(const Doc& d, const Value& tex, Image& img, std::string& why)
{
    const auto& images = arrayOf(d.json, "images");
    int src = intOr(tex, "source", -1);
    if (src < 0 || (size_t)src >= images.size()) { why = "invalid texture source index"; return false; }
    const Value& im = images[src];
    std::vector<uint8_t> holder;
    const uint8_t* p = nullptr;
    size_t n = 0;
    int bvi = intOr(im, "bufferView", -1);
    const Value* uri = member(im, "uri");
    if (bvi >= 0) {
        const auto& bvs = arrayOf(d.json, "bufferViews");
        if ((size_t)bvi >= bvs.size()) { why = "image buffer view out of range"; return false; }
        int bi = intOr(bvs[bvi], "buffer", -1);
        size_t off = (size_t)numOr(bvs[bvi], "byteOffset", 0), len = (size_t)numOr(bvs[bvi], "byteLength", 0);
        if (bi < 0 || (size_t)bi >= d.buffers.size() || off > d.buffers[bi].size() || len > d.buffers[bi].size() - off) { why = "image buffer view reaches past its buffer"; return false; }
        p = d.buffers[bi].data() + off; n = len;
    } else if (uri && uri->kind == Value::String) {
        if (!loadUri(uri->str, d.baseDir, holder)) { why = "cannot load image " + uri->str.substr(0, 64); return false; }
        p = holder.data(); n = holder.size();
    } else { why = "image without data"; return false; }
    const bool jpeg = n >= 2 && p[0] == 0xff && p[1] == 0xd8;
    if (jpeg ? !ptjpg::decode(p, n, img.pixels, img.width, img.height, why) : !ptpng::decode(p, n, img.pixels, img.width, img.height, why)) return false;
    img.components = 4;
    return true;
}

/* src/scene.cpp:93-193 */
PtMaterial importMaterial(const Value& gm, const std::vector<int>& textureMap)
{
    PtMaterial m;
    memset(&m, 0, sizeof(m));
    m.type = PT_PBR;
    m.baseColorTextureIdx = m.metallicRoughnessTextureIdx = m.normalTextureIdx = m.emissiveTextureIdx = m.occlusionTextureIdx = -1;
    m.transparency = 0.0f;
    m.emittance = 0.0f;
    m.indexOfRefraction = 1.5f;
    static const Value none;
    const Value* pbrp = member(gm, "pbrMetallicRoughness");
    const Value& pbr = (pbrp && pbrp->isObject()) ? *pbrp : none;

    double bc[4] = { 1.0, 1.0, 1.0, 1.0 };
    size_t bcN = 4;
    const Value* bcf = member(pbr, "baseColorFactor");
    if (bcf && bcf->isArray()) {
        bcN = bcf->arr.size();
        for (size_t i = 0; i < bcN && i < 4; ++i) bc[i] = bcf->arr[i].isNumber() ? bcf->arr[i].num : 0.0;
    }
    if (bcN >= 3) {
        m.color[0] = (float)bc[0]; m.color[1] = (float)bc[1]; m.color[2] = (float)bc[2];
        if (bcN >= 4) m.transparency = (float)(1.0 - bc[3]);
    } else m.color[0] = m.color[1] = m.color[2] = 1.0f;
    m.metallic = (float)numOr(pbr, "metallicFactor", 1.0);
    m.roughness = (float)numOr(pbr, "roughnessFactor", 1.0);

    auto slot = [&](const Value& owner, const char* key) -> int {
        const Value* t = member(owner, key);
        if (!t || !t->isObject()) return -1;
        int idx = intOr(*t, "index", -1);
        if (idx < 0 || (size_t)idx >= textureMap.size()) return -1;
        return textureMap[idx];
    };
    m.baseColorTextureIdx = slot(pbr, "baseColorTexture");
    m.metallicRoughnessTextureIdx = slot(pbr, "metallicRoughnessTexture");
    m.normalTextureIdx = slot(gm, "normalTexture");
    m.emissiveTextureIdx = slot(gm, "emissiveTexture");
    m.occlusionTextureIdx = slot(gm, "occlusionTexture");

    const Value* ef = member(gm, "emissiveFactor");
    if (ef && ef->isArray() && ef->arr.size() >= 3) {
        F3 e{ (float)ef->arr[0].num, (float)ef->arr[1].num, (float)ef->arr[2].num };
        put3(m.emissiveFactor, e);
        float len = sqrtf(dot3(e, e));
        if (len > 0.0f) { m.emittance = len; m.type = PT_EMITTING; }
    }
    const Value* am = member(gm, "alphaMode");
    if (am && am->kind == Value::String && am->str == "MASK")
        m.transparency = (numOr(gm, "alphaCutoff", 0.5) > (double)0.5f) ? 1.0f : 0.0f;
    return m;
}

} // namespace

bool load(const std::string& path, const ptx::Mat4& root, int materialBase, int textureBase, Import& out, std::string& err)
{
    Doc d;
    if (!parseContainer(path, d, err)) return false;

    /* textures first, then materials (src/scene.cpp:236-252) */
    std::vector<int> textureMap;
    const auto& gtex = arrayOf(d.json, "textures");
    for (size_t i = 0; i < gtex.size(); ++i) {
        Image img;
        std::string why;
        if (loadTextureImage(d, gtex[i], img, why)) {
            out.textures.push_back(std::move(img));
            textureMap.push_back(textureBase + (int)out.textures.size() - 1);
        } else {
            out.warnings.push_back("texture " + std::to_string(i) + " skipped: " + why);
            textureMap.push_back(-1);
        }
    }
    std::vector<int> materialMap;
    for (const Value& gm : arrayOf(d.json, "materials")) {
        out.materials.push_back(importMaterial(gm, textureMap));
        materialMap.push_back(materialBase + (int)out.materials.size() - 1);
    }
    if (materialMap.empty()) materialMap.push_back(0);

    const auto& nodes = arrayOf(d.json, "nodes");
    const auto& meshes = arrayOf(d.json, "meshes");
    bool ok = true;
    int depthGuard = 0;

    std::function<void(int, const ptx::Mat4&)> walk = [&](int ni, const ptx::Mat4& parent) {
        if (!ok || ni < 0 || (size_t)ni >= nodes.size()) return;
        if (++depthGuard > 512) { ok = false; err = "glTF node hierarchy too deep (cycle?)"; return; }
        const Value& node = nodes[ni];
        ptx::Mat4 local;
        const Value* mat = member(node, "matrix");
        if (mat && mat->isArray() && mat->arr.size() == 16) {
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) local.c[j][i] = (float)mat->arr[i * 4 + j].num;
        } else {
            ptx::Mat4 I = ptx::identity(), T = I, R = I, S = I;
            const Value* t = member(node, "translation");
            if (t && t->isArray() && t->arr.size() == 3) {
                float v[3] = { (float)t->arr[0].num, (float)t->arr[1].num, (float)t->arr[2].num };
                T = ptx::translate(I, v);
            }
            const Value* q = member(node, "rotation");
            if (q && q->isArray() && q->arr.size() == 4)
                R = quatToMat((float)q->arr[3].num, (float)q->arr[0].num, (float)q->arr[1].num, (float)q->arr[2].num);
            const Value* s = member(node, "scale");
            if (s && s->isArray() && s->arr.size() == 3) {
                float v[3] = { (float)s->arr[0].num, (float)s->arr[1].num, (float)s->arr[2].num };
                S = ptx::scale(I, v);
            }
            local = ptx::matmul(ptx::matmul(T, R), S);
        }
        const ptx::Mat4 M = ptx::matmul(parent, local);

        int mi = intOr(node, "mesh", -1);
        if (mi >= 0 && (size_t)mi < meshes.size()) {
            for (const Value& prim : arrayOf(meshes[mi], "primitives")) {
                if (intOr(prim, "mode", 4) != 4) { out.warnings.push_back("non-triangle primitive skipped"); continue; }
                const Value* attrs = member(prim, "attributes");
                if (!attrs || !attrs->contains("POSITION")) { out.warnings.push_back("primitive without POSITION skipped"); continue; }
                int gmat = intOr(prim, "material", -1);
                int sceneMat = (gmat >= 0 && (size_t)gmat < materialMap.size()) ? materialMap[gmat] : materialMap[0];

                View pos, nrm, uv, idx;
                bool haveN = false, haveUV = false, haveIdx = false;
                if (!accessorView(d, intOr(*attrs, "POSITION", -1), pos, err)) { ok = false; return; }
                /* an interleaved file comes out as the reference imports it — scrambled; say so, so that a caller can see why */
                auto strideNote = [&](const View& vw, const char* what) {
                    if (vw.ignoredByteStride)
                        out.warnings.push_back(std::string(what) + " accessor: buffer view byteStride " + std::to_string(vw.ignoredByteStride) +
                                               " is ignored, elements are read tightly packed as the reference reads them (src/scene.cpp:363-420); an interleaved file imports as wrong geometry");
                };
                strideNote(pos, "POSITION");
                if (pos.componentType != 5126 || pos.comps != 3) { ok = false; err = "POSITION must be float VEC3"; return; }
                if (attrs->contains("NORMAL")) {
                    if (!accessorView(d, intOr(*attrs, "NORMAL", -1), nrm, err)) { ok = false; return; }
                    strideNote(nrm, "NORMAL");
                    if (nrm.componentType != 5126 || nrm.comps != 3) { ok = false; err = "NORMAL must be float VEC3"; return; }
                    haveN = true;
                }
                if (attrs->contains("TEXCOORD_0")) {
                    if (!accessorView(d, intOr(*attrs, "TEXCOORD_0", -1), uv, err)) { ok = false; return; }
                    strideNote(uv, "TEXCOORD_0");
                    if (uv.componentType != 5126 || uv.comps != 2) { ok = false; err = "TEXCOORD_0 must be float VEC2"; return; }
                    haveUV = true;
                }
                int ia = intOr(prim, "indices", -1);
                if (ia >= 0) {
                    if (!accessorView(d, ia, idx, err)) { ok = false; return; }
                    haveIdx = true;
                    if (idx.componentType != 5123 && idx.componentType != 5125) { out.warnings.push_back("unsupported index type, primitive skipped"); continue; }
                }
                const Mat3 N = haveN ? normalMatrix(M) : Mat3{};
                const size_t corners = haveIdx ? idx.count : pos.count;
                for (size_t i = 0; i + 2 < corners; i += 3) {
                    size_t v[3];
                    for (int k = 0; k < 3; ++k) {
                        if (haveIdx) {
                            if (idx.componentType == 5123) { uint16_t t; memcpy(&t, idx.base + (i + k) * idx.stride, 2); v[k] = t; }
                            else { uint32_t t; memcpy(&t, idx.base + (i + k) * idx.stride, 4); v[k] = t; }
                        } else v[k] = i + k;
                    }
                    if (v[0] >= pos.count || v[1] >= pos.count || v[2] >= pos.count || (haveN && (v[0] >= nrm.count || v[1] >= nrm.count || v[2] >= nrm.count)) ||
                        (haveUV && (v[0] >= uv.count || v[1] >= uv.count || v[2] >= uv.count))) {
                        ok = false; err = "vertex index out of range"; return;
                    }
                    PtTriangle tri;
                    memset(&tri, 0, sizeof(tri));
                    F3 p0 = xformPoint(M, f3At(pos, v[0])), p1 = xformPoint(M, f3At(pos, v[1])), p2 = xformPoint(M, f3At(pos, v[2]));
                    put3(tri.v0, p0); put3(tri.v1, p1); put3(tri.v2, p2);
                    if (haveN) {
                        put3(tri.n0, normalize3(mul3(N, f3At(nrm, v[0]))));
                        put3(tri.n1, normalize3(mul3(N, f3At(nrm, v[1]))));
                        put3(tri.n2, normalize3(mul3(N, f3At(nrm, v[2]))));
                    } else {
                        F3 fn = normalize3(cross3(sub3(p1, p0), sub3(p2, p0)));
                        put3(tri.n0, fn); put3(tri.n1, fn); put3(tri.n2, fn);
                    }
                    if (haveUV) { f2At(uv, v[0], tri.uv0); f2At(uv, v[1], tri.uv1); f2At(uv, v[2], tri.uv2); }
                    tri.materialId = sceneMat;
                    out.triangles.push_back(tri);
                }
            }
        }
        for (const Value& c : arrayOf(node, "children"))
            if (c.isNumber()) walk((int)c.num, M);
        --depthGuard;
    };

    for (const Value& scene : arrayOf(d.json, "scenes"))
        for (const Value& n : arrayOf(scene, "nodes"))
            if (n.isNumber()) walk((int)n.num, root);
    return ok;
}

} // namespace ptgltf
