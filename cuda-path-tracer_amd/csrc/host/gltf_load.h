/*
 * gltf_load.h — glTF 2.0 (.glb / .gltf) mesh import, the role of Scene::loadGLTFModel,
 * loadGLTFMaterial and loadGLTFTexture (src/scene.cpp:44-568) without tinygltf.
 */
#pragma once
#include "../../../include/mi_pathtrace.h"
#include "xform.h"

#include <cstdint>
#include <string>
#include <vector>

namespace ptgltf {

struct Image {
    int width = 0, height = 0, components = 0;
    std::vector<uint8_t> pixels;
};

struct Import {
    std::vector<PtTriangle> triangles;   /* world space: root transform and node transforms applied */
    std::vector<PtMaterial> materials;   /* to be appended to the scene's table in this order */
    std::vector<Image> textures;         /* to be appended to the scene's table in this order */
    std::vector<std::string> warnings;
};

/* materialBase / textureBase: sizes of the scene's tables before the import (ids in the result are absolute). */
bool load(const std::string& path, const ptx::Mat4& root, int materialBase, int textureBase, Import& out, std::string& err);

} // namespace ptgltf
