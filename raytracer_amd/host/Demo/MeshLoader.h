// helpers::LoadMesh & co. -- the mesh ingestion of the reference's Demo (Demo/MeshLoader.h:14-17, MeshLoader.cpp).
#pragma once

#include "../Core/Shapes/MeshShape.h"
#include "../Core/Material/Material.h"
#include "../Core/Utils/Bitmap.h"
#include "../Core/Textures/Texture.h"

#include <map>
#include <memory>
#include <string>
#include <vector>

namespace helpers {

using MaterialsMap = std::map<std::string, rt::MaterialPtr>;

RAYLIB_API rt::BitmapPtr LoadBitmapObject(const std::string& baseDir, const std::string& path);
RAYLIB_API rt::TexturePtr LoadTexture(const std::string& baseDir, const std::string& path);
RAYLIB_API rt::MeshShapePtr LoadMesh(const std::string& filePath, MaterialsMap& outMaterials, const float scale = 1.0f);
RAYLIB_API rt::MaterialPtr CreateDefaultMaterial(MaterialsMap& outMaterials);

// The vertex streams LoadMesh hands to MeshShape::Initialize (MeshLoader.cpp:372-392), exposed for the parity tests.
struct MeshStreams
{
    std::vector<rt::uint32> vertexIndices, materialIndices;
    std::vector<rt::math::Float3> positions, normals, tangents;
    std::vector<rt::math::Float2> texCoords;
    std::vector<rt::MaterialPtr> materials;
};
RAYLIB_API bool LoadMeshStreams(const std::string& filePath, MaterialsMap& outMaterials, const float scale, MeshStreams& out);

} // namespace helpers
