// Borrowed-array description of a triangle mesh (reference: Core/Shapes/Mesh/VertexBufferDesc.h).
#pragma once

#include "../../Math/Math.h"

namespace rt {

class Material;
using MaterialPtr = std::shared_ptr<rt::Material>;

struct VertexBufferDesc
{
    uint32 numVertices = 0;
    uint32 numTriangles = 0;
    uint32 numMaterials = 0;

    const uint32* vertexIndexBuffer = nullptr;   // 3 per triangle
    const math::Float3* positions = nullptr;
    const math::Float3* normals = nullptr;
    const math::Float3* tangents = nullptr;
    const math::Float2* texCoords = nullptr;
    const uint32* materialIndexBuffer = nullptr; // 1 per triangle, UINT32_MAX => object's default material
    const MaterialPtr* materials = nullptr;
};

} // namespace rt
