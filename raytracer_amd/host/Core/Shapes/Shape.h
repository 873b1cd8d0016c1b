// Shapes (host-side descriptions).  Mirrors the constructors of the reference's IShape family
// (Core/Shapes/*.h); intersection / sampling / frame evaluation run on the device.
#pragma once

#include "../Math/Math.h"
#include "../BVH/BVH.h"
#include "Mesh/VertexBufferDesc.h"

namespace rt {

class IShape
{
public:
    enum class Kind : uint32 { Sphere = 0, Box = 1, Rect = 2, Mesh = 3 };   // == RtShapeKind
    virtual ~IShape() = default;
    virtual Kind GetKind() const = 0;
    virtual const math::Box GetBoundingBox() const = 0;   // local space
    virtual float GetSurfaceArea() const { return 0.0f; }
    // shapeParam / shapeParam2 of RtObject / RtLight
    virtual void GetParams(float param[4], float param2[4]) const = 0;
};
using ShapePtr = std::shared_ptr<IShape>;

class RAYLIB_API SphereShape : public IShape
{
public:
    explicit SphereShape(const float radius);
    Kind GetKind() const override { return Kind::Sphere; }
    const math::Box GetBoundingBox() const override;
    float GetSurfaceArea() const override;
    void GetParams(float param[4], float param2[4]) const override;
private:
    float mRadius, mInvRadius;
};

class RAYLIB_API BoxShape : public IShape
{
public:
    explicit BoxShape(const math::Vector4& size);   // half extents
    Kind GetKind() const override { return Kind::Box; }
    const math::Box GetBoundingBox() const override;
    float GetSurfaceArea() const override;
    void GetParams(float param[4], float param2[4]) const override;
private:
    math::Vector4 mSize, mInvSize;
};

class RAYLIB_API RectShape : public IShape
{
public:
    explicit RectShape(const math::Float2 size = math::Float2(FLT_MAX), const math::Float2 texScale = math::Float2(1.0f));
    Kind GetKind() const override { return Kind::Rect; }
    const math::Box GetBoundingBox() const override;
    float GetSurfaceArea() const override;
    void GetParams(float param[4], float param2[4]) const override;
private:
    math::Float2 mSize, mTextureScale;
};

struct MeshDesc
{
    VertexBufferDesc vertexBufferDesc;
    std::string path;
};

// Triangle mesh with its own BVH.  Initialize() copies the borrowed arrays, builds the SAH BVH over the
// triangles and reorders them into BVH leaf order (reference: Core/Shapes/MeshShape.cpp:34-112,
// Core/Shapes/Mesh/VertexBuffer.cpp:52-186).
class RAYLIB_API MeshShape : public IShape
{
public:
    struct Triangle { float v0[3], edge1[3], edge2[3]; };                 // == RtTriangle
    struct VertexIndices { uint32 i0, i1, i2, materialIndex; };           // mesh-local material index
    struct VertexShading { float normal[3], tangent[3], texCoord[2]; };   // == RtVertexShading

    MeshShape();
    ~MeshShape() override;
    bool Initialize(const MeshDesc& desc);

    Kind GetKind() const override { return Kind::Mesh; }
    const math::Box GetBoundingBox() const override { return mBoundingBox; }
    void GetParams(float param[4], float param2[4]) const override;

    const BVH& GetBVH() const { return mBVH; }
    const std::vector<Triangle>& GetTriangles() const { return mTriangles; }
    const std::vector<VertexIndices>& GetVertexIndices() const { return mIndices; }
    const std::vector<VertexShading>& GetVertexShading() const { return mShading; }
    const std::vector<MaterialPtr>& GetMaterials() const { return mMaterials; }

private:
    BVH mBVH;
    math::Box mBoundingBox;
    std::vector<Triangle> mTriangles;
    std::vector<VertexIndices> mIndices;
    std::vector<VertexShading> mShading;
    std::vector<MaterialPtr> mMaterials;
};

using MeshShapePtr = std::shared_ptr<MeshShape>;

} // namespace rt
