// Scene objects: transform + payload (reference: Core/Scene/Object/SceneObject*.h).
#pragma once

#include "../../Math/Math.h"
#include "../../Shapes/Shape.h"
#include "../../Material/Material.h"
#include "../Light/Light.h"

namespace rt {

class RAYLIB_API ISceneObject
{
public:
    enum class Type : uint8 { Shape, Light, Decal };
    ISceneObject();
    virtual ~ISceneObject();
    virtual Type GetType() const = 0;
    virtual math::Box GetBoundingBox() const = 0;   // world space
    void SetTransform(const math::Matrix4& matrix);   // also computes the inverse
    const math::Matrix4& GetBaseTransform() const { return mTransform; }
    const math::Matrix4 GetTransform(const float t = 0.0f) const { (void)t; return mTransform; }
    const math::Matrix4 GetInverseTransform(const float t = 0.0f) const { (void)t; return mInverseTranform; }
private:
    math::Matrix4 mTransform;
    math::Matrix4 mInverseTranform;
};
using SceneObjectPtr = std::unique_ptr<ISceneObject>;

class RAYLIB_API ShapeSceneObject : public ISceneObject
{
public:
    explicit ShapeSceneObject(const ShapePtr& shape);
    Type GetType() const override { return Type::Shape; }
    math::Box GetBoundingBox() const override;
    void SetDefaultMaterial(const MaterialPtr& material);
    const MaterialPtr& GetDefaultMaterial() const { return mDefaultMaterial; }
    const ShapePtr& GetShape() const { return mShape; }
private:
    ShapePtr mShape;
    MaterialPtr mDefaultMaterial;
};
using ShapeSceneObjectPtr = std::unique_ptr<ShapeSceneObject>;

class RAYLIB_API LightSceneObject : public ISceneObject
{
public:
    explicit LightSceneObject(LightPtr light);
    Type GetType() const override { return Type::Light; }
    math::Box GetBoundingBox() const override;
    const ILight& GetLight() const { return *mLight; }
private:
    LightPtr mLight;
};
using LightSceneObjectPtr = std::unique_ptr<LightSceneObject>;

} // namespace rt
