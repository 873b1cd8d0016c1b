// Lights (host-side descriptions).  Mirrors the constructors, types and flags of the reference's ILight
// family (Core/Scene/Light/*.h); Illuminate / GetRadiance / TestRayHit run on the device.
#pragma once

#include "../../Math/Math.h"
#include "../../Shapes/Shape.h"
#include "../../Textures/Texture.h"

namespace rt {

class ILight
{
public:
    static constexpr float CosEpsilon = 0.9999f;
    enum class Type : uint8 { Area, Background, Directional, Point, Spot };   // == RtLightType
    enum Flags : uint8 { Flag_None = 0, Flag_IsFinite = 1 << 0, Flag_IsDelta = 1 << 1 };

    explicit ILight(const math::Vector4& color = math::Vector4(1.0f)) : mColor(color) {}
    virtual ~ILight() = default;
    const math::Vector4& GetColor() const { return mColor; }
    void SetColor(const math::Vector4& color) { mColor = color; }
    virtual Type GetType() const = 0;
    virtual const math::Box GetBoundingBox() const = 0;
    virtual Flags GetFlags() const = 0;
private:
    math::Vector4 mColor;
};
using LightPtr = std::unique_ptr<ILight>;

class RAYLIB_API AreaLight : public ILight
{
public:
    AreaLight(ShapePtr shape, const math::Vector4& color);
    const ShapePtr& GetShape() const { return mShape; }
    Type GetType() const override { return Type::Area; }
    const math::Box GetBoundingBox() const override { return mShape->GetBoundingBox(); }
    Flags GetFlags() const override { return Flag_IsFinite; }
private:
    ShapePtr mShape;
};

class RAYLIB_API BackgroundLight : public ILight
{
public:
    BackgroundLight() = default;
    explicit BackgroundLight(const math::Vector4& color) : ILight(color) {}
    TexturePtr mTexture = nullptr;   // environment map, BackgroundLight.cpp:45-61 (public member in the reference too)
    Type GetType() const override { return Type::Background; }
    const math::Box GetBoundingBox() const override { return math::Box::Full(); }
    Flags GetFlags() const override { return Flag_None; }
};

class RAYLIB_API DirectionalLight : public ILight
{
public:
    DirectionalLight() : DirectionalLight(math::Vector4(1.0f)) {}
    explicit DirectionalLight(const math::Vector4& color, const float angle = 0.2f);
    Type GetType() const override { return Type::Directional; }
    const math::Box GetBoundingBox() const override { return math::Box::Full(); }
    Flags GetFlags() const override { return mIsDelta ? Flag_IsDelta : Flag_None; }
    float GetCosAngle() const { return mCosAngle; }
    bool IsDelta() const { return mIsDelta; }
private:
    float mCosAngle;
    bool mIsDelta;
};

class RAYLIB_API PointLight : public ILight
{
public:
    explicit PointLight(const math::Vector4& color) : ILight(color) {}
    Type GetType() const override { return Type::Point; }
    const math::Box GetBoundingBox() const override { return math::Box(math::Vector4::Zero(), 0.0f); }
    Flags GetFlags() const override { return Flags(Flag_IsFinite | Flag_IsDelta); }
};

class RAYLIB_API SpotLight : public ILight
{
public:
    SpotLight(const math::Vector4& color, const float angle);
    Type GetType() const override { return Type::Spot; }
    const math::Box GetBoundingBox() const override { return math::Box(math::Vector4::Zero(), 0.0f); }
    Flags GetFlags() const override { return mIsDelta ? Flags(Flag_IsFinite | Flag_IsDelta) : Flag_IsFinite; }
    float GetCosAngle() const { return mCosAngle; }
    bool IsDelta() const { return mIsDelta; }
private:
    float mCosAngle;
    bool mIsDelta;
};

} // namespace rt
