// Simple PBR material (host-side description).  Mirrors the public fields and SetBsdf/Compile of the
// reference's rt::Material (Core/Material/Material.h:25-117); the BSDF itself runs on the device, the
// host only records which of the nine kinds is selected and which textures modulate the parameters.
#pragma once

#include "../Math/Math.h"
#include "../Textures/Texture.h"

namespace rt {

template <typename T>
struct MaterialParameter
{
    T baseValue = T(1.0f);
    TexturePtr texture = nullptr;   // value = baseValue * texture->Evaluate(uv), MaterialParameter.h:22-32 (on the device)
    MaterialParameter() = default;
    MaterialParameter(const T v) : baseValue(v) {}
    MaterialParameter& operator=(const T v) { baseValue = v; return *this; }
};

class Material;
using MaterialPtr = std::shared_ptr<rt::Material>;

class RAYLIB_API Material
{
public:
    explicit Material(const char* debugName = "<unnamed>");

    static const char* DefaultBsdfName;
    static MaterialPtr Create();
    static const MaterialPtr& GetDefaultMaterial();

    std::string debugName;
    MaterialParameter<math::Vector4> emission = math::Vector4::Zero();
    MaterialParameter<math::Vector4> baseColor = math::Vector4(0.7f, 0.7f, 0.7f, 0.0f);
    MaterialParameter<float> roughness = 0.1f;
    MaterialParameter<float> metalness = 0.0f;
    float IoR = 1.5f;
    float K = 4.0f;
    TexturePtr normalMap = nullptr;    // Material::GetNormalVector, Material.cpp:120-138
    float normalMapStrength = 1.0f;

    // one of: null, diffuse, roughDiffuse, dielectric, roughDielectric, metal, roughMetal, plastic, roughPlastic
    void SetBsdf(const std::string& bsdfName);
    const std::string& GetBsdfName() const { return mBsdfName; }
    int GetBsdfKind() const { return mBsdfKind; }   // RtBsdf of include/rtgpu.h, -1 if unset

    // clamps emission to >= 0 and baseColor to [0, 1] (all four lanes), like the reference
    void Compile();

private:
    std::string mBsdfName;
    int mBsdfKind = -1;
};

} // namespace rt
