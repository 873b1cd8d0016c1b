// JSON scene files -> rt::Scene + rt::Camera, field by field as Demo/SceneLoader.cpp does it (same property names,
// same defaults, same required/optional rules, including rapidjson's integer-vs-real strictness in TryParseFloat).
// Not ingested: CSG shapes, texture-shaped bokeh, area-light textures (commented out in the reference's AreaLight too)
// -- LoadScene fails loudly on them.
#include "SceneLoader.h"
#include "Demo.h"
#include "Json.h"
#include "MeshLoader.h"

#include "../Core/Scene/Light/Light.h"
#include "../Core/Scene/Object/SceneObject_Shape.h"
#include "../Core/Scene/Object/SceneObject_Light.h"
#include "../Core/Shapes/BoxShape.h"
#include "../Core/Shapes/SphereShape.h"
#include "../Core/Shapes/RectShape.h"
#include "../Core/Textures/CheckerboardTexture.h"
#include "../Core/Textures/BitmapTexture.h"
#include "../Core/Textures/NoiseTexture.h"
#include "../Core/Textures/MixTexture.h"

#include <float.h>
#include <stdio.h>

RAYLIB_API Options gOptions;

namespace helpers {

using namespace rt;
using namespace rt::math;

using TexturesMap = std::map<std::string, TexturePtr>;
using json::Value;

#define LOAD_ERROR(...) do { fprintf(stderr, "[rt] ERROR: "); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } while (0)

static bool ParseVector2(const Value& value, Vector4& outVector)
{
    if (!value.IsArray()) { LOAD_ERROR("2D vector description must be an array"); return false; }
    if (value.Size() != 2) { LOAD_ERROR("Invalid array size for 2D vector"); return false; }
    outVector = Vector4(value[(size_t)0].GetFloat(), value[(size_t)1].GetFloat(), 0.0f, 0.0f);
    return true;
}

static bool ParseVector3(const Value& value, Vector4& outVector)
{
    if (!value.IsArray()) { LOAD_ERROR("3D vector description must be an array"); return false; }
    if (value.Size() != 3) { LOAD_ERROR("Invalid array size for 3D vector"); return false; }
    outVector = Vector4(value[(size_t)0].GetFloat(), value[(size_t)1].GetFloat(), value[(size_t)2].GetFloat(), 0.0f);
    return true;
}

static bool Missing(const char* name, bool optional)
{
    if (optional) return true;
    LOAD_ERROR("Missing '%s' property", name);
    return false;
}

static bool TryParseBool(const Value& value, const char* name, bool optional, bool& outValue)
{
    if (!value.HasMember(name)) return Missing(name, optional);
    if (!value[name].IsBool()) { LOAD_ERROR("Property '%s' must be a bool", name); return false; }
    outValue = value[name].GetBool();
    return true;
}

static bool TryParseFloat(const Value& value, const char* name, bool optional, float& outValue)
{
    if (!value.HasMember(name)) return Missing(name, optional);
    if (!value[name].IsFloat()) { LOAD_ERROR("Property '%s' must be a float", name); return false; }
    outValue = value[name].GetFloat();
    return true;
}

static bool TryParseVector2(const Value& value, const char* name, bool optional, Vector4& outValue)
{
    if (!value.HasMember(name)) return Missing(name, optional);
    return ParseVector2(value[name], outValue);
}

static bool TryParseVector3(const Value& value, const char* name, bool optional, Vector4& outValue)
{
    if (!value.HasMember(name)) return Missing(name, optional);
    return ParseVector3(value[name], outValue);
}

static bool TryParseTransform(const Value& parentValue, const char* name, Transform& outValue)
{
    if (!parentValue.HasMember(name)) return true;
    const Value& value = parentValue[name];
    if (!value.IsObject()) { LOAD_ERROR("Transform description must be a structure"); return false; }
    Vector4 translation = Vector4::Zero();
    if (!TryParseVector3(value, "translation", true, translation)) return false;
    Vector4 orientation = Vector4::Zero();
    if (!TryParseVector3(value, "orientation", true, orientation)) return false;
    orientation *= (RT_PI / 180.0f);
    outValue = Transform(translation, Quaternion::FromEulerAngles(orientation.ToFloat3()));
    return true;
}

static bool TryParseTextureName(const Value& value, const char* name, const TexturesMap& textures, TexturePtr& outValue)
{
    if (!value.HasMember(name)) return true;
    if (!value[name].IsString()) { LOAD_ERROR("Texture path '%s' must be a string", name); return false; }
    const char* textureName = value[name].GetString();
    const auto iter = textures.find(textureName);
    if (iter != textures.end()) { outValue = iter->second; return true; }
    outValue = helpers::LoadTexture(gOptions.dataPath, textureName);
    return true;
}

static bool TryParseMaterialName(const MaterialsMap& materials, const Value& value, const char* name, MaterialPtr& outValue)
{
    if (!value.HasMember(name)) return true;
    if (!value[name].IsString()) { LOAD_ERROR("Material name '%s' must be a string", name); return false; }
    const std::string materialName = value[name].GetString();
    const auto iter = materials.find(materialName);
    if (iter == materials.end()) { LOAD_ERROR("Material '%s' does not exist", materialName.c_str()); return false; }
    outValue = iter->second;
    return true;
}

static TexturePtr ParseTexture(const Value& value, const TexturesMap& textures, std::string& outName)
{
    if (!value.IsObject()) { LOAD_ERROR("Texture description must be a structure"); return nullptr; }
    if (!value.HasMember("name")) { LOAD_ERROR("Texture is missing 'name' field"); return nullptr; }
    const std::string name = value["name"].GetString();
    if (name.empty()) { LOAD_ERROR("Texture name cannot be empty"); return nullptr; }
    outName = name;
    if (!value.HasMember("type")) { LOAD_ERROR("Texture is missing 'type' field"); return nullptr; }
    const std::string type = value["type"].GetString();
    if (type.empty()) { LOAD_ERROR("Texture type cannot be empty"); return nullptr; }
    if (type == "bitmap")
    {
        if (!value.HasMember("path")) { LOAD_ERROR("Texture is missing 'path' field"); return nullptr; }
        const std::string path = value["path"].GetString();
        BitmapPtr bitmap = LoadBitmapObject(gOptions.dataPath, path);
        if (!bitmap || bitmap->GetWidth() == 0 || bitmap->GetHeight() == 0) return nullptr;
        return std::make_shared<BitmapTexture>(bitmap);
    }
    if (type == "checkerboard")
    {
        Vector4 colorA = Vector4::Zero(), colorB = Vector4::Zero();
        if (!TryParseVector3(value, "colorA", false, colorA)) return nullptr;
        if (!TryParseVector3(value, "colorB", false, colorB)) return nullptr;
        return std::make_shared<CheckerboardTexture>(colorA, colorB);
    }
    if (type == "noise")
    {
        Vector4 colorA = Vector4::Zero(), colorB = Vector4::Zero();
        if (!TryParseVector3(value, "colorA", false, colorA)) return nullptr;
        if (!TryParseVector3(value, "colorB", false, colorB)) return nullptr;
        int numOctaves = 1;
        if (value.HasMember("octaves"))
        {
            if (!value["octaves"].IsInt()) { LOAD_ERROR("Property 'octaves' must be an integer"); return nullptr; }
            numOctaves = value["octaves"].GetInt();
        }
        numOctaves = numOctaves < 1 ? 1 : (numOctaves > 20 ? 20 : numOctaves);
        return std::make_shared<NoiseTexture>(colorA, colorB, (uint32)numOctaves);
    }
    if (type == "mix")
    {
        TexturePtr texA, texB, texWeight;
        if (!TryParseTextureName(value, "textureA", textures, texA)) return nullptr;
        if (!TryParseTextureName(value, "textureB", textures, texB)) return nullptr;
        if (!TryParseTextureName(value, "weight", textures, texWeight)) return nullptr;
        if (!texA || !texB || !texWeight) { LOAD_ERROR("Mix texture '%s' needs textureA, textureB and weight", name.c_str()); return nullptr; }
        return std::make_shared<MixTexture>(texA, texB, texWeight);
    }
    LOAD_ERROR("Invalid texture type name: '%s'", type.c_str());
    return nullptr;
}

static MaterialPtr ParseMaterial(const Value& value, const TexturesMap& textures)
{
    if (!value.IsObject()) { LOAD_ERROR("Material description must be a structure"); return nullptr; }
    if (!value.HasMember("name")) { LOAD_ERROR("Material is missing 'name' field"); return nullptr; }
    const std::string name = value["name"].GetString();
    if (name.empty()) { LOAD_ERROR("Material name cannot be empty"); return nullptr; }
    std::string bsdfName = Material::DefaultBsdfName;
    if (value.HasMember("bsdf")) bsdfName = value["bsdf"].GetString();

    MaterialPtr material = Material::Create();
    material->debugName = name;
    material->SetBsdf(bsdfName);
    bool isDispersive = false;
    if (!TryParseBool(value, "dispersive", true, isDispersive)) return nullptr;   // spectral rendering only
    if (!TryParseVector3(value, "baseColor", true, material->baseColor.baseValue)) return nullptr;
    if (!TryParseVector3(value, "emissionColor", true, material->emission.baseValue)) return nullptr;
    if (!TryParseFloat(value, "roughness", true, material->roughness.baseValue)) return nullptr;
    if (!TryParseFloat(value, "metalness", true, material->metalness.baseValue)) return nullptr;
    if (!TryParseTextureName(value, "baseColorTexture", textures, material->baseColor.texture)) return nullptr;
    if (!TryParseTextureName(value, "emissionTexture", textures, material->emission.texture)) return nullptr;
    if (!TryParseTextureName(value, "roughnessTexture", textures, material->roughness.texture)) return nullptr;
    if (!TryParseTextureName(value, "metalnessTexture", textures, material->metalness.texture)) return nullptr;
    if (!TryParseTextureName(value, "normalMap", textures, material->normalMap)) return nullptr;
    TexturePtr maskMap;   // parsed like the reference; Material::GetMaskValue has no caller there
    if (!TryParseTextureName(value, "maskMap", textures, maskMap)) return nullptr;
    if (!TryParseFloat(value, "normalMapStrength", true, material->normalMapStrength)) return nullptr;
    if (!TryParseFloat(value, "IoR", true, material->IoR)) return nullptr;
    if (!TryParseFloat(value, "K", true, material->K)) return nullptr;
    material->Compile();
    return material;
}

static ShapePtr ParseShape(const Value& value, MaterialsMap& materials)
{
    if (!value.HasMember("type")) { LOAD_ERROR("Object is missing 'type' field"); return nullptr; }
    const std::string typeStr = value["type"].GetString();
    if (typeStr == "sphere")
    {
        float radius = 1.0f;
        if (!TryParseFloat(value, "radius", false, radius)) return nullptr;
        return std::make_shared<SphereShape>(radius);
    }
    if (typeStr == "box")
    {
        Vector4 size;
        if (!TryParseVector3(value, "size", false, size)) return nullptr;
        return std::make_shared<BoxShape>(size);
    }
    if (typeStr == "rect" || typeStr == "plane")
    {
        Vector4 size(FLT_MAX);
        if (!TryParseVector2(value, "size", false, size)) return nullptr;
        Vector4 textureScale(1.0f);
        if (!TryParseVector2(value, "textureScale", true, textureScale)) return nullptr;
        return std::make_shared<RectShape>(size.ToFloat2(), textureScale.ToFloat2());
    }
    if (typeStr == "mesh")
    {
        if (!value.HasMember("path")) { LOAD_ERROR("Missing 'path' property"); return nullptr; }
        if (!value["path"].IsString()) { LOAD_ERROR("Mesh path must be a string"); return nullptr; }
        float scale = 1.0f;
        if (!TryParseFloat(value, "scale", true, scale)) return nullptr;
        return helpers::LoadMesh(gOptions.dataPath + value["path"].GetString(), materials, scale);
    }
    if (typeStr == "csg") { LOAD_ERROR("CSG shapes are not supported by the device path"); return nullptr; }
    LOAD_ERROR("Unknown scene object type: '%s'", typeStr.c_str());
    return nullptr;
}

static bool ParseLight(const Value& value, Scene& scene, const TexturesMap& textures)
{
    if (!value.IsObject()) { LOAD_ERROR("Light description must be a structure"); return false; }
    if (!value.HasMember("type")) { LOAD_ERROR("Light is missing 'type' field"); return false; }
    Vector4 lightColor;
    if (!TryParseVector3(value, "color", false, lightColor)) return false;
    LightPtr light;
    const std::string typeStr = value["type"].GetString();
    if (typeStr == "area")
    {
        if (!value.HasMember("shape")) { LOAD_ERROR("Area light is missing 'shape' field"); return false; }
        MaterialsMap none;
        ShapePtr shape = ParseShape(value["shape"], none);
        if (!shape) return false;
        if (value.HasMember("texture")) { LOAD_ERROR("Area light textures are not evaluated (commented out in the reference's AreaLight.cpp:49-53, 136-144)"); }
        light = std::make_unique<AreaLight>(shape, lightColor);
    }
    else if (typeStr == "point") light = std::make_unique<PointLight>(lightColor);
    else if (typeStr == "spot")
    {
        float angle = 0.0f;
        if (!TryParseFloat(value, "angle", true, angle)) return false;
        const float angleRad = angle / 180.0f * RT_PI;
        light = std::make_unique<SpotLight>(lightColor, angleRad);
    }
    else if (typeStr == "directional")
    {
        float angle = 0.0f;
        if (!TryParseFloat(value, "angle", true, angle)) return false;
        light = std::make_unique<DirectionalLight>(lightColor, DegToRad(angle));
    }
    else if (typeStr == "background")
    {
        auto backgroundLight = std::make_unique<BackgroundLight>(lightColor);
        if (!TryParseTextureName(value, "texture", textures, backgroundLight->mTexture)) return false;
        light = std::move(backgroundLight);
    }
    else
    {
        // ("sphere" builds a light and drops it in the reference, leaving a null light object, SceneLoader.cpp:586-596)
        LOAD_ERROR("Unknown light type: '%s'", typeStr.c_str());
        return false;
    }
    auto lightObject = std::make_unique<LightSceneObject>(std::move(light));
    Transform transform;
    if (!TryParseTransform(value, "transform", transform)) return false;
    lightObject->SetTransform(transform.ToMatrix4());
    scene.AddObject(std::move(lightObject));
    return true;
}

static bool ParseObject(const Value& value, Scene& scene, MaterialsMap& materials)
{
    if (!value.IsObject()) { LOAD_ERROR("Object description must be a structure"); return false; }
    ShapePtr shape = ParseShape(value, materials);
    if (!shape) return false;
    ShapeSceneObjectPtr sceneObject = std::make_unique<ShapeSceneObject>(shape);
    MaterialPtr material;
    if (!TryParseMaterialName(materials, value, "material", material)) return false;
    sceneObject->SetDefaultMaterial(material);
    Transform transform;
    if (!TryParseTransform(value, "transform", transform)) return false;
    sceneObject->SetTransform(transform.ToMatrix4());
    scene.AddObject(std::move(sceneObject));
    return true;
}

static bool ParseCamera(const Value& value, rt::Camera& camera)
{
    if (!value.IsObject()) { LOAD_ERROR("Light description must be a structure"); return false; }
    Transform transform;
    if (!TryParseTransform(value, "transform", transform)) return false;
    float fov = 60.0f;
    if (!TryParseFloat(value, "fieldOfView", true, fov)) return false;
    camera.SetTransform(transform);
    camera.SetPerspective(1.0f, DegToRad(fov));
    if (!TryParseBool(value, "enableDOF", true, camera.mDOF.enable)) return false;
    if (!TryParseFloat(value, "aperture", true, camera.mDOF.aperture)) return false;
    if (!TryParseFloat(value, "focalPlaneDistance", true, camera.mDOF.focalPlaneDistance)) return false;
    if (value.HasMember("bokehTexture")) { LOAD_ERROR("Texture-shaped bokeh is not supported by the device path"); return false; }
    return true;
}

template <typename F>
static bool ForEachElement(const Value& document, const char* name, F&& f)
{
    if (!document.HasMember(name)) return true;
    const Value& array = document[name];
    if (!array.IsArray()) { LOAD_ERROR("'%s' is expected to be an array", name); return false; }
    for (size_t i = 0; i < array.Size(); i++) if (!f(array[i])) return false;
    return true;
}

bool LoadScene(const std::string& path, Scene& scene, rt::Camera& camera)
{
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) { LOAD_ERROR("Failed to open file: %s", path.c_str()); return false; }
    std::string text;
    char readBuffer[4096];
    size_t n;
    while ((n = fread(readBuffer, 1, sizeof(readBuffer), fp)) > 0) text.append(readBuffer, n);
    fclose(fp);

    Value d;
    std::string error;
    if (!json::Parse(text, d, error) || !d.IsObject())
    {
        LOAD_ERROR("Failed to parse scene file '%s': %s", path.c_str(), error.c_str());
        return false;
    }

    MaterialsMap materialsMap;
    TexturesMap texturesMap;
    if (!ForEachElement(d, "textures", [&](const Value& v) {
            std::string name;
            const TexturePtr texture = ParseTexture(v, texturesMap, name);
            if (!texture) return false;
            texturesMap[name] = texture;
            return true; })) return false;
    if (!ForEachElement(d, "materials", [&](const Value& v) {
            const MaterialPtr material = ParseMaterial(v, texturesMap);
            if (!material) return false;
            if (materialsMap.count(material->debugName) > 0) { LOAD_ERROR("Duplicated material: '%s'", material->debugName.c_str()); return false; }
            materialsMap[material->debugName] = material;
            return true; })) return false;
    if (!ForEachElement(d, "objects", [&](const Value& v) { return ParseObject(v, scene, materialsMap); })) return false;
    if (!ForEachElement(d, "lights", [&](const Value& v) { return ParseLight(v, scene, texturesMap); })) return false;
    if (d.HasMember("camera") && !ParseCamera(d["camera"], camera)) return false;
    return true;
}

} // namespace helpers
