// helpers::LoadScene: JSON scene files -> rt::Scene + rt::Camera.
//
// What the format IS comes from the reference's Demo (Demo/SceneLoader.cpp: property names, defaults, which properties are required,
// rapidjson's integer-vs-real strictness for scalar properties -- `"radius": 1` is refused, `"radius": 1.0` is not --, the section order
// textures -> materials -> objects -> lights -> camera).  How it is read here is this file's own: the format is written down as DATA --
// one property table per record kind (kMaterialSchema, kCameraSchema) and one registry per polymorphic section (kTextureKinds,
// kShapeKinds, kLightKinds: "type" string -> builder) -- and a small interpreter (Record) walks a JSON object against a table, keeping a
// breadcrumb ("materials[3].roughness") for its diagnostics.  Adding a property is a table row; adding a light kind is a registry row.
// Not ingested (LoadScene fails loudly): CSG shapes, texture-shaped bokeh; area-light textures are ignored with a warning (they are
// commented out in the reference's AreaLight too).
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
#include <stdarg.h>
#include <stdio.h>

#include <functional>

RAYLIB_API Options gOptions;

namespace helpers {

using namespace rt;
using namespace rt::math;
using json::Value;

namespace {

using TextureTable = std::map<std::string, TexturePtr>;

// What a scene file's sections have produced so far; builders look names up here.
struct LoadState
{
    TextureTable textures;
    MaterialsMap materials;
};

enum class Need { Optional, Required };

// ---- Record: one JSON object being read, with the path that led to it --------------------------------------------------------------
class Record
{
public:
    Record(const Value& value, std::string where) : mValue(value), mWhere(std::move(where)) {}

    bool IsStructure() const { return mValue.IsObject(); }
    bool Has(const char* key) const { return mValue.HasMember(key); }
    const Value& Raw(const char* key) const { return mValue[key]; }
    Record Child(const char* key) const { return Record(mValue[key], mWhere + "." + key); }
    const std::string& Where() const { return mWhere; }

    bool Complain(const char* format, ...) const
    {
        va_list args;
        va_start(args, format);
        fprintf(stderr, "[rt] ERROR: %s: ", mWhere.c_str());
        vfprintf(stderr, format, args);
        fprintf(stderr, "\n");
        va_end(args);
        return false;
    }

    // Every reader: true = the record is still good (value read, or absent and optional); false = diagnosed.
    bool Absent(const char* key, Need need) const { return need == Need::Optional ? true : Complain("property '%s' is required", key); }

    bool Flag(const char* key, Need need, bool& out) const
    {
        if (!Has(key)) return Absent(key, need);
        if (!mValue[key].IsBool()) return Complain("'%s' must be true or false", key);
        out = mValue[key].GetBool();
        return true;
    }
    // scalar: only a number written as a real (rapidjson's IsFloat, which the reference's loader insists on)
    bool Scalar(const char* key, Need need, float& out) const
    {
        if (!Has(key)) return Absent(key, need);
        if (!mValue[key].IsFloat()) return Complain("'%s' must be a real number (write 1.0, not 1)", key);
        out = mValue[key].GetFloat();
        return true;
    }
    bool Integer(const char* key, Need need, int& out) const
    {
        if (!Has(key)) return Absent(key, need);
        if (!mValue[key].IsInt()) return Complain("'%s' must be an integer", key);
        out = mValue[key].GetInt();
        return true;
    }
    // N-component vector: an array of exactly N numbers (integers are fine here, as in the reference); lanes past N become 0
    bool Vector(const char* key, Need need, size_t n, Vector4& out) const
    {
        if (!Has(key)) return Absent(key, need);
        const Value& a = mValue[key];
        if (!a.IsArray() || a.Size() != n) return Complain("'%s' must be an array of %u numbers", key, (unsigned)n);
        float lane[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
        for (size_t i = 0; i < n; ++i) lane[i] = a[i].GetFloat();
        out = Vector4(lane[0], lane[1], lane[2], lane[3]);
        return true;
    }
    bool Text(const char* key, Need need, std::string& out) const
    {
        if (!Has(key)) return Absent(key, need);
        if (!mValue[key].IsString()) return Complain("'%s' must be a string", key);
        out = mValue[key].GetString();
        return true;
    }
    // "transform": { "translation": [x, y, z], "orientation": [pitch, yaw, roll] in degrees }, both optional
    bool Placement(const char* key, Transform& out) const
    {
        if (!Has(key)) return true;
        const Record t = Child(key);
        if (!t.IsStructure()) return t.Complain("must be a structure");
        Vector4 translation = Vector4::Zero(), degrees = Vector4::Zero();
        if (!t.Vector("translation", Need::Optional, 3, translation) || !t.Vector("orientation", Need::Optional, 3, degrees)) return false;
        degrees *= (RT_PI / 180.0f);
        out = Transform(translation, Quaternion::FromEulerAngles(degrees.ToFloat3()));
        return true;
    }
    // a texture by name: one declared in the file's "textures" section, else a bitmap file under the data path
    bool TextureRef(const char* key, const LoadState& state, TexturePtr& out) const
    {
        if (!Has(key)) return true;
        if (!mValue[key].IsString()) return Complain("'%s' must name a texture", key);
        const std::string name = mValue[key].GetString();
        const auto found = state.textures.find(name);
        out = found != state.textures.end() ? found->second : helpers::LoadTexture(gOptions.dataPath, name);
        return true;
    }
    bool MaterialRef(const char* key, const LoadState& state, MaterialPtr& out) const
    {
        if (!Has(key)) return true;
        if (!mValue[key].IsString()) return Complain("'%s' must name a material", key);
        const auto found = state.materials.find(mValue[key].GetString());
        if (found == state.materials.end()) return Complain("material '%s' is not declared", mValue[key].GetString());
        out = found->second;
        return true;
    }

private:
    const Value& mValue;
    std::string mWhere;
};

// ---- property tables: a row binds a JSON key to a member of the object being filled in ----------------------------------------------
template <typename T>
struct Property
{
    enum class Kind { Flag, Scalar, Color, Texture } kind;
    const char* key;
    std::function<void*(T&)> member;     // bool* / float* / Vector4* / TexturePtr* according to `kind`; null: parsed and dropped
};

template <typename T, size_t N>
static bool ApplySchema(const Record& record, const Property<T> (&schema)[N], T& target, const LoadState& state)
{
    for (const Property<T>& p : schema)
    {
        bool flagSink = false; float scalarSink = 0.0f; Vector4 colorSink; TexturePtr textureSink;
        void* const where = p.member ? p.member(target) : nullptr;
        bool ok = true;
        switch (p.kind)
        {
        case Property<T>::Kind::Flag:    ok = record.Flag(p.key, Need::Optional, where ? *static_cast<bool*>(where) : flagSink); break;
        case Property<T>::Kind::Scalar:  ok = record.Scalar(p.key, Need::Optional, where ? *static_cast<float*>(where) : scalarSink); break;
        case Property<T>::Kind::Color:   ok = record.Vector(p.key, Need::Optional, 3, where ? *static_cast<Vector4*>(where) : colorSink); break;
        case Property<T>::Kind::Texture: ok = record.TextureRef(p.key, state, where ? *static_cast<TexturePtr*>(where) : textureSink); break;
        }
        if (!ok) return false;
    }
    return true;
}

using MP = Property<Material>;
static const MP kMaterialSchema[] = {
    { MP::Kind::Flag,    "dispersive",        nullptr },                                                  // spectral rendering only
    { MP::Kind::Color,   "baseColor",         [](Material& m) -> void* { return &m.baseColor.baseValue; } },
    { MP::Kind::Color,   "emissionColor",     [](Material& m) -> void* { return &m.emission.baseValue; } },
    { MP::Kind::Scalar,  "roughness",         [](Material& m) -> void* { return &m.roughness.baseValue; } },
    { MP::Kind::Scalar,  "metalness",         [](Material& m) -> void* { return &m.metalness.baseValue; } },
    { MP::Kind::Texture, "baseColorTexture",  [](Material& m) -> void* { return &m.baseColor.texture; } },
    { MP::Kind::Texture, "emissionTexture",   [](Material& m) -> void* { return &m.emission.texture; } },
    { MP::Kind::Texture, "roughnessTexture",  [](Material& m) -> void* { return &m.roughness.texture; } },
    { MP::Kind::Texture, "metalnessTexture",  [](Material& m) -> void* { return &m.metalness.texture; } },
    { MP::Kind::Texture, "normalMap",         [](Material& m) -> void* { return &m.normalMap; } },
    { MP::Kind::Texture, "maskMap",           nullptr },                                                  // read for validation; the renderer never asks for it
    { MP::Kind::Scalar,  "normalMapStrength", [](Material& m) -> void* { return &m.normalMapStrength; } },
    { MP::Kind::Scalar,  "IoR",               [](Material& m) -> void* { return &m.IoR; } },
    { MP::Kind::Scalar,  "K",                 [](Material& m) -> void* { return &m.K; } },
};

using CP = Property<rt::Camera>;
static const CP kCameraSchema[] = {
    { CP::Kind::Flag,   "enableDOF",          [](rt::Camera& c) -> void* { return &c.mDOF.enable; } },
    { CP::Kind::Scalar, "aperture",           [](rt::Camera& c) -> void* { return &c.mDOF.aperture; } },
    { CP::Kind::Scalar, "focalPlaneDistance", [](rt::Camera& c) -> void* { return &c.mDOF.focalPlaneDistance; } },
};

// ---- registries: "type" -> builder ---------------------------------------------------------------------------------------------------
template <typename Builder>
struct KindRow { const char* type; Builder build; };

template <typename Builder, size_t N>
static const Builder* FindKind(const KindRow<Builder> (&rows)[N], const std::string& type)
{
    for (const KindRow<Builder>& row : rows) if (type == row.type) return &row.build;
    return nullptr;
}

// the two colours checkerboard and noise textures share
static bool TwoColors(const Record& r, Vector4& a, Vector4& b)
{
    a = b = Vector4::Zero();
    return r.Vector("colorA", Need::Required, 3, a) && r.Vector("colorB", Need::Required, 3, b);
}

using TextureBuilder = TexturePtr (*)(const Record&, const LoadState&);
static const KindRow<TextureBuilder> kTextureKinds[] = {
    { "bitmap", [](const Record& r, const LoadState&) -> TexturePtr {
          std::string file;
          if (!r.Text("path", Need::Required, file)) return nullptr;
          const BitmapPtr bitmap = LoadBitmapObject(gOptions.dataPath, file);
          if (!bitmap || bitmap->GetWidth() == 0 || bitmap->GetHeight() == 0) return nullptr;
          return std::make_shared<BitmapTexture>(bitmap);
      } },
    { "checkerboard", [](const Record& r, const LoadState&) -> TexturePtr {
          Vector4 a, b;
          if (!TwoColors(r, a, b)) return nullptr;
          return std::make_shared<CheckerboardTexture>(a, b);
      } },
    { "noise", [](const Record& r, const LoadState&) -> TexturePtr {
          Vector4 a, b;
          int octaves = 1;
          if (!TwoColors(r, a, b) || !r.Integer("octaves", Need::Optional, octaves)) return nullptr;
          octaves = octaves < 1 ? 1 : (octaves > 20 ? 20 : octaves);
          return std::make_shared<NoiseTexture>(a, b, (uint32)octaves);
      } },
    { "mix", [](const Record& r, const LoadState& state) -> TexturePtr {
          TexturePtr a, b, weight;
          if (!r.TextureRef("textureA", state, a) || !r.TextureRef("textureB", state, b) || !r.TextureRef("weight", state, weight)) return nullptr;
          if (!a || !b || !weight) { r.Complain("a mix texture needs 'textureA', 'textureB' and 'weight'"); return nullptr; }
          return std::make_shared<MixTexture>(a, b, weight);
      } },
};

using ShapeBuilder = ShapePtr (*)(const Record&, LoadState&);
static const KindRow<ShapeBuilder> kShapeKinds[] = {
    { "sphere", [](const Record& r, LoadState&) -> ShapePtr {
          float radius = 1.0f;
          if (!r.Scalar("radius", Need::Required, radius)) return nullptr;
          return std::make_shared<SphereShape>(radius);
      } },
    { "box", [](const Record& r, LoadState&) -> ShapePtr {
          Vector4 size;
          if (!r.Vector("size", Need::Required, 3, size)) return nullptr;
          return std::make_shared<BoxShape>(size);
      } },
    { "rect", [](const Record& r, LoadState&) -> ShapePtr {
          Vector4 size(FLT_MAX), textureScale(1.0f);
          if (!r.Vector("size", Need::Required, 2, size) || !r.Vector("textureScale", Need::Optional, 2, textureScale)) return nullptr;
          return std::make_shared<RectShape>(size.ToFloat2(), textureScale.ToFloat2());
      } },
    { "mesh", [](const Record& r, LoadState& state) -> ShapePtr {
          std::string file;
          float scale = 1.0f;
          if (!r.Text("path", Need::Required, file) || !r.Scalar("scale", Need::Optional, scale)) return nullptr;
          return helpers::LoadMesh(gOptions.dataPath + file, state.materials, scale);    // the file's MTL materials join the table
      } },
    { "csg", [](const Record& r, LoadState&) -> ShapePtr { r.Complain("CSG shapes are not supported by the device path"); return nullptr; } },
};

static ShapePtr BuildShape(const Record& r, LoadState& state)
{
    std::string type;
    if (!r.Text("type", Need::Required, type)) return nullptr;
    if (type == "plane") type = "rect";     // the reference accepts both spellings
    const ShapeBuilder* build = FindKind(kShapeKinds, type);
    if (!build) { r.Complain("unknown shape type '%s'", type.c_str()); return nullptr; }
    return (*build)(r, state);
}

// a light builder gets the colour (required for every kind) already read
using LightBuilder = LightPtr (*)(const Record&, const Vector4& color, const LoadState&);
static const KindRow<LightBuilder> kLightKinds[] = {
    { "area", [](const Record& r, const Vector4& color, const LoadState&) -> LightPtr {
          if (!r.Has("shape")) { r.Complain("an area light needs a 'shape'"); return nullptr; }
          LoadState scratch;      // a mesh-shaped light keeps its MTL materials to itself
          const ShapePtr shape = BuildShape(r.Child("shape"), scratch);
          if (!shape) return nullptr;
          if (r.Has("texture")) fprintf(stderr, "[rt] warning: %s: area-light textures are not evaluated (nor by the reference, AreaLight.cpp:49-53)\n", r.Where().c_str());
          return std::make_unique<AreaLight>(shape, color);
      } },
    { "point", [](const Record&, const Vector4& color, const LoadState&) -> LightPtr { return std::make_unique<PointLight>(color); } },
    { "spot", [](const Record& r, const Vector4& color, const LoadState&) -> LightPtr {
          float degrees = 0.0f;
          if (!r.Scalar("angle", Need::Optional, degrees)) return nullptr;
          return std::make_unique<SpotLight>(color, degrees / 180.0f * RT_PI);
      } },
    { "directional", [](const Record& r, const Vector4& color, const LoadState&) -> LightPtr {
          float degrees = 0.0f;
          if (!r.Scalar("angle", Need::Optional, degrees)) return nullptr;
          return std::make_unique<DirectionalLight>(color, DegToRad(degrees));
      } },
    { "background", [](const Record& r, const Vector4& color, const LoadState& state) -> LightPtr {
          auto sky = std::make_unique<BackgroundLight>(color);
          if (!r.TextureRef("texture", state, sky->mTexture)) return nullptr;
          return sky;
      } },
    // (the reference also knows "sphere", builds the light and drops it -- a null light object in the scene, SceneLoader.cpp:586-596;
    //  refused here)
};

// ---- sections --------------------------------------------------------------------------------------------------------------------------
static bool ReadTexture(const Record& r, LoadState& state)
{
    if (!r.IsStructure()) return r.Complain("a texture must be a structure");
    std::string name, type;
    if (!r.Text("name", Need::Required, name) || !r.Text("type", Need::Required, type)) return false;
    if (name.empty() || type.empty()) return r.Complain("'name' and 'type' cannot be empty");
    const TextureBuilder* build = FindKind(kTextureKinds, type);
    if (!build) return r.Complain("unknown texture type '%s'", type.c_str());
    const TexturePtr texture = (*build)(r, state);
    if (!texture) return false;
    state.textures[name] = texture;     // a later texture of the same name replaces the earlier one
    return true;
}

static bool ReadMaterial(const Record& r, LoadState& state)
{
    if (!r.IsStructure()) return r.Complain("a material must be a structure");
    std::string name, bsdf = Material::DefaultBsdfName;
    if (!r.Text("name", Need::Required, name) || !r.Text("bsdf", Need::Optional, bsdf)) return false;
    if (name.empty()) return r.Complain("'name' cannot be empty");
    if (state.materials.count(name) != 0) return r.Complain("material '%s' is declared twice", name.c_str());
    const MaterialPtr material = Material::Create();
    material->debugName = name;
    material->SetBsdf(bsdf);
    if (!ApplySchema(r, kMaterialSchema, *material, state)) return false;
    material->Compile();
    state.materials[name] = material;
    return true;
}

static bool ReadObject(const Record& r, LoadState& state, Scene& scene)
{
    if (!r.IsStructure()) return r.Complain("an object must be a structure");
    const ShapePtr shape = BuildShape(r, state);
    if (!shape) return false;
    MaterialPtr material;
    Transform placement;
    if (!r.MaterialRef("material", state, material) || !r.Placement("transform", placement)) return false;
    ShapeSceneObjectPtr object = std::make_unique<ShapeSceneObject>(shape);
    object->SetDefaultMaterial(material);
    object->SetTransform(placement.ToMatrix4());
    scene.AddObject(std::move(object));
    return true;
}

static bool ReadLight(const Record& r, const LoadState& state, Scene& scene)
{
    if (!r.IsStructure()) return r.Complain("a light must be a structure");
    std::string type;
    Vector4 color;
    if (!r.Text("type", Need::Required, type) || !r.Vector("color", Need::Required, 3, color)) return false;
    const LightBuilder* build = FindKind(kLightKinds, type);
    if (!build) return r.Complain("unknown light type '%s'", type.c_str());
    LightPtr light = (*build)(r, color, state);
    Transform placement;
    if (!light || !r.Placement("transform", placement)) return false;
    auto object = std::make_unique<LightSceneObject>(std::move(light));
    object->SetTransform(placement.ToMatrix4());
    scene.AddObject(std::move(object));
    return true;
}

static bool ReadCamera(const Record& r, const LoadState& state, rt::Camera& camera)
{
    if (!r.IsStructure()) return r.Complain("the camera must be a structure");
    Transform placement;
    float fieldOfView = 60.0f;
    if (!r.Placement("transform", placement) || !r.Scalar("fieldOfView", Need::Optional, fieldOfView)) return false;
    camera.SetTransform(placement);
    camera.SetPerspective(1.0f, DegToRad(fieldOfView));
    if (!ApplySchema(r, kCameraSchema, camera, state)) return false;
    if (r.Has("bokehTexture")) return r.Complain("texture-shaped bokeh is not supported by the device path");
    return true;
}

// one top-level array section of the document; `each` sees every element with its breadcrumb
static bool ReadSection(const Value& document, const char* section, const std::function<bool(const Record&)>& each)
{
    if (!document.HasMember(section)) return true;
    const Value& list = document[section];
    if (!list.IsArray()) { fprintf(stderr, "[rt] ERROR: '%s' must be an array\n", section); return false; }
    for (size_t i = 0; i < list.Size(); ++i)
        if (!each(Record(list[i], std::string(section) + "[" + std::to_string(i) + "]"))) return false;
    return true;
}

static bool ReadWholeFile(const std::string& path, std::string& text)
{
    FILE* file = fopen(path.c_str(), "rb");
    if (!file) return false;
    char chunk[1 << 14];
    for (size_t got; (got = fread(chunk, 1, sizeof(chunk), file)) > 0;) text.append(chunk, got);
    fclose(file);
    return true;
}

} // namespace

bool LoadScene(const std::string& path, Scene& scene, rt::Camera& camera)
{
    std::string text, parseError;
    if (!ReadWholeFile(path, text)) { fprintf(stderr, "[rt] ERROR: cannot open scene file '%s'\n", path.c_str()); return false; }
    Value document;
    if (!json::Parse(text, document, parseError) || !document.IsObject())
    {
        fprintf(stderr, "[rt] ERROR: '%s' is not a JSON scene description: %s\n", path.c_str(), parseError.c_str());
        return false;
    }
    LoadState state;
    // the order matters: materials name textures, objects name materials (and meshes add their own), lights name textures
    if (!ReadSection(document, "textures", [&](const Record& r) { return ReadTexture(r, state); })) return false;
    if (!ReadSection(document, "materials", [&](const Record& r) { return ReadMaterial(r, state); })) return false;
    if (!ReadSection(document, "objects", [&](const Record& r) { return ReadObject(r, state, scene); })) return false;
    if (!ReadSection(document, "lights", [&](const Record& r) { return ReadLight(r, state, scene); })) return false;
    if (document.HasMember("camera") && !ReadCamera(Record(document["camera"], "camera"), state, camera)) return false;
    return true;
}

} // namespace helpers
