// Materials, shapes, lights, scene objects and the Scene flattening.  Host side, one-time work per scene.
#include "../Core/Textures/MixTexture.h"
#include "../Core/Scene/Scene.h"
#include "../Core/BVH/BVHBuilder.h"

#include <stdio.h>
#include <functional>
#include <map>

namespace rt {

using namespace math;

// ---------------------------------------------------------------------------------------------------
// Material
// ---------------------------------------------------------------------------------------------------
const char* Material::DefaultBsdfName = "diffuse";

Material::Material(const char* name) : debugName(name) {}

MaterialPtr Material::Create() { return MaterialPtr(new Material); }

void Material::SetBsdf(const std::string& bsdfName)
{
    static const struct { const char* name; int kind; } kTable[] = {
        { "null", RT_BSDF_NULL }, { "diffuse", RT_BSDF_DIFFUSE }, { "roughDiffuse", RT_BSDF_ROUGH_DIFFUSE },
        { "dielectric", RT_BSDF_DIELECTRIC }, { "roughDielectric", RT_BSDF_ROUGH_DIELECTRIC },
        { "metal", RT_BSDF_METAL }, { "roughMetal", RT_BSDF_ROUGH_METAL },
        { "plastic", RT_BSDF_PLASTIC }, { "roughPlastic", RT_BSDF_ROUGH_PLASTIC },
    };
    for (const auto& e : kTable)
    {
        if (bsdfName == e.name) { mBsdfName = bsdfName; mBsdfKind = e.kind; return; }
    }
    fprintf(stderr, "[rt] ERROR: Unknown BSDF name: '%s'\n", bsdfName.c_str());   // reference logs and keeps the old BSDF
}

void Material::Compile()
{
    emission.baseValue = Vector4::Max(Vector4::Zero(), emission.baseValue);
    baseColor.baseValue = Vector4::Max(Vector4::Zero(), Vector4::Min(VECTOR_ONE, baseColor.baseValue));
}

const MaterialPtr& Material::GetDefaultMaterial()
{
    static MaterialPtr sDefault = [] {
        MaterialPtr m = std::make_shared<Material>("default");
        m->SetBsdf(Material::DefaultBsdfName);
        m->Compile();
        return m;
    }();
    return sDefault;
}

// ---------------------------------------------------------------------------------------------------
// Shapes
// ---------------------------------------------------------------------------------------------------
SphereShape::SphereShape(const float radius) : mRadius(radius), mInvRadius(1.0f / radius) {}
const Box SphereShape::GetBoundingBox() const { return Box(Vector4::Zero(), mRadius); }
float SphereShape::GetSurfaceArea() const { return 4.0f * RT_PI * Sqr(mRadius); }
void SphereShape::GetParams(float p[4], float p2[4]) const
{
    p[0] = mRadius; p[1] = mInvRadius; p[2] = 0.0f; p[3] = 0.0f;
    p2[0] = p2[1] = p2[2] = p2[3] = 0.0f;
}

BoxShape::BoxShape(const Vector4& size) : mSize(size), mInvSize(VECTOR_ONE / size)
{
    mSize.w = 0.0f;
    mInvSize.w = 0.0f;
}
const Box BoxShape::GetBoundingBox() const { return Box(-mSize, mSize); }
float BoxShape::GetSurfaceArea() const { return 8.0f * (mSize.x * (mSize.y + mSize.z) + mSize.y * mSize.z); }
void BoxShape::GetParams(float p[4], float p2[4]) const
{
    p[0] = mSize.x; p[1] = mSize.y; p[2] = mSize.z; p[3] = 0.0f;
    p2[0] = mInvSize.x; p2[1] = mInvSize.y; p2[2] = mInvSize.z; p2[3] = 0.0f;
}

RectShape::RectShape(const Float2 size, const Float2 texScale) : mSize(size), mTextureScale(texScale) {}
const Box RectShape::GetBoundingBox() const { return Box(Vector4(-mSize.x, -mSize.y, 0.0f, 0.0f), Vector4(mSize.x, mSize.y, 0.0f, 0.0f)); }
float RectShape::GetSurfaceArea() const { return 4.0f * mSize.x * mSize.y; }
void RectShape::GetParams(float p[4], float p2[4]) const
{
    p[0] = mSize.x; p[1] = mSize.y; p[2] = mTextureScale.x; p[3] = mTextureScale.y;
    p2[0] = p2[1] = p2[2] = p2[3] = 0.0f;
}

MeshShape::MeshShape() : mBoundingBox(Box::Empty()) {}
MeshShape::~MeshShape() = default;
void MeshShape::GetParams(float p[4], float p2[4]) const { for (int i = 0; i < 4; ++i) { p[i] = 0.0f; p2[i] = 0.0f; } }

bool MeshShape::Initialize(const MeshDesc& desc)
{
    const VertexBufferDesc& vb = desc.vertexBufferDesc;
    mBoundingBox = Box::Empty();
    mTriangles.clear(); mIndices.clear(); mShading.clear(); mMaterials.clear();
    if (vb.numTriangles == 0) return true;
    if (!vb.positions) { fprintf(stderr, "[rt] ERROR: Positions buffer must be provided\n"); return false; }
    if (!vb.vertexIndexBuffer) { fprintf(stderr, "[rt] ERROR: Index buffer must be provided\n"); return false; }

    std::vector<Box> boxes;
    boxes.reserve(vb.numTriangles);
    for (uint32 i = 0; i < vb.numTriangles; ++i)
    {
        for (int k = 0; k < 3; ++k)
        {
            if (vb.vertexIndexBuffer[3 * i + k] >= vb.numVertices) { fprintf(stderr, "[rt] ERROR: Vertex index out of bounds\n"); return false; }
        }
        const Vector4 v0(vb.positions[vb.vertexIndexBuffer[3 * i + 0]]);
        const Vector4 v1(vb.positions[vb.vertexIndexBuffer[3 * i + 1]]);
        const Vector4 v2(vb.positions[vb.vertexIndexBuffer[3 * i + 2]]);
        const Box triBox(v0, v1, v2);
        boxes.push_back(triBox);
        mBoundingBox = Box(mBoundingBox, triBox);
    }

    BVHBuilder::Indices newOrder;
    BVHBuilder builder(mBVH);
    if (!builder.Build(boxes.data(), vb.numTriangles, BvhBuildingParams(), newOrder)) return false;

    // triangles in BVH leaf order: precomputed v0 / edge1 / edge2 + vertex & material indices
    mTriangles.resize(vb.numTriangles);
    mIndices.resize(vb.numTriangles);
    for (uint32 i = 0; i < vb.numTriangles; ++i)
    {
        const uint32 src = newOrder[i];
        const uint32 i0 = vb.vertexIndexBuffer[3 * src + 0], i1 = vb.vertexIndexBuffer[3 * src + 1], i2 = vb.vertexIndexBuffer[3 * src + 2];
        const Vector4 v0(vb.positions[i0]), v1(vb.positions[i1]), v2(vb.positions[i2]);
        const Vector4 e1 = v1 - v0, e2 = v2 - v0;
        Triangle& t = mTriangles[i];
        t.v0[0] = v0.x; t.v0[1] = v0.y; t.v0[2] = v0.z;
        t.edge1[0] = e1.x; t.edge1[1] = e1.y; t.edge1[2] = e1.z;
        t.edge2[0] = e2.x; t.edge2[1] = e2.y; t.edge2[2] = e2.z;
        const uint32 mat = vb.materialIndexBuffer ? vb.materialIndexBuffer[src] : UINT32_MAX;
        if (mat != UINT32_MAX && mat >= vb.numMaterials) { fprintf(stderr, "[rt] ERROR: Material index out of bounds\n"); return false; }
        mIndices[i] = { i0, i1, i2, mat };
    }

    mShading.resize(vb.numVertices);
    for (uint32 i = 0; i < vb.numVertices; ++i)
    {
        VertexShading& s = mShading[i];
        const Float3 n = vb.normals ? vb.normals[i] : Float3();
        const Float3 t = vb.tangents ? vb.tangents[i] : Float3();
        const Float2 uv = vb.texCoords ? vb.texCoords[i] : Float2();
        s.normal[0] = n.x; s.normal[1] = n.y; s.normal[2] = n.z;
        s.tangent[0] = t.x; s.tangent[1] = t.y; s.tangent[2] = t.z;
        s.texCoord[0] = uv.x; s.texCoord[1] = uv.y;
    }

    mMaterials.assign(vb.materials, vb.materials + vb.numMaterials);
    return true;
}

// ---------------------------------------------------------------------------------------------------
// Lights
// ---------------------------------------------------------------------------------------------------
AreaLight::AreaLight(ShapePtr shape, const Vector4& color) : ILight(color), mShape(std::move(shape)) {}

DirectionalLight::DirectionalLight(const Vector4& color, const float angle) : ILight(color)
{
    mCosAngle = cosf(angle);
    mIsDelta = mCosAngle > CosEpsilon;
}

SpotLight::SpotLight(const Vector4& color, const float angle) : ILight(color)
{
    mCosAngle = cosf(angle);
    mIsDelta = mCosAngle > CosEpsilon;
}

// ---------------------------------------------------------------------------------------------------
// Scene objects
// ---------------------------------------------------------------------------------------------------
ISceneObject::ISceneObject() : mTransform(Matrix4::Identity()), mInverseTranform(Matrix4::Identity()) {}
ISceneObject::~ISceneObject() = default;

void ISceneObject::SetTransform(const Matrix4& matrix)
{
    mTransform = matrix;
    mInverseTranform = matrix.Inverse();
}

ShapeSceneObject::ShapeSceneObject(const ShapePtr& shape) : mShape(shape), mDefaultMaterial(Material::GetDefaultMaterial()) {}

Box ShapeSceneObject::GetBoundingBox() const
{
    const Box local = mShape->GetBoundingBox();
    return Box(GetBaseTransform().TransformBox(local), GetTransform(1.0f).TransformBox(local));
}

void ShapeSceneObject::SetDefaultMaterial(const MaterialPtr& material)
{
    mDefaultMaterial = material;
    if (!mDefaultMaterial) mDefaultMaterial = Material::GetDefaultMaterial();
}

LightSceneObject::LightSceneObject(LightPtr light) : mLight(std::move(light)) {}

Box LightSceneObject::GetBoundingBox() const
{
    const Box local = mLight->GetBoundingBox();
    return Box(GetBaseTransform().TransformBox(local), GetTransform(1.0f).TransformBox(local));
}

// ---------------------------------------------------------------------------------------------------
// Scene
// ---------------------------------------------------------------------------------------------------
Scene::Scene() { memset(&mDesc, 0, sizeof(mDesc)); mDesc.abiVersion = RTGPU_ABI_VERSION; }
Scene::~Scene() = default;
Scene::Scene(Scene&&) = default;
Scene& Scene::operator=(Scene&&) = default;

void Scene::AddObject(SceneObjectPtr object)
{
    if (object->GetType() == ISceneObject::Type::Light) mLights.push_back(static_cast<const LightSceneObject*>(object.get()));
    mAllObjects.push_back(std::move(object));
}

bool Scene::BuildBVH()
{
    // classification: finite lights and shapes are traceable, infinite lights are "global"
    mTraceableObjects.clear(); mLights.clear(); mGlobalLights.clear();
    for (const auto& object : mAllObjects)
    {
        if (object->GetType() == ISceneObject::Type::Light)
        {
            const LightSceneObject* lightObject = static_cast<const LightSceneObject*>(object.get());
            mLights.push_back(lightObject);
            if (lightObject->GetLight().GetFlags() & ILight::Flag_IsFinite) mTraceableObjects.push_back(object.get());
            else mGlobalLights.push_back(lightObject);
        }
        else if (object->GetType() == ISceneObject::Type::Shape)
        {
            mTraceableObjects.push_back(object.get());
        }
    }

    std::vector<Box> boxes;
    boxes.reserve(mTraceableObjects.size());
    for (const ISceneObject* obj : mTraceableObjects) boxes.push_back(obj->GetBoundingBox());

    BVHBuilder::Indices newOrder;
    BVHBuilder builder(mTraceableObjectsBVH);
    if (!builder.Build(boxes.data(), (uint32)mTraceableObjects.size(), BvhBuildingParams(), newOrder)) return false;

    std::vector<const ISceneObject*> reordered;
    reordered.reserve(mTraceableObjects.size());
    for (uint32 i = 0; i < mTraceableObjects.size(); ++i) reordered.push_back(mTraceableObjects[newOrder[i]]);
    mTraceableObjects = std::move(reordered);

    return Flatten();
}

static void CopyNodes(const BVH& bvh, std::vector<RtNode>& out)
{
    static_assert(sizeof(RtNode) == sizeof(BVH::Node), "node layouts must match");
    const size_t first = out.size();
    out.resize(first + bvh.GetNumNodes());
    if (bvh.GetNumNodes()) memcpy(out.data() + first, bvh.GetNodes(), sizeof(RtNode) * bvh.GetNumNodes());
}

static void FillMaterial(const Material& m, RtMaterial& out)
{
    memset(&out, 0, sizeof(out));
    out.baseColorTexture = out.emissionTexture = out.roughnessTexture = out.metalnessTexture = out.normalMapTexture = RT_NO_TEXTURE;
    out.normalMapStrength = m.normalMapStrength;
    memcpy(out.emission, &m.emission.baseValue, 16);
    memcpy(out.baseColor, &m.baseColor.baseValue, 16);
    out.roughness = m.roughness.baseValue;
    out.metalness = m.metalness.baseValue;
    out.IoR = m.IoR;
    out.K = m.K;
    out.bsdf = (uint32)(m.GetBsdfKind() < 0 ? RT_BSDF_DIFFUSE : m.GetBsdfKind());
}

bool Scene::Flatten()
{
    mFlatTopNodes.clear(); mFlatMeshNodes.clear(); mFlatObjects.clear(); mFlatLights.clear(); mFlatGlobalLights.clear();
    mFlatMaterials.clear(); mFlatMeshes.clear(); mFlatTriangles.clear(); mFlatVertexIndices.clear(); mFlatVertexShading.clear();

    mFlatTextures.clear(); mFlatTexels.clear();
    bool texturesOk = true;
    std::map<const ITexture*, uint32> textureIds;
    std::function<uint32(const TexturePtr&)> internTexture = [&](const TexturePtr& t) -> uint32 {
        if (!t) return RT_NO_TEXTURE;
        auto it = textureIds.find(t.get());
        if (it != textureIds.end()) return it->second;
        RtTexture flat;
        if (!t->Describe(flat, mFlatTexels)) { texturesOk = false; return RT_NO_TEXTURE; }
        if (const MixTexture* mix = dynamic_cast<const MixTexture*>(t.get()))   // children first: their indices go into the mix
        {
            flat.mixA = internTexture(mix->GetTextureA()); flat.mixB = internTexture(mix->GetTextureB()); flat.mixWeight = internTexture(mix->GetTextureMask());
        }
        const uint32 id = (uint32)mFlatTextures.size();
        mFlatTextures.push_back(flat);
        textureIds[t.get()] = id;
        return id;
    };
    std::map<const Material*, uint32> materialIds;
    auto internMaterial = [&](const Material* m) -> uint32 {
        auto it = materialIds.find(m);
        if (it != materialIds.end()) return it->second;
        const uint32 id = (uint32)mFlatMaterials.size();
        RtMaterial flat; FillMaterial(*m, flat);
        flat.baseColorTexture = internTexture(m->baseColor.texture);
        flat.emissionTexture = internTexture(m->emission.texture);
        flat.roughnessTexture = internTexture(m->roughness.texture);
        flat.metalnessTexture = internTexture(m->metalness.texture);
        flat.normalMapTexture = internTexture(m->normalMap);
        mFlatMaterials.push_back(flat);
        materialIds[m] = id;
        return id;
    };
    std::map<const MeshShape*, uint32> meshIds;
    auto internMesh = [&](const MeshShape* mesh) -> uint32 {
        auto it = meshIds.find(mesh);
        if (it != meshIds.end()) return it->second;
        RtMesh flat; memset(&flat, 0, sizeof(flat));
        flat.firstNode = (uint32)mFlatMeshNodes.size();
        flat.numNodes = mesh->GetBVH().GetNumNodes();
        flat.firstTriangle = (uint32)mFlatTriangles.size();
        flat.numTriangles = (uint32)mesh->GetTriangles().size();
        flat.firstVertex = (uint32)mFlatVertexShading.size();
        flat.numVertices = (uint32)mesh->GetVertexShading().size();
        CopyNodes(mesh->GetBVH(), mFlatMeshNodes);
        std::vector<uint32> localToGlobalMaterial;
        for (const MaterialPtr& m : mesh->GetMaterials()) localToGlobalMaterial.push_back(m ? internMaterial(m.get()) : RT_NO_MATERIAL);
        for (size_t i = 0; i < mesh->GetTriangles().size(); ++i)
        {
            RtTriangle t; memcpy(&t, &mesh->GetTriangles()[i], sizeof(t));
            mFlatTriangles.push_back(t);
            const MeshShape::VertexIndices& vi = mesh->GetVertexIndices()[i];
            RtVertexIndices fi = { vi.i0, vi.i1, vi.i2, vi.materialIndex == UINT32_MAX ? RT_NO_MATERIAL : localToGlobalMaterial[vi.materialIndex] };
            mFlatVertexIndices.push_back(fi);
        }
        for (const MeshShape::VertexShading& s : mesh->GetVertexShading())
        {
            RtVertexShading fs; memcpy(&fs, &s, sizeof(fs));
            mFlatVertexShading.push_back(fs);
        }
        const uint32 id = (uint32)mFlatMeshes.size();
        mFlatMeshes.push_back(flat);
        meshIds[mesh] = id;
        return id;
    };

    // lights, in mLights order
    std::map<const LightSceneObject*, uint32> lightIds;
    for (const LightSceneObject* lo : mLights)
    {
        const ILight& light = lo->GetLight();
        RtLight L; memset(&L, 0, sizeof(L));
        lo->GetTransform().Store(L.transform);
        lo->GetInverseTransform().Store(L.invTransform);
        memcpy(L.color, &light.GetColor(), 16);
        L.type = (uint32)light.GetType();
        L.flags = (uint32)light.GetFlags();
        L.texture = RT_NO_TEXTURE;
        if (light.GetType() == ILight::Type::Background) L.texture = internTexture(static_cast<const BackgroundLight&>(light).mTexture);
        if (light.GetType() == ILight::Type::Area)
        {
            const AreaLight& al = static_cast<const AreaLight&>(light);
            if (al.GetShape()->GetKind() == IShape::Kind::Mesh) { fprintf(stderr, "[rt] ERROR: mesh area lights are not supported (MeshShape::Sample is 'Not implemented yet' in the reference too)\n"); return false; }
            L.shapeKind = (uint32)al.GetShape()->GetKind();
            al.GetShape()->GetParams(L.shapeParam, L.shapeParam2);
        }
        else if (light.GetType() == ILight::Type::Directional)
        {
            const DirectionalLight& dl = static_cast<const DirectionalLight&>(light);
            L.cosAngle = dl.GetCosAngle(); L.isDelta = dl.IsDelta() ? 1u : 0u;
        }
        else if (light.GetType() == ILight::Type::Spot)
        {
            const SpotLight& sl = static_cast<const SpotLight&>(light);
            L.cosAngle = sl.GetCosAngle(); L.isDelta = sl.IsDelta() ? 1u : 0u;
        }
        lightIds[lo] = (uint32)mFlatLights.size();
        mFlatLights.push_back(L);
    }
    for (const LightSceneObject* lo : mGlobalLights) mFlatGlobalLights.push_back(lightIds[lo]);

    // traceable objects in BVH leaf order
    for (const ISceneObject* obj : mTraceableObjects)
    {
        RtObject O; memset(&O, 0, sizeof(O));
        obj->GetTransform().Store(O.transform);
        obj->GetInverseTransform().Store(O.invTransform);
        O.materialIndex = RT_NO_MATERIAL;
        if (obj->GetType() == ISceneObject::Type::Light)
        {
            const LightSceneObject* lo = static_cast<const LightSceneObject*>(obj);
            O.objectKind = RT_OBJECT_LIGHT;
            O.lightIndex = lightIds[lo];
            const RtLight& L = mFlatLights[O.lightIndex];
            O.shapeKind = L.shapeKind;
            memcpy(O.shapeParam, L.shapeParam, 16); memcpy(O.shapeParam2, L.shapeParam2, 16);
        }
        else
        {
            const ShapeSceneObject* so = static_cast<const ShapeSceneObject*>(obj);
            O.objectKind = RT_OBJECT_SHAPE;
            O.shapeKind = (uint32)so->GetShape()->GetKind();
            so->GetShape()->GetParams(O.shapeParam, O.shapeParam2);
            O.materialIndex = internMaterial(so->GetDefaultMaterial().get());
            if (so->GetShape()->GetKind() == IShape::Kind::Mesh) O.meshIndex = internMesh(static_cast<const MeshShape*>(so->GetShape().get()));
        }
        mFlatObjects.push_back(O);
    }

    CopyNodes(mTraceableObjectsBVH, mFlatTopNodes);

    memset(&mDesc, 0, sizeof(mDesc));
    mDesc.abiVersion = RTGPU_ABI_VERSION;
    mDesc.numObjects = (uint32)mFlatObjects.size();
    mDesc.numTopNodes = (uint32)mFlatTopNodes.size();
    mDesc.numLights = (uint32)mFlatLights.size();
    mDesc.numGlobalLights = (uint32)mFlatGlobalLights.size();
    mDesc.numMaterials = (uint32)mFlatMaterials.size();
    mDesc.numMeshes = (uint32)mFlatMeshes.size();
    mDesc.numMeshNodes = (uint32)mFlatMeshNodes.size();
    mDesc.numTriangles = (uint32)mFlatTriangles.size();
    mDesc.numVertices = (uint32)mFlatVertexShading.size();
    mDesc.numTextures = (uint32)mFlatTextures.size();
    mDesc.textures = mFlatTextures.data();
    mDesc.texelData = mFlatTexels.data();
    mDesc.texelBytes = mFlatTexels.size();
    if (!texturesOk) return false;
    mDesc.topNodes = mFlatTopNodes.data();
    mDesc.objects = mFlatObjects.data();
    mDesc.lights = mFlatLights.data();
    mDesc.globalLights = mFlatGlobalLights.data();
    mDesc.materials = mFlatMaterials.data();
    mDesc.meshes = mFlatMeshes.data();
    mDesc.meshNodes = mFlatMeshNodes.data();
    mDesc.triangles = mFlatTriangles.data();
    mDesc.vertexIndices = mFlatVertexIndices.data();
    mDesc.vertexShading = mFlatVertexShading.data();
    mDesc.blueNoise = nullptr;
    mBuildId++;
    return true;
}

} // namespace rt
