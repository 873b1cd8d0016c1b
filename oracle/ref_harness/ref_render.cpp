// oracle/ref_harness/ref_render.cpp -- TEST / MEASUREMENT INFRASTRUCTURE, built in the build container only (the binary travels to the
// GPU box, the reference's sources do not).
//
// A headless driver over the REFERENCE'S OWN compiled translation units (oracle/_ref/libref_partial.a: every file under
// /root/reference/Core that builds on Linux as shipped, compiled unmodified from where it lies with the reference's AVX2 / FMA flags):
// rt::Viewport::Render (Core/Rendering/Viewport.cpp:185-289) -> rt::ThreadPool (Core/Utils/ThreadPool.cpp) ->
// rt::PathTracerMIS::RenderPixel (Core/Rendering/PathTracerMIS.cpp:254-415) -> GenericTraverse (Core/Traversal/Traversal_Single.h,
// instantiated HERE from the reference's header) -> MeshShape / shapes / lights / materials / BSDFs / samplers / SSE-AVX math: all the
// reference's machine code.  It is used
//   (1) as the CPU baseline of bench.py ("cpu_baseline.kind": "reference-partial"): the reference's AVX path timed on the GPU box's
//       host cores, numThreads = hardware threads and 1, on the same scene the GPU renders;
//   (2) to produce image-level golden statistics of the real integrator (tests/golden/ref_render_*.bin).
//
// What is NOT the reference's object code: three of its translation units do not compile here -- Scene/Scene.cpp and
// Rendering/Renderer.cpp include <Windows.h> through Utils/Profiler.h, Utils/MemoryHelpers.cpp needs <intrin.h> -- and no stand-in
// headers are written for them.  The handful of member functions they define are supplied below as GLUE, written against the
// reference's own headers: the scene's object list and its top-level BVH build, the 0 / 1 / N-object dispatch around GenericTraverse,
// the world -> object ray transform, the tangent-frame assembly of Scene::EvaluateIntersection, the empty IRenderer base-class hooks
// (LargeMemCopy and the allocator entry points: ref_glue.cpp).  No decals, no packet traversal (neither is on the measured path).  Per ray that is a few dozen
// instructions of dispatch around the reference's traversal, intersection and shading code; the baseline is labelled accordingly and
// no bit-level parity is claimed from this binary.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <memory>
#include <chrono>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <atomic>
#include <functional>
#include <algorithm>
#include <map>
#include <unordered_map>
#include <array>
#include <limits>
#include <cmath>
#include <sstream>
#include <fstream>
#include <iostream>
#include <set>
#include <deque>
#include <queue>
#include <list>
#include <future>
#include <type_traits>
#include <initializer_list>
#include <unistd.h>

// test-only access to private state (the generators' seeds); layouts are unchanged
#define private public
#define protected public
#include "PCH.h"
#include "Math/Math.h"
#include "Math/Vector4.h"
#include "Math/Random.h"
#include "Math/Transform.h"
#include "Math/SamplingHelpers.h"
#include "Math/Quaternion.h"
#include "Scene/Scene.h"
#include "Scene/Camera.h"
#include "Scene/Light/AreaLight.h"
#include "Scene/Light/BackgroundLight.h"
#include "Scene/Light/DirectionalLight.h"
#include "Scene/Light/PointLight.h"
#include "Scene/Light/SpotLight.h"
#include "Scene/Object/SceneObject_Shape.h"
#include "Scene/Object/SceneObject_Light.h"
#include "Scene/Object/SceneObject_Decal.h"
#include "Shapes/SphereShape.h"
#include "Shapes/BoxShape.h"
#include "Shapes/RectShape.h"
#include "Shapes/MeshShape.h"
#include "Material/Material.h"
#include "Textures/BitmapTexture.h"
#include "Utils/Bitmap.h"
#include "Rendering/Viewport.h"
#include "Rendering/Renderer.h"
#include "Rendering/PathTracerMIS.h"
#include "Rendering/PathDebugging.h"
#include "Rendering/ShadingData.h"
#include "Rendering/Context.h"
#include "BVH/BVHBuilder.h"
#include "Traversal/TraversalContext.h"
#include "Traversal/Traversal_Single.h"
#include "Utils/MemoryHelpers.h"
#undef private
#undef protected

using namespace rt;
using namespace rt::math;

// =====================================================================================================================================
// GLUE for Scene/Scene.cpp, Rendering/Renderer.cpp and Utils/MemoryHelpers.cpp (see the header comment)
// =====================================================================================================================================
namespace rt {

// Rendering/Renderer.cpp:10-43: the base class does nothing
IRenderer::IRenderer(const Scene& scene) : mScene(scene) {}
IRenderer::~IRenderer() {}
RendererContextPtr IRenderer::CreateContext() const { return RendererContextPtr(); }
void IRenderer::PreRender(uint32, const Film&) {}
void IRenderer::PreRender(uint32, RenderingContext&) {}
void IRenderer::PreRenderGlobal(RenderingContext&) {}
void IRenderer::PreRenderGlobal() {}
void IRenderer::Raytrace_Packet(RayPacket&, const Camera&, Film&, RenderingContext&) const {}

// Scene/Scene.cpp
Scene::Scene() {}
Scene::~Scene() {}

void Scene::AddObject(SceneObjectPtr object) { mAllObjects.PushBack(std::move(object)); }   // (:26-34; the light list is rebuilt by BuildBVH)

// :36-126 -- what is traceable, what is a global light, the top-level BVH over the traceable objects' boxes
bool Scene::BuildBVH()
{
    mTraceableObjects.Clear(); mLights.Clear(); mGlobalLights.Clear(); mDecals.Clear();
    for (const SceneObjectPtr& owned : mAllObjects)
    {
        const ISceneObject* object = owned.get();
        switch (object->GetType())
        {
        case ISceneObject::Type::Light:
        {
            const LightSceneObject* lightObject = static_cast<const LightSceneObject*>(object);
            mLights.PushBack(lightObject);
            if (lightObject->GetLight().GetFlags() & ILight::Flag_IsFinite) mTraceableObjects.PushBack(static_cast<const ITraceableSceneObject*>(object));
            else mGlobalLights.PushBack(lightObject);
            break;
        }
        case ISceneObject::Type::Shape: mTraceableObjects.PushBack(static_cast<const ITraceableSceneObject*>(object)); break;
        default: fprintf(stderr, "ref_render: decals are not supported by the glue\n"); return false;
        }
    }
    DynArray<Box> boxes;
    for (const ITraceableSceneObject* object : mTraceableObjects) boxes.PushBack(object->GetBoundingBox());
    BVHBuilder::Indices order;
    BVHBuilder builder(mTraceableObjectsBVH);
    if (!builder.Build(boxes.Data(), mTraceableObjects.Size(), BvhBuildingParams(), order)) return false;
    DynArray<const ITraceableSceneObject*> reordered;
    for (uint32 i = 0; i < mTraceableObjects.Size(); ++i) reordered.PushBack(mTraceableObjects[order[i]]);
    mTraceableObjects = std::move(reordered);
    return true;
}

// :128-165 -- a ray enters an object in the object's space, direction NOT renormalised (TransformRay_Unsafe)
void Scene::Traverse_Object(const SingleTraversalContext& context, const uint32 objectID) const
{
    const ITraceableSceneObject* object = mTraceableObjects[objectID];
    const Ray local = object->GetInverseTransform(context.context.time).TransformRay_Unsafe(context.ray);
    const SingleTraversalContext inner = { local, context.hitPoint, context.context };
    object->Traverse(inner, objectID);
}
bool Scene::Traverse_Object_Shadow(const SingleTraversalContext& context, const uint32 objectID) const
{
    const ITraceableSceneObject* object = mTraceableObjects[objectID];
    Ray local = object->GetInverseTransform(context.context.time).TransformRay_Unsafe(context.ray);
    local.originDivDir = local.origin * local.invDir;
    const SingleTraversalContext inner = { local, context.hitPoint, context.context };
    return object->Traverse_Shadow(inner);
}
// :167-195 -- a leaf of the top-level BVH is a run of objects
void Scene::Traverse_Leaf(const SingleTraversalContext& context, const uint32, const BVH::Node& node) const
{
    for (uint32 k = 0, n = node.numLeaves, first = node.childIndex; k < n; ++k) Traverse_Object(context, first + k);
}
bool Scene::Traverse_Leaf_Shadow(const SingleTraversalContext& context, const BVH::Node& node) const
{
    for (uint32 k = 0, n = node.numLeaves, first = node.childIndex; k < n; ++k) if (Traverse_Object_Shadow(context, first + k)) return true;
    return false;
}
// :219-261 -- nothing / the single object directly / the reference's GenericTraverse over the top-level tree
void Scene::Traverse(const SingleTraversalContext& context) const
{
    context.context.localCounters.Reset();
    const uint32 count = mTraceableObjects.Size();
    if (count == 1) Traverse_Object(context, 0);
    else if (count > 1) GenericTraverse(context, 0, this);
    context.context.counters.Append(context.context.localCounters);
}
bool Scene::Traverse_Shadow(const SingleTraversalContext& context) const
{
    const uint32 count = mTraceableObjects.Size();
    if (count == 0) return false;
    return count == 1 ? Traverse_Object_Shadow(context, 0) : GenericTraverse_Shadow(context, this);
}
void Scene::Traverse(const PacketTraversalContext&) const { fprintf(stderr, "ref_render: packet traversal is not part of the glue\n"); abort(); }

// :305-365 -- object-space hit point, the shape's frame, optional normal map, Gram-Schmidt on the tangent, back to world space
void Scene::EvaluateIntersection(const Ray& ray, const HitPoint& hitPoint, const float time, IntersectionData& outData) const
{
    const ITraceableSceneObject* object = mTraceableObjects[hitPoint.objectId];
    const Matrix4 toWorld = object->GetTransform(time);
    const Matrix4 toObject = toWorld.FastInverseNoScale();
    const Vector4 worldPosition = ray.GetAtDistance(hitPoint.distance);
    outData.frame[3] = toObject.TransformPoint(worldPosition);
    object->EvaluateIntersection(hitPoint, outData);

    Vector4 tangent = outData.frame[0], normal = outData.frame[2];
    const Vector4 bitangent = Vector4::Cross3(tangent, normal);
    if (outData.material && outData.material->normalMap)
    {
        const Vector4 mapped = outData.material->GetNormalVector(outData.texCoord);
        Vector4 bent = tangent * mapped.x;
        bent = Vector4::MulAndAdd(bitangent, mapped.y, bent);
        bent = Vector4::MulAndAdd(normal, mapped.z, bent);
        normal = bent.FastNormalized3();
    }
    tangent = Vector4::Orthogonalize(tangent, normal).Normalized3();
    outData.frame[2] = toWorld.TransformVector(normal);
    outData.frame[0] = toWorld.TransformVector(tangent);
    outData.frame[1] = Vector4::Cross3(outData.frame[0], outData.frame[2]);
    outData.frame[3] = worldPosition;
}
// :367-373 (no decals in the glue)
void Scene::EvaluateShadingData(ShadingData& shadingData, RenderingContext& context) const
{
    shadingData.intersection.material->EvaluateShadingData(context.wavelength, shadingData);
}

} // namespace rt

// =====================================================================================================================================
// Scene file (written by tests/ref_render.py) -> the reference's public API
// =====================================================================================================================================
struct Reader
{
    std::vector<uint8_t> data; size_t at = 0; bool ok = true;
    template <typename T> T get() { T v{}; if (at + sizeof(T) > data.size()) { ok = false; return v; } memcpy(&v, data.data() + at, sizeof(T)); at += sizeof(T); return v; }
    template <typename T> const T* array(size_t count) { if (at + count * sizeof(T) > data.size()) { ok = false; return nullptr; } const T* p = reinterpret_cast<const T*>(data.data() + at); at += count * sizeof(T); return p; }
};

static Matrix4 readMatrix(Reader& r)
{
    const float* m = r.array<float>(16);
    Matrix4 out;
    if (m) for (int i = 0; i < 4; ++i) out[i] = Vector4(m[4 * i], m[4 * i + 1], m[4 * i + 2], m[4 * i + 3]);
    return out;
}

static uint64_t splitmix64(uint64_t& s) { uint64_t z = (s += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
// the state the repo's host mirror gives a generator for a seed (raytracer_amd/host/src/Math.cpp, Random::Reset(seed)), so that both
// sides draw the same Halton scrambles and anti-aliasing offsets
static void seedRandom(Random& rng, uint64_t seed)
{
    uint64_t s = seed;
    rng.mSeed[0] = splitmix64(s); rng.mSeed[1] = splitmix64(s) | 1ULL;
    uint64_t simd[4];
    for (int i = 0; i < 4; ++i) simd[i] = splitmix64(s) | 1ULL;
    memcpy(&rng.mSeedSimd4[0], &simd[0], 16); memcpy(&rng.mSeedSimd4[1], &simd[2], 16);
}

static const char* kBsdfNames[] = { "null", "diffuse", "roughDiffuse", "dielectric", "roughDielectric", "metal", "roughMetal", "plastic", "roughPlastic" };

// ---- adaptive-rendering known-answer file (tests/golden/adaptive_kat.bin) ---------------------------------------------------------------
// The reference's own Viewport::BuildInitialBlocksList / ComputeBlockError / UpdateBlocksList (Viewport.cpp:552-581, :618-733; Viewport.cpp is
// one of the translation units that build here) driven on synthetic sum buffers: `rounds` times the two sum bitmaps are filled with a
// seeded field whose noise shrinks from round to round and differs across the frame (so that blocks split, drop and survive), the pass
// counter is advanced by two and UpdateBlocksList runs.  Recorded per round: the buffers, the block list before, every block's error as
// ComputeBlockError returns it, the block list after.  Layout: uint32 magic 'RAK1', width, height, rounds, numInitialPasses, minBlockSize,
// maxBlockSize, 0; float subdivisionTreshold, convergenceTreshold; uint32 numInitialBlocks; blocks[numInitialBlocks][4] (minX, maxX, minY,
// maxY); then per round: uint32 passesFinished, numBefore; float sum[h][w][3], secondary[h][w][3]; float errors[numBefore];
// uint32 numAfter; blocks[numAfter][4]; float converged; uint32 activePixels.
static int adaptiveKat(const char* outPath)
{
    const uint32 width = 72, height = 52, rounds = 8;
    Viewport viewport;
    RenderingParams params;
    params.numThreads = 1;
    params.adaptiveSettings.enable = true; params.adaptiveSettings.numInitialPasses = 4; params.adaptiveSettings.minBlockSize = 5;
    params.adaptiveSettings.maxBlockSize = 32; params.adaptiveSettings.subdivisionTreshold = 0.03f; params.adaptiveSettings.convergenceTreshold = 0.0015f;
    if (!viewport.SetRenderingParams(params) || !viewport.Resize(width, height)) return 2;
    viewport.Reset();     // -> BuildInitialBlocksList
    FILE* f = fopen(outPath, "wb");
    if (!f) return 2;
    const uint32 header[8] = { 0x314B4152u, width, height, rounds, params.adaptiveSettings.numInitialPasses, params.adaptiveSettings.minBlockSize, params.adaptiveSettings.maxBlockSize, 0u };
    fwrite(header, 4, 8, f);
    fwrite(&params.adaptiveSettings.subdivisionTreshold, 4, 1, f); fwrite(&params.adaptiveSettings.convergenceTreshold, 4, 1, f);
    auto writeBlocks = [&]() {
        const uint32 n = viewport.mBlocks.Size(); fwrite(&n, 4, 1, f);
        for (uint32 i = 0; i < n; ++i) { const uint32 b[4] = { viewport.mBlocks[i].minX, viewport.mBlocks[i].maxX, viewport.mBlocks[i].minY, viewport.mBlocks[i].maxY }; fwrite(b, 4, 4, f); }
    };
    writeBlocks();
    uint64_t state = 0x9E3779B97F4A7C15ULL;
    auto uniform = [&]() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return (float)((state >> 40) & 0xFFFFFFu) * (1.0f / 16777216.0f); };
    for (uint32 round = 1; round <= rounds; ++round)
    {
        const uint32 passes = 2u * round;
        viewport.mProgress.passesFinished = passes;
        for (uint32 y = 0; y < height; ++y)
            for (uint32 x = 0; x < width; ++x)
            {
                // a smooth image; noise that is strong in the lower right corner, weak in the upper left one and fades with the rounds
                const float base[3] = { 0.2f + 0.6f * (float)x / (float)width, 0.5f, 0.9f - 0.7f * (float)y / (float)height };
                const float noisy = (float)(x * y) / (float)(width * height);
                const float amplitude = (0.004f + 2.5f * noisy * noisy) / (float)round;
                Float3& s = viewport.mSum.GetPixelRef<Float3>(x, y); Float3& t = viewport.mSecondarySum.GetPixelRef<Float3>(x, y);
                float sv[3], tv[3];
                for (int k = 0; k < 3; ++k) { sv[k] = (float)passes * base[k]; tv[k] = 0.5f * sv[k] * (1.0f + amplitude * (uniform() - 0.5f)); }
                s = Float3(sv[0], sv[1], sv[2]); t = Float3(tv[0], tv[1], tv[2]);
            }
        const uint32 numBefore = viewport.mBlocks.Size();
        fwrite(&passes, 4, 1, f); fwrite(&numBefore, 4, 1, f);
        for (int which = 0; which < 2; ++which)
        {
            const Bitmap& b = which == 0 ? viewport.mSum : viewport.mSecondarySum;
            for (uint32 y = 0; y < height; ++y) fwrite(reinterpret_cast<const uint8_t*>(b.GetData()) + (size_t)y * b.GetStride(), 4, (size_t)width * 3, f);
        }
        for (uint32 i = 0; i < numBefore; ++i) { const float e = viewport.ComputeBlockError(viewport.mBlocks[i]); fwrite(&e, 4, 1, f); }
        viewport.UpdateBlocksList();
        writeBlocks();
        fwrite(&viewport.mProgress.converged, 4, 1, f); fwrite(&viewport.mProgress.activePixels, 4, 1, f);
        fprintf(stderr, "adaptive kat round %u: %u -> %u blocks, converged %.3f\n", round, numBefore, viewport.mBlocks.Size(), viewport.mProgress.converged);
    }
    fclose(f);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc == 3 && strcmp(argv[1], "--adaptive-kat") == 0) return adaptiveKat(argv[2]);
    if (argc < 3) { fprintf(stderr, "usage: ref_render <scene.bin> <out.bin> [threads] [passes]\n"); return 2; }
    Reader r;
    {
        FILE* f = fopen(argv[1], "rb");
        if (!f) { fprintf(stderr, "ref_render: cannot open %s\n", argv[1]); return 2; }
        fseek(f, 0, SEEK_END); const long size = ftell(f); fseek(f, 0, SEEK_SET);
        r.data.resize((size_t)size);
        if (fread(r.data.data(), 1, (size_t)size, f) != (size_t)size) { fclose(f); return 2; }
        fclose(f);
    }
    const uint64_t magic = r.get<uint64_t>();
    if (magic != 0x3230304645525452ULL) { fprintf(stderr, "ref_render: bad magic\n"); return 2; }   // "RTREF002"
    const uint32 width = r.get<uint32>(), height = r.get<uint32>(); uint32 passes = r.get<uint32>(), threads = r.get<uint32>();
    const uint32 maxRayDepth = r.get<uint32>(), minRouletteDepth = r.get<uint32>(), dimensions = r.get<uint32>(), blueNoise = r.get<uint32>(), samplingAll = r.get<uint32>();
    const uint32 numMaterials = r.get<uint32>(), numMeshes = r.get<uint32>(), numObjects = r.get<uint32>(), numLights = r.get<uint32>(), dumpImage = r.get<uint32>();
    const float aaSpread = r.get<float>(); (void)r.get<float>();
    const uint64_t seed = r.get<uint64_t>();
    const uint32 numTextures = r.get<uint32>(); (void)r.get<uint32>();
    if (argc > 3) threads = (uint32)atoi(argv[3]);
    if (argc > 4) passes = (uint32)atoi(argv[4]);

    // the reference opens "../Data/BlueNoise128_RGBA16.dat" relative to the working directory (Core/Sampling/GenericSampler.cpp:13):
    // RT_REF_CWD names a directory one level below a Data/ directory holding that table (the repo ships it as raytracer_amd/data)
    if (const char* cwd = getenv("RT_REF_CWD")) { if (chdir(cwd) != 0) { fprintf(stderr, "ref_render: cannot enter %s\n", cwd); return 2; } }

    // Demo/Main.cpp and Tests/Main.cpp flush denormals (Core/Math/Math.cpp:27-34); RT_REF_KEEP_DENORMALS=1 measures what that changes
    SetFlushDenormalsToZero(getenv("RT_REF_KEEP_DENORMALS") == nullptr);

    Camera camera;
    {
        const float* t = r.array<float>(3); const float* e = r.array<float>(3);
        const float fov = r.get<float>(), aspect = r.get<float>();
        const uint32 dof = r.get<uint32>(); const float focal = r.get<float>(), aperture = r.get<float>(); const uint32 bokeh = r.get<uint32>();
        if (!r.ok) return 2;
        Vector4 orientation(e[0], e[1], e[2], 0.0f);
        orientation *= (RT_PI / 180.0f);
        camera.SetTransform(Transform(Vector4(t[0], t[1], t[2], 0.0f), Quaternion::FromEulerAngles(orientation.ToFloat3())));
        camera.SetPerspective(aspect, fov);
        camera.mDOF.enable = dof != 0; camera.mDOF.focalPlaneDistance = focal; camera.mDOF.aperture = aperture;
        if (bokeh > 2u) return 2;   // circle, hexagon, square (Camera.h:21-28); NGon is commented out in the reference, Texture needs a bitmap
        camera.mDOF.bokehShape = (BokehShape)bokeh;
    }

    // bitmap textures: Bitmap::Init copies the texels; BitmapTexture(bitmap) is the constructor Demo/MeshLoader.cpp uses (default filter)
    std::vector<TexturePtr> textures;
    for (uint32 i = 0; i < numTextures; ++i)
    {
        const uint32 tw = r.get<uint32>(), th = r.get<uint32>(), format = r.get<uint32>(), linear = r.get<uint32>(), stride = r.get<uint32>(), bytes = r.get<uint32>();
        (void)r.get<uint32>(); (void)r.get<uint32>();
        const uint8_t* texels = r.array<uint8_t>(((size_t)bytes + 3u) & ~(size_t)3u);
        if (!r.ok || format == 0u || format > (uint32)Bitmap::Format::BC5) return 2;
        Bitmap::InitData init;
        init.width = tw; init.height = th; init.format = (Bitmap::Format)format; init.data = texels; init.stride = stride; init.linearSpace = linear != 0;
        BitmapPtr bitmap = std::make_shared<Bitmap>("fixture");
        if (!bitmap->Init(init)) { fprintf(stderr, "ref_render: Bitmap::Init failed\n"); return 2; }
        textures.push_back(std::make_shared<BitmapTexture>(bitmap));
    }

    std::vector<MaterialPtr> materials;
    for (uint32 i = 0; i < numMaterials; ++i)
    {
        const uint32 bsdf = r.get<uint32>(); const float* c = r.array<float>(10); const int32* t = r.array<int32>(5); const float normalMapStrength = r.get<float>();
        if (!r.ok || bsdf > 8) return 2;
        MaterialPtr m = Material::Create();
        m->SetBsdf(kBsdfNames[bsdf]);
        m->baseColor.baseValue = Vector4(c[0], c[1], c[2], 0.0f); m->emission.baseValue = Vector4(c[3], c[4], c[5], 0.0f);
        m->roughness.baseValue = c[6]; m->metalness.baseValue = c[7]; m->IoR = c[8]; m->K = c[9];
        auto textureOf = [&](int32 index) -> TexturePtr { return index >= 0 && (uint32)index < numTextures ? textures[(uint32)index] : TexturePtr(); };
        m->baseColor.texture = textureOf(t[0]); m->emission.texture = textureOf(t[1]); m->roughness.texture = textureOf(t[2]); m->metalness.texture = textureOf(t[3]);
        m->normalMap = textureOf(t[4]); m->normalMapStrength = normalMapStrength;
        m->Compile();
        materials.push_back(m);
    }

    std::vector<ShapePtr> meshes;
    std::vector<std::vector<MaterialPtr>> meshMaterials(numMeshes);
    for (uint32 i = 0; i < numMeshes; ++i)
    {
        const uint32 nv = r.get<uint32>(), nt = r.get<uint32>(), nm = r.get<uint32>(); (void)r.get<uint32>();
        MeshDesc desc;
        desc.vertexBufferDesc.numVertices = nv; desc.vertexBufferDesc.numTriangles = nt; desc.vertexBufferDesc.numMaterials = nm;
        desc.vertexBufferDesc.positions = r.array<Float3>(nv); desc.vertexBufferDesc.normals = r.array<Float3>(nv);
        desc.vertexBufferDesc.tangents = r.array<Float3>(nv); desc.vertexBufferDesc.texCoords = r.array<Float2>(nv);
        desc.vertexBufferDesc.vertexIndexBuffer = r.array<uint32>((size_t)nt * 3); desc.vertexBufferDesc.materialIndexBuffer = r.array<uint32>(nt);
        const uint32* table = r.array<uint32>(nm);
        if (!r.ok) return 2;
        for (uint32 k = 0; k < nm; ++k) { if (table[k] >= numMaterials) return 2; meshMaterials[i].push_back(materials[table[k]]); }
        desc.vertexBufferDesc.materials = meshMaterials[i].data();
        auto mesh = std::make_shared<MeshShape>();
        if (!mesh->Initialize(desc)) { fprintf(stderr, "ref_render: MeshShape::Initialize failed\n"); return 2; }
        meshes.push_back(mesh);
    }

    Scene scene;
    auto makeShape = [&](uint32 kind, const float* p) -> ShapePtr
    {
        if (kind == 0) return std::make_shared<SphereShape>(p[0]);
        if (kind == 1) return std::make_shared<BoxShape>(Vector4(p[0], p[1], p[2], 0.0f));
        if (kind == 2) return std::make_shared<RectShape>(Float2(p[0], p[1]), Float2(p[2], p[3]));
        return ShapePtr();
    };
    for (uint32 i = 0; i < numObjects; ++i)
    {
        const uint32 kind = r.get<uint32>(); const int32 material = r.get<int32>(); const uint32 meshIndex = r.get<uint32>(); (void)r.get<uint32>();
        const float* p = r.array<float>(4); const Matrix4 transform = readMatrix(r);
        if (!r.ok) return 2;
        ShapePtr shape = kind == 3 ? (meshIndex < meshes.size() ? meshes[meshIndex] : ShapePtr()) : makeShape(kind, p);
        if (!shape) return 2;
        auto object = std::make_unique<ShapeSceneObject>(shape);
        if (material >= 0 && (uint32)material < numMaterials) object->SetDefaultMaterial(materials[(uint32)material]);
        object->SetTransform(transform);
        scene.AddObject(std::move(object));
    }
    for (uint32 i = 0; i < numLights; ++i)
    {
        const uint32 kind = r.get<uint32>(), shapeKind = r.get<uint32>(), texturePlusOne = r.get<uint32>(); (void)r.get<uint32>();
        const float* c = r.array<float>(4); const float* p = r.array<float>(4); const Matrix4 transform = readMatrix(r);
        if (!r.ok) return 2;
        const Vector4 color(c[0], c[1], c[2], 0.0f);
        LightPtr light;
        if (kind == 0) { ShapePtr shape = makeShape(shapeKind, p); if (!shape) return 2; light = std::make_unique<AreaLight>(shape, color); }
        else if (kind == 1) light = std::make_unique<PointLight>(color);
        else if (kind == 2) light = std::make_unique<SpotLight>(color, p[0]);
        else if (kind == 3) light = std::make_unique<DirectionalLight>(color, p[0]);
        else if (kind == 4)
        {
            auto background = std::make_unique<BackgroundLight>(color);
            if (texturePlusOne != 0u) { if (texturePlusOne > numTextures) return 2; background->mTexture = textures[texturePlusOne - 1u]; }   // environment map (BackgroundLight.h:16)
            light = std::move(background);
        }
        else return 2;
        auto object = std::make_unique<LightSceneObject>(std::move(light));
        object->SetTransform(transform);
        scene.AddObject(std::move(object));
    }
    if (!scene.BuildBVH()) { fprintf(stderr, "ref_render: Scene::BuildBVH failed\n"); return 2; }

    // Viewport(), SetRenderingParams, Resize, SetRenderer, Reset: the call sequence of the repo's mirror (raytracer_amd.Viewport)
    Viewport viewport;
    RenderingParams params;
    params.numThreads = threads;
    params.samplingParams.dimensions = dimensions; params.samplingParams.useBlueNoiseDithering = blueNoise != 0;
    params.antiAliasingSpread = aaSpread;
    params.maxRayDepth = maxRayDepth; params.minRussianRouletteDepth = minRouletteDepth;
    params.traversalMode = TraversalMode::Single;
    params.lightSamplingStrategy = samplingAll ? LightSamplingStrategy::All : LightSamplingStrategy::Single;
    if (!viewport.SetRenderingParams(params)) return 2;
    seedRandom(viewport.mRandomGenerator, seed);
    seedRandom(viewport.mHaltonSequence.mRandom, seed ^ 0xA5A5A5A55A5A5A5AULL);
    for (uint32 i = 0; i < viewport.mThreadData.Size(); ++i) seedRandom(viewport.mThreadData[i].randomGenerator, seed + 0x1000u + i);
    if (!viewport.Resize(width, height)) return 2;
    RendererPtr renderer = std::make_shared<PathTracerMIS>(scene);
    if (!viewport.SetRenderer(renderer)) return 2;
    viewport.Reset();

#ifdef RT_REF_PATH_DEBUG
    // ref_paths: PathTracerMIS.cpp built WITHOUT RT_CONFIGURATION_FINAL records every vertex of every path (PathDebugData, the hook the
    // reference's Demo uses for its path inspector, PathTracerMIS.cpp:377-410 / Demo.cpp:288-291).  One thread, so one context sees all.
    PathDebugData pathDebug;
    if (threads != 1) { fprintf(stderr, "ref_paths needs one thread\n"); return 2; }
    viewport.mThreadData[0].pathDebugData = &pathDebug;
#endif
    RayTracingCounters total; total.Reset();
    std::vector<uint32> firstSeeds;
    float firstOffset[2] = { 0.0f, 0.0f };
    {
        // the anti-aliasing offset the first pass will draw (Viewport.cpp:235-242), from a copy of the generator
        Random copy = viewport.mRandomGenerator;
        const Vector4 u = SamplingHelpers::GetFloatNormal2(copy.GetFloat2()) * aaSpread;
        firstOffset[0] = u.x; firstOffset[1] = u.y;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32 p = 0; p < passes; ++p)
    {
        if (!viewport.Render(camera)) { fprintf(stderr, "ref_render: Viewport::Render failed\n"); return 2; }
        total.Append(viewport.GetCounters());
        if (p == 0)
        {
            // the per-pass constants the first pass ran with (what the repo's NextPassParams must reproduce)
            const GenericSampler& sampler = viewport.mThreadData[0].sampler;
            for (uint32 d = 0; d < dimensions && d < sampler.mCurrentSample.Size(); ++d) firstSeeds.push_back(sampler.mCurrentSample[d]);
        }
    }
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    const Bitmap& sum = viewport.GetSumBuffer();
    double mean[3] = { 0, 0, 0 }; uint64_t hash = 1469598103934665603ULL;
    for (uint32 y = 0; y < height; ++y)
    {
        const float* row = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(sum.GetData()) + (size_t)y * sum.GetStride());
        for (uint32 x = 0; x < width; ++x) for (int k = 0; k < 3; ++k) { mean[k] += row[3 * x + k]; uint32_t b; memcpy(&b, &row[3 * x + k], 4); hash = (hash ^ b) * 1099511628211ULL; }
    }
    for (int k = 0; k < 3; ++k) mean[k] /= (double)width * height * (passes ? passes : 1);
    printf("{\"seconds\": %.6f, \"passes\": %u, \"threads\": %u, \"hardware_threads\": %u, \"width\": %u, \"height\": %u, \"numRays\": %llu, \"numPrimaryRays\": %llu, "
           "\"numShadowRays\": %llu, \"numShadowRaysHit\": %llu, \"msamples_per_s\": %.6f, \"mean\": [%.9g, %.9g, %.9g], \"fnv1a\": \"%016llx\"}\n",
           seconds, passes, viewport.mThreadData.Size(), std::thread::hardware_concurrency(), width, height, (unsigned long long)total.numRays,
           (unsigned long long)total.numPrimaryRays, (unsigned long long)total.numShadowRays, (unsigned long long)total.numShadowRaysHit,
           (double)total.numRays / seconds / 1e6, mean[0], mean[1], mean[2], (unsigned long long)hash);

    if (FILE* f = fopen(argv[2], "wb"))
    {
        const uint32 header[8] = { 0x54554F52u, width, height, passes, (uint32)firstSeeds.size(), dumpImage, 0u, 0u };   // "ROUT"
        fwrite(header, 4, 8, f);
        const uint64_t counters[4] = { total.numRays, total.numPrimaryRays, total.numShadowRays, total.numShadowRaysHit };
        fwrite(counters, 8, 4, f);
        fwrite(&seconds, 8, 1, f);
        fwrite(firstOffset, 4, 2, f);
        if (!firstSeeds.empty()) fwrite(firstSeeds.data(), 4, firstSeeds.size(), f);
        if (dumpImage) for (uint32 y = 0; y < height; ++y) fwrite(reinterpret_cast<const uint8_t*>(sum.GetData()) + (size_t)y * sum.GetStride(), 4, (size_t)width * 3, f);
#ifdef RT_REF_PATH_DEBUG
        // every recorded vertex in the order the pixels were rendered: 28 floats each -- ray origin xyz, ray direction xyz, hit objectId,
        // subObjectId (bit-cast), distance, u, v, frame position xyz, normal xyz, tangent xyz, texCoord xy, throughput xyzw, bsdfEvent
        const uint32 numVertices = pathDebug.data.Size();
        fwrite(&numVertices, 4, 1, f);
        for (uint32 i = 0; i < numVertices; ++i)
        {
            const PathDebugData::HitPointData& d = pathDebug.data[i];
            float rec[28]; uint32 bits;
            rec[0] = d.rayOrigin.x; rec[1] = d.rayOrigin.y; rec[2] = d.rayOrigin.z; rec[3] = d.rayDir.x; rec[4] = d.rayDir.y; rec[5] = d.rayDir.z;
            bits = d.hitPoint.objectId; memcpy(&rec[6], &bits, 4); bits = d.hitPoint.subObjectId; memcpy(&rec[7], &bits, 4);
            rec[8] = d.hitPoint.distance; rec[9] = d.hitPoint.u; rec[10] = d.hitPoint.v;
            const Matrix4& fr = d.shadingData.intersection.frame;
            rec[11] = fr[3].x; rec[12] = fr[3].y; rec[13] = fr[3].z; rec[14] = fr[2].x; rec[15] = fr[2].y; rec[16] = fr[2].z; rec[17] = fr[0].x; rec[18] = fr[0].y; rec[19] = fr[0].z;
            rec[20] = d.shadingData.intersection.texCoord.x; rec[21] = d.shadingData.intersection.texCoord.y;
            rec[22] = d.throughput.value.x; rec[23] = d.throughput.value.y; rec[24] = d.throughput.value.z; rec[25] = d.throughput.value.w;
            bits = (uint32)d.bsdfEvent; memcpy(&rec[26], &bits, 4); rec[27] = 0.0f;
            fwrite(rec, 4, 28, f);
        }
#endif
        fclose(f);
    }
    return 0;
}
