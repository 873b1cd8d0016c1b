// rth_* : a flat C facade over the C++ host mirror (rt::Scene / rt::Viewport / ...), used by the Python
// test and benchmark drivers through ctypes.  Plumbing only: everything here forwards to the classes in
// Core/ which in turn call the device C-ABI of include/rtgpu.h.
#include "../Core/Rendering/Viewport.h"
#include "../Core/Rendering/PathTracerMIS.h"
#include "../Core/Rendering/VertexConnectionAndMerging.h"
#include "../Core/BVH/BVHBuilder.h"
#include "../Core/Textures/BitmapTexture.h"
#include "../Core/Textures/CheckerboardTexture.h"
#include "../Core/Textures/ConstTexture.h"
#include "../Core/Textures/NoiseTexture.h"
#include "../Core/Textures/MixTexture.h"
#include "../Demo/Demo.h"
#include "../Demo/MeshLoader.h"
#include "../Demo/ObjReader.h"
#include "../Demo/SceneLoader.h"

#include <stdio.h>

using namespace rt;
using namespace rt::math;

namespace {

struct SceneHandle
{
    Scene scene;
    std::vector<MaterialPtr> materials;
    std::vector<TexturePtr> textures;
};

struct ViewportHandle
{
    Viewport viewport;
    RendererPtr renderer;
};

Matrix4 LoadMatrix(const float* m)
{
    if (!m) return Matrix4::Identity();
    Matrix4 r;
    memcpy(r.rows, m, 64);
    return r;
}

Vector4 LoadColor(const float* c) { return Vector4(c[0], c[1], c[2], c[3]); }

MaterialPtr GetMaterial(SceneHandle* s, int id)
{
    if (id < 0 || id >= (int)s->materials.size()) return nullptr;
    return s->materials[(size_t)id];
}

} // namespace

extern "C" {

#define RTH_API __attribute__((visibility("default")))

// ---- transforms ------------------------------------------------------------------------------------
// translation + Euler orientation in DEGREES (pitch, yaw, roll), like the reference's JSON loader
// (Demo/SceneLoader.cpp:189-216)
RTH_API void rth_transform_from_euler(const float translation[3], const float orientationDeg[3], float out[16])
{
    Vector4 orientation(orientationDeg[0], orientationDeg[1], orientationDeg[2], 0.0f);
    orientation *= (RT_PI / 180.0f);
    const Transform t(Vector4(translation[0], translation[1], translation[2], 0.0f), Quaternion::FromEulerAngles(orientation.ToFloat3()));
    t.ToMatrix4().Store(out);
}

RTH_API void rth_matrix_inverse(const float in[16], float out[16]) { LoadMatrix(in).Inverse().Store(out); }

// ---- scene -----------------------------------------------------------------------------------------
RTH_API void* rth_scene_create() { return new SceneHandle(); }
RTH_API void rth_scene_destroy(void* s) { delete static_cast<SceneHandle*>(s); }

RTH_API int rth_material_create(void* sh, const char* bsdf, const float baseColor[4], const float emission[4],
                                float roughness, float metalness, float IoR, float K)
{
    SceneHandle* s = static_cast<SceneHandle*>(sh);
    MaterialPtr m = std::make_shared<Material>();
    m->SetBsdf(bsdf);
    if (m->GetBsdfKind() < 0) return -1;
    m->baseColor = LoadColor(baseColor);
    m->emission = LoadColor(emission);
    m->roughness = roughness;
    m->metalness = metalness;
    m->IoR = IoR;
    m->K = K;
    m->Compile();
    s->materials.push_back(m);
    return (int)s->materials.size() - 1;
}

// ---- textures ---------------------------------------------------------------------------------------
// format = rt::Bitmap::Format value; data: height rows of `stride` bytes (0 = tight); returns a texture id
RTH_API int rth_texture_bitmap(void* sh, uint32_t width, uint32_t height, uint32_t format, const void* data, uint32_t stride, int linearSpace, int filter)
{
    SceneHandle* s = static_cast<SceneHandle*>(sh);
    Bitmap::InitData init;
    init.width = width; init.height = height; init.format = (Bitmap::Format)format; init.data = data; init.stride = stride;
    if (init.format == Bitmap::Format::B8G8R8A8_UNorm_Palette) init.paletteSize = 256;
    init.linearSpace = linearSpace != 0;
    BitmapPtr bitmap = std::make_shared<Bitmap>("texture");
    if (!bitmap->Init(init)) return -1;
    auto tex = std::make_shared<BitmapTexture>(bitmap);
    tex->SetFilter((BitmapTextureFilter)filter);
    s->textures.push_back(tex);
    return (int)s->textures.size() - 1;
}
RTH_API int rth_texture_checkerboard(void* sh, const float colorA[4], const float colorB[4])
{
    SceneHandle* s = static_cast<SceneHandle*>(sh);
    s->textures.push_back(std::make_shared<CheckerboardTexture>(LoadColor(colorA), LoadColor(colorB)));
    return (int)s->textures.size() - 1;
}
RTH_API int rth_texture_const(void* sh, const float color[4])
{
    SceneHandle* s = static_cast<SceneHandle*>(sh);
    s->textures.push_back(std::make_shared<ConstTexture>(LoadColor(color)));
    return (int)s->textures.size() - 1;
}
RTH_API int rth_texture_noise(void* sh, const float colorA[4], const float colorB[4], uint32_t octaves)
{
    SceneHandle* s = static_cast<SceneHandle*>(sh);
    s->textures.push_back(std::make_shared<NoiseTexture>(LoadColor(colorA), LoadColor(colorB), octaves));
    return (int)s->textures.size() - 1;
}
RTH_API int rth_texture_mix(void* sh, int a, int b, int weight)
{
    SceneHandle* s = static_cast<SceneHandle*>(sh);
    const int n = (int)s->textures.size();
    if (a < 0 || b < 0 || weight < 0 || a >= n || b >= n || weight >= n) return -1;
    s->textures.push_back(std::make_shared<MixTexture>(s->textures[(size_t)a], s->textures[(size_t)b], s->textures[(size_t)weight]));
    return (int)s->textures.size() - 1;
}
// palette of a B8G8R8A8_UNorm_Palette bitmap texture: numEntries * 4 bytes (B, G, R, A)
RTH_API int rth_texture_set_palette(void* sh, int texture, const uint8_t* entries, uint32_t numEntries)
{
    SceneHandle* s = static_cast<SceneHandle*>(sh);
    if (texture < 0 || texture >= (int)s->textures.size()) return -1;
    BitmapTexture* bt = dynamic_cast<BitmapTexture*>(s->textures[(size_t)texture].get());
    if (!bt || !bt->GetBitmap() || bt->GetBitmap()->GetPaletteSize() < numEntries) return -1;
    memcpy(bt->GetBitmap()->GetPalette(), entries, (size_t)numEntries * 4u);
    return 0;
}
// slot: 0 baseColor, 1 emission, 2 roughness, 3 metalness, 4 normal map (strength = normalMapStrength)
RTH_API int rth_material_set_texture(void* sh, int material, int slot, int texture, float strength)
{
    SceneHandle* s = static_cast<SceneHandle*>(sh);
    MaterialPtr m = GetMaterial(s, material);
    if (!m || texture < 0 || texture >= (int)s->textures.size()) return -1;
    const TexturePtr& t = s->textures[(size_t)texture];
    switch (slot)
    {
    case 0: m->baseColor.texture = t; break;
    case 1: m->emission.texture = t; break;
    case 2: m->roughness.texture = t; break;
    case 3: m->metalness.texture = t; break;
    case 4: m->normalMap = t; m->normalMapStrength = strength; break;
    default: return -1;
    }
    return 0;
}

static int AddShape(SceneHandle* s, const ShapePtr& shape, const float* transform, int material)
{
    ShapeSceneObjectPtr obj = std::make_unique<ShapeSceneObject>(shape);
    obj->SetDefaultMaterial(GetMaterial(s, material));
    obj->SetTransform(LoadMatrix(transform));
    s->scene.AddObject(std::move(obj));
    return 0;
}

RTH_API int rth_add_sphere(void* sh, float radius, const float transform[16], int material)
{
    return AddShape(static_cast<SceneHandle*>(sh), std::make_shared<SphereShape>(radius), transform, material);
}

RTH_API int rth_add_box(void* sh, const float size[3], const float transform[16], int material)
{
    return AddShape(static_cast<SceneHandle*>(sh), std::make_shared<BoxShape>(Vector4(size[0], size[1], size[2], 0.0f)), transform, material);
}

RTH_API int rth_add_rect(void* sh, const float size[2], const float texScale[2], const float transform[16], int material)
{
    return AddShape(static_cast<SceneHandle*>(sh), std::make_shared<RectShape>(Float2(size[0], size[1]), Float2(texScale[0], texScale[1])), transform, material);
}

// positions/normals/tangents: numVertices*3 floats, texCoords: numVertices*2 (normals/tangents/texCoords may be NULL),
// indices: numTriangles*3, materialIndices: numTriangles (may be NULL => default material), materialIds: scene material ids
RTH_API int rth_add_mesh(void* sh, uint32_t numVertices, uint32_t numTriangles, const float* positions, const float* normals,
                         const float* tangents, const float* texCoords, const uint32_t* indices, const uint32_t* materialIndices,
                         uint32_t numMaterials, const int* materialIds, const float transform[16], int defaultMaterial)
{
    SceneHandle* s = static_cast<SceneHandle*>(sh);
    std::vector<MaterialPtr> mats;
    for (uint32_t i = 0; i < numMaterials; ++i)
    {
        MaterialPtr m = GetMaterial(s, materialIds[i]);
        if (!m) return -1;
        mats.push_back(m);
    }
    std::vector<uint32_t> noMaterial;
    if (!materialIndices) { noMaterial.assign(numTriangles, UINT32_MAX); materialIndices = noMaterial.data(); }
    MeshDesc desc;
    desc.vertexBufferDesc.numVertices = numVertices;
    desc.vertexBufferDesc.numTriangles = numTriangles;
    desc.vertexBufferDesc.numMaterials = numMaterials;
    desc.vertexBufferDesc.vertexIndexBuffer = indices;
    desc.vertexBufferDesc.positions = reinterpret_cast<const Float3*>(positions);
    desc.vertexBufferDesc.normals = reinterpret_cast<const Float3*>(normals);
    desc.vertexBufferDesc.tangents = reinterpret_cast<const Float3*>(tangents);
    desc.vertexBufferDesc.texCoords = reinterpret_cast<const Float2*>(texCoords);
    desc.vertexBufferDesc.materialIndexBuffer = materialIndices;
    desc.vertexBufferDesc.materials = mats.data();
    std::shared_ptr<MeshShape> mesh = std::make_shared<MeshShape>();
    if (!mesh->Initialize(desc)) return -2;
    return AddShape(s, mesh, transform, defaultMaterial);
}

static int AddLight(SceneHandle* s, LightPtr light, const float* transform)
{
    LightSceneObjectPtr obj = std::make_unique<LightSceneObject>(std::move(light));
    obj->SetTransform(LoadMatrix(transform));
    s->scene.AddObject(std::move(obj));
    return 0;
}

// shapeKind: 0 sphere (p[0] = radius), 1 box (p[0..2] = half extents), 2 rect (p[0..1] = half size)
RTH_API int rth_add_light_area(void* sh, int shapeKind, const float p[4], const float color[4], const float transform[16])
{
    ShapePtr shape;
    if (shapeKind == 0) shape = std::make_shared<SphereShape>(p[0]);
    else if (shapeKind == 1) shape = std::make_shared<BoxShape>(Vector4(p[0], p[1], p[2], 0.0f));
    else if (shapeKind == 2) shape = std::make_shared<RectShape>(Float2(p[0], p[1]));
    else return -1;
    return AddLight(static_cast<SceneHandle*>(sh), std::make_unique<AreaLight>(shape, LoadColor(color)), transform);
}
RTH_API int rth_add_light_background(void* sh, const float color[4])
{
    return AddLight(static_cast<SceneHandle*>(sh), std::make_unique<BackgroundLight>(LoadColor(color)), nullptr);
}
RTH_API int rth_add_light_background_textured(void* sh, const float color[4], int texture)
{
    SceneHandle* s = static_cast<SceneHandle*>(sh);
    if (texture < 0 || texture >= (int)s->textures.size()) return -1;
    auto light = std::make_unique<BackgroundLight>(LoadColor(color));
    light->mTexture = s->textures[(size_t)texture];
    return AddLight(s, std::move(light), nullptr);
}
RTH_API int rth_add_light_directional(void* sh, const float color[4], float angleRad, const float transform[16])
{
    return AddLight(static_cast<SceneHandle*>(sh), std::make_unique<DirectionalLight>(LoadColor(color), angleRad), transform);
}
RTH_API int rth_add_light_point(void* sh, const float color[4], const float transform[16])
{
    return AddLight(static_cast<SceneHandle*>(sh), std::make_unique<PointLight>(LoadColor(color)), transform);
}
RTH_API int rth_add_light_spot(void* sh, const float color[4], float angleRad, const float transform[16])
{
    return AddLight(static_cast<SceneHandle*>(sh), std::make_unique<SpotLight>(LoadColor(color), angleRad), transform);
}

// ---- ingestion (Demo/SceneLoader.cpp, Demo/MeshLoader.cpp) ---------------------------------------------------------
// helpers::LoadScene(path) into this scene; camera may be NULL.  dataPath = Options::dataPath (prefix of mesh / texture paths)
RTH_API int rth_load_scene(void* sh, void* camera, const char* path, const char* dataPath)
{
    SceneHandle* s = static_cast<SceneHandle*>(sh);
    gOptions.dataPath = dataPath ? dataPath : "";
    Camera scratch;
    return helpers::LoadScene(path, s->scene, camera ? *static_cast<Camera*>(camera) : scratch) ? 0 : -1;
}
// helpers::LoadMesh's vertex streams for the parity test: two calls, the first with NULL outputs returns the sizes
RTH_API int rth_load_mesh_streams(const char* path, float scale, uint32_t* numVertices, uint32_t* numTriangles, uint32_t* numMaterials,
                                  float* positions, float* normals, float* tangents, float* texCoords, uint32_t* indices, uint32_t* materialIndices)
{
    helpers::MaterialsMap materials;
    helpers::MeshStreams m;
    if (!helpers::LoadMeshStreams(path, materials, scale, m)) return -1;
    *numVertices = (uint32_t)m.positions.size(); *numTriangles = (uint32_t)(m.vertexIndices.size() / 3); *numMaterials = (uint32_t)m.materials.size();
    if (positions) memcpy(positions, m.positions.data(), m.positions.size() * 12);
    if (normals) memcpy(normals, m.normals.data(), m.normals.size() * 12);
    if (tangents) memcpy(tangents, m.tangents.data(), m.tangents.size() * 12);
    if (texCoords) memcpy(texCoords, m.texCoords.data(), m.texCoords.size() * 8);
    if (indices) memcpy(indices, m.vertexIndices.data(), m.vertexIndices.size() * 4);
    if (materialIndices) memcpy(materialIndices, m.materialIndices.data(), m.materialIndices.size() * 4);
    return 0;
}
RTH_API void rth_kat_world_to_screen(const float localToWorld[16], float aspectRatio, float tanHalfFoV, float out[16])
{
    Matrix4 l; memcpy(&l, localToWorld, 64);
    Camera::ComputeWorldToScreen(l, aspectRatio, tanHalfFoV).Store(out);
}
// Bitmap::Load on a file: out = { format, linearSpace, width, height, dataBytes, byteSum (sum * 31 + byte) }; -1 if it does not load
RTH_API int rth_kat_load_bitmap(const char* path, uint32_t out[6])
{
    Bitmap bitmap;
    if (!bitmap.Load(path)) return -1;
    const size_t bytes = (size_t)bitmap.GetHeight() * bitmap.GetStride();
    uint32_t sum = 0; const uint8_t* data = reinterpret_cast<const uint8_t*>(bitmap.GetData());
    for (size_t i = 0; i < bytes; ++i) sum = sum * 31u + data[i];
    out[0] = (uint32_t)bitmap.GetFormat(); out[1] = bitmap.IsLinearSpace() ? 1u : 0u; out[2] = bitmap.GetWidth(); out[3] = bitmap.GetHeight(); out[4] = (uint32_t)bytes; out[5] = sum;
    return 0;
}
RTH_API int rth_kat_load_bitmap_palette(const char* path, uint32_t out[2])   // { palette entries, checksum of the palette bytes }
{
    Bitmap bitmap;
    if (!bitmap.Load(path)) return -1;
    uint32_t sum = 0;
    for (size_t i = 0; i < (size_t)bitmap.GetPaletteSize() * 4u; ++i) sum = sum * 31u + bitmap.GetPalette()[i];
    out[0] = bitmap.GetPaletteSize(); out[1] = sum;
    return 0;
}
RTH_API int rth_kat_parse_double(const char* text, double* out) { return helpers::obj::TryParseDouble(text, text + strlen(text), out) ? 0 : -1; }

RTH_API int rth_scene_build(void* sh) { return static_cast<SceneHandle*>(sh)->scene.BuildBVH() ? 0 : -1; }
RTH_API const RtSceneDesc* rth_scene_desc(void* sh) { return &static_cast<SceneHandle*>(sh)->scene.GetDesc(); }

// ---- camera -----------------------------------------------------------------------------------------
RTH_API void* rth_camera_create() { return new Camera(); }
RTH_API void rth_camera_destroy(void* c) { delete static_cast<Camera*>(c); }
RTH_API void rth_camera_set_transform(void* c, const float translation[3], const float orientationDeg[3])
{
    Vector4 orientation(orientationDeg[0], orientationDeg[1], orientationDeg[2], 0.0f);
    orientation *= (RT_PI / 180.0f);
    static_cast<Camera*>(c)->SetTransform(Transform(Vector4(translation[0], translation[1], translation[2], 0.0f), Quaternion::FromEulerAngles(orientation.ToFloat3())));
}
RTH_API void rth_camera_set_perspective(void* c, float aspect, float fovRad) { static_cast<Camera*>(c)->SetPerspective(aspect, fovRad); }
RTH_API void rth_camera_set_dof(void* c, int enable, float focalPlaneDistance, float aperture)
{
    Camera* cam = static_cast<Camera*>(c);
    cam->mDOF.enable = enable != 0; cam->mDOF.focalPlaneDistance = focalPlaneDistance; cam->mDOF.aperture = aperture;
}
RTH_API void rth_camera_set_lens(void* c, uint32_t bokehShape, float barrelDistortionConstFactor, float barrelDistortionVariableFactor)
{
    Camera* cam = static_cast<Camera*>(c);
    cam->mDOF.bokehShape = (BokehShape)bokehShape;
    cam->barrelDistortionConstFactor = barrelDistortionConstFactor; cam->barrelDistortionVariableFactor = barrelDistortionVariableFactor;
}
RTH_API int rth_camera_desc(void* c, RtCamera* out) { return static_cast<Camera*>(c)->GetDesc(*out) ? 0 : -1; }

// ---- viewport ---------------------------------------------------------------------------------------
RTH_API void* rth_viewport_create() { return new ViewportHandle(); }
RTH_API void rth_viewport_destroy(void* v) { delete static_cast<ViewportHandle*>(v); }
RTH_API int rth_viewport_resize(void* v, uint32_t w, uint32_t h) { return static_cast<ViewportHandle*>(v)->viewport.Resize(w, h) ? 0 : -1; }
RTH_API int rth_viewport_set_params(void* v, uint32_t dimensions, int useBlueNoise, float antiAliasingSpread,
                                    uint32_t maxRayDepth, uint32_t minRussianRouletteDepth, int lightSamplingAll)
{
    RenderingParams p;
    p.samplingParams.dimensions = dimensions;
    p.samplingParams.useBlueNoiseDithering = useBlueNoise != 0;
    p.antiAliasingSpread = antiAliasingSpread;
    p.maxRayDepth = maxRayDepth;
    p.minRussianRouletteDepth = minRussianRouletteDepth;
    p.lightSamplingStrategy = lightSamplingAll ? LightSamplingStrategy::All : LightSamplingStrategy::Single;
    return static_cast<ViewportHandle*>(v)->viewport.SetRenderingParams(p) ? 0 : -1;
}
RTH_API void rth_viewport_set_seed(void* v, uint64_t seed) { static_cast<ViewportHandle*>(v)->viewport.SetSeed(seed); }
// device < 0: default (LOCAL_RANK or 0).  Returns -2 when no renderer could be created (no GPU / unknown name).
RTH_API int rth_viewport_set_renderer(void* v, void* sh, const char* name, int device)
{
    ViewportHandle* vh = static_cast<ViewportHandle*>(v);
    // the device applies to THIS renderer only: the process-wide default is put back afterwards
    const int previous = GetRendererDevice();
    if (device >= 0) SetRendererDevice(device);
    vh->renderer = CreateRenderer(name, static_cast<SceneHandle*>(sh)->scene);
    if (device >= 0) SetRendererDevice(previous);
    if (!vh->renderer) return -2;
    return vh->viewport.SetRenderer(vh->renderer) ? 0 : -1;
}
// the devices of renderers created afterwards (SetRendererDevices); n == 0: back to one device
RTH_API void rth_set_renderer_devices(const int* devices, uint32_t n) { SetRendererDevices(std::vector<int>(devices, devices + n)); }
// the public knobs of the "VCM" renderer (no-op with -3 when the viewport's renderer is not VCM); weights = 5 scalars:
// bsdf, light, vertexConnecting, cameraConnecting, vertexMerging
RTH_API int rth_viewport_set_vcm(void* v, uint32_t maxPathLength, int useVertexConnection, int useVertexMerging, float initialMergingRadius,
                                 float minMergingRadius, float mergingRadiusMultiplier, const float* weights)
{
    VertexConnectionAndMerging* r = dynamic_cast<VertexConnectionAndMerging*>(static_cast<ViewportHandle*>(v)->renderer.get());
    if (!r) return -3;
    r->mMaxPathLength = maxPathLength; r->mUseVertexConnection = useVertexConnection != 0; r->mUseVertexMerging = useVertexMerging != 0;
    r->mInitialMergingRadius = initialMergingRadius; r->mMinMergingRadius = minMergingRadius; r->mMergingRadiusMultiplier = mergingRadiusMultiplier;
    if (weights)
    {
        r->mBSDFSamplingWeight = Vector4(weights[0]); r->mLightSamplingWeight = Vector4(weights[1]); r->mVertexConnectingWeight = Vector4(weights[2]);
        r->mCameraConnectingWeight = Vector4(weights[3]); r->mVertexMergingWeight = Vector4(weights[4]);
    }
    return 0;
}
RTH_API int rth_viewport_set_debug_mode(void* v, uint32_t mode)
{
    DebugRenderer* r = dynamic_cast<DebugRenderer*>(static_cast<ViewportHandle*>(v)->renderer.get());
    if (!r || mode > (uint32_t)DebugRenderingMode::IoR) return -3;
    r->mRenderingMode = (DebugRenderingMode)mode;
    return 0;
}
RTH_API void rth_viewport_reset(void* v) { static_cast<ViewportHandle*>(v)->viewport.Reset(); }
RTH_API int rth_viewport_render(void* v, void* camera, uint32_t numPasses)
{
    ViewportHandle* vh = static_cast<ViewportHandle*>(v);
    for (uint32_t i = 0; i < numPasses; ++i) if (!vh->viewport.Render(*static_cast<Camera*>(camera))) return -1;
    return 0;
}
// out->seed points into storage owned by the viewport, valid until the next call
RTH_API int rth_viewport_next_pass_params(void* v, void* camera, RtPassParams* out)
{
    return static_cast<ViewportHandle*>(v)->viewport.NextPassParams(*static_cast<Camera*>(camera), *out) ? 0 : -1;
}
// one pass with explicit constants (uploads the scene on first use); the caller owns params->seed
RTH_API int rth_viewport_render_pass_with(void* v, const RtPassParams* params)
{
    ViewportHandle* vh = static_cast<ViewportHandle*>(v);
    if (!vh->renderer || !vh->renderer->RenderPass(*params)) return -1;
    vh->viewport.FinishPass();
    return 0;
}
RTH_API void rth_viewport_finish_pass(void* v) { static_cast<ViewportHandle*>(v)->viewport.FinishPass(); }
// Viewport::GetSumBuffer() alone: the accumulated frame is in the viewport's host bitmap when this returns (what bench.py times);
// rth_viewport_read_sum then only copies it out.
RTH_API int rth_viewport_fetch_sum(void* v) { (void)static_cast<ViewportHandle*>(v)->viewport.GetSumBuffer(); return 0; }
RTH_API int rth_viewport_read_sum(void* v, float* sum, float* secondary)
{
    ViewportHandle* vh = static_cast<ViewportHandle*>(v);
    if (sum) { const Bitmap& s = vh->viewport.GetSumBuffer(); memcpy(sum, s.GetData(), s.GetDataSize()); }
    if (secondary) { const Bitmap& s2 = vh->viewport.GetSecondarySumBuffer(); memcpy(secondary, s2.GetData(), s2.GetDataSize()); }
    return 0;
}
RTH_API int rth_viewport_counters(void* v, uint64_t out[16])
{
    const RayTracingCounters c = static_cast<ViewportHandle*>(v)->viewport.GetTotalCounters();
    memset(out, 0, 16 * sizeof(uint64_t));
    out[0] = c.numRays; out[1] = c.numShadowRays; out[2] = c.numShadowRaysHit; out[3] = c.numPrimaryRays;
    out[4] = c.numRayBoxTests; out[5] = c.numPassedRayBoxTests; out[6] = c.numRayTriangleTests; out[7] = c.numPassedRayTriangleTests;
    out[8] = c.numMeshHits; out[9] = c.numAnalyticHits; out[10] = c.numShadowRayBoxTests; out[11] = c.numShadowRayTriangleTests;
    return 0;
}
RTH_API void* rth_viewport_device_ctx(void* v)
{
    ViewportHandle* vh = static_cast<ViewportHandle*>(v);
    PathTracerMIS* pt = dynamic_cast<PathTracerMIS*>(vh->renderer.get());
    return pt ? pt->GetDeviceContext() : nullptr;
}
RTH_API int rth_viewport_set_shard(void* v, uint32_t rank, uint32_t world)
{
    ViewportHandle* vh = static_cast<ViewportHandle*>(v);
    PathTracerMIS* pt = dynamic_cast<PathTracerMIS*>(vh->renderer.get());
    if (!pt || !pt->SetShard(rank, world)) return -1;
    vh->viewport.ClearAccumulation();   // the device film was cleared with the ownership change: the host's sums, pass count and block list start over with it (the sample sequence does not)
    return 0;
}
RTH_API uint32_t rth_viewport_passes_finished(void* v) { return static_cast<ViewportHandle*>(v)->viewport.GetPassesFinished(); }
// adaptive rendering (RenderingParams::adaptiveSettings) and progress
RTH_API int rth_viewport_set_adaptive(void* v, int enable, uint32_t numInitialPasses, uint32_t minBlockSize, uint32_t maxBlockSize, float subdivisionTreshold, float convergenceTreshold)
{
    Viewport& vp = static_cast<ViewportHandle*>(v)->viewport;
    RenderingParams p = vp.GetRenderingParams();
    p.adaptiveSettings.enable = enable != 0; p.adaptiveSettings.numInitialPasses = numInitialPasses; p.adaptiveSettings.minBlockSize = minBlockSize;
    p.adaptiveSettings.maxBlockSize = maxBlockSize; p.adaptiveSettings.subdivisionTreshold = subdivisionTreshold; p.adaptiveSettings.convergenceTreshold = convergenceTreshold;
    if (!vp.SetRenderingParams(p)) return -1;
    vp.Reset();
    return 0;
}
// out: averageError, converged, activePixels, activeBlocks; blocks (may be NULL): up to maxBlocks x {minX, maxX, minY, maxY}; returns the block count
RTH_API int rth_viewport_progress(void* v, float* averageError, float* converged, uint32_t* activePixels, uint32_t* blocks, uint32_t maxBlocks)
{
    Viewport& vp = static_cast<ViewportHandle*>(v)->viewport;
    const RenderingProgress& p = vp.GetProgress();
    if (averageError) *averageError = p.averageError;
    if (converged) *converged = p.converged;
    if (activePixels) *activePixels = p.activePixels;
    const std::vector<RtBlock>& b = vp.GetBlocks();
    for (size_t i = 0; blocks && i < b.size() && i < maxBlocks; ++i) memcpy(blocks + 4 * i, &b[i], 16);
    return (int)b.size();
}

// Viewport::UpdateBlocksList's list walk on caller-supplied block errors (tests/golden/adaptive_kat.bin: the reference's own lists)
RTH_API int rth_viewport_kat_update_blocks(void* v, uint32_t passesFinished, const float* errors, uint32_t numErrors)
{
    Viewport& vp = static_cast<ViewportHandle*>(v)->viewport;
    if (numErrors != vp.GetBlocks().size()) return -1;
    vp.UpdateBlocksListWithErrors(passesFinished, std::vector<float>(errors, errors + numErrors));
    return (int)vp.GetBlocks().size();
}

// ---- known-answer-test entry points for the host-side algorithms -----------------------------------
// boxes: n * 6 floats (min xyz, max xyz).  outNodes: capacity 2n nodes of 8 uint32.  outOrder: n.
RTH_API int rth_kat_bvh_build(const float* boxes, uint32_t n, uint32_t* outNodes, uint32_t* outNumNodes, uint32_t* outOrder)
{
    std::vector<Box> b(n);
    for (uint32_t i = 0; i < n; ++i)
    {
        b[i].min = Vector4(boxes[6 * i + 0], boxes[6 * i + 1], boxes[6 * i + 2], 0.0f);
        b[i].max = Vector4(boxes[6 * i + 3], boxes[6 * i + 4], boxes[6 * i + 5], 0.0f);
    }
    BVH bvh;
    BVHBuilder builder(bvh);
    BVHBuilder::Indices order;
    if (!builder.Build(b.data(), n, BvhBuildingParams(), order)) return -1;
    *outNumNodes = bvh.GetNumNodes();
    if (bvh.GetNumNodes()) memcpy(outNodes, bvh.GetNodes(), (size_t)bvh.GetNumNodes() * 32);
    for (uint32_t i = 0; i < n; ++i) outOrder[i] = order[i];
    return 0;
}

// Halton seeds: state = scalar xoroshiro state of the sequence's private generator; out: numPasses * dims uint32
RTH_API int rth_kat_halton(const uint64_t scalarState[2], uint32_t dims, uint32_t numPasses, uint32_t* out)
{
    HaltonSequence h;
    const uint64_t simd[4] = { 1, 2, 3, 4 };
    h.GetRandom().SetState(scalarState, simd);
    h.Initialize(dims);
    for (uint32_t p = 0; p < numPasses; ++p)
    {
        h.NextSample();
        for (uint32_t d = 0; d < dims; ++d) out[(size_t)p * dims + d] = h.GetInt(d);
    }
    return 0;
}

// Random: outLongs[count] from GetLong, outVec4[count*4] from GetVector4 (independent streams of one state)
RTH_API int rth_kat_random(const uint64_t scalarState[2], const uint64_t simd4State[4], uint32_t count, uint64_t* outLongs, float* outVec4)
{
    Random r;
    r.SetState(scalarState, simd4State);
    for (uint32_t i = 0; i < count; ++i) outLongs[i] = r.GetLong();
    for (uint32_t i = 0; i < count; ++i)
    {
        const Vector4 v = r.GetVector4();
        outVec4[4 * i + 0] = v.x; outVec4[4 * i + 1] = v.y; outVec4[4 * i + 2] = v.z; outVec4[4 * i + 3] = v.w;
    }
    return 0;
}

RTH_API void rth_kat_float_normal2(float ux, float uy, float out[4])
{
    const Vector4 v = GetFloatNormal2(Float2(ux, uy));
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}

RTH_API void rth_set_global_seed(uint64_t seed) { Entropy::SetGlobalSeed(seed); }

} // extern "C"
