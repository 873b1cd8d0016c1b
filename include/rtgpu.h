/*
 * rtgpu.h -- C ABI of the MI355X path-tracing core (PathTracerMIS over the two-level BVH).
 *
 * This is the drop-in boundary.  The reference (Witek902/Raytracer) has no FFI: the integrator sits
 * behind the C++ virtual IRenderer::RenderPixel (Core/Rendering/Renderer.h:56) and is called once per
 * pixel from Viewport::RenderTile (Core/Rendering/Viewport.cpp:338).  A per-pixel virtual is not a GPU
 * boundary, so the seam is ONE PASS: everything Viewport::Render does between Viewport.cpp:200 and
 * Viewport.cpp:262 (per-pass sampler seeds -> tile fan-out -> RenderTile -> RenderPixel -> Film).
 *
 * Everything crossing this ABI is a plain pointer, size or POD struct.  No C++ types, no torch types,
 * no exceptions.  All functions return RTGPU_OK (0) or a negative RtgpuStatus; rtgpu_last_error()
 * returns a human readable message for the last failure on the calling thread.
 *
 * Matrices are 4 rows of 4 floats, row-vector convention exactly as rt::math::Matrix4
 * (Core/Math/Matrix4.h:20-27): point' = p.x*row0 + p.y*row1 + p.z*row2 + row3.
 */
#ifndef RTGPU_H
#define RTGPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTGPU_ABI_VERSION 3u

typedef enum RtgpuStatus
{
    RTGPU_OK = 0,
    RTGPU_ERR_INVALID_ARGUMENT = -1,
    RTGPU_ERR_NO_DEVICE = -2,
    RTGPU_ERR_OUT_OF_MEMORY = -3,
    RTGPU_ERR_DEVICE = -4,      /* a HIP call failed; see rtgpu_last_error() */
    RTGPU_ERR_NOT_READY = -5,   /* e.g. render_pass before upload_scene / resize */
    RTGPU_ERR_UNSUPPORTED = -6  /* feature outside the hot-path scope (textures, decals, CSG ...) */
} RtgpuStatus;

/* sentinel ids, Core/Traversal/HitPoint.h:8-9 */
#define RT_INVALID_OBJECT 0xFFFFFFFFu
#define RT_LIGHT_OBJECT   0xFFFFFFFEu
#define RT_NO_MATERIAL    0xFFFFFFFFu

/* ---------------------------------------------------------------------------------------------
 * Scene description (flat, read-only after upload).  Replaces the pointer graph the reference walks:
 * Scene::mTraceableObjects / mLights / mGlobalLights (Core/Scene/Scene.h:83-96), BVH::mNodes
 * (Core/BVH/BVH.h:22-30), VertexBuffer (Core/Shapes/Mesh/VertexBuffer.h:18-31), Material
 * (Core/Material/Material.h:47-73).
 * ------------------------------------------------------------------------------------------- */

/* 32-byte BVH node, bit-identical to rt::BVH::Node (Core/BVH/BVH.h:22-30).
 * leaves = numLeaves (low 30 bits) | splitAxis << 30.  numLeaves != 0 <=> leaf.
 * Interior: children are nodes[childIndex], nodes[childIndex+1].
 * Leaf: items [childIndex, childIndex + numLeaves) of the owning container
 * (scene objects for the top level, triangles for a mesh). */
typedef struct RtNode
{
    float    min[3];
    uint32_t childIndex;
    float    max[3];
    uint32_t leaves;
} RtNode;

/* 36 bytes, rt::math::ProcessedTriangle (Core/Math/Triangle.h:25-41) */
typedef struct RtTriangle
{
    float v0[3];
    float edge1[3];
    float edge2[3];
} RtTriangle;

/* 16 bytes, rt::VertexIndices (Core/Shapes/Mesh/VertexBuffer.h:18-24).
 * i0..i2 are relative to the owning mesh's first vertex; materialIndex is a GLOBAL index into
 * RtSceneDesc::materials or RT_NO_MATERIAL (=> the object's default material,
 * Core/Shapes/MeshShape.cpp:288-291). */
typedef struct RtVertexIndices
{
    uint32_t i0, i1, i2;
    uint32_t materialIndex;
} RtVertexIndices;

/* 32 bytes, rt::VertexShadingData (Core/Shapes/Mesh/VertexBuffer.h:26-31) */
typedef struct RtVertexShading
{
    float normal[3];
    float tangent[3];
    float texCoord[2];
} RtVertexShading;

typedef struct RtMesh
{
    uint32_t firstNode;     /* into RtSceneDesc::meshNodes; node childIndex values are mesh-relative */
    uint32_t numNodes;
    uint32_t firstTriangle; /* into triangles[] and vertexIndices[] (same order: BVH leaf order) */
    uint32_t numTriangles;
    uint32_t firstVertex;   /* into vertexShading[] */
    uint32_t numVertices;
    uint32_t _pad[2];
} RtMesh;

typedef enum RtShapeKind
{
    RT_SHAPE_SPHERE = 0, /* param = { radius, 1/radius, 0, 0 }            Core/Shapes/SphereShape.cpp:13 */
    RT_SHAPE_BOX    = 1, /* param = { size.xyz (half extents), 0 }        Core/Shapes/BoxShape.cpp:91   */
    RT_SHAPE_RECT   = 2, /* param = { size.x, size.y, texScale.x, .y }    Core/Shapes/RectShape.cpp:14  */
    RT_SHAPE_MESH   = 3  /* meshIndex valid                               Core/Shapes/MeshShape.cpp:34  */
} RtShapeKind;

typedef enum RtObjectKind
{
    RT_OBJECT_SHAPE = 0, /* rt::ShapeSceneObject  Core/Scene/Object/SceneObject_Shape.cpp */
    RT_OBJECT_LIGHT = 1  /* rt::LightSceneObject  Core/Scene/Object/SceneObject_Light.cpp (finite lights only) */
} RtObjectKind;

/* One traceable scene object, in top-level-BVH leaf order (Core/Scene/Scene.cpp:87-95). */
typedef struct RtObject
{
    float    transform[16];     /* local -> world, ISceneObject::mTransform         */
    float    invTransform[16];  /* world -> local, Matrix4::Inverse() of the above  (SceneObject.cpp:22).  Pass the reference's OWN inverse (GetInverseTransform()):
                                   the library never recomputes it, and a differently rounded inverse changes rendered bits (1-2 ulp under general rotations) */
    uint32_t objectKind;        /* RtObjectKind */
    uint32_t shapeKind;         /* RtShapeKind (for lights: shape of the area light; unused for point/spot) */
    uint32_t materialIndex;     /* default material (shapes); RT_NO_MATERIAL for lights */
    uint32_t meshIndex;         /* RT_SHAPE_MESH only */
    uint32_t lightIndex;        /* RT_OBJECT_LIGHT only: index into RtSceneDesc::lights */
    uint32_t _pad[3];
    float    shapeParam[4];
    float    shapeParam2[4];    /* box: 1/size (w=0); others unused */
} RtObject;

typedef enum RtLightType /* rt::ILight::Type, Core/Scene/Light/Light.h:27-34 */
{
    RT_LIGHT_AREA = 0,
    RT_LIGHT_BACKGROUND = 1,
    RT_LIGHT_DIRECTIONAL = 2,
    RT_LIGHT_POINT = 3,
    RT_LIGHT_SPOT = 4
} RtLightType;

#define RT_LIGHT_FLAG_FINITE 1u /* ILight::Flag_IsFinite */
#define RT_LIGHT_FLAG_DELTA  2u /* ILight::Flag_IsDelta  */

/* One light, in Scene::mLights order (= AddObject order, Core/Scene/Scene.cpp:45-48). */
typedef struct RtLight
{
    float    transform[16];
    float    invTransform[16];
    float    color[4];       /* Spectrum::rgbValues; all FOUR lanes are significant (RayColor::AlmostZero) */
    uint32_t type;           /* RtLightType */
    uint32_t flags;          /* RT_LIGHT_FLAG_* as returned by ILight::GetFlags() */
    uint32_t shapeKind;      /* area lights: RT_SHAPE_SPHERE / BOX / RECT */
    uint32_t isDelta;        /* directional / spot: mIsDelta (cos > 0.9999, Light.h:25) */
    float    cosAngle;       /* directional / spot: cosf(angle) computed by the host */
    uint32_t texture;        /* background light: BackgroundLight::mTexture (environment map, BackgroundLight.cpp:45-61)
                              * as an index into RtSceneDesc::textures, or RT_NO_TEXTURE */
    float    _pad[2];
    float    shapeParam[4];
    float    shapeParam2[4];
} RtLight;

typedef enum RtBsdf /* string names of Material::SetBsdf, Core/Material/Material.cpp:40-83 */
{
    RT_BSDF_NULL = 0,
    RT_BSDF_DIFFUSE = 1,
    RT_BSDF_ROUGH_DIFFUSE = 2,
    RT_BSDF_DIELECTRIC = 3,
    RT_BSDF_ROUGH_DIELECTRIC = 4,
    RT_BSDF_METAL = 5,
    RT_BSDF_ROUGH_METAL = 6,
    RT_BSDF_PLASTIC = 7,
    RT_BSDF_ROUGH_PLASTIC = 8
} RtBsdf;

#define RT_NO_TEXTURE 0xFFFFFFFFu

/* ITexture implementations that can sit on the shading path (Core/Textures/). */
typedef enum RtTextureKind
{
    RT_TEXTURE_BITMAP = 0,        /* BitmapTexture  (BitmapTexture.cpp:32-93) */
    RT_TEXTURE_CHECKERBOARD = 1,  /* CheckerboardTexture (CheckerboardTexture.cpp:31-40): colorA / colorB */
    RT_TEXTURE_CONST = 2,         /* ConstTexture: colorA */
    RT_TEXTURE_NOISE = 3,         /* NoiseTexture (NoiseTexture.cpp:58-164): simplex noise octaves, Lerp(colorA, colorB, value) */
    RT_TEXTURE_MIX = 4            /* MixTexture (MixTexture.cpp:23-30): Lerp(textures[mixA], textures[mixB], textures[mixWeight]);
                                   * a mix may reference mixes one level deep (rtgpu_upload_scene rejects deeper nesting) */
} RtTextureKind;

/* Texel formats: the values of rt::Bitmap::Format (Core/Utils/Bitmap.h:15-41), every one decoded on the device exactly
 * as Bitmap::GetPixel / GetPixelBlock do (Bitmap.cpp:335-832, Math/Packed.h, Utils/BlockCompression.cpp).  The
 * block-compressed formats address whole 4 x 4 blocks (width / 4 blocks per row, like the reference). */
typedef enum RtBitmapFormat
{
    RT_FORMAT_R8_UNORM = 1, RT_FORMAT_R8G8_UNORM = 2, RT_FORMAT_B8G8R8_UNORM = 3, RT_FORMAT_B8G8R8A8_UNORM = 4,
    RT_FORMAT_R8G8B8A8_UNORM = 5, RT_FORMAT_B8G8R8A8_UNORM_PALETTE = 6, RT_FORMAT_B5G6R5_UNORM = 7,
    RT_FORMAT_R16_UNORM = 8, RT_FORMAT_R16G16_UNORM = 9, RT_FORMAT_R16G16B16A16_UNORM = 10,
    RT_FORMAT_R32_FLOAT = 11, RT_FORMAT_R32G32_FLOAT = 12, RT_FORMAT_R32G32B32_FLOAT = 13, RT_FORMAT_R32G32B32A32_FLOAT = 14,
    RT_FORMAT_R11G11B10_FLOAT = 15,
    RT_FORMAT_R16_HALF = 16, RT_FORMAT_R16G16_HALF = 17, RT_FORMAT_R16G16B16_HALF = 18, RT_FORMAT_R16G16B16A16_HALF = 19,
    RT_FORMAT_R9G9B9E5_SHAREDEXP = 20, RT_FORMAT_BC1 = 21, RT_FORMAT_BC4 = 22, RT_FORMAT_BC5 = 23
} RtBitmapFormat;

typedef enum RtTextureFilter /* BitmapTextureFilter, Core/Textures/BitmapTexture.h */
{
    RT_FILTER_NEAREST = 0, RT_FILTER_BILINEAR = 1, RT_FILTER_BILINEAR_SMOOTHSTEP = 2
} RtTextureFilter;

/* 96 bytes */
typedef struct RtTexture
{
    uint32_t kind;          /* RtTextureKind */
    uint32_t format;        /* RtBitmapFormat */
    uint32_t width, height;
    uint32_t stride;        /* bytes per row, Bitmap::mStride (unused by the block-compressed formats) */
    uint32_t linearSpace;   /* Bitmap::mLinearSpace; 0 => Convert_sRGB_To_Linear on every fetched texel (all four lanes) */
    uint32_t filter;        /* RtTextureFilter */
    uint32_t numOctaves;    /* noise */
    uint64_t dataOffset;    /* byte offset of the first row (or block) in RtSceneDesc::texelData */
    uint64_t paletteOffset; /* palette format: byte offset of the B8G8R8A8 palette entries in texelData */
    float    colorA[4];     /* checkerboard / const / noise */
    float    colorB[4];
    uint32_t mixA, mixB, mixWeight;   /* mix: indices into RtSceneDesc::textures */
    uint32_t _pad;
} RtTexture;

/* 80 bytes; rt::Material after Compile() (Core/Material/Material.cpp:105-117): scalar parameters, the textures of the
 * four MaterialParameters (value = baseValue * texture->Evaluate(uv), MaterialParameter.h:22-32) and the normal map
 * (Material::GetNormalVector, Material.cpp:120-138; applied in Scene::EvaluateIntersection, Scene.cpp:327-337). */
typedef struct RtMaterial
{
    float    emission[4];   /* all four lanes significant */
    float    baseColor[4];
    float    roughness;
    float    metalness;
    float    IoR;
    float    K;
    uint32_t bsdf;          /* RtBsdf */
    uint32_t baseColorTexture;   /* indices into RtSceneDesc::textures, or RT_NO_TEXTURE */
    uint32_t emissionTexture;
    uint32_t roughnessTexture;
    uint32_t metalnessTexture;
    uint32_t normalMapTexture;
    float    normalMapStrength;
    uint32_t _pad;
} RtMaterial;

typedef struct RtSceneDesc
{
    uint32_t abiVersion;           /* RTGPU_ABI_VERSION */
    uint32_t numObjects;           /* traceable objects (shapes + finite lights) */
    uint32_t numTopNodes;          /* 0 when numObjects == 0 */
    uint32_t numLights;
    uint32_t numGlobalLights;
    uint32_t numMaterials;
    uint32_t numMeshes;
    uint32_t numMeshNodes;
    uint32_t numTriangles;
    uint32_t numVertices;
    uint32_t numTextures;
    uint32_t _pad;

    const RtNode*          topNodes;       /* [numTopNodes]  Scene::mTraceableObjectsBVH */
    const RtObject*        objects;        /* [numObjects]   */
    const RtLight*         lights;         /* [numLights]    Scene::mLights */
    const uint32_t*        globalLights;   /* [numGlobalLights] indices into lights[], Scene::mGlobalLights */
    const RtMaterial*      materials;      /* [numMaterials] */
    const RtMesh*          meshes;         /* [numMeshes]    */
    const RtNode*          meshNodes;      /* [numMeshNodes] */
    const RtTriangle*      triangles;      /* [numTriangles] */
    const RtVertexIndices* vertexIndices;  /* [numTriangles] */
    const RtVertexShading* vertexShading;  /* [numVertices]  */
    /* 128*128*4 uint16 = 131072 bytes, contents of Data/BlueNoise128_RGBA16.dat
     * (Core/Sampling/GenericSampler.cpp:13-52).  NULL => blue-noise dithering silently off (:72). */
    const uint16_t*        blueNoise;
    const RtTexture*       textures;       /* [numTextures] */
    const uint8_t*         texelData;      /* [texelBytes] rows of all bitmap textures (RtTexture::dataOffset points into it) */
    uint64_t               texelBytes;
} RtSceneDesc;

/* ---------------------------------------------------------------------------------------------
 * Per-pass constants: what Viewport::Render computes on the host before the tile fan-out
 * (Core/Rendering/Viewport.cpp:200-242) plus the RenderingParams fields the path reads
 * (Core/Rendering/Context.h:55-90) and the camera (Core/Scene/Camera.cpp:81-118).
 * ------------------------------------------------------------------------------------------- */

#define RTGPU_MAX_DIMENSIONS 4096u /* HaltonSequence::MaxDimensions */

typedef struct RtCamera
{
    float    localToWorld[16];   /* Camera::mLocalToWorld */
    float    aspectRatio;
    float    tanHalfFoV;         /* tanf(fov/2) computed by the host (Camera.cpp:37) */
    uint32_t dofEnable;
    uint32_t bokehShape;         /* rt::BokehShape (Camera.h:21-28): 0 circle, 1 hexagon, 2 square (NGon and Texture are refused) */
    float    focalPlaneDistance;
    float    aperture;
    float    barrelDistortionConstFactor;      /* Camera.cpp:86-91: applied when the variable factor is not zero; its random */
    float    barrelDistortionVariableFactor;   /* draw (ctx.randomGenerator.GetFloat()) comes from the per-pixel generator    */
    float    worldToScreen[16];  /* Camera::mWorldToScreen as SetPerspective left it (Camera.cpp:39-48): read by the
                                  * bidirectional integrator only (Camera::WorldToFilm, Camera.cpp:120-134) */
} RtCamera;

typedef enum RtLightSampling { RT_LIGHT_SAMPLING_SINGLE = 0, RT_LIGHT_SAMPLING_ALL = 1 } RtLightSampling;

typedef struct RtPassParams
{
    RtCamera camera;
    const uint32_t* seed;          /* [numDimensions] HaltonSequence::GetInt(d), Viewport.cpp:200-205 */
    uint32_t numDimensions;        /* SamplingParams::dimensions */
    uint32_t useBlueNoise;         /* SamplingParams::useBlueNoiseDithering */
    float    sampleOffset[2];      /* GetFloatNormal2(u) * antiAliasingSpread, Viewport.cpp:235-242 */
    uint32_t passIndex;            /* mProgress.passesFinished; even => also accumulate secondary sum */
    uint32_t maxRayDepth;
    uint32_t minRussianRouletteDepth;
    uint32_t lightSamplingStrategy;/* RtLightSampling */
    float    lightSamplingWeight[4]; /* PathTracerMIS::mLightSamplingWeight */
    float    bsdfSamplingWeight[4];  /* PathTracerMIS::mBSDFSamplingWeight  */
    /* Key of the per-pixel fallback generator.  The reference draws light picks
     * (PathTracerMIS.cpp:136) and samples past numDimensions (GenericSampler.cpp:108) from a
     * PER-THREAD xoroshiro128+ whose consumption order depends on dynamic tile scheduling; here every
     * pixel owns a xoroshiro128+ stream seeded from (rngKey, x, y) so the result is schedule-free. */
    uint64_t rngKey[2];
} RtPassParams;

/* rt::RayTracingCounters (Core/Rendering/Counters.h:36-93), intersection counters always on. */
typedef struct RtCounters
{
    uint64_t numRays;            /* sum over paths of depth+1  (PathTracerMIS.cpp:412) -- THE metric */
    uint64_t numShadowRays;
    uint64_t numShadowRaysHit;   /* reference naming: shadow rays that reached the light */
    uint64_t numPrimaryRays;
    /* The next four count CLOSEST-HIT traversals only, like the reference: Scene::Traverse resets and appends
     * the local counters (Scene.cpp:223,242) while Scene::Traverse_Shadow does neither, so the tests done by
     * shadow rays never reach RayTracingCounters there. */
    uint64_t numRayBoxTests;
    uint64_t numPassedRayBoxTests;
    uint64_t numRayTriangleTests;
    uint64_t numPassedRayTriangleTests;
    /* additions used by the roofline model */
    uint64_t numMeshHits;        /* EvaluateIntersection on a mesh triangle */
    uint64_t numAnalyticHits;    /* EvaluateIntersection on sphere/box/rect (incl. area lights) */
    uint64_t numShadowRayBoxTests;      /* box tests done by shadow rays (not counted by the reference) */
    uint64_t numShadowRayTriangleTests; /* triangle tests done by shadow rays */
    uint64_t numRetracedRays;    /* rays the default traversal kernel of single-mesh scenes did not trust (runner-up hit within its tolerance, NaN slab tests) and
                                  * handed to the binary-tree kernel, which walks the reference's order (performance statistic) */
    uint64_t _reserved[3];
} RtCounters;

typedef struct RtgpuContext RtgpuContext;

/* Optional tile-interleaved ownership for multi-GPU: this context renders only pixels whose 64x64
 * tile index (row-major) satisfies tile % worldSize == rank.  Non-owned pixels of the sum buffers
 * stay zero, so a sum-reduction (or gather) over ranks reproduces the 1-GPU image bit-exactly. */
typedef struct RtgpuShard
{
    uint32_t rank;
    uint32_t worldSize;
} RtgpuShard;

/* --- lifetime --------------------------------------------------------------------------------*/
/* Creates a context on HIP device `deviceIndex` (one context = one device = one host thread at a time). */
int  rtgpu_create(int deviceIndex, RtgpuContext** outCtx);
/* One context over several devices of a node: what the reference's ThreadPool does with the tiles of a frame across the cores of a
 * machine (Viewport.cpp:244-262, ThreadPool.cpp:176-260).  The frame's 64x64 tiles are dealt round-robin to the devices (RtgpuShard),
 * the scene is replicated by rtgpu_upload_scene, rtgpu_render_pass queues the pass on every device asynchronously, and the read-back
 * calls (rtgpu_read_sum, rtgpu_get_device_sum, rtgpu_postprocess, rtgpu_compute_block_errors) first gather the peers' tiles into the
 * first device's buffers (a kernel that reads the peers' HBM over xGMI; hipMemcpyPeerAsync staging without peer access or with
 * RTGPU_MULTI_STAGED=1).  Results are bit-identical to a one-device context; counters are summed.  deviceIndices == NULL: the first
 * numDevices visible devices (0 = all).  An index may repeat (two shards on one device: how the path is tested on a 1-GPU box).
 * Unsupported on such a context: rtgpu_set_shard, RT_INTEGRATOR_VCM / _LIGHT_TRACER (they splat over the whole frame);
 * rtgpu_get_kernel_times reports the first device. */
int  rtgpu_create_multi(const int* deviceIndices, uint32_t numDevices, RtgpuContext** outCtx);
int  rtgpu_num_devices(RtgpuContext* ctx, uint32_t* outCount);
void rtgpu_destroy(RtgpuContext* ctx);
const char* rtgpu_last_error(void);
uint32_t rtgpu_abi_version(void);

/* --- scene: replaces Scene::BuildBVH's pointer graph; copies everything, host memory may be freed */
int rtgpu_upload_scene(RtgpuContext* ctx, const RtSceneDesc* scene);

/* --- film: Viewport::Resize (Viewport.cpp:52-107) / Viewport::Reset (:120-138) ------------------*/
int rtgpu_resize(RtgpuContext* ctx, uint32_t width, uint32_t height);
int rtgpu_set_shard(RtgpuContext* ctx, RtgpuShard shard);
int rtgpu_reset(RtgpuContext* ctx);

/* --- the hot path: one pass over every owned pixel (Viewport.cpp:244-262 -> RenderTile :291-357 ->
 *     PathTracerMIS::RenderPixel PathTracerMIS.cpp:254-415 -> Film::AccumulateColor Film.cpp:31-39).
 *     Asynchronous on the context's stream; the params (incl. seed[]) are copied before return. */
int rtgpu_render_pass(RtgpuContext* ctx, const RtPassParams* params);

/* Blocks until all queued passes have finished. */
int rtgpu_synchronize(RtgpuContext* ctx);

/* --- readback: Viewport::GetSumBuffer (R32G32B32_Float, tight stride; row y of the bitmap is film
 *     row H-1-y, Viewport.cpp:309,354).  sumRGB / secondaryRGB: width*height*3 floats, either may be
 *     NULL.  Synchronises. */
int rtgpu_read_sum(RtgpuContext* ctx, float* sumRGB, float* secondaryRGB);

/* Optional: page-locks (hipHostRegister) a host buffer the caller passes to rtgpu_read_sum / rtgpu_postprocess again and again, so
 * that the copies run at the PCIe link's rate (a 1080p float3 frame: ~0.6 ms instead of ~3 ms).  Unregister before freeing the
 * buffer.  RTGPU_ERR_NO_DEVICE / RTGPU_ERR_DEVICE when it cannot be done: the buffer then simply stays pageable. */
int rtgpu_host_register(RtgpuContext* ctx, void* ptr, size_t bytes);
int rtgpu_host_unregister(RtgpuContext* ctx, void* ptr);

/* Device pointers of the float3 sum buffers (for RCCL gather/reduce by the caller).  Synchronises first (and gathers, on a
 * multi-device context): the buffers hold every pass queued so far. */
int rtgpu_get_device_sum(RtgpuContext* ctx, void** sumDevice, void** secondaryDevice, size_t* numFloats);

/* Counters accumulated since the last rtgpu_reset (Viewport::GetCounters is per pass; callers
 * difference two reads).  Synchronises. */
int rtgpu_get_counters(RtgpuContext* ctx, RtCounters* out);

/* Box / triangle test counters (numRayBoxTests ... numShadowRayTriangleTests).  In the reference they exist only
 * under the compile-time switch RT_ENABLE_INTERSECTION_COUNTERS (Core/Config.h:4, off by default); here they are a
 * run-time switch, OFF by default like there (the environment variable RTGPU_INTERSECTION_COUNTERS=1 turns them on for
 * contexts created afterwards).  They are the counters of the REFERENCE'S walk: with the switch on, every ray walks the binary
 * tree in the reference's order (k_trace); with it off, single-mesh scenes walk the 4-wide collapse of the same tree
 * (k_trace_wide: same hits, another visiting order; RTGPU_WIDE=0 keeps the binary walk).  numRays / numShadowRays /
 * numShadowRaysHit / numPrimaryRays / hit counts are always maintained.  Synchronises. */
int rtgpu_set_intersection_counters(RtgpuContext* ctx, int enable);

/* ---------------------------------------------------------------------------------------------
 * Integrator selection.  RT_INTEGRATOR_PATH_TRACER_MIS (default) is rt::PathTracerMIS; RT_INTEGRATOR_VCM is
 * rt::VertexConnectionAndMerging (Core/Rendering/VertexConnectionAndMerging.cpp, renderer name "VCM"): per pixel one light
 * sub-path (light vertices connected to the camera -> film splats, photons recorded for the next pass) and one camera
 * sub-path (light hits, next event estimation, connections to the pixel's light vertices, merging with the previous pass's
 * photons through the hash grid of Core/Utils/HashGrid.h).  RtVcmParams carries the public knobs of the class
 * (VertexConnectionAndMerging.h:35-53) with the constructor's defaults (.cpp:53-71).  RtPassParams::maxRayDepth,
 * lightSamplingStrategy, the Russian-roulette depth and the two weights are ignored by VCM (the reference's class does not
 * read them); passIndex == 0 restarts the merging radius and drops the recorded photons (PreRender, .cpp:84-138).
 * The draws the reference takes from the per-thread generator come from a per-pixel generator keyed by
 * RtPassParams::rngKey; film splats are float atomics (their summation order is not defined -- in the reference neither).
 * VCM needs the whole frame on one device: shard {0, 1} and no active-block restriction.  Synchronises. */
typedef enum RtIntegrator
{
    RT_INTEGRATOR_PATH_TRACER_MIS = 0,
    RT_INTEGRATOR_VCM = 1,
    RT_INTEGRATOR_PATH_TRACER = 2,  /* rt::PathTracer ("Path Tracer", Core/Rendering/PathTracer.cpp): BSDF sampling only -- no next event
                                     * estimation, no MIS; lightSamplingStrategy and the two weights of RtPassParams are ignored */
    RT_INTEGRATOR_DEBUG = 3,        /* rt::DebugRenderer ("Debug", Core/Rendering/DebugRenderer.cpp): one colour per pixel from the primary
                                     * ray's first hit, selected by rtgpu_set_debug_rendering_mode (default: TriangleID) */
    RT_INTEGRATOR_LIGHT_TRACER = 4  /* rt::LightTracer ("Light Tracer", Core/Rendering/LightTracer.cpp): one light path per pixel, every vertex
                                     * below RtPassParams::maxRayDepth connected to the camera (film splats); same per-pixel generator
                                     * convention and whole-frame requirement as VCM */
} RtIntegrator;
/* rt::DebugRenderingMode (Core/Rendering/DebugRenderer.h:7-33; the four intersection-counter modes exist in the reference only under
 * RT_ENABLE_INTERSECTION_COUNTERS, which is off, and are not provided) */
typedef enum RtDebugRenderingMode
{
    RT_DEBUG_CAMERA_LIGHT = 0, RT_DEBUG_TRIANGLE_ID, RT_DEBUG_DEPTH, RT_DEBUG_POSITION, RT_DEBUG_NORMALS, RT_DEBUG_TANGENTS, RT_DEBUG_BITANGENTS,
    RT_DEBUG_TEXCOORDS, RT_DEBUG_BASE_COLOR, RT_DEBUG_EMISSION, RT_DEBUG_ROUGHNESS, RT_DEBUG_METALNESS, RT_DEBUG_IOR
} RtDebugRenderingMode;
typedef struct RtVcmParams
{
    uint32_t maxPathLength;            /* mMaxPathLength = 10 */
    uint32_t useVertexConnection;      /* mUseVertexConnection = true */
    uint32_t useVertexMerging;         /* mUseVertexMerging = true */
    float    initialMergingRadius;     /* 0.02 */
    float    minMergingRadius;         /* 0.02 */
    float    mergingRadiusMultiplier;  /* 1.0 */
    uint32_t _pad[2];
    float    bsdfSamplingWeight[4];        /* mBSDFSamplingWeight      = 1 */
    float    lightSamplingWeight[4];       /* mLightSamplingWeight     = 1 */
    float    vertexConnectingWeight[4];    /* mVertexConnectingWeight  = 1 */
    float    cameraConnectingWeight[4];    /* mCameraConnectingWeight  = 1 */
    float    vertexMergingWeight[4];       /* mVertexMergingWeight     = 1 */
} RtVcmParams;
#define RT_VCM_MAX_PATH_LENGTH 16u
int rtgpu_set_integrator(RtgpuContext* ctx, uint32_t integrator, const RtVcmParams* vcm /* NULL: defaults */);
/* DebugRenderer::mRenderingMode; needs RT_INTEGRATOR_DEBUG.  Synchronises. */
int rtgpu_set_debug_rendering_mode(RtgpuContext* ctx, uint32_t mode);
/* number of photons recorded by the last VCM pass (the merge set of the next one).  Synchronises. */
int rtgpu_vcm_num_photons(RtgpuContext* ctx, uint32_t* outCount);

/* Batch lanes (1..6, default 4).  rtgpu_render_pass gathers passes into batches; consecutive batches run on
 * alternating HIP streams with their own path-state arenas, so the drain of one batch's traversal launches (a few
 * very long rays) overlaps with the next batch's kernels.  The film is still summed in pass order.  Performance
 * only: results do not depend on it.  1 = strictly serial kernels (what per-kernel timing wants).  Synchronises.
 * Replaces the thread-pool width of the reference (RenderingParams::numThreads, Viewport.cpp:44-50). */
int rtgpu_set_concurrency(RtgpuContext* ctx, uint32_t lanes);

/* Launch-sequence knobs of the PathTracerMIS pipeline (round 4).  Performance only: results do not depend on them (the parity tests run
 * every setting against the oracle).  Synchronises.
 *   RTGPU_SCHEDULE_TAIL_BOUNCE   the bounce at which a batch with dense path state hands its remaining paths to the fused tail kernel
 *                                (k_tail, rt_tail.hip: trace + the reference's own walk + shade in one persistent launch, per block);
 *                                0 = never, -1 = the library's policy (frames under 700 k owned pixels: bounce 6, else never).
 *   RTGPU_SCHEDULE_LOCAL_RETRACE 1 = a block of the 4-wide walks traces the rays it does not decide itself instead of handing them to a
 *                                launch of their own, 0 = never, -1 = the library's policy (frames under 400 k owned pixels). */
enum { RTGPU_SCHEDULE_TAIL_BOUNCE = 0, RTGPU_SCHEDULE_LOCAL_RETRACE = 1 };
int rtgpu_set_schedule(RtgpuContext* ctx, uint32_t knob, int32_t value);

/* ---------------------------------------------------------------------------------------------
 * Post-processing of the sum buffer into the displayable front buffer: Viewport::PostProcessTile
 * (Core/Rendering/Viewport.cpp:495-550) per pixel -- scale by 1 / numPasses, saturation, contrast as
 * FastExp(FastLog(c) * contrast), exposure and colour filter, tone mapping (Core/Color/ColorHelpers.h:78-132),
 * dithering, Vector4::ToBGR().  Bloom (bloomFactor > 0): the five blurred copies of the sum buffer of
 * Viewport::PerformPostProcess (:432-452; Bitmap::GaussianBlur, Core/Utils/Bitmap.cpp:880-1020, sigma = 2 * 2.5^i, 8 box
 * blurs per direction as running sums in the reference's order) are rebuilt from the current sum buffer by every call and
 * mixed in as PostProcessTile does (:512-524).  The reference's blur reads and writes out of bounds unless the width is a
 * multiple of 4, both sizes are <= 4096 and > 195 (its widest window): other sizes return RTGPU_ERR_UNSUPPORTED with
 * bloomFactor > 0.  Dithering uses a per-pixel hash of (x, y, ditherSeed) instead of the reference's
 * per-thread generator (which makes the reference's own front buffer thread-schedule dependent).
 * --------------------------------------------------------------------------------------------- */
typedef enum RtTonemapper { RT_TONEMAPPER_CLAMPED = 0, RT_TONEMAPPER_REINHARD = 1, RT_TONEMAPPER_HEJL_BURGESS_DAWSON = 2, RT_TONEMAPPER_ACES = 3 } RtTonemapper;

typedef struct RtPostprocessParams   /* PostprocessParams, Core/Rendering/PostProcess.h:10-28 (defaults: PostProcess.cpp:6-14) */
{
    float    colorFilter[4];
    float    exposure;            /* log2 scale: colorScale = colorFilter * 2^exposure (Viewport.cpp:455) */
    float    contrast;
    float    saturation;
    float    ditheringStrength;
    float    bloomFactor;         /* PostprocessParams::bloomFactor (0 = off) */
    uint32_t tonemapper;          /* RtTonemapper */
    uint32_t numPasses;           /* pixelScaling = 1 / numPasses (1 + passesFinished at the time of the call, :502) */
    uint32_t ditherSeed;
} RtPostprocessParams;

/* frontBufferBGRA: host, width * height uint32 (0x00RRGGBB), row y = sum-buffer row y.  Synchronises. */
int rtgpu_postprocess(RtgpuContext* ctx, const RtPostprocessParams* params, uint32_t* frontBufferBGRA);

/* ---------------------------------------------------------------------------------------------
 * Adaptive rendering support (Viewport::ComputeBlockError / UpdateBlocksList, Core/Rendering/Viewport.cpp:552-700).
 * The block list lives on the host (the mirror's rt::Viewport keeps the reference's splitting logic); the device
 * computes the error estimates and restricts the passes to the pixels of the active blocks.
 * --------------------------------------------------------------------------------------------- */
typedef struct RtBlock { uint32_t minX, maxX, minY, maxY; } RtBlock;   /* Block: [minX, maxX) x [minY, maxY) in sum-buffer coordinates */

/* outErrors[i] = Viewport::ComputeBlockError(blocks[i]) with imageScalingFactor = 1 / numPasses: per pixel
 * (|a-b|.x + 2 |a-b|.y + |a-b|.z) / sqrt(RT_EPSILON + a.x + 2 a.y + a.z), a = sum / n, b = 2 secondarySum / n, summed
 * row by row in the reference's order, times sqrt(blockArea / imageArea) / blockArea.  Synchronises. */
int rtgpu_compute_block_errors(RtgpuContext* ctx, uint32_t numPasses, uint32_t numBlocks, const RtBlock* blocks, float* outErrors);

/* Restricts the following passes to the pixels covered by the blocks (intersected with the shard's tiles);
 * numBlocks = 0 restores the whole image.  Blocks must not overlap.  Synchronises. */
int rtgpu_set_active_blocks(RtgpuContext* ctx, uint32_t numBlocks, const RtBlock* blocks);

/* Evaluates textures of the uploaded scene on the device: out[4*i..] = ITexture::Evaluate(textures[textureIndex[i]],
 * (uv[2*i], uv[2*i+1])).  Host pointers; synchronous.  Exists so that the device decode of every texel format can be
 * checked against the reference's vectors directly (tests/golden/texture_kat.bin). */
int rtgpu_evaluate_textures(RtgpuContext* ctx, uint32_t count, const uint32_t* textureIndex, const float* uv, float* out);

/* Known-answer-test hooks.  They evaluate the DEVICE implementation of one hot-path function (the code the traversal and shading
 * kernels call, rt_device_*.h) on caller-provided records, so that tests can hold the HIP functions directly against vectors produced
 * by the reference's own translation units (tests/golden/) without going through any CPU
 * restatement.  Host pointers, synchronous, not performance relevant.
 *   rtgpu_kat         func = function id of the .kat fixture header (tests/golden/README.md: Sin/SinCos/FastLog/FastACos/FastATan2,
 *                     SamplingHelpers, BuildOrthonormalBasis, Fresnel*, Refract3/Reflect3, Ray::Ray, TransformRay_Unsafe,
 *                     FastInverseNoScale, Intersect_BoxRay(_TwoSided), Intersect_TriangleRay, shape Intersect/Sample/Pdf/
 *                     EvaluateIntersection, ILight::Illuminate/GetRadiance/Emit, BSDF::Sample/Evaluate/Pdf, Camera::GenerateRay/
 *                     WorldToFilm/PdfW, Film splat pixel, Packed* photons, DebugRenderer triangle colour); in: n records of inStride
 *                     floats (integers bit-cast), out: n records of outStride floats.
 *   rtgpu_kat_sampler GenericSampler::ResetPixel + GetInt/GetFloat (Core/Sampling/GenericSampler.cpp:69-113): record =
 *                     {x, y, useBlueNoise, numDims, seed[numDims]}; count draws per record.  blueNoise: 128*128*4 uint16 or NULL.
 *   rtgpu_kat_mesh    MeshShape::Traverse / Traverse_Shadow / EvaluateIntersection (Core/Shapes/MeshShape.cpp:134-328) through the
 *                     traversal state machine of the path tracer, on the uploaded scene's single mesh object: rays = n * {origin[3],
 *                     direction[3], tmax}, out = n * 19 words {objectId (7 on a hit), triangle, distance, u, v, anyHit, tangent[4],
 *                     normal[4], texCoord[4], material}. */
int rtgpu_kat(RtgpuContext* ctx, uint32_t func, const float* in, uint32_t inStride, float* out, uint32_t outStride, uint32_t n);
int rtgpu_kat_sampler(RtgpuContext* ctx, const uint16_t* blueNoise, const uint32_t* in, uint32_t inStride, uint32_t count, uint32_t n,
                      uint32_t* outInts, float* outFloats);
int rtgpu_kat_mesh(RtgpuContext* ctx, const float* rays, uint32_t n, uint32_t* out);

/* --- measurement hooks (bench.py) ---------------------------------------------------------------
 * Per-kernel-class GPU time in milliseconds accumulated since rtgpu_reset, measured with HIP events
 * on the stream each kernel is launched on when timing is enabled (with more than one batch lane the kernels of
 * different batches overlap, so the classes' times can add up to more than the wall time).  names[i] are static
 * strings. */
#define RTGPU_NUM_KERNEL_CLASSES 8
int rtgpu_enable_timing(RtgpuContext* ctx, int enable);
int rtgpu_get_kernel_times(RtgpuContext* ctx, double ms[RTGPU_NUM_KERNEL_CLASSES],
                           uint64_t launches[RTGPU_NUM_KERNEL_CLASSES],
                           const char* names[RTGPU_NUM_KERNEL_CLASSES]);

/* Which traversal kernel serves the uploaded scene with the intersection counters off, and the bytes its walk fetches from: the node
 * records, the leaves' exact boxes and the triangles (what decides where the memory system serves the walk's divergent fetches from;
 * bench.py prices the kernel's access rate against the microbenchmark's figure for this footprint).  Diagnostic, like the timing calls. */
#define RTGPU_WALK_BINARY 0u   /* k_trace: the reference's binary tree (also every scene with the counters on) */
#define RTGPU_WALK_WIDE   1u   /* k_trace_wide: 4-wide collapse of a single mesh's tree */
#define RTGPU_WALK_WIDE2  2u   /* k_trace_wide2: 4-wide top level over 4-wide mesh trees */
typedef struct RtWalkInfo
{
    uint32_t kernel, reserved;
    uint64_t nodeBytes, leafBoxBytes, triangleBytes;
} RtWalkInfo;
int rtgpu_get_walk_info(RtgpuContext* ctx, RtWalkInfo* out);

/* How a multi-device context (rtgpu_create_multi) exchanges the peers' tiles at read-back, and why -- so that a first run on a real multi-GPU node
 * explains itself (the reference has no counterpart: its only fan-out is Core/Utils/ThreadPool.cpp:57-117 under Viewport.cpp:244-262).
 *   gatherMode    0 one device (nothing to gather), 1 a kernel on the first device reads the peers' sum buffers in place (peer access over xGMI),
 *                 2 hipMemcpyPeerAsync into staging buffers of the first device, then the same kernel
 *   gatherReason  why mode 2: 0 not staged, 1 RTGPU_MULTI_STAGED=1, 2 hipDeviceCanAccessPeer said no for `reasonDevice`,
 *                 3 hipDeviceEnablePeerAccess failed for `reasonDevice` (reasonError = the HIP error code)
 *   devices       the HIP device index of every shard (an index may repeat); peerAccess[k]: the first device addresses device k's memory
 *   gathers / lastGatherMs / totalGatherMs   exchanges so far and their host-side wall time (submission to completion on the first device's stream) */
#define RTGPU_GATHER_NONE 0u
#define RTGPU_GATHER_PEER_KERNEL 1u
#define RTGPU_GATHER_STAGED_COPY 2u
typedef struct RtMultiInfo
{
    uint32_t numDevices, gatherMode, gatherReason;
    int32_t  reasonDevice, reasonError;
    int32_t  devices[16];
    uint32_t peerAccess[16];
    uint32_t reserved;
    uint64_t gathers;
    double   lastGatherMs, totalGatherMs;
} RtMultiInfo;
int rtgpu_get_multi_info(RtgpuContext* ctx, RtMultiInfo* out);

#ifdef __cplusplus
}
#endif

#endif /* RTGPU_H */
