// Unidirectional path tracer with NEE + MIS, running on the GPU through include/rtgpu.h
// (reference: Core/Rendering/PathTracerMIS.h).
#pragma once

#include "Renderer.h"

namespace rt {

class RAYLIB_API PathTracerMIS : public IRenderer
{
public:
    explicit PathTracerMIS(const Scene& scene);
    ~PathTracerMIS() override;
    const char* GetName() const override;

    bool Resize(uint32 width, uint32 height) override;
    bool Reset() override;
    bool RenderPass(const RtPassParams& params) override;
    bool ReadSum(float* sumRGB, float* secondaryRGB) override;
    bool PinHostBuffer(void* data, size_t bytes) override;
    void UnpinHostBuffer(void* data) override;
    bool GetCounters(RayTracingCounters& outTotals) override;
    bool PostProcess(const PostprocessParams& params, uint32 numPasses, uint32* outBGRA) override;
    bool ComputeBlockErrors(uint32 numPasses, const std::vector<RtBlock>& blocks, std::vector<float>& outErrors) override;
    bool SetActiveBlocks(const std::vector<RtBlock>& blocks) override;

    // for debugging (the reference's UI pokes these: Demo/Demo_UserInterface.cpp:467-469)
    math::Vector4 mLightSamplingWeight;
    math::Vector4 mBSDFSamplingWeight;

    RtgpuContext* GetDeviceContext() const { return mCtx; }
    bool SetShard(uint32 rank, uint32 worldSize);

protected:
    // wholeFrameOnOneDevice: integrators that splat over the frame (Light Tracer, VCM) ignore SetRendererDevices
    PathTracerMIS(const Scene& scene, bool wholeFrameOnOneDevice);

private:
    bool EnsureSceneUploaded();
    RtgpuContext* mCtx = nullptr;
    uint64 mUploadedBuildId = ~0ull;
    std::vector<uint16> mBlueNoise;
};

// "Path Tracer": the same device pipeline without next event estimation and MIS (reference: Core/Rendering/PathTracer.h)
class RAYLIB_API PathTracer : public PathTracerMIS
{
public:
    explicit PathTracer(const Scene& scene);
    const char* GetName() const override;
};

// "Debug": one colour per pixel from the primary hit (reference: Core/Rendering/DebugRenderer.h)
enum class DebugRenderingMode : uint8
{
    CameraLight = 0, TriangleID, Depth, Position, Normals, Tangents, Bitangents, TexCoords, BaseColor, Emission, Roughness, Metalness, IoR,
};
class RAYLIB_API DebugRenderer : public PathTracerMIS
{
public:
    explicit DebugRenderer(const Scene& scene);
    const char* GetName() const override;
    bool RenderPass(const RtPassParams& params) override;
    DebugRenderingMode mRenderingMode;
private:
    int mAppliedMode = -1;
};

// "Light Tracer": light paths connected to the camera (reference: Core/Rendering/LightTracer.h)
class RAYLIB_API LightTracer : public PathTracerMIS
{
public:
    explicit LightTracer(const Scene& scene);
    const char* GetName() const override;
};

} // namespace rt
