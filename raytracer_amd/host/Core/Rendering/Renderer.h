// Renderer interface and factory (reference: Core/Rendering/Renderer.h:18-69, Renderer.cpp:45-69).
//
// The reference calls IRenderer::RenderPixel once per pixel from Viewport::RenderTile; a per-pixel
// virtual cannot be a device boundary, so the batch entry the reference leaves as a TODO
// (Renderer.h:33 "TODO batch & multisample rendering") is the one implemented here: RenderPass().
#pragma once

#include <stddef.h>

#include <vector>

#include "Context.h"
#include "../Scene/Scene.h"
#include "../Scene/Camera.h"
#include "../Utils/Bitmap.h"
#include "PostProcess.h"

namespace rt {

class RAYLIB_API IRenderer
{
public:
    explicit IRenderer(const Scene& scene) : mScene(scene) {}
    virtual ~IRenderer();
    virtual const char* GetName() const = 0;

    // device-pass interface used by Viewport
    virtual bool Resize(uint32 width, uint32 height) = 0;
    virtual bool Reset() = 0;
    virtual bool RenderPass(const RtPassParams& params) = 0;                 // asynchronous
    virtual bool ReadSum(float* sumRGB, float* secondaryRGB) = 0;            // synchronises
    virtual bool PinHostBuffer(void* /*data*/, size_t /*bytes*/) { return false; }   // page-locks a buffer ReadSum is given repeatedly
    virtual void UnpinHostBuffer(void* /*data*/) {}
    virtual bool GetCounters(RayTracingCounters& outTotals) = 0;             // totals since Reset; synchronises
    // Viewport::PostProcessTile over the whole sum buffer -> 0x00RRGGBB pixels; synchronises
    virtual bool PostProcess(const PostprocessParams& params, uint32 numPasses, uint32* outBGRA) = 0;
    // adaptive rendering: Viewport::ComputeBlockError for a list of blocks (synchronises); restrict the next passes to blocks (empty = all)
    virtual bool ComputeBlockErrors(uint32 numPasses, const std::vector<RtBlock>& blocks, std::vector<float>& outErrors) = 0;
    virtual bool SetActiveBlocks(const std::vector<RtBlock>& blocks) = 0;

protected:
    const Scene& mScene;
private:
    IRenderer(const IRenderer&) = delete;
    IRenderer& operator=(const IRenderer&) = delete;
};
using RendererPtr = std::shared_ptr<IRenderer>;

// The reference's factory (Core/Rendering/Renderer.cpp:20-46): "Path Tracer MIS" (the hot path), "Path Tracer", "Light Tracer",
// "Debug" and "VCM", all on the device; an unknown name logs and returns nullptr.
RAYLIB_API RendererPtr CreateRenderer(const std::string& name, const Scene& scene);

// Device selection for renderers created afterwards (default 0, or LOCAL_RANK when set).
RAYLIB_API void SetRendererDevice(int deviceIndex);
RAYLIB_API int GetRendererDevice();   // -1: the default
// Several devices of the node for ONE renderer (rtgpu_create_multi: the frame's 64x64 tiles are dealt to them, the read-back calls gather);
// an empty list goes back to one device.  The environment variable RTGPU_DEVICES ("0,1,2,3", or "all") does the same without a code change.
RAYLIB_API void SetRendererDevices(const std::vector<int>& deviceIndices);

} // namespace rt
