// Viewport: progressive accumulation driver (reference: Core/Rendering/Viewport.h:37-58).
// Render() performs the pass prologue of Viewport::Render (Core/Rendering/Viewport.cpp:200-242) on the
// host -- Halton seeds, anti-aliasing offset -- then hands ONE pass to the renderer instead of fanning
// 32x32 tiles out to a thread pool.  GetFrontBuffer() runs Viewport::PostProcessTile (Viewport.cpp:495-550) on the device for the
// whole image (bloom included: the five blurred copies are rebuilt from the sum buffer on demand).  Adaptive rendering keeps the reference's block list logic (Viewport.cpp:552-700) on the host; the
// error estimates come from the device.
#pragma once

#include "Renderer.h"
#include "../Sampling/HaltonSampler.h"

namespace rt {

struct RenderingProgress
{
    uint32 passesFinished = 0;
    uint32 activePixels = 0;
    uint32 activeBlocks = 0;
    float converged = 0.0f;
    float averageError = std::numeric_limits<float>::infinity();
};

class RAYLIB_API Viewport
{
public:
    Viewport();
    ~Viewport();

    bool Resize(uint32 width, uint32 height);
    bool SetRenderingParams(const RenderingParams& params);
    const RenderingParams& GetRenderingParams() const { return mParams; }
    bool SetRenderer(const RendererPtr& renderer);
    bool Render(const Camera& camera);
    void Reset();

    // Synchronises with the device and returns the accumulated (not tone-mapped) image.
    const Bitmap& GetSumBuffer();
    const Bitmap& GetSecondarySumBuffer();
    // tone-mapped B8G8R8A8 image of the passes finished so far (reference: Viewport::GetFrontBuffer, Viewport.h:47)
    const Bitmap& GetFrontBuffer();
    bool SetPostprocessParams(const PostprocessParams& params);
    const PostprocessParams& GetPostprocessParams() const { return mPostprocessParams; }
    uint32 GetWidth() const { return mWidth; }
    uint32 GetHeight() const { return mHeight; }
    // averageError: with adaptive rendering off the reference recomputes it after every second pass; here it is evaluated when
    // asked for, at an even pass count (otherwise the value of the last evaluation is returned)
    const RenderingProgress& GetProgress();
    const std::vector<RtBlock>& GetBlocks() const { return mBlocks; }
    // known-answer hook: the list walk of UpdateBlocksList with the pass counter and the block errors supplied by the caller
    void UpdateBlocksListWithErrors(uint32 passesFinished, std::vector<float> errors);
    uint32 GetPassesFinished() const { return mProgress.passesFinished; }
    // counters of the LAST pass (synchronises), like the reference
    const RayTracingCounters& GetCounters();
    // totals since Reset (synchronises)
    RayTracingCounters GetTotalCounters();

    // Extension (the reference has no seed API): re-seed the Viewport's generators.
    void SetSeed(uint64 seed);
    void ClearAccumulation();   // sums, progress, counters, block list start over; the sample sequence goes on

    // The per-pass constants Render() would use next; advances the Halton sequence and the generator
    // exactly like Render().  Exposed so parity tests can feed identical constants to the CPU oracle.
    bool NextPassParams(const Camera& camera, RtPassParams& outParams);
    void FinishPass();

private:
    RendererPtr mRenderer;
    math::Random mRandomGenerator;
    HaltonSequence mHaltonSequence;
    RenderingParams mParams;
    RenderingProgress mProgress;
    void BuildInitialBlocksList();
    bool UpdateBlocksList();
    void ApplyBlockErrors(std::vector<float>& errors);
    std::vector<RtBlock> mBlocks;
    uint32 mErrorEvaluatedAtPass = 0;
    Bitmap mSum, mSecondarySum, mFrontBuffer;
    PostprocessParams mPostprocessParams;
    uint32 mWidth = 0, mHeight = 0;
    bool mSumDirty = false, mSecondarySumDirty = false;   // the device holds passes the host bitmaps have not seen
    bool mSumPinned = false;
    void PinSumBuffers(bool pin);   // page-locks the two sum bitmaps for the read-back (IRenderer::PinHostBuffer)
    std::vector<uint32> mSeedStorage;
    uint64 mRngKey[2];
    RayTracingCounters mCounters, mTotalsAtLastPass, mTotalsBeforeLastPass;
};

} // namespace rt
