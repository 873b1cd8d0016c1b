// Run-time rendering parameters (reference: Core/Rendering/Context.h:25-90).
#pragma once

#include "Counters.h"

namespace rt {

enum class TraversalMode : uint8 { Single = 0, Packet };
enum class LightSamplingStrategy : uint8 { Single, All };

struct AdaptiveRenderingSettings
{
    bool enable = false;
    uint32 numInitialPasses = 10;
    uint32 minBlockSize = 4;
    uint32 maxBlockSize = 256;
    float subdivisionTreshold = 0.005f;
    float convergenceTreshold = 0.0001f;
};

struct SamplingParams
{
    uint32 dimensions = 64;
    bool useBlueNoiseDithering = true;
};

struct RenderingParams
{
    uint32 numThreads = 0;               // meaningless for the device renderer; kept for API compatibility
    SamplingParams samplingParams;
    float antiAliasingSpread = 0.5f;
    float motionBlurStrength = 0.5f;     // motion blur is a TODO in the reference (SceneObject.cpp:26-44): no effect
    uint32 maxRayDepth = 20;
    uint32 minRussianRouletteDepth = 1;
    uint16 tileSize = 32;                // CPU tiling; the device shards by 64x64 tiles across GPUs only
    TraversalMode traversalMode = TraversalMode::Single;
    LightSamplingStrategy lightSamplingStrategy = LightSamplingStrategy::Single;
    bool visualizeTimePerPixel = false;
    AdaptiveRenderingSettings adaptiveSettings;
};

} // namespace rt
