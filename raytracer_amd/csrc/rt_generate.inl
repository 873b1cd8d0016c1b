// rt_generate.inl -- camera rays of the slot-per-pixel pipeline.  Included by rt_trace.hip.
// Viewport::RenderTile per-pixel prologue + Camera::GenerateRay (Viewport.cpp:305-331, Camera.cpp:81-118)
__global__ void __launch_bounds__(RT_BLOCK) k_generate(const RtSceneDesc scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass,
                                                       const Paths paths, const uint32_t* __restrict__ slotPixel, uint32_t numSlots,
                                                       uint32_t* __restrict__ queue, uint32_t* __restrict__ queueCount,
                                                       unsigned long long* counters)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < numSlots; slot += stride)
    {
        // several passes ride in one batch: slot = passInBatch * slotsPerPass + pixelSlot
        const uint32_t passInBatch = slot / slotsPerPass;
        const DevPass& pass = passes[passInBatch];
        const uint32_t pix = slotPixel[slot - passInBatch * slotsPerPass];
        const uint32_t x = pix & 0xFFFFu, y = pix >> 16;
        const uint32_t realY = pass.height - 1u - y;
        // invSize = VECTOR_ONE2 / FromIntegers(w, h, 1, 1); coords = (FromIntegers(x, realY) + sampleOffset) * invSize
        const float invW = 1.0f / (float)(int32_t)pass.width, invH = 1.0f / (float)(int32_t)pass.height;
        const V4 coords(((float)(int32_t)x + pass.sampleOffset[0]) * invW, ((float)(int32_t)realY + pass.sampleOffset[1]) * invH, 0.0f, 0.0f);

        Sampler sampler;
        sampler.seed = pass.seed; sampler.numDims = pass.numDimensions; sampler.blueNoiseLayers = pass.blueNoiseLayers; sampler.blueNoise = scene.blueNoise;
        sampler.resetPixel(x, y, pass.rngKey[0], pass.rngKey[1]);

        // Camera::GenerateRay up to (not including) the Ray constructor, which trace/shade re-run from origin+direction
        V4 origin, direction;
        cameraGenerateRayParts(pass.camera, coords, sampler, origin, direction);

        prec(paths, R_ORIGIN, slot) = f4(origin.x, origin.y, origin.z, fbits(0x100u));   // depth 0, lastSpecular = true (PathTracerMIS.h:29-34)
        prec(paths, R_DIR, slot) = f4(direction.x, direction.y, direction.z, 1.0f);        // lastPdfW = 1
        prec(paths, R_TP, slot) = f4(1.0f, 1.0f, 1.0f, 1.0f);
        prec(paths, R_RESULT, slot) = f4(0.0f, 0.0f, 0.0f, fbits(pix));
        storeSampler(sampler, paths, slot, 0.0f, 0u);
        queue[slot] = slot;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
    {
        *queueCount = numSlots;
        atomicAdd(&counters[C_PRIMARY], (unsigned long long)numSlots);
    }
}
