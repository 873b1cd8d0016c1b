// rt_post.inl -- kernels behind rtgpu_postprocess / rtgpu_compute_block_errors / rtgpu_evaluate_textures.  Included by rt_trace.hip.
// Viewport::PostProcessTile (Viewport.cpp:495-550): sum buffer -> 0x00RRGGBB front buffer, one thread per pixel
__global__ void __launch_bounds__(RT_BLOCK) k_postprocess(const float* __restrict__ sum, uint32_t* __restrict__ front, uint32_t width, uint32_t height,
                                                          const RtPostprocessParams params, const PostScale colorScale)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= width * height) return;
    const uint32_t y = i / width, x = i - y * width;
    front[i] = postProcessPixel(sum[3 * (size_t)i + 0], sum[3 * (size_t)i + 1], sum[3 * (size_t)i + 2], x, y, params, colorScale.c);
}

// ---- bloom: Bitmap::GaussianBlur (Core/Utils/Bitmap.cpp:880-1020) ---------------------------------------------------------
// n box blurs per line, each a running sum in the reference's order (BoxBlur_Internal, :880-914), so a line is sequential;
// one thread per (line, colour channel).  The two line buffers live in global scratch, element-major (element e of
// thread t at [e * numThreads + t]) so that the threads of a wave touch consecutive words.
RT_DEV void boxBlurLine(float* __restrict__ dst, const float* __restrict__ src, uint32_t radius, uint32_t width, uint32_t stride)
{
    const float factor = 1.0f / (float)(2u * radius + 1u);
    uint32_t b = 0, e = 0, t = 0;
    const float firstValue = src[0], lastValue = src[(size_t)(width - 1u) * stride];
    float val = firstValue * (float)(radius + 1u);
    for (uint32_t j = 0; j < radius; j++) val = val + src[(size_t)(b++) * stride];
    for (uint32_t j = 0; j <= radius; j++) { val = val + (src[(size_t)(b++) * stride] - firstValue); dst[(size_t)(t++) * stride] = val * factor; }
    for (uint32_t j = radius + 1u; j < width - radius; j++) { val = val + (src[(size_t)(b++) * stride] - src[(size_t)(e++) * stride]); dst[(size_t)(t++) * stride] = val * factor; }
    for (uint32_t j = width - radius; j < width; j++) { val = val + (lastValue - src[(size_t)(e++) * stride]); dst[(size_t)(t++) * stride] = val * factor; }
}
__global__ void __launch_bounds__(RT_BLOCK) k_blur_lines(float* __restrict__ image, uint32_t width, uint32_t height, uint32_t vertical, const BlurPlan plan,
                                                         float* __restrict__ lineA, float* __restrict__ lineB)
{
    const uint32_t numLines = vertical ? width : height, length = vertical ? height : width;
    const uint32_t numThreads = numLines * 3u;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= numThreads) return;
    const uint32_t line = t / 3u, channel = t - line * 3u;
    const size_t pixelStride = vertical ? (size_t)width * 3u : 3u;
    float* px = image + (vertical ? (size_t)line * 3u : (size_t)line * width * 3u) + channel;
    // horizontal: source = B, target = A (:952-953); vertical: source = A, target = B (:983-984)
    float* source = (vertical ? lineA : lineB) + t;
    float* target = (vertical ? lineB : lineA) + t;
    for (uint32_t e = 0; e < length; ++e) source[(size_t)e * numThreads] = px[(size_t)e * pixelStride];
    for (uint32_t i = 0; i < plan.n; ++i)
    {
        const uint32_t radius = (float)i < plan.m ? plan.wl : plan.wu;
        boxBlurLine(target, source, radius, length, numThreads);
        float* tmp = source; source = target; target = tmp;
    }
    // horizontal reads targetLinePtr AFTER the last swap (:961-964: the buffer the last blur read from, i.e. n-1 blurs);
    // vertical reads tempLineA (:1003-1009: the last blur's output for even n)
    const float* result = vertical ? lineA + t : target;
    for (uint32_t e = 0; e < length; ++e) px[(size_t)e * pixelStride] = result[(size_t)e * numThreads];
}

// Viewport::PostProcessTile with bloom (:512-524): rgb * (1 - bloomFactor) + bloomFactor * sum of weighted blur levels
__global__ void __launch_bounds__(RT_BLOCK) k_postprocess_bloom(const float* __restrict__ sum, const BloomLevels blurred, uint32_t* __restrict__ front, uint32_t width, uint32_t height,
                                                                const RtPostprocessParams params, const PostScale colorScale)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= width * height) return;
    const uint32_t y = i / width, x = i - y * width;
    const float bloomWeights[5] = { 0.35f, 0.25f, 0.15f, 0.15f, 0.1f };
    float rgb[3];
    for (int k = 0; k < 3; ++k)
    {
        float v = sum[3 * (size_t)i + k] * (1.0f - params.bloomFactor);
        float bloomColor = 0.0f;
        for (int l = 0; l < 5; ++l) bloomColor = __fmaf_rn(blurred.level[l][3 * (size_t)i + k], bloomWeights[l], bloomColor);
        rgb[k] = __fmaf_rn(bloomColor, params.bloomFactor, v);
    }
    front[i] = postProcessPixel(rgb[0], rgb[1], rgb[2], x, y, params, colorScale.c);
}

// Viewport::ComputeBlockError (Viewport.cpp:552-581) in two steps that keep the reference's summation order: one thread
// per (block, row) adds the pixel errors of its row left to right, then one thread per block adds the rows top to bottom.
__global__ void __launch_bounds__(RT_BLOCK) k_block_error_rows(const float* __restrict__ sum, const float* __restrict__ secondary, uint32_t width,
                                                               const RtBlock* __restrict__ blocks, const ErrorRow* __restrict__ rows, uint32_t numRows,
                                                               float imageScalingFactor, float* __restrict__ rowErrors)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numRows) return;
    const RtBlock b = blocks[rows[i].block];
    const uint32_t y = rows[i].y;
    const float scaleB = 2.0f * imageScalingFactor;
    float rowError = 0.0f;
    for (uint32_t x = b.minX; x < b.maxX; ++x)
    {
        const size_t p = 3 * ((size_t)y * width + x);
        const float ax = imageScalingFactor * sum[p], ay = imageScalingFactor * sum[p + 1], az = imageScalingFactor * sum[p + 2];
        const float bx = scaleB * secondary[p], by = scaleB * secondary[p + 1], bz = scaleB * secondary[p + 2];
        const float dx = fabsf(ax - bx), dy = fabsf(ay - by), dz = fabsf(az - bz);
        const float error = (dx + 2.0f * dy + dz) / sqrtf(RTD_EPSILON + ax + 2.0f * ay + az);
        rowError += error;
    }
    rowErrors[i] = rowError;
}
__global__ void __launch_bounds__(RT_BLOCK) k_block_error_total(const RtBlock* __restrict__ blocks, const uint32_t* __restrict__ firstRow, uint32_t numBlocks,
                                                                const float* __restrict__ rowErrors, uint32_t totalArea, float* __restrict__ outErrors)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numBlocks) return;
    const RtBlock b = blocks[i];
    float totalError = 0.0f;
    for (uint32_t r = 0; r < b.maxY - b.minY; ++r) totalError += rowErrors[firstRow[i] + r];
    const uint32_t blockArea = (b.maxX - b.minX) * (b.maxY - b.minY);
    outErrors[i] = totalError * sqrtf((float)blockArea / (float)totalArea) / (float)blockArea;
}

// ITexture::Evaluate for a list of (texture, uv) pairs -- rtgpu_evaluate_textures
__global__ void __launch_bounds__(RT_BLOCK) k_evaluate_textures(const RtSceneDesc scene, uint32_t count, const uint32_t* __restrict__ textureIndex,
                                                                const float* __restrict__ uv, float* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const V4 c = textureEvaluate(scene, textureIndex[i], V4(uv[2 * i], uv[2 * i + 1], 0.0f, 0.0f));
    out[4 * i + 0] = c.x; out[4 * i + 1] = c.y; out[4 * i + 2] = c.z; out[4 * i + 3] = c.w;
}
