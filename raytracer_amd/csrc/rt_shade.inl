// rt_shade.inl -- PathTracerMIS / PathTracer / Debug shading over slot-per-pixel path state, and Film::AccumulateColor (k_accumulate).
// Included by rt_shade.hip.
RT_DEV float CombineMis(float samplePdf, float otherPdf) { return FastDivide(samplePdf, samplePdf + otherPdf); }        // PathTracerMIS.cpp:16-24
RT_DEV float PdfAtoW(float pdfA, float distance, float cosThere) { return FastDivide(pdfA * Sqr(distance), Abs(cosThere)); }   // :26-29

// PathTracerMIS::SampleLight up to the shadow ray (PathTracerMIS.cpp:43-79, 97-119): produces the NEE
// request {direction, tmax, contribution}; the occlusion test and the accumulation happen in k_trace_shadow.
template <int kLean>
__device__ __forceinline__ static bool computeLightSample(const RtSceneDesc& scene, const DevPass& pass, Sampler& sampler, const RtLight& light,
                               const ShadingData& sd, const RtMaterial& mat, uint32_t depth, float lightPickProbability,
                               float4& outDirTmax, float4& outContribution)
{
    float u[3]; u[0] = sampler.getFloat(); u[1] = sampler.getFloat(); u[2] = sampler.getFloat();
    float tmax = -1.0f; V4 dir = zero4(); V4 contribution = zero4();
    IlluminateResult ir;
    const V4 radiance = lightIlluminate<kLean>(scene, light, sd.intersection, u, ir);
    if (!almostZero4(radiance))
    {
        float bsdfPdfW = 0.0f;
        const V4 factor = materialEvaluate<kLean>(mat, sd, neg(ir.directionToLight), bsdfPdfW);
        if (!almostZero4(factor))
        {
            float weight = 1.0f;
            const bool isLastPathSegment = depth >= pass.maxRayDepth;
            if (!(light.flags & RT_LIGHT_FLAG_DELTA) && !isLastPathSegment)
            {
                const float continuationProbability = 1.0f;
                bsdfPdfW *= continuationProbability;
                weight = CombineMis(ir.directPdfW * lightPickProbability, bsdfPdfW);
            }
            contribution = (radiance * factor) * FastDivide(weight, lightPickProbability * ir.directPdfW);
            dir = ir.directionToLight;
            tmax = ir.distance * 0.999f;
        }
    }
    outDirTmax = f4(dir.x, dir.y, dir.z, tmax);
    outContribution = f4(contribution.x, contribution.y, contribution.z, 0.0f);
    return tmax >= 0.0f;   // a shadow ray has to be traced for this request
}
template <int kLean>
__device__ __forceinline__ static bool prepareLightSample(const RtSceneDesc& scene, const DevPass& pass, Sampler& sampler, const RtLight& light,
                               const ShadingData& sd, const RtMaterial& mat, uint32_t depth, float lightPickProbability,
                               const Paths& paths, uint32_t slot, uint32_t requestIndex)
{
    float4 dirTmax, contribution;
    const bool ray = computeLightSample<kLean>(scene, pass, sampler, light, sd, mat, depth, lightPickProbability, dirTmax, contribution);
    pshadow(paths, requestIndex, 0, slot) = dirTmax;
    pshadow(paths, requestIndex, 1, slot) = contribution;
    return ray;
}

// Folds the finished NEE requests of the path's previous vertex into its radiance:
// accumulatedColor = sum of the unoccluded SampleLight() results in light order, times mLightSamplingWeight,
// then resultColor.MulAndAccumulate(throughput, ...) (PathTracerMIS.cpp:141-151, 320).  k_trace_shadow marks
// occluded requests with tmax < 0.
RT_DEV void resolvePendingLightSamples(const Paths& paths, uint32_t slot, uint32_t numRequests, V4 lightSamplingWeight, V4& resultColor, Counters& cnt)
{
    if (numRequests == 0) return;
    V4 accumulated = zero4();
    bool any = false;
    for (uint32_t l = 0; l < numRequests; ++l)
    {
        if (pshadow(paths, l, 0, slot).w < 0.0f) continue;   // no shadow ray was needed, or k_trace found an occluder
        const float4 c = pshadow(paths, l, 1, slot);
        accumulated = accumulated + V4(c.x, c.y, c.z, 0.0f);
        any = true;
        cnt.c[C_SHADOW_HIT]++;   // counters.numShadowRaysHit: the shadow ray reached the light (PathTracerMIS.cpp:96-99)
    }
    if (!any) return;
    accumulated = accumulated * lightSamplingWeight;
    const float4 tp = prec(paths, R_SH_TP, slot);
    resultColor = mulAdd(V4(tp.x, tp.y, tp.z, 0.0f), accumulated, resultColor);
}

#define RT_APPEND_BUFFER 2048u

// Publishes a block's LDS append buffer with ONE global atomic and coalesced stores.  Called by all threads of
// the block at a block-uniform point (after a __syncthreads()).
RT_DEV void flushAppendBuffer(const uint32_t* buf, uint32_t& count, uint32_t& base, uint32_t* __restrict__ queue, uint32_t* __restrict__ queueCount)
{
    const uint32_t n = count;
    if (n != 0)
    {
        if (threadIdx.x == 0) base = atomicAdd(queueCount, n);
        __syncthreads();
        const uint32_t b = base;
        for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) queue[b + k] = buf[k];
        __syncthreads();
        if (threadIdx.x == 0) count = 0;
    }
    __syncthreads();
}

#ifndef RT_SHADE_FUNCTIONS_ONLY   // (rt_tail.hip takes the functions above and none of the kernels below)
// The body of PathTracerMIS::RenderPixel's loop for one path vertex (PathTracerMIS.cpp:276-395).
// kPlain: the renderer "Path Tracer" instead (PathTracer::RenderPixel, Core/Rendering/PathTracer.cpp:73-171): the same walk without
// next event estimation, MIS weights and sampling weights.
template <bool kLean, bool kPlain = false>
__global__ void __launch_bounds__(RT_BLOCK) k_shade(const RtSceneDesc scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass, const Paths paths,
                                                    const uint32_t* __restrict__ queueIn, const uint32_t* __restrict__ countIn,
                                                    uint32_t* __restrict__ queueOut, uint32_t* __restrict__ countOut,
                                                    uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount,
                                                    unsigned long long* counters)
{
    __shared__ uint32_t sPathBuf[RT_APPEND_BUFFER], sShadowBuf[RT_APPEND_BUFFER];
    __shared__ uint32_t sPathCount, sShadowCount, sPathBase, sShadowBase;
    if (threadIdx.x == 0) { sPathCount = 0; sShadowCount = 0; }
    __syncthreads();
    Counters cnt; zeroCounters(cnt);
    const uint32_t count = *countIn;
    const uint32_t stride = gridDim.x * blockDim.x;
    // the structural parameters are identical for all passes of a batch (the host flushes when they change);
    // seeds, camera, anti-aliasing offset and rng keys are per pass
    const DevPass pass = passes[0];
    const V4 lightSamplingWeight = load4(pass.lightSamplingWeight), bsdfSamplingWeight = load4(pass.bsdfSamplingWeight);
    // GetLightPickingProbability, PathTracerMIS.cpp:157-172
    const float lightPickProbability = pass.lightSamplingStrategy == RT_LIGHT_SAMPLING_SINGLE ? 1.0f / (float)scene.numLights : 1.0f;
    const uint32_t maxRequestsPerVertex = pass.lightSamplingStrategy == RT_LIGHT_SAMPLING_SINGLE ? 1u : (scene.numLights < 8u ? scene.numLights : 8u);

    // every lane of a wave runs the same number of iterations so that the ballot below sees whole waves
    const uint32_t rounded = (count + RT_BLOCK - 1) / RT_BLOCK * RT_BLOCK;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < rounded; i += stride)
    {
        bool alive = false;
        uint32_t slot = 0;
        unsigned long long rayMask = 0ull;   // NEE requests of this vertex that need a shadow ray (bit = request index)
        if (i < count)
        {
            slot = queueIn[i];
            const float4 rOrigin = prec(paths, R_ORIGIN, slot), rDir = prec(paths, R_DIR, slot), rTp = prec(paths, R_TP, slot);
            const float4 rResult = prec(paths, R_RESULT, slot), rHit = prec(paths, R_HIT, slot), rSampler = prec(paths, R_SAMPLER, slot);
            const uint32_t flags = ubits(rOrigin.w);
            uint32_t depth = flags & 0xFFu;
            const bool lastSpecular = (flags & 0x100u) != 0;
            const float lastPdfW = rDir.w;
            const uint32_t pix = ubits(rResult.w);
            const Ray ray = makePathRay(rOrigin, rDir, depth);
            V4 throughput(rTp.x, rTp.y, rTp.z, rTp.w);
            V4 resultColor(rResult.x, rResult.y, rResult.z, 0.0f);
            resolvePendingLightSamples(paths, slot, ubits(rSampler.w), lightSamplingWeight, resultColor, cnt);   // NEE of the previous vertex
            Hit hit;
            hit.objectId = ubits(rHit.x); hit.subObjectId = ubits(rHit.y); hit.distance = rHit.z; hit.u = rHit.w; hit.v = rSampler.x;
            bool samplerStored = false;

            do
            {
                if (hit.objectId == RT_INVALID_OBJECT)
                {
                    // EvaluateGlobalLights, PathTracerMIS.cpp:214-252
                    V4 result = zero4();
                    for (uint32_t g = 0; g < scene.numGlobalLights; ++g)
                    {
                        const RtLight& light = scene.lights[scene.globalLights[g]];
                        const Ray lightSpaceRay = transformRayUnsafe(loadM4(light.invTransform), ray);
                        float directPdfW = 0.0f;
                        const V4 lightContribution = lightGetRadiance<kLean>(scene, light, lightSpaceRay, zero4(), 1.0f, directPdfW);
                        if (kPlain) result = result + lightContribution;   // PathTracer::EvaluateGlobalLights, PathTracer.cpp:47-71
                        else if (!almostZero4(lightContribution))
                        {
                            float misWeight = 1.0f;
                            if (depth > 0 && !lastSpecular) misWeight = CombineMis(lastPdfW, directPdfW * lightPickProbability);
                            result = mulAdd(lightContribution, misWeight, result);
                        }
                    }
                    if (!kPlain) result = result * bsdfSamplingWeight;
                    resultColor = mulAdd(throughput, result, resultColor);
                    break;
                }

                ShadingData sd;
                // The reference keeps ONE ShadingData for the whole path (PathTracerMIS.cpp:258) and LightSceneObject::
                // EvaluateIntersection does not touch `material` (SceneObject_Light.cpp:62-73): when a path hits an area light,
                // IntersectionData::material is still the PREVIOUS vertex's, and its normal map (if any) is applied to the
                // light's frame (Scene.cpp:327).  The previous material rides in the flags word: (index + 1) << 9.
                sd.intersection.material = (flags >> 9) - 1u;   // 0 -> RT_NO_MATERIAL
                if (hit.distance < FLT_MAX) sceneEvaluateIntersection<kLean>(scene, ray, hit, sd.intersection, cnt);

                if (!kLean && hit.subObjectId == RT_LIGHT_OBJECT)
                {
                    // EvaluateLight, PathTracerMIS.cpp:174-212
                    const RtObject& obj = scene.objects[hit.objectId];
                    const RtLight& light = scene.lights[obj.lightIndex];
                    const M4 worldToLight = loadM4(obj.invTransform);
                    const Ray lightSpaceRay = transformRayUnsafe(worldToLight, ray);
                    const V4 lightSpaceHitPoint = transformPoint(worldToLight, sd.intersection.frame.r[3]);
                    const float cosAtLight = -dot3(sd.intersection.frame.r[2], ray.dir);
                    float directPdfA = 0.0f;
                    V4 lightContribution = lightGetRadiance<false>(scene, light, lightSpaceRay, lightSpaceHitPoint, cosAtLight, directPdfA);
                    if (kPlain) resultColor = mulAdd(throughput, lightContribution, resultColor);   // PathTracer::EvaluateLight, PathTracer.cpp:26-45
                    else if (!almostZero4(lightContribution))
                    {
                        float misWeight = 1.0f;
                        if (depth > 0 && !lastSpecular)
                        {
                            const float directPdfW = PdfAtoW(directPdfA, hit.distance, cosAtLight);
                            misWeight = CombineMis(lastPdfW, directPdfW * lightPickProbability);
                        }
                        lightContribution = lightContribution * bsdfSamplingWeight;
                        resultColor = mulAdd(throughput, lightContribution * misWeight, resultColor);
                    }
                    else
                    {
                        resultColor = mulAdd(throughput, zero4(), resultColor);
                    }
                    break;
                }

                sd.outgoingDirWorldSpace = neg(ray.dir);
                const RtMaterial& mat = scene.materials[sd.intersection.material];
                materialEvaluateShadingData<kLean>(scene, mat, sd);

                // emission, PathTracerMIS.cpp:309-317
                resultColor = mulAdd(throughput, kPlain ? sd.mp.emission : sd.mp.emission * bsdfSamplingWeight, resultColor);

                Sampler sampler; loadSampler(sampler, paths, slot, pix, rSampler, pass, scene.blueNoise);
                sampler.seed = passes[slot / slotsPerPass].seed;

                // SampleLights (next event estimation), PathTracerMIS.cpp:125-155
                uint32_t numRequests = 0;
                if (!kPlain && scene.numLights != 0)
                {
                    if (pass.lightSamplingStrategy == RT_LIGHT_SAMPLING_SINGLE)
                    {
                        uint32_t lightIndex = 0;
                        if (scene.numLights > 1) lightIndex = sampler.fallbackInt() % scene.numLights;
                        if (prepareLightSample<kLean>(scene, pass, sampler, scene.lights[lightIndex], sd, mat, depth, lightPickProbability, paths, slot, 0)) rayMask = 1ull;
                        numRequests = 1;
                    }
                    else
                    {
                        for (uint32_t l = 0; l < scene.numLights; ++l)
                        {
                            const bool ray = prepareLightSample<kLean>(scene, pass, sampler, scene.lights[l], sd, mat, depth, lightPickProbability, paths, slot, l);
                            if (ray)
                            {
                                if (l < 8u) rayMask |= 1ull << l;
                                else shadowQueue[atomicAdd(shadowCount, 1u)] = l * paths.capacity + slot;   // more than 64 lights: per-lane append
                            }
                        }
                        numRequests = scene.numLights;
                    }
                    prec(paths, R_SH_P, slot) = f4(sd.intersection.frame.r[3].x, sd.intersection.frame.r[3].y, sd.intersection.frame.r[3].z, 0.0f);
                    prec(paths, R_SH_TP, slot) = f4(throughput.x, throughput.y, throughput.z, 0.0f);
                }

                bool cont = true;
                if (depth >= pass.maxRayDepth) cont = false;

                // Russian roulette, PathTracerMIS.cpp:330-347
                if (cont && depth >= pass.minRussianRouletteDepth)
                {
                    const float minColorValue = 0.125f;
                    const float threshold = minColorValue + (1.0f - minColorValue) * colorMax(sd.mp.baseColor);
                    if (sampler.getFloat() > threshold) cont = false;
                    else throughput = throughput * (1.0f / threshold);
                }

                // BSDF sampling, PathTracerMIS.cpp:349-395
                if (cont)
                {
                    float pdf = 0.0f; V4 incomingDirWorldSpace = zero4(); uint32_t event = EV_NULL;
                    float u[3]; u[0] = sampler.getFloat(); u[1] = sampler.getFloat(); u[2] = sampler.getFloat();
                    const V4 bsdfValue = materialSample<kLean>(mat, sd, u, incomingDirWorldSpace, pdf, event);
                    if (event == EV_NULL) cont = false;
                    else
                    {
                        throughput = throughput * bsdfValue;
                        if (almostZero4(throughput)) cont = false;
                        else
                        {
                            prec(paths, R_ORIGIN, slot) = f4(sd.intersection.frame.r[3].x, sd.intersection.frame.r[3].y, sd.intersection.frame.r[3].z,
                                                             fbits((depth + 1u) | (((event & EV_SPECULAR) != 0) ? 0x100u : 0u) | ((sd.intersection.material + 1u) << 9)));
                            prec(paths, R_DIR, slot) = f4(incomingDirWorldSpace.x, incomingDirWorldSpace.y, incomingDirWorldSpace.z, pdf);
                            prec(paths, R_TP, slot) = f4(throughput.x, throughput.y, throughput.z, throughput.w);
                            alive = true;
                        }
                    }
                }
                storeSampler(sampler, paths, slot, hit.v, numRequests);
                samplerStored = true;
            } while (false);

            if (!samplerStored && ubits(rSampler.w) != 0u) prec(paths, R_SAMPLER, slot).w = fbits(0u);   // the resolved requests are spent
            prec(paths, R_RESULT, slot) = f4(resultColor.x, resultColor.y, resultColor.z, rResult.w);
            if (!alive) cnt.c[C_RAYS] += depth + 1u;   // counters.numRays += depth + 1, PathTracerMIS.cpp:412
        }

        // Queue appends go through per-block LDS buffers: a returning atomic on ONE global word sustains only ~88
        // operations per microsecond on this chip, so per-wave appends (hundreds of thousands per launch) would
        // dominate the kernel; a block publishes ~RT_APPEND_BUFFER entries per global atomic instead.
        for (unsigned long long pending = rayMask; pending != 0ull; pending &= pending - 1ull)
        {
            const uint32_t l = (uint32_t)(__ffsll((long long)pending) - 1);
            sShadowBuf[atomicAdd(&sShadowCount, 1u)] = l * paths.capacity + slot;
        }
        if (alive) sPathBuf[atomicAdd(&sPathCount, 1u)] = slot;
        __syncthreads();
        // flush when the next iteration could overflow a buffer (wave-uniform decision on block-shared counters)
        const bool last = (i - threadIdx.x) + stride >= rounded;
        if (last || sPathCount + RT_BLOCK > RT_APPEND_BUFFER) flushAppendBuffer(sPathBuf, sPathCount, sPathBase, queueOut, countOut);
        if (last || sShadowCount + RT_BLOCK * maxRequestsPerVertex > RT_APPEND_BUFFER) flushAppendBuffer(sShadowBuf, sShadowCount, sShadowBase, shadowQueue, shadowCount);
    }
    flushCounters(cnt, counters);
}

__global__ void __launch_bounds__(RT_BLOCK) k_debug_shade(const RtSceneDesc scene, const Paths paths, const uint32_t* __restrict__ queueIn, const uint32_t* __restrict__ countIn,
                                                          uint32_t mode, unsigned long long* counters)
{
    Counters cnt; zeroCounters(cnt);
    const uint32_t count = *countIn;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride)
    {
        const uint32_t slot = queueIn[i];
        const float4 rOrigin = prec(paths, R_ORIGIN, slot), rDir = prec(paths, R_DIR, slot), rHit = prec(paths, R_HIT, slot);
        const Ray ray = makePathRay(rOrigin, rDir, 0u);
        Hit hit; hit.objectId = ubits(rHit.x); hit.subObjectId = ubits(rHit.y); hit.distance = rHit.z; hit.u = rHit.w; hit.v = prec(paths, R_SAMPLER, slot).x;
        V4 color = zero4();
        if (hit.objectId != RT_INVALID_OBJECT)
        {
            if (hit.subObjectId == RT_LIGHT_OBJECT) color = V4(1.0f, 1.0f, 0.0f, 0.0f);
            else
            {
                ShadingData sd; sd.intersection.material = RT_NO_MATERIAL;
                if (mode != DBG_TRIANGLE_ID && mode != DBG_DEPTH)
                {
                    if (hit.distance < FLT_MAX) sceneEvaluateIntersection<false>(scene, ray, hit, sd.intersection, cnt);
                    materialEvaluateShadingData<false>(scene, scene.materials[sd.intersection.material], sd);
                }
                switch (mode)
                {
                case DBG_CAMERA_LIGHT: { const float NdotL = dot3(ray.dir, sd.intersection.frame.r[2]); color = sd.mp.baseColor * Abs(NdotL); break; }
                case DBG_DEPTH: { const float invDepth = 1.0f - 1.0f / (1.0f + hit.distance / 10.0f); color = splat(invDepth); break; }
                case DBG_TRIANGLE_ID:
                {
                    color = debugTriangleIdColor(hit.objectId, hit.subObjectId);
                    break;
                }
                case DBG_TANGENTS: color = min4(splat(1.0f), max4(zero4(), mulAdd(sd.intersection.frame.r[0], splat(0.5f), splat(0.5f)))); break;
                case DBG_BITANGENTS: color = min4(splat(1.0f), max4(zero4(), mulAdd(sd.intersection.frame.r[1], splat(0.5f), splat(0.5f)))); break;
                case DBG_NORMALS: color = min4(splat(1.0f), max4(zero4(), mulAdd(sd.intersection.frame.r[2], splat(0.5f), splat(0.5f)))); break;
                case DBG_POSITION: color = max4(zero4(), sd.intersection.frame.r[3]); break;
                case DBG_TEXCOORDS: color = V4(sd.intersection.texCoord.x - floorf(sd.intersection.texCoord.x), sd.intersection.texCoord.y - floorf(sd.intersection.texCoord.y), 0.0f, 0.0f); break;
                case DBG_BASE_COLOR: color = sd.mp.baseColor; break;
                case DBG_EMISSION: color = sd.mp.emission; break;
                case DBG_ROUGHNESS: color = splat(sd.mp.roughness); break;
                case DBG_METALNESS: color = splat(sd.mp.metalness); break;
                default: color = splat(sd.mp.IoR); break;
                }
            }
        }
        prec(paths, R_RESULT, slot) = f4(color.x, color.y, color.z, prec(paths, R_RESULT, slot).w);
    }
    flushCounters(cnt, counters);
}

// Film::AccumulateColor (Film.cpp:25-39): float3 sum buffers, tight stride, row y = tile row y.  The passes of a
// batch are added per pixel IN PASS ORDER, so the float sum is the one the reference builds pass after pass; the
// secondary sum receives the even passes (Viewport.cpp:303).
__global__ void __launch_bounds__(RT_BLOCK) k_accumulate(const Paths paths, uint32_t slotsPerPass, uint32_t numPasses, float* __restrict__ sum,
                                                         float* __restrict__ secondary, uint32_t width, const DevPass* __restrict__ passes,
                                                         unsigned long long* counters)
{
    Counters cnt; zeroCounters(cnt);
    const V4 lightSamplingWeight = load4(passes[0].lightSamplingWeight);
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t pixelSlot = blockIdx.x * blockDim.x + threadIdx.x; pixelSlot < slotsPerPass; pixelSlot += stride)
    {
        const uint32_t pix = ubits(prec(paths, R_RESULT, pixelSlot).w);
        const size_t idx = 3 * ((size_t)(pix >> 16) * width + (pix & 0xFFFFu));
        float sr = sum[idx + 0], sg = sum[idx + 1], sb = sum[idx + 2];
        float tr = secondary[idx + 0], tg = secondary[idx + 1], tb = secondary[idx + 2];
        for (uint32_t b = 0; b < numPasses; ++b)
        {
            const uint32_t slot = b * slotsPerPass + pixelSlot;
            const float4 rResult = prec(paths, R_RESULT, slot);
            V4 resultColor(rResult.x, rResult.y, rResult.z, 0.0f);
            resolvePendingLightSamples(paths, slot, ubits(prec(paths, R_SAMPLER, slot).w), lightSamplingWeight, resultColor, cnt);   // NEE of the path's last vertex
            sr = sr + resultColor.x; sg = sg + resultColor.y; sb = sb + resultColor.z;
            if ((passes[b].passIndex % 2u) == 0u) { tr = tr + resultColor.x; tg = tg + resultColor.y; tb = tb + resultColor.z; }
        }
        sum[idx + 0] = sr; sum[idx + 1] = sg; sum[idx + 2] = sb;
        secondary[idx + 0] = tr; secondary[idx + 1] = tg; secondary[idx + 2] = tb;
    }
    flushCounters(cnt, counters);
}
#endif   // RT_SHADE_FUNCTIONS_ONLY
