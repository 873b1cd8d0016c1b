// rt_dense.inl -- PathTracerMIS / PathTracer passes with DENSE path state (LightSamplingStrategy::Single).  Included by rt_shade.hip (and rt_tail.hip for the vertex body).
//
// The first layout kept a path in the slot of its pixel for its whole life: after a few bounces the live slots are sparse, every
// 16-byte record access of k_shade pulls its own 64/128-byte line from HBM (measured: 3.1x the bytes the kernel needs, L2 hit rate
// 27 %, profiles/r02_diag0_pmc_8.txt) and costs its own L1 access.  Here the state ping-pongs between two arenas: k_shade_dense reads
// the records of bounce k at consecutive slots and writes the survivors to consecutive slots of the other arena, so that every record
// access of every kernel is a fully coalesced 1 KB per wave.
//   * Slot allocation: a block counts its survivors in LDS and takes its range with ONE returning atomic per 256 vertices, on one of
//     RT_DENSE_SHARDS counters selected by the block index (a single word sustains only ~88 returning atomics per microsecond).  An
//     arena is therefore RT_DENSE_SHARDS regions; region s holds its live paths upwards from s * shardCapacity.
//   * A path that ends at a vertex whose next-event request still needs its shadow ray becomes a ZOMBIE: only what the resolution
//     needs is written, downwards from the top of the region; the next k_shade_dense folds the visibility result in and parks the
//     radiance.
//   * Finished paths park their radiance at home[pass * slotsPerPass + pixelSlot]; k_accumulate_home adds the passes of a batch to the
//     film per pixel in pass order -- the float sums are those of the reference's pass-after-pass accumulation, whatever order the
//     paths were compacted in.
// Arithmetic and consumption order of the samples are those of k_shade (same functions, same sequence): the images are bit-identical.
#ifndef RT_SHADE_FUNCTIONS_ONLY
// k_generate for dense state: slot i = home i; the regions of the first arena are simply filled one after the other
__global__ void __launch_bounds__(RT_BLOCK) k_generate_dense(const RtSceneDesc scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass, const Paths paths,
                                                             const uint32_t* __restrict__ slotPixel, uint32_t numSlots, uint32_t shardCapacity, uint32_t* __restrict__ counts,
                                                             unsigned long long* counters, uint32_t fullRecords)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < numSlots; slot += stride)
    {
        const uint32_t passInBatch = slot / slotsPerPass;
        const DevPass& pass = passes[passInBatch];
        const uint32_t pix = slotPixel[slot - passInBatch * slotsPerPass];
        const uint32_t x = pix & 0xFFFFu, y = pix >> 16;
        const uint32_t realY = pass.height - 1u - y;
        const float invW = 1.0f / (float)(int32_t)pass.width, invH = 1.0f / (float)(int32_t)pass.height;
        const V4 coords(((float)(int32_t)x + pass.sampleOffset[0]) * invW, ((float)(int32_t)realY + pass.sampleOffset[1]) * invH, 0.0f, 0.0f);
        Sampler sampler;
        sampler.seed = pass.seed; sampler.numDims = pass.numDimensions; sampler.blueNoiseLayers = pass.blueNoiseLayers; sampler.blueNoise = scene.blueNoise;
        sampler.resetPixel(x, y, pass.rngKey[0], pass.rngKey[1]);
        V4 origin, direction;
        cameraGenerateRayParts(pass.camera, coords, sampler, origin, direction);
        stStream(prec(paths, R_ORIGIN, slot), f4(origin.x, origin.y, origin.z, fbits(0x100u)));   // depth 0, lastSpecular = true (PathTracerMIS.h:29-34)
        stStream(prec(paths, R_DIR, slot), f4(direction.x, direction.y, direction.z, 1.0f));        // lastPdfW = 1
        if (fullRecords != 0u)
        {
            // (the consumers that do not rebuild a fresh path's other records from its slot: see denseShadeVertex)
            stStream(prec(paths, R_TP, slot), f4(1.0f, 1.0f, 1.0f, 1.0f));
            stStream(prec(paths, R_RESULT, slot), f4(0.0f, 0.0f, 0.0f, fbits(pix)));
            stStream(prec(paths, R_SH_TP, slot), f4(0.0f, 0.0f, 0.0f, fbits(slot)));                    // .w: the path's home
            storeSampler(sampler, paths, slot, 0.0f, 0u);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < RT_DENSE_SHARDS)
    {
        const uint32_t first = threadIdx.x * shardCapacity;
        counts[threadIdx.x] = first >= numSlots ? 0u : (numSlots - first < shardCapacity ? numSlots - first : shardCapacity);
        if (threadIdx.x == 0) atomicAdd(&counters[C_PRIMARY], (unsigned long long)numSlots);
    }
}

#endif   // RT_SHADE_FUNCTIONS_ONLY

// What a vertex leaves behind (denseShadeVertex): outcome 0 nothing (its radiance is parked at home[]), 1 a live path, 2 a zombie (radiance + one
// pending next-event request); the records of the survivor, for the caller to store -- k_shade_dense at a fresh dense slot of the other arena,
// k_tail (rt_tail.hip) in place.
struct DenseVertex
{
    uint32_t outcome;
    float4 oOrigin, oDir, oTp, oResult, oSampler, oRng;
    bool stagedShTp;     // stage[3][thread] holds R_SH_TP (throughput at the vertex | home); otherwise it is {0, 0, 0, home}
    bool rayNeeded;      // the vertex's next-event request needs its shadow ray
    uint32_t oHome, rayMask;
};

// The body of PathTracerMIS::RenderPixel's loop for one path vertex (PathTracerMIS.cpp:276-395) over the records of arena `in` at `slot`;
// kPlain: PathTracer::RenderPixel (Core/Rendering/PathTracer.cpp:73-171).  `stage` = four LDS rows of RT_BLOCK float4 (the next-event request
// waits there, see k_shade_dense).
template <int kLean, bool kPlain, bool kAll>
__device__ __forceinline__ static void denseShadeVertex(const RtSceneDesc& scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass, const DevPass& pass, const Paths& in,
                                                        uint32_t slot, bool zombie, V4 lightSamplingWeight, V4 bsdfSamplingWeight, float lightPickProbability,
                                                        float4 (*stage)[RT_BLOCK], float4* __restrict__ home, Counters& cnt, DenseVertex& v,
                                                        const uint32_t* __restrict__ primarySlotPixel = nullptr)
{
    // Bounce 0 (primarySlotPixel != null): k_generate_dense stores only what depends on the camera sample -- origin and direction.  The other five
    // records of a fresh path are functions of its slot (radiance 0 | pixel, throughput 1, home = slot, the sampler and the per-pixel generator as
    // the camera left them) and are rebuilt here instead of being written and read back: 80 of 112 bytes per path in each direction.
    const bool primary = primarySlotPixel != nullptr;
    float4 rResult, rSampler, rShTp;
    if (primary)
    {
        rResult = f4(0.0f, 0.0f, 0.0f, fbits(primarySlotPixel[slot - (slot / slotsPerPass) * slotsPerPass]));
        rSampler = f4(prec(in, R_SAMPLER, slot).x, 0.0f, 0.0f, fbits(0u));   // .x: the hit's v, written by the traversal; no request is pending
        rShTp = f4(0.0f, 0.0f, 0.0f, fbits(slot));
    }
    else { rResult = ldStream(prec(in, R_RESULT, slot)); rSampler = ldStream(prec(in, R_SAMPLER, slot)); rShTp = ldStream(prec(in, R_SH_TP, slot)); }
    const uint32_t pix = ubits(rResult.w), homeIndex = ubits(rShTp.w);
    v.oHome = homeIndex;
    V4 resultColor(rResult.x, rResult.y, rResult.z, 0.0f);
    resolvePendingLightSamples(in, slot, ubits(rSampler.w), lightSamplingWeight, resultColor, cnt);   // NEE of the previous vertex
    if (!zombie)
    {
        const float4 rOrigin = ldStream(prec(in, R_ORIGIN, slot)), rDir = ldStream(prec(in, R_DIR, slot)), rHit = ldStream(prec(in, R_HIT, slot));
        float4 rTp = f4(1.0f, 1.0f, 1.0f, 1.0f);
        if (!primary) rTp = ldStream(prec(in, R_TP, slot));
        const uint32_t flags = ubits(rOrigin.w);
        const uint32_t depth = flags & 0xFFu;
        const bool lastSpecular = (flags & 0x100u) != 0;
        const float lastPdfW = rDir.w;
        const Ray ray = makePathRay(rOrigin, rDir, depth);
        V4 throughput(rTp.x, rTp.y, rTp.z, rTp.w);
        Hit hit;
        hit.objectId = ubits(rHit.x); hit.subObjectId = ubits(rHit.y); hit.distance = rHit.z; hit.u = rHit.w; hit.v = rSampler.x;
        uint32_t numRequests = 0;
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
            sd.intersection.material = (flags >> 9) - 1u;   // the previous vertex's material (see k_shade)
            if (hit.distance < FLT_MAX) sceneEvaluateIntersection<kLean>(scene, ray, hit, sd.intersection, cnt);
            if (!RT_LEAN(kLean) && hit.subObjectId == RT_LIGHT_OBJECT)
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
                else resultColor = mulAdd(throughput, zero4(), resultColor);
                break;
            }
            sd.outgoingDirWorldSpace = neg(ray.dir);
            const RtMaterial& mat = scene.materials[sd.intersection.material];
            materialEvaluateShadingData<kLean>(scene, mat, sd);
            resultColor = mulAdd(throughput, kPlain ? sd.mp.emission : sd.mp.emission * bsdfSamplingWeight, resultColor);   // emission, :309-317

            Sampler sampler;
            if (primary)
            {
                // the sampler and the per-pixel generator as Camera::GenerateRay left them: reset as k_generate_dense does and let the camera draw again
                const DevPass& own = passes[homeIndex / slotsPerPass];
                const uint32_t x = pix & 0xFFFFu, y = pix >> 16;
                sampler.seed = own.seed; sampler.numDims = own.numDimensions; sampler.blueNoiseLayers = own.blueNoiseLayers; sampler.blueNoise = scene.blueNoise;
                sampler.resetPixel(x, y, own.rngKey[0], own.rngKey[1]);
                const float invW = 1.0f / (float)(int32_t)own.width, invH = 1.0f / (float)(int32_t)own.height;
                const V4 coords(((float)(int32_t)x + own.sampleOffset[0]) * invW, ((float)(int32_t)(own.height - 1u - y) + own.sampleOffset[1]) * invH, 0.0f, 0.0f);
                V4 cameraOrigin, cameraDirection;
                cameraGenerateRayParts(own.camera, coords, sampler, cameraOrigin, cameraDirection);
            }
            else { loadSampler(sampler, in, slot, pix, rSampler, pass, scene.blueNoise); sampler.seed = passes[homeIndex / slotsPerPass].seed; }

            // SampleLights (next event estimation), PathTracerMIS.cpp:125-155
            if (!kPlain && kAll && scene.numLights != 0)
            {
                for (uint32_t l = 0; l < scene.numLights; ++l)
                {
                    float4 dirTmax, contribution;
                    if (computeLightSample<kLean>(scene, pass, sampler, scene.lights[l], sd, mat, depth, lightPickProbability, dirTmax, contribution)) v.rayMask |= 1u << l;
                    pshadow(in, l, 0, slot) = dirTmax; pshadow(in, l, 1, slot) = contribution;
                }
                v.rayNeeded = v.rayMask != 0u;
                numRequests = v.rayNeeded ? scene.numLights : 0u;
                stage[2][threadIdx.x] = f4(sd.intersection.frame.r[3].x, sd.intersection.frame.r[3].y, sd.intersection.frame.r[3].z, 0.0f);
                stage[3][threadIdx.x] = f4(throughput.x, throughput.y, throughput.z, fbits(homeIndex));
                v.stagedShTp = true;
            }
            else if (!kPlain && scene.numLights != 0)
            {
                uint32_t lightIndex = 0;
                if (scene.numLights > 1) lightIndex = sampler.fallbackInt() % scene.numLights;
                float4 oShadow0, oShadow1;
                v.rayNeeded = computeLightSample<kLean>(scene, pass, sampler, scene.lights[lightIndex], sd, mat, depth, lightPickProbability, oShadow0, oShadow1);
                numRequests = v.rayNeeded ? 1u : 0u;   // a request without a ray contributes nothing (resolvePendingLightSamples skips it): not kept
                stage[0][threadIdx.x] = oShadow0; stage[1][threadIdx.x] = oShadow1;
                stage[2][threadIdx.x] = f4(sd.intersection.frame.r[3].x, sd.intersection.frame.r[3].y, sd.intersection.frame.r[3].z, 0.0f);
                stage[3][threadIdx.x] = f4(throughput.x, throughput.y, throughput.z, fbits(homeIndex));
                v.stagedShTp = true;
            }
            bool cont = depth < pass.maxRayDepth;
            if (cont && depth >= pass.minRussianRouletteDepth)   // Russian roulette, :330-347
            {
                const float minColorValue = 0.125f;
                const float threshold = minColorValue + (1.0f - minColorValue) * colorMax(sd.mp.baseColor);
                if (sampler.getFloat() > threshold) cont = false;
                else throughput = throughput * (1.0f / threshold);
            }
            if (cont)   // BSDF sampling, :349-395
            {
                float pdf = 0.0f; V4 incomingDirWorldSpace = zero4(); uint32_t event = EV_NULL;
                float u[3]; u[0] = sampler.getFloat(); u[1] = sampler.getFloat(); u[2] = sampler.getFloat();
                const V4 bsdfValue = materialSample<kLean>(mat, sd, u, incomingDirWorldSpace, pdf, event);
                if (event != EV_NULL)
                {
                    throughput = throughput * bsdfValue;
                    if (!almostZero4(throughput))
                    {
                        v.oOrigin = f4(sd.intersection.frame.r[3].x, sd.intersection.frame.r[3].y, sd.intersection.frame.r[3].z,
                                     fbits((depth + 1u) | (((event & EV_SPECULAR) != 0) ? 0x100u : 0u) | ((sd.intersection.material + 1u) << 9)));
                        v.oDir = f4(incomingDirWorldSpace.x, incomingDirWorldSpace.y, incomingDirWorldSpace.z, pdf);
                        v.oTp = f4(throughput.x, throughput.y, throughput.z, throughput.w);
                        v.outcome = 1;
                    }
                }
            }
            if (v.outcome == 1)
            {
                v.oSampler = f4(0.0f, fbits(sampler.salt), fbits(sampler.generated), fbits(numRequests));
                v.oRng = f4(fbits((uint32_t)sampler.fallback.s[0]), fbits((uint32_t)(sampler.fallback.s[0] >> 32)),
                          fbits((uint32_t)sampler.fallback.s[1]), fbits((uint32_t)(sampler.fallback.s[1] >> 32)));
                if (numRequests == 0u) v.stagedShTp = false;
            }
            else if (v.rayNeeded)
            {
                v.outcome = 2;   // the path ends here, its last next-event sample still needs its shadow ray
                v.oSampler = f4(0.0f, 0.0f, 0.0f, fbits(numRequests));
            }
        } while (false);
        if (v.outcome != 1) cnt.c[C_RAYS] += depth + 1u;   // counters.numRays += depth + 1, PathTracerMIS.cpp:412
    }
    v.oResult = f4(resultColor.x, resultColor.y, resultColor.z, fbits(pix));
    if (v.outcome == 0) stStream(home[homeIndex], f4(resultColor.x, resultColor.y, resultColor.z, 0.0f));
}

#ifndef RT_SHADE_FUNCTIONS_ONLY
// The body of PathTracerMIS::RenderPixel's loop for one path vertex (PathTracerMIS.cpp:276-395), as k_shade, reading arena `in`
// and writing the survivors densely into arena `out`.  kPlain: PathTracer::RenderPixel (Core/Rendering/PathTracer.cpp:73-171).
//
// kAll: LightSamplingStrategy::All (PathTracerMIS.cpp:141-147: every light is sampled at every vertex, up to RT_DENSE_MAX_LIGHTS of them here).
// The 2 x numLights request records of a vertex cannot wait in registers or LDS for its output slot, so they are written to the vertex's OWN
// slot of the input arena first -- its previous requests were folded in at the top of the iteration, the space is free -- and copied to the
// output slot once the block has allocated it (the copy reads what the same lane just wrote: L2 hits).
template <int kLean, bool kPlain = false, bool kAll = false>
__global__ void RT_SHADE_DENSE_ATTR(kLean, kAll) k_shade_dense(const RtSceneDesc scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass, const Paths in, const Paths out,
                                                          const DenseCounts dense, uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount,
                                                          float4* __restrict__ home, unsigned long long* counters)
{
    __shared__ uint32_t sShadowBuf[RT_APPEND_BUFFER];
    // The four records of a vertex's next-event request are known long before the vertex knows whether and where it survives (Russian roulette,
    // BSDF sampling and the block's slot allocation come after): they wait in LDS instead of 16 registers.  142 -> 127 VGPRs for the lean
    // variant = four waves per SIMD instead of three in a kernel that spends most of its time waiting for dependent gathers (-14 % kernel time).
    __shared__ float4 sStage[4][RT_BLOCK];   // [request direction | contribution | shading point | throughput at the vertex][thread]
    __shared__ uint32_t sShadowCount, sShadowBase;
    __shared__ uint32_t sLive, sZombies, sLiveBase, sZombieBase;
    __shared__ uint32_t sLivePrefix[RT_DENSE_SHARDS + 1u], sZombiePrefix[RT_DENSE_SHARDS + 1u];
    if (threadIdx.x == 0) { sShadowCount = 0; sLive = 0; sZombies = 0; }
    denseLoadPrefix(dense.in, sLivePrefix);
    if (threadIdx.x == 64)
    {
        uint32_t sum = 0;
        for (uint32_t s = 0; s < RT_DENSE_SHARDS; ++s)
        {
            sZombiePrefix[s] = sum; sum += dense.in[RT_DENSE_SHARDS + s];
            if (blockIdx.x == 0 && dense.in[s] + dense.in[RT_DENSE_SHARDS + s] > dense.shardCapacity) dense.errorFlags[0] = 1u;   // the launch before this one overfilled region s
        }
        sZombiePrefix[RT_DENSE_SHARDS] = sum;
    }
    __syncthreads();
    Counters cnt; zeroCounters(cnt);
    const uint32_t numLive = sLivePrefix[RT_DENSE_SHARDS], count = numLive + sZombiePrefix[RT_DENSE_SHARDS];
    const uint32_t stride = gridDim.x * blockDim.x;
    const DevPass pass = passes[0];   // the structural parameters are those of every pass of the batch
    const V4 lightSamplingWeight = load4(pass.lightSamplingWeight), bsdfSamplingWeight = load4(pass.bsdfSamplingWeight);
    const float lightPickProbability = kAll ? 1.0f : 1.0f / (float)(scene.numLights ? scene.numLights : 1u);   // GetLightPickingProbability, PathTracerMIS.cpp:157-172

    const uint32_t rounded = (count + RT_BLOCK - 1) / RT_BLOCK * RT_BLOCK;
    // the i-th vertex of the launch: live paths first (region by region), then the zombies (from the top of their regions)
    auto vertexSlot = [&](uint32_t idx, bool& zombie) -> uint32_t
    {
        zombie = idx >= numLive;
        if (!zombie) return denseLiveSlot(sLivePrefix, dense.shardCapacity, idx);
        const uint32_t z = idx - numLive, s = denseRegionOf(sZombiePrefix, z);
        return (s + 1u) * dense.shardCapacity - 1u - (z - sZombiePrefix[s]);
    };
    for (uint32_t first = blockIdx.x * blockDim.x; first < rounded; first += stride)
    {
        const uint32_t i = first + threadIdx.x;
        // what this vertex leaves behind: 0 nothing (radiance parked), 1 a live path, 2 a zombie (radiance + one pending request)
        DenseVertex v;
        v.outcome = 0; v.stagedShTp = false; v.rayNeeded = false; v.oHome = 0u; v.rayMask = 0u;
        uint32_t inSlot = 0u;
        if (i < count)
        {
            bool zombie;
            const uint32_t slot = vertexSlot(i, zombie);
            inSlot = slot;
            denseShadeVertex<kLean, kPlain, kAll>(scene, passes, slotsPerPass, pass, in, slot, zombie, lightSamplingWeight, bsdfSamplingWeight, lightPickProbability, sStage, home, cnt, v,
                                                  dense.primarySlotPixel);
        }
        uint32_t outcome = v.outcome;

        // dense slots of the other arena: ranks from LDS counters, the block's two ranges with one global atomic each.  The region follows
        // the CHUNK of 256 vertices, not the block: consecutive chunks take the regions in turn whatever the grid size, so the regions
        // fill evenly (a region is a sixteenth of the arena plus a margin of 65536 slots); what still does not fit raises a flag the host
        // checks instead of overwriting the neighbouring region.
        const uint32_t shard = (first / RT_BLOCK) & (RT_DENSE_SHARDS - 1u);
        uint32_t rank = 0;
        if (outcome == 1) rank = atomicAdd(&sLive, 1u);
        else if (outcome == 2) rank = atomicAdd(&sZombies, 1u);
        __syncthreads();
        if (threadIdx.x == 0 && sLive != 0u) sLiveBase = atomicAdd(&dense.out[shard], sLive);
        if (threadIdx.x == 64 && sZombies != 0u) sZombieBase = atomicAdd(&dense.out[RT_DENSE_SHARDS + shard], sZombies);
        __syncthreads();
        // never outside the region (live paths grow upwards from its start, zombies downwards from its end); whether the two met is
        // checked on the final counts by the next launch's prologue above
        if ((outcome == 1 && sLiveBase + rank >= dense.shardCapacity) || (outcome == 2 && sZombieBase + rank >= dense.shardCapacity)) 
        {
            // the vertex is dropped and the frame is invalid until the next reset (every synchronising call reports the flag); its home still gets what
            // the path had gathered, so that k_accumulate_home never adds a previous batch's value
            dense.errorFlags[0] = 1u; outcome = 0;
            stStream(home[v.oHome], f4(v.oResult.x, v.oResult.y, v.oResult.z, 0.0f));
        }
        if (outcome != 0)
        {
            const uint32_t slot = outcome == 1 ? shard * dense.shardCapacity + sLiveBase + rank : (shard + 1u) * dense.shardCapacity - 1u - (sZombieBase + rank);
            stStream(prec(out, R_RESULT, slot), v.oResult);
            stStream(prec(out, R_SAMPLER, slot), v.oSampler);
            stStream(prec(out, R_SH_TP, slot), v.stagedShTp ? sStage[3][threadIdx.x] : f4(0.0f, 0.0f, 0.0f, fbits(v.oHome)));
            if (outcome == 1) { stStream(prec(out, R_ORIGIN, slot), v.oOrigin); stStream(prec(out, R_DIR, slot), v.oDir); stStream(prec(out, R_TP, slot), v.oTp); stStream(prec(out, R_RNG, slot), v.oRng); }
            if (ubits(v.oSampler.w) != 0u)
            {
                stStream(prec(out, R_SH_P, slot), sStage[2][threadIdx.x]);
                if (kAll)
                {
                    for (uint32_t l = 0; l < scene.numLights; ++l)
                    {
                        stStream(pshadow(out, l, 0, slot), pshadow(in, l, 0, inSlot)); stStream(pshadow(out, l, 1, slot), pshadow(in, l, 1, inSlot));
                        if (v.rayMask & (1u << l)) sShadowBuf[atomicAdd(&sShadowCount, 1u)] = l * out.capacity + slot;
                    }
                }
                else
                {
                stStream(pshadow(out, 0, 0, slot), sStage[0][threadIdx.x]);
                stStream(pshadow(out, 0, 1, slot), sStage[1][threadIdx.x]);
                if (v.rayNeeded) sShadowBuf[atomicAdd(&sShadowCount, 1u)] = slot;   // request index = light 0 * capacity + slot
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) { sLive = 0; sZombies = 0; }
        const bool last = first + stride >= rounded;
        if (last || sShadowCount + RT_BLOCK * (kAll ? RT_DENSE_MAX_LIGHTS : 1u) > RT_APPEND_BUFFER) flushAppendBuffer(sShadowBuf, sShadowCount, sShadowBase, shadowQueue, shadowCount);
        else __syncthreads();
    }
    flushCounters(cnt, counters);
}

// Film::AccumulateColor (Film.cpp:25-39) from the parked radiance: the passes of a batch are added per pixel IN PASS ORDER; the
// secondary sum receives the even passes (Viewport.cpp:303)
__global__ void __launch_bounds__(RT_BLOCK) k_accumulate_home(const float4* __restrict__ home, const uint32_t* __restrict__ slotPixel, uint32_t slotsPerPass, uint32_t numPasses,
                                                              float* __restrict__ sum, float* __restrict__ secondary, uint32_t width, const DevPass* __restrict__ passes)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t pixelSlot = blockIdx.x * blockDim.x + threadIdx.x; pixelSlot < slotsPerPass; pixelSlot += stride)
    {
        const uint32_t pix = slotPixel[pixelSlot];
        const size_t idx = 3 * ((size_t)(pix >> 16) * width + (pix & 0xFFFFu));
        float sr = sum[idx + 0], sg = sum[idx + 1], sb = sum[idx + 2];
        float tr = secondary[idx + 0], tg = secondary[idx + 1], tb = secondary[idx + 2];
        for (uint32_t b = 0; b < numPasses; ++b)
        {
            const float4 c = ldStream(home[(size_t)b * slotsPerPass + pixelSlot]);
            sr = sr + c.x; sg = sg + c.y; sb = sb + c.z;
            if ((passes[b].passIndex % 2u) == 0u) { tr = tr + c.x; tg = tg + c.y; tb = tb + c.z; }
        }
        sum[idx + 0] = sr; sum[idx + 1] = sg; sum[idx + 2] = sb;
        secondary[idx + 0] = tr; secondary[idx + 1] = tg; secondary[idx + 2] = tb;
    }
}
#endif   // RT_SHADE_FUNCTIONS_ONLY
