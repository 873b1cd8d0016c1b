// rt_vcm.inl -- wavefront form of the reference's bidirectional integrator (Core/Rendering/VertexConnectionAndMerging.cpp),
// included by rt_shade.hip (it reuses the path-record arena, the persistent traversal kernel and the context).
//
// One pass over the whole frame (RenderPixel, :172-318, for every pixel):
//
//   generate                     camera rays (as for PathTracerMIS)                                   Viewport.cpp:305-331
//   photon grid                  HashGrid::Build over the photons recorded by the PREVIOUS pass        :126-170, HashGrid.h:17-71
//   vcm_emit                     GenerateLightSample                                                    :428-491
//   for vertex = 1 .. maxPathLength-1:                                         TraceLightPath,         :320-426
//       trace                    closest hit of the light sub-paths + the camera-connection shadow rays of the previous vertex
//       vcm_light_shade          splat the previous vertex's connection if visible; store the light vertex and the photon;
//                                ConnectToCamera (:908-966) up to the shadow ray; AdvancePath (:493-578)
//   trace + vcm_light_finish     the connections of each path's last vertex
//   for vertex = 1 .. maxPathLength:                                            camera sub-path,        :184-314
//       trace                    closest hit + the shadow rays queued by the previous vertex
//       vcm_camera_shade         fold the previous vertex's visible next-event / connection / merging terms into the radiance;
//                                light hit (EvaluateLight :580-635); SampleLights (:637-731), ConnectVertices (:746-821) to the
//                                pixel's light vertices and MergeVertices (:823-906) at this vertex; AdvancePath
//   trace + vcm_camera_finish    last vertex's terms, Film::AccumulateColor
//
// Shadow-ray results arrive one kernel after the terms that need them are computed; the terms are stored per request and
// folded in afterwards IN THE REFERENCE'S ORDER (lights, then light vertices, then merging), so the camera-path radiance is
// independent of the schedule (bit-identical to a scalar evaluation of the same pixel).  Film splats are float atomics.

#include "rt_vcm_state.h"

RT_DEV float vcmPdfWtoA(float pdfW, float distance, float cosThere) { return pdfW * Abs(cosThere) / Sqr(distance); }   // :25-28

RT_DEV void loadSimd(RandomSimd& r, const VcmArena& a, uint32_t slot)
{
    const float4 s0 = vrec(a, V_SIMD0, slot), s1 = vrec(a, V_SIMD1, slot);
    r.seed0[0] = (uint64_t)ubits(s0.x) | ((uint64_t)ubits(s0.y) << 32); r.seed0[1] = (uint64_t)ubits(s0.z) | ((uint64_t)ubits(s0.w) << 32);
    r.seed1[0] = (uint64_t)ubits(s1.x) | ((uint64_t)ubits(s1.y) << 32); r.seed1[1] = (uint64_t)ubits(s1.z) | ((uint64_t)ubits(s1.w) << 32);
}
RT_DEV void storeSimd(const RandomSimd& r, const VcmArena& a, uint32_t slot)
{
    vrec(a, V_SIMD0, slot) = f4(fbits((uint32_t)r.seed0[0]), fbits((uint32_t)(r.seed0[0] >> 32)), fbits((uint32_t)r.seed0[1]), fbits((uint32_t)(r.seed0[1] >> 32)));
    vrec(a, V_SIMD1, slot) = f4(fbits((uint32_t)r.seed1[0]), fbits((uint32_t)(r.seed1[0] >> 32)), fbits((uint32_t)r.seed1[1]), fbits((uint32_t)(r.seed1[1] >> 32)));
}

// Queue space for a whole block with ONE global atomic (a returning atomic on one word sustains only ~88 operations per
// microsecond on this chip): every thread asks for n entries (0 allowed) and gets the index of its first one.  Called by
// all threads of the block at a block-uniform point; *sCount must be 0 on entry and is 0 again on return.
RT_DEV uint32_t blockReserve(uint32_t n, uint32_t* __restrict__ globalCount, uint32_t* sCount, uint32_t* sBase)
{
    const uint32_t local = n ? atomicAdd(sCount, n) : 0u;
    __syncthreads();
    if (threadIdx.x == 0) { const uint32_t total = *sCount; *sBase = total ? atomicAdd(globalCount, total) : 0u; *sCount = 0u; }
    __syncthreads();
    return *sBase + local;
}

// ShadingData of a stored light vertex: fetch and decode are separate so that a loop can fetch the next vertex while it works on this one
struct LightVertexRecords { float4 r0, r1, r2, r3, r4, r5; };
RT_DEV LightVertexRecords fetchLightVertex(const VcmArena& a, uint32_t vertex, uint32_t slot)
{
    LightVertexRecords v;
    v.r0 = lvrec(a, vertex, 0, slot); v.r1 = lvrec(a, vertex, 1, slot); v.r2 = lvrec(a, vertex, 2, slot);
    v.r3 = lvrec(a, vertex, 3, slot); v.r4 = lvrec(a, vertex, 4, slot); v.r5 = lvrec(a, vertex, 5, slot);
    return v;
}
RT_DEV void decodeLightVertex(const RtSceneDesc& scene, const LightVertexRecords& v, ShadingData& sd, V4& throughput, float& dVC, float& dVCM, uint32_t& pathLength)
{
    const float4 r0 = v.r0, r1 = v.r1, r2 = v.r2, r3 = v.r3, r4 = v.r4, r5 = v.r5;
    sd.intersection.frame.r[0] = V4(r1.x, r1.y, r1.z, 0.0f);
    sd.intersection.frame.r[2] = V4(r2.x, r2.y, r2.z, 0.0f);
    sd.intersection.frame.r[1] = cross3(sd.intersection.frame.r[0], sd.intersection.frame.r[2]);   // as Scene::EvaluateIntersection left it (Scene.cpp:345)
    sd.intersection.frame.r[3] = V4(r0.x, r0.y, r0.z, 0.0f);
    sd.intersection.texCoord = zero4();
    sd.intersection.material = ubits(r0.w) & 0x00FFFFFFu;
    pathLength = ubits(r0.w) >> 24;
    sd.outgoingDirWorldSpace = V4(r3.x, r3.y, r3.z, 0.0f);
    sd.mp.baseColor = V4(r4.x, r4.y, r4.z, r4.w); sd.mp.emission = zero4();
    sd.mp.roughness = r1.w; sd.mp.metalness = r2.w; sd.mp.IoR = scene.materials[sd.intersection.material].IoR;
    throughput = V4(r5.x, r5.y, r5.z, 0.0f);
    dVC = r3.w; dVCM = r5.w;
}
RT_DEV void loadLightVertex(const RtSceneDesc& scene, const VcmArena& a, uint32_t vertex, uint32_t slot, ShadingData& sd, V4& throughput, float& dVC, float& dVCM, uint32_t& pathLength)
{
    decodeLightVertex(scene, fetchLightVertex(a, vertex, slot), sd, throughput, dVC, dVCM, pathLength);
}

// ---- light stage ---------------------------------------------------------------------------------------------------------

// GenerateLightSample, :428-491.  The scalar generator (Random::GetInt) is the pixel's Sampler::fallback stream, which lives in
// the CAMERA arena's R_RNG record (k_generate has reset it for this pass).
template <int kClass>
__global__ void RT_VCM_ATTR(k_vcm_emit) k_vcm_emit(const RtSceneDesc scene, const VcmBatch b, const Paths lp, const Paths cp,
                                                       const VcmArena a, const uint32_t* __restrict__ slotPixel, uint32_t numSlots,
                                                       uint32_t* __restrict__ queue, uint32_t* __restrict__ queueCount)
{
    __shared__ uint32_t sCount, sBase;
    if (threadIdx.x == 0) sCount = 0u;
    __syncthreads();
    const uint32_t rounded = (numSlots + RT_BLOCK - 1u) / RT_BLOCK * RT_BLOCK;   // whole blocks iterate together (blockReserve synchronises)
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < rounded; slot += stride)
    {
        bool ok = false;
        if (slot < numSlots)
        {
            const uint32_t pb = slot / b.slotsPerPass;
            const DevPass& pass = b.passes[pb];
            const VcmDev& vcm = b.vcms[pb];
            const uint32_t pix = slotPixel[slot - pb * b.slotsPerPass];
            const uint32_t x = pix & 0xFFFFu, y = pix >> 16;
            RandomSimd simd; simd.resetPixel(x, y, pass.rngKey[0], pass.rngKey[1]);
            a.lvCount[slot] = 0u; a.photonCount[slot] = 0u;
            prec(lp, R_SAMPLER, slot) = f4(0.0f, 0.0f, 0.0f, fbits(0u));
            if (scene.numLights != 0u)
            {
                const float4 rng = prec(cp, R_RNG, slot);
                Xoroshiro fb;
                fb.s[0] = (uint64_t)ubits(rng.x) | ((uint64_t)ubits(rng.y) << 32); fb.s[1] = (uint64_t)ubits(rng.z) | ((uint64_t)ubits(rng.w) << 32);
                const float lightPickProbability = 1.0f / (float)scene.numLights;
                const uint32_t lightIndex = (uint32_t)xoroshiroNext(fb) % scene.numLights;
                prec(cp, R_RNG, slot) = f4(fbits((uint32_t)fb.s[0]), fbits((uint32_t)(fb.s[0] >> 32)), fbits((uint32_t)fb.s[1]), fbits((uint32_t)(fb.s[1] >> 32)));
                const RtLight& light = scene.lights[lightIndex];
                const V4 ps = simd.getVector4(); const V4 ds = simd.getVector4();
                const float up[3] = { ps.x, ps.y, ps.z }, ud[2] = { ds.x, ds.y };
                EmitResult er; er.position = zero4(); er.direction = zero4(); er.directPdfA = er.emissionPdfW = er.cosAtLight = 0.0f;
                const V4 emitted = lightEmit<kClass>(scene, light, up, ud, er);
                if (!almostZero4(emitted))
                {
                    er.directPdfA *= lightPickProbability;
                    er.emissionPdfW *= lightPickProbability;
                    const float emissionInvPdfW = 1.0f / er.emissionPdfW;
                    er.position = er.position + er.direction * 0.0005f;
                    const V4 throughput = emitted * emissionInvPdfW;
                    const bool isFiniteLight = (light.flags & RT_LIGHT_FLAG_FINITE) != 0u;
                    const float dVCM = er.directPdfA * emissionInvPdfW;
                    float dVC = 0.0f;
                    if ((light.flags & RT_LIGHT_FLAG_DELTA) == 0u)
                    {
                        const float cosAtLight = isFiniteLight ? er.cosAtLight : 1.0f;
                        dVC = cosAtLight * emissionInvPdfW;
                    }
                    const float dVM = dVC * vcm.misVertexConnectionWeightFactorVC;
                    prec(lp, R_ORIGIN, slot) = f4(er.position.x, er.position.y, er.position.z, fbits(0u));   // "depth" 0: no 1e-3 offset on the emitted ray
                    prec(lp, R_DIR, slot) = f4(er.direction.x, er.direction.y, er.direction.z, 0.0f);
                    prec(lp, R_TP, slot) = f4(throughput.x, throughput.y, throughput.z, throughput.w);
                    vrec(a, V_MIS, slot) = f4(dVC, dVM, dVCM, fbits(1u | (isFiniteLight ? 0x200u : 0u)));
                    ok = true;
                }
            }
            storeSimd(simd, a, slot);
        }
        const uint32_t at = blockReserve(ok ? 1u : 0u, queueCount, &sCount, &sBase);
        if (ok) queue[at] = slot;
    }
}

// the pending camera connection of a light-path vertex: splat it if the shadow ray reached the camera
RT_DEV void resolveSplat(const Paths& lp, uint32_t slot, float* __restrict__ sum, float* __restrict__ secondary, bool evenPass, Counters& cnt)
{
    const float4 dirTmax = pshadow(lp, 0, 0, slot);
    if (dirTmax.w < 0.0f) return;
    cnt.c[C_SHADOW_HIT]++;
    const float4 c = pshadow(lp, 0, 1, slot);
    const uint32_t target = ubits(c.w);
    if (target == 0xFFFFFFFFu) return;
    atomicAdd(&sum[3 * (size_t)target + 0], c.x); atomicAdd(&sum[3 * (size_t)target + 1], c.y); atomicAdd(&sum[3 * (size_t)target + 2], c.z);
    if (evenPass) { atomicAdd(&secondary[3 * (size_t)target + 0], c.x); atomicAdd(&secondary[3 * (size_t)target + 1], c.y); atomicAdd(&secondary[3 * (size_t)target + 2], c.z); }
}

// AdvancePath, :493-578.  Returns false when the walk ends; on success the new ray / throughput / MIS quantities are stored.
RT_DEV bool vcmAdvancePath(const RtSceneDesc& scene, const VcmDev& vcm, const Paths& p, const VcmArena& a, uint32_t slot, const ShadingData& sd, const RtMaterial& mat,
                           const float sample[3], V4 throughput, float dVC, float dVM, float dVCM, uint32_t length, uint32_t extraBits)
{
    V4 incomingDirWorldSpace = zero4(); float bsdfDirPdf = 0.0f; uint32_t sampledEvent = EV_NULL;
    const V4 bsdfValue = materialSample<false>(mat, sd, sample, incomingDirWorldSpace, bsdfDirPdf, sampledEvent);
    const float cosThetaOut = Abs(dot3(incomingDirWorldSpace, sd.intersection.frame.r[2]));
    if (sampledEvent == EV_NULL) return false;
    throughput = throughput * bsdfValue;
    if (almostZero4(throughput)) return false;
    length++;
    bool lastSpecular;
    if (sampledEvent & EV_SPECULAR)
    {
        dVC *= cosThetaOut;
        dVM *= cosThetaOut;
        dVCM = 0.0f;
        lastSpecular = true;
    }
    else
    {
        const V4 outgoingLocal = worldToLocal(sd.intersection, sd.outgoingDirWorldSpace);
        const V4 incomingLocal = neg(worldToLocal(sd.intersection, incomingDirWorldSpace));
        const float bsdfRevPdf = bsdfPdf(mat.bsdf, mat, sd.mp, outgoingLocal, incomingLocal, true);
        const float invBsdfDirPdf = 1.0f / bsdfDirPdf;
        const float newVC = (cosThetaOut * invBsdfDirPdf) * (dVC * bsdfRevPdf + dVCM + vcm.misVertexMergingWeightFactorVC);
        const float newVM = (cosThetaOut * invBsdfDirPdf) * (dVM * bsdfRevPdf + dVCM * vcm.misVertexConnectionWeightFactorVC + 1.0f);
        dVC = newVC; dVM = newVM;
        dVCM = invBsdfDirPdf;
        lastSpecular = false;
    }
    // depth field > 0: the traversal adds the 1e-3 offset to Ray(origin, direction) (:528-529); bit 8 = lastSpecular, bits 9.. = material + 1
    prec(p, R_ORIGIN, slot) = f4(sd.intersection.frame.r[3].x, sd.intersection.frame.r[3].y, sd.intersection.frame.r[3].z,
                                 fbits(((length - 1u) & 0xFFu) | (lastSpecular ? 0x100u : 0u) | ((sd.intersection.material + 1u) << 9)));
    prec(p, R_DIR, slot) = f4(incomingDirWorldSpace.x, incomingDirWorldSpace.y, incomingDirWorldSpace.z, 0.0f);
    prec(p, R_TP, slot) = f4(throughput.x, throughput.y, throughput.z, throughput.w);
    vrec(a, V_MIS, slot) = f4(dVC, dVM, dVCM, fbits(length | extraBits));
    return true;
}

// One vertex of TraceLightPath's loop, :334-425
template <int kClass>
__global__ void RT_VCM_ATTR(k_vcm_light_shade) k_vcm_light_shade(const RtSceneDesc scene, const VcmBatch b, const Paths lp, const VcmArena a,
                                                              const uint32_t* __restrict__ queueIn, const uint32_t* __restrict__ countIn,
                                                              uint32_t* __restrict__ queueOut, uint32_t* __restrict__ countOut,
                                                              uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount,
                                                              float* __restrict__ sum, float* __restrict__ secondary, unsigned long long* counters)
{
    __shared__ uint32_t sCount, sBase;
    if (threadIdx.x == 0) sCount = 0u;
    __syncthreads();
    Counters cnt; zeroCounters(cnt);
    const uint32_t count = *countIn;
    const uint32_t rounded = (count + RT_BLOCK - 1u) / RT_BLOCK * RT_BLOCK;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < rounded; i += stride)
    {
        bool alive = false, needRay = false;
        uint32_t slot = 0;
        if (i < count)
        {
            slot = queueIn[i];
            const uint32_t pb = slot / b.slotsPerPass;
            const DevPass& pass = b.passes[pb];
            const VcmDev& vcm = b.vcms[pb];
            const bool evenPass = (pass.passIndex % 2u) == 0u;
            const float4 rOrigin = prec(lp, R_ORIGIN, slot), rDir = prec(lp, R_DIR, slot), rTp = prec(lp, R_TP, slot);
            const float4 rHit = prec(lp, R_HIT, slot), rSampler = prec(lp, R_SAMPLER, slot), rMis = vrec(a, V_MIS, slot);
            if (ubits(rSampler.w) != 0u) resolveSplat(lp, slot, sum, secondary, evenPass, cnt);
            uint32_t pending = 0u;
            const Ray ray = makePathRay(rOrigin, rDir, ubits(rOrigin.w) & 0xFFu);
            V4 throughput(rTp.x, rTp.y, rTp.z, rTp.w);
            float dVC = rMis.x, dVM = rMis.y, dVCM = rMis.z;
            const uint32_t length = ubits(rMis.w) & 0xFFu;
            const bool isFiniteLight = (ubits(rMis.w) & 0x200u) != 0u;
            Hit hit; hit.objectId = ubits(rHit.x); hit.subObjectId = ubits(rHit.y); hit.distance = rHit.z; hit.u = rHit.w; hit.v = rSampler.x;
            cnt.c[C_RAYS]++;
            if (hit.objectId != RT_INVALID_OBJECT && hit.subObjectId != RT_LIGHT_OBJECT)
            {
                ShadingData sd; sd.intersection.material = RT_NO_MATERIAL;
                sceneEvaluateIntersection<kClass>(scene, ray, hit, sd.intersection, cnt);
                sd.outgoingDirWorldSpace = neg(ray.dir);
                const RtMaterial& mat = scene.materials[sd.intersection.material];
                materialEvaluateShadingData<kClass>(scene, mat, sd);
                {
                    if (length > 1u || isFiniteLight) dVCM *= Sqr(hit.distance);
                    const float cosTheta = dot3(ray.dir, sd.intersection.frame.r[2]);
                    const float invMis = 1.0f / Abs(cosTheta);
                    dVCM *= invMis; dVC *= invMis; dVM *= invMis;
                }
                RandomSimd simd; loadSimd(simd, a, slot);
                if (!bsdfIsDelta(mat.bsdf))
                {
                    if (vcm.useVertexConnection)
                    {
                        const uint32_t k = a.lvCount[slot];
                        a.lvCount[slot] = k + 1u;
                        const V4 pos = sd.intersection.frame.r[3], tg = sd.intersection.frame.r[0], nr = sd.intersection.frame.r[2], og = sd.outgoingDirWorldSpace;
                        lvrec(a, k, 0, slot) = f4(pos.x, pos.y, pos.z, fbits(sd.intersection.material | (length << 24)));
                        lvrec(a, k, 1, slot) = f4(tg.x, tg.y, tg.z, sd.mp.roughness);
                        lvrec(a, k, 2, slot) = f4(nr.x, nr.y, nr.z, sd.mp.metalness);
                        lvrec(a, k, 3, slot) = f4(og.x, og.y, og.z, dVC);
                        lvrec(a, k, 4, slot) = f4(sd.mp.baseColor.x, sd.mp.baseColor.y, sd.mp.baseColor.z, sd.mp.baseColor.w);
                        lvrec(a, k, 5, slot) = f4(throughput.x, throughput.y, throughput.z, dVCM);

                        // ConnectToCamera, :908-966
                        const RtCamera& cam = pass.camera;
                        V4 dirToCamera = load4(cam.localToWorld + 12) - pos;
                        const float cameraDistanceSqr = sqrLength3(dirToCamera);
                        const float cameraDistance = sqrtf(cameraDistanceSqr);
                        dirToCamera = dirToCamera / cameraDistance;
                        float bsdfPdfW = 0.0f, bsdfRevPdfW = 0.0f;
                        const V4 cameraFactor = materialEvaluate<false>(mat, sd, neg(dirToCamera), bsdfPdfW, &bsdfRevPdfW);
                        float tmax = -1.0f; V4 contribution = zero4(); uint32_t target = 0xFFFFFFFFu;
                        V4 filmPos;
                        if (!almostZero4(cameraFactor) && cameraWorldToFilm(cam, pos, filmPos))
                        {
                            const V4 jitter = simd.getVector4();   // drawn when the connection is set up (the reference draws it inside Film::AccumulateColor,
                                                                    // i.e. only for visible connections; its stream is per-thread and entropy-seeded, so the position carries no meaning)
                            tmax = cameraDistance * 0.999f;
                            const float cosToCamera = dot3(dirToCamera, nr);
                            if (cosToCamera > FLT_EPSILON)
                            {
                                const float cameraPdfW = cameraDirectionPdfW(cam, neg(dirToCamera));
                                const float cameraPdfA = cameraPdfW * cosToCamera / cameraDistanceSqr;
                                const float wLight = cameraPdfA * (vcm.misVertexMergingWeightFactorVC + dVCM + dVC * bsdfRevPdfW);
                                const float misWeight = 1.0f / (wLight + 1.0f);
                                contribution = (cameraFactor * throughput) * (misWeight * cameraPdfA / (cosToCamera));
                                contribution = contribution * load4(vcm.cameraConnectingWeight);
                                uint32_t fx, fy;
                                if (filmSplatPixel(filmPos, pass.width, pass.height, jitter, fx, fy)) target = fy * pass.width + fx;
                            }
                        }
                        pshadow(lp, 0, 0, slot) = f4(dirToCamera.x, dirToCamera.y, dirToCamera.z, tmax);
                        pshadow(lp, 0, 1, slot) = f4(contribution.x, contribution.y, contribution.z, fbits(target));
                        prec(lp, R_SH_P, slot) = f4(pos.x, pos.y, pos.z, 0.0f);
                        pending = 1u;
                        needRay = tmax >= 0.0f;
                    }
                    if (vcm.useVertexMerging)
                    {
                        const uint32_t k = a.photonCount[slot];
                        a.photonCount[slot] = k + 1u;
                        float lum; uint32_t chroma;
                        packColorHdr(throughput, lum, chroma);
                        a.photonRaw[(size_t)(k * 2u + 0u) * a.capacity + slot] = f4(sd.intersection.frame.r[3].x, sd.intersection.frame.r[3].y, sd.intersection.frame.r[3].z, lum);
                        a.photonRaw[(size_t)(k * 2u + 1u) * a.capacity + slot] = f4(fbits(chroma), fbits(packUnitVector(sd.outgoingDirWorldSpace)), dVM, dVCM);
                    }
                }
                if (length + 2u <= vcm.maxPathLength)
                {
                    const V4 v = simd.getVector4();
                    const float sample[3] = { v.x, v.y, v.z };
                    alive = vcmAdvancePath(scene, vcm, lp, a, slot, sd, mat, sample, throughput, dVC, dVM, dVCM, length, isFiniteLight ? 0x200u : 0u);
                }
                storeSimd(simd, a, slot);
            }
            prec(lp, R_SAMPLER, slot).w = fbits(pending);
        }
        const uint32_t shadowAt = blockReserve(needRay ? 1u : 0u, shadowCount, &sCount, &sBase);
        if (needRay) shadowQueue[shadowAt] = slot;   // request id = light 0 * capacity + slot
        const uint32_t pathAt = blockReserve(alive ? 1u : 0u, countOut, &sCount, &sBase);
        if (alive) queueOut[pathAt] = slot;
    }
    flushCounters(cnt, counters);
}

// One vertex of LightTracer::RenderPixel's loop (Core/Rendering/LightTracer.cpp:69-181; renderer "Light Tracer"): like k_vcm_light_shade
// without MIS quantities, light vertices and photons; every vertex below maxRayDepth is connected to the camera with
// contribution = bsdf * throughput * PdfW / distance^2, and the shadow ray starts at samplePos + normal * 1e-4 (:138).
template <int kClass>
__global__ void RT_VCM_ATTR(k_lt_shade) k_lt_shade(const RtSceneDesc scene, const VcmBatch b, const Paths lp, const VcmArena a,
                                                       const uint32_t* __restrict__ queueIn, const uint32_t* __restrict__ countIn,
                                                       uint32_t* __restrict__ queueOut, uint32_t* __restrict__ countOut,
                                                       uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount,
                                                       float* __restrict__ sum, float* __restrict__ secondary, unsigned long long* counters)
{
    __shared__ uint32_t sCount, sBase;
    if (threadIdx.x == 0) sCount = 0u;
    __syncthreads();
    Counters cnt; zeroCounters(cnt);
    const DevPass& pass = b.passes[0];          // the Light Tracer runs one pass at a time
    const bool evenPass = (pass.passIndex % 2u) == 0u;
    const uint32_t maxRayDepth = b.vcms[0].maxPathLength;
    const uint32_t count = *countIn;
    const uint32_t rounded = (count + RT_BLOCK - 1u) / RT_BLOCK * RT_BLOCK;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < rounded; i += stride)
    {
        bool alive = false, needRay = false;
        uint32_t slot = 0;
        if (i < count)
        {
            slot = queueIn[i];
            const float4 rOrigin = prec(lp, R_ORIGIN, slot), rDir = prec(lp, R_DIR, slot), rTp = prec(lp, R_TP, slot);
            const float4 rHit = prec(lp, R_HIT, slot), rSampler = prec(lp, R_SAMPLER, slot);
            if (ubits(rSampler.w) != 0u) resolveSplat(lp, slot, sum, secondary, evenPass, cnt);
            uint32_t pending = 0u;
            const uint32_t depth = ubits(rOrigin.w) & 0xFFu;
            const Ray ray = makePathRay(rOrigin, rDir, depth);
            V4 throughput(rTp.x, rTp.y, rTp.z, rTp.w);
            Hit hit; hit.objectId = ubits(rHit.x); hit.subObjectId = ubits(rHit.y); hit.distance = rHit.z; hit.u = rHit.w; hit.v = rSampler.x;
            cnt.c[C_RAYS]++;
            if (hit.objectId != RT_INVALID_OBJECT && hit.subObjectId != RT_LIGHT_OBJECT)
            {
                ShadingData sd; sd.intersection.material = RT_NO_MATERIAL;
                if (hit.distance < FLT_MAX)
                {
                    sceneEvaluateIntersection<kClass>(scene, ray, hit, sd.intersection, cnt);
                    sd.outgoingDirWorldSpace = neg(ray.dir);
                    materialEvaluateShadingData<kClass>(scene, scene.materials[sd.intersection.material], sd);
                }
                if (depth < maxRayDepth)
                {
                    const RtMaterial& mat = scene.materials[sd.intersection.material];
                    RandomSimd simd; loadSimd(simd, a, slot);
                    {
                        const RtCamera& cam = pass.camera;
                        const V4 samplePos = sd.intersection.frame.r[3];
                        V4 dirToCamera = load4(cam.localToWorld + 12) - samplePos;
                        const float cameraDistanceSqr = sqrLength3(dirToCamera);
                        const float cameraDistance = sqrtf(cameraDistanceSqr);
                        dirToCamera = dirToCamera / cameraDistance;
                        float bsdfPdfW = 0.0f;
                        const V4 cameraFactor = materialEvaluate<false>(mat, sd, neg(dirToCamera), bsdfPdfW);
                        float tmax = -1.0f; V4 contribution = zero4(); uint32_t target = 0xFFFFFFFFu;
                        V4 filmPos;
                        if (!almostZero4(cameraFactor) && cameraWorldToFilm(cam, samplePos, filmPos))
                        {
                            const V4 jitter = simd.getVector4();
                            tmax = cameraDistance * 0.999f;
                            const float cameraPdfA = cameraDirectionPdfW(cam, neg(dirToCamera)) / cameraDistanceSqr;
                            contribution = (cameraFactor * throughput) * cameraPdfA;
                            uint32_t fx, fy;
                            if (filmSplatPixel(filmPos, pass.width, pass.height, jitter, fx, fy)) target = fy * pass.width + fx;
                        }
                        const V4 shadowOrigin = samplePos + sd.intersection.frame.r[2] * 0.0001f;
                        pshadow(lp, 0, 0, slot) = f4(dirToCamera.x, dirToCamera.y, dirToCamera.z, tmax);
                        pshadow(lp, 0, 1, slot) = f4(contribution.x, contribution.y, contribution.z, fbits(target));
                        prec(lp, R_SH_P, slot) = f4(shadowOrigin.x, shadowOrigin.y, shadowOrigin.z, 0.0f);
                        pending = 1u;
                        needRay = tmax >= 0.0f;
                    }
                    const V4 sv = simd.getVector4();
                    const float sample[3] = { sv.x, sv.y, sv.z };
                    V4 incomingDirWorldSpace = zero4(); float pdf = 0.0f; uint32_t event = EV_NULL;
                    const V4 bsdfValue = materialSample<false>(mat, sd, sample, incomingDirWorldSpace, pdf, event);
                    throughput = throughput * bsdfValue;
                    if (!almostZero4(throughput))
                    {
                        prec(lp, R_ORIGIN, slot) = f4(sd.intersection.frame.r[3].x, sd.intersection.frame.r[3].y, sd.intersection.frame.r[3].z, fbits(depth + 1u));
                        prec(lp, R_DIR, slot) = f4(incomingDirWorldSpace.x, incomingDirWorldSpace.y, incomingDirWorldSpace.z, 0.0f);
                        prec(lp, R_TP, slot) = f4(throughput.x, throughput.y, throughput.z, throughput.w);
                        alive = true;
                    }
                    storeSimd(simd, a, slot);
                }
            }
            prec(lp, R_SAMPLER, slot).w = fbits(pending);
        }
        const uint32_t shadowAt = blockReserve(needRay ? 1u : 0u, shadowCount, &sCount, &sBase);
        if (needRay) shadowQueue[shadowAt] = slot;
        const uint32_t pathAt = blockReserve(alive ? 1u : 0u, countOut, &sCount, &sBase);
        if (alive) queueOut[pathAt] = slot;
    }
    flushCounters(cnt, counters);
}

__global__ void __launch_bounds__(RT_BLOCK) k_vcm_light_finish(const VcmBatch b, const Paths lp, uint32_t numSlots, float* __restrict__ sum,
                                                               float* __restrict__ secondary, unsigned long long* counters)
{
    Counters cnt; zeroCounters(cnt);
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < numSlots; slot += stride)
        if (ubits(prec(lp, R_SAMPLER, slot).w) != 0u) resolveSplat(lp, slot, sum, secondary, (b.passes[slot / b.slotsPerPass].passIndex % 2u) == 0u, cnt);
    flushCounters(cnt, counters);
}

// ---- camera stage --------------------------------------------------------------------------------------------------------

// pending terms of the previous camera vertex: R_SAMPLER.w = numLightRequests | numConnectionRequests << 8 | merging << 16
RT_DEV void vcmResolvePending(const Paths& cp, const VcmArena& a, const VcmDev& vcm, uint32_t slot, uint32_t pendingBits, V4& resultColor, Counters& cnt)
{
    if (pendingBits == 0u) return;
    const uint32_t numLightRequests = pendingBits & 0xFFu, numConnections = (pendingBits >> 8) & 0xFFu;
    const float4 tp4 = prec(cp, R_SH_TP, slot);
    const V4 tp(tp4.x, tp4.y, tp4.z, 0.0f);
    if (pendingBits & 0x1000000u)   // SampleLights ran for this vertex, :719-731 and :254-259
    {
        // (the records of four requests are fetched together, unconditionally: the loop is bound by memory round trips, and a
        // fetch behind the visibility test would double them)
        V4 accumulatedColor = zero4();
        for (uint32_t base = 0; base < numLightRequests; base += 4u)
        {
            float tmax[4]; float4 c[4];
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k)
            {
                const uint32_t l = base + k < numLightRequests ? base + k : numLightRequests - 1u;
                tmax[k] = pshadow(cp, l, 0, slot).w; c[k] = pshadow(cp, l, 1, slot);
            }
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k)
            {
                if (base + k >= numLightRequests || tmax[k] < 0.0f) continue;
                cnt.c[C_SHADOW_HIT]++;
                accumulatedColor = accumulatedColor + V4(c[k].x, c[k].y, c[k].z, 0.0f);
            }
        }
        accumulatedColor = accumulatedColor * load4(vcm.lightSamplingWeight);
        resultColor = mulAdd(tp, accumulatedColor, resultColor);
    }
    if (pendingBits & 0x2000000u)   // vertex connections, :262-283
    {
        // (request j connects light vertex j: k_vcm_connect numbers them together, so the vertex's throughput record does not
        // have to wait for the request's)
        V4 vertexConnectionColor = zero4();
        for (uint32_t base = 0; base < numConnections; base += 4u)
        {
            float tmax[4]; float4 c[4], lvTp[4];
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k)
            {
                const uint32_t j = base + k < numConnections ? base + k : numConnections - 1u;
                tmax[k] = pshadow(cp, numLightRequests + j, 0, slot).w; c[k] = pshadow(cp, numLightRequests + j, 1, slot);
                lvTp[k] = lvrec(a, j, 5, slot);
            }
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k)
            {
                if (base + k >= numConnections || tmax[k] < 0.0f) continue;
                cnt.c[C_SHADOW_HIT]++;
                vertexConnectionColor = mulAdd(V4(lvTp[k].x, lvTp[k].y, lvTp[k].z, 0.0f), V4(c[k].x, c[k].y, c[k].z, 0.0f), vertexConnectionColor);
            }
        }
        vertexConnectionColor = vertexConnectionColor * load4(vcm.vertexConnectingWeight);
        resultColor = mulAdd(tp, vertexConnectionColor, resultColor);
    }
    if (pendingBits & 0x10000u)     // merging, :286-292
    {
        const float4 m = vrec(a, V_MERGE, slot);
        resultColor = mulAdd(V4(m.x, m.y, m.z, 0.0f), vcm.vertexMergingNormalizationFactor, resultColor);
    }
}

// EvaluateLight, :580-635 (isect == nullptr for global lights)
template <int kClass>
RT_DEV V4 vcmEvaluateLight(const RtSceneDesc& scene, const VcmDev& vcm, const RtLight& light, const float* invTransform, const Intersection* isect, const Ray& ray,
                           uint32_t length, bool lastSpecular, float dVC, float dVCM)
{
    const M4 worldToLight = loadM4(invTransform);
    const Ray lightSpaceRay = transformRayUnsafe(worldToLight, ray);
    const float cosAtLight = isect ? -dot3(isect->frame.r[2], ray.dir) : 1.0f;
    const V4 lightSpaceHitPoint = isect ? transformPoint(worldToLight, isect->frame.r[3]) : zero4();
    float directPdfA = 0.0f, emissionPdfW = 0.0f;
    V4 lightContribution = lightGetRadianceBidir<kClass>(scene, light, lightSpaceRay, lightSpaceHitPoint, cosAtLight, directPdfA, emissionPdfW);
    if (almostZero4(lightContribution)) return zero4();
    if (length > 1u)
    {
        const bool useVertexMerging = vcm.useVertexMerging && vcm.iteration > 0u;
        if (useVertexMerging && !vcm.useVertexConnection)
        {
            if (!lastSpecular) return zero4();
        }
        else
        {
            const float wCamera = directPdfA * dVCM + emissionPdfW * dVC;
            const float misWeight = 1.0f / (1.0f + wCamera);
            lightContribution = lightContribution * misWeight;
        }
    }
    lightContribution = lightContribution * load4(vcm.bsdfSamplingWeight);
    return lightContribution;
}

// One vertex of RenderPixel's loop, :201-314
template <int kClass>
__global__ void RT_VCM_ATTR(k_vcm_camera_shade) k_vcm_camera_shade(const RtSceneDesc scene, const VcmBatch b, const Paths cp, const VcmArena a,
                                                               const uint32_t* __restrict__ queueIn, const uint32_t* __restrict__ countIn,
                                                               uint32_t* __restrict__ queueOut, uint32_t* __restrict__ countOut,
                                                               uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount,
                                                               uint32_t* __restrict__ mergeQueue, uint32_t* __restrict__ mergeCount,
                                                               uint32_t* __restrict__ connectQueue, uint32_t* __restrict__ connectCount, unsigned long long* counters)
{
    __shared__ uint32_t sCount, sBase;
    if (threadIdx.x == 0) sCount = 0u;
    __syncthreads();
    Counters cnt; zeroCounters(cnt);
    const uint32_t count = *countIn;
    const uint32_t rounded = (count + RT_BLOCK - 1u) / RT_BLOCK * RT_BLOCK;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < rounded; i += stride)
    {
        bool alive = false, wantMerge = false, wantConnect = false;
        uint32_t slot = 0;
        unsigned long long rayMask = 0ull;   // requests of this vertex that need a shadow ray (at most 64 per vertex, checked by the host)
        if (i < count)
        {
            slot = queueIn[i];
            const uint32_t pb = slot / b.slotsPerPass;
            const DevPass& pass = b.passes[pb];
            const VcmDev& vcm = b.vcms[pb];
            const uint32_t gridPhotons = b.grids[pb].numPhotons;
            const float4 rOrigin = prec(cp, R_ORIGIN, slot), rDir = prec(cp, R_DIR, slot), rTp = prec(cp, R_TP, slot);
            const float4 rResult = prec(cp, R_RESULT, slot), rHit = prec(cp, R_HIT, slot), rSampler = prec(cp, R_SAMPLER, slot);
            const uint32_t flags = ubits(rOrigin.w);
            const uint32_t depth = flags & 0xFFu, length = depth + 1u;
            const bool lastSpecular = (flags & 0x100u) != 0u;
            const uint32_t pix = ubits(rResult.w);
            const Ray ray = makePathRay(rOrigin, rDir, depth);
            V4 throughput(rTp.x, rTp.y, rTp.z, rTp.w);
            V4 resultColor(rResult.x, rResult.y, rResult.z, 0.0f);
            vcmResolvePending(cp, a, vcm, slot, ubits(rSampler.w), resultColor, cnt);
            float dVC, dVM, dVCM;
            if (depth == 0u)   // :186-193
            {
                const float cameraPdf = cameraDirectionPdfW(pass.camera, ray.dir);
                dVC = 0.0f; dVM = 0.0f; dVCM = 1.0f / cameraPdf;
            }
            else { const float4 rMis = vrec(a, V_MIS, slot); dVC = rMis.x; dVM = rMis.y; dVCM = rMis.z; }
            Hit hit; hit.objectId = ubits(rHit.x); hit.subObjectId = ubits(rHit.y); hit.distance = rHit.z; hit.u = rHit.w; hit.v = rSampler.x;
            cnt.c[C_RAYS]++;
            uint32_t pendingBits = 0u;
            bool samplerStored = false;
            do
            {
                if (hit.objectId == RT_INVALID_OBJECT)
                {
                    V4 result = zero4();   // EvaluateGlobalLights, :733-744
                    for (uint32_t g = 0; g < scene.numGlobalLights; ++g)
                    {
                        const RtLight& light = scene.lights[scene.globalLights[g]];
                        result = result + vcmEvaluateLight<kClass>(scene, vcm, light, light.invTransform, nullptr, ray, length, lastSpecular, dVC, dVCM);
                    }
                    resultColor = mulAdd(throughput, result, resultColor);
                    break;
                }
                ShadingData sd;
                sd.intersection.material = (flags >> 9) - 1u;   // the path's one ShadingData keeps the previous vertex's material (see k_shade)
                sceneEvaluateIntersection<kClass>(scene, ray, hit, sd.intersection, cnt);
                {
                    const float cosTheta = dot3(ray.dir, sd.intersection.frame.r[2]);
                    const float invMis = 1.0f / Abs(cosTheta);
                    dVCM *= Sqr(hit.distance);
                    dVCM *= invMis; dVC *= invMis; dVM *= invMis;
                }
                if (hit.subObjectId == RT_LIGHT_OBJECT)
                {
                    const RtObject& obj = scene.objects[hit.objectId];
                    const V4 lightColor = vcmEvaluateLight<kClass>(scene, vcm, scene.lights[obj.lightIndex], obj.invTransform, &sd.intersection, ray, length, lastSpecular, dVC, dVCM);
                    resultColor = mulAdd(throughput, lightColor, resultColor);
                    break;
                }
                sd.outgoingDirWorldSpace = neg(ray.dir);
                const RtMaterial& mat = scene.materials[sd.intersection.material];
                materialEvaluateShadingData<kClass>(scene, mat, sd);
                resultColor = mulAdd(throughput, sd.mp.emission, resultColor);
                if (length >= vcm.maxPathLength) break;
                const bool isDeltaBsdf = bsdfIsDelta(mat.bsdf);
                const V4 pos = sd.intersection.frame.r[3];

                Sampler sampler; loadSampler(sampler, cp, slot, pix, rSampler, pass, scene.blueNoise);

                // SampleLights / SampleLight, :637-731
                uint32_t numLightRequests = 0u;
                if (!isDeltaBsdf && vcm.useVertexConnection)
                {
                    for (uint32_t l = 0; l < scene.numLights; ++l)
                    {
                        const RtLight& light = scene.lights[l];
                        float u[3]; u[0] = sampler.getFloat(); u[1] = sampler.getFloat(); u[2] = sampler.getFloat();
                        float tmax = -1.0f; V4 dir = zero4(); V4 contribution = zero4();
                        IlluminateResult ir; float emissionPdfW;
                        const V4 radiance = lightIlluminateBidir<kClass>(scene, light, sd.intersection, u, ir, emissionPdfW);
                        if (!almostZero4(radiance))
                        {
                            float bsdfPdfW = 0.0f, bsdfRevPdfW = 0.0f;
                            const V4 bsdfFactor = materialEvaluate<false>(mat, sd, neg(ir.directionToLight), bsdfPdfW, &bsdfRevPdfW);
                            if (!almostZero4(bsdfFactor))
                            {
                                dir = ir.directionToLight; tmax = ir.distance * 0.999f;
                                const float lightPickProbability = 1.0f;
                                const bool isDeltaLight = (light.flags & RT_LIGHT_FLAG_DELTA) != 0u;
                                const float continuationProbability = 1.0f;
                                bsdfPdfW *= isDeltaLight ? 0.0f : continuationProbability;
                                bsdfRevPdfW *= continuationProbability;
                                const float cosToLight = dot3(sd.intersection.frame.r[2], ir.directionToLight);
                                if (cosToLight > FLT_EPSILON)
                                {
                                    const float wLight = bsdfPdfW / (lightPickProbability * ir.directPdfW);
                                    const float wCamera = (emissionPdfW * cosToLight / (ir.directPdfW * ir.cosAtLight)) * (vcm.misVertexMergingWeightFactorVC + dVCM + dVC * bsdfRevPdfW);
                                    const float misWeight = 1.0f / (wLight + 1.0f + wCamera);
                                    contribution = (radiance * bsdfFactor) * (misWeight / (lightPickProbability * ir.directPdfW));
                                }
                            }
                        }
                        pshadow(cp, l, 0, slot) = f4(dir.x, dir.y, dir.z, tmax);
                        pshadow(cp, l, 1, slot) = f4(contribution.x, contribution.y, contribution.z, 0.0f);
                        if (tmax >= 0.0f) rayMask |= 1ull << l;
                    }
                    numLightRequests = scene.numLights;
                    pendingBits |= 0x1000000u;
                }

                // ConnectVertices to the light vertices of this pixel, :262-283 and :746-821: k_vcm_connect, from the vertex record below
                // (two BSDF evaluations per light vertex: kept out of this kernel's register budget)
                wantConnect = !isDeltaBsdf && vcm.useVertexConnection && a.lvCount[slot] > 0u;

                // MergeVertices, :823-906: the range query runs in k_vcm_merge (wave-cooperative for long photon lists); this
                // vertex is handed over as a record.  With no photons HashGrid::Process returns at once and the term is +0.
                const bool mergeHere = !isDeltaBsdf && vcm.useVertexMerging && vcm.iteration > 0u && gridPhotons != 0u;
                if (mergeHere || wantConnect)
                {
                    const V4 tg = sd.intersection.frame.r[0], nr = sd.intersection.frame.r[2], og = sd.outgoingDirWorldSpace;
                    cvrec(a, 0, slot) = f4(pos.x, pos.y, pos.z, fbits(sd.intersection.material));
                    cvrec(a, 1, slot) = f4(tg.x, tg.y, tg.z, sd.mp.roughness);
                    cvrec(a, 2, slot) = f4(nr.x, nr.y, nr.z, sd.mp.metalness);
                    cvrec(a, 3, slot) = f4(og.x, og.y, og.z, dVM);
                    cvrec(a, 4, slot) = f4(sd.mp.baseColor.x, sd.mp.baseColor.y, sd.mp.baseColor.z, sd.mp.baseColor.w);
                    cvrec(a, 5, slot) = f4(throughput.x, throughput.y, throughput.z, dVCM);
                }
                if (mergeHere) { wantMerge = true; pendingBits |= 0x10000u; }
                pendingBits |= numLightRequests;
                if (pendingBits != 0u || wantConnect)
                {
                    prec(cp, R_SH_P, slot) = f4(pos.x, pos.y, pos.z, dVC);                                         // .w: for k_vcm_connect
                    prec(cp, R_SH_TP, slot) = f4(throughput.x, throughput.y, throughput.z, fbits(length));
                }

                // AdvancePath, :493-578 (length > maxPathLength cannot hold here, :308-311)
                {
                    float sample[3]; sample[0] = sampler.getFloat(); sample[1] = sampler.getFloat(); sample[2] = sampler.getFloat();
                    alive = vcmAdvancePath(scene, vcm, cp, a, slot, sd, mat, sample, throughput, dVC, dVM, dVCM, length, 0u);
                }
                storeSampler(sampler, cp, slot, hit.v, pendingBits);
                samplerStored = true;
            } while (false);
            if (!samplerStored) prec(cp, R_SAMPLER, slot).w = fbits(pendingBits);
            prec(cp, R_RESULT, slot) = f4(resultColor.x, resultColor.y, resultColor.z, rResult.w);
        }
        // the shadow requests of this vertex, then the surviving path
        uint32_t shadowAt = blockReserve((uint32_t)__popcll(rayMask), shadowCount, &sCount, &sBase);
        for (unsigned long long pendingMask = rayMask; pendingMask != 0ull; pendingMask &= pendingMask - 1ull)
            shadowQueue[shadowAt++] = (uint32_t)(__ffsll((long long)pendingMask) - 1) * cp.capacity + slot;
        const uint32_t pathAt = blockReserve(alive ? 1u : 0u, countOut, &sCount, &sBase);
        if (alive) queueOut[pathAt] = slot;
        const uint32_t mergeAt = blockReserve(wantMerge ? 1u : 0u, mergeCount, &sCount, &sBase);
        if (wantMerge) mergeQueue[mergeAt] = slot;
        const uint32_t connectAt = blockReserve(wantConnect ? 1u : 0u, connectCount, &sCount, &sBase);
        if (wantConnect) connectQueue[connectAt] = slot;
    }
    flushCounters(cnt, counters);
}

// MergeVertices, :823-906 + HashGrid::Process, HashGrid.h:73-143, for the camera vertices queued by k_vcm_camera_shade.
// The reference adds the photons' terms with one fma chain in grid order; the chain is kept, the work in front of it is spread:
// a query with few candidate photons is one lane's loop; a query with many (a caustic, the footprint of a spot light:
// thousands of photons per cell) is taken by the whole wave -- 64 photons are tested and their BSDF terms evaluated at a time,
// then every lane replays the fma chain over the contributing lanes in order (shuffles), so the sum is the sequential one.
#define RT_VCM_SCAN_UNROLL 4u
struct MergeRanges { uint32_t start[8], end[8]; uint32_t numCells, total; };
RT_DEV void mergeCellRanges(const HashGridView& g, V4 queryPos, MergeRanges& r)
{
    const V4 distMin = queryPos - V4(g.boxMin[0], g.boxMin[1], g.boxMin[2], 0.0f);
    const V4 cellCoords = mulSub(distMin, splat(g.invCellSize), splat(0.5f));
    const int32_t cx = cvtT(cellCoords.x), cy = cvtT(cellCoords.y), cz = cvtT(cellCoords.z);
    uint32_t visitedCells[8];
    r.numCells = 0; r.total = 0;
#pragma unroll
    for (uint32_t i = 0; i < 8; ++i)
    {
        const uint32_t x = (uint32_t)cx + (i & 1), y = (uint32_t)cy + ((i >> 1) & 1), z = (uint32_t)cz + (i >> 2);
        const uint32_t ci = hashCellIndex(x, y, z, g.hashTableMask);
        bool visited = false;
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) if (j < r.numCells && visitedCells[j] == ci) visited = true;
        if (!visited)
        {
            visitedCells[r.numCells] = ci;
            r.start[r.numCells] = ci == 0 ? 0 : g.cellEnds[ci - 1]; r.end[r.numCells] = g.cellEnds[ci];
            r.total += r.end[r.numCells] - r.start[r.numCells];
            r.numCells++;
        }
    }
}
RT_DEV bool photonInRadius(const HashGridView& g, uint32_t j, V4 pos)   // HashGrid::Process's test, HashGrid.h:131-137
{
    const Photon& photon = g.photons[j];
    const float distSqr = sqrLength3(pos - V4(photon.px, photon.py, photon.pz, 0.0f));
    return distSqr <= g.radiusSqr;
}
// one photon against one camera vertex: false = no contribution; else term = cameraBsdfFactor * photon throughput, weight = misWeight / cosToLight
RT_DEV bool mergePhoton(const RtSceneDesc& scene, const VcmDev& vcm, const HashGridView& g, uint32_t j, const ShadingData& sd, V4 pos, float dVCM, float dVM, V4& term, float& weight)
{
    const Photon& photon = g.photons[j];
    if (!photonInRadius(g, j, pos)) return false;
    const V4 lightDirection = unpackUnitVector(photon.direction);
    const float cosToLight = dot3(sd.intersection.frame.r[2], lightDirection);
    if (cosToLight < FLT_EPSILON) return false;
    float cameraBsdfDirPdfW = 0.0f, cameraBsdfRevPdfW = 0.0f;
    const V4 cameraBsdfFactor = materialEvaluate<false>(scene.materials[sd.intersection.material], sd, neg(lightDirection), cameraBsdfDirPdfW, &cameraBsdfRevPdfW);
    if (almostZero4(cameraBsdfFactor)) return false;
    const V4 photonThroughput = unpackColorHdr(photon.lum, photon.chroma);
    const float wLight = photon.dVCM * vcm.misVertexConnectionWeightFactorVM + photon.dVM * cameraBsdfDirPdfW;
    const float wCamera = dVCM * vcm.misVertexConnectionWeightFactorVM + dVM * cameraBsdfRevPdfW;
    const float misWeight = 1.0f / (wLight + 1.0f + wCamera);
    weight = misWeight / cosToLight;
    term = cameraBsdfFactor * photonThroughput;
    return true;
}
RT_DEV float laneValue(float v, int lane) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), lane)); }   // lane: wave-uniform
RT_DEV V4 shfl3(V4 v, int lane) { return V4(__shfl(v.x, lane), __shfl(v.y, lane), __shfl(v.z, lane), 0.0f); }
RT_DEV void loadCameraVertex(const RtSceneDesc& scene, const VcmArena& a, uint32_t slot, ShadingData& sd, V4& throughput, float& dVM, float& dVCM)
{
    const float4 r0 = cvrec(a, 0, slot), r1 = cvrec(a, 1, slot), r2 = cvrec(a, 2, slot), r3 = cvrec(a, 3, slot), r4 = cvrec(a, 4, slot), r5 = cvrec(a, 5, slot);
    sd.intersection.frame.r[0] = V4(r1.x, r1.y, r1.z, 0.0f);
    sd.intersection.frame.r[2] = V4(r2.x, r2.y, r2.z, 0.0f);
    sd.intersection.frame.r[1] = cross3(sd.intersection.frame.r[0], sd.intersection.frame.r[2]);
    sd.intersection.frame.r[3] = V4(r0.x, r0.y, r0.z, 0.0f);
    sd.intersection.texCoord = zero4();
    sd.intersection.material = ubits(r0.w);
    sd.outgoingDirWorldSpace = V4(r3.x, r3.y, r3.z, 0.0f);
    sd.mp.baseColor = V4(r4.x, r4.y, r4.z, r4.w); sd.mp.emission = zero4();
    sd.mp.roughness = r1.w; sd.mp.metalness = r2.w; sd.mp.IoR = scene.materials[sd.intersection.material].IoR;
    throughput = V4(r5.x, r5.y, r5.z, 0.0f);
    dVM = r3.w; dVCM = r5.w;
}
// ConnectVertices, :746-821, for the camera vertices k_vcm_camera_shade queued: every light vertex of the pixel's light path is one
// request (direction | tmax, contribution | vertex) after the vertex's next-event requests; the visibility rays ride in the next k_trace
// launch and vcmResolvePending folds the visible ones in.  Reads the 96-byte vertex record the merge kernel uses, plus dVC and the path
// length from the spare lanes of R_SH_P / R_SH_TP.
__global__ void RT_VCM_CONNECT_ATTR k_vcm_connect(const RtSceneDesc scene, const VcmBatch b, const Paths cp, const VcmArena a,
                                                          const uint32_t* __restrict__ queue, const uint32_t* __restrict__ queueCount,
                                                          uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount)
{
    __shared__ uint32_t sCount, sBase;
    if (threadIdx.x == 0) sCount = 0u;
    __syncthreads();
    const uint32_t count = *queueCount;
    const uint32_t rounded = (count + RT_BLOCK - 1u) / RT_BLOCK * RT_BLOCK;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < rounded; i += stride)
    {
        uint32_t slot = 0;
        unsigned long long rayMask = 0ull;
        if (i < count)
        {
            slot = queue[i];
            const VcmDev& vcm = b.vcms[slot / b.slotsPerPass];
            ShadingData sd; V4 throughput; float dVM, dVCM;
            loadCameraVertex(scene, a, slot, sd, throughput, dVM, dVCM);
            const float dVC = prec(cp, R_SH_P, slot).w;
            const uint32_t length = ubits(prec(cp, R_SH_TP, slot).w);
            uint32_t pendingBits = ubits(prec(cp, R_SAMPLER, slot).w);
            const uint32_t numLightRequests = pendingBits & 0xFFu;
            const RtMaterial& mat = scene.materials[sd.intersection.material];
            const V4 pos = sd.intersection.frame.r[3];
            uint32_t numConnections = 0u;
            const uint32_t numLightVertices = a.lvCount[slot];
            LightVertexRecords next = fetchLightVertex(a, 0u, slot);   // (numLightVertices >= 1 for a queued vertex)
            for (uint32_t v = 0; v < numLightVertices; ++v)
            {
                ShadingData lsd; V4 lvThroughput; float lvVC, lvVCM; uint32_t lvLength;
                const LightVertexRecords cur = next;
                next = fetchLightVertex(a, v + 1u < numLightVertices ? v + 1u : v, slot);   // in flight during this vertex's two BSDF evaluations
                decodeLightVertex(scene, cur, lsd, lvThroughput, lvVC, lvVCM, lvLength);
                if (lvLength + length + 1u > vcm.maxPathLength) break;
                V4 lightDir = lsd.intersection.frame.r[3] - pos;
                const float distanceSqr = sqrLength3(lightDir);
                const float distance = sqrtf(distanceSqr);
                lightDir = lightDir / distance;
                const float cosCameraVertex = dot3(sd.intersection.frame.r[2], lightDir);
                const float cosLightVertex = dot3(lsd.intersection.frame.r[2], neg(lightDir));
                float tmax = -1.0f; V4 contribution = zero4();
                if (!(cosCameraVertex <= 0.0f || cosLightVertex <= 0.0f))
                {
                    const float geometryTerm = 1.0f / distanceSqr;
                    float cameraBsdfPdfW = 0.0f, cameraBsdfRevPdfW = 0.0f;
                    const V4 cameraFactor = materialEvaluate<false>(mat, sd, neg(lightDir), cameraBsdfPdfW, &cameraBsdfRevPdfW);
                    if (!almostZero4(cameraFactor))
                    {
                        float lightBsdfPdfW = 0.0f, lightBsdfRevPdfW = 0.0f;
                        const V4 lightFactor = materialEvaluate<false>(scene.materials[lsd.intersection.material], lsd, lightDir, lightBsdfPdfW, &lightBsdfRevPdfW);
                        if (!almostZero4(lightFactor))
                        {
                            tmax = distance * 0.999f;
                            const float continuationProbability = 1.0f;
                            lightBsdfPdfW *= continuationProbability;
                            lightBsdfRevPdfW *= continuationProbability;
                            const float cameraBsdfPdfA = vcmPdfWtoA(cameraBsdfPdfW, distance, cosLightVertex);
                            const float lightBsdfPdfA = vcmPdfWtoA(lightBsdfPdfW, distance, cosCameraVertex);
                            const float wLight = cameraBsdfPdfA * (vcm.misVertexMergingWeightFactorVC + lvVCM + lvVC * lightBsdfRevPdfW);
                            const float wCamera = lightBsdfPdfA * (vcm.misVertexMergingWeightFactorVC + dVCM + dVC * cameraBsdfRevPdfW);
                            const float misWeight = 1.0f / (wLight + 1.0f + wCamera);
                            contribution = (cameraFactor * lightFactor) * (geometryTerm * misWeight);
                        }
                    }
                }
                const uint32_t r = numLightRequests + numConnections;
                pshadow(cp, r, 0, slot) = f4(lightDir.x, lightDir.y, lightDir.z, tmax);
                pshadow(cp, r, 1, slot) = f4(contribution.x, contribution.y, contribution.z, fbits(v));
                if (tmax >= 0.0f) rayMask |= 1ull << r;
                numConnections++;
            }
            pendingBits |= (numConnections << 8) | 0x2000000u;
            prec(cp, R_SAMPLER, slot).w = fbits(pendingBits);
        }
        uint32_t shadowAt = blockReserve((uint32_t)__popcll(rayMask), shadowCount, &sCount, &sBase);
        for (unsigned long long pendingMask = rayMask; pendingMask != 0ull; pendingMask &= pendingMask - 1ull)
            shadowQueue[shadowAt++] = (uint32_t)(__ffsll((long long)pendingMask) - 1) * cp.capacity + slot;
    }
}

__global__ void __launch_bounds__(RT_BLOCK) k_vcm_merge(const RtSceneDesc scene, const VcmBatch b, const VcmArena a,
                                                        const uint32_t* __restrict__ queue, const uint32_t* __restrict__ queueCount, uint32_t cooperativeMin)
{
    __shared__ uint32_t sStart[RT_BLOCK][8], sEnd[RT_BLOCK][8];
    __shared__ uint32_t sCand[RT_BLOCK / 64][128], sOwner[RT_BLOCK / 64][128];
    const uint32_t count = *queueCount;
    const uint32_t rounded = (count + 63u) & ~63u;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t lane = threadIdx.x & 63u, waveBase = threadIdx.x & ~63u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < rounded; i += stride)
    {
        const bool valid = i < count;
        uint32_t slot = 0;
        ShadingData sd; V4 throughput = zero4(); float dVM = 0.0f, dVCM = 0.0f;
        MergeRanges r; r.numCells = 0; r.total = 0;
        V4 contribution = zero4();
        uint32_t pb = 0;
        if (valid)
        {
            slot = queue[i];
            pb = slot / b.slotsPerPass;
            loadCameraVertex(scene, a, slot, sd, throughput, dVM, dVCM);
            mergeCellRanges(b.grids[pb], sd.intersection.frame.r[3], r);
        }
        const bool big = valid && r.total >= cooperativeMin;
        // (a wave only reads what its own lanes wrote: no block barrier needed, the LDS writes are ordered before the reads of the same wave)
#pragma unroll
        for (int c = 0; c < 8; ++c) { sStart[threadIdx.x][c] = c < (int)r.numCells ? r.start[c] : 0u; sEnd[threadIdx.x][c] = c < (int)r.numCells ? r.end[c] : 0u; }
        uint32_t* cand = sCand[threadIdx.x >> 6];
        uint32_t* owners = sOwner[threadIdx.x >> 6];
        const unsigned long long lanesBelow = (1ull << lane) - 1ull;
        // SHORT photon lists: every lane scans its own candidates with the distance test only (one in five passes on a surface); the
        // photons inside the radius go, in scan order, to a wave-wide list of (owner lane, photon).  64 entries at a time are then
        // evaluated one per lane -- the owner's vertex comes through the crossbar -- and each owner adds its entries' terms in list
        // order, which is its own scan order: the reference's fma chain.  (A lane's own loop had 16 % of the lanes busy: different
        // list lengths, and a BSDF evaluation behind a branch one lane in five takes.)
        if (__ballot(valid && !big && r.total != 0u) != 0ull)
        {
            uint32_t numList = 0u;   // wave-uniform
            auto flush = [&](uint32_t n)
            {
                __builtin_amdgcn_wave_barrier();
                const int o = lane < n ? (int)owners[lane] : (int)lane;
                const uint32_t pj = lane < n ? cand[lane] : 0u;
                ShadingData osd;
                osd.intersection.frame.r[0] = shfl3(sd.intersection.frame.r[0], o);
                osd.intersection.frame.r[2] = shfl3(sd.intersection.frame.r[2], o);
                osd.intersection.frame.r[1] = cross3(osd.intersection.frame.r[0], osd.intersection.frame.r[2]);
                osd.intersection.frame.r[3] = shfl3(sd.intersection.frame.r[3], o);
                osd.intersection.texCoord = zero4();
                osd.intersection.material = (uint32_t)__shfl((int)sd.intersection.material, o);
                osd.outgoingDirWorldSpace = shfl3(sd.outgoingDirWorldSpace, o);
                osd.mp.baseColor = shfl3(sd.mp.baseColor, o); osd.mp.baseColor.w = __shfl(sd.mp.baseColor.w, o);
                osd.mp.emission = zero4();
                osd.mp.roughness = __shfl(sd.mp.roughness, o); osd.mp.metalness = __shfl(sd.mp.metalness, o); osd.mp.IoR = __shfl(sd.mp.IoR, o);
                const float oVM = __shfl(dVM, o), oVCM = __shfl(dVCM, o);
                const uint32_t opb = (uint32_t)__shfl((int)pb, o);
                V4 term = zero4(); float weight = 0.0f;
                const bool contributes = lane < n && mergePhoton(scene, b.vcms[opb], b.grids[opb], pj, osd, osd.intersection.frame.r[3], oVCM, oVM, term, weight);
                for (unsigned long long m = __ballot(contributes); m != 0ull; m &= m - 1ull)
                {
                    const int e = __ffsll((long long)m) - 1;
                    const int owner = __builtin_amdgcn_readlane(o, e);
                    const V4 t(laneValue(term.x, e), laneValue(term.y, e), laneValue(term.z, e), laneValue(term.w, e));
                    const float w = laneValue(weight, e);
                    if ((int)lane == owner) contribution = mulAdd(t, w, contribution);
                }
                __builtin_amdgcn_wave_barrier();
            };
            bool scanning = valid && !big && r.total != 0u;
            uint32_t c = 0u, j = sStart[threadIdx.x][0], endJ = sEnd[threadIdx.x][0];
            auto skipEmptyCells = [&]()
            {
                while (scanning && j >= endJ)
                {
                    if (++c >= r.numCells) scanning = false;
                    else { j = sStart[threadIdx.x][c]; endJ = sEnd[threadIdx.x][c]; }
                }
            };
            skipEmptyCells();
            while (__ballot(scanning) != 0ull)
            {
                // four consecutive photons of the cell per step: the loads are independent (the scan is bound by their latency)
                bool inside[RT_VCM_SCAN_UNROLL]; uint32_t myJ = 0u;
#pragma unroll
                for (uint32_t k = 0; k < RT_VCM_SCAN_UNROLL; ++k) inside[k] = false;
                if (scanning)
                {
                    const HashGridView& grid = b.grids[pb];
                    const uint32_t n = endJ - j < RT_VCM_SCAN_UNROLL ? endJ - j : RT_VCM_SCAN_UNROLL;
                    myJ = j;
#pragma unroll
                    for (uint32_t k = 0; k < RT_VCM_SCAN_UNROLL; ++k) inside[k] = photonInRadius(grid, j + (k < n ? k : n - 1u), sd.intersection.frame.r[3]) && k < n;
                    j += n;
                    skipEmptyCells();
                }
#pragma unroll
                for (uint32_t k = 0; k < RT_VCM_SCAN_UNROLL; ++k)
                {
                    const unsigned long long m = __ballot(inside[k]);
                    if (m == 0ull) continue;
                    if (inside[k]) { const uint32_t at = numList + (uint32_t)__popcll(m & lanesBelow); cand[at] = myJ + k; owners[at] = lane; }
                    numList += (uint32_t)__popcll(m);
                    if (numList >= 64u)
                    {
                        flush(64u);
                        const uint32_t restJ = lane < numList - 64u ? cand[64u + lane] : 0u, restOwner = lane < numList - 64u ? owners[64u + lane] : 0u;
                        __builtin_amdgcn_wave_barrier();
                        cand[lane] = restJ; owners[lane] = restOwner;
                        numList -= 64u;
                    }
                }
            }
            if (numList != 0u) flush(numList);
        }
        unsigned long long mBig = __ballot(big);
        if (mBig != 0ull)
        {
            for (; mBig != 0ull; mBig &= mBig - 1ull)
            {
                const int q = __ffsll((long long)mBig) - 1;
                const uint32_t qSlot = (uint32_t)__shfl((int)slot, q);
                const uint32_t qpb = qSlot / b.slotsPerPass;
                const VcmDev& vcm = b.vcms[qpb];
                const HashGridView& grid = b.grids[qpb];
                ShadingData qsd; V4 qThroughput; float qVM, qVCM;
                loadCameraVertex(scene, a, qSlot, qsd, qThroughput, qVM, qVCM);   // same address in all lanes: one broadcast fetch
                const V4 qPos = qsd.intersection.frame.r[3];
                V4 acc = zero4();
                // The photons inside the radius (one in five of a cell block's on a surface) are compacted, in order, into a wave-wide
                // list; 64 of them are evaluated at a time, then every lane replays the fma chain over the contributing lanes in order.
                uint32_t numCand = 0u;   // wave-uniform
                auto evaluate = [&](uint32_t n)
                {
                    __builtin_amdgcn_wave_barrier();
                    V4 term = zero4(); float weight = 0.0f;
                    const bool contributes = lane < n && mergePhoton(scene, vcm, grid, cand[lane], qsd, qPos, qVCM, qVM, term, weight);
                    for (unsigned long long m = __ballot(contributes); m != 0ull; m &= m - 1ull)
                    {
                        const int src = __ffsll((long long)m) - 1;
                        // (src is wave-uniform: v_readlane broadcasts through an SGPR instead of a trip through the LDS crossbar)
                        const V4 t(laneValue(term.x, src), laneValue(term.y, src), laneValue(term.z, src), laneValue(term.w, src));
                        acc = mulAdd(t, laneValue(weight, src), acc);
                    }
                };
                for (uint32_t c = 0; c < 8u; ++c)
                {
                    const uint32_t start = sStart[waveBase + (uint32_t)q][c], end = sEnd[waveBase + (uint32_t)q][c];
                    for (uint32_t base = start; base < end; base += 64u)
                    {
                        const uint32_t j = base + lane;
                        const bool inside = j < end && photonInRadius(grid, j, qPos);
                        const unsigned long long m = __ballot(inside);
                        if (inside) cand[numCand + (uint32_t)__popcll(m & lanesBelow)] = j;
                        numCand += (uint32_t)__popcll(m);
                        if (numCand >= 64u)
                        {
                            evaluate(64u);
                            __builtin_amdgcn_wave_barrier();
                            const uint32_t rest = lane < numCand - 64u ? cand[64u + lane] : 0u;
                            __builtin_amdgcn_wave_barrier();
                            cand[lane] = rest;
                            numCand -= 64u;
                        }
                    }
                }
                if (numCand != 0u) evaluate(numCand);
                __builtin_amdgcn_wave_barrier();
                if ((int)lane == q) contribution = acc;
            }
        }
        if (valid)
        {
            const V4 vertexMergingColor = contribution * load4(b.vcms[pb].vertexMergingWeight);
            const V4 m = throughput * vertexMergingColor;
            vrec(a, V_MERGE, slot) = f4(m.x, m.y, m.z, 0.0f);
        }
    }
}

// the last vertex's pending terms, then Film::AccumulateColor(x, y, color) (Film.cpp:25-39)
__global__ void __launch_bounds__(RT_BLOCK) k_vcm_camera_finish(const VcmBatch b, uint32_t numPasses, const Paths cp, const VcmArena a,
                                                                float* __restrict__ sum, float* __restrict__ secondary, uint32_t width, unsigned long long* counters)
{
    Counters cnt; zeroCounters(cnt);
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t pixelSlot = blockIdx.x * blockDim.x + threadIdx.x; pixelSlot < b.slotsPerPass; pixelSlot += stride)
    {
        const uint32_t pix = ubits(prec(cp, R_RESULT, pixelSlot).w);
        const size_t idx = 3 * ((size_t)(pix >> 16) * width + (pix & 0xFFFFu));
        float sr = sum[idx + 0], sg = sum[idx + 1], sb = sum[idx + 2];
        float tr = secondary[idx + 0], tg = secondary[idx + 1], tb = secondary[idx + 2];
        for (uint32_t p = 0; p < numPasses; ++p)   // the passes of the batch in pass order, like k_accumulate
        {
            const uint32_t slot = p * b.slotsPerPass + pixelSlot;
            const float4 rResult = prec(cp, R_RESULT, slot);
            V4 resultColor(rResult.x, rResult.y, rResult.z, 0.0f);
            vcmResolvePending(cp, a, b.vcms[p], slot, ubits(prec(cp, R_SAMPLER, slot).w), resultColor, cnt);
            sr = sr + resultColor.x; sg = sg + resultColor.y; sb = sb + resultColor.z;
            if ((b.passes[p].passIndex % 2u) == 0u) { tr = tr + resultColor.x; tg = tg + resultColor.y; tb = tb + resultColor.z; }
        }
        sum[idx + 0] = sr; sum[idx + 1] = sg; sum[idx + 2] = sb;
        secondary[idx + 0] = tr; secondary[idx + 1] = tg; secondary[idx + 2] = tb;
    }
    flushCounters(cnt, counters);
}
