// rt_kat.inl -- known-answer-test hook of the C-ABI (rtgpu_kat, rtgpu_kat_sampler, rtgpu_kat_mesh): the DEVICE functions of
// rt_device_math.h / rt_device_core.h / rt_device_vcm.h / rt_device_traverse.h evaluated on the record layouts of tests/golden/*.kat
// (written by the reference's own translation units, see tests/golden/README.md), so that every SURVEY 8(a) row has a
// device-vs-reference-vector check that does not go through the CPU restatement.  One thread per record; included by rt_trace.hip; the function ids are in rt_trace_kernels.h.
//
// Function ids and record layouts: tests/golden/README.md (the ids the fixtures carry in their headers).

RT_DEV void katPut(float* o, V4 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
RT_DEV void katLoad(void* dst, const float* src, uint32_t bytes)
{
    uint32_t* d = static_cast<uint32_t*>(dst);
    for (uint32_t k = 0; k < bytes / 4u; ++k) d[k] = __float_as_uint(src[k]);
}
// the BSDF fixtures carry the scalar part of RtMaterial (through `bsdf`, 52 bytes) in 16 float slots
RT_DEV RtMaterial katMaterial(const float* in)
{
    RtMaterial m; memset(&m, 0, sizeof(m));
    katLoad(&m, in, 52);
    m.baseColorTexture = m.emissionTexture = m.roughnessTexture = m.metalnessTexture = m.normalMapTexture = RT_NO_TEXTURE;
    return m;
}

__global__ void __launch_bounds__(64) k_kat(const RtSceneDesc scene, uint32_t func, const float* __restrict__ in, uint32_t inStride, float* __restrict__ out,
                                            uint32_t outStride, uint32_t n)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float* i = in + (size_t)r * inStride;
    float* o = out + (size_t)r * outStride;
    const uint32_t LW = (uint32_t)(sizeof(RtLight) / 4), CW = (uint32_t)(sizeof(RtCamera) / 4);
    switch (func)
    {
    case KAT_SIN_LANE: o[0] = sinLane(i[0]); break;
    case KAT_SINCOS: katPut(o, sinCos(i[0])); break;
    case KAT_FASTLOG: o[0] = fastLog(i[0]); break;
    case KAT_FASTACOS: o[0] = fastACos(i[0]); break;
    case KAT_FASTATAN2: o[0] = fastATan2(i[0], i[1]); break;
    case KAT_FLOAT_NORMAL2: katPut(o, getFloatNormal2(i[0], i[1])); break;
    case KAT_HEMISPHERE_COS: katPut(o, getHemisphereCos(i[0], i[1])); break;
    case KAT_SPHERE: katPut(o, getSphere(i[0], i[1])); break;
    case KAT_CIRCLE: katPut(o, getCircle(i[0], i[1])); break;
    case KAT_ORTHO_BASIS: { V4 u, v; buildOrthonormalBasis(load4(i), u, v); katPut(o, u); katPut(o + 4, v); break; }
    case KAT_FRESNEL_DIELECTRIC: o[0] = fresnelDielectric(i[0], i[1]); break;
    case KAT_FRESNEL_METAL: o[0] = fresnelMetal(i[0], i[1], i[2]); break;
    case KAT_REFRACT3: katPut(o, refract3(load4(i), load4(i + 4), i[8])); break;
    case KAT_REFLECT3: katPut(o, reflect3(load4(i), load4(i + 4))); break;
    case KAT_MAKE_RAY:
    {
        const Ray ray = makeRay(load4(i), load4(i + 4));
        katPut(o, ray.dir); katPut(o + 4, ray.invDir); katPut(o + 8, ray.originDivDir); break;
    }
    case KAT_TRANSFORM_RAY:   // in: matrix[16], origin[4], dir[4] (normalized world ray)  out: origin, dir, invDir, originDivDir
    {
        Ray w; w.origin = load4(i + 16); w.dir = load4(i + 20); w.invDir = zero4(); w.originDivDir = zero4();
        const Ray l = transformRayUnsafe(loadM4(i), w);
        katPut(o, l.origin); katPut(o + 4, l.dir); katPut(o + 8, l.invDir); katPut(o + 12, l.originDivDir); break;
    }
    case KAT_FAST_INVERSE: { const M4 m = fastInverseNoScale(loadM4(i)); for (int k = 0; k < 4; ++k) katPut(o + 4 * k, m.r[k]); break; }
    case KAT_TRANSFORM_SCALED:   // in: matrix[16] (rotation x scale + translation), v[4]  out: TransformPoint, TransformVector, FastInverseNoScale().TransformPoint
    {
        const M4 m = loadM4(i); const V4 v = load4(i + 16);
        katPut(o, transformPoint(m, v)); katPut(o + 4, transformVector(m, v)); katPut(o + 8, transformPoint(fastInverseNoScale(m), v)); break;
    }
    case KAT_FRAME_COMPOSE:      // Scene::EvaluateIntersection's frame, Scene.cpp:311-348 (record layout: tests/golden/README.md): the function the shading kernels call
    {
        const M4 transform = loadM4(i);
        Ray ray; ray.origin = load4(i + 16); ray.dir = load4(i + 20); ray.invDir = zero4(); ray.originDivDir = zero4();
        const V4 worldPosition = rayAt(ray, i[24]);
        M4 frame;
        composeShadingFrame(transform, worldPosition, load4(i + 28), load4(i + 32), i[25] != 0.0f, load4(i + 36), frame);
        katPut(o, transformPoint(fastInverseNoScale(transform), worldPosition));
        for (int k = 0; k < 4; ++k) katPut(o + 4 + 4 * k, frame.r[k]);
        break;
    }
    case KAT_BOX_RAY:         // in: origin[4], direction[4] (unnormalized), bmin[3], bmax[3]; both slab-test forms of the traversal kernels
    {
        const Ray ray = makeRay(load4(i), load4(i + 4));
        float d = 0.0f; bool h = intersectBoxRay(ray, load3(i + 8), load3(i + 11), d);
        if (rayIsNaNFree(ray))   // the hardware min/max form k_trace uses for such rays must agree bit for bit
        {
            float d2 = 0.0f; const bool h2 = intersectBoxRayNoNaN(ray, i[8], i[9], i[10], i[11], i[12], i[13], d2);
            if (h2 != h || (__float_as_uint(d2) != __float_as_uint(d) && !(d2 == 0.0f && d == 0.0f))) { h = !h; d = __uint_as_float(0x7fc00001u); }   // poison: the comparison fails
        }
        o[0] = __uint_as_float(h ? 1u : 0u); o[1] = d; break;
    }
    case KAT_BOX_RAY_TWOSIDED:
    {
        const Ray ray = makeRay(load4(i), load4(i + 4));
        float a = 0.0f, b = 0.0f; const bool h = intersectBoxRayTwoSided(ray, load3(i + 8), load3(i + 11), a, b);
        o[0] = __uint_as_float(h ? 1u : 0u); o[1] = a; o[2] = b; break;
    }
    case KAT_TRIANGLE_RAY:    // in: origin[4], direction[4], v0[3], e1[3], e2[3]
    {
        const Ray ray = makeRay(load4(i), load4(i + 4));
        float u = 0, v = 0, t = 0; const bool h = intersectTriangleRay(ray, load3(i + 8), load3(i + 11), load3(i + 14), u, v, t);
        o[0] = __uint_as_float(h ? 1u : 0u); o[1] = u; o[2] = v; o[3] = t; break;
    }
    case KAT_SHAPE_INTERSECT: // in: kind(bits), param[4], origin[4], direction[4]
    {
        const Ray ray = makeRay(load4(i + 5), load4(i + 9));
        ShapeHit sh; sh.nearDist = 0; sh.farDist = 0;
        const bool h = shapeIntersect(__float_as_uint(i[0]), i + 1, ray, sh);
        o[0] = __uint_as_float(h ? 1u : 0u); o[1] = h ? sh.nearDist : 0.0f; o[2] = h ? sh.farDist : 0.0f; o[3] = __uint_as_float(sh.subObjectId); break;
    }
    case KAT_SHAPE_SAMPLE:    // in: kind, param[4], ref[4], u[3]
    {
        ShapeSample s; s.direction = zero4(); s.distance = s.pdf = s.cosAtSurface = -1.0f;
        const bool h = shapeSampleFrom(__float_as_uint(i[0]), i + 1, load4(i + 5), i + 9, s);
        o[0] = __uint_as_float(h ? 1u : 0u);
        if (h) { katPut(o + 1, s.direction); o[5] = s.distance; o[6] = s.pdf; o[7] = s.cosAtSurface; }
        else { for (int k = 1; k < 8; ++k) o[k] = 0.0f; }
        break;
    }
    case KAT_SHAPE_PDF: o[0] = shapePdf(__float_as_uint(i[0]), i + 1, load4(i + 5), load4(i + 9)); break;
    case KAT_SHAPE_EVAL:      // in: kind, param[4], param2[4], localPos[4]   out: frame rows 0..2, texCoord
    {
        Intersection is; for (int k = 0; k < 4; ++k) is.frame.r[k] = zero4();
        is.frame.r[3] = load4(i + 9); is.texCoord = zero4(); is.material = 0;
        shapeEvaluateIntersection(__float_as_uint(i[0]), i + 1, i + 5, is);
        katPut(o, is.frame.r[0]); katPut(o + 4, is.frame.r[1]); katPut(o + 8, is.frame.r[2]); katPut(o + 12, is.texCoord); break;
    }
    case KAT_LIGHT_ILLUMINATE: // in: RtLight as floats (sizeof/4), frame[16], u[3]
    {
        RtLight L; katLoad(&L, i, sizeof(RtLight));
        Intersection is; is.frame = loadM4(i + LW); is.texCoord = zero4(); is.material = 0;
        IlluminateResult ir;
        const float u[3] = { i[LW + 16], i[LW + 17], i[LW + 18] };
        const V4 rad = lightIlluminate<false>(scene, L, is, u, ir);
        katPut(o, rad); katPut(o + 4, ir.directionToLight); o[8] = ir.distance; o[9] = ir.directPdfW; o[10] = ir.cosAtLight; break;
    }
    case KAT_LIGHT_RADIANCE:  // in: RtLight, ray origin[4], dir[4] (light space), hitPoint[4], cosAtLight
    {
        RtLight L; katLoad(&L, i, sizeof(RtLight));
        Ray ray; ray.origin = load4(i + LW); ray.dir = load4(i + LW + 4); ray.invDir = zero4(); ray.originDivDir = zero4();
        float pdf = 0.0f;
        const V4 rad = lightGetRadiance<false>(scene, L, ray, load4(i + LW + 8), i[LW + 12], pdf);
        katPut(o, rad); o[4] = pdf; break;
    }
    case KAT_BSDF_SAMPLE:     // in: the first 64 bytes of RtMaterial (16 floats; no textures), outgoingDir[4] (local), u[3]
    {
        const RtMaterial m = katMaterial(i);
        ShadingData sd; sd.intersection.texCoord = zero4(); materialEvaluateShadingData<false>(scene, m, sd);
        BsdfSample s;
        const float u[3] = { i[20], i[21], i[22] };
        const bool ok = bsdfSampleImpl(m.bsdf, m, sd.mp, u, load4(i + 16), s);
        o[0] = __uint_as_float(ok ? 1u : 0u);
        if (ok) { katPut(o + 1, s.color); katPut(o + 5, s.incomingDir); o[9] = s.pdf; o[10] = __uint_as_float(s.event); }
        else { for (int k = 1; k < 11; ++k) o[k] = 0.0f; }
        break;
    }
    case KAT_BSDF_EVALUATE:   // in: RtMaterial, outgoingDir[4], incomingDir[4] (local)
    {
        const RtMaterial m = katMaterial(i);
        ShadingData sd; sd.intersection.texCoord = zero4(); materialEvaluateShadingData<false>(scene, m, sd);
        float pdf = 0.0f;
        const V4 c = bsdfEvaluate(m.bsdf, m, sd.mp, load4(i + 16), load4(i + 20), pdf);
        katPut(o, c); o[4] = almostZero4(c) ? 0.0f : pdf; break;
    }
    case KAT_CAMERA_RAY:      // in: RtCamera (sizeof/4 floats), coords[2], dof samples from seed {u0,u1} bits, Random::mSeed[2] (barrel distortion)
    {
        RtCamera cam; katLoad(&cam, i, sizeof(RtCamera));
        const uint32_t seeds[2] = { __float_as_uint(i[CW + 2]), __float_as_uint(i[CW + 3]) };
        Sampler s; s.seed = seeds; s.numDims = 2; s.blueNoiseLayers = 0; s.blueNoise = nullptr;
        s.bx = s.by = 0; s.salt = 0; s.generated = 0;
        s.fallback.s[0] = (uint64_t)__float_as_uint(i[CW + 4]) | ((uint64_t)__float_as_uint(i[CW + 5]) << 32);   // Random::mSeed of ctx.randomGenerator
        s.fallback.s[1] = (uint64_t)__float_as_uint(i[CW + 6]) | ((uint64_t)__float_as_uint(i[CW + 7]) << 32);
        // what k_generate stores (cameraGenerateRayParts) followed by the Ray constructor every consumer re-runs
        const Ray ray = cameraGenerateRay(cam, V4(i[CW], i[CW + 1], 0.0f, 0.0f), s);
        katPut(o, ray.origin); katPut(o + 4, ray.dir); katPut(o + 8, ray.invDir); katPut(o + 12, ray.originDivDir); break;
    }
    case KAT_LIGHT_EMIT:      // in: RtLight, positionSample[3], directionSample[2]
    {
        RtLight L; katLoad(&L, i, sizeof(RtLight));
        EmitResult er; er.position = zero4(); er.direction = zero4(); er.directPdfA = er.emissionPdfW = er.cosAtLight = 0.0f;
        const float up[3] = { i[LW], i[LW + 1], i[LW + 2] }, ud[2] = { i[LW + 3], i[LW + 4] };
        const V4 c = lightEmit(scene, L, up, ud, er);
        katPut(o, c); katPut(o + 4, er.position); katPut(o + 8, er.direction); o[12] = er.directPdfA; o[13] = er.emissionPdfW; o[14] = er.cosAtLight; break;
    }
    case KAT_LIGHT_ILLUMINATE_BIDIR: // in: RtLight, frame[16], u[3]; rendererSupportsSolidAngleSampling = false
    {
        RtLight L; katLoad(&L, i, sizeof(RtLight));
        Intersection is; is.frame = loadM4(i + LW); is.texCoord = zero4(); is.material = 0;
        IlluminateResult ir; float emissionPdfW = 0.0f;
        const float u[3] = { i[LW + 16], i[LW + 17], i[LW + 18] };
        const V4 rad = lightIlluminateBidir(scene, L, is, u, ir, emissionPdfW);
        katPut(o, rad); katPut(o + 4, ir.directionToLight); o[8] = ir.distance; o[9] = ir.directPdfW; o[10] = emissionPdfW; o[11] = ir.cosAtLight; break;
    }
    case KAT_LIGHT_RADIANCE_BIDIR:
    {
        RtLight L; katLoad(&L, i, sizeof(RtLight));
        Ray ray; ray.origin = load4(i + LW); ray.dir = load4(i + LW + 4); ray.invDir = zero4(); ray.originDivDir = zero4();
        float pdfA = 0.0f, pdfW = 0.0f;
        const V4 rad = lightGetRadianceBidir(scene, L, ray, load4(i + LW + 8), i[LW + 12], pdfA, pdfW);
        katPut(o, rad); o[4] = almostZero4(rad) ? 0.0f : pdfA; o[5] = almostZero4(rad) ? 0.0f : pdfW; break;
    }
    case KAT_BSDF_PDFS:       // in: RtMaterial, outgoingDir[4], incomingDir[4]   out: colour, pdf, reverse pdf, Pdf(Forward), Pdf(Reverse)
    {
        const RtMaterial m = katMaterial(i);
        ShadingData sd; sd.intersection.texCoord = zero4(); materialEvaluateShadingData<false>(scene, m, sd);
        float pdf = 0.0f, rev = 0.0f;
        const V4 c = bsdfEvaluate(m.bsdf, m, sd.mp, load4(i + 16), load4(i + 20), pdf, &rev);
        katPut(o, c); o[4] = almostZero4(c) ? 0.0f : pdf; o[5] = almostZero4(c) ? 0.0f : rev;
        o[6] = bsdfPdf(m.bsdf, m, sd.mp, load4(i + 16), load4(i + 20), false);
        o[7] = bsdfPdf(m.bsdf, m, sd.mp, load4(i + 16), load4(i + 20), true); break;
    }
    case KAT_CAMERA_FILM:     // in: RtCamera, world position[4], direction[4]   out: visible, film coords[4], PdfW
    {
        RtCamera cam; katLoad(&cam, i, sizeof(RtCamera));
        V4 film = zero4();
        const bool ok = cameraWorldToFilm(cam, load4(i + CW), film);
        o[0] = __uint_as_float(ok ? 1u : 0u); o[1] = ok ? film.x : 0.0f; o[2] = ok ? film.y : 0.0f; o[3] = ok ? film.z : 0.0f; o[4] = ok ? film.w : 0.0f;
        o[5] = cameraDirectionPdfW(cam, load4(i + CW + 4)); break;
    }
    case KAT_FILM_SPLAT:      // in: pos[2], width, height, mSeedSimd4[0..1]   out: x, y (0xFFFFFFFF = outside), generator state after
    {
        RandomSimd rng; katLoad(rng.seed0, i + 4, 16); katLoad(rng.seed1, i + 8, 16);
        uint32_t x = 0xFFFFFFFFu, y = 0xFFFFFFFFu;
        if (!filmSplatPixel(V4(i[0], i[1], 0.0f, 0.0f), __float_as_uint(i[2]), __float_as_uint(i[3]), rng.getVector4(), x, y)) { x = y = 0xFFFFFFFFu; }
        o[0] = __uint_as_float(x); o[1] = __uint_as_float(y);
        for (int k = 0; k < 2; ++k)
        {
            o[2 + 2 * k] = __uint_as_float((uint32_t)rng.seed0[k]); o[3 + 2 * k] = __uint_as_float((uint32_t)(rng.seed0[k] >> 32));
            o[6 + 2 * k] = __uint_as_float((uint32_t)rng.seed1[k]); o[7 + 2 * k] = __uint_as_float((uint32_t)(rng.seed1[k] >> 32));
        }
        break;
    }
    case KAT_HSV_TO_RGB: katPut(o, debugTriangleIdColor(__float_as_uint(i[0]), __float_as_uint(i[1]))); break;
    case KAT_PACKED_PHOTON:   // in: direction[4], colour[4]   out: packed direction, packed colour (2 words), unpacked direction[4], colour[4]
    {
        const uint32_t pd = packUnitVector(load4(i));
        float py = 0.0f; uint32_t pc = 0u; packColorHdr(load4(i + 4), py, pc);
        o[0] = __uint_as_float(pd); o[1] = py; o[2] = __uint_as_float(pc);
        katPut(o + 3, unpackUnitVector(pd)); katPut(o + 7, unpackColorHdr(py, pc)); break;
    }
    default: break;
    }
}

// GenericSampler::ResetPixel + GetInt (GenericSampler.cpp:69-113) for dims [0, count) of one pixel per thread.
// in: x, y, useBlueNoise, numDims, seed[numDims] (uint32 bit-cast); out: count ints, then count GetFloat() values
__global__ void __launch_bounds__(64) k_kat_sampler(const uint16_t* __restrict__ blueNoise, const float* __restrict__ in, uint32_t inStride, float* __restrict__ out,
                                                    uint32_t count, uint32_t n)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint32_t* i = reinterpret_cast<const uint32_t*>(in + (size_t)r * inStride);
    float* o = out + (size_t)r * 2u * count;
    Sampler s; s.seed = i + 4; s.numDims = i[3]; s.blueNoise = blueNoise; s.blueNoiseLayers = (blueNoise && i[2]) ? 4u : 0u;
    s.resetPixel(i[0], i[1], 0ull, 0ull);
    for (uint32_t k = 0; k < count; ++k)
    {
        Sampler copy = s;
        o[k] = __uint_as_float(s.getInt());
        o[count + k] = copy.getFloat();
    }
}

// MeshShape::Traverse / Traverse_Shadow / EvaluateIntersection through the traversal state machine k_trace runs (rt_device_traverse.h)
// and meshEvaluateIntersection, on the single mesh object of the uploaded scene, objectID reported as 7 like the generator does
// (layout of tests/golden/mesh_kat.bin).  rays: n * 7 floats (origin, direction, tmax) in the mesh's space; out: n * 19 words
//   [objectId, subObjectId, distance, u, v, shadowHit, frame0.xyzw, frame2.xyzw, texCoord.xyzw, material]
__global__ void __launch_bounds__(64) k_kat_mesh(const RtSceneDesc scene, const float* __restrict__ rays, uint32_t n, uint32_t* __restrict__ out)
{
    __shared__ uint32_t sStack[RT_KAT_MESH_STACK * 64];
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const LdsStack stack = { sStack + threadIdx.x, 64u };
    const RtMesh& mesh = scene.meshes[scene.objects[0].meshIndex];
    const float* in = rays + 7 * (size_t)r;
    uint32_t* o = out + 19 * (size_t)r;
    const Ray ray = makeRay(V4(in[0], in[1], in[2], 0.0f), V4(in[3], in[4], in[5], 0.0f));
    Counters cnt; zeroCounters(cnt);
    Hit hp; hp.objectId = RT_INVALID_OBJECT; hp.subObjectId = 0; hp.distance = in[6]; hp.u = 0.0f; hp.v = 0.0f;
    auto reloadWorldRay = [&]() -> Ray { return ray; };
    for (int pass = 0; pass < 2; ++pass)   // 0: closest hit, 1: any hit
    {
        TravState s;
        s.ray = ray; s.nanFree = rayIsNaNFree(ray); s.shadow = pass == 1; s.occluded = false; s.hitDistance = in[6];
        s.stackSize = 0; s.levelBase = 0; s.leafNext = 1; s.leafEnd = 1; s.objectId = 7; s.triBase = mesh.firstTriangle;
        s.nodes = scene.meshNodes + mesh.firstNode; s.cur = packNode(s.nodes[0].childIndex, s.nodes[0].leaves); s.mode = mesh.numNodes ? TRAV_MESH : TRAV_DONE;
        auto onHit = [&](uint32_t objectId, uint32_t subObjectId, float distance, float u, float v)
        {
            hp.objectId = objectId; hp.subObjectId = subObjectId; hp.distance = distance; hp.u = u; hp.v = v;
        };
        while (s.mode != TRAV_DONE)
        {
            if (travIsInterior(s)) { if (s.nanFree) travStepInterior<false, false>(s, stack, cnt); else travStepInterior<false, true>(s, stack, cnt); }
            else travStepOther<false>(s, scene, stack, cnt, reloadWorldRay, onHit);
        }
        if (pass == 1) o[5] = s.occluded ? 1u : 0u;
    }
    const bool hit = hp.objectId == 7u;
    o[0] = hp.objectId; o[1] = hit ? hp.subObjectId : 0u; o[2] = __float_as_uint(hp.distance); o[3] = __float_as_uint(hit ? hp.u : 0.0f); o[4] = __float_as_uint(hit ? hp.v : 0.0f);
    for (int k = 6; k < 18; ++k) o[k] = 0u;
    o[18] = 0xFFFFFFFFu;
    if (hit)
    {
        Intersection is; for (int k = 0; k < 4; ++k) is.frame.r[k] = zero4();
        is.texCoord = zero4(); is.material = RT_NO_MATERIAL;
        meshEvaluateIntersection(scene, mesh, hp, is);
        const V4 f0 = is.frame.r[0], f2 = is.frame.r[2], tc = is.texCoord;
        o[6] = __float_as_uint(f0.x); o[7] = __float_as_uint(f0.y); o[8] = __float_as_uint(f0.z); o[9] = __float_as_uint(f0.w);
        o[10] = __float_as_uint(f2.x); o[11] = __float_as_uint(f2.y); o[12] = __float_as_uint(f2.z); o[13] = __float_as_uint(f2.w);
        o[14] = __float_as_uint(tc.x); o[15] = __float_as_uint(tc.y); o[16] = __float_as_uint(tc.z); o[17] = __float_as_uint(tc.w);
        o[18] = is.material;
    }
}
