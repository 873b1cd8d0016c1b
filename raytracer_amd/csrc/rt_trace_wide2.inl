// rt_trace_wide2.inl -- the 4-wide walk for TWO-LEVEL scenes (Scene::Traverse over the top-level BVH of the objects, Scene.cpp:219-261, into
// MeshShape::Traverse for mesh objects, Traverse_Object / Traverse_Object_Shadow for analytic shapes and light objects): a 4-wide collapse
// of the reference's top-level tree over 4-wide collapses of its mesh trees.  Included by rt_trace.hip (kernels: RT_DEVICE_KERNELS) and rt_runtime.hip (tree builders: RT_HOST_BUILDERS) after rt_trace_wide.inl, whose node
// format, slab test, sorting network and exactness argument it shares; what is new here is only the second level.
//
// Why.  Round 2's k_trace_wide serves scenes with ONE mesh object; the Cornell box (ten analytic instances), a Sponza with props, every
// scene with an area light fell back to the binary walk of k_trace.
//
// Levels.  WideScene::levels[o] describes the 4-wide tree of mesh object o (nodes, leaf gates, its own 16-bit grid in the mesh's local
// space), levels[numObjects] the top-level tree (grid in world space), whose leaves are runs of one or two OBJECTS.  A lane keeps the
// folded slab constants, origin and direction of the level it is in; the world ray is rebuilt from the path records when a mesh is left
// (as k_trace does), its exact slab terms -- invDir and the reference's STALE originDivDir (PathTracerMIS.cpp:392-393) -- wait in LDS
// for the top-level leaf gates.  The stack column is shared: the mesh level pushes above what the top level has deferred.
//
// Exactness (closest-hit rays).  Candidates are what the reference's walk could accept: an object's own hit distance (analytic shape: the
// near distance if positive, else the far one, as Traverse_Object picks; light: lightTestRayHit; mesh: its triangles behind their exact
// leaf boxes), for objects whose exact top-level leaf box the ray passes.  All of them are evaluated with the reference's arithmetic, in
// an order of our own; the running minimum and the runner-up are tracked across both levels, and a ray whose runner-up lies within tol
// (16 ulps of the largest slab term it has met, world or local, plus -- for bounce rays -- the 1e-3 by which the reference's top-level box
// distances are off, see the refill) of its best goes to the binary-tree kernel -- as do rays with a zero
// direction component at either level, origins far outside a grid, and rays whose stack would overflow.  Any-hit rays: an OR over the
// same candidates with the reference's per-object conditions against the fixed ray length.

struct WideLevel   // 64 bytes
{
    uint32_t nodeBase;    // first node of the level in WideScene::nodes, in nodes (a node = four float4)
    uint32_t gateBase;    // the exact box of the leaf whose first primitive is p: gate[gateBase + 2 p] (min), gate[gateBase + 2 p + 1] (max)
    uint32_t triBase;     // mesh levels: the mesh's first triangle in RtSceneDesc::triangles
    uint32_t valid;       // 0: an object without a tree (an empty mesh)
    float base[3], step[3], bound[3];
    uint32_t pad[3];
};
struct WideScene
{
    const float4* nodes;
    const float4* gate;
    const WideLevel* levels;   // [numObjects] mesh objects (valid only for those), [numObjects] = the top level
    uint32_t numObjects;       // >= 2 (one-object scenes are Scene::Traverse's bypass: no top-level tree; k_trace_wide or k_trace serve them)
};

#ifdef RT_DEVICE_KERNELS
#define RT_WIDE2_WORLD_WORDS 6u   // per lane in LDS: the world ray's invDir and (stale) originDivDir
#define RT_WIDE2_LOCAL_EXACT 64u

template <int kStack>
__global__ void __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(kStack <= 24 ? 5 : 1))) k_trace_wide2(const RtSceneDesc scene, const WideScene wide, const Paths paths,
                                                         const uint32_t* __restrict__ queue, const uint32_t* __restrict__ queueCount,
                                                         const uint32_t* __restrict__ shadowQueue, const uint32_t* __restrict__ shadowCount,
                                                         uint32_t* __restrict__ cursor, unsigned long long* counters, const WideTuning tune)
{
    __shared__ uint32_t sStack[kStack * RT_BLOCK];
    __shared__ float sWorld[RT_WIDE2_WORLD_WORDS * RT_BLOCK];
    __shared__ uint32_t sDensePrefix[RT_DENSE_SHARDS + 1u];
    // the block's own hand-over lists (rt_trace_wide.inl, WideLocal): 64 entries per kind keep five blocks of 30.7 KB on a CU
    __shared__ uint32_t sLocalExact[RT_WIDE2_LOCAL_EXACT], sLocalShadow[RT_WIDE2_LOCAL_EXACT], sLocalCounts[4];
    if (threadIdx.x < 4u) sLocalCounts[threadIdx.x] = 0u;
    __syncthreads();
    const WideLocal localLists = { sLocalExact, &sLocalCounts[0], sLocalShadow, &sLocalCounts[1], tune.localExact != 0u ? RT_WIDE2_LOCAL_EXACT : 0u };
    uint32_t* const stack = sStack + threadIdx.x;
    float* const worldTerms = sWorld + threadIdx.x;   // word w at worldTerms[w * RT_BLOCK]
    if (tune.denseCounts) { denseLoadPrefix(tune.denseCounts, sDensePrefix); __syncthreads(); }
    const uint32_t numClosest = tune.denseCounts ? sDensePrefix[RT_DENSE_SHARDS] : (queueCount ? *queueCount : 0u);
    const uint32_t count = numClosest + (shadowCount ? *shadowCount : 0u);
    const float inf = __uint_as_float(0x7f800000u);
    const WideLevel* const topLevel = wide.levels + wide.numObjects;

    // per-lane state: the ray of the CURRENT level (world at the top, object space inside a mesh)
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0;
    float ax = 0, ay = 0, az = 0, bx = 0, by = 0, bz = 0;
    float best = 0, second = 0, tol = 0;
    uint32_t cur = RT_QUANT_DONE, sp = 0, levelBase = 0, nodeBase = 0, slot = 0, light = 0;
    uint32_t objectId = 0;     // the mesh object being walked
    uint32_t leafRest = 0;     // objects of the current top-level leaf not yet visited: next index | how many << 30 (0: none)
    bool have = false, shadow = false, occluded = false, handOver = false, inMesh = false, exhausted = false;
    uint32_t numRetraced = 0, numShadowRays = 0;   // per WAVE (ballots at wave-uniform points: scalar registers)

    uint32_t chunkSize = count / (gridDim.x * ((uint32_t)RT_BLOCK / 64u) * 4u);
    chunkSize = chunkSize < tune.chunkMin ? tune.chunkMin : (chunkSize > 1024u ? 1024u : chunkSize);
    WaveChunk chunk = { 0u, 0u };

    // the world ray of the lane's request, exactly as at refill time (PathTracerMIS.cpp:86 / :392: Ray::Ray normalises, then the origin moves
    // along the direction and originDivDir stays what it was)
    auto loadWorldRay = [&]() -> Ray
    {
        float4 origin, dir; float offset;
        if (shadow) { origin = ldStream(prec(paths, R_SH_P, slot)); dir = ldStream(pshadow(paths, light, 0, slot)); offset = tune.shadowOffset; }
        else { origin = ldStream(prec(paths, R_ORIGIN, slot)); dir = ldStream(prec(paths, R_DIR, slot)); offset = 0.001f; }
        Ray world = makeRay(V4(origin.x, origin.y, origin.z, 0.0f), V4(dir.x, dir.y, dir.z, 0.0f));
        if (shadow || (ubits(origin.w) & 0xFFu) != 0u) world.origin = world.origin + world.dir * offset;
        return world;
    };
    // folded slab constants of `ray` on the grid of `level`; false: the reference's walk only (NaNs in its slab test, or the folded test's rounding
    // could eat the spare grid step).  Raises tol to 16 ulps of the largest slab term of this level.
    auto enterLevel = [&](const Ray& ray, const WideLevel& level) -> bool
    {
        const float mx = fabsf(ray.originDivDir.x) + level.bound[0] * fabsf(ray.invDir.x);
        const float my = fabsf(ray.originDivDir.y) + level.bound[1] * fabsf(ray.invDir.y);
        const float mz = fabsf(ray.originDivDir.z) + level.bound[2] * fabsf(ray.invDir.z);
        const float fold = 4.76837158203125e-07f;   // 2^-21
        if (!(rayIsNaNFree(ray) && mx * fold < level.step[0] * fabsf(ray.invDir.x) && my * fold < level.step[1] * fabsf(ray.invDir.y) && mz * fold < level.step[2] * fabsf(ray.invDir.z))) return false;
        ox = ray.origin.x; oy = ray.origin.y; oz = ray.origin.z; dx = ray.dir.x; dy = ray.dir.y; dz = ray.dir.z;
        ax = level.step[0] * ray.invDir.x; ay = level.step[1] * ray.invDir.y; az = level.step[2] * ray.invDir.z;
        bx = __fmaf_rn(level.base[0], ray.invDir.x, -ray.originDivDir.x);
        by = __fmaf_rn(level.base[1], ray.invDir.y, -ray.originDivDir.y);
        bz = __fmaf_rn(level.base[2], ray.invDir.z, -ray.originDivDir.z);
        if (!shadow) tol = fmaxf(tol, fmaxf(fmaxf(mx, my), mz) * 1.9073486328125e-06f);   // 2^-19
        nodeBase = level.nodeBase;
        return true;
    };
    // a candidate hit of a closest-hit ray (an analytic shape or a light object; triangles keep their pairwise form below)
    auto candidate = [&](float t, uint32_t object, uint32_t subObject)
    {
        if (t < best) { second = best; best = t; prec(paths, R_HIT, slot) = f4(fbits(object), fbits(subObject), t, 0.0f); prec(paths, R_SAMPLER, slot).x = 0.0f; }
        else second = fminf(second, t);
    };
    auto popOrDone = [&]() { if (sp == levelBase) cur = RT_QUANT_DONE; else { --sp; cur = stack[sp * RT_BLOCK]; } };

    for (;;)
    {
        const bool interior = have && (cur >> RT_NODE_LEAVES_SHIFT) == 0u;
        const bool other = have && !interior;
        const unsigned long long mI = __ballot(interior), mO = __ballot(other);
        const uint32_t nIdle = 64u - (uint32_t)__popcll(mI) - (uint32_t)__popcll(mO);
        if (!exhausted && (nIdle == 64u || nIdle >= tune.refillMinIdle))
        {
            // ---- refill ----
            if (chunk.next >= chunk.end)
            {
                waveClaimChunk(chunk, cursor, chunkSize, count);
                if (chunk.next >= chunk.end) { exhausted = true; continue; }
            }
            uint32_t idx = waveTake(!have, chunk);
            if (tune.reverseOrder != 0u && idx != 0xFFFFFFFFu) idx = count - 1u - idx;   // (as in k_trace_wide: the any-hit requests first)
            bool tookShadow = false;
            if (idx != 0xFFFFFFFFu)
            {
                shadow = idx >= numClosest;
                const uint32_t request = shadow ? shadowQueue[idx - numClosest]
                                                : (tune.denseCounts ? denseLiveSlot(sDensePrefix, tune.denseShardCapacity, idx) : (queue ? queue[idx] : idx));
                float maxDistance = inf;
                if (shadow) { light = request / paths.capacity; slot = request - light * paths.capacity; maxDistance = pshadow(paths, light, 0, slot).w; }
                else { slot = request; light = 0u; }
                const Ray world = loadWorldRay();
                best = maxDistance; second = inf; tol = 0.0f; occluded = false; handOver = false; inMesh = false;
                sp = 0u; levelBase = 0u; leafRest = 0u; have = true;
                worldTerms[0 * RT_BLOCK] = world.invDir.x; worldTerms[1 * RT_BLOCK] = world.invDir.y; worldTerms[2 * RT_BLOCK] = world.invDir.z;
                worldTerms[3 * RT_BLOCK] = world.originDivDir.x; worldTerms[4 * RT_BLOCK] = world.originDivDir.y; worldTerms[5 * RT_BLOCK] = world.originDivDir.z;
                if (!enterLevel(world, *topLevel)) { handOver = true; cur = RT_QUANT_DONE; }
                else
                {
                    cur = 0u;   // node 0 of the top level holds the children of the binary tree's root
                    // The reference tests the top-level boxes with the STALE originDivDir of a bounce ray (PathTracerMIS.cpp:392-393: the origin moved
                    // 1e-3 along the direction after Ray::Ray computed it), i.e. its box distances are measured from a point 1e-3 behind the origin
                    // the object hit distances are measured from.  Its walk therefore culls an object up to 1e-3 CLOSER than the hit it already
                    // holds, and which of two objects within 1e-3 of each other it returns depends on its visiting order: the runner-up window
                    // of such a ray is that offset plus the rounding term (found by bench.py's replay check on the Cornell box: 5 of 600 000 paths).
                    if (!shadow && (ubits(prec(paths, R_ORIGIN, slot).w) & 0xFFu) != 0u) tol += 0.00100098f;
                }
                tookShadow = shadow;
            }
            numShadowRays += (uint32_t)__popcll(__ballot(tookShadow));
            continue;
        }
        if ((mI | mO) == 0ull) break;
        if (mI != 0ull && (uint32_t)__popcll(mO) < tune.otherMinLanes)
        {
            // ---- interior phase (either level): four conservative slab tests per step ----
            bool in = interior;
            const float limit = best + (tol + tol);
            // byte selectors of the slab test, rebuilt per phase from the current level's constants (as k_trace_wide since round 5: three registers less across the other phases)
            const uint32_t selX = ax < 0.0f ? RT_WIDE_SEL_X_NEG : RT_WIDE_SEL_X_POS, selY = ay < 0.0f ? RT_WIDE_SEL_Y_NEG : RT_WIDE_SEL_Y_POS, selZ = az < 0.0f ? RT_WIDE_SEL_Z_NEG : RT_WIDE_SEL_Z_POS;
            for (;;)
            {
                if (in)
                {
                    const float4* p = wide.nodes + 4u * (size_t)(nodeBase + cur);
                    const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
                    float n0, f0, n1, f1, n2, f2, n3, f3;
                    RT_WIDE_SLAB(q0, n0, f0); RT_WIDE_SLAB(q1, n1, f1); RT_WIDE_SLAB(q2, n2, f2); RT_WIDE_SLAB(q3, n3, f3);
                    const bool h0 = f0 >= n0 && n0 < limit, h1 = f1 >= n1 && n1 < limit, h2 = f2 >= n2 && n2 < limit, h3 = f3 >= n3 && n3 < limit;
                    // any-hit rays walk the FARTHEST entered child next (rt_trace_wide.inl has the reasoning and the barrier's); `tol` is zero for any-hit rays only
                    float tolNow = tol;
                    asm volatile("" : "+v"(tolNow));
                    const uint32_t orderFlip = (tolNow == 0.0f && tune.anyHitFarFirst != 0u) ? 0u : 0x7FFFFFFFu;
                    uint32_t k0 = h0 ? orderFlip ^ ubits(n0) : 0xFFFFFFFFu, k1 = h1 ? orderFlip ^ ubits(n1) : 0xFFFFFFFFu;
                    uint32_t k2 = h2 ? orderFlip ^ ubits(n2) : 0xFFFFFFFFu, k3 = h3 ? orderFlip ^ ubits(n3) : 0xFFFFFFFFu;
                    uint32_t r0 = ubits(q0.w), r1 = ubits(q1.w), r2 = ubits(q2.w), r3 = ubits(q3.w);
#define RT_WIDE_CE(ka, ra, kb, rb) { const bool c_ = ka > kb; const uint32_t lo_ = min(ka, kb), hi_ = max(ka, kb), rl_ = c_ ? rb : ra, rh_ = c_ ? ra : rb; ka = lo_; kb = hi_; ra = rl_; rb = rh_; }
                    RT_WIDE_CE(k0, r0, k1, r1) RT_WIDE_CE(k2, r2, k3, r3) RT_WIDE_CE(k0, r0, k2, r2) RT_WIDE_CE(k1, r1, k3, r3) RT_WIDE_CE(k1, r1, k2, r2)
#undef RT_WIDE_CE
                    const uint32_t numHit = (h0 ? 1u : 0u) + (h1 ? 1u : 0u) + (h2 ? 1u : 0u) + (h3 ? 1u : 0u);
                    uint32_t* const top = stack + sp * RT_BLOCK;
                    top[0] = r0; top[RT_BLOCK] = r1; top[2 * RT_BLOCK] = r2;
                    if (numHit != 0u) { cur = numHit == 1u ? r0 : (numHit == 2u ? r1 : (numHit == 3u ? r2 : r3)); sp += numHit - 1u; }
                    else if (sp == levelBase) cur = RT_QUANT_DONE;
                    else { --sp; cur = stack[sp * RT_BLOCK]; }
                    if (sp + 3u > (uint32_t)kStack) { handOver = true; cur = RT_QUANT_DONE; }   // the next step could not push: the binary-tree kernel takes the ray
                }
                in = in && (cur >> RT_NODE_LEAVES_SHIFT) == 0u;
                const unsigned long long m = __ballot(in);
                if (m == 0ull || 64u - nIdle - (uint32_t)__popcll(m) >= tune.otherMinLanes) break;
            }
        }
        else
        {
        bool handedOver = false, uncountShadow = false;
        if (other)
        {
            // ---- everything else, one step per round: a mesh leaf, leaving a mesh, a top-level leaf, the next object of it, finishing ----
            bool finish = handOver;
            if (!finish && cur != RT_QUANT_DONE && (cur >> RT_NODE_LEAVES_SHIFT) == 3u) popOrDone();   // an unused slot of a node (RT_WIDE_EMPTY) whose corner point the ray happened to meet
            else if (!finish && inMesh)
            {
                if (RT_WIDE_IS_LEAF(cur))
                {
                    // MeshShape::Traverse_Leaf(_Shadow), MeshShape.cpp:134-207, as in k_trace_wide
                    const WideLevel& level = wide.levels[objectId];
                    const uint32_t numLeaves = cur >> RT_NODE_LEAVES_SHIFT, first = cur & RT_NODE_CHILD_MASK;
                    const RtTriangle* const tris = scene.triangles + level.triBase;
                    Ray ray; ray.origin = V4(ox, oy, oz, 0.0f); ray.dir = V4(dx, dy, dz, 0.0f);
                    V4 v0, e1, e2, nv0, ne1, ne2;
                    loadTriangle(tris + first, v0, e1, e2);
                    loadTriangle(tris + first + (numLeaves > 1u ? 1u : 0u), nv0, ne1, ne2);
                    float u0, v0_, t0, u1 = 0.0f, v1 = 0.0f, t1 = inf;
                    if (!intersectTriangleRay(ray, v0, e1, e2, u0, v0_, t0)) t0 = inf;
                    if (numLeaves > 1u && !intersectTriangleRay(ray, nv0, ne1, ne2, u1, v1, t1)) t1 = inf;
                    const float lo = fminf(t0, t1);
                    if (lo < best + tol)
                    {
                        const float4 gmin = wide.gate[level.gateBase + 2u * first], gmax = wide.gate[level.gateBase + 2u * first + 1u];
                        const Ray gateRay = makeRayUnsafe3(ray.origin, ray.dir);
                        float nearD;
                        const bool pass = intersectBoxRayNoNaN(gateRay, gmin.x, gmin.y, gmin.z, gmax.x, gmax.y, gmax.z, nearD) && (!shadow || nearD < best);
                        if (pass)
                        {
                            if (shadow) { if (lo < best) occluded = true; }
                            else
                            {
                                const float hi = fmaxf(t0, t1);
                                if (lo < best)
                                {
                                    second = fminf(best, hi);
                                    best = lo;
                                    const bool firstWins = t0 <= t1;
                                    prec(paths, R_HIT, slot) = f4(fbits(objectId), fbits(first + (firstWins ? 0u : 1u)), lo, firstWins ? u0 : u1);
                                    prec(paths, R_SAMPLER, slot).x = firstWins ? v0_ : v1;
                                }
                                else second = fminf(second, lo);
                            }
                        }
                    }
                    if (occluded) cur = RT_QUANT_DONE; else popOrDone();
                }
                if (cur == RT_QUANT_DONE)
                {
                    // GenericTraverse<MeshShape> returned: back to the object loop of the top-level leaf, in world space
                    inMesh = false; sp = levelBase; levelBase = 0u;
                    if (occluded) finish = true;
                    else
                    {
                        const Ray world = loadWorldRay();
                        ox = world.origin.x; oy = world.origin.y; oz = world.origin.z; dx = world.dir.x; dy = world.dir.y; dz = world.dir.z;
                        {
                            ax = topLevel->step[0] * world.invDir.x; ay = topLevel->step[1] * world.invDir.y; az = topLevel->step[2] * world.invDir.z;
                            bx = __fmaf_rn(topLevel->base[0], world.invDir.x, -world.originDivDir.x);
                            by = __fmaf_rn(topLevel->base[1], world.invDir.y, -world.originDivDir.y);
                            bz = __fmaf_rn(topLevel->base[2], world.invDir.z, -world.originDivDir.z);
                            nodeBase = topLevel->nodeBase;
                        }
                        if (leafRest == 0u) { popOrDone(); if (cur == RT_QUANT_DONE) finish = true; }
                    }
                }
            }
            else if (!finish)
            {
                // ---- top level ----
                if (leafRest == 0u)
                {
                    if (cur == RT_QUANT_DONE) finish = true;
                    else
                    {
                        // a top-level leaf: its objects count only if the ray passes the leaf's EXACT box (the test the reference's walk performs
                        // before it reaches them; any-hit rays: with the reference's entry-distance test against the fixed ray length)
                        const uint32_t first = cur & RT_NODE_CHILD_MASK;
                        const float4 gmin = wide.gate[topLevel->gateBase + 2u * first], gmax = wide.gate[topLevel->gateBase + 2u * first + 1u];
                        Ray gateRay;
                        gateRay.invDir = V4(worldTerms[0 * RT_BLOCK], worldTerms[1 * RT_BLOCK], worldTerms[2 * RT_BLOCK], 0.0f);
                        gateRay.originDivDir = V4(worldTerms[3 * RT_BLOCK], worldTerms[4 * RT_BLOCK], worldTerms[5 * RT_BLOCK], 0.0f);
                        float nearD;
                        const bool pass = intersectBoxRayNoNaN(gateRay, gmin.x, gmin.y, gmin.z, gmax.x, gmax.y, gmax.z, nearD) && (!shadow || nearD < best);
                        if (pass) leafRest = cur;
                        else { popOrDone(); if (cur == RT_QUANT_DONE) finish = true; }
                    }
                }
                if (!finish && leafRest != 0u)
                {
                    // Scene::Traverse_Object / Traverse_Object_Shadow for the next object of the leaf (Scene.cpp:128-217)
                    const uint32_t objectID = leafRest & RT_NODE_CHILD_MASK, remaining = leafRest >> RT_NODE_LEAVES_SHIFT;
                    leafRest = remaining > 1u ? ((objectID + 1u) | ((remaining - 1u) << RT_NODE_LEAVES_SHIFT)) : 0u;
                    const RtObject& obj = scene.objects[objectID];
                    Ray world; world.origin = V4(ox, oy, oz, 0.0f); world.dir = V4(dx, dy, dz, 0.0f);
                    const M4 invTransform = loadM4(obj.invTransform);
                    const Ray lray = makeRayUnsafe3(transformPoint(invTransform, world.origin), transformVector(invTransform, world.dir));   // = transformRayUnsafe
                    bool enteredMesh = false;
                    if (obj.objectKind == RT_OBJECT_LIGHT)
                    {
                        float lightDistance;
                        if (lightTestRayHit(scene.lights[obj.lightIndex], lray, lightDistance))
                        {
                            if (shadow) { if (lightDistance < best) occluded = true; }
                            else if (lightDistance > 0.0f) candidate(lightDistance, objectID, RT_LIGHT_OBJECT);
                        }
                    }
                    else if (obj.shapeKind == RT_SHAPE_MESH)
                    {
                        const WideLevel& level = wide.levels[objectID];
                        if (level.valid != 0u)
                        {
                            if (!enterLevel(lray, level)) { handOver = true; finish = true; }
                            else { inMesh = true; enteredMesh = true; objectId = objectID; levelBase = sp; cur = 0u; }
                        }
                    }
                    else
                    {
                        ShapeHit sh;
                        if (shapeIntersect(obj.shapeKind, obj.shapeParam, lray, sh))
                        {
                            if (shadow) { if (sh.farDist > 0.0f && sh.nearDist < best) occluded = true; }
                            else if (sh.nearDist > 0.0f) candidate(sh.nearDist, objectID, sh.subObjectId);
                            else if (sh.farDist > 0.0f) candidate(sh.farDist, objectID, sh.subObjectId);
                        }
                    }
                    if (occluded) finish = true;
                    else if (!finish && !enteredMesh && leafRest == 0u)
                    {
                        popOrDone(); if (cur == RT_QUANT_DONE) finish = true;
                    }
                }
            }
            if (finish)
            {
                if (handOver)
                {
                    if (shadow) { widePushExact(tune, localLists, true, light * paths.capacity + slot); uncountShadow = true; }   // counted by the kernel that resolves it
                    else widePushExact(tune, localLists, false, slot);
                    handedOver = true;
                }
                else if (shadow)
                {
                    if (occluded) pshadow(paths, light, 0, slot).w = -1.0f;
                }
                else if (best == inf) prec(paths, R_HIT, slot) = f4(fbits(RT_INVALID_OBJECT), fbits(0u), inf, 0.0f);
                else if (second <= best + tol)
                {
                    widePushExact(tune, localLists, false, slot);   // a runner-up too close to call: the reference's own walk decides
                    handedOver = true;
                }
                have = false; cur = RT_QUANT_DONE; leafRest = 0u; inMesh = false;
            }
        }
        numRetraced += (uint32_t)__popcll(__ballot(handedOver)); numShadowRays -= (uint32_t)__popcll(__ballot(uncountShadow));
        }
    }
    __shared__ uint32_t sTally[2];
    if (threadIdx.x < 2u) sTally[threadIdx.x] = 0u;
    __syncthreads();
    if ((threadIdx.x & 63u) == 0u)
    {
        if (numShadowRays) atomicAdd(&sTally[0], numShadowRays);
        if (numRetraced) atomicAdd(&sTally[1], numRetraced);
    }
    __syncthreads();
    if (threadIdx.x == 0u && sTally[0]) atomicAdd(&counters[C_SHADOW], (unsigned long long)sTally[0]);
    if (threadIdx.x == 1u && sTally[1]) atomicAdd(&counters[RT_COUNTER_RETRACED], (unsigned long long)sTally[1]);
    // the rays this block's walk did not decide, by the reference's own walk (as k_trace_wide)
    __syncthreads();
    if (threadIdx.x < 2u && sLocalCounts[threadIdx.x] > RT_WIDE2_LOCAL_EXACT) sLocalCounts[threadIdx.x] = RT_WIDE2_LOCAL_EXACT;
    __syncthreads();
    if (sLocalCounts[0] + sLocalCounts[1] != 0u)
    {
        const TravTuning exactTune = { tune.refillMinIdle, tune.otherMinLanes, tune.shadowOffset, nullptr, nullptr, RT_ABORT_CLOSEST_AFTER, nullptr, 0u, RT_RETRACE_SPLIT_AFTER };
        traceBinaryLoop<kStack, false>(scene, paths, sLocalExact, &sLocalCounts[0], sLocalShadow, &sLocalCounts[1], &sLocalCounts[2], counters, exactTune, sStack, sDensePrefix,
                                              (uint32_t)RT_BLOCK / 64u);
    }
}

#endif   // RT_DEVICE_KERNELS

#ifdef RT_HOST_BUILDERS
// ---- host: the 4-wide trees of a two-level scene --------------------------------------------------------------------------------------
struct WideLevelBuild
{
    std::vector<float4> nodes, gate;   // gate: 2 float4 per primitive index
    float base[3], step[3], bound[3];
    bool ok = false;
};

// One level (the top-level tree over the objects, or a mesh's tree over its triangles).  A tree whose root is a leaf (one or two
// primitives: a quad, two objects) gets a single wide node with that leaf as its only child.
static WideLevelBuild buildWideLevel(const RtNode* nodes, uint32_t numNodes, uint32_t numPrimitives, uint32_t depth)
{
    WideLevelBuild out;
    if (numNodes == 0u || numPrimitives == 0u) return out;
    const uint32_t rootLeaves = nodes[0].leaves & 0x3FFFFFFFu;
    if (rootLeaves == 0u)
    {
        const QuantBuild q = buildQuantBvh(nodes, numNodes, numPrimitives, depth);
        if (!q.ok) return out;
        const WideBuild w = buildWideBvh(nodes, numNodes, q);
        if (!w.ok) return out;
        out.nodes = w.nodes; out.gate = q.gate;
        memcpy(out.base, q.base, sizeof(q.base)); memcpy(out.step, q.step, sizeof(q.step)); memcpy(out.bound, q.bound, sizeof(q.bound));
        out.ok = true;
        return out;
    }
    if (rootLeaves > 2u || (uint64_t)nodes[0].childIndex + rootLeaves > numPrimitives) return out;
    // the grid of buildQuantBvh over the root's box; the leaf's stored box a full step outside the exact one
    const RtNode& root = nodes[0];
    float largest = 0.0f;
    for (int a = 0; a < 3; ++a) { if (!(root.min[a] <= root.max[a]) || !std::isfinite(root.min[a]) || !std::isfinite(root.max[a])) return out; largest = fmaxf(largest, root.max[a] - root.min[a]); }
    uint32_t v[6];
    for (int a = 0; a < 3; ++a)
    {
        out.step[a] = fmaxf(root.max[a] - root.min[a], 1e-6f * fmaxf(largest, 1e-30f)) / (RT_QUANT_GRID - 8.0f);
        out.base[a] = root.min[a] - 4.0f * out.step[a];
        out.bound[a] = fmaxf(fabsf(root.min[a]), fabsf(root.max[a])) + 8.0f * out.step[a];
        if (!(out.step[a] > 0.0f) || !std::isfinite(out.step[a])) return out;
        v[a] = 1u; v[3 + a] = 65534u;    // planes at base + step and base + 65534 step: more than a step outside [min, max] on both sides
        if ((double)out.base[a] + (double)out.step[a] > (double)root.min[a] - (double)out.step[a] || (double)out.base[a] + 65534.0 * (double)out.step[a] < (double)root.max[a] + (double)out.step[a]) return out;
    }
    const uint32_t d0 = v[0] | (v[1] << 16), d1 = v[2] | (v[3] << 16), d2 = v[4] | (v[5] << 16);
    out.nodes.assign(4u, make_float4(0.0f, 0.0f, 0.0f, __builtin_bit_cast(float, (uint32_t)RT_WIDE_EMPTY)));
    out.nodes[0] = make_float4(__builtin_bit_cast(float, d0), __builtin_bit_cast(float, d1), __builtin_bit_cast(float, d2), __builtin_bit_cast(float, root.childIndex | (rootLeaves << RT_NODE_LEAVES_SHIFT)));
    out.gate.assign((size_t)2 * numPrimitives, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    out.gate[2u * (size_t)root.childIndex] = make_float4(root.min[0], root.min[1], root.min[2], 0.0f);
    out.gate[2u * (size_t)root.childIndex + 1u] = make_float4(root.max[0], root.max[1], root.max[2], 0.0f);
    out.ok = true;
    return out;
}
#endif   // RT_HOST_BUILDERS
