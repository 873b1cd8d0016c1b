// rt_trace_binary.inl -- the reference's own walk: binary BVH, near child first (k_trace), and the cooperative search for degenerate
// closest-hit rays (k_trace_monster).  Included by rt_trace.hip and rt_tail.hip.
// ONE persistent traversal kernel per bounce: it serves the closest-hit rays of the paths alive at bounce k
// (Scene::Traverse, Scene.cpp:219-243) AND the NEE shadow rays queued by the shade of bounce k-1
// (Scene::Traverse_Shadow, Scene.cpp:245-261; PathTracerMIS.cpp:81-96).  The two sets are independent, and
// serving them from one work cursor halves the number of launches whose tail (a few long rays keeping the
// grid alive) would otherwise be paid twice.  An occluded NEE request is marked by tmax = -1; the contribution
// is folded in later by resolvePendingLightSamples.
// Occupancy: 5 waves per SIMD (<= 96 VGPRs; the register allocator gets there without spilling once the world ray and
// the hit record are not carried) x 24.6 KB of LDS stack per block = 5 blocks per CU.  The interior loop waits ~800 ns
// per dependent node fetch, so every extra wave is throughput.
// The loop itself is a device function: k_trace runs it over the launch's queues with its own LDS; k_trace_wide runs it over the block's own
// list of undecided rays when its walk is done (rt_trace_wide.inl), and k_tail between its block-local trace and shade phases (rt_tail.hip).
// `queue`, `shadowQueue`, the counts and `cursor` may live in LDS (generic pointers); `sharingWaves` = the waves that claim from `cursor`.
template <int kStack, bool kCount>
RT_DEV void traceBinaryLoop(const RtSceneDesc& scene, const Paths& paths, const uint32_t* queue, const uint32_t* queueCount,
                            const uint32_t* shadowQueue, const uint32_t* shadowCount, uint32_t* cursor, unsigned long long* counters, const TravTuning& tune,
                            uint32_t* sStack, uint32_t* sDensePrefix, uint32_t sharingWaves)
{
    const LdsStack stack = { sStack + threadIdx.x, RT_BLOCK };
    Counters cnt; zeroCounters(cnt);
    if (tune.denseCounts) { denseLoadPrefix(tune.denseCounts, sDensePrefix); __syncthreads(); }
    const uint32_t numClosest = tune.denseCounts ? sDensePrefix[RT_DENSE_SHARDS] : (queueCount ? *queueCount : 0u);
    const uint32_t count = numClosest + (shadowCount ? *shadowCount : 0u);
    TravState s; s.mode = TRAV_DONE; s.shadow = false;
    uint32_t slot = 0, light = 0;
    bool have = false, exhausted = false;
    // chunk size: large enough to make global atomics rare, small enough to keep the tail of the launch balanced
    uint32_t chunkSize = count / (sharingWaves * 4u);
    chunkSize = chunkSize < 64u ? 64u : (chunkSize > 1024u ? 1024u : chunkSize);
    WaveChunk chunk = { 0u, 0u };
    // the world ray of the lane's current request, rebuilt from the path state exactly as at refill time
    auto loadWorldRay = [&]() -> Ray
    {
        if (s.shadow)
        {
            const float4 origin = prec(paths, R_SH_P, slot), dirTmax = pshadow(paths, light, 0, slot);
            Ray shadowRay = makeRay(V4(origin.x, origin.y, origin.z, 0.0f), V4(dirTmax.x, dirTmax.y, dirTmax.z, 0.0f));
            shadowRay.origin = shadowRay.origin + shadowRay.dir * tune.shadowOffset;   // PathTracerMIS.cpp:86
            return shadowRay;
        }
        const float4 origin = prec(paths, R_ORIGIN, slot), dir = prec(paths, R_DIR, slot);
        return makePathRay(origin, dir, ubits(origin.w) & 0xFFu);
    };
    // HitPoint of a closest-hit ray, written through at every accepted hit (T6: objectId, subObjectId, distance, u, v)
    auto onHit = [&](uint32_t objectId, uint32_t subObjectId, float distance, float u, float v)
    {
        prec(paths, R_HIT, slot) = f4(fbits(objectId), fbits(subObjectId), distance, u);
        prec(paths, R_SAMPLER, slot).x = v;
    };
    // Drain-phase work sharing for any-hit rays.  When the queue is used up, a launch lasts as long as its longest
    // ray, and some NEE rays are very long: a direction that is exactly a coordinate axis makes two of the three slab
    // tests meaningless in the reference's box * invDir - origin * invDir formulation (inf - inf), and such a ray
    // walks every node whose remaining slab it overlaps -- tens of thousands of steps, alone in its wave.  Occlusion
    // is an OR over subtrees, so the deferred subtrees on a shadow ray's stack can be searched by other lanes: a
    // lane with nothing to do takes the OLDEST deferred node (the largest subtree) of a busy shadow ray in its wave
    // and searches it as a ray of its own with the same request id; whoever finds an occluder marks the request.
    // Closest-hit rays are never split (their box culling and tie-breaking depend on the visiting order), and the
    // counting variant does not split at all, so the intersection counters stay those of the serial traversal.
    const bool splitShadowRays = !kCount;
    const bool singleMeshLevel = scene.numObjects == 1u;   // bypass scenes: a mesh level is all a ray has (degenerate closest-hit rays can be handed over)
    // Single-mesh scenes (Scene::Traverse's one-object bypass, Scene.cpp:231-235, into MeshShape::Traverse): everything a ray
    // needs to enter the mesh is the same for all rays, so it is fetched ONCE per wave (uniform -> scalar registers) instead
    // of through three dependent loads (object -> mesh -> root node) behind every refill.
    bool bypassMesh = false;
    // (the object's inverse transform is fetched per refill through the constant address space -- scalar loads of a uniform address -- instead of living in
    //  sixteen scalar registers across the loop, whose header spilled two dozen of them into vector lanes every iteration)
    typedef const __attribute__((address_space(4))) float* ConstF;
    const ConstF bypassInvTransformWords = (ConstF)(uintptr_t)scene.objects[0].invTransform;
    const RtNode* bypassNodes = nullptr; uint32_t bypassTriBase = 0, bypassRoot = 0;
    if (scene.numObjects == 1u && scene.objects[0].objectKind == RT_OBJECT_SHAPE && scene.objects[0].shapeKind == RT_SHAPE_MESH)
    {
        const RtMesh& mesh = scene.meshes[scene.objects[0].meshIndex];
        if (mesh.numNodes != 0u)
        {
            bypassMesh = true;
            bypassNodes = scene.meshNodes + mesh.firstNode;
            bypassTriBase = mesh.firstTriangle;
            bypassRoot = packNode(bypassNodes[0].childIndex, bypassNodes[0].leaves);
        }
    }
    uint32_t drainIterations = 0, closestDrain = 0;
    for (;;)
    {
        const bool interior = have && travIsInterior(s);
        const bool other = have && !interior;
        const unsigned long long mI = __ballot(interior), mO = __ballot(other);
        const uint32_t nIdle = 64u - (uint32_t)__popcll(mI) - (uint32_t)__popcll(mO);
        // A closest-hit ray that is still running long after the queue ran dry is a degenerate one (an exactly axis-parallel
        // direction turns two of three slab tests into inf - inf, and the ray walks most of the tree alone -- tens of
        // milliseconds).  Such rays cannot be split like any-hit rays (the visiting order decides ties), so they are handed
        // to k_trace_monster, which finds the same hit cooperatively.  Single-mesh scenes, counters off.
        if (splitShadowRays && singleMeshLevel && tune.overflowQueue && exhausted && ++closestDrain > tune.abortClosestAfter)
        {
            const bool abortLane = have && !s.shadow;
            const unsigned long long mAbort = __ballot(abortLane);
            if (mAbort != 0ull)
            {
                const uint32_t lane = threadIdx.x & 63u;
                uint32_t base = 0;
                if (lane == (uint32_t)(__ffsll((long long)mAbort) - 1)) base = atomicAdd(tune.overflowCount, (uint32_t)__popcll(mAbort));
                base = (uint32_t)__shfl((int)base, __ffsll((long long)mAbort) - 1);
                if (abortLane) { tune.overflowQueue[base + (uint32_t)__popcll(mAbort & ((1ull << lane) - 1ull))] = slot; have = false; }
                continue;
            }
        }
        // Idle lanes get work from the queue (refill) or, once the queue is used up, from a busy shadow ray of the wave
        const bool refill = !exhausted && (nIdle == 64u || nIdle >= tune.refillMinIdle);
        unsigned long long mDonors = 0ull;
        const bool canDonate = have && s.shadow && s.mode == TRAV_MESH && s.stackSize > s.levelBase;
        if (splitShadowRays && exhausted && nIdle != 0u && ++drainIterations > (tune.splitAfter ? tune.splitAfter : RT_SPLIT_AFTER)) mDonors = __ballot(canDonate);
        if (refill || mDonors != 0ull)
        {
            uint32_t request = 0xFFFFFFFFu, donated = 0u, meshContextObject = 0u;
            bool shadowRequest = true;
            if (refill)
            {
                if (chunk.next >= chunk.end)
                {
                    waveClaimChunk(chunk, cursor, chunkSize, count);
                    if (chunk.next >= chunk.end) { exhausted = true; continue; }
                }
                const uint32_t idx = waveTake(!have, chunk);
                if (idx != 0xFFFFFFFFu)
                {
                    shadowRequest = idx >= numClosest;
                    request = shadowRequest ? shadowQueue[idx - numClosest] : (tune.denseCounts ? denseLiveSlot(sDensePrefix, tune.denseShardCapacity, idx) : queue[idx]);
                    if (shadowRequest) cnt.c[C_SHADOW]++;
                }
            }
            else
            {
                // the k-th idle lane takes the OLDEST deferred node (stack bottom of the level) of the k-th donor
                const unsigned long long mIdle = __ballot(!have);
                const uint32_t lane = threadIdx.x & 63u;
                const unsigned long long below = (1ull << lane) - 1ull;
                const uint32_t nDonors = (uint32_t)__popcll(mDonors), nTakers = (uint32_t)__popcll(mIdle);
                const uint32_t pairs = nDonors < nTakers ? nDonors : nTakers;
                const bool donate = canDonate && (uint32_t)__popcll(mDonors & below) < pairs;
                const uint32_t takerRank = (uint32_t)__popcll(mIdle & below);
                const bool take = !have && takerRank < pairs;
                uint32_t src = lane;
                if (take)
                {
                    unsigned long long m = mDonors;
                    for (uint32_t k = 0; k < takerRank; ++k) m &= m - 1ull;
                    src = (uint32_t)__ffsll((long long)m) - 1u;
                }
                uint32_t entry = 0u;
                if (donate)
                {
                    // the oldest entry of the mesh level leaves the donor's stack; the newest one takes its place (any-hit rays: the order
                    // of the remaining subtrees is free), so that nothing stale is left for the level below in a two-level scene
                    entry = stack.base[s.levelBase * stack.stride];
                    --s.stackSize;
                    stack.base[s.levelBase * stack.stride] = stack.base[s.stackSize * stack.stride];
                }
                const uint32_t donorRequest = (uint32_t)__shfl((int)(light * paths.capacity + slot), (int)src);
                donated = (uint32_t)__shfl((int)entry, (int)src);
                if (take) request = donorRequest;
                // two-level scenes: the taker continues inside the DONOR'S mesh (it rebuilds the local ray from the request like a mesh entry does)
                if (!bypassMesh) meshContextObject = (uint32_t)__shfl((int)s.objectId, (int)src);
            }
            if (request != 0xFFFFFFFFu)
            {
                float maxDistance = __uint_as_float(0x7f800000u);
                s.shadow = shadowRequest;
                if (!shadowRequest) slot = request;
                else
                {
                    light = request / paths.capacity; slot = request - light * paths.capacity;
                    maxDistance = pshadow(paths, light, 0, slot).w;   // hitPoint.distance = illuminateResult.distance * 0.999f
                }
                have = true;
                if (bypassMesh)
                {
                    // = travBegin + the object step of travStepOther for the one mesh object
                    M4 bypassInvTransform;
                    for (int r = 0; r < 4; ++r) bypassInvTransform.r[r] = V4(bypassInvTransformWords[4 * r], bypassInvTransformWords[4 * r + 1], bypassInvTransformWords[4 * r + 2], bypassInvTransformWords[4 * r + 3]);
                    s.ray = transformRayUnsafe(bypassInvTransform, loadWorldRay());
                    s.nanFree = rayIsNaNFree(s.ray);
                    s.hitDistance = maxDistance;
                    s.stackSize = 0; s.levelBase = 0; s.leafNext = 1; s.leafEnd = 1; s.objectId = 0; s.triBase = bypassTriBase;
                    s.occluded = false; s.nodes = bypassNodes; s.cur = bypassRoot; s.mode = TRAV_MESH;
                }
                else if (refill)
                {
                    travBegin(s, scene, loadWorldRay(), maxDistance, s.shadow);
                    // other single-object scenes start at the object loop (BVH bypass): enter the object right away instead
                    // of queueing for the "other" phase
                    if (!travIsInterior(s) && s.mode != TRAV_DONE) travStepOther<kCount>(s, scene, stack, cnt, loadWorldRay, onHit);
                }
                // a taken subtree: same ray, same mesh, but only the donated node instead of the root
                if (!refill && bypassMesh) { if (s.mode == TRAV_MESH) s.cur = donated; }
                else if (!refill)
                {
                    const RtObject& obj = scene.objects[meshContextObject];
                    const RtMesh& mesh = scene.meshes[obj.meshIndex];
                    s.ray = transformRayUnsafe(loadM4(obj.invTransform), loadWorldRay());   // = the donor's local ray (Scene::Traverse_Object_Shadow's)
                    s.nanFree = rayIsNaNFree(s.ray); s.hitDistance = maxDistance;
                    s.stackSize = 0; s.levelBase = 0; s.leafNext = 0; s.leafEnd = 0;   // nothing above the mesh: when its level is exhausted the ray is done
                    s.objectId = meshContextObject; s.triBase = mesh.firstTriangle; s.nodes = scene.meshNodes + mesh.firstNode;
                    s.occluded = false; s.cur = donated; s.mode = TRAV_MESH;
                }
            }
            continue;
        }
        if ((mI | mO) == 0ull) break;
        if (mI != 0ull && (uint32_t)__popcll(mO) < tune.otherMinLanes)
        {
            // INTERIOR PHASE as a tight inner loop: only (cur, stackSize) change per step, everything else of the lane
            // state is loop invariant, so the wave keeps stepping without re-evaluating the refill logic until enough
            // lanes wait at leaves / level exits.  Hardware min/max unless some lane's ray could produce a NaN in a
            // slab test (axis-parallel rays).
            bool in = interior;
            if (__all(!have || s.nanFree))
            {
                for (;;)
                {
                    if (in) travStepInterior<kCount, false>(s, stack, cnt);
                    in = in && (s.cur >> RT_NODE_LEAVES_SHIFT) == 0u;   // the mode does not change in here
                    const unsigned long long m = __ballot(in);
                    if (m == 0ull || 64u - nIdle - (uint32_t)__popcll(m) >= tune.otherMinLanes) break;
                }
            }
            else
            {
                for (;;)
                {
                    if (in) travStepInterior<kCount, true>(s, stack, cnt);
                    in = in && (s.cur >> RT_NODE_LEAVES_SHIFT) == 0u;   // the mode does not change in here
                    const unsigned long long m = __ballot(in);
                    if (m == 0ull || 64u - nIdle - (uint32_t)__popcll(m) >= tune.otherMinLanes) break;
                }
            }
        }
        else if (other)
        {
            if (s.mode != TRAV_DONE) travStepOther<kCount>(s, scene, stack, cnt, loadWorldRay, onHit);
            if (s.mode == TRAV_DONE)
            {
                if (s.shadow)
                {
                    if (s.occluded) pshadow(paths, light, 0, slot).w = -1.0f;   // unoccluded requests are tallied when they are resolved
                }
                else
                {
                    // nothing was hit: HitPoint stays {RT_INVALID_OBJECT, distance = FLT_MAX-ish infinity} (HitPoint.h:14-51)
                    if (s.hitDistance == __uint_as_float(0x7f800000u)) prec(paths, R_HIT, slot) = f4(fbits(RT_INVALID_OBJECT), fbits(0u), s.hitDistance, 0.0f);
                }
                have = false;
            }
        }
    }
    flushCounters(cnt, counters);
}

template <int kStack, bool kCount>
__global__ void __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(kStack <= 24 ? 5 : 1))) k_trace(const RtSceneDesc scene, const Paths paths,
                                                    const uint32_t* __restrict__ queue, const uint32_t* __restrict__ queueCount,
                                                    const uint32_t* __restrict__ shadowQueue, const uint32_t* __restrict__ shadowCount,
                                                    uint32_t* __restrict__ cursor, unsigned long long* counters, const TravTuning tune)
{
    __shared__ uint32_t sStack[kStack * RT_BLOCK];
    __shared__ uint32_t sDensePrefix[RT_DENSE_SHARDS + 1u];
    uint32_t blocks = gridDim.x;
    if (tune.baseBlocks != 0u && tune.baseBlocks < gridDim.x && !tune.denseCounts)
    {
        // a re-trace launch is issued with a full grid; for its usual few thousand rays only the first `baseBlocks` blocks stay (block-uniform decision)
        const uint32_t requests = (queueCount ? *queueCount : 0u) + (shadowCount ? *shadowCount : 0u);
        if (requests <= tune.fullGridAbove) { if (blockIdx.x >= tune.baseBlocks) return; blocks = tune.baseBlocks; }
    }
    traceBinaryLoop<kStack, kCount>(scene, paths, queue, queueCount, shadowQueue, shadowCount, cursor, counters, tune, sStack, sDensePrefix, blocks * (RT_BLOCK / 64u));
}

#ifndef RT_TRACE_FUNCTIONS_ONLY   // (rt_tail.hip takes the walk above and not this kernel)
// Closest hit of a degenerate ray, found by a whole block.  The sequential result is "smallest distance; among equal distances
// the triangle visited first".  The smallest distance does not depend on the order, so it is searched in parallel: a shared LDS
// stack of nodes, every thread pops one, tests the two children with the reference's slab test (same NaN behaviour, culling with
// <= the best distance so far so that every triangle AT the final distance is still visited) or the leaf's triangles, and
// publishes hits through a 64-bit atomic min of (distance bits, triangle).  If two different triangles ever report the same
// distance, or the stack overflows, one thread repeats the search sequentially with the ordinary state machine (exactly the
// reference's order); otherwise the winner is unique and its record (distance, u, v from the same Moller-Trumbore evaluation)
// is what the sequential traversal would have written.  Single-mesh scenes (the bypass path of k_trace).
__global__ void __launch_bounds__(RT_MONSTER_BLOCK) k_trace_monster(const RtSceneDesc scene, const Paths paths, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ queueCount)
{
    __shared__ uint32_t sStack[RT_MONSTER_STACK];
    __shared__ uint32_t sTop, sTaken, sFlags;            // sFlags: 1 = tie, 2 = stack overflow
    __shared__ unsigned long long sBest;
    __shared__ uint32_t sSerialStack[64];
    const uint32_t count = *queueCount;
    if (count == 0u) return;
    const RtMesh& mesh = scene.meshes[scene.objects[0].meshIndex];
    const RtNode* nodes = scene.meshNodes + mesh.firstNode;
    const RtTriangle* tris = scene.triangles + mesh.firstTriangle;
    const M4 invTransform = loadM4(scene.objects[0].invTransform);
    for (uint32_t q = blockIdx.x; q < count; q += gridDim.x)
    {
        const uint32_t slot = queue[q];
        const float4 origin = prec(paths, R_ORIGIN, slot), dir = prec(paths, R_DIR, slot);
        const Ray ray = transformRayUnsafe(invTransform, makePathRay(origin, dir, ubits(origin.w) & 0xFFu));
        if (threadIdx.x == 0)
        {
            sStack[0] = packNode(nodes[0].childIndex, nodes[0].leaves); sTop = 1u; sFlags = 0u;
            sBest = ((unsigned long long)0x7f800000u << 32) | 0xFFFFFFFFull;   // +inf, no triangle
        }
        __syncthreads();
        for (;;)
        {
            const uint32_t n = sTop;
            __syncthreads();
            if (n == 0u || sFlags != 0u) break;
            const uint32_t take = n < RT_MONSTER_BLOCK ? n : RT_MONSTER_BLOCK;
            uint32_t entry = 0u;
            const bool active = threadIdx.x < take;
            if (active) entry = sStack[n - 1u - threadIdx.x];
            if (threadIdx.x == 0) sTop = n - take;
            __syncthreads();
            if (active)
            {
                const float best = __uint_as_float((uint32_t)(sBest >> 32));
                const uint32_t numLeaves = entry >> RT_NODE_LEAVES_SHIFT, first = entry & RT_NODE_CHILD_MASK;
                if (numLeaves == 0u)
                {
                    const NodePair np = loadNodePair(nodes, first);
                    float distanceA, distanceB;
                    const bool hitA = intersectBoxRay(ray, V4(np.a0.x, np.a0.y, np.a0.z, 0.0f), V4(np.a1.x, np.a1.y, np.a1.z, 0.0f), distanceA) && distanceA <= best &&
                                      boxNearDegenerateAxes(ray, np.a0.x, np.a0.y, np.a0.z, np.a1.x, np.a1.y, np.a1.z);
                    const bool hitB = intersectBoxRay(ray, V4(np.b0.x, np.b0.y, np.b0.z, 0.0f), V4(np.b1.x, np.b1.y, np.b1.z, 0.0f), distanceB) && distanceB <= best &&
                                      boxNearDegenerateAxes(ray, np.b0.x, np.b0.y, np.b0.z, np.b1.x, np.b1.y, np.b1.z);
                    const uint32_t pushes = (hitA ? 1u : 0u) + (hitB ? 1u : 0u);
                    if (pushes != 0u)
                    {
                        const uint32_t at = atomicAdd(&sTop, pushes);
                        if (at + pushes > RT_MONSTER_STACK) atomicOr(&sFlags, 2u);
                        else
                        {
                            uint32_t k = at;
                            if (hitA) sStack[k++] = packNode(__float_as_uint(np.a0.w), __float_as_uint(np.a1.w));
                            if (hitB) sStack[k] = packNode(__float_as_uint(np.b0.w), __float_as_uint(np.b1.w));
                        }
                    }
                }
                else
                {
                    for (uint32_t i = 0; i < numLeaves; ++i)
                    {
                        const uint32_t triangleIndex = first + i;
                        V4 v0, e1, e2; loadTriangle(tris + triangleIndex, v0, e1, e2);
                        float u, v, dist;
                        if (intersectTriangleRay(ray, v0, e1, e2, u, v, dist) && dist <= best)
                        {
                            const unsigned long long key = ((unsigned long long)__float_as_uint(dist) << 32) | triangleIndex;
                            const unsigned long long old = atomicMin(&sBest, key);
                            if ((uint32_t)(old >> 32) == __float_as_uint(dist) && (uint32_t)old != triangleIndex) atomicOr(&sFlags, 1u);
                        }
                    }
                }
            }
            __syncthreads();
        }
        if (threadIdx.x == 0)
        {
            if (sFlags == 0u)
            {
                const uint32_t triangleIndex = (uint32_t)sBest;
                if (triangleIndex == 0xFFFFFFFFu) prec(paths, R_HIT, slot) = f4(fbits(RT_INVALID_OBJECT), fbits(0u), __uint_as_float(0x7f800000u), 0.0f);
                else
                {
                    V4 v0, e1, e2; loadTriangle(tris + triangleIndex, v0, e1, e2);
                    float u = 0.0f, v = 0.0f, dist = 0.0f;
                    (void)intersectTriangleRay(ray, v0, e1, e2, u, v, dist);
                    prec(paths, R_HIT, slot) = f4(fbits(0u), fbits(triangleIndex), dist, u);
                    prec(paths, R_SAMPLER, slot).x = v;
                }
            }
            else
            {
                // the sequential traversal, exactly as k_trace's bypass path runs it
                Counters cnt; zeroCounters(cnt);
                const LdsStack stack = { sSerialStack, 1u };
                TravState s;
                s.ray = ray; s.nanFree = false; s.shadow = false; s.occluded = false;
                s.hitDistance = __uint_as_float(0x7f800000u);
                s.stackSize = 0; s.levelBase = 0; s.leafNext = 1; s.leafEnd = 1; s.objectId = 0; s.triBase = mesh.firstTriangle;
                s.nodes = nodes; s.cur = packNode(nodes[0].childIndex, nodes[0].leaves); s.mode = TRAV_MESH;
                auto reloadWorldRay = [&]() -> Ray { return ray; };
                auto onHit = [&](uint32_t objectId, uint32_t subObjectId, float distance, float u, float v)
                {
                    prec(paths, R_HIT, slot) = f4(fbits(objectId), fbits(subObjectId), distance, u);
                    prec(paths, R_SAMPLER, slot).x = v;
                };
                while (s.mode != TRAV_DONE)
                {
                    if (travIsInterior(s)) travStepInterior<false, true>(s, stack, cnt);
                    else travStepOther<false>(s, scene, stack, cnt, reloadWorldRay, onHit);
                }
                if (s.hitDistance == __uint_as_float(0x7f800000u)) prec(paths, R_HIT, slot) = f4(fbits(RT_INVALID_OBJECT), fbits(0u), s.hitDistance, 0.0f);
            }
        }
        __syncthreads();
    }
}
#endif   // RT_TRACE_FUNCTIONS_ONLY
