// rt_trace_wide.inl -- the default traversal kernel of single-mesh scenes: FOUR LANES PER RAY over a 4-wide BVH collapsed from the
// reference's binary tree.  Included by rt_kernels.hip.
//
// Why.  The binary-tree kernel (k_trace, rt_device_traverse.h) spends its time in the vector L1: every lane fetches its own 64-byte
// node pair with four 16-byte loads, and a divergent 16-byte access costs the texture-cache pipeline one cycle however little of the
// line it uses -- 0.7 accesses per clock and CU measured, against a ceiling of one, with the memory behind it mostly idle
// (profiles/r02_diag0_*).  Here a ray owns a QUAD of lanes and a node has four children:
//   * lane j of the quad fetches child j: {min.xyz, child reference} and {max.xyz, -}: the quad's four 16-byte loads are 64 contiguous
//     bytes = ONE L1 access, two per node visit instead of four, and a 4-wide tree needs about half the visits of the binary one;
//   * a leaf's two triangles are stored lane-interleaved, {v0, v0', e1, e1', e2, e2'}: three accesses per leaf;
//   * the slab tests of the four children and the Moeller-Trumbore tests of the two triangles run in parallel on the quad's lanes;
//     ordering and the choice of the next node go through DPP quad permutes, only the deferred children go through the LDS stack;
//   * the top levels of the tree (breadth-first layout: the first RT_WIDE_LDS_NODES nodes) are staged in LDS by every block.
//
// Exactness.  The reference walks the BINARY tree near child first and keeps the first of equally distant hits
// (Traversal_Single.h:16-96, MeshShape.cpp:134-168), so its result could depend on the visiting order.  It does not, except in
// near-ties: (1) every child box of the wide tree IS a box of the binary tree and is tested with the same arithmetic; the slab test is
// monotone in the box planes (fma and min/max are), so a ray that passes a node's box passes the boxes of all its ancestors with
// a smaller entry distance -- skipping every other level changes nothing for rays without NaNs; (2) every triangle is tested with the
// same arithmetic, so each candidate hit has the same (t, u, v) in both walks; (3) the walks can only pick different candidates when
// two of them are closer together than the disagreement between a box's entry distance and its triangle's hit distance.  The kernel
// therefore culls with a slack (near < best + 2 tol), tracks the SECOND smallest candidate distance, and a ray whose runner-up is
// within tol of its best (tol = 16 ulps of the largest term of its slab tests) is not trusted: it goes to the exact queue and is
// traced again by the binary-tree kernel in the reference's order, as do rays with a zero direction component (their slab tests
// produce NaNs, which the reference's min/max operand order resolves in its own way).  Any-hit rays need none of this: occlusion is
// an OR over the same candidate set.  The box / triangle test counters of the reference only make sense for its own walk: with the
// intersection counters on, the binary-tree kernel runs alone.

#define RT_WIDE_EMPTY 0xFFFFFFFFu
#define RT_WIDE_DONE 0xFFFFFFFEu
#define RT_WIDE_LDS_NODES 85u    // 1 + 4 + 16 + 64 nodes = the top four levels when they are full (10.9 KB)

struct WideBvh
{
    const float4* nodes;     // 8 float4 per node: lo[4] = {min.xyz, ref}, hi[4] = {max.xyz, 0}; breadth-first order
    const float4* leaves;    // 6 float4 per leaf: v0, v0', e1, e1', e2, e2' (w of v0 / v0': index of the leaf's first triangle)
    uint32_t numNodes, numLeaves;
    uint32_t stackNeed;      // deepest stack a traversal can build
    float bound[3];          // largest |coordinate| of the mesh per axis (for the per-ray tolerance)
};

struct WideTuning
{
    uint32_t refillMinIdle, otherMinLanes;   // in lanes, as TravTuning
    float shadowOffset;
    uint32_t* exactQueue; uint32_t* exactCount;               // closest-hit rays handed to the binary-tree kernel
    uint32_t* exactShadowQueue; uint32_t* exactShadowCount;   // any-hit requests handed to it
};

// DPP quad permutes: lane j of every quad reads lane perm[j] of its quad
#define RT_QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
template <int kCtrl> RT_DEV float quadPermF(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), kCtrl, 0xF, 0xF, true)); }
template <int kCtrl> RT_DEV uint32_t quadPermU(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, kCtrl, 0xF, 0xF, true); }

template <int kStack, bool kLdsTop>
__global__ void __launch_bounds__(RT_BLOCK) k_trace_wide(const RtSceneDesc scene, const WideBvh bvh, const Paths paths,
                                                         const uint32_t* __restrict__ queue, const uint32_t* __restrict__ queueCount,
                                                         const uint32_t* __restrict__ shadowQueue, const uint32_t* __restrict__ shadowCount,
                                                         uint32_t* __restrict__ cursor, unsigned long long* counters, const WideTuning tune)
{
    __shared__ uint32_t sStack[kStack * (RT_BLOCK / 4)];
    __shared__ float4 sTop[kLdsTop ? RT_WIDE_LDS_NODES * 8u : 1u];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t j = threadIdx.x & 3u;                 // child / triangle this lane tests
    uint32_t* const stack = sStack + (threadIdx.x >> 2); // column of this quad; entry e at stack[e * (RT_BLOCK / 4)]
    const uint32_t tieMask = (j == 1u ? 4u : 0u) | (j == 2u ? 6u : 0u) | (j == 3u ? 7u : 0u);   // bit r-1: lane (j + r) & 3 sorts before lane j on equal keys
    const uint32_t ldsNodes = bvh.numNodes < RT_WIDE_LDS_NODES ? bvh.numNodes : RT_WIDE_LDS_NODES;
    if (kLdsTop)
    {
        for (uint32_t i = threadIdx.x; i < ldsNodes * 8u; i += RT_BLOCK) sTop[i] = bvh.nodes[i];
        __syncthreads();
    }
    const uint32_t numClosest = queueCount ? *queueCount : 0u;
    const uint32_t count = numClosest + (shadowCount ? *shadowCount : 0u);
    const M4 invTransform = loadM4(scene.objects[0].invTransform);
    const uint32_t triBase = 0u;   // triangle indices of the leaf records are mesh-relative, as HitPoint::subObjectId is

    // per-quad state, replicated in the quad's four lanes
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, ix = 0, iy = 0, iz = 0, qx = 0, qy = 0, qz = 0;   // local ray: origin, dir, invDir, originDivDir
    float best = 0, second = 0, tol = 0;
    uint32_t cur = RT_WIDE_DONE, sp = 0, slot = 0, light = 0;
    bool have = false, shadow = false, occluded = false, exhausted = false;
    uint32_t numRetraced = 0, numShadowRays = 0;

    uint32_t chunkSize = count / (gridDim.x * (RT_BLOCK / 64u) * 4u);
    chunkSize = chunkSize < 16u ? 16u : (chunkSize > 256u ? 256u : chunkSize);
    WaveChunk chunk = { 0u, 0u };
    const float inf = __uint_as_float(0x7f800000u);

    for (;;)
    {
        const bool interior = have && (cur >> RT_NODE_LEAVES_SHIFT) == 0u;
        const bool other = have && !interior;      // at a leaf, or finished
        const unsigned long long mI = __ballot(interior), mO = __ballot(other);
        const uint32_t nIdle = 64u - (uint32_t)__popcll(mI) - (uint32_t)__popcll(mO);
        if (!exhausted && (nIdle == 64u || nIdle >= tune.refillMinIdle))
        {
            // ---- refill: idle quads take the next rays of the wave's chunk ----
            if (chunk.next >= chunk.end)
            {
                waveClaimChunk(chunk, cursor, chunkSize, count);
                if (chunk.next >= chunk.end) { exhausted = true; continue; }
            }
            const unsigned long long mWant = __ballot(!have && j == 0u);
            const uint32_t rank = (uint32_t)__popcll(mWant & ((1ull << (lane & ~3u)) - 1ull));
            const uint32_t available = chunk.end - chunk.next;
            const uint32_t idx = (!have && rank < available) ? chunk.next + rank : 0xFFFFFFFFu;
            const uint32_t wanted = (uint32_t)__popcll(mWant);
            chunk.next += wanted < available ? wanted : available;
            if (idx != 0xFFFFFFFFu)
            {
                shadow = idx >= numClosest;
                const uint32_t request = shadow ? shadowQueue[idx - numClosest] : (queue ? queue[idx] : idx);
                Ray world;
                float maxDistance = inf;
                if (shadow)
                {
                    light = request / paths.capacity; slot = request - light * paths.capacity;
                    const float4 origin = prec(paths, R_SH_P, slot), dirTmax = pshadow(paths, light, 0, slot);
                    world = makeRay(V4(origin.x, origin.y, origin.z, 0.0f), V4(dirTmax.x, dirTmax.y, dirTmax.z, 0.0f));
                    world.origin = world.origin + world.dir * tune.shadowOffset;   // PathTracerMIS.cpp:86
                    maxDistance = dirTmax.w;                                       // hitPoint.distance = illuminateResult.distance * 0.999f
                }
                else
                {
                    slot = request; light = 0u;
                    const float4 origin = prec(paths, R_ORIGIN, slot), dir = prec(paths, R_DIR, slot);
                    world = makePathRay(origin, dir, ubits(origin.w) & 0xFFu);
                }
                const Ray local = transformRayUnsafe(invTransform, world);   // MeshShape is entered in object space, Scene.cpp:128-145
                if (!rayIsNaNFree(local))
                {
                    // a zero direction component: the reference's min/max operand order decides what its NaNs do -- its own walk only
                    if (j == 0u)
                    {
                        if (shadow) tune.exactShadowQueue[atomicAdd(tune.exactShadowCount, 1u)] = request;
                        else tune.exactQueue[atomicAdd(tune.exactCount, 1u)] = slot;
                        numRetraced++;
                    }
                }
                else
                {
                    ox = local.origin.x; oy = local.origin.y; oz = local.origin.z; dx = local.dir.x; dy = local.dir.y; dz = local.dir.z;
                    ix = local.invDir.x; iy = local.invDir.y; iz = local.invDir.z; qx = local.originDivDir.x; qy = local.originDivDir.y; qz = local.originDivDir.z;
                    // 16 ulps of the largest magnitude a slab test of this ray can produce
                    const float m = fmaxf(fmaxf(fabsf(qx) + bvh.bound[0] * fabsf(ix), fabsf(qy) + bvh.bound[1] * fabsf(iy)), fabsf(qz) + bvh.bound[2] * fabsf(iz));
                    tol = shadow ? 0.0f : m * 1.9073486328125e-06f;   // 2^-19
                    best = maxDistance; second = inf; occluded = false;
                    if (shadow && j == 0u) numShadowRays++;   // (a request handed to the binary-tree kernel is counted there)
                    sp = 0u; cur = 0u;   // the root's children are tested, not its own box (Traversal_Single.h:22-31)
                    have = true;
                }
            }
            continue;
        }
        if ((mI | mO) == 0ull) break;
        if (mI != 0ull && (uint32_t)__popcll(mO) < tune.otherMinLanes)
        {
            // ---- interior phase: four slab tests per quad and step, until enough quads wait at a leaf or are finished ----
            bool in = interior;
            for (;;)
            {
                if (in)
                {
                    float4 lo, hi;
                    if (kLdsTop && cur < ldsNodes) { lo = sTop[cur * 8u + j]; hi = sTop[cur * 8u + 4u + j]; }
                    else { const float4* p = bvh.nodes + (size_t)cur * 8u + j; lo = p[0]; hi = p[4]; }
                    const float ax = __fmaf_rn(lo.x, ix, -qx), bx = __fmaf_rn(hi.x, ix, -qx);
                    const float ay = __fmaf_rn(lo.y, iy, -qy), by = __fmaf_rn(hi.y, iy, -qy);
                    const float az = __fmaf_rn(lo.z, iz, -qz), bz = __fmaf_rn(hi.z, iz, -qz);
                    const float nearD = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
                    const float farD = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
                    const uint32_t ref = ubits(lo.w);
                    // box occlusion with the slack that keeps every candidate within tol of the final hit in the walk
                    const bool hit = (farD >= nearD) && (farD >= 0.0f) && (nearD < best + (tol + tol)) && ref != RT_WIDE_EMPTY;
                    const float key = hit ? nearD : inf;
                    const float k1 = quadPermF<RT_QP(1, 2, 3, 0)>(key), k2 = quadPermF<RT_QP(2, 3, 0, 1)>(key), k3 = quadPermF<RT_QP(3, 0, 1, 2)>(key);
                    const uint32_t rank = ((k1 < key || (k1 == key && (tieMask & 1u))) ? 1u : 0u) + ((k2 < key || (k2 == key && (tieMask & 2u))) ? 1u : 0u) +
                                          ((k3 < key || (k3 == key && (tieMask & 4u))) ? 1u : 0u);
                    const unsigned long long mHit = __ballot(hit);
                    const uint32_t nh = (uint32_t)__popc((uint32_t)(mHit >> (lane & ~3u)) & 0xFu);
                    uint32_t next = (hit && rank == 0u) ? ref : 0u;
                    next |= quadPermU<RT_QP(1, 0, 3, 2)>(next);
                    next |= quadPermU<RT_QP(2, 3, 0, 1)>(next);
                    if (hit && rank != 0u) stack[(sp + (nh - 1u - rank)) * (RT_BLOCK / 4u)] = ref;   // the farthest deepest
                    if (nh != 0u) { sp += nh - 1u; cur = next; }
                    else if (sp == 0u) cur = RT_WIDE_DONE;
                    else { --sp; cur = stack[sp * (RT_BLOCK / 4u)]; }
                }
                in = in && (cur >> RT_NODE_LEAVES_SHIFT) == 0u;
                const unsigned long long m = __ballot(in);
                if (m == 0ull || 64u - nIdle - (uint32_t)__popcll(m) >= tune.otherMinLanes) break;
            }
        }
        else if (other)
        {
            if (cur != RT_WIDE_DONE)
            {
                // ---- leaf: lanes 0 and 1 of the quad test the leaf's triangles (MeshShape::Traverse_Leaf, MeshShape.cpp:134-207) ----
                const uint32_t numLeaves = cur >> RT_NODE_LEAVES_SHIFT;
                const float4* q = bvh.leaves + (size_t)(cur & RT_NODE_CHILD_MASK) * 6u + (j & 1u);
                const float4 v0 = q[0], e1 = q[2], e2 = q[4];
                Ray ray; ray.origin = V4(ox, oy, oz, 0.0f); ray.dir = V4(dx, dy, dz, 0.0f);
                float u, v, dist;
                const bool accepted = intersectTriangleRay(ray, V4(v0.x, v0.y, v0.z, 0.0f), V4(e1.x, e1.y, e1.z, 0.0f), V4(e2.x, e2.y, e2.z, 0.0f), u, v, dist) && j < numLeaves;
                if (shadow)
                {
                    const unsigned long long mOcc = __ballot(accepted && dist < best);
                    if (((uint32_t)(mOcc >> (lane & ~3u)) & 0xFu) != 0u) { occluded = true; cur = RT_WIDE_DONE; }
                }
                else
                {
                    const float t = accepted ? dist : inf;
                    const float t0 = quadPermF<RT_QP(0, 0, 0, 0)>(t), t1 = quadPermF<RT_QP(1, 1, 1, 1)>(t);
                    const float lo = fminf(t0, t1), hi = fmaxf(t0, t1);
                    if (lo < best)
                    {
                        second = fminf(best, hi);
                        best = lo;
                        // HitPoint written through by the lane that holds the winner (lane 0 on an exact tie: the ray is retraced anyway)
                        if (accepted && dist == lo && (j == 0u || t0 != lo))
                        {
                            prec(paths, R_HIT, slot) = f4(fbits(0u), fbits(triBase + ubits(v0.w) + j), dist, u);
                            prec(paths, R_SAMPLER, slot).x = v;
                        }
                    }
                    else second = fminf(second, lo);
                }
                if (cur != RT_WIDE_DONE)
                {
                    if (sp == 0u) cur = RT_WIDE_DONE;
                    else { --sp; cur = stack[sp * (RT_BLOCK / 4u)]; }
                }
            }
            if (cur == RT_WIDE_DONE)
            {
                // ---- finished ----
                if (shadow)
                {
                    if (occluded && j == 0u) pshadow(paths, light, 0, slot).w = -1.0f;   // unoccluded requests are tallied when they are resolved
                }
                else if (j == 0u)
                {
                    if (best == inf) prec(paths, R_HIT, slot) = f4(fbits(RT_INVALID_OBJECT), fbits(0u), inf, 0.0f);   // HitPoint.h:14-51
                    else if (second <= best + tol)
                    {
                        tune.exactQueue[atomicAdd(tune.exactCount, 1u)] = slot;   // a runner-up too close to call: the reference's own walk decides
                        numRetraced++;
                    }
                }
                have = false;
            }
        }
    }
    // counters: shadow rays traced here, rays handed to the binary-tree kernel
    __shared__ uint32_t sTally[2];
    if (threadIdx.x < 2u) sTally[threadIdx.x] = 0u;
    __syncthreads();
    if (numShadowRays) atomicAdd(&sTally[0], numShadowRays);
    if (numRetraced) atomicAdd(&sTally[1], numRetraced);
    __syncthreads();
    if (threadIdx.x == 0u && sTally[0]) atomicAdd(&counters[C_SHADOW], (unsigned long long)sTally[0]);
    if (threadIdx.x == 1u && sTally[1]) atomicAdd(&counters[RT_COUNTER_RETRACED], (unsigned long long)sTally[1]);
}

// ---- host: collapse of the reference's binary BVH (BVH::Node, 32 bytes, children adjacent) into the 4-wide tree ----
struct WideBuild
{
    std::vector<float4> nodes, leaves;
    uint32_t numNodes = 0, numLeaves = 0, stackNeed = 0;
    float bound[3] = { 0.0f, 0.0f, 0.0f };
    bool ok = false;
};

static WideBuild buildWideBvh(const RtNode* nodes, uint32_t numNodes, const RtTriangle* tris, uint32_t numTriangles)
{
    WideBuild w;
    if (numNodes == 0u || (nodes[0].leaves & 0x3FFFFFFFu) != 0u) return w;   // a root that is a leaf: nothing to collapse
    auto isLeaf = [&](uint32_t n) { return (nodes[n].leaves & 0x3FFFFFFFu) != 0u; };
    auto area = [&](uint32_t n) { const float ex = nodes[n].max[0] - nodes[n].min[0], ey = nodes[n].max[1] - nodes[n].min[1], ez = nodes[n].max[2] - nodes[n].min[2]; return ex * ey + ey * ez + ez * ex; };
    // breadth-first: wide node k collapses binary node order[k]
    std::vector<uint32_t> order; order.push_back(0u);
    std::vector<uint32_t> depthSlack; depthSlack.push_back(0u);   // stack entries a walk can hold when it ARRIVES at wide node k
    std::unordered_map<uint32_t, uint32_t> leafOrdinal;
    for (size_t k = 0; k < order.size(); ++k)
    {
        const uint32_t n = order[k];
        uint32_t kids[4]; uint32_t numKids = 2u;
        kids[0] = nodes[n].childIndex; kids[1] = nodes[n].childIndex + 1u;
        if (kids[1] >= numNodes) return w;
        while (numKids < 4u)
        {
            int pick = -1; float bestArea = -1.0f;
            for (uint32_t c = 0; c < numKids; ++c) if (!isLeaf(kids[c]) && area(kids[c]) > bestArea) { bestArea = area(kids[c]); pick = (int)c; }
            if (pick < 0) break;
            const uint32_t p = kids[pick];
            if (nodes[p].childIndex + 1u >= numNodes) return w;
            kids[pick] = nodes[p].childIndex; kids[numKids++] = nodes[p].childIndex + 1u;
        }
        float4 rec[8];
        for (uint32_t c = 0; c < 4u; ++c)
        {
            if (c >= numKids) { rec[c] = make_float4(INFINITY, INFINITY, INFINITY, __builtin_bit_cast(float, RT_WIDE_EMPTY)); rec[4 + c] = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f); continue; }
            const RtNode& kid = nodes[kids[c]];
            uint32_t ref;
            if (isLeaf(kids[c]))
            {
                const uint32_t numLeaves = kid.leaves & 0x3FFFFFFFu, first = kid.childIndex;
                if (numLeaves > 2u || (uint64_t)first + numLeaves > numTriangles) return w;   // the quad tests two triangles per leaf
                const auto it = leafOrdinal.find(first);
                uint32_t ordinal;
                if (it != leafOrdinal.end()) ordinal = it->second;
                else
                {
                    ordinal = (uint32_t)(w.leaves.size() / 6u); leafOrdinal[first] = ordinal;
                    const RtTriangle& a = tris[first]; const RtTriangle& b = tris[first + (numLeaves > 1u ? 1u : 0u)];
                    const float firstBits = __builtin_bit_cast(float, first);
                    w.leaves.push_back(make_float4(a.v0[0], a.v0[1], a.v0[2], firstBits)); w.leaves.push_back(make_float4(b.v0[0], b.v0[1], b.v0[2], firstBits));
                    w.leaves.push_back(make_float4(a.edge1[0], a.edge1[1], a.edge1[2], 0.0f)); w.leaves.push_back(make_float4(b.edge1[0], b.edge1[1], b.edge1[2], 0.0f));
                    w.leaves.push_back(make_float4(a.edge2[0], a.edge2[1], a.edge2[2], 0.0f)); w.leaves.push_back(make_float4(b.edge2[0], b.edge2[1], b.edge2[2], 0.0f));
                }
                if (ordinal > RT_NODE_CHILD_MASK) return w;
                ref = ordinal | (numLeaves << RT_NODE_LEAVES_SHIFT);
            }
            else
            {
                ref = (uint32_t)order.size();
                if (ref >= RT_NODE_CHILD_MASK) return w;
                order.push_back(kids[c]);
                depthSlack.push_back(depthSlack[k] + numKids - 1u);
            }
            rec[c] = make_float4(kid.min[0], kid.min[1], kid.min[2], __builtin_bit_cast(float, ref));
            rec[4 + c] = make_float4(kid.max[0], kid.max[1], kid.max[2], 0.0f);
        }
        if (depthSlack[k] + numKids - 1u > w.stackNeed) w.stackNeed = depthSlack[k] + numKids - 1u;
        w.nodes.insert(w.nodes.end(), rec, rec + 8);
    }
    w.numNodes = (uint32_t)order.size(); w.numLeaves = (uint32_t)(w.leaves.size() / 6u);
    for (int a = 0; a < 3; ++a) w.bound[a] = fmaxf(fabsf(nodes[0].min[a]), fabsf(nodes[0].max[a]));
    // the root box of the reference's builder encloses everything; make sure of it from the children actually stored
    for (size_t i = 0; i < w.nodes.size(); i += 8)
        for (uint32_t c = 0; c < 4u; ++c)
        {
            if (__builtin_bit_cast(uint32_t, w.nodes[i + c].w) == RT_WIDE_EMPTY) continue;
            const float lo[3] = { w.nodes[i + c].x, w.nodes[i + c].y, w.nodes[i + c].z }, hi[3] = { w.nodes[i + 4 + c].x, w.nodes[i + 4 + c].y, w.nodes[i + 4 + c].z };
            for (int a = 0; a < 3; ++a) { w.bound[a] = fmaxf(w.bound[a], fmaxf(fabsf(lo[a]), fabsf(hi[a]))); }
        }
    w.ok = w.numLeaves != 0u || w.numNodes != 0u;
    return w;
}
