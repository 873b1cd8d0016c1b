// rt_trace_quant.inl -- an EXPERIMENT (RTGPU_QUANT=1; measured no faster than k_trace: 182 vs 183 ms) and the home of what the default
// walk of single-mesh scenes, k_trace_wide (rt_trace_wide.inl), shares with it: the 16-bit grid, the per-leaf exact boxes and the
// exactness argument below.  The kernel here walks the binary tree of the reference with its child pairs re-encoded in 32 bytes.
// Included by rt_trace.hip (kernels: RT_DEVICE_KERNELS) and rt_runtime.hip (tree builders: RT_HOST_BUILDERS).
//
// Why.  k_trace (rt_device_traverse.h) is bound by the vector L1: a lane fetches its 64-byte node pair with four 16-byte loads, and a
// divergent 16-byte access occupies the texture-cache pipeline for a cycle whatever it uses of the line -- 0.71 accesses per clock and
// CU measured against a ceiling of one (profiles/r02_diag0_pmc_3.txt), vector ALU issue at 54 %, the memory behind it mostly idle;
// fetching the same pair twice makes the kernel 49 % slower.  A first attempt to cut accesses by giving each ray a quad of lanes over a
// 4-wide tree (one fully coalesced access per node) was bit-exact and 1.7x SLOWER: a quarter of the rays per wave at the same
// instruction count per step, 96 % vector-ALU issue (profiles/r02_quadwide_pmc_*.txt).  This kernel keeps one ray per lane and
// halves the bytes instead:
//   * a child pair is two 16-byte records {min.xyz, max.xyz as 16-bit grid coordinates, child reference}: two accesses per visit
//     instead of four; the grid spans the mesh's bounds, planes are rounded OUTWARDS with a step to spare, so a stored box always
//     contains the reference's box;
//   * the slab test needs no decode step: t = fma(float(q), step * invDir, base * invDir - origin * invDir), two constants per axis
//     and ray, so a visit costs the twelve integer-to-float conversions on top of the old twelve fmas;
//   * boxes that are only conservative cannot decide what the reference tests, so a leaf's triangles count only if the ray also passes
//     the leaf's EXACT box (the test the reference's walk performs before it reaches them) -- fetched only when a triangle was
//     actually hit.
//
// Exactness.  Same argument as for any walk that visits a superset of the reference's leaves in another order: every candidate hit
// (a triangle the ray intersects inside a leaf whose exact box it passes) has the same (t, u, v) as in the reference's walk, because
// the triangle test and the exact box test are the reference's arithmetic; the slab test is monotone in the box planes, so passing a
// leaf's exact box implies passing every ancestor's box with a smaller entry distance, i.e. the reference's walk reaches exactly these
// candidates unless its running hit distance culls one -- which can only change the result when two candidates are closer together than
// the disagreement between a box's entry distance and its triangle's hit distance.  The kernel culls with a slack (near < best + 2 tol),
// tracks the SECOND smallest candidate distance, and a ray whose runner-up lies within tol of its best (tol = 16 ulps of the largest
// term of its slab tests) is not trusted: it goes to the exact queue and is traced again by k_trace in the reference's order.  So do
// rays with a zero direction component (their slab tests produce NaNs, which the reference's min/max operand order resolves in its
// own way) and rays that start so far outside the mesh that the folded slab test's rounding could eat the spare grid step.  Any-hit rays
// need no runner-up: occlusion is an OR over the same candidate set (their leaf gate includes the reference's entry-distance test
// against the fixed ray length).  The reference's box / triangle test counters belong to its own walk: with the intersection
// counters on, k_trace runs alone.

#define RT_QUANT_DONE 0xFFFFFFFFu   // cur: the ray is finished (same value as RT_LEVEL_EXHAUSTED: the mesh level has no node left)
#define RT_QUANT_GRID 65535.0f

struct QuantBvh
{
    const float4* pairs;     // record c of pair (childIndex, childIndex + 1) at pairs[childIndex + c]: {minx | miny << 16, minz | maxx << 16, maxy | maxz << 16, ref}
    const float4* gate;      // exact box of the leaf whose first triangle is t: gate[2 t] = {min.xyz, -}, gate[2 t + 1] = {max.xyz, -}
    uint32_t root;           // packed reference of the root node
    uint32_t stackNeed;
    float base[3], step[3];  // plane(q) = base + q * step
    float bound[3];          // largest |coordinate| of the mesh per axis (for the per-ray tolerance)
};

struct QuantTuning
{
    uint32_t refillMinIdle, otherMinLanes;
    float shadowOffset;
    uint32_t* exactQueue; uint32_t* exactCount;               // closest-hit rays handed to the binary-tree kernel
    uint32_t* exactShadowQueue; uint32_t* exactShadowCount;   // any-hit requests handed to it
};

#ifdef RT_DEVICE_KERNELS
template <int kStack>
__global__ void __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(kStack <= 24 ? 5 : 1))) k_trace_quant(const RtSceneDesc scene, const QuantBvh bvh, const Paths paths,
                                                          const uint32_t* __restrict__ queue, const uint32_t* __restrict__ queueCount,
                                                          const uint32_t* __restrict__ shadowQueue, const uint32_t* __restrict__ shadowCount,
                                                          uint32_t* __restrict__ cursor, unsigned long long* counters, const QuantTuning tune)
{
    __shared__ uint32_t sStack[kStack * RT_BLOCK];
    uint32_t* const stack = sStack + threadIdx.x;   // entry e at stack[e * RT_BLOCK]: bank = lane, conflict free at any depth
    const uint32_t numClosest = queueCount ? *queueCount : 0u;
    const uint32_t count = numClosest + (shadowCount ? *shadowCount : 0u);
    const M4 invTransform = loadM4(scene.objects[0].invTransform);
    const RtTriangle* const tris = scene.triangles + scene.meshes[scene.objects[0].meshIndex].firstTriangle;
    const float inf = __uint_as_float(0x7f800000u);

    // per-lane ray state
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0;     // local ray (triangle tests, leaf gate)
    float ax = 0, ay = 0, az = 0, bx = 0, by = 0, bz = 0;     // folded slab constants: t(q) = fma(q, a, b)
    float best = 0, second = 0, tol = 0;
    uint32_t cur = RT_QUANT_DONE, sp = 0, slot = 0, light = 0;
    bool have = false, shadow = false, occluded = false, exhausted = false;
    uint32_t numRetraced = 0, numShadowRays = 0;

    uint32_t chunkSize = count / (gridDim.x * (RT_BLOCK / 64u) * 4u);
    chunkSize = chunkSize < 64u ? 64u : (chunkSize > 1024u ? 1024u : chunkSize);
    WaveChunk chunk = { 0u, 0u };

    for (;;)
    {
        const bool interior = have && (cur >> RT_NODE_LEAVES_SHIFT) == 0u;
        const bool other = have && !interior;      // at a leaf, or finished
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
            const uint32_t idx = waveTake(!have, chunk);
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
                // largest magnitude a slab test of this ray can produce, per axis; 2^-21 of it bounds the folded test's rounding
                const float mx = fabsf(local.originDivDir.x) + bvh.bound[0] * fabsf(local.invDir.x);
                const float my = fabsf(local.originDivDir.y) + bvh.bound[1] * fabsf(local.invDir.y);
                const float mz = fabsf(local.originDivDir.z) + bvh.bound[2] * fabsf(local.invDir.z);
                const float fold = 4.76837158203125e-07f;   // 2^-21
                const bool trusted = rayIsNaNFree(local) &&
                                     mx * fold < bvh.step[0] * fabsf(local.invDir.x) && my * fold < bvh.step[1] * fabsf(local.invDir.y) && mz * fold < bvh.step[2] * fabsf(local.invDir.z);
                if (!trusted)
                {
                    // a zero direction component (NaNs in the reference's slab test) or an origin far outside the mesh: the reference's walk only
                    if (shadow) tune.exactShadowQueue[atomicAdd(tune.exactShadowCount, 1u)] = request;
                    else tune.exactQueue[atomicAdd(tune.exactCount, 1u)] = slot;
                    numRetraced++;
                }
                else
                {
                    ox = local.origin.x; oy = local.origin.y; oz = local.origin.z; dx = local.dir.x; dy = local.dir.y; dz = local.dir.z;
                    ax = bvh.step[0] * local.invDir.x; ay = bvh.step[1] * local.invDir.y; az = bvh.step[2] * local.invDir.z;
                    bx = __fmaf_rn(bvh.base[0], local.invDir.x, -local.originDivDir.x);
                    by = __fmaf_rn(bvh.base[1], local.invDir.y, -local.originDivDir.y);
                    bz = __fmaf_rn(bvh.base[2], local.invDir.z, -local.originDivDir.z);
                    tol = shadow ? 0.0f : fmaxf(fmaxf(mx, my), mz) * 1.9073486328125e-06f;   // 2^-19: 16 ulps
                    best = maxDistance; second = inf; occluded = false;
                    sp = 0u; cur = bvh.root;
                    have = true;
                    if (shadow) numShadowRays++;   // (a request handed to the binary-tree kernel is counted there)
                }
            }
            continue;
        }
        if ((mI | mO) == 0ull) break;
        if (mI != 0ull && (uint32_t)__popcll(mO) < tune.otherMinLanes)
        {
            // ---- interior phase: two conservative slab tests per step, until enough lanes wait at a leaf or are finished ----
            bool in = interior;
            const float limit = best + (tol + tol);   // box occlusion with the slack that keeps every candidate within tol of the final hit in the walk
            for (;;)
            {
                if (in)
                {
                    const float4* p = bvh.pairs + (cur & RT_NODE_CHILD_MASK);
                    const float4 qa = p[0], qb = p[1];
                    const uint32_t a0 = ubits(qa.x), a1 = ubits(qa.y), a2 = ubits(qa.z), b0 = ubits(qb.x), b1 = ubits(qb.y), b2 = ubits(qb.z);
                    const float aNx = __fmaf_rn((float)(a0 & 0xFFFFu), ax, bx), aNy = __fmaf_rn((float)(a0 >> 16), ay, by), aNz = __fmaf_rn((float)(a1 & 0xFFFFu), az, bz);
                    const float aXx = __fmaf_rn((float)(a1 >> 16), ax, bx), aXy = __fmaf_rn((float)(a2 & 0xFFFFu), ay, by), aXz = __fmaf_rn((float)(a2 >> 16), az, bz);
                    const float bNx = __fmaf_rn((float)(b0 & 0xFFFFu), ax, bx), bNy = __fmaf_rn((float)(b0 >> 16), ay, by), bNz = __fmaf_rn((float)(b1 & 0xFFFFu), az, bz);
                    const float bXx = __fmaf_rn((float)(b1 >> 16), ax, bx), bXy = __fmaf_rn((float)(b2 & 0xFFFFu), ay, by), bXz = __fmaf_rn((float)(b2 >> 16), az, bz);
                    const float nearA = fmaxf(fmaxf(fminf(aNx, aXx), fminf(aNy, aXy)), fminf(aNz, aXz));
                    const float farA = fminf(fminf(fmaxf(aNx, aXx), fmaxf(aNy, aXy)), fmaxf(aNz, aXz));
                    const float nearB = fmaxf(fmaxf(fminf(bNx, bXx), fminf(bNy, bXy)), fminf(bNz, bXz));
                    const float farB = fminf(fminf(fmaxf(bNx, bXx), fmaxf(bNy, bXy)), fmaxf(bNz, bXz));
                    const bool hitA = (farA >= nearA) && (farA >= 0.0f) && (nearA < limit);
                    const bool hitB = (farB >= nearB) && (farB >= 0.0f) && (nearB < limit);
                    const uint32_t a = ubits(qa.w), b = ubits(qb.w);
                    const bool both = hitA && hitB;
                    const bool swap = both && (nearB < nearA);   // nearer child first (any order gives the same candidates)
                    if (both) { stack[sp * RT_BLOCK] = swap ? a : b; ++sp; }
                    if (hitA || hitB) cur = (hitA && !swap) ? a : b;
                    else if (sp == 0u) cur = RT_QUANT_DONE;
                    else { --sp; cur = stack[sp * RT_BLOCK]; }
                }
                in = in && (cur >> RT_NODE_LEAVES_SHIFT) == 0u;
                const unsigned long long m = __ballot(in);
                if (m == 0ull || 64u - nIdle - (uint32_t)__popcll(m) >= tune.otherMinLanes) break;
            }
        }
        else if (other)
        {
            if (cur != RT_QUANT_DONE)
            {
                // ---- leaf: MeshShape::Traverse_Leaf(_Shadow), MeshShape.cpp:134-207 ----
                const uint32_t numLeaves = cur >> RT_NODE_LEAVES_SHIFT, first = cur & RT_NODE_CHILD_MASK;
                Ray ray; ray.origin = V4(ox, oy, oz, 0.0f); ray.dir = V4(dx, dy, dz, 0.0f);
                V4 v0, e1, e2, nv0, ne1, ne2;
                loadTriangle(tris + first, v0, e1, e2);
                loadTriangle(tris + first + (numLeaves > 1u ? 1u : 0u), nv0, ne1, ne2);   // the second triangle of the leaf rides in the same round trip
                float u0, v0_, t0, u1 = 0.0f, v1 = 0.0f, t1 = inf;
                if (!intersectTriangleRay(ray, v0, e1, e2, u0, v0_, t0)) t0 = inf;
                if (numLeaves > 1u && !intersectTriangleRay(ray, nv0, ne1, ne2, u1, v1, t1)) t1 = inf;
                const float lo = fminf(t0, t1);
                if (lo < best + tol)
                {
                    // a hit that matters: it counts only if the ray passes the leaf's exact box, as in the reference's walk
                    const float4 gmin = bvh.gate[2u * first], gmax = bvh.gate[2u * first + 1u];
                    const Ray gateRay = makeRayUnsafe(ray.origin, ray.dir);   // = the ray transformRayUnsafe built
                    float nearD;
                    const bool pass = intersectBoxRayNoNaN(gateRay, gmin.x, gmin.y, gmin.z, gmax.x, gmax.y, gmax.z, nearD) && (!shadow || nearD < best);
                    if (pass)
                    {
                        if (shadow) { if (lo < best) { occluded = true; cur = RT_QUANT_DONE; } }
                        else
                        {
                            const float hi = fmaxf(t0, t1);
                            if (lo < best)
                            {
                                second = fminf(best, hi);
                                best = lo;
                                const bool firstWins = t0 <= t1;   // HitPoint written through (an exact tie is retraced anyway)
                                prec(paths, R_HIT, slot) = f4(fbits(0u), fbits(first + (firstWins ? 0u : 1u)), lo, firstWins ? u0 : u1);
                                prec(paths, R_SAMPLER, slot).x = firstWins ? v0_ : v1;
                            }
                            else second = fminf(second, lo);
                        }
                    }
                }
                if (cur != RT_QUANT_DONE)
                {
                    if (sp == 0u) cur = RT_QUANT_DONE;
                    else { --sp; cur = stack[sp * RT_BLOCK]; }
                }
            }
            if (cur == RT_QUANT_DONE)
            {
                // ---- finished ----
                if (shadow)
                {
                    if (occluded) pshadow(paths, light, 0, slot).w = -1.0f;   // unoccluded requests are tallied when they are resolved
                }
                else if (best == inf) prec(paths, R_HIT, slot) = f4(fbits(RT_INVALID_OBJECT), fbits(0u), inf, 0.0f);   // HitPoint.h:14-51
                else if (second <= best + tol)
                {
                    tune.exactQueue[atomicAdd(tune.exactCount, 1u)] = slot;   // a runner-up too close to call: the reference's own walk decides
                    numRetraced++;
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

#endif   // RT_DEVICE_KERNELS

#ifdef RT_HOST_BUILDERS
// ---- host: the reference's binary BVH (BVH::Node, 32 bytes, children adjacent) re-encoded ----
struct QuantBuild
{
    std::vector<float4> pairs, gate;
    uint32_t root = 0, stackNeed = 0;
    float base[3] = { 0, 0, 0 }, step[3] = { 0, 0, 0 }, bound[3] = { 0, 0, 0 };
    bool ok = false;
};

static QuantBuild buildQuantBvh(const RtNode* nodes, uint32_t numNodes, uint32_t numTriangles, uint32_t depth)
{
    QuantBuild q;
    if (numNodes < 3u || (nodes[0].leaves & 0x3FFFFFFFu) != 0u) return q;   // a root that is a leaf: nothing to walk
    // bounds from every node actually stored (node 1 is never written by the reference's builder)
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    std::vector<uint8_t> reachable(numNodes, 0);
    {
        std::vector<uint32_t> todo; todo.push_back(0u); reachable[0] = 1;
        while (!todo.empty())
        {
            const uint32_t n = todo.back(); todo.pop_back();
            for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], nodes[n].min[a]); hi[a] = fmaxf(hi[a], nodes[n].max[a]); }
            if ((nodes[n].leaves & 0x3FFFFFFFu) != 0u) continue;
            const uint32_t c = nodes[n].childIndex;
            if ((c & 1u) != 0u || (uint64_t)c + 1u >= numNodes || reachable[c] || reachable[c + 1u]) return q;   // pairs are even-aligned in the reference's layout
            reachable[c] = reachable[c + 1u] = 1; todo.push_back(c); todo.push_back(c + 1u);
        }
    }
    for (int a = 0; a < 3; ++a)
    {
        if (!(lo[a] <= hi[a]) || !std::isfinite(lo[a]) || !std::isfinite(hi[a])) return q;
        const float ext = hi[a] - lo[a];
        const float largest = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
        q.step[a] = fmaxf(ext, 1e-6f * fmaxf(largest, 1e-30f)) / (RT_QUANT_GRID - 8.0f);
        q.base[a] = lo[a] - 4.0f * q.step[a];
        q.bound[a] = fmaxf(fabsf(lo[a]), fabsf(hi[a])) + 8.0f * q.step[a];
        if (!(q.step[a] > 0.0f) || !std::isfinite(q.step[a])) return q;
    }
    // plane(qv) as the device could see it at worst: the kernel folds base and step into the ray's constants, whose rounding is
    // covered by the one spare step; here the stored coordinate is pushed out until the plain float plane is a full step outside
    auto plane = [&](int a, uint32_t v) { return (double)q.base[a] + (double)v * (double)q.step[a]; };
    auto qmin = [&](int a, float x) { long v = (long)floor(((double)x - (double)q.base[a]) / (double)q.step[a]) - 1; while (v > 0 && plane(a, (uint32_t)v) > (double)x - (double)q.step[a]) --v; return (uint32_t)(v < 0 ? 0 : v); };
    auto qmax = [&](int a, float x) { long v = (long)ceil(((double)x - (double)q.base[a]) / (double)q.step[a]) + 1; while (v < 65535 && plane(a, (uint32_t)v) < (double)x + (double)q.step[a]) ++v; return (uint32_t)(v > 65535 ? 65535 : v); };
    q.pairs.assign(numNodes, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    q.gate.assign((size_t)2 * numTriangles, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    for (uint32_t n = 2u; n < numNodes; ++n)
    {
        if (!reachable[n]) continue;
        const RtNode& node = nodes[n];
        const uint32_t numLeaves = node.leaves & 0x3FFFFFFFu;
        if (numLeaves > 2u || node.childIndex > RT_NODE_CHILD_MASK) return q;   // the reference builds leaves of at most two triangles (BVHBuilder.h:16)
        uint32_t v[6];
        for (int a = 0; a < 3; ++a)
        {
            v[a] = qmin(a, node.min[a]); v[3 + a] = qmax(a, node.max[a]);
            if (plane(a, v[a]) > (double)node.min[a] - 0.5 * (double)q.step[a] || plane(a, v[3 + a]) < (double)node.max[a] + 0.5 * (double)q.step[a]) return q;   // the grid has room by construction
        }
        const uint32_t d0 = v[0] | (v[1] << 16), d1 = v[2] | (v[3] << 16), d2 = v[4] | (v[5] << 16);
        const uint32_t ref = node.childIndex | (numLeaves << RT_NODE_LEAVES_SHIFT);
        q.pairs[n] = make_float4(__builtin_bit_cast(float, d0), __builtin_bit_cast(float, d1), __builtin_bit_cast(float, d2), __builtin_bit_cast(float, ref));
        if (numLeaves != 0u)
        {
            if ((uint64_t)node.childIndex + numLeaves > numTriangles) return q;
            q.gate[2u * (size_t)node.childIndex] = make_float4(node.min[0], node.min[1], node.min[2], 0.0f);
            q.gate[2u * (size_t)node.childIndex + 1u] = make_float4(node.max[0], node.max[1], node.max[2], 0.0f);
        }
    }
    q.root = nodes[0].childIndex;   // interior: its packed reference is its child index
    q.stackNeed = depth;
    q.ok = true;
    return q;
}
#endif   // RT_HOST_BUILDERS
