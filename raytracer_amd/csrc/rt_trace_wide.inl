// rt_trace_wide.inl -- traversal of single-mesh scenes over a 4-WIDE tree collapsed from the reference's binary tree (same SAH splits,
// same leaves), one ray per lane.  Included by rt_trace.hip (kernels: RT_DEVICE_KERNELS) and rt_runtime.hip (tree builders: RT_HOST_BUILDERS) after rt_wide_grid.inl, whose grid, leaf gates and exactness argument it
// shares.
//
// Why.  k_trace waits ~0.8 us per dependent node fetch with 20 waves per CU to hide it: the walk is a chain of ~29 round
// trips per ray.  Two earlier attempts changed what a round trip moves -- half the bytes (k_trace_quant: no gain, the conversions ate
// it), a quad of lanes per ray (1.7x slower, instruction bound) -- not how many there are.  A 4-wide node shortens the chain to 17: one
// round trip fetches four children (64 bytes: four of k_trace_quant's 16-byte records), i.e. two levels of the binary tree at once, and
// the per-step overhead (stack, loop, scheduling ballots) is paid half as often.
// What came of it (profiles/r02_wide_*): the kernel is no longer latency-bound but ISSUE-bound -- vector ALU busy 90 % at 57 % lane
// utilisation -- so what counts is instructions per visit (137 after the permute / pair-sort step below, 161 before): 164.5 -> 147 ms
// per 69 passes of the benchmark against k_trace.  Variants that add live state (a leaf set aside, shared any-hit rays in the drain phase,
// fat leaves, one gate per ray, deferred children in slot order, one-wave blocks) spilled on the 96-VGPR cliff or lost otherwise; their
// measurements are in profiles/r02_wide_diag.txt and DESIGN 4, their code is in the history (round 2), not here.
//
// Node n = records nodes[4 n .. 4 n + 3], one per child: {min.xyz, max.xyz as 16-bit grid coordinates (rounded outwards, QuantBvh's
// grid), reference}.  A reference is an interior node's index, a leaf (first triangle | count << 30, the binary tree's own leaves of
// one or two triangles), or RT_WIDE_EMPTY behind a degenerate box for the unused slots of a node with two or three children.
//
// Exactness: k_trace_quant's argument word for word (conservative boxes visit a superset of the reference's leaves in another order;
// a leaf's triangles count only behind the leaf's exact box; runner-up within tol, zero direction components, far origins -> the
// binary-tree kernel decides), plus one more hand-over: a ray whose stack would overflow.

#define RT_WIDE_EMPTY 0xC0000000u    // leaf count 3 never occurs in the reference's trees (BVHBuilder.h:16): "no child"

struct WideBvh
{
    const float4* nodes;
    const float4* gate;      // QuantBvh::gate
    uint32_t numNodes;
    float base[3], step[3], bound[3];
};

struct WideTuning
{
    uint32_t refillMinIdle, otherMinLanes;
    float shadowOffset;
    uint32_t* exactQueue; uint32_t* exactCount;               // closest-hit rays handed to the binary-tree kernel
    uint32_t* exactShadowQueue; uint32_t* exactShadowCount;   // any-hit requests handed to it
    const uint32_t* denseCounts; uint32_t denseShardCapacity; // dense path state (TravTuning)
    uint32_t chunkMin;                                        // smallest piece of the work queue a wave claims at once
    uint32_t localExact;                                      // != 0: a block traces the rays its walk does not decide itself (k_trace_wide; RTGPU_LOCAL_EXACT=0: off)
    uint32_t drainAbortAfter;                                 // != 0: a wave whose work queue ran dry this many loop iterations ago hands the rays it still walks to the binary-tree kernel
    uint32_t reverseOrder;                                    // != 0: the queue is taken from its end (any-hit requests first, closest-hit rays last); 0 (default since round 6): front to back -- the launch's drain is then made of the any-hit rays, the short ones under the far-first order
    uint32_t anyHitFarFirst;                                  // != 0: an any-hit ray walks the FARTHEST child it enters next (round 6; RTGPU_ANYHIT_FAR_FIRST=0: nearest, as closest-hit rays do)
};

#ifdef RT_DEVICE_KERNELS
// slab test of one child record against the ray's folded constants; near is clamped to >= 0 (its bits then order like the float).
// Which of an axis's two planes the ray meets first is a property of the RAY (the sign of its direction), so three byte permutes with
// per-ray selectors (v_perm_b32) put {near plane, far plane} of every axis into one word and the six min / max of the textbook slab
// test disappear: 3 perm + 6 cvt (sub-word select) + 6 fma + max + max3 + min3 per child.
// Record words: w0 = minx | miny << 16, w1 = minz | maxx << 16, w2 = maxy | maxz << 16.  __builtin_amdgcn_perm(hi, lo, sel): byte i of
// the result is byte sel[i] of {lo = bytes 0-3, hi = bytes 4-7}.
#define RT_WIDE_SEL_X_POS 0x07060100u   // perm(w1, w0): minx (bytes 0,1) first, maxx (bytes 6,7) second
#define RT_WIDE_SEL_X_NEG 0x01000706u
#define RT_WIDE_SEL_Y_POS 0x05040302u   // perm(w2, w0): miny (bytes 2,3) first, maxy (bytes 4,5) second
#define RT_WIDE_SEL_Y_NEG 0x03020504u
#define RT_WIDE_SEL_Z_POS 0x07060100u   // perm(w2, w1): minz (bytes 0,1) first, maxz (bytes 6,7) second
#define RT_WIDE_SEL_Z_NEG 0x01000706u
#define RT_WIDE_SLAB(q, nearOut, farOut)                                                                                                          \
    {                                                                                                                                             \
        const uint32_t w0 = ubits(q.x), w1 = ubits(q.y), w2 = ubits(q.z);                                                                         \
        const uint32_t px = __builtin_amdgcn_perm(w1, w0, selX), py = __builtin_amdgcn_perm(w2, w0, selY), pz = __builtin_amdgcn_perm(w2, w1, selZ); \
        const float nx = __fmaf_rn((float)(px & 0xFFFFu), ax, bx), ny = __fmaf_rn((float)(py & 0xFFFFu), ay, by), nz = __fmaf_rn((float)(pz & 0xFFFFu), az, bz); \
        const float xx = __fmaf_rn((float)(px >> 16), ax, bx), xy = __fmaf_rn((float)(py >> 16), ay, by), xz = __fmaf_rn((float)(pz >> 16), az, bz);            \
        nearOut = fmaxf(fmaxf(nx, ny), fmaxf(nz, 0.0f));                                                                                           \
        farOut = fminf(fminf(xx, xy), xz);                                                                                                        \
    }
#define RT_WIDE_IS_LEAF(ref) ((((ref) >> RT_NODE_LEAVES_SHIFT) - 1u) < 2u)   // one or two triangles; not an interior node (0), not RT_WIDE_EMPTY / RT_QUANT_DONE (3)

// The block's own list of the rays its walk does not decide (LDS): they are traced by the reference's walk (traceBinaryLoop) in the same launch
// when the block's 4-wide walk is done, instead of by a launch of their own behind this one (ten launches of 70 ... 1600 us per batch for 0.1 %
// of the rays, profiles/r03_timeline_serial_start_of_round.txt).  What does not fit the list goes to the launch's queues as before.
struct WideLocal
{
    uint32_t* exact; uint32_t* exactCount;       // closest-hit rays (path slots)
    uint32_t* shadow; uint32_t* shadowCount;     // any-hit requests (light * capacity + slot)
    uint32_t capacity;                           // entries per list; 0: no local lists
};
RT_DEV void widePushExact(const WideTuning& tune, const WideLocal& local, bool shadowRequest, uint32_t request)
{
    if (local.capacity != 0u)
    {
        const uint32_t i = atomicAdd(shadowRequest ? local.shadowCount : local.exactCount, 1u);   // (the consumer clamps the count to the capacity)
        if (i < local.capacity) { (shadowRequest ? local.shadow : local.exact)[i] = request; return; }
    }
    if (tune.exactQueue == nullptr) return;   // k_tail: its lists hold every request a chunk can produce (rt_tail.hip states the invariant)
    if (shadowRequest) tune.exactShadowQueue[atomicAdd(tune.exactShadowCount, 1u)] = request;
    else tune.exactQueue[atomicAdd(tune.exactCount, 1u)] = request;
}
#define RT_WIDE_STACK 16           // stack entries per lane of the 4-wide walk
#define RT_WIDE_PARK 6u            // words per lane behind the stack (traceWideLoop's `park`)
#define RT_WIDE_LOCAL_EXACT 256u   // per block and kind: ~40 x what a block of the benchmark hands over per launch

// The walk as a device function (k_trace_wide below; k_tail, rt_tail.hip, runs it over a block's own queues in LDS: `queue`, `shadowQueue`,
// the counts and `cursor` are generic pointers, `sharingWaves` = the waves that claim from `cursor`).
template <int kStack, bool kDiag = false>
RT_DEV void traceWideLoop(const RtSceneDesc& scene, const WideBvh& bvh, const Paths& paths, const uint32_t* queue, const uint32_t* queueCount,
                          const uint32_t* shadowQueue, const uint32_t* shadowCount, uint32_t* cursor, unsigned long long* counters, const WideTuning& tune,
                          const WideLocal& handOver, uint32_t* sStack, uint32_t* sDensePrefix, uint32_t sharingWaves)
{
    uint32_t* const stack = sStack + threadIdx.x;   // entry e at stack[e * RT_BLOCK]: bank = lane, conflict free at any depth
    // Behind the stack's kStack entries: RT_WIDE_PARK words per lane that only the leaf phase reads -- the local ray's invDir and originDivDir, which the
    // exact box test of a leaf needs (round 5: rebuilding them there cost three IEEE divisions, ~100 of the leaf phase's 320 instructions, every time
    // some lane of the wave had a triangle hit to confirm)
    float* const park = reinterpret_cast<float*>(sStack + kStack * RT_BLOCK) + threadIdx.x;
    if (tune.denseCounts) { denseLoadPrefix(tune.denseCounts, sDensePrefix); __syncthreads(); }
    const uint32_t numClosest = tune.denseCounts ? sDensePrefix[RT_DENSE_SHARDS] : (queueCount ? *queueCount : 0u);
    const uint32_t count = numClosest + (shadowCount ? *shadowCount : 0u);
    // the mesh object's inverse transform is fetched per refill through the constant address space (scalar loads of a uniform address): sixteen scalar registers
    // less across the whole loop, whose header spilled a dozen of them into vector lanes every iteration
    typedef const __attribute__((address_space(4))) float* ConstF;
    const ConstF invTransformWords = (ConstF)(uintptr_t)scene.objects[0].invTransform;
    const RtTriangle* const tris = scene.triangles + scene.meshes[scene.objects[0].meshIndex].firstTriangle;
    const float inf = __uint_as_float(0x7f800000u);

    // per-lane ray state
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0;     // local ray (triangle tests, leaf gate)
    float ax = 0, ay = 0, az = 0, bx = 0, by = 0, bz = 0;     // folded slab constants: t(q) = fma(q, a, b)
    float best = 0, second = 0, tol = 0;
    uint32_t cur = RT_QUANT_DONE, sp = 0, slot = 0, light = 0;
    bool have = false, shadow = false, occluded = false, exhausted = false, overflow = false;
    // tallies per WAVE (ballots at wave-uniform points of the loop: scalar registers; as per-lane counters they were four of the 96 vector registers)
    uint32_t numRetraced = 0, numShadowRays = 0, numUntrusted = 0, numOverflow = 0;
    uint32_t diagVisits = 0, diagSlots = 0, diagLeaves = 0;   // kDiag: interior visits, lane slots of the interior loop (64 per wave step), leaf visits
    uint32_t diagGates = 0, diagHitWrites = 0;                // kDiag, RTGPU_WIDE_DIAG=3: exact-box fetches and hit records written through, per lane and launch (the walk's byte model, bench.py)
    uint32_t diagMaxSp = 0, diagDeep[3] = { 0u, 0u, 0u };     // kDiag, RTGPU_WIDE_DIAG=2: rays whose stack held more than 9 / 13 / 17 entries (what a 12 / 16 / 20-entry stack would hand over)

    uint32_t chunkSize = count / (sharingWaves * 4u);
    chunkSize = chunkSize < tune.chunkMin ? tune.chunkMin : (chunkSize > 1024u ? 1024u : chunkSize);
    WaveChunk chunk = { 0u, 0u };

    // kDiag: wave clock per phase (0 refill, 1 interior loop, 2 leaf / finish) and the number of times each ran
    unsigned long long diagClock[3] = { 0ull, 0ull, 0ull }, diagPrev = kDiag ? (unsigned long long)clock64() : 0ull, diagStart = diagPrev;
    uint32_t diagRuns[3] = { 0u, 0u, 0u }, diagPhase = 0u, diagClaims = 0u;
    unsigned long long diagClaimClock = 0ull, diagExhausted = 0ull;   // when this wave found the work queue empty   // inside the refill phase: waiting for the work cursor's atomic
    uint32_t drainIterations = 0u;   // (wave-uniform) loop iterations since the work queue ran dry
    for (;;)
    {
        if (kDiag) { const unsigned long long now = (unsigned long long)clock64(); diagClock[diagPhase] += now - diagPrev; diagPrev = now; }
        if (exhausted && tune.drainAbortAfter != 0u && ++drainIterations == tune.drainAbortAfter && have) { overflow = true; cur = RT_QUANT_DONE; }
        const bool interior = have && (cur >> RT_NODE_LEAVES_SHIFT) == 0u;
        const bool other = have && !interior;      // at a leaf, or finished
        const unsigned long long mI = __ballot(interior), mO = __ballot(other);
        const uint32_t nIdle = 64u - (uint32_t)__popcll(mI) - (uint32_t)__popcll(mO);
        if (!exhausted && (nIdle == 64u || nIdle >= tune.refillMinIdle))
        {
            // ---- refill ----
            if (kDiag) { diagPhase = 0u; diagRuns[0]++; }
            if (chunk.next >= chunk.end)
            {
                const unsigned long long claim0 = kDiag ? (unsigned long long)clock64() : 0ull;
                waveClaimChunk(chunk, cursor, chunkSize, count);
                if (kDiag) { diagClaimClock += (unsigned long long)clock64() - claim0; diagClaims++; }
                if (chunk.next >= chunk.end) { exhausted = true; if (kDiag) diagExhausted = (unsigned long long)clock64(); continue; }
            }
            uint32_t idx = waveTake(!have, chunk);
            if (tune.reverseOrder != 0u && idx != 0xFFFFFFFFu) idx = count - 1u - idx;
            bool tookUntrusted = false, tookShadow = false;
            if (idx != 0xFFFFFFFFu)
            {
                shadow = idx >= numClosest;
                const uint32_t request = shadow ? shadowQueue[idx - numClosest]
                                                : (tune.denseCounts ? denseLiveSlot(sDensePrefix, tune.denseShardCapacity, idx) : (queue ? queue[idx] : idx));
                // one ray construction for both kinds of request (a wave usually refills both at once): Ray::Ray normalises the direction
                // (PathTracerMIS.cpp:86 / :392), then the origin moves along it -- 1e-4 for an any-hit ray, 1e-3 for a bounce, not at all for
                // a primary ray
                float maxDistance = inf, offset;
                float4 origin, dir;
                if (shadow)
                {
                    // request = light * capacity + slot; one request per vertex (LightSamplingStrategy::Single: the arena holds one light's records) needs no division
                    if (paths.maxLights == 1u) { light = 0u; slot = request; } else { light = request / paths.capacity; slot = request - light * paths.capacity; }
                    origin = ldStream(prec(paths, R_SH_P, slot)); dir = ldStream(pshadow(paths, light, 0, slot));
                    maxDistance = dir.w;           // hitPoint.distance = illuminateResult.distance * 0.999f
                    offset = tune.shadowOffset;
                }
                else
                {
                    slot = request; light = 0u;
                    origin = ldStream(prec(paths, R_ORIGIN, slot)); dir = ldStream(prec(paths, R_DIR, slot));
                    offset = 0.001f;
                }
                Ray world = makeRay(V4(origin.x, origin.y, origin.z, 0.0f), V4(dir.x, dir.y, dir.z, 0.0f));
                if (shadow || (ubits(origin.w) & 0xFFu) != 0u) world.origin = world.origin + world.dir * offset;
                M4 invTransform;
                for (int r = 0; r < 4; ++r) invTransform.r[r] = V4(invTransformWords[4 * r], invTransformWords[4 * r + 1], invTransformWords[4 * r + 2], invTransformWords[4 * r + 3]);
                const Ray local = makeRayUnsafe3(transformPoint(invTransform, world.origin), transformVector(invTransform, world.dir));   // = transformRayUnsafe: MeshShape is entered in object space, Scene.cpp:128-145
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
                    widePushExact(tune, handOver, shadow, shadow ? request : slot);
                    tookUntrusted = true;
                }
                else
                {
                    ox = local.origin.x; oy = local.origin.y; oz = local.origin.z; dx = local.dir.x; dy = local.dir.y; dz = local.dir.z;
                    ax = bvh.step[0] * local.invDir.x; ay = bvh.step[1] * local.invDir.y; az = bvh.step[2] * local.invDir.z;
                    bx = __fmaf_rn(bvh.base[0], local.invDir.x, -local.originDivDir.x);
                    by = __fmaf_rn(bvh.base[1], local.invDir.y, -local.originDivDir.y);
                    bz = __fmaf_rn(bvh.base[2], local.invDir.z, -local.originDivDir.z);
                    park[0] = local.invDir.x; park[RT_BLOCK] = local.invDir.y; park[2 * RT_BLOCK] = local.invDir.z;
                    park[3 * RT_BLOCK] = local.originDivDir.x; park[4 * RT_BLOCK] = local.originDivDir.y; park[5 * RT_BLOCK] = local.originDivDir.z;
                    tol = shadow ? 0.0f : fmaxf(fmaxf(mx, my), mz) * 1.9073486328125e-06f;   // 2^-19: 16 ulps
                    best = maxDistance; second = inf; occluded = false; overflow = false;
                    sp = 0u; cur = 0u;   // node 0 holds the children of the binary tree's root
                    if (kDiag) diagMaxSp = 0u;
                    have = true;
                    tookShadow = shadow;   // (a request handed to the binary-tree kernel is counted there)
                }
            }
            { const uint32_t n = (uint32_t)__popcll(__ballot(tookUntrusted)); numRetraced += n; numUntrusted += n; numShadowRays += (uint32_t)__popcll(__ballot(tookShadow)); }
            continue;
        }
        if ((mI | mO) == 0ull) break;
        if (mI != 0ull && (uint32_t)__popcll(mO) < tune.otherMinLanes)
        {
            // ---- interior phase: four conservative slab tests per step, until enough lanes wait at a leaf or are finished ----
            if (kDiag) { diagPhase = 1u; diagRuns[1]++; }
            bool in = interior;
            const float limit = best + (tol + tol);   // box occlusion with the slack that keeps every candidate within tol of the final hit in the walk
            // which plane of an axis the ray meets first: byte selectors of the slab test, rebuilt per phase (three registers less across the leaf and refill phases)
            const uint32_t selX = ax < 0.0f ? RT_WIDE_SEL_X_NEG : RT_WIDE_SEL_X_POS, selY = ay < 0.0f ? RT_WIDE_SEL_Y_NEG : RT_WIDE_SEL_Y_POS, selZ = az < 0.0f ? RT_WIDE_SEL_Z_NEG : RT_WIDE_SEL_Z_POS;
            // Visiting order (round 6).  A closest-hit ray walks its NEAREST entered child next (hits shorten it).  An any-hit ray has nothing to shorten -- it ends
            // with the first occluder, wherever that lies -- and nearest-first is the worst order for it: a next-event ray starts ON a surface, so the nearest
            // boxes hold that surface's neighbours, which never occlude it.  FARTHEST child first finds the walls and roofs that do: the step model over the
            // benchmark's rays (tools/wide8/walk_model.cpp, profiles/r06_wide8_step_model.txt) gives 9.9 interior + 1.5 leaf visits per any-hit ray instead of
            // 16.0 + 2.6.  Occlusion is an OR over the same candidates: the result does not depend on the order.  Same instruction count: the sort key
            // 0x7FFFFFFF - bits(entry) = 0x7FFFFFFF ^ bits(entry) (entry >= 0: no borrow), and an any-hit lane xors with 0 instead.
            // (The flip is rebuilt in every iteration from `tol`, which is zero for any-hit rays only, behind an optimisation barrier: as a loop-invariant value
            //  it would be one more vector register live across the loop -- the 97th: 20 bytes of scratch -- for three instructions per visit saved.)
            for (;;)
            {
                if (kDiag) { diagSlots++; if (in) diagVisits++; }
                if (in)
                {
                    const float4* p = bvh.nodes + 4u * cur;
                    const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
                    float n0, f0, n1, f1, n2, f2, n3, f3;
                    RT_WIDE_SLAB(q0, n0, f0); RT_WIDE_SLAB(q1, n1, f1); RT_WIDE_SLAB(q2, n2, f2); RT_WIDE_SLAB(q3, n3, f3);
                        // (key, reference) pairs sorted so that the children the ray enters come first -- farthest first for a closest-hit ray, nearest
                        // first for an any-hit ray: the LAST entered one is walked next -- and the ones it misses last: key = orderFlip ^ bits(entry
                        // distance) (the distance is >= 0, so its bits order like the float), miss = all ones
                        const bool h0 = f0 >= n0 && n0 < limit, h1 = f1 >= n1 && n1 < limit, h2 = f2 >= n2 && n2 < limit, h3 = f3 >= n3 && n3 < limit;
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
                        // the first numHit - 1 references are deferred, the last one (the nearest child) is walked next.  The three stores are
                        // unconditional (what lands above the new top is free space; the overflow check keeps three entries in reserve)
                        uint32_t* const top = stack + sp * RT_BLOCK;
                        top[0] = r0; top[RT_BLOCK] = r1; top[2 * RT_BLOCK] = r2;
                        if (numHit != 0u) { cur = numHit == 1u ? r0 : (numHit == 2u ? r1 : (numHit == 3u ? r2 : r3)); sp += numHit - 1u; }
                        else if (sp == 0u) cur = RT_QUANT_DONE;
                        else { --sp; cur = stack[sp * RT_BLOCK]; }
                    if (kDiag && sp > diagMaxSp) diagMaxSp = sp;
                    if (sp + 3u > (uint32_t)kStack) { overflow = true; cur = RT_QUANT_DONE; }   // the next step could not push: the binary-tree kernel takes the ray
                }
                in = in && (cur >> RT_NODE_LEAVES_SHIFT) == 0u;
                const unsigned long long m = __ballot(in);
                if (m == 0ull || 64u - nIdle - (uint32_t)__popcll(m) >= tune.otherMinLanes) break;
            }
        }
        else
        {
        if (kDiag) { diagPhase = 2u; diagRuns[2]++; }
        bool handedOver = false, overflowed = false, uncountShadow = false;
        if (other)
        {
            // ---- leaves (the current one and the one set aside): MeshShape::Traverse_Leaf(_Shadow), MeshShape.cpp:134-207 ----
            for (int once = 0; once < 1; ++once)
            {
                const uint32_t leaf = cur;
                if (!RT_WIDE_IS_LEAF(leaf) || occluded) continue;
                if (kDiag) diagLeaves++;
                const uint32_t numLeaves = leaf >> RT_NODE_LEAVES_SHIFT, first = leaf & RT_NODE_CHILD_MASK;
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
                    if (kDiag) diagGates++;
                    Ray gateRay;   // = the ray transformRayUnsafe built at refill (makeRayUnsafe3 of the same origin and direction: its quotients were parked then)
                    gateRay.origin = ray.origin; gateRay.dir = ray.dir;
                    gateRay.invDir = V4(park[0], park[RT_BLOCK], park[2 * RT_BLOCK], 0.0f);
                    gateRay.originDivDir = V4(park[3 * RT_BLOCK], park[4 * RT_BLOCK], park[5 * RT_BLOCK], 0.0f);
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
                                const bool firstWins = t0 <= t1;   // HitPoint written through (an exact tie is retraced anyway)
                                prec(paths, R_HIT, slot) = f4(fbits(0u), fbits(first + (firstWins ? 0u : 1u)), lo, firstWins ? u0 : u1);
                                prec(paths, R_SAMPLER, slot).x = firstWins ? v0_ : v1;
                                if (kDiag) diagHitWrites++;
                            }
                            else second = fminf(second, lo);
                        }
                    }
                }
            }
            if (occluded) cur = RT_QUANT_DONE;
            if (cur != RT_QUANT_DONE)
            {
                if (sp == 0u) cur = RT_QUANT_DONE;
                else { --sp; cur = stack[sp * RT_BLOCK]; }
            }
            if (cur == RT_QUANT_DONE)
            {
                // ---- finished ----
                if (kDiag) { if (diagMaxSp > 9u) diagDeep[0]++; if (diagMaxSp > 13u) diagDeep[1]++; if (diagMaxSp > 17u) diagDeep[2]++; }
                if (overflow)
                {
                    widePushExact(tune, handOver, shadow, shadow ? light * paths.capacity + slot : slot);
                    handedOver = true; overflowed = true;
                    uncountShadow = shadow;   // counted by the kernel that resolves it
                }
                else if (shadow)
                {
                    if (occluded) pshadow(paths, light, 0, slot).w = -1.0f;   // unoccluded requests are tallied when they are resolved
                }
                else if (best == inf) prec(paths, R_HIT, slot) = f4(fbits(RT_INVALID_OBJECT), fbits(0u), inf, 0.0f);   // HitPoint.h:14-51
                else if (second <= best + tol)
                {
                    widePushExact(tune, handOver, false, slot);   // a runner-up too close to call: the reference's own walk decides
                    handedOver = true;
                }
                have = false;
            }
        }
        numRetraced += (uint32_t)__popcll(__ballot(handedOver)); numOverflow += (uint32_t)__popcll(__ballot(overflowed)); numShadowRays -= (uint32_t)__popcll(__ballot(uncountShadow));
        }
    }
    // counters: shadow rays traced here, rays handed to the binary-tree kernel
    __shared__ uint32_t sTally[4];
    if (threadIdx.x < 4u) sTally[threadIdx.x] = 0u;
    __syncthreads();
    if ((threadIdx.x & 63u) == 0u)
    {
        if (numShadowRays) atomicAdd(&sTally[0], numShadowRays);
        if (numRetraced) atomicAdd(&sTally[1], numRetraced);
        if (numUntrusted) atomicAdd(&sTally[2], numUntrusted);
        if (numOverflow) atomicAdd(&sTally[3], numOverflow);
    }
    __syncthreads();
    if (threadIdx.x == 0u && sTally[0]) atomicAdd(&counters[C_SHADOW], (unsigned long long)sTally[0]);
    if (threadIdx.x == 1u && sTally[1]) atomicAdd(&counters[RT_COUNTER_RETRACED], (unsigned long long)sTally[1]);
    if (!kDiag)
    {
        if (threadIdx.x == 2u && sTally[2]) atomicAdd(&counters[RT_COUNTER_RETRACED + 1], (unsigned long long)sTally[2]);   // diagnostics: untrusted at refill ...
        if (threadIdx.x == 3u && sTally[3]) atomicAdd(&counters[RT_COUNTER_RETRACED + 2], (unsigned long long)sTally[3]);   // ... and stack overflows
    }
    else
    {
        // RTGPU_WIDE_DIAG=1: the three spare counters hold the walk's statistics instead
        const bool deep = tune.localExact == 2u, bytes = tune.localExact == 3u;   // (the diagnostic kernel has no block-local lists: the field carries RTGPU_WIDE_DIAG's mode)
        atomicAdd(&counters[RT_COUNTER_RETRACED + 1], (unsigned long long)(deep ? diagDeep[0] : diagVisits));
        atomicAdd(&counters[RT_COUNTER_RETRACED + 2], (unsigned long long)(bytes ? diagGates : (deep ? diagDeep[1] : diagSlots)));   // 3: exact-box fetches
        atomicAdd(&counters[RT_COUNTER_RETRACED + 3], (unsigned long long)(deep ? diagDeep[2] : diagLeaves));
        // 3: the hit records written through get a 64-bit slot of their own (round 5 packed them into the upper half of the exact-box count, which wraps into
        // them past 2^32 boxes -- a 256-pass bench run is within a factor of two of that): the reference's shadow triangle-test counter, free in this walk; the
        // phase clocks that otherwise ride in these slots are not written in this mode
        if (bytes) atomicAdd(&counters[C_TRI_SHADOW], (unsigned long long)diagHitWrites);
        else if ((threadIdx.x & 63u) == 0u)
        {
            // the reference's intersection counters are not used by this walk: per-wave clocks and phase counts ride in their slots
            atomicAdd(&counters[C_BOX], diagClock[0]); atomicAdd(&counters[C_BOX_PASS], diagClock[1]); atomicAdd(&counters[C_TRI], diagClock[2]);
            atomicAdd(&counters[C_TRI_PASS], (unsigned long long)clock64() - diagStart);
            atomicAdd(&counters[C_BOX_SHADOW], (unsigned long long)diagRuns[1]);
            atomicAdd(&counters[C_TRI_SHADOW], diagExhausted ? diagPrev - diagExhausted : 0ull);   // the wave's drain phase: from the empty queue to its last ray
            atomicAdd(&counters[C_ANALYTIC_HITS], (unsigned long long)diagRuns[0]);   // (free in a mesh-only scene)
            atomicAdd(&counters[C_PRIMARY], diagClaimClock); atomicAdd(&counters[C_SHADOW_HIT], (unsigned long long)diagClaims);   // (k_shade adds to these two as well: subtract a run without RTGPU_WIDE_DIAG)
        }
    }
}


// kLocalExact: the block traces the rays its walk does not decide itself (small frames: rtgpu_set_schedule / launchTraceWide's policy).  A template
// parameter, not a run-time switch: the second walk's state cost the full-frame instantiation 148 bytes of scratch per lane and 2 KB of LDS
// (profiles/r04_kernel_stats_serial.txt against r03's) although a full frame never runs it.
template <int kStack, bool kDiag = false, bool kLocalExact = false>
__global__ void __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(kStack <= 24 ? 5 : 1))) k_trace_wide(const RtSceneDesc scene, const WideBvh bvh, const Paths paths,
                                                         const uint32_t* __restrict__ queue, const uint32_t* __restrict__ queueCount,
                                                         const uint32_t* __restrict__ shadowQueue, const uint32_t* __restrict__ shadowCount,
                                                         uint32_t* __restrict__ cursor, unsigned long long* counters, const WideTuning tune)
{
    // the 4-wide walk's own stack: RT_WIDE_STACK entries per lane (measured on the benchmark frame, RTGPU_WIDE_DIAG=2: 18 of 48.8 M rays ever hold
    // more than 13 deferred children, none more than 17 -- profiles/r05_wide_diag_phases.txt; a ray that would need more goes to the re-trace launch)
    // + the parked words; the block-local second walk runs the reference's binary walk on kStack entries of the same memory
    constexpr int kWords = kLocalExact && kStack > RT_WIDE_STACK + (int)RT_WIDE_PARK ? kStack : RT_WIDE_STACK + (int)RT_WIDE_PARK;
    __shared__ uint32_t sStack[kWords * RT_BLOCK];
    __shared__ uint32_t sDensePrefix[RT_DENSE_SHARDS + 1u];
    if constexpr (!kLocalExact)
    {
        const WideLocal none = { nullptr, nullptr, nullptr, nullptr, 0u };
        traceWideLoop<RT_WIDE_STACK, kDiag>(scene, bvh, paths, queue, queueCount, shadowQueue, shadowCount, cursor, counters, tune, none, sStack, sDensePrefix, gridDim.x * ((uint32_t)RT_BLOCK / 64u));
    }
    else
    {
        __shared__ uint32_t sLocalExact[RT_WIDE_LOCAL_EXACT], sLocalShadow[RT_WIDE_LOCAL_EXACT], sLocalCounts[4];   // counts: closest, any-hit, work cursor of the second walk
        if (threadIdx.x < 4u) sLocalCounts[threadIdx.x] = 0u;
        __syncthreads();
        const WideLocal local = { sLocalExact, &sLocalCounts[0], sLocalShadow, &sLocalCounts[1], RT_WIDE_LOCAL_EXACT };
        traceWideLoop<RT_WIDE_STACK, kDiag>(scene, bvh, paths, queue, queueCount, shadowQueue, shadowCount, cursor, counters, tune, local, sStack, sDensePrefix, gridDim.x * ((uint32_t)RT_BLOCK / 64u));
        // the rays this block's walk did not decide, by the reference's own walk (block-uniform branch: the counts are final behind the walk's barrier)
        __syncthreads();
        if (threadIdx.x < 2u && sLocalCounts[threadIdx.x] > RT_WIDE_LOCAL_EXACT) sLocalCounts[threadIdx.x] = RT_WIDE_LOCAL_EXACT;
        __syncthreads();
        if (sLocalCounts[0] + sLocalCounts[1] != 0u)
        {
            // a degenerate closest-hit ray (an axis-parallel direction: it walks most of the tree) does not keep the block: past RT_ABORT_RETRACE_AFTER rounds
            // it goes to the launch's exact queue, i.e. to the re-trace launch behind this one, which hands it on to k_trace_monster
            const TravTuning exactTune = { tune.refillMinIdle, tune.otherMinLanes, tune.shadowOffset, tune.exactQueue, tune.exactCount, RT_ABORT_RETRACE_AFTER, nullptr, 0u, RT_RETRACE_SPLIT_AFTER };
            traceBinaryLoop<kStack, false>(scene, paths, sLocalExact, &sLocalCounts[0], sLocalShadow, &sLocalCounts[1], &sLocalCounts[2], counters, exactTune, sStack, sDensePrefix,
                                                  (uint32_t)RT_BLOCK / 64u);
        }
    }
}
#endif   // RT_DEVICE_KERNELS

#ifdef RT_HOST_BUILDERS
// ---- host: collapse of the reference's binary tree.  A wide node starts as the two children of a binary node; while it has a free
// slot, its interior child with the largest surface area is replaced by that child's own two children.  Node indices are breadth first.
struct WideBuild
{
    std::vector<float4> nodes;
    bool ok = false;
};

static WideBuild buildWideBvh(const RtNode* nodes, uint32_t numNodes, const QuantBuild& q)
{
    WideBuild w;
    if (!q.ok) return w;
    auto isLeaf = [&](uint32_t n) { return (nodes[n].leaves & 0x3FFFFFFFu) != 0u; };
    auto area = [&](uint32_t n)
    {
        const double ex = (double)nodes[n].max[0] - nodes[n].min[0], ey = (double)nodes[n].max[1] - nodes[n].min[1], ez = (double)nodes[n].max[2] - nodes[n].min[2];
        return ex * ey + ey * ez + ez * ex;
    };
    std::vector<uint32_t> wideOf(numNodes, 0xFFFFFFFFu);   // binary interior node -> wide node
    std::vector<uint32_t> order;                            // wide node -> binary node
    order.push_back(0u); wideOf[0] = 0u;
    std::vector<uint32_t> children;                         // 4 binary node indices per wide node (0xFFFFFFFF: empty)
    for (size_t i = 0; i < order.size(); ++i)
    {
        const uint32_t n = order[i];
        uint32_t list[4]; uint32_t num = 2;
        list[0] = nodes[n].childIndex; list[1] = nodes[n].childIndex + 1u;
        while (num < 4u)
        {
            int pick = -1; double bestArea = -1.0;
            for (uint32_t k = 0; k < num; ++k) if (!isLeaf(list[k]) && area(list[k]) > bestArea) { bestArea = area(list[k]); pick = (int)k; }
            if (pick < 0) break;
            const uint32_t c = nodes[list[pick]].childIndex;
            list[pick] = c; list[num++] = c + 1u;
        }
        for (uint32_t k = 0; k < 4u; ++k)
        {
            const uint32_t child = k < num ? list[k] : 0xFFFFFFFFu;
            children.push_back(child);
            if (child != 0xFFFFFFFFu && !isLeaf(child))
            {
                if (order.size() >= RT_NODE_CHILD_MASK) return w;
                wideOf[child] = (uint32_t)order.size(); order.push_back(child);
            }
        }
    }
    w.nodes.resize(order.size() * 4u);
    for (size_t i = 0; i < order.size(); ++i)
        for (uint32_t k = 0; k < 4u; ++k)
        {
            const uint32_t child = children[i * 4u + k];
            float4 rec = make_float4(0.0f, 0.0f, 0.0f, __builtin_bit_cast(float, (uint32_t)RT_WIDE_EMPTY));   // a point at the grid's corner, outside the mesh's bounds
            if (child != 0xFFFFFFFFu)
            {
                rec = q.pairs[child];   // the child's box on the grid; its packed reference is replaced for interior children
                if (!isLeaf(child)) rec.w = __builtin_bit_cast(float, wideOf[child]);
            }
            w.nodes[i * 4u + k] = rec;
        }
    w.ok = true;
    return w;
}
#endif   // RT_HOST_BUILDERS
