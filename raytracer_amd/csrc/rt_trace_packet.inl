// rt_trace_packet.inl -- the camera rays of a dense batch as PACKETS: one wave = 64 consecutive paths of bounce 0 = an 8 x 8 pixel block of one pass
// (the slot -> pixel table), walking the 4-wide tree of rt_trace_wide.inl TOGETHER.  Included by rt_trace.hip behind rt_trace_wide.inl.
//
// Why.  k_trace_wide's step is paced by its dependent node fetch (profiles/r04_interior_step_probes.txt): every lane fetches its own 64-byte node, a
// wave step ends when the last of ~36 lines has arrived.  Camera rays of neighbouring pixels visit nearly the same nodes, so here the NODE IS UNIFORM:
// the wave keeps one current node and one stack (wave-uniform values: scalar registers, the stack in the lanes of two vector registers), a node is one
// scalar load of 64 bytes through the constant cache, a leaf's triangles and exact box likewise, and every lane tests its own ray against them -- no
// divergent fetch, no per-lane stack, no idle lanes inside a step.  A child is entered if ANY lane's ray enters it; the order is that of the first lane
// that does.  The first launch of a batch is 10 % of its traversal time (profiles/r03_timeline_serial_start_of_round.txt).
// Measured (profiles/r04_packet_ab.txt): 20.5 interior + 3.8 leaf steps per packet on the benchmark frame (one ray: 17 + 2), first launch of a five-pass batch
// 1.41 -> 1.02-1.2 ms, +2 % end to end; the launch is bound by the ~1.2 G instructions it issues (~240 per step, vector and scalar alike).
//
// Exactness: k_trace_wide's argument unchanged.  A lane sees a SUPERSET of the leaves its own walk would visit (the packet's union), in another order;
// a leaf's triangles count for a lane only behind the leaf's exact box and the `lo < best + tol` test, every candidate has the reference's (t, u, v), the
// runner-up is tracked per lane, and a lane whose runner-up is within tol of its best -- or whose ray is not trusted (zero direction component, far origin)
// -- goes to the exact queue for the reference's own walk (the k_trace launch behind this one).  Any-hit rays never come here (bounce 0 has none).

#ifdef RT_DEVICE_KERNELS
#define RT_PACKET_CLAIM 4u            // packets a wave claims per atomic (1: +1 ms per launch in atomics; 16: uneven tails; profiles/r04_packet_ab.txt)
#define RT_PACKET_STACK 128u          // entries: two vector registers' lanes (a 4-wide tree of depth d defers at most 3 d nodes)
typedef uint32_t PacketU4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) PacketU4* PacketConst4;    // constant address space: a uniform address is a scalar load
typedef const __attribute__((address_space(4))) float* PacketConstF;

// a 16-bit plane of a node word in a SCALAR register as a float: the sub-word select of the conversion reads the scalar register directly (no unpacking)
RT_DEV float packetPlaneLo(uint32_t w) { float f; asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(f) : "s"(w)); return f; }
RT_DEV float packetPlaneHi(uint32_t w) { float f; asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f) : "s"(w)); return f; }
#define RT_PACKET_CE(ka, ra, kb, rb) { if (ka > kb) { const uint32_t tk_ = ka, tr_ = ra; ka = kb; ra = rb; kb = tk_; rb = tr_; } }

__global__ void __launch_bounds__(RT_BLOCK) k_trace_packet(const RtSceneDesc scene, const WideBvh bvh, const Paths paths, uint32_t* __restrict__ cursor, unsigned long long* counters, const WideTuning tune)
{
    __shared__ uint32_t sDensePrefix[RT_DENSE_SHARDS + 1u];
    __shared__ uint32_t sTally[4];
    denseLoadPrefix(tune.denseCounts, sDensePrefix);
    if (threadIdx.x < 4u) sTally[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t count = sDensePrefix[RT_DENSE_SHARDS];
    const uint32_t lane = threadIdx.x & 63u;
    const M4 invTransform = loadM4(scene.objects[0].invTransform);
    const uint32_t firstTriangle = scene.meshes[scene.objects[0].meshIndex].firstTriangle;
    const PacketConst4 nodes = (PacketConst4)(uintptr_t)bvh.nodes;
    const PacketConst4 gates = (PacketConst4)(uintptr_t)bvh.gate;
    const PacketConstF triangles = (PacketConstF)(uintptr_t)(scene.triangles + firstTriangle);
    const float inf = __uint_as_float(0x7f800000u);
    uint32_t numRetraced = 0, numUntrusted = 0, numOverflow = 0;

    for (;;)
    {
        uint32_t base = 0u;
        if (lane == 0u) base = atomicAdd(cursor, 64u * RT_PACKET_CLAIM);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= count) break;
        for (uint32_t first = base; first < base + 64u * RT_PACKET_CLAIM && first < count; first += 64u)
        {
            // ---- the packet's rays: k_trace_wide's refill for closest-hit rays, word for word ----
            const uint32_t idx = first + lane;
            const bool valid = idx < count;
            uint32_t slot = 0u;
            bool act = false;
            float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, ax = 0, ay = 0, az = 0, bx = 0, by = 0, bz = 0, tol = 0;
            float best = inf, second = inf;
            if (valid)
            {
                slot = denseLiveSlot(sDensePrefix, tune.denseShardCapacity, idx);
                const float4 origin = ldStream(prec(paths, R_ORIGIN, slot)), dir = ldStream(prec(paths, R_DIR, slot));
                Ray world = makeRay(V4(origin.x, origin.y, origin.z, 0.0f), V4(dir.x, dir.y, dir.z, 0.0f));
                if ((ubits(origin.w) & 0xFFu) != 0u) world.origin = world.origin + world.dir * 0.001f;
                const Ray local = makeRayUnsafe3(transformPoint(invTransform, world.origin), transformVector(invTransform, world.dir));
                const float mx = fabsf(local.originDivDir.x) + bvh.bound[0] * fabsf(local.invDir.x);
                const float my = fabsf(local.originDivDir.y) + bvh.bound[1] * fabsf(local.invDir.y);
                const float mz = fabsf(local.originDivDir.z) + bvh.bound[2] * fabsf(local.invDir.z);
                const float fold = 4.76837158203125e-07f;   // 2^-21
                const bool trusted = rayIsNaNFree(local) &&
                                     mx * fold < bvh.step[0] * fabsf(local.invDir.x) && my * fold < bvh.step[1] * fabsf(local.invDir.y) && mz * fold < bvh.step[2] * fabsf(local.invDir.z);
                if (!trusted)
                {
                    tune.exactQueue[atomicAdd(tune.exactCount, 1u)] = slot;
                    numRetraced++; numUntrusted++;
                }
                else
                {
                    ox = local.origin.x; oy = local.origin.y; oz = local.origin.z; dx = local.dir.x; dy = local.dir.y; dz = local.dir.z;
                    ax = bvh.step[0] * local.invDir.x; ay = bvh.step[1] * local.invDir.y; az = bvh.step[2] * local.invDir.z;
                    bx = __fmaf_rn(bvh.base[0], local.invDir.x, -local.originDivDir.x);
                    by = __fmaf_rn(bvh.base[1], local.invDir.y, -local.originDivDir.y);
                    bz = __fmaf_rn(bvh.base[2], local.invDir.z, -local.originDivDir.z);
                    tol = fmaxf(fmaxf(mx, my), mz) * 1.9073486328125e-06f;   // 2^-19: 16 ulps
                    act = true;
                }
            }
            bool overflow = false;
            if (__ballot(act) != 0ull)
            {
                Ray ray; ray.origin = V4(ox, oy, oz, 0.0f); ray.dir = V4(dx, dy, dz, 0.0f);
                const Ray gateRay = makeRayUnsafe3(ray.origin, ray.dir);   // = the ray transformRayUnsafe built
                // ---- the shared walk: cur, sp and the stack are wave-uniform ----
                uint32_t stackLo = 0u, stackHi = 0u;   // entry e < 64 in lane e of stackLo, else in lane e - 64 of stackHi
                uint32_t sp = 0u, cur = 0u;            // node 0 holds the children of the binary tree's root
                for (;;)
                {
                    bool pop = true;
                    if ((cur >> RT_NODE_LEAVES_SHIFT) == 0u)
                    {
                        const float limit = best + (tol + tol);
                        uint32_t k0, k1, k2, k3, r0, r1, r2, r3;
                        const PacketU4 q0 = nodes[4u * cur], q1 = nodes[4u * cur + 1u], q2 = nodes[4u * cur + 2u], q3 = nodes[4u * cur + 3u];   // one 64-byte scalar load
#define RT_PACKET_CHILD(q, key, ref)                                                                                                             \
                        {                                                                                                                        \
                            const float nx0 = __fmaf_rn(packetPlaneLo(q.x), ax, bx), nx1 = __fmaf_rn(packetPlaneHi(q.y), ax, bx);            \
                            const float ny0 = __fmaf_rn(packetPlaneHi(q.x), ay, by), ny1 = __fmaf_rn(packetPlaneLo(q.z), ay, by);            \
                            const float nz0 = __fmaf_rn(packetPlaneLo(q.y), az, bz), nz1 = __fmaf_rn(packetPlaneHi(q.z), az, bz);            \
                            const float n = fmaxf(fmaxf(fminf(nx0, nx1), fminf(ny0, ny1)), fmaxf(fminf(nz0, nz1), 0.0f));                         \
                            const float f = fminf(fminf(fmaxf(nx0, nx1), fmaxf(ny0, ny1)), fmaxf(nz0, nz1));                                      \
                            const unsigned long long m = __ballot(act && f >= n && n < limit);                                                   \
                            ref = q.w;                                                                                                           \
                            key = 0xFFFFFFFFu;                                                                                                   \
                            if (m != 0ull) key = 0x7FFFFFFFu - (uint32_t)__builtin_amdgcn_readlane((int)ubits(n), __ffsll((long long)m) - 1);    \
                        }
                        RT_PACKET_CHILD(q0, k0, r0) RT_PACKET_CHILD(q1, k1, r1) RT_PACKET_CHILD(q2, k2, r2) RT_PACKET_CHILD(q3, k3, r3)
#undef RT_PACKET_CHILD
                        // entered children first, farthest first (the nearest is walked next), the others last: k_trace_wide's order, decided by the first lane inside each child
                        RT_PACKET_CE(k0, r0, k1, r1) RT_PACKET_CE(k2, r2, k3, r3) RT_PACKET_CE(k0, r0, k2, r2) RT_PACKET_CE(k1, r1, k3, r3) RT_PACKET_CE(k1, r1, k2, r2)
                        const uint32_t numHit = (k0 != 0xFFFFFFFFu ? 1u : 0u) + (k1 != 0xFFFFFFFFu ? 1u : 0u) + (k2 != 0xFFFFFFFFu ? 1u : 0u) + (k3 != 0xFFFFFFFFu ? 1u : 0u);
                        if (numHit != 0u)
                        {
                            if (sp + 3u > RT_PACKET_STACK) { overflow = true; break; }
#define RT_PACKET_PUSH(r) { if (sp < 64u) stackLo = lane == sp ? (r) : stackLo; else stackHi = lane == sp - 64u ? (r) : stackHi; ++sp; }   // (v_writelane_b32 cannot take value and lane from two scalar registers)
                            if (numHit > 1u) RT_PACKET_PUSH(r0)
                            if (numHit > 2u) RT_PACKET_PUSH(r1)
                            if (numHit > 3u) RT_PACKET_PUSH(r2)
#undef RT_PACKET_PUSH
                            cur = numHit == 1u ? r0 : (numHit == 2u ? r1 : (numHit == 3u ? r2 : r3));
                            pop = false;
                        }
                    }
                    else if (RT_WIDE_IS_LEAF(cur))
                    {
                        // ---- a leaf of the reference's tree: MeshShape::Traverse_Leaf, MeshShape.cpp:134-168, every lane against the same one or two triangles ----
                        const uint32_t numLeaves = cur >> RT_NODE_LEAVES_SHIFT, firstTri = cur & RT_NODE_CHILD_MASK;
                        const PacketConstF t = triangles + 9u * firstTri;
                        const V4 v0(t[0], t[1], t[2], 0.0f), e1(t[3], t[4], t[5], 0.0f), e2(t[6], t[7], t[8], 0.0f);
                        float u0, v0_, t0, u1 = 0.0f, v1 = 0.0f, t1 = inf;
                        if (!intersectTriangleRay(ray, v0, e1, e2, u0, v0_, t0)) t0 = inf;
                        if (numLeaves > 1u)
                        {
                            const V4 nv0(t[9], t[10], t[11], 0.0f), ne1(t[12], t[13], t[14], 0.0f), ne2(t[15], t[16], t[17], 0.0f);
                            if (!intersectTriangleRay(ray, nv0, ne1, ne2, u1, v1, t1)) t1 = inf;
                        }
                        const float lo = fminf(t0, t1);
                        const bool candidate = act && lo < best + tol;
                        if (__ballot(candidate) != 0ull)
                        {
                            const PacketU4 gmin = gates[2u * firstTri], gmax = gates[2u * firstTri + 1u];
                            float nearD;
                            const bool pass = intersectBoxRayNoNaN(gateRay, __uint_as_float(gmin.x), __uint_as_float(gmin.y), __uint_as_float(gmin.z),
                                                                   __uint_as_float(gmax.x), __uint_as_float(gmax.y), __uint_as_float(gmax.z), nearD);
                            if (candidate && pass)
                            {
                                const float hi = fmaxf(t0, t1);
                                if (lo < best)
                                {
                                    second = fminf(best, hi);
                                    best = lo;
                                    const bool firstWins = t0 <= t1;   // HitPoint written through (an exact tie is retraced anyway)
                                    prec(paths, R_HIT, slot) = f4(fbits(0u), fbits(firstTri + (firstWins ? 0u : 1u)), lo, firstWins ? u0 : u1);
                                    prec(paths, R_SAMPLER, slot).x = firstWins ? v0_ : v1;
                                }
                                else second = fminf(second, lo);
                            }
                        }
                    }
                    // (else: an unused child slot whose corner point a ray happened to meet)
                    if (pop)
                    {
                        if (sp == 0u) break;
                        --sp;
                        cur = sp < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)stackLo, (int)sp) : (uint32_t)__builtin_amdgcn_readlane((int)stackHi, (int)(sp - 64u));
                    }
                }
            }
            // ---- finished ----
            if (act)
            {
                if (overflow) { tune.exactQueue[atomicAdd(tune.exactCount, 1u)] = slot; numRetraced++; numOverflow++; }
                else if (best == inf) prec(paths, R_HIT, slot) = f4(fbits(RT_INVALID_OBJECT), fbits(0u), inf, 0.0f);   // HitPoint.h:14-51
                else if (second <= best + tol) { tune.exactQueue[atomicAdd(tune.exactCount, 1u)] = slot; numRetraced++; }   // a runner-up too close to call
            }
        }
    }
    if (numRetraced) atomicAdd(&sTally[1], numRetraced);
    if (numUntrusted) atomicAdd(&sTally[2], numUntrusted);
    if (numOverflow) atomicAdd(&sTally[3], numOverflow);
    __syncthreads();
    if (threadIdx.x == 1u && sTally[1]) atomicAdd(&counters[RT_COUNTER_RETRACED], (unsigned long long)sTally[1]);
    if (threadIdx.x == 2u && sTally[2]) atomicAdd(&counters[RT_COUNTER_RETRACED + 1], (unsigned long long)sTally[2]);
    if (threadIdx.x == 3u && sTally[3]) atomicAdd(&counters[RT_COUNTER_RETRACED + 2], (unsigned long long)sTally[3]);
}
#undef RT_PACKET_CE
#endif   // RT_DEVICE_KERNELS
