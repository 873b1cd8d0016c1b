// rt_wide_grid.inl -- what the 4-wide walks (rt_trace_wide.inl, rt_trace_wide2.inl, rt_trace_packet.inl) share: the 16-bit grid of conservative boxes, the
// per-leaf exact boxes ("gates") and the exactness argument below.  Included by rt_trace.hip / rt_tail.hip (RT_DEVICE_KERNELS) and rt_runtime.hip (tree
// builders: RT_HOST_BUILDERS).
// (Round 2 also had a kernel here, k_trace_quant: the reference's BINARY tree with its child pairs re-encoded in 32 bytes -- two accesses per visit instead
// of four, no decode step.  Bit-exact, and no faster than k_trace: 182 vs 183 ms, the twelve conversions per visit ate what the L1 gave back.  Removed in
// round 5 with the other measured-slower opt-in paths; the patch that restores it is profiles/r05_removed_optin_paths.patch.)
//
// The re-encoding:
//   * a child record is 16 bytes {min.xyz, max.xyz as 16-bit grid coordinates, child reference}; the grid spans the mesh's bounds, planes are
//     rounded OUTWARDS with a step to spare, so a stored box always contains the reference's box;
//   * the slab test needs no decode step: t = fma(float(q), step * invDir, base * invDir - origin * invDir), two constants per axis and ray;
//   * boxes that are only conservative cannot decide what the reference tests, so a leaf's triangles count only if the ray also passes the
//     leaf's EXACT box (the test the reference's walk performs before it reaches them) -- fetched only when a triangle was actually hit.
//
// Exactness.  Same argument as for any walk that visits a superset of the reference's leaves in another order: every candidate hit
// (a triangle the ray intersects inside a leaf whose exact box it passes) has the same (t, u, v) as in the reference's walk, because
// the triangle test and the exact box test are the reference's arithmetic; the slab test is monotone in the box planes, so passing a
// leaf's exact box implies passing every ancestor's box with a smaller entry distance, i.e. the reference's walk reaches exactly these
// candidates unless its running hit distance culls one -- which can only change the result when two candidates are closer together than
// the disagreement between a box's entry distance and its triangle's hit distance.  The walks cull with a slack (near < best + 2 tol),
// track the SECOND smallest candidate distance, and a ray whose runner-up lies within tol of its best (tol = 16 ulps of the largest
// term of its slab tests) is not trusted: it goes to the exact queue and is traced again by k_trace in the reference's order.  So do
// rays with a zero direction component (their slab tests produce NaNs, which the reference's min/max operand order resolves in its
// own way) and rays that start so far outside the mesh that the folded slab test's rounding could eat the spare grid step.  Any-hit rays
// need no runner-up: occlusion is an OR over the same candidate set (their leaf gate includes the reference's entry-distance test
// against the fixed ray length).  The reference's box / triangle test counters belong to its own walk: with the intersection
// counters on, k_trace runs alone.

#define RT_QUANT_DONE 0xFFFFFFFFu   // cur: the ray is finished (same value as RT_LEVEL_EXHAUSTED: the mesh level has no node left)
#define RT_QUANT_GRID 65535.0f

#ifdef RT_HOST_BUILDERS
// ---- host: the reference's binary BVH (BVH::Node, 32 bytes, children adjacent) re-encoded ----
struct QuantBuild
{
    std::vector<float4> pairs, gate;   // pairs: one 16-byte record per BINARY node, host-side only -- buildWideBvh / buildWide2 copy them into the 4-wide nodes they upload; gate: uploaded
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
