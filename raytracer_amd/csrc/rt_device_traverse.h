// rt_device_traverse.h -- the two-level BVH traversal as a per-lane STATE MACHINE, one node step at a time.
//
// Same per-ray algorithm as the nested loops of the reference (Scene::Traverse -> GenericTraverse<Scene> ->
// Traverse_Object -> GenericTraverse<MeshShape> -> Traverse_Leaf; Core/Scene/Scene.cpp:128-261,
// Core/Traversal/Traversal_Single.h, Core/Shapes/MeshShape.cpp:134-207): every ray performs exactly the same
// sequence of box / triangle / shape tests with the same running hit distance, so hits, tie-breaking and
// the intersection counters are unchanged.  What changes is the shape of the loop.  Because one call
// advances one lane by one step, a wave can
//   (a) refill lanes whose ray has finished with fresh rays from the queue (persistent threads),
//   (b) keep its "nodes to visit" stack in LDS instead of scratch, and
//   (c) run the two kinds of step as SEPARATE wave-wide phases: "interior" steps (two slab tests, ~90 % of all
//       steps, kept branch-light) and "other" steps (leaf triangles, per-object set-up, level exits, finishing).
//       Lanes that reach a leaf wait until enough lanes of the wave are at leaves too, instead of dragging the
//       whole wave through the long triangle code on nearly every iteration (with 64 lanes and ~1 leaf per 16
//       steps, some lane is at a leaf 98 % of the time).
//
// Stack: one uint32 column per lane in LDS, shared by both levels -- the mesh level continues above the
// entries the top level has pushed (meshBase remembers where the mesh's part starts).  An entry is the
// deferred node itself, packed as childIndex | numLeaves << 30, so a pop needs no memory access.
#pragma once

#include "rt_device_core.h"

namespace rtd {

enum TravMode : uint32_t
{
    TRAV_DONE = 0,      // nothing left to do for this ray
    TRAV_TOP_NODE = 1,  // cur is a top-level BVH node
    TRAV_TOP_LEAF = 2,  // iterating objects [leafNext, leafEnd) of a top-level leaf (or the single-object bypass)
    TRAV_MESH = 3       // cur is a node of the current mesh's BVH
};

#define RT_NODE_LEAVES_SHIFT 30u
#define RT_NODE_CHILD_MASK 0x3FFFFFFFu
#define RT_MAX_PACKED_LEAVES 3u    // numLeaves must fit two bits (the reference builds leaves of <= 2, BVHBuilder.h:16)
#define RT_LEVEL_EXHAUSTED 0xFFFFFFFFu   // cur: the current level has no node left (its part of the stack is empty)

RT_DEV uint32_t packNode(uint32_t childIndex, uint32_t leavesWord) { return childIndex | (leavesOf(leavesWord) << RT_NODE_LEAVES_SHIFT); }

struct TravState
{
    Ray ray;            // ray used for the tests of the current level: the world ray, or the mesh's local ray.  The world
                        // ray is NOT kept while a mesh is traversed (12 registers per lane that would cost a whole wave
                        // of occupancy); the caller rebuilds it from the path state when the mesh is left
    float hitDistance;  // running closest distance (the rest of the reference's HitPoint is written through to the
                        // path state by the caller's onHit at every accepted hit, not carried in registers)
    const RtNode* nodes;        // node array of the current level
    uint32_t mode;
    uint32_t cur;               // packed current node, or RT_LEVEL_EXHAUSTED
    uint32_t stackSize, levelBase;   // levelBase: stack size at which the current level is exhausted
    uint32_t leafNext, leafEnd;
    uint32_t objectId;          // object whose mesh is being traversed
    uint32_t triBase;           // offset of the current mesh in triangles[]
    bool nanFree;               // ray of the current level cannot produce NaNs in a slab test (rayIsNaNFree)
    bool shadow;                // any-hit ray (Scene::Traverse_Shadow) instead of closest-hit (Scene::Traverse)
    bool occluded;              // shadow rays: result
};

// LDS stack accessor: column `lane` (threadIdx.x) of a [capacity][blockDim.x] array -> bank = lane % 32 for
// every level, i.e. conflict free whatever levels the lanes are at.
struct LdsStack
{
    uint32_t* base;     // &stack[0][threadIdx.x]
    uint32_t stride;    // blockDim.x
    __device__ __forceinline__ void push(uint32_t& size, uint32_t v) const { base[size * stride] = v; ++size; }
    __device__ __forceinline__ uint32_t pop(uint32_t& size) const { --size; return base[size * stride]; }
};

RT_DEV bool travIsInterior(const TravState& s)
{
    return (s.mode == TRAV_MESH || s.mode == TRAV_TOP_NODE) && (s.cur >> RT_NODE_LEAVES_SHIFT) == 0;
}

// Scene::Traverse / Traverse_Shadow prologue (Scene.cpp:219-261): 0 objects, 1 object (BVH bypass), or the root.
RT_DEV void travBegin(TravState& s, const RtSceneDesc& d, const Ray& worldRay, float maxDistance, bool shadow)
{
    s.ray = worldRay; s.shadow = shadow;
    s.hitDistance = maxDistance;
    s.stackSize = 0; s.levelBase = 0; s.leafNext = 0; s.leafEnd = 0; s.objectId = 0; s.triBase = 0;
    s.cur = 0; s.occluded = false; s.nodes = d.topNodes; s.nanFree = rayIsNaNFree(worldRay);
    if (d.numObjects == 0) s.mode = TRAV_DONE;
    else if (d.numObjects == 1) { s.mode = TRAV_TOP_LEAF; s.leafNext = 0; s.leafEnd = 1; }
    else if (d.numTopNodes == 0) s.mode = TRAV_DONE;
    else { s.mode = TRAV_TOP_NODE; s.cur = packNode(d.topNodes[0].childIndex, d.topNodes[0].leaves); }
}

// INTERIOR step: test both children, descend / push / pop (Traversal_Single.h:44-91 and :127-170).
// Precondition: travIsInterior(s).  kExactMinMax selects the compare+select slab test that reproduces the
// _mm_min_ps/_mm_max_ps NaN behaviour; the caller uses it whenever a lane of the wave is not nanFree.

template <bool kCount, bool kExactMinMax>
RT_DEV void travStepInterior(TravState& s, const LdsStack& stack, Counters& cnt)
{
    const uint32_t firstChild = s.cur & RT_NODE_CHILD_MASK;
    const NodePair n = loadNodePair(s.nodes, firstChild);
    float distanceA, distanceB;
    bool hitA, hitB;
    if (kExactMinMax)
    {
        hitA = intersectBoxRay(s.ray, V4(n.a0.x, n.a0.y, n.a0.z, 0.0f), V4(n.a1.x, n.a1.y, n.a1.z, 0.0f), distanceA);
        hitB = intersectBoxRay(s.ray, V4(n.b0.x, n.b0.y, n.b0.z, 0.0f), V4(n.b1.x, n.b1.y, n.b1.z, 0.0f), distanceB);
        if (!kCount && !s.nanFree && s.mode == TRAV_MESH)
        {
            // an axis-parallel ray INSIDE A MESH: boxes clearly off its fixed coordinate hold nothing it can hit (rt_device_core.h, boxNearDegenerateAxes).
            // Never at the top level (round 6, found by the soak: seed 9606, case 765): the argument needs "whatever is in the box accepts only points of itself",
            // which holds for triangles and NOT for the reference's analytic shapes -- BoxShape::Intersect evaluates inf * 0 for such a ray and reports a hit on a
            // face of a box the ray passes a metre above (BoxShape.cpp:91-130 through Intersect_BoxRay_TwoSided); the reference does that, so this walk must too
            hitA = hitA && boxNearDegenerateAxes(s.ray, n.a0.x, n.a0.y, n.a0.z, n.a1.x, n.a1.y, n.a1.z);
            hitB = hitB && boxNearDegenerateAxes(s.ray, n.b0.x, n.b0.y, n.b0.z, n.b1.x, n.b1.y, n.b1.z);
        }
    }
    else
    {
        hitA = intersectBoxRayNoNaN(s.ray, n.a0.x, n.a0.y, n.a0.z, n.a1.x, n.a1.y, n.a1.z, distanceA);
        hitB = intersectBoxRayNoNaN(s.ray, n.b0.x, n.b0.y, n.b0.z, n.b1.x, n.b1.y, n.b1.z, distanceB);
    }
    hitA = hitA && (distanceA < s.hitDistance);   // box occlusion
    hitB = hitB && (distanceB < s.hitDistance);
    if (kCount)
    {
        cnt.c[C_BOX_SHADOW] += s.shadow ? 2u : 0u;
        cnt.c[C_BOX] += s.shadow ? 0u : 2u;
        cnt.c[C_BOX_PASS] += s.shadow ? 0u : ((hitA ? 1u : 0u) + (hitB ? 1u : 0u));
    }
    const uint32_t a = packNode(__float_as_uint(n.a0.w), __float_as_uint(n.a1.w));
    const uint32_t b = packNode(__float_as_uint(n.b0.w), __float_as_uint(n.b1.w));
    const bool both = hitA && hitB;
    const bool swap = !s.shadow && both && (distanceB < distanceA);   // closest: nearer child first; any-hit: A first
    if (both) stack.push(s.stackSize, swap ? a : b);
    if (hitA || hitB) s.cur = (hitA && !swap) ? a : b;
    else if (s.stackSize == s.levelBase) s.cur = RT_LEVEL_EXHAUSTED;        // handled by the "other" phase
    else s.cur = stack.pop(s.stackSize);
}

// next node of the current level after a leaf, or RT_LEVEL_EXHAUSTED
RT_DEV void travNext(TravState& s, const LdsStack& stack)
{
    if (s.stackSize == s.levelBase) s.cur = RT_LEVEL_EXHAUSTED;
    else s.cur = stack.pop(s.stackSize);
}

// OTHER step: a mesh leaf, a level exit, a top-level leaf header, or the next object of a top-level leaf.
// Precondition: s.mode != TRAV_DONE && !travIsInterior(s).  reloadWorldRay() returns the ray travBegin was given;
// onHit(objectId, subObjectId, distance, u, v) records a new closest hit (closest-hit rays only).
template <bool kCount, typename ReloadWorldRay, typename OnHit>
RT_DEV void travStepOther(TravState& s, const RtSceneDesc& d, const LdsStack& stack, Counters& cnt, ReloadWorldRay reloadWorldRay, OnHit onHit)
{
    if (s.mode == TRAV_MESH)
    {
        if (s.cur != RT_LEVEL_EXHAUSTED)
        {
            // MeshShape::Traverse_Leaf / Traverse_Leaf_Shadow
            const uint32_t numLeaves = s.cur >> RT_NODE_LEAVES_SHIFT;
            const uint32_t first = s.cur & RT_NODE_CHILD_MASK;
            if (kCount) { cnt.c[C_TRI_SHADOW] += s.shadow ? numLeaves : 0u; cnt.c[C_TRI] += s.shadow ? 0u : numLeaves; }
            const RtTriangle* tris = d.triangles + s.triBase;
            // the second triangle of the leaf (adjacent in memory) is fetched with the first: one memory round trip per leaf
            V4 v0, e1, e2, nv0, ne1, ne2;
            loadTriangle(tris + first, v0, e1, e2);
            loadTriangle(tris + first + (numLeaves > 1u ? 1u : 0u), nv0, ne1, ne2);
            for (uint32_t i = 0; i < numLeaves; ++i)
            {
                const uint32_t triangleIndex = first + i;
                if (i == 1u) { v0 = nv0; e1 = ne1; e2 = ne2; }
                else if (i > 1u) loadTriangle(tris + triangleIndex, v0, e1, e2);
                float u, v, dist;
                if (intersectTriangleRay(s.ray, v0, e1, e2, u, v, dist))
                {
                    if (dist < s.hitDistance)
                    {
                        s.hitDistance = dist;
                        if (s.shadow) { s.occluded = true; s.mode = TRAV_DONE; return; }
                        onHit(s.objectId, triangleIndex, dist, u, v);
                        if (kCount) cnt.c[C_TRI_PASS]++;
                    }
                }
            }
            travNext(s, stack);
            if (s.cur != RT_LEVEL_EXHAUSTED) return;
        }
        // GenericTraverse<MeshShape> returned: back to the object loop of the top-level leaf, in world space
        // (falls through to the next object of the leaf: one "other" step less per mesh visit)
        s.nodes = d.topNodes; s.levelBase = 0;
        s.mode = TRAV_TOP_LEAF;
        if (d.numObjects == 1) { s.mode = TRAV_DONE; return; }   // the bypass path: that was the only object
        s.ray = reloadWorldRay(); s.nanFree = rayIsNaNFree(s.ray);
    }
    if (s.mode == TRAV_TOP_NODE)
    {
        if (s.cur == RT_LEVEL_EXHAUSTED) { s.mode = TRAV_DONE; return; }
        // Scene::Traverse_Leaf(_Shadow): objects [first, first + numLeaves)
        s.leafNext = s.cur & RT_NODE_CHILD_MASK; s.leafEnd = s.leafNext + (s.cur >> RT_NODE_LEAVES_SHIFT);
        s.mode = TRAV_TOP_LEAF;
    }
    // TRAV_TOP_LEAF: Scene::Traverse_Object / Traverse_Object_Shadow for the next object of the leaf
    if (s.leafNext >= s.leafEnd)
    {
        if (d.numObjects == 1) { s.mode = TRAV_DONE; return; }   // the bypass path has no stack
        travNext(s, stack);
        s.mode = (s.cur == RT_LEVEL_EXHAUSTED) ? TRAV_DONE : TRAV_TOP_NODE;
        return;
    }
    const uint32_t objectID = s.leafNext++;
    const RtObject& obj = d.objects[objectID];
    const Ray lray = transformRayUnsafe(loadM4(obj.invTransform), s.ray);   // s.ray is the world ray here
    if (obj.objectKind == RT_OBJECT_LIGHT)
    {
        float lightDistance;
        if (lightTestRayHit(d.lights[obj.lightIndex], lray, lightDistance))
        {
            if (s.shadow)
            {
                if (lightDistance < s.hitDistance) { s.hitDistance = lightDistance; s.occluded = true; s.mode = TRAV_DONE; }
            }
            else if (lightDistance > 0.0f && lightDistance < s.hitDistance)
            {
                s.hitDistance = lightDistance; onHit(objectID, RT_LIGHT_OBJECT, lightDistance, 0.0f, 0.0f);   // u, v: mesh hits only
            }
        }
        return;
    }
    if (obj.shapeKind == RT_SHAPE_MESH)
    {
        const RtMesh& mesh = d.meshes[obj.meshIndex];
        if (mesh.numNodes == 0) return;
        s.ray = lray; s.nanFree = rayIsNaNFree(lray);
        s.objectId = objectID; s.triBase = mesh.firstTriangle;
        s.nodes = d.meshNodes + mesh.firstNode;
        s.levelBase = s.stackSize;
        s.cur = packNode(s.nodes[0].childIndex, s.nodes[0].leaves);
        s.mode = TRAV_MESH;
        return;
    }
    ShapeHit sh;
    if (shapeIntersect(obj.shapeKind, obj.shapeParam, lray, sh))
    {
        if (s.shadow)
        {
            if (sh.farDist > 0.0f && sh.nearDist < s.hitDistance) { s.occluded = true; s.mode = TRAV_DONE; }
        }
        else if (sh.nearDist > 0.0f && sh.nearDist < s.hitDistance) { s.hitDistance = sh.nearDist; onHit(objectID, sh.subObjectId, sh.nearDist, 0.0f, 0.0f); }
        else if (sh.farDist > 0.0f && sh.farDist < s.hitDistance) { s.hitDistance = sh.farDist; onHit(objectID, sh.subObjectId, sh.farDist, 0.0f, 0.0f); }
    }
}

} // namespace rtd
