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
//       steps) and "other" steps (leaf triangles, per-object set-up, finishing).  Lanes that reach a leaf wait
//       until enough lanes of the wave are at leaves too, instead of dragging the whole wave through the long
//       triangle code on nearly every iteration (with 64 lanes and ~1 leaf per 16 steps, some lane is at a
//       leaf 98 % of the time).
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
#define RT_MAX_PACKED_LEAVES 3u   // numLeaves must fit two bits (the reference builds leaves of <= 2, BVHBuilder.h:16)

RT_DEV uint32_t packNode(uint32_t childIndex, uint32_t leavesWord) { return childIndex | (leavesOf(leavesWord) << RT_NODE_LEAVES_SHIFT); }

struct TravState
{
    Ray ray;            // ray used for the tests of the current level (world ray, or the object's local ray)
    Ray worldRay;       // saved world ray while inside a mesh
    Hit hit;
    const RtNode* nodes;        // node array of the current level
    uint32_t mode;
    uint32_t cur;               // packed current node
    uint32_t stackSize, meshBase;
    uint32_t leafNext, leafEnd;
    uint32_t objectId;          // object whose mesh is being traversed
    uint32_t triBase;           // offset of the current mesh in triangles[]
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
RT_DEV void travBegin(TravState& s, const RtSceneDesc& d, const Ray& worldRay, float maxDistance)
{
    s.ray = worldRay; s.worldRay = worldRay;
    s.hit.objectId = RT_INVALID_OBJECT; s.hit.subObjectId = 0; s.hit.distance = maxDistance; s.hit.u = 0.0f; s.hit.v = 0.0f;
    s.stackSize = 0; s.meshBase = 0; s.leafNext = 0; s.leafEnd = 0; s.objectId = 0; s.triBase = 0;
    s.cur = 0; s.occluded = false; s.nodes = d.topNodes;
    if (d.numObjects == 0) s.mode = TRAV_DONE;
    else if (d.numObjects == 1) { s.mode = TRAV_TOP_LEAF; s.leafNext = 0; s.leafEnd = 1; }
    else if (d.numTopNodes == 0) s.mode = TRAV_DONE;
    else { s.mode = TRAV_TOP_NODE; s.cur = packNode(d.topNodes[0].childIndex, d.topNodes[0].leaves); }
}

// next node of the current level, or leave the level when its part of the stack is empty
RT_DEV void travPop(TravState& s, const RtSceneDesc& d, const LdsStack& stack)
{
    if (s.mode == TRAV_MESH)
    {
        if (s.stackSize == s.meshBase)
        {
            // GenericTraverse<MeshShape> returned: back to the object loop of the top-level leaf, in world space
            s.ray = s.worldRay; s.nodes = d.topNodes;
            s.mode = TRAV_TOP_LEAF;
            return;
        }
        s.cur = stack.pop(s.stackSize);
        return;
    }
    if (s.stackSize == 0) { s.mode = TRAV_DONE; return; }
    s.cur = stack.pop(s.stackSize);
    s.mode = TRAV_TOP_NODE;
}

// INTERIOR step: test both children, descend / push / pop (Traversal_Single.h:44-91 and :127-170).
// Precondition: travIsInterior(s).
template <bool kShadow>
RT_DEV void travStepInterior(TravState& s, const RtSceneDesc& d, const LdsStack& stack, Counters& cnt)
{
    const uint32_t firstChild = s.cur & RT_NODE_CHILD_MASK;
    const ChildTest t = testChildren<kShadow>(s.nodes, firstChild, s.ray, s.hit.distance, cnt);
    const uint32_t a = packNode(t.aChild, t.aLeaves), b = packNode(t.bChild, t.bLeaves);
    if (t.hitA && t.hitB)
    {
        const bool swap = kShadow ? false : (t.distanceB < t.distanceA);   // closest: nearer child first; any-hit: A first
        stack.push(s.stackSize, swap ? a : b);
        s.cur = swap ? b : a;
        return;
    }
    if (t.hitA) { s.cur = a; return; }
    if (t.hitB) { s.cur = b; return; }
    travPop(s, d, stack);
}

// OTHER step: a mesh leaf, a top-level leaf header, or the next object of a top-level leaf.
// Precondition: s.mode != TRAV_DONE && !travIsInterior(s).
template <bool kShadow>
RT_DEV void travStepOther(TravState& s, const RtSceneDesc& d, const LdsStack& stack, Counters& cnt)
{
    if (s.mode == TRAV_MESH)
    {
        // MeshShape::Traverse_Leaf / Traverse_Leaf_Shadow
        const uint32_t numLeaves = s.cur >> RT_NODE_LEAVES_SHIFT;
        const uint32_t first = s.cur & RT_NODE_CHILD_MASK;
        cnt.c[kShadow ? C_TRI_SHADOW : C_TRI] += numLeaves;
        const RtTriangle* tris = d.triangles + s.triBase;
        for (uint32_t i = 0; i < numLeaves; ++i)
        {
            const uint32_t triangleIndex = first + i;
            V4 v0, e1, e2; loadTriangle(tris + triangleIndex, v0, e1, e2);
            float u, v, dist;
            if (intersectTriangleRay(s.ray, v0, e1, e2, u, v, dist))
            {
                if (dist < s.hit.distance)
                {
                    s.hit.distance = dist;
                    if (kShadow) { s.occluded = true; s.mode = TRAV_DONE; return; }
                    s.hit.subObjectId = triangleIndex; s.hit.objectId = s.objectId; s.hit.u = u; s.hit.v = v;
                    cnt.c[C_TRI_PASS]++;
                }
            }
        }
        travPop(s, d, stack);
        return;
    }
    if (s.mode == TRAV_TOP_NODE)
    {
        // Scene::Traverse_Leaf(_Shadow): objects [first, first + numLeaves)
        s.leafNext = s.cur & RT_NODE_CHILD_MASK; s.leafEnd = s.leafNext + (s.cur >> RT_NODE_LEAVES_SHIFT);
        s.mode = TRAV_TOP_LEAF;
    }
    // TRAV_TOP_LEAF: Scene::Traverse_Object / Traverse_Object_Shadow for the next object of the leaf
    if (s.leafNext >= s.leafEnd)
    {
        if (d.numObjects == 1) { s.mode = TRAV_DONE; return; }   // the bypass path has no stack
        travPop(s, d, stack);
        return;
    }
    const uint32_t objectID = s.leafNext++;
    const RtObject& obj = d.objects[objectID];
    const Ray lray = transformRayUnsafe(loadM4(obj.invTransform), s.worldRay);
    if (obj.objectKind == RT_OBJECT_LIGHT)
    {
        float lightDistance;
        if (lightTestRayHit(d.lights[obj.lightIndex], lray, lightDistance))
        {
            if (kShadow)
            {
                if (lightDistance < s.hit.distance) { s.hit.distance = lightDistance; s.occluded = true; s.mode = TRAV_DONE; }
            }
            else if (lightDistance > 0.0f && lightDistance < s.hit.distance)
            {
                s.hit.distance = lightDistance; s.hit.objectId = objectID; s.hit.subObjectId = RT_LIGHT_OBJECT;
            }
        }
        return;
    }
    if (obj.shapeKind == RT_SHAPE_MESH)
    {
        const RtMesh& mesh = d.meshes[obj.meshIndex];
        if (mesh.numNodes == 0) return;
        s.ray = lray;
        s.objectId = objectID; s.triBase = mesh.firstTriangle;
        s.nodes = d.meshNodes + mesh.firstNode;
        s.meshBase = s.stackSize;
        s.cur = packNode(s.nodes[0].childIndex, s.nodes[0].leaves);
        s.mode = TRAV_MESH;
        return;
    }
    ShapeHit sh;
    if (shapeIntersect(obj.shapeKind, obj.shapeParam, lray, sh))
    {
        if (kShadow)
        {
            if (sh.farDist > 0.0f && sh.nearDist < s.hit.distance) { s.occluded = true; s.mode = TRAV_DONE; }
        }
        else if (sh.nearDist > 0.0f && sh.nearDist < s.hit.distance) { s.hit.distance = sh.nearDist; s.hit.objectId = objectID; s.hit.subObjectId = sh.subObjectId; }
        else if (sh.farDist > 0.0f && sh.farDist < s.hit.distance) { s.hit.distance = sh.farDist; s.hit.objectId = objectID; s.hit.subObjectId = sh.subObjectId; }
    }
}

} // namespace rtd
