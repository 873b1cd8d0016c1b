// Binary BVH container.  Node layout is bit-identical to the reference's rt::BVH::Node
// (Core/BVH/BVH.h:22-30) and to RtNode of include/rtgpu.h, so node arrays upload without repacking.
#pragma once

#include "../Math/Math.h"

namespace rt {

class BVH
{
public:
    static constexpr uint32 MaxDepth = 128;

    struct alignas(32) Node
    {
        math::Float3 min;
        uint32 childIndex;      // first child node / first leaf item
        math::Float3 max;
        uint32 numLeaves : 30;  // != 0 => leaf
        uint32 splitAxis : 2;

        bool IsLeaf() const { return numLeaves != 0; }
        uint32 GetSplitAxis() const { return splitAxis; }
        math::Box GetBox() const { return { math::Vector4(min), math::Vector4(max) }; }
    };
    static_assert(sizeof(Node) == 32, "BVH node must be 32 bytes");

    const Node* GetNodes() const { return mNodes.data(); }
    uint32 GetNumNodes() const { return mNumNodes; }

private:
    std::vector<Node> mNodes;
    uint32 mNumNodes = 0;
    friend class BVHBuilder;
};

} // namespace rt
