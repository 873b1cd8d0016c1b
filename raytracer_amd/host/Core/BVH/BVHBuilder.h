// Full-sweep SAH builder (host, one-time).  Same split rule, cost function, leaf size and leaf ordering
// as the reference's BVHBuilder (Core/BVH/BVHBuilder.cpp:37-306) so node order and leaf contents match
// (subObjectId == triangle index in leaf order).
#pragma once

#include "BVH.h"

namespace rt {

struct BvhBuildingParams
{
    enum class Heuristics { SurfaceArea, Volume };
    uint32 maxLeafNodeSize = 2;
    Heuristics heuristics = Heuristics::SurfaceArea;
};

class RAYLIB_API BVHBuilder
{
public:
    using Indices = std::vector<uint32>;

    explicit BVHBuilder(BVH& targetBVH);

    // construct the BVH and return the new order of the leaves
    bool Build(const math::Box* data, const uint32 numLeaves, const BvhBuildingParams& params, Indices& outLeavesOrder);

private:
    struct WorkSet
    {
        math::Box box;
        Indices leafIndices;
        uint32 sortedBy = 0xFFFFFFFFu;
        uint32 depth = 0;
    };
    struct Scratch
    {
        std::vector<math::Box> leftBoxes, rightBoxes;
        Indices sorted[3];
    };

    void BuildNode(const WorkSet& workSet, Scratch& scratch, uint32 targetNodeIndex);
    void SortLeaves(const WorkSet& workSet, Scratch& scratch) const;

    BvhBuildingParams mParams;
    const math::Box* mLeafBoxes = nullptr;
    uint32 mNumLeaves = 0;
    uint32 mNumGeneratedNodes = 0;
    uint32 mNumGeneratedLeaves = 0;
    Indices mLeavesOrder;
    BVH& mTarget;
};

} // namespace rt
