// Full-sweep SAH BVH builder, host side.  See Core/BVH/BVHBuilder.h.
#include "../Core/BVH/BVHBuilder.h"

#include <algorithm>

namespace rt {

using namespace math;

BVHBuilder::BVHBuilder(BVH& targetBVH) : mTarget(targetBVH) {}

bool BVHBuilder::Build(const Box* data, const uint32 numLeaves, const BvhBuildingParams& params, Indices& outLeavesOrder)
{
    mLeafBoxes = data;
    mNumLeaves = numLeaves;
    mParams = params;
    mNumGeneratedNodes = 0;
    mNumGeneratedLeaves = 0;
    mLeavesOrder.clear();
    mLeavesOrder.reserve(numLeaves);

    mTarget.mNodes.assign((size_t)2 * numLeaves, BVH::Node{});
    mTarget.mNumNodes = 0;
    outLeavesOrder.clear();
    if (numLeaves == 0) return true;

    WorkSet root;
    root.box = Box::Empty();
    root.leafIndices.resize(numLeaves);
    for (uint32 i = 0; i < numLeaves; ++i)
    {
        root.box = Box(root.box, mLeafBoxes[i]);
        root.leafIndices[i] = i;
    }

    Scratch scratch;
    scratch.leftBoxes.resize(numLeaves);
    scratch.rightBoxes.resize(numLeaves);

    // node 0 is the root; node 1 stays unused (children are always allocated in adjacent pairs)
    mNumGeneratedNodes = 2;
    BuildNode(root, scratch, 0);

    mTarget.mNumNodes = mNumGeneratedNodes;
    mTarget.mNodes.resize(mNumGeneratedNodes);
    outLeavesOrder = mLeavesOrder;
    return mNumGeneratedLeaves == numLeaves;
}

void BVHBuilder::SortLeaves(const WorkSet& workSet, Scratch& scratch) const
{
    for (uint32 axis = 0; axis < 3; ++axis)
    {
        if (workSet.sortedBy == axis) continue;   // the parent already sorted along this axis
        Indices& indices = scratch.sorted[axis];
        indices = workSet.leafIndices;
        const Box* boxes = mLeafBoxes;
        std::sort(indices.begin(), indices.end(), [boxes, axis](const uint32 a, const uint32 b) {
            const Vector4 centerA = boxes[a].max + boxes[a].min;
            const Vector4 centerB = boxes[b].max + boxes[b].min;
            return centerA[axis] < centerB[axis];
        });
    }
    if (workSet.sortedBy < 3) scratch.sorted[workSet.sortedBy] = workSet.leafIndices;
}

void BVHBuilder::BuildNode(const WorkSet& workSet, Scratch& scratch, uint32 targetNodeIndex)
{
    const uint32 numLeaves = (uint32)workSet.leafIndices.size();
    {
        BVH::Node& node = mTarget.mNodes[targetNodeIndex];
        node.min = workSet.box.min.ToFloat3();
        node.max = workSet.box.max.ToFloat3();
    }

    if (numLeaves <= mParams.maxLeafNodeSize)
    {
        BVH::Node& node = mTarget.mNodes[targetNodeIndex];
        node.numLeaves = numLeaves;
        node.splitAxis = 0;
        node.childIndex = mNumGeneratedLeaves;
        for (uint32 i = 0; i < numLeaves; ++i) mLeavesOrder.push_back(workSet.leafIndices[i]);
        mNumGeneratedLeaves += numLeaves;
        return;
    }

    uint32 bestAxis = 0, bestSplitPos = 0;
    float bestCost = FLT_MAX;
    Box bestLeftBox = Box::Empty(), bestRightBox = Box::Empty();

    SortLeaves(workSet, scratch);

    for (uint32 axis = 0; axis < 3; ++axis)
    {
        const Indices& sorted = scratch.sorted[axis];
        {
            Box acc = Box::Empty();
            for (uint32 i = 0; i < numLeaves; ++i) { acc = Box(acc, mLeafBoxes[sorted[i]]); scratch.leftBoxes[i] = acc; }
        }
        {
            Box acc = Box::Empty();
            for (uint32 i = numLeaves; i-- > 0;) { acc = Box(acc, mLeafBoxes[sorted[i]]); scratch.rightBoxes[i] = acc; }
        }
        for (uint32 splitPos = 0; splitPos < numLeaves - 1; ++splitPos)
        {
            const Box& leftBox = scratch.leftBoxes[splitPos];
            const Box& rightBox = scratch.rightBoxes[splitPos + 1];
            const bool area = mParams.heuristics == BvhBuildingParams::Heuristics::SurfaceArea;
            const float leftCost = area ? leftBox.SurfaceArea() : leftBox.Volume();
            const float rightCost = area ? rightBox.SurfaceArea() : rightBox.Volume();
            const uint32 leftCount = splitPos + 1;
            const uint32 rightCount = numLeaves - leftCount;
            const float totalCost = leftCost * static_cast<float>(leftCount) + rightCost * static_cast<float>(rightCount);
            if (totalCost < bestCost)
            {
                bestCost = totalCost; bestAxis = axis; bestSplitPos = splitPos;
                bestLeftBox = leftBox; bestRightBox = rightBox;
            }
        }
    }

    const uint32 leftCount = bestSplitPos + 1;
    const uint32 leftNodeIndex = mNumGeneratedNodes;
    mNumGeneratedNodes += 2;
    {
        BVH::Node& node = mTarget.mNodes[targetNodeIndex];
        node.childIndex = leftNodeIndex;
        node.numLeaves = 0;
        node.splitAxis = bestAxis;
    }

    WorkSet left, right;
    left.sortedBy = right.sortedBy = bestAxis;
    left.depth = right.depth = workSet.depth + 1;
    const Indices& sorted = scratch.sorted[bestAxis];
    left.leafIndices.assign(sorted.begin(), sorted.begin() + leftCount);
    right.leafIndices.assign(sorted.begin() + leftCount, sorted.begin() + numLeaves);
    left.box = bestLeftBox;
    right.box = bestRightBox;

    BuildNode(left, scratch, leftNodeIndex);
    BuildNode(right, scratch, leftNodeIndex + 1);
}

} // namespace rt
