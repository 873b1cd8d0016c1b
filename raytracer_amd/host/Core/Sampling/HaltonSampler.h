// Scrambled multidimensional Halton sequence (host side, double precision), the source of the per-pass
// sampler seeds RtPassParams::seed.  Follows the reference's HaltonSequence
// (Core/Sampling/HaltonSampler.cpp:61-206): random start per base, random digit permutation per base,
// incremental digit update.
#pragma once

#include "../Math/Math.h"

namespace rt {

class RAYLIB_API HaltonSequence
{
public:
    static constexpr uint32 MaxDimensions = 4096;
    static constexpr uint32 Width = 64;

    HaltonSequence();
    ~HaltonSequence();
    void Initialize(uint32 dimensions);
    uint32 GetNumDimensions() const { return mDimensions; }
    void NextSample();
    double GetDouble(uint32 dimension) const { return mRnd[dimension][0]; }
    uint32 GetInt(uint32 dimension) const { return uint32(mRnd[dimension][0] * (double)UINT32_MAX); }

    math::Random& GetRandom() { return mRandom; }

private:
    uint64 Permute(uint32 i, uint32 j) const { return mPermutations[i][mDigit[i][j]]; }
    void UpdatePlace(uint32 dimension, int32 place);   // one partial sum of the scrambled radical inverse (HaltonSampler.cpp)
    void InitPrimes();
    void InitStart();
    void InitPowerBuffer();
    void InitExpansion();
    void InitPermutation();

    uint32 mDimensions = 0;
    std::vector<uint64> mStarts;
    std::vector<uint32> mBase;
    std::vector<std::vector<double>> mRnd;
    std::vector<std::vector<uint64>> mDigit;
    std::vector<std::vector<uint64>> mPowerBuffer;
    std::vector<std::vector<uint64>> mPermutations;
    math::Random mRandom;
};

} // namespace rt
