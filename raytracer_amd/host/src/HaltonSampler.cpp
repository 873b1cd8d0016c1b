// Scrambled Halton sequence, host side (double precision).  See Core/Sampling/HaltonSampler.h.
// The arithmetic (expression order, double rounding) follows the reference's
// Core/Sampling/HaltonSampler.cpp:61-206 so that seed[d] is reproduced bit-exactly from the same
// generator state.
#include "../Core/Sampling/HaltonSampler.h"

#include <algorithm>

namespace rt {

HaltonSequence::HaltonSequence() = default;
HaltonSequence::~HaltonSequence() = default;

void HaltonSequence::Initialize(uint32 dimensions)
{
    mDimensions = dimensions > MaxDimensions ? MaxDimensions : dimensions;
    mRnd.assign(mDimensions, std::vector<double>(Width + 1, 0.0));   // +1: rnd[j + 1] is read at j = Width - 1
    mDigit.assign(mDimensions, std::vector<uint64>(Width, 0));
    mPowerBuffer.assign(mDimensions, std::vector<uint64>(Width, 0));
    mStarts.assign(mDimensions, 0);
    mBase.assign(mDimensions, 0);
    mPermutations.clear();

    if (mDimensions > 0)
    {
        InitPrimes();
        InitStart();
        InitPowerBuffer();
        InitPermutation();
        InitExpansion();
    }
}

// first mDimensions primes by trial division (HaltonSampler.cpp:138-158)
void HaltonSequence::InitPrimes()
{
    uint32 found = 0;
    for (uint32 candidate = 2; found < mDimensions; ++candidate)
    {
        bool isPrime = true;
        for (uint64 i = 2; i <= sqrt(candidate); i++)
        {
            if (candidate % i == 0) { isPrime = false; break; }
        }
        if (isPrime) mBase[found++] = candidate;
    }
}

// Start index of a dimension = a uniform double written out digit by digit in the dimension's base (HaltonSampler.cpp:160-183): the digit of
// weight base^-(k+1) is peeled off the remainder while that is still above 1e-16 and enters the index with weight base^k.  mStarts must be the
// reference's integers, so the three double expressions keep the reference's operations in its order: the quotient 1.0 / place, the product
// remainder * place under floor, and digit * 1.0 / place subtracted from the remainder.
static uint64 RadicalDigitsOf(double remainder, const uint64 base)
{
    uint64 index = 0;
    for (uint64 place = base; remainder > 1.0e-16; place *= base)
    {
        const double placeAsDouble = (double)place;
        if (remainder < 1.0 / placeAsDouble) continue;   // digit 0 at this place
        const uint64 digit = (uint64)floor(remainder * placeAsDouble);
        remainder -= (double)digit * 1.0 / placeAsDouble;
        index += digit * place / base;
    }
    return index;
}

void HaltonSequence::InitStart()
{
    for (uint32 d = 0; d < mDimensions; d++) mStarts[d] = RadicalDigitsOf(mRandom.GetDouble(), mBase[d]);
}

void HaltonSequence::InitPowerBuffer()   // :31-59
{
    for (uint32 d = 0; d < mDimensions; d++)
    {
        for (uint32 j = 0; j < Width; j++)
        {
            mPowerBuffer[d][j] = (j == 0) ? (uint64)mBase[d] : mPowerBuffer[d][j - 1] * mBase[d];
        }
    }
    for (auto& v : mRnd) std::fill(v.begin(), v.end(), 0.0);
    for (auto& v : mDigit) std::fill(v.begin(), v.end(), 0);
}

// random digit permutation per base, digit 0 may move too (:113-136)
void HaltonSequence::InitPermutation()
{
    mPermutations.resize(mDimensions);
    for (uint32 i = 0; i < mDimensions; i++)
    {
        std::vector<uint64>& perm = mPermutations[i];
        perm.resize(mBase[i]);
        for (uint64 j = 0; j < mBase[i]; j++) perm[j] = j;
        for (uint64 j = 1; j < mBase[i]; j++)
        {
            const uint64 tmp = (uint64)floor(mRandom.GetDouble() * mBase[i]);
            if (tmp != 0)
            {
                const uint64 k = perm[j];
                perm[j] = perm[tmp];
                perm[tmp] = k;
            }
        }
    }
}

// The scrambled radical inverse of a dimension is kept as partial sums: mRnd[d][j] = the contribution of digit places j, j + 1, ... -- so that a step of the
// sequence touches only the places whose digits changed.  One place: the permuted digit over base^(j + 1), added to the sum of the higher places.  The double
// expression is the reference's (HaltonSampler.cpp:78, :96, :103: `d * 1.0 / power`, then added to the right-hand neighbour): the seeds are these doubles.
void HaltonSequence::UpdatePlace(uint32 dimension, int32 place)
{
    const uint64 scrambled = Permute(dimension, (uint32)place);
    mRnd[dimension][place] = mRnd[dimension][place + 1] + scrambled * 1.0 / mPowerBuffer[dimension][place];
}

// the digit string of (start - 1), least significant place first, then every partial sum from the most significant place down (:61-82)
void HaltonSequence::InitExpansion()
{
    for (uint32 dimension = 0; dimension < mDimensions; dimension++)
    {
        const uint64 base = mBase[dimension];
        int32 places = 0;
        for (uint64 rest = mStarts[dimension] - 1; rest > 0; rest /= base) mDigit[dimension][places++] = rest % base;
        for (int32 place = places - 1; place >= 0; --place) UpdatePlace(dimension, place);
    }
}

// the digit string plus one: the places below the first one that does not overflow wrap to zero; the partial sums of the places that changed are rebuilt from
// the highest of them down (:84-106)
void HaltonSequence::NextSample()
{
    for (uint32 dimension = 0; dimension < mDimensions; dimension++)
    {
        std::vector<uint64>& digits = mDigit[dimension];
        int32 carry = 0;
        while (digits[carry] + 1 >= mBase[dimension]) ++carry;
        digits[carry]++;
        std::fill(digits.begin(), digits.begin() + carry, (uint64)0);
        for (int32 place = carry; place >= 0; --place) UpdatePlace(dimension, place);
    }
}

} // namespace rt
