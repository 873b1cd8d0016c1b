// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of this library's kernels (round-4 review, item 2).
// The MI355X guide calibrates FETCH_SIZE only for wide coalesced 16-byte-per-lane streams (it reports exactly half of their bytes there); the traversal
// kernels fetch one random 64-byte node per lane (four 16-byte loads of one half line) and random 72-byte triangle pairs.  Each kernel below moves a KNOWN
// number of bytes in one of these patterns over a table far larger than the 256 MB Infinity Cache, so that every distinct line comes from memory:
//   k_calib_stream16   every lane one 16-byte load, consecutive (the guide's pattern: expected factor 2)
//   k_calib_node64     every lane four 16-byte loads of ONE random 64-byte node (k_trace_wide's interior step)
//   k_calib_node16     every lane one 16-byte load of a random 64-byte node
//   k_calib_pair72     every lane 72 bytes at a random 36-byte stride (a leaf's two triangles: dwordx4, dwordx4, dword, x 2)
//   k_calib_store16    every lane one 16-byte store, consecutive (WRITE_SIZE: the path-record writes)
//   k_calib_store16r   every lane one 16-byte store to a random 16-byte slot (WRITE_SIZE: hit records written through)
// The program prints, per kernel, the bytes its lanes REQUESTED and the bytes of the distinct 64-byte and 128-byte lines they touched (computed on the host
// from the same index stream): the counter divided into these gives the factor and says which line size the fabric requests have.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/fetch_calib.hip -o tools/microbench/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o r -- tools/microbench/fetch_calib   (and again with WRITE_SIZE; tools/fetch_calib.sh does both)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <unordered_set>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hashIndex(uint32_t i, uint32_t salt)
{
    uint32_t x = i * 0x9E3779B1u + salt; x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x;
}
static uint32_t hashIndexHost(uint32_t i, uint32_t salt)
{
    uint32_t x = i * 0x9E3779B1u + salt; x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x;
}

__global__ void k_calib_stream16(const float4* __restrict__ table, float* __restrict__ sink, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 v = table[i];
    if (v.x == 12345.678f) sink[0] = v.y + v.z + v.w;
}
__global__ void k_calib_node64(const float4* __restrict__ table, float* __restrict__ sink, uint32_t n, uint32_t numNodes)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4* p = table + 4u * (size_t)(hashIndex(i, 1u) % numNodes);
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    if (a.x + b.x + c.x + d.x == 12345.678f) sink[0] = a.y;
}
__global__ void k_calib_node16(const float4* __restrict__ table, float* __restrict__ sink, uint32_t n, uint32_t numNodes)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a = table[4u * (size_t)(hashIndex(i, 2u) % numNodes)];
    if (a.x == 12345.678f) sink[0] = a.y;
}
__global__ void k_calib_pair72(const float* __restrict__ table, float* __restrict__ sink, uint32_t n, uint32_t numTriangles)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = table + 9u * (size_t)(hashIndex(i, 3u) % (numTriangles - 1u));
    float s = 0.0f;
    for (int k = 0; k < 18; ++k) s += p[k];
    if (s == 12345.678f) sink[0] = s;
}
__global__ void k_calib_store16(float4* __restrict__ table, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    table[i] = make_float4((float)i, 1.0f, 2.0f, 3.0f);
}
__global__ void k_calib_store16r(float4* __restrict__ table, uint32_t n, uint32_t numSlots)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    table[hashIndex(i, 4u) % numSlots] = make_float4((float)i, 1.0f, 2.0f, 3.0f);
}

int main(int argc, char** argv)
{
    const size_t tableBytes = (argc > 1 ? (size_t)atol(argv[1]) : 2048) << 20;     // MB; default 2 GB: eight times the Infinity Cache
    const uint32_t lanes = argc > 2 ? (uint32_t)atol(argv[2]) : (8u << 20);       // lanes per random-access kernel
    float4* table = nullptr; float* sink = nullptr;
    CHECK(hipMalloc((void**)&table, tableBytes));
    CHECK(hipMalloc((void**)&sink, 64));
    CHECK(hipMemset(table, 0, tableBytes));
    CHECK(hipDeviceSynchronize());
    const uint32_t numNodes = (uint32_t)(tableBytes / 64u), numTriangles = (uint32_t)(tableBytes / 36u), numSlots = (uint32_t)(tableBytes / 16u);
    const uint32_t streamLanes = (uint32_t)(tableBytes / 16u);
    const dim3 block(256);
    auto grid = [&](uint32_t n) { return dim3((n + 255u) / 256u); };
    // what the index streams touch (host replay of the same hashes)
    auto distinct = [&](auto&& address, uint32_t bytes, uint32_t line) -> double
    {
        std::unordered_set<uint64_t> seen; seen.reserve(lanes * 2u);
        for (uint32_t i = 0; i < lanes; ++i) { const uint64_t a = address(i); for (uint64_t l = a / line; l <= (a + bytes - 1u) / line; ++l) seen.insert(l); }
        return (double)seen.size() * line;
    };
    auto nodeAddr = [&](uint32_t salt) { return [=](uint32_t i) { return (uint64_t)64u * (hashIndexHost(i, salt) % numNodes); }; };
    auto pairAddr = [&](uint32_t i) { return (uint64_t)36u * (hashIndexHost(i, 3u) % (numTriangles - 1u)); };
    auto slotAddr = [&](uint32_t i) { return (uint64_t)16u * (hashIndexHost(i, 4u) % numSlots); };
    printf("{\"table_bytes\": %zu, \"lanes\": %u, \"kernels\": {\n", tableBytes, lanes);
    for (int rep = 0; rep < 2; ++rep)   // two launches each: the profiler's per-kernel sums are divided by the call count
    {
        hipLaunchKernelGGL(k_calib_stream16, grid(streamLanes), block, 0, 0, table, sink, streamLanes);
        hipLaunchKernelGGL(k_calib_node64, grid(lanes), block, 0, 0, table, sink, lanes, numNodes);
        hipLaunchKernelGGL(k_calib_node16, grid(lanes), block, 0, 0, table, sink, lanes, numNodes);
        hipLaunchKernelGGL(k_calib_pair72, grid(lanes), block, 0, 0, reinterpret_cast<const float*>(table), sink, lanes, numTriangles);
        hipLaunchKernelGGL(k_calib_store16, grid(streamLanes), block, 0, 0, table, streamLanes);
        hipLaunchKernelGGL(k_calib_store16r, grid(lanes), block, 0, 0, table, lanes, numSlots);
        CHECK(hipDeviceSynchronize());
    }
    printf(" \"k_calib_stream16\": {\"requested\": %.0f, \"lines64\": %.0f, \"lines128\": %.0f},\n", (double)streamLanes * 16.0, (double)tableBytes, (double)tableBytes);
    printf(" \"k_calib_node64\": {\"requested\": %.0f, \"lines64\": %.0f, \"lines128\": %.0f},\n", (double)lanes * 64.0, distinct(nodeAddr(1u), 64u, 64u), distinct(nodeAddr(1u), 64u, 128u));
    printf(" \"k_calib_node16\": {\"requested\": %.0f, \"lines64\": %.0f, \"lines128\": %.0f},\n", (double)lanes * 16.0, distinct(nodeAddr(2u), 16u, 64u), distinct(nodeAddr(2u), 16u, 128u));
    printf(" \"k_calib_pair72\": {\"requested\": %.0f, \"lines64\": %.0f, \"lines128\": %.0f},\n", (double)lanes * 72.0, distinct(pairAddr, 72u, 64u), distinct(pairAddr, 72u, 128u));
    printf(" \"k_calib_store16\": {\"requested\": %.0f, \"lines64\": %.0f, \"lines128\": %.0f},\n", (double)streamLanes * 16.0, (double)tableBytes, (double)tableBytes);
    printf(" \"k_calib_store16r\": {\"requested\": %.0f, \"lines64\": %.0f, \"lines128\": %.0f}\n}}\n", (double)lanes * 16.0, distinct(slotAddr, 16u, 64u), distinct(slotAddr, 16u, 128u));
    CHECK(hipFree(table)); CHECK(hipFree(sink));
    return 0;
}
