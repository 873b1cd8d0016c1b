// cadence.hip -- two microbenchmarks that decide which roof k_trace_wide sits under:
//   (1) VALU issue cadence on gfx950 for the instruction mix of the walk's inner loop (v_fma_f32, v_perm_b32, SDWA v_cvt_f32_u32,
//       v_max3 / v_min3, v_cndmask with an SGPR mask, v_min_u32 / v_max_u32, v_cmp): cycles per wave64 instruction per SIMD at 1 .. 8
//       resident waves per SIMD.  MI355X_MICROARCH.md says 2 cycles (SIMD-32); round 2's DESIGN assumed 4.
//   (2) vector L1 (TCP) rate for the walk's access pattern: every lane loads its own 64-byte node with four global_load_dwordx4
//       (64 distinct lines per instruction), from a table that fits L1 (16 KB per CU), L2 (2 MB) or neither (512 MB).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/cadence.hip -o tools/microbench/cadence ; run on the GPU box, prints a table.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Op { OP_FMA, OP_PERM, OP_CVT_SDWA, OP_CVT, OP_MAX3, OP_CNDMASK_SGPR, OP_MINU, OP_CMP, OP_PKFMA, OP_XOR, OP_MIX, OP_COUNT };
static const char* const kOpNames[OP_COUNT] = { "v_fma_f32", "v_perm_b32", "v_cvt_f32_u32_sdwa(WORD_1)", "v_cvt_f32_u32", "v_max3_f32", "v_cndmask_b32 (sgpr mask)",
                                                "v_min_u32", "v_cmp_gt_u32 (to sgpr pair)", "v_pk_fma_f32", "v_xor_b32", "the walk's mix (3 perm 6 cvt 6 fma max3 min3 max per child)" };

// 8 independent chains, 4 rounds = 32 instructions per iteration (one chain's instructions are 8 apart: no dependency stall)
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int kOp>
__global__ void __launch_bounds__(256) k_cadence(uint32_t iterations, float seed, float* out, unsigned long long* clocks)
{
    extern __shared__ uint32_t sPad[];   // dynamic LDS = 160 KB / (blocks per CU wanted): the dispatcher cannot pile more blocks on a CU
    if (seed == 123.0f) sPad[threadIdx.x] = 1u;
    float a[8]; uint32_t u[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + (float)(threadIdx.x + i); u[i] = threadIdx.x * 2654435761u + (uint32_t)i; }
    float b = seed * 0.5f, c = seed * 0.25f; uint32_t sel = 0x07060100u;
    float p2[2] = { b, c };
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (uint32_t it = 0; it < iterations; ++it)
    {
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
            if (kOp == OP_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                R8(X)
#undef X
            } else if (kOp == OP_PERM) {
#define X(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(sel));
                R8(X)
#undef X
            } else if (kOp == OP_CVT_SDWA) {
#define X(i) asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a[i]) : "v"(u[i]));
                R8(X)
#undef X
            } else if (kOp == OP_CVT) {
#define X(i) asm volatile("v_cvt_f32_u32_e32 %0, %1" : "=v"(a[i]) : "v"(u[i]));
                R8(X)
#undef X
            } else if (kOp == OP_MAX3) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                R8(X)
#undef X
            } else if (kOp == OP_CNDMASK_SGPR) {
                unsigned long long m = 0x5555555555555555ull;
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]) : "v"(sel), "s"(m));
                R8(X)
#undef X
            } else if (kOp == OP_MINU) {
#define X(i) asm volatile("v_min_u32_e32 %0, %0, %1" : "+v"(u[i]) : "v"(sel));
                R8(X)
#undef X
            } else if (kOp == OP_CMP) {
                unsigned long long m;
#define X(i) asm volatile("v_cmp_gt_u32_e64 %0, %1, %2" : "=s"(m) : "v"(u[i]), "v"(sel));
                R8(X)
#undef X
                asm volatile("" :: "s"(m));
            } else if (kOp == OP_PKFMA) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2* av = reinterpret_cast<f2*>(a); f2 bv = { p2[0], p2[1] };
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(av[i & 3]) : "v"(bv));
                R8(X)
#undef X
            } else if (kOp == OP_XOR) {
#define X(i) asm volatile("v_xor_b32_e32 %0, 0x7fffffff, %0" : "+v"(u[i]));
                R8(X)
#undef X
            } else if (kOp == OP_MIX) {
                // one child of the 4-wide step, twice (2 x 18 = 36 instructions): 3 perm, 6 sdwa cvt, 6 fma, max, max3, min3 -- two independent copies
#pragma unroll
                for (int h = 0; h < 2; ++h)
                {
                    uint32_t px, py, pz; float nx, ny, nz, xx, xy, xz;
                    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(px) : "v"(u[h * 4 + 1]), "v"(u[h * 4 + 0]), "v"(sel));
                    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(py) : "v"(u[h * 4 + 2]), "v"(u[h * 4 + 0]), "v"(sel));
                    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(pz) : "v"(u[h * 4 + 2]), "v"(u[h * 4 + 1]), "v"(sel));
                    asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(nx) : "v"(px));
                    asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(xx) : "v"(px));
                    asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(ny) : "v"(py));
                    asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(xy) : "v"(py));
                    asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(nz) : "v"(pz));
                    asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(xz) : "v"(pz));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(nx) : "v"(b), "v"(c));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(xx) : "v"(b), "v"(c));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ny) : "v"(b), "v"(c));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(xy) : "v"(b), "v"(c));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(nz) : "v"(b), "v"(c));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(xz) : "v"(b), "v"(c));
                    asm volatile("v_max_f32_e32 %0, 0, %0" : "+v"(nz));
                    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(a[h * 4 + 0]) : "v"(nx), "v"(ny), "v"(nz));
                    asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(a[h * 4 + 1]) : "v"(xx), "v"(xy), "v"(xz));
                }
            }
        }
    }
    const unsigned long long t1 = clock64();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + (float)u[i];
    if (s == 123.456f) out[0] = s;
    if ((threadIdx.x & 63u) == 0u) clocks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// every lane walks a random cycle of 64-byte nodes: four dwordx4 loads of ITS node per step (the walk's fetch), next = node.w of the first
__global__ void __launch_bounds__(256) k_l1(const float4* __restrict__ nodes, uint32_t numNodes, uint32_t steps, uint32_t activeLanes, float* out, unsigned long long* clocks)
{
    extern __shared__ uint32_t sPad[];
    if (steps == 0xFFFFFFFFu) sPad[threadIdx.x] = 1u;
    uint32_t cur = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u % numNodes;
    float acc = 0.0f;
    const bool active = (threadIdx.x & 63u) < activeLanes;
    __syncthreads();
    const unsigned long long t0 = clock64();
    if (active)
        for (uint32_t s = 0; s < steps; ++s)
        {
            const float4* p = nodes + 4u * (size_t)cur;
            const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
            acc += q0.x + q1.y + q2.z + q3.x;
            cur = __float_as_uint(q0.w);
        }
    const unsigned long long t1 = clock64();
    if (acc == 123.456f) out[0] = acc;
    if ((threadIdx.x & 63u) == 0u) clocks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// the same walk over nodes of `kLoads` x 16 bytes stored `stride16` x 16 bytes apart (48-byte nodes: dense = 3, one per 64-byte half line = 4)
template <int kLoads>
__global__ void __launch_bounds__(256) k_node(const float4* __restrict__ nodes, uint32_t numNodes, uint32_t stride16, uint32_t steps, uint32_t activeLanes, float* out, unsigned long long* clocks)
{
    extern __shared__ uint32_t sPad[];
    if (steps == 0xFFFFFFFFu) sPad[threadIdx.x] = 1u;
    uint32_t cur = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u % numNodes;
    float acc = 0.0f;
    const bool active = (threadIdx.x & 63u) < activeLanes;
    __syncthreads();
    const unsigned long long t0 = clock64();
    if (active)
        for (uint32_t s = 0; s < steps; ++s)
        {
            const float4* p = nodes + (size_t)stride16 * cur;
            float4 q[kLoads];
#pragma unroll
            for (int k = 0; k < kLoads; ++k) q[k] = p[k];
#pragma unroll
            for (int k = 0; k < kLoads; ++k) acc += q[k].x;
            cur = __float_as_uint(q[0].w);
        }
    const unsigned long long t1 = clock64();
    if (acc == 123.456f) out[0] = acc;
    if ((threadIdx.x & 63u) == 0u) clocks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// (4) step models for the next traversal kernel: every lane runs kRays independent random walks over nodes of kLoads x 16
// bytes; per step and ray it fetches its node and then issues kValu VALU instructions that DEPEND on the fetched words (fma chains seeded by the node),
// like the slab tests / sort network of k_trace_wide (137 per 4-wide visit).  Compared: today's step (4 loads, 137), an 8-wide node (8 loads = one
// 128-byte line, ~280: eight slab tests + a 19-exchange sort), two rays per lane (2 x (4 loads, 137) at the occupancy 120 VGPRs allow: 4 waves).
template <int kLoads, int kValu, int kRays>
__global__ void __launch_bounds__(256) k_step(const float4* __restrict__ nodes, uint32_t numNodes, uint32_t steps, uint32_t activeLanes, float* out, unsigned long long* clocks)
{
    extern __shared__ uint32_t sPad[];
    if (steps == 0xFFFFFFFFu) sPad[threadIdx.x] = 1u;
    uint32_t cur[kRays];
    for (int r = 0; r < kRays; ++r) cur[r] = ((blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + (uint32_t)r * 40503u) % numNodes;
    float acc[kRays][4];
    for (int r = 0; r < kRays; ++r) for (int k = 0; k < 4; ++k) acc[r][k] = 1.0f + (float)k;
    const bool active = (threadIdx.x & 63u) < activeLanes;
    __syncthreads();
    const unsigned long long t0 = clock64();
    if (active)
        for (uint32_t s = 0; s < steps; ++s)
        {
            float4 q[kRays][kLoads];
#pragma unroll
            for (int r = 0; r < kRays; ++r)
            {
                const float4* p = nodes + (size_t)kLoads * cur[r];
#pragma unroll
                for (int k = 0; k < kLoads; ++k) q[r][k] = p[k];
            }
#pragma unroll
            for (int r = 0; r < kRays; ++r)
            {
                // kValu dependent-on-the-node instructions in four independent chains (the compiler may not fold asm volatile)
                float b = q[r][kLoads - 1].x, c = q[r][kLoads / 2].y;
#pragma unroll
                for (int i = 0; i < kValu / 4; ++i)
                {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[r][0]) : "v"(b), "v"(c));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[r][1]) : "v"(b), "v"(c));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[r][2]) : "v"(b), "v"(c));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[r][3]) : "v"(b), "v"(c));
                }
                cur[r] = __float_as_uint(q[r][0].w);
            }
        }
    const unsigned long long t1 = clock64();
    float sum = 0; for (int r = 0; r < kRays; ++r) for (int k = 0; k < 4; ++k) sum += acc[r][k];
    if (sum == 123.456f) out[0] = sum;
    if ((threadIdx.x & 63u) == 0u) clocks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int kLoads, int kValu, int kRays>
static double runStep(uint32_t numCUs, const float4* dev, uint32_t numNodes, uint32_t wavesPerSimd, float* out, unsigned long long* clocksDev)
{
    const uint32_t steps = 3000u, activeLanes = 36u;
    const dim3 grid(numCUs * wavesPerSimd), block(256);
    const size_t lds = wavesPerSimd <= 1u ? (size_t)96 << 10 : ((size_t)160 << 10) / wavesPerSimd / 1024u * 1024u;
    CHECK(hipFuncSetAttribute((const void*)k_step<kLoads, kValu, kRays>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_step<kLoads, kValu, kRays>), grid, block, lds, 0, dev, numNodes, 200u, activeLanes, out, clocksDev);
    hipLaunchKernelGGL((k_step<kLoads, kValu, kRays>), grid, block, lds, 0, dev, numNodes, steps, activeLanes, out, clocksDev);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> clocks((size_t)grid.x * 4u);
    CHECK(hipMemcpy(clocks.data(), clocksDev, clocks.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double mean = 0; for (auto c : clocks) mean += (double)c; mean /= (double)clocks.size();
    return mean / steps;
}

// dynamic LDS that lets exactly `blocksPerCU` 256-thread blocks share a CU's 160 KB
static size_t ldsFor(uint32_t blocksPerCU) { return blocksPerCU <= 1u ? (size_t)96 << 10 : ((size_t)160 << 10) / blocksPerCU / 1024u * 1024u; }

template <int kOp>
static void runCadence(uint32_t numCUs, float* out, unsigned long long* clocksDev)
{
    const uint32_t iterations = 20000u, perIteration = kOp == OP_MIX ? 4u * 36u : 4u * 8u;   // 4 rounds of 8 (the mix: of 2 children x 18)
    printf("%-62s", kOpNames[kOp]);
    for (uint32_t wavesPerSimd : { 1u, 2u, 4u, 5u, 8u })
    {
        const dim3 grid(numCUs * wavesPerSimd), block(256);   // a 256-thread block = one wave per SIMD of its CU
        const size_t lds = ldsFor(wavesPerSimd);
        CHECK(hipFuncSetAttribute((const void*)k_cadence<kOp>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL((k_cadence<kOp>), grid, block, lds, 0, 100u, 1.0f, out, clocksDev);   // warm up
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_cadence<kOp>), grid, block, lds, 0, iterations, 1.0f, out, clocksDev);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> clocks((size_t)grid.x * 4u);
        CHECK(hipMemcpy(clocks.data(), clocksDev, clocks.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double mean = 0; for (auto c : clocks) mean += (double)c; mean /= (double)clocks.size();
        const double instr = (double)iterations * perIteration;
        // per SIMD: wavesPerSimd waves issued `instr` instructions each within `mean` clocks (if they all ran concurrently)
        // effective clock = ticks a wave ran / the kernel's wall time (every block is resident from start to end: one round of blocks)
        printf("  w%u: %5.2f clk @%4.2f GHz", wavesPerSimd, mean / (instr * wavesPerSimd), mean / (ms * 1e-3) * 1e-9);
    }
    printf("\n");
}

int main(int argc, char** argv)
{
    const bool onlyFormats = argc > 1 && argv[1][0] == '3';   // `cadence 3`: section (3) only
    const bool onlySteps = argc > 1 && argv[1][0] == '4';     // `cadence 4`: section (4) only
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const uint32_t numCUs = (uint32_t)prop.multiProcessorCount;
    printf("device %s, %u CUs, clock %d kHz (clock64 = s_memtime ticks)\n", prop.name, numCUs, prop.clockRate);
    float* out; unsigned long long* clocksDev;
    CHECK(hipMalloc((void**)&out, 64)); CHECK(hipMalloc((void**)&clocksDev, sizeof(unsigned long long) * numCUs * 8u * 4u * 2u));
    if (onlySteps)
    {
        printf("\n(4) step models: clocks per wave step; ray-steps per 1000 clocks and SIMD = waves x rays per lane / clocks; 4-wide-equivalent = x 17/11 for the 8-wide node\n");
        for (size_t tableBytes : { (size_t)4 << 20, (size_t)8 << 20, (size_t)22 << 20 })
        {
            // one table per node size: N random-cycle nodes of 64 bytes, or N / 2 of 128 bytes (the same bytes: an 8-wide collapse has half the nodes)
            auto makeTable = [&](uint32_t loads) -> std::pair<float4*, uint32_t>
            {
                const uint32_t numNodes = (uint32_t)(tableBytes / (16u * loads));
                std::vector<uint32_t> perm(numNodes);
                for (uint32_t i = 0; i < numNodes; ++i) perm[i] = i;
                uint64_t s = 88172645463325252ull;
                for (uint32_t i = numNodes - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; const uint32_t j = (uint32_t)(s % (i + 1)); std::swap(perm[i], perm[j]); }
                std::vector<float4> host((size_t)numNodes * loads, make_float4(1.0f, 0.5f, 3.0f, 0.0f));
                for (uint32_t i = 0; i < numNodes; ++i) host[(size_t)perm[i] * loads].w = __builtin_bit_cast(float, perm[(i + 1) % numNodes]);
                float4* dev; CHECK(hipMalloc((void**)&dev, host.size() * sizeof(float4)));
                CHECK(hipMemcpy(dev, host.data(), host.size() * sizeof(float4), hipMemcpyHostToDevice));
                return { dev, numNodes };
            };
            const auto t4 = makeTable(4u), t8 = makeTable(8u);
            const double a5 = runStep<4, 136, 1>(numCUs, t4.first, t4.second, 5u, out, clocksDev);
            const double a4 = runStep<4, 136, 1>(numCUs, t4.first, t4.second, 4u, out, clocksDev);
            const double b5 = runStep<8, 280, 1>(numCUs, t8.first, t8.second, 5u, out, clocksDev);
            const double b4 = runStep<8, 280, 1>(numCUs, t8.first, t8.second, 4u, out, clocksDev);
            const double c4 = runStep<4, 136, 2>(numCUs, t4.first, t4.second, 4u, out, clocksDev);
            const double c3 = runStep<4, 136, 2>(numCUs, t4.first, t4.second, 3u, out, clocksDev);
            const double f5 = runStep<4, 0, 1>(numCUs, t4.first, t4.second, 5u, out, clocksDev);
            printf("table %3zu MB:  today (4 loads, 136 VALU, 5 waves): %5.0f clk = %5.2f   at 4 waves: %5.0f clk = %5.2f   fetch only, 5 waves: %5.0f clk\n", tableBytes >> 20, a5, 5000.0 / a5, a4, 4000.0 / a4, f5);
            printf("               8-wide (8 loads, 280 VALU): 5 waves %5.0f clk = %5.2f (x17/11 = %5.2f)   4 waves %5.0f clk = %5.2f (x17/11 = %5.2f)\n", b5, 5000.0 / b5, 5000.0 / b5 * 17.0 / 11.0, b4, 4000.0 / b4, 4000.0 / b4 * 17.0 / 11.0);
            printf("               two rays per lane (2 x (4 loads, 136 VALU)): 4 waves %5.0f clk = %5.2f   3 waves %5.0f clk = %5.2f\n", c4, 8000.0 / c4, c3, 6000.0 / c3);
            CHECK(hipFree(t4.first)); CHECK(hipFree(t8.first));
        }
        return 0;
    }
    if (!onlyFormats) {
    printf("\n(1) cycles per wave64 VALU instruction per SIMD (clock64 ticks of one wave / instructions issued by all waves of its SIMD; wN = N resident waves per SIMD, enforced through the LDS allocation) and the effective clock\n");
    runCadence<OP_FMA>(numCUs, out, clocksDev); runCadence<OP_PERM>(numCUs, out, clocksDev); runCadence<OP_CVT_SDWA>(numCUs, out, clocksDev);
    runCadence<OP_CVT>(numCUs, out, clocksDev); runCadence<OP_MAX3>(numCUs, out, clocksDev); runCadence<OP_CNDMASK_SGPR>(numCUs, out, clocksDev);
    runCadence<OP_MINU>(numCUs, out, clocksDev); runCadence<OP_CMP>(numCUs, out, clocksDev); runCadence<OP_PKFMA>(numCUs, out, clocksDev);
    runCadence<OP_XOR>(numCUs, out, clocksDev); runCadence<OP_MIX>(numCUs, out, clocksDev);

    printf("\n(2) divergent node fetch: 4 x global_load_dwordx4 of one 64-byte node per lane and step; clocks per wave step and L1 accesses (active lanes x 4) per clock and CU\n");
    for (size_t tableBytes : { (size_t)8 << 10, (size_t)1 << 20, (size_t)8 << 20, (size_t)64 << 20, (size_t)1 << 30 })
    {
        const uint32_t numNodes = (uint32_t)(tableBytes / 64u);
        std::vector<float4> host((size_t)numNodes * 4u);
        std::vector<uint32_t> perm(numNodes);
        for (uint32_t i = 0; i < numNodes; ++i) perm[i] = i;
        uint64_t s = 88172645463325252ull;
        for (uint32_t i = numNodes - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; const uint32_t j = (uint32_t)(s % (i + 1)); std::swap(perm[i], perm[j]); }
        for (uint32_t i = 0; i < numNodes; ++i)
        {
            const uint32_t next = perm[(i + 1) % numNodes];   // one cycle through all nodes in random order
            for (int k = 0; k < 4; ++k) host[(size_t)perm[i] * 4u + k] = make_float4(1.0f, 2.0f, 3.0f, __builtin_bit_cast(float, next));
        }
        float4* dev; CHECK(hipMalloc((void**)&dev, host.size() * sizeof(float4)));
        CHECK(hipMemcpy(dev, host.data(), host.size() * sizeof(float4), hipMemcpyHostToDevice));
        for (uint32_t activeLanes : { 64u, 36u })
        {
            printf("table %7zu KB, %2u active lanes:", tableBytes >> 10, activeLanes);
            for (uint32_t wavesPerSimd : { 1u, 2u, 5u, 8u })
            {
                const uint32_t steps = tableBytes >= ((size_t)64 << 20) ? 2000u : 8000u;
                const dim3 grid(numCUs * wavesPerSimd), block(256);
                const size_t lds = ldsFor(wavesPerSimd);
                CHECK(hipFuncSetAttribute((const void*)k_l1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(k_l1, grid, block, lds, 0, dev, numNodes, 200u, activeLanes, out, clocksDev);
                hipLaunchKernelGGL(k_l1, grid, block, lds, 0, dev, numNodes, steps, activeLanes, out, clocksDev);
                CHECK(hipDeviceSynchronize());
                std::vector<unsigned long long> clocks((size_t)grid.x * 4u);
                CHECK(hipMemcpy(clocks.data(), clocksDev, clocks.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                double mean = 0; for (auto c : clocks) mean += (double)c; mean /= (double)clocks.size();
                const double perStep = mean / steps;
                printf("  w%u: %6.0f clk/step, %4.2f acc/clk/CU", wavesPerSimd, perStep, (double)activeLanes * 4.0 * 4.0 * wavesPerSimd / perStep);
            }
            printf("\n");
        }
        CHECK(hipFree(dev));
    }
    }
    printf("\n(3) node formats: the same random walk over N nodes of L x 16 bytes at a stride of S x 16 bytes, 5 waves per SIMD, 36 active lanes; clocks per wave step\n");
    for (uint32_t numNodes : { 65536u, 262144u, 1048576u })
    {
        std::vector<uint32_t> perm(numNodes);
        for (uint32_t i = 0; i < numNodes; ++i) perm[i] = i;
        uint64_t s = 88172645463325252ull;
        for (uint32_t i = numNodes - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; const uint32_t j = (uint32_t)(s % (i + 1)); std::swap(perm[i], perm[j]); }
        printf("%8u nodes:", numNodes);
        struct Format { int loads; uint32_t stride16; const char* name; };
        for (const Format f : { Format{ 4, 4u, "64 B / 4 loads" }, Format{ 3, 4u, "48 of 64 B / 3 loads" }, Format{ 3, 3u, "48 B dense / 3 loads" }, Format{ 2, 2u, "32 B / 2 loads" }, Format{ 2, 4u, "32 of 64 B / 2 loads" } })
        {
            std::vector<float4> host((size_t)numNodes * f.stride16, make_float4(1.0f, 2.0f, 3.0f, 0.0f));
            for (uint32_t i = 0; i < numNodes; ++i) host[(size_t)perm[i] * f.stride16].w = __builtin_bit_cast(float, perm[(i + 1) % numNodes]);
            float4* dev; CHECK(hipMalloc((void**)&dev, host.size() * sizeof(float4)));
            CHECK(hipMemcpy(dev, host.data(), host.size() * sizeof(float4), hipMemcpyHostToDevice));
            const uint32_t wavesPerSimd = 5u, steps = 4000u, activeLanes = 36u;
            const dim3 grid(numCUs * wavesPerSimd), block(256);
            const size_t lds = ldsFor(wavesPerSimd);
            auto launch = [&](uint32_t n)
            {
                if (f.loads == 4) { CHECK(hipFuncSetAttribute((const void*)k_node<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); hipLaunchKernelGGL(k_node<4>, grid, block, lds, 0, dev, numNodes, f.stride16, n, activeLanes, out, clocksDev); }
                else if (f.loads == 3) { CHECK(hipFuncSetAttribute((const void*)k_node<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); hipLaunchKernelGGL(k_node<3>, grid, block, lds, 0, dev, numNodes, f.stride16, n, activeLanes, out, clocksDev); }
                else { CHECK(hipFuncSetAttribute((const void*)k_node<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); hipLaunchKernelGGL(k_node<2>, grid, block, lds, 0, dev, numNodes, f.stride16, n, activeLanes, out, clocksDev); }
            };
            launch(200u); launch(steps);
            CHECK(hipDeviceSynchronize());
            std::vector<unsigned long long> clocks((size_t)grid.x * 4u);
            CHECK(hipMemcpy(clocks.data(), clocksDev, clocks.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            double mean = 0; for (auto c : clocks) mean += (double)c; mean /= (double)clocks.size();
            printf("  %s: %5.0f", f.name, mean / steps);
            CHECK(hipFree(dev));
        }
        printf("\n");
    }
    return 0;
}
