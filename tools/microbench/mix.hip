// mix.hip -- does v_fma_mix_f32 (an fp32 fma whose first source is an fp16 half of a register, converted on the fly) issue like v_fma_f32?
// One child of k_trace_wide's step in two forms at 5 waves per SIMD: (A) uint16 planes: 3 perm + 6 SDWA cvt + 6 fma + max + max3 + min3 = 18 instructions,
// (B) fp16 planes: 3 perm + 6 v_fma_mix_f32 + max + max3 + min3 = 12.  Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/mix.hip -o /tmp/mix ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int kForm>
__global__ void __launch_bounds__(256) k_mix(uint32_t iterations, float seed, float* out, unsigned long long* clocks)
{
    extern __shared__ uint32_t sPad[];
    if (seed == 123.0f) sPad[threadIdx.x] = 1u;
    float a[8]; uint32_t u[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + (float)(threadIdx.x + i); u[i] = 0x3C003C00u + ((threadIdx.x * 2654435761u + (uint32_t)i) & 0x03FF03FFu); }
    float b = seed * 0.5f, c = seed * 0.25f; uint32_t sel = 0x07060100u;
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (uint32_t it = 0; it < iterations; ++it)
    {
#pragma unroll
        for (int h = 0; h < 2; ++h)
        {
            uint32_t px, py, pz; float nx, ny, nz, xx, xy, xz;
            asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(px) : "v"(u[h * 4 + 1]), "v"(u[h * 4 + 0]), "v"(sel));
            asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(py) : "v"(u[h * 4 + 2]), "v"(u[h * 4 + 0]), "v"(sel));
            asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(pz) : "v"(u[h * 4 + 2]), "v"(u[h * 4 + 1]), "v"(sel));
            if (kForm == 0)
            {
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
            }
            else
            {
                asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(nx) : "v"(px), "v"(b), "v"(c));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(xx) : "v"(px), "v"(b), "v"(c));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(ny) : "v"(py), "v"(b), "v"(c));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(xy) : "v"(py), "v"(b), "v"(c));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(nz) : "v"(pz), "v"(b), "v"(c));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(xz) : "v"(pz), "v"(b), "v"(c));
            }
            asm volatile("v_max_f32_e32 %0, 0, %0" : "+v"(nz));
            asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(a[h * 4 + 0]) : "v"(nx), "v"(ny), "v"(nz));
            asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(a[h * 4 + 1]) : "v"(xx), "v"(xy), "v"(xz));
        }
    }
    const unsigned long long t1 = clock64();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + (float)u[i];
    if (s == 123.456f) out[0] = s;
    if ((threadIdx.x & 63u) == 0u) clocks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* out; unsigned long long* clocks; CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&clocks, sizeof(unsigned long long) * cus * 8 * 4));
    const uint32_t iterations = 20000;
    for (int wavesPerSimd : { 1, 2, 4, 5 })
        for (int form = 0; form < 2; ++form)
        {
            const int blocks = cus * wavesPerSimd; const size_t lds = 160 * 1024 / wavesPerSimd - 1024;
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; ++rep)
            {
                CHECK(hipEventRecord(e0));
                if (form == 0) hipLaunchKernelGGL(k_mix<0>, dim3(blocks), dim3(256), lds, 0, iterations, 1.5f, out, clocks);
                else hipLaunchKernelGGL(k_mix<1>, dim3(blocks), dim3(256), lds, 0, iterations, 1.5f, out, clocks);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            }
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h = 0; CHECK(hipMemcpy(&h, clocks, 8, hipMemcpyDeviceToHost));
            printf("waves/SIMD %d  %s: %8.3f ms, %6.1f wave clocks per child, %6.2f ns per child and SIMD-resident set\n", wavesPerSimd,
                   form == 0 ? "(A) uint16: 3 perm 6 cvt 6 fma 3 minmax = 18" : "(B) fp16:   3 perm 6 fma_mix    3 minmax = 12", ms, (double)h / (iterations * 2.0), 1e6 * ms / (iterations * 2.0));
        }
    return 0;
}
