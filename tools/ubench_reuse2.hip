// Memory side of a one-workgroup-per-polynomial NTT as a function of how many polynomials are in flight on the chip.
// Follow-up of tools/ubench_reuse.hip (no arithmetic, access patterns only): the workgroup count per CU is limited by
// a dynamic LDS request, so that the in-flight footprint (workgroups x 512 KB) can be put below the 256 MB Infinity
// Cache; a third variant keeps `keep` of the 16 column tiles' worth of rows in LDS between the passes.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_reuse2.hip -o tools/ubench_reuse2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
constexpr u32 POLY = 65536;

__device__ __forceinline__ void cols_tile(u64 *poly, u32 tile, u32 tid)
{
    const u32 c = tid & 15, tau = tid >> 4;
    u64 *p = poly + tile * 16;
    u64 x[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = p[(size_t)(e * 16 + tau) * 256 + c];
#pragma unroll
    for (int e = 0; e < 16; ++e) p[(size_t)(e * 16 + tau) * 256 + c] = x[e] + 1;
}
__device__ __forceinline__ void rows_tile(u64 *poly, u32 tile, u32 tid)
{
    u64 *p = poly + (size_t)tile * 4096;
    u64 x[16];
#pragma unroll
    for (int h = 0; h < 16; ++h) x[h] = p[h * 256 + tid];
#pragma unroll
    for (int h = 0; h < 16; ++h) p[h * 256 + tid] = x[h] + 1;
}
__global__ __launch_bounds__(256) void pass_cols(u64 *buf) { cols_tile(buf + (size_t)(blockIdx.x >> 4) * POLY, blockIdx.x & 15, threadIdx.x); }
__global__ __launch_bounds__(256) void pass_rows(u64 *buf) { rows_tile(buf + (size_t)(blockIdx.x >> 4) * POLY, blockIdx.x & 15, threadIdx.x); }

template <int T>
__global__ __launch_bounds__(T) void both_passes(u64 *buf, u32 polys)
{
    extern __shared__ u64 dyn[];
    if (threadIdx.x == 0xffffffffu) dyn[0] = 1; // keep the allocation
    const u32 sub = threadIdx.x >> 8, tid = threadIdx.x & 255;
    for (u32 pi = blockIdx.x; pi < polys; pi += gridDim.x)
    {
        u64 *poly = buf + (size_t)pi * POLY;
        for (u32 tile = sub; tile < 16; tile += T / 256) cols_tile(poly, tile, tid);
        __syncthreads();
        for (u32 tile = sub; tile < 16; tile += T / 256) rows_tile(poly, tile, tid);
        __syncthreads();
    }
}

// P polynomials per workgroup: all column passes, then the row passes row tile by row tile across the P polynomials
// (what sharing one staged twiddle tile between polynomials needs)
template <int T, int P>
__global__ __launch_bounds__(T) void both_passes_multi(u64 *buf, u32 polys)
{
    extern __shared__ u64 dyn[];
    if (threadIdx.x == 0xffffffffu) dyn[0] = 1;
    const u32 sub = threadIdx.x >> 8, tid = threadIdx.x & 255;
    for (u32 pi = blockIdx.x * P; pi < polys; pi += gridDim.x * P)
    {
        u64 *base = buf + (size_t)pi * POLY;
        for (u32 u = sub; u < 16 * P; u += T / 256) cols_tile(base + (size_t)(u >> 4) * POLY, u & 15, tid);
        __syncthreads();
        for (u32 u = sub; u < 16 * P; u += T / 256) rows_tile(base + (size_t)(u % P) * POLY, u / P, tid);
        __syncthreads();
    }
}

// column tiles write rows [0, KEEP*16) of the polynomial into LDS instead of memory (KEEP row tiles = KEEP*32 KB);
// the row pass takes those row tiles from LDS.  1024 threads.
template <int KEEP>
__global__ __launch_bounds__(1024) void both_passes_lds(u64 *buf, u32 polys)
{
    extern __shared__ u64 lds[]; // KEEP * 4096 words
    const u32 sub = threadIdx.x >> 8, tid = threadIdx.x & 255;
    for (u32 pi = blockIdx.x; pi < polys; pi += gridDim.x)
    {
        u64 *poly = buf + (size_t)pi * POLY;
        for (u32 tile = sub; tile < 16; tile += 4)
        {
            const u32 c = tid & 15, tau = tid >> 4;
            u64 *p = poly + tile * 16;
            u64 x[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) x[e] = p[(size_t)(e * 16 + tau) * 256 + c];
#pragma unroll
            for (int e = 0; e < 16; ++e)
            {
                const u32 row = e * 16 + tau; // e < KEEP <=> row tile (row / 16) == e < KEEP
                if (e < KEEP) lds[row * 256 + tile * 16 + c] = x[e] + 1;
                else p[(size_t)row * 256 + c] = x[e] + 1;
            }
        }
        __syncthreads();
        for (u32 tile = sub; tile < 16; tile += 4)
        {
            u64 *p = poly + (size_t)tile * 4096;
            u64 x[16];
            if (tile < KEEP)
            {
#pragma unroll
                for (int h = 0; h < 16; ++h) x[h] = lds[tile * 4096 + h * 256 + tid];
            }
            else
            {
#pragma unroll
                for (int h = 0; h < 16; ++h) x[h] = p[h * 256 + tid];
            }
#pragma unroll
            for (int h = 0; h < 16; ++h) p[h * 256 + tid] = x[h] + 1;
        }
        __syncthreads();
    }
}

int main(int argc, char **argv)
{
    const size_t polys = argc > 1 ? (size_t)atol(argv[1]) : 16384;
    const size_t bytes = polys * POLY * sizeof(u64);
    u64 *buf = nullptr;
    CK(hipMalloc(&buf, bytes));
    CK(hipMemset(buf, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int launches = 0;
    auto timed = [&](const char *name, auto launch) {
        launch();
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 4; ++r)
        {
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        CK(hipGetLastError());
        launches += 5;
        printf("%-58s %8.3f ms   %7.1f GB/s algorithmic\n", name, best, 2.0 * bytes / best / 1e6);
        fflush(stdout);
    };
    const u32 tiles = (u32)(polys * 16);
    timed("two launches (cols, rows)", [&] {
        hipLaunchKernelGGL(pass_cols, dim3(tiles), dim3(256), 0, 0, buf);
        hipLaunchKernelGGL(pass_rows, dim3(tiles), dim3(256), 0, 0, buf);
    });
    CK(hipFuncSetAttribute((const void *)both_passes<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)both_passes<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)both_passes<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    char name[128];
    // grid = all polys (hardware scheduler refills), LDS request limits workgroups per CU
    for (int per_cu : {1, 2, 3, 4, 8})
    {
        const u32 lds = per_cu == 8 ? 0 : (160 * 1024 / per_cu) & ~255u;
        snprintf(name, sizeof name, "1 WG/poly, 256 thr, <= %d WG/CU (grid = polys)", per_cu);
        timed(name, [&] { hipLaunchKernelGGL(both_passes<256>, dim3((u32)polys), dim3(256), lds, 0, buf, (u32)polys); });
    }
    for (int per_cu : {1, 2})
    {
        const u32 lds = (160 * 1024 / per_cu) & ~255u;
        snprintf(name, sizeof name, "1 WG/poly, 512 thr, <= %d WG/CU", per_cu);
        timed(name, [&] { hipLaunchKernelGGL(both_passes<512>, dim3((u32)polys), dim3(512), lds, 0, buf, (u32)polys); });
        snprintf(name, sizeof name, "1 WG/poly, 1024 thr, <= %d WG/CU", per_cu);
        timed(name, [&] { hipLaunchKernelGGL(both_passes<1024>, dim3((u32)polys), dim3(1024), lds, 0, buf, (u32)polys); });
    }
    // persistent form: grid = 256 x per_cu workgroups looping over polynomials
    for (int per_cu : {1, 2})
    {
        const u32 lds = (160 * 1024 / per_cu) & ~255u;
        snprintf(name, sizeof name, "persistent, 1024 thr, grid %d", 256 * per_cu);
        timed(name, [&] { hipLaunchKernelGGL(both_passes<1024>, dim3(256 * per_cu), dim3(1024), lds, 0, buf, (u32)polys); });
    }
    {
        auto run = [&](auto kern, const char *what, int T, int P) {
            CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            for (int per_cu : {1, 2})
            {
                if (T * per_cu > 2048) continue;
                const u32 lds = (160 * 1024 / per_cu) & ~255u;
                snprintf(name, sizeof name, "%s, <= %d WG/CU", what, per_cu);
                timed(name, [&] { hipLaunchKernelGGL(kern, dim3((u32)(polys / P)), dim3(T), lds, 0, buf, (u32)polys); });
            }
        };
        run(both_passes_multi<1024, 2>, "2 polys/WG, 1024 thr", 1024, 2);
        run(both_passes_multi<1024, 4>, "4 polys/WG, 1024 thr", 1024, 4);
        run(both_passes_multi<1024, 8>, "8 polys/WG, 1024 thr", 1024, 8);
        run(both_passes_multi<256, 2>, "2 polys/WG, 256 thr", 256, 2);
        run(both_passes_multi<256, 4>, "4 polys/WG, 256 thr", 256, 4);
    }
    CK(hipFuncSetAttribute((const void *)both_passes_lds<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)both_passes_lds<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)both_passes_lds<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    timed("1 WG/poly, 1024 thr, 2/16 of the rows kept in LDS", [&] { hipLaunchKernelGGL(both_passes_lds<2>, dim3((u32)polys), dim3(1024), 2 * 32768, 0, buf, (u32)polys); });
    timed("1 WG/poly, 1024 thr, 4/16 of the rows kept in LDS", [&] { hipLaunchKernelGGL(both_passes_lds<4>, dim3((u32)polys), dim3(1024), 4 * 32768, 0, buf, (u32)polys); });
    timed("1 WG/poly, 1024 thr, 5/16 of the rows kept in LDS", [&] { hipLaunchKernelGGL(both_passes_lds<5>, dim3((u32)polys), dim3(1024), 5 * 32768, 0, buf, (u32)polys); });
    u64 h[2];
    CK(hipMemcpy(h, buf, 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h + 1, buf + polys * POLY - 1, 8, hipMemcpyDeviceToHost));
    printf("check: first word %llu, last word %llu (expected %d)\n", h[0], h[1], 2 * launches);
    CK(hipFree(buf));
    return 0;
}
