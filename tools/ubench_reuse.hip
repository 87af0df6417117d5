// Does a second pass over a polynomial come from cache when the SAME workgroup made the first pass a moment ago?
// The batched NTT moves every polynomial (N = 2^16 words = 512 KB) through HBM twice (column pass, row pass).  This
// measures the memory side of the alternative without any arithmetic: one workgroup owns one polynomial, reads and
// rewrites it in the column pattern, synchronises, then reads and rewrites it in the row pattern -- against the same
// two patterns as two launches over the whole buffer.  Build:
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_reuse.hip -o tools/ubench_reuse
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr u32 POLY = 65536; // words

// column pattern of ntt_fwd_cols<.,8>: a tile = 16 columns x 256 rows; 256 threads, 16 words each
__device__ __forceinline__ void cols_tile(u64 *poly, u32 tile, u32 tid)
{
    const u32 c = tid & 15, tau = tid >> 4;
    u64 *p = poly + tile * 16;
    u64 x[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = p[(size_t)(e * 16 + tau) * 256 + c];
#pragma unroll
    for (int e = 0; e < 16; ++e) p[(size_t)(e * 16 + tau) * 256 + c] = x[e] + 1; // same words: no cross-thread hazard
}
// row pattern: a tile = 16 rows of 256 contiguous words
__device__ __forceinline__ void rows_tile(u64 *poly, u32 tile, u32 tid)
{
    u64 *p = poly + (size_t)tile * 4096;
    u64 x[16];
#pragma unroll
    for (int h = 0; h < 16; ++h) x[h] = p[h * 256 + tid];
#pragma unroll
    for (int h = 0; h < 16; ++h) p[h * 256 + tid] = x[h] + 1;
}

__global__ __launch_bounds__(256) void pass_cols(u64 *buf)
{
    cols_tile(buf + (size_t)(blockIdx.x >> 4) * POLY, blockIdx.x & 15, threadIdx.x);
}
__global__ __launch_bounds__(256) void pass_rows(u64 *buf)
{
    rows_tile(buf + (size_t)(blockIdx.x >> 4) * POLY, blockIdx.x & 15, threadIdx.x);
}
// one workgroup of T threads (T / 256 tiles at a time) owns polynomial blockIdx.x for both passes
template <int T>
__global__ __launch_bounds__(T) void both_passes(u64 *buf)
{
    u64 *poly = buf + (size_t)blockIdx.x * POLY;
    const u32 sub = threadIdx.x >> 8, tid = threadIdx.x & 255;
    for (u32 tile = sub; tile < 16; tile += T / 256) cols_tile(poly, tile, tid);
    __syncthreads(); // the workgroup's own stores are visible to it after the barrier (same CU, write-through L1)
    for (u32 tile = sub; tile < 16; tile += T / 256) rows_tile(poly, tile, tid);
}

// the same with P polynomials per workgroup (what sharing one staged twiddle tile between polynomials would need):
// all column passes first, then all row passes tile group by tile group across the P polynomials
template <int P>
__global__ __launch_bounds__(256) void both_passes_multi(u64 *buf)
{
    u64 *base = buf + (size_t)blockIdx.x * P * POLY;
    for (u32 p = 0; p < P; ++p)
        for (u32 tile = 0; tile < 16; ++tile) cols_tile(base + (size_t)p * POLY, tile, threadIdx.x);
    __syncthreads();
    for (u32 tile = 0; tile < 16; ++tile)
        for (u32 p = 0; p < P; ++p) rows_tile(base + (size_t)p * POLY, tile, threadIdx.x);
}

int main(int argc, char **argv)
{
    const size_t polys = argc > 1 ? (size_t)atol(argv[1]) : 16384; // 16384 x 512 KB = 8 GiB
    const size_t bytes = polys * POLY * sizeof(u64);
    u64 *buf = nullptr;
    CK(hipMalloc(&buf, bytes));
    CK(hipMemset(buf, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timed = [&](const char *name, auto launch) {
        launch();
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 5; ++r)
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
        // algorithmic bytes of one transform = read + write once
        printf("%-34s %8.3f ms   %7.1f GB/s algorithmic (2 x %.2f GB / t)\n", name, best, 2.0 * bytes / best / 1e6, bytes / 1e9);
    };
    const u32 tiles = (u32)(polys * 16);
    timed("two launches (cols, rows)", [&] {
        hipLaunchKernelGGL(pass_cols, dim3(tiles), dim3(256), 0, 0, buf);
        hipLaunchKernelGGL(pass_rows, dim3(tiles), dim3(256), 0, 0, buf);
    });
    timed("one workgroup per poly, 256 thr", [&] { hipLaunchKernelGGL(both_passes<256>, dim3((u32)polys), dim3(256), 0, 0, buf); });
    timed("one workgroup per poly, 512 thr", [&] { hipLaunchKernelGGL(both_passes<512>, dim3((u32)polys), dim3(512), 0, 0, buf); });
    timed("one workgroup per poly, 1024 thr", [&] { hipLaunchKernelGGL(both_passes<1024>, dim3((u32)polys), dim3(1024), 0, 0, buf); });
    timed("one workgroup per 2 polys, 256 thr", [&] { hipLaunchKernelGGL(both_passes_multi<2>, dim3((u32)(polys / 2)), dim3(256), 0, 0, buf); });
    timed("one workgroup per 4 polys, 256 thr", [&] { hipLaunchKernelGGL(both_passes_multi<4>, dim3((u32)(polys / 4)), dim3(256), 0, 0, buf); });
    timed("one workgroup per 8 polys, 256 thr", [&] { hipLaunchKernelGGL(both_passes_multi<8>, dim3((u32)(polys / 8)), dim3(256), 0, 0, buf); });
    // every word was incremented 2 x (1 warm-up + 5) x 7 variants times
    u64 h[2];
    CK(hipMemcpy(h, buf, 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h + 1, buf + polys * POLY - 1, 8, hipMemcpyDeviceToHost));
    printf("check: first word %llu, last word %llu (expected %d)\n", h[0], h[1], 2 * 6 * 7);
    CK(hipFree(buf));
    return 0;
}
