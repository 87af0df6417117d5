// Micro-benchmarks that bound the NTT kernels on gfx950: streaming copy, column-tile access pattern,
// and in-register Shoup butterfly throughput.  Build: hipcc -O3 --offload-arch=gfx950 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void copy16(const ulonglong2 *in, ulonglong2 *out, size_t n2)
{
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) out[i] = in[i];
}
__global__ __launch_bounds__(256) void copy8_tile(const u64 *in, u64 *out)
{
    // one block = 4096 contiguous words, 16 words per thread at stride 256 (row kernel pattern)
    size_t base = (size_t)blockIdx.x * 4096;
    u64 x[16];
#pragma unroll
    for (int h = 0; h < 16; ++h) x[h] = in[base + h * 256 + threadIdx.x];
#pragma unroll
    for (int h = 0; h < 16; ++h) out[base + h * 256 + threadIdx.x] = x[h];
}
__global__ __launch_bounds__(256) void copy8_cols(const u64 *in, u64 *out)
{
    // cols<8> pattern: block = 16 columns x 256 rows of a 65536-word poly; thread (tau, c): rows e*16+tau
    const u32 tile = blockIdx.x & 15;
    size_t base = (size_t)(blockIdx.x >> 4) * 65536 + tile * 16;
    const u32 c = threadIdx.x & 15, tau = threadIdx.x >> 4;
    u64 x[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = in[base + (size_t)(e * 16 + tau) * 256 + c];
#pragma unroll
    for (int e = 0; e < 16; ++e) out[base + (size_t)(tau * 16 + e) * 256 + c] = x[e];
}
__device__ __forceinline__ u64 mul_lazy(u64 x, u64 w, u64 wq, u64 q) { return w * x - __umul64hi(x, wq) * q; }
template <int MODE>
__global__ __launch_bounds__(256) void bfly(u64 *out, u64 q, u64 w, u64 wq, int iters)
{
    u64 x[16];
    const u64 q2 = q << 1;
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = (threadIdx.x * 16 + i + blockIdx.x) % q;
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int j = 0; j < 8; ++j)
        {
            if (MODE == 0)
            { // full Harvey butterfly
                u64 u = x[j] >= q2 ? x[j] - q2 : x[j];
                u64 v = mul_lazy(x[j + 8], w, wq, q);
                x[j] = u + v;
                x[j + 8] = u + q2 - v;
            }
            else if (MODE == 1)
            { // multiply only
                x[j + 8] = mul_lazy(x[j + 8], w, wq, q);
                x[j] += x[j + 8];
            }
            else
            { // add/guard only
                u64 u = x[j] >= q2 ? x[j] - q2 : x[j];
                u64 v = x[j + 8];
                x[j] = u + v;
                x[j + 8] = u + q2 - v;
            }
        }
        w += 2; // keep the compiler from hoisting
    }
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s ^= x[i];
    if (s == 12345) out[0] = s;
}
template <typename F>
float timeit(F f, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}
int main()
{
    const size_t words = (size_t)1 << 31; // 16 GiB per buffer... use 8 GiB
    const size_t W = (size_t)1 << 30;     // 8 GiB
    u64 *a, *b;
    CK(hipMalloc(&a, W * 8)); CK(hipMalloc(&b, W * 8));
    CK(hipMemset(a, 1, W * 8)); CK(hipMemset(b, 2, W * 8));
    (void)words;
    float ms = timeit([&] { hipLaunchKernelGGL(copy16, dim3(256 * 16), dim3(256), 0, 0, (const ulonglong2 *)a, (ulonglong2 *)b, W / 2); }, 5);
    printf("copy16 grid-stride : %.3f ms  %.1f GB/s (r+w)\n", ms, 2.0 * W * 8 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(copy8_tile, dim3(W / 4096), dim3(256), 0, 0, a, b); }, 5);
    printf("copy8 row tiles    : %.3f ms  %.1f GB/s (r+w)\n", ms, 2.0 * W * 8 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(copy8_cols, dim3(W / 4096), dim3(256), 0, 0, a, b); }, 5);
    printf("copy8 col tiles    : %.3f ms  %.1f GB/s (r+w)\n", ms, 2.0 * W * 8 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(copy8_cols, dim3(W / 4096), dim3(256), 0, 0, a, a); }, 5);
    printf("copy8 col in place : %.3f ms  %.1f GB/s (r+w)\n", ms, 2.0 * W * 8 / ms / 1e6);
    const u64 q = 1152921504606584833ULL, w = 288794978602139552ULL % q;
    const u64 wq = (u64)((((unsigned __int128)w) << 64) / q);
    const int iters = 2000, blocks = 256 * 8;
    const double nb = (double)blocks * 256 * 8 * iters;
    ms = timeit([&] { hipLaunchKernelGGL(bfly<0>, dim3(blocks), dim3(256), 0, 0, b, q, w, wq, iters); }, 3);
    printf("butterfly full     : %.3f ms  %.2f Gbfly/s  (cfg2 fwd NTT = 17.18 Gbfly -> %.2f ms)\n", ms, nb / ms / 1e6, 17.18e9 / (nb / ms * 1e3));
    ms = timeit([&] { hipLaunchKernelGGL(bfly<1>, dim3(blocks), dim3(256), 0, 0, b, q, w, wq, iters); }, 3);
    printf("shoup mul + add    : %.3f ms  %.2f Gmul/s\n", ms, nb / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(bfly<2>, dim3(blocks), dim3(256), 0, 0, b, q, w, wq, iters); }, 3);
    printf("guard/add/sub only : %.3f ms  %.2f Gop/s\n", ms, nb / ms / 1e6);
    return 0;
}
