// Cost-model probe: how integer multiplies and simple VALU ops add up when interleaved (8 and 4 waves/SIMD).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_mix.hip -o tools/ubench_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64; typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
// NM multiplies (kind MK) and NA simple ops per group, 8 independent chains
template <int NM, int NA, int MK>
__global__ __launch_bounds__(256) void mix(u32 *out, u32 a, int iters)
{
    u64 x[8]; u32 y[8];
    u32 c = a + threadIdx.x, d = a ^ (threadIdx.x * 7 + 1);
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 8 + i + a; y[i] = threadIdx.x + i; }
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int r = 0; r < (NM > NA ? NM : NA); ++r)
        {
#pragma unroll
            for (int i = 0; i < 8; ++i)
            {
                if (r < NM)
                {
                    if (MK == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[i]) : "v"(c), "v"(d) : "vcc");
                    else if (MK == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(y[i]) : "v"(c));
                    else asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(y[i]) : "v"(c));
                }
                if (r < NA)
                {
                    if (MK == 1 || MK == 2) { u32 lo = (u32)x[i]; asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(d)); x[i] = lo; }
                    else asm volatile("v_add_u32 %0, %0, %1" : "+v"(y[i]) : "v"(d));
                }
            }
        }
    }
    u64 s = 0;
    for (int i = 0; i < 8; ++i) s ^= x[i] ^ y[i];
    if (s == 0x12345) out[0] = (u32)s;
}
typedef void (*kfn)(u32 *, u32, int);
static void run(const char *name, kfn k, u32 *buf, int wps, int nm, int na)
{
    const int iters = 20000, blocks = 256 * wps;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, buf, 3u, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, buf, 3u, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double groups = (double)wps * iters * 8; // per SIMD: wave-groups (one group = NM mul + NA add)
    printf("%-22s w/simd=%d  %7.3f ms  %6.1f cyc per (%d mul + %d add)\n", name, wps, ms, ms * 1e-3 * 2.4e9 / groups, nm, na);
}
int main()
{
    u32 *buf; CK(hipMalloc(&buf, 4096));
    for (int wps : {8, 4, 2})
    {
        run("mad64 x10", mix<10, 0, 0>, buf, wps, 10, 0);
        run("mul_lo x10", mix<10, 0, 1>, buf, wps, 10, 0);
        run("add x20", mix<0, 20, 0>, buf, wps, 0, 20);
        run("add x10", mix<0, 10, 0>, buf, wps, 0, 10);
        run("mad64 x10 + add x10", mix<10, 10, 0>, buf, wps, 10, 10);
        run("mad64 x10 + add x20", mix<10, 20, 0>, buf, wps, 10, 20);
        run("mul_lo x10 + add x20", mix<10, 20, 1>, buf, wps, 10, 20);
        run("mad64 x5 + add x20", mix<5, 20, 0>, buf, wps, 5, 20);
    }
    return 0;
}
