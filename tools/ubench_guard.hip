// Cost of the conditional subtractions ("guards") and final reductions of the lazy NTT butterflies, per 64-bit value,
// with every SIMD kept at 4 waves: candidates for the in-transform guard of the Wide policy and for the last step
// that brings a value below 16q to [0, q).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_guard.hip -o tools/ubench_guard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ u64 add64(u64 a, u64 b)
{
    u64 r;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// A: the current guard
__device__ __forceinline__ u64 guard_cmp(u64 x, u64 K, u64 negK)
{
    const u64 t = add64(x, negK);
    return x >= K ? t : x;
}
// B: subtract K when the top bit is set (3 plain 32-bit ops + one 64-bit add)
__device__ __forceinline__ u64 guard_top(u64 x, u64 negK)
{
    const u32 m = (u32)((int)(u32)(x >> 32) >> 31);
    const u64 a = ((u64)(m & (u32)(negK >> 32)) << 32) | (m & (u32)negK);
    return add64(x, a);
}
// C: borrow of the subtraction selects
__device__ __forceinline__ u64 guard_borrow(u64 x, u64 K)
{
    u32 lo, hi;
    u64 bw;
    asm("v_sub_co_u32 %0, %2, %3, %5\n\tv_subb_co_u32 %1, %2, %4, %6, %2"
        : "=&v"(lo), "=&v"(hi), "=&s"(bw)
        : "v"((u32)x), "v"((u32)(x >> 32)), "v"((u32)K), "v"((u32)(K >> 32)));
    const u64 t = ((u64)hi << 32) | lo;
    // borrow set <=> x < K
    u32 rl, rh;
    asm("v_cndmask_b32 %0, %2, %3, %6\n\tv_cndmask_b32 %1, %4, %5, %6"
        : "=&v"(rl), "=&v"(rh)
        : "v"((u32)t), "v"((u32)x), "v"((u32)(t >> 32)), "v"((u32)(x >> 32)), "s"(bw));
    return ((u64)rh << 32) | rl;
}
// D: sign of the high word of x - K selects (valid while x - K < 2^63 and K <= 2^63)
__device__ __forceinline__ u64 guard_sign(u64 x, u64 negK)
{
    const u64 t = add64(x, negK);
    return (int)(u32)(t >> 32) < 0 ? x : t;
}
// final reductions of v < 16q
__device__ __forceinline__ u64 final4(u64 v, u64 q)
{
    const u64 nq = 0 - q;
    v = guard_cmp(v, q << 3, nq << 3);
    v = guard_cmp(v, q << 2, nq << 2);
    v = guard_cmp(v, q << 1, nq << 1);
    return guard_cmp(v, q, nq);
}
// float quotient estimate (never above floor(v / q), at most 1 below for v < 16q) + one guard
__device__ __forceinline__ u64 final_float(u64 v, u64 q, float inv, u64 nq)
{
    const u32 est = (u32)(__uint2float_rz((u32)(v >> 32)) * inv);
    u64 r;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(est), "v"((u32)nq), "v"(v) : "vcc");
    const u32 hi = (u32)(r >> 32) + est * (u32)(nq >> 32);
    r = ((u64)hi << 32) | (u32)r;
    r = guard_cmp(r, q << 1, nq << 1);
    return guard_cmp(r, q, nq);
}

template <int MODE>
__global__ __launch_bounds__(256) void k(u64 *out, u64 q, u64 seed, int iters, float inv)
{
    u64 x[8];
    const u64 K = q << 3, negK = 0 - K, nq = 0 - q;
    for (int i = 0; i < 8; ++i)
        x[i] = seed * (threadIdx.x * 8 + i + 1);
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int i = 0; i < 8; ++i)
        {
            u64 v = x[i] + seed; // one v_lshl_add-class op per value keeps the chains data dependent and moving
            if (MODE == 0) v = v;
            if (MODE == 1) v = guard_cmp(v, K, negK);
            if (MODE == 2) v = guard_top(v, negK);
            if (MODE == 3) v = guard_borrow(v, K);
            if (MODE == 4) v = guard_sign(v, negK);
            if (MODE == 5) v = final4(v, q);
            if (MODE == 6) v = final_float(v, q, inv, nq);
            x[i] = v;
        }
    }
    u64 s = 0;
    for (int i = 0; i < 8; ++i)
        s ^= x[i];
    if (s == 0x12345)
        out[0] = s;
}

template <int MODE>
static float run(const char *name, u64 *buf, float base)
{
    const int iters = 20000, blocks = 1024;
    const u64 q = 1152921504606584833ull;
    const float inv = 4294967296.0f / (float)q * 0.9999f;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, q, 0x9e3779b97f4a7c15ull, iters, inv);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, q, 0x9e3779b97f4a7c15ull, iters, inv);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double per = ms * 1e-3 * 2.4e9 / ((double)blocks * 4 / 1024.0 * iters * 8);
    printf("%-44s %.3f ms  %.2f cyc @2.4GHz per value, %.2f over the bare loop\n", name, ms, per, per - base);
    return (float)per;
}

int main()
{
    u64 *buf;
    CK(hipMalloc(&buf, 4096));
    const float base = run<0>("bare loop (one 64-bit add per value)", buf, 0);
    run<1>("A guard: add64 + cmp_u64 + 2 cndmask", buf, base);
    run<2>("B guard: top bit mask, 2 and, add64", buf, base);
    run<3>("C guard: sub_co/subb_co + 2 cndmask", buf, base);
    run<4>("D guard: add64 + cmp_i32 + 2 cndmask", buf, base);
    run<5>("final: 4 guards (8q, 4q, 2q, q)", buf, base);
    run<6>("final: float estimate + mad + 2 guards", buf, base);
    return 0;
}
