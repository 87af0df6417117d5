// What do the wait states between a VALU instruction that writes an SGPR / VCC and the VALU instruction that reads
// it cost on gfx950 (the compiler fills them with s_nop when it finds nothing else to place there)?
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_nop.hip -o tools/ubench_nop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// guard x >= K ? x + negK : x, the form the compiler emits: add64, cmp -> vcc, (wait states), 2 selects
__global__ __launch_bounds__(256) void k_guard_plain(u64 *out, u64 K, int iters)
{
    u64 x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 8 + i + K * 3;
    const u64 negK = 0 - K;
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int i = 0; i < 8; ++i)
        {
            u64 t;
            u32 lo = (u32)x[i], hi = (u32)(x[i] >> 32);
            asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(t) : "v"(x[i]), "v"(negK));
            // compare into VCC, the two wait states the compiler would insert, then the selects (volatile statements
            // keep their order; VCC carries the mask from one to the next)
            asm volatile("v_cmp_le_u64 vcc, %0, %1\n\ts_nop 1" : : "v"(K), "v"(x[i]) : "vcc");
            asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo) : "v"((u32)t) : "vcc");
            asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(hi) : "v"((u32)(t >> 32)) : "vcc");
            x[i] = (((u64)hi << 32) | lo) + it;
        }
    }
    u64 s = 0;
    for (int i = 0; i < 8; ++i) s ^= x[i];
    if (s == 0x12345) out[0] = s;
}

// the same guards, four at a time: four compares into four SGPR pairs, four adds, then the selects
__global__ __launch_bounds__(256) void k_guard_batched(u64 *out, u64 K, int iters)
{
    u64 x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 8 + i + K * 3;
    const u64 negK = 0 - K;
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int h = 0; h < 8; h += 4)
        {
            u64 m[4], t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                asm volatile("v_cmp_le_u64_e64 %0, %1, %2" : "=s"(m[i]) : "v"(K), "v"(x[h + i]));
#pragma unroll
            for (int i = 0; i < 4; ++i)
                asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(t[i]) : "v"(x[h + i]), "v"(negK));
#pragma unroll
            for (int i = 0; i < 4; ++i)
            {
                u32 lo = (u32)x[h + i], hi = (u32)(x[h + i] >> 32);
                asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(lo) : "v"((u32)t[i]), "s"(m[i]));
                asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(hi) : "v"((u32)(t[i] >> 32)), "s"(m[i]));
                x[h + i] = (((u64)hi << 32) | lo) + it;
            }
        }
    }
    u64 s = 0;
    for (int i = 0; i < 8; ++i) s ^= x[i];
    if (s == 0x12345) out[0] = s;
}

// 16 independent v_xor, with or without an s_nop 1 after each
template <int NOP>
__global__ __launch_bounds__(256) void k_xor_nop(u64 *out, u64 K, int iters)
{
    u32 x[8], y = (u32)K + threadIdx.x;
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 8 + i;
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i)
            {
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
                if (NOP == 1) asm volatile("s_nop 0");
                if (NOP == 2) asm volatile("s_nop 1");
            }
    }
    u32 s = 0;
    for (int i = 0; i < 8; ++i) s ^= x[i];
    if (s == 0x12345) out[0] = s;
}

typedef void (*kfn)(u64 *, u64, int);
static void run(const char *name, kfn k, u64 *buf, int waves_per_simd, int units)
{
    const int iters = 20000;
    const int blocks = 256 * waves_per_simd;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, buf, (u64)1 << 61, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, buf, (u64)1 << 61, iters);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double per = (double)waves_per_simd * iters * units;
    printf("%-22s w/simd=%d : %.3f ms  %.2f ns per unit per SIMD\n", name, waves_per_simd, ms, ms * 1e6 / per);
}

int main()
{
    u64 *buf;
    CK(hipMalloc(&buf, 4096));
    for (int w = 1; w <= 2; ++w)
    {
        run("guard plain", k_guard_plain, buf, w, 8);
        run("guard batched", k_guard_batched, buf, w, 8);
        run("xor", k_xor_nop<0>, buf, w, 16);
        run("xor + s_nop 0", k_xor_nop<1>, buf, w, 16);
        run("xor + s_nop 1", k_xor_nop<2>, buf, w, 16);
    }
    return 0;
}
