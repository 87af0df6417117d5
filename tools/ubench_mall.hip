// Does the Infinity Cache serve a chunk that was written just before?  The question behind a chunk-pipelined NTT
// (pass 2 of a chunk right behind its pass 1).  Same launches in both sequences, only the target of
// every second launch differs:
//   hot : for every chunk c:  rmw(c); rmw(c)              the second launch re-reads what the first wrote
//   cold: for every chunk c:  rmw(c); rmw(c + half)       the second launch touches memory last used long ago
// rmw = read 16 B per lane, add, write back in place (the access skeleton of an NTT pass), one trip per workgroup of 1024.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_mall.hip -o tools/ubench_mall
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <bool NT>
__global__ __launch_bounds__(1024) void rmw(u64x2 *p)
{
    u64x2 *q = p + (size_t)blockIdx.x * 4096 + threadIdx.x;
    u64x2 x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        x[u] = NT ? __builtin_nontemporal_load(q + u * 1024) : q[u * 1024];
#pragma unroll
    for (int u = 0; u < 4; ++u)
    {
        x[u].x += 1;
        if (NT)
            __builtin_nontemporal_store(x[u], q + u * 1024);
        else
            q[u * 1024] = x[u];
    }
}
template <bool NT>
__global__ __launch_bounds__(1024) void rd(const u64x2 *p, u64 *sink)
{
    const u64x2 *q = p + (size_t)blockIdx.x * 4096 + threadIdx.x;
    u64 s = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u)
    {
        const u64x2 v = NT ? __builtin_nontemporal_load(q + u * 1024) : q[u * 1024];
        s += v.x ^ v.y;
    }
    if (s == 0x123456789abcdefull)
        sink[0] = s;
}

int main()
{
    const size_t total = (size_t)8 << 30;
    char *buf;
    u64 *sink;
    CK(hipMalloc(&buf, total));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, total));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t half = total / 2;
    for (int nt = 0; nt < 2; ++nt)
        for (size_t mb : {8, 16, 32, 64, 128, 192, 256, 512})
        {
            const size_t S = mb << 20, chunks = half / S;
            const unsigned grid = (unsigned)(S / 65536);
            float ms[4];
            for (int mode = 0; mode < 4; ++mode) // 0 hot rmw+rmw, 1 cold rmw+rmw, 2 hot rmw+read, 3 cold rmw+read
            {
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep)
                {
                    CK(hipEventRecord(e0));
                    for (size_t c = 0; c < chunks; ++c)
                    {
                        u64x2 *a = (u64x2 *)(buf + c * S);
                        u64x2 *b = (u64x2 *)(buf + c * S + ((mode & 1) ? half : 0));
                        if (nt)
                            hipLaunchKernelGGL(rmw<true>, dim3(grid), dim3(1024), 0, 0, a);
                        else
                            hipLaunchKernelGGL(rmw<false>, dim3(grid), dim3(1024), 0, 0, a);
                        if (mode < 2)
                        {
                            if (nt)
                                hipLaunchKernelGGL(rmw<true>, dim3(grid), dim3(1024), 0, 0, b);
                            else
                                hipLaunchKernelGGL(rmw<false>, dim3(grid), dim3(1024), 0, 0, b);
                        }
                        else
                        {
                            if (nt)
                                hipLaunchKernelGGL(rd<true>, dim3(grid), dim3(1024), 0, 0, b, sink);
                            else
                                hipLaunchKernelGGL(rd<false>, dim3(grid), dim3(1024), 0, 0, b, sink);
                        }
                    }
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float t;
                    CK(hipEventElapsedTime(&t, e0, e1));
                    if (t < best)
                        best = t;
                }
                ms[mode] = best;
            }
            CK(hipGetLastError());
            printf("%s chunk %4zu MB x %4zu: rmw+rmw hot %7.3f ms cold %7.3f ms (hot/cold %.3f) | rmw+read hot %7.3f ms cold %7.3f ms (hot/cold %.3f)\n",
                   nt ? "nt     " : "default", mb, chunks, ms[0], ms[1], ms[0] / ms[1], ms[2], ms[3], ms[2] / ms[3]);
            fflush(stdout);
        }
    return 0;
}
