// What read+write rate does HBM give for the access shapes of the NTT passes?  In-place read-modify-write of a 16 GiB
// buffer (like the transform), by access width per lane and by the shape of a workgroup's footprint.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_copy.hip -o tools/ubench_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// A: 8 B per lane, a workgroup updates 32 KB contiguous (16 x 2 KB rows, the row-pass shape)
__global__ __launch_bounds__(256) void rmw_rows8(u64 *buf)
{
    u64 *p = buf + (size_t)blockIdx.x * 4096;
    u64 x[16];
#pragma unroll
    for (int h = 0; h < 16; ++h) x[h] = p[h * 256 + threadIdx.x];
#pragma unroll
    for (int h = 0; h < 16; ++h) p[h * 256 + threadIdx.x] = x[h] + 1;
}
// B: 16 B per lane, same footprint
__global__ __launch_bounds__(256) void rmw_rows16(u64 *buf)
{
    ulonglong2 *p = reinterpret_cast<ulonglong2 *>(buf + (size_t)blockIdx.x * 4096);
    ulonglong2 x[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) x[h] = p[h * 256 + threadIdx.x];
#pragma unroll
    for (int h = 0; h < 8; ++h) { x[h].x += 1; x[h].y += 1; p[h * 256 + threadIdx.x] = x[h]; }
}
// C: the column-pass shape: 256 rows x 128-byte segments, 8 B per lane (16 lanes per segment)
__global__ __launch_bounds__(256) void rmw_cols8(u64 *buf)
{
    u64 *p = buf + (size_t)(blockIdx.x >> 4) * 65536 + (blockIdx.x & 15) * 16;
    const u32 c = threadIdx.x & 15, tau = threadIdx.x >> 4;
    u64 x[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = p[(size_t)(e * 16 + tau) * 256 + c];
#pragma unroll
    for (int e = 0; e < 16; ++e) p[(size_t)(e * 16 + tau) * 256 + c] = x[e] + 1;
}
// D: column shape with 16 B per lane: 256 rows x 128-byte segments, 8 lanes per segment (a thread owns 2 columns, 8 rows)
__global__ __launch_bounds__(256) void rmw_cols16(u64 *buf)
{
    u64 *p = buf + (size_t)(blockIdx.x >> 4) * 65536 + (blockIdx.x & 15) * 16;
    const u32 c2 = threadIdx.x & 7, tau = threadIdx.x >> 3; // tau < 32
    ulonglong2 x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = *reinterpret_cast<ulonglong2 *>(p + (size_t)(e * 32 + tau) * 256 + c2 * 2);
#pragma unroll
    for (int e = 0; e < 8; ++e) { x[e].x += 1; x[e].y += 1; *reinterpret_cast<ulonglong2 *>(p + (size_t)(e * 32 + tau) * 256 + c2 * 2) = x[e]; }
}
// E: column shape with 256-byte segments (32 columns x 128 rows per workgroup... 2 workgroups per 32-column strip)
__global__ __launch_bounds__(256) void rmw_cols8_wide(u64 *buf)
{
    // tile = 32 columns x 128 rows: blockIdx -> (poly, strip of 32 columns, half of the rows)
    const u32 b = blockIdx.x & 15;
    u64 *p = buf + (size_t)(blockIdx.x >> 4) * 65536 + (b >> 1) * 32 + (size_t)(b & 1) * 128 * 256;
    const u32 c = threadIdx.x & 31, tau = threadIdx.x >> 5; // tau < 8
    u64 x[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = p[(size_t)(e * 8 + tau) * 256 + c];
#pragma unroll
    for (int e = 0; e < 16; ++e) p[(size_t)(e * 8 + tau) * 256 + c] = x[e] + 1;
}
// F: plain grid-stride 16 B per lane, out of place (a copy)
__global__ __launch_bounds__(256) void copy16(const ulonglong2 *a, ulonglong2 *b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
// G: nt variants of A
__global__ __launch_bounds__(256) void rmw_rows8_nt(u64 *buf)
{
    u64 *p = buf + (size_t)blockIdx.x * 4096;
    u64 x[16];
#pragma unroll
    for (int h = 0; h < 16; ++h) x[h] = __builtin_nontemporal_load(p + h * 256 + threadIdx.x);
#pragma unroll
    for (int h = 0; h < 16; ++h) __builtin_nontemporal_store(x[h] + 1, p + h * 256 + threadIdx.x);
}
// H: read only / write only of the row shape
__global__ __launch_bounds__(256) void read_rows8(u64 *buf, u64 *sink)
{
    u64 *p = buf + (size_t)blockIdx.x * 4096;
    u64 s = 0;
#pragma unroll
    for (int h = 0; h < 16; ++h) s += p[h * 256 + threadIdx.x];
    if (s == 0x1234567) sink[0] = s;
}
__global__ __launch_bounds__(256) void write_rows8(u64 *buf)
{
    u64 *p = buf + (size_t)blockIdx.x * 4096;
#pragma unroll
    for (int h = 0; h < 16; ++h) p[h * 256 + threadIdx.x] = blockIdx.x + h;
}

int main()
{
    const size_t polys = 16384, words = polys * 65536, bytes = words * 8; // 8 GiB
    u64 *buf, *buf2;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&buf2, bytes));
    CK(hipMemset(buf, 0, bytes));
    CK(hipMemset(buf2, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timed = [&](const char *name, double moved, auto launch) {
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
        printf("%-64s %8.3f ms   %7.1f GB/s (bytes read + written)\n", name, best, moved / best / 1e6);
        fflush(stdout);
    };
    const u32 tiles = (u32)(words / 4096);
    timed("A rows, 8 B/lane, in place", 2.0 * bytes, [&] { hipLaunchKernelGGL(rmw_rows8, dim3(tiles), dim3(256), 0, 0, buf); });
    timed("B rows, 16 B/lane, in place", 2.0 * bytes, [&] { hipLaunchKernelGGL(rmw_rows16, dim3(tiles), dim3(256), 0, 0, buf); });
    timed("C cols (128-B segments), 8 B/lane, in place", 2.0 * bytes, [&] { hipLaunchKernelGGL(rmw_cols8, dim3(tiles), dim3(256), 0, 0, buf); });
    timed("D cols (128-B segments), 16 B/lane, in place", 2.0 * bytes, [&] { hipLaunchKernelGGL(rmw_cols16, dim3(tiles), dim3(256), 0, 0, buf); });
    timed("E cols (256-B segments), 8 B/lane, in place", 2.0 * bytes, [&] { hipLaunchKernelGGL(rmw_cols8_wide, dim3(tiles), dim3(256), 0, 0, buf); });
    timed("F copy, 16 B/lane, grid-stride, 2048 workgroups", 2.0 * bytes, [&] { hipLaunchKernelGGL(copy16, dim3(2048), dim3(256), 0, 0, (const ulonglong2 *)buf, (ulonglong2 *)buf2, words / 2); });
    timed("F copy, 16 B/lane, grid-stride, 16384 workgroups", 2.0 * bytes, [&] { hipLaunchKernelGGL(copy16, dim3(16384), dim3(256), 0, 0, (const ulonglong2 *)buf, (ulonglong2 *)buf2, words / 2); });
    timed("G rows, 8 B/lane, in place, nontemporal", 2.0 * bytes, [&] { hipLaunchKernelGGL(rmw_rows8_nt, dim3(tiles), dim3(256), 0, 0, buf); });
    timed("H rows, read only", 1.0 * bytes, [&] { hipLaunchKernelGGL(read_rows8, dim3(tiles), dim3(256), 0, 0, buf, buf2); });
    timed("H rows, write only", 1.0 * bytes, [&] { hipLaunchKernelGGL(write_rows8, dim3(tiles), dim3(256), 0, 0, buf); });
    timed("hipMemcpyAsync device to device", 2.0 * bytes, [&] { CK(hipMemcpyAsync(buf2, buf, bytes, hipMemcpyDeviceToDevice, 0)); });
    return 0;
}
