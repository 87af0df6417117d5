// Which device copy reaches the highest read + write rate on this MI355X?  VERDICT r02 item 1(a): the guide quotes
// 6.29 TB/s for a float4 copy while the transforms' passes are anchored on copies at 4.9-5.0 TB/s.  This sweeps the
// launch shape (workgroups per CU, threads, 16-byte pieces in flight per lane), the cache policy of either side
// (nontemporal loads / stores), in place against out of place, and the offset between source and destination.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_copy_sweep.hip -o tools/ubench_copy_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
typedef unsigned long long u64;
typedef unsigned int u32;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// a workgroup moves U * blockDim consecutive 16-byte pieces per trip, all U loads of a lane issued before its stores;
// trips are handed out round robin over the grid (grid == number of trips: one trip per workgroup, not persistent)
template <int U, bool NTL, bool NTS, bool RMW>
__global__ void copy_k(const u64x2 *__restrict__ src, u64x2 *__restrict__ dst, size_t n_pieces)
{
    const size_t per_trip = (size_t)U * blockDim.x;
    for (size_t trip = blockIdx.x; trip * per_trip < n_pieces; trip += gridDim.x)
    {
        const size_t base = trip * per_trip + threadIdx.x;
        u64x2 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            x[u] = NTL ? __builtin_nontemporal_load(src + base + (size_t)u * blockDim.x) : src[base + (size_t)u * blockDim.x];
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            if (RMW)
                x[u].x += 1;
            if (NTS)
                __builtin_nontemporal_store(x[u], dst + base + (size_t)u * blockDim.x);
            else
                dst[base + (size_t)u * blockDim.x] = x[u];
        }
    }
}

// read only / write only at the same shapes (what each direction gives alone)
template <int U, bool NT>
__global__ void read_k(const u64x2 *__restrict__ src, u64 *sink, size_t n_pieces)
{
    const size_t per_trip = (size_t)U * blockDim.x;
    u64 s = 0;
    for (size_t trip = blockIdx.x; trip * per_trip < n_pieces; trip += gridDim.x)
    {
        const size_t base = trip * per_trip + threadIdx.x;
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            const u64x2 v = NT ? __builtin_nontemporal_load(src + base + (size_t)u * blockDim.x) : src[base + (size_t)u * blockDim.x];
            s += v.x ^ v.y;
        }
    }
    if (s == 0x123456789abcdefull)
        sink[0] = s;
}
template <int U, bool NT>
__global__ void write_k(u64x2 *__restrict__ dst, size_t n_pieces)
{
    const size_t per_trip = (size_t)U * blockDim.x;
    for (size_t trip = blockIdx.x; trip * per_trip < n_pieces; trip += gridDim.x)
    {
        const size_t base = trip * per_trip + threadIdx.x;
        u64x2 v;
        v.x = base;
        v.y = trip;
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            if (NT)
                __builtin_nontemporal_store(v, dst + base + (size_t)u * blockDim.x);
            else
                dst[base + (size_t)u * blockDim.x] = v;
        }
    }
}

struct Row
{
    double gbps;
    char text[160];
};

int main(int argc, char **argv)
{
    const size_t bytes = argc > 1 ? (size_t)atoll(argv[1]) << 20 : (size_t)8 << 30; // per buffer
    const size_t n_pieces = bytes / 16;
    char *pool;
    const size_t slack = 64u << 20;
    CK(hipMalloc(&pool, 2 * bytes + slack));
    CK(hipMemset(pool, 1, 2 * bytes + slack));
    u64 *sink;
    CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<Row> rows;
    auto timed = [&](const char *name, double moved, auto launch) {
        launch();
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 3; ++r)
        {
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best)
                best = ms;
        }
        CK(hipGetLastError());
        Row r;
        r.gbps = moved / best / 1e6;
        snprintf(r.text, sizeof r.text, "%-92s %8.3f ms %8.1f GB/s", name, best, r.gbps);
        printf("%s\n", r.text);
        fflush(stdout);
        rows.push_back(r);
    };
    const u64x2 *src = (const u64x2 *)pool;
    char name[160];

    // ---- 1. launch shape, default cache policy, out of place, destination right behind the source -------------
    const int threads_list[] = {256, 512, 1024};
    const int wg_per_cu[] = {0, 2, 4, 8, 16, 32}; // 0 = one trip per workgroup (grid = trips)
#define SHAPE(U)                                                                                                       \
    for (int threads : threads_list)                                                                                   \
        for (int wpc : wg_per_cu)                                                                                      \
        {                                                                                                              \
            if (wpc * threads > 2048 * 4 && wpc)                                                                       \
                continue;                                                                                              \
            const size_t trips = n_pieces / ((size_t)U * threads);                                                     \
            const u32 grid = wpc ? (u32)(256 * wpc) : (u32)trips;                                                      \
            snprintf(name, sizeof name, "copy U=%d (x16 B per lane) threads=%d grid=%s%d", U, threads,                 \
                     wpc ? "256x" : "trips=", wpc ? wpc : (int)grid);                                                  \
            u64x2 *dst = (u64x2 *)(pool + bytes);                                                                      \
            timed(name, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_k<U, false, false, false>), dim3(grid), dim3(threads), 0, 0, src, dst, n_pieces); }); \
        }
    SHAPE(1)
    SHAPE(2)
    SHAPE(4)
    SHAPE(8)

    // ---- 2. cache policy of either side at the three best shapes' neighbourhood ------------------------------
#define POLICY(U, NTL, NTS, threads, grid, label)                                                                      \
    {                                                                                                                  \
        snprintf(name, sizeof name, "copy U=%d threads=%d grid=%u loads=%s stores=%s %s", U, threads, (u32)(grid),     \
                 NTL ? "nt" : "default", NTS ? "nt" : "default", label);                                               \
        u64x2 *dst = (u64x2 *)(pool + bytes);                                                                          \
        timed(name, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_k<U, NTL, NTS, false>), dim3(grid), dim3(threads), 0, 0, src, dst, n_pieces); }); \
    }
    for (int wpc : {4, 8, 16})
    {
        POLICY(4, true, false, 256, 256 * wpc, "")
        POLICY(4, false, true, 256, 256 * wpc, "")
        POLICY(4, true, true, 256, 256 * wpc, "")
        POLICY(8, true, true, 256, 256 * wpc, "")
        POLICY(2, true, true, 512, 256 * wpc, "")
    }
    POLICY(1, true, true, 256, n_pieces / 256, "(one piece per thread)")
    POLICY(1, false, true, 256, n_pieces / 256, "(one piece per thread)")
    POLICY(1, true, false, 256, n_pieces / 256, "(one piece per thread)")
    POLICY(4, true, true, 256, n_pieces / 1024, "(one trip per workgroup)")
    POLICY(8, true, true, 256, n_pieces / 2048, "(one trip per workgroup)")

    // ---- 3. offset between source and destination (HBM channel / bank interleave), nt both sides and default -----
    for (size_t off : {(size_t)0, (size_t)256, (size_t)4096, (size_t)65536, (size_t)(1u << 20) + 4096, (size_t)(32u << 20) + 8192 + 256})
    {
        u64x2 *dst = (u64x2 *)(pool + bytes + off);
        snprintf(name, sizeof name, "copy U=4 threads=256 grid=256x8 nt both, destination = source + buffer + %zu B", off);
        timed(name, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_k<4, true, true, false>), dim3(2048), dim3(256), 0, 0, src, dst, n_pieces); });
        snprintf(name, sizeof name, "copy U=4 threads=256 grid=256x8 default, destination = source + buffer + %zu B", off);
        timed(name, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_k<4, false, false, false>), dim3(2048), dim3(256), 0, 0, src, dst, n_pieces); });
    }

    // ---- 4. in place (read, modify, write back the same lines) ------------------------------------------------------
    for (int wpc : {0, 4, 8, 16})
    {
        const u32 grid4 = wpc ? 256 * wpc : (u32)(n_pieces / 1024), grid8 = wpc ? 256 * wpc : (u32)(n_pieces / 2048);
        u64x2 *dst = (u64x2 *)pool;
        snprintf(name, sizeof name, "in place U=4 threads=256 grid=%u default", grid4);
        timed(name, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_k<4, false, false, true>), dim3(grid4), dim3(256), 0, 0, src, dst, n_pieces); });
        snprintf(name, sizeof name, "in place U=4 threads=256 grid=%u nt both", grid4);
        timed(name, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_k<4, true, true, true>), dim3(grid4), dim3(256), 0, 0, src, dst, n_pieces); });
        snprintf(name, sizeof name, "in place U=8 threads=256 grid=%u nt both", grid8);
        timed(name, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_k<8, true, true, true>), dim3(grid8), dim3(256), 0, 0, src, dst, n_pieces); });
        snprintf(name, sizeof name, "in place U=8 threads=256 grid=%u nt stores only", grid8);
        timed(name, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_k<8, false, true, true>), dim3(grid8), dim3(256), 0, 0, src, dst, n_pieces); });
    }

    // ---- 5. each direction alone ------------------------------------------------------------------------------------
    for (int wpc : {0, 8, 16})
    {
        const u32 grid = wpc ? 256 * wpc : (u32)(n_pieces / 1024);
        snprintf(name, sizeof name, "read only U=4 threads=256 grid=%u default", grid);
        timed(name, 1.0 * bytes, [&] { hipLaunchKernelGGL((read_k<4, false>), dim3(grid), dim3(256), 0, 0, src, sink, n_pieces); });
        snprintf(name, sizeof name, "read only U=4 threads=256 grid=%u nt", grid);
        timed(name, 1.0 * bytes, [&] { hipLaunchKernelGGL((read_k<4, true>), dim3(grid), dim3(256), 0, 0, src, sink, n_pieces); });
        snprintf(name, sizeof name, "write only U=4 threads=256 grid=%u default", grid);
        timed(name, 1.0 * bytes, [&] { hipLaunchKernelGGL((write_k<4, false>), dim3(grid), dim3(256), 0, 0, (u64x2 *)(pool + bytes), n_pieces); });
        snprintf(name, sizeof name, "write only U=4 threads=256 grid=%u nt", grid);
        timed(name, 1.0 * bytes, [&] { hipLaunchKernelGGL((write_k<4, true>), dim3(grid), dim3(256), 0, 0, (u64x2 *)(pool + bytes), n_pieces); });
    }
    timed("hipMemcpyAsync device to device", 2.0 * bytes, [&] { CK(hipMemcpyAsync(pool + bytes, pool, bytes, hipMemcpyDeviceToDevice, 0)); });

    std::sort(rows.begin(), rows.end(), [](const Row &a, const Row &b) { return a.gbps > b.gbps; });
    printf("\n---- best ten ----\n");
    for (size_t i = 0; i < rows.size() && i < 10; ++i)
        printf("%s\n", rows[i].text);
    return 0;
}
