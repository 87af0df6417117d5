// Access-only skeletons of the two NTT passes at N = 2^16 (no arithmetic), swept over the things ubench_copy_sweep
// showed to matter: bytes a thread has in flight, persistence of the workgroups, cache policy, in place against out of
// place.  Buffer = polys x limbs x 65536 words like BASELINE config 2 (fewer polys).  Rates are bytes read + written.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_pass_shapes.hip -o tools/ubench_pass_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned long long u64;
typedef unsigned int u32;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr u32 LIMBS = 32;
constexpr size_t N = 65536;

template <bool NT>
__device__ __forceinline__ u64x2 ld(const u64x2 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT>
__device__ __forceinline__ void st(u64x2 v, u64x2 *p)
{
    if (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}
template <bool NT>
__device__ __forceinline__ u64 ld8(const u64 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT>
__device__ __forceinline__ void st8(u64 v, u64 *p)
{
    if (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

// ---- row pass, one trip per workgroup: R rows (R * 2 KB contiguous) of one limb-polynomial, 16 B per lane ---------
// block -> (tile, poly, limb), tile fastest: the chip works on a compact window
// order 0: (tile, poly, limb) as the library's kernels (limb slowest: one prime's twiddles at a time);
// order 1: (tile, limb, poly): the chip walks the buffer front to back.  pad: extra words between polynomials' objects.
template <int R, int THREADS, bool NT>
__global__ __launch_bounds__(THREADS) void rows_once(const u64 *src, u64 *dst, u32 polys, int order = 0, size_t pad = 0)
{
    constexpr int U = R * 128 / THREADS; // 16-byte pieces per thread
    const u32 tiles = 256 / R;
    const u32 tile = blockIdx.x % tiles, rest = blockIdx.x / tiles;
    const u32 poly = order ? rest / LIMBS : rest % polys, limb = order ? rest % LIMBS : rest / polys;
    const size_t off = (size_t)poly * (LIMBS * N + pad) + limb * N + (size_t)tile * R * 256;
    const u64x2 *s = reinterpret_cast<const u64x2 *>(src + off) + threadIdx.x;
    u64x2 *d = reinterpret_cast<u64x2 *>(dst + off) + threadIdx.x;
    u64x2 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
        x[u] = ld<NT>(s + u * THREADS);
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
        x[u].x += 1;
        st<NT>(x[u], d + u * THREADS);
    }
}

// ---- row pass, streaming form (the shape of ntt_fwd_rows8): a workgroup stays on R rows of one limb and walks over
// the polynomials, ND polynomials in flight per thread, the next ND requested before the current ones are stored
template <int R, int ND, bool PREFETCH, bool NT, int WPS>
__global__ __launch_bounds__(256, WPS) void rows_stream(const u64 *src, u64 *dst, u32 polys, u32 chunks, size_t pad = 0)
{
    const size_t PS = LIMBS * N + pad;
    constexpr int U = R * 128 / 256;
    const u32 tiles = 256 / R;
    const u32 chunk = blockIdx.x % chunks, rest = blockIdx.x / chunks;
    const u32 tile = rest % tiles, limb = rest / tiles;
    const size_t base = (size_t)limb * N + (size_t)tile * R * 256;
    const u32 groups = polys / ND;
    u64x2 xn[ND][U];
    auto fetch = [&](u64x2 (&x)[ND][U], u32 g) {
#pragma unroll
        for (int n = 0; n < ND; ++n)
        {
            const u64x2 *s = reinterpret_cast<const u64x2 *>(src + base + (size_t)(g * ND + n) * PS) + threadIdx.x;
#pragma unroll
            for (int u = 0; u < U; ++u)
                x[n][u] = ld<NT>(s + u * 256);
        }
    };
    u32 g = chunk;
    if (PREFETCH && g < groups)
        fetch(xn, g);
    for (; g < groups; g += chunks)
    {
        u64x2 x[ND][U];
        if (PREFETCH)
        {
#pragma unroll
            for (int n = 0; n < ND; ++n)
#pragma unroll
                for (int u = 0; u < U; ++u)
                    x[n][u] = xn[n][u];
            if (g + chunks < groups)
                fetch(xn, g + chunks);
        }
        else
            fetch(x, g);
#pragma unroll
        for (int n = 0; n < ND; ++n)
        {
            u64x2 *d = reinterpret_cast<u64x2 *>(dst + base + (size_t)(g * ND + n) * PS) + threadIdx.x;
#pragma unroll
            for (int u = 0; u < U; ++u)
            {
                x[n][u].x += 1;
                st<NT>(x[n][u], d + u * 256);
            }
        }
    }
}

// ---- column pass: a workgroup owns C columns x 256 rows of one limb-polynomial ---------------------------------------
// B8: 8 bytes per lane (C lanes per row segment, as ntt_fwd_cols); otherwise 16 bytes per lane (C / 2 lanes per segment)
template <int C, int THREADS, bool B8, bool NT>
__global__ __launch_bounds__(THREADS) void cols_once(const u64 *src, u64 *dst, u32 polys, int order = 0, size_t pad = 0)
{
    const u32 tiles = 256 / C;
    const u32 tile = blockIdx.x % tiles, rest = blockIdx.x / tiles;
    const u32 poly = order ? rest / LIMBS : rest % polys, limb = order ? rest % LIMBS : rest / polys;
    const size_t off = (size_t)poly * (LIMBS * N + pad) + limb * N + (size_t)tile * C;
    if (B8)
    {
        constexpr int RPT = 256 * C / THREADS; // rows per thread
        const u32 c = threadIdx.x % C, tau = threadIdx.x / C;
        u64 x[RPT];
#pragma unroll
        for (int e = 0; e < RPT; ++e)
            x[e] = ld8<NT>(src + off + (size_t)(e * (THREADS / C) + tau) * 256 + c);
#pragma unroll
        for (int e = 0; e < RPT; ++e)
            st8<NT>(x[e] + 1, dst + off + (size_t)(e * (THREADS / C) + tau) * 256 + c);
    }
    else
    {
        constexpr int LPS = C / 2, RPT = 256 * LPS / THREADS;
        const u32 c = threadIdx.x % LPS, tau = threadIdx.x / LPS;
        u64x2 x[RPT];
#pragma unroll
        for (int e = 0; e < RPT; ++e)
            x[e] = ld<NT>(reinterpret_cast<const u64x2 *>(src + off + (size_t)(e * (THREADS / LPS) + tau) * 256) + c);
#pragma unroll
        for (int e = 0; e < RPT; ++e)
        {
            x[e].x += 1;
            st<NT>(x[e], reinterpret_cast<u64x2 *>(dst + off + (size_t)(e * (THREADS / LPS) + tau) * 256) + c);
        }
    }
}

// ---- the 128 x 512 split of N = 65536: a column tile of C columns x ROWS rows with rows RL words long -----------------
// (ROWS x RL = N; 7 column stages + 9 row stages instead of 8 + 8 would give 512-byte segments to a 512-thread workgroup)
template <int C, int ROWS, int RL, int THREADS>
__global__ __launch_bounds__(THREADS) void cols_shape(u64 *buf, u32 polys)
{
    const u32 tiles = RL / C;
    const u32 tile = blockIdx.x % tiles, rest = blockIdx.x / tiles;
    const u32 poly = rest / LIMBS, limb = rest % LIMBS; // (tile, limb, poly)
    const size_t off = (size_t)poly * (LIMBS * N) + limb * N + (size_t)tile * C;
    constexpr int RPT = ROWS * C / THREADS;
    const u32 c = threadIdx.x % C, tau = threadIdx.x / C;
    u64 x[RPT];
#pragma unroll
    for (int e = 0; e < RPT; ++e)
        x[e] = ld8<true>(buf + off + (size_t)(e * (THREADS / C) + tau) * RL + c);
#pragma unroll
    for (int e = 0; e < RPT; ++e)
        st8<true>(x[e] + 1, buf + off + (size_t)(e * (THREADS / C) + tau) * RL + c);
}
// rows RL words long, R of them per workgroup of 256 threads, 16 B per lane, one trip
template <int R, int RL>
__global__ __launch_bounds__(256) void rows_shape(u64 *buf, u32 polys)
{
    constexpr int U = R * RL / 2 / 256;
    u64x2 *p = reinterpret_cast<u64x2 *>(buf + (size_t)blockIdx.x * R * RL) + threadIdx.x;
    u64x2 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
        x[u] = ld<true>(p + u * 256);
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
        x[u].x += 1;
        st<true>(x[u], p + u * 256);
    }
}

struct Row
{
    double gbps;
    char text[176];
};

int main(int argc, char **argv)
{
    const u32 polys = argc > 1 ? (u32)atoi(argv[1]) : 256;
    const size_t words = (size_t)polys * LIMBS * N, bytes = words * 8;
    u64 *a, *b;
    const size_t PAD = 8192 + 512; // words: 68 KB between objects breaks the 16-MiB stride
    const size_t alloc = bytes + (size_t)polys * PAD * 8;
    CK(hipMalloc(&a, alloc));
    CK(hipMalloc(&b, alloc));
    CK(hipMemset(a, 1, alloc));
    CK(hipMemset(b, 1, alloc));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<Row> rows;
    auto timed = [&](const char *name, auto launch) {
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
        r.gbps = 2.0 * bytes / best / 1e6;
        snprintf(r.text, sizeof r.text, "%-104s %8.3f ms %8.1f GB/s", name, best, r.gbps);
        printf("%s\n", r.text);
        fflush(stdout);
        rows.push_back(r);
    };
    char name[176];
    const u32 lp = polys * LIMBS;

#define ROWS_ONCE(R, T, NT)                                                                                            \
    for (int oop = 0; oop < 2; ++oop)                                                                                  \
    {                                                                                                                  \
        snprintf(name, sizeof name, "rows, one trip: %2d rows per workgroup of %4d threads (%d x 16 B per lane) %s %s", R, T, \
                 R * 128 / T, NT ? "nt" : "default", oop ? "out of place" : "in place");                               \
        timed(name, [&] { hipLaunchKernelGGL((rows_once<R, T, NT>), dim3(lp * (256 / R)), dim3(T), 0, 0, a, oop ? b : a, polys); }); \
    }
    ROWS_ONCE(2, 256, false)
    ROWS_ONCE(2, 256, true)
    ROWS_ONCE(4, 256, false)
    ROWS_ONCE(4, 256, true)
    ROWS_ONCE(8, 256, false)
    ROWS_ONCE(8, 256, true)
    ROWS_ONCE(16, 256, false)
    ROWS_ONCE(16, 256, true)
    ROWS_ONCE(8, 512, true)
    ROWS_ONCE(16, 512, true)
    ROWS_ONCE(16, 1024, true)
    ROWS_ONCE(32, 1024, true)

#define ROWS_STREAM(R, ND, PF, NT, WPS, chunks)                                                                        \
    for (int oop = 0; oop < 2; ++oop)                                                                                  \
    {                                                                                                                  \
        snprintf(name, sizeof name, "rows, streaming: %d rows, %d polys in flight%s, %s, %d waves per SIMD, %d workgroups, %s", R, ND, \
                 PF ? " + next requested" : "", NT ? "nt" : "default", WPS, (int)(LIMBS * (256 / R) * chunks),       \
                 oop ? "out of place" : "in place");                                                                   \
        timed(name, [&] { hipLaunchKernelGGL((rows_stream<R, ND, PF, NT, WPS>), dim3(LIMBS * (256 / R) * chunks), dim3(256), 0, 0, a, oop ? b : a, polys, chunks); }); \
    }
    ROWS_STREAM(8, 2, true, true, 2, 8)   // the shape of ntt_fwd_rows8
    ROWS_STREAM(8, 2, true, false, 2, 8)
    ROWS_STREAM(8, 2, false, true, 2, 8)
    ROWS_STREAM(8, 1, true, true, 2, 8)
    ROWS_STREAM(8, 1, true, true, 4, 16)
    ROWS_STREAM(8, 1, false, true, 4, 16)
    ROWS_STREAM(8, 1, false, true, 8, 32)
    ROWS_STREAM(8, 2, true, true, 4, 16)
    ROWS_STREAM(4, 2, true, true, 4, 8)
    ROWS_STREAM(4, 1, false, true, 8, 16)
    ROWS_STREAM(2, 1, false, true, 8, 8)
    ROWS_STREAM(2, 1, false, false, 8, 8)
    ROWS_STREAM(16, 1, true, true, 2, 16)

#define COLS_ONCE(C, T, B8, NT)                                                                                        \
    for (int oop = 0; oop < 2; ++oop)                                                                                  \
    {                                                                                                                  \
        snprintf(name, sizeof name, "cols, one trip: %2d columns (%3d-B segments) x 256 rows, %4d threads, %2d B per lane, %s %s", C, C * 8, T, \
                 B8 ? 8 : 16, NT ? "nt" : "default", oop ? "out of place" : "in place");                               \
        timed(name, [&] { hipLaunchKernelGGL((cols_once<C, T, B8, NT>), dim3(lp * (256 / C)), dim3(T), 0, 0, a, oop ? b : a, polys); }); \
    }
    COLS_ONCE(16, 256, true, false)  // ntt_fwd_cols, 4096-word tiles
    COLS_ONCE(16, 256, true, true)
    COLS_ONCE(32, 512, true, false)  // ntt_fwd_cols, 8192-word tiles
    COLS_ONCE(32, 512, true, true)
    COLS_ONCE(64, 1024, true, false)
    COLS_ONCE(64, 1024, true, true)
    COLS_ONCE(16, 256, false, true)
    COLS_ONCE(32, 256, false, true)
    COLS_ONCE(32, 512, false, false)
    COLS_ONCE(32, 512, false, true)
    COLS_ONCE(64, 512, false, true)
    COLS_ONCE(64, 1024, false, false)
    COLS_ONCE(64, 1024, false, true)
    COLS_ONCE(128, 1024, false, true)
    COLS_ONCE(8, 256, true, true)
    COLS_ONCE(8, 128, true, true)

#define COLS_SHAPE(C, ROWS, RL, T)                                                                                     \
    snprintf(name, sizeof name, "cols, one trip: %2d columns (%3d-B segments) x %3d rows of %3d words, %4d threads, nt in place, (tile, limb, poly)", C, C * 8, ROWS, RL, T); \
    timed(name, [&] { hipLaunchKernelGGL((cols_shape<C, ROWS, RL, T>), dim3(lp * (RL / C)), dim3(T), 0, 0, a, polys); });
    COLS_SHAPE(32, 256, 256, 512)  // today
    COLS_SHAPE(64, 256, 256, 1024)
    COLS_SHAPE(64, 256, 256, 512)  // 32 rows per thread
    COLS_SHAPE(64, 128, 512, 512)  // the 7 + 9 split
    COLS_SHAPE(32, 128, 512, 256)
    COLS_SHAPE(128, 128, 512, 1024)
    COLS_SHAPE(64, 64, 1024, 256)  // 6 + 10
    COLS_SHAPE(128, 64, 1024, 512)
    timed("rows, one trip: 4 rows of 512 words per workgroup of 256 threads (4 x 16 B per lane), nt in place", [&] { hipLaunchKernelGGL((rows_shape<4, 512>), dim3(lp * 32), dim3(256), 0, 0, a, polys); });
    timed("rows, one trip: 8 rows of 256 words per workgroup of 256 threads (4 x 16 B per lane), nt in place", [&] { hipLaunchKernelGGL((rows_shape<8, 256>), dim3(lp * 32), dim3(256), 0, 0, a, polys); });

    // ---- order of the workgroups over the buffer and the stride between objects ------------------------------------
    if (argc <= 2)
    {
#define ROWS_ORDER(R, T)                                                                                               \
    for (int v = 0; v < 4; ++v)                                                                                        \
    {                                                                                                                  \
        const int order = v & 1;                                                                                       \
        const size_t pad = (v & 2) ? PAD : 0;                                                                          \
        snprintf(name, sizeof name, "rows, one trip: %2d rows, %4d threads, nt in place, order %s, object stride 16 MiB%s", R, T, \
                 order ? "(tile, limb, poly)" : "(tile, poly, limb)", pad ? " + 68 KiB" : "");                         \
        timed(name, [&] { hipLaunchKernelGGL((rows_once<R, T, true>), dim3(lp * (256 / R)), dim3(T), 0, 0, a, a, polys, order, pad); }); \
    }
        ROWS_ORDER(2, 256)
        ROWS_ORDER(8, 256)
        ROWS_ORDER(16, 256)
        ROWS_ORDER(32, 1024)
#define COLS_ORDER(C, T)                                                                                               \
    for (int v = 0; v < 4; ++v)                                                                                        \
    {                                                                                                                  \
        const int order = v & 1;                                                                                       \
        const size_t pad = (v & 2) ? PAD : 0;                                                                          \
        snprintf(name, sizeof name, "cols, one trip: %2d columns, %4d threads, 8 B per lane, nt in place, order %s, object stride 16 MiB%s", C, T, \
                 order ? "(tile, limb, poly)" : "(tile, poly, limb)", pad ? " + 68 KiB" : "");                         \
        timed(name, [&] { hipLaunchKernelGGL((cols_once<C, T, true, true>), dim3(lp * (256 / C)), dim3(T), 0, 0, a, a, polys, order, pad); }); \
    }
        COLS_ORDER(16, 256)
        COLS_ORDER(32, 512)
        COLS_ORDER(64, 1024)
        for (int v = 0; v < 2; ++v)
        {
            const size_t pad = v ? PAD : 0;
            for (u32 chunks : {1u, 2u, 8u})
            {
                snprintf(name, sizeof name, "rows, streaming: 8 rows, 2 polys in flight + next requested, nt, in place, %u workgroups, object stride 16 MiB%s",
                         LIMBS * 32 * chunks, pad ? " + 68 KiB" : "");
                timed(name, [&] { hipLaunchKernelGGL((rows_stream<8, 2, true, true, 2>), dim3(LIMBS * 32 * chunks), dim3(256), 0, 0, a, a, polys, chunks, pad); });
                snprintf(name, sizeof name, "rows, streaming: 8 rows, 1 poly in flight + next requested, nt, in place, %u workgroups, object stride 16 MiB%s",
                         LIMBS * 32 * chunks, pad ? " + 68 KiB" : "");
                timed(name, [&] { hipLaunchKernelGGL((rows_stream<8, 1, true, true, 2>), dim3(LIMBS * 32 * chunks), dim3(256), 0, 0, a, a, polys, chunks, pad); });
            }
        }
    }

    std::sort(rows.begin(), rows.end(), [](const Row &x, const Row &y) { return x.gbps > y.gbps; });
    printf("\n---- best twelve ----\n");
    for (size_t i = 0; i < rows.size() && i < 12; ++i)
        printf("%s\n", rows[i].text);
    return 0;
}
