// Memory side + synchronisation of a "team" NTT: 16 workgroups of one XCD share one polynomial (512 KB) at a time --
// each does one column tile, the team meets on a counter, each does one row tile -- so that the second pass reads what
// the first wrote a few microseconds earlier (same XCD: from that XCD's L2).  Access patterns only, no arithmetic.
// Follow-up of tools/ubench_reuse2.hip.  Persistent grid; teams are formed from the hardware XCC id at start.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_team.hip -o tools/ubench_team
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
constexpr u32 POLY = 65536;
constexpr u32 PAD = 32;         // counters on lines of their own (32 x 4 B = 128 B)
constexpr u32 MAX_TEAMS = 256;
constexpr u32 SPIN_LIMIT = 1u << 22;

struct State
{
    u32 xcd_slots[8 * PAD];
    u32 started[PAD];
    u32 error[PAD];
    u32 counter[MAX_TEAMS * PAD];
};

__device__ __forceinline__ u32 ld_relaxed(const u32 *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 ld_sc1(const u64 *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // global_load_dwordx2 ... sc1: bypasses this CU's L1
}

// one lane: wait until *p >= want, or give up (sets the error word; every spin of the launch then ends at once)
__device__ __forceinline__ bool spin_until(const u32 *p, u32 want, State *st)
{
    for (u32 i = 0; i < SPIN_LIMIT; ++i)
    {
        if (ld_relaxed(p) >= want)
            return true;
        if ((i & 63) == 63 && ld_relaxed(st->error))
            return false;
        __builtin_amdgcn_s_sleep(2);
    }
    __hip_atomic_store(st->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}

template <int MODE> // bit 0: sc1 loads in the row pass (plain = incorrect hand-off, timing only); bit 1: pipelined
__global__ __launch_bounds__(256) void team_passes(u64 *buf, u32 polys, State *st, u32 grid)
{
    __shared__ u32 sh[4];
    const u32 tid = threadIdx.x;
    if (tid == 0)
    {
        const u32 xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7; // HW_REG_XCC_ID[3:0]
        const u32 slot = atomicAdd(&st->xcd_slots[xcc * PAD], 1u);
        atomicAdd(st->started, 1u);
        bool ok = spin_until(st->started, grid, st);
        u32 rank = 0xffffffffu, nteams = 0;
        if (ok)
        {
            u32 before = 0;
            for (u32 x = 0; x < 8; ++x)
            {
                const u32 c = ld_relaxed(&st->xcd_slots[x * PAD]) / 16;
                if (x == xcc && slot / 16 < c)
                    rank = before + slot / 16;
                before += c;
            }
            nteams = before;
        }
        sh[0] = rank;
        sh[1] = slot & 15;
        sh[2] = nteams;
    }
    __syncthreads();
    const u32 rank = sh[0], member = sh[1], nteams = sh[2];
    if (rank == 0xffffffffu || nteams == 0)
        return; // a workgroup whose team did not fill (or a failed start): nothing to do
    const u32 c = tid & 15, tau = tid >> 4;
    auto cols = [&](u32 pi) {
        u64 *p = buf + (size_t)pi * POLY + member * 16;
        u64 x[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) x[e] = __builtin_nontemporal_load(p + (size_t)(e * 16 + tau) * 256 + c);
#pragma unroll
        for (int e = 0; e < 16; ++e) p[(size_t)(e * 16 + tau) * 256 + c] = x[e] + 1;
        // every wave's stores have reached L2, then one lane tells the team
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0)
            atomicAdd(&st->counter[rank * PAD], 1u);
    };
    auto rows = [&](u32 pi) {
        u64 *p = buf + (size_t)pi * POLY + (size_t)member * 4096;
        u64 x[16];
#pragma unroll
        for (int h = 0; h < 16; ++h) x[h] = (MODE & 1) ? ld_sc1(p + h * 256 + tid) : p[h * 256 + tid];
#pragma unroll
        for (int h = 0; h < 16; ++h) __builtin_nontemporal_store(x[h] + 1, p + h * 256 + tid);
    };
    auto wait = [&](u32 it) {
        if (tid == 0)
            sh[3] = spin_until(&st->counter[rank * PAD], 16 * (it + 1), st) ? 1u : 0u;
        __syncthreads();
        return sh[3] != 0;
    };
    u32 it = 0;
    if (MODE & 2)
    {
        // pipelined: the next polynomial's column tile is done before this one's meeting point is looked at
        if (rank < polys)
            cols(rank);
        for (u32 pi = rank; pi < polys; pi += nteams, ++it)
        {
            if (pi + nteams < polys)
                cols(pi + nteams);
            if (!wait(it))
                return;
            rows(pi);
        }
    }
    else
        for (u32 pi = rank; pi < polys; pi += nteams, ++it)
        {
            cols(pi);
            if (!wait(it))
                return;
            rows(pi);
        }
}

int main(int argc, char **argv)
{
    const size_t polys = argc > 1 ? (size_t)atol(argv[1]) : 16384;
    const size_t bytes = polys * POLY * sizeof(u64);
    u64 *buf = nullptr;
    State *st = nullptr;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&st, sizeof(State)));
    CK(hipMemset(buf, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int launches = 0;
    auto run = [&](const char *what, auto kern, u32 grid) {
        float best = 1e30f;
        u32 teams = 0, err = 0;
        for (int r = 0; r < 5; ++r)
        {
            CK(hipMemsetAsync(st, 0, sizeof(State), 0));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, buf, (u32)polys, st, grid);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < best) best = ms;
            State h;
            CK(hipMemcpy(&h, st, sizeof(State), hipMemcpyDeviceToHost));
            teams = 0;
            for (int x = 0; x < 8; ++x) teams += h.xcd_slots[x * PAD] / 16;
            err |= h.error[0];
            ++launches;
            if (r == 0)
            {
                printf("   slots per XCD:");
                for (int x = 0; x < 8; ++x) printf(" %u", h.xcd_slots[x * PAD]);
                printf("\n");
            }
        }
        CK(hipGetLastError());
        printf("%-44s grid %4u teams %3u err %u  %8.3f ms   %7.1f GB/s algorithmic\n", what, grid, teams, err, best, 2.0 * bytes / best / 1e6);
        fflush(stdout);
    };
    for (u32 grid : { 512u, 768u, 1024u, 1536u, 2048u })
    {
        run("teams of 16, sc1 loads in the row pass", team_passes<1>, grid);
        run("teams of 16, sc1 loads, pipelined", team_passes<3>, grid);
        run("teams of 16, plain loads, pipelined (timing)", team_passes<2>, grid);
    }
    u64 h[2];
    CK(hipMemcpy(h, buf, 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h + 1, buf + polys * POLY - 1, 8, hipMemcpyDeviceToHost));
    printf("check: first word %llu, last word %llu (expected %d)\n", h[0], h[1], 2 * launches);
    return 0;
}
