// VERDICT r02 item 1(b), measured at the level of the memory skeleton: the two NTT passes as ONE launch in which a ticket
// counter hands resident workgroups the tasks  pass1(c0), pass1(c1), pass2(c0), pass1(c2), pass2(c1) ...  over chunks of
// one limb x PB polynomials (PB x 512 KB), a pass-2 task waiting for its chunk's pass-1 completion counter (agent-scope
// release on the producer side, acquire poll on the consumer side, workgroup barrier before the loads), in place, so that
// pass 2 finds its chunk in the Infinity Cache.  No arithmetic: pass 1 is the column tile of ntt_fwd_cols (256 rows x
// 256-byte segments, 512 threads), pass 2 a 16-row tile (32 KB contiguous, 512 threads); every word gets +1 per pass, so
// the result (+2 everywhere) checks the hand-off.  Against it: the same two task kernels as two plain launches.
// Every spin is bounded (a stuck wait sets an error word and carries on), so a bug cannot hang the card.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_pipe.hip -o tools/ubench_pipe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr u32 LIMBS = 32;
constexpr size_t N = 65536;
constexpr u32 T1_PER_LP = 8;  // column tiles (32 columns) per limb-polynomial
constexpr u32 T2_PER_LP = 16; // row tiles (16 rows) per limb-polynomial

// agent-scope accesses that go around the XCD's L2 instead of flushing / invalidating all of it (the guide's hand-off form:
// write-through `sc1` stores, drained with vmcnt(0) before the count; `sc1` loads after the poll).  The compiler does not
// track inline-asm memory operations: the callers wait themselves.
__device__ __forceinline__ void st8_sc1(u64 *p, u64 v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
// four loads and their wait as ONE statement, so that the compiler cannot use the registers before the data has landed
__device__ __forceinline__ void ld16x4_sc1(u64x2 (&v)[4], const u64x2 *p0, const u64x2 *p1, const u64x2 *p2, const u64x2 *p3)
{
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                 "global_load_dwordx4 %1, %5, off sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc1\n\t"
                 "global_load_dwordx4 %3, %7, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
                 : "memory");
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// word offset of limb-polynomial (limb, poly)
__device__ __forceinline__ size_t lp_off(u32 limb, u32 poly) { return ((size_t)poly * LIMBS + limb) * N; }

template <bool NT, bool SC1 = false>
__device__ __forceinline__ void pass1_task(u64 *buf, u32 limb, u32 poly, u32 tile)
{
    u64 *p = buf + lp_off(limb, poly) + tile * 32;
    const u32 c = threadIdx.x & 31, tau = threadIdx.x >> 5; // 16 row groups
    u64 x[16];
#pragma unroll
    for (int e = 0; e < 16; ++e)
        x[e] = NT ? __builtin_nontemporal_load(p + (size_t)(e * 16 + tau) * 256 + c) : p[(size_t)(e * 16 + tau) * 256 + c];
#pragma unroll
    for (int e = 0; e < 16; ++e)
    {
        if (SC1)
            st8_sc1(p + (size_t)(e * 16 + tau) * 256 + c, x[e] + 1);
        else if (NT)
            __builtin_nontemporal_store(x[e] + 1, p + (size_t)(e * 16 + tau) * 256 + c);
        else
            p[(size_t)(e * 16 + tau) * 256 + c] = x[e] + 1;
    }
    if (SC1)
        wait_vm0(); // the write-through stores have reached memory
}

template <bool NT, bool SC1 = false>
__device__ __forceinline__ void pass2_task(u64 *buf, u32 limb, u32 poly, u32 tile)
{
    u64x2 *p = reinterpret_cast<u64x2 *>(buf + lp_off(limb, poly) + (size_t)tile * 16 * 256) + threadIdx.x;
    u64x2 x[4];
    if (SC1)
        ld16x4_sc1(x, p, p + 512, p + 1024, p + 1536);
    else
#pragma unroll
    for (int u = 0; u < 4; ++u)
        x[u] = NT ? __builtin_nontemporal_load(p + u * 512) : p[u * 512];
#pragma unroll
    for (int u = 0; u < 4; ++u)
    {
        x[u].x += 1;
        x[u].y += 1;
        if (NT)
            __builtin_nontemporal_store(x[u], p + u * 512);
        else
            p[u * 512] = x[u];
    }
}

// plain launches: grid = all tasks of one pass, (tile, poly, limb) with limb slowest = chunk order
template <bool NT>
__global__ __launch_bounds__(512) void pass1_all(u64 *buf, u32 polys)
{
    const u32 tile = blockIdx.x % T1_PER_LP, rest = blockIdx.x / T1_PER_LP;
    pass1_task<NT>(buf, rest / polys, rest % polys, tile);
}
template <bool NT>
__global__ __launch_bounds__(512) void pass2_all(u64 *buf, u32 polys)
{
    const u32 tile = blockIdx.x % T2_PER_LP, rest = blockIdx.x / T2_PER_LP;
    pass2_task<NT>(buf, rest / polys, rest % polys, tile);
}

struct Ctl
{
    u32 ticket;
    u32 error;
    u32 max_spins;
    u32 pad;
};

// one launch: ticket -> slot s of the sequence  P1(0), [P1(c+1), P2(c)] for c = 0 .. C-2, P2(C-1); chunk c = (limb, block of PB polys)
template <bool NT, bool SC1 = false>
__global__ __launch_bounds__(512) void pipelined(u64 *buf, u32 polys, u32 PB, u32 n_chunks, Ctl *ctl, u32 *done)
{
    __shared__ u32 s_ticket;
    if (threadIdx.x == 0)
        s_ticket = atomicAdd(&ctl->ticket, 1u);
    __syncthreads();
    const u32 t = s_ticket;
    const u32 t1 = PB * T1_PER_LP, t2 = PB * T2_PER_LP; // tasks of a chunk per pass
    // slots: [0, t1) = P1(0); then groups of (t1 + t2): P1(c + 1) followed by P2(c), for c = 0 .. C - 2; then P2(C - 1)
    u32 phase, chunk, local;
    if (t < t1)
        phase = 1, chunk = 0, local = t;
    else
    {
        const u32 r = t - t1, g = r / (t1 + t2), w = r % (t1 + t2);
        if (g < n_chunks - 1)
        {
            if (w < t1)
                phase = 1, chunk = g + 1, local = w;
            else
                phase = 2, chunk = g, local = w - t1;
        }
        else
        {
            // the tail: P2 of the last chunk (its slots follow the last full group)
            const u32 rr = r - (n_chunks - 1) * (t1 + t2);
            phase = 2, chunk = n_chunks - 1, local = rr;
            if (rr >= t2)
                return;
        }
    }
    const u32 blocks_per_limb = polys / PB;
    const u32 limb = chunk / blocks_per_limb, poly0 = (chunk % blocks_per_limb) * PB;
    if (phase == 1)
    {
        pass1_task<NT, SC1>(buf, limb, poly0 + local / T1_PER_LP, local % T1_PER_LP);
        if (!SC1)
            __threadfence(); // agent-scope release: this workgroup's stores are visible device-wide before the count
        __syncthreads();
        if (threadIdx.x == 0)
            atomicAdd(&done[chunk], 1u);
    }
    else
    {
        if (threadIdx.x == 0)
        {
            u32 spins = 0;
            while (__hip_atomic_load(&done[chunk], SC1 ? __ATOMIC_RELAXED : __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < t1)
            {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > ctl->max_spins)
                {
                    atomicAdd(&ctl->error, 1u);
                    break;
                }
            }
        }
        __syncthreads();
        if (!SC1)
            __atomic_thread_fence(__ATOMIC_ACQUIRE); // every wave of the workgroup acquires before its loads
        pass2_task<NT, SC1>(buf, limb, poly0 + local / T2_PER_LP, local % T2_PER_LP);
    }
}

int main(int argc, char **argv)
{
    const u32 polys = argc > 1 ? (u32)atoi(argv[1]) : 256;
    const size_t words = (size_t)polys * LIMBS * N, bytes = words * 8;
    u64 *buf;
    CK(hipMalloc(&buf, bytes));
    Ctl *ctl;
    u32 *done;
    CK(hipMalloc(&ctl, sizeof(Ctl)));
    CK(hipMalloc(&done, sizeof(u32) * LIMBS * polys));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<u64> probe(1 << 20);
    auto check = [&](u64 expect) {
        size_t bad = 0;
        for (size_t o : { (size_t)0, words / 3, words - probe.size() })
        {
            CK(hipMemcpy(probe.data(), buf + o, probe.size() * 8, hipMemcpyDeviceToHost));
            for (u64 v : probe)
                bad += v != expect;
        }
        return bad;
    };
    printf("two NTT passes, access only: %u polys x %u limbs x 65536 words = %.2f GB, 512-thread workgroups\n", polys, LIMBS, bytes / 1e9);
    for (int mode = 0; mode < 3; ++mode) // 0 default policy + fences, 1 nt + fences, 2 nt with sc1 hand-off accesses, no fences
    {
        const int nt = mode >= 1;
        const char *label = mode == 0 ? "default" : mode == 1 ? "nt" : "nt+sc1";
        // baseline: two plain launches
        CK(hipMemset(buf, 0, bytes));
        float best = 1e30f, best1 = 1e30f, best2 = 1e30f;
        for (int rep = 0; rep < 4; ++rep)
        {
            float a, b;
            hipEvent_t m;
            CK(hipEventCreate(&m));
            CK(hipEventRecord(e0));
            if (nt)
                hipLaunchKernelGGL(pass1_all<true>, dim3(polys * LIMBS * T1_PER_LP), dim3(512), 0, 0, buf, polys);
            else
                hipLaunchKernelGGL(pass1_all<false>, dim3(polys * LIMBS * T1_PER_LP), dim3(512), 0, 0, buf, polys);
            CK(hipEventRecord(m));
            if (nt)
                hipLaunchKernelGGL(pass2_all<true>, dim3(polys * LIMBS * T2_PER_LP), dim3(512), 0, 0, buf, polys);
            else
                hipLaunchKernelGGL(pass2_all<false>, dim3(polys * LIMBS * T2_PER_LP), dim3(512), 0, 0, buf, polys);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&a, e0, m));
            CK(hipEventElapsedTime(&b, m, e1));
            if (rep && a + b < best)
                best = a + b, best1 = a, best2 = b;
            CK(hipEventDestroy(m));
        }
        CK(hipGetLastError());
        printf("%-8s two launches:                         %7.3f ms (pass 1 %6.3f + pass 2 %6.3f)  %7.1f GB/s read + written  [%zu bad words]\n",
               label, best, best1, best2, 4.0 * bytes / best / 1e6, check(8));
        for (u32 PB : { 16u, 32u, 64u, 128u, 256u })
        {
            if (PB > polys || polys % PB)
                continue;
            const u32 n_chunks = LIMBS * (polys / PB);
            const u32 total = n_chunks * PB * (T1_PER_LP + T2_PER_LP);
            CK(hipMemset(buf, 0, bytes));
            float bestp = 1e30f;
            u32 errors = 0;
            const int reps = 4;
            for (int rep = 0; rep < reps; ++rep)
            {
                Ctl h = { 0, 0, 1u << 16, 0 };
                CK(hipMemcpy(ctl, &h, sizeof h, hipMemcpyHostToDevice));
                CK(hipMemset(done, 0, sizeof(u32) * n_chunks));
                CK(hipEventRecord(e0));
                if (mode == 2)
                    hipLaunchKernelGGL((pipelined<true, true>), dim3(total), dim3(512), 0, 0, buf, polys, PB, n_chunks, ctl, done);
                else if (nt)
                    hipLaunchKernelGGL(pipelined<true>, dim3(total), dim3(512), 0, 0, buf, polys, PB, n_chunks, ctl, done);
                else
                    hipLaunchKernelGGL(pipelined<false>, dim3(total), dim3(512), 0, 0, buf, polys, PB, n_chunks, ctl, done);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < bestp)
                    bestp = ms;
                CK(hipMemcpy(&h, ctl, sizeof h, hipMemcpyDeviceToHost));
                errors += h.error;
            }
            CK(hipGetLastError());
            printf("%-8s one launch, chunks of %3u polys (%4u MB): %7.3f ms                                  %7.1f GB/s read + written  [%zu bad words, %u waits timed out]\n",
                   label, PB, PB / 2, bestp, 4.0 * bytes / bestp / 1e6, check(2 * reps), errors);
            fflush(stdout);
        }
    }
    return 0;
}
