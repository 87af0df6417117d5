// Wave-shuffle A/B.  `north_star` names "wavefront shuffles for the final butterflies"; the library's
// row passes exchange through per-wave LDS instead (csrc/rows8.cuh).  This side kernel is the poly-streaming forward row
// pass (the shape of ntt_fwd_rows8: 8 rows per workgroup, twiddles staged in LDS, two polynomials in flight, next pair
// requested early) with rm_rows' LAST exchange -- the one in front of the final two stages, which stays inside groups of
// four lanes -- done either through LDS (as the library does) or with DPP quad permutes and selects, no LDS at all.
// Both variants run on a config-2 shaped buffer (polys x 32 limbs x 65536) with the same (random) twiddle words, their
// outputs are compared word for word, and both are timed.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude tools/ubench_shuffle.hip -o tools/ubench_shuffle
#include "../cat_amd/csrc/rows8.cuh"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

namespace
{
    // value of `v` in lane (lane ^ 1) / (lane ^ 2) of the same quad
    __device__ __forceinline__ u32 quad_xor1(u32 v) { return (u32)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); } // quad_perm [1,0,3,2]
    __device__ __forceinline__ u32 quad_xor2(u32 v) { return (u32)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true); } // quad_perm [2,3,0,1]
    template <int STEP>
    __device__ __forceinline__ u64 quad_xor64(u64 v)
    {
        const u32 lo = STEP == 1 ? quad_xor1((u32)v) : quad_xor2((u32)v);
        const u32 hi = STEP == 1 ? quad_xor1((u32)(v >> 32)) : quad_xor2((u32)(v >> 32));
        return ((u64)hi << 32) | lo;
    }

    // rm_rows' second exchange without LDS.  In: lane t = 4 A + c holds x[b] = row[A*32 + 4 b + c]; out: x[u] = row[8 t + u]
    // = row[A*32 + 8 c + u].  Inside a quad this is a 4 x 4 transpose between the lane index and register bits (b2 b1),
    // register bit b0 riding along: two butterfly steps (partner lane ^ 1 swaps register bit b1, partner lane ^ 2 swaps b2),
    // each a DPP move and two selects per 64-bit value moved, then a static renaming of the registers.
    template <int ND>
    __device__ __forceinline__ void exchange2_dpp(u64 (&x)[ND][8], u32 t)
    {
        const bool c0 = t & 1, c1 = t & 2;
#pragma unroll
        for (int n = 0; n < ND; ++n)
        {
            // step 1: registers differing in b1 (index bit 1)
#pragma unroll
            for (int r = 0; r < 8; ++r)
            {
                if (r & 2)
                    continue;
                const u64 r0 = x[n][r], r1 = x[n][r | 2];
                const u64 from0 = quad_xor64<1>(r0), from1 = quad_xor64<1>(r1);
                x[n][r] = c0 ? from1 : r0;
                x[n][r | 2] = c0 ? r1 : from0;
            }
            // step 2: registers differing in b2 (index bit 2)
#pragma unroll
            for (int r = 0; r < 8; ++r)
            {
                if (r & 4)
                    continue;
                const u64 r0 = x[n][r], r1 = x[n][r | 4];
                const u64 from0 = quad_xor64<2>(r0), from1 = quad_xor64<2>(r1);
                x[n][r] = c1 ? from1 : r0;
                x[n][r | 4] = c1 ? r1 : from0;
            }
            // now register (b2' b1' b0) of lane c holds old lane (b2' b1') register (c1 c0 b0); reader register u wants old
            // lane (u1 u0) register (c1 c0 u2): transposed register index ((u & 3) << 1) | (u >> 2)
            u64 y[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                y[u] = x[n][((u & 3) << 1) | (u >> 2)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                x[n][u] = y[u];
        }
    }

    // rm_rows (rows8.cuh) with the second exchange selectable; WIDE ranges, estimated-quotient final reduction
    template <int ND, bool DPP>
    __device__ __forceinline__ void rows_ab(u64 (&x)[ND][8], const u64 *tw, u64 *xs, u32 t, u64 q, u32 red_r, u32 red_shift)
    {
        const u64 K = q << 2, negK = 0 - K, nq = 0 - q;
        rm_stage<ND, 1, 2>(x, tw, 1, K, negK, nq);
        rm_stage<ND, 2, 0>(x, tw, 2, K, negK, nq);
        rm_stage<ND, 4, 2>(x, tw, 4, K, negK, nq);
        const u32 A = t >> 2, c = t & 3;
#pragma unroll
        for (int n = 0; n < ND; ++n)
        {
            rm_wave_sync();
#pragma unroll
            for (int a = 0; a < 8; ++a)
                xs[(a << 5) | (((t >> 2) ^ a) << 2) | c] = x[n][a];
            rm_wave_sync();
#pragma unroll
            for (int b = 0; b < 8; ++b)
                x[n][b] = xs[(A << 5) | ((b ^ A) << 2) | c];
        }
        rm_stage<ND, 1, 0>(x, tw, 8 + A, K, negK, nq);
        rm_stage<ND, 2, 2>(x, tw, 16 + 2 * A, K, negK, nq);
        rm_stage<ND, 4, 0>(x, tw, 32 + 4 * A, K, negK, nq);
        if (DPP)
            exchange2_dpp<ND>(x, t);
        else
        {
            const u32 bh = t & 3;
#pragma unroll
            for (int n = 0; n < ND; ++n)
            {
                rm_wave_sync();
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    xs[((((b & 1) << 2) | c) << 5) | (A << 2) | ((b >> 1) ^ c)] = x[n][b];
                rm_wave_sync();
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    x[n][u] = xs[(u << 5) | (A << 2) | (bh ^ (u & 3))];
            }
        }
        rm_stage<ND, 2, 2>(x, tw, 64 + 2 * t, K, negK, nq);
        rm_stage<ND, 4, 0>(x, tw, 128 + 2 * t, K, negK, nq, 192 + 2 * t);
#pragma unroll
        for (int n = 0; n < ND; ++n)
#pragma unroll
            for (int u = 0; u < 8; ++u)
                x[n][u] = reduce_below_2q(x[n][u], red_r, red_shift, nq);
    }

    // the streaming skeleton of ntt_fwd_rows8 (ntt.hip), out of place so that both variants see the same input
    template <bool DPP>
    __global__ __launch_bounds__(256, 2) void rows8_ab(const u64 *in, u64 *out, const u64 *table, u64 q, u32 red_r, u32 red_shift,
                                                      u32 n_polys, u32 n_limbs, u32 chunks)
    {
        __shared__ __align__(16) u64 lds[RM_TW_WORDS + RM_X_WORDS];
        const u32 log_n = 16, tiles = 256 / RM_ROWS;
        const u32 chunk = blockIdx.x % chunks, rest = blockIdx.x / chunks;
        const u32 tile = rest % tiles, limb = rest / tiles;
        rm_stage_twiddles(lds, table + ((size_t)limb * 3 << log_n), log_n, tile);
        __syncthreads();
        const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const u32 rl = wave * 2 + (lane >> 5), t = lane & 31;
        const u64 *tw = lds + rl * 256;
        u64 *xs = lds + RM_TW_WORDS + rl * 256;
        const size_t off = ((size_t)limb << log_n) + (size_t)(tile * RM_ROWS + rl) * 256;
        const size_t stride = (size_t)n_limbs << log_n;
        const u32 pairs = n_polys / 2;
        u64 xn[2][8];
        auto request = [&](u32 g) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    xn[n][e] = __builtin_nontemporal_load(in + off + (size_t)(2 * g + n) * stride + t + e * 32);
        };
        u32 pi = chunk;
        if (pi < pairs)
            request(pi);
        for (; pi < pairs; pi += chunks)
        {
            u64 x[2][8];
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    x[n][e] = xn[n][e];
            if (pi + chunks < pairs)
                request(pi + chunks);
            rows_ab<2, DPP>(x, tw, xs, t, q, red_r, red_shift);
#pragma unroll
            for (int n = 0; n < 2; ++n)
            {
                rm_wave_sync();
#pragma unroll
                for (int u = 0; u < 8; u += 2)
                {
                    ulonglong2 v;
                    v.x = x[n][u], v.y = x[n][u + 1];
                    *reinterpret_cast<ulonglong2 *>(xs + ((8 * t + u) ^ (((t >> 1) & 3) << 1))) = v;
                }
                rm_wave_sync();
                u64 *o = out + off + (size_t)(2 * pi + n) * stride;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                {
                    const u32 e = (i * 32 + t) * 2, sw = ((t >> 3) & 3) << 1;
                    const ulonglong2 w = *reinterpret_cast<const ulonglong2 *>(xs + (e ^ sw));
                    typedef u64 u64x2 __attribute__((ext_vector_type(2)));
                    u64x2 v2;
                    v2.x = w.x, v2.y = w.y;
                    __builtin_nontemporal_store(v2, reinterpret_cast<u64x2 *>(o + e));
                }
            }
        }
    }
} // namespace

int main(int argc, char **argv)
{
    const u32 polys = argc > 1 ? (u32)atoi(argv[1]) : 256, limbs = 32;
    const size_t N = 65536, words = (size_t)polys * limbs * N;
    const u64 q = 1152921504606584833ull; // the first 60-bit prime of BASELINE config 2
    const u32 red_r = (u32)((((unsigned __int128)1) << (31 + 60)) / q), red_shift = 60 - 33;
    u64 *in, *a, *b, *table;
    CK(hipMalloc(&in, words * 8));
    CK(hipMalloc(&a, words * 8));
    CK(hipMalloc(&b, words * 8));
    CK(hipMalloc(&table, (size_t)limbs * 3 * N * 8));
    {
        std::vector<u64> h((size_t)limbs * 3 * N);
        u64 s = 88172645463325252ull;
        auto next = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
        for (auto &w : h)
            w = next() % q; // any words: both variants do the same arithmetic on them
        CK(hipMemcpy(table, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        std::vector<u64> d(1 << 22);
        for (auto &w : d)
            w = next() % q;
        for (size_t o = 0; o < words; o += d.size())
            CK(hipMemcpy(in + o, d.data(), (words - o < d.size() ? words - o : d.size()) * 8, hipMemcpyHostToDevice));
    }
    const u32 groups = limbs * (256 / RM_ROWS);
    u32 chunks = (8192 + groups - 1) / groups;
    if (chunks > polys / 4)
        chunks = polys / 4;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best[2] = { 1e30f, 1e30f };
    for (int rep = 0; rep < 6; ++rep)
        for (int v = 0; v < 2; ++v)
        {
            CK(hipEventRecord(e0));
            if (v)
                hipLaunchKernelGGL(rows8_ab<true>, dim3(groups * chunks), dim3(256), 0, 0, in, b, table, q, red_r, red_shift, polys, limbs, chunks);
            else
                hipLaunchKernelGGL(rows8_ab<false>, dim3(groups * chunks), dim3(256), 0, 0, in, a, table, q, red_r, red_shift, polys, limbs, chunks);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best[v])
                best[v] = ms;
        }
    CK(hipGetLastError());
    // word-for-word comparison of the two outputs
    std::vector<u64> ha(1 << 22), hb(1 << 22);
    size_t bad = 0;
    for (size_t o = 0; o < words; o += ha.size())
    {
        const size_t cnt = words - o < ha.size() ? words - o : ha.size();
        CK(hipMemcpy(ha.data(), a + o, cnt * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb.data(), b + o, cnt * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < cnt; ++i)
            bad += ha[i] != hb[i];
    }
    const double bytes = 2.0 * words * 8;
    printf("forward row pass, %u polys x %u limbs x 65536 (%.2f GB), %u workgroups, out of place, best of 5\n", polys, limbs, words * 8 / 1e9, groups * chunks);
    printf("  last exchange through per-wave LDS (library form): %8.3f ms  %7.1f GB/s read + written\n", best[0], bytes / best[0] / 1e6);
    printf("  last exchange by DPP quad permutes + selects      : %8.3f ms  %7.1f GB/s read + written\n", best[1], bytes / best[1] / 1e6);
    printf("  outputs %s (%zu words differ)\n", bad ? "DIFFER" : "identical word for word", bad);
    return bad ? 1 : 0;
}
