// Row pass of the forward negacyclic NTT with 8 coefficients per thread and the row twiddles in LDS.
// Shared by the fused key-switch kernel (keyswitch.hip) and the poly-streaming row pass (ntt.hip).
//
// A workgroup of 256 threads owns RM_ROWS = 8 rows of 256 coefficients of one limb; a wave owns two of them.
// The last 8 stages of SEAL's forward network (util/dwthandler.h:94-192) stay inside a row and their twiddles
// tw[((n1 + row) << S) + G] (S = 0..7, G < 2^S) depend on the limb and the row only, not on the polynomial:
// the workgroup stages them once (split-Shoup form, 3 x 8 x 256 words = 48 KB) and every further row of the
// same limb and position reuses them from LDS.  Three register rounds of 3 + 3 + 2 stages; between them the wave
// exchanges its rows through 4 KB of LDS with XOR layouts that are conflict-free for writer and reader.
#pragma once
#include "modarith.cuh"

namespace
{
    constexpr int RM_ROWS = 8;
    constexpr int RM_TW_WORDS = 3 * RM_ROWS * 256; // [w0 | w1 | q0,q1][row][entry]; entry = rm_entry(S, G)
    constexpr int RM_X_WORDS = RM_ROWS * 256;

    // LDS position of twiddle (S, G), G < 2^S, inside a row's 256-word block: 2^S + G (groups of 2 and 4 consecutive
    // twiddles start 16-byte aligned, so they are fetched by ds_read_b128; at 2^S - 1 + G the compiler had to use
    // ds_read2_b64, whose lane stride of 2 / 4 words is a 2- / 4-way bank conflict); the last stage, where every lane
    // reads its own four twiddles, keeps pairs (G & 2 = 0) and (G & 2 = 2) in two halves of 64 words so that both 16-byte
    // reads are consecutive across lanes
    __device__ __forceinline__ u32 rm_entry(u32 S, u32 G)
    {
        return S == 7 ? 128 + ((G >> 2) << 1) + (G & 1) + ((G >> 1) & 1) * 64 : (1u << S) + G;
    }

    // all 256 threads: lds[0 .. RM_TW_WORDS) <- twiddles of rows tile*8 .. tile*8+7; table = u64[3][N] of the limb
    __device__ __forceinline__ void rm_stage_twiddles(u64 *lds, const u64 *table, u32 log_n, u32 tile)
    {
        const u32 n1 = 1u << (log_n - 8);
        const u32 e = threadIdx.x;
        if (e < 255)
        {
            const u32 S = 31 - __clz(e + 1), G = e + 1 - (1u << S);
            const u32 pos = rm_entry(S, G);
#pragma unroll
            for (int r = 0; r < RM_ROWS; ++r)
            {
                const u32 idx = ((n1 + tile * RM_ROWS + r) << S) + G;
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    lds[(a * RM_ROWS + r) * 256 + pos] = table[((size_t)a << log_n) + idx];
            }
        }
    }

    __device__ __forceinline__ u64 rm_guard(u64 x, u64 K, u64 negK)
    {
        const u64 t = add64(x, negK);
        return x >= K ? t : x;
    }

    // one radix-2 stage over the 8 registers of ND digits: CNT twiddles, register gap 4 / CNT
    // GM: guard of the additive operand: 0 none, 1 subtract K = 4q, 2 subtract 8q (wide ranges, ntt.hip policy Wide)
    // entry2: position of twiddles 2, 3 when CNT == 4 (entry + 2 except in the last stage, see rm_entry)
    template <int ND, int CNT, int GM = 1>
    __device__ __forceinline__ void rm_stage(u64 (&x)[ND][8], const u64 *tw, u32 entry, u64 K, u64 negK, u64 nq,
                                             u32 entry2 = 0xffffffffu)
    {
        constexpr int gap = 4 / CNT;
        MulOp3 w[CNT];
        if (entry2 == 0xffffffffu)
            entry2 = entry + 2;
#pragma unroll
        for (int g = 0; g < CNT; ++g)
        {
            const u32 at = g < 2 ? entry + g : entry2 + (g - 2);
            const u64 qq = tw[2 * RM_ROWS * 256 + at];
            w[g].w0 = tw[at];
            w[g].w1 = tw[RM_ROWS * 256 + at];
            w[g].q0 = (u32)qq;
            w[g].q1 = (u32)(qq >> 32);
        }
#pragma unroll
        for (int g = 0; g < CNT; ++g)
#pragma unroll
            for (int j = 0; j < gap; ++j)
#pragma unroll
                for (int n = 0; n < ND; ++n)
                {
                    const int i0 = g * 2 * gap + j;
                    const u64 u = GM == 0 ? x[n][i0] : GM == 1 ? rm_guard(x[n][i0], K, negK) : rm_guard(x[n][i0], K << 1, negK << 1);
                    const u64 v = mul_split_add(x[n][i0 + gap], w[g], nq, u);
                    x[n][i0] = v;
                    x[n][i0 + gap] = shl1_add64(u, K) - v;
                }
    }

    __device__ __forceinline__ void rm_wave_sync()
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // last 8 stages of ND limb-rows held as x[n][a] = row[a * 32 + t]; result x[n][u] = row[8 t + u], values < 4q
    // WIDE = 1: inputs < 16q, a subtraction of 8q before the even stages only (primes below 2^60)
    // WIDE = 1 with red_r != 0 (NttMod, primes of 36 bits and more): the final reduction from 16q goes by an estimated
    // quotient and the results are below 2q
    // WIDE = 2 (primes of 36 to 57 bits, ntt.hip policy Wider): inputs < 36q, no subtraction in any stage (< 68q < 2^64),
    // the same final reduction
    template <int ND, int WIDE = 0>
    __device__ __forceinline__ void rm_rows(u64 (&x)[ND][8], const u64 *tw, u64 *xs, u32 t, u64 q, u32 red_r = 0,
                                            u32 red_shift = 0)
    {
        const u64 K = q << 2, negK = 0 - K, nq = 0 - q;
        rm_stage<ND, 1, (WIDE == 2 ? 0 : WIDE ? 2 : 1)>(x, tw, 1, K, negK, nq);
        rm_stage<ND, 2, (WIDE ? 0 : 1)>(x, tw, 2, K, negK, nq);
        rm_stage<ND, 4, (WIDE == 2 ? 0 : WIDE ? 2 : 1)>(x, tw, 4, K, negK, nq);
        const u32 A = t >> 2, c = t & 3;
#pragma unroll
        for (int n = 0; n < ND; ++n)
        {
            rm_wave_sync(); // the previous reads of this region are done
#pragma unroll
            for (int a = 0; a < 8; ++a)
                xs[(a << 5) | (((t >> 2) ^ a) << 2) | c] = x[n][a]; // element a*32 + t
            rm_wave_sync();
#pragma unroll
            for (int b = 0; b < 8; ++b)
                x[n][b] = xs[(A << 5) | ((b ^ A) << 2) | c]; // element A*32 + 4b + c
        }
        rm_stage<ND, 1, (WIDE ? 0 : 1)>(x, tw, 8 + A, K, negK, nq);
        rm_stage<ND, 2, (WIDE == 2 ? 0 : WIDE ? 2 : 1)>(x, tw, 16 + 2 * A, K, negK, nq);
        rm_stage<ND, 4, (WIDE ? 0 : 1)>(x, tw, 32 + 4 * A, K, negK, nq);
        const u32 bh = t & 3; // reader t = A*4 + b_hi holds elements 8t + u
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
        rm_stage<ND, 2, (WIDE == 2 ? 0 : WIDE ? 2 : 1)>(x, tw, 64 + 2 * t, K, negK, nq);
        rm_stage<ND, 4, (WIDE ? 0 : 1)>(x, tw, 128 + 2 * t, K, negK, nq, 192 + 2 * t);
        if (WIDE == 2 || (WIDE && red_r != 0))
        {
#pragma unroll
            for (int n = 0; n < ND; ++n)
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    x[n][u] = reduce_below_2q(x[n][u], red_r, red_shift, nq);
            return;
        }
#pragma unroll
        for (int n = 0; n < ND; ++n)
#pragma unroll
            for (int u = 0; u < 8; ++u)
                x[n][u] = rm_guard(WIDE ? rm_guard(x[n][u], K << 1, negK << 1) : x[n][u], K, negK);
    }


    // ---- inverse (Gentleman-Sande) counterpart: the FIRST 8 stages of SEAL's inverse network ------------------
    // (util/dwthandler.h:203-357) stay inside a row; stage T (gap 2^T) uses inv_tw[N - (N >> T) + 1 + row (128 >> T) + G],
    // G < 128 >> T.  LDS entry of (T, G): 256 - (256 >> T) + G (T = 0: see rm_stage_twiddles_inv).  `table` = u64[3][N] of the limb in the shifted
    // inverse layout of the context (logical index i is stored at i - 1).
    __device__ __forceinline__ void rm_stage_twiddles_inv(u64 *lds, const u64 *table, u32 log_n, u32 tile)
    {
        const u32 n = 1u << log_n;
        const u32 e = threadIdx.x;
        if (e < 255)
        {
            const u32 T = 7 - (31 - __clz(255 - e)), G = e - (256 - (256 >> T));
            // first stage (every lane reads its own four twiddles): pairs (G & 2) = 0 and = 2 in two halves of 64 words,
            // so that both 16-byte reads are consecutive across lanes (as rm_entry does for the forward last stage)
            const u32 pos = T == 0 ? ((G >> 2) << 1) + (G & 1) + ((G >> 1) & 1) * 64 : e;
#pragma unroll
            for (int r = 0; r < RM_ROWS; ++r)
            {
                const u32 idx = n - (n >> T) + 1 + (tile * RM_ROWS + r) * (128 >> T) + G;
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    lds[(a * RM_ROWS + r) * 256 + pos] = table[((size_t)a << log_n) + idx - 1];
            }
        }
    }

    // MODE: reduction of the sums: 0 conditional subtraction of K = 4q, 1 none (operands below 4q), 2 estimated quotient
    // from below 16q to below 2q with the differences taken against 8q (ntt.hip, policy InvWide)
    template <int ND, int CNT, int MODE = 0>
    __device__ __forceinline__ void rm_stage_inv(u64 (&x)[ND][8], const u64 *tw, u32 entry, u64 K, u64 negK, u64 nq,
                                                 u32 entry2 = 0xffffffffu, u32 red_r = 0, u32 red_shift = 0)
    {
        constexpr int gap = 4 / CNT;
        MulOp3 w[CNT];
        if (entry2 == 0xffffffffu)
            entry2 = entry + 2;
#pragma unroll
        for (int g = 0; g < CNT; ++g)
        {
            const u32 at = g < 2 ? entry + g : entry2 + (g - 2);
            const u64 qq = tw[2 * RM_ROWS * 256 + at];
            w[g].w0 = tw[at];
            w[g].w1 = tw[RM_ROWS * 256 + at];
            w[g].q0 = (u32)qq;
            w[g].q1 = (u32)(qq >> 32);
        }
#pragma unroll
        for (int g = 0; g < CNT; ++g)
#pragma unroll
            for (int j = 0; j < gap; ++j)
#pragma unroll
                for (int n = 0; n < ND; ++n)
                {
                    const int i0 = g * 2 * gap + j;
                    const u64 u = x[n][i0], v = x[n][i0 + gap];
                    if (MODE == 0)
                        x[n][i0] = rm_guard(add64(u, v), K, negK);
                    else if (MODE == 1)
                        x[n][i0] = add64(u, v);
                    else
                        x[n][i0] = reduce_below_2q(add64(u, v), red_r, red_shift, nq);
                    x[n][i0 + gap] = mul_split(add64(u, MODE == 2 ? K << 1 : K) - v, w[g], nq);
                }
    }

    // in: x[n][u] = row[8 t + u] (values < 4q); out: x[n][a] = row[a * 32 + t], values < 4q
    // W (policy InvWide): stages alternate between no reduction and the estimated-quotient one
    template <int ND, bool W = false>
    __device__ __forceinline__ void rm_rows_inv(u64 (&x)[ND][8], const u64 *tw, u64 *xs, u32 t, u64 q, u32 red_r = 0,
                                                u32 red_shift = 0)
    {
        constexpr int MA = W ? 1 : 0, MB = W ? 2 : 0; // even / odd global stage
        constexpr u32 E2 = 0xffffffffu;
        const u64 K = q << 2, negK = 0 - K, nq = 0 - q;
        const u32 A = t >> 2, c = t & 3, bh = t & 3;
        rm_stage_inv<ND, 4, MA>(x, tw, 2 * t, K, negK, nq, 64 + 2 * t, red_r, red_shift);
        rm_stage_inv<ND, 2, MB>(x, tw, 128 + 2 * t, K, negK, nq, E2, red_r, red_shift);
#pragma unroll
        for (int n = 0; n < ND; ++n)
        {
            rm_wave_sync();
#pragma unroll
            for (int u = 0; u < 8; ++u)
                xs[(u << 5) | (A << 2) | (bh ^ (u & 3))] = x[n][u]; // element 8t + u
            rm_wave_sync();
#pragma unroll
            for (int b = 0; b < 8; ++b)
                x[n][b] = xs[((((b & 1) << 2) | c) << 5) | (A << 2) | ((b >> 1) ^ c)]; // element A*32 + 4b + c
        }
        rm_stage_inv<ND, 4, MA>(x, tw, 192 + 4 * A, K, negK, nq, E2, red_r, red_shift);
        rm_stage_inv<ND, 2, MB>(x, tw, 224 + 2 * A, K, negK, nq, E2, red_r, red_shift);
        rm_stage_inv<ND, 1, MA>(x, tw, 240 + A, K, negK, nq, E2, red_r, red_shift);
#pragma unroll
        for (int n = 0; n < ND; ++n)
        {
            rm_wave_sync();
#pragma unroll
            for (int b = 0; b < 8; ++b)
                xs[(A << 5) | ((b ^ A) << 2) | c] = x[n][b];
            rm_wave_sync();
#pragma unroll
            for (int a = 0; a < 8; ++a)
                x[n][a] = xs[(a << 5) | (((t >> 2) ^ a) << 2) | c]; // element a*32 + t
        }
        rm_stage_inv<ND, 4, MB>(x, tw, 248, K, negK, nq, E2, red_r, red_shift);
        rm_stage_inv<ND, 2, MA>(x, tw, 252, K, negK, nq, E2, red_r, red_shift);
        rm_stage_inv<ND, 1, MB>(x, tw, 254, K, negK, nq, E2, red_r, red_shift);
    }
} // namespace
