// Batched negacyclic NTT / inverse NTT for gfx950 (MI355X).
//
// Replaces util::ntt_negacyclic_harvey[_lazy] (util/ntt.cpp:393-436, util/dwthandler.h:94-192) and
// util::inverse_ntt_negacyclic_harvey[_lazy] (util/ntt.cpp:438-474, util/dwthandler.h:203-357).
// The butterfly network, twiddle order and lazy ranges are SEAL's, so even the *_lazy outputs are the
// reference's exact representatives; only the schedule is new:
//
//   N = N1 x 256 (N >= 4096).  A limb-polynomial is viewed as N1 rows of 256 words.
//   forward : "cols" kernel (first log2 N1 stages, stride-256 butterflies, tile = all N1 rows x C
//             columns, 4096 words staged through LDS) then "rows" kernel (last 8 stages on 16 rows).
//   inverse : "rows" kernel (first 8 stages) then "cols" kernel (last log2 N1 stages, n^-1 merged
//             into the final stage exactly like dwthandler.h:274-315).
//   Each thread keeps 16 coefficients in registers and runs up to 4 radix-2 stages on them between
//   LDS exchanges (one exchange per kernel in cols, two in rows: the second one re-shapes the tile so
//   that global loads/stores are 128-byte contiguous per 16 lanes).
//   N < 4096: one workgroup per limb-polynomial, stage-by-stage in LDS (tests / tiny parameters).
//
// HBM-bound by design: 16 B/coefficient algorithmic, twiddles (16 B each) come from L2.
#include "context.hpp"

namespace
{
    constexpr int ROW_LEN = 256;      // N2
    constexpr int ROW_PITCH = 272;    // 256 + 256/16 padding words (bank-conflict-free transposes)
    constexpr int TILE_WORDS = 4096;

    struct NttArgs
    {
        u64 *data;
        const DevMod *mods;
        const MulOp *tw; // [n_mod][n]
        LimbMap map;
        u32 n_limbs;
        u32 n_polys;
        u32 log_n;
        int lazy;
        u64 poly_stride; // words between consecutive objects (>= n_limbs * n)
        u32 limb_group;  // limb j uses modulus map.id[j / limb_group]
        u32 skip_diag;   // skip limbs with j % limb_group == j / limb_group (already transformed)
    };

    struct Where
    {
        u64 offset; // word offset of the limb-polynomial
        u32 mod;    // modulus index
        u32 tile;
        bool skip;
    };

    // limb is the slowest index so that concurrently running workgroups share one prime's twiddles in L2
    __device__ __forceinline__ Where locate(const NttArgs &a, u32 b, u32 tiles)
    {
        Where w;
        w.tile = b % tiles;
        const u32 rest = b / tiles;
        const u32 poly = rest % a.n_polys, limb = rest / a.n_polys;
        const u32 grp = limb / a.limb_group;
        w.mod = a.map.id[grp];
        w.skip = a.skip_diag && (limb - grp * a.limb_group == grp);
        w.offset = (u64)poly * a.poly_stride + ((u64)limb << a.log_n);
        return w;
    }

    __device__ __forceinline__ u64 guard2(u64 x, u64 q2)
    {
        return x >= q2 ? x - q2 : x;
    }

    // ---- 16-point register networks ---------------------------------------------------------------
    // Forward (Cooley-Tukey) local stage S: register gap 8 >> S, 2^S twiddles tw[(base << S) + g].
    template <int S>
    __device__ __forceinline__ void fwd_stage16(u64 (&x)[16], const MulOp *__restrict__ tw, u32 base, u64 q, u64 q2)
    {
        constexpr int gap = 8 >> S;
#pragma unroll
        for (int g = 0; g < (1 << S); ++g)
        {
            const MulOp w = tw[(base << S) + g];
#pragma unroll
            for (int j = 0; j < gap; ++j)
            {
                const int i0 = g * 2 * gap + j;
                const u64 u = guard2(x[i0], q2);
                const u64 v = mul_lazy(x[i0 + gap], w, q);
                x[i0] = u + v;
                x[i0 + gap] = u + q2 - v;
            }
        }
    }

    // Inverse (Gentleman-Sande) local stage S: register gap 1 << S, 8 >> S twiddles tw[root0 + g].
    template <int S>
    __device__ __forceinline__ void inv_stage16(u64 (&x)[16], const MulOp *__restrict__ tw, u32 root0, u64 q, u64 q2)
    {
        constexpr int gap = 1 << S;
#pragma unroll
        for (int g = 0; g < (8 >> S); ++g)
        {
            const MulOp w = tw[root0 + g];
#pragma unroll
            for (int j = 0; j < gap; ++j)
            {
                const int i0 = g * 2 * gap + j;
                const u64 u = x[i0], v = x[i0 + gap];
                x[i0] = guard2(u + v, q2);
                x[i0 + gap] = mul_lazy(u + q2 - v, w, q);
            }
        }
    }

    // Final inverse stage (single twiddle, merged n^-1): dwthandler.h:274-315.
    template <int S>
    __device__ __forceinline__ void inv_last_stage16(u64 (&x)[16], const DevMod &m, u64 q2)
    {
        constexpr int gap = 1 << S;
#pragma unroll
        for (int g = 0; g < (8 >> S); ++g)
        {
#pragma unroll
            for (int j = 0; j < gap; ++j)
            {
                const int i0 = g * 2 * gap + j;
                const u64 u = guard2(x[i0], q2), v = x[i0 + gap];
                x[i0] = mul_lazy(guard2(u + v, q2), m.inv_n_op, m.inv_n_quo, m.q);
                x[i0 + gap] = mul_lazy(u + q2 - v, m.inv_n_w_op, m.inv_n_w_quo, m.q);
            }
        }
    }

    // ---- forward, pass 1: columns ----------------------------------------------------------------------
    template <int LOGN1>
    __global__ __launch_bounds__(256) void ntt_fwd_cols(NttArgs a)
    {
        constexpr int K1 = LOGN1 - 4;          // stages done in round 1
        constexpr int N1 = 1 << LOGN1;
        constexpr int C = TILE_WORDS / N1;     // columns per workgroup
        constexpr int SUB = 16 >> K1;          // consecutive rows per thread in round 1
        __shared__ u64 lds[TILE_WORDS];

        const Where w = locate(a, blockIdx.x, ROW_LEN / C);
        if (w.skip)
            return;
        const DevMod m = a.mods[w.mod];
        const u64 q = m.q, q2 = q << 1;
        const MulOp *tw = a.tw + ((size_t)w.mod << a.log_n);
        u64 *p = a.data + w.offset + w.tile * C;

        const u32 c = threadIdx.x % C, tau = threadIdx.x / C;
        u64 x[16];
        if (K1 > 0)
        {
#pragma unroll
            for (int e = 0; e < 16; ++e)
            {
                const u32 row = (e >> (4 - K1)) * 16 + tau * SUB + (e & (SUB - 1));
                x[e] = p[row * ROW_LEN + c];
            }
            if (K1 > 0) fwd_stage16<0>(x, tw, 1, q, q2);
            if (K1 > 1) fwd_stage16<1>(x, tw, 1, q, q2);
            if (K1 > 2) fwd_stage16<2>(x, tw, 1, q, q2);
            if (K1 > 3) fwd_stage16<3>(x, tw, 1, q, q2);
#pragma unroll
            for (int e = 0; e < 16; ++e)
            {
                const u32 row = (e >> (4 - K1)) * 16 + tau * SUB + (e & (SUB - 1));
                lds[row * C + c] = x[e];
            }
            __syncthreads();
#pragma unroll
            for (int l = 0; l < 16; ++l)
                x[l] = lds[(tau * 16 + l) * C + c];
        }
        else
        {
#pragma unroll
            for (int l = 0; l < 16; ++l)
                x[l] = p[(tau * 16 + l) * ROW_LEN + c];
        }
        const u32 base = (1u << K1) + tau;
        fwd_stage16<0>(x, tw, base, q, q2);
        fwd_stage16<1>(x, tw, base, q, q2);
        fwd_stage16<2>(x, tw, base, q, q2);
        fwd_stage16<3>(x, tw, base, q, q2);
#pragma unroll
        for (int l = 0; l < 16; ++l)
            p[(tau * 16 + l) * ROW_LEN + c] = x[l];
    }

    // ---- forward, pass 2: rows (16 rows of 256 per workgroup) -------------------------------------------
    __global__ __launch_bounds__(256) void ntt_fwd_rows(NttArgs a)
    {
        __shared__ u64 lds[16 * ROW_PITCH];
        const u32 n1 = 1u << (a.log_n - 8);
        const Where w = locate(a, blockIdx.x, n1 / 16);
        if (w.skip)
            return;
        const DevMod m = a.mods[w.mod];
        const u64 q = m.q, q2 = q << 1;
        const MulOp *tw = a.tw + ((size_t)w.mod << a.log_n);
        const u32 r = threadIdx.x >> 4, t = threadIdx.x & 15;
        const u32 row = w.tile * 16 + r;
        u64 *p = a.data + w.offset + (size_t)row * ROW_LEN;
        u64 *s = lds + r * ROW_PITCH;

        u64 x[16];
#pragma unroll
        for (int h = 0; h < 16; ++h)
            x[h] = p[h * 16 + t];
        const u32 base1 = n1 + row;
        fwd_stage16<0>(x, tw, base1, q, q2);
        fwd_stage16<1>(x, tw, base1, q, q2);
        fwd_stage16<2>(x, tw, base1, q, q2);
        fwd_stage16<3>(x, tw, base1, q, q2);
#pragma unroll
        for (int h = 0; h < 16; ++h)
            s[h * 17 + t] = x[h];
        __syncthreads();
#pragma unroll
        for (int l = 0; l < 16; ++l)
            x[l] = s[t * 17 + l];
        const u32 base2 = base1 * 16 + t;
        fwd_stage16<0>(x, tw, base2, q, q2);
        fwd_stage16<1>(x, tw, base2, q, q2);
        fwd_stage16<2>(x, tw, base2, q, q2);
        fwd_stage16<3>(x, tw, base2, q, q2);
        if (!a.lazy)
        {
#pragma unroll
            for (int l = 0; l < 16; ++l)
                x[l] = csub(guard2(x[l], q2), q);
        }
        __syncthreads();
#pragma unroll
        for (int l = 0; l < 16; ++l)
            s[t * 17 + l] = x[l];
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 16; ++h)
            p[h * 16 + t] = s[h * 17 + t];
    }

    // ---- inverse, pass 1: rows ---------------------------------------------------------------------------
    __global__ __launch_bounds__(256) void ntt_inv_rows(NttArgs a)
    {
        __shared__ u64 lds[16 * ROW_PITCH];
        const u32 n = 1u << a.log_n, n1 = n >> 8;
        const Where w = locate(a, blockIdx.x, n1 / 16);
        if (w.skip)
            return;
        const DevMod m = a.mods[w.mod];
        const u64 q = m.q, q2 = q << 1;
        const MulOp *tw = a.tw + ((size_t)w.mod << a.log_n);
        const u32 r = threadIdx.x >> 4, t = threadIdx.x & 15;
        const u32 row = w.tile * 16 + r;
        u64 *p = a.data + w.offset + (size_t)row * ROW_LEN;
        u64 *s = lds + r * ROW_PITCH;

        u64 x[16];
#pragma unroll
        for (int h = 0; h < 16; ++h)
            s[h * 17 + t] = p[h * 16 + t];
        __syncthreads();
#pragma unroll
        for (int l = 0; l < 16; ++l)
            x[l] = s[t * 17 + l];
        // global stages 0..3: root index n - (n >> T) + 1 + group
        const u32 blk = row * 16 + t;
        inv_stage16<0>(x, tw, n - (n >> 0) + 1 + blk * 8, q, q2);
        inv_stage16<1>(x, tw, n - (n >> 1) + 1 + blk * 4, q, q2);
        inv_stage16<2>(x, tw, n - (n >> 2) + 1 + blk * 2, q, q2);
        inv_stage16<3>(x, tw, n - (n >> 3) + 1 + blk * 1, q, q2);
        __syncthreads();
#pragma unroll
        for (int l = 0; l < 16; ++l)
            s[t * 17 + l] = x[l];
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 16; ++h)
            x[h] = s[h * 17 + t];
        // global stages 4..7
        inv_stage16<0>(x, tw, n - (n >> 4) + 1 + row * 8, q, q2);
        inv_stage16<1>(x, tw, n - (n >> 5) + 1 + row * 4, q, q2);
        inv_stage16<2>(x, tw, n - (n >> 6) + 1 + row * 2, q, q2);
        inv_stage16<3>(x, tw, n - (n >> 7) + 1 + row * 1, q, q2);
#pragma unroll
        for (int h = 0; h < 16; ++h)
            p[h * 16 + t] = x[h];
    }

    // ---- inverse, pass 2: columns --------------------------------------------------------------------------
    template <int LOGN1>
    __global__ __launch_bounds__(256) void ntt_inv_cols(NttArgs a)
    {
        constexpr int K1 = LOGN1 - 4;
        constexpr int N1 = 1 << LOGN1;
        constexpr int C = TILE_WORDS / N1;
        constexpr int SUB = 16 >> K1;
        __shared__ u64 lds[TILE_WORDS];

        const u32 n = 1u << a.log_n;
        const Where w = locate(a, blockIdx.x, ROW_LEN / C);
        if (w.skip)
            return;
        const DevMod m = a.mods[w.mod];
        const u64 q = m.q, q2 = q << 1;
        const MulOp *tw = a.tw + ((size_t)w.mod << a.log_n);
        u64 *p = a.data + w.offset + w.tile * C;
        const u32 c = threadIdx.x % C, tau = threadIdx.x / C;

        u64 x[16];
#pragma unroll
        for (int l = 0; l < 16; ++l)
            x[l] = p[(tau * 16 + l) * ROW_LEN + c];
        // global stages 8..11
        inv_stage16<0>(x, tw, n - (n >> 8) + 1 + tau * 8, q, q2);
        inv_stage16<1>(x, tw, n - (n >> 9) + 1 + tau * 4, q, q2);
        inv_stage16<2>(x, tw, n - (n >> 10) + 1 + tau * 2, q, q2);
        if (K1 == 0)
            inv_last_stage16<3>(x, m, q2);
        else
            inv_stage16<3>(x, tw, n - (n >> 11) + 1 + tau, q, q2);
        if (K1 > 0)
        {
#pragma unroll
            for (int l = 0; l < 16; ++l)
                lds[(tau * 16 + l) * C + c] = x[l];
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 16; ++e)
            {
                const u32 row = (e >> (4 - K1)) * 16 + tau * SUB + (e & (SUB - 1));
                x[e] = lds[row * C + c];
            }
            // global stages 12 .. 12+K1-1 act on register stages S = 4-K1 .. 3; the last one is final
            if (K1 == 4) inv_stage16<0>(x, tw, n - (n >> 12) + 1, q, q2);
            if (K1 == 3) inv_stage16<1>(x, tw, n - (n >> 12) + 1, q, q2);
            if (K1 == 4) inv_stage16<1>(x, tw, n - (n >> 13) + 1, q, q2);
            if (K1 == 2) inv_stage16<2>(x, tw, n - (n >> 12) + 1, q, q2);
            if (K1 == 3) inv_stage16<2>(x, tw, n - (n >> 13) + 1, q, q2);
            if (K1 == 4) inv_stage16<2>(x, tw, n - (n >> 14) + 1, q, q2);
            inv_last_stage16<3>(x, m, q2);
        }
        if (!a.lazy)
        {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                x[e] = csub(x[e], q);
        }
        if (K1 > 0)
        {
#pragma unroll
            for (int e = 0; e < 16; ++e)
            {
                const u32 row = (e >> (4 - K1)) * 16 + tau * SUB + (e & (SUB - 1));
                p[row * ROW_LEN + c] = x[e];
            }
        }
        else
        {
#pragma unroll
            for (int l = 0; l < 16; ++l)
                p[(tau * 16 + l) * ROW_LEN + c] = x[l];
        }
    }

    // ---- small N (< 4096): one workgroup per limb-polynomial, stage by stage in LDS -----------------------
    template <bool FWD>
    __global__ __launch_bounds__(256) void ntt_small(NttArgs a)
    {
        __shared__ u64 s[2048];
        const u32 n = 1u << a.log_n, half = n >> 1;
        const Where w = locate(a, blockIdx.x, 1);
        if (w.skip)
            return;
        const DevMod m = a.mods[w.mod];
        const u64 q = m.q, q2 = q << 1;
        const MulOp *tw = a.tw + ((size_t)w.mod << a.log_n);
        u64 *p = a.data + w.offset;
        for (u32 i = threadIdx.x; i < n; i += blockDim.x)
            s[i] = p[i];
        __syncthreads();
        if (FWD)
        {
            u32 lg = a.log_n - 1; // log2(gap)
            for (u32 mm = 1; mm < n; mm <<= 1, --lg)
            {
                for (u32 j = threadIdx.x; j < half; j += blockDim.x)
                {
                    const u32 i = j >> lg, pos = j & ((1u << lg) - 1);
                    const u32 i0 = (i << (lg + 1)) + pos, i1 = i0 + (1u << lg);
                    const MulOp w = tw[mm + i];
                    const u64 u = guard2(s[i0], q2), v = mul_lazy(s[i1], w, q);
                    s[i0] = u + v;
                    s[i1] = u + q2 - v;
                }
                __syncthreads();
            }
            for (u32 i = threadIdx.x; i < n; i += blockDim.x)
                p[i] = a.lazy ? s[i] : csub(guard2(s[i], q2), q);
        }
        else
        {
            u32 lg = 0;
            for (u32 mm = half; mm > 1; mm >>= 1, ++lg)
            {
                for (u32 j = threadIdx.x; j < half; j += blockDim.x)
                {
                    const u32 i = j >> lg, pos = j & ((1u << lg) - 1);
                    const u32 i0 = (i << (lg + 1)) + pos, i1 = i0 + (1u << lg);
                    const MulOp w = tw[n - 2 * mm + 1 + i];
                    const u64 u = s[i0], v = s[i1];
                    s[i0] = guard2(u + v, q2);
                    s[i1] = mul_lazy(u + q2 - v, w, q);
                }
                __syncthreads();
            }
            for (u32 j = threadIdx.x; j < half; j += blockDim.x)
            {
                const u64 u = guard2(s[j], q2), v = s[j + half];
                s[j] = mul_lazy(guard2(u + v, q2), m.inv_n_op, m.inv_n_quo, q);
                s[j + half] = mul_lazy(u + q2 - v, m.inv_n_w_op, m.inv_n_w_quo, q);
            }
            __syncthreads();
            for (u32 i = threadIdx.x; i < n; i += blockDim.x)
                p[i] = a.lazy ? s[i] : csub(s[i], q);
        }
    }

    template <int LOGN1>
    void launch_cols(bool fwd, const NttArgs &a, u32 blocks, hipStream_t s)
    {
        if (fwd)
            hipLaunchKernelGGL(ntt_fwd_cols<LOGN1>, dim3(blocks), dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL(ntt_inv_cols<LOGN1>, dim3(blocks), dim3(256), 0, s, a);
    }
} // namespace

namespace cat
{
    cat_status launch_ntt(cat_ctx *ctx, bool forward, u64 *d_data, const LimbMap &map, uint32_t n_limbs,
                          uint32_t n_polys, bool lazy, hipStream_t s)
    {
        return launch_ntt_ex(ctx, forward, d_data, map, n_limbs, n_polys, lazy, (u64)n_limbs << ctx->log_n, 1, false,
                             s);
    }

    cat_status launch_ntt_ex(cat_ctx *ctx, bool forward, u64 *d_data, const LimbMap &map, uint32_t n_limbs,
                             uint32_t n_polys, bool lazy, u64 poly_stride, uint32_t limb_group, bool skip_diag,
                             hipStream_t s)
    {
        if (n_limbs == 0 || n_polys == 0)
            return CAT_OK;
        NttArgs a;
        a.data = d_data;
        a.mods = ctx->d_mods;
        a.tw = forward ? ctx->d_tw_fwd : ctx->d_tw_inv;
        a.map = map;
        a.n_limbs = n_limbs;
        a.n_polys = n_polys;
        a.log_n = ctx->log_n;
        a.lazy = lazy ? 1 : 0;
        a.poly_stride = poly_stride;
        a.limb_group = limb_group;
        a.skip_diag = skip_diag ? 1 : 0;
        const uint64_t n_lp = (uint64_t)n_limbs * n_polys;
        if (ctx->log_n < 12)
        {
            if (n_lp > 0x7fffffffu)
                return fail(CAT_ERR_LOGIC, "invalid parameters");
            const u32 threads = ctx->n / 2 >= 256 ? 256 : 64;
            if (forward)
                hipLaunchKernelGGL(ntt_small<true>, dim3((u32)n_lp), dim3(threads), 0, s, a);
            else
                hipLaunchKernelGGL(ntt_small<false>, dim3((u32)n_lp), dim3(threads), 0, s, a);
            CAT_HIP(hipGetLastError());
            return CAT_OK;
        }
        const u32 logn1 = ctx->log_n - 8;
        const u32 n1 = 1u << logn1;
        const uint64_t row_blocks = n_lp * (n1 / 16);
        const uint64_t col_blocks = n_lp * (ROW_LEN / (TILE_WORDS / n1));
        if (row_blocks > 0x7fffffffu || col_blocks > 0x7fffffffu)
            return fail(CAT_ERR_LOGIC, "invalid parameters");
        auto cols = [&]() {
            switch (logn1)
            {
            case 4: launch_cols<4>(forward, a, (u32)col_blocks, s); break;
            case 5: launch_cols<5>(forward, a, (u32)col_blocks, s); break;
            case 6: launch_cols<6>(forward, a, (u32)col_blocks, s); break;
            case 7: launch_cols<7>(forward, a, (u32)col_blocks, s); break;
            default: launch_cols<8>(forward, a, (u32)col_blocks, s); break;
            }
        };
        if (forward)
        {
            cols();
            hipLaunchKernelGGL(ntt_fwd_rows, dim3((u32)row_blocks), dim3(256), 0, s, a);
        }
        else
        {
            hipLaunchKernelGGL(ntt_inv_rows, dim3((u32)row_blocks), dim3(256), 0, s, a);
            cols();
        }
        CAT_HIP(hipGetLastError());
        return CAT_OK;
    }
} // namespace cat
