// Key switching (Evaluator::switch_key_inplace, evaluator.cpp:2523-2820, CKKS + BFV branches) and
// rescale / BFV mod-switch (RNSTool::divide_and_round_q_last[_ntt]_inplace, util/rns.cpp:789-901).
//
// Notation as in the reference: d = decomp_modulus_size (= L), k = key_modulus_size, special prime
// P = q_{k-1}, key_index(I) = I < d ? I : k-1.
//
//  step 1  t_target <- target (CKKS: inverse NTT, canonical)                         :2604-2611
//  step 2  raise: ext[I][J] = t_target[J] mod q_key_index(I)   (copy when q_J <= q)  :2643-2651
//          forward lazy NTT of every ext[I][J] except CKKS I == J (original limb)    :2635-2654
//  step 3  inner product with the key in 128-bit accumulators, one Barrett-128       :2658-2708
//  step 4  mod-down: INTT_P(t_last), +P/2, Barrett-64; per limb: reduce, +fix,
//          NTT (CKKS) / INTT of the product (BFV), subtract, * P^-1, add to ct       :2759-2818
//          BGV: the correction is t_last + P * [-t_last P^-1]_t instead of rounding  :2715-2758
// BGV shares steps 1-3 with CKKS (NTT-form ciphertexts) and, for mod switching,
// RNSTool::mod_t_and_divide_q_last_ntt_inplace (util/rns.cpp:1203-1246).
#include "context.hpp"
#include "rows8.cuh"
#include <cstdlib>

static_assert(CAT_MAX_KEYSWITCH_LIMBS < CAT_MAX_MODULI && CAT_MAX_KEYSWITCH_LIMBS <= 64,
              "rm_mac's carry-free high word holds at most 64 terms (modarith.cuh)");

namespace
{
    struct MacArgs
    {
        const u64 *ext;    // [batch][d+1][d][N]
        const u64 *target; // [batch][d][N] original (NTT form) for CKKS I == J
        const u64 *key;    // [k-1][2][k][N]
        u64 *prod;         // [batch][2][d+1][N]
        const DevMod *mods;
        u32 d, k, log_n;
        int use_diag;
        u64 target_stride; // words between consecutive target objects
        u64 batch;
        u64 total;         // ceil(batch / BB) * (d+1) * N/2 threads
    };

    // One thread per (batch tile, I, coefficient pair): the two key words of (J, component) are loaded once and
    // used for BB ciphertexts of the batch, so the key (2 L (L+1) N words, the largest stream of the call) is
    // read batch/BB times instead of batch times.
    template <int BB>
    __global__ __launch_bounds__(256) void ks_mac_kernel(MacArgs g)
    {
        const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
        if (i >= g.total)
            return;
        const u32 n2 = 1u << (g.log_n - 1);
        const u32 c = (u32)(i & (n2 - 1));
        const u64 r = i >> (g.log_n - 1);
        const u32 I = (u32)(r % (g.d + 1));
        const u64 b0 = (r / (g.d + 1)) * BB;
        const u32 ki = I == g.d ? g.k - 1 : I;
        const DevMod m = g.mods[ki];
        u64 lo[BB][4], hi[BB][4]; // [b][comp * 2 + coefficient]
#pragma unroll
        for (int b = 0; b < BB; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                lo[b][e] = hi[b][e] = 0;
        const ulonglong2 *ext = reinterpret_cast<const ulonglong2 *>(g.ext) + c;
        const ulonglong2 *tgt = reinterpret_cast<const ulonglong2 *>(g.target) + c;
        const ulonglong2 *kp = reinterpret_cast<const ulonglong2 *>(g.key) + ((u64)ki << (g.log_n - 1)) + c;
        for (u32 J = 0; J < g.d; ++J)
        {
            const ulonglong2 k0 = kp[((u64)(J * 2 + 0) * g.k) << (g.log_n - 1)];
            const ulonglong2 k1 = kp[((u64)(J * 2 + 1) * g.k) << (g.log_n - 1)];
            const bool diag = g.use_diag && I == J;
#pragma unroll
            for (int b = 0; b < BB; ++b)
            {
                if (b0 + b < g.batch)
                {
                    const ulonglong2 x = diag ? tgt[(b0 + b) * (g.target_stride >> 1) + ((u64)J << (g.log_n - 1))]
                                              : ext[((((b0 + b) * (g.d + 1) + I) * g.d + J) << (g.log_n - 1))];
                    mac128(lo[b][0], hi[b][0], x.x, k0.x);
                    mac128(lo[b][1], hi[b][1], x.y, k0.y);
                    mac128(lo[b][2], hi[b][2], x.x, k1.x);
                    mac128(lo[b][3], hi[b][3], x.y, k1.y);
                }
            }
        }
        // d < 256: the reference's lazy counter never triggers an intermediate reduction (:2619-2707)
#pragma unroll
        for (int b = 0; b < BB; ++b)
        {
            if (b0 + b < g.batch)
            {
                ulonglong2 *o = reinterpret_cast<ulonglong2 *>(g.prod) + ((((b0 + b) * 2) * (g.d + 1) + I) << (g.log_n - 1)) + c;
                ulonglong2 r0, r1;
                r0.x = barrett128(lo[b][0], hi[b][0], m.q, m.ratio0, m.ratio1);
                r0.y = barrett128(lo[b][1], hi[b][1], m.q, m.ratio0, m.ratio1);
                r1.x = barrett128(lo[b][2], hi[b][2], m.q, m.ratio0, m.ratio1);
                r1.y = barrett128(lo[b][3], hi[b][3], m.q, m.ratio0, m.ratio1);
                o[0] = r0;
                o[(u64)(g.d + 1) << (g.log_n - 1)] = r1;
            }
        }
    }

    // ---- fused second NTT pass + inner product (N >= 4096) --------------------------------------------------
    // One workgroup owns 8 rows (2048 coefficients) of ONE output limb I of one ciphertext and walks over all
    // decomposition digits J: for each digit it finishes the forward transform of ext[I][J] (the last 8 stages,
    // rows of 256; the first pass ran as ntt_fwd_cols with the modulus raise fused) and multiplies the result
    // straight from registers into the 128-bit accumulators of both key components.  The transformed extended
    // limbs -- d (d+1) N words, the largest buffer of the call -- are never written or re-read.
    //  * 8 coefficients per thread (three register rounds of 3 + 3 + 2 stages), two digits in flight per thread:
    //    they share every twiddle fetch and give the VALU two independent chains.
    //  * The row twiddles do not depend on the digit: the workgroup stages its 8 x 255 of them (split-Shoup form,
    //    48 KB) in LDS once and reads them 12 times; the vector-memory queue then carries only the two streams
    //    (digit data, key), so the compiler's counted waits never block on a twiddle behind an HBM load.
    //  * A wave owns two rows and exchanges them through its own 4 KB of LDS (wave-level sync only); both
    //    exchanges use XOR layouts that are conflict-free for their write and their read pattern.
    //  * Workgroup order: the `batch` workgroups that read the same key tile are 8 ids apart, i.e. on the same
    //    XCD at the same time, so the key tile comes from HBM once and from that XCD's L2 afterwards.

    struct RowsMacArgs
    {
        const u64 *ext;    // [batch][d+1][d][N], first pass done
        const u64 *target; // [batch][d][N] original (NTT form) for CKKS I == J
        const u64 *key;    // [k-1][2][k][N]
        u64 *prod;         // [batch][2][d+1][N]
        const DevMod *mods;
        const NttMod *nmods;
        const u64 *tw3;    // [n_mod][3][N] forward split-Shoup tables
        u32 d, k, log_n, batch;
        int use_diag;
        u64 target_stride;
        u32 groups;        // (d+1) * N / 2048
        u32 nb;            // ks_rows_mac_kernel<.., true>: ciphertexts per workgroup
    };

    __device__ __forceinline__ void rm_load8(u64 (&v)[8], const u64 *p)
    {
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            const ulonglong2 w = reinterpret_cast<const ulonglong2 *>(p)[i];
            v[2 * i] = w.x;
            v[2 * i + 1] = w.y;
        }
    }

    // Every prime of a scheme context is below 2^60 (cat_ctx_create), so the rows run with the wide lazy ranges and
    // their results (< 4q < 2^62) meet rm_mac's operand bound as they are.
    template <int WIDE, bool MULTI>
    __global__ __launch_bounds__(256, 2) void ks_rows_mac_kernel(RowsMacArgs g)
    {
        __shared__ __align__(16) u64 lds[RM_TW_WORDS + RM_X_WORDS];
        // id -> (group, ciphertext or run of nb ciphertexts): ids 8 apart share the group (= the key tile) and the XCD
        const u32 cb = MULTI ? (g.batch + g.nb - 1) / g.nb : g.batch;
        const u32 chunk = blockIdx.x / (8 * cb), within = blockIdx.x - chunk * 8 * cb;
        const u32 b0 = (within >> 3) * (MULTI ? g.nb : 1u), grp = chunk * 8 + (within & 7);
        if (grp >= g.groups)
            return;
        const u32 n1 = 1u << (g.log_n - 8), tiles = n1 / RM_ROWS;
        const u32 I = grp / tiles, tile = grp - I * tiles;
        const u32 ki = I == g.d ? g.k - 1 : I;
        const u64 q = g.nmods[ki].q;
        const u32 red_r = g.nmods[ki].red_r, red_shift = g.nmods[ki].red_shift;
        rm_stage_twiddles(lds, g.tw3 + ((size_t)ki * 3 << g.log_n), g.log_n, tile);
        __syncthreads();
        const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const u32 rl = wave * 2 + (lane >> 5), t = lane & 31;
        const u32 row = tile * RM_ROWS + rl;
        const u64 *tw = lds + rl * 256;
        u64 *xs = lds + RM_TW_WORDS + rl * 256;
        const size_t N = (size_t)1 << g.log_n;
        const size_t roff = (size_t)row * 256;

        // MULTI: the workgroup keeps its staged twiddles for g.nb ciphertexts in a row (launches with many more workgroups
        // than the card holds and few digits: the 48 KB of twiddles are a quarter of such a workgroup's traffic)
        for (u32 bb = 0; bb < (MULTI ? g.nb : 1u); ++bb)
        {
            const u32 b = b0 + bb;
            if (MULTI && b >= g.batch)
                break;
            Acc acc[2][8];
#pragma unroll
            for (int cidx = 0; cidx < 2; ++cidx)
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    acc[cidx][u].lo = acc[cidx][u].hi = 0, acc[cidx][u].cc = 0;
            const u64 *kbase = g.key + (size_t)ki * N + roff + 8 * t;
            const u64 *ebase = g.ext + ((size_t)b * (g.d + 1) + I) * g.d * N + roff + t;
            // one or two digits: their middle sums share one 64-bit word per accumulator and are folded once
            auto mac = [&](const u64 (&x0)[8], u32 J0, const u64 (&x1)[8], u32 J1, bool two) {
#pragma unroll
                for (int cidx = 0; cidx < 2; ++cidx)
                {
                    u64 kv[8], mid[8];
                    rm_load8(kv, kbase + (size_t)(J0 * 2 + cidx) * g.k * N);
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                    {
                        mid[u] = 0;
                        rm_mac(acc[cidx][u], mid[u], x0[u], kv[u]);
                    }
                    if (two)
                    {
                        rm_load8(kv, kbase + (size_t)(J1 * 2 + cidx) * g.k * N);
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            rm_mac(acc[cidx][u], mid[u], x1[u], kv[u]);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        rm_fold(acc[cidx][u], mid[u]);
                }
            };
            const bool diag = g.use_diag && I < g.d;
            if (diag)
            {
                u64 xv[8];
                rm_load8(xv, g.target + (size_t)b * g.target_stride + (size_t)I * N + roff + 8 * t);
                mac(xv, I, xv, I, false);
            }
            const u32 cnt = diag ? g.d - 1 : g.d;
            u32 jj = 0;
            for (; jj + 2 <= cnt; jj += 2)
            {
                const u32 J0 = (diag && jj >= I) ? jj + 1 : jj, J1 = (diag && jj + 1 >= I) ? jj + 2 : jj + 1;
                u64 x[2][8];
#pragma unroll
                for (int a = 0; a < 8; ++a)
                {
                    x[0][a] = ebase[(size_t)J0 * N + a * 32];
                    x[1][a] = ebase[(size_t)J1 * N + a * 32];
                }
                // key words of component 0 are requested before the transform (behind the digit data in the memory
                // queue), those of component 1 while component 0 is being accumulated
                const u64 *kp0 = kbase + (size_t)(J0 * 2) * g.k * N, *kp1 = kbase + (size_t)(J1 * 2) * g.k * N;
                u64 ka[8], kb[8];
                rm_load8(ka, kp0);
                rm_load8(kb, kp1);
                rm_rows<2, WIDE>(x, tw, xs, t, q, red_r, red_shift);
#pragma unroll
                for (int cidx = 0; cidx < 2; ++cidx)
                {
                    u64 kc[8], kd[8], mid[8];
                    if (cidx == 0)
                    {
                        rm_load8(kc, kp0 + (size_t)g.k * N);
                        rm_load8(kd, kp1 + (size_t)g.k * N);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        rm_mac2(acc[cidx][u], mid[u], x[0][u], ka[u], x[1][u], kb[u]);
#pragma unroll
                    for (int u = 0; u < 8; u += 4)
                        rm_fold4(acc[cidx][u], acc[cidx][u + 1], acc[cidx][u + 2], acc[cidx][u + 3], mid[u], mid[u + 1],
                                 mid[u + 2], mid[u + 3]);
                    if (cidx == 0)
                    {
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            ka[u] = kc[u], kb[u] = kd[u];
                    }
                }
            }
            if (jj < cnt)
            {
                const u32 J0 = (diag && jj >= I) ? jj + 1 : jj;
                u64 x[1][8];
#pragma unroll
                for (int a = 0; a < 8; ++a)
                    x[0][a] = ebase[(size_t)J0 * N + a * 32];
                rm_rows<1, WIDE>(x, tw, xs, t, q, red_r, red_shift);
                mac(x[0], J0, x[0], J0, false);
            }
            // d < 256: the reference's lazy counter never triggers an intermediate reduction (:2619-2707)
            const DevMod m = g.mods[ki];
#pragma unroll
            for (int cidx = 0; cidx < 2; ++cidx)
            {
                u64 *o = g.prod + (((size_t)b * 2 + cidx) * (g.d + 1) + I) * N + roff + 8 * t;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                {
                    ulonglong2 r;
                    const Acc &e0 = acc[cidx][2 * i], &e1 = acc[cidx][2 * i + 1];
                    r.x = barrett128(e0.lo, e0.hi + e0.cc, m.q, m.ratio0, m.ratio1);
                    r.y = barrett128(e1.lo, e1.hi + e1.cc, m.q, m.ratio0, m.ratio1);
                    reinterpret_cast<ulonglong2 *>(o)[i] = r;
                }
            }
        
        }
    }

    struct DownArgs
    {
        u64 *prod;      // [batch][2][d+1][N]; limb d holds INTT_P (lazy) of the special-prime product
        u64 *tmp;       // [batch][2][d][N]
        u64 *enc;       // [batch][2][d][N]
        const DevMod *mods;
        const MulOp *inv_mod; // [k][k]
        const u64 *half_mod;  // [k][k]
        const MulOp *q_mod;   // [k][k]: q_a mod q_b as a multiply operand
        u32 d, k, log_n;
        int lazy_shift;  // qi_lazy = q << lazy_shift (2 for CKKS, 1 for BFV)
        u64 enc_stride2; // vectors between consecutive ciphertexts in enc
        u64 total2;      // batch*2*d*N/2
    };

    // tmp[b][comp][j] = ((t_last + P/2 mod P) mod q_j) + (q_j - (P/2 mod q_j))          :2766-2789
    __global__ __launch_bounds__(256) void ks_down_prepare_kernel(DownArgs g)
    {
        const u64 stride = (u64)gridDim.x * blockDim.x;
        const u32 n2 = 1u << (g.log_n - 1);
        const DevMod P = g.mods[g.k - 1];
        const u64 half = P.q >> 1;
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < g.total2; i += stride)
        {
            const u32 c = (u32)(i & (n2 - 1));
            const u64 r = i >> (g.log_n - 1);
            const u32 j = (u32)(r % g.d);
            const u64 bc = r / g.d; // b*2 + comp
            const DevMod m = g.mods[j];
            ulonglong2 v =
                reinterpret_cast<const ulonglong2 *>(g.prod)[((bc * (g.d + 1) + g.d) << (g.log_n - 1)) + c];
            v.x = barrett64(v.x + half, P.q, P.ratio1);
            v.y = barrett64(v.y + half, P.q, P.ratio1);
            if (P.q > m.q)
            {
                v.x = barrett64(v.x, m.q, m.ratio1);
                v.y = barrett64(v.y, m.q, m.ratio1);
            }
            const u64 fix = m.q - g.half_mod[(g.k - 1) * g.k + j];
            v.x += fix;
            v.y += fix;
            reinterpret_cast<ulonglong2 *>(g.tmp)[i] = v;
        }
    }

    // BGV: the plain modulus t with Q^-1 mod t (Q = the prime being divided out), by value
    struct BgvArgs
    {
        u64 t, t_ratio1;
        MulOp inv_Q; // Q^-1 mod t as a multiply operand modulo t
    };

    // v canonical modulo Q.  k = [-v Q^-1]_t (evaluator.cpp:2727-2732, rns.cpp:1215-1222);
    // delta = ((k mod q) (Q mod q) + (v mod q)) mod q (evaluator.cpp:2738-2748, rns.cpp:1228-1237): v + k Q is
    // divisible by Q and congruent to 0 modulo t, so subtracting it keeps the plaintext.
    __device__ __forceinline__ u64 bgv_k(u64 v, const BgvArgs &pt)
    {
        const u64 k = neg_mod(barrett64(v, pt.t, pt.t_ratio1), pt.t);
        return mul_shoup(k, pt.inv_Q.op, pt.inv_Q.quo, pt.t);
    }
    // the part of delta that depends on the limb's prime; k = bgv_k(v) does not
    __device__ __forceinline__ u64 bgv_delta_k(u64 v, u64 k, const DevMod &m, const MulOp &Qm)
    {
        const u64 d = mul_shoup(barrett64(k, m.q, m.ratio1), Qm.op, Qm.quo, m.q);
        return add_mod(d, barrett64(v, m.q, m.ratio1), m.q);
    }
    __device__ __forceinline__ u64 bgv_delta(u64 v, const BgvArgs &pt, const DevMod &m, const MulOp &Qm)
    {
        return bgv_delta_k(v, bgv_k(v, pt), m, Qm);
    }

    // BGV: tmp[b][comp][j] = delta_j(t_last), t_last = INTT_P (canonical) of the special-prime product
    __global__ __launch_bounds__(256) void ks_down_prepare_bgv_kernel(DownArgs g, BgvArgs pt)
    {
        const u64 stride = (u64)gridDim.x * blockDim.x;
        const u32 n2 = 1u << (g.log_n - 1);
        const u64 pairs = g.total2 / g.d; // batch * 2 * N / 2: a thread walks over the limbs (k is formed once)
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += stride)
        {
            const u32 c = (u32)(i & (n2 - 1));
            const u64 bc = i >> (g.log_n - 1); // b*2 + comp
            const ulonglong2 v =
                reinterpret_cast<const ulonglong2 *>(g.prod)[((bc * (g.d + 1) + g.d) << (g.log_n - 1)) + c];
            const u64 kx = bgv_k(v.x, pt), ky = bgv_k(v.y, pt);
            ulonglong2 *o = reinterpret_cast<ulonglong2 *>(g.tmp) + ((bc * g.d) << (g.log_n - 1)) + c;
            for (u32 j = 0; j < g.d; ++j)
            {
                const DevMod m = g.mods[j];
                const MulOp Qm = g.q_mod[(g.k - 1) * g.k + j];
                ulonglong2 r;
                r.x = bgv_delta_k(v.x, kx, m, Qm);
                r.y = bgv_delta_k(v.y, ky, m, Qm);
                o[(u64)j << (g.log_n - 1)] = r;
            }
        }
    }

    // enc[b][comp][j] += (prod[b][comp][j] + qi_lazy - tmp[b][comp][j]) * P^-1 mod q_j   :2810-2816
    // INLINE (BFV: the correction needs no transform of its own, :2791-2808 is skipped in coefficient form): tmp is formed
    // here from the special-prime limb as ks_down_prepare_kernel would have written it, no tmp buffer, one kernel less
    template <bool INLINE>
    __global__ __launch_bounds__(256) void ks_down_finish_kernel(DownArgs g)
    {
        const u64 stride = (u64)gridDim.x * blockDim.x;
        const u32 n2 = 1u << (g.log_n - 1);
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < g.total2; i += stride)
        {
            const u32 c = (u32)(i & (n2 - 1));
            const u64 r = i >> (g.log_n - 1);
            const u32 j = (u32)(r % g.d);
            const u64 bc = r / g.d;
            const u64 q = g.mods[j].q;
            const MulOp pinv = g.inv_mod[(g.k - 1) * g.k + j];
            const u64 qi_lazy = q << g.lazy_shift;
            ulonglong2 t;
            if (INLINE)
            {
                const DevMod P = g.mods[g.k - 1], m = g.mods[j];
                const u64 half = P.q >> 1, fix = q - g.half_mod[(g.k - 1) * g.k + j];
                t = reinterpret_cast<const ulonglong2 *>(g.prod)[((bc * (g.d + 1) + g.d) << (g.log_n - 1)) + c];
                t.x = barrett64(t.x + half, P.q, P.ratio1);
                t.y = barrett64(t.y + half, P.q, P.ratio1);
                if (P.q > q)
                {
                    t.x = barrett64(t.x, q, m.ratio1);
                    t.y = barrett64(t.y, q, m.ratio1);
                }
                t.x += fix;
                t.y += fix;
            }
            else
                t = reinterpret_cast<const ulonglong2 *>(g.tmp)[i];
            const ulonglong2 p =
                reinterpret_cast<const ulonglong2 *>(g.prod)[((bc * (g.d + 1) + j) << (g.log_n - 1)) + c];
            ulonglong2 *ep = reinterpret_cast<ulonglong2 *>(g.enc) + (bc >> 1) * g.enc_stride2 +
                             (((bc & 1) * g.d + j) << (g.log_n - 1)) + c;
            ulonglong2 e = *ep;
            e.x = add_mod(mul_shoup(p.x + qi_lazy - t.x, pinv.op, pinv.quo, q), e.x, q);
            e.y = add_mod(mul_shoup(p.y + qi_lazy - t.y, pinv.op, pinv.quo, q), e.y, q);
            *ep = e;
        }
    }

    // ---- relinearize + rescale tail (launch_switch_key with rescale_out) -------------------------------------
    struct TailArgs
    {
        const u64 *prod; // [batch][2][d+1][N]
        const u64 *enc;  // [batch][2][d][N], enc_stride2 vectors apart
        u64 *buf;        // [batch*2][2][N]: {special-prime product, enc[d-1] + prod[d-1] P^-1}
        const DevMod *mods;
        const MulOp *inv_mod;
        const u64 *half_mod;
        const MulOp *q_mod; // [k][k]: q_a mod q_b as a multiply operand (BGV)
        u32 d, k, log_n;
        u64 enc_stride2;
        u64 total2; // batch*2*N/2
    };

    // buf[p][0] = prod[p][d];  buf[p][1] = enc[p][d-1] + prod[p][d-1] * P^-1 mod q_{d-1}
    __global__ __launch_bounds__(256) void tail_prepare_kernel(TailArgs g)
    {
        const u64 stride = (u64)gridDim.x * blockDim.x;
        const u32 n2 = 1u << (g.log_n - 1);
        const u64 q = g.mods[g.d - 1].q;
        const MulOp pinv = g.inv_mod[(g.k - 1) * g.k + (g.d - 1)];
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < g.total2; i += stride)
        {
            const u32 c = (u32)(i & (n2 - 1));
            const u64 p = i >> (g.log_n - 1); // b*2 + comp
            const ulonglong2 *pr = reinterpret_cast<const ulonglong2 *>(g.prod) + ((p * (g.d + 1)) << (g.log_n - 1)) + c;
            const ulonglong2 sp = pr[(u64)g.d << (g.log_n - 1)], pl = pr[(u64)(g.d - 1) << (g.log_n - 1)];
            const ulonglong2 e = reinterpret_cast<const ulonglong2 *>(g.enc)[(p >> 1) * g.enc_stride2 +
                                                                            ((((p & 1) * g.d) + g.d - 1) << (g.log_n - 1)) + c];
            ulonglong2 u;
            u.x = add_mod(e.x, mul_shoup(pl.x, pinv.op, pinv.quo, q), q);
            u.y = add_mod(e.y, mul_shoup(pl.y, pinv.op, pinv.quo, q), q);
            ulonglong2 *o = reinterpret_cast<ulonglong2 *>(g.buf) + ((p * 2) << (g.log_n - 1)) + c;
            o[0] = sp;
            o[n2] = u;
        }
    }

    // after the inverse transforms of both limbs of buf: buf[p][1] <- buf[p][1] - round_{q_{d-1}}(buf[p][0]) * P^-1,
    // the inverse transform of the relinearized last limb (the forward transform of the mod-down correction and
    // this inverse transform cancel)
    __global__ __launch_bounds__(256) void tail_last_kernel(TailArgs g)
    {
        const u64 stride = (u64)gridDim.x * blockDim.x;
        const u32 n2 = 1u << (g.log_n - 1);
        const DevMod P = g.mods[g.k - 1], m = g.mods[g.d - 1];
        const u64 half = P.q >> 1;
        const u64 fix = m.q - g.half_mod[(g.k - 1) * g.k + (g.d - 1)];
        const MulOp pinv = g.inv_mod[(g.k - 1) * g.k + (g.d - 1)];
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < g.total2; i += stride)
        {
            const u32 c = (u32)(i & (n2 - 1));
            const u64 p = i >> (g.log_n - 1);
            ulonglong2 *o = reinterpret_cast<ulonglong2 *>(g.buf) + ((p * 2) << (g.log_n - 1)) + c;
            ulonglong2 v = o[0];
            const ulonglong2 u = o[n2];
            v.x = barrett64(v.x + half, P.q, P.ratio1);
            v.y = barrett64(v.y + half, P.q, P.ratio1);
            if (P.q > m.q)
            {
                v.x = barrett64(v.x, m.q, m.ratio1);
                v.y = barrett64(v.y, m.q, m.ratio1);
            }
            ulonglong2 r;
            r.x = sub_mod(u.x, mul_shoup(v.x + fix, pinv.op, pinv.quo, m.q), m.q);
            r.y = sub_mod(u.y, mul_shoup(v.y + fix, pinv.op, pinv.quo, m.q), m.q);
            o[n2] = r;
        }
    }

    // BGV form of tail_last_kernel: the mod-down correction of the last limb is delta_{d-1}(buf[p][0]) (bgv_delta)
    __global__ __launch_bounds__(256) void tail_last_bgv_kernel(TailArgs g, BgvArgs ptP)
    {
        const u64 stride = (u64)gridDim.x * blockDim.x;
        const u32 n2 = 1u << (g.log_n - 1);
        const DevMod m = g.mods[g.d - 1];
        const MulOp Pm = g.q_mod[(g.k - 1) * g.k + (g.d - 1)];
        const MulOp pinv = g.inv_mod[(g.k - 1) * g.k + (g.d - 1)];
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < g.total2; i += stride)
        {
            const u32 c = (u32)(i & (n2 - 1));
            const u64 p = i >> (g.log_n - 1);
            ulonglong2 *o = reinterpret_cast<ulonglong2 *>(g.buf) + ((p * 2) << (g.log_n - 1)) + c;
            const ulonglong2 v = o[0], u = o[n2];
            ulonglong2 r;
            r.x = sub_mod(u.x, mul_shoup(bgv_delta(v.x, ptP, m, Pm), pinv.op, pinv.quo, m.q), m.q);
            r.y = sub_mod(u.y, mul_shoup(bgv_delta(v.y, ptP, m, Pm), pinv.op, pinv.quo, m.q), m.q);
            o[n2] = r;
        }
    }

    // BGV: tmp[p][j] = delta1_j P^-1 + delta2_j (mod q_j, canonical), j < d-1: delta1 from buf[p][0] (the special-prime
    // product, divided out by the mod-down), delta2 from buf[p][1] (the relinearized last limb, divided out by the
    // mod switch); one forward transform of this sum serves both corrections
    // A thread owns two coefficients of one polynomial and walks over the limbs: k = [-v Q^-1]_t (two reductions and a
    // product modulo t per value) does not depend on the limb and is formed once, not d - 1 times (541 us per BGV step
    // of batch 64 at N = 32768 when it was a thread per limb and coefficient pair)
    __global__ __launch_bounds__(256) void tail_delta_bgv_kernel(TailArgs g, BgvArgs ptP, BgvArgs ptL, u64 *tmp, u64 total2)
    {
        const u64 stride = (u64)gridDim.x * blockDim.x;
        const u32 n2 = 1u << (g.log_n - 1), dl = g.d - 1;
        const u64 pairs = total2 / dl; // polys * N / 2
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += stride)
        {
            const u32 c = (u32)(i & (n2 - 1));
            const u64 p = i >> (g.log_n - 1);
            const ulonglong2 *b = reinterpret_cast<const ulonglong2 *>(g.buf) + ((p * 2) << (g.log_n - 1)) + c;
            const ulonglong2 v = b[0], l = b[n2];
            const u64 kvx = bgv_k(v.x, ptP), kvy = bgv_k(v.y, ptP), klx = bgv_k(l.x, ptL), kly = bgv_k(l.y, ptL);
            ulonglong2 *o = reinterpret_cast<ulonglong2 *>(tmp) + ((p * dl) << (g.log_n - 1)) + c;
            for (u32 j = 0; j < dl; ++j)
            {
                const DevMod m = g.mods[j];
                const MulOp Pm = g.q_mod[(g.k - 1) * g.k + j], Lm = g.q_mod[dl * g.k + j];
                const MulOp pinv = g.inv_mod[(g.k - 1) * g.k + j];
                ulonglong2 r;
                r.x = add_mod(mul_shoup(bgv_delta_k(v.x, kvx, m, Pm), pinv.op, pinv.quo, m.q), bgv_delta_k(l.x, klx, m, Lm), m.q);
                r.y = add_mod(mul_shoup(bgv_delta_k(v.y, kvy, m, Pm), pinv.op, pinv.quo, m.q), bgv_delta_k(l.y, kly, m, Lm), m.q);
                o[(u64)j << (g.log_n - 1)] = r;
            }
        }
    }

    // ---- rescale ------------------------------------------------------------------------------------------
    struct RescaleArgs
    {
        const u64 *in;  // [polys][L][N]
        const u64 *last; // [polys][N]: INTT (canonical) of the last limb (CKKS) or the last limb itself (BFV)
        u64 *tmp;       // [polys][L-1][N]
        u64 *out;       // [polys][L-1][N]
        const DevMod *mods;
        const MulOp *inv_mod;
        const u64 *half_mod;
        const MulOp *q_mod;
        u32 L, k, log_n, size;
        u64 last_stride2; // vectors between consecutive `last` polys
        u64 in_stride2;   // vectors between consecutive input ciphertexts
        u64 out_stride2;  // vectors between consecutive output ciphertexts
        u64 total2;       // polys*(L-1)*N/2
    };

    // vector offset of limb li of poly p (= b*size + s) in the input / output batch
    __device__ __forceinline__ u64 rs_in(const RescaleArgs &g, u64 p, u32 li)
    {
        return (p / g.size) * g.in_stride2 + (((p % g.size) * g.L + li) << (g.log_n - 1));
    }
    __device__ __forceinline__ u64 rs_out(const RescaleArgs &g, u64 p, u32 li)
    {
        return (p / g.size) * g.out_stride2 + (((p % g.size) * (g.L - 1) + li) << (g.log_n - 1));
    }

    // CKKS: tmp[p][i] = ((last + half mod q_last) mod q_i) + (q_i - half mod q_i)       rns.cpp:857-878
    __global__ __launch_bounds__(256) void rescale_prepare_kernel(RescaleArgs g)
    {
        const u64 stride = (u64)gridDim.x * blockDim.x;
        const u32 n2 = 1u << (g.log_n - 1);
        const DevMod ql = g.mods[g.L - 1];
        const u64 half = ql.q >> 1;
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < g.total2; i += stride)
        {
            const u32 c = (u32)(i & (n2 - 1));
            const u64 r = i >> (g.log_n - 1);
            const u32 li = (u32)(r % (g.L - 1));
            const u64 p = r / (g.L - 1);
            const DevMod m = g.mods[li];
            ulonglong2 v = reinterpret_cast<const ulonglong2 *>(g.last)[p * g.last_stride2 + c];
            v.x = add_mod(v.x, half, ql.q);
            v.y = add_mod(v.y, half, ql.q);
            if (m.q < ql.q)
            {
                v.x = barrett64(v.x, m.q, m.ratio1);
                v.y = barrett64(v.y, m.q, m.ratio1);
            }
            const u64 neg_half = m.q - g.half_mod[(g.L - 1) * g.k + li];
            v.x += neg_half;
            v.y += neg_half;
            reinterpret_cast<ulonglong2 *>(g.tmp)[i] = v;
        }
    }

    // BGV: tmp[p][i] = delta_i(last), last = INTT (canonical) of the last limb            rns.cpp:1211-1237
    // (a thread owns two coefficients of a polynomial and walks over the limbs: k is formed once, see tail_delta_bgv_kernel)
    __global__ __launch_bounds__(256) void rescale_prepare_bgv_kernel(RescaleArgs g, BgvArgs pt)
    {
        const u64 stride = (u64)gridDim.x * blockDim.x;
        const u32 n2 = 1u << (g.log_n - 1), dl = g.L - 1;
        const u64 pairs = g.total2 / dl;
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += stride)
        {
            const u32 c = (u32)(i & (n2 - 1));
            const u64 p = i >> (g.log_n - 1);
            const ulonglong2 v = reinterpret_cast<const ulonglong2 *>(g.last)[p * g.last_stride2 + c];
            const u64 kx = bgv_k(v.x, pt), ky = bgv_k(v.y, pt);
            ulonglong2 *o = reinterpret_cast<ulonglong2 *>(g.tmp) + ((p * dl) << (g.log_n - 1)) + c;
            for (u32 li = 0; li < dl; ++li)
            {
                const DevMod m = g.mods[li];
                const MulOp Qm = g.q_mod[dl * g.k + li];
                ulonglong2 r;
                r.x = bgv_delta_k(v.x, kx, m, Qm);
                r.y = bgv_delta_k(v.y, ky, m, Qm);
                o[(u64)li << (g.log_n - 1)] = r;
            }
        }
    }

    // CKKS: out[p][i] = (in[p][i] + 4 q_i - tmp[p][i]) * q_last^-1 mod q_i              rns.cpp:879-899
    __global__ __launch_bounds__(256) void rescale_finish_kernel(RescaleArgs g)
    {
        const u64 stride = (u64)gridDim.x * blockDim.x;
        const u32 n2 = 1u << (g.log_n - 1);
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < g.total2; i += stride)
        {
            const u32 c = (u32)(i & (n2 - 1));
            const u64 r = i >> (g.log_n - 1);
            const u32 li = (u32)(r % (g.L - 1));
            const u64 p = r / (g.L - 1);
            const u64 q = g.mods[li].q;
            const MulOp inv = g.inv_mod[(g.L - 1) * g.k + li];
            const ulonglong2 t = reinterpret_cast<const ulonglong2 *>(g.tmp)[i];
            const ulonglong2 x = reinterpret_cast<const ulonglong2 *>(g.in)[rs_in(g, p, li) + c];
            ulonglong2 o;
            o.x = mul_shoup(x.x + (q << 2) - t.x, inv.op, inv.quo, q);
            o.y = mul_shoup(x.y + (q << 2) - t.y, inv.op, inv.quo, q);
            reinterpret_cast<ulonglong2 *>(g.out)[rs_out(g, p, li) + c] = o;
        }
    }

    // BFV (coefficient domain), rns.cpp:805-827: one pass
    __global__ __launch_bounds__(256) void rescale_coeff_kernel(RescaleArgs g)
    {
        const u64 stride = (u64)gridDim.x * blockDim.x;
        const u32 n2 = 1u << (g.log_n - 1);
        const DevMod ql = g.mods[g.L - 1];
        const u64 half = ql.q >> 1;
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < g.total2; i += stride)
        {
            const u32 c = (u32)(i & (n2 - 1));
            const u64 r = i >> (g.log_n - 1);
            const u32 li = (u32)(r % (g.L - 1));
            const u64 p = r / (g.L - 1);
            const DevMod m = g.mods[li];
            const MulOp inv = g.inv_mod[(g.L - 1) * g.k + li];
            const u64 half_mod = g.half_mod[(g.L - 1) * g.k + li];
            ulonglong2 v = reinterpret_cast<const ulonglong2 *>(g.in)[rs_in(g, p, g.L - 1) + c];
            const ulonglong2 x = reinterpret_cast<const ulonglong2 *>(g.in)[rs_in(g, p, li) + c];
            ulonglong2 o;
            u64 t = sub_mod(barrett64(add_mod(v.x, half, ql.q), m.q, m.ratio1), half_mod, m.q);
            o.x = mul_shoup(sub_mod(x.x, t, m.q), inv.op, inv.quo, m.q);
            t = sub_mod(barrett64(add_mod(v.y, half, ql.q), m.q, m.ratio1), half_mod, m.q);
            o.y = mul_shoup(sub_mod(x.y, t, m.q), inv.op, inv.quo, m.q);
            reinterpret_cast<ulonglong2 *>(g.out)[rs_out(g, p, li) + c] = o;
        }
    }

    // gather the last limb of every poly: last[p][N] = in[p][L-1][N]
    __global__ __launch_bounds__(256) void gather_last_kernel(RescaleArgs g, u64 *last, u64 total2)
    {
        const u64 stride = (u64)gridDim.x * blockDim.x;
        const u32 n2 = 1u << (g.log_n - 1);
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < total2; i += stride)
        {
            const u32 c = (u32)(i & (n2 - 1));
            const u64 p = i >> (g.log_n - 1);
            reinterpret_cast<ulonglong2 *>(last)[i] = reinterpret_cast<const ulonglong2 *>(g.in)[rs_in(g, p, g.L - 1) + c];
        }
    }

    BgvArgs bgv_args(const cat_ctx *ctx, u32 q_index)
    {
        BgvArgs a;
        a.t = ctx->pt.q;
        a.t_ratio1 = ctx->pt.ratio1;
        a.inv_Q = MulOp{ ctx->inv_q_mod_t[q_index], ctx->pt.shoup(ctx->inv_q_mod_t[q_index]) };
        return a;
    }

    u32 grid_for(u64 items)
    {
        u64 blocks = (items + 255) / 256;
        if (blocks > 256 * 16)
            blocks = 256 * 16;
        if (blocks == 0)
            blocks = 1;
        return (u32)blocks;
    }
} // namespace

namespace cat
{
    // Step 4 of switch_key_inplace (evaluator.cpp:2714-2818): the mod-down of prod [batch][2][d+1][N] (NTT form) into
    // enc [batch][2][d][N], every branch (CKKS, BFV, BGV); with rescale_out the relinearize + rescale tail.  tmp is
    // [batch][2][d][N] words of scratch, last [batch][2][2][N] (rescale_out only).  Shared by launch_switch_key and the
    // hoisted rotations.
    static cat_status mod_down(cat_ctx *ctx, u64 *prod, u64 *tmp, u64 *last, u64 *enc, size_t enc_stride, uint32_t L,
                               uint32_t batch, hipStream_t s, u64 *rescale_out, size_t out_stride)
    {
        const u32 d = L, k = ctx->k, rns = d + 1, log_n = ctx->log_n;
        const size_t N = ctx->n;
        const bool bgv = ctx->scheme == CAT_SCHEME_BGV;
        const bool ckks = ctx->scheme == CAT_SCHEME_CKKS || bgv;
        const size_t w_tmp = (size_t)batch * 2 * d * N;
        const LimbMap ident = ctx->identity_map();
        LimbMap mp{};
        mp.id[0] = (unsigned char)(k - 1);
        // the BGV branch transforms back fully (:2724), the others lazily (:2763)
        // (with rescale_out the special limb is transformed together with the last limb, below)
        cat_status st = CAT_OK;
        if (!rescale_out && ckks) // (BFV transforms all d + 1 limbs back in one launch, below)
            st = launch_ntt_ex(ctx, false, prod + (size_t)d * N, mp, 1, batch * 2, !bgv, (u64)rns * N, 1, false, s);
        if (st != CAT_OK)
            return st;
        DownArgs g;
        g.prod = prod;
        g.tmp = tmp;
        g.enc = enc;
        g.mods = ctx->d_mods;
        g.inv_mod = ctx->d_inv_mod;
        g.half_mod = ctx->d_half_mod;
        g.q_mod = ctx->d_q_mod;
        g.d = d;
        g.k = k;
        g.log_n = log_n;
        g.lazy_shift = ckks ? 2 : 1;
        g.enc_stride2 = enc_stride / 2;
        g.total2 = w_tmp / 2;
        if (bgv && !rescale_out)
        {
            // delta (canonical) by its own kernel, then its forward transform; the subtraction, the product with
            // P^-1 and the accumulation (:2749-2756) are the same finish step as CKKS, on a lazy transform
            hipLaunchKernelGGL(ks_down_prepare_bgv_kernel, dim3(grid_for(g.total2 / d)), dim3(256), 0, s, g,
                               bgv_args(ctx, k - 1));
            CAT_HIP(hipGetLastError());
            if (log_n >= 12)
            {
                NttFuse f;
                f.ep_other = prod;
                f.ep_other_b = (u64)2 * rns * N;
                f.ep_other_s = (u64)rns * N;
                f.ep_dest = enc;
                f.ep_dest_b = enc_stride;
                f.ep_dest_s = (u64)d * N;
                f.ep_S = 2;
                f.ep_acc = 1;
                f.ep_qidx = k - 1;
                return launch_ntt_ex(ctx, true, tmp, ident, d, batch * 2, true, (u64)d * N, 1, false, s, false, &f);
            }
            st = launch_ntt(ctx, true, tmp, ident, d, batch * 2, true, s);
            if (st != CAT_OK)
                return st;
            hipLaunchKernelGGL(ks_down_finish_kernel<false>, dim3(grid_for(g.total2)), dim3(256), 0, s, g);
            CAT_HIP(hipGetLastError());
            return CAT_OK;
        }
        if (rescale_out)
        {
            // relinearize + rescale_to_next with ONE forward transform per remaining limb.  Sequentially
            //   enc'[j] = enc[j] + (prod[j] - NTT(r1_j)) P^-1,   out[j] = (enc'[j] - NTT(r2_j)) q_last^-1,
            // r1 = the mod-down round source (from INTT_P of the special limb), r2 = the rescale round source
            // (from INTT of enc'[d-1], util/rns.cpp:857-878); the transform is linear, so
            //   out[j] = (enc[j] + prod[j] P^-1 - NTT(r1_j P^-1 + r2_j)) q_last^-1   (mod q_j, canonical).
            // The last limb first: the rescale needs INTT(enc'[d-1]), and
            //   INTT(enc'[d-1]) = INTT(enc[d-1] + prod[d-1] P^-1) - r1_{d-1} P^-1
            // (the forward transform of the correction and this inverse transform cancel), so one batched inverse
            // transform over {special limb, u = enc[d-1] + prod[d-1] P^-1} feeds both round sources.
            TailArgs ta;
            ta.prod = prod;
            ta.enc = enc;
            ta.buf = last;
            ta.mods = ctx->d_mods;
            ta.inv_mod = ctx->d_inv_mod;
            ta.half_mod = ctx->d_half_mod;
            ta.q_mod = ctx->d_q_mod;
            ta.d = d;
            ta.k = k;
            ta.log_n = log_n;
            ta.enc_stride2 = enc_stride / 2;
            ta.total2 = (u64)batch * 2 * N / 2;
            hipLaunchKernelGGL(tail_prepare_kernel, dim3(grid_for(ta.total2)), dim3(256), 0, s, ta);
            CAT_HIP(hipGetLastError());
            LimbMap m2{};
            m2.id[0] = (unsigned char)(k - 1);
            m2.id[1] = (unsigned char)(d - 1);
            st = launch_ntt(ctx, false, last, m2, 2, batch * 2, false, s);
            if (st != CAT_OK)
                return st;
            NttFuse g2;
            if (bgv)
            {
                // the same identities with the BGV corrections (bgv_delta; evaluator.cpp:2715-2758,
                // util/rns.cpp:1203-1246): r1_j = delta_j of the special-prime product, r2_j = delta_j of the
                // relinearized last limb.  Their sum is written out by its own kernel (the mod-t step does not fit
                // the round source), the forward transform keeps the two-term epilogue.
                const BgvArgs ptP = bgv_args(ctx, k - 1), ptL = bgv_args(ctx, d - 1);
                hipLaunchKernelGGL(tail_last_bgv_kernel, dim3(grid_for(ta.total2)), dim3(256), 0, s, ta, ptP);
                CAT_HIP(hipGetLastError());
                const u64 total2 = (u64)batch * 2 * (d - 1) * N / 2;
                hipLaunchKernelGGL(tail_delta_bgv_kernel, dim3(grid_for(total2 / (d - 1))), dim3(256), 0, s, ta, ptP, ptL, tmp,
                                   total2);
                CAT_HIP(hipGetLastError());
            }
            else
            {
                hipLaunchKernelGGL(tail_last_kernel, dim3(grid_for(ta.total2)), dim3(256), 0, s, ta);
                CAT_HIP(hipGetLastError());
                // (c) the remaining d-1 limbs: dual source, two-term epilogue
                g2.src = last;
                g2.src_stride = (u64)2 * N;
                g2.src_kind = 1;
                g2.src_qidx = k - 1;
                g2.src2 = last + N;
                g2.src2_stride = (u64)2 * N;
                g2.src2_qidx = d - 1;
            }
            g2.ep_other = enc;
            g2.ep_other_b = enc_stride;
            g2.ep_other_s = (u64)d * N;
            g2.ep_other2 = prod;
            g2.ep_other2_b = (u64)2 * rns * N;
            g2.ep_other2_s = (u64)rns * N;
            g2.ep_qidx2 = k - 1;
            g2.ep_dest = rescale_out;
            g2.ep_dest_b = out_stride;
            g2.ep_dest_s = (u64)(d - 1) * N;
            g2.ep_S = 2;
            g2.ep_acc = 0;
            g2.ep_qidx = d - 1;
            return launch_ntt_ex(ctx, true, tmp, ident, d - 1, batch * 2, true, (u64)(d - 1) * N, 1, false, s, false, &g2);
        }
        if (ckks && log_n >= 12)
        {
            // prepare and finish fused into the two passes of the forward transform of tmp (NttFuse)
            NttFuse f;
            f.src = prod + (size_t)d * N;
            f.src_stride = (u64)rns * N;
            f.src_kind = 1;
            f.src_qidx = k - 1;
            f.ep_other = prod;
            f.ep_other_b = (u64)2 * rns * N;
            f.ep_other_s = (u64)rns * N;
            f.ep_dest = enc;
            f.ep_dest_b = enc_stride;
            f.ep_dest_s = (u64)d * N;
            f.ep_S = 2;
            f.ep_acc = 1;
            f.ep_qidx = k - 1;
            return launch_ntt_ex(ctx, true, tmp, ident, d, batch * 2, true, (u64)d * N, 1, false, s, false, &f);
        }
        if (!ckks)
        {
            // BFV: the whole product, special limb included, goes back to coefficient form (lazy, :2763) in one launch and
            // the correction is applied there
            LimbMap ext{};
            for (u32 I = 0; I < rns; ++I)
                ext.id[I] = (unsigned char)(I == d ? k - 1 : I);
            st = launch_ntt_ex(ctx, false, prod, ext, rns, batch * 2, true, (u64)rns * N, 1, false, s);
            if (st != CAT_OK)
                return st;
            hipLaunchKernelGGL(ks_down_finish_kernel<true>, dim3(grid_for(g.total2)), dim3(256), 0, s, g);
            CAT_HIP(hipGetLastError());
            return CAT_OK;
        }
        hipLaunchKernelGGL(ks_down_prepare_kernel, dim3(grid_for(g.total2)), dim3(256), 0, s, g);
        CAT_HIP(hipGetLastError());
        st = launch_ntt(ctx, true, tmp, ident, d, batch * 2, true, s);
        if (st != CAT_OK)
            return st;
        hipLaunchKernelGGL(ks_down_finish_kernel<false>, dim3(grid_for(g.total2)), dim3(256), 0, s, g);
        CAT_HIP(hipGetLastError());

        return CAT_OK;
    }

    cat_status launch_switch_key(cat_ctx *ctx, u64 *enc, const u64 *target, const u64 *key, uint32_t L, uint32_t batch,
                                 size_t enc_stride, size_t target_stride, hipStream_t s, u64 *rescale_out, size_t out_stride)
    {
        const u32 d = L, k = ctx->k, rns = d + 1, log_n = ctx->log_n;
        const size_t N = ctx->n;
        const bool bgv = ctx->scheme == CAT_SCHEME_BGV;
        const bool ckks = ctx->scheme == CAT_SCHEME_CKKS || bgv; // NTT-form ciphertexts (:2608, :2635)
        const size_t w_target = (size_t)batch * d * N;
        const size_t w_ext = (size_t)batch * rns * d * N;
        const size_t w_prod = (size_t)batch * 2 * rns * N;
        const size_t w_tmp = (size_t)batch * 2 * d * N;
        const size_t w_last = rescale_out ? (size_t)batch * 2 * 2 * N : 0;
        static_assert(sizeof(u64) == 8, "");
        if (w_target + w_ext + w_prod + w_tmp + w_last != ws_words_switch_key(N, d, batch, rescale_out != nullptr))
            return fail(CAT_ERR_LOGIC, "workspace formula out of step");
        u64 *ws = (u64 *)workspace(ctx, ws_words_switch_key(N, d, batch, rescale_out != nullptr) * sizeof(u64));
        if (!ws)
            return workspace_status(ctx);
        u64 *t_target = ws, *ext = t_target + w_target, *prod = ext + w_ext, *tmp = prod + w_prod, *last = tmp + w_tmp;
        if (rescale_out && (!ckks || log_n < 12 || d < 2))
            return fail(CAT_ERR_LOGIC, "unsupported fused transform");
        const LimbMap ident = ctx->identity_map();

        // step 1
        const u64 *coeff_target = target;
        if (ckks)
        {
            cat_status st;
            if (log_n >= 12)
            {
                // out of place: the first inverse pass reads the target where it lies (it stays in NTT form for the
                // I == J terms) and writes t_target
                NttFuse in;
                in.src = target;
                in.src_stride = target_stride;
                st = launch_ntt_ex(ctx, false, t_target, ident, d, batch, false, (u64)d * N, 1, false, s, false, &in);
            }
            else
            {
                CAT_HIP(hipMemcpy2DAsync(t_target, d * N * sizeof(u64), target, target_stride * sizeof(u64),
                                         d * N * sizeof(u64), batch, hipMemcpyDeviceToDevice, s));
                st = launch_ntt(ctx, false, t_target, ident, d, batch, false, s);
            }
            if (st != CAT_OK)
                return st;
            coeff_target = t_target;
        }
        // step 2: modulus raise fused into the first NTT pass (ntt.hip Source): ext is written once, already
        // half transformed, instead of being written by a raise kernel and re-read
        const bool fused_mac = log_n >= 12;
        int wide = 0;
        {
            LimbMap map{};
            for (u32 I = 0; I < rns; ++I)
                map.id[I] = (unsigned char)(I == d ? k - 1 : I);
            wide = ntt_wide(ctx, map, rns); // the same predicate launch_ntt_ex applies to this map
            NttFuse raise;
            raise.src = coeff_target;
            raise.src_stride = ckks ? (u64)d * N : (u64)target_stride;
            raise.first_pass_only = fused_mac;
            cat_status st = launch_ntt_ex(ctx, true, ext, map, rns * d, batch, true, (u64)rns * d * N, d, ckks, s, false,
                                          &raise);
            if (st != CAT_OK)
                return st;
        }
        // step 3 (N >= 4096: together with the second NTT pass)
        if (fused_mac)
        {
            RowsMacArgs g;
            g.ext = ext;
            g.target = target;
            g.key = key;
            g.prod = prod;
            g.mods = ctx->d_mods;
            g.nmods = ctx->d_nttmods;
            g.tw3 = ctx->d_tw3_fwd;
            g.d = d;
            g.k = k;
            g.log_n = log_n;
            g.batch = batch;
            g.use_diag = ckks ? 1 : 0;
            g.target_stride = target_stride;
            g.groups = (u32)((u64)rns * (N / (RM_ROWS * 256)));
            // few digits (the 48 KB of staged twiddles weigh as much as a workgroup's own data: d = 3, N = 8192, batch 256
            // gains 2.8 %, d = 8 0.5 %, d >= 15 nothing or less) and far more workgroups than the card holds (512): two or
            // four ciphertexts per workgroup share its staged twiddles, as long as about four rounds of workgroups remain
            u32 nb = 1;
            while (d <= 8 && nb < 4 && (u64)g.groups * batch / (2 * nb) >= 2048 && 2 * nb <= batch)
                nb *= 2;
            g.nb = nb;
            const u64 blocks = (u64)((g.groups + 7) / 8) * 8 * ((batch + nb - 1) / nb);
            if (blocks > 0x7fffffffu)
                return fail(CAT_ERR_LOGIC, "invalid parameters");
            if (!wide)
                return fail(CAT_ERR_LOGIC, "invalid parameters"); // unreachable: scheme contexts hold primes < 2^60
            if (nb > 1 && wide == 2)
                hipLaunchKernelGGL((ks_rows_mac_kernel<2, true>), dim3((u32)blocks), dim3(256), 0, s, g);
            else if (nb > 1)
                hipLaunchKernelGGL((ks_rows_mac_kernel<1, true>), dim3((u32)blocks), dim3(256), 0, s, g);
            else if (wide == 2)
                hipLaunchKernelGGL((ks_rows_mac_kernel<2, false>), dim3((u32)blocks), dim3(256), 0, s, g);
            else
                hipLaunchKernelGGL((ks_rows_mac_kernel<1, false>), dim3((u32)blocks), dim3(256), 0, s, g);
            CAT_HIP(hipGetLastError());
        }
        else
        {
            MacArgs g;
            g.ext = ext;
            g.target = target;
            g.key = key;
            g.prod = prod;
            g.mods = ctx->d_mods;
            g.d = d;
            g.k = k;
            g.log_n = log_n;
            g.use_diag = ckks ? 1 : 0;
            g.target_stride = target_stride;
            g.batch = batch;
            constexpr int BB = 4;
            g.total = (u64)((batch + BB - 1) / BB) * rns * (N / 2);
            hipLaunchKernelGGL(ks_mac_kernel<BB>, dim3((u32)((g.total + 255) / 256)), dim3(256), 0, s, g);
            CAT_HIP(hipGetLastError());
        }
        // step 4
        return mod_down(ctx, prod, tmp, last, enc, enc_stride, L, batch, s, rescale_out, out_stride);
    }

    // Hoisted rotations (second half; no counterpart in the reference, whose rotate_internal runs one full
    // switch_key_inplace per Galois element, evaluator.cpp:2466-2521).  Several Galois elements g_1 .. g_r applied to the
    // SAME ciphertexts share everything that does not depend on the element: the inverse transform of c1, the modulus
    // raise and the d (d + 1) forward transforms of the extended digits -- all but a few per cent of a key switch's
    // arithmetic.  Per element there remain the inner product with that element's key, one permutation of the 2 (d + 1)
    // product limbs and the mod-down:
    //     sigma_g(ext_J) * key_J  =  sigma_g( ext_J * sigma_g^-1(key_J) )           (sigma_g permutes NTT-domain values)
    // so the inner product runs on the extended digits as they lie against the PREPARED key sigma_g^-1(key)
    // (cat_hoisted_key_prepare, once per key), and its result is permuted afterwards.
    // Not bit-identical to the reference's un-hoisted rotation, and it cannot be: the reference decomposes sigma_g(c1),
    // whose negated coefficients have the canonical digits q_J - t, and the residue of q_J - t modulo another prime is not
    // minus the residue of t; here the digits of c1 are permuted after the decomposition.  Both are key switches of the
    // same polynomial with digits below q_J (same noise bound); the results decrypt to the same rotated message.
    cat_status launch_rotate_hoisted(cat_ctx *ctx, const u64 *ct, const uint32_t *elts, const u64 *const *keys, uint32_t n_rot,
                                     u64 *const *outs, uint32_t L, uint32_t batch, hipStream_t s)
    {
        const u32 d = L, k = ctx->k, rns = d + 1, log_n = ctx->log_n;
        const size_t N = ctx->n, ctw = (size_t)2 * d * N;
        const bool ckks = ctx->scheme == CAT_SCHEME_CKKS || ctx->scheme == CAT_SCHEME_BGV; // NTT-form ciphertexts
        const size_t w_target = (size_t)batch * d * N, w_ext = (size_t)batch * rns * d * N, w_prod = (size_t)batch * 2 * rns * N,
                     w_tmp = (size_t)batch * 2 * d * N;
        if (w_target + w_ext + 2 * w_prod + 2 * w_tmp != ws_words_rotate_hoisted(N, d, batch))
            return fail(CAT_ERR_LOGIC, "workspace formula out of step");
        u64 *ws = (u64 *)workspace(ctx, ws_words_rotate_hoisted(N, d, batch) * sizeof(u64));
        if (!ws)
            return workspace_status(ctx);
        u64 *t_target = ws, *ext = t_target + w_target, *prod = ext + w_ext, *prod2 = prod + w_prod, *tmp = prod2 + w_prod,
            *perm = tmp + w_tmp;
        const LimbMap ident = ctx->identity_map();
        const u64 *c1 = ct + (size_t)d * N;
        // shared part: steps 1 and 2 of switch_key_inplace on c1 as it lies, every extended digit fully transformed
        const u64 *coeff_target = c1;
        size_t coeff_stride = ctw;
        if (ckks)
        {
            cat_status st;
            if (log_n >= 12)
            {
                NttFuse in;
                in.src = c1;
                in.src_stride = ctw;
                st = launch_ntt_ex(ctx, false, t_target, ident, d, batch, false, (u64)d * N, 1, false, s, false, &in);
            }
            else
            {
                CAT_HIP(hipMemcpy2DAsync(t_target, d * N * sizeof(u64), c1, ctw * sizeof(u64), d * N * sizeof(u64), batch,
                                         hipMemcpyDeviceToDevice, s));
                st = launch_ntt(ctx, false, t_target, ident, d, batch, false, s);
            }
            if (st != CAT_OK)
                return st;
            coeff_target = t_target;
            coeff_stride = (size_t)d * N;
        }
        {
            LimbMap map{};
            for (u32 I = 0; I < rns; ++I)
                map.id[I] = (unsigned char)(I == d ? k - 1 : I);
            NttFuse raise;
            raise.src = coeff_target;
            raise.src_stride = coeff_stride;
            const cat_status st = launch_ntt_ex(ctx, true, ext, map, rns * d, batch, true, (u64)rns * d * N, d, ckks, s, false, &raise);
            if (st != CAT_OK)
                return st;
        }
        for (uint32_t r = 0; r < n_rot; ++r)
        {
            MacArgs g;
            g.ext = ext;
            g.target = c1; // the I == J operands (NTT-form schemes): the original limb, as :2635-2638
            g.key = keys[r];
            g.prod = prod;
            g.mods = ctx->d_mods;
            g.d = d;
            g.k = k;
            g.log_n = log_n;
            g.use_diag = ckks ? 1 : 0;
            g.target_stride = ctw;
            g.batch = batch;
            constexpr int BB = 4;
            g.total = (u64)((batch + BB - 1) / BB) * rns * (N / 2);
            if (g.total > 0x7fffffffull * 256)
                return fail(CAT_ERR_LOGIC, "invalid parameters");
            hipLaunchKernelGGL(ks_mac_kernel<BB>, dim3((u32)((g.total + 255) / 256)), dim3(256), 0, s, g);
            CAT_HIP(hipGetLastError());
            cat_status st = launch_galois(ctx, true, prod, prod2, elts[r], ident, rns, batch * 2, s);
            if (st != CAT_OK)
                return st;
            // (sigma_g(c0), 0) as apply_galois_inplace leaves it (:2429-2449), then the mod-down adds the switched part
            u64 *out_r = outs[r];
            if ((st = launch_galois(ctx, ckks, ct, perm, elts[r], ident, d, batch * 2, s)) != CAT_OK)
                return st;
            if ((st = launch_copy_c0_zero_c1(ctx, perm, out_r, d, batch, s)) != CAT_OK)
                return st;
            if ((st = mod_down(ctx, prod2, tmp, nullptr, out_r, ctw, d, batch, s, nullptr, 0)) != CAT_OK)
                return st;
        }
        return CAT_OK;
    }

    cat_status launch_rescale(cat_ctx *ctx, const u64 *in, u64 *out, uint32_t L, uint32_t size, uint32_t batch,
                              size_t in_stride, size_t out_stride, hipStream_t s, u64 *scratch, size_t scratch_bytes)
    {
        const size_t N = ctx->n;
        const u64 polys = (u64)batch * size;
        const size_t w_last = (size_t)polys * N, w_tmp = (size_t)polys * (L - 1) * N;
        // outputs overlap not-yet-read inputs whenever the two batches share memory
        const u64 *in_end = in + (size_t)(batch ? batch - 1 : 0) * in_stride + (size_t)size * L * N;
        const u64 *out_end = out + (size_t)(batch ? batch - 1 : 0) * out_stride + (size_t)size * (L - 1) * N;
        const bool alias = !(out_end <= in || in_end <= out);
        const bool ckks = ctx->scheme != CAT_SCHEME_BFV;
        const bool bgv = ctx->scheme == CAT_SCHEME_BGV;
        u64 *ws = scratch;
        if (scratch)
        {
            if (alias)
                return fail(CAT_ERR_INVALID_ARG, "destination must not alias the inputs");
            if ((w_last + w_tmp) * sizeof(u64) > scratch_bytes)
                return fail(CAT_ERR_INVALID_ARG, "operands exceed the static scratch buffer");
        }
        else
            ws = (u64 *)workspace(ctx, ws_words_rescale(N, L, polys) * sizeof(u64)); // = w_last + 2 * w_tmp
        if (!ws)
            return workspace_status(ctx);
        u64 *last = ws, *tmp = last + w_last, *stage = tmp + w_tmp; // stage is used only when alias (never with scratch)
        RescaleArgs g;
        g.in = in;
        g.last = last;
        g.tmp = tmp;
        g.out = alias ? stage : out;
        g.mods = ctx->d_mods;
        g.inv_mod = ctx->d_inv_mod;
        g.half_mod = ctx->d_half_mod;
        g.q_mod = ctx->d_q_mod;
        g.L = L;
        g.k = ctx->k;
        g.log_n = ctx->log_n;
        g.size = size;
        g.in_stride2 = in_stride / 2;
        g.out_stride2 = alias ? (u64)size * (L - 1) * N / 2 : out_stride / 2;
        g.last_stride2 = N / 2;
        g.total2 = w_tmp / 2;
        if (g.total2 == 0)
            return CAT_OK;
        if (ckks)
        {
            hipLaunchKernelGGL(gather_last_kernel, dim3(grid_for(w_last / 2)), dim3(256), 0, s, g, last,
                               (u64)w_last / 2);
            CAT_HIP(hipGetLastError());
            LimbMap ml{};
            ml.id[0] = (unsigned char)(L - 1);
            cat_status st = launch_ntt(ctx, false, last, ml, 1, (uint32_t)polys, false, s);
            if (st != CAT_OK)
                return st;
            if (bgv)
            {
                hipLaunchKernelGGL(rescale_prepare_bgv_kernel, dim3(grid_for(g.total2 / (L - 1))), dim3(256), 0, s, g,
                                   bgv_args(ctx, L - 1));
                CAT_HIP(hipGetLastError());
            }
            if (ctx->log_n >= 12)
            {
                NttFuse f;
                if (!bgv)
                {
                    f.src = last;
                    f.src_stride = N;
                    f.src_kind = 1;
                    f.src_qidx = L - 1;
                }
                f.ep_other = in;
                f.ep_other_b = in_stride;
                f.ep_other_s = (u64)L * N;
                f.ep_dest = g.out;
                f.ep_dest_b = g.out_stride2 * 2;
                f.ep_dest_s = (u64)(L - 1) * N;
                f.ep_S = size;
                f.ep_acc = 0;
                f.ep_qidx = L - 1;
                st = launch_ntt_ex(ctx, true, tmp, ctx->identity_map(), L - 1, (uint32_t)polys, true, (u64)(L - 1) * N, 1,
                                   false, s, false, &f);
                if (st != CAT_OK)
                    return st;
                if (alias)
                    CAT_HIP(hipMemcpy2DAsync(out, out_stride * sizeof(u64), stage,
                                             (size_t)size * (L - 1) * N * sizeof(u64),
                                             (size_t)size * (L - 1) * N * sizeof(u64), batch, hipMemcpyDeviceToDevice, s));
                return CAT_OK;
            }
            if (!bgv)
            {
                hipLaunchKernelGGL(rescale_prepare_kernel, dim3(grid_for(g.total2)), dim3(256), 0, s, g);
                CAT_HIP(hipGetLastError());
            }
            st = launch_ntt(ctx, true, tmp, ctx->identity_map(), L - 1, (uint32_t)polys, true, s);
            if (st != CAT_OK)
                return st;
            hipLaunchKernelGGL(rescale_finish_kernel, dim3(grid_for(g.total2)), dim3(256), 0, s, g);
            CAT_HIP(hipGetLastError());
        }
        else
        {
            hipLaunchKernelGGL(rescale_coeff_kernel, dim3(grid_for(g.total2)), dim3(256), 0, s, g);
            CAT_HIP(hipGetLastError());
        }
        if (alias)
            CAT_HIP(hipMemcpy2DAsync(out, out_stride * sizeof(u64), stage, (size_t)size * (L - 1) * N * sizeof(u64),
                                     (size_t)size * (L - 1) * N * sizeof(u64), batch, hipMemcpyDeviceToDevice, s));
        return CAT_OK;
    }
} // namespace cat
