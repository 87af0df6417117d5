// BFV BEHZ multiplication (Evaluator::bfv_multiply, evaluator.cpp:395-567) and the RNSTool pieces it is
// made of (util/rns.cpp: BaseConverter::fast_convert_array :418-463, RNSTool::initialize :578-787,
// fastbconv_sk :903-979, sm_mrq :981-1041, fast_floor :1043-1090, fastbconv_m_tilde :1092-1131).
//
// Bases of a level with L limbs (rns.cpp:607-635): q = primes[0..L); the 61-bit auxiliary primes
// get_primes(2N, 61, |B| + 2) = {m_sk, gamma, B_0, B_1, ...}; Bsk = B U {m_sk}; m_tilde = 2^32.
// The context holds the descending 61-bit list once (moduli k.., capi.cpp), so every level's bases are
// prefixes of it: m_sk = modulus k, B_i = modulus k + 2 + i.
//
// All conversions are per-coefficient small dense mat-vecs: one thread per coefficient stages the
// |ibase| scaled residues x_i * (q/q_i)^-1 mod q_i in LDS ([limb][thread], conflict-free) and then forms
// each output residue as a 128-bit dot product with one Barrett-128 (the reference reduces every 16
// terms, uintarithsmallmod.cpp:110-173; the canonical result is the same).
#include "context.hpp"

#include <cstring>
#include <type_traits>

namespace
{
    constexpr u64 MT = 1ull << 32; // m_tilde (rns.cpp:635)

    struct BconvArgs
    {
        const u64 *in;   // [polys][in_limbs_total][N]; the conversion reads limbs 0..isz-1
        u64 *out;        // [polys][out_limbs_total][N]; the conversion writes limbs out_off..out_off+osz-1
        const DevMod *mods;
        const MulOp *inv_punc; // [isz]: (base/b_i)^-1 mod b_i, times the caller's constant when there is one
        const u64 *matrix;    // [osz][isz]
        const u64 *mt_matrix; // [isz] or nullptr
        LimbMap imap, omap;
        u32 isz, osz, log_n;
        u64 in_stride, out_stride; // words between consecutive polys
        // output placement of poly p: (p / grp) * out_obj_stride + (p % grp) * out_stride (grp = 1: p * out_obj_stride)
        u32 grp;
        u64 out_obj_stride;
        u64 *copy_out;             // optional: the input limbs as read, same placement rule with the copy strides
        u64 copy_stride, copy_obj_stride;
        u32 out_off, mt_limb;
        u64 total; // polys * N
        // fused tail (template parameter EP of bconv_kernel)
        const MulOp *ep_c0, *ep_c1; // EP 3: B mod q_o, -B mod q_o per output limb
        const u64 *ep_in; // EP 2: the Bsk part of the input, [polys][..][N]
        u64 ep_in_stride;
        u64 ep_k0;        // EP 1: -q^-1 mod m_tilde
        MulOp ep_k1;      // EP 3: B^-1 mod m_sk
        u32 ep_msk_id;    // EP 3: modulus id of m_sk; its matrix row follows the osz output rows, its input limb the isz read
    };

    // EP: what happens to each converted residue v before it is stored -- the tails the reference runs as separate loops
    // over the same coefficients, each needing only values this thread already holds:
    //   0  out = v (and the residue modulo m_tilde when mt_matrix is given: fastbconv_m_tilde, rns.cpp:1092-1131)
    //   1  sm_mrq (rns.cpp:981-1041) on top of fastbconv_m_tilde: r = v_mt * (-q^-1) mod m_tilde, centred;
    //        out_i = (r * (q mod m_i) + v_i) * m_tilde^-1 mod m_i
    //   2  fast_floor tail (rns.cpp:1070-1088): out_i = (in_Bsk_i [* t] + (m_i - v_i)) * q^-1 mod m_i
    //   3  fastbconv_sk tail (rns.cpp:930-977): alpha = (v_sk + (m_sk - in_sk)) * B^-1 mod m_sk, centred;
    //        out_i = v_i + (alpha > m_sk/2 ? (m_sk - alpha) * (B mod q_i) : alpha * (-B mod q_i))      ep_c0 = B, ep_c1 = -B
    // Same values as the separate kernels (cat_sm_mrq keeps the standalone form); one write instead of write + read + write.
    // Every tail is affine in v, so it costs one more multiply-accumulate before the reduction, not multiplications after
    // it: EP 1 and 2 come with a matrix of isz + 1 columns that has the tail's factor folded in (build_level) and the
    // tail's own operand (r, in_Bsk_i) meets the extra column; EP 3 adds alpha x (+-B mod q_i).  Operand bounds of rm_mac:
    // r and alpha are below a 61-bit prime, in_Bsk_i is a residue below 2^62 (canonical or lazy); at most 31 + 1 terms.
    // W: coefficients per thread (1 or 2).  Two make every global access 16 bytes per lane and every matrix word (a scalar
    // load) serve two dot products.  This kernel (residues staged in LDS under a run-time index) serves more than 16 input
    // limbs with W = 1; up to 16 limbs take bconv_reg_kernel below.
    template <int W>
    __device__ __forceinline__ void ldw(u64 (&v)[W], const u64 *p)
    {
        if constexpr (W == 2)
        {
            const ulonglong2 t = *reinterpret_cast<const ulonglong2 *>(p);
            v[0] = t.x, v[1] = t.y;
        }
        else
            v[0] = *p;
    }
    template <int W>
    __device__ __forceinline__ void stw(u64 *p, const u64 (&v)[W])
    {
        if constexpr (W == 2)
            *reinterpret_cast<ulonglong2 *>(p) = make_ulonglong2(v[0], v[1]);
        else
            *p = v[0];
    }

    template <int EP, int W>
    __global__ __launch_bounds__(256) void bconv_kernel(BconvArgs g)
    {
        extern __shared__ __align__(16) u64 stage[]; // [isz][256][W]
        const u64 i = ((u64)blockIdx.x * 256 + threadIdx.x) * W;
        if (i >= g.total)
            return;
        const u32 c = (u32)(i & ((1u << g.log_n) - 1));
        const u64 p = i >> g.log_n;
        const u64 *src = g.in + p * g.in_stride + c;
        const u64 obj = p / g.grp, sub = p - obj * g.grp;
        u64 *cpy = g.copy_out ? g.copy_out + obj * g.copy_obj_stride + sub * g.copy_stride + c : nullptr;
        u64 *mine = stage + (size_t)threadIdx.x * W; // + j * 256 * W
        for (u32 j = 0; j < g.isz; ++j)
        {
            const u64 q = g.mods[g.imap.id[j]].q;
            u64 x[W];
            ldw<W>(x, src + ((u64)j << g.log_n));
            if (cpy)
                stw<W>(cpy + ((u64)j << g.log_n), x);
#pragma unroll
            for (int w = 0; w < W; ++w)
                x[w] = mul_shoup(x[w], g.inv_punc[j].op, g.inv_punc[j].quo, q);
            stw<W>(mine + j * 256 * W, x);
        }
        u64 *dst = g.out + obj * g.out_obj_stride + sub * g.out_stride + c;
        // one dot product of the staged residues with a matrix row, both coefficients
        constexpr bool XCOL = EP == 1 || EP == 2; // the matrix carries the tail's column
        const u32 ms = g.isz + (XCOL ? 1 : 0);    // words per matrix row
        u64 mt_r[W], alpha[W];
        bool alpha_neg[W];
#pragma unroll
        for (int w = 0; w < W; ++w)
            mt_r[w] = alpha[w] = 0, alpha_neg[w] = false;
        // the tail's term of output o: operand ex, multiplier eb
        auto extra = [&](u32 o, u64 q, const u64 *row, u64 (&ex)[W], u64 (&eb)[W]) {
            if (EP == 1)
            {
#pragma unroll
                for (int w = 0; w < W; ++w)
                {
                    u64 t = mt_r[w];
                    if (t >= (MT >> 1))
                        t += q - MT;
                    ex[w] = t;
                    eb[w] = row[g.isz];
                }
            }
            if (EP == 2)
            {
                ldw<W>(ex, g.ep_in + p * g.ep_in_stride + ((u64)o << g.log_n) + c);
#pragma unroll
                for (int w = 0; w < W; ++w)
                    eb[w] = row[g.isz];
            }
            if (EP == 3)
            {
#pragma unroll
                for (int w = 0; w < W; ++w)
                {
                    ex[w] = alpha[w];
                    eb[w] = alpha_neg[w] ? g.ep_c0[o].op : g.ep_c1[o].op;
                }
            }
        };
        // xo >= 0: with the tail's term of output xo
        auto dot = [&](const u64 *row, const DevMod &m, u64 (&v)[W], int xo = -1) {
            Acc acc[W];
#pragma unroll
            for (int w = 0; w < W; ++w)
                acc[w].lo = acc[w].hi = 0, acc[w].cc = 0;
            if (EP != 0 && xo >= 0)
            {
                u64 ex[W], eb[W];
                extra((u32)xo, m.q, row, ex, eb);
#pragma unroll
                for (int w = 0; w < W; ++w)
                {
                    u64 mid = 0;
                    rm_mac(acc[w], mid, ex[w], eb[w]);
                    rm_fold(acc[w], mid);
                }
            }
            u32 j = 0;
            // scaled residues and matrix entries are canonical modulo primes of at most 61 bits: rm_mac's bounds
            for (; j + 2 <= g.isz; j += 2)
            {
                u64 x0[W], x1[W];
                ldw<W>(x0, mine + j * 256 * W);
                ldw<W>(x1, mine + (j + 1) * 256 * W);
#pragma unroll
                for (int w = 0; w < W; ++w)
                {
                    u64 mid;
                    rm_mac2(acc[w], mid, x0[w], row[j], x1[w], row[j + 1]);
                    rm_fold(acc[w], mid);
                }
            }
            if (j < g.isz)
            {
                u64 x0[W];
                ldw<W>(x0, mine + j * 256 * W);
#pragma unroll
                for (int w = 0; w < W; ++w)
                {
                    u64 mid = 0;
                    rm_mac(acc[w], mid, x0[w], row[j]);
                    rm_fold(acc[w], mid);
                }
            }
#pragma unroll
            for (int w = 0; w < W; ++w)
                v[w] = reduce128(acc[w].lo, acc[w].hi + acc[w].cc, m.q, m.ratio1, m.r64_quo);
        };
        if (EP == 1)
        {
            u64 acc[W];
#pragma unroll
            for (int w = 0; w < W; ++w)
                acc[w] = 0;
            for (u32 j = 0; j < g.isz; ++j)
            {
                u64 x[W];
                ldw<W>(x, mine + j * 256 * W);
#pragma unroll
                for (int w = 0; w < W; ++w)
                    acc[w] += x[w] * g.mt_matrix[j];
            }
#pragma unroll
            for (int w = 0; w < W; ++w)
                mt_r[w] = ((acc[w] & (MT - 1)) * g.ep_k0) & (MT - 1);
        }
        if (EP == 3)
        {
            const DevMod m = g.mods[g.ep_msk_id];
            u64 conv_sk[W], in_sk[W];
            dot(g.matrix + (u64)g.osz * ms, m, conv_sk);
            ldw<W>(in_sk, src + ((u64)g.isz << g.log_n));
#pragma unroll
            for (int w = 0; w < W; ++w)
            {
                alpha[w] = mul_shoup(conv_sk[w] + (m.q - in_sk[w]), g.ep_k1.op, g.ep_k1.quo, m.q);
                alpha_neg[w] = alpha[w] > (m.q >> 1);
                if (alpha_neg[w])
                    alpha[w] = m.q - alpha[w];
            }
        }
        auto finish_store = [&](u32 o, u64 (&v)[W]) { stw<W>(dst + ((u64)(g.out_off + o) << g.log_n), v); };
        u32 o = 0;
        // four output residues at a time: one read of the staged residues serves four dot products, and their folds go
        // through rm_fold4 (3 instructions per accumulator instead of the compiler's 8)
        for (; o + 4 <= g.osz; o += 4)
        {
            const u64 *row = g.matrix + (u64)o * ms;
            Acc acc[W][4];
#pragma unroll
            for (int w = 0; w < W; ++w)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    acc[w][k].lo = acc[w][k].hi = 0, acc[w][k].cc = 0;
            if (EP != 0)
            {
                u64 ex[4][W], eb[4][W];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    extra(o + k, g.mods[g.omap.id[o + k]].q, row + k * ms, ex[k], eb[k]);
#pragma unroll
                for (int w = 0; w < W; ++w)
                {
                    u64 mid[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                    {
                        mid[k] = 0;
                        rm_mac(acc[w][k], mid[k], ex[k][w], eb[k][w]);
                    }
                    rm_fold4(acc[w][0], acc[w][1], acc[w][2], acc[w][3], mid[0], mid[1], mid[2], mid[3]);
                }
            }
            u32 j = 0;
            for (; j + 2 <= g.isz; j += 2)
            {
                u64 x0[W], x1[W];
                ldw<W>(x0, mine + j * 256 * W);
                ldw<W>(x1, mine + (j + 1) * 256 * W);
                u64 m0[4], m1[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    m0[k] = row[k * ms + j], m1[k] = row[k * ms + j + 1];
#pragma unroll
                for (int w = 0; w < W; ++w)
                {
                    u64 mid[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        rm_mac2(acc[w][k], mid[k], x0[w], m0[k], x1[w], m1[k]);
                    rm_fold4(acc[w][0], acc[w][1], acc[w][2], acc[w][3], mid[0], mid[1], mid[2], mid[3]);
                }
            }
            if (j < g.isz)
            {
                u64 x0[W];
                ldw<W>(x0, mine + j * 256 * W);
#pragma unroll
                for (int w = 0; w < W; ++w)
                {
                    u64 mid[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                    {
                        mid[k] = 0;
                        rm_mac(acc[w][k], mid[k], x0[w], row[k * ms + j]);
                    }
                    rm_fold4(acc[w][0], acc[w][1], acc[w][2], acc[w][3], mid[0], mid[1], mid[2], mid[3]);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
                const DevMod m = g.mods[g.omap.id[o + k]];
                u64 v[W];
#pragma unroll
                for (int w = 0; w < W; ++w)
                    v[w] = reduce128(acc[w][k].lo, acc[w][k].hi + acc[w][k].cc, m.q, m.ratio1, m.r64_quo);
                finish_store(o + k, v);
            }
        }
        for (; o < g.osz; ++o)
        {
            const DevMod m = g.mods[g.omap.id[o]];
            u64 v[W];
            dot(g.matrix + (u64)o * ms, m, v, (int)o);
            finish_store(o, v);
        }
        if (EP == 0 && g.mt_matrix)
        {
            u64 acc[W];
#pragma unroll
            for (int w = 0; w < W; ++w)
                acc[w] = 0;
            for (u32 j = 0; j < g.isz; ++j)
            {
                u64 x[W];
                ldw<W>(x, mine + j * 256 * W);
#pragma unroll
                for (int w = 0; w < W; ++w)
                    acc[w] += x[w] * g.mt_matrix[j];
            }
#pragma unroll
            for (int w = 0; w < W; ++w)
                acc[w] &= MT - 1;
            stw<W>(dst + ((u64)g.mt_limb << g.log_n), acc);
        }
    }

    // The same conversion for at most IT input limbs (IT = 4, 8, 12, 16), two coefficients per thread, with the scaled residues
    // in REGISTERS and every loop over the input limbs unrolled: no LDS (the staged residues were private to their thread
    // anyway, LDS only gave them a run-time index, and 32 KB per workgroup capped the CU at five workgroups), the matrix
    // words of a block of four outputs are requested together instead of pair by pair behind a wait each.  Limbs beyond
    // isz contribute a zero residue (their matrix index is clamped, the word read does not matter).
    template <int EP, int IT>
    __global__ __launch_bounds__(256) void bconv_reg_kernel(BconvArgs g)
    {
        constexpr int W = 2;
        const u64 i = ((u64)blockIdx.x * 256 + threadIdx.x) * W;
        if (i >= g.total)
            return;
        const u32 c = (u32)(i & ((1u << g.log_n) - 1));
        const u64 p = i >> g.log_n;
        const u64 *src = g.in + p * g.in_stride + c;
        const u64 obj = p / g.grp, sub = p - obj * g.grp;
        u64 *cpy = g.copy_out ? g.copy_out + obj * g.copy_obj_stride + sub * g.copy_stride + c : nullptr;
        u64 *dst = g.out + obj * g.out_obj_stride + sub * g.out_stride + c;
        const u32 last = g.isz - 1;
        u64 sx[IT][W];
#pragma unroll
        for (int j = 0; j < IT; ++j)
        {
            sx[j][0] = sx[j][1] = 0;
            if ((u32)j < g.isz)
                ldw<W>(sx[j], src + ((u64)j << g.log_n));
        }
#pragma unroll
        for (int j = 0; j < IT; ++j)
        {
            if ((u32)j < g.isz)
            {
                const u64 q = g.mods[g.imap.id[j]].q;
                if (cpy)
                    stw<W>(cpy + ((u64)j << g.log_n), sx[j]);
#pragma unroll
                for (int w = 0; w < W; ++w)
                    sx[j][w] = mul_shoup(sx[j][w], g.inv_punc[j].op, g.inv_punc[j].quo, q);
            }
        }
        constexpr bool XCOL = EP == 1 || EP == 2;
        const u32 ms = g.isz + (XCOL ? 1 : 0);
        u64 mt_r[W], alpha[W];
        bool alpha_neg[W];
#pragma unroll
        for (int w = 0; w < W; ++w)
            mt_r[w] = alpha[w] = 0, alpha_neg[w] = false;
        auto clamp = [&](int j) { return (u32)j < g.isz ? (u32)j : last; };
        auto extra = [&](u32 o, u64 q, const u64 *row, u64 (&ex)[W], u64 (&eb)[W]) {
            if (EP == 1)
            {
#pragma unroll
                for (int w = 0; w < W; ++w)
                {
                    u64 t = mt_r[w];
                    if (t >= (MT >> 1))
                        t += q - MT;
                    ex[w] = t;
                    eb[w] = row[g.isz];
                }
            }
            if (EP == 2)
            {
                ldw<W>(ex, g.ep_in + p * g.ep_in_stride + ((u64)o << g.log_n) + c);
#pragma unroll
                for (int w = 0; w < W; ++w)
                    eb[w] = row[g.isz];
            }
            if (EP == 3)
            {
#pragma unroll
                for (int w = 0; w < W; ++w)
                {
                    ex[w] = alpha[w];
                    eb[w] = alpha_neg[w] ? g.ep_c0[o].op : g.ep_c1[o].op;
                }
            }
        };
        // K outputs o .. o + K - 1 (K = 4 or 1); with_tail: the tail's term goes in as well
        auto block = [&](u32 o, auto kc, bool with_tail, const u64 *row, u64 (&res)[decltype(kc)::value][W]) {
            constexpr int K = decltype(kc)::value;
            Acc acc[W][K];
#pragma unroll
            for (int w = 0; w < W; ++w)
#pragma unroll
                for (int k = 0; k < K; ++k)
                    acc[w][k].lo = acc[w][k].hi = 0, acc[w][k].cc = 0;
            if (EP != 0 && with_tail)
            {
                u64 ex[K][W], eb[K][W];
#pragma unroll
                for (int k = 0; k < K; ++k)
                    extra(o + k, g.mods[g.omap.id[o + k]].q, row + k * ms, ex[k], eb[k]);
#pragma unroll
                for (int w = 0; w < W; ++w)
#pragma unroll
                    for (int k = 0; k < K; ++k)
                    {
                        u64 mid = 0;
                        rm_mac(acc[w][k], mid, ex[k][w], eb[k][w]);
                        rm_fold(acc[w][k], mid);
                    }
            }
#pragma unroll
            for (int j = 0; j < IT; j += 2)
            {
                if ((u32)j >= g.isz)
                    break;
                u64 m0[K], m1[K];
#pragma unroll
                for (int k = 0; k < K; ++k)
                    m0[k] = row[k * ms + j], m1[k] = row[k * ms + clamp(j + 1)];
#pragma unroll
                for (int w = 0; w < W; ++w)
                {
                    u64 mid[K];
#pragma unroll
                    for (int k = 0; k < K; ++k)
                        rm_mac2(acc[w][k], mid[k], sx[j][w], m0[k], sx[j + 1][w], m1[k]);
                    if constexpr (K == 4)
                        rm_fold4(acc[w][0], acc[w][1], acc[w][2], acc[w][3], mid[0], mid[1], mid[2], mid[3]);
                    else
                        rm_fold(acc[w][0], mid[0]);
                }
            }
#pragma unroll
            for (int k = 0; k < K; ++k)
            {
                const DevMod m = with_tail || EP != 3 ? g.mods[g.omap.id[o + k]] : g.mods[g.ep_msk_id];
#pragma unroll
                for (int w = 0; w < W; ++w)
                    res[k][w] = reduce128(acc[w][k].lo, acc[w][k].hi + acc[w][k].cc, m.q, m.ratio1, m.r64_quo);
            }
        };
        if (EP == 1)
        {
            u64 acc[W] = { 0, 0 };
#pragma unroll
            for (int j = 0; j < IT; ++j)
#pragma unroll
                for (int w = 0; w < W; ++w)
                    acc[w] += sx[j][w] * g.mt_matrix[clamp(j)];
#pragma unroll
            for (int w = 0; w < W; ++w)
                mt_r[w] = ((acc[w] & (MT - 1)) * g.ep_k0) & (MT - 1);
        }
        if (EP == 3)
        {
            const DevMod m = g.mods[g.ep_msk_id];
            u64 conv_sk[1][W], in_sk[W];
            block(0, std::integral_constant<int, 1>{}, false, g.matrix + (u64)g.osz * ms, conv_sk);
            ldw<W>(in_sk, src + ((u64)g.isz << g.log_n));
#pragma unroll
            for (int w = 0; w < W; ++w)
            {
                alpha[w] = mul_shoup(conv_sk[0][w] + (m.q - in_sk[w]), g.ep_k1.op, g.ep_k1.quo, m.q);
                alpha_neg[w] = alpha[w] > (m.q >> 1);
                if (alpha_neg[w])
                    alpha[w] = m.q - alpha[w];
            }
        }
        u32 o = 0;
        for (; o + 4 <= g.osz; o += 4)
        {
            u64 res[4][W];
            block(o, std::integral_constant<int, 4>{}, true, g.matrix + (u64)o * ms, res);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                stw<W>(dst + ((u64)(g.out_off + o + k) << g.log_n), res[k]);
        }
        for (; o < g.osz; ++o)
        {
            u64 res[1][W];
            block(o, std::integral_constant<int, 1>{}, true, g.matrix + (u64)o * ms, res);
            stw<W>(dst + ((u64)(g.out_off + o) << g.log_n), res[0]);
        }
        if (EP == 0 && g.mt_matrix)
        {
            u64 acc[W] = { 0, 0 };
#pragma unroll
            for (int j = 0; j < IT; ++j)
#pragma unroll
                for (int w = 0; w < W; ++w)
                    acc[w] += sx[j][w] * g.mt_matrix[clamp(j)];
#pragma unroll
            for (int w = 0; w < W; ++w)
                acc[w] &= MT - 1;
            stw<W>(dst + ((u64)g.mt_limb << g.log_n), acc);
        }
    }

    struct PieceArgs
    {
        const u64 *in;
        u64 *out;
        const DevMod *mods;
        const MulOp *c0; // per-limb constants (meaning depends on the kernel)
        const MulOp *c1;
        const MulOp *c2;
        LimbMap map;
        u32 n_limbs, log_n, aux_limb;
        u64 in_stride, out_stride;
        u64 k0; // scalar constant
        MulOp k1;
        u64 total; // polys * n_limbs * N
    };

    // sm_mrq (rns.cpp:981-1041): in [polys][Bsk+1][N] (last limb modulo m_tilde) -> out [polys][Bsk][N]
    //   r = in_mt * (-q^-1 mod m_tilde) mod m_tilde; r >= m_tilde/2 -> r += m_i - m_tilde;
    //   out_i = (r * (q mod m_i) + in_i) * m_tilde^-1 mod m_i         c0 = q mod m_i, c1 = m_tilde^-1 mod m_i
    __global__ __launch_bounds__(256) void sm_mrq_kernel(PieceArgs g)
    {
        const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
        if (i >= g.total)
            return;
        const u32 c = (u32)(i & ((1u << g.log_n) - 1));
        const u64 r = i >> g.log_n;
        const u32 j = (u32)(r % g.n_limbs);
        const u64 p = r / g.n_limbs;
        const DevMod m = g.mods[g.map.id[j]];
        const u64 *src = g.in + p * g.in_stride + c;
        u64 t = (src[(u64)g.n_limbs << g.log_n] * g.k0) & (MT - 1);
        if (t >= (MT >> 1))
            t += m.q - MT;
        const u64 a = mul_shoup(t, g.c0[j].op, g.c0[j].quo, m.q);
        const u64 b = barrett64(src[(u64)j << g.log_n], m.q, m.ratio1);
        g.out[p * g.out_stride + ((u64)j << g.log_n) + c] = mul_shoup(add_mod(a, b, m.q), g.c1[j].op, g.c1[j].quo, m.q);
    }

    // (x0,x1) (x) (y0,y1) over an arbitrary limb list; inputs may be lazy (< 4q): evaluator.cpp:497-541
    struct TensorMapArgs
    {
        const u64 *x; // [batch][..] poly 0, 1 at x, x + poly
        const u64 *y;
        u64 *out; // [batch][3][nl][N]
        const DevMod *mods;
        LimbMap map;
        u32 n_limbs, log_n;
        u64 in_stride; // words between consecutive batch elements of x and of y
        u64 total;     // batch * nl * N
    };

    __global__ __launch_bounds__(256) void tensor_map_kernel(TensorMapArgs g)
    {
        const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
        if (i >= g.total)
            return;
        const u64 poly = (u64)g.n_limbs << g.log_n;
        const u64 b = i / poly, off = i % poly;
        const DevMod m = g.mods[g.map.id[off >> g.log_n]];
        const u64 *x = g.x + b * g.in_stride + off, *y = g.y + b * g.in_stride + off;
        const u64 x0 = x[0], x1 = x[poly], y0 = y[0], y1 = y[poly];
        u64 *o = g.out + b * 3 * poly + off;
        o[0] = reduce128(x0 * y0, mulhi64(x0, y0), m.q, m.ratio1, m.r64_quo);
        // the middle term as one 128-bit sum and ONE reduction (operands below 2^63: the sum is below 2^127, and
        // Barrett-128 takes any 128-bit value); the canonical result is that of the reference's two products and add
        u64 lo = 0, hi = 0;
        mac128(lo, hi, x1, y0);
        mac128(lo, hi, x0, y1);
        o[poly] = reduce128(lo, hi, m.q, m.ratio1, m.r64_quo);
        o[2 * poly] = reduce128(x1 * y1, mulhi64(x1, y1), m.q, m.ratio1, m.r64_quo);
    }

    u32 blocks_for(u64 items)
    {
        return (u32)((items + 255) / 256);
    }

    MulOp mk(const cat::HostMod &m, u64 v)
    {
        v %= m.q;
        return MulOp{ v, m.shoup(v) };
    }
} // namespace

namespace cat
{
    // RNSTool::initialize (util/rns.cpp:578-787) for the level with L limbs; device tables in one blob.
    static cat_status build_level(cat_ctx *ctx, uint32_t L)
    {
        BfvLevel &lv = ctx->bfv[L];
        const uint32_t k = ctx->k;
        std::vector<u64> q(L);
        for (uint32_t i = 0; i < L; ++i)
            q[i] = ctx->hmods[i].q;
        const std::vector<u64> prod_q = mp_product(q, SIZE_MAX);
        // rns.cpp:607-615: |B| = L, one more when 32 + bits(t) + bits(q) >= 61 L + 61
        int t_bits = 64 - __builtin_clzll(ctx->plain_modulus);
        uint32_t B = L;
        if (32 + t_bits + mp_bits(prod_q) >= 61 * (int)L + 61)
            B++;
        if (k + 2 + B > ctx->n_mod)
            return fail(CAT_ERR_LOGIC, "invalid rns bases");
        lv.L = L;
        lv.B = B;
        lv.Bsk = B + 1;
        lv.idx_msk = k;
        std::memset(&lv.map_q, 0, sizeof(LimbMap));
        std::memset(&lv.map_Bsk, 0, sizeof(LimbMap));
        for (uint32_t i = 0; i < L; ++i)
            lv.map_q.id[i] = (unsigned char)i;
        for (uint32_t i = 0; i < B; ++i)
            lv.map_Bsk.id[i] = (unsigned char)(k + 2 + i);
        lv.map_Bsk.id[B] = (unsigned char)k;
        auto hm = [&](uint32_t id) -> const HostMod & { return ctx->hmods[id]; };
        std::vector<u64> bvals(B);
        for (uint32_t i = 0; i < B; ++i)
            bvals[i] = hm(k + 2 + i).q;
        const std::vector<u64> prod_B = mp_product(bvals, SIZE_MAX);

        std::vector<u64> blob;
        auto put_ops = [&](const std::vector<MulOp> &v) {
            size_t off = blob.size();
            for (const MulOp &o : v)
            {
                blob.push_back(o.op);
                blob.push_back(o.quo);
            }
            return off;
        };
        auto put_words = [&](const std::vector<u64> &v) {
            size_t off = blob.size();
            blob.insert(blob.end(), v.begin(), v.end());
            if (blob.size() & 1)
                blob.push_back(0); // keep MulOp tables 16-byte aligned
            return off;
        };
        std::vector<MulOp> ops;
        std::vector<u64> words;
        u64 inv = 0;

        // base q: (q/q_i)^-1 mod q_i, m_tilde mod q_i, t mod q_i, B mod q_i, -B mod q_i
        std::vector<std::vector<u64>> punc_q(L), punc_B(B);
        for (uint32_t i = 0; i < L; ++i)
            punc_q[i] = mp_product(q, i);
        for (uint32_t i = 0; i < B; ++i)
            punc_B[i] = mp_product(bvals, i);
        ops.clear();
        for (uint32_t i = 0; i < L; ++i)
        {
            if (!invert_mod(mp_mod(punc_q[i], q[i]), q[i], inv))
                return fail(CAT_ERR_LOGIC, "invalid rns bases");
            ops.push_back(mk(hm(i), inv));
        }
        lv.o_inv_punc_q = put_ops(ops);
        // fastbconv_m_tilde multiplies by m_tilde and fast_floor's caller by t before the conversion scales by
        // (q/q_i)^-1 (rns.cpp:1104-1113, evaluator.cpp:549-556): x a mod q then b mod q is x (a b mod q) mod q, one multiply
        const std::vector<MulOp> inv_punc = ops;
        auto merged = [&](u64 c) {
            std::vector<MulOp> v;
            for (uint32_t i = 0; i < L; ++i)
                v.push_back(mk(hm(i), (u64)((unsigned __int128)(c % q[i]) * inv_punc[i].op % q[i])));
            return v;
        };
        lv.o_mtilde_q = put_ops(merged(MT));
        lv.o_t_q = put_ops(merged(ctx->plain_modulus));
        ops.clear();
        for (uint32_t i = 0; i < L; ++i)
            ops.push_back(mk(hm(i), mp_mod(prod_B, q[i])));
        lv.o_prodB_q = put_ops(ops);
        ops.clear();
        for (uint32_t i = 0; i < L; ++i)
            ops.push_back(mk(hm(i), q[i] - mp_mod(prod_B, q[i])));
        lv.o_negprodB_q = put_ops(ops);
        // q -> Bsk matrix [Bsk][L], q -> m_tilde [L]
        words.clear();
        for (uint32_t o = 0; o < lv.Bsk; ++o)
            for (uint32_t i = 0; i < L; ++i)
                words.push_back(mp_mod(punc_q[i], hm(lv.map_Bsk.id[o]).q));
        lv.o_q_to_Bsk = put_words(words);
        words.clear();
        for (uint32_t i = 0; i < L; ++i)
            words.push_back(mp_mod(punc_q[i], MT));
        lv.o_q_to_mt = put_words(words);
        // base B: (B/b_i)^-1 mod b_i; B -> q U {m_sk} matrix [L + 1][B]
        ops.clear();
        for (uint32_t i = 0; i < B; ++i)
        {
            if (!invert_mod(mp_mod(punc_B[i], bvals[i]), bvals[i], inv))
                return fail(CAT_ERR_LOGIC, "invalid rns bases");
            ops.push_back(mk(hm(k + 2 + i), inv));
        }
        lv.o_inv_punc_B = put_ops(ops);
        words.clear();
        for (uint32_t o = 0; o < L; ++o)
            for (uint32_t i = 0; i < B; ++i)
                words.push_back(mp_mod(punc_B[i], q[o]));
        for (uint32_t i = 0; i < B; ++i)
            words.push_back(mp_mod(punc_B[i], hm(k).q));
        lv.o_B_to_q = put_words(words);
        lv.o_B_to_msk = lv.o_B_to_q + (size_t)L * B;
        // Bsk: q mod m_i, m_tilde^-1 mod m_i, q^-1 mod m_i, t mod m_i
        std::vector<MulOp> pq, imt, iq, tb;
        for (uint32_t i = 0; i < lv.Bsk; ++i)
        {
            const HostMod &m = hm(lv.map_Bsk.id[i]);
            const u64 qm = mp_mod(prod_q, m.q);
            pq.push_back(mk(m, qm));
            if (!invert_mod(MT % m.q, m.q, inv))
                return fail(CAT_ERR_LOGIC, "invalid rns bases");
            imt.push_back(mk(m, inv));
            if (!invert_mod(qm, m.q, inv))
                return fail(CAT_ERR_LOGIC, "invalid rns bases");
            iq.push_back(mk(m, inv));
            tb.push_back(mk(m, ctx->plain_modulus));
        }
        lv.o_prod_q_Bsk = put_ops(pq);
        lv.o_inv_mt_Bsk = put_ops(imt);
        lv.o_inv_q_Bsk = put_ops(iq);
        lv.o_t_Bsk = put_ops(tb);
        // sm_mrq and the fast_floor tail are affine in the converted residue v_o = sum_i s_i (q/q_i) mod m_o:
        //   ((r (q mod m_o) + v_o) m_tilde^-1,   (x [t] + (m_o - v_o)) q^-1
        // so each is ONE dot product over L + 1 terms (the tail's own operand r or x against one more column) and one
        // reduction -- the canonical residue of the same number the reference reaches in three reductions (bconv_kernel)
        auto mulm = [](u64 a, u64 b, u64 m) { return (u64)((unsigned __int128)a * b % m); };
        std::vector<u64> m_mrq, m_floor, m_floor_t;
        for (uint32_t o = 0; o < lv.Bsk; ++o)
        {
            const u64 m = hm(lv.map_Bsk.id[o]).q;
            for (uint32_t i = 0; i < L; ++i)
            {
                const u64 e = mp_mod(punc_q[i], m);
                m_mrq.push_back(mulm(e, imt[o].op, m));
                const u64 f = mulm(e, iq[o].op, m);
                m_floor.push_back(f ? m - f : 0);
                m_floor_t.push_back(f ? m - f : 0);
            }
            m_mrq.push_back(mulm(pq[o].op, imt[o].op, m));
            m_floor.push_back(iq[o].op);
            m_floor_t.push_back(mulm(tb[o].op, iq[o].op, m));
        }
        lv.o_q_to_Bsk_mrq = put_words(m_mrq);
        lv.o_q_to_Bsk_floor = put_words(m_floor);
        lv.o_q_to_Bsk_floor_t = put_words(m_floor_t);
        // scalars
        if (!invert_mod(mp_mod(prod_q, MT), MT, inv))
            return fail(CAT_ERR_LOGIC, "invalid rns bases");
        lv.neg_inv_q_mt = (MT - inv) & (MT - 1);
        if (!invert_mod(mp_mod(prod_B, hm(k).q), hm(k).q, inv))
            return fail(CAT_ERR_LOGIC, "invalid rns bases");
        lv.inv_B_msk = mk(hm(k), inv);

        CAT_HIP(hipMalloc((void **)&lv.d_blob, blob.size() * sizeof(u64)));
        CAT_HIP(hipMemcpy(lv.d_blob, blob.data(), blob.size() * sizeof(u64), hipMemcpyHostToDevice));
        lv.ready = true;
        return CAT_OK;
    }

    cat_status build_bfv_levels(cat_ctx *ctx)
    {
        ctx->bfv.assign(ctx->k + 1, BfvLevel{});
        // data levels have 1 .. max(1, k-1) limbs (context.cu:607-637); L = k exists only when k == 1
        const uint32_t top = ctx->k > 1 ? ctx->k - 1 : 1;
        for (uint32_t L = 1; L <= top; ++L)
        {
            cat_status st = build_level(ctx, L);
            if (st != CAT_OK)
                return st;
        }
        return CAT_OK;
    }

    void destroy_bfv_levels(cat_ctx *ctx)
    {
        for (auto &l : ctx->bfv)
            if (l.d_blob)
                (void)hipFree(l.d_blob);
        ctx->bfv.clear();
    }
} // namespace cat

namespace
{
    using namespace cat;

    const MulOp *ops_at(const BfvLevel &lv, size_t off)
    {
        return reinterpret_cast<const MulOp *>(lv.d_blob + off);
    }

    // dev: the caller's DeviceScope (the context's device stays current until the entry point returns)
    cat_status level_of(cat_ctx *ctx, uint32_t L, const BfvLevel **out, DeviceScope &dev)
    {
        const cat_status ds = dev.enter(ctx);
        if (ds != CAT_OK)
            return ds;
        if (ctx->scheme != CAT_SCHEME_BFV)
            return fail(CAT_ERR_LOGIC, "unsupported scheme");
        if (L == 0 || L >= ctx->bfv.size() || !ctx->bfv[L].ready)
            return fail(CAT_ERR_INVALID_ARG, "encrypted is not valid for encryption parameters");
        *out = &ctx->bfv[L];
        return CAT_OK;
    }

    cat_status launch_bconv(const BconvArgs &g, int ep, hipStream_t s)
    {
        if (g.total == 0)
            return CAT_OK;
        // up to 16 input limbs (and an even number of coefficients): residues in registers, two coefficients per thread
        if (g.isz <= 16 && (g.total & 1) == 0)
        {
            const dim3 grid(blocks_for(g.total / 2)), block(256);
            const int it = (int)((g.isz + 3) / 4);
#define CAT_BCONV_REG(EP)                                                                                              \
    switch (it)                                                                                                        \
    {                                                                                                                  \
    case 1: hipLaunchKernelGGL((bconv_reg_kernel<EP, 4>), grid, block, 0, s, g); break;                                \
    case 2: hipLaunchKernelGGL((bconv_reg_kernel<EP, 8>), grid, block, 0, s, g); break;                                \
    case 3: hipLaunchKernelGGL((bconv_reg_kernel<EP, 12>), grid, block, 0, s, g); break;                               \
    default: hipLaunchKernelGGL((bconv_reg_kernel<EP, 16>), grid, block, 0, s, g); break;                              \
    }
            switch (ep)
            {
            case 0: CAT_BCONV_REG(0) break;
            case 1: CAT_BCONV_REG(1) break;
            case 2: CAT_BCONV_REG(2) break;
            default: CAT_BCONV_REG(3) break;
            }
#undef CAT_BCONV_REG
            CAT_HIP(hipGetLastError());
            return CAT_OK;
        }
        // more input limbs: residues staged in LDS under a run-time index, one coefficient per thread
        const dim3 grid(blocks_for(g.total)), block(256);
        const size_t lds = g.isz * 256 * sizeof(u64);
        switch (ep)
        {
        case 0: hipLaunchKernelGGL((bconv_kernel<0, 1>), grid, block, lds, s, g); break;
        case 1: hipLaunchKernelGGL((bconv_kernel<1, 1>), grid, block, lds, s, g); break;
        case 2: hipLaunchKernelGGL((bconv_kernel<2, 1>), grid, block, lds, s, g); break;
        default: hipLaunchKernelGGL((bconv_kernel<3, 1>), grid, block, lds, s, g); break;
        }
        CAT_HIP(hipGetLastError());
        return CAT_OK;
    }

    // FastBConv q -> Bsk: in [polys][in_limbs][N] -> out limbs 0..Bsk-1
    //   ep 0: plain (+ the residue modulo m_tilde in limb Bsk when with_mt: fastbconv_m_tilde)
    //   ep 1: fastbconv_m_tilde followed by sm_mrq, out [polys][Bsk][N] (evaluator.cpp:466-474 as one pass)
    //   ep 2: fast_floor (rns.cpp:1043-1090): in_B = the Bsk part of the input; times_t multiplies both parts by t first
    // scale: the input is multiplied by 1 (PRE_NONE), m_tilde (PRE_MT) or t (PRE_T) first
    enum { PRE_NONE, PRE_MT, PRE_T };
    cat_status conv_q_to_Bsk(cat_ctx *ctx, const BfvLevel &lv, const u64 *in, u64 in_stride, u64 *out, u64 out_stride,
                             int scale, bool with_mt, u64 polys, hipStream_t s, int ep = 0, const u64 *in_B = nullptr,
                             u64 in_B_stride = 0, const BconvArgs *place = nullptr)
    {
        BconvArgs g{};
        g.in = in;
        g.out = out;
        g.mods = ctx->d_mods;
        g.inv_punc = ops_at(lv, scale == PRE_MT ? lv.o_mtilde_q : scale == PRE_T ? lv.o_t_q : lv.o_inv_punc_q);
        g.grp = 1;
        g.out_obj_stride = out_stride;
        if (place) // output grouping and the copy of the input (BFV multiply)
        {
            g.grp = place->grp;
            g.out_obj_stride = place->out_obj_stride;
            g.copy_out = place->copy_out;
            g.copy_stride = place->copy_stride;
            g.copy_obj_stride = place->copy_obj_stride;
        }
        g.matrix = lv.d_blob + (ep == 1 ? lv.o_q_to_Bsk_mrq : ep == 2 ? (scale == PRE_T ? lv.o_q_to_Bsk_floor_t : lv.o_q_to_Bsk_floor)
                                                        : lv.o_q_to_Bsk);
        g.mt_matrix = with_mt || ep == 1 ? lv.d_blob + lv.o_q_to_mt : nullptr;
        g.imap = lv.map_q;
        g.omap = lv.map_Bsk;
        g.isz = lv.L;
        g.osz = lv.Bsk;
        g.log_n = ctx->log_n;
        g.in_stride = in_stride;
        g.out_stride = out_stride;
        g.out_off = 0;
        g.mt_limb = lv.Bsk;
        g.total = polys << ctx->log_n;
        if (ep == 1)
            g.ep_k0 = lv.neg_inv_q_mt;
        else if (ep == 2)
        {
            g.ep_in = in_B;
            g.ep_in_stride = in_B_stride;
        }
        return launch_bconv(g, ep, s);
    }

    cat_status run_sm_mrq(cat_ctx *ctx, const BfvLevel &lv, const u64 *in, u64 in_stride, u64 *out, u64 out_stride,
                          u64 polys, hipStream_t s)
    {
        PieceArgs g{};
        g.in = in;
        g.out = out;
        g.mods = ctx->d_mods;
        g.c0 = ops_at(lv, lv.o_prod_q_Bsk);
        g.c1 = ops_at(lv, lv.o_inv_mt_Bsk);
        g.map = lv.map_Bsk;
        g.n_limbs = lv.Bsk;
        g.log_n = ctx->log_n;
        g.in_stride = in_stride;
        g.out_stride = out_stride;
        g.k0 = lv.neg_inv_q_mt;
        g.total = (polys * lv.Bsk) << ctx->log_n;
        if (g.total == 0)
            return CAT_OK;
        hipLaunchKernelGGL(sm_mrq_kernel, dim3(blocks_for(g.total)), dim3(256), 0, s, g);
        CAT_HIP(hipGetLastError());
        return CAT_OK;
    }

    // in_q / in_B: the q part and the Bsk part of the input (each with its own poly stride); `out` must not overlap in_B
    cat_status run_fast_floor(cat_ctx *ctx, const BfvLevel &lv, const u64 *in_q, u64 in_q_stride, const u64 *in_B,
                              u64 in_B_stride, u64 *out, u64 out_stride, bool times_t, u64 polys, hipStream_t s)
    {
        return conv_q_to_Bsk(ctx, lv, in_q, in_q_stride, out, out_stride, times_t ? PRE_T : PRE_NONE, false, polys, s, 2, in_B,
                             in_B_stride);
    }

    // in [polys][Bsk][N] -> out [polys][L][N] (out_stride words apart): FastBConv_{B -> q} and FastBConv_{B -> m_sk} of
    // the same staged residues, alpha, and the correction, per coefficient in one thread
    cat_status run_fastbconv_sk(cat_ctx *ctx, const BfvLevel &lv, const u64 *in, u64 in_stride, u64 *out, u64 out_stride,
                                u64 polys, hipStream_t s)
    {
        BconvArgs b{};
        b.in = in;
        b.out = out;
        b.mods = ctx->d_mods;
        b.inv_punc = ops_at(lv, lv.o_inv_punc_B);
        b.matrix = lv.d_blob + lv.o_B_to_q; // [L + 1][|B|]: the rows of q, then the row of m_sk
        b.imap = lv.map_Bsk; // first |B| entries are base B; entry |B| is m_sk (the limb alpha needs)
        b.omap = lv.map_q;
        b.isz = lv.B;
        b.osz = lv.L;
        b.log_n = ctx->log_n;
        b.in_stride = in_stride;
        b.out_stride = out_stride;
        b.grp = 1;
        b.out_obj_stride = out_stride;
        b.total = polys << ctx->log_n;
        b.ep_c0 = ops_at(lv, lv.o_prodB_q);
        b.ep_c1 = ops_at(lv, lv.o_negprodB_q);
        b.ep_k1 = lv.inv_B_msk;
        b.ep_msk_id = lv.idx_msk;
        return launch_bconv(b, 3, s);
    }
} // namespace

extern "C" {

uint32_t cat_bfv_base_Bsk_size(const cat_ctx *ctx, uint32_t L)
{
    if (!ctx || L >= ctx->bfv.size() || !ctx->bfv[L].ready)
        return 0;
    return ctx->bfv[L].Bsk;
}

cat_status cat_bfv_base_Bsk(const cat_ctx *ctx, uint32_t L, uint64_t *out_moduli)
{
    if (!ctx || !out_moduli || L >= ctx->bfv.size() || !ctx->bfv[L].ready)
        return fail(CAT_ERR_INVALID_ARG, "no BEHZ bases for this level");
    const BfvLevel &lv = ctx->bfv[L];
    for (uint32_t i = 0; i < lv.Bsk; ++i)
        out_moduli[i] = ctx->hmods[lv.map_Bsk.id[i]].q;
    return CAT_OK;
}

cat_status cat_fastbconv_m_tilde(cat_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, uint32_t L, uint32_t n_polys,
                                 cat_stream s)
{
    const BfvLevel *lv;
    DeviceScope dev;
    cat_status st = level_of(ctx, L, &lv, dev);
    if (st != CAT_OK)
        return st;
    if (n_polys && (!d_in || !d_out))
        return fail(CAT_ERR_INVALID_ARG, "input");
    if (((uintptr_t)d_in | (uintptr_t)d_out) & 15)
        return fail(CAT_ERR_INVALID_ARG, "device pointers must be 16-byte aligned");
    return conv_q_to_Bsk(ctx, *lv, (const u64 *)d_in, (u64)L * ctx->n, (u64 *)d_out, (u64)(lv->Bsk + 1) * ctx->n,
                         PRE_MT, true, n_polys, (hipStream_t)s);
}

cat_status cat_sm_mrq(cat_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, uint32_t L, uint32_t n_polys, cat_stream s)
{
    const BfvLevel *lv;
    DeviceScope dev;
    cat_status st = level_of(ctx, L, &lv, dev);
    if (st != CAT_OK)
        return st;
    if (n_polys && (!d_in || !d_out))
        return fail(CAT_ERR_INVALID_ARG, "input");
    if (((uintptr_t)d_in | (uintptr_t)d_out) & 15)
        return fail(CAT_ERR_INVALID_ARG, "device pointers must be 16-byte aligned");
    return run_sm_mrq(ctx, *lv, (const u64 *)d_in, (u64)(lv->Bsk + 1) * ctx->n, (u64 *)d_out, (u64)lv->Bsk * ctx->n,
                      n_polys, (hipStream_t)s);
}

cat_status cat_fast_floor(cat_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, uint32_t L, uint32_t n_polys,
                          cat_stream s)
{
    const BfvLevel *lv;
    DeviceScope dev;
    cat_status st = level_of(ctx, L, &lv, dev);
    if (st != CAT_OK)
        return st;
    if (n_polys && (!d_in || !d_out))
        return fail(CAT_ERR_INVALID_ARG, "input");
    if (((uintptr_t)d_in | (uintptr_t)d_out) & 15)
        return fail(CAT_ERR_INVALID_ARG, "device pointers must be 16-byte aligned");
    const u64 stride = (u64)(L + lv->Bsk) * ctx->n;
    return run_fast_floor(ctx, *lv, (const u64 *)d_in, stride, (const u64 *)d_in + (u64)L * ctx->n, stride,
                          (u64 *)d_out, (u64)lv->Bsk * ctx->n, false, n_polys, (hipStream_t)s);
}

cat_status cat_fastbconv_sk(cat_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, uint32_t L, uint32_t n_polys,
                            cat_stream s)
{
    const BfvLevel *lv;
    DeviceScope dev;
    cat_status st = level_of(ctx, L, &lv, dev);
    if (st != CAT_OK)
        return st;
    if (n_polys && (!d_in || !d_out))
        return fail(CAT_ERR_INVALID_ARG, "input");
    if (((uintptr_t)d_in | (uintptr_t)d_out) & 15)
        return fail(CAT_ERR_INVALID_ARG, "device pointers must be 16-byte aligned");
    return run_fastbconv_sk(ctx, *lv, (const u64 *)d_in, (u64)lv->Bsk * ctx->n, (u64 *)d_out, (u64)L * ctx->n, n_polys,
                            (hipStream_t)s);
}

cat_status cat_bfv_multiply(cat_ctx *ctx, const uint64_t *d_x, const uint64_t *d_y, uint64_t *d_out, uint32_t L,
                            uint32_t batch, cat_stream stream)
{
    const BfvLevel *lvp;
    DeviceScope dev;
    cat_status st = level_of(ctx, L, &lvp, dev);
    if (st != CAT_OK)
        return st;
    if (batch == 0)
        return CAT_OK;
    if (!d_x || !d_y || !d_out)
        return fail(CAT_ERR_INVALID_ARG, "encrypted1 or encrypted2 is not valid");
    if (((uintptr_t)d_x | (uintptr_t)d_y | (uintptr_t)d_out) & 15)
        return fail(CAT_ERR_INVALID_ARG, "device pointers must be 16-byte aligned");
    if (d_out == d_x || d_out == d_y)
        return fail(CAT_ERR_INVALID_ARG, "destination must not alias the inputs");
    const BfvLevel &lv = *lvp;
    hipStream_t s = (hipStream_t)stream;
    const u64 N = ctx->n, Bsk = lv.Bsk, b = batch;
    const u64 pq = (u64)L * N, pB = Bsk * N;
    // workspace (words): inq [b][4][L][N] | inB [b][4][Bsk][N] | tq [b][3][L][N] | tB [b][3][Bsk][N] | fl [b][3][Bsk][N]
    const size_t w_inq = b * 4 * pq, w_inB = b * 4 * pB, w_tq = b * 3 * pq, w_tB = b * 3 * pB, w_fl = b * 3 * pB;
    if (w_inq + w_inB + w_tq + w_tB + w_fl != ws_words_bfv_multiply(N, L, Bsk, b))
        return fail(CAT_ERR_LOGIC, "workspace formula out of step");
    u64 *ws = (u64 *)workspace(ctx, ws_words_bfv_multiply(N, L, Bsk, b) * sizeof(u64));
    if (!ws)
        return workspace_status(ctx);
    u64 *inq = ws, *inB = inq + w_inq, *tq = inB + w_inB, *tB = tq + w_tq, *fl = tB + w_tB;
    // steps (1)-(3), evaluator.cpp:452-490: per input polynomial, lazy NTT in q; lift to Bsk, lazy NTT in Bsk.
    // fastbconv_m_tilde + sm_mrq run as one pass over the coefficients of x, then of y, where they lie (the lifted
    // [Bsk + 1] residues are never stored); the same pass leaves the copy of the q limbs that is transformed in place.
    for (int half = 0; half < 2; ++half)
    {
        BconvArgs place{};
        place.grp = 2;
        place.out_obj_stride = 4 * pB;
        place.copy_out = inq + half * 2 * pq;
        place.copy_stride = pq;
        place.copy_obj_stride = 4 * pq;
        if ((st = conv_q_to_Bsk(ctx, lv, (const u64 *)(half ? d_y : d_x), pq, inB + half * 2 * pB, pB, PRE_MT, false, b * 2, s, 1,
                                nullptr, 0, &place)) != CAT_OK)
            return st;
    }
    if ((st = launch_ntt(ctx, true, inq, lv.map_q, L, batch * 4, true, s)) != CAT_OK)
        return st;
    if ((st = launch_ntt(ctx, true, inB, lv.map_Bsk, (uint32_t)Bsk, batch * 4, true, s)) != CAT_OK)
        return st;
    // step (4), :493-541
    {
        TensorMapArgs g{};
        g.mods = ctx->d_mods;
        g.log_n = ctx->log_n;
        g.x = inq;
        g.y = inq + 2 * pq;
        g.out = tq;
        g.map = lv.map_q;
        g.n_limbs = L;
        g.in_stride = 4 * pq;
        g.total = b * pq;
        hipLaunchKernelGGL(tensor_map_kernel, dim3(blocks_for(g.total)), dim3(256), 0, s, g);
        g.x = inB;
        g.y = inB + 2 * pB;
        g.out = tB;
        g.map = lv.map_Bsk;
        g.n_limbs = (u32)Bsk;
        g.in_stride = 4 * pB;
        g.total = b * pB;
        hipLaunchKernelGGL(tensor_map_kernel, dim3(blocks_for(g.total)), dim3(256), 0, s, g);
        CAT_HIP(hipGetLastError());
    }
    // step (5), :545-546
    if ((st = launch_ntt(ctx, false, tq, lv.map_q, L, batch * 3, true, s)) != CAT_OK)
        return st;
    if ((st = launch_ntt(ctx, false, tB, lv.map_Bsk, (uint32_t)Bsk, batch * 3, true, s)) != CAT_OK)
        return st;
    // steps (6)-(8), :549-566: multiply by t, fast_floor, fastbconv_sk
    if ((st = run_fast_floor(ctx, lv, tq, pq, tB, pB, fl, pB, true, b * 3, s)) != CAT_OK)
        return st;
    return run_fastbconv_sk(ctx, lv, fl, pB, (u64 *)d_out, pq, b * 3, s);
}

} // extern "C"
