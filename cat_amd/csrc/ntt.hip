// Batched negacyclic NTT / inverse NTT for gfx950 (MI355X).
//
// Replaces util::ntt_negacyclic_harvey[_lazy] (util/ntt.cpp:393-436, util/dwthandler.h:94-192) and
// util::inverse_ntt_negacyclic_harvey[_lazy] (util/ntt.cpp:438-474, util/dwthandler.h:203-357).
// The butterfly network and twiddle order are SEAL's; only the schedule is new:
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
// Two arithmetic flavours share the kernels (policy template parameter):
//   Exact : SEAL's Harvey butterflies with the 64-bit Shoup product (lazy range K = 2q).  Every intermediate
//           value, hence every *_lazy output, is the reference's exact representative.  Used for the C ABI's
//           lazy calls.
//   Fast  : the same network with the split-Shoup product of modarith.cuh (r < 4q, lazy range K = 4q) and the
//           same output contracts (forward lazy < 4q, inverse lazy < 2q, otherwise canonical).  Values agree
//           with Exact modulo q everywhere, so canonical outputs are bit-identical; it needs ~25 % fewer VALU
//           cycles per butterfly, which is what bounds these kernels.  Used for fully
//           reduced transforms and by every internal caller (key switching, rescale, BFV multiply).
//
// HBM traffic is 16 B/coefficient per pass (two passes); twiddles come from L2.
#include "context.hpp"
#include "rows8.cuh"
#include <cstdlib>
#include <type_traits>

// The tilings below are sized for gfx950: 160 KB of LDS per CU (the 8192-word column tiles take 70 KB per workgroup, the
// streaming row passes 64 KB at two workgroups per CU) and 64-lane waves.  Other targets are not supported.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "cat_amd's kernels are written for gfx950 (MI355X): build with --offload-arch=gfx950"
#endif

namespace
{
    constexpr int ROW_LEN = 256;      // N2
    constexpr int ROW_PITCH = 272;    // 256 + 256/16 padding words (bank-conflict-free transposes)
    constexpr int TILE_WORDS = 4096;
#ifndef ROWS8_TARGET_BLOCKS
#define ROWS8_TARGET_BLOCKS 8192 // poly-streaming row passes: workgroups aimed at (4096 and 16384 measured 1 % slower, 1024 and 2048 6 and 3 %)
#endif

    // streaming accesses of the poly-streaming row passes (data touched once per launch): non-temporal hints
#define NT_LOAD(p) __builtin_nontemporal_load(p)
#define NT_STORE(v, p) __builtin_nontemporal_store((v), (p))
    typedef u64 u64x2 __attribute__((ext_vector_type(2)));
    __device__ __forceinline__ ulonglong2 NT_LOAD2(const u64 *p)
    {
        const u64x2 v = NT_LOAD(reinterpret_cast<const u64x2 *>(p));
        ulonglong2 r;
        r.x = v.x;
        r.y = v.y;
        return r;
    }
    __device__ __forceinline__ void NT_STORE2(const ulonglong2 &w, u64 *p)
    {
        u64x2 v;
        v.x = w.x;
        v.y = w.y;
        NT_STORE(v, reinterpret_cast<u64x2 *>(p));
    }

    struct NttArgs
    {
        u64 *data;
        const NttMod *mods;
        const void *tw; // Exact: MulOp [n_mod][n];  Fast: u64 [n_mod][3][n] (w0 | w1 | q0,q1)
        LimbMap map;
        u32 n_limbs;
        u32 n_polys;
        u32 log_n;
        int lazy;
        u64 poly_stride; // words between consecutive objects (>= n_limbs * n)
        u32 limb_group;  // limb j uses modulus map.id[j / limb_group]
        u32 skip_diag;   // skip limbs with j % limb_group == j / limb_group (already transformed)
        // fused modulus raise of key switching (evaluator.cpp:2643-2651), forward only: when src != nullptr the
        // first pass reads limb (j % limb_group) of object `poly` of src (coefficient form, canonical modulo
        // prime j % limb_group), reduces it modulo the limb's own prime when that one is smaller, and writes the
        // transformed limb to data as usual
        const u64 *src;
        u64 src_stride; // words between consecutive src objects
        u32 src_kind;   // 0 raise, 1 round-and-fix (context.hpp NttFuse)
        u32 src_qidx;
        // fused epilogue of the rows pass (NttFuse): active when ep_dest != nullptr
        const u64 *ep_other;
        u64 *ep_dest;
        u64 ep_other_b, ep_other_s, ep_dest_b, ep_dest_s;
        u32 ep_S, ep_acc, ep_qidx;
        const u64 *half_mod;  // [kdim][kdim]: (q_a >> 1) mod q_b
        const MulOp *inv_mod; // [kdim][kdim]: q_a^-1 mod q_b
        u32 kdim;
        // second "round" source and two-term epilogue (NttFuse, relinearize + rescale in one transform)
        const u64 *src2;
        u64 src2_stride;
        u32 src2_qidx;
        u32 ep_qidx2;
        const u64 *ep_other2;
        u64 ep_other2_b, ep_other2_s;
    };

    struct Where
    {
        u64 offset; // word offset of the limb-polynomial
        u32 mod;    // modulus index
        u32 tile;
        bool skip;
        u64 src_offset; // fused raise: word offset of the source limb and its prime
        u32 src_mod;
        u32 poly, limb;
    };

    // Row passes (POLY_MAJOR = false): limb is the slowest index so that concurrently running workgroups share one
    // prime's row twiddles in L2 (1.5 MB per prime at N = 2^16).  Column passes (POLY_MAJOR = true) need 6 KB of
    // twiddles per prime, staged in LDS, and walk the buffer front to back instead -- tile, limb, polynomial: the
    // workgroups in flight then cover one contiguous window rather than the same offsets of many objects a power of two
    // apart, which HBM serves 8-10 % faster (profiles/r03_pass_shapes.txt: 5.13 -> 5.55 TB/s for this tile shape;
    // config 2 forward 14.26 -> 13.93 ms, metric op +0.9 %, same box)
    template <bool POLY_MAJOR = false>
    __device__ __forceinline__ Where locate(const NttArgs &a, u32 b, u32 tiles)
    {
        Where w;
        w.tile = b % tiles;
        const u32 rest = b / tiles;
        const u32 poly = POLY_MAJOR ? rest / a.n_limbs : rest % a.n_polys, limb = POLY_MAJOR ? rest % a.n_limbs : rest / a.n_polys;
        const u32 grp = limb / a.limb_group;
        w.mod = a.map.id[grp];
        w.skip = a.skip_diag && (limb - grp * a.limb_group == grp);
        w.offset = (u64)poly * a.poly_stride + ((u64)limb << a.log_n);
        w.poly = poly;
        w.limb = limb;
        w.src_mod = limb - grp * a.limb_group;
        w.src_offset = (u64)poly * a.src_stride + ((u64)w.src_mod << a.log_n);
        return w;
    }

    // LDS hand-off between lanes of ONE wave (the rows kernels: a wave owns its 4 rows of the tile): LDS operations
    // of a wave execute in order, so only the compiler has to be kept from reordering and the writes waited for;
    // no s_barrier, the four waves of a workgroup drift freely and cover each other's memory waits.
    __device__ __forceinline__ void wave_lds_sync()
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // x >= K ? x - K : x for x < 2K
    __device__ __forceinline__ u64 guard(u64 x, u64 K, u64 negK)
    {
        const u64 t = add64(x, negK);
        return x >= K ? t : x;
    }

    // Input side of a forward first pass: the data itself, or values derived from another buffer (NttFuse).
    struct Source
    {
        const u64 *p;
        u64 q, ratio1;   // the limb's own prime
        u64 Q, Qratio1;  // kind 1: prime of the source values
        u64 half, fix;
        int mode;        // 0 plain, 1 reduce modulo q, 2 round-and-fix
        bool reduce2;
        bool close;      // mode 1: the source prime is below 2q, one conditional subtraction reduces
        __device__ __forceinline__ u64 derive(u64 v) const
        {
            if (mode == 1)
                v = close ? guard(v, q, 0 - q) : barrett64(v, q, ratio1);
            else if (mode == 2)
            {
                v = barrett64(v + half, Q, Qratio1);
                if (reduce2)
                    v = barrett64(v, q, ratio1);
                v += fix;
            }
            return v;
        }
        __device__ __forceinline__ u64 operator()(u32 i) const { return derive(p[i]); }
        // 16 values at once: every load is issued before the (wave-uniform) mode is looked at, so the loads stay
        // one batch with one wait instead of sixteen dependent round trips
        template <class Index>
        __device__ __forceinline__ void load16(u64 (&x)[16], Index index) const
        {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                x[e] = p[index(e)];
            if (mode != 0)
            {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    x[e] = derive(x[e]);
            }
        }
    };

    // wide_input: the transform takes operands below 4q (Wide column pass), so values modulo a prime q_J <= 4q go in
    // as they are -- the transform is linear modulo q and every consumer reduces its output, only the class counts
    __device__ __forceinline__ Source source(const NttArgs &a, const Where &w, u64 extra, bool wide_input)
    {
        Source s;
        s.q = s.ratio1 = s.Q = s.Qratio1 = s.half = s.fix = 0;
        s.reduce2 = false;
        s.close = false;
        if (!a.src)
        {
            s.p = a.data + w.offset + extra;
            s.mode = 0;
            return s;
        }
        const NttMod m = a.mods[w.mod];
        s.q = m.q;
        s.ratio1 = m.ratio1;
        if (a.src_kind == 0)
        {
            s.p = a.src + w.src_offset + extra;
            const u64 qs = a.mods[w.src_mod].q;
            s.mode = (qs > m.q && !(wide_input && qs <= (m.q << 2))) ? 1 : 0; // copy when q_J <= q_I (evaluator.cpp:2643)
            s.close = qs < (m.q << 1);
        }
        else
        {
            const NttMod Q = a.mods[a.src_qidx];
            s.p = a.src + (u64)w.poly * a.src_stride + extra;
            s.Q = Q.q;
            s.Qratio1 = Q.ratio1;
            s.half = Q.q >> 1;
            s.fix = m.q - a.half_mod[a.src_qidx * a.kdim + w.mod];
            s.reduce2 = Q.q > m.q;
            s.mode = 2;
        }
        return s;
    }

    // the second "round" source of a dual transform: src2 + poly * src2_stride, values modulo prime src2_qidx
    __device__ __forceinline__ Source source2(const NttArgs &a, const Where &w, u64 extra)
    {
        Source s;
        const NttMod m = a.mods[w.mod], Q = a.mods[a.src2_qidx];
        s.q = m.q;
        s.ratio1 = m.ratio1;
        s.p = a.src2 + (u64)w.poly * a.src2_stride + extra;
        s.Q = Q.q;
        s.Qratio1 = Q.ratio1;
        s.half = Q.q >> 1;
        s.fix = m.q - a.half_mod[a.src2_qidx * a.kdim + w.mod];
        s.reduce2 = Q.q > m.q;
        s.close = false;
        s.mode = 2;
        return s;
    }

    // DUAL: x = round(src; Q) * Q^-1 + round(src2; Q2), both "round" sources of NttFuse (< 3q, canonical product)
    template <class Index>
    __device__ __forceinline__ void load16_dual(u64 (&x)[16], const Source &in1, const Source &in2, const MulOp &inv,
                                                Index index)
    {
        u64 y[16];
#pragma unroll
        for (int e = 0; e < 16; ++e)
        {
            x[e] = in1.p[index(e)];
            y[e] = in2.p[index(e)];
        }
        // both sources are canonical here (full inverse transforms), so (v + Q/2) mod Q is one conditional subtraction,
        // and so is the reduction modulo q when Q < 2q (wave-uniform choices)
        const u64 q = in1.q, nq = 0 - q;
        const bool close1 = in1.Q < (q << 1), close2 = in2.Q < (q << 1);
#pragma unroll
        for (int e = 0; e < 16; ++e)
        {
            u64 a = guard(x[e] + in1.half, in1.Q, 0 - in1.Q), b = guard(y[e] + in2.half, in2.Q, 0 - in2.Q);
            if (in1.reduce2)
                a = close1 ? guard(a, q, nq) : barrett64(a, q, in1.ratio1);
            if (in2.reduce2)
                b = close2 ? guard(b, q, nq) : barrett64(b, q, in2.ratio1);
            x[e] = mul_shoup(a + in1.fix, inv.op, inv.quo, q) + (b + in2.fix);
        }
    }

    // ---- arithmetic policies ----------------------------------------------------------------------------
    struct Exact
    {
        static constexpr int WIDE = 0;
        static constexpr bool INV_WIDE = false;
        typedef MulOp Tw;
        typedef const MulOp *Table;
        struct Mod
        {
            u64 q, K, negK, nq;
            MulOp inv_n, inv_n_w;
        };
        __device__ static __forceinline__ Mod mod(const NttMod &m)
        {
            Mod r;
            r.q = m.q;
            r.K = m.q << 1;
            r.negK = 0 - r.K;
            r.nq = 0 - m.q;
            r.inv_n = m.inv_n;
            r.inv_n_w = m.inv_n_w;
            return r;
        }
        __device__ static __forceinline__ Table table(const void *tw, u32 mod, u32 log_n, bool)
        {
            return reinterpret_cast<const MulOp *>(tw) + ((size_t)mod << log_n);
        }
        template <int CNT>
        __device__ static __forceinline__ void load(Table t, u32 idx, Tw (&w)[CNT])
        {
#pragma unroll
            for (int g = 0; g < CNT; ++g)
                w[g] = t[idx + g];
        }
        __device__ static __forceinline__ u64 mul(u64 x, const Tw &w, const Mod &m)
        {
            return mul_lazy(x, w, m.q);
        }
        __device__ static __forceinline__ u64 mul_add(u64 x, const Tw &w, const Mod &m, u64 a)
        {
            return add64(a, mul_lazy(x, w, m.q));
        }
        __device__ static __forceinline__ Tw inv_n(const Mod &m) { return m.inv_n; }
        __device__ static __forceinline__ Tw inv_n_w(const Mod &m) { return m.inv_n_w; }
        // forward values are < 2K = 4q, inverse values < K = 2q
        __device__ static __forceinline__ u64 finish_fwd(u64 x, const Mod &m, bool lazy)
        {
            return lazy ? x : guard(guard(x, m.K, m.negK), m.q, m.nq);
        }
        __device__ static __forceinline__ u64 finish_inv(u64 x, const Mod &m, bool lazy)
        {
            return lazy ? x : guard(x, m.q, m.nq);
        }
    };

    struct Fast
    {
        static constexpr int WIDE = 0; // 0: guard in every stage, 1: policy Wide, 2: policy Wider
        static constexpr bool INV_WIDE = false;
        typedef MulOp3 Tw;
        struct Table
        {
            const u64 *w0, *w1, *qq;
        };
        struct Mod
        {
            u64 q, K, negK, nq;
            MulOp3 inv_n, inv_n_w;
            u32 red_r, red_shift; // reduce_below_2q (InvWide)
        };
        __device__ static __forceinline__ Mod mod(const NttMod &m)
        {
            Mod r;
            r.q = m.q;
            r.K = m.q << 2;
            r.negK = 0 - r.K;
            r.nq = 0 - m.q;
            r.inv_n = m.inv_n3;
            r.inv_n_w = m.inv_n_w3;
            r.red_r = m.red_r;
            r.red_shift = m.red_shift;
            return r;
        }
        // the inverse table is stored shifted by one entry (entry j = inv_root_powers[j + 1]) so that every
        // group of 2, 4 or 8 consecutive twiddles starts 16-byte aligned
        __device__ static __forceinline__ Table table(const void *tw, u32 mod, u32 log_n, bool inverse)
        {
            const u64 *b = reinterpret_cast<const u64 *>(tw) + ((size_t)mod * 3 << log_n) - (inverse ? 1 : 0);
            Table t;
            t.w0 = b;
            t.w1 = b + ((size_t)1 << log_n);
            t.qq = b + ((size_t)2 << log_n);
            return t;
        }
        template <int CNT>
        __device__ static __forceinline__ void load(const Table &t, u32 idx, Tw (&w)[CNT])
        {
            if (CNT == 1)
            {
                const u64 qq = t.qq[idx];
                w[0].w0 = t.w0[idx];
                w[0].w1 = t.w1[idx];
                w[0].q0 = (u32)qq;
                w[0].q1 = (u32)(qq >> 32);
            }
            else
            {
#pragma unroll
                for (int g = 0; g < CNT; g += 2)
                {
                    const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(t.w0 + idx + g);
                    const ulonglong2 b = *reinterpret_cast<const ulonglong2 *>(t.w1 + idx + g);
                    const ulonglong2 c = *reinterpret_cast<const ulonglong2 *>(t.qq + idx + g);
                    w[g].w0 = a.x, w[g].w1 = b.x, w[g].q0 = (u32)c.x, w[g].q1 = (u32)(c.x >> 32);
                    w[g + 1].w0 = a.y, w[g + 1].w1 = b.y, w[g + 1].q0 = (u32)c.y, w[g + 1].q1 = (u32)(c.y >> 32);
                }
            }
        }
        __device__ static __forceinline__ u64 mul(u64 x, const Tw &w, const Mod &m)
        {
            return mul_split(x, w, m.nq);
        }
        __device__ static __forceinline__ u64 mul_add(u64 x, const Tw &w, const Mod &m, u64 a)
        {
            return mul_split_add(x, w, m.nq, a);
        }
        __device__ static __forceinline__ Tw inv_n(const Mod &m) { return m.inv_n; }
        __device__ static __forceinline__ Tw inv_n_w(const Mod &m) { return m.inv_n_w; }
        // forward values are < 2K = 8q, inverse values < K = 4q; contracts: lazy forward < 4q, lazy inverse < 2q
        __device__ static __forceinline__ u64 finish_fwd(u64 x, const Mod &m, bool lazy)
        {
            x = guard(x, m.K, m.negK);
            return lazy ? x : guard(guard(x, m.q << 1, m.nq << 1), m.q, m.nq);
        }
        __device__ static __forceinline__ u64 finish_inv(u64 x, const Mod &m, bool lazy)
        {
            x = guard(x, m.q << 1, m.nq << 1);
            return lazy ? x : guard(x, m.q, m.nq);
        }
    };

    // Fast arithmetic with wide lazy ranges, forward transforms whose primes are all below 2^60: a butterfly adds at
    // most 4q to the range of its inputs and the split-Shoup product takes any 64-bit operand, so values may grow
    // to 16q < 2^64 and ONE conditional subtraction of 8q every second stage replaces the per-stage guard
    // (-2 of 18 instructions per butterfly).  Range in units of q, column pass from inputs < 4: stages 0,1,2
    // unguarded (< 8, 12, 16), then odd stages subtract 8q first (< 8 -> < 12) and even stages do not (< 16).  Row
    // passes accept < 16q: even local stages subtract, odd ones do not; results < 16q are brought to the usual
    // contract (< 4q lazy, canonical otherwise) by the final reduction.
    struct Wide : Fast
    {
        static constexpr int WIDE = 1;
    };
    // The same with NO guard at all, forward transforms whose primes all have 36 to 57 bits -- every default coefficient
    // modulus of the reference (CoeffModulus::BFVDefault, util/globals.cpp:25-70: at most 56 bits) and CKKS chains of
    // such primes.  Sixteen stages add at most 64q to inputs below 4q: 68q < 2^64 for q < 2^57, so the values simply grow
    // and the last pass brings them down with one estimated-quotient reduction (reduce_below_2q takes any 64-bit
    // operand when the prime has 36 bits or more): 14 instead of 16 instructions per butterfly.  A first-pass-only
    // result holds values below 36q.
    struct Wider : Fast
    {
        static constexpr int WIDE = 2;
    };
    // Fast arithmetic with wide lazy ranges for INVERSE transforms whose primes all have 36 to 60 bits.  A Gentleman-
    // Sande butterfly doubles the range of its sum and its product comes out below 4q whatever went in, so with values
    // below 4q an even stage needs no reduction (sums below 8q, differences taken against 4q) and the odd stage after
    // it (sums below 16q <= 2^64, differences against 8q) brings its sums below 2q with the estimated-quotient
    // reduction (modarith.cuh reduce_below_2q, 5 instructions): 2.5 instructions per butterfly and stage instead of
    // the 4 of a conditional subtraction in every stage.  The merged last stage multiplies both outputs and needs no
    // reduction at all; it takes its difference against 8q when it sits at an odd stage.
    struct InvWide : Fast
    {
        static constexpr bool INV_WIDE = true;
    };
    // reduction mode of inverse global stage G: 0 conditional subtraction of K (Exact: 2q, Fast: 4q), 1 none, 2 estimated
    template <class P>
    __device__ __forceinline__ constexpr int inv_mode(int G)
    {
        return !P::INV_WIDE ? 0 : (G & 1) ? 2 : 1;
    }
    // guard mode of a forward stage: 0 none, 1 subtract K = 2q (Exact) / 4q (Fast), 2 subtract 8q
    template <class P>
    __device__ __forceinline__ constexpr int col_guard(int global_stage)
    {
        return P::WIDE == 2 ? 0 : !P::WIDE ? 1 : ((global_stage & 1) && global_stage >= 3) ? 2 : 0;
    }
    template <class P>
    __device__ __forceinline__ constexpr int row_guard(int local_stage)
    {
        return P::WIDE == 2 ? 0 : !P::WIDE ? 1 : (local_stage & 1) ? 0 : 2;
    }

    // end of a forward transform: from the policy's range to the contract (lazy: below 4q; else canonical)
    template <class P>
    __device__ __forceinline__ u64 finish_fwd_any(u64 x, const typename P::Mod &m, bool lazy)
    {
        if constexpr (P::WIDE == 2)
        {
            x = reduce_below_2q(x, m.red_r, m.red_shift, m.nq);
            return lazy ? x : guard(x, m.q, m.nq);
        }
        else
            return P::finish_fwd(P::WIDE ? guard(x, m.K << 1, m.negK << 1) : x, m, lazy);
    }

    // Column passes, Fast policy: the first log2(N1) <= 8 forward stages use table entries 1 .. N1-1, the last
    // log2(N1) inverse stages the entries N - N1 + 1 .. N - 1 -- at most 255 twiddles per limb, the same for every
    // column and polynomial.  A workgroup copies them to LDS (6 KB) and the butterflies read them with ds_read
    // instead of sending 30 vector loads per thread to L2.
    __device__ __forceinline__ Fast::Table stage_col_twiddles(u64 *twl, const Fast::Table &g, u32 log_n, bool inverse)
    {
        const u32 n = 1u << log_n;
        const u32 first = inverse ? n - 256 + 1 : 0; // logical index of twl[0]
        const u32 e = threadIdx.x;
        if (e < 256)
        {
            twl[e] = g.w0[first + e];
            twl[256 + e] = g.w1[first + e];
            twl[512 + e] = g.qq[first + e];
        }
        Fast::Table t;
        t.w0 = twl - first;
        t.w1 = twl + 256 - first;
        t.qq = twl + 512 - first;
        return t;
    }
    __device__ __forceinline__ Exact::Table stage_col_twiddles(u64 *, const Exact::Table &g, u32, bool) { return g; }

    // ---- 16-point register networks ---------------------------------------------------------------
#ifndef TW_CHUNK
#define TW_CHUNK 8 // 4 saves ~7 VGPRs in the forward kernels but measured slower on MI355X
#endif
    // Forward (Cooley-Tukey) local stage S: register gap 8 >> S, 2^S twiddles tw[(base << S) + g].
    template <class P, int S, int GM = 1>
    __device__ __forceinline__ void fwd_stage16(u64 (&x)[16], const typename P::Table &tw, u32 base,
                                                const typename P::Mod &m)
    {
        constexpr int gap = 8 >> S;
        constexpr int CNT = 1 << S, CH = CNT < TW_CHUNK ? CNT : TW_CHUNK; // twiddles are fetched CH at a time
#pragma unroll
        for (int g0 = 0; g0 < CNT; g0 += CH)
        {
            typename P::Tw w[CH];
            P::template load<CH>(tw, (base << S) + g0, w);
#pragma unroll
            for (int g = g0; g < g0 + CH; ++g)
            {
#pragma unroll
                for (int j = 0; j < gap; ++j)
                {
                    // X' = u + w*Y with u folded into the multiply chain; Y' = u - w*Y + K = 2u + K - X'
                    const int i0 = g * 2 * gap + j;
                    const u64 u = GM == 0 ? x[i0] : GM == 1 ? guard(x[i0], m.K, m.negK) : guard(x[i0], m.K << 1, m.negK << 1);
                    const u64 s = P::mul_add(x[i0 + gap], w[g - g0], m, u);
                    x[i0] = s;
                    x[i0 + gap] = shl1_add64(u, m.K) - s;
                }
            }
        }
    }

    // Inverse (Gentleman-Sande) local stage S: register gap 1 << S, 8 >> S twiddles tw[root0 + g].
    template <class P, int S, int G = 0, int CHUNK = TW_CHUNK>
    __device__ __forceinline__ void inv_stage16(u64 (&x)[16], const typename P::Table &tw, u32 root0,
                                                const typename P::Mod &m)
    {
        constexpr int MODE = inv_mode<P>(G);
        constexpr int gap = 1 << S;
        constexpr int CNT = 8 >> S, CH = CNT < CHUNK ? CNT : CHUNK; // twiddles are fetched CH at a time
#pragma unroll
        for (int g0 = 0; g0 < CNT; g0 += CH)
        {
            typename P::Tw w[CH];
            P::template load<CH>(tw, root0 + g0, w);
#pragma unroll
            for (int g = g0; g < g0 + CH; ++g)
            {
#pragma unroll
                for (int j = 0; j < gap; ++j)
                {
                    const int i0 = g * 2 * gap + j;
                    const u64 u = x[i0], v = x[i0 + gap];
                    if constexpr (MODE == 0)
                        x[i0] = guard(add64(u, v), m.K, m.negK);
                    else if constexpr (MODE == 1)
                        x[i0] = add64(u, v);
                    else
                        x[i0] = reduce_below_2q(add64(u, v), m.red_r, m.red_shift, m.nq);
                    x[i0 + gap] = P::mul(add64(u, MODE == 2 ? m.K << 1 : m.K) - v, w[g - g0], m);
                }
            }
        }
    }

    // Final inverse stage (single twiddle, merged n^-1): dwthandler.h:274-315.
    template <class P, int S, int G = 0>
    __device__ __forceinline__ void inv_last_stage16(u64 (&x)[16], const typename P::Mod &m)
    {
        constexpr int gap = 1 << S;
        const typename P::Tw wn = P::inv_n(m), wnw = P::inv_n_w(m);
#pragma unroll
        for (int g = 0; g < (8 >> S); ++g)
        {
#pragma unroll
            for (int j = 0; j < gap; ++j)
            {
                const int i0 = g * 2 * gap + j;
                if constexpr (std::is_base_of<Fast, P>::value)
                {
                    // operands below K = 4q and the split product takes any 64-bit word: no reduction before either
                    // multiply (u + v and u + K - v stay below 8q < 2^64)
                    const u64 u = x[i0], v = x[i0 + gap];
                    x[i0] = P::mul(add64(u, v), wn, m);
                    x[i0 + gap] = P::mul(add64(u, inv_mode<P>(G) == 2 ? m.K << 1 : m.K) - v, wnw, m);
                }
                else
                {
                    const u64 u = guard(x[i0], m.K, m.negK), v = x[i0 + gap];
                    x[i0] = P::mul(guard(add64(u, v), m.K, m.negK), wn, m);
                    x[i0 + gap] = P::mul(add64(u, m.K) - v, wnw, m);
                }
            }
        }
    }

    // ---- forward, pass 1: columns ----------------------------------------------------------------------
    // one column tile (C columns x N1 rows) of the limb-polynomial w points at; tw = the staged column twiddles
    template <class P, int LOGN1, bool DUAL, int TW = TILE_WORDS>
    __device__ __forceinline__ void fwd_cols_tile(const NttArgs &a, const Where &w, const typename P::Mod &m,
                                                  const typename P::Table &tw, u64 *lds)
    {
        constexpr int K1 = LOGN1 - 4;          // stages done in round 1
        constexpr int N1 = 1 << LOGN1;
        constexpr int C = TW / N1;             // columns per workgroup (TW / 16 threads)
        constexpr int SUB = 16 >> K1;          // consecutive rows per thread in round 1
        u64 *p = a.data + w.offset + w.tile * C;
        const Source in = source(a, w, w.tile * C, P::WIDE != 0);

        const u32 c = threadIdx.x % C, tau = threadIdx.x / C;
        u64 x[16];
        if (K1 > 0)
        {
            auto idx = [&](int e) { return ((e >> (4 - K1)) * 16 + tau * SUB + (e & (SUB - 1))) * ROW_LEN + c; };
            if (DUAL)
                load16_dual(x, in, source2(a, w, w.tile * C), a.inv_mod[a.src_qidx * a.kdim + w.mod], idx);
            else
                in.load16(x, idx);
            if (std::is_base_of<Fast, P>::value)
                __syncthreads(); // staged twiddles visible
            if (K1 > 0) fwd_stage16<P, 0, col_guard<P>(0)>(x, tw, 1, m);
            if (K1 > 1) fwd_stage16<P, 1, col_guard<P>(1)>(x, tw, 1, m);
            if (K1 > 2) fwd_stage16<P, 2, col_guard<P>(2)>(x, tw, 1, m);
            if (K1 > 3) fwd_stage16<P, 3, col_guard<P>(3)>(x, tw, 1, m);
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
            auto idx = [&](int l) { return (tau * 16 + l) * ROW_LEN + c; };
            if (DUAL)
                load16_dual(x, in, source2(a, w, w.tile * C), a.inv_mod[a.src_qidx * a.kdim + w.mod], idx);
            else
                in.load16(x, idx);
            if (std::is_base_of<Fast, P>::value)
                __syncthreads();
        }
        const u32 base = (1u << K1) + tau;
        fwd_stage16<P, 0, col_guard<P>(K1 + 0)>(x, tw, base, m);
        fwd_stage16<P, 1, col_guard<P>(K1 + 1)>(x, tw, base, m);
        fwd_stage16<P, 2, col_guard<P>(K1 + 2)>(x, tw, base, m);
        fwd_stage16<P, 3, col_guard<P>(K1 + 3)>(x, tw, base, m);
#pragma unroll
        for (int l = 0; l < 16; ++l)
            p[(tau * 16 + l) * ROW_LEN + c] = x[l];
    }

    // TW = words per workgroup tile (TW / 16 threads).  8192 at N = 2^16 makes the row segments a workgroup reads and
    // writes 256 bytes long instead of 128: HBM gives that shape 5.04 TB/s against 4.77 (profiles/r02_access_shape_rates.txt)
    template <class P, int LOGN1, bool DUAL = false, int TW = TILE_WORDS>
    __global__ __launch_bounds__(TW / 16, 4) void ntt_fwd_cols(NttArgs a)
    {
        static_assert(TW / 16 >= 256, "stage_col_twiddles copies with the first 256 threads of the workgroup");
        static_assert((TW + 3 * 256) * sizeof(u64) <= 80 * 1024, "two column workgroups per CU must fit 160 KB of LDS");
        __shared__ u64 lds[TW];
        const Where w = locate<true>(a, blockIdx.x, ROW_LEN / (TW >> LOGN1));
        if (w.skip)
            return;
        const typename P::Mod m = P::mod(a.mods[w.mod]);
        __shared__ u64 twl[3 * 256];
        const typename P::Table tw = stage_col_twiddles(twl, P::table(a.tw, w.mod, a.log_n, false), a.log_n, false);
        fwd_cols_tile<P, LOGN1, DUAL, TW>(a, w, m, tw, lds);
    }

    // ---- forward, pass 2: rows (16 rows of 256 per workgroup) -------------------------------------------
    template <class P>
    __device__ __forceinline__ void fwd_rows_tile(const NttArgs &a, const Where &w, const typename P::Mod &m,
                                                  const typename P::Table &tw, u64 *lds)
    {
        const u32 n1 = 1u << (a.log_n - 8);
        const u32 r = threadIdx.x >> 4, t = threadIdx.x & 15;
        const u32 row = w.tile * 16 + r;
        u64 *p = a.data + w.offset + (size_t)row * ROW_LEN;
        u64 *s = lds + r * ROW_PITCH;

        u64 x[16];
#pragma unroll
        for (int h = 0; h < 16; ++h)
            x[h] = p[h * 16 + t];
        const u32 base1 = n1 + row;
        fwd_stage16<P, 0, row_guard<P>(0)>(x, tw, base1, m);
        fwd_stage16<P, 1, row_guard<P>(1)>(x, tw, base1, m);
        fwd_stage16<P, 2, row_guard<P>(2)>(x, tw, base1, m);
        fwd_stage16<P, 3, row_guard<P>(3)>(x, tw, base1, m);
#pragma unroll
        for (int h = 0; h < 16; ++h)
            s[h * 17 + t] = x[h];
        wave_lds_sync();
#pragma unroll
        for (int l = 0; l < 16; ++l)
            x[l] = s[t * 17 + l];
        const u32 base2 = base1 * 16 + t;
        fwd_stage16<P, 0, row_guard<P>(4)>(x, tw, base2, m);
        fwd_stage16<P, 1, row_guard<P>(5)>(x, tw, base2, m);
        fwd_stage16<P, 2, row_guard<P>(6)>(x, tw, base2, m);
        fwd_stage16<P, 3, row_guard<P>(7)>(x, tw, base2, m);
#pragma unroll
        for (int l = 0; l < 16; ++l)
            x[l] = finish_fwd_any<P>(x[l], m, a.lazy != 0);
        wave_lds_sync();
#pragma unroll
        for (int l = 0; l < 16; ++l)
            s[t * 17 + l] = x[l];
        wave_lds_sync();
        if (a.ep_dest)
        {
            const u32 b = w.poly / a.ep_S, si = w.poly - b * a.ep_S;
            const u64 in_limb = ((u64)w.limb << a.log_n) + (size_t)row * ROW_LEN;
            const u64 *other = a.ep_other + b * a.ep_other_b + si * a.ep_other_s + in_limb;
            u64 *dest = a.ep_dest + b * a.ep_dest_b + si * a.ep_dest_s + in_limb;
            const MulOp inv = a.inv_mod[a.ep_qidx * a.kdim + w.mod];
            const u64 q4 = m.q << 2;
            if (a.ep_other2)
            {
                // two-term form: dest = ((other + other2 * Q2^-1) + 4q - t) * Q^-1, other and other2 canonical
                const u64 *other2 = a.ep_other2 + b * a.ep_other2_b + si * a.ep_other2_s + in_limb;
                const MulOp inv2 = a.inv_mod[a.ep_qidx2 * a.kdim + w.mod];
#pragma unroll
                for (int h = 0; h < 16; ++h)
                {
                    const u64 base_v = other[h * 16 + t] + mul_shoup(other2[h * 16 + t], inv2.op, inv2.quo, m.q);
                    dest[h * 16 + t] = mul_shoup(base_v + q4 - s[h * 17 + t], inv.op, inv.quo, m.q);
                }
                return;
            }
#pragma unroll
            for (int h = 0; h < 16; ++h)
            {
                u64 v = mul_shoup(other[h * 16 + t] + q4 - s[h * 17 + t], inv.op, inv.quo, m.q);
                if (a.ep_acc)
                    v = add_mod(v, dest[h * 16 + t], m.q);
                dest[h * 16 + t] = v;
            }
            return;
        }
#pragma unroll
        for (int h = 0; h < 16; ++h)
            p[h * 16 + t] = s[h * 17 + t];
    }

    template <class P>
    __global__ __launch_bounds__(256) void ntt_fwd_rows(NttArgs a)
    {
        __shared__ u64 lds[16 * ROW_PITCH];
        const u32 n1 = 1u << (a.log_n - 8);
        const Where w = locate(a, blockIdx.x, n1 / 16);
        if (w.skip)
            return;
        const typename P::Mod m = P::mod(a.mods[w.mod]);
        const typename P::Table tw = P::table(a.tw, w.mod, a.log_n, false);
        fwd_rows_tile<P>(a, w, m, tw, lds);
    }

    // ---- forward, pass 2, poly-streaming form (Fast policy, no fused epilogue) ------------------------------
    // A workgroup owns 8 rows of one limb at one position, stages their twiddles in LDS once (rows8.cuh) and
    // streams the rows of `n_polys / chunks` polynomials through them, two polynomials in flight per thread and
    // the next pair already requested.  Against ntt_fwd_rows: no per-tile twiddle traffic from L2 (3x the data
    // volume there) and nothing but the data stream in the vector-memory queue.
    // ND: polynomials in flight per thread (they share every twiddle read from LDS).  ND = 1 halves the bytes a thread
    // has in flight, which the access-only skeleton likes (profiles/r03_pass_shapes.txt: 5.41 against 4.96 TB/s), but the
    // kernel itself takes 3.5 % longer with it (config 2 forward 14.60 against 14.08 ms, same box): ND = 2 stays
    template <int WIDE, int ND = 2>
    __global__ __launch_bounds__(256, 2) void ntt_fwd_rows8(NttArgs a, u32 chunks)
    {
        __shared__ __align__(16) u64 lds[RM_TW_WORDS + RM_X_WORDS];
        const u32 n1 = 1u << (a.log_n - 8), tiles = n1 / RM_ROWS;
        const u32 chunk = blockIdx.x % chunks, rest = blockIdx.x / chunks;
        const u32 tile = rest % tiles, limb = rest / tiles;
        const u32 grp = limb / a.limb_group;
        if (a.skip_diag && limb - grp * a.limb_group == grp)
            return;
        const u32 mod = a.map.id[grp];
        const u64 q = a.mods[mod].q, nq = 0 - q;
        const u32 red_r = a.mods[mod].red_r, red_shift = a.mods[mod].red_shift;
        const bool below_2q = WIDE != 0 && red_r != 0; // what rm_rows hands back
        rm_stage_twiddles(lds, reinterpret_cast<const u64 *>(a.tw) + ((size_t)mod * 3 << a.log_n), a.log_n, tile);
        __syncthreads();
        const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const u32 rl = wave * 2 + (lane >> 5), t = lane & 31;
        const u64 *tw = lds + rl * 256;
        u64 *xs = lds + RM_TW_WORDS + rl * 256;
        u64 *base = a.data + ((u64)limb << a.log_n) + (size_t)(tile * RM_ROWS + rl) * 256;
        const bool lazy = a.lazy != 0;

        const u32 groups = (a.n_polys + ND - 1) / ND;
        u64 xn[ND][8];
        auto request = [&](u32 g) {
#pragma unroll
            for (int n = 0; n < ND; ++n)
            {
                const u32 poly = ND * g + n < a.n_polys ? ND * g + n : ND * g; // a ragged last group repeats its first
                const u64 *src = base + (u64)poly * a.poly_stride + t;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    xn[n][e] = NT_LOAD(src + e * 32);
            }
        };
        u32 pi = chunk;
        if (pi < groups)
            request(pi);
        for (; pi < groups; pi += chunks)
        {
            u64 x[ND][8];
#pragma unroll
            for (int n = 0; n < ND; ++n)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    x[n][e] = xn[n][e];
            if (pi + chunks < groups)
                request(pi + chunks);
            rm_rows<ND, WIDE>(x, tw, xs, t, q, red_r, red_shift);
            // third exchange: from 8 consecutive coefficients per thread to 16-byte pieces that are consecutive
            // across the lanes of a row, so that every store instruction writes whole 512-byte runs
#pragma unroll
            for (int n = 0; n < ND; ++n)
            {
                if (ND * pi + n >= a.n_polys)
                    break;
                rm_wave_sync();
#pragma unroll
                for (int u = 0; u < 8; u += 2)
                {
                    ulonglong2 v;
                    v.x = x[n][u], v.y = x[n][u + 1];
                    if (!lazy)
                    {
                        v.x = guard(below_2q ? v.x : guard(v.x, q << 1, nq << 1), q, nq);
                        v.y = guard(below_2q ? v.y : guard(v.y, q << 1, nq << 1), q, nq);
                    }
                    // 16-byte pieces, piece index XOR (t >> 1) & 3: eight consecutive lanes of a ds_write_b128 (64-byte
                    // stride: slots 0 and 4 of 8 only) land in eight different 16-byte slots, the pieces stay whole (no
                    // swap of their halves on the way out) and the 16-byte reads below are consecutive across lanes up
                    // to a constant XOR
                    *reinterpret_cast<ulonglong2 *>(xs + ((8 * t + u) ^ (((t >> 1) & 3) << 1))) = v;
                }
                rm_wave_sync();
                u64 *o = base + (u64)(ND * pi + n) * a.poly_stride;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                {
                    const u32 e = (i * 32 + t) * 2, sw = ((t >> 3) & 3) << 1; // writer lane 8 i + (t >> 2)
                    const ulonglong2 w = *reinterpret_cast<const ulonglong2 *>(xs + (e ^ sw));
                    NT_STORE2(w, o + e);
                }
            }
        }
    }

    // ---- inverse, pass 1, poly-streaming form: the mirror image of ntt_fwd_rows8 ----------------------------
    template <bool INV_WIDE>
    __global__ __launch_bounds__(256, 2) void ntt_inv_rows8(NttArgs a, u32 chunks)
    {
        __shared__ __align__(16) u64 lds[RM_TW_WORDS + RM_X_WORDS];
        const u32 n1 = 1u << (a.log_n - 8), tiles = n1 / RM_ROWS;
        const u32 chunk = blockIdx.x % chunks, rest = blockIdx.x / chunks;
        const u32 tile = rest % tiles, limb = rest / tiles;
        const u32 grp = limb / a.limb_group;
        if (a.skip_diag && limb - grp * a.limb_group == grp)
            return;
        const u32 mod = a.map.id[grp];
        const u64 q = a.mods[mod].q;
        rm_stage_twiddles_inv(lds, reinterpret_cast<const u64 *>(a.tw) + ((size_t)mod * 3 << a.log_n), a.log_n, tile);
        __syncthreads();
        const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const u32 rl = wave * 2 + (lane >> 5), t = lane & 31;
        const u64 *tw = lds + rl * 256;
        u64 *xs = lds + RM_TW_WORDS + rl * 256;
        u64 *base = a.data + ((u64)limb << a.log_n) + (size_t)(tile * RM_ROWS + rl) * 256;
        // out-of-place form (NttFuse::src on an inverse transform): the rows come from another buffer of the same
        // limb layout, objects src_stride apart
        const u64 *in_base = a.src + ((u64)limb << a.log_n) + (size_t)(tile * RM_ROWS + rl) * 256;

        // loads: 16-byte pieces consecutive across the lanes of a row, turned into 8 consecutive coefficients per
        // thread through LDS (element e at e ^ ((e >> 5) & 7), as in the forward kernel's store side)
        auto fetch = [&](ulonglong2 (&raw)[2][4], u32 pi) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
            {
                const u32 poly = 2 * pi + n < a.n_polys ? 2 * pi + n : 2 * pi;
                const u64 *src = a.src ? in_base + (u64)poly * a.src_stride : base + (u64)poly * a.poly_stride;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    raw[n][i] = NT_LOAD2(src + (i * 32 + t) * 2);
            }
        };
        const u32 pairs = (a.n_polys + 1) / 2;
        ulonglong2 rn[2][4];
        u32 pi = chunk;
        if (pi < pairs)
            fetch(rn, pi);
        for (; pi < pairs; pi += chunks)
        {
            u64 x[2][8];
#pragma unroll
            for (int n = 0; n < 2; ++n)
            {
                rm_wave_sync();
#pragma unroll
                for (int i = 0; i < 4; ++i)
                {
                    // the mirror image of the forward kernel's last exchange: whole 16-byte pieces, piece index XOR
                    // (lane >> 2) & 3 of the lane that reads them, 8 i + (t >> 2): a ds_read_b128 serves 16 lanes
                    // at a time (a ds_write_b128 eight, hence (lane >> 1) & 3 in the forward kernel)
                    const u32 e = (i * 32 + t) * 2, sw = ((2 * i + (t >> 4)) & 3) << 1;
                    *reinterpret_cast<ulonglong2 *>(xs + (e ^ sw)) = rn[n][i];
                }
                rm_wave_sync();
#pragma unroll
                for (int u = 0; u < 8; u += 2)
                {
                    const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(xs + ((8 * t + u) ^ (((t >> 2) & 3) << 1)));
                    x[n][u] = v.x, x[n][u + 1] = v.y;
                }
            }
            if (pi + chunks < pairs)
                fetch(rn, pi + chunks);
            rm_rows_inv<2, INV_WIDE>(x, tw, xs, t, q, a.mods[mod].red_r, a.mods[mod].red_shift);
            const bool two = 2 * pi + 1 < a.n_polys;
#pragma unroll
            for (int n = 0; n < 2; ++n)
            {
                if (n == 1 && !two)
                    break;
                u64 *o = base + (u64)(2 * pi + n) * a.poly_stride + t;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    NT_STORE(x[n][e], o + e * 32);
            }
        }
    }

    // ---- inverse, pass 1: rows ---------------------------------------------------------------------------
    template <class P>
    __global__ __launch_bounds__(256) void ntt_inv_rows(NttArgs a)
    {
        __shared__ u64 lds[16 * ROW_PITCH];
        const u32 n = 1u << a.log_n, n1 = n >> 8;
        const Where w = locate(a, blockIdx.x, n1 / 16);
        if (w.skip)
            return;
        const typename P::Mod m = P::mod(a.mods[w.mod]);
        const typename P::Table tw = P::table(a.tw, w.mod, a.log_n, true);
        const u32 r = threadIdx.x >> 4, t = threadIdx.x & 15;
        const u32 row = w.tile * 16 + r;
        u64 *p = a.data + w.offset + (size_t)row * ROW_LEN;
        u64 *s = lds + r * ROW_PITCH;

        u64 x[16];
        const u64 *pin = a.src ? a.src + (u64)w.poly * a.src_stride + ((u64)w.limb << a.log_n) + (size_t)row * ROW_LEN : p;
#pragma unroll
        for (int h = 0; h < 16; ++h)
            s[h * 17 + t] = pin[h * 16 + t];
        wave_lds_sync();
#pragma unroll
        for (int l = 0; l < 16; ++l)
            x[l] = s[t * 17 + l];
        // global stages 0..3: root index n - (n >> T) + 1 + group
        const u32 blk = row * 16 + t;
        inv_stage16<P, 0, 0>(x, tw, n - (n >> 0) + 1 + blk * 8, m);
        inv_stage16<P, 1, 1>(x, tw, n - (n >> 1) + 1 + blk * 4, m);
        inv_stage16<P, 2, 2>(x, tw, n - (n >> 2) + 1 + blk * 2, m);
        inv_stage16<P, 3, 3>(x, tw, n - (n >> 3) + 1 + blk * 1, m);
        wave_lds_sync();
#pragma unroll
        for (int l = 0; l < 16; ++l)
            s[t * 17 + l] = x[l];
        wave_lds_sync();
#pragma unroll
        for (int h = 0; h < 16; ++h)
            x[h] = s[h * 17 + t];
        // global stages 4..7
        inv_stage16<P, 0, 4>(x, tw, n - (n >> 4) + 1 + row * 8, m);
        inv_stage16<P, 1, 5>(x, tw, n - (n >> 5) + 1 + row * 4, m);
        inv_stage16<P, 2, 6>(x, tw, n - (n >> 6) + 1 + row * 2, m);
        inv_stage16<P, 3, 7>(x, tw, n - (n >> 7) + 1 + row * 1, m);
#pragma unroll
        for (int h = 0; h < 16; ++h)
            p[h * 16 + t] = x[h];
    }

    // ---- inverse, pass 2: columns --------------------------------------------------------------------------
    template <class P, int LOGN1, int TW = TILE_WORDS>
    __global__ __launch_bounds__(TW / 16, 4) void ntt_inv_cols(NttArgs a)
    {
        constexpr int K1 = LOGN1 - 4;
        constexpr int N1 = 1 << LOGN1;
        constexpr int C = TW / N1;
        constexpr int SUB = 16 >> K1;
        static_assert(TW / 16 >= 256, "stage_col_twiddles copies with the first 256 threads of the workgroup");
        static_assert((TW + 3 * 256) * sizeof(u64) <= 80 * 1024, "two column workgroups per CU must fit 160 KB of LDS");
        __shared__ u64 lds[TW];

        const u32 n = 1u << a.log_n;
        const Where w = locate<true>(a, blockIdx.x, ROW_LEN / C);
        if (w.skip)
            return;
        const typename P::Mod m = P::mod(a.mods[w.mod]);
        __shared__ u64 twl[3 * 256];
        const typename P::Table tw = stage_col_twiddles(twl, P::table(a.tw, w.mod, a.log_n, true), a.log_n, true);
        u64 *p = a.data + w.offset + w.tile * C;
        const u32 c = threadIdx.x % C, tau = threadIdx.x / C;

        u64 x[16];
#pragma unroll
        for (int l = 0; l < 16; ++l)
            x[l] = p[(tau * 16 + l) * ROW_LEN + c];
        if (std::is_base_of<Fast, P>::value)
            __syncthreads(); // staged twiddles visible
        // global stages 8..11
        inv_stage16<P, 0, 8>(x, tw, n - (n >> 8) + 1 + tau * 8, m);
        inv_stage16<P, 1, 9>(x, tw, n - (n >> 9) + 1 + tau * 4, m);
        inv_stage16<P, 2, 10>(x, tw, n - (n >> 10) + 1 + tau * 2, m);
        if (K1 == 0)
            inv_last_stage16<P, 3, 11>(x, m);
        else
            inv_stage16<P, 3, 11>(x, tw, n - (n >> 11) + 1 + tau, m);
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
            if (K1 == 4) inv_stage16<P, 0, 12>(x, tw, n - (n >> 12) + 1, m);
            if (K1 == 3) inv_stage16<P, 1, 12>(x, tw, n - (n >> 12) + 1, m);
            if (K1 == 4) inv_stage16<P, 1, 13>(x, tw, n - (n >> 13) + 1, m);
            if (K1 == 2) inv_stage16<P, 2, 12>(x, tw, n - (n >> 12) + 1, m);
            if (K1 == 3) inv_stage16<P, 2, 13>(x, tw, n - (n >> 13) + 1, m);
            if (K1 == 4) inv_stage16<P, 2, 14>(x, tw, n - (n >> 14) + 1, m);
            inv_last_stage16<P, 3, 11 + K1>(x, m);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e)
            x[e] = P::finish_inv(x[e], m, a.lazy != 0);
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
    template <class P, bool FWD>
    __global__ __launch_bounds__(256) void ntt_small(NttArgs a)
    {
        __shared__ u64 s[2048];
        const u32 n = 1u << a.log_n, half = n >> 1;
        const Where w = locate(a, blockIdx.x, 1);
        if (w.skip)
            return;
        const typename P::Mod m = P::mod(a.mods[w.mod]);
        const typename P::Table tw = P::table(a.tw, w.mod, a.log_n, !FWD);
        u64 *p = a.data + w.offset;
        if (FWD)
        {
            const Source in = source(a, w, 0, false);
            for (u32 i = threadIdx.x; i < n; i += blockDim.x)
                s[i] = in(i);
        }
        else
            for (u32 i = threadIdx.x; i < n; i += blockDim.x)
                s[i] = p[i];
        __syncthreads();
        typename P::Tw tv[1];
        if (FWD)
        {
            u32 lg = a.log_n - 1; // log2(gap)
            for (u32 mm = 1; mm < n; mm <<= 1, --lg)
            {
                for (u32 j = threadIdx.x; j < half; j += blockDim.x)
                {
                    const u32 i = j >> lg, pos = j & ((1u << lg) - 1);
                    const u32 i0 = (i << (lg + 1)) + pos, i1 = i0 + (1u << lg);
                    P::template load<1>(tw, mm + i, tv);
                    const u64 u = guard(s[i0], m.K, m.negK), v = P::mul(s[i1], tv[0], m);
                    s[i0] = u + v;
                    s[i1] = u + m.K - v;
                }
                __syncthreads();
            }
            for (u32 i = threadIdx.x; i < n; i += blockDim.x)
                p[i] = P::finish_fwd(s[i], m, a.lazy != 0);
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
                    P::template load<1>(tw, n - 2 * mm + 1 + i, tv);
                    const u64 u = s[i0], v = s[i1];
                    s[i0] = guard(u + v, m.K, m.negK);
                    s[i1] = P::mul(u + m.K - v, tv[0], m);
                }
                __syncthreads();
            }
            const typename P::Tw wn = P::inv_n(m), wnw = P::inv_n_w(m);
            for (u32 j = threadIdx.x; j < half; j += blockDim.x)
            {
                const u64 u = guard(s[j], m.K, m.negK), v = s[j + half];
                s[j] = P::mul(guard(u + v, m.K, m.negK), wn, m);
                s[j + half] = P::mul(u + m.K - v, wnw, m);
            }
            __syncthreads();
            for (u32 i = threadIdx.x; i < n; i += blockDim.x)
                p[i] = P::finish_inv(s[i], m, a.lazy != 0);
        }
    }

    // ---- N = 4096, 8192 (written for 16384 too): the whole limb-polynomial in ONE workgroup, one trip over HBM ------
    // A limb of 32 / 64 / 128 KB fits the 160 KB of LDS, so these sizes need no second pass over memory: 16 B per
    // coefficient of HBM traffic, which is the algorithmic figure (the two-pass form moves 32).  Same butterfly network
    // and twiddle order as everything else here (dwthandler.h:94-357).  log2 N = K + 12: the forward transform runs
    // K stages on registers loaded at stride N/16, then three rounds of four stages at strides 256, 16 and 1 inside
    // blocks of 4096, 256 and 16 coefficients; the inverse is the mirror image.  Between rounds the data goes through
    // LDS in place (element i at i + i / 16: every access pattern of the three strides is bank-conflict free or 2-way
    // at worst), one workgroup barrier per round.  N = 16384 would run as 512 threads x 2 sets of 16 coefficients, one
    // workgroup per CU; it is bit-exact but slower than the two passes (launch_policy) and is not instantiated.  Twiddles come from L2
    // (workgroups are ordered limb-major: one prime's table at a time).  Fast-family policies only; the Exact policy,
    // the fused sources and epilogues stay on the two-pass kernels.
    template <int LOGN>
    struct OneCfg
    {
        static constexpr int N = 1 << LOGN, K = LOGN - 12;
        static constexpr int THREADS = LOGN == 12 ? 256 : 512;
        static constexpr int SETS = (N / 16) / THREADS;
        static constexpr int WAVES = LOGN == 14 ? 2 : 4; // per SIMD: LDS allows 4 / 2 / 1 workgroups per CU
        // the inverse kernel at N = 4096 spills 7 registers at 128 (a spill is HBM traffic): three workgroups per CU instead
        static constexpr int WAVES_INV = LOGN == 12 ? 3 : WAVES;
        static constexpr int LDS_WORDS = N + N / 16;
    };
    __device__ __forceinline__ u32 one_pad(u32 i) { return i + (i >> 4); }
    // padded LDS position of element e of a 16-point network at stride S (S = 1 or a multiple of 16) whose element 0 is
    // `first`: linear in e, so the sixteen accesses of a thread share one address register
    template <int S>
    __device__ __forceinline__ u32 one_at(u32 first_padded, int e)
    {
        return first_padded + e * (S == 1 ? 1 : S + S / 16);
    }

    template <class P, int G0>
    __device__ __forceinline__ void fwd_round16(u64 (&x)[16], const typename P::Table &tw, u32 base, const typename P::Mod &m)
    {
        fwd_stage16<P, 0, col_guard<P>(G0 + 0)>(x, tw, base, m);
        fwd_stage16<P, 1, col_guard<P>(G0 + 1)>(x, tw, base, m);
        fwd_stage16<P, 2, col_guard<P>(G0 + 2)>(x, tw, base, m);
        fwd_stage16<P, 3, col_guard<P>(G0 + 3)>(x, tw, base, m);
    }

    // one in-place round through LDS: 16-point networks at stride S inside blocks of 16 S, global stages G0 .. G0 + 3
    template <class P, int LOGN, int G0, int S>
    __device__ __forceinline__ void fwd_one_round(u64 *lds, const typename P::Table &tw, const typename P::Mod &m)
    {
        typedef OneCfg<LOGN> C;
#pragma unroll
        for (int set = 0; set < C::SETS; ++set)
        {
            const u32 v = set * C::THREADS + threadIdx.x, blk = v / S, r = v % S;
            const u32 at = one_pad(blk * 16 * S + r); // element e: blk * 16 S + e S + r
            u64 x[16];
#pragma unroll
            for (int e = 0; e < 16; ++e)
                x[e] = lds[one_at<S>(at, e)];
            fwd_round16<P, G0>(x, tw, (C::N >> 4) / S + blk, m);
#pragma unroll
            for (int e = 0; e < 16; ++e)
                lds[one_at<S>(at, e)] = x[e];
        }
    }

    // FUSED: with the sources and epilogues of NttFuse (the "round" source of mod-down and rescale, the dual source and
    // the one- and two-term finish of relinearize + rescale; not the key-switch raise, whose row pass is the fused
    // key-switch kernel): at N = 4096 and 8192 these steps, too, cross HBM once.  A separate instantiation so that the
    // plain transform keeps its registers.
    template <class P, int LOGN, bool FUSED = false>
    __global__ __launch_bounds__(OneCfg<LOGN>::THREADS, OneCfg<LOGN>::WAVES) void ntt_fwd_one(NttArgs a)
    {
        typedef OneCfg<LOGN> C;
        constexpr int N = C::N, K = C::K, T = C::THREADS, Q = N / 16;
        __shared__ u64 lds[C::LDS_WORDS];
        const Where w = locate(a, blockIdx.x, 1);
        if (w.skip)
            return;
        const typename P::Mod m = P::mod(a.mods[w.mod]);
        const typename P::Table tw = P::table(a.tw, w.mod, LOGN, false);
        u64 *p = a.data + w.offset;
        {
            u64 x[C::SETS][16];
#pragma unroll
            for (int set = 0; set < C::SETS; ++set)
            {
                if constexpr (FUSED)
                {
                    const Source in = source(a, w, 0, P::WIDE != 0);
                    auto idx = [&](int e) { return e * Q + set * T + threadIdx.x; };
                    if (a.src2)
                        load16_dual(x[set], in, source2(a, w, 0), a.inv_mod[a.src_qidx * a.kdim + w.mod], idx);
                    else
                        in.load16(x[set], idx);
                }
                else
                {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        x[set][e] = p[e * Q + set * T + threadIdx.x];
                }
            }
#pragma unroll
            for (int set = 0; set < C::SETS; ++set)
            {
                // registers at stride N/16 cover the whole polynomial: its first K stages, or (K = 0) the first full round
                if (K == 0)
                    fwd_round16<P, 0>(x[set], tw, 1, m);
                if (K >= 1)
                    fwd_stage16<P, 0, col_guard<P>(0)>(x[set], tw, 1, m);
                if (K >= 2)
                    fwd_stage16<P, 1, col_guard<P>(1)>(x[set], tw, 1, m);
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    lds[one_at<Q>(one_pad(set * T + threadIdx.x), e)] = x[set][e];
            }
        }
        __syncthreads();
        if (K > 0)
        {
            fwd_one_round<P, LOGN, K, 256>(lds, tw, m);
            __syncthreads();
        }
        fwd_one_round<P, LOGN, K + 4, 16>(lds, tw, m);
        __syncthreads();
        fwd_one_round<P, LOGN, K + 8, 1>(lds, tw, m);
        __syncthreads();
        const bool lazy = a.lazy != 0;
        if (FUSED && a.ep_dest)
        {
            // the finish steps of fwd_rows_tile on the transformed limb t (< 4q), which is not stored
            const u32 b = w.poly / a.ep_S, si = w.poly - b * a.ep_S;
            const u64 in_limb = (u64)w.limb << LOGN;
            const u64 *other = a.ep_other + b * a.ep_other_b + si * a.ep_other_s + in_limb;
            const u64 *other2 = a.ep_other2 ? a.ep_other2 + b * a.ep_other2_b + si * a.ep_other2_s + in_limb : nullptr;
            u64 *dest = a.ep_dest + b * a.ep_dest_b + si * a.ep_dest_s + in_limb;
            const MulOp inv = a.inv_mod[a.ep_qidx * a.kdim + w.mod];
            const MulOp inv2 = a.ep_other2 ? a.inv_mod[a.ep_qidx2 * a.kdim + w.mod] : inv;
            const u64 q4 = m.q << 2;
#pragma unroll
            for (int j = 0; j < N / (2 * T); ++j)
            {
                const u32 i = 2 * (j * T + threadIdx.x);
                const u64 t0 = finish_fwd_any<P>(lds[one_pad(i)], m, true), t1 = finish_fwd_any<P>(lds[one_pad(i + 1)], m, true);
                ulonglong2 o = *reinterpret_cast<const ulonglong2 *>(other + i), v;
                if (other2)
                {
                    const ulonglong2 o2 = *reinterpret_cast<const ulonglong2 *>(other2 + i);
                    o.x += mul_shoup(o2.x, inv2.op, inv2.quo, m.q);
                    o.y += mul_shoup(o2.y, inv2.op, inv2.quo, m.q);
                }
                v.x = mul_shoup(o.x + q4 - t0, inv.op, inv.quo, m.q);
                v.y = mul_shoup(o.y + q4 - t1, inv.op, inv.quo, m.q);
                if (a.ep_acc && !other2)
                {
                    const ulonglong2 d = *reinterpret_cast<const ulonglong2 *>(dest + i);
                    v.x = add_mod(v.x, d.x, m.q);
                    v.y = add_mod(v.y, d.y, m.q);
                }
                *reinterpret_cast<ulonglong2 *>(dest + i) = v;
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < N / (2 * T); ++j)
        {
            const u32 i = 2 * (j * T + threadIdx.x);
            ulonglong2 v;
            v.x = lds[one_pad(i)];
            v.y = lds[one_pad(i + 1)];
            v.x = finish_fwd_any<P>(v.x, m, lazy);
            v.y = finish_fwd_any<P>(v.y, m, lazy);
            *reinterpret_cast<ulonglong2 *>(p + i) = v;
        }
    }

    template <class P, int LOGN, int G0, int S>
    __device__ __forceinline__ void inv_one_round(u64 *lds, const typename P::Table &tw, const typename P::Mod &m)
    {
        typedef OneCfg<LOGN> C;
        constexpr u32 n = C::N;
#pragma unroll
        for (int set = 0; set < C::SETS; ++set)
        {
            const u32 v = set * C::THREADS + threadIdx.x, blk = v / S, r = v % S;
            const u32 at = one_pad(blk * 16 * S + r);
            u64 x[16];
#pragma unroll
            for (int e = 0; e < 16; ++e)
                x[e] = lds[one_at<S>(at, e)];
            // the first stage's eight twiddles in two fetches of four: with eight at once these kernels spill a few
            // registers at 128 VGPRs, and spills are HBM traffic (PMC: 1.22 x algorithmic instead of 1.03 x)
            inv_stage16<P, 0, G0 + 0, 4>(x, tw, n - (n >> (G0 + 0)) + 1 + blk * 8, m);
            inv_stage16<P, 1, G0 + 1>(x, tw, n - (n >> (G0 + 1)) + 1 + blk * 4, m);
            inv_stage16<P, 2, G0 + 2>(x, tw, n - (n >> (G0 + 2)) + 1 + blk * 2, m);
            inv_stage16<P, 3, G0 + 3>(x, tw, n - (n >> (G0 + 3)) + 1 + blk * 1, m);
#pragma unroll
            for (int e = 0; e < 16; ++e)
                lds[one_at<S>(at, e)] = x[e];
        }
    }

    template <class P, int LOGN>
    __global__ __launch_bounds__(OneCfg<LOGN>::THREADS, OneCfg<LOGN>::WAVES_INV) void ntt_inv_one(NttArgs a)
    {
        typedef OneCfg<LOGN> C;
        constexpr int N = C::N, K = C::K, T = C::THREADS, Q = N / 16;
        constexpr u32 n = N;
        __shared__ u64 lds[C::LDS_WORDS];
        const Where w = locate(a, blockIdx.x, 1);
        if (w.skip)
            return;
        const typename P::Mod m = P::mod(a.mods[w.mod]);
        const typename P::Table tw = P::table(a.tw, w.mod, LOGN, true);
        u64 *p = a.data + w.offset;
        // out-of-place form (NttFuse::src on an inverse transform): same limb layout, objects src_stride apart
        const u64 *pin = a.src ? a.src + (u64)w.poly * a.src_stride + ((u64)w.limb << LOGN) : p;
#pragma unroll
        for (int j = 0; j < N / (2 * T); ++j)
        {
            const u32 i = 2 * (j * T + threadIdx.x);
            const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(pin + i);
            lds[one_pad(i)] = v.x;
            lds[one_pad(i + 1)] = v.y;
        }
        __syncthreads();
        inv_one_round<P, LOGN, 0, 1>(lds, tw, m);
        __syncthreads();
        inv_one_round<P, LOGN, 4, 16>(lds, tw, m);
        __syncthreads();
        if (K > 0)
        {
            inv_one_round<P, LOGN, 8, 256>(lds, tw, m);
            __syncthreads();
        }
        const bool lazy = a.lazy != 0;
#pragma unroll
        for (int set = 0; set < C::SETS; ++set)
        {
            u64 x[16];
#pragma unroll
            for (int e = 0; e < 16; ++e)
                x[e] = lds[one_at<Q>(one_pad(set * T + threadIdx.x), e)];
            if (K == 0)
            {
                // stages 8 .. 11 at stride 256 = N/16: the last one carries n^-1 (dwthandler.h:274-315)
                inv_stage16<P, 0, 8, 4>(x, tw, n - (n >> 8) + 1, m);
                inv_stage16<P, 1, 9>(x, tw, n - (n >> 9) + 1, m);
                inv_stage16<P, 2, 10>(x, tw, n - (n >> 10) + 1, m);
                inv_last_stage16<P, 3, 11>(x, m);
            }
            if (K == 1)
                inv_last_stage16<P, 3, 12>(x, m);
            if (K == 2)
            {
                inv_stage16<P, 2, 12>(x, tw, n - (n >> 12) + 1, m);
                inv_last_stage16<P, 3, 13>(x, m);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e)
                p[e * Q + set * T + threadIdx.x] = P::finish_inv(x[e], m, lazy);
        }
    }

    // forward transforms of large batches (N = 2^16), plain or with the key-switch raise: column tiles twice as wide,
    // 14.7 against 15.1 ms on BASELINE config 2 and -2.7 % on the raise pass of the metric op (same box); 1024-thread
    // tiles of 64 columns measured slower (16.0 ms)
    inline bool wide_tiles(const NttArgs &a)
    {
        return (u64)a.n_limbs * a.n_polys >= 2048;
    }

    template <class P, int LOGN1>
    void launch_cols(bool fwd, const NttArgs &a, u32 blocks, hipStream_t s)
    {
        if (fwd && a.src2)
            hipLaunchKernelGGL((ntt_fwd_cols<P, LOGN1, true>), dim3(blocks), dim3(256), 0, s, a);
        else if (fwd)
        {
            if constexpr (LOGN1 == 8 && std::is_base_of<Fast, P>::value)
            {
                if (wide_tiles(a))
                {
                    hipLaunchKernelGGL((ntt_fwd_cols<P, LOGN1, false, 2 * TILE_WORDS>), dim3(blocks / 2), dim3(512), 0, s, a);
                    return;
                }
            }
            hipLaunchKernelGGL((ntt_fwd_cols<P, LOGN1>), dim3(blocks), dim3(256), 0, s, a);
        }
        else
        {
            // (with the Fast policy the inverse column pass is bound by its instruction stream, 93 % VALU-busy, and wide
            // tiles change nothing; InvWide's 93 instead of 103 instructions per coefficient leave it on its accesses)
            typedef typename std::conditional<P::WIDE != 0, Fast, P>::type Q;
            if constexpr (LOGN1 == 8 && P::INV_WIDE)
            {
                if (wide_tiles(a))
                {
                    hipLaunchKernelGGL((ntt_inv_cols<Q, LOGN1, 2 * TILE_WORDS>), dim3(blocks / 2), dim3(512), 0, s, a);
                    return;
                }
            }
            hipLaunchKernelGGL((ntt_inv_cols<Q, LOGN1>), dim3(blocks), dim3(256), 0, s, a);
        }
    }

    template <class P>
    cat_status launch_policy(cat_ctx *ctx, bool forward, NttArgs &a, hipStream_t s, bool first_only)
    {
        const uint64_t n_lp = (uint64_t)a.n_limbs * a.n_polys;
        if (ctx->log_n < 12)
        {
            if (n_lp > 0x7fffffffu)
                return cat::fail(CAT_ERR_LOGIC, "invalid parameters");
            const u32 threads = ctx->n / 2 >= 256 ? 256 : 64;
            typedef typename std::conditional<P::WIDE != 0, Fast, P>::type Q;
            if (forward)
                hipLaunchKernelGGL((ntt_small<Q, true>), dim3((u32)n_lp), dim3(threads), 0, s, a);
            else
                hipLaunchKernelGGL((ntt_small<Q, false>), dim3((u32)n_lp), dim3(threads), 0, s, a);
            CAT_HIP(hipGetLastError());
            return CAT_OK;
        }
        // N = 4096 and 8192, plain transforms (an inverse one may read another buffer): one trip over HBM.  Same box, 1 GiB
        // of 55-bit limbs (profiles/r03_ntt_by_ring_size.txt): forward 0.950 -> 0.730 ms at N = 4096 and 0.859 -> 0.758 ms at
        // N = 8192, inverse 0.835 -> 0.818 and 0.898 -> 0.892 ms.  N = 16384 (128 KB of LDS, one workgroup per CU: its load,
        // compute and store phases do not overlap with anything) measured 0.980 / 1.147 ms against 0.848 / 0.882 ms for
        // the two passes and stays on them.
        // Forward transforms with a fused "round" / dual source or a fused finish (mod-down, rescale, relinearize + rescale)
        // take the FUSED instantiation; the key-switch raise (first pass only) stays on the column kernel.
        if constexpr (std::is_base_of<Fast, P>::value)
        if (ctx->log_n <= 13 && forward && !first_only && (a.ep_dest || a.src) && (!a.src || a.src_kind == 1) && !a.skip_diag)
        {
            if (n_lp > 0x7fffffffu)
                return cat::fail(CAT_ERR_LOGIC, "invalid parameters");
            if (ctx->log_n == 12)
                hipLaunchKernelGGL((ntt_fwd_one<P, 12, true>), dim3((u32)n_lp), dim3(OneCfg<12>::THREADS), 0, s, a);
            else
                hipLaunchKernelGGL((ntt_fwd_one<P, 13, true>), dim3((u32)n_lp), dim3(OneCfg<13>::THREADS), 0, s, a);
            CAT_HIP(hipGetLastError());
            return CAT_OK;
        }
        if constexpr (std::is_base_of<Fast, P>::value)
        if (ctx->log_n <= 13 && !first_only && !a.ep_dest && !a.src2 &&
            (forward ? !a.src : (!a.src || (a.src_kind == 0 && a.limb_group == 1))))
        {
            if (n_lp > 0x7fffffffu)
                return cat::fail(CAT_ERR_LOGIC, "invalid parameters");
            typedef typename std::conditional<P::WIDE != 0, Fast, P>::type QI; // Wide is a forward policy
            const dim3 grid((u32)n_lp);
            if (forward)
                switch (ctx->log_n)
                {
                case 12: hipLaunchKernelGGL((ntt_fwd_one<P, 12>), grid, dim3(OneCfg<12>::THREADS), 0, s, a); break;
                default: hipLaunchKernelGGL((ntt_fwd_one<P, 13>), grid, dim3(OneCfg<13>::THREADS), 0, s, a); break;
                }
            else
                switch (ctx->log_n)
                {
                case 12: hipLaunchKernelGGL((ntt_inv_one<QI, 12>), grid, dim3(OneCfg<12>::THREADS), 0, s, a); break;
                default: hipLaunchKernelGGL((ntt_inv_one<QI, 13>), grid, dim3(OneCfg<13>::THREADS), 0, s, a); break;
                }
            CAT_HIP(hipGetLastError());
            return CAT_OK;
        }
        const u32 logn1 = ctx->log_n - 8;
        const u32 n1 = 1u << logn1;
        const uint64_t row_blocks = n_lp * (n1 / 16);
        const uint64_t col_blocks = n_lp * (ROW_LEN / (TILE_WORDS / n1));
        if (row_blocks > 0x7fffffffu || col_blocks > 0x7fffffffu)
            return cat::fail(CAT_ERR_LOGIC, "invalid parameters");
        auto cols = [&]() {
            switch (logn1)
            {
            case 4: launch_cols<P, 4>(forward, a, (u32)col_blocks, s); break;
            case 5: launch_cols<P, 5>(forward, a, (u32)col_blocks, s); break;
            case 6: launch_cols<P, 6>(forward, a, (u32)col_blocks, s); break;
            case 7: launch_cols<P, 7>(forward, a, (u32)col_blocks, s); break;
            default: launch_cols<P, 8>(forward, a, (u32)col_blocks, s); break;
            }
        };
        if (forward)
        {
            cols();
            if (first_only)
                ;
            else if (std::is_base_of<Fast, P>::value && !a.ep_dest && a.n_polys >= 4)
            {
                // poly-streaming row pass: workgroups = limbs x 8-row positions x chunks of the polynomials
                const u64 groups = (u64)a.n_limbs * (n1 / RM_ROWS);
                u32 chunks = (u32)((ROWS8_TARGET_BLOCKS + groups - 1) / groups);
                const u32 max_chunks = (a.n_polys + 1) / 2 / 2; // at least two pairs per workgroup
                if (chunks > max_chunks)
                    chunks = max_chunks;
                if (chunks < 1)
                    chunks = 1;
                hipLaunchKernelGGL((ntt_fwd_rows8<P::WIDE, 2>), dim3((u32)(groups * chunks)), dim3(256), 0, s, a, chunks);
            }
            else
                hipLaunchKernelGGL((ntt_fwd_rows<P>), dim3((u32)row_blocks), dim3(256), 0, s, a);
        }
        else
        {
            if (std::is_base_of<Fast, P>::value && a.n_polys >= 4)
            {
                const u64 groups = (u64)a.n_limbs * (n1 / RM_ROWS);
                u32 chunks = (u32)((ROWS8_TARGET_BLOCKS + groups - 1) / groups);
                const u32 max_chunks = (a.n_polys + 1) / 2 / 2;
                if (chunks > max_chunks)
                    chunks = max_chunks;
                if (chunks < 1)
                    chunks = 1;
                hipLaunchKernelGGL(ntt_inv_rows8<P::INV_WIDE>, dim3((u32)(groups * chunks)), dim3(256), 0, s, a, chunks);
            }
            else
                hipLaunchKernelGGL((ntt_inv_rows<typename std::conditional<P::WIDE != 0, Fast, P>::type>),
                                   dim3((u32)row_blocks), dim3(256), 0, s, a);
            cols();
        }
        CAT_HIP(hipGetLastError());
        return CAT_OK;
    }
} // namespace

namespace cat
{
    int ntt_wide(const cat_ctx *ctx, const LimbMap &map, uint32_t n_groups)
    {
        int level = 2;
        for (uint32_t i = 0; i < n_groups; ++i)
        {
            const u64 q = ctx->hmods[map.id[i]].q;
            if (q >> 60)
                return 0;
            if ((q >> 57) || !(q >> 35))
                level = 1;
        }
        return level;
    }

    // every prime of 36 to 60 bits: the inverse transform's wide ranges (policy InvWide) need 16q <= 2^64 and the
    // estimated-quotient reduction
    static bool ntt_inv_wide(const cat_ctx *ctx, const LimbMap &map, uint32_t n_groups)
    {
        for (uint32_t i = 0; i < n_groups; ++i)
        {
            const u64 q = ctx->hmods[map.id[i]].q;
            if ((q >> 60) || !(q >> 35))
                return false;
        }
        return true;
    }

    cat_status launch_ntt(cat_ctx *ctx, bool forward, u64 *d_data, const LimbMap &map, uint32_t n_limbs,
                          uint32_t n_polys, bool lazy, hipStream_t s, bool exact)
    {
        return launch_ntt_ex(ctx, forward, d_data, map, n_limbs, n_polys, lazy, (u64)n_limbs << ctx->log_n, 1, false,
                             s, exact, nullptr);
    }

    cat_status launch_ntt_ex(cat_ctx *ctx, bool forward, u64 *d_data, const LimbMap &map, uint32_t n_limbs,
                             uint32_t n_polys, bool lazy, u64 poly_stride, uint32_t limb_group, bool skip_diag,
                             hipStream_t s, bool exact, const NttFuse *fuse)
    {
        if (n_limbs == 0 || n_polys == 0)
            return CAT_OK;
        NttArgs a;
        a.data = d_data;
        a.mods = ctx->d_nttmods;
        a.map = map;
        a.n_limbs = n_limbs;
        a.n_polys = n_polys;
        a.log_n = ctx->log_n;
        a.lazy = lazy ? 1 : 0;
        a.poly_stride = poly_stride;
        a.limb_group = limb_group;
        a.skip_diag = skip_diag ? 1 : 0;
        a.src = nullptr;
        a.src_stride = 0;
        a.src_kind = a.src_qidx = 0;
        a.ep_other = nullptr;
        a.ep_dest = nullptr;
        a.ep_other_b = a.ep_other_s = a.ep_dest_b = a.ep_dest_s = 0;
        a.ep_S = 1;
        a.ep_acc = a.ep_qidx = 0;
        a.half_mod = ctx->d_half_mod;
        a.inv_mod = ctx->d_inv_mod;
        a.kdim = ctx->k;
        a.src2 = nullptr;
        a.src2_stride = 0;
        a.src2_qidx = a.ep_qidx2 = 0;
        a.ep_other2 = nullptr;
        a.ep_other2_b = a.ep_other2_s = 0;
        if (fuse)
        {
            // inverse transforms take only the out-of-place source (src_kind 0, N >= 4096, one modulus per limb)
            if ((!forward && (fuse->ep_dest || fuse->first_pass_only || fuse->src_kind != 0 || ctx->log_n < 12 ||
                              limb_group != 1)) ||
                (fuse->ep_dest && (ctx->log_n < 12 || !lazy)) ||
                (fuse->first_pass_only && (ctx->log_n < 12 || exact || fuse->ep_dest)))
                return fail(CAT_ERR_LOGIC, "unsupported fused transform");
            a.src = fuse->src;
            a.src_stride = fuse->src_stride;
            a.src_kind = (u32)fuse->src_kind;
            a.src_qidx = fuse->src_qidx;
            a.ep_other = fuse->ep_other;
            a.ep_dest = fuse->ep_dest;
            a.ep_other_b = fuse->ep_other_b;
            a.ep_other_s = fuse->ep_other_s;
            a.ep_dest_b = fuse->ep_dest_b;
            a.ep_dest_s = fuse->ep_dest_s;
            a.ep_S = fuse->ep_S;
            a.ep_acc = (u32)fuse->ep_acc;
            a.ep_qidx = fuse->ep_qidx;
            // the dual form: a second round source and the two-term epilogue come together, on a forward transform
            // whose first source is a round source
            // (the two-term epilogue also runs alone, over data prepared by the caller: the BGV tail)
            if ((fuse->src2 && !fuse->ep_other2) ||
                (fuse->ep_other2 && (!forward || !fuse->ep_dest || fuse->ep_acc || exact)) ||
                (fuse->src2 && (fuse->src_kind != 1 || !fuse->src)))
                return fail(CAT_ERR_LOGIC, "unsupported fused transform");
            a.src2 = fuse->src2;
            a.src2_stride = fuse->src2_stride;
            a.src2_qidx = fuse->src2_qidx;
            a.ep_qidx2 = fuse->ep_qidx2;
            a.ep_other2 = fuse->ep_other2;
            a.ep_other2_b = fuse->ep_other2_b;
            a.ep_other2_s = fuse->ep_other2_s;
        }
        // N = 2 has odd-aligned twiddle groups nowhere, but keep the tiny sizes on the exact path: nothing to gain
        if (exact || ctx->log_n < 2)
        {
            a.tw = forward ? (const void *)ctx->d_tw_fwd : (const void *)ctx->d_tw_inv;
            return launch_policy<Exact>(ctx, forward, a, s, false);
        }
        a.tw = forward ? (const void *)ctx->d_tw3_fwd : (const void *)ctx->d_tw3_inv;
        if (forward && ctx->log_n >= 12)
        {
            const int level = ntt_wide(ctx, map, (n_limbs + limb_group - 1) / limb_group);
            if (level == 2)
                return launch_policy<Wider>(ctx, forward, a, s, fuse && fuse->first_pass_only);
            if (level == 1)
                return launch_policy<Wide>(ctx, forward, a, s, fuse && fuse->first_pass_only);
        }
        if (!forward && ctx->log_n >= 12 && ntt_inv_wide(ctx, map, (n_limbs + limb_group - 1) / limb_group))
            return launch_policy<InvWide>(ctx, forward, a, s, false);
        return launch_policy<Fast>(ctx, forward, a, s, fuse && fuse->first_pass_only);
    }
} // namespace cat
