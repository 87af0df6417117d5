// Device-side 64-bit modular arithmetic for gfx950 (wave64, 32-bit integer multipliers).
// Semantics follow SEAL's word arithmetic so that lazy representatives match the reference bit for
// bit: util/uintarithsmallmod.cuh:188-226 (Barrett-128), :312-333 (Barrett-64), :416-429 (Shoup lazy).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;
namespace cat
{
    using ::u32;
}

// {operand, quotient = floor(operand * 2^64 / q)}  (MultiplyUIntModOperand, uintarithsmallmod.cuh:363-393)
struct __attribute__((aligned(16))) MulOp
{
    u64 op;
    u64 quo;
};

// Per-modulus constants, 64 bytes.
struct __attribute__((aligned(64))) DevMod
{
    u64 q;
    u64 ratio0; // floor(2^128/q) low word  (modulus.cpp:87-98)
    u64 ratio1; // floor(2^128/q) high word
    u64 inv_n_op; // n^-1 mod q (util/ntt.cpp:289-295)
    u64 inv_n_quo;
    u64 inv_n_w_op; // inv_root_powers[n-1] * n^-1 mod q (mul_root_scalar, util/ntt.h:50-56)
    u64 inv_n_w_quo;
    u64 r64_quo; // floor((2^64 mod q) 2^64 / q): reduce128()
};

__device__ __forceinline__ u64 mulhi64(u64 a, u64 b)
{
    return __umul64hi(a, b);
}

// x*w mod q in [0, 2q) for any 64-bit x (multiply_uint_mod_lazy)
__device__ __forceinline__ u64 mul_lazy(u64 x, u64 w, u64 wq, u64 q)
{
    return w * x - mulhi64(x, wq) * q;
}

__device__ __forceinline__ u64 mul_lazy(u64 x, const MulOp &w, u64 q)
{
    return w.op * x - mulhi64(x, w.quo) * q;
}

__device__ __forceinline__ u64 csub(u64 x, u64 q)
{
    return x >= q ? x - q : x;
}

// multiply_uint_mod(x, MultiplyUIntModOperand) -> [0, q)
__device__ __forceinline__ u64 mul_shoup(u64 x, u64 w, u64 wq, u64 q)
{
    return csub(mul_lazy(x, w, wq, q), q);
}

__device__ __forceinline__ u64 barrett64(u64 x, u64 q, u64 ratio1)
{
    u64 r = x - mulhi64(x, ratio1) * q;
    return csub(r, q);
}

__device__ __forceinline__ u64 barrett128(u64 lo, u64 hi, u64 q, u64 r0, u64 r1)
{
    u64 carry = mulhi64(lo, r0);
    u64 t2lo = lo * r1, t2hi = mulhi64(lo, r1);
    u64 tmp1 = t2lo + carry;
    u64 tmp3 = t2hi + (tmp1 < t2lo);
    u64 t3lo = hi * r0, t3hi = mulhi64(hi, r0);
    u64 s = tmp1 + t3lo;
    carry = t3hi + (s < tmp1);
    u64 quot = hi * r1 + tmp3 + carry;
    u64 r = lo - quot * q;
    return csub(r, q);
}

// hi 2^64 + lo mod q as (hi (2^64 mod q) mod q) + (lo mod q): a Shoup product with the per-prime constant and a Barrett-64
// step, 13 VALU instructions fewer than Barrett-128 and the same canonical residue.  q < 2^62; any hi, lo.
__device__ __forceinline__ u64 reduce128(u64 lo, u64 hi, u64 q, u64 ratio1, u64 r64_quo)
{
    const u64 r64 = 0 - ratio1 * q; // 2^64 mod q
    const u64 t1 = r64 * hi - mulhi64(hi, r64_quo) * q; // [0, 2q)
    const u64 t0 = lo - mulhi64(lo, ratio1) * q;        // [0, 2q)
    return csub(csub(t0 + t1, q << 1), q);
}

// multiply_uint_mod(a, b, modulus): full product then Barrett-128
__device__ __forceinline__ u64 mul_mod(u64 a, u64 b, u64 q, u64 r0, u64 r1)
{
    return barrett128(a * b, mulhi64(a, b), q, r0, r1);
}

__device__ __forceinline__ u64 add_mod(u64 a, u64 b, u64 q)
{
    return csub(a + b, q);
}

__device__ __forceinline__ u64 sub_mod(u64 a, u64 b, u64 q)
{
    u64 t = a - b;
    return a < b ? t + q : t;
}

__device__ __forceinline__ u64 neg_mod(u64 a, u64 q)
{
    return a ? q - a : 0;
}

// 128-bit accumulate: (lo,hi) += a*b
__device__ __forceinline__ void mac128(u64 &lo, u64 &hi, u64 a, u64 b)
{
    u64 pl = a * b, ph = mulhi64(a, b);
    lo += pl;
    hi += ph + (lo < pl);
}

// ---- "split Shoup" multiply (internal fast path; representatives differ from SEAL's lazy ones) ------------
// For a constant w < q:  W0 = w, W1 = w * 2^32 mod q, Q0 = floor(W0 * 2^32 / q), Q1 = floor(W1 * 2^32 / q).
// With x = x1 * 2^32 + x0:  x * w = x0 * W0 + x1 * W1 (mod q) and each 32-bit Shoup step loses less than 2q, so
//   r = x0*W0 + x1*W1 - (mulhi32(x0,Q0) + mulhi32(x1,Q1)) * q   lies in [0, 4q)   for ANY 64-bit x.
// 10 32-bit multiplies like the 64-bit Shoup product, but no 64x64 high product: two v_mad_u64_u32 chains with
// natural register pairs (no carry handling, no pair building), which is what the VALU-bound NTT needs
// (tools/ubench_bfly.hip: 1.37 -> 1.59 T butterflies/s on MI355X).
struct __attribute__((aligned(8))) MulOp3
{
    u64 w0;
    u64 w1;
    u32 q0;
    u32 q1;
};

// nq = 2^64 - q.  Two independent v_mad_u64_u32 chains: the low-word products (a natural register pair) and the
// high-word cross terms, whose low word is added into the upper word by one 32-bit add (no pair building).
__device__ __forceinline__ u64 mul_split(u64 x, u64 W0, u64 W1, u32 Q0, u32 Q1, u64 nq)
{
    u64 L, C;
    u32 h0, h1;
    asm("v_mul_hi_u32 %2, %4, %10\n\t"
        "v_mul_hi_u32 %3, %5, %11\n\t"
        "v_mad_u64_u32 %1, vcc, %4, %7, 0\n\t"
        "v_mad_u64_u32 %0, vcc, %4, %6, 0\n\t"
        "v_mad_u64_u32 %1, vcc, %5, %9, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %8, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %2, %13, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %2, %12, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %3, %13, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %12, %0"
        : "=&v"(L), "=&v"(C), "=&v"(h0), "=&v"(h1)
        : "v"((u32)x), "v"((u32)(x >> 32)), "v"((u32)W0), "v"((u32)(W0 >> 32)), "v"((u32)W1), "v"((u32)(W1 >> 32)),
          "v"(Q0), "v"(Q1), "v"((u32)nq), "v"((u32)(nq >> 32))
        : "vcc");
    u32 hi = (u32)(L >> 32);
    asm("v_add_u32 %0, %0, %1" : "+v"(hi) : "v"((u32)C));
    return ((u64)hi << 32) | (u32)L;
}

// a + x*w in the same lazy sense (result < a + 4q): two independent chains -- the low-word products accumulate on
// top of `a`, the high-word cross terms on their own -- joined by ONE 32-bit add into the upper word.
__device__ __forceinline__ u64 mul_split_add(u64 x, u64 W0, u64 W1, u32 Q0, u32 Q1, u64 nq, u64 a)
{
    u64 L, C;
    u32 h0, h1;
    asm("v_mul_hi_u32 %2, %4, %10\n\t"
        "v_mul_hi_u32 %3, %5, %11\n\t"
        "v_mad_u64_u32 %1, vcc, %4, %7, 0\n\t"
        "v_mad_u64_u32 %0, vcc, %4, %6, %14\n\t"
        "v_mad_u64_u32 %1, vcc, %5, %9, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %8, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %2, %13, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %2, %12, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %3, %13, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %12, %0"
        : "=&v"(L), "=&v"(C), "=&v"(h0), "=&v"(h1)
        : "v"((u32)x), "v"((u32)(x >> 32)), "v"((u32)W0), "v"((u32)(W0 >> 32)), "v"((u32)W1), "v"((u32)(W1 >> 32)),
          "v"(Q0), "v"(Q1), "v"((u32)nq), "v"((u32)(nq >> 32)), "v"(a)
        : "vcc");
    u32 hi = (u32)(L >> 32);
    asm("v_add_u32 %0, %0, %1" : "+v"(hi) : "v"((u32)C));
    return ((u64)hi << 32) | (u32)L;
}

// 64-bit add as one v_lshl_add_u64 (the compiler prefers a two-instruction carry chain for x + constant)
__device__ __forceinline__ u64 add64(u64 a, u64 b)
{
    u64 r;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// 2a + b as one v_lshl_add_u64
__device__ __forceinline__ u64 shl1_add64(u64 a, u64 b)
{
    u64 r;
    asm("v_lshl_add_u64 %0, %1, 1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ u64 mul_split(u64 x, const MulOp3 &w, u64 nq)
{
    return mul_split(x, w.w0, w.w1, w.q0, w.q1, nq);
}
__device__ __forceinline__ u64 mul_split_add(u64 x, const MulOp3 &w, u64 nq, u64 a)
{
    return mul_split_add(x, w.w0, w.w1, w.q0, w.q1, nq, a);
}

// 128-bit lazy accumulation without a 64x64 high product.  a b = a0 b0 + (a0 b1 + a1 b0) 2^32 + a1 b1 2^64 with
//   lo  += a0 b0            one v_mad_u64_u32, its carry-out counted in `cc` (weight 2^64),
//   mid += a0 b1 + a1 b0    summed over at most TWO terms, then folded into lo/hi by rm_fold(),
//   hi  += a1 b1            plain 64-bit accumulation.
// Operand bounds the callers keep: (a < 2^62, b < 2^60) or (a, b < 2^61); then a term's middle sum is below 2^63,
// so two terms fit 64 bits, and a1 b1 < 2^58, so up to 64 terms need no carry.
struct Acc
{
    u64 lo, hi;
    u32 cc;
};
__device__ __forceinline__ void rm_mac(Acc &acc, u64 &mid, u64 a, u64 b)
{
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc.lo), "=s"(carry) : "v"(a0), "v"(b0));
    asm("v_addc_co_u32 %0, vcc, 0, %0, %1" : "+v"(acc.cc) : "s"(carry) : "vcc");
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc.hi) : "v"(a1), "v"(b1) : "vcc");
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %4, %0"
        : "+v"(mid)
        : "v"(a0), "v"(b1), "v"(a1), "v"(b0)
        : "vcc");
}
__device__ __forceinline__ void rm_fold(Acc &acc, u64 mid)
{
    const u64 t = acc.lo + (mid << 32);
    acc.cc += t < acc.lo;
    acc.lo = t;
    acc.hi += mid >> 32;
}

// The same accumulation for TWO digits of one coefficient as one statement: the carry of the low product is consumed
// three instructions after it is produced (gfx950 wants two wait states there) and the middle sum starts from the
// literal 0, so no zeroing move.
__device__ __forceinline__ void rm_mac2(Acc &acc, u64 &mid, u64 a0, u64 b0, u64 a1, u64 b1)
{
    u64 sc;
    asm("v_mad_u64_u32 %0, %4, %5, %9, %0\n\t"
        "v_mad_u64_u32 %3, vcc, %5, %11, 0\n\t"
        "v_mad_u64_u32 %1, vcc, %7, %11, %1\n\t"
        "v_mad_u64_u32 %3, vcc, %7, %9, %3\n\t"
        "v_addc_co_u32 %2, vcc, 0, %2, %4\n\t"
        "v_mad_u64_u32 %0, %4, %6, %10, %0\n\t"
        "v_mad_u64_u32 %3, vcc, %6, %12, %3\n\t"
        "v_mad_u64_u32 %1, vcc, %8, %12, %1\n\t"
        "v_mad_u64_u32 %3, vcc, %8, %10, %3\n\t"
        "v_addc_co_u32 %2, vcc, 0, %2, %4"
        : "+v"(acc.lo), "+v"(acc.hi), "+v"(acc.cc), "=&v"(mid), "=&s"(sc)
        : "v"((u32)a0), "v"((u32)a1), "v"((u32)(a0 >> 32)), "v"((u32)(a1 >> 32)), "v"((u32)b0), "v"((u32)b1),
          "v"((u32)(b0 >> 32)), "v"((u32)(b1 >> 32))
        : "vcc");
}
// rm_fold for four accumulators: the low word of mid goes into the high word of lo (carry counted in cc), its high word
// into hi by one v_mad_u64_u32 (x 1); the compiler's form builds mid << 32 and mid >> 32 as register pairs with moves.
__device__ __forceinline__ void rm_fold4(Acc &e0, Acc &e1, Acc &e2, Acc &e3, u64 m0, u64 m1, u64 m2, u64 m3)
{
    u32 h0 = (u32)(e0.lo >> 32), h1 = (u32)(e1.lo >> 32), h2 = (u32)(e2.lo >> 32), h3 = (u32)(e3.lo >> 32);
    u64 s0, s1, s2, s3;
    asm("v_add_co_u32 %0, %12, %0, %16\n\t"
        "v_add_co_u32 %1, %13, %1, %17\n\t"
        "v_add_co_u32 %2, %14, %2, %18\n\t"
        "v_add_co_u32 %3, %15, %3, %19\n\t"
        "v_mad_u64_u32 %4, vcc, %20, 1, %4\n\t"
        "v_mad_u64_u32 %5, vcc, %21, 1, %5\n\t"
        "v_mad_u64_u32 %6, vcc, %22, 1, %6\n\t"
        "v_mad_u64_u32 %7, vcc, %23, 1, %7\n\t"
        "v_addc_co_u32 %8, vcc, 0, %8, %12\n\t"
        "v_addc_co_u32 %9, vcc, 0, %9, %13\n\t"
        "v_addc_co_u32 %10, vcc, 0, %10, %14\n\t"
        "v_addc_co_u32 %11, vcc, 0, %11, %15"
        : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(e0.hi), "+v"(e1.hi), "+v"(e2.hi), "+v"(e3.hi), "+v"(e0.cc),
          "+v"(e1.cc), "+v"(e2.cc), "+v"(e3.cc), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3)
        : "v"((u32)m0), "v"((u32)m1), "v"((u32)m2), "v"((u32)m3), "v"((u32)(m0 >> 32)), "v"((u32)(m1 >> 32)),
          "v"((u32)(m2 >> 32)), "v"((u32)(m3 >> 32))
        : "vcc");
    e0.lo = ((u64)h0 << 32) | (u32)e0.lo;
    e1.lo = ((u64)h1 << 32) | (u32)e1.lo;
    e2.lo = ((u64)h2 << 32) | (u32)e2.lo;
    e3.lo = ((u64)h3 << 32) | (u32)e3.lo;
}

// Per-modulus constants of the NTT kernels (both arithmetic flavours)
struct __attribute__((aligned(16))) NttMod
{
    u64 q;
    u64 ratio1;     // high word of floor(2^128/q): Barrett-64 of the fused modulus raise
    MulOp inv_n;    // n^-1 mod q                              (util/ntt.cpp:289-295)
    MulOp inv_n_w;  // inv_root_powers[n-1] * n^-1 mod q      (util/ntt.h:50-56)
    MulOp3 inv_n3;
    MulOp3 inv_n_w3;
    // reduce_below_2q(): red_r = floor(2^(31+b) / q), red_shift = b - 33 for a prime of b >= 36 bits, else red_r = 0
    u32 red_r, red_shift;
    u64 pad;
};

// x < 16q < 2^64, q a prime of b >= 36 bits: x - e q in [0, 2q) with e = floor(x / q) or one less, estimated from the
// high word: mulhi32(x_hi, floor(2^(31+b)/q)) = (x/q) 2^(b-33) - (less than 3), so e = that >> (b - 33) never exceeds
// floor(x/q) and misses it by at most one.  5 instructions (v_mul_hi_u32, shift, v_mad_u64_u32, v_mul_lo_u32, add)
// where the two conditional subtractions of 8q and 4q take 8 and leave the value below 4q only.
__device__ __forceinline__ u64 reduce_below_2q(u64 x, u32 red_r, u32 red_shift, u64 nq)
{
    const u32 e = __umulhi((u32)(x >> 32), red_r) >> red_shift;
    u64 r = x + (u64)e * (u32)nq;
    r += (u64)(e * (u32)(nq >> 32)) << 32;
    return r;
}

// Limb position -> modulus index map, passed to kernels by value (CAT_MAX_MODULI entries).
struct LimbMap
{
    unsigned char id[64];
};
