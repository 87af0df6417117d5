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
    u64 pad;
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

// Limb position -> modulus index map, passed to kernels by value (CAT_MAX_MODULI entries).
struct LimbMap
{
    unsigned char id[64];
};
