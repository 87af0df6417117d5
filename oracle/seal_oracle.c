/*
 * seal_oracle.c -- CPU restatement of the reference hot path (see seal_oracle.h).
 * TEST INFRASTRUCTURE: never linked into the product library.
 * Reference paths are relative to /root/reference/native/src/seal/.
 */
#include "seal_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------------------------------
 * Word arithmetic
 * ---------------------------------------------------------------------------------------------- */

static inline uint64_t mul_hi64(uint64_t a, uint64_t b) /* util/uintarith.cuh:805-910 multiply_uint64_hw64 */
{
    return (uint64_t)(((u128)a * b) >> 64);
}

static int sig_bits(uint64_t v)
{
    int b = 0;
    while (v)
    {
        b++;
        v >>= 1;
    }
    return b;
}

/* modulus.cpp:66-105 Modulus::set_value: const_ratio = floor(2^128 / value), remainder */
void so_modulus_init(so_modulus *m, uint64_t value)
{
    m->value = value;
    m->bit_count = sig_bits(value);
    if (value == 0)
    {
        m->ratio[0] = m->ratio[1] = m->ratio[2] = 0;
        return;
    }
    u128 all = ~(u128)0; /* 2^128 - 1 */
    u128 q = all / value;
    u128 r = all % value;
    r += 1;
    if (r == value)
    {
        q += 1;
        r = 0;
    }
    m->ratio[0] = (uint64_t)q;
    m->ratio[1] = (uint64_t)(q >> 64);
    m->ratio[2] = (uint64_t)r;
}

/* util/uintarithsmallmod.cuh:312-333 */
uint64_t so_barrett_reduce_64(uint64_t x, const so_modulus *m)
{
    uint64_t t = mul_hi64(x, m->ratio[1]);
    uint64_t r = x - t * m->value;
    return r >= m->value ? r - m->value : r;
}

/* util/uintarithsmallmod.cuh:188-226 */
uint64_t so_barrett_reduce_128(const uint64_t in[2], const so_modulus *m)
{
    uint64_t tmp1, tmp3, carry;
    u128 tmp2;
    carry = mul_hi64(in[0], m->ratio[0]);
    tmp2 = (u128)in[0] * m->ratio[1];
    {
        uint64_t lo = (uint64_t)tmp2, hi = (uint64_t)(tmp2 >> 64);
        tmp1 = lo + carry;
        tmp3 = hi + (tmp1 < lo);
    }
    tmp2 = (u128)in[1] * m->ratio[0];
    {
        uint64_t lo = (uint64_t)tmp2, hi = (uint64_t)(tmp2 >> 64);
        uint64_t s = tmp1 + lo;
        carry = hi + (s < tmp1);
    }
    tmp1 = in[1] * m->ratio[1] + tmp3 + carry;
    tmp3 = in[0] - tmp1 * m->value;
    return tmp3 >= m->value ? tmp3 - m->value : tmp3;
}

/* util/uintarithsmallmod.cpp:175-188 */
uint64_t so_multiply_uint_mod(uint64_t a, uint64_t b, const so_modulus *m)
{
    u128 z = (u128)a * b;
    uint64_t w[2] = { (uint64_t)z, (uint64_t)(z >> 64) };
    return so_barrett_reduce_128(w, m);
}

/* util/uintarithsmallmod.cuh:368-392 */
void so_mulmod_op_set(so_mulmod_op *op, uint64_t operand, const so_modulus *m)
{
    op->operand = operand;
    op->quotient = (uint64_t)((((u128)operand) << 64) / m->value);
}

/* util/uintarithsmallmod.cuh:416-429 */
uint64_t so_multiply_uint_mod_lazy(uint64_t x, so_mulmod_op y, const so_modulus *m)
{
    uint64_t t = mul_hi64(x, y.quotient);
    return y.operand * x - t * m->value;
}

/* util/uintarithsmallmod.cpp:190-204 */
uint64_t so_multiply_uint_mod_op(uint64_t x, so_mulmod_op y, const so_modulus *m)
{
    uint64_t r = so_multiply_uint_mod_lazy(x, y, m);
    return r >= m->value ? r - m->value : r;
}

/* util/uintarithsmallmod.cuh:115-131 */
uint64_t so_add_uint_mod(uint64_t a, uint64_t b, const so_modulus *m)
{
    a += b;
    return a >= m->value ? a - m->value : a;
}

/* util/uintarithsmallmod.cuh:146-167 */
uint64_t so_sub_uint_mod(uint64_t a, uint64_t b, const so_modulus *m)
{
    uint64_t t = a - b;
    return a < b ? t + m->value : t;
}

/* util/uintarithsmallmod.cuh:64-78 */
uint64_t so_negate_uint_mod(uint64_t a, const so_modulus *m)
{
    return a ? m->value - a : 0;
}

/* util/uintarithsmallmod.cuh:564-569 */
static uint64_t multiply_add_uint_mod_op(uint64_t a, so_mulmod_op b, uint64_t c, const so_modulus *m)
{
    return so_add_uint_mod(so_multiply_uint_mod_op(a, b, m), so_barrett_reduce_64(c, m), m);
}

/* util/uintarithsmallmod.cpp exponentiate_uint_mod (square and multiply) */
uint64_t so_exponentiate_uint_mod(uint64_t a, uint64_t e, const so_modulus *m)
{
    uint64_t r = 1, p = a;
    if (e == 0)
        return 1;
    while (e)
    {
        if (e & 1)
            r = so_multiply_uint_mod(r, p, m);
        e >>= 1;
        if (e)
            p = so_multiply_uint_mod(p, p, m);
    }
    return r;
}

/* util/numth.h try_invert_uint_mod via extended Euclid; returns 0 when not invertible */
int so_try_invert_uint_mod(uint64_t a, uint64_t modulus, uint64_t *result)
{
    if (a == 0)
        return 0;
    __int128 r0 = modulus, r1 = a % modulus, t0 = 0, t1 = 1;
    if (modulus == 1)
        return 0;
    while (r1 != 0)
    {
        __int128 q = r0 / r1;
        __int128 r2 = r0 - q * r1;
        __int128 t2 = t0 - q * t1;
        r0 = r1;
        r1 = r2;
        t0 = t1;
        t1 = t2;
    }
    if (r0 != 1)
        return 0;
    if (t0 < 0)
        t0 += modulus;
    *result = (uint64_t)t0;
    return 1;
}

/* util/uintarithsmallmod.cpp:110-173: sum of products in 128 bits (<= 16 terms per block), Barrett-128 */
uint64_t so_dot_product_mod(const uint64_t *a, const uint64_t *b, size_t count, const so_modulus *m)
{
    if (count == 0)
        return 0;
    u128 acc = 0;
    if (count > 16)
    {
        acc = so_dot_product_mod(a + 16, b + 16, count - 16, m);
        count = 16;
    }
    for (size_t i = 0; i < count; i++)
        acc += (u128)a[i] * b[i];
    uint64_t w[2] = { (uint64_t)acc, (uint64_t)(acc >> 64) };
    return so_barrett_reduce_128(w, m);
}

/* util/uintarithsmallmod.cuh:472-504: multi-word value mod q */
uint64_t so_modulo_uint(const uint64_t *value, size_t count, const so_modulus *m)
{
    if (count == 1)
        return value[0] < m->value ? value[0] : so_barrett_reduce_64(value[0], m);
    uint64_t temp[2] = { 0, value[count - 1] };
    for (size_t k = count - 1; k--;)
    {
        temp[0] = value[k];
        temp[1] = so_barrett_reduce_128(temp, m);
    }
    return temp[1];
}

/* ------------------------------------------------------------------------------------------------
 * Number theory
 * ---------------------------------------------------------------------------------------------- */

static uint64_t mulmod_plain(uint64_t a, uint64_t b, uint64_t q)
{
    return (uint64_t)(((u128)a * b) % q);
}

static uint64_t powmod_plain(uint64_t a, uint64_t e, uint64_t q)
{
    uint64_t r = 1;
    a %= q;
    while (e)
    {
        if (e & 1)
            r = mulmod_plain(r, a, q);
        a = mulmod_plain(a, a, q);
        e >>= 1;
    }
    return r;
}

/* util/numth.cpp:180-276 is a randomized Miller-Rabin; for 64-bit inputs the fixed base set below is
 * a proof of primality, so the accepted set is identical. */
int so_is_prime(uint64_t n)
{
    static const uint64_t bases[] = { 2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37 };
    if (n < 2)
        return 0;
    for (size_t i = 0; i < sizeof(bases) / sizeof(bases[0]); i++)
    {
        if (n == bases[i])
            return 1;
        if (n % bases[i] == 0)
            return 0;
    }
    uint64_t d = n - 1;
    int r = 0;
    while (!(d & 1))
    {
        d >>= 1;
        r++;
    }
    for (size_t i = 0; i < sizeof(bases) / sizeof(bases[0]); i++)
    {
        uint64_t x = powmod_plain(bases[i], d, n);
        if (x == 1 || x == n - 1)
            continue;
        int comp = 1;
        for (int j = 1; j < r; j++)
        {
            x = mulmod_plain(x, x, n);
            if (x == n - 1)
            {
                comp = 0;
                break;
            }
        }
        if (comp)
            return 0;
    }
    return 1;
}

/* util/numth.cpp:278-311 */
int so_get_primes(uint64_t factor, int bit_size, size_t count, uint64_t *out)
{
    uint64_t value = (((uint64_t)1 << bit_size) - 1) / factor * factor + 1;
    uint64_t lower = (uint64_t)1 << (bit_size - 1);
    size_t found = 0;
    while (found < count && value > lower)
    {
        if (so_is_prime(value))
            out[found++] = value;
        value -= factor;
    }
    return found == count ? 0 : -1;
}

/* modulus.cpp:150-190 CoeffModulus::Create: per bit size the primes are handed out smallest-first
 * (pop_back of the descending get_primes list). */
int so_coeff_modulus_create(size_t n, const int *bit_sizes, size_t count, uint64_t *out)
{
    int cnt[64] = { 0 };
    uint64_t *tab[64] = { 0 };
    int rc = 0;
    for (size_t i = 0; i < count; i++)
    {
        if (bit_sizes[i] < 2 || bit_sizes[i] > 60)
            return -1;
        cnt[bit_sizes[i]]++;
    }
    for (int b = 2; b <= 60; b++)
    {
        if (!cnt[b])
            continue;
        tab[b] = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)cnt[b]);
        if (so_get_primes(2 * (uint64_t)n, b, (size_t)cnt[b], tab[b]) != 0)
            rc = -1;
    }
    if (rc == 0)
    {
        for (size_t i = 0; i < count; i++)
        {
            int b = bit_sizes[i];
            out[i] = tab[b][--cnt[b]];
        }
    }
    for (int b = 2; b <= 60; b++)
        free(tab[b]);
    return rc;
}

/* util/globals.cpp:24-70 (128-bit security defaults): data table */
int so_bfv_default(size_t n, uint64_t *out, size_t *count)
{
    static const uint64_t p4096[] = { 0xffffee001ULL, 0xffffc4001ULL, 0x1ffffe0001ULL };
    static const uint64_t p8192[] = { 0x7fffffd8001ULL, 0x7fffffc8001ULL, 0xfffffffc001ULL, 0xffffff6c001ULL,
                                      0xfffffebc001ULL };
    static const uint64_t p16384[] = { 0xfffffffd8001ULL,  0xfffffffa0001ULL,  0xfffffff00001ULL,
                                       0x1fffffff68001ULL, 0x1fffffff50001ULL, 0x1ffffffee8001ULL,
                                       0x1ffffffea0001ULL, 0x1ffffffe88001ULL, 0x1ffffffe48001ULL };
    static const uint64_t p32768[] = { 0x7fffffffe90001ULL, 0x7fffffffbf0001ULL, 0x7fffffffbd0001ULL,
                                       0x7fffffffba0001ULL, 0x7fffffffaa0001ULL, 0x7fffffffa50001ULL,
                                       0x7fffffff9f0001ULL, 0x7fffffff7e0001ULL, 0x7fffffff770001ULL,
                                       0x7fffffff380001ULL, 0x7fffffff330001ULL, 0x7fffffff2d0001ULL,
                                       0x7fffffff170001ULL, 0x7fffffff150001ULL, 0x7ffffffef00001ULL,
                                       0xfffffffff70001ULL };
    const uint64_t *src;
    size_t c;
    switch (n)
    {
    case 4096:
        src = p4096;
        c = 3;
        break;
    case 8192:
        src = p8192;
        c = 5;
        break;
    case 16384:
        src = p16384;
        c = 9;
        break;
    case 32768:
        src = p32768;
        c = 16;
        break;
    default:
        return -1;
    }
    memcpy(out, src, c * sizeof(uint64_t));
    *count = c;
    return 0;
}

/* util/numth.cpp:313-412.  try_primitive_root starts from a random element; the minimal root found
 * by try_minimal_primitive_root does not depend on the start, so a deterministic search is used. */
int so_try_minimal_primitive_root(uint64_t degree, const so_modulus *m, uint64_t *out)
{
    uint64_t q = m->value;
    if ((q - 1) % degree != 0)
        return 0;
    uint64_t quotient_group = (q - 1) / degree;
    uint64_t root = 0;
    for (uint64_t g = 2; g < q && g < 100000; g++)
    {
        uint64_t cand = so_exponentiate_uint_mod(g, quotient_group, m);
        if (cand != 0 && so_exponentiate_uint_mod(cand, degree >> 1, m) == q - 1)
        {
            root = cand;
            break;
        }
    }
    if (!root)
        return 0;
    uint64_t generator_sq = so_multiply_uint_mod(root, root, m);
    uint64_t current = root;
    for (uint64_t i = 0; i < degree; i += 2)
    {
        if (current < root)
            root = current;
        current = so_multiply_uint_mod(current, generator_sq, m);
    }
    *out = root;
    return 1;
}

/* util/common.h reverse_bits(operand, bit_count) */
uint32_t so_reverse_bits(uint32_t x, int bits)
{
    uint32_t r = 0;
    for (int i = 0; i < bits; i++)
        if (x & ((uint32_t)1 << i))
            r |= (uint32_t)1 << (bits - 1 - i);
    return r;
}

/* ------------------------------------------------------------------------------------------------
 * NTT
 * ---------------------------------------------------------------------------------------------- */

/* util/ntt.cpp:240-299 NTTTables::initialize */
so_ntt_tables *so_ntt_tables_create(int log_n, uint64_t modulus)
{
    so_ntt_tables *t = (so_ntt_tables *)calloc(1, sizeof(*t));
    t->log_n = log_n;
    t->n = (size_t)1 << log_n;
    so_modulus_init(&t->mod, modulus);
    if (!so_try_minimal_primitive_root(2 * (uint64_t)t->n, &t->mod, &t->root) ||
        !so_try_invert_uint_mod(t->root, modulus, &t->inv_root))
    {
        free(t);
        return NULL;
    }
    t->root_powers = (so_mulmod_op *)malloc(sizeof(so_mulmod_op) * t->n);
    t->inv_root_powers = (so_mulmod_op *)malloc(sizeof(so_mulmod_op) * t->n);
    so_mulmod_op root;
    so_mulmod_op_set(&root, t->root, &t->mod);
    uint64_t power = t->root;
    for (size_t i = 1; i < t->n; i++)
    {
        so_mulmod_op_set(&t->root_powers[so_reverse_bits((uint32_t)i, log_n)], power, &t->mod);
        power = so_multiply_uint_mod_op(power, root, &t->mod);
    }
    so_mulmod_op_set(&t->root_powers[0], 1, &t->mod);

    so_mulmod_op_set(&root, t->inv_root, &t->mod);
    power = t->inv_root;
    for (size_t i = 1; i < t->n; i++)
    {
        so_mulmod_op_set(&t->inv_root_powers[so_reverse_bits((uint32_t)(i - 1), log_n) + 1], power, &t->mod);
        power = so_multiply_uint_mod_op(power, root, &t->mod);
    }
    so_mulmod_op_set(&t->inv_root_powers[0], 1, &t->mod);

    uint64_t inv_n;
    if (!so_try_invert_uint_mod((uint64_t)t->n, modulus, &inv_n))
    {
        so_ntt_tables_destroy(t);
        return NULL;
    }
    so_mulmod_op_set(&t->inv_degree, inv_n, &t->mod);
    return t;
}

void so_ntt_tables_destroy(so_ntt_tables *t)
{
    if (!t)
        return;
    free(t->root_powers);
    free(t->inv_root_powers);
    free(t);
}

uint64_t so_ntt_root_power(const so_ntt_tables *t, size_t i)
{
    return t->root_powers[i].operand;
}

uint64_t so_ntt_inv_root_power(const so_ntt_tables *t, size_t i)
{
    return t->inv_root_powers[i].operand;
}

/* util/dwthandler.h:94-192 transform_to_rev with util/ntt.h:20-67 lazy arithmetic
 * (guard: x >= 2q ? x-2q : x; add: a+b; sub: a+2q-b; mul_root: Shoup lazy), then
 * util/ntt.cpp:407-436 final correction when !lazy. */
void so_ntt_forward(uint64_t *values, const so_ntt_tables *t, int lazy)
{
    const so_modulus *m = &t->mod;
    const uint64_t q = m->value, two_q = q << 1;
    const size_t n = t->n;
    const so_mulmod_op *roots = t->root_powers;
    size_t gap = n >> 1, mm = 1;
    for (; mm < n; mm <<= 1)
    {
        size_t offset = 0;
        for (size_t i = 0; i < mm; i++)
        {
            so_mulmod_op r = *++roots;
            uint64_t *x = values + offset, *y = x + gap;
            for (size_t j = 0; j < gap; j++)
            {
                uint64_t u = *x >= two_q ? *x - two_q : *x;
                uint64_t v = so_multiply_uint_mod_lazy(*y, r, m);
                *x++ = u + v;
                *y++ = u + two_q - v;
            }
            offset += gap << 1;
        }
        gap >>= 1;
    }
    if (!lazy)
    {
        for (size_t i = 0; i < n; i++)
        {
            uint64_t v = values[i];
            if (v >= two_q)
                v -= two_q;
            if (v >= q)
                v -= q;
            values[i] = v;
        }
    }
}

/* util/dwthandler.h:203-357 transform_from_rev with scalar = n^-1 merged into the last stage
 * (util/ntt.cpp:438-450), then util/ntt.cpp:452-474 final correction when !lazy. */
void so_ntt_inverse(uint64_t *values, const so_ntt_tables *t, int lazy)
{
    const so_modulus *m = &t->mod;
    const uint64_t q = m->value, two_q = q << 1;
    const size_t n = t->n;
    const so_mulmod_op *roots = t->inv_root_powers;
    size_t gap = 1, mm = n >> 1;
    for (; mm > 1; mm >>= 1)
    {
        size_t offset = 0;
        for (size_t i = 0; i < mm; i++)
        {
            so_mulmod_op r = *++roots;
            uint64_t *x = values + offset, *y = x + gap;
            for (size_t j = 0; j < gap; j++)
            {
                uint64_t u = *x, v = *y;
                uint64_t s = u + v;
                *x++ = s >= two_q ? s - two_q : s;
                *y++ = so_multiply_uint_mod_lazy(u + two_q - v, r, m);
            }
            offset += gap << 1;
        }
        gap <<= 1;
    }
    {
        so_mulmod_op r = *++roots;
        so_mulmod_op scaled_r;
        /* util/ntt.h:50-56 mul_root_scalar */
        so_mulmod_op_set(&scaled_r, so_multiply_uint_mod_op(r.operand, t->inv_degree, m), m);
        uint64_t *x = values, *y = x + gap;
        for (size_t j = 0; j < gap; j++)
        {
            uint64_t u = *x >= two_q ? *x - two_q : *x;
            uint64_t v = *y;
            uint64_t s = u + v;
            s = s >= two_q ? s - two_q : s;
            *x++ = so_multiply_uint_mod_lazy(s, t->inv_degree, m);
            *y++ = so_multiply_uint_mod_lazy(u + two_q - v, scaled_r, m);
        }
    }
    if (!lazy)
    {
        for (size_t i = 0; i < n; i++)
            if (values[i] >= q)
                values[i] -= q;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Coefficient-wise ops (one limb), util/polyarithsmallmod.cpp
 * ---------------------------------------------------------------------------------------------- */

void so_modulo_poly_coeffs(const uint64_t *a, size_t n, const so_modulus *m, uint64_t *r) /* :18-41 */
{
    for (size_t i = 0; i < n; i++)
        r[i] = so_barrett_reduce_64(a[i], m);
}

void so_add_poly_coeffmod(const uint64_t *a, const uint64_t *b, size_t n, const so_modulus *m, uint64_t *r) /* :43-86 */
{
    for (size_t i = 0; i < n; i++)
        r[i] = so_add_uint_mod(a[i], b[i], m);
}

void so_sub_poly_coeffmod(const uint64_t *a, const uint64_t *b, size_t n, const so_modulus *m, uint64_t *r) /* :88-131 */
{
    for (size_t i = 0; i < n; i++)
        r[i] = so_sub_uint_mod(a[i], b[i], m);
}

void so_negate_poly_coeffmod(const uint64_t *a, size_t n, const so_modulus *m, uint64_t *r) /* .cuh:84-117 */
{
    for (size_t i = 0; i < n; i++)
        r[i] = so_negate_uint_mod(a[i], m);
}

void so_add_poly_scalar_coeffmod(const uint64_t *a, size_t n, uint64_t s, const so_modulus *m, uint64_t *r) /* :133-161 */
{
    for (size_t i = 0; i < n; i++)
        r[i] = so_add_uint_mod(a[i], s, m);
}

void so_sub_poly_scalar_coeffmod(const uint64_t *a, size_t n, uint64_t s, const so_modulus *m, uint64_t *r) /* :163-191 */
{
    for (size_t i = 0; i < n; i++)
        r[i] = so_sub_uint_mod(a[i], s, m);
}

static void mul_poly_op(const uint64_t *a, size_t n, so_mulmod_op s, const so_modulus *m, uint64_t *r) /* :193-225 */
{
    for (size_t i = 0; i < n; i++)
        r[i] = so_multiply_uint_mod_op(a[i], s, m);
}

/* util/polyarithsmallmod.cuh:477-486: scalar first reduced mod q, then Shoup multiply */
void so_multiply_poly_scalar_coeffmod(const uint64_t *a, size_t n, uint64_t s, const so_modulus *m, uint64_t *r)
{
    so_mulmod_op op;
    so_mulmod_op_set(&op, so_barrett_reduce_64(s, m), m);
    mul_poly_op(a, n, op, m, r);
}

void so_dyadic_product_coeffmod(const uint64_t *a, const uint64_t *b, size_t n, const so_modulus *m, uint64_t *r) /* :227-285 */
{
    for (size_t i = 0; i < n; i++)
        r[i] = so_multiply_uint_mod(a[i], b[i], m);
}

/* ------------------------------------------------------------------------------------------------
 * Galois, util/galois.cpp
 * ---------------------------------------------------------------------------------------------- */

uint32_t so_galois_elt_from_step(int log_n, int step) /* :53-95 */
{
    uint32_t n = (uint32_t)1 << log_n;
    uint64_t m = 2 * (uint64_t)n;
    if (step == 0)
        return (uint32_t)(m - 1);
    int sign = step < 0;
    uint32_t pos = (uint32_t)(step < 0 ? -step : step);
    if (pos >= (n >> 1))
        return 0; /* reference throws invalid_argument("step count too large") */
    int s = sign ? (int)(n >> 1) - (int)pos : (int)pos;
    uint64_t elt = 1;
    while (s--)
    {
        elt *= 3;
        elt &= m - 1;
    }
    return (uint32_t)elt;
}

void so_galois_table_ntt(int log_n, uint32_t elt, uint32_t *table) /* :18-51 */
{
    size_t n = (size_t)1 << log_n;
    for (size_t i = n; i < 2 * n; i++)
    {
        uint32_t reversed = so_reverse_bits((uint32_t)i, log_n + 1);
        uint64_t index_raw = ((uint64_t)elt * reversed) >> 1;
        index_raw &= (uint64_t)(n - 1);
        table[i - n] = so_reverse_bits((uint32_t)index_raw, log_n);
    }
}

void so_apply_galois_ntt(const uint64_t *in, int log_n, uint32_t elt, uint64_t *out) /* :192-218 */
{
    size_t n = (size_t)1 << log_n;
    uint32_t *table = (uint32_t *)malloc(sizeof(uint32_t) * n);
    so_galois_table_ntt(log_n, elt, table);
    for (size_t i = 0; i < n; i++)
        out[i] = in[table[i]];
    free(table);
}

void so_apply_galois(const uint64_t *in, int log_n, uint32_t elt, const so_modulus *m, uint64_t *out) /* :148-190 */
{
    uint64_t nm1 = ((uint64_t)1 << log_n) - 1;
    uint64_t index_raw = 0;
    for (uint64_t i = 0; i <= nm1; i++, index_raw += elt)
    {
        uint64_t index = index_raw & nm1;
        uint64_t v = in[i];
        if ((index_raw >> log_n) & 1)
            v = v ? m->value - v : 0;
        out[index] = v;
    }
}

/* ------------------------------------------------------------------------------------------------
 * RNS bases, base converter, RNSTool (util/rns.cpp)
 * ---------------------------------------------------------------------------------------------- */

struct so_rnsbase
{
    size_t size;
    so_modulus *base;
    uint64_t *base_prod;                /* size words */
    uint64_t *punctured_prod;           /* size x size words */
    so_mulmod_op *inv_punctured_mod_base; /* size */
};

/* multi-word helpers (little endian) */
static void mp_mul_word(const uint64_t *a, size_t n, uint64_t w, uint64_t *r /* n words, truncating */)
{
    uint64_t carry = 0;
    for (size_t i = 0; i < n; i++)
    {
        u128 t = (u128)a[i] * w + carry;
        r[i] = (uint64_t)t;
        carry = (uint64_t)(t >> 64);
    }
}

/* util/rns.cpp:212-257 RNSBase::initialize */
static so_rnsbase *rnsbase_create(const uint64_t *vals, size_t size)
{
    so_rnsbase *b = (so_rnsbase *)calloc(1, sizeof(*b));
    b->size = size;
    b->base = (so_modulus *)malloc(sizeof(so_modulus) * size);
    for (size_t i = 0; i < size; i++)
        so_modulus_init(&b->base[i], vals[i]);
    b->base_prod = (uint64_t *)calloc(size, sizeof(uint64_t));
    b->punctured_prod = (uint64_t *)calloc(size * size, sizeof(uint64_t));
    b->inv_punctured_mod_base = (so_mulmod_op *)malloc(sizeof(so_mulmod_op) * size);
    if (size > 1)
    {
        uint64_t *tmp = (uint64_t *)calloc(size, sizeof(uint64_t));
        for (size_t i = 0; i < size; i++)
        {
            uint64_t *pp = b->punctured_prod + i * size;
            memset(pp, 0, size * sizeof(uint64_t));
            pp[0] = 1;
            for (size_t j = 0; j < size; j++)
            {
                if (j == i)
                    continue;
                mp_mul_word(pp, size, vals[j], tmp);
                memcpy(pp, tmp, size * sizeof(uint64_t));
            }
        }
        mp_mul_word(b->punctured_prod, size, vals[0], b->base_prod);
        for (size_t i = 0; i < size; i++)
        {
            uint64_t t = so_modulo_uint(b->punctured_prod + i * size, size, &b->base[i]);
            uint64_t inv = 0;
            so_try_invert_uint_mod(t, vals[i], &inv);
            so_mulmod_op_set(&b->inv_punctured_mod_base[i], inv, &b->base[i]);
        }
        free(tmp);
    }
    else
    {
        b->base_prod[0] = vals[0];
        b->punctured_prod[0] = 1;
        so_mulmod_op_set(&b->inv_punctured_mod_base[0], 1, &b->base[0]);
    }
    return b;
}

static void rnsbase_destroy(so_rnsbase *b)
{
    if (!b)
        return;
    free(b->base);
    free(b->base_prod);
    free(b->punctured_prod);
    free(b->inv_punctured_mod_base);
    free(b);
}

typedef struct
{
    so_rnsbase *ibase, *obase; /* not owned */
    uint64_t *matrix;          /* obase.size x ibase.size; util/rns.cpp:541-562 */
} base_conv;

static base_conv *conv_create(so_rnsbase *ibase, so_rnsbase *obase)
{
    base_conv *c = (base_conv *)calloc(1, sizeof(*c));
    c->ibase = ibase;
    c->obase = obase;
    c->matrix = (uint64_t *)malloc(sizeof(uint64_t) * ibase->size * obase->size);
    for (size_t o = 0; o < obase->size; o++)
        for (size_t i = 0; i < ibase->size; i++)
            c->matrix[o * ibase->size + i] =
                so_modulo_uint(ibase->punctured_prod + i * ibase->size, ibase->size, &obase->base[o]);
    return c;
}

static void conv_destroy(base_conv *c)
{
    if (!c)
        return;
    free(c->matrix);
    free(c);
}

/* util/rns.cpp:418-463 BaseConverter::fast_convert_array; in: [isz][count], out: [osz][count] */
static void conv_fast_convert_array(const base_conv *c, const uint64_t *in, size_t count, uint64_t *out)
{
    size_t isz = c->ibase->size, osz = c->obase->size;
    uint64_t *temp = (uint64_t *)malloc(sizeof(uint64_t) * count * isz);
    for (size_t i = 0; i < isz; i++)
    {
        const so_modulus *m = &c->ibase->base[i];
        so_mulmod_op inv = c->ibase->inv_punctured_mod_base[i];
        if (inv.operand == 1)
            for (size_t k = 0; k < count; k++)
                temp[k * isz + i] = so_barrett_reduce_64(in[i * count + k], m);
        else
            for (size_t k = 0; k < count; k++)
                temp[k * isz + i] = so_multiply_uint_mod_op(in[i * count + k], inv, m);
    }
    for (size_t o = 0; o < osz; o++)
        for (size_t k = 0; k < count; k++)
            out[o * count + k] = so_dot_product_mod(temp + k * isz, c->matrix + o * isz, isz, &c->obase->base[o]);
    free(temp);
}

void so_fast_convert_array(const uint64_t *ibase, size_t isz, const uint64_t *obase, size_t osz, const uint64_t *in,
                           size_t count, uint64_t *out)
{
    so_rnsbase *ib = rnsbase_create(ibase, isz), *ob = rnsbase_create(obase, osz);
    base_conv *c = conv_create(ib, ob);
    conv_fast_convert_array(c, in, count, out);
    conv_destroy(c);
    rnsbase_destroy(ib);
    rnsbase_destroy(ob);
}

struct so_rnstool
{
    size_t n;
    so_modulus t;
    so_rnsbase *base_q, *base_B, *base_Bsk, *base_Bsk_m_tilde, *base_m_tilde, *base_m_sk;
    base_conv *q_to_Bsk, *q_to_m_tilde, *B_to_q, *B_to_m_sk;
    so_modulus m_sk, gamma, m_tilde;
    uint64_t *prod_B_mod_q;
    so_mulmod_op *inv_prod_q_mod_Bsk;
    so_mulmod_op inv_prod_B_mod_m_sk;
    so_mulmod_op *inv_m_tilde_mod_Bsk;
    so_mulmod_op neg_inv_prod_q_mod_m_tilde;
    uint64_t *prod_q_mod_Bsk;
    so_mulmod_op *inv_q_last_mod_q;
    uint64_t inv_q_last_mod_t; /* rns.cpp:778-786, BGV */
    so_ntt_tables **Bsk_tables;
    int log_n;
};

/* util/rns.cpp:578-787 RNSTool::initialize (BFV decryption t/gamma parts are out of scope) */
static so_rnstool *rnstool_create_ex(size_t n, const uint64_t *q, size_t q_size, uint64_t t, int with_aux)
{
    so_rnstool *r = (so_rnstool *)calloc(1, sizeof(*r));
    r->n = n;
    r->log_n = sig_bits(n) - 1;
    so_modulus_init(&r->t, t);
    r->base_q = rnsbase_create(q, q_size);

    int total_bits = 0;
    for (size_t i = q_size; i--;)
        if (r->base_q->base_prod[i])
        {
            total_bits = (int)(64 * i) + sig_bits(r->base_q->base_prod[i]);
            break;
        }
    size_t B_size = q_size;
    if (32 + r->t.bit_count + total_bits >= 61 * (int)q_size + 61)
        B_size++;
    size_t Bsk_size = B_size + 1;
    size_t Bsk_mt_size = Bsk_size + 1;

    /* rns.cpp:625-635: get_primes(2n, 61, Bsk_m_tilde_size): m_sk, gamma, then base B */
    uint64_t *primes = (uint64_t *)malloc(sizeof(uint64_t) * Bsk_mt_size);
    if (!with_aux || r->log_n < 1 || so_get_primes(2 * (uint64_t)n, 61, Bsk_mt_size, primes) != 0)
    {
        /* the BEHZ auxiliary bases are only needed by the BFV path; rescale data below stays valid */
        free(primes);
        primes = NULL;
    }
    if (primes)
    {
        so_modulus_init(&r->m_sk, primes[0]);
        so_modulus_init(&r->gamma, primes[1]);
        so_modulus_init(&r->m_tilde, (uint64_t)1 << 32);
        uint64_t *bsk = (uint64_t *)malloc(sizeof(uint64_t) * Bsk_mt_size);
        memcpy(bsk, primes + 2, sizeof(uint64_t) * B_size);
        bsk[B_size] = primes[0];
        bsk[B_size + 1] = r->m_tilde.value;
        r->base_B = rnsbase_create(bsk, B_size);
        r->base_Bsk = rnsbase_create(bsk, Bsk_size);
        r->base_Bsk_m_tilde = rnsbase_create(bsk, Bsk_mt_size);
        r->base_m_tilde = rnsbase_create(&r->m_tilde.value, 1);
        r->base_m_sk = rnsbase_create(&r->m_sk.value, 1);
        r->q_to_Bsk = conv_create(r->base_q, r->base_Bsk);
        r->q_to_m_tilde = conv_create(r->base_q, r->base_m_tilde);
        r->B_to_q = conv_create(r->base_B, r->base_q);
        r->B_to_m_sk = conv_create(r->base_B, r->base_m_sk);

        r->Bsk_tables = (so_ntt_tables **)calloc(Bsk_size, sizeof(so_ntt_tables *));
        for (size_t i = 0; i < Bsk_size; i++)
            r->Bsk_tables[i] = so_ntt_tables_create(r->log_n, bsk[i]);

        r->prod_B_mod_q = (uint64_t *)malloc(sizeof(uint64_t) * q_size);
        for (size_t i = 0; i < q_size; i++)
            r->prod_B_mod_q[i] = so_modulo_uint(r->base_B->base_prod, B_size, &r->base_q->base[i]);

        r->inv_prod_q_mod_Bsk = (so_mulmod_op *)malloc(sizeof(so_mulmod_op) * Bsk_size);
        r->inv_m_tilde_mod_Bsk = (so_mulmod_op *)malloc(sizeof(so_mulmod_op) * Bsk_size);
        r->prod_q_mod_Bsk = (uint64_t *)malloc(sizeof(uint64_t) * Bsk_size);
        uint64_t temp;
        for (size_t i = 0; i < Bsk_size; i++)
        {
            const so_modulus *m = &r->base_Bsk->base[i];
            temp = so_modulo_uint(r->base_q->base_prod, q_size, m);
            r->prod_q_mod_Bsk[i] = temp;
            so_try_invert_uint_mod(temp, m->value, &temp);
            so_mulmod_op_set(&r->inv_prod_q_mod_Bsk[i], temp, m);
            so_try_invert_uint_mod(so_barrett_reduce_64(r->m_tilde.value, m), m->value, &temp);
            so_mulmod_op_set(&r->inv_m_tilde_mod_Bsk[i], temp, m);
        }
        temp = so_modulo_uint(r->base_B->base_prod, B_size, &r->m_sk);
        so_try_invert_uint_mod(temp, r->m_sk.value, &temp);
        so_mulmod_op_set(&r->inv_prod_B_mod_m_sk, temp, &r->m_sk);

        temp = so_modulo_uint(r->base_q->base_prod, q_size, &r->m_tilde);
        so_try_invert_uint_mod(temp, r->m_tilde.value, &temp);
        so_mulmod_op_set(&r->neg_inv_prod_q_mod_m_tilde, so_negate_uint_mod(temp, &r->m_tilde), &r->m_tilde);
        free(bsk);
        free(primes);
    }

    /* rns.cpp:767-776 */
    if (q_size > 1)
    {
        r->inv_q_last_mod_q = (so_mulmod_op *)malloc(sizeof(so_mulmod_op) * (q_size - 1));
        for (size_t i = 0; i + 1 < q_size; i++)
        {
            uint64_t temp = 0;
            so_try_invert_uint_mod(q[q_size - 1], q[i], &temp);
            so_mulmod_op_set(&r->inv_q_last_mod_q[i], temp, &r->base_q->base[i]);
        }
    }
    /* rns.cpp:778-786 */
    r->inv_q_last_mod_t = 1;
    if (t != 0)
        so_try_invert_uint_mod(q[q_size - 1], t, &r->inv_q_last_mod_t);
    return r;
}

so_rnstool *so_rnstool_create(size_t n, const uint64_t *q, size_t q_size, uint64_t t)
{
    return rnstool_create_ex(n, q, q_size, t, 1);
}

void so_rnstool_destroy(so_rnstool *r)
{
    if (!r)
        return;
    if (r->Bsk_tables)
    {
        for (size_t i = 0; i < r->base_Bsk->size; i++)
            so_ntt_tables_destroy(r->Bsk_tables[i]);
        free(r->Bsk_tables);
    }
    conv_destroy(r->q_to_Bsk);
    conv_destroy(r->q_to_m_tilde);
    conv_destroy(r->B_to_q);
    conv_destroy(r->B_to_m_sk);
    rnsbase_destroy(r->base_q);
    rnsbase_destroy(r->base_B);
    rnsbase_destroy(r->base_Bsk);
    rnsbase_destroy(r->base_Bsk_m_tilde);
    rnsbase_destroy(r->base_m_tilde);
    rnsbase_destroy(r->base_m_sk);
    free(r->prod_B_mod_q);
    free(r->inv_prod_q_mod_Bsk);
    free(r->inv_m_tilde_mod_Bsk);
    free(r->prod_q_mod_Bsk);
    free(r->inv_q_last_mod_q);
    free(r);
}

size_t so_rnstool_base_B_size(const so_rnstool *r)
{
    return r->base_B ? r->base_B->size : 0;
}

size_t so_rnstool_base_Bsk_size(const so_rnstool *r)
{
    return r->base_Bsk ? r->base_Bsk->size : 0;
}

void so_rnstool_base_Bsk(const so_rnstool *r, uint64_t *out)
{
    for (size_t i = 0; i < r->base_Bsk->size; i++)
        out[i] = r->base_Bsk->base[i].value;
}

uint64_t so_rnstool_m_tilde(const so_rnstool *r)
{
    return r->m_tilde.value;
}

void so_rnstool_inv_q_last_mod_q(const so_rnstool *r, uint64_t *operands)
{
    for (size_t i = 0; i + 1 < r->base_q->size; i++)
        operands[i] = r->inv_q_last_mod_q[i].operand;
}

/* util/rns.cpp:789-828 (BFV / coefficient domain) */
void so_divide_and_round_q_last_inplace(const so_rnstool *r, uint64_t *input)
{
    size_t L = r->base_q->size, n = r->n;
    uint64_t *last = input + (L - 1) * n;
    const so_modulus *lm = &r->base_q->base[L - 1];
    uint64_t half = lm->value >> 1;
    so_add_poly_scalar_coeffmod(last, n, half, lm, last);
    uint64_t *temp = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (size_t i = 0; i + 1 < L; i++)
    {
        const so_modulus *m = &r->base_q->base[i];
        so_modulo_poly_coeffs(last, n, m, temp);
        uint64_t half_mod = so_barrett_reduce_64(half, m);
        so_sub_poly_scalar_coeffmod(temp, n, half_mod, m, temp);
        so_sub_poly_coeffmod(input + i * n, temp, n, m, input + i * n);
        mul_poly_op(input + i * n, n, r->inv_q_last_mod_q[i], m, input + i * n);
    }
    free(temp);
}

/* util/rns.cpp:1203-1246 (BGV / NTT domain) */
void so_mod_t_and_divide_q_last_ntt_inplace(const so_rnstool *r, uint64_t *input, so_ntt_tables *const *tables)
{
    size_t L = r->base_q->size, n = r->n;
    const so_modulus *lm = &r->base_q->base[L - 1];
    uint64_t *c_last = input + (L - 1) * n;
    uint64_t *neg = (uint64_t *)malloc(sizeof(uint64_t) * n), *delta = (uint64_t *)malloc(sizeof(uint64_t) * n);
    so_ntt_inverse(c_last, tables[L - 1], 0);
    so_modulo_poly_coeffs(c_last, n, &r->t, neg);
    so_negate_poly_coeffmod(neg, n, &r->t, neg);
    if (r->inv_q_last_mod_t != 1)
        so_multiply_poly_scalar_coeffmod(neg, n, r->inv_q_last_mod_t, &r->t, neg);
    for (size_t i = 0; i + 1 < L; i++)
    {
        const so_modulus *m = &r->base_q->base[i];
        so_modulo_poly_coeffs(neg, n, m, delta);
        so_multiply_poly_scalar_coeffmod(delta, n, lm->value, m, delta);
        for (size_t k = 0; k < n; k++)
            delta[k] = so_add_uint_mod(delta[k], so_barrett_reduce_64(c_last[k], m), m);
        so_ntt_forward(delta, tables[i], 0);
        so_sub_poly_coeffmod(input + i * n, delta, n, m, input + i * n);
        mul_poly_op(input + i * n, n, r->inv_q_last_mod_q[i], m, input + i * n);
    }
    free(neg);
    free(delta);
}

uint64_t so_rnstool_inv_q_last_mod_t(const so_rnstool *r)
{
    return r->inv_q_last_mod_t;
}

/* util/rns.cpp:830-901 (CKKS / NTT domain), SEAL_USER_MOD_BIT_COUNT_MAX <= 60 branch (defines.h:47) */
void so_divide_and_round_q_last_ntt_inplace(const so_rnstool *r, uint64_t *input, so_ntt_tables *const *tables)
{
    size_t L = r->base_q->size, n = r->n;
    uint64_t *last = input + (L - 1) * n;
    const so_modulus *lm = &r->base_q->base[L - 1];
    so_ntt_inverse(last, tables[L - 1], 0);
    uint64_t half = lm->value >> 1;
    so_add_poly_scalar_coeffmod(last, n, half, lm, last);
    uint64_t *temp = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (size_t i = 0; i + 1 < L; i++)
    {
        const so_modulus *m = &r->base_q->base[i];
        if (m->value < lm->value)
            so_modulo_poly_coeffs(last, n, m, temp);
        else
            memcpy(temp, last, sizeof(uint64_t) * n);
        uint64_t neg_half_mod = m->value - so_barrett_reduce_64(half, m);
        for (size_t k = 0; k < n; k++)
            temp[k] += neg_half_mod;
        uint64_t qi_lazy = m->value << 2;
        so_ntt_forward(temp, tables[i], 1);
        uint64_t *dst = input + i * n;
        for (size_t k = 0; k < n; k++)
            dst[k] += qi_lazy - temp[k];
        mul_poly_op(dst, n, r->inv_q_last_mod_q[i], m, dst);
    }
    free(temp);
}

/* util/rns.cpp:1092-1131: in: base q [q][n] -> out: Bsk U {m_tilde} */
void so_fastbconv_m_tilde(const so_rnstool *r, const uint64_t *in, uint64_t *out)
{
    size_t qs = r->base_q->size, n = r->n, Bsk = r->base_Bsk->size;
    uint64_t *temp = (uint64_t *)malloc(sizeof(uint64_t) * qs * n);
    for (size_t i = 0; i < qs; i++)
        so_multiply_poly_scalar_coeffmod(in + i * n, n, r->m_tilde.value, &r->base_q->base[i], temp + i * n);
    conv_fast_convert_array(r->q_to_Bsk, temp, n, out);
    conv_fast_convert_array(r->q_to_m_tilde, temp, n, out + Bsk * n);
    free(temp);
}

/* util/rns.cpp:981-1041: in: Bsk U {m_tilde} -> out: Bsk */
void so_sm_mrq(const so_rnstool *r, const uint64_t *in, uint64_t *out)
{
    size_t n = r->n, Bsk = r->base_Bsk->size;
    const uint64_t *in_mt = in + Bsk * n;
    uint64_t mt_div2 = r->m_tilde.value >> 1;
    uint64_t *r_mt = (uint64_t *)malloc(sizeof(uint64_t) * n);
    mul_poly_op(in_mt, n, r->neg_inv_prod_q_mod_m_tilde, &r->m_tilde, r_mt);
    for (size_t i = 0; i < Bsk; i++)
    {
        const so_modulus *m = &r->base_Bsk->base[i];
        so_mulmod_op pq;
        so_mulmod_op_set(&pq, r->prod_q_mod_Bsk[i], m);
        for (size_t k = 0; k < n; k++)
        {
            uint64_t temp = r_mt[k];
            if (temp >= mt_div2)
                temp += m->value - r->m_tilde.value;
            out[i * n + k] = so_multiply_uint_mod_op(multiply_add_uint_mod_op(temp, pq, in[i * n + k], m),
                                                     r->inv_m_tilde_mod_Bsk[i], m);
        }
    }
    free(r_mt);
}

/* util/rns.cpp:1043-1090: in: q U Bsk -> out: Bsk */
void so_fast_floor(const so_rnstool *r, const uint64_t *in, uint64_t *out)
{
    size_t qs = r->base_q->size, n = r->n, Bsk = r->base_Bsk->size;
    conv_fast_convert_array(r->q_to_Bsk, in, n, out);
    in += qs * n;
    for (size_t i = 0; i < Bsk; i++)
    {
        const so_modulus *m = &r->base_Bsk->base[i];
        for (size_t k = 0; k < n; k++)
            out[i * n + k] =
                so_multiply_uint_mod_op(in[i * n + k] + (m->value - out[i * n + k]), r->inv_prod_q_mod_Bsk[i], m);
    }
}

/* util/rns.cpp:903-979: in: Bsk -> out: q (Shenoy-Kumaresan) */
void so_fastbconv_sk(const so_rnstool *r, const uint64_t *in, uint64_t *out)
{
    size_t qs = r->base_q->size, n = r->n, Bs = r->base_B->size;
    conv_fast_convert_array(r->B_to_q, in, n, out);
    uint64_t *temp = (uint64_t *)malloc(sizeof(uint64_t) * n);
    uint64_t *alpha = (uint64_t *)malloc(sizeof(uint64_t) * n);
    conv_fast_convert_array(r->B_to_m_sk, in, n, temp);
    const uint64_t *in_sk = in + Bs * n;
    for (size_t k = 0; k < n; k++)
        alpha[k] = so_multiply_uint_mod_op(temp[k] + (r->m_sk.value - in_sk[k]), r->inv_prod_B_mod_m_sk, &r->m_sk);
    uint64_t m_sk_div2 = r->m_sk.value >> 1;
    for (size_t i = 0; i < qs; i++)
    {
        const so_modulus *m = &r->base_q->base[i];
        so_mulmod_op pb, npb;
        so_mulmod_op_set(&pb, r->prod_B_mod_q[i], m);
        so_mulmod_op_set(&npb, m->value - r->prod_B_mod_q[i], m);
        for (size_t k = 0; k < n; k++)
        {
            if (alpha[k] > m_sk_div2)
                out[i * n + k] =
                    multiply_add_uint_mod_op(so_negate_uint_mod(alpha[k], &r->m_sk), pb, out[i * n + k], m);
            else
                out[i * n + k] = multiply_add_uint_mod_op(alpha[k], npb, out[i * n + k], m);
        }
    }
    free(temp);
    free(alpha);
}

/* ------------------------------------------------------------------------------------------------
 * Context (modulus chain levels: context.cu:607-637, 726-765) and Evaluator-level ops
 * ---------------------------------------------------------------------------------------------- */

struct so_context
{
    int log_n;
    size_t n, k;
    int scheme;
    uint64_t plain_modulus;
    so_modulus *mod;        /* k key-level primes; level with L limbs uses mod[0..L-1] */
    so_ntt_tables **tables; /* k */
    so_rnstool **tools;     /* tools[L], L = 1..k, created lazily (BFV aux bases are costly) */
    uint64_t *primes;
};

so_context *so_context_create(int log_n, const uint64_t *primes, size_t k, int scheme, uint64_t plain_modulus)
{
    so_context *c = (so_context *)calloc(1, sizeof(*c));
    c->log_n = log_n;
    c->n = (size_t)1 << log_n;
    c->k = k;
    c->scheme = scheme;
    c->plain_modulus = plain_modulus;
    c->mod = (so_modulus *)malloc(sizeof(so_modulus) * k);
    c->tables = (so_ntt_tables **)calloc(k, sizeof(so_ntt_tables *));
    c->tools = (so_rnstool **)calloc(k + 1, sizeof(so_rnstool *));
    c->primes = (uint64_t *)malloc(sizeof(uint64_t) * k);
    memcpy(c->primes, primes, sizeof(uint64_t) * k);
    for (size_t i = 0; i < k; i++)
    {
        so_modulus_init(&c->mod[i], primes[i]);
        c->tables[i] = so_ntt_tables_create(log_n, primes[i]);
        if (!c->tables[i])
        {
            so_context_destroy(c);
            return NULL;
        }
    }
    return c;
}

void so_context_destroy(so_context *c)
{
    if (!c)
        return;
    for (size_t i = 0; i < c->k; i++)
        so_ntt_tables_destroy(c->tables[i]);
    for (size_t i = 0; i <= c->k; i++)
        so_rnstool_destroy(c->tools[i]);
    free(c->tables);
    free(c->tools);
    free(c->mod);
    free(c->primes);
    free(c);
}

const so_ntt_tables *so_context_tables(const so_context *c, size_t i)
{
    return c->tables[i];
}

const so_rnstool *so_context_rnstool(so_context *c, size_t L)
{
    if (!c->tools[L])
        c->tools[L] = rnstool_create_ex(c->n, c->primes, L,
                                        c->scheme == SO_SCHEME_BFV || c->scheme == SO_SCHEME_BGV ? c->plain_modulus : 0,
                                        c->scheme == SO_SCHEME_BFV);
    return c->tools[L];
}

/* evaluator.cpp:604-663 (dest_size == 3 path; tiling does not change results) */
void so_ckks_multiply(const so_context *c, size_t L, uint64_t *x, const uint64_t *y)
{
    size_t n = c->n;
    uint64_t *temp = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (size_t i = 0; i < L; i++)
    {
        const so_modulus *m = &c->mod[i];
        uint64_t *x0 = x + i * n, *x1 = x + (L + i) * n, *x2 = x + (2 * L + i) * n;
        const uint64_t *y0 = y + i * n, *y1 = y + (L + i) * n;
        so_dyadic_product_coeffmod(x1, y1, n, m, x2);
        so_dyadic_product_coeffmod(x1, y0, n, m, temp);
        so_dyadic_product_coeffmod(x0, y1, n, m, x1);
        so_add_poly_coeffmod(x1, temp, n, m, x1);
        so_dyadic_product_coeffmod(x0, y0, n, m, x0);
    }
    free(temp);
}

/* evaluator.cpp:1022-1078 (size 2 -> 3): x has room for 3 polys */
void so_ckks_square(const so_context *c, size_t L, uint64_t *x)
{
    size_t n = c->n;
    for (size_t i = 0; i < L; i++)
    {
        const so_modulus *m = &c->mod[i];
        uint64_t *x0 = x + i * n, *x1 = x + (L + i) * n, *x2 = x + (2 * L + i) * n;
        so_dyadic_product_coeffmod(x1, x1, n, m, x2);
        so_dyadic_product_coeffmod(x0, x1, n, m, x1);
        so_add_poly_coeffmod(x1, x1, n, m, x1);
        so_dyadic_product_coeffmod(x0, x0, n, m, x0);
    }
}

/* evaluator.cpp:2135-2172: every polynomial of the ciphertext times the NTT-form plaintext (L limbs) */
void so_multiply_plain_ntt(const so_context *c, size_t L, uint64_t *ct, size_t size, const uint64_t *plain)
{
    size_t n = c->n;
    for (size_t p = 0; p < size; p++)
        for (size_t i = 0; i < L; i++)
            so_dyadic_product_coeffmod(ct + (p * L + i) * n, plain + i * n, n, &c->mod[i], ct + (p * L + i) * n);
}

/* evaluator.cpp:1737-1843 / 1845-1951, CKKS branch (:1812-1817, :1920-1925): c0 +- plain */
void so_add_plain_ckks(const so_context *c, size_t L, uint64_t *ct, const uint64_t *plain, int subtract)
{
    size_t n = c->n;
    for (size_t i = 0; i < L; i++)
    {
        if (subtract)
            so_sub_poly_coeffmod(ct + i * n, plain + i * n, n, &c->mod[i], ct + i * n);
        else
            so_add_poly_coeffmod(ct + i * n, plain + i * n, n, &c->mod[i], ct + i * n);
    }
}

/* evaluator.cpp:664-700 */
void so_ckks_multiply_general(const so_context *c, size_t L, uint64_t *x, size_t xs, const uint64_t *y, size_t ys)
{
    size_t n = c->n, ds = xs + ys - 1, poly = L * n;
    uint64_t *temp = (uint64_t *)calloc(ds * poly, sizeof(uint64_t));
    uint64_t *prod = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (size_t I = 0; I < ds; I++)
    {
        size_t x_last = I < xs - 1 ? I : xs - 1;
        size_t y_first = I < ys - 1 ? I : ys - 1;
        size_t x_first = I - y_first;
        size_t steps = x_last - x_first + 1;
        for (size_t s = 0; s < steps; s++)
        {
            const uint64_t *xp = x + (x_first + s) * poly;
            const uint64_t *yp = y + (y_first - s) * poly;
            for (size_t i = 0; i < L; i++)
            {
                so_dyadic_product_coeffmod(xp + i * n, yp + i * n, n, &c->mod[i], prod);
                so_add_poly_coeffmod(prod, temp + I * poly + i * n, n, &c->mod[i], temp + I * poly + i * n);
            }
        }
    }
    memcpy(x, temp, sizeof(uint64_t) * ds * poly);
    free(temp);
    free(prod);
}

/* evaluator.cpp:395-567, sizes 2 x 2 -> 3, coefficient form, level with L limbs */
void so_bfv_multiply(so_context *c, size_t L, uint64_t *x, const uint64_t *y)
{
    const so_rnstool *r = so_context_rnstool(c, L);
    size_t n = c->n, Bsk = r->base_Bsk->size;
    size_t pq = L * n, pB = Bsk * n;
    uint64_t *xq = (uint64_t *)malloc(sizeof(uint64_t) * 2 * pq), *yq = (uint64_t *)malloc(sizeof(uint64_t) * 2 * pq);
    uint64_t *xB = (uint64_t *)malloc(sizeof(uint64_t) * 2 * pB), *yB = (uint64_t *)malloc(sizeof(uint64_t) * 2 * pB);
    uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * (Bsk + 1) * n);
    const uint64_t *src[2] = { x, y };
    uint64_t *dq[2] = { xq, yq }, *dB[2] = { xB, yB };
    /* steps (1)-(3), :452-490 */
    for (int w = 0; w < 2; w++)
        for (size_t p = 0; p < 2; p++)
        {
            memcpy(dq[w] + p * pq, src[w] + p * pq, sizeof(uint64_t) * pq);
            for (size_t i = 0; i < L; i++)
                so_ntt_forward(dq[w] + p * pq + i * n, c->tables[i], 1);
            so_fastbconv_m_tilde(r, src[w] + p * pq, tmp);
            so_sm_mrq(r, tmp, dB[w] + p * pB);
            for (size_t i = 0; i < Bsk; i++)
                so_ntt_forward(dB[w] + p * pB + i * n, r->Bsk_tables[i], 1);
        }
    /* step (4), :493-541 */
    uint64_t *tq = (uint64_t *)calloc(3 * pq, sizeof(uint64_t)), *tB = (uint64_t *)calloc(3 * pB, sizeof(uint64_t));
    uint64_t *prod = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (size_t I = 0; I < 3; I++)
    {
        size_t x_last = I < 1 ? I : 1, y_first = I < 1 ? I : 1, x_first = I - y_first;
        size_t steps = x_last - x_first + 1;
        for (size_t s = 0; s < steps; s++)
        {
            size_t xi = x_first + s, yi = y_first - s;
            for (size_t i = 0; i < L; i++)
            {
                so_dyadic_product_coeffmod(xq + xi * pq + i * n, yq + yi * pq + i * n, n, &c->mod[i], prod);
                so_add_poly_coeffmod(prod, tq + I * pq + i * n, n, &c->mod[i], tq + I * pq + i * n);
            }
            for (size_t i = 0; i < Bsk; i++)
            {
                const so_modulus *m = &r->base_Bsk->base[i];
                so_dyadic_product_coeffmod(xB + xi * pB + i * n, yB + yi * pB + i * n, n, m, prod);
                so_add_poly_coeffmod(prod, tB + I * pB + i * n, n, m, tB + I * pB + i * n);
            }
        }
    }
    /* step (5), :545-546 */
    for (size_t I = 0; I < 3; I++)
    {
        for (size_t i = 0; i < L; i++)
            so_ntt_inverse(tq + I * pq + i * n, c->tables[i], 1);
        for (size_t i = 0; i < Bsk; i++)
            so_ntt_inverse(tB + I * pB + i * n, r->Bsk_tables[i], 1);
    }
    /* steps (6)-(8), :549-566 */
    uint64_t *tqB = (uint64_t *)malloc(sizeof(uint64_t) * (pq + pB));
    uint64_t *tBsk = (uint64_t *)malloc(sizeof(uint64_t) * pB);
    for (size_t I = 0; I < 3; I++)
    {
        for (size_t i = 0; i < L; i++)
            so_multiply_poly_scalar_coeffmod(tq + I * pq + i * n, n, c->plain_modulus, &c->mod[i], tqB + i * n);
        for (size_t i = 0; i < Bsk; i++)
            so_multiply_poly_scalar_coeffmod(tB + I * pB + i * n, n, c->plain_modulus, &r->base_Bsk->base[i],
                                             tqB + pq + i * n);
        so_fast_floor(r, tqB, tBsk);
        so_fastbconv_sk(r, tBsk, x + I * pq);
    }
    free(xq);
    free(yq);
    free(xB);
    free(yB);
    free(tmp);
    free(tq);
    free(tB);
    free(prod);
    free(tqB);
    free(tBsk);
}

/* evaluator.cpp:2523-2820, CKKS and BFV branches (BGV out of scope).
 * d = L (decomp_modulus_size), k = key_modulus_size, modswitch_factors = key-level inv_q_last_mod_q. */
void so_switch_key_inplace(const so_context *c, size_t L, uint64_t *encrypted, const uint64_t *target,
                           const uint64_t *key)
{
    const size_t n = c->n, k = c->k, d = L, rns = d + 1;
    const int bgv = c->scheme == SO_SCHEME_BGV;
    const int ckks = c->scheme == SO_SCHEME_CKKS || bgv; /* NTT-form target: :2608, :2639 */
    so_context *cm = (so_context *)c;
    const so_rnstool *key_tool = so_context_rnstool(cm, k);

    /* :2604-2611 */
    uint64_t *t_target = (uint64_t *)malloc(sizeof(uint64_t) * d * n);
    memcpy(t_target, target, sizeof(uint64_t) * d * n);
    if (ckks)
        for (size_t j = 0; j < d; j++)
            so_ntt_inverse(t_target + j * n, c->tables[j], 0);

    /* :2614 t_poly_prod[comp][I][n] */
    uint64_t *t_poly_prod = (uint64_t *)calloc(2 * rns * n, sizeof(uint64_t));
    uint64_t *t_ntt = (uint64_t *)malloc(sizeof(uint64_t) * n);
    u128 *acc = (u128 *)malloc(sizeof(u128) * 2 * n);

    for (size_t I = 0; I < rns; I++) /* :2616 */
    {
        size_t key_index = (I == d) ? k - 1 : I;
        const so_modulus *km = &c->mod[key_index];
        size_t bound = 256, counter = bound; /* defines.h:76 SEAL_MULTIPLY_ACCUMULATE_USER_MOD_MAX */
        memset(acc, 0, sizeof(u128) * 2 * n);
        for (size_t J = 0; J < d; J++) /* :2630 */
        {
            const uint64_t *t_operand;
            if (ckks && I == J)
                t_operand = target + J * n;
            else
            {
                if (c->mod[J].value <= km->value)
                    memcpy(t_ntt, t_target + J * n, sizeof(uint64_t) * n);
                else
                    so_modulo_poly_coeffs(t_target + J * n, n, km, t_ntt);
                so_ntt_forward(t_ntt, c->tables[key_index], 1);
                t_operand = t_ntt;
            }
            for (size_t comp = 0; comp < 2; comp++) /* :2658-2682 */
            {
                const uint64_t *kp = key + ((J * 2 + comp) * k + key_index) * n;
                u128 *a = acc + comp * n;
                if (!counter)
                {
                    for (size_t l = 0; l < n; l++)
                    {
                        u128 s = (u128)t_operand[l] * kp[l] + a[l];
                        uint64_t w[2] = { (uint64_t)s, (uint64_t)(s >> 64) };
                        a[l] = so_barrett_reduce_128(w, km);
                    }
                }
                else
                    for (size_t l = 0; l < n; l++)
                        a[l] += (u128)t_operand[l] * kp[l];
            }
            if (!--counter)
                counter = bound;
        }
        for (size_t comp = 0; comp < 2; comp++) /* :2691-2708 */
        {
            uint64_t *dst = t_poly_prod + (comp * rns + I) * n;
            const u128 *a = acc + comp * n;
            if (counter == bound)
                for (size_t l = 0; l < n; l++)
                    dst[l] = (uint64_t)a[l];
            else
                for (size_t l = 0; l < n; l++)
                {
                    uint64_t w[2] = { (uint64_t)a[l], (uint64_t)(a[l] >> 64) };
                    dst[l] = so_barrett_reduce_128(w, km);
                }
        }
    }

    const so_modulus *pk = &c->mod[k - 1];
    const uint64_t qk = pk->value, qk_half = qk >> 1;
    /* :2715-2758 BGV branch */
    for (size_t comp = 0; bgv && comp < 2; comp++)
    {
        so_modulus tm;
        so_modulus_init(&tm, c->plain_modulus);
        uint64_t *t_last = t_poly_prod + (comp * rns + d) * n;
        uint64_t *kk = (uint64_t *)malloc(sizeof(uint64_t) * n), *cq = (uint64_t *)malloc(sizeof(uint64_t) * n);
        so_ntt_inverse(t_last, c->tables[k - 1], 0);
        so_modulo_poly_coeffs(t_last, n, &tm, kk);
        so_negate_poly_coeffmod(kk, n, &tm, kk);
        if (key_tool->inv_q_last_mod_t != 1)
            so_multiply_poly_scalar_coeffmod(kk, n, key_tool->inv_q_last_mod_t, &tm, kk);
        for (size_t j = 0; j < d; j++)
        {
            const so_modulus *m = &c->mod[j];
            so_modulo_poly_coeffs(kk, n, m, t_ntt);
            so_multiply_poly_scalar_coeffmod(t_ntt, n, qk, m, t_ntt);
            so_modulo_poly_coeffs(t_last, n, m, cq);
            so_add_poly_coeffmod(t_ntt, cq, n, m, t_ntt);
            so_ntt_forward(t_ntt, c->tables[j], 0);
            uint64_t *tp = t_poly_prod + (comp * rns + j) * n;
            so_sub_poly_coeffmod(tp, t_ntt, n, m, tp);
            mul_poly_op(tp, n, key_tool->inv_q_last_mod_q[j], m, tp);
            uint64_t *e = encrypted + (comp * d + j) * n;
            so_add_poly_coeffmod(tp, e, n, m, e);
        }
        free(kk);
        free(cq);
    }
    /* :2759-2818 non-BGV branch */
    for (size_t comp = 0; !bgv && comp < 2; comp++)
    {
        uint64_t *t_last = t_poly_prod + (comp * rns + d) * n;
        so_ntt_inverse(t_last, c->tables[k - 1], 1);
        for (size_t l = 0; l < n; l++)
            t_last[l] = so_barrett_reduce_64(t_last[l] + qk_half, pk);
        for (size_t j = 0; j < d; j++)
        {
            const so_modulus *m = &c->mod[j];
            uint64_t qi = m->value;
            if (qk > qi)
                so_modulo_poly_coeffs(t_last, n, m, t_ntt);
            else
                memcpy(t_ntt, t_last, sizeof(uint64_t) * n);
            uint64_t fix = qi - so_barrett_reduce_64(qk_half, m);
            for (size_t l = 0; l < n; l++)
                t_ntt[l] += fix;
            uint64_t qi_lazy = qi << 1;
            uint64_t *tp = t_poly_prod + (comp * rns + j) * n;
            if (ckks)
            {
                so_ntt_forward(t_ntt, c->tables[j], 1);
                qi_lazy = qi << 2; /* SEAL_USER_MOD_BIT_COUNT_MAX <= 60 */
            }
            else
                so_ntt_inverse(tp, c->tables[j], 1);
            for (size_t l = 0; l < n; l++)
                tp[l] += qi_lazy - t_ntt[l];
            mul_poly_op(tp, n, key_tool->inv_q_last_mod_q[j], m, tp);
            uint64_t *e = encrypted + (comp * d + j) * n;
            so_add_poly_coeffmod(tp, e, n, m, e);
        }
    }
    free(t_target);
    free(t_poly_prod);
    free(t_ntt);
    free(acc);
}

/* evaluator.cpp:1144-1199, encrypted size 3 -> 2 (RelinKeys::get_index(2) = 0, relinkeys.h:58-64) */
void so_relinearize(const so_context *c, size_t L, uint64_t *ct, const uint64_t *relin_key)
{
    so_switch_key_inplace(c, L, ct, ct + 2 * L * c->n, relin_key);
}

/* evaluator.cpp:1201-1280 */
void so_mod_switch_scale_to_next(so_context *c, size_t L, const uint64_t *in, size_t size, uint64_t *out)
{
    const so_rnstool *r = so_context_rnstool(c, L);
    size_t n = c->n;
    uint64_t *copy = (uint64_t *)malloc(sizeof(uint64_t) * size * L * n);
    memcpy(copy, in, sizeof(uint64_t) * size * L * n);
    for (size_t p = 0; p < size; p++)
    {
        if (c->scheme == SO_SCHEME_CKKS)
            so_divide_and_round_q_last_ntt_inplace(r, copy + p * L * n, c->tables);
        else if (c->scheme == SO_SCHEME_BGV)
            so_mod_t_and_divide_q_last_ntt_inplace(r, copy + p * L * n, c->tables);
        else
            so_divide_and_round_q_last_inplace(r, copy + p * L * n);
        memcpy(out + p * (L - 1) * n, copy + p * L * n, sizeof(uint64_t) * (L - 1) * n);
    }
    free(copy);
}

/* evaluator.cpp:1282-1345 */
void so_mod_switch_drop_to_next(const so_context *c, size_t L, const uint64_t *in, size_t size, uint64_t *out)
{
    size_t n = c->n;
    for (size_t p = 0; p < size; p++)
        memmove(out + p * (L - 1) * n, in + p * L * n, sizeof(uint64_t) * (L - 1) * n);
}

/* evaluator.cpp:2362-2464 */
void so_apply_galois_inplace(const so_context *c, size_t L, uint64_t *ct, uint32_t elt, const uint64_t *galois_key)
{
    size_t n = c->n;
    uint64_t *temp = (uint64_t *)malloc(sizeof(uint64_t) * L * n);
    for (size_t p = 0; p < 2; p++)
    {
        for (size_t i = 0; i < L; i++)
        {
            if (c->scheme == SO_SCHEME_CKKS || c->scheme == SO_SCHEME_BGV) /* :2430-2443 */
                so_apply_galois_ntt(ct + (p * L + i) * n, c->log_n, elt, temp + i * n);
            else
                so_apply_galois(ct + (p * L + i) * n, c->log_n, elt, &c->mod[i], temp + i * n);
        }
        if (p == 0)
            memcpy(ct, temp, sizeof(uint64_t) * L * n);
    }
    memset(ct + L * n, 0, sizeof(uint64_t) * L * n);
    so_switch_key_inplace(c, L, ct, temp, galois_key);
    free(temp);
}

/* evaluator.cpp:2267-2360 */
void so_transform_to_ntt(const so_context *c, size_t L, uint64_t *ct, size_t size)
{
    for (size_t p = 0; p < size; p++)
        for (size_t i = 0; i < L; i++)
            so_ntt_forward(ct + (p * L + i) * c->n, c->tables[i], 0);
}

void so_transform_from_ntt(const so_context *c, size_t L, uint64_t *ct, size_t size)
{
    for (size_t p = 0; p < size; p++)
        for (size_t i = 0; i < L; i++)
            so_ntt_inverse(ct + (p * L + i) * c->n, c->tables[i], 0);
}

/* ------------------------------------------------------------------------------------------------
 * Key material and RLWE helpers for tests.  PRNG = xoshiro256** seeded by splitmix64 (harness's own).
 * ---------------------------------------------------------------------------------------------- */

typedef struct
{
    uint64_t s[4];
} rng_t;

static uint64_t splitmix64(uint64_t *x)
{
    uint64_t z = (*x += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

static void rng_seed(rng_t *r, uint64_t seed)
{
    for (int i = 0; i < 4; i++)
        r->s[i] = splitmix64(&seed);
}

static inline uint64_t rotl(uint64_t x, int k)
{
    return (x << k) | (x >> (64 - k));
}

static uint64_t rng_next(rng_t *r)
{
    uint64_t *s = r->s;
    uint64_t result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return result;
}

void so_fill_uniform(const so_context *c, size_t L, size_t polys, uint64_t seed, uint64_t *out)
{
    rng_t r;
    rng_seed(&r, seed);
    for (size_t p = 0; p < polys; p++)
        for (size_t i = 0; i < L; i++)
            for (size_t l = 0; l < c->n; l++)
                out[(p * L + i) * c->n + l] = rng_next(&r) % c->mod[i].value;
}

/* ternary secret in NTT form over all k key primes (keygenerator.cpp:56-120 shape) */
void so_sample_ternary_ntt(const so_context *c, uint64_t seed, uint64_t *sk)
{
    rng_t r;
    rng_seed(&r, seed);
    for (size_t l = 0; l < c->n; l++)
    {
        uint64_t v = rng_next(&r) % 3; /* 0,1,2 -> 0,1,-1 */
        for (size_t i = 0; i < c->k; i++)
            sk[i * c->n + l] = v == 2 ? c->mod[i].value - 1 : v;
    }
    for (size_t i = 0; i < c->k; i++)
        so_ntt_forward(sk + i * c->n, c->tables[i], 0);
}

/* centred binomial noise (sum of 21 coin differences, sigma ~ 3.2), coefficient form, limbs [0,L) + extra */
static void sample_noise(const so_context *c, const size_t *limbs, size_t nl, rng_t *r, uint64_t *out)
{
    for (size_t l = 0; l < c->n; l++)
    {
        uint64_t bits = rng_next(r);
        int e = __builtin_popcountll(bits & 0x1fffff) - __builtin_popcountll((bits >> 21) & 0x1fffff);
        for (size_t i = 0; i < nl; i++)
        {
            uint64_t q = c->mod[limbs[i]].value;
            out[i * c->n + l] = e >= 0 ? (uint64_t)e : q - (uint64_t)(-e);
        }
    }
}

/* (c0, c1) = (-(a s + e), a) over the given limbs, NTT form (util/rlwe.cpp:336-400) */
static void encrypt_zero_ntt(const so_context *c, const size_t *limbs, size_t nl, const uint64_t *sk, rng_t *r,
                             uint64_t *c0, uint64_t *c1)
{
    size_t n = c->n;
    sample_noise(c, limbs, nl, r, c0);
    for (size_t i = 0; i < nl; i++)
    {
        const so_modulus *m = &c->mod[limbs[i]];
        for (size_t l = 0; l < n; l++)
            c1[i * n + l] = rng_next(r) % m->value;
        so_ntt_forward(c0 + i * n, c->tables[limbs[i]], 0);
        if (c->scheme == SO_SCHEME_BGV) /* noise = t e (rlwe.cpp:379-385) */
            so_multiply_poly_scalar_coeffmod(c0 + i * n, n, c->plain_modulus, m, c0 + i * n);
        for (size_t l = 0; l < n; l++)
        {
            uint64_t as = so_multiply_uint_mod(c1[i * n + l], sk[limbs[i] * n + l], m);
            c0[i * n + l] = so_negate_uint_mod(so_add_uint_mod(as, c0[i * n + l], m), m);
        }
    }
}

/* keygenerator.cpp:301-334 generate_one_kswitch_key; key layout [k-1][2][k][N] */
void so_gen_kswitch_key(const so_context *c, const uint64_t *sk, const uint64_t *new_key, uint64_t seed, uint64_t *key)
{
    size_t n = c->n, k = c->k, d = k - 1;
    size_t *limbs = (size_t *)malloc(sizeof(size_t) * k);
    for (size_t i = 0; i < k; i++)
        limbs[i] = i;
    rng_t r;
    rng_seed(&r, seed);
    for (size_t J = 0; J < d; J++)
    {
        uint64_t *c0 = key + (J * 2 + 0) * k * n, *c1 = key + (J * 2 + 1) * k * n;
        encrypt_zero_ntt(c, limbs, k, sk, &r, c0, c1);
        const so_modulus *m = &c->mod[J];
        uint64_t factor = so_barrett_reduce_64(c->mod[k - 1].value, m);
        for (size_t l = 0; l < n; l++)
        {
            uint64_t t = so_multiply_uint_mod(new_key[J * n + l], factor, m);
            c0[J * n + l] = so_add_uint_mod(c0[J * n + l], t, m);
        }
    }
    free(limbs);
}

/* keygenerator.cpp:122-170: new key = s^2 */
void so_gen_relin_key(const so_context *c, const uint64_t *sk, uint64_t seed, uint64_t *key)
{
    size_t n = c->n, k = c->k;
    uint64_t *s2 = (uint64_t *)malloc(sizeof(uint64_t) * k * n);
    for (size_t i = 0; i < k; i++)
        so_dyadic_product_coeffmod(sk + i * n, sk + i * n, n, &c->mod[i], s2 + i * n);
    so_gen_kswitch_key(c, sk, s2, seed, key);
    free(s2);
}

/* keygenerator.cpp:172-260: new key = galois(s) (NTT-domain permutation) */
void so_gen_galois_key(const so_context *c, const uint64_t *sk, uint32_t elt, uint64_t seed, uint64_t *key)
{
    size_t n = c->n, k = c->k;
    uint64_t *rs = (uint64_t *)malloc(sizeof(uint64_t) * k * n);
    for (size_t i = 0; i < k; i++)
        so_apply_galois_ntt(sk + i * n, c->log_n, elt, rs + i * n);
    so_gen_kswitch_key(c, sk, rs, seed, key);
    free(rs);
}

void so_encrypt_symmetric(const so_context *c, size_t L, const uint64_t *sk, const uint64_t *msg, uint64_t seed,
                          int ntt_form, uint64_t *ct)
{
    size_t n = c->n;
    size_t *limbs = (size_t *)malloc(sizeof(size_t) * L);
    for (size_t i = 0; i < L; i++)
        limbs[i] = i;
    rng_t r;
    rng_seed(&r, seed);
    uint64_t *c0 = ct, *c1 = ct + L * n;
    encrypt_zero_ntt(c, limbs, L, sk, &r, c0, c1);
    uint64_t *m = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (size_t i = 0; i < L; i++)
    {
        memcpy(m, msg + i * n, sizeof(uint64_t) * n);
        if (ntt_form)
            so_ntt_forward(m, c->tables[i], 0);
        else
        {
            so_ntt_inverse(c0 + i * n, c->tables[i], 0);
            so_ntt_inverse(c1 + i * n, c->tables[i], 0);
        }
        so_add_poly_coeffmod(c0 + i * n, m, n, &c->mod[i], c0 + i * n);
    }
    free(m);
    free(limbs);
}

void so_decrypt_phase(const so_context *c, size_t L, const uint64_t *sk, const uint64_t *ct, size_t size, int ntt_form,
                      uint64_t *phase)
{
    size_t n = c->n;
    uint64_t *t = (uint64_t *)malloc(sizeof(uint64_t) * n), *sp = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (size_t i = 0; i < L; i++)
    {
        const so_modulus *m = &c->mod[i];
        uint64_t *ph = phase + i * n;
        memset(ph, 0, sizeof(uint64_t) * n);
        for (size_t l = 0; l < n; l++)
            sp[l] = 1;
        for (size_t p = 0; p < size; p++)
        {
            memcpy(t, ct + (p * L + i) * n, sizeof(uint64_t) * n);
            if (!ntt_form)
                so_ntt_forward(t, c->tables[i], 0);
            for (size_t l = 0; l < n; l++)
            {
                ph[l] = so_add_uint_mod(ph[l], so_multiply_uint_mod(t[l], sp[l], m), m);
                sp[l] = so_multiply_uint_mod(sp[l], sk[i * n + l], m);
            }
        }
        so_ntt_inverse(ph, c->tables[i], 0);
    }
    free(t);
    free(sp);
}

/* ------------------------------------------------------------------------------------------------
 * the reference's own randomness -- Blake2xb PRNG (randomgen.cpp:176-211, randomgen.h:300-560),
 * RLWE samplers (util/rlwe.cpp:21-131), key generation (keygenerator.cpp:56-363), encrypt_zero_symmetric /
 * _asymmetric (util/rlwe.cpp:189-412), Encryptor::encrypt_zero_internal (encryptor.cpp:84-180) and
 * Decryptor::dot_product_ct_sk_array (decryptor.cu:544-626).  A "factory seed" plays the role of a
 * Blake2xbPRNGFactory(default_seed): every create() starts a generator from that same seed (randomgen.h:436-447).
 * BLAKE2b follows RFC 7693, BLAKE2Xb the reference's util/blake2xb.c:32-149; tests/test_random_keys.py checks both
 * against the reference's C files compiled as they lie (oracle/_ref/libblake2_ref.so).
 * ---------------------------------------------------------------------------------------------- */

static const uint64_t b2_iv[8] = { 0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                   0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL };
static const uint8_t b2_sigma[10][16] = {
    { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 }, { 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3 },
    { 11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4 }, { 7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8 },
    { 9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13 }, { 2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9 },
    { 12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11 }, { 13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10 },
    { 6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5 }, { 10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0 }
};

static inline uint64_t b2_rotr(uint64_t x, int c)
{
    return (x >> c) | (x << (64 - c));
}

/* RFC 7693 3.2: compression of one 128-byte block; t = bytes hashed so far including this block */
static void b2_compress(uint64_t h[8], const uint8_t block[128], uint64_t t, int last)
{
    uint64_t m[16], v[16];
    memcpy(m, block, 128);
    for (int i = 0; i < 8; i++)
    {
        v[i] = h[i];
        v[i + 8] = b2_iv[i];
    }
    v[12] ^= t; /* inputs stay far below 2^64 bytes: the high counter word is zero */
    if (last)
        v[14] = ~v[14];
#define B2_G(a, b, c, d, x, y)            \
    v[a] = v[a] + v[b] + (x);             \
    v[d] = b2_rotr(v[d] ^ v[a], 32);      \
    v[c] = v[c] + v[d];                   \
    v[b] = b2_rotr(v[b] ^ v[c], 24);      \
    v[a] = v[a] + v[b] + (y);             \
    v[d] = b2_rotr(v[d] ^ v[a], 16);      \
    v[c] = v[c] + v[d];                   \
    v[b] = b2_rotr(v[b] ^ v[c], 63);
    for (int r = 0; r < 12; r++)
    {
        const uint8_t *s = b2_sigma[r % 10];
        B2_G(0, 4, 8, 12, m[s[0]], m[s[1]])
        B2_G(1, 5, 9, 13, m[s[2]], m[s[3]])
        B2_G(2, 6, 10, 14, m[s[4]], m[s[5]])
        B2_G(3, 7, 11, 15, m[s[6]], m[s[7]])
        B2_G(0, 5, 10, 15, m[s[8]], m[s[9]])
        B2_G(1, 6, 11, 12, m[s[10]], m[s[11]])
        B2_G(2, 7, 8, 13, m[s[12]], m[s[13]])
        B2_G(3, 4, 9, 14, m[s[14]], m[s[15]])
    }
#undef B2_G
    for (int i = 0; i < 8; i++)
        h[i] ^= v[i] ^ v[i + 8];
}

/* hash of a whole message given as up to two pieces (an optional 128-byte key block, then the input) under a
 * 64-byte parameter block */
static void b2_hash(const uint8_t param[64], const uint8_t *key_block, const uint8_t *in, size_t inlen, uint8_t *out,
                    size_t outlen)
{
    uint64_t h[8];
    for (int i = 0; i < 8; i++)
    {
        uint64_t w;
        memcpy(&w, param + 8 * i, 8);
        h[i] = b2_iv[i] ^ w;
    }
    size_t total = (key_block ? 128 : 0) + inlen;
    size_t blocks = total ? (total + 127) / 128 : 1;
    for (size_t b = 0; b < blocks; b++)
    {
        uint8_t block[128];
        memset(block, 0, 128);
        size_t start = b * 128, end = start + 128 < total ? start + 128 : total;
        for (size_t pos = start; pos < end; pos++)
            block[pos - start] = (key_block && pos < 128) ? key_block[pos] : in[pos - (key_block ? 128 : 0)];
        b2_compress(h, block, end, b + 1 == blocks);
    }
    uint8_t full[64];
    memcpy(full, h, 64);
    memcpy(out, full, outlen);
}

int so_blake2b(void *out, size_t outlen, const void *in, size_t inlen, const void *key, size_t keylen)
{
    if (!out || !outlen || outlen > 64 || keylen > 64 || (!in && inlen) || (!key && keylen))
        return -1;
    uint8_t param[64] = { 0 }, kb[128] = { 0 };
    param[0] = (uint8_t)outlen;
    param[1] = (uint8_t)keylen;
    param[2] = 1;
    param[3] = 1;
    if (keylen)
        memcpy(kb, key, keylen);
    b2_hash(param, keylen ? kb : NULL, (const uint8_t *)in, inlen, (uint8_t *)out, outlen);
    return 0;
}

/* util/blake2xb.c:36-181 */
int so_blake2xb(void *out, size_t outlen, const void *in, size_t inlen, const void *key, size_t keylen)
{
    if (!out || !outlen || outlen > 0xFFFFFFFFUL || keylen > 64 || (!in && inlen) || (!key && keylen))
        return -1;
    uint8_t param[64] = { 0 }, kb[128] = { 0 }, root[64];
    uint32_t xof = (uint32_t)outlen, leaf = 64;
    param[0] = 64;
    param[1] = (uint8_t)keylen;
    param[2] = 1;
    param[3] = 1;
    memcpy(param + 12, &xof, 4);
    if (keylen)
        memcpy(kb, key, keylen);
    b2_hash(param, keylen ? kb : NULL, (const uint8_t *)in, inlen, root, 64);
    param[1] = param[2] = param[3] = 0; /* :119-125 */
    memcpy(param + 4, &leaf, 4);
    param[16] = 0;
    param[17] = 64;
    for (uint32_t i = 0; outlen > 0; i++)
    {
        size_t bs = outlen < 64 ? outlen : 64;
        param[0] = (uint8_t)bs;
        memcpy(param + 8, &i, 4);
        b2_hash(param, NULL, root, 64, (uint8_t *)out + (size_t)i * 64, bs);
        outlen -= bs;
    }
    return 0;
}

#define SO_PRNG_BUFFER 4096 /* randomgen.h:391 */
struct so_prng
{
    uint64_t seed[8];
    uint64_t counter;
    size_t head;
    uint8_t buffer[SO_PRNG_BUFFER];
};

so_prng *so_prng_create(const uint64_t seed[8])
{
    so_prng *p = (so_prng *)calloc(1, sizeof(*p));
    memcpy(p->seed, seed, 64);
    p->head = SO_PRNG_BUFFER; /* empty: the first draw refills (randomgen.h:331-336) */
    return p;
}

void so_prng_destroy(so_prng *p)
{
    free(p);
}

/* randomgen.cpp:176-211 */
void so_prng_generate(so_prng *p, size_t count, void *dest)
{
    uint8_t *d = (uint8_t *)dest;
    while (count)
    {
        size_t cur = SO_PRNG_BUFFER - p->head;
        if (cur > count)
            cur = count;
        memcpy(d, p->buffer + p->head, cur);
        p->head += cur;
        d += cur;
        count -= cur;
        if (p->head == SO_PRNG_BUFFER)
        {
            so_blake2xb(p->buffer, SO_PRNG_BUFFER, &p->counter, sizeof(p->counter), p->seed, 64);
            p->counter++;
            p->head = 0;
        }
    }
}

static uint32_t prng_u32(so_prng *p)
{
    uint32_t r;
    so_prng_generate(p, 4, &r);
    return r;
}

/* std::uniform_int_distribution<uint64_t>(0, 2) over the 32-bit adapter (randomtostd.h) as libstdc++ 11 computes
 * it: Lemire's method, product = word * 3, redraw while the low half is below 2^32 mod 3 = 1 */
static uint64_t prng_uniform3(so_prng *p)
{
    uint64_t product = (uint64_t)prng_u32(p) * 3u;
    uint32_t low = (uint32_t)product;
    if (low < 3u)
    {
        uint32_t threshold = (uint32_t)(0u - 3u) % 3u;
        while (low < threshold)
        {
            product = (uint64_t)prng_u32(p) * 3u;
            low = (uint32_t)product;
        }
    }
    return product >> 32;
}

/* util/rlwe.cpp:21-38 */
void so_sample_poly_ternary(so_prng *p, const uint64_t *q, size_t limbs, size_t n, uint64_t *dest)
{
    for (size_t i = 0; i < n; i++)
    {
        uint64_t rand = prng_uniform3(p);
        uint64_t flag = (uint64_t)(-(int64_t)(rand == 0));
        for (size_t j = 0; j < limbs; j++)
            dest[i + j * n] = rand + (flag & q[j]) - 1;
    }
}

/* util/rlwe.cpp:66-101 */
void so_sample_poly_cbd(so_prng *p, const uint64_t *q, size_t limbs, size_t n, uint64_t *dest)
{
    for (size_t i = 0; i < n; i++)
    {
        unsigned char x[6];
        so_prng_generate(p, 6, x);
        x[2] &= 0x1F;
        x[5] &= 0x1F;
        int32_t noise = __builtin_popcount(x[0]) + __builtin_popcount(x[1]) + __builtin_popcount(x[2]) -
                        __builtin_popcount(x[3]) - __builtin_popcount(x[4]) - __builtin_popcount(x[5]);
        uint64_t flag = (uint64_t)(-(int64_t)(noise < 0));
        for (size_t j = 0; j < limbs; j++)
            dest[i + j * n] = (uint64_t)(int64_t)noise + (flag & q[j]);
    }
}

/* util/rlwe.cpp:103-131 */
void so_sample_poly_uniform(so_prng *p, const uint64_t *q, size_t limbs, size_t n, uint64_t *dest)
{
    so_prng_generate(p, limbs * n * sizeof(uint64_t), dest);
    for (size_t j = 0; j < limbs; j++)
    {
        so_modulus m;
        so_modulus_init(&m, q[j]);
        uint64_t max_random = 0xFFFFFFFFFFFFFFFFULL;
        uint64_t max_multiple = max_random - so_barrett_reduce_64(max_random, &m) - 1;
        for (size_t i = 0; i < n; i++)
        {
            uint64_t rand = dest[i + j * n];
            while (rand >= max_multiple)
                so_prng_generate(p, sizeof(rand), &rand);
            dest[i + j * n] = so_barrett_reduce_64(rand, &m);
        }
    }
}

/* keygenerator.cpp:56-91: sk over all k key primes, NTT form */
void so_seal_gen_secret_key(const so_context *c, const uint64_t factory_seed[8], uint64_t *sk)
{
    so_prng *p = so_prng_create(factory_seed);
    so_sample_poly_ternary(p, c->primes, c->k, c->n, sk);
    so_prng_destroy(p);
    for (size_t i = 0; i < c->k; i++)
        so_ntt_forward(sk + i * c->n, c->tables[i], 0);
}

/* UniformRandomGeneratorInfo::save, compr_mode none (randomgen.cpp:99-121 inside serialization.cpp:224-265): a
 * 16-byte SEALHeader {magic 0xA15E, header size 0x10, version 4.0, compr_mode 0, reserved, size = 81}, the prng_type
 * byte (1 = blake2xb), the 64-byte seed */
#define SO_PRNG_INFO_BYTES 81
static void prng_info_save(const uint64_t seed[8], uint8_t out[SO_PRNG_INFO_BYTES])
{
    const uint8_t head[8] = { 0x5E, 0xA1, 0x10, 4, 0, 0, 0, 0 };
    uint64_t size = SO_PRNG_INFO_BYTES;
    memcpy(out, head, 8);
    memcpy(out + 8, &size, 8);
    out[16] = 1;
    memcpy(out + 17, seed, 64);
}

/* util/rlwe.cpp:282-412: ct = 2 polys x L limbs; with save_seed the second polynomial ends as the indicator word
 * 0xFFFFFFFFFFFFFFFF followed by the serialized generator info (:404-411) */
static void seal_encrypt_zero_symmetric(const so_context *c, size_t L, const uint64_t *sk, const uint64_t factory_seed[8],
                                        int ntt_form, int save_seed, uint64_t *ct);

void so_seal_encrypt_zero_symmetric(const so_context *c, size_t L, const uint64_t *sk, const uint64_t factory_seed[8],
                                    int ntt_form, uint64_t *ct)
{
    seal_encrypt_zero_symmetric(c, L, sk, factory_seed, ntt_form, 0, ct);
}

void so_seal_encrypt_zero_symmetric_seeded(const so_context *c, size_t L, const uint64_t *sk,
                                           const uint64_t factory_seed[8], int ntt_form, uint64_t *ct)
{
    seal_encrypt_zero_symmetric(c, L, sk, factory_seed, ntt_form, 1, ct);
}

/* Ciphertext::expand_seed (ciphertext.cpp:153-186, version 4): the second polynomial of a seeded ciphertext */
void so_seal_expand_seed(const so_context *c, size_t L, const uint64_t seed[8], uint64_t *c1)
{
    so_prng *p = so_prng_create(seed);
    so_sample_poly_uniform(p, c->primes, L, c->n, c1);
    so_prng_destroy(p);
}

static void seal_encrypt_zero_symmetric(const so_context *c, size_t L, const uint64_t *sk, const uint64_t factory_seed[8],
                                        int ntt_form, int save_seed, uint64_t *ct)
{
    size_t n = c->n;
    uint64_t *c0 = ct, *c1 = ct + L * n;
    so_prng *bootstrap = so_prng_create(factory_seed);
    uint64_t public_seed[8];
    so_prng_generate(bootstrap, 64, public_seed);
    so_prng *cprng = so_prng_create(public_seed);
    size_t info_words = (SO_PRNG_INFO_BYTES + 7) / 8;
    if (save_seed && L * n < info_words + 1) /* :305-317 */
        save_seed = 0;
    so_sample_poly_uniform(cprng, c->primes, L, n, c1);
    if (!ntt_form && save_seed) /* :347-356 */
        for (size_t i = 0; i < L; i++)
            so_ntt_forward(c1 + i * n, c->tables[i], 0);
    uint64_t *noise = (uint64_t *)malloc(sizeof(uint64_t) * L * n);
    so_sample_poly_cbd(bootstrap, c->primes, L, n, noise);
    for (size_t i = 0; i < L; i++)
    {
        const so_modulus *m = &c->mod[i];
        so_dyadic_product_coeffmod(sk + i * n, c1 + i * n, n, m, c0 + i * n);
        if (ntt_form)
            so_ntt_forward(noise + i * n, c->tables[i], 0);
        else
            so_ntt_inverse(c0 + i * n, c->tables[i], 0);
        if (c->scheme == SO_SCHEME_BGV)
            so_multiply_poly_scalar_coeffmod(noise + i * n, n, c->plain_modulus, m, noise + i * n);
        so_add_poly_coeffmod(noise + i * n, c0 + i * n, n, m, c0 + i * n);
        so_negate_poly_coeffmod(c0 + i * n, n, m, c0 + i * n);
    }
    if (!ntt_form && !save_seed)
        for (size_t i = 0; i < L; i++)
            so_ntt_inverse(c1 + i * n, c->tables[i], 0);
    if (save_seed)
    {
        uint8_t info[SO_PRNG_INFO_BYTES];
        prng_info_save(public_seed, info);
        c1[0] = 0xFFFFFFFFFFFFFFFFULL;
        memcpy(c1 + 1, info, SO_PRNG_INFO_BYTES);
    }
    free(noise);
    so_prng_destroy(cprng);
    so_prng_destroy(bootstrap);
}

/* util/rlwe.cpp:189-280 at the level with L limbs; pk = 2 polys x k limbs (key level, NTT form) */
void so_seal_encrypt_zero_asymmetric(const so_context *c, size_t L, const uint64_t *pk, const uint64_t factory_seed[8],
                                     int ntt_form, uint64_t *ct)
{
    size_t n = c->n, k = c->k;
    so_prng *p = so_prng_create(factory_seed);
    uint64_t *u = (uint64_t *)malloc(sizeof(uint64_t) * L * n);
    so_sample_poly_ternary(p, c->primes, L, n, u);
    for (size_t i = 0; i < L; i++)
    {
        so_ntt_forward(u + i * n, c->tables[i], 0);
        for (size_t j = 0; j < 2; j++)
        {
            uint64_t *d = ct + (j * L + i) * n;
            so_dyadic_product_coeffmod(u + i * n, pk + (j * k + i) * n, n, &c->mod[i], d);
            if (!ntt_form)
                so_ntt_inverse(d, c->tables[i], 0);
        }
    }
    for (size_t j = 0; j < 2; j++)
    {
        so_sample_poly_cbd(p, c->primes, L, n, u);
        for (size_t i = 0; i < L; i++)
        {
            uint64_t *e = u + i * n, *d = ct + (j * L + i) * n;
            if (c->scheme == SO_SCHEME_BGV)
            {
                if (ntt_form)
                    so_ntt_forward(e, c->tables[i], 1);
                so_multiply_poly_scalar_coeffmod(e, n, c->plain_modulus, &c->mod[i], e);
            }
            else if (ntt_form)
                so_ntt_forward(e, c->tables[i], 0);
            so_add_poly_coeffmod(e, d, n, &c->mod[i], d);
        }
    }
    free(u);
    so_prng_destroy(p);
}

/* Encryptor::encrypt_zero_internal (encryptor.cpp:84-180): below the key level a public-key encryption is made one
 * level up and switched down */
void so_seal_encrypt_zero(so_context *c, size_t L, const uint64_t *pk, const uint64_t *sk, const uint64_t factory_seed[8],
                          uint64_t *ct)
{
    int ntt_form = c->scheme != SO_SCHEME_BFV;
    if (!pk)
    {
        so_seal_encrypt_zero_symmetric(c, L, sk, factory_seed, ntt_form, ct);
        return;
    }
    if (L == c->k)
    {
        so_seal_encrypt_zero_asymmetric(c, L, pk, factory_seed, ntt_form, ct);
        return;
    }
    uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * 2 * (L + 1) * c->n);
    so_seal_encrypt_zero_asymmetric(c, L + 1, pk, factory_seed, ntt_form, tmp);
    so_mod_switch_scale_to_next(c, L + 1, tmp, 2, ct);
    free(tmp);
}

/* keygenerator.cpp:301-334 with the reference's generators; key layout [k-1][2][k][N] */
void so_seal_gen_kswitch_key(const so_context *c, const uint64_t *sk, const uint64_t *new_key,
                             const uint64_t factory_seed[8], uint64_t *key)
{
    size_t n = c->n, k = c->k, d = k - 1;
    uint64_t *t = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (size_t J = 0; J < d; J++)
    {
        uint64_t *c0 = key + (J * 2) * k * n;
        so_seal_encrypt_zero_symmetric(c, k, sk, factory_seed, 1, c0);
        const so_modulus *m = &c->mod[J];
        uint64_t factor = so_barrett_reduce_64(c->mod[k - 1].value, m);
        so_multiply_poly_scalar_coeffmod(new_key + J * n, n, factor, m, t);
        so_add_poly_coeffmod(c0 + J * n, t, n, m, c0 + J * n);
    }
    free(t);
}

void so_seal_gen_relin_key(const so_context *c, const uint64_t *sk, const uint64_t factory_seed[8], uint64_t *key)
{
    size_t n = c->n, k = c->k;
    uint64_t *s2 = (uint64_t *)malloc(sizeof(uint64_t) * k * n);
    for (size_t i = 0; i < k; i++)
        so_dyadic_product_coeffmod(sk + i * n, sk + i * n, n, &c->mod[i], s2 + i * n);
    so_seal_gen_kswitch_key(c, sk, s2, factory_seed, key);
    free(s2);
}

void so_seal_gen_galois_key(const so_context *c, const uint64_t *sk, uint32_t elt, const uint64_t factory_seed[8],
                            uint64_t *key)
{
    size_t n = c->n, k = c->k;
    uint64_t *rs = (uint64_t *)malloc(sizeof(uint64_t) * k * n);
    for (size_t i = 0; i < k; i++)
        so_apply_galois_ntt(sk + i * n, c->log_n, elt, rs + i * n);
    so_seal_gen_kswitch_key(c, sk, rs, factory_seed, key);
    free(rs);
}

/* decryptor.cu:544-626: out (L limbs) = c0 + c1 s + ... in the form of the ciphertext */
void so_seal_dot_product_ct_sk(const so_context *c, size_t L, const uint64_t *sk, const uint64_t *ct, size_t size,
                               int ntt_form, uint64_t *out)
{
    size_t n = c->n;
    uint64_t *t = (uint64_t *)malloc(sizeof(uint64_t) * n), *sp = (uint64_t *)malloc(sizeof(uint64_t) * n),
             *acc = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (size_t i = 0; i < L; i++)
    {
        const so_modulus *m = &c->mod[i];
        memset(acc, 0, sizeof(uint64_t) * n);
        memcpy(sp, sk + i * n, sizeof(uint64_t) * n);
        for (size_t p = 1; p < size; p++)
        {
            memcpy(t, ct + (p * L + i) * n, sizeof(uint64_t) * n);
            if (!ntt_form)
                so_ntt_forward(t, c->tables[i], 1);
            so_dyadic_product_coeffmod(t, sp, n, m, t);
            so_add_poly_coeffmod(acc, t, n, m, acc);
            so_dyadic_product_coeffmod(sp, sk + i * n, n, m, sp);
        }
        if (!ntt_form)
            so_ntt_inverse(acc, c->tables[i], 0);
        so_add_poly_coeffmod(acc, ct + i * n, n, m, out + i * n);
    }
    free(t);
    free(sp);
    free(acc);
}

/* evaluator.cpp:56-118 balance_correction_factors: out = {new factor, e1, e2} with e1 f1 = e2 f2 (mod t) and
 * |e1| + |e2| smallest among the remainders of the extended Euclidean run on (t, f2 / f1) */
static uint64_t gcd_u64(uint64_t a, uint64_t b)
{
    while (b)
    {
        uint64_t r = a % b;
        a = b;
        b = r;
    }
    return a;
}

int so_balance_correction_factors(uint64_t factor1, uint64_t factor2, uint64_t t, uint64_t out[3])
{
    so_modulus pm;
    so_modulus_init(&pm, t);
    uint64_t half_t = t / 2, ratio = 1;
    if (!so_try_invert_uint_mod(factor1, t, &ratio))
        return -1;
    ratio = so_multiply_uint_mod(ratio, factor2, &pm);
#define SO_BAL(x) ((int64_t)((x) > half_t ? (x)-t : (x)))
#define SO_SUM_ABS(x, y) (llabs(SO_BAL(x)) + llabs(SO_BAL(y)))
    uint64_t e1 = ratio, e2 = 1;
    int64_t sum = SO_SUM_ABS(e1, e2);
    int64_t prev_a = (int64_t)t, prev_b = 0, a = (int64_t)ratio, b = 1;
    while (a != 0)
    {
        int64_t q = prev_a / a, temp = prev_a % a;
        prev_a = a;
        a = temp;
        temp = prev_b - b * q;
        prev_b = b;
        b = temp;
        uint64_t a_mod = so_barrett_reduce_64((uint64_t)llabs(a), &pm);
        if (a < 0)
            a_mod = so_negate_uint_mod(a_mod, &pm);
        uint64_t b_mod = so_barrett_reduce_64((uint64_t)llabs(b), &pm);
        if (b < 0)
            b_mod = so_negate_uint_mod(b_mod, &pm);
        if (a_mod != 0 && gcd_u64(a_mod, t) == 1)
        {
            int64_t new_sum = SO_SUM_ABS(a_mod, b_mod);
            if (new_sum < sum)
            {
                sum = new_sum;
                e1 = a_mod;
                e2 = b_mod;
            }
        }
    }
#undef SO_BAL
#undef SO_SUM_ABS
    out[0] = so_multiply_uint_mod(e1, factor1, &pm);
    out[1] = e1;
    out[2] = e2;
    return 0;
}
