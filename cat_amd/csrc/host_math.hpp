// Host-side number theory and table construction for the context (product code; independent of
// oracle/).  Produces exactly the constants SEAL's SEALContext would hold for this path:
// Barrett ratios (modulus.cpp:87-98), minimal primitive roots (util/numth.cpp:386-412), NTT tables
// in SEAL's orders (util/ntt.cpp:268-287), Shoup quotients (util/uintarithsmallmod.cuh:368-380).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace cat
{
    using u64 = unsigned long long; // same type as the device-side ::u64 (modarith.cuh)
    using u128 = unsigned __int128;

    struct HostMod
    {
        u64 q = 0;
        u64 ratio0 = 0, ratio1 = 0;

        HostMod() = default;
        explicit HostMod(u64 value) : q(value)
        {
            // floor(2^128 / q) from floor((2^128 - 1) / q)
            u128 ones = ~u128(0);
            u128 quo = ones / q;
            if (ones % q + 1 == q)
                quo += 1;
            ratio0 = u64(quo);
            ratio1 = u64(quo >> 64);
        }
        u64 mul(u64 a, u64 b) const
        {
            return u64((u128(a) * b) % q);
        }
        u64 reduce(u64 a) const
        {
            return a % q;
        }
        u64 pow(u64 a, u64 e) const
        {
            u64 r = 1 % q;
            a %= q;
            while (e)
            {
                if (e & 1)
                    r = mul(r, a);
                a = mul(a, a);
                e >>= 1;
            }
            return r;
        }
        // floor(w * 2^64 / q), w < q
        u64 shoup(u64 w) const
        {
            return u64((u128(w) << 64) / q);
        }
        // split-Shoup constants of w < q (modarith.cuh): out = {W0, W1, Q0 | Q1 << 32}
        void split(u64 w, u64 out[3]) const
        {
            const u64 w1 = u64((u128(w) << 32) % q);
            out[0] = w;
            out[1] = w1;
            out[2] = u64((u128(w) << 32) / q) | (u64((u128(w1) << 32) / q) << 32);
        }
    };

    inline bool invert_mod(u64 a, u64 m, u64 &out)
    {
        if (m < 2)
            return false;
        __int128 r0 = m, r1 = a % m, t0 = 0, t1 = 1;
        while (r1)
        {
            __int128 qq = r0 / r1;
            __int128 r2 = r0 - qq * r1, t2 = t0 - qq * t1;
            r0 = r1, r1 = r2, t0 = t1, t1 = t2;
        }
        if (r0 != 1)
            return false;
        out = u64(t0 < 0 ? t0 + m : t0);
        return true;
    }

    inline bool is_prime_u64(u64 n)
    {
        if (n < 2)
            return false;
        static const u64 small[] = { 2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37 };
        for (u64 p : small)
        {
            if (n == p)
                return true;
            if (n % p == 0)
                return false;
        }
        HostMod m(n);
        u64 d = n - 1;
        int r = 0;
        while (!(d & 1))
            d >>= 1, ++r;
        for (u64 a : small) // deterministic for all 64-bit n
        {
            u64 x = m.pow(a, d);
            if (x == 1 || x == n - 1)
                continue;
            bool composite = true;
            for (int i = 1; i < r && composite; ++i)
            {
                x = m.mul(x, x);
                composite = x != n - 1;
            }
            if (composite)
                return false;
        }
        return true;
    }

    // util::get_primes (numth.cpp:278-311): descending primes = 1 (mod factor) of exactly bit_size bits
    inline std::vector<u64> primes_descending(u64 factor, int bit_size, std::size_t count)
    {
        std::vector<u64> out;
        u64 value = ((u64(1) << bit_size) - 1) / factor * factor + 1;
        u64 lower = u64(1) << (bit_size - 1);
        while (out.size() < count && value > lower)
        {
            if (is_prime_u64(value))
                out.push_back(value);
            value -= factor;
        }
        if (out.size() < count)
            throw std::logic_error("failed to find enough qualifying primes");
        return out;
    }

    // Smallest primitive degree-th root of unity mod q (degree a power of two dividing q-1).
    inline bool minimal_primitive_root(u64 degree, const HostMod &m, u64 &out)
    {
        if ((m.q - 1) % degree)
            return false;
        u64 cofactor = (m.q - 1) / degree, root = 0;
        for (u64 g = 2; g < m.q; ++g)
        {
            u64 c = m.pow(g, cofactor);
            if (m.pow(c, degree >> 1) == m.q - 1)
            {
                root = c;
                break;
            }
            if (g > (1u << 20))
                return false;
        }
        // all primitive roots are the odd powers of `root`
        u64 sq = m.mul(root, root), cur = root, best = root;
        for (u64 i = 0; i < degree; i += 2)
        {
            if (cur < best)
                best = cur;
            cur = m.mul(cur, sq);
        }
        out = best;
        return true;
    }

    inline std::uint32_t bit_reverse(std::uint32_t x, int bits)
    {
        std::uint32_t r = 0;
        for (int i = 0; i < bits; ++i)
            r |= ((x >> i) & 1u) << (bits - 1 - i);
        return r;
    }

    struct HostNttTables
    {
        u64 root = 0, inv_root = 0, inv_n = 0;
        std::vector<u64> fwd; // interleaved {operand, quotient}, 2n words, bit-reversed powers of psi
        std::vector<u64> inv; // interleaved, scrambled powers of psi^-1
        u64 inv_n_w = 0;      // inv[n-1].operand * n^-1 mod q
    };

    inline HostNttTables build_ntt_tables(int log_n, const HostMod &m)
    {
        HostNttTables t;
        std::size_t n = std::size_t(1) << log_n;
        if (!minimal_primitive_root(2 * u64(n), m, t.root) || !invert_mod(t.root, m.q, t.inv_root) ||
            !invert_mod(u64(n) % m.q, m.q, t.inv_n))
            throw std::invalid_argument("invalid modulus");
        t.fwd.assign(2 * n, 0);
        t.inv.assign(2 * n, 0);
        auto put = [&](std::vector<u64> &v, std::size_t idx, u64 w) {
            v[2 * idx] = w;
            v[2 * idx + 1] = m.shoup(w);
        };
        u64 p = t.root;
        for (std::size_t i = 1; i < n; ++i)
        {
            put(t.fwd, bit_reverse(std::uint32_t(i), log_n), p);
            p = m.mul(p, t.root);
        }
        put(t.fwd, 0, 1);
        p = t.inv_root;
        for (std::size_t i = 1; i < n; ++i)
        {
            put(t.inv, bit_reverse(std::uint32_t(i - 1), log_n) + 1, p);
            p = m.mul(p, t.inv_root);
        }
        put(t.inv, 0, 1);
        t.inv_n_w = m.mul(t.inv[2 * (n - 1)], t.inv_n);
        return t;
    }

    // little-endian multi-word helpers for the BEHZ base products (util/rns.cpp:212-257)
    inline std::vector<u64> mp_mul_word(const std::vector<u64> &a, u64 w)
    {
        std::vector<u64> r(a.size() + 1, 0);
        u64 carry = 0;
        for (std::size_t i = 0; i < a.size(); ++i)
        {
            u128 t = u128(a[i]) * w + carry;
            r[i] = u64(t);
            carry = u64(t >> 64);
        }
        r[a.size()] = carry;
        return r;
    }

    inline u64 mp_mod(const std::vector<u64> &a, u64 q)
    {
        u128 r = 0;
        for (std::size_t i = a.size(); i--;)
            r = ((r << 64) | a[i]) % q;
        return u64(r);
    }

    inline int mp_bits(const std::vector<u64> &a)
    {
        for (std::size_t i = a.size(); i--;)
            if (a[i])
                return int(64 * i) + (64 - __builtin_clzll(a[i]));
        return 0;
    }

    // product of all values except index `skip` (skip = SIZE_MAX: full product)
    inline std::vector<u64> mp_product(const std::vector<u64> &vals, std::size_t skip)
    {
        std::vector<u64> r{ 1 };
        for (std::size_t i = 0; i < vals.size(); ++i)
            if (i != skip)
                r = mp_mul_word(r, vals[i]);
        return r;
    }
} // namespace cat
