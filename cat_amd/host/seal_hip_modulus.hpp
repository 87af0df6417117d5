// seal::Modulus, seal::CoeffModulus, seal::PlainModulus and sec_level_type for the facade (host only).
// References (paths relative to /root/reference/native/src/seal/): modulus.h / modulus.cpp:66-230 (Modulus::set_value,
// const_ratio, CoeffModulus::MaxBitCount / BFVDefault / Create, PlainModulus::Batching), util/numth.cpp:245-311
// (is_prime, get_primes), util/hestdparms.h (classical security tables), util/globals.cpp:24-70 (128-bit defaults).
#pragma once
#include <algorithm>
#include <array>
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <unordered_map>
#include <vector>

namespace seal
{
    enum class sec_level_type : int
    {
        none = 0,
        tc128 = 128,
        tc192 = 192,
        tc256 = 256
    };

    namespace util
    {
        inline std::uint64_t mulmod_u64(std::uint64_t a, std::uint64_t b, std::uint64_t m)
        {
            return (std::uint64_t)((unsigned __int128)a * b % m);
        }
        inline std::uint64_t powmod_u64(std::uint64_t a, std::uint64_t e, std::uint64_t m)
        {
            std::uint64_t r = 1 % m;
            a %= m;
            for (; e; e >>= 1, a = mulmod_u64(a, a, m))
                if (e & 1)
                    r = mulmod_u64(r, a, m);
            return r;
        }
        // deterministic Miller-Rabin for 64-bit values (the reference's is_prime is probabilistic, numth.cpp:245-276;
        // on primes and composites below 2^64 the two agree)
        inline bool is_prime_u64(std::uint64_t n)
        {
            if (n < 2)
                return false;
            for (std::uint64_t p : { 2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull })
                if (n % p == 0)
                    return n == p;
            std::uint64_t d = n - 1;
            int r = 0;
            while (!(d & 1))
                d >>= 1, ++r;
            for (std::uint64_t a : { 2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull })
            {
                std::uint64_t x = powmod_u64(a, d, n);
                if (x == 1 || x == n - 1)
                    continue;
                bool composite = true;
                for (int i = 1; i < r && composite; ++i)
                {
                    x = mulmod_u64(x, x, n);
                    composite = x != n - 1;
                }
                if (composite)
                    return false;
            }
            return true;
        }
        inline int significant_bits(std::uint64_t v)
        {
            int b = 0;
            for (; v; v >>= 1)
                ++b;
            return b;
        }
    } // namespace util

    // modulus.h: value, bit_count, const_ratio = {floor(2^128 / value) low, high, 2^128 mod value}, is_prime
    class Modulus
    {
    public:
        Modulus(std::uint64_t value = 0)
        {
            set_value(value);
        }
        Modulus &operator=(std::uint64_t value)
        {
            set_value(value);
            return *this;
        }
        std::uint64_t value() const noexcept
        {
            return value_;
        }
        const std::uint64_t *data() const noexcept
        {
            return &value_;
        }
        int bit_count() const noexcept
        {
            return bit_count_;
        }
        std::size_t uint64_count() const noexcept
        {
            return 1; // also for the zero modulus (modulus.h: uint64_count_ = 1; tests/seal/modulus.cpp:21)
        }
        const std::array<std::uint64_t, 3> &const_ratio() const noexcept
        {
            return const_ratio_;
        }
        bool is_zero() const noexcept
        {
            return value_ == 0;
        }
        bool is_prime() const noexcept
        {
            return is_prime_;
        }
        // modulus.cpp:107-114
        std::uint64_t reduce(std::uint64_t value) const
        {
            if (value_ == 0)
                throw std::logic_error("cannot reduce modulo a zero modulus");
            return value % value_;
        }
        // the facade's own code reads a modulus as its value (not in the reference, which spells .value())
        operator std::uint64_t() const noexcept
        {
            return value_;
        }
        bool operator==(const Modulus &o) const noexcept { return value_ == o.value_; }
        bool operator!=(const Modulus &o) const noexcept { return value_ != o.value_; }
        bool operator<(const Modulus &o) const noexcept { return value_ < o.value_; }
        bool operator<=(const Modulus &o) const noexcept { return value_ <= o.value_; }
        bool operator>(const Modulus &o) const noexcept { return value_ > o.value_; }
        bool operator>=(const Modulus &o) const noexcept { return value_ >= o.value_; }

    private:
        // modulus.cpp:66-105
        void set_value(std::uint64_t value)
        {
            if (value == 0)
            {
                value_ = 0;
                bit_count_ = 0;
                const_ratio_ = { { 0, 0, 0 } };
                is_prime_ = false;
                return;
            }
            if ((value >> 61) != 0 || value == 1) // SEAL_MOD_BIT_COUNT_MAX = 61
                throw std::invalid_argument("value can be at most 61-bit and cannot be 1");
            value_ = value;
            bit_count_ = util::significant_bits(value);
            // floor(2^128 / value) by two steps of 128-by-64 division
            const unsigned __int128 top = ((unsigned __int128)1 << 64);
            const std::uint64_t hi = (std::uint64_t)(top / value);
            const unsigned __int128 rem = ((top % value) << 64);
            const std::uint64_t lo = (std::uint64_t)(rem / value);
            const_ratio_ = { { lo, hi, (std::uint64_t)(rem % value) } };
            is_prime_ = util::is_prime_u64(value);
        }
        std::uint64_t value_ = 0;
        int bit_count_ = 0;
        std::array<std::uint64_t, 3> const_ratio_{ { 0, 0, 0 } };
        bool is_prime_ = false;
    };

    namespace util
    {
        // numth.cpp:278-311: primes = 1 mod factor of exactly bit_size bits, largest first
        inline std::vector<Modulus> get_primes(std::uint64_t factor, int bit_size, std::size_t count)
        {
            if (bit_size > 61 || bit_size < 2)
                throw std::invalid_argument("failed to find enough qualifying primes");
            std::vector<Modulus> out;
            std::uint64_t value = ((std::uint64_t(1) << bit_size) - 1) / factor * factor + 1;
            const std::uint64_t lower = std::uint64_t(1) << (bit_size - 1);
            while (count > 0 && value > lower)
            {
                if (is_prime_u64(value))
                {
                    out.emplace_back(value);
                    --count;
                }
                if (value < factor)
                    break;
                value -= factor;
            }
            if (count > 0)
                throw std::logic_error("failed to find enough qualifying primes");
            return out;
        }
    } // namespace util

    class CoeffModulus
    {
    public:
        CoeffModulus() = delete;
        // util/hestdparms.h (classical tables)
        static int MaxBitCount(std::size_t poly_modulus_degree, sec_level_type sec_level = sec_level_type::tc128)
        {
            static const std::size_t degrees[] = { 1024, 2048, 4096, 8192, 16384, 32768 };
            static const int t128[] = { 27, 54, 109, 218, 438, 881 }, t192[] = { 19, 37, 75, 152, 305, 611 },
                             t256[] = { 14, 29, 58, 118, 237, 476 };
            if (sec_level == sec_level_type::none)
                return 2147483647;
            for (int i = 0; i < 6; ++i)
                if (degrees[i] == poly_modulus_degree)
                    return sec_level == sec_level_type::tc128 ? t128[i] : sec_level == sec_level_type::tc192 ? t192[i] : t256[i];
            return 0;
        }
        // modulus.cpp:116-142 with the 128-bit table of util/globals.cpp:24-70 (the 192- and 256-bit tables are not carried)
        static std::vector<Modulus> BFVDefault(std::size_t poly_modulus_degree, sec_level_type sec_level = sec_level_type::tc128)
        {
            if (!MaxBitCount(poly_modulus_degree, sec_level))
                throw std::invalid_argument("non-standard poly_modulus_degree");
            if (sec_level == sec_level_type::none)
                throw std::invalid_argument("invalid security level");
            if (sec_level != sec_level_type::tc128)
                throw std::logic_error("only the 128-bit default coefficient moduli are available");
            switch (poly_modulus_degree)
            {
            case 1024: return { 0x7e00001 };
            case 2048: return { 0x3fffffff000001 };
            case 4096: return { 0xffffee001, 0xffffc4001, 0x1ffffe0001 };
            case 8192: return { 0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001 };
            case 16384:
                return { 0xfffffffd8001,  0xfffffffa0001,  0xfffffff00001,  0x1fffffff68001, 0x1fffffff50001,
                         0x1ffffffee8001, 0x1ffffffea0001, 0x1ffffffe88001, 0x1ffffffe48001 };
            default:
                return { 0x7fffffffe90001, 0x7fffffffbf0001, 0x7fffffffbd0001, 0x7fffffffba0001, 0x7fffffffaa0001,
                         0x7fffffffa50001, 0x7fffffff9f0001, 0x7fffffff7e0001, 0x7fffffff770001, 0x7fffffff380001,
                         0x7fffffff330001, 0x7fffffff2d0001, 0x7fffffff170001, 0x7fffffff150001, 0x7ffffffef00001,
                         0xfffffffff70001 };
            }
        }
        // modulus.cpp:144-190: equal bit sizes are handed out smallest prime first
        static std::vector<Modulus> Create(std::size_t poly_modulus_degree, std::vector<int> bit_sizes)
        {
            return create(poly_modulus_degree, 2 * (std::uint64_t)poly_modulus_degree, bit_sizes);
        }
        // modulus.cpp:192-229: primes that are also 1 modulo the plain modulus (BGV)
        static std::vector<Modulus> Create(std::size_t poly_modulus_degree, const Modulus &plain_modulus, std::vector<int> bit_sizes)
        {
            std::uint64_t factor = 2 * (std::uint64_t)poly_modulus_degree, a = plain_modulus.value(), b = factor;
            while (b)
            {
                const std::uint64_t r = a % b;
                a = b;
                b = r;
            }
            const unsigned __int128 f = (unsigned __int128)factor * (plain_modulus.value() / a);
            if (f >> 64)
                throw std::logic_error("unsigned overflow");
            return create(poly_modulus_degree, (std::uint64_t)f, bit_sizes);
        }

    private:
        static std::vector<Modulus> create(std::size_t n, std::uint64_t factor, const std::vector<int> &bit_sizes)
        {
            if (n > 131072 || n < 2 || (n & (n - 1)))
                throw std::invalid_argument("poly_modulus_degree is invalid");
            if (bit_sizes.size() > 256)
                throw std::invalid_argument("bit_sizes is invalid");
            for (int b : bit_sizes)
                if (b > 60 || b < 2) // SEAL_USER_MOD_BIT_COUNT_MAX / _MIN
                    throw std::invalid_argument("bit_sizes is invalid");
            std::unordered_map<int, std::size_t> count;
            for (int b : bit_sizes)
                ++count[b];
            std::unordered_map<int, std::vector<Modulus>> table;
            for (const auto &e : count)
                table[e.first] = util::get_primes(factor, e.first, e.second);
            std::vector<Modulus> result;
            for (int b : bit_sizes)
            {
                result.emplace_back(table[b].back());
                table[b].pop_back();
            }
            return result;
        }
    };

    class PlainModulus
    {
    public:
        PlainModulus() = delete;
        // modulus.h: a prime of the given size that supports batching (= 1 modulo 2N)
        static Modulus Batching(std::size_t poly_modulus_degree, int bit_size)
        {
            return CoeffModulus::Create(poly_modulus_degree, { bit_size })[0];
        }
        static std::vector<Modulus> Batching(std::size_t poly_modulus_degree, std::vector<int> bit_sizes)
        {
            return CoeffModulus::Create(poly_modulus_degree, bit_sizes);
        }
    };
} // namespace seal
