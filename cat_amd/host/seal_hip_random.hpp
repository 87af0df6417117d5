// seal_hip_random.hpp -- host-side randomness of the facade: the reference's default PRNG
// (Blake2xb in counter mode, randomgen.h:300-560, randomgen.cpp:176-211) and its RLWE samplers
// (util/rlwe.cpp:21-131), so that keys and fresh ciphertexts made here equal the reference's CPU output for the same
// seed.  Pure host C++: the samplers fill host vectors that the callers upload; nothing here touches the device.
//
// BLAKE2b is written from RFC 7693 and BLAKE2Xb from the BLAKE2X specification (the reference vendors the BLAKE2
// authors' C files, util/blake2b.c and util/blake2xb.c; tests/test_random_keys.py checks this file's stream against
// those files compiled as they lie).
#pragma once

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <random>
#include <stdexcept>
#include <vector>

namespace seal
{
    constexpr std::size_t prng_seed_uint64_count = 8; // randomgen.h:20-24
    constexpr std::size_t prng_seed_byte_count = prng_seed_uint64_count * 8;
    using prng_seed_type = std::array<std::uint64_t, prng_seed_uint64_count>;

    enum class prng_type : std::uint8_t // randomgen.h:29-36
    {
        unknown = 0,
        blake2xb = 1,
        shake256 = 2
    };

    namespace util
    {
        // BLAKE2b, RFC 7693.  `param` is the 64-byte parameter block (digest length, key length, fanout, depth,
        // leaf length, node offset, XOF length, node depth, inner length, reserved, salt, personalisation).
        class Blake2b
        {
        public:
            void init(const std::uint8_t param[64])
            {
                static const std::uint64_t iv[8] = { 0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                                                     0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                                                     0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL };
                for (int i = 0; i < 8; ++i)
                {
                    std::uint64_t w;
                    std::memcpy(&w, param + 8 * i, 8);
                    h_[i] = iv[i] ^ w;
                }
                t_[0] = t_[1] = 0;
                buflen_ = 0;
            }
            void update(const void *in, std::size_t len)
            {
                const std::uint8_t *p = static_cast<const std::uint8_t *>(in);
                while (len)
                {
                    if (buflen_ == 128) // more input follows: this block is not the last one
                    {
                        add_counter(128);
                        compress(false);
                        buflen_ = 0;
                    }
                    const std::size_t take = std::min<std::size_t>(128 - buflen_, len);
                    std::memcpy(buf_ + buflen_, p, take);
                    buflen_ += take;
                    p += take;
                    len -= take;
                }
            }
            void final(void *out, std::size_t outlen)
            {
                add_counter(buflen_);
                std::memset(buf_ + buflen_, 0, 128 - buflen_);
                compress(true);
                std::uint8_t full[64];
                std::memcpy(full, h_, 64);
                std::memcpy(out, full, outlen);
            }

        private:
            void add_counter(std::uint64_t inc)
            {
                t_[0] += inc;
                t_[1] += t_[0] < inc;
            }
            static std::uint64_t rotr(std::uint64_t x, int c)
            {
                return (x >> c) | (x << (64 - c));
            }
            void compress(bool last)
            {
                static const std::uint8_t sigma[12][16] = {
                    { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 }, { 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3 },
                    { 11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4 }, { 7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8 },
                    { 9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13 }, { 2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9 },
                    { 12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11 }, { 13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10 },
                    { 6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5 }, { 10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0 },
                    { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 }, { 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3 }
                };
                static const std::uint64_t iv[8] = { 0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                                                     0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                                                     0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL };
                std::uint64_t m[16], v[16];
                std::memcpy(m, buf_, 128);
                for (int i = 0; i < 8; ++i)
                {
                    v[i] = h_[i];
                    v[i + 8] = iv[i];
                }
                v[12] ^= t_[0];
                v[13] ^= t_[1];
                if (last)
                    v[14] = ~v[14];
                auto G = [&](int r, int i, int a, int b, int c, int d) {
                    v[a] = v[a] + v[b] + m[sigma[r][2 * i]];
                    v[d] = rotr(v[d] ^ v[a], 32);
                    v[c] = v[c] + v[d];
                    v[b] = rotr(v[b] ^ v[c], 24);
                    v[a] = v[a] + v[b] + m[sigma[r][2 * i + 1]];
                    v[d] = rotr(v[d] ^ v[a], 16);
                    v[c] = v[c] + v[d];
                    v[b] = rotr(v[b] ^ v[c], 63);
                };
                for (int r = 0; r < 12; ++r)
                {
                    G(r, 0, 0, 4, 8, 12);
                    G(r, 1, 1, 5, 9, 13);
                    G(r, 2, 2, 6, 10, 14);
                    G(r, 3, 3, 7, 11, 15);
                    G(r, 4, 0, 5, 10, 15);
                    G(r, 5, 1, 6, 11, 12);
                    G(r, 6, 2, 7, 8, 13);
                    G(r, 7, 3, 4, 9, 14);
                }
                for (int i = 0; i < 8; ++i)
                    h_[i] ^= v[i] ^ v[i + 8];
            }

            std::uint64_t h_[8], t_[2];
            std::uint8_t buf_[128];
            std::size_t buflen_ = 0;
        };

        // One-shot BLAKE2b with an optional key (the key is the first, zero-padded block)
        inline void blake2b(void *out, std::size_t outlen, const void *in, std::size_t inlen, const void *key,
                            std::size_t keylen)
        {
            if (!outlen || outlen > 64 || keylen > 64)
                throw std::invalid_argument("blake2b: invalid length");
            std::uint8_t param[64] = { 0 };
            param[0] = (std::uint8_t)outlen;
            param[1] = (std::uint8_t)keylen;
            param[2] = 1; // fanout
            param[3] = 1; // depth
            Blake2b s;
            s.init(param);
            if (keylen)
            {
                std::uint8_t block[128] = { 0 };
                std::memcpy(block, key, keylen);
                s.update(block, 128);
            }
            s.update(in, inlen);
            s.final(out, outlen);
        }

        // BLAKE2Xb: a root hash of (key block, input) whose parameter block carries the XOF length, expanded in
        // counter mode, 64 bytes per node (same argument order as the reference's blake2xb(), util/blake2xb.c:151)
        inline void blake2xb(void *out, std::size_t outlen, const void *in, std::size_t inlen, const void *key,
                             std::size_t keylen)
        {
            if (!out || !outlen || outlen > 0xFFFFFFFFULL || keylen > 64 || (!in && inlen) || (!key && keylen))
                throw std::invalid_argument("blake2xb: invalid argument");
            std::uint8_t param[64] = { 0 };
            const std::uint32_t xof = (std::uint32_t)outlen;
            param[0] = 64;
            param[1] = (std::uint8_t)keylen;
            param[2] = 1;
            param[3] = 1;
            std::memcpy(param + 12, &xof, 4);
            Blake2b root;
            root.init(param);
            if (keylen)
            {
                std::uint8_t block[128] = { 0 };
                std::memcpy(block, key, keylen);
                root.update(block, 128);
            }
            root.update(in, inlen);
            std::uint8_t h0[64];
            root.final(h0, 64);
            // expansion nodes: key length 0, fanout 0, depth 0, leaf length 64, node offset i, inner length 64
            param[1] = 0;
            param[2] = 0;
            param[3] = 0;
            const std::uint32_t leaf = 64;
            std::memcpy(param + 4, &leaf, 4);
            param[16] = 0;  // node depth
            param[17] = 64; // inner length
            std::uint8_t *o = static_cast<std::uint8_t *>(out);
            for (std::uint32_t i = 0; outlen; ++i)
            {
                const std::size_t block = std::min<std::size_t>(outlen, 64);
                param[0] = (std::uint8_t)block;
                std::memcpy(param + 8, &i, 4);
                Blake2b node;
                node.init(param);
                node.update(h0, 64);
                node.final(o, block);
                o += block;
                outlen -= block;
            }
        }
    } // namespace util

    // randomgen.h:300-420: a buffered byte source; generate() hands out buffer bytes and refills at the end
    class UniformRandomGenerator
    {
    public:
        explicit UniformRandomGenerator(const prng_seed_type &seed) : seed_(seed), buffer_(buffer_size_), head_(buffer_size_)
        {}
        virtual ~UniformRandomGenerator() = default;
        void generate(std::size_t byte_count, void *destination) // randomgen.cpp:176-193
        {
            std::uint8_t *d = static_cast<std::uint8_t *>(destination);
            while (byte_count)
            {
                const std::size_t cur = std::min(byte_count, buffer_size_ - head_);
                std::memcpy(d, buffer_.data() + head_, cur);
                head_ += cur;
                d += cur;
                byte_count -= cur;
                if (head_ == buffer_size_)
                {
                    refill_buffer();
                    head_ = 0;
                }
            }
        }
        std::uint32_t generate() // randomgen.h:343-348
        {
            std::uint32_t r;
            generate(sizeof(r), &r);
            return r;
        }
        const prng_seed_type &seed() const noexcept
        {
            return seed_;
        }
        virtual prng_type type() const noexcept = 0;

    protected:
        virtual void refill_buffer() = 0;
        static constexpr std::size_t buffer_size_ = 4096; // randomgen.h:391
        prng_seed_type seed_;
        std::vector<std::uint8_t> buffer_;
        std::size_t head_;
    };

    // randomgen.cpp:201-211: buffer = BLAKE2Xb(out = 4096 bytes, in = 64-bit counter, key = the 64-byte seed)
    class Blake2xbPRNG : public UniformRandomGenerator
    {
    public:
        using UniformRandomGenerator::UniformRandomGenerator;
        prng_type type() const noexcept override
        {
            return prng_type::blake2xb;
        }

    protected:
        void refill_buffer() override
        {
            util::blake2xb(buffer_.data(), buffer_size_, &counter_, sizeof(counter_), seed_.data(), prng_seed_byte_count);
            ++counter_;
        }

    private:
        std::uint64_t counter_ = 0;
    };

    // randomgen.h:422-520: create() seeds from the system unless the factory was given a default seed, in which
    // case EVERY generator it creates starts from that seed (the reference's way of making runs reproducible)
    class UniformRandomGeneratorFactory
    {
    public:
        UniformRandomGeneratorFactory() : use_random_seed_(true)
        {}
        explicit UniformRandomGeneratorFactory(const prng_seed_type &default_seed)
            : default_seed_(default_seed), use_random_seed_(false)
        {}
        virtual ~UniformRandomGeneratorFactory() = default;
        std::shared_ptr<UniformRandomGenerator> create()
        {
            if (!use_random_seed_)
                return create_impl(default_seed_);
            prng_seed_type seed;
            std::random_device rd("/dev/urandom"); // random_bytes, randomgen.cpp:32-46
            for (auto &w : seed)
                w = ((std::uint64_t)rd() << 32) | rd();
            return create_impl(seed);
        }
        std::shared_ptr<UniformRandomGenerator> create(const prng_seed_type &seed)
        {
            return create_impl(seed);
        }
        bool use_random_seed() const noexcept
        {
            return use_random_seed_;
        }
        const prng_seed_type &default_seed() const noexcept
        {
            return default_seed_;
        }
        static std::shared_ptr<UniformRandomGeneratorFactory> DefaultFactory();

    protected:
        virtual std::shared_ptr<UniformRandomGenerator> create_impl(const prng_seed_type &seed) = 0;

    private:
        prng_seed_type default_seed_{};
        bool use_random_seed_;
    };

    class Blake2xbPRNGFactory : public UniformRandomGeneratorFactory
    {
    public:
        using UniformRandomGeneratorFactory::UniformRandomGeneratorFactory;

    protected:
        std::shared_ptr<UniformRandomGenerator> create_impl(const prng_seed_type &seed) override
        {
            return std::make_shared<Blake2xbPRNG>(seed);
        }
    };

    inline std::shared_ptr<UniformRandomGeneratorFactory> UniformRandomGeneratorFactory::DefaultFactory()
    {
        // SEAL_DEFAULT_PRNG = Blake2xb (CMakeLists.txt SEAL_DEFAULT_PRNG, util/defines.h)
        static std::shared_ptr<UniformRandomGeneratorFactory> f = std::make_shared<Blake2xbPRNGFactory>();
        return f;
    }

    // randomtostd.h:18-75: the std-style adapter std::uniform_int_distribution draws 32-bit words through
    class RandomToStandardAdapter
    {
    public:
        using result_type = std::uint32_t;
        explicit RandomToStandardAdapter(std::shared_ptr<UniformRandomGenerator> g) : g_(std::move(g))
        {
            if (!g_)
                throw std::invalid_argument("generator cannot be null");
        }
        result_type operator()()
        {
            return g_->generate();
        }
        static constexpr result_type min() noexcept
        {
            return 0;
        }
        static constexpr result_type max() noexcept
        {
            return 0xFFFFFFFFu;
        }

    private:
        std::shared_ptr<UniformRandomGenerator> g_;
    };

    namespace util
    {
        // util/rlwe.cpp:21-38.  destination: [limbs][N]; coefficient i is -1, 0 or 1 in every limb.  Like the
        // reference this draws through std::uniform_int_distribution, whose algorithm belongs to the C++ library
        // (libstdc++ 11: Lemire's method on one 32-bit word); built with the same library, the same seed gives the
        // reference's secret key.
        inline void sample_poly_ternary(std::shared_ptr<UniformRandomGenerator> prng, const std::vector<std::uint64_t> &q,
                                        std::size_t n, std::uint64_t *destination)
        {
            RandomToStandardAdapter engine(std::move(prng));
            std::uniform_int_distribution<std::uint64_t> dist(0, 2);
            for (std::size_t i = 0; i < n; ++i)
            {
                const std::uint64_t rand = dist(engine);
                const std::uint64_t flag = (std::uint64_t)(-(std::int64_t)(rand == 0));
                for (std::size_t j = 0; j < q.size(); ++j)
                    destination[i + j * n] = rand + (flag & q[j]) - 1;
            }
        }

        // util/rlwe.cpp:66-101 (SEAL_NOISE_SAMPLER without SEAL_USE_GAUSSIAN_NOISE, util/defines.h:195-199): centred
        // binomial, 21 coin pairs from 6 bytes per coefficient, standard deviation 3.2
        inline void sample_poly_cbd(const std::shared_ptr<UniformRandomGenerator> &prng, const std::vector<std::uint64_t> &q,
                                    std::size_t n, std::uint64_t *destination)
        {
            for (std::size_t i = 0; i < n; ++i)
            {
                unsigned char x[6];
                prng->generate(6, x);
                x[2] &= 0x1F;
                x[5] &= 0x1F;
                const std::int32_t noise = __builtin_popcount(x[0]) + __builtin_popcount(x[1]) + __builtin_popcount(x[2]) -
                                           __builtin_popcount(x[3]) - __builtin_popcount(x[4]) - __builtin_popcount(x[5]);
                const std::uint64_t flag = (std::uint64_t)(-(std::int64_t)(noise < 0));
                for (std::size_t j = 0; j < q.size(); ++j)
                    destination[i + j * n] = (std::uint64_t)(std::int64_t)noise + (flag & q[j]);
            }
        }

        // util/rlwe.cpp:103-131: the buffer is filled in one draw, then every limb is made uniform modulo its prime
        // by redrawing single words at or above the largest multiple of the prime
        inline void sample_poly_uniform(const std::shared_ptr<UniformRandomGenerator> &prng, const std::vector<std::uint64_t> &q,
                                        std::size_t n, std::uint64_t *destination)
        {
            prng->generate(q.size() * n * sizeof(std::uint64_t), destination);
            constexpr std::uint64_t max_random = 0xFFFFFFFFFFFFFFFFULL;
            for (std::size_t j = 0; j < q.size(); ++j)
            {
                const std::uint64_t max_multiple = max_random - max_random % q[j] - 1;
                for (std::size_t i = 0; i < n; ++i)
                {
                    std::uint64_t rand = destination[i + j * n];
                    while (rand >= max_multiple)
                        prng->generate(sizeof(rand), &rand);
                    destination[i + j * n] = rand % q[j];
                }
            }
        }
    } // namespace util
} // namespace seal
