// Host-only check (no GPU): the facade's PRNG and samplers (cat_amd/host/seal_hip_random.hpp, product code) against
// the CPU oracle's restatement (oracle/seal_oracle.c) and, when it was built, against the reference's own BLAKE2 C
// files (oracle/_ref/libblake2_ref.so, opened at run time).  Built by `make random_test`, run by
// tests/test_random_keys.py.
#include "../../cat_amd/host/seal_hip_random.hpp"
#include "../../oracle/seal_oracle.h"

#include <cstdio>
#include <dlfcn.h>
#include <string>

static int failures = 0;
#define EXPECT(cond)                                                    \
    do                                                                  \
    {                                                                   \
        if (!(cond))                                                    \
        {                                                               \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++failures;                                                 \
        }                                                               \
    } while (0)

typedef int (*xb_fn)(void *, size_t, const void *, size_t, const void *, size_t);

int main(int argc, char **argv)
{
    using namespace seal;
    unsigned char in[300], key[64];
    for (int i = 0; i < 300; ++i)
        in[i] = (unsigned char)(i * 7 + 1);
    for (int i = 0; i < 64; ++i)
        key[i] = (unsigned char)i;
    xb_fn ref_xb = nullptr, ref_b = nullptr;
    if (argc > 1)
        if (void *so = dlopen(argv[1], RTLD_NOW))
        {
            ref_xb = (xb_fn)dlsym(so, "blake2xb");
            ref_b = (xb_fn)dlsym(so, "blake2b");
        }
    const size_t outs[] = { 1, 32, 63, 64, 65, 128, 200, 4096, 5000 }, ins[] = { 0, 1, 8, 127, 128, 129, 256, 300 },
                 keys[] = { 0, 1, 32, 64 };
    for (size_t o : outs)
        for (size_t il : ins)
            for (size_t kl : keys)
            {
                std::vector<unsigned char> a(o), b(o), c(o);
                util::blake2xb(a.data(), o, in, il, kl ? key : nullptr, kl);
                EXPECT(so_blake2xb(b.data(), o, in, il, kl ? key : nullptr, kl) == 0 && a == b);
                if (ref_xb)
                    EXPECT(ref_xb(c.data(), o, in, il, kl ? key : nullptr, kl) == 0 && a == c);
                if (o <= 64)
                {
                    util::blake2b(a.data(), o, in, il, kl ? key : nullptr, kl);
                    EXPECT(so_blake2b(b.data(), o, in, il, kl ? key : nullptr, kl) == 0 && a == b);
                    if (ref_b)
                        EXPECT(ref_b(c.data(), o, in, il, kl ? key : nullptr, kl) == 0 && a == c);
                }
            }

    // generator stream in ragged pieces across buffer refills (randomgen.cpp:176-193)
    const prng_seed_type seed = { 1, 2, 3, 4, 5, 6, 7, 8 };
    {
        Blake2xbPRNG g(seed);
        so_prng *p = so_prng_create(seed.data());
        const size_t pieces[] = { 1, 4, 6, 4095, 1, 8192, 13, 4096, 64 };
        for (size_t n : pieces)
        {
            std::vector<unsigned char> a(n), b(n);
            g.generate(n, a.data());
            so_prng_generate(p, n, b.data());
            EXPECT(a == b);
        }
        so_prng_destroy(p);
        // a fixed-seed factory starts every generator from the same seed (randomgen.h:436-447)
        Blake2xbPRNGFactory f(seed);
        EXPECT(!f.use_random_seed() && f.create()->generate() == f.create()->generate());
        EXPECT(Blake2xbPRNGFactory().use_random_seed());
    }

    // samplers (util/rlwe.cpp:21-131): same words from the same generator state, several prime shapes
    const std::vector<std::vector<std::uint64_t>> prime_sets = { { 0xffffffffffc0001ULL, 0xfffffffff840001ULL, 1099511603713ULL },
                                                                 { 12289 },
                                                                 { 0x1fffffffffe00001ULL, 65537 } };
    for (const auto &q : prime_sets)
    {
        const size_t n = 2048;
        std::vector<std::uint64_t> a(q.size() * n), b(q.size() * n);
        auto g = std::make_shared<Blake2xbPRNG>(seed);
        so_prng *p = so_prng_create(seed.data());
        // one generator feeds all three in turn, as in encrypt_zero_asymmetric
        util::sample_poly_ternary(g, q, n, a.data());
        so_sample_poly_ternary(p, q.data(), q.size(), n, b.data());
        EXPECT(a == b);
        size_t minus = 0, zero = 0, plus = 0;
        for (size_t i = 0; i < n; ++i)
        {
            minus += a[i] == q[0] - 1;
            zero += a[i] == 0;
            plus += a[i] == 1;
        }
        EXPECT(minus + zero + plus == n && minus > n / 4 && zero > n / 4 && plus > n / 4);
        util::sample_poly_cbd(g, q, n, a.data());
        so_sample_poly_cbd(p, q.data(), q.size(), n, b.data());
        EXPECT(a == b);
        for (size_t i = 0; i < n; ++i) // |noise| <= 21, the same signed value in every limb
        {
            const std::int64_t v = a[i] > q[0] / 2 ? (std::int64_t)a[i] - (std::int64_t)q[0] : (std::int64_t)a[i];
            EXPECT(v >= -21 && v <= 21);
            for (size_t j = 1; j < q.size(); ++j)
                EXPECT((a[i + j * n] > q[j] / 2 ? (std::int64_t)a[i + j * n] - (std::int64_t)q[j] : (std::int64_t)a[i + j * n]) == v);
        }
        util::sample_poly_uniform(g, q, n, a.data());
        so_sample_poly_uniform(p, q.data(), q.size(), n, b.data());
        EXPECT(a == b);
        for (size_t j = 0; j < q.size(); ++j)
            for (size_t i = 0; i < n; ++i)
                EXPECT(a[i + j * n] < q[j]);
        // still in step afterwards
        EXPECT(g->generate() == [&] { std::uint32_t r; so_prng_generate(p, 4, &r); return r; }());
        so_prng_destroy(p);
    }
    if (failures)
    {
        std::printf("%d check(s) failed\n", failures);
        return 1;
    }
    std::printf("random ok: facade PRNG and samplers match the oracle%s\n", ref_xb ? " and the reference's BLAKE2 build" : "");
    return 0;
}
