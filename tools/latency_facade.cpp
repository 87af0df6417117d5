// Single-ciphertext latencies at the reference's published configuration: CKKS, N = 32768,
// CoeffModulus::BFVDefault(32768) = 16 primes, one ciphertext at a time through seal::Evaluator -- the shape of
// native/examples/6_performance.cpp:348-603 (`count` runs per operation, wall clock around each call; the reference's
// GPU build synchronises inside its calls, here the clock stops after cat_stream_sync) -- and the same operations through
// the C ABI at batch 1.  Prints one JSON object.  Median and mean of `count` timed runs after one warm-up run.
//   g++ -std=c++17 -O2 tools/latency_facade.cpp -o tools/latency_facade -Lcat_amd/lib -lcat_amd -Wl,-rpath,'$ORIGIN/../cat_amd/lib'
#include "../cat_amd/host/seal_hip.hpp"
#include "../cat_amd/host/seal_hip_crypto.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <random>
#include <string>
#include <vector>

using namespace seal;
using clk = std::chrono::steady_clock;

static double us_since(clk::time_point t0)
{
    return std::chrono::duration<double, std::micro>(clk::now() - t0).count();
}

struct Series
{
    std::vector<double> v;
    double median() const
    {
        std::vector<double> s(v);
        std::sort(s.begin(), s.end());
        return s.empty() ? 0 : (s.size() & 1) ? s[s.size() / 2] : 0.5 * (s[s.size() / 2 - 1] + s[s.size() / 2]);
    }
    double mean() const
    {
        double t = 0;
        for (double x : v)
            t += x;
        return v.empty() ? 0 : t / v.size();
    }
};

int main(int argc, char **argv)
{
    const int count = argc > 1 ? std::atoi(argv[1]) : 10;
    const std::size_t n = argc > 2 ? (std::size_t)std::atoll(argv[2]) : 32768;
    try
    {
        EncryptionParameters parms(scheme_type::ckks);
        parms.set_poly_modulus_degree(n);
        parms.set_coeff_modulus(CoeffModulus::BFVDefault(n));
        const prng_seed_type seed = { 1, 2, 3, 4, 5, 6, 7, 8 };
        parms.set_random_generator(std::make_shared<Blake2xbPRNGFactory>(seed));
        const auto t_ctx = clk::now();
        SEALContext context(parms);
        cat_ctx *h = context.handle();
        const double ctx_s = us_since(t_ctx) * 1e-6;
        const std::size_t k = context.key_modulus_size(), L = context.first_level_size();

        const auto t_keys = clk::now();
        KeyGenerator keygen(context);
        PublicKey public_key;
        keygen.create_public_key(public_key);
        RelinKeys relin_keys(context);
        keygen.create_relin_keys(relin_keys);
        // KeyGenerator::create_galois_keys(GaloisKeys &) of the reference: every power-of-two step both ways and the
        // conjugation (GaloisTool::get_elts_all, util/galois.cpp:97-130)
        GaloisKeys gal_keys(context);
        std::vector<int> steps;
        for (std::size_t s = 1; s < n / 2; s <<= 1)
        {
            steps.push_back((int)s);
            steps.push_back(-(int)s);
        }
        keygen.create_galois_keys(steps, gal_keys);
        keygen.create_galois_keys(std::vector<std::uint32_t>{ (std::uint32_t)(2 * n - 1) }, gal_keys);
        const double keys_s = us_since(t_keys) * 1e-6;

        Encryptor encryptor(context, public_key);
        Evaluator evaluator(context);
        const double scale = std::pow(2.0, 40);
        std::mt19937_64 rng(7);

        // an NTT-form plaintext at the first level with uniform residues (the encoder is not on the path)
        Plaintext plain(context, L);
        {
            const std::vector<std::uint64_t> q = context.parms().coeff_modulus_words();
            std::vector<std::uint64_t> w(L * n);
            for (std::size_t j = 0; j < L; ++j)
                for (std::size_t i = 0; i < n; ++i)
                    w[j * n + i] = rng() % q[j];
            plain.upload(w.data());
            plain.scale() = scale;
        }

        std::map<std::string, Series> t;
        auto timed = [&](const char *name, bool record, const std::function<void()> &f) {
            detail::check(cat_stream_sync(h, nullptr));
            const auto t0 = clk::now();
            f();
            detail::check(cat_stream_sync(h, nullptr));
            const double us = us_since(t0);
            if (record)
                t[name].v.push_back(us);
        };

        for (int run = -1; run < count; ++run)
        {
            const bool rec = run >= 0;
            Ciphertext e1, e2, e;
            encryptor.encrypt_zero(e1);
            encryptor.encrypt_zero(e2);
            encryptor.encrypt_zero(e);
            e1.scale() = e2.scale() = e.scale() = scale;
            timed("add", rec, [&] {
                evaluator.add_inplace(e1, e1);
                evaluator.add_inplace(e2, e2);
                evaluator.add_inplace(e1, e2);
            });
            timed("multiply", rec, [&] { evaluator.multiply_inplace(e1, e2); });
            timed("multiply_plain", rec, [&] { evaluator.multiply_plain_inplace(e2, plain); });
            timed("square", rec, [&] { evaluator.square_inplace(e2); });
            timed("relinearize", rec, [&] { evaluator.relinearize_inplace(e1, relin_keys); });
            timed("rescale", rec, [&] { evaluator.rescale_to_next_inplace(e1); });
            timed("rotate_vector_one_step", rec, [&] {
                evaluator.rotate_vector_inplace(e, 1, gal_keys);
                evaluator.rotate_vector_inplace(e, -1, gal_keys);
            });
            const int random_rotation = (int)(rng() & (n / 2 - 1));
            timed("rotate_vector_random", rec, [&] { evaluator.rotate_vector_inplace(e, random_rotation, gal_keys); });
            timed("complex_conjugate", rec, [&] { evaluator.complex_conjugate_inplace(e, gal_keys); });
            // not in the reference: eight different rotations of one ciphertext with the decomposition shared
            // (Evaluator::rotate_hoisted -> cat_rotate_hoisted), against the same eight as separate rotate_vector calls
            const std::vector<int> eight = { 1, 2, 4, 8, 16, 32, 64, 128 };
            std::vector<Ciphertext> rotated;
            timed("rotate_hoisted_x8_per_rotation", rec, [&] { evaluator.rotate_hoisted(e, eight, gal_keys, rotated); });
            timed("rotate_vector_x8_per_rotation", rec, [&] {
                for (int st : eight)
                {
                    Ciphertext r2;
                    evaluator.rotate_vector(e, st, gal_keys, r2);
                }
            });
        }
        // per-call figures as the reference prints them: add / 3, one-step rotation / 2 (6_performance.cpp:573-586)
        for (double &x : t["add"].v)
            x /= 3;
        for (double &x : t["rotate_vector_one_step"].v)
            x /= 2;
        for (double &x : t["rotate_hoisted_x8_per_rotation"].v)
            x /= 8;
        for (double &x : t["rotate_vector_x8_per_rotation"].v)
            x /= 8;

        // ---- the same operations through the C ABI, one ciphertext, device buffers held by the caller ---------------
        std::map<std::string, Series> c;
        {
            detail::DeviceWords x(h, 2 * L * n), y(h, 2 * L * n), p3(h, 3 * L * n), out(h, 2 * (L - 1) * n), ct(h, 2 * L * n);
            std::vector<std::uint64_t> w(3 * L * n);
            const std::vector<std::uint64_t> q = context.parms().coeff_modulus_words();
            auto fill = [&](detail::DeviceWords &d, std::size_t polys) {
                for (std::size_t pz = 0; pz < polys; ++pz)
                    for (std::size_t j = 0; j < L; ++j)
                        for (std::size_t i = 0; i < n; ++i)
                            w[(pz * L + j) * n + i] = rng() % q[j];
                detail::check(cat_upload(h, d.get(), w.data(), polys * L * n * 8, nullptr));
                detail::check(cat_stream_sync(h, nullptr));
            };
            fill(x, 2);
            fill(y, 2);
            fill(ct, 2);
            const std::uint64_t *rk = relin_keys.d_key(RelinKeys::get_index(2));
            const std::uint32_t elt1 = cat_galois_elt_from_step(cat_ctx_log_n(h), 1);
            const std::uint64_t *gk = gal_keys.d_key(GaloisKeys::get_index(elt1));
            auto ctimed = [&](const char *name, bool record, const std::function<void()> &f) {
                detail::check(cat_stream_sync(h, nullptr));
                const auto t0 = clk::now();
                f();
                detail::check(cat_stream_sync(h, nullptr));
                if (record)
                    c[name].v.push_back(us_since(t0));
            };
            for (int run = -1; run < count; ++run)
            {
                const bool rec = run >= 0;
                ctimed("cat_add", rec, [&] { detail::check(cat_add(h, x.get(), y.get(), out.get(), nullptr, (std::uint32_t)L, 1, nullptr)); });
                ctimed("cat_ckks_multiply", rec, [&] { detail::check(cat_ckks_multiply(h, x.get(), y.get(), p3.get(), (std::uint32_t)L, 1, nullptr)); });
                ctimed("cat_relinearize", rec, [&] { detail::check(cat_relinearize(h, p3.get(), rk, (std::uint32_t)L, 1, nullptr)); });
                ctimed("cat_rescale", rec, [&] { detail::check(cat_rescale_strided(h, p3.get(), 3 * L * n, out.get(), 2 * (L - 1) * n, (std::uint32_t)L, 2, 1, nullptr)); });
                ctimed("cat_ckks_multiply+cat_relinearize_rescale", rec, [&] {
                    detail::check(cat_ckks_multiply(h, x.get(), y.get(), p3.get(), (std::uint32_t)L, 1, nullptr));
                    detail::check(cat_relinearize_rescale(h, p3.get(), rk, out.get(), (std::uint32_t)L, 1, nullptr));
                });
                ctimed("cat_apply_galois_inplace", rec, [&] { detail::check(cat_apply_galois_inplace(h, ct.get(), elt1, gk, (std::uint32_t)L, 1, nullptr)); });
                ctimed("cat_ntt_forward (one ciphertext, 2 x L limbs)", rec, [&] { detail::check(cat_ntt_forward(h, ct.get(), nullptr, (std::uint32_t)L, 2, 0, nullptr)); });
                ctimed("cat_ntt_inverse (one ciphertext, 2 x L limbs)", rec, [&] { detail::check(cat_ntt_inverse(h, ct.get(), nullptr, (std::uint32_t)L, 2, 0, nullptr)); });
            }
        }

        std::printf("{\"workload\": \"CKKS N=%zu, BFVDefault(%zu) = %zu primes (%zu ciphertext limbs + special), one ciphertext per call\", "
                    "\"count\": %d, \"context_s\": %.2f, \"keygen_s\": %.2f, \"galois_keys\": %zu, \"workspace_grew\": %llu, "
                    "\"timing\": \"host clock around each call, stopped after cat_stream_sync (6_performance.cpp:348-603 times the same way)\", ",
                    n, n, k, L, count, ctx_s, keys_s, steps.size() + 1, (unsigned long long)cat_ctx_workspace_grew(h));
        auto dump = [&](const char *key, const std::map<std::string, Series> &m, const char *which) {
            std::printf("\"%s\": {", key);
            bool first = true;
            for (const auto &kv : m)
            {
                std::printf("%s\"%s\": %.1f", first ? "" : ", ", kv.first.c_str(),
                            std::string(which) == "median" ? kv.second.median() : kv.second.mean());
                first = false;
            }
            std::printf("}");
        };
        dump("evaluator_median_us", t, "median");
        std::printf(", ");
        dump("evaluator_mean_us", t, "mean");
        std::printf(", ");
        dump("c_abi_batch1_median_us", c, "median");
        std::printf("}\n");
    }
    catch (const std::exception &e)
    {
        std::printf("{\"error\": \"%s\"}\n", e.what());
        return 1;
    }
    return 0;
}
