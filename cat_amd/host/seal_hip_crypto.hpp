// seal_hip_crypto.hpp -- key generation, encryption and decryption of the facade with the
// reference's class names: SecretKey, PublicKey, KeyGenerator (keygenerator.cpp:56-363), Encryptor
// (encryptor.cpp:84-324), Decryptor (decryptor.cu:544-670) and util::encrypt_zero_symmetric / _asymmetric
// (util/rlwe.cpp:189-412).
//
// Randomness is drawn on the host exactly as the reference draws it (seal_hip_random.hpp: Blake2xb PRNG, ternary /
// centred-binomial / uniform samplers), so a fixed-seed factory gives the reference's keys and ciphertexts; every
// polynomial operation (NTT, dyadic products, sums, Galois permutation of the secret key, the modulus switch of a
// public-key encryption) runs on the device through the C ABI.  A level is named by its limb count L as in
// seal_hip.hpp; the key level has all k primes.
//
// BFV / BGV plaintexts are host coefficient vectors modulo t: their scaling (util/scalingvariant.cpp:69-113) or
// lift (encryptor.cpp:261-304) and the RNS tail of decryption (RNSTool::decrypt_scale_and_round, decrypt_modt,
// util/rns.cpp:466-539, 1133-1252) are small per-coefficient host loops around the device work.  Not restated:
// the encoders (CKKSEncoder, BatchEncoder).
#pragma once

#include "seal_hip.hpp"

#include <sstream>

namespace seal
{
    namespace detail
    {
        // words of device memory owned through the C ABI
        class DeviceWords
        {
        public:
            DeviceWords(cat_ctx *h, std::size_t words) : h_(h), words_(words)
            {
                check(cat_malloc(h_, words * 8, (void **)&d_));
            }
            DeviceWords(cat_ctx *h, const std::vector<std::uint64_t> &host) : DeviceWords(h, host.size())
            {
                upload(host.data());
            }
            DeviceWords(const DeviceWords &) = delete;
            DeviceWords &operator=(const DeviceWords &) = delete;
            ~DeviceWords()
            {
                if (d_)
                    cat_free(h_, d_);
            }
            void upload(const std::uint64_t *host)
            {
                check(cat_upload(h_, d_, host, words_ * 8, nullptr));
                check(cat_stream_sync(h_, nullptr));
            }
            std::vector<std::uint64_t> to_cpu() const
            {
                std::vector<std::uint64_t> out(words_);
                check(cat_download(h_, out.data(), d_, words_ * 8, nullptr));
                check(cat_stream_sync(h_, nullptr));
                return out;
            }
            std::uint64_t *get() const noexcept
            {
                return d_;
            }
            std::size_t words() const noexcept
            {
                return words_;
            }

        private:
            cat_ctx *h_;
            std::uint64_t *d_ = nullptr;
            std::size_t words_;
        };

        // ---- small multi-word helpers (little-endian words) for the per-level plaintext constants -------------------
        using u128 = unsigned __int128;
        inline std::vector<std::uint64_t> mp_mul_word(const std::vector<std::uint64_t> &a, std::uint64_t w)
        {
            std::vector<std::uint64_t> r(a.size() + 1, 0);
            std::uint64_t carry = 0;
            for (std::size_t i = 0; i < a.size(); ++i)
            {
                const u128 t = (u128)a[i] * w + carry;
                r[i] = (std::uint64_t)t;
                carry = (std::uint64_t)(t >> 64);
            }
            r[a.size()] = carry;
            return r;
        }
        inline std::uint64_t mp_mod_word(const std::vector<std::uint64_t> &a, std::uint64_t p)
        {
            u128 r = 0;
            for (std::size_t i = a.size(); i--;)
                r = ((r << 64) | a[i]) % p;
            return (std::uint64_t)r;
        }
        inline std::vector<std::uint64_t> mp_div_word(const std::vector<std::uint64_t> &a, std::uint64_t w)
        {
            std::vector<std::uint64_t> q(a.size(), 0);
            u128 r = 0;
            for (std::size_t i = a.size(); i--;)
            {
                const u128 cur = (r << 64) | a[i];
                q[i] = (std::uint64_t)(cur / w);
                r = cur % w;
            }
            return q;
        }
        inline std::vector<std::uint64_t> mp_product(const std::vector<std::uint64_t> &vals, std::size_t skip)
        {
            std::vector<std::uint64_t> r{ 1 };
            for (std::size_t i = 0; i < vals.size(); ++i)
                if (i != skip)
                    r = mp_mul_word(r, vals[i]);
            return r;
        }
        inline std::uint64_t mul_mod(std::uint64_t a, std::uint64_t b, std::uint64_t p)
        {
            return (std::uint64_t)((u128)a * b % p);
        }
        inline bool try_invert_mod(std::uint64_t a, std::uint64_t m, std::uint64_t &out)
        {
            __int128 r0 = m, r1 = a % m, t0 = 0, t1 = 1;
            while (r1)
            {
                const __int128 q = r0 / r1, r2 = r0 - q * r1, t2 = t0 - q * t1;
                r0 = r1, r1 = r2, t0 = t1, t1 = t2;
            }
            if (r0 != 1)
                return false;
            out = (std::uint64_t)(t0 < 0 ? t0 + m : t0);
            return true;
        }

        // What SEALContext::ContextData and RNSTool hold for the plaintext side of one level (context.cu:441-491,
        // util/rns.cpp:640-765): host constants, built when an Encryptor / Decryptor first meets the level.
        struct PlainTools
        {
            std::vector<std::uint64_t> q;            // the level's primes
            std::uint64_t t = 0, gamma = 0;
            std::uint64_t q_mod_t = 0, threshold = 0; // coeff_modulus_mod_plain_modulus, plain_upper_half_threshold
            std::vector<std::uint64_t> div;          // coeff_div_plain_modulus: floor(q / t) mod q_i
            std::vector<std::uint64_t> inv_punc;     // ((q / q_i) mod q_i)^-1 mod q_i
            std::vector<std::uint64_t> to_t, to_gamma; // (q / q_i) mod t, mod gamma
            std::vector<std::uint64_t> prod_t_gamma; // t gamma mod q_i
            std::uint64_t neg_inv_q_t = 0, neg_inv_q_gamma = 0, inv_gamma_t = 0;

            PlainTools(const std::vector<std::uint64_t> &primes, std::uint64_t plain_modulus, std::uint64_t gamma_prime)
                : q(primes), t(plain_modulus), gamma(gamma_prime)
            {
                const std::vector<std::uint64_t> Q = mp_product(q, (std::size_t)-1);
                q_mod_t = mp_mod_word(Q, t);
                threshold = (t + 1) >> 1;
                const std::vector<std::uint64_t> Q_div_t = mp_div_word(Q, t);
                for (std::size_t i = 0; i < q.size(); ++i)
                {
                    div.push_back(mp_mod_word(Q_div_t, q[i]));
                    const std::vector<std::uint64_t> punc = mp_product(q, i);
                    std::uint64_t inv = 0;
                    if (!try_invert_mod(mp_mod_word(punc, q[i]), q[i], inv))
                        throw std::logic_error("invalid rns bases");
                    inv_punc.push_back(inv);
                    to_t.push_back(mp_mod_word(punc, t));
                    if (gamma)
                    {
                        to_gamma.push_back(mp_mod_word(punc, gamma));
                        prod_t_gamma.push_back(mul_mod(t % q[i], gamma % q[i], q[i]));
                    }
                }
                if (gamma)
                {
                    std::uint64_t inv = 0;
                    if (!try_invert_mod(q_mod_t, t, inv))
                        throw std::logic_error("invalid rns bases");
                    neg_inv_q_t = inv ? t - inv : 0;
                    if (!try_invert_mod(mp_mod_word(Q, gamma), gamma, inv))
                        throw std::logic_error("invalid rns bases");
                    neg_inv_q_gamma = gamma - inv;
                    if (!try_invert_mod(gamma % t, t, inv_gamma_t))
                        throw std::logic_error("invalid rns bases");
                }
            }

            // multiply_add_plain_with_scaling_variant (util/scalingvariant.cpp:69-113): out[i][idx] = round(q m / t) mod q_i
            void scale_plain(const std::vector<std::uint64_t> &m, std::size_t n, std::uint64_t *out) const
            {
                std::fill(out, out + q.size() * n, 0);
                for (std::size_t idx = 0; idx < m.size(); ++idx)
                {
                    const std::uint64_t fix = (std::uint64_t)(((u128)m[idx] * q_mod_t + threshold) / t);
                    for (std::size_t i = 0; i < q.size(); ++i)
                        out[i * n + idx] = (std::uint64_t)(((u128)m[idx] * div[i] + fix) % q[i]);
                }
            }
            // the BGV lift (encryptor.cpp:261-304): the upper half of [0, t) stands for negatives; both branches of the
            // reference (fast lift or decomposition of m + q - t) leave (m - t) mod q_i
            void lift_plain(const std::vector<std::uint64_t> &m, std::size_t n, std::uint64_t *out) const
            {
                std::fill(out, out + q.size() * n, 0);
                for (std::size_t idx = 0; idx < m.size(); ++idx)
                    for (std::size_t i = 0; i < q.size(); ++i)
                    {
                        const std::uint64_t v = m[idx] % q[i];
                        out[i * n + idx] = m[idx] >= threshold ? (v + q[i] - t % q[i]) % q[i] : v;
                    }
            }
            // BaseConverter::fast_convert_array (util/rns.cpp:418-463) for one coefficient into one modulus
            std::uint64_t convert(const std::uint64_t *temp, const std::vector<std::uint64_t> &matrix, std::uint64_t p) const
            {
                u128 acc = 0;
                for (std::size_t i = 0; i < q.size(); ++i)
                    acc = (acc + (u128)temp[i] * matrix[i]) % p;
                return (std::uint64_t)acc;
            }
            // RNSTool::decrypt_scale_and_round (util/rns.cpp:1133-1201): phase [L][N] -> N coefficients modulo t
            void decrypt_scale_and_round(const std::uint64_t *phase, std::size_t n, std::uint64_t *out) const
            {
                std::vector<std::uint64_t> temp(q.size());
                const std::uint64_t gamma_div_2 = gamma >> 1;
                for (std::size_t j = 0; j < n; ++j)
                {
                    for (std::size_t i = 0; i < q.size(); ++i)
                        temp[i] = mul_mod(mul_mod(phase[i * n + j], prod_t_gamma[i], q[i]), inv_punc[i], q[i]);
                    const std::uint64_t vt = mul_mod(convert(temp.data(), to_t, t), neg_inv_q_t, t);
                    const std::uint64_t vg = mul_mod(convert(temp.data(), to_gamma, gamma), neg_inv_q_gamma, gamma);
                    std::uint64_t r;
                    if (vg > gamma_div_2)
                        r = (vt + (gamma - vg) % t) % t;
                    else
                        r = (vt + t - vg % t) % t;
                    out[j] = r ? mul_mod(r, inv_gamma_t, t) : 0;
                }
            }
            // RNSTool::decrypt_modt = BaseConverter::exact_convert_array (util/rns.cpp:466-539): the rounding term is
            // a double-precision sum in limb order, as in the reference
            void decrypt_modt(const std::uint64_t *phase, std::size_t n, std::uint64_t *out) const
            {
                std::vector<std::uint64_t> temp(q.size());
                for (std::size_t j = 0; j < n; ++j)
                {
                    double v = 0.0;
                    for (std::size_t i = 0; i < q.size(); ++i)
                    {
                        temp[i] = mul_mod(phase[i * n + j], inv_punc[i], q[i]);
                        v += static_cast<double>(temp[i]) / static_cast<double>(q[i]);
                    }
                    v += 0.5;
                    const std::uint64_t rounded = static_cast<std::uint64_t>(v);
                    const std::uint64_t sum = convert(temp.data(), to_t, t);
                    const std::uint64_t vq = mul_mod(rounded % t, q_mod_t, t);
                    out[j] = (sum + t - vq) % t;
                }
            }
        };

        inline std::vector<std::uint64_t> level_primes(const SEALContext &context, std::size_t L)
        {
            const auto &q = context.parms().coeff_modulus_words();
            if (L == 0 || L > q.size())
                throw std::invalid_argument("parms_id is not valid for encryption parameters");
            return std::vector<std::uint64_t>(q.begin(), q.begin() + (std::ptrdiff_t)L);
        }
    } // namespace detail

    // secretkey.cuh: the ternary secret in NTT form over all k key-level primes, [k][N] words, device-resident
    class SecretKey
    {
    public:
        SecretKey() = default;
        SecretKey(const SEALContext &context, const std::uint64_t *host_ntt_form)
            : data_(std::make_shared<detail::DeviceWords>(context.handle(), context.key_modulus_size() * context.poly_modulus_degree()))
        {
            data_->upload(host_ntt_form);
        }
        const std::uint64_t *d_data() const
        {
            if (!data_)
                throw std::logic_error("secret key is not set");
            return data_->get();
        }
        std::vector<std::uint64_t> to_cpu() const
        {
            if (!data_)
                throw std::logic_error("secret key is not set");
            return data_->to_cpu();
        }
        bool is_set() const noexcept
        {
            return (bool)data_;
        }
        // SecretKey::save / unsafe_load (secretkey.cuh:140-260): the key is kept as a Plaintext at the key level --
        // key parms_id, coeff_count = k N, scale 1.0, data (plaintext.cpp:207-233)
        std::streamoff save(const SEALContext &context, std::ostream &stream) const
        {
            const std::vector<std::uint64_t> host = to_cpu();
            return Serialization::Save(
                [&](std::ostream &s) {
                    util::write_pod(s, context.key_parms_id());
                    util::write_pod<std::uint64_t>(s, host.size());
                    util::write_pod<double>(s, 1.0);
                    util::dynarray_save(s, host.data(), host.size());
                },
                static_cast<std::streamoff>(sizeof(Serialization::SEALHeader) + sizeof(parms_id_type) + 16) +
                    util::dynarray_save_size(host.size()),
                stream);
        }
        std::streamoff load(const SEALContext &context, std::istream &stream)
        {
            return Serialization::Load(
                [&](std::istream &s, SEALVersion) {
                    const parms_id_type id = util::read_pod<parms_id_type>(s);
                    const std::uint64_t coeff_count = util::read_pod<std::uint64_t>(s);
                    (void)util::read_pod<double>(s);
                    const std::size_t words = context.key_modulus_size() * context.poly_modulus_degree();
                    if (id != context.key_parms_id() || coeff_count != words) // is_valid_for(SecretKey), valcheck.cpp
                        throw std::logic_error("SecretKey data is invalid");
                    const std::vector<std::uint64_t> data = util::dynarray_load(s, words);
                    if (data.size() != words)
                        throw std::logic_error("SecretKey data is invalid");
                    const auto &q = context.parms().coeff_modulus_words();
                    for (std::size_t j = 0; j < q.size(); ++j)
                        for (std::size_t i = 0; i < context.poly_modulus_degree(); ++i)
                            if (data[j * context.poly_modulus_degree() + i] >= q[j])
                                throw std::logic_error("SecretKey data is invalid");
                    *this = SecretKey(context, data.data());
                },
                stream);
        }

    private:
        friend class KeyGenerator;
        std::shared_ptr<detail::DeviceWords> data_;
    };

    // publickey.h: an encryption of zero at the key level (2 x k x N words, NTT form)
    class PublicKey
    {
    public:
        Ciphertext &data() noexcept
        {
            return ct_;
        }
        const Ciphertext &data() const noexcept
        {
            return ct_;
        }
        // publickey.h:84-190: a PublicKey is saved as its Ciphertext; a loaded one must sit at the key level
        std::streamoff save(std::ostream &stream) const
        {
            return ct_.save(stream);
        }
        std::streamoff load(const SEALContext &context, std::istream &stream)
        {
            Ciphertext fresh;
            const std::streamoff in_size = fresh.load(context, stream);
            if (fresh.size() != 2 || fresh.coeff_modulus_size() != context.key_modulus_size() || !fresh.is_ntt_form())
                throw std::logic_error("PublicKey data is invalid");
            ct_ = std::move(fresh);
            return in_size;
        }

    private:
        Ciphertext ct_;
    };

    namespace util
    {
        // util/rlwe.cpp:282-412 without the seed-saving branch: (c0, c1) = (-(a s + e), a), BGV: -(a s + t e).
        // d_c0, d_c1: L x N device words each.  `a` comes from a second generator seeded with 64 bytes of the
        // first one (:327-334); the noise from the first one (:360).
        // save_seed (:305-317, :347-356, :404-411): c1 is left as an indicator word followed by the serialized
        // generator info, from which loading re-expands it; off when a polynomial is too small to hold them.
        inline void encrypt_zero_symmetric(const SecretKey &secret_key, const SEALContext &context, std::size_t L,
                                           bool is_ntt_form, std::uint64_t *d_c0, std::uint64_t *d_c1, bool save_seed = false)
        {
            cat_ctx *h = context.handle();
            const std::size_t n = context.poly_modulus_degree();
            const std::vector<std::uint64_t> q = detail::level_primes(context, L);
            const std::uint32_t limbs = (std::uint32_t)L;

            auto bootstrap_prng = context.parms().random_generator()->create();
            prng_seed_type public_prng_seed;
            bootstrap_prng->generate(prng_seed_byte_count, public_prng_seed.data());
            auto ciphertext_prng = UniformRandomGeneratorFactory::DefaultFactory()->create(public_prng_seed);

            const std::size_t info_bytes = (std::size_t)UniformRandomGeneratorInfo::SaveSize();
            const std::size_t info_words = (info_bytes + 7) / 8;
            if (save_seed && L * n < info_words + 1)
                save_seed = false;

            std::vector<std::uint64_t> a(L * n), noise(L * n);
            // taken as the NTT form of a (:342-346), or as its coefficient form when a seed is kept (:347-356)
            sample_poly_uniform(ciphertext_prng, q, n, a.data());
            sample_poly_cbd(bootstrap_prng, q, n, noise.data());
            detail::check(cat_upload(h, d_c1, a.data(), a.size() * 8, nullptr));
            detail::check(cat_stream_sync(h, nullptr));
            if (!is_ntt_form && save_seed)
                detail::check(cat_ntt_forward(h, d_c1, nullptr, limbs, 1, 0, nullptr));
            detail::DeviceWords d_noise(h, noise);

            detail::check(cat_dyadic(h, secret_key.d_data(), d_c1, d_c0, nullptr, limbs, 1, nullptr)); // :366-368
            if (is_ntt_form)
                detail::check(cat_ntt_forward(h, d_noise.get(), nullptr, limbs, 1, 0, nullptr)); // :372
            else
                detail::check(cat_ntt_inverse(h, d_c0, nullptr, limbs, 1, 0, nullptr)); // :376
            if (context.parms().scheme() == scheme_type::bgv)
            {
                const std::vector<std::uint64_t> t(L, context.parms().plain_modulus());
                detail::check(cat_mul_scalar(h, d_noise.get(), t.data(), d_noise.get(), nullptr, limbs, 1, nullptr)); // :379-385
            }
            detail::check(cat_add(h, d_noise.get(), d_c0, d_c0, nullptr, limbs, 1, nullptr)); // :388-390
            detail::check(cat_negate(h, d_c0, d_c0, nullptr, limbs, 1, nullptr));             // :392
            if (!is_ntt_form && !save_seed)
                detail::check(cat_ntt_inverse(h, d_c1, nullptr, limbs, 1, 0, nullptr)); // :395-402
            detail::check(cat_stream_sync(h, nullptr));
            if (save_seed)
            {
                // the indicator word, then the info written over the following bytes (the tail of its last word
                // keeps what the polynomial held there, as in the reference)
                std::vector<std::uint64_t> head(info_words + 1);
                detail::check(cat_download(h, head.data(), d_c1, head.size() * 8, nullptr));
                detail::check(cat_stream_sync(h, nullptr));
                head[0] = 0xFFFFFFFFFFFFFFFFULL;
                std::ostringstream os(std::ios::binary);
                UniformRandomGeneratorInfo(ciphertext_prng->type(), ciphertext_prng->seed()).save(os);
                const std::string bytes = os.str();
                std::memcpy(head.data() + 1, bytes.data(), info_bytes);
                detail::check(cat_upload(h, d_c1, head.data(), head.size() * 8, nullptr));
                detail::check(cat_stream_sync(h, nullptr));
            }
        }

        inline void encrypt_zero_symmetric(const SecretKey &secret_key, const SEALContext &context, std::size_t L,
                                           bool is_ntt_form, bool save_seed, Ciphertext &destination)
        {
            destination.resize(context, L, 2);
            destination.is_ntt_form() = is_ntt_form;
            destination.scale() = 1.0;
            destination.correction_factor() = 1;
            encrypt_zero_symmetric(secret_key, context, L, is_ntt_form, destination.d_data(0), destination.d_data(1), save_seed);
        }

        // util/rlwe.cpp:189-280: c[j] = pk[j] u + e[j] (BGV: + t e[j]), u ternary; one generator for u, e[0], e[1]
        inline void encrypt_zero_asymmetric(const PublicKey &public_key, const SEALContext &context, std::size_t L,
                                            bool is_ntt_form, Ciphertext &destination)
        {
            cat_ctx *h = context.handle();
            const std::size_t n = context.poly_modulus_degree();
            const std::vector<std::uint64_t> q = detail::level_primes(context, L);
            const std::uint32_t limbs = (std::uint32_t)L;
            const Ciphertext &pk = public_key.data();
            if (pk.size() != 2 || pk.coeff_modulus_size() != context.key_modulus_size() || !pk.is_ntt_form())
                throw std::invalid_argument("public key is not valid for the encryption parameters");
            const bool bgv = context.parms().scheme() == scheme_type::bgv;

            destination.resize(context, L, 2);
            destination.is_ntt_form() = is_ntt_form;
            destination.scale() = 1.0;
            destination.correction_factor() = 1;

            auto prng = context.parms().random_generator()->create();
            std::vector<std::uint64_t> tmp(L * n);
            sample_poly_ternary(prng, q, n, tmp.data());
            detail::DeviceWords d_u(h, tmp);
            detail::check(cat_ntt_forward(h, d_u.get(), nullptr, limbs, 1, 0, nullptr));
            for (std::size_t j = 0; j < 2; ++j)
            {
                // limb i of polynomial j of the key-level public key (its first L limbs are this level's primes)
                const std::uint64_t *pk_j = pk.d_data() + j * pk.coeff_modulus_size() * n;
                detail::check(cat_dyadic(h, d_u.get(), pk_j, destination.d_data(j), nullptr, limbs, 1, nullptr));
                if (!is_ntt_form)
                    detail::check(cat_ntt_inverse(h, destination.d_data(j), nullptr, limbs, 1, 0, nullptr));
            }
            detail::DeviceWords d_e(h, L * n);
            const std::vector<std::uint64_t> t(L, context.parms().plain_modulus());
            for (std::size_t j = 0; j < 2; ++j)
            {
                sample_poly_cbd(prng, q, n, tmp.data());
                d_e.upload(tmp.data());
                if (is_ntt_form) // BGV transforms lazily and reduces in the scalar product (:259-264): same residues
                    detail::check(cat_ntt_forward(h, d_e.get(), nullptr, limbs, 1, 0, nullptr));
                if (bgv)
                    detail::check(cat_mul_scalar(h, d_e.get(), t.data(), d_e.get(), nullptr, limbs, 1, nullptr));
                detail::check(cat_add(h, d_e.get(), destination.d_data(j), destination.d_data(j), nullptr, limbs, 1, nullptr));
            }
            detail::check(cat_stream_sync(h, nullptr));
        }
    } // namespace util

    class KeyGenerator
    {
    public:
        // keygenerator.cpp:19-54
        explicit KeyGenerator(const SEALContext &context) : context_(context)
        {
            generate_sk();
        }
        KeyGenerator(const SEALContext &context, const SecretKey &secret_key) : context_(context), secret_key_(secret_key)
        {
            if (!secret_key.is_set())
                throw std::invalid_argument("secret key is not valid for encryption parameters");
        }
        const SecretKey &secret_key() const
        {
            return secret_key_;
        }

        // generate_pk, keygenerator.cpp:93-120
        void create_public_key(PublicKey &destination) const
        {
            util::encrypt_zero_symmetric(secret_key_, context_, context_.key_modulus_size(), true, false, destination.data());
        }

        // create_relin_keys, keygenerator.cpp:122-160, count = 1: the new key is s^2 (compute_secret_key_array :232-299)
        void create_relin_keys(RelinKeys &destination) const
        {
            cat_ctx *h = context_.handle();
            const std::size_t k = context_.key_modulus_size(), n = context_.poly_modulus_degree();
            detail::DeviceWords s2(h, k * n);
            detail::check(cat_dyadic(h, secret_key_.d_data(), secret_key_.d_data(), s2.get(), nullptr, (std::uint32_t)k, 1, nullptr));
            generate_one_kswitch_key(s2.get(), destination.allocate_key(RelinKeys::get_index(2)));
        }

        // create_galois_keys, keygenerator.cpp:162-221
        void create_galois_keys(const std::vector<std::uint32_t> &galois_elts, GaloisKeys &destination) const
        {
            cat_ctx *h = context_.handle();
            const std::size_t k = context_.key_modulus_size(), n = context_.poly_modulus_degree();
            detail::DeviceWords rotated(h, k * n);
            for (std::uint32_t galois_elt : galois_elts)
            {
                if (!(galois_elt & 1) || galois_elt >= (n << 1))
                    throw std::invalid_argument("Galois element is not valid");
                if (destination.has_key(galois_elt))
                    continue;
                detail::check(cat_apply_galois_ntt(h, secret_key_.d_data(), rotated.get(), galois_elt, (std::uint32_t)k, 1, nullptr));
                generate_one_kswitch_key(rotated.get(), destination.allocate_key(GaloisKeys::get_index(galois_elt)));
            }
        }
        // by rotation steps (keygenerator.h create_galois_keys(const vector<int> &steps, ...))
        void create_galois_keys(const std::vector<int> &steps, GaloisKeys &destination) const
        {
            std::vector<std::uint32_t> elts;
            for (int s : steps)
            {
                const std::uint32_t e = cat_galois_elt_from_step(cat_ctx_log_n(context_.handle()), s);
                if (!e)
                    throw std::invalid_argument("step count too large");
                elts.push_back(e);
            }
            create_galois_keys(elts, destination);
        }

    private:
        // keygenerator.cpp:56-91
        void generate_sk()
        {
            const std::size_t k = context_.key_modulus_size(), n = context_.poly_modulus_degree();
            std::vector<std::uint64_t> s(k * n);
            util::sample_poly_ternary(context_.parms().random_generator()->create(), context_.parms().coeff_modulus_words(), n, s.data());
            secret_key_.data_ = std::make_shared<detail::DeviceWords>(context_.handle(), s);
            detail::check(cat_ntt_forward(context_.handle(), secret_key_.data_->get(), nullptr, (std::uint32_t)k, 1, 0, nullptr));
            detail::check(cat_stream_sync(context_.handle(), nullptr));
        }

        // keygenerator.cpp:301-334: entry J is an encryption of zero at the key level whose c0 gets
        // (P mod q_J) * new_key added in limb J.  d_dest: [k-1][2][k][N]
        void generate_one_kswitch_key(const std::uint64_t *d_new_key, std::uint64_t *d_dest) const
        {
            if (!context_.using_keyswitching())
                throw std::logic_error("keyswitching is not supported by the context");
            cat_ctx *h = context_.handle();
            const std::size_t k = context_.key_modulus_size(), n = context_.poly_modulus_degree();
            const std::size_t decomp_mod_count = context_.first_level_size();
            const auto &key_modulus = context_.parms().coeff_modulus_words();
            detail::DeviceWords temp(h, n);
            for (std::size_t J = 0; J < decomp_mod_count; ++J)
            {
                std::uint64_t *c0 = d_dest + (J * 2) * k * n, *c1 = c0 + k * n;
                util::encrypt_zero_symmetric(secret_key_, context_, k, true, c0, c1);
                const std::uint64_t factor = key_modulus.back() % key_modulus[J];
                const std::uint32_t id = (std::uint32_t)J;
                detail::check(cat_mul_scalar(h, d_new_key + J * n, &factor, temp.get(), &id, 1, 1, nullptr));
                detail::check(cat_add(h, c0 + J * n, temp.get(), c0 + J * n, &id, 1, 1, nullptr));
            }
            detail::check(cat_stream_sync(h, nullptr));
        }

        SEALContext context_;
        SecretKey secret_key_;
    };

    class Encryptor
    {
    public:
        Encryptor(const SEALContext &context, const PublicKey &public_key) : context_(context)
        {
            set_public_key(public_key);
        }
        Encryptor(const SEALContext &context, const SecretKey &secret_key) : context_(context), secret_key_(secret_key)
        {}
        Encryptor(const SEALContext &context, const PublicKey &public_key, const SecretKey &secret_key)
            : context_(context), secret_key_(secret_key)
        {
            set_public_key(public_key);
        }
        void set_public_key(const PublicKey &public_key)
        {
            public_key_.data() = public_key.data();
            has_public_key_ = true;
        }
        void set_secret_key(const SecretKey &secret_key)
        {
            secret_key_ = secret_key;
        }

        // encryptor.h encrypt_zero(parms_id, destination): the level is named by its limb count
        void encrypt_zero(std::size_t coeff_modulus_size, Ciphertext &destination) const
        {
            encrypt_zero_internal(coeff_modulus_size, true, destination);
        }
        void encrypt_zero(Ciphertext &destination) const
        {
            encrypt_zero_internal(context_.first_level_size(), true, destination);
        }
        void encrypt_zero_symmetric(std::size_t coeff_modulus_size, Ciphertext &destination) const
        {
            encrypt_zero_internal(coeff_modulus_size, false, destination);
        }
        void encrypt_zero_symmetric(Ciphertext &destination) const
        {
            encrypt_zero_internal(context_.first_level_size(), false, destination);
        }

        // encrypt_internal, CKKS branch (encryptor.cpp:218-244): the NTT-form plaintext is added to c0
        void encrypt(const Plaintext &plain, Ciphertext &destination) const
        {
            encrypt_plain(plain, true, destination);
        }
        void encrypt_symmetric(const Plaintext &plain, Ciphertext &destination) const
        {
            encrypt_plain(plain, false, destination);
        }

    private:
        // encryptor.cpp:84-180
        void encrypt_zero_internal(std::size_t L, bool is_asymmetric, Ciphertext &destination) const
        {
            if (is_asymmetric && !has_public_key_)
                throw std::logic_error("public key is not set");
            if (!is_asymmetric && !secret_key_.is_set())
                throw std::logic_error("secret key is not set");
            const scheme_type scheme = context_.parms().scheme();
            bool is_ntt_form = false;
            if (scheme == scheme_type::ckks || scheme == scheme_type::bgv)
                is_ntt_form = true;
            else if (scheme != scheme_type::bfv)
                throw std::invalid_argument("unsupported scheme");
            if (L == 0 || L > context_.key_modulus_size())
                throw std::invalid_argument("parms_id is not valid for encryption parameters");
            if (!is_asymmetric)
            {
                util::encrypt_zero_symmetric(secret_key_, context_, L, is_ntt_form, false, destination);
                return;
            }
            if (L == context_.key_modulus_size())
            {
                util::encrypt_zero_asymmetric(public_key_, context_, L, is_ntt_form, destination);
                return;
            }
            // a level below the key level: encrypt one level up, then switch the modulus down (:124-166)
            Ciphertext temp;
            util::encrypt_zero_asymmetric(public_key_, context_, L + 1, is_ntt_form, temp);
            destination.resize(context_, L, 2);
            detail::check(cat_rescale(context_.handle(), temp.d_data(), destination.d_data(), (std::uint32_t)(L + 1), 2, 1, nullptr));
            detail::check(cat_stream_sync(context_.handle(), nullptr));
            destination.is_ntt_form() = is_ntt_form;
            destination.scale() = temp.scale();
            destination.correction_factor() = temp.correction_factor();
        }
        void encrypt_plain(const Plaintext &plain, bool is_asymmetric, Ciphertext &destination) const
        {
            const scheme_type scheme = context_.parms().scheme();
            if (scheme == scheme_type::bfv || scheme == scheme_type::bgv)
            {
                // encryptor.cpp:205-217 (BFV) and :245-318 (BGV): a fresh encryption of zero at the first level whose c0
                // gets round(q m / t), or the transform of the lifted plaintext
                if (plain.is_ntt_form())
                    throw std::invalid_argument("plain cannot be in NTT form");
                const std::size_t L = context_.first_level_size(), n = context_.poly_modulus_degree();
                const std::uint64_t t = context_.parms().plain_modulus();
                if (plain.coeff_count() > n)
                    throw std::invalid_argument("plain is not valid for encryption parameters");
                for (std::uint64_t c : plain.data())
                    if (c >= t)
                        throw std::invalid_argument("plain is not valid for encryption parameters");
                encrypt_zero_internal(L, is_asymmetric, destination);
                const detail::PlainTools tools(detail::level_primes(context_, L), t, 0);
                std::vector<std::uint64_t> add(L * n);
                if (scheme == scheme_type::bfv)
                    tools.scale_plain(plain.data(), n, add.data());
                else
                    tools.lift_plain(plain.data(), n, add.data());
                detail::DeviceWords d_add(context_.handle(), add);
                if (scheme == scheme_type::bgv)
                    detail::check(cat_ntt_forward(context_.handle(), d_add.get(), nullptr, (std::uint32_t)L, 1, 0, nullptr));
                detail::check(cat_add(context_.handle(), destination.d_data(0), d_add.get(), destination.d_data(0), nullptr,
                                      (std::uint32_t)L, 1, nullptr));
                detail::check(cat_stream_sync(context_.handle(), nullptr));
                return;
            }
            if (!plain.is_ntt_form())
                throw std::invalid_argument("plain must be in NTT form");
            const std::size_t L = plain.coeff_modulus_size();
            encrypt_zero_internal(L, is_asymmetric, destination);
            detail::check(cat_add(context_.handle(), destination.d_data(0), plain.d_data(), destination.d_data(0), nullptr,
                                  (std::uint32_t)L, 1, nullptr));
            detail::check(cat_stream_sync(context_.handle(), nullptr));
            destination.scale() = plain.scale();
        }

        SEALContext context_;
        PublicKey public_key_;
        bool has_public_key_ = false;
        SecretKey secret_key_;
    };

    // Evaluator::add_plain_inplace / sub_plain_inplace, BFV and BGV (evaluator.cpp:1737-1951).  BFV: c0 +- round(q m / t)
    // at the ciphertext's level (multiply_add / multiply_sub_plain_with_scaling_variant, util/scalingvariant.cpp:69-165);
    // BGV: m times the ciphertext's correction factor modulo t, lifted and transformed
    // (transform_to_ntt_inplace(Plaintext &), evaluator.cpp:2174-2265), added to or subtracted from c0.
    inline void Evaluator::plain_addsub_coeff(Ciphertext &encrypted, const Plaintext &plain, bool sub) const
    {
        const scheme_type scheme = context_.parms().scheme();
        const bool bfv = scheme == scheme_type::bfv;
        if (!bfv && scheme != scheme_type::bgv)
            throw std::invalid_argument("unsupported scheme");
        if (bfv ? encrypted.is_ntt_form() : !encrypted.is_ntt_form())
            throw std::invalid_argument(bfv ? "BFV encrypted cannot be in NTT form" : "BGV encrypted must be in NTT form");
        if (plain.is_ntt_form())
            throw std::invalid_argument(bfv ? "BFV plain cannot be in NTT form" : "BGV plain cannot be in NTT form");
        const std::size_t L = encrypted.coeff_modulus_size(), n = context_.poly_modulus_degree();
        const std::uint64_t t = context_.parms().plain_modulus();
        if (plain.coeff_count() > n)
            throw std::invalid_argument("plain is not valid for encryption parameters");
        for (std::uint64_t c : plain.data())
            if (c >= t)
                throw std::invalid_argument("plain is not valid for encryption parameters");
        const detail::PlainTools tools(detail::level_primes(context_, L), t, 0);
        std::vector<std::uint64_t> add(L * n);
        if (bfv)
            tools.scale_plain(plain.data(), n, add.data());
        else
        {
            std::vector<std::uint64_t> scaled(plain.data());
            for (std::uint64_t &c : scaled)
                c = mul_mod_t(c, encrypted.correction_factor());
            tools.lift_plain(scaled, n, add.data());
        }
        detail::DeviceWords d_add(context_.handle(), add);
        if (!bfv)
            detail::check(cat_ntt_forward(context_.handle(), d_add.get(), nullptr, (std::uint32_t)L, 1, 0, nullptr));
        detail::check((sub ? cat_sub : cat_add)(context_.handle(), encrypted.d_data(0), d_add.get(), encrypted.d_data(0), nullptr,
                                                (std::uint32_t)L, 1, nullptr));
        detail::check(cat_stream_sync(context_.handle(), nullptr));
    }

    // Evaluator::multiply_plain_inplace with a coefficient-form plaintext modulo t (evaluator.cpp:1953-2133): the
    // plaintext is lifted (coefficients at or above (t + 1) / 2 stand for negatives) and transformed once; an NTT-form
    // ciphertext is multiplied as it is, a coefficient-form one through forward transform, product, inverse transform
    // per polynomial.  (The reference's monomial shortcut gives the same words: both are exact modulo q.)
    inline void Evaluator::multiply_plain_coeff(Ciphertext &encrypted, const Plaintext &plain) const
    {
        const std::size_t L = encrypted.coeff_modulus_size(), n = context_.poly_modulus_degree();
        const std::uint64_t t = context_.parms().plain_modulus();
        if (context_.parms().scheme() == scheme_type::ckks || t == 0)
            throw std::invalid_argument("plain is not valid for encryption parameters");
        if (plain.coeff_count() > n || plain.coeff_count() == 0)
            throw std::invalid_argument("plain is not valid for encryption parameters");
        for (std::uint64_t c : plain.data())
            if (c >= t)
                throw std::invalid_argument("plain is not valid for encryption parameters");
        const detail::PlainTools tools(detail::level_primes(context_, L), t, 0);
        std::vector<std::uint64_t> lifted(L * n);
        tools.lift_plain(plain.data(), n, lifted.data());
        detail::DeviceWords d_plain(context_.handle(), lifted);
        cat_ctx *c = context_.handle();
        const std::uint32_t size = (std::uint32_t)encrypted.size();
        detail::check(cat_ntt_forward(c, d_plain.get(), nullptr, (std::uint32_t)L, 1, 0, nullptr));
        const bool coeff_form = !encrypted.is_ntt_form();
        if (coeff_form)
            detail::check(cat_ntt_forward(c, encrypted.d_data(), nullptr, (std::uint32_t)L, size, 0, nullptr));
        detail::check(cat_multiply_plain_ntt(c, encrypted.d_data(), d_plain.get(), 0, encrypted.d_data(), (std::uint32_t)L, size, 1,
                                             nullptr));
        if (coeff_form)
            detail::check(cat_ntt_inverse(c, encrypted.d_data(), nullptr, (std::uint32_t)L, size, 0, nullptr));
        detail::check(cat_stream_sync(c, nullptr));
    }

    inline void Evaluator::transform_to_ntt_inplace(Plaintext &plain, const parms_id_type &parms_id) const
    {
        const std::size_t L = context_.level_of(parms_id), n = context_.poly_modulus_degree();
        const std::uint64_t t = context_.parms().plain_modulus();
        if (!L)
            throw std::invalid_argument("parms_id is not valid for the current context");
        if (plain.is_ntt_form())
            throw std::invalid_argument("plain is already in NTT form");
        if (context_.parms().scheme() == scheme_type::ckks || t == 0 || plain.coeff_count() > n)
            throw std::invalid_argument("plain is not valid for encryption parameters");
        for (std::uint64_t c : plain.data())
            if (c >= t)
                throw std::invalid_argument("plain is not valid for encryption parameters");
        const detail::PlainTools tools(detail::level_primes(context_, L), t, 0);
        std::vector<std::uint64_t> lifted(L * n);
        tools.lift_plain(plain.data(), n, lifted.data());
        Plaintext dest(context_, L);
        dest.upload(lifted.data());
        detail::check(cat_ntt_forward(context_.handle(), dest.d_data(), nullptr, (std::uint32_t)L, 1, 0, nullptr));
        detail::check(cat_stream_sync(context_.handle(), nullptr));
        plain = std::move(dest);
    }

    class Decryptor
    {
    public:
        Decryptor(const SEALContext &context, const SecretKey &secret_key) : context_(context), secret_key_(secret_key)
        {
            if (!secret_key.is_set())
                throw std::invalid_argument("secret key is not valid for encryption parameters");
        }

        // dot_product_ct_sk_array (decryptor.cu:544-626): d_destination (L x N device words) = c0 + c1 s + c2 s^2 + ...
        // in the form (NTT or coefficient) of the ciphertext.  For BFV and BGV this phase is what
        // RNSTool::decrypt_scale_and_round / decrypt_modt would finish.
        void dot_product_ct_sk_array(const Ciphertext &encrypted, std::uint64_t *d_destination)
        {
            cat_ctx *h = context_.handle();
            const std::size_t n = context_.poly_modulus_degree(), k = context_.key_modulus_size();
            const std::size_t L = encrypted.coeff_modulus_size(), size = encrypted.size();
            if (!encrypted.d_data() || size < 2 || L == 0 || L > k)
                throw std::invalid_argument("encrypted is not valid for encryption parameters");
            const std::uint32_t limbs = (std::uint32_t)L;
            compute_secret_key_array(size - 1);
            // c_1 ... c_{size-1} in NTT form (the reference transforms lazily, :582 and :603: same residues)
            detail::DeviceWords copy(h, (size - 1) * L * n);
            std::vector<std::uint64_t> host = encrypted.to_cpu();
            copy.upload(host.data() + L * n);
            if (!encrypted.is_ntt_form())
                detail::check(cat_ntt_forward(h, copy.get(), nullptr, limbs, (std::uint32_t)(size - 1), 0, nullptr));
            for (std::size_t p = 0; p + 1 < size; ++p)
            {
                std::uint64_t *c = copy.get() + p * L * n;
                detail::check(cat_dyadic(h, c, powers_->get() + p * k * n, c, nullptr, limbs, 1, nullptr));
                if (p)
                    detail::check(cat_add(h, copy.get(), c, copy.get(), nullptr, limbs, 1, nullptr));
            }
            if (!encrypted.is_ntt_form())
                detail::check(cat_ntt_inverse(h, copy.get(), nullptr, limbs, 1, 0, nullptr));
            detail::check(cat_add(h, copy.get(), encrypted.d_data(), d_destination, nullptr, limbs, 1, nullptr));
            detail::check(cat_stream_sync(h, nullptr));
        }

        // decrypt -> ckks_decrypt (decryptor.cu:628-670): the plaintext is the phase, NTT form, at the level and scale
        // of the ciphertext
        void decrypt(const Ciphertext &encrypted, Plaintext &destination)
        {
            const scheme_type scheme = context_.parms().scheme();
            if (scheme == scheme_type::bfv || scheme == scheme_type::bgv)
            {
                // bfv_decrypt (decryptor.cu:672-730) / bgv_decrypt (:103-150): the phase on the device, its RNS tail
                // (decrypt_scale_and_round / decrypt_modt and the correction factor) on the host
                const bool bfv = scheme == scheme_type::bfv;
                if (bfv && encrypted.is_ntt_form())
                    throw std::invalid_argument("encrypted cannot be in NTT form");
                if (!bfv && !encrypted.is_ntt_form())
                    throw std::invalid_argument("encrypted must be in NTT form");
                const std::size_t L = encrypted.coeff_modulus_size(), n = context_.poly_modulus_degree();
                const std::uint64_t t = context_.parms().plain_modulus();
                detail::DeviceWords d_phase(context_.handle(), L * n);
                dot_product_ct_sk_array(encrypted, d_phase.get());
                if (!bfv)
                    detail::check(cat_ntt_inverse(context_.handle(), d_phase.get(), nullptr, (std::uint32_t)L, 1, 0, nullptr));
                const std::vector<std::uint64_t> phase = d_phase.to_cpu();
                // gamma = the second auxiliary prime of the BEHZ tool (util/rns.cpp:625-631); a BFV context lists its
                // auxiliary primes after the coefficient primes
                const std::uint64_t gamma = bfv ? cat_ctx_modulus(context_.handle(), (std::uint32_t)context_.key_modulus_size() + 1) : 0;
                const detail::PlainTools tools(detail::level_primes(context_, L), t, gamma);
                std::vector<std::uint64_t> m(n);
                if (bfv)
                    tools.decrypt_scale_and_round(phase.data(), n, m.data());
                else
                {
                    tools.decrypt_modt(phase.data(), n, m.data());
                    if (encrypted.correction_factor() != 1)
                    {
                        std::uint64_t fix = 1;
                        if (!detail::try_invert_mod(encrypted.correction_factor(), t, fix))
                            throw std::logic_error("invalid correction factor");
                        for (auto &c : m)
                            c = detail::mul_mod(c, fix, t);
                    }
                }
                // the plaintext keeps its significant coefficients (decryptor.cu:724-728)
                std::size_t count = n;
                while (count > 1 && m[count - 1] == 0)
                    --count;
                m.resize(count);
                destination = Plaintext(std::move(m));
                return;
            }
            if (!encrypted.is_ntt_form())
                throw std::invalid_argument("encrypted must be in NTT form");
            if (destination.coeff_modulus_size() != encrypted.coeff_modulus_size())
                destination = Plaintext(context_, encrypted.coeff_modulus_size());
            dot_product_ct_sk_array(encrypted, destination.d_data());
            destination.is_ntt_form() = true;
            destination.scale() = encrypted.scale();
        }

    private:
        // decryptor.cpp:160-223: powers s, s^2, ... of the secret key, [power][k][N], NTT form
        void compute_secret_key_array(std::size_t max_power)
        {
            if (max_power <= powers_count_)
                return;
            cat_ctx *h = context_.handle();
            const std::size_t n = context_.poly_modulus_degree(), k = context_.key_modulus_size();
            auto next = std::make_shared<detail::DeviceWords>(h, max_power * k * n);
            std::vector<std::uint64_t> s = secret_key_.to_cpu();
            detail::check(cat_upload(h, next->get(), s.data(), s.size() * 8, nullptr));
            detail::check(cat_stream_sync(h, nullptr));
            for (std::size_t p = 1; p < max_power; ++p)
                detail::check(cat_dyadic(h, next->get() + (p - 1) * k * n, next->get(), next->get() + p * k * n, nullptr,
                                         (std::uint32_t)k, 1, nullptr));
            detail::check(cat_stream_sync(h, nullptr));
            powers_ = std::move(next);
            powers_count_ = max_power;
        }

        SEALContext context_;
        SecretKey secret_key_;
        std::shared_ptr<detail::DeviceWords> powers_;
        std::size_t powers_count_ = 0;
    };
} // namespace seal
