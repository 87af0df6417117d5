// seal_hip.hpp -- C++ host facade over the C ABI (include/cat_amd.h) that keeps the reference's class and
// function names for the hot path, so code written against Rayman96/CAT's seal::Evaluator /
// seal::util::ntt_negacyclic_harvey keeps its shape.  Header-only; links against libcat_amd.so only.
//
// What is mirrored (paths relative to /root/reference/native/src/seal/):
//   seal::util::ntt_negacyclic_harvey[_lazy], inverse_ntt_negacyclic_harvey[_lazy]   util/ntt.h:201-270
//   seal::util::{add,sub,negate}_poly_coeffmod, dyadic_product_coeffmod               util/polyarithsmallmod.cuh:84-611
//   seal::Ciphertext (size/coeff_modulus_size/is_ntt_form/scale, to_gpu/to_cpu)        ciphertext.cuh:356-385, 496-735
//   seal::KSwitchKeys / RelinKeys / GaloisKeys (get_index, has_key)                   kswitchkeys.h:36-131, relinkeys.h:58-64, galoiskeys.h:48-50
//   seal::Evaluator::{add,sub,negate,multiply,square,multiply_plain,add_plain,sub_plain,relinearize,rescale_to_next,
//                     mod_switch_to_next,transform_{to,from}_ntt,apply_galois,rotate_vector,rotate_rows,
//                     complex_conjugate}_inplace
//                                                                                      evaluator.h:88-1298
// Differences that are deliberate: data lives on the device (the reference keeps a host array plus a
// device shadow; here upload()/download() are the to_gpu()/to_cpu() of ciphertext.cuh:361-381); key
// generation, encoding and encryption stay with the caller; errors are
// the reference's exception types with the reference's messages, never exit().
#pragma once

#include "../../include/cat_amd.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <initializer_list>
#include <limits>
#include <map>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "seal_hip_modulus.hpp"
#include "seal_hip_serialization.hpp" // brings seal_hip_random.hpp

namespace seal
{
    enum class scheme_type : std::uint8_t
    {
        none = 0x0,
        bfv = 0x1,
        ckks = 0x2,
        bgv = 0x3
    };

    namespace detail
    {
        // cat_status -> the exception the reference would have thrown (evaluator.cpp:2534-2573)
        inline void check(cat_status st)
        {
            switch (st)
            {
            case CAT_OK:
                return;
            case CAT_ERR_INVALID_ARG:
                throw std::invalid_argument(cat_last_error());
            case CAT_ERR_LOGIC:
                throw std::logic_error(cat_last_error());
            case CAT_ERR_OUT_OF_RANGE:
                throw std::out_of_range(cat_last_error());
            case CAT_ERR_NOMEM:
                throw std::bad_alloc();
            default:
                throw std::runtime_error(cat_last_error());
            }
        }
        // util::are_close<double> (util/common.h:568-574), what are_same_scale compares with (evaluator.cpp:24-26)
        inline bool are_close(double a, double b) noexcept
        {
            const double scale = std::max(std::max(std::fabs(a), std::fabs(b)), 1.0);
            return std::fabs(a - b) < std::numeric_limits<double>::epsilon() * scale;
        }
    } // namespace detail

    class EncryptionParameters
    {
    public:
        explicit EncryptionParameters(scheme_type scheme) : scheme_(scheme)
        {}
        void set_poly_modulus_degree(std::size_t n)
        {
            n_ = n;
        }
        // encryptionparams.cuh set_coeff_modulus(const std::vector<Modulus> &); the facade also takes plain words
        void set_coeff_modulus(const std::vector<Modulus> &q)
        {
            qm_ = q;
            q_.clear();
            for (const Modulus &m : q)
                q_.push_back(m.value());
        }
        void set_coeff_modulus(const std::vector<std::uint64_t> &q)
        {
            std::vector<Modulus> m;
            for (std::uint64_t v : q)
                m.emplace_back(v); // throws like Modulus::set_value on values above 61 bits
            set_coeff_modulus(m);
        }
        void set_coeff_modulus(std::initializer_list<std::uint64_t> q)
        {
            set_coeff_modulus(std::vector<std::uint64_t>(q));
        }
        void set_plain_modulus(const Modulus &t)
        {
            t_ = t;
        }
        void set_plain_modulus(std::uint64_t t)
        {
            t_ = Modulus(t);
        }
        scheme_type scheme() const noexcept
        {
            return scheme_;
        }
        std::size_t poly_modulus_degree() const noexcept
        {
            return n_;
        }
        const std::vector<Modulus> &coeff_modulus() const noexcept
        {
            return qm_;
        }
        // the primes as plain words, the form the C ABI takes (not in the reference)
        const std::vector<std::uint64_t> &coeff_modulus_words() const noexcept
        {
            return q_;
        }
        const Modulus &plain_modulus() const noexcept
        {
            return t_;
        }
        // encryptionparams.cuh set_random_generator / random_generator: nullptr = the default factory
        void set_random_generator(std::shared_ptr<UniformRandomGeneratorFactory> random_generator) noexcept
        {
            random_generator_ = std::move(random_generator);
        }
        std::shared_ptr<UniformRandomGeneratorFactory> random_generator() const
        {
            return random_generator_ ? random_generator_ : UniformRandomGeneratorFactory::DefaultFactory();
        }

    private:
        scheme_type scheme_;
        std::size_t n_ = 0;
        std::vector<std::uint64_t> q_;
        std::vector<Modulus> qm_;
        Modulus t_;
        std::shared_ptr<UniformRandomGeneratorFactory> random_generator_;
    };

    // SEALContext for this path: the modulus chain, per-prime NTT tables and per-level RNS tools live in the
    // device context (context.cu:292-356, 607-765).  A level is named by its number of limbs L: the key
    // level has all k primes, the first data level k-1 (or 1 when k == 1), each next level one fewer.
    class SEALContext
    {
    public:
        explicit SEALContext(const EncryptionParameters &parms, int device = 0) : parms_(parms)
        {
            const std::size_t n = parms.poly_modulus_degree();
            std::uint32_t log_n = 0;
            while ((std::size_t(1) << log_n) < n)
                ++log_n;
            if (n == 0 || (std::size_t(1) << log_n) != n)
                throw std::invalid_argument("poly_modulus_degree is invalid");
            cat_ctx *h = nullptr;
            detail::check(cat_ctx_create(log_n, parms.coeff_modulus_words().data(),
                                         (std::uint32_t)parms.coeff_modulus_words().size(), (int)parms.scheme(),
                                         parms.plain_modulus().value(), device, &h));
            h_.reset(h, [](cat_ctx *p) { cat_ctx_destroy(p); });
            // temporaries of every operation on one ciphertext at any level, allocated once here so that no Evaluator
            // call ever synchronises or allocates on its way (cat_ctx_workspace_need; larger batches through the C ABI
            // grow the block on first use, or call cat_ctx_reserve themselves)
            detail::check(cat_ctx_reserve(h, cat_ctx_workspace_need(h, 1)));
            // the reference's device pool (gpu_memorypool.h): blocks of resized / destroyed ciphertexts are kept for the
            // next one of that size instead of hipFree (a device-wide synchronisation) + hipMalloc per Evaluator call.
            // Safe here because every call of the facade goes to the default stream (cat_ctx_set_pool's contract).
            detail::check(cat_ctx_set_pool(h, std::size_t(1) << 30));
        }
        cat_ctx *handle() const noexcept
        {
            return h_.get();
        }
        const EncryptionParameters &parms() const noexcept
        {
            return parms_;
        }
        std::size_t poly_modulus_degree() const noexcept
        {
            return parms_.poly_modulus_degree();
        }
        std::size_t key_modulus_size() const noexcept
        {
            return parms_.coeff_modulus_words().size();
        }
        std::size_t first_level_size() const noexcept
        {
            const std::size_t k = key_modulus_size();
            return k > 1 ? k - 1 : k;
        }
        bool using_keyswitching() const noexcept
        {
            return key_modulus_size() > 1;
        }
        // parms_id of the level with L limbs (the hash of that level's EncryptionParameters, encryptionparams.cpp:124-158;
        // chain: context.cu:607-637, 726-765) and the way back; 0 = no such level
        parms_id_type parms_id(std::size_t coeff_modulus_size) const
        {
            if (coeff_modulus_size == 0 || coeff_modulus_size > key_modulus_size())
                throw std::invalid_argument("parms_id is not valid for encryption parameters");
            return util::compute_parms_id((std::uint8_t)parms_.scheme(), parms_.poly_modulus_degree(),
                                          parms_.coeff_modulus_words().data(), coeff_modulus_size,
                                          parms_.plain_modulus().value());
        }
        parms_id_type key_parms_id() const
        {
            return parms_id(key_modulus_size());
        }
        parms_id_type first_parms_id() const
        {
            return parms_id(first_level_size());
        }
        std::size_t level_of(const parms_id_type &id) const
        {
            for (std::size_t L = key_modulus_size(); L >= 1; --L)
                if (parms_id(L) == id)
                    return L;
            return 0;
        }

    private:
        EncryptionParameters parms_;
        std::shared_ptr<cat_ctx> h_;
    };

    // Device-resident ciphertext: size x coeff_modulus_size x N words, poly[limb][coeff] (ciphertext.cuh:423-434).
    class Ciphertext
    {
    public:
        Ciphertext() = default;
        Ciphertext(const SEALContext &context, std::size_t coeff_modulus_size, std::size_t size = 2)
        {
            resize(context, coeff_modulus_size, size);
        }
        // ciphertext.cuh:60-120: Ciphertext(context, parms_id [, size_capacity]) -- an empty ciphertext of that level
        Ciphertext(const SEALContext &context, const parms_id_type &parms_id, std::size_t size = 2)
        {
            resize(context, parms_id, size);
        }
        Ciphertext(const Ciphertext &o)
        {
            *this = o;
        }
        Ciphertext &operator=(const Ciphertext &o)
        {
            if (this == &o)
                return *this;
            ctx_ = o.ctx_;
            n_ = o.n_;
            is_ntt_form_ = o.is_ntt_form_;
            scale_ = o.scale_;
            correction_factor_ = o.correction_factor_;
            reserve_words(o.words());
            size_ = o.size_;
            L_ = o.L_;
            if (o.words())
            {
                // device to device on the default stream, in order with the work already queued on either buffer
                if (cat_copy(ctx_->handle(), d_, o.d_, o.words() * sizeof(std::uint64_t), nullptr) != CAT_OK)
                    throw std::runtime_error(cat_last_error());
            }
            return *this;
        }
        Ciphertext(Ciphertext &&o) noexcept
        {
            *this = std::move(o);
        }
        Ciphertext &operator=(Ciphertext &&o) noexcept
        {
            if (this != &o)
            {
                release();
                ctx_ = std::move(o.ctx_);
                d_ = o.d_;
                cap_ = o.cap_;
                size_ = o.size_;
                L_ = o.L_;
                n_ = o.n_;
                is_ntt_form_ = o.is_ntt_form_;
                scale_ = o.scale_;
                correction_factor_ = o.correction_factor_;
                o.d_ = nullptr;
                o.cap_ = o.size_ = o.L_ = 0;
            }
            return *this;
        }
        ~Ciphertext()
        {
            release();
        }

        // Ciphertext::resize(context, parms_id, size) (ciphertext.cpp:76-107); contents are kept when the
        // limb count is unchanged, exactly like the reference's resize of the backing array.
        void resize(const SEALContext &context, std::size_t coeff_modulus_size, std::size_t size)
        {
            if (size < 2 && size != 0)
                throw std::invalid_argument("invalid size");
            if (coeff_modulus_size == 0 || coeff_modulus_size > context.key_modulus_size())
                throw std::invalid_argument("parms_id is not valid for encryption parameters");
            ctx_ = std::make_shared<SEALContext>(context);
            n_ = context.poly_modulus_degree();
            const std::size_t want = size * coeff_modulus_size * n_;
            if (want > cap_)
            {
                std::uint64_t *nd = nullptr;
                detail::check(cat_malloc(ctx_->handle(), want * sizeof(std::uint64_t), (void **)&nd));
                if (d_)
                {
                    if (L_ == coeff_modulus_size && words())
                    {
                        std::vector<std::uint64_t> tmp(words());
                        download(tmp.data());
                        detail::check(cat_upload(ctx_->handle(), nd, tmp.data(), tmp.size() * 8, nullptr));
                        detail::check(cat_stream_sync(ctx_->handle(), nullptr));
                    }
                    cat_free(ctx_->handle(), d_);
                }
                d_ = nd;
                cap_ = want;
            }
            size_ = size;
            L_ = coeff_modulus_size;
        }
        void resize(const SEALContext &context, const parms_id_type &parms_id, std::size_t size)
        {
            const std::size_t L = context.level_of(parms_id);
            if (!L)
                throw std::invalid_argument("parms_id is not valid for encryption parameters");
            resize(context, L, size);
        }
        void resize(std::size_t size)
        {
            if (!ctx_)
                throw std::logic_error("ciphertext is not bound to a context");
            resize(*ctx_, L_, size);
        }

        // to_gpu() / to_cpu() (ciphertext.cuh:361-381)
        void upload(const std::uint64_t *host)
        {
            detail::check(cat_upload(ctx_->handle(), d_, host, words() * 8, nullptr));
            detail::check(cat_stream_sync(ctx_->handle(), nullptr));
        }
        void download(std::uint64_t *host) const
        {
            detail::check(cat_download(ctx_->handle(), host, d_, words() * 8, nullptr));
            detail::check(cat_stream_sync(ctx_->handle(), nullptr));
        }
        std::vector<std::uint64_t> to_cpu() const
        {
            std::vector<std::uint64_t> out(words());
            if (!out.empty())
                download(out.data());
            return out;
        }

        std::uint64_t *d_data() noexcept
        {
            return d_;
        }
        const std::uint64_t *d_data() const noexcept
        {
            return d_;
        }
        std::uint64_t *d_data(std::size_t poly_index) noexcept
        {
            return d_ + poly_index * L_ * n_;
        }
        std::size_t size() const noexcept
        {
            return size_;
        }
        std::size_t coeff_modulus_size() const noexcept
        {
            return L_;
        }
        std::size_t poly_modulus_degree() const noexcept
        {
            return n_;
        }
        bool is_ntt_form() const noexcept
        {
            return is_ntt_form_;
        }
        bool &is_ntt_form() noexcept
        {
            return is_ntt_form_;
        }
        double scale() const noexcept
        {
            return scale_;
        }
        double &scale() noexcept
        {
            return scale_;
        }
        // BGV only (ciphertext.h correction_factor): decryption divides the message by it modulo t
        std::uint64_t correction_factor() const noexcept
        {
            return correction_factor_;
        }
        std::uint64_t &correction_factor() noexcept
        {
            return correction_factor_;
        }
        std::size_t words() const noexcept
        {
            return size_ * L_ * n_;
        }
        const SEALContext *context() const noexcept
        {
            return ctx_.get();
        }
        parms_id_type parms_id() const
        {
            return ctx_ && L_ ? ctx_->parms_id(L_) : parms_id_zero;
        }

        // Ciphertext::save_members / save_size (ciphertext.cpp:188-286), uncompressed: parms_id, NTT flag, size, N, limbs,
        // scale, correction factor, then the data as a DynArray -- or, when the second polynomial starts with the seed
        // marker (util/rlwe.cpp:404-411), the first polynomial and the generator info
        std::streamoff save_size(bool seeded = false) const
        {
            const std::size_t members = sizeof(parms_id_type) + 1 + 4 * 8 + 8;
            return static_cast<std::streamoff>(sizeof(Serialization::SEALHeader) + members) +
                   (seeded ? util::dynarray_save_size(words() / 2) + UniformRandomGeneratorInfo::SaveSize()
                           : util::dynarray_save_size(words()));
        }
        std::streamoff save(std::ostream &stream) const
        {
            const std::vector<std::uint64_t> host = to_cpu();
            const bool seeded = size_ == 2 && !host.empty() && host[L_ * n_] == 0xFFFFFFFFFFFFFFFFULL;
            return Serialization::Save(
                [&](std::ostream &s) {
                    util::write_pod(s, parms_id());
                    util::write_pod<std::uint8_t>(s, is_ntt_form_ ? 1 : 0);
                    util::write_pod<std::uint64_t>(s, size_);
                    util::write_pod<std::uint64_t>(s, n_);
                    util::write_pod<std::uint64_t>(s, L_);
                    util::write_pod(s, scale_);
                    util::write_pod(s, correction_factor_);
                    if (seeded)
                    {
                        util::dynarray_save(s, host.data(), host.size() / 2);
                        // the info was stored as a serialized object after the marker word: copy it through
                        s.write(reinterpret_cast<const char *>(host.data() + L_ * n_ + 1),
                                static_cast<std::streamsize>(UniformRandomGeneratorInfo::SaveSize()));
                    }
                    else
                        util::dynarray_save(s, host.data(), host.size());
                },
                save_size(seeded), stream);
        }
        // Ciphertext::load_members (ciphertext.cpp:288-400): metadata checked against the context, the data bounded by
        // the size the metadata promises, a seeded ciphertext expanded with the generator it names (:153-186)
        std::streamoff load(const SEALContext &context, std::istream &stream)
        {
            Ciphertext fresh;
            const std::streamoff in_size = Serialization::Load(
                [&](std::istream &s, SEALVersion version) {
                    const parms_id_type id = util::read_pod<parms_id_type>(s);
                    const std::uint8_t ntt = util::read_pod<std::uint8_t>(s);
                    const std::uint64_t size64 = util::read_pod<std::uint64_t>(s);
                    const std::uint64_t n64 = util::read_pod<std::uint64_t>(s);
                    const std::uint64_t L64 = util::read_pod<std::uint64_t>(s);
                    const double scale = util::read_pod<double>(s);
                    std::uint64_t correction_factor = 1;
                    if (version.major == 4)
                        correction_factor = util::read_pod<std::uint64_t>(s);
                    // is_metadata_valid_for(..., allow_pure_key_levels = true), valcheck.cpp
                    const std::size_t L = context.level_of(id);
                    if (!L || n64 != context.poly_modulus_degree() || L64 != L || size64 < 2 || size64 > 16) // SEAL_CIPHERTEXT_SIZE_MIN .. _MAX, util/defines.h:66-70
                        throw std::logic_error("ciphertext data is invalid");
                    const std::size_t total = (std::size_t)size64 * L * (std::size_t)n64;
                    std::vector<std::uint64_t> data = util::dynarray_load(s, total);
                    if (data.size() == L * n64 && size64 == 2)
                    {
                        UniformRandomGeneratorInfo info;
                        if (version.major == 4 || (version.major == 3 && version.minor >= 6))
                            info.load(s);
                        else
                            throw std::logic_error("incompatible version"); // the 3.4 / 3.5 samplers are not restated
                        auto prng = info.make_prng();
                        if (!prng)
                            throw std::logic_error("unsupported prng_type");
                        data.resize(total);
                        const auto &q = context.parms().coeff_modulus_words();
                        util::sample_poly_uniform(prng, std::vector<std::uint64_t>(q.begin(), q.begin() + (std::ptrdiff_t)L),
                                                  (std::size_t)n64, data.data() + L * n64);
                    }
                    if (data.size() != total) // is_buffer_valid
                        throw std::logic_error("ciphertext data is invalid");
                    fresh.resize(context, L, (std::size_t)size64);
                    fresh.upload(data.data());
                    fresh.is_ntt_form_ = ntt != 0;
                    fresh.scale_ = scale;
                    fresh.correction_factor_ = correction_factor;
                },
                stream);
            *this = std::move(fresh);
            return in_size;
        }

    private:
        void reserve_words(std::size_t want)
        {
            if (want > cap_)
            {
                release_buffer();
                detail::check(cat_malloc(ctx_->handle(), want * sizeof(std::uint64_t), (void **)&d_));
                cap_ = want;
            }
        }
        void release_buffer()
        {
            if (d_ && ctx_)
                cat_free(ctx_->handle(), d_);
            d_ = nullptr;
            cap_ = 0;
        }
        void release()
        {
            release_buffer();
            size_ = L_ = 0;
        }

        std::shared_ptr<SEALContext> ctx_;
        std::uint64_t *d_ = nullptr;
        std::size_t cap_ = 0, size_ = 0, L_ = 0, n_ = 0;
        bool is_ntt_form_ = false;
        double scale_ = 1.0;
        std::uint64_t correction_factor_ = 1;
    };

    // Device-resident NTT-form CKKS plaintext: coeff_modulus_size x N words (plaintext.h; produced by CKKSEncoder on
    // the host, which stays with the caller).
    class Plaintext
    {
    public:
        Plaintext(const SEALContext &context, std::size_t coeff_modulus_size) : buf_(context, coeff_modulus_size, 2)
        {
            // storage is borrowed from Ciphertext (2 polys allocated, the first one used)
            buf_.is_ntt_form() = true;
        }
        // A BFV / BGV plaintext (plaintext.h): up to N coefficients modulo t on the HOST, parms_id zero, not NTT form.
        // It reaches the device inside Encryptor::encrypt and comes back from Decryptor::decrypt.
        Plaintext() = default;
        explicit Plaintext(std::vector<std::uint64_t> coeffs) : coeffs_(std::move(coeffs))
        {}
        std::size_t coeff_count() const noexcept
        {
            return coeffs_.size();
        }
        const std::vector<std::uint64_t> &data() const noexcept
        {
            return coeffs_;
        }
        std::vector<std::uint64_t> &data() noexcept
        {
            return coeffs_;
        }
        void upload(const std::uint64_t *host)
        {
            std::vector<std::uint64_t> tmp(buf_.words(), 0);
            std::copy(host, host + buf_.words() / 2, tmp.begin());
            buf_.upload(tmp.data());
        }
        const std::uint64_t *d_data() const noexcept
        {
            return buf_.d_data();
        }
        std::uint64_t *d_data() noexcept
        {
            return buf_.d_data();
        }
        // the coeff_modulus_size x N words of the plaintext
        std::vector<std::uint64_t> to_cpu() const
        {
            std::vector<std::uint64_t> all = buf_.to_cpu();
            all.resize(all.size() / 2);
            return all;
        }
        std::size_t coeff_modulus_size() const noexcept
        {
            return buf_.coeff_modulus_size();
        }
        bool is_ntt_form() const noexcept
        {
            return buf_.is_ntt_form();
        }
        bool &is_ntt_form() noexcept
        {
            return buf_.is_ntt_form();
        }
        double scale() const noexcept
        {
            return buf_.scale();
        }
        double &scale() noexcept
        {
            return buf_.scale();
        }
        // plaintext.h: parms_id_zero for a coefficient-form (BFV / BGV) plaintext, the level's id in NTT form
        parms_id_type parms_id() const
        {
            return buf_.d_data() ? buf_.parms_id() : parms_id_zero;
        }
        // Plaintext::save_members / load_members (plaintext.cpp:207-290) for an NTT-form plaintext: parms_id of its
        // level, coeff_count = limbs x N, scale, data
        std::streamoff save(std::ostream &stream) const
        {
            const std::vector<std::uint64_t> host = to_cpu();
            return Serialization::Save(
                [&](std::ostream &s) {
                    util::write_pod(s, buf_.parms_id());
                    util::write_pod<std::uint64_t>(s, host.size());
                    util::write_pod(s, buf_.scale());
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
                    const double scale = util::read_pod<double>(s);
                    const std::size_t L = context.level_of(id);
                    if (!L) // parms_id_zero marks a coefficient-form (BFV/BGV) plaintext, which has no device form here
                        throw std::logic_error("unsupported: only NTT-form plaintexts are device-resident");
                    if (coeff_count != L * context.poly_modulus_degree())
                        throw std::logic_error("plaintext data is invalid");
                    const std::vector<std::uint64_t> data = util::dynarray_load(s, (std::size_t)coeff_count);
                    if (data.size() != coeff_count)
                        throw std::logic_error("plaintext data is invalid");
                    Plaintext fresh(context, L);
                    fresh.upload(data.data());
                    fresh.scale() = scale;
                    *this = std::move(fresh);
                },
                stream);
        }

    private:
        Ciphertext buf_;
        std::vector<std::uint64_t> coeffs_;
    };

    // KSwitchKeys::data()[index] = vector<PublicKey>, one PublicKey per decomposition prime, each 2 x k x N
    // words in NTT form (kswitchkeys.h:90-101).  Here every index holds one device array [k-1][2][k][N].
    class KSwitchKeys
    {
    public:
        explicit KSwitchKeys(const SEALContext &context) : ctx_(std::make_shared<SEALContext>(context))
        {}
        KSwitchKeys(const KSwitchKeys &) = delete;
        KSwitchKeys &operator=(const KSwitchKeys &) = delete;
        virtual ~KSwitchKeys()
        {
            for (auto *p : keys_)
                if (p)
                    cat_free(ctx_->handle(), p);
        }
        std::size_t key_words() const noexcept
        {
            const std::size_t k = ctx_->key_modulus_size();
            return (k - 1) * 2 * k * ctx_->poly_modulus_degree();
        }
        // host: [k-1][2][k][N] words as produced by KeyGenerator (keygenerator.cpp:301-334) on the CPU
        void set_key(std::size_t index, const std::uint64_t *host)
        {
            if (!ctx_->using_keyswitching())
                throw std::logic_error("keyswitching is not supported by the context");
            if (keys_.size() <= index)
                keys_.resize(index + 1, nullptr);
            if (!keys_[index])
                detail::check(cat_malloc(ctx_->handle(), key_words() * 8, (void **)&keys_[index]));
            detail::check(cat_upload(ctx_->handle(), keys_[index], host, key_words() * 8, nullptr));
            detail::check(cat_stream_sync(ctx_->handle(), nullptr));
        }
        // device storage of one entry, for keys generated on the device (KeyGenerator, seal_hip_crypto.hpp)
        std::uint64_t *allocate_key(std::size_t index)
        {
            if (!ctx_->using_keyswitching())
                throw std::logic_error("keyswitching is not supported by the context");
            if (keys_.size() <= index)
                keys_.resize(index + 1, nullptr);
            if (!keys_[index])
                detail::check(cat_malloc(ctx_->handle(), key_words() * 8, (void **)&keys_[index]));
            return keys_[index];
        }
        std::vector<std::uint64_t> key_to_cpu(std::size_t index) const
        {
            std::vector<std::uint64_t> out(key_words());
            detail::check(cat_download(ctx_->handle(), out.data(), d_key(index), out.size() * 8, nullptr));
            detail::check(cat_stream_sync(ctx_->handle(), nullptr));
            return out;
        }
        bool has_index(std::size_t index) const noexcept
        {
            return index < keys_.size() && keys_[index];
        }
        const std::uint64_t *d_key(std::size_t index) const
        {
            if (!has_index(index))
                throw std::invalid_argument("key not present");
            return keys_[index];
        }
        std::size_t size() const noexcept
        {
            std::size_t c = 0;
            for (auto *p : keys_)
                c += p != nullptr;
            return c;
        }

        // KSwitchKeys::save_members / load_members (kswitchkeys.cpp:42-160): key-level parms_id, the number of entries,
        // then per entry the number of PublicKeys (0 = absent) and each PublicKey = a size-2 key-level NTT-form
        // Ciphertext object.  `entries` is keys_.size() of the reference: the key count for RelinKeys, N for GaloisKeys
        // (keygenerator.cpp:188).
        std::streamoff save(std::ostream &stream) const
        {
            const std::size_t k = ctx_->key_modulus_size(), n = ctx_->poly_modulus_degree(), entries = saved_entries();
            Ciphertext pk(*ctx_, k, 2);
            pk.is_ntt_form() = true;
            std::streamoff raw = static_cast<std::streamoff>(sizeof(Serialization::SEALHeader) + sizeof(parms_id_type) + 8 + 8 * entries);
            for (std::size_t i = 0; i < entries; ++i)
                if (has_index(i))
                    raw += static_cast<std::streamoff>(k - 1) * pk.save_size();
            return Serialization::Save(
                [&](std::ostream &s) {
                    util::write_pod(s, ctx_->key_parms_id());
                    util::write_pod<std::uint64_t>(s, entries);
                    for (std::size_t i = 0; i < entries; ++i)
                    {
                        util::write_pod<std::uint64_t>(s, has_index(i) ? k - 1 : 0);
                        if (!has_index(i))
                            continue;
                        const std::vector<std::uint64_t> host = key_to_cpu(i);
                        for (std::size_t j = 0; j + 1 < k; ++j)
                        {
                            pk.upload(host.data() + j * 2 * k * n);
                            pk.save(s);
                        }
                    }
                },
                raw, stream);
        }
        std::streamoff load(const SEALContext &context, std::istream &stream)
        {
            if (context.handle() != ctx_->handle())
                throw std::invalid_argument("encryption parameters are not set correctly");
            const std::size_t k = ctx_->key_modulus_size(), n = ctx_->poly_modulus_degree();
            std::vector<std::vector<std::uint64_t>> loaded;
            const std::streamoff in_size = Serialization::Load(
                [&](std::istream &s, SEALVersion) {
                    if (util::read_pod<parms_id_type>(s) != ctx_->key_parms_id())
                        throw std::logic_error("KSwitchKeys data is invalid");
                    const std::uint64_t dim1 = util::read_pod<std::uint64_t>(s);
                    if (dim1 > n) // SEAL_CIPHERTEXT_SIZE_MAX - 2 for RelinKeys, N for GaloisKeys
                        throw std::logic_error("KSwitchKeys data is invalid");
                    loaded.resize((std::size_t)dim1);
                    for (std::uint64_t i = 0; i < dim1; ++i)
                    {
                        const std::uint64_t dim2 = util::read_pod<std::uint64_t>(s);
                        if (dim2 != 0 && dim2 != k - 1)
                            throw std::logic_error("KSwitchKeys data is invalid");
                        for (std::uint64_t j = 0; j < dim2; ++j)
                        {
                            Ciphertext pk;
                            pk.load(context, s);
                            if (pk.size() != 2 || pk.coeff_modulus_size() != k || !pk.is_ntt_form())
                                throw std::logic_error("PublicKey data is invalid");
                            const std::vector<std::uint64_t> w = pk.to_cpu();
                            loaded[(std::size_t)i].insert(loaded[(std::size_t)i].end(), w.begin(), w.end());
                        }
                    }
                },
                stream);
            for (auto *p : keys_)
                if (p)
                    cat_free(ctx_->handle(), p);
            keys_.assign(loaded.size(), nullptr);
            for (std::size_t i = 0; i < loaded.size(); ++i)
                if (!loaded[i].empty())
                    set_key(i, loaded[i].data());
            return in_size;
        }

    protected:
        // keys_.size() as the reference would have it
        virtual std::size_t saved_entries() const
        {
            return keys_.size();
        }

        std::shared_ptr<SEALContext> ctx_;
        std::vector<std::uint64_t *> keys_;
    };

    class RelinKeys : public KSwitchKeys
    {
    public:
        using KSwitchKeys::KSwitchKeys;
        // relinkeys.h:58-64
        static std::size_t get_index(std::size_t key_power)
        {
            if (key_power < 2)
                throw std::invalid_argument("key_power cannot be less than 2");
            return key_power - 2;
        }
        bool has_key(std::size_t key_power) const
        {
            return has_index(get_index(key_power));
        }
    };

    class GaloisKeys : public KSwitchKeys
    {
    public:
        using KSwitchKeys::KSwitchKeys;
        // galoiskeys.h:48-50, util/galois.h:144-153
        static std::size_t get_index(std::uint32_t galois_elt)
        {
            if (!(galois_elt & 1))
                throw std::invalid_argument("galois_elt is not valid");
            return (galois_elt - 1) >> 1;
        }
        bool has_key(std::uint32_t galois_elt) const
        {
            return has_index(get_index(galois_elt));
        }
        ~GaloisKeys() override
        {
            for (auto &kv : hoisted_)
                cat_free(ctx_->handle(), kv.second);
        }
        // Not in the reference: the key of galois_elt prepared for hoisted rotations (cat_hoisted_key_prepare: sigma^-1
        // applied to every limb), made on first use and kept beside the key.  Changing a key with set_key / load after
        // its first hoisted use is not supported.
        const std::uint64_t *d_hoisted_key(std::uint32_t galois_elt) const
        {
            auto it = hoisted_.find(galois_elt);
            if (it != hoisted_.end())
                return it->second;
            const std::uint64_t *key = d_key(get_index(galois_elt));
            std::uint64_t *prepared = nullptr;
            detail::check(cat_malloc(ctx_->handle(), key_words() * 8, (void **)&prepared));
            const cat_status st = cat_hoisted_key_prepare(ctx_->handle(), key, galois_elt, prepared, nullptr);
            if (st != CAT_OK)
            {
                cat_free(ctx_->handle(), prepared);
                detail::check(st);
            }
            hoisted_[galois_elt] = prepared;
            return prepared;
        }

    protected:
        mutable std::map<std::uint32_t, std::uint64_t *> hoisted_;
        std::size_t saved_entries() const override // galois_keys.data().resize(coeff_count), keygenerator.cpp:188
        {
            return ctx_->poly_modulus_degree();
        }
    };

    namespace util
    {
        // NTTTables of one prime of the context (util/ntt.h:69-199); the tables themselves are device-resident.
        class NTTTables
        {
        public:
            NTTTables(const SEALContext &context, std::uint32_t modulus_index)
                : ctx_(context.handle()), index_(modulus_index)
            {}
            std::uint64_t get_root() const
            {
                return cat_ctx_root(ctx_, index_);
            }
            std::uint64_t modulus() const
            {
                return cat_ctx_modulus(ctx_, index_);
            }
            cat_ctx *ctx() const noexcept
            {
                return ctx_;
            }
            std::uint32_t index() const noexcept
            {
                return index_;
            }

        private:
            cat_ctx *ctx_;
            std::uint32_t index_;
        };
        using Modulus = NTTTables; // the coefficient-wise ops only need "which prime"

        // CoeffIter overloads (util/ntt.h:201-270): operand = N device words
        inline void ntt_negacyclic_harvey_lazy(std::uint64_t *d_operand, const NTTTables &t)
        {
            const std::uint32_t id = t.index();
            detail::check(cat_ntt_forward(t.ctx(), d_operand, &id, 1, 1, 1, nullptr));
        }
        inline void ntt_negacyclic_harvey(std::uint64_t *d_operand, const NTTTables &t)
        {
            const std::uint32_t id = t.index();
            detail::check(cat_ntt_forward(t.ctx(), d_operand, &id, 1, 1, 0, nullptr));
        }
        inline void inverse_ntt_negacyclic_harvey_lazy(std::uint64_t *d_operand, const NTTTables &t)
        {
            const std::uint32_t id = t.index();
            detail::check(cat_ntt_inverse(t.ctx(), d_operand, &id, 1, 1, 1, nullptr));
        }
        inline void inverse_ntt_negacyclic_harvey(std::uint64_t *d_operand, const NTTTables &t)
        {
            const std::uint32_t id = t.index();
            detail::check(cat_ntt_inverse(t.ctx(), d_operand, &id, 1, 1, 0, nullptr));
        }
        // (operand1, operand2, coeff_count, modulus, result), polyarithsmallmod.cuh:172-175, 544-546;
        // coeff_count must be the context's N (one limb per call, like the CoeffIter overloads)
        inline void add_poly_coeffmod(const std::uint64_t *a, const std::uint64_t *b, std::size_t, const Modulus &m,
                                      std::uint64_t *r)
        {
            const std::uint32_t id = m.index();
            detail::check(cat_add(m.ctx(), a, b, r, &id, 1, 1, nullptr));
        }
        inline void sub_poly_coeffmod(const std::uint64_t *a, const std::uint64_t *b, std::size_t, const Modulus &m,
                                      std::uint64_t *r)
        {
            const std::uint32_t id = m.index();
            detail::check(cat_sub(m.ctx(), a, b, r, &id, 1, 1, nullptr));
        }
        inline void negate_poly_coeffmod(const std::uint64_t *a, std::size_t, const Modulus &m, std::uint64_t *r)
        {
            const std::uint32_t id = m.index();
            detail::check(cat_negate(m.ctx(), a, r, &id, 1, 1, nullptr));
        }
        inline void dyadic_product_coeffmod(const std::uint64_t *a, const std::uint64_t *b, std::size_t,
                                            const Modulus &m, std::uint64_t *r)
        {
            const std::uint32_t id = m.index();
            detail::check(cat_dyadic(m.ctx(), a, b, r, &id, 1, 1, nullptr));
        }
    } // namespace util

    class Evaluator
    {
    public:
        explicit Evaluator(const SEALContext &context) : context_(context)
        {}

        void negate_inplace(Ciphertext &encrypted) const
        {
            verify(encrypted, "encrypted");
            detail::check(cat_negate(h(), encrypted.d_data(), encrypted.d_data(), nullptr,
                                     (std::uint32_t)encrypted.coeff_modulus_size(), (std::uint32_t)encrypted.size(),
                                     nullptr));
        }

        // evaluator.cpp:145-237 and :265-350: operands of different sizes (the longer tail is copied, or negated for a
        // subtraction) and, in BGV, of different correction factors (both sides are scaled first, :196-216)
        void add_inplace(Ciphertext &encrypted1, const Ciphertext &encrypted2) const
        {
            add_sub(encrypted1, encrypted2, false);
        }
        void sub_inplace(Ciphertext &encrypted1, const Ciphertext &encrypted2) const
        {
            add_sub(encrypted1, encrypted2, true);
        }

        // evaluator.cpp:352-393 -> ckks_multiply :569-708 / bfv_multiply :395-567 (sizes 2 x 2 -> 3)
        void multiply_inplace(Ciphertext &encrypted1, const Ciphertext &encrypted2) const
        {
            verify(encrypted1, "encrypted1");
            verify(encrypted2, "encrypted2");
            if (encrypted1.coeff_modulus_size() != encrypted2.coeff_modulus_size())
                throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
            const std::uint32_t L = (std::uint32_t)encrypted1.coeff_modulus_size();
            if (encrypted1.size() != 2 || encrypted2.size() != 2)
            {
                if (context_.parms().scheme() == scheme_type::bfv)
                    throw std::logic_error("unsupported: BFV operands larger than size 2 are not multiplied on the device");
                multiply_general(encrypted1, encrypted2);
                return;
            }
            Ciphertext dest(context_, L, 3);
            switch (context_.parms().scheme())
            {
            case scheme_type::ckks:
                if (!encrypted1.is_ntt_form() || !encrypted2.is_ntt_form())
                    throw std::invalid_argument("encrypted1 or encrypted2 must be in NTT form");
                detail::check(cat_ckks_multiply(h(), encrypted1.d_data(), encrypted2.d_data(), dest.d_data(), L, 1, nullptr));
                dest.is_ntt_form() = true;
                dest.scale() = encrypted1.scale() * encrypted2.scale();
                break;
            case scheme_type::bfv:
                if (encrypted1.is_ntt_form() || encrypted2.is_ntt_form())
                    throw std::invalid_argument("encrypted1 or encrypted2 cannot be in NTT form");
                detail::check(cat_bfv_multiply(h(), encrypted1.d_data(), encrypted2.d_data(), dest.d_data(), L, 1, nullptr));
                dest.scale() = encrypted1.scale();
                break;
            case scheme_type::bgv: // evaluator.cpp:710-841
                if (!encrypted1.is_ntt_form() || !encrypted2.is_ntt_form())
                    throw std::invalid_argument("encrypted1 or encrypted2 must be in NTT form");
                detail::check(cat_bgv_multiply(h(), encrypted1.d_data(), encrypted2.d_data(), dest.d_data(), L, 1, nullptr));
                dest.is_ntt_form() = true;
                dest.scale() = encrypted1.scale();
                dest.correction_factor() = mul_mod_t(encrypted1.correction_factor(), encrypted2.correction_factor());
                break;
            default:
                throw std::invalid_argument("unsupported scheme");
            }
            encrypted1 = std::move(dest);
        }
        void multiply(const Ciphertext &encrypted1, const Ciphertext &encrypted2, Ciphertext &destination) const
        {
            if (&encrypted2 == &destination)
            {
                Ciphertext tmp(encrypted1);
                multiply_inplace(tmp, encrypted2);
                destination = std::move(tmp);
                return;
            }
            destination = encrypted1;
            multiply_inplace(destination, encrypted2);
        }

        // evaluator.cpp:993-1078 (CKKS, size 2 -> 3)
        void square_inplace(Ciphertext &encrypted) const
        {
            verify(encrypted, "encrypted");
            if (context_.parms().scheme() != scheme_type::ckks)
            {
                Ciphertext copy(encrypted);
                multiply_inplace(encrypted, copy);
                return;
            }
            if (!encrypted.is_ntt_form())
                throw std::invalid_argument("encrypted must be in NTT form");
            if (encrypted.size() != 2)
                throw std::logic_error("unsupported: only size-2 operands are squared on the device");
            const std::uint32_t L = (std::uint32_t)encrypted.coeff_modulus_size();
            Ciphertext dest(context_, L, 3);
            detail::check(cat_ckks_square(h(), encrypted.d_data(), dest.d_data(), L, 1, nullptr));
            dest.is_ntt_form() = true;
            dest.scale() = encrypted.scale() * encrypted.scale();
            encrypted = std::move(dest);
        }

        // evaluator.cpp:1953-1983 -> multiply_plain_ntt :2135-2172
        void multiply_plain_inplace(Ciphertext &encrypted, const Plaintext &plain) const
        {
            verify(encrypted, "encrypted");
            if (!plain.is_ntt_form())
            {
                // coefficient-form plaintext modulo t (BFV / BGV): multiply_plain_normal :1985-2133, or the lifted
                // plaintext's transform against an NTT-form ciphertext :1972-1977
                multiply_plain_coeff(encrypted, plain);
                return;
            }
            if (!encrypted.is_ntt_form())
            {
                // :1978-1983
                transform_to_ntt_inplace(encrypted);
                multiply_plain_inplace(encrypted, plain);
                transform_from_ntt_inplace(encrypted);
                return;
            }
            if (encrypted.coeff_modulus_size() != plain.coeff_modulus_size())
                throw std::invalid_argument("encrypted_ntt and plain_ntt parameter mismatch");
            detail::check(cat_multiply_plain_ntt(h(), encrypted.d_data(), plain.d_data(), 0, encrypted.d_data(),
                                                 (std::uint32_t)encrypted.coeff_modulus_size(),
                                                 (std::uint32_t)encrypted.size(), 1, nullptr));
            encrypted.scale() *= plain.scale();
        }

        // evaluator.cpp:1737-1843 / 1845-1951, CKKS branch
        void add_plain_inplace(Ciphertext &encrypted, const Plaintext &plain) const
        {
            plain_addsub(encrypted, plain, false);
        }
        void sub_plain_inplace(Ciphertext &encrypted, const Plaintext &plain) const
        {
            plain_addsub(encrypted, plain, true);
        }

        // evaluator.cpp:1144-1199 (destination_size 2): one key switch per dropped polynomial
        void relinearize_inplace(Ciphertext &encrypted, const RelinKeys &relin_keys) const
        {
            verify(encrypted, "encrypted");
            std::size_t size = encrypted.size();
            if (size < 2)
                throw std::invalid_argument("destination_size must be at least 2 and less than or equal to current count");
            if (relin_keys.size() < size - 2)
                throw std::invalid_argument("not enough relinearization keys");
            const std::uint32_t L = (std::uint32_t)encrypted.coeff_modulus_size();
            while (size > 2)
            {
                const std::uint64_t *key = relin_keys.d_key(RelinKeys::get_index(size - 1));
                detail::check(cat_switch_key(h(), encrypted.d_data(), encrypted.d_data(size - 1), key, L, 1, nullptr));
                --size;
            }
            encrypted.resize(2);
        }

        // evaluator.cpp:1481-1519
        void rescale_to_next(const Ciphertext &encrypted, Ciphertext &destination) const
        {
            verify(encrypted, "encrypted");
            if (encrypted.coeff_modulus_size() < 2)
                throw std::invalid_argument("end of modulus switching chain reached");
            if (context_.parms().scheme() != scheme_type::ckks)
                throw std::invalid_argument("unsupported operation for scheme type");
            if (!encrypted.is_ntt_form())
                throw std::invalid_argument("CKKS encrypted must be in NTT form");
            scale_to_next(encrypted, destination);
        }
        void rescale_to_next_inplace(Ciphertext &encrypted) const
        {
            Ciphertext dest;
            rescale_to_next(encrypted, dest);
            encrypted = std::move(dest);
        }

        // Not in the reference: relinearize_inplace directly followed by rescale_to_next_inplace (CKKS) or
        // mod_switch_to_next_inplace (BFV, BGV) -- the usual tail of a multiplication -- as one device call
        // (cat_relinearize_rescale: one forward transform per limb serves both steps).  Same words, same metadata
        // and same exceptions as the two calls; sizes other than 3 take the two calls.
        void relinearize_rescale_inplace(Ciphertext &encrypted, const RelinKeys &relin_keys) const
        {
            verify(encrypted, "encrypted");
            const scheme_type scheme = context_.parms().scheme();
            auto second = [&] {
                if (scheme == scheme_type::ckks)
                    rescale_to_next_inplace(encrypted);
                else
                    mod_switch_to_next_inplace(encrypted);
            };
            if (encrypted.size() != 3)
            {
                relinearize_inplace(encrypted, relin_keys);
                second();
                return;
            }
            if (relin_keys.size() < 1)
                throw std::invalid_argument("not enough relinearization keys");
            const std::uint32_t L = (std::uint32_t)encrypted.coeff_modulus_size();
            if (L < 2)
                throw std::invalid_argument("end of modulus switching chain reached");
            if (scheme == scheme_type::bfv ? encrypted.is_ntt_form() : !encrypted.is_ntt_form())
                throw std::invalid_argument(scheme == scheme_type::bfv    ? "BFV encrypted cannot be in NTT form"
                                            : scheme == scheme_type::bgv ? "BGV encrypted must be in NTT form"
                                                                         : "CKKS encrypted must be in NTT form");
            Ciphertext dest(context_, L - 1, 2);
            detail::check(cat_relinearize_rescale(h(), encrypted.d_data(), relin_keys.d_key(RelinKeys::get_index(2)),
                                                  dest.d_data(), L, 1, nullptr));
            scaled_metadata(encrypted, dest, L);
            encrypted = std::move(dest);
        }

        // Not in the reference: rotate_vector_inplace directly followed by rescale_to_next_inplace (CKKS) as one
        // device call when the step has its own key (cat_apply_galois_rescale); otherwise the two calls.
        void rotate_vector_rescale_inplace(Ciphertext &encrypted, int steps, const GaloisKeys &galois_keys) const
        {
            if (context_.parms().scheme() != scheme_type::ckks)
                throw std::logic_error("unsupported scheme");
            verify(encrypted, "encrypted");
            const std::uint32_t elt = steps ? cat_galois_elt_from_step(log_n(), steps) : 0;
            const std::uint32_t L = (std::uint32_t)encrypted.coeff_modulus_size();
            if (!elt || !galois_keys.has_key(elt) || encrypted.size() != 2 || L < 2 || !encrypted.is_ntt_form() ||
                !context_.using_keyswitching())
            {
                rotate_internal(encrypted, steps, galois_keys);
                rescale_to_next_inplace(encrypted);
                return;
            }
            Ciphertext dest(context_, L - 1, 2);
            detail::check(cat_apply_galois_rescale(h(), encrypted.d_data(), elt, galois_keys.d_key(GaloisKeys::get_index(elt)),
                                                   dest.d_data(), L, 1, nullptr));
            scaled_metadata(encrypted, dest, L);
            encrypted = std::move(dest);
        }

        // evaluator.cpp:1384-1419: BFV and BGV -> mod_switch_scale_to_next, CKKS -> mod_switch_drop_to_next
        void mod_switch_to_next_inplace(Ciphertext &encrypted) const
        {
            verify(encrypted, "encrypted");
            if (encrypted.coeff_modulus_size() < 2)
                throw std::invalid_argument("end of modulus switching chain reached");
            Ciphertext dest;
            if (context_.parms().scheme() == scheme_type::bfv)
            {
                if (encrypted.is_ntt_form())
                    throw std::invalid_argument("BFV encrypted cannot be in NTT form");
                scale_to_next(encrypted, dest);
                dest.scale() = encrypted.scale();
            }
            else if (context_.parms().scheme() == scheme_type::bgv)
            {
                if (!encrypted.is_ntt_form())
                    throw std::invalid_argument("BGV encrypted must be in NTT form");
                scale_to_next(encrypted, dest);
                dest.scale() = encrypted.scale();
            }
            else if (context_.parms().scheme() == scheme_type::ckks)
            {
                if (!encrypted.is_ntt_form())
                    throw std::invalid_argument("CKKS encrypted must be in NTT form");
                const std::uint32_t L = (std::uint32_t)encrypted.coeff_modulus_size();
                dest.resize(context_, L - 1, encrypted.size());
                detail::check(cat_mod_drop(h(), encrypted.d_data(), dest.d_data(), L, (std::uint32_t)encrypted.size(), 1, nullptr));
                dest.is_ntt_form() = true;
                dest.scale() = encrypted.scale();
            }
            else
                throw std::invalid_argument("unsupported scheme");
            encrypted = std::move(dest);
        }

        // evaluator.cpp:2267-2360
        void transform_to_ntt_inplace(Ciphertext &encrypted) const
        {
            verify(encrypted, "encrypted");
            if (encrypted.is_ntt_form())
                throw std::invalid_argument("encrypted is already in NTT form");
            detail::check(cat_ntt_forward(h(), encrypted.d_data(), nullptr, (std::uint32_t)encrypted.coeff_modulus_size(),
                                          (std::uint32_t)encrypted.size(), 0, nullptr));
            encrypted.is_ntt_form() = true;
        }
        void transform_from_ntt_inplace(Ciphertext &encrypted_ntt) const
        {
            verify(encrypted_ntt, "encrypted_ntt");
            if (!encrypted_ntt.is_ntt_form())
                throw std::invalid_argument("encrypted_ntt is not in NTT form");
            detail::check(cat_ntt_inverse(h(), encrypted_ntt.d_data(), nullptr,
                                          (std::uint32_t)encrypted_ntt.coeff_modulus_size(),
                                          (std::uint32_t)encrypted_ntt.size(), 0, nullptr));
            encrypted_ntt.is_ntt_form() = false;
        }

        // evaluator.cpp:2362-2464
        void apply_galois_inplace(Ciphertext &encrypted, std::uint32_t galois_elt, const GaloisKeys &galois_keys) const
        {
            verify(encrypted, "encrypted");
            if (!(galois_elt & 1) || galois_elt >= 2 * (std::uint64_t)context_.poly_modulus_degree())
                throw std::invalid_argument("Galois element is not valid");
            if (!galois_keys.has_key(galois_elt))
                throw std::invalid_argument("Galois key not present");
            if (encrypted.size() > 2)
                throw std::invalid_argument("encrypted size must be 2");
            detail::check(cat_apply_galois_inplace(h(), encrypted.d_data(), galois_elt,
                                                   galois_keys.d_key(GaloisKeys::get_index(galois_elt)),
                                                   (std::uint32_t)encrypted.coeff_modulus_size(), 1, nullptr));
        }

        // Not in the reference: several Galois elements applied to ONE ciphertext with the
        // decomposition, the modulus raise and the forward transforms of the extended digits shared (cat_rotate_hoisted).
        // Every element needs its own key (no NAF chain: a chain is sequential and has nothing to share).  The outputs
        // decrypt to what apply_galois_inplace's do; they are not its words (cat_amd.h).
        void apply_galois_hoisted(const Ciphertext &encrypted, const std::vector<std::uint32_t> &galois_elts,
                                  const GaloisKeys &galois_keys, std::vector<Ciphertext> &destination) const
        {
            verify(encrypted, "encrypted");
            if (!context_.using_keyswitching())
                throw std::logic_error("encryption parameters do not support batching");
            if (encrypted.size() > 2)
                throw std::invalid_argument("encrypted size must be 2");
            const std::uint32_t L = (std::uint32_t)encrypted.coeff_modulus_size();
            std::vector<const std::uint64_t *> keys;
            for (std::uint32_t elt : galois_elts)
            {
                if (!(elt & 1) || elt >= 2 * (std::uint64_t)context_.poly_modulus_degree())
                    throw std::invalid_argument("Galois element is not valid");
                if (!galois_keys.has_key(elt))
                    throw std::invalid_argument("Galois key not present");
                keys.push_back(galois_keys.d_hoisted_key(elt));
            }
            std::vector<Ciphertext> result(galois_elts.size());
            std::vector<std::uint64_t *> outs;
            for (Ciphertext &c : result)
            {
                c.resize(context_, L, 2);
                c.is_ntt_form() = encrypted.is_ntt_form();
                c.scale() = encrypted.scale();
                c.correction_factor() = encrypted.correction_factor();
                outs.push_back(c.d_data());
            }
            detail::check(cat_rotate_hoisted(h(), encrypted.d_data(), galois_elts.data(), keys.data(),
                                             (std::uint32_t)galois_elts.size(), outs.data(), L, 1, nullptr));
            destination = std::move(result);
        }
        // rotate_vector (CKKS) / rotate_rows (BFV, BGV) by several step counts at once
        void rotate_hoisted(const Ciphertext &encrypted, const std::vector<int> &steps, const GaloisKeys &galois_keys,
                            std::vector<Ciphertext> &destination) const
        {
            std::vector<std::uint32_t> elts;
            for (int st : steps)
            {
                const std::uint32_t elt = st == 0 ? 0 : cat_galois_elt_from_step(log_n(), st);
                if (elt == 0)
                    throw std::invalid_argument(st == 0 ? "a rotation by zero steps has no Galois element" : "step count too large");
                elts.push_back(elt);
            }
            apply_galois_hoisted(encrypted, elts, galois_keys, destination);
        }

        // evaluator.h:1195-1237 -> rotate_internal (evaluator.cpp:2466-2521), incl. the NAF fallback
        void rotate_vector_inplace(Ciphertext &encrypted, int steps, const GaloisKeys &galois_keys) const
        {
            if (context_.parms().scheme() != scheme_type::ckks)
                throw std::logic_error("unsupported scheme");
            rotate_internal(encrypted, steps, galois_keys);
        }
        void rotate_rows_inplace(Ciphertext &encrypted, int steps, const GaloisKeys &galois_keys) const
        {
            if (context_.parms().scheme() != scheme_type::bfv && context_.parms().scheme() != scheme_type::bgv)
                throw std::logic_error("unsupported scheme"); // evaluator.h:1063
            rotate_internal(encrypted, steps, galois_keys);
        }
        void complex_conjugate_inplace(Ciphertext &encrypted, const GaloisKeys &galois_keys) const
        {
            if (context_.parms().scheme() != scheme_type::ckks)
                throw std::logic_error("unsupported scheme");
            apply_galois_inplace(encrypted, cat_galois_elt_from_step(log_n(), 0), galois_keys);
        }

        // ---- the reference's out-of-place and chained forms (evaluator.h): each is the copy / loop it is there ----
        void negate(const Ciphertext &encrypted, Ciphertext &destination) const // evaluator.h:111-115
        {
            destination = encrypted;
            negate_inplace(destination);
        }
        void add(const Ciphertext &encrypted1, const Ciphertext &encrypted2, Ciphertext &destination) const // :139-150
        {
            if (&encrypted2 == &destination)
                add_inplace(destination, encrypted1);
            else
            {
                destination = encrypted1;
                add_inplace(destination, encrypted2);
            }
        }
        void add_many(const std::vector<Ciphertext> &encrypteds, Ciphertext &destination) const // evaluator.cpp:242-261
        {
            if (encrypteds.empty())
                throw std::invalid_argument("encrypteds cannot be empty");
            for (const Ciphertext &c : encrypteds)
                if (&c == &destination)
                    throw std::invalid_argument("encrypteds must be different from destination");
            destination = encrypteds[0];
            for (std::size_t i = 1; i < encrypteds.size(); ++i)
                add_inplace(destination, encrypteds[i]);
        }
        void sub(const Ciphertext &encrypted1, const Ciphertext &encrypted2, Ciphertext &destination) const // :187-199
        {
            if (&encrypted2 == &destination)
            {
                sub_inplace(destination, encrypted1);
                negate_inplace(destination);
            }
            else
            {
                destination = encrypted1;
                sub_inplace(destination, encrypted2);
            }
        }
        void square(const Ciphertext &encrypted, Ciphertext &destination) const // :262-267
        {
            destination = encrypted;
            square_inplace(destination);
        }
        void relinearize(const Ciphertext &encrypted, const RelinKeys &relin_keys, Ciphertext &destination) const // :327-333
        {
            destination = encrypted;
            relinearize_inplace(destination, relin_keys);
        }
        void mod_switch_to_next(const Ciphertext &encrypted, Ciphertext &destination) const // :352-355
        {
            destination = encrypted;
            mod_switch_to_next_inplace(destination);
        }
        // NTT-form plaintexts drop their last limb: mod_switch_drop_to_next(Plaintext &) evaluator.cpp:1347-1382,
        // mod_switch_to_next_inplace :1421-1427, mod_switch_to_inplace :1453-1479
        void mod_switch_to_next_inplace(Plaintext &plain) const
        {
            if (!plain.d_data() || plain.coeff_modulus_size() == 0)
                throw std::invalid_argument("plain is not valid for encryption parameters");
            if (!plain.is_ntt_form())
                throw std::invalid_argument("plain is not in NTT form");
            const std::uint32_t L = (std::uint32_t)plain.coeff_modulus_size();
            if (L < 2)
                throw std::invalid_argument("end of modulus switching chain reached");
            Plaintext dest(context_, L - 1);
            detail::check(cat_mod_drop(h(), plain.d_data(), dest.d_data(), L, 1, 1, nullptr));
            dest.scale() = plain.scale();
            plain = std::move(dest);
        }
        void mod_switch_to_next(const Plaintext &plain, Plaintext &destination) const
        {
            destination = plain;
            mod_switch_to_next_inplace(destination);
        }
        void mod_switch_to_inplace(Plaintext &plain, const parms_id_type &parms_id) const
        {
            const std::size_t have = plain.d_data() ? plain.coeff_modulus_size() : 0, want = context_.level_of(parms_id);
            if (!have)
                throw std::invalid_argument("plain is not valid for encryption parameters");
            if (!want)
                throw std::invalid_argument("parms_id is not valid for encryption parameters");
            if (!plain.is_ntt_form())
                throw std::invalid_argument("plain is not in NTT form");
            if (have < want)
                throw std::invalid_argument("cannot switch to higher level modulus");
            while (plain.coeff_modulus_size() != want)
                mod_switch_to_next_inplace(plain);
        }
        void mod_switch_to(const Plaintext &plain, const parms_id_type &parms_id, Plaintext &destination) const
        {
            destination = plain;
            mod_switch_to_inplace(destination, parms_id);
        }
        // evaluator.cpp:1429-1451
        void mod_switch_to_inplace(Ciphertext &encrypted, const parms_id_type &parms_id) const
        {
            check_target_level(encrypted, parms_id);
            while (encrypted.parms_id() != parms_id)
                mod_switch_to_next_inplace(encrypted);
        }
        void mod_switch_to(const Ciphertext &encrypted, const parms_id_type &parms_id, Ciphertext &destination) const
        {
            destination = encrypted;
            mod_switch_to_inplace(destination, parms_id);
        }
        // evaluator.cpp:1521-1570
        void rescale_to_inplace(Ciphertext &encrypted, const parms_id_type &parms_id) const
        {
            verify(encrypted, "encrypted");
            check_target_level(encrypted, parms_id);
            if (context_.parms().scheme() != scheme_type::ckks)
                throw std::invalid_argument("unsupported operation for scheme type");
            while (encrypted.parms_id() != parms_id)
                rescale_to_next_inplace(encrypted);
        }
        void rescale_to(const Ciphertext &encrypted, const parms_id_type &parms_id, Ciphertext &destination) const
        {
            destination = encrypted;
            rescale_to_inplace(destination, parms_id);
        }
        // evaluator.cpp:1627-1702: the pairwise product tree, relinearized after every product (BFV and BGV)
        void multiply_many(const std::vector<Ciphertext> &encrypteds, const RelinKeys &relin_keys, Ciphertext &destination) const
        {
            if (encrypteds.empty())
                throw std::invalid_argument("encrypteds vector must not be empty");
            for (const Ciphertext &c : encrypteds)
                if (&c == &destination)
                    throw std::invalid_argument("encrypteds must be different from destination");
            verify(encrypteds[0], "encrypteds");
            if (context_.parms().scheme() != scheme_type::bfv && context_.parms().scheme() != scheme_type::bgv)
                throw std::logic_error("unsupported scheme");
            if (encrypteds.size() == 1)
            {
                destination = encrypteds[0];
                return;
            }
            std::vector<Ciphertext> product_vec;
            for (std::size_t i = 0; i + 1 < encrypteds.size(); i += 2)
            {
                Ciphertext temp;
                if (encrypteds[i].d_data() == encrypteds[i + 1].d_data())
                    square(encrypteds[i], temp);
                else
                    multiply(encrypteds[i], encrypteds[i + 1], temp);
                relinearize_inplace(temp, relin_keys);
                product_vec.emplace_back(std::move(temp));
            }
            if (encrypteds.size() & 1)
                product_vec.emplace_back(encrypteds.back());
            for (std::size_t i = 0; i + 1 < product_vec.size(); i += 2)
            {
                Ciphertext temp;
                multiply(product_vec[i], product_vec[i + 1], temp);
                relinearize_inplace(temp, relin_keys);
                product_vec.emplace_back(std::move(temp));
            }
            destination = product_vec.back();
        }
        // evaluator.cpp:1704-1736
        void exponentiate_inplace(Ciphertext &encrypted, std::uint64_t exponent, const RelinKeys &relin_keys) const
        {
            verify(encrypted, "encrypted");
            if (exponent == 0)
                throw std::invalid_argument("exponent cannot be 0");
            if (exponent == 1)
                return;
            std::vector<Ciphertext> exp_vector(static_cast<std::size_t>(exponent), encrypted);
            multiply_many(exp_vector, relin_keys, encrypted);
        }
        void exponentiate(const Ciphertext &encrypted, std::uint64_t exponent, const RelinKeys &relin_keys,
                          Ciphertext &destination) const
        {
            destination = encrypted;
            exponentiate_inplace(destination, exponent, relin_keys);
        }
        void multiply_plain(const Ciphertext &encrypted, const Plaintext &plain, Ciphertext &destination) const
        {
            destination = encrypted;
            multiply_plain_inplace(destination, plain);
        }
        void add_plain(const Ciphertext &encrypted, const Plaintext &plain, Ciphertext &destination) const
        {
            destination = encrypted;
            add_plain_inplace(destination, plain);
        }
        void sub_plain(const Ciphertext &encrypted, const Plaintext &plain, Ciphertext &destination) const
        {
            destination = encrypted;
            sub_plain_inplace(destination, plain);
        }
        void transform_to_ntt(const Ciphertext &encrypted, Ciphertext &destination_ntt) const
        {
            destination_ntt = encrypted;
            transform_to_ntt_inplace(destination_ntt);
        }
        void transform_from_ntt(const Ciphertext &encrypted_ntt, Ciphertext &destination) const
        {
            destination = encrypted_ntt;
            transform_from_ntt_inplace(destination);
        }
        void apply_galois(const Ciphertext &encrypted, std::uint32_t galois_elt, const GaloisKeys &galois_keys,
                          Ciphertext &destination) const
        {
            destination = encrypted;
            apply_galois_inplace(destination, galois_elt, galois_keys);
        }
        void rotate_vector(const Ciphertext &encrypted, int steps, const GaloisKeys &galois_keys, Ciphertext &destination) const
        {
            destination = encrypted;
            rotate_vector_inplace(destination, steps, galois_keys);
        }
        void rotate_rows(const Ciphertext &encrypted, int steps, const GaloisKeys &galois_keys, Ciphertext &destination) const
        {
            destination = encrypted;
            rotate_rows_inplace(destination, steps, galois_keys);
        }
        // evaluator.h:1127-1141: the column swap of the batching matrix = Galois element 2N - 1
        void rotate_columns_inplace(Ciphertext &encrypted, const GaloisKeys &galois_keys) const
        {
            if (context_.parms().scheme() != scheme_type::bfv && context_.parms().scheme() != scheme_type::bgv)
                throw std::logic_error("unsupported scheme");
            apply_galois_inplace(encrypted, cat_galois_elt_from_step(log_n(), 0), galois_keys);
        }
        void rotate_columns(const Ciphertext &encrypted, const GaloisKeys &galois_keys, Ciphertext &destination) const
        {
            destination = encrypted;
            rotate_columns_inplace(destination, galois_keys);
        }
        void complex_conjugate(const Ciphertext &encrypted, const GaloisKeys &galois_keys, Ciphertext &destination) const
        {
            destination = encrypted;
            complex_conjugate_inplace(destination, galois_keys);
        }

    private:
        // mod_switch_to / rescale_to: both levels exist and the target is not above the operand (evaluator.cpp:1432-1446)
        void check_target_level(const Ciphertext &encrypted, const parms_id_type &parms_id) const
        {
            const std::size_t have = context_.level_of(encrypted.parms_id()), want = context_.level_of(parms_id);
            if (!have)
                throw std::invalid_argument("encrypted is not valid for encryption parameters");
            if (!want)
                throw std::invalid_argument("parms_id is not valid for encryption parameters");
            if (have < want)
                throw std::invalid_argument("cannot switch to higher level modulus");
        }
        cat_ctx *h() const noexcept
        {
            return context_.handle();
        }
        std::uint32_t log_n() const noexcept
        {
            return cat_ctx_log_n(h());
        }
        void verify(const Ciphertext &c, const char *name) const
        {
            if (!c.d_data() || c.size() < 2 || c.poly_modulus_degree() != context_.poly_modulus_degree() ||
                c.coeff_modulus_size() == 0 || c.coeff_modulus_size() > context_.key_modulus_size() ||
                (c.context() && c.context()->handle() != h()))
                throw std::invalid_argument(std::string(name) + " is not valid for encryption parameters");
        }
        void check_pair(const Ciphertext &a, const Ciphertext &b) const
        {
            verify(a, "encrypted1");
            verify(b, "encrypted2");
            if (a.coeff_modulus_size() != b.coeff_modulus_size())
                throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
            if (a.is_ntt_form() != b.is_ntt_form())
                throw std::invalid_argument("NTT form mismatch");
            if (!detail::are_close(a.scale(), b.scale()))
                throw std::invalid_argument("scale mismatch");
        }
        // balance_correction_factors (evaluator.cpp:56-118): {new factor, e1, e2} with e1 f1 = e2 f2 (mod t) and
        // |e1| + |e2| smallest among the remainders of the extended Euclidean run on (t, f2 / f1)
        std::array<std::uint64_t, 3> balance_correction_factors(std::uint64_t factor1, std::uint64_t factor2) const
        {
            using u128 = unsigned __int128;
            const std::uint64_t t = context_.parms().plain_modulus(), half_t = t / 2;
            auto bal = [&](std::uint64_t x) { return (std::int64_t)(x > half_t ? x - t : x); };
            auto sum_abs = [&](std::uint64_t x, std::uint64_t y) { return std::llabs(bal(x)) + std::llabs(bal(y)); };
            auto gcd = [](std::uint64_t a, std::uint64_t b) {
                while (b)
                {
                    const std::uint64_t r = a % b;
                    a = b;
                    b = r;
                }
                return a;
            };
            // f1^-1 mod t by the extended Euclidean algorithm
            std::uint64_t ratio = 0;
            {
                __int128 r0 = t, r1 = factor1 % t, t0 = 0, t1 = 1;
                while (r1)
                {
                    const __int128 q = r0 / r1, r2 = r0 - q * r1, t2 = t0 - q * t1;
                    r0 = r1, r1 = r2, t0 = t1, t1 = t2;
                }
                if (r0 != 1)
                    throw std::logic_error("invalid correction factor1");
                ratio = (std::uint64_t)(t0 < 0 ? t0 + t : t0);
            }
            ratio = (std::uint64_t)((u128)ratio * factor2 % t);
            std::uint64_t e1 = ratio, e2 = 1;
            std::int64_t sum = sum_abs(e1, e2);
            std::int64_t prev_a = (std::int64_t)t, prev_b = 0, a = (std::int64_t)ratio, b = 1;
            while (a != 0)
            {
                const std::int64_t q = prev_a / a;
                std::int64_t temp = prev_a % a;
                prev_a = a;
                a = temp;
                temp = prev_b - b * q;
                prev_b = b;
                b = temp;
                std::uint64_t a_mod = (std::uint64_t)std::llabs(a) % t;
                if (a < 0 && a_mod)
                    a_mod = t - a_mod;
                std::uint64_t b_mod = (std::uint64_t)std::llabs(b) % t;
                if (b < 0 && b_mod)
                    b_mod = t - b_mod;
                if (a_mod != 0 && gcd(a_mod, t) == 1)
                {
                    const std::int64_t new_sum = sum_abs(a_mod, b_mod);
                    if (new_sum < sum)
                    {
                        sum = new_sum;
                        e1 = a_mod;
                        e2 = b_mod;
                    }
                }
            }
            return { (std::uint64_t)((u128)e1 * factor1 % t), e1, e2 };
        }
        void add_sub(Ciphertext &encrypted1, const Ciphertext &encrypted2, bool sub) const
        {
            check_pair(encrypted1, encrypted2);
            const std::uint32_t L = (std::uint32_t)encrypted1.coeff_modulus_size();
            if (encrypted1.correction_factor() != encrypted2.correction_factor())
            {
                const auto factors = balance_correction_factors(encrypted1.correction_factor(), encrypted2.correction_factor());
                const std::vector<std::uint64_t> f1(L, factors[1]), f2(L, factors[2]);
                detail::check(cat_mul_scalar(h(), encrypted1.d_data(), f1.data(), encrypted1.d_data(), nullptr, L,
                                             (std::uint32_t)encrypted1.size(), nullptr));
                Ciphertext encrypted2_copy(encrypted2);
                detail::check(cat_mul_scalar(h(), encrypted2_copy.d_data(), f2.data(), encrypted2_copy.d_data(), nullptr, L,
                                             (std::uint32_t)encrypted2_copy.size(), nullptr));
                encrypted1.correction_factor() = factors[0];
                encrypted2_copy.correction_factor() = factors[0];
                add_sub(encrypted1, encrypted2_copy, sub);
                return;
            }
            const std::size_t size1 = encrypted1.size(), size2 = encrypted2.size(), min_count = std::min(size1, size2);
            if (size1 < size2)
                encrypted1.resize(size2); // keeps the polynomials it has (ciphertext.cpp:76-107)
            auto fn = sub ? cat_sub : cat_add;
            detail::check(fn(h(), encrypted1.d_data(), encrypted2.d_data(), encrypted1.d_data(), nullptr, L,
                             (std::uint32_t)min_count, nullptr));
            if (size1 < size2)
            {
                // the rest of encrypted2: copied (reducing a reduced value copies it) or negated
                const std::uint64_t *src = encrypted2.d_data() + min_count * L * context_.poly_modulus_degree();
                if (sub)
                    detail::check(cat_negate(h(), src, encrypted1.d_data(size1), nullptr, L, (std::uint32_t)(size2 - size1), nullptr));
                else
                    detail::check(cat_modulo(h(), src, encrypted1.d_data(size1), nullptr, L, (std::uint32_t)(size2 - size1), nullptr));
            }
            detail::check(cat_stream_sync(h(), nullptr));
        }
        std::uint64_t mul_mod_t(std::uint64_t a, std::uint64_t b) const
        {
            return (std::uint64_t)((unsigned __int128)a * b % context_.parms().plain_modulus());
        }
        // ckks_multiply / bgv_multiply for other sizes (evaluator.cpp:664-700, 800-836): output polynomial I is the sum
        // of the dyadic products x[a] y[I - a]
        void multiply_general(Ciphertext &encrypted1, const Ciphertext &encrypted2) const
        {
            if (!encrypted1.is_ntt_form() || !encrypted2.is_ntt_form())
                throw std::invalid_argument("encrypted1 or encrypted2 must be in NTT form");
            const std::size_t xs = encrypted1.size(), ys = encrypted2.size(), dest_size = xs + ys - 1;
            if (dest_size > 16) // SEAL_CIPHERTEXT_SIZE_MAX
                throw std::logic_error("invalid parameters");
            const std::uint32_t L = (std::uint32_t)encrypted1.coeff_modulus_size();
            const std::size_t poly = (std::size_t)L * context_.poly_modulus_degree();
            Ciphertext dest(context_, L, dest_size), prod(context_, L, 2);
            for (std::size_t I = 0; I < dest_size; ++I)
            {
                const std::size_t last1 = std::min(I, xs - 1), first2 = std::min(I, ys - 1), first1 = I - first2;
                for (std::size_t a = first1; a <= last1; ++a)
                {
                    std::uint64_t *out = a == first1 ? dest.d_data(I) : prod.d_data();
                    detail::check(cat_dyadic(h(), encrypted1.d_data() + a * poly, encrypted2.d_data() + (I - a) * poly, out,
                                             nullptr, L, 1, nullptr));
                    if (a != first1)
                        detail::check(cat_add(h(), dest.d_data(I), prod.d_data(), dest.d_data(I), nullptr, L, 1, nullptr));
                }
            }
            detail::check(cat_stream_sync(h(), nullptr));
            dest.is_ntt_form() = true;
            if (context_.parms().scheme() == scheme_type::bgv)
            {
                dest.scale() = encrypted1.scale();
                dest.correction_factor() = mul_mod_t(encrypted1.correction_factor(), encrypted2.correction_factor());
            }
            else
                dest.scale() = encrypted1.scale() * encrypted2.scale();
            encrypted1 = std::move(dest);
        }
        // BFV / BGV branches of add_plain / sub_plain (coefficient-form plaintext modulo t, evaluator.cpp:1737-1951);
        // defined in seal_hip_crypto.hpp next to the plaintext scaling / lifting it shares with Encryptor
        void plain_addsub_coeff(Ciphertext &encrypted, const Plaintext &plain, bool sub) const;
        void multiply_plain_coeff(Ciphertext &encrypted, const Plaintext &plain) const;

    public:
        // evaluator.cpp:2174-2265: a coefficient-form plaintext modulo t lifted to the level parms_id names and
        // transformed (defined in seal_hip_crypto.hpp like the branches above)
        void transform_to_ntt_inplace(Plaintext &plain, const parms_id_type &parms_id) const;
        void transform_to_ntt(const Plaintext &plain, const parms_id_type &parms_id, Plaintext &destination_ntt) const
        {
            destination_ntt = plain;
            transform_to_ntt_inplace(destination_ntt, parms_id);
        }

    private:
        void plain_addsub(Ciphertext &encrypted, const Plaintext &plain, bool sub) const
        {
            verify(encrypted, "encrypted");
            if (context_.parms().scheme() != scheme_type::ckks)
            {
                plain_addsub_coeff(encrypted, plain, sub);
                return;
            }
            if (!encrypted.is_ntt_form() || !plain.is_ntt_form())
                throw std::invalid_argument("NTT form mismatch");
            if (encrypted.coeff_modulus_size() != plain.coeff_modulus_size())
                throw std::invalid_argument("encrypted and plain parameter mismatch");
            if (!detail::are_close(encrypted.scale(), plain.scale()))
                throw std::invalid_argument("scale mismatch");
            auto fn = sub ? cat_sub_plain : cat_add_plain;
            detail::check(fn(h(), encrypted.d_data(), plain.d_data(), 0, encrypted.d_data(),
                             (std::uint32_t)encrypted.coeff_modulus_size(), (std::uint32_t)encrypted.size(), 1, nullptr));
        }
        // mod_switch_scale_to_next (evaluator.cpp:1201-1280)
        void scale_to_next(const Ciphertext &encrypted, Ciphertext &destination) const
        {
            const std::uint32_t L = (std::uint32_t)encrypted.coeff_modulus_size();
            Ciphertext dest(context_, L - 1, encrypted.size());
            detail::check(cat_rescale(h(), encrypted.d_data(), dest.d_data(), L, (std::uint32_t)encrypted.size(), 1, nullptr));
            scaled_metadata(encrypted, dest, L);
            destination = std::move(dest);
        }
        // what mod_switch_scale_to_next leaves in the destination's header (evaluator.cpp:1256-1279)
        void scaled_metadata(const Ciphertext &encrypted, Ciphertext &dest, std::uint32_t L) const
        {
            dest.is_ntt_form() = encrypted.is_ntt_form();
            dest.scale() = encrypted.scale();
            if (context_.parms().scheme() == scheme_type::ckks)
                dest.scale() = encrypted.scale() / (double)cat_ctx_modulus(h(), L - 1);
            else if (context_.parms().scheme() == scheme_type::bgv) // :1274-1279
                dest.correction_factor() = mul_mod_t(encrypted.correction_factor(), cat_bgv_inv_q_last_mod_t(h(), L));
            else
                dest.correction_factor() = encrypted.correction_factor();
        }
        void rotate_internal(Ciphertext &encrypted, int steps, const GaloisKeys &galois_keys) const
        {
            if (!context_.using_keyswitching())
                throw std::logic_error("encryption parameters do not support batching");
            if (steps == 0)
                return;
            const std::size_t slots = context_.poly_modulus_degree() >> 1;
            const std::uint32_t elt = cat_galois_elt_from_step(log_n(), steps);
            if (elt == 0)
                throw std::invalid_argument("step count too large");
            if (galois_keys.has_key(elt))
            {
                apply_galois_inplace(encrypted, elt, galois_keys);
                return;
            }
            // NAF of steps; every term must have its power-of-two key (evaluator.cpp:2501-2519)
            std::vector<int> naf;
            {
                long long v = steps;
                int bit = 0;
                const bool neg = v < 0;
                if (neg)
                    v = -v;
                while (v)
                {
                    int z = 0;
                    if (v & 1)
                    {
                        z = 2 - (int)(v & 3);
                        v -= z;
                    }
                    if (z)
                        naf.push_back((neg ? -z : z) * (1 << bit));
                    v >>= 1;
                    ++bit;
                }
            }
            if (naf.size() == 1)
                throw std::invalid_argument("Galois key not present");
            for (int step : naf)
                if ((std::size_t)std::abs(step) != slots)
                    rotate_internal(encrypted, step, galois_keys);
        }

        SEALContext context_;
    };
} // namespace seal
