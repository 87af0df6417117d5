// seal_hip_serialization.hpp -- the reference's wire format for the facade's objects, so that
// ciphertexts and keys written by the reference's CPU build load onto the device path and the other way round:
// SEALHeader and Save/Load (serialization.h:58-160, serialization.cpp:137-452), DynArray (dynarray.h:664-730),
// UniformRandomGeneratorInfo (randomgen.cpp:99-158), parms_id (encryptionparams.cpp:124-158, util/hash.h).
// The object layouts themselves are written out next to the classes in seal_hip.hpp / seal_hip_crypto.hpp.
// Uncompressed only: this fork builds with zlib and zstd off (CMakeLists.txt:175,196), so compr_mode_default is
// `none` and its loader rejects anything else with "unsupported compression mode" -- as this one does.  Host code.
#pragma once

#include <array>
#include <cstdint>
#include <cstring>
#include <functional>
#include <istream>
#include <ostream>
#include <stdexcept>
#include <vector>

#include "seal_hip_random.hpp"

namespace seal
{
    using parms_id_type = std::array<std::uint64_t, 4>; // util/hash.h hash_block_type
    constexpr parms_id_type parms_id_zero{ { 0, 0, 0, 0 } };

    enum class compr_mode_type : std::uint8_t // serialization.h:27-40
    {
        none = 0,
        zlib = 1,
        zstd = 2
    };

    struct SEALVersion
    {
        std::uint8_t major = 4, minor = 0, patch = 0, tweak = 0;
    };

    class Serialization
    {
    public:
        static constexpr compr_mode_type compr_mode_default = compr_mode_type::none;
        static constexpr std::uint16_t seal_magic = 0xA15E;
        static constexpr std::uint8_t seal_header_size = 0x10;
        static constexpr std::uint8_t version_major = 4, version_minor = 0; // CMakeLists.txt:37

        struct SEALHeader // serialization.h:76-94: 16 bytes
        {
            std::uint16_t magic = seal_magic;
            std::uint8_t header_size = seal_header_size;
            std::uint8_t version_major = Serialization::version_major;
            std::uint8_t version_minor = Serialization::version_minor;
            compr_mode_type compr_mode = compr_mode_type::none;
            std::uint16_t reserved = 0;
            std::uint64_t size = 0; // the whole object including this header
        };
        static_assert(sizeof(SEALHeader) == seal_header_size, "");

        static bool IsSupportedComprMode(std::uint8_t compr_mode) noexcept
        {
            return compr_mode == static_cast<std::uint8_t>(compr_mode_type::none);
        }
        static bool IsCompatibleVersion(const SEALHeader &header) noexcept // serialization.h:144-161
        {
            if (header.version_major == version_major && header.version_minor == version_minor)
                return true;
            if (header.version_major != version_major && header.version_major != 3)
                return false;
            return header.version_major == 3 && header.version_minor >= 4;
        }
        static bool IsValidHeader(const SEALHeader &header) noexcept // serialization.h:168-190
        {
            return header.magic == seal_magic && header.header_size == seal_header_size && IsCompatibleVersion(header) &&
                   IsSupportedComprMode(static_cast<std::uint8_t>(header.compr_mode));
        }

        // serialization.cpp:224-338, compr_mode none: header with the final size, then the members
        static std::streamoff Save(const std::function<void(std::ostream &)> &save_members, std::streamoff raw_size,
                                   std::ostream &stream, compr_mode_type compr_mode = compr_mode_default)
        {
            if (!save_members)
                throw std::invalid_argument("save_members is invalid");
            if (raw_size < static_cast<std::streamoff>(sizeof(SEALHeader)))
                throw std::invalid_argument("raw_size is too small");
            if (!IsSupportedComprMode(static_cast<std::uint8_t>(compr_mode)))
                throw std::invalid_argument("unsupported compression mode");
            Guard guard(stream);
            try
            {
                const auto start = stream.tellp();
                SEALHeader header;
                header.size = static_cast<std::uint64_t>(raw_size);
                stream.write(reinterpret_cast<const char *>(&header), sizeof(header));
                save_members(stream);
                const std::streamoff written = stream.tellp() - start;
                if (written != raw_size)
                    throw std::logic_error("invalid data size");
                return written;
            }
            catch (const std::ios_base::failure &)
            {
                throw std::runtime_error("I/O error");
            }
        }

        // serialization.cpp:340-452
        static std::streamoff Load(const std::function<void(std::istream &, SEALVersion)> &load_members, std::istream &stream)
        {
            if (!load_members)
                throw std::invalid_argument("load_members is invalid");
            Guard guard(stream);
            try
            {
                const auto start = stream.tellg();
                SEALHeader header;
                stream.read(reinterpret_cast<char *>(&header), sizeof(header));
                if (!IsCompatibleVersion(header))
                    throw std::logic_error("incompatible version");
                if (header.magic != seal_magic || header.header_size != seal_header_size)
                    throw std::logic_error("loaded SEALHeader is invalid");
                if (!IsSupportedComprMode(static_cast<std::uint8_t>(header.compr_mode)))
                    throw std::invalid_argument("unsupported compression mode");
                SEALVersion version;
                version.major = header.version_major;
                version.minor = header.version_minor;
                load_members(stream, version);
                if (header.size != static_cast<std::uint64_t>(stream.tellg() - start))
                    throw std::logic_error("invalid data size");
                return static_cast<std::streamoff>(header.size);
            }
            catch (const std::ios_base::failure &)
            {
                throw std::runtime_error("I/O error");
            }
        }

    private:
        // streams throw on failure while an object is being read or written (the reference's exception masks)
        template <class Stream>
        struct Guard
        {
            explicit Guard(Stream &s) : s_(s), old_(s.exceptions())
            {
                s_.exceptions(std::ios_base::badbit | std::ios_base::failbit);
            }
            ~Guard()
            {
                try
                {
                    s_.exceptions(old_);
                }
                catch (...)
                {}
            }
            Stream &s_;
            std::ios_base::iostate old_;
        };
    };

    namespace util
    {
        // EncryptionParameters::compute_parms_id (encryptionparams.cpp:124-158): BLAKE2b-256 of the words
        // [scheme, N, q_0 .. q_{L-1}, t] (Modulus(0) still counts one word, modulus.cpp:69-73)
        inline parms_id_type compute_parms_id(std::uint8_t scheme, std::size_t poly_modulus_degree, const std::uint64_t *q,
                                              std::size_t coeff_modulus_size, std::uint64_t plain_modulus)
        {
            std::vector<std::uint64_t> words;
            words.push_back(scheme);
            words.push_back(poly_modulus_degree);
            words.insert(words.end(), q, q + coeff_modulus_size);
            words.push_back(plain_modulus);
            parms_id_type id;
            blake2b(id.data(), sizeof(id), words.data(), words.size() * 8, nullptr, 0);
            if (id == parms_id_zero)
                throw std::logic_error("parms_id cannot be zero");
            return id;
        }

        template <typename T>
        inline void write_pod(std::ostream &s, const T &v)
        {
            s.write(reinterpret_cast<const char *>(&v), sizeof(T));
        }
        template <typename T>
        inline T read_pod(std::istream &s)
        {
            T v;
            s.read(reinterpret_cast<char *>(&v), sizeof(T));
            return v;
        }

        // DynArray<uint64_t>::save / load (dynarray.h:580-730): an object of its own = header, count, words
        inline std::streamoff dynarray_save_size(std::size_t count)
        {
            return static_cast<std::streamoff>(sizeof(Serialization::SEALHeader) + 8 + count * 8);
        }
        inline void dynarray_save(std::ostream &stream, const std::uint64_t *data, std::size_t count)
        {
            Serialization::Save(
                [&](std::ostream &s) {
                    write_pod<std::uint64_t>(s, count);
                    if (count)
                        s.write(reinterpret_cast<const char *>(data), static_cast<std::streamsize>(count * 8));
                },
                dynarray_save_size(count), stream);
        }
        inline std::vector<std::uint64_t> dynarray_load(std::istream &stream, std::size_t in_size_bound)
        {
            std::vector<std::uint64_t> out;
            Serialization::Load(
                [&](std::istream &s, SEALVersion) {
                    const std::uint64_t size64 = read_pod<std::uint64_t>(s);
                    if (in_size_bound && size64 > in_size_bound)
                        throw std::logic_error("unexpected size");
                    out.resize(static_cast<std::size_t>(size64));
                    if (size64)
                        s.read(reinterpret_cast<char *>(out.data()), static_cast<std::streamsize>(size64 * 8));
                },
                stream);
            return out;
        }
    } // namespace util

    // randomgen.h:38-290, randomgen.cpp:99-174: generator type and seed, what a seeded ciphertext carries instead
    // of its second polynomial
    class UniformRandomGeneratorInfo
    {
    public:
        UniformRandomGeneratorInfo() = default;
        UniformRandomGeneratorInfo(prng_type type, const prng_seed_type &seed) : type_(type), seed_(seed)
        {}
        prng_type type() const noexcept
        {
            return type_;
        }
        const prng_seed_type &seed() const noexcept
        {
            return seed_;
        }
        bool has_valid_prng_type() const noexcept
        {
            return type_ == prng_type::unknown || type_ == prng_type::blake2xb || type_ == prng_type::shake256;
        }
        static std::streamoff SaveSize()
        {
            return static_cast<std::streamoff>(sizeof(Serialization::SEALHeader) + sizeof(prng_type) + prng_seed_byte_count);
        }
        std::streamoff save(std::ostream &stream) const
        {
            return Serialization::Save(
                [&](std::ostream &s) {
                    util::write_pod(s, type_);
                    s.write(reinterpret_cast<const char *>(seed_.data()), prng_seed_byte_count);
                },
                SaveSize(), stream);
        }
        std::streamoff load(std::istream &stream)
        {
            return Serialization::Load(
                [&](std::istream &s, SEALVersion) {
                    UniformRandomGeneratorInfo info;
                    info.type_ = util::read_pod<prng_type>(s);
                    if (!info.has_valid_prng_type())
                        throw std::logic_error("prng_type is invalid");
                    s.read(reinterpret_cast<char *>(info.seed_.data()), prng_seed_byte_count);
                    *this = info;
                },
                stream);
        }
        // make_prng, randomgen.cpp:160-174 (the SHAKE256 generator is not restated)
        std::shared_ptr<UniformRandomGenerator> make_prng() const
        {
            if (type_ == prng_type::blake2xb)
                return std::make_shared<Blake2xbPRNG>(seed_);
            return nullptr;
        }

    private:
        prng_type type_ = prng_type::unknown;
        prng_seed_type seed_{};
    };
} // namespace seal
