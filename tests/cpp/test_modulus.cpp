// Host-only check of the facade's seal::Modulus / CoeffModulus / PlainModulus (cat_amd/host/seal_hip_modulus.hpp)
// against the known answers the reference's own unit tests hold (native/tests/seal/modulus.cpp:15-69 CreateModulus,
// :160-182 Reduce, :184-226 CustomExceptionTest, :229-299 CustomTest) and its default tables (util/globals.cpp:24-70,
// util/hestdparms.h).  Only the expected VALUES are taken from there.  No GPU, no library: g++ alone.
#include "../../cat_amd/host/seal_hip_modulus.hpp"

#include <cstdio>
#include <functional>

using namespace seal;

static int failures = 0;
#define EXPECT(c)                                                       \
    do                                                                  \
    {                                                                   \
        if (!(c))                                                       \
        {                                                               \
            std::printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #c);  \
            ++failures;                                                 \
        }                                                               \
    } while (0)

template <class E>
static bool throws(const std::function<void()> &f)
{
    try
    {
        f();
    }
    catch (const E &)
    {
        return true;
    }
    catch (...)
    {
        return false;
    }
    return false;
}

static int bits(std::uint64_t v)
{
    int b = 0;
    while (v)
        ++b, v >>= 1;
    return b;
}

static void create_modulus()
{
    Modulus mod;
    EXPECT(mod.is_zero() && mod.value() == 0 && mod.bit_count() == 0 && mod.uint64_count() == 1);
    EXPECT(mod.const_ratio()[0] == 0 && mod.const_ratio()[1] == 0 && mod.const_ratio()[2] == 0 && !mod.is_prime());

    mod = 3;
    EXPECT(!mod.is_zero() && mod.value() == 3 && mod.bit_count() == 2 && mod.uint64_count() == 1);
    EXPECT(mod.const_ratio()[0] == 6148914691236517205ULL && mod.const_ratio()[1] == 6148914691236517205ULL &&
           mod.const_ratio()[2] == 1 && mod.is_prime());

    Modulus mod2(2), mod3(3);
    EXPECT(mod.value() != mod2.value() && mod.value() == mod3.value());

    mod = 0;
    EXPECT(mod.is_zero() && mod.bit_count() == 0 && mod.const_ratio()[0] == 0 && mod.const_ratio()[1] == 0 &&
           mod.const_ratio()[2] == 0);

    mod = 0xF00000F00000FULL;
    EXPECT(mod.value() == 0xF00000F00000FULL && mod.bit_count() == 52 && mod.uint64_count() == 1);
    EXPECT(mod.const_ratio()[0] == 1224979098644774929ULL && mod.const_ratio()[1] == 4369ULL &&
           mod.const_ratio()[2] == 281470698520321ULL && !mod.is_prime());

    mod = 0xF00000F000079ULL;
    EXPECT(mod.bit_count() == 52);
    EXPECT(mod.const_ratio()[0] == 1224979096621368355ULL && mod.const_ratio()[1] == 4369ULL &&
           mod.const_ratio()[2] == 1144844808538997ULL && mod.is_prime());

    // values out of range (modulus.cpp:66-75: at most 61 bits, not 1)
    EXPECT(throws<std::invalid_argument>([] { Modulus m(1); }));
    EXPECT(throws<std::invalid_argument>([] { Modulus m(std::uint64_t(1) << 61); }));
}

static void reduce()
{
    Modulus zero;
    EXPECT(throws<std::logic_error>([&] { (void)zero.reduce(10); }));
    Modulus two(2);
    EXPECT(two.reduce(0) == 0 && two.reduce(1) == 1 && two.reduce(2) == 0 && two.reduce(0xF0F0F0) == 0);
    Modulus ten(10);
    EXPECT(ten.reduce(0) == 0 && ten.reduce(1) == 1 && ten.reduce(8) == 8 && ten.reduce(1234567) == 7 &&
           ten.reduce(12345670) == 0);
}

static void coeff_modulus_exceptions()
{
    typedef std::invalid_argument IA;
    typedef std::logic_error LE;
    EXPECT(throws<IA>([] { CoeffModulus::Create(1, { 2 }); }));
    EXPECT(throws<IA>([] { CoeffModulus::Create(262144, { 30 }); }));
    EXPECT(throws<IA>([] { CoeffModulus::Create(1023, { 20 }); }));
    EXPECT(throws<IA>([] { CoeffModulus::Create(2048, { 0 }); }));
    EXPECT(throws<IA>([] { CoeffModulus::Create(2048, { -30 }); }));
    EXPECT(throws<IA>([] { CoeffModulus::Create(2048, { 30, -30 }); }));
    EXPECT(throws<LE>([] { CoeffModulus::Create(2, { 2 }); }));
    EXPECT(throws<LE>([] { CoeffModulus::Create(2, { 3, 3, 3 }); }));
    EXPECT(throws<LE>([] { CoeffModulus::Create(1024, { 8 }); }));
    EXPECT(throws<IA>([] { CoeffModulus::Create(1, Modulus(2), { 2 }); }));
    EXPECT(throws<IA>([] { CoeffModulus::Create(262144, Modulus(2), { 30 }); }));
    EXPECT(throws<IA>([] { CoeffModulus::Create(1023, Modulus(2), { 20 }); }));
    EXPECT(throws<IA>([] { CoeffModulus::Create(2048, Modulus(2), { 0 }); }));
    EXPECT(throws<IA>([] { CoeffModulus::Create(2048, Modulus(2), { -30 }); }));
    EXPECT(throws<IA>([] { CoeffModulus::Create(2048, Modulus(2), { 30, -30 }); }));
    EXPECT(throws<LE>([] { CoeffModulus::Create(2048, Modulus(std::uint64_t(1) << 53), { 20 }); }));
    EXPECT(throws<LE>([] { CoeffModulus::Create(2, Modulus(2), { 2 }); }));
    EXPECT(throws<LE>([] { CoeffModulus::Create(2, Modulus(30), { 6, 6 }); }));
    EXPECT(throws<LE>([] { CoeffModulus::Create(1024, Modulus(257), { 20 }); }));
    EXPECT(throws<LE>([] { CoeffModulus::Create(1024, Modulus(255), { 22, 22, 22 }); }));
}

static void coeff_modulus_values()
{
    auto values = [](const std::vector<Modulus> &v) {
        std::vector<std::uint64_t> r;
        for (const Modulus &m : v)
            r.push_back(m.value());
        return r;
    };
    typedef std::vector<std::uint64_t> V;
    EXPECT(CoeffModulus::Create(2, {}).empty());
    EXPECT(values(CoeffModulus::Create(2, { 3 })) == V({ 5 }));
    EXPECT(values(CoeffModulus::Create(2, { 3, 4 })) == V({ 5, 13 }));
    EXPECT(values(CoeffModulus::Create(2, { 3, 5, 4, 5 })) == V({ 5, 17, 13, 29 }));
    EXPECT(CoeffModulus::Create(2, Modulus(4), {}).empty());
    EXPECT(values(CoeffModulus::Create(2, Modulus(4), { 3 })) == V({ 5 }));
    EXPECT(values(CoeffModulus::Create(2, Modulus(4), { 3, 4 })) == V({ 5, 13 }));
    EXPECT(values(CoeffModulus::Create(2, Modulus(4), { 3, 5, 4, 5 })) == V({ 5, 17, 13, 29 }));
    for (int with_plain = 0; with_plain < 2; ++with_plain)
    {
        const std::vector<Modulus> cm = with_plain ? CoeffModulus::Create(32, Modulus(64), { 30, 40, 30, 30, 40 })
                                                   : CoeffModulus::Create(32, { 30, 40, 30, 30, 40 });
        const int want[5] = { 30, 40, 30, 30, 40 };
        EXPECT(cm.size() == 5);
        for (std::size_t i = 0; i < cm.size() && i < 5; ++i)
            EXPECT(bits(cm[i].value()) == want[i] && cm[i].value() % 64 == 1 && cm[i].is_prime());
        EXPECT(cm[0].value() != cm[2].value() && cm[2].value() != cm[3].value() && cm[1].value() != cm[4].value());
    }
    EXPECT(values(CoeffModulus::Create(1024, Modulus(255), { 22, 22 })) == V({ 3133441ULL, 3655681ULL }));
}

static void defaults()
{
    // util/hestdparms.h (classical 128 / 192 / 256 bit) and util/globals.cpp:24-70
    EXPECT(CoeffModulus::MaxBitCount(1024) == 27 && CoeffModulus::MaxBitCount(2048) == 54 &&
           CoeffModulus::MaxBitCount(4096) == 109 && CoeffModulus::MaxBitCount(8192) == 218 &&
           CoeffModulus::MaxBitCount(16384) == 438 && CoeffModulus::MaxBitCount(32768) == 881);
    EXPECT(CoeffModulus::MaxBitCount(32768, sec_level_type::tc192) == 611 &&
           CoeffModulus::MaxBitCount(32768, sec_level_type::tc256) == 476 &&
           CoeffModulus::MaxBitCount(4096, sec_level_type::tc192) == 75);
    const std::size_t degrees[] = { 1024, 2048, 4096, 8192, 16384, 32768 };
    const std::size_t counts[] = { 1, 1, 3, 5, 9, 16 };
    for (int i = 0; i < 6; ++i)
    {
        const std::vector<Modulus> d = CoeffModulus::BFVDefault(degrees[i]);
        EXPECT(d.size() == counts[i]);
        int total = 0;
        for (const Modulus &m : d)
        {
            EXPECT(m.is_prime() && m.value() % (2 * degrees[i]) == 1);
            total += m.bit_count();
        }
        EXPECT(total == CoeffModulus::MaxBitCount(degrees[i]));
    }
    EXPECT(CoeffModulus::BFVDefault(4096)[0].value() == 0xffffee001ULL);      // globals.cpp:38-43
    EXPECT(CoeffModulus::BFVDefault(8192)[0].value() == 0x7fffffd8001ULL &&
           CoeffModulus::BFVDefault(8192)[4].value() == 0xfffffebc001ULL); // globals.cpp:45-50
    EXPECT(throws<std::invalid_argument>([] { CoeffModulus::BFVDefault(3000); }));
    const Modulus t = PlainModulus::Batching(4096, 20);
    EXPECT(t.is_prime() && t.bit_count() == 20 && t.value() % 8192 == 1);
}

int main()
{
    create_modulus();
    reduce();
    coeff_modulus_exceptions();
    coeff_modulus_values();
    defaults();
    if (failures)
    {
        std::printf("modulus test: %d failure(s)\n", failures);
        return 1;
    }
    std::printf("modulus ok: Modulus / CoeffModulus / PlainModulus reproduce the reference's known answers\n");
    return 0;
}
