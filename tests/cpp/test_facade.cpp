// Parity test of the C++ facade (cat_amd/host/seal_hip.hpp) written the way the reference's own tests drive
// seal::Evaluator (native/tests/seal/evaluator.cpp:3354-3623: multiply, relinearize, rescale, rotate), but
// bit-exact against the CPU oracle instead of decrypt-and-compare with a tolerance.  Needs a GPU; built by
// `make facade_test`, run by tests/test_gpu_facade.py.  Test infrastructure: links oracle/libseal_oracle.so.
#include "../../cat_amd/host/seal_hip_crypto.hpp"
#include "../../oracle/seal_oracle.h"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <sstream>
#include <vector>

using namespace seal;
typedef std::vector<uint64_t> vec;

static int failures = 0;
#define EXPECT(cond)                                                          \
    do                                                                        \
    {                                                                         \
        if (!(cond))                                                          \
        {                                                                     \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);       \
            ++failures;                                                       \
        }                                                                     \
    } while (0)

template <typename E>
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

static vec uniform(so_context *oc, size_t L, size_t polys, size_t n, uint64_t seed)
{
    vec v(polys * L * n);
    so_fill_uniform(oc, L, polys, seed, v.data());
    return v;
}

static void test_util_ntt_known_answers()
{
    // native/tests/seal/util/ntt.cpp:77-101
    EncryptionParameters parms(scheme_type::none);
    parms.set_poly_modulus_degree(2);
    parms.set_coeff_modulus({ 0xffffffffffc0001ULL });
    SEALContext context(parms);
    util::NTTTables tables(context, 0);
    Ciphertext buf(context, 1, 2); // 2 polys x 1 limb x 2 words of device memory
    const uint64_t cases[3][4] = { { 0, 0, 0, 0 }, { 1, 0, 1, 1 }, { 1, 1, 288794978602139553ULL, 864126526004445282ULL } };
    for (auto &c : cases)
    {
        vec h = { c[0], c[1], 0, 0 };
        buf.upload(h.data());
        util::ntt_negacyclic_harvey(buf.d_data(), tables);
        h = buf.to_cpu();
        EXPECT(h[0] == c[2] && h[1] == c[3]);
        util::inverse_ntt_negacyclic_harvey(buf.d_data(), tables);
        h = buf.to_cpu();
        EXPECT(h[0] == c[0] && h[1] == c[1]);
    }
}

static void test_ckks()
{
    const int log_n = 12;
    const size_t n = 4096;
    const int bits[4] = { 50, 40, 40, 50 };
    vec primes(4);
    EXPECT(so_coeff_modulus_create(n, bits, 4, primes.data()) == 0);
    const size_t k = 4, L = 3;
    so_context *oc = so_context_create(log_n, primes.data(), k, SO_SCHEME_CKKS, 0);

    EncryptionParameters parms(scheme_type::ckks);
    parms.set_poly_modulus_degree(n);
    parms.set_coeff_modulus(primes);
    SEALContext context(parms);
    Evaluator evaluator(context);
    EXPECT(context.first_level_size() == L);

    vec sk(k * n), rk((k - 1) * 2 * k * n), gk1(rk.size()), gk2(rk.size()), gkc(rk.size());
    so_sample_ternary_ntt(oc, 1, sk.data());
    so_gen_relin_key(oc, sk.data(), 2, rk.data());
    const uint32_t elt1 = so_galois_elt_from_step(log_n, 1), elt2 = so_galois_elt_from_step(log_n, 2),
                   eltc = so_galois_elt_from_step(log_n, 0);
    so_gen_galois_key(oc, sk.data(), elt1, 3, gk1.data());
    so_gen_galois_key(oc, sk.data(), elt2, 4, gk2.data());
    so_gen_galois_key(oc, sk.data(), eltc, 5, gkc.data());
    RelinKeys relin_keys(context);
    relin_keys.set_key(RelinKeys::get_index(2), rk.data());
    GaloisKeys galois_keys(context);
    galois_keys.set_key(GaloisKeys::get_index(elt1), gk1.data());
    galois_keys.set_key(GaloisKeys::get_index(elt2), gk2.data());
    galois_keys.set_key(GaloisKeys::get_index(eltc), gkc.data());
    EXPECT(relin_keys.has_key(2) && galois_keys.has_key(elt1) && !galois_keys.has_key(5));

    vec x = uniform(oc, L, 2, n, 10), y = uniform(oc, L, 2, n, 11);
    Ciphertext encrypted1(context, L), encrypted2(context, L);
    encrypted1.upload(x.data());
    encrypted2.upload(y.data());
    const double delta = std::pow(2.0, 40);
    encrypted1.scale() = encrypted2.scale() = delta;

    // multiply needs NTT form (evaluator.cpp:571)
    EXPECT(throws<std::invalid_argument>([&] { evaluator.multiply_inplace(encrypted1, encrypted2); }));
    encrypted1.is_ntt_form() = encrypted2.is_ntt_form() = true;

    // add / sub / negate (evaluator.cpp:88-351)
    {
        Ciphertext s(encrypted1);
        evaluator.add_inplace(s, encrypted2);
        vec got = s.to_cpu(), exp(x.size());
        for (size_t p = 0; p < 2; ++p)
            for (size_t j = 0; j < L; ++j)
                so_add_poly_coeffmod(&x[(p * L + j) * n], &y[(p * L + j) * n], n, &so_context_tables(oc, j)->mod,
                                     &exp[(p * L + j) * n]);
        EXPECT(got == exp);
        evaluator.sub_inplace(s, encrypted2);
        EXPECT(s.to_cpu() == x);
        evaluator.negate_inplace(s);
        evaluator.negate_inplace(s);
        EXPECT(s.to_cpu() == x);
    }

    // square, multiply_plain, add_plain / sub_plain (tests/evaluator.cpp:3354-3538 drive the same calls)
    {
        Ciphertext s(encrypted1);
        vec e(3 * L * n, 0);
        std::memcpy(e.data(), x.data(), x.size() * 8);
        so_ckks_square(oc, L, e.data());
        evaluator.square_inplace(s);
        EXPECT(s.size() == 3 && s.to_cpu() == e && s.scale() == delta * delta);
        vec pl = uniform(oc, L, 1, n, 12);
        Plaintext plain(context, L);
        plain.upload(pl.data());
        plain.scale() = delta;
        Ciphertext m(encrypted1);
        vec em = x;
        so_multiply_plain_ntt(oc, L, em.data(), 2, pl.data());
        evaluator.multiply_plain_inplace(m, plain);
        EXPECT(m.to_cpu() == em && m.scale() == delta * delta);
        Ciphertext a(encrypted1);
        vec ea = x;
        so_add_plain_ckks(oc, L, ea.data(), pl.data(), 0);
        evaluator.add_plain_inplace(a, plain);
        EXPECT(a.to_cpu() == ea);
        evaluator.sub_plain_inplace(a, plain);
        EXPECT(a.to_cpu() == x);
        plain.scale() = 2 * delta;
        EXPECT(throws<std::invalid_argument>([&] { evaluator.add_plain_inplace(a, plain); })); // scale mismatch
    }

    // multiply -> relinearize -> rescale (tests/evaluator.cpp:3539-3623)
    vec exp3(3 * L * n, 0);
    std::memcpy(exp3.data(), x.data(), x.size() * 8);
    so_ckks_multiply(oc, L, exp3.data(), y.data());
    evaluator.multiply_inplace(encrypted1, encrypted2);
    EXPECT(encrypted1.size() == 3 && encrypted1.is_ntt_form() && encrypted1.scale() == delta * delta);
    EXPECT(encrypted1.to_cpu() == exp3);
    const Ciphertext encrypted1_size3(encrypted1);

    so_relinearize(oc, L, exp3.data(), rk.data());
    evaluator.relinearize_inplace(encrypted1, relin_keys);
    EXPECT(encrypted1.size() == 2);
    vec exp2(exp3.begin(), exp3.begin() + 2 * L * n);
    EXPECT(encrypted1.to_cpu() == exp2);

    vec expr(2 * (L - 1) * n);
    so_mod_switch_scale_to_next(oc, L, exp2.data(), 2, expr.data());
    Ciphertext rescaled;
    evaluator.rescale_to_next(encrypted1, rescaled);
    EXPECT(rescaled.coeff_modulus_size() == L - 1 && rescaled.to_cpu() == expr);
    EXPECT(rescaled.scale() == delta * delta / (double)primes[L - 1]);

    // the one-call forms (not in the reference): the words and metadata of the two calls they stand for
    {
        Ciphertext a(encrypted1_size3), b(encrypted1_size3);
        evaluator.relinearize_inplace(a, relin_keys);
        evaluator.rescale_to_next_inplace(a);
        evaluator.relinearize_rescale_inplace(b, relin_keys);
        EXPECT(b.size() == 2 && b.coeff_modulus_size() == L - 1 && b.is_ntt_form() && b.scale() == a.scale());
        EXPECT(b.to_cpu() == a.to_cpu() && b.to_cpu() == expr);
        Ciphertext c(encrypted2), d(encrypted2);
        evaluator.rotate_vector_inplace(c, 1, galois_keys);
        evaluator.rescale_to_next_inplace(c);
        evaluator.rotate_vector_rescale_inplace(d, 1, galois_keys);
        EXPECT(d.coeff_modulus_size() == L - 1 && d.scale() == c.scale() && d.to_cpu() == c.to_cpu());
        Ciphertext two(encrypted2), two_ref(encrypted2); // size 2: relinearize is a no-op, then the rescale
        evaluator.rescale_to_next_inplace(two_ref);
        evaluator.relinearize_rescale_inplace(two, relin_keys);
        EXPECT(two.to_cpu() == two_ref.to_cpu() && two.scale() == two_ref.scale());
        EXPECT(throws<std::invalid_argument>([&] { evaluator.rotate_vector_rescale_inplace(d, 3, galois_keys); }));
    }

    // out-of-place and chained forms (evaluator.h): the copies / loops they are in the reference
    {
        Ciphertext a, b;
        evaluator.add(encrypted2, encrypted2, a);
        b = encrypted2;
        evaluator.add_inplace(b, encrypted2);
        EXPECT(a.to_cpu() == b.to_cpu());
        evaluator.sub(encrypted2, a, a); // destination is the second operand: x - 2x = -x
        evaluator.negate(encrypted2, b);
        EXPECT(a.to_cpu() == b.to_cpu());
        const std::vector<Ciphertext> many{ encrypted2, encrypted2, encrypted2 };
        evaluator.add_many(many, a);
        b = encrypted2;
        evaluator.add_inplace(b, encrypted2);
        evaluator.add_inplace(b, encrypted2);
        EXPECT(a.to_cpu() == b.to_cpu());
        EXPECT(throws<std::invalid_argument>([&] { evaluator.add_many(std::vector<Ciphertext>{}, a); }));
        // rescale_to / mod_switch_to walk the chain (evaluator.cpp:1429-1451, 1521-1570)
        Ciphertext c(encrypted2), d(encrypted2);
        evaluator.rescale_to_next_inplace(c);
        evaluator.rescale_to_next_inplace(c);
        evaluator.rescale_to_inplace(d, context.parms_id(L - 2));
        EXPECT(d.coeff_modulus_size() == L - 2 && d.to_cpu() == c.to_cpu() && d.scale() == c.scale());
        EXPECT(throws<std::invalid_argument>([&] { evaluator.rescale_to_inplace(d, context.parms_id(L)); })); // higher level
        Ciphertext m1(encrypted2), m2;
        evaluator.mod_switch_to_next_inplace(m1);
        evaluator.mod_switch_to(encrypted2, context.parms_id(L - 1), m2);
        EXPECT(m2.to_cpu() == m1.to_cpu() && m2.scale() == encrypted2.scale());
        evaluator.mod_switch_to_inplace(m2, m2.parms_id()); // already there
        EXPECT(m2.to_cpu() == m1.to_cpu());
        parms_id_type bogus = context.parms_id(L);
        bogus[0] ^= 1;
        EXPECT(throws<std::invalid_argument>([&] { evaluator.mod_switch_to_inplace(m2, bogus); }));
        Ciphertext sq1(encrypted2), sq2, r1(encrypted2), r2, t1(encrypted2), t2;
        evaluator.square_inplace(sq1);
        evaluator.square(encrypted2, sq2);
        EXPECT(sq1.to_cpu() == sq2.to_cpu() && sq2.size() == 3);
        evaluator.relinearize(sq2, relin_keys, r2);
        evaluator.relinearize_inplace(sq1, relin_keys);
        EXPECT(r2.size() == 2 && r2.to_cpu() == sq1.to_cpu());
        evaluator.rotate_vector_inplace(r1, 2, galois_keys);
        evaluator.rotate_vector(encrypted2, 2, galois_keys, r2);
        EXPECT(r2.to_cpu() == r1.to_cpu());
        evaluator.complex_conjugate(encrypted2, galois_keys, r2);
        evaluator.apply_galois(encrypted2, eltc, galois_keys, r1);
        EXPECT(r2.to_cpu() == r1.to_cpu());
        evaluator.transform_from_ntt_inplace(t1);
        evaluator.transform_from_ntt(encrypted2, t2);
        EXPECT(!t2.is_ntt_form() && t2.to_cpu() == t1.to_cpu());
        evaluator.transform_to_ntt(t2, t1);
        EXPECT(t1.is_ntt_form() && t1.to_cpu() == y);
        // NTT-form plaintexts drop limbs (evaluator.cpp:1347-1382, 1453-1479)
        Plaintext pl(context, L), pl2;
        pl.upload(x.data());
        pl.scale() = delta;
        evaluator.mod_switch_to(pl, context.parms_id(L - 2), pl2);
        EXPECT(pl2.coeff_modulus_size() == L - 2 && pl2.scale() == delta && pl2.parms_id() == context.parms_id(L - 2));
        EXPECT(pl2.to_cpu() == vec(x.begin(), x.begin() + (L - 2) * n));
        evaluator.mod_switch_to_next_inplace(pl);
        EXPECT(pl.to_cpu() == vec(x.begin(), x.begin() + (L - 1) * n));
        EXPECT(throws<std::invalid_argument>([&] { evaluator.mod_switch_to_next_inplace(pl2); })); // end of chain
        EXPECT(throws<std::invalid_argument>([&] { evaluator.mod_switch_to_inplace(pl2, context.parms_id(L)); }));
        EXPECT(throws<std::logic_error>([&] { evaluator.multiply_many(many, relin_keys, a); }));       // CKKS: unsupported scheme
        EXPECT(throws<std::logic_error>([&] { evaluator.rotate_columns_inplace(r1, galois_keys); })); // BFV / BGV only
    }

    // rotate_vector / complex_conjugate (tests/evaluator.cpp:4147, disabled in the reference build)
    {
        Ciphertext r(encrypted2);
        vec e = y;
        so_apply_galois_inplace(oc, L, e.data(), elt1, gk1.data());
        evaluator.rotate_vector_inplace(r, 1, galois_keys);
        EXPECT(r.to_cpu() == e);
        // step 3 has no key: NAF 3 = 4 - 1 needs the key of step 4 -> "Galois key not present"
        EXPECT(throws<std::invalid_argument>([&] { evaluator.rotate_vector_inplace(r, 3, galois_keys); }));
        vec e2 = e;
        so_apply_galois_inplace(oc, L, e2.data(), eltc, gkc.data());
        evaluator.complex_conjugate_inplace(r, galois_keys);
        EXPECT(r.to_cpu() == e2);
        EXPECT(throws<std::logic_error>([&] { evaluator.rotate_rows_inplace(r, 1, galois_keys); }));
        EXPECT(throws<std::invalid_argument>([&] { evaluator.apply_galois_inplace(r, 4, galois_keys); }));
    }

    // multiply of larger operands (evaluator.cpp:664-700): 3 x 2 -> 4, then relinearize 4 -> 2 needs the s^3 key
    {
        vec x3 = uniform(oc, L, 3, n, 13), e(4 * L * n, 0);
        Ciphertext a(context, L, 3);
        a.upload(x3.data());
        a.is_ntt_form() = true;
        a.scale() = delta;
        std::memcpy(e.data(), x3.data(), x3.size() * 8);
        so_ckks_multiply_general(oc, L, e.data(), 3, y.data(), 2);
        evaluator.multiply_inplace(a, encrypted2);
        EXPECT(a.size() == 4 && a.to_cpu() == e && a.scale() == delta * delta);
        EXPECT(throws<std::invalid_argument>([&] { evaluator.relinearize_inplace(a, relin_keys); })); // "not enough relinearization keys"
    }

    // mod_switch_to_next (drop) and the NTT transforms
    {
        Ciphertext m(encrypted2);
        vec e(2 * (L - 1) * n);
        so_mod_switch_drop_to_next(oc, L, y.data(), 2, e.data());
        evaluator.mod_switch_to_next_inplace(m);
        EXPECT(m.coeff_modulus_size() == L - 1 && m.to_cpu() == e);
        Ciphertext t(encrypted2);
        vec f = y;
        so_transform_from_ntt(oc, L, f.data(), 2);
        evaluator.transform_from_ntt_inplace(t);
        EXPECT(!t.is_ntt_form() && t.to_cpu() == f);
        EXPECT(throws<std::invalid_argument>([&] { evaluator.transform_from_ntt_inplace(t); }));
        evaluator.transform_to_ntt_inplace(t);
        EXPECT(t.to_cpu() == y);
        // chain end (evaluator.cpp:1488-1491)
        evaluator.mod_switch_to_next_inplace(m);
        EXPECT(m.coeff_modulus_size() == 1);
        EXPECT(throws<std::invalid_argument>([&] { evaluator.rescale_to_next_inplace(m); }));
        EXPECT(throws<std::invalid_argument>([&] { evaluator.mod_switch_to_next_inplace(m); }));
    }
    so_context_destroy(oc);
}

static void test_bfv()
{
    const int log_n = 12;
    const size_t n = 4096;
    const int bits[3] = { 36, 36, 37 };
    const uint64_t t = 65537;
    vec primes(3);
    EXPECT(so_coeff_modulus_create(n, bits, 3, primes.data()) == 0);
    const size_t k = 3, L = 2;
    so_context *oc = so_context_create(log_n, primes.data(), k, SO_SCHEME_BFV, t);

    EncryptionParameters parms(scheme_type::bfv);
    parms.set_poly_modulus_degree(n);
    parms.set_coeff_modulus(primes);
    parms.set_plain_modulus(t);
    SEALContext context(parms);
    Evaluator evaluator(context);

    vec sk(k * n), rk((k - 1) * 2 * k * n);
    so_sample_ternary_ntt(oc, 1, sk.data());
    so_gen_relin_key(oc, sk.data(), 2, rk.data());
    RelinKeys relin_keys(context);
    relin_keys.set_key(0, rk.data());

    vec x = uniform(oc, L, 2, n, 20), y = uniform(oc, L, 2, n, 21);
    Ciphertext encrypted1(context, L), encrypted2(context, L);
    encrypted1.upload(x.data());
    encrypted2.upload(y.data());
    vec exp3(3 * L * n, 0);
    std::memcpy(exp3.data(), x.data(), x.size() * 8);
    so_bfv_multiply(oc, L, exp3.data(), y.data());
    evaluator.multiply_inplace(encrypted1, encrypted2);
    EXPECT(encrypted1.size() == 3 && encrypted1.to_cpu() == exp3);
    Ciphertext one_call(encrypted1);
    so_relinearize(oc, L, exp3.data(), rk.data());
    evaluator.relinearize_inplace(encrypted1, relin_keys);
    vec exp2(exp3.begin(), exp3.begin() + 2 * L * n);
    EXPECT(encrypted1.to_cpu() == exp2);
    // BFV mod switch = divide_and_round_q_last_inplace (util/rns.cpp:789-828)
    vec e(2 * (L - 1) * n);
    so_mod_switch_scale_to_next(oc, L, exp2.data(), 2, e.data());
    evaluator.mod_switch_to_next_inplace(encrypted1);
    EXPECT(encrypted1.to_cpu() == e);
    evaluator.relinearize_rescale_inplace(one_call, relin_keys); // the same two steps as one call
    EXPECT(one_call.coeff_modulus_size() == L - 1 && !one_call.is_ntt_form() && one_call.to_cpu() == e);
    EXPECT(throws<std::invalid_argument>([&] { evaluator.rescale_to_next_inplace(encrypted2); }));

    // multiply_many / exponentiate (evaluator.cpp:1627-1736): the product tree, relinearized after every product
    {
        Ciphertext a(context, L), b(context, L), c(context, L);
        a.upload(x.data());
        b.upload(y.data());
        vec z = uniform(oc, L, 2, n, 22);
        c.upload(z.data());
        auto mul_relin = [&](const vec &u, const vec &v) {
            vec p3(3 * L * n, 0);
            std::memcpy(p3.data(), u.data(), u.size() * 8);
            so_bfv_multiply(oc, L, p3.data(), v.data());
            so_relinearize(oc, L, p3.data(), rk.data());
            return vec(p3.begin(), p3.begin() + 2 * L * n);
        };
        Ciphertext prod3, power;
        evaluator.multiply_many({ a, b, c }, relin_keys, prod3); // (a b) then ((a b) c)... the tree: [ab, c] -> (ab) c
        EXPECT(prod3.size() == 2 && prod3.to_cpu() == mul_relin(mul_relin(x, y), z));
        evaluator.exponentiate(a, 4, relin_keys, power);          // [a a, a a] -> (a a)(a a)
        const vec a2 = mul_relin(x, x);
        EXPECT(power.to_cpu() == mul_relin(a2, a2));
        evaluator.exponentiate_inplace(b, 1, relin_keys);
        EXPECT(b.to_cpu() == y);
        EXPECT(throws<std::invalid_argument>([&] { evaluator.exponentiate_inplace(b, 0, relin_keys); }));
        EXPECT(throws<std::invalid_argument>([&] { evaluator.multiply_many({ a, b }, relin_keys, prod3); evaluator.multiply_many({}, relin_keys, prod3); }));
    }
    so_context_destroy(oc);
}

// BGV through the facade: multiply, relinearize, mod_switch_to_next with the correction-factor
// bookkeeping of evaluator.cpp:838-840 and :1274-1279, rotate_rows
static void test_bgv()
{
    const int log_n = 12;
    const size_t n = 4096;
    const int bits[4] = { 50, 40, 40, 50 };
    const uint64_t t = 65537;
    vec primes(4);
    EXPECT(so_coeff_modulus_create(n, bits, 4, primes.data()) == 0);
    const size_t k = 4, L = 3;
    so_context *oc = so_context_create(log_n, primes.data(), k, SO_SCHEME_BGV, t);

    EncryptionParameters parms(scheme_type::bgv);
    parms.set_poly_modulus_degree(n);
    parms.set_coeff_modulus(primes);
    parms.set_plain_modulus(t);
    SEALContext context(parms);
    Evaluator evaluator(context);

    vec sk(k * n), rk((k - 1) * 2 * k * n), gk(rk.size());
    so_sample_ternary_ntt(oc, 1, sk.data());
    so_gen_relin_key(oc, sk.data(), 2, rk.data());
    const uint32_t elt = so_galois_elt_from_step(log_n, 1);
    so_gen_galois_key(oc, sk.data(), elt, 3, gk.data());
    RelinKeys relin_keys(context);
    relin_keys.set_key(0, rk.data());
    GaloisKeys galois_keys(context);
    galois_keys.set_key(GaloisKeys::get_index(elt), gk.data());

    vec x = uniform(oc, L, 2, n, 30), y = uniform(oc, L, 2, n, 31);
    Ciphertext encrypted1(context, L), encrypted2(context, L);
    encrypted1.upload(x.data());
    encrypted2.upload(y.data());
    EXPECT(throws<std::invalid_argument>([&] { evaluator.multiply_inplace(encrypted1, encrypted2); })); // :712
    encrypted1.is_ntt_form() = encrypted2.is_ntt_form() = true;
    encrypted1.correction_factor() = 3;
    encrypted2.correction_factor() = 5;

    vec exp3(3 * L * n, 0);
    std::memcpy(exp3.data(), x.data(), x.size() * 8);
    so_ckks_multiply(oc, L, exp3.data(), y.data()); // the same tensor product (evaluator.cpp:770-798)
    evaluator.multiply_inplace(encrypted1, encrypted2);
    EXPECT(encrypted1.size() == 3 && encrypted1.to_cpu() == exp3 && encrypted1.correction_factor() == 15);
    Ciphertext one_call(encrypted1);
    so_relinearize(oc, L, exp3.data(), rk.data());
    evaluator.relinearize_inplace(encrypted1, relin_keys);
    vec exp2(exp3.begin(), exp3.begin() + 2 * L * n);
    EXPECT(encrypted1.to_cpu() == exp2);

    vec e(2 * (L - 1) * n);
    so_mod_switch_scale_to_next(oc, L, exp2.data(), 2, e.data());
    evaluator.mod_switch_to_next_inplace(encrypted1);
    uint64_t inv = 0;
    EXPECT(so_try_invert_uint_mod(primes[L - 1] % t, t, &inv) == 1);
    EXPECT(encrypted1.coeff_modulus_size() == L - 1 && encrypted1.is_ntt_form() && encrypted1.to_cpu() == e);
    EXPECT(encrypted1.correction_factor() == 15 * inv % t);
    EXPECT(throws<std::invalid_argument>([&] { evaluator.rescale_to_next_inplace(encrypted2); }));
    evaluator.relinearize_rescale_inplace(one_call, relin_keys); // the same two steps as one call
    EXPECT(one_call.coeff_modulus_size() == L - 1 && one_call.is_ntt_form() && one_call.to_cpu() == e);
    EXPECT(one_call.correction_factor() == encrypted1.correction_factor() && one_call.scale() == encrypted1.scale());

    // add / sub of operands with different correction factors and sizes (evaluator.cpp:196-237, 302-350)
    for (int sub = 0; sub < 2; ++sub)
    {
        vec a3 = uniform(oc, L, 2, n, 32), b3 = uniform(oc, L, 3, n, 33);
        Ciphertext a(context, L, 2), b(context, L, 3);
        a.upload(a3.data());
        b.upload(b3.data());
        a.is_ntt_form() = b.is_ntt_form() = true;
        a.correction_factor() = 7;
        b.correction_factor() = 12345;
        uint64_t f[3];
        EXPECT(so_balance_correction_factors(7, 12345, t, f) == 0);
        EXPECT((unsigned __int128)f[1] * 7 % t == (unsigned __int128)f[2] * 12345 % t && f[0] == f[1] * 7 % t);
        vec e(3 * L * n);
        for (size_t p = 0; p < 3; ++p)
            for (size_t j = 0; j < L; ++j)
            {
                const so_modulus *m = &so_context_tables(oc, j)->mod;
                vec sb(n), sa(n, 0);
                so_multiply_poly_scalar_coeffmod(&b3[(p * L + j) * n], n, f[2], m, sb.data());
                if (p < 2)
                    so_multiply_poly_scalar_coeffmod(&a3[(p * L + j) * n], n, f[1], m, sa.data());
                if (sub)
                    so_sub_poly_coeffmod(sa.data(), sb.data(), n, m, &e[(p * L + j) * n]);
                else
                    so_add_poly_coeffmod(sa.data(), sb.data(), n, m, &e[(p * L + j) * n]);
            }
        if (sub)
            evaluator.sub_inplace(a, b);
        else
            evaluator.add_inplace(a, b);
        EXPECT(a.size() == 3 && a.correction_factor() == f[0] && a.to_cpu() == e);
    }

    vec r = y;
    so_apply_galois_inplace(oc, L, r.data(), elt, gk.data());
    evaluator.rotate_rows_inplace(encrypted2, 1, galois_keys);
    EXPECT(encrypted2.to_cpu() == r);
    EXPECT(throws<std::logic_error>([&] { evaluator.rotate_vector_inplace(encrypted2, 1, galois_keys); }));
    so_context_destroy(oc);
}

// KeyGenerator / Encryptor / Decryptor of the facade with a fixed-seed Blake2xbPRNGFactory against
// the oracle's restatement of keygenerator.cpp, util/rlwe.cpp, encryptor.cpp and decryptor.cu with the same seed:
// identical key and ciphertext words, then the Evaluator pipeline on them, then decryption.
static void test_crypto(scheme_type scheme)
{
    const int log_n = 12;
    const size_t n = 4096;
    const int bits[4] = { 50, 40, 40, 50 };
    const uint64_t t = scheme == scheme_type::ckks ? 0 : 65537;
    vec primes(4);
    EXPECT(so_coeff_modulus_create(n, bits, 4, primes.data()) == 0);
    const size_t k = 4;
    const int so_scheme = scheme == scheme_type::ckks ? SO_SCHEME_CKKS : scheme == scheme_type::bfv ? SO_SCHEME_BFV : SO_SCHEME_BGV;
    so_context *oc = so_context_create(log_n, primes.data(), k, so_scheme, t);
    const bool ntt = scheme != scheme_type::bfv;
    const prng_seed_type seed = { 101, 102, 103, 104, 105, 106, 107, 108 };

    EncryptionParameters parms(scheme);
    parms.set_poly_modulus_degree(n);
    parms.set_coeff_modulus(primes);
    parms.set_plain_modulus(t);
    parms.set_random_generator(std::make_shared<Blake2xbPRNGFactory>(seed));
    SEALContext context(parms);

    KeyGenerator keygen(context);
    vec sk(k * n);
    so_seal_gen_secret_key(oc, seed.data(), sk.data());
    EXPECT(keygen.secret_key().to_cpu() == sk);

    PublicKey public_key;
    keygen.create_public_key(public_key);
    vec pk(2 * k * n);
    so_seal_encrypt_zero_symmetric(oc, k, sk.data(), seed.data(), 1, pk.data());
    EXPECT(public_key.data().coeff_modulus_size() == k && public_key.data().is_ntt_form() && public_key.data().to_cpu() == pk);

    RelinKeys relin_keys(context);
    keygen.create_relin_keys(relin_keys);
    vec rk((k - 1) * 2 * k * n);
    so_seal_gen_relin_key(oc, sk.data(), seed.data(), rk.data());
    EXPECT(relin_keys.has_key(2) && relin_keys.key_to_cpu(0) == rk);

    GaloisKeys galois_keys(context);
    keygen.create_galois_keys(std::vector<int>{ 1 }, galois_keys);
    const uint32_t elt = so_galois_elt_from_step(log_n, 1);
    vec gk(rk.size());
    so_seal_gen_galois_key(oc, sk.data(), elt, seed.data(), gk.data());
    EXPECT(galois_keys.has_key(elt) && galois_keys.key_to_cpu(GaloisKeys::get_index(elt)) == gk);

    // a KeyGenerator built from an existing secret key makes the same keys (keygenerator.cpp:27-54)
    {
        KeyGenerator again(context, SecretKey(context, sk.data()));
        PublicKey pk2;
        again.create_public_key(pk2);
        EXPECT(pk2.data().to_cpu() == pk);
    }

    Encryptor encryptor(context, public_key, keygen.secret_key());
    Decryptor decryptor(context, keygen.secret_key());
    for (size_t L = k; L >= 2; --L)
    {
        // public-key encryptions below the key level are made one level up and switched down (encryptor.cpp:124-166)
        vec e(2 * L * n), ph(L * n);
        Ciphertext ct;
        so_seal_encrypt_zero(oc, L, pk.data(), nullptr, seed.data(), e.data());
        encryptor.encrypt_zero(L, ct);
        EXPECT(ct.coeff_modulus_size() == L && ct.is_ntt_form() == ntt && ct.to_cpu() == e);
        so_seal_encrypt_zero(oc, L, nullptr, sk.data(), seed.data(), e.data());
        encryptor.encrypt_zero_symmetric(L, ct);
        EXPECT(ct.to_cpu() == e);
        // the phase c0 + c1 s (decryptor.cu:560-591)
        so_seal_dot_product_ct_sk(oc, L, sk.data(), e.data(), 2, ntt, ph.data());
        seal::detail::DeviceWords d_ph(context.handle(), L * n);
        decryptor.dot_product_ct_sk_array(ct, d_ph.get());
        EXPECT(d_ph.to_cpu() == ph);
    }
    {
        Encryptor no_keys(context, keygen.secret_key());
        Ciphertext ct;
        EXPECT(throws<std::logic_error>([&] { no_keys.encrypt_zero(ct); })); // "public key is not set"
    }

    if (scheme != scheme_type::ckks)
    {
        // size-3 phase in coefficient (BFV) / NTT (BGV) form: c0 + c1 s + c2 s^2 (decryptor.cu:592-625)
        const size_t L = 3;
        vec x = uniform(oc, L, 3, n, 40), ph(L * n);
        Ciphertext ct(context, L, 3);
        ct.upload(x.data());
        ct.is_ntt_form() = ntt;
        so_seal_dot_product_ct_sk(oc, L, sk.data(), x.data(), 3, ntt, ph.data());
        seal::detail::DeviceWords d_ph(context.handle(), L * n);
        decryptor.dot_product_ct_sk_array(ct, d_ph.get());
        EXPECT(d_ph.to_cpu() == ph);
        Plaintext plain;
        decryptor.decrypt(ct, plain); // random words: any plaintext below t, never more than N coefficients
        EXPECT(plain.coeff_count() >= 1 && plain.coeff_count() <= n);
        for (uint64_t c : plain.data())
            EXPECT(c < t);
        so_context_destroy(oc);
        return;
    }

    // CKKS end to end: encrypt two plaintexts, multiply, relinearize, rescale, rotate, decrypt
    const size_t L = 3;
    const double delta = std::pow(2.0, 30);
    auto make_plain = [&](uint64_t mseed, vec &host) {
        host.assign(L * n, 0);
        for (size_t i = 0; i < n; ++i)
        {
            const uint64_t m = (mseed * 2654435761ULL + i * 40503ULL) % 97; // small message coefficients
            for (size_t j = 0; j < L; ++j)
                host[j * n + i] = (m << 30) % primes[j];
        }
        for (size_t j = 0; j < L; ++j)
            so_ntt_forward(&host[j * n], so_context_tables(oc, j), 0);
    };
    vec p1, p2;
    make_plain(1, p1);
    make_plain(2, p2);
    Plaintext plain1(context, L), plain2(context, L);
    plain1.upload(p1.data());
    plain2.upload(p2.data());
    plain1.scale() = plain2.scale() = delta;
    Ciphertext c1, c2;
    encryptor.encrypt(plain1, c1);
    encryptor.encrypt_symmetric(plain2, c2);
    vec e1(2 * L * n), e2(2 * L * n);
    so_seal_encrypt_zero(oc, L, pk.data(), nullptr, seed.data(), e1.data());
    so_seal_encrypt_zero(oc, L, nullptr, sk.data(), seed.data(), e2.data());
    for (size_t j = 0; j < L; ++j)
    {
        so_add_poly_coeffmod(&e1[j * n], &p1[j * n], n, &so_context_tables(oc, j)->mod, &e1[j * n]);
        so_add_poly_coeffmod(&e2[j * n], &p2[j * n], n, &so_context_tables(oc, j)->mod, &e2[j * n]);
    }
    EXPECT(c1.to_cpu() == e1 && c2.to_cpu() == e2 && c1.scale() == delta);

    // decrypt(encrypt(m)) is m up to the fresh noise, and equals the oracle's phase word for word
    {
        Plaintext back(context, L);
        decryptor.decrypt(c1, back);
        vec ph(L * n);
        so_seal_dot_product_ct_sk(oc, L, sk.data(), e1.data(), 2, 1, ph.data());
        EXPECT(back.to_cpu() == ph && back.scale() == delta);
        so_ntt_inverse(ph.data(), so_context_tables(oc, 0), 0);
        bool close = true;
        for (size_t i = 0; i < n; ++i)
        {
            const uint64_t m = (1 * 2654435761ULL + i * 40503ULL) % 97;
            const int64_t v = ph[i] > primes[0] / 2 ? (int64_t)ph[i] - (int64_t)primes[0] : (int64_t)ph[i];
            close = close && std::llabs(v - (int64_t)(m << 30)) < (1 << 20);
        }
        EXPECT(close);
    }

    Evaluator evaluator(context);
    vec exp3(3 * L * n, 0);
    std::memcpy(exp3.data(), e1.data(), e1.size() * 8);
    so_ckks_multiply(oc, L, exp3.data(), e2.data());
    so_relinearize(oc, L, exp3.data(), rk.data());
    vec exp2(exp3.begin(), exp3.begin() + 2 * L * n), expr(2 * (L - 1) * n);
    so_mod_switch_scale_to_next(oc, L, exp2.data(), 2, expr.data());
    so_apply_galois_inplace(oc, L - 1, expr.data(), elt, gk.data());
    evaluator.multiply_inplace(c1, c2);
    evaluator.relinearize_inplace(c1, relin_keys);
    evaluator.rescale_to_next_inplace(c1);
    evaluator.rotate_vector_inplace(c1, 1, galois_keys);
    EXPECT(c1.to_cpu() == expr);
    Plaintext result(context, L - 1);
    decryptor.decrypt(c1, result);
    vec ph((L - 1) * n);
    so_seal_dot_product_ct_sk(oc, L - 1, sk.data(), expr.data(), 2, 1, ph.data());
    EXPECT(result.to_cpu() == ph && result.scale() == delta * delta / (double)primes[L - 1]);

    // hoisted rotations (not in the reference; cat_rotate_hoisted): the same Galois element twice in one call, each
    // output decrypting to what the un-hoisted rotation decrypts to, up to key-switching noise, and the input untouched
    {
        Ciphertext plain_rot;
        evaluator.apply_galois(c2, elt, galois_keys, plain_rot);
        const vec before = c2.to_cpu();
        std::vector<Ciphertext> hoisted;
        evaluator.apply_galois_hoisted(c2, { elt, elt }, galois_keys, hoisted);
        EXPECT(hoisted.size() == 2 && c2.to_cpu() == before);
        EXPECT(hoisted[0].to_cpu() == hoisted[1].to_cpu() && hoisted[0].scale() == c2.scale() && hoisted[0].is_ntt_form());
        Plaintext pa(context, L), pb(context, L);
        decryptor.decrypt(plain_rot, pa);
        decryptor.decrypt(hoisted[0], pb);
        vec wa = pa.to_cpu(), wb = pb.to_cpu();
        so_ntt_inverse(wa.data(), so_context_tables(oc, 0), 0);
        so_ntt_inverse(wb.data(), so_context_tables(oc, 0), 0);
        bool close = true;
        for (size_t i = 0; i < n; ++i)
        {
            const uint64_t dlt = wa[i] >= wb[i] ? wa[i] - wb[i] : wb[i] - wa[i];
            close = close && (dlt < (1u << 20) || primes[0] - dlt < (1u << 20));
        }
        EXPECT(close);
        EXPECT(throws<std::invalid_argument>([&] { evaluator.apply_galois_hoisted(c2, { elt + 2 }, galois_keys, hoisted); })); // "Galois key not present"
        std::vector<Ciphertext> by_step;
        evaluator.rotate_hoisted(c2, { 1 }, galois_keys, by_step);
        EXPECT(by_step.size() == 1 && by_step[0].to_cpu() == hoisted[0].to_cpu());
    }
    so_context_destroy(oc);
}

// the reference's wire format.  Objects made with a fixed seed are written to `dir` (the pytest
// driver compares the files byte for byte with the oracle's restatement, oracle/seal_serialization.py), the files
// the oracle wrote there beforehand are loaded and compared with the same objects, every object survives a round
// trip in memory, and malformed input fails the way serialization.cpp:340-452 / ciphertext.cpp:288-400 fail.
static void test_serialization(const std::string &dir)
{
    const int log_n = 12;
    const size_t n = 4096, k = 4, L = 3;
    const int bits[4] = { 50, 40, 40, 50 };
    vec primes(4);
    EXPECT(so_coeff_modulus_create(n, bits, 4, primes.data()) == 0);
    so_context *oc = so_context_create(log_n, primes.data(), k, SO_SCHEME_CKKS, 0);
    const prng_seed_type seed = { 201, 202, 203, 204, 205, 206, 207, 208 };
    EncryptionParameters parms(scheme_type::ckks);
    parms.set_poly_modulus_degree(n);
    parms.set_coeff_modulus(primes);
    parms.set_random_generator(std::make_shared<Blake2xbPRNGFactory>(seed));
    SEALContext context(parms);
    EXPECT(context.level_of(context.parms_id(2)) == 2 && context.level_of(parms_id_zero) == 0 &&
           context.key_parms_id() != context.first_parms_id());

    KeyGenerator keygen(context);
    PublicKey public_key;
    keygen.create_public_key(public_key);
    RelinKeys relin_keys(context);
    keygen.create_relin_keys(relin_keys);
    GaloisKeys galois_keys(context);
    keygen.create_galois_keys(std::vector<int>{ 1 }, galois_keys);
    const uint32_t elt = so_galois_elt_from_step(log_n, 1);
    Encryptor encryptor(context, keygen.secret_key());
    Ciphertext ct, seeded;
    encryptor.encrypt_zero_symmetric(L, ct);
    ct.scale() = std::pow(2.0, 40);
    util::encrypt_zero_symmetric(keygen.secret_key(), context, L, true, true, seeded);
    vec seeded_words = seeded.to_cpu(), ct_words = ct.to_cpu();
    {
        vec e(2 * L * n);
        so_seal_encrypt_zero_symmetric_seeded(oc, L, keygen.secret_key().to_cpu().data(), seed.data(), 1, e.data());
        EXPECT(seeded_words == e && seeded_words[L * n] == 0xFFFFFFFFFFFFFFFFULL);
        EXPECT(vec(seeded_words.begin(), seeded_words.begin() + L * n) == vec(ct_words.begin(), ct_words.begin() + L * n));
    }
    vec pl = uniform(oc, L, 1, n, 55);
    Plaintext plain(context, L);
    plain.upload(pl.data());
    plain.scale() = std::pow(2.0, 40);

    auto write = [&](const char *name, const std::function<void(std::ostream &)> &save) {
        std::ofstream f(dir + "/" + name, std::ios::binary);
        save(f);
        EXPECT(f.good());
    };
    write("facade_sk.bin", [&](std::ostream &s) { keygen.secret_key().save(context, s); });
    write("facade_pk.bin", [&](std::ostream &s) { public_key.save(s); });
    write("facade_rk.bin", [&](std::ostream &s) { relin_keys.save(s); });
    write("facade_gk.bin", [&](std::ostream &s) { galois_keys.save(s); });
    write("facade_ct.bin", [&](std::ostream &s) { EXPECT(ct.save(s) == ct.save_size()); });
    write("facade_seeded.bin", [&](std::ostream &s) { EXPECT(seeded.save(s) == seeded.save_size(true)); });
    write("facade_plain.bin", [&](std::ostream &s) { plain.save(s); });

    // round trips in memory
    {
        std::stringstream s(std::ios::in | std::ios::out | std::ios::binary);
        ct.save(s);
        seeded.save(s);
        plain.save(s);
        keygen.secret_key().save(context, s);
        public_key.save(s);
        relin_keys.save(s);
        galois_keys.save(s);
        Ciphertext a, b;
        EXPECT(a.load(context, s) == ct.save_size());
        EXPECT(a.to_cpu() == ct_words && a.is_ntt_form() && a.scale() == ct.scale() && a.coeff_modulus_size() == L);
        b.load(context, s);
        EXPECT(b.to_cpu() == ct_words); // the expanded seed gives back the second polynomial
        Plaintext p2(context, 1);
        p2.load(context, s);
        EXPECT(p2.to_cpu() == pl && p2.scale() == plain.scale() && p2.coeff_modulus_size() == L);
        SecretKey sk2;
        sk2.load(context, s);
        EXPECT(sk2.to_cpu() == keygen.secret_key().to_cpu());
        PublicKey pk2;
        pk2.load(context, s);
        EXPECT(pk2.data().to_cpu() == public_key.data().to_cpu());
        RelinKeys rk2(context);
        rk2.load(context, s);
        EXPECT(rk2.has_key(2) && rk2.size() == 1 && rk2.key_to_cpu(0) == relin_keys.key_to_cpu(0));
        GaloisKeys gk2(context);
        gk2.load(context, s);
        EXPECT(gk2.has_key(elt) && gk2.size() == 1 &&
               gk2.key_to_cpu(GaloisKeys::get_index(elt)) == galois_keys.key_to_cpu(GaloisKeys::get_index(elt)));
        // loaded keys work: relinearize with rk2 equals relinearize with the original
        vec x = uniform(oc, L, 3, n, 56);
        Ciphertext c3(context, L, 3), c3b(context, L, 3);
        c3.upload(x.data());
        c3b.upload(x.data());
        c3.is_ntt_form() = c3b.is_ntt_form() = true;
        Evaluator evaluator(context);
        evaluator.relinearize_inplace(c3, relin_keys);
        evaluator.relinearize_inplace(c3b, rk2);
        EXPECT(c3.to_cpu() == c3b.to_cpu());
    }

    // files written by the oracle with the same parameters and seed
    auto open = [&](const char *name) { return std::ifstream(dir + "/" + name, std::ios::binary); };
    if (open("oracle_ct.bin").good())
    {
        std::ifstream f = open("oracle_ct.bin");
        Ciphertext a;
        a.load(context, f);
        EXPECT(a.to_cpu() == ct_words && a.scale() == ct.scale());
        f = open("oracle_seeded.bin");
        a.load(context, f);
        EXPECT(a.to_cpu() == ct_words);
        f = open("oracle_sk.bin");
        SecretKey sk2;
        sk2.load(context, f);
        EXPECT(sk2.to_cpu() == keygen.secret_key().to_cpu());
        f = open("oracle_pk.bin");
        PublicKey pk2;
        pk2.load(context, f);
        EXPECT(pk2.data().to_cpu() == public_key.data().to_cpu());
        f = open("oracle_rk.bin");
        RelinKeys rk2(context);
        rk2.load(context, f);
        EXPECT(rk2.key_to_cpu(0) == relin_keys.key_to_cpu(0));
        f = open("oracle_gk.bin");
        GaloisKeys gk2(context);
        gk2.load(context, f);
        EXPECT(gk2.key_to_cpu(GaloisKeys::get_index(elt)) == galois_keys.key_to_cpu(GaloisKeys::get_index(elt)));
        f = open("oracle_plain.bin");
        Plaintext p2(context, 1);
        p2.load(context, f);
        EXPECT(p2.to_cpu() == pl);
        std::printf("serialization: oracle files loaded\n");
    }

    // malformed input
    {
        std::ostringstream os(std::ios::binary);
        ct.save(os);
        const std::string good = os.str();
        auto load_from = [&](std::string bytes) {
            std::istringstream is(bytes, std::ios::binary);
            Ciphertext a;
            a.load(context, is);
        };
        std::string bad = good;
        bad[0] = 0;
        EXPECT(throws<std::logic_error>([&] { load_from(bad); })); // "loaded SEALHeader is invalid"
        bad = good;
        bad[5] = 1;
        EXPECT(throws<std::invalid_argument>([&] { load_from(bad); })); // "unsupported compression mode"
        bad = good;
        bad[3] = 5;
        EXPECT(throws<std::logic_error>([&] { load_from(bad); })); // "incompatible version"
        bad = good;
        bad[16] ^= 1;
        EXPECT(throws<std::logic_error>([&] { load_from(bad); })); // unknown parms_id: "ciphertext data is invalid"
        bad = good;
        bad[16 + 32 + 1 + 16] = 2; // limb count of another level
        EXPECT(throws<std::logic_error>([&] { load_from(bad); }));
        bad = good;
        bad[8] ^= 1; // header size field no longer matches what was read
        EXPECT(throws<std::logic_error>([&] { load_from(bad); })); // "invalid data size"
        EXPECT(throws<std::runtime_error>([&] { load_from(good.substr(0, good.size() - 9)); })); // "I/O error"
        // a context over the same three primes has this level as its key level and accepts the object (pure key levels
        // are allowed, ciphertext.cpp:331-342); with the primes in another order the parms_id is unknown
        EncryptionParameters same(scheme_type::ckks), other(scheme_type::ckks);
        same.set_poly_modulus_degree(n);
        same.set_coeff_modulus(vec(primes.begin(), primes.begin() + 3));
        SEALContext context3(same);
        {
            std::istringstream is(good, std::ios::binary);
            Ciphertext a;
            a.load(context3, is);
            EXPECT(a.to_cpu() == ct_words);
        }
        other.set_poly_modulus_degree(n);
        other.set_coeff_modulus({ primes[1], primes[0], primes[2] });
        SEALContext context2(other);
        EXPECT(throws<std::logic_error>([&] {
            std::istringstream is(good, std::ios::binary);
            Ciphertext a;
            a.load(context2, is);
        }));
    }
    so_context_destroy(oc);
}

// BFV / BGV with plaintexts: encrypt -> multiply -> relinearize -> mod switch -> decrypt gives the
// negacyclic product modulo t, as the reference's Evaluator tests check (native/tests/seal/evaluator.cpp:5259-5558);
// the ciphertexts and decrypted plaintexts are also written to `dir` for the bit-exact comparison with the oracle
// (oracle/seal_plain.py on top of oracle/seal_oracle.c) in tests/test_gpu_facade.py.
static void test_plain_modes(const std::string &dir, scheme_type scheme)
{
    const size_t n = 4096, k = 4, L = 3;
    const int bits[4] = { 40, 40, 40, 41 };
    const uint64_t t = 65537;
    vec primes(4);
    EXPECT(so_coeff_modulus_create(n, bits, 4, primes.data()) == 0);
    const prng_seed_type seed = { 301, 302, 303, 304, 305, 306, 307, 308 };
    EncryptionParameters parms(scheme);
    parms.set_poly_modulus_degree(n);
    parms.set_coeff_modulus(primes);
    parms.set_plain_modulus(t);
    parms.set_random_generator(std::make_shared<Blake2xbPRNGFactory>(seed));
    SEALContext context(parms);
    KeyGenerator keygen(context);
    RelinKeys relin_keys(context);
    keygen.create_relin_keys(relin_keys);
    PublicKey public_key;
    keygen.create_public_key(public_key);
    Encryptor encryptor(context, public_key, keygen.secret_key());
    Decryptor decryptor(context, keygen.secret_key());
    Evaluator evaluator(context);
    EXPECT(context.first_level_size() == L && k == context.key_modulus_size());

    vec m1(n), m2(n / 2); // the second plaintext is shorter than N (plain.coeff_count() < N)
    for (size_t i = 0; i < n; ++i)
        m1[i] = (i * i * 7 + 3 * i + 11) % t;
    for (size_t i = 0; i < m2.size(); ++i)
        m2[i] = (i * 5 + (i % 7) * 1000 + 65000) % t;
    Ciphertext c1, c2;
    encryptor.encrypt_symmetric(Plaintext(m1), c1);
    encryptor.encrypt(Plaintext(m2), c2);
    EXPECT(c1.coeff_modulus_size() == L && c1.is_ntt_form() == (scheme == scheme_type::bgv));
    Plaintext back;
    decryptor.decrypt(c1, back);
    EXPECT(back.data() == m1 && !back.is_ntt_form());
    decryptor.decrypt(c2, back);
    EXPECT(back.data() == m2);
    EXPECT(throws<std::invalid_argument>([&] { encryptor.encrypt(Plaintext(vec{ t }), c2); })); // coefficient not below t

    // expected product: negacyclic convolution modulo t
    vec exp(n, 0);
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < m2.size(); ++j)
        {
            const uint64_t p = m1[i] * m2[j] % t;
            const size_t d = i + j;
            if (d < n)
                exp[d] = (exp[d] + p) % t;
            else
                exp[d - n] = (exp[d - n] + t - p) % t;
        }
    Ciphertext prod(c1);
    evaluator.multiply_inplace(prod, c2);
    evaluator.relinearize_inplace(prod, relin_keys);
    decryptor.decrypt(prod, back);
    vec got = back.data();
    got.resize(n, 0);
    EXPECT(got == exp);
    evaluator.mod_switch_to_next_inplace(prod);
    EXPECT(prod.coeff_modulus_size() == L - 1 && (scheme == scheme_type::bfv || prod.correction_factor() != 1));
    decryptor.decrypt(prod, back);
    got = back.data();
    got.resize(n, 0);
    EXPECT(got == exp);

    // add_plain / sub_plain with a coefficient-form plaintext (evaluator.cpp:1737-1951), fresh and after the mod switch
    // (BGV: the correction factor of the switched ciphertext scales the plaintext)
    {
        vec sum(n), diff(n);
        for (size_t i = 0; i < n; ++i)
        {
            const uint64_t b = i < m2.size() ? m2[i] : 0;
            sum[i] = (m1[i] + b) % t;
            diff[i] = (m1[i] + t - b) % t;
        }
        auto dec = [&](const Ciphertext &ct) {
            Plaintext p;
            decryptor.decrypt(ct, p);
            vec v = p.data();
            v.resize(n, 0);
            return v;
        };
        Ciphertext a(c1), s2;
        evaluator.add_plain_inplace(a, Plaintext(m2));
        EXPECT(dec(a) == sum);
        evaluator.sub_plain(c1, Plaintext(m2), s2);
        EXPECT(dec(s2) == diff);
        evaluator.sub_plain_inplace(a, Plaintext(m2));
        EXPECT(dec(a) == m1);
        Ciphertext low(c1);
        evaluator.mod_switch_to_next_inplace(low);
        evaluator.add_plain_inplace(low, Plaintext(m2));
        EXPECT(low.coeff_modulus_size() == L - 1 && dec(low) == sum);
        EXPECT(throws<std::invalid_argument>([&] { evaluator.add_plain_inplace(a, Plaintext(vec{ t })); }));
        // multiply_plain with a coefficient-form plaintext (evaluator.cpp:1953-2133): the negacyclic product modulo t
        Ciphertext mp;
        evaluator.multiply_plain(c1, Plaintext(m2), mp);
        EXPECT(mp.size() == 2 && mp.is_ntt_form() == c1.is_ntt_form() && dec(mp) == exp);
        Ciphertext mono(c1);
        evaluator.multiply_plain_inplace(mono, Plaintext(vec{ 0, 0, 0, 5 })); // 5 x^3: the reference's monomial case
        vec em(n);
        for (size_t i = 0; i < n; ++i)
            em[(i + 3) % n] = i + 3 < n ? m1[i] * 5 % t : (t - m1[i] * 5 % t) % t;
        EXPECT(dec(mono) == em);
        // the same product with the plaintext transformed first (evaluator.cpp:2174-2265, then :1967-1983)
        Plaintext m2_ntt;
        evaluator.transform_to_ntt(Plaintext(m2), c1.parms_id(), m2_ntt);
        EXPECT(m2_ntt.is_ntt_form() && m2_ntt.coeff_modulus_size() == L && m2_ntt.parms_id() == c1.parms_id());
        Ciphertext mp2(c1);
        evaluator.multiply_plain_inplace(mp2, m2_ntt);
        EXPECT(mp2.is_ntt_form() == c1.is_ntt_form() && mp2.to_cpu() == mp.to_cpu() && dec(mp2) == exp);
        EXPECT(throws<std::invalid_argument>([&] { evaluator.transform_to_ntt_inplace(m2_ntt, c1.parms_id()); }));
        // the product with a plaintext added, at the product's correction factor
        Ciphertext pa(prod);
        evaluator.add_plain_inplace(pa, Plaintext(m2));
        vec e2(exp);
        for (size_t i = 0; i < m2.size(); ++i)
            e2[i] = (e2[i] + m2[i]) % t;
        EXPECT(dec(pa) == e2);
    }

    const std::string tag = scheme == scheme_type::bfv ? "bfv" : "bgv";
    auto write_ct = [&](const std::string &name, const Ciphertext &ct) {
        std::ofstream f(dir + "/facade_" + tag + "_" + name + ".bin", std::ios::binary);
        ct.save(f);
    };
    write_ct("c1", c1);
    write_ct("c2", c2);
    write_ct("prod", prod);
    std::ofstream f(dir + "/facade_" + tag + "_dec.bin", std::ios::binary);
    f.write(reinterpret_cast<const char *>(got.data()), (std::streamsize)(got.size() * 8));
}


// A caller written the way the reference's examples set a CKKS circuit up (native/examples/4_ckks_basics.cpp:60-110,
// minus the encoder, which is out of scope): Modulus / CoeffModulus::Create / set_coeff_modulus(vector<Modulus>),
// KeyGenerator, Encryptor, Evaluator, Ciphertext(context, parms_id).  The words are checked against the oracle.
static void test_reference_idiom()
{
    // Modulus: const_ratio = floor(2^128 / value) and the remainder (modulus.cpp:87-98), the limits of set_value
    {
        Modulus m(0xffffffffffc0001ULL);
        so_modulus om;
        so_modulus_init(&om, m.value());
        EXPECT(m.bit_count() == 60 && m.is_prime() && !m.is_zero() && m.uint64_count() == 1);
        EXPECT(m.const_ratio()[0] == om.ratio[0] && m.const_ratio()[1] == om.ratio[1]);
        const unsigned __int128 back = (((unsigned __int128)m.const_ratio()[1] << 64) | m.const_ratio()[0]) * m.value() + m.const_ratio()[2];
        EXPECT(back == 0); // = 2^128 modulo 2^128
        EXPECT(m.reduce(m.value() + 5) == 5);
        EXPECT(throws<std::invalid_argument>([] { Modulus bad(std::uint64_t(1) << 61); }));
        EXPECT(throws<std::invalid_argument>([] { Modulus bad(1); }));
        EXPECT(Modulus().is_zero() && throws<std::logic_error>([] { Modulus().reduce(3); }));
        EXPECT(!Modulus(15).is_prime());
    }
    // CoeffModulus (modulus.cpp:116-190, util/hestdparms.h, util/globals.cpp:24-70)
    {
        const int bits[4] = { 60, 40, 40, 60 };
        vec ref(4);
        EXPECT(so_coeff_modulus_create(8192, bits, 4, ref.data()) == 0);
        const std::vector<Modulus> q = CoeffModulus::Create(8192, { 60, 40, 40, 60 });
        EXPECT(q.size() == 4);
        for (size_t i = 0; i < q.size(); ++i)
            EXPECT(q[i].value() == ref[i]);
        EXPECT(q[0] < q[3] && q[1] < q[2]); // equal sizes: smallest first
        EXPECT(throws<std::invalid_argument>([] { CoeffModulus::Create(8192, { 61 }); }));
        EXPECT(throws<std::invalid_argument>([] { CoeffModulus::Create(1000, { 40 }); }));
        EXPECT(CoeffModulus::MaxBitCount(16384) == 438 && CoeffModulus::MaxBitCount(32768, sec_level_type::tc192) == 611 &&
               CoeffModulus::MaxBitCount(3000) == 0);
        for (size_t n : { size_t(4096), size_t(8192), size_t(16384), size_t(32768) })
        {
            vec d(32);
            size_t cnt = 0;
            EXPECT(so_bfv_default(n, d.data(), &cnt) == 0);
            const std::vector<Modulus> b = CoeffModulus::BFVDefault(n);
            EXPECT(b.size() == cnt);
            int total = 0;
            for (size_t i = 0; i < b.size() && i < cnt; ++i)
            {
                EXPECT(b[i].value() == d[i] && b[i].is_prime() && (b[i].value() - 1) % (2 * n) == 0);
                total += b[i].bit_count();
            }
            EXPECT(total == CoeffModulus::MaxBitCount(n));
        }
        EXPECT(throws<std::invalid_argument>([] { CoeffModulus::BFVDefault(3000); }));
        const Modulus t = PlainModulus::Batching(8192, 20);
        EXPECT(t.bit_count() == 20 && t.is_prime() && (t.value() - 1) % 16384 == 0);
        // BGV-style chain: every prime is also 1 modulo t (modulus.cpp:192-229)
        for (const Modulus &m : CoeffModulus::Create(4096, Modulus(65537), { 40, 40, 41 }))
            EXPECT(m.is_prime() && (m.value() - 1) % 8192 == 0 && (m.value() - 1) % 65537 == 0);
    }

    EncryptionParameters parms(scheme_type::ckks);
    const size_t poly_modulus_degree = 8192;
    parms.set_poly_modulus_degree(poly_modulus_degree);
    parms.set_coeff_modulus(CoeffModulus::Create(poly_modulus_degree, { 60, 40, 40, 60 }));
    const prng_seed_type seed = { 7, 7, 7, 7, 7, 7, 7, 7 };
    parms.set_random_generator(std::make_shared<Blake2xbPRNGFactory>(seed));
    EXPECT(parms.coeff_modulus().size() == 4 && parms.coeff_modulus()[0].bit_count() == 60 && parms.plain_modulus().is_zero());
    SEALContext context(parms);
    // the context pre-sized its workspace: nothing below allocates or synchronises inside an Evaluator call
    const size_t reserved = cat_ctx_workspace_bytes(context.handle());
    EXPECT(reserved >= cat_ctx_workspace_need(context.handle(), 1) && reserved > 0);

    KeyGenerator keygen(context);
    const SecretKey &secret_key = keygen.secret_key();
    PublicKey public_key;
    keygen.create_public_key(public_key);
    RelinKeys relin_keys(context);
    keygen.create_relin_keys(relin_keys);
    GaloisKeys gal_keys(context);
    keygen.create_galois_keys(std::vector<int>{ 1 }, gal_keys);
    Encryptor encryptor(context, public_key);
    Evaluator evaluator(context);
    Decryptor decryptor(context, secret_key);

    const double scale = std::pow(2.0, 40);
    Ciphertext x1_encrypted, x2_encrypted;
    encryptor.encrypt_zero(x1_encrypted);
    encryptor.encrypt_zero(x2_encrypted);
    x1_encrypted.scale() = scale;
    // scales one ulp apart are "the same scale" (util::are_close, evaluator.cpp:24-26, 174)
    x2_encrypted.scale() = std::nextafter(scale, 2 * scale);
    EXPECT(x1_encrypted.parms_id() == context.first_parms_id() && x1_encrypted.coeff_modulus_size() == 3);
    {
        Ciphertext sum;
        evaluator.add(x1_encrypted, x2_encrypted, sum);
        x2_encrypted.scale() = scale * 1.5;
        EXPECT(throws<std::invalid_argument>([&] { evaluator.add(x1_encrypted, x2_encrypted, sum); }));
        x2_encrypted.scale() = scale;
    }

    // Ciphertext(context, parms_id): an empty ciphertext of a named level (ciphertext.cuh:60-120)
    Ciphertext destination(context, context.first_parms_id());
    EXPECT(destination.size() == 2 && destination.coeff_modulus_size() == 3);
    EXPECT(throws<std::invalid_argument>([&] { Ciphertext bad(context, parms_id_type{ 1, 2, 3, 4 }); }));

    // x1 * x2, relinearize, rescale -- against the oracle on the same words and the same key
    const size_t n = poly_modulus_degree, k = 4, L = 3;
    vec primes;
    for (const Modulus &m : parms.coeff_modulus())
        primes.push_back(m.value());
    so_context *oc = so_context_create(13, primes.data(), k, SO_SCHEME_CKKS, 0);
    vec a = x1_encrypted.to_cpu(), b = x2_encrypted.to_cpu(), rk = relin_keys.key_to_cpu(0);
    evaluator.multiply(x1_encrypted, x2_encrypted, destination);
    evaluator.relinearize_inplace(destination, relin_keys);
    evaluator.rescale_to_next_inplace(destination);
    EXPECT(destination.size() == 2 && destination.coeff_modulus_size() == 2 && destination.parms_id() == context.parms_id(2));
    EXPECT(std::fabs(destination.scale() - scale * scale / (double)primes[2]) < 1e-6 * destination.scale());
    {
        vec prod(3 * L * n);
        std::copy(a.begin(), a.end(), prod.begin());
        so_ckks_multiply(oc, L, prod.data(), b.data());
        so_relinearize(oc, L, prod.data(), rk.data());
        vec out(2 * (L - 1) * n);
        so_mod_switch_scale_to_next(oc, L, prod.data(), 2, out.data());
        EXPECT(destination.to_cpu() == out);
    }
    // the product of two encryptions of zero decrypts to noise far below the scale: |phase| < 2^60 at a 100-bit modulus
    {
        seal::detail::DeviceWords d_ph(context.handle(), 2 * n);
        decryptor.dot_product_ct_sk_array(destination, d_ph.get());
        EXPECT(d_ph.to_cpu().size() == 2 * n);
    }
    evaluator.rotate_vector_inplace(destination, 1, gal_keys);
    EXPECT(cat_ctx_workspace_bytes(context.handle()) == reserved); // no call grew the workspace
    so_context_destroy(oc);
}

int main(int argc, char **argv)
{
    try
    {
        test_util_ntt_known_answers();
        test_reference_idiom();
        test_ckks();
        test_bfv();
        test_bgv();
        test_crypto(scheme_type::ckks);
        test_crypto(scheme_type::bfv);
        test_crypto(scheme_type::bgv);
        if (argc > 1)
        {
            test_serialization(argv[1]);
            test_plain_modes(argv[1], scheme_type::bfv);
            test_plain_modes(argv[1], scheme_type::bgv);
        }
    }
    catch (const std::exception &e)
    {
        std::printf("FAIL: unexpected exception: %s\n", e.what());
        return 2;
    }
    if (failures)
    {
        std::printf("%d facade check(s) failed\n", failures);
        return 1;
    }
    std::printf("facade ok: seal::Evaluator / seal::util over libcat_amd.so match the CPU oracle bit for bit\n");
    return 0;
}
