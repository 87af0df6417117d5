/*
 * seal_oracle.h -- CPU restatement of the reference's (Rayman96/CAT = SEAL 4.0 fork) RNS
 * polynomial arithmetic hot path.  TEST INFRASTRUCTURE ONLY: imported by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the checker / timed CPU
 * baseline.  Nothing under cat_amd/ (the product) may include, link or call this.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/native/src/seal/).  Pinning: tests/test_oracle_known_answers.py checks this
 * file against the literal known-answer vectors of the reference's own unit tests
 * (tests/golden/reference_known_answers.json, transcribed from the .cpp files under native/tests/seal/util/).
 * The reference itself is NOT buildable in this image under the build rules (it needs a
 * cmake-generated util/config.h and CUDA/cuRAND headers), see DESIGN.md.
 */
#ifndef SEAL_ORACLE_H
#define SEAL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* modulus.cpp:66-105 -- value, bit count, const_ratio = floor(2^128/value) lo/hi + remainder */
typedef struct {
    uint64_t value;
    uint64_t ratio[3];
    int bit_count;
} so_modulus;

/* util/uintarithsmallmod.cuh:363-393 */
typedef struct {
    uint64_t operand;
    uint64_t quotient;
} so_mulmod_op;

/* util/ntt.h:69-199 */
typedef struct {
    int log_n;
    size_t n;
    so_modulus mod;
    uint64_t root;
    uint64_t inv_root;
    so_mulmod_op *root_powers;     /* bit-reversed order, ntt.cpp:268-277 */
    so_mulmod_op *inv_root_powers; /* scrambled order,   ntt.cpp:279-287 */
    so_mulmod_op inv_degree;
} so_ntt_tables;

/* ---- word arithmetic (util/uintarithsmallmod.cuh, util/uintarith.cuh) ---- */
void so_modulus_init(so_modulus *m, uint64_t value);
uint64_t so_barrett_reduce_64(uint64_t x, const so_modulus *m);
uint64_t so_barrett_reduce_128(const uint64_t in[2], const so_modulus *m);
uint64_t so_multiply_uint_mod(uint64_t a, uint64_t b, const so_modulus *m);
void so_mulmod_op_set(so_mulmod_op *op, uint64_t operand, const so_modulus *m);
uint64_t so_multiply_uint_mod_op(uint64_t x, so_mulmod_op y, const so_modulus *m);
uint64_t so_multiply_uint_mod_lazy(uint64_t x, so_mulmod_op y, const so_modulus *m);
uint64_t so_add_uint_mod(uint64_t a, uint64_t b, const so_modulus *m);
uint64_t so_sub_uint_mod(uint64_t a, uint64_t b, const so_modulus *m);
uint64_t so_negate_uint_mod(uint64_t a, const so_modulus *m);
uint64_t so_exponentiate_uint_mod(uint64_t a, uint64_t e, const so_modulus *m);
int so_try_invert_uint_mod(uint64_t a, uint64_t modulus, uint64_t *result);
uint64_t so_dot_product_mod(const uint64_t *a, const uint64_t *b, size_t count, const so_modulus *m);
uint64_t so_modulo_uint(const uint64_t *value, size_t count, const so_modulus *m);

/* ---- number theory (util/numth.cpp) ---- */
int so_is_prime(uint64_t value);
int so_get_primes(uint64_t factor, int bit_size, size_t count, uint64_t *out);
int so_coeff_modulus_create(size_t n, const int *bit_sizes, size_t count, uint64_t *out);
int so_bfv_default(size_t n, uint64_t *out, size_t *count);
int so_try_minimal_primitive_root(uint64_t degree, const so_modulus *m, uint64_t *root);
uint32_t so_reverse_bits(uint32_t x, int bits);

/* ---- NTT (util/ntt.cpp, util/dwthandler.h) ---- */
so_ntt_tables *so_ntt_tables_create(int log_n, uint64_t modulus);
void so_ntt_tables_destroy(so_ntt_tables *t);
void so_ntt_forward(uint64_t *x, const so_ntt_tables *t, int lazy);
void so_ntt_inverse(uint64_t *x, const so_ntt_tables *t, int lazy);
uint64_t so_ntt_root_power(const so_ntt_tables *t, size_t i);
uint64_t so_ntt_inv_root_power(const so_ntt_tables *t, size_t i);

/* ---- coefficient-wise ops on one limb (util/polyarithsmallmod.cpp) ---- */
void so_modulo_poly_coeffs(const uint64_t *a, size_t n, const so_modulus *m, uint64_t *r);
void so_add_poly_coeffmod(const uint64_t *a, const uint64_t *b, size_t n, const so_modulus *m, uint64_t *r);
void so_sub_poly_coeffmod(const uint64_t *a, const uint64_t *b, size_t n, const so_modulus *m, uint64_t *r);
void so_negate_poly_coeffmod(const uint64_t *a, size_t n, const so_modulus *m, uint64_t *r);
void so_add_poly_scalar_coeffmod(const uint64_t *a, size_t n, uint64_t s, const so_modulus *m, uint64_t *r);
void so_sub_poly_scalar_coeffmod(const uint64_t *a, size_t n, uint64_t s, const so_modulus *m, uint64_t *r);
void so_multiply_poly_scalar_coeffmod(const uint64_t *a, size_t n, uint64_t s, const so_modulus *m, uint64_t *r);
void so_dyadic_product_coeffmod(const uint64_t *a, const uint64_t *b, size_t n, const so_modulus *m, uint64_t *r);

/* ---- Galois (util/galois.cpp) ---- */
uint32_t so_galois_elt_from_step(int log_n, int step);
void so_galois_table_ntt(int log_n, uint32_t elt, uint32_t *table);
void so_apply_galois_ntt(const uint64_t *in, int log_n, uint32_t elt, uint64_t *out);
void so_apply_galois(const uint64_t *in, int log_n, uint32_t elt, const so_modulus *m, uint64_t *out);

/* ---- RNS bases / BEHZ tool (util/rns.cpp) ---- */
typedef struct so_rnsbase so_rnsbase;
typedef struct so_rnstool so_rnstool;

so_rnstool *so_rnstool_create(size_t n, const uint64_t *q, size_t q_size, uint64_t t);
void so_rnstool_destroy(so_rnstool *r);
size_t so_rnstool_base_B_size(const so_rnstool *r);
size_t so_rnstool_base_Bsk_size(const so_rnstool *r);
void so_rnstool_base_Bsk(const so_rnstool *r, uint64_t *out); /* B primes then m_sk */
uint64_t so_rnstool_m_tilde(const so_rnstool *r);
void so_rnstool_inv_q_last_mod_q(const so_rnstool *r, uint64_t *operands);
uint64_t so_rnstool_inv_q_last_mod_t(const so_rnstool *r);
void so_mod_t_and_divide_q_last_ntt_inplace(const so_rnstool *r, uint64_t *input, so_ntt_tables *const *tables);
void so_divide_and_round_q_last_inplace(const so_rnstool *r, uint64_t *input);
void so_divide_and_round_q_last_ntt_inplace(const so_rnstool *r, uint64_t *input, so_ntt_tables *const *tables);
void so_fastbconv_m_tilde(const so_rnstool *r, const uint64_t *in, uint64_t *out);
void so_sm_mrq(const so_rnstool *r, const uint64_t *in, uint64_t *out);
void so_fast_floor(const so_rnstool *r, const uint64_t *in, uint64_t *out);
void so_fastbconv_sk(const so_rnstool *r, const uint64_t *in, uint64_t *out);
/* generic BaseConverter::fast_convert_array for tests (rns.cpp:418-463) */
void so_fast_convert_array(const uint64_t *ibase, size_t isz, const uint64_t *obase, size_t osz,
                           const uint64_t *in, size_t count, uint64_t *out);

/* ---- Evaluator-level context + ops (context.cu:607-765, evaluator.cpp) ---- */
enum { SO_SCHEME_BFV = 1, SO_SCHEME_CKKS = 2, SO_SCHEME_BGV = 3 }; /* encryptionparams.h scheme_type */

typedef struct so_context so_context;
/* primes: the key-level modulus chain (k primes, the last is the special prime). */
so_context *so_context_create(int log_n, const uint64_t *primes, size_t k, int scheme, uint64_t plain_modulus);
void so_context_destroy(so_context *c);
const so_ntt_tables *so_context_tables(const so_context *c, size_t i);
const so_rnstool *so_context_rnstool(so_context *c, size_t L);

/* evaluator.cpp:569-708: x has room for 3 polys (x0,x1 filled), y 2 polys; L limbs, NTT form */
void so_ckks_multiply(const so_context *c, size_t L, uint64_t *x, const uint64_t *y);
/* general sizes, evaluator.cpp:664-700; x must have room for (xs+ys-1) polys */
void so_ckks_multiply_general(const so_context *c, size_t L, uint64_t *x, size_t xs, const uint64_t *y, size_t ys);
/* evaluator.cpp:1022-1078, 2135-2172, 1737-1951 (CKKS branch) */
void so_ckks_square(const so_context *c, size_t L, uint64_t *x);
void so_multiply_plain_ntt(const so_context *c, size_t L, uint64_t *ct, size_t size, const uint64_t *plain);
void so_add_plain_ckks(const so_context *c, size_t L, uint64_t *ct, const uint64_t *plain, int subtract);
/* evaluator.cpp:395-567 (sizes 2x2 -> 3, coefficient form) */
void so_bfv_multiply(so_context *c, size_t L, uint64_t *x, const uint64_t *y);
/* evaluator.cpp:2523-2820; encrypted: 2 polys x L limbs; target: L limbs;
   key: [k-1][2][k][N] (KSwitchKeys::data()[index], kswitchkeys.h:90-101) */
void so_switch_key_inplace(const so_context *c, size_t L, uint64_t *encrypted, const uint64_t *target,
                           const uint64_t *key);
/* evaluator.cpp:1144-1199 with encrypted size 3 -> 2: ct has 3 polys on entry */
void so_relinearize(const so_context *c, size_t L, uint64_t *ct, const uint64_t *relin_key);
/* evaluator.cpp:1201-1280: in: size polys x L limbs -> out: size polys x (L-1) limbs */
void so_mod_switch_scale_to_next(so_context *c, size_t L, const uint64_t *in, size_t size, uint64_t *out);
/* evaluator.cpp:1282-1345 */
void so_mod_switch_drop_to_next(const so_context *c, size_t L, const uint64_t *in, size_t size, uint64_t *out);
/* evaluator.cpp:2362-2464 (size-2 ciphertext) */
void so_apply_galois_inplace(const so_context *c, size_t L, uint64_t *ct, uint32_t elt, const uint64_t *galois_key);
/* evaluator.cpp:2267-2360 */
void so_transform_to_ntt(const so_context *c, size_t L, uint64_t *ct, size_t size);
void so_transform_from_ntt(const so_context *c, size_t L, uint64_t *ct, size_t size);

/* ---- key material for tests (keygenerator.cpp:56-334, util/rlwe.cpp:282-400); the PRNG is the
 * harness's own xoshiro256** (bit-exactness is asserted on identical key bytes) ---- */
void so_sample_ternary_ntt(const so_context *c, uint64_t seed, uint64_t *sk_ntt /* k limbs */);
void so_gen_kswitch_key(const so_context *c, const uint64_t *sk_ntt, const uint64_t *new_key_ntt, uint64_t seed,
                        uint64_t *key /* [k-1][2][k][N] */);
void so_gen_relin_key(const so_context *c, const uint64_t *sk_ntt, uint64_t seed, uint64_t *key);
void so_gen_galois_key(const so_context *c, const uint64_t *sk_ntt, uint32_t elt, uint64_t seed, uint64_t *key);
/* symmetric encryption of a coefficient-form RNS plaintext m (L limbs) -> ct (2 polys x L limbs);
   ntt_form selects CKKS (NTT) or BFV (coefficient) output. */
void so_encrypt_symmetric(const so_context *c, size_t L, const uint64_t *sk_ntt, const uint64_t *m, uint64_t seed,
                          int ntt_form, uint64_t *ct);
/* phase = sum_i c_i s^i in coefficient form, L limbs */
void so_decrypt_phase(const so_context *c, size_t L, const uint64_t *sk_ntt, const uint64_t *ct, size_t size,
                      int ntt_form, uint64_t *phase);
void so_fill_uniform(const so_context *c, size_t L, size_t polys, uint64_t seed, uint64_t *out);

/* ---- the reference's own randomness and key material: Blake2xb PRNG (randomgen.cpp:176-211),
 * samplers (util/rlwe.cpp:21-131), KeyGenerator (keygenerator.cpp:56-363), encrypt_zero_* (util/rlwe.cpp:189-412),
 * Encryptor::encrypt_zero_internal (encryptor.cpp:84-180), Decryptor::dot_product_ct_sk_array (decryptor.cu:544-626).
 * factory_seed = the default seed of a Blake2xbPRNGFactory: every generator created starts from it. ---- */
typedef struct so_prng so_prng;
int so_blake2b(void *out, size_t outlen, const void *in, size_t inlen, const void *key, size_t keylen);
int so_blake2xb(void *out, size_t outlen, const void *in, size_t inlen, const void *key, size_t keylen);
so_prng *so_prng_create(const uint64_t seed[8]);
void so_prng_destroy(so_prng *p);
void so_prng_generate(so_prng *p, size_t count, void *dest);
void so_sample_poly_ternary(so_prng *p, const uint64_t *q, size_t limbs, size_t n, uint64_t *dest);
void so_sample_poly_cbd(so_prng *p, const uint64_t *q, size_t limbs, size_t n, uint64_t *dest);
void so_sample_poly_uniform(so_prng *p, const uint64_t *q, size_t limbs, size_t n, uint64_t *dest);
void so_seal_gen_secret_key(const so_context *c, const uint64_t factory_seed[8], uint64_t *sk /* k limbs, NTT */);
void so_seal_encrypt_zero_symmetric(const so_context *c, size_t L, const uint64_t *sk, const uint64_t factory_seed[8],
                                    int ntt_form, uint64_t *ct /* 2 x L x N */);
/* save_seed = true (util/rlwe.cpp:305-317, 347-356, 404-411) and Ciphertext::expand_seed (ciphertext.cpp:153-186) */
void so_seal_encrypt_zero_symmetric_seeded(const so_context *c, size_t L, const uint64_t *sk,
                                           const uint64_t factory_seed[8], int ntt_form, uint64_t *ct);
void so_seal_expand_seed(const so_context *c, size_t L, const uint64_t seed[8], uint64_t *c1 /* L x N */);
void so_seal_encrypt_zero_asymmetric(const so_context *c, size_t L, const uint64_t *pk /* 2 x k x N */,
                                     const uint64_t factory_seed[8], int ntt_form, uint64_t *ct);
/* pk == NULL: symmetric with sk */
void so_seal_encrypt_zero(so_context *c, size_t L, const uint64_t *pk, const uint64_t *sk, const uint64_t factory_seed[8],
                          uint64_t *ct);
void so_seal_gen_kswitch_key(const so_context *c, const uint64_t *sk, const uint64_t *new_key,
                             const uint64_t factory_seed[8], uint64_t *key /* [k-1][2][k][N] */);
void so_seal_gen_relin_key(const so_context *c, const uint64_t *sk, const uint64_t factory_seed[8], uint64_t *key);
void so_seal_gen_galois_key(const so_context *c, const uint64_t *sk, uint32_t elt, const uint64_t factory_seed[8],
                            uint64_t *key);
void so_seal_dot_product_ct_sk(const so_context *c, size_t L, const uint64_t *sk, const uint64_t *ct, size_t size,
                               int ntt_form, uint64_t *out /* L limbs */);

/* evaluator.cpp:56-118: BGV operands with different correction factors; out = {new factor, scalar for the first,
 * scalar for the second}; -1 when factor1 is not invertible */
int so_balance_correction_factors(uint64_t factor1, uint64_t factor2, uint64_t t, uint64_t out[3]);

#ifdef __cplusplus
}
#endif
#endif
