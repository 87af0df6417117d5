/*
 * cat_amd.h -- C ABI of the MI355X-native RNS polynomial arithmetic library (libcat_amd.so).
 *
 * This is the drop-in boundary for the hot path of Rayman96/CAT (a CUDA fork of SEAL 4.0): every
 * entry point below names the reference interface it replaces (file:line, paths relative to
 * /root/reference/native/src/seal/).  The reference has no FFI layer of its own (the CUDA code is
 * compiled into the seal:: classes); the C++ facade in cat_amd/host/seal_hip.hpp binds
 * these symbols behind the reference's class and function names.
 *
 * Conventions
 *  - All pointers named d_* are DEVICE pointers (hipMalloc'ed or torch-owned), 16-byte aligned (the kernels move
 *    16 bytes per lane; a misaligned operand is refused with CAT_ERR_INVALID_ARG).
 *  - Polynomial layout is the reference's: poly[limb][coeff], limbs contiguous, N words each
 *    (RNSIter, util/iterator.h:951); a ciphertext is size x L x N words (ciphertext.cuh:423-434);
 *    a batch is `batch` such objects back to back.
 *  - Limb i of a level-L object is reduced modulo primes[i] of the context (context.cu:607-637).
 *  - Every call returns a cat_status, takes an explicit stream (hipStream_t as void*, NULL = default
 *    stream), is asynchronous with respect to the host, never throws and never exits.
 *  - A call makes the context's device current while it runs and restores the calling thread's current
 *    device before it returns.
 *  - Error codes mirror the reference's exceptions (evaluator.cpp:2534-2573): CAT_ERR_INVALID_ARG =
 *    std::invalid_argument, CAT_ERR_LOGIC = std::logic_error, CAT_ERR_OUT_OF_RANGE =
 *    std::out_of_range.  cat_last_error() returns the message of the calling thread's last failure.
 *  - Temporaries come from a context-owned device workspace that grows on demand (hipMalloc happens
 *    only when a call needs more than any earlier call; cat_ctx_reserve() pre-sizes it).  Growing inside a
 *    call means a device-wide synchronisation, hipFree and hipMalloc: cat_ctx_workspace_grew() counts those
 *    events and cat_ctx_set_workspace_strict() turns them into CAT_ERR_LOGIC.
 *  - One context per device and per host thread of control; calls on one context must not overlap.
 */
#ifndef CAT_AMD_H
#define CAT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cat_ctx cat_ctx;
typedef void *cat_stream; /* hipStream_t */

typedef enum {
    CAT_OK = 0,
    CAT_ERR_INVALID_ARG = 1,
    CAT_ERR_LOGIC = 2,
    CAT_ERR_OUT_OF_RANGE = 3,
    CAT_ERR_HIP = 4,
    CAT_ERR_NOMEM = 5
} cat_status;

enum { CAT_SCHEME_NONE = 0, CAT_SCHEME_BFV = 1, CAT_SCHEME_CKKS = 2, CAT_SCHEME_BGV = 3 }; /* scheme_type, encryptionparams.h */

#define CAT_MAX_MODULI 64 /* coefficient primes + BFV auxiliary primes per context */
#define CAT_MAX_KEYSWITCH_LIMBS 63 /* decomposition digits of one key switch (= ciphertext limbs L); < CAT_MAX_MODULI */
#define CAT_MAX_LOG_N 16
#define CAT_MIN_LOG_N 1

/* ---- library / context -------------------------------------------------------------------------- */

const char *cat_version(void);
const char *cat_last_error(void);

/* Replaces SEALContext construction for this path (context.cu:292-356, 607-765): validates the
 * primes (prime, = 1 mod 2N, <= 61 bits), builds per-prime NTTTables (util/ntt.cpp:240-299), Barrett
 * ratios (modulus.cpp:87-98), q_a^-1 mod q_b tables (util/rns.cpp:767-776) and, for BFV, the BEHZ
 * bases of every level (util/rns.cpp:578-787), and uploads them to `device`.
 * primes[0..n_primes) is the key-level chain; when n_primes > 1 the last prime is the special prime
 * of key switching.  plain_modulus is used by BFV and BGV (BGV: it must be coprime to every prime, since the
 * levels hold q_last^-1 mod t, util/rns.cpp:767-776). */
cat_status cat_ctx_create(uint32_t log_n, const uint64_t *primes, uint32_t n_primes, int scheme,
                          uint64_t plain_modulus, int device, cat_ctx **out);
cat_status cat_ctx_destroy(cat_ctx *ctx);
cat_status cat_ctx_reserve(cat_ctx *ctx, size_t workspace_bytes);
size_t cat_ctx_workspace_bytes(const cat_ctx *ctx);
/* Bytes of workspace the largest call on `batch` ciphertexts (size <= 3, any level, any entry point) needs: reserving
 * that much once (cat_ctx_reserve) means no later call synchronises, frees or allocates.  The C++ facade does so for
 * batch 1 when it builds a SEALContext. */
size_t cat_ctx_workspace_need(const cat_ctx *ctx, uint32_t batch);
/* No counterpart in the reference (its pool allocates silently, gpu_memorypool.h:64-131).  How many times a call had to
 * grow the workspace itself (each time: hipDeviceSynchronize + hipFree + hipMalloc inside an otherwise asynchronous
 * call); cat_ctx_reserve does not count.  A caller that reserved cat_ctx_workspace_need(batch) expects this to stay 0. */
uint64_t cat_ctx_workspace_grew(const cat_ctx *ctx);
/* on != 0: a call that would have to grow the workspace fails with CAT_ERR_LOGIC ("the workspace would have to grow to
 * ... bytes") before launching anything, instead of synchronising and reallocating: for callers under HIP graph capture
 * or with latency budgets.  cat_ctx_reserve still grows it. */
cat_status cat_ctx_set_workspace_strict(cat_ctx *ctx, int on);
uint32_t cat_ctx_log_n(const cat_ctx *ctx);
uint32_t cat_ctx_n_primes(const cat_ctx *ctx);
/* NTTTables::get_root() (util/ntt.h:101): the minimal primitive 2N-th root used for prime i */
uint64_t cat_ctx_root(const cat_ctx *ctx, uint32_t prime_index);
/* Number of moduli known to the context (coefficient primes first, then BFV auxiliary primes:
 * m_sk, gamma, B_0.., util/rns.cpp:625-635) and their values. */
uint32_t cat_ctx_n_moduli(const cat_ctx *ctx);
uint64_t cat_ctx_modulus(const cat_ctx *ctx, uint32_t modulus_index);

/* Device memory helpers (Ciphertext::to_gpu/to_cpu, ciphertext.cuh:361-381).  Optional: any device
 * pointer works with the calls below. */
cat_status cat_malloc(cat_ctx *ctx, size_t bytes, void **d_ptr);
cat_status cat_free(cat_ctx *ctx, void *d_ptr);
/* The reference keeps freed device blocks in a pool (gpu_memorypool.h:64-131, size-keyed free lists).  max_cached_bytes
 * > 0: cat_free keeps blocks of this context (up to that many bytes in total) and cat_malloc hands a kept block of exactly
 * the requested size out again -- no hipFree (which synchronises the whole device) and no hipMalloc on a caller's
 * steady-state path: Evaluator::multiply_inplace's resize to three polynomials went from about 45 us to about 1 us.
 * Contract: a kept block is reused in STREAM order only, so with the pool on the caller issues the work that touches
 * cat_malloc'ed blocks on one stream (the C++ facade uses the default stream throughout), or synchronises before
 * cat_free.  0 switches the pool off and releases what it holds (the default: cat_free = hipFree). */
cat_status cat_ctx_set_pool(cat_ctx *ctx, size_t max_cached_bytes);
cat_status cat_upload(cat_ctx *ctx, void *d_dst, const void *h_src, size_t bytes, cat_stream s);
cat_status cat_download(cat_ctx *ctx, void *h_dst, const void *d_src, size_t bytes, cat_stream s);
/* device-to-device copy of `bytes` bytes (the reference copies a Ciphertext's device shadow the same way,
 * ciphertext.cuh:284-316); the ranges must not overlap */
cat_status cat_copy(cat_ctx *ctx, void *d_dst, const void *d_src, size_t bytes, cat_stream s);
cat_status cat_stream_sync(cat_ctx *ctx, cat_stream s);

/* ---- a1/a2/a13: negacyclic NTT -------------------------------------------------------------------
 * util::ntt_negacyclic_harvey[_lazy] (util/ntt.cpp:393-436, ntt.h:201-270) and
 * util::inverse_ntt_negacyclic_harvey[_lazy] (util/ntt.cpp:438-474), batched: d_data holds n_polys
 * objects of n_limbs limbs; limb j is transformed with the tables of modulus limb_ids[j]
 * (limb_ids == NULL means 0..n_limbs-1).  In place.  Forward: input < 4q, output bit-reversed order,
 * < 4q when lazy else < q.  Inverse: input < 2q bit-reversed order, output < 2q when lazy else < q.
 * The lazy outputs are the reference's exact representatives (same butterfly network). */
cat_status cat_ntt_forward(cat_ctx *ctx, uint64_t *d_data, const uint32_t *limb_ids, uint32_t n_limbs,
                           uint32_t n_polys, int lazy, cat_stream s);
cat_status cat_ntt_inverse(cat_ctx *ctx, uint64_t *d_data, const uint32_t *limb_ids, uint32_t n_limbs,
                           uint32_t n_polys, int lazy, cat_stream s);

/* ---- a4/a5/a6: coefficient-wise ops (util/polyarithsmallmod.cpp:18-285) ---------------------------
 * Operands are n_polys x n_limbs x N; limb j uses modulus limb_ids[j] (NULL = 0..n_limbs-1).
 * Outputs may alias inputs. */
cat_status cat_add(cat_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out,
                   const uint32_t *limb_ids, uint32_t n_limbs, uint32_t n_polys, cat_stream s); /* :43-86 */
cat_status cat_sub(cat_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out,
                   const uint32_t *limb_ids, uint32_t n_limbs, uint32_t n_polys, cat_stream s); /* :88-131 */
cat_status cat_negate(cat_ctx *ctx, const uint64_t *d_a, uint64_t *d_out, const uint32_t *limb_ids,
                      uint32_t n_limbs, uint32_t n_polys, cat_stream s); /* .cuh:84-117 */
cat_status cat_dyadic(cat_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out,
                      const uint32_t *limb_ids, uint32_t n_limbs, uint32_t n_polys, cat_stream s); /* :227-285 */
/* scalars[j] is the scalar of limb j (any 64-bit value; reduced mod q_j first, .cuh:477-486) */
cat_status cat_mul_scalar(cat_ctx *ctx, const uint64_t *d_a, const uint64_t *scalars, uint64_t *d_out,
                          const uint32_t *limb_ids, uint32_t n_limbs, uint32_t n_polys, cat_stream s); /* :193-225 */
/* scalars[j] must be < q_j (the reference's precondition, :150-153) */
cat_status cat_add_scalar(cat_ctx *ctx, const uint64_t *d_a, const uint64_t *scalars, uint64_t *d_out,
                          const uint32_t *limb_ids, uint32_t n_limbs, uint32_t n_polys, cat_stream s); /* :133-161 */
cat_status cat_sub_scalar(cat_ctx *ctx, const uint64_t *d_a, const uint64_t *scalars, uint64_t *d_out,
                          const uint32_t *limb_ids, uint32_t n_limbs, uint32_t n_polys, cat_stream s); /* :163-191 */
cat_status cat_modulo(cat_ctx *ctx, const uint64_t *d_a, uint64_t *d_out, const uint32_t *limb_ids,
                      uint32_t n_limbs, uint32_t n_polys, cat_stream s); /* :18-41 */

/* ---- a7: Evaluator::ckks_multiply, size 2 x 2 -> 3 (evaluator.cpp:569-708) -------------------------
 * d_x: batch x 2 x L x N, d_y: batch x 2 x L x N, d_out: batch x 3 x L x N (NTT form).  d_out must
 * not alias the inputs. */
cat_status cat_ckks_multiply(cat_ctx *ctx, const uint64_t *d_x, const uint64_t *d_y, uint64_t *d_out, uint32_t L,
                             uint32_t batch, cat_stream s);

/* ---- f4 ("next" row 4): BGV ----------------------------------------------------------------
 * With a CAT_SCHEME_BGV context (NTT-form ciphertexts) cat_switch_key / cat_relinearize /
 * cat_apply_galois_inplace take the BGV mod-down branch (evaluator.cpp:2715-2758) and cat_rescale[_strided] is
 * RNSTool::mod_t_and_divide_q_last_ntt_inplace (util/rns.cpp:1203-1246) = mod_switch_to_next.
 * Evaluator::bgv_multiply, size 2 x 2 -> 3 (evaluator.cpp:710-841): same layout and contract as cat_ckks_multiply. */
cat_status cat_bgv_multiply(cat_ctx *ctx, const uint64_t *d_x, const uint64_t *d_y, uint64_t *d_out, uint32_t L,
                            uint32_t batch, cat_stream s);
/* RNSTool::inv_q_last_mod_t() of the level with L limbs (util/rns.cpp:778-786): the factor mod switching multiplies
 * into Ciphertext::correction_factor() (evaluator.cpp:1274-1279; bookkeeping stays with the caller).  0 when the
 * context is not BGV or L is out of range. */
uint64_t cat_bgv_inv_q_last_mod_t(const cat_ctx *ctx, uint32_t L);

/* ---- f1 ("next" row 1): squares and plaintext operands, same layout and kernels' arithmetic -------
 * Evaluator::ckks_square, size 2 -> 3 (evaluator.cpp:1022-1078): (c0^2, 2 c0 c1, c1^2); d_out must not alias d_x. */
cat_status cat_ckks_square(cat_ctx *ctx, const uint64_t *d_x, uint64_t *d_out, uint32_t L, uint32_t batch, cat_stream s);
/* Evaluator::multiply_plain_ntt (evaluator.cpp:2135-2172): every polynomial of each ciphertext times the NTT-form
 * plaintext.  d_ct, d_out: batch x size x L x N (may alias); d_plain: L x N words per plaintext, plain_stride words
 * apart (0 = one plaintext shared by the batch).  The scale bookkeeping stays with the caller. */
cat_status cat_multiply_plain_ntt(cat_ctx *ctx, const uint64_t *d_ct, const uint64_t *d_plain, size_t plain_stride,
                                  uint64_t *d_out, uint32_t L, uint32_t size, uint32_t batch, cat_stream s);
/* Evaluator::add_plain_inplace / sub_plain_inplace, CKKS branch (evaluator.cpp:1737-1951): c0 +- plain, both in NTT
 * form at the same level and scale; the other polynomials are copied when d_out != d_ct. */
cat_status cat_add_plain(cat_ctx *ctx, const uint64_t *d_ct, const uint64_t *d_plain, size_t plain_stride,
                         uint64_t *d_out, uint32_t L, uint32_t size, uint32_t batch, cat_stream s);
cat_status cat_sub_plain(cat_ctx *ctx, const uint64_t *d_ct, const uint64_t *d_plain, size_t plain_stride,
                         uint64_t *d_out, uint32_t L, uint32_t size, uint32_t batch, cat_stream s);

/* ---- a8: Evaluator::switch_key_inplace (evaluator.cpp:2523-2820), CKKS and BFV branches ------------
 * d_encrypted: batch x 2 x L x N, updated in place (encrypted[comp][j] += mod-down result).
 * d_target:    batch x L x N (NTT form for CKKS, coefficient form for BFV); must not alias d_encrypted.
 * d_key:       one KSwitchKeys::data()[index] entry = [n_primes-1][2][n_primes][N] words, NTT form
 *              (kswitchkeys.h:90-101), shared by the whole batch.
 * Errors: L == 0 or L >= n_primes -> CAT_ERR_INVALID_ARG; scheme NONE -> CAT_ERR_LOGIC
 * ("keyswitching is not supported by the context", :2543-2546). */
cat_status cat_switch_key(cat_ctx *ctx, uint64_t *d_encrypted, const uint64_t *d_target, const uint64_t *d_key,
                          uint32_t L, uint32_t batch, cat_stream s);
/* No counterpart in the reference (single GPU, single process: SURVEY.md 2a).  Replicates a key (or any table of
 * n_words words) from rank `root` of an RCCL communicator to every rank's d_key, in place, in buckets of bucket_words
 * words (0 = 256 MB): the one collective of the path, issued once per key outside the steady state.
 * comm is the caller's ncclComm_t (one rank per GPU, created with ncclCommInitRank); comm == NULL means a single process
 * and is a no-op.  librccl is bound at run time; CAT_ERR_LOGIC when it cannot be loaded, CAT_ERR_HIP with RCCL's message
 * when the collective fails.  Asynchronous on stream s like every other call. */
cat_status cat_key_broadcast(cat_ctx *ctx, void *comm, uint64_t *d_key, size_t n_words, int root, size_t bucket_words,
                             cat_stream s);
/* No counterpart in the reference either.  The optional second collective: every rank contributes the
 * n_words words of its result slice at d_local and receives all slices in rank order in d_all (world x n_words words;
 * ncclAllGather, so equal counts -- a caller with uneven slices pads the short ones).  d_all + rank * n_words may be
 * d_local (in place).  comm == NULL: a single process, d_all receives a copy of d_local.  Errors as cat_key_broadcast. */
cat_status cat_result_gather(cat_ctx *ctx, void *comm, const uint64_t *d_local, size_t n_words, uint64_t *d_all,
                             cat_stream s);
/* Evaluator::relinearize_internal for size 3 -> 2 (evaluator.cpp:1144-1199): d_ct3 is
 * batch x 3 x L x N; on return the first two polys of each ciphertext hold the result (the caller
 * "resizes" by ignoring the third).  d_key = RelinKeys key of index get_index(2) = 0 (relinkeys.h:58-64). */
cat_status cat_relinearize(cat_ctx *ctx, uint64_t *d_ct3, const uint64_t *d_key, uint32_t L, uint32_t batch,
                           cat_stream s);

/* Evaluator::relinearize_inplace (size 3 -> 2, evaluator.cpp:1144-1199) followed by rescale_to_next / BFV-BGV
 * mod_switch_to_next (:1201-1280, :1481-1519) as one call: d_out (batch x 2 x (L-1) x N, must not overlap d_ct3) receives
 * exactly what the two calls give one after the other.  The reference has no such method; for CKKS and BGV at N >= 4096 the
 * mod-down correction and the rescale correction of a limb share one forward transform (the transform is linear and
 * the result canonical, so the words are identical), which removes (L-1) x 2 transforms per ciphertext.  d_ct3
 * (batch x 3 x L x N) is scratch afterwards.  Errors as cat_relinearize and cat_rescale. */
cat_status cat_relinearize_rescale(cat_ctx *ctx, uint64_t *d_ct3, const uint64_t *d_key, uint64_t *d_out, uint32_t L,
                                   uint32_t batch, cat_stream s);

/* ---- a9/a10: mod switching ------------------------------------------------------------------------
 * Evaluator::mod_switch_scale_to_next (evaluator.cpp:1201-1280) = rescale_to_next (:1481-1519):
 * CKKS: RNSTool::divide_and_round_q_last_ntt_inplace (util/rns.cpp:830-901); BFV:
 * divide_and_round_q_last_inplace (:789-828).  d_in: batch x size x L x N -> d_out: batch x size x
 * (L-1) x N.  d_out may equal d_in (the compaction is ordered).  The `scale_ /= q_last` bookkeeping
 * (:1271) stays with the caller. */
cat_status cat_rescale(cat_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, uint32_t L, uint32_t size,
                       uint32_t batch, cat_stream s);
/* Same, for batches whose ciphertexts are not back to back: in_stride / out_stride are the words
 * between consecutive ciphertexts (e.g. 3*L*N right after cat_relinearize, whose "resize to 2",
 * evaluator.cpp:1191, leaves the third polynomial in place). */
cat_status cat_rescale_strided(cat_ctx *ctx, const uint64_t *d_in, size_t in_stride, uint64_t *d_out,
                               size_t out_stride, uint32_t L, uint32_t size, uint32_t batch, cat_stream s);
/* No counterpart in the reference.  cat_rescale for callers that cannot allocate device memory: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  Same arguments, same words in d_out; its temporaries
 * (batch x size last limbs and batch x size x (L-1) correction limbs) live in a static device array of
 * CAT_NOALLOC_SCRATCH_BYTES per GPU instead of the context workspace, so the call performs no hipMalloc / hipFree and
 * never synchronises.  Limits: batch * size * L * N * 8 bytes <= CAT_NOALLOC_SCRATCH_BYTES (else CAT_ERR_INVALID_ARG,
 * "operands exceed the static scratch buffer"); d_out must not overlap d_in (CAT_ERR_INVALID_ARG).  Every call on a GPU
 * shares that GPU's array, from any context or thread: calls that overlap in time on one GPU race.  Nothing in the array
 * needs to start at zero (every word is written before it is read within a call). */
#define CAT_NOALLOC_SCRATCH_BYTES (64u << 20)
cat_status cat_rescale_noalloc(cat_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, uint32_t L, uint32_t size,
                               uint32_t batch, cat_stream s);
/* Evaluator::mod_switch_drop_to_next (evaluator.cpp:1282-1345): drop the last limb of every poly. */
cat_status cat_mod_drop(cat_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, uint32_t L, uint32_t size,
                        uint32_t batch, cat_stream s);

/* ---- a11: Galois ------------------------------------------------------------------------------------
 * GaloisTool::get_elt_from_step (util/galois.cpp:53-95); returns 0 when |step| >= N/2 (the reference
 * throws invalid_argument("step count too large")). */
uint32_t cat_galois_elt_from_step(uint32_t log_n, int step);
/* GaloisTool::apply_galois_ntt (util/galois.cpp:192-218) / apply_galois (:148-190) on n_polys x n_limbs
 * x N; d_out must not alias d_in (:160-163 -> CAT_ERR_INVALID_ARG). */
cat_status cat_apply_galois_ntt(cat_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, uint32_t galois_elt,
                                uint32_t n_limbs, uint32_t n_polys, cat_stream s);
cat_status cat_apply_galois(cat_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, uint32_t galois_elt,
                            const uint32_t *limb_ids, uint32_t n_limbs, uint32_t n_polys, cat_stream s);
/* Evaluator::apply_galois_inplace (evaluator.cpp:2362-2464) on batch x 2 x L x N ciphertexts with one
 * GaloisKeys entry (index GaloisKeys::get_index(elt), galoiskeys.h:48-50); rotate_vector is this with
 * elt = cat_galois_elt_from_step(step) (evaluator.cpp:2466-2521). */
cat_status cat_apply_galois_inplace(cat_ctx *ctx, uint64_t *d_ct, uint32_t galois_elt, const uint64_t *d_key,
                                    uint32_t L, uint32_t batch, cat_stream s);

/* apply_galois_inplace (rotate_vector) directly followed by rescale_to_next / mod_switch_to_next, as one call with the
 * same words as the two (see cat_relinearize_rescale): d_out is batch x 2 x (L-1) x N and must not overlap d_ct, which
 * is scratch afterwards. */
cat_status cat_apply_galois_rescale(cat_ctx *ctx, uint64_t *d_ct, uint32_t galois_elt, const uint64_t *d_key,
                                    uint64_t *d_out, uint32_t L, uint32_t batch, cat_stream s);

/* ---- f4 ("next" row 4), second half: hoisted rotations ------------------------------------------------
 * No counterpart in the reference: Evaluator::rotate_internal runs one full switch_key_inplace per Galois element
 * (evaluator.cpp:2466-2521).  cat_rotate_hoisted applies n_rot Galois elements to the SAME batch of ciphertexts and shares
 * what does not depend on the element -- the inverse transform of c1, the modulus raise and the d (d + 1) forward
 * transforms of the extended digits (evaluator.cpp:2604-2654) -- so that each further element costs an inner product, one
 * permutation of the product and the mod-down.
 *   d_ct:   batch x 2 x L x N (CKKS / BGV: NTT form, BFV: coefficient form), read only
 *   d_keys_prepared[r]: device pointer to the key of galois_elts[r] after cat_hoisted_key_prepare (host array of n_rot pointers)
 *   d_outs[r]: device pointer to the output batch of galois_elts[r], batch x 2 x L x N each (host array of n_rot pointers);
 *           none may overlap d_ct, and they must not overlap each other
 * The results are NOT the reference's words: the reference decomposes sigma_g(c1), this call permutes the digits of c1
 * after the decomposition, and the canonical digits of a negated coefficient (q_J - t) do not reduce to minus the digits'
 * residues.  Both are key switches with digits below q_J: every output decrypts to what apply_galois_inplace's output
 * decrypts to, within the key-switching noise (tests/test_gpu_hoisted.py checks exactly that against the oracle). */
cat_status cat_rotate_hoisted(cat_ctx *ctx, const uint64_t *d_ct, const uint32_t *galois_elts,
                              const uint64_t *const *d_keys_prepared, uint32_t n_rot, uint64_t *const *d_outs, uint32_t L,
                              uint32_t batch, cat_stream s);
/* d_key_out <- sigma_g^-1 applied to every limb of the GaloisKeys entry of galois_elt (same shape, [n_primes-1][2][n_primes][N];
 * must not alias d_key): sigma_g(ext) * key = sigma_g(ext * sigma_g^-1(key)) lets the hoisted inner product read the
 * extended digits where they lie.  Once per key. */
cat_status cat_hoisted_key_prepare(cat_ctx *ctx, const uint64_t *d_key, uint32_t galois_elt, uint64_t *d_key_out, cat_stream s);

/* ---- a12: Evaluator::bfv_multiply, size 2 x 2 -> 3 (evaluator.cpp:395-567) with the BEHZ tool
 * (util/rns.cpp:418-463, 903-1131).  d_x, d_y: batch x 2 x L x N coefficient form; d_out: batch x 3 x L x N. */
cat_status cat_bfv_multiply(cat_ctx *ctx, const uint64_t *d_x, const uint64_t *d_y, uint64_t *d_out, uint32_t L,
                            uint32_t batch, cat_stream s);
/* BEHZ pieces, exposed for parity tests against tests/util/rns.cpp:460-853.  All operate on one
 * polynomial of the level with L limbs: q-limbs first, then Bsk (B primes, m_sk), then m_tilde. */
uint32_t cat_bfv_base_Bsk_size(const cat_ctx *ctx, uint32_t L);
cat_status cat_bfv_base_Bsk(const cat_ctx *ctx, uint32_t L, uint64_t *out_moduli);
cat_status cat_fastbconv_m_tilde(cat_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, uint32_t L, uint32_t n_polys,
                                 cat_stream s); /* rns.cpp:1092-1131: q -> Bsk U {m_tilde} */
cat_status cat_sm_mrq(cat_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, uint32_t L, uint32_t n_polys,
                      cat_stream s); /* rns.cpp:981-1041: Bsk U {m_tilde} -> Bsk */
cat_status cat_fast_floor(cat_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, uint32_t L, uint32_t n_polys,
                          cat_stream s); /* rns.cpp:1043-1090: q U Bsk -> Bsk; input residues canonical, as the reference
                                          * requires (values up to 2^62 are accepted: the Bsk part enters a lazy 128-bit sum) */
cat_status cat_fastbconv_sk(cat_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, uint32_t L, uint32_t n_polys,
                            cat_stream s); /* rns.cpp:903-979: Bsk -> q */

#ifdef __cplusplus
}
#endif
#endif
