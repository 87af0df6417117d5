// Internal context shared by the C-ABI translation units (not installed).
#pragma once
#include "../../include/cat_amd.h"
#include "host_math.hpp"
#include "modarith.cuh"

#include <hip/hip_runtime.h>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

namespace cat
{
    // BEHZ tool of one level (util/rns.cpp:578-787), device-resident constants.
    struct BfvLevel
    {
        bool ready = false;
        uint32_t L = 0, B = 0, Bsk = 0; // |q|, |B|, |Bsk| = B + 1
        LimbMap map_q{}, map_Bsk{};     // modulus indices
        uint32_t idx_msk = 0;
        // device arrays (u64 unless noted)
        u64 *d_blob = nullptr;
        // offsets into d_blob (in words)
        size_t o_inv_punc_q = 0;   // [L] MulOp: (q/q_i)^-1 mod q_i
        size_t o_mtilde_q = 0;     // [L] MulOp: m_tilde * (q/q_i)^-1 mod q_i (the two reference multiplications as one constant)
        size_t o_q_to_Bsk = 0;     // [Bsk][L]: (q/q_i) mod Bsk_o
        size_t o_q_to_mt = 0;      // [L]: (q/q_i) mod m_tilde
        // the conversion matrix with a tail folded in (bfv.hip bconv_kernel EP 1 / 2): one more column, one reduction per output
        size_t o_q_to_Bsk_mrq = 0;      // [Bsk][L+1]: (q/q_i) m_tilde^-1 mod Bsk_o | (q mod Bsk_o) m_tilde^-1
        size_t o_q_to_Bsk_floor = 0;    // [Bsk][L+1]: -(q/q_i) q^-1 mod Bsk_o | q^-1 mod Bsk_o
        size_t o_q_to_Bsk_floor_t = 0;  // [Bsk][L+1]: the same                | t q^-1 mod Bsk_o
        size_t o_inv_punc_B = 0;   // [B] MulOp: (B/b_i)^-1 mod b_i
        size_t o_B_to_q = 0;       // [L][B]: (B/b_i) mod q_o
        size_t o_B_to_msk = 0;     // [B]: (B/b_i) mod m_sk
        size_t o_prod_q_Bsk = 0;   // [Bsk] MulOp: q mod Bsk_i
        size_t o_inv_mt_Bsk = 0;   // [Bsk] MulOp: m_tilde^-1 mod Bsk_i
        size_t o_inv_q_Bsk = 0;    // [Bsk] MulOp: q^-1 mod Bsk_i
        size_t o_prodB_q = 0;      // [L] MulOp: B mod q_i
        size_t o_negprodB_q = 0;   // [L] MulOp: -B mod q_i
        size_t o_t_q = 0;          // [L] MulOp: t * (q/q_i)^-1 mod q_i
        size_t o_t_Bsk = 0;        // [Bsk] MulOp: t mod Bsk_i
        u64 neg_inv_q_mt = 0;      // -q^-1 mod m_tilde
        MulOp inv_B_msk{};         // B^-1 mod m_sk
    };
} // namespace cat

struct cat_ctx
{
    int device = 0;
    uint32_t log_n = 0;
    size_t n = 0;
    uint32_t k = 0;     // coefficient primes (key level)
    uint32_t n_mod = 0; // k + BFV auxiliary primes
    int scheme = 0;
    cat::u64 plain_modulus = 0;
    std::vector<cat::HostMod> hmods;
    std::vector<cat::u64> roots;

    DevMod *d_mods = nullptr;
    MulOp *d_tw_fwd = nullptr; // [n_mod][n]
    MulOp *d_tw_inv = nullptr; // [n_mod][n]
    cat::u64 *d_tw3_fwd = nullptr; // [n_mod][3][n] split-Shoup form of the same twiddles (fast path): w0 | w1 | q0,q1
    cat::u64 *d_tw3_inv = nullptr; // [n_mod][3][n], shifted by one entry
    NttMod *d_nttmods = nullptr; // [n_mod]
    MulOp *d_inv_mod = nullptr; // [k][k]: entry [a][b] = q_a^-1 mod q_b
    std::vector<MulOp> h_inv_mod;
    cat::u64 *d_half_mod = nullptr; // [k][k]: entry [a][b] = (q_a >> 1) mod q_b
    std::vector<cat::u64> h_half_mod;
    MulOp *d_q_mod = nullptr; // [k][k]: entry [a][b] = q_a mod q_b as a multiply operand modulo q_b (BGV)
    // BGV (util/rns.cpp:767-776): the plain modulus t and q_i^-1 mod t for every coefficient prime
    cat::HostMod pt;
    std::vector<cat::u64> inv_q_mod_t;

    void *ws = nullptr;
    size_t ws_bytes = 0;
    uint64_t ws_grew = 0;   // times the workspace had to grow inside a call (synchronise + hipFree + hipMalloc)
    bool ws_strict = false; // cat_ctx_set_workspace_strict: a call that would grow it fails with CAT_ERR_LOGIC instead
    cat_status ws_status = CAT_OK; // why the last workspace() request returned nullptr

    // cat_ctx_set_pool: blocks handed back with cat_free are kept (by exact size) for the next cat_malloc
    size_t pool_cap = 0, pool_held = 0;
    std::multimap<size_t, void *> pool_free;
    std::unordered_map<void *, size_t> pool_live; // blocks cat_malloc handed out while the pool was on

    std::vector<cat::BfvLevel> bfv; // index L

    LimbMap identity_map() const
    {
        LimbMap m{};
        for (int i = 0; i < 64; ++i)
            m.id[i] = (unsigned char)i;
        return m;
    }
};

namespace cat
{
    void set_error(const std::string &msg);
    cat_status fail(cat_status code, const std::string &msg);
    cat_status hip_fail(hipError_t e, const char *what);
    // Grows the context workspace to at least `bytes`; returns nullptr (and sets the error and ctx->ws_status) on
    // failure: out of memory, or a growth refused in strict mode.
    void *workspace(cat_ctx *ctx, size_t bytes);
    inline cat_status workspace_status(const cat_ctx *ctx) { return ctx->ws_status == CAT_OK ? CAT_ERR_NOMEM : ctx->ws_status; }

    // Workspace words of every entry point that takes temporaries.  The launchers size their request with these and
    // cat_ctx_workspace_need takes their maximum, so the two cannot drift apart.
    inline size_t ws_words_switch_key(size_t N, size_t L, size_t batch, bool with_rescale)
    {
        // t_target [b][L] | ext [b][L+1][L] | prod [b][2][L+1] | tmp [b][2][L] | last [b][2][2] (relinearize + rescale)
        return batch * (L + (L + 1) * L + 2 * (L + 1) + 2 * L + (with_rescale ? 4 : 0)) * N;
    }
    inline size_t ws_words_galois(size_t N, size_t L, size_t batch)
    {
        return ws_words_switch_key(N, L, batch, true) + batch * 2 * L * N; // the permuted copy behind the key-switch region
    }
    inline size_t ws_words_rotate_hoisted(size_t N, size_t L, size_t batch)
    {
        // t_target [b][L] | ext [b][L+1][L] | prod, permuted prod [b][2][L+1] each | tmp, permuted ct [b][2][L] each
        return batch * (L + (L + 1) * L + 4 * (L + 1) + 4 * L) * N;
    }
    inline size_t ws_words_rescale(size_t N, size_t L, size_t polys)
    {
        return polys * N + 2 * polys * (L - 1) * N; // last | tmp | stage (aliased output)
    }
    inline size_t ws_words_mod_drop(size_t N, size_t L, size_t polys) { return polys * (L - 1) * N; }
    inline size_t ws_words_bfv_multiply(size_t N, size_t L, size_t Bsk, size_t batch)
    {
        // inq [b][4][L] | inB [b][4][Bsk] | tq [b][3][L] | tB [b][3][Bsk] | fl [b][3][Bsk]
        return batch * (4 * L + 4 * Bsk + 3 * L + 3 * Bsk + 3 * Bsk) * N;
    }

    // Makes the context's device current for the duration of an entry point and restores the caller's device on the
    // way out (the entry points used to leave the calling thread on the context's device).
    struct DeviceScope
    {
        int prev = -1;
        bool switched = false;
        cat_status enter(const cat_ctx *ctx);
        cat_status enter_device(int device);
        ~DeviceScope()
        {
            if (switched)
                (void)hipSetDevice(prev);
        }
    };
    cat_status make_map(const cat_ctx *ctx, const uint32_t *limb_ids, uint32_t n_limbs, LimbMap &out);

    // ---- kernel launchers (ntt.hip, polyops.hip, keyswitch.hip, bfv.hip) ----
    // exact = true reproduces SEAL's lazy representatives bit for bit (the C ABI's lazy calls); internal callers
    // and fully reduced transforms use the faster split-Shoup arithmetic (same values modulo q, same ranges).
    cat_status launch_ntt(cat_ctx *ctx, bool forward, u64 *d_data, const LimbMap &map, uint32_t n_limbs,
                          uint32_t n_polys, bool lazy, hipStream_t s, bool exact = false);
    // Work fused into a forward transform (key switching and rescale, keyswitch.hip):
    //  - source of the first pass: instead of the data itself, limb values derived from another buffer
    //      kind 0 "raise"  (evaluator.cpp:2643-2651): limb j of object p reads limb (j % limb_group) of object p of
    //                       src, canonical modulo that limb's prime, reduced modulo prime of j when that is smaller;
    //      kind 1 "round"  (evaluator.cpp:2766-2789, util/rns.cpp:857-878): every limb of object p reads the single
    //                       polynomial src + p * src_stride holding values < 2Q (Q = prime src_qidx) and forms
    //                       ((v + Q/2) mod Q) mod q_j + (q_j - (Q/2 mod q_j));
    //  - epilogue of the last pass (evaluator.cpp:2810-2816, util/rns.cpp:879-899): the transformed limb t (< 4q) is
    //      not stored; dest = (other + 4q - t) * Q^-1 mod q  [+ dest when ep_acc], with object p = b * ep_S + s
    //      addressed as base + b * *_b + s * *_s + limb * N.
    // range policy of a forward transform over the primes the map names (first n_groups entries), ntt.hip:
    //   1  every prime below 2^60: wide lazy ranges (policy Wide), a first-pass-only result holds values < 16q instead of < 8q
    //   2  every prime of 36 to 57 bits: no guards at all (policy Wider), a first-pass-only result holds values < 36q
    //   0  otherwise
    int ntt_wide(const cat_ctx *ctx, const LimbMap &map, uint32_t n_groups);
    struct NttFuse
    {
        const u64 *src = nullptr;
        u64 src_stride = 0;
        int src_kind = 0;
        uint32_t src_qidx = 0;
        const u64 *ep_other = nullptr;
        u64 *ep_dest = nullptr;
        u64 ep_other_b = 0, ep_other_s = 0, ep_dest_b = 0, ep_dest_s = 0;
        uint32_t ep_S = 1;
        int ep_acc = 0;
        uint32_t ep_qidx = 0;
        bool first_pass_only = false; // forward, N >= 4096: run the column pass alone (the caller finishes the rows)
        // Dual form (relinearize followed by rescale as ONE transform per limb, keyswitch.hip): the input is
        //   round(src; Q) * Q^-1 + round(src2; Q2)      (Q = prime src_qidx, Q2 = prime src2_qidx)
        // and the epilogue dest = ((other + other2 * Q2e^-1) + 4q - t) * Qe^-1 with Qe = prime ep_qidx,
        // Q2e = prime ep_qidx2.  By linearity of the transform this equals the mod-down finish followed by the
        // rescale finish modulo q, and the output is canonical.  The two-term epilogue may also come without src2,
        // over data the caller prepared itself (the BGV tail writes its two corrections with its own kernel).
        const u64 *src2 = nullptr;
        u64 src2_stride = 0;
        uint32_t src2_qidx = 0;
        const u64 *ep_other2 = nullptr;
        u64 ep_other2_b = 0, ep_other2_s = 0;
        uint32_t ep_qidx2 = 0;
    };
    // general form: objects `poly_stride` words apart; limb j uses modulus map.id[j / limb_group];
    // skip_diag skips limbs with j % limb_group == j / limb_group (key switching's I == J operands)
    cat_status launch_ntt_ex(cat_ctx *ctx, bool forward, u64 *d_data, const LimbMap &map, uint32_t n_limbs,
                             uint32_t n_polys, bool lazy, u64 poly_stride, uint32_t limb_group, bool skip_diag,
                             hipStream_t s, bool exact = false, const NttFuse *fuse = nullptr);
    cat_status launch_ew(cat_ctx *ctx, int op, const u64 *a, const u64 *b, u64 *out, const LimbMap &map,
                         uint32_t n_limbs, uint32_t n_polys, hipStream_t s);
    cat_status launch_scalar(cat_ctx *ctx, int op, const u64 *a, const u64 *scalars, u64 *out, const LimbMap &map,
                             uint32_t n_limbs, uint32_t n_polys, hipStream_t s);
    cat_status launch_ckks_tensor(cat_ctx *ctx, const u64 *x, const u64 *y, u64 *out, uint32_t L, uint32_t batch,
                                  hipStream_t s);
    cat_status launch_ckks_square(cat_ctx *ctx, const u64 *x, u64 *out, uint32_t L, uint32_t batch, hipStream_t s);
    // op: 0 = multiply, 1 = add, 2 = subtract; plain_stride in words (0 = one plaintext for the whole batch)
    cat_status launch_plain(cat_ctx *ctx, int op, const u64 *ct, const u64 *plain, size_t plain_stride, u64 *out,
                            uint32_t L, uint32_t size, uint32_t batch, hipStream_t s);
    cat_status launch_galois(cat_ctx *ctx, bool ntt_domain, const u64 *in, u64 *out, uint32_t elt, const LimbMap &map,
                             uint32_t n_limbs, uint32_t n_polys, hipStream_t s);
    cat_status launch_drop_limb(cat_ctx *ctx, const u64 *in, u64 *out, uint32_t L, uint32_t n_polys, hipStream_t s);
    // out[b][0] = in[b][0], out[b][1] = 0 for `batch` size-2 ciphertexts of L limbs
    cat_status launch_copy_c0_zero_c1(cat_ctx *ctx, const u64 *in, u64 *out, uint32_t L, uint32_t batch, hipStream_t s);
    // enc_stride / target_stride: words between consecutive ciphertexts / targets of the batch
    // rescale_out != nullptr (CKKS, N >= 4096, L >= 2): the mod-down is followed by rescale_to_next of the size-2
    // result, written to rescale_out ([batch][2][L-1][N], out_stride words apart); enc keeps only its last limb updated
    cat_status launch_switch_key(cat_ctx *ctx, u64 *enc, const u64 *target, const u64 *key, uint32_t L, uint32_t batch,
                                 size_t enc_stride, size_t target_stride, hipStream_t s, u64 *rescale_out = nullptr,
                                 size_t out_stride = 0);
    // hoisted rotations (keyswitch.hip): ct [batch][2][L][N], keys[r] prepared with sigma_g^-1, outs[r] [batch][2][L][N]
    cat_status launch_rotate_hoisted(cat_ctx *ctx, const u64 *ct, const uint32_t *elts, const u64 *const *keys, uint32_t n_rot,
                                     u64 *const *outs, uint32_t L, uint32_t batch, hipStream_t s);
    // scratch != nullptr: temporaries come from that caller-provided device block of scratch_bytes bytes instead of the
    // context workspace (the no-allocation entry point, capi.cpp); input and output must then not overlap
    cat_status launch_rescale(cat_ctx *ctx, const u64 *in, u64 *out, uint32_t L, uint32_t size, uint32_t batch,
                              size_t in_stride, size_t out_stride, hipStream_t s, u64 *scratch = nullptr,
                              size_t scratch_bytes = 0);
    enum { EW_ADD = 0, EW_SUB = 1, EW_NEG = 2, EW_DYADIC = 3, EW_MODULO = 4, SCALAR_MUL = 0, SCALAR_ADD = 1, SCALAR_SUB = 2 };
} // namespace cat

#define CAT_HIP(call)                                        \
    do                                                       \
    {                                                        \
        hipError_t e__ = (call);                             \
        if (e__ != hipSuccess)                               \
            return cat::hip_fail(e__, #call);                \
    } while (0)
