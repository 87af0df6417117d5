// C ABI of libcat_amd.so (include/cat_amd.h): argument validation in the reference's terms, context
// construction (table precompute + upload) and dispatch to the HIP kernels.
#include "context.hpp"

#include <atomic>
#include <iterator>
#include <cstring>
#include <initializer_list>
#include <mutex>
#include <new>

using namespace cat;

namespace
{
    thread_local std::string g_last_error;
}

// ---- scratch of the no-allocation entry point (cat_rescale_noalloc) ------------------------------------------
// One file-scope device array per GPU the library runs on, held until the process ends.  `cat_noalloc_scratch` is a
// host-side stand-in for the symbol, NOT a device address: host code only ever uses the per-device address that
// noalloc_scratch_init() obtained with hipGetSymbolAddress (each GPU has its own copy at its own address).
static __device__ __align__(256) unsigned char cat_noalloc_scratch[CAT_NOALLOC_SCRATCH_BYTES];

namespace
{
    constexpr int MAX_DEVICES = 64;
    std::mutex g_scratch_mutex; // serialises the one-time look-up per device; readers do not take it
    std::atomic<void *> g_scratch_ptr[MAX_DEVICES] = {};

    // once per device, with that device current (called from cat_ctx_create after hipSetDevice)
    cat_status noalloc_scratch_init(int device)
    {
        if (device < 0 || device >= MAX_DEVICES)
            return CAT_OK; // such a device simply has no no-allocation entry point (it reports that itself)
        if (g_scratch_ptr[device].load(std::memory_order_acquire))
            return CAT_OK;
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (g_scratch_ptr[device].load(std::memory_order_relaxed))
            return CAT_OK;
        void *p = nullptr;
        CAT_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(cat_noalloc_scratch)));
        g_scratch_ptr[device].store(p, std::memory_order_release);
        return CAT_OK;
    }

    void *noalloc_scratch(int device)
    {
        if (device < 0 || device >= MAX_DEVICES)
            return nullptr;
        // written once per device before any context of that device exists (cat_ctx_create): a plain acquire load
        return g_scratch_ptr[device].load(std::memory_order_acquire);
    }
}

namespace cat
{
    void set_error(const std::string &msg)
    {
        g_last_error = msg;
    }

    cat_status fail(cat_status code, const std::string &msg)
    {
        g_last_error = msg;
        return code;
    }

    cat_status hip_fail(hipError_t e, const char *what)
    {
        g_last_error = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what;
        return e == hipErrorOutOfMemory ? CAT_ERR_NOMEM : CAT_ERR_HIP;
    }

    void *workspace(cat_ctx *ctx, size_t bytes)
    {
        ctx->ws_status = CAT_OK;
        if (bytes <= ctx->ws_bytes)
            return ctx->ws;
        if (ctx->ws_strict)
        {
            // strict mode (cat_ctx_set_workspace_strict): growing means a device-wide synchronisation, hipFree and
            // hipMalloc inside an asynchronous call; the caller asked to be told instead
            set_error("the workspace would have to grow to " + std::to_string(bytes) + " bytes (cat_ctx_reserve it first)");
            ctx->ws_status = CAT_ERR_LOGIC;
            return nullptr;
        }
        ctx->ws_status = CAT_ERR_NOMEM;
        ++ctx->ws_grew;
        // earlier launches may still use the old block
        if (hipDeviceSynchronize() != hipSuccess)
        {
            set_error("HIP error while growing the workspace");
            return nullptr;
        }
        if (ctx->ws)
            (void)hipFree(ctx->ws);
        ctx->ws = nullptr;
        ctx->ws_bytes = 0;
        size_t want = bytes + bytes / 8;
        void *p = nullptr;
        if (hipMalloc(&p, want) != hipSuccess)
        {
            want = bytes;
            if (hipMalloc(&p, want) != hipSuccess)
            {
                set_error("out of device memory for the workspace");
                return nullptr;
            }
        }
        ctx->ws = p;
        ctx->ws_bytes = want;
        ctx->ws_status = CAT_OK;
        return p;
    }

    cat_status DeviceScope::enter_device(int device)
    {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess)
            return hip_fail(e, "hipGetDevice");
        if (prev != device)
        {
            e = hipSetDevice(device);
            if (e != hipSuccess)
                return hip_fail(e, "hipSetDevice");
            switched = true;
        }
        return CAT_OK;
    }
    cat_status DeviceScope::enter(const cat_ctx *ctx)
    {
        if (!ctx)
            return fail(CAT_ERR_INVALID_ARG, "ctx");
        return enter_device(ctx->device);
    }

    cat_status make_map(const cat_ctx *ctx, const uint32_t *limb_ids, uint32_t n_limbs, LimbMap &out)
    {
        if (n_limbs > CAT_MAX_MODULI)
            return fail(CAT_ERR_INVALID_ARG, "too many limbs");
        std::memset(&out, 0, sizeof(out));
        for (uint32_t j = 0; j < n_limbs; ++j)
        {
            const uint32_t id = limb_ids ? limb_ids[j] : j;
            if (id >= ctx->n_mod)
                return fail(CAT_ERR_OUT_OF_RANGE, "limb_ids");
            out.id[j] = (unsigned char)id;
        }
        return CAT_OK;
    }

    cat_status build_bfv_levels(cat_ctx *ctx); // bfv.hip
    void destroy_bfv_levels(cat_ctx *ctx);
} // namespace cat

namespace
{
    // the kernels move 16 bytes per lane (ulonglong2): every device operand must be 16-byte aligned
    cat_status aligned16(std::initializer_list<const void *> ptrs)
    {
        for (const void *p : ptrs)
            if ((reinterpret_cast<uintptr_t>(p) & 15) != 0)
                return fail(CAT_ERR_INVALID_ARG, "device pointers must be 16-byte aligned");
        return CAT_OK;
    }

    template <typename T>
    cat_status upload_vec(T **dptr, const std::vector<T> &v)
    {
        CAT_HIP(hipMalloc((void **)dptr, v.size() * sizeof(T)));
        CAT_HIP(hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        return CAT_OK;
    }
} // namespace

extern "C" {

const char *cat_version(void)
{
    return "cat_amd 0.1 (gfx950)";
}

const char *cat_last_error(void)
{
    return g_last_error.c_str();
}

cat_status cat_ctx_create(uint32_t log_n, const uint64_t *primes, uint32_t n_primes, int scheme,
                          uint64_t plain_modulus, int device, cat_ctx **out)
{
    if (!out)
        return fail(CAT_ERR_INVALID_ARG, "out");
    *out = nullptr;
    if (log_n < CAT_MIN_LOG_N || log_n > CAT_MAX_LOG_N)
        return fail(CAT_ERR_INVALID_ARG, "poly_modulus_degree is invalid");
    if (!primes || n_primes == 0 || n_primes > CAT_MAX_MODULI)
        return fail(CAT_ERR_INVALID_ARG, "coeff_modulus is invalid");
    if (scheme != CAT_SCHEME_NONE && scheme != CAT_SCHEME_BFV && scheme != CAT_SCHEME_CKKS && scheme != CAT_SCHEME_BGV)
        return fail(CAT_ERR_INVALID_ARG, "unsupported scheme");
    const size_t n = size_t(1) << log_n;
    for (uint32_t i = 0; i < n_primes; ++i)
    {
        const uint64_t q = primes[i];
        if (q < 2 || (q >> 61) != 0)
            return fail(CAT_ERR_INVALID_ARG, "value can be at most 61-bit and cannot be 1");
        // SEAL_USER_MOD_BIT_COUNT_MAX = 60 (util/defines.h:48; context.cu:202-206: invalid_coeff_modulus_bit_count).
        // It is what keeps the reference's 128-bit lazy accumulators of key switching from wrapping
        // (evaluator.cpp:2618-2671: up to 256 products of a lazy operand < 4q and a key word < q); with 61-bit primes
        // and 21 or more digits they do wrap there, and no exact implementation reproduces that.  Contexts without a
        // scheme (transforms and coefficient-wise operations only) keep the 61-bit limit of Modulus (modulus.cpp:150).
        if (scheme != CAT_SCHEME_NONE && (q >> 60) != 0)
            return fail(CAT_ERR_INVALID_ARG, "coeff_modulus primes of a scheme context can be at most 60-bit");
        if ((q - 1) % (2 * n) != 0 || !is_prime_u64(q))
            return fail(CAT_ERR_INVALID_ARG, "invalid modulus");
        for (uint32_t j = 0; j < i; ++j)
            if (primes[j] == q)
                return fail(CAT_ERR_INVALID_ARG, "coeff_modulus is invalid");
    }
    if ((scheme == CAT_SCHEME_BFV || scheme == CAT_SCHEME_BGV) && (plain_modulus < 2 || (plain_modulus >> 60) != 0))
        return fail(CAT_ERR_INVALID_ARG, "plain_modulus is invalid");
    DeviceScope dev;
    {
        cat_status st = dev.enter_device(device);
        if (st == CAT_OK)
            st = noalloc_scratch_init(device);
        if (st != CAT_OK)
            return st;
    }

    cat_ctx *ctx = new (std::nothrow) cat_ctx();
    if (!ctx)
        return fail(CAT_ERR_NOMEM, "out of host memory");
    ctx->device = device;
    ctx->log_n = log_n;
    ctx->n = n;
    ctx->k = n_primes;
    ctx->scheme = scheme;
    ctx->plain_modulus = plain_modulus;
    std::vector<uint64_t> all(primes, primes + n_primes);
    try
    {
        if (scheme == CAT_SCHEME_BFV)
        {
            // auxiliary 61-bit primes of the BEHZ bases (util/rns.cpp:625-635): m_sk, gamma, B_0 ...
            // the largest level needs |B| <= k + 1 primes
            std::vector<u64> aux = primes_descending(2 * n, 61, n_primes + 3);
            for (u64 a : aux)
                all.push_back(a);
            if (all.size() > CAT_MAX_MODULI)
                throw std::invalid_argument("coeff_modulus is invalid");
        }
        ctx->n_mod = (uint32_t)all.size();
        std::vector<DevMod> dm(ctx->n_mod);
        std::vector<MulOp> fwd(ctx->n_mod * n), inv(ctx->n_mod * n);
        // split-Shoup tables, structure of arrays per prime: [w0 | w1 | q0,q1]; the inverse one is stored shifted by
        // one entry (entry j = inv_root_powers[j + 1], ntt.hip Fast::table)
        std::vector<u64> fwd3(ctx->n_mod * 3 * n, 0), inv3(ctx->n_mod * 3 * n, 0);
        std::vector<NttMod> nm(ctx->n_mod);
        auto split3 = [](const HostMod &m, u64 w) {
            u64 t[3];
            m.split(w, t);
            return MulOp3{ t[0], t[1], (u32)t[2], (u32)(t[2] >> 32) };
        };
        for (uint32_t i = 0; i < ctx->n_mod; ++i)
        {
            HostMod hm(all[i]);
            HostNttTables t = build_ntt_tables((int)log_n, hm);
            ctx->hmods.push_back(hm);
            ctx->roots.push_back(t.root);
            dm[i].q = hm.q;
            dm[i].ratio0 = hm.ratio0;
            dm[i].ratio1 = hm.ratio1;
            dm[i].r64_quo = hm.shoup((u64)(0 - hm.ratio1 * hm.q));
            dm[i].inv_n_op = t.inv_n;
            dm[i].inv_n_quo = hm.shoup(t.inv_n);
            dm[i].inv_n_w_op = t.inv_n_w;
            dm[i].inv_n_w_quo = hm.shoup(t.inv_n_w);
            std::memcpy(&fwd[i * n], t.fwd.data(), n * sizeof(MulOp));
            std::memcpy(&inv[i * n], t.inv.data(), n * sizeof(MulOp));
            for (size_t j = 0; j < n; ++j)
            {
                u64 c[3];
                hm.split(t.fwd[2 * j], c);
                for (int e = 0; e < 3; ++e)
                    fwd3[(i * 3 + e) * n + j] = c[e];
                if (j + 1 < n)
                {
                    hm.split(t.inv[2 * (j + 1)], c);
                    for (int e = 0; e < 3; ++e)
                        inv3[(i * 3 + e) * n + j] = c[e];
                }
            }
            nm[i].q = hm.q;
            nm[i].ratio1 = hm.ratio1;
            nm[i].inv_n = MulOp{ t.inv_n, hm.shoup(t.inv_n) };
            nm[i].inv_n_w = MulOp{ t.inv_n_w, hm.shoup(t.inv_n_w) };
            nm[i].inv_n3 = split3(hm, t.inv_n);
            nm[i].inv_n_w3 = split3(hm, t.inv_n_w);
            int bits = 0;
            while (bits < 64 && (hm.q >> bits))
                ++bits;
            nm[i].red_r = 0, nm[i].red_shift = 0, nm[i].pad = 0;
            if (bits >= 36 && bits <= 60)
            {
                nm[i].red_r = (u32)((((unsigned __int128)1) << (31 + bits)) / hm.q);
                nm[i].red_shift = (u32)(bits - 33);
            }
        }
        const uint32_t k = ctx->k;
        ctx->h_inv_mod.assign((size_t)k * k, MulOp{ 0, 0 });
        ctx->h_half_mod.assign((size_t)k * k, 0);
        std::vector<MulOp> q_mod((size_t)k * k, MulOp{ 0, 0 });
        for (uint32_t a = 0; a < k; ++a)
            for (uint32_t b = 0; b < k; ++b)
            {
                if (a == b)
                    continue;
                const HostMod &mb = ctx->hmods[b];
                u64 iv = 0;
                if (!invert_mod(ctx->hmods[a].q % mb.q, mb.q, iv))
                    throw std::logic_error("invalid rns bases");
                ctx->h_inv_mod[(size_t)a * k + b] = MulOp{ iv, mb.shoup(iv) };
                ctx->h_half_mod[(size_t)a * k + b] = (ctx->hmods[a].q >> 1) % mb.q;
                q_mod[(size_t)a * k + b] = MulOp{ ctx->hmods[a].q % mb.q, mb.shoup(ctx->hmods[a].q % mb.q) };
            }
        if (scheme == CAT_SCHEME_BGV)
        {
            // RNSTool::initialize, util/rns.cpp:767-776: inv_q_last_mod_t of every level (and of the key level)
            ctx->pt = HostMod(plain_modulus);
            ctx->inv_q_mod_t.assign(k, 0);
            for (uint32_t a = 0; a < k; ++a)
                if (!invert_mod(ctx->hmods[a].q % plain_modulus, plain_modulus, ctx->inv_q_mod_t[a]))
                    throw std::logic_error("invalid rns bases");
        }
        cat_status st = upload_vec(&ctx->d_mods, dm);
        if (st == CAT_OK)
            st = upload_vec(&ctx->d_tw_fwd, fwd);
        if (st == CAT_OK)
            st = upload_vec(&ctx->d_tw_inv, inv);
        if (st == CAT_OK)
            st = upload_vec(&ctx->d_tw3_fwd, fwd3);
        if (st == CAT_OK)
            st = upload_vec(&ctx->d_tw3_inv, inv3);
        if (st == CAT_OK)
            st = upload_vec(&ctx->d_nttmods, nm);
        if (st == CAT_OK)
            st = upload_vec(&ctx->d_inv_mod, ctx->h_inv_mod);
        if (st == CAT_OK)
            st = upload_vec(&ctx->d_half_mod, ctx->h_half_mod);
        if (st == CAT_OK)
            st = upload_vec(&ctx->d_q_mod, q_mod);
        if (st == CAT_OK && scheme == CAT_SCHEME_BFV)
            st = build_bfv_levels(ctx);
        if (st != CAT_OK)
        {
            cat_ctx_destroy(ctx);
            return st;
        }
    }
    catch (const std::invalid_argument &e)
    {
        cat_ctx_destroy(ctx);
        return fail(CAT_ERR_INVALID_ARG, e.what());
    }
    catch (const std::logic_error &e)
    {
        cat_ctx_destroy(ctx);
        return fail(CAT_ERR_LOGIC, e.what());
    }
    catch (const std::bad_alloc &)
    {
        cat_ctx_destroy(ctx);
        return fail(CAT_ERR_NOMEM, "out of host memory");
    }
    *out = ctx;
    return CAT_OK;
}

cat_status cat_ctx_destroy(cat_ctx *ctx)
{
    if (!ctx)
        return CAT_OK;
    DeviceScope dev;
    (void)dev.enter(ctx);
    (void)hipDeviceSynchronize();
    destroy_bfv_levels(ctx);
    if (ctx->d_mods) (void)hipFree(ctx->d_mods);
    if (ctx->d_tw_fwd) (void)hipFree(ctx->d_tw_fwd);
    if (ctx->d_tw_inv) (void)hipFree(ctx->d_tw_inv);
    if (ctx->d_tw3_fwd) (void)hipFree(ctx->d_tw3_fwd);
    if (ctx->d_tw3_inv) (void)hipFree(ctx->d_tw3_inv);
    if (ctx->d_nttmods) (void)hipFree(ctx->d_nttmods);
    if (ctx->d_inv_mod) (void)hipFree(ctx->d_inv_mod);
    if (ctx->d_half_mod) (void)hipFree(ctx->d_half_mod);
    if (ctx->d_q_mod) (void)hipFree(ctx->d_q_mod);
    if (ctx->ws) (void)hipFree(ctx->ws);
    for (auto &kv : ctx->pool_free)
        (void)hipFree(kv.second);
    delete ctx;
    return CAT_OK;
}

cat_status cat_ctx_reserve(cat_ctx *ctx, size_t bytes)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    // reserving is the sanctioned way to grow: strict mode does not apply to it
    const bool strict = ctx->ws_strict;
    ctx->ws_strict = false;
    const uint64_t grew = ctx->ws_grew;
    void *p = workspace(ctx, bytes);
    ctx->ws_strict = strict;
    ctx->ws_grew = grew;
    return p || bytes == 0 ? CAT_OK : workspace_status(ctx);
}

cat_status cat_ctx_set_workspace_strict(cat_ctx *ctx, int on)
{
    if (!ctx)
        return fail(CAT_ERR_INVALID_ARG, "ctx");
    ctx->ws_strict = on != 0;
    return CAT_OK;
}

uint64_t cat_ctx_workspace_grew(const cat_ctx *ctx)
{
    return ctx ? ctx->ws_grew : 0;
}

size_t cat_ctx_workspace_need(const cat_ctx *ctx, uint32_t batch)
{
    // the largest request any entry point makes for `batch` ciphertexts (size <= 3) at any level of this context: the
    // maximum of the launchers' own formulas (context.hpp ws_words_*); every one of them grows with the level
    if (!ctx || batch == 0)
        return 0;
    const size_t N = ctx->n, b = batch, k = ctx->k;
    const size_t L = k > 1 ? k - 1 : 1;
    size_t words = 0;
    auto take = [&](size_t w) { words = w > words ? w : words; };
    if (k > 1 && ctx->scheme != CAT_SCHEME_NONE)
    {
        take(ws_words_galois(N, L, b)); // >= ws_words_switch_key with or without the rescale tail
        take(ws_words_rotate_hoisted(N, L, b));
    }
    take(ws_words_rescale(N, k, b * 3)); // a rescale / mod switch may start at the key level
    take(ws_words_mod_drop(N, k, b * 3));
    if (ctx->scheme == CAT_SCHEME_BFV)
        for (size_t l = 1; l < ctx->bfv.size(); ++l)
            if (ctx->bfv[l].ready)
                take(ws_words_bfv_multiply(N, l, ctx->bfv[l].Bsk, b));
    return words * sizeof(u64);
}

size_t cat_ctx_workspace_bytes(const cat_ctx *ctx)
{
    return ctx ? ctx->ws_bytes : 0;
}

uint32_t cat_ctx_log_n(const cat_ctx *ctx)
{
    return ctx ? ctx->log_n : 0;
}

uint32_t cat_ctx_n_primes(const cat_ctx *ctx)
{
    return ctx ? ctx->k : 0;
}

uint64_t cat_ctx_root(const cat_ctx *ctx, uint32_t i)
{
    return ctx && i < ctx->n_mod ? ctx->roots[i] : 0;
}

uint32_t cat_ctx_n_moduli(const cat_ctx *ctx)
{
    return ctx ? ctx->n_mod : 0;
}

uint64_t cat_ctx_modulus(const cat_ctx *ctx, uint32_t i)
{
    return ctx && i < ctx->n_mod ? ctx->hmods[i].q : 0;
}

cat_status cat_malloc(cat_ctx *ctx, size_t bytes, void **d_ptr)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    if (!d_ptr)
        return fail(CAT_ERR_INVALID_ARG, "d_ptr");
    if (!bytes)
        bytes = 8;
    if (ctx->pool_cap)
    {
        auto it = ctx->pool_free.find(bytes);
        if (it != ctx->pool_free.end())
        {
            *d_ptr = it->second;
            ctx->pool_held -= bytes;
            ctx->pool_free.erase(it);
            ctx->pool_live[*d_ptr] = bytes;
            return CAT_OK;
        }
    }
    CAT_HIP(hipMalloc(d_ptr, bytes));
    if (ctx->pool_cap)
        ctx->pool_live[*d_ptr] = bytes;
    return CAT_OK;
}

cat_status cat_free(cat_ctx *ctx, void *d_ptr)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    if (d_ptr)
    {
        auto it = ctx->pool_live.find(d_ptr);
        if (it != ctx->pool_live.end())
        {
            const size_t bytes = it->second;
            ctx->pool_live.erase(it);
            if (ctx->pool_cap && ctx->pool_held + bytes <= ctx->pool_cap)
            {
                // kept for the next cat_malloc of this size: no hipFree, hence no device-wide synchronisation; work
                // queued on the block is ordered against its next user by the stream (cat_ctx_set_pool's contract)
                ctx->pool_free.emplace(bytes, d_ptr);
                ctx->pool_held += bytes;
                return CAT_OK;
            }
        }
    }
    CAT_HIP(hipFree(d_ptr));
    return CAT_OK;
}

cat_status cat_ctx_set_pool(cat_ctx *ctx, size_t max_cached_bytes)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    ctx->pool_cap = max_cached_bytes;
    // shrinking (or switching off) releases what is held beyond the new limit
    while (ctx->pool_held > ctx->pool_cap && !ctx->pool_free.empty())
    {
        auto it = std::prev(ctx->pool_free.end());
        ctx->pool_held -= it->first;
        CAT_HIP(hipFree(it->second));
        ctx->pool_free.erase(it);
    }
    return CAT_OK;
}

cat_status cat_upload(cat_ctx *ctx, void *d_dst, const void *h_src, size_t bytes, cat_stream s)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    CAT_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)s));
    return CAT_OK;
}

cat_status cat_download(cat_ctx *ctx, void *h_dst, const void *d_src, size_t bytes, cat_stream s)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    CAT_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)s));
    return CAT_OK;
}

cat_status cat_copy(cat_ctx *ctx, void *d_dst, const void *d_src, size_t bytes, cat_stream s)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    if (bytes == 0)
        return CAT_OK;
    if (!d_dst || !d_src)
        return fail(CAT_ERR_INVALID_ARG, "operand");
    const char *a = (const char *)d_src, *b = (const char *)d_dst;
    if (!(a + bytes <= b || b + bytes <= a))
        return fail(CAT_ERR_INVALID_ARG, "destination must not alias the inputs");
    CAT_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
    return CAT_OK;
}

cat_status cat_stream_sync(cat_ctx *ctx, cat_stream s)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    CAT_HIP(hipStreamSynchronize((hipStream_t)s));
    return CAT_OK;
}

static cat_status ntt_common(cat_ctx *ctx, bool fwd, uint64_t *d_data, const uint32_t *limb_ids, uint32_t n_limbs,
                             uint32_t n_polys, int lazy, cat_stream s)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    if (!d_data && n_limbs && n_polys)
        return fail(CAT_ERR_INVALID_ARG, "operand");
    if ((st = aligned16({ d_data })) != CAT_OK)
        return st;
    LimbMap map;
    st = make_map(ctx, limb_ids, n_limbs, map);
    if (st != CAT_OK)
        return st;
    // lazy results are promised to be the reference's exact representatives -> exact arithmetic
    return launch_ntt(ctx, fwd, (u64 *)d_data, map, n_limbs, n_polys, lazy != 0, (hipStream_t)s, lazy != 0);
}

cat_status cat_ntt_forward(cat_ctx *ctx, uint64_t *d_data, const uint32_t *limb_ids, uint32_t n_limbs,
                           uint32_t n_polys, int lazy, cat_stream s)
{
    return ntt_common(ctx, true, d_data, limb_ids, n_limbs, n_polys, lazy, s);
}

cat_status cat_ntt_inverse(cat_ctx *ctx, uint64_t *d_data, const uint32_t *limb_ids, uint32_t n_limbs,
                           uint32_t n_polys, int lazy, cat_stream s)
{
    return ntt_common(ctx, false, d_data, limb_ids, n_limbs, n_polys, lazy, s);
}

static cat_status ew_common(cat_ctx *ctx, int op, const uint64_t *a, const uint64_t *b, uint64_t *out,
                            const uint32_t *limb_ids, uint32_t n_limbs, uint32_t n_polys, cat_stream s, bool binary)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    if (n_limbs && n_polys)
    {
        if (!a)
            return fail(CAT_ERR_INVALID_ARG, binary ? "operand1" : "poly");
        if (binary && !b)
            return fail(CAT_ERR_INVALID_ARG, "operand2");
        if (!out)
            return fail(CAT_ERR_INVALID_ARG, "result");
    }
    if ((st = aligned16({ a, b, out })) != CAT_OK)
        return st;
    LimbMap map;
    st = make_map(ctx, limb_ids, n_limbs, map);
    if (st != CAT_OK)
        return st;
    return launch_ew(ctx, op, (const u64 *)a, (const u64 *)b, (u64 *)out, map, n_limbs, n_polys, (hipStream_t)s);
}

cat_status cat_add(cat_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, const uint32_t *ids, uint32_t nl,
                   uint32_t np, cat_stream s)
{
    return ew_common(ctx, EW_ADD, a, b, out, ids, nl, np, s, true);
}

cat_status cat_sub(cat_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, const uint32_t *ids, uint32_t nl,
                   uint32_t np, cat_stream s)
{
    return ew_common(ctx, EW_SUB, a, b, out, ids, nl, np, s, true);
}

cat_status cat_negate(cat_ctx *ctx, const uint64_t *a, uint64_t *out, const uint32_t *ids, uint32_t nl, uint32_t np,
                      cat_stream s)
{
    return ew_common(ctx, EW_NEG, a, nullptr, out, ids, nl, np, s, false);
}

cat_status cat_dyadic(cat_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, const uint32_t *ids,
                      uint32_t nl, uint32_t np, cat_stream s)
{
    return ew_common(ctx, EW_DYADIC, a, b, out, ids, nl, np, s, true);
}

cat_status cat_modulo(cat_ctx *ctx, const uint64_t *a, uint64_t *out, const uint32_t *ids, uint32_t nl, uint32_t np,
                      cat_stream s)
{
    return ew_common(ctx, EW_MODULO, a, nullptr, out, ids, nl, np, s, false);
}

static cat_status scalar_common(cat_ctx *ctx, int op, const uint64_t *a, const uint64_t *scalars, uint64_t *out,
                                const uint32_t *ids, uint32_t nl, uint32_t np, cat_stream s)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    if (nl && np && (!a || !out))
        return fail(CAT_ERR_INVALID_ARG, a ? "result" : "poly");
    if (nl && !scalars)
        return fail(CAT_ERR_INVALID_ARG, "scalar");
    if ((st = aligned16({ a, out })) != CAT_OK)
        return st;
    LimbMap map;
    st = make_map(ctx, ids, nl, map);
    if (st != CAT_OK)
        return st;
    return launch_scalar(ctx, op, (const u64 *)a, (const u64 *)scalars, (u64 *)out, map, nl, np, (hipStream_t)s);
}

cat_status cat_mul_scalar(cat_ctx *ctx, const uint64_t *a, const uint64_t *sc, uint64_t *out, const uint32_t *ids,
                          uint32_t nl, uint32_t np, cat_stream s)
{
    return scalar_common(ctx, SCALAR_MUL, a, sc, out, ids, nl, np, s);
}

cat_status cat_add_scalar(cat_ctx *ctx, const uint64_t *a, const uint64_t *sc, uint64_t *out, const uint32_t *ids,
                          uint32_t nl, uint32_t np, cat_stream s)
{
    return scalar_common(ctx, SCALAR_ADD, a, sc, out, ids, nl, np, s);
}

cat_status cat_sub_scalar(cat_ctx *ctx, const uint64_t *a, const uint64_t *sc, uint64_t *out, const uint32_t *ids,
                          uint32_t nl, uint32_t np, cat_stream s)
{
    return scalar_common(ctx, SCALAR_SUB, a, sc, out, ids, nl, np, s);
}

static cat_status level_check(cat_ctx *ctx, uint32_t L, bool needs_next)
{
    // a level with L limbs uses primes 0..L-1; data levels have L <= k-1 when k > 1 (context.cu:607-637)
    if (L == 0 || L > ctx->k)
        return fail(CAT_ERR_INVALID_ARG, "encrypted is not valid for encryption parameters");
    if (needs_next && L < 2)
        return fail(CAT_ERR_INVALID_ARG, "end of modulus switching chain reached");
    return CAT_OK;
}

cat_status cat_ckks_multiply(cat_ctx *ctx, const uint64_t *x, const uint64_t *y, uint64_t *out, uint32_t L,
                             uint32_t batch, cat_stream s)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    if ((st = level_check(ctx, L, false)) != CAT_OK)
        return st;
    if (batch && (!x || !y || !out))
        return fail(CAT_ERR_INVALID_ARG, "encrypted1 or encrypted2 is not valid");
    if (out == x || out == y)
        return fail(CAT_ERR_INVALID_ARG, "destination must not alias the inputs");
    if ((st = aligned16({ x, y, out })) != CAT_OK)
        return st;
    return launch_ckks_tensor(ctx, (const u64 *)x, (const u64 *)y, (u64 *)out, L, batch, (hipStream_t)s);
}

cat_status cat_bgv_multiply(cat_ctx *ctx, const uint64_t *x, const uint64_t *y, uint64_t *out, uint32_t L,
                            uint32_t batch, cat_stream s)
{
    // Evaluator::bgv_multiply (evaluator.cpp:710-841) computes the same three dyadic combinations
    return cat_ckks_multiply(ctx, x, y, out, L, batch, s);
}

uint64_t cat_bgv_inv_q_last_mod_t(const cat_ctx *ctx, uint32_t L)
{
    if (!ctx || ctx->scheme != CAT_SCHEME_BGV || L == 0 || L > ctx->k)
        return 0;
    return ctx->inv_q_mod_t[L - 1];
}

cat_status cat_ckks_square(cat_ctx *ctx, const uint64_t *x, uint64_t *out, uint32_t L, uint32_t batch, cat_stream s)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    if ((st = level_check(ctx, L, false)) != CAT_OK)
        return st;
    if (batch && (!x || !out))
        return fail(CAT_ERR_INVALID_ARG, "encrypted is not valid for encryption parameters");
    if (out == x)
        return fail(CAT_ERR_INVALID_ARG, "destination must not alias the input");
    if ((st = aligned16({ x, out })) != CAT_OK)
        return st;
    return launch_ckks_square(ctx, (const u64 *)x, (u64 *)out, L, batch, (hipStream_t)s);
}

static cat_status plain_common(cat_ctx *ctx, int op, const uint64_t *ct, const uint64_t *plain, size_t plain_stride,
                               uint64_t *out, uint32_t L, uint32_t size, uint32_t batch, cat_stream s)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    if ((st = level_check(ctx, L, false)) != CAT_OK)
        return st;
    if (size * batch == 0)
        return CAT_OK;
    if (!ct || !out)
        return fail(CAT_ERR_INVALID_ARG, "encrypted is not valid for encryption parameters");
    if (!plain)
        return fail(CAT_ERR_INVALID_ARG, "plain is not valid for encryption parameters");
    if (plain_stride != 0 && (plain_stride < (size_t)L * ctx->n || (plain_stride & 1)))
        return fail(CAT_ERR_INVALID_ARG, "stride");
    if ((st = aligned16({ ct, plain, out })) != CAT_OK)
        return st;
    return launch_plain(ctx, op, (const u64 *)ct, (const u64 *)plain, plain_stride, (u64 *)out, L, size, batch,
                        (hipStream_t)s);
}

cat_status cat_multiply_plain_ntt(cat_ctx *ctx, const uint64_t *ct, const uint64_t *plain, size_t plain_stride,
                                  uint64_t *out, uint32_t L, uint32_t size, uint32_t batch, cat_stream s)
{
    return plain_common(ctx, 0, ct, plain, plain_stride, out, L, size, batch, s);
}

cat_status cat_add_plain(cat_ctx *ctx, const uint64_t *ct, const uint64_t *plain, size_t plain_stride, uint64_t *out,
                         uint32_t L, uint32_t size, uint32_t batch, cat_stream s)
{
    return plain_common(ctx, 1, ct, plain, plain_stride, out, L, size, batch, s);
}

cat_status cat_sub_plain(cat_ctx *ctx, const uint64_t *ct, const uint64_t *plain, size_t plain_stride, uint64_t *out,
                         uint32_t L, uint32_t size, uint32_t batch, cat_stream s)
{
    return plain_common(ctx, 2, ct, plain, plain_stride, out, L, size, batch, s);
}

cat_status cat_switch_key(cat_ctx *ctx, uint64_t *enc, const uint64_t *target, const uint64_t *key, uint32_t L,
                          uint32_t batch, cat_stream s)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    if (ctx->scheme == CAT_SCHEME_NONE || ctx->k < 2)
        return fail(CAT_ERR_LOGIC, "keyswitching is not supported by the context");
    if (L == 0 || L >= ctx->k)
        return fail(CAT_ERR_INVALID_ARG, "encrypted is not valid for encryption parameters");
    if (!target)
        return fail(CAT_ERR_INVALID_ARG, "target_iter");
    if (!enc)
        return fail(CAT_ERR_INVALID_ARG, "encrypted is not valid for encryption parameters");
    if (!key)
        return fail(CAT_ERR_INVALID_ARG, "kswitch_keys is not valid for encryption parameters");
    // the lazy accumulators of the inner product hold at most 64 terms without a carry (modarith.cuh: rm_mac); a context
    // has at most CAT_MAX_MODULI = 64 primes, so L <= 63 here, and the check says so explicitly
    if (L > CAT_MAX_KEYSWITCH_LIMBS)
        return fail(CAT_ERR_LOGIC, "invalid parameters");
    if ((st = aligned16({ enc, target, key })) != CAT_OK)
        return st;
    if (batch == 0)
        return CAT_OK;
    return launch_switch_key(ctx, (u64 *)enc, (const u64 *)target, (const u64 *)key, L, batch, (size_t)2 * L * ctx->n,
                             (size_t)L * ctx->n, (hipStream_t)s);
}

cat_status cat_relinearize(cat_ctx *ctx, uint64_t *ct3, const uint64_t *key, uint32_t L, uint32_t batch, cat_stream s)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    if (!key)
        return fail(CAT_ERR_INVALID_ARG, "relin_keys is not valid for encryption parameters");
    if (!ct3)
        return fail(CAT_ERR_INVALID_ARG, "encrypted is not valid for encryption parameters");
    if (ctx->scheme == CAT_SCHEME_NONE || ctx->k < 2)
        return fail(CAT_ERR_LOGIC, "keyswitching is not supported by the context");
    if (L == 0 || L >= ctx->k || L > CAT_MAX_KEYSWITCH_LIMBS)
        return fail(CAT_ERR_INVALID_ARG, "encrypted is not valid for encryption parameters");
    if ((st = aligned16({ ct3, key })) != CAT_OK)
        return st;
    if (batch == 0)
        return CAT_OK;
    // the target of ciphertext b is its own third polynomial (stride 3*L*N for both)
    const size_t stride = (size_t)3 * L * ctx->n;
    return launch_switch_key(ctx, (u64 *)ct3, (const u64 *)ct3 + (size_t)2 * L * ctx->n, (const u64 *)key, L, batch,
                             stride, stride, (hipStream_t)s);
}

cat_status cat_relinearize_rescale(cat_ctx *ctx, uint64_t *ct3, const uint64_t *key, uint64_t *out, uint32_t L,
                                   uint32_t batch, cat_stream s)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    if (!key)
        return fail(CAT_ERR_INVALID_ARG, "relin_keys is not valid for encryption parameters");
    if (!ct3 || !out)
        return fail(CAT_ERR_INVALID_ARG, "encrypted is not valid for encryption parameters");
    if (ctx->scheme == CAT_SCHEME_NONE || ctx->k < 2)
        return fail(CAT_ERR_LOGIC, "keyswitching is not supported by the context");
    if (L == 0 || L >= ctx->k || L > CAT_MAX_KEYSWITCH_LIMBS)
        return fail(CAT_ERR_INVALID_ARG, "encrypted is not valid for encryption parameters");
    if (L < 2)
        return fail(CAT_ERR_INVALID_ARG, "end of modulus switching chain reached");
    if ((st = aligned16({ ct3, key, out })) != CAT_OK)
        return st;
    if (batch == 0)
        return CAT_OK;
    const size_t N = ctx->n, stride = (size_t)3 * L * N, out_stride = (size_t)2 * (L - 1) * N;
    const u64 *in_end = (const u64 *)ct3 + (size_t)batch * stride, *out_end = (const u64 *)out + (size_t)batch * out_stride;
    if (!(out_end <= (const u64 *)ct3 || in_end <= (const u64 *)out))
        return fail(CAT_ERR_INVALID_ARG, "destination must not alias the inputs");
    if ((ctx->scheme == CAT_SCHEME_CKKS || ctx->scheme == CAT_SCHEME_BGV) && ctx->log_n >= 12)
        return launch_switch_key(ctx, (u64 *)ct3, (const u64 *)ct3 + (size_t)2 * L * N, (const u64 *)key, L, batch, stride,
                                 stride, (hipStream_t)s, (u64 *)out, out_stride);
    // BFV and small degrees: the two steps one after the other
    st = launch_switch_key(ctx, (u64 *)ct3, (const u64 *)ct3 + (size_t)2 * L * N, (const u64 *)key, L, batch, stride, stride,
                           (hipStream_t)s);
    if (st != CAT_OK)
        return st;
    return launch_rescale(ctx, (const u64 *)ct3, (u64 *)out, L, 2, batch, stride, out_stride, (hipStream_t)s);
}

cat_status cat_rescale(cat_ctx *ctx, const uint64_t *in, uint64_t *out, uint32_t L, uint32_t size, uint32_t batch,
                       cat_stream s)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    if ((st = level_check(ctx, L, true)) != CAT_OK)
        return st;
    if (size * batch && (!in || !out))
        return fail(CAT_ERR_INVALID_ARG, "encrypted is not valid for encryption parameters");
    if ((st = aligned16({ in, out })) != CAT_OK)
        return st;
    return launch_rescale(ctx, (const u64 *)in, (u64 *)out, L, size, batch, (size_t)size * L * ctx->n,
                          (size_t)size * (L - 1) * ctx->n, (hipStream_t)s);
}

cat_status cat_rescale_noalloc(cat_ctx *ctx, const uint64_t *in, uint64_t *out, uint32_t L, uint32_t size, uint32_t batch,
                               cat_stream s)
{
    // cat_rescale with its temporaries in the static device array above: no hipMalloc / hipFree / allocating
    // library routine on this path (argument checks, kernel launches and nothing else)
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    if ((st = level_check(ctx, L, true)) != CAT_OK)
        return st;
    if (size * batch && (!in || !out))
        return fail(CAT_ERR_INVALID_ARG, "encrypted is not valid for encryption parameters");
    if ((st = aligned16({ in, out })) != CAT_OK)
        return st;
    void *scratch = noalloc_scratch(ctx->device);
    if (!scratch)
        return fail(CAT_ERR_LOGIC, "no static scratch buffer on this device");
    // (polys last limbs) + (polys x (L-1) correction limbs), the same sum launch_rescale checks
    const unsigned __int128 need = (unsigned __int128)size * batch * L * ctx->n * sizeof(u64);
    if (need > CAT_NOALLOC_SCRATCH_BYTES)
        return fail(CAT_ERR_INVALID_ARG, "operands exceed the static scratch buffer");
    return launch_rescale(ctx, (const u64 *)in, (u64 *)out, L, size, batch, (size_t)size * L * ctx->n,
                          (size_t)size * (L - 1) * ctx->n, (hipStream_t)s, (u64 *)scratch, CAT_NOALLOC_SCRATCH_BYTES);
}

cat_status cat_rescale_strided(cat_ctx *ctx, const uint64_t *in, size_t in_stride, uint64_t *out, size_t out_stride,
                               uint32_t L, uint32_t size, uint32_t batch, cat_stream s)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    if ((st = level_check(ctx, L, true)) != CAT_OK)
        return st;
    if (size * batch && (!in || !out))
        return fail(CAT_ERR_INVALID_ARG, "encrypted is not valid for encryption parameters");
    if ((st = aligned16({ in, out })) != CAT_OK)
        return st;
    if (in_stride < (size_t)size * L * ctx->n || out_stride < (size_t)size * (L - 1) * ctx->n || (in_stride & 1) ||
        (out_stride & 1))
        return fail(CAT_ERR_INVALID_ARG, "stride");
    return launch_rescale(ctx, (const u64 *)in, (u64 *)out, L, size, batch, in_stride, out_stride, (hipStream_t)s);
}

cat_status cat_mod_drop(cat_ctx *ctx, const uint64_t *in, uint64_t *out, uint32_t L, uint32_t size, uint32_t batch,
                        cat_stream s)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    if ((st = level_check(ctx, L, true)) != CAT_OK)
        return st;
    const uint32_t polys = size * batch;
    if (polys && (!in || !out))
        return fail(CAT_ERR_INVALID_ARG, "encrypted is not valid for encryption parameters");
    if ((st = aligned16({ in, out })) != CAT_OK)
        return st;
    if (in == out && polys > 1)
    {
        const size_t bytes = ws_words_mod_drop(ctx->n, L, polys) * sizeof(u64);
        u64 *ws = (u64 *)workspace(ctx, bytes);
        if (!ws)
            return workspace_status(ctx);
        st = launch_drop_limb(ctx, (const u64 *)in, ws, L, polys, (hipStream_t)s);
        if (st != CAT_OK)
            return st;
        CAT_HIP(hipMemcpyAsync(out, ws, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
        return CAT_OK;
    }
    if (in == out)
        return CAT_OK;
    return launch_drop_limb(ctx, (const u64 *)in, (u64 *)out, L, polys, (hipStream_t)s);
}

uint32_t cat_galois_elt_from_step(uint32_t log_n, int step)
{
    const uint32_t n = 1u << log_n;
    const uint64_t m = 2ull * n;
    if (step == 0)
        return (uint32_t)(m - 1);
    const bool neg = step < 0;
    const uint32_t pos = (uint32_t)(neg ? -(int64_t)step : step);
    if (pos >= (n >> 1))
        return 0;
    uint32_t e = neg ? (n >> 1) - pos : pos;
    uint64_t elt = 1;
    while (e--)
        elt = (elt * 3) & (m - 1);
    return (uint32_t)elt;
}

static cat_status galois_check(cat_ctx *ctx, const uint64_t *in, uint64_t *out, uint32_t elt)
{
    if (!in)
        return fail(CAT_ERR_INVALID_ARG, "operand");
    if (!out)
        return fail(CAT_ERR_INVALID_ARG, "result");
    if (in == out)
        return fail(CAT_ERR_INVALID_ARG, "result cannot point to the same value as operand");
    if (aligned16({ in, out }) != CAT_OK)
        return CAT_ERR_INVALID_ARG;
    if (!(elt & 1) || elt >= 2 * (uint64_t)ctx->n)
        return fail(CAT_ERR_INVALID_ARG, "Galois element is not valid");
    return CAT_OK;
}

cat_status cat_apply_galois_ntt(cat_ctx *ctx, const uint64_t *in, uint64_t *out, uint32_t elt, uint32_t nl,
                                uint32_t np, cat_stream s)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    if ((st = galois_check(ctx, in, out, elt)) != CAT_OK)
        return st;
    return launch_galois(ctx, true, (const u64 *)in, (u64 *)out, elt, ctx->identity_map(), nl ? nl : 1, nl ? np : 0,
                         (hipStream_t)s);
}

cat_status cat_apply_galois(cat_ctx *ctx, const uint64_t *in, uint64_t *out, uint32_t elt, const uint32_t *ids,
                            uint32_t nl, uint32_t np, cat_stream s)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    if ((st = galois_check(ctx, in, out, elt)) != CAT_OK)
        return st;
    LimbMap map;
    st = make_map(ctx, ids, nl, map);
    if (st != CAT_OK)
        return st;
    return launch_galois(ctx, false, (const u64 *)in, (u64 *)out, elt, map, nl ? nl : 1, nl ? np : 0, (hipStream_t)s);
}

static cat_status galois_inplace(cat_ctx *ctx, uint64_t *ct, uint32_t elt, const uint64_t *key, uint32_t L, uint32_t batch,
                                 cat_stream s, uint64_t *rescale_out);

// g^-1 modulo 2N for an odd g (the Galois group of the 2N-th cyclotomic ring)
static uint32_t galois_inverse(uint32_t elt, uint32_t log_n)
{
    const uint64_t m = 2ull << log_n;
    uint64_t inv = 1, base = elt % m;
    // the group has exponent N: g^(N-1) = g^-1
    for (uint64_t e = (m >> 1) - 1; e; e >>= 1)
    {
        if (e & 1)
            inv = inv * base % m;
        base = base * base % m;
    }
    return (uint32_t)inv;
}

cat_status cat_hoisted_key_prepare(cat_ctx *ctx, const uint64_t *key, uint32_t elt, uint64_t *key_out, cat_stream s)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    if (ctx->scheme == CAT_SCHEME_NONE || ctx->k < 2)
        return fail(CAT_ERR_LOGIC, "keyswitching is not supported by the context");
    if (!key || !key_out)
        return fail(CAT_ERR_INVALID_ARG, "Galois key not present");
    if (key == key_out)
        return fail(CAT_ERR_INVALID_ARG, "result cannot point to the same value as operand");
    if ((st = aligned16({ key, key_out })) != CAT_OK)
        return st;
    if (!(elt & 1) || elt >= 2 * (uint64_t)ctx->n)
        return fail(CAT_ERR_INVALID_ARG, "Galois element is not valid");
    // every limb of the key is in NTT form (kswitchkeys.h:90-101): [k-1][2][k] limbs
    return launch_galois(ctx, true, (const u64 *)key, (u64 *)key_out, galois_inverse(elt, ctx->log_n), ctx->identity_map(),
                         ctx->k, (ctx->k - 1) * 2, (hipStream_t)s);
}

cat_status cat_rotate_hoisted(cat_ctx *ctx, const uint64_t *ct, const uint32_t *elts, const uint64_t *const *keys,
                              uint32_t n_rot, uint64_t *const *outs, uint32_t L, uint32_t batch, cat_stream s)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    if (ctx->scheme == CAT_SCHEME_NONE || ctx->k < 2)
        return fail(CAT_ERR_LOGIC, "keyswitching is not supported by the context");
    if (!ct || L == 0 || L >= ctx->k || L > CAT_MAX_KEYSWITCH_LIMBS)
        return fail(CAT_ERR_INVALID_ARG, "encrypted is not valid for encryption parameters");
    if (n_rot && (!elts || !keys))
        return fail(CAT_ERR_INVALID_ARG, "Galois key not present");
    if (n_rot && !outs)
        return fail(CAT_ERR_INVALID_ARG, "destination is not valid");
    if ((st = aligned16({ ct })) != CAT_OK)
        return st;
    const size_t ctw = (size_t)2 * L * ctx->n;
    const u64 *in_end = (const u64 *)ct + (size_t)batch * ctw;
    for (uint32_t r = 0; r < n_rot; ++r)
    {
        if (!outs[r])
            return fail(CAT_ERR_INVALID_ARG, "destination is not valid");
        if ((st = aligned16({ outs[r] })) != CAT_OK)
            return st;
        if (!((const u64 *)outs[r] + (size_t)batch * ctw <= (const u64 *)ct || in_end <= (const u64 *)outs[r]))
            return fail(CAT_ERR_INVALID_ARG, "destination must not alias the inputs");
        if (!keys[r])
            return fail(CAT_ERR_INVALID_ARG, "Galois key not present");
        if ((st = aligned16({ keys[r] })) != CAT_OK)
            return st;
        if (!(elts[r] & 1) || elts[r] >= 2 * (uint64_t)ctx->n)
            return fail(CAT_ERR_INVALID_ARG, "Galois element is not valid");
    }
    if (batch == 0 || n_rot == 0)
        return CAT_OK;
    return launch_rotate_hoisted(ctx, (const u64 *)ct, elts, (const u64 *const *)keys, n_rot, (u64 *const *)outs, L, batch,
                                 (hipStream_t)s);
}

cat_status cat_apply_galois_inplace(cat_ctx *ctx, uint64_t *ct, uint32_t elt, const uint64_t *key, uint32_t L,
                                    uint32_t batch, cat_stream s)
{
    return galois_inplace(ctx, ct, elt, key, L, batch, s, nullptr);
}

cat_status cat_apply_galois_rescale(cat_ctx *ctx, uint64_t *ct, uint32_t elt, const uint64_t *key, uint64_t *out, uint32_t L,
                                    uint32_t batch, cat_stream s)
{
    if (!out)
        return fail(CAT_ERR_INVALID_ARG, "destination is not valid");
    if (ctx && L < 2)
        return fail(CAT_ERR_INVALID_ARG, "end of modulus switching chain reached");
    if (ctx && batch)
    {
        const u64 *in_end = (const u64 *)ct + (size_t)batch * 2 * L * ctx->n,
                  *out_end = (const u64 *)out + (size_t)batch * 2 * (L - 1) * ctx->n;
        if (ct && !(out_end <= (const u64 *)ct || in_end <= (const u64 *)out))
            return fail(CAT_ERR_INVALID_ARG, "destination must not alias the inputs");
    }
    return galois_inplace(ctx, ct, elt, key, L, batch, s, out);
}

static cat_status galois_inplace(cat_ctx *ctx, uint64_t *ct, uint32_t elt, const uint64_t *key, uint32_t L, uint32_t batch,
                                 cat_stream s, uint64_t *rescale_out)
{
    DeviceScope dev;
    cat_status st = dev.enter(ctx);
    if (st != CAT_OK)
        return st;
    if (ctx->scheme == CAT_SCHEME_NONE || ctx->k < 2)
        return fail(CAT_ERR_LOGIC, "keyswitching is not supported by the context");
    if (!ct || L == 0 || L >= ctx->k)
        return fail(CAT_ERR_INVALID_ARG, "encrypted is not valid for encryption parameters");
    if (!key)
        return fail(CAT_ERR_INVALID_ARG, "Galois key not present");
    if ((st = aligned16({ ct, key, rescale_out })) != CAT_OK)
        return st;
    if (!(elt & 1) || elt >= 2 * (uint64_t)ctx->n)
        return fail(CAT_ERR_INVALID_ARG, "Galois element is not valid");
    if (batch == 0)
        return CAT_OK;
    // perm = [batch][2][L][N] permuted copy of the ciphertexts; c0 <- perm(c0), c1 <- 0 and the key switch reads its
    // target perm(c1) in place, one ciphertext stride apart (:2410-2449).  The permuted copy must survive
    // launch_switch_key's own workspace use, so it is carved from the same block, after the key-switch region.
    const size_t N = ctx->n, ctw = (size_t)2 * L * N;
    const size_t ks_words = ws_words_switch_key(N, L, batch, true);
    u64 *ws = (u64 *)workspace(ctx, ws_words_galois(N, L, batch) * sizeof(u64));
    if (!ws)
        return workspace_status(ctx);
    u64 *perm = ws + ks_words;
    const u64 *tgt = perm + (size_t)L * N;
    const bool ntt = ctx->scheme == CAT_SCHEME_CKKS || ctx->scheme == CAT_SCHEME_BGV; // :2429
    st = launch_galois(ctx, ntt, (const u64 *)ct, perm, elt, ctx->identity_map(), L, batch * 2, (hipStream_t)s);
    if (st != CAT_OK)
        return st;
    st = launch_copy_c0_zero_c1(ctx, perm, (u64 *)ct, L, batch, (hipStream_t)s);
    if (st != CAT_OK)
        return st;
    if (!rescale_out)
        return launch_switch_key(ctx, (u64 *)ct, tgt, (const u64 *)key, L, batch, ctw, ctw, (hipStream_t)s);
    if ((ctx->scheme == CAT_SCHEME_CKKS || ctx->scheme == CAT_SCHEME_BGV) && ctx->log_n >= 12)
        return launch_switch_key(ctx, (u64 *)ct, tgt, (const u64 *)key, L, batch, ctw, ctw, (hipStream_t)s,
                                 (u64 *)rescale_out, (size_t)2 * (L - 1) * N);
    st = launch_switch_key(ctx, (u64 *)ct, tgt, (const u64 *)key, L, batch, ctw, ctw, (hipStream_t)s);
    if (st != CAT_OK)
        return st;
    return launch_rescale(ctx, (const u64 *)ct, (u64 *)rescale_out, L, 2, batch, ctw, (size_t)2 * (L - 1) * N, (hipStream_t)s);
}

} // extern "C"
