// Key replication and result gathering over RCCL: the two collectives of the path, both outside the
// steady state (keys once at start, results at the end if the caller wants them in one place).  The reference has no counterpart (single GPU, single process; no nccl/MPI anywhere under native/src).
// librccl is bound at run time (dlopen) so that single-GPU users of libcat_amd.so need no RCCL at all.
#include "context.hpp"

#include <dlfcn.h>
#include <mutex>

namespace
{
    typedef int (*bcast_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
    typedef int (*allgather_fn)(const void *, void *, size_t, int, void *, hipStream_t);
    typedef const char *(*errstr_fn)(int);
    constexpr int NCCL_UINT64 = 5; // ncclUint64, rccl.h

    std::mutex g_mutex;
    bool g_tried = false;
    bcast_fn g_bcast = nullptr;
    allgather_fn g_allgather = nullptr;
    errstr_fn g_errstr = nullptr;

    bool bind_rccl()
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        if (g_tried)
            return g_bcast != nullptr;
        g_tried = true;
        // The communicator belongs to the RCCL instance that created it: use the ncclBroadcast already visible in the
        // process (the caller's RCCL, e.g. the copy PyTorch bundles, or one the caller loaded with RTLD_GLOBAL) before
        // loading any copy of our own -- a comm must never cross two RCCL instances.
        g_bcast = (bcast_fn)dlsym(RTLD_DEFAULT, "ncclBroadcast");
        if (g_bcast)
        {
            g_allgather = (allgather_fn)dlsym(RTLD_DEFAULT, "ncclAllGather");
            g_errstr = (errstr_fn)dlsym(RTLD_DEFAULT, "ncclGetErrorString");
            return true;
        }
        void *h = nullptr;
        for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" })
        {
            // RTLD_NOLOAD first: a copy loaded without RTLD_GLOBAL is in the process but not in the default scope
            if ((h = dlopen(name, RTLD_NOW | RTLD_NOLOAD)) != nullptr)
                break;
        }
        if (!h)
            for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" })
                if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) != nullptr)
                    break;
        if (!h)
            return false;
        g_bcast = (bcast_fn)dlsym(h, "ncclBroadcast");
        g_allgather = (allgather_fn)dlsym(h, "ncclAllGather");
        g_errstr = (errstr_fn)dlsym(h, "ncclGetErrorString");
        return g_bcast != nullptr;
    }
} // namespace

extern "C" cat_status cat_key_broadcast(cat_ctx *ctx, void *comm, uint64_t *d_key, size_t n_words, int root,
                                        size_t bucket_words, cat_stream s)
{
    cat::DeviceScope dev;
    {
        const cat_status st = dev.enter(ctx);
        if (st != CAT_OK)
            return st;
    }
    if (!comm || n_words == 0)
        return CAT_OK; // one process holds the key already
    if (!d_key)
        return cat::fail(CAT_ERR_INVALID_ARG, "kswitch_keys is not valid for encryption parameters");
    if (root < 0)
        return cat::fail(CAT_ERR_INVALID_ARG, "root");
    if (!bind_rccl())
        return cat::fail(CAT_ERR_LOGIC, "librccl.so is not available");
    // one-to-all over xGMI is bound by the root's links: a few large buckets, not one call per limb
    if (bucket_words == 0)
        bucket_words = (size_t)32 << 20; // 256 MB
    for (size_t off = 0; off < n_words; off += bucket_words)
    {
        const size_t cnt = n_words - off < bucket_words ? n_words - off : bucket_words;
        const int rc = g_bcast(d_key + off, d_key + off, cnt, NCCL_UINT64, root, comm, (hipStream_t)s);
        if (rc != 0)
            return cat::fail(CAT_ERR_HIP, std::string("RCCL error: ") + (g_errstr ? g_errstr(rc) : "ncclBroadcast failed"));
    }
    return CAT_OK;
}

extern "C" cat_status cat_result_gather(cat_ctx *ctx, void *comm, const uint64_t *d_local, size_t n_words, uint64_t *d_all,
                                        cat_stream s)
{
    cat::DeviceScope dev;
    {
        const cat_status st = dev.enter(ctx);
        if (st != CAT_OK)
            return st;
    }
    if (n_words == 0)
        return CAT_OK;
    if (!d_local || !d_all)
        return cat::fail(CAT_ERR_INVALID_ARG, "destination is not valid");
    if (!comm)
    {
        // one process: its slice is the whole batch
        if (d_all != d_local)
            CAT_HIP(hipMemcpyAsync(d_all, d_local, n_words * sizeof(uint64_t), hipMemcpyDeviceToDevice, (hipStream_t)s));
        return CAT_OK;
    }
    if (!bind_rccl() || !g_allgather)
        return cat::fail(CAT_ERR_LOGIC, "librccl.so is not available");
    const int rc = g_allgather(d_local, d_all, n_words, NCCL_UINT64, comm, (hipStream_t)s);
    if (rc != 0)
        return cat::fail(CAT_ERR_HIP, std::string("RCCL error: ") + (g_errstr ? g_errstr(rc) : "ncclAllGather failed"));
    return CAT_OK;
}
