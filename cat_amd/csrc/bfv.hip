// BFV BEHZ multiplication (Evaluator::bfv_multiply, evaluator.cpp:395-567; util/rns.cpp:418-463, 903-1131).
#include "context.hpp"

namespace cat
{
    cat_status build_bfv_levels(cat_ctx *ctx)
    {
        ctx->bfv.assign(ctx->k + 1, BfvLevel{});
        return CAT_OK;
    }

    void destroy_bfv_levels(cat_ctx *ctx)
    {
        for (auto &l : ctx->bfv)
            if (l.d_blob)
                (void)hipFree(l.d_blob);
        ctx->bfv.clear();
    }
} // namespace cat

extern "C" {
uint32_t cat_bfv_base_Bsk_size(const cat_ctx *, uint32_t) { return 0; }
cat_status cat_bfv_base_Bsk(const cat_ctx *, uint32_t, uint64_t *) { return cat::fail(CAT_ERR_LOGIC, "unsupported scheme"); }
cat_status cat_bfv_multiply(cat_ctx *, const uint64_t *, const uint64_t *, uint64_t *, uint32_t, uint32_t, cat_stream) { return cat::fail(CAT_ERR_LOGIC, "unsupported scheme"); }
cat_status cat_fastbconv_m_tilde(cat_ctx *, const uint64_t *, uint64_t *, uint32_t, uint32_t, cat_stream) { return cat::fail(CAT_ERR_LOGIC, "unsupported scheme"); }
cat_status cat_sm_mrq(cat_ctx *, const uint64_t *, uint64_t *, uint32_t, uint32_t, cat_stream) { return cat::fail(CAT_ERR_LOGIC, "unsupported scheme"); }
cat_status cat_fast_floor(cat_ctx *, const uint64_t *, uint64_t *, uint32_t, uint32_t, cat_stream) { return cat::fail(CAT_ERR_LOGIC, "unsupported scheme"); }
cat_status cat_fastbconv_sk(cat_ctx *, const uint64_t *, uint64_t *, uint32_t, uint32_t, cat_stream) { return cat::fail(CAT_ERR_LOGIC, "unsupported scheme"); }
}
