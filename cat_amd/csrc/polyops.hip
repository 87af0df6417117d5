// Coefficient-wise kernels: add/sub/negate/dyadic/scalar ops (util/polyarithsmallmod.cpp:18-285),
// the fused CKKS 2x2->3 tensor product (evaluator.cpp:604-663), Galois permutations
// (util/galois.cpp:148-218) and limb-dropping copies (evaluator.cpp:1282-1345).
// All are pure streaming kernels: 16-byte accesses per lane, grid-stride, HBM-bound.
#include "context.hpp"

namespace
{
    struct EwArgs
    {
        const u64 *a;
        const u64 *b;
        u64 *out;
        const DevMod *mods;
        LimbMap map;
        u32 n_limbs;
        u32 log_n;
        u64 total2; // number of 2-word vectors
    };

    struct ScalarArgs
    {
        u64 op[64];
        u64 quo[64];
    };

    enum EwOp { OP_ADD, OP_SUB, OP_NEG, OP_DYADIC, OP_MODULO };

    template <int OP>
    __device__ __forceinline__ u64 ew_apply(u64 x, u64 y, const DevMod &m)
    {
        if (OP == OP_ADD) return add_mod(x, y, m.q);
        if (OP == OP_SUB) return sub_mod(x, y, m.q);
        if (OP == OP_NEG) return neg_mod(x, m.q);
        if (OP == OP_DYADIC) return mul_mod(x, y, m.q, m.ratio0, m.ratio1);
        return barrett64(x, m.q, m.ratio1);
    }

    template <int OP>
    __global__ __launch_bounds__(256) void ew_kernel(EwArgs g)
    {
        const u64 stride = (u64)gridDim.x * blockDim.x;
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < g.total2; i += stride)
        {
            const u32 limb = (u32)(((i * 2) >> g.log_n) % g.n_limbs);
            const DevMod m = g.mods[g.map.id[limb]];
            const ulonglong2 x = reinterpret_cast<const ulonglong2 *>(g.a)[i];
            ulonglong2 y = x;
            if (OP == OP_ADD || OP == OP_SUB || OP == OP_DYADIC)
                y = reinterpret_cast<const ulonglong2 *>(g.b)[i];
            ulonglong2 r;
            r.x = ew_apply<OP>(x.x, y.x, m);
            r.y = ew_apply<OP>(x.y, y.y, m);
            reinterpret_cast<ulonglong2 *>(g.out)[i] = r;
        }
    }

    enum ScOp { SC_MUL, SC_ADD, SC_SUB };

    template <int OP>
    __global__ __launch_bounds__(256) void scalar_kernel(EwArgs g, ScalarArgs sc)
    {
        const u64 stride = (u64)gridDim.x * blockDim.x;
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < g.total2; i += stride)
        {
            const u32 limb = (u32)(((i * 2) >> g.log_n) % g.n_limbs);
            const u64 q = g.mods[g.map.id[limb]].q;
            const u64 so = sc.op[limb], sq = sc.quo[limb];
            const ulonglong2 x = reinterpret_cast<const ulonglong2 *>(g.a)[i];
            ulonglong2 r;
            if (OP == SC_MUL)
            {
                r.x = mul_shoup(x.x, so, sq, q);
                r.y = mul_shoup(x.y, so, sq, q);
            }
            else if (OP == SC_ADD)
            {
                r.x = add_mod(x.x, so, q);
                r.y = add_mod(x.y, so, q);
            }
            else
            {
                r.x = sub_mod(x.x, so, q);
                r.y = sub_mod(x.y, so, q);
            }
            reinterpret_cast<ulonglong2 *>(g.out)[i] = r;
        }
    }

    // (x0,x1) (x) (y0,y1) -> (x0y0, x0y1 + x1y0, x1y1), one pass: 4 reads + 3 writes per coefficient.
    struct TensorArgs
    {
        const u64 *x;
        const u64 *y;
        u64 *out;
        const DevMod *mods;
        u32 L;
        u32 log_n;
        u64 poly2;  // L*N/2 vectors per polynomial
        u64 total2; // batch * poly2
    };

    __global__ __launch_bounds__(256) void ckks_tensor_kernel(TensorArgs g)
    {
        const u64 stride = (u64)gridDim.x * blockDim.x;
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < g.total2; i += stride)
        {
            const u64 ct = i / g.poly2, off = i % g.poly2;
            const u32 limb = (u32)((off * 2) >> g.log_n);
            const DevMod m = g.mods[limb];
            const ulonglong2 *x = reinterpret_cast<const ulonglong2 *>(g.x) + ct * 2 * g.poly2 + off;
            const ulonglong2 *y = reinterpret_cast<const ulonglong2 *>(g.y) + ct * 2 * g.poly2 + off;
            ulonglong2 *o = reinterpret_cast<ulonglong2 *>(g.out) + ct * 3 * g.poly2 + off;
            const ulonglong2 x0 = x[0], x1 = x[g.poly2], y0 = y[0], y1 = y[g.poly2];
            ulonglong2 r0, r1, r2;
            r0.x = mul_mod(x0.x, y0.x, m.q, m.ratio0, m.ratio1);
            r0.y = mul_mod(x0.y, y0.y, m.q, m.ratio0, m.ratio1);
            r2.x = mul_mod(x1.x, y1.x, m.q, m.ratio0, m.ratio1);
            r2.y = mul_mod(x1.y, y1.y, m.q, m.ratio0, m.ratio1);
            r1.x = add_mod(mul_mod(x0.x, y1.x, m.q, m.ratio0, m.ratio1), mul_mod(x1.x, y0.x, m.q, m.ratio0, m.ratio1), m.q);
            r1.y = add_mod(mul_mod(x0.y, y1.y, m.q, m.ratio0, m.ratio1), mul_mod(x1.y, y0.y, m.q, m.ratio0, m.ratio1), m.q);
            o[0] = r0;
            o[g.poly2] = r1;
            o[2 * g.poly2] = r2;
        }
    }

    // ckks_square (evaluator.cpp:1022-1078), size 2 -> 3: (c0^2, 2 c0 c1, c1^2); 2 reads + 3 writes per coefficient
    __global__ __launch_bounds__(256) void ckks_square_kernel(TensorArgs g)
    {
        const u64 stride = (u64)gridDim.x * blockDim.x;
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < g.total2; i += stride)
        {
            const u64 ct = i / g.poly2, off = i % g.poly2;
            const u32 limb = (u32)((off * 2) >> g.log_n);
            const DevMod m = g.mods[limb];
            const ulonglong2 *x = reinterpret_cast<const ulonglong2 *>(g.x) + ct * 2 * g.poly2 + off;
            ulonglong2 *o = reinterpret_cast<ulonglong2 *>(g.out) + ct * 3 * g.poly2 + off;
            const ulonglong2 x0 = x[0], x1 = x[g.poly2];
            ulonglong2 r0, r1, r2;
            r0.x = mul_mod(x0.x, x0.x, m.q, m.ratio0, m.ratio1);
            r0.y = mul_mod(x0.y, x0.y, m.q, m.ratio0, m.ratio1);
            r2.x = mul_mod(x1.x, x1.x, m.q, m.ratio0, m.ratio1);
            r2.y = mul_mod(x1.y, x1.y, m.q, m.ratio0, m.ratio1);
            r1.x = mul_mod(x0.x, x1.x, m.q, m.ratio0, m.ratio1);
            r1.y = mul_mod(x0.y, x1.y, m.q, m.ratio0, m.ratio1);
            r1.x = add_mod(r1.x, r1.x, m.q);
            r1.y = add_mod(r1.y, r1.y, m.q);
            o[0] = r0;
            o[g.poly2] = r1;
            o[2 * g.poly2] = r2;
        }
    }

    // Plaintext operand ops on batch x size x L x N ciphertexts; the plaintext is L x N (stride 0 = shared by the
    // batch).  MUL: every polynomial times plain (multiply_plain_ntt, evaluator.cpp:2135-2172); ADD / SUB: c0 +- plain,
    // the other polynomials copied (add_plain / sub_plain, CKKS branch, evaluator.cpp:1737-1951).
    struct PlainArgs
    {
        const u64 *ct;
        const u64 *plain;
        u64 *out;
        const DevMod *mods;
        u32 L, log_n, size;
        u64 plain_stride2; // vectors between consecutive plaintexts (0 = one plaintext for the whole batch)
        u64 total2;        // batch * size * L * N / 2
    };

    enum PlainOp { PL_MUL, PL_ADD, PL_SUB };

    template <int OP>
    __global__ __launch_bounds__(256) void plain_kernel(PlainArgs g)
    {
        const u64 stride = (u64)gridDim.x * blockDim.x;
        const u64 poly2 = ((u64)g.L << g.log_n) >> 1;
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < g.total2; i += stride)
        {
            const u64 pidx = i / poly2, off = i % poly2;
            const u64 b = pidx / g.size;
            const u32 poly = (u32)(pidx % g.size);
            const DevMod m = g.mods[(off * 2) >> g.log_n];
            ulonglong2 x = reinterpret_cast<const ulonglong2 *>(g.ct)[i];
            if (OP == PL_MUL || poly == 0)
            {
                const ulonglong2 p = reinterpret_cast<const ulonglong2 *>(g.plain)[b * g.plain_stride2 + off];
                if (OP == PL_MUL)
                {
                    x.x = mul_mod(x.x, p.x, m.q, m.ratio0, m.ratio1);
                    x.y = mul_mod(x.y, p.y, m.q, m.ratio0, m.ratio1);
                }
                else if (OP == PL_ADD)
                {
                    x.x = add_mod(x.x, p.x, m.q);
                    x.y = add_mod(x.y, p.y, m.q);
                }
                else
                {
                    x.x = sub_mod(x.x, p.x, m.q);
                    x.y = sub_mod(x.y, p.y, m.q);
                }
            }
            reinterpret_cast<ulonglong2 *>(g.out)[i] = x;
        }
    }

    // Galois: out[i] = in[bitrev(((elt * bitrev_{logn+1}(N + i)) >> 1) & (N-1))]  (galois.cpp:36-43, 217)
    struct GaloisArgs
    {
        const u64 *in;
        u64 *out;
        const DevMod *mods;
        LimbMap map;
        u32 n_limbs;
        u32 log_n;
        u32 elt;
        u64 total;
    };

    __global__ __launch_bounds__(256) void galois_ntt_kernel(GaloisArgs g)
    {
        const u64 stride = (u64)gridDim.x * blockDim.x;
        const u32 nm1 = (1u << g.log_n) - 1;
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < g.total; i += stride)
        {
            const u32 idx = (u32)i & nm1;
            const u64 base = i - idx;
            const u32 reversed = __brev(idx + nm1 + 1) >> (31 - g.log_n);
            const u32 raw = (u32)(((u64)g.elt * reversed) >> 1) & nm1;
            const u32 src = __brev(raw) >> (32 - g.log_n);
            g.out[i] = g.in[base + src];
        }
    }

    // coefficient domain (galois.cpp:174-189): out[(i*elt) mod N] = +-in[i]
    __global__ __launch_bounds__(256) void galois_coeff_kernel(GaloisArgs g)
    {
        const u64 stride = (u64)gridDim.x * blockDim.x;
        const u32 nm1 = (1u << g.log_n) - 1;
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < g.total; i += stride)
        {
            const u32 idx = (u32)i & nm1;
            const u64 base = i - idx;
            const u32 limb = (u32)((i >> g.log_n) % g.n_limbs);
            const u64 q = g.mods[g.map.id[limb]].q;
            const u64 raw = (u64)idx * g.elt;
            u64 v = g.in[i];
            if ((raw >> g.log_n) & 1)
                v = neg_mod(v, q);
            g.out[base + (raw & nm1)] = v;
        }
    }

    // copy the first (L-1) limbs of every polynomial (stride compaction)
    __global__ __launch_bounds__(256) void drop_limb_kernel(const u64 *in, u64 *out, u32 L, u32 log_n, u64 total2)
    {
        const u64 stride = (u64)gridDim.x * blockDim.x;
        const u64 out_poly2 = ((u64)(L - 1) << log_n) >> 1, in_poly2 = ((u64)L << log_n) >> 1;
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < total2; i += stride)
        {
            const u64 p = i / out_poly2, off = i % out_poly2;
            reinterpret_cast<ulonglong2 *>(out)[i] = reinterpret_cast<const ulonglong2 *>(in)[p * in_poly2 + off];
        }
    }

    // apply_galois_inplace (evaluator.cpp:2410-2449): out[b][0] = in[b][0], out[b][1] = 0 for ciphertexts of 2 x words2
    // 16-byte pieces each, in one pass
    __global__ __launch_bounds__(256) void copy_c0_zero_c1_kernel(const u64 *in, u64 *out, u64 words2, u64 total2)
    {
        const u64 stride = (u64)gridDim.x * blockDim.x;
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < total2; i += stride)
        {
            const u64 b = i / words2, off = i - b * words2;
            const ulonglong2 v = reinterpret_cast<const ulonglong2 *>(in)[b * 2 * words2 + off];
            ulonglong2 z;
            z.x = z.y = 0;
            reinterpret_cast<ulonglong2 *>(out)[b * 2 * words2 + off] = v;
            reinterpret_cast<ulonglong2 *>(out)[(b * 2 + 1) * words2 + off] = z;
        }
    }

    u32 grid_for(u64 items)
    {
        u64 blocks = (items + 255) / 256;
        if (blocks > 256 * 16)
            blocks = 256 * 16;
        if (blocks == 0)
            blocks = 1;
        return (u32)blocks;
    }
} // namespace

namespace cat
{
    cat_status launch_ew(cat_ctx *ctx, int op, const u64 *a, const u64 *b, u64 *out, const LimbMap &map,
                         uint32_t n_limbs, uint32_t n_polys, hipStream_t s)
    {
        EwArgs g;
        g.a = a;
        g.b = b;
        g.out = out;
        g.mods = ctx->d_mods;
        g.map = map;
        g.n_limbs = n_limbs;
        g.log_n = ctx->log_n;
        g.total2 = ((u64)n_polys * n_limbs << ctx->log_n) >> 1;
        if (g.total2 == 0)
            return CAT_OK;
        const u32 grid = grid_for(g.total2);
        switch (op)
        {
        case OP_ADD: hipLaunchKernelGGL(ew_kernel<OP_ADD>, dim3(grid), dim3(256), 0, s, g); break;
        case OP_SUB: hipLaunchKernelGGL(ew_kernel<OP_SUB>, dim3(grid), dim3(256), 0, s, g); break;
        case OP_NEG: hipLaunchKernelGGL(ew_kernel<OP_NEG>, dim3(grid), dim3(256), 0, s, g); break;
        case OP_DYADIC: hipLaunchKernelGGL(ew_kernel<OP_DYADIC>, dim3(grid), dim3(256), 0, s, g); break;
        default: hipLaunchKernelGGL(ew_kernel<OP_MODULO>, dim3(grid), dim3(256), 0, s, g); break;
        }
        CAT_HIP(hipGetLastError());
        return CAT_OK;
    }

    cat_status launch_scalar(cat_ctx *ctx, int op, const u64 *a, const u64 *scalars, u64 *out, const LimbMap &map,
                             uint32_t n_limbs, uint32_t n_polys, hipStream_t s)
    {
        EwArgs g;
        g.a = a;
        g.b = nullptr;
        g.out = out;
        g.mods = ctx->d_mods;
        g.map = map;
        g.n_limbs = n_limbs;
        g.log_n = ctx->log_n;
        g.total2 = ((u64)n_polys * n_limbs << ctx->log_n) >> 1;
        if (g.total2 == 0)
            return CAT_OK;
        ScalarArgs sc{};
        for (uint32_t j = 0; j < n_limbs; ++j)
        {
            const HostMod &m = ctx->hmods[map.id[j]];
            if (op == SC_MUL)
            {
                sc.op[j] = scalars[j] % m.q; // scalar is first reduced modulo q (polyarithsmallmod.cuh:477-486)
                sc.quo[j] = m.shoup(sc.op[j]);
            }
            else
            {
                if (scalars[j] >= m.q)
                    return fail(CAT_ERR_INVALID_ARG, "scalar");
                sc.op[j] = scalars[j];
            }
        }
        const u32 grid = grid_for(g.total2);
        switch (op)
        {
        case SC_MUL: hipLaunchKernelGGL(scalar_kernel<SC_MUL>, dim3(grid), dim3(256), 0, s, g, sc); break;
        case SC_ADD: hipLaunchKernelGGL(scalar_kernel<SC_ADD>, dim3(grid), dim3(256), 0, s, g, sc); break;
        default: hipLaunchKernelGGL(scalar_kernel<SC_SUB>, dim3(grid), dim3(256), 0, s, g, sc); break;
        }
        CAT_HIP(hipGetLastError());
        return CAT_OK;
    }

    cat_status launch_ckks_tensor(cat_ctx *ctx, const u64 *x, const u64 *y, u64 *out, uint32_t L, uint32_t batch,
                                  hipStream_t s)
    {
        TensorArgs g;
        g.x = x;
        g.y = y;
        g.out = out;
        g.mods = ctx->d_mods;
        g.L = L;
        g.log_n = ctx->log_n;
        g.poly2 = ((u64)L << ctx->log_n) >> 1;
        g.total2 = g.poly2 * batch;
        if (g.total2 == 0)
            return CAT_OK;
        hipLaunchKernelGGL(ckks_tensor_kernel, dim3(grid_for(g.total2)), dim3(256), 0, s, g);
        CAT_HIP(hipGetLastError());
        return CAT_OK;
    }

    cat_status launch_ckks_square(cat_ctx *ctx, const u64 *x, u64 *out, uint32_t L, uint32_t batch, hipStream_t s)
    {
        TensorArgs g;
        g.x = x;
        g.y = x;
        g.out = out;
        g.mods = ctx->d_mods;
        g.L = L;
        g.log_n = ctx->log_n;
        g.poly2 = ((u64)L << ctx->log_n) >> 1;
        g.total2 = g.poly2 * batch;
        if (g.total2 == 0)
            return CAT_OK;
        hipLaunchKernelGGL(ckks_square_kernel, dim3(grid_for(g.total2)), dim3(256), 0, s, g);
        CAT_HIP(hipGetLastError());
        return CAT_OK;
    }

    cat_status launch_plain(cat_ctx *ctx, int op, const u64 *ct, const u64 *plain, size_t plain_stride, u64 *out,
                            uint32_t L, uint32_t size, uint32_t batch, hipStream_t s)
    {
        PlainArgs g;
        g.ct = ct;
        g.plain = plain;
        g.out = out;
        g.mods = ctx->d_mods;
        g.L = L;
        g.log_n = ctx->log_n;
        g.size = size;
        g.plain_stride2 = plain_stride / 2;
        g.total2 = (((u64)batch * size * L) << ctx->log_n) >> 1;
        if (g.total2 == 0)
            return CAT_OK;
        const u32 grid = grid_for(g.total2);
        switch (op)
        {
        case PL_MUL: hipLaunchKernelGGL(plain_kernel<PL_MUL>, dim3(grid), dim3(256), 0, s, g); break;
        case PL_ADD: hipLaunchKernelGGL(plain_kernel<PL_ADD>, dim3(grid), dim3(256), 0, s, g); break;
        default: hipLaunchKernelGGL(plain_kernel<PL_SUB>, dim3(grid), dim3(256), 0, s, g); break;
        }
        CAT_HIP(hipGetLastError());
        return CAT_OK;
    }

    cat_status launch_galois(cat_ctx *ctx, bool ntt_domain, const u64 *in, u64 *out, uint32_t elt, const LimbMap &map,
                             uint32_t n_limbs, uint32_t n_polys, hipStream_t s)
    {
        GaloisArgs g;
        g.in = in;
        g.out = out;
        g.mods = ctx->d_mods;
        g.map = map;
        g.n_limbs = n_limbs;
        g.log_n = ctx->log_n;
        g.elt = elt;
        g.total = (u64)n_polys * n_limbs << ctx->log_n;
        if (g.total == 0)
            return CAT_OK;
        if (ntt_domain)
            hipLaunchKernelGGL(galois_ntt_kernel, dim3(grid_for(g.total)), dim3(256), 0, s, g);
        else
            hipLaunchKernelGGL(galois_coeff_kernel, dim3(grid_for(g.total)), dim3(256), 0, s, g);
        CAT_HIP(hipGetLastError());
        return CAT_OK;
    }

    cat_status launch_copy_c0_zero_c1(cat_ctx *ctx, const u64 *in, u64 *out, uint32_t L, uint32_t batch, hipStream_t s)
    {
        const u64 words2 = ((u64)L << ctx->log_n) >> 1, total2 = words2 * batch;
        if (total2 == 0)
            return CAT_OK;
        hipLaunchKernelGGL(copy_c0_zero_c1_kernel, dim3(grid_for(total2)), dim3(256), 0, s, in, out, words2, total2);
        CAT_HIP(hipGetLastError());
        return CAT_OK;
    }

    cat_status launch_drop_limb(cat_ctx *ctx, const u64 *in, u64 *out, uint32_t L, uint32_t n_polys, hipStream_t s)
    {
        const u64 total2 = ((u64)n_polys * (L - 1) << ctx->log_n) >> 1;
        if (total2 == 0)
            return CAT_OK;
        hipLaunchKernelGGL(drop_limb_kernel, dim3(grid_for(total2)), dim3(256), 0, s, in, out, L, ctx->log_n, total2);
        CAT_HIP(hipGetLastError());
        return CAT_OK;
    }
} // namespace cat
