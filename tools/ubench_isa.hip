// Instruction-throughput probe for the integer VALU ops the 64-bit modular butterflies are built from.
// Prints SIMD cycles per wave-instruction (assuming 2.4 GHz) with every SIMD kept full (8 waves/SIMD).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_isa.hip -o tools/ubench_isa
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// 8 independent chains per lane, ITER iterations
#define KERNEL32(NAME, ASM)                                                                        \
    __global__ __launch_bounds__(256) void NAME(u32 *out, u32 a, u32 b, int iters)                 \
    {                                                                                              \
        u32 x[8], y = a + threadIdx.x, z = b ^ threadIdx.x;                                        \
        for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 8 + i + a;                                \
        for (int it = 0; it < iters; ++it)                                                         \
        {                                                                                          \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(x[i]) : "v"(y), "v"(z)); \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(x[i]) : "v"(y), "v"(z)); \
        }                                                                                          \
        u32 s = 0;                                                                                 \
        for (int i = 0; i < 8; ++i) s ^= x[i];                                                     \
        if (s == 0x12345) out[0] = s;                                                              \
    }

#define KERNEL64(NAME, ASM)                                                                        \
    __global__ __launch_bounds__(256) void NAME(u32 *out, u32 a, u32 b, int iters)                 \
    {                                                                                              \
        u64 x[8];                                                                                  \
        u32 y = a + threadIdx.x, z = b ^ threadIdx.x;                                              \
        u64 w = ((u64)y << 32) | z;                                                                \
        for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 8 + i + a;                                \
        for (int it = 0; it < iters; ++it)                                                         \
        {                                                                                          \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(x[i]) : "v"(y), "v"(z), "v"(w) : "vcc"); \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(x[i]) : "v"(y), "v"(z), "v"(w) : "vcc"); \
        }                                                                                          \
        u64 s = 0;                                                                                 \
        for (int i = 0; i < 8; ++i) s ^= x[i];                                                     \
        if (s == 0x12345) out[0] = (u32)s;                                                         \
    }

KERNEL32(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL32(k_add3_u32, "v_add3_u32 %0, %0, %1, %2")
KERNEL32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1")
KERNEL32(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_min_u32, "v_min_u32 %0, %0, %1")
KERNEL32(k_add_co, "v_add_co_u32 %0, vcc, %0, %1")
KERNEL32(k_addc_co, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNEL32(k_cmp_u32, "v_cmp_le_u32 vcc, %0, %1")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7")
KERNEL32(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL32(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL32(k_sub_nc, "v_sub_u32 %0, %0, %1")
KERNEL32(k_mul_lo_u16, "v_mul_lo_u16 %0, %0, %1")
KERNEL32(k_mad_u16, "v_mad_u16 %0, %0, %1, %2")
KERNEL32(k_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
KERNEL32(k_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
KERNEL32(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL32(k_dot4_u32_u8, "v_dot4_u32_u8 %0, %0, %1, %2")

KERNEL32(k_and_b32, "v_and_b32 %0, %0, %1")
KERNEL32(k_ashrrev_i32, "v_ashrrev_i32 %0, 31, %0")
KERNEL32(k_cmp_lt_i32, "v_cmp_lt_i32 vcc, %0, %1")
KERNEL32(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
KERNEL32(k_cvt_u32_f32, "v_cvt_u32_f32 %0, %0")
KERNEL32(k_mul_f32, "v_mul_f32 %0, %0, %1")
KERNEL32(k_not_b32, "v_not_b32 %0, %0")
KERNEL32(k_sub_co, "v_sub_co_u32 %0, vcc, %0, %1")
KERNEL32(k_subb_co, "v_subb_co_u32 %0, vcc, %0, %1, vcc")
KERNEL32(k_bfe_u32, "v_bfe_u32 %0, %0, 3, 9")
KERNEL32(k_lshrrev_b32, "v_lshrrev_b32 %0, 28, %0")
KERNEL32(k_and_or_b32, "v_and_or_b32 %0, %0, %1, %2")
KERNEL32(k_mov_b32, "v_mov_b32 %0, %1")

KERNEL64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
KERNEL64(k_mad_u64_u32_z, "v_mad_u64_u32 %0, vcc, %1, %2, 0")
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %3")
KERNEL64(k_cmp_u64, "v_cmp_le_u64 vcc, %0, %3")
KERNEL64(k_fma_f64, "v_fma_f64 %0, %0, %3, %3")
KERNEL64(k_mul_f64, "v_mul_f64 %0, %0, %3")
KERNEL64(k_add_f64, "v_add_f64 %0, %0, %3")
KERNEL64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %3, %3")
KERNEL64(k_lshlrev_b64, "v_lshlrev_b64 %0, 3, %0")
KERNEL64(k_mov_b64, "v_mov_b64 %0, %3")
KERNEL64(k_pk_add_u32x, "v_pk_mul_f32 %0, %0, %3")

typedef void (*kfn)(u32 *, u32, u32, int);

static void run(const char *name, kfn k, u32 *buf, int waves_per_simd)
{
    const int iters = 40000;
    const int blocks = 256 * waves_per_simd; // 256 CUs x 4 SIMDs x waves_per_simd waves, 4 waves per block
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, buf, 3u, 5u, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, buf, 3u, 5u, iters);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    // wave-instructions per SIMD: blocks*4 waves / 1024 SIMDs * iters*16
    const double wi = (double)blocks * 4 / 1024.0 * iters * 16;
    printf("%-18s w/simd=%d : %.3f ms  %.2f cyc/wave-instr @2.4GHz\n", name, blocks * 4 / 1024, ms, ms * 1e-3 * 2.4e9 / wi);
}

int main()
{
    u32 *buf;
    CK(hipMalloc(&buf, 4096));
#define R(k) run(#k, k, buf, 4);
    R(k_add_u32) R(k_add3_u32) R(k_mul_lo_u32) R(k_mul_hi_u32) R(k_mul_u32_u24) R(k_mul_hi_u32_u24) R(k_mad_u32_u24)
    R(k_cndmask) R(k_min_u32) R(k_add_co) R(k_addc_co) R(k_cmp_u32) R(k_alignbit) R(k_xor) R(k_lshl_add) R(k_sub_nc)
    R(k_and_b32) R(k_ashrrev_i32) R(k_cmp_lt_i32) R(k_cvt_f32_u32) R(k_cvt_u32_f32) R(k_mul_f32) R(k_not_b32) R(k_sub_co)
    R(k_subb_co) R(k_bfe_u32) R(k_lshrrev_b32) R(k_and_or_b32) R(k_mov_b32)
    R(k_mul_lo_u16) R(k_mad_u16) R(k_pk_mul_lo_u16) R(k_pk_mad_u16) R(k_fma_f32) R(k_dot4_u32_u8)
    R(k_mad_u64_u32) R(k_mad_u64_u32_z) R(k_lshl_add_u64) R(k_cmp_u64) R(k_fma_f64) R(k_mul_f64) R(k_add_f64)
    R(k_pk_fma_f32) R(k_lshlrev_b64) R(k_mov_b64) R(k_pk_add_u32x)
    run("k_mad_u64_u32 w1", k_mad_u64_u32, buf, 1);
    run("k_mul_lo_u32 w1", k_mul_lo_u32, buf, 1);
    run("k_add_u32 w1", k_add_u32, buf, 1);
    return 0;
}
