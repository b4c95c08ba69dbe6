// valu_rate.hip -- issue rate of vector instruction classes on gfx950: 8 waves per SIMD resident, eight
// independent chains per wave, the instruction under test written in inline assembly so the optimiser cannot fold it.
// Prints the cycles one SIMD spends per wave-instruction.   hipcc --offload-arch=gfx950 -O3 -std=c++17 valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REPS 8192
#define CHAINS 8

#define KERNEL2(name, ASMSTR)                                                                       \
    __global__ __launch_bounds__(256) void name(unsigned* out, unsigned seed)                       \
    {                                                                                               \
        unsigned a[CHAINS];                                                                         \
        unsigned b = seed + threadIdx.x * 3u + 1u;                                                  \
        for (int k = 0; k < CHAINS; k++) a[k] = seed * (k + 1) + threadIdx.x;                       \
        for (int r = 0; r < REPS / CHAINS; r++) {                                                   \
            _Pragma("unroll") for (int k = 0; k < CHAINS; k++) asm volatile(ASMSTR : "+v"(a[k]) : "v"(b) : "vcc", "s10", "s11"); \
        }                                                                                           \
        unsigned acc = 0;                                                                           \
        for (int k = 0; k < CHAINS; k++) acc ^= a[k];                                               \
        if (acc == 0x12345678u) out[0] = acc;                                                       \
    }

// 64-bit operand classes (packed fp32, fp64, 64-bit integer multiply-add)
#define KERNEL64(name, ASMSTR)                                                                      \
    __global__ __launch_bounds__(256) void name(unsigned* out, unsigned seed)                       \
    {                                                                                               \
        unsigned long long a[CHAINS];                                                               \
        unsigned long long b = 0x3F8000003F800000ull + seed + threadIdx.x;                          \
        for (int k = 0; k < CHAINS; k++) a[k] = 0x3F8000003F800000ull + seed * (k + 1) + threadIdx.x; \
        for (int r = 0; r < REPS / CHAINS; r++) {                                                   \
            _Pragma("unroll") for (int k = 0; k < CHAINS; k++) asm volatile(ASMSTR : "+v"(a[k]) : "v"(b) : "vcc", "s10", "s11"); \
        }                                                                                           \
        unsigned long long acc = 0;                                                                 \
        for (int k = 0; k < CHAINS; k++) acc ^= a[k];                                               \
        if (acc == 0x12345678u) out[0] = (unsigned)acc;                                             \
    }

KERNEL2(k_and, "v_and_b32 %0, %0, %1")
KERNEL2(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL2(k_min_u32, "v_min_u32 %0, %0, %1")
KERNEL2(k_lshl_or, "v_lshl_or_b32 %0, %0, 3, %1")
KERNEL2(k_add3, "v_add3_u32 %0, %0, %1, %1")
KERNEL2(k_bfe, "v_bfe_u32 %0, %0, 3, 8")
KERNEL2(k_pk_min_i16, "v_pk_min_i16 %0, %0, %1")
KERNEL2(k_pk_sub_i16, "v_pk_sub_i16 %0, %0, %1")
KERNEL2(k_pk_min_f16, "v_pk_min_f16 %0, %0, %1")
KERNEL2(k_pk_add_f16, "v_pk_add_f16 %0, %0, %1")
KERNEL2(k_min_f32, "v_min_f32 %0, %0, %1")
KERNEL2(k_add_f32, "v_add_f32 %0, %0, %1")
KERNEL2(k_fma_f32, "v_fma_f32 %0, %0, %1, %1")
KERNEL2(k_mul_i24, "v_mul_i32_i24 %0, %0, %1")
KERNEL2(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL2(k_perm, "v_perm_b32 %0, %0, %1, %1")
KERNEL2(k_alignbit, "v_alignbit_b32 %0, %0, %1, 13")
KERNEL2(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL2(k_cmp, "v_cmp_lt_u32 vcc, %0, %1")
KERNEL2(k_dot4, "v_dot4_u32_u8 %0, %0, %1, %0")
KERNEL2(k_sad, "v_sad_u8 %0, %0, %1, %0")
KERNEL2(k_bcnt, "v_bcnt_u32_b32 %0, %1, %0")
KERNEL2(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
KERNEL2(k_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
KERNEL2(k_dpp, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL2(k_mov_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL2(k_cvt, "v_cvt_f32_ubyte0 %0, %1")

KERNEL2(k_or, "v_or_b32 %0, %0, %1")
KERNEL2(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL2(k_sub_u32, "v_sub_u32 %0, %0, %1")
KERNEL2(k_subrev_u32, "v_subrev_u32 %0, %0, %1")
KERNEL2(k_lshlrev, "v_lshlrev_b32 %0, 3, %0")
KERNEL2(k_lshrrev, "v_lshrrev_b32 %0, 3, %0")
KERNEL2(k_lshlrev_v, "v_lshlrev_b32 %0, %1, %0")
KERNEL2(k_ashrrev, "v_ashrrev_i32 %0, 3, %0")
KERNEL2(k_mul_f32, "v_mul_f32 %0, %0, %1")
KERNEL2(k_sub_f32, "v_sub_f32 %0, %0, %1")
KERNEL2(k_max_f32, "v_max_f32 %0, %0, %1")
KERNEL2(k_fmac_f32, "v_fmac_f32 %0, %1, %1")
KERNEL2(k_fma_f32_3, "v_fma_f32 %0, %0, %1, %0")
KERNEL2(k_max_u32, "v_max_u32 %0, %0, %1")
KERNEL2(k_max_i32, "v_max_i32 %0, %0, %1")
KERNEL2(k_mov, "v_mov_b32 %0, %1")
KERNEL2(k_cndmask_e32, "v_cndmask_b32_e32 %0, %0, %1, vcc")
KERNEL2(k_cndmask_e64, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
KERNEL2(k_and_or, "v_and_or_b32 %0, %0, %1, %1")
KERNEL2(k_or3, "v_or3_b32 %0, %0, %1, %1")
KERNEL2(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
KERNEL2(k_add_lshl, "v_add_lshl_u32 %0, %0, %1, 1")
KERNEL2(k_xad, "v_xad_u32 %0, %0, %1, %1")
KERNEL2(k_add_u16, "v_add_u16 %0, %0, %1")
KERNEL2(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
KERNEL2(k_pk_max_i16, "v_pk_max_i16 %0, %0, %1")
KERNEL2(k_pk_lshl, "v_pk_lshlrev_b16 %0, 1, %0")
KERNEL2(k_pk_mad_i16, "v_pk_mad_i16 %0, %0, %1, %1")
KERNEL2(k_min3, "v_min3_u32 %0, %0, %1, %1")
KERNEL2(k_med3, "v_med3_i32 %0, %0, %1, %1")
KERNEL2(k_add_e64, "v_add_u32_e64 %0, %0, %1")
KERNEL2(k_and_e64, "v_and_b32_e64 %0, %0, %1")
KERNEL2(k_add_sgpr, "v_add_u32 %0, s4, %0")
KERNEL2(k_add_lit, "v_add_u32 %0, 0x12345, %0")
KERNEL2(k_and_lit, "v_and_b32 %0, 0x00ff00ff, %0")
KERNEL2(k_add_inl, "v_add_u32 %0, 7, %0")
KERNEL2(k_add_co, "v_add_co_u32 %0, vcc, %0, %1")
KERNEL2(k_cmp_e64, "v_cmp_lt_u32_e64 s[10:11], %0, %1")
KERNEL2(k_cmp_f32, "v_cmp_lt_f32 vcc, %0, %1")
KERNEL2(k_bfi, "v_bfi_b32 %0, %1, %0, %1")
KERNEL2(k_not, "v_not_b32 %0, %0")
KERNEL2(k_ffbh, "v_ffbh_u32 %0, %1")
KERNEL2(k_cvt_i, "v_cvt_f32_i32 %0, %1")
KERNEL2(k_rcp, "v_rcp_f32 %0, %1")
KERNEL2(k_readlane, "v_readfirstlane_b32 s10, %0")
KERNEL2(k_lshl_or3, "v_lshl_or_b32 %0, %1, 8, %0")
KERNEL2(k_msad, "v_msad_u8 %0, %0, %1, %0")
KERNEL2(k_sad_u16, "v_sad_u16 %0, %0, %1, %0")

// operand-pattern probes: three distinct registers, mixed streams
#define KERNEL3(name, ASMSTR)                                                                       \
    __global__ __launch_bounds__(256) void name(unsigned* out, unsigned seed)                       \
    {                                                                                               \
        unsigned a[CHAINS];                                                                         \
        unsigned b = seed + threadIdx.x * 3u + 1u, c = seed ^ (threadIdx.x * 7u);                   \
        for (int k = 0; k < CHAINS; k++) a[k] = seed * (k + 1) + threadIdx.x;                       \
        for (int r = 0; r < REPS / CHAINS; r++) {                                                   \
            _Pragma("unroll") for (int k = 0; k < CHAINS; k++) asm volatile(ASMSTR : "+v"(a[k]) : "v"(b), "v"(c) : "vcc", "s10", "s11"); \
        }                                                                                           \
        unsigned acc = 0;                                                                           \
        for (int k = 0; k < CHAINS; k++) acc ^= a[k];                                               \
        if (acc == 0x12345678u) out[0] = acc;                                                       \
    }
KERNEL3(k3_add_bc, "v_add_u32 %0, %1, %2")
KERNEL3(k3_and_bc, "v_and_b32 %0, %1, %2")
KERNEL3(k3_fma_abc, "v_fma_f32 %0, %0, %1, %2")
KERNEL3(k3_fma_bca, "v_fma_f32 %0, %1, %2, %0")
KERNEL3(k3_fmac, "v_fmac_f32 %0, %1, %2")
KERNEL3(k3_lshrrev_v, "v_lshrrev_b32 %0, %1, %0")
KERNEL3(k3_lshl1, "v_lshlrev_b32 %0, 1, %0")
KERNEL3(k3_lshl16, "v_lshlrev_b32 %0, 16, %1")
KERNEL3(k3_lshr_b, "v_lshrrev_b32 %0, 8, %1")
KERNEL3(k3_mix2, "v_and_b32 %0, %0, %1\n v_pk_min_i16 %0, %0, %2")
KERNEL3(k3_mix3, "v_and_b32 %0, %0, %1\n v_add_u32 %0, %0, %2\n v_pk_min_i16 %0, %0, %2")
KERNEL3(k3_cmp_cnd, "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
KERNEL3(k3_sub_f32, "v_sub_f32 %0, %1, %2")
KERNEL3(k3_mul_f32, "v_mul_f32 %0, %1, %2")
KERNEL3(k3_mul_u24, "v_mul_u32_u24 %0, %1, %2")
KERNEL3(k3_mad_u24, "v_mad_u32_u24 %0, %1, %2, %0")
KERNEL3(k3_and_sgpr, "v_and_b32 %0, s4, %0")
KERNEL3(k3_xor_b, "v_xor_b32 %0, %1, %0")
KERNEL3(k3_sub_co, "v_sub_co_u32 %0, vcc, %0, %1")
KERNEL3(k3_max_i16, "v_max_i16 %0, %0, %1")
KERNEL3(k3_max_u16, "v_max_u16 %0, %0, %1")
KERNEL3(k3_sub_u16, "v_sub_u16 %0, %0, %1")
KERNEL3(k3_mul_lo_u16, "v_mul_lo_u16 %0, %0, %1")
KERNEL3(k3_lshr_u16, "v_lshrrev_b16 %0, 1, %0")
KERNEL3(k3_add_f16, "v_add_f16 %0, %0, %1")
KERNEL3(k3_mul_f16, "v_mul_f16 %0, %0, %1")
KERNEL3(k3_max_f16, "v_max_f16 %0, %0, %1")
KERNEL64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %1")
KERNEL64(k_pk_add_f32, "v_pk_add_f32 %0, %0, %1")
KERNEL64(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
KERNEL64(k_fma_f64, "v_fma_f64 %0, %0, %1, %1")
__global__ __launch_bounds__(256) void k_mad_u64(unsigned* out, unsigned seed)
{
    unsigned long long a[CHAINS];
    unsigned b = seed + threadIdx.x * 3u + 1u;
    for (int k = 0; k < CHAINS; k++) a[k] = seed * (k + 1) + threadIdx.x;
    for (int r = 0; r < REPS / CHAINS; r++) {
        _Pragma("unroll") for (int k = 0; k < CHAINS; k++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(a[k]) : "v"(b) : "vcc");
    }
    unsigned long long acc = 0;
    for (int k = 0; k < CHAINS; k++) acc ^= a[k];
    if (acc == 0x12345678u) out[0] = (unsigned)acc;
}
KERNEL64(k_lshl_b64, "v_lshlrev_b64 %0, 3, %0")

template <class K>
static void run(const char* name, K kern)
{
    unsigned* d;
    (void)hipMalloc(&d, 64);
    const int blocks = 256 * 8 * 4;   // 8 waves per SIMD resident, four rounds
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int w = 0; w < 5; w++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 7u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a, 0);
    const int launches = 10;
    for (int w = 0; w < launches; w++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 7u + w);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double winstr_per_simd = (double)blocks * 4.0 * REPS * launches / 1024.0;
    const double ns_per = ms * 1e6 / winstr_per_simd;
    printf("%-16s %8.3f ms  %6.3f ns per wave-instruction and SIMD = %5.2f cycles at 2.4 GHz\n", name, ms, ns_per, ns_per * 2.4);
    (void)hipFree(d);
}

#define RUN(k) run(#k, k)

int main()
{
    RUN(k_and); RUN(k_add_u32); RUN(k_min_u32); RUN(k_lshl_or); RUN(k_add3); RUN(k_bfe);
    RUN(k_pk_min_i16); RUN(k_pk_sub_i16); RUN(k_pk_min_f16); RUN(k_pk_add_f16);
    RUN(k_min_f32); RUN(k_add_f32); RUN(k_fma_f32); RUN(k_mul_i24); RUN(k_mul_lo);
    RUN(k_perm); RUN(k_alignbit); RUN(k_cndmask); RUN(k_cmp); RUN(k_dot4); RUN(k_sad); RUN(k_bcnt); RUN(k_mbcnt);
    RUN(k_sdwa); RUN(k_dpp); RUN(k_mov_dpp); RUN(k_cvt);
    RUN(k_or); RUN(k_xor); RUN(k_sub_u32); RUN(k_subrev_u32); RUN(k_lshlrev); RUN(k_lshrrev); RUN(k_lshlrev_v); RUN(k_ashrrev); RUN(k_mul_f32); RUN(k_sub_f32); RUN(k_max_f32); RUN(k_fmac_f32); RUN(k_fma_f32_3); RUN(k_max_u32); RUN(k_max_i32); RUN(k_mov); RUN(k_cndmask_e32); RUN(k_cndmask_e64); RUN(k_and_or); RUN(k_or3); RUN(k_lshl_add); RUN(k_add_lshl); RUN(k_xad); RUN(k_add_u16); RUN(k_pk_add_u16); RUN(k_pk_max_i16); RUN(k_pk_lshl); RUN(k_pk_mad_i16); RUN(k_min3); RUN(k_med3); RUN(k_add_e64); RUN(k_and_e64); RUN(k_add_sgpr); RUN(k_add_lit); RUN(k_and_lit); RUN(k_add_inl); RUN(k_add_co); RUN(k_cmp_e64); RUN(k_cmp_f32); RUN(k_bfi); RUN(k_not); RUN(k_ffbh); RUN(k_cvt_i); RUN(k_rcp); RUN(k_readlane); RUN(k_lshl_or3); RUN(k_msad); RUN(k_sad_u16);
    RUN(k3_add_bc); RUN(k3_and_bc); RUN(k3_fma_abc); RUN(k3_fma_bca); RUN(k3_fmac); RUN(k3_lshrrev_v); RUN(k3_lshl1); RUN(k3_lshl16); RUN(k3_lshr_b); RUN(k3_mix2); RUN(k3_mix3); RUN(k3_cmp_cnd); RUN(k3_sub_f32); RUN(k3_mul_f32); RUN(k3_mul_u24); RUN(k3_mad_u24); RUN(k3_and_sgpr); RUN(k3_xor_b); RUN(k3_sub_co); RUN(k3_max_i16); RUN(k3_max_u16); RUN(k3_sub_u16); RUN(k3_mul_lo_u16); RUN(k3_lshr_u16); RUN(k3_add_f16); RUN(k3_mul_f16); RUN(k3_max_f16);
    RUN(k_pk_fma_f32); RUN(k_pk_add_f32); RUN(k_pk_mul_f32); RUN(k_fma_f64); RUN(k_mad_u64); RUN(k_lshl_b64);
    return 0;
}
