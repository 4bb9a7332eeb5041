// cycles a SIMD spends per wave64 instruction of one kind, at 8 waves per SIMD (developer tool; which vector instructions of the match
// kernel's mix are the dear ones):  hipcc --offload-arch=gfx950 -O2 -o tools/valu_rates tools/valu_rates.hip && tools/valu_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef unsigned long long u64;
#define ATTR __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
#define R8(x) x x x x x x x x
#define KERNEL(name, body, ...)                                                        \
    ATTR void name(uint32_t *out, int iters) {                                         \
        uint32_t a = threadIdx.x, b = a * 3u + 1u, c = a ^ 0x55u, d = a + 7u;          \
        u64 p = ((u64)a << 32) | b, q = ((u64)c << 32) | d;                            \
        uint32_t s0 = 0;                                                               \
        for (int i = 0; i < iters; i++) asm volatile(R8(R8(body)) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(p), "+v"(q), "+s"(s0) : : "vcc", "scc"); \
        if ((a ^ b ^ c ^ d ^ (uint32_t)p ^ (uint32_t)q ^ s0) == 0x13572468u) out[0] = a; \
    }
// (two independent chains where the instruction has a destination: the dependent-issue latency is hidden by the other seven waves anyway)
KERNEL(k_add, "v_add_u32 %0, %0, %1\n")
KERNEL(k_and, "v_and_b32 %0, %0, %1\n")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc\n")
KERNEL(k_cmp32, "v_cmp_eq_u32 vcc, %0, %1\n")
KERNEL(k_cmp64, "v_cmp_eq_u64 vcc, %4, %5\n")
KERNEL(k_lshl_add_u64, "v_lshl_add_u64 %4, %4, 3, %5\n")
KERNEL(k_lshlrev_b64, "v_lshlrev_b64 %4, 3, %4\n")
KERNEL(k_lshrrev_b64, "v_lshrrev_b64 %4, %0, %4\n")
KERNEL(k_mad_u64_u32, "v_mad_u64_u32 %4, vcc, %0, %1, %5\n")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1\n")
KERNEL(k_mul_hi, "v_mul_hi_u32 %0, %0, %1\n")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, %2\n")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 4, 3\n")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2\n")
KERNEL(k_min3, "v_min3_u32 %0, %0, %1, %2\n")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2\n")
KERNEL(k_lshl_add_u32, "v_lshl_add_u32 %0, %0, 2, %1\n")
KERNEL(k_bfrev, "v_bfrev_b32 %0, %0\n")
KERNEL(k_readlane, "v_readlane_b32 %6, %0, 5\n")
KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0\n")
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_add_lit, "v_add_u32 %0, 0x12345678, %0\n")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2\n")
KERNEL(k_or, "v_or_b32 %0, %0, %1\n")
KERNEL(k_xor, "v_xor_b32 %0, %0, %1\n")
KERNEL(k_sub, "v_sub_u32 %0, %0, %1\n")
KERNEL(k_lshlrev_b32, "v_lshlrev_b32 %0, 3, %0\n")
KERNEL(k_lshrrev_b32, "v_lshrrev_b32 %0, %1, %0\n")
KERNEL(k_min_u32, "v_min_u32 %0, %0, %1\n")
KERNEL(k_mov, "v_mov_b32 %0, %1\n")
KERNEL(k_mul_u24, "v_mul_u32_u24 %0, %0, %1\n")
KERNEL(k_add_e64, "v_add_u32_e64 %0, %0, %1\n")
KERNEL(k_and_sgpr, "v_and_b32 %0, s40, %0\n")
KERNEL(k_cmp_cnd, "v_cmp_eq_u32 vcc, %0, %1\nv_cndmask_b32 %2, %2, %3, vcc\n")
KERNEL(k_addc, "v_add_co_u32 %0, vcc, %0, %1\nv_addc_co_u32 %2, vcc, %2, %3, vcc\n")
KERNEL(k_and_inline, "v_and_b32 %0, 15, %0\n")
KERNEL(k_and_lit, "v_and_b32 %0, 0x7fff1234, %0\n")
KERNEL(k_add_inline, "v_add_u32 %0, 7, %0\n")
KERNEL(k_add_sgpr, "v_add_u32 %0, s40, %0\n")
KERNEL(k_lshlrev_vgpr, "v_lshlrev_b32 %0, %1, %0\n")
KERNEL(k_lshrrev_inline, "v_lshrrev_b32 %0, 3, %0\n")
KERNEL(k_and_2chains, "v_and_b32 %0, %0, %1\nv_and_b32 %2, %2, %3\n")
KERNEL(k_bfe_2chains, "v_bfe_u32 %0, %0, 4, 3\nv_bfe_u32 %2, %2, 4, 3\n")
KERNEL(k_and_indep, "v_and_b32 %0, %1, %2\n")
KERNEL(k_cmp_cnd3, "v_cmp_eq_u32 vcc, %0, %1\nv_cndmask_b32 %2, %2, %3, vcc\nv_cndmask_b32 %0, %0, %3, vcc\nv_cndmask_b32 %1, %1, %3, vcc\n")
KERNEL(k_cnd_e64_vcc, "v_cndmask_b32_e64 %0, %0, %1, vcc\n")
KERNEL(k_cnd_alt, "v_cndmask_b32 %0, %0, %1, vcc\nv_and_b32 %2, %2, %3\n")
KERNEL(k_cnd2_and, "v_cndmask_b32 %0, %0, %1, vcc\nv_cndmask_b32 %2, %2, %3, vcc\nv_and_b32 %1, %1, %3\n")
KERNEL(k_cnd64_cnd32, "v_cndmask_b32_e64 %0, %0, %1, vcc\nv_cndmask_b32 %2, %2, %3, vcc\n")
KERNEL(k_cnd2_indep, "v_cndmask_b32 %0, %1, %3, vcc\nv_cndmask_b32 %2, %1, %3, vcc\n")
KERNEL(k_cmp_cnd2, "v_cmp_eq_u32 vcc, %0, %1\nv_cndmask_b32 %2, %2, %3, vcc\nv_cndmask_b32 %0, %0, %3, vcc\n")
KERNEL(k_cmp_cnd_and_cnd, "v_cmp_eq_u32 vcc, %0, %1\nv_cndmask_b32 %2, %2, %3, vcc\nv_and_b32 %1, %1, %3\nv_cndmask_b32 %0, %0, %3, vcc\n")
KERNEL(k_cmp_cnd_nop_cnd, "v_cmp_eq_u32 vcc, %0, %1\nv_cndmask_b32 %2, %2, %3, vcc\ns_nop 0\nv_cndmask_b32 %0, %0, %3, vcc\n")
KERNEL(k_cmp64s_cnd64x2, "v_cmp_eq_u32 s[40:41], %0, %1\nv_cndmask_b32_e64 %2, %2, %3, s[40:41]\nv_cndmask_b32_e64 %0, %0, %3, s[40:41]\n")
KERNEL(k_cmp_cnd64x2, "v_cmp_eq_u32 vcc, %0, %1\nv_cndmask_b32_e64 %2, %2, %3, vcc\nv_cndmask_b32_e64 %0, %0, %3, vcc\n")
KERNEL(k_cmp_cnd32_cnd64, "v_cmp_eq_u32 vcc, %0, %1\nv_cndmask_b32 %2, %2, %3, vcc\nv_cndmask_b32_e64 %0, %0, %3, vcc\n")
KERNEL(k_cmp_sgpr, "v_cmp_eq_u32 s[40:41], %0, %1\n")
KERNEL(k_cndmask_sgpr, "v_cndmask_b32 %0, %0, %1, s[40:41]\n")

int main() {
    uint32_t *d;
    hipMalloc(&d, 4096);
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int n_cu = pr.multiProcessorCount, iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
#define RUN(name)                                                                                   \
    {                                                                                               \
        hipLaunchKernelGGL(name, dim3(n_cu * 8), dim3(256), 0, 0, d, 100);                           \
        hipDeviceSynchronize();                                                                     \
        hipEventRecord(e0);                                                                         \
        hipLaunchKernelGGL(name, dim3(n_cu * 8), dim3(256), 0, 0, d, iters);                         \
        hipEventRecord(e1);                                                                         \
        hipEventSynchronize(e1);                                                                    \
        float ms;                                                                                   \
        hipEventElapsedTime(&ms, e0, e1);                                                           \
        const double inst_per_simd = (double)iters * 64.0 * 8.0; /* 64 per iteration, 8 waves */   \
        printf("%-16s %6.2f cycles per wave64 instruction and SIMD (at 2.4 GHz)\n", #name, ms * 1e-3 * 2.4e9 / inst_per_simd); \
    }
    RUN(k_add) RUN(k_and) RUN(k_cndmask) RUN(k_cmp32) RUN(k_cmp64) RUN(k_lshl_add_u64) RUN(k_lshlrev_b64) RUN(k_lshrrev_b64) RUN(k_mad_u64_u32)
    RUN(k_mul_lo) RUN(k_mul_hi) RUN(k_alignbit) RUN(k_bfe) RUN(k_perm) RUN(k_min3) RUN(k_add3) RUN(k_lshl_add_u32) RUN(k_bfrev) RUN(k_readlane)
    RUN(k_mbcnt) RUN(k_mov_dpp) RUN(k_add_lit) RUN(k_and_or) RUN(k_or) RUN(k_xor) RUN(k_sub) RUN(k_lshlrev_b32) RUN(k_lshrrev_b32) RUN(k_min_u32) RUN(k_mov) RUN(k_mul_u24) RUN(k_add_e64) RUN(k_and_sgpr) RUN(k_cmp_cnd) RUN(k_addc) RUN(k_and_inline) RUN(k_and_lit) RUN(k_add_inline) RUN(k_add_sgpr) RUN(k_lshlrev_vgpr) RUN(k_lshrrev_inline) RUN(k_and_2chains) RUN(k_bfe_2chains) RUN(k_and_indep) RUN(k_cmp_cnd3) RUN(k_cnd_e64_vcc) RUN(k_cnd_alt) RUN(k_cnd2_and) RUN(k_cnd64_cnd32) RUN(k_cnd2_indep) RUN(k_cmp_cnd2) RUN(k_cmp_cnd_and_cnd) RUN(k_cmp_cnd_nop_cnd) RUN(k_cmp64s_cnd64x2) RUN(k_cmp_cnd64x2) RUN(k_cmp_cnd32_cnd64) RUN(k_cmp_sgpr) RUN(k_cndmask_sgpr)
    return 0;
}
