// gs_calib.hip -- ceilings of the device a measurement runs on (gs_calibrate, include/gsgpu.h).
//
// bench.py prices the match / filter kernels against resources; the ceilings of those resources are MEASURED here, in the
// same process and on the same chip, instead of being quoted: how many wave64 VALU instructions a SIMD issues per second at
// 8 waves per SIMD (pure full-rate ops, and the match kernel's own mix of 32-bit logic, compares, selects, 64-bit shifts,
// multiplies and cross-lane reads), how many scalar instructions a CU issues, how many scattered vector loads a CU's
// address unit takes, and how many random 64-byte lines per second the memory side delivers from a table of a given
// footprint.  Nothing here is on the product path.
//
// Every issue kernel runs 8 waves per SIMD (256-thread workgroups, 8 per CU, amdgpu_waves_per_eu(8,8)) -- the match kernel's
// occupancy -- and a loop body of hand-placed instructions (inline asm: the count per iteration is exact), long enough that
// the loop's own s_add / s_cmp / s_cbranch are noise.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gsgpu.h"

typedef unsigned long long u64;

#define CAL_ATTR __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))

// ---- 1. pure VALU: 64 independent full-rate 32-bit instructions per iteration, 8 chains
CAL_ATTR void cal_valu_pure(uint32_t *out, int iters) {
    uint32_t a = threadIdx.x, b = a * 3u + 1u, c = a ^ 0x55u, d = a + 7u, e = a | 9u, f = a * 5u, g = a + 11u, h = a ^ 0x1234u;
    for (int i = 0; i < iters; i++) {
#define V8                                 \
    "v_xor_b32 %0, %0, %1\n"               \
    "v_add_u32 %1, %1, %2\n"               \
    "v_and_b32 %2, %2, %3\n"               \
    "v_or_b32 %3, %3, %4\n"                \
    "v_xor_b32 %4, %4, %5\n"               \
    "v_add_u32 %5, %5, %6\n"               \
    "v_xor_b32 %6, %6, %7\n"               \
    "v_add_u32 %7, %7, %0\n"
        asm volatile(V8 V8 V8 V8 V8 V8 V8 V8 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
#undef V8
    }
    if ((a ^ b ^ c ^ d ^ e ^ f ^ g ^ h) == 0x13572468u) out[0] = a;
}
#define CAL_VALU_PURE_PER_ITER 64

// ---- 2. the match kernel's VALU mix (static histogram of gs_match_kernel<true,false,31>'s vector instructions, per 64):
// 13 and, 11 cndmask, 5 mov, 6 32-bit compares, 3 64-bit compares, 3 lshl_add_u64, 3 readlane, 3 add, 3 lshrrev_b32,
// 2 lshlrev_b64, 2 or, 1 xor, 1 lshl_add_u32, 1 lshrrev_b64, 1 bfrev, 1 mad_u64_u32, 1 min3, 1 mul_lo, 1 mbcnt pair (2), 1 bfe
CAL_ATTR void cal_valu_mix(uint32_t *out, int iters) {
    uint32_t a = threadIdx.x, b = a * 3u + 1u, c = a ^ 0x55u, d = a + 7u, e = a | 9u, f = a * 5u, g = a + 11u, h = a ^ 0x1234u;
    u64 p = ((u64)a << 32) | b, q = ((u64)c << 32) | d;
    uint32_t s0;
    for (int i = 0; i < iters; i++) {
        asm volatile(
            // 16
            "v_and_b32 %0, %0, %1\n"
            "v_cmp_eq_u32 vcc, %2, %3\n"
            "v_cndmask_b32 %1, %1, %4, vcc\n"
            "v_and_b32 %2, %2, %5\n"
            "v_lshlrev_b64 %8, 3, %8\n"
            "v_mov_b32 %3, %6\n"
            "v_and_b32 %4, %4, %7\n"
            "v_cndmask_b32 %5, %5, %0, vcc\n"
            "v_readlane_b32 %10, %6, 5\n"
            "v_add_u32 %6, %6, %1\n"
            "v_lshrrev_b32 %7, 1, %7\n"
            "v_cmp_gt_i32 vcc, %0, %2\n"
            "v_cndmask_b32 %3, %3, %5, vcc\n"
            "v_and_b32 %0, %0, %4\n"
            "v_lshl_add_u64 %9, %9, 2, %8\n"
            "v_or_b32 %1, %1, %6\n"
            // 32
            "v_and_b32 %2, %2, %7\n"
            "v_cmp_eq_u64 vcc, %8, %9\n"
            "v_cndmask_b32 %4, %4, %1, vcc\n"
            "v_mov_b32 %5, %3\n"
            "v_and_b32 %6, %6, %0\n"
            "v_mul_lo_u32 %7, %7, %2\n"
            "v_cndmask_b32 %0, %0, %3, vcc\n"
            "v_bfrev_b32 %1, %1\n"
            "v_and_b32 %2, %2, %4\n"
            "v_cmp_ne_u32 vcc, %5, %6\n"
            "v_cndmask_b32 %3, %3, %7, vcc\n"
            "v_lshrrev_b64 %8, 5, %8\n"
            "v_add_u32 %4, %4, %0\n"
            "v_and_b32 %5, %5, %1\n"
            "v_min3_u32 %6, %6, %2, %3\n"
            "v_mov_b32 %7, %4\n"
            // 48
            "v_and_b32 %0, %0, %5\n"
            "v_cmp_eq_u64 vcc, %9, %8\n"
            "v_cndmask_b32 %1, %1, %6, vcc\n"
            "v_lshl_add_u64 %9, %9, 1, %8\n"
            "v_readlane_b32 %10, %7, 9\n"
            "v_lshrrev_b32 %2, 3, %2\n"
            "v_and_b32 %3, %3, %0\n"
            "v_cndmask_b32 %4, %4, %1, vcc\n"
            "v_mad_u64_u32 %8, vcc, %5, %6, %9\n"
            "v_mov_b32 %5, %2\n"
            "v_cmp_lt_i32 vcc, %3, %4\n"
            "v_cndmask_b32 %6, %6, %7, vcc\n"
            "v_and_b32 %7, %7, %0\n"
            "v_lshl_add_u32 %0, %0, 2, %1\n"
            "v_mbcnt_lo_u32_b32 %1, -1, 0\n"
            "v_mbcnt_hi_u32_b32 %1, -1, %1\n"
            // 64
            "v_and_b32 %2, %2, %3\n"
            "v_cmp_eq_u32 vcc, %4, %5\n"
            "v_cndmask_b32 %3, %3, %6, vcc\n"
            "v_lshlrev_b64 %9, 7, %9\n"
            "v_or_b32 %4, %4, %7\n"
            "v_cmp_ne_u64 vcc, %8, %9\n"
            "v_cndmask_b32 %5, %5, %0, vcc\n"
            "v_readlane_b32 %10, %1, 33\n"
            "v_add_u32 %6, %6, %2\n"
            "v_lshl_add_u64 %8, %8, 3, %9\n"
            "v_xor_b32 %7, %7, %3\n"
            "v_lshrrev_b32 %0, 2, %0\n"
            "v_bfe_u32 %1, %4, 3, 9\n"
            "v_mov_b32 %2, %5\n"
            "v_cmp_gt_i32 vcc, %6, %7\n"
            "v_cndmask_b32 %3, %3, %0, vcc\n"
            : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(p), "+v"(q), "=s"(s0)
            :
            : "vcc");
    }
    if ((a ^ b ^ c ^ d ^ e ^ f ^ g ^ h ^ (uint32_t)p ^ (uint32_t)q ^ s0) == 0x13572468u) out[0] = a;
}
#define CAL_VALU_MIX_PER_ITER 64

// ---- 3. scalar ALU: 64 instructions per iteration on 8 chains
CAL_ATTR void cal_salu(uint32_t *out, int iters) {
    uint32_t a = blockIdx.x, b = a * 3u + 1u, c = a ^ 0x55u, d = a + 7u, e = a | 9u, f = a * 5u, g = a + 11u, h = a ^ 0x1234u;
    for (int i = 0; i < iters; i++) {
#define S8                                 \
    "s_xor_b32 %0, %0, %1\n"               \
    "s_add_u32 %1, %1, %2\n"               \
    "s_and_b32 %2, %2, %3\n"               \
    "s_or_b32 %3, %3, %4\n"                \
    "s_lshl_b32 %4, %4, 1\n"               \
    "s_add_u32 %5, %5, %6\n"               \
    "s_xor_b32 %6, %6, %7\n"               \
    "s_add_u32 %7, %7, %0\n"
        asm volatile(S8 S8 S8 S8 S8 S8 S8 S8 : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(e), "+s"(f), "+s"(g), "+s"(h) : : "scc");
#undef S8
    }
    if ((a ^ b ^ c ^ d ^ e ^ f ^ g ^ h) == 0x13572468u) out[0] = a;
}
#define CAL_SALU_PER_ITER 64

// ---- 4. VALU and SALU side by side in one stream (32 + 32 per iteration, alternating): do the two issue ports overlap?
CAL_ATTR void cal_valu_salu(uint32_t *out, int iters) {
    uint32_t a = threadIdx.x, b = a * 3u + 1u, c = a ^ 0x55u, d = a + 7u;
    uint32_t sa = blockIdx.x, sb = sa * 3u + 1u, sc_ = sa ^ 0x55u, sd = sa + 7u;
    for (int i = 0; i < iters; i++) {
#define VS8                                \
    "v_xor_b32 %0, %0, %1\n"               \
    "s_xor_b32 %4, %4, %5\n"               \
    "v_add_u32 %1, %1, %2\n"               \
    "s_add_u32 %5, %5, %6\n"               \
    "v_and_b32 %2, %2, %3\n"               \
    "s_and_b32 %6, %6, %7\n"               \
    "v_add_u32 %3, %3, %0\n"               \
    "s_add_u32 %7, %7, %4\n"
        asm volatile(VS8 VS8 VS8 VS8 VS8 VS8 VS8 VS8
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(sa), "+s"(sb), "+s"(sc_), "+s"(sd)
                     :
                     : "scc");
#undef VS8
    }
    if ((a ^ b ^ c ^ d ^ sa ^ sb ^ sc_ ^ sd) == 0x13572468u) out[0] = a;
}
#define CAL_VS_PER_ITER 64

// ---- 5. vector-memory issue: loads that hit the CU's L1 / the XCD's L2 (64 KiB footprint), 8 independent loads in flight
// per wave.  KIND 0: byte loads of 64 consecutive bytes (the match kernel's base loads); 1: dword loads, 8 distinct words per
// wave (gate words); 2: 16-byte loads, groups of 8 lanes share a 64-byte line (record planes); 3: 16-byte loads, every lane
// its own line (table walk / filter)
template <int KIND>
CAL_ATTR void cal_vmem(const uint8_t *buf, uint32_t *out, int iters) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            x = x * 1664525u + 1013904223u;
            const uint32_t wave_x = __builtin_amdgcn_readfirstlane(x);
            if (KIND == 0) {
                acc += buf[((wave_x >> 8) & 0xffc0u) + lane];
            } else if (KIND == 1) {
                acc += *reinterpret_cast<const uint32_t *>(buf + (((wave_x >> 8) + (lane >> 3) * 4160u) & 0xfffcu));
            } else if (KIND == 2) {
                const uint4 v = *reinterpret_cast<const uint4 *>(buf + (((wave_x >> 8) + (lane >> 3) * 4160u) & 0xffc0u));
                acc += v.x ^ v.y ^ v.z ^ v.w;
            } else {
                const uint4 v = *reinterpret_cast<const uint4 *>(buf + ((x >> 8) & 0xffc0u));
                acc += v.x ^ v.y ^ v.z ^ v.w;
            }
        }
    }
    if (acc == 0x13572468u) out[0] = acc;
}
#define CAL_VMEM_PER_ITER 8

// ---- 6. random 64-byte lines from a table of a given footprint (the store probe's memory-side ceiling): every lane reads
// 16 bytes of its own random line; the best of three shapes (4 or 8 independent lines in flight per lane, 8 or 16 workgroups
// per CU) is the ceiling
__device__ __forceinline__ u64 cal_mix(u64 x) {
    x ^= x >> 31;
    x *= 0x7fb5d329728ea185ULL;
    x ^= x >> 27;
    x *= 0x81dadef4bc2dd44dULL;
    x ^= x >> 33;
    return x;
}
template <int INFLIGHT>
__global__ __launch_bounds__(256) void cal_random_lines(const u64 *table, u64 n_lines, int iters, u64 *out) {
    const u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 acc = 0, state = cal_mix(tid + 12345);
    for (int it = 0; it < iters; it++) {
        u64 idx[INFLIGHT];
#pragma unroll
        for (int j = 0; j < INFLIGHT; j++) {
            state = cal_mix(state + j + 1);
            idx[j] = (u64)(((unsigned __int128)state * n_lines) >> 64);
        }
#pragma unroll
        for (int j = 0; j < INFLIGHT; j++) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(table + idx[j] * 8);
            acc += v.x ^ v.y;
        }
    }
    if (acc == 0x1234567) out[0] = acc;
}

// ---------------------------------------------------------------------------------------------------
static int cal_fail(int code) { return code; }

#define CAL_TRY(x)                                   \
    do {                                             \
        if ((x) != hipSuccess) return cal_fail(GS_E_HIP); \
    } while (0)

template <typename L>
static int cal_time(L launch, double *ms_out) {
    hipEvent_t e0, e1;
    CAL_TRY(hipEventCreate(&e0));
    CAL_TRY(hipEventCreate(&e1));
    launch(true);  // warm-up (code object load, clocks)
    CAL_TRY(hipDeviceSynchronize());
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
        CAL_TRY(hipEventRecord(e0, 0));
        launch(false);
        CAL_TRY(hipEventRecord(e1, 0));
        CAL_TRY(hipEventSynchronize(e1));
        float ms = 0;
        CAL_TRY(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (hipGetLastError() != hipSuccess) return cal_fail(GS_E_HIP);
    *ms_out = best;
    return GS_OK;
}

// out[0] = the rate (see gsgpu.h), out[1] = milliseconds of the timed launch, out[2] = instructions / loads / lines it issued,
// out[3] = compute units of the device
extern "C" int gs_calibrate(int device, int what, int64_t arg, double out[4]) {
    if (!out) return GS_E_INVALID;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) return GS_E_NODEVICE;
    if (device < 0 || device >= n_dev) return GS_E_INVALID;
    CAL_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    CAL_TRY(hipGetDeviceProperties(&prop, device));
    const int n_cu = prop.multiProcessorCount;
    const int grid = n_cu * 8;  // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    uint32_t *d_out = nullptr;
    CAL_TRY(hipMalloc((void **)&d_out, 64));
    double ms = 0, count = 0;
    int rc = GS_OK;
    const double waves = (double)grid * 4;
    switch (what) {
    case GS_CAL_VALU_PURE: {
        const int iters = 20000;
        rc = cal_time([&](bool warm) { hipLaunchKernelGGL(cal_valu_pure, dim3(grid), dim3(256), 0, 0, d_out, warm ? 16 : iters); }, &ms);
        count = waves * iters * CAL_VALU_PURE_PER_ITER;
        break;
    }
    case GS_CAL_VALU_MIX: {
        const int iters = 20000;
        rc = cal_time([&](bool warm) { hipLaunchKernelGGL(cal_valu_mix, dim3(grid), dim3(256), 0, 0, d_out, warm ? 16 : iters); }, &ms);
        count = waves * iters * CAL_VALU_MIX_PER_ITER;
        break;
    }
    case GS_CAL_SALU: {
        const int iters = 20000;
        rc = cal_time([&](bool warm) { hipLaunchKernelGGL(cal_salu, dim3(grid), dim3(256), 0, 0, d_out, warm ? 16 : iters); }, &ms);
        count = waves * iters * CAL_SALU_PER_ITER;
        break;
    }
    case GS_CAL_VALU_SALU: {
        const int iters = 20000;
        rc = cal_time([&](bool warm) { hipLaunchKernelGGL(cal_valu_salu, dim3(grid), dim3(256), 0, 0, d_out, warm ? 16 : iters); }, &ms);
        count = waves * iters * CAL_VS_PER_ITER;
        break;
    }
    case GS_CAL_VMEM_BYTES:
    case GS_CAL_VMEM_WORDS:
    case GS_CAL_VMEM_SHARED_LINES:
    case GS_CAL_VMEM_SCATTERED: {
        uint8_t *buf = nullptr;
        CAL_TRY(hipMalloc((void **)&buf, 1 << 17));
        CAL_TRY(hipMemset(buf, 1, 1 << 17));
        const int iters = what == GS_CAL_VMEM_SCATTERED ? 400 : 2000;
        rc = cal_time(
            [&](bool warm) {
                const int n = warm ? 4 : iters;
                if (what == GS_CAL_VMEM_BYTES)
                    hipLaunchKernelGGL(cal_vmem<0>, dim3(grid), dim3(256), 0, 0, buf, d_out, n);
                else if (what == GS_CAL_VMEM_WORDS)
                    hipLaunchKernelGGL(cal_vmem<1>, dim3(grid), dim3(256), 0, 0, buf, d_out, n);
                else if (what == GS_CAL_VMEM_SHARED_LINES)
                    hipLaunchKernelGGL(cal_vmem<2>, dim3(grid), dim3(256), 0, 0, buf, d_out, n);
                else
                    hipLaunchKernelGGL(cal_vmem<3>, dim3(grid), dim3(256), 0, 0, buf, d_out, n);
            },
            &ms);
        count = waves * iters * CAL_VMEM_PER_ITER;
        hipFree(buf);
        break;
    }
    case GS_CAL_RANDOM_LINES: {
        if (arg < (1 << 16) || arg > ((int64_t)64 << 30)) {
            hipFree(d_out);
            return GS_E_INVALID;
        }
        const u64 n_lines = (u64)arg / 64;
        u64 *table = nullptr;
        if (hipMalloc((void **)&table, n_lines * 64) != hipSuccess) {
            hipFree(d_out);
            return GS_E_NOMEM;
        }
        CAL_TRY(hipMemset(table, 1, n_lines * 64));
        double best_rate = 0;
        for (int shape = 0; shape < 3 && rc == GS_OK; shape++) {
            // (every shape requests the same number of lines: n_cu * 8 * 256 * 400)
            const int inflight = shape == 1 ? 8 : 4, blocks = n_cu * (shape == 2 ? 16 : 8), iters = (shape == 2 ? 200 : 400) / inflight;
            double t = 0;
            rc = cal_time(
                [&](bool warm) {
                    if (inflight == 8)
                        hipLaunchKernelGGL(cal_random_lines<8>, dim3(blocks), dim3(256), 0, 0, table, n_lines, warm ? 2 : iters, (u64 *)d_out);
                    else
                        hipLaunchKernelGGL(cal_random_lines<4>, dim3(blocks), dim3(256), 0, 0, table, n_lines, warm ? 2 : iters, (u64 *)d_out);
                },
                &t);
            const double c = (double)blocks * 256 * iters * inflight;
            if (rc == GS_OK && c / t > best_rate) {
                best_rate = c / t;
                ms = t;
                count = c;
            }
        }
        hipFree(table);
        break;
    }
    default:
        hipFree(d_out);
        return GS_E_INVALID;
    }
    hipFree(d_out);
    if (rc) return rc;
    out[0] = count / (ms * 1e-3);
    out[1] = ms;
    out[2] = count;
    out[3] = (double)n_cu;
    return GS_OK;
}
