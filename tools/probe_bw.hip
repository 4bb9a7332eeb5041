// probe_bw.hip -- microbenchmark: how many random 64-byte lines per second can MI355X deliver from a table of
// a given size (L2 / Infinity Cache / HBM resident)?  Sets the ceiling for the k-mer store probe.
//   ./probe_bw <table_MiB> <lines_in_flight_per_lane> <bytes_per_lane_per_line: 8|16|64> [blocks_per_cu]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef unsigned long long u64;

__device__ __forceinline__ u64 mix(u64 x) {
    x ^= x >> 31; x *= 0x7fb5d329728ea185ULL; x ^= x >> 27; x *= 0x81dadef4bc2dd44dULL; x ^= x >> 33; return x;
}

template <int INFLIGHT, int BYTES>
__global__ __launch_bounds__(256) void probe(const u64 *table, u64 mask, int iters, u64 *out) {
    u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 acc = 0;
    u64 state = mix(tid + 12345);
    for (int it = 0; it < iters; it++) {
        u64 idx[INFLIGHT];
#pragma unroll
        for (int j = 0; j < INFLIGHT; j++) { state = mix(state + j + 1); idx[j] = state & mask; }
#pragma unroll
        for (int j = 0; j < INFLIGHT; j++) {
            const u64 *p = table + idx[j] * 8;
            if (BYTES == 8) acc += p[0];
            else if (BYTES == 16) { ulonglong2 v = *(const ulonglong2 *)p; acc += v.x ^ v.y; }
            else { const ulonglong2 *q = (const ulonglong2 *)p; ulonglong2 a = q[0], b = q[1], c = q[2], d = q[3];
                   acc += a.x ^ a.y ^ b.x ^ b.y ^ c.x ^ c.y ^ d.x ^ d.y; }
        }
    }
    if (acc == 0x1234567) out[0] = acc;
}

template <int INFLIGHT, int BYTES>
double run(const u64 *table, u64 mask, int grid, int iters, u64 *out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<INFLIGHT, BYTES>), dim3(grid), dim3(256), 0, 0, table, mask, 2, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<INFLIGHT, BYTES>), dim3(grid), dim3(256), 0, 0, table, mask, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double lines = (double)grid * 256 * iters * INFLIGHT;
    return lines / (ms * 1e-3);
}

int main(int argc, char **argv) {
    size_t mib = argc > 1 ? atol(argv[1]) : 64;
    int bpc = argc > 2 ? atoi(argv[2]) : 8;
    size_t n_lines = mib * 1024 * 1024 / 64;
    u64 mask = n_lines - 1;
    u64 *table, *out;
    hipMalloc(&table, n_lines * 64); hipMemset(table, 1, n_lines * 64); hipMalloc(&out, 8);
    int grid = 256 * bpc;
    int iters = 400;
    printf("table %zu MiB, grid %d x 256\n", mib, grid);
#define R(I, B) { double r = run<I, B>(table, mask, grid, iters / I + 1, out); printf("inflight %d bytes %2d: %.1f G lines/s = %.2f TB/s (64B lines)\n", I, B, r / 1e9, r * 64 / 1e12); }
    R(1, 8) R(2, 8) R(4, 8) R(8, 8) R(1, 16) R(4, 16) R(1, 64) R(2, 64) R(4, 64) R(8, 64)
    return 0;
}
