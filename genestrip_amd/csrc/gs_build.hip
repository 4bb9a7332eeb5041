// gs_build.hip -- DB construction on the device (include/gsgpu.h, gs_dbbuild_*): the compute core of the reference's
// FillDBGoal + DBGoal (C/goals/refseq/FillDBGoal.java:297-end, C/goals/refseq/DBGoal.java:188-311 over
// C/refseq/AbstractStoreFastaReader.java:87-115 and C/util/CGATLongBuffer.java:137-229):
//   every k-mer of every genome region -> (canonical k-mer, region) pairs        gs_build_kmers_kernel
//   all pairs sorted by k-mer                                                    rocPRIM radix sort (stable)
//   per distinct k-mer: stored iff a FILL region holds it (KMerSortedArray.putLong, first writer), value = the node of
//   the first such region, then the lowest common ancestor with the node of every UPDATE region that holds it
//   (KMerStore.update + TaxTree.getLowestCommonAncestor, C/tax/TaxTree.java:160-187)       gs_build_reduce_kernel
//   compaction of the stored k-mers in ascending order                                      scan + scatter
// The reference walks the genomes twice with a hash-free sorted array and a Bloom filter in front of it; here one sort
// replaces both walks.  Byte / integer work, HBM-bound (12 bytes per genome base through an 8-pass radix sort).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

typedef unsigned long long u64;

#define GS_BUILD_NONE 0xffffffffffffffffULL
#define GS_BUILD_UPDATE 0x80000000u

// 2-bit code of the reference (C/util/CGAT.java:66-74: C0 G1 A2 T3), 4 = not a base.  lower: enableLowerCaseBases
// (AbstractStoreFastaReader.java:100: CGAT.cgatToUpperCase)
__device__ __forceinline__ uint32_t gs_build_code(uint8_t c, int lower) {
    if (lower && c >= 'a') c = (uint8_t)(c - 32);
    return c == 'C' ? 0u : c == 'G' ? 1u : c == 'A' ? 2u : c == 'T' ? 3u : 4u;
}

// one thread per base position p of the concatenated regions: the k-mer that STARTS at p, if it lies in one region and holds
// only bases.  The reference's ring buffer is reset by a non-base and at a region start
// (CGATLongBuffer.put :141-144, AbstractStoreFastaReader.startRegion), and emits the window of the last k bases whenever it
// is filled and (bases of the region so far) % stepSize == 0 (:103-104): the k-mer over region bases [s, s + k) is taken iff
// all k are bases and (s + k) % stepSize == 0.
__global__ __launch_bounds__(256) void gs_build_kmers_kernel(const uint8_t *seq, const u64 *off, int64_t n_regions, int64_t total, int k,
                                                             int lower, int step, uint32_t first_region, uint32_t update_flag,
                                                             u64 *keys, uint32_t *vals) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        // region of p: the last r with off[r] <= base + p  (off is relative to seq, off[0] = 0)
        int64_t lo = 0, hi = n_regions;  // invariant: off[lo] <= p < off[hi]
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (off[mid] <= (u64)p)
                lo = mid;
            else
                hi = mid;
        }
        u64 key = GS_BUILD_NONE;
        const int64_t s = p - (int64_t)off[lo];
        if ((u64)p + (u64)k <= off[lo + 1] && (s + k) % step == 0) {
            u64 fwd = 0, rev = 0;
            bool ok = true;
            for (int i = 0; i < k; i++) {
                const uint32_t c = gs_build_code(seq[p + i], lower);
                ok = ok && c < 4u;
                fwd = (fwd << 2) | (u64)(c & 3u);
                rev = (rev >> 2) | ((u64)((c & 3u) ^ 1u) << (2 * (k - 1)));  // complement: C<->G, A<->T (CGAT.java:71-74)
            }
            if (ok) key = fwd > rev ? fwd : rev;  // CGAT.standardKMer (:145-147)
        }
        keys[p] = key;
        vals[p] = update_flag | (first_region + (uint32_t)lo);
    }
}

// number of real pairs in the sorted key array (the placeholders sort to the end)
__global__ void gs_build_count_kernel(const u64 *keys, int64_t n, u64 *n_valid) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (keys[i] != GS_BUILD_NONE && (i + 1 == n || keys[i + 1] == GS_BUILD_NONE)) *n_valid = (u64)(i + 1);
}

// TaxTree.getLowestCommonAncestor (C/tax/TaxTree.java:160-187) over value indices; one tree (the API refuses forests)
__device__ __forceinline__ int gs_build_lca(const int32_t *parent, const int32_t *depth, int a, int b) {
    while (depth[a] > depth[b]) a = parent[a];
    while (depth[b] > depth[a]) b = parent[b];
    while (a != b) {
        a = parent[a];
        b = parent[b];
    }
    return a;
}

// the head of every run of equal keys folds its run: flag[i] = 1 and value[i] = node iff the k-mer is stored
__global__ __launch_bounds__(256) void gs_build_reduce_kernel(const u64 *keys, const uint32_t *vals, int64_t n, const int32_t *node_of_region,
                                                              const int32_t *parent, const int32_t *depth, uint32_t *flag, int32_t *value) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const u64 key = keys[i];
        uint32_t f = 0;
        int32_t out = -1;
        if (i == 0 || keys[i - 1] != key) {
            uint32_t first_fill = 0xffffffffu;
            int upd = -1;
            for (int64_t j = i; j < n && keys[j] == key; j++) {
                const uint32_t v = vals[j];
                if (v & GS_BUILD_UPDATE) {
                    const int nd = node_of_region[v & ~GS_BUILD_UPDATE];
                    upd = upd < 0 ? nd : (upd == nd ? upd : gs_build_lca(parent, depth, upd, nd));
                } else
                    first_fill = v < first_fill ? v : first_fill;
            }
            if (first_fill != 0xffffffffu) {
                f = 1;
                out = node_of_region[first_fill];
                if (upd >= 0 && upd != out) out = gs_build_lca(parent, depth, out, upd);
            }
        }
        flag[i] = f;
        value[i] = out;
    }
}

__global__ __launch_bounds__(256) void gs_build_scatter_kernel(const u64 *keys, const int32_t *value, const uint32_t *flag, const u64 *pos,
                                                               int64_t n, int64_t *out_keys, int32_t *out_vals) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (flag[i]) {
            out_keys[pos[i]] = (int64_t)keys[i];
            out_vals[pos[i]] = value[i];
        }
}

static int gs_build_grid(int64_t n) {
    int64_t g = (n + 255) / 256;
    if (g > 65536) g = 65536;
    return g < 1 ? 1 : (int)g;
}

extern "C" hipError_t gs_launch_build_kmers(const uint8_t *seq, const u64 *off, int64_t n_regions, int64_t total, int k, int lower, int step,
                                            uint32_t first_region, int update, u64 *keys, uint32_t *vals, hipStream_t stream) {
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(gs_build_kmers_kernel, dim3(gs_build_grid(total)), dim3(256), 0, stream, seq, off, n_regions, total, k, lower, step,
                       first_region, update ? GS_BUILD_UPDATE : 0u, keys, vals);
    return hipGetLastError();
}

// keys / vals: n pairs, sorted in place through the alternate buffers (keys_alt / vals_alt: n each).  Returns the sorted arrays
// in *keys_out / *vals_out (one of the two buffers each).
extern "C" hipError_t gs_build_sort(u64 *keys, u64 *keys_alt, uint32_t *vals, uint32_t *vals_alt, int64_t n, u64 **keys_out,
                                    uint32_t **vals_out, hipStream_t stream) {
    *keys_out = keys;
    *vals_out = vals;
    if (n <= 1) return hipSuccess;
    rocprim::double_buffer<u64> dk(keys, keys_alt);
    rocprim::double_buffer<uint32_t> dv(vals, vals_alt);
    size_t tmp_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, dk, dv, (size_t)n, 0, 64, stream);
    if (e != hipSuccess) return e;
    void *tmp = nullptr;
    e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1);
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_pairs(tmp, tmp_bytes, dk, dv, (size_t)n, 0, 64, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    hipFree(tmp);
    *keys_out = dk.current();
    *vals_out = dv.current();
    return e;
}

extern "C" hipError_t gs_launch_build_count(const u64 *keys, int64_t n, u64 *n_valid, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(n_valid, 0, sizeof(u64), stream);
    if (e != hipSuccess || n <= 0) return e;
    hipLaunchKernelGGL(gs_build_count_kernel, dim3(gs_build_grid(n)), dim3(256), 0, stream, keys, n, n_valid);
    return hipGetLastError();
}

// flag / value / pos: n entries of scratch each.  *n_out = number of stored k-mers; out_keys / out_vals must hold them (call
// with out_keys == nullptr first to learn the count: the reduce and the scan are then already done).
extern "C" hipError_t gs_build_reduce(const u64 *keys, const uint32_t *vals, int64_t n, const int32_t *node_of_region, const int32_t *parent,
                                      const int32_t *depth, uint32_t *flag, int32_t *value, u64 *pos, int64_t *n_out, hipStream_t stream) {
    *n_out = 0;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gs_build_reduce_kernel, dim3(gs_build_grid(n)), dim3(256), 0, stream, keys, vals, n, node_of_region, parent, depth, flag,
                       value);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    size_t tmp_bytes = 0;
    e = rocprim::exclusive_scan(nullptr, tmp_bytes, flag, pos, (u64)0, (size_t)n, rocprim::plus<u64>(), stream);
    if (e != hipSuccess) return e;
    void *tmp = nullptr;
    e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1);
    if (e != hipSuccess) return e;
    e = rocprim::exclusive_scan(tmp, tmp_bytes, flag, pos, (u64)0, (size_t)n, rocprim::plus<u64>(), stream);
    u64 last_pos = 0;
    uint32_t last_flag = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&last_pos, pos + (n - 1), sizeof(u64), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&last_flag, flag + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    hipFree(tmp);
    if (e == hipSuccess) *n_out = (int64_t)(last_pos + last_flag);
    return e;
}

extern "C" hipError_t gs_launch_build_scatter(const u64 *keys, const int32_t *value, const uint32_t *flag, const u64 *pos, int64_t n,
                                              int64_t *out_keys, int32_t *out_vals, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gs_build_scatter_kernel, dim3(gs_build_grid(n)), dim3(256), 0, stream, keys, value, flag, pos, n, out_keys, out_vals);
    return hipGetLastError();
}
