// gs_build.hip -- DB construction on the device (include/gsgpu.h, gs_dbbuild_*): the compute core of the reference's
// FillDBGoal + DBGoal (C/goals/refseq/FillDBGoal.java:297-end, C/goals/refseq/DBGoal.java:188-311 over
// C/refseq/AbstractStoreFastaReader.java:87-115 and C/util/CGATLongBuffer.java:137-229):
//   every k-mer of every genome region -> (canonical k-mer, region) pairs        gs_build_kmers_kernel
//   all pairs sorted by k-mer                                                    rocPRIM radix sort over the 2k key bits
//   per distinct k-mer: stored iff a FILL region holds it (KMerSortedArray.putLong, first writer), value = the node of
//   the first such region, then the lowest common ancestor with the node of every UPDATE region that holds it
//   (KMerStore.update + TaxTree.getLowestCommonAncestor, C/tax/TaxTree.java:160-187)       gs_build_reduce_kernel
//   compaction of the stored k-mers in ascending order                                      scan + scatter
// The reference walks the genomes twice with a hash-free sorted array and a Bloom filter in front of it; here one sort
// replaces both walks.  Byte / integer work, HBM-bound (12 bytes per genome base through a radix sort of 2k / 8 passes).
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

// bit i of v -> bit 2 i
__device__ __forceinline__ u64 gs_build_spread(uint32_t v) {
    u64 x = v;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFULL;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFULL;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0FULL;
    x = (x | (x << 2)) & 0x3333333333333333ULL;
    x = (x | (x << 1)) & 0x5555555555555555ULL;
    return x;
}

// The reference's low-complexity score of a k-mer (CGATLongBuffer.getDustValue, C/util/CGATLongBuffer.java:146-229; its test
// T/util/CGATLongBufferTest.java:280-313 states it for a window): for the periods 1, 2, 3 every maximal run of L consecutive
// positions whose base equals the base `period` earlier adds fib(L), fib = 0, 1, 2, 3, 5, 8 ...  On planes: one mask of matches
// per period, then a walk over its runs of ones.
__device__ __forceinline__ int gs_build_dust(uint32_t fhi, uint32_t flo, int k) {
    int d = 0;
    for (int p = 1; p <= 3 && p < k; p++) {
        uint32_t m = ~((fhi ^ (fhi >> p)) | (flo ^ (flo >> p))) & ((1u << (k - p)) - 1u);
        while (m) {
            m >>= __builtin_ctz(m);
            const int len = __builtin_ctz(~m);  // (m < 2^31: a zero bit always follows)
            int a = 1, b = 2;                   // fib(1), fib(2)
            for (int i = 1; i < len; i++) {
                const int c = a + b;
                a = b;
                b = c;
            }
            d += a;
            m >>= len;
        }
    }
    return d;
}

// k bits of the 128-bit string {b (high), a (low)} from bit s (s in 0..63, k <= 31)
__device__ __forceinline__ uint32_t gs_build_funnel(u64 a, u64 b, int s, uint32_t kmask) {
    return (uint32_t)((a >> s) | ((b << 1) << (63 - s))) & kmask;
}

// One wave per tile of 64 consecutive base positions of the concatenated regions; lane = the k-mer that STARTS at position
// p0 + lane.  The tile's 64 + k - 1 bytes become three ballot planes (code bit 1, code bit 0, "not a base"), a lane cuts
// its k-mer out of the planes with a funnel shift and interleaves them into the reference's encoding (first base in the top
// bits) -- ~40 integer instructions per k-mer instead of a 31-step loop over bytes.  The reference's ring buffer is reset by
// a non-base and at a region start (CGATLongBuffer.put :141-144, AbstractStoreFastaReader.startRegion) and emits the window of
// the last k bases whenever it is filled and (bases of the region so far) % stepSize == 0 (:103-104): the k-mer over region
// bases [s, s + k) is taken iff all k are bases and (s + k) % stepSize == 0.  Only real pairs are written (compacted per wave).
#define GS_BUILD_BLOCK 1024  // 16 waves: one atomic on the pair counter per workgroup and step
__global__ __launch_bounds__(GS_BUILD_BLOCK) void gs_build_kmers_kernel(const uint8_t *seq, const u64 *off, int64_t n_regions, int64_t total, int k,
                                                             int lower, int step, int max_dust, uint32_t first_region, uint32_t update_flag,
                                                             u64 range_lo, u64 range_hi, u64 *keys, uint32_t *vals, u64 *n_out) {
    constexpr int WAVES = GS_BUILD_BLOCK / 64;
    __shared__ uint32_t s_cnt[WAVES];
    __shared__ u64 s_base;
    const int lane = (int)(threadIdx.x & 63), wib = (int)(threadIdx.x >> 6);
    const uint32_t kmask = (uint32_t)((1ULL << k) - 1);
    const int64_t n_tiles = (total + 63) >> 6;
    // (the loop is uniform over the workgroup: its waves take WAVES consecutive tiles per step and meet at the barriers)
    for (int64_t tile0 = (int64_t)blockIdx.x * WAVES; tile0 < n_tiles; tile0 += (int64_t)gridDim.x * WAVES) {
        const int64_t tile = tile0 + wib;
        const int64_t p0 = tile << 6, p = p0 + lane;  // (tile >= n_tiles: p >= total, every lane idles)
        // bytes p0 .. p0 + 63 and p0 + 64 .. p0 + 64 + k - 2
        const uint32_t c0 = p < total ? gs_build_code(seq[p], lower) : 4u;
        const uint32_t c1 = (lane < k - 1 && p + 64 < total) ? gs_build_code(seq[p + 64], lower) : 4u;
        const u64 hi0 = __ballot((c0 >> 1) & 1u), lo0 = __ballot(c0 & 1u), bad0 = __ballot(c0 >> 2);
        const u64 hi1 = __ballot((c1 >> 1) & 1u), lo1 = __ballot(c1 & 1u), bad1 = __ballot(c1 >> 2);
        // region of the tile's first position: the last r with off[r] <= p0 (the same search on every lane: scalar loads)
        int64_t lo = 0, hi = n_regions;  // invariant: off[lo] <= p0 < off[hi]
        const u64 p0c = p0 < total ? (u64)p0 : (u64)(total - 1);
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (off[mid] <= p0c)
                lo = mid;
            else
                hi = mid;
        }
        int64_t r = lo;
        u64 r_begin = off[lo], r_end = off[lo + 1];
        if (p0 < total && r_end < (u64)(p0 + 64 + k - 1) && r_end < (u64)total) {  // a region ends inside the tile: every lane finds its own
            int64_t l2 = lo, h2 = n_regions;
            while (h2 - l2 > 1) {
                const int64_t mid = (l2 + h2) >> 1;
                if (off[mid] <= (u64)p)
                    l2 = mid;
                else
                    h2 = mid;
            }
            r = l2;
            r_begin = off[l2];
            r_end = off[l2 + 1];
        }
        u64 key = GS_BUILD_NONE;
        const uint32_t wbad = gs_build_funnel(bad0, bad1, lane, kmask);
        const int64_t s_in = p - (int64_t)r_begin;
        if (p < total && wbad == 0 && (u64)p + (u64)k <= r_end && (s_in + k) % step == 0) {
            // planes: bit i = base p + i.  Reference encoding: base p in the top bit pair.
            const uint32_t fhi = gs_build_funnel(hi0, hi1, lane, kmask), flo = gs_build_funnel(lo0, lo1, lane, kmask);
            const uint32_t rhi = __brev(fhi) >> (32 - k), rlo = __brev(flo) >> (32 - k);
            const u64 fwd = (gs_build_spread(rhi) << 1) | gs_build_spread(rlo);
            const u64 rev = (gs_build_spread(fhi) << 1) | gs_build_spread((flo ^ kmask) & kmask);  // complement: C<->G, A<->T, reversed
            key = fwd > rev ? fwd : rev;  // CGAT.standardKMer (:145-147)
            if (key < range_lo || key >= range_hi) key = GS_BUILD_NONE;  // (gs_dbbuild_set_range: another pass takes it)
            if (max_dust >= 0 && key != GS_BUILD_NONE && gs_build_dust(fhi, flo, k) > max_dust) key = GS_BUILD_NONE;  // isDust(): skipped (:105-107)
        }
        // the workgroup's pairs go behind each other at the end of the pair arrays: one atomic per workgroup and step
        const u64 have = __ballot(key != GS_BUILD_NONE);
        if (lane == 0) s_cnt[wib] = (uint32_t)__popcll(have);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t sum = 0;
            for (int w = 0; w < WAVES; w++) sum += s_cnt[w];
            s_base = sum ? atomicAdd(n_out, (u64)sum) : 0;
        }
        __syncthreads();
        if (key != GS_BUILD_NONE) {
            u64 at = s_base + (u64)__popcll(have & ((1ULL << lane) - 1));
            for (int w = 0; w < wib; w++) at += s_cnt[w];
            keys[at] = key;
            vals[at] = update_flag | (first_region + (uint32_t)r);
        }
        __syncthreads();  // (s_cnt / s_base are rewritten by the next step)
    }
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
static int gs_build_tile_grid(int64_t total) {  // 16 waves per workgroup, one tile of 64 positions per wave and step
    int64_t g = ((total + 63) / 64 + 15) / 16;
    if (g > 4096) g = 4096;
    return g < 1 ? 1 : (int)g;
}

// keys / vals: room for `total` more pairs behind *n_out (device counter of the pairs written so far)
extern "C" hipError_t gs_launch_build_kmers(const uint8_t *seq, const u64 *off, int64_t n_regions, int64_t total, int k, int lower, int step,
                                            int max_dust, uint32_t first_region, int update, u64 range_lo, u64 range_hi, u64 *keys, uint32_t *vals,
                                            u64 *n_out, hipStream_t stream) {
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(gs_build_kmers_kernel, dim3(gs_build_tile_grid(total)), dim3(GS_BUILD_BLOCK), 0, stream, seq, off, n_regions, total, k, lower, step,
                       max_dust, first_region, update ? GS_BUILD_UPDATE : 0u, range_lo, range_hi, keys, vals, n_out);
    return hipGetLastError();
}

// keys / vals: n pairs, sorted in place through the alternate buffers (keys_alt / vals_alt: n each).  Returns the sorted arrays
// in *keys_out / *vals_out (one of the two buffers each).
extern "C" hipError_t gs_build_sort(u64 *keys, u64 *keys_alt, uint32_t *vals, uint32_t *vals_alt, int64_t n, int key_bits, u64 **keys_out,
                                    uint32_t **vals_out, hipStream_t stream) {
    *keys_out = keys;
    *vals_out = vals;
    if (n <= 1) return hipSuccess;
    rocprim::double_buffer<u64> dk(keys, keys_alt);
    rocprim::double_buffer<uint32_t> dv(vals, vals_alt);
    size_t tmp_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, dk, dv, (size_t)n, 0, (unsigned)key_bits, stream);
    if (e != hipSuccess) return e;
    void *tmp = nullptr;
    e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1);
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_pairs(tmp, tmp_bytes, dk, dv, (size_t)n, 0, (unsigned)key_bits, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    hipFree(tmp);
    *keys_out = dk.current();
    *vals_out = dv.current();
    return e;
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


// ---------------------------------------------------------------------------------------------------
// Index filter construction (BloomIndexGoal, C/goals/refseq/BloomIndexGoal.java:66-113): AbstractKMerBloomFilter.putLong
// (C/bloom/AbstractKMerBloomFilter.java:196-203) of every k-mer with the XOR hashes (C/bloom/XORKMerBloomFilter.java:43-59: bit
// abs((factor_i ^ kmer) % bits) for every hash i) or the Murmur ones (MurmurKMerBloomFilter).  One thread per (k-mer, hash).
// ---------------------------------------------------------------------------------------------------
// MurmurKMerBloomFilter.hash: MurmurHash3DropIn.hash64(kmer, factor) (C/util/MurmurHash3DropIn.java:60-87)
__device__ __forceinline__ int64_t gs_build_murmur64(int64_t data_, int64_t base) {
    const u64 data = (u64)data_;
    u64 hash = (u64)base;
    u64 kk = __builtin_bswap64(data);
    kk *= 0x87c37b91114253d5ULL;
    kk = (kk << 31) | (kk >> 33);
    kk *= 0x4cf5ad432745937fULL;
    hash ^= kk;
    hash = ((hash << 27) | (hash >> 37)) * 5 + 0x52dce729ULL;
    hash ^= 8;
    hash ^= hash >> 33;
    hash *= 0xff51afd7ed558ccdULL;
    hash ^= hash >> 33;
    hash *= 0xc4ceb9fe1a85ec53ULL;
    hash ^= hash >> 33;
    return (int64_t)(hash ^ data);
}

__global__ __launch_bounds__(256) void gs_bloom_xor_put_kernel(const int64_t *keys, int64_t n, int64_t bits, const int64_t *factors, int n_hashes,
                                                              int murmur, u64 *words) {
    const int64_t total = n * (int64_t)n_hashes;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t f = factors[i % n_hashes], key = keys[i / n_hashes];
        const int64_t r = (murmur ? gs_build_murmur64(key, f) : (f ^ key)) % bits;  // (truncating remainder, as in Java)
        const u64 b = (u64)(r < 0 ? -r : r);
        const u64 m = 1ULL << (b & 63);
        if ((words[b >> 6] & m) == 0) atomicOr(&words[b >> 6], m);
    }
}

extern "C" hipError_t gs_launch_bloom_xor_put(const int64_t *keys, int64_t n, int64_t bits, const int64_t *factors, int n_hashes, int murmur,
                                              u64 *words, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gs_bloom_xor_put_kernel, dim3(gs_build_grid(n * (int64_t)n_hashes)), dim3(256), 0, stream, keys, n, bits, factors, n_hashes,
                       murmur, words);
    return hipGetLastError();
}
