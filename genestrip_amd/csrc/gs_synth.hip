// gs_synth.hip -- deterministic synthetic read generator, NOT part of the hot path:
// bench.py and the tests use it to produce the same reads on the host (for the CPU oracle) and directly in
// HBM (so that a 10 M-read batch never crosses PCIe).  Every byte is a pure function of
// (seed, global read index, base index) through a counter-based splitmix64 hash, so host and device agree.
//
// Read recipe (L bases, fixed length): 50 % sampled from a DB genome (uniform species, position, strand)
// with 1 % i.i.d. substitutions, 50 % uniform ACGT background; 0.1 % of the reads get one 'N'.
#include <hip/hip_runtime.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define SY_HD __host__ __device__ __forceinline__
#else
#define SY_HD static inline
#endif

SY_HD uint64_t sy_hash(uint64_t seed, uint64_t read, uint64_t field) {
    uint64_t z = seed + read * 0x9E3779B97F4A7C15ULL + field * 0xD1B54A32D192ED03ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

SY_HD uint8_t sy_comp(uint8_t c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c; }

// base j of read g
SY_HD uint8_t sy_base(uint64_t seed, uint64_t g, int j, int L, const uint8_t *genomes, int n_genomes, int genome_len) {
    const char ACGT[4] = {'A', 'C', 'G', 'T'};
    const uint64_t hn = sy_hash(seed, g, 4);
    if (hn % 1000 == 0 && (int)((hn >> 16) % (uint64_t)L) == j) return 'N';
    const uint64_t hb = sy_hash(seed, g, 16 + (uint64_t)j);
    if (sy_hash(seed, g, 0) & 1) return (uint8_t)ACGT[hb & 3];  // background read
    const int s = (int)(sy_hash(seed, g, 1) % (uint64_t)n_genomes);
    const int pos = (int)(sy_hash(seed, g, 2) % (uint64_t)(genome_len - L + 1));
    const bool rc = sy_hash(seed, g, 3) & 1;
    uint8_t c = rc ? sy_comp(genomes[(size_t)s * genome_len + pos + (L - 1 - j)]) : genomes[(size_t)s * genome_len + pos + j];
    if (hb % 100 == 0) {  // substitution: one of the three other bases
        int idx = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3;
        c = (uint8_t)ACGT[(idx + 1 + (int)((hb >> 8) % 3)) & 3];
    }
    return c;
}

__global__ void sy_reads_kernel(uint64_t seed, uint64_t first, int64_t n_reads, int L, const uint8_t *genomes, int n_genomes,
                                int genome_len, uint8_t *seq, uint64_t *off) {
    const int64_t total = n_reads * (int64_t)L;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / L;
        const int j = (int)(i - r * L);
        seq[i] = sy_base(seed, first + (uint64_t)r, j, L, genomes, n_genomes, genome_len);
    }
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n_reads; r += (int64_t)gridDim.x * blockDim.x)
        off[r] = (uint64_t)r * (uint64_t)L;
}

extern "C" void gs_synth_reads_host(uint64_t seed, uint64_t first, int64_t n_reads, int L, const uint8_t *genomes,
                                    int n_genomes, int genome_len, uint8_t *seq, uint64_t *off) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n_reads; r++)
        for (int j = 0; j < L; j++) seq[r * L + j] = sy_base(seed, first + (uint64_t)r, j, L, genomes, n_genomes, genome_len);
    for (int64_t r = 0; r <= n_reads; r++) off[r] = (uint64_t)r * (uint64_t)L;
}

// genomes/seq/off are device pointers; runs on the current device's default stream and synchronises
extern "C" int gs_synth_reads_device(uint64_t seed, uint64_t first, int64_t n_reads, int L, const uint8_t *genomes,
                                     int n_genomes, int genome_len, uint8_t *seq, uint64_t *off) {
    hipLaunchKernelGGL(sy_reads_kernel, dim3(4096), dim3(256), 0, 0, seed, first, n_reads, L, genomes, n_genomes, genome_len,
                       seq, off);
    hipError_t e = hipDeviceSynchronize();
    return e == hipSuccess ? 0 : -(int)e;
}

// ---------------------------------------------------------------------------------------------------
// XOR index filter as BloomIndexGoal builds it (input manufacture for the filter workloads at configs[2] scale; the
// tests check this builder against the CPU oracle's filter on small inputs).  Every key sets n_hashes bits:
// bit index = abs((factor[i] ^ key) % bits) with Java's truncated remainder (C/bloom/XORKMerBloomFilter.java:43-59,
// AbstractKMerBloomFilter.java:193-203), bit b lives in word b >> 6 at position b & 63 (LargeBitVector).
// ---------------------------------------------------------------------------------------------------
SY_HD uint64_t sy_xor_bit(int64_t key, int64_t factor, int64_t bits) {
    const int64_t r = (factor ^ key) % bits;
    return (uint64_t)(r < 0 ? -r : r);
}

__global__ void sy_bloom_xor_kernel(const int64_t *keys, int64_t n, int64_t bits, const int64_t *factors, int n_hashes,
                                    unsigned long long *words) {
    const int64_t total = n * (int64_t)n_hashes;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t key = keys[i / n_hashes];
        const uint64_t b = sy_xor_bit(key, factors[i % n_hashes], bits);
        atomicOr(&words[b >> 6], 1ULL << (b & 63));
    }
}

extern "C" void gs_synth_bloom_xor_host(const int64_t *keys, int64_t n, int64_t bits, const int64_t *factors, int n_hashes,
                                        uint64_t *words) {
    for (int64_t i = 0; i < n; i++)
        for (int h = 0; h < n_hashes; h++) {
            const uint64_t b = sy_xor_bit(keys[i], factors[h], bits);
            words[b >> 6] |= 1ULL << (b & 63);
        }
}

// keys / factors / words are device pointers; words must be zeroed by the caller; synchronises
extern "C" int gs_synth_bloom_xor_device(const int64_t *keys, int64_t n, int64_t bits, const int64_t *factors, int n_hashes,
                                         uint64_t *words) {
    hipLaunchKernelGGL(sy_bloom_xor_kernel, dim3(8192), dim3(256), 0, 0, keys, n, bits, factors, n_hashes,
                       (unsigned long long *)words);
    hipError_t e = hipDeviceSynchronize();
    return e == hipSuccess ? 0 : -(int)e;
}

// ---------------------------------------------------------------------------------------------------
// Synthetic store builder (SynthDB in genestrip_amd/synth.py): canonical k-mers of every species genome, each stored
// with the LCA of the species containing it (mirrors FillDBGoal + DBGoal's LCA update, C/goals/refseq/DBGoal.java:233-256,
// for the two-level genus/species tree of the recipe).  Same result as the numpy path, on all host cores: the 47 M-k-mer
// store of configs[2..3] scale takes seconds instead of a minute.
// ---------------------------------------------------------------------------------------------------
#include <algorithm>
#include <vector>

namespace {
struct SyKv {
    uint64_t k;
    int32_t v;
};
struct SyDb {
    std::vector<int64_t> kmers;
    std::vector<int32_t> vals;
};
inline uint32_t sy_code(uint8_t c) { return c == 'C' ? 0u : c == 'G' ? 1u : c == 'A' ? 2u : 3u; }  // CGAT.java:66-69
}  // namespace

// genomes: n_species x genome_len upper-case ACGT; species_vi[n_species]; parent_vi[n_values] (root = value 0).
// Returns a handle (NULL on failure); *n_out = number of distinct canonical k-mers.
extern "C" void *gs_synth_db_build(const uint8_t *genomes, int n_species, int genome_len, int k, const int32_t *species_vi,
                                   const int32_t *parent_vi, int64_t *n_out) try {
    const int nb_bits = 10, nb = 1 << nb_bits;
    const int top_shift = 2 * k > nb_bits ? 2 * k - nb_bits : 0;
    const uint64_t mask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
    const int n = genome_len - k + 1;
    std::vector<std::vector<uint64_t>> per((size_t)n_species);
#pragma omp parallel for schedule(dynamic, 1)
    for (int s = 0; s < n_species; s++) {
        std::vector<uint64_t> &u = per[(size_t)s];
        if (n <= 0) continue;
        u.resize((size_t)n);
        const uint8_t *g = genomes + (size_t)s * genome_len;
        uint64_t fwd = 0, rev = 0;
        for (int j = 0; j < genome_len; j++) {
            const uint64_t c = sy_code(g[j]);
            fwd = ((fwd << 2) | c) & mask;                          // CGAT.nextKMerStraight :208-214
            rev = (rev >> 2) | ((c ^ 1ULL) << (2 * (k - 1)));       // CGAT.nextKMerReverse :226-232
            if (j >= k - 1) u[(size_t)(j - k + 1)] = fwd > rev ? fwd : rev;  // standardKMer :145-147
        }
        std::sort(u.begin(), u.end());
        u.erase(std::unique(u.begin(), u.end()), u.end());
    }
    // bucket by the top bits, so that the buckets can be sorted side by side and concatenated in order
    std::vector<int64_t> cnt((size_t)nb + 1, 0);
    for (int s = 0; s < n_species; s++)
        for (uint64_t x : per[(size_t)s]) cnt[(size_t)(x >> top_shift) + 1]++;
    for (int b = 0; b < nb; b++) cnt[(size_t)b + 1] += cnt[(size_t)b];
    std::vector<SyKv> all((size_t)cnt[(size_t)nb]);
    {
        std::vector<int64_t> cur(cnt.begin(), cnt.end() - 1);
        for (int s = 0; s < n_species; s++) {
            for (uint64_t x : per[(size_t)s]) all[(size_t)cur[(size_t)(x >> top_shift)]++] = SyKv{x, species_vi[s]};
            std::vector<uint64_t>().swap(per[(size_t)s]);
        }
    }
    std::vector<int64_t> out_cnt((size_t)nb + 1, 0);
#pragma omp parallel for schedule(dynamic, 4)
    for (int b = 0; b < nb; b++) {
        SyKv *lo = all.data() + cnt[(size_t)b], *hi = all.data() + cnt[(size_t)b + 1];
        std::sort(lo, hi, [](const SyKv &a, const SyKv &c) { return a.k < c.k || (a.k == c.k && a.v < c.v); });
        // in place: one entry per distinct k-mer with the LCA of its smallest / largest species (pre-order value indices:
        // the species of one genus are contiguous)
        SyKv *w = lo;
        for (SyKv *p = lo; p < hi;) {
            SyKv *q = p;
            while (q + 1 < hi && q[1].k == p->k) q++;
            const int32_t vmin = p->v, vmax = q->v;
            const int32_t val = vmin == vmax ? vmin : (parent_vi[vmin] == parent_vi[vmax] ? parent_vi[vmin] : 0);
            *w++ = SyKv{p->k, val};
            p = q + 1;
        }
        out_cnt[(size_t)b + 1] = w - lo;
    }
    for (int b = 0; b < nb; b++) out_cnt[(size_t)b + 1] += out_cnt[(size_t)b];
    SyDb *db = new SyDb();
    db->kmers.resize((size_t)out_cnt[(size_t)nb]);
    db->vals.resize((size_t)out_cnt[(size_t)nb]);
#pragma omp parallel for schedule(dynamic, 4)
    for (int b = 0; b < nb; b++) {
        const SyKv *lo = all.data() + cnt[(size_t)b];
        const int64_t m = out_cnt[(size_t)b + 1] - out_cnt[(size_t)b];
        for (int64_t i = 0; i < m; i++) {
            db->kmers[(size_t)(out_cnt[(size_t)b] + i)] = (int64_t)lo[i].k;
            db->vals[(size_t)(out_cnt[(size_t)b] + i)] = lo[i].v;
        }
    }
    *n_out = (int64_t)db->kmers.size();
    return db;
} catch (...) {
    return nullptr;
}

extern "C" void gs_synth_db_fetch(void *handle, int64_t *kmers, int32_t *vals) {
    SyDb *db = (SyDb *)handle;
    std::copy(db->kmers.begin(), db->kmers.end(), kmers);
    std::copy(db->vals.begin(), db->vals.end(), vals);
    delete db;
}
