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
