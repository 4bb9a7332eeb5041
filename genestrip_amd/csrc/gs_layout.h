// gs_layout.h -- HBM layout of the device k-mer store, shared by the host builder and the kernels.
//
// Key encoding ("planar"): for a k-mer s_0..s_{k-1} with the reference's 2-bit codes (C=0 G=1 A=2 T=3,
// C/util/CGAT.java:66-69) the key is  (hi << 31) | lo  where bit i of hi/lo is the high/low code bit of
// base i.  A wave builds it from two 64-lane ballots; reverse complement = bit-reverse both planes and
// flip lo (complement is code^1).  The canonical orientation is chosen exactly as the reference does
// (numeric max of the interleaved encodings = lexicographic max from base 0, CGAT.java:145-147), so the
// set of reachable keys is identical.
//
// Table: n_buckets (power of two) buckets of 8 x uint64 slots = one 64-byte line per probe.
//   h      = gs_mix62(key)                 bijection on [0, 2^62)
//   bucket = h & (n_buckets-1),  rem = h >> bucket_bits
//   slot   = rem << (vbits+3) | disp << (vbits+1) | (value_index+1) << 1 | seen        (0 = empty)
// `seen` is the unique-k-mer mark of the running match (KMerUniqueCounterBits): set with one atomicOr by the first
// wave that hits the slot with a clear bit; every later probe reads it for free with the slot itself.
// An entry lives in bucket (home + disp) & mask, disp in 0..3; it is displaced only past FULL buckets, so
// a probe walks home, home+1, .. while the bucket it just read is full and holds no match.
// For the multi-GPU merge the seen bits are extracted into a compact bitmap (bit index = bucket*8 + slot).
#pragma once
#include <stdint.h>

#define GS_SLOTS_PER_BUCKET 8
#define GS_MAX_DISP 3
#define GS_KEY_BITS 62
#define GS_PLANE_SHIFT 31

#if defined(__HIPCC__)
#define GS_HD __host__ __device__ __forceinline__
#else
#define GS_HD static inline
#endif

GS_HD uint64_t gs_mix62(uint64_t x) {
    const uint64_t M = (1ULL << GS_KEY_BITS) - 1;
    x ^= x >> 31;
    x = (x * 0x7fb5d329728ea185ULL) & M;
    x ^= x >> 27;
    x = (x * 0x81dadef4bc2dd44dULL) & M;
    x ^= x >> 33;
    return x;
}

// Gate ("is this k-mer possibly in the store?"): a word-blocked Bloom filter, 4 bits per key inside one 64-bit
// word, sized to stay resident in each XCD's 4 MiB L2 (random 8-byte reads from <= 4 MiB run at ~250 G/s on
// MI355X against ~59 G/s for 64-byte lines of a 64 MiB table: tools/probe_bw.hip).  It plays the role of the
// reference's Bloom pre-filter in KMerStore.getLong (C/store/KMerSortedArray.java:299-301): no false
// negatives, so results are unchanged; most misses never touch the table.  Only built when it fits.
//   word = gate[(h >> bucket_bits) & gate_mask],  bits = 4 x 6-bit fields of h >> 38
#define GS_GATE_FIELD_SHIFT 38

// DB-partitioned stores (config 5): rank (h >> GS_OWNER_SHIFT) % n_parts owns the key
#define GS_OWNER_SHIFT 40

GS_HD uint64_t gs_gate_field_bits(uint32_t f) {  // f = h >> GS_GATE_FIELD_SHIFT (24 bits)
    return (1ULL << (f & 63)) | (1ULL << ((f >> 6) & 63)) | (1ULL << ((f >> 12) & 63)) | (1ULL << ((f >> 18) & 63));
}

GS_HD uint64_t gs_gate_bits(uint64_t h) { return gs_gate_field_bits((uint32_t)(h >> GS_GATE_FIELD_SHIFT)); }

// Minimizer gate: a second, much smaller filter over the MINIMIZERS of the stored k-mers.  The minimizer of a k-mer
// is the canonical 15-mer with the smallest hash among its k-14 15-mers (strand symmetric).  A k-mer can only be in
// the store if its minimizer is in this set, and consecutive k-mers of a read share their minimizer for ~(k-13)/2
// positions, so the 64 lanes of a wave ask for only ~8 distinct words: the lookups coalesce to a handful of requests
// instead of one per k-mer.  Used by the fused kernels when k >= GS_MIN_K; the word gate above stays for the
// key-only probe of the DB-partitioned mode.
#define GS_MIN_L 15
#define GS_MIN_K 19

GS_HD uint32_t gs_lmer_hash(uint32_t fh, uint32_t fl) {  // 15-bit planes of a 15-mer (base 0 in bit 0) -> order hash
    const uint32_t M = (1u << GS_MIN_L) - 1u;
    const uint32_t rh = __builtin_bitreverse32(fh) >> (32 - GS_MIN_L);
    const uint32_t rl = (__builtin_bitreverse32(fl) >> (32 - GS_MIN_L)) ^ M;
    const uint32_t f = (fh << GS_MIN_L) | fl, r = (rh << GS_MIN_L) | rl;
    uint32_t g = f < r ? f : r;  // canonical 15-mer (30 bits); the mix below is a bijection on 32 bits
    g *= 0x9E3779B1u;
    g ^= g >> 15;
    g *= 0x85EBCA77u;
    g ^= g >> 13;
    return g;
}

GS_HD uint64_t gs_mgate_bits(uint32_t m) {  // 3 bits inside one 64-bit word
    uint32_t x = m * 0xC2B2AE35u;
    x ^= x >> 16;
    return (1ULL << (x & 63)) | (1ULL << ((x >> 6) & 63)) | (1ULL << ((x >> 12) & 63));
}

struct GsDbDev {
    const unsigned long long *table;  // n_buckets * 8 slots
    const unsigned long long *gate;   // gate_mask+1 words, or nullptr
    uint64_t gate_mask;
    const unsigned long long *mgate;  // minimizer gate: mgate_mask+1 words, or nullptr
    uint64_t mgate_mask;
    uint32_t bucket_bits;
    uint32_t vbits;
    uint64_t bucket_mask;
    int32_t k;
    int32_t n_values;
    // per value index (tree node) arrays; tin/tout = pre-order interval, depth root = 0
    const int32_t *parent;
    const int32_t *depth;
    const int32_t *tin;
    const int32_t *tout;
};
