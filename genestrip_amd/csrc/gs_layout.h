// gs_layout.h -- HBM layout of the device k-mer store, shared by the host builder and the kernels.
//
// Key encoding ("planar"): for a k-mer s_0..s_{k-1} with the reference's 2-bit codes (C=0 G=1 A=2 T=3,
// C/util/CGAT.java:66-69) the key is  (hi << 31) | lo  where bit i of hi/lo is the high/low code bit of
// base i.  A wave builds it from two 64-lane ballots; reverse complement = bit-reverse both planes and
// flip lo (complement is code^1).  The reference looks a k-mer up under max(fwd, revcomp) of its interleaved
// encoding (CGAT.java:145-147); the table files the same k-mer under the orientation whose (hi, lo) plane pair is
// the larger one (gs_rep_planes: one 64-bit compare on the register pair).  Both rules pick one representative
// per {k-mer, reverse complement} class, and the builder converts every reachable stored k-mer to that
// representative, so hit/miss and values are identical.
//
// Table: n_buckets (power of two) buckets of 8 x uint64 slots = one 64-byte line per probe.
//   h      = gs_mix_planes(hi, lo)         bijection on [0, 2^62)
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

GS_HD uint32_t gs_brev32(uint32_t x) { return __builtin_bitreverse32(x); }

// representative orientation of the k-mer with forward planes (fhi, flo): planes of the larger (hi:lo) pair
GS_HD void gs_rep_planes(uint32_t fhi, uint32_t flo, int k, uint32_t kmask, uint32_t &a, uint32_t &b) {
    const uint32_t rhi = gs_brev32(fhi) >> (32 - k);
    const uint32_t rlo = (gs_brev32(flo) >> (32 - k)) ^ kmask;
    const bool fwd = (((uint64_t)fhi << 32) | flo) >= (((uint64_t)rhi << 32) | rlo);
    a = fwd ? fhi : rhi;
    b = fwd ? flo : rlo;
}

// 31-bit round function of the Feistel network below: both halves of the 32x32 -> 64-bit product folded together,
// so every output bit depends on every input bit (one v_mad_u64_u32 + xor + shift)
GS_HD uint32_t gs_fold31(uint32_t x, uint32_t c) {
    const uint64_t p = (uint64_t)x * c;
    return ((uint32_t)p ^ (uint32_t)(p >> 32)) >> 1;
}

// (hi, lo) planes, 31 bits each -> h in [0, 2^62).  A three-round Feistel network is a bijection whatever the round
// function is, so (bucket, remainder) identifies the key exactly and the slots need no full key.
GS_HD uint64_t gs_mix_planes(uint32_t a, uint32_t b) {
    a ^= gs_fold31(b, 0x9E3779B1u);
    b ^= gs_fold31(a, 0x85EBCA77u);
    a ^= gs_fold31(b, 0xC2B2AE3Du);
    return ((uint64_t)a << GS_PLANE_SHIFT) | b;
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
// instead of one per k-mer.  32-bit words, two bits per entry, 16-32 bits per distinct minimizer.  Used by the fused
// kernels when k >= GS_MIN_K; the word gate above stays for the key-only probe of the DB-partitioned mode.
#define GS_MIN_L 15
#define GS_MIN_K 19

// 15-bit planes of a 15-mer (base 0 in bit 0) -> (canonical 15-mer << 1) | (1 if the given strand is the canonical one).
// A 15-mer is never its own reverse complement (odd length).
GS_HD uint32_t gs_lmer_canon(uint32_t fh, uint32_t fl) {
    const uint32_t M = (1u << GS_MIN_L) - 1u;
    const uint32_t rh = gs_brev32(fh) >> (32 - GS_MIN_L);
    const uint32_t rl = (gs_brev32(fl) >> (32 - GS_MIN_L)) ^ M;
    const uint32_t f = (fh << GS_MIN_L) | fl, r = (rh << GS_MIN_L) | rl;
    return f < r ? (f << 1) | 1u : r << 1;
}
// order hash of a canonical 15-mer (30 bits): odd multiplier = a bijection on 32 bits, the order is set by the well-mixed top bits
GS_HD uint32_t gs_canon_hash(uint32_t g) { return g * 0x9E3779B1u; }
GS_HD uint32_t gs_lmer_hash(uint32_t fh, uint32_t fl) { return gs_canon_hash(gs_lmer_canon(fh, fl) >> 1); }

// Which 15-mer of a k-mer is "its" minimizer: the occurrence with the smallest RANK = order hash with the low 8 bits
// replaced by the position (a wave finds minimum and position with one min3 chain over its LDS row), i.e. the smallest
// hash by its top 24 bits, leftmost on a tie.  "Leftmost" depends on the strand a k-mer is read from, so a stored k-mer
// can have two choices (the same 15-mer twice inside the k-mer, or two 15-mers whose hashes agree in 24 bits: ~1e-5 of
// the k-mers): the builder evaluates BOTH strand views of every stored k-mer (gs_choose_minimizer on the planes and on
// their reverse complement) and files the k-mer under each, so whatever strand a read shows, the probe's choice is one
// the builder has seen.  Everything keyed by the minimizer (gate, record bucket) uses the EXACT canonical 15-mer of the
// chosen occurrence (gs_min_oriented), never the truncated rank.
GS_HD uint32_t gs_lmer_rank(uint32_t hash, uint32_t idx) { return (hash & 0xffffff00u) | idx; }

GS_HD int gs_choose_minimizer(uint32_t hi, uint32_t lo, int k) {  // host / reference form of the wave's min3 chain
    uint32_t best = 0xffffffffu;
    for (int d = 0; d + GS_MIN_L <= k; d++) {
        const uint32_t x = gs_lmer_rank(gs_lmer_hash((hi >> d) & 0x7fffu, (lo >> d) & 0x7fffu), (uint32_t)d);
        best = x < best ? x : best;
    }
    return (int)(best & 0xffu);
}

// The chosen occurrence (offset p inside the k-mer with forward planes fhi/flo and reverse-complement planes rhi/rlo):
// gh = order hash of its canonical 15-mer (a bijection of the 15-mer: the key of gate and record bucket), and the k-mer
// in the orientation in which that 15-mer is the canonical one (ohi, olo), where it sits at offset (k-15) - j.
// cf = gs_lmer_canon of the chosen 15-mer (the wave reads it back from its LDS row instead of recomputing it)
GS_HD void gs_min_oriented_cf(uint32_t cf, uint32_t fhi, uint32_t flo, uint32_t rhi, uint32_t rlo, int k, int p, uint32_t &gh,
                              uint32_t &ohi, uint32_t &olo, int &j) {
    const bool fwd = (cf & 1u) != 0;
    gh = gs_canon_hash(cf >> 1);
    ohi = fwd ? fhi : rhi;
    olo = fwd ? flo : rlo;
    j = fwd ? (k - GS_MIN_L) - p : p;
}
GS_HD void gs_min_oriented(uint32_t fhi, uint32_t flo, uint32_t rhi, uint32_t rlo, int k, int p, uint32_t &gh,
                           uint32_t &ohi, uint32_t &olo, int &j) {
    const uint32_t M = (1u << GS_MIN_L) - 1u;
    gs_min_oriented_cf(gs_lmer_canon((fhi >> p) & M, (flo >> p) & M), fhi, flo, rhi, rlo, k, p, gh, ohi, olo, j);
}

// Super-k-mer records (fused kernels, k >= GS_MIN_K): the k-mers of a read that share a minimizer occurrence are ~9
// consecutive positions, and in the store they are the substrings of ONE window of 2k-15 bases around that 15-mer.  A
// record is one 64-byte line holding such a window in the minimizer's canonical orientation:
//   w0 = window code-hi plane (47 bits) | seen bits  << 47      seen[j]: unique-k-mer mark of the k-mer at offset j
//   w1 = window code-lo plane (47 bits) | valid bits << 47      valid[j]: the k-mer at offset j (bases j..j+k-1) is stored
//   w2..w7: value indices, three per word (21 bits each; offset j in word 2 + j/3), bit 63 of EVERY word = `more`
// A k-mer whose minimizer sits at offset c - j of its oriented form (c = k-15) is stored iff valid[j] and its planes equal
// the window's bits [j, j+k).  One bucket holds one window, and a window lives in one of the TWO buckets of its minimizer,
// gs_rec_bucket(gh, 0 / 1) (cuckoo placement by the builder: at <= 0.4 windows per bucket practically every window finds
// a place); a probe loads both lines at once, so there is no dependent second access.  What cannot be placed -- a third
// window of the same minimizer, the rare k-mers with two strand views -- goes to the ordinary table below, and the
// `more` bit of both buckets tells a mismatching probe to look there.  The lanes of a wave that share a minimizer load
// the same lines, so a read from the store costs ~26 record lines instead of ~110 bucket lines.
#define GS_REC_WORDS 8
#define GS_REC_WIN_BITS 47
#define GS_REC_VAL_BITS 21
#define GS_REC_MAX_VALUES (1 << GS_REC_VAL_BITS)
#define GS_REC_MORE (1ULL << 63)
#define GS_REC_SLOTS 32  // virtual slots per record bucket (hit counters, compact bitmap): 17 used
#define GS_MAX_STRIPES 8 // devices a striped store spans (one node: 8 GPUs on xGMI)
// first bucket of stripe p when 2^rec_bits buckets are split n_parts ways (the inverse of (b * n_parts) >> rec_bits)
GS_HD uint64_t gs_stripe_first(uint32_t rec_bits, uint32_t n_parts, uint32_t p) {
    return (((uint64_t)p << rec_bits) + n_parts - 1) / n_parts;
}
// ... and the first table bucket of stripe p (groups of four buckets; bucket_bits >= 5 in a striped store)
GS_HD uint64_t gs_tab_stripe_first(uint32_t bucket_bits, uint32_t n_parts, uint32_t p) {
    return 4 * gs_stripe_first(bucket_bits - 2, n_parts, p);
}

#ifndef GS_ONE_PRODUCT
#define GS_ONE_PRODUCT 0  // experiment ("quarter-rate multiplies"): gate bits / second bucket from the bits of ONE product
#endif
GS_HD uint32_t gs_bitrev32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bitreverse32(x);
#else
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x >> 8) & 0x00ff00ffu) | ((x & 0x00ff00ffu) << 8);
    return (x >> 16) | (x << 16);
#endif
}
GS_HD uint32_t gs_rec_bucket(uint32_t gh, uint32_t rec_bits, int choice) {
#if GS_ONE_PRODUCT
    const uint32_t p = gh * 0x27D4EB2Fu;
    return (choice ? gs_bitrev32(p) : p) >> (32 - rec_bits);
#else
    return (gh * (choice ? 0x165667B1u : 0x27D4EB2Fu)) >> (32 - rec_bits);
#endif
}

// minimizer gate word (32 bits) and the two bits an entry sets in it
GS_HD uint32_t gs_mgate_word(uint32_t m, uint32_t word_bits) { return (m * 0x85EBCA77u) >> (32 - word_bits); }
GS_HD uint32_t gs_mgate_bits(uint32_t m) {
#if GS_ONE_PRODUCT
    const uint32_t y = m * 0x85EBCA77u;  // (the word's product: its low bits)
    return (1u << (y & 31)) | (1u << ((y >> 5) & 31));
#else
    const uint32_t y = m * 0xC2B2AE3Du;  // a second product: independent of the word index at any gate size
    return (1u << (y >> 27)) | (1u << ((y >> 22) & 31));
#endif
}
// A third bit of the same word, the "second bucket" hint: set for a gate key iff a window of its minimizer was placed in the
// minimizer's SECOND candidate bucket (gs_rec_bucket(gh, 1)).  A probe whose hint is clear loads the first bucket's line only --
// cuckoo placement tries the first choice first, so about two windows in three sit there --; a hint set by a neighbour of the
// word costs the second line, nothing else.  (Windows without a bucket need no hint: their k-mers are in the table, which the
// `more` bit of EITHER bucket line announces.)
#if GS_ONE_PRODUCT
GS_HD uint32_t gs_mgate_hint(uint32_t m) { return 1u << (((m * 0x85EBCA77u) >> 10) & 31); }
#else
GS_HD uint32_t gs_mgate_hint(uint32_t m) { return 1u << (((m * 0xC2B2AE3Du) >> 17) & 31); }
#endif
// the two context keys of a window (gs_gate_ctx_key of its k-mers with j >= 4 / j <= 3) from the window planes; c = k - 15
// The context key keeps its top 24 bits from the MINIMIZER alone and takes the low 8 from minimizer and context together: the gate
// word of a context key (gs_mgate_word_ctx) then lies in a 64-byte line chosen by the minimizer, so that the two keys a run of
// k-mers around one minimizer asks for (front / behind) -- and the keys of all windows of that minimizer -- cost ONE line request
// where they cost two (round 4: the 473 M-k-mer store sits at 0.89 of the fabric's random-line rate, a third of its 65 requests
// per read are gate words).  Which word of the line and which bits of the word still depend on the whole key.
GS_HD uint32_t gs_gate_ctx_key_raw(uint32_t gh, uint32_t ctx) {
    return ((gh * 0x85EBCA77u) & 0xffffff00u) | (((gh ^ ((ctx + 1u) * 0x9E3779B1u)) * 0xC2B2AE3Du) >> 24);
}
// word of a context-keyed gate: line = top bits of the key (minimizer only, up to 2^24 lines = 1 GiB of gate), word of the line = a hash of the key
GS_HD uint32_t gs_mgate_word_ctx(uint32_t ck, uint32_t word_bits) {
    return word_bits > 4u ? ((ck >> (36u - word_bits)) << 4) | ((ck * 0x9E3779B1u) >> 28) : (ck * 0x85EBCA77u) >> (32u - word_bits);
}
GS_HD uint32_t gs_window_ctx(uint64_t w_hi, uint64_t w_lo, int k, bool behind) {
    const int pos = behind ? (k - GS_MIN_L) + GS_MIN_L : (k - GS_MIN_L) - 4;
    return ((uint32_t)(w_hi >> pos) & 15u) | (((uint32_t)(w_lo >> pos) & 15u) << 4) | (behind ? 256u : 0u);
}

// Context-keyed gate (big stores).  The 15-mer minimizer space is ~60 M; a store of several hundred million k-mers uses most of
// it (473 M k-mers: 40 M distinct minimizers), so that a filter over the minimizers alone lets 68 % of the positions of a read
// that is NOT from the store through -- every one of them then fetches two record lines from HBM.  For such stores the gate is
// keyed by the minimizer AND four bases next to it: a k-mer at window offset j holds the four bases behind its minimizer when
// j >= 4, else (k >= 22) the four bases in front of it -- which side is a function of j alone, so store and probe agree -- and a
// window has ONE such quadruple per side, so the gate holds <= 2 entries per window instead of 1 per minimizer, while a random
// k-mer that shares a stored minimizer matches its quadruple with probability 2^-8.  No false negatives as before (results do
// not depend on the gate); record buckets stay keyed by the minimizer alone.
//   ohi / olo: the k-mer in the orientation in which its minimizer is canonical (gs_min_oriented), minimizer at base (k-15) - j
#define GS_CTX_MIN_K 22
GS_HD uint32_t gs_gate_ctx_key(uint32_t gh, uint32_t ohi, uint32_t olo, int j, int k) {
    const int m = (k - GS_MIN_L) - j;  // first base of the minimizer inside the oriented k-mer
    const bool behind = j >= 4;
    const int pos = behind ? m + GS_MIN_L : m - 4;
    const uint32_t ctx = ((ohi >> pos) & 15u) | (((olo >> pos) & 15u) << 4) | (behind ? 256u : 0u);
    return gs_gate_ctx_key_raw(gh, ctx);
}

struct GsDbDev {
    const unsigned long long *table;  // n_buckets * 8 slots
    const unsigned long long *gate;   // gate_mask+1 words, or nullptr
    uint64_t gate_mask;
    const uint32_t *mgate;  // minimizer gate: 2^mgate_bits 32-bit words, or nullptr
    uint32_t mgate_bits;
    uint32_t rec_bits;      // 2^rec_bits record buckets (rec != nullptr)
    const unsigned long long *rec;  // super-k-mer records, GS_REC_WORDS words per bucket, or nullptr
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
    // striped store (gs_db_create_striped / gs_db_create_stripe): the record buckets are split into n_parts contiguous
    // stripes, stripe p in the HBM of one device; a bucket b belongs to stripe (b * n_parts) >> rec_bits and its line is
    // at rec_biased[p] + b * GS_REC_WORDS (the stripe's base pointer minus its first bucket).  Foreign stripes are read
    // over xGMI peer access and never written: the seen bits of a striped store live in the run's own bitmap.
    // n_parts <= 1: plain store (`rec` above).
    // The overflow table is striped the same way in units of four buckets (so that a stripe starts on a word of the slot
    // bitmap): bucket b belongs to stripe ((b >> 2) * n_parts) >> (bucket_bits - 2), its line is at tab_biased[p] +
    // b * GS_SLOTS_PER_BUCKET; `table` is nullptr then.
    uint32_t n_parts;
    uint32_t mgate_ctx;  // 1: the minimizer gate is keyed by gs_gate_ctx_key (minimizer + four neighbouring bases), 0: by the minimizer
    const unsigned long long *rec_biased[GS_MAX_STRIPES];
    const unsigned long long *tab_biased[GS_MAX_STRIPES];
};
