/*
 * gs_oracle.c -- CPU restatement of Genestrip's `match` / `filter` hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see gs_oracle.h).  Plain C, sequential semantics of the
 * reference's single-threaded (`threads=0`) run; the optional OpenMP mode only shards
 * reads over threads the way the reference's consumer threads do and merges integer
 * stats, which are order independent (reference test K4,
 * core/src/test/java/org/metagene/genestrip/match/FastqKMerMatcherTest.java:243-313).
 *
 * Parity: pinned by K1..K10 in tests/test_oracle_golden.py.
 *
 * Citations: C/ = core/src/main/java/org/metagene/genestrip/,
 *            B/ = base/src/main/java/org/metagene/genestrip/  (under /root/reference).
 */
#include "gs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * java.util.Random (documented LCG)
 * ---------------------------------------------------------------------------------------- */
#define JR_MULT 0x5DEECE66DULL
#define JR_MASK ((1ULL << 48) - 1)

void orc_jrandom_init(orc_jrandom *r, int64_t seed) { r->seed = ((uint64_t)seed ^ JR_MULT) & JR_MASK; }

int32_t orc_jrandom_next(orc_jrandom *r, int bits) {
    r->seed = (r->seed * JR_MULT + 0xBULL) & JR_MASK;
    return (int32_t)(int64_t)(r->seed >> (48 - bits)); /* (int)(seed >>> (48-bits)) */
}

int64_t orc_jrandom_next_long(orc_jrandom *r) {
    int64_t hi = (int64_t)orc_jrandom_next(r, 32);
    int64_t lo = (int64_t)orc_jrandom_next(r, 32);
    return (int64_t)(((uint64_t)hi << 32) + (uint64_t)lo);
}

int32_t orc_jrandom_next_int(orc_jrandom *r, int32_t bound) {
    int32_t rr = orc_jrandom_next(r, 31);
    int32_t m = bound - 1;
    if ((bound & m) == 0) return (int32_t)(((int64_t)bound * (int64_t)rr) >> 31);
    for (int32_t u = rr; u - (rr = u % bound) + m < 0; u = orc_jrandom_next(r, 31)) {
    }
    return rr;
}

/* ------------------------------------------------------------------------------------------
 * 2-bit codec, C/util/CGAT.java
 * ---------------------------------------------------------------------------------------- */
static inline int code_straight(uint8_t c) { /* CGAT_JUMP_TABLE :66-69 */
    switch (c) {
    case 'C': return 0;
    case 'G': return 1;
    case 'A': return 2;
    case 'T': return 3;
    default: return -1;
    }
}
static inline int code_reverse(uint8_t c) { /* CGAT_REVERSE_JUMP_TABLE :71-74 */
    switch (c) {
    case 'C': return 1;
    case 'G': return 0;
    case 'A': return 3;
    case 'T': return 2;
    default: return -1;
    }
}

int64_t orc_kmer_straight(const uint8_t *seq, int start, int k, int *bad_pos) {
    uint64_t res = 0;
    if (bad_pos) *bad_pos = -1;
    for (int i = start; i < start + k; i++) {
        int c = code_straight(seq[i]);
        res = (res << 2) | (res >> 62); /* rotateLeft(res, 2) */
        if (c < 0) {
            if (bad_pos) *bad_pos = i;
            return -1;
        }
        res += (uint64_t)c;
    }
    return (int64_t)res;
}

int64_t orc_kmer_reverse(const uint8_t *seq, int start, int k, int *bad_pos) {
    uint64_t res = 0;
    if (bad_pos) *bad_pos = -1;
    for (int i = start + k - 1; i >= start; i--) {
        int c = code_reverse(seq[i]);
        res = (res << 2) | (res >> 62);
        if (c < 0) {
            if (bad_pos) *bad_pos = i;
            return -1;
        }
        res += (uint64_t)c;
    }
    return (int64_t)res;
}

static inline uint64_t straight_filter(int k) { /* SHIFT_FILTERS_STRAIGHT :80-83 */
    return k >= 32 ? ~0ULL : ~(~0ULL << (2 * k));
}

int64_t orc_next_straight(int64_t kmer, uint8_t bp, int k) {
    int c = code_straight(bp);
    if (c < 0) return -1;
    return (int64_t)((((uint64_t)kmer << 2) & straight_filter(k)) | (uint64_t)c);
}

int64_t orc_next_reverse(int64_t kmer, uint8_t bp, int k) {
    int c = code_reverse(bp);
    if (c < 0) return -1;
    return (int64_t)(((uint64_t)kmer >> 2) | ((uint64_t)c << (2 * (k - 1))));
}

int64_t orc_standard_kmer(int64_t straight, int64_t reverse) { return straight > reverse ? straight : reverse; }

int64_t orc_kmer_canonical(const uint8_t *seq, int start, int k, int *bad_pos) {
    int64_t r = orc_kmer_reverse(seq, start, k, bad_pos);
    int64_t s = orc_kmer_straight(seq, start, k, bad_pos);
    return orc_standard_kmer(r, s);
}

/* ------------------------------------------------------------------------------------------
 * Bloom filters, C/bloom/
 * ---------------------------------------------------------------------------------------- */
struct orc_bloom {
    int kind;
    int64_t bits;    /* XOR/Murmur: bit count ; Blocked: bucket count */
    int32_t hashes;
    int64_t *factors; /* hashes entries ; Blocked: 1 entry = seed */
    uint64_t *words;
    int64_t n_words;
};

static inline int64_t jabs_mod(int64_t v, int64_t m) { /* Math.abs(v % m): % truncates in C99 as in Java */
    int64_t r = v % m;
    return r < 0 ? -r : r;
}

int64_t orc_murmur_hash64(int64_t data_, int64_t hash_base) {
    const uint64_t C1 = 0x87c37b91114253d5ULL, C2 = 0x4cf5ad432745937fULL;
    uint64_t data = (uint64_t)data_, hash = (uint64_t)hash_base;
    uint64_t k = __builtin_bswap64(data); /* Long.reverseBytes, inlined at :64-65 */
    k *= C1;
    k = (k << 31) | (k >> 33);
    k *= C2;
    hash ^= k;
    hash = ((hash << 27) | (hash >> 37)) * 5 + 0x52dce729ULL;
    hash ^= 8; /* length = Long.BYTES */
    hash ^= hash >> 33;
    hash *= 0xff51afd7ed558ccdULL;
    hash ^= hash >> 33;
    hash *= 0xc4ceb9fe1a85ec53ULL;
    hash ^= hash >> 33;
    return (int64_t)(hash ^ data);
}

orc_bloom *orc_bloom_create(int kind, int64_t n, double fpp) {
    orc_bloom *b = (orc_bloom *)calloc(1, sizeof(*b));
    orc_jrandom rnd;
    orc_jrandom_init(&rnd, 42);
    b->kind = kind;
    if (kind == ORC_BLOOM_BLOCKED) {
        /* BlockedKMerBloomFilter.ensureExpectedSize :201-219, bitsPerKey = 10 */
        int64_t entries = n < 1 ? 1 : n;
        int64_t bits = entries * 10;
        b->bits = (bits + 63) / 64;
        b->n_words = b->bits + 16 + 1;
        b->hashes = 0;
        b->factors = (int64_t *)malloc(sizeof(int64_t));
        b->factors[0] = orc_jrandom_next_long(&rnd); /* :92 */
    } else {
        /* AbstractKMerBloomFilter.optimalNumOfBits :183-185, optimalNumOfHashFunctions :172-174 */
        double dbits = -(double)n * log(fpp) / (log(2.0) * log(2.0));
        int64_t bits = (int64_t)dbits;
        if (bits < 1) bits = 1;
        b->bits = bits;
        double dh = ((double)bits) / (double)n * log(2.0);
        int64_t h = (int64_t)floor(dh + 0.5); /* Math.round */
        b->hashes = (int32_t)(h < 1 ? 1 : h);
        b->factors = (int64_t *)malloc(sizeof(int64_t) * (size_t)b->hashes);
        for (int i = 0; i < b->hashes; i++) b->factors[i] = orc_jrandom_next_long(&rnd);
        b->n_words = (bits + 63) / 64;
    }
    b->words = (uint64_t *)calloc((size_t)b->n_words, sizeof(uint64_t));
    return b;
}

void orc_bloom_destroy(orc_bloom *b) {
    if (!b) return;
    free(b->factors);
    free(b->words);
    free(b);
}

static inline int64_t bloom_index(const orc_bloom *b, int64_t key, int i) {
    int64_t h = b->kind == ORC_BLOOM_XOR ? (b->factors[i] ^ key) /* XORKMerBloomFilter.java:43-46 */
                                         : orc_murmur_hash64(key, b->factors[i]);
    return jabs_mod(h, b->bits); /* reduce :57-59 */
}

typedef struct { int64_t s1, s2; uint64_t m1, m2; } blocked_probe;

static inline blocked_probe blocked_prepare(const orc_bloom *b, int64_t key) {
    /* BlockedKMerBloomFilter.containsLong :181-199; Java shift counts use the low 6 bits, >> is arithmetic */
    blocked_probe p;
    int64_t hash = b->factors[0] ^ key;
    p.s1 = jabs_mod(hash, b->bits);
    uint64_t uh = (uint64_t)hash;
    uh ^= (uh << 32) | (uh >> 32);
    int64_t sh = (int64_t)uh;
    p.m1 = (1ULL << (sh & 63)) | (1ULL << ((sh >> 6) & 63));
    p.m2 = (1ULL << ((sh >> 12) & 63)) | (1ULL << ((sh >> 18) & 63));
    p.s2 = p.s1 + 1 + (int64_t)(uh >> 60);
    return p;
}

void orc_bloom_put(orc_bloom *b, int64_t key) {
    if (b->kind == ORC_BLOOM_BLOCKED) {
        blocked_probe p = blocked_prepare(b, key);
        b->words[p.s1] |= p.m1;
        b->words[p.s2] |= p.m2;
    } else {
        for (int i = 0; i < b->hashes; i++) {
            int64_t idx = bloom_index(b, key, i);
            b->words[idx >> 6] |= 1ULL << (idx & 63);
        }
    }
}

void orc_bloom_put_many(orc_bloom *b, const int64_t *keys, int64_t n) {
    for (int64_t i = 0; i < n; i++) orc_bloom_put(b, keys[i]);
}

/* the same bits, set from several threads (bit sets commute): for filters of tens of millions of keys */
void orc_bloom_put_many_mt(orc_bloom *b, const int64_t *keys, int64_t n, int threads) {
    if (threads <= 1 || b->kind == ORC_BLOOM_BLOCKED) {
        orc_bloom_put_many(b, keys, n);
        return;
    }
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t i = 0; i < n; i++)
        for (int h = 0; h < b->hashes; h++) {
            int64_t idx = bloom_index(b, keys[i], h);
            __atomic_fetch_or(&b->words[idx >> 6], 1ULL << (idx & 63), __ATOMIC_RELAXED);
        }
}

int orc_bloom_contains(const orc_bloom *b, int64_t key) {
    if (b->kind == ORC_BLOOM_BLOCKED) {
        blocked_probe p = blocked_prepare(b, key);
        return ((p.m1 & b->words[p.s1]) == p.m1) && ((p.m2 & b->words[p.s2]) == p.m2);
    }
    for (int i = 0; i < b->hashes; i++) { /* AbstractKMerBloomFilter.containsLong :209-216 */
        int64_t idx = bloom_index(b, key, i);
        if (!((b->words[idx >> 6] >> (idx & 63)) & 1ULL)) return 0;
    }
    return 1;
}

int orc_bloom_kind(const orc_bloom *b) { return b->kind; }
int64_t orc_bloom_bits(const orc_bloom *b) { return b->bits; }
int32_t orc_bloom_hashes(const orc_bloom *b) { return b->hashes; }
const int64_t *orc_bloom_hash_factors(const orc_bloom *b) { return b->factors; }
const uint64_t *orc_bloom_words(const orc_bloom *b) { return b->words; }
int64_t orc_bloom_n_words(const orc_bloom *b) { return b->n_words; }

/* FastqBloomFilter.isAcceptRead :120-161 */
int orc_filter_accept_read(const orc_bloom *b, int k, int min_pos_count, double positive_ratio,
                           const uint8_t *read, int read_size) {
    int max = read_size - k + 1;
    int pos_threshold = min_pos_count > 0 ? min_pos_count : (int)((double)max * positive_ratio);
    int neg_threshold = max - pos_threshold;
    int64_t kmer = -1, rkmer = -1;
    int counter = 0, neg = 0, bad;
    for (int i = 0; i < max; i++) {
        if (kmer == -1) {
            kmer = orc_kmer_straight(read, i, k, &bad);
            if (kmer == -1)
                i = bad;
            else
                rkmer = orc_kmer_reverse(read, i, k, NULL);
        } else {
            kmer = orc_next_straight(kmer, read[i + k - 1], k);
            if (kmer == -1)
                i += k - 1;
            else
                rkmer = orc_next_reverse(rkmer, read[i + k - 1], k);
        }
        if (kmer != -1) {
            if (orc_bloom_contains(b, orc_standard_kmer(kmer, rkmer))) {
                if (++counter >= pos_threshold) return 1;
            } else {
                if (++neg > neg_threshold) return 0;
            }
        }
    }
    return 0;
}

void orc_filter_batch(const orc_bloom *b, int k, int min_pos_count, double positive_ratio,
                      const uint8_t *seq, const uint64_t *off, int64_t n, uint8_t *accept, int threads) {
    (void)threads;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(threads > 1 ? threads : 1)
    for (int64_t r = 0; r < n; r++)
        accept[r] = (uint8_t)orc_filter_accept_read(b, k, min_pos_count, positive_ratio, seq + off[r],
                                                    (int)(off[r + 1] - off[r]));
}

/* ------------------------------------------------------------------------------------------
 * Store + tree
 * ---------------------------------------------------------------------------------------- */
struct orc_db {
    int k;
    int64_t n;
    int64_t *kmers;
    int32_t *vidx;
    int32_t n_values;
    int32_t *parent; /* n_values; -1 root; -2 not a node */
    int32_t *depth;
    int has_tree;
    orc_bloom *gate;
    /* RadixKMerStore layout (C/store/RadixKMerStore.java), radix_bits == 0: KMerSortedArray layout.
     * radix_entries = the buckets radixIndex[0], radixIndex[1], .. back to back (every bucket exactly full),
     * bucket_off[r] = bucketOffset[r] after optimize() (:651-655), bucket_off[2^radix_bits] = entries.
     * kmers[] / vidx[] are then in visit() order (:714-729), i.e. indexed by the global position `pos`. */
    int radix_bits;
    int remaining_bits;
    uint64_t remaining_mask;
    uint64_t *radix_entries;
    int64_t *bucket_off;
};

orc_db *orc_db_create(int k, int64_t n, const int64_t *kmers, const int32_t *vidx, int32_t n_values,
                      const int32_t *parent_vi, int bloom_gate) {
    orc_db *db = (orc_db *)calloc(1, sizeof(*db));
    db->k = k;
    db->n = n;
    db->kmers = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    db->vidx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    memcpy(db->kmers, kmers, sizeof(int64_t) * (size_t)n);
    memcpy(db->vidx, vidx, sizeof(int32_t) * (size_t)n);
    db->n_values = n_values;
    db->parent = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_values > 0 ? n_values : 1));
    db->depth = (int32_t *)calloc((size_t)(n_values > 0 ? n_values : 1), sizeof(int32_t));
    db->has_tree = parent_vi != NULL;
    for (int32_t v = 0; v < n_values; v++) db->parent[v] = parent_vi ? parent_vi[v] : -1;
    for (int32_t v = 0; v < n_values; v++) {
        int d = 0;
        if (db->parent[v] == -2) continue;
        for (int32_t a = db->parent[v]; a >= 0; a = db->parent[a]) d++; /* getLevel() */
        db->depth[v] = d;
    }
    if (bloom_gate) {
        /* AbstractKMerStore.createOptimizedFilter :271-284 with optimizedFpp = 0.01 => Blocked */
        db->gate = orc_bloom_create(ORC_BLOOM_BLOCKED, n, 0.01);
        for (int64_t i = 0; i < n; i++) orc_bloom_put(db->gate, kmers[i]);
    }
    return db;
}

/* ---- RadixKMerStore (opt-in `useRadixStore`): entry = valueIndex << remainingBits | (kmer >>> radixBits),
 * bucket = low radixBits bits of the k-mer, buckets sorted by the remaining bits (optimize, :632-655) ---- */
static uint64_t g_radix_sort_mask; /* qsort has no context argument; the oracle builds stores from one thread */
static int radix_entry_cmp(const void *a, const void *b) { /* remainingComparator :640-641 */
    uint64_t x = *(const uint64_t *)a & g_radix_sort_mask, y = *(const uint64_t *)b & g_radix_sort_mask;
    return x < y ? -1 : (x > y ? 1 : 0);
}

int32_t orc_radix_max_values(int radix_bits) { /* maxValuesForRadix :160-164 */
    if (radix_bits < 16 || radix_bits > 30) return -1; /* checkRadixBits :166-171 */
    int value_bits = 64 - (62 - radix_bits);
    if (value_bits > 30) value_bits = 30;
    return (int32_t)1 << value_bits;
}

orc_db *orc_db_create_radix(int k, int radix_bits, int64_t n, const int64_t *kmers, const int32_t *vidx,
                            int32_t n_values, const int32_t *parent_vi, int bloom_gate) {
    if (orc_radix_max_values(radix_bits) < 0 || n_values > orc_radix_max_values(radix_bits)) return NULL;
    /* tree arrays and scalar fields as for the sorted layout; the entry arrays are rebuilt below */
    orc_db *db = orc_db_create(k, 0, kmers, vidx, n_values, parent_vi, 0);
    const int64_t n_buckets = (int64_t)1 << radix_bits;
    const uint64_t radix_mask = (uint64_t)n_buckets - 1;
    db->radix_bits = radix_bits;
    db->remaining_bits = 62 - radix_bits; /* remainingBitsForRadix :146-148 */
    db->remaining_mask = (1ULL << db->remaining_bits) - 1;
    db->n = n;
    /* the caller of the constructor has counted the k-mers per bucket with radixOf (:306-308); capacities are exact */
    db->bucket_off = (int64_t *)calloc((size_t)n_buckets + 1, sizeof(int64_t));
    for (int64_t i = 0; i < n; i++) db->bucket_off[((uint64_t)kmers[i] & radix_mask) + 1]++;
    for (int64_t r = 0; r < n_buckets; r++) db->bucket_off[r + 1] += db->bucket_off[r];
    db->radix_entries = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n > 0 ? n : 1));
    int64_t *fill = (int64_t *)calloc((size_t)n_buckets, sizeof(int64_t));
    for (int64_t i = 0; i < n; i++) { /* putLong :319-364 in the given order (no fill-time filter) */
        const uint64_t radix = (uint64_t)kmers[i] & radix_mask;
        const uint64_t remaining = (uint64_t)kmers[i] >> radix_bits;                       /* remainingOf :311-313 */
        db->radix_entries[db->bucket_off[radix] + fill[radix]++] =
            ((uint64_t)(int64_t)vidx[i] << db->remaining_bits) | remaining;                 /* entryOf :315-317 */
    }
    free(fill);
    g_radix_sort_mask = db->remaining_mask; /* optimize :632-650 */
    for (int64_t r = 0; r < n_buckets; r++) {
        const int64_t f = db->bucket_off[r + 1] - db->bucket_off[r];
        if (f > 1) qsort(db->radix_entries + db->bucket_off[r], (size_t)f, sizeof(uint64_t), radix_entry_cmp);
    }
    /* the flat (kmer, value index) arrays in visit() order (:714-729): position pos = bucketOffset[r] + i */
    free(db->kmers);
    free(db->vidx);
    db->kmers = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    db->vidx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t r = 0; r < n_buckets; r++)
        for (int64_t p = db->bucket_off[r]; p < db->bucket_off[r + 1]; p++) {
            const uint64_t e = db->radix_entries[p];
            db->kmers[p] = (int64_t)(((e & db->remaining_mask) << radix_bits) | (uint64_t)r);
            db->vidx[p] = (int32_t)(e >> db->remaining_bits);
        }
    if (bloom_gate) { /* optimize :656-671: the optimized filter is refilled from the reassembled k-mers */
        db->gate = orc_bloom_create(ORC_BLOOM_BLOCKED, n, 0.01);
        for (int64_t i = 0; i < n; i++) orc_bloom_put(db->gate, db->kmers[i]);
    }
    return db;
}

/* KMerStore.visit (KMerSortedArray.java:426-439 / RadixKMerStore.java:714-729): entry at position pos */
int64_t orc_db_entries(const orc_db *db) { return db->n; }
void orc_db_visit(const orc_db *db, int64_t *kmers, int32_t *vidx) {
    memcpy(kmers, db->kmers, sizeof(int64_t) * (size_t)db->n);
    memcpy(vidx, db->vidx, sizeof(int32_t) * (size_t)db->n);
}

/* RadixKMerStore.getLong :369-412 (sorted branch) */
static int32_t radix_db_get(const orc_db *db, int64_t kmer, int64_t *pos) {
    const uint64_t radix = (uint64_t)kmer & (((uint64_t)1 << db->radix_bits) - 1);
    const int64_t fill = db->bucket_off[radix + 1] - db->bucket_off[radix];
    if (fill == 0) return -1; /* null bucket: return early, even before the filter (:372-375) */
    if (db->gate && !orc_bloom_contains(db->gate, kmer)) return -1;
    const uint64_t remaining = (uint64_t)kmer >> db->radix_bits;
    const uint64_t *bucket = db->radix_entries + db->bucket_off[radix];
    int64_t lo = 0, hi = fill - 1;
    while (lo <= hi) {
        const int64_t mid = (int64_t)(((uint64_t)lo + (uint64_t)hi) >> 1);
        const uint64_t mid_rem = bucket[mid] & db->remaining_mask;
        if (mid_rem < remaining)
            lo = mid + 1;
        else if (mid_rem > remaining)
            hi = mid - 1;
        else {
            if (pos) *pos = db->bucket_off[radix] + mid; /* posStore[0] = bucketOffset[radix] + pos (:408-410) */
            const int32_t vi = (int32_t)(bucket[mid] >> db->remaining_bits);
            if (db->parent[vi] == -2) return -1;
            return vi;
        }
    }
    return -1;
}

void orc_db_destroy(orc_db *db) {
    if (!db) return;
    free(db->radix_entries);
    free(db->bucket_off);
    free(db->kmers);
    free(db->vidx);
    free(db->parent);
    free(db->depth);
    orc_bloom_destroy(db->gate);
    free(db);
}

/* KMerSortedArray.getLong :298-349 (sorted branch): gate, then binary search; a value without a
 * tree node converts to null (Database.convertKMerStore :136-143) and so reads as a miss. */
int32_t orc_db_get(const orc_db *db, int64_t kmer, int64_t *pos) {
    if (db->radix_bits) return radix_db_get(db, kmer, pos);
    if (db->gate && !orc_bloom_contains(db->gate, kmer)) return -1;
    int64_t lo = 0, hi = db->n - 1;
    while (lo <= hi) { /* java.util.Arrays.binarySearch */
        int64_t mid = (int64_t)(((uint64_t)lo + (uint64_t)hi) >> 1);
        int64_t v = db->kmers[mid];
        if (v < kmer)
            lo = mid + 1;
        else if (v > kmer)
            hi = mid - 1;
        else {
            if (pos) *pos = mid;
            int32_t vi = db->vidx[mid];
            if (db->parent[vi] == -2) return -1;
            return vi;
        }
    }
    return -1;
}

int orc_tree_is_ancestor_of(const orc_db *db, int32_t node, int32_t anc) { /* SmallTaxTree :242-252 */
    while (node >= 0) {
        if (node == anc) return 1;
        node = db->parent[node];
    }
    return 0;
}

int32_t orc_tree_lca(const orc_db *db, int32_t a, int32_t b) { /* :263-289, -1 == null */
    if (a == b) return a;
    if (a < 0 || b < 0) return -1;
    while (db->depth[a] > db->depth[b]) a = db->parent[a];
    while (db->depth[b] > db->depth[a]) b = db->parent[b];
    while (a != b) {
        a = db->parent[a];
        b = db->parent[b];
        if (a < 0 || b < 0) return -1;
    }
    return a;
}

/* ------------------------------------------------------------------------------------------
 * matchRead, C/match/FastqKMerMatcher.java:327-535
 * ---------------------------------------------------------------------------------------- */
#define NODE_NULL (-1)
#define NODE_INVALID (-2)

typedef struct {
    /* per consumer (thread) state: SmallTaxIdNode.counts/countsInitKeys (:643-658) and
     * readNoPerCPerStat (:78,:134) */
    int32_t *cnt;
    int64_t *cnt_key;
    int64_t *read_no_row;
    int64_t *table;  /* n_values x ORC_N_COLS (thread private, merged at the end)  */
    double *dtable;  /* n_values x ORC_N_DCOLS */
    int32_t *path_node;
    int32_t *path_cnt;
} consumer_t;

struct orc_run {
    const orc_db *db;
    orc_match_cfg cfg;
    int64_t *table;
    double *dtable;
    uint64_t *unique_bits; /* KMerUniqueCounterBits bit vector, indexed by store rank */
    int16_t *hit_counts;   /* countsVector (LargeShortVector), indexed by store rank; NULL unless max_kmer_res_counts > 0 */
    int n_consumers;
    consumer_t *cons;
};

static void consumer_init(consumer_t *c, const orc_db *db, int max_paths) {
    size_t nv = (size_t)(db->n_values > 0 ? db->n_values : 1);
    c->cnt = (int32_t *)calloc(nv, sizeof(int32_t));
    c->cnt_key = (int64_t *)malloc(nv * sizeof(int64_t));
    c->read_no_row = (int64_t *)malloc(nv * sizeof(int64_t));
    c->table = (int64_t *)calloc(nv * ORC_N_COLS, sizeof(int64_t));
    c->dtable = (double *)calloc(nv * ORC_N_DCOLS, sizeof(double));
    c->path_node = (int32_t *)malloc(sizeof(int32_t) * (size_t)(max_paths > 0 ? max_paths : 1));
    c->path_cnt = (int32_t *)malloc(sizeof(int32_t) * (size_t)(max_paths > 0 ? max_paths : 1));
    for (size_t i = 0; i < nv; i++) {
        c->cnt_key[i] = -1;
        c->read_no_row[i] = -1;
        c->table[i * ORC_N_COLS + ORC_C_MAX_CONTIG_READ_NO] = -1;
    }
}

static void consumer_free(consumer_t *c) {
    free(c->cnt);
    free(c->cnt_key);
    free(c->read_no_row);
    free(c->table);
    free(c->dtable);
    free(c->path_node);
    free(c->path_cnt);
}

orc_run *orc_match_begin(const orc_db *db, const orc_match_cfg *cfg) {
    orc_run *run = (orc_run *)calloc(1, sizeof(*run));
    size_t nv = (size_t)(db->n_values > 0 ? db->n_values : 1);
    run->db = db;
    run->cfg = *cfg;
    run->table = (int64_t *)calloc(nv * ORC_N_COLS, sizeof(int64_t));
    run->dtable = (double *)calloc(nv * ORC_N_DCOLS, sizeof(double));
    for (size_t i = 0; i < nv; i++) run->table[i * ORC_N_COLS + ORC_C_MAX_CONTIG_READ_NO] = -1;
    run->unique_bits = (uint64_t *)calloc((size_t)((db->n + 63) / 64 + 1), sizeof(uint64_t));
    if (cfg->max_kmer_res_counts > 0) run->hit_counts = (int16_t *)calloc((size_t)(db->n > 0 ? db->n : 1), sizeof(int16_t));
    return run;
}

void orc_match_destroy(orc_run *run) {
    if (!run) return;
    for (int i = 0; i < run->n_consumers; i++) consumer_free(&run->cons[i]);
    free(run->cons);
    free(run->table);
    free(run->dtable);
    free(run->unique_bits);
    free(run->hit_counts);
    free(run);
}

static inline void inc_count(consumer_t *c, int32_t node, int64_t key) { /* SmallTaxIdNode.incCount */
    if (c->cnt_key[node] == key)
        c->cnt[node]++;
    else {
        c->cnt_key[node] = key;
        c->cnt[node] = 1;
    }
}

static int sum_counts(const orc_db *db, const consumer_t *c, int32_t node, int64_t key) { /* :184-193 */
    int res = 0;
    while (node >= 0) {
        if (c->cnt_key[node] == key) res += c->cnt[node];
        node = db->parent[node];
    }
    return res;
}

static int32_t lowest_node_sum_above(const orc_db *db, const consumer_t *c, int32_t node, int64_t key,
                                     int threshold) { /* :208-221 */
    int res = 0;
    while (node >= 0) {
        if (c->cnt_key[node] == key) {
            res += c->cnt[node];
            if (res >= threshold) return node;
        }
        node = db->parent[node];
    }
    return -1;
}

/* contig flush, FastqKMerMatcher.java:396-411 / :457-471.  "first read with the strict maximum"
 * is what a single-threaded run records as max contig descriptor; across threads we keep
 * (len, smallest readNo). */
static inline void flush_contig(int64_t *row, int contig_len, int64_t read_no) {
    row[ORC_C_KMERS] += contig_len;
    row[ORC_C_CONTIGS]++;
    row[ORC_C_CONTIG_LEN_SQ_SUM] += (int64_t)contig_len * contig_len;
    if (contig_len > row[ORC_C_MAX_CONTIG_LEN]) {
        row[ORC_C_MAX_CONTIG_LEN] = contig_len;
        row[ORC_C_MAX_CONTIG_READ_NO] = read_no;
    }
}

static void merge_path(const orc_db *db, consumer_t *c, int *used, int max_paths, int32_t node) { /* :568-586 */
    for (int i = 0; i < *used; i++) {
        if (orc_tree_is_ancestor_of(db, node, c->path_node[i])) {
            c->path_node[i] = node;
            return;
        } else if (orc_tree_is_ancestor_of(db, c->path_node[i], node)) {
            return;
        }
    }
    if (*used < max_paths) c->path_node[(*used)++] = node;
}

/* one read; returns ORC_F_* flags, *class_vi = entry.classNode */
static int match_read(orc_run *run, consumer_t *c, const uint8_t *read, int read_size, int64_t read_no,
                      int32_t *class_vi_out) {
    const orc_db *db = run->db;
    const orc_match_cfg *cfg = &run->cfg;
    const int k = db->k;
    int found = 0;
    int tax_err = cfg->classify ? 0 : -1;
    int max = read_size - k + 1;
    double max_err_times_max = cfg->max_read_tax_err * (double)max;
    int32_t last = NODE_NULL, node;
    int contig_len = 0, used_paths = 0;
    int64_t *stats = NULL;
    int64_t kmer = -1, rkmer = -1, pos = 0;
    int old_index = 0, bad;
    int32_t class_vi = -1;

    for (int i = 0; i < max; i++) {
        if (kmer == -1) {
            kmer = orc_kmer_straight(read, i, k, &bad);
            if (kmer == -1) {
                old_index = i;
                i = bad;
            } else
                rkmer = orc_kmer_reverse(read, i, k, NULL);
        } else {
            uint8_t lb = read[i + k - 1];
            kmer = orc_next_straight(kmer, lb, k);
            if (kmer == -1) {
                old_index = i;
                i += k - 1;
            } else
                rkmer = orc_next_reverse(rkmer, lb, k);
        }
        node = kmer == -1 ? NODE_INVALID : orc_db_get(db, orc_standard_kmer(kmer, rkmer), &pos);
        int new_contig = node != last;
        if (tax_err != -1) {
            if (node < 0) { /* null or INVALID */
                tax_err++;
                if (cfg->max_read_tax_err >= 0) {
                    if ((cfg->max_read_tax_err >= 1 && (double)tax_err > cfg->max_read_tax_err) ||
                        ((double)tax_err > max_err_times_max))
                        tax_err = -1;
                }
            } else {
                inc_count(c, node, read_no);
                if (new_contig) merge_path(db, c, &used_paths, cfg->max_paths, node);
            }
        }
        if (new_contig) {
            if (contig_len > 0) {
                if (stats) flush_contig(stats, contig_len, read_no);
                contig_len = 0;
            }
        }
        if (node == NODE_INVALID)
            contig_len += i >= max ? max - old_index : i - old_index + 1;
        else
            contig_len++;
        last = node;
        if (node >= 0) {
            found = 1;
            if (new_contig) {
                stats = c->table + (size_t)node * ORC_N_COLS;
                if (c->read_no_row[node] != read_no) {
                    c->read_no_row[node] = read_no;
                    stats[ORC_C_READS_1KMER]++;
                }
            }
            if (cfg->count_unique) { /* KMerUniqueCounterBits.putInlined :117-143 */
#pragma omp atomic
                run->unique_bits[pos >> 6] |= 1ULL << (pos & 63);
                if (run->hit_counts) { /* ++countsVector.shorts[index]: Java short arithmetic wraps (:134-140) */
#pragma omp critical(orc_hit_counts)
                    run->hit_counts[pos] = (int16_t)(uint16_t)((uint16_t)run->hit_counts[pos] + 1u);
                }
            }
        } else
            stats = NULL;
    }

    int flags = 0;
    if (found) {
        flags |= ORC_F_FOUND | ORC_F_RETURNED;
        if (contig_len > 0 && stats) flush_contig(stats, contig_len, read_no);
        if (tax_err != -1) {
            int ties = 0;
            for (int i = 0; i < cfg->max_paths; i++) c->path_cnt[i] = 0; /* nextEntry :277-280 */
            for (int i = 0; i < used_paths; i++) {
                int sum = sum_counts(db, c, c->path_node[i], read_no);
                if (sum > c->path_cnt[0]) {
                    c->path_cnt[0] = sum;
                    c->path_node[0] = c->path_node[i];
                    ties = 0;
                } else if (sum == c->path_cnt[0]) {
                    ties++;
                    c->path_cnt[ties] = sum;
                    c->path_node[ties] = c->path_node[i];
                }
            }
            if (cfg->threshold > 1)
                for (int i = 0; i <= ties; i++)
                    c->path_node[i] = lowest_node_sum_above(db, c, c->path_node[i], read_no, cfg->threshold);
            int32_t cn = c->path_node[0];
            for (int i = 1; i <= ties; i++) cn = orc_tree_lca(db, cn, c->path_node[i]);
            class_vi = cn;
            if (cn < 0) {
                flags &= ~ORC_F_RETURNED; /* "return false" :497-500 */
            } else {
                int read_kmers = (ties > 0 || cfg->threshold > 1) ? sum_counts(db, c, c->path_node[0], read_no)
                                                                  : c->path_cnt[0];
                int class_err = max - read_kmers;
                double mc = cfg->max_read_class_err;
                if (mc < 0 || (mc >= 1 && (double)class_err <= mc) || ((double)class_err <= mc * (double)max)) {
                    double err = ((double)tax_err) / (double)max;
                    double cerr = ((double)class_err) / (double)max;
                    int64_t *row = c->table + (size_t)cn * ORC_N_COLS;
                    double *drow = c->dtable + (size_t)cn * ORC_N_DCOLS;
                    row[ORC_C_READS]++;
                    row[ORC_C_READS_KMERS] += read_kmers;
                    row[ORC_C_READS_BPS] += read_size;
                    drow[ORC_D_ERR_SUM] += err;
                    drow[ORC_D_ERR_SQ_SUM] += err * err;
                    drow[ORC_D_CLASS_ERR_SUM] += cerr;
                    drow[ORC_D_CLASS_ERR_SQ_SUM] += cerr * cerr;
                    flags |= ORC_F_COUNTED;
                }
            }
        }
    }
    *class_vi_out = class_vi;
    return flags;
}

int orc_match_submit(orc_run *run, const uint8_t *seq, const uint64_t *off, int64_t n, int64_t first_read_no,
                     int32_t *class_vi, uint8_t *flags, int threads) {
    if (threads < 1) threads = 1;
    if (run->n_consumers < threads) {
        run->cons = (consumer_t *)realloc(run->cons, sizeof(consumer_t) * (size_t)threads);
        for (int i = run->n_consumers; i < threads; i++) consumer_init(&run->cons[i], run->db, run->cfg.max_paths);
        run->n_consumers = threads;
    }
#pragma omp parallel num_threads(threads)
    {
#ifdef _OPENMP
        consumer_t *c = &run->cons[omp_get_thread_num()];
#else
        consumer_t *c = &run->cons[0];
#endif
#pragma omp for schedule(dynamic, 512)
        for (int64_t r = 0; r < n; r++) {
            int32_t cv;
            int f = match_read(run, c, seq + off[r], (int)(off[r + 1] - off[r]), first_read_no + r, &cv);
            if (class_vi) class_vi[r] = cv;
            if (flags) flags[r] = (uint8_t)f;
        }
    }
    return 0;
}

static void merge_consumers(orc_run *run);

int orc_match_finish(orc_run *run, int64_t *table, double *dtable) {
    const orc_db *db = run->db;
    size_t nv = (size_t)db->n_values;
    merge_consumers(run);
    /* KMerUniqueCounterBits.getUniqueKmerCounts :146-163 ; -1 when counting is off (:226-230) */
    for (size_t v = 0; v < nv; v++) run->table[v * ORC_N_COLS + ORC_C_UNIQUE_KMERS] = run->cfg.count_unique ? 0 : -1;
    if (run->cfg.count_unique)
        for (int64_t i = 0; i < db->n; i++)
            if ((run->unique_bits[i >> 6] >> (i & 63)) & 1ULL) run->table[(size_t)db->vidx[i] * ORC_N_COLS + ORC_C_UNIQUE_KMERS]++;
    memcpy(table, run->table, sizeof(int64_t) * nv * ORC_N_COLS);
    if (dtable) memcpy(dtable, run->dtable, sizeof(double) * nv * ORC_N_DCOLS);
    return 0;
}

static void update_max_counts(int16_t count, int16_t *target, int n) { /* :198-209 */
    for (int j = 0; j < n; j++)
        if (count > target[j]) {
            for (int k = n - 1; k > j; k--) target[k] = target[k - 1];
            target[j] = count;
            return;
        }
}

int orc_match_max_counts(orc_run *run, int16_t *out) {
    const orc_db *db = run->db;
    const int n = run->cfg.max_kmer_res_counts;
    if (n <= 0 || !run->hit_counts) return -1;
    size_t nv = (size_t)db->n_values;
    memset(out, 0, sizeof(int16_t) * (nv + 1) * (size_t)n);
    for (int64_t i = 0; i < db->n; i++) {
        if (!((run->unique_bits[i >> 6] >> (i & 63)) & 1ULL)) continue;
        if (db->parent[db->vidx[i]] == -2) continue; /* taxid == null */
        update_max_counts(run->hit_counts[i], out + (size_t)db->vidx[i] * n, n);
        update_max_counts(run->hit_counts[i], out + nv * (size_t)n, n);
    }
    return 0;
}

/* raw accumulator state of a run, for the multi-rank merge tests: the whole table (ORC_N_COLS columns, the
 * unique column is not meaningful here), and the unique bit vector (one bit per store rank). */
int64_t orc_match_bitmap_words(const orc_run *run) { return (run->db->n + 63) / 64 + 1; }

int orc_match_export(orc_run *run, int64_t *table, uint64_t *bitmap) {
    size_t nv = (size_t)run->db->n_values;
    merge_consumers(run);
    memcpy(table, run->table, sizeof(int64_t) * nv * ORC_N_COLS);
    memcpy(bitmap, run->unique_bits, sizeof(uint64_t) * (size_t)orc_match_bitmap_words(run));
    return 0;
}

int orc_match_import(orc_run *run, const int64_t *table, const uint64_t *bitmap) {
    size_t nv = (size_t)run->db->n_values;
    merge_consumers(run);
    memcpy(run->table, table, sizeof(int64_t) * nv * ORC_N_COLS);
    memcpy(run->unique_bits, bitmap, sizeof(uint64_t) * (size_t)orc_match_bitmap_words(run));
    return 0;
}

static void merge_consumers(orc_run *run) {
    const orc_db *db = run->db;
    size_t nv = (size_t)db->n_values;
    /* merge consumers into the run table */
    for (int t = 0; t < run->n_consumers; t++) {
        consumer_t *c = &run->cons[t];
        for (size_t v = 0; v < nv; v++) {
            int64_t *d = run->table + v * ORC_N_COLS, *s = c->table + v * ORC_N_COLS;
            d[ORC_C_READS] += s[ORC_C_READS];
            d[ORC_C_READS_KMERS] += s[ORC_C_READS_KMERS];
            d[ORC_C_KMERS] += s[ORC_C_KMERS];
            d[ORC_C_CONTIGS] += s[ORC_C_CONTIGS];
            d[ORC_C_CONTIG_LEN_SQ_SUM] += s[ORC_C_CONTIG_LEN_SQ_SUM];
            d[ORC_C_READS_1KMER] += s[ORC_C_READS_1KMER];
            d[ORC_C_READS_BPS] += s[ORC_C_READS_BPS];
            if (s[ORC_C_MAX_CONTIG_LEN] > d[ORC_C_MAX_CONTIG_LEN] ||
                (s[ORC_C_MAX_CONTIG_LEN] == d[ORC_C_MAX_CONTIG_LEN] && s[ORC_C_MAX_CONTIG_LEN] > 0 &&
                 s[ORC_C_MAX_CONTIG_READ_NO] < d[ORC_C_MAX_CONTIG_READ_NO])) {
                d[ORC_C_MAX_CONTIG_LEN] = s[ORC_C_MAX_CONTIG_LEN];
                d[ORC_C_MAX_CONTIG_READ_NO] = s[ORC_C_MAX_CONTIG_READ_NO];
            }
            for (int j = 0; j < ORC_N_DCOLS; j++) run->dtable[v * ORC_N_DCOLS + j] += c->dtable[v * ORC_N_DCOLS + j];
            memset(s, 0, sizeof(int64_t) * ORC_N_COLS);
            s[ORC_C_MAX_CONTIG_READ_NO] = -1;
            memset(c->dtable + v * ORC_N_DCOLS, 0, sizeof(double) * ORC_N_DCOLS);
        }
    }
}

/* Kraken-style segments: the (lastTaxid, contigLen) pairs printKrakenStyleOut receives (:391-394,:452-454) */
int orc_match_segments(const orc_db *db, const uint8_t *read, int read_size, int32_t *codes, int32_t *lens, int cap) {
    const int k = db->k;
    int max = read_size - k + 1, nseg = 0, contig_len = 0, old_index = 0, bad;
    int32_t last = NODE_NULL, node;
    int64_t kmer = -1, rkmer = -1;
    for (int i = 0; i < max; i++) {
        if (kmer == -1) {
            kmer = orc_kmer_straight(read, i, k, &bad);
            if (kmer == -1) {
                old_index = i;
                i = bad;
            } else
                rkmer = orc_kmer_reverse(read, i, k, NULL);
        } else {
            uint8_t lb = read[i + k - 1];
            kmer = orc_next_straight(kmer, lb, k);
            if (kmer == -1) {
                old_index = i;
                i += k - 1;
            } else
                rkmer = orc_next_reverse(rkmer, lb, k);
        }
        node = kmer == -1 ? NODE_INVALID : orc_db_get(db, orc_standard_kmer(kmer, rkmer), NULL);
        if (node != last && contig_len > 0) {
            if (nseg < cap) {
                codes[nseg] = last;
                lens[nseg] = contig_len;
            }
            nseg++;
            contig_len = 0;
        }
        if (node == NODE_INVALID)
            contig_len += i >= max ? max - old_index : i - old_index + 1;
        else
            contig_len++;
        last = node;
    }
    if (contig_len > 0) {
        if (nseg < cap) {
            codes[nseg] = last;
            lens[nseg] = contig_len;
        }
        nseg++;
    }
    return nseg;
}

/* ------------------------------------------------------------------------------------------
 * FASTQ / FASTA parser
 * ---------------------------------------------------------------------------------------- */
typedef struct { const uint8_t *d; size_t len, pos; } lreader;
typedef struct { uint8_t *p; size_t n, cap; } bytebuf;
typedef struct { uint64_t *p; size_t n, cap; } offbuf;

static void bb_reserve(bytebuf *b, size_t extra) {
    if (b->n + extra > b->cap) {
        size_t nc = b->cap ? b->cap : 4096;
        while (b->n + extra > nc) nc *= 2;
        b->p = (uint8_t *)realloc(b->p, nc);
        b->cap = nc;
    }
}
static void ob_push(offbuf *b, uint64_t v) {
    if (b->n == b->cap) {
        b->cap = b->cap ? b->cap * 2 : 1024;
        b->p = (uint64_t *)realloc(b->p, b->cap * sizeof(uint64_t));
    }
    b->p[b->n++] = v;
}

/* BufferedLineReader.nextLine(target, startPos) :160-182 with an unbounded target: appends the next
 * line INCLUDING its '\n' to out (NUL bytes dropped) and returns the number of bytes appended
 * (0 at EOF).  `nextLine() - 1` in the callers is therefore "appended - 1". */
static size_t next_line(lreader *r, bytebuf *out) {
    size_t start = out->n;
    while (r->pos < r->len) {
        uint8_t c = r->d[r->pos++];
        if (c != 0) {
            bb_reserve(out, 1);
            out->p[out->n++] = c;
        }
        if (c == '\n') break;
    }
    return out->n - start;
}

orc_reads *orc_parse_fastq(const uint8_t *data, size_t len, int fasta, int k) {
    orc_reads *res = (orc_reads *)calloc(1, sizeof(*res));
    lreader rd = {data, len, 0};
    bytebuf seq = {0}, desc = {0}, qual = {0}, tmp = {0};
    offbuf so = {0}, dof = {0}, qo = {0};
    ob_push(&so, 0);
    ob_push(&dof, 0);
    ob_push(&qo, 0);
    if (!fasta) {
        /* doReadFastq :288-368.  Sizes are "nextLine() - 1": the '\n' is counted and removed, so a last
         * line without '\n' loses its final byte. */
        for (;;) {
            size_t dstart = desc.n;
            size_t got = next_line(&rd, &desc);
            if (got == 0) break; /* readDescriptorSize = -1 */
            desc.n = dstart + got - 1;
            size_t sstart = seq.n;
            got = next_line(&rd, &seq);
            size_t read_size = got ? got - 1 : 0; /* at EOF Java would run into an exception; stop */
            seq.n = sstart + read_size;
            int eof_err = got == 0;
            for (;;) {
                size_t lstart = seq.n;
                got = next_line(&rd, &seq);
                if (got == 0) {
                    eof_err = 1;
                    break;
                }
                if (seq.p[lstart] == '+') {
                    seq.n = lstart;
                    break;
                }
                seq.n = lstart + got - 1;
            }
            if (eof_err) { /* truncated record: the reference throws; drop the partial record */
                desc.n = dstart;
                seq.n = sstart;
                break;
            }
            read_size = seq.n - sstart;
            /* quality lines until >= read_size chars (:327-341) */
            size_t qstart = qual.n;
            got = next_line(&rd, &qual);
            long qsize = (long)got - 1; /* readProbsSize = nextLine(readProbs) - 1 */
            while (qsize < (long)read_size) {
                long old = qsize;
                /* nextLine(readProbs, readProbsSize): continue writing over the previous '\n' */
                qual.n = qstart + (size_t)(qsize < 0 ? 0 : qsize);
                got = next_line(&rd, &qual);
                qsize = got ? (long)(qual.n - qstart) - 1 : old - 1;
                if (qsize == old - 1) break; /* EOF */
            }
            if (qsize < 0) qsize = 0; /* the reference would fault here (truncated file) */
            qual.n = qstart + (size_t)qsize;
            ob_push(&so, seq.n);
            ob_push(&dof, desc.n);
            ob_push(&qo, qual.n);
            res->n_reads++;
            if ((int64_t)read_size >= k) res->total_kmers += (int64_t)read_size - k + 1;
            res->total_bps += (int64_t)read_size;
        }
    } else {
        /* doReadFasta :375-438: '>' line => descriptor with byte 0 rewritten to '@'; sequence lines
         * concatenated until the next line starting with '>' or EOF. */
        size_t got = next_line(&rd, &tmp);
        int have = got > 0;
        size_t dlen = got ? got - 1 : 0;
        while (have) {
            size_t dstart = desc.n;
            bb_reserve(&desc, dlen + 1);
            memcpy(desc.p + desc.n, tmp.p, dlen);
            desc.n += dlen;
            if (dlen > 0) desc.p[dstart] = '@';
            size_t sstart = seq.n;
            have = 0;
            for (;;) {
                size_t lstart = seq.n;
                got = next_line(&rd, &seq);
                if (got == 0) break; /* EOF */
                if (seq.p[lstart] == '>') {
                    /* next header: the reference copies newSize - readSize bytes, i.e. without '\n' */
                    tmp.n = 0;
                    bb_reserve(&tmp, got);
                    memcpy(tmp.p, seq.p + lstart, got - 1);
                    dlen = got - 1;
                    seq.n = lstart;
                    have = 1;
                    break;
                }
                seq.n = lstart + got - 1;
            }
            size_t read_size = seq.n - sstart;
            ob_push(&so, seq.n);
            ob_push(&dof, desc.n);
            ob_push(&qo, qual.n); /* readProbsSize = -1: no qualities */
            res->n_reads++;
            if ((int64_t)read_size >= k) res->total_kmers += (int64_t)read_size - k + 1;
            res->total_bps += (int64_t)read_size;
        }
    }
    free(tmp.p);
    bb_reserve(&seq, 1);
    bb_reserve(&desc, 1);
    bb_reserve(&qual, 1);
    res->seq = seq.p;
    res->seq_off = so.p;
    res->desc = desc.p;
    res->desc_off = dof.p;
    res->qual = qual.p;
    res->qual_off = qo.p;
    return res;
}

void orc_reads_free(orc_reads *r) {
    if (!r) return;
    free(r->seq);
    free(r->seq_off);
    free(r->desc);
    free(r->desc_off);
    free(r->qual);
    free(r->qual_off);
    free(r);
}


/* =====================================================================================================
 * DB construction (test infrastructure like everything in this file): FillDBGoal + DBGoal, one region after the other
 * ===================================================================================================== */
struct orc_build {
    int k, lower, step, max_dust;
    int32_t n_values;
    int32_t *parent, *depth;
    int64_t *kmers;   /* entries in insertion order, sorted by orc_build_optimize */
    int32_t *vals;
    int64_t n, cap;
    int64_t *set;     /* open addressing over the stored k-mers (putLong's duplicate check, exact); -1 = empty */
    int64_t set_cap;
    int sorted;
};

static int32_t build_depth(const int32_t *parent, int32_t v) {
    int32_t d = 0;
    while (parent[v] >= 0) {
        v = parent[v];
        d++;
    }
    return d;
}

int32_t orc_taxtree_lca(int32_t n_values, const int32_t *parent_vi, int32_t a, int32_t b) {
    (void)n_values;
    if (a == b) return a;       /* :162-164 */
    if (a < 0 || b < 0) return -1; /* :165-167 */
    int32_t da = build_depth(parent_vi, a), db = build_depth(parent_vi, b);
    while (da > db) {           /* :175-177 */
        a = parent_vi[a];
        da--;
    }
    while (db > da) {           /* :178-180 */
        b = parent_vi[b];
        db--;
    }
    while (a != b) {            /* :181-184 (parent of a root = null) */
        a = a >= 0 ? parent_vi[a] : -1;
        b = b >= 0 ? parent_vi[b] : -1;
        if (a < 0 || b < 0) return a == b ? a : -1;
    }
    return a;
}

orc_build *orc_build_begin(int k, int32_t n_values, const int32_t *parent_vi, int lower_case_bases, int step_size) {
    return orc_build_begin_dust(k, n_values, parent_vi, lower_case_bases, step_size, -1);
}

orc_build *orc_build_begin_dust(int k, int32_t n_values, const int32_t *parent_vi, int lower_case_bases, int step_size, int max_dust) {
    orc_build *b = (orc_build *)calloc(1, sizeof(orc_build));
    b->max_dust = max_dust;
    b->k = k;
    b->lower = lower_case_bases;
    b->step = step_size;
    b->n_values = n_values;
    b->parent = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_values);
    memcpy(b->parent, parent_vi, sizeof(int32_t) * (size_t)n_values);
    b->cap = 1024;
    b->kmers = (int64_t *)malloc(sizeof(int64_t) * (size_t)b->cap);
    b->vals = (int32_t *)malloc(sizeof(int32_t) * (size_t)b->cap);
    b->set_cap = 4096;
    b->set = (int64_t *)malloc(sizeof(int64_t) * (size_t)b->set_cap);
    for (int64_t i = 0; i < b->set_cap; i++) b->set[i] = -1;
    return b;
}

static uint64_t build_mix(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    return x;
}

/* 1: newly inserted, 0: was there */
static int build_set_put(orc_build *b, int64_t kmer) {
    if ((b->n + 1) * 2 > b->set_cap) {
        const int64_t nc = b->set_cap * 2;
        int64_t *ns = (int64_t *)malloc(sizeof(int64_t) * (size_t)nc);
        for (int64_t i = 0; i < nc; i++) ns[i] = -1;
        for (int64_t i = 0; i < b->set_cap; i++)
            if (b->set[i] >= 0) {
                uint64_t h = build_mix((uint64_t)b->set[i]) & (uint64_t)(nc - 1);
                while (ns[h] >= 0) h = (h + 1) & (uint64_t)(nc - 1);
                ns[h] = b->set[i];
            }
        free(b->set);
        b->set = ns;
        b->set_cap = nc;
    }
    uint64_t h = build_mix((uint64_t)kmer) & (uint64_t)(b->set_cap - 1);
    while (b->set[h] >= 0) {
        if (b->set[h] == kmer) return 0;
        h = (h + 1) & (uint64_t)(b->set_cap - 1);
    }
    b->set[h] = kmer;
    return 1;
}

/* one region through CGATLongBuffer.put + AbstractStoreFastaReader.dataLine; handle(b, kmer, node) per taken k-mer */
static void build_region(orc_build *b, const uint8_t *seq, int64_t len, int32_t node, void (*handle)(orc_build *, int64_t, int32_t)) {
    const int k = b->k;
    int64_t kmer = 0, rev = 0;      /* reset(): :263-268 */
    int bp_counter = 0, filled = 0;
    int64_t bps_in_region = 0;      /* AbstractRefSeqFastaReader.java:149 */
    /* the streaming low-complexity score (maxDust >= 0): CGATLongBuffer.java:96-110 (weights), :149-228 (put), :263-281 (reset) */
    const int dust = b->max_dust >= 0;
    int diff[32], srl0b[32], srl1b[32], srl2b[32];
    int d = 0, srl0 = 0, srl1 = 0, srl2 = 0, l1 = -1, l2 = -1, l3 = -1;
    for (int i = 0; i < 32; i++) {
        diff[i] = i < 3 ? 1 : diff[i - 1] + diff[i - 2];
        srl0b[i] = srl1b[i] = srl2b[i] = 0;
    }
    const int64_t mask = k == 32 ? -1 : (((int64_t)1 << (2 * k)) - 1);
    for (int64_t i = 0; i < len; i++) {
        uint8_t c = seq[i];
        if (b->lower) {             /* CGAT.cgatToUpperCase, C/util/CGAT.java:91-99 */
            if (c == 'a') c = 'A';
            else if (c == 'c') c = 'C';
            else if (c == 'g') c = 'G';
            else if (c == 't') c = 'T';
        }
        int bp = c == 'C' ? 0 : c == 'G' ? 1 : c == 'A' ? 2 : c == 'T' ? 3 : -1;  /* CGAT_JUMP_TABLE */
        if (bp < 0) {               /* :141-144 */
            kmer = rev = 0;
            bp_counter = 0;
            filled = 0;
            d = srl0 = srl1 = srl2 = 0;
            l1 = l2 = l3 = -1;
            for (int j = 0; j < k; j++) srl0b[j] = srl1b[j] = srl2b[j] = 0;
        } else {
            kmer = ((kmer << 2) & mask) | (int64_t)bp;                                  /* :146 */
            rev = (int64_t)((uint64_t)rev >> 2) | ((int64_t)(bp ^ 1) << (2 * (k - 1)));  /* :147 */
            if (dust) {
                if ((int)c == l1) {  /* :149-159 */
                    int pos = bp_counter - 1 - srl0;
                    if (pos < 0) pos += k;
                    srl0b[pos]++;
                    d += diff[srl0];
                    if (srl0 < k - 1) srl0++;
                } else
                    srl0 = 0;
                if ((int)c == l2) {  /* :160-172 */
                    int pos = bp_counter - 2 - srl1;
                    if (pos < 0) pos += k;
                    srl1b[pos]++;
                    d += diff[srl1];
                    if (srl1 < k - 2) srl1++;
                } else
                    srl1 = 0;
                if ((int)c == l3) {  /* :173-185 */
                    int pos = bp_counter - 3 - srl2;
                    if (pos < 0) pos += k;
                    srl2b[pos]++;
                    d += diff[srl2];
                    if (srl2 < k - 3) srl2++;
                } else
                    srl2 = 0;
                l3 = l2;
                l2 = l1;
                l1 = (int)c;
            }
            const int old_bp = bp_counter;
            bp_counter++;                                                                /* :196-201 */
            if (bp_counter == k) {
                bp_counter = 0;
                filled = 1;
            }
            if (filled && dust) {                                                        /* :202-227 */
                int oc = srl0b[old_bp];
                srl0b[old_bp] = 0;
                if (oc > 0) {
                    d -= diff[oc - 1];
                    srl0b[bp_counter] = oc - 1;
                }
                oc = srl1b[old_bp];
                srl1b[old_bp] = 0;
                if (oc > 0) {
                    d -= diff[oc - 1];
                    srl1b[bp_counter] = oc - 1;
                }
                oc = srl2b[old_bp];
                srl2b[old_bp] = 0;
                if (oc > 0) {
                    d -= diff[oc - 1];
                    srl2b[bp_counter] = oc - 1;
                }
            }
        }
        bps_in_region++;                                   /* AbstractStoreFastaReader.java:102 */
        if (bps_in_region % b->step == 0 && filled) {      /* :103-104 */
            if (dust && d > b->max_dust) continue;         /* isDust(): dustCounter++, not stored (:105-106) */
            handle(b, kmer > rev ? kmer : rev, node);      /* getStandardKMer: CGAT.standardKMer */
        }
    }
}

static void build_put(orc_build *b, int64_t kmer, int32_t node) { /* KMerSortedArray.putLong :168-202 */
    if (!build_set_put(b, kmer)) return;
    if (b->n == b->cap) {
        b->cap *= 2;
        b->kmers = (int64_t *)realloc(b->kmers, sizeof(int64_t) * (size_t)b->cap);
        b->vals = (int32_t *)realloc(b->vals, sizeof(int32_t) * (size_t)b->cap);
    }
    b->kmers[b->n] = kmer;
    b->vals[b->n] = node;
    b->n++;
}

void orc_build_fill(orc_build *b, const uint8_t *seq, const uint64_t *offsets, const int32_t *node_vi, int64_t n_regions) {
    for (int64_t r = 0; r < n_regions; r++)
        build_region(b, seq + offsets[r], (int64_t)(offsets[r + 1] - offsets[r]), node_vi[r], build_put);
}

static int64_t *g_sort_keys;
static int build_cmp(const void *x, const void *y) {
    const int64_t a = g_sort_keys[*(const int64_t *)x], c = g_sort_keys[*(const int64_t *)y];
    return a < c ? -1 : a > c;
}

int64_t orc_build_optimize(orc_build *b) {
    int64_t *idx = (int64_t *)malloc(sizeof(int64_t) * (size_t)(b->n ? b->n : 1));
    for (int64_t i = 0; i < b->n; i++) idx[i] = i;
    g_sort_keys = b->kmers;
    qsort(idx, (size_t)b->n, sizeof(int64_t), build_cmp);
    int64_t *nk = (int64_t *)malloc(sizeof(int64_t) * (size_t)(b->n ? b->n : 1));
    int32_t *nv = (int32_t *)malloc(sizeof(int32_t) * (size_t)(b->n ? b->n : 1));
    for (int64_t i = 0; i < b->n; i++) {
        nk[i] = b->kmers[idx[i]];
        nv[i] = b->vals[idx[i]];
    }
    free(idx);
    free(b->kmers);
    free(b->vals);
    b->kmers = nk;
    b->vals = nv;
    b->cap = b->n ? b->n : 1;
    b->sorted = 1;
    return b->n;
}

static void build_update(orc_build *b, int64_t kmer, int32_t node) { /* KMerSortedArray.update + DBGoal's provider :233-256 */
    int64_t lo = 0, hi = b->n - 1;
    while (lo <= hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (b->kmers[mid] < kmer)
            lo = mid + 1;
        else if (b->kmers[mid] > kmer)
            hi = mid - 1;
        else {
            const int32_t l = orc_taxtree_lca(b->n_values, b->parent, b->vals[mid], node);
            if (l >= 0) b->vals[mid] = l;  /* lcaNode != null ? lcaNode.getTaxId() : oldValue */
            return;
        }
    }
}

void orc_build_update(orc_build *b, const uint8_t *seq, const uint64_t *offsets, const int32_t *node_vi, int64_t n_regions) {
    for (int64_t r = 0; r < n_regions; r++)
        build_region(b, seq + offsets[r], (int64_t)(offsets[r + 1] - offsets[r]), node_vi[r], build_update);
}

void orc_build_fetch(const orc_build *b, int64_t *kmers, int32_t *value_idx) {
    memcpy(kmers, b->kmers, sizeof(int64_t) * (size_t)b->n);
    memcpy(value_idx, b->vals, sizeof(int32_t) * (size_t)b->n);
}

void orc_build_destroy(orc_build *b) {
    if (!b) return;
    free(b->parent);
    free(b->depth);
    free(b->kmers);
    free(b->vals);
    free(b->set);
    free(b);
}


/* the streaming score after the bytes of `s` (a fresh buffer of size k): CGATLongBuffer.getDustValue, for the known answers of
 * T/util/CGATLongBufferTest.java:56-108.  -1 while the buffer is not filled... no: the reference reports d as it stands. */
static void dust_probe_handle(orc_build *b, int64_t kmer, int32_t node) {
    (void)kmer;
    (void)node;
    b->n++; /* counts the k-mers that passed */
}

int64_t orc_dust_passed(int k, int max_dust, const uint8_t *s, int64_t len) {
    const int32_t parent[1] = {-1};
    orc_build *b = orc_build_begin_dust(k, 1, parent, 0, 1, max_dust);
    build_region(b, s, len, 0, dust_probe_handle);
    const int64_t n = b->n;
    b->n = 0;
    orc_build_destroy(b);
    return n;
}
