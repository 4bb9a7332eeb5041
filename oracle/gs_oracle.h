/*
 * gs_oracle.h -- CPU restatement of Genestrip's `match` / `filter` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported CPU baseline.
 *
 * Parity status: PINNED by the reference's own known-answer tests and
 * fixtures -- see tests/test_oracle_golden.py.
 * The reference (Java 11, Maven) cannot be compiled or run in the build
 * container (no JDK), so there is no oracle/_ref build.
 *
 * Citations are relative to /root/reference/, with
 *   C/ = core/src/main/java/org/metagene/genestrip/
 *   B/ = base/src/main/java/org/metagene/genestrip/
 */
#ifndef GS_ORACLE_H
#define GS_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- java.util.Random (pins Bloom seeds, K10, and K3 reads) ---- */
typedef struct { uint64_t seed; } orc_jrandom;
void orc_jrandom_init(orc_jrandom *r, int64_t seed);
int32_t orc_jrandom_next(orc_jrandom *r, int bits);
int64_t orc_jrandom_next_long(orc_jrandom *r);
int32_t orc_jrandom_next_int(orc_jrandom *r, int32_t bound);

/* ---- 2-bit codec, C/util/CGAT.java ---- */
int64_t orc_kmer_straight(const uint8_t *seq, int start, int k, int *bad_pos);  /* :159-180 */
int64_t orc_kmer_reverse(const uint8_t *seq, int start, int k, int *bad_pos);   /* :245-265 */
int64_t orc_next_straight(int64_t kmer, uint8_t bp, int k);                     /* :208-214 */
int64_t orc_next_reverse(int64_t kmer, uint8_t bp, int k);                      /* :226-232 */
int64_t orc_standard_kmer(int64_t straight, int64_t reverse);                   /* :145-147 */
int64_t orc_kmer_canonical(const uint8_t *seq, int start, int k, int *bad_pos); /* :127-131 */

/* ---- Bloom filters, C/bloom/ ---- */
enum { ORC_BLOOM_XOR = 0, ORC_BLOOM_MURMUR = 1, ORC_BLOOM_BLOCKED = 2 };
typedef struct orc_bloom orc_bloom;
/* XOR / Murmur: sized from (expected_insertions, fpp) as AbstractKMerBloomFilter.java:172-185,
 * hash factors from Random(42) (:78,:105-109).  Blocked: bits_per_key 10, seed Random(42).nextLong()
 * (BlockedKMerBloomFilter.java:92,:201-219); fpp ignored. */
orc_bloom *orc_bloom_create(int kind, int64_t expected_insertions, double fpp);
void orc_bloom_destroy(orc_bloom *b);
void orc_bloom_put(orc_bloom *b, int64_t key);
void orc_bloom_put_many(orc_bloom *b, const int64_t *keys, int64_t n);
void orc_bloom_put_many_mt(orc_bloom *b, const int64_t *keys, int64_t n, int threads);
int orc_bloom_contains(const orc_bloom *b, int64_t key);
int orc_bloom_kind(const orc_bloom *b);
int64_t orc_bloom_bits(const orc_bloom *b);      /* XOR/Murmur: #bits ; Blocked: #buckets */
int32_t orc_bloom_hashes(const orc_bloom *b);    /* XOR/Murmur only */
const int64_t *orc_bloom_hash_factors(const orc_bloom *b); /* Blocked: pointer to the seed */
const uint64_t *orc_bloom_words(const orc_bloom *b);
int64_t orc_bloom_n_words(const orc_bloom *b);
int64_t orc_murmur_hash64(int64_t data, int64_t hash_base); /* C/util/MurmurHash3DropIn.java:60-87 */

/* FastqBloomFilter.isAcceptRead, C/bloom/FastqBloomFilter.java:120-161 */
int orc_filter_accept_read(const orc_bloom *b, int k, int min_pos_count, double positive_ratio,
                           const uint8_t *read, int read_size);
/* whole batch; accept[n] gets 0/1; threads<=1 => sequential */
void orc_filter_batch(const orc_bloom *b, int k, int min_pos_count, double positive_ratio,
                      const uint8_t *seq, const uint64_t *offsets, int64_t n_reads,
                      uint8_t *accept, int threads);

/* ---- k-mer store + tree (KMerSortedArray.getLong, SmallTaxTree) ---- */
typedef struct orc_db orc_db;
/* kmers_sorted ascending & distinct (Java canonical encoding); value_idx[i] in [0,n_values).
 * parent_vi[v]: value index of the parent node, -1 for the root, -2 if value v has no tree node
 * (convertKMerStore maps it to null => behaves as a miss, Database.java:136-143).
 * parent_vi may be NULL when the tree is not needed (classification off): every value is a node.
 * bloom_gate != 0 builds the Blocked Bloom pre-filter gate (KMerSortedArray.java:299-301). */
orc_db *orc_db_create(int k, int64_t n_entries, const int64_t *kmers_sorted, const int32_t *value_idx,
                      int32_t n_values, const int32_t *parent_vi, int bloom_gate);
/* The opt-in RadixKMerStore layout (C/store/RadixKMerStore.java): kmers distinct, in ANY order (putLong order);
 * radix_bits in [16, 30] (:91-93), n_values <= orc_radix_max_values(radix_bits) (:160-164); NULL otherwise.
 * getLong follows :369-412: bucket = low radix_bits bits, null bucket => miss before the filter, binary search
 * over the remaining bits, pos = bucketOffset[radix] + local position. */
orc_db *orc_db_create_radix(int k, int radix_bits, int64_t n_entries, const int64_t *kmers, const int32_t *value_idx,
                            int32_t n_values, const int32_t *parent_vi, int bloom_gate);
int32_t orc_radix_max_values(int radix_bits);
/* KMerStore.visit order of either layout: (kmer, value index) of positions 0 .. entries-1
 * (KMerSortedArray.java:426-439: ascending k-mers; RadixKMerStore.java:714-729: by bucket, then remaining bits) */
int64_t orc_db_entries(const orc_db *db);
void orc_db_visit(const orc_db *db, int64_t *kmers, int32_t *value_idx);
void orc_db_destroy(orc_db *db);
/* returns value index or -1; *pos gets the store position on a hit (rank in the sorted array / radix position) */
int32_t orc_db_get(const orc_db *db, int64_t kmer, int64_t *pos);
int32_t orc_tree_lca(const orc_db *db, int32_t a, int32_t b);            /* SmallTaxTree.java:263-289 */
int orc_tree_is_ancestor_of(const orc_db *db, int32_t node, int32_t anc); /* :242-252 */

/* ---- matchRead state machine, C/match/FastqKMerMatcher.java:327-535 ---- */
typedef struct {
    int32_t classify;            /* taxTree != null                                   */
    int32_t count_unique;        /* uniqueCounter != null                             */
    int32_t max_paths;           /* maxClassificationPaths (default 10)               */
    int32_t threshold;           /* minKMersForClass (default 1)                      */
    double max_read_tax_err;     /* maxReadTaxErrorCount (default -1)                 */
    double max_read_class_err;   /* maxReadClassErrorCount (default -1)               */
    int32_t max_kmer_res_counts; /* maxKMerResCounts (default 0): per-k-mer hit counters */
    int32_t pad;
} orc_match_cfg;

/* integer table columns, one row per value index */
enum {
    ORC_C_READS = 0, ORC_C_READS_KMERS, ORC_C_KMERS, ORC_C_UNIQUE_KMERS, ORC_C_CONTIGS,
    ORC_C_CONTIG_LEN_SQ_SUM, ORC_C_MAX_CONTIG_LEN, ORC_C_READS_1KMER, ORC_C_READS_BPS,
    ORC_C_MAX_CONTIG_READ_NO, ORC_N_COLS
};
/* double table columns */
enum { ORC_D_ERR_SUM = 0, ORC_D_ERR_SQ_SUM, ORC_D_CLASS_ERR_SUM, ORC_D_CLASS_ERR_SQ_SUM, ORC_N_DCOLS };

/* per-read flag bits */
enum { ORC_F_FOUND = 1,      /* >=1 k-mer hit                                         */
       ORC_F_RETURNED = 2,   /* matchRead return value (=> filtered FASTQ write)      */
       ORC_F_COUNTED = 4 };  /* read passed the class-error gate and was counted      */

typedef struct orc_run orc_run;
orc_run *orc_match_begin(const orc_db *db, const orc_match_cfg *cfg);
/* process n_reads reads; read i = seq[offsets[i] .. offsets[i+1]), readNo = first_read_no + i.
 * class_vi (may be NULL): entry.classNode's value index or -1.  flags (may be NULL): ORC_F_*.
 * threads > 1 uses that many OpenMP threads (integer results are order independent, K4). */
int orc_match_submit(orc_run *run, const uint8_t *seq, const uint64_t *offsets, int64_t n_reads,
                     int64_t first_read_no, int32_t *class_vi, uint8_t *flags, int threads);
/* table: n_values x ORC_N_COLS int64 ; dtable: n_values x ORC_N_DCOLS double (may be NULL) */
int orc_match_finish(orc_run *run, int64_t *table, double *dtable);
void orc_match_destroy(orc_run *run);
/* KMerUniqueCounterBits.getMaxCountsCounts (:173-211): (n_values + 1) x max_kmer_res_counts shorts, last row = total */
int orc_match_max_counts(orc_run *run, int16_t *out);
/* raw accumulator state for the multi-rank merge tests: table n_values x ORC_N_COLS, bitmap one bit per store rank */
int64_t orc_match_bitmap_words(const orc_run *run);
int orc_match_export(orc_run *run, int64_t *table, uint64_t *bitmap);
int orc_match_import(orc_run *run, const int64_t *table, const uint64_t *bitmap);

/* Kraken-style segments of one read (FastqKMerMatcher.java:597-611): writes up to cap (code,len)
 * pairs, code = value index, -1 (miss, "0") or -2 (INVALID, "A"); returns number of segments. */
int orc_match_segments(const orc_db *db, const uint8_t *read, int read_size, int32_t *codes,
                       int32_t *lens, int cap);

/* ---- FASTQ / FASTA parser, C/fastq/AbstractFastqReader.java:288-438 over
 *      B/io/BufferedLineReader.java:114-182 ---- */
typedef struct {
    int64_t n_reads;
    uint8_t *seq;        /* concatenated read bytes            */
    uint64_t *seq_off;   /* n_reads+1                          */
    uint8_t *desc;       /* concatenated descriptors (incl '@')*/
    uint64_t *desc_off;  /* n_reads+1                          */
    uint8_t *qual;       /* concatenated quality bytes         */
    uint64_t *qual_off;  /* n_reads+1                          */
    int64_t total_kmers; /* sum max(0, L-k+1) (:343-346)       */
    int64_t total_bps;
} orc_reads;
/* parse an in-memory (already inflated) FASTQ (fasta=0) or FASTA (fasta=1) byte stream */
orc_reads *orc_parse_fastq(const uint8_t *data, size_t len, int fasta, int k);
void orc_reads_free(orc_reads *r);

/* ---- DB construction: FillDBGoal (C/goals/refseq/FillDBGoal.java:297ff) + store.optimize + DBGoal
 *      (C/goals/refseq/DBGoal.java:188-311) over AbstractStoreFastaReader.dataLine (C/refseq/
 *      AbstractStoreFastaReader.java:87-115) and CGATLongBuffer.put (C/util/CGATLongBuffer.java:137-229, maxDust = -1).
 *      Regions are walked one after the other, as a single reader thread does; putLong's Bloom filter is taken as exact
 *      (no false positives: C/store/KMerSortedArray.java:175-186 would drop a few new k-mers, which ones depends on the
 *      insertion order of the reference's reader threads). ---- */
typedef struct orc_build orc_build;
orc_build *orc_build_begin(int k, int32_t n_values, const int32_t *parent_vi, int lower_case_bases, int step_size);
/* with GSConfigKey maxDust (-1 off): k-mers whose streaming low-complexity score exceeds it are skipped (CGATLongBuffer.isDust) */
orc_build *orc_build_begin_dust(int k, int32_t n_values, const int32_t *parent_vi, int lower_case_bases, int step_size, int max_dust);
/* number of k-mers of `s` (one region) whose score is <= max_dust: for the known answers of T/util/CGATLongBufferTest.java */
int64_t orc_dust_passed(int k, int max_dust, const uint8_t *s, int64_t len);
void orc_build_fill(orc_build *b, const uint8_t *seq, const uint64_t *offsets, const int32_t *node_vi, int64_t n_regions);
int64_t orc_build_optimize(orc_build *b);  /* store.optimize(): sorted, returns the number of entries */
void orc_build_update(orc_build *b, const uint8_t *seq, const uint64_t *offsets, const int32_t *node_vi, int64_t n_regions);
void orc_build_fetch(const orc_build *b, int64_t *kmers, int32_t *value_idx);
void orc_build_destroy(orc_build *b);
/* TaxTree.getLowestCommonAncestor (C/tax/TaxTree.java:160-187) over value indices; -1 = null */
int32_t orc_taxtree_lca(int32_t n_values, const int32_t *parent_vi, int32_t a, int32_t b);

#ifdef __cplusplus
}
#endif
#endif
