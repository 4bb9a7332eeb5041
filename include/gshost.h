/*
 * gshost.h -- C++ host layer above the C ABI (libgshost.so): the pieces of the reference's Java host that sit
 * directly on either side of the hot path, written natively because the reference's toolchain (JDK) is absent
 * here and its host is compiled code.  It is what a no-JVM deployment (or the test-suite) drives; the Java host of
 * INTEGRATION.md keeps using its own parser/reporter and only needs include/gsgpu.h.
 *
 *   gs_fastq_*            FASTQ / FASTA ingest with the reference's exact record semantics
 *                         (C/fastq/AbstractFastqReader.java:288-438 over B/io/BufferedLineReader.java:114-182;
 *                         gzip by file suffix, B/io/StreamProvider.java:92-100,148-150), batched for gs_match_submit
 *   gs_host_match_files   FastqKMerMatcher.runMatcher (C/match/FastqKMerMatcher.java:181-235): files -> batches ->
 *                         GPU -> per-taxid table, totals, optional filtered FASTQ (:304-307) and Kraken-style
 *                         output (:308-314, :723-756); parsing of batch i+1 overlaps the GPU work of batch i
 *   gs_host_filter_files  FastqBloomFilter.runFilter (C/bloom/FastqBloomFilter.java:80-105)
 *   gs_host_write_csv     MatchingResult.completeResults + ResultReporter.printMatchResult
 *                         (C/match/MatchingResult.java:84-118, C/match/ResultReporter.java:190-279)
 * (C/ = core/src/main/java/org/metagene/genestrip/, B/ = base/src/main/java/org/metagene/genestrip/)
 */
#ifndef GSHOST_H
#define GSHOST_H

#include "gsgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- FASTQ / FASTA reader ---- */
typedef struct gs_fastq gs_fastq;

typedef struct {
    int64_t n_reads;
    const uint8_t *seq;        /* concatenated reads                                   */
    const uint64_t *seq_off;   /* n_reads + 1                                          */
    const uint8_t *desc;       /* descriptors incl. the leading '@' (FASTA: rewritten) */
    const uint64_t *desc_off;
    const uint8_t *qual;       /* quality bytes; FASTA: empty                          */
    const uint64_t *qual_off;
    int64_t first_read_no;     /* 0-based index of the first read within its file      */
} gs_read_batch;

/* fasta: 0 = FASTQ, 1 = FASTA, -1 = decide by suffix (.fasta/.fa/.fna/.fas[.gz|.gzip], C/goals/FastqMapGoal.java:188-201) */
int gs_fastq_open(gs_fastq **out, const char *path, int fasta, int k);
/* next batch of at most max_reads reads / max_bytes sequence bytes; n_reads == 0 at end of file.
 * The pointers stay valid until the next call on the same reader. */
int gs_fastq_next(gs_fastq *r, int64_t max_reads, int64_t max_bytes, gs_read_batch *batch);
/* totals so far: reads, k-mers (sum of max(0, L-k+1)), base pairs (AbstractFastqReader.java:343-349) */
int gs_fastq_totals(const gs_fastq *r, int64_t *reads, int64_t *kmers, int64_t *bps);
int gs_fastq_close(gs_fastq *r);

/* ---- runMatcher ---- */
typedef struct {
    const char *filtered_path;     /* reads with matchRead() == true, rewritten like ReadEntry.write; NULL = off;
                                      gzip when the name ends in .gz/.gzip                                     */
    const char *kraken_out_path;   /* Kraken-style per-read lines; NULL = off                                  */
    int32_t write_all;             /* writeAll: also lines of unclassified reads (C/GSConfigKey.java:315)      */
    const char *const *taxids;     /* taxid string per value index (needed for the Kraken-style output)        */
    int64_t batch_reads;           /* 0 = default (1 Mi reads)                                                 */
    int32_t with_probs;            /* withProbs (C/GSConfigKey.java:364): written reads keep their quality line(s)
                                      (ReadEntry.write, AbstractFastqReader.java:570-584) instead of '~' x length */
    uint8_t *max_contig_desc;      /* NULL, or n_values x max_contig_desc_stride bytes: per value index the name of the read
                                      that holds the longest contig (CountsPerTaxid.maxContigDescriptor, FastqKMerMatcher.java
                                      :401-407: the descriptor behind its first character up to the first blank), NUL-terminated,
                                      cut to fit; empty when there is none.  Costs one synchronisation per chunk.  Reads of FASTA /
                                      multi-line FASTQ chunks that the device parsed leave the name empty.              */
    int32_t max_contig_desc_stride;
} gs_host_match_opts;

typedef struct {
    int64_t reads, kmers, bps;     /* totalReads / totalKMers / totalBPs                                        */
    int64_t filtered_reads;        /* reads written to filtered_path                                            */
    double seconds_total, seconds_parse, seconds_gpu;
} gs_host_totals;

/* processes the files in order with ONE gs_run (begin..finish); table/dtable as gs_match_finish */
int gs_host_match_files(gs_db *db, const gs_match_cfg *cfg, const char *const *paths, int n_paths,
                        const gs_host_match_opts *opts, int64_t *table, double *dtable, gs_host_totals *totals);

/* The same into a run the caller began and will finish (gs_match_begin ... gs_match_finish), per-read outputs included: what a
 * host that keeps ONE run per matcher calls once per runMatcher -- java/src/.../match/GpuFastqKMerMatcher.java: a store allows one
 * unique-counting run at a time, and the matcher's own run is it.  Read numbers run over the files in order, from 0. */
int gs_host_match_run(gs_run *run, gs_db *db, const char *const *paths, int n_paths, const gs_host_match_opts *opts,
                      gs_host_totals *totals);

/* The files of a run into a gs_run that the caller began and will finish -- for one-process-per-GPU runs that share
 * the files of a sample (genestrip_amd/distributed.py: match_files_sharded): every process takes some of the files,
 * merges its run's device state with the others (gs_match_device_state) and finishes.  file_index[n_paths] = position
 * of each file in the global file order; read numbers on the device are (file_index << 32 | read in file), so the
 * max-contig tie-break keeps the global file order; reads_of_file[n_paths] receives the read counts, which turn those
 * numbers into running ones after the merge.  No per-read outputs.  file_index[i] must lie in [0, GS_HOST_MAX_FILE_INDEX)
 * (the max-contig key has 40 bits for the read number: 8 for the file, 32 for the read); GS_E_UNSUPPORTED otherwise. */
#define GS_HOST_MAX_FILE_INDEX 256
int gs_host_match_into(gs_run *run, gs_db *db, const char *const *paths, int n_paths, const int32_t *file_index,
                       int64_t *reads_of_file, gs_host_totals *totals);

/* runMatcher over the files of a sample on several devices of THIS process (what a JVM host with 8 GPUs calls):
 * dbs[d] = a replica of the store on device d (or the handles of ONE striped store, gs_db_create_striped); file i goes to replica i % n_dbs, every replica runs on a thread of its
 * own (as gs_host_match_into), the runs are merged with gs_match_merge (RCCL between devices) and finished once.  The
 * table equals the one gs_host_match_files returns for the same files in the same order.  No per-read outputs. */
int gs_host_match_files_multi(gs_db *const *dbs, int n_dbs, const gs_match_cfg *cfg, const char *const *paths, int n_paths,
                              int64_t *table, double *dtable, gs_host_totals *totals);

/* diagnostics: which = 0: chunks of text that went through the general (multi-line) FASTQ device path of the match goal in this
 * process so far; 1: FASTA / general FASTQ chunks that the filter goal handled on the device */
int64_t gs_host_stat(int which);

/* ---- runFilter: accepted reads -> filtered_path, the rest -> rest_path (either may be NULL); with_probs as above ---- */
int gs_host_filter_files(gs_bloom *bloom, int k, int min_pos_count, double positive_ratio, const char *const *paths,
                         int n_paths, const char *filtered_path, const char *rest_path, int with_probs,
                         gs_host_totals *totals);

/* ---- completeResults + CSV ---- */
typedef struct {
    int32_t n_values;
    const int32_t *parent_vi;      /* as gs_db_create                                                   */
    const int32_t *position;       /* tree sort position per value index (SmallTaxIdNode position); NULL = pre-order */
    const char *const *taxids;
    const char *const *names;      /* may be NULL                                                       */
    const char *const *ranks;      /* Rank.toString() per node, may be NULL                             */
    const int64_t *db_kmers;       /* k-mers stored per value index (Database.getStats)                 */
    int64_t db_kmers_total;
    const char *const *max_contig_desc; /* per value index, may be NULL                                 */
    const int16_t *max_kmer_counts;     /* gs_match_max_counts output ((n_values+1) x max_kmer_res_counts) or NULL */
    int32_t max_kmer_res_counts;        /* > 0 adds the experimental "max kmer counts" column (pos 2001)          */
} gs_host_tax_info;

int gs_host_write_csv(const char *path, const gs_host_tax_info *tax, const int64_t *table, const double *dtable,
                      const gs_host_totals *totals);

/* The host layer keeps page-locked blocks and the device decoders of gzip input (gigabytes of HBM) from call to call; this hands
 * them back (a long-lived JVM host before it loads a big store).  No other thread may be inside the host layer meanwhile. */
int gs_host_release_pools(void);

/* message of the last failure raised inside the host layer itself (failures of the C ABI: gs_last_error) */
const char *gs_host_last_error(void);

/* The ingest path's own gzip decoder (genestrip_amd/csrc/gs_inflate.h; replaces java.util.zip.GZIPInputStream,
 * B/io/StreamProvider.java:92-100) on a memory range, `block` output bytes per decode call -- exposed for the tests:
 * out receives the concatenated members; GS_E_INVALID for a corrupt stream (CRC-32 and ISIZE are checked). */
int gs_host_gunzip(const uint8_t *in, size_t n_in, uint8_t *out, size_t out_cap, size_t *n_out, size_t block);
/* ... and through the multi-threaded decoder of the gzip text path (speculative starts inside the stream, checked and
 * repaired by the in-order resolver): `threads` workers on compressed chunks of `chunk` bytes */
int gs_host_gunzip_parallel(const uint8_t *in, size_t n_in, uint8_t *out, size_t out_cap, size_t *n_out, int threads,
                            size_t chunk, size_t block);

/* Double.toString(double) -- exposed for the tests of the CSV writer */
int gs_host_java_double(double v, char *buf, int cap);

#ifdef __cplusplus
}
#endif
#endif
