/*
 * gsgpu.h -- C ABI of the MI355X-native engine for Genestrip's `match` / `filter` hot path.
 *
 * This is the drop-in boundary.  The reference (pure Java) has no FFI for this
 * path; its seams are Java virtual methods.  Each entry point below names the reference interface it
 * replaces (paths relative to /root/reference, C/ = core/src/main/java/org/metagene/genestrip/).
 * INTEGRATION.md shows the JNI stub + Java subclasses a maintainer would add on the reference side.
 *
 * Conventions
 *  - plain C types only; every function returns 0 (GS_OK) or a negative GS_E_* code; the message of the
 *    last failure on the calling thread is available from gs_last_error().  No exceptions cross the ABI.
 *  - arrays passed to *_create are borrowed for the duration of the call (copied to HBM before return).
 *  - handles are not thread safe; use one submitting thread per handle.  Handles on different devices
 *    (one process per GPU) are independent.
 *  - `mem` says where the batch pointers of a submit call live: GS_MEM_HOST (pageable or pinned host
 *    memory; the library stages it to HBM) or GS_MEM_DEVICE (already resident in HBM of the handle's
 *    device; used by bench.py and by callers that ingest on the GPU).  It applies to inputs and outputs.
 *  - every handle works on its own HIP stream.  Device buffers handed to a call must be COMPLETE: the
 *    caller synchronises the stream that produced them first; outputs are complete after the matching
 *    *_sync / *_finish call.  (genestrip_amd/binding.py waits for torch's current stream for you.)
 */
#ifndef GSGPU_H
#define GSGPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_ABI_VERSION 3

enum {
    GS_OK = 0,
    GS_E_INVALID = -1,   /* bad argument                                   */
    GS_E_NOMEM = -2,     /* host or device allocation failed               */
    GS_E_HIP = -3,       /* a HIP runtime call failed                      */
    GS_E_UNSUPPORTED = -4,
    GS_E_STATE = -5,     /* call order violated                            */
    GS_E_NODEVICE = -6,  /* no usable gfx950 device                        */
    GS_E_IO = -7         /* reading or writing a file failed (host layer)  */
};

enum { GS_MEM_HOST = 0, GS_MEM_DEVICE = 1, GS_MEM_DEVICE_TEXT = 2 /* gs_match_submit_text / gs_filter_submit_text: the text in HBM, the per-read results to host memory */ };

const char *gs_last_error(void);
const char *gs_strerror(int code);

/* The library keeps the device buffers of finished runs (text banks, queues, result arrays: up to GS_DEVICE_CACHE_MB, default
 * 4096) for the next run of the same shape -- allocating and unmapping them is a fifth of a 30 ms file-level call.  This frees
 * what waits (it is also freed, by itself, when an allocation fails).  The reference has no counterpart: its buffers are JVM heap. */
int gs_device_cache_trim(void);
int gs_abi_version(void);
int gs_device_count(int *n);

/* ---------------------------------------------------------------------------------------------------
 * Device-resident k-mer store + taxonomy
 *
 * Replaces, for lookups: KMerStore.getLong (C/store/KMerSortedArray.java:298-349,
 * C/store/RadixKMerStore.java:369-412) and the SmallTaxTree ancestor/LCA queries
 * (C/tax/SmallTaxTree.java:184-289).  Filled from the Java side through KMerStore.visit
 * (C/store/KMerSortedArray.java:426-439): (kmer, valueIndex) in ascending kmer order.
 *
 *   kmers_sorted[n_entries]  canonical k-mers in the reference encoding (CGAT.java:66-74,145-147), distinct, in
 *                            KMerStore.visit order: ascending for the default KMerSortedArray; by radix bucket and
 *                            remaining bits for the opt-in RadixKMerStore (C/store/RadixKMerStore.java:714-729).
 *                            Any order of distinct k-mers is accepted (ascending input skips the distinctness sort).
 *   value_idx[n_entries]     store value index of each k-mer, in [0, n_values)
 *   parent_vi[n_values]      value index of the parent tree node; -1 for the root; -2 if the value has
 *                            no tree node (Database.convertKMerStore maps it to null => k-mer is a miss,
 *                            C/store/Database.java:136-143).  Every tree node has a value index
 *                            (Database.initStoreIndices, C/store/Database.java:107-128).
 *                            NULL => no tree (classification must be off); every value is a node.
 *
 * The device layout (super-k-mer records, overflow table, minimizer gate: genestrip_amd/csrc/gs_layout.h) is laid out ON the
 * device for stores with records (k >= 19, at most 2^21 values): 473 M k-mers in under a second, the same bytes from the same
 * arrays every time (runs on separately built replicas stay mergeable).  GS_BUILD_HOST=1 selects the host builder, which also
 * serves the other stores.
 * ------------------------------------------------------------------------------------------------- */
typedef struct gs_db gs_db;

typedef struct {
    int32_t k;
    int32_t n_values;
    int64_t n_entries;      /* entries handed in                                            */
    int64_t n_stored;       /* entries in the device table (reachable + with a tree node)   */
    int64_t n_buckets;      /* 64-byte buckets of 8 slots (table)                           */
    int64_t table_bytes;
    int32_t max_displacement;
    int32_t value_bits;
    int64_t gate_bytes;     /* L2-resident pre-filter (0 = not built: store too large for it)  */
    int64_t mgate_bytes;    /* minimizer gate (0 = not built: k < 19)                           */
    int64_t rec_bytes;      /* super-k-mer records (0 = not built: k < 19, partition, > 2^21 values) */
    int64_t n_in_records;   /* stored k-mers that live in records (the others are table slots)  */
    int32_t n_stripes;      /* striped store: devices the record table is split over (else 0)  */
    int32_t stripe;         /* ... and which stripe this handle's device holds                  */
    int64_t stripe_bytes;   /* ... and its size                                                  */
} gs_db_info;

int gs_db_create(gs_db **out, int device, int k, int64_t n_entries, const int64_t *kmers_sorted,
                 const int32_t *value_idx, int32_t n_values, const int32_t *parent_vi);
int gs_db_get_info(const gs_db *db, gs_db_info *info);
int gs_db_destroy(gs_db *db);

/* Native store file: the built device image (table, gate, taxonomy arrays), so that a later
 * process loads it straight into HBM instead of going through Java deserialisation (Database.load,
 * C/store/Database.java:265-314) + KMerStore.visit + gs_db_create.  Not portable across layout versions. */
int gs_db_save(gs_db *db, const char *path);
int gs_db_load(gs_db **out, int device, const char *path);

/* ---------------------------------------------------------------------------------------------------
 * match
 *
 * Replaces FastqKMerMatcher.runMatcher / matchRead (C/match/FastqKMerMatcher.java:181-235, :327-535)
 * plus KMerUniqueCounterBits (C/store/KMerUniqueCounterBits.java:117-163).  One gs_run corresponds to
 * one runMatcher call (one output key): stats and the unique bitmap start cleared (:192-193).
 * ------------------------------------------------------------------------------------------------- */
typedef struct gs_run gs_run;

typedef struct {
    int32_t classify;           /* taxTree != null (MatchResultGoal.java:125-127)                */
    int32_t count_unique;       /* GSConfigKey countUniqueKMers (C/GSConfigKey.java:305)         */
    int32_t max_paths;          /* maxClassificationPaths (:350), 1..128                         */
    int32_t threshold;          /* minKMersForClass (:341)                                       */
    double max_read_tax_err;    /* maxReadTaxErrorCount (:328), -1 = off                         */
    double max_read_class_err;  /* maxReadClassErrorCount (:337), -1 = off                       */
    int32_t profile;            /* != 0: record HIP events around the kernels of each submit     */
    int32_t max_kmer_res_counts; /* maxKMerResCounts (:344): > 0 keeps per-k-mer hit counters for gs_match_max_counts */
} gs_match_cfg;

/* integer result table: one row per value index (CountsPerTaxid fields, C/match/CountsPerTaxid.java:127-159) */
enum {
    GS_C_READS = 0,
    GS_C_READS_KMERS,
    GS_C_KMERS,
    GS_C_UNIQUE_KMERS,       /* -1 in every row when count_unique == 0 (FastqKMerMatcher.java:226-230) */
    GS_C_CONTIGS,            /* Java field is int; the table keeps the exact 64-bit count               */
    GS_C_CONTIG_LEN_SQ_SUM,
    GS_C_MAX_CONTIG_LEN,
    GS_C_READS_1KMER,
    GS_C_READS_BPS,
    GS_C_MAX_CONTIG_READ_NO, /* readNo of the first read (file order) with the max contig, -1 if none   */
    GS_N_COLS
};
enum { GS_D_ERR_SUM = 0, GS_D_ERR_SQ_SUM, GS_D_CLASS_ERR_SUM, GS_D_CLASS_ERR_SQ_SUM, GS_N_DCOLS };

/* per-read flags (optional output) */
enum {
    GS_F_FOUND = 1,     /* >= 1 k-mer of the read hit the store                                   */
    GS_F_RETURNED = 2,  /* matchRead's return value: read goes to the filtered FASTQ (:304-307)   */
    GS_F_COUNTED = 4    /* read passed the class-error gate and was added to `reads` (:508-526)   */
};

int gs_match_begin(gs_run **out, gs_db *db, const gs_match_cfg *cfg);

/* One batch of reads: read i = seq[offsets[i] .. offsets[i+1]), readNo = first_read_no + i
 * (AbstractFastqReader.java:343).  class_vi / flags may be NULL; when given they receive, per read,
 * entry.classNode's value index (-1 = null) and GS_F_* bits.  Asynchronous for GS_MEM_DEVICE (call
 * gs_match_sync / gs_match_finish before reading outputs); synchronous for GS_MEM_HOST. */
int gs_match_submit(gs_run *run, const uint8_t *seq, const uint64_t *offsets, int64_t n_reads,
                    int64_t first_read_no, int mem, int32_t *class_vi, uint8_t *flags);
/* The same for reads of ONE length lying back to back (read i = seq[i * read_len .. (i + 1) * read_len)): no offsets array to
 * build, stage or read -- on the device the offsets' round trip in front of every read's bases is gone.  What a host holds after
 * parsing a sequencer's FASTQ (fixed cycles) and what bench.py's resident reads are. */
int gs_match_submit_fixed(gs_run *run, const uint8_t *seq, int32_t read_len, int64_t n_reads, int64_t first_read_no, int mem,
                          int32_t *class_vi, uint8_t *flags);
int gs_match_sync(gs_run *run);
/* The asynchronous form for HOST batches (the producer thread of AbstractFastqReader fills batch i+1 while batch i is
 * on the device): returns as soon as the work is queued -- for page-locked arrays (gs_pinned_alloc) that is at once,
 * and the copy of this batch runs under the kernel of the previous one -- and hands out a ticket.  gs_match_wait(ticket)
 * returns when class_vi / flags of that batch are filled in and seq / offsets may be reused; at most two batches should
 * be under way (a third submit waits for the first on the device).  gs_match_sync / gs_match_finish wait for all. */
int gs_match_submit_async(gs_run *run, const uint8_t *seq, const uint64_t *offsets, int64_t n_reads,
                          int64_t first_read_no, int32_t *class_vi, uint8_t *flags, int64_t *ticket);
int gs_match_wait(gs_run *run, int64_t ticket);

/* Text mode: a chunk of RAW FASTQ text made of whole four-line records; the device finds the records itself
 * (replaces the producer-thread parse of AbstractFastqReader.doReadFastq, C/fastq/AbstractFastqReader.java:288-368,
 * for the common file shape; read i of the chunk is its (4i+2)-th line, '\r' kept, readNo = first_read_no + i).
 * n_lines = number of '\n' in the chunk as counted by the caller: a multiple of 4, the chunk ends with one.
 * The chunk is REFUSED on the device -- nothing of it, nor of any later text chunk, reaches the run's state -- if
 * it is not exactly what the reference would split into one record per four lines: a NUL byte (the reference drops
 * those, B/io/BufferedLineReader.java:176-178), a third line that does not start with '+' (multi-line sequence,
 * :301-308), a quality line shorter than its sequence (:320-341), or a newline count other than n_lines.
 * gs_match_text_status reports the ticket of the first refused chunk (-1: none); the caller re-parses from there
 * with the general parser, calls gs_match_text_clear_error and goes on.  Asynchronous: with GS_MEM_HOST the text
 * must stay untouched until gs_match_text_wait_copy(ticket) returns (use gs_pinned_alloc for overlap); class_vi /
 * flags (n_lines / 4 entries, same memory kind as text, may be NULL) are complete after gs_match_sync.
 * totals[3] = reads, k-mers (sum of max(0, L-k+1)) and bases of all ACCEPTED chunks since begin/reset
 * (AbstractFastqReader.java:343-349).  Chunks are limited to 1 GiB. */
int gs_match_submit_text(gs_run *run, const uint8_t *text, int64_t n_bytes, int64_t n_lines, int mem,
                         int64_t first_read_no, int32_t *class_vi, uint8_t *flags, int64_t *ticket);
/* The same for FASTA (AbstractFastqReader.doReadFasta, C/fastq/AbstractFastqReader.java:375-438): a chunk of raw FASTA
 * text made of whole records -- it starts with a header line ('>'), ends with a newline, n_lines = its newlines,
 * n_records = its header lines (lines whose first byte is '>'), both counted by the caller.  Read i is the concatenation
 * of the sequence lines of record i ('\r' kept; a record without sequence lines is a read of length 0).  The device finds
 * the records with two prefix sums over the lines and gathers the sequences.  Refused (as above, same status calls) if
 * the counts do not match, the chunk does not start with a header, it holds a NUL byte, or ANY line is empty (there the
 * reference's loop looks at a stale byte of its buffer, :386-392: such input belongs to the reference-exact parser).
 * n_records < 2^24 per chunk.  class_vi / flags: n_records entries. */
int gs_match_submit_fasta(gs_run *run, const uint8_t *text, int64_t n_bytes, int64_t n_lines, int64_t n_records, int mem,
                          int64_t first_read_no, int32_t *class_vi, uint8_t *flags, int64_t *ticket);
/* General FASTQ (AbstractFastqReader.doReadFastq, C/fastq/AbstractFastqReader.java:288-368, with sequence and quality over any
 * number of lines): `text` = whole lines (n_lines newlines, the last byte is one) that START with a record's descriptor line.  The
 * device finds the record structure -- per line where the next record would start if one started here, then the orbit of line 0
 * by pointer doubling -- and matches the records that END inside the chunk; *n_records is their number, *consumed_bytes what they
 * cover (*consumed_lines, may be NULL: their lines): the caller puts the rest in front of the next chunk.  The call waits for the
 * structure (not for the match; the text has been copied when it returns).  A chunk with a NUL byte or a record of more than
 * 4096 lines is refused like a malformed four-line chunk: *n_records = -1, nothing is matched, gs_match_text_status /
 * gs_match_text_clear_error as there. */
int gs_match_submit_fastq_ml(gs_run *run, const uint8_t *text, int64_t n_bytes, int64_t n_lines, int mem, int64_t first_read_no,
                             int32_t *class_vi, uint8_t *flags, int64_t *n_records, int64_t *consumed_bytes, int64_t *consumed_lines,
                             int64_t *ticket);
/* (class_vi / flags: optional per-read outputs as for gs_match_submit_text, room for n_lines / 4 + 1 reads.)  After such a chunk:
 * classes[0 .. consumed_lines) = 1 for a record's descriptor line, 2 for its sequence lines, 0 otherwise; with
 * gs_match_text_newlines this is the record geometry for per-read output. */
int gs_match_text_line_classes(gs_run *run, uint8_t *classes);
int gs_match_text_wait_copy(gs_run *run, int64_t ticket);
int gs_match_text_status(gs_run *run, int64_t *failed_ticket, int64_t *first_bad_record, int64_t totals[3]);
int gs_match_text_clear_error(gs_run *run);
/* Files that are read side by side are independent streams of chunks: a refusal in one must not silence the others.
 * The refusal state and the totals exist in 16 banks; submit / status / clear_error work on the selected one (0 at
 * begin).  Use first_read_no to keep the read numbers of the files apart. */
int gs_match_text_select(gs_run *run, int bank);
/* page-locked host memory for the text blocks (so that the H2D copy overlaps the caller's file reads) */
int gs_pinned_alloc(void **p, size_t bytes);
int gs_pinned_free(void *p);

/* table: n_values x GS_N_COLS int64 ; dtable (may be NULL): n_values x GS_N_DCOLS double (sums of doubles
 * accumulate in device order: not bit-reproducible, as with threads > 0 in the reference).  Host pointers. */
int gs_match_finish(gs_run *run, int64_t *table, double *dtable);
/* The reference copies the read's descriptor whenever a contig beats the current maximum of its tax id
 * (CountsPerTaxid.maxContigDescriptor, FastqKMerMatcher.java:401-407).  A host that cannot keep the descriptors of all
 * reads asks after every batch: read_no[n_values] (host) = read number that currently holds the maximum (-1: none).  A
 * maximum only ever moves to a LATER batch's read by beating it, so the descriptor a host copies when the answer falls
 * into the batch it just submitted is, at the end, the one gs_match_finish reports.  Synchronises. */
int gs_match_max_contig_reads(gs_run *run, int64_t *read_no);
int gs_match_reset(gs_run *run); /* same matcher, next key: clears stats + unique bitmap */
int gs_match_destroy(gs_run *run);

/* maxKMerResCounts > 0 (experimental CSV column "max kmer counts"; KMerUniqueCounterBits.getMaxCountsCounts,
 * C/store/KMerUniqueCounterBits.java:173-211): the `max_kmer_res_counts` largest per-k-mer hit counts of every value
 * index (rows 0..n_values-1) and over all values (row n_values), descending, zero padded; counts are Java shorts
 * (they wrap at 2^15 exactly like `++countsVector.shorts[i]`).  out: (n_values + 1) x max_kmer_res_counts int16, host.
 * Not supported in DB-partitioned mode. */
int gs_match_max_counts(gs_run *run, int16_t *out);

/* Multi-GPU (read-sharded) merge hooks: raw device pointers of the run's accumulators so that the host
 * can reduce them over RCCL before gs_match_finish (sum over ranks for `sums`, max for `max_keys`,
 * all-gather + gs_match_or_bitmap for the unique bitmap).
 *   sums      int64  [n_values][GS_N_SUMS]   additive columns
 *   max_keys  int64  [n_values]              (maxContigLen << 40) | (2^40-1 - readNo), 0 = none
 *   dsums     double [n_values][GS_N_DCOLS]
 *   bitmap    uint32 [bitmap_words]          one bit per table slot                                */
enum { GS_S_READS = 0, GS_S_READS_KMERS, GS_S_KMERS, GS_S_CONTIGS, GS_S_CONTIG_LEN_SQ_SUM, GS_S_READS_1KMER,
       GS_S_READS_BPS, GS_N_SUMS };
int gs_match_device_state(gs_run *run, void **sums, void **max_keys, void **dsums, void **bitmap,
                          int64_t *bitmap_words);
/* bitmap |= OR of n_parts device bitmaps laid out back to back at `parts` (each bitmap_words long) */
int gs_match_or_bitmap(gs_run *run, const void *parts, int64_t n_parts);

/* The same merge for runs that live in ONE process -- what a JVM host does with one gs_run per GPU of the node (the
 * reference is a single process: C/fastq/AbstractFastqReader.java:85-104; it has no counterpart of torch.distributed).
 * Every run must work on a replica of the same store (same arrays through gs_db_create / the same store file).  Runs on
 * the same device are reduced by kernels, the devices among each other by RCCL collectives over xGMI (all-reduce SUM of
 * `sums` / `dsums`, all-reduce MAX of `max_keys`, all-gather + OR of the unique bitmaps; librccl is loaded on first
 * use).  Afterwards EVERY run holds the global state: gs_match_finish on any of them returns the table a single run
 * over all the reads would have produced (give the runs disjoint read numbers: first_read_no).  Synchronous. */
int gs_match_merge(gs_run *const *runs, int n_runs);

/* ---------------------------------------------------------------------------------------------------
 * Striped store (BASELINE.json configs[4]: a store that does not fit one GPU).  The super-k-mer
 * record table -- the bulk of a store -- is split into n_stripes runs of consecutive buckets, stripe p in the HBM of one
 * device; gates, overflow table and tree are small and live on every device.  Reads stay on their home GPU and go
 * through the SAME fused kernel as with a replicated store (gs_match_submit* / gs_match_segments): a record line of a
 * foreign stripe is simply loaded over xGMI peer access (64 bytes per minimizer run that passed the gate), there is no
 * exchange step, no materialised key stream and no host synchronisation per batch.  Record lines are never written:
 * the seen bits of a striped store are kept in each run's own bitmap, and gs_match_merge / the torch.distributed merge
 * OR them exactly as for replicas.  With one stripe this IS gs_db_create.  No counterpart in the single-process
 * reference.
 *
 * gs_db_create_striped: all stripes from ONE process (a JVM host with one gs_run per GPU): out[p] = handle on
 *   devices[p] (2..8 devices that can access each other; a device may appear more than once, which puts several stripes
 *   on it -- how the tests rehearse this on one GPU).  The stripes are freed with the last of the handles.
 * gs_db_create_stripe: ONE stripe per process (torch.distributed, one process per GPU): every process builds the layout
 *   from the same arrays and keeps stripe `stripe`; then every process exports its stripe (gs_db_stripe_export: 64 opaque
 *   bytes, a HIP IPC memory handle), the hosts exchange them (all-gather), and each attaches the others
 *   (gs_db_stripe_attach).  gs_match_begin refuses a handle whose stripes are not all attached.  The exporting process
 *   must keep its handle alive while others use the stripe.
 * ------------------------------------------------------------------------------------------------- */
#define GS_MAX_STRIPES 8
#define GS_STRIPE_HANDLE_BYTES 64
int gs_db_create_striped(gs_db **out, const int *devices, int n_stripes, int k, int64_t n_entries, const int64_t *kmers,
                         const int32_t *value_idx, int32_t n_values, const int32_t *parent_vi);
int gs_db_create_stripe(gs_db **out, int device, int n_stripes, int stripe, int k, int64_t n_entries, const int64_t *kmers,
                        const int32_t *value_idx, int32_t n_values, const int32_t *parent_vi);
int gs_db_stripe_export(gs_db *db, void *handle);
int gs_db_stripe_attach(gs_db *db, int stripe, const void *handle);
/* The same from a store file (gs_db_save of the store built ONCE by gs_db_create): nobody rebuilds the layout, every process
 * reads the image, checks it as gs_db_load does and keeps its stripe.  The usual way to bring a big store up. */
int gs_db_load_striped(gs_db **out, const int *devices, int n_stripes, const char *path);
int gs_db_load_stripe(gs_db **out, int device, int n_stripes, int stripe, const char *path);

/* ---------------------------------------------------------------------------------------------------
 * DB construction: the compute core of FillDBGoal + DBGoal -- every k-mer of every genome region of
 * the requested taxa is stored under its tax node (C/goals/refseq/FillDBGoal.java:297ff -> KMerSortedArray.putLong,
 * C/store/KMerSortedArray.java:168-202: the first region that holds a k-mer sets its value), then every region of the whole
 * reference collection updates the k-mers it shares with the store to the lowest common ancestor of the old value and its own
 * node (C/goals/refseq/DBGoal.java:233-256, :299-311 -> KMerStore.update, TaxTree.getLowestCommonAncestor,
 * C/tax/TaxTree.java:160-187).  K-mers are formed as AbstractStoreFastaReader.dataLine does (C/refseq/
 * AbstractStoreFastaReader.java:87-115 over C/util/CGATLongBuffer.java:137-229): per region a window of the last k bases,
 * reset by any byte that is not C, G, A, T (after upper-casing a, c, g, t if lower_case_bases: GSConfigKey lowerCaseBases,
 * default on), taken when (bases of the region so far) % step_size == 0, stored as CGAT.standardKMer.
 *
 *   gs_dbbuild_begin   tree as for gs_db_create (parent_vi: -1 the ONE root, -2 no node); max_dust = GSConfigKey maxDust (-1 off):
 *                      k-mers whose low-complexity score (CGATLongBuffer.getDustValue, the sum of fib(run length) over the runs of
 *                      period 1, 2 and 3 inside the k-mer) exceeds it are skipped in both passes (:105-107).
 *   gs_dbbuild_add     regions = FASTA records without their header and line ends (seq, offsets[n_regions + 1], offsets[0] = 0;
 *                      `mem` says where seq / offsets live), node_vi[n_regions] (host) = value index of each region's node
 *                      (FillDBGoal / DBGoal reworkNode: the host's mapping).  update = 0: a FillDBGoal region (its k-mers are
 *                      stored), update = 1: a DBGoal region (only LCA updates).  Regions count in the order they are added.
 *   gs_dbbuild_finish  one stable radix sort of all (k-mer, region) pairs + one pass over the runs of equal k-mers.
 *   gs_dbbuild_fetch   kmers ascending (the reference's encoding) + value_idx: the arrays gs_db_create takes.
 *
 * Differences to the reference, all of them run-to-run variations of the reference itself: putLong drops a new k-mer
 * when the fill Bloom filter reports a false positive (:175-186) -- which k-mers depends on the insertion order of its
 * reader threads -- and with several reader threads "first" is a race; here every k-mer of a fill region is stored and the
 * first region in the order of the add calls wins.  maxKMersPerTaxid / maxGenomesPerTaxid / a full store (:187-190) are the
 * host's business (they decide which regions are handed in).
 * ------------------------------------------------------------------------------------------------- */
typedef struct gs_dbbuild gs_dbbuild;
int gs_dbbuild_begin(gs_dbbuild **out, int device, int k, int32_t n_values, const int32_t *parent_vi, int lower_case_bases, int max_dust,
                     int step_size);
int gs_dbbuild_add(gs_dbbuild *b, const uint8_t *seq, const uint64_t *offsets, const int32_t *node_vi, int64_t n_regions, int mem,
                   int update);
/* A collection whose pairs do not fit the GPU (40 bytes per genome base at the peak) is built in passes over disjoint ranges of
 * the canonical k-mer: before the first gs_dbbuild_add, keep only the k-mers in [lo, hi); hand the same regions to one builder
 * per range; the results of ascending ranges, one behind the other, are the result of a single build.  (Canonical k-mers are
 * the larger of two strands, so equal shares of the k-mers lie between 4^k * sqrt(i / n): genestrip_amd.binding.kmer_ranges.) */
int gs_dbbuild_set_range(gs_dbbuild *b, uint64_t lo, uint64_t hi);
int gs_dbbuild_finish(gs_dbbuild *b, int64_t *n_kmers);
int gs_dbbuild_fetch(gs_dbbuild *b, int64_t *kmers, int32_t *value_idx);
/* ... or straight into a store on the builder's device, without the arrays ever leaving the GPU: genomes -> gs_dbbuild_add ->
 * gs_dbbuild_finish -> gs_dbbuild_to_db -> gs_match_begin (and gs_db_save for later processes).  Every k-mer of the build must
 * be reachable, which it is by construction (canonical k-mers).  Stores without records: GS_E_UNSUPPORTED (fetch + gs_db_create). */
int gs_dbbuild_to_db(gs_dbbuild *b, gs_db **out);
int gs_dbbuild_destroy(gs_dbbuild *b);

/* ---------------------------------------------------------------------------------------------------
 * DB-partitioned match, the split pipeline of round 1 (kept: it also serves stores without records): the store is split
 * over the GPUs of a node by key hash (gs_db_create_part keeps the keys with (h >> 40) % n_parts == part, h = the library's mixed key), reads stay
 * on their home GPU.  Per batch: gs_match_encode (reads -> h of every k-mer position; ~0 marks a window with a
 * non-CGAT base) -> all-to-all of the keys to their owners -> gs_match_probe_keys on the owner (node = value index,
 * -1 miss, -2 invalid; marks unique k-mers in the owner's table) -> all-to-all back -> gs_match_reduce on the home
 * GPU (the per-read state machine over the node stream).  Unique-k-mer counts of the partitions are disjoint and
 * simply add up.  pos_off[n_reads+1] = exclusive prefix of max(0, L-k+1).  All pointers are DEVICE pointers; the
 * calls are asynchronous on the run's stream (gs_match_sync).  No counterpart in the single-process reference.
 * ------------------------------------------------------------------------------------------------- */
int gs_db_create_part(gs_db **out, int device, int k, int64_t n_entries, const int64_t *kmers_sorted,
                      const int32_t *value_idx, int32_t n_values, const int32_t *parent_vi, int n_parts, int part);
int gs_match_encode(gs_run *run, const uint8_t *seq, const uint64_t *offsets, int64_t n_reads, const uint64_t *pos_off,
                    uint64_t *keys);
int gs_match_probe_keys(gs_run *run, const uint64_t *keys, int64_t n_keys, int32_t *nodes);
/* gs_match_encode + gs_route_keys in one pass, without the 8-byte key per k-mer position in between: the keys of owner o
 * go to send_keys[o * cap ..), send_idx holds their position in the batch (~0: a slot that was handed out to a wave but
 * not used -- its key is the invalid-window sentinel, the owner answers it with -2 and gs_unroute_region skips it); the
 * waves take slots 2048 at a time, so counts[o] (HOST array, slots handed out for owner o) is a multiple of 2048 and at
 * most cap; `nodes` receives the node of every position that is NOT routed (the routed ones are overwritten by
 * gs_unroute_region).  cap: a multiple of 2048; *overflow != 0: some region was too small, nothing of the batch may be
 * used (call gs_match_encode + gs_route_keys instead).  Synchronous. */
int gs_match_encode_route(gs_run *run, const uint8_t *seq, const uint64_t *offsets, int64_t n_reads, const uint64_t *pos_off,
                          int n_parts, int64_t cap, uint64_t *send_keys, uint32_t *send_idx, int32_t *nodes, int64_t *counts,
                          int *overflow);
/* What a caller needs to size `cap`: the waves gs_match_encode_route launches for a batch of n_reads reads on this run's device
 * (each may leave one chunk partly used per owner) and the chunk size in slots (2048).  No device work. */
int gs_match_route_geometry(const gs_run *run, int64_t n_reads, int32_t *n_waves, int32_t *chunk);
/* nodes[idx[i]] = back[i] for the n answers of one owner region (idx ~0 skipped); asynchronous on the run's stream */
int gs_unroute_region(gs_run *run, const uint32_t *idx, const int32_t *back, int64_t n, int32_t *nodes);
int gs_match_reduce(gs_run *run, const uint8_t *seq, const uint64_t *offsets, int64_t n_reads, int64_t first_read_no,
                    const uint64_t *pos_off, const int32_t *nodes, int32_t *class_vi, uint8_t *flags);
/* Routing helpers around the two all-to-alls (device counting sort by owner rank, synchronous):
 * gs_route_keys groups the valid keys by owner: send_keys[n_valid] (owner 0 first), idx[n_valid] = position of each
 * routed key in `keys`, counts[n_parts] (HOST array) = keys per owner; nodes (may be NULL): the positions that are
 * NOT routed get their node right here -- -2 (invalid window, key ~0) or -1 (key ~0 - 1: gs_match_encode found that
 * the store's minimizer gate, which every partition builds over the keys of ALL partitions, rules the k-mer out, so it
 * is a miss without asking its owner).  gs_unroute_nodes scatters the returned nodes back: nodes[idx[i]] = back[i];
 * with keys != NULL it first writes the unrouted positions as above (keys = NULL: gs_route_keys has done it).
 * n_keys < 2^32, n_parts <= 64. */
int gs_route_keys(gs_run *run, const uint64_t *keys, int64_t n_keys, int n_parts, uint64_t *send_keys, uint32_t *idx,
                  int64_t *counts, int32_t *nodes);
int gs_unroute_nodes(gs_run *run, const uint64_t *keys, const uint32_t *idx, const int32_t *back, int64_t n_routed,
                     int32_t *nodes, int64_t n_keys);

/* Kraken-style per-read segments (writeKrakenStyleOut; FastqKMerMatcher.printKrakenStyleOut, :597-611): the maximal
 * runs of equal tax node over the k-mer positions of each read, in read order.  gs_match_segments probes the batch,
 * fills seg_off[n_reads+1] (host array, exclusive prefix of the per-read segment counts) and keeps the segments on
 * the device; gs_match_segments_fetch copies them out: codes[i] = value index, -1 (miss, printed "0") or -2
 * (window with a non-CGAT base, printed "A"), starts[i] = first k-mer position of the run; a run ends where the
 * next one starts (or at L-k+1).  Does not touch the run's statistics.  Outputs are host pointers. */
int gs_match_segments(gs_run *run, const uint8_t *seq, const uint64_t *offsets, int64_t n_reads, int mem,
                      uint64_t *seg_off);
int gs_match_segments_fetch(gs_run *run, int32_t *codes, int32_t *starts);
/* the same for the reads of the most recent text chunk (gs_match_submit_text; it must not have been refused), and the
 * byte offsets of that chunk's newlines (n_lines entries, host memory) -- the record geometry a caller needs to write
 * the filtered FASTQ and the Kraken-style lines from its copy of the raw text.  Both synchronise. */
int gs_match_segments_text(gs_run *run, uint64_t *seg_off);
int gs_match_text_newlines(gs_run *run, uint32_t *newlines);
/* ... and the descriptor LINES (the '@' included) of n of that chunk's records (0-based in the chunk), each into `stride` bytes of
 * host memory, NUL-terminated, cut to fit: what a host that never sees the reads needs for CountsPerTaxid.maxContigDescriptor
 * (C/match/FastqKMerMatcher.java:401-407) once gs_match_max_contig_reads has named the reads.  Synchronises. */
int gs_match_text_descriptors(gs_run *run, const int64_t *records, int32_t n, uint8_t *out, int32_t stride);
/* After a FASTA or general-FASTQ chunk: bounds[0 .. n_records] of its reads in the gathered sequence buffer, i.e. the read
 * lengths (bounds[r + 1] - bounds[r]); waits for the chunk. */
int gs_match_text_read_bounds(gs_run *run, uint64_t *bounds);

/* accumulated device time of the match kernel launches since gs_match_begin (cfg.profile != 0) */
int gs_match_kernel_time(gs_run *run, int64_t *launches, double *total_ms);
/* the device the run's store lives on (for helpers that work beside a run: gs_inflater_create) */
int gs_match_get_device(gs_run *run, int *device);

/* ---------------------------------------------------------------------------------------------------
 * filter
 *
 * Replaces FastqBloomFilter.isAcceptRead (C/bloom/FastqBloomFilter.java:120-161) over a device copy of
 * the reference's index filter: XORKMerBloomFilter / MurmurKMerBloomFilter
 * (C/bloom/AbstractKMerBloomFilter.java:209-216, fields :52-62) or BlockedKMerBloomFilter
 * (C/bloom/BlockedKMerBloomFilter.java:181-199).  The bit array and hash factors are replicated
 * exactly, so false positives are identical.
 *   kind GS_BLOOM_XOR/MURMUR: bits = number of bits, hash_factors[n_hashes]
 *   kind GS_BLOOM_BLOCKED:    bits = number of buckets, hash_factors[0] = seed, n_hashes ignored
 * ------------------------------------------------------------------------------------------------- */
typedef struct gs_bloom gs_bloom;
enum { GS_BLOOM_XOR = 0, GS_BLOOM_MURMUR = 1, GS_BLOOM_BLOCKED = 2 };

int gs_bloom_create(gs_bloom **out, int device, int kind, int64_t bits, int32_t n_hashes,
                    const int64_t *hash_factors, const uint64_t *words, int64_t n_words);
/* The index filter built on the device (BloomIndexGoal, C/goals/refseq/BloomIndexGoal.java:66-113): an XORKMerBloomFilter sized
 * for `expected_insertions` k-mers at false-positive rate `fpp` (AbstractKMerBloomFilter.java:172-185: bits, number of hashes;
 * hash factors = the first nextLong() values of java.util.Random(42), :105-109) with putLong of every k-mer (kmers in host or
 * device memory, the reference's encoding -- e.g. the result of gs_dbbuild for the requested taxa).  Bit for bit the filter the
 * reference builds from the same k-mers.  gs_bloom_get returns geometry and contents (words may be NULL; n_words >= (bits+63)/64). */
int gs_bloom_build(gs_bloom **out, int device, int kind, const int64_t *kmers, int64_t n_kmers, int mem, int64_t expected_insertions,
                   double fpp);
int gs_bloom_get(gs_bloom *bloom, int64_t *bits, int32_t *n_hashes, int64_t *hash_factors, uint64_t *words, int64_t n_words);
int gs_bloom_destroy(gs_bloom *bloom);
/* accept[i] = isAcceptRead(read i) ? 1 : 0.  profile != 0 records kernel time (gs_filter_kernel_time). */
int gs_filter_submit(gs_bloom *bloom, int k, int min_pos_count, double positive_ratio, const uint8_t *seq,
                     const uint64_t *offsets, int64_t n_reads, int mem, uint8_t *accept, int profile);
int gs_filter_sync(gs_bloom *bloom);
/* Text mode of the filter, as gs_match_submit_text: a chunk of raw four-line FASTQ, records found (and checked) on the
 * device.  accept[n_lines / 4] as gs_filter_submit; newlines (may be NULL) receives the byte offset of every '\n' of
 * the chunk, i.e. the record geometry the caller needs to rewrite the accepted reads (FastqBloomFilter.java:92-105).
 * Both live in the same memory kind as text and are complete after gs_filter_sync; a refused chunk yields all-zero
 * accept flags and shows up in gs_filter_text_status.  gs_filter_text_reset clears the refusal (and, if asked, the
 * totals of the accepted chunks). */
int gs_filter_submit_text(gs_bloom *bloom, int k, int min_pos_count, double positive_ratio, const uint8_t *text,
                          int64_t n_bytes, int64_t n_lines, int mem, uint8_t *accept, uint32_t *newlines, int profile,
                          int64_t *ticket);
int gs_filter_text_wait_copy(gs_bloom *bloom, int64_t ticket);
int gs_filter_get_device(gs_bloom *bloom, int *device);
int gs_filter_text_status(gs_bloom *bloom, int64_t *failed_ticket, int64_t *first_bad_record, int64_t totals[3]);
int gs_filter_text_reset(gs_bloom *bloom, int clear_totals);

/* FASTA chunks and general (multi-line) FASTQ chunks through the filter: the same record search as gs_match_submit_fasta /
 * gs_match_submit_fastq_ml (AbstractFastqReader.java:375-438, :301-308), the reads gathered on the device, accept[r] per
 * record in file order.  `newlines` (may be NULL) receives the offset of every newline of the chunk (n_lines of them).
 * gs_filter_submit_fastq_ml reports the records that END in the chunk and the bytes / lines they cover (n_records = -1: refused,
 * see gs_filter_text_status; the caller carries the rest into its next chunk).  Afterwards gs_filter_text_read_bounds copies
 * the n_records + 1 offsets of the gathered reads (read lengths = differences) and gs_filter_text_line_classes the class of
 * every line of a general FASTQ chunk (1 descriptor, 2 sequence, 0 '+' / quality); both wait for the chunk. */
int gs_filter_submit_fasta(gs_bloom *bloom, int k, int min_pos_count, double positive_ratio, const uint8_t *text,
                           int64_t n_bytes, int64_t n_lines, int64_t n_records, int mem, uint8_t *accept, uint32_t *newlines,
                           int64_t *ticket);
int gs_filter_submit_fastq_ml(gs_bloom *bloom, int k, int min_pos_count, double positive_ratio, const uint8_t *text,
                              int64_t n_bytes, int64_t n_lines, int mem, uint8_t *accept, uint32_t *newlines,
                              int64_t *n_records, int64_t *consumed_bytes, int64_t *consumed_lines, int64_t *ticket);
int gs_filter_text_read_bounds(gs_bloom *bloom, uint64_t *bounds);
int gs_filter_text_line_classes(gs_bloom *bloom, uint8_t *classes);
int gs_filter_kernel_time(gs_bloom *bloom, int64_t *launches, double *total_ms);

/* ---------------------------------------------------------------------------------------------------
 * Block-gzip input inflated on the device
 *
 * Replaces java.util.zip.GZIPInputStream in front of the FASTQ parser (B/io/StreamProvider.java:92-100, 148-150) for BGZF
 * files (bgzip / htslib; also what this library writes for .gz outputs): gzip members of at most 64 KiB of text that state
 * their compressed size, so the host lists them without inflating and ships the COMPRESSED bytes; one wave inflates one member
 * (genestrip_amd/csrc/gs_inflate_dev.hip), ISIZE and CRC-32 are checked as GZIPInputStream checks them.
 *
 *   payload_offset / payload_len   the member's DEFLATE stream inside `file` (behind the gzip header, in front of the trailer)
 *   isize / crc32                  the member's trailer: size and CRC-32 of its text
 *
 * gs_inflate_members: one shot, host buffers (status[i]: 0 = ok, else the member's inflate error; may be NULL).
 * gs_inflater_*: the streaming form in front of gs_match_submit_text / gs_filter_submit_text.  Every feed appends the text of a
 * run of members behind what the previous feed left over and returns, as DEVICE memory, the longest prefix made of whole
 * four-line records (n_lines a multiple of 4, n_bytes up to and including that newline); the rest is carried into the next
 * feed.  next_lo / next_hi: the byte range of `file` the NEXT feed will need (0, 0: none) -- it is copied to the device while
 * this feed's kernels run.  The returned pointer stays valid until the NEXT feed (hand it to gs_match_submit_text with
 * GS_MEM_DEVICE and wait for that chunk's copy -- gs_match_text_wait_copy -- before feeding again).  gs_inflater_tail: what is left after the last feed (fewer
 * than four lines, or a last line without newline), for the caller's parser.  A member that does not inflate to its ISIZE and
 * CRC-32 fails the feed with GS_E_INVALID (gs_inflate_last_error).
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    int64_t payload_offset;
    uint32_t payload_len;
    uint32_t isize;
    uint32_t crc32;
    uint32_t reserved;
} gs_inflate_member;
typedef struct gs_inflater gs_inflater;
int gs_inflate_members(int device, const uint8_t *file, const gs_inflate_member *members, int64_t n_members, uint8_t *out,
                       int64_t out_cap, int32_t *status);
int gs_inflater_create(gs_inflater **out, int device);
int gs_inflater_feed(gs_inflater *inf, const uint8_t *file, const gs_inflate_member *members, int64_t n_members, int64_t next_lo,
                     int64_t next_hi, int last, const uint8_t **text, int64_t *n_bytes, int64_t *n_lines, int64_t *tail_bytes);
/* A SINGLE-MEMBER gzip stream (gzip, pigz) inflated on the device: block boundaries found speculatively, segments decoded side by
 * side with markers for the unknown 32 KiB window, windows resolved in a second pass (gs_inflate_dev.hip), CRC-32 and ISIZE checked
 * as java.util.zip.GZIPInputStream does (B/io/StreamProvider.java:92-100).
 * gs_gunzipper_*: the stream in BATCHES of about one segment per wave slot of the device (a file of any size through buffers of a few
 * gigabytes).  gs_gunzipper_next: the text of the next batch in device memory, behind the last `keep_tail` bytes of the text the call
 * before returned (what lay behind the caller's last whole record); the pointer is valid until the next call.  *last: 0 = more
 * batches follow, 1 = the file is through (every member's CRC-32 and ISIZE were right; members behind one another -- cat a.gz b.gz --
 * give their texts behind one another, what is not a member behind the last trailer is ignored as GZIPInputStream ignores it).
 * GS_E_UNSUPPORTED (from open or from any batch): not a stream this
 * path takes from here on (a segment that outgrows its buffer, a block boundary that was a mirage and could not be repaired) -- the
 * caller inflates the rest on the host; GS_E_INVALID: the stream is damaged (bad code, CRC-32 or ISIZE mismatch).  `gz` must stay
 * readable until gs_gunzipper_close, _reopen or _park (a stream of up to 16 GiB is uploaded by a thread of the object's own while the
 * batches are decoded).  gs_gunzipper_info: [0] segments, [1] chunks searched, [2] batches, [3] block starts that were
 * mirages (decoded again).
 * gs_gunzip_plan_device: the whole stream into ONE device buffer (released with gs_gunzip_free; several members: GS_E_UNSUPPORTED);
 * gs_gunzip_device copies it to `out` (tests, tools).  info as gs_gunzipper_info. */
typedef struct gs_gunzipper gs_gunzipper;
int gs_gunzipper_open(gs_gunzipper **out, int device, const uint8_t *gz, int64_t n);
int gs_gunzipper_reopen(gs_gunzipper *g, const uint8_t *gz, int64_t n); /* the same object and its device buffers on another file */
int gs_gunzipper_next(gs_gunzipper *g, int64_t keep_tail, const uint8_t **d_text, int64_t *n_text, int *last);
int gs_gunzipper_info(const gs_gunzipper *g, int64_t info[4]);
int gs_gunzipper_first_span(gs_gunzipper *g, int64_t bytes); /* the first batch takes at most `bytes` of compressed data: first text early (callers with writers) */
int gs_gunzipper_park(gs_gunzipper *g); /* the caller is through with the file: the upload thread is stopped (`gz` may go away), the buffers stay */
int gs_gunzipper_close(gs_gunzipper *g);
int gs_gunzip_plan_device(int device, const uint8_t *gz, int64_t n, uint8_t **d_text, int64_t *n_text, int64_t info[4]);
int gs_gunzip_free(int device, uint8_t *d_text);
int gs_gunzip_device(int device, const uint8_t *gz, int64_t n, uint8_t *out, int64_t out_cap, int64_t *n_text, int64_t info[4]);
/* lines and the offset behind the last newline that closes a four-line record (count a multiple of four) in n bytes of device text */
int gs_text_cut_device(int device, const uint8_t *d_text, int64_t n, int64_t *n_lines, int64_t *cut);
/* n bytes of device memory to host memory (page-locked for speed); synchronous */
int gs_device_fetch(int device, const uint8_t *d_src, uint8_t *out, int64_t n);
int gs_inflater_tail(gs_inflater *inf, uint8_t *out, int64_t cap, int64_t *n);
/* the first n_bytes of the text the LAST gs_inflater_feed returned, copied to host memory (page-locked for speed); synchronous.
 * For callers that need the text on the host as well -- the filter goal's writers (C/bloom/FastqBloomFilter.java:92-105 rewriteInput). */
int gs_inflater_fetch(gs_inflater *inf, uint8_t *out, int64_t n_bytes);
int gs_inflater_reset(gs_inflater *inf);  /* ready for another file; its device and page-locked buffers stay (allocating them costs more than inflating a file) */
int gs_inflater_destroy(gs_inflater *inf);
const char *gs_inflate_last_error(void);

/* ---------------------------------------------------------------------------------------------------
 * The output side on the device (genestrip_amd/csrc/gs_deflate_dev.hip)
 *
 * Replaces, for the records of a four-line text chunk, the per-read writers of the reference: ReadEntry.write
 * (C/fastq/AbstractFastqReader.java:570-584) as called by FastqBloomFilter.nextEntry -> rewriteInput (C/bloom/FastqBloomFilter.java:92-105,
 * AbstractFastqReader.java:465-470) and by FastqKMerMatcher.afterMatch (C/match/FastqKMerMatcher.java:304-307), and the
 * java.util.zip.GZIPOutputStream behind them (B/io/StreamProvider.java getOutputStreamForFile; gzipFastqOutput is the reference's
 * default, C/GSConfigKey.java:155).
 *
 * gs_filter_compact_text / gs_match_compact_text: the records of the handle's most recent four-line chunk (gs_filter_submit_text /
 * gs_match_submit_text with per-read flags; it must not have been refused) that the writer wants -- filter: accept flag set
 * (which != 0) or clear (which == 0: the reference's dump file); match: GS_F_RETURNED set -- rewritten exactly as ReadEntry.write
 * does (descriptor, '\n', read, "\n+\n", the record's quality line if with_probs else '~' x length, '\n'), back to back in a
 * device buffer of the handle's: *d_out (valid until the next compact call with the same `which` and `slot`; slot 0 / 1: two
 * buffers, so that chunk i can be on its way out while chunk i + 1 is gathered), *n_bytes, *n_records.  To be called before the
 * next chunk is submitted on the handle.  Synchronises the handle's stream.
 *
 * gs_deflater_*: n bytes of device text -> block-gzip (BGZF) members in host memory: gzip members (RFC 1952) of at most 63 KiB of
 * text, each with its size in a 'BC' extra subfield (what bgzip writes; GZIPInputStream / zcat read the file as one stream, this
 * library's device inflater takes the members side by side).  One wave per member: run / hash matches chosen per lane, greedy
 * parse, Huffman code from a counting pass over a sample of the call's text, built on the host, shared by the call's members;
 * CRC-32 and ISIZE as GZIPOutputStream writes them.  out_cap >= gs_deflate_bound(n).  No end-of-file block (28 bytes, the
 * caller's: an empty member).  Synchronous.  gs_deflate_host: host text through the same path (tests, tools);
 * gs_deflate_host_reference: the same format from a plain CPU loop over the same tables (a test of the table builder that runs
 * without a device).  gs_deflater_info: [0] members, [1] text bytes, [2] compressed bytes so far.
 * gs_deflater_append / _pending / _flush: chunks of a few MiB are better compressed together (a call costs ~0.6 ms whatever its
 * size): append copies d_text[0, n) behind the text that is waiting on the device (the source is free when it returns), flush
 * compresses what waits (out_cap >= gs_deflate_bound(gs_deflater_pending(d))).
 * ------------------------------------------------------------------------------------------------- */
typedef struct gs_deflater gs_deflater;
int gs_filter_compact_text(gs_bloom *bloom, int which, int with_probs, int slot, const uint8_t **d_out, int64_t *n_bytes, int64_t *n_records);
int gs_match_compact_text(gs_run *run, int with_probs, int slot, const uint8_t **d_out, int64_t *n_bytes, int64_t *n_records);
int gs_deflater_create(gs_deflater **out, int device);
int gs_deflater_pack(gs_deflater *d, const uint8_t *d_text, int64_t n, uint8_t *out, int64_t out_cap, int64_t *n_out);
int gs_deflater_info(const gs_deflater *d, int64_t info[3]);
int gs_deflater_append(gs_deflater *d, const uint8_t *d_text, int64_t n);
int64_t gs_deflater_pending(const gs_deflater *d);
int gs_deflater_flush(gs_deflater *d, uint8_t *out, int64_t out_cap, int64_t *n_out);
int gs_deflater_destroy(gs_deflater *d);
int64_t gs_deflate_bound(int64_t n);
int gs_deflate_host(int device, const uint8_t *text, int64_t n, uint8_t *out, int64_t out_cap, int64_t *n_out);
int gs_deflate_host_reference(const uint8_t *text, int64_t n, uint8_t *out, int64_t out_cap, int64_t *n_out);
const char *gs_deflate_last_error(void);

/* ---------------------------------------------------------------------------------------------------
 * Measurement support (no reference counterpart; not on the data path): the ceilings of the device the kernels run on,
 * measured with small calibration kernels at the match kernel's occupancy (8 waves per SIMD), so that a benchmark prices
 * its kernels against numbers taken in the same process on the same chip (bench.py `roofline`).
 *   out[0] = rate: wave64 instructions per second over the whole device (VALU_PURE: full-rate 32-bit ops; VALU_MIX: the
 *            match kernel's instruction mix; SALU; VALU_SALU: an alternating stream, both kinds counted), wave-level load
 *            instructions per second (VMEM_*: byte loads of 64 consecutive bytes / dword loads of 8 distinct words /
 *            16-byte loads with 8 lanes per 64-byte line / 16-byte loads with a line per lane, all cache resident), or
 *            random 64-byte lines per second from a table of `arg` bytes (RANDOM_LINES)
 *   out[1] = milliseconds of the timed launch, out[2] = instructions / loads / lines it issued, out[3] = compute units */
enum {
    GS_CAL_VALU_PURE = 0, GS_CAL_VALU_MIX = 1, GS_CAL_SALU = 2, GS_CAL_VALU_SALU = 3,
    GS_CAL_VMEM_BYTES = 4, GS_CAL_VMEM_WORDS = 5, GS_CAL_VMEM_SHARED_LINES = 6, GS_CAL_VMEM_SCATTERED = 7,
    GS_CAL_RANDOM_LINES = 8
};
int gs_calibrate(int device, int what, int64_t arg, double out[4]);

#ifdef __cplusplus
}
#endif
#endif
