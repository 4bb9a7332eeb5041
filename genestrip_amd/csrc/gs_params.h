// gs_params.h -- kernel parameter blocks and launch constants shared by gs_kernels.hip and gs_api.cpp
#pragma once
#include <stdint.h>

#include "../../include/gsgpu.h"
#include "gs_layout.h"

#define GS_BLOCK 256   // 4 waves per workgroup
#ifndef GS_NV_LDS
#define GS_NV_LDS 128  // per-taxid counters are privatised in LDS up to this many value indices: 25 values 10.2 ms in LDS against 49 ms with global atomics on 25 hot counters; 211 values 11.5 ms in LDS (the footprint costs occupancy) against 10.7 ms global
#endif
#define GS_NV_TREE_LDS 2048   // up to here the taxonomy arrays (12 B per value) still travel in LDS
#define GS_STAT_REC_MAX_VALUES 10240  // deferred statistics (GsStatRec): at most 16 passes of 640 values over the records

// Deferred statistics of a read whose hit k-mers all carry ONE tax id (the usual case).  When the per-taxid counters do
// not fit the LDS, such a read writes this record instead of ~20 global atomics, and gs_stat_reduce_kernel adds the
// records up in LDS tables of its own (a kernel that has the whole LDS for them).
struct GsStatRec {
    int32_t vi;          // value index, -1: nothing deferred (no hit, several tax ids, long read)
    int32_t contigs;
    int64_t kmers;
    int64_t sq;          // sum of squared contig lengths
    uint64_t max_key;    // (longest contig << 40) | (2^40 - 1 - read number)
    int32_t counted;     // 1: the read was classified to vi and passed the class-error gate
    int32_t read_kmers;
    int32_t read_len;
    int32_t pad;
    double err, cerr;
};

struct GsMatchParams {
    GsDbDev db;
    const uint8_t *seq;
    const uint64_t *off;
    int64_t n_reads;
    int64_t first_read_no;
    int32_t classify, count_unique, max_paths, threshold;
    double max_read_tax_err, max_read_class_err;
    int64_t *sums;         // [n_values][GS_N_SUMS]
    int64_t *max_keys;     // [n_values]
    double *dsums;         // [n_values][GS_N_DCOLS]
    uint32_t *bitmap;      // one bit per table slot
    uint32_t *hit_counts;  // per table slot, or nullptr (maxKMerResCounts == 0)
    int32_t *class_vi;     // optional per read
    uint8_t *flags;        // optional per read
    unsigned int *long_count;  // [0] entries of the queue of reads with more than 128 k-mer positions (chunks of 64 per wave), [1] the long-read kernel's chunk cursor
    uint32_t *long_list;
    // DB-partitioned mode only: node of every k-mer position, looked up by the owning rank (nullptr: probe locally)
    const int32_t *nodes;
    const unsigned long long *pos_off;  // n_reads + 1: first position of read r in `nodes`
    // text mode (gs_match_submit_text): `off` holds (start, end) PAIRS of the in-place sequence lines (off_stride 2,
    // else 1: n_reads + 1 running offsets), and the whole launch is skipped when *skip != 0 (chunk refused by the
    // device-side record scan, gs_text.hip)
    // off_stride 0: NO offsets at all -- every read is fixed_len bytes, read r at r * fixed_len (gs_match_submit_fixed): the offsets'
    // round trip in front of the bases' (14 % of a wave's cycles on a store that does not fit the caches) is gone
    int32_t off_stride;
    int32_t fixed_len;
    int32_t pad0;
    // direct global-atomic counters (n_values > GS_NV_LDS) exist in `stat_copies` copies, one per group of workgroups
    // (blockIdx % stat_copies): the atomics of one tax id are spread over that many cache lines; the copies are
    // folded into copy 0 before anything reads the accumulators
    int32_t stat_copies;
    const uint32_t *skip;
    GsStatRec *stat_recs;  // deferred statistics (global-atomic counters only) or nullptr: room for n_reads + 64 per wave
    unsigned long long *stat_rec_count;  // records handed out so far (waves take them 64 at a time)
    // Reads of huge_min k-mer positions and more (assembled contigs, chromosomes: the reference grows its read buffer for them,
    // AbstractFastqReader.java:593-604) are taken apart over MANY waves: the long-read kernel hands the first huge_slots of a batch
    // to gs_match_huge_kernel (chunks of whole iterations, one wave each; per read: vote counts and first positions per node, the
    // runs at the chunks' ends) and gs_match_huge_finish_kernel puts each read together again (seams, distinct nodes in order of first
    // appearance, classification).  nullptr: off (DB-partitioned mode, batches of short fixed-length reads).
    unsigned int *huge_count;          // reads handed over so far
    uint32_t *huge_list;               // [huge_slots] their numbers in the batch
    struct GsHugeHead *huge_head;      // [huge_slots][GS_HUGE_COPIES]
    struct GsHugeChunk *huge_chunks;   // [huge_slots][GS_HUGE_MAX_CHUNKS]
    uint32_t *huge_cnt;                // [huge_slots][GS_HUGE_COPIES][n_values] positions that hold the node (chunk c votes on copy c mod GS_HUGE_COPIES)
    uint32_t *huge_first;              // [huge_slots][GS_HUGE_COPIES][n_values] its first position (~0: none)
    uint32_t *huge_touch;              // [huge_slots][GS_HUGE_COPIES * n_values] the nodes with a count, in no order, once per copy (GsHugeHead.n_touch of them)
    uint32_t *huge_fold;               // [huge_slots][2][n_values] the copies folded by the finish kernel: votes, first position
    // long_count: three queues x (entries, cursor) -- queue 0 the long-read kernel's, queues 1 / 2 those of gs_match_wide_kernel<3 / 4>
    // (reads of 129 .. 192 / 193 .. 256 positions, where wide_mask says that kernel serves this run); long_list: queue q at q * long_cap
    int64_t long_cap;
    int32_t wide_mask;
    int32_t pad2;
    int32_t huge_slots;                // <= GS_HUGE_SLOTS
    int32_t huge_min;                  // k-mer positions from which a read goes this way
    int32_t huge_chunk_min;            // k-mer positions per chunk at least (a multiple of 128)
    int32_t pad1;
};
#define GS_HUGE_MIN (1 << 15)
#define GS_HUGE_SLOTS 256       // (= GS_BLOCK: one thread per slot where the chunks are counted)
#define GS_HUGE_MAX_CHUNKS 8192
#define GS_HUGE_COPIES 16      // of a read's vote rows and counters (a power of two, at most 64)
#define GS_HUGE_CHUNK_MIN 1024  // a longer read than GS_HUGE_MAX_CHUNKS of these is cut into GS_HUGE_MAX_CHUNKS chunks
struct GsHugeHead {
    unsigned int n_miss, bad_lo, flags, n_touch;  // flags: 1 = some k-mer hit, 2 = a bad base at or behind position max - 1
};
struct GsHugeChunk {
    int32_t head_node, head_len;  // the run the chunk starts with (it may continue the chunk before): node, positions
    int32_t tail_node, tail_len;  // the run that is open at its end; tail_len < 0: the whole chunk is ONE run (head_len positions)
};

// device-side FASTQ record scan (gs_text.hip)
enum { GS_TS_STICKY = 0, GS_TS_CHUNK_ERR = 1, GS_TS_FIRST_BAD = 2, GS_TS_SKIP = 3, GS_TS_FAILED_TICKET = 4, GS_TS_WORDS = 8 };
enum { GS_TE_NUL = 1, GS_TE_COUNT = 2, GS_TE_SHAPE = 4 };

struct GsTextParams {
    const uint8_t *text;   // n_bytes, padded with blanks to a multiple of 4096
    int64_t n_bytes;
    int64_t n_lines;       // newlines the host counted (a multiple of 4)
    uint32_t *tile_count;  // one per 4096-byte tile
    uint32_t *nl;          // n_lines newline offsets
    unsigned long long *off2;          // 2 * (n_lines / 4): (start, end) of every record's sequence line
    unsigned long long *chunk_totals;  // [3] scratch: -, k-mers, bases of this chunk
    unsigned long long *run_totals;    // [3] reads, k-mers, bases accepted so far
    uint32_t *status;      // GS_TS_WORDS words
    int32_t k;
    int32_t pad;
    // FASTA mode (gs_match_submit_fasta): n_records > = 0 header lines expected, reads gathered into fa_seq
    int64_t n_records;                // -1: FASTQ mode
    unsigned long long *fa_scan;      // per line: exclusive prefix inside its block of (header ? 1 << 40 : length)
    unsigned long long *fa_block;     // per block of GS_FA_BLOCK lines: total, then exclusive prefix
    uint32_t *line_dst;               // per line: destination of its bytes in fa_seq, ~0 for header lines
    uint8_t *fa_seq;                  // the sequences of the chunk's records back to back; off2[0 .. n_records] = their bounds
    // general FASTQ (gs_match_submit_fastq_ml: sequence and quality over any number of lines): the record structure is found
    // on the device first, then the FASTA kernels gather the sequence lines under these classes
    const uint8_t *line_class;        // per line: 1 descriptor line, 2 sequence line, 0 anything else; nullptr in the other modes
    uint32_t *ml_next, *ml_plus;      // per line i, IF a record starts there: the line behind its last quality line, its '+' line
    uint32_t *ml_jump_a, *ml_jump_b;  // pointer doubling
    uint8_t *ml_mark;                 // 1: a record starts at this line
    unsigned long long *ml_out;       // [0] complete records [1] lines they cover [2] bytes they cover
};
#define GS_ML_NONE 0xffffffffu
#define GS_ML_TOO_LONG 0xfffffffeu
#define GS_ML_MAX_LINES 4096  // lines one record may span before the chunk is refused (the host parser takes over)
#define GS_FA_BLOCK 256

struct GsFilterParams {
    int32_t kind;           // GS_BLOOM_*
    int32_t k;
    int32_t min_pos_count;
    int32_t n_hashes;
    double positive_ratio;
    uint64_t bits;          // XOR/Murmur: bit count ; Blocked: bucket count
    uint64_t magic;         // unsigned division of a 64-bit value by `bits` (round-up method)
    int32_t magic_shift;    // ceil(log2(bits))
    int32_t pad;
    const unsigned long long *words;
    const int64_t *factors;
    const uint8_t *seq;
    const uint64_t *off;
    int64_t n_reads;
    uint8_t *accept;
    int32_t off_stride;     // 1: running offsets; 2: (start, end) pairs (text mode, gs_text.hip)
    int32_t pad2;
    const uint32_t *skip;   // text mode: the launch does nothing when *skip != 0
};

struct GsSegParams {
    GsDbDev db;
    const uint8_t *seq;
    const uint64_t *off;
    int64_t n_reads;
    uint32_t *seg_count;            // count pass: segments per read
    const unsigned long long *seg_off;  // write pass: exclusive prefix of seg_count
    int32_t *seg_code;
    int32_t *seg_start;
    int32_t off_stride;  // 1: running offsets; 2: (start, end) pairs (text mode)
    int32_t huge_min;    // reads of this many k-mer positions and more are left out (count pass: seg_count = GS_SEG_HUGE) ...
    // ... and come back cut into pieces of whole iterations, one wave each (a chromosome on one wave: 150 ms per 5 Mbp and pass)
    const struct GsSegPiece *pieces;
    struct GsSegPieceOut *piece_out;  // count pass
    int64_t n_pieces;
};
#define GS_SEG_HUGE 0xffffffffu
#define GS_SEG_PIECE_ITERS 32  // iterations (of 128 positions) per piece
struct GsSegPiece {
    uint32_t read;
    int32_t it0, n_iter;  // its iterations
    int32_t carry;        // write pass: the node of the position in front of it (GS_NODE_NONE: the read starts here)
    unsigned long long out_off;  // write pass: where its first run goes, relative to the read's first
};
struct GsSegPieceOut {
    int32_t first_node, last_node;  // of its first and its last position
    uint32_t count, pad;            // runs that start in it, its first position taken as a start
};

struct GsEncodeParams {
    int32_t k;
    int32_t pad;
    const uint8_t *seq;
    const uint64_t *off;
    int64_t n_reads;
    const unsigned long long *pos_off;  // n_reads + 1 (exclusive prefix of max(0, L-k+1))
    unsigned long long *keys;           // gs_mix62(canonical planar key), ~0 for windows with a non-CGAT base,
                                        // ~0 - 1 for k-mers the minimizer gate rules out
    const uint32_t *mgate;              // the store's minimizer gate (covers every partition's keys) or NULL
    uint32_t mgate_bits;
    uint32_t pad2;
};

// fused encode + routing of the DB-partitioned mode (gs_encode_route_kernel): owner o's keys go to
// send_keys[o * cap ..), handed out to the waves in chunks of GS_ROUTE_CHUNK slots; idx = position of the key in the
// batch (~0: a slot that was handed out but not used, its key is the invalid-window sentinel)
#define GS_ROUTE_CHUNK 2048
struct GsRouteParams {
    int32_t n_parts;
    int32_t pad;
    unsigned long long cap;        // slots per owner region (a multiple of GS_ROUTE_CHUNK)
    unsigned long long *cursors;   // [n_parts] slots handed out per owner; [64] = overflow flag
    unsigned long long *send_keys;
    uint32_t *send_idx;
    int32_t *nodes;                // per k-mer position: node of the positions that are not routed
};
