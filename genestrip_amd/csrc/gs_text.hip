// gs_text.hip -- device-side FASTQ record scan (host ingest).
//
// The reference parses FASTQ on one producer thread (AbstractFastqReader.doReadFastq, C/fastq/AbstractFastqReader.java
// :288-368 over BufferedLineReader.nextLine, B/io/BufferedLineReader.java:160-182).  For the overwhelmingly common
// shape of a FASTQ file -- four lines per record -- that parse is a newline scan, and a newline scan is streaming work
// the GPU does at HBM speed: the host only reads the file into pinned blocks and cuts them at record boundaries.
//
// A chunk is accepted only if it is EXACTLY what the reference's state machine would turn into one record per four
// lines, so that the in-place sequence lines are the reads the reference would have produced:
//   * no NUL byte (the reference drops them, :176-178 of BufferedLineReader),
//   * line 3 of every record starts with '+' (the reference takes further lines as sequence until one does, :301-308),
//   * the quality line is at least as long as the sequence line (shorter: the reference keeps reading quality
//     lines, :320-341),
//   * the number of newlines equals what the host counted and the text ends with one.
// '\r' stays part of its line exactly as in the reference (and is then an invalid base).  Anything else sets a sticky
// error: this chunk and all later ones are skipped by the match kernel (GsMatchParams::skip), the run's state stays
// untouched, and the host re-parses from the start of the failing chunk -- a file that fails in its very first chunk once more
// through the general FASTQ kernels at the end of this file (records over any number of lines), the rest with the general parser.
//
// Kernels (tiles of 4096 bytes, one uint4 per thread):
//   gs_text_count_kernel    newlines per tile, NUL check                      streaming, 1 B read per byte
//   gs_text_scan_kernel     exclusive prefix over the tile counts (one block)
//   gs_text_lines_kernel    byte offset of every newline -> nl[]               streaming
//   gs_text_records_kernel  one thread per record: checks, (start, end) of the sequence line, totals
//   gs_text_commit_kernel   one thread: sticky error / skip flag / run totals
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "gs_params.h"

typedef unsigned long long u64;

#define GS_TEXT_TILE 4096
#define GS_TEXT_BLOCK 256

// 0x80 in every byte of w that equals c
__device__ __forceinline__ uint32_t gs_eq_bytes(uint32_t w, uint32_t c4) {
    const uint32_t x = w ^ c4;
    return ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;
}

__global__ __launch_bounds__(GS_TEXT_BLOCK) void gs_text_count_kernel(GsTextParams P) {
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const uint4 v = reinterpret_cast<const uint4 *>(P.text)[(size_t)blockIdx.x * GS_TEXT_BLOCK + threadIdx.x];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t c = 0, z = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        c += __popc(gs_eq_bytes(w[i], 0x0a0a0a0au));
        z |= gs_eq_bytes(w[i], 0u);
    }
    if (c) atomicAdd(&s_cnt, c);
    if (z) atomicOr(&P.status[GS_TS_CHUNK_ERR], GS_TE_NUL);
    __syncthreads();
    if (threadIdx.x == 0) P.tile_count[blockIdx.x] = s_cnt;
}

// one block: tile_count -> exclusive prefix (in place); also resets the per-chunk words
__global__ __launch_bounds__(1024) void gs_text_scan_kernel(GsTextParams P, int64_t n_tiles) {
    __shared__ u64 s_part[1024];
    const int t = threadIdx.x;
    // A thread's run of counts is a stream of cache lines of its own (1 024 such streams): with one dword per load every line was asked
    // of L2 sixteen times (0.22 ms for the 131 072 tiles of a 512 MiB slice).  Runs of a multiple of 16 counts, read a line -- four
    // 16-byte loads back to back -- at a time.
    const int64_t per = ((n_tiles + 1023) / 1024 + 15) & ~(int64_t)15;
    const int64_t a = (int64_t)t * per < n_tiles ? (int64_t)t * per : n_tiles, b = a + per < n_tiles ? a + per : n_tiles;
    u64 sum = 0;
    int64_t i0 = a;
    for (; i0 + 16 <= b; i0 += 16) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = reinterpret_cast<const uint4 *>(P.tile_count + i0)[u];
#pragma unroll
        for (int u = 0; u < 4; u++) sum += (u64)v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (int64_t i = i0; i < b; i++) sum += P.tile_count[i];
    s_part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan
        const u64 x = t >= d ? s_part[t - d] : 0;
        __syncthreads();
        s_part[t] += x;
        __syncthreads();
    }
    u64 run = s_part[t] - sum;
    int64_t i1 = a;
    for (; i1 + 16 <= b; i1 += 16) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = reinterpret_cast<const uint4 *>(P.tile_count + i1)[u];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t c0 = v[u].x, c1 = v[u].y, c2 = v[u].z, c3 = v[u].w;
            v[u].x = (uint32_t)run;
            v[u].y = (uint32_t)(run + c0);
            v[u].z = (uint32_t)(run + c0 + c1);
            v[u].w = (uint32_t)(run + c0 + c1 + c2);
            run += (u64)c0 + c1 + c2 + c3;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) reinterpret_cast<uint4 *>(P.tile_count + i1)[u] = v[u];
    }
    for (int64_t i = i1; i < b; i++) {
        const uint32_t c = P.tile_count[i];
        P.tile_count[i] = (uint32_t)run;
        run += c;
    }
    if (t == 1023) {
        if (s_part[1023] != (u64)P.n_lines) atomicOr(&P.status[GS_TS_CHUNK_ERR], GS_TE_COUNT);
        if (P.n_bytes > 0 && P.text[P.n_bytes - 1] != '\n') atomicOr(&P.status[GS_TS_CHUNK_ERR], GS_TE_COUNT);
        P.chunk_totals[0] = P.chunk_totals[1] = P.chunk_totals[2] = 0;
    }
}

__global__ __launch_bounds__(GS_TEXT_BLOCK) void gs_text_lines_kernel(GsTextParams P) {
    __shared__ uint32_t s_wave[GS_TEXT_BLOCK / 64];
    const size_t base = ((size_t)blockIdx.x * GS_TEXT_BLOCK + threadIdx.x) * 16;
    const uint4 v = reinterpret_cast<const uint4 *>(P.text)[(size_t)blockIdx.x * GS_TEXT_BLOCK + threadIdx.x];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t m[4], c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        m[i] = gs_eq_bytes(w[i], 0x0a0a0a0au);
        c += __popc(m[i]);
    }
    // exclusive prefix of c over the block: wave scan + wave totals through LDS
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t x = __shfl_up(inc, d);
        if (lane >= d) inc += x;
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    uint32_t before = P.tile_count[blockIdx.x];
    for (int i = 0; i < wv; i++) before += s_wave[i];
    uint32_t idx = before + inc - c;
    if (c == 0) return;
    if ((u64)idx + c > (u64)P.n_lines) return;  // count mismatch (flagged by the scan kernel): stay inside nl[]
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t mm = m[i];
        while (mm) {
            const int bit = __ffs(mm) - 1;  // 7, 15, 23, 31
            mm &= mm - 1;
            P.nl[idx++] = (uint32_t)(base + 4 * i + (bit >> 3));
        }
    }
}

__global__ __launch_bounds__(GS_TEXT_BLOCK) void gs_text_records_kernel(GsTextParams P) {
    __shared__ u64 s_tot[2];
    if (threadIdx.x < 2) s_tot[threadIdx.x] = 0;
    __syncthreads();
    const int64_t n_rec = P.n_lines >> 2;
    const int64_t r = (int64_t)blockIdx.x * GS_TEXT_BLOCK + threadIdx.x;
    if (r < n_rec && P.status[GS_TS_CHUNK_ERR] == 0) {  // (a count mismatch leaves nl[] partly unwritten)
        const uint32_t e0 = P.nl[4 * r], e1 = P.nl[4 * r + 1], e2 = P.nl[4 * r + 2], e3 = P.nl[4 * r + 3];
        const uint32_t s1 = e0 + 1, s2 = e1 + 1, s3 = e2 + 1;
        const int64_t len = (int64_t)e1 - (int64_t)s1;
        const int64_t qlen = (int64_t)e3 - (int64_t)s3;
        const bool ok = P.text[s2] == '+' && qlen >= len;
        if (!ok) {
            atomicOr(&P.status[GS_TS_CHUNK_ERR], GS_TE_SHAPE);
            atomicMin(&P.status[GS_TS_FIRST_BAD], (uint32_t)r);
        }
        P.off2[2 * r] = s1;
        P.off2[2 * r + 1] = e1;
        if (len >= P.k) atomicAdd(&s_tot[0], (u64)(len - P.k + 1));
        atomicAdd(&s_tot[1], (u64)len);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_tot[0]) atomicAdd(&P.chunk_totals[1], s_tot[0]);
        if (s_tot[1]) atomicAdd(&P.chunk_totals[2], s_tot[1]);
    }
}

// ---------------------------------------------------------------------------------------------------
// FASTA mode (AbstractFastqReader.doReadFasta, C/fastq/AbstractFastqReader.java:375-438): a record is a header line
// (first byte '>') and the sequence lines up to the next header line, concatenated.  '>' never opens a sequence line,
// so the record boundaries are known from the first byte of every line and the whole parse is two prefix sums over the
// LINES of the chunk: headers before a line = its record, sequence bytes before a line = where its bytes go in the
// compacted read buffer.  Accepted only if the chunk is what the reference would split the same way: it starts with a
// header line, holds exactly the header lines the host counted, no NUL byte and NO EMPTY LINE (on an empty line the
// reference's loop looks at a stale byte of its read buffer, :386-392: such input goes to the reference-exact parser).
//   gs_fasta_lines_kernel   per line: header? length; exclusive scan inside blocks of 256 lines
//   gs_fasta_scan_kernel    one block: prefix over the block totals, header count check, off2[n_records]
//   gs_fasta_emit_kernel    per line: record start offsets (off2) / destination of a sequence line
//   gs_fasta_gather_kernel  one wave per sequence line: bytes -> fa_seq
//   gs_fasta_totals_kernel  per record: k-mer and base totals
// ---------------------------------------------------------------------------------------------------
#define GS_FA_HDR (1ULL << 40)
#define GS_FA_LEN_MASK (GS_FA_HDR - 1)

__device__ __forceinline__ void gs_fa_line(const GsTextParams &P, int64_t i, uint32_t &start, uint32_t &len, bool &hdr) {
    start = i ? P.nl[i - 1] + 1 : 0;
    len = P.nl[i] - start;
    if (P.line_class != nullptr) {  // general FASTQ: the classes were found by gs_ml_class_kernel
        const uint8_t c = P.line_class[i];
        hdr = c == 1;
        if (c != 2) len = 0;
        return;
    }
    hdr = len > 0 && P.text[start] == '>';
}

__global__ __launch_bounds__(GS_FA_BLOCK) void gs_fasta_lines_kernel(GsTextParams P) {
    __shared__ u64 s_wave[GS_FA_BLOCK / 64];
    const int64_t i = (int64_t)blockIdx.x * GS_FA_BLOCK + threadIdx.x;
    const bool live = i < P.n_lines && P.status[GS_TS_CHUNK_ERR] == 0;  // (a count mismatch leaves nl[] partly unwritten)
    u64 v = 0;
    if (live) {
        uint32_t start, len;
        bool hdr;
        gs_fa_line(P, i, start, len, hdr);
        if (P.line_class == nullptr && (len == 0 || (i == 0 && !hdr))) {  // (FASTA: no empty lines, a header first)
            atomicOr(&P.status[GS_TS_CHUNK_ERR], GS_TE_SHAPE);
            atomicMin(&P.status[GS_TS_FIRST_BAD], (uint32_t)i);
        }
        v = hdr ? GS_FA_HDR : (u64)len;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    u64 inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u64 x = __shfl_up(inc, d);
        if (lane >= d) inc += x;
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    u64 before = 0;
    for (int w = 0; w < wv; w++) before += s_wave[w];
    if (i < P.n_lines) P.fa_scan[i] = before + inc - v;
    if (threadIdx.x == GS_FA_BLOCK - 1) P.fa_block[blockIdx.x] = before + inc;
}

__global__ __launch_bounds__(1024) void gs_fasta_scan_kernel(GsTextParams P, int64_t n_blocks) {
    __shared__ u64 s_part[1024];
    const int t = threadIdx.x;
    const int64_t per = (n_blocks + 1023) / 1024;
    const int64_t a = (int64_t)t * per, b = a + per < n_blocks ? a + per : n_blocks;
    u64 sum = 0;
    for (int64_t i = a; i < b; i++) sum += P.fa_block[i];
    s_part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const u64 x = t >= d ? s_part[t - d] : 0;
        __syncthreads();
        s_part[t] += x;
        __syncthreads();
    }
    u64 run = s_part[t] - sum;
    for (int64_t i = a; i < b; i++) {
        const u64 c = P.fa_block[i];
        P.fa_block[i] = run;
        run += c;
    }
    if (t == 1023) {
        const u64 total = s_part[1023];
        if ((int64_t)(total >> 40) != P.n_records)
            atomicOr(&P.status[GS_TS_CHUNK_ERR], GS_TE_COUNT);
        else
            P.off2[P.n_records] = total & GS_FA_LEN_MASK;
    }
}

__global__ __launch_bounds__(GS_FA_BLOCK) void gs_fasta_emit_kernel(GsTextParams P) {
    const int64_t i = (int64_t)blockIdx.x * GS_FA_BLOCK + threadIdx.x;
    if (i >= P.n_lines || P.status[GS_TS_CHUNK_ERR] != 0) return;
    uint32_t start, len;
    bool hdr;
    gs_fa_line(P, i, start, len, hdr);
    const u64 pre = P.fa_scan[i] + P.fa_block[blockIdx.x];
    if (hdr) {
        P.off2[pre >> 40] = pre & GS_FA_LEN_MASK;  // (the header count was checked: pre >> 40 < n_records)
        P.line_dst[i] = 0xffffffffu;
    } else {
        P.line_dst[i] = len ? (uint32_t)(pre & GS_FA_LEN_MASK) : 0xffffffffu;  // (general FASTQ: '+' and quality lines carry nothing)
    }
}

__global__ __launch_bounds__(256) void gs_fasta_gather_kernel(GsTextParams P) {
    if (P.status[GS_TS_CHUNK_ERR] != 0) return;
    const int lane = threadIdx.x & 63;
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < P.n_lines; i += n_waves) {
        const uint32_t dst = P.line_dst[i];
        if (dst == 0xffffffffu) continue;
        const uint32_t start = i ? P.nl[i - 1] + 1 : 0, len = P.nl[i] - start;  // (a line with a destination is copied whole)
        for (uint32_t j = (uint32_t)lane; j < len; j += 64) P.fa_seq[(size_t)dst + j] = P.text[(size_t)start + j];
    }
}

__global__ __launch_bounds__(GS_TEXT_BLOCK) void gs_fasta_totals_kernel(GsTextParams P) {
    __shared__ u64 s_tot[2];
    if (threadIdx.x < 2) s_tot[threadIdx.x] = 0;
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * GS_TEXT_BLOCK + threadIdx.x;
    if (r < P.n_records && P.status[GS_TS_CHUNK_ERR] == 0) {
        const int64_t len = (int64_t)(P.off2[r + 1] - P.off2[r]);
        if (len >= P.k) atomicAdd(&s_tot[0], (u64)(len - P.k + 1));
        atomicAdd(&s_tot[1], (u64)len);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_tot[0]) atomicAdd(&P.chunk_totals[1], s_tot[0]);
        if (s_tot[1]) atomicAdd(&P.chunk_totals[2], s_tot[1]);
    }
}

// ---------------------------------------------------------------------------------------------------
// General FASTQ: AbstractFastqReader.doReadFastq (C/fastq/AbstractFastqReader.java:288-368) lets the sequence and the quality of
// a record run over any number of lines: descriptor line (ANY line: no '@' is asked for); one sequence line, then further ones
// until a line STARTS with '+' (:301-308); one quality line, then further ones until they hold at least as many characters as
// the sequence (:320-341).  Where a record starts therefore depends on everything before it -- a '@' may open a quality line
// -- which is a sequential parse on the face of it.  On the device:
//   gs_ml_next_kernel   per line i: IF a record started here, where would the next one start (and where is its '+' line)?  A
//                       handful of line lengths and first bytes per line, every line at once
//   gs_ml_orbit_*       the records of the chunk are the orbit of line 0 under that map: pointer doubling marks it in
//                       log2(lines) rounds (one small launch per round)
//   gs_ml_class_kernel  every marked line classifies the lines of its record (descriptor / sequence / other) and counts it;
//                       the first marked line whose record does not end inside the chunk is where the next chunk has to start
// and then the FASTA kernels above gather the sequence lines of every record into one read (gs_fa_line reads the classes).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t gs_ml_start(const GsTextParams &P, uint32_t i) { return i ? P.nl[i - 1] + 1 : 0; }
__device__ __forceinline__ uint32_t gs_ml_len(const GsTextParams &P, uint32_t i) { return P.nl[i] - gs_ml_start(P, i); }

__global__ __launch_bounds__(256) void gs_ml_next_kernel(GsTextParams P) {
    const uint32_t n = (uint32_t)P.n_lines;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n || P.status[GS_TS_CHUNK_ERR] != 0) return;
    uint32_t next = GS_ML_NONE, plus = GS_ML_NONE;
    const uint32_t s = i + 1;  // the first sequence line, whatever it starts with
    if (s < n) {
        u64 L = gs_ml_len(P, s);
        uint32_t p = s + 1, steps = 0;
        for (; p < n; p++, steps++) {
            if (steps == GS_ML_MAX_LINES) break;
            const uint32_t st = gs_ml_start(P, p);
            if (P.nl[p] > st && P.text[st] == '+') break;
            L += P.nl[p] - st;
        }
        if (steps == GS_ML_MAX_LINES) {
            next = GS_ML_TOO_LONG;  // (an error only if a record really starts here: gs_ml_class_kernel)
        } else if (p < n) {
            plus = p;
            uint32_t q = p + 1;  // the first quality line, whatever its length
            if (q < n) {
                u64 acc = gs_ml_len(P, q);
                q++;
                steps = 0;
                while (acc < L && q < n && steps < GS_ML_MAX_LINES) {
                    acc += gs_ml_len(P, q);
                    q++;
                    steps++;
                }
                if (steps == GS_ML_MAX_LINES)
                    next = GS_ML_TOO_LONG;
                else if (acc >= L)
                    next = q;  // (q <= n: the record ends inside the chunk)
            }
        }
    }
    P.ml_next[i] = next;
    P.ml_plus[i] = plus;
}

__global__ __launch_bounds__(256) void gs_ml_orbit_init_kernel(GsTextParams P) {
    const uint32_t n = (uint32_t)P.n_lines;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n || P.status[GS_TS_CHUNK_ERR] != 0) return;
    P.ml_mark[i] = i == 0;
    P.ml_jump_a[i] = P.ml_next[i] < n ? P.ml_next[i] : GS_ML_NONE;  // (next == n: the record ends with the chunk, nothing follows)
}

// one round of pointer doubling: ja = next^(2^r) -> jb = next^(2^(r+1)); every marked line marks where ja leads.  (A mark that
// appears during the round may or may not be passed on in the same round: marks only ever land on starts of records, and after
// round r all starts within 2^(r+1) records of line 0 are marked either way.)
__global__ __launch_bounds__(256) void gs_ml_orbit_round_kernel(GsTextParams P, const uint32_t *ja, uint32_t *jb) {
    const uint32_t n = (uint32_t)P.n_lines;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n || P.status[GS_TS_CHUNK_ERR] != 0) return;
    const uint32_t j = ja[i];
    if (j != GS_ML_NONE && P.ml_mark[i]) P.ml_mark[j] = 1;
    jb[i] = j != GS_ML_NONE ? ja[j] : GS_ML_NONE;
}

__global__ __launch_bounds__(256) void gs_ml_class_kernel(GsTextParams P, uint8_t *line_class) {
    const uint32_t n = (uint32_t)P.n_lines;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n || P.status[GS_TS_CHUNK_ERR] != 0 || !P.ml_mark[i]) return;
    const uint32_t next = P.ml_next[i];
    if (next == GS_ML_TOO_LONG) {  // a record of more than GS_ML_MAX_LINES lines: the host parser's case
        atomicOr(&P.status[GS_TS_CHUNK_ERR], GS_TE_SHAPE);
        atomicMin(&P.status[GS_TS_FIRST_BAD], i);
        return;
    }
    if (next == GS_ML_NONE) {  // the record goes on behind the chunk: the next chunk starts with this line
        atomicMin(&P.ml_out[1], (u64)i);
        return;
    }
    line_class[i] = 1;
    for (uint32_t j = i + 1; j < P.ml_plus[i]; j++) line_class[j] = 2;
    atomicAdd(&P.ml_out[0], 1ULL);
}

__global__ void gs_text_commit_kernel(GsTextParams P, uint32_t ticket) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (P.status[GS_TS_STICKY] == 0 && P.status[GS_TS_CHUNK_ERR] != 0) {
        P.status[GS_TS_STICKY] = P.status[GS_TS_CHUNK_ERR];
        P.status[GS_TS_FAILED_TICKET] = ticket;
    }
    if (P.status[GS_TS_STICKY] != 0) {
        P.status[GS_TS_SKIP] = 1;
    } else {
        P.status[GS_TS_SKIP] = 0;
        P.run_totals[0] += (u64)(P.n_records >= 0 ? P.n_records : (P.n_lines >> 2));
        P.run_totals[1] += P.chunk_totals[1];
        P.run_totals[2] += P.chunk_totals[2];
    }
    P.status[GS_TS_CHUNK_ERR] = 0;
}

// general FASTQ, first half: newline offsets + record structure.  ml_out: [0] = 0, [1] = n_lines before the launch; afterwards
// [0] complete records, [1] the lines they cover (the caller reads them back, sets n_lines / n_records / line_class and calls
// gs_launch_text_scan for the second half with lines_done = 1)
extern "C" hipError_t gs_launch_text_ml(const GsTextParams *P, uint8_t *line_class, hipStream_t stream) {
    const int64_t n_tiles = (P->n_bytes + GS_TEXT_TILE - 1) / GS_TEXT_TILE;
    if (n_tiles <= 0 || P->n_lines <= 0) return hipSuccess;
    hipLaunchKernelGGL(gs_text_count_kernel, dim3((unsigned)n_tiles), dim3(GS_TEXT_BLOCK), 0, stream, *P);
    hipLaunchKernelGGL(gs_text_scan_kernel, dim3(1), dim3(1024), 0, stream, *P, n_tiles);
    hipLaunchKernelGGL(gs_text_lines_kernel, dim3((unsigned)n_tiles), dim3(GS_TEXT_BLOCK), 0, stream, *P);
    const unsigned g = (unsigned)((P->n_lines + 255) / 256);
    hipLaunchKernelGGL(gs_ml_next_kernel, dim3(g), dim3(256), 0, stream, *P);
    hipLaunchKernelGGL(gs_ml_orbit_init_kernel, dim3(g), dim3(256), 0, stream, *P);
    uint32_t *ja = P->ml_jump_a, *jb = P->ml_jump_b;
    for (int64_t span = 1; span < P->n_lines; span <<= 1) {
        hipLaunchKernelGGL(gs_ml_orbit_round_kernel, dim3(g), dim3(256), 0, stream, *P, ja, jb);
        std::swap(ja, jb);
    }
    hipLaunchKernelGGL(gs_ml_class_kernel, dim3(g), dim3(256), 0, stream, *P, line_class);
    return hipGetLastError();
}

extern "C" hipError_t gs_launch_text_scan(const GsTextParams *P, uint32_t ticket, hipStream_t stream) {
    const int64_t n_tiles = (P->n_bytes + GS_TEXT_TILE - 1) / GS_TEXT_TILE;
    if (n_tiles > 0) {
        if (P->line_class == nullptr) {  // (general FASTQ: gs_launch_text_ml has found the newlines already)
            hipLaunchKernelGGL(gs_text_count_kernel, dim3((unsigned)n_tiles), dim3(GS_TEXT_BLOCK), 0, stream, *P);
            hipLaunchKernelGGL(gs_text_scan_kernel, dim3(1), dim3(1024), 0, stream, *P, n_tiles);
            hipLaunchKernelGGL(gs_text_lines_kernel, dim3((unsigned)n_tiles), dim3(GS_TEXT_BLOCK), 0, stream, *P);
        }
        if (P->n_records >= 0) {  // FASTA, general FASTQ
            const int64_t n_blocks = (P->n_lines + GS_FA_BLOCK - 1) / GS_FA_BLOCK;
            if (n_blocks > 0) {
                hipLaunchKernelGGL(gs_fasta_lines_kernel, dim3((unsigned)n_blocks), dim3(GS_FA_BLOCK), 0, stream, *P);
                hipLaunchKernelGGL(gs_fasta_scan_kernel, dim3(1), dim3(1024), 0, stream, *P, n_blocks);
                hipLaunchKernelGGL(gs_fasta_emit_kernel, dim3((unsigned)n_blocks), dim3(GS_FA_BLOCK), 0, stream, *P);
                hipLaunchKernelGGL(gs_fasta_gather_kernel, dim3((unsigned)std::min<int64_t>((P->n_lines + 3) / 4, 8192)), dim3(256), 0, stream, *P);
            }
            if (P->n_records > 0)
                hipLaunchKernelGGL(gs_fasta_totals_kernel, dim3((unsigned)((P->n_records + GS_TEXT_BLOCK - 1) / GS_TEXT_BLOCK)),
                                   dim3(GS_TEXT_BLOCK), 0, stream, *P);
        } else {
            const int64_t n_rec = P->n_lines >> 2;
            if (n_rec > 0)
                hipLaunchKernelGGL(gs_text_records_kernel, dim3((unsigned)((n_rec + GS_TEXT_BLOCK - 1) / GS_TEXT_BLOCK)),
                                   dim3(GS_TEXT_BLOCK), 0, stream, *P);
        }
    }
    hipLaunchKernelGGL(gs_text_commit_kernel, dim3(1), dim3(64), 0, stream, *P, ticket);
    return hipGetLastError();
}
