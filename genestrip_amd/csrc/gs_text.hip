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
// untouched, and the host re-parses from the start of the failing chunk with the general parser.
//
// Kernels (tiles of 4096 bytes, one uint4 per thread):
//   gs_text_count_kernel    newlines per tile, NUL check                      streaming, 1 B read per byte
//   gs_text_scan_kernel     exclusive prefix over the tile counts (one block)
//   gs_text_lines_kernel    byte offset of every newline -> nl[]               streaming
//   gs_text_records_kernel  one thread per record: checks, (start, end) of the sequence line, totals
//   gs_text_commit_kernel   one thread: sticky error / skip flag / run totals
#include <hip/hip_runtime.h>
#include <stdint.h>

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
    const int64_t per = (n_tiles + 1023) / 1024;
    const int64_t a = (int64_t)t * per, b = a + per < n_tiles ? a + per : n_tiles;
    u64 sum = 0;
    for (int64_t i = a; i < b; i++) sum += P.tile_count[i];
    s_part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan
        const u64 x = t >= d ? s_part[t - d] : 0;
        __syncthreads();
        s_part[t] += x;
        __syncthreads();
    }
    u64 run = s_part[t] - sum;
    for (int64_t i = a; i < b; i++) {
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
        P.run_totals[0] += (u64)(P.n_lines >> 2);
        P.run_totals[1] += P.chunk_totals[1];
        P.run_totals[2] += P.chunk_totals[2];
    }
    P.status[GS_TS_CHUNK_ERR] = 0;
}

extern "C" hipError_t gs_launch_text_scan(const GsTextParams *P, uint32_t ticket, hipStream_t stream) {
    const int64_t n_tiles = (P->n_bytes + GS_TEXT_TILE - 1) / GS_TEXT_TILE;
    if (n_tiles > 0) {
        hipLaunchKernelGGL(gs_text_count_kernel, dim3((unsigned)n_tiles), dim3(GS_TEXT_BLOCK), 0, stream, *P);
        hipLaunchKernelGGL(gs_text_scan_kernel, dim3(1), dim3(1024), 0, stream, *P, n_tiles);
        hipLaunchKernelGGL(gs_text_lines_kernel, dim3((unsigned)n_tiles), dim3(GS_TEXT_BLOCK), 0, stream, *P);
        const int64_t n_rec = P->n_lines >> 2;
        if (n_rec > 0)
            hipLaunchKernelGGL(gs_text_records_kernel, dim3((unsigned)((n_rec + GS_TEXT_BLOCK - 1) / GS_TEXT_BLOCK)),
                               dim3(GS_TEXT_BLOCK), 0, stream, *P);
    }
    hipLaunchKernelGGL(gs_text_commit_kernel, dim3(1), dim3(64), 0, stream, *P, ticket);
    return hipGetLastError();
}
