// gs_deflate_dev.hip -- the OUTPUT side of the file pipelines on the device: the accepted (or rejected) records of a text chunk
// gathered in HBM, and DEFLATE written there as block-gzip (BGZF) members.
//
// The reference writes its FASTQ outputs gzip-compressed by default (GSConfigKey.java:155 gzipFastqOutput = true; FilterGoal.java:76,
// MatchResultGoal.java:106 -> StreamProvider.getOutputStreamForFile -> java.util.zip.GZIPOutputStream, one thread), record by record
// through ReadEntry.write (C/fastq/AbstractFastqReader.java:570-584).  With the kernels two orders of magnitude ahead of that
// thread -- and of sixteen zlib threads, which is what round 3 had -- the output side decides what `filter` delivers: 1.4 Gbp/s.
// Here
//   * gc_*: the records a writer wants (accept flag set / clear) are rewritten as ReadEntry.write does -- descriptor, read, "+",
//     the quality line or '~' x length -- back to back in device memory: one length per record, a two-level prefix sum, one wave
//     per record for the bytes.  Only what will be written ever crosses PCIe.
//   * gd_*: that text is cut into pieces of 63 KiB and ONE WAVE turns a piece into one gzip member (RFC 1952) whose extra field
//     states its size (BGZF, SAM spec 4.1: what bgzip writes; zcat / GZIPInputStream read the file as one stream, bgzip-aware
//     readers -- and this library's device inflater -- take the members side by side): 64 positions at a time,
//       - every lane looks for a match of its position: the run (distance 1: the '~' line of a rewritten record, homopolymers)
//         and ONE candidate from a hash of its next four bytes (the same header a record earlier, repeats), extended eight
//         bytes per step; a match is kept only if it pays under the code at hand (bases cost two bits: most of what zlib's
//         level 1 "finds" in DNA costs more than the literals it replaces);
//       - a short scalar walk over the lanes that hold a match picks the tokens greedily (first match wins, the positions it
//         covers are skipped, as zlib's fast levels do);
//       - every token lane looks its Huffman code up in LDS (a length's code and extra bits as one precomputed word), a DPP
//         prefix sum over the bit counts places the tokens, and the bits are ORed into a small LDS ring whose finished words
//         leave as whole dwords.
//     The Huffman code is SEMI-STATIC: a counting pass of the same tokenizer over a sample of the text (up to 2 048 ranges of 1 KB, spread
//     over it) gives the symbol frequencies, the host builds the length-limited code and the dynamic block header ONCE per call,
//     and every member of the call carries that header: FASTQ is stationary, the loss against a code per member is under one
//     per cent, and no tree is ever built on the device.  Every literal has a code; a piece that would grow is stored.
//     CRC-32 of the pieces by gi_crc_kernel (gs_inflate_dev.hip), 64 lanes over equal slices.
//   * the members are compacted (prefix sum of their sizes, byte gather) and leave the device in one copy.
// Output is valid DEFLATE for any input bytes; what it is tuned for is FASTQ.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/gsgpu.h"

typedef unsigned long long u64;

#define GD_PIECE 64512u   // most text bytes per member: 63 KiB = 64 slices of 1008 bytes (the CRC kernel's fast join wants a multiple of 1024)
#define GD_PIECE_MIN 16384u  // a short text is cut finer, so that the device's wave slots have a piece each (a lone wave takes 2.4 ms for 63 KiB)
#define GD_SLOT_EXTRA 1024u  // output room per member beyond its text (a stored piece takes 18 + 5 + 8 bytes more, a piece that is given up on at most 384 + 31)
#define GD_WAVES 4
#define GD_HBITS 11
#define GD_HSIZE (1u << GD_HBITS)
#define GD_RING 128u      // dwords of output under construction per wave
#define GD_MAXLEN 258u
#define GD_PREFIX_WORDS 128
#define GD_SAMPLE 2048         // ranges of the text the counting pass looks at ...
#define GD_SAMPLE_BYTES 1008u  // ... of this many bytes each: one wave per range -- a chain of 16 steps (a whole piece: 1 000 steps, 2.4 ms; ranges of 4 032 bytes 0.15-0.57 ms)

extern "C" int gs_crc_tiles_device(const uint8_t *d_text, int64_t n, uint32_t tile, uint32_t *d_crc, hipStream_t stream);
extern "C" uint32_t gs_crc_init_term(uint64_t n);
extern "C" const char *gs_inflate_last_error(void);

// what a call's members share (device copy of GdCode's tables)
struct GdTables {
    uint32_t lit[288];   // literal / length symbol: bits << 24 | code, bit-reversed (the counting pass: frequencies)
    uint32_t len[256];   // match length 3 .. 258: total bits << 24 | (symbol code | extra bits behind it)
    uint32_t dist[32];   // distance symbol: bits << 24 | code, bit-reversed (counting pass: frequencies)
    uint32_t prefix[GD_PREFIX_WORDS];  // BGZF header (18 bytes, BSIZE blank) + BFINAL / BTYPE + the dynamic header, as a bit string
    uint32_t prefix_bits;
    uint32_t eob;        // bits << 24 | code of symbol 256
    uint32_t pad[2];
};

struct GdWaveLds {
    uint16_t hash[GD_HSIZE];
    uint32_t ring[GD_RING];
};

__device__ __forceinline__ int gd_lane() { return (int)__lane_id(); }
__device__ __forceinline__ uint32_t gd_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ void gd_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t gd_scan_incl(uint32_t x) {  // inclusive prefix sum over the wave (DPP, as gi_scan_incl)
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}
typedef u64 __attribute__((aligned(1))) gd_u64_any;
typedef uint32_t __attribute__((aligned(1))) gd_u32_any;
__device__ __forceinline__ u64 gd_load64(const uint8_t *p) { return *reinterpret_cast<const gd_u64_any *>(p); }
__device__ __forceinline__ uint32_t gd_load32(const uint8_t *p) { return *reinterpret_cast<const gd_u32_any *>(p); }

// bytes that t[c ..) and t[p ..) share, at most maxlen (p + maxlen <= n)
__device__ __forceinline__ uint32_t gd_common(const uint8_t *t, uint32_t p, uint32_t c, uint32_t maxlen) {
    uint32_t l = 0;
    while (l + 8u <= maxlen) {
        const u64 x = gd_load64(t + p + l) ^ gd_load64(t + c + l);
        if (x) return l + ((uint32_t)__builtin_ctzll(x) >> 3);
        l += 8;
    }
    while (l < maxlen && t[p + l] == t[c + l]) l++;
    return l;
}

// does a match of `len` bytes at `dist` pay?  Literals cost about two bits when they are bases and six otherwise (judged by the
// match's first byte), a match a length code of about seven bits, a distance code of about three and log2(dist) - 1 extra bits.
__device__ __forceinline__ bool gd_pays(uint32_t len, uint32_t dist, uint32_t b0) {
    const bool base = b0 == 'A' || b0 == 'C' || b0 == 'G' || b0 == 'T';
    const uint32_t have = len * (base ? 9u : 24u);                                  // quarter bits
    const uint32_t need = 4u * (9u + (dist > 1u ? 31u - (uint32_t)__builtin_clz(dist) : 0u));
    return have > need;
}

// length -> literal / length symbol (RFC 1951 3.2.5)
__host__ __device__ inline uint32_t gd_len_symbol(uint32_t len) {
    if (len == 258u) return 285u;
    const uint32_t l = len - 3u;
    if (l < 8u) return 257u + l;
    uint32_t msb = 31u;
    while (!((l >> msb) & 1u)) msb--;
    const uint32_t extra = msb - 2u;
    return 261u + 4u * extra + ((l >> extra) & 3u);
}

// One piece through the tokenizer.  COUNT: the symbols are counted into T->lit / T->dist (LDS, shared by the workgroup).  Else the
// member is written to `out` (GD_SLOT bytes); returns its size (wave-uniform), 0 if the piece has to be stored.
template <bool COUNT>
__device__ __forceinline__ uint32_t gd_piece(const uint8_t *t, uint32_t n, GdTables *T, GdWaveLds &W, uint32_t *out) {
    const int lane = gd_lane();
    for (uint32_t i = (uint32_t)lane; i < GD_HSIZE / 2u; i += 64u) reinterpret_cast<uint32_t *>(W.hash)[i] = 0xffffffffu;
    uint32_t bitpos = 0;
    const uint32_t limit = (18u + 5u + n) * 8u;
    if (!COUNT) {
        for (uint32_t i = (uint32_t)lane; i < GD_RING; i += 64u) W.ring[i] = 0;
        gd_lds_sync();
        const uint32_t pb = T->prefix_bits, pw = pb >> 5;
        for (uint32_t i = (uint32_t)lane; i < pw; i += 64u)
            if (i != 4u) out[i] = T->prefix[i];  // (dword 4 holds BSIZE: written last)
        if (lane == 0 && (pb & 31u)) W.ring[pw & (GD_RING - 1u)] = T->prefix[pw];
        bitpos = pb;
    }
    gd_lds_sync();
    uint32_t skip = 0;
    bool failed = false;
    for (uint32_t g0 = 0; g0 < n; g0 += 64u) {
        if (skip >= 64u) {
            skip -= 64u;
            continue;
        }
        const uint32_t p = g0 + (uint32_t)lane;
        const bool in = p < n;
        const uint32_t b0 = in ? (uint32_t)t[p] : 0u;
        uint32_t mlen = 0, mdist = 0;
        // the hash of the next four bytes: candidate read, own position written (every lane of the piece, covered or not)
        uint32_t cand = 0xffffu;
        const bool hashed = in && p + 4u <= n;
        uint32_t h = 0;
        if (hashed) {
            h = (gd_load32(t + p) * 2654435761u) >> (32 - GD_HBITS);
            cand = W.hash[h];
        }
        gd_lds_sync();
        if (hashed) W.hash[h] = (uint16_t)p;
        const bool search = in && (uint32_t)lane >= skip && p + 4u <= n;
        if (search) {
            const uint32_t maxlen = n - p < GD_MAXLEN ? n - p : GD_MAXLEN;
            if (p >= 1u) {
                const uint32_t l1 = gd_common(t, p, p - 1u, maxlen);
                if (l1 >= 4u && gd_pays(l1, 1u, b0)) {
                    mlen = l1;
                    mdist = 1u;
                }
            }
            if (cand != 0xffffu && cand < p && p - cand <= 32768u && p - cand > 1u && mlen < maxlen) {
                const uint32_t l2 = gd_common(t, p, cand, maxlen);
                if (l2 >= 4u && l2 > mlen && gd_pays(l2, p - cand, b0)) {
                    mlen = l2;
                    mdist = p - cand;
                }
            }
        }
        // greedy choice along the positions: the first match wins and covers what follows it
        const u64 M = __ballot(mlen > 0u);
        const u64 valid = __ballot(in);
        u64 lit_mask = 0, match_mask = 0;
        uint32_t at = skip;
        skip = 0;
        for (;;) {
            const u64 rest = M & (~0ull << at);
            if (!rest) {
                lit_mask |= ~0ull << at;
                break;
            }
            const uint32_t m = (uint32_t)__builtin_ctzll(rest);
            lit_mask |= (~0ull << at) & ~(~0ull << m);
            match_mask |= 1ull << m;
            const uint32_t L = (uint32_t)__builtin_amdgcn_readlane((int)mlen, (int)m);
            at = m + L;
            if (at >= 64u) {
                skip = at - 64u;
                break;
            }
        }
        lit_mask &= valid;
        const bool is_lit = (lit_mask >> lane) & 1ull, is_match = (match_mask >> lane) & 1ull;
        uint32_t dsym = 0, dextra_bits = 0, dextra = 0;
        if (is_match) {
            const uint32_t d = mdist - 1u;
            if (d < 4u) {
                dsym = d;
            } else {
                const uint32_t msb = 31u - (uint32_t)__builtin_clz(d);
                dsym = 2u * msb + ((d >> (msb - 1u)) & 1u);
                dextra_bits = msb - 1u;
                dextra = d & ((1u << dextra_bits) - 1u);
            }
        }
        if (COUNT) {
            if (is_lit) atomicAdd(&T->lit[b0], 1u);
            if (is_match) {
                atomicAdd(&T->lit[gd_len_symbol(mlen)], 1u);
                atomicAdd(&T->dist[dsym], 1u);
            }
            continue;
        }
        u64 bits = 0;
        uint32_t nb = 0;
        if (is_lit) {
            const uint32_t e = T->lit[b0];
            nb = e >> 24;
            bits = e & 0xffffffu;
            if (nb == 0u) failed = true;
        } else if (is_match) {
            const uint32_t e = T->len[mlen - 3u], de = T->dist[dsym];
            nb = e >> 24;
            bits = e & 0xffffffu;
            if (nb == 0u || (de >> 24) == 0u) failed = true;
            bits |= (u64)(de & 0xffffffu) << nb;
            nb += de >> 24;
            bits |= (u64)dextra << nb;
            nb += dextra_bits;
        }
        const uint32_t incl = gd_scan_incl(nb);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (nb) {
            const uint32_t off = bitpos + incl - nb, w = off >> 5, sh = off & 31u;
            const u64 a = bits << sh;
            const uint32_t v0 = (uint32_t)a, v1 = (uint32_t)(a >> 32), v2 = sh ? (uint32_t)(bits >> (64u - sh)) : 0u;
            atomicOr(&W.ring[w & (GD_RING - 1u)], v0);
            if (v1) atomicOr(&W.ring[(w + 1u) & (GD_RING - 1u)], v1);
            if (v2) atomicOr(&W.ring[(w + 2u) & (GD_RING - 1u)], v2);
        }
        gd_lds_sync();
        const uint32_t first = bitpos >> 5, last = (bitpos + total) >> 5;
        for (uint32_t i = first + (uint32_t)lane; i < last; i += 64u) {
            out[i] = W.ring[i & (GD_RING - 1u)];
            W.ring[i & (GD_RING - 1u)] = 0;
        }
        gd_lds_sync();
        bitpos += total;
        if (__ballot(failed) != 0ull || bitpos > limit) return 0u;  // a byte without a code, or no gain: stored
    }
    if (COUNT) {
        if (lane == 0) atomicAdd(&T->lit[256], 1u);
        return 0u;
    }
    // end of block, padding to a byte, the trailer's place
    {
        const uint32_t e = T->eob;
        u64 bits = e & 0xffffffu;
        uint32_t nb = e >> 24;
        if (nb == 0u) return 0u;
        nb = (bitpos + nb + 7u & ~7u) - bitpos;  // (the padding bits are zeros)
        if (lane == 0) {
            const uint32_t off = bitpos, w = off >> 5, sh = off & 31u;
            const u64 a = bits << sh;
            atomicOr(&W.ring[w & (GD_RING - 1u)], (uint32_t)a);
            if ((uint32_t)(a >> 32)) atomicOr(&W.ring[(w + 1u) & (GD_RING - 1u)], (uint32_t)(a >> 32));
        }
        gd_lds_sync();
        const uint32_t first = bitpos >> 5, last = (bitpos + nb) >> 5;
        for (uint32_t i = first + (uint32_t)lane; i < last; i += 64u) {
            out[i] = W.ring[i & (GD_RING - 1u)];
            W.ring[i & (GD_RING - 1u)] = 0;
        }
        gd_lds_sync();
        bitpos += nb;
    }
    return bitpos >> 3;  // (bytes so far, the words below bitpos >> 5 have left the ring; the caller appends CRC-32 and ISIZE: gd_finish)
}

// CRC-32 + ISIZE behind `bytes` bytes of which the words below bitpos >> 5 have left the ring; BSIZE; returns the member's size
__device__ __forceinline__ uint32_t gd_finish(GdTables *T, GdWaveLds &W, uint32_t *out, uint32_t bytes, uint32_t flushed_words, uint32_t crc, uint32_t isize) {
    const int lane = gd_lane();
    const uint32_t bitpos = bytes * 8u;
    if (lane == 0) {
        const u64 bits = (u64)crc | ((u64)isize << 32);
        const uint32_t w = bitpos >> 5, sh = bitpos & 31u;
        const u64 a = bits << sh;
        atomicOr(&W.ring[w & (GD_RING - 1u)], (uint32_t)a);
        if ((uint32_t)(a >> 32)) atomicOr(&W.ring[(w + 1u) & (GD_RING - 1u)], (uint32_t)(a >> 32));
        if (sh) atomicOr(&W.ring[(w + 2u) & (GD_RING - 1u)], (uint32_t)(bits >> (64u - sh)));
    }
    gd_lds_sync();
    const uint32_t total = bytes + 8u, words = (total + 3u) >> 2;
    for (uint32_t i = flushed_words + (uint32_t)lane; i < words; i += 64u) out[i] = W.ring[i & (GD_RING - 1u)];
    if (lane == 0) out[4] = (T->prefix[4] & 0xffff0000u) | (total - 1u);
    return total;
}

__global__ __launch_bounds__(64 * GD_WAVES) void gd_count_kernel(const uint8_t *text, int64_t n, int64_t n_pieces, int n_sample, uint32_t *hist) {
    __shared__ GdTables T;
    __shared__ GdWaveLds W[GD_WAVES];
    for (uint32_t i = threadIdx.x; i < 288u; i += blockDim.x) T.lit[i] = 0;
    if (threadIdx.x < 32u) T.dist[threadIdx.x] = 0;
    __syncthreads();
    const int wv = (int)(threadIdx.x >> 6);
    const int64_t s = (int64_t)blockIdx.x * GD_WAVES + wv;
    if (s < n_sample) {  // range s of n_sample, spread evenly over the n_pieces ranges the text has
        const int64_t r = s * n_pieces / n_sample;
        const int64_t base = r * (int64_t)GD_SAMPLE_BYTES;
        const uint32_t len = (uint32_t)(n - base < (int64_t)GD_SAMPLE_BYTES ? n - base : (int64_t)GD_SAMPLE_BYTES);
        gd_piece<true>(text + base, len, &T, W[wv], nullptr);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 288u; i += blockDim.x)
        if (T.lit[i]) atomicAdd(&hist[i], T.lit[i]);
    if (threadIdx.x < 32u && T.dist[threadIdx.x]) atomicAdd(&hist[288u + threadIdx.x], T.dist[threadIdx.x]);
}

// one wave per piece (pieces drawn in order: blockIdx * GD_WAVES + wave, striding over the grid)
__global__ __launch_bounds__(64 * GD_WAVES) __attribute__((amdgpu_waves_per_eu(7, 7))) void gd_deflate_kernel(const uint8_t *text, int64_t n, int64_t n_pieces, uint32_t piece_bytes, const GdTables *tables,
                                                                   const uint32_t *crc_raw, uint32_t init_full, uint32_t init_last, uint8_t *slots, uint32_t *sizes) {
    __shared__ GdTables T;
    __shared__ GdWaveLds W[GD_WAVES];
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(tables);
        uint32_t *dst = reinterpret_cast<uint32_t *>(&T);
        for (uint32_t i = threadIdx.x; i < sizeof(GdTables) / 4u; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const int wv = (int)(threadIdx.x >> 6), lane = gd_lane();
    for (int64_t piece = (int64_t)blockIdx.x * GD_WAVES + wv; piece < n_pieces; piece += (int64_t)gridDim.x * GD_WAVES) {
        const int64_t base = piece * (int64_t)piece_bytes;
        const uint32_t len = (uint32_t)(n - base < (int64_t)piece_bytes ? n - base : (int64_t)piece_bytes);
        const uint8_t *t = text + base;
        uint32_t *out = reinterpret_cast<uint32_t *>(slots + (size_t)piece * (piece_bytes + GD_SLOT_EXTRA));
        const uint32_t crc = ~(crc_raw[piece] ^ (len == piece_bytes ? init_full : init_last));
        uint32_t bytes = gd_piece<false>(t, len, &T, W[wv], out);
        uint32_t total;
        if (bytes) {
            total = gd_finish(&T, W[wv], out, bytes, (bytes * 8u) >> 5, crc, len);
        } else {
            // stored: the BGZF header, one stored block (BFINAL, LEN, ~LEN), the bytes, the trailer
            uint8_t *o = reinterpret_cast<uint8_t *>(out);
            total = 18u + 5u + len + 8u;
            {   // (every byte from exactly one lane)
                uint32_t b = 0;
                if (lane < 16)
                    b = T.prefix[lane >> 2] >> (8 * (lane & 3));
                else if (lane < 18)
                    b = (total - 1u) >> (8 * (lane - 16));
                else if (lane == 18)
                    b = 1;
                else if (lane < 21)
                    b = len >> (8 * (lane - 19));
                else if (lane < 23)
                    b = ~len >> (8 * (lane - 21));
                if (lane < 23) o[lane] = (uint8_t)b;
                if (lane >= 32 && lane < 40) o[23u + len + (uint32_t)(lane - 32)] = (uint8_t)((lane < 36 ? crc : len) >> (8 * ((lane - 32) & 3)));
            }
            for (uint32_t i = (uint32_t)lane; i < len; i += 64u) o[23u + i] = t[i];
        }
        if (lane == 0) sizes[piece] = total;
        gd_lds_sync();
    }
}

// exclusive prefix of the members' sizes (one block); off[n] = the total
__global__ __launch_bounds__(1024) void gd_offsets_kernel(const uint32_t *sizes, int64_t n, u64 *off) {
    __shared__ u64 s_part[1024];
    const int t = (int)threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t a = (int64_t)t * per, b = a + per < n ? a + per : n;
    u64 sum = 0;
    for (int64_t i = a; i < b; i++) sum += sizes[i];
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
        off[i] = run;
        run += sizes[i];
    }
    if (t == 1023) off[n] = s_part[1023];
}

__global__ __launch_bounds__(256) void gd_gather_kernel(const uint8_t *slots, uint32_t slot_bytes, const uint32_t *sizes, const u64 *off, int64_t n, uint8_t *out) {
    const int64_t piece = blockIdx.x;
    if (piece >= n) return;
    const uint8_t *src = slots + (size_t)piece * slot_bytes;
    uint8_t *dst = out + off[piece];
    const uint32_t sz = sizes[piece];
    // whole dwords where the destination is aligned, bytes at the ends
    const uint32_t head = (uint32_t)((4u - ((uintptr_t)dst & 3u)) & 3u);
    const uint32_t h = head < sz ? head : sz;
    if (threadIdx.x < h) dst[threadIdx.x] = src[threadIdx.x];
    const uint32_t words = (sz - h) >> 2;
    for (uint32_t i = threadIdx.x; i < words; i += 256u) reinterpret_cast<uint32_t *>(dst + h)[i] = gd_load32(src + h + 4u * i);
    const uint32_t done = h + 4u * words;
    if (threadIdx.x < sz - done) dst[done + threadIdx.x] = src[done + threadIdx.x];
}

// ---------------------------------------------------------------------------------------------------
// the records a writer wants, rewritten as ReadEntry.write does (C/fastq/AbstractFastqReader.java:570-584)
// ---------------------------------------------------------------------------------------------------
#define GC_BLOCK 256
struct GcRec {
    uint32_t d0, head, q0, qn;  // descriptor start; bytes of "descriptor \n read \n" (contiguous in the chunk); quality line
};
__device__ __forceinline__ GcRec gc_record(const uint32_t *nl, int64_t r) {
    GcRec g;
    g.d0 = r ? nl[4 * r - 1] + 1u : 0u;
    g.head = nl[4 * r + 1] + 1u - g.d0;
    g.q0 = nl[4 * r + 2] + 1u;
    g.qn = nl[4 * r + 3] - g.q0;
    return g;
}
__device__ __forceinline__ uint32_t gc_out_len(const uint32_t *nl, int64_t r, bool probs) {
    const GcRec g = gc_record(nl, r);
    const uint32_t sl = nl[4 * r + 1] - nl[4 * r] - 1u;
    return g.head + 2u + (probs ? g.qn : sl) + 1u;
}

__global__ __launch_bounds__(GC_BLOCK) void gc_len_kernel(const uint32_t *nl, int64_t n_rec, const uint8_t *flags, uint32_t mask, uint32_t want, int probs, uint32_t *len,
                                                          u64 *block_sum, u64 *block_cnt) {
    __shared__ u64 s_sum, s_cnt;
    if (threadIdx.x == 0) s_sum = s_cnt = 0;
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * GC_BLOCK + threadIdx.x;
    uint32_t l = 0;
    if (r < n_rec && (uint32_t)((flags[r] & mask) != 0u) == want) l = gc_out_len(nl, r, probs != 0);
    if (r < n_rec) len[r] = l;
    if (l) {
        atomicAdd(&s_sum, (u64)l);
        atomicAdd(&s_cnt, (u64)1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        block_sum[blockIdx.x] = s_sum;
        block_cnt[blockIdx.x] = s_cnt;
    }
}

// one block: block_sum -> exclusive prefix in place; totals[0] = bytes, totals[1] = records
__global__ __launch_bounds__(1024) void gc_scan_kernel(u64 *block_sum, const u64 *block_cnt, int64_t n_blocks, u64 *totals) {
    __shared__ u64 s_part[1024], s_c[1024];
    const int t = (int)threadIdx.x;
    const int64_t per = (n_blocks + 1023) / 1024;
    const int64_t a = (int64_t)t * per, b = a + per < n_blocks ? a + per : n_blocks;
    u64 sum = 0, cnt = 0;
    for (int64_t i = a; i < b; i++) {
        sum += block_sum[i];
        cnt += block_cnt[i];
    }
    s_part[t] = sum;
    s_c[t] = cnt;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const u64 x = t >= d ? s_part[t - d] : 0, y = t >= d ? s_c[t - d] : 0;
        __syncthreads();
        s_part[t] += x;
        s_c[t] += y;
        __syncthreads();
    }
    u64 run = s_part[t] - sum;
    for (int64_t i = a; i < b; i++) {
        const u64 c = block_sum[i];
        block_sum[i] = run;
        run += c;
    }
    if (t == 1023) {
        totals[0] = s_part[1023];
        totals[1] = s_c[1023];
    }
}

__global__ __launch_bounds__(GC_BLOCK) void gc_copy_kernel(const uint8_t *text, const uint32_t *nl, int64_t n_rec, const uint32_t *len, const u64 *block_off, int probs,
                                                           uint8_t *out) {
    __shared__ uint32_t s_wave[GC_BLOCK / 64];
    const int lane = gd_lane(), wv = (int)(threadIdx.x >> 6);
    const int64_t r = (int64_t)blockIdx.x * GC_BLOCK + threadIdx.x;
    const uint32_t l = r < n_rec ? len[r] : 0u;
    const uint32_t incl = gd_scan_incl(l);
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    u64 before = block_off[blockIdx.x];
    for (int i = 0; i < wv; i++) before += s_wave[i];
    const u64 my_off = before + incl - l;
    GcRec g{0, 0, 0, 0};
    if (l) g = gc_record(nl, r);
    u64 todo = __ballot(l != 0u);
    while (todo) {  // one record after the other, all lanes on its bytes
        const int j = (int)__builtin_ctzll(todo);
        todo &= todo - 1;
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)g.d0, j), head = (uint32_t)__builtin_amdgcn_readlane((int)g.head, j);
        const uint32_t q0 = (uint32_t)__builtin_amdgcn_readlane((int)g.q0, j), ol = (uint32_t)__builtin_amdgcn_readlane((int)l, j);
        const uint32_t olo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_off, j), ohi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_off >> 32), j);
        uint8_t *o = out + (((u64)ohi << 32) | olo);
        for (uint32_t i = (uint32_t)lane; i < ol; i += 64u) {
            uint8_t b;
            if (i < head)
                b = text[d0 + i];
            else if (i == head)
                b = '+';
            else if (i == head + 1u || i == ol - 1u)
                b = '\n';
            else
                b = probs ? text[q0 + (i - head - 2u)] : (uint8_t)'~';
            o[i] = b;
        }
    }
}

// the descriptor lines of a few records of a four-line chunk, each into `stride` bytes (NUL-terminated, cut to fit): for a host
// that wants the name of a read it has never seen (CountsPerTaxid.maxContigDescriptor, FastqKMerMatcher.java:401-407)
__global__ __launch_bounds__(64) void gc_desc_kernel(const uint8_t *text, const uint32_t *nl, const int64_t *records, int n, uint8_t *out, int stride) {
    const int i = (int)blockIdx.x;
    if (i >= n) return;
    const int64_t r = records[i];
    const uint32_t d0 = r ? nl[4 * r - 1] + 1u : 0u, len = nl[4 * r] - d0;
    const uint32_t m = len < (uint32_t)(stride - 1) ? len : (uint32_t)(stride - 1);
    uint8_t *o = out + (size_t)i * (size_t)stride;
    for (uint32_t j = threadIdx.x; j < m; j += 64u) o[j] = text[d0 + j];
    if (threadIdx.x == 0) o[m] = 0;
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
static thread_local std::string gd_err;
static int gd_fail(int code, const std::string &m) {
    gd_err = m;
    return code;
}
extern "C" const char *gs_deflate_last_error(void) { return gd_err.c_str(); }
#define GD_TRY(x)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) return gd_fail(e_ == hipErrorOutOfMemory ? GS_E_NOMEM : GS_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Code lengths of a Huffman code over freq[0 .. n) with no length above max_bits and a COMPLETE code (Kraft sum exactly one; zlib's
// inflate refuses an incomplete literal / length code).  Symbols with frequency 0 get length 0; a single used symbol gets a partner.
static void gd_code_lengths(const uint64_t *freq_in, int n, int max_bits, uint8_t *lens) {
    std::vector<uint64_t> freq(freq_in, freq_in + n);
    std::vector<int> used;
    for (int i = 0; i < n; i++)
        if (freq[(size_t)i]) used.push_back(i);
    for (int i = 0; i < n; i++) lens[i] = 0;
    if (used.empty()) return;
    if (used.size() == 1) {  // (a second code so that the code is complete)
        const int other = used[0] == 0 ? 1 : 0;
        freq[(size_t)other] = 1;
        used.push_back(other);
        std::sort(used.begin(), used.end());
    }
    std::stable_sort(used.begin(), used.end(), [&](int a, int b) { return freq[(size_t)a] < freq[(size_t)b]; });
    const int m = (int)used.size();
    // two queues: the leaves in ascending order, the inner nodes in the order they are made (ascending as well)
    std::vector<uint64_t> w((size_t)(2 * m));
    std::vector<int> parent((size_t)(2 * m), -1);
    for (int i = 0; i < m; i++) w[(size_t)i] = freq[(size_t)used[(size_t)i]];
    int leaf = 0, inner = m, made = m;
    auto take = [&]() {
        if (leaf < m && (inner >= made || w[(size_t)leaf] <= w[(size_t)inner])) return leaf++;
        return inner++;
    };
    while (made < 2 * m - 1) {
        const int a = take(), b = take();
        w[(size_t)made] = w[(size_t)a] + w[(size_t)b];
        parent[(size_t)a] = parent[(size_t)b] = made;
        made++;
    }
    std::vector<int> depth((size_t)(2 * m - 1), 0);
    for (int i = 2 * m - 3; i >= 0; i--) depth[(size_t)i] = depth[(size_t)parent[(size_t)i]] + 1;
    std::vector<int> l((size_t)m);
    for (int i = 0; i < m; i++) l[(size_t)i] = std::min(depth[(size_t)i], max_bits);
    // Kraft sum in units of 2^-max_bits; too large: lengthen the rarest symbols that are not at the limit; too small: shorten
    uint64_t K = 0;
    const uint64_t one = (uint64_t)1 << max_bits;
    for (int i = 0; i < m; i++) K += one >> l[(size_t)i];
    while (K > one) {
        // the longest code below the limit (ties: the rarer symbol, i.e. the lower index in `used`)
        int best = -1;
        for (int i = 0; i < m; i++)
            if (l[(size_t)i] < max_bits && (best < 0 || l[(size_t)i] > l[(size_t)best])) best = i;
        K -= one >> (l[(size_t)best] + 1);
        l[(size_t)best]++;
    }
    while (K < one) {
        // shorten the most frequent symbol whose step still fits (the deficit is a multiple of the longest code's weight)
        int best = -1;
        for (int i = m - 1; i >= 0; i--)
            if (l[(size_t)i] > 1 && K + (one >> l[(size_t)i]) <= one) {
                best = i;
                break;
            }
        if (best < 0) break;
        K += one >> l[(size_t)best];
        l[(size_t)best]--;
    }
    for (int i = 0; i < m; i++) lens[used[(size_t)i]] = (uint8_t)l[(size_t)i];
}

// canonical codes (RFC 1951 3.2.2), bit-reversed: Huffman codes are packed starting with their most significant bit
static void gd_canonical(const uint8_t *lens, int n, uint32_t *codes) {
    uint32_t count[16] = {0}, next[16] = {0};
    for (int i = 0; i < n; i++) count[lens[i]]++;
    count[0] = 0;
    uint32_t code = 0;
    for (int b = 1; b < 16; b++) {
        code = (code + count[b - 1]) << 1;
        next[b] = code;
    }
    for (int i = 0; i < n; i++) {
        const int l = lens[i];
        if (!l) {
            codes[i] = 0;
            continue;
        }
        uint32_t c = next[l]++, r = 0;
        for (int b = 0; b < l; b++) r |= ((c >> b) & 1u) << (l - 1 - b);
        codes[i] = r;
    }
}

struct GdBitWriter {
    std::vector<uint32_t> w;
    uint32_t bits = 0;
    void put(uint32_t v, int n) {
        for (int i = 0; i < n; i++) {
            if ((bits >> 5) >= w.size()) w.push_back(0);
            if ((v >> i) & 1u) w[bits >> 5] |= 1u << (bits & 31u);
            bits++;
        }
    }
};

// tables + prefix from the symbol frequencies (hist[0..288): literal / length, hist[288..320): distance); false: the header does not fit
static bool gd_build_tables(const uint32_t *hist, GdTables *T, uint8_t *lit_lens_out = nullptr, uint8_t *dist_lens_out = nullptr) {
    memset(T, 0, sizeof(*T));
    uint64_t lf[286], df[30];
    // every symbol keeps a code (any byte may turn up in a piece that was not sampled): counts are scaled, the floor is one
    for (int i = 0; i < 286; i++) lf[i] = (uint64_t)hist[i] * 64u + 1u;
    for (int i = 0; i < 30; i++) df[i] = (uint64_t)hist[288 + i] * 64u + 1u;
    uint8_t ll[286], dl[30];
    gd_code_lengths(lf, 286, 15, ll);
    gd_code_lengths(df, 30, 15, dl);
    uint32_t lc[286], dc[30];
    gd_canonical(ll, 286, lc);
    gd_canonical(dl, 30, dc);
    for (int i = 0; i < 286; i++) T->lit[i] = ((uint32_t)ll[i] << 24) | lc[i];
    T->eob = T->lit[256];
    for (int i = 0; i < 30; i++) T->dist[i] = ((uint32_t)dl[i] << 24) | dc[i];
    for (uint32_t len = 3; len <= 258; len++) {
        const uint32_t sym = gd_len_symbol(len);
        uint32_t extra = 0, base = 0;
        const uint32_t s = sym - 257u;
        if (s < 8) {
            base = 3 + s;
        } else if (s == 28) {
            base = 258;
        } else {
            extra = (s >> 2) - 1;
            base = 3 + ((4 + (s & 3)) << extra);
        }
        const uint32_t nb = ll[sym], code = lc[sym] | ((len - base) << nb);
        T->len[len - 3] = ((nb + extra) << 24) | code;
    }
    // the dynamic header: code lengths run-length coded (16: repeat previous 3-6, 17: zeros 3-10, 18: zeros 11-138)
    std::vector<uint8_t> all(ll, ll + 286);
    all.insert(all.end(), dl, dl + 30);
    std::vector<std::pair<int, int>> rle;  // (symbol, extra value)
    for (size_t i = 0; i < all.size();) {
        size_t j = i;
        while (j < all.size() && all[j] == all[i]) j++;
        size_t run = j - i;
        if (all[i] == 0) {
            while (run >= 11) {
                const size_t r = std::min<size_t>(run, 138);
                rle.push_back({18, (int)r - 11});
                run -= r;
            }
            if (run >= 3) {
                rle.push_back({17, (int)run - 3});
                run = 0;
            }
            while (run--) rle.push_back({0, 0});
        } else {
            rle.push_back({all[i], 0});
            run--;
            while (run >= 3) {
                const size_t r = std::min<size_t>(run, 6);
                rle.push_back({16, (int)r - 3});
                run -= r;
            }
            while (run--) rle.push_back({all[i], 0});
        }
        i = j;
    }
    uint64_t cf[19] = {0};
    for (auto &x : rle) cf[x.first]++;
    uint8_t cl[19];
    gd_code_lengths(cf, 19, 7, cl);
    uint32_t cc[19];
    gd_canonical(cl, 19, cc);
    static const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19;
    while (hclen > 4 && cl[order[hclen - 1]] == 0) hclen--;
    GdBitWriter bw;
    static const uint8_t head[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0, 0};
    for (int i = 0; i < 18; i++) bw.put(head[i], 8);
    bw.put(1, 1);  // BFINAL
    bw.put(2, 2);  // BTYPE = dynamic
    bw.put(286 - 257, 5);
    bw.put(30 - 1, 5);
    bw.put((uint32_t)(hclen - 4), 4);
    for (int i = 0; i < hclen; i++) bw.put(cl[order[i]], 3);
    for (auto &x : rle) {
        bw.put(cc[x.first], cl[x.first]);
        if (x.first == 16) bw.put((uint32_t)x.second, 2);
        if (x.first == 17) bw.put((uint32_t)x.second, 3);
        if (x.first == 18) bw.put((uint32_t)x.second, 7);
    }
    if (bw.w.size() + 1 > GD_PREFIX_WORDS) return false;
    for (size_t i = 0; i < bw.w.size(); i++) T->prefix[i] = bw.w[i];
    T->prefix_bits = bw.bits;
    if (lit_lens_out) memcpy(lit_lens_out, ll, 286);
    if (dist_lens_out) memcpy(dist_lens_out, dl, 30);
    return true;
}

// text bytes per member for a call over n bytes: about one piece per wave slot of the device, between 16 and 63 KiB (a multiple of 1024)
static uint32_t gd_piece_bytes(int64_t n) {
    int64_t p = ((n + 4095) / 4096 + 1023) / 1024 * 1024;
    if (const char *e = getenv("GS_DEFLATE_PIECE")) p = atoll(e) / 1024 * 1024;  // (tests)
    return (uint32_t)std::max<int64_t>(GD_PIECE_MIN, std::min<int64_t>(GD_PIECE, p));
}

// CPU check of the table builder (no device): `text` as BGZF members through the SAME tables and header, tokens chosen by a plain
// greedy loop (runs and a four-byte hash, the kernel's acceptance rule).  zlib must inflate the result to `text`.
extern "C" int gs_deflate_host_reference(const uint8_t *text, int64_t n, uint8_t *out, int64_t out_cap, int64_t *n_out) try {
    if (n < 0 || (n > 0 && !text) || !out || !n_out) return gd_fail(GS_E_INVALID, "bad argument");
    *n_out = 0;
    struct Tok {
        uint32_t len, dist;
        uint8_t lit;
    };
    auto pays = [](uint32_t len, uint32_t dist, uint32_t b0) {
        const bool base = b0 == 'A' || b0 == 'C' || b0 == 'G' || b0 == 'T';
        uint32_t lg = 0;
        while (dist > 1 && (dist >> (lg + 1))) lg++;
        return len * (base ? 9u : 24u) > 4u * (9u + (dist > 1 ? lg : 0u));
    };
    auto tokenize = [&](const uint8_t *t, uint32_t len, std::vector<Tok> &toks) {
        std::vector<uint16_t> hash(GD_HSIZE, 0xffff);
        toks.clear();
        for (uint32_t p = 0; p < len;) {
            uint32_t ml = 0, md = 0;
            const uint32_t maxlen = std::min(len - p, GD_MAXLEN);
            if (p + 4 <= len) {
                if (p >= 1) {
                    uint32_t l = 0;
                    while (l < maxlen && t[p + l] == t[p - 1 + l]) l++;
                    if (l >= 4 && pays(l, 1, t[p])) ml = l, md = 1;
                }
                uint32_t w;
                memcpy(&w, t + p, 4);
                const uint32_t h = (w * 2654435761u) >> (32 - GD_HBITS);
                const uint32_t c = hash[h];
                hash[h] = (uint16_t)p;
                if (c != 0xffff && c < p && p - c <= 32768 && p - c > 1) {
                    uint32_t l = 0;
                    while (l < maxlen && t[p + l] == t[c + l]) l++;
                    if (l >= 4 && l > ml && pays(l, p - c, t[p])) ml = l, md = p - c;
                }
            }
            if (ml) {
                toks.push_back({ml, md, 0});
                p += ml;
            } else {
                toks.push_back({0, 0, t[p]});
                p++;
            }
        }
    };
    const uint32_t piece_bytes = gd_piece_bytes(n);
    const int64_t n_pieces = (n + piece_bytes - 1) / piece_bytes;
    std::vector<uint32_t> hist(320, 0);
    std::vector<Tok> toks;
    const int64_t n_ranges = (n + GD_SAMPLE_BYTES - 1) / GD_SAMPLE_BYTES;
    const int n_sample = (int)std::min<int64_t>(n_ranges, GD_SAMPLE);
    for (int s = 0; s < n_sample; s++) {
        const int64_t r = (int64_t)s * n_ranges / n_sample, base = r * GD_SAMPLE_BYTES;
        tokenize(text + base, (uint32_t)std::min<int64_t>(GD_SAMPLE_BYTES, n - base), toks);
        for (const Tok &k : toks) {
            if (k.len) {
                hist[gd_len_symbol(k.len)]++;
                const uint32_t d = k.dist - 1;
                uint32_t ds = d;
                if (d >= 4) {
                    uint32_t msb = 31;
                    while (!((d >> msb) & 1u)) msb--;
                    ds = 2 * msb + ((d >> (msb - 1)) & 1u);
                }
                hist[288 + ds]++;
            } else
                hist[k.lit]++;
        }
        hist[256]++;
    }
    GdTables T;
    if (!gd_build_tables(hist.data(), &T)) return gd_fail(GS_E_INVALID, "the dynamic header does not fit");
    int64_t at = 0;
    for (int64_t piece = 0; piece < n_pieces; piece++) {
        const int64_t base = piece * piece_bytes;
        const uint32_t len = (uint32_t)std::min<int64_t>(piece_bytes, n - base);
        tokenize(text + base, len, toks);
        GdBitWriter bw;
        bw.w.assign(T.prefix, T.prefix + ((T.prefix_bits + 31) >> 5));
        bw.bits = T.prefix_bits;
        for (const Tok &k : toks) {
            if (k.len) {
                const uint32_t e = T.len[k.len - 3];
                bw.put(e & 0xffffff, (int)(e >> 24));
                const uint32_t d = k.dist - 1;
                uint32_t ds = d, eb = 0, ev = 0;
                if (d >= 4) {
                    uint32_t msb = 31;
                    while (!((d >> msb) & 1u)) msb--;
                    ds = 2 * msb + ((d >> (msb - 1)) & 1u);
                    eb = msb - 1;
                    ev = d & ((1u << eb) - 1);
                }
                bw.put(T.dist[ds] & 0xffffff, (int)(T.dist[ds] >> 24));
                bw.put(ev, (int)eb);
            } else
                bw.put(T.lit[k.lit] & 0xffffff, (int)(T.lit[k.lit] >> 24));
        }
        bw.put(T.eob & 0xffffff, (int)(T.eob >> 24));
        while (bw.bits & 7u) bw.put(0, 1);
        uint32_t crc = 0xffffffffu;
        for (uint32_t i = 0; i < len; i++) {
            crc ^= text[base + i];
            for (int b = 0; b < 8; b++) crc = (crc >> 1) ^ ((crc & 1u) ? 0xedb88320u : 0u);
        }
        crc = ~crc;
        if ((bw.bits >> 3) > 18u + 5u + len) {  // no gain: one stored block, as the kernel writes it
            bw.w.assign(T.prefix, T.prefix + 5);
            bw.w[4] &= 0xffffu;  // (BSIZE's place; behind it the dynamic header started)
            bw.bits = 144;
            bw.put(1, 8);
            bw.put(len & 0xffffu, 16);
            bw.put(~len & 0xffffu, 16);
            for (uint32_t i = 0; i < len; i++) bw.put(text[base + i], 8);
        }
        bw.put(crc, 32);
        bw.put(len, 32);
        const uint32_t total = bw.bits >> 3;
        if (at + total > out_cap) return gd_fail(GS_E_NOMEM, "output buffer too small");
        memcpy(out + at, bw.w.data(), total);
        out[at + 16] = (uint8_t)(total - 1);
        out[at + 17] = (uint8_t)((total - 1) >> 8);
        at += total;
    }
    *n_out = at;
    return GS_OK;
} catch (const std::bad_alloc &) {
    return gd_fail(GS_E_NOMEM, "out of host memory");
}

struct gs_deflater {
    int device = 0, n_cu = 256;
    hipStream_t stream = nullptr;
    uint8_t *d_slots = nullptr, *d_out = nullptr;
    size_t slots_cap = 0, out_cap = 0;
    uint32_t *d_sizes = nullptr, *d_crc = nullptr, *d_hist = nullptr;
    u64 *d_off = nullptr;
    size_t pieces_cap = 0;
    GdTables *d_tables = nullptr;
    GdTables *h_tables = nullptr;  // page-locked
    uint32_t *h_hist = nullptr;    // page-locked, 320 words
    u64 *h_total = nullptr;        // page-locked
    int64_t stored_members = 0, members = 0, bytes_in = 0, bytes_out = 0;
    // text waiting for gs_deflater_flush (gs_deflater_append): chunks of a few MiB are compressed together -- a call costs ~0.6 ms of
    // launches, the code's construction on the host and two waits whatever its size
    uint8_t *d_acc = nullptr;
    size_t acc_cap = 0;
    int64_t acc_n = 0;
};

extern "C" int gs_deflater_create(gs_deflater **out, int device) {
    if (!out) return gd_fail(GS_E_INVALID, "NULL argument");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return gd_fail(GS_E_NODEVICE, "no usable gfx950 device");
    if (device < 0 || device >= n) return gd_fail(GS_E_INVALID, "bad device");
    GD_TRY(hipSetDevice(device));
    gs_deflater *d = new gs_deflater();
    d->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) d->n_cu = prop.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&d->d_tables, sizeof(GdTables));
    if (e == hipSuccess) e = hipMalloc((void **)&d->d_hist, 320 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipHostMalloc((void **)&d->h_tables, sizeof(GdTables));
    if (e == hipSuccess) e = hipHostMalloc((void **)&d->h_hist, 320 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipHostMalloc((void **)&d->h_total, sizeof(u64));
    if (e != hipSuccess) {
        hipFree(d->d_tables);
        hipFree(d->d_hist);
        hipHostFree(d->h_tables);
        hipHostFree(d->h_hist);
        hipHostFree(d->h_total);
        if (d->stream) hipStreamDestroy(d->stream);
        delete d;
        return gd_fail(e == hipErrorOutOfMemory ? GS_E_NOMEM : GS_E_HIP, std::string("deflater: ") + hipGetErrorString(e));
    }
    *out = d;
    return GS_OK;
}

extern "C" int gs_deflater_destroy(gs_deflater *d) {
    if (!d) return GS_OK;
    hipSetDevice(d->device);
    if (d->stream) hipStreamSynchronize(d->stream);
    for (void *p : {(void *)d->d_slots, (void *)d->d_out, (void *)d->d_sizes, (void *)d->d_crc, (void *)d->d_hist, (void *)d->d_off, (void *)d->d_tables, (void *)d->d_acc}) hipFree(p);
    hipHostFree(d->h_tables);
    hipHostFree(d->h_hist);
    hipHostFree(d->h_total);
    if (d->stream) hipStreamDestroy(d->stream);
    delete d;
    return GS_OK;
}

// the most a call can write for n bytes of text: every piece stored
extern "C" int64_t gs_deflate_bound(int64_t n) {
    const int64_t pieces = (n + GD_PIECE_MIN - 1) / GD_PIECE_MIN;
    return n + pieces * (18 + 5 + 8);
}

template <typename T>
static int gd_grow(T **p, size_t *cap, size_t need) {
    if (*cap >= need && *p) return GS_OK;
    hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t want = need + need / 8 + 64;
    const hipError_t e = hipMalloc((void **)p, want * sizeof(T));
    if (e != hipSuccess) return gd_fail(e == hipErrorOutOfMemory ? GS_E_NOMEM : GS_E_HIP, std::string("gs_deflater: ") + hipGetErrorString(e));
    *cap = want;
    return GS_OK;
}

// n bytes of DEVICE text (complete: the caller has synchronised whatever produced them) -> BGZF members in `out` (host memory,
// page-locked for speed; out_cap >= gs_deflate_bound(n)); *n_out bytes.  No end-of-file block (the file's writer appends it).
extern "C" int gs_deflater_pack(gs_deflater *d, const uint8_t *d_text, int64_t n, uint8_t *out, int64_t out_cap, int64_t *n_out) try {
    if (!d || n < 0 || (n > 0 && !d_text) || !n_out || (n > 0 && !out)) return gd_fail(GS_E_INVALID, "bad argument");
    *n_out = 0;
    if (n == 0) return GS_OK;
    if (out_cap < gs_deflate_bound(n)) return gd_fail(GS_E_INVALID, "output buffer smaller than gs_deflate_bound");
    GD_TRY(hipSetDevice(d->device));
    const uint32_t piece = gd_piece_bytes(n), slot = piece + GD_SLOT_EXTRA;
    const int64_t n_pieces = (n + piece - 1) / piece;
    int rc;
    if (d->pieces_cap < (size_t)n_pieces + 1) {
        GD_TRY(hipStreamSynchronize(d->stream));
        hipFree(d->d_sizes);
        hipFree(d->d_crc);
        hipFree(d->d_off);
        d->d_sizes = d->d_crc = nullptr;
        d->d_off = nullptr;
        d->pieces_cap = 0;
        const size_t want = (size_t)n_pieces + (size_t)n_pieces / 4 + 64;
        GD_TRY(hipMalloc((void **)&d->d_sizes, want * sizeof(uint32_t)));
        GD_TRY(hipMalloc((void **)&d->d_crc, want * sizeof(uint32_t)));
        GD_TRY(hipMalloc((void **)&d->d_off, (want + 1) * sizeof(u64)));
        d->pieces_cap = want;
    }
    if (d->slots_cap < (size_t)n_pieces * slot) GD_TRY(hipStreamSynchronize(d->stream));
    if ((rc = gd_grow(&d->d_slots, &d->slots_cap, (size_t)n_pieces * slot))) return rc;
    if ((rc = gd_grow(&d->d_out, &d->out_cap, (size_t)gs_deflate_bound(n)))) return rc;
    // 1. symbol frequencies from a sample of the pieces
    const int64_t n_ranges = (n + GD_SAMPLE_BYTES - 1) / GD_SAMPLE_BYTES;
    const int n_sample = (int)std::min<int64_t>(n_ranges, GD_SAMPLE);
    GD_TRY(hipMemsetAsync(d->d_hist, 0, 320 * sizeof(uint32_t), d->stream));
    hipLaunchKernelGGL((gd_count_kernel), dim3((unsigned)((n_sample + GD_WAVES - 1) / GD_WAVES)), dim3(64 * GD_WAVES), 0, d->stream, d_text, n, n_ranges, n_sample, d->d_hist);
    GD_TRY(hipGetLastError());
    GD_TRY(hipMemcpyAsync(d->h_hist, d->d_hist, 320 * sizeof(uint32_t), hipMemcpyDeviceToHost, d->stream));
    // (the CRC-32 of the pieces meanwhile)
    if (gs_crc_tiles_device(d_text, n, piece, d->d_crc, d->stream) != GS_OK) return gd_fail(GS_E_HIP, std::string("CRC-32 of the pieces: ") + gs_inflate_last_error());
    GD_TRY(hipStreamSynchronize(d->stream));
    // 2. the code and the header every member of this call carries
    if (!gd_build_tables(d->h_hist, d->h_tables)) return gd_fail(GS_E_INVALID, "gs_deflater: the dynamic header does not fit its buffer");
    GD_TRY(hipMemcpyAsync(d->d_tables, d->h_tables, sizeof(GdTables), hipMemcpyHostToDevice, d->stream));
    // 3. one wave per piece
    const int64_t last_len = n - (n_pieces - 1) * (int64_t)piece;
    static const int blocks_per_cu = getenv("GS_DEFLATE_BLOCKS_PER_CU") ? std::max(1, atoi(getenv("GS_DEFLATE_BLOCKS_PER_CU"))) : 7;  // (seven waves per SIMD: 65 VGPRs when the compiler is told to, 21.3 KB of LDS per block; 4 / 5 / 6 / 7 blocks: 26.8 / 24.1 / 23.2 / 21.4 ms per 1.26 GB)
    const int grid = (int)std::min<int64_t>((n_pieces + GD_WAVES - 1) / GD_WAVES, (int64_t)d->n_cu * blocks_per_cu);
    hipLaunchKernelGGL((gd_deflate_kernel), dim3((unsigned)grid), dim3(64 * GD_WAVES), 0, d->stream, d_text, n, n_pieces, piece, d->d_tables, d->d_crc, gs_crc_init_term(piece),
                       gs_crc_init_term((uint64_t)last_len), d->d_slots, d->d_sizes);
    hipLaunchKernelGGL((gd_offsets_kernel), dim3(1), dim3(1024), 0, d->stream, d->d_sizes, n_pieces, d->d_off);
    hipLaunchKernelGGL((gd_gather_kernel), dim3((unsigned)n_pieces), dim3(256), 0, d->stream, d->d_slots, slot, d->d_sizes, d->d_off, n_pieces, d->d_out);
    GD_TRY(hipGetLastError());
    GD_TRY(hipMemcpyAsync(d->h_total, d->d_off + n_pieces, sizeof(u64), hipMemcpyDeviceToHost, d->stream));
    GD_TRY(hipStreamSynchronize(d->stream));
    const int64_t total = (int64_t)*d->h_total;
    if (total <= 0 || total > out_cap) return gd_fail(GS_E_HIP, "gs_deflater: the members' sizes do not add up");
    GD_TRY(hipMemcpyAsync(out, d->d_out, (size_t)total, hipMemcpyDeviceToHost, d->stream));
    GD_TRY(hipStreamSynchronize(d->stream));
    d->members += n_pieces;
    d->bytes_in += n;
    d->bytes_out += total;
    *n_out = total;
    return GS_OK;
} catch (const std::bad_alloc &) {
    return gd_fail(GS_E_NOMEM, "out of host memory");
}

// d_text[0, n) (device memory, complete) behind the text that is waiting; the source is free again when the call returns
extern "C" int gs_deflater_append(gs_deflater *d, const uint8_t *d_text, int64_t n) {
    if (!d || n < 0 || (n > 0 && !d_text)) return gd_fail(GS_E_INVALID, "bad argument");
    if (n == 0) return GS_OK;
    GD_TRY(hipSetDevice(d->device));
    const size_t need = (size_t)(d->acc_n + n) + 64;
    if (d->acc_cap < need) {
        uint8_t *p = nullptr;
        const size_t cap = std::max(need + need / 2, (size_t)48 << 20);
        GD_TRY(hipStreamSynchronize(d->stream));
        if (hipMalloc((void **)&p, cap) != hipSuccess) return gd_fail(GS_E_NOMEM, "gs_deflater_append: no device memory");
        if (d->acc_n > 0 && hipMemcpy(p, d->d_acc, (size_t)d->acc_n, hipMemcpyDeviceToDevice) != hipSuccess) {
            hipFree(p);
            return gd_fail(GS_E_HIP, "gs_deflater_append: copy");
        }
        hipFree(d->d_acc);
        d->d_acc = p;
        d->acc_cap = cap;
    }
    GD_TRY(hipMemcpyAsync(d->d_acc + d->acc_n, d_text, (size_t)n, hipMemcpyDeviceToDevice, d->stream));
    GD_TRY(hipMemsetAsync(d->d_acc + d->acc_n + n, 0, 64, d->stream));  // (the kernels read a little past the end)
    GD_TRY(hipStreamSynchronize(d->stream));
    d->acc_n += n;
    return GS_OK;
}
extern "C" int64_t gs_deflater_pending(const gs_deflater *d) { return d ? d->acc_n : 0; }
// the waiting text as BGZF members into out (room for gs_deflate_bound(gs_deflater_pending)); nothing waits afterwards
extern "C" int gs_deflater_flush(gs_deflater *d, uint8_t *out, int64_t out_cap, int64_t *n_out) {
    if (!d || !n_out) return gd_fail(GS_E_INVALID, "bad argument");
    *n_out = 0;
    if (d->acc_n == 0) return GS_OK;
    const int rc = gs_deflater_pack(d, d->d_acc, d->acc_n, out, out_cap, n_out);
    if (rc == GS_OK) d->acc_n = 0;
    return rc;
}

// [0] members written, [1] text bytes, [2] compressed bytes so far
extern "C" int gs_deflater_info(const gs_deflater *d, int64_t info[3]) {
    if (!d || !info) return gd_fail(GS_E_INVALID, "NULL argument");
    info[0] = d->members;
    info[1] = d->bytes_in;
    info[2] = d->bytes_out;
    return GS_OK;
}

// host text -> BGZF members in a host buffer, through the device (tests, tools)
extern "C" int gs_deflate_host(int device, const uint8_t *text, int64_t n, uint8_t *out, int64_t out_cap, int64_t *n_out) {
    if (n < 0 || (n > 0 && !text) || !n_out) return gd_fail(GS_E_INVALID, "bad argument");
    gs_deflater *d = nullptr;
    int rc = gs_deflater_create(&d, device);
    if (rc) return rc;
    uint8_t *d_text = nullptr;
    hipError_t e = hipMalloc((void **)&d_text, (size_t)n + 64);
    if (e == hipSuccess) e = hipMemcpy(d_text, text, (size_t)n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_text + n, 0, 64);
    if (e != hipSuccess) {
        hipFree(d_text);
        gs_deflater_destroy(d);
        return gd_fail(e == hipErrorOutOfMemory ? GS_E_NOMEM : GS_E_HIP, std::string("gs_deflate_host: ") + hipGetErrorString(e));
    }
    rc = gs_deflater_pack(d, d_text, n, out, out_cap, n_out);
    hipFree(d_text);
    gs_deflater_destroy(d);
    return rc;
}

// The records of a four-line chunk whose flag says so -- ((flags[r] & mask) != 0) == (want != 0) -- rewritten into d_out (room for
// the chunk's bytes): launched on `stream`, totals[0] = bytes, totals[1] = records land in h_totals (page-locked) when the stream
// has been synchronised.  d_len: n_records words, d_blocks: 2 * (n_records / 256 + 1) + 2 words of scratch.
extern "C" int gs_compact_records_device(hipStream_t stream, const uint8_t *d_text, const uint32_t *d_nl, int64_t n_records, const uint8_t *d_flags, int mask, int want,
                                         int with_probs, uint8_t *d_out, uint32_t *d_len, u64 *d_blocks, u64 *h_totals) {
    if (n_records < 0 || !h_totals) return gd_fail(GS_E_INVALID, "bad argument");
    h_totals[0] = h_totals[1] = 0;
    if (n_records == 0) return GS_OK;
    const int64_t n_blocks = (n_records + GC_BLOCK - 1) / GC_BLOCK;
    u64 *sum = d_blocks, *cnt = d_blocks + n_blocks, *tot = d_blocks + 2 * n_blocks;
    hipLaunchKernelGGL(gc_len_kernel, dim3((unsigned)n_blocks), dim3(GC_BLOCK), 0, stream, d_nl, n_records, d_flags, (uint32_t)mask, (uint32_t)(want != 0), with_probs, d_len, sum, cnt);
    hipLaunchKernelGGL(gc_scan_kernel, dim3(1), dim3(1024), 0, stream, sum, cnt, n_blocks, tot);
    hipLaunchKernelGGL(gc_copy_kernel, dim3((unsigned)n_blocks), dim3(GC_BLOCK), 0, stream, d_text, d_nl, n_records, d_len, sum, with_probs, d_out);
    GD_TRY(hipGetLastError());
    GD_TRY(hipMemcpyAsync(h_totals, tot, 2 * sizeof(u64), hipMemcpyDeviceToHost, stream));
    return GS_OK;
}

// descriptor lines of records[0 .. n) of a four-line chunk (device arrays) into d_out (n x stride bytes); asynchronous on `stream`
extern "C" int gs_gather_descriptors_device(hipStream_t stream, const uint8_t *d_text, const uint32_t *d_nl, const int64_t *d_records, int n, uint8_t *d_out, int stride) {
    if (n <= 0) return GS_OK;
    if (stride < 2) return gd_fail(GS_E_INVALID, "stride < 2");
    hipLaunchKernelGGL(gc_desc_kernel, dim3((unsigned)n), dim3(64), 0, stream, d_text, d_nl, d_records, n, d_out, stride);
    GD_TRY(hipGetLastError());
    return GS_OK;
}
