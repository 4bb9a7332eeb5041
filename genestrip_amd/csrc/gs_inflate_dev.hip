// gs_inflate_dev.hip -- DEFLATE on the device: block-gzip input (BGZF: bgzip / htslib, and this library's own .gz outputs), and -- in the
// second half of the file -- single-member gzip streams (gzip, pigz) taken apart into segments.
//
// The reference reads every .fastq.gz through java.util.zip.GZIPInputStream (B/io/StreamProvider.java:92-100): one thread, ~0.2 GB/s
// of text.  A BGZF file is a chain of gzip members of at most 64 KiB of text that state their compressed size (the 'BC' extra
// subfield), so the host finds every member by hopping from header to header without inflating anything, the COMPRESSED bytes go
// over PCIe (a fifth of the text), and the members are inflated side by side on the GPU straight into the text buffer the record
// scan reads (gs_match_submit_text with GS_MEM_DEVICE): ONE WAVE PER MEMBER.
//
// Block headers and Huffman tables are handled in lock step (every lane holds the same bit buffer; the replicated entries of a
// code are filled by the lanes in parallel).  The SYMBOLS of a block are decoded 64 bit offsets at a time (gi_token): lane l
// decodes the complete token -- literal, or length + extra bits + distance + extra bits, two table reads each from LDS -- that
// WOULD start l bits behind the reader; a short scalar walk (offset += token bits, one v_readlane per token) then picks the
// lanes that really are token starts, a prefix sum over their output lengths gives every token its place in the text, and the
// group's text is written ONE LANE PER SYMBOL, 64 symbols at a time: every token leaves a descriptor (its lane, literal or distance)
// at the index of its first symbol, a prefix maximum hands it to the symbols behind it, and symbol j is the literal or the symbol
// `dist` in front of it -- out[j] = out[j - dist], overlapping copies included --: from the LDS ring (below), from the text (far
// sources, one gather), or from lane j - dist of the same 64 after a few rounds of pointer jumping.  (Groups of more than 256
// symbols -- runs, maximal matches -- keep the first version's way: literals to their places, matches copied in order by all lanes.)
// A FASTQ file at level 1 is 99 % matches of 3 .. 6 bases a few hundred symbols back, 5.8 tokens and 46 symbols per 64 bits.
// The first version decoded one symbol per walk of the scalar unit: 19 scalar instructions per byte of text, the CU's one scalar
// issue port 80 % busy (rocprofv3 --pmc) -- the port, not memory, was the limit.  Here the per-token work is vector work.
// A block whose code does not fit the tables takes the one-symbol-at-a-time loop with the canonical decoder (gi_slow).
// The last GI_RING bytes of a wave's text are kept in LDS as well (a ring): a match whose distance fits the ring -- the header and
// quality lines of FASTQ always, and whatever a fast compression level finds in the bases -- is copied from there, a hundred
// cycles instead of a trip through L2 behind the wave's own stores (which was 90 % of the kernel's time).  A longer distance reads
// the text back through the same CU's L1 / L2, and the wave waits for its own stores only when such a match reaches into the
// bytes stored since its last wait.  Behind the last block: the member's ISIZE must be met exactly, and its CRC-32 is recomputed
// from the text (64 lanes x slicing-by-4 over equal slices, tables in LDS, combined with x^(8 n) mod P).
//
// gs_inflater_feed turns a run of members into "whole four-line records": the text is appended behind the tail the previous call
// left over, a second pair of kernels counts the newlines and finds the last one that closes a record (count a multiple of four),
// and the caller hands exactly that much to gs_match_submit_text / gs_filter_submit_text.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/gsgpu.h"
#include "gs_upload.h"

typedef unsigned long long u64;
struct __attribute__((aligned(16))) gs_u16x8 {
    uint16_t v[8];
};

#define GI_WAVES 4                 // waves per workgroup
#define GI_LROOT 9                 // root bits of the literal / length table
#define GI_DROOT 8                 // root bits of the distance table
#define GI_LSIZE 864               // entries: 2^9 root + sub-tables (zlib's proven bound for 286 symbols, root 9, 15 bits: 852)
#define GI_DSIZE 592               // 2^8 root + sub-tables; a code that needs more is decoded the canonical way (gi_slow)
#ifndef GI_RING
#define GI_RING 2048               // bytes of recent text per wave in LDS (a power of two); with it a wave takes 8.6 KB: 16 waves per CU
#endif
#define GI_RING_SYM 1024           // symbols of recent text per wave of gi_segment_kernel (2 KB: with it five workgroups fit a CU)
#define GI_LANE_SYMS 256u          // a group of at most this many symbols is written one lane per symbol (gi_decode_blocks)
#define GI_PAR_MAX 16u             // matches of at most this length are copied one lane per match (gi_inflate_kernel)
#define GI_ON(mask) __builtin_amdgcn_inverse_ballot_w64(mask)

// table entry (16 bits -- the tables of a wave take 4.7 KB of LDS, so that 32 waves fit a CU): bits 0-3 code bits to drop (in a
// sub-table: the bits beyond the root), bits 4-6 kind, bits 7-15 value: the literal, the length / distance SYMBOL (base and extra
// bits are arithmetic) -- or, kind GI_SUB, where the sub-table starts behind the root table, with its index bits in bits 0-3
enum { GI_BAD = 0, GI_LIT = 1, GI_LEN = 2, GI_EOB = 3, GI_SUB = 4, GI_DIST = 5 };
#define GI_ENTRY(bits, kind, value) ((uint16_t)((uint32_t)(bits) | ((uint32_t)(kind) << 4) | ((uint32_t)(value) << 7)))

// status of a member (0 = inflated, ISIZE and CRC-32 as announced)
enum { GI_OK = 0, GI_E_HEADER = 1, GI_E_TABLE = 2, GI_E_CODE = 3, GI_E_DIST = 4, GI_E_OVERRUN = 5, GI_E_SIZE = 6, GI_E_CRC = 7, GI_E_INPUT = 8 };

struct GiCode {            // what the canonical decoder needs (and the table builder starts from)
    uint16_t count[16];    // codes per length
    uint16_t *work;        // symbols ordered by (length, symbol)
};

struct GiWave {
    uint32_t cbuf[132];      // two 256-byte pieces of the compressed payload (piece j in half j & 1) for the token decoder; [128 ..] = [0 .. 3] again
    uint16_t ltab[GI_LSIZE];
    uint16_t dtab[GI_DSIZE];
    union {
        uint16_t lwork[288];  // (the table builder and the canonical decoder)
        uint32_t slots[128];  // 64 symbols of a group: token descriptors by output index, and the symbols' states (gi_decode_blocks; only
                              // while a block's symbols are decoded through the tables -- the canonical decoder never runs next to it)
    };
    uint16_t dwork[32];
    uint16_t lcount[16], dcount[16], offs[16];
    union {
        uint8_t lens[320 + 32];  // (the lengths of a dynamic block are decoded at an offset of 24 and moved into place)
        uint32_t lut[64];        // while a block's symbols are decoded (the lengths have become tables by then): base | extra bits << 16
                                 // of the length symbols (0..28) and, from 32 on, of the distance symbols (0..29)
    };
};

__constant__ uint8_t gi_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
__constant__ uint32_t gi_crc_table[4 * 256];  // slicing-by-4 tables of the reflected CRC-32

__device__ __forceinline__ int gi_lane() { return (int)__lane_id(); }
__device__ __forceinline__ uint32_t gi_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ void gi_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// base and extra bits of a length symbol s = code - 257 (0..28) and of a distance symbol d (0..29): RFC 1951 3.2.5 in closed form
__device__ __forceinline__ void gi_len_sym(uint32_t s, uint32_t &base, uint32_t &extra) {
    if (s < 8) {
        base = 3 + s;
        extra = 0;
    } else if (s == 28) {
        base = 258;
        extra = 0;
    } else {
        extra = (s >> 2) - 1;
        base = 3 + ((4 + (s & 3)) << extra);
    }
}
__device__ __forceinline__ void gi_dist_sym(uint32_t d, uint32_t &base, uint32_t &extra) {
    if (d < 4) {
        base = 1 + d;
        extra = 0;
    } else {
        extra = (d >> 1) - 1;
        base = 1 + ((2 + (d & 1)) << extra);
    }
}

// ---- the bit reader: 64-bit buffer, refilled 32 bits at a time from the wave's current 256-byte piece of the input
struct GiBits {
    const uint8_t *in;    // the member's deflate payload
    uint32_t in_len;      // its length
    uint32_t piece_at;    // byte offset of the piece in `cur`
    uint32_t idx;         // next dword of the piece (0..64)
    uint32_t cur, nxt;    // lane i: dword i of the current / the following piece
    u64 bb;               // bit buffer (low bits first)
    int bc;               // valid bits in it
    uint32_t taken;       // dwords handed to the bit buffer so far
    __device__ __forceinline__ uint32_t load_piece(uint32_t at, int lane) const {
        // dword `lane` of the piece at byte offset `at`: bytes beyond the payload read as 0 (an overrun is caught by `taken`)
        const uint32_t o = at + 4u * (uint32_t)lane;
        uint32_t v = 0;
        if (o + 4u <= in_len) {
            typedef uint32_t __attribute__((aligned(1))) u32_any;  // (the payload starts at any byte; the hardware reads unaligned dwords)
            v = *reinterpret_cast<const u32_any *>(in + o);
        } else if (o < in_len) {
            for (uint32_t b = 0; o + b < in_len; b++) v |= (uint32_t)in[o + b] << (8 * b);
        }
        return v;
    }
    __device__ __forceinline__ void start(const uint8_t *p, uint32_t n, int lane) {
        in = p;
        in_len = n;
        piece_at = 0;
        idx = 0;
        taken = 0;
        bb = 0;
        bc = 0;
        cur = load_piece(0, lane);
        nxt = load_piece(256, lane);
    }
    __device__ __forceinline__ void refill(int lane) {
        while (bc <= 32) {
            if (idx == 64) {
                cur = nxt;
                piece_at += 256;
                nxt = load_piece(piece_at + 256, lane);
                idx = 0;
            }
            const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)idx);
            bb |= (u64)w << bc;
            bc += 32;
            idx++;
            taken++;
        }
    }
    // continue at byte `at` of the payload (behind a stored block)
    __device__ __forceinline__ void seek(uint32_t at, int lane) {
        piece_at = at & ~3u;
        cur = load_piece(piece_at, lane);
        nxt = load_piece(piece_at + 256, lane);
        idx = 0;
        bb = 0;
        bc = 0;
        taken = piece_at / 4;
        refill(lane);
        const int skip = 8 * (int)(at & 3u);
        bb >>= skip;
        bc -= skip;
    }
    __device__ __forceinline__ void seek_bit(u64 bit, int lane) {  // continue at bit `bit` of the payload
        seek((uint32_t)(bit >> 3), lane);
        const int skip = (int)(bit & 7u);
        bb >>= skip;
        bc -= skip;
    }
    __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)bb & ((1u << n) - 1u); }
    __device__ __forceinline__ void drop(int n) {
        bb >>= n;
        bc -= n;
    }
    __device__ __forceinline__ uint32_t get(int n) {
        const uint32_t v = peek(n);
        drop(n);
        return v;
    }
    // bits consumed so far; a stream that ran beyond its payload has consumed more than 8 * in_len
    __device__ __forceinline__ u64 consumed() const { return (u64)taken * 32u - (u64)bc; }
};

// ---- Huffman code from code lengths (canonical codes, deflate bit order): counts + symbols in canonical order (always), and the
// root table + sub-tables for the fast path (*fast = false when the sub-tables do not fit `cap`: the block is then decoded the
// canonical way, gi_slow -- correct for every code, an order of magnitude slower, and not needed by any stream zlib writes).
// type 0: literal / length codes (n = 288), 1: distance codes (n = 32), 2: code length codes (n = 19, root 7, no sub-tables).
// All lanes run the same statements; only the replication of an entry over its copies is spread over the lanes.
// Returns false for an over-subscribed or (where deflate forbids it) incomplete set of lengths.
__device__ bool gi_build(GiWave &w, int type, int n, int lens_at, uint16_t *tab, uint16_t *count, uint16_t *work,
                         int root, int cap, int lane, bool *fast) {
    *fast = true;
    for (int i = lane; i < 16; i += 64) count[i] = 0;
    for (int i = lane; i < (1 << root); i += 64) tab[i] = 0;
    gi_lds_sync();
    if (lane == 0)
        for (int s = 0; s < n; s++) count[w.lens[lens_at + s]]++;
    gi_lds_sync();
    int max = 15;
    while (max >= 1 && count[max] == 0) max--;
    if (max == 0) return true;  // no code at all (the distance code of a block of literals): every lookup ends in an invalid entry
    int left = 1;
    for (int len = 1; len <= 15; len++) {
        left <<= 1;
        left -= (int)count[len];
        if (left < 0) return false;  // over-subscribed
    }
    if (left > 0 && (type == 2 || max != 1)) return false;  // incomplete: only a single code of one bit may stand alone
    if (lane == 0) {
        w.offs[1] = 0;
        for (int len = 1; len < 15; len++) w.offs[len + 1] = (uint16_t)(w.offs[len] + count[len]);
        for (int s = 0; s < n; s++) {
            const int l = w.lens[lens_at + s];
            if (l) work[w.offs[l]++] = (uint16_t)s;
        }
    }
    gi_lds_sync();
    // codes in canonical order; `huff` is the bit-reversed code (the stream delivers codes low bit first)
    uint32_t huff = 0;
    int len = 1;
    while (count[len] == 0) len++;
    int remaining = (int)count[len];
    int used = 1 << root;   // entries in use (root table first)
    int sub_base = 0, sub_bits = 0, n_sub = 0;
    uint32_t sub_prefix = 0xffffffffu;
    int n_codes = 0;
    for (int l = 1; l <= 15; l++) n_codes += (int)count[l];
    for (int i = 0; i < n_codes; i++) {
        const int sym = (int)work[i];
        uint32_t kind, value;
        if (type == 0) {
            kind = sym < 256 ? GI_LIT : (sym == 256 ? GI_EOB : (sym <= 285 ? GI_LEN : GI_BAD));  // 286, 287 never appear in a valid stream
            value = sym < 256 ? (uint32_t)sym : (sym <= 285 ? (uint32_t)(sym - 257) & 31u : 0u);
        } else if (type == 1) {
            kind = sym <= 29 ? GI_DIST : GI_BAD;
            value = (uint32_t)sym;
        } else {
            kind = GI_LIT;
            value = (uint32_t)sym;
        }
        if (len <= root) {
            const uint16_t e = GI_ENTRY(len, kind, value);
            const int reps = 1 << (root - len);
            for (int r = lane; r < reps; r += 64) tab[huff + ((uint32_t)r << len)] = e;
        } else {
            const uint32_t prefix = huff & ((1u << root) - 1u);
            if (prefix != sub_prefix) {  // a new sub-table: as many bits as the longest code under this prefix needs
                sub_prefix = prefix;
                int curr = len - root, room = 1 << curr, l2 = len, cnt = remaining;
                while (l2 < max) {
                    room -= cnt;
                    if (room <= 0) break;
                    curr++;
                    l2++;
                    room <<= 1;
                    cnt = (int)count[l2];
                }
                sub_bits = curr;
                sub_base = used;
                used += 1 << curr;
                if (used > cap) {
                    *fast = false;
                    gi_lds_sync();
                    return true;
                }
                for (int r = lane; r < (1 << curr); r += 64) tab[sub_base + r] = 0;
                if (lane == 0) tab[prefix] = GI_ENTRY(sub_bits, GI_SUB, sub_base - (1 << root));  // (at most 352 / 336 entries behind the root: nine bits)
                n_sub++;
            }
            const uint16_t e = GI_ENTRY(len - root, kind, value);
            const int reps = 1 << (sub_bits - (len - root));
            for (int r = lane; r < reps; r += 64) tab[sub_base + (huff >> root) + ((uint32_t)r << (len - root))] = e;
        }
        // next code of this length in bit-reversed form
        uint32_t incr = 1u << (len - 1);
        while (huff & incr) incr >>= 1;
        huff = incr ? (huff & (incr - 1)) + incr : 0;
        if (--remaining == 0 && i + 1 < n_codes) {
            len++;
            while (count[len] == 0) len++;
            remaining = (int)count[len];
        }
    }
    gi_lds_sync();
    return true;
}

// fast path: (kind << 16 | value << 4 | bits) of the next code; bits = 0: no such code
__device__ __forceinline__ uint32_t gi_lookup(const uint16_t *tab, int root, const GiBits &b) {
    uint32_t e = gi_uni(tab[b.peek(root)]);
    uint32_t bits = e & 15u;
    if (((e >> 4) & 7u) == GI_SUB) {
        e = gi_uni(tab[(1u << root) + (e >> 7) + (((uint32_t)(b.bb >> root)) & ((1u << bits) - 1u))]);
        bits = (e & 15u) ? (e & 15u) + (uint32_t)root : 0u;
    }
    return (((e >> 4) & 7u) << 16) | ((e >> 7) << 4) | bits;
}

// the canonical decoder (no table): walks the lengths, one bit at a time; same result format.  type as in gi_build.
__device__ uint32_t gi_slow(const uint16_t *count, const uint16_t *work, int type, const GiBits &b) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; len++) {
        code |= (int)((b.bb >> (len - 1)) & 1u);
        const int cnt = (int)count[len];
        if (code - cnt < first) {
            const int sym = (int)gi_uni(work[index + (code - first)]);
            uint32_t kind, value;
            if (type == 0) {
                kind = sym < 256 ? GI_LIT : (sym == 256 ? GI_EOB : (sym <= 285 ? GI_LEN : GI_BAD));
                value = sym < 256 ? (uint32_t)sym : (sym <= 285 ? (uint32_t)(sym - 257) & 31u : 0u);
            } else {
                kind = sym <= 29 ? GI_DIST : GI_BAD;
                value = (uint32_t)sym;
            }
            return (kind << 16) | (value << 4) | (uint32_t)len;
        }
        index += cnt;
        first += cnt;
        first <<= 1;
        code <<= 1;
    }
    return 0;
}

// ---- the token decoder: what starts at this lane's bit offset (w0 = the next 32 bits, w1 = the 32 behind them)
//   kind: GI_LIT / GI_LEN / GI_EOB / GI_BAD; t: bits of the whole token (a match: length code + extra + distance code + extra,
//   at most 48); olen: bytes of text it produces; dist: the distance of a match; lit: the byte of a literal
__device__ __forceinline__ void gi_token(const GiWave &w, uint32_t w0, uint32_t w1, uint32_t &kind, uint32_t &t, uint32_t &olen, uint32_t &dist,
                                         uint32_t &lit) {
    const u64 bits = ((u64)w1 << 32) | w0;
    uint32_t e = w.ltab[w0 & ((1u << GI_LROOT) - 1u)];
    uint32_t n = e & 15u;
    kind = (e >> 4) & 7u;
    if (kind == GI_SUB) {
        e = w.ltab[(1u << GI_LROOT) + (e >> 7) + ((w0 >> GI_LROOT) & ((1u << n) - 1u))];
        n = (e & 15u) ? (e & 15u) + GI_LROOT : 0u;
        kind = (e >> 4) & 7u;
    }
    const uint32_t val = e >> 7;
    if (n == 0 || (kind != GI_LIT && kind != GI_LEN && kind != GI_EOB)) kind = GI_BAD;
    lit = val & 0xffu;
    t = n;
    olen = kind == GI_LIT ? 1u : 0u;
    dist = 0;
    if (kind == GI_LEN) {
        const uint32_t lx = w.lut[val & 31u];
        const uint32_t lbase = lx & 0xffffu, lextra = lx >> 16;
        olen = lbase + ((uint32_t)(bits >> n) & ((1u << lextra) - 1u));
        const uint32_t u = (uint32_t)(bits >> (n + lextra));  // (n + lextra <= 20: 44 bits are left, a distance takes at most 28)
        uint32_t de = w.dtab[u & ((1u << GI_DROOT) - 1u)];
        uint32_t dn = de & 15u, dk = (de >> 4) & 7u;
        if (dk == GI_SUB) {
            de = w.dtab[(1u << GI_DROOT) + (de >> 7) + ((u >> GI_DROOT) & ((1u << dn) - 1u))];
            dn = (de & 15u) ? (de & 15u) + GI_DROOT : 0u;
            dk = (de >> 4) & 7u;
        }
        const uint32_t dx = w.lut[32u + ((de >> 7) & 31u)];
        const uint32_t dbase = dx & 0xffffu, dextra = dx >> 16;
        dist = dbase + ((u >> dn) & ((1u << dextra) - 1u));
        t = n + lextra + dn + dextra;
        if (dk != GI_DIST || dn == 0) kind = GI_BAD;
    }
}

// inclusive prefix sum over the 64 lanes (DPP: four shifts inside the rows of 16, two row broadcasts)
__device__ __forceinline__ uint32_t gi_scan_incl(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);   // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);   // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);   // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);   // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return x;
}

// inclusive prefix maximum over the 64 lanes (the same six steps)
__device__ __forceinline__ uint32_t gi_scan_max(uint32_t x) {
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));
    x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));
    return x;
}

// x^(8 n) mod P (reflected CRC-32 polynomial arithmetic), and a * b mod P
__device__ uint32_t gi_gf_mul(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) {
        if (a & 0x80000000u) p ^= b;
        a <<= 1;
        b = (b >> 1) ^ ((b & 1u) ? 0xedb88320u : 0u);
    }
    return p;
}
__device__ uint32_t gi_x_pow_8n(uint32_t n) {
    uint32_t r = 0x80000000u, sq = 0x00800000u;  // x^0, x^8 (bit 31 = x^0 in the reflected representation)
    while (n) {
        if (n & 1u) r = gi_gf_mul(r, sq);
        sq = gi_gf_mul(sq, sq);
        n >>= 1;
    }
    return r;
}

struct GiBlock {
    u64 in_off;        // deflate payload of the member inside the compressed buffer
    uint32_t in_len;
    uint32_t out_len;  // ISIZE
    u64 out_off;       // where its text goes
    uint32_t crc;
    uint32_t pad;
};

// the header of a dynamic block behind its three type bits: code length code, the lengths of the two codes, their tables
// (GI_OK, or what is wrong with it -- the block finder of a single-member stream lives off the second answer)
__device__ int gi_dynamic_header(GiWave &w, GiBits &b, int lane, bool *lfast_out, bool *dfast_out) {
    bool lfast = true, dfast = true;
    b.refill(lane);
    const int hlit = (int)b.get(5) + 257, hdist = (int)b.get(5) + 1, hclen = (int)b.get(4) + 4;
    if (hlit > 286 || hdist > 30) {
        return GI_E_HEADER;
    }
    for (int s = lane; s < 19; s += 64) w.lens[s] = 0;
    gi_lds_sync();
    for (int i = 0; i < hclen; i++) {
        b.refill(lane);
        const uint32_t v = b.get(3);
        if (lane == 0) w.lens[gi_cl_order[i]] = (uint8_t)v;
    }
    gi_lds_sync();
    bool clfast = true;  // (19 codes of at most 7 bits under a 7-bit root: no sub-tables, always fast)
    if (!gi_build(w, 2, 19, 0, w.dtab, w.dcount, w.dwork, 7, GI_DSIZE, lane, &clfast)) {  // (in dtab for a moment)
        return GI_E_TABLE;
    }
    int at = 0, prev = 0;
    const int total = hlit + hdist;
    // the lengths are decoded into lens[19 ..] (behind the code length code's own lengths) and moved down afterwards
    while (at < total) {
        b.refill(lane);
        const uint32_t e = gi_lookup(w.dtab, 7, b);
        if ((e >> 16) != GI_LIT || (e & 15u) == 0) {
            return GI_E_CODE;
        }
        b.drop((int)(e & 15u));
        const int sym = (int)((e >> 4) & 0xffu);
        int rep = 1, val = sym;
        if (sym == 16) {
            if (at == 0) return GI_E_CODE;
            rep = 3 + (int)b.get(2);
            val = prev;
        } else if (sym == 17) {
            rep = 3 + (int)b.get(3);
            val = 0;
        } else if (sym == 18) {
            rep = 11 + (int)b.get(7);
            val = 0;
        }
        if (at + rep > total) {
            return GI_E_CODE;
        }
        // literal / length lengths at lens[24 + i] for now, distance lengths behind them
        for (int r = lane; r < rep; r += 64) w.lens[24 + at + r] = (uint8_t)val;
        at += rep;
        prev = val;
    }
    gi_lds_sync();
    {   // into place: lens[0 .. 288) literal / length (unused ones 0), lens[288 .. 320) distances
        uint8_t mine[5];
        for (int q = 0; q < 5; q++) {
            const int s = lane + 64 * q;  // 0 .. 319
            uint8_t v = 0;
            if (s < 288) {
                if (s < hlit) v = w.lens[24 + s];
            } else if (s - 288 < hdist)
                v = w.lens[24 + hlit + (s - 288)];
            mine[q] = v;
        }
        gi_lds_sync();
        for (int q = 0; q < 5; q++) w.lens[lane + 64 * q] = mine[q];
        gi_lds_sync();
    }
    if (w.lens[256] == 0) {  // a block without an end-of-block code never ends
        return GI_E_TABLE;
    }
    if (!gi_build(w, 0, 288, 0, w.ltab, w.lcount, w.lwork, GI_LROOT, GI_LSIZE, lane, &lfast) ||
        !gi_build(w, 1, 32, 288, w.dtab, w.dcount, w.dwork, GI_DROOT, GI_DSIZE, lane, &dfast)) {
        return GI_E_TABLE;
    }
    *lfast_out = lfast;
    *dfast_out = dfast;
    return GI_OK;
}

// ---- the blocks of one unit of work, from the reader's position: a BGZF member to its final block (MARK = false: bytes), or a
// SEGMENT of a single-member stream from one block boundary to the next one that was found (MARK = true, gi_segment_kernel:
// 16-bit symbols; what a match copies from in front of the segment is a MARKER 0x8000 | window position, resolved later --
// dst[-GI_WINDOW .. 0) and the ring hold those markers when the call starts, so that no copy has to know).
template <bool MARK>
struct GiOut {
    typedef uint8_t T;
};
template <>
struct GiOut<true> {
    typedef uint16_t T;
};
#define GI_WINDOW 32768u
enum { GI_E_SYNC = 9,     // a segment did not end on the block boundary the next segment starts at
       GI_FINAL = 10 };    // (not an error) the member's final block lay inside the segment: it is the member's last, its text is good

template <bool MARK>
__device__ int gi_decode_blocks(GiWave &w, typename GiOut<MARK>::T *ring, GiBits &b, uint32_t in_len, u64 stop_bit, bool to_final,
                                typename GiOut<MARK>::T *dst, uint32_t cap, int force_slow, int lane, uint32_t *produced) {
    typedef typename GiOut<MARK>::T OutT;
    constexpr uint32_t RING = MARK ? GI_RING_SYM : GI_RING;
    constexpr uint32_t RM = RING - 1u;
    constexpr uint32_t BACK = MARK ? GI_WINDOW : 0u;  // how far in front of the unit's first symbol a match may reach
    bool final_inside = false;
    uint32_t pos = 0;          // bytes of text produced (stored or pending)
    uint32_t npend = 0;        // pending literals (lane j < npend holds byte pos - npend + j)
    uint32_t pbyte = 0;
    uint32_t visible = 0;      // every byte below this offset is known to have reached memory
    int err = GI_OK;
    bool last = false, lfast = true, dfast = true;
    auto flush = [&]() {
        if (npend) {
            const uint32_t at = pos - npend + (uint32_t)lane;
            if ((uint32_t)lane < npend && at < cap) {  // (never beyond the member's own text)
                dst[at] = (OutT)pbyte;
                ring[at & RM] = (OutT)pbyte;
            }
            npend = 0;
        }
    };
    while (!last && err == GI_OK) {
        if (MARK && !to_final && b.consumed() >= stop_bit) break;  // the next segment's first block (checked behind the loop)
        b.refill(lane);
        last = b.get(1) != 0;
        if (MARK && last && !to_final) final_inside = true;  // the member ends inside this segment (another member follows: GI_FINAL)
        const uint32_t btype = b.get(2);
        if (btype == 0) {  // stored: to the byte boundary, LEN, ~LEN, bytes
            b.drop(b.bc & 7);
            b.refill(lane);
            const uint32_t len = b.get(16), nlen = b.get(16);
            if ((len ^ nlen) != 0xffffu) {
                err = GI_E_HEADER;
                break;
            }
            flush();
            if (pos + len > cap) {
                err = GI_E_OVERRUN;
                break;
            }
            // the reader stands on a byte boundary: the bytes are copied straight from the payload, the reader re-seated behind them
            const uint32_t src = (uint32_t)(b.consumed() >> 3);
            if ((u64)src + len > in_len) {
                err = GI_E_INPUT;
                break;
            }
            for (uint32_t i = (uint32_t)lane; i < len; i += 64) {
                const OutT c = (OutT)b.in[src + i];
                dst[pos + i] = c;
                if (len - i <= RING) ring[(pos + i) & RM] = c;  // (the last GI_RING bytes of the block)
            }
            b.seek(src + len, lane);
            pos += len;
            continue;
        }
        if (btype == 3) {
            err = GI_E_HEADER;
            break;
        }
        if (btype == 1) {  // fixed codes
            for (int s = lane; s < 288; s += 64) w.lens[s] = s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8));
            for (int s = lane; s < 32; s += 64) w.lens[288 + s] = 5;
            gi_lds_sync();
            if (!gi_build(w, 0, 288, 0, w.ltab, w.lcount, w.lwork, GI_LROOT, GI_LSIZE, lane, &lfast) ||
                !gi_build(w, 1, 32, 288, w.dtab, w.dcount, w.dwork, GI_DROOT, GI_DSIZE, lane, &dfast)) {
                err = GI_E_TABLE;
                break;
            }
        } else {  // dynamic codes
            err = gi_dynamic_header(w, b, lane, &lfast, &dfast);
            if (err != GI_OK) break;
        }
        if (force_slow & 1) lfast = dfast = false;  // (test hook: every code through the canonical decoder)
        if (lfast && dfast) {
            // ---- the symbols of the block, 64 bit offsets at a time (see the head of the file)
            flush();
            // Is this block's text mostly literals of two or three bits (bases under a code of their own: what this library's writer makes
            // of FASTQ; quality strings of few symbols)?  Then 64 bits hold 25 tokens and more, and the scalar walk over them -- six
            // dependent instructions and a taken branch per token -- is what a group waits for: such a block takes the walk four tokens
            // at a time (below).  Decided once per block from the code lengths: half the code space in literals of at most three bits.
            bool dense;
            {
                uint32_t mass = 0;  // in eighths of the code space
                for (int sidx = lane; sidx < 256; sidx += 64) {
                    const uint32_t l = w.lens[sidx];
                    mass += (l >= 1u && l <= 3u) ? (8u >> l) : 0u;
                }
                dense = (uint32_t)__builtin_amdgcn_readlane((int)gi_scan_incl(mass), 63) >= 4u;
                gi_lds_sync();
            }
            {   // the token decoder's table of bases and extra bits (RFC 1951 3.2.5), where the block's code lengths were
                uint32_t base = 0, extra = 0;
                if (lane < 29)
                    gi_len_sym((uint32_t)lane, base, extra);
                else if (lane >= 32 && lane < 62)
                    gi_dist_sym((uint32_t)lane - 32u, base, extra);
                w.lut[lane] = base | (extra << 16);
            }
            u64 P = b.consumed();                         // the reader's position, bits from the start of the payload
            const u64 plimit = (u64)in_len * 8u;      // a token that starts behind it: the stream has run off its payload
            uint32_t k = (uint32_t)(P >> 11);             // pieces k and k + 1 are in w.cbuf, piece k + 2 is on its way
            {
                const uint32_t p0 = b.load_piece(k * 256u, lane), p1 = b.load_piece((k + 1u) * 256u, lane);
                w.cbuf[(k & 1u) * 64u + (uint32_t)lane] = p0;
                w.cbuf[((k + 1u) & 1u) * 64u + (uint32_t)lane] = p1;
                if (lane < 4) w.cbuf[128 + lane] = (k & 1u) ? p1 : p0;  // (a lane's 96 bits are read without wrapping the index)
            }
            uint32_t ahead = b.load_piece((k + 2u) * 256u, lane);
            gi_lds_sync();
            bool eob = false;
            while (!eob) {
                if (P > plimit) {  // (every group takes at least one bit, so the loop ends)
                    err = GI_E_INPUT;
                    break;
                }
                if ((uint32_t)(P >> 11) != k) {  // one piece further: piece k + 2 takes the place of piece k
                    w.cbuf[(k & 1u) * 64u + (uint32_t)lane] = ahead;
                    if ((k & 1u) == 0u && lane < 4) w.cbuf[128 + lane] = ahead;
                    k++;
                    ahead = b.load_piece((k + 2u) * 256u, lane);
                    gi_lds_sync();
                }
                const uint32_t q = (uint32_t)(P & 4095u) + (uint32_t)lane;  // this lane's bit offset inside the 512 staged bytes
                const uint32_t d0 = q >> 5, sh = q & 31u;
                const uint32_t x0 = w.cbuf[d0], x1 = w.cbuf[d0 + 1u], x2 = w.cbuf[d0 + 2u];  // (d0 <= 129)
                uint32_t kind, t, olen, dist, lit;
                gi_token(w, __builtin_amdgcn_alignbit(x1, x0, sh), __builtin_amdgcn_alignbit(x2, x1, sh), kind, t, olen, dist, lit);
                // which lanes are token starts: lane 0 is one, and every token names the next
                const bool stop = kind == GI_EOB || kind == GI_BAD;
                const uint32_t step = stop ? 64u : t;  // (an end marker ends the walk)
                u64 chain = 0;
                uint32_t adv;  // bits of this group
                if (dense) {
                    // The lanes follow the links first -- where the second, third and fourth token behind every offset starts (three
                    // cross-lane reads) --, the scalar walk takes every FOURTH token, and the three behind each of those are marked
                    // through LDS.
                    const uint32_t j1 = min((uint32_t)lane + step, 64u);
                    auto hop = [&](uint32_t from, uint32_t table) {  // table[from], or 64 behind the end
                        const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((from & 63u) << 2), (int)table);
                        return from < 64u ? v : 64u;
                    };
                    const uint32_t j2 = hop(j1, j1), j3 = hop(j2, j1), j4 = hop(j2, j2);
                    u64 fourth = 0;
                    uint32_t at = 0;
                    do {
                        fourth |= 1ULL << at;
                        at = (uint32_t)__builtin_amdgcn_readlane((int)j4, (int)at);
                    } while (at < 64u);
                    uint32_t *const fl = w.slots;
                    fl[lane] = 0u;
                    gi_lds_sync();
                    if (GI_ON(fourth)) {
                        if (j1 < 64u) fl[j1] = 1u;
                        if (j2 < 64u) fl[j2] = 1u;
                        if (j3 < 64u) fl[j3] = 1u;
                    }
                    gi_lds_sync();
                    chain = fourth | __ballot(fl[lane] != 0u);
                    gi_lds_sync();
                    const int el = 63 - __builtin_clzll(chain);
                    adv = (uint32_t)el + (uint32_t)__builtin_amdgcn_readlane((int)step, el);
                } else {
                    uint32_t at = 0;
                    do {  // (a token has at least one bit: at most 64 trips)
                        chain |= 1ULL << at;
                        at += (uint32_t)__builtin_amdgcn_readlane((int)step, (int)at);
                    } while (at < 64u);
                    adv = at;
                }
                if ((__ballot(stop) & chain) != 0) {  // the chain's last token is an end marker
                    const int el = 63 - __builtin_clzll(chain);
                    if ((uint32_t)__builtin_amdgcn_readlane((int)kind, el) == GI_BAD) {
                        err = GI_E_CODE;
                        break;
                    }
                    eob = true;
                    adv = (uint32_t)el + (uint32_t)__builtin_amdgcn_readlane((int)t, el);
                }
                const bool on = GI_ON(chain);
                const uint32_t ol = on ? olen : 0u;
                const uint32_t incl = gi_scan_incl(ol);
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                const uint32_t mpos = pos + (incl - ol);  // where this lane's token writes
                if (pos + total > cap) {
                    err = GI_E_OVERRUN;
                    break;
                }
                const bool is_match = on && kind == GI_LEN;
                const u64 mm = __ballot(is_match);
                if (!MARK && mm != 0 && __ballot(is_match && dist > mpos + BACK) != 0) {  // (a segment's window takes every distance deflate can state)
                    err = GI_E_DIST;
                    break;
                }
                if (total <= GI_LANE_SYMS) {
                    // ---- a group of at most 256 symbols (all but runs and maximal matches): ONE LANE PER SYMBOL OF TEXT, 64 symbols
                    // at a time.  Every token leaves a descriptor (its lane, literal byte or distance) at the index of its first
                    // symbol, a prefix maximum hands it to the symbols behind it (the lane number in the top bits grows with the
                    // index), and symbol j is the literal, or -- out[j] = out[j - dist], which holds for overlapping copies too --
                    // the symbol `dist` in front of it: in front of these 64 (the ring; the text when the distance is larger than
                    // the ring), or among them, and then lane j - dist knows it, now or after a few rounds of pointer jumping (a
                    // chain of copies inside the 64; a run of n symbols takes log2 n rounds).  All reads of the ring come before
                    // the writes of the same 64 symbols.
                    uint32_t *const sa = w.slots, *const sb = w.slots + 64;
                    const uint32_t rel = mpos - pos;
                    const uint32_t desc = ((uint32_t)lane << 26) | (1u << 25) | (kind == GI_LIT ? (1u << 24) | lit : dist);
                    uint32_t carry = 0;
                    for (uint32_t base = 0; base < total; base += 64u) {
                        sa[lane] = 0;
                        gi_lds_sync();
                        if (ol != 0u && rel - base < 64u) sa[rel - base] = desc;
                        gi_lds_sync();
                        const uint32_t d = max(gi_scan_max(sa[lane]), carry);
                        carry = (uint32_t)__builtin_amdgcn_readlane((int)d, 63);
                        const uint32_t at0 = pos + base;  // the first of these 64 symbols
                        const bool act = base + (uint32_t)lane < total;
                        const bool islit = ((d >> 24) & 1u) != 0;
                        const uint32_t dd = d & 0xffffu;
                        int32_t p = (int32_t)lane - (int32_t)dd;  // (of a match) index of the source among the 64, or negative
                        const uint32_t ab = at0 + (uint32_t)p;    // its offset in the text
                        uint32_t val = ring[ab & RM];
                        const bool far = act && !islit && dd > RING;
                        if (__ballot(far) != 0) {
                            if (__ballot(far && (int32_t)ab >= (int32_t)visible) != 0) {  // the source reaches into symbols this wave stored since its last wait
                                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                                visible = pos;
                            }
                            if (far) val = dst[(int32_t)ab];
                        }
                        if (islit) val = dd;
                        bool pend = act && !islit && p >= 0;
                        while (__ballot(pend) != 0) {
                            sb[lane] = pend ? (uint32_t)p : (0x80000000u | val);
                            gi_lds_sync();
                            const uint32_t s = sb[pend ? p : lane];
                            gi_lds_sync();
                            if (pend) {
                                if (s & 0x80000000u) {
                                    val = s & 0xffffu;
                                    pend = false;
                                } else
                                    p = (int32_t)s;
                            }
                        }
                        if (act) {
                            ring[(at0 + (uint32_t)lane) & RM] = (OutT)val;
                            dst[at0 + (uint32_t)lane] = (OutT)val;
                        }
                        gi_lds_sync();
                    }
                    pos += total;
                    P += adv;
                    continue;
                }
                // A larger group (long matches) of at most RING / 4 symbols is put together in the ring and stored from there in whole
                // lines (below); one beyond that is stored piece by piece as it is produced.
                const bool big = total > RING / 4u;
                if (on && kind == GI_LIT) {
                    if (big) dst[mpos] = (OutT)lit;
                    ring[mpos & RM] = (OutT)lit;
                }
                if (mm != 0) {
                    gi_lds_sync();
                    // Nothing this group writes lands on a ring slot whose old byte a match with dist + total + 64 <= GI_RING
                    // still reads (the slot of byte p is reused by byte p + GI_RING).
                    const uint32_t from = mpos - dist;
                    // side by side: short, not overlapping, source in the ring and in front of the group's first byte
                    const bool par = is_match && olen <= GI_PAR_MAX && dist >= olen && dist + total + 64u <= RING && (mpos - pos) + olen <= dist;
                    const u64 pm = __ballot(par);
                    if (pm != 0) {
                        for (uint32_t i = 0; i < GI_PAR_MAX; i++) {
                            const bool go = par && i < olen;
                            if (__ballot(go) == 0) break;
                            if (go) {
                                const OutT c = ring[(from + i) & RM];
                                if (big) dst[mpos + i] = c;
                                ring[(mpos + i) & RM] = c;
                            }
                        }
                        gi_lds_sync();
                    }
                    for (u64 sm = mm & ~pm; sm != 0; sm &= sm - 1) {  // the others in order, all lanes on one match
                        const int ml = __builtin_ctzll(sm);
                        const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)olen, ml), dd = (uint32_t)__builtin_amdgcn_readlane((int)dist, ml);
                        const uint32_t mp = (uint32_t)__builtin_amdgcn_readlane((int)mpos, ml), fr = mp - dd;
                        if (dd + total + 64u <= RING) {
                            if (dd == 1) {  // a run of one byte
                                const OutT c = ring[fr & RM];
                                for (uint32_t i = (uint32_t)lane; i < len; i += 64) {
                                    if (big) dst[mp + i] = c;
                                    ring[(mp + i) & RM] = c;
                                }
                            } else {
                                for (uint32_t i = (uint32_t)lane; i < len; i += 64) {
                                    const OutT c = ring[(fr + (dd >= len ? i : i % dd)) & RM];
                                    if (big) dst[mp + i] = c;
                                    ring[(mp + i) & RM] = c;
                                }
                            }
                            gi_lds_sync();
                        } else {
                            // (a small group: dd > RING - 64 - RING / 4 > the group's length + 258 -- the source ends in front of `pos`,
                            // in symbols that were stored by the groups before)
                            if ((int32_t)(fr + (len < dd ? len : dd)) > (int32_t)visible) {  // the source reaches into bytes this wave stored since its last wait
                                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                                visible = big ? mp : pos;
                            }
                            for (uint32_t i = (uint32_t)lane; i < len; i += 64) {
                                const OutT c = dst[(int32_t)(fr + (dd >= len ? i : i % dd))];
                                if (big) dst[mp + i] = c;
                                ring[(mp + i) & RM] = c;
                            }
                            gi_lds_sync();
                        }
                    }
                }
                if (!big) {
                    gi_lds_sync();
                    for (uint32_t i = (uint32_t)lane; i < total; i += 64) dst[pos + i] = ring[(pos + i) & RM];
                }
                if (total > RING) {
                    // A group of more than RING symbols (a handful of maximal matches in 64 bits of input: runs) wrote some ring slots
                    // twice, and not in the order of the positions -- the literals go first, the matches after them: a literal
                    // behind 1 Ki symbols of run lost its slot to the run (found by tools/gunzip_fuzz.py: the newline between a
                    // run of 'A' and the first FASTQ record, copied 216 symbols later).  The text itself is right (every copy of
                    // such a group reads and writes memory): the ring is filled again from its last RING symbols.
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    const uint32_t from_p = pos + total - RING;
                    for (uint32_t i = (uint32_t)lane; i < RING; i += 64) ring[(from_p + i) & RM] = dst[(int32_t)(from_p + i)];
                    gi_lds_sync();
                    visible = pos + total;
                }
                pos += total;
                P += adv;
            }
            if (err != GI_OK) break;
            b.seek_bit(P, lane);
            continue;  // the next block
        }
        // ---- one symbol at a time (a code that did not fit the tables; GS_INFLATE_FORCE_SLOW).  Literals wait in `pbyte` (lane j:
        // the j-th pending byte) and are stored 62..64 at a time; whether they fit the member's announced size is checked when
        // they are stored (flush), not per literal.
        const uint32_t tlimit = in_len / 4u + 4u;  // dwords the reader may take before the stream has run off its payload
        for (;;) {
            if (b.bc <= 32) {
                b.refill(lane);
                if (b.taken > tlimit) {
                    err = GI_E_INPUT;
                    break;
                }
            }
            const uint32_t e = lfast ? gi_lookup(w.ltab, GI_LROOT, b) : gi_slow(w.lcount, w.lwork, 0, b);
            const uint32_t kind = e >> 16;
            if ((e & 15u) == 0) {
                err = GI_E_CODE;
                break;
            }
            b.drop((int)(e & 15u));
            if (kind == GI_LIT) {
                if ((uint32_t)lane == npend) pbyte = (e >> 4) & 0xffu;
                npend++;
                pos++;
                if (npend >= 63) {
                    if (pos > cap) {
                        err = GI_E_OVERRUN;
                        break;
                    }
                    flush();
                }
                continue;
            }
            if (kind == GI_EOB) break;
            if (kind != GI_LEN) {
                err = GI_E_CODE;
                break;
            }
            uint32_t lbase, lextra, dbase, dextra;
            gi_len_sym((e >> 4) & 31u, lbase, lextra);
            const uint32_t len = lbase + b.get((int)lextra);
            b.refill(lane);
            const uint32_t d = dfast ? gi_lookup(w.dtab, GI_DROOT, b) : gi_slow(w.dcount, w.dwork, 1, b);
            if ((d >> 16) != GI_DIST || (d & 15u) == 0) {
                err = GI_E_CODE;
                break;
            }
            b.drop((int)(d & 15u));
            gi_dist_sym((d >> 4) & 31u, dbase, dextra);
            const uint32_t dist = dbase + b.get((int)dextra);
            if (dist > pos + BACK) {
                err = GI_E_DIST;
                break;
            }
            if (pos + len > cap) {
                err = GI_E_OVERRUN;
                break;
            }
            flush();
            const uint32_t from = pos - dist;
            if (dist <= RING - 64u) {
                // the source lies in the ring (every source byte is in front of `pos`, and a byte is overwritten GI_RING
                // bytes later: not by this copy).  LDS operations of a wave execute in order: the barrier is for the compiler.
                gi_lds_sync();
                if (dist >= len) {  // the usual case: source and destination do not overlap
                    for (uint32_t i = (uint32_t)lane; i < len; i += 64) {
                        const OutT c = ring[(from + i) & RM];
                        dst[pos + i] = c;
                        ring[(pos + i) & RM] = c;
                    }
                } else if (dist == 1) {  // a run of one byte
                    const OutT c = ring[from & RM];
                    for (uint32_t i = (uint32_t)lane; i < len; i += 64) {
                        dst[pos + i] = c;
                        ring[(pos + i) & RM] = c;
                    }
                } else {  // the pattern of `dist` bytes repeats
                    for (uint32_t i = (uint32_t)lane; i < len; i += 64) {
                        const OutT c = ring[(from + i % dist) & RM];
                        dst[pos + i] = c;
                        ring[(pos + i) & RM] = c;
                    }
                }
                gi_lds_sync();
            } else {
                if ((int32_t)(from + (len < dist ? len : dist)) > (int32_t)visible) {  // the source reaches into bytes this wave stored since its last wait
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    visible = pos;
                }
                // (dist > GI_RING - 64 >= len is not guaranteed for a small ring: keep the general form)
                for (uint32_t i = (uint32_t)lane; i < len; i += 64) {
                    const OutT c = dst[(int32_t)(from + (dist >= len ? i : i % dist))];
                    dst[pos + i] = c;
                    ring[(pos + i) & RM] = c;
                }
            }
            pos += len;
        }
    }
    if (err == GI_OK && pos > cap) err = GI_E_OVERRUN;
    if (err == GI_OK) flush();
    if (MARK && err == GI_OK && final_inside) err = GI_FINAL;
    if (MARK && err == GI_OK && !to_final && b.consumed() != stop_bit) err = GI_E_SYNC;
    *produced = pos;
    return err;
}

__global__ __launch_bounds__(64 * GI_WAVES) __attribute__((amdgpu_waves_per_eu(5, 5))) void gi_inflate_kernel(const uint8_t *comp, const GiBlock *blocks, int64_t n_blocks, uint8_t *out,
                                                                    int32_t *status, int force_slow, unsigned long long *next_member) {
    __shared__ GiWave s_w[GI_WAVES];
    __shared__ uint8_t s_ring[GI_WAVES][GI_RING];  // text byte p of a wave's member at [p % GI_RING]
    __shared__ uint32_t s_crc[4 * 256];
    for (int i = (int)threadIdx.x; i < 4 * 256; i += 64 * GI_WAVES) s_crc[i] = gi_crc_table[i];
    __syncthreads();
    const int lane = gi_lane();
    const int wib = (int)gi_uni(threadIdx.x >> 6);
    GiWave &w = s_w[wib];
    // the waves draw members from a shared counter: a member takes 5 .. 15 ms of a wave, a fixed assignment would leave the waves
    // with one member fewer idle for that long
    for (;;) {
        __builtin_amdgcn_wave_barrier();
        unsigned long long take = 0;
        if (lane == 0) take = atomicAdd(next_member, 1ULL);
        const int64_t bi = (int64_t)(((u64)gi_uni((uint32_t)(take >> 32)) << 32) | gi_uni((uint32_t)take));
        if (bi >= n_blocks) break;
        const GiBlock blk = blocks[bi];
        uint8_t *const dst = out + blk.out_off;
        const uint32_t cap = blk.out_len;
        GiBits b;
        b.start(comp + blk.in_off, blk.in_len, lane);
        int err = GI_OK;
        uint32_t pos = 0;
        err = gi_decode_blocks<false>(w, s_ring[wib], b, blk.in_len, 0, true, dst, cap, force_slow, lane, &pos);
        if (err == GI_OK && pos != cap) err = GI_E_SIZE;
        if (err == GI_OK && b.consumed() > (u64)blk.in_len * 8u) err = GI_E_INPUT;
        if (err == GI_OK && cap > 0) {  // CRC-32 of the text: 64 equal slices, then crc = crc_0 * x^(8 (n - s)) + crc_1 * x^(8 (n - 2s)) + ...
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const uint32_t slice = ((cap + 63u) / 64u + 15u) & ~15u;
            const uint32_t lo = (uint32_t)lane * slice, hi = lo + slice < cap ? lo + slice : cap;
            uint32_t c = 0;  // (raw register: the pre / post inversion is applied once, on the combined value)
            if (lane == 0) c = 0xffffffffu;
            uint32_t i = lo;
            typedef uint32_t __attribute__((ext_vector_type(4), aligned(1))) u32x4_any;  // (a member's text starts at any byte)
            for (; i + 64u <= hi && lo < cap; i += 64) {  // a line's four loads together (gi_crc_kernel: a line asked of L2 once, not four times)
                u32x4_any v4[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v4[u] = *reinterpret_cast<const u32x4_any *>(dst + i + 16u * (uint32_t)u);
#pragma unroll
                for (int u = 0; u < 4; u++) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        c ^= v4[u][q];
                        c = s_crc[768 + (c & 0xffu)] ^ s_crc[512 + ((c >> 8) & 0xffu)] ^ s_crc[256 + ((c >> 16) & 0xffu)] ^ s_crc[c >> 24];
                    }
                }
            }
            for (; i + 16u <= hi && lo < cap; i += 16) {  // sixteen bytes per load; slicing-by-4: four independent table reads per dword
                const u32x4_any v = *reinterpret_cast<const u32x4_any *>(dst + i);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    c ^= v[q];
                    c = s_crc[768 + (c & 0xffu)] ^ s_crc[512 + ((c >> 8) & 0xffu)] ^ s_crc[256 + ((c >> 16) & 0xffu)] ^ s_crc[c >> 24];
                }
            }
            for (; i < hi && lo < cap; i++) c = s_crc[(c ^ dst[i]) & 0xffu] ^ (c >> 8);
            // combine: the register after slice j is R_j; the whole register = sum_j R_j * x^(8 * bytes behind slice j)
            const uint32_t behind = hi < cap ? cap - hi : 0;
            uint32_t part = (lo < cap) ? gi_gf_mul(c, gi_x_pow_8n(behind)) : 0u;
            for (int o = 32; o >= 1; o >>= 1) part ^= (uint32_t)__shfl_xor((int)part, o);
            if ((part ^ 0xffffffffu) != blk.crc) err = GI_E_CRC;
        }
        // (every lane stores the same word.  With `if (lane == 0)` here AND around the atomic at the top of the loop, clang -O2 threads
        // the two tests through the back edge: lane 0 and the other 63 lanes then go around the loop separately, the 63 read
        // their own -- zero -- `take` with readfirstlane and inflate member 0 for ever.  Seen with ROCm 7.2 once the loop body
        // changed; the barrier at the top is there for the same reason.)
        status[bi] = err;
    }
}

// =====================================================================================================================
// A single-member gzip stream (gzip, pigz: what every sequencer pipeline writes) on the device.  Its deflate blocks follow each other
// at arbitrary BIT positions and every match may reach 32 KiB back, so the stream is taken apart the way pugz / rapidgzip do it on
// CPU threads:
//   1. gi_find_kernel: one wave per 8 KiB of the compressed stream finds EVERY bit offset at which a non-final DYNAMIC block starts
//      (and the final one at the stream's end), through three sieves of falling width and rising cost (see the kernel); it runs behind
//      the upload of the compressed bytes, launch by launch;
//   2. gi_segment_kernel: one wave per SEGMENT -- a whole number of blocks, one round of segments per batch -- decodes to 16-bit
//      symbols; what a match takes from in front of the segment is a marker 0x8000 | window position; a segment must END exactly where
//      the next one starts (else that start was a mirage: the segment is decoded again up to the start after it);
//   3. gi_window_prep / gi_win_compose / gi_win_groups / gi_win_apply: the last 32 KiB of every segment, markers replaced through the
//      window of the segment before -- the one sequential dependence, walked in two levels because window maps compose;
//   4. gi_resolve_kernel: every segment's symbols to bytes (markers through the window in front of it), at its place in the text;
//   5. the CRC-32 of the text per 64 KiB tile (gi_crc_kernel), combined on the host, against the member's trailer.
// =====================================================================================================================
struct GiSeg {
    u64 start_bit;     // of the segment's first block, from the first byte of the deflate stream
    u64 stop_bit;      // where the next segment starts (unused for the last one)
    u64 out_off;       // first symbol of the segment in the symbol buffer (GI_WINDOW marker symbols lie in front of it)
    uint32_t out_cap;  // symbols it may produce
    uint32_t to_final; // the last segment: to the stream's final block
};

// the bits from bit `p` of the stream on (at least 57 of them; the buffer has slack behind the stream)
__device__ __forceinline__ u64 gi_bits_at(const uint8_t *in, u64 p) {
    typedef u64 __attribute__((aligned(1))) u64_any;
    return *reinterpret_cast<const u64_any *>(in + (p >> 3)) >> (p & 7u);
}

// The block finder's third test, ONE CANDIDATE PER LANE (the lanes diverge; nothing here is wave-uniform): the code lengths of the
// dynamic header at bit `o` are decoded the canonical way -- the code length code's counts in a register, its symbols in canonical
// order in the lane's 20 bytes of LDS -- and only summed up: true when the repeat codes fit, the literal / length code is complete and
// has an end-of-block code, the distance code is complete (or the one 1-bit code, or none), and -- text_only -- no byte >= 128 has a
// code.  Stricter than gi_dynamic_header in corners no compressor produces (an incomplete literal code of one symbol); what passes is
// checked by gi_dynamic_header, what does not is simply not a segment boundary.  A random candidate costs the whole wave ~100 trips
// of this loop for up to 64 candidates, where gi_dynamic_header took ~30 us for each of them.
__device__ bool gi_header_plausible(const uint8_t *in, uint32_t in_len, u64 o, uint8_t *sorted, bool text_only) {
    const u64 end = (u64)in_len * 8u;
    if (o + 17u + 57u + 64u > end) return false;  // (a block start this close to the end of the stream is of no use as a boundary)
    const uint32_t head = (uint32_t)gi_bits_at(in, o);
    const uint32_t hlit = ((head >> 3) & 31u) + 257u, hdist = ((head >> 8) & 31u) + 1u, hclen = ((head >> 13) & 15u) + 4u;
    u64 pos = o + 17u;
    const u64 cl = gi_bits_at(in, pos) & ((1ULL << (3u * hclen)) - 1ULL);
    pos += 3u * hclen;
    constexpr uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    u64 by_sym = 0;  // 3 bits per symbol 0 .. 18
#pragma unroll
    for (int i = 0; i < 19; i++) by_sym |= ((cl >> (3 * i)) & 7ULL) << (3 * order[i]);
    u64 cnts = 0;    // 8 bits per length 0 .. 7
#pragma unroll
    for (int sidx = 0; sidx < 19; sidx++) cnts += 1ULL << (8u * (uint32_t)((by_sym >> (3 * sidx)) & 7ULL));
    cnts &= ~0xffULL;
    u64 run = cnts * 0x0101010101010101ULL - cnts;  // per length: the codes of all shorter lengths = where its symbols start
#pragma unroll
    for (int sidx = 0; sidx < 19; sidx++) {
        const uint32_t l = (uint32_t)((by_sym >> (3 * sidx)) & 7ULL);
        if (l) {
            sorted[(uint32_t)(run >> (8u * l)) & 31u] = (uint8_t)sidx;
            run += 1ULL << (8u * l);
        }
    }
    const uint32_t total = hlit + hdist;
    uint32_t at = 0, prev = 0, lit_k = 0, dist_k = 0;
    bool eob = false, high = false;
    u64 buf = gi_bits_at(in, pos);
    int have = 57;
    while (at < total) {
        if (have < 14) {
            if (pos + 64u > end) return false;
            buf = gi_bits_at(in, pos);
            have = 57;
        }
        int code = 0, first = 0, index = 0, used = 0;
        uint32_t sym = 0;
#pragma unroll
        for (int len = 1; len <= 7; len++) {
            if (used == 0) {
                code |= (int)((buf >> (len - 1)) & 1u);
                const int cnt = (int)((cnts >> (8 * len)) & 255u);
                if (code - cnt < first) {
                    sym = sorted[index + (code - first)];
                    used = len;
                }
                index += cnt;
                first += cnt;
                first <<= 1;
                code <<= 1;
            }
        }
        if (used == 0) return false;  // (not with a complete code)
        buf >>= used;
        uint32_t rep = 1, val = sym, extra = 0;
        if (sym == 16) {
            if (at == 0) return false;
            rep = 3u + ((uint32_t)buf & 3u);
            val = prev;
            extra = 2;
        } else if (sym == 17) {
            rep = 3u + ((uint32_t)buf & 7u);
            val = 0;
            extra = 3;
        } else if (sym == 18) {
            rep = 11u + ((uint32_t)buf & 127u);
            val = 0;
            extra = 7;
        }
        buf >>= extra;
        have -= used + (int)extra;
        pos += (u64)(used + (int)extra);
        if (at + rep > total) return false;
        if (val) {
            const uint32_t wgt = 32768u >> val;
            const uint32_t nl = at < hlit ? (rep < hlit - at ? rep : hlit - at) : 0u;
            lit_k += nl * wgt;
            dist_k += (rep - nl) * wgt;
            if (nl && at <= 256u && at + nl > 256u) eob = true;
            if (nl && at < 256u && at + nl > 128u) high = true;
        }
        at += rep;
        prev = val;
    }
    return eob && !(text_only && high) && lit_k == 32768u && (dist_k == 32768u || dist_k == 16384u || dist_k == 0u);
}

// The block finder: per chunk of the compressed stream the bit offsets at which a non-final dynamic block starts (the first
// GI_FIND_MAX of them, in stream order; in the stream's last MiB the final block as well), through three sieves of falling width and
// rising cost --
//   1. every offset: the type bits and the two code counts (one offset in nine passes).  The chunk goes through registers in pieces
//      of 512 bytes (lane i holds dwords i and 64 + i, the next piece is on its way); a lane tests the 32 offsets that start in its
//      dword AT ONCE -- the three type bits and "HLIT, HDIST are not 30 or 31" are ~25 bit operations on the dword pair shifted by
//      the fields' places (the first version tested one offset per lane: 22 instructions per 64 offsets instead of per 2 048) --
//      and hands its survivors to a queue in LDS, one per trip;
//   2. 64 queued offsets at a time: the code length code is complete (Kraft sum exactly one: one in 250 of those);
//   3. 64 of those at a time, one per lane: gi_header_plausible (above);
// and what is left -- real block starts, and a mirage per 100 MB -- through gi_dynamic_header by the whole wave.
// (The first version: sieves 1 + 2 in one step for every offset, from memory, and gi_dynamic_header for each of the ~40 survivors
// per chunk, up to the chunk's first block start only: 5.9 ms for 4 096 chunks of 58 KB.  This one reads the whole stream.)
#define GI_FIND_MAX 16
#define GI_FIND_LAUNCHES 64  // finder launches per batch at most (each has a work counter of its own; what is left when they are used up goes into the last one)
__global__ __launch_bounds__(64 * GI_WAVES) __attribute__((amdgpu_waves_per_eu(5, 5))) void gi_find_kernel(const uint8_t *in, uint32_t in_len, uint32_t chunk_bytes,
                                                                                                        int64_t first, int64_t n_chunks, u64 *start_bit, unsigned long long *next_chunk,
                                                                                                        int text_only, int64_t n_real, int64_t fin_first) {
    __shared__ GiWave s_w[GI_WAVES];
    __shared__ uint32_t s_q1[GI_WAVES][256];  // sieve 1's survivors (a ring; bit offsets from the chunk's first bit)
    __shared__ uint32_t s_q2[GI_WAVES][128];  // sieve 2's survivors
    __shared__ uint8_t s_kraft[4096];  // four code lengths of 3 bits -> their Kraft sum in 1 / 128 (255: more than one)
    for (uint32_t i = threadIdx.x; i < 4096u; i += 64 * GI_WAVES) {
        uint32_t sum = 0;
        for (uint32_t f = 0; f < 4; f++) {
            const uint32_t len = (i >> (3 * f)) & 7u;
            sum += len ? (128u >> len) : 0u;
        }
        s_kraft[i] = (uint8_t)(sum < 255u ? sum : 255u);
    }
    __syncthreads();
    const int lane = gi_lane();
    const int wib = (int)gi_uni(threadIdx.x >> 6);
    GiWave &w = s_w[wib];
    uint32_t *const q1 = s_q1[wib], *const q2 = s_q2[wib];
    typedef uint32_t __attribute__((aligned(1))) u32_any;
    for (;;) {
        __builtin_amdgcn_wave_barrier();
        unsigned long long take = 0;
        if (lane == 0) take = atomicAdd(next_chunk, 1ULL);
        const int64_t ci = first + (int64_t)(((u64)gi_uni((uint32_t)(take >> 32)) << 32) | gi_uni((uint32_t)take));
        if (ci >= n_chunks) break;  // (this launch: the chunks first .. n_chunks, `next_chunk` counts from 0)
        // Chunks n_real .. n_chunks are the chunks fin_first .. n_real (the stream's last MiB, when the upload reaches its end) once
        // more, searched for the FINAL block: without it the last segment is the stream's last two blocks, and a batch takes as long
        // as its longest segment.  (A search of its own: twice the candidates in one chunk would make that wave the kernel's last.)
        const bool fin = ci >= n_real;
        const u64 lo_byte = (u64)(fin ? fin_first + (ci - n_real) : ci) * chunk_bytes;
        const u64 lo = lo_byte * 8u;
        const u64 hi_byte = lo_byte + chunk_bytes < (u64)in_len ? lo_byte + chunk_bytes : (u64)in_len;
        const uint32_t n_bits = (uint32_t)(hi_byte - lo_byte) * 8u;  // (offsets are kept relative to `lo`: a chunk is less than 512 MiB)
        uint32_t n1 = 0, h1 = 0, n2 = 0, n_found = 0, grp = 0;      // (wave-uniform; grp: groups of 64 offsets searched so far)
        uint32_t pend = 0, pend_base = 0;  // sieve 1's survivors among this lane's 32 offsets from pend_base + 32 * lane on, not yet queued
        const uint8_t *const p0 = in + lo_byte + 4u * (uint32_t)lane;
        // the piece being searched: dwords 0 .. 63, 64 .. 127 and 128 of it; the one behind it (the buffer is zero for 1 KiB behind the stream)
        uint32_t r0 = 0, r1 = 0, r2 = 0;
        uint32_t x0 = *reinterpret_cast<const u32_any *>(p0), x1 = *reinterpret_cast<const u32_any *>(p0 + 256);
        uint32_t x2 = *reinterpret_cast<const u32_any *>(in + lo_byte + 512);
        for (;;) {
            const bool scanned = grp * 64u >= n_bits && __ballot(pend != 0u) == 0;
            if (n2 >= 64u || (scanned && n1 == 0u && n2 > 0u)) {
                // sieve 3, one candidate per lane; then the whole header for what is left, in offset order
                gi_lds_sync();
                const uint32_t cnt = n2 < 64u ? n2 : 64u;
                const bool act = (uint32_t)lane < cnt;
                const uint32_t rel = q2[act ? lane : 0];
                const uint32_t rest = q2[64u + (uint32_t)lane < n2 ? 64u + (uint32_t)lane : 0u];
                // (a lane's 20 bytes of symbols in canonical order lie in the wave's literal table, which gi_dynamic_header builds anew
                // below: five workgroups of 28.4 KB fit a CU, with 96 VGPRs five waves a SIMD)
                const bool good = act && gi_header_plausible(in, in_len, lo + rel, reinterpret_cast<uint8_t *>(w.ltab) + 20 * lane, text_only != 0);
                for (u64 cand = __ballot(good); cand != 0; cand &= cand - 1) {
                    const u64 oc = lo + (u64)(uint32_t)__builtin_amdgcn_readlane((int)rel, __builtin_ctzll(cand));
                    GiBits b;
                    b.in = in;
                    b.in_len = in_len;
                    b.seek_bit(oc + 3, lane);
                    bool lf, df;
                    if (gi_dynamic_header(w, b, lane, &lf, &df) == GI_OK && lf && df) {
                        // FASTQ / FASTA are text: a block whose code gives a length to a byte >= 128 is not taken for a start (a bit pattern
                        // that parses as a header by chance -- about one per 100 MB -- does so with all but certainty; a real block
                        // with such bytes is then simply not a segment boundary)
                        bool high = false;
                        if (text_only)
                            for (int sidx = 128 + lane; sidx < 256; sidx += 64) high |= w.lens[sidx] != 0;
                        if (__ballot(high) == 0) {
                            if (n_found < GI_FIND_MAX) start_bit[ci * GI_FIND_MAX + n_found] = oc;  // (every lane the same word)
                            n_found++;
                        }
                    }
                }
                gi_lds_sync();
                if (64u + (uint32_t)lane < n2) q2[lane] = rest;  // (fewer than 64 are left: one lane each)
                n2 -= cnt;
                continue;
            }
            if (n1 >= 64u || (scanned && n1 > 0u)) {
                // sieve 2: the code length code is complete
                gi_lds_sync();
                const uint32_t cnt = n1 < 64u ? n1 : 64u;
                const bool act = (uint32_t)lane < cnt;
                const uint32_t rel = q1[(h1 + (uint32_t)lane) & 255u];
                const u64 o = lo + (act ? rel : 0u);
                const uint8_t *p = in + ((o >> 5) << 2);
                const uint32_t d0 = *reinterpret_cast<const u32_any *>(p), d1 = *reinterpret_cast<const u32_any *>(p + 4);
                const uint32_t d2 = *reinterpret_cast<const u32_any *>(p + 8), d3 = *reinterpret_cast<const u32_any *>(p + 12);
                const uint32_t sh2 = (uint32_t)o & 31u;
                const uint32_t w0 = __builtin_amdgcn_alignbit(d1, d0, sh2), w1 = __builtin_amdgcn_alignbit(d2, d1, sh2), w2 = __builtin_amdgcn_alignbit(d3, d2, sh2);
                const uint32_t hclen = ((w0 >> 13) & 15u) + 4u;
                const uint32_t na = hclen < 15u ? hclen : 15u, nc = hclen - na;  // the first 15 lengths, and lengths 15 .. 18
                const u64 a = (((u64)w1 << 32 | w0) >> 17) & ((1ULL << (3u * na)) - 1ULL);
                const uint32_t c = (uint32_t)(((u64)w2 << 32 | w1) >> 30) & ((1u << (3u * nc)) - 1u);
                // (the sum by table, four lengths at a time: 19 x shift / mask / compare / select were most of this sieve)
                const uint32_t kraft = (uint32_t)s_kraft[(uint32_t)a & 0xfffu] + s_kraft[(uint32_t)(a >> 12) & 0xfffu] + s_kraft[(uint32_t)(a >> 24) & 0xfffu] +
                                       s_kraft[(uint32_t)(a >> 36)] + s_kraft[c];
                const bool ok = act && kraft == 128u;
                const u64 m = __ballot(ok);
                if (ok) q2[n2 + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = rel;
                n2 += (uint32_t)__builtin_popcountll(m);  // (fewer than 64 before: the queue holds 128)
                h1 += cnt;
                n1 -= cnt;
                continue;
            }
            if (__ballot(pend != 0u) != 0) {
                // sieve 1's survivors to the queue, one per lane and trip (a lane's 32 offsets hold three or four of them; at most 64
                // a trip, and the queue is drained above before it holds 64: 63 + 64 of its 256 places)
                const bool has = pend != 0u;
                const u64 m = __ballot(has);
                const uint32_t rel = pend_base + 32u * (uint32_t)lane + (uint32_t)__builtin_ctz(has ? pend : 1u);
                if (has) q1[(h1 + n1 + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))) & 255u] = rel;
                pend &= pend - 1u;
                n1 += (uint32_t)__builtin_popcountll(m);
                continue;
            }
            if (scanned) break;
            // sieve 1 over the next 2 048 offsets, 32 to a lane and all of them at once: not final, dynamic, at most 286 / 30 codes
            // as bit operations on (x >> s) for the shifts the header's fields sit at
            if ((grp & 63u) == 0u) {
                r0 = x0;
                r1 = x1;
                r2 = x2;
                const uint32_t next_at = (grp >> 6) * 512u + 512u;
                if (next_at * 8u < n_bits) {  // the piece behind this one, while this one is searched
                    x0 = *reinterpret_cast<const u32_any *>(p0 + next_at);
                    x1 = *reinterpret_cast<const u32_any *>(p0 + next_at + 256);
                    x2 = *reinterpret_cast<const u32_any *>(in + lo_byte + next_at + 512);
                }
            }
            {
                const bool second = (grp & 32u) != 0u;
                const uint32_t cur = second ? r1 : r0;                                                 // this lane's 32 offsets start in its dword
                const uint32_t edge = (uint32_t)__builtin_amdgcn_readlane((int)(second ? r2 : r1), 0);  // the dword behind lane 63's
                uint32_t nxt = (uint32_t)__shfl_down((int)cur, 1);
                if (lane == 63) nxt = edge;
#define GI_SH(sft) __builtin_amdgcn_alignbit(nxt, cur, sft)
                uint32_t m = (fin ? cur : ~cur) & ~GI_SH(1) & GI_SH(2);      // type bits 1 0 1 (final) / 0 0 1, lowest first
                m &= ~(GI_SH(4) & GI_SH(5) & GI_SH(6) & GI_SH(7));          // HLIT <= 29: not 1111x
                m &= ~(GI_SH(9) & GI_SH(10) & GI_SH(11) & GI_SH(12));       // HDIST <= 29
#undef GI_SH
                pend_base = grp * 64u;
                const uint32_t mine = pend_base + 32u * (uint32_t)lane;  // offsets at and beyond n_bits are not the chunk's
                if (mine >= n_bits)
                    m = 0;
                else if (n_bits - mine < 32u)
                    m &= (1u << (n_bits - mine)) - 1u;
                pend = m;
                grp += 32u;
            }
        }
    }
}

// (80 VGPRs and 6.6 KB of LDS per wave: six waves per SIMD, 24 per CU -- the LDS of a CU is handed out in pieces of 1 280 bytes:
// a workgroup of 32 064 bytes fits four times, not five, which cost a third of the kernel's speed before it was noticed)
__global__ __launch_bounds__(64 * GI_WAVES) __attribute__((amdgpu_waves_per_eu(6, 6))) void gi_segment_kernel(const uint8_t *in, uint32_t in_len, const GiSeg *segs, int64_t n_segs,
                                                                                                           uint16_t *sym, int32_t *status, uint32_t *out_len,
                                                                                                           u64 *end_bit, unsigned long long *next_seg) {
    __shared__ GiWave s_w[GI_WAVES];
    __shared__ uint16_t s_ring[GI_WAVES][GI_RING_SYM];
    const int lane = gi_lane();
    const int wib = (int)gi_uni(threadIdx.x >> 6);
    GiWave &w = s_w[wib];
    for (;;) {
        __builtin_amdgcn_wave_barrier();
        unsigned long long take = 0;
        if (lane == 0) take = atomicAdd(next_seg, 1ULL);
        const int64_t si = (int64_t)(((u64)gi_uni((uint32_t)(take >> 32)) << 32) | gi_uni((uint32_t)take));
        if (si >= n_segs) break;
        const GiSeg seg = segs[si];
        uint16_t *const dst = sym + seg.out_off;
        // the unknown window in front of the segment: markers, in memory and in the ring
        for (uint32_t j = (uint32_t)lane; j < GI_WINDOW; j += 64) dst[(int32_t)j - (int32_t)GI_WINDOW] = (uint16_t)(0x8000u | j);
        for (uint32_t j = (uint32_t)lane; j < GI_RING_SYM; j += 64) s_ring[wib][j] = (uint16_t)(0x8000u | (GI_WINDOW - GI_RING_SYM + j));
        gi_lds_sync();
        GiBits b;
        b.in = in;
        b.in_len = in_len;
        b.seek_bit(seg.start_bit, lane);
        uint32_t pos = 0;
        const int err = gi_decode_blocks<true>(w, s_ring[wib], b, in_len, seg.stop_bit, seg.to_final != 0, dst, seg.out_cap, 0, lane, &pos);
        status[si] = err;
        out_len[si] = pos;
        end_bit[si] = b.consumed();
    }
}

// win[i]: the last GI_WINDOW bytes of the text up to the end of segment i.  gi_window_prep_kernel copies every segment's last GI_WINDOW
// symbols into an array of their own (side by side); the markers in them are then replaced through the window before -- the only
// sequential dependence of the whole decoder.  (Repetitive text keeps its markers to the end of a segment -- every quality line is a
// copy of the one before --, so every slot has to be looked at; one workgroup walking all segments, the window in LDS, takes 2.2 us
// per segment: 9 ms for 4 096.)
__global__ __launch_bounds__(256) void gi_window_prep_kernel(const uint16_t *sym, const GiSeg *segs, const uint32_t *out_len, int64_t n_segs, uint16_t *win16) {
    const int64_t i = blockIdx.x;
    if (i >= n_segs) return;
    const uint16_t *dst = sym + segs[i].out_off;
    const int64_t n = (int64_t)out_len[i];
    uint16_t *cur = win16 + (size_t)i * GI_WINDOW;
    for (uint32_t j = threadIdx.x; j < GI_WINDOW; j += 256) cur[j] = dst[n - (int64_t)GI_WINDOW + (int64_t)j];  // (in front of the segment: its marker prefix)
}

// The walk in two levels.  A segment's window symbols are a MAP from the window before it (slot -> a byte, or "slot p of the window
// before"), and maps compose: gi_win_compose_kernel walks the segments of a GROUP in order (the groups side by side, one workgroup
// each, the composed map in LDS) and leaves every segment's map relative to the window before its group; gi_win_groups_kernel walks
// the groups' last segments in order (one workgroup, the window bytes in LDS); gi_win_apply_kernel fills in all other windows side by
// side from the end of the group before.  4 096 segments: 32 + 128 sequential steps instead of 4 096.
// Thread t owns the slots 8 t + 8192 q .. + 8 (q < 4); the LDS copy is updated in place between two barriers.
__global__ __launch_bounds__(1024) void gi_win_compose_kernel(uint16_t *win16, int64_t n_segs, int64_t per_group) {
    __shared__ uint16_t s_c[GI_WINDOW];
    const uint32_t t = threadIdx.x;
    const int64_t a = (int64_t)blockIdx.x * per_group, b = a + per_group < n_segs ? a + per_group : n_segs;
    if (a >= b) return;
    gs_u16x8 nxt[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        nxt[q] = *reinterpret_cast<const gs_u16x8 *>(win16 + (size_t)a * GI_WINDOW + 8u * t + 8192u * (uint32_t)q);
        *reinterpret_cast<gs_u16x8 *>(&s_c[8u * t + 8192u * (uint32_t)q]) = nxt[q];
    }
    if (a + 1 < b) {
#pragma unroll
        for (int q = 0; q < 4; q++) nxt[q] = *reinterpret_cast<const gs_u16x8 *>(win16 + (size_t)(a + 1) * GI_WINDOW + 8u * t + 8192u * (uint32_t)q);
    }
    __syncthreads();
    for (int64_t i = a + 1; i < b; i++) {
        gs_u16x8 cur[4];
#pragma unroll
        for (int q = 0; q < 4; q++) cur[q] = nxt[q];
        if (i + 1 < b) {
#pragma unroll
            for (int q = 0; q < 4; q++) nxt[q] = *reinterpret_cast<const gs_u16x8 *>(win16 + (size_t)(i + 1) * GI_WINDOW + 8u * t + 8192u * (uint32_t)q);
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint16_t v = cur[q].v[e];
                cur[q].v[e] = v < 0x8000u ? v : s_c[v & 0x7fffu];
            }
        __syncthreads();  // (every read of the map before is through)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            *reinterpret_cast<gs_u16x8 *>(&s_c[8u * t + 8192u * (uint32_t)q]) = cur[q];
            *reinterpret_cast<gs_u16x8 *>(win16 + (size_t)i * GI_WINDOW + 8u * t + 8192u * (uint32_t)q) = cur[q];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void gi_win_groups_kernel(const uint16_t *win16, int64_t n_segs, int64_t per_group, uint8_t *win, const uint8_t *win0) {
    __shared__ uint8_t s_w[GI_WINDOW];
    const uint32_t t = threadIdx.x;
    const int64_t n_groups = (n_segs + per_group - 1) / per_group;
    // win0: the text in front of the first segment (a later batch of the stream), or nullptr (the stream starts here: zeros)
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t slot = 8u * t + 8192u * (uint32_t)q;
        *reinterpret_cast<uint64_t *>(&s_w[slot]) = win0 ? *reinterpret_cast<const uint64_t *>(win0 + slot) : 0ULL;
    }
    __syncthreads();
    for (int64_t g = 0; g < n_groups; g++) {
        const int64_t last = ((g + 1) * per_group < n_segs ? (g + 1) * per_group : n_segs) - 1;
        uint64_t pack[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const gs_u16x8 c = *reinterpret_cast<const gs_u16x8 *>(win16 + (size_t)last * GI_WINDOW + 8u * t + 8192u * (uint32_t)q);
            uint8_t bytes[8];
#pragma unroll
            for (int e = 0; e < 8; e++) bytes[e] = c.v[e] < 0x8000u ? (uint8_t)c.v[e] : s_w[c.v[e] & 0x7fffu];
            memcpy(&pack[q], bytes, 8);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t slot = 8u * t + 8192u * (uint32_t)q;
            *reinterpret_cast<uint64_t *>(&s_w[slot]) = pack[q];
            *reinterpret_cast<uint64_t *>(win + (size_t)last * GI_WINDOW + slot) = pack[q];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void gi_win_apply_kernel(const uint16_t *win16, int64_t n_segs, int64_t per_group, uint8_t *win, const uint8_t *win0) {
    const int64_t i = blockIdx.x;
    if (i >= n_segs) return;
    const int64_t g = i / per_group;
    const int64_t last = ((g + 1) * per_group < n_segs ? (g + 1) * per_group : n_segs) - 1;
    if (i == last) return;  // (gi_win_groups_kernel has written it)
    const uint8_t *prev = g ? win + (size_t)(g * per_group - 1) * GI_WINDOW : win0;
    const uint16_t *c = win16 + (size_t)i * GI_WINDOW;
    uint8_t *out = win + (size_t)i * GI_WINDOW;
    for (uint32_t j = threadIdx.x; j < GI_WINDOW; j += 256) {
        const uint16_t v = c[j];
        out[j] = v < 0x8000u ? (uint8_t)v : (prev ? prev[v & 0x7fffu] : (uint8_t)0);
    }
}

__global__ __launch_bounds__(256) void gi_resolve_kernel(const uint16_t *sym, const GiSeg *segs, const uint32_t *out_len, const u64 *text_off, int64_t n_segs,
                                                         const uint8_t *win, const uint8_t *win0, uint8_t *text) {
    for (int64_t i = blockIdx.y; i < n_segs; i += gridDim.y) {
        const uint16_t *dst = sym + segs[i].out_off;
        const uint32_t n = out_len[i];
        const uint8_t *prev = i ? win + (size_t)(i - 1) * GI_WINDOW : win0;
        uint8_t *out = text + text_off[i];
        // eight symbols to a thread (a segment's symbols start on a 16-byte boundary: its room is a multiple of eight), then the rest
        typedef u64 __attribute__((aligned(1))) u64_any;
        const uint32_t n8 = n / 8u;
        for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n8; j += gridDim.x * blockDim.x) {
            const gs_u16x8 v = *reinterpret_cast<const gs_u16x8 *>(dst + 8u * (size_t)j);
            u64 o = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint16_t c = v.v[q];
                const uint8_t b = c < 0x8000u ? (uint8_t)c : (prev ? prev[c & 0x7fffu] : (uint8_t)0);
                o |= (u64)b << (8 * q);
            }
            *reinterpret_cast<u64_any *>(out + 8u * (size_t)j) = o;
        }
        for (uint32_t j = 8u * n8 + blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
            const uint16_t v = dst[j];
            out[j] = v < 0x8000u ? (uint8_t)v : (prev ? prev[v & 0x7fffu] : (uint8_t)0);
        }
    }
}

// raw CRC-32 register (no pre / post inversion) over tile t of `tile` bytes: the host combines them
struct GiCrcPow {  // x^(8 * slice * 2^s) mod P for a whole tile's slice (tile / 64 bytes), s = 0 .. 5: computed once on the host
    uint32_t p[6];
};

__global__ __launch_bounds__(256) void gi_crc_kernel(const uint8_t *text, int64_t n, uint32_t tile, uint32_t *crc, GiCrcPow pw) {
    __shared__ uint32_t s_crc[4 * 256];
    for (int i = (int)threadIdx.x; i < 4 * 256; i += 256) s_crc[i] = gi_crc_table[i];
    __syncthreads();
    const int lane = gi_lane();
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t base = t * (int64_t)tile;
    if (base >= n) return;
    const uint32_t cap = (uint32_t)(n - base < (int64_t)tile ? n - base : (int64_t)tile);
    const uint8_t *dst = text + base;
    const uint32_t slice = ((cap + 63u) / 64u + 15u) & ~15u;
    const uint32_t lo = (uint32_t)lane * slice, hi = lo + slice < cap ? lo + slice : cap;
    uint32_t c = 0;
    uint32_t i = lo;
    typedef uint32_t __attribute__((ext_vector_type(4), aligned(1))) u32x4_any;  // (the text starts at any byte)
    // A lane's slice is a stream of cache lines of its own, and the CU's L1 (32 KB for 32 waves x 64 such streams) keeps none of them
    // from one load to the next: with one 16-byte load per trip every line was asked of L2 four times, and the kernel ran at L2's
    // request rate (1.26 GB in 1.25 ms = 63 G requests/s).  Four loads of one line together: the misses behind the first merge.
    for (; i + 64u <= hi && lo < cap; i += 64) {
        u32x4_any v4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v4[u] = *reinterpret_cast<const u32x4_any *>(dst + i + 16u * (uint32_t)u);
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                c ^= v4[u][q];
                c = s_crc[768 + (c & 0xffu)] ^ s_crc[512 + ((c >> 8) & 0xffu)] ^ s_crc[256 + ((c >> 16) & 0xffu)] ^ s_crc[c >> 24];
            }
        }
    }
    for (; i + 16u <= hi && lo < cap; i += 16) {  // (what is left of the slice, sixteen bytes per load)
        const u32x4_any v = *reinterpret_cast<const u32x4_any *>(dst + i);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            c ^= v[q];
            c = s_crc[768 + (c & 0xffu)] ^ s_crc[512 + ((c >> 8) & 0xffu)] ^ s_crc[256 + ((c >> 16) & 0xffu)] ^ s_crc[c >> 24];
        }
    }
    for (; i < hi && lo < cap; i++) c = s_crc[(c ^ dst[i]) & 0xffu] ^ (c >> 8);
    if (cap == tile && slice * 64u == tile) {
        // a whole tile, 64 equal slices: neighbours are joined pairwise, crc(A B) = crc(A) x^(8 |B|) + crc(B), six times (a lane's own
        // x^(8 * bytes behind it) -- a square-and-multiply per lane -- took longer than the slice itself)
        uint32_t part = c;
#pragma unroll
        for (int sidx = 0; sidx < 6; sidx++) {
            const uint32_t other = (uint32_t)__shfl_down((int)part, 1 << sidx);
            if ((lane & ((2 << sidx) - 1)) == 0) part = gi_gf_mul(part, pw.p[sidx]) ^ other;
        }
        if (lane == 0) crc[t] = part;
        return;
    }
    const uint32_t behind = hi < cap ? cap - hi : 0;
    uint32_t part = (lo < cap) ? gi_gf_mul(c, gi_x_pow_8n(behind)) : 0u;
    for (int o = 32; o >= 1; o >>= 1) part ^= (uint32_t)__shfl_xor((int)part, o);
    if (lane == 0) crc[t] = part;
}

// ---- the four-line cut: newlines per 4 KiB tile, then (one workgroup) the offset of the last newline whose count is a multiple of 4
__global__ __launch_bounds__(256) void gi_count_kernel(const uint8_t *text, int64_t n, uint32_t *tile_count) {
    const int64_t tile = blockIdx.x;
    const int64_t at = tile * 4096 + (int64_t)threadIdx.x * 16;
    uint32_t c = 0;
    if (at + 16 <= n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(text + at);
        const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
        for (int q = 0; q < 4; q++) {
            const uint32_t x = wds[q] ^ 0x0a0a0a0au;
            c += __popc(((x - 0x01010101u) & ~x & 0x80808080u));
        }
    } else {
        for (int64_t i = at; i < n && i < at + 16; i++) c += text[i] == '\n';
    }
    __shared__ uint32_t s[256];
    s[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_count[tile] = s[0];
}

// out[0] = newlines in [0, n), out[1] = bytes up to and including the newline number (out[0] & ~3) (0 if that is 0)
// inclusive prefix sum over the 1024 threads of the block (s: 1024 words of LDS; returns this thread's sum, *total the block's)
__device__ u64 gi_block_scan(u64 v, u64 *s, int t, u64 *total) {
    s[t] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const u64 add = t >= o ? s[t - o] : 0;
        __syncthreads();
        s[t] += add;
        __syncthreads();
    }
    const u64 mine = s[t];
    *total = s[1023];
    __syncthreads();
    return mine;
}

// out[0]: newlines in text[0 .. n); out[1]: the bytes up to and including the last newline whose number is a multiple of four (the
// whole four-line records).  One block: the tiles' counts summed per thread and scanned, the thread whose range holds the target
// newline opened up over the block (its tiles, then the 4 096 bytes of the one tile), a scan each time.  (The first version walked
// the last two levels on one thread, a dependent load per tile and per byte: 0.46 ms for 512 MiB of text.)
__global__ __launch_bounds__(1024) void gi_cut_kernel(const uint8_t *text, int64_t n, const uint32_t *tile_count, int64_t n_tiles, u64 *out) {
    __shared__ u64 s_scan[1024];
    __shared__ u64 s_lo, s_before, s_tile;
    const int t = (int)threadIdx.x;
    const int64_t per = ((n_tiles + 1023) / 1024 + 15) & ~(int64_t)15;  // (a multiple of 16 counts: a thread's run is read a cache line at a time, as in gs_text_scan_kernel)
    const int64_t lo = (int64_t)t * per < n_tiles ? (int64_t)t * per : n_tiles, hi = lo + per < n_tiles ? lo + per : n_tiles;
    u64 mine = 0;
    int64_t i0 = lo;
    for (; i0 + 16 <= hi; i0 += 16) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = reinterpret_cast<const uint4 *>(tile_count + i0)[u];
#pragma unroll
        for (int u = 0; u < 4; u++) mine += (u64)v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (int64_t i = i0; i < hi; i++) mine += tile_count[i];
    u64 total = 0;
    const u64 incl = gi_block_scan(mine, s_scan, t, &total);
    const u64 target = total & ~3ULL;  // the newline with this number (1-based) ends the last whole record
    if (target == 0) {
        if (t == 0) {
            out[0] = total;
            out[1] = 0;
        }
        return;
    }
    if (incl - mine < target && target <= incl) {  // (exactly one thread)
        s_lo = (u64)lo;
        s_before = incl - mine;
    }
    __syncthreads();
    // that thread's tiles, 1024 at a time
    int64_t tile = -1;
    u64 before = s_before;
    const int64_t first = (int64_t)s_lo, behind = first + per < n_tiles ? first + per : n_tiles;
    for (int64_t base = first; tile < 0 && base < behind; base += 1024) {
        const int64_t i = base + t;
        const u64 c = i < behind ? tile_count[i] : 0;
        u64 sum = 0;
        const u64 inc = gi_block_scan(c, s_scan, t, &sum);
        if (before + inc - c < target && target <= before + inc) {
            s_tile = (u64)i;
            s_before = before + inc - c;
        }
        __syncthreads();
        if (before + sum >= target) tile = (int64_t)s_tile;
        before += sum;
    }
    if (tile < 0) return;  // (not reached: the target lies in that thread's tiles)
    // the tile's bytes, four to a thread
    before = s_before;
    const int64_t at = tile * 4096 + 4 * (int64_t)t;
    uint32_t c = 0;
    for (int q = 0; q < 4; q++) c += (at + q < n && text[at + q] == '\n') ? 1u : 0u;
    u64 sum = 0;
    const u64 inc = gi_block_scan(c, s_scan, t, &sum);
    if (before + inc - c < target && target <= before + inc) {
        u64 seen = before + inc - c;
        for (int q = 0; q < 4; q++)
            if (at + q < n && text[at + q] == '\n' && ++seen == target) out[1] = (u64)(at + q) + 1;
        out[0] = total;
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
static thread_local std::string gi_err;
extern "C" const char *gs_inflate_last_error(void) { return gi_err.c_str(); }
static int gi_fail(int code, const std::string &m) {
    gi_err = m;
    return code;
}
#define GI_TRY(x)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) return gi_fail(e_ == hipErrorOutOfMemory ? GS_E_NOMEM : GS_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); \
    } while (0)

// The occupancy the two decoders are launched for is an LDS budget: a CU's 160 KB are handed out in pieces of 1 280 bytes, so a workgroup
// that grows by a few bytes across a multiple of that loses a whole workgroup per CU (gi_segment_kernel 10.3 -> 15.5 ms when it did).
constexpr size_t GI_LDS_PIECE = 1280, GI_LDS_CU = 160 * 1024;
constexpr size_t gi_lds_pieces(size_t bytes) { return (bytes + GI_LDS_PIECE - 1) / GI_LDS_PIECE * GI_LDS_PIECE; }
static_assert(6 * gi_lds_pieces(GI_WAVES * (sizeof(GiWave) + GI_RING_SYM * sizeof(uint16_t))) <= GI_LDS_CU, "gi_segment_kernel: six workgroups no longer fit a CU's LDS");
static_assert(5 * gi_lds_pieces(GI_WAVES * (sizeof(GiWave) + GI_RING) + 4 * 256 * sizeof(uint32_t)) <= GI_LDS_CU, "gi_inflate_kernel: five workgroups no longer fit a CU's LDS");

static int gi_wgs_per_cu() {  // workgroups of four waves per CU (the LDS of a CU holds five of 31.4 KB -- it is handed out in pieces of 1 280 bytes --, 96 VGPRs)
    int v = 5;
    if (const char *e = getenv("GS_INFLATE_WGS")) v = std::max(1, std::min(8, atoi(e)));
    return v;
}

static int gi_seg_wgs_per_cu() {  // ... of gi_segment_kernel: six (26.3 KB of LDS per workgroup in pieces of 1 280 bytes, 80 VGPRs)
    int v = 6;
    if (const char *e = getenv("GS_GUNZIP_WGS")) v = std::max(1, std::min(8, atoi(e)));
    return v;
}

static int gi_force_slow() {  // GS_INFLATE_FORCE_SLOW=1: every Huffman code through the canonical decoder (tests)
    const char *e = getenv("GS_INFLATE_FORCE_SLOW");
    return e && atoi(e) != 0;
}

static int gi_upload_crc_table() {
    static std::atomic<bool> done[64];  // (zero-initialised; two threads may both upload the same table, neither reads a torn flag)
    int dev = 0;
    GI_TRY(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && done[dev].load(std::memory_order_acquire)) return GS_OK;
    uint32_t t[4 * 256];
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1u) ? 0xedb88320u : 0u);
        t[i] = c;
    }
    for (int q = 1; q < 4; q++)  // t[q][i]: the register after byte i and q zero bytes
        for (uint32_t i = 0; i < 256; i++) t[256 * q + i] = (t[256 * (q - 1) + i] >> 8) ^ t[t[256 * (q - 1) + i] & 0xffu];
    GI_TRY(hipMemcpyToSymbol(HIP_SYMBOL(gi_crc_table), t, sizeof(t)));
    if (dev >= 0 && dev < 64) done[dev].store(true, std::memory_order_release);
    return GS_OK;
}

struct gs_inflater {
    int device = 0;
    int n_cu = 256;
    hipStream_t stream = nullptr;
    uint8_t *d_comp[2] = {nullptr, nullptr};
    size_t comp_cap[2] = {0, 0};
    uint8_t *h_comp[2] = {nullptr, nullptr};  // page-locked staging: the file is a page-cache mapping, which the runtime copies at ~0.1 GB/s
    size_t h_comp_cap[2] = {0, 0};
    int64_t comp_lo[2] = {-1, -1}, comp_hi[2] = {-1, -1};  // the range of the file each staging buffer holds
    hipEvent_t comp_ready[2] = {nullptr, nullptr};
    uint8_t *d_text[2] = {nullptr, nullptr};
    size_t text_cap[2] = {0, 0};
    int cur = 0;            // text buffer of the current call
    const uint8_t *last_text = nullptr;  // what the last feed returned (gs_inflater_fetch)
    int64_t last_bytes = 0;
    int64_t tail = 0;       // bytes carried into d_text[cur] by the previous call
    GiBlock *d_blocks = nullptr;
    size_t blocks_cap = 0;
    int32_t *d_status = nullptr;
    size_t status_cap = 0;
    uint32_t *d_tiles = nullptr;
    size_t tiles_cap = 0;
    u64 *d_cut = nullptr;
    // pinned host mirrors
    GiBlock *h_blocks = nullptr;
    int32_t *h_status = nullptr;
    size_t h_cap = 0;
    u64 *h_cut = nullptr;
    double kernel_ms = 0;
    int64_t launches = 0;
};

extern "C" int gs_inflater_create(gs_inflater **out, int device) {
    if (!out) return gi_fail(GS_E_INVALID, "NULL argument");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return gi_fail(GS_E_NODEVICE, "no usable gfx950 device");
    if (device < 0 || device >= n) return gi_fail(GS_E_INVALID, "bad device");
    GI_TRY(hipSetDevice(device));
    int rc = gi_upload_crc_table();
    if (rc) return rc;
    gs_inflater *g = new gs_inflater();
    g->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) g->n_cu = prop.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&g->comp_ready[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void **)&g->d_cut, 4 * sizeof(u64));  // [0..1] the cut, [2] the member queue's cursor
    if (e == hipSuccess) e = hipHostMalloc((void **)&g->h_cut, 2 * sizeof(u64));
    if (e != hipSuccess) {
        delete g;
        return gi_fail(GS_E_HIP, std::string("inflater: ") + hipGetErrorString(e));
    }
    *out = g;
    return GS_OK;
}

// forget the carried tail and the staged ranges: the inflater is ready for another file (its buffers stay)
extern "C" int gs_inflater_reset(gs_inflater *g) {
    if (!g) return gi_fail(GS_E_INVALID, "NULL argument");
    hipSetDevice(g->device);
    if (g->stream) hipStreamSynchronize(g->stream);
    g->tail = 0;
    g->cur = 0;
    g->last_text = nullptr;
    g->last_bytes = 0;
    for (int i = 0; i < 2; i++) g->comp_lo[i] = g->comp_hi[i] = -1;
    return GS_OK;
}

extern "C" int gs_inflater_destroy(gs_inflater *g) {
    if (!g) return GS_OK;
    hipSetDevice(g->device);
    if (g->stream) hipStreamSynchronize(g->stream);
    for (int i = 0; i < 2; i++) {
        hipFree(g->d_comp[i]);
        hipHostFree(g->h_comp[i]);
        hipFree(g->d_text[i]);
        if (g->comp_ready[i]) hipEventDestroy(g->comp_ready[i]);
    }
    hipFree(g->d_blocks);
    hipFree(g->d_status);
    hipFree(g->d_tiles);
    hipFree(g->d_cut);
    hipHostFree(g->h_blocks);
    hipHostFree(g->h_status);
    hipHostFree(g->h_cut);
    if (g->stream) hipStreamDestroy(g->stream);
    delete g;
    return GS_OK;
}

template <typename T>
static int gi_grow(T **p, size_t *cap, size_t need, hipStream_t s) {
    if (*cap >= need) return GS_OK;
    GI_TRY(hipStreamSynchronize(s));
    hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t want = need + need / 4 + 4096;
    GI_TRY(hipMalloc((void **)p, want * sizeof(T)));
    *cap = want;
    return GS_OK;
}

// the same for a buffer whose first `keep` bytes matter (the carried tail of the text)
static int gi_grow_keep(uint8_t **p, size_t *cap, size_t need, size_t keep, hipStream_t s) {
    if (*cap >= need) return GS_OK;
    GI_TRY(hipStreamSynchronize(s));
    uint8_t *q = nullptr;
    const size_t want = need + need / 4 + 4096;
    GI_TRY(hipMalloc((void **)&q, want));
    if (keep > 0 && *p) {
        hipError_t e = hipMemcpy(q, *p, keep, hipMemcpyDeviceToDevice);
        if (e != hipSuccess) {
            hipFree(q);
            return gi_fail(GS_E_HIP, std::string("text buffer: ") + hipGetErrorString(e));
        }
    }
    hipFree(*p);
    *p = q;
    *cap = want;
    return GS_OK;
}

// copy file[lo, hi) into staging buffer `which` (asynchronous on the inflater's stream)
static int gi_stage(gs_inflater *g, int which, const uint8_t *file, int64_t lo, int64_t hi) {
    if (g->comp_lo[which] == lo && g->comp_hi[which] == hi) return GS_OK;  // prefetched by the previous call
    int rc = gi_grow(&g->d_comp[which], &g->comp_cap[which], (size_t)(hi - lo) + 1024, g->stream);
    if (rc) return rc;
    if (g->h_comp_cap[which] < (size_t)(hi - lo)) {
        GI_TRY(hipStreamSynchronize(g->stream));
        hipHostFree(g->h_comp[which]);
        g->h_comp[which] = nullptr;
        g->h_comp_cap[which] = 0;
        const size_t want = (size_t)(hi - lo) + (size_t)(hi - lo) / 4 + 4096;
        GI_TRY(hipHostMalloc((void **)&g->h_comp[which], want));
        g->h_comp_cap[which] = want;
    }
    {   // the page-cache mapping into the page-locked buffer, on several threads (one thread moves ~5 GB/s: less than the device inflates)
        const size_t n = (size_t)(hi - lo);
        int n_thr = (int)std::min<size_t>(8, n >> 22);
        if (const char *e = getenv("GS_INFLATE_COPY_THREADS")) n_thr = std::max(1, std::min(32, atoi(e)));
        if (n_thr <= 1) {
            memcpy(g->h_comp[which], file + lo, n);
        } else {
            std::vector<std::thread> th;
            for (int t = 0; t < n_thr; t++) {
                const size_t a = n * (size_t)t / (size_t)n_thr, b = n * ((size_t)t + 1) / (size_t)n_thr;
                th.emplace_back([=] { memcpy(g->h_comp[which] + a, file + lo + a, b - a); });
            }
            for (auto &x : th) x.join();
        }
    }
    GI_TRY(hipMemcpyAsync(g->d_comp[which], g->h_comp[which], (size_t)(hi - lo), hipMemcpyHostToDevice, g->stream));
    GI_TRY(hipMemsetAsync(g->d_comp[which] + (hi - lo), 0, 1024, g->stream));
    g->comp_lo[which] = lo;
    g->comp_hi[which] = hi;
    return GS_OK;
}

extern "C" int gs_inflater_feed(gs_inflater *g, const uint8_t *file, const gs_inflate_member *members, int64_t n_members, int64_t next_lo,
                                int64_t next_hi, int last, const uint8_t **text, int64_t *n_bytes, int64_t *n_lines, int64_t *tail_bytes) {
    if (!g || !text || !n_bytes || !n_lines || n_members < 0 || (n_members > 0 && (!file || !members)))
        return gi_fail(GS_E_INVALID, "bad argument");
    GI_TRY(hipSetDevice(g->device));
    *text = nullptr;
    *n_bytes = *n_lines = 0;
    int64_t lo = 0, hi = 0, total_out = 0;
    for (int64_t i = 0; i < n_members; i++) {
        const gs_inflate_member &m = members[i];
        if (m.payload_len > (1u << 20) || m.isize > (1u << 16) || m.payload_offset < 0)
            return gi_fail(GS_E_INVALID, "a BGZF member holds at most 64 KiB of text");
        if (i == 0) lo = m.payload_offset;
        if (m.payload_offset < hi && i > 0) return gi_fail(GS_E_INVALID, "members must be in file order");
        hi = m.payload_offset + (int64_t)m.payload_len;
        total_out += m.isize;
    }
    lo &= ~(int64_t)3;  // (the payloads are read as aligned dwords)
    const int cb = g->cur;
    int rc;
    if ((rc = gi_grow_keep(&g->d_text[cb], &g->text_cap[cb], (size_t)(g->tail + total_out) + 8192, (size_t)g->tail, g->stream))) return rc;
    if ((rc = gi_grow(&g->d_blocks, &g->blocks_cap, (size_t)n_members + 1, g->stream))) return rc;
    if ((rc = gi_grow(&g->d_status, &g->status_cap, (size_t)n_members + 1, g->stream))) return rc;
    if (g->h_cap < (size_t)n_members + 1) {
        GI_TRY(hipStreamSynchronize(g->stream));
        hipHostFree(g->h_blocks);
        hipHostFree(g->h_status);
        g->h_blocks = nullptr;
        g->h_status = nullptr;
        g->h_cap = 0;
        const size_t want = (size_t)n_members * 2 + 1024;
        GI_TRY(hipHostMalloc((void **)&g->h_blocks, want * sizeof(GiBlock)));
        GI_TRY(hipHostMalloc((void **)&g->h_status, want * sizeof(int32_t)));
        g->h_cap = want;
    }
    if (n_members > 0) {
        if ((rc = gi_stage(g, cb, file, lo, hi))) return rc;
        int64_t at = g->tail;
        for (int64_t i = 0; i < n_members; i++) {
            GiBlock &b = g->h_blocks[i];
            b.in_off = (u64)(members[i].payload_offset - lo);
            b.in_len = members[i].payload_len;
            b.out_len = members[i].isize;
            b.out_off = (u64)at;
            b.crc = members[i].crc32;
            b.pad = 0;
            at += members[i].isize;
        }
        GI_TRY(hipMemcpyAsync(g->d_blocks, g->h_blocks, sizeof(GiBlock) * (size_t)n_members, hipMemcpyHostToDevice, g->stream));
        const int grid = (int)std::min<int64_t>((n_members + GI_WAVES - 1) / GI_WAVES, (int64_t)g->n_cu * gi_wgs_per_cu());
        GI_TRY(hipMemsetAsync(g->d_cut + 2, 0, sizeof(u64), g->stream));
        hipLaunchKernelGGL(gi_inflate_kernel, dim3(grid), dim3(64 * GI_WAVES), 0, g->stream, g->d_comp[cb], g->d_blocks, n_members, g->d_text[cb],
                           g->d_status, gi_force_slow(), g->d_cut + 2);
        GI_TRY(hipGetLastError());
        GI_TRY(hipMemcpyAsync(g->h_status, g->d_status, sizeof(int32_t) * (size_t)n_members, hipMemcpyDeviceToHost, g->stream));
    }
    const int64_t have = g->tail + total_out;
    const int64_t n_tiles = (have + 4095) / 4096;
    if ((rc = gi_grow(&g->d_tiles, &g->tiles_cap, (size_t)n_tiles + 1, g->stream))) return rc;
    if (have > 0) {
        hipLaunchKernelGGL(gi_count_kernel, dim3((unsigned)n_tiles), dim3(256), 0, g->stream, g->d_text[cb], have, g->d_tiles);
        hipLaunchKernelGGL(gi_cut_kernel, dim3(1), dim3(1024), 0, g->stream, g->d_text[cb], have, g->d_tiles, n_tiles, g->d_cut);
        GI_TRY(hipGetLastError());
        GI_TRY(hipMemcpyAsync(g->h_cut, g->d_cut, 2 * sizeof(u64), hipMemcpyDeviceToHost, g->stream));
    } else {
        g->h_cut[0] = g->h_cut[1] = 0;
    }
    // the compressed bytes of the NEXT call travel while this call's kernels run
    if (next_hi > next_lo && file) {
        if ((rc = gi_stage(g, cb ^ 1, file, next_lo & ~(int64_t)3, next_hi))) return rc;
    }
    GI_TRY(hipStreamSynchronize(g->stream));
    for (int64_t i = 0; i < n_members; i++)
        if (g->h_status[i] != GI_OK)
            return gi_fail(GS_E_INVALID, "corrupt BGZF member " + std::to_string(i) + " of this run (inflate status " + std::to_string(g->h_status[i]) + ")");
    int64_t lines = (int64_t)g->h_cut[0], cut = (int64_t)g->h_cut[1];
    int64_t whole_lines = lines & ~(int64_t)3;
    if (last && (lines & 3) == 0 && cut < have) {
        // (the file ends without a final newline behind whole records: the tail is an unterminated line the caller must deal with)
    }
    *text = g->d_text[cb];
    g->last_text = g->d_text[cb];
    g->last_bytes = cut;
    *n_bytes = cut;
    *n_lines = whole_lines;
    const int64_t tail = have - cut;
    if (tail_bytes) *tail_bytes = tail;
    // the rest opens the other buffer for the next call
    const int nb = cb ^ 1;
    if ((rc = gi_grow(&g->d_text[nb], &g->text_cap[nb], (size_t)tail + 8192, g->stream))) return rc;
    if (tail > 0) GI_TRY(hipMemcpyAsync(g->d_text[nb], g->d_text[cb] + cut, (size_t)tail, hipMemcpyDeviceToDevice, g->stream));
    GI_TRY(hipStreamSynchronize(g->stream));
    g->tail = tail;
    g->cur = nb;
    return GS_OK;
}

// the tail the last call left over (an unterminated last line, or fewer than four lines): copied to the host for the caller's parser
extern "C" int gs_inflater_tail(gs_inflater *g, uint8_t *out, int64_t cap, int64_t *n) {
    if (!g || !n) return gi_fail(GS_E_INVALID, "NULL argument");
    *n = g->tail;
    if (g->tail == 0) return GS_OK;
    if (!out || cap < g->tail) return gi_fail(GS_E_INVALID, "tail buffer too small");
    GI_TRY(hipSetDevice(g->device));
    GI_TRY(hipMemcpy(out, g->d_text[g->cur], (size_t)g->tail, hipMemcpyDeviceToHost));
    return GS_OK;
}

extern "C" int gs_inflater_fetch(gs_inflater *g, uint8_t *out, int64_t n_bytes) {
    if (!g || n_bytes < 0 || (n_bytes > 0 && !out)) return gi_fail(GS_E_INVALID, "bad argument");
    if (n_bytes == 0) return GS_OK;
    if (g->last_text == nullptr || n_bytes > g->last_bytes) return gi_fail(GS_E_STATE, "more bytes than the last feed returned");
    GI_TRY(hipSetDevice(g->device));
    GI_TRY(hipMemcpy(out, g->last_text, (size_t)n_bytes, hipMemcpyDeviceToHost));
    return GS_OK;
}

// ---- single-member gzip on the device (kernels above: gi_find_kernel .. gi_crc_kernel)
static uint32_t gi_h_gf_mul(uint32_t a, uint32_t b) {  // (the device functions gi_gf_mul / gi_x_pow_8n again, for the host)
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) {
        if (a & 0x80000000u) p ^= b;
        a <<= 1;
        b = (b >> 1) ^ ((b & 1u) ? 0xedb88320u : 0u);
    }
    return p;
}
static uint32_t gi_h_x_pow_8n(uint64_t n) {
    uint32_t r = 0x80000000u, sq = 0x00800000u;
    while (n) {
        if (n & 1u) r = gi_h_gf_mul(r, sq);
        sq = gi_h_gf_mul(sq, sq);
        n >>= 1;
    }
    return r;
}

// host memory of any kind (a page-cache mapping is copied by the runtime at a crawl) to the device through two page-locked buffers,
// filled by several threads while the other one is on its way
static double gi_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static bool gi_trace() {  // GS_HOST_TRACE: the host layer's timeline on stderr; here the stages of a gunzip batch
    static const bool on = getenv("GS_HOST_TRACE") != nullptr;
    return on;
}

// The device as the upload threads of gs_upload.h see it: a stream, events, and two page-locked pieces of 32 MiB taken from a per-device
// free list for the length of ONE copy -- a second decoder on the same device (a filter and a match job, two threads of a JVM) gets a
// set of its own instead of waiting for a whole stream.
struct GiHipDev {
    typedef hipEvent_t Event;
    int device = 0;
    hipStream_t stream = nullptr;
    struct Staging {
        uint8_t *h[2] = {nullptr, nullptr};
        hipEvent_t done[2] = {nullptr, nullptr};
        size_t piece = 0;
    };
    struct StagingList {
        std::mutex m;
        std::vector<Staging> idle;
    };
    static StagingList &list(int dev) {
        static StagingList per_device[64];
        return per_device[(dev >= 0 && dev < 64) ? dev : 0];
    }
    int bind() { return hipSetDevice(device) == hipSuccess ? GS_OK : GS_E_HIP; }
    int take_staging(uint8_t *h[2], Event ev[2], size_t piece) {
        Staging st;
        {
            StagingList &sl = list(device);
            std::lock_guard<std::mutex> l(sl.m);
            for (size_t i = 0; i < sl.idle.size(); i++)
                if (sl.idle[i].piece == piece) {
                    st = sl.idle[i];
                    sl.idle.erase(sl.idle.begin() + (long)i);
                    break;
                }
        }
        st.piece = piece;
        for (int i = 0; i < 2; i++) {
            if (!st.h[i] && hipHostMalloc((void **)&st.h[i], piece) != hipSuccess) st.h[i] = nullptr;
            if (!st.done[i] && hipEventCreateWithFlags(&st.done[i], hipEventDisableTiming) != hipSuccess) st.done[i] = nullptr;
            h[i] = st.h[i];
            ev[i] = st.done[i];
        }
        if (!h[0] || !h[1] || !ev[0] || !ev[1]) {
            give_staging(h, ev);
            return GS_E_NOMEM;
        }
        piece_ = piece;
        return GS_OK;
    }
    void give_staging(uint8_t *h[2], Event ev[2]) {
        Staging st;
        for (int i = 0; i < 2; i++) {
            st.h[i] = h[i];
            st.done[i] = ev[i];
        }
        st.piece = piece_;
        StagingList &sl = list(device);
        std::lock_guard<std::mutex> l(sl.m);
        sl.idle.push_back(st);
    }
    int copy_async(uint8_t *d_dst, const uint8_t *h_src, size_t n) { return hipMemcpyAsync(d_dst, h_src, n, hipMemcpyHostToDevice, stream) == hipSuccess ? GS_OK : GS_E_HIP; }
    int record(Event ev) { return hipEventRecord(ev, stream) == hipSuccess ? GS_OK : GS_E_HIP; }
    int wait_event(Event ev) { return hipEventSynchronize(ev) == hipSuccess ? GS_OK : GS_E_HIP; }
    int drain() { return hipStreamSynchronize(stream) == hipSuccess ? GS_OK : GS_E_HIP; }
    size_t piece_ = 0;
};
static const size_t GI_UPLOAD_PIECE = (size_t)32 << 20;
static int gi_copy_threads() {
    if (const char *e = getenv("GS_INFLATE_COPY_THREADS")) return std::max(1, std::min(32, atoi(e)));
    return 0;  // (by the size of the piece: gs_staged_copy)
}

// `after_piece(bytes uploaded so far, the event behind that piece's copy)`: work on other streams that waits for the piece
static int gi_h2d_staged(uint8_t *d_dst, const uint8_t *src, size_t n, hipStream_t stream, const std::function<int(size_t, hipEvent_t)> &after_piece) {
    GiHipDev dev;
    GI_TRY(hipGetDevice(&dev.device));
    dev.stream = stream;
    const int rc = gs_staged_copy<GiHipDev>(dev, d_dst, src, n, GI_UPLOAD_PIECE, gi_copy_threads(), after_piece);
    if (rc && rc != GS_E_STATE) return gi_fail(rc, "the copy of the compressed bytes to the device failed");
    return rc;
}

struct GiDevBufs {  // freed on every way out
    std::vector<void *> p;
    template <typename T>
    hipError_t get(T **q, size_t bytes) {
        void *v = nullptr;
        const hipError_t e = hipMalloc(&v, bytes ? bytes : 1);
        if (e == hipSuccess) p.push_back(v);
        *q = static_cast<T *>(v);
        return e;
    }
    void keep(void *v) { p.erase(std::remove(p.begin(), p.end(), v), p.end()); }
    ~GiDevBufs() {
        for (void *v : p) hipFree(v);
    }
};

// the CRC kernel for other translation units (gs_deflate_dev.hip: the members it writes): raw registers (no pre / post inversion)
// of the tiles of `tile` bytes of d_text[0, n) into d_crc, on `stream`; a tile that is a multiple of 1024 bytes takes the fast join
extern "C" int gs_crc_tiles_device(const uint8_t *d_text, int64_t n, uint32_t tile, uint32_t *d_crc, hipStream_t stream) {
    if (n <= 0) return GS_OK;
    if (tile < 64 || (tile & 63u)) return gi_fail(GS_E_INVALID, "gs_crc_tiles_device: the tile must be a multiple of 64 bytes");
    int rc = gi_upload_crc_table();
    if (rc) return rc;
    GiCrcPow pw;
    for (int sidx = 0; sidx < 6; sidx++) pw.p[sidx] = gi_h_x_pow_8n((uint64_t)(tile / 64u) << sidx);
    const int64_t n_tiles = (n + tile - 1) / tile;
    hipLaunchKernelGGL(gi_crc_kernel, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, stream, d_text, n, tile, d_crc, pw);
    GI_TRY(hipGetLastError());
    return GS_OK;
}
// CRC-32 of n bytes from their raw register: ~(raw ^ 0xffffffff x^(8 n))
extern "C" uint32_t gs_crc_finish_raw(uint32_t raw, uint64_t n) { return ~(raw ^ gi_h_gf_mul(0xffffffffu, gi_h_x_pow_8n(n))); }
extern "C" uint32_t gs_crc_init_term(uint64_t n) { return gi_h_gf_mul(0xffffffffu, gi_h_x_pow_8n(n)); }

// ---- the streaming form: the stream is taken in BATCHES of about one segment per wave slot of the device (its compressed bytes, the
// finder, the segments, the window pass chained to the window the batch before left, the text, the CRC-32 folded into a running
// register), so that a file of any size goes through buffers of a few gigabytes.
struct gs_gunzipper {
    int device = 0, n_cu = 256;
    const uint8_t *gz = nullptr;  // the mapped file (stays the caller's)
    int64_t n = 0, hdr = 0, in_len = 0;
    uint32_t want_crc = 0, want_isize = 0;
    u64 bit = 0;          // of the next batch's first block, from the first byte of the deflate stream
    bool done = false, more_members = false;
    uint32_t raw = 0xffffffffu;  // running CRC-32 register (before the final inversion)
    u64 total = 0;               // text bytes so far
    uint32_t chunk = 65536, ratio = 16;
    uint32_t fchunk = 8192;   // the finder's unit of work (GS_GUNZIP_FIND_CHUNK)
    double block_bytes = 0;   // compressed bytes per deflate block, as the batch before found them (0: not known yet)
    int64_t first_span = 0;   // > 0: the first batch takes at most this many compressed bytes (gs_gunzipper_first_span)
    // A stream of up to GS_GUNZIP_WHOLE_MAX bytes (16 GiB) is uploaded WHOLE, by a thread of its own, while the batches are decoded: only
    // the first batch waits for its bytes, and nothing is uploaded twice.  (Larger streams: every batch uploads its own span, d_in.)
    bool whole = false;
    uint8_t *d_all = nullptr;
    size_t all_cap = 0;
    GiHipDev up_dev;            // the upload's view of the device (stream s_up)
    GsUploader<GiHipDev> up;    // the thread itself, the bytes that have arrived, cancel / park: gs_upload.h (sanitizer harness: tests/native/uploader_sanitize.cpp)
    int text_only = 1;
    hipStream_t s_up = nullptr, s_find[2] = {nullptr, nullptr};  // the upload, and the finder launches behind its pieces
    unsigned long long *d_fq = nullptr;                        // one work counter per finder launch
    int64_t n_batches = 0, n_segments = 0, n_mirages = 0, n_chunks = 0, n_members = 1;
    // device buffers, grown as needed
    uint8_t *d_in = nullptr, *d_win = nullptr, *d_prev = nullptr, *d_text = nullptr, *d_tail = nullptr;
    u64 *d_start = nullptr, *d_end = nullptr, *d_off = nullptr;
    unsigned long long *d_q = nullptr;
    GiSeg *d_segs = nullptr;
    int32_t *d_status = nullptr;
    uint32_t *d_len = nullptr, *d_crc = nullptr;
    uint16_t *d_sym = nullptr, *d_win16 = nullptr;
    size_t in_cap = 0, start_cap = 0, seg_cap = 0, sym_cap = 0, win_cap = 0, win16_cap = 0, text_cap = 0, tail_cap = 0, crc_cap = 0;
    bool have_prev = false;
    int64_t last_n_text = 0;
};

template <typename T>
static int gu_grow(T **p, size_t *cap, size_t need, size_t slack_pct = 12) {
    if (*cap >= need && *p) return GS_OK;
    hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t want = need + need * slack_pct / 100 + 256;
    const hipError_t e = hipMalloc((void **)p, want * sizeof(T));
    if (e != hipSuccess) return gi_fail(e == hipErrorOutOfMemory ? GS_E_NOMEM : GS_E_HIP, std::string("gs_gunzipper: ") + hipGetErrorString(e));
    *cap = want;
    return GS_OK;
}

static int64_t gu_slots(int n_cu) {  // chunks per batch: the device's wave slots (GS_GUNZIP_SLOTS: tests force many small batches)
    if (const char *e = getenv("GS_GUNZIP_SLOTS")) return std::max(1, std::min(1 << 20, atoi(e)));
    return (int64_t)n_cu * GI_WAVES * gi_seg_wgs_per_cu();
}

// A caller with writers behind it wants the first text early, not the most text per batch: the FIRST batch of the stream then takes at
// most `bytes` of compressed data (0: as much as a batch holds).  Call between open / reopen and the first gs_gunzipper_next.
extern "C" int gs_gunzipper_first_span(gs_gunzipper *g, int64_t bytes) {
    if (!g || bytes < 0) return gi_fail(GS_E_INVALID, "bad argument");
    g->first_span = bytes;
    return GS_OK;
}

// the caller is through with the file (or gives up on it): the upload thread is stopped -- `gz` may go away; the buffers stay
extern "C" int gs_gunzipper_park(gs_gunzipper *g) {
    if (!g) return GS_OK;
    g->up.park();
    return GS_OK;
}

static void gu_start_upload(gs_gunzipper *g) {  // (throws std::system_error when no thread can be had)
    g->up_dev.device = g->device;
    g->up_dev.stream = g->s_up;
    g->up.start(&g->up_dev, g->d_all, g->gz + g->hdr, g->in_len, GI_UPLOAD_PIECE, gi_copy_threads());
}

// until `need` bytes of the stream have arrived; *have: how many have
static int gu_wait_uploaded(gs_gunzipper *g, int64_t need, int64_t *have) {
    const int rc = g->up.wait(need, have);
    if (rc) return gi_fail(rc, "gs_gunzipper: the upload of the stream failed");
    return GS_OK;
}

extern "C" int gs_gunzipper_close(gs_gunzipper *g) {
    if (!g) return GS_OK;
    gs_gunzipper_park(g);
    hipSetDevice(g->device);
    hipDeviceSynchronize();
    for (hipStream_t st : {g->s_up, g->s_find[0], g->s_find[1]})
        if (st) hipStreamDestroy(st);
    for (void *p : {(void *)g->d_in, (void *)g->d_win, (void *)g->d_prev, (void *)g->d_text, (void *)g->d_tail, (void *)g->d_start, (void *)g->d_end, (void *)g->d_off,
                    (void *)g->d_all, (void *)g->d_fq, (void *)g->d_q, (void *)g->d_segs, (void *)g->d_status, (void *)g->d_len, (void *)g->d_crc, (void *)g->d_sym, (void *)g->d_win16})
        hipFree(p);
    delete g;
    return GS_OK;
}

// length of the member header at p (RFC 1952), or -1: not a gzip member / truncated
static int64_t gi_gzip_header(const uint8_t *gz, int64_t n) {
    if (n < 18 || gz[0] != 0x1f || gz[1] != 0x8b || gz[2] != 8 || (gz[3] & 0xe0)) return -1;
    const int flg = gz[3];
    int64_t hdr = 10;
    if (flg & 4) {
        if (hdr + 2 > n) return -1;
        hdr += 2 + ((int64_t)gz[hdr] | ((int64_t)gz[hdr + 1] << 8));
    }
    for (int bit : {8, 16})
        if (flg & bit) {
            while (hdr < n && gz[hdr] != 0) hdr++;
            hdr++;
        }
    if (flg & 2) hdr += 2;
    return hdr + 8 >= n ? -1 : hdr;
}

extern "C" int gs_gunzipper_reopen(gs_gunzipper *g, const uint8_t *gz, int64_t n);

extern "C" int gs_gunzipper_open(gs_gunzipper **out, int device, const uint8_t *gz, int64_t n) {
    if (!out || !gz || n < 18) return gi_fail(GS_E_INVALID, "bad argument");
    *out = nullptr;
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd < 1) return gi_fail(GS_E_NODEVICE, "no usable gfx950 device");
    const int64_t hdr = gi_gzip_header(gz, n);
    if (hdr < 0) return gi_fail(GS_E_INVALID, "not a gzip stream, or a truncated one");
    GI_TRY(hipSetDevice(device));
    int rc = gi_upload_crc_table();
    if (rc) return rc;
    gs_gunzipper *g = new gs_gunzipper();
    g->device = device;
    hipDeviceProp_t prop;
    g->n_cu = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    hipError_t e = hipMalloc((void **)&g->d_q, 2 * sizeof(u64));
    if (e == hipSuccess) e = hipMalloc((void **)&g->d_fq, GI_FIND_LAUNCHES * sizeof(u64));
    if (e == hipSuccess) e = hipMalloc((void **)&g->d_prev, GI_WINDOW);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->s_up, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->s_find[0], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->s_find[1], hipStreamNonBlocking);
    if (e != hipSuccess) {
        gs_gunzipper_close(g);
        return gi_fail(GS_E_NOMEM, "gs_gunzipper_open");
    }
    rc = gs_gunzipper_reopen(g, gz, n);
    if (rc) {
        gs_gunzipper_close(g);
        return rc;
    }
    *out = g;
    return GS_OK;
}

// the same object (and its device buffers: a batch's symbols alone are gigabytes) on another file
extern "C" int gs_gunzipper_reopen(gs_gunzipper *g, const uint8_t *gz, int64_t n) {
    if (!g || !gz || n < 18) return gi_fail(GS_E_INVALID, "bad argument");
    const int64_t hdr = gi_gzip_header(gz, n);
    if (hdr < 0) return gi_fail(GS_E_INVALID, "not a gzip stream, or a truncated one");
    gs_gunzipper_park(g);
    GI_TRY(hipSetDevice(g->device));
    GI_TRY(hipDeviceSynchronize());  // (nobody reads the last file's text any more)
    g->gz = gz;
    g->n = n;
    g->hdr = hdr;
    g->in_len = n - hdr;  // (what lies behind the deflate stream -- the trailer, further members -- is found when the final block is)
    g->bit = 0;
    g->done = g->more_members = g->have_prev = false;
    g->raw = 0xffffffffu;
    g->total = 0;
    g->n_batches = g->n_segments = g->n_mirages = g->n_chunks = 0;
    g->block_bytes = 0;
    g->first_span = 0;
    g->n_members = 1;
    g->last_n_text = 0;
    g->text_only = 1;
    const int64_t slots = gu_slots(g->n_cu);
    // The finder's unit of work, and the unit a batch is measured in (one chunk per wave slot): not less than 16 KiB of compressed data
    // nor more than 64 KiB (320 MiB on 256 CUs: then the stream takes several batches).
    g->chunk = (uint32_t)std::min<int64_t>(65536, std::max<int64_t>(16384, g->in_len / slots));
    if (const char *e = getenv("GS_GUNZIP_CHUNK")) g->chunk = (uint32_t)std::max(4096, std::min(1 << 24, atoi(e)));
    g->fchunk = std::min<uint32_t>(g->chunk, 8192);
    if (const char *e = getenv("GS_GUNZIP_FIND_CHUNK")) g->fchunk = (uint32_t)std::max(4096, std::min(1 << 24, atoi(e)));
    if (const char *e = getenv("GS_GUNZIP_ANY_BYTES")) g->text_only = atoi(e) == 0;  // block starts whose literal code covers bytes >= 128 count as well
    // symbols a segment may produce per byte of its compressed span: twice what FASTQ does (4 .. 6 : 1); a segment that outgrows its
    // room is decoded again with eight times as much (deflate's limit is 1032 : 1), and when the slack for that is used up the rest
    // of the file goes to the host decoders
    g->ratio = 12;
    if (g->in_len < ((int64_t)64 << 20)) {  // a small stream says how far it expands (ISIZE, modulo 2^32)
        const uint8_t *t = gz + n - 8;
        u64 est = (u64)t[4] | ((u64)t[5] << 8) | ((u64)t[6] << 16) | ((u64)t[7] << 24);
        g->ratio = (uint32_t)std::min<u64>(1040, std::max<u64>(12, 4 * (est / (u64)std::max<int64_t>(g->in_len, 1)) + 4));
    }
    if (const char *e = getenv("GS_GUNZIP_RATIO")) g->ratio = (uint32_t)std::max(2, std::min(1040, atoi(e)));
    int64_t whole_max = (int64_t)16 << 30;
    if (const char *e = getenv("GS_GUNZIP_WHOLE_MAX")) whole_max = atoll(e);
    g->whole = g->in_len <= whole_max;
    if (g->whole && (size_t)g->in_len + 1024 > g->all_cap) {  // (a buffer that has to grow: at most a quarter of what is free -- the batches need room too)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || (size_t)g->in_len > free_b / 4) g->whole = false;
    }
    if (g->whole && gu_grow(&g->d_all, &g->all_cap, (size_t)g->in_len + 1024) != GS_OK) g->whole = false;  // (no room: batch by batch)
    if (g->whole) {
        GI_TRY(hipMemset(g->d_all + g->in_len, 0, 1024));
        GI_TRY(hipDeviceSynchronize());
        try {
            gu_start_upload(g);
        } catch (const std::system_error &) {  // no thread to be had: every batch uploads its own span
            g->whole = false;
        }
    }
    return GS_OK;
}

// One batch.  keep_tail: the last `keep_tail` bytes of the text the call before returned stay in front of the new text (what lay behind
// the caller's last whole record).  *last: 0 = more batches follow (of this member, or of the member behind it), 1 = the file is through
// (every member's CRC-32 and ISIZE were right).  The pointer is valid until the next call.
extern "C" int gs_gunzipper_next(gs_gunzipper *g, int64_t keep_tail, const uint8_t **d_text_out, int64_t *n_text, int *last) {
    if (!g || !d_text_out || !n_text || !last || keep_tail < 0 || keep_tail > g->last_n_text) return gi_fail(GS_E_INVALID, "bad argument");
    *d_text_out = nullptr;
    *n_text = 0;
    *last = 0;
    if (g->done) return gi_fail(GS_E_STATE, "the stream is through");
    GI_TRY(hipSetDevice(g->device));
    int rc;
    // the tail of the batch before, out of the way
    if (keep_tail > 0) {
        if ((rc = gu_grow(&g->d_tail, &g->tail_cap, (size_t)keep_tail))) return rc;
        GI_TRY(hipMemcpy(g->d_tail, g->d_text + (g->last_n_text - keep_tail), (size_t)keep_tail, hipMemcpyDeviceToDevice));
    }
    const int64_t slots = gu_slots(g->n_cu);
    const int wgs = g->n_cu * gi_seg_wgs_per_cu();
    const u64 base = (g->bit / 8u) & ~(u64)3;  // of the device copy, in the deflate stream
    const u64 rel0 = g->bit - base * 8u;
    const int64_t remain = g->in_len - (int64_t)base;
    int64_t nb_chunks = std::min<int64_t>(slots, (remain + g->chunk - 1) / g->chunk);
    if (g->n_batches == 0 && g->first_span > 0) nb_chunks = std::min<int64_t>(nb_chunks, std::max<int64_t>(1, g->first_span / g->chunk));
    if (g->block_bytes > 0 && (remain + g->chunk - 1) / g->chunk > slots) {  // (what is left fits one batch: all of it, m = ceil below)
        // as many bytes as hold a whole number of blocks per wave slot (and 3 % more: blocks differ a little): what lies behind the
        // batch's last segment is uploaded and searched again by the next batch
        const double room = (double)nb_chunks * g->chunk;
        const int64_t per_slot = std::max<int64_t>(1, (int64_t)(room / ((double)slots * g->block_bytes)));
        nb_chunks = std::min<int64_t>(nb_chunks, (int64_t)((double)per_slot * (double)slots * g->block_bytes * 1.03 / g->chunk) + 1);
    }
    const int64_t span = std::min<int64_t>(remain, (nb_chunks + 64) * (int64_t)g->chunk);
    const bool to_end = span == remain;
    const int64_t n_chunks = (span + g->chunk - 1) / g->chunk;
    if (span >= ((int64_t)1 << 31)) return gi_fail(GS_E_UNSUPPORTED, "a batch of more than 2 GiB");
    const uint32_t in_len = (uint32_t)span;
    if (!g->whole && (rc = gu_grow(&g->d_in, &g->in_cap, (size_t)span + 1024))) return rc;
    const uint8_t *const d_in = g->whole ? g->d_all + base : g->d_in;  // (whole: what lies behind the span is the stream itself, and zeros behind its end)
    // 1. the compressed bytes in pieces, and behind every piece the block finder over the chunks that are complete with it (all block
    // starts: the first GI_FIND_MAX of a chunk): the search runs while the next pieces are copied
    const int64_t fc = g->fchunk;
    const int64_t n_fchunks = (span + fc - 1) / fc;
    const int64_t fin_first = to_end ? std::max<int64_t>(0, n_fchunks - (((int64_t)1 << 20) + fc - 1) / fc) : n_fchunks;
    const int64_t n_search = n_fchunks + (n_fchunks - fin_first);  // (the last MiB twice: gi_find_kernel)
    if ((rc = gu_grow(&g->d_start, &g->start_cap, (size_t)n_search * GI_FIND_MAX))) return rc;
    const double t_0 = gi_now_ms();
    if (!g->whole) GI_TRY(hipMemset(g->d_in + span, 0, 1024));
    GI_TRY(hipMemset(g->d_start, 0xff, sizeof(u64) * (size_t)n_search * GI_FIND_MAX));
    GI_TRY(hipMemset(g->d_fq, 0, GI_FIND_LAUNCHES * sizeof(u64)));
    GI_TRY(hipStreamSynchronize(0));  // (the upload and the finder run on streams of their own, which do not wait for the null stream)
    int64_t searched = 0;
    int n_launch = 0;
    auto find_upto = [&](int64_t upto, hipEvent_t ev) -> int {  // the work items searched .. upto
        if (upto <= searched) return GS_OK;
        hipStream_t st = g->s_find[n_launch & 1];
        if (ev) GI_TRY(hipStreamWaitEvent(st, ev, 0));
        const int64_t n_items = upto - searched;
        hipLaunchKernelGGL(gi_find_kernel, dim3((unsigned)std::min<int64_t>((n_items + GI_WAVES - 1) / GI_WAVES, (int64_t)g->n_cu * gi_wgs_per_cu())), dim3(64 * GI_WAVES), 0, st,
                           d_in, in_len, g->fchunk, searched, upto, g->d_start, g->d_fq + n_launch, g->text_only, n_fchunks, fin_first);
        GI_TRY(hipGetLastError());
        searched = upto;
        n_launch++;
        return GS_OK;
    };
    // (A launch behind every second piece of 32 MiB, 8 KiB to a wave: 8 192 chunks, two rounds over the wave slots, about as long as the
    // two pieces take to arrive.  The launches share one hardware queue -- the process has more streams than queues -- and run one after
    // the other: smaller launches leave the device half empty, larger ones leave more to do when the last piece is there.
    // 238 MB: 4.9 ms of upload + 2.2 ms against 4.9 + 4.5 with one launch behind the upload.)
    int n_pieces = 0, every = 2;
    if (const char *e = getenv("GS_GUNZIP_FIND_EVERY")) every = std::max(1, atoi(e));
    if (g->whole) {
        // the upload thread is ahead of the batches: the first one follows it in steps of `every` pieces, the later ones find their
        // bytes in place (one launch, the whole device)
        int64_t seen = 0;
        while (seen < span) {
            int64_t have = 0;
            if ((rc = gu_wait_uploaded(g, (int64_t)base + std::min<int64_t>(span, seen + (int64_t)every * (32 << 20)), &have))) return rc;
            seen = std::min<int64_t>(span, have - (int64_t)base);
            if (seen < span && n_launch < GI_FIND_LAUNCHES - 1 && (rc = find_upto((seen - 4096) / fc, nullptr))) return rc;
        }
    } else {
        rc = gi_h2d_staged(g->d_in, g->gz + g->hdr + base, (size_t)span, g->s_up, [&](size_t up, hipEvent_t ev) -> int {
            if (++n_pieces % every != 0 || (int64_t)up >= span || n_launch >= GI_FIND_LAUNCHES - 1) return GS_OK;  // (the rest in one launch, below)
            return find_upto(((int64_t)up - 4096) / fc, ev);  // (a candidate is looked at up to ~600 bytes behind its chunk)
        });
        if (rc) return rc;
    }
    const double t_up = gi_now_ms();
    if ((rc = find_upto(n_search, nullptr))) return rc;
    GI_TRY(hipStreamSynchronize(g->s_find[0]));
    GI_TRY(hipStreamSynchronize(g->s_find[1]));
    std::vector<u64> found((size_t)n_search * GI_FIND_MAX);
    GI_TRY(hipMemcpy(found.data(), g->d_start, sizeof(u64) * found.size(), hipMemcpyDeviceToHost));
    const double t_find = gi_now_ms();
    std::vector<u64> cands{rel0}, beyond;  // block starts in the batch's chunks (the batch's first block in front), and behind them
    const u64 batch_bits = (u64)nb_chunks * g->chunk * 8u;
    for (int64_t c = 0; c < n_search; c++)
        for (int j = 0; j < GI_FIND_MAX; j++) {
            const u64 v = found[(size_t)c * GI_FIND_MAX + (size_t)j];
            if (v == ~0ULL) break;
            if (v > rel0) (v < batch_bits ? cands : beyond).push_back(v);
        }
    // (in stream order: the final block's place among the others; and the finder queues the candidates of 2 048 offsets lane by lane)
    std::sort(cands.begin(), cands.end());
    std::sort(beyond.begin(), beyond.end());
    // One segment per wave slot and ONE ROUND of segments per batch: a segment is a wave's work from beginning to end, so the batch
    // takes as long as its longest segment whatever the others do -- blocks of the stream (50 KB of gzip -1 .. -6 FASTQ, 300 KB of
    // text, ~12 ms) are dealt out whole, m = floor(blocks / slots) to a segment, and what is left over when the blocks are not a
    // multiple of the slots waits for the next batch (found again there) instead of costing this one a second round.  At the end of
    // the stream nothing follows that the blocks left over could join: m = ceil(blocks / slots) then, and one batch less.
    const size_t n_cand = cands.size();
    if (n_cand > 16) g->block_bytes = (double)(cands.back() - cands.front()) / 8.0 / (double)(n_cand - 1);
    size_t per_seg = std::max<size_t>(1, n_cand / (size_t)slots);
    if (to_end && n_cand > (size_t)slots * per_seg) per_seg++;
    const size_t keep = std::min(n_cand, (size_t)slots * per_seg);
    std::vector<GiSeg> segs;
    for (size_t i = 0; i < keep; i += per_seg) {
        if (!segs.empty()) segs.back().stop_bit = cands[i];
        GiSeg sg{};
        sg.start_bit = cands[i];
        segs.push_back(sg);
    }
    std::vector<u64> stops;  // found starts behind the batch's segments: where its last segment may end (the first, or -- mirages -- one of the next)
    for (size_t i = keep; i < n_cand && stops.size() < 8; i++) stops.push_back(cands[i]);
    for (size_t i = 0; i < beyond.size() && stops.size() < 8; i++) stops.push_back(beyond[i]);
    size_t stop_at = 0;
    if (stops.empty() && !to_end) return gi_fail(GS_E_UNSUPPORTED, "no block start in 64 chunks behind a batch: host decoders");
    u64 sym_total = 0;
    for (size_t i = 0; i < segs.size(); i++) {
        const bool is_last = i + 1 == segs.size();
        if (is_last) {
            segs[i].to_final = stops.empty();
            segs[i].stop_bit = stops.empty() ? 0 : stops[0];
        }
        const u64 end = is_last ? (stops.empty() ? (u64)in_len * 8u : stops.back()) : segs[i].stop_bit;  // (room up to the farthest stop: it may have to run on)
        const u64 span_b = (end - segs[i].start_bit) / 8u + 1u;
        const u64 cap = (span_b * g->ratio + 65536u + 7u) & ~(u64)7;  // (a multiple of eight: gi_resolve_kernel reads 16 bytes at a time)
        if (cap > 0xffff0000ull) return gi_fail(GS_E_UNSUPPORTED, "a segment of more than 4 G symbols");
        segs[i].out_cap = (uint32_t)cap;
        sym_total += GI_WINDOW;
        segs[i].out_off = sym_total;
        sym_total += cap;
    }
    const int64_t n_segs = (int64_t)segs.size();
    // 2. segments
    u64 sym_used = sym_total;
    const u64 sym_room = sym_total + sym_total / 8 + ((u64)64 << 20);  // slack for segments that are decoded again (mirages, below)
    if ((rc = gu_grow(&g->d_sym, &g->sym_cap, (size_t)sym_room + 64, 0))) return rc;
    if (g->seg_cap < (size_t)n_segs) {
        size_t c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
        if ((rc = gu_grow(&g->d_segs, &c1, (size_t)n_segs)) || (rc = gu_grow(&g->d_status, &c2, (size_t)n_segs)) || (rc = gu_grow(&g->d_len, &c3, (size_t)n_segs)) ||
            (rc = gu_grow(&g->d_end, &c4, (size_t)n_segs)) || (rc = gu_grow(&g->d_off, &c5, (size_t)n_segs)))
            return rc;
        g->seg_cap = std::min(std::min(c1, c2), std::min(std::min(c3, c4), c5));
    }
    GI_TRY(hipMemcpy(g->d_segs, segs.data(), sizeof(GiSeg) * (size_t)n_segs, hipMemcpyHostToDevice));
    GI_TRY(hipMemset(g->d_q + 1, 0, sizeof(u64)));
    hipLaunchKernelGGL(gi_segment_kernel, dim3((unsigned)std::min<int64_t>((n_segs + GI_WAVES - 1) / GI_WAVES, wgs)), dim3(64 * GI_WAVES), 0, 0, d_in, in_len, g->d_segs,
                       n_segs, g->d_sym, g->d_status, g->d_len, g->d_end, g->d_q + 1);
    GI_TRY(hipGetLastError());
    std::vector<int32_t> st((size_t)n_segs);
    std::vector<uint32_t> len((size_t)n_segs);
    std::vector<u64> endb((size_t)n_segs);
    GI_TRY(hipMemcpy(st.data(), g->d_status, sizeof(int32_t) * (size_t)n_segs, hipMemcpyDeviceToHost));
    GI_TRY(hipMemcpy(len.data(), g->d_len, sizeof(uint32_t) * (size_t)n_segs, hipMemcpyDeviceToHost));
    GI_TRY(hipMemcpy(endb.data(), g->d_end, sizeof(u64) * (size_t)n_segs, hipMemcpyDeviceToHost));
    const double t_seg = gi_now_ms();
    // A block start that was a mirage (a bit pattern that parses as a complete dynamic header: about one per 100 MB of compressed
    // data without the text test) shows as the segment IN FRONT of it running past it: that segment is decoded again up to the start
    // after the mirage, into the slack behind the symbols (the mirage's own segment is dropped); for the batch's last segment the
    // next of the starts behind the batch takes the mirage's place.  A few rounds: two mirages may follow each other.
    bool member_end = false;
    for (int round = 0; round < 6; round++) {
        {   // the member's final block inside a segment (another member follows): that segment is the last one -- once nothing in front
            // of it is in doubt; what the batch decoded behind it belongs to the next member and is decoded again from its first block
            size_t f = segs.size(), y = segs.size();
            for (size_t i = 0; i < segs.size(); i++) {
                if (st[i] == GI_FINAL && f == segs.size()) f = i;
                if (st[i] == GI_E_SYNC && y == segs.size()) y = i;
            }
            if (f < segs.size() && y > f) {
                segs.resize(f + 1);
                st.resize(f + 1);
                len.resize(f + 1);
                endb.resize(f + 1);
                segs[f].to_final = 1;
                st[f] = GI_OK;
                member_end = true;
            }
        }
        std::vector<GiSeg> redo, kept;
        std::vector<size_t> redo_at;
        std::vector<int32_t> kst;
        std::vector<uint32_t> klen;
        std::vector<u64> kend;
        for (size_t i = 0; i < segs.size(); i++) {
            const bool is_last = i + 1 == segs.size();
            if (st[i] == GI_E_SYNC && !member_end && (!is_last || (!segs[i].to_final && (stop_at + 1 < stops.size() || to_end)))) {
                GiSeg m = segs[i];
                u64 cap = m.out_cap;
                if (!is_last) {
                    const GiSeg &gone = segs[i + 1];
                    m.stop_bit = gone.stop_bit;
                    m.to_final = gone.to_final;
                    cap += gone.out_cap;
                } else {
                    stop_at++;
                    m.to_final = stop_at >= stops.size();
                    m.stop_bit = m.to_final ? 0 : stops[stop_at];
                }
                if (cap > 0xffff0000ull || sym_used + GI_WINDOW + cap > sym_room) return gi_fail(GS_E_UNSUPPORTED, "no room to decode a segment again: host decoders");
                m.out_cap = (uint32_t)cap;
                sym_used += GI_WINDOW;
                m.out_off = sym_used;
                sym_used += cap;
                redo.push_back(m);
                redo_at.push_back(kept.size());
                kept.push_back(m);
                kst.push_back(GI_OK);
                klen.push_back(0);
                kend.push_back(0);
                g->n_mirages++;
                if (!is_last) i++;  // (the segment behind the mirage is gone)
            } else if (st[i] == GI_E_OVERRUN && (u64)segs[i].out_cap * 8u <= 0xffff0000ull && sym_used + GI_WINDOW + (u64)segs[i].out_cap * 8u <= sym_room) {
                // a segment that expands more than the room it was given (a run of one base, of one quality): once more with eight times the room
                GiSeg m = segs[i];
                m.out_cap *= 8u;
                sym_used += GI_WINDOW;
                m.out_off = sym_used;
                sym_used += m.out_cap;
                redo.push_back(m);
                redo_at.push_back(kept.size());
                kept.push_back(m);
                kst.push_back(GI_OK);
                klen.push_back(0);
                kend.push_back(0);
            } else {
                kept.push_back(segs[i]);
                kst.push_back(st[i]);
                klen.push_back(len[i]);
                kend.push_back(endb[i]);
            }
        }
        if (redo.empty()) break;
        const int64_t n_redo = (int64_t)redo.size();
        GI_TRY(hipMemcpy(g->d_segs, redo.data(), sizeof(GiSeg) * (size_t)n_redo, hipMemcpyHostToDevice));  // (the master copy goes back below)
        GI_TRY(hipMemset(g->d_q + 1, 0, sizeof(u64)));
        hipLaunchKernelGGL(gi_segment_kernel, dim3((unsigned)std::min<int64_t>((n_redo + GI_WAVES - 1) / GI_WAVES, wgs)), dim3(64 * GI_WAVES), 0, 0, d_in, in_len, g->d_segs,
                           n_redo, g->d_sym, g->d_status, g->d_len, g->d_end, g->d_q + 1);
        GI_TRY(hipGetLastError());
        std::vector<int32_t> rst((size_t)n_redo);
        std::vector<uint32_t> rlen((size_t)n_redo);
        std::vector<u64> rend((size_t)n_redo);
        GI_TRY(hipMemcpy(rst.data(), g->d_status, sizeof(int32_t) * (size_t)n_redo, hipMemcpyDeviceToHost));
        GI_TRY(hipMemcpy(rlen.data(), g->d_len, sizeof(uint32_t) * (size_t)n_redo, hipMemcpyDeviceToHost));
        GI_TRY(hipMemcpy(rend.data(), g->d_end, sizeof(u64) * (size_t)n_redo, hipMemcpyDeviceToHost));
        for (size_t r = 0; r < redo.size(); r++) {
            kst[redo_at[r]] = rst[r];
            klen[redo_at[r]] = rlen[r];
            kend[redo_at[r]] = rend[r];
        }
        segs.swap(kept);
        st.swap(kst);
        len.swap(klen);
        endb.swap(kend);
    }
    const int64_t n_fin = (int64_t)segs.size();
    GI_TRY(hipMemcpy(g->d_segs, segs.data(), sizeof(GiSeg) * (size_t)n_fin, hipMemcpyHostToDevice));
    GI_TRY(hipMemcpy(g->d_len, len.data(), sizeof(uint32_t) * (size_t)n_fin, hipMemcpyHostToDevice));
    std::vector<u64> off((size_t)n_fin + 1, 0);
    for (int64_t i = 0; i < n_fin; i++) {
        const int e = st[(size_t)i];
        if (e == GI_E_SYNC || e == GI_E_OVERRUN)
            return gi_fail(GS_E_UNSUPPORTED, "segment " + std::to_string(i) + " of " + std::to_string(n_fin) + (e == GI_E_SYNC ? " does not end where the next one starts" : " outgrows its buffer") + ": host decoders");
        if (e != GI_OK && i == 0 && g->n_batches == 0) return gi_fail(GS_E_INVALID, "corrupt gzip stream (inflate status " + std::to_string(e) + " in the first segment)");
        if (e != GI_OK) return gi_fail(GS_E_UNSUPPORTED, "segment " + std::to_string(i) + ": inflate status " + std::to_string(e) + " (a damaged stream or a false block start): host decoders");
        off[(size_t)i + 1] = off[(size_t)i] + len[(size_t)i];
    }
    const int64_t n_new = (int64_t)off[(size_t)n_fin];
    const bool final_batch = segs.back().to_final != 0;
    // 3. windows (chained to the batch before), 4. text behind the kept tail, 5. CRC-32 of the new text
    const uint32_t tile = 65536;
    const int64_t n_tiles = (n_new + tile - 1) / tile;
    if ((rc = gu_grow(&g->d_win, &g->win_cap, (size_t)n_fin * GI_WINDOW))) return rc;
    if ((rc = gu_grow(&g->d_win16, &g->win16_cap, (size_t)n_fin * GI_WINDOW))) return rc;
    if ((rc = gu_grow(&g->d_text, &g->text_cap, (size_t)(keep_tail + n_new) + 8192))) return rc;
    if ((rc = gu_grow(&g->d_crc, &g->crc_cap, (size_t)std::max<int64_t>(n_tiles, 1)))) return rc;
    GI_TRY(hipMemcpy(g->d_off, off.data(), sizeof(u64) * (size_t)n_fin, hipMemcpyHostToDevice));
    if (keep_tail > 0) GI_TRY(hipMemcpy(g->d_text, g->d_tail, (size_t)keep_tail, hipMemcpyDeviceToDevice));
    const uint8_t *win0 = g->have_prev ? g->d_prev : nullptr;
    hipLaunchKernelGGL(gi_window_prep_kernel, dim3((unsigned)n_fin), dim3(256), 0, 0, g->d_sym, g->d_segs, g->d_len, n_fin, g->d_win16);
    {
        const int64_t per_group = std::max<int64_t>(1, (n_fin + 127) / 128), n_groups = (n_fin + per_group - 1) / per_group;
        if (per_group > 1) hipLaunchKernelGGL(gi_win_compose_kernel, dim3((unsigned)n_groups), dim3(1024), 0, 0, g->d_win16, n_fin, per_group);
        hipLaunchKernelGGL(gi_win_groups_kernel, dim3(1), dim3(1024), 0, 0, g->d_win16, n_fin, per_group, g->d_win, win0);
        if (per_group > 1) hipLaunchKernelGGL(gi_win_apply_kernel, dim3((unsigned)n_fin), dim3(256), 0, 0, g->d_win16, n_fin, per_group, g->d_win, win0);
    }
    hipLaunchKernelGGL(gi_resolve_kernel, dim3(16, (unsigned)std::min<int64_t>(n_fin, 16384)), dim3(256), 0, 0, g->d_sym, g->d_segs, g->d_len, g->d_off, n_fin, g->d_win, win0,
                       g->d_text + keep_tail);
    GiCrcPow pw;
    for (int sidx = 0; sidx < 6; sidx++) pw.p[sidx] = gi_h_x_pow_8n((uint64_t)(tile / 64u) << sidx);
    if (n_tiles > 0) hipLaunchKernelGGL(gi_crc_kernel, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, 0, g->d_text + keep_tail, n_new, tile, g->d_crc, pw);
    GI_TRY(hipGetLastError());
    GI_TRY(hipMemcpy(g->d_prev, g->d_win + (size_t)(n_fin - 1) * GI_WINDOW, GI_WINDOW, hipMemcpyDeviceToDevice));  // (a short batch's window reaches into the one before: gi_win_* took it from win0)
    g->have_prev = true;
    std::vector<uint32_t> crc((size_t)std::max<int64_t>(n_tiles, 1), 0);
    if (n_tiles > 0) GI_TRY(hipMemcpy(crc.data(), g->d_crc, sizeof(uint32_t) * (size_t)n_tiles, hipMemcpyDeviceToHost));
    {   // the tiles' registers behind one another (Horner: one multiplication per tile), then behind the register so far:
        // R = R_before x^(8 n) + sum_t R_t x^(8 bytes behind tile t)
        uint32_t add = 0;
        const uint32_t x_tile = gi_h_x_pow_8n(tile);
        for (int64_t t = 0; t < n_tiles; t++)
            add = gi_h_gf_mul(add, t == n_tiles - 1 ? gi_h_x_pow_8n((uint64_t)(n_new - t * (int64_t)tile)) : x_tile) ^ crc[(size_t)t];
        g->raw = gi_h_gf_mul(g->raw, gi_h_x_pow_8n((uint64_t)n_new)) ^ add;
    }
    g->total += (u64)n_new;
    g->n_batches++;
    if (gi_trace()) {
        uint32_t longest = 0;
        std::vector<uint32_t> sorted_len(len.begin(), len.end());
        std::sort(sorted_len.begin(), sorted_len.end());
        for (uint32_t v : len) longest = std::max(longest, v);
        fprintf(stderr, "  segments' text: shortest %u, 10 %% %u, median %u, 90 %% %u, longest %u bytes\n", sorted_len.front(), sorted_len[sorted_len.size() / 10], sorted_len[sorted_len.size() / 2],
                sorted_len[sorted_len.size() * 9 / 10], longest);
    }
    if (gi_trace())
        fprintf(stderr, "gunzip batch %lld: %lld bytes up %.2f ms, find %.2f (%zu starts in %lld chunks), %lld segments %.2f, windows + text + CRC %.2f: %lld bytes of text\n",
                (long long)g->n_batches, (long long)span, t_up - t_0, t_find - t_up, n_cand, (long long)n_chunks, (long long)n_fin, t_seg - t_find, gi_now_ms() - t_seg, (long long)n_new);
    g->n_segments += n_fin;
    g->n_chunks += n_chunks;
    g->last_n_text = keep_tail + n_new;
    *d_text_out = g->d_text;
    *n_text = keep_tail + n_new;
    if (final_batch) {
        g->done = true;
        // the trailer behind the final block's last byte; whatever follows it is another member
        const u64 end_byte = base + (endb.back() + 7u) / 8u;
        if ((int64_t)end_byte + 8 > g->in_len) return gi_fail(GS_E_INVALID, "corrupt gzip stream: no trailer behind the final block");
        const uint8_t *t = g->gz + g->hdr + end_byte;
        const uint32_t want_crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        const uint32_t want_isize = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
        if ((uint32_t)g->total != want_isize) return gi_fail(GS_E_INVALID, "corrupt gzip stream: ISIZE does not match the inflated text");
        if ((g->raw ^ 0xffffffffu) != want_crc && !getenv("GS_GUNZIP_DEBUG_KEEP_BAD_CRC"))  // (tools/gunzip_debug.py: the text as it came out)
            return gi_fail(GS_E_INVALID, "corrupt gzip stream: CRC-32 of the inflated text does not match");
        // what follows the trailer: another member (cat a.gz b.gz: its text simply follows), or nothing a gzip reader takes (ignored,
        // as java.util.zip.GZIPInputStream does)
        const int64_t nxt = (int64_t)end_byte + 8;
        const int64_t h2 = nxt < g->in_len ? gi_gzip_header(g->gz + g->hdr + nxt, g->in_len - nxt) : -1;
        if (h2 >= 0) {
            g->done = false;
            g->bit = (u64)(nxt + h2) * 8u;
            g->have_prev = false;
            g->raw = 0xffffffffu;
            g->total = 0;
            g->n_members++;
            *last = 0;
        } else
            *last = 1;
    } else {
        g->bit = base * 8u + segs.back().stop_bit;
    }
    return GS_OK;
}

extern "C" int gs_gunzipper_info(const gs_gunzipper *g, int64_t info[4]) {
    if (!g || !info) return gi_fail(GS_E_INVALID, "NULL argument");
    info[0] = g->n_segments;
    info[1] = g->n_chunks;
    info[2] = g->n_batches;
    info[3] = g->n_mirages;
    return GS_OK;
}

// the whole stream into one device buffer (tests, tools; a file the host pipeline takes goes batch by batch)
extern "C" int gs_gunzip_plan_device(int device, const uint8_t *gz, int64_t n, uint8_t **d_text_out, int64_t *n_text, int64_t info[4]) {
    if (!d_text_out || !n_text) return gi_fail(GS_E_INVALID, "bad argument");
    *d_text_out = nullptr;
    *n_text = 0;
    gs_gunzipper *g = nullptr;
    int rc = gs_gunzipper_open(&g, device, gz, n);
    if (rc) return rc;
    uint8_t *all = nullptr;
    size_t cap = 0, have = 0;
    for (;;) {
        const uint8_t *t = nullptr;
        int64_t nt = 0;
        int last = 0;
        rc = gs_gunzipper_next(g, 0, &t, &nt, &last);
        if (!rc && have + (size_t)nt + 8192 > cap) {
            uint8_t *bigger = nullptr;
            const size_t want = (have + (size_t)nt) * (last ? 1 : 2) + 8192;
            if (hipMalloc((void **)&bigger, want) != hipSuccess) rc = gi_fail(GS_E_NOMEM, "gs_gunzip_plan_device");
            if (!rc && have && hipMemcpy(bigger, all, have, hipMemcpyDeviceToDevice) != hipSuccess) rc = gi_fail(GS_E_HIP, "gs_gunzip_plan_device");
            if (!rc) {
                hipFree(all);
                all = bigger;
                cap = want;
            } else
                hipFree(bigger);
        }
        if (!rc && nt > 0 && hipMemcpy(all + have, t, (size_t)nt, hipMemcpyDeviceToDevice) != hipSuccess) rc = gi_fail(GS_E_HIP, "gs_gunzip_plan_device");
        if (rc) break;
        have += (size_t)nt;
        if (last) break;
    }
    if (!rc && info) gs_gunzipper_info(g, info);
    gs_gunzipper_close(g);
    if (rc) {
        hipFree(all);
        return rc;
    }
    *d_text_out = all;
    *n_text = (int64_t)have;
    return GS_OK;
}

extern "C" int gs_gunzip_free(int device, uint8_t *d_text) {
    if (!d_text) return GS_OK;
    GI_TRY(hipSetDevice(device));
    GI_TRY(hipFree(d_text));
    return GS_OK;
}

extern "C" int gs_gunzip_device(int device, const uint8_t *gz, int64_t n, uint8_t *out, int64_t out_cap, int64_t *n_text, int64_t info[4]) {
    if (!n_text) return gi_fail(GS_E_INVALID, "NULL argument");
    uint8_t *d_text = nullptr;
    int rc = gs_gunzip_plan_device(device, gz, n, &d_text, n_text, info);
    if (rc) return rc;
    if (*n_text > out_cap || (*n_text > 0 && !out)) {
        hipFree(d_text);
        return gi_fail(GS_E_INVALID, "output buffer too small");
    }
    const hipError_t e = *n_text > 0 ? hipMemcpy(out, d_text, (size_t)*n_text, hipMemcpyDeviceToHost) : hipSuccess;
    hipFree(d_text);
    if (e != hipSuccess) return gi_fail(GS_E_HIP, std::string("gs_gunzip_device: ") + hipGetErrorString(e));
    return GS_OK;
}

extern "C" int gs_device_fetch(int device, const uint8_t *d_src, uint8_t *out, int64_t n) {
    if (n < 0 || (n > 0 && (!d_src || !out))) return gi_fail(GS_E_INVALID, "bad argument");
    if (n == 0) return GS_OK;
    GI_TRY(hipSetDevice(device));
    GI_TRY(hipMemcpy(out, d_src, (size_t)n, hipMemcpyDeviceToHost));
    return GS_OK;
}

extern "C" int gs_text_cut_device(int device, const uint8_t *d_text, int64_t n, int64_t *n_lines, int64_t *cut) {
    if (!n_lines || !cut || n < 0 || (n > 0 && !d_text)) return gi_fail(GS_E_INVALID, "bad argument");
    *n_lines = 0;
    *cut = 0;
    if (n == 0) return GS_OK;
    GI_TRY(hipSetDevice(device));
    const int64_t n_tiles = (n + 4095) / 4096;
    GiDevBufs bufs;
    uint32_t *d_tiles = nullptr;
    u64 *d_cut = nullptr;
    hipError_t e = bufs.get(&d_tiles, sizeof(uint32_t) * (size_t)n_tiles);
    if (e == hipSuccess) e = bufs.get(&d_cut, 4 * sizeof(u64));
    if (e != hipSuccess) return gi_fail(GS_E_NOMEM, "gs_text_cut_device");
    hipLaunchKernelGGL(gi_count_kernel, dim3((unsigned)n_tiles), dim3(256), 0, 0, d_text, n, d_tiles);
    hipLaunchKernelGGL(gi_cut_kernel, dim3(1), dim3(1024), 0, 0, d_text, n, d_tiles, n_tiles, d_cut);
    u64 h[2] = {0, 0};
    GI_TRY(hipMemcpy(h, d_cut, sizeof(h), hipMemcpyDeviceToHost));
    *n_lines = (int64_t)h[0] & ~(int64_t)3;
    *cut = (int64_t)h[1];
    return GS_OK;
}

// one-shot form (tests, tools): inflate the members of a compressed buffer on the host into a host buffer
extern "C" int gs_inflate_members(int device, const uint8_t *file, const gs_inflate_member *members, int64_t n_members, uint8_t *out,
                                  int64_t out_cap, int32_t *status) {
    if (n_members < 0 || (n_members > 0 && (!file || !members || !out))) return gi_fail(GS_E_INVALID, "bad argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return gi_fail(GS_E_NODEVICE, "no usable gfx950 device");
    if (n_members == 0) return GS_OK;
    GI_TRY(hipSetDevice(device));
    int rc = gi_upload_crc_table();
    if (rc) return rc;
    int64_t lo = members[0].payload_offset & ~(int64_t)3, hi = 0, total = 0;
    std::vector<GiBlock> hb((size_t)n_members);
    for (int64_t i = 0; i < n_members; i++) {
        if (members[i].payload_len > (1u << 20) || members[i].isize > (1u << 16) || members[i].payload_offset < lo)
            return gi_fail(GS_E_INVALID, "a BGZF member holds at most 64 KiB of text; members in file order");
        hb[(size_t)i] = GiBlock{(u64)(members[i].payload_offset - lo), members[i].payload_len, members[i].isize, (u64)total, members[i].crc32, 0};
        total += members[i].isize;
        hi = std::max<int64_t>(hi, members[i].payload_offset + (int64_t)members[i].payload_len);
    }
    if (total > out_cap) return gi_fail(GS_E_INVALID, "output buffer too small");
    uint8_t *d_comp = nullptr, *d_out = nullptr;
    GiBlock *d_b = nullptr;
    int32_t *d_s = nullptr;
    unsigned long long *d_q = nullptr;
    hipError_t e = hipMalloc((void **)&d_comp, (size_t)(hi - lo) + 1024);
    if (e == hipSuccess) e = hipMalloc((void **)&d_q, sizeof(u64));
    if (e == hipSuccess) e = hipMemset(d_q, 0, sizeof(u64));
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, (size_t)total + 64);
    if (e == hipSuccess) e = hipMalloc((void **)&d_b, sizeof(GiBlock) * (size_t)n_members);
    if (e == hipSuccess) e = hipMalloc((void **)&d_s, sizeof(int32_t) * (size_t)n_members);
    if (e == hipSuccess) e = hipMemset(d_comp, 0, (size_t)(hi - lo) + 1024);
    if (e == hipSuccess) e = hipMemcpy(d_comp, file + lo, (size_t)(hi - lo), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_b, hb.data(), sizeof(GiBlock) * (size_t)n_members, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipDeviceProp_t prop;
        const int n_cu = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        const int grid = (int)std::min<int64_t>((n_members + GI_WAVES - 1) / GI_WAVES, (int64_t)n_cu * gi_wgs_per_cu());
        hipLaunchKernelGGL(gi_inflate_kernel, dim3(grid), dim3(64 * GI_WAVES), 0, 0, d_comp, d_b, n_members, d_out, d_s, gi_force_slow(), d_q);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    std::vector<int32_t> st((size_t)n_members, -1);
    if (e == hipSuccess) e = hipMemcpy(st.data(), d_s, sizeof(int32_t) * (size_t)n_members, hipMemcpyDeviceToHost);
    if (e == hipSuccess && total > 0) e = hipMemcpy(out, d_out, (size_t)total, hipMemcpyDeviceToHost);
    hipFree(d_comp);
    hipFree(d_out);
    hipFree(d_b);
    hipFree(d_s);
    hipFree(d_q);
    if (e != hipSuccess) return gi_fail(GS_E_HIP, std::string("gs_inflate_members: ") + hipGetErrorString(e));
    int bad = 0;
    for (int64_t i = 0; i < n_members; i++) {
        if (status) status[i] = st[(size_t)i];
        bad += st[(size_t)i] != GI_OK;
    }
    return bad ? gi_fail(GS_E_INVALID, std::to_string(bad) + " member(s) did not inflate to their ISIZE / CRC-32") : GS_OK;
}
