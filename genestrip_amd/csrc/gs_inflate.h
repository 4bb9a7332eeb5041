// gs_inflate.h -- resumable gzip/DEFLATE decoder for the ingest path (RFC 1951 / RFC 1952), written for throughput:
// 64-bit bit buffer refilled eight bytes at a time, one table lookup per literal/length symbol (11-bit primary table
// with sub-tables for the long codes), word-wise match copies.  The reference reads gzip through
// java.util.zip.GZIPInputStream (B/io/StreamProvider.java:92-100,148-150); FASTQ is almost always gzip-compressed,
// and a general-purpose zlib inflate on one thread (~0.5 GB/s of text) is what bounds the whole file pipeline.
//
// The compressed input is one contiguous range (the file is mapped); the output is delivered in caller-sized blocks.
// A block may end anywhere -- inside a DEFLATE block, inside a match -- and decode() continues from there on the next
// call.  Back-references reach at most 32 KiB back: the caller keeps the previous 32 KiB of output directly in front
// of the new block (`history` bytes are readable before `out`).
//
// Integrity: the ISIZE trailer of every member is checked here; the CRC-32 is either checked here as well (init(...,
// true), zlib's crc32()) or left to the consumer of the blocks (init(..., false)): member_ends() then lists, per
// member, the output offset where it ended and the CRC-32 its trailer announced, so that another thread can run the
// checksum over the delivered blocks while this one keeps decoding (the CRC costs a third of the decode time).
// GZIPInputStream checks both.  Concatenated members are decoded one after the other (RFC 1952 2.2).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <zlib.h>

// CRC-32 (gzip polynomial), sixteen table look-ups per 16 input bytes: ~1.5x zlib 1.2.11's crc32()
class GsCrc32 {
public:
    static uint32_t update(uint32_t crc, const uint8_t *p, size_t n) {
        static const Tables t;
        crc = ~crc;
        while (n >= 16) {
            uint64_t a, b;
            memcpy(&a, p, 8);
            memcpy(&b, p + 8, 8);
            a ^= crc;
            crc = t.T[15][a & 255] ^ t.T[14][(a >> 8) & 255] ^ t.T[13][(a >> 16) & 255] ^ t.T[12][(a >> 24) & 255] ^
                  t.T[11][(a >> 32) & 255] ^ t.T[10][(a >> 40) & 255] ^ t.T[9][(a >> 48) & 255] ^ t.T[8][a >> 56] ^
                  t.T[7][b & 255] ^ t.T[6][(b >> 8) & 255] ^ t.T[5][(b >> 16) & 255] ^ t.T[4][(b >> 24) & 255] ^
                  t.T[3][(b >> 32) & 255] ^ t.T[2][(b >> 40) & 255] ^ t.T[1][(b >> 48) & 255] ^ t.T[0][b >> 56];
            p += 16;
            n -= 16;
        }
        while (n--) crc = (crc >> 8) ^ t.T[0][(crc ^ *p++) & 255];
        return ~crc;
    }

private:
    struct Tables {
        uint32_t T[16][256];
        Tables() {
            for (uint32_t i = 0; i < 256; i++) {
                uint32_t c = i;
                for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1)));
                T[0][i] = c;
            }
            for (int t = 1; t < 16; t++)
                for (int i = 0; i < 256; i++) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 255];
        }
    };
};

// OutT = uint8_t: the decoded bytes.  OutT = uint16_t: "marker mode" for decoding that starts in the middle of a stream
// (GsParallelGunzip below): values < 256 are bytes, values >= 0x8000 stand for byte (v - 0x8000) of the 32 KiB of output
// that precede the start and are not known yet -- the caller puts these 32768 marker values in front of the buffer, so
// that back-references into the unknown window simply copy them.
template <typename OutT>
class GsInflateT {
public:
    enum Status { NEED_OUTPUT = 0, DONE = 1, AT_BOUNDARY = 2, CORRUPT = -1 };
    struct MemberEnd {
        uint64_t out_offset;  // total output bytes when the member ended
        uint32_t crc;         // CRC-32 of the member's data according to its trailer
        uint32_t isize;       // ... and its length mod 2^32
    };

    // start INSIDE a stream, at the first bit of a DEFLATE block header (GsParallelGunzip): no gzip header is read, the
    // size of the member that is in progress cannot be checked here (its end is reported with the others)
    void init_at(const uint8_t *in, size_t n_in, uint64_t bit_offset) {
        init(in, n_in, false);
        in_ = in + (bit_offset >> 3);
        refill();
        const int r = (int)(bit_offset & 7);
        bitbuf_ >>= r;
        bitcnt_ -= r;
        state_ = S_BLOCK_HEADER;
        partial_member_ = true;
        any_member_ = true;
    }
    // bits of the input consumed so far
    uint64_t bit_position() const { return (uint64_t)(in_ - in_begin_) * 8 - (uint64_t)bitcnt_; }
    // decode() returns AT_BOUNDARY when it stands in front of a non-final dynamic-Huffman block at or behind this bit
    // (the kind of block start a decoder that begins mid-stream can recognise)
    void stop_at_boundary_from(uint64_t bit_limit) { stop_limit_ = bit_limit; }

    void init(const uint8_t *in, size_t n_in, bool check_crc = true) {
        in_begin_ = in;
        stop_limit_ = ~0ULL;
        partial_member_ = false;
        check_crc_ = check_crc;
        total_out_ = 0;
        n_ends_ = 0;
        in_ = in;
        in_end_ = in + n_in;
        bitbuf_ = 0;
        bitcnt_ = 0;
        state_ = S_HEADER;
        final_ = false;
        pend_len_ = 0;
        pend_dist_ = 0;
        stored_left_ = 0;
        crc_ = 0;
        isize_ = 0;
        any_member_ = false;
    }

    // (check_crc = false) member ends since the last clear_member_ends(); decode() reports a full block when the list
    // is full, so the caller drains it after every call
    int n_member_ends() const { return n_ends_; }
    const MemberEnd *member_ends() const { return ends_; }
    void clear_member_ends() { n_ends_ = 0; }

    // fills out[0, cap) as far as the stream goes; *produced bytes were written.  `history` = number of valid bytes
    // directly in front of `out` (the tail of the previous block, at least min(32768, total output so far)).
    Status decode(OutT *out, size_t cap, size_t history, size_t *produced) {
        OutT *o = out, *const o_end = out + cap;
        OutT *const out0 = out;
        const OutT *const o_min = out - history;
        Status st = NEED_OUTPUT;
        for (;;) {
            if (state_ == S_HEADER) {
                align_to_byte();
                if (avail_bytes() == 0 && any_member_) {  // clean end after >= 1 member
                    st = DONE;
                    break;
                }
                if (!parse_header()) {
                    // trailing garbage after a complete member is ignored the way gzip tools do when it is not a header
                    st = any_member_ && !header_started_ ? DONE : CORRUPT;
                    break;
                }
                crc_ = (uint32_t)crc32(0L, Z_NULL, 0);
                isize_ = 0;
                state_ = S_BLOCK_HEADER;
                final_ = false;
            } else if (state_ == S_BLOCK_HEADER) {
                if (final_) {
                    state_ = S_TRAILER;
                    continue;
                }
                if (!need_bits(3)) return fail_corrupt(out0, o, produced);
                if (bit_position() >= stop_limit_ && (bitbuf_ & 7u) == 4u) {  // BFINAL = 0, BTYPE = 2
                    st = AT_BOUNDARY;
                    break;
                }
                final_ = take(1) != 0;
                const uint32_t type = take(2);
                if (type == 0) {
                    align_to_byte();
                    if (!need_bits(32)) return fail_corrupt(out0, o, produced);
                    const uint32_t len = take(16), nlen = take(16);
                    if ((len ^ nlen) != 0xffffu) return fail_corrupt(out0, o, produced);
                    stored_left_ = len;
                    state_ = S_STORED;
                } else if (type == 1) {
                    build_fixed();
                    state_ = S_CODES;
                } else if (type == 2) {
                    if (!read_dynamic()) return fail_corrupt(out0, o, produced);
                    state_ = S_CODES;
                } else
                    return fail_corrupt(out0, o, produced);
            } else if (state_ == S_STORED) {
                // the bit buffer is byte aligned: hand its whole bytes back to the input first
                unread_bit_buffer();
                size_t n = stored_left_;
                if (n > (size_t)(o_end - o)) n = (size_t)(o_end - o);
                if (n > (size_t)(in_end_ - in_)) return fail_corrupt(out0, o, produced);
                for (size_t q = 0; q < n; q++) o[q] = (OutT)in_[q];  // (uint8_t: a plain copy)
                o += n;
                in_ += n;
                stored_left_ -= (uint32_t)n;
                if (stored_left_ == 0)
                    state_ = S_BLOCK_HEADER;
                else
                    break;  // output full
            } else if (state_ == S_CODES) {
                const int r = decode_codes(o, o_end, o_min);
                if (r < 0) return fail_corrupt(out0, o, produced);
                if (r == 0) break;  // output full
                state_ = S_BLOCK_HEADER;
            } else if (state_ == S_TRAILER_PENDING) {
                if (n_ends_ == MAX_ENDS) break;
                ends_[n_ends_++] = {total_out_ + (uint64_t)(o - out0), pend_crc_, pend_isize_};
                any_member_ = true;
                state_ = S_HEADER;
            } else {  // S_TRAILER
                // account the output of this call before comparing
                flush_crc(out, o);
                out = o;
                align_to_byte();
                if (!need_bits(32)) return fail_corrupt(out0, o, produced);
                const uint32_t crc = take(32);
                if (!need_bits(32)) return fail_corrupt(out0, o, produced);
                const uint32_t isz = take(32);
                if ((check_crc_ && crc != crc_) || (!partial_member_ && isz != isize_)) return fail_corrupt(out0, o, produced);
                partial_member_ = false;
                pend_isize_ = isz;
                if (!check_crc_) {
                    if (n_ends_ == MAX_ENDS) {  // the caller has to drain the list first: stop at this member boundary
                        // put the trailer back and report the block as full
                        state_ = S_TRAILER_PENDING;
                        pend_crc_ = crc;
                        break;
                    }
                    ends_[n_ends_++] = {total_out_ + (uint64_t)(o - out0), crc, isz};
                }
                any_member_ = true;
                state_ = S_HEADER;
            }
        }
        flush_crc(out, o);
        *produced = (size_t)(o - out0);
        total_out_ += (uint64_t)(o - out0);
        return st;
    }

private:
    enum { S_HEADER, S_BLOCK_HEADER, S_STORED, S_CODES, S_TRAILER, S_TRAILER_PENDING };
    enum { MAX_ENDS = 4096 };
    enum { LL_BITS = 11, D_BITS = 8 };
    // table entry: bits 0-3 code length consumed by THIS lookup, bits 4-7 kind, bits 8-12 extra bit count,
    // bits 16-31 value (literal / length base / distance base / sub-table offset); kind: 0 literal, 1 length, 2 end of
    // block, 3 sub-table pointer (bits 8-12 = sub-table bits), 4 distance, 15 invalid
    typedef uint32_t Entry;
    static Entry mk(uint32_t len, uint32_t kind, uint32_t extra, uint32_t value) { return len | (kind << 4) | (extra << 8) | (value << 16); }

    const uint8_t *in_ = nullptr, *in_end_ = nullptr;
    uint64_t bitbuf_ = 0;
    int bitcnt_ = 0;
    int state_ = S_HEADER;
    bool final_ = false, any_member_ = false, header_started_ = false;
    uint32_t pend_len_ = 0, pend_dist_ = 0, stored_left_ = 0;
    uint32_t crc_ = 0, isize_ = 0, pend_crc_ = 0, pend_isize_ = 0;
    const uint8_t *in_begin_ = nullptr;
    uint64_t stop_limit_ = ~0ULL;
    bool partial_member_ = false;
    bool check_crc_ = true;
    uint64_t total_out_ = 0;
    int n_ends_ = 0;
    MemberEnd ends_[MAX_ENDS];
    // sub-table room: with exactly sized sub-tables the totals stay below zlib's ENOUGH bounds (852 / 592 entries for
    // smaller roots); build() refuses a table that would not fit rather than overrun
    Entry ll_[(1 << LL_BITS) + 2048], dd_[(1 << D_BITS) + 1024];
    uint8_t pmax_[1 << LL_BITS];
    Entry pair_[1 << LL_BITS];

    size_t avail_bytes() const { return (size_t)(in_end_ - in_) + (size_t)(bitcnt_ >> 3); }

    void refill() {
        if (in_end_ - in_ >= 8) {
            uint64_t w;
            memcpy(&w, in_, 8);
            bitbuf_ |= w << bitcnt_;
            const int take_bytes = (63 - bitcnt_) >> 3;
            in_ += take_bytes;
            bitcnt_ += take_bytes << 3;
        } else {
            while (bitcnt_ <= 56 && in_ < in_end_) {
                bitbuf_ |= (uint64_t)(*in_++) << bitcnt_;
                bitcnt_ += 8;
            }
        }
    }
    bool need_bits(int n) {
        if (bitcnt_ < n) refill();
        return bitcnt_ >= n;
    }
    uint32_t take(int n) {
        const uint32_t v = (uint32_t)(bitbuf_ & ((n == 32) ? 0xffffffffull : ((1ull << n) - 1)));
        bitbuf_ >>= n;
        bitcnt_ -= n;
        return v;
    }
    void align_to_byte() {
        const int drop = bitcnt_ & 7;
        bitbuf_ >>= drop;
        bitcnt_ -= drop;
    }
    void unread_bit_buffer() {  // byte aligned: give the buffered bytes back
        in_ -= bitcnt_ >> 3;
        bitbuf_ = 0;
        bitcnt_ = 0;
    }
    void flush_crc(const OutT *from_t, const OutT *to_t) {
        size_t n = (size_t)(to_t - from_t);
        isize_ += (uint32_t)n;
        if (!check_crc_ || sizeof(OutT) != 1) return;
        const uint8_t *from = reinterpret_cast<const uint8_t *>(from_t);
        while (n) {
            const uInt c = n > 0x40000000u ? 0x40000000u : (uInt)n;
            crc_ = (uint32_t)crc32(crc_, from, c);
            from += c;
            n -= c;
        }
    }
    Status fail_corrupt(OutT *out, OutT *o, size_t *produced) {
        *produced = (size_t)(o - out);
        return CORRUPT;
    }

    bool parse_header() {  // RFC 1952 2.3
        header_started_ = false;
        unread_bit_buffer();
        const uint8_t *p = in_;
        if (in_end_ - p < 10) return false;
        if (p[0] != 0x1f || p[1] != 0x8b) return false;
        header_started_ = true;
        if (p[2] != 8) return false;
        const int flg = p[3];
        p += 10;
        if (flg & 4) {  // FEXTRA
            if (in_end_ - p < 2) return false;
            const size_t xl = (size_t)p[0] | ((size_t)p[1] << 8);
            p += 2;
            if ((size_t)(in_end_ - p) < xl) return false;
            p += xl;
        }
        for (int bit = 8; bit <= 16; bit <<= 1)  // FNAME, FCOMMENT: zero-terminated
            if (flg & bit) {
                const void *z = memchr(p, 0, (size_t)(in_end_ - p));
                if (!z) return false;
                p = (const uint8_t *)z + 1;
            }
        if (flg & 2) {  // FHCRC
            if (in_end_ - p < 2) return false;
            p += 2;
        }
        in_ = p;
        return true;
    }

    // canonical Huffman decode table: `n` code lengths -> primary table of `tbits` bits + sub-tables
    // is_dist selects the symbol -> entry mapping.  Returns false for an over-subscribed or (non-trivially) incomplete code.
    bool build(const uint8_t *lens, int n, Entry *tab, int tbits, int tab_cap, bool is_dist, bool pair_literals = false) {
        if (pair_literals) memset(pair_, 0, sizeof(pair_));
        static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        int count[16] = {0};
        for (int i = 0; i < n; i++) count[lens[i]]++;
        count[0] = 0;
        int left = 1, used = 0;
        for (int l = 1; l <= 15; l++) {
            left = (left << 1) - count[l];
            if (left < 0) return false;
            used += count[l];
        }
        const Entry invalid = mk(1, 15, 0, 0);
        for (int i = 0; i < (1 << tbits); i++) tab[i] = invalid;
        if (used == 0) return true;  // no codes at all (a block without distances): any use is an error
        if (left > 0 && !(used == 1)) return false;  // incomplete, except the one-code case zlib also accepts
        uint32_t next[16];
        {
            uint32_t code = 0;
            for (int l = 1; l <= 15; l++) {
                code = (code + (uint32_t)count[l - 1]) << 1;
                next[l] = code;
            }
        }
        // pass 1: bit-reversed code of every symbol (DEFLATE packs codes MSB first, the tables are indexed LSB first)
        // and, per primary prefix, the longest code that shares it (= the size of its sub-table)
        uint16_t rev_of[288 + 32];
        uint8_t *pmax = pmax_;
        memset(pmax, 0, (size_t)1 << tbits);
        for (int sym = 0; sym < n; sym++) {
            const int l = lens[sym];
            if (!l) continue;
            const uint32_t code = next[l]++;
            uint32_t rev = 0;
            for (int b = 0; b < l; b++) rev |= ((code >> b) & 1u) << (l - 1 - b);
            rev_of[sym] = (uint16_t)rev;
            if (l > tbits) {
                uint8_t &m = pmax[rev & ((1u << tbits) - 1)];
                if (l > m) m = (uint8_t)l;
            }
        }
        int sub_next = 1 << tbits;
        for (int sym = 0; sym < n; sym++) {
            const int l = lens[sym];
            if (!l) continue;
            const uint32_t rev = rev_of[sym];
            Entry e;
            if (!is_dist) {
                if (sym < 256)
                    e = mk(0, 0, 0, (uint32_t)sym);
                else if (sym == 256)
                    e = mk(0, 2, 0, 0);
                else if (sym <= 285)
                    e = mk(0, 1, len_extra[sym - 257], len_base[sym - 257]);
                else
                    e = mk(0, 15, 0, 0);
            } else {
                e = sym < 30 ? mk(0, 4, dist_extra[sym], dist_base[sym]) : mk(0, 15, 0, 0);
            }
            if (l <= tbits) {
                e |= (uint32_t)l;
                for (uint32_t i = rev; i < (1u << tbits); i += 1u << l) tab[i] = e;
            } else {
                const uint32_t prefix = rev & ((1u << tbits) - 1);
                Entry &pe = tab[prefix];
                if (((pe >> 4) & 15) != 3) {
                    const int sbits = pmax[prefix] - tbits;
                    if (sub_next + (1 << sbits) > tab_cap) return false;
                    pe = mk((uint32_t)tbits, 3, (uint32_t)sbits, (uint32_t)sub_next);
                    for (int i = 0; i < (1 << sbits); i++) tab[sub_next + i] = invalid;
                    sub_next += 1 << sbits;
                }
                const int sbits = (int)((pe >> 8) & 31);
                const uint32_t base = pe >> 16;
                e |= (uint32_t)(l - tbits);
                for (uint32_t i = rev >> tbits; i < (1u << sbits); i += 1u << (l - tbits)) tab[base + i] = e;
            }
        }
        if (pair_literals) {
            // two literals in one look-up where both codes fit the primary index (FASTQ text: bases and the common
            // quality values have 2-4 bit codes): kind 5, value = first | second << 8, length = both code lengths
            const uint32_t n_prim = 1u << tbits;
            for (uint32_t i = 0; i < n_prim; i++) {
                const Entry e1 = tab[i];
                if ((e1 & 0xf0u) != 0) continue;
                const uint32_t l1 = e1 & 15;
                const Entry e2 = tab[(i >> l1) & (n_prim - 1)];  // valid only if its code lies inside the known bits
                if ((e2 & 0xf0u) != 0) continue;
                const uint32_t l2 = e2 & 15;
                if (l1 + l2 > (uint32_t)tbits) continue;
                pair_[i] = mk(l1 + l2, 5, l1, (e1 >> 16) | ((e2 >> 16) << 8));  // (extra field: length of the first code)
            }
            for (uint32_t i = 0; i < n_prim; i++)
                if (((pair_[i] >> 4) & 15) == 5) tab[i] = pair_[i];
        }
        return true;
    }

    void build_fixed() {
        uint8_t lens[288 + 32];
        int i = 0;
        for (; i < 144; i++) lens[i] = 8;
        for (; i < 256; i++) lens[i] = 9;
        for (; i < 280; i++) lens[i] = 7;
        for (; i < 288; i++) lens[i] = 8;
        build(lens, 288, ll_, LL_BITS, (int)(sizeof(ll_) / sizeof(ll_[0])), false, true);
        for (i = 0; i < 32; i++) lens[i] = 5;
        build(lens, 32, dd_, D_BITS, (int)(sizeof(dd_) / sizeof(dd_[0])), true);
    }

    bool read_dynamic() {
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        if (!need_bits(14)) return false;
        const int hlit = (int)take(5) + 257, hdist = (int)take(5) + 1, hclen = (int)take(4) + 4;
        if (hlit > 286 || hdist > 30) return false;
        uint8_t cl[19] = {0};
        for (int i = 0; i < hclen; i++) {
            if (!need_bits(3)) return false;
            cl[order[i]] = (uint8_t)take(3);
        }
        Entry ct[1 << 7];
        if (!build(cl, 19, ct, 7, 1 << 7, false)) return false;
        uint8_t lens[286 + 30];
        int i = 0;
        while (i < hlit + hdist) {
            if (!need_bits(7 + 7)) {
                if (bitcnt_ < 1) return false;  // fewer than 14 bits left in the whole input: go on carefully
            }
            const Entry e = ct[bitbuf_ & 127];
            const int l = (int)(e & 15);
            if (((e >> 4) & 15) == 15 || l > bitcnt_) return false;
            take(l);
            const uint32_t sym = e >> 16;  // literal kind: value = symbol (0..18)
            if (sym < 16) {
                lens[i++] = (uint8_t)sym;
            } else {
                int rep, val = 0;
                if (sym == 16) {
                    if (i == 0 || !need_bits(2)) return false;
                    val = lens[i - 1];
                    rep = 3 + (int)take(2);
                } else if (sym == 17) {
                    if (!need_bits(3)) return false;
                    rep = 3 + (int)take(3);
                } else {
                    if (!need_bits(7)) return false;
                    rep = 11 + (int)take(7);
                }
                if (i + rep > hlit + hdist) return false;
                while (rep--) lens[i++] = (uint8_t)val;
            }
        }
        if (lens[256] == 0) return false;  // no end-of-block code
        if (!build(lens, hlit, ll_, LL_BITS, (int)(sizeof(ll_) / sizeof(ll_[0])), false, true)) return false;
        if (!build(lens + hlit, hdist, dd_, D_BITS, (int)(sizeof(dd_) / sizeof(dd_[0])), true)) return false;
        return true;
    }

    // 1: end of block, 0: output full (state saved), -1: corrupt
    __attribute__((optimize("O3"))) int decode_codes(OutT *&o_ref, OutT *o_end, const OutT *o_min) {
        OutT *o = o_ref;
        // a match that did not fit the previous block
        if (pend_len_) {
            if ((size_t)(o - o_min) < pend_dist_) return -1;
            while (pend_len_ && o < o_end) {
                *o = *(o - pend_dist_);
                o++;
                pend_len_--;
            }
            if (pend_len_) {
                o_ref = o;
                return 0;
            }
        }
        for (;;) {
            // fast loop: room for a maximal match plus copy slop, and >= 8 input bytes for unconditional refills
            while (o_end - o >= 258 + 16 && in_end_ - in_ >= 16) {
                refill();  // >= 56 bits: litlen (15) + extra (5) + dist (15) + extra (13) = 48
                Entry e = ll_[bitbuf_ & ((1u << LL_BITS) - 1)];
                if ((e & 0xf0u) == 0 || (e & 0xf0u) == 0x50u) {  // literal or literal pair: three look-ups per refill (<= 45 bits)
#define GS_INF_LITERALS()                                   \
    bitbuf_ >>= (e & 15);                                   \
    bitcnt_ -= (int)(e & 15);                               \
    if ((e & 0xf0u) == 0x50u) {                             \
        o[0] = (OutT)((e >> 16) & 0xffu);                   \
        o[1] = (OutT)(e >> 24);                             \
        o += 2;                                             \
    } else                                                  \
        *o++ = (OutT)(e >> 16);
                    GS_INF_LITERALS()
                    e = ll_[bitbuf_ & ((1u << LL_BITS) - 1)];
                    if ((e & 0xf0u) != 0 && (e & 0xf0u) != 0x50u) goto not_literal;
                    GS_INF_LITERALS()
                    e = ll_[bitbuf_ & ((1u << LL_BITS) - 1)];
                    if ((e & 0xf0u) != 0 && (e & 0xf0u) != 0x50u) goto not_literal;
                    GS_INF_LITERALS()
#undef GS_INF_LITERALS
                    continue;
                }
            not_literal:
                if (bitcnt_ < 48) continue;  // (after literals) a match needs up to 48 bits: refill first
                if (((e >> 4) & 15) == 3) {
                    bitbuf_ >>= LL_BITS;
                    bitcnt_ -= LL_BITS;
                    e = ll_[(e >> 16) + (bitbuf_ & ((1u << ((e >> 8) & 31)) - 1))];
                }
                const uint32_t kind = (e >> 4) & 15;
                bitbuf_ >>= (e & 15);
                bitcnt_ -= (int)(e & 15);
                if (kind == 0) {
                    *o++ = (OutT)(e >> 16);
                    continue;
                }
                if (kind == 2) {
                    o_ref = o;
                    return 1;
                }
                if (kind != 1) return -1;
                const int xb = (int)((e >> 8) & 31);
                uint32_t len = (e >> 16) + (uint32_t)(bitbuf_ & ((1u << xb) - 1));
                bitbuf_ >>= xb;
                bitcnt_ -= xb;
                Entry d = dd_[bitbuf_ & ((1u << D_BITS) - 1)];
                if (((d >> 4) & 15) == 3) {
                    bitbuf_ >>= D_BITS;
                    bitcnt_ -= D_BITS;
                    d = dd_[(d >> 16) + (bitbuf_ & ((1u << ((d >> 8) & 31)) - 1))];
                }
                if (((d >> 4) & 15) != 4) return -1;
                bitbuf_ >>= (d & 15);
                bitcnt_ -= (int)(d & 15);
                const int db = (int)((d >> 8) & 31);
                const uint32_t dist = (d >> 16) + (uint32_t)(bitbuf_ & ((1u << db) - 1));
                bitbuf_ >>= db;
                bitcnt_ -= db;
                if ((size_t)(o - o_min) < dist) return -1;
                const OutT *s = o - dist;
                OutT *const e_out = o + len;
                constexpr uint32_t W = 8 / sizeof(OutT);  // elements per 8-byte word
                if (dist >= W) {  // word-wise; may write up to 7 bytes past the match (room is guaranteed)
                    do {
                        uint64_t w;
                        memcpy(&w, s, 8);
                        memcpy(o, &w, 8);
                        s += W;
                        o += W;
                    } while (o < e_out);
                } else if (dist == 1) {
                    const OutT c = *s;
                    for (uint32_t q = 0; q < len; q++) o[q] = c;
                } else {
                    do {
                        *o++ = *s++;
                    } while (o < e_out);
                }
                o = e_out;
            }
            // careful path: one symbol with every check
            if (!need_bits(1)) return -1;
            refill();
            Entry e = ll_[bitbuf_ & ((1u << LL_BITS) - 1)];
            int used = 0;
            if (((e >> 4) & 15) == 3) {
                if (bitcnt_ < LL_BITS) return -1;
                used = LL_BITS;
                e = ll_[(e >> 16) + ((bitbuf_ >> LL_BITS) & ((1u << ((e >> 8) & 31)) - 1))];
            }
            uint32_t kind = (e >> 4) & 15;
            if (kind == 5) {  // a literal pair: take only its first literal here
                e = mk((e >> 8) & 31, 0, 0, (e >> 16) & 0xffu);
                kind = 0;
            }
            used += (int)(e & 15);
            if (kind == 15 || used > bitcnt_) return -1;
            if (kind == 0) {
                if (o == o_end) {  // no room: leave the symbol in the bit buffer
                    o_ref = o;
                    return 0;
                }
                bitbuf_ >>= used;
                bitcnt_ -= used;
                *o++ = (OutT)(e >> 16);
                continue;
            }
            bitbuf_ >>= used;
            bitcnt_ -= used;
            if (kind == 2) {
                o_ref = o;
                return 1;
            }
            if (kind != 1) return -1;
            const int xb = (int)((e >> 8) & 31);
            if (!need_bits(xb)) return -1;
            uint32_t len = (e >> 16) + take(xb);
            if (!need_bits(1)) return -1;
            refill();
            Entry d = dd_[bitbuf_ & ((1u << D_BITS) - 1)];
            used = 0;
            if (((d >> 4) & 15) == 3) {
                if (bitcnt_ < D_BITS) return -1;
                used = D_BITS;
                d = dd_[(d >> 16) + ((bitbuf_ >> D_BITS) & ((1u << ((d >> 8) & 31)) - 1))];
            }
            used += (int)(d & 15);
            if (((d >> 4) & 15) != 4 || used > bitcnt_) return -1;
            bitbuf_ >>= used;
            bitcnt_ -= used;
            const int db = (int)((d >> 8) & 31);
            if (!need_bits(db)) return -1;
            const uint32_t dist = (d >> 16) + take(db);
            if ((size_t)(o - o_min) < dist) return -1;
            while (len && o < o_end) {
                *o = *(o - dist);
                o++;
                len--;
            }
            if (len) {  // the rest goes into the next block
                pend_len_ = len;
                pend_dist_ = dist;
                o_ref = o;
                return 0;
            }
        }
    }
};

typedef GsInflateT<uint8_t> GsInflate;

// ---------------------------------------------------------------------------------------------------------------------
// GsParallelGunzip -- one gzip stream inflated by several threads.
//
// A DEFLATE stream can only be decoded from its start: every block may copy from the 32 KiB before it.  But it can be
// decoded SPECULATIVELY from the middle: a worker looks for something that parses as the header of a non-final
// dynamic-Huffman block (complete code-length, literal and distance codes, an end-of-block code, and 64 Ki symbols that
// decode without an error), starts there and writes 16-bit symbols -- bytes, or "marker" values that stand for a byte of
// the unknown 32 KiB window.  When the chunk before it is complete, the window is known and the markers are replaced.
// A guess is never trusted: the resolver checks that every chunk starts exactly at the bit where the previous one
// stopped (a decoder stops at the first recognisable block start behind its chunk's end) and otherwise decodes the gap
// itself, so a wrong or missing guess costs time, not correctness.  (The scheme is the one of pugz / rapidgzip.)
// ---------------------------------------------------------------------------------------------------------------------
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

class GsParallelGunzip {
public:
    typedef GsInflateT<uint16_t> Dec;
    typedef Dec::MemberEnd MemberEnd;

    ~GsParallelGunzip() { stop(); }
    // diagnostics: chunks the sequencer had to decode itself / worker chunks it could not use
    size_t n_decoded_here = 0, n_dropped = 0, n_used = 0;

    void start(const uint8_t *in, size_t n_in, int n_threads, size_t chunk_bytes = (size_t)1 << 20) {
        in_ = in;
        n_in_ = n_in;
        chunk_ = chunk_bytes < 65536 ? 65536 : chunk_bytes;
        n_chunks_ = n_in ? (n_in + chunk_ - 1) / chunk_ : 1;
        chunks_.resize(n_chunks_);
        for (auto &c : chunks_) c.reset(new Chunk());
        next_claim_ = 0;
        consumed_ = 0;
        max_ahead_ = (size_t)n_threads * 2 + 2;
        stop_ = false;
        for (int t = 0; t < n_threads; t++) threads_.emplace_back([this] { worker(); });
        threads_.emplace_back([this] { sequencer(); });
    }

    void stop() {
        {
            std::lock_guard<std::mutex> l(m_);
            stop_ = true;
        }
        cv_.notify_all();
        q_cv_.notify_all();
        for (auto &c : chunks_) c->cv.notify_all();
        for (auto &t : threads_) t.join();
        threads_.clear();
        order_.clear();
        for (auto &b : pool_) free(b.first);
        pool_.clear();
    }

    // the next bytes of the decoded stream, in order: up to cap bytes into out.  CRC-32 and ISIZE of every member are
    // checked here (from per-chunk checksums the workers computed, combined in stream order).  *done: the stream is
    // complete; returns false on corruption.  ends (may be NULL): member ends inside these bytes, offsets relative to out.
    bool read(uint8_t *out, size_t cap, size_t *produced, std::vector<MemberEnd> *ends, bool *done) {
        size_t got = 0;
        *done = false;
        *produced = 0;
        while (got < cap) {
            if (!cur_) {
                std::unique_lock<std::mutex> l(q_m_);
                q_cv_.wait(l, [&] { return stop_flag() || !order_.empty() || seq_done_ || seq_error_; });
                if (order_.empty()) {
                    if (seq_error_) return false;
                    if (seq_done_) {
                        *done = true;
                        break;
                    }
                    return false;  // stopped
                }
                cur_item_ = std::move(order_.front());
                order_.pop_front();
                l.unlock();
                cur_ = cur_item_.c;
                {
                    std::unique_lock<std::mutex> lc(cur_->m);
                    cur_->cv.wait(lc, [&] { return cur_->bytes_ready || stop_flag(); });
                    if (!cur_->bytes_ready) return false;
                }
                cur_off_ = cur_end_ = 0;
                // member checks: the chunk's segment checksums, combined with what earlier chunks left open
                for (size_t k = 0; k < cur_->seg.size(); k++) {
                    run_crc_ = (uint32_t)crc32_combine(run_crc_, cur_->seg[k].second, (z_off_t)cur_->seg[k].first);
                    run_len_ += cur_->seg[k].first;
                    if (k < cur_->ends.size()) {
                        if (run_crc_ != cur_->ends[k].crc || (uint32_t)run_len_ != cur_->ends[k].isize) return false;
                        run_crc_ = 0;
                        run_len_ = 0;
                    }
                }
            }
            const size_t n = std::min(cur_->size - cur_off_, cap - got);
            memcpy(out + got, reinterpret_cast<const uint8_t *>(cur_->data + 32768) + cur_off_, n);
            while (cur_end_ < cur_->ends.size() && cur_->ends[cur_end_].out_offset <= cur_off_ + n) {
                if (ends) {
                    MemberEnd e = cur_->ends[cur_end_];
                    e.out_offset = got + (e.out_offset - cur_off_);
                    ends->push_back(e);
                }
                cur_end_++;
            }
            got += n;
            cur_off_ += n;
            if (cur_off_ == cur_->size) {
                release(cur_item_);
                cur_ = nullptr;
            }
        }
        *produced = got;
        if (!cur_ && !*done) {  // the stream may just have ended
            std::lock_guard<std::mutex> l(q_m_);
            if (order_.empty() && seq_done_ && !seq_error_) *done = true;
        }
        return true;
    }

private:
    struct Chunk {
        std::mutex m;
        std::condition_variable cv;
        bool ready = false, bytes_ready = false, found = false, corrupt = false, last = false;
        uint64_t start_bit = 0, end_bit = 0;
        uint16_t *data = nullptr;  // 32768 markers, then `size` symbols (malloc: grown without being cleared); after the
                                   // resolve task the `size` BYTES of the chunk sit at the start of the symbol area
        size_t cap = 0, size = 0;
        std::vector<MemberEnd> ends;
        std::vector<std::pair<uint64_t, uint32_t>> seg;  // (length, CRC-32) of the byte ranges between member ends
        uint8_t window[32768];                           // the 32 KiB in front of the chunk
        ~Chunk() { free(data); }
        void reserve(size_t n) {
            if (n <= cap) return;
            data = (uint16_t *)realloc(data, n * sizeof(uint16_t));
            cap = n;
        }
    };
    struct Item {
        Chunk *c = nullptr;
        std::unique_ptr<Chunk> own;  // a chunk the sequencer decoded itself (gap or wrong guess)
        size_t idx = (size_t)-1;     // index in chunks_ (worker chunk)
    };

    const uint8_t *in_ = nullptr;
    size_t n_in_ = 0, chunk_ = 0, n_chunks_ = 0;
    std::vector<std::unique_ptr<Chunk>> chunks_;
    std::vector<std::thread> threads_;
    std::mutex m_;  // claims, tasks, in-flight accounting
    std::condition_variable cv_;
    size_t next_claim_ = 0, consumed_ = 0, max_ahead_ = 4;
    size_t inflight_bytes_ = 0;  // symbol buffers of chunks that are decoded but not yet consumed
    std::deque<Chunk *> tasks_;  // chunks whose window is known: symbols -> bytes + checksums
    bool stop_ = false;
    // symbol buffers go back to a pool instead of to the OS: fresh pages cost a fault and a clear each
    std::mutex pool_m_;
    std::vector<std::pair<uint16_t *, size_t>> pool_;
    // sequencer -> reader
    std::mutex q_m_;
    std::condition_variable q_cv_;
    std::deque<Item> order_;
    bool seq_done_ = false, seq_error_ = false;
    // reader state
    Item cur_item_;
    Chunk *cur_ = nullptr;
    size_t cur_off_ = 0, cur_end_ = 0;
    uint32_t run_crc_ = 0;
    uint64_t run_len_ = 0;

    bool stop_flag() {
        std::lock_guard<std::mutex> l(m_);
        return stop_;
    }
    void take_buffer(Chunk &c) {
        std::lock_guard<std::mutex> l(pool_m_);
        if (!pool_.empty()) {
            c.data = pool_.back().first;
            c.cap = pool_.back().second;
            pool_.pop_back();
        }
    }
    void give_buffer(Chunk &c) {
        if (!c.data) return;
        std::lock_guard<std::mutex> l(pool_m_);
        pool_.push_back({c.data, c.cap});
        c.data = nullptr;
        c.cap = 0;
    }
    static void fill_markers(uint16_t *v) {
        for (int i = 0; i < 32768; i++) v[i] = (uint16_t)(0x8000 + i);
    }

    // decode from where `d` stands until it stops by itself (boundary behind stop_bit, end of stream, corruption)
    void run(Dec &d, Chunk &c, uint64_t stop_bit) {
        take_buffer(c);
        c.reserve(32768 + ((size_t)4 << 20));
        fill_markers(c.data);
        d.stop_at_boundary_from(stop_bit);
        size_t size = 0;
        for (;;) {
            if (c.cap - 32768 - size < ((size_t)1 << 20)) c.reserve(32768 + 2 * size + ((size_t)4 << 20));  // grow geometrically
            const size_t room = c.cap - 32768 - size;
            size_t p = 0;
            const Dec::Status st = d.decode(c.data + 32768 + size, room, 32768 + size, &p);
            for (int e = 0; e < d.n_member_ends(); e++) c.ends.push_back(d.member_ends()[e]);
            d.clear_member_ends();
            size += p;
            if (st == Dec::NEED_OUTPUT) continue;
            c.corrupt = st == Dec::CORRUPT;
            c.last = st == Dec::DONE;
            break;
        }
        c.size = size;
        c.end_bit = d.bit_position();
    }

    // first position >= from_bit that looks like the start of a non-final dynamic block
    bool find_start(uint64_t from_bit, uint64_t *found) const {
        std::unique_ptr<Dec> d(new Dec());
        std::vector<uint16_t> scratch(32768 + 65536);
        fill_markers(scratch.data());
        // inside the chunk only: a stretch without the start of a non-final dynamic block (stored data, or members that
        // are one final block each) is left to the sequencer instead of being searched to the end of the file
        const uint64_t end_bit = std::min((uint64_t)n_in_ * 8, from_bit + (uint64_t)chunk_ * 8 + 160);
        for (uint64_t p = from_bit; p + 160 < end_bit; p++) {
            // cheap tests first (a candidate costs a full header parse otherwise): BFINAL = 0 and BTYPE = 2, at most 286
            // literal/length and 30 distance codes, and a COMPLETE code-length code (Kraft sum exactly 1)
            const size_t byte = (size_t)(p >> 3);
            uint64_t w;
            memcpy(&w, in_ + byte, 8);
            w >>= (p & 7);  // >= 56 valid bits
            if ((w & 7u) != 4u) continue;
            if (((w >> 3) & 31u) > 29u || ((w >> 8) & 31u) > 29u) continue;
            const int hclen = (int)((w >> 13) & 15u) + 4;
            {
                // 3 bits per code length, hclen <= 19 of them from bit 17 of the header on = bit (p & 7) + 1 of byte + 2
                uint64_t v[2];
                memcpy(v, in_ + byte + 2, 16);
                const int sh = (int)(p & 7) + 1;
                const uint64_t lo = (v[0] >> sh) | (v[1] << (64 - sh));
                uint32_t kraft = 0;
                int used = 0;
                for (int q = 0; q < hclen; q++) {
                    const uint32_t len = (uint32_t)(lo >> (3 * q)) & 7u;
                    if (len) {
                        kraft += 128u >> len;
                        used++;
                    }
                }
                if (kraft != 128u && !(used == 1)) continue;
            }
            d->init_at(in_, n_in_, p);
            size_t prod = 0;
            const Dec::Status st = d->decode(scratch.data() + 32768, 65536, 32768, &prod);
            if (st == Dec::CORRUPT) continue;
            *found = p;
            return true;
        }
        return false;
    }

    static void build_lut(const uint8_t *window, uint8_t *lut) {
        for (int i = 0; i < 256; i++) lut[i] = (uint8_t)i;
        memset(lut + 256, 0, 0x8000 - 256);
        memcpy(lut + 0x8000, window, 32768);
    }

    // 16-bit symbols -> bytes, in place, plus the checksums of the ranges between member ends.  In FASTQ nearly every
    // record copies its header from the one before, so markers do not fade with the distance from the chunk start:
    // every symbol goes through a 64 Ki-entry table (bytes map to themselves, marker 0x8000 + i to byte i of the window).
    void resolve_task(Chunk &c) {
        std::unique_ptr<uint8_t[]> lut(new uint8_t[65536]);
        build_lut(c.window, lut.get());
        const uint16_t *src = c.data + 32768;
        uint8_t *dst = reinterpret_cast<uint8_t *>(c.data + 32768);  // dst[i] is written after src[i] was read
        const size_t n = c.size;
        size_t i = 0;
        for (; i + 8 <= n; i += 8) {
            const uint8_t b0 = lut[src[i]], b1 = lut[src[i + 1]], b2 = lut[src[i + 2]], b3 = lut[src[i + 3]];
            const uint8_t b4 = lut[src[i + 4]], b5 = lut[src[i + 5]], b6 = lut[src[i + 6]], b7 = lut[src[i + 7]];
            dst[i] = b0, dst[i + 1] = b1, dst[i + 2] = b2, dst[i + 3] = b3;
            dst[i + 4] = b4, dst[i + 5] = b5, dst[i + 6] = b6, dst[i + 7] = b7;
        }
        for (; i < n; i++) dst[i] = lut[src[i]];
        size_t at = 0;
        for (const MemberEnd &e : c.ends) {
            c.seg.push_back({e.out_offset - at, GsCrc32::update(0, dst + at, (size_t)e.out_offset - at)});
            at = (size_t)e.out_offset;
        }
        c.seg.push_back({n - at, GsCrc32::update(0, dst + at, n - at)});
        {
            std::lock_guard<std::mutex> l(c.m);
            c.bytes_ready = true;
        }
        c.cv.notify_all();
    }

    void worker() {
        std::unique_ptr<Dec> d(new Dec());
        for (;;) {
            Chunk *task = nullptr;
            size_t i = 0;
            {
                std::unique_lock<std::mutex> l(m_);
                // resolve tasks first; else decode a new chunk -- not too far ahead of the reader, neither in chunks
                // nor in bytes (highly compressible data can inflate a thousandfold: DEFLATE tops out at 1032:1)
                cv_.wait(l, [&] {
                    return stop_ || !tasks_.empty() ||
                           (next_claim_ < n_chunks_ && next_claim_ < consumed_ + max_ahead_ &&
                            (inflight_bytes_ < ((size_t)2 << 30) || next_claim_ == consumed_));
                });
                if (stop_) return;
                if (!tasks_.empty()) {
                    task = tasks_.front();
                    tasks_.pop_front();
                } else
                    i = next_claim_++;
            }
            if (task) {
                resolve_task(*task);
                continue;
            }
            Chunk &c = *chunks_[i];
            const uint64_t stop_bit = (uint64_t)(i + 1) * chunk_ * 8;
            if (i == 0) {
                d->init(in_, n_in_, false);
                c.found = true;
                c.start_bit = 0;
                run(*d, c, stop_bit);
            } else {
                uint64_t s = 0;
                if (find_start((uint64_t)i * chunk_ * 8, &s)) {
                    c.found = true;
                    c.start_bit = s;
                    d->init_at(in_, n_in_, s);
                    run(*d, c, stop_bit > s ? stop_bit : s + 1);
                }
            }
            {
                std::lock_guard<std::mutex> l(m_);
                inflight_bytes_ += c.size * sizeof(uint16_t);
            }
            {
                std::lock_guard<std::mutex> l(c.m);
                c.ready = true;
            }
            c.cv.notify_all();
        }
    }

    // puts the chunks into stream order: a worker's chunk is used if it starts exactly at the bit where the stream
    // stands, gaps and wrong guesses are decoded here; hands every chunk its window and queues its resolve task
    void sequencer() {
        uint64_t pos_bit = 0;
        bool first = true, finished = false, error = false;
        std::vector<uint8_t> tail(32768, 0), lut(65536);
        std::unique_ptr<Dec> d(new Dec());
        auto accept = [&](Item item) {
            Chunk &c = *item.c;
            memcpy(c.window, tail.data(), 32768);
            // the window of the NEXT chunk: the last 32 KiB of (window + this chunk's bytes)
            const size_t t = c.size < 32768 ? c.size : 32768;
            build_lut(c.window, lut.data());
            if (t < 32768) memmove(tail.data(), tail.data() + t, 32768 - t);
            const uint16_t *src = c.data + 32768 + (c.size - t);
            uint8_t *dst = tail.data() + (32768 - t);
            for (size_t q = 0; q < t; q++) dst[q] = lut[src[q]];
            pos_bit = c.end_bit;
            if (c.last) finished = true;
            Chunk *cp = item.c;
            {
                std::lock_guard<std::mutex> l(q_m_);
                order_.push_back(std::move(item));
            }
            q_cv_.notify_all();
            {
                std::lock_guard<std::mutex> l(m_);
                tasks_.push_back(cp);
            }
            cv_.notify_all();
        };
        auto decode_here = [&](uint64_t stop_bit) {
            n_decoded_here++;
            Item item;
            item.own.reset(new Chunk());
            item.c = item.own.get();
            if (first && pos_bit == 0)
                d->init(in_, n_in_, false);
            else
                d->init_at(in_, n_in_, pos_bit);
            run(*d, *item.c, stop_bit);
            first = false;
            if (item.c->corrupt) {
                error = true;
                return;
            }
            if (item.c->size == 0 && !item.c->last) {  // (cannot happen: a decoder that does not stop makes progress)
                pos_bit = item.c->end_bit;
                give_buffer(*item.c);
                return;
            }
            accept(std::move(item));
        };
        size_t idx = 0;
        while (!finished && !error) {
            if (stop_flag()) return;
            if (idx >= n_chunks_) {
                if (first && n_in_ == 0) {
                    error = true;
                    break;
                }
                decode_here(~0ULL);  // the worker chunks are used up but the stream has not ended
                continue;
            }
            Chunk &c = *chunks_[idx];
            {
                std::unique_lock<std::mutex> l(c.m);
                c.cv.wait(l, [&] { return c.ready || stop_flag_unlocked(); });
                if (!c.ready) return;
            }
            if (!c.found || c.start_bit < pos_bit || (first && idx != 0)) {  // nothing usable in this chunk
                n_dropped++;
                drop(idx++);
                continue;
            }
            if (c.start_bit > pos_bit) {  // a gap in front of it: decode up to its start here, then look again
                decode_here(c.start_bit);
                continue;
            }
            if (c.corrupt) {
                error = true;
                break;
            }
            n_used++;
            first = false;
            Item item;
            item.c = &c;
            item.idx = idx++;
            accept(std::move(item));
        }
        {
            std::lock_guard<std::mutex> l(q_m_);
            seq_done_ = true;
            seq_error_ = error;
        }
        q_cv_.notify_all();
    }

    bool stop_flag_unlocked() {  // (called with a chunk's mutex held: m_ is a different lock)
        std::lock_guard<std::mutex> l(m_);
        return stop_;
    }

    void drop(size_t idx) {
        give_buffer(*chunks_[idx]);
        {
            std::lock_guard<std::mutex> l(m_);
            if (idx + 1 > consumed_) consumed_ = idx + 1;
            const size_t b = chunks_[idx]->size * sizeof(uint16_t);
            inflight_bytes_ = inflight_bytes_ > b ? inflight_bytes_ - b : 0;
        }
        cv_.notify_all();
    }

    void release(Item &item) {
        if (item.idx != (size_t)-1)
            drop(item.idx);
        else if (item.own) {
            give_buffer(*item.own);
            item.own.reset();
        }
        item.c = nullptr;
    }
};

// ---------------------------------------------------------------------------------------------------
// BGZF (SAM/BAM spec 4.1; bgzip, htslib, many sequencing pipelines): a gzip file whose members are blocks of at most
// 64 KiB of text and say how long they are -- an extra subfield 'B' 'C' holds the compressed size of the member --
// so the blocks of a buffer can be found without decoding and inflated independently: no speculative starts, no
// marker symbols, no second pass.  read() delivers the next bytes of the stream like GsParallelGunzip::read(), with
// the CRC-32 and ISIZE of every block checked by the thread that inflates it.  A member that is not a BGZF block ends
// this reader's part of the file: *done is set and rest_offset() says where the general decoder has to go on.
// ---------------------------------------------------------------------------------------------------
#include "gs_pool.h"

class GsBgzfReader {
public:
    // a BGZF block at in[o .. o + *block_len): gzip member with FEXTRA and a 'BC' subfield of two bytes (BSIZE)
    static bool block_at(const uint8_t *in, size_t n, size_t o, size_t *block_len, uint32_t *isize) {
        if (o > n || n - o < 28) return false;  // the empty block (end-of-file marker) has 28 bytes
        const uint8_t *p = in + o;
        if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 4) == 0) return false;
        const size_t xlen = (size_t)p[10] | ((size_t)p[11] << 8);
        if (n - o < 12 + xlen + 10) return false;
        size_t bsize = 0;
        bool found = false;
        for (size_t q = 12; q + 4 <= 12 + xlen;) {
            const size_t slen = (size_t)p[q + 2] | ((size_t)p[q + 3] << 8);
            if (p[q] == 'B' && p[q + 1] == 'C' && slen == 2 && q + 6 <= 12 + xlen) {
                bsize = (size_t)p[q + 4] | ((size_t)p[q + 5] << 8);
                found = true;
            }
            q += 4 + slen;
        }
        if (!found) return false;
        const size_t len = bsize + 1;
        if (len < 12 + xlen + 10 || len > n - o) return false;
        const uint8_t *t = p + len - 4;
        const uint32_t sz = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        if (sz > 65536) return false;
        *block_len = len;
        *isize = sz;
        return true;
    }
    static bool looks_like(const uint8_t *in, size_t n) {
        size_t len;
        uint32_t isize;
        return block_at(in, n, 0, &len, &isize);
    }

    GsBgzfReader(const uint8_t *in, size_t n_in, int threads) : in_(in), n_(n_in), pool_(threads), dec_((size_t)std::max(1, threads)) {}

    // first byte of the input this reader did not take (== n_in when the whole file was BGZF)
    size_t rest_offset() const { return pos_; }

    // up to cap bytes of the decoded stream into out; *done: this reader's part of the file is through (see
    // rest_offset()).  Returns false on corruption (a block that does not inflate to its ISIZE / CRC-32).
    bool read(uint8_t *out, size_t cap, size_t *produced, bool *done) {
        size_t got = 0;
        *done = false;
        if (spill_at_ < spill_.size()) {  // the rest of a block that did not fit the previous buffer
            const size_t m = std::min(cap, spill_.size() - spill_at_);
            memcpy(out, spill_.data() + spill_at_, m);
            spill_at_ += m;
            got = m;
        }
        tasks_.clear();
        size_t off = got;
        bool spilled = false;
        while (off < cap && !ended_) {
            size_t len;
            uint32_t isize;
            if (pos_ >= n_ || !block_at(in_, n_, pos_, &len, &isize)) {
                ended_ = true;
                break;
            }
            if (isize <= cap - off) {
                tasks_.push_back({pos_, len, out + off, isize});
                off += isize;
                pos_ += len;
            } else {  // does not fit any more: decoded on the side, handed out in two parts
                spill_.resize(isize);
                spill_at_ = 0;
                tasks_.push_back({pos_, len, spill_.data(), isize});
                pos_ += len;
                spilled = true;
                break;
            }
        }
        std::atomic<bool> ok{true};
        pool_.run((int64_t)tasks_.size(), [&](int t, int64_t lo, int64_t hi) {
            GsInflate &d = dec_[(size_t)t];
            for (int64_t i = lo; i < hi && ok.load(std::memory_order_relaxed); i++) {
                const Task &k = tasks_[(size_t)i];
                uint8_t dummy;
                size_t p = 0;
                d.init(in_ + k.pos, k.len, true);
                GsInflate::Status st = d.decode(k.isize ? k.out : &dummy, k.isize ? k.isize : 1, 0, &p);
                if (st == GsInflate::NEED_OUTPUT && p == k.isize && k.isize) {
                    // the text is complete, the end-of-block symbol and the trailer are still to come
                    size_t more = 0;
                    st = d.decode(&dummy, 1, 0, &more);
                    p += more;
                }
                if (st != GsInflate::DONE || p != k.isize) ok = false;
            }
        }, 4);
        if (!ok) return false;
        if (spilled) {
            const size_t m = std::min(cap - off, spill_.size());
            memcpy(out + off, spill_.data(), m);
            spill_at_ = m;
            off += m;
        }
        *produced = off;
        *done = ended_ && spill_at_ >= spill_.size();
        return true;
    }

private:
    struct Task {
        size_t pos, len;
        uint8_t *out;
        uint32_t isize;
    };
    const uint8_t *in_;
    size_t n_, pos_ = 0;
    GsRangePool pool_;
    std::vector<GsInflate> dec_;
    std::vector<Task> tasks_;
    std::vector<uint8_t> spill_;
    size_t spill_at_ = 0;
    bool ended_ = false;
};
