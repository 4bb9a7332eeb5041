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

class GsInflate {
public:
    enum Status { NEED_OUTPUT = 0, DONE = 1, CORRUPT = -1 };
    struct MemberEnd {
        uint64_t out_offset;  // total output bytes when the member ended
        uint32_t crc;         // CRC-32 of the member's data according to its trailer
    };

    void init(const uint8_t *in, size_t n_in, bool check_crc = true) {
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
    Status decode(uint8_t *out, size_t cap, size_t history, size_t *produced) {
        uint8_t *o = out, *const o_end = out + cap;
        uint8_t *const out0 = out;
        const uint8_t *const o_min = out - history;
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
                memcpy(o, in_, n);
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
                ends_[n_ends_++] = {total_out_ + (uint64_t)(o - out0), pend_crc_};
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
                if ((check_crc_ && crc != crc_) || isz != isize_) return fail_corrupt(out0, o, produced);
                if (!check_crc_) {
                    if (n_ends_ == MAX_ENDS) {  // the caller has to drain the list first: stop at this member boundary
                        // put the trailer back and report the block as full
                        state_ = S_TRAILER_PENDING;
                        pend_crc_ = crc;
                        break;
                    }
                    ends_[n_ends_++] = {total_out_ + (uint64_t)(o - out0), crc};
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
    uint32_t crc_ = 0, isize_ = 0, pend_crc_ = 0;
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
    void flush_crc(const uint8_t *from, const uint8_t *to) {
        size_t n = (size_t)(to - from);
        isize_ += (uint32_t)n;
        if (!check_crc_) return;
        while (n) {
            const uInt c = n > 0x40000000u ? 0x40000000u : (uInt)n;
            crc_ = (uint32_t)crc32(crc_, from, c);
            from += c;
            n -= c;
        }
    }
    Status fail_corrupt(uint8_t *out, uint8_t *o, size_t *produced) {
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
    __attribute__((optimize("O3"))) int decode_codes(uint8_t *&o_ref, uint8_t *o_end, const uint8_t *o_min) {
        uint8_t *o = o_ref;
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
        const uint16_t two = (uint16_t)(e >> 16);           \
        memcpy(o, &two, 2);                                 \
        o += 2;                                             \
    } else                                                  \
        *o++ = (uint8_t)(e >> 16);
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
                    *o++ = (uint8_t)(e >> 16);
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
                const uint8_t *s = o - dist;
                uint8_t *const e_out = o + len;
                if (dist >= 8) {  // word-wise; may write up to 7 bytes past the match (room is guaranteed)
                    do {
                        uint64_t w;
                        memcpy(&w, s, 8);
                        memcpy(o, &w, 8);
                        s += 8;
                        o += 8;
                    } while (o < e_out);
                } else if (dist == 1) {
                    memset(o, *s, len);
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
                *o++ = (uint8_t)(e >> 16);
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
