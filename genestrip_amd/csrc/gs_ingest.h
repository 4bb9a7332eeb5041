// gs_ingest.h -- the byte-level side of the host layer (include/gshost.h): file names, the line reader and record
// parser with the reference's exact semantics (C/fastq/AbstractFastqReader.java:288-438 over
// B/io/BufferedLineReader.java:114-182), and the block readers of the text path (parallel pread / one inflating thread
// into pooled pinned blocks).  Header-only, included by gs_host.cpp and gs_report.cpp.
#pragma once
#include "../../include/gshost.h"
#include "gs_inflate.h"
#include "gs_pool.h"

#include <errno.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <queue>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

namespace gs_host {


inline thread_local std::string g_host_err;
inline int hfail(int code, const std::string &m) {
    g_host_err = m;
    return code;
}

inline bool ends_with(const std::string &s, const char *suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}
inline bool is_gzip_name(const std::string &s) { return ends_with(s, ".gz") || ends_with(s, ".gzip"); }  // StreamProvider.java:148-150

inline bool is_fasta_name(const std::string &s) {  // FastqMapGoal.java:64,188-201
    static const char *suf[] = {"fasta", "fa", "fna", "fas", "fasta.gz", "fa.gz", "fna.gz", "fas.gz",
                                "fasta.gzip", "fa.gzip", "fna.gzip", "fas.gzip"};
    for (const char *x : suf)
        if (ends_with(s, x)) return true;
    return false;
}

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---------------------------------------------------------------------------------------------------
// byte source + line reader (BufferedLineReader.nextLine semantics: NUL bytes dropped, '\r' kept, the '\n' is part
// of the returned line; callers take "length - 1")
// ---------------------------------------------------------------------------------------------------
class LineReader {
public:
    bool open(const std::string &path, int64_t offset = 0) {
        gz_ = gzopen(path.c_str(), "rb");  // zlib reads plain files transparently, gzip by content
        if (!gz_) return false;
        gzbuffer(gz_, 1 << 20);
        buf_.resize(1 << 20);
        return offset == 0 || gzseek(gz_, (z_off_t)offset, SEEK_SET) == (z_off_t)offset;
    }
    void open_mem(const uint8_t *p, size_t n) {  // a byte range that is already in memory
        buf_.assign(p, p + n);
        fill_ = n;
        pos_ = 0;
        eof_ = true;
    }
    ~LineReader() {
        if (gz_) gzclose(gz_);
    }
    // appends the next line (incl. '\n', NULs dropped) to out; returns the number of bytes appended (0 at EOF)
    size_t next_line(std::vector<uint8_t> &out) {
        const size_t start = out.size();
        for (;;) {
            if (pos_ == fill_) {
                if (eof_) break;
                const int n = gzread(gz_, buf_.data(), (unsigned)buf_.size());
                if (n <= 0) {
                    eof_ = true;
                    break;
                }
                fill_ = (size_t)n;
                pos_ = 0;
            }
            const uint8_t *p = buf_.data() + pos_;
            const size_t avail = fill_ - pos_;
            const uint8_t *nl = (const uint8_t *)memchr(p, '\n', avail);
            const size_t take = nl ? (size_t)(nl - p) + 1 : avail;
            if (memchr(p, 0, take) == nullptr) {
                out.insert(out.end(), p, p + take);
            } else {
                for (size_t i = 0; i < take; i++)
                    if (p[i] != 0) out.push_back(p[i]);
            }
            pos_ += take;
            if (nl) break;
        }
        return out.size() - start;
    }

private:
    gzFile gz_ = nullptr;
    std::vector<uint8_t> buf_;
    size_t pos_ = 0, fill_ = 0;
    bool eof_ = false;
};

struct Batch {
    std::vector<uint8_t> seq, desc, qual;
    std::vector<uint64_t> seq_off{0}, desc_off{0}, qual_off{0};
    int64_t first_read_no = 0;
    bool has_qual = true;  // false for FASTA records (readProbsSize = -1, AbstractFastqReader.java:417)
    int64_t n() const { return (int64_t)seq_off.size() - 1; }
    void clear() {
        seq.clear();
        desc.clear();
        qual.clear();
        seq_off.assign(1, 0);
        desc_off.assign(1, 0);
        qual_off.assign(1, 0);
        has_qual = true;
    }
};

// AbstractFastqReader.doReadFastq (:288-368) / doReadFasta (:375-438) with unbounded buffers
class FastqParser {
public:
    FastqParser(int k, bool fasta) : k_(k), fasta_(fasta) {}
    bool open(const std::string &path, int64_t offset = 0) { return lr_.open(path, offset); }
    void open_mem(const uint8_t *p, size_t n) { lr_.open_mem(p, n); }

    // appends up to max_reads records / max_bytes sequence bytes to b; returns false at end of file
    bool parse(Batch &b, int64_t max_reads, int64_t max_bytes) {
        b.clear();
        b.first_read_no = reads_;
        while (!done_ && b.n() < max_reads && (int64_t)b.seq.size() < max_bytes) {
            if (!(fasta_ ? next_fasta(b) : next_fastq(b))) done_ = true;
        }
        return b.n() > 0;
    }
    int64_t reads_ = 0, kmers_ = 0, bps_ = 0;

private:
    void account(Batch &b, size_t read_size) {
        b.seq_off.push_back(b.seq.size());
        b.desc_off.push_back(b.desc.size());
        b.qual_off.push_back(b.qual.size());
        reads_++;
        if ((int64_t)read_size >= k_) kmers_ += (int64_t)read_size - k_ + 1;
        bps_ += (int64_t)read_size;
    }

    bool next_fastq(Batch &b) {
        const size_t dstart = b.desc.size(), sstart = b.seq.size(), qstart = b.qual.size();
        size_t got = lr_.next_line(b.desc);
        if (got == 0) return false;  // readDescriptorSize == -1
        b.desc.resize(dstart + got - 1);
        got = lr_.next_line(b.seq);
        if (got == 0) {  // truncated record: the reference runs into an exception here
            b.desc.resize(dstart);
            return false;
        }
        b.seq.resize(sstart + got - 1);
        for (;;) {  // sequence lines until a line STARTING with '+' (:301-308)
            const size_t lstart = b.seq.size();
            got = lr_.next_line(b.seq);
            if (got == 0) {
                b.desc.resize(dstart);
                b.seq.resize(sstart);
                return false;
            }
            if (b.seq[lstart] == '+') {
                b.seq.resize(lstart);
                break;
            }
            b.seq.resize(lstart + got - 1);
        }
        const long read_size = (long)(b.seq.size() - sstart);
        // quality lines until >= readSize characters (:320-341)
        got = lr_.next_line(b.qual);
        long qsize = (long)got - 1;
        while (qsize < read_size) {
            const long old = qsize;
            b.qual.resize(qstart + (size_t)(qsize < 0 ? 0 : qsize));  // continue over the previous '\n'
            got = lr_.next_line(b.qual);
            qsize = got ? (long)(b.qual.size() - qstart) - 1 : old - 1;
            if (qsize == old - 1) break;  // EOF
        }
        if (qsize < 0) qsize = 0;
        b.qual.resize(qstart + (size_t)qsize);
        account(b, (size_t)read_size);
        return true;
    }

    bool next_fasta(Batch &b) {
        b.has_qual = false;
        if (!have_header_) {
            header_.clear();
            const size_t got = lr_.next_line(header_);
            if (got == 0) return false;
            header_.resize(got - 1);
            have_header_ = true;
        }
        const size_t dstart = b.desc.size(), sstart = b.seq.size();
        b.desc.insert(b.desc.end(), header_.begin(), header_.end());
        if (!header_.empty()) b.desc[dstart] = '@';  // :380
        have_header_ = false;
        bool more = true;
        for (;;) {
            const size_t lstart = b.seq.size();
            const size_t got = lr_.next_line(b.seq);
            if (got == 0) {
                more = false;
                break;
            }
            if (b.seq[lstart] == '>') {  // next header: copied without its '\n' (:405-413)
                header_.assign(b.seq.begin() + (long)lstart, b.seq.begin() + (long)(lstart + got - 1));
                b.seq.resize(lstart);
                have_header_ = true;
                break;
            }
            b.seq.resize(lstart + got - 1);
        }
        account(b, b.seq.size() - sstart);
        return more || have_header_;
    }

    int k_;
    bool fasta_, done_ = false, have_header_ = false;
    std::vector<uint8_t> header_;
    LineReader lr_;
};

inline void append_int(std::string &s, long long v) {
    char buf[24];
    auto r = std::to_chars(buf, buf + sizeof(buf), v);
    s.append(buf, r.ptr);
}

// Double.toString: shortest round-trip digits; decimal notation for 1e-3 <= |d| < 1e7, else d.dddE[-]n
inline std::string java_double(double v) {
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v > 0 ? "Infinity" : "-Infinity";
    if (v == 0) return std::signbit(v) ? "-0.0" : "0.0";
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf), std::fabs(v), std::chars_format::scientific);
    std::string sci(buf, r.ptr);  // d[.ddd]e[+-]XX
    const size_t epos = sci.find('e');
    std::string digits = sci.substr(0, epos);
    const int exp10 = atoi(sci.c_str() + epos + 1);
    digits.erase(std::remove(digits.begin(), digits.end(), '.'), digits.end());
    std::string out = v < 0 ? "-" : "";
    const double a = std::fabs(v);
    if (a >= 1e-3 && a < 1e7) {
        if (exp10 >= 0) {
            std::string ip = digits.substr(0, std::min(digits.size(), (size_t)exp10 + 1));
            while ((int)ip.size() < exp10 + 1) ip.push_back('0');
            std::string fp = digits.size() > (size_t)exp10 + 1 ? digits.substr((size_t)exp10 + 1) : "0";
            out += ip + "." + fp;
        } else {
            out += "0." + std::string((size_t)(-exp10 - 1), '0') + digits;
        }
    } else {
        out += digits.substr(0, 1) + "." + (digits.size() > 1 ? digits.substr(1) : "0") + "E";
        append_int(out, exp10);
    }
    return out;
}


// ---- fast path for plain four-line FASTQ: reader threads fill pinned blocks, the device finds the records -----------
// (gs_match_submit_text).  The host only counts newlines to cut the stream at record boundaries.
// newlines in p[0, n): 16 bytes per step where SSE2 is there (every x86-64); the readers of the file pipelines run this
// over everything they read, so a byte-at-a-time loop (3 GB/s) would be what bounds them
inline size_t count_newlines(const uint8_t *p, size_t n) {
    size_t c = 0, i = 0;
#if defined(__SSE2__)
    const __m128i nl = _mm_set1_epi8('\n'), zero = _mm_setzero_si128();
    __m128i total = zero;
    while (n - i >= 16) {
        // byte counters take 255 steps before they could overflow; two independent chains
        size_t steps = (n - i) / 32;
        if (steps > 255) steps = 255;
        __m128i a = zero, b = zero;
        for (size_t s = 0; s < steps; s++, i += 32) {
            a = _mm_sub_epi8(a, _mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(p + i)), nl));
            b = _mm_sub_epi8(b, _mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(p + i + 16)), nl));
        }
        if (steps == 0) {
            a = _mm_sub_epi8(a, _mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(p + i)), nl));
            i += 16;
        }
        total = _mm_add_epi64(total, _mm_add_epi64(_mm_sad_epu8(a, zero), _mm_sad_epu8(b, zero)));
    }
    c = (size_t)_mm_cvtsi128_si64(total) + (size_t)_mm_cvtsi128_si64(_mm_unpackhi_epi64(total, total));
#endif
    for (; i < n; i++) c += p[i] == '\n';
    return c;
}

struct TextSlot {
    uint8_t *buf = nullptr;  // pinned: headroom (for the carried partial record) + block
    size_t n = 0;            // bytes read into the block
    int64_t newlines = 0;
    int64_t last4[4] = {-1, -1, -1, -1};  // offsets of the last four newlines of the block, last first
    int state = 0;           // 0 empty, 1 full, 2 decoded (gzip input: its newlines are still to be counted)
    bool eof = false, io_error = false;
    struct GzEnd {
        uint32_t offset, crc, isize;  // gzip input: a member ended at this offset of the block; CRC-32 / ISIZE of its trailer
    };
    std::vector<GzEnd> member_ends;
};

// page-locked blocks are expensive to create (the driver pins every page): the pipelines of one process reuse them
// from file to file.  At most 64 idle blocks are kept; they are deliberately not released at exit (the HIP runtime
// may already be gone when static destructors run).
class PinnedPool {
public:
    int get(size_t bytes, uint8_t **out) {
        {
            std::lock_guard<std::mutex> l(m_);
            for (size_t i = 0; i < idle_.size(); i++)
                if (idle_[i].second == bytes) {
                    *out = idle_[i].first;
                    idle_.erase(idle_.begin() + (long)i);
                    return GS_OK;
                }
        }
        void *p = nullptr;
        const int rc = gs_pinned_alloc(&p, bytes);
        *out = (uint8_t *)p;
        return rc;
    }
    void put(uint8_t *p, size_t bytes) {
        if (!p) return;
        {
            std::lock_guard<std::mutex> l(m_);
            if (idle_.size() < 64) {
                idle_.push_back({p, bytes});
                return;
            }
        }
        gs_pinned_free(p);
    }
    void release_all() {  // gs_host_release_pools
        std::lock_guard<std::mutex> l(m_);
        for (auto &x : idle_) gs_pinned_free(x.first);
        idle_.clear();
    }

private:
    std::mutex m_;
    std::vector<std::pair<uint8_t *, size_t>> idle_;
};
inline PinnedPool &pinned_pool() {
    static PinnedPool *pool = new PinnedPool();
    return *pool;
}

struct TextReader {
    int fd = -1;
    size_t block = 0, headroom = 0;
    int n_slots = 0, n_threads = 0;
    std::vector<TextSlot> slots;
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv;
    bool stop = false;
    int64_t start_off = 0;

    // gzip input: the file is mapped and ONE thread inflates it into the blocks, in order (a gzip stream is serial);
    // the CRC-32 of the members is left to the consumer of the blocks (verify_gzip), which has the time
    bool gz = false;
    const uint8_t *map = nullptr;
    size_t map_len = 0;
    uint32_t run_crc = 0;
    bool consumer_checks_crc = false;  // gzip input: verify_gzip() runs the CRC-32 over the delivered blocks
    uint64_t run_size = 0;
    int gz_threads_hint = 0;

    int open(const std::string &path, size_t block_bytes, int readers, bool gzip) {
        if (gzip) {
            gz = true;
            fd = ::open(path.c_str(), O_RDONLY);
            if (fd < 0) return hfail(GS_E_INVALID, "cannot open " + path);
            struct stat sb;
            if (fstat(fd, &sb) != 0) return hfail(GS_E_INVALID, "cannot stat " + path);
            map_len = (size_t)sb.st_size;
            if (map_len) {
                void *mapped = mmap(nullptr, map_len, PROT_READ, MAP_PRIVATE, fd, 0);
                if (mapped == MAP_FAILED) return hfail(GS_E_INVALID, "cannot map " + path);
                map = (const uint8_t *)mapped;
                madvise(mapped, map_len, MADV_SEQUENTIAL);
            }
            if (block_bytes < ((size_t)64 << 10)) block_bytes = (size_t)64 << 10;  // the 32 KiB window lives in the headroom
            gz_threads_hint = readers;  // inflating threads behind the ONE thread that fills the blocks
            readers = 1;
        } else {
            fd = ::open(path.c_str(), O_RDONLY);
            if (fd < 0) return hfail(GS_E_INVALID, "cannot open " + path);
#ifdef POSIX_FADV_SEQUENTIAL
            posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);
#endif
        }
        block = block_bytes;
        headroom = block_bytes;
        // small files: no more readers and pinned blocks than the file has blocks (pinning memory is not free)
        {
            struct stat sb2;
            const size_t fsize = fstat(fd, &sb2) == 0 ? (size_t)sb2.st_size : ~(size_t)0;
            const size_t est = gzip ? fsize * 8 : fsize;  // (gzip: a generous guess of the text size)
            const size_t n_blocks = est / block + 1;
            if (!gzip && (size_t)readers > n_blocks) readers = (int)n_blocks;
            n_threads = readers;
            // block i lives in slot i % n_slots and is read by thread i % n_threads: n_slots must stay a multiple of
            // n_threads, so that the blocks sharing a slot are filled by ONE thread, in order
            n_slots = gzip ? 4 : 2 * readers;
        }
        slots.resize((size_t)n_slots);  // (each filling thread pins its own blocks when it first needs them: in parallel)
        return GS_OK;
    }
    // the pinned block of a slot, allocated by the thread that fills it; false: out of (pinnable) memory
    bool slot_buffer(TextSlot &sl) {
        if (sl.buf) return true;
        return pinned_pool().get(headroom + block, &sl.buf) == GS_OK && sl.buf != nullptr;
    }
    void fill_newlines(TextSlot &sl, const uint8_t *dst, size_t got, GsRangePool *pool = nullptr) {
        if (pool && got >= ((size_t)1 << 20)) {  // (gzip input: ONE thread counts for all the inflating threads)
            std::atomic<int64_t> total{0};
            const int64_t pieces = (int64_t)((got + 262143) >> 18);
            pool->run(pieces, [&](int, int64_t lo, int64_t hi) {
                const size_t a = (size_t)lo << 18, b = std::min(got, (size_t)hi << 18);
                total += (int64_t)count_newlines(dst + a, b - a);
            }, 1);
            sl.newlines = total.load();
        } else
            sl.newlines = (int64_t)count_newlines(dst, got);
        size_t end = got;
        for (int j = 0; j < 4; j++) {
            const void *q = end ? memrchr(dst, '\n', end) : nullptr;
            sl.last4[j] = q ? (int64_t)((const uint8_t *)q - dst) : -1;
            end = q ? (size_t)((const uint8_t *)q - dst) : 0;
        }
    }
    // gzip input: the stream is inflated by GsParallelGunzip (several threads, speculative starts inside the stream)
    // when the machine has the cores, by one GsInflate otherwise; either way ONE thread hands the decoded bytes to
    // the blocks in order.  member_ends carry (offset in the block, CRC-32, ISIZE) for the consumer's check.
    void start_gzip() {
        int gz_threads = gz_threads_hint > 0 ? gz_threads_hint : (int)std::min<unsigned>(8, std::thread::hardware_concurrency() / 2);
        if (const char *e = getenv("GS_GZ_THREADS")) gz_threads = std::max(1, std::min(32, atoi(e)));
        bool use_bgzf = GsBgzfReader::looks_like(map, map_len);
        if (const char *e = getenv("GS_BGZF")) use_bgzf = use_bgzf && atoi(e) != 0;
        // only the one-thread decoder leaves the CRC-32 to the consumer of the blocks (verify_gzip); the multi-threaded
        // readers check it themselves, and running it again there would bound the whole pipeline by one core's CRC rate
        consumer_checks_crc = !use_bgzf && !(gz_threads >= 2 && map_len >= ((size_t)1 << 20));
        threads.emplace_back([this, gz_threads, use_bgzf] {
            std::unique_ptr<GsInflate> inf;
            std::unique_ptr<GsParallelGunzip> par;
            std::unique_ptr<GsBgzfReader> bgzf;  // blocks that say how long they are: inflated side by side
            std::vector<GsParallelGunzip::MemberEnd> ends;
            // the general decoders, for the file from `from` on
            auto start_general = [&](size_t from) {
                if (gz_threads >= 2 && map_len - from >= ((size_t)1 << 20)) {
                    par.reset(new GsParallelGunzip());
                    par->start(map + from, map_len - from, gz_threads, (size_t)1 << 20);
                } else {
                    // (from the start of the file the consumer of the blocks runs the CRC-32, verify_gzip; behind
                    // BGZF blocks the decoder checks it itself)
                    inf.reset(new GsInflate());
                    inf->init(map + from, map_len - from, from != 0);
                }
            };
            if (use_bgzf)
                bgzf.reset(new GsBgzfReader(map, map_len, gz_threads));
            else
                start_general(0);

            std::vector<uint8_t> window(32768);
            size_t hist = 0;
            bool done = map_len == 0;
            for (int64_t i = 0;; i++) {
                TextSlot &sl = slots[(size_t)(i % n_slots)];
                {
                    std::unique_lock<std::mutex> l(m);
                    cv.wait(l, [&] { return stop || sl.state == 0; });
                    if (stop) break;
                }
                bool err = !slot_buffer(sl);
                uint8_t *dst = err ? nullptr : sl.buf + headroom;
                size_t got = 0;
                sl.member_ends.clear();
                if (bgzf) {
                    while (!done && !err && got < block) {
                        size_t p = 0;
                        bool fin = false;
                        if (!bgzf->read(dst + got, block - got, &p, &fin)) err = true;  // (checks CRC-32 / ISIZE itself)
                        got += p;
                        if (fin) {
                            const size_t rest = bgzf->rest_offset();
                            bgzf.reset();
                            // ordinary gzip members may follow the BGZF blocks; anything that does not even start like
                            // a member is trailing garbage, which the general decoder ignores as well
                            if (rest < map_len && map_len - rest >= 10 && map[rest] == 0x1f && map[rest + 1] == 0x8b)
                                start_general(rest);
                            else
                                done = true;
                            break;
                        }
                        if (p == 0 && !err) err = true;
                    }
                }
                if (par) {
                    while (!done && !err && got < block) {
                        size_t p = 0;
                        ends.clear();
                        bool fin = false;
                        if (!par->read(dst + got, block - got, &p, nullptr, &fin)) err = true;  // (checks CRC-32 / ISIZE itself)
                        got += p;
                        if (fin) done = true;
                        if (p == 0 && !fin && !err) err = true;  // (cannot happen: read() blocks until it has bytes)
                    }
                } else if (inf && !err) {
                    memcpy(dst - hist, window.data() + (32768 - hist), hist);
                    while (!done && got < block) {
                        size_t p = 0;
                        const GsInflate::Status st = inf->decode(dst + got, block - got, hist + got, &p);
                        const uint64_t block_start = (uint64_t)i * block;
                        for (int e = 0; e < inf->n_member_ends(); e++)
                            sl.member_ends.push_back({(uint32_t)(inf->member_ends()[e].out_offset - block_start), inf->member_ends()[e].crc,
                                                      inf->member_ends()[e].isize});
                        inf->clear_member_ends();
                        got += p;
                        if (st == GsInflate::CORRUPT) err = true;
                        if (st != GsInflate::NEED_OUTPUT) done = true;
                    }
                    const size_t keep = got < 32768 ? got : 32768;  // (a short block is the last one)
                    if (keep == 32768) memcpy(window.data(), dst + got - 32768, 32768);
                    hist = keep == 32768 ? 32768 : hist;
                }
                sl.n = got;
                sl.eof = got < block || done;
                sl.io_error = err;
                {
                    std::lock_guard<std::mutex> l(m);
                    sl.state = 2;  // the thread below counts the newlines while this one goes on decoding
                }
                cv.notify_all();
                if (sl.eof || err) break;
            }
            if (par) par->stop();
        });
        threads.emplace_back([this] {
            GsRangePool counters(4);
            for (int64_t i = 0;; i++) {
                TextSlot &sl = slots[(size_t)(i % n_slots)];
                {
                    std::unique_lock<std::mutex> l(m);
                    cv.wait(l, [&] { return stop || sl.state == 2; });
                    if (stop) break;
                }
                fill_newlines(sl, sl.buf ? sl.buf + headroom : nullptr, sl.buf ? sl.n : 0, &counters);
                const bool last = sl.eof || sl.io_error;
                {
                    std::lock_guard<std::mutex> l(m);
                    sl.state = 1;
                }
                cv.notify_all();
                if (last) break;
            }
        });
    }
    // consumer side: CRC-32 of the gzip members over the delivered block (GZIPInputStream checks it while reading)
    bool verify_gzip(const TextSlot &sl) {
        if (!gz || !consumer_checks_crc) return true;
        const uint8_t *p = sl.buf + headroom;
        size_t at = 0;
        for (const auto &me : sl.member_ends) {
            run_crc = GsCrc32::update(run_crc, p + at, me.offset - at);
            run_size += me.offset - at;
            if (run_crc != me.crc || (uint32_t)run_size != me.isize) return false;
            run_crc = 0;
            run_size = 0;
            at = me.offset;
        }
        run_crc = GsCrc32::update(run_crc, p + at, sl.n - at);
        run_size += sl.n - at;
        return true;
    }
    void start() {
        if (gz) {
            start_gzip();
            return;
        }
        for (int t = 0; t < n_threads; t++)
            threads.emplace_back([this, t] {
                for (int64_t i = t;; i += n_threads) {
                    TextSlot &sl = slots[(size_t)(i % n_slots)];
                    {
                        std::unique_lock<std::mutex> l(m);
                        cv.wait(l, [&] { return stop || sl.state == 0; });
                        if (stop) return;
                    }
                    bool err = !slot_buffer(sl);
                    uint8_t *dst = err ? nullptr : sl.buf + headroom;
                    size_t got = 0;
                    while (!err && got < block) {
                        const ssize_t r = pread(fd, dst + got, block - got, (off_t)(start_off + i * (int64_t)block + (int64_t)got));
                        if (r < 0) {
                            if (errno == EINTR) continue;
                            err = true;
                            break;
                        }
                        if (r == 0) break;
                        got += (size_t)r;
                    }
                    sl.n = got;
                    sl.eof = got < block;
                    sl.io_error = err;
                    fill_newlines(sl, dst, got);
                    {
                        std::lock_guard<std::mutex> l(m);
                        sl.state = 1;
                    }
                    cv.notify_all();
                    if (sl.eof || err) return;  // later blocks are past the end: the consumer stops at this one
                }
            });
    }
    bool is_full(int64_t i) {
        std::lock_guard<std::mutex> l(m);
        return slots[(size_t)(i % n_slots)].state == 1;
    }
    TextSlot &wait_full(int64_t i) {
        TextSlot &sl = slots[(size_t)(i % n_slots)];
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return sl.state == 1; });
        return sl;
    }
    void release(int64_t i) {
        {
            std::lock_guard<std::mutex> l(m);
            slots[(size_t)(i % n_slots)].state = 0;
        }
        cv.notify_all();
    }
    void close() {
        {
            std::lock_guard<std::mutex> l(m);
            stop = true;
        }
        cv.notify_all();
        for (auto &t : threads) t.join();
        threads.clear();
        for (auto &sl : slots) pinned_pool().put(sl.buf, headroom + block);
        slots.clear();
        if (map) munmap((void *)map, map_len);
        map = nullptr;
        if (fd >= 0) ::close(fd);
        fd = -1;
    }
};

// 0: not for the text path, 1: plain FASTQ (parallel pread), 2: gzip FASTQ (inflating threads), 3 / 4: the same for FASTA
// (file type by name as the reference decides it, FastqMapGoal.java:188-201)
inline int text_path_kind(const std::string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return 0;
    unsigned char mg[2] = {0, 0};
    const size_t n = fread(mg, 1, 2, f);
    fclose(f);
    const bool gzip_content = n == 2 && mg[0] == 0x1f && mg[1] == 0x8b;  // zlib decides by content, so do we
    return (gzip_content ? 2 : 1) + (is_fasta_name(path) ? 2 : 0);
}


// ---------------------------------------------------------------------------------------------------
// output helper: plain or gzip by suffix (StreamProvider.getOutputStreamForFile)
// ---------------------------------------------------------------------------------------------------
// Writes happen on a thread of the file's own (started with the first buffer): the pipeline hands over whole buffers
// (one or a few per chunk, in order) and goes on with the next chunk while the previous one is on its way to the page
// cache or through deflate.  At most 64 buffers (a few chunks' worth) wait; close() reports a failed or short write.
class OutFile {
public:
    bool open(const char *path) {
        if (!path) return true;
        gzip_ = is_gzip_name(path);
        f_ = fopen(path, "wb");
        return f_ != nullptr;
    }
    bool active() const { return f_ != nullptr; }
    // A gzip file is written as a sequence of gzip members (RFC 1952 2.2; gunzip, zcat and java.util.zip.GZIPInputStream
    // read them as one stream), so that the threads that format a chunk can also compress their part of it: pack()
    // turns a buffer into members (BGZF blocks, level 1 like the "wb1" stream it replaces) and write(.., true) passes
    // them through.  Small buffers are better left to the writer thread, which collects >= 1 MiB before it packs.
    bool gzip() const { return gzip_; }
    bool pack(std::vector<uint8_t> &buf) {
        if (!gzip_ || !active()) return false;
        // the members are BGZF blocks (SAM spec 4.1: at most 64 KiB of text each, the member size in a 'BC' extra
        // subfield), so that bgzip-aware tools -- and GsBgzfReader -- can inflate the file block-parallel
        const size_t piece = 65280;
        const size_t n_pieces = (buf.size() + piece - 1) / piece;
        std::vector<uint8_t> out = take();
        out.resize(buf.size() + n_pieces * 64 + 64);  // (a block that does not compress is stored: 5 bytes + 26 of frame)
        z_stream z{};
        bool ok = deflateInit2(&z, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) == Z_OK;
        size_t produced = 0;
        for (size_t i = 0; i < n_pieces && ok; i++) {
            const size_t at = i * piece, n = std::min(piece, buf.size() - at);
            uint8_t *o = out.data() + produced;
            ok = out.size() - produced >= n + 64 && deflateReset(&z) == Z_OK;
            if (!ok) break;
            z.next_in = buf.data() + at;
            z.avail_in = (uInt)n;
            z.next_out = o + 18;
            z.avail_out = (uInt)(n + 64 - 26);
            ok = deflate(&z, Z_FINISH) == Z_STREAM_END;
            const size_t clen = (size_t)z.total_out, bsize = 18 + clen + 8 - 1;
            ok = ok && bsize < 65536;
            static const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
            memcpy(o, head, 16);
            o[16] = (uint8_t)bsize;
            o[17] = (uint8_t)(bsize >> 8);
            const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), buf.data() + at, (uInt)n), isz = (uint32_t)n;
            uint8_t *t = o + 18 + clen;
            for (int q = 0; q < 4; q++) t[q] = (uint8_t)(crc >> (8 * q));
            for (int q = 0; q < 4; q++) t[4 + q] = (uint8_t)(isz >> (8 * q));
            produced += 18 + clen + 8;
        }
        deflateEnd(&z);
        if (!ok) {
            failed_ = true;
            give_back(std::move(out));
            return false;
        }
        out.resize(produced);
        buf.swap(out);
        give_back(std::move(out));
        return true;
    }
    // text of this file that waits somewhere else (DeviceWriter: chunks gathered on the device are compressed together) has to be
    // in the queue before bytes from the host are: called in front of every host-side write and of close()
    std::function<void()> before_host_write;
    void write(std::vector<uint8_t> &&buf, bool packed = false) {
        if (!active() || buf.empty()) return;
        if (before_host_write) before_host_write();
        std::unique_lock<std::mutex> l(m_);
        if (!th_.joinable()) th_ = std::thread([this] { drain(); });
        cv_.wait(l, [&] { return q_.size() < 64; });
        Item it;
        it.buf = std::move(buf);
        it.packed = packed;
        q_.push(std::move(it));
        cv_.notify_all();
    }
    void write(const void *p, size_t n) {
        const uint8_t *b = static_cast<const uint8_t *>(p);
        write(std::vector<uint8_t>(b, b + n));
    }
    // The same without taking the bytes over: [p, p + n) -- text for a plain file, finished gzip members for a gzip file (what the
    // device writer delivers: gs_deflater_pack) -- stays the caller's until `done` has been called from the writer thread.
    void write_ref(const uint8_t *p, size_t n, std::function<void()> done) {
        if (!active() || n == 0) {
            if (done) done();
            return;
        }
        std::unique_lock<std::mutex> l(m_);
        if (!th_.joinable()) th_ = std::thread([this] { drain(); });
        cv_.wait(l, [&] { return q_.size() < 64; });
        Item it;
        it.packed = true;
        it.ref = p;
        it.ref_n = n;
        it.done = std::move(done);
        q_.push(std::move(it));
        cv_.notify_all();
    }
    // an empty buffer that keeps the capacity of one written earlier (fresh memory costs page faults)
    std::vector<uint8_t> take() {
        std::lock_guard<std::mutex> l(m_);
        if (free_.empty()) return {};
        std::vector<uint8_t> b = std::move(free_.back());
        free_.pop_back();
        return b;
    }
    // flushes and closes; false if any write failed
    bool close() {
        if (active() && before_host_write) before_host_write();
        if (th_.joinable()) {
            {
                std::lock_guard<std::mutex> l(m_);
                done_ = true;
            }
            cv_.notify_all();
            th_.join();
        }
        if (f_) {
            if (gzip_) {  // BGZF's end-of-file marker: an empty block (also what makes an empty output a valid gzip file)
                static const uint8_t eof_block[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0,
                                                      3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                if (!failed_ && fwrite(eof_block, 1, sizeof eof_block, f_) != sizeof eof_block) failed_ = true;
            }
            if (fclose(f_) != 0) failed_ = true;
        }
        f_ = nullptr;
        return !failed_;
    }
    ~OutFile() { close(); }

private:
    struct Item {
        std::vector<uint8_t> buf;
        bool packed = false;
        const uint8_t *ref = nullptr;  // write_ref: the bytes lie here (buf is empty)
        size_t ref_n = 0;
        std::function<void()> done;
    };
    void give_back(std::vector<uint8_t> &&b) {
        b.clear();
        std::lock_guard<std::mutex> l(m_);
        if (free_.size() < 64) free_.push_back(std::move(b));
    }
    void put(const std::vector<uint8_t> &b) {
        if (failed_ || b.empty()) return;
        if (fwrite(b.data(), 1, b.size(), f_) != b.size()) failed_ = true;
    }
    void flush_pending() {
        if (pending_.empty()) return;
        if (pack(pending_)) put(pending_);
        pending_.clear();
    }
    void drain() {
        for (;;) {
            Item it;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return done_ || !q_.empty(); });
                if (q_.empty()) break;
                it = std::move(q_.front());
                q_.pop();
            }
            cv_.notify_all();
            // (a dead file keeps taking buffers so that the producer never blocks on it)
            if (it.ref) {
                if (gzip_) flush_pending();  // order
                if (!failed_ && fwrite(it.ref, 1, it.ref_n, f_) != it.ref_n) failed_ = true;
                if (it.done) it.done();
                continue;
            }
            if (!gzip_ || it.packed) {
                if (gzip_) flush_pending();  // order
                put(it.buf);
            } else {
                pending_.insert(pending_.end(), it.buf.begin(), it.buf.end());
                if (pending_.size() >= ((size_t)1 << 20)) flush_pending();
            }
            give_back(std::move(it.buf));
        }
        if (gzip_) flush_pending();
    }
    bool gzip_ = false;
    FILE *f_ = nullptr;
    std::thread th_;
    std::mutex m_;
    std::condition_variable cv_;
    std::queue<Item> q_;
    std::vector<std::vector<uint8_t>> free_;
    std::vector<uint8_t> pending_;  // writer thread only
    bool done_ = false;
    std::atomic<bool> failed_{false};
};

typedef GsRangePool FormatPool;

inline int format_threads() {
    int t = (int)std::min<unsigned>(8, std::max<unsigned>(1, std::thread::hardware_concurrency() / 2));
    if (const char *e = getenv("GS_HOST_FORMAT_THREADS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 64) t = v;
    }
    return t;
}

}  // namespace gs_host
