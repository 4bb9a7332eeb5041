// gs_host.cpp -- C++ host layer above the C ABI (include/gshost.h): FASTQ/FASTA ingest with the reference's
// record semantics, the runMatcher / runFilter file pipelines (parse of batch i+1 overlaps the GPU work of batch i),
// Kraken-style and filtered-FASTQ writers, completeResults + CSV.  Plain C++17 + zlib; all GPU work goes through
// the C ABI of include/gsgpu.h.
#include "../../include/gshost.h"
#include "gs_inflate.h"

#include <errno.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <charconv>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_host_err;
int hfail(int code, const std::string &m) {
    g_host_err = m;
    return code;
}

bool ends_with(const std::string &s, const char *suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}
bool is_gzip_name(const std::string &s) { return ends_with(s, ".gz") || ends_with(s, ".gzip"); }  // StreamProvider.java:148-150

bool is_fasta_name(const std::string &s) {  // FastqMapGoal.java:64,188-201
    static const char *suf[] = {"fasta", "fa", "fna", "fas", "fasta.gz", "fa.gz", "fna.gz", "fas.gz",
                                "fasta.gzip", "fa.gzip", "fna.gzip", "fas.gzip"};
    for (const char *x : suf)
        if (ends_with(s, x)) return true;
    return false;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

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
    int64_t n() const { return (int64_t)seq_off.size() - 1; }
    void clear() {
        seq.clear();
        desc.clear();
        qual.clear();
        seq_off.assign(1, 0);
        desc_off.assign(1, 0);
        qual_off.assign(1, 0);
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

// ---------------------------------------------------------------------------------------------------
// output helper: plain or gzip by suffix (StreamProvider.getOutputStreamForFile)
// ---------------------------------------------------------------------------------------------------
class OutFile {
public:
    bool open(const char *path) {
        if (!path) return true;
        if (is_gzip_name(path)) {
            gz_ = gzopen(path, "wb1");
            return gz_ != nullptr;
        }
        f_ = fopen(path, "wb");
        return f_ != nullptr;
    }
    bool active() const { return gz_ || f_; }
    void write(const void *p, size_t n) {
        if (gz_)
            gzwrite(gz_, p, (unsigned)n);
        else if (f_)
            fwrite(p, 1, n, f_);
    }
    void put(char c) { write(&c, 1); }
    ~OutFile() {
        if (gz_) gzclose(gz_);
        if (f_) fclose(f_);
    }

private:
    gzFile gz_ = nullptr;
    FILE *f_ = nullptr;
};

// ReadEntry.write (AbstractFastqReader.java:570-584); qualities are '~' x L unless with_probs and present
void write_read(OutFile &out, const Batch &b, int64_t i, bool with_probs, std::vector<uint8_t> &tmp) {
    tmp.clear();
    const size_t d0 = b.desc_off[i], d1 = b.desc_off[i + 1], s0 = b.seq_off[i], s1 = b.seq_off[i + 1];
    tmp.insert(tmp.end(), b.desc.begin() + (long)d0, b.desc.begin() + (long)d1);
    tmp.push_back('\n');
    tmp.insert(tmp.end(), b.seq.begin() + (long)s0, b.seq.begin() + (long)s1);
    tmp.push_back('\n');
    tmp.push_back('+');
    tmp.push_back('\n');
    const size_t q0 = b.qual_off[i], q1 = b.qual_off[i + 1];
    if (with_probs && q1 > q0)
        tmp.insert(tmp.end(), b.qual.begin() + (long)q0, b.qual.begin() + (long)q1);
    else
        tmp.insert(tmp.end(), s1 - s0, (uint8_t)'~');
    tmp.push_back('\n');
    out.write(tmp.data(), tmp.size());
}

// bounded producer/consumer hand-off of parsed batches (depth 2: parse i+1 while the GPU works on i)
class BatchQueue {
public:
    void push(std::unique_ptr<Batch> b) {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return q_.size() < 2; });
        q_.push(std::move(b));
        cv_.notify_all();
    }
    std::unique_ptr<Batch> pop() {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return !q_.empty(); });
        auto b = std::move(q_.front());
        q_.pop();
        cv_.notify_all();
        return b;
    }

private:
    std::mutex m_;
    std::condition_variable cv_;
    std::queue<std::unique_ptr<Batch>> q_;
};

void append_int(std::string &s, long long v) {
    char buf[24];
    auto r = std::to_chars(buf, buf + sizeof(buf), v);
    s.append(buf, r.ptr);
}

// Double.toString: shortest round-trip digits; decimal notation for 1e-3 <= |d| < 1e7, else d.dddE[-]n
std::string java_double(double v) {
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

}  // namespace

// ---------------------------------------------------------------------------------------------------
// C API
// ---------------------------------------------------------------------------------------------------
struct gs_fastq {
    std::unique_ptr<FastqParser> parser;
    Batch batch;
};

extern "C" int gs_fastq_open(gs_fastq **out, const char *path, int fasta, int k) {
    if (!out || !path) return hfail(GS_E_INVALID, "NULL argument");
    const bool fa = fasta < 0 ? is_fasta_name(path) : fasta != 0;
    auto r = std::make_unique<gs_fastq>();
    r->parser = std::make_unique<FastqParser>(k, fa);
    if (!r->parser->open(path)) return hfail(GS_E_INVALID, std::string("cannot open ") + path);
    *out = r.release();
    return GS_OK;
}

extern "C" int gs_fastq_next(gs_fastq *r, int64_t max_reads, int64_t max_bytes, gs_read_batch *b) {
    if (!r || !b) return hfail(GS_E_INVALID, "NULL argument");
    r->parser->parse(r->batch, max_reads, max_bytes);
    b->n_reads = r->batch.n();
    b->seq = r->batch.seq.data();
    b->seq_off = r->batch.seq_off.data();
    b->desc = r->batch.desc.data();
    b->desc_off = r->batch.desc_off.data();
    b->qual = r->batch.qual.data();
    b->qual_off = r->batch.qual_off.data();
    b->first_read_no = r->batch.first_read_no;
    return GS_OK;
}

extern "C" int gs_fastq_totals(const gs_fastq *r, int64_t *reads, int64_t *kmers, int64_t *bps) {
    if (!r) return hfail(GS_E_INVALID, "NULL argument");
    if (reads) *reads = r->parser->reads_;
    if (kmers) *kmers = r->parser->kmers_;
    if (bps) *bps = r->parser->bps_;
    return GS_OK;
}

extern "C" int gs_fastq_close(gs_fastq *r) {
    delete r;
    return GS_OK;
}

extern "C" const char *gs_host_last_error(void) { return g_host_err.c_str(); }

// the ingest path's gzip decoder on a memory range, delivering `block` bytes per decode call (test hook)
extern "C" int gs_host_gunzip(const uint8_t *in, size_t n_in, uint8_t *out, size_t out_cap, size_t *n_out, size_t block) {
    if ((!in && n_in) || !out || !n_out || block == 0) return hfail(GS_E_INVALID, "bad argument");
    std::unique_ptr<GsInflate> inf(new GsInflate());
    inf->init(in, n_in, true);
    size_t total = 0;
    for (;;) {
        size_t room = out_cap - total;
        if (room > block) room = block;
        size_t p = 0;
        const GsInflate::Status st = inf->decode(out + total, room, total, &p);
        total += p;
        if (st == GsInflate::CORRUPT) return hfail(GS_E_INVALID, "corrupt gzip stream");
        if (st == GsInflate::DONE) break;
        if (total == out_cap) {  // the stream may just have ended: one more call without room tells
            const GsInflate::Status st2 = inf->decode(out + total, 0, total, &p);
            if (st2 == GsInflate::DONE) break;
            return hfail(st2 == GsInflate::CORRUPT ? GS_E_INVALID : GS_E_NOMEM, st2 == GsInflate::CORRUPT ? "corrupt gzip stream" : "output buffer too small");
        }
    }
    *n_out = total;
    return GS_OK;
}

extern "C" int gs_host_java_double(double v, char *buf, int cap) {
    const std::string s = java_double(v);
    if (!buf || cap <= (int)s.size()) return GS_E_INVALID;
    memcpy(buf, s.c_str(), s.size() + 1);
    return GS_OK;
}

namespace {

// parse one source on a producer thread; the consumer gets batches in order; a null batch ends the stream
struct Producer {
    std::thread th;
    BatchQueue q;
    int64_t reads = 0, kmers = 0, bps = 0;
    double seconds = 0;
    std::string error;
    // path from byte `offset` on, or the memory range [mem, mem + mem_n) when path is empty
    void start(const std::string &path, int64_t offset, const uint8_t *mem, size_t mem_n, int k, int64_t batch_reads) {
        th = std::thread([this, path, offset, mem, mem_n, k, batch_reads] {
            FastqParser parser(k, !path.empty() && is_fasta_name(path));
            bool ok_open = true;
            if (path.empty())
                parser.open_mem(mem, mem_n);
            else
                ok_open = parser.open(path, offset);
            if (!ok_open) {
                error = "cannot open " + path;
            } else {
                for (;;) {
                    auto b = std::make_unique<Batch>();
                    const double t0 = now_s();
                    const bool ok = parser.parse(*b, batch_reads, (int64_t)1 << 30);
                    seconds += now_s() - t0;
                    if (!ok) break;
                    q.push(std::move(b));
                }
                reads = parser.reads_;  // totalReads += reads (AbstractLoggingFastqStreamer.java:123-125)
                kmers = parser.kmers_;
                bps = parser.bps_;
            }
            q.push(nullptr);
        });
    }
};

// ---- fast path for plain four-line FASTQ: reader threads fill pinned blocks, the device finds the records -----------
// (gs_match_submit_text).  The host only counts newlines to cut the stream at record boundaries.
size_t count_newlines(const uint8_t *p, size_t n) {
    size_t c = 0;
    for (size_t i = 0; i < n; i++) c += p[i] == '\n';  // vectorised by the compiler
    return c;
}

struct TextSlot {
    uint8_t *buf = nullptr;  // pinned: headroom (for the carried partial record) + block
    size_t n = 0;            // bytes read into the block
    int64_t newlines = 0;
    int64_t last4[4] = {-1, -1, -1, -1};  // offsets of the last four newlines of the block, last first
    int state = 0;           // 0 empty, 1 full
    bool eof = false, io_error = false;
    std::vector<std::pair<uint32_t, uint32_t>> member_ends;  // gzip input: (offset in the block, CRC-32 of the trailer)
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

private:
    std::mutex m_;
    std::vector<std::pair<uint8_t *, size_t>> idle_;
};
PinnedPool &pinned_pool() {
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

    int open(const std::string &path, size_t block_bytes, int readers, bool gzip) {
        if (gzip) {
            gz = true;
            fd = ::open(path.c_str(), O_RDONLY);
            if (fd < 0) return hfail(GS_E_INVALID, "cannot open " + path);
            struct stat sb;
            if (fstat(fd, &sb) != 0) return hfail(GS_E_INVALID, "cannot stat " + path);
            map_len = (size_t)sb.st_size;
            if (map_len) {
                void *m = mmap(nullptr, map_len, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m == MAP_FAILED) return hfail(GS_E_INVALID, "cannot map " + path);
                map = (const uint8_t *)m;
                madvise(m, map_len, MADV_SEQUENTIAL);
            }
            if (block_bytes < ((size_t)64 << 10)) block_bytes = (size_t)64 << 10;  // the 32 KiB window lives in the headroom
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
        n_threads = readers;
        n_slots = gzip ? 4 : 2 * readers;
        slots.resize((size_t)n_slots);
        for (auto &sl : slots) {
            int rc = pinned_pool().get(headroom + block, &sl.buf);
            if (rc) return rc;
        }
        return GS_OK;
    }
    void fill_newlines(TextSlot &sl, const uint8_t *dst, size_t got) {
        sl.newlines = (int64_t)count_newlines(dst, got);
        size_t end = got;
        for (int j = 0; j < 4; j++) {
            const void *q = end ? memrchr(dst, '\n', end) : nullptr;
            sl.last4[j] = q ? (int64_t)((const uint8_t *)q - dst) : -1;
            end = q ? (size_t)((const uint8_t *)q - dst) : 0;
        }
    }
    void start_gzip() {
        threads.emplace_back([this] {
            std::unique_ptr<GsInflate> inf(new GsInflate());
            inf->init(map, map_len, false);
            std::vector<uint8_t> window(32768);
            size_t hist = 0;
            bool done = map_len == 0;
            for (int64_t i = 0;; i++) {
                TextSlot &sl = slots[(size_t)(i % n_slots)];
                {
                    std::unique_lock<std::mutex> l(m);
                    cv.wait(l, [&] { return stop || sl.state == 0; });
                    if (stop) return;
                }
                uint8_t *dst = sl.buf + headroom;
                memcpy(dst - hist, window.data() + (32768 - hist), hist);
                size_t got = 0;
                bool err = false;
                sl.member_ends.clear();
                while (!done && got < block) {
                    size_t p = 0;
                    const GsInflate::Status st = inf->decode(dst + got, block - got, hist + got, &p);
                    const uint64_t block_start = (uint64_t)i * block;
                    for (int e = 0; e < inf->n_member_ends(); e++)
                        sl.member_ends.push_back({(uint32_t)(inf->member_ends()[e].out_offset - block_start), inf->member_ends()[e].crc});
                    inf->clear_member_ends();
                    got += p;
                    if (st == GsInflate::CORRUPT) err = true;
                    if (st != GsInflate::NEED_OUTPUT) done = true;
                }
                const size_t keep = got < 32768 ? got : 32768;  // (a short block is the last one)
                if (keep == 32768)
                    memcpy(window.data(), dst + got - 32768, 32768);
                hist = keep == 32768 ? 32768 : hist;
                sl.n = got;
                sl.eof = got < block || done;
                sl.io_error = err;
                fill_newlines(sl, dst, got);
                {
                    std::lock_guard<std::mutex> l(m);
                    sl.state = 1;
                }
                cv.notify_all();
                if (sl.eof || err) return;
            }
        });
    }
    // consumer side: CRC-32 of the gzip members over the delivered block (GZIPInputStream checks it while reading)
    bool verify_gzip(const TextSlot &sl) {
        if (!gz) return true;
        const uint8_t *p = sl.buf + headroom;
        size_t at = 0;
        for (const auto &me : sl.member_ends) {
            run_crc = GsCrc32::update(run_crc, p + at, me.first - at);
            if (run_crc != me.second) return false;
            run_crc = 0;
            at = me.first;
        }
        run_crc = GsCrc32::update(run_crc, p + at, sl.n - at);
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
                    uint8_t *dst = sl.buf + headroom;
                    size_t got = 0;
                    bool err = false;
                    while (got < block) {
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

// 0: not for the text path (FASTA), 1: plain FASTQ (parallel pread), 2: gzip FASTQ (one inflating thread)
int text_path_kind(const std::string &path) {
    if (is_fasta_name(path)) return 0;
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return 0;
    unsigned char mg[2] = {0, 0};
    const size_t n = fread(mg, 1, 2, f);
    fclose(f);
    const bool gzip_content = n == 2 && mg[0] == 0x1f && mg[1] == 0x8b;  // zlib decides by content, so do we
    return gzip_content ? 2 : 1;
}

}  // namespace

namespace {

// everything one runMatcher call carries from batch to batch
struct MatchCtx {
    gs_run *run = nullptr;
    gs_db_info info{};
    const gs_host_match_opts *opts = nullptr;
    OutFile filtered, kraken;
    std::vector<int32_t> cls, seg_code, seg_start;
    std::vector<uint8_t> flags, tmp;
    std::vector<uint64_t> seg_off;
    std::string line;
    std::vector<uint8_t> out_buf, flt_buf, nl_bytes;  // Kraken lines / filtered records of one batch, written in one go
    int64_t global_read_no = 0, filtered_reads = 0;  // read numbers run over all files of the call (file order)
    int64_t reads = 0, kmers = 0, bps = 0;
    double t_gpu = 0, t_parse = 0;
};

// MatcherReadEntry.writeMatchDetails (:723-756) for read i of the current batch / chunk (c.cls, c.seg_*): descriptor
// up to the first blank without its '@', class taxid, length, runs "taxid:n"
void kraken_line(MatchCtx &c, const uint8_t *desc, size_t dlen, int64_t L, int64_t i) {
    const gs_host_match_opts *opts = c.opts;
    const uint64_t s0 = c.seg_off[(size_t)i], s1 = c.seg_off[(size_t)i + 1];
    const int32_t cl = c.cls[(size_t)i];
    if (s1 == s0 || !(opts->write_all || cl >= 0)) return;
    const int64_t maxp = L - c.info.k + 1;
    std::string &line = c.line;
    line.assign(cl >= 0 ? "C\t" : "U\t");
    size_t de = dlen;
    for (size_t j = 1; j < dlen; j++)
        if (desc[j] == ' ') {
            de = j;
            break;
        }
    if (dlen > 1) line.append((const char *)desc + 1, de - 1);
    line.push_back('\t');
    line.append(cl >= 0 ? opts->taxids[cl] : "0");
    line.push_back('\t');
    append_int(line, L);
    line.push_back('\t');
    for (uint64_t sg = s0; sg < s1; sg++) {
        if (sg > s0) line.push_back(' ');
        const int32_t code = c.seg_code[(size_t)sg];
        if (code == -2)
            line.push_back('A');
        else if (code < 0)
            line.push_back('0');
        else
            line.append(opts->taxids[code]);
        line.push_back(':');
        append_int(line, (sg + 1 < s1 ? c.seg_start[(size_t)sg + 1] : maxp) - c.seg_start[(size_t)sg]);
    }
    line.push_back('\n');
    c.out_buf.insert(c.out_buf.end(), line.begin(), line.end());
}

// one parsed batch through the GPU and the per-read writers
int consume_batch(MatchCtx &c, Batch &b, int64_t &read_no) {
    const int64_t n = b.n();
    c.cls.resize((size_t)n);
    c.flags.resize((size_t)n);
    if (b.seq.empty()) b.seq.push_back(0);
    const double t0 = now_s();
    int err = gs_match_submit(c.run, b.seq.data(), b.seq_off.data(), n, read_no, GS_MEM_HOST, c.cls.data(), c.flags.data());
    if (!err && c.kraken.active()) {
        c.seg_off.resize((size_t)n + 1);
        err = gs_match_segments(c.run, b.seq.data(), b.seq_off.data(), n, GS_MEM_HOST, c.seg_off.data());
        if (!err) {
            c.seg_code.resize((size_t)c.seg_off[(size_t)n]);
            c.seg_start.resize((size_t)c.seg_off[(size_t)n]);
            err = gs_match_segments_fetch(c.run, c.seg_code.data(), c.seg_start.data());
        }
    }
    c.t_gpu += now_s() - t0;
    if (err) return err;
    read_no += n;
    c.out_buf.clear();
    for (int64_t i = 0; i < n; i++) {
        if (c.filtered.active() && (c.flags[(size_t)i] & GS_F_RETURNED)) {  // afterMatch (:304-307)
            write_read(c.filtered, b, i, false, c.tmp);
            c.filtered_reads++;
        }
        if (c.kraken.active()) {
            const size_t d0 = b.desc_off[(size_t)i], d1 = b.desc_off[(size_t)i + 1];
            const int64_t L = (int64_t)(b.seq_off[(size_t)i + 1] - b.seq_off[(size_t)i]);
            kraken_line(c, b.desc.data() + d0, d1 - d0, L, i);
        }
    }
    if (!c.out_buf.empty()) c.kraken.write(c.out_buf.data(), c.out_buf.size());
    return GS_OK;
}

// the general path: the reference's record parser on a producer thread (file from `offset`, or a memory range)
int parsed_source(MatchCtx &c, const std::string &path, int64_t offset, const uint8_t *mem, size_t mem_n, int64_t &read_no) {
    Producer prod;
    prod.start(path, offset, mem, mem_n, c.info.k, c.opts->batch_reads > 0 ? c.opts->batch_reads : (int64_t)1 << 20);
    int err = GS_OK;
    for (;;) {
        std::unique_ptr<Batch> b = prod.q.pop();
        if (!b) break;
        if (err) continue;  // keep draining so the producer can finish
        err = consume_batch(c, *b, read_no);
    }
    prod.th.join();
    if (!err && !prod.error.empty()) err = hfail(GS_E_INVALID, prod.error);
    c.reads += prod.reads;
    c.kmers += prod.kmers;
    c.bps += prod.bps;
    c.t_parse += prod.seconds;
    return err;
}

struct TextChunk {
    int64_t file_off;  // of the chunk's first byte
    int64_t reads_before;  // reads of this file in earlier chunks
    int64_t ticket;
};

// ReadEntry.write of record i of a raw chunk (newline offsets nl[]): descriptor, read, "+", '~' x length
// (appended to `buf`; the caller writes one buffer per chunk)
void append_text_record(std::vector<uint8_t> &buf, const uint8_t *text, const uint32_t *nl, int64_t i) {
    const size_t d0 = i == 0 ? 0 : (size_t)nl[4 * i - 1] + 1, d1 = nl[4 * i], s0 = d1 + 1, s1 = nl[4 * i + 1];
    const size_t at = buf.size(), dl = d1 - d0, sl = s1 - s0;
    buf.resize(at + dl + 2 * sl + 5);
    uint8_t *o = buf.data() + at;
    memcpy(o, text + d0, dl);
    o += dl;
    *o++ = '\n';
    memcpy(o, text + s0, sl);
    o += sl;
    *o++ = '\n';
    *o++ = '+';
    *o++ = '\n';
    memset(o, '~', sl);
    o += sl;
    *o = '\n';
}

// One FASTQ file (plain or gzip) going to the device as raw text blocks.  Falls back to
// parsed_source() from the first chunk the device refuses (gs_match_text_status), so any file the general path accepts
// gives the same result.  step() handles one block; several jobs can be stepped in turn (files read side by side),
// each with its own status bank on the device and its own range of read numbers.
struct TextJob {
    MatchCtx &c;
    std::string path;
    int bank;
    int64_t read_no;  // number of the next read of this file
    TextReader tr;
    std::vector<uint8_t> carry;
    std::vector<TextChunk> chunks;
    int64_t carry_lines = 0, reads_in_file = 0, carry_file_off = 0, first_ticket = -1, next_block = 0;
    int64_t base_tot[3] = {0, 0, 0}, tot[3] = {0, 0, 0};
    bool done = false;
    double t0 = 0;

    TextJob(MatchCtx &ctx, const std::string &p, int bank_, int64_t first_read_no) : c(ctx), path(p), bank(bank_), read_no(first_read_no) {}

    int open(bool gzip, int readers) {
        // measured on the MI355X box (tools/file_rate_sweep.sh, 5 GB file in the page cache): 8 readers x 8 MiB blocks
        // 24.8 GB/s of file, 4 x 32 MiB 10.6 GB/s, 8 x 128 MiB 9.1 GB/s -- blocks that stay in the CPU caches between
        // pread and the newline count win
        size_t block = (size_t)8 << 20;
        if (const char *e = getenv("GS_HOST_BLOCK_BYTES")) {
            const long long v = atoll(e);
            if (v >= 64 && v <= ((long long)1 << 29)) block = (size_t)v;
        }
        if (const char *e = getenv("GS_HOST_READERS")) {
            const int v = atoi(e);
            if (v >= 1 && v <= 32) readers = v;
        }
        t0 = now_s();
        int err = tr.open(path, block, readers, gzip);
        if (!err) err = gs_match_text_select(c.run, bank);
        int64_t failed = -1, bad = -1;
        if (!err) err = gs_match_text_status(c.run, &failed, &bad, base_tot);  // totals this bank has seen before
        if (!err) tr.start();
        return err;
    }

    // 1: a block was handled, 0: none ready (blocking = false only); `done` is set when the file is through
    int step(bool blocking, int *err_out) {
        int err = GS_OK;
        const int64_t i = next_block;
        if (!blocking && !tr.is_full(i)) return 0;
        TextSlot &sl = tr.wait_full(i);
        int64_t fallback_off = -1, fallback_reads = 0;
        bool last = false;
        if (sl.io_error || !tr.verify_gzip(sl)) {
            err = hfail(GS_E_INVALID, (tr.gz ? "corrupt gzip stream in " : "read error on ") + path);
        } else {
            err = gs_match_text_select(c.run, bank);
            uint8_t *blk = sl.buf + tr.headroom;
            const int64_t total = carry_lines + sl.newlines;
            const int64_t rem = total & 3, usable = total - rem;
            last = sl.eof;
            if (err) {
            } else if (usable == 0) {  // not one whole record yet: keep everything
                carry.insert(carry.end(), blk, blk + sl.n);
                carry_lines = total;
                if (carry.size() > tr.headroom && !last) {  // a record longer than a block: the general parser takes over
                    fallback_off = carry_file_off;
                    fallback_reads = reads_in_file;
                }
            } else if (carry.size() > tr.headroom) {
                fallback_off = carry_file_off;
                fallback_reads = reads_in_file;
            } else {
                const int64_t cut = sl.last4[rem];  // the newline with `rem` newlines behind it ends the last whole record
                uint8_t *start = blk - carry.size();
                if (!carry.empty()) memcpy(start, carry.data(), carry.size());
                int64_t ticket = -1;
                const bool per_read = c.filtered.active() || c.kraken.active();
                const int64_t n_chunk = usable >> 2;
                if (per_read) {
                    c.cls.resize((size_t)n_chunk);
                    c.flags.resize((size_t)n_chunk);
                }
                err = gs_match_submit_text(c.run, start, (int64_t)carry.size() + cut + 1, usable, GS_MEM_HOST, read_no + reads_in_file,
                                           per_read ? c.cls.data() : nullptr, per_read ? c.flags.data() : nullptr, &ticket);
                if (!err && per_read) {  // the writers need this chunk's results now
                    chunks.push_back({carry_file_off, reads_in_file, ticket});
                    err = check_refusal(&fallback_off, &fallback_reads);
                    chunks.pop_back();
                    if (!err && fallback_off < 0) err = write_chunk_outputs(start, n_chunk);
                }
                if (!err && fallback_off < 0) {
                    if (first_ticket < 0) first_ticket = ticket;
                    chunks.push_back({carry_file_off, reads_in_file, ticket});
                    reads_in_file += usable >> 2;
                    carry_file_off = i * (int64_t)tr.block + cut + 1;
                    carry.assign(blk + cut + 1, blk + sl.n);
                    carry_lines = rem;
                    err = gs_match_text_wait_copy(c.run, ticket);  // the pinned block goes back to its reader
                }
                // a file that is not four-line FASTQ fails in its first chunk: look early, then now and again
                if (!err && fallback_off < 0 && !per_read && (chunks.size() == 1 || (chunks.size() & 15) == 0))
                    err = check_refusal(&fallback_off, &fallback_reads);
            }
        }
        tr.release(i);
        next_block = i + 1;
        if (err || last || fallback_off >= 0) err = finish(err, fallback_off, fallback_reads);
        *err_out = err;
        return 1;
    }

private:
    // filtered FASTQ (afterMatch, :304-307) and Kraken-style lines (:723-756) of the chunk that was just matched, from
    // the raw block: the device returns the record geometry (newline offsets) and the segments
    int write_chunk_outputs(const uint8_t *text, int64_t n) {
        c.nl_bytes.resize((size_t)n * 4 * sizeof(uint32_t));
        uint32_t *nl = reinterpret_cast<uint32_t *>(c.nl_bytes.data());
        int err = gs_match_text_newlines(c.run, nl);
        if (!err && c.kraken.active()) {
            c.seg_off.resize((size_t)n + 1);
            err = gs_match_segments_text(c.run, c.seg_off.data());
            if (!err) {
                c.seg_code.resize((size_t)c.seg_off[(size_t)n]);
                c.seg_start.resize((size_t)c.seg_off[(size_t)n]);
                err = gs_match_segments_fetch(c.run, c.seg_code.data(), c.seg_start.data());
            }
        }
        if (err) return err;
        c.out_buf.clear();
        c.flt_buf.clear();
        for (int64_t r = 0; r < n; r++) {
            if (c.filtered.active() && (c.flags[(size_t)r] & GS_F_RETURNED)) {
                append_text_record(c.flt_buf, text, nl, r);
                c.filtered_reads++;
            }
            if (c.kraken.active()) {
                const size_t d0 = r == 0 ? 0 : (size_t)nl[4 * r - 1] + 1, d1 = nl[4 * r];
                kraken_line(c, text + d0, d1 - d0, (int64_t)nl[4 * r + 1] - (int64_t)d1 - 1, r);
            }
        }
        if (!c.flt_buf.empty()) c.filtered.write(c.flt_buf.data(), c.flt_buf.size());
        if (!c.out_buf.empty()) c.kraken.write(c.out_buf.data(), c.out_buf.size());
        return GS_OK;
    }

    int check_refusal(int64_t *fallback_off, int64_t *fallback_reads) {
        int64_t failed = -1, bad = -1;
        int err = gs_match_text_status(c.run, &failed, &bad, tot);
        if (err || failed < 0) return err;
        for (const TextChunk &ch : chunks)
            if (ch.ticket == failed) {
                *fallback_off = ch.file_off;
                *fallback_reads = ch.reads_before;
            }
        return gs_match_text_clear_error(c.run);
    }

    int finish(int err, int64_t fallback_off, int64_t fallback_reads) {
        done = true;
        tr.close();
        c.t_parse += now_s() - t0;
        if (err) return err;
        err = gs_match_text_select(c.run, bank);
        if (!err && fallback_off < 0) err = check_refusal(&fallback_off, &fallback_reads);  // also fetches the final totals
        if (err) return err;
        if (fallback_off >= 0) {  // `tot` was read after the refusal: it holds exactly the accepted chunks
            int64_t failed = -1, bad = -1;
            err = gs_match_text_status(c.run, &failed, &bad, tot);
            if (err) return err;
        }
        c.reads += tot[0] - base_tot[0];
        c.kmers += tot[1] - base_tot[1];
        c.bps += tot[2] - base_tot[2];
        if (fallback_off >= 0) {
            read_no += fallback_reads;
            return parsed_source(c, path, fallback_off, nullptr, 0, read_no);
        }
        read_no += reads_in_file;
        // what is left after the last whole four-line group (no final newline, truncated record): the general parser
        if (!carry.empty()) return parsed_source(c, std::string(), 0, carry.data(), carry.size(), read_no);
        return GS_OK;
    }
};

}  // namespace

namespace {

// the files of one runMatcher call into c.run (begin and finish are the caller's).  file_index (may be NULL): the
// position of each file in the global file order when several processes share the files of a run; read numbers are then
// (file_index << 32 | read in file).  reads_of_file[n_paths] receives the read counts, *composite says whether the
// max-contig read numbers of the run are in that (file, read) form.
int run_files(MatchCtx &c, const char *const *paths, int n_paths, const int32_t *file_index, std::vector<int64_t> &reads_of_file_out,
              bool *composite) {
    bool fast = true;
    if (const char *e = getenv("GS_HOST_FAST")) fast = atoi(e) != 0;
    int err = GS_OK;
    std::vector<int> kind((size_t)n_paths, 0);
    int n_gzip = 0;
    for (int i = 0; i < n_paths; i++) {
        kind[(size_t)i] = fast ? text_path_kind(paths[i]) : 0;
        n_gzip += kind[(size_t)i] == 2;
    }
    const int default_readers = (int)std::min<unsigned>(8, std::max<unsigned>(2, std::thread::hardware_concurrency() / 2));
    // Several gzip files: each is bound by its single inflating thread, so they are read side by side (up to 8 at a
    // time).  The read numbers of file f then start at f << 32, which keeps "first read in file order" (the max-contig
    // tie-break) intact; the column is converted back to running read numbers at the end.
    // (per-read outputs follow the read order: one file after the other)
    bool side_by_side = n_gzip >= 2 && n_paths <= 256 && !c.filtered.active() && !c.kraken.active();
    if (const char *e = getenv("GS_HOST_PARALLEL_FILES")) side_by_side = side_by_side && atoi(e) != 0;
    if (file_index) side_by_side = true;  // read numbers are (file << 32 | read): the files are independent anyway
    std::vector<int64_t> reads_of_file((size_t)n_paths, 0);
    if (!side_by_side) {
        int64_t read_no = 0;
        for (int i = 0; i < n_paths && !err; i++) {
            const std::string path(paths[i]);
            if (kind[(size_t)i]) {
                TextJob job(c, path, 0, read_no);
                err = job.open(kind[(size_t)i] == 2, default_readers);
                while (!err && !job.done) job.step(true, &err);
                if (!job.done) job.tr.close();
                read_no = job.read_no;
            } else {
                err = parsed_source(c, path, 0, nullptr, 0, read_no);
            }
        }
    } else {
        std::vector<std::unique_ptr<TextJob>> active;
        std::vector<int> file_of;
        int next = 0;
        while (!err && (next < n_paths || !active.empty())) {
            while (!err && next < n_paths && (int)active.size() < 8) {
                const int64_t base = (int64_t)(file_index ? file_index[next] : next) << 32;
                if (kind[(size_t)next]) {
                    int bank = 0;  // a free bank
                    for (;; bank++) {
                        bool used = false;
                        for (auto &j : active) used = used || j->bank == bank;
                        if (!used) break;
                    }
                    auto job = std::make_unique<TextJob>(c, std::string(paths[next]), bank, base);
                    err = job->open(kind[(size_t)next] == 2, 2);
                    active.push_back(std::move(job));
                    file_of.push_back(next);
                } else {  // FASTA etc.: the general parser, on its own
                    int64_t read_no = base;
                    err = parsed_source(c, std::string(paths[next]), 0, nullptr, 0, read_no);
                    reads_of_file[(size_t)next] = read_no - base;
                }
                next++;
            }
            bool progressed = false;
            for (size_t j = 0; j < active.size() && !err; j++) progressed = active[j]->step(false, &err) > 0 || progressed;
            for (size_t j = 0; j < active.size();) {
                if (active[j]->done) {
                    reads_of_file[(size_t)file_of[j]] = active[j]->read_no - ((int64_t)(file_index ? file_index[file_of[j]] : file_of[j]) << 32);
                    if (reads_of_file[(size_t)file_of[j]] >= ((int64_t)1 << 32) && !err)
                        err = hfail(GS_E_UNSUPPORTED, "more than 2^32 reads in one of several files read side by side (set GS_HOST_PARALLEL_FILES=0)");
                    active.erase(active.begin() + (long)j);
                    file_of.erase(file_of.begin() + (long)j);
                } else
                    j++;
            }
            if (!progressed && !active.empty()) std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        for (auto &j : active) j->tr.close();
    }
    reads_of_file_out = reads_of_file;
    *composite = side_by_side;
    return err;
}

}  // namespace

extern "C" int gs_host_match_files(gs_db *db, const gs_match_cfg *cfg, const char *const *paths, int n_paths,
                                   const gs_host_match_opts *opts, int64_t *table, double *dtable,
                                   gs_host_totals *totals) {
    if (!db || !cfg || !paths || n_paths < 0 || !table) return hfail(GS_E_INVALID, "NULL argument");
    MatchCtx c;
    int rc = gs_db_get_info(db, &c.info);
    if (rc) return rc;
    const gs_host_match_opts none{};
    if (!opts) opts = &none;
    c.opts = opts;
    if (opts->kraken_out_path && !opts->taxids) return hfail(GS_E_INVALID, "Kraken-style output needs the taxid strings");
    if (!c.filtered.open(opts->filtered_path) || !c.kraken.open(opts->kraken_out_path)) return hfail(GS_E_INVALID, "cannot open output file");
    rc = gs_match_begin(&c.run, db, cfg);
    if (rc) return rc;
    const double t_start = now_s();
    std::vector<int64_t> reads_of_file;
    bool side_by_side = false;
    int err = run_files(c, paths, n_paths, nullptr, reads_of_file, &side_by_side);
    if (!err) err = gs_match_finish(c.run, table, dtable);
    if (!err && side_by_side) {  // (file << 32 | read in file) -> running read number over the files in order
        std::vector<int64_t> before((size_t)n_paths + 1, 0);
        for (int i = 0; i < n_paths; i++) before[(size_t)i + 1] = before[(size_t)i] + reads_of_file[(size_t)i];
        for (int32_t v = 0; v < c.info.n_values; v++) {
            int64_t &x = table[(size_t)v * GS_N_COLS + GS_C_MAX_CONTIG_READ_NO];
            if (x >= 0) x = before[(size_t)(x >> 32)] + (x & 0xffffffffLL);
        }
    }
    gs_match_destroy(c.run);
    if (totals) {
        totals->reads = c.reads;
        totals->kmers = c.kmers;
        totals->bps = c.bps;
        totals->filtered_reads = c.filtered_reads;
        totals->seconds_total = now_s() - t_start;
        totals->seconds_parse = c.t_parse;
        totals->seconds_gpu = c.t_gpu;
    }
    return err;
}


// The same, into a run the caller began and will finish: for one-process-per-GPU runs that share the files of a sample
// (genestrip_amd/distributed.py: match_files_sharded) -- every process takes some of the files, merges the device state
// of its run with the others (gs_match_device_state) and finishes.  file_index[n_paths] = position of each file in the
// global file order; the read numbers handed to the device are (file_index << 32 | read in file), reads_of_file[n_paths]
// receives the read counts (needed to turn the max-contig read numbers into running ones after the merge).
extern "C" int gs_host_match_into(gs_run *run, gs_db *db, const char *const *paths, int n_paths, const int32_t *file_index,
                                  int64_t *reads_of_file, gs_host_totals *totals) {
    if (!run || !db || !paths || n_paths < 0 || !file_index || !reads_of_file) return hfail(GS_E_INVALID, "NULL argument");
    MatchCtx c;
    int rc = gs_db_get_info(db, &c.info);
    if (rc) return rc;
    const gs_host_match_opts none{};
    c.opts = &none;
    c.run = run;
    const double t_start = now_s();
    std::vector<int64_t> rof;
    bool composite = false;
    const int err = run_files(c, paths, n_paths, file_index, rof, &composite);
    for (int i = 0; i < n_paths && (size_t)i < rof.size(); i++) reads_of_file[i] = rof[(size_t)i];
    if (totals) {
        totals->reads = c.reads;
        totals->kmers = c.kmers;
        totals->bps = c.bps;
        totals->filtered_reads = 0;
        totals->seconds_total = now_s() - t_start;
        totals->seconds_parse = c.t_parse;
        totals->seconds_gpu = c.t_gpu;
    }
    return err;
}

namespace {

struct FilterCtx {
    gs_bloom *bloom = nullptr;
    int k = 31, min_pos_count = 1;
    double positive_ratio = 0.2;
    OutFile acc_out, rest_out;
    std::vector<uint8_t> accept, tmp;
    int64_t accepted = 0, reads = 0, kmers = 0, bps = 0;
    double t_gpu = 0, t_parse = 0;
};

// the general path for one source (file from `offset`, or a memory range): reference parser -> batches -> GPU -> writers
int filter_parsed_source(FilterCtx &c, const std::string &path, int64_t offset, const uint8_t *mem, size_t mem_n) {
    Producer prod;
    prod.start(path, offset, mem, mem_n, c.k, (int64_t)1 << 20);
    int err = GS_OK;
    for (;;) {
        std::unique_ptr<Batch> b = prod.q.pop();
        if (!b) break;
        if (err) continue;
        const int64_t n = b->n();
        c.accept.resize((size_t)n);
        if (b->seq.empty()) b->seq.push_back(0);
        const double t0 = now_s();
        err = gs_filter_submit(c.bloom, c.k, c.min_pos_count, c.positive_ratio, b->seq.data(), b->seq_off.data(), n, GS_MEM_HOST,
                               c.accept.data(), 0);
        c.t_gpu += now_s() - t0;
        if (err) continue;
        for (int64_t i = 0; i < n; i++) {  // nextEntry (FastqBloomFilter.java:92-105), input order
            if (c.accept[(size_t)i]) {
                c.accepted++;
                if (c.acc_out.active()) write_read(c.acc_out, *b, i, false, c.tmp);
            } else if (c.rest_out.active())
                write_read(c.rest_out, *b, i, false, c.tmp);
        }
    }
    prod.th.join();
    if (!err && !prod.error.empty()) err = hfail(GS_E_INVALID, prod.error);
    c.reads += prod.reads;
    c.kmers += prod.kmers;
    c.bps += prod.bps;
    c.t_parse += prod.seconds;
    return err;
}

// plain FASTQ: raw text blocks to the device (gs_filter_submit_text); accept flags and record geometry come back
int filter_text_file(FilterCtx &c, const std::string &path, bool gzip) {
    size_t block = (size_t)8 << 20;
    if (const char *e = getenv("GS_HOST_BLOCK_BYTES")) {
        const long long v = atoll(e);
        if (v >= 64 && v <= ((long long)1 << 29)) block = (size_t)v;
    }
    int readers = (int)std::min<unsigned>(8, std::max<unsigned>(2, std::thread::hardware_concurrency() / 2));
    if (const char *e = getenv("GS_HOST_READERS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 32) readers = v;
    }
    TextReader tr;
    int err = tr.open(path, block, readers, gzip);
    if (err) {
        tr.close();
        return err;
    }
    err = gs_filter_text_reset(c.bloom, 1);
    // results of a chunk land in pinned memory: accept flags + newline offsets
    uint8_t *h_acc = nullptr;
    uint32_t *h_nl = nullptr;
    size_t acc_cap = 0, nl_cap = 0;
    std::vector<uint8_t> carry, acc_buf, rest_buf;
    int64_t carry_lines = 0, carry_file_off = 0, fallback_off = -1;
    int64_t tot[3] = {0, 0, 0}, failed = -1, bad = -1;
    const double t0 = now_s();
    if (!err) tr.start();
    for (int64_t i = 0; !err; i++) {
        TextSlot &sl = tr.wait_full(i);
        if (sl.io_error || !tr.verify_gzip(sl)) {
            err = hfail(GS_E_INVALID, (tr.gz ? "corrupt gzip stream in " : "read error on ") + path);
            break;
        }
        uint8_t *blk = sl.buf + tr.headroom;
        const int64_t total = carry_lines + sl.newlines;
        const int64_t rem = total & 3, usable = total - rem;
        const bool eof = sl.eof;
        if (usable == 0) {
            carry.insert(carry.end(), blk, blk + sl.n);
            carry_lines = total;
            if (carry.size() > tr.headroom && !eof) fallback_off = carry_file_off;
        } else if (carry.size() > tr.headroom) {
            fallback_off = carry_file_off;
        } else {
            const int64_t cut = sl.last4[rem];
            uint8_t *start = blk - carry.size();
            if (!carry.empty()) memcpy(start, carry.data(), carry.size());
            const int64_t n_reads = usable >> 2;
            if (acc_cap < (size_t)n_reads) {
                gs_pinned_free(h_acc);
                h_acc = nullptr;
                acc_cap = (size_t)n_reads + (size_t)n_reads / 4;
                void *p = nullptr;
                if ((err = gs_pinned_alloc(&p, acc_cap))) break;
                h_acc = (uint8_t *)p;
            }
            if (nl_cap < (size_t)usable) {
                gs_pinned_free(h_nl);
                h_nl = nullptr;
                nl_cap = (size_t)usable + (size_t)usable / 4;
                void *p = nullptr;
                if ((err = gs_pinned_alloc(&p, nl_cap * sizeof(uint32_t)))) break;
                h_nl = (uint32_t *)p;
            }
            int64_t ticket = -1;
            const double tg = now_s();
            err = gs_filter_submit_text(c.bloom, c.k, c.min_pos_count, c.positive_ratio, start, (int64_t)carry.size() + cut + 1,
                                        usable, GS_MEM_HOST, h_acc, h_nl, 0, &ticket);
            if (!err) err = gs_filter_text_status(c.bloom, &failed, &bad, tot);  // synchronises: results are needed now
            c.t_gpu += now_s() - tg;
            if (err) break;
            if (failed >= 0) {  // not four-line FASTQ from here on: the general parser continues at this chunk
                fallback_off = carry_file_off;
            } else {
                acc_buf.clear();
                rest_buf.clear();
                for (int64_t r = 0; r < n_reads; r++) {  // nextEntry (FastqBloomFilter.java:92-105), input order
                    if (h_acc[r]) {
                        c.accepted++;
                        if (c.acc_out.active()) append_text_record(acc_buf, start, h_nl, r);
                    } else if (c.rest_out.active())
                        append_text_record(rest_buf, start, h_nl, r);
                }
                if (!acc_buf.empty()) c.acc_out.write(acc_buf.data(), acc_buf.size());
                if (!rest_buf.empty()) c.rest_out.write(rest_buf.data(), rest_buf.size());
                carry_file_off = i * (int64_t)tr.block + cut + 1;
                carry.assign(blk + cut + 1, blk + sl.n);
                carry_lines = rem;
            }
        }
        tr.release(i);
        if (eof || fallback_off >= 0) break;
    }
    tr.close();
    gs_pinned_free(h_acc);
    gs_pinned_free(h_nl);
    c.t_parse += now_s() - t0;
    if (err) return err;
    c.reads += tot[0];
    c.kmers += tot[1];
    c.bps += tot[2];
    if (fallback_off >= 0) {
        err = gs_filter_text_reset(c.bloom, 1);
        if (err) return err;
        return filter_parsed_source(c, path, fallback_off, nullptr, 0);
    }
    if (!carry.empty()) return filter_parsed_source(c, std::string(), 0, carry.data(), carry.size());
    return GS_OK;
}

}  // namespace

extern "C" int gs_host_filter_files(gs_bloom *bloom, int k, int min_pos_count, double positive_ratio,
                                    const char *const *paths, int n_paths, const char *filtered_path,
                                    const char *rest_path, gs_host_totals *totals) {
    if (!bloom || !paths || n_paths < 0) return hfail(GS_E_INVALID, "NULL argument");
    FilterCtx c;
    c.bloom = bloom;
    c.k = k;
    c.min_pos_count = min_pos_count;
    c.positive_ratio = positive_ratio;
    if (!c.acc_out.open(filtered_path) || !c.rest_out.open(rest_path)) return hfail(GS_E_INVALID, "cannot open output file");
    const double t_start = now_s();
    bool fast = true;
    if (const char *e = getenv("GS_HOST_FAST")) fast = atoi(e) != 0;
    int err = GS_OK;
    for (int f = 0; f < n_paths && !err; f++) {
        const std::string path(paths[f]);
        const int kind = fast ? text_path_kind(path) : 0;
        if (kind)
            err = filter_text_file(c, path, kind == 2);
        else
            err = filter_parsed_source(c, path, 0, nullptr, 0);
    }
    if (totals) {
        totals->reads = c.reads;
        totals->kmers = c.kmers;
        totals->bps = c.bps;
        totals->filtered_reads = c.accepted;
        totals->seconds_total = now_s() - t_start;
        totals->seconds_parse = c.t_parse;
        totals->seconds_gpu = c.t_gpu;
    }
    return err;
}

// ---------------------------------------------------------------------------------------------------
// completeResults + CSV
// ---------------------------------------------------------------------------------------------------
extern "C" int gs_host_write_csv(const char *path, const gs_host_tax_info *tax, const int64_t *table, const double *dtable,
                                 const gs_host_totals *totals) {
    if (!path || !tax || !table || !totals || !tax->parent_vi || !tax->taxids || !tax->db_kmers)
        return hfail(GS_E_INVALID, "NULL argument");
    const int nv = tax->n_values;
    // rows: every value with a CountsPerTaxid (>= 1 hit k-mer or >= 1 classified read) plus all their ancestors
    // (MatchingResult.java:88-98)
    std::vector<char> present((size_t)nv, 0);
    for (int v = 0; v < nv; v++) {
        const int64_t *row = table + (size_t)v * GS_N_COLS;
        if (tax->parent_vi[v] != -2 && (row[GS_C_READS] > 0 || row[GS_C_READS_1KMER] > 0)) present[(size_t)v] = 1;
    }
    for (int v = 0; v < nv; v++)
        if (present[(size_t)v] == 1)
            for (int a = tax->parent_vi[v]; a >= 0; a = tax->parent_vi[a])
                if (!present[(size_t)a]) present[(size_t)a] = 2;
    // tree order (sortTaxidsViaTree): by position; default pre-order over children in value-index order
    std::vector<int> pos((size_t)nv, 0);
    if (tax->position) {
        for (int v = 0; v < nv; v++) pos[(size_t)v] = tax->position[v];
    } else {
        std::vector<std::vector<int>> kids((size_t)nv);
        std::vector<int> roots, stack;
        for (int v = 0; v < nv; v++) {
            if (tax->parent_vi[v] >= 0)
                kids[(size_t)tax->parent_vi[v]].push_back(v);
            else if (tax->parent_vi[v] == -1)
                roots.push_back(v);
        }
        int counter = 0;
        for (auto it = roots.rbegin(); it != roots.rend(); ++it) stack.push_back(*it);
        while (!stack.empty()) {
            const int v = stack.back();
            stack.pop_back();
            pos[(size_t)v] = counter++;
            for (auto it = kids[(size_t)v].rbegin(); it != kids[(size_t)v].rend(); ++it) stack.push_back(*it);
        }
    }
    std::vector<int> rows;
    for (int v = 0; v < nv; v++)
        if (present[(size_t)v]) rows.push_back(v);
    std::sort(rows.begin(), rows.end(), [&](int a, int b) { return pos[(size_t)a] < pos[(size_t)b]; });
    // accumulate into ancestors in tree order (:104-117); value types READS, KMERS, READS_BPS, READS_1KMER, READS_KMERS
    static const int vcol[5] = {GS_C_READS, GS_C_KMERS, GS_C_READS_BPS, GS_C_READS_1KMER, GS_C_READS_KMERS};
    static const char *vname[5] = {"reads", "kmers", "reads bps", "read >=1 kmer", "reads kmers"};
    std::vector<int64_t> acc((size_t)nv * 5, 0);
    std::vector<double> accn((size_t)nv * 5, 0.0), accd((size_t)nv * 4, 0.0);
    auto val = [&](int v, int t) { return present[(size_t)v] == 1 ? table[(size_t)v * GS_N_COLS + vcol[t]] : (int64_t)0; };
    auto dval = [&](int v, int j) { return (present[(size_t)v] == 1 && dtable) ? dtable[(size_t)v * GS_N_DCOLS + j] : 0.0; };
    for (int v : rows) {
        const int64_t dbk = tax->db_kmers[v];
        for (int t = 0; t < 5; t++) {
            acc[(size_t)v * 5 + t] += val(v, t);
            accn[(size_t)v * 5 + t] += dbk > 0 ? (double)val(v, t) / (double)dbk : 0.0;
        }
        for (int j = 0; j < 4; j++) accd[(size_t)v * 4 + j] += dval(v, j);
    }
    // a descendant adds its OWN values to every ancestor, in tree order
    for (int v : rows) {
        const int64_t dbk = tax->db_kmers[v];
        for (int a = tax->parent_vi[v]; a >= 0; a = tax->parent_vi[a]) {
            for (int t = 0; t < 5; t++) {
                acc[(size_t)a * 5 + t] += val(v, t);
                accn[(size_t)a * 5 + t] += dbk > 0 ? (double)val(v, t) / (double)dbk : 0.0;
            }
            for (int j = 0; j < 4; j++) accd[(size_t)a * 4 + j] += dval(v, j);
        }
    }
    FILE *f = fopen(path, "wb");
    if (!f) return hfail(GS_E_INVALID, std::string("cannot open ") + path);
    std::string o;
    o = "pos;level;name;rank;taxid;reads;kmers from reads;kmers;unique kmers;contigs;average contig length;max contig length;"
        "reads >=1 kmer;reads bps;avg. read length;db coverage;exp. unique kmers;unique kmers / exp.;db kmers;parent taxid;"
        "mean error;kmer error std. dev.;mean class error;class error std. dev.;contig len std. dev.;";
    for (int t = 0; t < 5; t++) o += std::string("norm. ") + vname[t] + ";";
    for (int t = 0; t < 5; t++) o += std::string("acc. ") + vname[t] + ";acc. norm. " + vname[t] + ";";
    o += "max contig desc.;acc. mean error;acc. error std. dev.;acc. mean class error;acc. class error std. dev.;";
    const int nmc = (tax->max_kmer_counts && tax->max_kmer_res_counts > 0) ? tax->max_kmer_res_counts : 0;
    if (nmc) o += "max kmer counts;";  // ResultReporter.java:213, :262-271
    o.push_back('\n');
    auto max_counts = [&](int row) {
        for (int i = 0; i < nmc; i++) {
            if (i > 0) o.push_back(';');
            append_int(o, tax->max_kmer_counts[(size_t)row * (size_t)nmc + (size_t)i]);
        }
        if (nmc) o.push_back(';');
    };
    auto dbl = [&](double v, bool total_row, bool always = false) {  // ResultReporter.java:249-253
        if (!std::isnan(v) && !std::isinf(v) && (!total_row || always)) o += java_double(v);
        o.push_back(';');
    };
    // TOTAL row (pos 0): reads, kmers, reads bps, db kmers; everything else 0 / blank
    o += "0;0;TOTAL;;;";
    append_int(o, totals->reads);
    o += ";0;";
    append_int(o, totals->kmers);
    o += ";0;0;";
    dbl(0.0 / 0.0, true);  // average contig length: NaN -> blank
    o += "0;0;";
    append_int(o, totals->bps);
    o.push_back(';');
    dbl(totals->reads ? (double)totals->bps / (double)totals->reads : 0.0 / 0.0, true, true);
    dbl(0, true);
    dbl(0, true);
    dbl(0, true);
    append_int(o, tax->db_kmers_total);
    o += ";;";
    for (int j = 0; j < 5; j++) dbl(0, true);
    for (int t = 0; t < 5; t++) dbl(0, true);
    for (int t = 0; t < 10; t++) o.push_back(';');
    o.push_back(';');
    for (int j = 0; j < 4; j++) dbl(0, true);
    max_counts(nv);
    o.push_back('\n');
    int p = 1;
    for (int v : rows) {
        const int64_t *row = table + (size_t)v * GS_N_COLS;
        const bool own = present[(size_t)v] == 1;
        auto col = [&](int c) { return own ? row[c] : (int64_t)0; };
        int level = 0;
        for (int a = tax->parent_vi[v]; a >= 0; a = tax->parent_vi[a]) level++;
        const int64_t reads = col(GS_C_READS), kmers = col(GS_C_KMERS), contigs = col(GS_C_CONTIGS);
        const int64_t uniq = own ? row[GS_C_UNIQUE_KMERS] : 0, dbk = tax->db_kmers[v];
        append_int(o, p++);
        o.push_back(';');
        append_int(o, level);
        o.push_back(';');
        if (tax->names && tax->names[v]) o += tax->names[v];
        o.push_back(';');
        if (tax->ranks && tax->ranks[v]) o += tax->ranks[v];
        o.push_back(';');
        o += tax->taxids[v];
        o.push_back(';');
        append_int(o, reads);
        o.push_back(';');
        append_int(o, col(GS_C_READS_KMERS));
        o.push_back(';');
        append_int(o, kmers);
        o.push_back(';');
        append_int(o, uniq);
        o.push_back(';');
        append_int(o, (int32_t)contigs);  // Java field is int
        o.push_back(';');
        dbl((double)kmers / (double)contigs, false);
        append_int(o, col(GS_C_MAX_CONTIG_LEN));
        o.push_back(';');
        append_int(o, col(GS_C_READS_1KMER));
        o.push_back(';');
        append_int(o, col(GS_C_READS_BPS));
        o.push_back(';');
        dbl((double)col(GS_C_READS_BPS) / (double)reads, false);
        dbl((double)uniq / (double)dbk, false);
        const double expu = (1 - std::pow(1 - 1.0 / (double)dbk, (double)kmers)) * (double)dbk;
        dbl(expu, false);
        dbl((double)uniq / expu, false);
        append_int(o, dbk);
        o.push_back(';');
        if (tax->parent_vi[v] >= 0) o += tax->taxids[tax->parent_vi[v]];
        o.push_back(';');
        const double es = dval(v, GS_D_ERR_SUM), es2 = dval(v, GS_D_ERR_SQ_SUM), cs = dval(v, GS_D_CLASS_ERR_SUM),
                     cs2 = dval(v, GS_D_CLASS_ERR_SQ_SUM);
        dbl(es / (double)reads, false);
        dbl(std::sqrt((es2 - es * es / (double)reads) / (double)(reads - 1)), false);
        dbl(cs / (double)reads, false);
        dbl(std::sqrt((cs2 - cs * cs / (double)reads) / (double)(reads - 1)), false);
        dbl(std::sqrt(((double)col(GS_C_CONTIG_LEN_SQ_SUM) - ((double)kmers * (double)kmers) / (double)contigs) / (double)(contigs - 1)), false);
        for (int t = 0; t < 5; t++) dbl((double)val(v, t) / (double)dbk, false);
        for (int t = 0; t < 5; t++) {
            append_int(o, acc[(size_t)v * 5 + t]);
            o.push_back(';');
            o += java_double(accn[(size_t)v * 5 + t]);
            o.push_back(';');
        }
        if (tax->max_contig_desc && tax->max_contig_desc[v]) o += tax->max_contig_desc[v];
        o.push_back(';');
        const double areads = (double)acc[(size_t)v * 5 + 0];
        const double aes = accd[(size_t)v * 4 + 0], aes2 = accd[(size_t)v * 4 + 1], acs = accd[(size_t)v * 4 + 2],
                     acs2 = accd[(size_t)v * 4 + 3];
        dbl(aes / areads, false);
        dbl(std::sqrt((aes2 - aes * aes / areads) / (areads - 1)), false);
        dbl(acs / areads, false);
        dbl(std::sqrt((acs2 - acs * acs / areads) / (areads - 1)), false);
        if (own) max_counts(v);
        else if (nmc) o.push_back(';');  // maxKMerCounts == null for rows added as missing ancestors
        o.push_back('\n');
    }
    fwrite(o.data(), 1, o.size(), f);
    fclose(f);
    return GS_OK;
}
