// gs_host.cpp -- C++ host layer above the C ABI (include/gshost.h): the runMatcher / runFilter file pipelines -- raw
// text blocks to the device where the file allows it (TextJob), the reference-exact parser otherwise and as the
// fallback -- with the Kraken-style and filtered-FASTQ writers.  The byte-level side (readers, parser, gzip decoder)
// is in gs_ingest.h / gs_inflate.h, the CSV report in gs_report.cpp.  Plain C++17 + zlib; all GPU work goes through the
// C ABI of include/gsgpu.h.
#include "gs_ingest.h"

using namespace gs_host;

namespace {

// ReadEntry.write (AbstractFastqReader.java:570-584) appended to `buf`; qualities are '~' x L unless with_probs and
// present
void append_read(std::vector<uint8_t> &buf, const Batch &b, int64_t i, bool with_probs) {
    const size_t d0 = b.desc_off[i], d1 = b.desc_off[i + 1], s0 = b.seq_off[i], s1 = b.seq_off[i + 1];
    buf.insert(buf.end(), b.desc.begin() + (long)d0, b.desc.begin() + (long)d1);
    buf.push_back('\n');
    buf.insert(buf.end(), b.seq.begin() + (long)s0, b.seq.begin() + (long)s1);
    buf.push_back('\n');
    buf.push_back('+');
    buf.push_back('\n');
    const size_t q0 = b.qual_off[i], q1 = b.qual_off[i + 1];
    if (with_probs && b.has_qual)
        buf.insert(buf.end(), b.qual.begin() + (long)q0, b.qual.begin() + (long)q1);
    else
        buf.insert(buf.end(), s1 - s0, (uint8_t)'~');
    buf.push_back('\n');
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

}  // namespace


// ---------------------------------------------------------------------------------------------------
// C API
// ---------------------------------------------------------------------------------------------------
struct gs_fastq {
    std::unique_ptr<FastqParser> parser;
    Batch batch;
};

extern "C" int gs_fastq_open(gs_fastq **out, const char *path, int fasta, int k) try {
    if (!out || !path) return hfail(GS_E_INVALID, "NULL argument");
    const bool fa = fasta < 0 ? is_fasta_name(path) : fasta != 0;
    auto r = std::make_unique<gs_fastq>();
    r->parser = std::make_unique<FastqParser>(k, fa);
    if (!r->parser->open(path)) return hfail(GS_E_INVALID, std::string("cannot open ") + path);
    *out = r.release();
    return GS_OK;
} catch (const std::bad_alloc &) {
    return hfail(GS_E_NOMEM, "out of host memory");
} catch (const std::exception &e) {  // (nothing may leave through the C ABI)
    return hfail(GS_E_INVALID, std::string("unexpected exception: ") + e.what());
}

extern "C" int gs_fastq_next(gs_fastq *r, int64_t max_reads, int64_t max_bytes, gs_read_batch *b) try {
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
} catch (const std::bad_alloc &) {
    return hfail(GS_E_NOMEM, "out of host memory");
} catch (const std::exception &e) {  // (nothing may leave through the C ABI)
    return hfail(GS_E_INVALID, std::string("unexpected exception: ") + e.what());
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
extern "C" int gs_host_gunzip(const uint8_t *in, size_t n_in, uint8_t *out, size_t out_cap, size_t *n_out, size_t block) try {
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
} catch (const std::bad_alloc &) {
    return hfail(GS_E_NOMEM, "out of host memory");
} catch (const std::exception &e) {  // (nothing may leave through the C ABI)
    return hfail(GS_E_INVALID, std::string("unexpected exception: ") + e.what());
}


// the same stream through GsParallelGunzip (test hook): `threads` workers, compressed chunks of `chunk` bytes;
// CRC-32 and ISIZE of every member are checked inside read()
extern "C" int gs_host_gunzip_parallel(const uint8_t *in, size_t n_in, uint8_t *out, size_t out_cap, size_t *n_out, int threads,
                                       size_t chunk, size_t block) try {
    if ((!in && n_in) || !out || !n_out || threads < 1 || block == 0) return hfail(GS_E_INVALID, "bad argument");
    size_t total = 0;
    std::vector<uint8_t> spill(block);
    // BGZF blocks are inflated side by side (GsBgzfReader); ordinary members (behind them, or the whole file) go
    // through the speculative decoder -- the same hand-over as in the file pipeline (TextReader::start_gzip)
    size_t from = 0;
    if (GsBgzfReader::looks_like(in, n_in)) {
        GsBgzfReader bg(in, n_in, threads);
        bool done = false;
        while (!done) {
            uint8_t *dst = total < out_cap ? out + total : spill.data();
            const size_t room = total < out_cap ? std::min(block, out_cap - total) : block;
            size_t p = 0;
            if (!bg.read(dst, room, &p, &done)) return hfail(GS_E_INVALID, "corrupt gzip stream");
            if (dst == spill.data() && p > 0) return hfail(GS_E_NOMEM, "output buffer too small");
            total += p;
        }
        from = bg.rest_offset();
        if (!(from < n_in && n_in - from >= 10 && in[from] == 0x1f && in[from + 1] == 0x8b)) from = n_in;  // trailing garbage
        if (from >= n_in) {
            *n_out = total;
            return GS_OK;
        }
    }
    GsParallelGunzip pg;
    pg.start(in + from, n_in - from, threads, chunk);
    bool done = false;
    while (!done) {
        uint8_t *dst = total < out_cap ? out + total : spill.data();
        const size_t room = total < out_cap ? std::min(block, out_cap - total) : block;
        size_t p = 0;
        if (!pg.read(dst, room, &p, nullptr, &done)) return hfail(GS_E_INVALID, "corrupt gzip stream");
        if (dst == spill.data() && p > 0) return hfail(GS_E_NOMEM, "output buffer too small");
        total += p;
    }
    *n_out = total;
    return GS_OK;
} catch (const std::bad_alloc &) {
    return hfail(GS_E_NOMEM, "out of host memory");
} catch (const std::exception &e) {  // (nothing may leave through the C ABI)
    return hfail(GS_E_INVALID, std::string("unexpected exception: ") + e.what());
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
    void start(const std::string &path, int64_t offset, const uint8_t *mem, size_t mem_n, int k, int64_t batch_reads, bool mem_fasta = false) {
        th = std::thread([this, path, offset, mem, mem_n, k, batch_reads, mem_fasta] {
            FastqParser parser(k, path.empty() ? mem_fasta : is_fasta_name(path));
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

}  // namespace

namespace {

// everything one runMatcher call carries from batch to batch
// grow-only array in pinned host memory: per-read results come back from the device into these (a copy into pageable
// memory is staged by the runtime and several times slower); resize() does not keep the contents
template <class T>
struct PinnedVec {
    T *p = nullptr;
    size_t cap = 0, n = 0;
    int resize(size_t m) {
        if (m > cap) {
            gs_pinned_free(p);
            p = nullptr;
            cap = 0;
            void *q = nullptr;
            const size_t want = m + m / 4 + 64;
            const int err = gs_pinned_alloc(&q, want * sizeof(T));
            if (err) return err;
            p = static_cast<T *>(q);
            cap = want;
        }
        n = m;
        return GS_OK;
    }
    T *data() { return p; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    PinnedVec() = default;
    PinnedVec(const PinnedVec &) = delete;
    PinnedVec &operator=(const PinnedVec &) = delete;
    ~PinnedVec() { gs_pinned_free(p); }
};

// Big page-locked buffers kept from call to call (locking half a gigabyte of pages costs ~0.1 s -- more than a file of four million
// reads takes to filter): get() hands out an idle buffer of at least `bytes` (or allocates), put() takes it back.  Never freed.
struct PinnedPool {
    std::mutex m;
    std::vector<std::pair<void *, size_t>> idle;
    void *get(size_t bytes, size_t *cap) {
        {
            std::lock_guard<std::mutex> l(m);
            for (size_t i = 0; i < idle.size(); i++)
                if (idle[i].second >= bytes) {
                    void *q = idle[i].first;
                    *cap = idle[i].second;
                    idle.erase(idle.begin() + (long)i);
                    return q;
                }
            if (!idle.empty()) {  // too small: give the pages back before asking for more
                gs_pinned_free(idle.back().first);
                idle.pop_back();
            }
        }
        void *q = nullptr;
        const size_t want = bytes + bytes / 8 + 4096;
        if (gs_pinned_alloc(&q, want) != GS_OK) return nullptr;
        *cap = want;
        return q;
    }
    void put(void *q, size_t cap) {
        if (!q) return;
        std::lock_guard<std::mutex> l(m);
        idle.emplace_back(q, cap);
    }
};
inline PinnedPool &pinned_pool() {
    static PinnedPool *p = new PinnedPool();  // (never destroyed: the runtime may be gone by the time statics are torn down)
    return *p;
}
struct PooledBuf {  // one buffer of the pool, returned when it goes out of scope
    void *p = nullptr;
    size_t cap = 0;
    int need(size_t bytes) {
        if (bytes <= cap) return GS_OK;
        pinned_pool().put(p, cap);
        p = pinned_pool().get(bytes, &cap);
        if (!p) {
            cap = 0;
            return hfail(GS_E_NOMEM, "page-locked memory for a text chunk");
        }
        return GS_OK;
    }
    PooledBuf() = default;
    PooledBuf(const PooledBuf &) = delete;
    PooledBuf &operator=(const PooledBuf &) = delete;
    ~PooledBuf() { pinned_pool().put(p, cap); }
};

// ... and the device DEFLATE writers of .gz outputs (slots and output buffer of the size of a chunk's text)
struct DeflaterPool {
    std::mutex m;
    std::vector<std::pair<int, gs_deflater *>> idle;
    gs_deflater *get(int device) {
        {
            std::lock_guard<std::mutex> l(m);
            for (size_t i = 0; i < idle.size(); i++)
                if (idle[i].first == device) {
                    gs_deflater *g = idle[i].second;
                    idle.erase(idle.begin() + (long)i);
                    return g;
                }
        }
        gs_deflater *g = nullptr;
        return gs_deflater_create(&g, device) == GS_OK ? g : nullptr;
    }
    void put(int device, gs_deflater *g) {
        if (!g) return;
        std::lock_guard<std::mutex> l(m);
        idle.emplace_back(device, g);
    }
};
inline DeflaterPool &deflater_pool() {
    static DeflaterPool *p = new DeflaterPool();  // (never destroyed, as the inflaters)
    return *p;
}
// GS_DEVICE_OUTPUT=0: the per-read writers format (and zlib compresses) on host threads, as before round 4
inline bool device_output() {
    if (const char *e = getenv("GS_DEVICE_OUTPUT")) return atoi(e) != 0;
    return true;
}

// The device side of one output file of the filter / match goal: the records the file wants have been gathered on the device
// (gs_filter_compact_text / gs_match_compact_text); emit() compresses them there when the file is gzip (gs_deflater_pack: BGZF
// members, what OutFile::pack makes with zlib on host threads) or fetches them as they are, into one of two page-locked buffers,
// and hands that buffer to the file's writer thread by reference.  Called on the chunk's formatting thread, chunk after chunk.
struct DeviceWriter {
    OutFile *out = nullptr;
    int device = 0;
    gs_deflater *defl = nullptr;
    PooledBuf buf[2];
    std::future<void> written[2];
    int flip = 0;
    int64_t bytes_text = 0, bytes_file = 0;
    // text of chunks waits on the device until this much is there (gzip files): 150 chunks of 8 MiB, compressed one by one, were
    // 150 x 0.6 ms of launches and waits (a plain file into a .gz: 4.1 Gbp/s against 12.7 from feeds of 256 MiB)
    static constexpr int64_t kTogether = (int64_t)32 << 20;
    int late_err = GS_OK;  // of a flush the file itself asked for
    std::recursive_mutex mu;
    void begin(OutFile *o, int dev) {
        out = o;
        device = dev;
        if (out) out->before_host_write = [this] {
            const int e = flush_pending();
            if (e && !late_err) late_err = e;
        };
    }
    // n_out bytes of buf[flip] to the file's writer thread, by reference
    void hand_over(int64_t n_out) {
        auto pr = std::make_shared<std::promise<void>>();
        written[flip] = pr->get_future();
        out->write_ref(static_cast<const uint8_t *>(buf[flip].p), (size_t)n_out, [pr] { pr->set_value(); });
        bytes_file += n_out;
        flip ^= 1;
    }
    int flush_pending() {
        std::lock_guard<std::recursive_mutex> l(mu);
        const int64_t n = defl ? gs_deflater_pending(defl) : 0;
        if (n <= 0) return GS_OK;
        if (written[flip].valid()) written[flip].get();  // (the buffer before last is on disk)
        const int64_t cap = gs_deflate_bound(n);
        int err = buf[flip].need((size_t)cap);
        if (err) return err;
        int64_t n_out = 0;
        if (gs_deflater_flush(defl, static_cast<uint8_t *>(buf[flip].p), cap, &n_out) != GS_OK)
            return hfail(GS_E_HIP, std::string("device DEFLATE writer: ") + gs_deflate_last_error());
        hand_over(n_out);
        return GS_OK;
    }
    // (set: the caller's chunk parity -- d_text stays valid until the chunk after next is gathered; nothing here depends on it)
    int emit(int set, const uint8_t *d_text, int64_t n_bytes) {
        (void)set;
        if (!out || !out->active() || n_bytes <= 0) return GS_OK;
        std::lock_guard<std::recursive_mutex> l(mu);
        if (late_err) return late_err;
        bytes_text += n_bytes;
        if (out->gzip()) {
            if (!defl && !(defl = deflater_pool().get(device))) return hfail(GS_E_NOMEM, "no device DEFLATE writer");
            if (n_bytes >= kTogether && gs_deflater_pending(defl) == 0) {  // enough for a call of its own: from where it lies
                if (written[flip].valid()) written[flip].get();
                const int64_t cap = gs_deflate_bound(n_bytes);
                int err = buf[flip].need((size_t)cap);
                if (err) return err;
                int64_t n_out = 0;
                if (gs_deflater_pack(defl, d_text, n_bytes, static_cast<uint8_t *>(buf[flip].p), cap, &n_out) != GS_OK)
                    return hfail(GS_E_HIP, std::string("device DEFLATE writer: ") + gs_deflate_last_error());
                hand_over(n_out);
                return GS_OK;
            }
            if (gs_deflater_append(defl, d_text, n_bytes) != GS_OK) return hfail(GS_E_HIP, std::string("device DEFLATE writer: ") + gs_deflate_last_error());
            return gs_deflater_pending(defl) >= kTogether ? flush_pending() : GS_OK;
        }
        if (written[flip].valid()) written[flip].get();
        int err = buf[flip].need((size_t)n_bytes);
        if (err) return err;
        if (gs_device_fetch(device, d_text, static_cast<uint8_t *>(buf[flip].p), n_bytes) != GS_OK) return hfail(GS_E_HIP, gs_inflate_last_error());
        hand_over(n_bytes);
        return GS_OK;
    }
    // what still waits is compressed, every buffer handed to the writer thread has been written; the deflater goes back to its pool
    int finish() {
        std::lock_guard<std::recursive_mutex> l(mu);
        int err = late_err;
        if (out && out->active()) {
            const int e = flush_pending();
            if (!err) err = e;
        }
        if (out) out->before_host_write = nullptr;
        for (auto &w : written)
            if (w.valid()) w.get();
        if (defl) {
            if (gs_deflater_pending(defl) > 0) {  // (an error on the way: nothing of this file may wait in a pooled object)
                int64_t dummy = 0;
                PooledBuf tmp;
                if (tmp.need((size_t)gs_deflate_bound(gs_deflater_pending(defl))) == GS_OK)
                    gs_deflater_flush(defl, static_cast<uint8_t *>(tmp.p), gs_deflate_bound(gs_deflater_pending(defl)), &dummy);
            }
            deflater_pool().put(device, defl);
        }
        defl = nullptr;
        return err;
    }
    ~DeviceWriter() { finish(); }
};

struct MatchCtx {
    gs_run *run = nullptr;
    gs_db_info info{};
    const gs_host_match_opts *opts = nullptr;
    OutFile filtered, kraken;
    // per-read results of a batch; two sets, so that the writers can work on one chunk while the device fills the
    // other (TextJob)
    struct Results {
        PinnedVec<int32_t> cls, seg_code, seg_start;
        PinnedVec<uint8_t> flags;
        PinnedVec<uint64_t> seg_off;
        PinnedVec<uint32_t> nl;
    } res[2];
    FormatPool pool{format_threads()};  // the per-read writers format a batch on these threads
    std::vector<uint32_t> taxid_len;    // strlen of opts->taxids[vi] (Kraken-style lines)
    size_t taxid_max = 1;
    int64_t global_read_no = 0, filtered_reads = 0;  // read numbers run over all files of the call (file order)
    int64_t reads = 0, kmers = 0, bps = 0;
    double t_gpu = 0, t_parse = 0;
    DeviceWriter filtered_dev;  // the filtered file fed from the device (TextJob::emit_filtered_device)
    // CountsPerTaxid.maxContigDescriptor for a host that never sees the reads (opts->max_contig_desc): after every chunk the device
    // names the read that holds each tax id's longest contig (gs_match_max_contig_reads); a holder that lies in the chunk just
    // submitted gets its name fetched while the chunk's text is at hand.  A maximum only moves to a later read by beating it, so
    // the name kept at the end is the one gs_match_finish's read number stands for.
    std::vector<int64_t> max_holder, max_now;
    bool track_desc() const { return opts && opts->max_contig_desc != nullptr && opts->max_contig_desc_stride >= 2; }
    // fetch(records in the chunk, n, out, stride) -> the records' descriptor lines, NUL-terminated; null: the text is not at hand
    int update_max_contig(int64_t first_no, int64_t n_reads, const std::function<int(const int64_t *, int32_t, uint8_t *, int32_t)> &fetch) {
        if (!track_desc() || n_reads <= 0) return GS_OK;
        const size_t nv = (size_t)info.n_values;
        if (max_holder.size() != nv) max_holder.assign(nv, -1);
        max_now.resize(nv);
        int err = gs_match_max_contig_reads(run, max_now.data());
        if (err) return err;
        std::vector<int64_t> recs;
        std::vector<size_t> vis;
        for (size_t v = 0; v < nv; v++) {
            const int64_t r = max_now[v];
            if (r != max_holder[v] && r >= first_no && r < first_no + n_reads) {
                recs.push_back(r - first_no);
                vis.push_back(v);
            }
            max_holder[v] = r;
        }
        if (recs.empty()) return GS_OK;
        const int32_t stride = opts->max_contig_desc_stride;
        std::vector<uint8_t> lines(recs.size() * (size_t)stride + 1, 0);
        if (fetch && (err = fetch(recs.data(), (int32_t)recs.size(), lines.data(), stride))) return err;
        for (size_t i = 0; i < recs.size(); i++) {
            const uint8_t *l = lines.data() + i * (size_t)stride;
            uint8_t *o = opts->max_contig_desc + vis[i] * (size_t)stride;
            int32_t j = 1;  // (behind the line's first character, up to the first blank: FastqKMerMatcher.java:404-407)
            for (; l[0] && j < stride && l[j] && l[j] != ' '; j++) o[j - 1] = l[j];
            o[j - 1] = 0;
        }
        return GS_OK;
    }
};

// MatcherReadEntry.writeMatchDetails (:723-756) for read i of the current batch / chunk (c.cls, c.seg_*): descriptor
// up to the first blank without its '@', class taxid, length, runs "taxid:n"
inline uint8_t *put_uint(uint8_t *o, uint64_t v) {
    char tmp[20];
    int n = 0;
    do {
        tmp[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (n) *o++ = (uint8_t)tmp[--n];
    return o;
}

void kraken_line(const MatchCtx &c, const MatchCtx::Results &rs, std::vector<uint8_t> &out, const uint8_t *desc, size_t dlen,
                 int64_t L, int64_t i) {
    const gs_host_match_opts *opts = c.opts;
    const uint64_t s0 = rs.seg_off[(size_t)i], s1 = rs.seg_off[(size_t)i + 1];
    const int32_t cl = rs.cls[(size_t)i];
    if (s1 == s0 || !(opts->write_all || cl >= 0)) return;
    const int64_t maxp = L - c.info.k + 1;
    // written in place: make room for the longest this line can get, cut back to what it took
    const size_t at = out.size();
    const size_t room = 2 + dlen + 1 + c.taxid_max + 1 + 20 + 1 + (size_t)(s1 - s0) * (c.taxid_max + 23) + 1;
    if (out.capacity() < at + room) out.reserve(std::max(2 * out.capacity(), at + room + ((size_t)1 << 16)));
    out.resize(at + room);
    uint8_t *o = out.data() + at;
    *o++ = cl >= 0 ? 'C' : 'U';
    *o++ = '\t';
    if (dlen > 1) {
        const void *sp = memchr(desc + 1, ' ', dlen - 1);
        const size_t n = sp ? (size_t)((const uint8_t *)sp - desc) - 1 : dlen - 1;
        memcpy(o, desc + 1, n);
        o += n;
    }
    *o++ = '\t';
    if (cl >= 0) {
        memcpy(o, opts->taxids[cl], c.taxid_len[(size_t)cl]);
        o += c.taxid_len[(size_t)cl];
    } else
        *o++ = '0';
    *o++ = '\t';
    o = put_uint(o, (uint64_t)L);
    *o++ = '\t';
    for (uint64_t sg = s0; sg < s1; sg++) {
        if (sg > s0) *o++ = ' ';
        const int32_t code = rs.seg_code[(size_t)sg];
        if (code == -2)
            *o++ = 'A';
        else if (code < 0)
            *o++ = '0';
        else {
            memcpy(o, opts->taxids[code], c.taxid_len[(size_t)code]);
            o += c.taxid_len[(size_t)code];
        }
        *o++ = ':';
        const int64_t cnt = (sg + 1 < s1 ? rs.seg_start[(size_t)sg + 1] : maxp) - rs.seg_start[(size_t)sg];
        if (cnt < 0) *o++ = '-';  // (cannot happen for segments the device produced; printed like the reference's int)
        o = put_uint(o, (uint64_t)(cnt < 0 ? -cnt : cnt));
    }
    *o++ = '\n';
    out.resize((size_t)(o - out.data()));
}

// per-thread output of one batch: Kraken lines and filtered records of a contiguous range of reads
struct FormatPart {
    std::vector<uint8_t> kraken, filtered;
    bool kraken_packed = false, filtered_packed = false;  // already a gzip member (OutFile::pack)
    int64_t n_filtered = 0;
    // a part that is big enough is compressed by the thread that made it
    void pack(OutFile &kr, OutFile &flt) {
        const size_t worth_it = (size_t)64 << 10;
        kraken_packed = kr.gzip() && kraken.size() >= worth_it && kr.pack(kraken);
        filtered_packed = flt.gzip() && filtered.size() >= worth_it && flt.pack(filtered);
    }
};

// hands the parts to the writers in read order
void write_parts(MatchCtx &c, std::vector<FormatPart> &parts) {
    for (FormatPart &p : parts) {
        c.filtered_reads += p.n_filtered;
        c.filtered.write(std::move(p.filtered), p.filtered_packed);
        c.kraken.write(std::move(p.kraken), p.kraken_packed);
    }
}

// one parsed batch through the GPU and the per-read writers
int consume_batch(MatchCtx &c, Batch &b, int64_t &read_no) {
    const int64_t n = b.n();
    MatchCtx::Results &rs = c.res[0];
    int err = rs.cls.resize((size_t)n);
    if (!err) err = rs.flags.resize((size_t)n);
    if (err) return err;
    if (b.seq.empty()) b.seq.push_back(0);
    const double t0 = now_s();
    err = gs_match_submit(c.run, b.seq.data(), b.seq_off.data(), n, read_no, GS_MEM_HOST, rs.cls.data(), rs.flags.data());
    if (!err && c.kraken.active()) {
        err = rs.seg_off.resize((size_t)n + 1);
        if (!err) err = gs_match_segments(c.run, b.seq.data(), b.seq_off.data(), n, GS_MEM_HOST, rs.seg_off.data());
        if (!err) err = rs.seg_code.resize((size_t)rs.seg_off[(size_t)n]);
        if (!err) err = rs.seg_start.resize((size_t)rs.seg_off[(size_t)n]);
        if (!err) err = gs_match_segments_fetch(c.run, rs.seg_code.data(), rs.seg_start.data());
    }
    c.t_gpu += now_s() - t0;
    if (err) return err;
    err = c.update_max_contig(read_no, n, [&b](const int64_t *recs, int32_t m, uint8_t *out, int32_t stride) {
        for (int32_t i = 0; i < m; i++) {
            const size_t d0 = b.desc_off[(size_t)recs[i]], d1 = b.desc_off[(size_t)recs[i] + 1];
            const size_t len = std::min<size_t>(d1 - d0, (size_t)stride - 1);
            memcpy(out + (size_t)i * (size_t)stride, b.desc.data() + d0, len);
            out[(size_t)i * (size_t)stride + len] = 0;
        }
        return (int)GS_OK;
    });
    if (err) return err;
    read_no += n;
    if (!c.filtered.active() && !c.kraken.active()) return GS_OK;
    std::vector<FormatPart> parts((size_t)c.pool.threads());
    c.pool.run(n, [&](int t, int64_t lo, int64_t hi) {
        FormatPart &p = parts[(size_t)t];
        p.filtered = c.filtered.take();
        p.kraken = c.kraken.take();
        for (int64_t i = lo; i < hi; i++) {
            if (c.filtered.active() && (rs.flags[(size_t)i] & GS_F_RETURNED)) {  // afterMatch (:304-307)
                append_read(p.filtered, b, i, c.opts->with_probs != 0);
                p.n_filtered++;
            }
            if (c.kraken.active()) {
                const size_t d0 = b.desc_off[(size_t)i], d1 = b.desc_off[(size_t)i + 1];
                const int64_t L = (int64_t)(b.seq_off[(size_t)i + 1] - b.seq_off[(size_t)i]);
                kraken_line(c, rs, p.kraken, b.desc.data() + d0, d1 - d0, L, i);
            }
        }
        p.pack(c.kraken, c.filtered);
    });
    write_parts(c, parts);
    return GS_OK;
}

// the general path: the reference's record parser on a producer thread (file from `offset`, or a memory range)
int parsed_source(MatchCtx &c, const std::string &path, int64_t offset, const uint8_t *mem, size_t mem_n, int64_t &read_no,
                  bool mem_fasta = false) {
    Producer prod;
    prod.start(path, offset, mem, mem_n, c.info.k, c.opts->batch_reads > 0 ? c.opts->batch_reads : (int64_t)1 << 20, mem_fasta);
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

std::atomic<int64_t> g_ml_chunks{0};  // chunks matched through the general FASTQ device path (gs_host_stat(0))
std::atomic<int64_t> g_filter_general_chunks{0};  // FASTA / general FASTQ chunks filtered on the device (gs_host_stat(1))

struct TextChunk {
    int64_t file_off;  // of the chunk's first byte
    int64_t reads_before;  // reads of this file in earlier chunks
    int64_t ticket;
};

// ReadEntry.write of record i of a raw chunk (newline offsets nl[]): descriptor, read, "+", then '~' x length or
// (with_probs) the record's quality line -- in a chunk the device accepted that is ONE line at least as long as the read
// (appended to `buf`; the caller writes one buffer per chunk)
void append_text_record(std::vector<uint8_t> &buf, const uint8_t *text, const uint32_t *nl, int64_t i, bool with_probs) {
    const size_t d0 = i == 0 ? 0 : (size_t)nl[4 * i - 1] + 1, d1 = nl[4 * i], s0 = d1 + 1, s1 = nl[4 * i + 1];
    const size_t at = buf.size(), dl = d1 - d0, sl = s1 - s0;
    if (with_probs) {
        const size_t q0 = (size_t)nl[4 * i + 2] + 1, ql = (size_t)nl[4 * i + 3] - q0;
        buf.resize(at + dl + sl + ql + 5);
        uint8_t *o = buf.data() + at;
        memcpy(o, text + d0, dl + 1 + sl + 1);  // descriptor and read lines as they stand, newlines included
        o += dl + sl + 2;
        *o++ = '+';
        *o++ = '\n';
        memcpy(o, text + q0, ql + 1);
        return;
    }
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
// One record of a FASTA or general FASTQ chunk as four-line FASTQ (ReadEntry.write, AbstractFastqReader.java:570-584): lines h
// (descriptor) to next - 1 of the chunk, newline offsets nl, line classes cls (1 descriptor, 2 sequence, 0 '+' / quality), read
// length L.  Descriptor (FASTA: '>' replaced by '@', :380), the read in ONE line, "+", then the quality characters of the record
// (general FASTQ with withProbs: every quality line that was consumed, joined) or '~' x length.
void append_general_record(std::vector<uint8_t> &o, const uint8_t *text, const uint32_t *nl, const uint8_t *cls, int64_t h, int64_t next,
                           int64_t L, bool is_fasta, bool probs) {
    auto line_start = [nl](int64_t i) { return i ? (size_t)nl[i - 1] + 1 : (size_t)0; };
    const size_t d0 = line_start(h), dlen = (size_t)nl[h] - d0, at = o.size();
    o.insert(o.end(), text + d0, text + d0 + dlen);
    if (is_fasta && dlen > 0) o[at] = '@';
    o.push_back('\n');
    int64_t i = h + 1;
    for (; i < next && cls[(size_t)i] == 2; i++) o.insert(o.end(), text + line_start(i), text + nl[i]);
    o.push_back('\n');
    o.push_back('+');
    o.push_back('\n');
    if (probs) {
        for (i++; i < next; i++) o.insert(o.end(), text + line_start(i), text + nl[i]);  // (behind the '+' line)
    } else
        o.insert(o.end(), (size_t)L, (uint8_t)'~');
    o.push_back('\n');
}

// Where a FASTA chunk may end inside a block: header lines ('>' at a line start) are counted by memchr over the block ('>' is
// rare); the chunk ends in front of the block's last header line -- everything up to there is whole records --, at the end of
// the file behind the final newline.  cut < 0: no record boundary in this block.
struct FastaCut {
    int64_t headers = 0;      // header lines that start inside the block
    int64_t cut = -1;         // the chunk ends here (exclusive, offset in the block)
    int64_t cut_headers = 0;  // headers in front of `cut`
    int64_t tail_lines = 0;   // newlines at or behind `cut`
};

FastaCut fasta_cut(const uint8_t *blk, int64_t n, bool last, const std::vector<uint8_t> &carry) {
    FastaCut fc;
    const bool at_line_start = carry.empty() || carry.back() == '\n';
    int64_t last_hdr = -1;
    for (const uint8_t *p = blk, *end = blk + n; p < end;) {
        const uint8_t *q = (const uint8_t *)memchr(p, '>', (size_t)(end - p));
        if (!q) break;
        if (q == blk ? at_line_start : q[-1] == '\n') {
            fc.headers++;
            last_hdr = q - blk;
        }
        p = q + 1;
    }
    if (last && n > 0 && blk[n - 1] == '\n') {
        fc.cut = n;
        fc.cut_headers = fc.headers;
    } else if (last && n == 0 && !carry.empty() && carry.back() == '\n') {
        fc.cut = 0;
    } else if (last_hdr > 0 || (last_hdr == 0 && !carry.empty())) {
        fc.cut = last_hdr;
        fc.cut_headers = fc.headers - 1;
    }
    if (fc.cut >= 0)
        for (const uint8_t *p = blk + fc.cut, *end = blk + n; p < end;) {
            const uint8_t *q = (const uint8_t *)memchr(p, '\n', (size_t)(end - p));
            if (!q) break;
            fc.tail_lines++;
            p = q + 1;
        }
    return fc;
}

// device inflaters are kept for the life of the process, per device: their buffers (two text buffers, two staging buffers, page-locked
// mirrors) take longer to allocate and free than a file takes to inflate
struct InflaterPool {
    std::mutex m;
    std::vector<std::pair<int, gs_inflater *>> idle;
    gs_inflater *get(int device) {
        {
            std::lock_guard<std::mutex> l(m);
            for (size_t i = 0; i < idle.size(); i++)
                if (idle[i].first == device) {
                    gs_inflater *g = idle[i].second;
                    idle.erase(idle.begin() + (long)i);
                    return g;
                }
        }
        gs_inflater *g = nullptr;
        return gs_inflater_create(&g, device) == GS_OK ? g : nullptr;
    }
    void put(int device, gs_inflater *g) {
        if (gs_inflater_reset(g) != GS_OK) {
            gs_inflater_destroy(g);
            return;
        }
        std::lock_guard<std::mutex> l(m);
        idle.emplace_back(device, g);
    }
};
// ... and the decoders of single-member streams, whose device buffers are gigabytes (one idle object per device is kept)
struct GunzipperPool {
    std::mutex m;
    std::vector<std::pair<int, gs_gunzipper *>> idle;
    int open(gs_gunzipper **out, int device, const uint8_t *gz, int64_t n) {
        gs_gunzipper *g = nullptr;
        {
            std::lock_guard<std::mutex> l(m);
            for (size_t i = 0; i < idle.size(); i++)
                if (idle[i].first == device) {
                    g = idle[i].second;
                    idle.erase(idle.begin() + (long)i);
                    break;
                }
        }
        if (g) {
            const int rc = gs_gunzipper_reopen(g, gz, n);
            if (rc == GS_OK) {
                *out = g;
                return GS_OK;
            }
            gs_gunzipper_close(g);
            *out = nullptr;
            return rc;
        }
        return gs_gunzipper_open(out, device, gz, n);
    }
    void put(int device, gs_gunzipper *g) {
        if (!g) return;
        gs_gunzipper_park(g);  // (the file is about to be unmapped: no upload thread may read it any more)
        {
            std::lock_guard<std::mutex> l(m);
            bool have = false;
            for (auto &x : idle) have = have || x.first == device;
            if (!have) {
                idle.emplace_back(device, g);
                return;
            }
        }
        gs_gunzipper_close(g);
    }
};
// compressed bytes of a stream's first batch when writers wait for its text (GS_HOST_GUNZIP_FIRST; 0: a full batch)
inline int64_t gunzip_first_span() {
    if (const char *e = getenv("GS_HOST_GUNZIP_FIRST")) return std::max<int64_t>(0, atoll(e));
    return (int64_t)64 << 20;
}

// ... and when nobody waits for it but the match kernel (GS_HOST_GUNZIP_FIRST_MATCH; 0: a full batch): the first batch is the only one
// that waits for its compressed bytes to cross PCIe
inline int64_t gunzip_first_span_match() {
    if (const char *e = getenv("GS_HOST_GUNZIP_FIRST_MATCH")) return std::max<int64_t>(0, atoll(e));
    return 0;
}

inline GunzipperPool &gunzipper_pool() {
    static GunzipperPool *p = new GunzipperPool();  // (never destroyed, as the inflaters)
    return *p;
}

inline InflaterPool &inflater_pool() {
    static InflaterPool *p = new InflaterPool();  // (never destroyed: the runtime may be gone by the time statics are torn down)
    return *p;
}

// every byte of the mapped file belongs to a BGZF member: list them (payload, ISIZE, CRC-32); false: not (only) BGZF
inline bool bgzf_member_list(const uint8_t *map, size_t map_len, std::vector<gs_inflate_member> &members) {
    members.clear();
    size_t o = 0;
    while (o < map_len) {
        size_t len = 0;
        uint32_t isize = 0;
        if (!GsBgzfReader::block_at(map, map_len, o, &len, &isize)) return false;
        const uint8_t *p = map + o;
        if (p[3] != 4) return false;  // (name / comment / header CRC: not what bgzip writes -- the general decoder knows them)
        const size_t hdr = 12 + ((size_t)p[10] | ((size_t)p[11] << 8));
        const uint8_t *t = p + len - 8;
        gs_inflate_member m{};
        m.payload_offset = (int64_t)(o + hdr);
        m.payload_len = (uint32_t)(len - hdr - 8);
        m.isize = isize;
        m.crc32 = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        if (isize > 0) members.push_back(m);
        o += len;
    }
    return !members.empty();
}

// text per feed of the device inflater (GS_HOST_BGZF_TEXT): a feed is ~8000 members at 512 MiB, two rounds over the device's wave slots
inline int64_t bgzf_text_target() {
    int64_t t = (int64_t)512 << 20;
    if (const char *e = getenv("GS_HOST_BGZF_TEXT")) {
        const long long v = atoll(e);
        if (v >= 65536 && v <= ((long long)1 << 29)) t = v;
    }
    return t;
}

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
    std::future<void> formatting;  // per-read outputs of the previous chunk on their way to the writers
    int64_t n_formatted = 0;
    int64_t held_ticket = -1, held_block = -1;
    // FASTA files (AbstractFastqReader.doReadFasta): chunks are cut in front of a header line, the device finds the
    // records (gs_match_submit_fasta); no per-read outputs on this path
    bool fasta = false;
    int64_t carry_headers = 0;
    // general FASTQ (sequence / quality over several lines): chunks of whole lines that start at a record's descriptor line; the
    // device finds the records (gs_match_submit_fastq_ml) and says how much of the chunk they cover, the rest is carried into the
    // next one.  A FASTQ file whose first chunk is not four-line FASTQ is read again this way (finish()); no per-read outputs.
    bool general = false;
    bool gz_ = false;
    int readers_ = 2;
    // block-gzip (BGZF) input: the members are listed from their headers, the COMPRESSED bytes go to the device and are inflated
    // there (gs_inflater_feed, one wave per member); without per-read outputs the text never exists on the host, with them
    // (Kraken-style lines, filtered FASTQ) it comes back once per feed, page-locked, for the writers.  What the device path cannot
    // take (a chunk the record scan refuses, the unterminated tail of the file) goes the usual way.
    bool dev_bgzf = false;
    gs_inflater *inf_ = nullptr;
    int inf_device_ = 0;
    std::vector<gs_inflate_member> members_;
    size_t next_member_ = 0;
    int64_t dev_tickets_[2] = {-1, -1};
    int64_t n_feeds_ = 0;
    PooledBuf dev_text_[2];  // the text of a feed on the host, for the per-read writers
    // a single-member gzip file inflated on the device as a whole (gs_gunzip_plan_device): its text lies in HBM, step_gunzip hands it
    // to the record scan in slices of whole records
    bool dev_gz = false;
    gs_gunzipper *gzr_ = nullptr;
    const uint8_t *gz_text_ = nullptr;  // the current batch (gs_gunzipper_next)
    int64_t gz_n_ = 0, gz_off_ = 0, gz_ticket_ = -1;
    int gz_last_ = 0;                   // 1: the file is through (members behind one another are decoded on the device, each with its own CRC-32 / ISIZE)

    TextJob(MatchCtx &ctx, const std::string &p, int bank_, int64_t first_read_no, bool fasta_ = false)
        : c(ctx), path(p), bank(bank_), read_no(first_read_no), fasta(fasta_) {}
    ~TextJob() { abort(); }
    // stops the readers (after the writers of the last chunk are through with its block)
    void abort() {
        drain();
        release_held();
        release_gunzipper();  // (before the file is unmapped: its upload thread reads the mapping)
        tr.close();
        if (inf_) {
            inflater_pool().put(inf_device_, inf_);
            inf_ = nullptr;
        }
    }
    // The device gunzipper goes back to its pool -- which parks its upload thread -- BEFORE tr.close() unmaps the file the thread
    // is copying from (a stream of up to 16 GiB is uploaded whole while the batches run; finish() is reached mid-stream by every
    // refusal or fallback).  Its device text stays valid until the object is reopened.
    void release_gunzipper() {
        if (!gzr_) return;
        gs_match_sync(c.run);  // (the record scan may still be copying out of its text)
        gunzipper_pool().put(inf_device_, gzr_);
        gzr_ = nullptr;
        gz_text_ = nullptr;
        gz_n_ = gz_off_ = 0;
    }

    bool list_bgzf_members() { return bgzf_member_list(tr.map, tr.map_len, members_); }

    // the chunk that was just submitted: first read number, reads; four_line: its descriptor lines can be fetched from the device
    int chunk_submitted(int64_t first_no, int64_t n_reads, bool four_line) {
        if (!c.track_desc()) return GS_OK;
        gs_run *run = c.run;
        if (!four_line) return c.update_max_contig(first_no, n_reads, nullptr);
        return c.update_max_contig(first_no, n_reads, [run](const int64_t *recs, int32_t m, uint8_t *out, int32_t stride) {
            return gs_match_text_descriptors(run, recs, m, out, stride);
        });
    }

    // Filtered FASTQ without Kraken-style lines: the reads matchRead returned true for (afterMatch, FastqKMerMatcher.java:304-307) are
    // gathered on the device and -- for a .gz file -- compressed there; the chunk's text never comes to the host.
    bool device_filtered() const { return device_output() && c.filtered.active() && !c.kraken.active(); }
    int dev_err_ = GS_OK;
    // after the chunk's flags are in (check_refusal has synchronised): gather now, compress / fetch / write on a thread of its own
    int emit_filtered_device() {
        const int set = (int)(n_formatted & 1);
        const uint8_t *d = nullptr;
        int64_t nb = 0, nr = 0;
        int err = gs_match_compact_text(c.run, c.opts->with_probs != 0, set, &d, &nb, &nr);
        if (err) return err;
        c.filtered_reads += nr;
        drain();  // one chunk at a time: output order
        if (dev_err_) return dev_err_;
        n_formatted++;
        auto job = [this, set, d, nb] {
            const int e = c.filtered_dev.emit(set, d, nb);
            if (e) dev_err_ = e;
        };
        try {
            formatting = std::async(std::launch::async, job);
        } catch (const std::system_error &) {  // no thread to be had: on this one
            job();
        }
        return GS_OK;
    }

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
        gz_ = gzip;
        readers_ = readers;
        int err = tr.open(path, block, readers, gzip);
        if (!err) err = gs_match_text_select(c.run, bank);
        int64_t failed = -1, bad = -1;
        if (!err) err = gs_match_text_status(c.run, &failed, &bad, base_tot);  // totals this bank has seen before
        if (!err && gzip && !fasta && !general) {
            bool want = true;
            if (const char *e = getenv("GS_DEVICE_INFLATE")) want = atoi(e) != 0;
            if (want && tr.map_len >= 28 && list_bgzf_members()) {
                int device = 0;
                if (gs_match_get_device(c.run, &device) == GS_OK && (inf_ = inflater_pool().get(device)) != nullptr) {
                    inf_device_ = device;
                    dev_bgzf = true;
                }
            }
        }
        if (!err && gzip && !fasta && !general && !dev_bgzf && tr.map_len >= 18) {
            // not block-gzip: a single-member stream (gzip, pigz) is inflated on the device as a whole -- block starts found speculatively,
            // segments decoded side by side, windows resolved in a second pass.  Whatever that path does not take (several members, a
            // damaged stream: the host decoders report it) is inflated on the host as before.
            bool want = true;
            if (const char *e = getenv("GS_DEVICE_INFLATE")) want = atoi(e) != 0;
            if (const char *e = getenv("GS_DEVICE_GUNZIP")) want = want && atoi(e) != 0;
            int device = 0;
            if (want && gs_match_get_device(c.run, &device) == GS_OK) {
                static const bool trace = getenv("GS_HOST_TRACE") != nullptr;
                const double tg = now_s();
                int grc = gunzipper_pool().open(&gzr_, device, tr.map, (int64_t)tr.map_len);
                // (writers behind this job: a small first batch, so that they start after 10 ms and not after the 27 a full batch takes)
                if (grc == GS_OK) grc = gs_gunzipper_first_span(gzr_, (c.filtered.active() || c.kraken.active()) ? gunzip_first_span() : gunzip_first_span_match());
                if (grc == GS_OK) grc = gs_gunzipper_next(gzr_, 0, &gz_text_, &gz_n_, &gz_last_);  // (the first batch now: a stream this path does not take shows here)
                if (trace) fprintf(stderr, "gunzip on the device: rc %d, first batch %lld bytes of text, %.2f ms%s%s\n", grc, (long long)gz_n_, (now_s() - tg) * 1e3, grc ? ": " : "", grc ? gs_inflate_last_error() : "");
                if (grc == GS_OK) {
                    dev_gz = true;
                    inf_device_ = device;
                } else if (gzr_) {
                    gunzipper_pool().put(device, gzr_);
                    gzr_ = nullptr;
                    gz_text_ = nullptr;
                    gz_n_ = 0;
                }
            }
        }
        if (!err && c.filtered.active()) {
            int device = 0;
            if (gs_match_get_device(c.run, &device) == GS_OK) c.filtered_dev.begin(&c.filtered, device);
        }
        if (!err && !dev_bgzf && !dev_gz) tr.start();
        return err;
    }

    // the next slice of the device text: whole four-line records up to the feed size, the leftover of the last slice to the host
    int step_gunzip(int *err_out) {
        int err = GS_OK;
        const bool per_read = c.filtered.active() || c.kraken.active();
        const int64_t text_target = per_read && !getenv("GS_HOST_BGZF_TEXT") ? ((int64_t)128 << 20) : bgzf_text_target();
        int64_t fallback_off = -1, fallback_reads = 0;
        int64_t n_lines = 0, n_bytes = 0;
        static const bool trace = getenv("GS_HOST_TRACE") != nullptr;
        const double ts0 = now_s();
        for (;;) {  // a slice with a whole record in it: from this batch, or with the next one behind what is left of this
            const int64_t rest = gz_n_ - gz_off_, look = std::min(rest, text_target);
            n_lines = n_bytes = 0;
            if (look > 0 && gs_text_cut_device(inf_device_, gz_text_ + gz_off_, look, &n_lines, &n_bytes) != GS_OK) err = hfail(GS_E_HIP, gs_inflate_last_error());
            if (err || n_lines > 0 || gz_last_ || look < rest) break;  // (look < rest: a full slice without a record -- the general parser, below)
            if (gz_ticket_ >= 0) {  // the scan's copy out of this batch's text must be through before the text is replaced
                err = gs_match_text_wait_copy(c.run, gz_ticket_);
                gz_ticket_ = -1;
                if (err) break;
            }
            const int grc = gs_gunzipper_next(gzr_, rest, &gz_text_, &gz_n_, &gz_last_);
            gz_off_ = 0;
            if (grc == GS_E_UNSUPPORTED || grc == GS_E_NOMEM) {  // from here on the host decoders (the tail that was kept belongs to them as well)
                fallback_off = carry_file_off;
                fallback_reads = reads_in_file;
                gz_n_ = 0;
                break;
            }
            if (grc != GS_OK) {
                err = hfail(GS_E_INVALID, std::string("corrupt gzip stream in ") + path + ": " + gs_inflate_last_error());
                break;
            }
        }
        const int64_t rest = gz_n_ - gz_off_;
        const bool last = gz_last_ != 0 && std::min(rest, text_target) == rest;
        const uint8_t *text = gz_text_ + gz_off_;
        const double ts1 = now_s();
        if (!err) err = gs_match_text_select(c.run, bank);
        if (!err && n_lines > 0 && per_read) {
            const int64_t n_chunk = n_lines >> 2;
            MatchCtx::Results &rs = c.res[n_formatted & 1];  // (the set of the chunk before last: its writers are done)
            PooledBuf &tb = dev_text_[n_formatted & 1];
            int64_t ticket = -1;
            const bool dev_f = device_filtered();
            err = rs.cls.resize((size_t)n_chunk);
            if (!err) err = rs.flags.resize((size_t)n_chunk);
            if (!err && !dev_f) err = tb.need((size_t)n_bytes);
            if (!err) err = gs_match_submit_text(c.run, text, n_bytes, n_lines, GS_MEM_DEVICE_TEXT, read_no + reads_in_file, rs.cls.data(), rs.flags.data(), &ticket);
            if (!err) err = chunk_submitted(read_no + reads_in_file, n_chunk, true);
            if (!err && !dev_f && gs_device_fetch(inf_device_, text, static_cast<uint8_t *>(tb.p), n_bytes) != GS_OK) err = hfail(GS_E_HIP, gs_inflate_last_error());
            if (!err) {
                chunks.push_back({carry_file_off, reads_in_file, ticket});
                err = check_refusal(&fallback_off, &fallback_reads);  // (synchronises: the results are needed now)
                if (!err && fallback_off < 0 && !dev_f) err = fetch_chunk_results(rs, n_chunk);
                if (err || fallback_off >= 0) chunks.pop_back();
            }
            if (!err && fallback_off < 0 && dev_f) {
                if (first_ticket < 0) first_ticket = ticket;
                reads_in_file += n_chunk;
                carry_file_off += n_bytes;
                err = emit_filtered_device();
            } else if (!err && fallback_off < 0) {
                if (first_ticket < 0) first_ticket = ticket;
                reads_in_file += n_chunk;
                carry_file_off += n_bytes;
                drain();  // one chunk at a time: output order, and the other result set becomes free
                n_formatted++;
                const uint8_t *h_text = static_cast<const uint8_t *>(tb.p);
                try {
                    formatting = std::async(std::launch::async, [this, &rs, h_text, n_chunk] { format_chunk(rs, h_text, n_chunk, -1); });
                } catch (const std::system_error &) {  // no thread to be had: on this one
                    format_chunk(rs, h_text, n_chunk, -1);
                }
            }
        } else if (!err && n_lines > 0) {
            int64_t ticket = -1;
            err = gs_match_submit_text(c.run, text, n_bytes, n_lines, GS_MEM_DEVICE, read_no + reads_in_file, nullptr, nullptr, &ticket);
            if (!err) err = chunk_submitted(read_no + reads_in_file, n_lines >> 2, true);
            if (!err) {
                gz_ticket_ = ticket;
                if (first_ticket < 0) first_ticket = ticket;
                chunks.push_back({carry_file_off, reads_in_file, ticket});
                reads_in_file += n_lines >> 2;
                carry_file_off += n_bytes;
                if (chunks.size() == 1 || (chunks.size() & 15) == 0) err = check_refusal(&fallback_off, &fallback_reads);
            }
        } else if (!err && fallback_off < 0 && !last) {  // not one whole record in a full slice: the general parser
            fallback_off = carry_file_off;
            fallback_reads = reads_in_file;
        }
        if (!err && fallback_off < 0) gz_off_ += n_bytes;
        const double ts2 = now_s();
        if (err || last || fallback_off >= 0) {
            if (!err && fallback_off < 0 && gz_off_ < gz_n_) {  // what is left behind the last whole record
                carry.resize((size_t)(gz_n_ - gz_off_));
                if (gs_device_fetch(inf_device_, gz_text_ + gz_off_, carry.data(), gz_n_ - gz_off_) != GS_OK) err = hfail(GS_E_HIP, gs_inflate_last_error());
            }
            err = finish(err, fallback_off, fallback_reads);
        }
        if (trace)
            fprintf(stderr, "gunzip slice: %lld bytes, %lld lines: cut (+ next batch) %.2f ms, submit %.2f ms, finish %.2f ms\n", (long long)n_bytes, (long long)n_lines, (ts1 - ts0) * 1e3,
                    (ts2 - ts1) * 1e3, (now_s() - ts2) * 1e3);
        *err_out = err;
        return 1;
    }

    // one run of members: inflate on the device, submit the whole records, carry the rest (on the device)
    int step_bgzf(int *err_out) {
        int err = GS_OK;
        // text per feed: a wave inflates a member in ~6 ms whatever else runs, so the rate is the number of members under way --
        // 512 MiB are ~8000 members, two rounds over the device's wave slots
        const bool per_read = c.filtered.active() || c.kraken.active();
        // (with writers behind it a feed is 128 MiB of text, not 512: they start four times earlier -- as filter_bgzf_file)
        const int64_t text_target = per_read && !getenv("GS_HOST_BGZF_TEXT") ? ((int64_t)128 << 20) : bgzf_text_target();
        auto run_end = [&](size_t from) {
            int64_t sum = 0;
            size_t e = from;
            while (e < members_.size() && (e == from || sum + members_[e].isize <= text_target)) sum += members_[e++].isize;
            return e;
        };
        const size_t a = next_member_, b = run_end(a), b2 = run_end(b);
        const bool last = b == members_.size();
        int64_t next_lo = 0, next_hi = 0;
        if (b2 > b) {
            next_lo = members_[b].payload_offset;
            next_hi = members_[b2 - 1].payload_offset + (int64_t)members_[b2 - 1].payload_len;
        }
        // The previous feed's text went to the device scan, which takes its own copy (device to device, a fraction of a
        // millisecond): that copy must be through before this feed runs -- the feed ends by moving its leftover into the OTHER
        // text buffer, which is the one the scan is copying from.
        const int slot = 0;
        if (dev_tickets_[slot] >= 0) {
            err = gs_match_text_wait_copy(c.run, dev_tickets_[slot]);
            dev_tickets_[slot] = -1;
        }
        const uint8_t *text = nullptr;
        int64_t n_bytes = 0, n_lines = 0, tail = 0;
        int64_t fallback_off = -1, fallback_reads = 0;
        static const bool trace = getenv("GS_HOST_TRACE") != nullptr;
        const double tt0 = now_s();
        if (!err && gs_inflater_feed(inf_, tr.map, members_.data() + a, (int64_t)(b - a), next_lo, next_hi, last ? 1 : 0, &text, &n_bytes, &n_lines, &tail) != GS_OK)
            err = hfail(GS_E_INVALID, std::string("corrupt gzip stream in ") + path + ": " + gs_inflate_last_error());
        n_feeds_++;
        next_member_ = b;
        const double tt1 = now_s();
        if (!err) err = gs_match_text_select(c.run, bank);
        if (!err && n_lines > 0 && per_read) {
            // the writers need the chunk's results and its text: class / flags come to host arrays (GS_MEM_DEVICE_TEXT), the text of
            // the feed is fetched while the match kernel runs, then the chunk is formatted on a thread of its own
            const int64_t n_chunk = n_lines >> 2;
            MatchCtx::Results &rs = c.res[n_formatted & 1];  // (the set of the chunk before last: its writers are done)
            PooledBuf &tb = dev_text_[n_formatted & 1];
            int64_t ticket = -1;
            const bool dev_f = device_filtered();
            err = rs.cls.resize((size_t)n_chunk);
            if (!err) err = rs.flags.resize((size_t)n_chunk);
            if (!err && !dev_f) err = tb.need((size_t)n_bytes);
            if (!err) err = gs_match_submit_text(c.run, text, n_bytes, n_lines, GS_MEM_DEVICE_TEXT, read_no + reads_in_file, rs.cls.data(), rs.flags.data(), &ticket);
            if (!err) err = chunk_submitted(read_no + reads_in_file, n_chunk, true);
            if (!err && !dev_f && gs_inflater_fetch(inf_, static_cast<uint8_t *>(tb.p), n_bytes) != GS_OK) err = hfail(GS_E_HIP, gs_inflate_last_error());
            if (!err) {
                chunks.push_back({carry_file_off, reads_in_file, ticket});
                err = check_refusal(&fallback_off, &fallback_reads);  // (synchronises: the results are needed now)
                if (!err && fallback_off < 0 && !dev_f) err = fetch_chunk_results(rs, n_chunk);
                if (err || fallback_off >= 0) chunks.pop_back();
            }
            if (!err && fallback_off < 0 && dev_f) {
                if (first_ticket < 0) first_ticket = ticket;
                reads_in_file += n_chunk;
                carry_file_off += n_bytes;
                err = emit_filtered_device();
            } else if (!err && fallback_off < 0) {
                if (first_ticket < 0) first_ticket = ticket;
                reads_in_file += n_chunk;
                carry_file_off += n_bytes;
                drain();  // one chunk at a time: output order, and the other result set becomes free
                n_formatted++;
                const uint8_t *h_text = static_cast<const uint8_t *>(tb.p);
                try {
                    formatting = std::async(std::launch::async, [this, &rs, h_text, n_chunk] { format_chunk(rs, h_text, n_chunk, -1); });
                } catch (const std::system_error &) {  // no thread to be had: on this one
                    format_chunk(rs, h_text, n_chunk, -1);
                }
            }
        } else if (!err && n_lines > 0) {
            int64_t ticket = -1;
            err = gs_match_submit_text(c.run, text, n_bytes, n_lines, GS_MEM_DEVICE, read_no + reads_in_file, nullptr, nullptr, &ticket);
            if (!err) err = chunk_submitted(read_no + reads_in_file, n_lines >> 2, true);
            if (!err) {
                dev_tickets_[slot] = ticket;
                if (first_ticket < 0) first_ticket = ticket;
                chunks.push_back({carry_file_off, reads_in_file, ticket});
                reads_in_file += n_lines >> 2;
                carry_file_off += n_bytes;
                if (chunks.size() == 1 || (chunks.size() & 15) == 0) err = check_refusal(&fallback_off, &fallback_reads);
            }
            if (trace) fprintf(stderr, "bgzf feed %lld: members %zu, feed %.2f ms, submit+check %.2f ms, %lld bytes %lld lines tail %lld\n", (long long)n_feeds_, b - a, (tt1 - tt0) * 1e3, (now_s() - tt1) * 1e3, (long long)n_bytes, (long long)n_lines, (long long)tail);
        } else if (!err && tail > ((int64_t)256 << 20) && !last) {  // no record boundary in a quarter of a gigabyte: the general parser
            fallback_off = carry_file_off;
            fallback_reads = reads_in_file;
        }
        if (err || last || fallback_off >= 0) {
            if (!err && fallback_off < 0) {  // what is left behind the last whole record
                int64_t n = 0;
                carry.resize((size_t)tail);
                if (tail > 0 && gs_inflater_tail(inf_, carry.data(), tail, &n) != GS_OK) err = hfail(GS_E_HIP, gs_inflate_last_error());
            }
            for (int q = 0; q < 2; q++)
                if (dev_tickets_[q] >= 0) {
                    const int e2 = gs_match_text_wait_copy(c.run, dev_tickets_[q]);
                    if (!err) err = e2;
                    dev_tickets_[q] = -1;
                }
            err = finish(err, fallback_off, fallback_reads);
        }
        *err_out = err;
        return 1;
    }

    // 1: a block was handled, 0: none ready (blocking = false only); `done` is set when the file is through
    int step(bool blocking, int *err_out) {
        if (dev_bgzf) return step_bgzf(err_out);
        if (dev_gz) return step_gunzip(err_out);
        if (general) return step_general(blocking, err_out);
        if (fasta) return step_fasta(blocking, err_out);
        int err = GS_OK;
        const int64_t i = next_block;
        bool keep_block = false;
        if (!blocking && !tr.is_full(i)) return 0;
        TextSlot &sl = tr.wait_full(i);
        int64_t fallback_off = -1, fallback_reads = 0;
        bool last = false;
        if (sl.io_error || !tr.verify_gzip(sl)) {
            err = hfail(tr.gz ? GS_E_INVALID : GS_E_IO, (tr.gz ? "corrupt gzip stream in " : "read error on ") + path);
        } else {
            err = gs_match_text_select(c.run, bank);
            uint8_t *blk = sl.buf + tr.headroom;
            const int64_t total = carry_lines + sl.newlines;
            const int64_t rem = total & 3, usable = total - rem;
            last = sl.eof;
            if (err) {
                // (the bank could not be selected: reported below)
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
                bool format_it = false;
                const bool per_read = c.filtered.active() || c.kraken.active();
                const int64_t n_chunk = usable >> 2;
                MatchCtx::Results &rs = c.res[n_formatted & 1];  // (the set of the chunk before last: its writers are done)
                if (per_read) {
                    err = rs.cls.resize((size_t)n_chunk);
                    if (!err) err = rs.flags.resize((size_t)n_chunk);
                }
                if (!err)
                    err = gs_match_submit_text(c.run, start, (int64_t)carry.size() + cut + 1, usable, GS_MEM_HOST, read_no + reads_in_file,
                                               per_read ? rs.cls.data() : nullptr, per_read ? rs.flags.data() : nullptr, &ticket);
                if (!err) err = chunk_submitted(read_no + reads_in_file, usable >> 2, true);
                const bool dev_f = per_read && device_filtered() && c.filtered.gzip();  // (a plain file: formatted from the reader's block, which is here anyway)
                if (!err && per_read) {  // the writers need this chunk's results
                    chunks.push_back({carry_file_off, reads_in_file, ticket});
                    err = check_refusal(&fallback_off, &fallback_reads);
                    chunks.pop_back();
                    if (!err && fallback_off < 0 && !dev_f) err = fetch_chunk_results(rs, n_chunk);
                    format_it = !err && fallback_off < 0 && !dev_f;
                    if (!err && fallback_off < 0 && dev_f) err = emit_filtered_device();
                }
                if (!err && fallback_off < 0) {
                    if (first_ticket < 0) first_ticket = ticket;
                    chunks.push_back({carry_file_off, reads_in_file, ticket});
                    reads_in_file += usable >> 2;
                    carry_file_off = i * (int64_t)tr.block + cut + 1;
                    carry.assign(blk + cut + 1, blk + sl.n);
                    carry_lines = rem;
                    if (format_it) {
                        err = gs_match_text_wait_copy(c.run, ticket);  // (the writers hand the block back, below)
                    } else {
                        // the pinned block goes back to its reader when its copy is through: looked at one chunk later,
                        // so that this thread is already submitting the next copy while this one runs
                        err = release_held();
                        held_ticket = ticket;
                        held_block = i;
                        keep_block = true;
                    }
                }
                if (format_it && !err) {
                    // (only now: the block returns to its reader when the writers are through with it, and the
                    // carry above had to be taken out first)
                    drain();  // one chunk at a time: output order, and the other result set becomes free
                    n_formatted++;
                    keep_block = true;
                    try {
                        formatting = std::async(std::launch::async, [this, &rs, start, n_chunk, i] { format_chunk(rs, start, n_chunk, i); });
                    } catch (const std::system_error &) {  // no thread to be had: on this one
                        format_chunk(rs, start, n_chunk, i);
                    }
                }
                // a file that is not four-line FASTQ fails in its first chunk: look early, then now and again
                if (!err && fallback_off < 0 && !per_read && (chunks.size() == 1 || (chunks.size() & 15) == 0))
                    err = check_refusal(&fallback_off, &fallback_reads);
            }
        }
        if (!keep_block) tr.release(i);  // (else: format_chunk releases it)
        next_block = i + 1;
        if (err || last || fallback_off >= 0) err = finish(err, fallback_off, fallback_reads);
        *err_out = err;
        return 1;
    }

    // The FASTA form of step(): the chunk ends in front of the block's last header line (everything up to there is whole
    // records), the rest is carried into the next block.  Headers and newlines are counted here (memchr over the block:
    // '>' is rare, the tail behind the last header is one record), the device checks the counts.
    int step_fasta(bool blocking, int *err_out) {
        int err = GS_OK;
        const int64_t i = next_block;
        bool keep_block = false;
        if (!blocking && !tr.is_full(i)) return 0;
        TextSlot &sl = tr.wait_full(i);
        int64_t fallback_off = -1, fallback_reads = 0;
        bool last = false;
        if (sl.io_error || !tr.verify_gzip(sl)) {
            err = hfail(tr.gz ? GS_E_INVALID : GS_E_IO, (tr.gz ? "corrupt gzip stream in " : "read error on ") + path);
        } else {
            err = gs_match_text_select(c.run, bank);
            uint8_t *blk = sl.buf + tr.headroom;
            const int64_t n = (int64_t)sl.n;
            last = sl.eof;
            const FastaCut fc = fasta_cut(blk, n, last, carry);
            const int64_t headers = fc.headers, cut = fc.cut, cut_headers = fc.cut_headers, tail_lines = fc.tail_lines;
            if (err) {
                // (the bank could not be selected: reported below)
            } else if (cut < 0) {  // no record boundary in this block: keep everything
                carry.insert(carry.end(), blk, blk + n);
                carry_lines += sl.newlines;
                carry_headers += headers;
                if (carry.size() > tr.headroom && !last) {  // a record longer than a block: the general parser takes over
                    fallback_off = carry_file_off;
                    fallback_reads = reads_in_file;
                }
            } else if (carry.size() > tr.headroom) {
                fallback_off = carry_file_off;
                fallback_reads = reads_in_file;
            } else {
                const int64_t lines = carry_lines + sl.newlines - tail_lines, records = carry_headers + cut_headers;
                uint8_t *start = blk - carry.size();
                if (!carry.empty()) memcpy(start, carry.data(), carry.size());
                int64_t ticket = -1;
                if (records >= ((int64_t)1 << 24)) {  // (more records than one chunk may hold: the general parser)
                    fallback_off = carry_file_off;
                    fallback_reads = reads_in_file;
                } else if ((int64_t)carry.size() + cut > 0) {
                    const bool kr = c.kraken.active() || c.filtered.active();
                    MatchCtx::Results &rs = c.res[0];
                    if (kr) {
                        err = rs.cls.resize((size_t)std::max<int64_t>(records, 1));
                        if (!err) err = rs.flags.resize((size_t)std::max<int64_t>(records, 1));
                    }
                    if (!err)
                        err = gs_match_submit_fasta(c.run, start, (int64_t)carry.size() + cut, lines, records, GS_MEM_HOST,
                                                    read_no + reads_in_file, kr ? rs.cls.data() : nullptr, kr ? rs.flags.data() : nullptr, &ticket);
                    if (!err) err = chunk_submitted(read_no + reads_in_file, records, false);
                    if (!err && kr && records > 0) {  // the per-read outputs of this chunk's records, before the block goes back
                        chunks.push_back({carry_file_off, reads_in_file, ticket});
                        err = check_refusal(&fallback_off, &fallback_reads);
                        chunks.pop_back();
                        if (!err && fallback_off < 0) err = outputs_general(rs, start, lines, records, true);
                    }
                }
                if (!err && fallback_off < 0) {
                    if (ticket >= 0) {
                        if (first_ticket < 0) first_ticket = ticket;
                        chunks.push_back({carry_file_off, reads_in_file, ticket});
                    }
                    reads_in_file += records;
                    carry_file_off = i * (int64_t)tr.block + cut;
                    carry.assign(blk + cut, blk + n);
                    carry_lines = tail_lines;
                    carry_headers = headers - cut_headers;
                    if (ticket >= 0) {
                        err = release_held();
                        held_ticket = ticket;
                        held_block = i;
                        keep_block = true;
                    }
                }
                if (!err && fallback_off < 0 && (chunks.size() == 1 || (chunks.size() & 15) == 0))
                    err = check_refusal(&fallback_off, &fallback_reads);
            }
        }
        if (!keep_block) tr.release(i);
        next_block = i + 1;
        if (err || last || fallback_off >= 0) err = finish(err, fallback_off, fallback_reads);
        *err_out = err;
        return 1;
    }

    // The general form of step(): everything up to the block's last newline goes to the device together with what the last chunk
    // left over; the device reports how many records END in it and how many bytes they cover.
    int step_general(bool blocking, int *err_out) {
        int err = GS_OK;
        const int64_t i = next_block;
        if (!blocking && !tr.is_full(i)) return 0;
        TextSlot &sl = tr.wait_full(i);
        int64_t fallback_off = -1, fallback_reads = 0;
        bool last = false;
        if (sl.io_error || !tr.verify_gzip(sl)) {
            err = hfail(tr.gz ? GS_E_INVALID : GS_E_IO, (tr.gz ? "corrupt gzip stream in " : "read error on ") + path);
        } else {
            err = gs_match_text_select(c.run, bank);
            uint8_t *blk = sl.buf + tr.headroom;
            const int64_t n = (int64_t)sl.n;
            last = sl.eof;
            if (err) {
                // (the bank could not be selected: reported below)
            } else if (sl.newlines == 0) {  // not one whole line: keep everything
                carry.insert(carry.end(), blk, blk + n);
                if (carry.size() > tr.headroom && !last) {
                    fallback_off = carry_file_off;
                    fallback_reads = reads_in_file;
                }
            } else if (carry.size() > tr.headroom) {  // a record longer than a block: the general parser takes over
                fallback_off = carry_file_off;
                fallback_reads = reads_in_file;
            } else {
                const int64_t cut = (int64_t)sl.last4[0] + 1;  // behind the block's last newline
                uint8_t *start = blk - carry.size();
                if (!carry.empty()) memcpy(start, carry.data(), carry.size());
                const int64_t bytes = (int64_t)carry.size() + cut, lines = carry_lines + sl.newlines;
                int64_t n_rec = 0, used = 0, used_lines = 0, ticket = -1;
                const bool kr = c.kraken.active() || c.filtered.active();
                MatchCtx::Results &rs = c.res[0];
                if (kr) {
                    err = rs.cls.resize((size_t)(lines / 4 + 2));
                    if (!err) err = rs.flags.resize((size_t)(lines / 4 + 2));
                }
                if (!err)
                    err = gs_match_submit_fastq_ml(c.run, start, bytes, lines, GS_MEM_HOST, read_no + reads_in_file, kr ? rs.cls.data() : nullptr,
                                                   kr ? rs.flags.data() : nullptr, &n_rec, &used, &used_lines, &ticket);
                if (!err && n_rec > 0) err = chunk_submitted(read_no + reads_in_file, n_rec, false);
                if (!err && kr && n_rec > 0) err = outputs_general(rs, start, used_lines, n_rec, false);
                if (!err && n_rec < 0) {  // refused (NUL byte, a record of thousands of lines): the general parser from here
                    err = gs_match_text_clear_error(c.run);
                    fallback_off = carry_file_off;
                    fallback_reads = reads_in_file;
                } else if (!err) {
                    g_ml_chunks.fetch_add(1);
                    if (first_ticket < 0) first_ticket = ticket;
                    reads_in_file += n_rec;
                    carry_file_off += used;
                    // what the records did not cover + what lies behind the last newline (the text has been copied)
                    std::vector<uint8_t> rest(start + used, start + bytes);
                    rest.insert(rest.end(), blk + cut, blk + n);
                    carry.swap(rest);
                    carry_lines = lines - used_lines;
                }
            }
        }
        tr.release(i);
        next_block = i + 1;
        if (err || last || fallback_off >= 0) err = finish(err, fallback_off, fallback_reads);
        *err_out = err;
        return 1;
    }

    // Per-read outputs of a FASTA or general FASTQ chunk that has just been matched.  Record geometry: the newline offsets from the
    // device and a class per line (1 descriptor, 2 sequence, 0 '+' / quality) -- from the device for general FASTQ, by the
    // first byte for FASTA.  Kraken-style lines (MatcherReadEntry.writeMatchDetails, :723-756): descriptor up to the first blank
    // without its first character, class, length, runs.  Filtered FASTQ: append_general_record.
    int outputs_general(MatchCtx::Results &rs, const uint8_t *text, int64_t n_lines, int64_t n_records, bool is_fasta) {
        std::vector<uint64_t> bounds((size_t)n_records + 1);
        std::vector<uint8_t> cls((size_t)std::max<int64_t>(n_lines, 1));
        int err = gs_match_text_read_bounds(c.run, bounds.data());  // (waits for the chunk: cls / flags are complete)
        if (!err) err = rs.nl.resize((size_t)std::max<int64_t>(n_lines, 1));
        if (!err) err = gs_match_text_newlines(c.run, rs.nl.data());
        if (!err && !is_fasta) err = gs_match_text_line_classes(c.run, cls.data());
        if (!err && c.kraken.active()) {
            err = rs.seg_off.resize((size_t)n_records + 1);
            if (!err) err = gs_match_segments_text(c.run, rs.seg_off.data());
            if (!err) err = rs.seg_code.resize((size_t)rs.seg_off[(size_t)n_records]);
            if (!err) err = rs.seg_start.resize((size_t)rs.seg_off[(size_t)n_records]);
            if (!err) err = gs_match_segments_fetch(c.run, rs.seg_code.data(), rs.seg_start.data());
        }
        if (err) return err;
        const uint32_t *nl = rs.nl.p;
        auto line_start = [nl](int64_t i) { return i ? (size_t)nl[i - 1] + 1 : (size_t)0; };
        if (is_fasta)
            for (int64_t i = 0; i < n_lines; i++) cls[(size_t)i] = text[line_start(i)] == '>' && nl[i] > line_start(i) ? 1 : 2;
        std::vector<int64_t> head;  // descriptor line of every record, + n_lines
        head.reserve((size_t)n_records + 1);
        for (int64_t i = 0; i < n_lines; i++)
            if (cls[(size_t)i] == 1) head.push_back(i);
        if ((int64_t)head.size() != n_records) return hfail(GS_E_INVALID, "text chunk: the descriptor lines do not match the device's record count");
        head.push_back(n_lines);
        std::vector<FormatPart> parts((size_t)c.pool.threads());
        MatchCtx &cc = c;
        const bool probs = c.opts->with_probs != 0 && !is_fasta;
        c.pool.run(n_records, [&](int t, int64_t lo, int64_t hi) {
            FormatPart &p = parts[(size_t)t];
            p.filtered = cc.filtered.take();
            p.kraken = cc.kraken.take();
            for (int64_t r = lo; r < hi; r++) {
                const int64_t h = head[(size_t)r], next = head[(size_t)r + 1];
                const size_t d0 = line_start(h), dlen = (size_t)nl[h] - d0;
                const int64_t L = (int64_t)(bounds[(size_t)r + 1] - bounds[(size_t)r]);
                if (cc.filtered.active() && (rs.flags[(size_t)r] & GS_F_RETURNED)) {
                    append_general_record(p.filtered, text, nl, cls.data(), h, next, L, is_fasta, probs);
                    p.n_filtered++;
                }
                if (cc.kraken.active()) kraken_line(cc, rs, p.kraken, text + d0, dlen, L, r);
            }
            p.pack(cc.kraken, cc.filtered);
        });
        write_parts(c, parts);
        return GS_OK;
    }

private:
    // filtered FASTQ (afterMatch, :304-307) and Kraken-style lines (:723-756) of the chunk that was just matched, from
    // the raw block: the device returns the record geometry (newline offsets) and the segments (fetch_chunk_results,
    // on the submitting thread), the lines are formatted and handed to the writers on a thread of their own
    // (format_chunk) while the next chunk is on the device; the chunk's block goes back to its reader afterwards
    int fetch_chunk_results(MatchCtx::Results &rs, int64_t n) {
        int err = rs.nl.resize((size_t)n * 4);
        if (!err) err = gs_match_text_newlines(c.run, rs.nl.data());
        if (!err && c.kraken.active()) {
            err = rs.seg_off.resize((size_t)n + 1);
            if (!err) err = gs_match_segments_text(c.run, rs.seg_off.data());
            if (!err) err = rs.seg_code.resize((size_t)rs.seg_off[(size_t)n]);
            if (!err) err = rs.seg_start.resize((size_t)rs.seg_off[(size_t)n]);
            if (!err) err = gs_match_segments_fetch(c.run, rs.seg_code.data(), rs.seg_start.data());
        }
        return err;
    }

    void format_chunk(const MatchCtx::Results &rs, const uint8_t *text, int64_t n, int64_t block) {
        std::vector<FormatPart> parts((size_t)c.pool.threads());
        MatchCtx &cc = c;
        const uint32_t *nl = rs.nl.p;
        c.pool.run(n, [&cc, &rs, &parts, text, nl](int t, int64_t lo, int64_t hi) {
            FormatPart &p = parts[(size_t)t];
            p.filtered = cc.filtered.take();
            p.kraken = cc.kraken.take();
            for (int64_t r = lo; r < hi; r++) {
                if (cc.filtered.active() && (rs.flags[(size_t)r] & GS_F_RETURNED)) {
                    append_text_record(p.filtered, text, nl, r, cc.opts->with_probs != 0);
                    p.n_filtered++;
                }
                if (cc.kraken.active()) {
                    const size_t d0 = r == 0 ? 0 : (size_t)nl[4 * r - 1] + 1, d1 = nl[4 * r];
                    kraken_line(cc, rs, p.kraken, text + d0, d1 - d0, (int64_t)nl[4 * r + 1] - (int64_t)d1 - 1, r);
                }
            }
            p.pack(cc.kraken, cc.filtered);
        });
        write_parts(c, parts);
        if (block >= 0) tr.release(block);  // (-1: the text was not a reader's block -- device-inflated input)
    }

    // waits for the chunk that is being formatted (its result set and its block are free afterwards)
    void drain() {
        if (formatting.valid()) formatting.get();
    }
    // the block whose copy to the device was still running when the thread went on
    int release_held() {
        if (held_ticket < 0) return GS_OK;
        const int err = gs_match_text_wait_copy(c.run, held_ticket);
        tr.release(held_block);
        held_ticket = -1;
        return err;
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
        drain();
        if (!err) err = dev_err_;
        const int held_err = release_held();  // (the blocks return to the pool in close(): no copy may still read them)
        if (!err) err = held_err;
        release_gunzipper();  // (parks the upload thread: it reads the mapping that close() removes)
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
            // a FASTQ file that is not four lines per record from its very first chunk: once more with the records found on the
            // device (GS_HOST_ML=0: straight to the reference-exact parser, which also takes over whatever that pass refuses)
            bool ml = fallback_off == 0 && fallback_reads == 0 && !fasta && !general;
            if (const char *e = getenv("GS_HOST_ML")) ml = ml && atoi(e) != 0;
            if (ml) {
                TextJob g(c, path, bank, read_no, false);
                g.general = true;
                int gerr = g.open(gz_, readers_);
                while (!gerr && !g.done) g.step(true, &gerr);
                if (!g.done) g.abort();
                read_no = g.read_no;
                return gerr;
            }
            return parsed_source(c, path, fallback_off, nullptr, 0, read_no);
        }
        read_no += reads_in_file;
        // what is left after the last whole four-line group (no final newline, truncated record): the general parser
        if (!carry.empty()) return parsed_source(c, std::string(), 0, carry.data(), carry.size(), read_no, fasta);
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
              bool *composite, bool allow_side_by_side = true) {
    bool fast = true;
    if (const char *e = getenv("GS_HOST_FAST")) fast = atoi(e) != 0;
    int err = GS_OK;
    std::vector<int> kind((size_t)n_paths, 0);
    int n_gzip = 0;
    for (int i = 0; i < n_paths; i++) {
        kind[(size_t)i] = fast ? text_path_kind(paths[i]) : 0;
        n_gzip += kind[(size_t)i] == 2 || kind[(size_t)i] == 4;
    }
    const int default_readers = (int)std::min<unsigned>(8, std::max<unsigned>(2, std::thread::hardware_concurrency() / 2));
    // one gzip file at a time: its inflating threads are all the parallelism there is (measured on the MI355X box,
    // tools/gz_threads_sweep.py: 8 threads 1.7 Gbp/s, 12 2.3, 16 2.9, 24 3.3)
    const int gzip_threads = (int)std::min<unsigned>(16, std::max<unsigned>(2, std::thread::hardware_concurrency() / 2));
    // Several gzip files: each is bound by its single inflating thread, so they are read side by side (up to 8 at a
    // time).  The read numbers of file f then start at f << 32, which keeps "first read in file order" (the max-contig
    // tie-break) intact; the column is converted back to running read numbers at the end.
    // (per-read outputs follow the read order: one file after the other)
    bool side_by_side = n_gzip >= 2 && n_paths <= 256 && !c.filtered.active() && !c.kraken.active();
    if (const char *e = getenv("GS_HOST_PARALLEL_FILES")) side_by_side = side_by_side && atoi(e) != 0;
    side_by_side = side_by_side && allow_side_by_side;
    if (file_index) side_by_side = true;  // read numbers are (file << 32 | read): the files are independent anyway
    std::vector<int64_t> reads_of_file((size_t)n_paths, 0);
    if (!side_by_side) {
        int64_t read_no = 0;
        for (int i = 0; i < n_paths && !err; i++) {
            const std::string path(paths[i]);
            if (kind[(size_t)i]) {
                const bool gz = kind[(size_t)i] == 2 || kind[(size_t)i] == 4;
                TextJob job(c, path, 0, read_no, kind[(size_t)i] >= 3);
                err = job.open(gz, gz ? gzip_threads : default_readers);
                while (!err && !job.done) job.step(true, &err);
                if (!job.done) job.abort();
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
                    auto job = std::make_unique<TextJob>(c, std::string(paths[next]), bank, base, kind[(size_t)next] >= 3);
                    err = job->open(kind[(size_t)next] == 2 || kind[(size_t)next] == 4, std::max(2, 2 * default_readers / std::min(n_paths, 8)));
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
        for (auto &j : active) j->abort();
    }
    reads_of_file_out = reads_of_file;
    *composite = side_by_side;
    return err;
}

}  // namespace

extern "C" int gs_host_match_into(gs_run *run, gs_db *db, const char *const *paths, int n_paths, const int32_t *file_index,
                                  int64_t *reads_of_file, gs_host_totals *totals);

extern "C" int gs_host_match_files(gs_db *db, const gs_match_cfg *cfg, const char *const *paths, int n_paths,
                                   const gs_host_match_opts *opts, int64_t *table, double *dtable,
                                   gs_host_totals *totals) try {
    if (!db || !cfg || !paths || n_paths < 0 || !table) return hfail(GS_E_INVALID, "NULL argument");
    MatchCtx c;
    int rc = gs_db_get_info(db, &c.info);
    if (rc) return rc;
    const gs_host_match_opts none{};
    if (!opts) opts = &none;
    c.opts = opts;
    if (opts->kraken_out_path && !opts->taxids) return hfail(GS_E_INVALID, "Kraken-style output needs the taxid strings");
    if (opts->kraken_out_path) {
        c.taxid_len.resize((size_t)c.info.n_values);
        for (int32_t v = 0; v < c.info.n_values; v++) {
            if (!opts->taxids[v]) return hfail(GS_E_INVALID, "Kraken-style output: a taxid string is NULL");
            c.taxid_len[(size_t)v] = (uint32_t)strlen(opts->taxids[v]);
            c.taxid_max = std::max(c.taxid_max, (size_t)c.taxid_len[(size_t)v]);
        }
    }
    if (c.track_desc()) memset(opts->max_contig_desc, 0, (size_t)c.info.n_values * (size_t)opts->max_contig_desc_stride);
    if (!c.filtered.open(opts->filtered_path) || !c.kraken.open(opts->kraken_out_path)) return hfail(GS_E_INVALID, "cannot open output file");
    const double t_begin = now_s();
    rc = gs_match_begin(&c.run, db, cfg);
    if (rc) return rc;
    const double t_start = now_s();
    std::vector<int64_t> reads_of_file;
    bool side_by_side = false;
    int err = run_files(c, paths, n_paths, nullptr, reads_of_file, &side_by_side);
    const double t_files = now_s();
    if (!err) err = gs_match_finish(c.run, table, dtable);
    const double t_fin = now_s();
    if (!err && side_by_side) {  // (file << 32 | read in file) -> running read number over the files in order
        std::vector<int64_t> before((size_t)n_paths + 1, 0);
        for (int i = 0; i < n_paths; i++) before[(size_t)i + 1] = before[(size_t)i] + reads_of_file[(size_t)i];
        for (int32_t v = 0; v < c.info.n_values; v++) {
            int64_t &x = table[(size_t)v * GS_N_COLS + GS_C_MAX_CONTIG_READ_NO];
            if (x >= 0) x = before[(size_t)(x >> 32)] + (x & 0xffffffffLL);
        }
    }
    gs_match_destroy(c.run);
    const bool wrote = c.filtered.close() & c.kraken.close();  // (both are flushed before the clock stops)
    if (!err) err = c.filtered_dev.late_err;
    if (getenv("GS_HOST_TRACE") != nullptr)
        fprintf(stderr, "match files: begin %.2f ms, files %.2f, finish %.2f, destroy + close %.2f\n", (t_start - t_begin) * 1e3, (t_files - t_start) * 1e3, (t_fin - t_files) * 1e3,
                (now_s() - t_fin) * 1e3);
    if (!err && !wrote) err = hfail(GS_E_IO, "write to an output file failed");
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
} catch (const std::bad_alloc &) {
    return hfail(GS_E_NOMEM, "out of host memory");
} catch (const std::exception &e) {  // (nothing may leave through the C ABI)
    return hfail(GS_E_INVALID, std::string("unexpected exception: ") + e.what());
}


// runMatcher's body for a host that keeps its own run (begin / reset ... finish): the files into `run`, per-read outputs included
extern "C" int gs_host_match_run(gs_run *run, gs_db *db, const char *const *paths, int n_paths, const gs_host_match_opts *opts,
                                 gs_host_totals *totals) try {
    if (!run || !db || !paths || n_paths < 0) return hfail(GS_E_INVALID, "NULL argument");
    MatchCtx c;
    int rc = gs_db_get_info(db, &c.info);
    if (rc) return rc;
    const gs_host_match_opts none{};
    if (!opts) opts = &none;
    c.opts = opts;
    if (opts->kraken_out_path && !opts->taxids) return hfail(GS_E_INVALID, "Kraken-style output needs the taxid strings");
    if (opts->kraken_out_path) {
        c.taxid_len.resize((size_t)c.info.n_values);
        for (int32_t v = 0; v < c.info.n_values; v++) {
            if (!opts->taxids[v]) return hfail(GS_E_INVALID, "Kraken-style output: a taxid string is NULL");
            c.taxid_len[(size_t)v] = (uint32_t)strlen(opts->taxids[v]);
            c.taxid_max = std::max(c.taxid_max, (size_t)c.taxid_len[(size_t)v]);
        }
    }
    if (c.track_desc()) memset(opts->max_contig_desc, 0, (size_t)c.info.n_values * (size_t)opts->max_contig_desc_stride);
    if (!c.filtered.open(opts->filtered_path) || !c.kraken.open(opts->kraken_out_path)) return hfail(GS_E_INVALID, "cannot open output file");
    c.run = run;
    const double t_start = now_s();
    std::vector<int64_t> reads_of_file;
    bool composite = false;
    int err = run_files(c, paths, n_paths, nullptr, reads_of_file, &composite, false);  // (one file after the other: running read numbers)
    if (!err) err = gs_match_sync(run);
    const bool wrote = c.filtered.close() & c.kraken.close();
    if (!err) err = c.filtered_dev.late_err;
    if (!err && !wrote) err = hfail(GS_E_IO, "write to an output file failed");
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
} catch (const std::bad_alloc &) {
    return hfail(GS_E_NOMEM, "out of host memory");
} catch (const std::exception &e) {  // (nothing may leave through the C ABI)
    return hfail(GS_E_INVALID, std::string("unexpected exception: ") + e.what());
}

// runMatcher over the files of a sample on SEVERAL devices of this process: dbs[d] = a replica of the store on device d
// (the same arrays through gs_db_create, or the same store file).  File i goes to replica i % n_dbs; every replica has its
// own run and its own worker thread (reader pool, device text path, as gs_host_match_into); read numbers are
// (file << 32 | read), so that the max-contig tie-break keeps the file order; the runs are merged by gs_match_merge
// (kernels on one device, RCCL between devices) and finished once.  No per-read outputs (they would interleave).
extern "C" int gs_host_match_files_multi(gs_db *const *dbs, int n_dbs, const gs_match_cfg *cfg, const char *const *paths,
                                         int n_paths, int64_t *table, double *dtable, gs_host_totals *totals) try {
    if (!dbs || n_dbs < 1 || !cfg || !paths || n_paths < 0 || !table) return hfail(GS_E_INVALID, "NULL argument");
    if (n_paths > GS_HOST_MAX_FILE_INDEX) return hfail(GS_E_UNSUPPORTED, "at most 256 files per call");
    const double t_start = now_s();
    std::vector<gs_run *> runs((size_t)n_dbs, nullptr);
    int err = GS_OK;
    for (int d = 0; d < n_dbs && !err; d++) {
        if (!dbs[d]) err = hfail(GS_E_INVALID, "a store is NULL");
        if (!err) err = gs_match_begin(&runs[(size_t)d], dbs[d], cfg);
    }
    std::vector<std::vector<int32_t>> mine((size_t)n_dbs);
    for (int i = 0; i < n_paths; i++) mine[(size_t)(i % n_dbs)].push_back(i);
    std::vector<int64_t> reads_of_file((size_t)n_paths, 0);
    std::vector<gs_host_totals> tot((size_t)n_dbs);
    std::vector<int> rcs((size_t)n_dbs, GS_OK);
    std::vector<std::string> msgs((size_t)n_dbs);
    if (!err) {
        std::vector<std::thread> th;
        for (int d = 0; d < n_dbs; d++)
            th.emplace_back([&, d] {
                const std::vector<int32_t> &idx = mine[(size_t)d];
                if (idx.empty()) return;
                std::vector<const char *> p;
                for (int32_t i : idx) p.push_back(paths[i]);
                std::vector<int64_t> rof(idx.size(), 0);
                rcs[(size_t)d] = gs_host_match_into(runs[(size_t)d], dbs[d], p.data(), (int)p.size(), idx.data(), rof.data(), &tot[(size_t)d]);
                if (rcs[(size_t)d]) msgs[(size_t)d] = gs_host_last_error();  // (the message is per thread)
                for (size_t x = 0; x < idx.size(); x++) reads_of_file[(size_t)idx[x]] = rof[x];
            });
        for (auto &t : th) t.join();
        for (int d = 0; d < n_dbs && !err; d++)
            if (rcs[(size_t)d]) err = hfail(rcs[(size_t)d], msgs[(size_t)d]);
    }
    if (!err) err = gs_match_merge(runs.data(), n_dbs);
    if (!err) err = gs_match_finish(runs[0], table, dtable);
    if (!err) {  // (file << 32 | read in file) -> running read number over the files in order
        gs_db_info info{};
        gs_db_get_info(dbs[0], &info);
        std::vector<int64_t> before((size_t)n_paths + 1, 0);
        for (int i = 0; i < n_paths; i++) before[(size_t)i + 1] = before[(size_t)i] + reads_of_file[(size_t)i];
        for (int32_t v = 0; v < info.n_values; v++) {
            int64_t &x = table[(size_t)v * GS_N_COLS + GS_C_MAX_CONTIG_READ_NO];
            if (x >= 0) x = before[(size_t)(x >> 32)] + (x & 0xffffffffLL);
        }
    }
    for (gs_run *r : runs)
        if (r) gs_match_destroy(r);
    if (totals) {
        *totals = gs_host_totals{};
        for (const gs_host_totals &t : tot) {
            totals->reads += t.reads;
            totals->kmers += t.kmers;
            totals->bps += t.bps;
            totals->seconds_parse += t.seconds_parse;
            totals->seconds_gpu += t.seconds_gpu;
        }
        totals->seconds_total = now_s() - t_start;
    }
    return err;
} catch (const std::bad_alloc &) {
    return hfail(GS_E_NOMEM, "out of host memory");
} catch (const std::exception &e) {  // (nothing may leave through the C ABI)
    return hfail(GS_E_INVALID, std::string("unexpected exception: ") + e.what());
}

// The same, into a run the caller began and will finish: for one-process-per-GPU runs that share the files of a sample
// (genestrip_amd/distributed.py: match_files_sharded) -- every process takes some of the files, merges the device state
// of its run with the others (gs_match_device_state) and finishes.  file_index[n_paths] = position of each file in the
// global file order; the read numbers handed to the device are (file_index << 32 | read in file), reads_of_file[n_paths]
// receives the read counts (needed to turn the max-contig read numbers into running ones after the merge).
extern "C" int gs_host_match_into(gs_run *run, gs_db *db, const char *const *paths, int n_paths, const int32_t *file_index,
                                  int64_t *reads_of_file, gs_host_totals *totals) try {
    if (!run || !db || !paths || n_paths < 0 || !file_index || !reads_of_file) return hfail(GS_E_INVALID, "NULL argument");
    // the max-contig key keeps 40 bits of the read number (gs_kernels.hip: key_lo): 8 of them are the file, 32 the read
    for (int i = 0; i < n_paths; i++)
        if (file_index[i] < 0 || file_index[i] >= GS_HOST_MAX_FILE_INDEX)
            return hfail(GS_E_UNSUPPORTED, "file_index must be in [0, 256): read numbers are (file << 32 | read) in a 40-bit field");
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
} catch (const std::bad_alloc &) {
    return hfail(GS_E_NOMEM, "out of host memory");
} catch (const std::exception &e) {  // (nothing may leave through the C ABI)
    return hfail(GS_E_INVALID, std::string("unexpected exception: ") + e.what());
}

namespace {

struct FilterCtx {
    gs_bloom *bloom = nullptr;
    int k = 31, min_pos_count = 1;
    double positive_ratio = 0.2;
    bool with_probs = false;
    OutFile acc_out, rest_out;
    std::vector<uint8_t> accept;
    int64_t accepted = 0, reads = 0, kmers = 0, bps = 0;
    double t_gpu = 0, t_parse = 0;
    FormatPool pool{format_threads()};
    DeviceWriter acc_dev, rest_dev;  // (declared behind the files: they wait for their writes before the files close)
};

struct FilterPart {
    std::vector<uint8_t> acc, rest;
    bool acc_packed = false, rest_packed = false;
    int64_t n_accepted = 0;
    void pack(OutFile &a, OutFile &r) {
        const size_t worth_it = (size_t)64 << 10;
        acc_packed = a.gzip() && acc.size() >= worth_it && a.pack(acc);
        rest_packed = r.gzip() && rest.size() >= worth_it && r.pack(rest);
    }
};

void write_filter_parts(FilterCtx &c, std::vector<FilterPart> &parts) {
    for (FilterPart &p : parts) {
        c.accepted += p.n_accepted;
        c.acc_out.write(std::move(p.acc), p.acc_packed);
        c.rest_out.write(std::move(p.rest), p.rest_packed);
    }
}

// the general path for one source (file from `offset`, or a memory range): reference parser -> batches -> GPU -> writers
int filter_parsed_source(FilterCtx &c, const std::string &path, int64_t offset, const uint8_t *mem, size_t mem_n, bool mem_fasta = false) {
    Producer prod;
    prod.start(path, offset, mem, mem_n, c.k, (int64_t)1 << 20, mem_fasta);
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
        std::vector<FilterPart> parts((size_t)c.pool.threads());
        const Batch &bb = *b;
        c.pool.run(n, [&](int t, int64_t lo, int64_t hi) {
            FilterPart &p = parts[(size_t)t];
            p.acc = c.acc_out.take();
            p.rest = c.rest_out.take();
            for (int64_t i = lo; i < hi; i++) {  // nextEntry (FastqBloomFilter.java:92-105), input order
                if (c.accept[(size_t)i]) {
                    p.n_accepted++;
                    if (c.acc_out.active()) append_read(p.acc, bb, i, c.with_probs);
                } else if (c.rest_out.active())
                    append_read(p.rest, bb, i, c.with_probs);
            }
            p.pack(c.acc_out, c.rest_out);
        });
        write_filter_parts(c, parts);
    }
    prod.th.join();
    if (!err && !prod.error.empty()) err = hfail(GS_E_INVALID, prod.error);
    c.reads += prod.reads;
    c.kmers += prod.kmers;
    c.bps += prod.bps;
    c.t_parse += prod.seconds;
    return err;
}

// block size and reader / inflating threads of the filter goal's text pipelines (GS_HOST_BLOCK_BYTES, GS_HOST_READERS)
void filter_reader_shape(bool gzip, size_t *block, int *readers) {
    *block = (size_t)8 << 20;
    if (const char *e = getenv("GS_HOST_BLOCK_BYTES")) {
        const long long v = atoll(e);
        if (v >= 64 && v <= ((long long)1 << 29)) *block = (size_t)v;
    }
    *readers = (int)std::min<unsigned>(gzip ? 16 : 8, std::max<unsigned>(2, std::thread::hardware_concurrency() / 2));
    if (const char *e = getenv("GS_HOST_READERS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 32) *readers = v;
    }
}

int filter_general_file(FilterCtx &c, const std::string &path, bool gzip, bool fasta);

// the records of one chunk of four-line FASTQ to the writers: nextEntry (FastqBloomFilter.java:92-105), input order
void format_text_chunk(FilterCtx &c, const uint8_t *start, const uint8_t *h_acc, const uint32_t *h_nl, int64_t n_reads) {
    std::vector<FilterPart> parts((size_t)c.pool.threads());
    c.pool.run(n_reads, [&](int t, int64_t lo, int64_t hi) {
        FilterPart &p = parts[(size_t)t];
        p.acc = c.acc_out.take();
        p.rest = c.rest_out.take();
        for (int64_t r = lo; r < hi; r++) {
            if (h_acc[r]) {
                p.n_accepted++;
                if (c.acc_out.active()) append_text_record(p.acc, start, h_nl, r, c.with_probs);
            } else if (c.rest_out.active())
                append_text_record(p.rest, start, h_nl, r, c.with_probs);
        }
        p.pack(c.acc_out, c.rest_out);
    });
    write_filter_parts(c, parts);
}

// Block-gzip (BGZF) FASTQ: the members are listed from their headers, the COMPRESSED bytes go to the device and are inflated there
// (gs_inflater_feed), the filter runs on the text where it lies (GS_MEM_DEVICE_TEXT), and the text comes back ONCE, page-locked, for
// the writers -- while the filter kernel runs.  *handled = false: not (only) BGZF, or no inflater: the caller takes its usual path.
int filter_bgzf_file(FilterCtx &c, const std::string &path, bool *handled) {
    *handled = false;
    if (const char *e = getenv("GS_DEVICE_INFLATE"))
        if (atoi(e) == 0) return GS_OK;
    size_t block;
    int readers;
    filter_reader_shape(true, &block, &readers);
    TextReader tr;  // (for the mapping only: its readers are never started)
    int err = tr.open(path, block, readers, true);
    std::vector<gs_inflate_member> members;
    int device = 0;
    gs_inflater *inf = nullptr;
    // (not block-gzip: a single-member stream is inflated on the device as a whole, gs_gunzip_plan_device, and handed on in slices)
    gs_gunzipper *gzr = nullptr;
    const uint8_t *gz_text = nullptr;  // the current batch
    int64_t gz_n = 0, gz_off = 0;
    int gz_last = 0;
    if (err || tr.map_len < 18 || gs_filter_get_device(c.bloom, &device) != GS_OK) {
        tr.close();
        return GS_OK;
    }
    if (tr.map_len >= 28 && bgzf_member_list(tr.map, tr.map_len, members)) {
        if ((inf = inflater_pool().get(device)) == nullptr) {
            tr.close();
            return GS_OK;
        }
    } else {
        members.clear();
        bool want = true;
        if (const char *e = getenv("GS_DEVICE_GUNZIP")) want = atoi(e) != 0;
        if (!want || gunzipper_pool().open(&gzr, device, tr.map, (int64_t)tr.map_len) != GS_OK || gs_gunzipper_first_span(gzr, gunzip_first_span()) != GS_OK ||
            gs_gunzipper_next(gzr, 0, &gz_text, &gz_n, &gz_last) != GS_OK) {
            gunzipper_pool().put(device, gzr);
            tr.close();
            return GS_OK;  // (a stream this path does not take, a damaged one: the host decoders take it -- and report it)
        }
    }
    const bool whole = inf == nullptr;
    *handled = true;
    err = gs_filter_text_reset(c.bloom, 1);
    PooledBuf text_sets[2], nl_sets[2];
    PinnedVec<uint8_t> acc_sets[2];
    std::future<void> formatting;
    std::future<int> dev_job;  // device output: gather -> (deflate) -> fetch -> writer of the chunk before
    c.acc_dev.begin(&c.acc_out, device);
    c.rest_dev.begin(&c.rest_out, device);
    int64_t n_formatted = 0, text_off = 0, fallback_off = -1;
    int64_t tot[3] = {0, 0, 0}, failed = -1, bad = -1;
    std::vector<uint8_t> carry;
    const double t0 = now_s();
    // (feeds of 128 MiB here, not 512: the writers get their first chunk four times earlier, and what they have not written when
    // the last feed is through is what the file waits for in the end -- 4 M reads: 285 ms with 512 MiB feeds, 175 ms with 128)
    // (with the writers' side on the device -- nothing to format, a sixth of the bytes to write -- the feeds are 256 MiB: 12.1 against
    // 11.2 Gbp/s gz -> gz at 16 M reads)
    const bool dev_out = device_output();
    const int64_t text_target = getenv("GS_HOST_BGZF_TEXT") ? bgzf_text_target() : ((int64_t)(dev_out ? 256 : 128) << 20);
    auto run_end = [&](size_t from) {
        int64_t sum = 0;
        size_t e = from;
        while (e < members.size() && (e == from || sum + members[e].isize <= text_target)) sum += members[e++].isize;
        return e;
    };
    for (size_t a = 0; !err && (whole || a < members.size());) {
        bool last = false;
        const uint8_t *text = nullptr;
        int64_t n_bytes = 0, n_lines = 0, tail = 0;
        const double tg = now_s();
        if (whole) {  // the next slice of the device text: whole records up to the feed size
            bool refused = false;
            for (;;) {  // (a slice with a whole record in it: from this batch, or with the next one behind what is left of this)
                const int64_t rest = gz_n - gz_off, look = std::min(rest, text_target);
                n_lines = n_bytes = 0;
                if (look > 0 && gs_text_cut_device(device, gz_text + gz_off, look, &n_lines, &n_bytes) != GS_OK) {
                    err = hfail(GS_E_HIP, gs_inflate_last_error());
                    break;
                }
                if (n_lines > 0 || gz_last || look < rest) break;
                const int grc = gs_gunzipper_next(gzr, rest, &gz_text, &gz_n, &gz_last);  // (every earlier slice has been waited for: gs_filter_text_status)
                gz_off = 0;
                if (grc == GS_E_UNSUPPORTED || grc == GS_E_NOMEM) {
                    refused = true;
                    gz_n = 0;
                    break;
                }
                if (grc != GS_OK) {
                    err = hfail(GS_E_INVALID, std::string("corrupt gzip stream in ") + path + ": " + gs_inflate_last_error());
                    break;
                }
            }
            if (err) break;
            const int64_t rest = gz_n - gz_off;
            last = gz_last != 0 && std::min(rest, text_target) == rest;
            text = gz_text + gz_off;
            tail = last ? rest - n_bytes : (n_lines > 0 ? 0 : ((int64_t)1 << 40));  // (no record in a full slice: the general parser, below)
            if (refused) {  // the host decoders from here: a batch the device path does not take
                fallback_off = text_off;
                break;
            }
            gz_off += n_bytes;
        } else {
            const size_t b = run_end(a), b2 = run_end(b);
            last = b == members.size();
            int64_t next_lo = 0, next_hi = 0;
            if (b2 > b) {
                next_lo = members[b].payload_offset;
                next_hi = members[b2 - 1].payload_offset + (int64_t)members[b2 - 1].payload_len;
            }
            if (gs_inflater_feed(inf, tr.map, members.data() + a, (int64_t)(b - a), next_lo, next_hi, last ? 1 : 0, &text, &n_bytes, &n_lines, &tail) != GS_OK) {
                err = hfail(GS_E_INVALID, std::string("corrupt gzip stream in ") + path + ": " + gs_inflate_last_error());
                break;
            }
            a = b;
        }
        if (n_lines > 0 && dev_out) {
            // The writers' side stays on the device: the records each file wants are gathered there (gs_filter_compact_text), a .gz
            // file's are compressed there (DeviceWriter::emit -> gs_deflater_pack), and only what the files will hold crosses PCIe --
            // on a thread of its own, while the next feed is inflated and filtered.
            const int64_t n_reads = n_lines >> 2;
            const int set = (int)(n_formatted & 1);
            if ((err = acc_sets[set].resize((size_t)n_reads))) break;
            uint8_t *h_acc = acc_sets[set].data();
            int64_t ticket = -1;
            static const bool trace = getenv("GS_HOST_TRACE") != nullptr;
            const double t1 = now_s();
            err = gs_filter_submit_text(c.bloom, c.k, c.min_pos_count, c.positive_ratio, text, n_bytes, n_lines, GS_MEM_DEVICE_TEXT, h_acc, nullptr, 0, &ticket);
            if (!err) err = gs_filter_text_status(c.bloom, &failed, &bad, tot);  // synchronises: results are needed now
            const double t2 = now_s();
            c.t_gpu += t2 - tg;
            if (err) break;
            if (failed >= 0) {  // not four-line FASTQ from here on: the general parser continues at this chunk
                fallback_off = text_off;
                break;
            }
            const uint8_t *d_a = nullptr, *d_r = nullptr;
            int64_t nb_a = 0, nr_a = 0, nb_r = 0, nr_r = 0;
            if (c.acc_out.active()) err = gs_filter_compact_text(c.bloom, 1, c.with_probs ? 1 : 0, set, &d_a, &nb_a, &nr_a);
            if (!err && c.rest_out.active()) err = gs_filter_compact_text(c.bloom, 0, c.with_probs ? 1 : 0, set, &d_r, &nb_r, &nr_r);
            if (err) break;
            if (c.acc_out.active())
                c.accepted += nr_a;
            else if (c.rest_out.active())
                c.accepted += n_reads - nr_r;
            else
                for (int64_t r = 0; r < n_reads; r++) c.accepted += h_acc[r] != 0;
            const double t3 = now_s();
            if (dev_job.valid() && (err = dev_job.get())) break;  // one chunk at a time: output order
            if (trace)
                fprintf(stderr, "filter feed (device output): %lld bytes, inflate %.2f ms, filter %.2f, gather %.2f (%lld + %lld bytes), writers of the chunk before %.2f\n",
                        (long long)n_bytes, (t1 - tg) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (long long)nb_a, (long long)nb_r, (now_s() - t3) * 1e3);
            n_formatted++;
            auto job = [&c, set, d_a, nb_a, d_r, nb_r]() -> int {
                const int e1 = c.acc_dev.emit(set, d_a, nb_a);
                const int e2 = c.rest_dev.emit(set, d_r, nb_r);
                return e1 ? e1 : e2;
            };
            try {
                dev_job = std::async(std::launch::async, job);
            } catch (const std::system_error &) {  // no thread to be had: on this one
                if ((err = job())) break;
            }
            text_off += n_bytes;
        } else if (n_lines > 0) {
            const int64_t n_reads = n_lines >> 2;
            const int set = (int)(n_formatted & 1);  // (the set of the chunk before last: its writers are through)
            if ((err = acc_sets[set].resize((size_t)n_reads)) || (err = nl_sets[set].need(sizeof(uint32_t) * (size_t)n_lines)) || (err = text_sets[set].need((size_t)n_bytes)))
                break;
            uint8_t *h_acc = acc_sets[set].data(), *h_text = static_cast<uint8_t *>(text_sets[set].p);
            uint32_t *h_nl = static_cast<uint32_t *>(nl_sets[set].p);
            int64_t ticket = -1;
            static const bool trace = getenv("GS_HOST_TRACE") != nullptr;
            const double t1 = now_s();
            err = gs_filter_submit_text(c.bloom, c.k, c.min_pos_count, c.positive_ratio, text, n_bytes, n_lines, GS_MEM_DEVICE_TEXT, h_acc, h_nl, 0, &ticket);
            const double t2 = now_s();
            if (!err && (whole ? gs_device_fetch(device, text, h_text, n_bytes) : gs_inflater_fetch(inf, h_text, n_bytes)) != GS_OK)
                err = hfail(GS_E_HIP, gs_inflate_last_error());  // (while the kernel runs)
            const double t3 = now_s();
            if (!err) err = gs_filter_text_status(c.bloom, &failed, &bad, tot);  // synchronises: results are needed now
            const double t4 = now_s();
            c.t_gpu += t4 - tg;
            if (err) break;
            if (failed >= 0) {  // not four-line FASTQ from here on: the general parser continues at this chunk
                fallback_off = text_off;
                break;
            }
            if (formatting.valid()) formatting.get();  // one chunk at a time: output order, the other set is free
            if (trace)
                fprintf(stderr, "filter bgzf feed: %lld bytes, inflate + buffers %.2f ms, submit %.2f, text back %.2f, status %.2f, writers of the chunk before %.2f\n",
                        (long long)n_bytes, (t1 - tg) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (now_s() - t4) * 1e3);
            n_formatted++;
            auto write_chunk = [&c, h_text, h_acc, h_nl, n_reads] { format_text_chunk(c, h_text, h_acc, h_nl, n_reads); };
            try {
                formatting = std::async(std::launch::async, write_chunk);
            } catch (const std::system_error &) {  // no thread to be had: on this one
                write_chunk();
            }
            text_off += n_bytes;
        } else if (tail > ((int64_t)256 << 20) && !last) {  // no record boundary in a quarter of a gigabyte: the general parser
            fallback_off = text_off;
            break;
        }
        if (last) {  // what is left behind the last whole record
            int64_t n = 0;
            carry.resize((size_t)tail);
            if (tail > 0 && (whole ? gs_device_fetch(device, gz_text + gz_off, carry.data(), tail) : gs_inflater_tail(inf, carry.data(), tail, &n)) != GS_OK)
                err = hfail(GS_E_HIP, gs_inflate_last_error());
            break;
        }
    }
    const double te0 = now_s();
    if (formatting.valid()) formatting.get();
    if (dev_job.valid()) {
        const int e2 = dev_job.get();
        if (!err) err = e2;
    }
    const double te1 = now_s();
    if (inf) inflater_pool().put(device, inf);
    if (gzr) gunzipper_pool().put(device, gzr);  // (every slice's filter run has been waited for: gs_filter_text_status)
    tr.close();
    if (getenv("GS_HOST_TRACE") != nullptr)
        fprintf(stderr, "filter bgzf: loop %.2f ms (from open), last writers %.2f, inflater back + unmap %.2f\n", (te0 - t0) * 1e3, (te1 - te0) * 1e3, (now_s() - te1) * 1e3);
    c.t_parse += now_s() - t0;
    if (err) return err;
    c.reads += tot[0];
    c.kmers += tot[1];
    c.bps += tot[2];
    if (fallback_off >= 0) {
        err = gs_filter_text_reset(c.bloom, 1);
        if (err) return err;
        // not four lines per record from the very first chunk: once more with the records found on the device, as filter_text_file
        bool ml = fallback_off == 0;
        if (const char *e = getenv("GS_HOST_ML")) ml = ml && atoi(e) != 0;
        if (ml) return filter_general_file(c, path, true, false);
        return filter_parsed_source(c, path, fallback_off, nullptr, 0);
    }
    if (!carry.empty()) return filter_parsed_source(c, std::string(), 0, carry.data(), carry.size());
    return GS_OK;
}

// plain FASTQ: raw text blocks to the device (gs_filter_submit_text); accept flags and record geometry come back
int filter_text_file(FilterCtx &c, const std::string &path, bool gzip) {
    size_t block;
    int readers;
    filter_reader_shape(gzip, &block, &readers);
    TextReader tr;
    int err = tr.open(path, block, readers, gzip);
    if (err) {
        tr.close();
        return err;
    }
    err = gs_filter_text_reset(c.bloom, 1);
    // results of a chunk land in pinned memory: accept flags + newline offsets; two sets, so that the writers can work
    // on one chunk (on a thread of their own) while the device is busy with the next
    PinnedVec<uint8_t> acc_sets[2];
    PinnedVec<uint32_t> nl_sets[2];
    std::future<void> formatting;
    std::future<int> dev_job;
    // (plain outputs from a plain file are formatted from the reader's page-locked block, which is on the host anyway)
    int device = 0;
    const bool dev_out = device_output() && ((c.acc_out.active() && c.acc_out.gzip()) || (c.rest_out.active() && c.rest_out.gzip())) &&
                         gs_filter_get_device(c.bloom, &device) == GS_OK;
    c.acc_dev.begin(&c.acc_out, device);
    c.rest_dev.begin(&c.rest_out, device);
    int64_t n_formatted = 0;
    std::vector<uint8_t> carry;
    int64_t carry_lines = 0, carry_file_off = 0, fallback_off = -1;
    int64_t tot[3] = {0, 0, 0}, failed = -1, bad = -1;
    const double t0 = now_s();
    if (!err) tr.start();
    for (int64_t i = 0; !err; i++) {
        TextSlot &sl = tr.wait_full(i);
        if (sl.io_error || !tr.verify_gzip(sl)) {
            err = hfail(tr.gz ? GS_E_INVALID : GS_E_IO, (tr.gz ? "corrupt gzip stream in " : "read error on ") + path);
            break;
        }
        uint8_t *blk = sl.buf + tr.headroom;
        const int64_t total = carry_lines + sl.newlines;
        const int64_t rem = total & 3, usable = total - rem;
        const bool eof = sl.eof;
        bool keep_block = false;
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
            if ((err = acc_sets[n_formatted & 1].resize((size_t)n_reads))) break;  // (the set of the chunk before last)
            if ((err = nl_sets[n_formatted & 1].resize((size_t)usable))) break;
            uint8_t *h_acc = acc_sets[n_formatted & 1].data();
            uint32_t *h_nl = nl_sets[n_formatted & 1].data();
            int64_t ticket = -1;
            const double tg = now_s();
            err = gs_filter_submit_text(c.bloom, c.k, c.min_pos_count, c.positive_ratio, start, (int64_t)carry.size() + cut + 1,
                                        usable, GS_MEM_HOST, h_acc, h_nl, 0, &ticket);
            if (!err) err = gs_filter_text_status(c.bloom, &failed, &bad, tot);  // synchronises: results are needed now
            c.t_gpu += now_s() - tg;
            if (err) break;
            if (failed >= 0) {  // not four-line FASTQ from here on: the general parser continues at this chunk
                fallback_off = carry_file_off;
            } else if (dev_out) {
                // a .gz output: the records are gathered and compressed on the device (the chunk's text is there already), the block
                // goes straight back to its reader
                const int set = (int)(n_formatted & 1);
                const uint8_t *d_a = nullptr, *d_r = nullptr;
                int64_t nb_a = 0, nr_a = 0, nb_r = 0, nr_r = 0;
                if (c.acc_out.active()) err = gs_filter_compact_text(c.bloom, 1, c.with_probs ? 1 : 0, set, &d_a, &nb_a, &nr_a);
                if (!err && c.rest_out.active()) err = gs_filter_compact_text(c.bloom, 0, c.with_probs ? 1 : 0, set, &d_r, &nb_r, &nr_r);
                if (err) break;
                c.accepted += c.acc_out.active() ? nr_a : n_reads - nr_r;
                carry_file_off = i * (int64_t)tr.block + cut + 1;
                carry.assign(blk + cut + 1, blk + sl.n);
                carry_lines = rem;
                if (dev_job.valid() && (err = dev_job.get())) break;  // one chunk at a time: output order
                n_formatted++;
                auto job = [&c, set, d_a, nb_a, d_r, nb_r]() -> int {
                    const int e1 = c.acc_dev.emit(set, d_a, nb_a);
                    const int e2 = c.rest_dev.emit(set, d_r, nb_r);
                    return e1 ? e1 : e2;
                };
                try {
                    dev_job = std::async(std::launch::async, job);
                } catch (const std::system_error &) {  // no thread to be had: on this one
                    if ((err = job())) break;
                }
            } else {
                // (the carry is taken out first: the block returns to its reader when the writers are through with it)
                carry_file_off = i * (int64_t)tr.block + cut + 1;
                carry.assign(blk + cut + 1, blk + sl.n);
                carry_lines = rem;
                if (formatting.valid()) formatting.get();  // one chunk at a time: output order, the other set is free
                n_formatted++;
                keep_block = true;
                auto write_chunk = [&c, &tr, h_acc, h_nl, start, n_reads, i] {
                    format_text_chunk(c, start, h_acc, h_nl, n_reads);
                    tr.release(i);  // the block goes back to its reader
                };
                try {
                    formatting = std::async(std::launch::async, write_chunk);
                } catch (const std::system_error &) {  // no thread to be had: on this one
                    write_chunk();
                }
            }
        }
        if (!keep_block) tr.release(i);
        if (eof || fallback_off >= 0) break;
    }
    if (formatting.valid()) formatting.get();
    if (dev_job.valid()) {
        const int e2 = dev_job.get();
        if (!err) err = e2;
    }
    tr.close();
    c.t_parse += now_s() - t0;
    if (err) return err;
    c.reads += tot[0];
    c.kmers += tot[1];
    c.bps += tot[2];
    if (fallback_off >= 0) {
        err = gs_filter_text_reset(c.bloom, 1);
        if (err) return err;
        // not four lines per record from the very first chunk: once more with the records found on the device (GS_HOST_ML=0:
        // straight to the reference-exact parser, which also takes over whatever that pass refuses)
        bool ml = fallback_off == 0;
        if (const char *e = getenv("GS_HOST_ML")) ml = ml && atoi(e) != 0;
        if (ml) return filter_general_file(c, path, gzip, false);
        return filter_parsed_source(c, path, fallback_off, nullptr, 0);
    }
    if (!carry.empty()) return filter_parsed_source(c, std::string(), 0, carry.data(), carry.size());
    return GS_OK;
}

// FASTA and general FASTQ (sequence / quality over several lines): chunks of whole records (FASTA: cut in front of a header
// line) or of whole lines (general FASTQ: the device says how many records end in the chunk and what they cover) go to the
// device (gs_filter_submit_fasta / gs_filter_submit_fastq_ml); every record is written as four-line FASTQ.  What the device
// refuses and the tail of the file go through the reference-exact parser.
int filter_general_file(FilterCtx &c, const std::string &path, bool gzip, bool fasta) {
    size_t block;
    int readers;
    filter_reader_shape(gzip, &block, &readers);
    TextReader tr;
    int err = tr.open(path, block, readers, gzip);
    if (err) {
        tr.close();
        return err;
    }
    err = gs_filter_text_reset(c.bloom, 1);
    // two result sets: the records of chunk i are formatted and handed to the writers on a thread of their own while chunk i + 1 is
    // on the device (as the four-line path does; the chunk's block goes back to its reader when the formatting is through)
    struct Res {
        PinnedVec<uint8_t> acc;
        PinnedVec<uint32_t> nls;
        std::vector<uint64_t> bounds;
        std::vector<uint8_t> cls;
        std::vector<int64_t> head;
    } res[2];
    std::future<void> formatting;
    int64_t n_chunks = 0;
    std::vector<uint8_t> carry;
    int64_t carry_lines = 0, carry_headers = 0, carry_file_off = 0, fallback_off = -1;
    int64_t tot[3] = {0, 0, 0}, failed = -1, bad = -1;
    const double t0 = now_s();
    if (!err) tr.start();
    for (int64_t i = 0; !err; i++) {
        TextSlot &sl = tr.wait_full(i);
        if (sl.io_error || !tr.verify_gzip(sl)) {
            err = hfail(tr.gz ? GS_E_INVALID : GS_E_IO, (tr.gz ? "corrupt gzip stream in " : "read error on ") + path);
            break;
        }
        uint8_t *blk = sl.buf + tr.headroom;
        const int64_t n = (int64_t)sl.n;
        const bool eof = sl.eof;
        // the chunk: [start, start + bytes) = the carry + the block up to `cut`; `rest` = what stays for the next block
        FastaCut fc;
        if (fasta)
            fc = fasta_cut(blk, n, eof, carry);
        else if (sl.newlines > 0)
            fc.cut = (int64_t)sl.last4[0] + 1;  // behind the block's last newline
        if (fc.cut < 0) {  // no boundary in this block: keep everything
            carry.insert(carry.end(), blk, blk + n);
            carry_lines += sl.newlines;
            carry_headers += fc.headers;
            if (carry.size() > tr.headroom && !eof) fallback_off = carry_file_off;  // a record longer than a block
        } else if (carry.size() > tr.headroom) {
            fallback_off = carry_file_off;
        } else {
            uint8_t *start = blk - carry.size();
            if (!carry.empty()) memcpy(start, carry.data(), carry.size());
            Res &rs = res[n_chunks & 1];
            PinnedVec<uint8_t> &acc = rs.acc;
            PinnedVec<uint32_t> &nls = rs.nls;
            std::vector<uint64_t> &bounds = rs.bounds;
            std::vector<uint8_t> &cls = rs.cls;
            std::vector<int64_t> &head = rs.head;
            std::function<void()> format_job;
            const int64_t bytes = (int64_t)carry.size() + fc.cut;
            int64_t lines = carry_lines + sl.newlines - fc.tail_lines, records = carry_headers + fc.cut_headers, used = bytes, ticket = -1;
            if (fasta && records >= ((int64_t)1 << 24)) {  // (more records than one chunk may hold)
                fallback_off = carry_file_off;
            } else if (bytes > 0) {
                if ((err = acc.resize((size_t)(fasta ? std::max<int64_t>(records, 1) : lines / 4 + 2)))) break;
                if ((err = nls.resize((size_t)std::max<int64_t>(lines, 1)))) break;
                const double tg = now_s();
                if (fasta) {
                    err = gs_filter_submit_fasta(c.bloom, c.k, c.min_pos_count, c.positive_ratio, start, bytes, lines, records, GS_MEM_HOST,
                                                 acc.data(), nls.data(), &ticket);
                } else {
                    int64_t all_lines = lines;
                    err = gs_filter_submit_fastq_ml(c.bloom, c.k, c.min_pos_count, c.positive_ratio, start, bytes, all_lines, GS_MEM_HOST,
                                                    acc.data(), nls.data(), &records, &used, &lines, &ticket);
                    carry_lines = all_lines - lines;  // (lines the records did not cover)
                }
                if (!err) err = gs_filter_text_status(c.bloom, &failed, &bad, tot);  // synchronises: results are needed now
                if (!err && failed < 0 && records > 0) {
                    bounds.resize((size_t)records + 1);
                    cls.resize((size_t)lines);
                    err = gs_filter_text_read_bounds(c.bloom, bounds.data());
                    if (!err && !fasta) err = gs_filter_text_line_classes(c.bloom, cls.data());
                }
                c.t_gpu += now_s() - tg;
                if (err) break;
                if (failed >= 0 || records < 0) {  // refused: the general parser continues at this chunk
                    fallback_off = carry_file_off;
                } else if (records > 0) {
                    g_filter_general_chunks.fetch_add(1);
                    const uint32_t *nl = nls.p;
                    auto line_start = [nl](int64_t j) { return j ? (size_t)nl[j - 1] + 1 : (size_t)0; };
                    if (fasta)
                        for (int64_t j = 0; j < lines; j++) cls[(size_t)j] = start[line_start(j)] == '>' && nl[j] > line_start(j) ? 1 : 2;
                    head.clear();  // descriptor line of every record, + lines
                    for (int64_t j = 0; j < lines; j++)
                        if (cls[(size_t)j] == 1) head.push_back(j);
                    if ((int64_t)head.size() != records) {
                        err = hfail(GS_E_INVALID, "text chunk: the descriptor lines do not match the device's record count");
                        break;
                    }
                    head.push_back(lines);
                    const bool probs = c.with_probs && !fasta;
                    const int64_t n_rec = records;
                    Res *rp = &rs;
                    format_job = [&c, rp, start, nl, n_rec, fasta, probs] {
                        std::vector<FilterPart> parts((size_t)c.pool.threads());
                        const Res &r_ = *rp;
                        c.pool.run(n_rec, [&](int t, int64_t lo, int64_t hi) {
                            FilterPart &p = parts[(size_t)t];
                            p.acc = c.acc_out.take();
                            p.rest = c.rest_out.take();
                            for (int64_t r = lo; r < hi; r++) {  // nextEntry (FastqBloomFilter.java:92-105), input order
                                const int64_t L = (int64_t)(r_.bounds[(size_t)r + 1] - r_.bounds[(size_t)r]);
                                if (r_.acc[(size_t)r]) {
                                    p.n_accepted++;
                                    if (c.acc_out.active())
                                        append_general_record(p.acc, start, nl, r_.cls.data(), r_.head[(size_t)r], r_.head[(size_t)r + 1], L, fasta, probs);
                                } else if (c.rest_out.active())
                                    append_general_record(p.rest, start, nl, r_.cls.data(), r_.head[(size_t)r], r_.head[(size_t)r + 1], L, fasta, probs);
                            }
                            p.pack(c.acc_out, c.rest_out);
                        });
                        write_filter_parts(c, parts);
                    };
                }
            }
            if (fallback_off < 0) {
                carry_file_off += used;
                // what the records did not cover + what lies behind the cut
                std::vector<uint8_t> rest(start + used, start + bytes);
                rest.insert(rest.end(), blk + fc.cut, blk + n);
                carry.swap(rest);
                if (fasta) {
                    carry_lines = fc.tail_lines;
                    carry_headers = fc.headers - fc.cut_headers;
                }
            }
            if (format_job) {  // (the carry has been taken out of the block: the formatting thread may hand it back)
                if (formatting.valid()) formatting.get();  // one chunk at a time: output order, and the other result set is free again
                n_chunks++;
                try {
                    formatting = std::async(std::launch::async, [format_job, &tr, i] {
                        format_job();
                        tr.release(i);
                    });
                } catch (const std::system_error &) {  // no thread to be had: on this one
                    format_job();
                    tr.release(i);
                }
                if (eof || fallback_off >= 0) break;
                continue;
            }
        }
        tr.release(i);
        if (eof || fallback_off >= 0) break;
    }
    if (formatting.valid()) formatting.get();
    tr.close();
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
    if (!carry.empty()) return filter_parsed_source(c, std::string(), 0, carry.data(), carry.size(), fasta);
    return GS_OK;
}

}  // namespace

extern "C" int gs_host_filter_files(gs_bloom *bloom, int k, int min_pos_count, double positive_ratio,
                                    const char *const *paths, int n_paths, const char *filtered_path,
                                    const char *rest_path, int with_probs, gs_host_totals *totals) try {
    if (!bloom || !paths || n_paths < 0) return hfail(GS_E_INVALID, "NULL argument");
    FilterCtx c;
    c.with_probs = with_probs != 0;
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
        if (kind >= 3)
            err = filter_general_file(c, path, kind == 4, true);
        else if (kind) {
            bool handled = false;
            if (kind == 2) err = filter_bgzf_file(c, path, &handled);  // (block-gzip: inflated on the device)
            if (!err && !handled) err = filter_text_file(c, path, kind == 2);
        }
        else
            err = filter_parsed_source(c, path, 0, nullptr, 0);
    }
    const double tc0 = now_s();
    const bool wrote = c.acc_out.close() & c.rest_out.close();
    if (!err) err = c.acc_dev.late_err ? c.acc_dev.late_err : c.rest_dev.late_err;
    if (getenv("GS_HOST_TRACE") != nullptr) fprintf(stderr, "filter files: %.2f ms before the outputs were closed, closing %.2f ms\n", (tc0 - t_start) * 1e3, (now_s() - tc0) * 1e3);
    if (!err && !wrote) err = hfail(GS_E_IO, "write to an output file failed");
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
} catch (const std::bad_alloc &) {
    return hfail(GS_E_NOMEM, "out of host memory");
} catch (const std::exception &e) {  // (nothing may leave through the C ABI)
    return hfail(GS_E_INVALID, std::string("unexpected exception: ") + e.what());
}

// The pools above keep page-locked blocks and the device decoders' buffers (a parked gunzipper holds up to a quarter of the free
// HBM) for the life of the process.  A long-lived host -- a JVM that next loads a big store -- hands them back with this call;
// nothing may be using the host layer on another thread meanwhile.
namespace {
void release_pools_impl() {
    {
        PinnedPool &pp = pinned_pool();  // (the big buffers of this file)
        std::lock_guard<std::mutex> l(pp.m);
        for (auto &x : pp.idle) gs_pinned_free(x.first);
        pp.idle.clear();
    }
    gs_host::pinned_pool().release_all();  // (the readers' blocks)
    {
        InflaterPool &ip = inflater_pool();
        std::lock_guard<std::mutex> l(ip.m);
        for (auto &x : ip.idle) gs_inflater_destroy(x.second);
        ip.idle.clear();
    }
    {
        GunzipperPool &gp = gunzipper_pool();
        std::lock_guard<std::mutex> l(gp.m);
        for (auto &x : gp.idle) gs_gunzipper_close(x.second);
        gp.idle.clear();
    }
    {
        DeflaterPool &dp = deflater_pool();
        std::lock_guard<std::mutex> l(dp.m);
        for (auto &x : dp.idle) gs_deflater_destroy(x.second);
        dp.idle.clear();
    }
}
}  // namespace

extern "C" int gs_host_release_pools(void) try {
    release_pools_impl();
    gs_device_cache_trim();  // (the pools' device memory went through the C ABI library's block cache)
    return GS_OK;
} catch (const std::exception &e) {  // (nothing may leave through the C ABI)
    return hfail(GS_E_INVALID, std::string("unexpected exception: ") + e.what());
}

extern "C" int64_t gs_host_stat(int which) {
    return which == 0 ? g_ml_chunks.load() : (which == 1 ? g_filter_general_chunks.load() : -1);
}
