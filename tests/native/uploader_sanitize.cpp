// Sanitizer harness for the host threads of the device gunzip (genestrip_amd/csrc/gs_upload.h: the staged copy and the upload thread
// of gs_gunzipper_*), against a MOCK device: page-locked memory is malloc, the copy queue is a worker thread that executes copies and
// event records in order, "device memory" is a heap buffer -- and what arrives there is read back by the HOST decoder of gs_inflate.h,
// which must produce the text the stream was made from.  Built by tests/test_host_cpu.py with -fsanitize=thread and with
// -fsanitize=address,undefined (GPU sanitizers are not available on this pool).  Scenarios: a stream followed batch by batch; park()
// in the middle with the source FREED right after it; the next stream on the same object (reopen); a copy helper that fails; an
// uploader destroyed while it runs; two uploaders on one device.
#include "../../genestrip_amd/csrc/gs_inflate.h"
#include "../../genestrip_amd/csrc/gs_upload.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <deque>
#include <memory>
#include <random>

static std::vector<uint8_t> gz(const std::vector<uint8_t> &in, int level) {
    z_stream z{};
    deflateInit2(&z, level, Z_DEFLATED, 31, 8, Z_DEFAULT_STRATEGY);
    std::vector<uint8_t> out(deflateBound(&z, in.size()) + 64);
    z.next_in = (Bytef *)in.data();
    z.avail_in = (uInt)in.size();
    z.next_out = out.data();
    z.avail_out = (uInt)out.size();
    deflate(&z, Z_FINISH);
    out.resize(z.total_out);
    deflateEnd(&z);
    return out;
}
static std::vector<uint8_t> fastq(size_t n, uint64_t seed) {
    std::mt19937_64 rng(seed);
    std::vector<uint8_t> v;
    const char *b = "ACGT";
    unsigned long long id = 0;
    while (v.size() < n) {
        char d[96];
        int m = snprintf(d, 96, "@A00123:45:HXX:1:1101:%llu:%llu 1:N:0:ACGT\n", 1000 + (id / 50) % 30000, 1000 + (id * 37) % 40000);
        id++;
        v.insert(v.end(), d, d + m);
        for (int i = 0; i < 150; i++) v.push_back(b[rng() & 3]);
        v.push_back('\n');
        v.push_back('+');
        v.push_back('\n');
        for (int i = 0; i < 150; i++) v.push_back(rng() % 100 < 88 ? 'F' : ':');
        v.push_back('\n');
    }
    v.resize(n);
    return v;
}

// the mock device: one in-order queue served by a worker thread
struct MockDev {
    typedef int Event;
    struct Op {
        int kind;  // 0 copy, 1 record
        uint8_t *dst;
        const uint8_t *src;
        size_t n;
        int ev;
    };
    std::mutex m;
    std::condition_variable cv;
    std::deque<Op> q;
    std::vector<long> ev_done;  // per event: how many records of it have completed
    std::vector<long> ev_asked;
    bool stop = false, busy = false;
    std::thread worker;
    std::atomic<int> copies{0};
    int fail_at_copy = -1;  // the copy with this number (from 0) fails
    int slow_us = 0;        // per copy, so that a consumer can overtake the upload
    MockDev() {
        worker = std::thread([this] {
            for (;;) {
                Op op;
                {
                    std::unique_lock<std::mutex> l(m);
                    cv.wait(l, [&] { return stop || !q.empty(); });
                    if (q.empty()) return;
                    op = q.front();
                    q.pop_front();
                    busy = true;
                }
                if (op.kind == 0) {
                    if (slow_us) std::this_thread::sleep_for(std::chrono::microseconds(slow_us));
                    memcpy(op.dst, op.src, op.n);
                }
                {
                    std::lock_guard<std::mutex> l(m);
                    if (op.kind == 1) ev_done[(size_t)op.ev]++;
                    busy = false;
                }
                cv.notify_all();
            }
        });
    }
    ~MockDev() {
        {
            std::lock_guard<std::mutex> l(m);
            stop = true;
        }
        cv.notify_all();
        worker.join();
    }
    int bind() { return 0; }
    int take_staging(uint8_t *h[2], Event ev[2], size_t piece) {
        std::lock_guard<std::mutex> l(m);
        for (int i = 0; i < 2; i++) {
            h[i] = (uint8_t *)malloc(piece);
            ev[i] = (int)ev_done.size();
            ev_done.push_back(0);
            ev_asked.push_back(0);
        }
        return 0;
    }
    void give_staging(uint8_t *h[2], Event ev[2]) {
        (void)ev;
        for (int i = 0; i < 2; i++) free(h[i]);
    }
    int copy_async(uint8_t *d, const uint8_t *s, size_t n) {
        if (copies++ == fail_at_copy) return -3;
        {
            std::lock_guard<std::mutex> l(m);
            q.push_back({0, d, s, n, 0});
        }
        cv.notify_all();
        return 0;
    }
    int record(Event e) {
        {
            std::lock_guard<std::mutex> l(m);
            ev_asked[(size_t)e]++;
            q.push_back({1, nullptr, nullptr, 0, e});
        }
        cv.notify_all();
        return 0;
    }
    int wait_event(Event e) {
        std::unique_lock<std::mutex> l(m);
        const long want = ev_asked[(size_t)e];
        cv.wait(l, [&] { return ev_done[(size_t)e] >= want; });
        return 0;
    }
    int drain() {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return q.empty() && !busy; });
        return 0;
    }
};

// what the host decoder makes of the bytes that arrived in "device memory"
static bool inflates_to(const uint8_t *dev_mem, size_t n, const std::vector<uint8_t> &text) {
    std::unique_ptr<GsInflate> inf(new GsInflate());
    inf->init(dev_mem, n, true);
    std::vector<uint8_t> out(text.size() + 1);
    size_t total = 0;
    for (;;) {
        size_t p = 0;
        const GsInflate::Status st = inf->decode(out.data() + total, std::min<size_t>(100000, out.size() - total), total, &p);
        total += p;
        if (st == GsInflate::CORRUPT) return false;
        if (st == GsInflate::DONE) break;
        if (total == out.size()) return false;
    }
    return total == text.size() && memcmp(out.data(), text.data(), total) == 0;
}

int main() {
    int fails = 0;
    const auto text = fastq(5000000, 1);
    const auto comp = gz(text, 6);
    const int64_t n = (int64_t)comp.size();
    // 1. a stream followed batch by batch, as gs_gunzipper_next follows its upload thread; pieces of 64 KiB, two and four copy threads
    for (int threads : {1, 2, 4}) {
        MockDev dev;
        dev.slow_us = 50;
        std::vector<uint8_t> dmem((size_t)n + 64, 0xee);
        GsUploader<MockDev> up;
        up.start(&dev, dmem.data(), comp.data(), n, 65536, threads);
        for (int64_t need = 150000; ; need += 150000) {
            int64_t have = 0;
            const int64_t want = std::min(need, n);
            if (up.wait(want, &have) != 0 || have < want) fails++;
            if (memcmp(dmem.data(), comp.data(), (size_t)have) != 0) fails++;  // what is said to have arrived has arrived
            if (want == n) break;
        }
        up.park();
        if (!inflates_to(dmem.data(), (size_t)n, text)) fails++;
    }
    // 2. park in the middle, the source freed at once; then the next stream on the same uploader (gs_gunzipper_reopen)
    {
        MockDev dev;
        dev.slow_us = 200;
        std::vector<uint8_t> dmem((size_t)n + 64);
        GsUploader<MockDev> up;
        auto *src = new std::vector<uint8_t>(comp);
        up.start(&dev, dmem.data(), src->data(), n, 65536, 2);
        int64_t have = 0;
        if (up.wait(200000, &have) != 0) fails++;
        up.park();
        delete src;  // (ASan: a thread that still read the source would show here)
        if (up.wait(n, &have) == 0 && have < n) fails++;  // a parked upload never reports bytes it has not copied
        const auto text2 = fastq(1500000, 7);
        const auto comp2 = gz(text2, 1);
        std::vector<uint8_t> dmem2(comp2.size() + 64);
        up.start(&dev, dmem2.data(), comp2.data(), (int64_t)comp2.size(), 65536, 1);
        if (up.wait((int64_t)comp2.size(), &have) != 0) fails++;
        up.park();
        if (!inflates_to(dmem2.data(), comp2.size(), text2)) fails++;
    }
    // 3. a copy helper that fails: the waiting batch gets the error, nothing hangs, what arrived before stays valid
    {
        MockDev dev;
        dev.fail_at_copy = 5;
        std::vector<uint8_t> dmem((size_t)n + 64);
        GsUploader<MockDev> up;
        up.start(&dev, dmem.data(), comp.data(), n, 65536, 1);
        int64_t have = 0;
        if (up.wait(n, &have) == 0) fails++;
        if (have > 5 * 65536) fails++;
        if (up.wait(std::min<int64_t>(have, 65536), &have) != 0 && have >= 65536) fails++;
        if (memcmp(dmem.data(), comp.data(), (size_t)have) != 0) fails++;
        up.park();
    }
    // 4. an uploader destroyed while its copy runs
    {
        MockDev dev;
        dev.slow_us = 500;
        std::vector<uint8_t> dmem((size_t)n + 64);
        {
            GsUploader<MockDev> up;
            up.start(&dev, dmem.data(), comp.data(), n, 65536, 2);
            int64_t have = 0;
            up.wait(70000, &have);
        }
        dev.drain();
    }
    // 5. two uploaders on one device (a filter job and a match job), each with a staging set of its own
    {
        MockDev dev;
        std::vector<uint8_t> d1((size_t)n + 64), d2((size_t)n + 64);
        GsUploader<MockDev> a, b;
        a.start(&dev, d1.data(), comp.data(), n, 65536, 2);
        b.start(&dev, d2.data(), comp.data(), n, 131072, 1);
        int64_t have = 0;
        if (a.wait(n, &have) != 0 || b.wait(n, &have) != 0) fails++;
        a.park();
        b.park();
        if (!inflates_to(d1.data(), (size_t)n, text) || !inflates_to(d2.data(), (size_t)n, text)) fails++;
    }
    // 6. the staged copy on the calling thread (a batch that uploads its own span), cancelled by its callback
    {
        MockDev dev;
        std::vector<uint8_t> dmem((size_t)n + 64);
        int pieces = 0;
        const int rc = gs_staged_copy<MockDev>(dev, dmem.data(), comp.data(), (size_t)n, 65536, 3, [&](size_t, int) { return ++pieces == 4 ? -5 : 0; });
        if (rc != -5 || pieces != 4) fails++;
        if (memcmp(dmem.data(), comp.data(), 4 * 65536) != 0) fails++;  // (drained before it returned)
        if (gs_staged_copy<MockDev>(dev, dmem.data(), comp.data(), (size_t)n, 100000, 0, nullptr) != 0 || !inflates_to(dmem.data(), (size_t)n, text)) fails++;
    }
    printf("fails %d\n", fails);
    return fails;
}
