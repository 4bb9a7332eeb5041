// gs_pool.h -- a small pool of worker threads for short data-parallel loops of the host layer.
#ifndef GS_POOL_H
#define GS_POOL_H

#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// A few threads for short data-parallel loops of the host layer (formatting the per-read outputs of a chunk, inflating
// the BGZF blocks of a buffer): run(n, f, grain) calls f(t, lo, hi) for contiguous ranges of [0, n) of at least `grain`
// items, range t on worker t, the last one on the caller's thread, and returns when all are done.  Small inputs run
// inline.  (Threads are kept: such a loop takes well under a millisecond, less than starting them would take.)
class GsRangePool {
public:
    explicit GsRangePool(int threads) : n_(std::max(1, threads)) {}
    int threads() const { return n_; }
    template <class F>
    void run(int64_t n, F f, int64_t grain = 1024) {
        const int T = (int)std::max<int64_t>(1, std::min<int64_t>(n_, n / grain));
        if (T == 1) {
            f(0, (int64_t)0, n);
            return;
        }
        {
            std::lock_guard<std::mutex> l(m_);
            if (th_.empty())
                for (int t = 0; t + 1 < n_; t++) th_.emplace_back([this, t] { worker(t); });
            job_ = [&f, n, T](int t) { f(t, n * t / T, n * (t + 1) / T); };
            active_ = T - 1;
            pending_ = T - 1;
            gen_++;
        }
        cv_.notify_all();
        f(T - 1, n * (T - 1) / T, n);
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [&] { return pending_ == 0; });
    }
    ~GsRangePool() {
        {
            std::lock_guard<std::mutex> l(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &x : th_) x.join();
    }

private:
    void worker(int t) {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> l(m_);
            cv_.wait(l, [&] { return stop_ || gen_ != seen; });
            if (stop_) return;
            seen = gen_;
            if (t >= active_) continue;
            l.unlock();
            job_(t);
            l.lock();
            if (--pending_ == 0) done_.notify_all();
        }
    }
    const int n_;
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::function<void(int)> job_;
    uint64_t gen_ = 0;
    int active_ = 0, pending_ = 0;
    bool stop_ = false;
};

#endif
