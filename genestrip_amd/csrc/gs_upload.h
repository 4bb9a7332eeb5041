// gs_upload.h -- the host threads of the device gunzip (gs_inflate_dev.hip: gs_gunzipper_*), separated from the HIP calls they make so
// that the same code runs against a mock device under ThreadSanitizer / AddressSanitizer (tests/native/uploader_sanitize.cpp; GPU
// sanitizers are not available on this pool).
//
//   gs_staged_copy<Dev>   host memory of any kind (a page-cache mapping is copied by the runtime at a crawl) to the device through
//                         two page-locked pieces, filled by several threads while the other one is on its way; after every piece
//                         a callback may queue work behind it or cancel the copy
//   GsUploader<Dev>       a thread that runs one such copy of a whole compressed stream while the caller decodes batches of it:
//                         wait(need) blocks until the first `need` bytes have arrived (or the copy has failed), park() stops the
//                         thread -- the source may then go away --, start() begins the next stream on the same object
//
// Dev (a device as these threads see it):
//   typedef Event;  int bind();                       make the device current on the calling thread
//   int take_staging(uint8_t *h[2], Event ev[2], size_t piece);   two page-locked pieces + events, for the length of one copy
//   void give_staging(uint8_t *h[2], Event ev[2]);
//   int copy_async(uint8_t *d_dst, const uint8_t *h_src, size_t n);   queued in order behind the earlier copies
//   int record(Event ev);  int wait_event(Event ev);  int drain();    drain: everything queued so far has completed
// Every int is 0 or a negative GS_E_* code.
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#ifndef GS_UPLOAD_E_STATE
#define GS_UPLOAD_E_STATE (-5)  // GS_E_STATE: the copy was cancelled
#define GS_UPLOAD_E_NOMEM (-2)
#define GS_UPLOAD_E_HIP (-3)
#endif

// after_piece(bytes queued so far, the event behind that piece's copy) -> 0, or a code that ends the copy
template <class Dev>
int gs_staged_copy(Dev &dev, uint8_t *d_dst, const uint8_t *src, size_t n, size_t piece, int copy_threads,
                   const std::function<int(size_t, typename Dev::Event)> &after_piece) {
    uint8_t *h[2] = {nullptr, nullptr};
    typename Dev::Event done[2];
    int rc = dev.take_staging(h, done, piece);
    if (rc) return rc;
    struct Return {  // (back to the device on every way out; every way out has drained the queue first)
        Dev &dev;
        uint8_t **h;
        typename Dev::Event *ev;
        ~Return() { dev.give_staging(h, ev); }
    } give_back{dev, h, done};
    int which = 0;
    bool used[2] = {false, false};
    for (size_t at = 0; at < n; at += piece, which ^= 1) {
        const size_t len = std::min(piece, n - at);
        if (used[which] && (rc = dev.wait_event(done[which]))) {
            dev.drain();
            return rc;
        }
        int n_thr = copy_threads > 0 ? copy_threads : (int)std::min<size_t>(8, std::max<size_t>(1, len >> 22));
        {
            std::vector<std::thread> th;
            struct Join {  // (a thread that could not be started must not leave the others unjoined)
                std::vector<std::thread> &th;
                ~Join() {
                    for (auto &x : th)
                        if (x.joinable()) x.join();
                }
            } join{th};
            size_t started = 1;
            for (int t = 1; t < n_thr; t++) {
                const size_t a = len * (size_t)t / (size_t)n_thr, b = len * ((size_t)t + 1) / (size_t)n_thr;
                uint8_t *dst = h[which];
                try {
                    th.emplace_back([=] { memcpy(dst + a, src + at + a, b - a); });
                    started++;
                } catch (const std::system_error &) {  // no more threads: this one copies the rest
                    memcpy(dst + a, src + at + a, len - a);
                    break;
                }
            }
            (void)started;
            memcpy(h[which], src + at, len / (size_t)n_thr);
        }
        rc = dev.copy_async(d_dst + at, h[which], len);
        if (!rc) rc = dev.record(done[which]);
        if (rc) {
            dev.drain();
            return rc;
        }
        used[which] = true;
        if (after_piece) {
            rc = after_piece(at + len, done[which]);
            if (rc) {  // (cancelled or failed: the copies under way still read h[], which the next copy fills)
                dev.drain();
                return rc;
            }
        }
    }
    return dev.drain();
}

template <class Dev>
class GsUploader {
public:
    GsUploader() = default;
    GsUploader(const GsUploader &) = delete;
    GsUploader &operator=(const GsUploader &) = delete;
    ~GsUploader() { park(); }

    // src[0, n) -> d_dst on a thread of this object's own; throws std::system_error when no thread can be had (nothing is started)
    void start(Dev *dev, uint8_t *d_dst, const uint8_t *src, int64_t n, size_t piece = (size_t)32 << 20, int copy_threads = 0) {
        park();
        done_ = 0;
        rc_ = 0;
        cancel_ = finished_ = false;
        thr_ = std::thread([this, dev, d_dst, src, n, piece, copy_threads] {
            int rc = dev->bind();
            typename Dev::Event prev{};
            bool have_prev = false;
            size_t prev_end = 0;
            try {
                if (!rc)
                    rc = gs_staged_copy<Dev>(*dev, d_dst, src, (size_t)n, piece, copy_threads, [&](size_t up, typename Dev::Event ev) -> int {
                        if (have_prev) {  // the piece before this one has arrived when its event has (this one is on its way)
                            const int wrc = dev->wait_event(prev);
                            if (wrc) return wrc;
                            publish((int64_t)prev_end, 0, false);
                        }
                        prev = ev;
                        have_prev = true;
                        prev_end = up;
                        std::lock_guard<std::mutex> l(m_);
                        return cancel_ ? GS_UPLOAD_E_STATE : 0;
                    });
            } catch (...) {  // (the copy helpers' threads, memory: the batches then fail with this code)
                rc = GS_UPLOAD_E_NOMEM;
            }
            publish(rc ? 0 : n, rc, true);
        });
    }

    // until `need` bytes of the stream have arrived; *have: how many have.  Non-zero: the copy failed (or was cancelled) before that.
    int wait(int64_t need, int64_t *have) {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return done_ >= need || finished_; });
        if (have) *have = done_;
        if (done_ < need) return rc_ ? rc_ : GS_UPLOAD_E_HIP;
        return 0;
    }

    // the caller is through with the source (or gives up on it): the thread is stopped and joined; the source may go away
    void park() {
        if (thr_.joinable()) {
            {
                std::lock_guard<std::mutex> l(m_);
                cancel_ = true;
            }
            thr_.join();
        }
        cancel_ = false;
    }

    bool running() const { return thr_.joinable(); }

private:
    void publish(int64_t done, int rc_now, bool fin) {
        {
            std::lock_guard<std::mutex> l(m_);
            done_ = std::max(done_, done);
            if (rc_now) rc_ = rc_now;
            finished_ = finished_ || fin;
        }
        cv_.notify_all();
    }
    std::thread thr_;
    std::mutex m_;
    std::condition_variable cv_;
    int64_t done_ = 0;  // bytes of the stream that have arrived (under m_)
    int rc_ = 0;
    bool cancel_ = false, finished_ = false;
};
