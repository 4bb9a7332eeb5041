// gs_api.cpp -- C ABI (include/gsgpu.h) over the gfx950 kernels: host-side table build, HBM residency,
// stream/event plumbing.  No CPU fallback exists: every entry point needs a HIP device and fails with
// GS_E_NODEVICE / GS_E_HIP otherwise.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "gs_layout.h"
#include "gs_params.h"

typedef unsigned long long u64;

// ---------------------------------------------------------------------------------------------------
// Device blocks between calls.  A file-level call (gs_host_match_files: begin, a few submits, finish, destroy) allocates some twenty
// buffers -- text banks, queues, result arrays -- and frees them 30 ms later; hipMalloc and hipFree of those cost 5.5 of the 30 ms
// (GS_HOST_TRACE: the first submit 2.9 ms against 0.09 for the next ones, destroy 2.6).  Every hipMalloc / hipFree of THIS file goes
// through a small cache instead: a freed block waits (per device, by exact size) for the next request of its size.  hipFree's
// contract is kept -- the device is idle when it returns --, only the unmapping is saved.  At most GS_DEVICE_CACHE_MB (default 4096)
// wait, the oldest go first; a block of more than a quarter of that (a big store) is never kept; when an allocation fails everything
// that waits is freed and the allocation tried again; gs_device_cache_trim() frees it all (gs_host_release_pools calls it).
// ---------------------------------------------------------------------------------------------------
#include <deque>
#include <mutex>
#include <unordered_map>
namespace gs_cache {
struct Block {
    void *p;
    size_t n;
    int dev;
};
struct State {
    std::mutex mu;
    std::unordered_map<void *, std::pair<size_t, int>> live;  // blocks handed out: size, device
    std::deque<Block> idle;                                    // blocks that wait, oldest first
    size_t idle_bytes = 0;
};
static State &state() {
    static State *s = new State();  // (never destroyed: blocks may be freed from static destructors)
    return *s;
}
static size_t cap_bytes() {
    static const size_t v = [] {
        const char *e = getenv("GS_DEVICE_CACHE_MB");
        return (size_t)(e ? std::max(0, atoi(e)) : 4096) << 20;
    }();
    return v;
}
static void drop_idle_locked(State &st, size_t keep_bytes) {
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    while (!st.idle.empty() && st.idle_bytes > keep_bytes) {
        const Block b = st.idle.front();
        st.idle.pop_front();
        st.idle_bytes -= b.n;
        hipSetDevice(b.dev);
        hipFree(b.p);
    }
    if (have) hipSetDevice(cur);
}
static hipError_t alloc(void **p, size_t n) {
    State &st = state();
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lk(st.mu);
        for (auto it = st.idle.begin(); it != st.idle.end(); ++it)
            if (it->n == n && it->dev == dev) {
                *p = it->p;
                st.idle_bytes -= n;
                st.idle.erase(it);
                st.live[*p] = {n, dev};
                return hipSuccess;
            }
    }
    static const bool trace = getenv("GS_CACHE_TRACE") != nullptr;  // (developer: which requests the cache could not serve)
    if (trace) fprintf(stderr, "device cache: %zu bytes allocated\n", n);
    e = hipMalloc(p, n);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> lk(st.mu);
            drop_idle_locked(st, 0);
        }
        e = hipMalloc(p, n);
    }
    if (e == hipSuccess && n != 0 && n <= cap_bytes() / 4) {
        std::lock_guard<std::mutex> lk(st.mu);
        st.live[*p] = {n, dev};
    }
    return e;
}
static hipError_t release(void *p) {
    if (!p) return hipSuccess;
    State &st = state();
    std::unique_lock<std::mutex> lk(st.mu);
    const auto it = st.live.find(p);
    if (it == st.live.end()) {
        lk.unlock();
        return hipFree(p);
    }
    const Block b{p, it->second.first, it->second.second};
    st.live.erase(it);
    lk.unlock();
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    if (have && cur != b.dev) hipSetDevice(b.dev);
    const hipError_t e = hipDeviceSynchronize();  // (what hipFree guarantees: nothing on the device uses the block any more)
    if (have && cur != b.dev) hipSetDevice(cur);
    lk.lock();
    st.idle.push_back(b);
    st.idle_bytes += b.n;
    drop_idle_locked(st, cap_bytes());
    return e;
}
static void trim() {
    State &st = state();
    std::lock_guard<std::mutex> lk(st.mu);
    drop_idle_locked(st, 0);
}
}  // namespace gs_cache
template <typename T>
static inline hipError_t gs_cached_malloc(T **p, size_t n) {
    return gs_cache::alloc(reinterpret_cast<void **>(p), n);
}
static inline hipError_t gs_cached_free(void *p) { return gs_cache::release(p); }
#define hipMalloc gs_cached_malloc
#define hipFree gs_cached_free

extern "C" int gs_device_cache_trim(void) {
    gs_cache::trim();
    return GS_OK;
}

extern "C" hipError_t gs_launch_match(const GsMatchParams *P, int grid, hipStream_t stream);
extern "C" hipError_t gs_launch_match_huge(const GsMatchParams *P, int grid, hipStream_t stream);
extern "C" hipError_t gs_launch_match_wide(const GsMatchParams *P, int ns, int n_cu, hipStream_t stream);
extern "C" int gs_match_wide_mask(const GsMatchParams *P);
extern "C" hipError_t gs_launch_classify(const GsMatchParams *P, hipStream_t stream);
extern "C" hipError_t gs_launch_fold_stats(long long *sums, unsigned long long *maxk, double *dsums, long long nv, int copies, hipStream_t stream);
extern "C" hipError_t gs_launch_match_long(const GsMatchParams *P, int grid, int32_t *scratch, uint32_t *serial,
                                           hipStream_t stream);
extern "C" hipError_t gs_launch_unique_count(const u64 *table, const uint32_t *bitmap, int64_t n_slots, uint32_t vbits,
                                              int32_t n_values, u64 *unique, const u64 *rec, int64_t n_rec, hipStream_t stream);
extern "C" hipError_t gs_launch_rec_unique_count(const u64 *rec, const uint32_t *bitmap_rec, int64_t n_rec, int32_t n_values,
                                                  u64 *unique, hipStream_t stream);
extern "C" hipError_t gs_launch_clear_seen(u64 *table, int64_t n_slots, u64 *rec, int64_t n_rec, hipStream_t stream);
extern "C" hipError_t gs_launch_bitmap_extract(const u64 *table, int64_t n_slots, uint32_t *bitmap, const u64 *rec, int64_t n_rec,
                                                hipStream_t stream);
extern "C" hipError_t gs_launch_bitmap_or(uint32_t *dst, const uint32_t *parts, int64_t n_words, int64_t n_parts,
                                           hipStream_t stream);
extern "C" hipError_t gs_launch_segments(const struct GsSegParams *P, int write, int grid, hipStream_t stream);
extern "C" hipError_t gs_launch_encode(const struct GsEncodeParams *P, int grid, hipStream_t stream);
extern "C" hipError_t gs_launch_probe_keys(const GsDbDev *db, const u64 *keys, int64_t n, int32_t *nodes, int count_unique,
                                            hipStream_t stream);
extern "C" hipError_t gs_launch_filter(const struct GsFilterParams *P, int grid, hipStream_t stream);
extern "C" hipError_t gs_launch_stat_reduce(const GsStatRec *recs, const void *count, int64_t n_max, int n_values, void *sums, void *maxk,
                                             void *dsums, int32_t *vi_scratch, hipStream_t stream);
extern "C" int gs_match_occupancy(int n_values);
extern "C" int gs_match_long_occupancy(int n_values);
extern "C" int gs_filter_occupancy();

// ---------------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(e_ == hipErrorOutOfMemory ? GS_E_NOMEM : GS_E_HIP,                              \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                             \
    } while (0)

extern "C" const char *gs_last_error(void) { return g_err.c_str(); }

extern "C" const char *gs_strerror(int code) {
    switch (code) {
    case GS_OK: return "ok";
    case GS_E_INVALID: return "invalid argument";
    case GS_E_NOMEM: return "out of memory";
    case GS_E_HIP: return "HIP runtime error";
    case GS_E_UNSUPPORTED: return "unsupported";
    case GS_E_STATE: return "invalid call order";
    case GS_E_NODEVICE: return "no usable gfx950 device";
    case GS_E_IO: return "file input/output failed";
    default: return "unknown error";
    }
}

extern "C" int gs_abi_version(void) { return GS_ABI_VERSION; }

extern "C" int gs_device_count(int *n) {
    if (!n) return fail(GS_E_INVALID, "n is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess || c <= 0) {
        *n = 0;
        return fail(GS_E_NODEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *n = c;
    return GS_OK;
}

static int use_device(int device) {
    int c = 0;
    int rc = gs_device_count(&c);
    if (rc) return rc;
    if (device < 0 || device >= c) return fail(GS_E_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------------
// store build
// ---------------------------------------------------------------------------------------------------
struct gs_db {
    int device = 0;
    gs_db_info info{};
    GsDbDev dev{};
    u64 *d_table = nullptr;
    u64 *d_gate = nullptr;
    uint32_t *d_mgate = nullptr;
    u64 *d_rec = nullptr;       // super-k-mer records (gs_layout.h), GS_REC_WORDS words per bucket
    int64_t n_rec = 0;          // record buckets (a power of two) or 0
    int32_t *d_tree = nullptr;  // parent | depth | tin | tout
    int64_t n_slots() const { return info.n_buckets * GS_SLOTS_PER_BUCKET; }
    int n_cu = 256;
    struct gs_run *unique_owner = nullptr;  // the slots' seen bits belong to one unique-counting run at a time
    // Runs keep a pointer to their store.  A host with garbage-collected wrappers (Java finalizers, Python __del__) may
    // destroy the store before its runs: gs_db_destroy then only marks it, and the last gs_match_destroy frees it.
    int live_runs = 0;
    bool destroy_pending = false;
    // striped store (gs_layout.h, GsDbDev::rec_biased): d_rec is THIS handle's stripe, buckets [rec_first, rec_first +
    // rec_local) of the n_rec buckets; stripe_base[q] = where this process sees stripe q
    int n_parts = 0, part = 0;
    int64_t rec_first = 0, rec_local = 0;
    int64_t tab_first = 0, tab_local = 0;      // its buckets of the overflow table: they follow the record lines in d_rec
    const u64 *stripe_base[GS_MAX_STRIPES] = {};
    unsigned present = 0;                      // bit q: stripe q is known
    std::shared_ptr<struct StripeGroup> group; // all stripes in one process: they are freed with the last handle
    std::vector<void *> ipc_opened;            // stripes of other processes (hipIpcOpenMemHandle)
    bool striped() const { return n_parts > 1; }
    bool complete() const { return n_parts <= 1 || present == (1u << n_parts) - 1u; }
};

struct StripeGroup {
    std::vector<std::pair<int, void *>> allocs;
    ~StripeGroup() {
        for (auto &a : allocs) {
            hipSetDevice(a.first);
            hipFree(a.second);
        }
    }
};

static void db_set_stripe(gs_db *db, int q, const u64 *base) {
    db->stripe_base[q] = base;
    db->present |= 1u << q;
    // (biased: the kernels add bucket * GS_REC_WORDS with the GLOBAL bucket number)
    const size_t first = (size_t)gs_stripe_first(db->dev.rec_bits, (uint32_t)db->n_parts, (uint32_t)q);
    const size_t local = (size_t)gs_stripe_first(db->dev.rec_bits, (uint32_t)db->n_parts, (uint32_t)q + 1) - first;
    db->dev.rec_biased[q] = base - first * GS_REC_WORDS;
    db->dev.tab_biased[q] = base + local * GS_REC_WORDS -
                            (size_t)gs_tab_stripe_first(db->dev.bucket_bits, (uint32_t)db->n_parts, (uint32_t)q) * GS_SLOTS_PER_BUCKET;
}
// the table buckets of stripe q as this process sees them
static const u64 *stripe_table(const gs_db *db, int q, int64_t *first, int64_t *local) {
    const int64_t rf = (int64_t)gs_stripe_first(db->dev.rec_bits, (uint32_t)db->n_parts, (uint32_t)q);
    const int64_t rl = (int64_t)gs_stripe_first(db->dev.rec_bits, (uint32_t)db->n_parts, (uint32_t)q + 1) - rf;
    *first = (int64_t)gs_tab_stripe_first(db->dev.bucket_bits, (uint32_t)db->n_parts, (uint32_t)q);
    *local = (int64_t)gs_tab_stripe_first(db->dev.bucket_bits, (uint32_t)db->n_parts, (uint32_t)q + 1) - *first;
    return db->stripe_base[q] + (size_t)rl * GS_REC_WORDS;
}

static void db_free(gs_db *db) {
    hipSetDevice(db->device);
    for (void *p : db->ipc_opened) hipIpcCloseMemHandle(p);
    if (!db->striped()) hipFree(db->d_table);  // (a stripe's table buckets live behind its record lines)
    hipFree(db->d_gate);
    hipFree(db->d_mgate);
    if (!db->group) hipFree(db->d_rec);
    hipFree(db->d_tree);
    delete db;
}

// reference (interleaved, first base in the top bits) -> forward planes; also reports reachability:
// the reference only ever queries max(fwd, revcomp) (CGAT.java:145-147), so a stored key that is smaller
// than its reverse complement can never be hit.
static inline bool java_to_planes(u64 x, int k, uint32_t &hi, uint32_t &lo) {
    hi = lo = 0;
    u64 rc = 0;
    for (int i = 0; i < k; i++) {
        uint32_t c = (uint32_t)(x >> (2 * (k - 1 - i))) & 3u;
        hi |= (c >> 1) << i;
        lo |= (c & 1u) << i;
        rc |= (u64)(c ^ 1u) << (2 * i);  // base i complemented lands at position k-1-i from the top
    }
    return x >= rc;
}

static int bits_for(u64 v) {
    int b = 0;
    while (v) {
        b++;
        v >>= 1;
    }
    return b;
}

static int db_create_impl(gs_db **out, int device, int k, int64_t n, const int64_t *kmers, const int32_t *vidx,
                          int32_t n_values, const int32_t *parent_vi, int n_parts, int part, bool fused, int stripes = 1,
                          const int *stripe_devices = nullptr, int stripe_only = -1);
static int db_self_check(gs_db *db);

extern "C" int gs_db_create(gs_db **out, int device, int k, int64_t n, const int64_t *kmers, const int32_t *vidx,
                            int32_t n_values, const int32_t *parent_vi) try {
    return db_create_impl(out, device, k, n, kmers, vidx, n_values, parent_vi, 1, 0, true);
} catch (const std::bad_alloc &) {
    return fail(GS_E_NOMEM, "out of host memory");
} catch (const std::exception &e) {  // (nothing may leave through the C ABI)
    return fail(GS_E_INVALID, std::string("unexpected exception: ") + e.what());
}

extern "C" int gs_db_create_part(gs_db **out, int device, int k, int64_t n, const int64_t *kmers, const int32_t *vidx,
                                 int32_t n_values, const int32_t *parent_vi, int n_parts, int part) try {
    if (n_parts < 1 || part < 0 || part >= n_parts) return fail(GS_E_INVALID, "bad partition");
    return db_create_impl(out, device, k, n, kmers, vidx, n_values, parent_vi, n_parts, part, false);
} catch (const std::bad_alloc &) {
    return fail(GS_E_NOMEM, "out of host memory");
} catch (const std::exception &e) {  // (nothing may leave through the C ABI)
    return fail(GS_E_INVALID, std::string("unexpected exception: ") + e.what());
}

// ---- striped store (include/gsgpu.h)
extern "C" int gs_db_create_striped(gs_db **out, const int *devices, int n_stripes, int k, int64_t n, const int64_t *kmers,
                                    const int32_t *vidx, int32_t n_values, const int32_t *parent_vi) try {
    if (!out || !devices || n_stripes < 2 || n_stripes > GS_MAX_STRIPES) return fail(GS_E_INVALID, "a striped store spans 2..8 devices");
    for (int p = 0; p < n_stripes; p++) {
        out[p] = nullptr;
        const int rc = use_device(devices[p]);
        if (rc) return rc;
    }
    return db_create_impl(out, devices[0], k, n, kmers, vidx, n_values, parent_vi, 1, 0, true, n_stripes, devices, -1);
} catch (const std::bad_alloc &) {
    return fail(GS_E_NOMEM, "out of host memory");
} catch (const std::exception &e) {
    return fail(GS_E_INVALID, std::string("unexpected exception: ") + e.what());
}

extern "C" int gs_db_create_stripe(gs_db **out, int device, int n_stripes, int stripe, int k, int64_t n, const int64_t *kmers,
                                   const int32_t *vidx, int32_t n_values, const int32_t *parent_vi) try {
    if (!out || n_stripes < 2 || n_stripes > GS_MAX_STRIPES || stripe < 0 || stripe >= n_stripes)
        return fail(GS_E_INVALID, "a striped store spans 2..8 devices");
    return db_create_impl(out, device, k, n, kmers, vidx, n_values, parent_vi, 1, 0, true, n_stripes, nullptr, stripe);
} catch (const std::bad_alloc &) {
    return fail(GS_E_NOMEM, "out of host memory");
} catch (const std::exception &e) {
    return fail(GS_E_INVALID, std::string("unexpected exception: ") + e.what());
}

extern "C" int gs_db_stripe_export(gs_db *db, void *handle) {
    if (!db || !handle) return fail(GS_E_INVALID, "NULL argument");
    if (!db->striped() || db->group) return fail(GS_E_STATE, "not a stripe of gs_db_create_stripe");
    static_assert(sizeof(hipIpcMemHandle_t) <= GS_STRIPE_HANDLE_BYTES, "IPC handle size");
    HIP_TRY(hipSetDevice(db->device));
    hipIpcMemHandle_t h;
    HIP_TRY(hipIpcGetMemHandle(&h, db->d_rec));
    memset(handle, 0, GS_STRIPE_HANDLE_BYTES);
    memcpy(handle, &h, sizeof(h));
    return GS_OK;
}

extern "C" int gs_db_stripe_attach(gs_db *db, int stripe, const void *handle) {
    if (!db || !handle) return fail(GS_E_INVALID, "NULL argument");
    if (!db->striped() || db->group) return fail(GS_E_STATE, "not a stripe of gs_db_create_stripe");
    if (stripe < 0 || stripe >= db->n_parts || stripe == db->part) return fail(GS_E_INVALID, "stripe out of range (or this handle's own)");
    if (db->present & (1u << stripe)) return fail(GS_E_STATE, "stripe already attached");
    if (db->live_runs > 0) return fail(GS_E_STATE, "the store has runs");
    HIP_TRY(hipSetDevice(db->device));
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    void *p = nullptr;
    HIP_TRY(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    db->ipc_opened.push_back(p);
    db_set_stripe(db, stripe, (const u64 *)p);
    return GS_OK;
}

// ---- a built store on the host (from the builder or from a store file) and its way into HBM
struct StoreImage {
    int k = 0, n_values = 0;
    int64_t n_entries = 0, n_stored = 0, n_in_records = 0;
    int b = 0, vbits = 0, rec_bits = 0, max_disp = 0;
    const u64 *table = nullptr;
    size_t table_words = 0;
    const u64 *gate = nullptr;
    size_t gate_words = 0;
    const uint32_t *mgate = nullptr;
    size_t mgate_words = 0;
    uint32_t mgate_ctx = 0;  // the gate is keyed by gs_gate_ctx_key
    const u64 *rec = nullptr;
    size_t rec_words = 0;
    const int32_t *parent = nullptr, *depth = nullptr, *tin = nullptr, *tout = nullptr;
};  // (table / gate / mgate / rec may lie in host memory or in a device's: store_upload copies with hipMemcpyDefault)

// one handle: the whole store (stripes <= 1), or stripe `part` of `stripes` -- the record buckets gs_stripe_first(part) up
// to gs_stripe_first(part + 1) and the table buckets gs_tab_stripe_first(..) in ONE allocation (one IPC handle), and
// everything else (gates, tree) in full
static int store_upload(const StoreImage &im, int dev_no, int stripes, int part, gs_db **res) {
    *res = nullptr;
    HIP_TRY(hipSetDevice(dev_no));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev_no));
    gs_db *db = new gs_db();
    db->device = dev_no;
    db->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const int n_values = im.n_values;
    const int64_t n_rec_total = im.rec_words ? (int64_t)1 << im.rec_bits : 0;
    int64_t rfirst = 0, rlocal = n_rec_total, tfirst = 0, tlocal = (int64_t)1 << im.b;
    if (stripes > 1) {
        rfirst = (int64_t)gs_stripe_first((uint32_t)im.rec_bits, (uint32_t)stripes, (uint32_t)part);
        rlocal = (int64_t)gs_stripe_first((uint32_t)im.rec_bits, (uint32_t)stripes, (uint32_t)part + 1) - rfirst;
        tfirst = (int64_t)gs_tab_stripe_first((uint32_t)im.b, (uint32_t)stripes, (uint32_t)part);
        tlocal = (int64_t)gs_tab_stripe_first((uint32_t)im.b, (uint32_t)stripes, (uint32_t)part + 1) - tfirst;
    }
    const size_t tbytes = (size_t)tlocal * GS_SLOTS_PER_BUCKET * sizeof(u64);
    const size_t rbytes = (size_t)rlocal * GS_REC_WORDS * sizeof(u64);
    const u64 *rsrc = im.rec + (size_t)rfirst * GS_REC_WORDS;
    const u64 *tsrc = im.table + (size_t)tfirst * GS_SLOTS_PER_BUCKET;
    hipError_t e = hipSuccess;
    if (stripes > 1) {
        e = hipMalloc((void **)&db->d_rec, rbytes + tbytes);
        if (e == hipSuccess) e = hipMemcpy(db->d_rec, rsrc, rbytes, hipMemcpyDefault);
        if (e == hipSuccess) db->d_table = db->d_rec + (size_t)rlocal * GS_REC_WORDS;
    } else {
        e = hipMalloc((void **)&db->d_table, tbytes);
        if (e == hipSuccess && rbytes) e = hipMalloc((void **)&db->d_rec, rbytes);
        if (e == hipSuccess && rbytes) e = hipMemcpy(db->d_rec, rsrc, rbytes, hipMemcpyDefault);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&db->d_tree, sizeof(int32_t) * 4 * (size_t)n_values);
    if (e == hipSuccess && im.gate_words) e = hipMalloc((void **)&db->d_gate, im.gate_words * sizeof(u64));
    if (e == hipSuccess && im.gate_words) e = hipMemcpy(db->d_gate, im.gate, im.gate_words * sizeof(u64), hipMemcpyDefault);
    if (e == hipSuccess && im.mgate_words) e = hipMalloc((void **)&db->d_mgate, im.mgate_words * sizeof(uint32_t));
    if (e == hipSuccess && im.mgate_words) e = hipMemcpy(db->d_mgate, im.mgate, im.mgate_words * sizeof(uint32_t), hipMemcpyDefault);
    if (e == hipSuccess) e = hipMemcpy(db->d_table, tsrc, tbytes, hipMemcpyDefault);
    if (e == hipSuccess) e = hipMemcpy(db->d_tree, im.parent, sizeof(int32_t) * n_values, hipMemcpyDefault);
    if (e == hipSuccess) e = hipMemcpy(db->d_tree + n_values, im.depth, sizeof(int32_t) * n_values, hipMemcpyDefault);
    if (e == hipSuccess) e = hipMemcpy(db->d_tree + 2 * (size_t)n_values, im.tin, sizeof(int32_t) * n_values, hipMemcpyDefault);
    if (e == hipSuccess) e = hipMemcpy(db->d_tree + 3 * (size_t)n_values, im.tout, sizeof(int32_t) * n_values, hipMemcpyDefault);
    if (e != hipSuccess) {
        if (stripes > 1) db->d_table = nullptr;  // (inside d_rec)
        db_free(db);
        return fail(e == hipErrorOutOfMemory ? GS_E_NOMEM : GS_E_HIP, std::string("store upload: ") + hipGetErrorString(e));
    }
    db->info.k = im.k;
    db->info.n_values = n_values;
    db->info.n_entries = im.n_entries;
    db->info.n_stored = im.n_stored;
    db->info.n_in_records = im.n_in_records;
    db->info.rec_bytes = (int64_t)(im.rec_words * sizeof(u64));
    db->n_rec = n_rec_total;
    db->dev.rec = stripes > 1 ? nullptr : db->d_rec;
    db->dev.rec_bits = (uint32_t)im.rec_bits;
    db->info.n_buckets = (int64_t)1 << im.b;
    db->info.table_bytes = (int64_t)(im.table_words * sizeof(u64));
    db->info.max_displacement = im.max_disp;
    db->info.value_bits = im.vbits;
    db->dev.table = stripes > 1 ? nullptr : db->d_table;
    db->dev.gate = db->d_gate;
    db->dev.gate_mask = im.gate_words ? (u64)im.gate_words - 1 : 0;
    db->info.gate_bytes = (int64_t)(im.gate_words * sizeof(u64));
    db->dev.mgate = db->d_mgate;
    db->dev.mgate_bits = 0;
    while (((size_t)1 << db->dev.mgate_bits) < im.mgate_words) db->dev.mgate_bits++;
    db->dev.mgate_ctx = im.mgate_ctx;
    db->info.mgate_bytes = (int64_t)(im.mgate_words * sizeof(uint32_t));
    db->dev.bucket_bits = (uint32_t)im.b;
    db->dev.vbits = (uint32_t)im.vbits;
    db->dev.bucket_mask = (1ULL << im.b) - 1;
    db->dev.k = im.k;
    db->dev.n_values = n_values;
    db->dev.parent = db->d_tree;
    db->dev.depth = db->d_tree + n_values;
    db->dev.tin = db->d_tree + 2 * (size_t)n_values;
    db->dev.tout = db->d_tree + 3 * (size_t)n_values;
    if (stripes > 1) {
        db->n_parts = stripes;
        db->part = part;
        db->rec_first = rfirst;
        db->rec_local = rlocal;
        db->tab_first = tfirst;
        db->tab_local = tlocal;
        db->info.n_stripes = stripes;
        db->info.stripe = part;
        db->info.stripe_bytes = (int64_t)(rbytes + tbytes);
        db->dev.n_parts = (uint32_t)stripes;
        db_set_stripe(db, part, db->d_rec);
    }
    *res = db;
    return GS_OK;
}

// stripes <= 1: out[0] on `device`.  stripe_only >= 0: out[0] = that stripe on `device` (the others arrive through
// gs_db_stripe_attach).  Else every stripe in this process, out[p] on stripe_devices[p]: the handles know each other's
// stripes, which belong to all of them together.
static int store_place(const StoreImage &im, int device, int stripes, const int *stripe_devices, int stripe_only, gs_db **out) {
    if (stripes > 1 && im.rec_words == 0)
        return fail(GS_E_UNSUPPORTED, "a striped store needs super-k-mer records (k >= 19, at most 2^21 values, a non-empty store)");
    if (stripes <= 1) return store_upload(im, device, 1, 0, out);
    if (stripe_only >= 0) return store_upload(im, device, stripes, stripe_only, out);
    int rc = GS_OK;
    auto group = std::make_shared<StripeGroup>();
    std::vector<gs_db *> made;
    for (int p = 0; p < stripes && rc == GS_OK; p++) {
        gs_db *db = nullptr;
        rc = store_upload(im, stripe_devices[p], stripes, p, &db);
        if (rc == GS_OK) {
            made.push_back(db);
            group->allocs.push_back({db->device, db->d_rec});
            db->group = group;
        }
    }
    for (int p = 0; p < stripes && rc == GS_OK; p++)
        for (int q = 0; q < stripes && rc == GS_OK; q++) {
            if (q == p) continue;
            if (made[(size_t)p]->device != made[(size_t)q]->device) {
                hipSetDevice(made[(size_t)p]->device);
                int can = 0;
                hipDeviceCanAccessPeer(&can, made[(size_t)p]->device, made[(size_t)q]->device);
                if (!can)
                    rc = fail(GS_E_UNSUPPORTED, "the devices of a striped store need peer access to each other");
                else {
                    const hipError_t pe = hipDeviceEnablePeerAccess(made[(size_t)q]->device, 0);
                    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) rc = fail(GS_E_HIP, std::string("peer access: ") + hipGetErrorString(pe));
                    (void)hipGetLastError();
                }
            }
            if (rc == GS_OK) db_set_stripe(made[(size_t)p], q, made[(size_t)q]->d_rec);
        }
    if (rc != GS_OK) {
        for (gs_db *db : made) db_free(db);
        return rc;
    }
    for (int p = 0; p < stripes; p++) out[p] = made[(size_t)p];
    return GS_OK;
}

// a zero-filled array whose pages are first touched by whoever writes them (calloc of a large block hands out untouched
// zero pages): std::vector would fill gigabytes from one thread before the parallel writers start
template <typename T>
struct Zeroed {
    T *p = nullptr;
    size_t n = 0;
    Zeroed() = default;
    Zeroed(const Zeroed &) = delete;
    Zeroed &operator=(const Zeroed &) = delete;
    size_t mapped = 0;
    ~Zeroed() { release(); }
    void release() {
        if (p) munmap(p, mapped);
        p = nullptr;
        n = mapped = 0;
    }
    void reset(size_t count) {
        release();
        if (count == 0) return;
        // anonymous mapping = untouched zero pages; transparent huge pages where the kernel grants them: the builder's
        // random accesses into gigabyte arrays (cuckoo slots, record lines, gate) otherwise miss the TLB every time
        const size_t bytes = (count * sizeof(T) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) throw std::bad_alloc();
        madvise(m, bytes, MADV_HUGEPAGE);
        p = (T *)m;
        n = count;
        mapped = bytes;
    }
    bool empty() const { return n == 0; }
    size_t size() const { return n; }
    T *data() { return p; }
    const T *data() const { return p; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
};

// [0, n) in n_thr contiguous slices, one thread each (fn(t, lo, hi)); slices do not depend on timing
template <typename F>
static void parallel_slices(int n_thr, int64_t n, F fn) {
    if (n_thr <= 1 || n < 4096) {
        fn(0, (int64_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < n_thr; t++) th.emplace_back([=] { fn(t, n * t / n_thr, n * (t + 1) / n_thr); });
    for (auto &x : th) x.join();
}

// GS_BUILD_TRACE=1: phase times of the host builder on stderr (developer aid)
struct BuildTrace {
    bool on;
    std::chrono::steady_clock::time_point t0;
    BuildTrace() : on(getenv("GS_BUILD_TRACE") != nullptr && atoi(getenv("GS_BUILD_TRACE")) != 0), t0(std::chrono::steady_clock::now()) {}
    void mark(const char *what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[gs_db_create] %-28s %8.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    }
};


// parent / depth / pre-order interval of every value index from parent_vi (-1 root, -2 no node; NULL: every value a root)
static int tree_arrays(int32_t n_values, const int32_t *parent_vi, std::vector<int32_t> &parent, std::vector<int32_t> &depth,
                       std::vector<int32_t> &tin, std::vector<int32_t> &tout) {
    parent.assign((size_t)n_values, -1);
    depth.assign((size_t)n_values, 0);
    tin.assign((size_t)n_values, 0);
    tout.assign((size_t)n_values, 0);
    for (int32_t v = 0; v < n_values; v++) {
        int32_t p = parent_vi ? parent_vi[v] : -1;
        if (p < -2 || p >= n_values || p == v) return fail(GS_E_INVALID, "parent_vi out of range");
        parent[v] = p;
    }
    std::vector<std::vector<int32_t>> kids(n_values);
    std::vector<int32_t> roots;
    for (int32_t v = 0; v < n_values; v++) {
        if (parent[v] >= 0) {
            if (parent[parent[v]] == -2) return fail(GS_E_INVALID, "parent_vi points at a value without a node");
            kids[parent[v]].push_back(v);
        } else if (parent[v] == -1)
            roots.push_back(v);
    }
    int32_t counter = 0, visited = 0;
    std::vector<std::pair<int32_t, size_t>> stack;
    for (int32_t root : roots) {
        stack.push_back({root, 0});
        tin[root] = counter++;
        depth[root] = 0;
        visited++;
        while (!stack.empty()) {
            auto &top = stack.back();
            if (top.second < kids[top.first].size()) {
                int32_t c = kids[top.first][top.second++];
                tin[c] = counter++;
                depth[c] = (int32_t)stack.size();
                visited++;
                stack.push_back({c, 0});
            } else {
                tout[top.first] = counter;
                stack.pop_back();
            }
        }
    }
    int32_t nodes = 0;
    for (int32_t v = 0; v < n_values; v++) nodes += parent[v] != -2;
    if (visited != nodes) return fail(GS_E_INVALID, "parent_vi contains a cycle");
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------------
// the layout built on the device (gs_layout_build.hip): same rules as the host builder below
// ---------------------------------------------------------------------------------------------------
enum { GS_LB_N_E = 0, GS_LB_N_T, GS_LB_N_M, GS_LB_N_H, GS_LB_IN_REC, GS_LB_OVERFLOW, GS_LB_MAX_DISP, GS_LB_N_WIN, GS_LB_N_CTX, GS_LB_N_HINT, GS_LB_COUNTERS };
extern "C" hipError_t gs_lb_perkey(const int64_t *kmers, const int32_t *vidx, int64_t n, int k, const int32_t *parent, uint32_t *e_gh, uint32_t *e_ohi,
                                   uint32_t *e_olo, uint32_t *e_vj, u64 *e_sort, u64 *e_sort2, u64 *t_key, int32_t *t_val, uint32_t *m_gh, uint32_t *h_gh, uint32_t *h_ctx,
                                   u64 *cnt, hipStream_t stream);
extern "C" hipError_t gs_lb_sort_entries(u64 *e_sort, u64 *e_sort2, u64 *sort_alt, uint32_t *perm, uint32_t *perm_alt, int64_t n, const uint32_t *e_gh,
                                         const uint32_t *e_ohi, const uint32_t *e_olo, const uint32_t *e_vj, uint32_t *s_gh, uint32_t *s_ohi,
                                         uint32_t *s_olo, uint32_t *s_vj, hipStream_t stream);
extern "C" hipError_t gs_lb_groups(const uint32_t *s_gh, int64_t n, uint32_t *head, uint32_t *group, uint32_t *g_start, int64_t *n_groups,
                                   hipStream_t stream);
extern "C" hipError_t gs_lb_cluster(const uint32_t *s_gh, const uint32_t *s_ohi, const uint32_t *s_olo, const uint32_t *s_vj, const uint32_t *g_start,
                                    int64_t n_groups, int k, uint8_t *assign, u64 *w_hi, u64 *w_lo, uint32_t *w_valid, uint32_t *w_gh, u64 *cnt,
                                    hipStream_t stream);
extern "C" hipError_t gs_lb_place(const uint32_t *w_valid, const uint32_t *w_gh, int64_t n_w, uint32_t rec_bits, int max_rounds, uint32_t *slot,
                                  uint32_t *slot2, uint32_t *claim, uint32_t *state, uint32_t *win_bucket, u64 *changes, int *rounds_done,
                                  hipStream_t stream);
extern "C" hipError_t gs_lb_lines(const uint32_t *slot, uint32_t rec_bits, const u64 *w_hi, const u64 *w_lo, const uint32_t *w_valid, const uint32_t *s_gh,
                                  const uint32_t *s_ohi, const uint32_t *s_olo, const uint32_t *s_vj, const uint32_t *group, const uint8_t *assign,
                                  const uint32_t *win_bucket, int64_t n_e, int k, u64 *rec, u64 *t_key, int32_t *t_val, uint32_t *m_gh, u64 *cnt,
                                  hipStream_t stream);
extern "C" hipError_t gs_lb_more(const uint32_t *m_gh, int64_t n_m, uint32_t rec_bits, u64 *rec, hipStream_t stream);
extern "C" hipError_t gs_lb_table(const u64 *t_key, const int32_t *t_val, int64_t n_t, int b, int vbits, u64 *rot_a, u64 *rot_b, int32_t *val_b,
                                  int32_t *val_c, uint32_t *start, uint32_t *left, uint32_t *pos, uint32_t *perm, uint32_t *perm_alt, uint32_t *fill_a,
                                  uint32_t *fill_b, u64 *table, int64_t *overflow, int *max_disp, hipStream_t stream);
extern "C" hipError_t gs_lb_distinct(uint32_t *h_gh, uint32_t *h_alt, int64_t n_h, u64 *d_scratch, int64_t *distinct, uint32_t **sorted,
                                     hipStream_t stream);
extern "C" hipError_t gs_lb_gate(const uint32_t *h_gh, int64_t n_h, int ctx, uint32_t mgate_bits, uint32_t *mgate, hipStream_t stream);
extern "C" hipError_t gs_lb_hint_collect(const uint32_t *w_valid, const uint32_t *w_gh, const u64 *w_hi, const u64 *w_lo, const uint32_t *win_bucket,
                                         int64_t n_w, uint32_t rec_bits, int k, uint32_t *hint_gh, uint32_t *hint_cx, u64 *cnt, hipStream_t stream);
extern "C" hipError_t gs_lb_hint(const uint32_t *hint_gh, const uint32_t *hint_cx, int64_t n, int ctx, uint32_t mgate_bits, uint32_t *mgate,
                                 hipStream_t stream);

// device scratch that goes when the build is over (or fails)
struct DevPool {
    std::vector<void *> all;
    hipError_t err = hipSuccess;
    ~DevPool() {
        for (void *p : all) hipFree(p);
    }
    template <typename T>
    T *get(size_t count) {
        void *p = nullptr;
        if (err == hipSuccess) err = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
        if (err != hipSuccess) return nullptr;
        all.push_back(p);
        return (T *)p;
    }
    void drop(void *p) {
        for (void *&x : all)
            if (x == p && p) {
                hipFree(p);
                x = nullptr;
            }
    }
    void *keep(void *p) {  // the pointer leaves the pool (it becomes part of the store)
        for (void *&x : all)
            if (x == p) x = nullptr;
        return p;
    }
};

// 1: built (*out), 0: this store is for the host builder (no record entries), < 0: error
// kmers / vidx: host arrays, or (on_dev) arrays in this device's memory, which are only read
static int db_create_on_device(gs_db **out, int device, int k, int64_t n, const int64_t *kmers, const int32_t *vidx, int32_t n_values,
                               const std::vector<int32_t> &parent, const std::vector<int32_t> &depth, const std::vector<int32_t> &tin,
                               const std::vector<int32_t> &tout, BuildTrace &trace, bool on_dev = false) {
    hipStream_t stream = nullptr;
    DevPool pool;
    auto bad = [&](hipError_t e, const char *what) {
        return fail(e == hipErrorOutOfMemory ? GS_E_NOMEM : GS_E_HIP, std::string("device layout build (") + what + "): " + hipGetErrorString(e));
    };
#define LB_TRY(expr, what)                                        \
    do {                                                          \
        hipError_t e_ = (expr);                                   \
        if (e_ == hipSuccess && pool.err != hipSuccess) e_ = pool.err; \
        if (e_ != hipSuccess) return bad(e_, what);               \
    } while (0)
    const size_t sn = (size_t)n;
    int64_t *d_k = on_dev ? const_cast<int64_t *>(kmers) : pool.get<int64_t>(sn);
    int32_t *d_v = on_dev ? const_cast<int32_t *>(vidx) : pool.get<int32_t>(sn);
    int32_t *d_parent = pool.get<int32_t>((size_t)n_values);
    uint32_t *e_gh = pool.get<uint32_t>(sn), *e_ohi = pool.get<uint32_t>(sn), *e_olo = pool.get<uint32_t>(sn), *e_vj = pool.get<uint32_t>(sn);
    u64 *e_sort = pool.get<u64>(sn), *e_sort2 = pool.get<u64>(sn);
    u64 *t_key = pool.get<u64>(sn);
    int32_t *t_val = pool.get<int32_t>(sn);
    uint32_t *m_gh = pool.get<uint32_t>(2 * sn), *h_gh = pool.get<uint32_t>(2 * sn);
    // context keys of the gate (gs_gate_ctx_key): collected for stores big enough to possibly need them
    int64_t ctx_min_distinct = 12000000;  // ~20 % of the 15-mer minimizer space: a filter over the minimizers alone starts to leak
    if (const char *e = getenv("GS_GATE_CTX_MIN_DISTINCT")) ctx_min_distinct = std::max<int64_t>(0, atoll(e));
    const bool ctx_possible = k >= GS_CTX_MIN_K && n >= ctx_min_distinct;
    uint32_t *h_ctx = ctx_possible ? pool.get<uint32_t>(2 * sn) : nullptr;
    u64 *cnt = pool.get<u64>(GS_LB_COUNTERS);
    LB_TRY(pool.err, "buffers");
    if (!on_dev) {
        LB_TRY(hipMemcpy(d_k, kmers, sn * sizeof(int64_t), hipMemcpyHostToDevice), "upload");
        LB_TRY(hipMemcpy(d_v, vidx, sn * sizeof(int32_t), hipMemcpyHostToDevice), "upload");
    }
    LB_TRY(hipMemcpy(d_parent, parent.data(), (size_t)n_values * sizeof(int32_t), hipMemcpyHostToDevice), "upload");
    LB_TRY(hipMemset(cnt, 0, GS_LB_COUNTERS * sizeof(u64)), "counters");
    trace.mark("device: upload");
    LB_TRY(gs_lb_perkey(d_k, d_v, n, k, d_parent, e_gh, e_ohi, e_olo, e_vj, e_sort, e_sort2, t_key, t_val, m_gh, h_gh, h_ctx, cnt, stream), "per key");
    u64 c[GS_LB_COUNTERS];
    LB_TRY(hipMemcpy(c, cnt, sizeof(c), hipMemcpyDeviceToHost), "per key");
    if (!on_dev) {
        pool.drop(d_k);
        pool.drop(d_v);
    }
    const int64_t n_e = (int64_t)c[GS_LB_N_E], n_h = (int64_t)c[GS_LB_N_H], n_ctx = (int64_t)c[GS_LB_N_CTX];
    trace.mark("device: per key");
    if (n_e == 0) return 0;  // nothing for records: the host builder's table-only store
    // ---- entries sorted by minimizer, minimizers -> windows
    const size_t se = (size_t)n_e;
    u64 *sort_alt = pool.get<u64>(se);
    uint32_t *perm = pool.get<uint32_t>(se), *perm_alt = pool.get<uint32_t>(se);
    uint32_t *s_gh = pool.get<uint32_t>(se), *s_ohi = pool.get<uint32_t>(se), *s_olo = pool.get<uint32_t>(se), *s_vj = pool.get<uint32_t>(se);
    LB_TRY(pool.err, "sort buffers");
    LB_TRY(gs_lb_sort_entries(e_sort, e_sort2, sort_alt, perm, perm_alt, n_e, e_gh, e_ohi, e_olo, e_vj, s_gh, s_ohi, s_olo, s_vj, stream), "sort");
    pool.drop(e_sort2);
    pool.drop(e_gh);
    pool.drop(e_ohi);
    pool.drop(e_olo);
    pool.drop(e_vj);
    pool.drop(e_sort);
    pool.drop(sort_alt);
    pool.drop(perm);
    pool.drop(perm_alt);
    trace.mark("device: sort");
    uint32_t *head = pool.get<uint32_t>(se), *group = pool.get<uint32_t>(se), *g_start = pool.get<uint32_t>(se + 1);
    LB_TRY(pool.err, "group buffers");
    int64_t n_groups = 0;
    LB_TRY(gs_lb_groups(s_gh, n_e, head, group, g_start, &n_groups, stream), "groups");
    pool.drop(head);
    const size_t sw = 2 * (size_t)n_groups;
    uint8_t *assign = pool.get<uint8_t>(se);
    u64 *w_hi = pool.get<u64>(sw), *w_lo = pool.get<u64>(sw);
    uint32_t *w_valid = pool.get<uint32_t>(sw), *w_gh = pool.get<uint32_t>(sw);
    LB_TRY(pool.err, "window buffers");
    LB_TRY(gs_lb_cluster(s_gh, s_ohi, s_olo, s_vj, g_start, n_groups, k, assign, w_hi, w_lo, w_valid, w_gh, cnt, stream), "cluster");
    LB_TRY(hipMemcpy(c, cnt, sizeof(c), hipMemcpyDeviceToHost), "cluster");
    pool.drop(g_start);
    const int64_t n_win = (int64_t)c[GS_LB_N_WIN];
    trace.mark("device: cluster");
    // ---- buckets
    double rload = 0.4;
    if (const char *e = getenv("GS_REC_LOAD")) {
        const double v = atof(e);
        if (v > 0.01 && v <= 0.5) rload = v;
    }
    int rec_bits = 4;
    while (rec_bits < 29 && (double)((size_t)1 << rec_bits) * rload < (double)n_win) rec_bits++;
    const size_t n_rec = (size_t)1 << rec_bits;
    int max_rounds = 64, rounds_done = 0;
    if (const char *e = getenv("GS_REC_ROUNDS")) max_rounds = std::max(2, std::min(5000, atoi(e)));
    uint32_t *slot = pool.get<uint32_t>(n_rec), *slot2 = pool.get<uint32_t>(n_rec), *claim = pool.get<uint32_t>(n_rec);
    uint32_t *wstate = pool.get<uint32_t>(sw), *win_bucket = pool.get<uint32_t>(sw);
    u64 *d_rec = pool.get<u64>(n_rec * GS_REC_WORDS);
    LB_TRY(pool.err, "record buffers");
    LB_TRY(gs_lb_place(w_valid, w_gh, (int64_t)sw, (uint32_t)rec_bits, max_rounds, slot, slot2, claim, wstate, win_bucket, cnt + GS_LB_OVERFLOW,
                       &rounds_done, stream), "placement");
    pool.drop(wstate);
    pool.drop(slot2);
    pool.drop(claim);
    LB_TRY(gs_lb_lines(slot, (uint32_t)rec_bits, w_hi, w_lo, w_valid, s_gh, s_ohi, s_olo, s_vj, group, assign, win_bucket, n_e, k, d_rec, t_key, t_val,
                       m_gh, cnt, stream), "lines");
    LB_TRY(hipMemcpy(c, cnt, sizeof(c), hipMemcpyDeviceToHost), "lines");
    const int64_t n_t = (int64_t)c[GS_LB_N_T], n_m = (int64_t)c[GS_LB_N_M], n_in_records = (int64_t)c[GS_LB_IN_REC];
    LB_TRY(gs_lb_more(m_gh, n_m, (uint32_t)rec_bits, d_rec, stream), "more bits");
    // the windows in their second bucket (gs_mgate_hint); at most every window
    uint32_t *hint_gh = pool.get<uint32_t>(std::max<size_t>(sw, 1)), *hint_cx = pool.get<uint32_t>(std::max<size_t>(sw, 1));
    LB_TRY(pool.err, "hint buffers");
    LB_TRY(gs_lb_hint_collect(w_valid, w_gh, w_hi, w_lo, win_bucket, (int64_t)sw, (uint32_t)rec_bits, k, hint_gh, hint_cx, cnt, stream), "hints");
    LB_TRY(hipMemcpy(c, cnt, sizeof(c), hipMemcpyDeviceToHost), "hints");
    const int64_t n_hint = (int64_t)c[GS_LB_N_HINT];
    LB_TRY(hipDeviceSynchronize(), "more bits");
    for (void *p : {(void *)slot, (void *)win_bucket, (void *)assign, (void *)w_hi, (void *)w_lo, (void *)w_valid, (void *)w_gh, (void *)s_gh, (void *)s_ohi,
                    (void *)s_olo, (void *)s_vj, (void *)group, (void *)m_gh})
        pool.drop(p);
    trace.mark("device: placement + lines");
    if (trace.on)
        fprintf(stderr, "[gs_db_create] device: %lld entries, %lld minimizers, %lld windows in %lld buckets (%.3f, %d bidding rounds), %lld k-mers in records, %lld in the table\n",
                (long long)n_e, (long long)n_groups, (long long)n_win, (long long)n_rec, (double)n_win / (double)n_rec, rounds_done,
                (long long)n_in_records, (long long)n_t);
    // ---- overflow table
    const int vbits = std::max(1, bits_for((u64)n_values));
    double load = 3.0;
    if (const char *e = getenv("GS_BUCKET_LOAD")) {
        double v = atof(e);
        if (v > 0.05 && v <= 6.0) load = v;
    }
    int b = std::max(vbits + 1, 4);
    while ((double)(1ULL << b) * load < (double)n_t) b++;
    u64 *d_table = nullptr;
    int max_disp = 0;
    {
        const size_t st = (size_t)std::max<int64_t>(n_t, 1);
        u64 *rot_a = pool.get<u64>(st), *rot_b = pool.get<u64>(st);
        int32_t *val_b = pool.get<int32_t>(st), *val_c = pool.get<int32_t>(st);
        uint32_t *t_start = pool.get<uint32_t>(st), *t_left = pool.get<uint32_t>(st), *t_pos = pool.get<uint32_t>(st), *t_perm = pool.get<uint32_t>(st),
                 *t_perm2 = pool.get<uint32_t>(st);
        LB_TRY(pool.err, "table buffers");
        for (;; b++) {
            if (b > 29) return fail(GS_E_UNSUPPORTED, "store too large for 32-bit slot indices");
            uint32_t *fill_a = pool.get<uint32_t>((size_t)1 << b), *fill_b = pool.get<uint32_t>((size_t)1 << b);
            d_table = pool.get<u64>(((size_t)1 << b) * GS_SLOTS_PER_BUCKET);
            LB_TRY(pool.err, "table buffers");
            int64_t overflow = 0;
            LB_TRY(gs_lb_table(t_key, t_val, n_t, b, vbits, rot_a, rot_b, val_b, val_c, t_start, t_left, t_pos, t_perm, t_perm2, fill_a, fill_b, d_table,
                               &overflow, &max_disp, stream), "table");
            pool.drop(fill_a);
            pool.drop(fill_b);
            if (overflow == 0) break;
            pool.drop(d_table);  // a key found no slot within GS_MAX_DISP buckets: twice the buckets
        }
        for (void *p : {(void *)rot_a, (void *)rot_b, (void *)val_b, (void *)val_c, (void *)t_start, (void *)t_left, (void *)t_pos, (void *)t_perm, (void *)t_perm2}) pool.drop(p);
    }
    pool.drop(t_key);
    pool.drop(t_val);
    trace.mark("device: overflow table");
    // ---- minimizer gate
    uint32_t *h_alt = pool.get<uint32_t>((size_t)std::max<int64_t>(n_h, 1));
    LB_TRY(pool.err, "gate buffers");
    int64_t distinct = 0;
    uint32_t *h_sorted = h_gh;
    LB_TRY(gs_lb_distinct(h_gh, h_alt, n_h, cnt, &distinct, &h_sorted, stream), "distinct minimizers");
    // a store whose minimizers fill a good part of the minimizer space gets the gate keyed by minimizer + context instead
    int64_t n_gate = n_h;
    uint32_t mgate_ctx = 0;
    if (h_ctx != nullptr && distinct >= ctx_min_distinct) {
        pool.drop(h_alt);
        h_alt = pool.get<uint32_t>((size_t)std::max<int64_t>(n_ctx, 1));
        LB_TRY(pool.err, "gate buffers");
        LB_TRY(gs_lb_distinct(h_ctx, h_alt, n_ctx, cnt, &distinct, &h_sorted, stream), "distinct context keys");
        n_gate = n_ctx;
        mgate_ctx = 1;
    }
    double bits_per_min = 16.0;
    if (const char *e = getenv("GS_MGATE_BITS_PER_MIN")) bits_per_min = std::max(1.0, atof(e));
    int mgate_bits = 6;
    while (mgate_bits < 30 && (double)((size_t)32 << mgate_bits) < (double)distinct * bits_per_min) mgate_bits++;
    uint32_t *d_mgate = pool.get<uint32_t>((size_t)1 << mgate_bits);
    LB_TRY(pool.err, "gate");
    LB_TRY(gs_lb_gate(h_sorted, n_gate, (int)mgate_ctx, (uint32_t)mgate_bits, d_mgate, stream), "gate");
    LB_TRY(gs_lb_hint(hint_gh, hint_cx, n_hint, (int)mgate_ctx, (uint32_t)mgate_bits, d_mgate, stream), "gate hints");
    int32_t *d_tree = pool.get<int32_t>(4 * (size_t)n_values);
    LB_TRY(pool.err, "tree");
    LB_TRY(hipMemcpy(d_tree, parent.data(), sizeof(int32_t) * (size_t)n_values, hipMemcpyHostToDevice), "tree");
    LB_TRY(hipMemcpy(d_tree + n_values, depth.data(), sizeof(int32_t) * (size_t)n_values, hipMemcpyHostToDevice), "tree");
    LB_TRY(hipMemcpy(d_tree + 2 * (size_t)n_values, tin.data(), sizeof(int32_t) * (size_t)n_values, hipMemcpyHostToDevice), "tree");
    LB_TRY(hipMemcpy(d_tree + 3 * (size_t)n_values, tout.data(), sizeof(int32_t) * (size_t)n_values, hipMemcpyHostToDevice), "tree");
    LB_TRY(hipDeviceSynchronize(), "gate");
    trace.mark("device: gate");
#undef LB_TRY
    hipDeviceProp_t prop;
    gs_db *db = new gs_db();
    db->device = device;
    db->n_cu = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    db->d_rec = (u64 *)pool.keep(d_rec);
    db->d_table = (u64 *)pool.keep(d_table);
    db->d_mgate = (uint32_t *)pool.keep(d_mgate);
    db->d_tree = (int32_t *)pool.keep(d_tree);
    db->n_rec = (int64_t)n_rec;
    db->info.k = k;
    db->info.n_values = n_values;
    db->info.n_entries = n;
    db->info.n_stored = n_t + n_in_records;
    db->info.n_in_records = n_in_records;
    db->info.rec_bytes = (int64_t)(n_rec * GS_REC_WORDS * sizeof(u64));
    db->info.n_buckets = (int64_t)1 << b;
    db->info.table_bytes = (int64_t)(((size_t)1 << b) * GS_SLOTS_PER_BUCKET * sizeof(u64));
    db->info.max_displacement = max_disp;
    db->info.value_bits = vbits;
    db->info.gate_bytes = 0;
    db->info.mgate_bytes = (int64_t)(((size_t)1 << mgate_bits) * sizeof(uint32_t));
    db->dev.table = db->d_table;
    db->dev.gate = nullptr;
    db->dev.gate_mask = 0;
    db->dev.mgate = db->d_mgate;
    db->dev.mgate_bits = (uint32_t)mgate_bits;
    db->dev.mgate_ctx = mgate_ctx;
    db->dev.rec = db->d_rec;
    db->dev.rec_bits = (uint32_t)rec_bits;
    db->dev.bucket_bits = (uint32_t)b;
    db->dev.vbits = (uint32_t)vbits;
    db->dev.bucket_mask = (1ULL << b) - 1;
    db->dev.k = k;
    db->dev.n_values = n_values;
    db->dev.parent = db->d_tree;
    db->dev.depth = db->d_tree + n_values;
    db->dev.tin = db->d_tree + 2 * (size_t)n_values;
    db->dev.tout = db->d_tree + 3 * (size_t)n_values;
    *out = db;
    return 1;
}

// fused: the store serves the fused kernels (gs_match_submit*, gs_match_segments) and may keep k-mers in super-k-mer
// records; a partition store (gs_db_create_part, any n_parts) keeps every key in the table, where gs_match_probe_keys looks
static int db_create_impl(gs_db **out, int device, int k, int64_t n, const int64_t *kmers, const int32_t *vidx,
                          int32_t n_values, const int32_t *parent_vi, int n_parts, int part, bool fused, int stripes,
                          const int *stripe_devices, int stripe_only) {
    if (!out) return fail(GS_E_INVALID, "out is NULL");
    for (int p = 0; p < (stripes > 1 && stripe_only < 0 ? stripes : 1); p++) out[p] = nullptr;
    if (k < 1 || k > 31) return fail(GS_E_INVALID, "k must be in [1,31]");
    if (n < 0 || n_values < 1 || n_values > (1 << 24) || (n > 0 && (!kmers || !vidx)))
        return fail(GS_E_INVALID, "bad store arrays (n_values must be in [1, 2^24])");
    // GS_BUILD_DRYRUN=1 (developer aid, with GS_BUILD_TRACE): the host layout is built and timed, nothing is uploaded
    const bool dryrun = getenv("GS_BUILD_DRYRUN") != nullptr && atoi(getenv("GS_BUILD_DRYRUN")) != 0;
    int rc = dryrun ? GS_OK : use_device(device);
    if (rc) return rc;
    BuildTrace trace;
    bool ascending = true;
    for (int64_t i = 0; i < n; i++) {
        if (vidx[i] < 0 || vidx[i] >= n_values) return fail(GS_E_INVALID, "value_idx out of range");
        if (i && kmers[i] <= kmers[i - 1]) ascending = false;
        if (kmers[i] < 0 || (k < 31 && (u64)kmers[i] >> (2 * k))) return fail(GS_E_INVALID, "k-mer exceeds 2k bits");
    }
    if (!ascending) {
        // KMerStore.visit order of a RadixKMerStore (C/store/RadixKMerStore.java:714-729: by radix bucket, then by the
        // remaining bits) is not ascending.  The table does not care about the order, only that the keys are distinct.
        std::vector<int64_t> sorted(kmers, kmers + n);
        std::sort(sorted.begin(), sorted.end());
        if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) return fail(GS_E_INVALID, "k-mers must be distinct");
    }
    // ---- tree arrays
    std::vector<int32_t> parent, depth, tin, tout;
    rc = tree_arrays(n_values, parent_vi, parent, depth, tin, tout);
    if (rc) return rc;
    trace.mark("checks + tree");
    // ---- the layout on the device: plain fused stores with records (k >= 19, at most 2^21 values).  GS_BUILD_HOST=1 keeps
    // the host builder (the reference for the layout rules; striped / partition / table-only stores are always its job)
    {
        bool on_device = fused && n_parts == 1 && !dryrun && k >= GS_MIN_K && n_values <= GS_REC_MAX_VALUES && n > 0 && n < ((int64_t)1 << 31);
        if (const char *e = getenv("GS_BUILD_HOST")) on_device = on_device && atoi(e) == 0;
        if (const char *e = getenv("GS_MGATE")) on_device = on_device && atoi(e) != 0;
        if (const char *e = getenv("GS_RECORDS")) on_device = on_device && atoi(e) != 0;
        if (on_device && stripes <= 1) {
            const int drc = db_create_on_device(out, device, k, n, kmers, vidx, n_values, parent, depth, tin, tout, trace);
            if (drc == 1 && getenv("GS_BUILD_VERIFY") != nullptr && atoi(getenv("GS_BUILD_VERIFY")) != 0) {
                // developer aid: the image through the checks gs_db_load applies to a store file
                const int vrc = db_self_check(*out);
                if (vrc) {
                    db_free(*out);
                    *out = nullptr;
                    return vrc;
                }
            }
            if (drc != 0) return drc < 0 ? drc : GS_OK;
        } else if (on_device) {
            // a striped store: the layout on one device (the first stripe's, or this process's), the stripes cut from that
            gs_db *whole = nullptr;
            const int bdev = stripe_only >= 0 ? device : stripe_devices[0];
            HIP_TRY(hipSetDevice(bdev));
            const int drc = db_create_on_device(&whole, bdev, k, n, kmers, vidx, n_values, parent, depth, tin, tout, trace);
            if (drc < 0) return drc;
            if (drc == 1) {
                // (the stripes are copied out of the build's arrays, device to device: no host image)
                StoreImage im{};
                im.k = k;
                im.n_values = n_values;
                im.n_entries = n;
                im.n_stored = whole->info.n_stored;
                im.n_in_records = whole->info.n_in_records;
                im.b = (int)whole->dev.bucket_bits;
                im.vbits = (int)whole->dev.vbits;
                im.rec_bits = (int)whole->dev.rec_bits;
                im.max_disp = whole->info.max_displacement;
                im.table = whole->d_table;
                im.table_words = (size_t)whole->info.n_buckets * GS_SLOTS_PER_BUCKET;
                im.mgate = whole->d_mgate;
                im.mgate_words = (size_t)1 << whole->dev.mgate_bits;
                im.mgate_ctx = whole->dev.mgate_ctx;
                im.rec = whole->d_rec;
                im.rec_words = (size_t)whole->n_rec * GS_REC_WORDS;
                im.parent = parent.data();
                im.depth = depth.data();
                im.tin = tin.data();
                im.tout = tout.data();
                rc = store_place(im, device, stripes, stripe_devices, stripe_only, out);
                db_free(whole);
                trace.mark("stripes");
                return rc;
            }
        }
    }
    // ---- keys
    // Every reachable stored k-mer is looked at from both strands (gs_layout.h: gs_choose_minimizer).  If both views pick
    // the same minimizer occurrence the k-mer can live in a super-k-mer record; otherwise (and for every k-mer when the
    // store has no records) it goes to the ordinary table under its mixed key.
    struct RecEntry {
        uint32_t bucket, gh, ohi, olo, vj;  // vj = value index << 5 | offset j of the k-mer in its window
    };
    std::vector<u64> hkey;
    std::vector<int32_t> hval;
    std::vector<uint32_t> hmin;   // order hash of the minimizer(s) of every reachable key of ANY partition (k >= GS_MIN_K)
    std::vector<uint32_t> hmore;  // minimizers whose record bucket must send mismatching probes on to the table
    std::vector<uint32_t> hint_min;  // minimizers with a window in their second candidate bucket (gs_mgate_hint)
    std::vector<RecEntry> rents;
    hkey.reserve((size_t)n);
    hval.reserve((size_t)n);
    bool want_mgate = k >= GS_MIN_K;
    if (const char *e = getenv("GS_MGATE")) want_mgate = want_mgate && atoi(e) != 0;
    bool want_rec = want_mgate && fused && n_parts == 1 && n_values <= GS_REC_MAX_VALUES;
    if (const char *e = getenv("GS_RECORDS")) want_rec = want_rec && atoi(e) != 0;
    if (want_mgate) hmin.reserve((size_t)n);
    const uint32_t kmask = (1u << k) - 1u;
    {
        // per-key work (planes, representative, hash, minimizer) on all host cores: slices are converted independently
        // and concatenated in order, so the result does not depend on the thread count
        int n_thr = (int)std::min<int64_t>(std::max<unsigned>(1, std::thread::hardware_concurrency()), 32);
        if (const char *e = getenv("GS_BUILD_THREADS")) n_thr = std::max(1, std::min(64, atoi(e)));
        if (n < 200000) n_thr = 1;
        struct Slice {
            std::vector<u64> h;
            std::vector<int32_t> v;
            std::vector<uint32_t> m, more;
            std::vector<RecEntry> r;
        };
        std::vector<Slice> sl((size_t)n_thr);
        auto work = [&](int t) {
            const int64_t lo = n * t / n_thr, hi = n * (t + 1) / n_thr;
            Slice &o = sl[(size_t)t];
            if (want_rec)
                o.r.reserve((size_t)(hi - lo));
            else {
                o.h.reserve((size_t)(hi - lo));
                o.v.reserve((size_t)(hi - lo));
            }
            if (want_mgate) o.m.reserve((size_t)(hi - lo));
            for (int64_t i = lo; i < hi; i++) {
                uint32_t fhi, flo, phi, plo;
                bool reachable = java_to_planes((u64)kmers[i], k, fhi, flo);
                if (!reachable || parent[vidx[i]] == -2) continue;
                bool in_record = false;
                // the minimizer gate of a partition covers the keys of ALL partitions: the rank that encodes a read
                // (gs_match_encode) uses it to decide which k-mers are worth routing to their owner at all
                if (want_mgate) {
                    const uint32_t rhi = gs_brev32(fhi) >> (32 - k), rlo = (gs_brev32(flo) >> (32 - k)) ^ kmask;
                    uint32_t gh1, gh2, ohi1, olo1, ohi2, olo2;
                    int j1, j2;
                    // the reverse strand sees the same canonical 15-mers at mirrored offsets: one pass over the order hashes
                    // serves both views (the leftmost of equal ranks in one view is the rightmost in the other)
                    int p1 = 0, p2 = 0;
                    {
                        uint32_t best1 = 0xffffffffu, best2 = 0xffffffffu;
                        const int last = k - GS_MIN_L;
                        for (int d = 0; d <= last; d++) {
                            const uint32_t h = gs_lmer_hash((fhi >> d) & 0x7fffu, (flo >> d) & 0x7fffu);
                            const uint32_t x1 = gs_lmer_rank(h, (uint32_t)d), x2 = gs_lmer_rank(h, (uint32_t)(last - d));
                            best1 = x1 < best1 ? x1 : best1;
                            best2 = x2 < best2 ? x2 : best2;
                        }
                        p1 = (int)(best1 & 0xffu);
                        p2 = (int)(best2 & 0xffu);
                    }
                    gs_min_oriented(fhi, flo, rhi, rlo, k, p1, gh1, ohi1, olo1, j1);
                    gs_min_oriented(rhi, rlo, fhi, flo, k, p2, gh2, ohi2, olo2, j2);
                    const bool same = gh1 == gh2 && j1 == j2 && ohi1 == ohi2 && olo1 == olo2;
                    o.m.push_back(gh1);
                    if (gh2 != gh1) o.m.push_back(gh2);
                    if (want_rec) {
                        if (same) {
                            o.r.push_back(RecEntry{0u, gh1, ohi1, olo1, ((uint32_t)vidx[i] << 5) | (uint32_t)j1});
                            in_record = true;
                        } else {  // two strand views: one table slot, reachable from both record buckets
                            o.more.push_back(gh1);
                            o.more.push_back(gh2);
                        }
                    }
                }
                if (in_record) continue;
                gs_rep_planes(fhi, flo, k, kmask, phi, plo);  // the orientation the table files this k-mer under
                const u64 hk = gs_mix_planes(phi, plo);
                if (n_parts > 1 && (int)((hk >> GS_OWNER_SHIFT) % (u64)n_parts) != part) continue;  // another rank owns it
                o.h.push_back(hk);
                o.v.push_back(vidx[i]);
            }
        };
        if (n_thr == 1) {
            work(0);
        } else {
            std::vector<std::thread> th;
            for (int t = 0; t < n_thr; t++) th.emplace_back(work, t);
            for (auto &x : th) x.join();
        }
        // the slices one behind the other (every thread copies its own)
        std::vector<size_t> oh((size_t)n_thr + 1, 0), om((size_t)n_thr + 1, 0), omo((size_t)n_thr + 1, 0), orr((size_t)n_thr + 1, 0);
        for (int t = 0; t < n_thr; t++) {
            oh[(size_t)t + 1] = oh[(size_t)t] + sl[(size_t)t].h.size();
            om[(size_t)t + 1] = om[(size_t)t] + sl[(size_t)t].m.size();
            omo[(size_t)t + 1] = omo[(size_t)t] + sl[(size_t)t].more.size();
            orr[(size_t)t + 1] = orr[(size_t)t] + sl[(size_t)t].r.size();
        }
        hkey.resize(oh[(size_t)n_thr]);
        hval.resize(oh[(size_t)n_thr]);
        hmin.resize(om[(size_t)n_thr]);
        hmore.resize(omo[(size_t)n_thr]);
        rents.resize(orr[(size_t)n_thr]);
        auto gather = [&](int t) {
            Slice &o = sl[(size_t)t];
            std::copy(o.h.begin(), o.h.end(), hkey.begin() + (ptrdiff_t)oh[(size_t)t]);
            std::copy(o.v.begin(), o.v.end(), hval.begin() + (ptrdiff_t)oh[(size_t)t]);
            std::copy(o.m.begin(), o.m.end(), hmin.begin() + (ptrdiff_t)om[(size_t)t]);
            std::copy(o.more.begin(), o.more.end(), hmore.begin() + (ptrdiff_t)omo[(size_t)t]);
            std::copy(o.r.begin(), o.r.end(), rents.begin() + (ptrdiff_t)orr[(size_t)t]);
            Slice().h.swap(o.h);
            std::vector<int32_t>().swap(o.v);
            std::vector<uint32_t>().swap(o.m);
            std::vector<RecEntry>().swap(o.r);
        };
        if (n_thr == 1)
            gather(0);
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < n_thr; t++) th.emplace_back(gather, t);
            for (auto &x : th) x.join();
        }
    }
    trace.mark("per-key minimizers");
    // ---- number of DISTINCT minimizers, estimated by linear counting on a 2^26-bit sketch (a sort of the 32-bit hashes
    // costs seconds for tens of millions of keys; +-1 % is plenty for power-of-two sizes): sizes the gate and the records
    const int64_t nm = (int64_t)hmin.size();
    double distinct_min = 0;
    if (want_mgate && nm > 0) {
        const size_t sketch_bits = (size_t)1 << 26;
        std::vector<u64> sketch(sketch_bits / 64, 0);
        for (int64_t i = 0; i < nm; i++) {
            const uint32_t x = hmin[(size_t)i] * 0x9E3779B1u;  // (the hash is a bijection of the 15-mer: no extra collisions)
            sketch[(x >> 6) & (sketch.size() - 1)] |= 1ULL << (x & 63);
        }
        size_t ones = 0;
        for (u64 w : sketch) ones += (size_t)__builtin_popcountll(w);
        const double zero_frac = std::max(1e-9, 1.0 - (double)ones / (double)sketch_bits);
        distinct_min = std::min((double)nm, -(double)sketch_bits * std::log(zero_frac));
    }
    trace.mark("distinct-minimizer sketch");
    // ---- super-k-mer records (gs_layout.h): the k-mers of one minimizer are clustered into windows, every window gets one
    // of its minimizer's two buckets (cuckoo placement); the k-mers of windows that find no place join the table keys
    Zeroed<u64> rec;
    int rec_bits = 0;
    int64_t n_in_records = 0;
    if (want_rec && !rents.empty()) {
        const int64_t ne = (int64_t)rents.size();
        int n_thr = (int)std::min<int64_t>(std::max<unsigned>(1, std::thread::hardware_concurrency()), 32);
        if (const char *e = getenv("GS_BUILD_THREADS")) n_thr = std::max(1, std::min(64, atoi(e)));
        if (ne < 200000) n_thr = 1;
        // counting sort by the top bits of the minimizer hash, then every chunk is sorted and clustered on its own
        const int cbits = 8, cshift = 32 - cbits, n_chunks = 1 << cbits;
        std::vector<int64_t> cstart((size_t)n_chunks + 1, 0);
        std::vector<RecEntry> sorted((size_t)ne);
        {   // stable: thread t's entries of a chunk go behind those of the threads before it
            std::vector<std::vector<int64_t>> hist((size_t)n_thr, std::vector<int64_t>((size_t)n_chunks, 0));
            parallel_slices(n_thr, ne, [&](int t, int64_t lo, int64_t hi) {
                std::vector<int64_t> &h = hist[(size_t)t];
                for (int64_t i = lo; i < hi; i++) h[(size_t)(rents[(size_t)i].gh >> cshift)]++;
            });
            int64_t run = 0;
            for (int c = 0; c < n_chunks; c++) {
                cstart[(size_t)c] = run;
                for (int t = 0; t < n_thr; t++) {
                    const int64_t x = hist[(size_t)t][(size_t)c];
                    hist[(size_t)t][(size_t)c] = run;
                    run += x;
                }
            }
            cstart[(size_t)n_chunks] = run;
            parallel_slices(n_thr, ne, [&](int t, int64_t lo, int64_t hi) {
                std::vector<int64_t> &cur = hist[(size_t)t];
                for (int64_t i = lo; i < hi; i++) sorted[(size_t)cur[(size_t)(rents[(size_t)i].gh >> cshift)]++] = rents[(size_t)i];
            });
        }
        std::vector<RecEntry>().swap(rents);
        trace.mark("counting sort into chunks");
        struct Win {
            u64 whi, wlo, known;
            uint32_t valid, gh;
            int32_t val[17];  // offsets 0 .. k - 15 <= 16
        };
        std::vector<std::vector<Win>> chunk_wins((size_t)n_chunks);
        auto cluster_chunk = [&](int c) {
            RecEntry *lo = sorted.data() + cstart[(size_t)c], *hi = sorted.data() + cstart[(size_t)c + 1];
            std::sort(lo, hi, [](const RecEntry &a, const RecEntry &b) {
                if (a.gh != b.gh) return a.gh < b.gh;
                if ((a.vj & 31u) != (b.vj & 31u)) return (a.vj & 31u) < (b.vj & 31u);
                if (a.ohi != b.ohi) return a.ohi < b.ohi;
                return a.olo < b.olo;
            });
            std::vector<Win> &wins = chunk_wins[(size_t)c];
            for (RecEntry *g0 = lo; g0 < hi;) {  // the entries of one minimizer
                RecEntry *g1 = g0;
                while (g1 < hi && g1->gh == g0->gh) g1++;
                const size_t first = wins.size();
                for (RecEntry *e = g0; e < g1; e++) {
                    const int j = (int)(e->vj & 31u);
                    const u64 eh = (u64)e->ohi << j, el = (u64)e->olo << j, km = (u64)kmask << j;
                    size_t w = first;
                    for (; w < wins.size(); w++) {  // first window that agrees on every base both know and has offset j free
                        const Win &W = wins[w];
                        if (!((W.valid >> j) & 1u) && ((W.whi ^ eh) & W.known & km) == 0 && ((W.wlo ^ el) & W.known & km) == 0) break;
                    }
                    if (w == wins.size()) {
                        wins.push_back(Win{});
                        wins.back().gh = e->gh;
                    }
                    Win &W = wins[w];
                    W.whi |= eh;
                    W.wlo |= el;
                    W.known |= km;
                    W.valid |= 1u << j;
                    W.val[j] = (int32_t)(e->vj >> 5);
                }
                // the fullest windows of a minimizer first: only two of them can find a bucket
                std::stable_sort(wins.begin() + (ptrdiff_t)first, wins.end(),
                                 [](const Win &a, const Win &b) { return __builtin_popcount(a.valid) > __builtin_popcount(b.valid); });
                g0 = g1;
            }
        };
        if (n_thr == 1) {
            for (int c = 0; c < n_chunks; c++) cluster_chunk(c);
        } else {
            std::atomic<int> next{0};
            std::vector<std::thread> th;
            for (int t = 0; t < n_thr; t++)
                th.emplace_back([&] {
                    for (int c; (c = next.fetch_add(1)) < n_chunks;) cluster_chunk(c);
                });
            for (auto &x : th) x.join();
        }
        std::vector<RecEntry>().swap(sorted);
        trace.mark("sort + cluster windows");
        size_t n_win = 0;
        for (auto &v : chunk_wins) n_win += v.size();
        // windows per bucket before the bucket count doubles: with the power-of-two rounding 0.2 .. 0.4 (two-choice cuckoo
        // placement with one window per bucket works up to 0.5)
        double rload = 0.4;
        if (const char *e = getenv("GS_REC_LOAD")) {
            const double v = atof(e);
            if (v > 0.01 && v <= 0.5) rload = v;
        }
        rec_bits = 4;
        while (rec_bits < 29 && (double)((size_t)1 << rec_bits) * rload < (double)n_win) rec_bits++;
        const size_t n_rec = (size_t)1 << rec_bits;
        // cuckoo placement.  Windows are numbered in chunk order (within a minimizer: fullest first).  Two parallel rounds
        // whose outcome does not depend on timing -- every window bids for its first bucket, then the losers for their
        // second bucket among those still free, and the LOWEST number wins a bucket (atomic min) -- place ~95 % of the
        // windows; the rest goes through the sequential eviction walk.
        std::vector<const Win *> order;
        order.reserve(n_win);
        for (auto &v : chunk_wins)
            for (const Win &W0 : v) order.push_back(&W0);
        if (order.size() >= 0xffffffffull) return fail(GS_E_UNSUPPORTED, "store too large: more than 2^32 windows");
        const uint32_t EMPTY = 0xffffffffu;
        const int64_t nw = (int64_t)order.size();
        std::vector<uint32_t> slot(n_rec, EMPTY), bid;
        std::vector<uint8_t> placed((size_t)nw, 0);
        parallel_slices(n_thr, nw, [&](int, int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; i++) {
                uint32_t *p = &slot[gs_rec_bucket(order[(size_t)i]->gh, (uint32_t)rec_bits, 0)];
                uint32_t seen = __atomic_load_n(p, __ATOMIC_RELAXED);
                while ((uint32_t)i < seen && !__atomic_compare_exchange_n(p, &seen, (uint32_t)i, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
                }
            }
        });
        parallel_slices(n_thr, nw, [&](int, int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; i++) placed[(size_t)i] = slot[gs_rec_bucket(order[(size_t)i]->gh, (uint32_t)rec_bits, 0)] == (uint32_t)i;
        });
        bid.assign(n_rec, EMPTY);
        parallel_slices(n_thr, nw, [&](int, int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; i++) {
                if (placed[(size_t)i]) continue;
                const uint32_t b1 = gs_rec_bucket(order[(size_t)i]->gh, (uint32_t)rec_bits, 1);
                if (slot[b1] != EMPTY) continue;  // (round 1 is over: slot is read-only here)
                uint32_t *p = &bid[b1];
                uint32_t seen = __atomic_load_n(p, __ATOMIC_RELAXED);
                while ((uint32_t)i < seen && !__atomic_compare_exchange_n(p, &seen, (uint32_t)i, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
                }
            }
        });
        parallel_slices(n_thr, (int64_t)n_rec, [&](int, int64_t lo, int64_t hi) {
            for (int64_t b = lo; b < hi; b++)
                if (bid[(size_t)b] != EMPTY) {  // (only buckets that were free after round 1 got bids)
                    slot[(size_t)b] = bid[(size_t)b];
                    placed[bid[(size_t)b]] = 1;
                }
        });
        std::vector<uint32_t>().swap(bid);
        std::vector<const Win *> homeless;
        uint64_t rng = 0x9E3779B97F4A7C15ULL;
        // A walk that has not ended after 16 evictions rarely ends at all: what blocks it is not the load but the
        // minimizers with two windows -- their two buckets are taken for good, and a window whose both buckets are such
        // (or a second pair that shares a bucket with the first) has no place whatever the walk does.  At 473 M k-mers
        // 19 % of the minimizers have two windows and 5 % more (the space of 15-mers that can BE a minimizer is only
        // ~4^15 / 2 / 9 = 60 M): 2.1 M windows are third siblings, 2.9 M walks are hopeless (the same count with 48 or 500
        // steps; 500 steps of three cache misses each made this loop a third of the build).  Those windows go to the table.
        int max_kicks = 16;
        if (const char *e = getenv("GS_REC_KICKS")) max_kicks = std::max(1, std::min(5000, atoi(e)));
        int64_t n_walks = 0, n_kicks = 0, n_sib_first = 0, n_sib_later = 0, n_exhausted = 0;
        for (int64_t wi = 0; wi < nw; wi++) {
            if (placed[(size_t)wi]) continue;
            n_walks++;
            uint32_t cur = (uint32_t)wi;
            uint32_t at = gs_rec_bucket(order[cur]->gh, (uint32_t)rec_bits, 0);
            for (int kick = 0; cur != EMPTY && kick < max_kicks; kick++) {
                n_kicks++;
                const uint32_t gh = order[cur]->gh;
                const uint32_t b0 = gs_rec_bucket(gh, (uint32_t)rec_bits, 0), b1 = gs_rec_bucket(gh, (uint32_t)rec_bits, 1);
                if (slot[b0] == EMPTY) {
                    slot[b0] = cur;
                    cur = EMPTY;
                } else if (slot[b1] == EMPTY) {
                    slot[b1] = cur;
                    cur = EMPTY;
                } else {  // both taken: evict one of them (not the bucket we just came from) and carry it on
                    const bool sib0 = order[slot[b0]]->gh == gh, sib1 = order[slot[b1]]->gh == gh;
                    if (sib0 && sib1) {  // its own sibling windows hold both buckets: no place for a third
                        (kick == 0 ? n_sib_first : n_sib_later)++;
                        break;
                    }
                    rng = rng * 6364136223846793005ULL + 1442695040888963407ULL;
                    uint32_t victim = (rng >> 40) & 1 ? b1 : b0;
                    if (kick > 0 && victim == at && b0 != b1) victim = victim == b0 ? b1 : b0;
                    if (order[slot[victim]]->gh == gh) victim = victim == b0 ? b1 : b0;  // (fuller siblings stay)
                    std::swap(cur, slot[victim]);
                    at = victim;
                    if (kick == max_kicks - 1) n_exhausted++;
                }
            }
            if (cur != EMPTY) homeless.push_back(order[cur]);
        }
        std::vector<uint8_t>().swap(placed);
        if (trace.on) {
            int64_t hist[5] = {0, 0, 0, 0, 0}, wsum[5] = {0, 0, 0, 0, 0};
            for (int64_t i = 0; i < nw;) {
                int64_t j = i;
                while (j < nw && order[(size_t)j]->gh == order[(size_t)i]->gh) j++;
                const int64_t c = j - i, slot_no = c >= 17 ? 4 : c >= 3 ? 3 : c;
                hist[slot_no]++;
                wsum[slot_no] += c;
                i = j;
            }
            fprintf(stderr, "[gs_db_create] minimizers with 1 / 2 / 3..16 / 17+ windows: %lld / %lld / %lld / %lld (windows: %lld / %lld / %lld / %lld)\n",
                    (long long)hist[1], (long long)hist[2], (long long)hist[3], (long long)hist[4], (long long)wsum[1], (long long)wsum[2],
                    (long long)wsum[3], (long long)wsum[4]);
        }
        if (trace.on)
            fprintf(stderr, "[gs_db_create] %lld windows in %lld buckets (%.3f), %lld walks, %lld steps, %lld windows to the table (third sibling at once %lld, later %lld, walk given up %lld)\n",
                    (long long)nw, (long long)n_rec, (double)nw / (double)n_rec, (long long)n_walks, (long long)n_kicks, (long long)homeless.size(),
                    (long long)n_sib_first, (long long)n_sib_later, (long long)n_exhausted);
        trace.mark("cuckoo placement");
        rec.reset(n_rec * GS_REC_WORDS);
        {
            std::vector<int64_t> cnt((size_t)n_thr, 0);
            std::vector<std::vector<uint32_t>> hint_of((size_t)n_thr);
            parallel_slices(n_thr, (int64_t)n_rec, [&](int t, int64_t lo, int64_t hi) {
                int64_t mine = 0;
                for (int64_t b = lo; b < hi; b++) {
                    if (b + 8 < hi && slot[(size_t)b + 8] != EMPTY) __builtin_prefetch(order[slot[(size_t)b + 8]]);
                    if (slot[(size_t)b] == EMPTY) continue;
                    const Win *W = order[slot[(size_t)b]];
                    if ((uint32_t)b != gs_rec_bucket(W->gh, (uint32_t)rec_bits, 0)) hint_of[(size_t)t].push_back(W->gh);  // gs_mgate_hint
                    u64 *rp = rec.data() + (size_t)b * GS_REC_WORDS;
                    rp[0] = W->whi;
                    rp[1] = W->wlo | ((u64)W->valid << GS_REC_WIN_BITS);
                    for (uint32_t m = W->valid; m; m &= m - 1) {
                        const int j = __builtin_ctz(m);
                        rp[2 + j / 3] |= (u64)W->val[j] << (GS_REC_VAL_BITS * (j % 3));
                        mine++;
                    }
                }
                cnt[(size_t)t] = mine;
            });
            for (int64_t x : cnt) n_in_records += x;
            for (const auto &h : hint_of) hint_min.insert(hint_min.end(), h.begin(), h.end());
        }
        for (const Win *W : homeless) {  // their k-mers become table keys, reachable through both buckets' `more` bit
            for (uint32_t m = W->valid; m; m &= m - 1) {
                const int j = __builtin_ctz(m);
                uint32_t phi, plo;
                gs_rep_planes((uint32_t)(W->whi >> j) & kmask, (uint32_t)(W->wlo >> j) & kmask, k, kmask, phi, plo);
                hkey.push_back(gs_mix_planes(phi, plo));
                hval.push_back(W->val[j]);
            }
            hmore.push_back(W->gh);
        }
        // the `more` bit of both buckets of these minimizers (OR is order-free: threads + atomics give the same lines)
        parallel_slices(n_thr, (int64_t)hmore.size(), [&](int, int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; i++)
                for (int ch = 0; ch < 2; ch++) {
                    u64 *rp = rec.data() + (size_t)gs_rec_bucket(hmore[(size_t)i], (uint32_t)rec_bits, ch) * GS_REC_WORDS;
                    if (__atomic_load_n(&rp[2], __ATOMIC_RELAXED) & GS_REC_MORE) continue;  // (set by an earlier entry: every word of the line has it then, or will)
                    for (int x = 2; x < GS_REC_WORDS; x++) __atomic_fetch_or(&rp[x], GS_REC_MORE, __ATOMIC_RELAXED);
                }
        });
        trace.mark("record lines + more bits");
    }  // (no eligible k-mer at all: no records, every key is in the table and the kernels take the table path)
    const int64_t ns = (int64_t)hkey.size();
    const int vbits = std::max(1, bits_for((u64)n_values));
    // keys per 8-slot bucket before the bucket count doubles: with the power-of-two rounding the table ends up at
    // 1.5 .. 3 keys per bucket.  Measured on a 47 M-k-mer store: 2.8 keys/bucket (1 GiB) is 5 % faster than 1.4 (2 GiB).
    double load = 3.0;
    if (const char *e = getenv("GS_BUCKET_LOAD")) {
        double v = atof(e);
        if (v > 0.05 && v <= 6.0) load = v;
    }
    int b = std::max(vbits + 1, stripes > 1 ? 5 : 4);  // slot bits: (62-b) rem + 2 disp + vbits value + 1 seen <= 64
    while ((double)(1ULL << b) * load < (double)ns) b++;
    std::vector<u64> table;
    int max_disp = 0;
    for (;; b++) {
        if (b > 29) return fail(GS_E_UNSUPPORTED, "store too large for 32-bit slot indices");
        const u64 nb = 1ULL << b, mask = nb - 1;
        table.assign((size_t)nb * GS_SLOTS_PER_BUCKET, 0);
        std::vector<uint8_t> fill((size_t)nb, 0);
        bool ok = true;
        max_disp = 0;
        for (int64_t i = 0; i < ns && ok; i++) {
            const u64 h = hkey[i], home = h & mask, rem = h >> b;
            int d = 0;
            for (; d <= GS_MAX_DISP; d++) {
                const u64 bk = (home + d) & mask;
                if (fill[bk] < GS_SLOTS_PER_BUCKET) {
                    table[bk * GS_SLOTS_PER_BUCKET + fill[bk]++] =
                        (rem << (vbits + 3)) | ((u64)d << (vbits + 1)) | ((u64)(hval[i] + 1) << 1);
                    break;
                }
            }
            if (d > GS_MAX_DISP) ok = false;
            max_disp = std::max(max_disp, d);
        }
        if (ok) break;
    }
    trace.mark("overflow table");
    // ---- L2-resident gate (see gs_layout.h): ~8-16 bits per key, only if it fits the per-XCD L2 budget
    std::vector<u64> gate;
    {
        size_t max_bytes = (size_t)2 << 20;
        if (const char *e = getenv("GS_GATE_MAX_BYTES")) max_bytes = (size_t)atoll(e);
        size_t bits_per_key = 8;  // rounded up to a power-of-two word count: 8..16 bits per key
        if (const char *e = getenv("GS_GATE_BITS_PER_KEY")) bits_per_key = (size_t)std::max(1, atoi(e));
        int wbits = 3;
        while (((size_t)64 << wbits) < (size_t)ns * bits_per_key) wbits++;
        if (ns > 0 && ((size_t)8 << wbits) <= max_bytes && b + wbits <= GS_GATE_FIELD_SHIFT) {
            gate.assign((size_t)1 << wbits, 0);
            const u64 gmask = ((u64)1 << wbits) - 1;
            for (int64_t i = 0; i < ns; i++) gate[(hkey[i] >> b) & gmask] |= gs_gate_bits(hkey[i]);
        }
    }
    // ---- minimizer gate (gs_layout.h): 16-32 bits per DISTINCT minimizer, 2 bits set per entry, 32-bit words
    Zeroed<uint32_t> mgate;
    int mgate_bits = 0;
    if (want_mgate && nm > 0) {
        const double distinct = distinct_min;
        double bits_per_min = 16.0;  // rounded up to a power-of-two word count: 16..32 bits per distinct minimizer
        if (const char *e = getenv("GS_MGATE_BITS_PER_MIN")) bits_per_min = std::max(1.0, atof(e));
        mgate_bits = 6;
        while (mgate_bits < 30 && (double)((size_t)32 << mgate_bits) < distinct * bits_per_min) mgate_bits++;
        mgate.reset((size_t)1 << mgate_bits);
        int g_thr = (int)std::min<int64_t>(std::max<unsigned>(1, std::thread::hardware_concurrency()), 32);
        if (const char *e = getenv("GS_BUILD_THREADS")) g_thr = std::max(1, std::min(64, atoi(e)));
        parallel_slices(nm < 200000 ? 1 : g_thr, nm, [&](int, int64_t lo, int64_t hi) {  // (OR is order-free)
            for (int64_t i = lo; i < hi; i++) {
                uint32_t *w = mgate.data() + gs_mgate_word(hmin[(size_t)i], (uint32_t)mgate_bits);
                const uint32_t bits = gs_mgate_bits(hmin[(size_t)i]);
                if ((__atomic_load_n(w, __ATOMIC_RELAXED) & bits) != bits) __atomic_fetch_or(w, bits, __ATOMIC_RELAXED);
            }
        });
        for (uint32_t h : hint_min) mgate.data()[gs_mgate_word(h, (uint32_t)mgate_bits)] |= gs_mgate_hint(h);  // windows in their second bucket
    }
    trace.mark("gates");
    StoreImage im{};
    im.k = k;
    im.n_values = n_values;
    im.n_entries = n;
    im.n_stored = ns + n_in_records;
    im.n_in_records = n_in_records;
    im.b = b;
    im.vbits = vbits;
    im.rec_bits = rec_bits;
    im.max_disp = max_disp;
    im.table = table.data();
    im.table_words = table.size();
    im.gate = gate.data();
    im.gate_words = gate.size();
    im.mgate = mgate.data();
    im.mgate_words = mgate.size();
    im.rec = rec.data();
    im.rec_words = rec.size();
    im.parent = parent.data();
    im.depth = depth.data();
    im.tin = tin.data();
    im.tout = tout.data();
    if (dryrun) return fail(GS_E_UNSUPPORTED, "GS_BUILD_DRYRUN: layout built, nothing uploaded");
    rc = store_place(im, device, stripes, stripe_devices, stripe_only, out);
    trace.mark("upload");
    return rc;
}

extern "C" int gs_db_get_info(const gs_db *db, gs_db_info *info) {
    if (!db || !info) return fail(GS_E_INVALID, "NULL argument");
    *info = db->info;
    return GS_OK;
}

// ---- native store file: the built device image (table, gate, tree) so that a later process skips the rebuild.
// Layout: GsStoreFileHeader | table (n_buckets*8 u64) | gate (gate_bytes) | tree (4*n_values int32)
struct GsStoreFileHeader {
    char magic[8];  // "GSSTORE9"
    gs_db_info info;
    uint32_t bucket_bits, vbits;  // (bit 31 of vbits: the minimizer gate is keyed by gs_gate_ctx_key, GsDbDev::mgate_ctx)
    uint64_t gate_words;
    uint64_t mgate_words;  // 32-bit words, a power of two
    uint64_t rec_buckets;  // super-k-mer record buckets (GS_REC_WORDS words each), a power of two or 0
    uint64_t checksum;     // store_checksum over table | gate | mgate | records | tree
};

// two running 64-bit sums over the payload's 32-bit words (Fletcher style): position sensitive, one pass
struct StoreChecksum {
    uint64_t a = 1, b = 0;
    // a and b after the words of [p, p + bytes): slices are summed by all cores with a = b = 0 -- S = sum of the words, T = sum
    // of their prefix sums -- and chained: a' = a + S, b' = b + n * a + T (the same numbers as the word-by-word loop, mod 2^64)
    void add(const void *p, size_t bytes) {
        const uint32_t *w = (const uint32_t *)p;
        const size_t n = bytes / 4;
        int n_thr = n < ((size_t)1 << 22) ? 1 : (int)std::min<unsigned>(32, std::max<unsigned>(1, std::thread::hardware_concurrency()));
        std::vector<uint64_t> S((size_t)n_thr, 0), T((size_t)n_thr, 0);
        auto part = [&](int t) {
            const size_t lo = n * (size_t)t / (size_t)n_thr, hi = n * ((size_t)t + 1) / (size_t)n_thr;
            uint64_t sa = 0, sb = 0;
            for (size_t i = lo; i < hi; i++) {
                sa += w[i];
                sb += sa;
            }
            S[(size_t)t] = sa;
            T[(size_t)t] = sb;
        };
        if (n_thr == 1)
            part(0);
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < n_thr; t++) th.emplace_back(part, t);
            for (auto &x : th) x.join();
        }
        for (int t = 0; t < n_thr; t++) {
            const size_t lo = n * (size_t)t / (size_t)n_thr, hi = n * ((size_t)t + 1) / (size_t)n_thr;
            b += (uint64_t)(hi - lo) * a + T[(size_t)t];
            a += S[(size_t)t];
        }
    }
    uint64_t value() const { return a ^ (b << 1) ^ (b >> 63); }
};

// what the kernels index with: every field of a store image that came from a file is checked before it reaches HBM
template <typename Words>
static const char *store_image_defect(const GsStoreFileHeader &h, const Words &table, const Words &rec, const std::vector<int32_t> &tree) {
    const gs_db_info &in = h.info;
    if (in.k < 1 || in.k > 31) return "k outside [1, 31]";
    if (in.n_values < 1 || in.n_values > (1 << 24)) return "n_values outside [1, 2^24]";
    if (h.vbits < 1 || h.vbits > 25 || ((u64)in.n_values >> h.vbits) != 0) return "value bits do not hold n_values";
    if ((int)h.bucket_bits < std::max((int)h.vbits + 1, 4) || h.bucket_bits > 29) return "bucket bits out of range";
    if (in.n_buckets != ((int64_t)1 << h.bucket_bits) || in.table_bytes != in.n_buckets * 64) return "table size does not match the bucket count";
    if (in.value_bits != (int32_t)h.vbits) return "value bits disagree";
    if ((h.gate_words & (h.gate_words - 1)) != 0 || h.gate_words > ((uint64_t)1 << 28)) return "gate size is not a power of two";
    if (h.gate_words && h.bucket_bits + (uint32_t)bits_for(h.gate_words - 1) > GS_GATE_FIELD_SHIFT) return "gate too large for its index field";
    if ((h.mgate_words & (h.mgate_words - 1)) != 0 || h.mgate_words > ((uint64_t)1 << 30)) return "minimizer gate size is not a power of two";
    if (h.mgate_words && in.k < GS_MIN_K) return "minimizer gate on a store with k < 19";
    if (in.gate_bytes != (int64_t)(h.gate_words * 8) || in.mgate_bytes != (int64_t)(h.mgate_words * 4)) return "gate sizes disagree";
    if ((h.rec_buckets & (h.rec_buckets - 1)) != 0 || h.rec_buckets > ((uint64_t)1 << 29) || (h.rec_buckets && !h.mgate_words)) return "record bucket count is not a power of two";
    if (in.rec_bytes != (int64_t)(h.rec_buckets * GS_REC_WORDS * 8)) return "record size disagrees";
    if (h.rec_buckets && in.n_values > GS_REC_MAX_VALUES) return "records on a store with more than 2^21 values";
    if (in.max_displacement < 0 || in.max_displacement > GS_MAX_DISP) return "displacement out of range";
    const int32_t nv = in.n_values;
    const int32_t *parent = tree.data(), *depth = parent + nv, *tin = depth + nv, *tout = tin + nv;
    for (int32_t v = 0; v < nv; v++) {
        if (parent[v] < -2 || parent[v] >= nv || parent[v] == v) return "tree: parent out of range";
        if (parent[v] == -2) continue;
        if (tin[v] < 0 || tout[v] <= tin[v] || tout[v] > nv || depth[v] < 0 || depth[v] >= nv) return "tree: pre-order interval out of range";
        if (parent[v] >= 0) {  // a child's interval nests strictly inside its parent's, one level down: walks end at a root
            const int32_t p = parent[v];
            if (parent[p] == -2 || depth[v] != depth[p] + 1 || tin[v] <= tin[p] || tout[v] > tout[p]) return "tree: child interval not nested in its parent's";
        } else if (depth[v] != 0)
            return "tree: root with a depth";
    }
    // the slots and the record lines on all cores (slices; the first defect of the lowest slice is reported)
    const u64 vmask = ((u64)1 << h.vbits) - 1;
    const int c_off = in.k - GS_MIN_L;
    const size_t n_lines = rec.size() / GS_REC_WORDS;
    const int n_thr = (table.size() + rec.size()) < ((size_t)1 << 22) ? 1 : (int)std::min<unsigned>(32, std::max<unsigned>(1, std::thread::hardware_concurrency()));
    std::vector<int64_t> c_stored((size_t)n_thr, 0), c_rec((size_t)n_thr, 0);
    std::vector<const char *> why((size_t)n_thr, nullptr);
    auto part = [&](int t) {
        const char *bad = nullptr;
        int64_t st = 0, ir = 0;
        for (size_t i = table.size() * (size_t)t / (size_t)n_thr, e = table.size() * ((size_t)t + 1) / (size_t)n_thr; i < e && !bad; i++) {
            const u64 s = table[i];
            if (s == 0) continue;
            const u64 v1 = (s >> 1) & vmask;
            if (v1 == 0 || v1 > (u64)nv || parent[v1 - 1] == -2) bad = "table: slot value without a tree node";
            else if (s & 1) bad = "table: seen bit set in a stored image";
            st++;
        }
        for (size_t b = n_lines * (size_t)t / (size_t)n_thr, e = n_lines * ((size_t)t + 1) / (size_t)n_thr; b < e && !bad; b++) {
            const u64 *rp = rec.data() + b * GS_REC_WORDS;
            if (rp[0] >> GS_REC_WIN_BITS) bad = "records: seen bit set in a stored image";
            uint32_t valid = (uint32_t)(rp[1] >> GS_REC_WIN_BITS);
            if (!bad && (valid >> (c_off + 1))) bad = "records: k-mer offset beyond the window";
            for (; valid && !bad; valid &= valid - 1) {
                const int j = __builtin_ctz(valid);
                const u64 vi = (rp[2 + j / 3] >> (GS_REC_VAL_BITS * (j % 3))) & (GS_REC_MAX_VALUES - 1);
                if (vi >= (u64)nv || parent[vi] == -2) bad = "records: value without a tree node";
                ir++;
            }
        }
        c_stored[(size_t)t] = st;
        c_rec[(size_t)t] = ir;
        why[(size_t)t] = bad;
    };
    if (n_thr == 1)
        part(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_thr; t++) th.emplace_back(part, t);
        for (auto &x : th) x.join();
    }
    int64_t stored = 0, in_records = 0;
    for (int t = 0; t < n_thr; t++) {
        if (why[(size_t)t]) return why[(size_t)t];
        stored += c_stored[(size_t)t];
        in_records += c_rec[(size_t)t];
    }
    if (in_records != in.n_in_records) return "records: entry count disagrees with the header";
    if (stored + in_records != in.n_stored) return "table: entry count disagrees with the header";
    return nullptr;
}

// the store's image back to the host and through store_image_defect (GS_BUILD_VERIFY=1: after every device build)
static int db_self_check(gs_db *db) {
    GsStoreFileHeader h{};
    h.info = db->info;
    h.bucket_bits = db->dev.bucket_bits;
    h.vbits = db->dev.vbits;
    h.gate_words = db->d_gate ? db->dev.gate_mask + 1 : 0;
    h.mgate_words = db->d_mgate ? (uint64_t)1 << db->dev.mgate_bits : 0;
    h.rec_buckets = (uint64_t)db->n_rec;
    const size_t nv = (size_t)db->info.n_values;
    Zeroed<u64> table, rec;
    table.reset((size_t)db->info.n_buckets * GS_SLOTS_PER_BUCKET);
    rec.reset((size_t)h.rec_buckets * GS_REC_WORDS);
    std::vector<int32_t> tree(4 * nv);
    HIP_TRY(hipMemcpy(table.data(), db->d_table, table.size() * sizeof(u64), hipMemcpyDeviceToHost));
    if (!rec.empty()) HIP_TRY(hipMemcpy(rec.data(), db->d_rec, rec.size() * sizeof(u64), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(tree.data(), db->d_tree, tree.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (const char *why = store_image_defect(h, table, rec, tree)) return fail(GS_E_HIP, std::string("GS_BUILD_VERIFY: the device-built store fails the image checks: ") + why);
    if (db->info.n_stored > db->info.n_entries) return fail(GS_E_HIP, "GS_BUILD_VERIFY: more k-mers stored than handed in");
    return GS_OK;
}

extern "C" int gs_db_save(gs_db *db, const char *path) try {
    if (!db || !path) return fail(GS_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(db->device));
    if (db->unique_owner) return fail(GS_E_STATE, "a unique-counting run is active on this store (seen bits are set)");
    if (db->striped()) return fail(GS_E_UNSUPPORTED, "a striped store is not saved as such: save the store built by gs_db_create");
    GsStoreFileHeader h{};
    memcpy(h.magic, "GSSTORE9", 8);
    h.info = db->info;
    h.bucket_bits = db->dev.bucket_bits;
    h.vbits = db->dev.vbits;
    h.gate_words = db->d_gate ? db->dev.gate_mask + 1 : 0;
    h.mgate_words = db->d_mgate ? (uint64_t)1 << db->dev.mgate_bits : 0;
    h.rec_buckets = (uint64_t)db->n_rec;
    const size_t nv = (size_t)db->info.n_values;
    Zeroed<u64> table, rec;  // (untouched pages: the copy from the device is the first to touch them)
    table.reset((size_t)db->info.n_buckets * GS_SLOTS_PER_BUCKET);
    rec.reset((size_t)h.rec_buckets * GS_REC_WORDS);
    std::vector<u64> gate((size_t)h.gate_words);
    if (!rec.empty()) HIP_TRY(hipMemcpy(rec.data(), db->d_rec, rec.size() * sizeof(u64), hipMemcpyDeviceToHost));
    std::vector<uint32_t> mgate((size_t)h.mgate_words);
    std::vector<int32_t> tree(4 * nv);
    HIP_TRY(hipMemcpy(table.data(), db->d_table, table.size() * sizeof(u64), hipMemcpyDeviceToHost));
    if (!gate.empty()) HIP_TRY(hipMemcpy(gate.data(), db->d_gate, gate.size() * sizeof(u64), hipMemcpyDeviceToHost));
    if (!mgate.empty()) HIP_TRY(hipMemcpy(mgate.data(), db->d_mgate, mgate.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(tree.data(), db->d_tree, tree.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    {
        StoreChecksum cs;
        cs.add(table.data(), table.size() * sizeof(u64));
        cs.add(gate.data(), gate.size() * sizeof(u64));
        cs.add(mgate.data(), mgate.size() * sizeof(uint32_t));
        cs.add(rec.data(), rec.size() * sizeof(u64));
        cs.add(tree.data(), tree.size() * sizeof(int32_t));
        h.checksum = cs.value();
    }
    FILE *f = fopen(path, "wb");
    if (!f) return fail(GS_E_IO, std::string("cannot open ") + path);
    if (db->dev.mgate_ctx) h.vbits |= 0x80000000u;  // (the flag travels in bit 31 of the header's vbits word)
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fflush(f) == 0;
    // the payload with several threads (pwrite), as gs_db_load reads it
    const int fd = fileno(f);
    off_t at = (off_t)sizeof(h);
    auto par_write = [&](const void *src, size_t bytes) {
        const int n_thr = bytes < ((size_t)64 << 20) ? 1 : (int)std::min<unsigned>(16, std::max<unsigned>(1, std::thread::hardware_concurrency()));
        std::atomic<bool> good{true};
        auto part = [&](int t) {
            size_t lo = bytes * (size_t)t / (size_t)n_thr;
            const size_t hi = bytes * ((size_t)t + 1) / (size_t)n_thr;
            while (lo < hi) {
                const ssize_t r = pwrite(fd, (const char *)src + lo, std::min<size_t>(hi - lo, (size_t)1 << 30), at + (off_t)lo);
                if (r <= 0) {
                    good = false;
                    return;
                }
                lo += (size_t)r;
            }
        };
        if (n_thr == 1)
            part(0);
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < n_thr; t++) th.emplace_back(part, t);
            for (auto &x : th) x.join();
        }
        at += (off_t)bytes;
        return good.load();
    };
    ok = ok && par_write(table.data(), table.size() * sizeof(u64)) && par_write(gate.data(), gate.size() * sizeof(u64)) &&
         par_write(mgate.data(), mgate.size() * sizeof(uint32_t)) && par_write(rec.data(), rec.size() * sizeof(u64)) &&
         par_write(tree.data(), tree.size() * sizeof(int32_t));
    ok = (fclose(f) == 0) && ok;
    return ok ? GS_OK : fail(GS_E_IO, std::string("short write to ") + path);
} catch (const std::bad_alloc &) {
    return fail(GS_E_NOMEM, "out of host memory");
} catch (const std::exception &e) {  // (nothing may leave through the C ABI)
    return fail(GS_E_INVALID, std::string("unexpected exception: ") + e.what());
}

// a store file into HBM: whole (stripes <= 1) or striped, as store_place
static int db_load_impl(gs_db **out, int device, const char *path, int stripes, const int *stripe_devices, int stripe_only) {
    if (!out || !path) return fail(GS_E_INVALID, "NULL argument");
    for (int p = 0; p < (stripes > 1 && stripe_only < 0 ? stripes : 1); p++) out[p] = nullptr;
    int rc = use_device(device);
    if (rc) return rc;
    FILE *f = fopen(path, "rb");
    if (!f) return fail(GS_E_INVALID, std::string("cannot open ") + path);
    GsStoreFileHeader h{};
    uint32_t file_mgate_ctx = 0;
    const bool got_header = fread(&h, sizeof(h), 1, f) == 1;
    if (got_header) {
        file_mgate_ctx = h.vbits >> 31;
        h.vbits &= 0x7fffffffu;
    }
    if (!got_header || memcmp(h.magic, "GSSTORE9", 8) != 0 || h.info.n_values < 1 || h.info.n_values > (1 << 24) ||
        h.bucket_bits > 29 || h.info.n_buckets != ((int64_t)1 << h.bucket_bits) || h.vbits > 25 ||
        h.gate_words > ((uint64_t)1 << 28) || h.mgate_words > ((uint64_t)1 << 30) || h.rec_buckets > ((uint64_t)1 << 29)) {
        fclose(f);
        return fail(GS_E_INVALID, std::string(path) + " is not a gsgpu store file (or one of another layout version)");
    }
    const size_t nv = (size_t)h.info.n_values;
    {   // the payload the header announces must be exactly what the file holds
        const uint64_t want = (uint64_t)sizeof(h) + (uint64_t)h.info.n_buckets * 64 + h.gate_words * 8 + h.mgate_words * 4 + h.rec_buckets * GS_REC_WORDS * 8 + (uint64_t)nv * 16;
        if (fseeko(f, 0, SEEK_END) != 0 || (uint64_t)ftello(f) != want || fseeko(f, (off_t)sizeof(h), SEEK_SET) != 0) {
            fclose(f);
            return fail(GS_E_INVALID, std::string(path) + ": file size does not match its header (truncated or damaged)");
        }
    }
    // (the big arrays as untouched pages: the reading threads are the first to touch them)
    Zeroed<u64> table, rec;
    table.reset((size_t)h.info.n_buckets * GS_SLOTS_PER_BUCKET);
    rec.reset((size_t)h.rec_buckets * GS_REC_WORDS);
    std::vector<u64> gate((size_t)h.gate_words);
    std::vector<uint32_t> mgate((size_t)h.mgate_words);
    std::vector<int32_t> tree(4 * nv);
    // the payload with several threads (pread): one thread copies a page-cached file at 2-3 GB/s, a big store has 10 GB
    const int fd = fileno(f);
    off_t at = (off_t)sizeof(h);
    auto par_read = [&](void *dst, size_t bytes) {
        const int n_thr = bytes < ((size_t)64 << 20) ? 1 : (int)std::min<unsigned>(16, std::max<unsigned>(1, std::thread::hardware_concurrency()));
        std::atomic<bool> good{true};
        auto part = [&](int t) {
            size_t lo = bytes * (size_t)t / (size_t)n_thr;
            const size_t hi = bytes * ((size_t)t + 1) / (size_t)n_thr;
            while (lo < hi) {
                const ssize_t r = pread(fd, (char *)dst + lo, std::min<size_t>(hi - lo, (size_t)1 << 30), at + (off_t)lo);
                if (r <= 0) {
                    good = false;
                    return;
                }
                lo += (size_t)r;
            }
        };
        if (n_thr == 1)
            part(0);
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < n_thr; t++) th.emplace_back(part, t);
            for (auto &x : th) x.join();
        }
        at += (off_t)bytes;
        return good.load();
    };
    const bool ok = par_read(table.data(), table.size() * sizeof(u64)) && par_read(gate.data(), gate.size() * sizeof(u64)) &&
                    par_read(mgate.data(), mgate.size() * sizeof(uint32_t)) && par_read(rec.data(), rec.size() * sizeof(u64)) &&
                    par_read(tree.data(), tree.size() * sizeof(int32_t));
    fclose(f);
    if (!ok) return fail(GS_E_INVALID, std::string(path) + " is truncated");
    {
        StoreChecksum cs;
        cs.add(table.data(), table.size() * sizeof(u64));
        cs.add(gate.data(), gate.size() * sizeof(u64));
        cs.add(mgate.data(), mgate.size() * sizeof(uint32_t));
        cs.add(rec.data(), rec.size() * sizeof(u64));
        cs.add(tree.data(), tree.size() * sizeof(int32_t));
        if (cs.value() != h.checksum) return fail(GS_E_INVALID, std::string(path) + ": payload checksum mismatch (damaged file)");
    }
    if (const char *why = store_image_defect(h, table, rec, tree)) return fail(GS_E_INVALID, std::string(path) + ": " + why);
    StoreImage im{};
    im.k = h.info.k;
    im.n_values = h.info.n_values;
    im.n_entries = h.info.n_entries;
    im.n_stored = h.info.n_stored;
    im.n_in_records = h.info.n_in_records;
    im.b = (int)h.bucket_bits;
    im.vbits = (int)h.vbits;
    im.rec_bits = 0;
    while (((uint64_t)1 << im.rec_bits) < h.rec_buckets) im.rec_bits++;
    im.max_disp = h.info.max_displacement;
    im.table = table.data();
    im.table_words = table.size();
    im.gate = gate.data();
    im.gate_words = gate.size();
    im.mgate = mgate.data();
    im.mgate_words = mgate.size();
    im.mgate_ctx = (file_mgate_ctx && h.info.k >= GS_CTX_MIN_K && !mgate.empty()) ? 1u : 0u;
    im.rec = rec.data();
    im.rec_words = rec.size();
    im.parent = tree.data();
    im.depth = tree.data() + nv;
    im.tin = tree.data() + 2 * nv;
    im.tout = tree.data() + 3 * nv;
    return store_place(im, device, stripes, stripe_devices, stripe_only, out);
}

#define GS_API_CATCH                                                                  \
    catch (const std::bad_alloc &) { return fail(GS_E_NOMEM, "out of host memory"); } \
    catch (const std::exception &e) { return fail(GS_E_INVALID, std::string("unexpected exception: ") + e.what()); }

extern "C" int gs_db_load(gs_db **out, int device, const char *path) try {
    return db_load_impl(out, device, path, 1, nullptr, -1);
}
GS_API_CATCH

extern "C" int gs_db_load_striped(gs_db **out, const int *devices, int n_stripes, const char *path) try {
    if (!out || !devices || n_stripes < 2 || n_stripes > GS_MAX_STRIPES) return fail(GS_E_INVALID, "a striped store spans 2..8 devices");
    for (int p = 0; p < n_stripes; p++) {
        out[p] = nullptr;
        const int rc = use_device(devices[p]);
        if (rc) return rc;
    }
    return db_load_impl(out, devices[0], path, n_stripes, devices, -1);
}
GS_API_CATCH

extern "C" int gs_db_load_stripe(gs_db **out, int device, int n_stripes, int stripe, const char *path) try {
    if (!out || n_stripes < 2 || n_stripes > GS_MAX_STRIPES || stripe < 0 || stripe >= n_stripes)
        return fail(GS_E_INVALID, "a striped store spans 2..8 devices");
    return db_load_impl(out, device, path, n_stripes, nullptr, stripe);
}
GS_API_CATCH

extern "C" int gs_db_destroy(gs_db *db) {
    if (!db) return GS_OK;
    if (db->live_runs > 0) {  // freed by the last gs_match_destroy
        db->destroy_pending = true;
        return GS_OK;
    }
    db_free(db);
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------------
// match runs
// ---------------------------------------------------------------------------------------------------
// device state of the text mode (raw FASTQ chunk -> records, gs_text.hip); one per run / per filter handle
struct TextScan {
    // the chunk's bytes on the device: two buffers, filled by turns on a stream of their own, so that the copy of chunk
    // i+1 runs while the kernels of chunk i work on the other buffer.  d_text = the buffer of the latest chunk.
    uint8_t *d_text = nullptr;
    uint8_t *d_buf[2] = {nullptr, nullptr};
    size_t buf_cap[2] = {0, 0};
    hipStream_t copy_stream = nullptr;
    hipEvent_t done[2] = {};       // the kernels that read d_buf[b] have been issued up to here (on the owner's stream)
    bool done_valid[2] = {false, false};
    uint32_t *d_tile = nullptr, *d_nl = nullptr;
    size_t tile_cap = 0, nl_cap = 0;
    u64 *d_off2 = nullptr;
    size_t off2_cap = 0;
    // FASTA mode: per-line scan words, per-block totals, per-line destinations, the gathered sequences
    u64 *d_fa_scan = nullptr, *d_fa_block = nullptr;
    uint32_t *d_line_dst = nullptr;
    uint8_t *d_fa_seq = nullptr;
    size_t fa_scan_cap = 0, fa_block_cap = 0, line_dst_cap = 0, fa_seq_cap = 0;
    bool last_fasta = false;
    // general FASTQ (gs_match_submit_fastq_ml): per-line record structure
    uint32_t *d_ml_next = nullptr, *d_ml_plus = nullptr, *d_ml_ja = nullptr, *d_ml_jb = nullptr;
    uint8_t *d_ml_mark = nullptr, *d_ml_class = nullptr;
    u64 *d_ml_out = nullptr;
    size_t ml_next_cap = 0, ml_plus_cap = 0, ml_ja_cap = 0, ml_jb_cap = 0, ml_mark_cap = 0, ml_class_cap = 0;
    // GS_TEXT_BANKS independent streams of chunks (files read side by side): a refusal in one must not silence the
    // others, so the status words and totals exist once per bank; `bank` is the one the next calls work on
    uint32_t *d_status = nullptr;  // GS_TEXT_BANKS x GS_TS_WORDS
    u64 *d_totals = nullptr;       // GS_TEXT_BANKS x ([3] chunk scratch | [3] totals of the accepted chunks)
    int bank = 0;
    int64_t last_reads = 0, last_lines = 0;  // of the most recent chunk (per-read follow-up calls refer to it)
    hipEvent_t copied[8] = {};     // H2D of ticket t has completed: copied[t % 8]
    int64_t tickets = 0;
    // the writers' gather on the device (gs_*_compact_text, gs_deflate_dev.hip): output buffers [which][slot], per-record lengths,
    // per-block sums, page-locked totals; d_last_flags = where the last chunk's per-read flags lie on the device
    uint8_t *d_compact[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    size_t compact_cap[2][2] = {{0, 0}, {0, 0}};
    uint32_t *d_clen = nullptr;
    u64 *d_cblocks = nullptr, *h_ctotals = nullptr;
    size_t clen_cap = 0, cblocks_cap = 0;
    const uint8_t *d_last_flags = nullptr;
    int64_t last_bytes = 0;
    bool last_four_line = false;
};
enum { GS_TEXT_BANKS = 16 };

// gs_match_submit_async: two sets of device staging buffers, filled by turns on a copy stream, so that the copy of
// batch i+1 runs while the kernel of batch i works on the other set
struct BatchStage {
    uint8_t *d_seq[2] = {nullptr, nullptr};
    uint64_t *d_off[2] = {nullptr, nullptr};
    int32_t *d_class[2] = {nullptr, nullptr};
    uint8_t *d_flags[2] = {nullptr, nullptr};
    size_t seq_cap[2] = {0, 0}, reads_cap[2] = {0, 0};
    std::vector<uint64_t> rel[2];  // rebased offsets of a batch whose offsets[0] != 0 (must outlive the copy)
    hipStream_t copy_stream = nullptr;
    hipEvent_t copied[2] = {}, finished[4] = {};  // finished[t % 4]: everything of batch t is through
    int64_t tickets = 0;
};

struct gs_run {
    BatchStage stage;
    gs_db *db = nullptr;
    gs_match_cfg cfg{};
    hipStream_t stream = nullptr;
    int64_t *d_sums = nullptr;   // [stat_copies][n_values][GS_N_SUMS]; copy 0 is the one everything outside the kernels sees
    u64 *h_result = nullptr;     // page-locked landing area of gs_match_finish: sums | max keys | unique counts | double sums
    int64_t *d_max = nullptr;
    double *d_dsums = nullptr;
    u64 *d_route_cursors = nullptr;    // gs_match_encode_route: slots handed out per owner + overflow flag
    GsStatRec *d_stat_recs = nullptr;  // deferred statistics of the current batch (global-atomic counters only)
    u64 *d_stat_rec_count = nullptr;
    size_t stat_recs_cap = 0;
    int32_t *d_stat_vi = nullptr;      // the records' value indices as an array (stores with more values than one reduce pass takes)
    size_t stat_vi_cap = 0;
    bool use_stat_recs = false;
    int stat_copies = 1;         // > 1: global-atomic counters spread over several copies (GsMatchParams::stat_copies)
    bool stats_spread = false;   // some copy other than 0 may be non-zero: fold_stats() before reading
    uint32_t *d_hit_counts = nullptr;  // per slot, only when cfg.max_kmer_res_counts > 0
    uint32_t *d_bitmap = nullptr;  // compact copy of the slots' seen bits (built on demand: finish / device_state)
    int64_t bitmap_words = 0;
    bool seen_dirty = false;    // some slot may carry a seen bit
    bool bitmap_merged = false; // d_bitmap holds a merged (multi-rank) bitmap: do not re-extract
    u64 *d_unique = nullptr;
    unsigned int *d_long_count = nullptr;
    uint32_t *d_long_list = nullptr;
    int64_t long_cap = 0;
    int32_t *d_scratch = nullptr;  // long-read kernel: per wave tag[n_values] | cnt[n_values]
    uint32_t *d_serial = nullptr;
    // reads of tens of thousands of positions and more, taken apart over many waves (gs_match_huge_kernel): one allocation
    // [count u32 x 4 | list | heads | chunk records | cnt rows | first rows | touch rows]
    unsigned char *d_huge = nullptr;
    int huge_slots = 0, huge_min = GS_HUGE_MIN, huge_chunk_min = GS_HUGE_CHUNK_MIN;
    int long_grid = 0;
    // host staging (GS_MEM_HOST)
    uint8_t *d_seq = nullptr;
    uint64_t *d_off = nullptr;
    int32_t *d_class = nullptr;
    uint8_t *d_flags = nullptr;
    size_t seq_cap = 0, reads_cap = 0;
    int grid = 0;
    // Kraken-style segments of the last gs_match_segments call
    uint32_t *d_seg_count = nullptr;
    u64 *d_seg_off = nullptr;
    int32_t *d_seg_code = nullptr, *d_seg_start = nullptr;
    int64_t seg_total = 0;
    TextScan text;  // text mode (gs_match_submit_text)
    // profiling
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    int64_t launches = 0;
    double total_ms = 0;
};


static void text_free_fasta(TextScan &t) {
    hipFree(t.d_ml_next);
    hipFree(t.d_ml_plus);
    hipFree(t.d_ml_ja);
    hipFree(t.d_ml_jb);
    hipFree(t.d_ml_mark);
    hipFree(t.d_ml_class);
    hipFree(t.d_ml_out);
    hipFree(t.d_fa_scan);
    hipFree(t.d_fa_block);
    hipFree(t.d_line_dst);
    hipFree(t.d_fa_seq);
}

static void text_free(TextScan &t) {
    text_free_fasta(t);
    if (t.copy_stream) {
        hipStreamSynchronize(t.copy_stream);
        hipStreamDestroy(t.copy_stream);
    }
    for (hipEvent_t ev : t.done)
        if (ev) hipEventDestroy(ev);
    hipFree(t.d_buf[0]);
    hipFree(t.d_buf[1]);
    hipFree(t.d_tile);
    hipFree(t.d_nl);
    hipFree(t.d_off2);
    hipFree(t.d_status);
    hipFree(t.d_totals);
    for (auto &w : t.d_compact)
        for (uint8_t *q : w) hipFree(q);
    hipFree(t.d_clen);
    hipFree(t.d_cblocks);
    if (t.h_ctotals) hipHostFree(t.h_ctotals);
    for (hipEvent_t ev : t.copied)
        if (ev) hipEventDestroy(ev);
    t = TextScan();
}

template <typename T>
static int grow(T **p, size_t *cap, size_t need, hipStream_t stream) {
    if (*cap >= need) return GS_OK;
    HIP_TRY(hipStreamSynchronize(stream));
    hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t n = need + need / 4;
    HIP_TRY(hipMalloc((void **)p, n * sizeof(T)));
    *cap = n;
    return GS_OK;
}

static int text_reset_bank(TextScan &t, int bank, bool totals, hipStream_t stream) {
    uint32_t *st = t.d_status + (size_t)bank * GS_TS_WORDS;
    HIP_TRY(hipMemsetAsync(st, 0, sizeof(uint32_t) * GS_TS_WORDS, stream));
    HIP_TRY(hipMemsetAsync(st + GS_TS_FIRST_BAD, 0xff, sizeof(uint32_t), stream));
    if (totals) HIP_TRY(hipMemsetAsync(t.d_totals + (size_t)bank * 6, 0, sizeof(u64) * 6, stream));
    return GS_OK;
}

// the selected bank, or (all = true: begin / reset of the handle) every bank
static int text_reset(TextScan &t, bool totals, hipStream_t stream, bool all = false) {
    if (!t.d_status) return GS_OK;
    if (!all) return text_reset_bank(t, t.bank, totals, stream);
    for (int b = 0; b < GS_TEXT_BANKS; b++) {
        int rc = text_reset_bank(t, b, totals, stream);
        if (rc) return rc;
    }
    return GS_OK;
}

extern "C" hipError_t gs_launch_text_scan(const GsTextParams *P, uint32_t ticket, hipStream_t stream);
extern "C" hipError_t gs_launch_text_ml(const GsTextParams *P, uint8_t *line_class, hipStream_t stream);

// copies the chunk to the device and runs the record scan; *ticket identifies the chunk.  After it the (start, end)
// pairs of the sequence lines are in t.d_off2, the newline offsets in t.d_nl and the skip flag in t.d_status.
// fasta_records < 0: four-line FASTQ; >= 0: FASTA with that many header lines (gs_text.hip)
// ml_out != nullptr: general FASTQ -- the record structure is found on the device first (this call then waits for it) and
// ml_out[0] = complete records (-1: the chunk was refused), ml_out[1] = bytes, ml_out[2] = lines they cover
static int text_submit(TextScan &t, hipStream_t stream, const uint8_t *text, int64_t n_bytes, int64_t n_lines, int mem, int k,
                       int64_t *ticket, int64_t fasta_records = -1, int64_t *ml_out = nullptr) {
    const bool ml = ml_out != nullptr;
    const bool fasta = fasta_records >= 0 || ml;
    if (ml) {
        ml_out[0] = ml_out[1] = ml_out[2] = 0;
        fasta_records = 0;
    }
    if (n_bytes < 0 || n_lines < 0 || (!fasta && (n_lines & 3) != 0) || (n_bytes > 0 && !text) || n_lines > n_bytes)
        return fail(GS_E_INVALID, "bad text chunk (n_lines must be a multiple of 4)");
    if (ml && n_lines >= ((int64_t)1 << 26)) return fail(GS_E_INVALID, "general FASTQ chunks are limited to 2^26 lines");
    if (fasta && (fasta_records > n_lines || fasta_records >= ((int64_t)1 << 24)))
        return fail(GS_E_INVALID, "bad FASTA chunk (at most 2^24 - 1 records, not more records than lines)");
    if (n_bytes > ((int64_t)1 << 30)) return fail(GS_E_INVALID, "text chunks are limited to 1 GiB");
    if (mem != GS_MEM_HOST && mem != GS_MEM_DEVICE) return fail(GS_E_INVALID, "mem must be GS_MEM_HOST or GS_MEM_DEVICE");
    if (!t.d_status) {
        HIP_TRY(hipMalloc((void **)&t.d_status, sizeof(uint32_t) * GS_TS_WORDS * GS_TEXT_BANKS));
        HIP_TRY(hipMalloc((void **)&t.d_totals, sizeof(u64) * 6 * GS_TEXT_BANKS));
        int rc = text_reset(t, true, stream, true);
        if (rc) return rc;
        for (hipEvent_t &ev : t.copied) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        for (hipEvent_t &ev : t.done) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIP_TRY(hipStreamCreateWithFlags(&t.copy_stream, hipStreamNonBlocking));
    }
    int64_t n_reads = ml ? n_lines / 4 + 1 : (fasta ? fasta_records : (n_lines >> 2));  // (ml: an upper bound for now)
    const size_t padded = ((size_t)n_bytes + 4095) & ~(size_t)4095;
    const int64_t tk = t.tickets;
    const int b = (int)(tk & 1);
    int rc;
    if (t.buf_cap[b] < padded + 4096) HIP_TRY(hipStreamSynchronize(t.copy_stream));  // (grow() waits for `stream` itself)
    if ((rc = grow(&t.d_buf[b], &t.buf_cap[b], padded + 4096, stream))) return rc;
    if ((rc = grow(&t.d_tile, &t.tile_cap, padded / 4096 + 1, stream))) return rc;
    if ((rc = grow(&t.d_nl, &t.nl_cap, (size_t)n_lines + 4, stream))) return rc;
    if ((rc = grow(&t.d_off2, &t.off2_cap, 2 * (size_t)n_reads + 2, stream))) return rc;
    if (fasta) {
        if ((rc = grow(&t.d_fa_scan, &t.fa_scan_cap, (size_t)n_lines + 1, stream))) return rc;
        if ((rc = grow(&t.d_fa_block, &t.fa_block_cap, (size_t)n_lines / GS_FA_BLOCK + 2, stream))) return rc;
        if ((rc = grow(&t.d_line_dst, &t.line_dst_cap, (size_t)n_lines + 1, stream))) return rc;
        if ((rc = grow(&t.d_fa_seq, &t.fa_seq_cap, (size_t)n_bytes + 256, stream))) return rc;
    }
    if (ml) {
        const size_t nl1 = (size_t)n_lines + 1;
        if ((rc = grow(&t.d_ml_next, &t.ml_next_cap, nl1, stream))) return rc;
        if ((rc = grow(&t.d_ml_plus, &t.ml_plus_cap, nl1, stream))) return rc;
        if ((rc = grow(&t.d_ml_ja, &t.ml_ja_cap, nl1, stream))) return rc;
        if ((rc = grow(&t.d_ml_jb, &t.ml_jb_cap, nl1, stream))) return rc;
        if ((rc = grow(&t.d_ml_mark, &t.ml_mark_cap, nl1, stream))) return rc;
        if ((rc = grow(&t.d_ml_class, &t.ml_class_cap, nl1, stream))) return rc;
        if (!t.d_ml_out) HIP_TRY(hipMalloc((void **)&t.d_ml_out, sizeof(u64) * 4));
    }
    t.d_text = t.d_buf[b];
    // the copy waits for the kernels of the chunk before last (they read this buffer), the scan for the copy
    if (t.done_valid[b]) HIP_TRY(hipStreamWaitEvent(t.copy_stream, t.done[b], 0));
    if (n_bytes > 0)
        HIP_TRY(hipMemcpyAsync(t.d_text, text, (size_t)n_bytes, mem == GS_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                               t.copy_stream));
    HIP_TRY(hipEventRecord(t.copied[tk & 7], t.copy_stream));
    HIP_TRY(hipStreamWaitEvent(stream, t.copied[tk & 7], 0));
    if (padded > (size_t)n_bytes) HIP_TRY(hipMemsetAsync(t.d_text + n_bytes, ' ', padded - (size_t)n_bytes, stream));
    GsTextParams T{};
    T.text = t.d_text;
    T.n_bytes = n_bytes;
    T.n_lines = n_lines;
    T.tile_count = t.d_tile;
    T.nl = t.d_nl;
    T.off2 = (unsigned long long *)t.d_off2;
    T.chunk_totals = (unsigned long long *)t.d_totals + (size_t)t.bank * 6;
    T.run_totals = T.chunk_totals + 3;
    T.status = t.d_status + (size_t)t.bank * GS_TS_WORDS;
    T.k = k;
    T.n_records = fasta ? fasta_records : -1;
    T.fa_scan = (unsigned long long *)t.d_fa_scan;
    T.fa_block = (unsigned long long *)t.d_fa_block;
    T.line_dst = t.d_line_dst;
    T.fa_seq = t.d_fa_seq;
    t.last_fasta = fasta;
    if (ml) {
        // first half: newlines + record structure; then this thread waits for the counts (the match launch needs the number
        // of reads, the caller the bytes that belong to the next chunk)
        T.ml_next = t.d_ml_next;
        T.ml_plus = t.d_ml_plus;
        T.ml_jump_a = t.d_ml_ja;
        T.ml_jump_b = t.d_ml_jb;
        T.ml_mark = t.d_ml_mark;
        T.ml_out = (unsigned long long *)t.d_ml_out;
        const u64 init[4] = {0, (u64)n_lines, 0, 0};
        HIP_TRY(hipMemcpyAsync(t.d_ml_out, init, sizeof(init), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemsetAsync(t.d_ml_class, 0, (size_t)n_lines + 1, stream));
        HIP_TRY(gs_launch_text_ml(&T, t.d_ml_class, stream));
        u64 got[4] = {0, 0, 0, 0};
        uint32_t st[GS_TS_WORDS];
        HIP_TRY(hipMemcpyAsync(got, t.d_ml_out, sizeof(got), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(st, T.status, sizeof(st), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        int64_t lines_done = 0, bytes_done = 0;
        n_reads = 0;
        if (st[GS_TS_CHUNK_ERR] == 0 && st[GS_TS_STICKY] == 0 && n_lines > 0) {
            n_reads = (int64_t)got[0];
            lines_done = (int64_t)got[1];
            if (lines_done > 0) {
                uint32_t last_nl = 0;
                HIP_TRY(hipMemcpy(&last_nl, t.d_nl + (lines_done - 1), sizeof(uint32_t), hipMemcpyDeviceToHost));
                bytes_done = (int64_t)last_nl + 1;
            }
        }
        ml_out[0] = (st[GS_TS_CHUNK_ERR] != 0 || st[GS_TS_STICKY] != 0) ? -1 : n_reads;
        ml_out[1] = bytes_done;
        ml_out[2] = lines_done;
        // second half on what is whole: the FASTA kernels under the line classes (or, after an error, only the commit)
        T.n_lines = lines_done;
        T.n_bytes = bytes_done > 0 ? bytes_done : n_bytes;
        T.n_records = n_reads;
        T.line_class = t.d_ml_class;
        n_lines = lines_done;
    }
    HIP_TRY(gs_launch_text_scan(&T, (uint32_t)tk, stream));
    HIP_TRY(hipEventRecord(t.done[b], stream));
    t.done_valid[b] = true;
    t.tickets = tk + 1;
    t.last_reads = n_reads;
    t.last_lines = n_lines;
    t.last_bytes = n_bytes;
    t.last_four_line = !fasta;
    t.d_last_flags = nullptr;
    if (ticket) *ticket = tk;
    return GS_OK;
}

extern "C" int gs_compact_records_device(hipStream_t stream, const uint8_t *d_text, const uint32_t *d_nl, int64_t n_records, const uint8_t *d_flags, int mask, int want,
                                         int with_probs, uint8_t *d_out, uint32_t *d_len, u64 *d_blocks, u64 *h_totals);
extern "C" const char *gs_deflate_last_error(void);

static int text_touched(TextScan &t, hipStream_t stream);

// the records of the last chunk whose flag says so, as ReadEntry.write writes them, into t.d_compact[which][slot]; synchronises
static int text_compact(TextScan &t, hipStream_t stream, int mask, int want, int which, int slot, int with_probs, const uint8_t **d_out, int64_t *n_bytes,
                        int64_t *n_records) {
    if (!d_out || !n_bytes || !n_records) return fail(GS_E_INVALID, "NULL argument");
    *d_out = nullptr;
    *n_bytes = *n_records = 0;
    if (which < 0 || which > 1 || slot < 0 || slot > 1) return fail(GS_E_INVALID, "which and slot must be 0 or 1");
    if (t.tickets == 0) return fail(GS_E_STATE, "no text chunk has been submitted");
    if (!t.last_four_line) return fail(GS_E_STATE, "the last chunk was not four-line FASTQ");
    if (!t.d_last_flags) return fail(GS_E_STATE, "the last chunk was submitted without per-read flags");
    const int64_t n = t.last_reads;
    if (n == 0) return GS_OK;
    int rc;
    if ((rc = grow(&t.d_compact[which][slot], &t.compact_cap[which][slot], (size_t)t.last_bytes + 64, stream))) return rc;
    if ((rc = grow(&t.d_clen, &t.clen_cap, (size_t)n, stream))) return rc;
    if ((rc = grow(&t.d_cblocks, &t.cblocks_cap, 2 * ((size_t)n / 256 + 1) + 2, stream))) return rc;
    if (!t.h_ctotals) HIP_TRY(hipHostMalloc((void **)&t.h_ctotals, 2 * sizeof(u64)));
    if (gs_compact_records_device(stream, t.d_text, t.d_nl, n, t.d_last_flags, mask, want, with_probs, t.d_compact[which][slot], t.d_clen, t.d_cblocks, t.h_ctotals) != GS_OK)
        return fail(GS_E_HIP, std::string("record gather: ") + gs_deflate_last_error());
    if ((rc = text_touched(t, stream))) return rc;
    HIP_TRY(hipStreamSynchronize(stream));
    *d_out = t.d_compact[which][slot];
    *n_bytes = (int64_t)t.h_ctotals[0];
    *n_records = (int64_t)t.h_ctotals[1];
    return GS_OK;
}

// to be called after every launch on `stream` that reads the latest chunk's bytes (match / filter / segment kernels):
// the buffer may be overwritten by the chunk after next only when these are through
static int text_touched(TextScan &t, hipStream_t stream) {
    if (t.tickets == 0) return GS_OK;
    HIP_TRY(hipEventRecord(t.done[(t.tickets - 1) & 1], stream));
    return GS_OK;
}

static int text_wait_copy(TextScan &t, int64_t ticket) {
    if (ticket < 0 || ticket >= t.tickets) return fail(GS_E_INVALID, "unknown ticket");
    if (ticket + 8 <= t.tickets) return GS_OK;  // its event has been re-recorded by a later submit: long done
    HIP_TRY(hipEventSynchronize(t.copied[ticket & 7]));
    return GS_OK;
}

// synchronises the stream
static int text_status(TextScan &t, hipStream_t stream, int64_t *failed_ticket, int64_t *first_bad_record, int64_t totals[3]) {
    if (failed_ticket) *failed_ticket = -1;
    if (first_bad_record) *first_bad_record = -1;
    if (totals) totals[0] = totals[1] = totals[2] = 0;
    if (!t.d_status) return GS_OK;
    uint32_t st[GS_TS_WORDS];
    u64 tt[6];
    HIP_TRY(hipMemcpyAsync(st, t.d_status + (size_t)t.bank * GS_TS_WORDS, sizeof(st), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(tt, t.d_totals + (size_t)t.bank * 6, sizeof(tt), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (st[GS_TS_STICKY]) {
        if (failed_ticket) *failed_ticket = (int64_t)st[GS_TS_FAILED_TICKET];
        if (first_bad_record && st[GS_TS_FIRST_BAD] != 0xffffffffu) *first_bad_record = (int64_t)st[GS_TS_FIRST_BAD];
    }
    if (totals)
        for (int i = 0; i < 3; i++) totals[i] = (int64_t)tt[3 + i];
    return GS_OK;
}

static int run_clear(gs_run *run) {
    const size_t nv = (size_t)run->db->info.n_values;
    const size_t cp = (size_t)run->stat_copies;
    // (sums, max keys, double sums and the long-read queue counters [queue length, consumer cursor] are one block)
    HIP_TRY(hipMemsetAsync(run->d_sums, 0, sizeof(int64_t) * nv * (GS_N_SUMS + 1 + GS_N_DCOLS) * cp + 2 * sizeof(unsigned int), run->stream));
    run->stats_spread = false;
    HIP_TRY(hipMemsetAsync(run->d_bitmap, 0, sizeof(uint32_t) * (size_t)run->bitmap_words, run->stream));
    if (run->d_hit_counts)
        HIP_TRY(hipMemsetAsync(run->d_hit_counts, 0, sizeof(uint32_t) * (size_t)(run->db->n_slots() + run->db->n_rec * GS_REC_SLOTS), run->stream));
    if (run->seen_dirty && !run->db->striped())  // (a striped store's seen bits are the bitmap that was just cleared)
        HIP_TRY(gs_launch_clear_seen(run->db->d_table, run->db->n_slots(), run->db->d_rec, run->db->n_rec, run->stream));
    run->seen_dirty = false;
    run->bitmap_merged = false;
    return GS_OK;
}

static void run_free(gs_run *run) {
    if (!run) return;
    hipSetDevice(run->db->device);
    for (auto &p : run->pending) {
        hipEventDestroy(p.first);
        hipEventDestroy(p.second);
    }
    hipHostFree(run->h_result);
    hipFree(run->d_sums);  // (d_max, d_dsums, d_long_count lie inside)
    hipFree(run->d_stat_recs);
    hipFree(run->d_stat_vi);
    hipFree(run->d_stat_rec_count);
    hipFree(run->d_route_cursors);
    hipFree(run->d_bitmap);
    hipFree(run->d_hit_counts);
    hipFree(run->d_unique);
    hipFree(run->d_long_list);
    hipFree(run->d_scratch);
    hipFree(run->d_serial);
    hipFree(run->d_huge);
    hipFree(run->d_seq);
    hipFree(run->d_off);
    hipFree(run->d_class);
    hipFree(run->d_flags);
    hipFree(run->d_seg_count);
    hipFree(run->d_seg_off);
    hipFree(run->d_seg_code);
    hipFree(run->d_seg_start);
    text_free(run->text);
    {
        BatchStage &g = run->stage;
        if (g.copy_stream) {
            hipStreamSynchronize(g.copy_stream);
            hipStreamDestroy(g.copy_stream);
        }
        for (int b = 0; b < 2; b++) {
            hipFree(g.d_seq[b]);
            hipFree(g.d_off[b]);
            hipFree(g.d_class[b]);
            hipFree(g.d_flags[b]);
            if (g.copied[b]) hipEventDestroy(g.copied[b]);
        }
        for (hipEvent_t ev : g.finished)
            if (ev) hipEventDestroy(ev);
    }
    if (run->stream) hipStreamDestroy(run->stream);
    delete run;
}

extern "C" int gs_match_begin(gs_run **out, gs_db *db, const gs_match_cfg *cfg) {
    if (!out || !db || !cfg) return fail(GS_E_INVALID, "NULL argument");
    *out = nullptr;
    if (cfg->max_paths < 1 || cfg->max_paths > 128) return fail(GS_E_INVALID, "max_paths must be in [1,128] (C/GSConfigKey.java:350)");
    HIP_TRY(hipSetDevice(db->device));
    if (!db->complete()) return fail(GS_E_STATE, "striped store: not every stripe is attached yet (gs_db_stripe_attach)");
    if (cfg->count_unique && db->unique_owner && !db->striped())
        return fail(GS_E_STATE, "this store already has an active unique-counting run (the seen bits live in the table)");
    gs_run *run = new gs_run();
    run->db = db;
    run->cfg = *cfg;
    const size_t nv = (size_t)db->info.n_values;
    run->bitmap_words = (db->n_slots() + 31) / 32 + db->n_rec;  // one word per record bucket behind the table slots' bits
    hipError_t e = hipStreamCreateWithFlags(&run->stream, hipStreamNonBlocking);
    // Counters that do not fit the LDS are global atomics: 12 bytes x 8 per tax id in three cache lines that every CU
    // hits.  Measured on the 47 M-k-mer / 526-value store: they cost 8 of 16.7 ms on reads from the store; spread over
    // 16 copies (one per group of workgroups) the lines are 16 times colder.
    const char *force_global = getenv("GS_FORCE_GLOBAL_STATS");  // (developer knob, see gs_launch_match)
    if (nv > GS_NV_LDS || (force_global != nullptr && atoi(force_global) != 0)) {
        int copies = 16;
        if (const char *ev = getenv("GS_STAT_COPIES")) copies = std::max(1, std::min(64, atoi(ev)));
        while (copies > 1 && (size_t)copies * nv * 96 > ((size_t)64 << 20)) copies /= 2;
        run->stat_copies = copies;
        // and most reads need no atomics at all: the hit k-mers of a read usually carry one tax id, whose statistics go
        // into one 64-byte record per read that a second kernel adds up in LDS (gs_stat_reduce_kernel)
        run->use_stat_recs = nv <= GS_STAT_REC_MAX_VALUES;
        if (const char *ev = getenv("GS_STAT_RECS")) run->use_stat_recs = run->use_stat_recs && atoi(ev) != 0;
    }
    else {
        // counters in LDS: every workgroup adds its table to the device's when it ends -- 2 048 workgroups x the same few cache lines,
        // and atomics on one line execute one after the other (~11 ns each: a tail of 0.1-0.2 ms behind the kernel).  16 copies here too.
        run->stat_copies = 16;
        if (const char *ev = getenv("GS_STAT_COPIES")) run->stat_copies = std::max(1, std::min(64, atoi(ev)));
    }
    const size_t cp = (size_t)run->stat_copies;
    // sums | max keys | double sums | long-read queue counters in ONE allocation: one memset clears them (gs_match_reset)
    if (e == hipSuccess) e = hipMalloc((void **)&run->d_sums, sizeof(int64_t) * nv * (GS_N_SUMS + 1 + GS_N_DCOLS) * cp + 8 * sizeof(unsigned int));
    if (e == hipSuccess) {
        run->d_max = run->d_sums + nv * GS_N_SUMS * cp;
        run->d_dsums = reinterpret_cast<double *>(run->d_max + nv * cp);
        run->d_long_count = reinterpret_cast<unsigned int *>(run->d_dsums + nv * GS_N_DCOLS * cp);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&run->d_bitmap, sizeof(uint32_t) * (size_t)run->bitmap_words);
    if (e == hipSuccess) e = hipMalloc((void **)&run->d_unique, sizeof(u64) * nv);
    if (e == hipSuccess && cfg->max_kmer_res_counts > 0)
        e = hipMalloc((void **)&run->d_hit_counts, sizeof(uint32_t) * (size_t)(db->n_slots() + db->n_rec * GS_REC_SLOTS));
    if (e != hipSuccess) {
        run_free(run);
        return fail(e == hipErrorOutOfMemory ? GS_E_NOMEM : GS_E_HIP, std::string("run alloc: ") + hipGetErrorString(e));
    }
    int occ = gs_match_occupancy(db->info.n_values);
    if (occ < 1) occ = 1;
    if (occ > 8) occ = 8;  // 8 workgroups (32 waves) per CU: the kernel is built for 64 VGPRs
    if (const char *ev = getenv("GS_MATCH_BLOCKS_PER_CU")) {
        int v = atoi(ev);
        if (v >= 1 && v <= 16) occ = v;
    }
    run->grid = db->n_cu * occ;
    run->seen_dirty = cfg->count_unique != 0;  // a previous owner may have left seen bits behind
    int rc = run_clear(run);
    if (rc) {
        run_free(run);
        return rc;
    }
    if (cfg->count_unique && !db->striped()) db->unique_owner = run;  // (a striped store is read-only: any number of runs)
    db->live_runs++;
    *out = run;
    return GS_OK;
}

static int collect_events(gs_run *run) {
    for (auto &p : run->pending) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, p.first, p.second));
        run->total_ms += ms;
        run->launches++;
        hipEventDestroy(p.first);
        hipEventDestroy(p.second);
    }
    run->pending.clear();
    return GS_OK;
}

static int ensure_long(gs_run *run, int64_t n_reads) {
    // (the queue is filled in chunks of 64 entries per wave of the match kernel: every wave may leave one partly used)
    const int64_t room = n_reads + (int64_t)run->grid * (GS_BLOCK / 64) * 64;
    if (run->long_cap < room) {  // (three queues: long reads, reads for the three- and the four-sub-round kernel)
        HIP_TRY(hipStreamSynchronize(run->stream));
        hipFree(run->d_long_list);
        run->d_long_list = nullptr;
        run->long_cap = 0;
        int64_t cap = std::max<int64_t>(room, 1024);
        HIP_TRY(hipMalloc((void **)&run->d_long_list, sizeof(uint32_t) * (size_t)cap * 3));
        run->long_cap = cap;
    }
    if (!run->d_scratch) {
        // (where the longest reads leave the one-wave path, gs_match_huge_kernel: tunables and test hooks)
        if (const char *e = getenv("GS_HUGE_MIN")) run->huge_min = std::max(129, atoi(e));
        if (const char *e = getenv("GS_HUGE_CHUNK")) run->huge_chunk_min = std::max(128, (atoi(e) + 127) & ~127);
        // long-read kernel: a persistent grid that fills the device (the widest variant's occupancy; one wave per SIMD ran
        // 7-16x below the short-read kernel's rate per base), each wave owns tag[n_values] + cnt[n_values] in HBM
        const size_t nv = (size_t)run->db->info.n_values;
        int occ = gs_match_long_occupancy((int)nv);
        if (occ < 1) occ = 1;
        if (occ > 8) occ = 8;
        if (const char *e = getenv("GS_LONG_BLOCKS_PER_CU")) {
            const int v = atoi(e);
            if (v >= 1 && v <= 8) occ = v;
        }
        int grid = run->db->n_cu * occ;
        while (grid > 8 && (size_t)grid * 4 * nv * 2 * sizeof(int32_t) > ((size_t)2 << 30)) grid /= 2;
        run->long_grid = grid;
        const size_t waves = (size_t)grid * (GS_BLOCK / 64);
        HIP_TRY(hipMalloc((void **)&run->d_scratch, waves * nv * 2 * sizeof(int32_t)));
        HIP_TRY(hipMalloc((void **)&run->d_serial, waves * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(run->d_scratch, 0, waves * nv * 2 * sizeof(int32_t), run->stream));
        HIP_TRY(hipMemsetAsync(run->d_serial, 0, waves * sizeof(uint32_t), run->stream));
        if (const char *e = getenv("GS_TEST_LONG_SERIAL")) {
            // test hook for the serial wrap of gs_match_long_kernel: start every wave's serial just below 2^32 and fill
            // its tag / count rows with the values the serials take right after the wrap (1, 2, ..): without the clear
            // on wrap the first reads after it would take stale tags for their own
            const unsigned long v = strtoul(e, nullptr, 0);
            HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)run->d_serial, (int)(uint32_t)v, waves, run->stream));
            HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)run->d_scratch, 1, waves * nv * 2, run->stream));
        }
    }
    return GS_OK;
}

// the rows of the huge-read kernels; they leave them clean (gs_match_huge_finish_kernel), so this runs once per run
static int ensure_huge(gs_run *run, GsMatchParams *P) {
    const size_t nv = (size_t)run->db->info.n_values;
    const size_t per_slot_words = (3 * (size_t)GS_HUGE_COPIES + 2) * nv;  // votes, first positions, touch list per copy + the folded pair
    if (!run->d_huge) {
        int slots = (int)std::min<size_t>(GS_HUGE_SLOTS, std::max<size_t>(4, ((size_t)256 << 20) / (per_slot_words * 4)));
        if (const char *e = getenv("GS_HUGE_SLOTS")) slots = std::max(1, std::min(GS_HUGE_SLOTS, atoi(e)));
        const size_t fixed = 16 + sizeof(uint32_t) * GS_HUGE_SLOTS + sizeof(GsHugeHead) * GS_HUGE_SLOTS * GS_HUGE_COPIES;
        const size_t chunks = sizeof(GsHugeChunk) * (size_t)slots * GS_HUGE_MAX_CHUNKS, row = sizeof(uint32_t) * (size_t)slots * GS_HUGE_COPIES * nv;
        const size_t fold = sizeof(uint32_t) * (size_t)slots * 2 * nv;
        HIP_TRY(hipMalloc((void **)&run->d_huge, fixed + chunks + 3 * row + fold));
        HIP_TRY(hipMemsetAsync(run->d_huge, 0, fixed + chunks + row, run->stream));
        HIP_TRY(hipMemsetAsync(run->d_huge + fixed + chunks + row, 0xff, row, run->stream));  // first positions: none
        HIP_TRY(hipMemsetAsync(run->d_huge + fixed + chunks + 3 * row, 0, fold, run->stream));
        run->huge_slots = slots;
    }
    const size_t slots = (size_t)run->huge_slots;
    unsigned char *p = run->d_huge;
    P->huge_count = reinterpret_cast<unsigned int *>(p);
    p += 16;
    P->huge_list = reinterpret_cast<uint32_t *>(p);
    p += sizeof(uint32_t) * GS_HUGE_SLOTS;
    P->huge_head = reinterpret_cast<GsHugeHead *>(p);
    p += sizeof(GsHugeHead) * GS_HUGE_SLOTS * GS_HUGE_COPIES;
    P->huge_chunks = reinterpret_cast<GsHugeChunk *>(p);
    p += sizeof(GsHugeChunk) * slots * GS_HUGE_MAX_CHUNKS;
    P->huge_cnt = reinterpret_cast<uint32_t *>(p);
    p += sizeof(uint32_t) * slots * GS_HUGE_COPIES * nv;
    P->huge_first = reinterpret_cast<uint32_t *>(p);
    p += sizeof(uint32_t) * slots * GS_HUGE_COPIES * nv;
    P->huge_touch = reinterpret_cast<uint32_t *>(p);
    p += sizeof(uint32_t) * slots * GS_HUGE_COPIES * nv;
    P->huge_fold = reinterpret_cast<uint32_t *>(p);
    P->huge_slots = run->huge_slots;
    P->huge_min = run->huge_min;
    P->huge_chunk_min = run->huge_chunk_min;
    return hipMemsetAsync(P->huge_count, 0, sizeof(unsigned int), run->stream) == hipSuccess ? GS_OK : fail(GS_E_HIP, "huge-read counter");
}

static int launch_batch(gs_run *run, const uint8_t *d_seq, const uint64_t *d_off, int64_t n_reads, int64_t first_read_no,
                        int32_t *d_class, uint8_t *d_flags, const int32_t *d_nodes = nullptr,
                        const uint64_t *d_pos_off = nullptr, int off_stride = 1, const uint32_t *d_skip = nullptr, int fixed_len = 0) {
    if (n_reads > (int64_t)0xffffffffLL) return fail(GS_E_INVALID, "more than 2^32-1 reads in one batch");
    int rc = ensure_long(run, n_reads);
    if (rc) return rc;
    GsMatchParams P{};
    P.db = run->db->dev;
    P.seq = d_seq;
    P.off = d_off;
    P.n_reads = n_reads;
    P.first_read_no = first_read_no;
    P.classify = run->cfg.classify;
    P.count_unique = run->cfg.count_unique;
    P.max_paths = run->cfg.max_paths;
    P.threshold = run->cfg.threshold;
    P.max_read_tax_err = run->cfg.max_read_tax_err;
    P.max_read_class_err = run->cfg.max_read_class_err;
    P.sums = run->d_sums;
    P.max_keys = run->d_max;
    P.dsums = run->d_dsums;
    P.stat_copies = run->stat_copies;
    run->stats_spread = run->stat_copies > 1;
    P.bitmap = run->d_bitmap;
    P.hit_counts = run->d_hit_counts;
    P.class_vi = d_class;
    P.flags = d_flags;
    P.long_count = run->d_long_count;
    P.long_list = run->d_long_list;
    P.long_cap = run->long_cap;
    P.nodes = d_nodes;
    P.pos_off = (const unsigned long long *)d_pos_off;
    P.off_stride = off_stride;
    P.fixed_len = fixed_len;
    P.skip = d_skip;
    int grid = (int)std::min<int64_t>(run->grid, (n_reads + (GS_BLOCK / 64) - 1) / (GS_BLOCK / 64));
    if (grid < 1) grid = 1;
    const int64_t rec_room = n_reads + (int64_t)grid * (GS_BLOCK / 64) * 64;  // every wave may leave one chunk of 64 partly used
    if (run->use_stat_recs) {
        if ((rc = grow(&run->d_stat_recs, &run->stat_recs_cap, (size_t)rec_room, run->stream))) return rc;
        /* (640 = GS_REDUCE_VALUES, the values one pass of gs_stat_reduce_kernel takes) */
        if (run->db->info.n_values > 640 && (rc = grow(&run->d_stat_vi, &run->stat_vi_cap, (size_t)rec_room, run->stream))) return rc;
        if (!run->d_stat_rec_count) HIP_TRY(hipMalloc((void **)&run->d_stat_rec_count, sizeof(u64)));
        HIP_TRY(hipMemsetAsync(run->d_stat_rec_count, 0, sizeof(u64), run->stream));
        P.stat_recs = run->d_stat_recs;
        P.stat_rec_count = (unsigned long long *)run->d_stat_rec_count;
    }
    HIP_TRY(hipMemsetAsync(run->d_long_count, 0, 6 * sizeof(unsigned int), run->stream));
    // (reads of one short length each, or nodes that came from other ranks: nothing for the huge-read kernels)
    const bool huge = !d_nodes && (off_stride != 0 || fixed_len - run->db->info.k + 1 >= run->huge_min);
    if (huge && (rc = ensure_huge(run, &P))) return rc;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (run->cfg.profile) {
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, run->stream));
    }
    if (run->cfg.count_unique && !d_nodes) {
        run->seen_dirty = true;
        run->bitmap_merged = false;
    }
    // reads of one length with more than 128 k-mer positions (250-bp pairs, 150 bp at k < 23): every one of them would be queued for the
    // long-read kernel -- the pass that queues them is skipped (2.0 of 16.2 ms for 10 M reads of 150 bp at k = 16), the queue is the batch
    const bool all_long = off_stride == 0 && !d_nodes && !d_skip && fixed_len - run->db->info.k + 1 > 128 && !huge;
    static const bool wide_on = !getenv("GS_WIDE") || atoi(getenv("GS_WIDE")) != 0;  // (GS_WIDE=0: every read above 128 positions on the long-read path)
    P.wide_mask = wide_on ? gs_match_wide_mask(&P) : 0;
    if (all_long)
        P.long_list = nullptr;
    else {
        // reads of more than 128 positions into the queues of the kernels that take them (reads of one short length: there are none)
        if (off_stride != 0 || fixed_len - run->db->info.k + 1 > 128) HIP_TRY(gs_launch_classify(&P, run->stream));
        HIP_TRY(gs_launch_match(&P, grid, run->stream));
    }
    if (P.stat_recs && !all_long)  // (into copy 0 of the counters; part of the timed region)
        HIP_TRY(gs_launch_stat_reduce(P.stat_recs, run->d_stat_rec_count, rec_room, run->db->info.n_values, run->d_sums, run->d_max,
                                      run->d_dsums, run->d_stat_vi, run->stream));
    if (run->cfg.profile) {
        HIP_TRY(hipEventRecord(e1, run->stream));
        run->pending.push_back({e0, e1});
    }
    const int pos_fixed = fixed_len - run->db->info.k + 1;
    if (all_long && pos_fixed <= 256 && ((P.wide_mask >> (pos_fixed <= 192 ? 0 : 1)) & 1)) {  // the whole batch in trips of three / four sub-rounds
        HIP_TRY(gs_launch_match_wide(&P, pos_fixed <= 192 ? 3 : 4, run->db->n_cu, run->stream));
        return GS_OK;
    }
    if (!all_long && P.wide_mask) {  // what gs_match_kernel put into queues 1 and 2 (the kernels return at once when their queue is empty)
        if (P.wide_mask & 1) HIP_TRY(gs_launch_match_wide(&P, 3, run->db->n_cu, run->stream));
        if (P.wide_mask & 2) HIP_TRY(gs_launch_match_wide(&P, 4, run->db->n_cu, run->stream));
    }
    // reads with more than 128 k-mer positions were queued; the long-read kernel drains the queue
    HIP_TRY(gs_launch_match_long(&P, run->long_grid, run->d_scratch, run->d_serial, run->stream));
    if (huge) HIP_TRY(gs_launch_match_huge(&P, run->long_grid, run->stream));  // ... and hands the longest ones on
    return GS_OK;
}

extern "C" int gs_match_submit(gs_run *run, const uint8_t *seq, const uint64_t *offsets, int64_t n_reads,
                               int64_t first_read_no, int mem, int32_t *class_vi, uint8_t *flags) {
    if (!run) return fail(GS_E_INVALID, "run is NULL");
    if (n_reads < 0 || (n_reads > 0 && (!seq || !offsets))) return fail(GS_E_INVALID, "bad batch arrays");
    if (n_reads == 0) return GS_OK;
    HIP_TRY(hipSetDevice(run->db->device));
    if (mem == GS_MEM_DEVICE) return launch_batch(run, seq, offsets, n_reads, first_read_no, class_vi, flags);
    if (mem != GS_MEM_HOST) return fail(GS_E_INVALID, "mem must be GS_MEM_HOST or GS_MEM_DEVICE");
    // host batch: stage to HBM (synchronous)
    const size_t nbytes = (size_t)(offsets[n_reads] - offsets[0]);
    if (run->seq_cap < nbytes + 1) {
        HIP_TRY(hipStreamSynchronize(run->stream));
        hipFree(run->d_seq);
        run->d_seq = nullptr;
        run->seq_cap = 0;
        HIP_TRY(hipMalloc((void **)&run->d_seq, nbytes + 1));
        run->seq_cap = nbytes + 1;
    }
    if (run->reads_cap < (size_t)n_reads) {
        HIP_TRY(hipStreamSynchronize(run->stream));
        hipFree(run->d_off);
        hipFree(run->d_class);
        hipFree(run->d_flags);
        run->d_off = nullptr;
        run->d_class = nullptr;
        run->d_flags = nullptr;
        run->reads_cap = 0;
        HIP_TRY(hipMalloc((void **)&run->d_off, sizeof(uint64_t) * ((size_t)n_reads + 1)));
        HIP_TRY(hipMalloc((void **)&run->d_class, sizeof(int32_t) * (size_t)n_reads));
        HIP_TRY(hipMalloc((void **)&run->d_flags, (size_t)n_reads));
        run->reads_cap = (size_t)n_reads;
    }
    // offsets are rebased so that the staged slice starts at 0
    std::vector<uint64_t> rel;
    const uint64_t *hoff = offsets;
    if (offsets[0] != 0) {
        rel.resize((size_t)n_reads + 1);
        for (int64_t i = 0; i <= n_reads; i++) rel[(size_t)i] = offsets[i] - offsets[0];
        hoff = rel.data();
    }
    HIP_TRY(hipMemcpyAsync(run->d_seq, seq + offsets[0], nbytes, hipMemcpyHostToDevice, run->stream));
    HIP_TRY(hipMemcpyAsync(run->d_off, hoff, sizeof(uint64_t) * ((size_t)n_reads + 1), hipMemcpyHostToDevice, run->stream));
    int rc = launch_batch(run, run->d_seq, run->d_off, n_reads, first_read_no, class_vi ? run->d_class : nullptr,
                          flags ? run->d_flags : nullptr);
    if (rc) return rc;
    if (class_vi)
        HIP_TRY(hipMemcpyAsync(class_vi, run->d_class, sizeof(int32_t) * (size_t)n_reads, hipMemcpyDeviceToHost, run->stream));
    if (flags) HIP_TRY(hipMemcpyAsync(flags, run->d_flags, (size_t)n_reads, hipMemcpyDeviceToHost, run->stream));
    HIP_TRY(hipStreamSynchronize(run->stream));
    return collect_events(run);
}

// reads of ONE length, back to back, without an offsets array (what a sequencer's FASTQ gives once parsed: `read_len` bases each)
extern "C" int gs_match_submit_fixed(gs_run *run, const uint8_t *seq, int32_t read_len, int64_t n_reads, int64_t first_read_no, int mem,
                                     int32_t *class_vi, uint8_t *flags) {
    if (!run) return fail(GS_E_INVALID, "run is NULL");
    if (n_reads < 0 || read_len < 0 || (n_reads > 0 && !seq)) return fail(GS_E_INVALID, "bad batch arrays");
    if (n_reads == 0) return GS_OK;
    HIP_TRY(hipSetDevice(run->db->device));
    if (mem == GS_MEM_DEVICE)
        return launch_batch(run, seq, nullptr, n_reads, first_read_no, class_vi, flags, nullptr, nullptr, 0, nullptr, read_len);
    if (mem != GS_MEM_HOST) return fail(GS_E_INVALID, "mem must be GS_MEM_HOST or GS_MEM_DEVICE");
    const size_t nbytes = (size_t)n_reads * (size_t)read_len;
    if (run->seq_cap < nbytes + 1) {
        HIP_TRY(hipStreamSynchronize(run->stream));
        hipFree(run->d_seq);
        run->d_seq = nullptr;
        run->seq_cap = 0;
        HIP_TRY(hipMalloc((void **)&run->d_seq, nbytes + 1));
        run->seq_cap = nbytes + 1;
    }
    if (run->reads_cap < (size_t)n_reads) {
        HIP_TRY(hipStreamSynchronize(run->stream));
        hipFree(run->d_off);
        hipFree(run->d_class);
        hipFree(run->d_flags);
        run->d_off = nullptr;
        run->d_class = nullptr;
        run->d_flags = nullptr;
        run->reads_cap = 0;
        HIP_TRY(hipMalloc((void **)&run->d_off, sizeof(uint64_t) * ((size_t)n_reads + 1)));
        HIP_TRY(hipMalloc((void **)&run->d_class, sizeof(int32_t) * (size_t)n_reads));
        HIP_TRY(hipMalloc((void **)&run->d_flags, (size_t)n_reads));
        run->reads_cap = (size_t)n_reads;
    }
    HIP_TRY(hipMemcpyAsync(run->d_seq, seq, nbytes, hipMemcpyHostToDevice, run->stream));
    int rc = launch_batch(run, run->d_seq, nullptr, n_reads, first_read_no, class_vi ? run->d_class : nullptr, flags ? run->d_flags : nullptr, nullptr, nullptr,
                          0, nullptr, read_len);
    if (rc) return rc;
    if (class_vi) HIP_TRY(hipMemcpyAsync(class_vi, run->d_class, sizeof(int32_t) * (size_t)n_reads, hipMemcpyDeviceToHost, run->stream));
    if (flags) HIP_TRY(hipMemcpyAsync(flags, run->d_flags, (size_t)n_reads, hipMemcpyDeviceToHost, run->stream));
    HIP_TRY(hipStreamSynchronize(run->stream));
    return collect_events(run);
}

// the asynchronous form of a host batch: returns when the work is queued; gs_match_wait(ticket) returns when the
// batch's per-read outputs are in place and its input arrays may be reused.  Two batches can be under way.
extern "C" int gs_match_submit_async(gs_run *run, const uint8_t *seq, const uint64_t *offsets, int64_t n_reads,
                                     int64_t first_read_no, int32_t *class_vi, uint8_t *flags, int64_t *ticket) try {
    if (!run || !ticket) return fail(GS_E_INVALID, "NULL argument");
    if (n_reads <= 0 || !seq || !offsets) return fail(GS_E_INVALID, "bad batch arrays");
    HIP_TRY(hipSetDevice(run->db->device));
    BatchStage &g = run->stage;
    if (!g.copy_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&g.copy_stream, hipStreamNonBlocking));
        for (hipEvent_t &ev : g.copied) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        for (hipEvent_t &ev : g.finished) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    const int64_t tk = g.tickets;
    const int b = (int)(tk & 1);
    const size_t nbytes = (size_t)(offsets[n_reads] - offsets[0]);
    if (tk >= 2) HIP_TRY(hipEventSynchronize(g.copied[b]));  // rel[b] of the batch before last is free (long done)
    if (g.seq_cap[b] < nbytes + 1 || g.reads_cap[b] < (size_t)n_reads) {
        HIP_TRY(hipStreamSynchronize(g.copy_stream));
        HIP_TRY(hipStreamSynchronize(run->stream));
        if (g.seq_cap[b] < nbytes + 1) {
            hipFree(g.d_seq[b]);
            g.d_seq[b] = nullptr;
            g.seq_cap[b] = 0;
            HIP_TRY(hipMalloc((void **)&g.d_seq[b], nbytes + nbytes / 4 + 1));
            g.seq_cap[b] = nbytes + nbytes / 4 + 1;
        }
        if (g.reads_cap[b] < (size_t)n_reads) {
            hipFree(g.d_off[b]);
            hipFree(g.d_class[b]);
            hipFree(g.d_flags[b]);
            g.d_off[b] = nullptr;
            g.d_class[b] = nullptr;
            g.d_flags[b] = nullptr;
            g.reads_cap[b] = 0;
            const size_t cap = (size_t)n_reads + (size_t)n_reads / 4;
            HIP_TRY(hipMalloc((void **)&g.d_off[b], sizeof(uint64_t) * (cap + 1)));
            HIP_TRY(hipMalloc((void **)&g.d_class[b], sizeof(int32_t) * cap));
            HIP_TRY(hipMalloc((void **)&g.d_flags[b], cap));
            g.reads_cap[b] = cap;
        }
    }
    const uint64_t *hoff = offsets;
    if (offsets[0] != 0) {  // the staged slice starts at 0
        g.rel[b].resize((size_t)n_reads + 1);
        for (int64_t i = 0; i <= n_reads; i++) g.rel[b][(size_t)i] = offsets[i] - offsets[0];
        hoff = g.rel[b].data();
    }
    // the batch before last must be through (its kernel read these buffers, its outputs left from them).  Waited for
    // on the host: a copy that has to wait for an event on the device was seen to start only after the kernel of the
    // previous batch had ended (rocprofv3 --memory-copy-trace), i.e. not to overlap at all
    if (tk >= 2) HIP_TRY(hipEventSynchronize(g.finished[(tk - 2) & 3]));
    HIP_TRY(hipMemcpyAsync(g.d_seq[b], seq + offsets[0], nbytes, hipMemcpyHostToDevice, g.copy_stream));
    HIP_TRY(hipMemcpyAsync(g.d_off[b], hoff, sizeof(uint64_t) * ((size_t)n_reads + 1), hipMemcpyHostToDevice, g.copy_stream));
    HIP_TRY(hipEventRecord(g.copied[b], g.copy_stream));
    HIP_TRY(hipStreamWaitEvent(run->stream, g.copied[b], 0));
    int rc = launch_batch(run, g.d_seq[b], g.d_off[b], n_reads, first_read_no, class_vi ? g.d_class[b] : nullptr,
                          flags ? g.d_flags[b] : nullptr);
    if (rc) return rc;
    if (class_vi)
        HIP_TRY(hipMemcpyAsync(class_vi, g.d_class[b], sizeof(int32_t) * (size_t)n_reads, hipMemcpyDeviceToHost, run->stream));
    if (flags) HIP_TRY(hipMemcpyAsync(flags, g.d_flags[b], (size_t)n_reads, hipMemcpyDeviceToHost, run->stream));
    HIP_TRY(hipEventRecord(g.finished[tk & 3], run->stream));
    g.tickets = tk + 1;
    *ticket = tk;
    return GS_OK;
} catch (const std::bad_alloc &) {
    return fail(GS_E_NOMEM, "out of host memory");
}

extern "C" int gs_match_wait(gs_run *run, int64_t ticket) {
    if (!run) return fail(GS_E_INVALID, "run is NULL");
    BatchStage &g = run->stage;
    if (ticket < 0 || ticket >= g.tickets) return fail(GS_E_INVALID, "unknown ticket");
    HIP_TRY(hipSetDevice(run->db->device));
    // (a batch three or more behind the latest is through: the submit of the batch two after it waited for it)
    if (ticket + 3 <= g.tickets) return GS_OK;
    HIP_TRY(hipEventSynchronize(g.finished[ticket & 3]));
    return GS_OK;
}

// ---- text mode: raw 4-line FASTQ chunks, records found on the device (gs_text.hip) ------------------------------
extern "C" int gs_pinned_alloc(void **p, size_t bytes) {
    if (!p) return fail(GS_E_INVALID, "NULL argument");
    *p = nullptr;
    hipError_t e = hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? GS_E_NOMEM : GS_E_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    return GS_OK;
}

extern "C" int gs_pinned_free(void *p) {
    if (p) HIP_TRY(hipHostFree(p));
    return GS_OK;
}

static int match_submit_text(gs_run *run, const uint8_t *text, int64_t n_bytes, int64_t n_lines, int mem, int64_t first_read_no,
                             int32_t *class_vi, uint8_t *flags, int64_t *ticket, int64_t fasta_records, int64_t *ml_out = nullptr);

extern "C" int gs_match_submit_fastq_ml(gs_run *run, const uint8_t *text, int64_t n_bytes, int64_t n_lines, int mem, int64_t first_read_no,
                                        int32_t *class_vi, uint8_t *flags, int64_t *n_records, int64_t *consumed_bytes, int64_t *consumed_lines,
                                        int64_t *ticket) {
    if (!n_records || !consumed_bytes) return fail(GS_E_INVALID, "NULL argument");
    int64_t out[3] = {0, 0, 0};
    const int rc = match_submit_text(run, text, n_bytes, n_lines, mem, first_read_no, class_vi, flags, ticket, -1, out);
    *n_records = out[0];
    *consumed_bytes = out[1];
    if (consumed_lines) *consumed_lines = out[2];
    return rc;
}

extern "C" int gs_match_submit_text(gs_run *run, const uint8_t *text, int64_t n_bytes, int64_t n_lines, int mem,
                                    int64_t first_read_no, int32_t *class_vi, uint8_t *flags, int64_t *ticket) {
    return match_submit_text(run, text, n_bytes, n_lines, mem, first_read_no, class_vi, flags, ticket, -1);
}

extern "C" int gs_match_submit_fasta(gs_run *run, const uint8_t *text, int64_t n_bytes, int64_t n_lines, int64_t n_records, int mem,
                                     int64_t first_read_no, int32_t *class_vi, uint8_t *flags, int64_t *ticket) {
    if (n_records < 0) return fail(GS_E_INVALID, "n_records < 0");
    return match_submit_text(run, text, n_bytes, n_lines, mem, first_read_no, class_vi, flags, ticket, n_records);
}

static int match_submit_text(gs_run *run, const uint8_t *text, int64_t n_bytes, int64_t n_lines, int mem, int64_t first_read_no,
                             int32_t *class_vi, uint8_t *flags, int64_t *ticket, int64_t fasta_records, int64_t *ml_out) {
    if (!run) return fail(GS_E_INVALID, "run is NULL");
    HIP_TRY(hipSetDevice(run->db->device));
    const bool fasta = fasta_records >= 0 || ml_out != nullptr;
    if (mem != GS_MEM_HOST && mem != GS_MEM_DEVICE && mem != GS_MEM_DEVICE_TEXT) return fail(GS_E_INVALID, "bad mem");
    int rc = text_submit(run->text, run->stream, text, n_bytes, n_lines, mem == GS_MEM_HOST ? GS_MEM_HOST : GS_MEM_DEVICE, run->db->info.k, ticket,
                         fasta_records, ml_out);
    if (rc) return rc;
    const int64_t n_reads = ml_out ? std::max<int64_t>(ml_out[0], 0) : (fasta ? fasta_records : (n_lines >> 2));
    if (n_reads == 0) return GS_OK;
    const bool dev_out = mem == GS_MEM_DEVICE;  // (GS_MEM_DEVICE_TEXT: the text is in HBM, class_vi / flags are host arrays)
    if ((class_vi || flags) && !dev_out && run->reads_cap < (size_t)n_reads) {
        HIP_TRY(hipStreamSynchronize(run->stream));
        hipFree(run->d_off);
        hipFree(run->d_class);
        hipFree(run->d_flags);
        run->d_off = nullptr;
        run->d_class = nullptr;
        run->d_flags = nullptr;
        run->reads_cap = 0;
        HIP_TRY(hipMalloc((void **)&run->d_off, sizeof(uint64_t) * ((size_t)n_reads + 1)));
        HIP_TRY(hipMalloc((void **)&run->d_class, sizeof(int32_t) * (size_t)n_reads));
        HIP_TRY(hipMalloc((void **)&run->d_flags, (size_t)n_reads));
        run->reads_cap = (size_t)n_reads;
    }
    int32_t *dc = class_vi ? (dev_out ? class_vi : run->d_class) : nullptr;
    uint8_t *df = flags ? (dev_out ? flags : run->d_flags) : nullptr;
    // FASTQ: the sequence lines in place, (start, end) pairs; FASTA: the gathered sequences, running offsets
    rc = launch_batch(run, fasta ? run->text.d_fa_seq : run->text.d_text, (const uint64_t *)run->text.d_off2, n_reads, first_read_no, dc,
                      df, nullptr, nullptr, fasta ? 1 : 2, run->text.d_status + (size_t)run->text.bank * GS_TS_WORDS + GS_TS_SKIP);
    if (rc) return rc;
    if ((rc = text_touched(run->text, run->stream))) return rc;
    run->text.d_last_flags = df;
    if (!dev_out) {  // complete after gs_match_sync
        if (class_vi) HIP_TRY(hipMemcpyAsync(class_vi, run->d_class, sizeof(int32_t) * (size_t)n_reads, hipMemcpyDeviceToHost, run->stream));
        if (flags) HIP_TRY(hipMemcpyAsync(flags, run->d_flags, (size_t)n_reads, hipMemcpyDeviceToHost, run->stream));
    }
    return GS_OK;
}

// the reads matchRead() returned true for (GS_F_RETURNED), as afterMatch writes them (FastqKMerMatcher.java:304-307)
extern "C" int gs_match_compact_text(gs_run *run, int with_probs, int slot, const uint8_t **d_out, int64_t *n_bytes, int64_t *n_records) {
    if (!run) return fail(GS_E_INVALID, "run is NULL");
    HIP_TRY(hipSetDevice(run->db->device));
    return text_compact(run->text, run->stream, GS_F_RETURNED, 1, 1, slot, with_probs, d_out, n_bytes, n_records);
}

extern "C" int gs_gather_descriptors_device(hipStream_t stream, const uint8_t *d_text, const uint32_t *d_nl, const int64_t *d_records, int n, uint8_t *d_out, int stride);

// the descriptor lines (whole first lines, '@' included) of a few records of the last four-line chunk: out[i * stride ..], NUL-terminated
extern "C" int gs_match_text_descriptors(gs_run *run, const int64_t *records, int32_t n, uint8_t *out, int32_t stride) {
    if (!run || n < 0 || (n > 0 && (!records || !out)) || stride < 2) return fail(GS_E_INVALID, "bad argument");
    if (n == 0) return GS_OK;
    TextScan &t = run->text;
    if (t.tickets == 0 || !t.last_four_line) return fail(GS_E_STATE, "the last chunk was not four-line FASTQ");
    for (int32_t i = 0; i < n; i++)
        if (records[i] < 0 || records[i] >= t.last_reads) return fail(GS_E_INVALID, "record outside the last chunk");
    HIP_TRY(hipSetDevice(run->db->device));
    int64_t *d_rec = nullptr;
    uint8_t *d_out = nullptr;
    hipError_t e = hipMalloc((void **)&d_rec, sizeof(int64_t) * (size_t)n);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, (size_t)n * (size_t)stride);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rec, records, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, run->stream);
    int rc = GS_OK;
    if (e == hipSuccess && gs_gather_descriptors_device(run->stream, t.d_text, t.d_nl, d_rec, n, d_out, stride) != GS_OK) rc = fail(GS_E_HIP, gs_deflate_last_error());
    if (e == hipSuccess && !rc) e = hipMemcpyAsync(out, d_out, (size_t)n * (size_t)stride, hipMemcpyDeviceToHost, run->stream);
    if (e == hipSuccess && !rc) e = hipStreamSynchronize(run->stream);
    hipFree(d_rec);
    hipFree(d_out);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? GS_E_NOMEM : GS_E_HIP, std::string("gs_match_text_descriptors: ") + hipGetErrorString(e));
    return rc;
}

extern "C" int gs_match_text_wait_copy(gs_run *run, int64_t ticket) {
    if (!run) return fail(GS_E_INVALID, "run is NULL");
    HIP_TRY(hipSetDevice(run->db->device));
    return text_wait_copy(run->text, ticket);
}

extern "C" int gs_match_text_status(gs_run *run, int64_t *failed_ticket, int64_t *first_bad_record, int64_t totals[3]) {
    if (!run) return fail(GS_E_INVALID, "run is NULL");
    HIP_TRY(hipSetDevice(run->db->device));
    int rc = text_status(run->text, run->stream, failed_ticket, first_bad_record, totals);
    if (rc) return rc;
    return collect_events(run);
}

extern "C" int gs_match_text_select(gs_run *run, int bank) {
    if (!run) return fail(GS_E_INVALID, "run is NULL");
    if (bank < 0 || bank >= GS_TEXT_BANKS) return fail(GS_E_INVALID, "text bank out of range");
    run->text.bank = bank;
    return GS_OK;
}

extern "C" int gs_match_text_clear_error(gs_run *run) {
    if (!run) return fail(GS_E_INVALID, "run is NULL");
    HIP_TRY(hipSetDevice(run->db->device));
    return text_reset(run->text, false, run->stream);
}

extern "C" hipError_t gs_launch_merge_i64(void *dst, const void *src, int64_t n, int op, hipStream_t stream);
extern "C" hipError_t gs_launch_merge_f64(void *dst, const void *src, int64_t n, hipStream_t stream);

// seen bits of the store -> the run's compact bitmap.  The kernels of a striped store write the bitmap themselves (the
// store's memory, partly another GPU's, is never written).
static int run_extract_bitmap(gs_run *run) {
    if (run->db->striped()) return GS_OK;
    HIP_TRY(gs_launch_bitmap_extract(run->db->d_table, run->db->n_slots(), run->d_bitmap, run->db->d_rec, run->db->n_rec, run->stream));
    return GS_OK;
}

// unique k-mers per value index from the bitmap (own or merged) into d_unique
static int run_unique_counts(gs_run *run) {
    const gs_db *db = run->db;
    HIP_TRY(hipMemsetAsync(run->d_unique, 0, sizeof(u64) * (size_t)db->info.n_values, run->stream));
    if (!db->striped()) {
        HIP_TRY(gs_launch_unique_count(db->d_table, run->d_bitmap, db->n_slots(), db->dev.vbits, db->info.n_values, run->d_unique,
                                       db->d_rec, db->n_rec, run->stream));
        return GS_OK;
    }
    // every stripe against its part of the bitmap (a foreign stripe is read over xGMI, once)
    const uint32_t *brec = run->d_bitmap + (db->n_slots() + 31) / 32;
    for (int q = 0; q < db->n_parts; q++) {
        int64_t tfirst = 0, tlocal = 0;
        const u64 *tq = stripe_table(db, q, &tfirst, &tlocal);
        HIP_TRY(gs_launch_unique_count(tq, run->d_bitmap + tfirst * GS_SLOTS_PER_BUCKET / 32, tlocal * GS_SLOTS_PER_BUCKET, db->dev.vbits,
                                       db->info.n_values, run->d_unique, nullptr, 0, run->stream));
        const int64_t first = (int64_t)gs_stripe_first(db->dev.rec_bits, (uint32_t)db->n_parts, (uint32_t)q);
        const int64_t local = (int64_t)gs_stripe_first(db->dev.rec_bits, (uint32_t)db->n_parts, (uint32_t)q + 1) - first;
        HIP_TRY(gs_launch_rec_unique_count(db->stripe_base[q], brec + first, local, db->info.n_values, run->d_unique, run->stream));
    }
    return GS_OK;
}

// the copies of the global-atomic counters into copy 0 (the others start from zero again)
static int fold_stats(gs_run *run) {
    if (!run->stats_spread) return GS_OK;
    HIP_TRY(gs_launch_fold_stats((long long *)run->d_sums, (unsigned long long *)run->d_max, run->d_dsums, (long long)run->db->info.n_values, run->stat_copies, run->stream));
    run->stats_spread = false;
    return GS_OK;
}

extern "C" int gs_match_get_device(gs_run *run, int *device) {
    if (!run || !device) return fail(GS_E_INVALID, "NULL argument");
    *device = run->db->device;
    return GS_OK;
}

extern "C" int gs_match_sync(gs_run *run) {
    if (!run) return fail(GS_E_INVALID, "run is NULL");
    HIP_TRY(hipSetDevice(run->db->device));
    HIP_TRY(hipStreamSynchronize(run->stream));
    return collect_events(run);
}

extern "C" int gs_match_finish(gs_run *run, int64_t *table, double *dtable) {
    if (!run || !table) return fail(GS_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(run->db->device));
    const size_t nv = (size_t)run->db->info.n_values;
    // the four result arrays land in page-locked memory (a copy into pageable memory is staged by the runtime: ~20 us each,
    // one after the other -- a third of the time between two batches of the bench)
    if (!run->h_result) HIP_TRY(hipHostMalloc((void **)&run->h_result, sizeof(u64) * nv * (GS_N_SUMS + 2 + GS_N_DCOLS), hipHostMallocDefault));
    int64_t *sums = reinterpret_cast<int64_t *>(run->h_result);
    int64_t *maxk = sums + nv * GS_N_SUMS;
    u64 *uniq = reinterpret_cast<u64 *>(maxk + nv);
    double *dsums = reinterpret_cast<double *>(uniq + nv);
    {
        const int frc = fold_stats(run);
        if (frc) return frc;
    }
    if (run->cfg.count_unique) {
        if (!run->bitmap_merged) {
            const int xrc = run_extract_bitmap(run);
            if (xrc) return xrc;
        }
        {
            const int urc = run_unique_counts(run);
            if (urc) return urc;
        }
        HIP_TRY(hipMemcpyAsync(uniq, run->d_unique, sizeof(u64) * nv, hipMemcpyDeviceToHost, run->stream));
    }
    HIP_TRY(hipMemcpyAsync(sums, run->d_sums, sizeof(int64_t) * nv * GS_N_SUMS, hipMemcpyDeviceToHost, run->stream));
    HIP_TRY(hipMemcpyAsync(maxk, run->d_max, sizeof(int64_t) * nv, hipMemcpyDeviceToHost, run->stream));
    if (dtable) HIP_TRY(hipMemcpyAsync(dsums, run->d_dsums, sizeof(double) * nv * GS_N_DCOLS, hipMemcpyDeviceToHost, run->stream));
    HIP_TRY(hipStreamSynchronize(run->stream));
    if (dtable) memcpy(dtable, dsums, sizeof(double) * nv * GS_N_DCOLS);
    int rc = collect_events(run);
    if (rc) return rc;
    for (size_t v = 0; v < nv; v++) {
        int64_t *row = table + v * GS_N_COLS;
        const int64_t *s = sums + v * GS_N_SUMS;
        row[GS_C_READS] = s[GS_S_READS];
        row[GS_C_READS_KMERS] = s[GS_S_READS_KMERS];
        row[GS_C_KMERS] = s[GS_S_KMERS];
        row[GS_C_UNIQUE_KMERS] = run->cfg.count_unique ? (int64_t)uniq[v] : -1;
        row[GS_C_CONTIGS] = s[GS_S_CONTIGS];
        row[GS_C_CONTIG_LEN_SQ_SUM] = s[GS_S_CONTIG_LEN_SQ_SUM];
        row[GS_C_READS_1KMER] = s[GS_S_READS_1KMER];
        row[GS_C_READS_BPS] = s[GS_S_READS_BPS];
        const u64 key = (u64)maxk[v];
        row[GS_C_MAX_CONTIG_LEN] = (int64_t)(key >> 40);
        row[GS_C_MAX_CONTIG_READ_NO] = key ? (int64_t)(((1ULL << 40) - 1) - (key & ((1ULL << 40) - 1))) : -1;
    }
    return GS_OK;
}

extern "C" int gs_match_max_contig_reads(gs_run *run, int64_t *read_no) {
    if (!run || !read_no) return fail(GS_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(run->db->device));
    const size_t nv = (size_t)run->db->info.n_values;
    std::vector<int64_t> maxk(nv);
    {
        const int frc = fold_stats(run);
        if (frc) return frc;
    }
    HIP_TRY(hipMemcpyAsync(maxk.data(), run->d_max, sizeof(int64_t) * nv, hipMemcpyDeviceToHost, run->stream));
    HIP_TRY(hipStreamSynchronize(run->stream));
    int rc = collect_events(run);
    if (rc) return rc;
    for (size_t v = 0; v < nv; v++) {
        const u64 key = (u64)maxk[v];
        read_no[v] = key ? (int64_t)(((1ULL << 40) - 1) - (key & ((1ULL << 40) - 1))) : -1;
    }
    return GS_OK;
}

extern "C" int gs_match_reset(gs_run *run) {
    if (!run) return fail(GS_E_INVALID, "run is NULL");
    HIP_TRY(hipSetDevice(run->db->device));
    if (!run->pending.empty()) HIP_TRY(hipStreamSynchronize(run->stream));
    int rc = collect_events(run);
    if (rc) return rc;
    rc = run_clear(run);  // kernel-time counters stay cumulative over the life of the handle
    if (rc) return rc;
    return text_reset(run->text, true, run->stream, true);
}

extern "C" int gs_match_destroy(gs_run *run) {
    if (!run) return GS_OK;
    hipSetDevice(run->db->device);
    hipStreamSynchronize(run->stream);
    if (run->db->unique_owner == run) {
        if (run->seen_dirty && !run->db->striped())
            gs_launch_clear_seen(run->db->d_table, run->db->n_slots(), run->db->d_rec, run->db->n_rec, run->stream);
        hipStreamSynchronize(run->stream);
        run->db->unique_owner = nullptr;
    }
    gs_db *db = run->db;
    run_free(run);
    if (--db->live_runs == 0 && db->destroy_pending) db_free(db);
    return GS_OK;
}

extern "C" int gs_match_device_state(gs_run *run, void **sums, void **max_keys, void **dsums, void **bitmap,
                                     int64_t *bitmap_words) {
    if (!run) return fail(GS_E_INVALID, "run is NULL");
    HIP_TRY(hipSetDevice(run->db->device));
    {
        const int frc = fold_stats(run);
        if (frc) return frc;
    }
    if (bitmap && run->cfg.count_unique && !run->bitmap_merged) {  // refresh the compact copy of the seen bits
        const int xrc = run_extract_bitmap(run);
        if (xrc) return xrc;
    }
    HIP_TRY(hipStreamSynchronize(run->stream));
    if (sums) *sums = run->d_sums;
    if (max_keys) *max_keys = run->d_max;
    if (dsums) *dsums = run->d_dsums;
    if (bitmap) *bitmap = run->d_bitmap;
    if (bitmap_words) *bitmap_words = run->bitmap_words;
    return GS_OK;
}

extern "C" int gs_match_or_bitmap(gs_run *run, const void *parts, int64_t n_parts) {
    if (!run || !parts || n_parts < 1) return fail(GS_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(run->db->device));
    HIP_TRY(gs_launch_bitmap_or(run->d_bitmap, (const uint32_t *)parts, run->bitmap_words, n_parts, run->stream));
    HIP_TRY(hipStreamSynchronize(run->stream));
    run->bitmap_merged = true;  // gs_match_finish counts from this merged bitmap
    return GS_OK;
}

// ---- merge of runs that live in one process (gs_merge.hip)
extern "C" int gs_rccl_merge_leaders(int n_dev, const int *devices, void *const *sums, void *const *maxk, void *const *dsums,
                                     void *const *bitmap, void *const *gather, int64_t n_sums, int64_t n_max, int64_t n_dsums,
                                     int64_t n_words, const hipStream_t *streams, const char **msg);

extern "C" int gs_match_merge(gs_run *const *runs, int n_runs) try {
    if (!runs || n_runs < 1) return fail(GS_E_INVALID, "bad argument");
    for (int i = 0; i < n_runs; i++) {
        if (!runs[i]) return fail(GS_E_INVALID, "run is NULL");
        for (int j = 0; j < i; j++)
            if (runs[j] == runs[i]) return fail(GS_E_INVALID, "the same run twice");
    }
    const gs_run *r0 = runs[0];
    const gs_db_info &i0 = r0->db->info;
    for (int i = 1; i < n_runs; i++) {  // the stores must be replicas: the bitmap is indexed by table slot / record offset
        const gs_db_info &x = runs[i]->db->info;
        if (x.k != i0.k || x.n_values != i0.n_values || x.n_stored != i0.n_stored || x.n_buckets != i0.n_buckets ||
            x.rec_bytes != i0.rec_bytes || x.n_in_records != i0.n_in_records || runs[i]->bitmap_words != r0->bitmap_words)
            return fail(GS_E_INVALID, "gs_match_merge needs runs on replicas of one store");
        if (runs[i]->cfg.count_unique != r0->cfg.count_unique) return fail(GS_E_INVALID, "runs differ in count_unique");
    }
    const size_t nv = (size_t)i0.n_values;
    const bool uniq = r0->cfg.count_unique != 0;
    const int64_t words = r0->bitmap_words;
    // every run's compact bitmap up to date, every stream drained
    for (int i = 0; i < n_runs; i++) {
        gs_run *run = runs[i];
        HIP_TRY(hipSetDevice(run->db->device));
        {
            const int frc = fold_stats(run);
            if (frc) return frc;
        }
        if (uniq && !run->bitmap_merged) {
            const int xrc = run_extract_bitmap(run);
            if (xrc) return xrc;
        }
        HIP_TRY(hipStreamSynchronize(run->stream));
    }
    // ---- stage A: the runs of one device into that device's first run (its leader)
    std::vector<int> devices;
    std::vector<gs_run *> leader;
    for (int i = 0; i < n_runs; i++) {
        gs_run *run = runs[i];
        size_t d = 0;
        while (d < devices.size() && devices[d] != run->db->device) d++;
        if (d == devices.size()) {
            devices.push_back(run->db->device);
            leader.push_back(run);
            continue;
        }
        gs_run *L = leader[d];
        HIP_TRY(hipSetDevice(L->db->device));
        HIP_TRY(gs_launch_merge_i64(L->d_sums, run->d_sums, (int64_t)(nv * GS_N_SUMS), 0, L->stream));
        HIP_TRY(gs_launch_merge_i64(L->d_max, run->d_max, (int64_t)nv, 1, L->stream));
        HIP_TRY(gs_launch_merge_f64(L->d_dsums, run->d_dsums, (int64_t)(nv * GS_N_DCOLS), L->stream));
        if (uniq) HIP_TRY(gs_launch_bitmap_or(L->d_bitmap, run->d_bitmap, words, 1, L->stream));
    }
    // ---- stage B: the leaders among each other over RCCL (GS_MERGE_FORCE_RCCL=1: also for a single device, as a rehearsal)
    const int n_dev = (int)devices.size();
    const char *force = getenv("GS_MERGE_FORCE_RCCL");
    if (n_dev > 1 || (force && atoi(force) != 0)) {
        std::vector<void *> ps((size_t)n_dev), pm((size_t)n_dev), pd((size_t)n_dev), pb((size_t)n_dev), pg((size_t)n_dev, nullptr);
        std::vector<hipStream_t> st((size_t)n_dev);
        int rc = GS_OK;
        for (int d = 0; d < n_dev && rc == GS_OK; d++) {
            gs_run *L = leader[(size_t)d];
            ps[(size_t)d] = L->d_sums;
            pm[(size_t)d] = L->d_max;
            pd[(size_t)d] = L->d_dsums;
            pb[(size_t)d] = L->d_bitmap;
            st[(size_t)d] = L->stream;
            if (uniq) {
                hipSetDevice(L->db->device);
                if (hipMalloc(&pg[(size_t)d], sizeof(uint32_t) * (size_t)words * (size_t)n_dev) != hipSuccess)
                    rc = fail(GS_E_NOMEM, "merge scratch");
            }
        }
        const char *msg = "";
        if (rc == GS_OK) {
            const int m = gs_rccl_merge_leaders(n_dev, devices.data(), ps.data(), pm.data(), pd.data(), pb.data(), pg.data(),
                                                (int64_t)(nv * GS_N_SUMS), (int64_t)nv, (int64_t)(nv * GS_N_DCOLS), uniq ? words : 0,
                                                st.data(), &msg);
            if (m != 0) rc = fail(m == -4 ? GS_E_UNSUPPORTED : GS_E_HIP, msg);
        }
        for (int d = 0; d < n_dev && rc == GS_OK; d++) {
            gs_run *L = leader[(size_t)d];
            hipSetDevice(L->db->device);
            if (uniq && gs_launch_bitmap_or(L->d_bitmap, (const uint32_t *)pg[(size_t)d], words, n_dev, L->stream) != hipSuccess)
                rc = fail(GS_E_HIP, "bitmap OR");
            if (rc == GS_OK && hipStreamSynchronize(L->stream) != hipSuccess) rc = fail(GS_E_HIP, "merge sync");
        }
        for (int d = 0; d < n_dev; d++)
            if (pg[(size_t)d]) {
                hipSetDevice(devices[(size_t)d]);
                hipFree(pg[(size_t)d]);
            }
        if (rc != GS_OK) return rc;
    }
    // ---- stage C: the global state back into every run of the device
    for (int i = 0; i < n_runs; i++) {
        gs_run *run = runs[i];
        size_t d = 0;
        while (devices[d] != run->db->device) d++;
        gs_run *L = leader[d];
        HIP_TRY(hipSetDevice(L->db->device));
        if (run != L) {
            HIP_TRY(hipMemcpyAsync(run->d_sums, L->d_sums, sizeof(int64_t) * nv * GS_N_SUMS, hipMemcpyDeviceToDevice, L->stream));
            HIP_TRY(hipMemcpyAsync(run->d_max, L->d_max, sizeof(int64_t) * nv, hipMemcpyDeviceToDevice, L->stream));
            HIP_TRY(hipMemcpyAsync(run->d_dsums, L->d_dsums, sizeof(double) * nv * GS_N_DCOLS, hipMemcpyDeviceToDevice, L->stream));
            if (uniq) HIP_TRY(hipMemcpyAsync(run->d_bitmap, L->d_bitmap, sizeof(uint32_t) * (size_t)words, hipMemcpyDeviceToDevice, L->stream));
        }
        run->bitmap_merged = uniq;  // gs_match_finish counts from the merged bitmap
    }
    for (gs_run *L : leader) {
        HIP_TRY(hipSetDevice(L->db->device));
        HIP_TRY(hipStreamSynchronize(L->stream));
    }
    return GS_OK;
} catch (const std::bad_alloc &) {
    return fail(GS_E_NOMEM, "out of host memory");
}


// ---------------------------------------------------------------------------------------------------
// DB construction on the device (gs_build.hip; include/gsgpu.h "gs_dbbuild")
// ---------------------------------------------------------------------------------------------------
extern "C" hipError_t gs_launch_build_kmers(const uint8_t *seq, const u64 *off, int64_t n_regions, int64_t total, int k, int lower, int step,
                                            int max_dust, uint32_t first_region, int update, u64 range_lo, u64 range_hi, u64 *keys, uint32_t *vals,
                                            u64 *n_out, hipStream_t stream);
extern "C" hipError_t gs_build_sort(u64 *keys, u64 *keys_alt, uint32_t *vals, uint32_t *vals_alt, int64_t n, int key_bits, u64 **keys_out,
                                    uint32_t **vals_out, hipStream_t stream);
extern "C" hipError_t gs_build_reduce(const u64 *keys, const uint32_t *vals, int64_t n, const int32_t *node_of_region, const int32_t *parent,
                                      const int32_t *depth, uint32_t *flag, int32_t *value, u64 *pos, int64_t *n_out, hipStream_t stream);
extern "C" hipError_t gs_launch_build_scatter(const u64 *keys, const int32_t *value, const uint32_t *flag, const u64 *pos, int64_t n,
                                              int64_t *out_keys, int32_t *out_vals, hipStream_t stream);

struct gs_dbbuild {
    int device = 0, k = 0, lower = 1, step = 1, max_dust = -1;
    int32_t n_values = 0;
    hipStream_t stream = nullptr;
    int32_t *d_tree = nullptr;  // parent | depth
    u64 *d_keys = nullptr;      // the (k-mer, region | update bit) pairs handed in so far
    uint32_t *d_vals = nullptr;
    u64 *d_count = nullptr;     // their number, counted by the kernel
    size_t cap = 0, n_pairs = 0;
    u64 range_lo = 0, range_hi = ~0ULL;  // gs_dbbuild_set_range
    std::vector<int32_t> node_of_region;
    std::vector<int32_t> parent;
    uint8_t *d_seq = nullptr;  // staging of host input
    size_t seq_cap = 0;
    u64 *d_off = nullptr;
    size_t off_cap = 0;
    int64_t *d_out_keys = nullptr;
    int32_t *d_out_vals = nullptr;
    int64_t n_out = 0, n_kmers_seen = 0;
    bool finished = false;
    bool failed = false;  // gs_dbbuild_finish failed half-way: the pairs are gone, nothing can be fetched (sticky GS_E_STATE)
};

static void dbbuild_free(gs_dbbuild *b) {
    hipSetDevice(b->device);
    if (b->stream) hipStreamSynchronize(b->stream);
    hipFree(b->d_tree);
    hipFree(b->d_keys);
    hipFree(b->d_vals);
    hipFree(b->d_count);
    hipFree(b->d_seq);
    hipFree(b->d_off);
    hipFree(b->d_out_keys);
    hipFree(b->d_out_vals);
    if (b->stream) hipStreamDestroy(b->stream);
    delete b;
}

extern "C" int gs_dbbuild_begin(gs_dbbuild **out, int device, int k, int32_t n_values, const int32_t *parent_vi, int lower_case_bases,
                                int max_dust, int step_size) try {
    if (!out) return fail(GS_E_INVALID, "out is NULL");
    *out = nullptr;
    if (k < 1 || k > 31) return fail(GS_E_INVALID, "k must be in [1,31]");
    if (n_values < 1 || n_values > (1 << 24) || !parent_vi) return fail(GS_E_INVALID, "bad tree arrays (n_values must be in [1, 2^24])");
    if (step_size < 1) return fail(GS_E_INVALID, "stepSize must be >= 1 (C/GSConfigKey.java:236)");
    if (max_dust > 32767) return fail(GS_E_INVALID, "maxDust > Short.MAX_VALUE (C/util/CGATLongBuffer.java:78-80)");
    // depths by parent walks; exactly one root: TaxTree.getLowestCommonAncestor answers null for nodes of different trees
    // and the update then keeps the old value (DBGoal.java:243), which depends on the order of the regions
    std::vector<int32_t> depth((size_t)n_values, -1);
    int roots = 0;
    for (int32_t v = 0; v < n_values; v++) {
        const int32_t p = parent_vi[v];
        if (p < -2 || p >= n_values || p == v) return fail(GS_E_INVALID, "parent_vi out of range");
        roots += p == -1;
    }
    if (roots != 1) return fail(GS_E_UNSUPPORTED, "gs_dbbuild needs a tree with exactly one root");
    for (int32_t v = 0; v < n_values; v++) {
        if (parent_vi[v] == -2 || depth[(size_t)v] >= 0) continue;
        std::vector<int32_t> path;
        int32_t x = v;
        while (x >= 0 && depth[(size_t)x] < 0) {
            if (parent_vi[x] == -2) return fail(GS_E_INVALID, "parent_vi points at a value without a node");
            path.push_back(x);
            if ((int32_t)path.size() > n_values) return fail(GS_E_INVALID, "parent_vi contains a cycle");
            x = parent_vi[x];
        }
        int32_t d = x >= 0 ? depth[(size_t)x] + 1 : 0;
        for (size_t i = path.size(); i-- > 0;) depth[(size_t)path[i]] = d++;
    }
    int rc = use_device(device);
    if (rc) return rc;
    gs_dbbuild *b = new gs_dbbuild();
    b->device = device;
    b->k = k;
    b->lower = lower_case_bases != 0;
    b->step = step_size;
    b->max_dust = max_dust < 0 ? -1 : max_dust;
    b->n_values = n_values;
    b->parent.assign(parent_vi, parent_vi + n_values);
    hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_count, sizeof(u64));
    if (e == hipSuccess) e = hipMemset(b->d_count, 0, sizeof(u64));
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_tree, sizeof(int32_t) * 2 * (size_t)n_values);
    if (e == hipSuccess) e = hipMemcpy(b->d_tree, parent_vi, sizeof(int32_t) * (size_t)n_values, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(b->d_tree + n_values, depth.data(), sizeof(int32_t) * (size_t)n_values, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        dbbuild_free(b);
        return fail(e == hipErrorOutOfMemory ? GS_E_NOMEM : GS_E_HIP, std::string("gs_dbbuild_begin: ") + hipGetErrorString(e));
    }
    *out = b;
    return GS_OK;
}
GS_API_CATCH

extern "C" int gs_dbbuild_add(gs_dbbuild *b, const uint8_t *seq, const uint64_t *offsets, const int32_t *node_vi, int64_t n_regions, int mem,
                              int update) try {
    if (!b || n_regions < 0 || (n_regions > 0 && (!seq || !offsets || !node_vi))) return fail(GS_E_INVALID, "bad argument");
    if (b->finished) return fail(GS_E_STATE, "gs_dbbuild_finish has been called");
    if (n_regions == 0) return GS_OK;
    if (mem != GS_MEM_HOST && mem != GS_MEM_DEVICE) return fail(GS_E_INVALID, "mem must be GS_MEM_HOST or GS_MEM_DEVICE");
    HIP_TRY(hipSetDevice(b->device));
    if (b->node_of_region.size() + (size_t)n_regions >= ((size_t)1 << 31)) return fail(GS_E_UNSUPPORTED, "more than 2^31 regions");
    // node_vi is a host array in both cases (one entry per region)
    for (int64_t r = 0; r < n_regions; r++)
        if (node_vi[r] < 0 || node_vi[r] >= b->n_values || b->parent[(size_t)node_vi[r]] == -2) return fail(GS_E_INVALID, "node_vi: not a node of the tree");
    std::vector<uint64_t> hoff;
    const uint64_t *off_host = offsets;
    if (mem == GS_MEM_DEVICE) {
        hoff.resize((size_t)n_regions + 1);
        HIP_TRY(hipMemcpy(hoff.data(), offsets, sizeof(uint64_t) * ((size_t)n_regions + 1), hipMemcpyDeviceToHost));
        off_host = hoff.data();
    }
    if (off_host[0] != 0) return fail(GS_E_INVALID, "offsets[0] must be 0");
    for (int64_t r = 0; r < n_regions; r++)
        if (off_host[r + 1] < off_host[r]) return fail(GS_E_INVALID, "offsets must not decrease");
    const int64_t total = (int64_t)off_host[n_regions];
    const uint8_t *d_seq = seq;
    const u64 *d_off = (const u64 *)offsets;
    if (mem == GS_MEM_HOST) {
        int rc = grow(&b->d_seq, &b->seq_cap, (size_t)std::max<int64_t>(total, 1), b->stream);
        if (!rc) rc = grow(&b->d_off, &b->off_cap, (size_t)n_regions + 1, b->stream);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(b->d_seq, seq, (size_t)total, hipMemcpyHostToDevice, b->stream));
        HIP_TRY(hipMemcpyAsync(b->d_off, offsets, sizeof(u64) * ((size_t)n_regions + 1), hipMemcpyHostToDevice, b->stream));
        d_seq = b->d_seq;
        d_off = b->d_off;
    }
    if (b->n_pairs + (size_t)total > b->cap) {  // the pair buffers grow by doubling
        size_t want = std::max(b->cap * 2, b->n_pairs + (size_t)total);
        want = std::max<size_t>(want, 1 << 20);
        u64 *nk = nullptr;
        uint32_t *nv = nullptr;
        hipError_t e = hipMalloc((void **)&nk, want * sizeof(u64));
        if (e == hipSuccess) e = hipMalloc((void **)&nv, want * sizeof(uint32_t));
        if (e == hipSuccess && b->n_pairs) e = hipMemcpyAsync(nk, b->d_keys, b->n_pairs * sizeof(u64), hipMemcpyDeviceToDevice, b->stream);
        if (e == hipSuccess && b->n_pairs) e = hipMemcpyAsync(nv, b->d_vals, b->n_pairs * sizeof(uint32_t), hipMemcpyDeviceToDevice, b->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
        if (e != hipSuccess) {
            hipFree(nk);
            hipFree(nv);
            return fail(e == hipErrorOutOfMemory ? GS_E_NOMEM : GS_E_HIP, std::string("gs_dbbuild_add: ") + hipGetErrorString(e));
        }
        hipFree(b->d_keys);
        hipFree(b->d_vals);
        b->d_keys = nk;
        b->d_vals = nv;
        b->cap = want;
    }
    HIP_TRY(gs_launch_build_kmers(d_seq, d_off, n_regions, total, b->k, b->lower, b->step, b->max_dust, (uint32_t)b->node_of_region.size(), update != 0,
                                  b->range_lo, b->range_hi, b->d_keys, b->d_vals, b->d_count, b->stream));
    u64 have = 0;
    HIP_TRY(hipMemcpyAsync(&have, b->d_count, sizeof(u64), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));  // (the caller's arrays and the staging buffers are free again)
    b->n_pairs = (size_t)have;
    b->node_of_region.insert(b->node_of_region.end(), node_vi, node_vi + n_regions);
    return GS_OK;
}
GS_API_CATCH

extern "C" int gs_dbbuild_set_range(gs_dbbuild *b, uint64_t lo, uint64_t hi) {
    if (!b || lo >= hi) return fail(GS_E_INVALID, "bad range");
    if (b->n_pairs > 0 || !b->node_of_region.empty()) return fail(GS_E_STATE, "gs_dbbuild_set_range comes before the first gs_dbbuild_add");
    b->range_lo = lo;
    b->range_hi = hi;
    return GS_OK;
}

extern "C" int gs_dbbuild_finish(gs_dbbuild *b, int64_t *n_kmers) try {
    if (!b || !n_kmers) return fail(GS_E_INVALID, "NULL argument");
    if (b->failed) return fail(GS_E_STATE, "this builder's gs_dbbuild_finish failed: its pairs are gone (start over with gs_dbbuild_begin)");
    if (b->finished) {
        *n_kmers = b->n_out;
        return GS_OK;
    }
    *n_kmers = 0;
    HIP_TRY(hipSetDevice(b->device));
    hipFree(b->d_seq);
    hipFree(b->d_off);
    b->d_seq = nullptr;
    b->d_off = nullptr;
    b->seq_cap = b->off_cap = 0;
    const int64_t n = (int64_t)b->n_pairs;
    int rc = GS_OK;
    u64 *keys_alt = nullptr, *pos = nullptr;
    uint32_t *vals_alt = nullptr, *flag = nullptr;
    int32_t *value = nullptr, *d_nor = nullptr;
    auto cleanup = [&] {
        hipFree(keys_alt);
        hipFree(vals_alt);
        hipFree(pos);
        hipFree(flag);
        hipFree(value);
        hipFree(d_nor);
    };
    auto check = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == GS_OK) rc = fail(e == hipErrorOutOfMemory ? GS_E_NOMEM : GS_E_HIP, std::string("gs_dbbuild_finish (") + what + "): " + hipGetErrorString(e));
        return e == hipSuccess;
    };
    if (n > 0) {
        u64 *ks = nullptr;
        uint32_t *vs = nullptr;
        const u64 n_valid = (u64)n;
        bool ok = check(hipMalloc((void **)&keys_alt, (size_t)n * sizeof(u64)), "sort buffers") &&
                  check(hipMalloc((void **)&vals_alt, (size_t)n * sizeof(uint32_t)), "sort buffers") &&
                  check(gs_build_sort(b->d_keys, keys_alt, b->d_vals, vals_alt, n, 2 * b->k, &ks, &vs, b->stream), "sort");
        b->n_kmers_seen = (int64_t)n_valid;
        if (ok && n_valid > 0) {
            const size_t nr = std::max<size_t>(b->node_of_region.size(), 1);
            ok = check(hipMalloc((void **)&d_nor, nr * sizeof(int32_t)), "regions") &&
                 check(hipMemcpy(d_nor, b->node_of_region.data(), b->node_of_region.size() * sizeof(int32_t), hipMemcpyHostToDevice), "regions") &&
                 check(hipMalloc((void **)&flag, (size_t)n_valid * sizeof(uint32_t)), "scratch") &&
                 check(hipMalloc((void **)&value, (size_t)n_valid * sizeof(int32_t)), "scratch") &&
                 check(hipMalloc((void **)&pos, (size_t)n_valid * sizeof(u64)), "scratch") &&
                 check(gs_build_reduce(ks, vs, (int64_t)n_valid, d_nor, b->d_tree, b->d_tree + b->n_values, flag, value, pos, &b->n_out, b->stream), "reduce");
            if (ok && b->n_out > 0)
                ok = check(hipMalloc((void **)&b->d_out_keys, (size_t)b->n_out * sizeof(int64_t)), "result") &&
                     check(hipMalloc((void **)&b->d_out_vals, (size_t)b->n_out * sizeof(int32_t)), "result") &&
                     check(gs_launch_build_scatter(ks, value, flag, pos, (int64_t)n_valid, b->d_out_keys, b->d_out_vals, b->stream), "scatter") &&
                     check(hipStreamSynchronize(b->stream), "scatter");
        }
    }
    cleanup();
    hipFree(b->d_keys);
    hipFree(b->d_vals);
    b->d_keys = nullptr;
    b->d_vals = nullptr;
    b->cap = b->n_pairs = 0;
    if (rc != GS_OK) {  // nothing half-built may be handed out later: no results, no second finish
        hipFree(b->d_out_keys);
        hipFree(b->d_out_vals);
        b->d_out_keys = nullptr;
        b->d_out_vals = nullptr;
        b->n_out = 0;
        b->failed = true;
        return rc;
    }
    b->finished = true;
    *n_kmers = b->n_out;
    return GS_OK;
}
GS_API_CATCH

extern "C" int gs_dbbuild_fetch(gs_dbbuild *b, int64_t *kmers, int32_t *value_idx) {
    if (!b || (b->n_out > 0 && (!kmers || !value_idx))) return fail(GS_E_INVALID, "NULL argument");
    if (!b->finished) return fail(GS_E_STATE, "gs_dbbuild_finish first");
    HIP_TRY(hipSetDevice(b->device));
    if (b->n_out > 0) {
        HIP_TRY(hipMemcpy(kmers, b->d_out_keys, (size_t)b->n_out * sizeof(int64_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(value_idx, b->d_out_vals, (size_t)b->n_out * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    return GS_OK;
}

extern "C" int gs_dbbuild_to_db(gs_dbbuild *b, gs_db **out) try {
    if (!b || !out) return fail(GS_E_INVALID, "NULL argument");
    *out = nullptr;
    if (!b->finished) return fail(GS_E_STATE, "gs_dbbuild_finish first");
    if (b->n_out <= 0 || b->k < GS_MIN_K || b->n_values > GS_REC_MAX_VALUES || b->n_out >= ((int64_t)1 << 31))
        return fail(GS_E_UNSUPPORTED, "gs_dbbuild_to_db serves stores with records (k >= 19, at most 2^21 values, 1 .. 2^31 - 1 k-mers): fetch the arrays and call gs_db_create");
    HIP_TRY(hipSetDevice(b->device));
    std::vector<int32_t> parent, depth, tin, tout;
    int rc = tree_arrays(b->n_values, b->parent.data(), parent, depth, tin, tout);
    if (rc) return rc;
    BuildTrace trace;
    rc = db_create_on_device(out, b->device, b->k, b->n_out, b->d_out_keys, b->d_out_vals, b->n_values, parent, depth, tin, tout, trace, true);
    if (rc == 0) return fail(GS_E_UNSUPPORTED, "gs_dbbuild_to_db: no k-mer of this build fits a record: fetch the arrays and call gs_db_create");
    return rc < 0 ? rc : GS_OK;
}
GS_API_CATCH

extern "C" int gs_dbbuild_destroy(gs_dbbuild *b) {
    if (b) dbbuild_free(b);
    return GS_OK;
}

// ---- DB-partitioned mode: encode / probe / reduce as separate steps (all pointers are device pointers)
extern "C" int gs_match_encode(gs_run *run, const uint8_t *seq, const uint64_t *offsets, int64_t n_reads,
                               const uint64_t *pos_off, uint64_t *keys) {
    if (!run || (n_reads > 0 && (!seq || !offsets || !pos_off || !keys))) return fail(GS_E_INVALID, "NULL argument");
    if (run->db->n_rec > 0) return fail(GS_E_STATE, "the split pipeline needs a partition store (gs_db_create_part): this store keeps k-mers in super-k-mer records");
    if (n_reads <= 0) return GS_OK;
    HIP_TRY(hipSetDevice(run->db->device));
    GsEncodeParams P{};
    P.k = run->db->info.k;
    P.seq = seq;
    P.off = offsets;
    P.n_reads = n_reads;
    P.pos_off = (const unsigned long long *)pos_off;
    P.keys = (unsigned long long *)keys;
    P.mgate = run->db->dev.mgate;
    P.mgate_bits = run->db->dev.mgate_bits;
    int grid = (int)std::min<int64_t>((int64_t)run->db->n_cu * 8, (n_reads + 3) / 4);
    if (grid < 1) grid = 1;
    HIP_TRY(gs_launch_encode(&P, grid, run->stream));
    return GS_OK;
}

extern "C" hipError_t gs_launch_encode_route(const GsEncodeParams *P, const GsRouteParams *R, int grid, hipStream_t stream);
extern "C" hipError_t gs_launch_unroute_region(const uint32_t *idx, const int32_t *back, int64_t n, int32_t *nodes, hipStream_t stream);

static int route_grid(const gs_run *run, int64_t n_reads) {  // workgroups of GS_BLOCK / 64 waves (one read per wave at a time)
    return (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)run->db->n_cu * 8, (n_reads + 3) / 4));
}

extern "C" int gs_match_route_geometry(const gs_run *run, int64_t n_reads, int32_t *n_waves, int32_t *chunk) {
    if (!run || n_reads < 0) return fail(GS_E_INVALID, "bad argument");
    if (n_waves) *n_waves = route_grid(run, n_reads) * (GS_BLOCK / 64);
    if (chunk) *chunk = GS_ROUTE_CHUNK;
    return GS_OK;
}

extern "C" int gs_match_encode_route(gs_run *run, const uint8_t *seq, const uint64_t *offsets, int64_t n_reads, const uint64_t *pos_off,
                                     int n_parts, int64_t cap, uint64_t *send_keys, uint32_t *send_idx, int32_t *nodes,
                                     int64_t *counts, int *overflow) {
    if (!run || !counts || !overflow || n_parts < 1 || n_parts > 64 || cap < GS_ROUTE_CHUNK || (cap % GS_ROUTE_CHUNK) != 0 ||
        (n_reads > 0 && (!seq || !offsets || !pos_off || !send_keys || !send_idx || !nodes)))
        return fail(GS_E_INVALID, "bad argument (cap must be a multiple of 2048)");
    if (run->db->n_rec > 0) return fail(GS_E_STATE, "the split pipeline needs a partition store (gs_db_create_part): this store keeps k-mers in super-k-mer records");
    for (int i = 0; i < n_parts; i++) counts[i] = 0;
    *overflow = 0;
    if (n_reads <= 0) return GS_OK;
    HIP_TRY(hipSetDevice(run->db->device));
    {   // a routed key carries its position in the batch as 32 bits (send_idx, ~0 = unused slot): as gs_route_keys, refuse a batch
        // whose positions do not fit instead of scattering the answers to truncated positions
        u64 n_pos = 0;
        HIP_TRY(hipMemcpyAsync(&n_pos, pos_off + n_reads, sizeof(u64), hipMemcpyDeviceToHost, run->stream));
        HIP_TRY(hipStreamSynchronize(run->stream));
        if (n_pos >= 0xffffffffULL) return fail(GS_E_INVALID, "more than 2^32-2 k-mer positions in one batch: split the batch");
    }
    if (!run->d_route_cursors) HIP_TRY(hipMalloc((void **)&run->d_route_cursors, sizeof(u64) * 65));
    HIP_TRY(hipMemsetAsync(run->d_route_cursors, 0, sizeof(u64) * 65, run->stream));
    GsEncodeParams P{};
    P.k = run->db->info.k;
    P.seq = seq;
    P.off = offsets;
    P.n_reads = n_reads;
    P.pos_off = (const unsigned long long *)pos_off;
    P.keys = nullptr;
    P.mgate = run->db->dev.mgate;
    P.mgate_bits = run->db->dev.mgate_bits;
    GsRouteParams R{};
    R.n_parts = n_parts;
    R.cap = (unsigned long long)cap;
    R.cursors = (unsigned long long *)run->d_route_cursors;
    R.send_keys = (unsigned long long *)send_keys;
    R.send_idx = send_idx;
    R.nodes = nodes;
    HIP_TRY(gs_launch_encode_route(&P, &R, route_grid(run, n_reads), run->stream));
    u64 h[65];
    HIP_TRY(hipMemcpyAsync(h, run->d_route_cursors, sizeof(h), hipMemcpyDeviceToHost, run->stream));
    HIP_TRY(hipStreamSynchronize(run->stream));
    for (int i = 0; i < n_parts; i++) counts[i] = (int64_t)std::min<u64>(h[i], (u64)cap);
    *overflow = h[64] != 0;
    return GS_OK;
}

extern "C" int gs_unroute_region(gs_run *run, const uint32_t *idx, const int32_t *back, int64_t n, int32_t *nodes) {
    if (!run || (n > 0 && (!idx || !back || !nodes))) return fail(GS_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(run->db->device));
    HIP_TRY(gs_launch_unroute_region(idx, back, n, nodes, run->stream));
    return GS_OK;
}

extern "C" int gs_match_probe_keys(gs_run *run, const uint64_t *keys, int64_t n_keys, int32_t *nodes) {
    if (!run || (n_keys > 0 && (!keys || !nodes))) return fail(GS_E_INVALID, "NULL argument");
    if (run->db->n_rec > 0) return fail(GS_E_STATE, "the split pipeline needs a partition store (gs_db_create_part): this store keeps k-mers in super-k-mer records");
    if (n_keys <= 0) return GS_OK;
    HIP_TRY(hipSetDevice(run->db->device));
    if (run->cfg.count_unique) {
        run->seen_dirty = true;
        run->bitmap_merged = false;
    }
    HIP_TRY(gs_launch_probe_keys(&run->db->dev, (const u64 *)keys, n_keys, nodes, run->cfg.count_unique, run->stream));
    return GS_OK;
}

extern "C" int gs_match_reduce(gs_run *run, const uint8_t *seq, const uint64_t *offsets, int64_t n_reads,
                               int64_t first_read_no, const uint64_t *pos_off, const int32_t *nodes, int32_t *class_vi,
                               uint8_t *flags) {
    if (!run || (n_reads > 0 && (!seq || !offsets || !pos_off || !nodes))) return fail(GS_E_INVALID, "NULL argument");
    if (n_reads <= 0) return GS_OK;
    HIP_TRY(hipSetDevice(run->db->device));
    return launch_batch(run, seq, offsets, n_reads, first_read_no, class_vi, flags, nodes, pos_off);
}

extern "C" hipError_t gs_launch_route_count(const u64 *keys, int64_t n, int n_parts, u64 *counts, hipStream_t stream);
extern "C" hipError_t gs_launch_route_scatter(const u64 *keys, int64_t n, int n_parts, u64 *cursors, u64 *send_keys,
                                               uint32_t *idx, int32_t *nodes, hipStream_t stream);
extern "C" hipError_t gs_launch_unroute(const u64 *keys, const uint32_t *idx, const int32_t *back, int64_t n_routed, int32_t *nodes,
                                         int64_t n_keys, hipStream_t stream);

// groups the valid keys by owner rank (counting sort on the device); counts[n_parts] is a HOST array
extern "C" int gs_route_keys(gs_run *run, const uint64_t *keys, int64_t n_keys, int n_parts, uint64_t *send_keys,
                             uint32_t *idx, int64_t *counts, int32_t *nodes) {
    if (!run || !counts || n_parts < 1 || n_parts > 64 || (n_keys > 0 && (!keys || !send_keys || !idx)))
        return fail(GS_E_INVALID, "bad argument");
    if (n_keys >= ((int64_t)1 << 32)) return fail(GS_E_INVALID, "more than 2^32-1 keys in one batch");
    HIP_TRY(hipSetDevice(run->db->device));
    for (int i = 0; i < n_parts; i++) counts[i] = 0;
    if (n_keys <= 0) return GS_OK;
    u64 *d_counts = nullptr;
    HIP_TRY(hipMalloc((void **)&d_counts, sizeof(u64) * 64));
    hipError_t e = hipMemsetAsync(d_counts, 0, sizeof(u64) * 64, run->stream);
    if (e == hipSuccess) e = gs_launch_route_count((const u64 *)keys, n_keys, n_parts, d_counts, run->stream);
    u64 h[64];
    if (e == hipSuccess) e = hipMemcpyAsync(h, d_counts, sizeof(u64) * 64, hipMemcpyDeviceToHost, run->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(run->stream);
    u64 cur[64], acc = 0;
    for (int i = 0; i < 64; i++) {
        cur[i] = acc;
        if (i < n_parts) {
            counts[i] = (int64_t)h[i];
            acc += h[i];
        }
    }
    if (e == hipSuccess) e = hipMemcpyAsync(d_counts, cur, sizeof(u64) * 64, hipMemcpyHostToDevice, run->stream);
    if (e == hipSuccess)
        e = gs_launch_route_scatter((const u64 *)keys, n_keys, n_parts, d_counts, (u64 *)send_keys, idx, nodes, run->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(run->stream);
    hipFree(d_counts);
    if (e != hipSuccess) return fail(GS_E_HIP, std::string("gs_route_keys: ") + hipGetErrorString(e));
    return GS_OK;
}

// inverse of gs_route_keys for the nodes that came back: nodes[idx[i]] = back[i]; unrouted positions read -2 (key ~0)
// or -1 (key ~0 - 1: ruled out by the gate of the encoding rank)
extern "C" int gs_unroute_nodes(gs_run *run, const uint64_t *keys, const uint32_t *idx, const int32_t *back,
                                int64_t n_routed, int32_t *nodes, int64_t n_keys) {
    if (!run || (n_keys > 0 && !nodes) || (n_routed > 0 && (!idx || !back))) return fail(GS_E_INVALID, "bad argument");
    if (n_keys <= 0) return GS_OK;
    HIP_TRY(hipSetDevice(run->db->device));
    HIP_TRY(gs_launch_unroute((const u64 *)keys, idx, back, n_routed, nodes, n_keys, run->stream));
    return GS_OK;
}

// ---- Kraken-style segments (two passes: count, host prefix sum, write)
static int stage_batch(gs_run *run, const uint8_t *seq, const uint64_t *offsets, int64_t n_reads, int mem,
                       const uint8_t **d_seq, const uint64_t **d_off) {
    if (mem == GS_MEM_DEVICE) {
        *d_seq = seq;
        *d_off = offsets;
        return GS_OK;
    }
    if (mem != GS_MEM_HOST) return fail(GS_E_INVALID, "mem must be GS_MEM_HOST or GS_MEM_DEVICE");
    const size_t nbytes = (size_t)(offsets[n_reads] - offsets[0]);
    HIP_TRY(hipStreamSynchronize(run->stream));
    if (run->seq_cap < nbytes + 1) {
        hipFree(run->d_seq);
        run->d_seq = nullptr;
        run->seq_cap = 0;
        HIP_TRY(hipMalloc((void **)&run->d_seq, nbytes + 1));
        run->seq_cap = nbytes + 1;
    }
    if (run->reads_cap < (size_t)n_reads) {
        hipFree(run->d_off);
        hipFree(run->d_class);
        hipFree(run->d_flags);
        run->d_off = nullptr;
        run->d_class = nullptr;
        run->d_flags = nullptr;
        run->reads_cap = 0;
        HIP_TRY(hipMalloc((void **)&run->d_off, sizeof(uint64_t) * ((size_t)n_reads + 1)));
        HIP_TRY(hipMalloc((void **)&run->d_class, sizeof(int32_t) * (size_t)n_reads));
        HIP_TRY(hipMalloc((void **)&run->d_flags, (size_t)n_reads));
        run->reads_cap = (size_t)n_reads;
    }
    std::vector<uint64_t> rel((size_t)n_reads + 1);
    for (int64_t i = 0; i <= n_reads; i++) rel[(size_t)i] = offsets[i] - offsets[0];
    HIP_TRY(hipMemcpy(run->d_seq, seq + offsets[0], nbytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(run->d_off, rel.data(), sizeof(uint64_t) * ((size_t)n_reads + 1), hipMemcpyHostToDevice));
    *d_seq = run->d_seq;
    *d_off = run->d_off;
    return GS_OK;
}

static int segments_core(gs_run *run, const uint8_t *d_seq, const uint64_t *d_off, int64_t n_reads, int off_stride, uint64_t *seg_off) {
    hipFree(run->d_seg_count);
    hipFree(run->d_seg_off);
    hipFree(run->d_seg_code);
    hipFree(run->d_seg_start);
    run->d_seg_count = nullptr;
    run->d_seg_off = nullptr;
    run->d_seg_code = run->d_seg_start = nullptr;
    HIP_TRY(hipMalloc((void **)&run->d_seg_count, sizeof(uint32_t) * (size_t)n_reads));
    HIP_TRY(hipMalloc((void **)&run->d_seg_off, sizeof(u64) * ((size_t)n_reads + 1)));
    GsSegParams P{};
    P.db = run->db->dev;
    P.seq = d_seq;
    P.off = d_off;
    P.off_stride = off_stride;
    P.n_reads = n_reads;
    P.seg_count = run->d_seg_count;
    P.huge_min = GS_HUGE_MIN;
    if (const char *e = getenv("GS_HUGE_MIN")) P.huge_min = std::max(129, atoi(e));
    int grid = (int)std::min<int64_t>(run->grid, (n_reads + 3) / 4);
    if (grid < 1) grid = 1;
    HIP_TRY(gs_launch_segments(&P, 0, grid, run->stream));
    std::vector<uint32_t> counts((size_t)n_reads);
    HIP_TRY(hipMemcpyAsync(counts.data(), run->d_seg_count, sizeof(uint32_t) * (size_t)n_reads, hipMemcpyDeviceToHost,
                           run->stream));
    HIP_TRY(hipStreamSynchronize(run->stream));
    // Reads of tens of thousands of positions and more were left out (one wave would walk a chromosome for 150 ms per pass): they come
    // back in pieces of GS_SEG_PIECE_ITERS iterations, one wave each.  A piece counts its first position as the start of a run; the
    // seams where the node goes on are taken off here, and the write pass gets the node in front of every piece.
    std::vector<GsSegPiece> pieces;
    std::vector<size_t> first_piece;  // per left-out read, + end
    std::vector<int64_t> huge_reads;
    struct Dev {  // (freed on every way out)
        GsSegPiece *pieces = nullptr;
        GsSegPieceOut *out = nullptr;
        ~Dev() {
            hipFree(pieces);
            hipFree(out);
        }
    } dv;
    for (int64_t r = 0; r < n_reads; r++)
        if (counts[(size_t)r] == GS_SEG_HUGE) huge_reads.push_back(r);
    if (!huge_reads.empty()) {
        const int k = run->db->info.k;
        for (int64_t r : huge_reads) {
            uint64_t se[2];
            HIP_TRY(hipMemcpy(se, d_off + r * off_stride, sizeof(se), hipMemcpyDeviceToHost));
            const int64_t max = (int64_t)(se[1] - se[0]) - k + 1, n_iter = (max + 127) >> 7;
            first_piece.push_back(pieces.size());
            for (int64_t it = 0; it < n_iter; it += GS_SEG_PIECE_ITERS) {
                GsSegPiece pc{};
                pc.read = (uint32_t)r;
                pc.it0 = (int32_t)it;
                pc.n_iter = (int32_t)std::min<int64_t>(GS_SEG_PIECE_ITERS, n_iter - it);
                pc.carry = -3;  // GS_NODE_NONE
                pieces.push_back(pc);
            }
        }
        first_piece.push_back(pieces.size());
        HIP_TRY(hipMalloc((void **)&dv.pieces, sizeof(GsSegPiece) * pieces.size()));
        HIP_TRY(hipMalloc((void **)&dv.out, sizeof(GsSegPieceOut) * pieces.size()));
        HIP_TRY(hipMemcpyAsync(dv.pieces, pieces.data(), sizeof(GsSegPiece) * pieces.size(), hipMemcpyHostToDevice, run->stream));
        GsSegParams Q = P;
        Q.pieces = dv.pieces;
        Q.piece_out = dv.out;
        Q.n_pieces = (int64_t)pieces.size();
        const int pgrid = (int)std::max<int64_t>(1, std::min<int64_t>(run->grid, ((int64_t)pieces.size() + 3) / 4));
        HIP_TRY(gs_launch_segments(&Q, 0, pgrid, run->stream));
        std::vector<GsSegPieceOut> outs(pieces.size());
        HIP_TRY(hipMemcpyAsync(outs.data(), dv.out, sizeof(GsSegPieceOut) * pieces.size(), hipMemcpyDeviceToHost, run->stream));
        HIP_TRY(hipStreamSynchronize(run->stream));
        for (size_t h = 0; h < huge_reads.size(); h++) {
            uint64_t total = 0;
            for (size_t i = first_piece[h]; i < first_piece[h + 1]; i++) {
                uint32_t c = outs[i].count;
                if (i > first_piece[h]) {
                    pieces[i].carry = outs[i - 1].last_node;
                    if (outs[i].first_node == outs[i - 1].last_node) c--;  // the run goes on across the seam
                }
                pieces[i].out_off = total;
                total += c;
            }
            if (total >= GS_SEG_HUGE) return fail(GS_E_INVALID, "more than 2^32 - 2 segments in one read");
            counts[(size_t)huge_reads[h]] = (uint32_t)total;
        }
    }
    for (int64_t i = 0; i < n_reads; i++) seg_off[i + 1] = seg_off[i] + counts[(size_t)i];
    run->seg_total = (int64_t)seg_off[n_reads];
    if (run->seg_total > 0) {
        HIP_TRY(hipMalloc((void **)&run->d_seg_code, sizeof(int32_t) * (size_t)run->seg_total));
        HIP_TRY(hipMalloc((void **)&run->d_seg_start, sizeof(int32_t) * (size_t)run->seg_total));
        HIP_TRY(hipMemcpyAsync(run->d_seg_off, seg_off, sizeof(u64) * ((size_t)n_reads + 1), hipMemcpyHostToDevice, run->stream));
        P.seg_off = run->d_seg_off;
        P.seg_code = run->d_seg_code;
        P.seg_start = run->d_seg_start;
        HIP_TRY(gs_launch_segments(&P, 1, grid, run->stream));
        if (!pieces.empty()) {
            HIP_TRY(hipMemcpyAsync(dv.pieces, pieces.data(), sizeof(GsSegPiece) * pieces.size(), hipMemcpyHostToDevice, run->stream));
            GsSegParams Q = P;
            Q.pieces = dv.pieces;
            Q.n_pieces = (int64_t)pieces.size();
            const int pgrid = (int)std::max<int64_t>(1, std::min<int64_t>(run->grid, ((int64_t)pieces.size() + 3) / 4));
            HIP_TRY(gs_launch_segments(&Q, 1, pgrid, run->stream));
        }
        HIP_TRY(hipStreamSynchronize(run->stream));
    }
    return GS_OK;
}

extern "C" int gs_match_segments(gs_run *run, const uint8_t *seq, const uint64_t *offsets, int64_t n_reads, int mem,
                                 uint64_t *seg_off) {
    if (!run || !seg_off) return fail(GS_E_INVALID, "NULL argument");
    if (n_reads < 0 || (n_reads > 0 && (!seq || !offsets))) return fail(GS_E_INVALID, "bad batch arrays");
    HIP_TRY(hipSetDevice(run->db->device));
    seg_off[0] = 0;
    run->seg_total = 0;
    if (n_reads == 0) return GS_OK;
    const uint8_t *d_seq = nullptr;
    const uint64_t *d_off = nullptr;
    int rc = stage_batch(run, seq, offsets, n_reads, mem, &d_seq, &d_off);
    if (rc) return rc;
    return segments_core(run, d_seq, d_off, n_reads, 1, seg_off);
}

// the same for the reads of the most recent text chunk (which must not have been refused: gs_match_text_status)
extern "C" int gs_match_segments_text(gs_run *run, uint64_t *seg_off) {
    if (!run || !seg_off) return fail(GS_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(run->db->device));
    seg_off[0] = 0;
    run->seg_total = 0;
    const int64_t n_reads = run->text.last_reads;
    if (run->text.tickets == 0) return fail(GS_E_STATE, "no text chunk has been submitted");
    if (n_reads == 0) return GS_OK;
    const bool fasta = run->text.last_fasta;  // (the gathered sequences stay in place until the next FASTA chunk)
    const int rc = segments_core(run, fasta ? run->text.d_fa_seq : run->text.d_text, (const uint64_t *)run->text.d_off2, n_reads,
                                 fasta ? 1 : 2, seg_off);
    if (rc) return rc;
    return text_touched(run->text, run->stream);
}

// newline offsets of the most recent text chunk (the record geometry for per-read writers); synchronises
extern "C" int gs_match_text_read_bounds(gs_run *run, uint64_t *bounds) {
    if (!run || !bounds) return fail(GS_E_INVALID, "NULL argument");
    if (run->text.tickets == 0) return fail(GS_E_STATE, "no text chunk has been submitted");
    if (!run->text.last_fasta) return fail(GS_E_STATE, "the last chunk was four-line FASTQ: its reads lie in the text (gs_match_text_newlines)");
    HIP_TRY(hipSetDevice(run->db->device));
    HIP_TRY(hipMemcpyAsync(bounds, run->text.d_off2, sizeof(uint64_t) * ((size_t)run->text.last_reads + 1), hipMemcpyDeviceToHost, run->stream));
    HIP_TRY(hipStreamSynchronize(run->stream));
    return collect_events(run);
}

extern "C" int gs_match_text_line_classes(gs_run *run, uint8_t *classes) {
    if (!run || !classes) return fail(GS_E_INVALID, "NULL argument");
    if (run->text.tickets == 0 || !run->text.d_ml_class) return fail(GS_E_STATE, "no general FASTQ chunk has been submitted");
    HIP_TRY(hipSetDevice(run->db->device));
    if (run->text.last_lines > 0)
        HIP_TRY(hipMemcpyAsync(classes, run->text.d_ml_class, (size_t)run->text.last_lines, hipMemcpyDeviceToHost, run->stream));
    HIP_TRY(hipStreamSynchronize(run->stream));
    return collect_events(run);
}

extern "C" int gs_match_text_newlines(gs_run *run, uint32_t *newlines) {
    if (!run || !newlines) return fail(GS_E_INVALID, "NULL argument");
    if (run->text.tickets == 0) return fail(GS_E_STATE, "no text chunk has been submitted");
    HIP_TRY(hipSetDevice(run->db->device));
    if (run->text.last_lines > 0)
        HIP_TRY(hipMemcpyAsync(newlines, run->text.d_nl, sizeof(uint32_t) * (size_t)run->text.last_lines, hipMemcpyDeviceToHost, run->stream));
    HIP_TRY(hipStreamSynchronize(run->stream));
    return collect_events(run);
}

extern "C" int gs_match_segments_fetch(gs_run *run, int32_t *codes, int32_t *starts) {
    if (!run || (run->seg_total > 0 && (!codes || !starts))) return fail(GS_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(run->db->device));
    if (run->seg_total > 0) {
        HIP_TRY(hipMemcpy(codes, run->d_seg_code, sizeof(int32_t) * (size_t)run->seg_total, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(starts, run->d_seg_start, sizeof(int32_t) * (size_t)run->seg_total, hipMemcpyDeviceToHost));
    }
    return GS_OK;
}

extern "C" int gs_match_max_counts(gs_run *run, int16_t *out) {
    if (!run || !out) return fail(GS_E_INVALID, "NULL argument");
    const int N = run->cfg.max_kmer_res_counts;
    if (N <= 0 || !run->d_hit_counts) return fail(GS_E_STATE, "the run was not begun with max_kmer_res_counts > 0");
    if (!run->cfg.count_unique) return fail(GS_E_STATE, "max k-mer counts need count_unique");
    HIP_TRY(hipSetDevice(run->db->device));
    const size_t n_slots = (size_t)run->db->info.n_buckets * GS_SLOTS_PER_BUCKET;
    const size_t nv = (size_t)run->db->info.n_values;
    std::vector<u64> table(n_slots);
    std::vector<uint32_t> counts(n_slots);
    HIP_TRY(hipStreamSynchronize(run->stream));
    if (run->db->striped()) {  // the slices of the table from their stripes, the seen bits from the run's bitmap
        std::vector<uint32_t> tseen((n_slots + 31) / 32);
        HIP_TRY(hipMemcpy(tseen.data(), run->d_bitmap, tseen.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (int q = 0; q < run->db->n_parts; q++) {
            int64_t tfirst = 0, tlocal = 0;
            const u64 *tq = stripe_table(run->db, q, &tfirst, &tlocal);
            HIP_TRY(hipMemcpy(table.data() + (size_t)tfirst * GS_SLOTS_PER_BUCKET, tq, (size_t)tlocal * GS_SLOTS_PER_BUCKET * sizeof(u64), hipMemcpyDeviceToHost));
        }
        for (size_t i = 0; i < n_slots; i++) table[i] = (table[i] & ~1ULL) | ((tseen[i >> 5] >> (i & 31)) & 1u);
    } else
        HIP_TRY(hipMemcpy(table.data(), run->db->d_table, n_slots * sizeof(u64), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(counts.data(), run->d_hit_counts, n_slots * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::fill(out, out + (nv + 1) * (size_t)N, (int16_t)0);
    const u64 vmask = ((u64)1 << run->db->dev.vbits) - 1;
    auto update = [N](int16_t count, int16_t *target) {  // updateMaxCounts (:198-209)
        for (int j = 0; j < N; j++)
            if (count > target[j]) {
                for (int kk = N - 1; kk > j; kk--) target[kk] = target[kk - 1];
                target[j] = count;
                return;
            }
    };
    for (size_t i = 0; i < n_slots; i++) {
        const u64 s = table[i];
        if (!(s & 1ULL)) continue;  // bitVector.get(i)
        const int vi = (int)((s >> 1) & vmask) - 1;
        if (vi < 0) continue;
        const int16_t c = (int16_t)(uint16_t)(counts[i] & 0xffffu);  // Java short arithmetic wraps
        update(c, out + (size_t)vi * N);
        update(c, out + nv * (size_t)N);
    }
    if (run->db->n_rec > 0) {  // the k-mers that live in super-k-mer records
        const size_t n_rec = (size_t)run->db->n_rec;
        std::vector<u64> rec(n_rec * GS_REC_WORDS);
        std::vector<uint32_t> rcounts(n_rec * GS_REC_SLOTS);
        std::vector<uint32_t> rseen;  // striped store: the seen bits are the run's own (one word per record bucket)
        const gs_db *db = run->db;
        if (db->striped()) {
            for (int q = 0; q < db->n_parts; q++) {
                const size_t first = (size_t)gs_stripe_first(db->dev.rec_bits, (uint32_t)db->n_parts, (uint32_t)q);
                const size_t local = (size_t)gs_stripe_first(db->dev.rec_bits, (uint32_t)db->n_parts, (uint32_t)q + 1) - first;
                HIP_TRY(hipMemcpy(rec.data() + first * GS_REC_WORDS, db->stripe_base[q], local * GS_REC_WORDS * sizeof(u64), hipMemcpyDeviceToHost));
            }
            rseen.resize(n_rec);
            HIP_TRY(hipMemcpy(rseen.data(), run->d_bitmap + (n_slots + 31) / 32, n_rec * sizeof(uint32_t), hipMemcpyDeviceToHost));
        } else
            HIP_TRY(hipMemcpy(rec.data(), run->db->d_rec, rec.size() * sizeof(u64), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(rcounts.data(), run->d_hit_counts + n_slots, rcounts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (size_t b = 0; b < n_rec; b++) {
            const u64 *rp = rec.data() + b * GS_REC_WORDS;
            uint32_t seen = (db->striped() ? rseen[b] : (uint32_t)(rp[0] >> GS_REC_WIN_BITS)) & (uint32_t)(rp[1] >> GS_REC_WIN_BITS);
            for (; seen; seen &= seen - 1) {
                const int j = __builtin_ctz(seen);
                const size_t vi = (size_t)((rp[2 + j / 3] >> (GS_REC_VAL_BITS * (j % 3))) & (GS_REC_MAX_VALUES - 1));
                if (vi >= nv) continue;
                const int16_t c = (int16_t)(uint16_t)(rcounts[b * GS_REC_SLOTS + (size_t)j] & 0xffffu);
                update(c, out + vi * N);
                update(c, out + nv * (size_t)N);
            }
        }
    }
    return GS_OK;
}

extern "C" int gs_match_kernel_time(gs_run *run, int64_t *launches, double *total_ms) {
    if (!run) return fail(GS_E_INVALID, "run is NULL");
    HIP_TRY(hipSetDevice(run->db->device));
    HIP_TRY(hipStreamSynchronize(run->stream));
    int rc = collect_events(run);
    if (rc) return rc;
    if (launches) *launches = run->launches;
    if (total_ms) *total_ms = run->total_ms;
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------------
// filter
// ---------------------------------------------------------------------------------------------------
struct gs_bloom {
    int device = 0;
    int kind = 0;
    int64_t bits = 0;
    int32_t n_hashes = 0;
    int64_t n_words = 0;
    u64 *d_words = nullptr;
    int64_t *d_factors = nullptr;
    hipStream_t stream = nullptr;
    int n_cu = 256;
    uint8_t *d_seq = nullptr;
    uint64_t *d_off = nullptr;
    uint8_t *d_accept = nullptr;
    size_t seq_cap = 0, reads_cap = 0;
    TextScan text;  // text mode (gs_filter_submit_text)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    int64_t launches = 0;
    double total_ms = 0;
};

extern "C" int gs_bloom_create(gs_bloom **out, int device, int kind, int64_t bits, int32_t n_hashes,
                               const int64_t *hash_factors, const uint64_t *words, int64_t n_words) try {
    if (!out) return fail(GS_E_INVALID, "out is NULL");
    *out = nullptr;
    if (kind < GS_BLOOM_XOR || kind > GS_BLOOM_BLOCKED) return fail(GS_E_INVALID, "unknown bloom kind");
    if (bits < 1 || !hash_factors || !words || n_words < 1) return fail(GS_E_INVALID, "bad bloom arrays");
    if (kind != GS_BLOOM_BLOCKED) {
        if (n_hashes < 1 || n_hashes > 64) return fail(GS_E_INVALID, "n_hashes must be in [1,64]");
        if (n_words < (bits + 63) / 64) return fail(GS_E_INVALID, "words shorter than bits");
        if (bits > ((int64_t)1 << 37)) return fail(GS_E_UNSUPPORTED, "filters above 2^37 bits (16 GiB) are not supported");
    } else {
        n_hashes = 1;
        if (n_words < bits + 17) return fail(GS_E_INVALID, "blocked filter needs buckets+17 words");
    }
    int rc = use_device(device);
    if (rc) return rc;
    gs_bloom *b = new gs_bloom();
    b->device = device;
    b->kind = kind;
    b->bits = bits;
    b->n_hashes = n_hashes;
    b->n_words = n_words;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) b->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_words, sizeof(u64) * (size_t)n_words);
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_factors, sizeof(int64_t) * (size_t)n_hashes);
    if (e == hipSuccess) e = hipMemcpy(b->d_words, words, sizeof(u64) * (size_t)n_words, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(b->d_factors, hash_factors, sizeof(int64_t) * (size_t)n_hashes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        gs_bloom_destroy(b);
        return fail(e == hipErrorOutOfMemory ? GS_E_NOMEM : GS_E_HIP, std::string("bloom upload: ") + hipGetErrorString(e));
    }
    *out = b;
    return GS_OK;
} catch (const std::bad_alloc &) {
    return fail(GS_E_NOMEM, "out of host memory");
} catch (const std::exception &e) {  // (nothing may leave through the C ABI)
    return fail(GS_E_INVALID, std::string("unexpected exception: ") + e.what());
}

extern "C" hipError_t gs_launch_bloom_xor_put(const int64_t *keys, int64_t n, int64_t bits, const int64_t *factors, int n_hashes, int murmur,
                                              u64 *words, hipStream_t stream);

// java.util.Random (the hash factors of the reference's filters are its first nextLong() values for seed 42,
// C/bloom/AbstractKMerBloomFilter.java:78,105-109)
struct JavaRandom {
    uint64_t s;
    explicit JavaRandom(int64_t seed) : s(((uint64_t)seed ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1)) {}
    int32_t next(int bits) {
        s = (s * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
        return (int32_t)(int64_t)(s >> (48 - bits));
    }
    int64_t next_long() {
        const int64_t hi = (int64_t)next(32), lo = (int64_t)next(32);
        return (int64_t)(((uint64_t)hi << 32) + (uint64_t)lo);
    }
};

extern "C" int gs_bloom_build(gs_bloom **out, int device, int kind, const int64_t *kmers, int64_t n_kmers, int mem, int64_t expected_insertions,
                              double fpp) try {
    if (!out) return fail(GS_E_INVALID, "out is NULL");
    *out = nullptr;
    if (kind != GS_BLOOM_XOR && kind != GS_BLOOM_MURMUR)
        return fail(GS_E_UNSUPPORTED, "gs_bloom_build makes XOR and Murmur filters (the reference's index filters); blocked filters: gs_bloom_create");
    if (n_kmers < 0 || (n_kmers > 0 && !kmers) || expected_insertions < 1 || !(fpp > 0.0 && fpp < 1.0)) return fail(GS_E_INVALID, "bad argument");
    if (mem != GS_MEM_HOST && mem != GS_MEM_DEVICE) return fail(GS_E_INVALID, "mem must be GS_MEM_HOST or GS_MEM_DEVICE");
    // AbstractKMerBloomFilter.optimalNumOfBits :183-185, optimalNumOfHashFunctions :172-174 (Java double arithmetic)
    const double dbits = -(double)expected_insertions * std::log(fpp) / (std::log(2.0) * std::log(2.0));
    int64_t bits = (int64_t)dbits;
    if (bits < 1) bits = 1;
    if (bits > ((int64_t)1 << 37)) return fail(GS_E_UNSUPPORTED, "filters above 2^37 bits (16 GiB) are not supported");
    int64_t nh = (int64_t)std::floor((double)bits / (double)expected_insertions * std::log(2.0) + 0.5);  // Math.round
    if (nh < 1) nh = 1;
    if (nh > 64) return fail(GS_E_UNSUPPORTED, "more than 64 hash functions");
    std::vector<int64_t> factors((size_t)nh);
    JavaRandom rnd(42);
    for (int64_t i = 0; i < nh; i++) factors[(size_t)i] = rnd.next_long();
    int rc = use_device(device);
    if (rc) return rc;
    gs_bloom *b = new gs_bloom();
    b->device = device;
    b->kind = kind;
    b->bits = bits;
    b->n_hashes = (int32_t)nh;
    b->n_words = (bits + 63) / 64;
    hipDeviceProp_t prop;
    int64_t *d_keys = nullptr;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) b->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_words, sizeof(u64) * (size_t)b->n_words);
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_factors, sizeof(int64_t) * (size_t)nh);
    if (e == hipSuccess) e = hipMemsetAsync(b->d_words, 0, sizeof(u64) * (size_t)b->n_words, b->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(b->d_factors, factors.data(), sizeof(int64_t) * (size_t)nh, hipMemcpyHostToDevice, b->stream);
    const int64_t *keys = kmers;
    if (e == hipSuccess && mem == GS_MEM_HOST && n_kmers > 0) {
        e = hipMalloc((void **)&d_keys, sizeof(int64_t) * (size_t)n_kmers);
        if (e == hipSuccess) e = hipMemcpyAsync(d_keys, kmers, sizeof(int64_t) * (size_t)n_kmers, hipMemcpyHostToDevice, b->stream);
        keys = d_keys;
    }
    if (e == hipSuccess) e = gs_launch_bloom_xor_put(keys, n_kmers, bits, b->d_factors, (int)nh, kind == GS_BLOOM_MURMUR, b->d_words, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    hipFree(d_keys);
    if (e != hipSuccess) {
        gs_bloom_destroy(b);
        return fail(e == hipErrorOutOfMemory ? GS_E_NOMEM : GS_E_HIP, std::string("gs_bloom_build: ") + hipGetErrorString(e));
    }
    *out = b;
    return GS_OK;
}
GS_API_CATCH

// geometry and contents of a filter (tests, and hosts that keep the filter in the reference's own object): words may be NULL
extern "C" int gs_bloom_get(gs_bloom *b, int64_t *bits, int32_t *n_hashes, int64_t *hash_factors, uint64_t *words, int64_t n_words) {
    if (!b) return fail(GS_E_INVALID, "bloom is NULL");
    HIP_TRY(hipSetDevice(b->device));
    if (bits) *bits = b->bits;
    if (n_hashes) *n_hashes = b->n_hashes;
    if (hash_factors) HIP_TRY(hipMemcpy(hash_factors, b->d_factors, sizeof(int64_t) * (size_t)b->n_hashes, hipMemcpyDeviceToHost));
    if (words) {
        if (n_words < b->n_words) return fail(GS_E_INVALID, "words is shorter than the filter");
        HIP_TRY(hipMemcpy(words, b->d_words, sizeof(u64) * (size_t)b->n_words, hipMemcpyDeviceToHost));
    }
    return GS_OK;
}

extern "C" int gs_bloom_destroy(gs_bloom *b) {
    if (!b) return GS_OK;
    hipSetDevice(b->device);
    if (b->stream) hipStreamSynchronize(b->stream);
    for (auto &p : b->pending) {
        hipEventDestroy(p.first);
        hipEventDestroy(p.second);
    }
    hipFree(b->d_words);
    hipFree(b->d_factors);
    hipFree(b->d_seq);
    hipFree(b->d_off);
    hipFree(b->d_accept);
    text_free(b->text);
    if (b->stream) hipStreamDestroy(b->stream);
    delete b;
    return GS_OK;
}

static int bloom_collect(gs_bloom *b) {
    for (auto &p : b->pending) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, p.first, p.second));
        b->total_ms += ms;
        b->launches++;
        hipEventDestroy(p.first);
        hipEventDestroy(p.second);
    }
    b->pending.clear();
    return GS_OK;
}

// signed magic for floor-free truncated division by a positive 63-bit constant d:
//   q = mulhi_u64(|v|, magic) >> shift  is exact for |v| < 2^63 when magic = ceil(2^(64+shift) / d)
// computed with 128-bit arithmetic on the host (see gs_filter_kernel for the use).
static void magic_u64(u64 d, u64 &magic, int &shift) {
    // smallest l with 2^l >= d ; magic = floor(2^64 * (2^l - d) / d) + 1 ; q = (mulhi(n,magic) + ((n - mulhi)>>1)) >> (l-1)
    int l = 0;
    while (l < 64 && ((u64)1 << l) < d) l++;
    unsigned __int128 num = ((unsigned __int128)((l == 64 ? 0 : ((u64)1 << l)) - d)) << 64;
    magic = (u64)(num / d) + 1;
    shift = l;
}

static int filter_launch(gs_bloom *b, int k, int min_pos_count, double positive_ratio, const uint8_t *d_seq,
                         const uint64_t *d_off, int64_t n_reads, uint8_t *d_acc, int off_stride, const uint32_t *d_skip,
                         int profile) {
    GsFilterParams P{};
    P.kind = b->kind;
    P.k = k;
    P.min_pos_count = min_pos_count;
    P.positive_ratio = positive_ratio;
    P.bits = (u64)b->bits;
    P.n_hashes = b->n_hashes;
    P.words = b->d_words;
    P.factors = b->d_factors;
    {
        u64 mg = 0;
        int sh = 0;
        magic_u64((u64)b->bits, mg, sh);
        P.magic = mg;
        P.magic_shift = sh;
    }
    P.seq = d_seq;
    P.off = d_off;
    P.n_reads = n_reads;
    P.accept = d_acc;
    P.off_stride = off_stride;
    P.skip = d_skip;
    int occ = gs_filter_occupancy();
    if (occ < 1) occ = 1;
    if (const char *ev = getenv("GS_FILTER_BLOCKS_PER_CU")) {
        const int v = atoi(ev);
        if (v >= 1 && v <= 16) occ = v;
    }
    int grid = (int)std::min<int64_t>((int64_t)b->n_cu * occ, (n_reads + 3) / 4);
    if (grid < 1) grid = 1;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (profile) {
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, b->stream));
    }
    HIP_TRY(gs_launch_filter(&P, grid, b->stream));
    if (profile) {
        HIP_TRY(hipEventRecord(e1, b->stream));
        b->pending.push_back({e0, e1});
    }
    return GS_OK;
}

extern "C" int gs_filter_submit(gs_bloom *b, int k, int min_pos_count, double positive_ratio, const uint8_t *seq,
                                const uint64_t *offsets, int64_t n_reads, int mem, uint8_t *accept, int profile) {
    if (!b) return fail(GS_E_INVALID, "bloom is NULL");
    if (k < 1 || k > 31) return fail(GS_E_INVALID, "k must be in [1,31]");
    if (n_reads < 0 || (n_reads > 0 && (!seq || !offsets || !accept))) return fail(GS_E_INVALID, "bad batch arrays");
    if (n_reads == 0) return GS_OK;
    HIP_TRY(hipSetDevice(b->device));
    const uint8_t *d_seq = seq;
    const uint64_t *d_off = offsets;
    uint8_t *d_acc = accept;
    std::vector<uint64_t> rel;
    if (mem == GS_MEM_HOST) {
        const size_t nbytes = (size_t)(offsets[n_reads] - offsets[0]);
        if (b->seq_cap < nbytes + 1) {
            HIP_TRY(hipStreamSynchronize(b->stream));
            hipFree(b->d_seq);
            b->d_seq = nullptr;
            b->seq_cap = 0;
            HIP_TRY(hipMalloc((void **)&b->d_seq, nbytes + 1));
            b->seq_cap = nbytes + 1;
        }
        if (b->reads_cap < (size_t)n_reads) {
            HIP_TRY(hipStreamSynchronize(b->stream));
            hipFree(b->d_off);
            hipFree(b->d_accept);
            b->d_off = nullptr;
            b->d_accept = nullptr;
            b->reads_cap = 0;
            HIP_TRY(hipMalloc((void **)&b->d_off, sizeof(uint64_t) * ((size_t)n_reads + 1)));
            HIP_TRY(hipMalloc((void **)&b->d_accept, (size_t)n_reads));
            b->reads_cap = (size_t)n_reads;
        }
        const uint64_t *hoff = offsets;
        if (offsets[0] != 0) {
            rel.resize((size_t)n_reads + 1);
            for (int64_t i = 0; i <= n_reads; i++) rel[(size_t)i] = offsets[i] - offsets[0];
            hoff = rel.data();
        }
        HIP_TRY(hipMemcpyAsync(b->d_seq, seq + offsets[0], nbytes, hipMemcpyHostToDevice, b->stream));
        HIP_TRY(hipMemcpyAsync(b->d_off, hoff, sizeof(uint64_t) * ((size_t)n_reads + 1), hipMemcpyHostToDevice, b->stream));
        d_seq = b->d_seq;
        d_off = b->d_off;
        d_acc = b->d_accept;
    } else if (mem != GS_MEM_DEVICE)
        return fail(GS_E_INVALID, "mem must be GS_MEM_HOST or GS_MEM_DEVICE");

    int rc = filter_launch(b, k, min_pos_count, positive_ratio, d_seq, d_off, n_reads, d_acc, 1, nullptr, profile);
    if (rc) return rc;
    if (mem == GS_MEM_HOST) {
        HIP_TRY(hipMemcpyAsync(accept, b->d_accept, (size_t)n_reads, hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));
        return bloom_collect(b);
    }
    return GS_OK;
}

extern "C" int gs_filter_sync(gs_bloom *b) {
    if (!b) return fail(GS_E_INVALID, "bloom is NULL");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return bloom_collect(b);
}

// ---- text mode of the filter (see gs_match_submit_text) ----
static int filter_submit_text(gs_bloom *b, int k, int min_pos_count, double positive_ratio, const uint8_t *text, int64_t n_bytes,
                              int64_t n_lines, int mem, uint8_t *accept, uint32_t *newlines, int profile, int64_t *ticket,
                              int64_t fasta_records, int64_t *ml_out) {
    if (!b) return fail(GS_E_INVALID, "bloom is NULL");
    if (k < 1 || k > 31) return fail(GS_E_INVALID, "k must be in [1,31]");
    if (n_lines > 0 && !accept) return fail(GS_E_INVALID, "accept is NULL");
    HIP_TRY(hipSetDevice(b->device));
    const bool fasta = fasta_records >= 0 || ml_out != nullptr;  // the reads are gathered, not in place
    if (mem != GS_MEM_HOST && mem != GS_MEM_DEVICE && mem != GS_MEM_DEVICE_TEXT) return fail(GS_E_INVALID, "bad mem");
    int rc = text_submit(b->text, b->stream, text, n_bytes, n_lines, mem == GS_MEM_HOST ? GS_MEM_HOST : GS_MEM_DEVICE, k, ticket, fasta_records, ml_out);
    if (rc) return rc;
    const int64_t n_reads = ml_out ? std::max<int64_t>(ml_out[0], 0) : (fasta ? fasta_records : (n_lines >> 2));
    if (n_reads == 0) return GS_OK;
    const bool dev_out = mem == GS_MEM_DEVICE;  // (GS_MEM_DEVICE_TEXT: the text is in HBM, accept / newlines are host arrays)
    if (!dev_out && b->reads_cap < (size_t)n_reads) {
        HIP_TRY(hipStreamSynchronize(b->stream));
        hipFree(b->d_off);
        hipFree(b->d_accept);
        b->d_off = nullptr;
        b->d_accept = nullptr;
        b->reads_cap = 0;
        HIP_TRY(hipMalloc((void **)&b->d_off, sizeof(uint64_t) * ((size_t)n_reads + 1)));
        HIP_TRY(hipMalloc((void **)&b->d_accept, (size_t)n_reads));
        b->reads_cap = (size_t)n_reads;
    }
    uint8_t *d_acc = dev_out ? accept : b->d_accept;
    // a refused chunk leaves `accept` untouched: zero it so that stale flags never look like results
    HIP_TRY(hipMemsetAsync(d_acc, 0, (size_t)n_reads, b->stream));
    rc = filter_launch(b, k, min_pos_count, positive_ratio, fasta ? b->text.d_fa_seq : b->text.d_text, (const uint64_t *)b->text.d_off2,
                       n_reads, d_acc, fasta ? 1 : 2, b->text.d_status + (size_t)b->text.bank * GS_TS_WORDS + GS_TS_SKIP, profile);
    if (rc) return rc;
    if ((rc = text_touched(b->text, b->stream))) return rc;
    if (!fasta) b->text.d_last_flags = d_acc;
    const hipMemcpyKind kind = dev_out ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (!dev_out) HIP_TRY(hipMemcpyAsync(accept, b->d_accept, (size_t)n_reads, kind, b->stream));
    if (newlines) HIP_TRY(hipMemcpyAsync(newlines, b->text.d_nl, sizeof(uint32_t) * (size_t)n_lines, kind, b->stream));
    return GS_OK;
}

extern "C" int gs_filter_submit_text(gs_bloom *b, int k, int min_pos_count, double positive_ratio, const uint8_t *text,
                                     int64_t n_bytes, int64_t n_lines, int mem, uint8_t *accept, uint32_t *newlines,
                                     int profile, int64_t *ticket) {
    return filter_submit_text(b, k, min_pos_count, positive_ratio, text, n_bytes, n_lines, mem, accept, newlines, profile, ticket, -1, nullptr);
}

extern "C" int gs_filter_submit_fasta(gs_bloom *b, int k, int min_pos_count, double positive_ratio, const uint8_t *text,
                                      int64_t n_bytes, int64_t n_lines, int64_t n_records, int mem, uint8_t *accept,
                                      uint32_t *newlines, int64_t *ticket) {
    if (n_records < 0) return fail(GS_E_INVALID, "n_records < 0");
    return filter_submit_text(b, k, min_pos_count, positive_ratio, text, n_bytes, n_lines, mem, accept, newlines, 0, ticket, n_records, nullptr);
}

extern "C" int gs_filter_submit_fastq_ml(gs_bloom *b, int k, int min_pos_count, double positive_ratio, const uint8_t *text,
                                         int64_t n_bytes, int64_t n_lines, int mem, uint8_t *accept, uint32_t *newlines,
                                         int64_t *n_records, int64_t *consumed_bytes, int64_t *consumed_lines, int64_t *ticket) {
    if (!n_records || !consumed_bytes) return fail(GS_E_INVALID, "NULL argument");
    int64_t out[3] = {0, 0, 0};
    const int rc = filter_submit_text(b, k, min_pos_count, positive_ratio, text, n_bytes, n_lines, mem, accept, newlines, 0, ticket, -1, out);
    *n_records = out[0];
    *consumed_bytes = out[1];
    if (consumed_lines) *consumed_lines = out[2];
    return rc;
}

// nextEntry's rewriteInput on the device (FastqBloomFilter.java:92-105): the accepted (which != 0) or the other records of the last chunk
extern "C" int gs_filter_compact_text(gs_bloom *b, int which, int with_probs, int slot, const uint8_t **d_out, int64_t *n_bytes, int64_t *n_records) {
    if (!b) return fail(GS_E_INVALID, "bloom is NULL");
    HIP_TRY(hipSetDevice(b->device));
    return text_compact(b->text, b->stream, 0xff, which != 0, which != 0, slot, with_probs, d_out, n_bytes, n_records);
}

extern "C" int gs_filter_text_read_bounds(gs_bloom *b, uint64_t *bounds) {
    if (!b || !bounds) return fail(GS_E_INVALID, "NULL argument");
    if (b->text.tickets == 0) return fail(GS_E_STATE, "no text chunk has been submitted");
    if (!b->text.last_fasta) return fail(GS_E_STATE, "the last chunk was four-line FASTQ: its reads lie in the text (newlines)");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipMemcpyAsync(bounds, b->text.d_off2, sizeof(uint64_t) * ((size_t)b->text.last_reads + 1), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return bloom_collect(b);
}

extern "C" int gs_filter_text_line_classes(gs_bloom *b, uint8_t *classes) {
    if (!b || !classes) return fail(GS_E_INVALID, "NULL argument");
    if (b->text.tickets == 0 || !b->text.d_ml_class) return fail(GS_E_STATE, "no general FASTQ chunk has been submitted");
    HIP_TRY(hipSetDevice(b->device));
    if (b->text.last_lines > 0)
        HIP_TRY(hipMemcpyAsync(classes, b->text.d_ml_class, (size_t)b->text.last_lines, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return bloom_collect(b);
}

extern "C" int gs_filter_get_device(gs_bloom *b, int *device) {
    if (!b || !device) return fail(GS_E_INVALID, "NULL argument");
    *device = b->device;
    return GS_OK;
}

extern "C" int gs_filter_text_wait_copy(gs_bloom *b, int64_t ticket) {
    if (!b) return fail(GS_E_INVALID, "bloom is NULL");
    HIP_TRY(hipSetDevice(b->device));
    return text_wait_copy(b->text, ticket);
}

extern "C" int gs_filter_text_status(gs_bloom *b, int64_t *failed_ticket, int64_t *first_bad_record, int64_t totals[3]) {
    if (!b) return fail(GS_E_INVALID, "bloom is NULL");
    HIP_TRY(hipSetDevice(b->device));
    int rc = text_status(b->text, b->stream, failed_ticket, first_bad_record, totals);
    if (rc) return rc;
    return bloom_collect(b);
}

extern "C" int gs_filter_text_reset(gs_bloom *b, int clear_totals) {
    if (!b) return fail(GS_E_INVALID, "bloom is NULL");
    HIP_TRY(hipSetDevice(b->device));
    return text_reset(b->text, clear_totals != 0, b->stream);
}

extern "C" int gs_filter_kernel_time(gs_bloom *b, int64_t *launches, double *total_ms) {
    int rc = gs_filter_sync(b);
    if (rc) return rc;
    if (launches) *launches = b->launches;
    if (total_ms) *total_ms = b->total_ms;
    return GS_OK;
}
