// orr_api.hip -- the C ABI of libomnirecall_hip.so (include/omnirecall_hip.h):
// corpus shard management, search orchestration on the index's HIP stream, and
// the host-side exact finish of the k' survivors.
//
// Replaces RecallSearchService.cs:26-37 (GetRecentChunksAsync + Select(ScoreChunk)
// + OrderByDescending/ThenByDescending/Take) for a batch of queries.  There is NO
// CPU scoring path here: without a gfx950 device every compute entry point
// returns ORR_EDEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <chrono>
#include <pthread.h>
#include <dlfcn.h>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <cstring>
#include <limits>
#include <mutex>
#include <shared_mutex>
#include <deque>
#include <new>
#include <numeric>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "orr_kernels.h"
#include "orr_token_index.h"

namespace {

thread_local std::string g_last_error = "";

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                 \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            return fail(_e == hipErrorOutOfMemory ? ORR_ENOMEM : ORR_EDEVICE, "%s failed: %s (%s:%d)", \
                        #expr, hipGetErrorString(_e), __FILE__, __LINE__);                            \
    } while (0)

#define ORR_TRY(expr)          \
    do {                       \
        int _r = (expr);       \
        if (_r != ORR_OK) return _r; \
    } while (0)

// Grow-only device buffer.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return ORR_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(ORR_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        }
        cap = want;
        return ORR_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
    }
    template <typename T> T *as() const { return static_cast<T *>(p); }
};

// Grow-only pinned host buffer (device-visible): small uploads, the query copy, and the
// candidate records the last kernel writes straight into host memory.
struct PinnedBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return ORR_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(ORR_ENOMEM, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        }
        cap = want;
        return ORR_OK;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
    }
    template <typename T> T *as() const { return static_cast<T *>(p); }
};

struct KernelStat {
    std::string name;
    int64_t launches = 0;
    double total_ms = 0.0;
    double algo_bytes = 0.0;
};

struct PendingEvent {
    int stat;
    hipEvent_t start, stop;
};

}  // namespace

struct orr_index {
    int device = 0;
    int32_t dim = 0;
    int64_t row_base = 0;
    hipStream_t stream = nullptr;      // main stream: dots, fused score, selection
    hipStream_t stream_kw = nullptr;   // keyword scan runs beside the HBM-bound dot kernel
    hipStream_t stream_aux = nullptr;  // what must not sit in front of the keyword chain: the clearing of the term bitmaps behind a
                                       // search (0.1 ms at 10M rows x 256 queries) and the norms of device-resident queries
    hipEvent_t ev_bm_clean = nullptr;  // ... recorded behind that clearing; the next search's posting expansion waits for it
    bool bm_clean_pending = false;
    bool kw_counters_clean = false;    // the hit counter and the per-term hit counts were zeroed behind the last search
    const void *kw_counters_of[2] = {nullptr, nullptr};
    hipEvent_t ev_inputs = nullptr, ev_kw_done = nullptr, ev_main_ready = nullptr, ev_range[15] = {};     // (one per row range but the first: 16 ranges at most)
    std::mutex mu;

    // corpus, in append order until seal, in candidate order afterwards
    int64_t n_rows = 0, cap_rows = 0;
    float *d_emb = nullptr;
    int64_t *d_created = nullptr;
    int64_t *d_row_ids = nullptr;
    // lowercased content: row r = d_pool[d_cstart[r] .. +d_clen[r]), starts 16-byte aligned,
    // each row followed by 1..16 spaces (see keyword_scan_kernel)
    uint64_t *d_cstart = nullptr;      // [cap_rows]
    uint32_t *d_clen = nullptr;        // [cap_rows]
    uint8_t *d_pool = nullptr;
    uint64_t pool_len = 0, pool_cap = 0;
    double *d_norm_b = nullptr;
    // after seal the raw content is replaced by the token index (orr_token_index.cpp)
    int64_t n_tokens = 0;
    uint8_t *d_vpool = nullptr;        // vocabulary in the scan kernel's row layout
    uint64_t *d_vstart = nullptr;      // [n_tokens]
    uint32_t *d_vlen = nullptr;        // [n_tokens]
    uint64_t *d_post_off = nullptr;    // [n_tokens+1]
    uint32_t *d_post_rows = nullptr;   // ascending candidate positions per token
    uint64_t n_postings = 0;
    // vocabulary tokens longer than 16 bytes (URLs and the like) go through the wave-per-token scan; the rest is
    // matched one lane per token.  Built at the first search with terms (ensure_vlong); -1 = not yet.
    int64_t n_vlong = -1;
    int64_t n_vmid = 0;                // the first n_vmid entries of the list are the tokens of 17..32 bytes (one lane per token too)
    DevBuf vlong_start, vlong_len, vlong_id;
    // stored row bitmaps of the FREQUENT vocabulary tokens (a posting list of at least rows / 64 entries: the bitmap is at most
    // twice its bytes), built at the first search with terms over a large shard (ensure_token_bitmaps): a query term whose only
    // hit is such a token uses the stored bitmap as it is instead of expanding the posting list again for every batch
    int64_t n_tok_bm = -1;             // -1: not looked at yet; 0: none
    int64_t tok_bm_words = 0;
    DevBuf tok_bm, tok_bm_index;       // [n_tok_bm][tok_bm_words] u32; [n_tokens] int32 bitmap number or -1
    std::vector<int64_t> h_created;    // host mirror (seal-time ordering)
    std::vector<uint32_t> h_clen;      // host mirror of content lengths
    std::vector<uint64_t> h_cprefix;   // after seal: bytes of content in rows [0, r)
    std::vector<double> h_norm_a;      // exact query norms of the batch in flight (run_shard -> host finish)
    std::vector<uint32_t> h_survivors; // two-stage pass: (query,row) pairs the screen kept, per query of the batch in flight (else empty)
    uint32_t kw_hits_cap = 16u << 20;  // (term, token) matches the hit list of the keyword chain holds; grows to the measured count when a batch exceeds it
    uint32_t survivor_cap = 8192;      // entries per query of the survivors' buffers; grows when a query overflows it (clustered corpora)
    uint32_t pass_cap = 8192;          // what the pass in flight uses: survivor_cap, halved until a batch's buffers stay below 2 GiB
    orr_search_stats sstats{};         // orr_index_search_stats
    bool sealed = false;
    bool is_view = false;              // a second search lane over another index's sealed corpus (orr_index_view)
    bool opt_fuse_epilogue = false;
    int opt_two_stage = 1;             // 0 off, 1 on (bf16 shadow when it fits), 2 on without the shadow
    int opt_shard_pass = 0;            // orr_search_shard: 0 the library picks the pass, 1 unfused batched pass, 2 exact pass
    int opt_shard_topk = 0;            // orr_search_shard: the caller's topK when > 0 (the two-stage floor then comes from the k-th best, not the k'-th)
    int sample_boost = 1;              // two-stage pass: the sampled prefix is this many times the default (1..16), steered by the survivors measured
    DevBuf emb_shadow;                 // bf16(E), [n_rows][dim]: operand of the screening GEMM (two-stage pass)
    bool shadow_ready = false, shadow_failed = false;
    DevBuf emb_i8, i8_scale, i8_rel_err, i8_rel_hat, i8_rowf;   // int8 shadow: streaming screen of 1..4 queries (K2i), screening GEMM (K2j)
    bool i8_ready = false, i8_failed = false;
    // deleted rows (orr_index_delete_rows): ascending positions, mirrored on the device for the record flags
    std::vector<int64_t> dead;
    DevBuf d_dead;
    int64_t dead_before = 0;           // deleted rows in the shards in front of this one ("dead_rows_before")
    const orr_index *parent = nullptr; // views: the deleted set lives in the owning index
    std::vector<std::pair<int64_t, int64_t>> id_index;   // (row id, position) ascending, built at the first delete

    // search workspace
    DevBuf ws_q, ws_dot, ws_dotf, ws_sel, ws_cand, ws_qc, ws_rowc, ws_tau, ws_qsplit, ws_fcnt, ws_fbuf, ws_fqf, ws_fany, ws_tsL, ws_tskey, ws_qtiled, ws_fdot, ws_pbuf, ws_psel, ws_q8, ws_q8s1, ws_q8err, ws_zero, ws_norm_a;
    DevBuf ws_keys_a, ws_keys_b, ws_vals_a, ws_vals_b, ws_sort_tmp, ws_raw, ws_src_start, ws_qsub;
    DevBuf ws_vmatch, ws_bitmaps, ws_hits, ws_counter, ws_meta, ws_tickets, ws_kwalias;
    size_t bitmaps_clean = 0;          // leading bytes of ws_bitmaps known to be zero (cleared again behind every search)
    const void *bitmaps_clean_of = nullptr;
    PinnedBuf pin_meta, pin_q, pin_qc, pin_cand, pin_norm, pin_cnt, pin_kwcnt;
    hipEvent_t ev_q = nullptr;

    // search lanes (owning index only): concurrent searches on ONE handle each take a lane -- the index itself, or one of up
    // to max_lanes - 1 internal views (own streams and workspaces, shared corpus and shadows) created when first needed
    std::mutex lanes_mu;
    std::condition_variable lanes_cv;
    std::vector<orr_index *> lanes;        // internal views (lane 0 is the index itself)
    std::vector<char> lane_busy;           // [1 + lanes.size()]; a reserved slot whose view is still being made counts as busy
    bool self_busy = false;
    int lanes_reserved = 0;                // views being created right now
    int max_lanes = 4;
    bool internal_lane = false;            // this view belongs to its parent's lane pool (not handed to the caller)
    int64_t dead_before_pub = 0, dead_count_pub = 0;   // (under lanes_mu) copies of dead_before / dead.size() a cluster reads without waiting for searches
    uint32_t survivor_cap_hint = 0;        // (under lanes_mu) a lane measured that the survivors' buffers must be at least this large
    std::atomic<int> user_views{0};        // views handed to the caller (orr_index_view) that are still alive: they pin the shard's layout

    // profiling
    int profiling = 0;                     // 0 off, 1 every kernel, 2 only the pass over all rows (the kernel a roofline is quoted on)
    std::vector<KernelStat> stats;
    std::vector<PendingEvent> pending;
    std::vector<hipEvent_t> event_pool;
};

static int make_view(orr_index *parent, orr_index **out, bool internal);

namespace {

// ---- search lanes ------------------------------------------------------------------------------------------------
// The request path is concurrent by nature (RecallSearchService is scoped, one instance per request, Program.cs:59; the store
// behind it is lock-free, InMemoryIngestionStore.cs:8-9).  A search on an OWNING index takes a free lane: the index's own
// workspaces, or those of an internal view (created on demand, at most max_lanes - 1 of them; corpus and shadows are shared,
// nothing is copied).  Searches from different threads on one handle then run side by side; the caller never sees a view.
// A view handle the caller made itself (orr_index_view) is its own single lane, as before.
struct Lane {
    orr_index *owner = nullptr;
    orr_index *lane = nullptr;
    int slot = -1;                                     // 0: the index itself; i >= 1: owner->lanes[i - 1]
    Lane() = default;
    Lane(const Lane &) = delete;
    Lane &operator=(const Lane &) = delete;
    Lane(Lane &&o) noexcept : owner(o.owner), lane(o.lane), slot(o.slot) { o.owner = nullptr; o.lane = nullptr; o.slot = -1; }
    int acquire(orr_index *idx)
    {
        if (idx->is_view) { lane = idx; return ORR_OK; }
        owner = idx;
        std::unique_lock<std::mutex> lk(idx->lanes_mu);
        for (;;) {
            auto take = [&](orr_index *l, int sl) {
                slot = sl; lane = l;
                if (idx->survivor_cap_hint > l->survivor_cap) l->survivor_cap = idx->survivor_cap_hint;   // what another lane measured
                return ORR_OK;
            };
            if (!idx->self_busy) { idx->self_busy = true; return take(idx, 0); }
            for (size_t i = 0; i < idx->lanes.size(); ++i)
                if (idx->lanes[i] && !idx->lane_busy[i]) { idx->lane_busy[i] = 1; return take(idx->lanes[i], (int)i + 1); }
            if (idx->sealed && (int)idx->lanes.size() + 1 < idx->max_lanes) {
                const size_t i = idx->lanes.size();
                idx->lanes.push_back(nullptr);                     // the slot is reserved (and busy) while its view is made
                idx->lane_busy.push_back(1);
                lk.unlock();
                orr_index *v = nullptr;
                const int r = make_view(idx, &v, true);            // (waits for the search that holds the index's own lane)
                lk.lock();
                if (r == ORR_OK) {
                    idx->lanes[i] = v; slot = (int)i + 1; lane = v;
                    return ORR_OK;
                }
                // no room for another set of workspaces (or the index is being torn down): make do with the lanes there are
                idx->lane_busy[i] = 0;
                idx->max_lanes = 1;
                for (orr_index *l : idx->lanes) if (l) ++idx->max_lanes;
                continue;
            }
            idx->lanes_cv.wait(lk);
        }
    }
    void release()
    {
        if (!owner) { lane = nullptr; return; }
        {
            std::lock_guard<std::mutex> lk(owner->lanes_mu);
            if (slot == 0) owner->self_busy = false;
            else if (slot > 0) owner->lane_busy[(size_t)slot - 1] = 0;
        }
        owner->lanes_cv.notify_all();
        owner = nullptr; lane = nullptr; slot = -1;
    }
    ~Lane() { release(); }
};

// Everything that changes what the lanes share or reads their counters (deletes, options, statistics, save, destroy) waits until
// no search is in flight on any lane and keeps new ones out meanwhile.
struct AllLanes {
    orr_index *owner = nullptr;
    explicit AllLanes(orr_index *idx)
    {
        if (!idx || idx->is_view) return;
        owner = idx;
        std::unique_lock<std::mutex> lk(idx->lanes_mu);
        idx->lanes_cv.wait(lk, [idx] {
            if (idx->self_busy) return false;
            for (size_t i = 0; i < idx->lanes.size(); ++i) if (idx->lane_busy[i]) return false;
            return true;
        });
        idx->self_busy = true;
        for (size_t i = 0; i < idx->lanes.size(); ++i) idx->lane_busy[i] = 1;
    }
    ~AllLanes()
    {
        if (!owner) return;
        {
            std::lock_guard<std::mutex> lk(owner->lanes_mu);
            owner->self_busy = false;
            for (size_t i = 0; i < owner->lanes.size(); ++i) owner->lane_busy[i] = 0;
        }
        owner->lanes_cv.notify_all();
    }
    AllLanes(const AllLanes &) = delete;
    AllLanes &operator=(const AllLanes &) = delete;
};

int stat_slot(orr_index *idx, const char *name)
{
    for (size_t i = 0; i < idx->stats.size(); ++i)
        if (idx->stats[i].name == name) return (int)i;
    KernelStat s;
    s.name = name;
    idx->stats.push_back(s);
    return (int)idx->stats.size() - 1;
}

hipEvent_t take_event(orr_index *idx)
{
    if (!idx->event_pool.empty()) {
        hipEvent_t e = idx->event_pool.back();
        idx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

// Brackets one launch with events on the index's stream when profiling is on.
struct Timed {
    // the one launch per search that streams every row (an event pair costs a few microseconds of stream time each, which a
    // one-query search notices: level 2 times just this kernel)
    static bool over_all_rows(const char *name)
    {
        return (strncmp(name, "screen_", 7) == 0 && !strstr(name, "prefix")) || strncmp(name, "gemm_dot", 8) == 0 ||
               strcmp(name, "dot_exact") == 0 || strcmp(name, "gemv_mfma") == 0;
    }
    orr_index *idx;
    PendingEvent pe;
    bool on;
    hipStream_t st;
    Timed(orr_index *i, const char *name, double algo_bytes, hipStream_t stream = nullptr)
        : idx(i), on(i->profiling == 1 || (i->profiling == 2 && over_all_rows(name))), st(stream ? stream : i->stream)
    {
        if (!on) return;
        pe.stat = stat_slot(idx, name);
        idx->stats[pe.stat].algo_bytes += algo_bytes;
        pe.start = take_event(idx);
        pe.stop = take_event(idx);
        (void)hipEventRecord(pe.start, st);
    }
    ~Timed()
    {
        if (!on) return;
        (void)hipEventRecord(pe.stop, st);
        idx->pending.push_back(pe);
    }
};

// After a stream synchronise: fold the finished event pairs into the counters.
void collect_events(orr_index *idx)
{
    for (auto &pe : idx->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pe.start, pe.stop) == hipSuccess) {
            idx->stats[pe.stat].launches += 1;
            idx->stats[pe.stat].total_ms += (double)ms;
        }
        idx->event_pool.push_back(pe.start);
        idx->event_pool.push_back(pe.stop);
    }
    idx->pending.clear();
}

int bind_device(const orr_index *idx)
{
    HIP_TRY(hipSetDevice(idx->device));
    return ORR_OK;
}

template <typename T> int dev_alloc(T **out, size_t count)
{
    *out = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(out), count * sizeof(T));
    if (e != hipSuccess) {
        *out = nullptr;
        return fail(ORR_ENOMEM, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
    }
    return ORR_OK;
}

// Reallocates a device array keeping the first `keep` elements.
template <typename T> int dev_grow(T **buf, size_t keep, size_t new_count, hipStream_t s)
{
    T *nb = nullptr;
    ORR_TRY(dev_alloc(&nb, new_count));
    if (*buf && keep) {
        hipError_t e = hipMemcpyAsync(nb, *buf, keep * sizeof(T), hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) {
            (void)hipFree(nb);
            return fail(ORR_EDEVICE, "device copy failed: %s", hipGetErrorString(e));
        }
    }
    if (*buf) (void)hipFree(*buf);
    *buf = nb;
    return ORR_OK;
}

int ensure_row_capacity(orr_index *idx, int64_t rows)
{
    if (rows <= idx->cap_rows) return ORR_OK;
    int64_t nc = std::max<int64_t>(rows, idx->cap_rows + idx->cap_rows / 2);
    nc = std::max<int64_t>(nc, 1024);
    if (idx->dim > 0) ORR_TRY(dev_grow(&idx->d_emb, (size_t)idx->n_rows * idx->dim, (size_t)nc * idx->dim, idx->stream));
    ORR_TRY(dev_grow(&idx->d_created, (size_t)idx->n_rows, (size_t)nc, idx->stream));
    ORR_TRY(dev_grow(&idx->d_row_ids, (size_t)idx->n_rows, (size_t)nc, idx->stream));
    ORR_TRY(dev_grow(&idx->d_cstart, (size_t)idx->n_rows, (size_t)nc, idx->stream));
    ORR_TRY(dev_grow(&idx->d_clen, (size_t)idx->n_rows, (size_t)nc, idx->stream));
    idx->cap_rows = nc;
    return ORR_OK;
}

int ensure_pool_capacity(orr_index *idx, uint64_t bytes)
{
    if (bytes + orr::kScanPoolSlack <= idx->pool_cap) return ORR_OK;
    uint64_t nc = std::max<uint64_t>(bytes + orr::kScanPoolSlack, idx->pool_cap + idx->pool_cap / 2);
    nc = std::max<uint64_t>(nc, 1u << 16);
    ORR_TRY(dev_grow(&idx->d_pool, (size_t)idx->pool_len, (size_t)nc, idx->stream));
    idx->pool_cap = nc;
    return ORR_OK;
}

// Exact sum_i (double)fl32(q_i*q_i) in index order: normA of RecallSearchService.cs:80.
double exact_norm(const float *q, int32_t dim)
{
    double acc = 0.0;
    for (int32_t i = 0; i < dim; ++i) {
        float p = q[i] * q[i];
        acc += (double)p;
    }
    return acc;
}

// A few persistent host threads for the per-query work around a batch (exact norms of host-resident queries, the
// host finish): queries are independent, and starting threads per call costs more than the work of a small batch.
// One parallel region at a time; a second caller (another search lane) simply runs its tasks itself.
class HostPool {
public:
    static HostPool &get()
    {
        static HostPool *pool = new HostPool((int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency())) - 1);
        return *pool;                                  // never destroyed: its threads sleep until the process ends
    }
    int width() const { return (int)workers_.size() + 1; }
    void run(int n_tasks, const std::function<void(int)> &fn)
    {
        std::unique_lock<std::mutex> region(region_mu_, std::try_to_lock);
        if (!region.owns_lock() || workers_.empty() || n_tasks <= 1 || forked_.load(std::memory_order_relaxed)) {
            for (int i = 0; i < n_tasks; ++i) fn(i);
            return;
        }
        {
            std::lock_guard<std::mutex> l(mu_);
            fn_ = &fn; n_tasks_ = n_tasks; next_.store(0); active_ = (int)workers_.size(); ++generation_;
        }
        cv_work_.notify_all();
        for (int i; (i = next_.fetch_add(1)) < n_tasks;) fn(i);
        std::unique_lock<std::mutex> l(mu_);
        cv_done_.wait(l, [&] { return active_ == 0; });
    }

private:
    explicit HostPool(int n_workers)
    {
        for (int i = 0; i < n_workers; ++i) workers_.emplace_back([this] { loop(); });
        for (auto &w : workers_) w.detach();
        // a forked child inherits this object but none of its threads: there every region runs in the caller
        pthread_atfork(nullptr, nullptr, [] { forked_.store(true); });
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> l(mu_);
            cv_work_.wait(l, [&] { return generation_ != seen; });
            seen = generation_;
            const std::function<void(int)> *fn = fn_;
            const int n = n_tasks_;
            l.unlock();
            for (int i; (i = next_.fetch_add(1)) < n;) (*fn)(i);
            l.lock();
            if (--active_ == 0) cv_done_.notify_one();
        }
    }
    static inline std::atomic<bool> forked_{false};
    std::mutex region_mu_, mu_;
    std::condition_variable cv_work_, cv_done_;
    std::vector<std::thread> workers_;
    const std::function<void(int)> *fn_ = nullptr;
    int n_tasks_ = 0, active_ = 0;
    std::atomic<int> next_{0};
    uint64_t generation_ = 0;
};

// The same sums for a batch.  Each query's sum is a chain of dependent fp64 additions (the order is
// part of the reference's arithmetic), so eight queries are walked in lock step to keep the adder busy.
void exact_norms_range(const float *q, int32_t B, int32_t dim, double *out);
void exact_norms(const float *q, int32_t B, int32_t dim, double *out)
{
    if ((int64_t)B * dim < (1 << 18)) { exact_norms_range(q, B, dim, out); return; }
    const int32_t groups = (B + 7) / 8;                 // groups of eight queries, shared out over the pool
    HostPool::get().run(groups, [&](int g) {
        const int32_t b0 = g * 8, nb = std::min<int32_t>(8, B - b0);
        exact_norms_range(q + (size_t)b0 * dim, nb, dim, out + b0);
    });
}

void exact_norms_range(const float *q, int32_t B, int32_t dim, double *out)
{
    int32_t b = 0;
    for (; b + 8 <= B; b += 8) {
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const float *r = q + (size_t)b * dim;
        for (int32_t i = 0; i < dim; ++i)
            for (int j = 0; j < 8; ++j) {
                float p = r[(size_t)j * dim + i] * r[(size_t)j * dim + i];
                acc[j] += (double)p;
            }
        for (int j = 0; j < 8; ++j) out[b + j] = acc[j];
    }
    for (; b < B; ++b) out[b] = exact_norm(q + (size_t)b * dim, dim);
}

// double.CompareTo
int compare_double(double a, double b)
{
    if (a < b) return -1;
    if (a > b) return 1;
    if (a == b) return 0;
    if (std::isnan(a)) return std::isnan(b) ? 0 : -1;
    return 1;
}

struct Ranked {
    double score;
    int64_t order_key;
    int64_t row_id;
};

// RecallSearchService.cs:59-67 for one surviving candidate, in the reference's own
// arithmetic on the host (libm exp = what Math.Exp calls), from the exact pieces
// the device produced.
double exact_score(const orr_candidate &c, bool use_cos, double norm_a, int32_t n_terms, int64_t now_ticks)
{
    double cosv = 0.0;
    if (use_cos) {
        if (norm_a <= 0.0 || c.norm_b <= 0.0)
            cosv = 0.0;
        else
            cosv = c.dot / (std::sqrt(norm_a) * std::sqrt(c.norm_b));
    }
    double kw = n_terms > 0 ? (double)c.matches / (double)n_terms : 0.0;
    double total_days = (double)(now_ticks - c.created_ticks) / 864000000000.0;
    double age_days = total_days > 0.0 ? total_days : 0.0;
    double rec = std::exp(-age_days / 30.0);
    return (cosv * 0.7) + (kw * 0.2) + (rec * 0.1);
}

// Absolute slack between the device's selection score and the exact host score
// of the same row: identical IEEE operations except exp (ocml vs libm, a few ulp
// of a value <= 1, times 0.1).
constexpr double kCertifyEps = 1e-13;

}  // namespace

namespace {

// Shard file -> device array whose values index something (orr_index_load): every element is checked on its way through the host buffer
// (a shard file is input: posting rows >= n_rows would make expand_hits write outside the term bitmaps).
template <typename T, typename CHECK>
int read_device_array_checked(FILE *f, T *dptr, size_t count, std::vector<uint8_t> &buf, CHECK ok_run, const char *what)
{
    const size_t per = buf.size() / sizeof(T);
    for (size_t off = 0; off < count; off += per) {
        const size_t m = std::min(per, count - off);
        if (fread(buf.data(), sizeof(T), m, f) != m) return fail(ORR_EINVAL, "shard file is truncated");
        if (!ok_run(reinterpret_cast<const T *>(buf.data()), off, m)) return fail(ORR_EINVAL, "shard file: malformed %s", what);
        HIP_TRY(hipMemcpy(dptr + off, buf.data(), m * sizeof(T), hipMemcpyHostToDevice));
    }
    return ORR_OK;
}

}  // namespace

// the events that release the later row ranges' count words (one per range but the first)
static bool create_range_events(hipEvent_t (&ev)[15])
{
    for (hipEvent_t &e : ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
    return true;
}

extern "C" {

int orr_abi_version(void) { return ORR_ABI_VERSION; }

const char *orr_last_error(void) { return g_last_error.c_str(); }

int orr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

int orr_index_create(const orr_config *cfg, orr_index **out)
{
    if (!cfg || !out) return fail(ORR_EINVAL, "orr_index_create: null argument");
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(orr_config)) return fail(ORR_EINVAL, "orr_config.struct_size mismatch");
    if (cfg->dim < 0 || cfg->capacity_rows < 0 || cfg->row_base < 0) return fail(ORR_EINVAL, "negative size in orr_config");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(ORR_EDEVICE, "no HIP device available: libomnirecall_hip has no CPU path");
    if (cfg->device < 0 || cfg->device >= n_dev) return fail(ORR_EINVAL, "device %d out of range (%d visible)", cfg->device, n_dev);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(ORR_EDEVICE, "device %d is %s; this library carries gfx950 code only", cfg->device, prop.gcnArchName);
    orr_index *idx = new (std::nothrow) orr_index();
    if (!idx) return fail(ORR_ENOMEM, "out of host memory");
    idx->device = cfg->device;
    idx->dim = cfg->dim;
    idx->row_base = cfg->row_base;
    if (hipSetDevice(idx->device) != hipSuccess || hipStreamCreateWithFlags(&idx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&idx->stream_kw, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&idx->stream_aux, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&idx->ev_bm_clean, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&idx->ev_inputs, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&idx->ev_kw_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&idx->ev_main_ready, hipEventDisableTiming) != hipSuccess ||
        !create_range_events(idx->ev_range) ||
        hipEventCreateWithFlags(&idx->ev_q, hipEventDisableTiming) != hipSuccess) {
        orr_index_destroy(idx);
        return fail(ORR_EDEVICE, "cannot create streams on device %d", cfg->device);
    }
    if (cfg->capacity_rows > 0) {
        int r = ensure_row_capacity(idx, cfg->capacity_rows);
        if (r != ORR_OK) { orr_index_destroy(idx); return r; }
    }
    *out = idx;
    return ORR_OK;
}

void orr_index_destroy(orr_index *idx)
{
    if (!idx) return;
    if (idx->is_view && !idx->internal_lane && idx->parent) const_cast<orr_index *>(idx->parent)->user_views.fetch_sub(1);
    if (!idx->is_view) {                               // the internal lanes go first (they borrow the corpus)
        std::vector<orr_index *> lanes;
        {
            AllLanes all(idx);
            lanes.swap(idx->lanes);
            idx->lane_busy.clear();
            idx->max_lanes = 1;
        }
        for (orr_index *l : lanes) if (l) orr_index_destroy(l);
    }
    (void)hipSetDevice(idx->device);
    if (idx->stream) (void)hipStreamSynchronize(idx->stream);
    if (idx->stream_kw) (void)hipStreamSynchronize(idx->stream_kw);
    if (idx->stream_aux) (void)hipStreamSynchronize(idx->stream_aux);
    if (idx->ev_bm_clean) (void)hipEventDestroy(idx->ev_bm_clean);
    if (idx->stream_aux) (void)hipStreamDestroy(idx->stream_aux);
    if (idx->ev_inputs) (void)hipEventDestroy(idx->ev_inputs);
    if (idx->ev_kw_done) (void)hipEventDestroy(idx->ev_kw_done);
    if (idx->ev_main_ready) (void)hipEventDestroy(idx->ev_main_ready);
    for (hipEvent_t e : idx->ev_range) if (e) (void)hipEventDestroy(e);
    if (idx->stream_kw) (void)hipStreamDestroy(idx->stream_kw);
    for (auto &pe : idx->pending) { (void)hipEventDestroy(pe.start); (void)hipEventDestroy(pe.stop); }
    for (auto e : idx->event_pool) (void)hipEventDestroy(e);
    if (!idx->is_view) {                               // a view borrows the corpus and the shadow
        if (idx->d_emb) (void)hipFree(idx->d_emb);
        if (idx->d_created) (void)hipFree(idx->d_created);
        if (idx->d_row_ids) (void)hipFree(idx->d_row_ids);
        if (idx->d_cstart) (void)hipFree(idx->d_cstart);
        if (idx->d_clen) (void)hipFree(idx->d_clen);
        if (idx->d_pool) (void)hipFree(idx->d_pool);
        if (idx->d_norm_b) (void)hipFree(idx->d_norm_b);
        if (idx->d_vpool) (void)hipFree(idx->d_vpool);
        if (idx->d_vstart) (void)hipFree(idx->d_vstart);
        if (idx->d_vlen) (void)hipFree(idx->d_vlen);
        if (idx->d_post_off) (void)hipFree(idx->d_post_off);
        if (idx->d_post_rows) (void)hipFree(idx->d_post_rows);
    } else {
        idx->emb_shadow.p = nullptr; idx->emb_shadow.cap = 0;
        for (DevBuf *b : {&idx->emb_i8, &idx->i8_scale, &idx->i8_rel_err, &idx->i8_rel_hat, &idx->i8_rowf, &idx->tok_bm, &idx->tok_bm_index}) { b->p = nullptr; b->cap = 0; }
    }
    idx->tok_bm.release(); idx->tok_bm_index.release();
    idx->d_dead.release();
    if (!idx->is_view) { idx->vlong_start.release(); idx->vlong_len.release(); idx->vlong_id.release(); }
    idx->ws_norm_a.release();
    DevBuf *bufs[] = {&idx->ws_q, &idx->ws_dot, &idx->ws_dotf, &idx->ws_rowc, &idx->ws_tau, &idx->ws_qsplit, &idx->ws_fcnt,
                      &idx->ws_fbuf, &idx->ws_fqf, &idx->ws_fany, &idx->ws_tsL, &idx->ws_tskey, &idx->ws_qtiled, &idx->ws_fdot, &idx->ws_pbuf, &idx->ws_psel, &idx->ws_q8, &idx->ws_q8s1, &idx->ws_q8err, &idx->ws_zero, &idx->ws_sel, &idx->ws_cand, &idx->ws_qc, &idx->ws_keys_a, &idx->ws_keys_b,
                      &idx->ws_vals_a, &idx->ws_vals_b, &idx->ws_sort_tmp, &idx->ws_raw, &idx->ws_src_start, &idx->ws_qsub,
                      &idx->ws_vmatch, &idx->ws_bitmaps, &idx->ws_hits, &idx->ws_counter, &idx->ws_meta, &idx->ws_tickets, &idx->ws_kwalias};
    for (auto b : bufs) b->release();
    idx->emb_shadow.release();
    idx->emb_i8.release(); idx->i8_scale.release(); idx->i8_rel_err.release(); idx->i8_rel_hat.release(); idx->i8_rowf.release();
    idx->pin_meta.release(); idx->pin_q.release(); idx->pin_qc.release(); idx->pin_cand.release(); idx->pin_norm.release(); idx->pin_cnt.release(); idx->pin_kwcnt.release();
    if (idx->ev_q) (void)hipEventDestroy(idx->ev_q);
    if (idx->stream) (void)hipStreamDestroy(idx->stream);
    delete idx;
}

int orr_index_set_row_base(orr_index *idx, int64_t row_base)
{
    if (!idx || row_base < 0) return fail(ORR_EINVAL, "orr_index_set_row_base: bad argument");
    AllLanes all(idx);
    std::lock_guard<std::mutex> lock(idx->mu);
    idx->row_base = row_base;
    for (orr_index *l : idx->lanes) if (l) l->row_base = row_base;
    return ORR_OK;
}

int64_t orr_index_rows(const orr_index *idx) { return idx ? idx->n_rows : 0; }
int32_t orr_index_dim(const orr_index *idx) { return idx ? idx->dim : 0; }

int orr_index_append(orr_index *idx, int64_t n, int32_t dim, const float *emb, const int64_t *created_ticks,
                     const uint8_t *content_lower, const uint64_t *content_off, const int64_t *row_ids)
{
    if (!idx) return fail(ORR_EINVAL, "orr_index_append: null index");
    std::lock_guard<std::mutex> lock(idx->mu);
    if (idx->sealed) return fail(ORR_ESTATE, "orr_index_append: index is sealed");
    if (n < 0) return fail(ORR_EINVAL, "orr_index_append: negative row count");
    if (n == 0) return ORR_OK;
    if (!created_ticks || !content_off) return fail(ORR_EINVAL, "orr_index_append: created_ticks and content_off are required");
    if (dim != 0 && dim != idx->dim)
        return fail(ORR_EDIM, "orr_index_append: dim %d differs from the index dimension %d", dim, idx->dim);
    if (dim != 0 && !emb) return fail(ORR_EINVAL, "orr_index_append: emb is NULL with dim %d", dim);
    if (idx->n_rows + n >= (int64_t)0xFFFFFFFFll) return fail(ORR_EINVAL, "orr_index_append: more than 2^32-1 rows in one shard");
    ORR_TRY(bind_device(idx));
    ORR_TRY(ensure_row_capacity(idx, idx->n_rows + n));

    // content offsets: bring to the host, rebase onto the pool
    std::vector<uint64_t> off((size_t)n + 1);
    HIP_TRY(hipMemcpy(off.data(), content_off, sizeof(uint64_t) * ((size_t)n + 1), hipMemcpyDefault));
    for (int64_t i = 0; i < n; ++i)
        if (off[i + 1] < off[i]) return fail(ORR_EINVAL, "orr_index_append: content_off is not monotone at row %lld", (long long)i);
    const uint64_t bytes = off[n] - off[0];
    if (bytes > 0 && !content_lower) return fail(ORR_EINVAL, "orr_index_append: content_lower is NULL");
    // re-lay the rows out for the scan kernel: 16-byte aligned starts, space padding behind each row
    std::vector<uint64_t> src_start((size_t)n), dst_start((size_t)n);
    std::vector<uint32_t> lens((size_t)n);
    uint64_t cursor = idx->pool_len;
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t len = off[i + 1] - off[i];
        if (len >= (1ull << 31)) return fail(ORR_EINVAL, "orr_index_append: content of row %lld exceeds 2 GiB", (long long)i);
        src_start[i] = off[i] - off[0];
        dst_start[i] = cursor;
        lens[i] = (uint32_t)len;
        cursor += orr::padded_row_bytes(len);
    }
    ORR_TRY(ensure_pool_capacity(idx, cursor));
    HIP_TRY(hipMemsetAsync(idx->d_pool + idx->pool_len, 0x20, cursor - idx->pool_len, idx->stream));
    HIP_TRY(hipMemcpyAsync(idx->d_cstart + idx->n_rows, dst_start.data(), sizeof(uint64_t) * (size_t)n, hipMemcpyHostToDevice, idx->stream));
    HIP_TRY(hipMemcpyAsync(idx->d_clen + idx->n_rows, lens.data(), sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice, idx->stream));
    if (bytes > 0) {
        ORR_TRY(idx->ws_raw.reserve(bytes));
        ORR_TRY(idx->ws_src_start.reserve(sizeof(uint64_t) * (size_t)n));
        HIP_TRY(hipMemcpyAsync(idx->ws_raw.p, content_lower + off[0], bytes, hipMemcpyDefault, idx->stream));
        HIP_TRY(hipMemcpyAsync(idx->ws_src_start.p, src_start.data(), sizeof(uint64_t) * (size_t)n, hipMemcpyHostToDevice, idx->stream));
        HIP_TRY(orr::launch_gather_content(idx->ws_raw.as<uint8_t>(), idx->ws_src_start.as<uint64_t>(), idx->d_clen + idx->n_rows,
                                           idx->d_pool, idx->d_cstart + idx->n_rows, nullptr, n, idx->stream));
    }
    HIP_TRY(hipStreamSynchronize(idx->stream));

    // timestamps (host mirror + device)
    const size_t old = idx->h_created.size();
    idx->h_created.resize(old + (size_t)n);
    HIP_TRY(hipMemcpy(idx->h_created.data() + old, created_ticks, sizeof(int64_t) * (size_t)n, hipMemcpyDefault));
    HIP_TRY(hipMemcpyAsync(idx->d_created + idx->n_rows, idx->h_created.data() + old, sizeof(int64_t) * (size_t)n,
                           hipMemcpyHostToDevice, idx->stream));

    // embeddings: copy, or zero rows (a zero row has normB = 0 -> cosine 0, the same
    // value the null/empty guard of RecallSearchService.cs:71 yields)
    if (idx->dim > 0) {
        float *dst = idx->d_emb + (size_t)idx->n_rows * idx->dim;
        const size_t eb = sizeof(float) * (size_t)n * idx->dim;
        if (dim == idx->dim)
            HIP_TRY(hipMemcpyAsync(dst, emb, eb, hipMemcpyDefault, idx->stream));
        else
            HIP_TRY(hipMemsetAsync(dst, 0, eb, idx->stream));
    }
    if (row_ids)
        HIP_TRY(hipMemcpyAsync(idx->d_row_ids + idx->n_rows, row_ids, sizeof(int64_t) * (size_t)n, hipMemcpyDefault, idx->stream));
    else
        HIP_TRY(orr::launch_iota_i64(idx->d_row_ids + idx->n_rows, n, idx->row_base + idx->n_rows, idx->stream));
    HIP_TRY(hipStreamSynchronize(idx->stream));
    idx->h_clen.insert(idx->h_clen.end(), lens.begin(), lens.end());
    idx->pool_len = cursor;
    idx->n_rows += n;
    return ORR_OK;
}

int orr_index_seal(orr_index *idx)
{
    if (!idx) return fail(ORR_EINVAL, "orr_index_seal: null index");
    std::lock_guard<std::mutex> lock(idx->mu);
    if (idx->sealed) return ORR_OK;
    ORR_TRY(bind_device(idx));
    const int64_t n = idx->n_rows;

    // candidate order: stable sort by CreatedAt descending (InMemoryIngestionStore.cs:61)
    std::vector<int64_t> perm((size_t)n);
    std::iota(perm.begin(), perm.end(), (int64_t)0);
    const int64_t *cr = idx->h_created.data();
    bool identity = true;
    for (int64_t i = 1; i < n && identity; ++i) identity = cr[i - 1] >= cr[i];
    if (!identity) {
        std::stable_sort(perm.begin(), perm.end(), [cr](int64_t a, int64_t b) { return cr[a] > cr[b]; });
        int64_t *d_perm = nullptr;
        ORR_TRY(dev_alloc(&d_perm, (size_t)n));
        HIP_TRY(hipMemcpyAsync(d_perm, perm.data(), sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, idx->stream));
        if (idx->dim > 0) {
            float *ne = nullptr;
            ORR_TRY(dev_alloc(&ne, (size_t)idx->cap_rows * idx->dim));
            HIP_TRY(orr::launch_gather_rows_f32(idx->d_emb, ne, d_perm, n, idx->dim, idx->stream));
            HIP_TRY(hipStreamSynchronize(idx->stream));
            (void)hipFree(idx->d_emb);
            idx->d_emb = ne;
        }
        int64_t *nc = nullptr, *nr = nullptr;
        ORR_TRY(dev_alloc(&nc, (size_t)idx->cap_rows));
        ORR_TRY(dev_alloc(&nr, (size_t)idx->cap_rows));
        HIP_TRY(orr::launch_gather_i64(idx->d_created, nc, d_perm, n, idx->stream));
        HIP_TRY(orr::launch_gather_i64(idx->d_row_ids, nr, d_perm, n, idx->stream));
        std::vector<uint64_t> nstart((size_t)n);
        std::vector<uint32_t> nlen((size_t)n);
        uint64_t cursor = 0;
        for (int64_t p = 0; p < n; ++p) {
            nlen[p] = idx->h_clen[perm[p]];
            nstart[p] = cursor;
            cursor += orr::padded_row_bytes(nlen[p]);
        }
        uint64_t *d_nstart = nullptr;
        uint32_t *d_nlen = nullptr;
        uint8_t *npool = nullptr;
        ORR_TRY(dev_alloc(&d_nstart, (size_t)idx->cap_rows));
        ORR_TRY(dev_alloc(&d_nlen, (size_t)idx->cap_rows));
        ORR_TRY(dev_alloc(&npool, (size_t)idx->pool_cap));
        HIP_TRY(hipMemsetAsync(npool, 0x20, cursor, idx->stream));
        HIP_TRY(hipMemcpyAsync(d_nstart, nstart.data(), sizeof(uint64_t) * (size_t)n, hipMemcpyHostToDevice, idx->stream));
        HIP_TRY(orr::launch_gather_content(idx->d_pool, idx->d_cstart, idx->d_clen, npool, d_nstart, d_perm, n, idx->stream));
        HIP_TRY(orr::launch_gather_u32(idx->d_clen, d_nlen, d_perm, n, idx->stream));
        HIP_TRY(hipStreamSynchronize(idx->stream));
        (void)hipFree(idx->d_created); idx->d_created = nc;
        (void)hipFree(idx->d_row_ids); idx->d_row_ids = nr;
        (void)hipFree(idx->d_cstart); idx->d_cstart = d_nstart;
        (void)hipFree(idx->d_clen); idx->d_clen = d_nlen;
        (void)hipFree(idx->d_pool); idx->d_pool = npool;
        idx->pool_len = cursor;
        (void)hipFree(d_perm);
        std::vector<int64_t> sorted_created((size_t)n);
        for (int64_t p = 0; p < n; ++p) sorted_created[p] = cr[perm[p]];
        idx->h_created.swap(sorted_created);
        idx->h_clen.swap(nlen);
    }
    idx->h_cprefix.assign((size_t)n + 1, 0);
    for (int64_t p = 0; p < n; ++p) idx->h_cprefix[p + 1] = idx->h_cprefix[p] + idx->h_clen[p];

    // token index: content goes to the host once, comes back as vocabulary + postings,
    // and the raw text leaves HBM
    if (n > 0) {
        std::vector<uint8_t> h_pool((size_t)idx->pool_len + 16);
        std::vector<uint64_t> h_cstart((size_t)n);
        if (idx->pool_len) HIP_TRY(hipMemcpy(h_pool.data(), idx->d_pool, (size_t)idx->pool_len, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(h_cstart.data(), idx->d_cstart, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToHost));
        orr::TokenIndexHost ti;
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        orr::build_token_index(h_pool.data(), h_cstart.data(), idx->h_clen.data(), n, (int)std::min(hw, 32u), ti);
        idx->n_tokens = (int64_t)ti.vstart.size();
        idx->n_postings = ti.post_rows.size();
        ORR_TRY(dev_alloc(&idx->d_vpool, ti.vpool.size() + orr::kScanPoolSlack));
        ORR_TRY(dev_alloc(&idx->d_vstart, ti.vstart.size()));
        ORR_TRY(dev_alloc(&idx->d_vlen, ti.vlen.size()));
        ORR_TRY(dev_alloc(&idx->d_post_off, ti.post_off.size()));
        ORR_TRY(dev_alloc(&idx->d_post_rows, ti.post_rows.size()));
        HIP_TRY(hipMemsetAsync(idx->d_vpool, 0x20, ti.vpool.size() + orr::kScanPoolSlack, idx->stream));
        if (!ti.vpool.empty()) HIP_TRY(hipMemcpyAsync(idx->d_vpool, ti.vpool.data(), ti.vpool.size(), hipMemcpyHostToDevice, idx->stream));
        if (idx->n_tokens) {
            HIP_TRY(hipMemcpyAsync(idx->d_vstart, ti.vstart.data(), sizeof(uint64_t) * ti.vstart.size(), hipMemcpyHostToDevice, idx->stream));
            HIP_TRY(hipMemcpyAsync(idx->d_vlen, ti.vlen.data(), sizeof(uint32_t) * ti.vlen.size(), hipMemcpyHostToDevice, idx->stream));
        }
        HIP_TRY(hipMemcpyAsync(idx->d_post_off, ti.post_off.data(), sizeof(uint64_t) * ti.post_off.size(), hipMemcpyHostToDevice, idx->stream));
        if (idx->n_postings)
            HIP_TRY(hipMemcpyAsync(idx->d_post_rows, ti.post_rows.data(), sizeof(uint32_t) * ti.post_rows.size(), hipMemcpyHostToDevice, idx->stream));
        HIP_TRY(hipStreamSynchronize(idx->stream));
        (void)hipFree(idx->d_pool); idx->d_pool = nullptr; idx->pool_cap = 0;
        (void)hipFree(idx->d_cstart); idx->d_cstart = nullptr;
        (void)hipFree(idx->d_clen); idx->d_clen = nullptr;
    }

    // K0: exact row norms, sum_i (double)fl32(e_i*e_i) (RecallSearchService.cs:81)
    ORR_TRY(dev_alloc(&idx->d_norm_b, (size_t)std::max<int64_t>(n, 1)));
    if (idx->dim > 0 && n > 0) {
        Timed t(idx, "row_norms_exact", 4.0 * (double)n * idx->dim + 8.0 * (double)n);
        HIP_TRY(orr::launch_dot_exact(idx->d_emb, n, idx->dim, nullptr, 1, true, idx->d_norm_b, n, idx->stream));
    } else if (n > 0) {
        HIP_TRY(hipMemsetAsync(idx->d_norm_b, 0, sizeof(double) * (size_t)n, idx->stream));
    }
    HIP_TRY(hipStreamSynchronize(idx->stream));
    collect_events(idx);
    idx->sealed = true;
    return ORR_OK;
}

namespace {

struct ShardHeader {
    char magic[8];               // "ORRSHD1\0"
    uint32_t version, dim;
    int64_t n_rows, n_tokens;
    uint64_t n_postings, vpool_bytes;
    uint64_t reserved[4];
};
constexpr size_t kIoChunk = 64u << 20;

int write_device_array(FILE *f, const void *dptr, size_t bytes, std::vector<uint8_t> &buf)
{
    const uint8_t *p = static_cast<const uint8_t *>(dptr);
    for (size_t off = 0; off < bytes; off += kIoChunk) {
        const size_t m = std::min(kIoChunk, bytes - off);
        HIP_TRY(hipMemcpy(buf.data(), p + off, m, hipMemcpyDeviceToHost));
        if (fwrite(buf.data(), 1, m, f) != m) return fail(ORR_EINVAL, "short write to the shard file");
    }
    return ORR_OK;
}

int read_device_array(FILE *f, void *dptr, size_t bytes, std::vector<uint8_t> &buf)
{
    uint8_t *p = static_cast<uint8_t *>(dptr);
    for (size_t off = 0; off < bytes; off += kIoChunk) {
        const size_t m = std::min(kIoChunk, bytes - off);
        if (fread(buf.data(), 1, m, f) != m) return fail(ORR_EINVAL, "shard file is truncated");
        HIP_TRY(hipMemcpy(p + off, buf.data(), m, hipMemcpyHostToDevice));
    }
    return ORR_OK;
}

}  // namespace

int orr_index_save(orr_index *idx, const char *path)
{
    if (!idx || !path) return fail(ORR_EINVAL, "orr_index_save: null argument");
    AllLanes all(idx);
    std::lock_guard<std::mutex> lock(idx->mu);
    if (!idx->sealed) return fail(ORR_ESTATE, "orr_index_save: index is not sealed");
    if (idx->is_view) return fail(ORR_ESTATE, "orr_index_save: save the owning index, not a view");
    ORR_TRY(bind_device(idx));
    FILE *f = fopen(path, "wb");
    if (!f) return fail(ORR_EINVAL, "orr_index_save: cannot open %s", path);
    ShardHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "ORRSHD1", 8);
    h.version = 1; h.dim = (uint32_t)idx->dim; h.n_rows = idx->n_rows; h.n_tokens = idx->n_tokens;
    h.n_postings = idx->n_postings;
    h.reserved[0] = (uint64_t)idx->dead.size();          // deleted positions follow the last array
    uint64_t vpool_bytes = 0;
    std::vector<uint64_t> vstart((size_t)idx->n_tokens);
    std::vector<uint32_t> vlen((size_t)idx->n_tokens);
    if (idx->n_tokens > 0) {
        hipError_t e1 = hipMemcpy(vstart.data(), idx->d_vstart, sizeof(uint64_t) * vstart.size(), hipMemcpyDeviceToHost);
        hipError_t e2 = hipMemcpy(vlen.data(), idx->d_vlen, sizeof(uint32_t) * vlen.size(), hipMemcpyDeviceToHost);
        if (e1 != hipSuccess || e2 != hipSuccess) { fclose(f); return fail(ORR_EDEVICE, "orr_index_save: device copy failed"); }
        vpool_bytes = vstart.back() + orr::padded_row_bytes(vlen.back());
    }
    h.vpool_bytes = vpool_bytes;
    std::vector<uint8_t> buf(kIoChunk);
    const size_t n = (size_t)idx->n_rows, V = (size_t)idx->n_tokens;
    int r = fwrite(&h, sizeof(h), 1, f) == 1 ? ORR_OK : fail(ORR_EINVAL, "short write to the shard file");
    if (r == ORR_OK && idx->dim > 0) r = write_device_array(f, idx->d_emb, sizeof(float) * n * idx->dim, buf);
    if (r == ORR_OK) r = write_device_array(f, idx->d_norm_b, sizeof(double) * n, buf);
    if (r == ORR_OK) r = write_device_array(f, idx->d_created, sizeof(int64_t) * n, buf);
    if (r == ORR_OK) r = write_device_array(f, idx->d_row_ids, sizeof(int64_t) * n, buf);
    if (r == ORR_OK && n) r = fwrite(idx->h_clen.data(), sizeof(uint32_t), n, f) == n ? ORR_OK : fail(ORR_EINVAL, "short write");
    if (r == ORR_OK && V) {
        r = fwrite(vstart.data(), sizeof(uint64_t), V, f) == V && fwrite(vlen.data(), sizeof(uint32_t), V, f) == V
                ? ORR_OK : fail(ORR_EINVAL, "short write");
        if (r == ORR_OK) r = write_device_array(f, idx->d_vpool, vpool_bytes, buf);
    }
    if (r == ORR_OK && n) r = write_device_array(f, idx->d_post_off, sizeof(uint64_t) * (V + 1), buf);
    if (r == ORR_OK && idx->n_postings) r = write_device_array(f, idx->d_post_rows, sizeof(uint32_t) * (size_t)idx->n_postings, buf);
    if (r == ORR_OK && !idx->dead.empty())
        r = fwrite(idx->dead.data(), sizeof(int64_t), idx->dead.size(), f) == idx->dead.size() ? ORR_OK : fail(ORR_EINVAL, "short write");
    if (fclose(f) != 0 && r == ORR_OK) r = fail(ORR_EINVAL, "orr_index_save: close failed");
    return r;
}

int orr_index_load(const orr_config *cfg, const char *path, orr_index **out)
{
    if (!cfg || !path || !out) return fail(ORR_EINVAL, "orr_index_load: null argument");
    *out = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) return fail(ORR_EINVAL, "orr_index_load: cannot open %s", path);
    ShardHeader h;
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "ORRSHD1", 8) != 0 || h.version != 1 || h.n_rows < 0 || h.n_tokens < 0) {
        fclose(f);
        return fail(ORR_EINVAL, "orr_index_load: %s is not a version-1 shard file", path);
    }
    if (cfg->dim != 0 && cfg->dim != (int32_t)h.dim) {
        fclose(f);
        return fail(ORR_EDIM, "orr_index_load: file dimension %u differs from the requested %d", h.dim, cfg->dim);
    }
    {   // the header's counts against the size of the file, before anything is allocated from them
        long at = ftell(f), end = -1;
        if (at >= 0 && fseek(f, 0, SEEK_END) == 0) end = ftell(f);
        if (at < 0 || end < 0 || fseek(f, at, SEEK_SET) != 0) { fclose(f); return fail(ORR_EINVAL, "orr_index_load: cannot size %s", path); }
        const long double nn = (long double)h.n_rows, vv = (long double)h.n_tokens;
        long double want = (long double)sizeof(h) + nn * (4.0L * h.dim + 8 + 8 + 8 + 4) + (long double)h.n_postings * 4 + (long double)h.reserved[0] * 8;
        if (h.n_tokens) want += vv * 12 + (long double)h.vpool_bytes;
        if (h.n_rows) want += (vv + 1) * 8;
        if (want != (long double)end) {
            fclose(f);
            return fail(ORR_EINVAL, "orr_index_load: %s: the header's counts do not add up to the file's size", path);
        }
    }
    orr_config c = *cfg;
    c.dim = (int32_t)h.dim;
    c.capacity_rows = h.n_rows;
    orr_index *idx = nullptr;
    int r = orr_index_create(&c, &idx);
    if (r != ORR_OK) { fclose(f); return r; }
    // idx is private to this call until it is returned: no locking needed
    const size_t n = (size_t)h.n_rows, V = (size_t)h.n_tokens;
    std::vector<uint8_t> buf(kIoChunk);
    idx->n_rows = h.n_rows; idx->n_tokens = h.n_tokens; idx->n_postings = h.n_postings;
    auto body = [&]() -> int {
        if (idx->dim > 0) ORR_TRY(read_device_array(f, idx->d_emb, sizeof(float) * n * idx->dim, buf));
        ORR_TRY(dev_alloc(&idx->d_norm_b, std::max<size_t>(n, 1)));
        ORR_TRY(read_device_array(f, idx->d_norm_b, sizeof(double) * n, buf));
        idx->h_created.resize(n);
        const long pos = ftell(f);
        if (n && fread(idx->h_created.data(), sizeof(int64_t), n, f) != n) return fail(ORR_EINVAL, "shard file is truncated");
        fseek(f, pos, SEEK_SET);
        ORR_TRY(read_device_array(f, idx->d_created, sizeof(int64_t) * n, buf));
        ORR_TRY(read_device_array(f, idx->d_row_ids, sizeof(int64_t) * n, buf));
        idx->h_clen.resize(n);
        if (n && fread(idx->h_clen.data(), sizeof(uint32_t), n, f) != n) return fail(ORR_EINVAL, "shard file is truncated");
        idx->h_cprefix.assign(n + 1, 0);
        for (size_t p = 0; p < n; ++p) idx->h_cprefix[p + 1] = idx->h_cprefix[p] + idx->h_clen[p];
        if (V) {
            ORR_TRY(dev_alloc(&idx->d_vstart, V));
            ORR_TRY(dev_alloc(&idx->d_vlen, V));
            ORR_TRY(dev_alloc(&idx->d_vpool, (size_t)h.vpool_bytes + orr::kScanPoolSlack));
            std::vector<uint64_t> vstart_h(V);
            ORR_TRY(read_device_array_checked(f, idx->d_vstart, V, buf, [&](const uint64_t *x, size_t off, size_t m) {
                for (size_t i = 0; i < m; ++i) { if (x[i] > h.vpool_bytes) return false; vstart_h[off + i] = x[i]; }
                return true; }, "vocabulary offsets"));
            ORR_TRY(read_device_array_checked(f, idx->d_vlen, V, buf, [&](const uint32_t *x, size_t off, size_t m) {
                for (size_t i = 0; i < m; ++i) if (vstart_h[off + i] + orr::padded_row_bytes(x[i]) > h.vpool_bytes) return false;
                return true; }, "vocabulary lengths"));
            HIP_TRY(hipMemset(idx->d_vpool, 0x20, (size_t)h.vpool_bytes + orr::kScanPoolSlack));
            ORR_TRY(read_device_array(f, idx->d_vpool, (size_t)h.vpool_bytes, buf));
        }
        if (n) {
            ORR_TRY(dev_alloc(&idx->d_post_off, V + 1));
            uint64_t prev = 0;
            ORR_TRY(read_device_array_checked(f, idx->d_post_off, V + 1, buf, [&](const uint64_t *x, size_t off, size_t m) {
                for (size_t i = 0; i < m; ++i) {
                    if (x[i] < prev || x[i] > h.n_postings || (off + i == 0 && x[i] != 0) || (off + i == V && x[i] != h.n_postings)) return false;
                    prev = x[i];
                }
                return true; }, "posting offsets"));
        } else if (h.n_postings) {
            return fail(ORR_EINVAL, "shard file: postings without rows");
        }
        ORR_TRY(dev_alloc(&idx->d_post_rows, std::max<size_t>((size_t)h.n_postings, 1)));
        if (h.n_postings)
            ORR_TRY(read_device_array_checked(f, idx->d_post_rows, (size_t)h.n_postings, buf, [&](const uint32_t *x, size_t, size_t m) {
                for (size_t i = 0; i < m; ++i) if ((uint64_t)x[i] >= (uint64_t)n) return false;
                return true; }, "posting rows"));
        if (h.reserved[0]) {                          // deleted rows: norms and timestamps in the file are already overwritten
            if (h.reserved[0] > (uint64_t)n) return fail(ORR_EINVAL, "shard file lists more deleted rows than rows");
            idx->dead.resize((size_t)h.reserved[0]);
            if (fread(idx->dead.data(), sizeof(int64_t), idx->dead.size(), f) != idx->dead.size()) return fail(ORR_EINVAL, "shard file is truncated");
            for (size_t i = 0; i < idx->dead.size(); ++i)
                if (idx->dead[i] < 0 || idx->dead[i] >= (int64_t)n || (i && idx->dead[i] <= idx->dead[i - 1]))
                    return fail(ORR_EINVAL, "shard file has a malformed deleted-row list");
            ORR_TRY(idx->d_dead.reserve(sizeof(int64_t) * idx->dead.size()));
            HIP_TRY(hipMemcpy(idx->d_dead.p, idx->dead.data(), sizeof(int64_t) * idx->dead.size(), hipMemcpyHostToDevice));
            idx->dead_count_pub = (int64_t)idx->dead.size();
        }
        return ORR_OK;
    };
    r = body();
    fclose(f);
    if (r != ORR_OK) {
        const std::string keep = g_last_error;
        orr_index_destroy(idx);
        g_last_error = keep;
        return r;
    }
    // the raw content buffers of an unsealed index are not part of a shard file
    if (idx->d_cstart) { (void)hipFree(idx->d_cstart); idx->d_cstart = nullptr; }
    if (idx->d_clen) { (void)hipFree(idx->d_clen); idx->d_clen = nullptr; }
    idx->sealed = true;
    *out = idx;
    return ORR_OK;
}

// bf16 shadow of the sealed embeddings for the screening GEMM.  Half the master's bytes; when the
// allocation does not fit, the two-stage pass converts in the kernel instead (orr_gemm.hip, PROD = 1).
static int ensure_shadow(orr_index *idx)
{
    if (idx->is_view) return ORR_OK;                   // taken from the parent at creation, or absent
    if (idx->shadow_ready || idx->shadow_failed || !idx->sealed || idx->n_rows <= 0 || idx->dim <= 0 || idx->dim % 64 != 0) return ORR_OK;
    const size_t bytes = orr::bf16_tiled_bytes(idx->n_rows, idx->dim);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + bytes / 8 + ((size_t)4 << 30)) {   // keep 4 GiB for workspaces
        idx->shadow_failed = true;
        return ORR_OK;
    }
    if (idx->emb_shadow.reserve(bytes) != ORR_OK) {
        (void)hipGetLastError();
        idx->shadow_failed = true;
        return ORR_OK;
    }
    HIP_TRY(orr::launch_bf16_tiled(idx->d_emb, idx->n_rows, idx->dim, idx->emb_shadow.p, idx->stream));
    HIP_TRY(hipStreamSynchronize(idx->stream));
    idx->shadow_ready = true;
    return ORR_OK;
}

// Int8 shadow (a quarter of the master's bytes): operand of the streaming screen of 1..4 queries (K2i) and of
// the screening GEMM of larger batches (K2j).  Where it exists the bf16 shadow is only built for what it
// does not cover (5..8 queries, dimensions that are not a multiple of 128).
static int ensure_i8_shadow(orr_index *idx)
{
    if (idx->is_view || idx->i8_ready || idx->i8_failed || !idx->sealed || idx->n_rows <= 0 || idx->dim <= 0 || idx->dim % 128 != 0) return ORR_OK;
    const size_t bytes = orr::i8_tiled_bytes(idx->n_rows, idx->dim) + 28 * (size_t)idx->n_rows;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + bytes / 8 + ((size_t)8 << 30)) {   // keep 8 GiB for workspaces
        idx->i8_failed = true;
        return ORR_OK;
    }
    if (idx->emb_i8.reserve(orr::i8_tiled_bytes(idx->n_rows, idx->dim)) != ORR_OK || idx->i8_scale.reserve(sizeof(float) * (size_t)idx->n_rows) != ORR_OK ||
        idx->i8_rel_err.reserve(sizeof(float) * (size_t)idx->n_rows) != ORR_OK ||
        idx->i8_rel_hat.reserve(sizeof(float) * (size_t)idx->n_rows) != ORR_OK ||
        idx->i8_rowf.reserve(sizeof(float4) * (size_t)idx->n_rows) != ORR_OK) {
        (void)hipGetLastError();
        idx->emb_i8.release(); idx->i8_scale.release(); idx->i8_rel_err.release(); idx->i8_rel_hat.release(); idx->i8_rowf.release();
        idx->i8_failed = true;
        return ORR_OK;
    }
    HIP_TRY(orr::launch_i8_shadow(idx->d_emb, idx->d_norm_b, idx->n_rows, idx->dim, idx->emb_i8.p, idx->i8_scale.as<float>(),
                                  idx->i8_rel_err.as<float>(), idx->i8_rel_hat.as<float>(), idx->stream));
    HIP_TRY(orr::launch_i8_rowf(idx->i8_scale.as<float>(), idx->i8_rel_err.as<float>(), idx->i8_rel_hat.as<float>(), idx->n_rows,
                                idx->i8_rowf.as<float4>(), idx->stream));
    HIP_TRY(hipStreamSynchronize(idx->stream));
    idx->i8_ready = true;
    return ORR_OK;
}

// The long tokens of the vocabulary as their own row list for the wave-per-token scan.
static int ensure_vlong(orr_index *idx)
{
    if (idx->n_vlong >= 0 || idx->is_view) return ORR_OK;          // a view copies its parent's at creation
    const size_t V = (size_t)idx->n_tokens;
    std::vector<uint64_t> start, vstart(V);
    std::vector<uint32_t> len, id, vlen(V);
    if (V) {
        HIP_TRY(hipMemcpy(vstart.data(), idx->d_vstart, sizeof(uint64_t) * V, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(vlen.data(), idx->d_vlen, sizeof(uint32_t) * V, hipMemcpyDeviceToHost));
    }
    for (int pass = 0; pass < 2; ++pass) {                         // 17..32 bytes first, the longer ones behind them
        for (size_t v = 0; v < V; ++v)
            if (vlen[v] > 16 && (vlen[v] <= 32) == (pass == 0)) { start.push_back(vstart[v]); len.push_back(vlen[v]); id.push_back((uint32_t)v); }
        if (pass == 0) idx->n_vmid = (int64_t)id.size();
    }
    if (!id.empty()) {
        ORR_TRY(idx->vlong_start.reserve(sizeof(uint64_t) * id.size()));
        ORR_TRY(idx->vlong_len.reserve(sizeof(uint32_t) * id.size()));
        ORR_TRY(idx->vlong_id.reserve(sizeof(uint32_t) * id.size()));
        HIP_TRY(hipMemcpy(idx->vlong_start.p, start.data(), sizeof(uint64_t) * id.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(idx->vlong_len.p, len.data(), sizeof(uint32_t) * id.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(idx->vlong_id.p, id.data(), sizeof(uint32_t) * id.size(), hipMemcpyHostToDevice));
    }
    idx->n_vlong = (int64_t)id.size();
    return ORR_OK;
}

// Row bitmaps of the frequent vocabulary tokens (see orr_index::tok_bm), through the same posting expansion a batch runs.
static int ensure_token_bitmaps(orr_index *idx)
{
    if (idx->is_view || idx->n_tok_bm >= 0 || !idx->sealed) return ORR_OK;    // a view copies its parent's at creation
    idx->n_tok_bm = 0;
    if (const char *e = getenv("ORR_TOKEN_BITMAPS")) { if (atoi(e) == 0) return ORR_OK; }      // diagnostic: A/B against per-batch expansion
    const int64_t n = idx->n_rows, V = idx->n_tokens;
    if (n < 48 * (int64_t)orr::kSelSegRows || V <= 0 || idx->n_postings == 0) return ORR_OK;   // small shards expand in microseconds
    const int64_t words = ((n + 31) / 32 + 3) / 4 * 4;
    std::vector<uint64_t> off((size_t)V + 1);
    HIP_TRY(hipMemcpy(off.data(), idx->d_post_off, sizeof(uint64_t) * ((size_t)V + 1), hipMemcpyDeviceToHost));
    std::vector<uint32_t> toks;
    for (int64_t v = 0; v < V; ++v)
        if ((off[(size_t)v + 1] - off[(size_t)v]) * 64 >= (uint64_t)n) toks.push_back((uint32_t)v);
    if (toks.empty()) return ORR_OK;
    const size_t bytes = toks.size() * (size_t)words * sizeof(uint32_t);
    size_t free_b = 0, total_b = 0;
    if (bytes > ((size_t)32 << 30) || hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + ((size_t)12 << 30)) return ORR_OK;   // no room: expand per batch
    if (idx->tok_bm.reserve(bytes) != ORR_OK || idx->tok_bm_index.reserve(sizeof(int32_t) * (size_t)V) != ORR_OK) {
        (void)hipGetLastError();
        idx->tok_bm.release(); idx->tok_bm_index.release();
        return ORR_OK;
    }
    std::vector<int32_t> index((size_t)V, -1);
    std::vector<orr::KwHit> hits(toks.size());
    uint64_t chunks = 0;
    for (size_t j = 0; j < toks.size(); ++j) {
        const uint32_t v = toks[j];
        index[v] = (int32_t)j;
        orr::KwHit h;
        h.post_begin = off[v]; h.post_len = (uint32_t)(off[(size_t)v + 1] - off[v]); h.chunk_base = (uint32_t)chunks; h.term = (uint32_t)j; h.token = v;
        hits[j] = h;
        chunks += (h.post_len + orr::kPostChunk - 1) / orr::kPostChunk;
    }
    if (chunks >= ((uint64_t)1 << 32)) { idx->tok_bm.release(); idx->tok_bm_index.release(); return ORR_OK; }
    const unsigned long long counter = ((unsigned long long)toks.size() << 32) | (unsigned long long)chunks;
    DevBuf d_hits, d_counter;
    ORR_TRY(d_hits.reserve(sizeof(orr::KwHit) * hits.size()));
    ORR_TRY(d_counter.reserve(sizeof(counter)));
    hipStream_t s = idx->stream;
    hipError_t e = hipMemcpyAsync(d_hits.p, hits.data(), sizeof(orr::KwHit) * hits.size(), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_counter.p, &counter, sizeof(counter), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(idx->tok_bm_index.p, index.data(), sizeof(int32_t) * (size_t)V, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(idx->tok_bm.p, 0, bytes, s);
    if (e == hipSuccess) e = orr::launch_expand_hits(d_hits.as<orr::KwHit>(), d_counter.as<unsigned long long>(), (uint32_t)hits.size(), idx->d_post_rows,
                                                     idx->tok_bm.as<uint32_t>(), words, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    d_hits.release(); d_counter.release();
    if (e != hipSuccess) {
        idx->tok_bm.release(); idx->tok_bm_index.release();
        return fail(ORR_EDEVICE, "token bitmaps: %s", hipGetErrorString(e));
    }
    idx->n_tok_bm = (int64_t)toks.size();
    idx->tok_bm_words = words;
    return ORR_OK;
}

int64_t orr_index_live_rows(const orr_index *idx)
{
    return idx ? idx->n_rows - (int64_t)(idx->parent ? idx->parent->dead.size() : idx->dead.size()) : 0;
}

int orr_index_delete_rows(orr_index *idx, int64_t n, const int64_t *row_ids, int64_t *out_deleted)
{
    if (out_deleted) *out_deleted = 0;
    if (!idx || n < 0 || (n > 0 && !row_ids)) return fail(ORR_EINVAL, "orr_index_delete_rows: bad argument");
    AllLanes all(idx);                                 // no search in flight on any lane while the rows change
    std::lock_guard<std::mutex> lock(idx->mu);
    if (!idx->sealed) return fail(ORR_ESTATE, "orr_index_delete_rows: the index is not sealed");
    if (idx->is_view) return fail(ORR_EINVAL, "orr_index_delete_rows: delete on the owning index, not on a view");
    if (n == 0 || idx->n_rows == 0) return ORR_OK;
    HIP_TRY(hipSetDevice(idx->device));
    const size_t rows = (size_t)idx->n_rows;
    if (idx->id_index.empty()) {                       // ids -> positions, once
        std::vector<int64_t> ids(rows);
        HIP_TRY(hipMemcpy(ids.data(), idx->d_row_ids, sizeof(int64_t) * rows, hipMemcpyDeviceToHost));
        idx->id_index.resize(rows);
        for (size_t p = 0; p < rows; ++p) idx->id_index[p] = {ids[p], (int64_t)p};
        std::sort(idx->id_index.begin(), idx->id_index.end());
    }
    std::vector<int64_t> want((size_t)n);
    HIP_TRY(hipMemcpy(want.data(), row_ids, sizeof(int64_t) * (size_t)n, hipMemcpyDefault));
    std::vector<int64_t> fresh;
    for (int64_t id : want) {
        auto it = std::lower_bound(idx->id_index.begin(), idx->id_index.end(), std::make_pair(id, (int64_t)-1));
        for (; it != idx->id_index.end() && it->first == id; ++it)
            if (!std::binary_search(idx->dead.begin(), idx->dead.end(), it->second)) fresh.push_back(it->second);
    }
    std::sort(fresh.begin(), fresh.end());
    fresh.erase(std::unique(fresh.begin(), fresh.end()), fresh.end());
    if (fresh.empty()) return ORR_OK;
    if ((idx->dead.size() + fresh.size()) * 4 > rows)
        return fail(ORR_ESTATE, "orr_index_delete_rows: more than a quarter of the shard's %lld rows would be deleted: rebuild the shard",
                    (long long)idx->n_rows);
    std::vector<int64_t> merged(idx->dead.size() + fresh.size());
    std::merge(idx->dead.begin(), idx->dead.end(), fresh.begin(), fresh.end(), merged.begin());
    DevBuf tmp;
    ORR_TRY(tmp.reserve(sizeof(int64_t) * fresh.size()));
    int r = ORR_OK;
    if (hipMemcpy(tmp.p, fresh.data(), sizeof(int64_t) * fresh.size(), hipMemcpyHostToDevice) != hipSuccess ||
        orr::launch_tombstone_rows(tmp.as<int64_t>(), (int32_t)fresh.size(), idx->d_norm_b, idx->d_created, idx->stream) != hipSuccess ||
        hipStreamSynchronize(idx->stream) != hipSuccess)
        r = fail(ORR_EDEVICE, "orr_index_delete_rows: device update failed");
    tmp.release();
    ORR_TRY(r);
    ORR_TRY(idx->d_dead.reserve(sizeof(int64_t) * merged.size()));
    HIP_TRY(hipMemcpy(idx->d_dead.p, merged.data(), sizeof(int64_t) * merged.size(), hipMemcpyHostToDevice));
    idx->dead.swap(merged);
    { std::lock_guard<std::mutex> pl(idx->lanes_mu); idx->dead_count_pub = (int64_t)idx->dead.size(); }
    if (out_deleted) *out_deleted = (int64_t)fresh.size();
    return ORR_OK;
}

int orr_index_compact(orr_index *idx, int64_t *out_removed)
{
    if (out_removed) *out_removed = 0;
    if (!idx) return fail(ORR_EINVAL, "orr_index_compact: null index");
    if (idx->is_view) return fail(ORR_EINVAL, "orr_index_compact: compact the owning index, not a view");
    std::vector<orr_index *> old_lanes;
    AllLanes all(idx);                                 // no search in flight while rows move
    std::lock_guard<std::mutex> lock(idx->mu);
    if (!idx->sealed) return fail(ORR_ESTATE, "orr_index_compact: the index is not sealed");
    if (idx->user_views.load() > 0)
        return fail(ORR_ESTATE, "orr_index_compact: %d view(s) of this index are alive (orr_index_view): destroy them first", idx->user_views.load());
    if (idx->dead.empty()) return ORR_OK;
    HIP_TRY(hipSetDevice(idx->device));
    hipStream_t s = idx->stream;
    const int64_t n = idx->n_rows, n_dead = (int64_t)idx->dead.size(), n_new = n - n_dead;
    // live positions, ascending (= the new candidate order), and how far each old position moves up
    std::vector<int64_t> live((size_t)n_new);
    std::vector<uint32_t> shift((size_t)n + 1);
    {
        size_t d = 0, w = 0;
        for (int64_t p = 0; p < n; ++p) {
            shift[(size_t)p] = (uint32_t)d;
            if (d < idx->dead.size() && idx->dead[d] == p) { ++d; continue; }
            live[w++] = p;
        }
        shift[(size_t)n] = (uint32_t)d;
    }
    DevBuf d_live;
    ORR_TRY(d_live.reserve(sizeof(int64_t) * (size_t)std::max<int64_t>(n_new, 1)));
    int r = ORR_OK;
    auto body = [&]() -> int {
        if (n_new > 0) HIP_TRY(hipMemcpy(d_live.p, live.data(), sizeof(int64_t) * (size_t)n_new, hipMemcpyHostToDevice));
        // ---- embeddings: IN PLACE, chunk by chunk through a bounce buffer (a second copy of a 150 GB shard does not fit).  Rows only
        // move towards lower positions, and chunks go in ascending order, so a chunk's destination never reaches rows a later chunk
        // still has to read.
        if (idx->dim > 0 && n_new > 0) {
            const int64_t first_moved = idx->dead.front();                  // rows in front of the first deleted one stay where they are
            const int64_t chunk = std::max<int64_t>(1, ((int64_t)256 << 20) / ((int64_t)sizeof(float) * idx->dim));
            DevBuf bounce;
            ORR_TRY(bounce.reserve(sizeof(float) * (size_t)chunk * idx->dim));
            int64_t w0 = first_moved;                                       // live[w0] is the first row that moves (w0 rows in front of it are live)
            w0 = (int64_t)(std::lower_bound(live.begin(), live.end(), first_moved) - live.begin());
            for (int64_t w = w0; w < n_new; w += chunk) {
                const int64_t m = std::min<int64_t>(chunk, n_new - w);
                const hipError_t e1 = orr::launch_gather_rows_f32(idx->d_emb, bounce.as<float>(), d_live.as<int64_t>() + w, m, idx->dim, s);
                if (e1 != hipSuccess) { bounce.release(); return fail(ORR_EDEVICE, "orr_index_compact: gather failed: %s", hipGetErrorString(e1)); }
                const hipError_t e2 = hipMemcpyAsync(idx->d_emb + (size_t)w * idx->dim, bounce.p, sizeof(float) * (size_t)m * idx->dim, hipMemcpyDeviceToDevice, s);
                if (e2 != hipSuccess) { bounce.release(); return fail(ORR_EDEVICE, "orr_index_compact: copy failed: %s", hipGetErrorString(e2)); }
            }
            const hipError_t e3 = hipStreamSynchronize(s);
            bounce.release();
            if (e3 != hipSuccess) return fail(ORR_EDEVICE, "orr_index_compact: %s", hipGetErrorString(e3));
        }
        // ---- per-row scalars: gathered into new arrays (8 bytes per row each)
        int64_t *nc = nullptr, *nr = nullptr, *nn = nullptr;
        ORR_TRY(dev_alloc(&nc, (size_t)idx->cap_rows));
        ORR_TRY(dev_alloc(&nr, (size_t)idx->cap_rows));
        ORR_TRY(dev_alloc(&nn, (size_t)std::max<int64_t>(n, 1)));
        if (n_new > 0) {
            HIP_TRY(orr::launch_gather_i64(idx->d_created, nc, d_live.as<int64_t>(), n_new, s));
            HIP_TRY(orr::launch_gather_i64(idx->d_row_ids, nr, d_live.as<int64_t>(), n_new, s));
            HIP_TRY(orr::launch_gather_i64(reinterpret_cast<const int64_t *>(idx->d_norm_b), nn, d_live.as<int64_t>(), n_new, s));   // (the norms as bit patterns)
        }
        HIP_TRY(hipStreamSynchronize(s));
        (void)hipFree(idx->d_created); idx->d_created = nc;
        (void)hipFree(idx->d_row_ids); idx->d_row_ids = nr;
        (void)hipFree(idx->d_norm_b); idx->d_norm_b = reinterpret_cast<double *>(nn);
        // ---- token index: every posting list loses the deleted positions and is renumbered (on the host: one pass over the postings)
        if (idx->n_postings > 0) {
            const size_t V = (size_t)idx->n_tokens;
            std::vector<uint64_t> off(V + 1), noff(V + 1);
            std::vector<uint32_t> rows((size_t)idx->n_postings);
            HIP_TRY(hipMemcpy(off.data(), idx->d_post_off, sizeof(uint64_t) * (V + 1), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(rows.data(), idx->d_post_rows, sizeof(uint32_t) * rows.size(), hipMemcpyDeviceToHost));
            size_t w = 0;
            for (size_t t = 0; t < V; ++t) {
                noff[t] = w;
                for (uint64_t i = off[t]; i < off[t + 1]; ++i) {
                    const uint32_t p = rows[(size_t)i];
                    if (shift[(size_t)p + 1] != shift[(size_t)p]) continue;       // a deleted row
                    rows[w++] = p - shift[(size_t)p];
                }
            }
            noff[V] = w;
            HIP_TRY(hipMemcpy(idx->d_post_off, noff.data(), sizeof(uint64_t) * (V + 1), hipMemcpyHostToDevice));
            if (w) HIP_TRY(hipMemcpy(idx->d_post_rows, rows.data(), sizeof(uint32_t) * w, hipMemcpyHostToDevice));
            idx->n_postings = w;
        }
        return ORR_OK;
    };
    r = body();
    d_live.release();
    if (r != ORR_OK) return r;                         // (a failure half way leaves the shard unusable: the caller rebuilds it)
    // ---- host mirrors
    {
        std::vector<int64_t> hc((size_t)n_new);
        std::vector<uint32_t> hl((size_t)n_new);
        for (int64_t w = 0; w < n_new; ++w) { hc[(size_t)w] = idx->h_created[(size_t)live[(size_t)w]]; hl[(size_t)w] = idx->h_clen[(size_t)live[(size_t)w]]; }
        idx->h_created.swap(hc);
        idx->h_clen.swap(hl);
        idx->h_cprefix.assign((size_t)n_new + 1, 0);
        for (int64_t w = 0; w < n_new; ++w) idx->h_cprefix[(size_t)w + 1] = idx->h_cprefix[(size_t)w] + idx->h_clen[(size_t)w];
    }
    idx->n_rows = n_new;
    idx->dead.clear();
    idx->d_dead.release();
    idx->id_index.clear();
    // derived copies are rebuilt at the next search that wants them; the lanes' borrowed pointers die with the lanes
    idx->emb_shadow.release(); idx->shadow_ready = false; idx->shadow_failed = false;
    idx->emb_i8.release(); idx->i8_scale.release(); idx->i8_rel_err.release(); idx->i8_rel_hat.release(); idx->i8_rowf.release();
    idx->i8_ready = false; idx->i8_failed = false;
    idx->bitmaps_clean = 0; idx->bitmaps_clean_of = nullptr;
    idx->tok_bm.release(); idx->tok_bm_index.release(); idx->n_tok_bm = -1; idx->tok_bm_words = 0;
    {
        std::lock_guard<std::mutex> ll(idx->lanes_mu);
        old_lanes.swap(idx->lanes);
        idx->lane_busy.clear();
        idx->dead_count_pub = 0;
    }
    for (orr_index *l : old_lanes) if (l) orr_index_destroy(l);
    if (out_removed) *out_removed = n_dead;
    return ORR_OK;
}

int orr_index_set_option(orr_index *idx, const char *name, int64_t value)
{
    if (!idx || !name) return fail(ORR_EINVAL, "orr_index_set_option: null argument");
    AllLanes all(idx);                                 // options apply to every lane of the index
    std::lock_guard<std::mutex> lock(idx->mu);
    auto lanes = [&](auto &&fn) { fn(idx); for (orr_index *l : idx->lanes) if (l) fn(l); };
    if (strcmp(name, "fuse_epilogue") == 0) { lanes([&](orr_index *x) { x->opt_fuse_epilogue = value != 0; }); return ORR_OK; }
    if (strcmp(name, "dead_rows_before") == 0) {
        if (value < 0) return fail(ORR_EINVAL, "orr_index_set_option: dead_rows_before must be >= 0");
        lanes([&](orr_index *x) { x->dead_before = value; });
        { std::lock_guard<std::mutex> pl(idx->lanes_mu); idx->dead_before_pub = value; }
        return ORR_OK;
    }
    if (strcmp(name, "kw_hits_cap") == 0) {
        if (value < 1 || value > (int64_t)0x7FFFFFFF) return fail(ORR_EINVAL, "orr_index_set_option: kw_hits_cap must be in 1 .. 2^31-1");
        lanes([&](orr_index *x) { x->kw_hits_cap = (uint32_t)value; });
        return ORR_OK;
    }
    if (strcmp(name, "max_lanes") == 0) {
        if (value < 1 || value > 16) return fail(ORR_EINVAL, "orr_index_set_option: max_lanes must be in 1 .. 16");
        if (idx->is_view) return fail(ORR_EINVAL, "orr_index_set_option: max_lanes applies to the owning index");
        idx->max_lanes = std::max<int>((int)value, 1 + (int)idx->lanes.size());      // (lanes that exist stay)
        return ORR_OK;
    }
    if (strcmp(name, "shard_topk") == 0) {
        if (value < 0 || value > 1 << 30) return fail(ORR_EINVAL, "orr_index_set_option: shard_topk must be >= 0");
        lanes([&](orr_index *x) { x->opt_shard_topk = (int)value; });
        return ORR_OK;
    }
    if (strcmp(name, "shard_pass") == 0) {
        if (value < 0 || value > 2) return fail(ORR_EINVAL, "orr_index_set_option: shard_pass takes 0, 1 or 2");
        lanes([&](orr_index *x) { x->opt_shard_pass = (int)value; });
        return ORR_OK;
    }
    if (strcmp(name, "two_stage") == 0) {
        if (value < 0 || value > 2) return fail(ORR_EINVAL, "orr_index_set_option: two_stage takes 0, 1 or 2");
        idx->opt_two_stage = (int)value;
        if (value == 1) {
            HIP_TRY(hipSetDevice(idx->device));
            ORR_TRY(ensure_i8_shadow(idx));
            if (!idx->i8_ready) ORR_TRY(ensure_shadow(idx));
        }
        for (orr_index *l : idx->lanes) {               // the lanes borrow whatever shadow exists now
            if (!l) continue;
            l->opt_two_stage = (int)value;
            l->emb_shadow.p = idx->emb_shadow.p; l->shadow_ready = idx->shadow_ready; l->shadow_failed = !idx->shadow_ready;
            l->emb_i8.p = idx->emb_i8.p; l->i8_scale.p = idx->i8_scale.p; l->i8_rel_err.p = idx->i8_rel_err.p;
            l->i8_rel_hat.p = idx->i8_rel_hat.p; l->i8_rowf.p = idx->i8_rowf.p;
            l->i8_ready = idx->i8_ready; l->i8_failed = !idx->i8_ready;
        }
        return ORR_OK;
    }
    return fail(ORR_EINVAL, "orr_index_set_option: unknown option %s", name);
}

static int make_view(orr_index *parent, orr_index **out, bool internal);

int orr_index_view(orr_index *parent, orr_index **out) { return make_view(parent, out, false); }

static int make_view(orr_index *parent, orr_index **out, bool internal)
{
    if (!parent || !out) return fail(ORR_EINVAL, "orr_index_view: null argument");
    *out = nullptr;
    std::lock_guard<std::mutex> lock(parent->mu);
    if (!parent->sealed) return fail(ORR_ESTATE, "orr_index_view: the index is not sealed");
    if (parent->is_view) return fail(ORR_EINVAL, "orr_index_view: take views of the owning index");
    HIP_TRY(hipSetDevice(parent->device));
    // the shadow is shared, so it has to exist before the view does -- but only shards the two-stage pass applies to get one
    if (parent->opt_two_stage == 1 && parent->n_rows >= 48 * (int64_t)orr::kSelSegRows) {
        ORR_TRY(ensure_i8_shadow(parent));
        if (!parent->i8_ready) ORR_TRY(ensure_shadow(parent));
    }
    ORR_TRY(ensure_vlong(parent));             // (before the view exists: a failure here must not leak it)
    ORR_TRY(ensure_token_bitmaps(parent));
    orr_index *v = new (std::nothrow) orr_index();
    if (!v) return fail(ORR_ENOMEM, "out of host memory");
    v->is_view = true;
    v->n_vlong = parent->n_vlong;
    v->n_vmid = parent->n_vmid;
    v->n_tok_bm = parent->n_tok_bm; v->tok_bm_words = parent->tok_bm_words;
    v->tok_bm.p = parent->tok_bm.p; v->tok_bm_index.p = parent->tok_bm_index.p;                                        // borrowed
    v->vlong_start.p = parent->vlong_start.p; v->vlong_len.p = parent->vlong_len.p; v->vlong_id.p = parent->vlong_id.p;   // borrowed
    v->parent = parent; v->dead_before = parent->dead_before;
    v->device = parent->device; v->dim = parent->dim; v->row_base = parent->row_base;
    v->n_rows = parent->n_rows; v->cap_rows = parent->cap_rows;
    v->d_emb = parent->d_emb; v->d_created = parent->d_created; v->d_row_ids = parent->d_row_ids; v->d_norm_b = parent->d_norm_b;
    v->n_tokens = parent->n_tokens; v->d_vpool = parent->d_vpool; v->d_vstart = parent->d_vstart; v->d_vlen = parent->d_vlen;
    v->d_post_off = parent->d_post_off; v->d_post_rows = parent->d_post_rows; v->n_postings = parent->n_postings;
    v->sealed = true;
    v->opt_fuse_epilogue = parent->opt_fuse_epilogue; v->opt_two_stage = parent->opt_two_stage;
    v->emb_shadow.p = parent->emb_shadow.p; v->emb_shadow.cap = 0;           // borrowed, never freed here
    v->shadow_ready = parent->shadow_ready; v->shadow_failed = !parent->shadow_ready;
    v->emb_i8.p = parent->emb_i8.p; v->i8_scale.p = parent->i8_scale.p; v->i8_rel_err.p = parent->i8_rel_err.p;
    v->i8_rel_hat.p = parent->i8_rel_hat.p; v->i8_rowf.p = parent->i8_rowf.p;   // borrowed as well
    v->i8_ready = parent->i8_ready; v->i8_failed = !parent->i8_ready;
    if (hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&v->stream_kw, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&v->stream_aux, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&v->ev_bm_clean, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&v->ev_inputs, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&v->ev_kw_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&v->ev_main_ready, hipEventDisableTiming) != hipSuccess ||
        !create_range_events(v->ev_range) ||
        hipEventCreateWithFlags(&v->ev_q, hipEventDisableTiming) != hipSuccess) {
        v->internal_lane = true;                       // (not counted yet: the destroy must not count it down)
        orr_index_destroy(v);
        return fail(ORR_EDEVICE, "cannot create streams on device %d", parent->device);
    }
    v->internal_lane = internal;
    if (!internal) parent->user_views.fetch_add(1);
    *out = v;
    return ORR_OK;
}

int orr_index_screen_dots(orr_index *idx, int32_t B, int32_t dim, const float *q, float *out)
{
    if (!idx || !q || !out || B <= 0) return fail(ORR_EINVAL, "orr_index_screen_dots: bad argument");
    AllLanes all(idx);
    std::lock_guard<std::mutex> lock(idx->mu);
    if (!idx->sealed) return fail(ORR_ESTATE, "orr_index_screen_dots: the index is not sealed");
    if (dim != idx->dim || dim <= 0 || dim % 64 != 0) return fail(ORR_EINVAL, "orr_index_screen_dots: dim must equal the index dimension and be a multiple of 64");
    if (idx->n_rows <= 0) return ORR_OK;
    HIP_TRY(hipSetDevice(idx->device));
    ORR_TRY(ensure_shadow(idx));
    if (!idx->shadow_ready) return fail(ORR_ENOMEM, "orr_index_screen_dots: the bf16 shadow does not fit in device memory");
    hipStream_t s = idx->stream;
    ORR_TRY(idx->ws_q.reserve(sizeof(float) * (size_t)B * dim));
    ORR_TRY(idx->ws_qtiled.reserve(orr::bf16_tiled_bytes(B, dim)));
    ORR_TRY(idx->ws_dotf.reserve(sizeof(float) * (size_t)B * (size_t)idx->n_rows));
    HIP_TRY(hipMemcpyAsync(idx->ws_q.p, q, sizeof(float) * (size_t)B * dim, hipMemcpyDefault, s));
    HIP_TRY(orr::launch_bf16_tiled(idx->ws_q.as<float>(), B, dim, idx->ws_qtiled.p, s));
    {
        Timed t(idx, "screen_bf16", 2.0 * (double)idx->n_rows * dim + 2.0 * (double)B * dim + 4.0 * (double)B * (double)idx->n_rows);
        HIP_TRY(orr::launch_screen_bf16(idx->ws_qtiled.p, B, idx->emb_shadow.p, 0, idx->n_rows, dim, idx->ws_dotf.as<float>(), idx->n_rows,
                                        nullptr, s));
    }
    HIP_TRY(hipMemcpyAsync(out, idx->ws_dotf.p, sizeof(float) * (size_t)B * (size_t)idx->n_rows, hipMemcpyDefault, s));
    HIP_TRY(hipStreamSynchronize(s));
    collect_events(idx);
    return ORR_OK;
}

int orr_index_screen_i8_dots(orr_index *idx, int32_t B, int32_t dim, const float *q, int32_t form, int32_t nt_rows, int32_t *out_dots,
                             int8_t *out_iq, int8_t *out_ie)
{
    if (!idx || !q || B <= 0 || form < 0 || form > 2) return fail(ORR_EINVAL, "orr_index_screen_i8_dots: bad argument");
    AllLanes all(idx);
    std::lock_guard<std::mutex> lock(idx->mu);
    if (!idx->sealed) return fail(ORR_ESTATE, "orr_index_screen_i8_dots: the index is not sealed");
    if (dim != idx->dim || dim <= 0 || dim % 128 != 0) return fail(ORR_EINVAL, "orr_index_screen_i8_dots: dim must equal the index dimension and be a multiple of 128");
    if (form > 0 && dim / 64 <= 6) return fail(ORR_EINVAL, "orr_index_screen_i8_dots: the four-wave forms need dim >= 448");
    if (idx->n_rows <= 0) return ORR_OK;
    HIP_TRY(hipSetDevice(idx->device));
    ORR_TRY(ensure_i8_shadow(idx));
    if (!idx->i8_ready) return fail(ORR_ENOMEM, "orr_index_screen_i8_dots: the int8 shadow does not fit in device memory");
    hipStream_t s = idx->stream;
    const size_t n = (size_t)idx->n_rows;
    ORR_TRY(idx->ws_q.reserve(sizeof(float) * (size_t)B * dim));
    ORR_TRY(idx->ws_q8.reserve(2 * (size_t)B * dim));
    ORR_TRY(idx->ws_q8s1.reserve(sizeof(float) * (size_t)B));
    ORR_TRY(idx->ws_q8err.reserve(2 * sizeof(double) * (size_t)B));
    ORR_TRY(idx->ws_qtiled.reserve(orr::i8_tiled_bytes(B, dim)));
    HIP_TRY(hipMemcpyAsync(idx->ws_q.p, q, sizeof(float) * (size_t)B * dim, hipMemcpyDefault, s));
    // the same quantisation and tiling the searches use (one int8 level per query)
    HIP_TRY(orr::launch_i8_queries(idx->ws_q.as<float>(), B, dim, idx->ws_q8.p, idx->ws_q8s1.as<float>(), idx->ws_q8err.as<double>(), s,
                                   idx->ws_q8err.as<double>() + B));
    HIP_TRY(orr::launch_i8_tile_queries(idx->ws_q8.p, B, dim, idx->ws_qtiled.p, s));
    if (out_dots) {
        ORR_TRY(idx->ws_dotf.reserve(sizeof(int32_t) * (size_t)B * n));
        HIP_TRY(hipMemsetAsync(idx->ws_dotf.p, 0xAB, sizeof(int32_t) * (size_t)B * n, s));     // (an element the kernel never writes shows up as 0xABABABAB)
        ORR_TRY(idx->ws_tickets.reserve(sizeof(uint32_t) * 8 * 16));
        HIP_TRY(hipMemsetAsync(idx->ws_tickets.p, 0, sizeof(uint32_t) * 8 * 16, s));
        {
            Timed t(idx, form == 0 ? "screen_i8_dots_w8" : form == 1 ? "screen_i8_dots_w4" : "screen_i8_dots_w16",
                    1.0 * (double)n * dim + 1.0 * (double)B * dim + 4.0 * (double)B * (double)n);
            HIP_TRY(orr::launch_screen_i8_dots_raw(idx->ws_qtiled.p, B, idx->emb_i8.p, idx->n_rows, dim, idx->ws_dotf.as<int32_t>(),
                                                   idx->n_rows, form, nt_rows != 0, s, idx->ws_tickets.as<uint32_t>()));
        }
        HIP_TRY(hipMemcpyAsync(out_dots, idx->ws_dotf.p, sizeof(int32_t) * (size_t)B * n, hipMemcpyDefault, s));
    }
    if (out_iq) HIP_TRY(hipMemcpyAsync(out_iq, idx->ws_q8.p, (size_t)B * dim, hipMemcpyDefault, s));
    if (out_ie) {
        ORR_TRY(idx->ws_raw.reserve(n * (size_t)dim));
        HIP_TRY(orr::launch_i8_untile(idx->emb_i8.p, idx->n_rows, dim, idx->ws_raw.p, s));
        HIP_TRY(hipMemcpyAsync(out_ie, idx->ws_raw.p, n * (size_t)dim, hipMemcpyDefault, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    collect_events(idx);
    return ORR_OK;
}

int orr_index_set_profiling(orr_index *idx, int32_t enabled)
{
    if (!idx) return fail(ORR_EINVAL, "null index");
    AllLanes all(idx);
    std::lock_guard<std::mutex> lock(idx->mu);
    const int level = enabled == 2 ? 2 : (enabled != 0 ? 1 : 0);
    idx->profiling = level;
    idx->stats.clear();
    for (orr_index *l : idx->lanes) if (l) { l->profiling = level; l->stats.clear(); }
    return ORR_OK;
}

int orr_index_kernel_stats(orr_index *idx, orr_kernel_stat *out, int32_t cap)
{
    if (!idx) return fail(ORR_EINVAL, "null index");
    AllLanes all(idx);
    std::lock_guard<std::mutex> lock(idx->mu);
    std::vector<KernelStat> sum = idx->stats;           // every lane's launches, by kernel name
    for (orr_index *l : idx->lanes) {
        if (!l) continue;
        for (const KernelStat &ks : l->stats) {
            size_t i = 0;
            while (i < sum.size() && sum[i].name != ks.name) ++i;
            if (i == sum.size()) { sum.push_back(ks); continue; }
            sum[i].launches += ks.launches; sum[i].total_ms += ks.total_ms; sum[i].algo_bytes += ks.algo_bytes;
        }
    }
    const int32_t n = (int32_t)sum.size();
    for (int32_t i = 0; i < n && i < cap && out; ++i) {
        memset(&out[i], 0, sizeof(orr_kernel_stat));
        strncpy(out[i].name, sum[(size_t)i].name.c_str(), sizeof(out[i].name) - 1);
        out[i].launches = sum[(size_t)i].launches;
        out[i].total_ms = sum[(size_t)i].total_ms;
        out[i].algo_bytes = sum[(size_t)i].algo_bytes;
    }
    return n;
}

}  // extern "C"

namespace {

struct BatchArgs {
    int32_t B, dim;
    const float *q;
    const uint8_t *terms_utf8;
    const uint32_t *term_off;
    const uint32_t *query_term_off;
    int64_t now_ticks;
    int64_t candidate_limit;
    int32_t topk = 10;             // the caller's k (two-stage floor); kprime is passed separately
    bool force_exact = false;      // skip the MFMA candidate pass (escalation after a failed certificate)
    mutable bool used_mfma = false; // set by run_shard
    bool no_fuse = false;          // keep the batched pass unfused (retry after a candidate-buffer overflow)
    const double *norms_host = nullptr; // exact normA of every query, already computed by the caller (orr_cluster: once for all shards)
    orr_candidate *out_dev = nullptr;   // orr_search_shard with a device-resident `out`: the kernels write the records there
    mutable bool used_fused = false;
    mutable bool used_two_stage = false;   // the pass kept survivors in per-query buffers (idx->h_survivors holds their counts)
};

const orr_index *owner_of(const orr_index *idx) { return idx->parent ? idx->parent : idx; }

int check_batch(const orr_index *idx, const BatchArgs &a, const char *fn)
{
    if (!idx) return fail(ORR_EINVAL, "%s: null index", fn);
    if (!idx->sealed) return fail(ORR_ESTATE, "%s: index is not sealed", fn);
    if (a.B <= 0) return fail(ORR_EINVAL, "%s: batch size must be positive", fn);
    if (a.dim < 0) return fail(ORR_EINVAL, "%s: negative query dimension", fn);
    if (a.dim > 0 && !a.q) return fail(ORR_EINVAL, "%s: q is NULL with dim %d", fn, a.dim);
    if (!a.query_term_off) return fail(ORR_EINVAL, "%s: query_term_off is required", fn);
    return ORR_OK;
}

// Rows of this shard that take part: the global candidate prefix clipped to the shard.
// Deleted rows do not count: the prefix ends behind the shard's local_live-th live row.
int64_t participating_rows(const orr_index *idx, int64_t candidate_limit)
{
    const int64_t limit = std::max<int64_t>(1, candidate_limit);     // Take(Math.Max(1, maxCount))
    const int64_t local_live = limit - (idx->row_base - idx->dead_before);
    if (local_live <= 0) return 0;
    int64_t p = local_live;                                          // smallest p with p - dead(< p) == local_live
    for (int64_t d : owner_of(idx)->dead) {
        if (d < p) ++p; else break;
        if (p >= idx->n_rows) break;
    }
    return std::min<int64_t>(p, idx->n_rows);
}

bool is_device_pointer(const void *p)
{
    hipPointerAttribute_t attr;
    memset(&attr, 0, sizeof(attr));
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();            // plain malloc'd memory: not known to the runtime
        return false;
    }
    return attr.type == hipMemoryTypeDevice;
}

// Rows of the sampled prefix of the two-stage pass, in selection segments (4096 rows each).  A larger
// sample gives a tighter floor (fewer survivors to re-score) and costs a larger ranking pass; the survivors
// of one query are about 2 k n / sample when scores are continuous (far fewer when keyword matches make
// them step-like), and they must stay well inside the 8192-entry buffers: k n / 2000 rows.  Measured (full
// hybrid scores): 1M rows x 1024 queries, 4 / 8 / 16 segments = 9.7 / 10.4 / 11.3 ms per batch; 12.5M rows x
// 1024 queries, 4 / 16 / 64 segments = 84.3 / 86.8 / 95.4 ms; the re-score stays below 0.4 ms throughout.
// The streaming form (1..8 queries) re-scores in parallel waves whose time does not grow with the number
// of survivors, so it goes down to two segments.
// boost (1..16, orr_index::sample_boost): scores without steps (cosine-only queries: config C4) leave 10..50 times as
// many survivors as hybrid ones, because the int8 bound is then comparable to the spacing of the scores around the
// floor; the index doubles the sample while the measured survivors per query stay above 4096 and halves it again
// below 512 (12.5M rows, 256 cosine-only queries, k' = 32: 16,000 survivors per query and 10 ms of exact re-scoring
// with the default sample of 200k rows against 14 ms for the screen itself; four times the sample, a quarter of both).
static int32_t sample_segments(int32_t n_seg_all, int64_t n, int32_t k, bool small_batch, int boost)
{
    const int64_t rows = (int64_t)std::max<int32_t>(1, k) * n / 2000 * std::max(1, boost);
    const int64_t segs = (rows + orr::kSelSegRows - 1) / orr::kSelSegRows;
    const int64_t most = std::min<int64_t>(64 * (int64_t)std::max(1, boost), std::max<int64_t>(n_seg_all / 8, 4));   // never more than an eighth of the rows
    return (int32_t)std::min<int64_t>(most, std::max<int64_t>(small_batch ? 2 : 4, segs));
}

// ORR_HOST_TIMING=1: where the host side of a search spends its time, printed to stderr every 64 calls (diagnostic,
// one searching thread).
struct HostTiming {
    bool on = getenv("ORR_HOST_TIMING") != nullptr;
    double acc[8] = {0};
    int64_t calls = 0;
    std::chrono::steady_clock::time_point last;
    void start() { if (on) last = std::chrono::steady_clock::now(); }
    void mark(int i)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        acc[i] += std::chrono::duration<double, std::milli>(now - last).count();
        last = now;
    }
    void done()
    {
        if (!on || ++calls % 64 != 0) return;
        fprintf(stderr, "[orr host ms/call] query_setup+cos_launch %.3f | keyword_prep+launch %.3f | norms+consts %.3f | select_launches %.3f | "
                        "wait_gpu %.3f | finish %.3f\n", acc[0] / 64, acc[1] / 64, acc[2] / 64, acc[3] / 64, acc[4] / 64, acc[5] / 64);
        for (double &x : acc) x = 0;
    }
};
thread_local HostTiming g_ht;          // per searching thread: lanes (views) and cluster workers each time their own calls

// ---- K3, the keyword side of one pass, on its own stream (with the int8 prefix the main stream has little to do before it
// needs the bitmaps, so this chain is the critical path of a batch): distinct terms -> vocabulary scan -> (term, token) hits ->
// per distinct term a row bitmap: a stored token bitmap where the term's only hit has one (ensure_token_bitmaps), else its
// posting lists OR-ed into the batch's own bitmaps (orr_token_index.cpp: why this equals RecallSearchService.cs:111).
struct KwSide {
    orr::KwView view{nullptr, 0, nullptr, nullptr, nullptr};
    size_t bm_bytes = 0, bm_clean_before = 0;     // the batch's own bitmaps: bytes used, bytes known to be zero on entry
    bool overflow_possible = false;               // the hit list may have been too short (checked behind the pass)
    uint32_t max_hits = 0;
};

int launch_keyword_side(orr_index *idx, const BatchArgs &a, const std::vector<uint32_t> &qoff, KwSide &out)
{
    const int32_t B = a.B;
    const uint32_t t_begin = qoff[0], n_terms_total = qoff[(size_t)B] - qoff[0];
    if (n_terms_total == 0) return ORR_OK;
    {

        // distinct terms of the batch: open addressing on an FNV-1a hash of the bytes (a batch of 1024 queries has
        // ~3000 terms; the node-based map this replaces cost as much as the GPU side of a small batch)
        std::vector<std::string_view> dterms;
        dterms.reserve(n_terms_total);
        uint32_t table_size = 16;
        while (table_size < 2 * n_terms_total + 2) table_size <<= 1;
        std::vector<uint32_t> table(table_size, 0xFFFFFFFFu);
        std::vector<uint32_t> qmeta((size_t)n_terms_total + (size_t)B + 1);   // [term -> distinct idx][query offsets]
        for (uint32_t t = 0; t < n_terms_total; ++t) {
            const uint32_t o = a.term_off[t_begin + t], e = a.term_off[t_begin + t + 1];
            if (e < o) return fail(ORR_EINVAL, "term_off is not monotone at term %u", t_begin + t);
            std::string_view sv(reinterpret_cast<const char *>(a.terms_utf8) + o, e - o);
            uint64_t h = 1469598103934665603ull;
            for (unsigned char ch : sv) h = (h ^ ch) * 1099511628211ull;
            uint32_t slot = (uint32_t)(h ^ (h >> 32)) & (table_size - 1);
            while (table[slot] != 0xFFFFFFFFu && dterms[table[slot]] != sv) slot = (slot + 1) & (table_size - 1);
            if (table[slot] == 0xFFFFFFFFu) {
                table[slot] = (uint32_t)dterms.size();
                dterms.push_back(sv);
            }
            qmeta[t] = table[slot];
        }
        for (int32_t b = 0; b <= B; ++b) qmeta[n_terms_total + b] = qoff[b] - t_begin;
        const uint32_t TT = (uint32_t)dterms.size();
        size_t pool_bytes = 0;
        for (auto &d : dterms) pool_bytes += d.size();
        // one pinned block, one upload: [ScanTerm x TT][qmeta][iota 0..64][term bytes]
        const size_t off_terms = 0;
        const size_t off_qmeta = off_terms + sizeof(orr::ScanTerm) * TT;
        const size_t off_iota = off_qmeta + sizeof(uint32_t) * qmeta.size();
        const size_t off_match = off_iota + sizeof(uint32_t) * 65;
        const size_t off_match8 = off_match + sizeof(orr::MatchTerm) * TT;
        const size_t off_lk = off_match8 + sizeof(orr::MatchTerm8) * TT;       // [8 u32 class boundaries][TT keys][TT term numbers]
        const size_t off_pool = off_lk + sizeof(uint32_t) * (8 + 2 * (size_t)TT);
        const size_t off_bloom = (off_pool + pool_bytes + 16 + 15) / 16 * 16;   // (last: only uploaded when the lookup form runs)
        // (the lookup pays from tens of millions of (token, term) pairs on: below, sorting the terms on the host -- 0.15 ms for the
        // 3,000 terms of 1024 queries -- costs more than comparing them all on the GPU)
        const bool vocab_lookup = [&] { static const int e = [] { const char *v = getenv("ORR_VOCAB_LOOKUP"); return v ? atoi(v) : -1; }();
                                        return e >= 0 ? e != 0 : (TT >= 64 && (int64_t)TT * idx->n_tokens >= ((int64_t)1 << 25)); }();
        const size_t meta_bytes = vocab_lookup ? off_bloom + orr::kVocabBloomBits / 8 : off_pool + pool_bytes + 16;
        ORR_TRY(idx->pin_meta.reserve(meta_bytes));
        ORR_TRY(idx->ws_meta.reserve(meta_bytes));
        uint8_t *hm = idx->pin_meta.as<uint8_t>();
        orr::ScanTerm *st = reinterpret_cast<orr::ScanTerm *>(hm + off_terms);
        uint32_t cursor = 0;
        for (uint32_t t = 0; t < TT; ++t) {
            st[t].off = cursor;
            st[t].len = (uint32_t)dterms[t].size();
            uint32_t pre = 0, msk = 0;
            for (uint32_t k = 0; k < 4 && k < st[t].len; ++k) {
                pre |= (uint32_t)(uint8_t)dterms[t][k] << (8 * k);
                msk |= 0xFFu << (8 * k);
            }
            st[t].prefix = pre;
            st[t].mask = msk;
            orr::MatchTerm &mt = reinterpret_cast<orr::MatchTerm *>(hm + off_match)[t];
            memset(&mt, 0, sizeof(mt));
            mt.len = st[t].len;
            for (uint32_t k = 0; k < 16 && k < st[t].len; ++k) {
                mt.w[k >> 2] |= (uint32_t)(uint8_t)dterms[t][k] << (8 * (k & 3));
                mt.m[k >> 2] |= 0xFFu << (8 * (k & 3));
            }
            orr::MatchTerm8 &m8 = reinterpret_cast<orr::MatchTerm8 *>(hm + off_match8)[t];
            memset(&m8, 0, sizeof(m8));
            m8.len = st[t].len;
            for (uint32_t k = 0; k < 32 && k < st[t].len; ++k) {
                m8.w[k >> 2] |= (uint32_t)(uint8_t)dterms[t][k] << (8 * (k & 3));
                m8.m[k >> 2] |= 0xFFu << (8 * (k & 3));
            }
            memcpy(hm + off_pool + cursor, dterms[t].data(), dterms[t].size());
            cursor += st[t].len;
        }
        if (vocab_lookup) {   // the terms of at most 16 bytes sorted by (class = min(bytes, 4), length-masked first dword): vocab_match_lookup
            uint32_t *lk = reinterpret_cast<uint32_t *>(hm + off_lk), *keys = lk + 8, *tix = keys + TT;
            const orr::MatchTerm *mts = reinterpret_cast<const orr::MatchTerm *>(hm + off_match);
            std::vector<std::pair<uint64_t, uint32_t>> order;
            order.reserve(TT);
            for (uint32_t t = 0; t < TT; ++t)
                if (st[t].len >= 1 && st[t].len <= 16)
                    order.emplace_back(((uint64_t)(std::min<uint32_t>(st[t].len, 4u) - 1u) << 32) | mts[t].w[0], t);
            std::sort(order.begin(), order.end());
            for (int c = 0; c <= 4; ++c) lk[c] = 0;
            for (size_t i = 0; i < order.size(); ++i) {
                keys[i] = (uint32_t)order[i].first;
                tix[i] = order[i].second;
                lk[(order[i].first >> 32) + 1] = (uint32_t)i + 1;
            }
            for (int c = 1; c <= 4; ++c) lk[c] = std::max(lk[c], lk[c - 1]);     // empty classes inherit the boundary in front of them
            lk[5] = lk[6] = lk[7] = 0;
            uint32_t *bloom = reinterpret_cast<uint32_t *>(hm + off_bloom);
            memset(bloom, 0, orr::kVocabBloomBits / 8);
            for (const auto &o : order) {
                const uint32_t hb = orr::vocab_bloom_hash((uint32_t)o.first, (uint32_t)(o.first >> 32));
                bloom[hb >> 5] |= 1u << (hb & 31u);
            }
        }
        memcpy(hm + off_qmeta, qmeta.data(), sizeof(uint32_t) * qmeta.size());
        uint32_t *iota = reinterpret_cast<uint32_t *>(hm + off_iota);
        for (uint32_t i = 0; i < 65; ++i) iota[i] = i;

        const int64_t V = idx->n_tokens;
        const int64_t words = ((idx->n_rows + 31) / 32 + 3) / 4 * 4;       // 16-byte aligned bitmaps
        const uint64_t want_hits = (uint64_t)std::max<int64_t>(V, 1) * TT;
        const uint32_t max_hits = (uint32_t)std::min<uint64_t>(want_hits, idx->kw_hits_cap);
        ORR_TRY(ensure_vlong(idx));
        ORR_TRY(ensure_token_bitmaps(idx));
        const int64_t VL = idx->n_vlong;
        ORR_TRY(idx->ws_vmatch.reserve(sizeof(uint16_t) * (size_t)TT * (size_t)std::max<int64_t>(VL, 1)));
        ORR_TRY(idx->ws_bitmaps.reserve(sizeof(uint32_t) * (size_t)TT * (size_t)words));
        // The term bitmaps must start out zero.  Every search clears what it used again when it is done (on this side
        // stream, behind its last kernel, while the host finishes the batch), so the next one only clears what lies
        // beyond: the memset (270 MB at 1024 queries x 1M rows) leaves the critical path of the keyword chain.
        out.bm_bytes = sizeof(uint32_t) * (size_t)TT * (size_t)words;
        size_t bm_clean = idx->bitmaps_clean_of == idx->ws_bitmaps.p ? idx->bitmaps_clean : 0;
        idx->bitmaps_clean = 0;                        // until this search has cleaned up after itself
        idx->bitmaps_clean_of = idx->ws_bitmaps.p;
        out.bm_clean_before = bm_clean;
        ORR_TRY(idx->ws_hits.reserve(sizeof(orr::KwHit) * (size_t)max_hits));
        ORR_TRY(idx->ws_counter.reserve(sizeof(unsigned long long)));
        ORR_TRY(idx->pin_kwcnt.reserve(sizeof(unsigned long long)));
        *idx->pin_kwcnt.as<unsigned long long>() = 0ull;
        hipStream_t k = idx->stream_kw;
        uint8_t *dm = idx->ws_meta.as<uint8_t>();
        HIP_TRY(hipMemcpyAsync(dm, hm, meta_bytes, hipMemcpyHostToDevice, k));
        // (the chain's counters were zeroed behind the last search on this stream, unless this is the first one or they moved)
        const bool counters_clean = idx->kw_counters_clean && idx->kw_counters_of[0] == idx->ws_counter.p;
        idx->kw_counters_clean = false;
        if (!counters_clean) HIP_TRY(hipMemsetAsync(idx->ws_counter.p, 0, sizeof(unsigned long long), k));
        bool bitmaps_settled = false;
        auto settle_bitmaps = [&]() -> int {
            if (bitmaps_settled) return ORR_OK;
            bitmaps_settled = true;
            if (idx->bm_clean_pending) {
                HIP_TRY(hipStreamWaitEvent(k, idx->ev_bm_clean, 0));
                idx->bm_clean_pending = false;
            }
            if (bm_clean < out.bm_bytes)
                HIP_TRY(hipMemsetAsync(static_cast<uint8_t *>(idx->ws_bitmaps.p) + bm_clean, 0, out.bm_bytes - bm_clean, k));
            return ORR_OK;
        };
        if (V > 0) {
            const orr::ScanTerm *d_terms = reinterpret_cast<const orr::ScanTerm *>(dm + off_terms);
            {   // tokens of at most 16 bytes: one lane per token against every distinct term, hits reserved in place; where tokens x terms
                // is large the terms are looked up (sorted by first dword) instead of compared one by one (ORR_VOCAB_LOOKUP=0|1 forces either)
                Timed t(idx, "vocab_match", 0.0, k);
                if (vocab_lookup) {
                    const uint32_t *lk = reinterpret_cast<const uint32_t *>(dm + off_lk);
                    HIP_TRY(orr::launch_vocab_match_lookup(idx->d_vpool, idx->d_vstart, idx->d_vlen, V,
                                                           reinterpret_cast<const orr::MatchTerm *>(dm + off_match), lk, lk + 8, lk + 8 + TT,
                                                           reinterpret_cast<const uint32_t *>(dm + off_bloom), idx->d_post_off,
                                                           idx->ws_counter.as<unsigned long long>(), idx->ws_hits.as<orr::KwHit>(), max_hits, k));
                } else {
                    HIP_TRY(orr::launch_vocab_match_short(idx->d_vpool, idx->d_vstart, idx->d_vlen, V,
                                                          reinterpret_cast<const orr::MatchTerm *>(dm + off_match), (int32_t)TT, idx->d_post_off,
                                                          idx->ws_counter.as<unsigned long long>(), idx->ws_hits.as<orr::KwHit>(), max_hits, k));
                }
            }
            // tokens of 17..32 bytes (the front of the list of longer tokens): one lane per token as well (ORR_VOCAB_MID=0: through
            // the wave-per-token scan like the longer ones, A/B)
            static const bool mid_off = [] { const char *e = getenv("ORR_VOCAB_MID"); return e && atoi(e) == 0; }();
            const int64_t n_mid = mid_off ? 0 : std::min<int64_t>(idx->n_vmid, VL);
            if (n_mid > 0) {
                Timed t(idx, "vocab_match_mid", 0.0, k);
                HIP_TRY(orr::launch_vocab_match_mid(idx->d_vpool, idx->vlong_start.as<uint64_t>(), idx->vlong_len.as<uint32_t>(),
                                                    idx->vlong_id.as<uint32_t>(), n_mid, reinterpret_cast<const orr::MatchTerm8 *>(dm + off_match8),
                                                    (int32_t)TT, idx->d_post_off, idx->ws_counter.as<unsigned long long>(),
                                                    idx->ws_hits.as<orr::KwHit>(), max_hits, k));
            }
            if (VL - n_mid > 0) {   // the rest: every distinct term is its own 1-term "query" of the wave-per-token scan
                const int64_t VS = VL - n_mid;
                {
                    Timed t(idx, "vocab_scan", 0.0, k);
                    HIP_TRY(orr::launch_vocab_scan(idx->d_vpool, idx->vlong_start.as<uint64_t>() + n_mid, idx->vlong_len.as<uint32_t>() + n_mid, VS,
                                                   dm + off_pool, d_terms, (int32_t)TT, reinterpret_cast<const uint32_t *>(dm + off_iota),
                                                   idx->ws_vmatch.as<uint16_t>(), k));
                }
                {
                    Timed t(idx, "vocab_hits", 0.0, k);
                    HIP_TRY(orr::launch_vocab_hits(idx->ws_vmatch.as<uint16_t>(), VS, (int32_t)TT, idx->vlong_id.as<uint32_t>() + n_mid, idx->d_post_off,
                                                   idx->ws_counter.as<unsigned long long>(), idx->ws_hits.as<orr::KwHit>(), max_hits, k));
                }
            }
            // terms whose only hit is a token with a stored bitmap use that bitmap as it is (no expansion, nothing copied)
            const uint8_t *skip = nullptr;
            if (idx->n_tok_bm > 0 && idx->tok_bm_words == words) {
                const size_t o_tok = sizeof(uint32_t) * (size_t)TT, o_off = (2 * o_tok + 7) / 8 * 8, o_alias = o_off + sizeof(int64_t) * (size_t)TT;
                ORR_TRY(idx->ws_kwalias.reserve(o_alias + (size_t)TT + 16));
                uint8_t *wa = idx->ws_kwalias.as<uint8_t>();
                if (!(counters_clean && idx->kw_counters_of[1] == idx->ws_kwalias.p))
                    HIP_TRY(hipMemsetAsync(wa, 0, o_tok, k));              // the hit counts
                const int64_t delta = (int64_t)(idx->tok_bm.as<uint32_t>() - idx->ws_bitmaps.as<uint32_t>());   // words from the batch's bitmaps to the token store
                Timed t(idx, "kw_alias", 0.0, k);
                HIP_TRY(orr::launch_kw_alias(idx->ws_hits.as<orr::KwHit>(), idx->ws_counter.as<unsigned long long>(), max_hits, (int32_t)TT,
                                             idx->tok_bm_index.as<int32_t>(), delta, words, reinterpret_cast<uint32_t *>(wa),
                                             reinterpret_cast<uint32_t *>(wa + o_tok), reinterpret_cast<int64_t *>(wa + o_off), wa + o_alias, k));
                out.view.term_word_off = reinterpret_cast<const int64_t *>(wa + o_off);
                skip = wa + o_alias;
            }
            // the bitmaps are first written here: the clearing behind the last search (on the auxiliary stream) must be done,
            // and what lies beyond the cleared part is cleared now
            ORR_TRY(settle_bitmaps());
            {
                Timed t(idx, "expand_hits", 0.0, k);
                HIP_TRY(orr::launch_expand_hits(idx->ws_hits.as<orr::KwHit>(), idx->ws_counter.as<unsigned long long>(), max_hits,
                                                idx->d_post_rows, idx->ws_bitmaps.as<uint32_t>(), words, k,
                                                idx->pin_kwcnt.as<unsigned long long>(), skip));     // hits of this pass: statistics
            }
        }
        ORR_TRY(settle_bitmaps());                     // (a corpus without tokens: nothing expanded, the bitmaps are read all the same)
        HIP_TRY(hipEventRecord(idx->ev_kw_done, k));
        out.view.bitmaps = idx->ws_bitmaps.as<uint32_t>();
        out.view.words_per_term = words;
        out.view.q_term_idx = reinterpret_cast<const uint32_t *>(dm + off_qmeta);
        out.view.q_term_off = out.view.q_term_idx + n_terms_total;
        out.overflow_possible = want_hits > (uint64_t)max_hits;
        out.max_hits = max_hits;
    }
    return ORR_OK;
}

// Generic path for large k (topK beyond the 64 entries a wave keeps): every score of a query, a stable descending radix sort
// (hipCUB), the first k' rows as records -- query by query.
int run_large_k(orr_index *idx, const BatchArgs &a, int32_t kprime, int64_t n, const double *d_dot, const orr::KwView &kw,
                const orr::QueryConst *qc, orr_candidate *d_cand, hipStream_t s)
{
    const int32_t B = a.B;
    // generic large-k path: full stable sort of every score, query by query
    ORR_TRY(idx->ws_keys_a.reserve(sizeof(unsigned long long) * (size_t)n));
    ORR_TRY(idx->ws_keys_b.reserve(sizeof(unsigned long long) * (size_t)n));
    ORR_TRY(idx->ws_vals_a.reserve(sizeof(uint32_t) * (size_t)n));
    ORR_TRY(idx->ws_vals_b.reserve(sizeof(uint32_t) * (size_t)n));
    size_t tmp_bytes = 0;
    HIP_TRY(orr::sort_pairs_desc(nullptr, tmp_bytes, idx->ws_keys_a.as<unsigned long long>(),
                                 idx->ws_keys_b.as<unsigned long long>(), idx->ws_vals_a.as<uint32_t>(),
                                 idx->ws_vals_b.as<uint32_t>(), n, s));
    ORR_TRY(idx->ws_sort_tmp.reserve(tmp_bytes));
    for (int32_t b = 0; b < B; ++b) {
        const double *dq = d_dot ? d_dot + (size_t)b * n : nullptr;
        {
            Timed t(idx, "score_keys", (double)n * 36.0);
            HIP_TRY(orr::launch_score_keys(dq, idx->d_norm_b, idx->d_created, kw, b, qc[b], a.now_ticks, n,
                                           idx->ws_keys_a.as<unsigned long long>(), idx->ws_vals_a.as<uint32_t>(), s));
        }
        {
            Timed t(idx, "radix_sort_desc", (double)n * 12.0 * 2.0 * 8.0);
            size_t tb = idx->ws_sort_tmp.cap;
            HIP_TRY(orr::sort_pairs_desc(idx->ws_sort_tmp.p, tb, idx->ws_keys_a.as<unsigned long long>(),
                                         idx->ws_keys_b.as<unsigned long long>(), idx->ws_vals_a.as<uint32_t>(),
                                         idx->ws_vals_b.as<uint32_t>(), n, s));
        }
        HIP_TRY(orr::launch_records_from_sorted(idx->ws_keys_b.as<unsigned long long>(), idx->ws_vals_b.as<uint32_t>(),
                                                kprime, n, idx->row_base, d_dot, n, idx->d_norm_b, idx->d_created,
                                                idx->d_row_ids, kw, b, 1, d_cand + (size_t)b * (kprime + 1), s));
    }
    return ORR_OK;
}

// up to this many queries (a grid dimension) the tail of the two-stage pass is finish_survivors: one launch, two from 64 queries
// on (12.5M rows x 1024 queries: 0.26 ms against 0.70 for the four separate kernels below, which remain for dim % 256 != 0)
constexpr int kFinishFusedMaxB = 65535;
constexpr int kRetryPass = 1;          // run_shard_once: a workspace was too small and has been enlarged; the same pass again

// The tail of the two-stage pass: exact re-score of every buffered survivor (fp32 master, reference arithmetic), the best k' of
// them as records with their exact dots.  dim % 256 == 0: finish_survivors (four lanes per survivor, or a wave per survivor for
// the smallest batches; the workgroup that draws a query's last ticket -- or a second launch -- merges its lists and writes the
// records, straight into pinned host memory when the record set is small); else four launches.  The survivors' counts go back
// with the records (idx->pin_cnt).
int two_stage_tail(orr_index *idx, const BatchArgs &a, int32_t kprime, int64_t n, const float *d_q, const orr::KwView &kw,
                   const orr::FusedEpilogue &epi, uint32_t kCap, int32_t buf_lists, bool host_records, size_t rec_bytes,
                   orr_candidate **d_cand_io, bool *direct_host_io, hipStream_t s)
{
    const int32_t B = a.B;
    orr_candidate *d_cand = *d_cand_io;
    ORR_TRY(idx->ws_fdot.reserve(sizeof(double) * (size_t)B * kCap));
    ORR_TRY(idx->pin_cnt.reserve(sizeof(uint32_t) * (size_t)B));
    if (B <= kFinishFusedMaxB && idx->dim % 256 == 0) {
        // the tail in one launch; small record sets go straight into pinned host memory (they are final when written)
        if (host_records && !a.out_dev && rec_bytes <= (256u << 10)) {
            ORR_TRY(idx->pin_cand.reserve(rec_bytes));
            d_cand = idx->pin_cand.as<orr_candidate>();
            *direct_host_io = true;
        }
        Timed t(idx, "finish_survivors", 0.0);
        HIP_TRY(orr::launch_finish_survivors(idx->d_emb, idx->dim, d_q, B, idx->d_norm_b, idx->d_created, idx->d_row_ids, kw,
                                             idx->ws_qc.as<orr::QueryConst>(), a.now_ticks, epi.cnt, idx->ws_fcnt.as<uint32_t>() + 2 * B,
                                             kCap, epi.buf, idx->ws_fdot.as<double>(), idx->ws_sel.as<orr::SelEntry>(), kprime, n,
                                             idx->row_base, idx->ws_tsL.as<double>(), d_cand, idx->pin_cnt.as<uint32_t>(), s));
    } else {
        {
            Timed t(idx, "rescore_buffer_exact", 0.0);
            HIP_TRY(orr::launch_rescore_buffer_exact(idx->d_emb, idx->dim, d_q, B, idx->d_norm_b, idx->d_created, kw,
                                                     idx->ws_qc.as<orr::QueryConst>(), a.now_ticks, epi.cnt, kCap, epi.buf,
                                                     idx->ws_fdot.as<double>(), s));
        }
        {
            Timed t(idx, "buffer_to_lists", 0.0);
            HIP_TRY(orr::launch_buffer_to_lists(epi.buf, epi.cnt, kCap, B, 0, buf_lists, idx->ws_sel.as<orr::SelEntry>(), s));
        }
        {
            Timed t(idx, "select_final", (double)B * (double)buf_lists * orr::kSelWidth * sizeof(orr::SelEntry));
            HIP_TRY(orr::launch_select_final(idx->ws_sel.as<orr::SelEntry>(), buf_lists, B, kprime, n, idx->row_base,
                                             nullptr, nullptr, 0, idx->d_norm_b, idx->d_created, idx->d_row_ids, kw,
                                             0, 0.0, nullptr, epi.cnt, kCap, idx->ws_tsL.as<double>(), d_cand, s));
        }
        {   // the records' exact dots come out of the buffer: no second K6 pass
            Timed t(idx, "records_dot_from_buffer", 0.0);
            HIP_TRY(orr::launch_records_dot_from_buffer(epi.buf, idx->ws_fdot.as<double>(), epi.cnt, kCap, B, kprime, idx->row_base,
                                                        d_cand, s));
        }
        // the survivors' counts go back with the records: per-query escalation and orr_index_search_stats
        HIP_TRY(hipMemcpyAsync(idx->pin_cnt.p, epi.cnt, sizeof(uint32_t) * (size_t)B, hipMemcpyDeviceToHost, s));
    }
    *d_cand_io = d_cand;
    return ORR_OK;
}

// The int8 screening GEMM (K2j) with the fused scoring epilogue over all participating rows, in n_ranges row ranges: the later
// ranges' count words are formed on the keyword stream while the earlier ranges are multiplied (released when the main
// stream gets here), every launch draws its output tiles from its own set of tickets.
int screen_i8_in_ranges(orr_index *idx, const BatchArgs &a, int64_t n, const orr::KwView &kw, orr::FusedEpilogue epi, int n_ranges,
                        const int64_t *range_row, double plane_bytes_per_row, hipStream_t s, bool tickets_cleared)
{
    const int32_t B = a.B;
    // algorithmic bytes: the int8 rows once, per row its constants (rowc 16 B, i8_rowf 16 B) and, with query terms,
    // 16 B of count words per 32 queries; the query image once
    if (n_ranges > 1) {
        // the later ranges' count words: released when the main stream reaches the first range's GEMM
        hipStream_t k = idx->stream_kw;
        HIP_TRY(hipEventRecord(idx->ev_main_ready, s));
        HIP_TRY(hipStreamWaitEvent(k, idx->ev_main_ready, 0));
        for (int r = 1; r < n_ranges; ++r) {
            {
                Timed t(idx, "count_planes", plane_bytes_per_row * (double)(range_row[r + 1] - range_row[r]), k);
                HIP_TRY(orr::launch_query_count_planes(kw, B, n, epi.plane_stride, idx->ws_fany.as<uint32_t>(), k, range_row[r],
                                                       range_row[r + 1], epi.count_bits == 2 ? 2 : 4));
            }
            HIP_TRY(hipEventRecord(idx->ev_range[r - 1], k));
        }
    }
    // output-tile tickets of the 16 x 16 x 64 form: eight counters per launch, cleared once per pass
    ORR_TRY(idx->ws_tickets.reserve(sizeof(uint32_t) * 8 * 16));
    if (!tickets_cleared) HIP_TRY(hipMemsetAsync(idx->ws_tickets.p, 0, sizeof(uint32_t) * 8 * 16, s));
    for (int r = 0; r < n_ranges; ++r) {
        if (r > 0) HIP_TRY(hipStreamWaitEvent(s, idx->ev_range[r - 1], 0));
        epi.tickets = idx->ws_tickets.as<uint32_t>() + 8 * r;
        const double rows_r = (double)(range_row[r + 1] - range_row[r]);
        Timed t(idx, "screen_i8_fused", rows_r * ((double)idx->dim + 32.0 + (epi.count_planes ? (epi.count_bits == 2 ? 8.0 : 16.0) * (double)((B + 31) / 32) : 0.0)) +
                                        1.0 * (double)B * idx->dim);
        HIP_TRY(orr::launch_screen_i8(idx->ws_qtiled.p, B, idx->emb_i8.p, range_row[r + 1], idx->dim, epi, s, range_row[r]));
    }
    return ORR_OK;
}

// Device side of one batch: exact dots, keyword bitmaps, fused scores, selection.
// Records ([B][kprime+1]) land in pinned host memory (*recs_host) when host_records is set
// and they are small, otherwise in idx->ws_cand (*recs_host = nullptr).  *q_host points at
// the query vectors in host memory (valid until the next call).  Caller holds the lock.
int run_shard_once(orr_index *idx, const BatchArgs &a, int32_t kprime, bool host_records, const float **q_host,
                   const orr_candidate **recs_host)
{
    g_ht.start();
    ORR_TRY(bind_device(idx));
    const int64_t n = participating_rows(idx, a.candidate_limit);
    const int32_t B = a.B;
    const bool use_cos = a.dim > 0 && a.dim == idx->dim;
    hipStream_t s = idx->stream;
    if (q_host) *q_host = nullptr;
    if (recs_host) *recs_host = nullptr;

    std::vector<uint32_t> qoff((size_t)B + 1);
    memcpy(qoff.data(), a.query_term_off, sizeof(uint32_t) * ((size_t)B + 1));
    for (int32_t b = 0; b < B; ++b) {
        if (qoff[b + 1] < qoff[b]) return fail(ORR_EINVAL, "query_term_off is not monotone at query %d", b);
        if (qoff[b + 1] - qoff[b] > 65535) return fail(ORR_EINVAL, "query %d has more than 65535 terms", b);
    }
    const uint32_t t_begin = qoff[0], t_end = qoff[B];
    const uint32_t n_terms_total = t_end - t_begin;
    if (n_terms_total > 0 && (!a.term_off || !a.terms_utf8)) return fail(ORR_EINVAL, "terms are referenced but term_off/terms_utf8 is NULL");

    // ---- record destination
    const size_t rec_count = (size_t)B * ((size_t)kprime + 1);
    const size_t rec_bytes = sizeof(orr_candidate) * rec_count;
    // Batched candidate pass on the matrix cores (K2) + exact re-score (K6) from this batch size
    // up; below it the HBM-bound exact kernel is as fast and needs no second pass.
    constexpr int mfma_min_batch = 5;
    // 1..8 queries over a large shard with a shadow in place: the streaming form of the two-stage pass
    // (stream over a sampled prefix -> floor, stream over all rows -> survivors, exact re-score)
    bool ts_stream = false, ts_i8 = false;
    if (use_cos && !a.force_exact && !a.no_fuse && idx->opt_two_stage == 1 && idx->dim % 64 == 0 && kprime <= orr::kSelWidth &&
        (n + orr::kSelSegRows - 1) / orr::kSelSegRows >= 48 && std::max<int32_t>(1, a.topk) <= orr::kSelWidth &&
        B <= orr::kMaxGemvScreenQ) {
        ORR_TRY(ensure_i8_shadow(idx));
        ts_i8 = idx->i8_ready;
        if (!ts_i8) ORR_TRY(ensure_shadow(idx));
        ts_stream = ts_i8 || idx->shadow_ready;
        // 5..8 queries on the int8 shadow: the screening GEMM with one live query tile is HBM-bound as well and
        // reads the rows once, the stream would need two launches (1M x 3072: 8 queries 1.39 -> 0.97 ms)
        if (ts_i8 && B > orr::kMaxI8ScreenQ) { ts_i8 = false; ts_stream = false; }
    }
    const bool use_mfma = use_cos && !a.force_exact && (B >= mfma_min_batch || ts_stream) && idx->dim % 64 == 0 && kprime <= orr::kSelWidth;
    const bool approx_pass = use_mfma;                   // records carry no dot yet: filled in exactly on the device
    a.used_mfma = approx_pass;
    a.used_fused = false;
    a.used_two_stage = false;
    idx->h_survivors.clear();
    bool direct_host = host_records && !approx_pass && rec_bytes <= (256u << 10);
    orr_candidate *d_cand = nullptr;
    if (direct_host) {
        ORR_TRY(idx->pin_cand.reserve(rec_bytes));
        d_cand = idx->pin_cand.as<orr_candidate>();        // pinned host memory is device-writable
    } else if (a.out_dev) {
        d_cand = a.out_dev;
    } else {
        ORR_TRY(idx->ws_cand.reserve(rec_bytes));
        d_cand = idx->ws_cand.as<orr_candidate>();
    }

    // ---- query vectors: the dot kernel reads them where they are (device) or from one upload
    const float *d_q = nullptr;
    bool q_download_pending = false;
    // Queries that already live on the device get their exact norms there (the kernel that computes the rows' norms):
    // no download of the vectors, no host pass over them.  The generic large-k path scores with host-side constants.
    bool dev_norms = false;
    idx->h_norm_a.assign((size_t)B, 0.0);
    if (use_cos) {
        const size_t qbytes = sizeof(float) * (size_t)B * a.dim;
        constexpr int dev_norm_min = 16;
        if (is_device_pointer(a.q) && kprime <= orr::kSelWidth && B >= dev_norm_min && !a.norms_host) {
            // (a handful of queries: the download and the host's pass cost less than the kernel's 3072-step chains)
            d_q = a.q;
            dev_norms = true;
            ORR_TRY(idx->ws_norm_a.reserve(sizeof(double) * (size_t)B));
            ORR_TRY(idx->pin_norm.reserve(sizeof(double) * (size_t)B));
            HIP_TRY(orr::launch_dot_exact(d_q, B, a.dim, nullptr, 1, true, idx->ws_norm_a.as<double>(), B, idx->stream_aux));   // beside the first cosine kernel
            HIP_TRY(hipEventRecord(idx->ev_q, idx->stream_aux));
        } else if (is_device_pointer(a.q)) {
            ORR_TRY(idx->pin_q.reserve(qbytes));
            d_q = a.q;
            HIP_TRY(hipMemcpyAsync(idx->pin_q.p, a.q, qbytes, hipMemcpyDeviceToHost, idx->stream_kw));
            HIP_TRY(hipEventRecord(idx->ev_q, idx->stream_kw));
            q_download_pending = true;
        } else {
            ORR_TRY(idx->pin_q.reserve(qbytes));
            memcpy(idx->pin_q.p, a.q, qbytes);
            ORR_TRY(idx->ws_q.reserve(qbytes));
            HIP_TRY(hipMemcpyAsync(idx->ws_q.p, idx->pin_q.p, qbytes, hipMemcpyHostToDevice, s));
            d_q = idx->ws_q.as<float>();
        }
        if (q_host && !dev_norms) *q_host = idx->pin_q.as<float>();
    }

    // 1..4 queries on the int8 shadow: their int8 images are the first thing the main stream needs and depend on nothing
    // else, so that launch goes out before the host prepares the keyword side; it also clears the pass's counters
    bool counters_cleared = false;
    if (n > 0 && ts_stream && ts_i8) {
        ORR_TRY(idx->ws_q8.reserve(2 * (size_t)B * idx->dim));
        ORR_TRY(idx->ws_q8s1.reserve(sizeof(float) * (size_t)B));
        ORR_TRY(idx->ws_q8err.reserve(sizeof(double) * (size_t)B));
        ORR_TRY(idx->ws_fcnt.reserve(sizeof(uint32_t) * 3 * (size_t)B));
        HIP_TRY(orr::launch_i8_queries(d_q, B, idx->dim, idx->ws_q8.p, idx->ws_q8s1.as<float>(), idx->ws_q8err.as<double>(), s, nullptr,
                                       idx->ws_fcnt.as<uint32_t>(), 3 * B));
        counters_cleared = true;
    }

    if (n == 0) {   // nothing on this shard takes part: empty records + trailers
        std::vector<orr_candidate> empty(rec_count);
        for (auto &c : empty) { memset(&c, 0, sizeof(c)); c.row_id = -1; c.order_key = -1; }
        for (int32_t b = 0; b < B; ++b) {
            orr_candidate &t = empty[(size_t)b * (kprime + 1) + kprime];
            t.approx_score = -std::numeric_limits<double>::infinity();
            t.order_key = 0; t.matches = 0; t.flags = ORR_CAND_TRAILER;
        }
        if (direct_host) memcpy(d_cand, empty.data(), rec_bytes);
        else HIP_TRY(hipMemcpyAsync(d_cand, empty.data(), rec_bytes, hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (q_download_pending || dev_norms) HIP_TRY(hipEventSynchronize(idx->ev_q));    // nothing reads the caller's vectors after the call
        if (recs_host && direct_host) *recs_host = d_cand;
        return ORR_OK;
    }

    // ---- K3 keyword side first, on its own stream
    KwSide kws;
    ORR_TRY(launch_keyword_side(idx, a, qoff, kws));
    const orr::KwView kw = kws.view;
    const size_t bm_bytes = kws.bm_bytes, bm_clean_before = kws.bm_clean_before;
    const bool kw_overflow_possible = kws.overflow_possible;
    const uint32_t kw_max_hits = kws.max_hits;

    g_ht.mark(1);
    // the hi/lo bf16 split of the queries, launched in front of the first kernel that reads it (the int8 forms never do)
    bool queries_split = false;
    auto need_split = [&]() -> int {
        if (queries_split) return ORR_OK;
        queries_split = true;
        ORR_TRY(idx->ws_qsplit.reserve(sizeof(float) * (size_t)B * idx->dim));
        HIP_TRY(orr::launch_split_queries(d_q, B, idx->dim, idx->ws_qsplit.p, s));
        return ORR_OK;
    };
    // ---- cosine numerators
    double *d_dot = nullptr;
    float *d_dotf = nullptr;
    double approx_eps = 0.0;
    bool bf16_split = false;
    int32_t fused_sample_seg = 0;      // > 0: fused epilogue behind a sampled prefix of that many segments
    bool two_stage = false;            // plain-bf16 first stage over all rows + exact second stage
    bool records_have_dots = false;    // two-stage: exact dots copied from the survivors' buffer
    bool ts_gemv = false;              // two-stage with the streaming screen (1..8 queries)
    bool prefix_i8 = false;            // the sampled prefix went through the int8 screening GEMM: its keys are lower bounds
    int pass_mode = 0;                 // orr_search_stats.pass_mode of this pass
    int64_t dotf_rows = n;             // columns of d_dotf
    if (use_cos && use_mfma) {
        if (!ts_stream && B <= 64) {
            ORR_TRY(idx->ws_dotf.reserve(sizeof(float) * (size_t)B * (size_t)n));
            d_dotf = idx->ws_dotf.as<float>();
        }
        // Where the two-stage pass applies it wins from the first MFMA batch on (1M x 3072 rows: 8 queries 2.19 ms
        // against 2.35 ms streaming, 32 queries 2.28 against 3.07, 64 queries 2.4 against 6.3); ORR_TS_MIN_BATCH moves that
        const bool ts_eligible = idx->opt_two_stage != 0 && !a.no_fuse && (n + orr::kSelSegRows - 1) / orr::kSelSegRows >= 48 &&
                                 std::max<int32_t>(1, a.topk) <= orr::kSelWidth;
        constexpr int ts_min_batch = 5;
        if (ts_stream) {
            // 1..8 queries: HBM-bound, so no GEMM tile: the streaming screen (K2i on the int8 shadow, else K2g on the
            // bf16 one) runs over the sample for the floor and then over all rows
            ts_gemv = true;
            two_stage = true;
            const int32_t n_seg_all = (int32_t)((n + orr::kSelSegRows - 1) / orr::kSelSegRows);
            fused_sample_seg = sample_segments(n_seg_all, n, a.topk, true, idx->sample_boost);
            dotf_rows = (int64_t)fused_sample_seg * orr::kSelSegRows;
            if (ts_i8) {
                // (quantised above, before the keyword side was prepared)
            } else {
                ORR_TRY(need_split());
            }
        } else if (B < ts_min_batch || (B <= 64 && !ts_eligible)) {   // HBM-bound streaming form over all rows, 32 queries per launch
            for (int32_t b0 = 0; b0 < B; b0 += 32) {
                const int32_t nq = std::min<int32_t>(32, B - b0);
                Timed t(idx, "gemv_mfma", 4.0 * (double)n * idx->dim + 4.0 * (double)nq * idx->dim + 4.0 * (double)nq * (double)n);
                HIP_TRY(orr::launch_gemv_mfma(d_q + (size_t)b0 * a.dim, nq, idx->d_emb, n, idx->dim, d_dotf + (size_t)b0 * n, n, s));
            }
        } else {
            {   // split bf16: queries split once; with enough rows the GEMM over everything behind a
                // sampled prefix runs with the fused scoring epilogue (launched further down, once the
                // floor keys exist) and only the prefix's dots go through HBM
                const int32_t n_seg_all = (int32_t)((n + orr::kSelSegRows - 1) / orr::kSelSegRows);
                two_stage = idx->opt_two_stage != 0 && !a.no_fuse && n_seg_all >= 48 && std::max<int32_t>(1, a.topk) <= orr::kSelWidth;
                fused_sample_seg = ((idx->opt_fuse_epilogue || two_stage) && !a.no_fuse && n_seg_all >= 48)
                                       ? sample_segments(n_seg_all, n, a.topk, false, idx->sample_boost) : 0;
                dotf_rows = fused_sample_seg > 0 ? (int64_t)fused_sample_seg * orr::kSelSegRows : n;
                ORR_TRY(idx->ws_dotf.reserve(sizeof(float) * (size_t)B * (size_t)dotf_rows));
                d_dotf = idx->ws_dotf.as<float>();
                // Where the int8 shadow exists the sampled prefix goes through the int8 screening GEMM as well (integer
                // dots out; fuse_select turns them into LOWER bounds of the scores with the per-pair bound), which reads
                // a quarter of the bytes of the split pass and runs at twice its MFMA rate.
                if (two_stage && fused_sample_seg > 0 && idx->opt_two_stage == 1 && idx->dim % 128 == 0) {
                    ORR_TRY(ensure_i8_shadow(idx));
                    prefix_i8 = idx->i8_ready;
                }
                if (prefix_i8) {
                    ORR_TRY(idx->ws_q8.reserve(2 * (size_t)B * idx->dim));
                    ORR_TRY(idx->ws_q8s1.reserve(sizeof(float) * (size_t)B));
                    ORR_TRY(idx->ws_q8err.reserve(2 * sizeof(double) * (size_t)B));
                    ORR_TRY(idx->ws_qtiled.reserve(orr::i8_tiled_bytes(B, idx->dim)));
                    HIP_TRY(orr::launch_i8_queries(d_q, B, idx->dim, idx->ws_q8.p, idx->ws_q8s1.as<float>(), idx->ws_q8err.as<double>(), s,
                                                   idx->ws_q8err.as<double>() + B));
                    HIP_TRY(orr::launch_i8_tile_queries(idx->ws_q8.p, B, idx->dim, idx->ws_qtiled.p, s));
                    const int64_t pre_rows = std::min<int64_t>(dotf_rows, n);
                    Timed t(idx, "screen_i8_prefix", 1.0 * (double)pre_rows * idx->dim + 1.0 * (double)B * idx->dim + 4.0 * (double)B * (double)pre_rows);
                    HIP_TRY(orr::launch_screen_i8_dots(idx->ws_qtiled.p, B, idx->emb_i8.p, pre_rows, idx->dim, d_dotf, dotf_rows, s));
                } else {
                    ORR_TRY(need_split());
                    Timed t(idx, "gemm_dot_bf16x3", 4.0 * (double)dotf_rows * idx->dim + 4.0 * (double)B * idx->dim + 4.0 * (double)B * (double)dotf_rows);
                    HIP_TRY(orr::launch_gemm_dot_bf16x3(idx->ws_qsplit.p, B, idx->d_emb, 0, dotf_rows, idx->dim, d_dotf, dotf_rows, nullptr, 3, s));
                    bf16_split = true;
                }
            }
        }
        // Error of the approximate dot against the reference sum, relative to sum|q_k e_k| (<= |q||e| by
        // Cauchy-Schwarz, which turns it into a bound on the cosine).  The matrix cores' internal
        // summation order and rounding mode are not documented, so every addition (and every fp32
        // product) is charged one full unit in the last place, 2^-23, of a partial sum that never
        // exceeds sum|terms|:
        //   f32 MFMA    D products + D additions                      -> (2 D + 2) 2^-23
        //   split bf16  3 D additions of exact products, plus the dropped lo*lo and the second-order
        //               residuals of the split (u = 2^-8 per bf16 rounding) -> 3.1 u^2 + 3.06 D 2^-23
        const double u23 = 1.1920928955078125e-07, u16 = 1.52587890625e-05;
        const double eps_cos = bf16_split ? 3.1 * u16 + 3.06 * (double)idx->dim * u23
                                          : (2.0 * (double)idx->dim + 2.0) * u23;
        approx_eps = 0.7 * 1.01 * eps_cos + 1e-12;
        // streaming form: the prefix is scored by the plain-bf16 stream itself, so its floor carries that bound
        if (ts_gemv) approx_eps = 0.7 * 1.01 * (0.0078125 * (1.0 + 0.001953125) + 1.02 * (double)idx->dim * u23) + 1e-12;
        if (ts_i8 || prefix_i8) approx_eps = 1e-12;    // int8 forms: the sample's keys are already lower bounds
    } else if (use_cos) {
        ORR_TRY(idx->ws_dot.reserve(sizeof(double) * (size_t)B * (size_t)n));
        d_dot = idx->ws_dot.as<double>();
        for (int32_t b0 = 0; b0 < B; b0 += orr::kMaxExactQ) {
            const int32_t nq = std::min<int32_t>(orr::kMaxExactQ, B - b0);
            Timed t(idx, "dot_exact", 4.0 * (double)n * idx->dim + 4.0 * nq * idx->dim + 8.0 * nq * (double)n);
            HIP_TRY(orr::launch_dot_exact(idx->d_emb, n, idx->dim, d_q + (size_t)b0 * a.dim, nq, false,
                                          d_dot + (size_t)b0 * n, n, s));
        }
    }

    g_ht.mark(0);
    // ---- per-query constants (exact normA needs the vectors on the host)
    const bool batched_score = (B >= 4 || ts_stream) && kprime <= orr::kSelWidth;     // per-row pieces once per batch
    if (q_download_pending) HIP_TRY(hipEventSynchronize(idx->ev_q));
    ORR_TRY(idx->pin_qc.reserve(sizeof(orr::QueryConst) * (size_t)B));
    ORR_TRY(idx->ws_qc.reserve(sizeof(orr::QueryConst) * (size_t)B));
    orr::QueryConst *qc = idx->pin_qc.as<orr::QueryConst>();
    if (use_cos && !dev_norms) {
        if (a.norms_host) memcpy(idx->h_norm_a.data(), a.norms_host, sizeof(double) * (size_t)B);
        else exact_norms(idx->pin_q.as<float>(), B, a.dim, idx->h_norm_a.data());
    }
    for (int32_t b = 0; b < B; ++b) {
        qc[b].use_cos = use_cos ? 1 : 0;
        qc[b].norm_a = idx->h_norm_a[(size_t)b];
        qc[b].n_terms = (int32_t)(qoff[b + 1] - qoff[b]);
        qc[b].inv_n_terms = qc[b].n_terms > 0 ? 1.0 / (double)qc[b].n_terms : 0.0;
        qc[b].inv_sqrt_na = 0.0;
        if (batched_score && use_cos && !dev_norms) {
            if (qc[b].norm_a <= 0.0) qc[b].use_cos = 0;                    // guard :84 -> cosine 0 for every row
            else qc[b].inv_sqrt_na = 1.0 / std::sqrt(qc[b].norm_a);        // NaN stays NaN
        }
    }
    if (dev_norms) {        // (the kernel that adds the norms reads the host's constants in place: no upload command)
        HIP_TRY(hipStreamWaitEvent(s, idx->ev_q, 0));
        HIP_TRY(orr::launch_patch_query_norms(idx->ws_qc.as<orr::QueryConst>(), idx->ws_norm_a.as<double>(), B, batched_score, s,
                                              idx->pin_norm.as<double>(), qc));
    } else {
        HIP_TRY(hipMemcpyAsync(idx->ws_qc.p, qc, sizeof(orr::QueryConst) * (size_t)B, hipMemcpyHostToDevice, s));
    }
    // per-row selection constants do not depend on the keyword side: enqueued before the main stream waits for it
    const double2 *d_rowc_early = nullptr;
    const bool rowc_inline = ts_stream && ts_i8;        // the int8 stream forms them in its epilogue (one or two uses per row)
    if (batched_score && kprime <= orr::kSelWidth && !rowc_inline) {
        ORR_TRY(idx->ws_rowc.reserve(sizeof(double2) * (size_t)n));
        Timed t(idx, "row_consts", 32.0 * (double)n);
        HIP_TRY(orr::launch_row_consts(idx->d_norm_b, idx->d_created, a.now_ticks, n, idx->ws_rowc.as<double2>(), s));
        d_rowc_early = idx->ws_rowc.as<double2>();
    }
    if (n_terms_total > 0) HIP_TRY(hipStreamWaitEvent(s, idx->ev_kw_done, 0));
    g_ht.mark(2);

    // ---- K4/K5 fused score + selection
    if (kprime <= orr::kSelWidth) {
        const int64_t n_seg = (n + orr::kSelSegRows - 1) / orr::kSelSegRows;
        ORR_TRY(idx->ws_sel.reserve(sizeof(orr::SelEntry) * (size_t)B * (size_t)n_seg * orr::kSelWidth));
        const double2 *d_rowc = d_rowc_early;
        const int32_t n_seg32 = (int32_t)n_seg;
        unsigned long long *d_tau = nullptr;
        if (fused_sample_seg > 0) {
            // ---- fused batched pass: prefix lists -> floor keys -> GEMM with the scoring epilogue ->
            // survivors' buffers -> lists; the final merge reads prefix lists + buffer lists
            a.used_fused = true;
            uint32_t kCap = idx->survivor_cap;                              // survivors kept per query (a multiple of 64)
            while (kCap > 8192 && (size_t)B * kCap * 40 > ((size_t)2 << 30)) kCap >>= 1;
            idx->pass_cap = kCap;
            const int32_t buf_lists = (int32_t)(kCap / orr::kSelWidth);
            const int32_t lists_total = fused_sample_seg + buf_lists;
            const int32_t finish_group = orr::finish_survivors_group(B, idx->dim);      // (finish_survivors' lists: one per group)
            const int32_t lists_room = std::max<int32_t>(lists_total, finish_group ? (int32_t)(kCap / (uint32_t)finish_group) : 0);
            ORR_TRY(idx->ws_sel.reserve(sizeof(orr::SelEntry) * (size_t)B * (size_t)lists_room * orr::kSelWidth));
            ORR_TRY(idx->ws_tau.reserve(sizeof(unsigned long long) * (size_t)B));
            ORR_TRY(idx->ws_fcnt.reserve(sizeof(uint32_t) * 3 * (size_t)B));      // [survivors][sampled prefix][workgroups done]
            ORR_TRY(idx->ws_fbuf.reserve(sizeof(orr::SelEntry) * (size_t)B * kCap));
            d_tau = idx->ws_tau.as<unsigned long long>();
            // 0: the prefix's rows are ranked in full; 1: per-lane maxima, sorted; 2: wave maxima, nothing sorted (where 16 per
            // segment are at least 8 k candidates per query)
            static const char *prefix_env = getenv("ORR_PREFIX_FULL_RANKING");        // =1: form 0; =2: form 1 (A/B)
            const int prefix_floor_form = !(two_stage && d_rowc != nullptr) || (prefix_env && atoi(prefix_env) == 1) ? 0
                                          : ((int64_t)fused_sample_seg * 16 >= 8 * (int64_t)std::max<int32_t>(1, a.topk) && !(prefix_env && atoi(prefix_env) == 2)) ? 2 : 1;
            if (!ts_gemv) {
                Timed t(idx, "fuse_select", (double)B * (double)dotf_rows * 28.0);
                orr::I8Prefix i8p;
                if (prefix_i8) { i8p.rowf = idx->i8_rowf.as<float4>(); i8p.qs1 = idx->ws_q8s1.as<float>(); i8p.qerr2 = idx->ws_q8err.as<double>() + B; }
                HIP_TRY(orr::launch_fuse_select(nullptr, d_dotf, dotf_rows, idx->d_norm_b,
                                                idx->d_created, d_rowc, kw, idx->ws_qc.as<orr::QueryConst>(), a.now_ticks,
                                                std::min<int64_t>(dotf_rows, n), B, 0, fused_sample_seg, nullptr, idx->ws_sel.as<orr::SelEntry>(),
                                                lists_total, s, i8p,
                                                // two-stage: the prefix only yields the floor (the records come out of the survivors' buffers)
                                                prefix_floor_form));
            }
            ORR_TRY(idx->ws_fqf.reserve(sizeof(float4) * 2 * (size_t)B));
            orr::FusedEpilogue epi{};
            epi.count_planes = nullptr;
            epi.plane_stride = (n + 63) / 64 * 64;
            // Large shards go through the int8 screening GEMM in FOUR ROW RANGES: only the first range's count words are formed
            // in front of the GEMM; those of the later ranges are formed on the keyword stream while the earlier ranges are
            // multiplied (the GEMM leaves half of the HBM bandwidth unused; in front of it the count words were 0.5 of the
            // 1.6 ms a 10M-row, 256-query batch spends before its GEMM starts, 2.7 of 6.9 ms at 12.5M rows x 1024 queries).
            int n_ranges = 1;
            int64_t range_row[17];
            range_row[0] = 0;
            for (int r = 1; r <= 16; ++r) range_row[r] = n;
            double plane_bytes_per_row = 4.0 * orr::kCountPlanes * (double)((B + 31) / 32);
            if (kw.bitmaps && !ts_gemv) {
                ORR_TRY(idx->ws_fany.reserve(sizeof(uint32_t) * orr::kCountPlanes * (size_t)((B + 31) / 32) * (size_t)epi.plane_stride));
                if (prefix_i8 && two_stage && n >= (int64_t)2000000) {
                    // four ranges, eight for more than 256 queries (the first range's count words sit in front of the GEMM: 4 planes
                    // x 4 B per row and 32 queries)
                    n_ranges = B > 256 ? 8 : 4;                     // (12.5M rows x 1024 queries: 43.0 / 41.1 / 41.0 ms per batch with 4 / 8 / 16)
                    // (boundaries on whole rounds of the persistent GEMM -- 256 workgroups x 256-row tiles: every workgroup of a
                    // launch then multiplies the same number of tiles; only the last range ends on a partial round)
                    constexpr int64_t kRound = 256 * 256;
                    for (int r = 1; r < n_ranges; ++r) range_row[r] = (n * r / n_ranges + kRound / 2) / kRound * kRound;
                }
                // two-bit count words where every query has at most three terms and the 16 x 16 x 64 form screens (its epilogue reads
                // them): half the words written here and read there
                uint32_t max_terms = 0;
                for (int32_t b = 0; b < B; ++b) max_terms = std::max(max_terms, qoff[(size_t)b + 1] - qoff[(size_t)b]);
                if (max_terms <= 3 && idx->opt_two_stage == 1 && two_stage && prefix_i8 && orr::screen_i8_uses_tile16(B, n, idx->dim, epi.plane_stride) &&
                    !getenv("ORR_COUNT_BITS4"))
                    epi.count_bits = 2;
                plane_bytes_per_row = (epi.count_bits == 2 ? 8.0 : 16.0) * (double)((B + 31) / 32);
                Timed t(idx, "count_planes", plane_bytes_per_row * (double)(range_row[1] - range_row[0]));
                HIP_TRY(orr::launch_query_count_planes(kw, B, n, epi.plane_stride, idx->ws_fany.as<uint32_t>(), s, 0, range_row[1], epi.count_bits == 2 ? 2 : 4));
                epi.count_planes = idx->ws_fany.as<uint32_t>();
            }
            epi.qf = idx->ws_fqf.as<float4>();
            epi.rowc = d_rowc; epi.qc = idx->ws_qc.as<orr::QueryConst>(); epi.kw = kw;
            epi.cnt = idx->ws_fcnt.as<uint32_t>(); epi.buf = idx->ws_fbuf.as<orr::SelEntry>(); epi.cap = kCap;
            // (the counters are cleared by the query-constants launch in front of the screening launches; the streaming forms
            // cleared them with the queries' images)
            const bool clear_with_consts = !counters_cleared && !ts_gemv;
            if (!counters_cleared && !clear_with_consts) HIP_TRY(hipMemsetAsync(idx->ws_fcnt.p, 0, sizeof(uint32_t) * 3 * (size_t)B, s));
            ORR_TRY(idx->ws_tickets.reserve(sizeof(uint32_t) * 8 * 16));
            if (two_stage) {
                // ---- two-stage: floor from the k-th best split-pass score of the prefix; ONE plain-bf16
                // product over ALL rows keeps every row that can still reach it; those are re-scored
                // exactly; the best k' of them become the records
                const int32_t kth = std::max<int32_t>(1, a.topk);
                // plain bf16: (1 + u)^2 - 1 per product with u = 2^-8, D additions charged 2^-23 each
                const bool i8_gemm_planned = idx->opt_two_stage == 1 && !ts_gemv && idx->dim % 128 == 0 &&
                                             (idx->i8_ready || (!idx->i8_failed && !idx->is_view));
                if (i8_gemm_planned) ORR_TRY(ensure_i8_shadow(idx));
                const double eps1 = (ts_i8 || (i8_gemm_planned && idx->i8_ready)) ? 0.0     // int8 forms: the per-pair bound is added inside the kernel
                                          : 0.7 * 1.01 * (0.0078125 * (1.0 + 0.001953125) + 1.02 * (double)idx->dim * 1.1920928955078125e-07) + 1e-12;
                ORR_TRY(idx->ws_tsL.reserve(sizeof(double) * (size_t)B));
                ORR_TRY(idx->ws_tskey.reserve(sizeof(unsigned long long) * (size_t)B));
                // the floor comes out of the sampling selection's own launch
                orr::FloorOut floor;
                floor.floor_key = idx->ws_tskey.as<unsigned long long>();
                floor.L = idx->ws_tsL.as<double>();
                floor.eps3 = approx_eps; floor.eps1 = eps1;
                if (ts_gemv) {
                    // the sample goes through the stream too: floor keys of 0 keep every sampled row, their
                    // approximate keys are sorted in lists of 64 and the k-th best one per query is the floor's base
                    const uint32_t cap_p = (uint32_t)dotf_rows;                 // a multiple of 4096
                    ORR_TRY(idx->ws_pbuf.reserve(sizeof(orr::SelEntry) * (size_t)B * cap_p));
                    if (idx->ws_zero.cap < sizeof(unsigned long long) * (size_t)B) {   // floor keys of 0, never written again
                        ORR_TRY(idx->ws_zero.reserve(sizeof(unsigned long long) * (size_t)std::max<int32_t>(B, 64)));
                        HIP_TRY(hipMemsetAsync(idx->ws_zero.p, 0, idx->ws_zero.cap, s));
                    }
                    orr::FusedEpilogue pre = epi;
                    pre.tau = idx->ws_zero.as<unsigned long long>();
                    pre.buf = idx->ws_pbuf.as<orr::SelEntry>();
                    pre.cap = cap_p;
                    pre.cnt = idx->ws_fcnt.as<uint32_t>() + B;         // its own counters: one clearing for both launches
                    const int64_t pre_rows = std::min<int64_t>(dotf_rows, n);
                    const bool pre_lists = ts_i8 && orr::screen_gemv_i8_prefix_makes_lists(idx->dim);   // sorted lists straight from the kernel
                    ORR_TRY(idx->ws_psel.reserve(sizeof(orr::SelEntry) * (size_t)B * cap_p));
                    if (pre_lists) pre.buf = idx->ws_psel.as<orr::SelEntry>();
                    if (!ts_i8) ORR_TRY(need_split());
                    {
                        Timed t(idx, "screen_gemv_prefix", (ts_i8 ? 1.0 : 2.0) * (double)dotf_rows * idx->dim + 2.0 * (double)B * idx->dim);
                        if (ts_i8)
                            HIP_TRY(orr::launch_screen_gemv_i8(idx->ws_q8.p, idx->ws_q8s1.as<float>(), idx->ws_q8err.as<double>(), B, idx->emb_i8.p,
                                                               idx->i8_scale.as<float>(), idx->i8_rel_err.as<float>(), idx->i8_rel_hat.as<float>(),
                                                               idx->d_norm_b, idx->d_created, a.now_ticks, pre_rows, idx->dim, pre, true, s));
                        else
                            HIP_TRY(orr::launch_screen_gemv_bf16(idx->ws_qsplit.p, B, idx->emb_shadow.p, pre_rows, idx->dim, pre, s));
                    }
                    {   // lists of 64 sorted in parallel (by the int8 stream itself where it can), then the k-th best key per query
                        Timed t(idx, "select_floor", 0.0);
                        const int32_t lists_all = (int32_t)(cap_p / orr::kSelWidth);
                        const int32_t lists_p = pre_lists ? (int32_t)((pre_rows + orr::kSelWidth - 1) / orr::kSelWidth) : lists_all;
                        if (!pre_lists)
                            HIP_TRY(orr::launch_buffer_to_lists(pre.buf, pre.cnt, cap_p, B, 0, lists_all, idx->ws_psel.as<orr::SelEntry>(), s));
                        HIP_TRY(orr::launch_select_final_sample(idx->ws_psel.as<orr::SelEntry>(), lists_all, lists_p, B, kth, d_tau, s, floor));
                    }
                } else {
                    Timed t(idx, "select_floor", 0.0);
                    HIP_TRY(orr::launch_select_final_sample(idx->ws_sel.as<orr::SelEntry>(), lists_total, fused_sample_seg, B, kth, d_tau, s, floor,
                                                            prefix_floor_form == 2 ? 1 : 0));
                }
                // the screening GEMM runs on the int8 shadow where there is one (K2j), else on the bf16 shadow (K2c),
                // else it converts the fp32 rows itself
                bool gemm_i8 = false, tickets_cleared = false;
                if (idx->opt_two_stage == 1 && !ts_gemv) {
                    ORR_TRY(ensure_i8_shadow(idx));
                    gemm_i8 = idx->i8_ready;
                    if (!gemm_i8) ORR_TRY(ensure_shadow(idx));
                }
                if (gemm_i8) {
                    if (!prefix_i8) {              // (the int8 prefix quantised and tiled the queries already)
                        ORR_TRY(idx->ws_q8.reserve(2 * (size_t)B * idx->dim));
                        ORR_TRY(idx->ws_q8s1.reserve(sizeof(float) * (size_t)B));
                        ORR_TRY(idx->ws_q8err.reserve(2 * sizeof(double) * (size_t)B));
                        HIP_TRY(orr::launch_i8_queries(d_q, B, idx->dim, idx->ws_q8.p, idx->ws_q8s1.as<float>(), idx->ws_q8err.as<double>(), s,
                                                       idx->ws_q8err.as<double>() + B));
                    }
                    epi.i8_rowf = idx->i8_rowf.as<float4>();
                    epi.i8_qs1 = idx->ws_q8s1.as<float>();
                }
                if (!ts_gemv) {                  // the streaming kernels score in fp64 directly, no fp32 pre-filter constants
                    // (ws_fqf holds two arrays of B: qf, and behind it the NaN-safe copy the 16 x 16 x 64 form stages)
                    HIP_TRY(orr::launch_fused_query_consts(idx->ws_qc.as<orr::QueryConst>(), idx->ws_tskey.as<unsigned long long>(), B,
                                                           idx->ws_fqf.as<float4>(), s, gemm_i8 ? idx->ws_q8s1.as<float>() : nullptr,
                                                           gemm_i8 ? idx->ws_q8err.as<double>() + B : nullptr,
                                                           gemm_i8 ? idx->ws_fqf.as<float4>() + B : nullptr,
                                                           clear_with_consts ? idx->ws_fcnt.as<uint32_t>() : nullptr, 3 * B,
                                                           idx->ws_tickets.as<uint32_t>(), 8 * 16));
                    tickets_cleared = true;
                    epi.qf16 = gemm_i8 ? idx->ws_fqf.as<float4>() + B : nullptr;
                }
                epi.tau = idx->ws_tskey.as<unsigned long long>();
                if (n_ranges > 1 && !gemm_i8) {         // (ranges were planned for the int8 GEMM: the other forms take one launch)
                    Timed t(idx, "count_planes", plane_bytes_per_row * (double)(n - range_row[1]));
                    HIP_TRY(orr::launch_query_count_planes(kw, B, n, epi.plane_stride, idx->ws_fany.as<uint32_t>(), s, range_row[1], n, epi.count_bits == 2 ? 2 : 4));
                    n_ranges = 1;
                }
                if (gemm_i8) {
                    if (!prefix_i8) {
                        ORR_TRY(idx->ws_qtiled.reserve(orr::i8_tiled_bytes(B, idx->dim)));
                        HIP_TRY(orr::launch_i8_tile_queries(idx->ws_q8.p, B, idx->dim, idx->ws_qtiled.p, s));
                    }
                    ORR_TRY(screen_i8_in_ranges(idx, a, n, kw, epi, n_ranges, range_row, plane_bytes_per_row, s, tickets_cleared));
                } else if (ts_i8) {
                    Timed t(idx, "screen_gemv_i8", 1.0 * (double)n * idx->dim + 28.0 * (double)n + 2.0 * (double)B * idx->dim);   // per row: scale, two relative norms (12 B), normB and created (16 B)
                    HIP_TRY(orr::launch_screen_gemv_i8(idx->ws_q8.p, idx->ws_q8s1.as<float>(), idx->ws_q8err.as<double>(), B, idx->emb_i8.p,
                                                       idx->i8_scale.as<float>(), idx->i8_rel_err.as<float>(), idx->i8_rel_hat.as<float>(),
                                                       idx->d_norm_b, idx->d_created, a.now_ticks, n, idx->dim, epi, false, s));
                } else if (ts_gemv) {
                    ORR_TRY(need_split());
                    Timed t(idx, "screen_gemv_bf16", 2.0 * (double)n * idx->dim + 2.0 * (double)B * idx->dim);
                    HIP_TRY(orr::launch_screen_gemv_bf16(idx->ws_qsplit.p, B, idx->emb_shadow.p, n, idx->dim, epi, s));
                } else if (idx->opt_two_stage == 1 && idx->shadow_ready) {
                    ORR_TRY(idx->ws_qtiled.reserve(orr::bf16_tiled_bytes(B, idx->dim)));
                    HIP_TRY(orr::launch_bf16_tiled(d_q, B, idx->dim, idx->ws_qtiled.p, s));
                    Timed t(idx, "screen_bf16_fused", 2.0 * (double)n * idx->dim + 2.0 * (double)B * idx->dim);
                    HIP_TRY(orr::launch_screen_bf16(idx->ws_qtiled.p, B, idx->emb_shadow.p, 0, n, idx->dim, nullptr, 0, &epi, s));
                } else {
                    ORR_TRY(need_split());
                    Timed t(idx, "gemm_dot_bf16x1_fused", 4.0 * (double)n * idx->dim + 2.0 * (double)B * idx->dim);
                    HIP_TRY(orr::launch_gemm_dot_bf16x3(idx->ws_qsplit.p, B, idx->d_emb, 0, n, idx->dim, nullptr, 0, &epi, 1, s));
                }
                ORR_TRY(two_stage_tail(idx, a, kprime, n, d_q, kw, epi, kCap, buf_lists, host_records, rec_bytes, &d_cand, &direct_host, s));
                records_have_dots = true;
                a.used_two_stage = true;
                pass_mode = (gemm_i8 || ts_i8) ? 1 : (((ts_gemv && !ts_i8) || (idx->opt_two_stage == 1 && idx->shadow_ready)) ? 2 : 3);
            } else {
            {
                Timed t(idx, "select_floor", 0.0);
                HIP_TRY(orr::launch_select_final_sample(idx->ws_sel.as<orr::SelEntry>(), lists_total, fused_sample_seg, B, kprime, d_tau, s));
            }
            HIP_TRY(orr::launch_fused_query_consts(idx->ws_qc.as<orr::QueryConst>(), d_tau, B, idx->ws_fqf.as<float4>(), s, nullptr, nullptr, nullptr,
                                                   clear_with_consts ? idx->ws_fcnt.as<uint32_t>() : nullptr, 3 * B));
            epi.tau = d_tau;
            {
                ORR_TRY(need_split());
                Timed t(idx, "gemm_dot_bf16x3_fused", 4.0 * (double)(n - dotf_rows) * idx->dim + 4.0 * (double)B * idx->dim);
                HIP_TRY(orr::launch_gemm_dot_bf16x3(idx->ws_qsplit.p, B, idx->d_emb, dotf_rows, n, idx->dim, nullptr, 0, &epi, 3, s));
            }
            {
                Timed t(idx, "buffer_to_lists", 0.0);
                HIP_TRY(orr::launch_buffer_to_lists(epi.buf, epi.cnt, kCap, B, fused_sample_seg, lists_total,
                                                    idx->ws_sel.as<orr::SelEntry>(), s));
            }
            {
                Timed t(idx, "select_final", (double)B * (double)lists_total * orr::kSelWidth * sizeof(orr::SelEntry));
                HIP_TRY(orr::launch_select_final(idx->ws_sel.as<orr::SelEntry>(), lists_total, B, kprime, n, idx->row_base,
                                                 nullptr, nullptr, 0, idx->d_norm_b, idx->d_created, idx->d_row_ids, kw,
                                                 0, approx_eps, nullptr, epi.cnt, kCap, nullptr, d_cand, s));
            }
            }
        } else {
        // Large batches: scan a prefix first, take its k'-th best key per query as a floor, and let
        // the rest of the corpus skip every 64-row batch that cannot beat it.
        const int32_t sample_seg = (B >= 8 && n_seg32 >= 48) ? std::min<int32_t>(64, std::max<int32_t>(16, n_seg32 / 16)) : 0;
        if (sample_seg > 0) {
            ORR_TRY(idx->ws_tau.reserve(sizeof(unsigned long long) * (size_t)B));
            d_tau = idx->ws_tau.as<unsigned long long>();
            {
                Timed t(idx, "fuse_select", (double)B * (double)sample_seg * orr::kSelSegRows * 28.0);
                HIP_TRY(orr::launch_fuse_select(d_dot, d_dotf, n, idx->d_norm_b, idx->d_created, d_rowc, kw,
                                                idx->ws_qc.as<orr::QueryConst>(), a.now_ticks, n, B, 0, sample_seg, nullptr,
                                                idx->ws_sel.as<orr::SelEntry>(), 0, s));
            }
            {
                Timed t(idx, "select_floor", 0.0);
                HIP_TRY(orr::launch_select_final_sample(idx->ws_sel.as<orr::SelEntry>(), n_seg32, sample_seg, B, kprime, d_tau, s));
            }
        }
        {
            Timed t(idx, "fuse_select", (double)B * (double)n * (8.0 * (use_cos ? 1 : 0) + 8.0 + 8.0));
            HIP_TRY(orr::launch_fuse_select(d_dot, d_dotf, n, idx->d_norm_b, idx->d_created, d_rowc, kw,
                                            idx->ws_qc.as<orr::QueryConst>(), a.now_ticks, n, B, sample_seg, n_seg32 - sample_seg,
                                            d_tau, idx->ws_sel.as<orr::SelEntry>(), 0, s));
        }
        {
            Timed t(idx, "select_final", (double)B * (double)n_seg * orr::kSelWidth * sizeof(orr::SelEntry));
            HIP_TRY(orr::launch_select_final(idx->ws_sel.as<orr::SelEntry>(), (int32_t)n_seg, B, kprime, n, idx->row_base,
                                             d_dot, d_dotf, n, idx->d_norm_b, idx->d_created, idx->d_row_ids, kw,
                                             use_mfma ? 0 : 1, approx_eps, nullptr, nullptr, 0u, nullptr, d_cand, s));
        }
        }
        if (approx_pass && !records_have_dots) {   // K6: the survivors' dots again, now in the reference's own arithmetic
            Timed t(idx, "rescore_exact", (double)B * kprime * 4.0 * idx->dim);
            HIP_TRY(orr::launch_rescore_exact(idx->d_emb, idx->dim, d_q, B, kprime, idx->row_base, d_cand, s));
        }
    } else {
        ORR_TRY(run_large_k(idx, a, kprime, n, d_dot, kw, qc, d_cand, s));
    }
    if (!owner_of(idx)->dead.empty())      // records of deleted rows are dropped by the host finish
        HIP_TRY(orr::launch_mark_dead_records(d_cand, B, kprime, owner_of(idx)->d_dead.as<int64_t>(),
                                              (int32_t)owner_of(idx)->dead.size(), idx->row_base, s));
    if (host_records && !direct_host) {     // large record sets: one asynchronous copy into pinned memory behind the last kernel
        ORR_TRY(idx->pin_cand.reserve(rec_bytes));
        HIP_TRY(hipMemcpyAsync(idx->pin_cand.p, d_cand, rec_bytes, hipMemcpyDeviceToHost, s));
    }
    g_ht.mark(3);
    idx->sstats.pass_mode = pass_mode;
    HIP_TRY(hipStreamSynchronize(s));
    if (bm_bytes) {            // every kernel that read the bitmaps is done: clear them for the next search
        if (bm_bytes >= ((size_t)16 << 20)) {   // (small ones stay on the keyword stream: a cross-stream wait costs a one-query call more)
            HIP_TRY(hipMemsetAsync(idx->ws_bitmaps.p, 0, bm_bytes, idx->stream_aux));
            HIP_TRY(hipEventRecord(idx->ev_bm_clean, idx->stream_aux));
            idx->bm_clean_pending = true;
        } else {
            HIP_TRY(hipMemsetAsync(idx->ws_bitmaps.p, 0, bm_bytes, idx->stream_kw));
        }
        idx->bitmaps_clean = std::max(bm_bytes, bm_clean_before);
    }
    if (n_terms_total > 0 && idx->ws_counter.p) {   // ... and the keyword chain's counters (two memsets less in front of the next chain)
        HIP_TRY(hipMemsetAsync(idx->ws_counter.p, 0, sizeof(unsigned long long), idx->stream_kw));
        if (idx->ws_kwalias.p) HIP_TRY(hipMemsetAsync(idx->ws_kwalias.p, 0, idx->ws_kwalias.cap, idx->stream_kw));
        idx->kw_counters_clean = true;
        idx->kw_counters_of[0] = idx->ws_counter.p;
        idx->kw_counters_of[1] = idx->ws_kwalias.p;
    }
    if (dev_norms) memcpy(idx->h_norm_a.data(), idx->pin_norm.p, sizeof(double) * (size_t)B);
    if (n_terms_total > 0) {
        idx->sstats.kw_hits_total += (int64_t)(*idx->pin_kwcnt.as<unsigned long long>() >> 32);
        idx->sstats.kw_passes += 1;
    }
    if (a.used_two_stage) {
        idx->h_survivors.assign(idx->pin_cnt.as<uint32_t>(), idx->pin_cnt.as<uint32_t>() + B);
        uint64_t sum = 0;
        for (uint32_t cnt : idx->h_survivors) sum += cnt;
        const uint64_t mean = sum / (uint64_t)B;
        if (mean > 4096 && idx->sample_boost < 16) idx->sample_boost *= 2;
        else if (mean < 512 && idx->sample_boost > 1) idx->sample_boost /= 2;
    }
    g_ht.mark(4);
    collect_events(idx);
    if (kw_overflow_possible) {
        // (the counter as expand_hits left it in pinned memory: the device copy is zeroed again behind the pass)
        const unsigned long long cnt = *idx->pin_kwcnt.as<unsigned long long>();
        const uint32_t hits = (uint32_t)(cnt >> 32);
        if (hits > kw_max_hits) {
            // the distinct terms of this batch match more vocabulary tokens than the hit list holds (short terms against a
            // large vocabulary): the bitmaps are incomplete, so this pass's records are discarded; the list grows to the
            // measured count and the pass runs again (the index keeps the larger list)
            if ((uint64_t)hits * sizeof(orr::KwHit) > ((uint64_t)8 << 30))
                return fail(ORR_ENOMEM, "keyword terms matched %u vocabulary tokens: a hit list of that size is refused (8 GiB)", hits);
            idx->kw_hits_cap = hits + hits / 4 + 1024u;
            return kRetryPass;
        }
    }
    if (recs_host && direct_host) *recs_host = d_cand;
    else if (recs_host && host_records) *recs_host = idx->pin_cand.as<orr_candidate>();
    return ORR_OK;
}

int run_shard(orr_index *idx, const BatchArgs &a, int32_t kprime, bool host_records, const float **q_host,
              const orr_candidate **recs_host)
{
    for (int attempt = 0;; ++attempt) {
        const int r = run_shard_once(idx, a, kprime, host_records, q_host, recs_host);
        if (r != kRetryPass) return r;
        idx->sstats.passes += 1;
        if (attempt >= 3) return fail(ORR_EDEVICE, "the keyword hit list kept overflowing");
    }
}

// Host finish for one query over records from any number of shards.
// Returns the number of results; *certified tells whether rows outside the
// candidate sets were provably unable to reach the top-k.
int32_t finish_query(const orr_candidate *const *shard_recs, int32_t n_shards, int32_t kprime, bool use_cos,
                     double norm_a, int32_t n_terms, int64_t now_ticks, int32_t topk, int64_t *out_rows,
                     double *out_scores, bool *certified, int *err)
{
    std::vector<Ranked> ranked;
    double cutoff = -std::numeric_limits<double>::infinity();
    double eps = kCertifyEps;
    bool any_cut = false, overflow = false;
    double need_score = -std::numeric_limits<double>::infinity();   // two-stage: rows never offered score below this
    *err = ORR_OK;
    for (int32_t sidx = 0; sidx < n_shards; ++sidx) {
        const orr_candidate *rec = shard_recs[sidx];
        const orr_candidate &tr = rec[kprime];
        if (!(tr.flags & ORR_CAND_TRAILER) || tr.matches < 0 || tr.matches > kprime) {
            *err = fail(ORR_ECOMM, "orr_merge_candidates: shard %d has a malformed trailer", sidx);
            return 0;
        }
        for (int32_t i = 0; i < tr.matches; ++i) {
            const orr_candidate &c = rec[i];
            if (c.row_id < 0 && c.order_key < 0) continue;
            if (c.flags & ORR_CAND_DEAD) continue;                    // deleted row (orr_index_delete_rows)
            Ranked r;
            r.score = exact_score(c, use_cos, norm_a, n_terms, now_ticks);
            r.order_key = c.order_key;
            r.row_id = c.row_id;
            ranked.push_back(r);
        }
        if (tr.dot > eps) eps = tr.dot;            // bound of the pass that produced this shard's records
        if (tr.flags & ORR_CAND_OVERFLOW) overflow = true;
        if ((tr.flags & ORR_CAND_TWO_STAGE) && tr.norm_b > need_score) need_score = tr.norm_b;
        if (tr.approx_score != -std::numeric_limits<double>::infinity()) {
            any_cut = true;
            // NaN cut-off: everything left out is NaN too (NaN sorts last), harmless
            if (!(tr.approx_score != tr.approx_score) && tr.approx_score > cutoff) cutoff = tr.approx_score;
        }
    }
    std::sort(ranked.begin(), ranked.end(), [](const Ranked &x, const Ranked &y) {
        const int c = compare_double(x.score, y.score);
        if (c != 0) return c > 0;                 // OrderByDescending(score)       :34
        return x.order_key < y.order_key;         // ThenByDescending(created), stable == candidate order  :35
    });
    const int32_t take = std::max<int32_t>(1, topk);                                 // :36
    const int32_t n_out = (int32_t)std::min<size_t>((size_t)take, ranked.size());
    for (int32_t i = 0; i < n_out; ++i) {
        out_rows[i] = ranked[i].row_id;
        out_scores[i] = ranked[i].score;
    }
    const bool lower_bounded = need_score != -std::numeric_limits<double>::infinity();
    if (overflow) {
        *certified = false;                       // some survivors were dropped: repeat unfused
    } else if (!any_cut && !lower_bounded) {
        *certified = true;
    } else if (n_out < take) {
        *certified = false;                       // fewer results than asked while rows were cut
    } else {
        const double sk = ranked[n_out - 1].score;
        *certified = !any_cut || sk > cutoff + eps;           // false for NaN
        // two-stage: rows that were never offered score below need_score, and the construction makes
        // S_k >= need_score; a violated bound must not pass silently
        if (*certified && lower_bounded) *certified = sk >= need_score - 1e-12;
    }
    return n_out;
}

int merge_impl(int32_t n_shards, int32_t B, int32_t kprime, const orr_candidate *all, int32_t dim, bool use_cos,
               const float *q_host, const double *norms, const uint32_t *query_term_off, int64_t now_ticks, int32_t topk,
               int64_t *out_rows, double *out_scores, int32_t *out_counts, int32_t *out_uncertified,
               uint8_t *out_certified = nullptr)
{
    const int32_t take = std::max<int32_t>(1, topk);
    int32_t unc = 0;
    std::vector<double> own_norms;
    if (use_cos && !norms) {
        own_norms.resize((size_t)B);
        exact_norms(q_host, B, dim, own_norms.data());
        norms = own_norms.data();
    }
    // queries are independent: large merges (many queries x many shards) are split over a few host threads
    auto work = [&](int32_t b_begin, int32_t b_end, int32_t *unc_out, int *err_out) {
        std::vector<const orr_candidate *> recs((size_t)n_shards);
        for (int32_t b = b_begin; b < b_end; ++b) {
            for (int32_t sidx = 0; sidx < n_shards; ++sidx)
                recs[sidx] = all + ((size_t)sidx * B + b) * ((size_t)kprime + 1);
            const double norm_a = use_cos ? norms[b] : 0.0;
            const int32_t n_terms = (int32_t)(query_term_off[b + 1] - query_term_off[b]);
            bool cert = false;
            int err = ORR_OK;
            for (int32_t i = 0; i < take; ++i) { out_rows[(size_t)b * take + i] = -1; out_scores[(size_t)b * take + i] = 0.0; }
            const int32_t cnt = finish_query(recs.data(), n_shards, kprime, use_cos, norm_a, n_terms, now_ticks, topk,
                                             out_rows + (size_t)b * take, out_scores + (size_t)b * take, &cert, &err);
            if (err != ORR_OK) { *err_out = err; return; }
            if (out_counts) out_counts[b] = cnt;
            if (out_certified) out_certified[b] = cert ? 1 : 0;
            if (!cert) ++*unc_out;
        }
    };
    const int64_t n_records = (int64_t)B * n_shards * kprime;
    int n_thr = n_records >= 4096 ? std::min(HostPool::get().width() * 2, (B + 15) / 16) : 1;     // tasks of >= 16 queries
    if (n_thr < 1) n_thr = 1;
    std::vector<int32_t> t_unc((size_t)n_thr, 0);
    std::vector<int> t_err((size_t)n_thr, ORR_OK);
    if (n_thr == 1) {
        work(0, B, &t_unc[0], &t_err[0]);
    } else {
        HostPool::get().run(n_thr, [&](int t) {
            work((int32_t)((int64_t)B * t / n_thr), (int32_t)((int64_t)B * (t + 1) / n_thr), &t_unc[(size_t)t], &t_err[(size_t)t]);
        });
    }
    for (int t = 0; t < n_thr; ++t) {
        if (t_err[(size_t)t] != ORR_OK)
            return n_thr == 1 ? t_err[0] : fail(t_err[(size_t)t], "orr_merge_candidates: a shard's records are malformed");   // the detail was set on a pool thread
        unc += t_unc[(size_t)t];
    }
    if (out_uncertified) *out_uncertified = unc;
    return ORR_OK;
}


// ---- one batch through the passes, escalating ONLY the queries that could not be certified ----------------------------
// A query whose top-k could not be certified (a tie at the cut, a survivors' buffer that overflowed, k' too small for a
// mass of equal scores) goes through the next more exact pass as part of a compacted sub-batch; the others keep their
// results.  Order of escalation: larger survivors' buffers (when that was the only problem and they stay affordable) ->
// unfused batched pass -> the reference-arithmetic pass over all rows -> k' x 4.  Passes whose workspace grows with
// (queries x rows) are run over slices of the sub-batch, so the workspace stays bounded (kPassWorkspaceBytes).
constexpr size_t kPassWorkspaceBytes = (size_t)4 << 30;

struct SubBatch {                  // storage of a compacted sub-batch (the vectors live in idx->ws_qsub when they are device-resident)
    std::vector<float> q_host;
    std::vector<uint8_t> pool;
    std::vector<uint32_t> term_off, qoff;
};

int build_subset(orr_index *idx, const BatchArgs &orig, const std::vector<int32_t> &ids, SubBatch &sb, BatchArgs &out)
{
    out = orig;
    const int32_t nb = (int32_t)ids.size();
    out.B = nb;
    out.used_mfma = out.used_fused = out.used_two_stage = false;
    if (orig.dim > 0) {
        const size_t row = (size_t)orig.dim;
        if (is_device_pointer(orig.q)) {
            ORR_TRY(idx->ws_qsub.reserve(sizeof(float) * row * (size_t)nb));
            for (int32_t i = 0; i < nb; ++i)
                HIP_TRY(hipMemcpyAsync(idx->ws_qsub.as<float>() + (size_t)i * row, orig.q + (size_t)ids[(size_t)i] * row, sizeof(float) * row,
                                       hipMemcpyDeviceToDevice, idx->stream));
            HIP_TRY(hipStreamSynchronize(idx->stream));
            out.q = idx->ws_qsub.as<float>();
        } else {
            sb.q_host.resize(row * (size_t)nb);
            for (int32_t i = 0; i < nb; ++i)
                memcpy(sb.q_host.data() + (size_t)i * row, orig.q + (size_t)ids[(size_t)i] * row, sizeof(float) * row);
            out.q = sb.q_host.data();
        }
    }
    sb.pool.clear(); sb.term_off.assign(1, 0u); sb.qoff.assign(1, 0u);
    for (int32_t i = 0; i < nb; ++i) {
        const int32_t b = ids[(size_t)i];
        for (uint32_t t = orig.query_term_off[b]; t < orig.query_term_off[b + 1]; ++t) {
            const uint32_t o = orig.term_off[t], e = orig.term_off[t + 1];
            if (e < o) return fail(ORR_EINVAL, "term_off is not monotone at term %u", t);
            sb.pool.insert(sb.pool.end(), orig.terms_utf8 + o, orig.terms_utf8 + e);
            sb.term_off.push_back((uint32_t)sb.pool.size());
        }
        sb.qoff.push_back((uint32_t)sb.term_off.size() - 1u);
    }
    sb.pool.push_back(0);
    out.terms_utf8 = sb.pool.data();
    out.term_off = sb.term_off.data();
    out.query_term_off = sb.qoff.data();
    return ORR_OK;
}

// ids: queries of `orig` to answer (ascending); whole = ids is the entire batch in order.  flags (no_fuse / force_exact) are
// carried in `orig` for sub-batches.  Results are written to out_*[ids[i]].
int search_ids(orr_index *idx, const BatchArgs &orig, const std::vector<int32_t> &ids, bool whole, int64_t kprime, int64_t n,
               int64_t *out_rows, double *out_scores, int32_t *out_counts, int depth)
{
    const int32_t nb = (int32_t)ids.size();
    const int32_t take = std::max<int32_t>(1, orig.topk);
    const bool use_cos = orig.dim > 0 && orig.dim == idx->dim;
    // passes that keep a number per (query,row): slices of the sub-batch
    if ((orig.no_fuse || orig.force_exact) && nb > 1 && (size_t)nb * (size_t)std::max<int64_t>(n, 1) * 8 > kPassWorkspaceBytes) {
        const int32_t per = (int32_t)std::max<size_t>(1, kPassWorkspaceBytes / ((size_t)std::max<int64_t>(n, 1) * 8));
        for (int32_t i0 = 0; i0 < nb; i0 += per) {
            std::vector<int32_t> part(ids.begin() + i0, ids.begin() + std::min<int32_t>(nb, i0 + per));
            ORR_TRY(search_ids(idx, orig, part, false, kprime, n, out_rows, out_scores, out_counts, depth));
        }
        return ORR_OK;
    }
    SubBatch sb;
    BatchArgs cur = orig;
    if (!whole) ORR_TRY(build_subset(idx, orig, ids, sb, cur));
    cur.no_fuse = orig.no_fuse; cur.force_exact = orig.force_exact;

    const float *q_host = nullptr;
    const orr_candidate *recs = nullptr;
    ORR_TRY(run_shard(idx, cur, (int32_t)kprime, true, &q_host, &recs));
    idx->sstats.passes += 1;
    if (depth > 0) idx->sstats.requeried += nb;
    std::vector<orr_candidate> copied;
    if (!recs) {                                  // large record sets stay on the device until here
        copied.resize((size_t)nb * ((size_t)kprime + 1));
        HIP_TRY(hipMemcpy(copied.data(), idx->ws_cand.p, sizeof(orr_candidate) * copied.size(), hipMemcpyDeviceToHost));
        recs = copied.data();
    }
    std::vector<uint8_t> cert((size_t)nb, 1);
    int32_t unc = 0;
    if (whole) {
        ORR_TRY(merge_impl(1, nb, (int32_t)kprime, recs, cur.dim, use_cos, q_host, use_cos ? idx->h_norm_a.data() : nullptr,
                           cur.query_term_off, cur.now_ticks, cur.topk, out_rows, out_scores, out_counts, &unc, cert.data()));
    } else {
        std::vector<int64_t> rows((size_t)nb * take);
        std::vector<double> scores((size_t)nb * take);
        std::vector<int32_t> counts((size_t)nb);
        ORR_TRY(merge_impl(1, nb, (int32_t)kprime, recs, cur.dim, use_cos, q_host, use_cos ? idx->h_norm_a.data() : nullptr,
                           cur.query_term_off, cur.now_ticks, cur.topk, rows.data(), scores.data(), counts.data(), &unc, cert.data()));
        for (int32_t i = 0; i < nb; ++i) {
            const size_t b = (size_t)ids[(size_t)i];
            memcpy(out_rows + b * take, rows.data() + (size_t)i * take, sizeof(int64_t) * take);
            memcpy(out_scores + b * take, scores.data() + (size_t)i * take, sizeof(double) * take);
            if (out_counts) out_counts[b] = counts[(size_t)i];
        }
    }
    g_ht.mark(5);
    // survivors of the screening pass (two-stage): statistics, and the buffer size the next pass needs
    uint32_t worst_unc_survivors = 0;
    bool unc_only_overflow = unc > 0;
    if (cur.used_two_stage && (int32_t)idx->h_survivors.size() == nb) {
        for (int32_t i = 0; i < nb; ++i) {
            const uint32_t c = idx->h_survivors[(size_t)i];
            idx->sstats.survivors_total += c;
            idx->sstats.survivor_samples += 1;
            if ((int64_t)c > idx->sstats.survivors_max) idx->sstats.survivors_max = c;
            if (c > idx->pass_cap) idx->sstats.overflowed_queries += 1;
            if (!cert[(size_t)i]) {
                if (c > idx->pass_cap) worst_unc_survivors = std::max(worst_unc_survivors, c);
                else unc_only_overflow = false;
            }
        }
    } else {
        unc_only_overflow = false;
    }
    idx->sstats.survivor_capacity = idx->survivor_cap;
    if (unc == 0) return ORR_OK;

    std::vector<int32_t> again;
    for (int32_t i = 0; i < nb; ++i) if (!cert[(size_t)i]) again.push_back(ids[(size_t)i]);
    BatchArgs next = orig;
    next.no_fuse = cur.no_fuse; next.force_exact = cur.force_exact;
    if (unc_only_overflow && worst_unc_survivors < (1u << 19) && (int64_t)worst_unc_survivors * 2 < n &&
        (size_t)again.size() * (size_t)worst_unc_survivors * 96 < ((size_t)2 << 30)) {
        // the screen kept more pairs than the buffers hold (rows clustered around the query): the same pass again for
        // these queries with buffers sized from the measured counts; the index keeps the larger size for later searches
        uint32_t cap = idx->pass_cap;
        while (cap < worst_unc_survivors + worst_unc_survivors / 8) cap *= 2;
        if (cap > idx->survivor_cap) idx->survivor_cap = cap;
        idx->sstats.buffer_growths += 1;
        if (!idx->is_view || idx->internal_lane) {      // the other lanes of the handle start from the measured size as well
            orr_index *own = const_cast<orr_index *>(owner_of(idx));
            std::lock_guard<std::mutex> ll(own->lanes_mu);
            own->survivor_cap_hint = std::max(own->survivor_cap_hint, cap);
        }
    } else if (cur.used_fused && !cur.no_fuse) {
        next.no_fuse = true;                                   // a tie at the cut or an overflow too large to buffer: unfused pass
    } else if (cur.used_mfma) {
        next.force_exact = true;                               // then the exact pass, same k'
        idx->sstats.exact_pass_queries += (int64_t)again.size();
    } else if (kprime >= n) {
        return ORR_OK;                                         // every participating row was a candidate: nothing more exact exists
    } else {
        kprime = std::min<int64_t>(n, kprime * 4);
    }
    if (depth > 40) return fail(ORR_EDEVICE, "orr_search_batch: escalation did not terminate");
    return search_ids(idx, next, again, false, kprime, n, out_rows, out_scores, out_counts, depth + 1);
}

}  // namespace

extern "C" {

int orr_index_search_stats(orr_index *idx, orr_search_stats *out, int32_t reset)
{
    if (!idx) return fail(ORR_EINVAL, "orr_index_search_stats: null index");
    AllLanes all(idx);
    std::lock_guard<std::mutex> lock(idx->mu);
    idx->sstats.survivor_capacity = idx->survivor_cap;
    idx->sstats.vocab_tokens = idx->n_tokens;
    if (out) {
        *out = idx->sstats;
        for (orr_index *l : idx->lanes) {               // the counters of every lane of this handle
            if (!l) continue;
            const orr_search_stats &t = l->sstats;
            out->searches += t.searches; out->queries += t.queries; out->passes += t.passes; out->requeried += t.requeried;
            out->overflowed_queries += t.overflowed_queries; out->buffer_growths += t.buffer_growths;
            out->exact_pass_queries += t.exact_pass_queries; out->survivors_total += t.survivors_total;
            out->survivor_samples += t.survivor_samples; out->survivors_max = std::max(out->survivors_max, t.survivors_max);
            out->survivor_capacity = std::max<int64_t>(out->survivor_capacity, l->survivor_cap);
            out->kw_hits_total += t.kw_hits_total; out->kw_passes += t.kw_passes;
            if (out->pass_mode == 0) out->pass_mode = t.pass_mode;
        }
    }
    if (reset) {
        auto clear = [](orr_index *x) { const int64_t cap = x->survivor_cap; x->sstats = orr_search_stats{}; x->sstats.survivor_capacity = cap; };
        clear(idx);
        for (orr_index *l : idx->lanes) if (l) clear(l);
    }
    return ORR_OK;
}

// One shard pass of `a` (all of its queries) with the records written to out[first .. first + a.B) (host or device memory), then --
// inside the call -- the queries whose survivors' buffers overflowed again with buffers sized from the measured counts
// (clustered rows, cosine-only scores): a caller that only saw ORR_CAND_OVERFLOW could but repeat the whole batch through
// the exact pass on every shard.  Caller holds idx->mu.
static int shard_pass_into(orr_index *idx, BatchArgs a, int32_t kprime, int64_t candidate_limit, orr_candidate *out, size_t first,
                           bool dev_out)
{
    const int32_t B = a.B;
    const size_t rec_q = sizeof(orr_candidate) * ((size_t)kprime + 1);
    unsigned char *dst = reinterpret_cast<unsigned char *>(out) + rec_q * first;
    a.out_dev = dev_out ? reinterpret_cast<orr_candidate *>(dst) : nullptr;   // records written where the caller wants them (the all-gather's send buffer)
    ORR_TRY(run_shard(idx, a, kprime, false, nullptr, nullptr));
    if (!dev_out) HIP_TRY(hipMemcpy(dst, idx->ws_cand.p, rec_q * (size_t)B, hipMemcpyDefault));
    idx->sstats.passes += 1;
    const int64_t n = participating_rows(idx, candidate_limit);
    std::vector<int32_t> active((size_t)B);             // the queries the last pass answered, in the batch's numbering
    std::iota(active.begin(), active.end(), 0);
    bool two_stage = a.used_two_stage;
    for (int round = 0; round < 4 && two_stage && idx->h_survivors.size() == active.size(); ++round) {
        std::vector<int32_t> over;
        uint32_t worst = 0;
        for (size_t i = 0; i < active.size(); ++i) {       // (statistics: only the queries this pass ran)
            const uint32_t cnt = idx->h_survivors[i];
            idx->sstats.survivors_total += cnt; idx->sstats.survivor_samples += 1;
            if ((int64_t)cnt > idx->sstats.survivors_max) idx->sstats.survivors_max = cnt;
            if (cnt > idx->pass_cap) { over.push_back(active[i]); worst = std::max(worst, cnt); }
        }
        if (over.empty()) break;
        idx->sstats.overflowed_queries += (int64_t)over.size();
        if (worst >= (1u << 19) || (int64_t)worst * 2 >= n || over.size() * (size_t)worst * 96 >= ((size_t)2 << 30)) break;   // the caller's escalation
        uint32_t cap = idx->pass_cap;
        while (cap < worst + worst / 8) cap *= 2;
        if (cap > idx->survivor_cap) idx->survivor_cap = cap;
        idx->sstats.buffer_growths += 1;
        SubBatch sb;
        BatchArgs sub;
        a.out_dev = nullptr;
        ORR_TRY(build_subset(idx, a, over, sb, sub));
        sub.no_fuse = a.no_fuse; sub.force_exact = a.force_exact;
        ORR_TRY(run_shard(idx, sub, kprime, false, nullptr, nullptr));          // records in idx->ws_cand
        idx->sstats.passes += 1; idx->sstats.requeried += (int64_t)over.size();
        for (size_t i = 0; i < over.size(); ++i)
            HIP_TRY(hipMemcpy(dst + rec_q * (size_t)over[i], static_cast<const unsigned char *>(idx->ws_cand.p) + rec_q * i, rec_q, hipMemcpyDefault));
        two_stage = sub.used_two_stage;
        active.swap(over);
    }
    idx->sstats.survivor_capacity = idx->survivor_cap;
    return ORR_OK;
}

int orr_search_shard_ex(orr_index *idx, int32_t B, int32_t dim, const float *q, const uint8_t *terms_utf8,
                        const uint32_t *term_off, const uint32_t *query_term_off, int64_t now_ticks, int32_t kprime,
                        int64_t candidate_limit, int32_t topk, int32_t pass, orr_candidate *out)
{
    BatchArgs a{B, dim, q, terms_utf8, term_off, query_term_off, now_ticks, candidate_limit, kprime};
    ORR_TRY(check_batch(idx, a, "orr_search_shard"));
    if (kprime < 1) return fail(ORR_EINVAL, "orr_search_shard: kprime must be >= 1");
    if (!out) return fail(ORR_EINVAL, "orr_search_shard: out is NULL");
    if (pass < 0 || pass > 2 || topk < 0) return fail(ORR_EINVAL, "orr_search_shard_ex: pass takes 0, 1 or 2 and topk must be >= 0");
    Lane ln;                                           // concurrent calls on one handle run on different lanes
    ORR_TRY(ln.acquire(idx));
    idx = ln.lane;
    std::lock_guard<std::mutex> lock(idx->mu);
    // the caller's escalation after a merge that could not certify every query (orr_merge_candidates)
    a.no_fuse = pass >= 1;
    a.force_exact = pass >= 2;
    // the floor of the two-stage pass: from the k-th best of the sample when the caller told its topK (valid across shards:
    // the global k-th best is at least every shard's), else from the k'-th
    if (topk > 0) a.topk = std::min<int32_t>(kprime, topk);
    const bool dev_out = is_device_pointer(out);
    idx->sstats.searches += 1; idx->sstats.queries += B;
    // passes that keep one number per (query,row) -- the unfused and the exact one -- run over slices of the batch, so that
    // their workspace stays bounded whatever the batch (1024 queries x 12.5M rows x 8 B = 102 GB in one piece)
    const int64_t n = std::max<int64_t>(1, participating_rows(idx, candidate_limit));
    if (pass >= 1 && B > 1 && (size_t)B * (size_t)n * 8 > kPassWorkspaceBytes) {
        const int32_t per = (int32_t)std::max<size_t>(1, kPassWorkspaceBytes / ((size_t)n * 8));
        for (int32_t b0 = 0; b0 < B; b0 += per) {
            std::vector<int32_t> part((size_t)std::min<int32_t>(per, B - b0));
            std::iota(part.begin(), part.end(), b0);
            SubBatch sb;
            BatchArgs sub;
            ORR_TRY(build_subset(idx, a, part, sb, sub));
            sub.no_fuse = a.no_fuse; sub.force_exact = a.force_exact;
            ORR_TRY(shard_pass_into(idx, sub, kprime, candidate_limit, out, (size_t)b0, dev_out));
        }
        return ORR_OK;
    }
    return shard_pass_into(idx, a, kprime, candidate_limit, out, 0, dev_out);
}

int orr_search_shard(orr_index *idx, int32_t B, int32_t dim, const float *q, const uint8_t *terms_utf8,
                     const uint32_t *term_off, const uint32_t *query_term_off, int64_t now_ticks, int32_t kprime,
                     int64_t candidate_limit, orr_candidate *out)
{
    if (!idx) return fail(ORR_EINVAL, "orr_search_shard: null index");
    int32_t topk = 0, pass = 0;
    {   // the sticky per-index forms of the two arguments ("shard_topk", "shard_pass"); orr_search_shard_ex takes them per call
        std::lock_guard<std::mutex> lock(idx->mu);
        topk = idx->opt_shard_topk; pass = idx->opt_shard_pass;
    }
    return orr_search_shard_ex(idx, B, dim, q, terms_utf8, term_off, query_term_off, now_ticks, kprime, candidate_limit, topk, pass, out);
}

int orr_merge_candidates(int32_t n_shards, int32_t B, int32_t kprime, const orr_candidate *all, int32_t index_dim,
                         int32_t dim, const float *q_host, const uint32_t *query_term_off, int64_t now_ticks,
                         int32_t topk, int64_t *out_rows, double *out_scores, int32_t *out_counts,
                         int32_t *out_uncertified)
{
    if (n_shards < 1 || B < 1 || kprime < 1) return fail(ORR_EINVAL, "orr_merge_candidates: sizes must be positive");
    if (!all || !query_term_off || !out_rows || !out_scores) return fail(ORR_EINVAL, "orr_merge_candidates: null argument");
    if (dim < 0 || index_dim < 0) return fail(ORR_EINVAL, "orr_merge_candidates: negative dimension");
    const bool use_cos = dim > 0 && dim == index_dim;
    if (use_cos && !q_host) return fail(ORR_EINVAL, "orr_merge_candidates: q_host is required with dim %d", dim);
    return merge_impl(n_shards, B, kprime, all, dim, use_cos, q_host, nullptr, query_term_off, now_ticks, topk, out_rows,
                      out_scores, out_counts, out_uncertified);
}

int orr_merge_candidates_ex(int32_t n_shards, int32_t B, int32_t kprime, const orr_candidate *all, int32_t index_dim,
                            int32_t dim, const float *q_host, const uint32_t *query_term_off, int64_t now_ticks,
                            int32_t topk, int64_t *out_rows, double *out_scores, int32_t *out_counts,
                            int32_t *out_uncertified, uint8_t *out_certified)
{
    if (n_shards < 1 || B < 1 || kprime < 1) return fail(ORR_EINVAL, "orr_merge_candidates: sizes must be positive");
    if (!all || !query_term_off || !out_rows || !out_scores) return fail(ORR_EINVAL, "orr_merge_candidates: null argument");
    if (dim < 0 || index_dim < 0) return fail(ORR_EINVAL, "orr_merge_candidates: negative dimension");
    const bool use_cos = dim > 0 && dim == index_dim;
    if (use_cos && !q_host) return fail(ORR_EINVAL, "orr_merge_candidates: q_host is required with dim %d", dim);
    return merge_impl(n_shards, B, kprime, all, dim, use_cos, q_host, nullptr, query_term_off, now_ticks, topk, out_rows,
                      out_scores, out_counts, out_uncertified, out_certified);
}

int orr_search_batch(orr_index *idx, int32_t B, int32_t dim, const float *q, const uint8_t *terms_utf8,
                     const uint32_t *term_off, const uint32_t *query_term_off, int64_t now_ticks, int32_t topk,
                     int64_t candidate_limit, int64_t *out_rows, double *out_scores, int32_t *out_counts)
{
    BatchArgs a{B, dim, q, terms_utf8, term_off, query_term_off, now_ticks, candidate_limit, topk};
    ORR_TRY(check_batch(idx, a, "orr_search_batch"));
    if (!out_rows || !out_scores) return fail(ORR_EINVAL, "orr_search_batch: output buffers are required");
    Lane ln;                                           // concurrent calls on one handle run on different lanes
    ORR_TRY(ln.acquire(idx));
    idx = ln.lane;
    std::lock_guard<std::mutex> lock(idx->mu);
    const int32_t take = std::max<int32_t>(1, topk);
    const int64_t n = participating_rows(idx, candidate_limit);

    // k': the asked k plus a margin, escalated until every query certifies.
    int64_t kprime = std::min<int64_t>(std::max<int64_t>(1, n), std::max<int64_t>((int64_t)take + 22, 32));
    if (kprime > orr::kSelWidth && take + 8 <= orr::kSelWidth) kprime = orr::kSelWidth;
    idx->sstats.searches += 1;
    idx->sstats.queries += B;
    std::vector<int32_t> all((size_t)B);
    std::iota(all.begin(), all.end(), 0);
    const int r = search_ids(idx, a, all, true, kprime, n, out_rows, out_scores, out_counts, 0);
    g_ht.done();
    return r;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// orr_cluster: several shards behind ONE handle in ONE process (the reference host is a single process: Program.cs:59,
// IngestionServiceCollectionExtensions.cs:22-23).  Shard i lives on devices[i] and holds a contiguous range of the global
// candidate order (rows of shard i are all at least as new as those of shard i + 1).  A search runs orr_search_shard's
// device side on every shard at once (one host thread per shard, each bound to its device), the [B][k'+1] records of
// every shard come back through pinned host memory, and the host finishes all queries exactly as orr_merge_candidates
// does -- the record exchange of the multi-process path (RCCL all-gather, sharded.py) without the collective, because
// here every record is wanted in ONE address space.  Escalation is per query, as in orr_search_batch.
// ---------------------------------------------------------------------------------------------------------------------
// RCCL, bound at run time (dlopen: the library stays loadable without it, and the default record exchange does not use it).
// Declarations as in <rccl/rccl.h> (NCCL-compatible ABI): opaque communicator, int result (0 = success), ncclInt8 = 0.
struct RcclApi {
    void *lib = nullptr;
    int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*AllGather)(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t stream) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    static RcclApi *get()
    {
        static RcclApi *api = [] {
            RcclApi *a = new RcclApi();
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                a->lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (a->lib) break;
            }
            if (!a->lib) return a;
            a->CommInitAll = reinterpret_cast<decltype(a->CommInitAll)>(dlsym(a->lib, "ncclCommInitAll"));
            a->CommDestroy = reinterpret_cast<decltype(a->CommDestroy)>(dlsym(a->lib, "ncclCommDestroy"));
            a->AllGather = reinterpret_cast<decltype(a->AllGather)>(dlsym(a->lib, "ncclAllGather"));
            a->GroupStart = reinterpret_cast<decltype(a->GroupStart)>(dlsym(a->lib, "ncclGroupStart"));
            a->GroupEnd = reinterpret_cast<decltype(a->GroupEnd)>(dlsym(a->lib, "ncclGroupEnd"));
            a->GetErrorString = reinterpret_cast<decltype(a->GetErrorString)>(dlsym(a->lib, "ncclGetErrorString"));
            return a;
        }();
        return api;
    }
    bool ok() const { return lib && CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd; }
    const char *text(int r) const { return GetErrorString ? GetErrorString(r) : "RCCL error"; }
};

struct orr_cluster {
    std::vector<orr_index *> shards;
    int32_t dim = 0;
    bool sealed = false;
    std::shared_mutex mu;              // searches share it (they run side by side, each shard search on a lane of its shard); seal / destroy take it alone
    std::mutex stats_mu;
    orr_search_stats sstats{};
    // optional record exchange over RCCL ("exchange" = 1): one communicator per shard device, one all-gather of the per-shard
    // [B][k'+1] records on the shards' exchange streams, the merge reads device 0's gathered copy.  One exchange at a time.
    int exchange = 0;                  // 0: pinned host memory (default); 1: RCCL all-gather over xGMI
    std::mutex rccl_mu;
    std::vector<void *> comms;         // [G], created at the first exchange
    std::vector<hipStream_t> xstreams; // [G]
    std::vector<DevBuf> xsend, xrecv;  // [G] grow-only
    PinnedBuf xhost;
    int64_t rccl_exchanges = 0;        // all-gathers done (orr_cluster_search_stats reports them in reserved[0])
};

namespace {

// Persistent host threads for the shard halves of cluster searches: a search hands shards 1.. to the pool and runs shard 0
// itself (round 2 started G - 1 threads per call).  The pool grows with demand (concurrent searches each need G - 1 workers)
// up to a cap; its threads sleep between tasks and live as long as the process.
class ShardPool {
public:
    static ShardPool &get() { static ShardPool *p = new ShardPool(); return *p; }
    void submit(std::function<void()> task)
    {
        std::unique_lock<std::mutex> l(mu_);
        queue_.push_back(std::move(task));
        if (idle_ == 0 && (int)threads_ < kMaxThreads && !forked_.load(std::memory_order_relaxed)) {
            ++threads_;
            std::thread([this] { loop(); }).detach();
        }
        l.unlock();
        cv_.notify_one();
    }
    bool usable() const { return !forked_.load(std::memory_order_relaxed); }

private:
    static constexpr int kMaxThreads = 128;
    ShardPool() { pthread_atfork(nullptr, nullptr, [] { forked_.store(true); }); }
    void loop()
    {
        std::unique_lock<std::mutex> l(mu_);
        for (;;) {
            ++idle_;
            cv_.wait(l, [this] { return !queue_.empty(); });
            --idle_;
            std::function<void()> task = std::move(queue_.front());
            queue_.pop_front();
            l.unlock();
            task();
            l.lock();
        }
    }
    static inline std::atomic<bool> forked_{false};
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> queue_;
    int idle_ = 0;
    unsigned threads_ = 0;
};

// fn(i) for every shard at once (the calling thread takes shard 0, pool threads the others); returns the first failure
int for_each_shard(int32_t n, const std::function<int(int32_t)> &fn)
{
    std::vector<int> rc((size_t)n, ORR_OK);
    std::vector<std::string> msg((size_t)n);
    std::mutex done_mu;
    std::condition_variable done_cv;
    int pending = 0;
    const bool pooled = ShardPool::get().usable();
    for (int32_t i = 1; i < n; ++i) {
        auto task = [&, i] {
            rc[(size_t)i] = fn(i);
            if (rc[(size_t)i] != ORR_OK) msg[(size_t)i] = g_last_error;
            std::lock_guard<std::mutex> l(done_mu);
            if (--pending == 0) done_cv.notify_one();
        };
        {
            std::lock_guard<std::mutex> l(done_mu);
            ++pending;
        }
        if (pooled) ShardPool::get().submit(task);
        else task();
    }
    rc[0] = fn(0);
    if (rc[0] != ORR_OK) msg[0] = g_last_error;
    {
        std::unique_lock<std::mutex> l(done_mu);
        done_cv.wait(l, [&] { return pending == 0; });
    }
    for (int32_t i = 0; i < n; ++i)
        if (rc[(size_t)i] != ORR_OK) { g_last_error = msg[(size_t)i]; return rc[(size_t)i]; }     // (the detail was set on that shard's thread)
    return ORR_OK;
}

// Communicators, exchange streams and buffers of a cluster's RCCL record exchange (caller holds c->rccl_mu).  Any failure
// switches the cluster back to the pinned-host exchange for good and says why in orr_last_error().
int rccl_prepare(orr_cluster *c, size_t bytes_per_shard)
{
    const int G = (int)c->shards.size();
    RcclApi *api = RcclApi::get();
    auto give_up = [&](const char *why, const char *detail) {
        c->exchange = 0;
        return fail(ORR_ECOMM, "orr_cluster: RCCL exchange disabled (%s%s%s); records travel through pinned host memory", why, detail ? ": " : "", detail ? detail : "");
    };
    if (!api->ok()) return give_up("librccl.so could not be loaded", dlerror());
    if (c->comms.empty()) {
        std::vector<int> devs((size_t)G);
        for (int g = 0; g < G; ++g) {
            devs[(size_t)g] = c->shards[(size_t)g]->device;
            for (int h = 0; h < g; ++h)
                if (devs[(size_t)h] == devs[(size_t)g]) return give_up("two shards share a device, a communicator needs distinct ones", nullptr);
        }
        std::vector<void *> comms((size_t)G, nullptr);
        const int r = api->CommInitAll(comms.data(), G, devs.data());
        if (r != 0) return give_up("ncclCommInitAll failed", api->text(r));
        c->comms = comms;
        c->xstreams.assign((size_t)G, nullptr);
        c->xsend.resize((size_t)G);
        c->xrecv.resize((size_t)G);
        for (int g = 0; g < G; ++g) {
            if (hipSetDevice(devs[(size_t)g]) != hipSuccess || hipStreamCreateWithFlags(&c->xstreams[(size_t)g], hipStreamNonBlocking) != hipSuccess)
                return give_up("cannot create an exchange stream", nullptr);
        }
    }
    for (int g = 0; g < G; ++g) {
        if (hipSetDevice(c->shards[(size_t)g]->device) != hipSuccess) return give_up("hipSetDevice failed", nullptr);
        if (c->xsend[(size_t)g].reserve(bytes_per_shard) != ORR_OK || c->xrecv[(size_t)g].reserve(bytes_per_shard * (size_t)G) != ORR_OK)
            return give_up("no device memory for the exchange buffers", nullptr);
    }
    if (c->xhost.reserve(bytes_per_shard * (size_t)G) != ORR_OK) return give_up("no pinned memory for the gathered records", nullptr);
    return ORR_OK;
}

// One all-gather of the shards' records (grouped: one thread drives every device), then device 0's gathered copy -> out.
int rccl_all_gather(orr_cluster *c, size_t bytes_per_shard, orr_candidate *out)
{
    const int G = (int)c->shards.size();
    RcclApi *api = RcclApi::get();
    int r = api->GroupStart();
    for (int g = 0; g < G && r == 0; ++g)
        r = api->AllGather(c->xsend[(size_t)g].p, c->xrecv[(size_t)g].p, bytes_per_shard, /* ncclInt8 */ 0, c->comms[(size_t)g], c->xstreams[(size_t)g]);
    const int r2 = api->GroupEnd();
    if (r == 0) r = r2;
    if (r != 0) { c->exchange = 0; return fail(ORR_ECOMM, "orr_cluster: ncclAllGather failed: %s", api->text(r)); }
    HIP_TRY(hipSetDevice(c->shards[0]->device));
    HIP_TRY(hipMemcpyAsync(c->xhost.p, c->xrecv[0].p, bytes_per_shard * (size_t)G, hipMemcpyDeviceToHost, c->xstreams[0]));
    for (int g = 0; g < G; ++g) {                       // every device's part of the collective is over before the buffers are reused
        HIP_TRY(hipSetDevice(c->shards[(size_t)g]->device));
        HIP_TRY(hipStreamSynchronize(c->xstreams[(size_t)g]));
    }
    memcpy(out, c->xhost.p, bytes_per_shard * (size_t)G);
    c->rccl_exchanges += 1;
    return ORR_OK;
}

int cluster_search_ids(orr_cluster *c, const BatchArgs &orig, const std::vector<int32_t> &ids, bool whole, int64_t kprime, int64_t n_total,
                       int64_t *out_rows, double *out_scores, int32_t *out_counts, int depth)
{
    const int32_t nb = (int32_t)ids.size(), G = (int32_t)c->shards.size();
    const int32_t take = std::max<int32_t>(1, orig.topk);
    const bool use_cos = orig.dim > 0 && orig.dim == c->dim;
    int64_t n_max = 1;
    for (orr_index *sh : c->shards) n_max = std::max<int64_t>(n_max, participating_rows(sh, orig.candidate_limit));
    if ((orig.no_fuse || orig.force_exact) && nb > 1 && (size_t)nb * (size_t)n_max * 8 > kPassWorkspaceBytes) {
        const int32_t per = (int32_t)std::max<size_t>(1, kPassWorkspaceBytes / ((size_t)n_max * 8));
        for (int32_t i0 = 0; i0 < nb; i0 += per) {
            std::vector<int32_t> part(ids.begin() + i0, ids.begin() + std::min<int32_t>(nb, i0 + per));
            ORR_TRY(cluster_search_ids(c, orig, part, false, kprime, n_total, out_rows, out_scores, out_counts, depth));
        }
        return ORR_OK;
    }
    // the sub-batch in host memory (the cluster's queries are host-resident by contract), built once for all shards
    SubBatch sb;
    BatchArgs cur = orig;
    if (!whole) ORR_TRY(build_subset(c->shards[0], orig, ids, sb, cur));
    std::vector<double> norms;
    if (use_cos) {
        norms.resize((size_t)nb);
        exact_norms(cur.q, nb, cur.dim, norms.data());
        cur.norms_host = norms.data();
    }
    const size_t rec_per_shard = (size_t)nb * ((size_t)kprime + 1);
    std::vector<orr_candidate> all((size_t)G * rec_per_shard);
    std::vector<uint8_t> used_two_stage((size_t)G, 0), used_fused((size_t)G, 0), used_mfma((size_t)G, 0);
    // every shard's half runs on a LANE of that shard (concurrent cluster searches take different lanes); the lanes stay held
    // until this pass has looked at what the screen kept on them
    std::vector<Lane> lanes((size_t)G);
    std::vector<orr_index *> on((size_t)G, nullptr);
    // "exchange" = 1: the shards write their records into per-device send buffers and ONE RCCL all-gather brings every shard's
    // records to every device; the merge reads device 0's copy.  (One exchange at a time per cluster: the communicators are
    // not shared between concurrent collectives.)
    std::unique_lock<std::mutex> rccl_lock(c->rccl_mu, std::defer_lock);
    bool via_rccl = false;
    const size_t rec_bytes_shard = sizeof(orr_candidate) * rec_per_shard;
    if (c->exchange == 1) {
        rccl_lock.lock();
        via_rccl = rccl_prepare(c, rec_bytes_shard) == ORR_OK;
        if (!via_rccl) rccl_lock.unlock();
    }
    ORR_TRY(for_each_shard(G, [&](int32_t g) -> int {
        ORR_TRY(lanes[(size_t)g].acquire(c->shards[(size_t)g]));
        orr_index *sh = lanes[(size_t)g].lane;
        on[(size_t)g] = sh;
        std::lock_guard<std::mutex> lock(sh->mu);
        BatchArgs mine = cur;
        const float *qh = nullptr;
        const orr_candidate *recs = nullptr;
        if (via_rccl) {
            mine.out_dev = c->xsend[(size_t)g].as<orr_candidate>();          // complete when run_shard returns (it synchronises its stream)
            ORR_TRY(run_shard(sh, mine, (int32_t)kprime, false, &qh, &recs));
        } else {
            ORR_TRY(run_shard(sh, mine, (int32_t)kprime, true, &qh, &recs));
            if (recs) memcpy(all.data() + (size_t)g * rec_per_shard, recs, sizeof(orr_candidate) * rec_per_shard);
            else HIP_TRY(hipMemcpy(all.data() + (size_t)g * rec_per_shard, sh->ws_cand.p, sizeof(orr_candidate) * rec_per_shard, hipMemcpyDeviceToHost));
        }
        used_two_stage[(size_t)g] = mine.used_two_stage; used_fused[(size_t)g] = mine.used_fused; used_mfma[(size_t)g] = mine.used_mfma;
        return ORR_OK;
    }));
    if (via_rccl) {
        ORR_TRY(rccl_all_gather(c, rec_bytes_shard, all.data()));
        rccl_lock.unlock();
    }
    std::unique_lock<std::mutex> stats_lock(c->stats_mu);
    c->sstats.passes += 1;
    c->sstats.pass_mode = on[0]->sstats.pass_mode;
    if (depth > 0) c->sstats.requeried += nb;
    stats_lock.unlock();
    std::vector<uint8_t> cert((size_t)nb, 1);
    int32_t unc = 0;
    std::vector<int64_t> rows((size_t)nb * take);
    std::vector<double> scores((size_t)nb * take);
    std::vector<int32_t> counts((size_t)nb);
    ORR_TRY(merge_impl(G, nb, (int32_t)kprime, all.data(), cur.dim, use_cos, cur.q, use_cos ? norms.data() : nullptr, cur.query_term_off,
                       cur.now_ticks, cur.topk, rows.data(), scores.data(), counts.data(), &unc, cert.data()));
    for (int32_t i = 0; i < nb; ++i) {                                      // (a later pass overwrites what could not be certified)
        const size_t b = (size_t)ids[(size_t)i];
        memcpy(out_rows + b * take, rows.data() + (size_t)i * take, sizeof(int64_t) * take);
        memcpy(out_scores + b * take, scores.data() + (size_t)i * take, sizeof(double) * take);
        if (out_counts) out_counts[b] = counts[(size_t)i];
    }
    // survivors of the screening pass, per shard: statistics, and the buffer size a repeat needs
    bool any_fused = false, any_mfma = false, grow = false, only_overflow = unc > 0;
    stats_lock.lock();
    for (int32_t g = 0; g < G; ++g) {
        orr_index *sh = on[(size_t)g];
        any_fused = any_fused || used_fused[(size_t)g];
        any_mfma = any_mfma || used_mfma[(size_t)g];
        if (!used_two_stage[(size_t)g] || (int32_t)sh->h_survivors.size() != nb) continue;
        uint32_t worst = 0;
        for (int32_t i = 0; i < nb; ++i) {
            const uint32_t cnt = sh->h_survivors[(size_t)i];
            c->sstats.survivors_total += cnt;
            if ((int64_t)cnt > c->sstats.survivors_max) c->sstats.survivors_max = cnt;
            if (cnt > sh->pass_cap) { c->sstats.overflowed_queries += 1; if (!cert[(size_t)i]) worst = std::max(worst, cnt); }
        }
        c->sstats.survivor_samples += nb;
        if (worst > 0 && worst < (1u << 19) && (int64_t)worst * 2 < participating_rows(sh, orig.candidate_limit) &&
            (size_t)unc * (size_t)worst * 96 < ((size_t)2 << 30)) {
            uint32_t cap = sh->pass_cap;
            while (cap < worst + worst / 8) cap *= 2;
            if (cap > sh->survivor_cap) { sh->survivor_cap = cap; grow = true; }
            if (grow) {                                                     // (the repeat may run on another lane of this shard: the owner carries the size too)
                orr_index *own = c->shards[(size_t)g];
                std::lock_guard<std::mutex> ll(own->lanes_mu);
                own->survivor_cap_hint = std::max(own->survivor_cap_hint, cap);
            }
        } else if (worst > 0) {
            only_overflow = false;                                          // too many survivors to buffer: a more exact pass instead
        }
        c->sstats.survivor_capacity = std::max<int64_t>(c->sstats.survivor_capacity, sh->survivor_cap);
    }
    stats_lock.unlock();
    if (unc == 0) return ORR_OK;
    // queries uncertified for a reason other than an overflowing buffer need a more exact pass whatever the buffers do
    if (grow) {
        for (int32_t g = 0; g < G && only_overflow; ++g) {
            orr_index *sh = on[(size_t)g];
            if (!used_two_stage[(size_t)g] || (int32_t)sh->h_survivors.size() != nb) { only_overflow = false; break; }
        }
        if (only_overflow)
            for (int32_t i = 0; i < nb && only_overflow; ++i) {
                if (cert[(size_t)i]) continue;
                bool over = false;
                for (int32_t g = 0; g < G; ++g) over = over || on[(size_t)g]->h_survivors[(size_t)i] > on[(size_t)g]->pass_cap;
                only_overflow = over;
            }
    }
    // (a grown buffer size belongs to the lane that measured it; the shard's other lanes learn it when they overflow themselves)
    lanes.clear();                                      // the repeat below takes lanes of its own
    std::vector<int32_t> again;
    for (int32_t i = 0; i < nb; ++i) if (!cert[(size_t)i]) again.push_back(ids[(size_t)i]);
    BatchArgs next = orig;
    if (grow && only_overflow) {
        std::lock_guard<std::mutex> l(c->stats_mu);
        c->sstats.buffer_growths += 1;                                       // the same pass again with buffers sized from the measured counts
    } else if (any_fused && !orig.no_fuse) {
        next.no_fuse = true;
    } else if (any_mfma && !orig.force_exact) {
        next.force_exact = true;
        std::lock_guard<std::mutex> l(c->stats_mu);
        c->sstats.exact_pass_queries += (int64_t)again.size();
    } else if (kprime >= n_total) {
        return ORR_OK;
    } else {
        kprime = std::min<int64_t>(n_total, kprime * 4);
    }
    if (depth >= 40) return fail(ORR_EDEVICE, "orr_cluster_search_batch: escalation did not terminate");
    return cluster_search_ids(c, next, again, false, kprime, n_total, out_rows, out_scores, out_counts, depth + 1);
}

}  // namespace

extern "C" {

int orr_cluster_create(const int32_t *devices, int32_t n_shards, int32_t dim, int64_t capacity_rows_per_shard, orr_cluster **out)
{
    if (!devices || !out || n_shards < 1 || n_shards > 64 || dim < 0 || capacity_rows_per_shard < 0)
        return fail(ORR_EINVAL, "orr_cluster_create: bad argument");
    *out = nullptr;
    orr_cluster *c = new (std::nothrow) orr_cluster();
    if (!c) return fail(ORR_ENOMEM, "out of host memory");
    c->dim = dim;
    for (int32_t i = 0; i < n_shards; ++i) {
        orr_config cfg{(int32_t)sizeof(orr_config), devices[i], dim, 0, capacity_rows_per_shard, 0};
        orr_index *sh = nullptr;
        const int r = orr_index_create(&cfg, &sh);
        if (r != ORR_OK) { const std::string keep = g_last_error; orr_cluster_destroy(c); g_last_error = keep; return r; }
        c->shards.push_back(sh);
    }
    *out = c;
    return ORR_OK;
}

void orr_cluster_destroy(orr_cluster *c)
{
    if (!c) return;
    if (!c->comms.empty()) {
        RcclApi *api = RcclApi::get();
        for (size_t g = 0; g < c->comms.size(); ++g) {
            if (g < c->shards.size()) (void)hipSetDevice(c->shards[g]->device);
            if (g < c->xstreams.size() && c->xstreams[g]) { (void)hipStreamSynchronize(c->xstreams[g]); (void)hipStreamDestroy(c->xstreams[g]); }
            if (c->comms[g] && api->CommDestroy) (void)api->CommDestroy(c->comms[g]);
            if (g < c->xsend.size()) c->xsend[g].release();
            if (g < c->xrecv.size()) c->xrecv[g].release();
        }
    }
    c->xhost.release();
    for (orr_index *sh : c->shards) orr_index_destroy(sh);
    delete c;
}

int orr_cluster_set_option(orr_cluster *c, const char *name, int64_t value)
{
    if (!c || !name) return fail(ORR_EINVAL, "orr_cluster_set_option: null argument");
    std::unique_lock<std::shared_mutex> lock(c->mu);
    if (strcmp(name, "exchange") == 0) {
        if (value != 0 && value != 1) return fail(ORR_EINVAL, "orr_cluster_set_option: exchange takes 0 (pinned host memory) or 1 (RCCL all-gather)");
        if (value == 1) {
            if (!RcclApi::get()->ok()) return fail(ORR_ECOMM, "orr_cluster_set_option: librccl.so could not be loaded");
            for (size_t g = 0; g < c->shards.size(); ++g)
                for (size_t h = 0; h < g; ++h)
                    if (c->shards[h]->device == c->shards[g]->device)
                        return fail(ORR_EINVAL, "orr_cluster_set_option: the RCCL exchange needs every shard on a device of its own (shards %zu and %zu share device %d)",
                                    h, g, c->shards[g]->device);
        }
        std::lock_guard<std::mutex> l(c->rccl_mu);
        c->exchange = (int)value;
        return ORR_OK;
    }
    return fail(ORR_EINVAL, "orr_cluster_set_option: unknown option %s", name);
}

int32_t orr_cluster_shards(const orr_cluster *c) { return c ? (int32_t)c->shards.size() : 0; }

orr_index *orr_cluster_shard(orr_cluster *c, int32_t i)
{
    if (!c || i < 0 || i >= (int32_t)c->shards.size()) { (void)fail(ORR_EINVAL, "orr_cluster_shard: no shard %d", i); return nullptr; }
    return c->shards[(size_t)i];
}

int64_t orr_cluster_rows(const orr_cluster *c)
{
    int64_t n = 0;
    if (c) for (const orr_index *sh : c->shards) n += sh->n_rows;
    return n;
}

int orr_cluster_seal(orr_cluster *c)
{
    if (!c) return fail(ORR_EINVAL, "orr_cluster_seal: null cluster");
    std::unique_lock<std::shared_mutex> lock(c->mu);
    ORR_TRY(for_each_shard((int32_t)c->shards.size(), [&](int32_t g) { return orr_index_seal(c->shards[(size_t)g]); }));
    int64_t base = 0, dead = 0;
    const orr_index *prev = nullptr;
    for (orr_index *sh : c->shards) {
        // the global candidate order must be shard 0's rows, then shard 1's, ...: nothing in a shard may be newer than a row in front of it
        if (prev && !prev->h_created.empty() && !sh->h_created.empty() && sh->h_created.front() > prev->h_created.back())
            return fail(ORR_EINVAL, "orr_cluster_seal: a shard holds a row newer than one of the shard in front of it; "
                                    "partition the rows by CreatedAtUtc, newest first");
        ORR_TRY(orr_index_set_row_base(sh, base));
        ORR_TRY(orr_index_set_option(sh, "dead_rows_before", dead));
        base += sh->n_rows;
        dead += (int64_t)sh->dead.size();
        if (sh->n_rows > 0) prev = sh;
    }
    c->sealed = true;
    return ORR_OK;
}

// places the shards in the global candidate order: row_base and the deleted rows in front of each
static int place_shards(orr_cluster *c)
{
    int64_t base = 0, dead = 0;
    for (orr_index *sh : c->shards) {
        ORR_TRY(orr_index_set_row_base(sh, base));
        ORR_TRY(orr_index_set_option(sh, "dead_rows_before", dead));
        base += sh->n_rows;
        dead += (int64_t)sh->dead.size();
    }
    return ORR_OK;
}

int orr_cluster_compact(orr_cluster *c, int64_t *out_removed)
{
    if (out_removed) *out_removed = 0;
    if (!c) return fail(ORR_EINVAL, "orr_cluster_compact: null cluster");
    std::unique_lock<std::shared_mutex> lock(c->mu);   // no search in flight on the cluster
    if (!c->sealed) return fail(ORR_ESTATE, "orr_cluster_compact: the cluster is not sealed");
    std::vector<int64_t> removed(c->shards.size(), 0);
    ORR_TRY(for_each_shard((int32_t)c->shards.size(), [&](int32_t g) { return orr_index_compact(c->shards[(size_t)g], &removed[(size_t)g]); }));
    ORR_TRY(place_shards(c));                          // the shards behind a compacted one move up in the global order
    if (out_removed) for (int64_t r : removed) *out_removed += r;
    return ORR_OK;
}

int orr_cluster_search_batch(orr_cluster *c, int32_t B, int32_t dim, const float *q_host, const uint8_t *terms_utf8,
                             const uint32_t *term_off, const uint32_t *query_term_off, int64_t now_ticks, int32_t topk,
                             int64_t candidate_limit, int64_t *out_rows, double *out_scores, int32_t *out_counts)
{
    if (!c) return fail(ORR_EINVAL, "orr_cluster_search_batch: null cluster");
    BatchArgs a{B, dim, q_host, terms_utf8, term_off, query_term_off, now_ticks, candidate_limit, topk};
    ORR_TRY(check_batch(c->shards[0], a, "orr_cluster_search_batch"));
    if (!out_rows || !out_scores) return fail(ORR_EINVAL, "orr_cluster_search_batch: output buffers are required");
    if (dim > 0 && is_device_pointer(q_host)) return fail(ORR_EINVAL, "orr_cluster_search_batch: the query vectors must be in host memory (every shard's device reads them)");
    std::shared_lock<std::shared_mutex> lock(c->mu);   // searches run side by side; seal and destroy are exclusive
    if (!c->sealed) return fail(ORR_ESTATE, "orr_cluster_search_batch: the cluster is not sealed");
    {   // rows deleted from a shard since the seal (orr_index_delete_rows on orr_cluster_shard(i)) shift the candidate_limit prefix
        // of every shard behind it: when the dead-row prefix sums changed they are set again (an option of the shard: every lane
        // of it is idle while it changes)
        int64_t dead = 0;
        for (orr_index *sh : c->shards) {
            int64_t have, mine;
            {
                std::lock_guard<std::mutex> sl(sh->lanes_mu);
                have = sh->dead_before_pub;
                mine = sh->dead_count_pub;
            }
            if (have != dead) ORR_TRY(orr_index_set_option(sh, "dead_rows_before", dead));
            dead += mine;
        }
    }
    const int32_t take = std::max<int32_t>(1, topk);
    int64_t n_total = 0;
    for (orr_index *sh : c->shards) n_total += participating_rows(sh, candidate_limit);
    // k' per shard as orr_search_batch picks it for one shard: any shard may hold the whole top-k
    int64_t kprime = std::min<int64_t>(std::max<int64_t>(1, n_total), std::max<int64_t>((int64_t)take + 22, 32));
    if (kprime > orr::kSelWidth && take + 8 <= orr::kSelWidth) kprime = orr::kSelWidth;
    {
        std::lock_guard<std::mutex> l(c->stats_mu);
        c->sstats.searches += 1;
        c->sstats.queries += B;
    }
    for (int64_t i = 0; i < (int64_t)B * take; ++i) { out_rows[i] = -1; out_scores[i] = 0.0; }
    std::vector<int32_t> all((size_t)B);
    std::iota(all.begin(), all.end(), 0);
    return cluster_search_ids(c, a, all, true, kprime, n_total, out_rows, out_scores, out_counts, 0);
}

int orr_cluster_search_stats(orr_cluster *c, orr_search_stats *out, int32_t reset)
{
    if (!c) return fail(ORR_EINVAL, "orr_cluster_search_stats: null cluster");
    std::lock_guard<std::mutex> lock(c->stats_mu);
    if (out) { *out = c->sstats; out->reserved[0] = c->rccl_exchanges; }
    if (reset) c->sstats = orr_search_stats{};
    return ORR_OK;
}

}  // extern "C"
