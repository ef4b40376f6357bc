// orr_batcher.cpp -- request micro-batcher in front of orr_search_batch
// (include/omnirecall_host.h).  Callers block on a condition variable until their slice of the
// batch result is ready.  Two worker threads when the index is sealed: the second one searches
// through a view of the index (orr_index_view: own streams and workspaces, shared corpus), so
// while one batch is in its screening pass the next one is already collected and running its
// keyword chain and ranking pass.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/omnirecall_hip.h"
#include "../../../include/omnirecall_host.h"
#include "orr_store.h"

namespace {

struct Request {
    int32_t dim;
    const float *q;
    const uint8_t *terms;
    const uint32_t *term_off;
    int32_t n_terms;
    int64_t now_ticks;
    int32_t topk;
    int64_t candidate_limit;
    int64_t *out_rows;
    double *out_scores;
    int32_t *out_count;
    int status = 1;            // 1 = pending
    bool done = false;
    int64_t batch_now = 0;     // the clock the batch was answered at (see run_batch)
    std::string error;         // the library's message when status != ORR_OK (it was set on the worker's thread)
};

}  // namespace

struct orrh_batcher {
    orr_index *index = nullptr;
    int32_t max_batch = 64;
    int32_t max_wait_us = 200;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<Request *> queue;
    bool stop = false;
    std::thread worker, worker2;
    orr_index *view = nullptr;     // second lane (null when the index was not sealed at creation)
    int64_t batches = 0, requests = 0;
    int32_t largest = 0;
};

namespace {

// Requests that may share a batch.  NOT their clocks: the C# shim passes DateTime.UtcNow.Ticks per request (100 ns
// resolution), so no two production requests ever carry the same now_ticks; the batch is answered at ONE clock, the
// latest of its requests' (the reference reads the clock per chunk, RecallSearchService.cs:117; one frozen clock per
// search is this build's documented rule F3, and requests of one batch are at most max_wait_us apart).
// ... but only clocks that can belong to one batch: requests that really arrive together are at most the collection window
// plus the time they queued behind a running batch apart (one second of slack covers that); a caller that passes explicit or
// replayed clocks (tests, deterministic re-ranking, backfills) hours apart gets its own batch per clock instead of scores
// computed at somebody else's now_ticks.
bool compatible(const Request *a, const Request *b, int64_t clock_slack_ticks)
{
    const int64_t d = a->now_ticks > b->now_ticks ? a->now_ticks - b->now_ticks : b->now_ticks - a->now_ticks;
    return a->dim == b->dim && a->candidate_limit == b->candidate_limit && d <= clock_slack_ticks;
}

void run_batch(orr_index *index, std::vector<Request *> &batch)
{
    const int32_t B = (int32_t)batch.size();
    const int32_t dim = batch[0]->dim;
    int32_t topk = 1;
    int64_t now = batch[0]->now_ticks;
    for (auto r : batch) { topk = std::max(topk, std::max(1, r->topk)); now = std::max(now, r->now_ticks); }
    std::vector<float> q((size_t)B * (size_t)std::max(dim, 0) + 1);
    std::vector<uint8_t> terms;
    std::vector<uint32_t> term_off{0}, qoff{0};
    for (int32_t i = 0; i < B; ++i) {
        const Request *r = batch[i];
        if (dim > 0) memcpy(q.data() + (size_t)i * dim, r->q, sizeof(float) * (size_t)dim);
        for (int32_t t = 0; t < r->n_terms; ++t) {
            const uint32_t o = r->term_off[t], e = r->term_off[t + 1];
            terms.insert(terms.end(), r->terms + o, r->terms + e);
            term_off.push_back((uint32_t)terms.size());
        }
        qoff.push_back((uint32_t)term_off.size() - 1);
    }
    terms.push_back(0);
    std::vector<int64_t> rows((size_t)B * topk, -1);
    std::vector<double> scores((size_t)B * topk, 0.0);
    std::vector<int32_t> counts((size_t)B, 0);
    const int st = orr_search_batch(index, B, dim, dim > 0 ? q.data() : nullptr, terms.data(), term_off.data(), qoff.data(),
                                    now, topk, batch[0]->candidate_limit, rows.data(), scores.data(),
                                    counts.data());
    const std::string err = st == ORR_OK ? std::string() : std::string(orr_last_error());
    for (int32_t i = 0; i < B; ++i) {
        Request *r = batch[i];
        r->status = st;
        r->batch_now = now;
        r->error = err;
        if (st == ORR_OK) {
            const int32_t k = std::max(1, r->topk);
            const int32_t n = std::min(k, counts[i]);          // a smaller topk is a prefix of the batch's
            for (int32_t j = 0; j < k; ++j) {
                r->out_rows[j] = j < n ? rows[(size_t)i * topk + j] : -1;
                r->out_scores[j] = j < n ? scores[(size_t)i * topk + j] : 0.0;
            }
            if (r->out_count) *r->out_count = n;
        }
    }
}

void worker_loop(orrh_batcher *b, orr_index *index)
{
    std::unique_lock<std::mutex> lk(b->mu);
    for (;;) {
        b->cv_work.wait(lk, [b] { return b->stop || !b->queue.empty(); });
        if (b->stop && b->queue.empty()) return;
        // collect: the head request and every compatible one that shows up within the window
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(b->max_wait_us);
        std::vector<Request *> batch;
        for (;;) {
            for (auto it = b->queue.begin(); it != b->queue.end() && (int32_t)batch.size() < b->max_batch;) {
                if (batch.empty() || compatible(batch[0], *it, (int64_t)b->max_wait_us * 10 + 10000000)) {     // (10 ticks per microsecond)
                    batch.push_back(*it);
                    it = b->queue.erase(it);
                } else {
                    ++it;
                }
            }
            if ((int32_t)batch.size() >= b->max_batch || b->stop) break;
            if (b->cv_work.wait_until(lk, deadline) == std::cv_status::timeout) break;
        }
        lk.unlock();
        run_batch(index, batch);
        lk.lock();
        b->batches += 1;
        b->requests += (int64_t)batch.size();
        b->largest = std::max(b->largest, (int32_t)batch.size());
        for (auto r : batch) r->done = true;
        b->cv_done.notify_all();
    }
}

}  // namespace

extern "C" {

orrh_batcher *orrh_batcher_create(void *index, int32_t max_batch, int32_t max_wait_us)
{
    if (!index || max_batch < 1 || max_wait_us < 0) return nullptr;
    orrh_batcher *b = new orrh_batcher();
    b->index = static_cast<orr_index *>(index);
    b->max_batch = max_batch;
    b->max_wait_us = max_wait_us;
    b->worker = std::thread(worker_loop, b, b->index);
    if (orr_index_view(b->index, &b->view) == ORR_OK) b->worker2 = std::thread(worker_loop, b, b->view);
    return b;
}

void orrh_batcher_destroy(orrh_batcher *b)
{
    if (!b) return;
    {
        std::lock_guard<std::mutex> l(b->mu);
        b->stop = true;
    }
    b->cv_work.notify_all();
    if (b->worker.joinable()) b->worker.join();
    if (b->worker2.joinable()) b->worker2.join();
    if (b->view) orr_index_destroy(b->view);
    delete b;
}

int orrh_batcher_search_at(orrh_batcher *b, int32_t dim, const float *q, const uint8_t *terms_utf8, const uint32_t *term_off,
                           int32_t n_terms, int64_t now_ticks, int32_t topk, int64_t candidate_limit, int64_t *out_rows,
                           double *out_scores, int32_t *out_count, int64_t *out_batch_now)
{
    if (!b || !out_rows || !out_scores || dim < 0 || n_terms < 0 || (dim > 0 && !q) || (n_terms > 0 && (!terms_utf8 || !term_off)))
        return orrh_detail::fail(ORR_EINVAL, "orrh_batcher_search: bad argument");
    Request r{dim, q, terms_utf8, term_off, n_terms, now_ticks, topk, candidate_limit, out_rows, out_scores, out_count};
    std::unique_lock<std::mutex> lk(b->mu);
    if (b->stop) return orrh_detail::fail(ORR_ESTATE, "orrh_batcher_search: the batcher is shutting down");
    b->queue.push_back(&r);
    b->cv_work.notify_one();
    b->cv_done.wait(lk, [&r] { return r.done; });
    if (out_batch_now) *out_batch_now = r.batch_now;
    if (r.status != ORR_OK) return orrh_detail::fail(r.status, r.error);      // the text crosses from the worker's thread to the caller's
    return r.status;
}

int orrh_batcher_search(orrh_batcher *b, int32_t dim, const float *q, const uint8_t *terms_utf8, const uint32_t *term_off,
                        int32_t n_terms, int64_t now_ticks, int32_t topk, int64_t candidate_limit, int64_t *out_rows,
                        double *out_scores, int32_t *out_count)
{
    return orrh_batcher_search_at(b, dim, q, terms_utf8, term_off, n_terms, now_ticks, topk, candidate_limit, out_rows, out_scores,
                                  out_count, nullptr);
}

void orrh_batcher_stats(orrh_batcher *b, int64_t *batches, int64_t *requests, int32_t *largest_batch)
{
    if (!b) return;
    std::lock_guard<std::mutex> l(b->mu);
    if (batches) *batches = b->batches;
    if (requests) *requests = b->requests;
    if (largest_batch) *largest_batch = b->largest;
}

}  // extern "C"
