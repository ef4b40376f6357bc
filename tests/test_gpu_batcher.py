"""Micro-batcher: many request threads, one query each, coalesced into batched
orr_search_batch calls; every caller must get exactly what a lone search returns, and concurrent
un-batched searches from several threads must be safe too (per-index serialisation)."""
import threading

import numpy as np
import pytest

from helpers import DAY, NOW, build_index, oracle_corpus, pkg, random_corpus

pytestmark = pytest.mark.gpu


def test_concurrent_requests_are_batched_and_exact():
    P = pkg()
    rng = np.random.default_rng(17)
    n, dim = 4000, 128
    c = random_corpus(rng, n, dim)
    idx = build_index(c)
    corpus = oracle_corpus(c)
    n_req = 96
    qs = rng.standard_normal((n_req, dim)).astype(np.float32)
    texts = [["alpha kubernetes", "the helm", "GAMMA delta zzz", "net ab"][i % 4] for i in range(n_req)]
    topks = [[10, 3, 12, 1][i % 4] for i in range(n_req)]
    batcher = P.MicroBatcher(idx, max_batch=32, max_wait_us=20000)
    results = [None] * n_req
    barrier = threading.Barrier(n_req)

    def work(i):
        barrier.wait()
        results[i] = batcher.search(qs[i], P.text.query_terms(texts[i]), NOW, topks[i], candidate_limit=n)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(n_req)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    st = batcher.stats()
    assert st["requests"] == n_req and st["batches"] < n_req and st["largest_batch"] > 1, st
    for i in range(n_req):
        orow, osc, _ = corpus.search(qs[i], texts[i], NOW, topks[i], candidate_limit=n)
        rows, scores = results[i]
        assert list(rows) == list(orow) and np.array_equal(scores, osc), i
    batcher.close()
    idx.close()


def test_requests_with_their_own_clocks_still_share_batches():
    """The C# shim passes DateTime.UtcNow.Ticks per request: no two requests carry the same now_ticks.  They must still
    be coalesced; a batch is answered at its latest clock, which the caller can ask for -- the result then equals the
    oracle's at that clock."""
    P = pkg()
    rng = np.random.default_rng(19)
    n, dim = 3000, 64
    c = random_corpus(rng, n, dim)
    idx = build_index(c)
    corpus = oracle_corpus(c)
    n_req = 64
    qs = rng.standard_normal((n_req, dim)).astype(np.float32)
    texts = [["alpha kubernetes", "the helm", "GAMMA delta zzz", "net ab"][i % 4] for i in range(n_req)]
    nows = [NOW + 1_000 * i + int(rng.integers(0, 999)) for i in range(n_req)]          # all different, 100 us apart
    batcher = P.MicroBatcher(idx, max_batch=32, max_wait_us=20000)
    results = [None] * n_req
    barrier = threading.Barrier(n_req)

    def work(i):
        barrier.wait()
        results[i] = batcher.search(qs[i], P.text.query_terms(texts[i]), nows[i], 5, candidate_limit=n, with_clock=True)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(n_req)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    st = batcher.stats()
    assert st["requests"] == n_req and st["largest_batch"] > 1 and st["batches"] < n_req, st
    for i in range(n_req):
        rows, scores, clock = results[i]
        assert clock in nows and clock >= nows[i], (i, clock)
        orow, osc, _ = corpus.search(qs[i], texts[i], clock, 5, candidate_limit=n)
        assert list(rows) == list(orow) and np.array_equal(scores, osc), i
    batcher.close()
    idx.close()


def test_replayed_clocks_hours_apart_do_not_share_a_batch():
    """Requests with explicit clocks far apart (tests, deterministic re-ranking, backfills) are answered at THEIR OWN clock:
    only clocks within the collection window (+ 1 s of queueing slack) may share a batch."""
    P = pkg()
    rng = np.random.default_rng(23)
    n, dim = 2000, 64
    c = random_corpus(rng, n, dim)
    idx = build_index(c)
    corpus = oracle_corpus(c)
    n_req = 12
    qs = rng.standard_normal((n_req, dim)).astype(np.float32)
    hour = 36_000_000_000
    nows = [NOW - (i % 3) * 5 * hour + i for i in range(n_req)]                       # three clock groups, five hours apart
    batcher = P.MicroBatcher(idx, max_batch=32, max_wait_us=50000)
    results = [None] * n_req
    barrier = threading.Barrier(n_req)

    def work(i):
        barrier.wait()
        results[i] = batcher.search(qs[i], P.text.query_terms("alpha helm"), nows[i], 5, candidate_limit=n, with_clock=True)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(n_req)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert batcher.stats()["batches"] >= 3
    for i in range(n_req):
        rows, scores, clock = results[i]
        assert abs(clock - nows[i]) < hour, (i, clock - nows[i])                       # answered within its own group
        orow, osc, _ = corpus.search(qs[i], "alpha helm", clock, 5, candidate_limit=n)
        assert list(rows) == list(orow) and np.array_equal(scores, osc), i
    batcher.close()
    idx.close()


def test_concurrent_unbatched_searches_are_serialised_safely():
    P = pkg()
    rng = np.random.default_rng(18)
    n, dim = 3000, 64
    c = random_corpus(rng, n, dim)
    idx = build_index(c)
    corpus = oracle_corpus(c)
    qs = rng.standard_normal((24, dim)).astype(np.float32)
    out = [None] * 24

    def work(i):
        out[i] = idx.search(qs[i:i + 1], [P.text.query_terms("alpha beta")], NOW, 5, candidate_limit=n)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(24)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for i in range(24):
        orow, osc, _ = corpus.search(qs[i], "alpha beta", NOW, 5, candidate_limit=n)
        assert list(out[i][0][0]) == list(orow) and np.array_equal(out[i][1][0], osc)
    idx.close()


def test_views_search_concurrently_and_agree_with_the_owner():
    """orr_index_view: two lanes over one sealed shard searched from two threads at once return what the
    owner returns alone (single queries and batches, two-stage pass included)."""
    import threading
    P = pkg()
    rng = np.random.default_rng(123)
    n, dim = 200_000, 64
    emb = rng.standard_normal((n, dim)).astype(np.float32)
    created = np.sort(NOW - rng.integers(0, 300 * DAY, n))[::-1].astype(np.int64)
    words = np.array(["alpha", "beta", "gamma", "delta", "kubernetes", "helm"])
    contents = [" ".join(w).encode() for w in words[rng.integers(0, len(words), (n, 4))]]
    idx = P.RecallIndex(dim=dim)
    for r0 in range(0, n, 50_000):
        idx.append(emb[r0:r0 + 50_000], created[r0:r0 + 50_000], contents[r0:r0 + 50_000])
    idx.seal()
    view = idx.view()
    with pytest.raises(P.OrrError):
        view.append(emb[:1], created[:1], contents[:1])
    with pytest.raises(P.OrrError):
        view.view()                                    # views are taken of the owner
    with pytest.raises(P.OrrError):
        view.save("/tmp/orr_view_must_not_save.bin")
    qs = rng.standard_normal((64, dim)).astype(np.float32)
    terms = [[b"alpha"], [b"helm", b"beta"], []] * 21 + [[b"gamma"]]
    want_single = [idx.search(qs[b:b + 1], terms[b:b + 1], NOW, 10, candidate_limit=n) for b in range(16)]
    want_batch = idx.search(qs, terms, NOW, 10, candidate_limit=n)
    got = {}

    def lane(name, index):
        out = []
        for rep in range(3):
            for b in range(16):
                out.append(index.search(qs[b:b + 1], terms[b:b + 1], NOW, 10, candidate_limit=n))
            out.append(index.search(qs, terms, NOW, 10, candidate_limit=n))
        got[name] = out

    threads = [threading.Thread(target=lane, args=("owner", idx)), threading.Thread(target=lane, args=("view", view))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for name in ("owner", "view"):
        res = got[name]
        assert len(res) == 3 * 17
        for rep in range(3):
            for b in range(16):
                assert all(np.array_equal(x, y) for x, y in zip(res[rep * 17 + b], want_single[b])), (name, rep, b)
            assert all(np.array_equal(x, y) for x, y in zip(res[rep * 17 + 16], want_batch)), (name, rep)
    view.close()
    idx.close()
