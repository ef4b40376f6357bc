"""Concurrent searches on ONE handle ("orr_search_batch must be re-entrant on a sealed index"; the reference
creates a scoped RecallSearchService per request, Program.cs:59, over a lock-free store, InMemoryIngestionStore.cs:8-9).
The library gives every in-flight search a lane of its own (the index's workspaces or an internal view's); callers share
one handle and never see a view."""
import threading
import time

import numpy as np
import pytest

from helpers import NOW, DAY, orc, pkg

pytestmark = pytest.mark.gpu

TEXTS = ["alpha the helm", "kubernetes", "what is the", "GAMMA zzz", "azure cosmos vector search", ""]


def _corpus(rng, n, dim):
    emb = rng.standard_normal((n, dim)).astype(np.float32)
    created = np.sort(NOW - rng.integers(0, 300 * DAY, n))[::-1].astype(np.int64)
    words = np.array(["alpha", "beta", "gamma", "delta", "kubernetes", "helm", "azure", "cosmos", "vector", "search"])
    contents = [" ".join(w) for w in words[rng.integers(0, len(words), (n, 5))]]
    return emb, created, contents


def test_eight_threads_on_one_handle_equal_the_oracle_and_overlap():
    P = pkg()
    rng = np.random.default_rng(31)
    n, dim = 300_000, 128
    emb, created, contents = _corpus(rng, n, dim)
    idx = P.RecallIndex(dim=dim)
    for r0 in range(0, n, 100_000):
        idx.append(emb[r0:r0 + 100_000], created[r0:r0 + 100_000], [s.encode() for s in contents[r0:r0 + 100_000]])
    idx.seal()
    corpus = orc.OracleCorpus(emb, created, contents)
    n_threads, per = 8, 40
    qs = rng.standard_normal((n_threads * per, dim)).astype(np.float32)
    texts = [TEXTS[i % len(TEXTS)] for i in range(n_threads * per)]
    terms = [P.text.query_terms(t) for t in texts]
    # serial reference: the same single-query searches one after the other from one thread
    idx.search(qs[:1], terms[:1], NOW, 10, candidate_limit=n)
    t0 = time.perf_counter()
    serial = [idx.search(qs[i:i + 1], terms[i:i + 1], NOW, 10, candidate_limit=n) for i in range(n_threads * per)]
    t_serial = time.perf_counter() - t0
    out = [None] * (n_threads * per)
    errors = []
    barrier = threading.Barrier(n_threads)

    def work(t):
        try:
            barrier.wait()
            for i in range(t * per, (t + 1) * per):
                out[i] = idx.search(qs[i:i + 1], terms[i:i + 1], NOW, 10, candidate_limit=n)
        except Exception as exc:                      # noqa: BLE001
            errors.append(repr(exc))

    for _ in range(2):                                # (the first round creates the lanes and their workspaces)
        threads = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
        t0 = time.perf_counter()
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        t_conc = time.perf_counter() - t0
    assert not errors, errors[:3]
    for i in range(n_threads * per):
        assert np.array_equal(out[i][0], serial[i][0]) and np.array_equal(out[i][1], serial[i][1]) and np.array_equal(out[i][2], serial[i][2]), i
    for i in (0, 7, 41, 133, 319):
        orow, osc, _ = corpus.search(qs[i], texts[i], NOW, 10, candidate_limit=n, threads=8)
        assert list(out[i][0][0, :out[i][2][0]]) == list(orow) and np.array_equal(out[i][1][0, :out[i][2][0]], osc), i
    st = idx.search_stats()
    assert st["searches"] >= 3 * n_threads * per          # every lane's counters are in the handle's statistics
    print("one handle, %d threads x %d single-query searches: serial %.1f ms, concurrent %.1f ms (x%.2f)" %
          (n_threads, per, 1e3 * t_serial, 1e3 * t_conc, t_serial / t_conc))
    assert t_conc < 0.75 * t_serial, (t_serial, t_conc)    # the searches really overlap (4 lanes by default)
    # options reach every lane; max_lanes = 1 serialises again (and stays correct)
    idx.set_option("two_stage", 0)
    r0 = idx.search(qs[:4], terms[:4], NOW, 10, candidate_limit=n)
    idx.set_option("two_stage", 1)
    r1 = idx.search(qs[:4], terms[:4], NOW, 10, candidate_limit=n)
    assert np.array_equal(r0[0], r1[0]) and np.array_equal(r0[1], r1[1])
    # deletes wait for the lanes and are seen by all of them
    victim = int(out[0][0][0, 0])
    assert idx.delete_rows([victim]) == 1
    res = [None] * 4

    def after_delete(t):
        res[t] = idx.search(qs[:1], terms[:1], NOW, 10, candidate_limit=n)

    ths = [threading.Thread(target=after_delete, args=(t,)) for t in range(4)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    for t in range(4):
        assert victim not in list(res[t][0][0]) and np.array_equal(res[t][0], res[0][0])
    idx.close()


def test_concurrent_cluster_searches_equal_the_oracle():
    P = pkg()
    rng = np.random.default_rng(32)
    n, dim = 6000, 64
    emb, created, contents = _corpus(rng, n, dim)
    corpus = orc.OracleCorpus(emb, created, contents)
    cl = P.RecallCluster([0, 0, 0], dim)
    bounds = [0, 2000, 4000, n]
    for g in range(3):
        lo, hi = bounds[g], bounds[g + 1]
        cl.shard(g).append(emb[lo:hi], created[lo:hi], [s.encode() for s in contents[lo:hi]], row_ids=np.arange(lo, hi, dtype=np.int64))
    cl.seal()
    n_threads, per = 6, 10
    qs = rng.standard_normal((n_threads * per, 3, dim)).astype(np.float32)
    out = [None] * (n_threads * per)
    errors = []

    def work(t):
        try:
            for i in range(t * per, (t + 1) * per):
                tt = [P.text.query_terms(TEXTS[(i + j) % len(TEXTS)]) for j in range(3)]
                out[i] = cl.search(qs[i], tt, NOW, 7, candidate_limit=n if i % 3 else 2500)
        except Exception as exc:                      # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    for i in range(n_threads * per):
        rows, scores, counts = out[i]
        for j in range(3):
            orow, osc, _ = corpus.search(qs[i, j], TEXTS[(i + j) % len(TEXTS)], NOW, 7, candidate_limit=n if i % 3 else 2500)
            assert list(rows[j, :counts[j]]) == list(orow) and np.array_equal(scores[j, :counts[j]], osc), (i, j)
    assert cl.search_stats()["searches"] == n_threads * per
    cl.close()
