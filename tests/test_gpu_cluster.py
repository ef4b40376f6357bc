"""orr_cluster: several shards behind one handle in one process (the single-process multi-GPU form of the C ABI).
Two and three "devices" are shards on cuda:0 here; results must equal the oracle over the whole corpus, including
the escalation paths (ties at the cut, k' growth, buffers that overflow on one shard only)."""
import numpy as np
import pytest

from helpers import NOW, DAY, orc, pkg, random_corpus

pytestmark = pytest.mark.gpu

QUERY_TEXTS = ["alpha the helm", "kubernetes", "what is the", "GAMMA zzz", "azure cosmos vector search", ""]


def _sorted_corpus(rng, n, dim, **kw):
    c = random_corpus(rng, n, dim, **kw)
    order = np.argsort(-c["created"], kind="stable")
    return {"emb": [c["emb"][i] for i in order], "created": c["created"][order], "contents": [c["contents"][i] for i in order], "dim": dim}


def _fill(P, cl, c, bounds):
    lower = [P.text.lower_invariant(s) for s in c["contents"]]
    for g in range(len(bounds) - 1):
        lo, hi = bounds[g], bounds[g + 1]
        sh = cl.shard(g)
        r = lo
        while r < hi:                                    # runs of rows with / without an embedding
            has = c["emb"][r] is not None
            e = r
            while e < hi and (c["emb"][e] is not None) == has:
                e += 1
            emb = np.stack(c["emb"][r:e]).astype(np.float32) if has and c["dim"] > 0 else None
            sh.append(emb, c["created"][r:e], lower[r:e], row_ids=np.arange(r, e, dtype=np.int64))
            r = e
    cl.seal()


@pytest.mark.parametrize("n_shards", [2, 3])
def test_cluster_equals_the_oracle(n_shards):
    P = pkg()
    rng = np.random.default_rng(500 + n_shards)
    n, dim = 2400, 64
    c = _sorted_corpus(rng, n, dim)
    corpus = orc.OracleCorpus(c["emb"], c["created"], c["contents"])
    cl = P.RecallCluster([0] * n_shards, dim)
    bounds = [n * g // n_shards for g in range(n_shards + 1)]
    _fill(P, cl, c, bounds)
    assert cl.rows == n and cl.n_shards == n_shards
    B = 9
    qs = rng.standard_normal((B, dim)).astype(np.float32)
    qs[3] = 0.0
    texts = [QUERY_TEXTS[b % len(QUERY_TEXTS)] for b in range(B)]
    terms = [P.text.query_terms(t) for t in texts]
    for topk, limit in ((10, n), (3, 300), (40, n), (70, n), (5, bounds[1] + 7), (1, 1)):
        rows, scores, counts = cl.search(qs, terms, NOW, topk, candidate_limit=limit)
        for b in range(B):
            orow, osc, _ = corpus.search(qs[b], texts[b], NOW, topk, candidate_limit=limit)
            assert counts[b] == len(orow), (topk, limit, b)
            assert list(rows[b, :counts[b]]) == list(orow), (topk, limit, b)
            assert np.array_equal(scores[b, :counts[b]], osc), (topk, limit, b)
    # queries without a vector (NoOp embedder): keyword + recency only
    rows, scores, counts = cl.search(None, terms, NOW, 5, candidate_limit=n)
    for b in range(B):
        orow, osc, _ = corpus.search([], texts[b], NOW, 5, candidate_limit=n)
        assert list(rows[b, :counts[b]]) == list(orow) and np.array_equal(scores[b, :counts[b]], osc), b
    cl.close()


def test_cluster_two_stage_shards_with_ties_and_an_overflow_on_one_shard():
    """Shards large enough for the int8 two-stage pass; 20,000 identical rows sit in the second shard only, so one
    query's survivors overflow that shard's buffers while the other shard certifies at once."""
    P = pkg()
    rng = np.random.default_rng(77)
    n, dim = 420_000, 128
    emb = rng.standard_normal((n, dim)).astype(np.float32)
    emb[300_000:320_000] = emb[300_000]
    emb[100_000:100_030] = emb[100_000]
    created = np.sort(NOW - rng.integers(0, 300 * DAY, n))[::-1].astype(np.int64)
    words = np.array(["alpha", "beta", "gamma", "delta", "kubernetes", "helm", "azure", "cosmos"])
    contents = [" ".join(w) for w in words[rng.integers(0, len(words), (n, 4))]]
    corpus = orc.OracleCorpus(emb, created, contents)
    cl = P.RecallCluster([0, 0], dim)
    half = n // 2
    for g, (lo, hi) in enumerate(((0, half), (half, n))):
        for r0 in range(lo, hi, 70_000):
            r1 = min(hi, r0 + 70_000)
            cl.shard(g).append(emb[r0:r1], created[r0:r1], [s.encode() for s in contents[r0:r1]], row_ids=np.arange(r0, r1, dtype=np.int64))
    cl.seal()
    B = 12
    qs = rng.standard_normal((B, dim)).astype(np.float32)
    qs[0] = emb[300_000]                      # 20,000 exact ties in shard 1
    qs[1] = emb[100_000]                      # 30 exact ties in shard 0
    qs[2] = emb[n - 3] * 2.0
    texts = [QUERY_TEXTS[b % len(QUERY_TEXTS)] for b in range(B)]
    terms = [P.text.query_terms(t) for t in texts]
    cl.search_stats(reset=True)
    rows, scores, counts = cl.search(qs, terms, NOW, 10, candidate_limit=n)
    st = cl.search_stats()
    assert st["overflowed_queries"] >= 1 and st["passes"] >= 2 and st["requeried"] < B * (st["passes"] - 1), st   # only some queries were repeated
    for b in range(B):
        orow, osc, _ = corpus.search(qs[b], texts[b], NOW, 10, candidate_limit=n, threads=8)
        assert list(rows[b, :counts[b]]) == list(orow), b
        assert np.array_equal(scores[b, :counts[b]], osc), b
    # one query, streaming form on both shards
    rows, scores, counts = cl.search(qs[2:3], terms[2:3], NOW, 10, candidate_limit=n)
    orow, osc, _ = corpus.search(qs[2], texts[2], NOW, 10, candidate_limit=n, threads=8)
    assert list(rows[0, :counts[0]]) == list(orow) and np.array_equal(scores[0, :counts[0]], osc)
    cl.close()


def test_cluster_rejects_shards_out_of_order():
    P = pkg()
    cl = P.RecallCluster([0, 0], 4)
    e = np.ones((2, 4), dtype=np.float32)
    cl.shard(0).append(e, np.array([NOW - 5 * DAY, NOW - 6 * DAY]), [b"a", b"b"], row_ids=np.array([0, 1]))
    cl.shard(1).append(e, np.array([NOW - 1 * DAY, NOW - 9 * DAY]), [b"c", b"d"], row_ids=np.array([2, 3]))   # newer than shard 0's oldest
    with pytest.raises(P.OrrError):
        cl.seal()
    cl.close()


def test_cluster_deletes_after_the_seal_shift_the_candidate_limit_of_later_shards():
    """orr_index_delete_rows on shard 0 AFTER orr_cluster_seal: candidate_limit counts live rows only, so a limit that
    crosses the shard boundary must reach further into shard 1 -- exactly as one index (and the oracle) over the
    surviving rows does.  (Round 2 fixed each shard's dead-rows-before at seal time.)"""
    P = pkg()
    rng = np.random.default_rng(612)
    n, dim = 1800, 32
    c = _sorted_corpus(rng, n, dim, p_null=0.0)
    cl = P.RecallCluster([0, 0, 0], dim)
    bounds = [0, 600, 1200, n]
    _fill(P, cl, c, bounds)
    # deletes in shard 0 (40 rows) and shard 1 (25 rows), after the seal
    dead0 = sorted(int(x) for x in rng.choice(np.arange(0, 600), 40, replace=False))
    dead1 = sorted(int(x) for x in rng.choice(np.arange(600, 1200), 25, replace=False))
    assert cl.shard(0).delete_rows(dead0) == 40
    assert cl.shard(1).delete_rows(dead1) == 25
    gone = set(dead0) | set(dead1)
    keep = [r for r in range(n) if r not in gone]
    corpus = orc.OracleCorpus([c["emb"][r] for r in keep], c["created"][keep], [c["contents"][r] for r in keep])
    B = 6
    qs = rng.standard_normal((B, dim)).astype(np.float32)
    texts = [QUERY_TEXTS[b % len(QUERY_TEXTS)] for b in range(B)]
    terms = [P.text.query_terms(t) for t in texts]
    # limits (in LIVE rows): inside shard 0, just across the first boundary, across both, everything
    for topk, limit in ((5, 300), (8, 560 + 15), (8, 600 + 30), (10, 1135 + 20), (10, 1200 + 50), (10, n)):
        rows, scores, counts = cl.search(qs, terms, NOW, topk, candidate_limit=limit)
        for b in range(B):
            orow, osc, _ = corpus.search(qs[b], texts[b], NOW, topk, candidate_limit=limit)
            want = [keep[int(r)] for r in orow]                      # the oracle numbers the surviving rows 0..; ids are the original positions
            assert list(rows[b, :counts[b]]) == want, (topk, limit, b)
            assert np.array_equal(scores[b, :counts[b]], osc), (topk, limit, b)
    # more deletes between two searches are seen by the next one
    more = [r for r in range(0, 600) if r not in gone][:10]
    assert cl.shard(0).delete_rows(more) == 10
    gone |= set(more)
    keep = [r for r in range(n) if r not in gone]
    corpus = orc.OracleCorpus([c["emb"][r] for r in keep], c["created"][keep], [c["contents"][r] for r in keep])
    rows, scores, counts = cl.search(qs, terms, NOW, 8, candidate_limit=600)
    for b in range(B):
        orow, osc, _ = corpus.search(qs[b], texts[b], NOW, 8, candidate_limit=600)
        assert list(rows[b, :counts[b]]) == [keep[int(r)] for r in orow] and np.array_equal(scores[b, :counts[b]], osc), b
    cl.close()


def test_cluster_compaction_moves_the_later_shards_up():
    P = pkg()
    rng = np.random.default_rng(613)
    n, dim = 1500, 32
    c = _sorted_corpus(rng, n, dim, p_null=0.0)
    cl = P.RecallCluster([0, 0, 0], dim)
    _fill(P, cl, c, [0, 500, 1000, n])
    gone = set(int(x) for x in rng.choice(n, 300, replace=False))
    for g, (lo, hi) in enumerate(((0, 500), (500, 1000), (1000, n))):
        cl.shard(g).delete_rows(sorted(r for r in gone if lo <= r < hi))
    assert cl.compact() == len(gone) and cl.rows == n - len(gone)
    keep = [r for r in range(n) if r not in gone]
    corpus = orc.OracleCorpus([c["emb"][r] for r in keep], c["created"][keep], [c["contents"][r] for r in keep])
    qs = rng.standard_normal((5, dim)).astype(np.float32)
    texts = [QUERY_TEXTS[b % len(QUERY_TEXTS)] for b in range(5)]
    terms = [P.text.query_terms(t) for t in texts]
    for topk, limit in ((10, n), (6, 420), (6, 800), (3, 1)):
        rows, scores, counts = cl.search(qs, terms, NOW, topk, candidate_limit=limit)
        for b in range(5):
            orow, osc, _ = corpus.search(qs[b], texts[b], NOW, topk, candidate_limit=limit)
            assert list(rows[b, :counts[b]]) == [keep[int(r)] for r in orow] and np.array_equal(scores[b, :counts[b]], osc), (topk, limit, b)
    cl.close()


def test_cluster_record_exchange_over_rccl():
    """"exchange" = 1: the shards' records travel by ONE ncclAllGather (RCCL, bound at run time) instead of through pinned host
    memory.  A communicator needs distinct devices, so on a one-GPU box only the one-shard cluster can run it (an all-gather of
    one rank: the whole path -- dlopen, communicator, grouped call on the exchange stream, device 0's copy to the host, merge --
    is the same); two shards on one card are refused, and the pinned-host path keeps working afterwards."""
    P = pkg()
    rng = np.random.default_rng(71)
    n, dim = 1800, 32
    c = _sorted_corpus(rng, n, dim)
    corpus = orc.OracleCorpus(c["emb"], c["created"], c["contents"])
    qs = rng.standard_normal((7, dim)).astype(np.float32)
    texts = [QUERY_TEXTS[b % len(QUERY_TEXTS)] for b in range(7)]
    terms = [P.text.query_terms(t) for t in texts]
    one = P.RecallCluster([0], dim)
    _fill(P, one, c, [0, n])
    one.set_option("exchange", 1)
    for topk, limit in ((10, n), (3, 300), (70, n)):                     # (70: the large-k path; k' grows)
        rows, scores, counts = one.search(qs, terms, NOW, topk, candidate_limit=limit)
        for b in range(7):
            orow, osc, _ = corpus.search(qs[b], texts[b], NOW, topk, candidate_limit=limit)
            assert list(rows[b, :counts[b]]) == list(orow) and np.array_equal(scores[b, :counts[b]], osc), (topk, limit, b)
    st = one.search_stats()
    assert st["rccl_exchanges"] >= 3, st
    one.set_option("exchange", 0)
    before = st["rccl_exchanges"]
    one.search(qs, terms, NOW, 5, candidate_limit=n)
    assert one.search_stats()["rccl_exchanges"] == before
    one.close()
    two = P.RecallCluster([0, 0], dim)
    _fill(P, two, c, [0, 900, n])
    with pytest.raises(P.OrrError):
        two.set_option("exchange", 1)                                    # two shards on one device: no communicator
    rows, scores, counts = two.search(qs, terms, NOW, 5, candidate_limit=n)
    for b in range(7):
        orow, osc, _ = corpus.search(qs[b], texts[b], NOW, 5, candidate_limit=n)
        assert list(rows[b, :counts[b]]) == list(orow) and np.array_equal(scores[b, :counts[b]], osc), b
    two.close()
