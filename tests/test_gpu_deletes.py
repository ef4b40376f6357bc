"""Deletes without a reseal (orr_index_delete_rows): after a delete every search must
answer exactly like a shard rebuilt without those rows -- the oracle runs on the compacted corpus and its
row numbers are mapped back through the list of kept rows.  Reference behaviour being matched:
InMemoryIngestionStore.DeleteDocumentAsync (InMemoryIngestionStore.cs:50-55) followed by
GetRecentChunksAsync (:57-65), which simply no longer enumerates the document's chunks."""
import numpy as np
import pytest

from helpers import DAY, NOW, build_index, orc, pkg, random_corpus

pytestmark = pytest.mark.gpu

TEXTS = ["alpha", "the kubernetes helm", "GAMMA delta zzz", "what is the", "azure cosmos vector search"]


def _compacted(c, deleted):
    keep = np.array([r for r in range(len(c["created"])) if r not in deleted], dtype=np.int64)
    sub = {"emb": [c["emb"][r] for r in keep], "created": c["created"][keep], "contents": [c["contents"][r] for r in keep]}
    return keep, orc.OracleCorpus(sub["emb"], sub["created"], sub["contents"])


def _check(idx, keep, corpus, qvec, text, topk, limit, threads=1):
    P = pkg()
    q = None if qvec is None else np.asarray(qvec, np.float32).reshape(1, -1)
    rows, scores, counts = idx.search(q, [P.text.query_terms(text)], NOW, topk, candidate_limit=limit)
    orow, osc, _ = corpus.search([] if qvec is None else qvec, text, NOW, topk, candidate_limit=limit, threads=threads)
    k = int(counts[0])
    assert list(rows[0, :k]) == [int(keep[r]) for r in orow], (text, topk, limit)
    a, b = scores[0, :k], np.asarray(osc)
    assert ((a == b) | (np.isnan(a) & np.isnan(b))).all(), (text, topk, limit)
    return rows[0, :k]


@pytest.mark.parametrize("seed,n,dim", [(11, 400, 3), (12, 3000, 64), (13, 6000, 128)])
def test_deleted_rows_behave_like_a_rebuilt_shard(seed, n, dim):
    rng = np.random.default_rng(seed)
    c = random_corpus(rng, n, dim)
    idx = build_index(c, chunk=977)
    qvecs = [rng.standard_normal(dim).astype(np.float32), None,
             next(e for e in c["emb"] if e is not None).copy()]
    deleted = set()
    assert idx.live_rows == n
    for wave in range(4):
        if wave == 1:      # whatever ranks first for each query right now
            keep, corpus = _compacted(c, deleted)
            victims = set()
            for qv in qvecs:
                for text in TEXTS[:3]:
                    victims.update(int(r) for r in _check(idx, keep, corpus, qv, text, 10, n)[:4])
        elif wave == 2:    # the newest rows: the front of the candidate prefix
            order = np.argsort(-c["created"], kind="stable")
            victims = set(int(r) for r in order[:40:2])
        else:
            victims = set(int(r) for r in rng.choice(n, size=n // 20, replace=False))
        fresh = victims - deleted
        ids = list(victims) + list(victims)[:3] + [n + 17, -5]            # repeats and unknown ids are skipped
        assert idx.delete_rows(ids) == len(fresh)
        assert idx.delete_rows(list(victims)) == 0                         # idempotent
        deleted |= victims
        assert idx.live_rows == n - len(deleted) and idx.rows == n
        keep, corpus = _compacted(c, deleted)
        for qv in qvecs:
            for ti, text in enumerate(TEXTS):
                for topk, limit in ((10, n), (1, 300), (40, n), (len(keep) + 5, n), (64, 300), (3, 2), (10, len(keep) - 1)):
                    _check(idx, keep, corpus, qv, text, topk, limit)
    with pytest.raises(pkg().native.OrrError) as e:                        # a quarter of the shard: rebuild instead
        idx.delete_rows(list(range(n)))
    assert e.value.code == pkg().native.ORR_ESTATE
    idx.close()


def test_deletes_through_the_two_stage_pass_views_and_the_shard_file(tmp_path):
    """200,000 rows: the int8 streaming screen (1..8 queries) and the int8 screening GEMM (more) with deleted
    rows among each query's best; a view sees the deletes of its parent; save/load keeps them."""
    P = pkg()
    rng = np.random.default_rng(14)
    n, dim, B = 200_000, 128, 40
    emb = rng.standard_normal((n, dim)).astype(np.float32)
    created = np.sort(NOW - rng.integers(0, 300 * DAY, n))[::-1].astype(np.int64)
    words = np.array(["alpha", "beta", "gamma", "delta", "kubernetes", "helm", "azure", "cosmos"])
    contents = [" ".join(w) for w in words[rng.integers(0, len(words), (n, 5))]]
    idx = P.RecallIndex(dim=dim)
    for r0 in range(0, n, 50_000):
        idx.append(emb[r0:r0 + 50_000], created[r0:r0 + 50_000], [s.encode() for s in contents[r0:r0 + 50_000]])
    idx.seal()
    planted = rng.integers(0, n, B)
    qs = (emb[planted] + 0.2 * rng.standard_normal((B, dim))).astype(np.float32)
    texts = [TEXTS[b % len(TEXTS)] for b in range(B)]
    terms = [P.text.query_terms(t) for t in texts]
    before = idx.search(qs, terms, NOW, 10, candidate_limit=n)
    assert (before[0][:, 0] == planted).all()
    view = idx.view()                                                     # created BEFORE the deletes
    victims = set(int(r) for r in before[0][:, :3].ravel())               # the three best rows of every query
    victims |= set(int(r) for r in rng.choice(n, 2000, replace=False))
    victims |= set(range(0, 64))                                          # part of the sampled prefix
    assert idx.delete_rows(sorted(victims)) == len(victims)
    keep = np.array([r for r in range(n) if r not in victims], dtype=np.int64)
    corpus = orc.OracleCorpus(emb[keep], created[keep], [contents[r] for r in keep])
    idx.set_profiling(True)
    rows, scores, counts = idx.search(qs, terms, NOW, 10, candidate_limit=n)
    stats = idx.kernel_stats()
    idx.set_profiling(False)
    assert stats["screen_i8_fused"]["launches"] == 1 and "dot_exact" not in stats, stats.keys()   # no fallback
    assert not (set(int(r) for r in rows.ravel()) & victims)
    for b in (0, 1, 2, 3, 17, 39):
        orow, osc, _ = corpus.search(qs[b], texts[b], NOW, 10, candidate_limit=n, threads=8)
        assert list(rows[b, :counts[b]]) == [int(keep[r]) for r in orow], b
        assert np.array_equal(scores[b, :counts[b]], osc), b
    for handle in (idx, view):                                            # streaming form, both lanes
        for b0, nb in ((0, 1), (5, 4), (8, 8)):
            r1, s1, c1 = handle.search(qs[b0:b0 + nb], terms[b0:b0 + nb], NOW, 10, candidate_limit=n)
            assert np.array_equal(r1, rows[b0:b0 + nb]) and np.array_equal(s1, scores[b0:b0 + nb])
    # candidate_limit counts live rows: the prefix of 70,000 live rows ends behind position 70,000
    lim = 70_000
    r2, s2, c2 = idx.search(qs[:6], terms[:6], NOW, 10, candidate_limit=lim)
    for b in range(6):
        orow, osc, _ = corpus.search(qs[b], texts[b], NOW, 10, candidate_limit=lim, threads=8)
        assert list(r2[b, :c2[b]]) == [int(keep[r]) for r in orow] and np.array_equal(s2[b, :c2[b]], osc), b
    # exact kernel over all rows (two_stage off) agrees
    idx.set_option("two_stage", 0)
    r3, s3, _ = idx.search(qs[:3], terms[:3], NOW, 10, candidate_limit=n)
    assert np.array_equal(r3, rows[:3]) and np.array_equal(s3, scores[:3])
    idx.set_option("two_stage", 1)
    # the shard file keeps the deleted set
    path = str(tmp_path / "shard.orr")
    idx.save(path)
    view.close()
    idx.close()
    again = P.RecallIndex.load(path)
    assert again.rows == n and again.live_rows == n - len(victims)
    r4, s4, c4 = again.search(qs, terms, NOW, 10, candidate_limit=n)
    assert np.array_equal(r4, rows) and np.array_equal(s4, scores) and np.array_equal(c4, counts)
    assert again.delete_rows(sorted(victims)[:10]) == 0
    again.close()


def test_deletes_across_two_shards_keep_the_global_candidate_limit():
    """Two shards on one GPU: the second is told how many deleted rows lie in front of it, so that
    candidate_limit keeps counting live rows in the global order."""
    P = pkg()
    rng = np.random.default_rng(15)
    n, dim = 5000, 64
    c = random_corpus(rng, n, dim, sorted_created=True)
    c["created"] = np.sort(NOW - rng.choice(400 * DAY, n, replace=False))[::-1].astype(np.int64)   # distinct: the split is unambiguous
    half = 2200
    parts = []
    for lo, hi in ((0, half), (half, n)):
        sub = {"emb": c["emb"][lo:hi], "created": c["created"][lo:hi], "contents": c["contents"][lo:hi], "dim": dim}
        parts.append(build_index(sub, row_base=lo))
    victims = set(int(r) for r in rng.choice(n, 300, replace=False)) | set(range(half - 5, half + 5))
    d0 = parts[0].delete_rows(sorted(r for r in victims if r < half))
    d1 = parts[1].delete_rows(sorted(r for r in victims if r >= half))
    assert d0 + d1 == len(victims)
    parts[1].set_option("dead_rows_before", d0)
    keep, corpus = _compacted(c, victims)
    q = rng.standard_normal(dim).astype(np.float32)
    for text in TEXTS[:3]:
        terms = [P.text.query_terms(text)]
        for topk, limit in ((10, n), (10, 2100), (10, half - d0 + 3), (25, 3000), (5, 1)):
            kp = 32
            while True:                                   # the escalation every multi-shard caller runs
                recs = np.stack([p.search_shard(q[None, :], terms, NOW, kp, limit) for p in parts])
                rows, scores, counts, unc = P.merge_candidates(recs, dim, q[None, :], terms, NOW, topk)
                if unc == 0 or kp >= n:
                    break
                kp *= 4
            orow, osc, _ = corpus.search(q, text, NOW, topk, candidate_limit=limit)
            assert list(rows[0, :counts[0]]) == [int(keep[r]) for r in orow], (text, topk, limit)
            assert np.array_equal(scores[0, :counts[0]], osc)
    for p in parts:
        p.close()


def test_compaction_in_place_equals_a_rebuilt_shard(tmp_path):
    """orr_index_compact: the deleted rows leave the device arrays (embeddings moved up in place, scalars gathered, posting
    lists renumbered, shadows rebuilt on demand); ids are kept; every search then equals the oracle over the surviving rows,
    through the exact kernel, the streaming screen and the int8 GEMM; the quarter-of-the-shard limit is lifted; deletes,
    a second compaction and a shard-file round trip work on the compacted shard."""
    P = pkg()
    rng = np.random.default_rng(41)
    n, dim = 230_000, 128
    emb = rng.standard_normal((n, dim)).astype(np.float32)
    created = np.sort(NOW - rng.integers(0, 300 * DAY, n))[::-1].astype(np.int64)
    words = np.array(["alpha", "beta", "gamma", "delta", "kubernetes", "helm", "azure", "cosmos"])
    contents = [" ".join(w) for w in words[rng.integers(0, len(words), (n, 4))]]
    ids = np.arange(n, dtype=np.int64) * 3 + 7                              # caller's ids, not positions
    idx = P.RecallIndex(dim=dim)
    for r0 in range(0, n, 50_000):
        idx.append(emb[r0:r0 + 50_000], created[r0:r0 + 50_000], [s.encode() for s in contents[r0:r0 + 50_000]], row_ids=ids[r0:r0 + 50_000])
    idx.seal()
    B = 12
    qs = rng.standard_normal((B, dim)).astype(np.float32)
    texts = [TEXTS[b % len(TEXTS)] for b in range(B)]
    terms = [P.text.query_terms(t) for t in texts]
    idx.search(qs, terms, NOW, 10, candidate_limit=n)                        # (the int8 shadow exists before the rows move)
    deleted = set()

    def check(handle, what):
        keep = np.array([r for r in range(n) if r not in deleted], dtype=np.int64)
        corpus = orc.OracleCorpus(emb[keep], created[keep], [contents[r] for r in keep])
        for limit in (len(keep), 150_000, 300):
            rows, scores, counts = handle.search(qs, terms, NOW, 10, candidate_limit=limit)
            for b in (0, 1, 5, 11):
                orow, osc, _ = corpus.search(qs[b], texts[b], NOW, 10, candidate_limit=limit, threads=8)
                assert list(rows[b, :counts[b]]) == [int(ids[keep[r]]) for r in orow], (what, limit, b)
                assert np.array_equal(scores[b, :counts[b]], osc), (what, limit, b)
        r1, s1, c1 = handle.search(qs[:1], terms[:1], NOW, 10, candidate_limit=len(keep))       # streaming screen
        rows, scores, counts = handle.search(qs, terms, NOW, 10, candidate_limit=len(keep))
        assert np.array_equal(r1[0], rows[0]) and np.array_equal(s1[0], scores[0])
        return rows

    # a first wave: 20 % of the rows, among them whatever ranks first, the newest rows and a whole stretch
    top = check(idx, "before")
    victims = set(int((r - 7) // 3) for r in top[:, :3].ravel()) | set(range(0, 40)) | set(range(100_000, 100_700))
    victims |= set(int(r) for r in rng.choice(n, n // 5, replace=False))
    assert idx.delete_rows([int(ids[r]) for r in sorted(victims)]) == len(victims)
    deleted |= victims
    check(idx, "tombstoned")
    # a second wave would pass a quarter of the shard: refused ...
    more = set(int(r) for r in rng.choice(n, n // 10, replace=False)) - deleted
    with pytest.raises(P.OrrError):
        idx.delete_rows([int(ids[r]) for r in sorted(more)])
    # ... until the shard is compacted
    view = idx.view()
    with pytest.raises(P.OrrError):                                         # a caller's view pins the arrays
        idx.compact()
    view.close()
    assert idx.compact() == len(deleted)
    assert idx.rows == n - len(deleted) and idx.live_rows == idx.rows
    assert idx.compact() == 0
    check(idx, "compacted")
    assert idx.delete_rows([int(ids[r]) for r in sorted(more)]) == len(more)
    deleted |= more
    check(idx, "compacted + deleted")
    idx.set_option("two_stage", 0)
    check(idx, "compacted + deleted, exact kernel")
    idx.set_option("two_stage", 1)
    assert idx.compact() == len(more)
    rows_final = check(idx, "compacted twice")
    path = str(tmp_path / "compacted.orr")
    idx.save(path)
    idx.close()
    again = P.RecallIndex.load(path)
    assert again.rows == n - len(deleted)
    r2, _, _ = again.search(qs, terms, NOW, 10, candidate_limit=again.rows)
    assert np.array_equal(r2, rows_final)
    again.close()
