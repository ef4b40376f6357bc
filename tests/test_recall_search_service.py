"""The reference's own recall-search tests, restated against the C++ store/service mirror
over the HIP scorer (tests/OmniRecall.Api.Tests/Services/RecallSearchServiceTests.cs:8-117,
Endpoints/RecallEndpointTests.cs:10-30), plus response-contract checks against the oracle."""
import numpy as np
import pytest

from helpers import has_gpu, orc, pkg

NOW = 639144000000000000


def _svc():
    P = pkg()
    return P.service


def seed(store, S, now=NOW):
    """SeedAsync, RecallSearchServiceTests.cs:51-117."""
    for i, (fid, fname) in enumerate((("doc-1", "notes-azure.md"), ("doc-2", "notes-devops.md"), ("doc-3", "notes-common.md"))):
        store.UpsertDocument(S.CosmosDocumentRecord(fid, fname, now))
    store.UpsertChunks([S.CosmosChunkRecord("doc-1:0000", "doc-1", 0, "azure cosmos db vector search", [1.0, 0.0], now)])
    store.UpsertChunks([S.CosmosChunkRecord("doc-2:0000", "doc-2", 0, "kubernetes deployment yaml and helm chart", [0.0, 1.0], now)])
    store.UpsertChunks([S.CosmosChunkRecord("doc-3:0000", "doc-3", 0, "what is the and of for", [0.0, 0.0], now)])


def test_blank_query_is_argument_error_without_touching_the_gpu():
    S = _svc()
    store = S.InMemoryIngestionStore()
    sut = S.RecallSearchService(store, S.StubQueryEmbeddingClient([]), now_ticks=NOW)
    for q in ("", "   ", "\t\n"):
        with pytest.raises(S.HostError) as ei:
            sut.Search(q, 3)
        assert ei.value.code == -1 and str(ei.value) == "Query is required."        # RecallSearchService.cs:22-23
    sut.close()
    store.close()


def test_store_upsert_replace_delete_counts():
    S = _svc()
    store = S.InMemoryIngestionStore()
    seed(store, S)
    assert store.ChunkCount() == 3
    store.UpsertChunks([S.CosmosChunkRecord("doc-1:0001", "doc-1", 1, "b", None, NOW),
                        S.CosmosChunkRecord("doc-1:0000", "doc-1", 0, "a", None, NOW)])     # replaces doc-1's list
    assert store.ChunkCount() == 4
    store.DeleteDocument("doc-2")
    assert store.ChunkCount() == 3
    store.close()


@pytest.mark.gpu
def test_SearchAsync_WithEmbeddings_ReturnsMostSimilarChunkFirst():
    S = _svc()
    store = S.InMemoryIngestionStore()
    seed(store, S)
    sut = S.RecallSearchService(store, S.StubQueryEmbeddingClient([1.0, 0.0]), now_ticks=NOW)
    result = sut.Search("azure", 3)
    assert result["citations"]
    assert result["citations"][0]["documentId"] == "doc-1"
    assert result["citations"][0]["fileName"] == "notes-azure.md"
    assert result["citations"][0]["score"] == 1.0            # Math.Round(0.9999999999999999, 4)
    assert [c["score"] for c in result["citations"]] == [1.0, 0.1, 0.1]
    sut.close(); store.close()


@pytest.mark.gpu
def test_SearchAsync_NoQueryEmbedding_FallsBackToKeywordScore():
    S = _svc()
    store = S.InMemoryIngestionStore()
    seed(store, S)
    sut = S.RecallSearchService(store, S.StubQueryEmbeddingClient([]), now_ticks=NOW)
    result = sut.Search("kubernetes", 3)
    assert result["citations"] and result["citations"][0]["documentId"] == "doc-2"
    assert result["citations"][0]["score"] == 0.3
    sut.close(); store.close()


@pytest.mark.gpu
def test_SearchAsync_StopWordsDoNotDiluteKeywordMatch():
    S = _svc()
    store = S.InMemoryIngestionStore()
    seed(store, S)
    sut = S.RecallSearchService(store, S.StubQueryEmbeddingClient([]), now_ticks=NOW)
    result = sut.Search("what is the kubernetes", 3)
    assert result["citations"] and result["citations"][0]["documentId"] == "doc-2"
    sut.close(); store.close()


@pytest.mark.gpu
def test_SearchRecall_AfterUpload_ReturnsCitations_contract():
    """RecallEndpointTests.cs:10-30 with the NoOp embedder, and the whole response body."""
    S = _svc()
    store = S.InMemoryIngestionStore()
    text = "nebula architecture notes for azure functions and angular app"
    created = NOW - 1234567
    store.UpsertDocument(S.CosmosDocumentRecord("d1", "nebula-notes.md", created))
    store.UpsertChunks([S.CosmosChunkRecord("d1:0000", "d1", 0, text, None, created)])
    sut = S.RecallSearchService(store, S.StubQueryEmbeddingClient([]), now_ticks=NOW)
    body = sut.Search("nebula", 3)
    assert body["query"] == "nebula" and len(body["citations"]) == 1
    c = body["citations"][0]
    assert set(c) == {"documentId", "fileName", "chunkId", "chunkIndex", "snippet", "score", "createdAtUtc"}
    assert c["fileName"] == "nebula-notes.md" and c["chunkId"] == "d1:0000" and c["chunkIndex"] == 0
    assert c["snippet"] == text
    assert c["score"] == orc.round4(0.2 + orc.recency(created, NOW) * 0.1)
    assert c["createdAtUtc"] == "2026-05-14T23:59:59.8765433Z"
    sut.close(); store.close()


@pytest.mark.gpu
def test_service_matches_oracle_on_a_multi_document_store():
    """Unknown file names, snippets cut at 180 chars, chunk-index ordering inside a document,
    store mutation between searches, candidate_limit 300 vs all."""
    S = _svc()
    rng = np.random.default_rng(4)
    store = S.InMemoryIngestionStore()
    words = ["alpha", "beta", "Gamma", "delta", "kubernetes", "azure", "the", "of"]
    chunks_flat = []
    for d in range(40):
        did = "doc-%02d" % d
        created = NOW - int(rng.integers(0, 200)) * 864000000000 - int(rng.integers(0, 10**9))
        if d % 5:
            store.UpsertDocument(S.CosmosDocumentRecord(did, "file-%02d.md" % d, created))
        cs = []
        for i in rng.permutation(10):
            content = " ".join(rng.choice(words, size=int(rng.integers(5, 60)))) + ("\nline two " * (i % 3))
            cs.append(S.CosmosChunkRecord("%s:%04d" % (did, i), did, int(i), content,
                                          rng.standard_normal(8).astype(np.float32), created))
        store.UpsertChunks(cs)
        chunks_flat += sorted(cs, key=lambda c: c.ChunkIndex)

    def expect(chunks, qv, text, k, limit):
        cor = orc.OracleCorpus([c.Embedding for c in chunks], [c.CreatedAtTicks for c in chunks], [c.Content for c in chunks])
        rows, scores, rounded = cor.search(qv, text, NOW, k, candidate_limit=limit)
        return [(chunks[r].Id, rd, orc.snippet(chunks[r].Content).decode()) for r, rd in zip(rows, rounded)]

    qv = rng.standard_normal(8).astype(np.float32)
    for limit in (300, 10**6):
        sut = S.RecallSearchService(store, S.StubQueryEmbeddingClient(qv), candidate_limit=limit, now_ticks=NOW)
        for text, k in (("alpha kubernetes", 5), ("the GAMMA", 12), ("zzz", 1)):
            body = sut.Search(text, k)
            got = [(c["chunkId"], c["score"], c["snippet"]) for c in body["citations"]]
            assert got == expect(chunks_flat, qv, text, k, limit)
            for c in body["citations"]:
                d = int(c["documentId"][4:])
                assert c["fileName"] == ("file-%02d.md" % d if d % 5 else "unknown")       # :47
        store.DeleteDocument("doc-03")
        remaining = [c for c in chunks_flat if c.DocumentId != "doc-03"]
        body = sut.Search("alpha", 7)                                                       # index rebuilt after the change
        assert [(c["chunkId"], c["score"], c["snippet"]) for c in body["citations"]] == expect(remaining, qv, "alpha", 7, limit)
        chunks_flat = remaining
        sut.close()
    store.close()


@pytest.mark.gpu
def test_chat_evidence_guard_consumer_sees_the_same_scores():
    """ChatOrchestrationService.HasSufficientEvidence (ChatOrchestrationService.cs:58-65) gates the LLM call on
    'citation count >= minimum and some Citation.Score >= 0.25' (appsettings.json:13-16), and the prompt prints
    score={c.Score:F4} (:85).  Both read the ROUNDED scores of this path, so the decision and the text must be
    what the oracle's scores give."""
    S = _svc()
    rng = np.random.default_rng(8)
    store = S.InMemoryIngestionStore()
    chunks = []
    for d in range(12):
        created = NOW - int(rng.integers(0, 90)) * 864000000000
        store.UpsertDocument(S.CosmosDocumentRecord("d%d" % d, "f%d.md" % d, created))
        cs = [S.CosmosChunkRecord("d%d:%04d" % (d, i), "d%d" % d, i, " ".join(rng.choice(["alpha", "beta", "gamma"], 6)),
                                  rng.standard_normal(16).astype(np.float32), created) for i in range(5)]
        store.UpsertChunks(cs)
        chunks += cs
    cor = orc.OracleCorpus([c.Embedding for c in chunks], [c.CreatedAtTicks for c in chunks], [c.Content for c in chunks])

    def guard(scores, min_citations=1, min_score=0.25):
        return len(scores) >= min_citations and any(s >= min_score for s in scores)

    for trial in range(12):
        qv = (rng.standard_normal(16) * (0.05 if trial % 3 == 0 else 1.0)).astype(np.float32)
        if trial % 4 == 0:
            qv = chunks[trial].Embedding                       # cosine 1: clearly above the threshold
        sut = S.RecallSearchService(store, S.StubQueryEmbeddingClient(qv), candidate_limit=300, now_ticks=NOW)
        body = sut.Search("zeta" if trial % 2 else "alpha zeta", 5)
        got = [c["score"] for c in body["citations"]]
        _, _, rounded = cor.search(qv, "zeta" if trial % 2 else "alpha zeta", NOW, 5, candidate_limit=300)
        assert got == list(rounded)
        assert guard(got) == guard(list(rounded))
        assert ["%.4f" % s for s in got] == ["%.4f" % s for s in rounded]      # score={c.Score:F4}
        sut.close()
    store.close()


@pytest.mark.gpu
def test_uploads_after_the_first_build_become_delta_shards():
    """newer documents are indexed as small delta shards in front of the existing ones
    (no full rebuild); searches over several shards stay identical to the oracle over the whole store;
    deletes and replaced chunk lists drop rows in place (orr_index_delete_rows); out-of-order timestamps
    and deleting more than a quarter of a shard fall back to a rebuild."""
    S = _svc()
    rng = np.random.default_rng(31)
    store = S.InMemoryIngestionStore()
    words = ["alpha", "beta", "gamma", "delta", "kubernetes", "azure"]
    chunks_flat = []

    def upload(doc, created, n_chunks=6):
        store.UpsertDocument(S.CosmosDocumentRecord(doc, doc + ".md", created))
        cs = [S.CosmosChunkRecord("%s:%04d" % (doc, i), doc, i, " ".join(rng.choice(words, 8)),
                                  rng.standard_normal(16).astype(np.float32), created) for i in range(n_chunks)]
        store.UpsertChunks(cs)
        chunks_flat.extend(cs)

    def replace(doc, created, n_chunks):
        at = next(i for i, c in enumerate(chunks_flat) if c.DocumentId == doc)      # the document keeps its place in the enumeration
        chunks_flat[:] = [c for c in chunks_flat if c.DocumentId != doc]
        tail = chunks_flat[at:]
        del chunks_flat[at:]
        upload(doc, created, n_chunks)
        chunks_flat.extend(tail)

    def check(sut, k=8):
        cor = orc.OracleCorpus([c.Embedding for c in chunks_flat], [c.CreatedAtTicks for c in chunks_flat],
                               [c.Content for c in chunks_flat])
        for text in ("alpha kubernetes", "the gamma", "zzz"):
            body = sut.Search(text, k)
            rows, _, rounded = cor.search(qv, text, NOW, k, candidate_limit=sut_limit)
            assert [(c["chunkId"], c["score"]) for c in body["citations"]] == \
                   [(chunks_flat[r].Id, rd) for r, rd in zip(rows, rounded)]

    qv = rng.standard_normal(16).astype(np.float32)
    base_time = NOW - 100 * 864000000000
    for d in range(30):
        upload("old-%02d" % d, base_time + d * 1000)
    for sut_limit in (300, 10**6):
        sut = S.RecallSearchService(store, S.StubQueryEmbeddingClient(qv), candidate_limit=sut_limit, now_ticks=NOW)
        check(sut)
        assert sut.Stats() == {"shards": 1, "full_rebuilds": 1, "delta_builds": 0, "tombstoned_rows": 0, "compactions": 0, "delta_merges": 0}
        t = base_time + 10**9 * (1 if sut_limit == 300 else 5)
        for step in range(3):                                  # three rounds of newer uploads -> three delta shards
            for j in range(2):
                t += 7777
                upload("new-%d-%d-%d" % (sut_limit, step, j), t)
            check(sut)
        st = sut.Stats()
        assert st["shards"] == 4 and st["full_rebuilds"] == 1 and st["delta_builds"] == 3, st
        upload("late-%d" % sut_limit, base_time - 5)            # older than what is indexed: order would break -> rebuild
        check(sut)
        assert sut.Stats()["shards"] == 1 and sut.Stats()["full_rebuilds"] == 2
        store.DeleteDocument("old-03")                          # delete -> its rows are dropped in place, no rebuild
        chunks_flat[:] = [c for c in chunks_flat if c.DocumentId != "old-03"]
        check(sut)
        st = sut.Stats()
        assert st["full_rebuilds"] == 2 and st["shards"] == 1 and st["tombstoned_rows"] == 6, st
        # reindex (DocumentIngestionService.cs:210-291): a document's chunk list is replaced by a newer one ->
        # the old rows are dropped in place and the new list becomes a delta shard
        prev = sum(1 for c in chunks_flat if c.DocumentId == "old-05")
        t += 10**7
        replace("old-05", t, 4)
        check(sut)
        st = sut.Stats()
        assert st["full_rebuilds"] == 2 and st["shards"] == 2 and st["tombstoned_rows"] == 6 + prev, st
        check(sut, k=300)                                       # more than the live rows inside candidate_limit
        gone = ["old-%02d" % d for d in range(6, 20)]           # more than a quarter of a shard at once: rebuild
        for g in gone:
            store.DeleteDocument(g)
        chunks_flat[:] = [c for c in chunks_flat if c.DocumentId not in gone]
        check(sut)
        assert sut.Stats()["full_rebuilds"] == 3 and sut.Stats()["shards"] == 1
        sut.close()
        # restore for the second pass
        upload("old-03", base_time + 3 * 1000 + 1)
        for g in gone:
            upload(g, base_time + int(g[4:]) * 1000 + 1)
    store.close()


@pytest.mark.gpu
def test_the_ninth_shard_merges_the_delta_shards_and_leaves_the_oldest_alone():
    """A corpus that grows by uploads: with eight shards in place the next upload merges the seven delta shards and the new
    chunks into ONE shard in front of the oldest (the large one), which stays on the device as it is -- no full rebuild.
    Rows a delta shard lost in place before the merge stay out; results equal the oracle over the store after every step."""
    S = _svc()
    rng = np.random.default_rng(77)
    store = S.InMemoryIngestionStore()
    words = ["alpha", "beta", "gamma", "delta", "kubernetes", "azure"]
    chunks_flat = []

    def upload(doc, created, n_chunks=5):
        store.UpsertDocument(S.CosmosDocumentRecord(doc, doc + ".md", created))
        cs = [S.CosmosChunkRecord("%s:%04d" % (doc, i), doc, i, " ".join(rng.choice(words, 8)),
                                  rng.standard_normal(16).astype(np.float32), created) for i in range(n_chunks)]
        store.UpsertChunks(cs)
        chunks_flat.extend(cs)

    qv = rng.standard_normal(16).astype(np.float32)
    base_time = NOW - 60 * 864000000000
    for d in range(40):
        upload("base-%02d" % d, base_time + d * 1000)
    sut = S.RecallSearchService(store, S.StubQueryEmbeddingClient(qv), candidate_limit=10**6, now_ticks=NOW)

    def check(note):
        cor = orc.OracleCorpus([c.Embedding for c in chunks_flat], [c.CreatedAtTicks for c in chunks_flat], [c.Content for c in chunks_flat])
        for text in ("alpha kubernetes", "the gamma", "zzz"):
            body = sut.Search(text, 12)
            rows, _, rounded = cor.search(qv, text, NOW, 12, candidate_limit=10**6)
            assert [(c["chunkId"], c["score"]) for c in body["citations"]] == [(chunks_flat[r].Id, rd) for r, rd in zip(rows, rounded)], note

    check("built")
    t = base_time + 10**9
    for step in range(7):                                      # seven rounds of newer uploads: seven delta shards
        t += 5000
        upload("up-%d-a" % step, t, 6)
        upload("up-%d-b" % step, t + 1, 1)
        check("delta %d" % step)
    st = sut.Stats()
    assert st["shards"] == 8 and st["delta_builds"] == 7 and st["full_rebuilds"] == 1 and st["delta_merges"] == 0, st
    store.DeleteDocument("up-2-b")                             # a row dropped in place inside a delta shard (a seventh of it)
    chunks_flat[:] = [c for c in chunks_flat if c.DocumentId != "up-2-b"]
    check("deleted inside a delta shard")
    t += 5000
    upload("up-7", t, 6)                                       # the ninth shard: merge
    check("merged")
    st = sut.Stats()
    assert st["shards"] == 2 and st["delta_merges"] == 1 and st["full_rebuilds"] == 1 and st["tombstoned_rows"] == 1, st
    for step in range(8, 11):                                  # and on it goes: new delta shards in front of the merged one
        t += 5000
        upload("up-%d" % step, t, 2)
        check("delta after the merge %d" % step)
    st = sut.Stats()
    assert st["shards"] == 5 and st["full_rebuilds"] == 1, st
    store.DeleteDocument("up-3-b")                             # a document that now lives in the merged shard
    chunks_flat[:] = [c for c in chunks_flat if c.DocumentId != "up-3-b"]
    check("deleted inside the merged shard")
    assert sut.Stats()["full_rebuilds"] == 1
    sut.close()
    store.close()


@pytest.mark.gpu
def test_deletes_past_a_quarter_of_a_shard_compact_it_instead_of_rebuilding():
    """Two waves of deleted documents, each a fifth of the store: the second would take the shard past a quarter of
    tombstones, so the mirror compacts it in place (orr_index_compact: the first wave's rows leave the device arrays, ids
    stay) and drops the second wave's rows -- no rebuild, results still the oracle's over the surviving chunks."""
    S = _svc()
    rng = np.random.default_rng(57)
    store = S.InMemoryIngestionStore()
    words = ["alpha", "beta", "gamma", "delta", "kubernetes", "azure"]
    chunks_flat = []
    base_time = NOW - 50 * 864000000000
    for d in range(40):
        doc = "doc-%02d" % d
        store.UpsertDocument(S.CosmosDocumentRecord(doc, doc + ".md", base_time + d * 1000))
        cs = [S.CosmosChunkRecord("%s:%04d" % (doc, i), doc, i, " ".join(rng.choice(words, 8)),
                                  rng.standard_normal(16).astype(np.float32), base_time + d * 1000) for i in range(5)]
        store.UpsertChunks(cs)
        chunks_flat.extend(cs)
    qv = rng.standard_normal(16).astype(np.float32)
    sut = S.RecallSearchService(store, S.StubQueryEmbeddingClient(qv), candidate_limit=10**6, now_ticks=NOW)

    def check(limit_note):
        cor = orc.OracleCorpus([c.Embedding for c in chunks_flat], [c.CreatedAtTicks for c in chunks_flat], [c.Content for c in chunks_flat])
        for text in ("alpha kubernetes", "the gamma", "zzz"):
            body = sut.Search(text, 9)
            rows, _, rounded = cor.search(qv, text, NOW, 9, candidate_limit=10**6)
            assert [(c["chunkId"], c["score"]) for c in body["citations"]] == [(chunks_flat[r].Id, rd) for r, rd in zip(rows, rounded)], limit_note

    check("built")
    for wave, docs in enumerate((range(0, 40, 5), range(1, 40, 5))):       # 8 documents = 20 % of the rows each
        for d in docs:
            store.DeleteDocument("doc-%02d" % d)
        gone = {"doc-%02d" % d for d in docs}
        chunks_flat[:] = [c for c in chunks_flat if c.DocumentId not in gone]
        check("wave %d" % wave)
    st = sut.Stats()
    assert st["full_rebuilds"] == 1 and st["compactions"] == 1 and st["tombstoned_rows"] == 80 and st["shards"] == 1, st
    # newer uploads still become delta shards in front of the compacted one
    doc = "doc-new"
    store.UpsertDocument(S.CosmosDocumentRecord(doc, doc + ".md", NOW - 1000))
    cs = [S.CosmosChunkRecord("%s:%04d" % (doc, i), doc, i, "alpha kubernetes " + " ".join(rng.choice(words, 4)),
                              rng.standard_normal(16).astype(np.float32), NOW - 1000) for i in range(3)]
    store.UpsertChunks(cs)
    chunks_flat.extend(cs)
    check("delta shard in front of a compacted shard")
    assert sut.Stats()["shards"] == 2 and sut.Stats()["full_rebuilds"] == 1
    sut.close()
    store.close()


def test_evidence_guard_and_prompt_score_text_consume_the_same_scores():
    """ChatOrchestrationService only consumes the 4-decimal scores -- HasSufficientEvidence
    (ChatOrchestrationService.cs:58-65, defaults MinimumCitationCount 1 / MinimumStrongCitationScore 0.25 in
    appsettings.json:13-16) and the `score={c.Score:F4}` prompt text (:85)."""
    P = pkg()
    T = P.text
    assert T.has_sufficient_evidence([0.31, 0.12], 1, 0.25) is True
    assert T.has_sufficient_evidence([0.2499, 0.12], 1, 0.25) is False
    assert T.has_sufficient_evidence([0.25], 1, 0.25) is True                  # >= threshold
    assert T.has_sufficient_evidence([], 1, 0.25) is False
    assert T.has_sufficient_evidence([0.9], 0, 0.25) is True                   # Math.Max(1, count)
    assert T.has_sufficient_evidence([0.9], 2, 0.25) is False
    assert T.has_sufficient_evidence([0.0], 1, -3.0) is True                   # Math.Max(0d, threshold)
    assert T.has_sufficient_evidence([float("nan"), 0.1], 1, 0.25) is False
    for raw, text in ((0.30000000000000004, "0.3000"), (0.9999999999999999, "1.0000"), (0.1, "0.1000"),
                      (0.12345, "0.1234"), (0.12355, "0.1236"), (0.0, "0.0000"), (12.5, "12.5000")):
        assert T.format_score_f4(T.round4(raw)) == text, raw


def _guard_cases():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chat_guard_cases.json")) as f:
        return json.load(f)


def test_evidence_guard_on_the_references_own_fixtures():
    """The three guard cases the reference's tests hold (ChatOrchestrationServiceTests.cs:97-181), as data."""
    T = pkg().text
    for case in _guard_cases()["cases"]:
        got = T.has_sufficient_evidence(case["citation_scores"], case["minimum_citation_count"], case["minimum_strong_citation_score"])
        assert got is case["sufficient"], case["source"]


@pytest.mark.gpu
def test_chat_evidence_guard_and_prompt_text_fed_by_the_hip_path():
    """SURVEY §8(f) #4 as a parity test: citations produced by the HIP scorer (orrh_service_search_json: the
    /api/recall/search body with Math.Round(score, 4)) go through HasSufficientEvidence and the `score={c.Score:F4}`
    prompt text (ChatOrchestrationService.cs:58-65,85) at every threshold pair the reference's tests and defaults use;
    the oracle's rounded scores for the same store go through an independent restatement of both.  The decisions and
    the prompt lines must be identical, for queries whose best score lies on either side of each threshold."""
    S = _svc()
    T = pkg().text
    rng = np.random.default_rng(4242)
    dim, n_docs = 16, 40
    store = S.InMemoryIngestionStore()
    words = ["alpha", "beta", "gamma", "delta", "kubernetes", "helm", "azure", "cosmos", "vector", "search", "the", "of"]
    emb, created, contents = [], [], []
    for d in range(n_docs):
        t = NOW - int(rng.integers(0, 200)) * 864000000000 - d
        store.UpsertDocument(S.CosmosDocumentRecord("doc-%02d" % d, "file-%02d.md" % d, t))
        chunks = []
        for i in range(3):
            e = None if (d + i) % 7 == 0 else (rng.standard_normal(dim) * (0.05 if d % 5 == 0 else 1.0)).astype(np.float32)
            text = " ".join(rng.choice(words, size=int(rng.integers(3, 9))))
            chunks.append(S.CosmosChunkRecord("doc-%02d:%04d" % (d, i), "doc-%02d" % d, i, text, None if e is None else e.tolist(), t))
            emb.append(e); created.append(t); contents.append(text)
        store.UpsertChunks(chunks)
    corpus = orc.OracleCorpus(emb, np.asarray(created, dtype=np.int64), contents)
    thresholds = _guard_cases()["thresholds_used_by_the_reference"]
    outcomes = set()
    queries = [("kubernetes helm", rng.standard_normal(dim)), ("zzz unknown", rng.standard_normal(dim) * 1e-3), ("azure", None),
               ("the of", rng.standard_normal(dim)), ("vector search cosmos", np.asarray(emb[4]) * 3.0), ("nothing here", None)]
    for text, vec in queries:
        v = [] if vec is None else np.asarray(vec, dtype=np.float32)
        sut = S.RecallSearchService(store, S.StubQueryEmbeddingClient(v), candidate_limit=300, now_ticks=NOW)
        for topk in (1, 2, 5):
            body = sut.Search(text, topk)
            got_scores = [c["score"] for c in body["citations"]]
            orow, osc, ornd = corpus.search(v, text, NOW, topk, candidate_limit=300)
            assert got_scores == list(ornd), (text, topk)                        # the 4-decimal scores the consumer sees
            for th in thresholds:
                cnt, thr = th["minimum_citation_count"], th["minimum_strong_citation_score"]
                want = len(ornd) >= max(1, cnt) and any(x >= max(0.0, thr) for x in ornd)        # :58-65 restated
                got = T.has_sufficient_evidence(got_scores, cnt, thr)
                assert got is want, (text, topk, th)
                outcomes.add((cnt, thr, got))
            for c, r in zip(body["citations"], ornd):                            # :85  score={c.Score:F4}
                line = "[1] file=%s chunk=%d score=%s" % (c["fileName"], c["chunkIndex"], T.format_score_f4(c["score"]))
                assert line.endswith("score=%.4f" % r), (line, r)
        sut.close()
    # both outcomes occurred at every threshold pair: the test saw scores on either side
    for th in thresholds:
        key = (th["minimum_citation_count"], th["minimum_strong_citation_score"])
        assert (key[0], key[1], True) in outcomes and (key[0], key[1], False) in outcomes, (key, sorted(outcomes))
    store.close()
