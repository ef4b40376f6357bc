"""The reference's durable corpus format in and out of the store mirror:
Cosmos items = CosmosDocumentRecord / CosmosChunkRecord (CosmosIngestionRecords.cs:5-30) written by
System.Text.Json under JsonNamingPolicy.CamelCase (CosmosIngestionStore.cs:34-40).  The fixture
tests/golden/cosmos_items.json is hand-written in that shape (two query pages with system properties,
escapes, a null and an empty embedding, a repeated item id, an item of another type)."""
import json
import os
import struct

import numpy as np
import pytest

from helpers import NOW, orc, pkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TICKS_PER_SECOND = 10_000_000


def _svc():
    return pkg().service


def _ticks(y, mo, d, h=0, mi=0, s=0, frac=0):
    import datetime
    days = (datetime.date(y, mo, d) - datetime.date(1, 1, 1)).days
    return ((days * 24 + h) * 60 + mi) * 60 * TICKS_PER_SECOND + s * TICKS_PER_SECOND + frac


def _f32(x):
    return struct.unpack("<f", struct.pack("<f", x))[0]


def _items(store):
    return json.loads(store.ExportCosmosJson().decode("utf-8"))


def test_fixture_imports_as_system_text_json_would_read_it():
    S = _svc()
    store = S.InMemoryIngestionStore()
    raw = open(os.path.join(ROOT, "tests", "golden", "cosmos_items.json"), "rb").read()
    assert store.ImportCosmosJson(raw) == (2, 4)
    assert store.ChunkCount() == 4
    items = _items(store)
    docs = {i["id"]: i for i in items if i["type"] == "document"}
    chunks = [i for i in items if i["type"] == "chunk"]
    assert docs["doc-azure"]["fileName"] == "azure-notes.md" and docs["doc-azure"]["createdAtUtc"] == "2026-05-14T08:30:00Z"
    assert docs["doc-k8s"]["fileName"] == 'k8s "helm" guide.txt'
    assert docs["doc-k8s"]["createdAtUtc"] == "2026-05-13T23:59:59.1234567Z" and docs["doc-k8s"]["chunkCount"] == 2
    # chunk lists are ordered by chunkIndex (InMemoryIngestionStore.cs:23), documents in first-seen order
    assert [c["id"] for c in chunks] == ["doc-azure:0000", "doc-azure:0001", "doc-k8s:0000", "doc-k8s:0001"]
    a0, a1, k0, k1 = chunks
    assert a0["content"] == "azure cosmos db vector search\nsecond line\ttabbed \\ backslash / slash"
    assert a1["content"] == "Vector search in Azure Cosmos DB — naïve café 🚀 test"
    assert a0["embedding"] == [1, 0, 0, 0, 0]
    assert a1["embedding"] == [0.25, -0.5, 1e-05, 3, 0.1]                  # shortest text of the binary32 values
    assert k0["embedding"] is None and k0["createdAtUtc"] == "2026-05-13T23:59:59.1234567Z"   # +02:00 converted to UTC
    # the repeated item id replaced the earlier item; numbers were rounded straight to binary32
    assert k1["content"] == "KUBERNETES Helm values override" and k1["createdAtUtc"] == "2026-05-13T23:59:59.12Z"
    want = [_f32(0.3), 16777216.0, -_f32(1.17549435e-38), _f32(3.4028235e38), 0.0]
    assert [_f32(v) for v in k1["embedding"]] == want
    store.close()


def test_export_import_round_trip_is_the_identity():
    S = _svc()
    rng = np.random.default_rng(5)
    store = S.InMemoryIngestionStore()
    for d in range(12):
        t = NOW - int(rng.integers(0, 10**15))
        store.UpsertDocument(S.CosmosDocumentRecord("d%02d" % d, "file %d é中\U0001F680.md" % d, t))
        cs = []
        for i in range(int(rng.integers(1, 6))):
            emb = None if rng.random() < 0.2 else (rng.standard_normal(24) * 10.0 ** rng.integers(-30, 30)).astype(np.float32)
            cs.append(S.CosmosChunkRecord("d%02d:%04d" % (d, i), "d%02d" % d, i, "text \"%d\" \\ \x01\x1f \n café" % i, emb, t + i))
        store.UpsertChunks(cs)
    first = store.ExportCosmosJson()
    json.loads(first.decode("utf-8"))                                      # well-formed
    other = S.InMemoryIngestionStore()
    nd, nc = other.ImportCosmosJson(first)
    assert (nd, nc) == (12, store.ChunkCount())
    assert other.ExportCosmosJson() == first
    # JSON lines and a bare array are read alike
    lines = "\n".join(json.dumps(i) for i in json.loads(first.decode("utf-8")))
    third = S.InMemoryIngestionStore()
    assert third.ImportCosmosJson(lines) == (nd, nc)
    assert third.ExportCosmosJson() == first
    for s in (store, other, third):
        s.close()


@pytest.mark.parametrize("bad,why", [
    ('[{"id":"a:0","type":"chunk","documentId":"a","chunkIndex":0,"content":"x","embedding":[1,2', "expected ']'"),
    ('[{"id":"a:0","type":"chunk","documentId":"a","chunkIndex":0,"embedding":["one"]}]', "not a floating-point literal"),
    ('[{"id":"a:0","type":"chunk","documentId":"a","chunkIndex":1.5}]', "integer"),
    ('[{"id":"a:0","type":"chunk","documentId":"a","createdAtUtc":"yesterday"}]', "ISO 8601"),
    ('[{"id":"a:0","type":"chunk","documentId":"a","createdAtUtc":"2026-02-30T00:00:00Z"}]', "ISO 8601"),
    ('[{"id":"a:0","type":"chunk","chunkIndex":0}]', "documentId"),
    ('[{"type":"document","fileName":"x"}]', "no id"),
    ('[{"id":"a","type":"document","fileName":"x"}] trailing', "after the closing"),
    ('{"id":"a","type":"document","fileName":"\\q"}', "escape"),
])
def test_malformed_input_is_an_argument_error_and_leaves_the_store_untouched(bad, why):
    S = _svc()
    store = S.InMemoryIngestionStore()
    store.UpsertDocument(S.CosmosDocumentRecord("keep", "keep.md", 5))
    store.UpsertChunks([S.CosmosChunkRecord("keep:0000", "keep", 0, "kept", [1.0, 2.0], 5)])
    before = store.ExportCosmosJson()
    good_then_bad = '{"id":"new","type":"document","fileName":"n"}\n' + bad if not bad.startswith("[") else bad
    with pytest.raises(S.HostError) as e:
        store.ImportCosmosJson(good_then_bad)
    assert e.value.code == -1 and why in str(e.value), str(e.value)
    assert store.ExportCosmosJson() == before
    store.close()


def test_dates_and_named_float_literals():
    S = _svc()
    store = S.InMemoryIngestionStore()
    store.ImportCosmosJson(json.dumps([
        {"id": "d", "type": "document", "fileName": "f", "createdAtUtc": "0001-01-01T00:00:00"},
        {"id": "d:0", "type": "chunk", "documentId": "d", "chunkIndex": 0, "content": "c",
         "embedding": ["NaN", "Infinity", "-Infinity", -0.0], "createdAtUtc": "2024-02-29T12:34:56.7891234567Z"},
        {"id": "d:1", "documentId": "d", "chunkIndex": 1, "content": "no type: documentId makes it a chunk",
         "createdAtUtc": "1999-12-31T23:59"},
    ]))
    items = _items(store)
    assert items[0]["createdAtUtc"] == "0001-01-01T00:00:00Z"
    assert items[1]["embedding"][:3] == ["NaN", "Infinity", "-Infinity"]
    assert b'"-Infinity",-0]' in store.ExportCosmosJson()                   # the sign of zero survives
    assert items[1]["createdAtUtc"] == "2024-02-29T12:34:56.7891234Z"      # beyond 100 ns: truncated
    assert items[2]["createdAtUtc"] == "1999-12-31T23:59:00Z" and items[2]["embedding"] is None
    store.close()


@pytest.mark.gpu
def test_search_over_an_imported_store_matches_the_oracle():
    """Import -> index -> search: the citations equal the oracle's ranking over the same items as Python reads them."""
    S = _svc()
    raw = open(os.path.join(ROOT, "tests", "golden", "cosmos_items.json"), "rb").read()
    store = S.InMemoryIngestionStore()
    store.ImportCosmosJson(raw)
    items = _items(store)
    chunks = [i for i in items if i["type"] == "chunk"]
    now = _ticks(2026, 5, 15)
    embs = [None if c["embedding"] is None else np.asarray(c["embedding"], np.float32) for c in chunks]
    import datetime

    def ticks_of(text):
        head, _, frac = text.rstrip("Z").partition(".")
        dt = datetime.datetime.strptime(head, "%Y-%m-%dT%H:%M:%S")
        return _ticks(dt.year, dt.month, dt.day, dt.hour, dt.minute, dt.second, int((frac + "0000000")[:7]) if frac else 0)

    created = [ticks_of(c["createdAtUtc"]) for c in chunks]
    cor = orc.OracleCorpus(embs, created, [c["content"] for c in chunks])
    qv = np.asarray([0.2, -0.4, 0.0, 0.5, 0.1], np.float32)
    sut = S.RecallSearchService(store, S.StubQueryEmbeddingClient(qv), candidate_limit=300, now_ticks=now)
    for text in ("azure vector search", "the kubernetes helm", "what is the"):
        body = sut.Search(text, 3)
        rows, _, rounded = cor.search(qv, text, now, 3, candidate_limit=300)
        assert [(c["chunkId"], c["score"]) for c in body["citations"]] == [(chunks[r]["id"], rd) for r, rd in zip(rows, rounded)]
    assert body["citations"][0]["fileName"] in ("azure-notes.md", 'k8s "helm" guide.txt')
    sut.close()
    store.close()
