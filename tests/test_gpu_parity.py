"""Parity tests proper: the HIP path, called through the C ABI, against the CPU
oracle on the same inputs.  Bar: identical row ids in identical order, fp64 scores
bit-identical (the survivors are finished in the reference's own arithmetic)."""
import importlib

import os

import numpy as np
import pytest

from helpers import DAY, NOW, assert_same_ranking, build_index, oracle_corpus, orc, pkg, random_corpus

pytestmark = pytest.mark.gpu


def _kat_corpus(seed):
    chunks = seed["chunks"]
    dim = max((len(c["embedding"]) for c in chunks if c["embedding"]), default=0)
    return {"emb": [None if not c["embedding"] else np.asarray(c["embedding"], np.float32) for c in chunks],
            "created": np.asarray([c["createdTicks"] for c in chunks], dtype=np.int64),
            "contents": [c["content"] for c in chunks], "dim": dim}


@pytest.mark.parametrize("i", range(5))
def test_reference_known_answers_through_the_c_abi(kats, i):
    case = kats["cases"][i]
    c = _kat_corpus(case["seed"])
    idx = build_index(c)
    corpus = oracle_corpus(c)
    qv = case["queryVector"] or None
    rows, scores = assert_same_ranking(idx, corpus, c, qv, case["query"], case["topK"], 300, now=kats["nowTicks"])
    chunks = case["seed"]["chunks"]
    files = {d["id"]: d["fileName"] for d in case["seed"]["documents"]}
    a = case["asserted"]
    if "rank1DocumentId" in a:
        assert chunks[rows[0]]["documentId"] == a["rank1DocumentId"]
    if "rank1FileName" in a:
        assert files[chunks[rows[0]]["documentId"]] == a["rank1FileName"]
    assert [chunks[r]["documentId"] for r in rows] == case["derivedOrder"]
    for r, s in zip(rows, scores):
        want = case.get("derivedScores", {}).get(chunks[r]["documentId"])
        if want is not None:
            assert s == want
    idx.close()


@pytest.mark.parametrize("dim", [2, 3, 64, 768, 3072])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 700])
def test_exact_dot_and_norm_kernels(dim, n):
    """K0/K1e against the oracle's sequential fp64 sums, read back through the candidate records."""
    P = pkg()
    rng = np.random.default_rng(dim * 1000 + n)
    emb = (rng.standard_normal((n, dim)) * rng.choice([1.0, 1e-3, 50.0], size=(n, 1))).astype(np.float32)
    created = (NOW - np.arange(n, dtype=np.int64) * 1000)
    idx = P.RecallIndex(dim=dim)
    idx.append(emb, created, [b"x"] * n)
    idx.seal()
    B = 3
    qs = rng.standard_normal((B, dim)).astype(np.float32)
    kp = min(n, 64)
    rec = idx.search_shard(qs, [[b"x"]] * B, NOW, kp, n)
    for b in range(B):
        tr = rec[b, kp]
        assert tr["flags"] == P.native.ORR_CAND_TRAILER and tr["matches"] == kp and tr["order_key"] == n
        for c in rec[b, :kp]:
            r = int(c["row_id"])
            assert c["dot"] == orc.dot(qs[b], emb[r])
            assert c["norm_b"] == orc.dot(emb[r], emb[r])
            assert c["matches"] == 1 and c["created_ticks"] == created[r] and c["order_key"] == r
    idx.close()


QUERY_TEXTS = ["alpha", "the kubernetes helm", "what is the", "GAMMA delta zzz", "net ab abcdefghij", "Été naïve",
               "q? x1 db", "abcd abcde abc", "kubernetes " * 3 + "Deployment yaml chart azure cosmos vector search"]


@pytest.mark.parametrize("seed,n,dim", [(1, 50, 2), (2, 300, 3), (3, 1500, 64), (4, 4200, 128), (5, 9000, 768)])
def test_search_matches_oracle_on_mixed_corpora(seed, n, dim):
    """Unsorted appends (seal permutes), null embeddings, duplicates (exact ties),
    future timestamps, empty contents; topK and candidate_limit edge values."""
    rng = np.random.default_rng(seed)
    c = random_corpus(rng, n, dim)
    idx = build_index(c, chunk=977)
    corpus = oracle_corpus(c)
    qvecs = [rng.standard_normal(dim).astype(np.float32), None, np.zeros(dim, np.float32),
             rng.standard_normal(dim + 1).astype(np.float32)]            # last: dimension mismatch -> cosine 0
    some_row = next(r for r in range(n) if c["emb"][r] is not None)
    qvecs.append(c["emb"][some_row].copy())                               # cosine 1 on that row and its duplicates
    for qi, qv in enumerate(qvecs):
        for ti in (qi, qi + 4):
            text = QUERY_TEXTS[ti % len(QUERY_TEXTS)]
            for topk, limit in ((10, n), (1, 300), (-1, n), (0, 1), (3, 2), (40, n), (n + 5, n), (64, 300)):
                assert_same_ranking(idx, corpus, c, qv, text, topk, limit)
    idx.close()


def test_nan_and_inf_embeddings_rank_like_double_compareto():
    rng = np.random.default_rng(7)
    c = random_corpus(rng, 400, 64, p_null=0.0)
    c["emb"][5][3] = np.nan
    c["emb"][77][0] = np.inf
    c["emb"][200][:] = 0.0
    idx = build_index(c)
    corpus = oracle_corpus(c)
    q = rng.standard_normal(64).astype(np.float32)
    for topk in (10, 400, 405):
        assert_same_ranking(idx, corpus, c, q, "alpha beta", topk, 400)
    qn = q.copy()
    qn[1] = np.nan
    assert_same_ranking(idx, corpus, c, qn, "alpha", 12, 400)
    idx.close()


def test_corpus_without_embeddings_is_keyword_and_recency_only():
    """Default configuration of the reference: NoOpEmbeddingClient, cosine 0 everywhere (F7)."""
    rng = np.random.default_rng(8)
    c = random_corpus(rng, 2500, 0)
    idx = build_index(c)
    corpus = oracle_corpus(c)
    for text in QUERY_TEXTS[:5]:
        for topk, limit in ((5, 300), (10, 2500), (70, 2500)):
            assert_same_ranking(idx, corpus, c, None, text, topk, limit)
    idx.close()


def test_massive_ties_fall_back_to_candidate_order():
    """Every row identical: the top-k is the first k rows in candidate order (F4)."""
    n, dim = 5000, 64
    e = np.ones(dim, np.float32)
    c = {"emb": [e.copy() for _ in range(n)], "created": np.full(n, NOW - DAY, np.int64),
         "contents": ["alpha beta"] * n, "dim": dim}
    idx = build_index(c)
    corpus = oracle_corpus(c)
    rows, _ = assert_same_ranking(idx, corpus, c, e, "alpha", 10, n)
    assert list(rows) == list(range(10))
    assert_same_ranking(idx, corpus, c, None, "zzz", 100, n)
    idx.close()


def test_long_contents_and_many_terms():
    """Contents longer than one 1 KiB scan step, matches that straddle step boundaries,
    terms of 1..12 bytes, and a query with more than 64 terms (sliced launches)."""
    rng = np.random.default_rng(12)
    n = 300
    contents = []
    for r in range(n):
        k = int(rng.integers(100, 600))
        words = list(rng.choice(["lorem", "ipsum", "dolor", "sit", "amet", "consectetur", "adipiscing", "elit"], size=k))
        if r % 3 == 0:
            words.insert(int(rng.integers(0, k)), "needle%d" % (r % 7))
        contents.append(" ".join(words))
    c = {"emb": [None] * n, "created": (NOW - rng.integers(0, 100 * DAY, n)).astype(np.int64), "contents": contents, "dim": 0}
    idx = build_index(c)
    corpus = oracle_corpus(c)
    many = " ".join("t%03d" % i for i in range(150)) + " needle3 lorem"
    for text in ("needle3", "needle1 needle2 zzz", "m", "or", "sit amet", "consectetur adipiscing", many,
                 "lorem ipsum dolor sit amet consectetur adipiscing elit needle0 needle6"):
        assert_same_ranking(idx, corpus, c, None, text, 15, n)
    idx.close()


def test_keyword_terms_against_tokens_of_every_length():
    """The token index matches short vocabulary tokens (<= 16 bytes) one lane per token and longer ones one
    wave per token: terms of 1..40 bytes at every offset of tokens of 1..40 bytes, incl. the 4-, 8-, 12- and
    16-byte edges, multi-byte UTF-8, and terms that only almost match.  No embeddings: the score is the
    keyword fraction plus recency, so every row's match count is compared through the ranking."""
    rng = np.random.default_rng(13)
    alphabet = list("abcdefghijklmnopqrstuvwxyz0123456789-_.:/") + ["é", "ß", "中", "🚀"]
    tokens = []
    for ln in list(range(1, 41)) * 6:
        tokens.append("".join(rng.choice(alphabet, size=ln)))
    tokens += ["a" * 16, "a" * 17, "ab" * 8, "ab" * 9, "abcdefghijklmnop", "abcdefghijklmnopq", "xabcdefghijklmnop"]
    n = 900
    contents = [" ".join(rng.choice(tokens, size=int(rng.integers(1, 9)))) for _ in range(n)]
    created = (NOW - rng.integers(0, 100 * DAY, n)).astype(np.int64)
    c = {"emb": [None] * n, "created": created, "contents": contents, "dim": 0}
    idx = build_index(c)
    corpus = oracle_corpus(c)
    P = pkg()
    queries = []
    for _ in range(60):
        terms = []
        for _ in range(int(rng.integers(1, 6))):
            tok = tokens[int(rng.integers(0, len(tokens)))]
            a = int(rng.integers(0, len(tok)))
            b = int(rng.integers(a + 1, len(tok) + 1))
            term = tok[a:b]
            if rng.random() < 0.25:                          # break it somewhere: must not match by its prefix alone
                k = int(rng.integers(0, len(term)))
                term = term[:k] + "#" + term[k + 1:]
            terms.append(term)
        queries.append(" ".join(terms))
    queries += ["a" * 16, "a" * 17, "a" * 15 + "b", "abcdefghijklmnop", "bcdefghijklmnopq", "abcdefghijklmnopq", "mnop", "mnopq",
                "ab" * 8, "ba" * 8, "é", "中🚀", "-", "🚀"]
    for text in queries:
        assert_same_ranking(idx, corpus, c, None, text, n, n)
    # the same through one batch (distinct terms are matched once for all queries)
    terms = [P.text.query_terms(t) for t in queries]
    rows, scores, counts = idx.search(None, terms, NOW, 25, candidate_limit=n)
    for b, text in enumerate(queries):
        orow, osc, _ = corpus.search([], text, NOW, 25, candidate_limit=n)
        assert list(rows[b, :counts[b]]) == list(orow) and np.array_equal(scores[b, :counts[b]], osc), text
    idx.close()


def test_batched_queries_equal_single_queries():
    P = pkg()
    rng = np.random.default_rng(21)
    c = random_corpus(rng, 3000, 128, sorted_created=True)
    idx = build_index(c)
    B = 11
    qs = rng.standard_normal((B, 128)).astype(np.float32)
    texts = [QUERY_TEXTS[b % len(QUERY_TEXTS)] for b in range(B)]
    terms = [P.text.query_terms(t) for t in texts]
    rows, scores, counts = idx.search(qs, terms, NOW, 10, candidate_limit=3000)
    corpus = oracle_corpus(c)
    for b in range(B):
        orow, osc, _ = corpus.search(qs[b], texts[b], NOW, 10, candidate_limit=3000)
        assert list(rows[b]) == list(orow) and np.array_equal(scores[b], osc)
    idx.close()


def test_config_c1_1k_by_768_limit_300_and_1000():
    """BASELINE.json configs[0]: 1k chunks x 768-d, single query, candidate_limit 300 (reference) and 1000."""
    P = pkg()
    syn = importlib.import_module("omni_recall_rag_amd.synthetic")
    n, dim = 1000, 768
    emb = syn.embeddings(0, n, dim).numpy()
    created = syn.created_ticks(0, n, n).numpy()
    pool, off = syn.contents(0, n)
    idx = P.RecallIndex(dim=dim)
    idx.append(emb, created, pool.numpy(), off.numpy())
    idx.seal()
    corpus = orc.OracleCorpus(emb, created, (pool.numpy(), off.numpy()))
    for b in range(4):
        q = syn.query_vectors(b, 1, dim, n).numpy()[0]
        text = syn.query_texts(b, 1, n)[0]
        for limit in (300, 1000):
            rows, scores, counts = idx.search(q[None, :], [P.text.query_terms(text)], syn.NOW_TICKS, 10, candidate_limit=limit)
            orow, osc, _ = corpus.search(q, text, syn.NOW_TICKS, 10, candidate_limit=limit)
            assert list(rows[0]) == list(orow) and np.array_equal(scores[0], osc)
        assert rows[0, 0] == syn.planted_rows(b, 1, n)[0]
    idx.close()


def test_two_shards_on_one_gpu_equal_one_shard():
    """Row-sharded search + host merge gives the same ranking as the single index."""
    P = pkg()
    rng = np.random.default_rng(33)
    n, dim, B, k = 6000, 128, 5, 10
    c = random_corpus(rng, n, dim)
    order = np.argsort(-c["created"], kind="stable")          # shards are contiguous ranges of the candidate order
    c = {"emb": [c["emb"][i] for i in order], "created": c["created"][order],
         "contents": [c["contents"][i] for i in order], "dim": dim}
    whole = build_index(c)
    cut = 2500
    parts = []
    for lo, hi in ((0, cut), (cut, n)):
        sub = {"emb": c["emb"][lo:hi], "created": c["created"][lo:hi], "contents": c["contents"][lo:hi], "dim": dim}
        parts.append(build_index(sub, row_base=lo))
    qs = rng.standard_normal((B, dim)).astype(np.float32)
    texts = [QUERY_TEXTS[b] for b in range(B)]
    terms = [P.text.query_terms(t) for t in texts]
    for limit in (n, 300, 3000):
        rows, scores, counts = whole.search(qs, terms, NOW, k, candidate_limit=limit)
        recs = np.stack([p.search_shard(qs, terms, NOW, 32, limit) for p in parts])
        mrows, mscores, mcounts, unc = P.merge_candidates(recs, dim, qs, terms, NOW, k)
        assert unc == 0
        assert np.array_equal(rows, mrows) and np.array_equal(scores, mscores) and np.array_equal(counts, mcounts)
    whole.close()
    for p in parts:
        p.close()


@pytest.mark.parametrize("B,n,dim", [(5, 130, 64), (9, 700, 64), (33, 1000, 64), (40, 5000, 128), (130, 3000, 256), (256, 20000, 768)])
def test_batched_mfma_candidate_pass_plus_exact_rescore_matches_oracle(B, n, dim):
    """Batches >= 5 take K2 (f32 MFMA candidate pass: streaming form up to 64 queries, tiled GEMM
    above) + K6 (exact re-score): the final ranking and scores must still be bit-identical."""
    P = pkg()
    rng = np.random.default_rng(B * 7 + n)
    c = random_corpus(rng, n, dim, sorted_created=False)
    idx = build_index(c)
    corpus = oracle_corpus(c)
    qs = rng.standard_normal((B, dim)).astype(np.float32)
    some = [r for r in range(n) if c["emb"][r] is not None][:3]
    for i, r in enumerate(some):
        qs[i] = c["emb"][r]                                   # exact cosine-1 rows and their duplicates
    texts = [QUERY_TEXTS[b % len(QUERY_TEXTS)] for b in range(B)]
    terms = [P.text.query_terms(t) for t in texts]
    idx.set_profiling(True)
    rows, scores, counts = idx.search(qs, terms, NOW, 10, candidate_limit=n)
    stats = idx.kernel_stats()
    assert ("gemm_dot_bf16x3" if B > 64 else "gemv_mfma") in stats and "rescore_exact" in stats, stats.keys()
    check = range(B) if n <= 5000 else range(0, B, 16)
    for b in check:
        orow, osc, _ = corpus.search(qs[b], texts[b], NOW, 10, candidate_limit=n, threads=8)
        assert list(rows[b, :counts[b]]) == list(orow), b
        assert np.array_equal(scores[b, :counts[b]], osc), b
    idx.close()


def test_gemm_candidate_dots_are_within_the_stated_bound():
    """K2 alone: |fp32 MFMA dot - reference dot| <= (D+2) 2^-24 sum|q_k e_k| (the bound the certificate uses)."""
    P = pkg()
    rng = np.random.default_rng(99)
    n, dim, B = 300, 128, 16
    emb = rng.standard_normal((n, dim)).astype(np.float32)
    idx = P.RecallIndex(dim=dim)
    idx.append(emb, NOW - np.arange(n, dtype=np.int64), [b"x"] * n)
    idx.seal()
    qs = rng.standard_normal((B, dim)).astype(np.float32)
    rec = idx.search_shard(qs, [[]] * B, NOW, 64, n)
    for b in range(B):
        for c in rec[b, :64]:
            r = int(c["row_id"])
            assert c["flags"] & P.native.ORR_CAND_DOT_EXACT
            assert c["dot"] == orc.dot(qs[b], emb[r])          # after K6 the record holds the exact dot
        assert rec[b, 64]["dot"] > 0                            # trailer carries the pass's epsilon
    idx.close()


def test_shard_file_round_trip(tmp_path):
    """orr_index_save / orr_index_load: the reloaded shard answers exactly like the original
    (exact path, batched MFMA path, keyword-only, large k), and bad files are rejected."""
    P = pkg()
    rng = np.random.default_rng(41)
    n, dim = 2500, 128
    c = random_corpus(rng, n, dim)
    idx = build_index(c)
    path = str(tmp_path / "shard.orr")
    idx.save(path)
    re = P.RecallIndex.load(path)
    assert re.rows == n and re.dim == dim
    corpus = oracle_corpus(c)
    q = rng.standard_normal(dim).astype(np.float32)
    for topk, limit in ((10, n), (5, 300), (70, n)):
        assert_same_ranking(re, corpus, c, q, "alpha kubernetes the", topk, limit)
        assert_same_ranking(re, corpus, c, None, "GAMMA delta", topk, limit)
    B = 20
    qs = rng.standard_normal((B, dim)).astype(np.float32)
    terms = [P.text.query_terms("helm azure")] * B
    r1 = idx.search(qs, terms, NOW, 10, candidate_limit=n)
    r2 = re.search(qs, terms, NOW, 10, candidate_limit=n)
    assert all(np.array_equal(a, b) for a, b in zip(r1, r2))
    bad = tmp_path / "bad.orr"
    bad.write_bytes(b"not a shard file at all" * 10)
    with pytest.raises(P.OrrError) as ei:
        P.RecallIndex.load(str(bad))
    assert ei.value.code == P.native.ORR_EINVAL
    trunc = tmp_path / "trunc.orr"
    trunc.write_bytes(open(path, "rb").read()[:4096])
    with pytest.raises(P.OrrError):
        P.RecallIndex.load(str(trunc))
    # a file is input: counts that do not add up to its size, and indices that point outside what they index, are refused
    # (header: magic[8], version u32, dim u32, n_rows i64, n_tokens i64, n_postings u64, vpool_bytes u64, reserved u64[4])
    good = bytearray(open(path, "rb").read())
    import struct
    n_rows_f, n_tokens_f, n_post_f, vpool_f = struct.unpack_from("<qqQQ", good, 16)
    assert n_rows_f == n and n_post_f > 0 and n_tokens_f > 0
    inflated = bytearray(good)
    struct.pack_into("<Q", inflated, 32, n_post_f + (1 << 40))              # a terabyte of postings that the file does not hold
    (tmp_path / "inflated.orr").write_bytes(bytes(inflated))
    with pytest.raises(P.OrrError) as ei:
        P.RecallIndex.load(str(tmp_path / "inflated.orr"))
    assert ei.value.code == P.native.ORR_EINVAL and "add up" in str(ei.value)
    wild = bytearray(good)                                                   # the posting rows are the file's last array
    struct.pack_into("<I", wild, len(wild) - 4, n + 5)                      # (no deleted rows in this shard)
    (tmp_path / "wild.orr").write_bytes(bytes(wild))
    with pytest.raises(P.OrrError) as ei:
        P.RecallIndex.load(str(tmp_path / "wild.orr"))
    assert ei.value.code == P.native.ORR_EINVAL and "posting rows" in str(ei.value)
    idx.close()
    re.close()


def test_fused_batched_pass_matches_oracle():
    """> 96 queries over >= 48 selection segments: the bf16x3 GEMM scores and filters in its epilogue
    (scores never reach HBM behind the sampled prefix).  Same bit-exact results required."""
    P = pkg()
    rng = np.random.default_rng(77)
    n, dim, B = 200_000, 64, 130
    emb = rng.standard_normal((n, dim)).astype(np.float32)
    created = np.sort(NOW - rng.integers(0, 300 * DAY, n))[::-1].astype(np.int64)
    words = np.array(["alpha", "beta", "gamma", "delta", "kubernetes", "helm", "azure", "cosmos"])
    contents = [" ".join(w) for w in words[rng.integers(0, len(words), (n, 5))]]
    idx = P.RecallIndex(dim=dim)
    step = 50_000
    for r0 in range(0, n, step):
        idx.append(emb[r0:r0 + step], created[r0:r0 + step], [c.encode() for c in contents[r0:r0 + step]])
    idx.seal()
    qs = rng.standard_normal((B, dim)).astype(np.float32)
    qs[0] = emb[n - 7]                       # winners deep behind the prefix
    qs[1] = emb[123_456] * 3.0
    texts = [QUERY_TEXTS[b % len(QUERY_TEXTS)] for b in range(B)]
    terms = [P.text.query_terms(t) for t in texts]
    idx.set_option("two_stage", 0)
    plain = idx.search(qs, terms, NOW, 10, candidate_limit=n)          # split pass, dots through HBM
    idx.set_option("fuse_epilogue", 1)
    with pytest.raises(P.OrrError):
        idx.set_option("no_such_option", 1)
    idx.set_profiling(True)
    rows, scores, counts = idx.search(qs, terms, NOW, 10, candidate_limit=n)
    stats = idx.kernel_stats()
    assert "gemm_dot_bf16x3_fused" in stats and "buffer_to_lists" in stats, stats.keys()
    assert all(np.array_equal(x, y) for x, y in zip(plain, (rows, scores, counts)))
    corpus = orc.OracleCorpus(emb, created, contents)
    assert rows[0, 0] == n - 7 and rows[1, 0] == 123_456
    for b in list(range(0, 6)) + [64, 129]:
        orow, osc, _ = corpus.search(qs[b], texts[b], NOW, 10, candidate_limit=n, threads=8)
        assert list(rows[b, :counts[b]]) == list(orow), b
        assert np.array_equal(scores[b, :counts[b]], osc), b
    idx.close()


def test_screening_dots_stay_inside_the_bound_the_two_stage_pass_uses():
    """orr_index_screen_dots: sum_k bf16(q_k) bf16(e_k) from the bf16 shadow, 256 x 256 x 64 LDS-DMA tiles.
    Checked against a float64 evaluation of the same bf16-rounded operands (tight: only fp32
    accumulation differs) and against the exact dot with the bound the pass relies on."""
    import torch
    P = pkg()
    rng = np.random.default_rng(91)
    for n, dim, B in ((777, 64, 3), (1000, 192, 70), (5000, 3072, 300)):
        emb = (rng.standard_normal((n, dim)) * rng.choice([1e-3, 1.0, 50.0], (n, 1))).astype(np.float32)
        created = np.sort(NOW - rng.integers(0, 300 * DAY, n))[::-1].astype(np.int64)
        idx = P.RecallIndex(dim=dim)
        idx.append(emb, created, [b"x"] * n)
        idx.seal()
        qs = rng.standard_normal((B, dim)).astype(np.float32)
        qs[0] = emb[n - 1]
        got = idx.screen_dots(qs).astype(np.float64)
        eh = torch.from_numpy(emb).to(torch.bfloat16).to(torch.float64).numpy()
        qh = torch.from_numpy(qs).to(torch.bfloat16).to(torch.float64).numpy()
        want_bf16 = qh @ eh.T
        scale = np.abs(qh) @ np.abs(eh).T                                      # sum |q_k e_k| of the rounded operands
        assert np.all(np.abs(got - want_bf16) <= 1.02 * dim * 2.0 ** -23 * scale + 1e-30), (n, dim, B)
        exact = qs.astype(np.float64) @ emb.astype(np.float64).T
        bound = (2.0 ** -7 * (1 + 2.0 ** -9) + 1.02 * dim * 2.0 ** -23) * (np.abs(qs).astype(np.float64) @ np.abs(emb).astype(np.float64).T)
        assert np.all(np.abs(got - exact) <= bound + 1e-30), (n, dim, B)
        idx.close()


def test_two_stage_batched_pass_matches_oracle():
    """Option "two_stage": the split GEMM only covers a sampled prefix; the corpus is screened by ONE
    plain-bf16 product whose survivors are re-scored in the reference's arithmetic on the device.
    Results must stay bit-identical, with and without a retry through the unfused pass."""
    P = pkg()
    rng = np.random.default_rng(78)
    n, dim, B = 200_000, 128, 130
    emb = rng.standard_normal((n, dim)).astype(np.float32)
    emb[150_000:150_040] = emb[150_000]          # 40 identical rows: ties across the floor
    emb[100_000:120_000] = emb[100_000]          # 20,000 identical rows: more survivors than a query's buffer holds
    created = np.sort(NOW - rng.integers(0, 300 * DAY, n))[::-1].astype(np.int64)
    words = np.array(["alpha", "beta", "gamma", "delta", "kubernetes", "helm", "azure", "cosmos"])
    contents = [" ".join(w) for w in words[rng.integers(0, len(words), (n, 5))]]
    idx = P.RecallIndex(dim=dim)
    step = 50_000
    for r0 in range(0, n, step):
        idx.append(emb[r0:r0 + step], created[r0:r0 + step], [c.encode() for c in contents[r0:r0 + step]])
    idx.seal()
    qs = rng.standard_normal((B, dim)).astype(np.float32)
    qs[0] = emb[n - 7]
    qs[1] = emb[123_456] * 3.0
    qs[2] = emb[150_000]
    qs[3] = 0.0                                  # normA == 0: cosine 0 everywhere, keyword + recency decide
    qs[4] = emb[5] * 1e-3
    texts = [QUERY_TEXTS[b % len(QUERY_TEXTS)] for b in range(B)]
    terms = [P.text.query_terms(t) for t in texts]
    corpus = orc.OracleCorpus(emb, created, contents)
    # a query on the 20,000 identical rows overflows its survivor buffer: THAT QUERY ALONE goes through further passes
    # (larger buffers sized from the measured count first), the other 129 keep the results of the first pass
    q_over = qs.copy()
    q_over[5] = emb[100_000]
    idx.set_profiling(True)
    idx.reset_search_stats()
    rows, scores, counts = idx.search(q_over, terms, NOW, 10, candidate_limit=n)
    st = idx.kernel_stats()
    ss = idx.search_stats()
    assert st["screen_i8_prefix"]["launches"] == 1 and st["screen_i8_fused"]["launches"] == 1 and "gemm_dot_bf16x3" not in st, st
    assert ss["overflowed_queries"] >= 1 and ss["buffer_growths"] >= 1 and ss["survivors_max"] >= 20_000, ss
    assert ss["passes"] >= 2 and ss["requeried"] == ss["passes"] - 1, ss        # every repeat carried exactly one query
    assert ss["survivor_capacity"] >= 20_000, ss
    idx.set_profiling(False)
    for b in (0, 5, 6):
        orow, osc, _ = corpus.search(q_over[b], texts[b], NOW, 10, candidate_limit=n, threads=8)
        assert list(rows[b, :counts[b]]) == list(orow) and np.array_equal(scores[b, :counts[b]], osc), b
    for topk, mode in ((10, 1), (30, 1), (10, 2), (30, 2)):
        idx.set_option("two_stage", 0)
        plain = idx.search(qs, terms, NOW, topk, candidate_limit=n)
        idx.set_option("two_stage", mode)         # 1: bf16 shadow + LDS-DMA screening kernel; 2: converts in the kernel
        idx.set_profiling(True)
        rows, scores, counts = idx.search(qs, terms, NOW, topk, candidate_limit=n)
        stats = idx.kernel_stats()
        idx.set_profiling(False)
        screen = "screen_i8_fused" if mode == 1 else "gemm_dot_bf16x1_fused"      # dim 128: mode 1 screens on the int8 shadow
        assert screen in stats and "rescore_buffer_exact" in stats, stats.keys()
        if mode == 1:     # the sampled prefix goes through the int8 GEMM too; no retry through the unfused pass
            assert stats["screen_i8_prefix"]["launches"] == 1 and stats[screen]["launches"] == 1 and "gemm_dot_bf16x3" not in stats, stats
        else:
            assert stats["gemm_dot_bf16x3"]["launches"] == 1, stats
        assert all(np.array_equal(x, y) for x, y in zip(plain, (rows, scores, counts)))
        assert rows[0, 0] == n - 7 and rows[1, 0] == 123_456
        for b in list(range(0, 8)) + [64, 129]:
            orow, osc, _ = corpus.search(qs[b], texts[b], NOW, topk, candidate_limit=n, threads=8)
            assert list(rows[b, :counts[b]]) == list(orow), (topk, b)
            assert np.array_equal(scores[b, :counts[b]], osc), (topk, b)
    # 1..4 queries: the int8 shadow is streamed without the matrix core (the sample first, for the floor);
    # 5..8 queries take the screening GEMM with one live query tile
    idx.set_option("two_stage", 1)
    for b0, nb in ((0, 1), (2, 1), (3, 1), (0, 4), (4, 3), (100, 2), (0, 8), (1, 6)):
        idx.set_profiling(True)
        rows, scores, counts = idx.search(qs[b0:b0 + nb], terms[b0:b0 + nb], NOW, 10, candidate_limit=n)
        st = idx.kernel_stats()
        assert ("screen_gemv_i8" if nb <= 4 else "screen_i8_fused") in st, sorted(st)     # dim 128: the int8 shadow applies
        idx.set_profiling(False)
        for b in range(nb):
            orow, osc, _ = corpus.search(qs[b0 + b], texts[b0 + b], NOW, 10, candidate_limit=n, threads=8)
            assert list(rows[b, :counts[b]]) == list(orow) and np.array_equal(scores[b, :counts[b]], osc), (b0, b)
    rows, scores, counts = idx.search(q_over[5:6], terms[5:6], NOW, 10, candidate_limit=n)     # buffer overflow -> exact pass
    orow, osc, _ = corpus.search(q_over[5], texts[5], NOW, 10, candidate_limit=n, threads=8)
    assert list(rows[0, :counts[0]]) == list(orow) and np.array_equal(scores[0, :counts[0]], osc)
    # the sharded entry point takes the same route (floor from the k'-th best) and merges exactly
    kprime = 32
    recs = idx.search_shard(qs, terms, NOW, kprime, candidate_limit=n)
    mrows, mscores, mcounts, unc = P.merge_candidates(recs[None], dim, qs, terms, NOW, 10)
    assert unc == 0
    idx.set_option("two_stage", 0)
    plain = idx.search(qs, terms, NOW, 10, candidate_limit=n)
    assert np.array_equal(mrows, plain[0]) and np.array_equal(mscores, plain[1])
    # ... and so does a small sharded batch (the multi-GPU bench sends one query per rank): streaming form
    idx.set_option("two_stage", 1)
    for nb in (1, 3, 8):
        recs = idx.search_shard(qs[:nb], terms[:nb], NOW, kprime, candidate_limit=n)
        mrows, mscores, mcounts, unc = P.merge_candidates(recs[None], dim, qs[:nb], terms[:nb], NOW, 10)
        assert unc == 0
        assert np.array_equal(mrows, plain[0][:nb]) and np.array_equal(mscores, plain[1][:nb]), nb
    idx.close()


def test_two_stage_pass_with_rows_that_are_hot_for_every_query():
    """The 1,500 newest rows are near-duplicates of one chunk and carry the full recency credit, and no query
    has terms: for a query they score within the screening margin of each other, so whole tiles of the
    screening GEMM pass the pre-filter for every query that likes that chunk at all.  The epilogue must cope
    (per-thread queues overflow into direct appends) without falling back, and the result stays exact."""
    P = pkg()
    rng = np.random.default_rng(79)
    n, dim, B = 200_000, 64, 130
    emb = (rng.standard_normal((n, dim)) * 0.2).astype(np.float32)
    emb[:1500] = emb[0] + (rng.standard_normal((1500, dim)) * 1e-4).astype(np.float32)   # near-duplicates of one chunk
    created = np.empty(n, dtype=np.int64)
    created[:1500] = NOW - rng.integers(0, DAY // 4, 1500)             # a few hours old
    created[1500:] = NOW - rng.integers(200 * DAY, 300 * DAY, n - 1500)
    created = np.sort(created)[::-1].astype(np.int64)
    contents = [b"x"] * n
    idx = P.RecallIndex(dim=dim)
    for r0 in range(0, n, 50_000):
        idx.append(emb[r0:r0 + 50_000], created[r0:r0 + 50_000], contents[r0:r0 + 50_000])
    idx.seal()
    qs = (emb[0][None, :] * rng.uniform(0.5, 2.0, (B, 1)) + rng.standard_normal((B, dim)) * 0.15).astype(np.float32)   # all like that chunk
    terms = [[] for _ in range(B)]
    idx.set_profiling(True)
    rows, scores, counts = idx.search(qs, terms, NOW, 10, candidate_limit=n)
    stats = idx.kernel_stats()
    idx.set_profiling(False)
    assert stats["screen_bf16_fused"]["launches"] == 1 and stats["gemm_dot_bf16x3"]["launches"] == 1, stats
    assert (rows < 1500).all()                                             # every result is one of the hot rows
    corpus = orc.OracleCorpus(emb, created, contents)
    for b in (0, 1, 64, 129):
        orow, osc, _ = corpus.search(qs[b], "", NOW, 10, candidate_limit=n, threads=8)
        assert list(rows[b, :counts[b]]) == list(orow), b
        assert np.array_equal(scores[b, :counts[b]], osc), b
    r1, s1, c1 = idx.search(qs[:1], terms[:1], NOW, 10, candidate_limit=n)    # streaming form, same rows
    assert np.array_equal(r1[0], rows[0]) and np.array_equal(s1[0], scores[0])
    idx.close()


def test_int8_shadow_with_rows_and_queries_that_quantise_badly():
    """The int8 screens (stream for 1..8 queries, MFMA GEMM beyond) rely on a per-pair bound computed from the
    actual quantisation errors.  Rows and queries chosen to make one scale per vector a poor fit -- a single
    huge coordinate, tiny and huge magnitudes, zero rows, a constant vector -- must still give the reference's
    answer bit for bit (such rows simply survive the screen)."""
    P = pkg()
    rng = np.random.default_rng(80)
    n, dim = 200_000, 128
    emb = rng.standard_normal((n, dim)).astype(np.float32)
    emb[0:100] *= np.float32(1e-6)
    emb[100:200, 7] = np.float32(1000.0)                                  # one coordinate dominates the scale
    emb[200:300] = 0.0
    emb[300:400] *= np.float32(1e30)
    emb[400:500] = np.float32(0.37)                                       # constant rows
    emb[500:600, ::2] = 0.0
    perm = rng.permutation(n)
    emb = emb[perm]                                                       # spread the odd rows over the corpus
    where = np.empty(n, dtype=np.int64)
    where[perm] = np.arange(n)
    created = np.sort(NOW - rng.integers(0, 300 * DAY, n))[::-1].astype(np.int64)
    words = np.array(["alpha", "beta", "gamma", "delta"])
    contents = [" ".join(w).encode() for w in words[rng.integers(0, len(words), (n, 3))]]
    idx = P.RecallIndex(dim=dim)
    for r0 in range(0, n, 50_000):
        idx.append(emb[r0:r0 + 50_000], created[r0:r0 + 50_000], contents[r0:r0 + 50_000])
    idx.seal()
    B = 40
    qs = rng.standard_normal((B, dim)).astype(np.float32)
    qs[0] = 0.0; qs[0, 7] = 100.0                                         # spiky query: likes the spiky rows
    qs[1] *= np.float32(1e-8)
    qs[2] = emb[where[150]]
    qs[3] = emb[where[350]]                                               # huge magnitude
    qs[4] = 0.37
    qs[5] = emb[where[50]]                                                # tiny magnitude
    qs[6, 1::2] = 0.0
    texts = ["alpha", "", "beta gamma", "delta", "", "alpha beta", ""] + [""] * (B - 7)
    terms = [P.text.query_terms(t) if t else [] for t in texts]
    corpus = orc.OracleCorpus(emb, created, contents)
    idx.set_profiling(True)
    r4, s4, c4 = idx.search(qs[:4], terms[:4], NOW, 10, candidate_limit=n)            # int8 stream
    r7, s7, c7 = idx.search(qs[:7], terms[:7], NOW, 10, candidate_limit=n)            # int8 GEMM, one live query tile
    rB, sB, cB = idx.search(qs, terms, NOW, 10, candidate_limit=n)                    # int8 GEMM
    stats = idx.kernel_stats()
    idx.set_profiling(False)
    assert "screen_gemv_i8" in stats and "screen_i8_fused" in stats, sorted(stats)
    for b in range(8):
        orow, osc, _ = corpus.search(qs[b], texts[b], NOW, 10, candidate_limit=n, threads=8)
        for name, (rr, ss, cc) in (("stream", (r4, s4, c4)), ("stream2", (r7, s7, c7)), ("gemm", (rB, sB, cB))):
            if b >= len(cc):
                continue
            assert list(rr[b, :cc[b]]) == list(orow), (name, b)
            assert np.array_equal(ss[b, :cc[b]], osc), (name, b)
    idx.close()


def test_two_stage_pass_at_a_dimension_that_is_a_multiple_of_256():
    """dim % 256 == 0 lets small batches re-score their survivors four lanes per row (the sum's order stays
    the reference's; only the loads are spread) inside the one-launch tail of the pass (re-score, lists, final
    selection, records); checked against the oracle directly."""
    P = pkg()
    rng = np.random.default_rng(81)
    n, dim = 200_000, 256
    emb = rng.standard_normal((n, dim)).astype(np.float32)
    emb[:64] *= np.float32(1e3)
    created = np.sort(NOW - rng.integers(0, 300 * DAY, n))[::-1].astype(np.int64)
    words = np.array(["alpha", "beta", "gamma", "delta", "kubernetes", "helm"])
    contents = [" ".join(w) for w in words[rng.integers(0, len(words), (n, 4))]]
    idx = P.RecallIndex(dim=dim)
    for r0 in range(0, n, 50_000):
        idx.append(emb[r0:r0 + 50_000], created[r0:r0 + 50_000], [c.encode() for c in contents[r0:r0 + 50_000]])
    idx.seal()
    B = 70
    qs = rng.standard_normal((B, dim)).astype(np.float32)
    qs[0] = emb[n - 3]
    qs[1] = emb[10] * np.float32(0.5)
    texts = [QUERY_TEXTS[b % len(QUERY_TEXTS)] for b in range(B)]
    terms = [P.text.query_terms(t) for t in texts]
    corpus = orc.OracleCorpus(emb, created, contents)
    for b0, nb in ((0, 1), (0, 4), (1, 7), (0, 17), (0, 64), (0, 70)):
        idx.set_profiling(True)
        rows, scores, counts = idx.search(qs[b0:b0 + nb], terms[b0:b0 + nb], NOW, 10, candidate_limit=n)
        st = idx.kernel_stats()
        idx.set_profiling(False)
        # dim % 256 == 0: the tail of the pass is finish_survivors (one launch; two from 64 queries on), else the separate kernels
        assert "finish_survivors" in st and "rescore_buffer_exact" not in st and "dot_exact" not in st, sorted(st)
        for b in sorted({0, min(1, nb - 1), nb - 1}):
            orow, osc, _ = corpus.search(qs[b0 + b], texts[b0 + b], NOW, 10, candidate_limit=n, threads=8)
            assert list(rows[b, :counts[b]]) == list(orow) and np.array_equal(scores[b, :counts[b]], osc), (b0, nb, b)
    # more than one 256-query tile: the persistent workgroups of the screening GEMM walk (query tile, row tile) pairs
    B2 = 300
    q2 = rng.standard_normal((B2, dim)).astype(np.float32)
    q2[255] = emb[77_777]
    q2[256] = emb[n - 1]
    q2[299] = emb[123_456] * np.float32(2.0)
    t2 = [QUERY_TEXTS[b % len(QUERY_TEXTS)] for b in range(B2)]
    terms2 = [P.text.query_terms(t) for t in t2]
    idx.set_option("two_stage", 0)
    plain = idx.search(q2, terms2, NOW, 10, candidate_limit=n)
    idx.set_option("two_stage", 1)
    idx.set_profiling(True)
    got = idx.search(q2, terms2, NOW, 10, candidate_limit=n)
    st = idx.kernel_stats()
    idx.set_profiling(False)
    assert st["screen_i8_fused"]["launches"] == 1 and "gemm_dot_bf16x3" not in st and "finish_survivors" in st, sorted(st)
    assert all(np.array_equal(x, y) for x, y in zip(plain, got))
    assert got[0][255, 0] == 77_777 and got[0][256, 0] == n - 1 and got[0][299, 0] == 123_456
    for b in (0, 255, 256, 299):
        orow, osc, _ = corpus.search(q2[b], t2[b], NOW, 10, candidate_limit=n, threads=8)
        assert list(got[0][b, :got[2][b]]) == list(orow) and np.array_equal(got[1][b, :got[2][b]], osc), b
    import torch
    qd = torch.from_numpy(qs).to("cuda:0")                                 # device-resident batch: norms computed on the device
    for nb in (70, 20, 3):
        h = idx.search(qs[:nb], terms[:nb], NOW, 10, candidate_limit=n)
        d = idx.search(qd[:nb], terms[:nb], NOW, 10, candidate_limit=n)
        assert all(np.array_equal(x, y) for x, y in zip(h, d)), nb
    idx.close()


@pytest.mark.parametrize("n,dim", [(3000, 3), (5000, 64), (4000, 256)])
def test_device_resident_query_batches_get_the_same_norms_as_host_ones(n, dim):
    """Batches of 16+ queries that already live on the device have their exact norms computed there (the kernel
    that computes the rows' norms); host-resident queries get them on the host.  Same queries, both ways, plus
    the oracle: zero, tiny, huge, NaN and duplicate queries included."""
    import torch
    P = pkg()
    rng = np.random.default_rng(82 + dim)
    c = random_corpus(rng, n, dim)
    idx = build_index(c)
    corpus = oracle_corpus(c)
    B = 40
    qs = rng.standard_normal((B, dim)).astype(np.float32)
    qs[1] = 0.0
    qs[2] *= np.float32(1e-20)
    qs[3] *= np.float32(1e18)
    qs[4, 0] = np.nan
    qs[5] = qs[6]
    qs[7] = next(e for e in c["emb"] if e is not None)
    texts = [QUERY_TEXTS[b % len(QUERY_TEXTS)] for b in range(B)]
    terms = [P.text.query_terms(t) for t in texts]
    host = idx.search(qs, terms, NOW, 10, candidate_limit=n)
    qd = torch.from_numpy(qs).to("cuda:0")
    for nb in (B, 16, 15):
        dev = idx.search(qd[:nb], terms[:nb], NOW, 10, candidate_limit=n)
        for x, y in zip(host, dev):
            assert np.array_equal(x[:nb], y, equal_nan=True), nb
    for b in range(9):
        orow, osc, _ = corpus.search(qs[b], texts[b], NOW, 10, candidate_limit=n)
        assert list(host[0][b, :host[2][b]]) == list(orow), b
        a = host[1][b, :host[2][b]]
        assert ((a == osc) | (np.isnan(a) & np.isnan(osc))).all(), b
    idx.close()


def test_two_stage_pass_on_the_bf16_shadow_at_dim_192():
    """dim % 128 != 0: no int8 shadow, the two-stage pass runs on the bf16 one -- streaming screen for 1..8 queries,
    the bf16 screening GEMM (persistent workgroups: 782 row tiles, 6 K-tiles each) beyond, split-bf16 prefix."""
    P = pkg()
    rng = np.random.default_rng(83)
    n, dim = 200_000, 192
    emb = rng.standard_normal((n, dim)).astype(np.float32)
    created = np.sort(NOW - rng.integers(0, 300 * DAY, n))[::-1].astype(np.int64)
    words = np.array(["alpha", "beta", "gamma", "delta", "kubernetes", "helm"])
    contents = [" ".join(w) for w in words[rng.integers(0, len(words), (n, 4))]]
    idx = P.RecallIndex(dim=dim)
    for r0 in range(0, n, 50_000):
        idx.append(emb[r0:r0 + 50_000], created[r0:r0 + 50_000], [c.encode() for c in contents[r0:r0 + 50_000]])
    idx.seal()
    B = 300
    qs = rng.standard_normal((B, dim)).astype(np.float32)
    qs[0] = emb[n - 5]
    qs[256] = emb[31_337]
    texts = [QUERY_TEXTS[b % len(QUERY_TEXTS)] for b in range(B)]
    terms = [P.text.query_terms(t) for t in texts]
    corpus = orc.OracleCorpus(emb, created, contents)
    idx.set_option("two_stage", 0)
    plain = idx.search(qs, terms, NOW, 10, candidate_limit=n)
    idx.set_option("two_stage", 1)
    for b0, nb, kernel in ((0, 1, "screen_gemv_bf16"), (3, 8, "screen_gemv_bf16"), (0, 20, "screen_bf16_fused"),
                           (0, 100, "screen_bf16_fused"), (0, 300, "screen_bf16_fused")):
        idx.set_profiling(True)
        got = idx.search(qs[b0:b0 + nb], terms[b0:b0 + nb], NOW, 10, candidate_limit=n)
        st = idx.kernel_stats()
        idx.set_profiling(False)
        assert kernel in st and "screen_i8_fused" not in st and "screen_gemv_i8" not in st, sorted(st)
        assert all(np.array_equal(x[b0:b0 + nb], y) for x, y in zip(plain, got)), (b0, nb)
    assert plain[0][0, 0] == n - 5 and plain[0][256, 0] == 31_337
    for b in (0, 1, 256, 299):
        orow, osc, _ = corpus.search(qs[b], texts[b], NOW, 10, candidate_limit=n, threads=8)
        assert list(plain[0][b, :plain[2][b]]) == list(orow) and np.array_equal(plain[1][b, :plain[2][b]], osc), b
    idx.close()


def test_empty_and_degenerate_inputs():
    """Empty corpus, rows with empty content, queries without terms or vectors, API misuse."""
    P = pkg()
    empty = P.RecallIndex(dim=8)
    empty.seal()
    rows, scores, counts = empty.search(np.ones((2, 8), np.float32), [[b"x"], []], NOW, 5, candidate_limit=300)
    assert list(counts) == [0, 0] and (rows == -1).all()
    with pytest.raises(P.OrrError) as ei:
        empty.append(np.ones((1, 8), np.float32), [NOW], [b"late"])          # append after seal
    assert ei.value.code == P.native.ORR_ESTATE
    empty.close()

    unsealed = P.RecallIndex(dim=4)
    unsealed.append(np.ones((2, 4), np.float32), [NOW, NOW], [b"a", b"b"])
    with pytest.raises(P.OrrError) as ei:
        unsealed.search(np.ones((1, 4), np.float32), [[b"a"]], NOW, 1)
    assert ei.value.code == P.native.ORR_ESTATE
    with pytest.raises(P.OrrError) as ei:
        unsealed.append(np.ones((1, 5), np.float32), [NOW], [b"c"])           # wrong dimension
    assert ei.value.code == P.native.ORR_EDIM
    unsealed.close()

    c = {"emb": [np.array([1, 0, 0, 0], np.float32), None, np.array([0, 1, 0, 0], np.float32), np.zeros(4, np.float32)],
         "created": np.array([NOW - DAY, NOW, NOW - 2 * DAY, NOW + DAY], np.int64), "contents": ["", "   ", "Alpha", ""], "dim": 4}
    idx = build_index(c)
    corpus = oracle_corpus(c)
    for qv, text in (([1, 0, 0, 0], "alpha"), (None, "alpha"), ([0, 0, 0, 0], "the"), ([1, 0, 0, 0], "zzz")):
        for topk in (1, 3, 50):
            assert_same_ranking(idx, corpus, c, qv, text, topk, 300)
    # batch mixing a query with terms and one whose text is only stop words / symbols
    qs = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]], np.float32)
    texts = ["alpha", "the", "?!"]
    rows, scores, counts = idx.search(qs, [P.text.query_terms(t) for t in texts], NOW, 4, candidate_limit=300)
    for b in range(3):
        orow, osc, _ = corpus.search(qs[b], texts[b], NOW, 4, candidate_limit=300)
        assert list(rows[b, :counts[b]]) == list(orow) and np.array_equal(scores[b, :counts[b]], osc)
    idx.close()


def test_keyword_hit_list_overflow_grows_the_list_and_repeats_the_pass():
    """Short terms against a large vocabulary match more (term, token) pairs than the hit list holds: round 1 failed the
    whole batch with ORR_ENOMEM.  Now the list grows to the measured count and the pass runs again; forced here with a
    16-entry list."""
    P = pkg()
    rng = np.random.default_rng(321)
    n, dim = 3000, 32
    syll = ["ka", "re", "mi", "to", "ne", "su", "lo", "vi", "da", "po", "er", "in"]
    contents = [" ".join("".join(rng.choice(syll, size=int(rng.integers(2, 5)))) for _ in range(int(rng.integers(3, 12)))) for _ in range(n)]
    c = {"emb": [rng.standard_normal(dim).astype(np.float32) for _ in range(n)],
         "created": (NOW - rng.integers(0, 200 * DAY, n)).astype(np.int64), "contents": contents, "dim": dim}
    idx = build_index(c)
    corpus = oracle_corpus(c)
    idx.set_option("kw_hits_cap", 16)
    qs = rng.standard_normal((6, dim)).astype(np.float32)
    texts = ["e in", "ka re mi", "er", "to ne su lo", "zzz", "in er ka"]            # one- and two-letter terms: substrings of hundreds of tokens
    terms = [P.text.query_terms(t) for t in texts]
    idx.reset_search_stats()
    rows, scores, counts = idx.search(qs, terms, NOW, 10, candidate_limit=n)
    ss = idx.search_stats()
    assert ss["passes"] >= 2, ss                                                   # the overflowing pass + its repeat
    for b in range(6):
        orow, osc, _ = corpus.search(qs[b], texts[b], NOW, 10, candidate_limit=n)
        assert list(rows[b, :counts[b]]) == list(orow) and np.array_equal(scores[b, :counts[b]], osc), b
    # the larger list stays: the same batch again needs no repeat
    idx.reset_search_stats()
    rows2, scores2, _ = idx.search(qs, terms, NOW, 10, candidate_limit=n)
    assert idx.search_stats()["passes"] == 1 and np.array_equal(rows2, rows) and np.array_equal(scores2, scores)
    idx.close()


def test_four_wave_screening_kernel_variants_match_oracle():
    """From 65 queries up and 6+ K-tiles (dim >= 384) the int8 screening GEMM runs in its four-wave form
    (screen_tile4_kernel): 65..128 queries as 1 x 4 waves, 129..256 as 2 x 2 with the rows requested non-temporal, more than
    256 with several query tiles per row tile.  Row count not a multiple of the 256-row tile, batch sizes that are not
    multiples of 32, rows that quantise badly (they must survive the screen), a zero query, duplicates of a row
    (ties at the cut), queries with and without terms -- each batch against the oracle, bit for bit."""
    P = pkg()
    rng = np.random.default_rng(4242)
    n, dim = 200_019, 512                                                # 8 K-tiles; the last row tile holds 83 rows
    emb = rng.standard_normal((n, dim)).astype(np.float32)
    emb[1000:1040, 3] = np.float32(500.0)                                # one coordinate dominates the scale
    emb[2000:2040] = 0.0
    emb[3000:3040] *= np.float32(1e25)
    emb[n - 83:n - 60] = emb[5]                                          # duplicates inside the partial tile
    created = np.sort(NOW - rng.integers(0, 300 * DAY, n))[::-1].astype(np.int64)
    words = np.array(["alpha", "beta", "gamma", "delta", "kubernetes", "helm", "rollout"])
    contents = [" ".join(w).encode() for w in words[rng.integers(0, len(words), (n, 4))]]
    B = 300
    qs = rng.standard_normal((B, dim)).astype(np.float32)
    qs[0] = emb[5]
    qs[1] = emb[1010] * np.float32(0.01)
    qs[2] = 0.0
    qs[3] = emb[3005]
    qs[64] = emb[n - 1]
    qs[129] = emb[77_777] * np.float32(3.0)
    qs[257] = emb[n - 70]
    qs[299] = emb[123]
    # rows whose whole similarity to a query sits in ONE K-tile (the last 64 dimensions, the first 64), in output tiles that are
    # not a workgroup's first: a stream that hands a K-tile of the neighbouring output tile to the K loop loses exactly them
    qs[5] = 0.0
    qs[5, dim - 64:] = rng.standard_normal(64).astype(np.float32)
    qs[6] = 0.0
    qs[6, :64] = rng.standard_normal(64).astype(np.float32)
    for r in (70_001, 150_003, 199_990):
        emb[r] = np.float32(0.01) * emb[r] + np.float32(2.0) * qs[5]
        emb[r + 5] = np.float32(0.01) * emb[r + 5] + np.float32(3.0) * qs[6]
    idx = P.RecallIndex(dim=dim)
    for r0 in range(0, n, 50_000):
        idx.append(emb[r0:r0 + 50_000], created[r0:r0 + 50_000], contents[r0:r0 + 50_000])
    idx.seal()
    texts = [QUERY_TEXTS[b % len(QUERY_TEXTS)] if b % 3 else "" for b in range(B)]
    terms = [P.text.query_terms(t) if t else [] for t in texts]
    corpus = orc.OracleCorpus(emb, created, contents)
    check = {0, 1, 2, 3, 5, 6, 64, 65, 69, 127, 129, 199, 255, 256, 257, 299}
    for nb in (70, 128, 130, 256, 300):
        idx.set_profiling(True)
        idx.reset_search_stats()
        rows, scores, counts = idx.search(qs[:nb], terms[:nb], NOW, 10, candidate_limit=n)
        st = idx.kernel_stats()
        ss = idx.search_stats()
        idx.set_profiling(False)
        assert st["screen_i8_fused"]["launches"] >= 1 and "dot_exact" not in st, sorted(st)
        assert ss["exact_pass_queries"] == 0, ss                          # nothing had to fall back to the exact pass
        for b in sorted(x for x in check if x < nb):
            orow, osc, _ = corpus.search(qs[b], texts[b], NOW, 10, candidate_limit=n, threads=8)
            assert list(rows[b, :counts[b]]) == list(orow), (nb, b)
            assert np.array_equal(scores[b, :counts[b]], osc), (nb, b)
    # a deeper cut: the top 50 of unstructured queries, where the scores around the cut lie a hair apart -- any row the
    # screening pass under-estimates by more than its margin is missing from such a list
    rows50, scores50, counts50 = idx.search(qs[:256], terms[:256], NOW, 50, candidate_limit=n)
    for b in (7, 100, 200, 255):
        orow, osc, _ = corpus.search(qs[b], texts[b], NOW, 50, candidate_limit=n, threads=8)
        assert list(rows50[b, :counts50[b]]) == list(orow), b
        assert np.array_equal(scores50[b, :counts50[b]], osc), b
    # the same two queries through the kernels for small batches (one-query stream, 2..4, 5..64 queries)
    for nb in (1, 2, 4, 8, 33, 64):
        sel = ([5, 6] + list(range(7, 7 + nb)))[:nb]
        rows_s, scores_s, counts_s = idx.search(qs[sel], [terms[b] for b in sel], NOW, 10, candidate_limit=n)
        for pos, b in enumerate(sel[:2]):
            orow, osc, _ = corpus.search(qs[b], texts[b], NOW, 10, candidate_limit=n, threads=8)
            assert list(rows_s[pos, :counts_s[pos]]) == list(orow), (nb, b)
            assert np.array_equal(scores_s[pos, :counts_s[pos]], osc), (nb, b)
    # which workgroup gets which output tiles must not show: the same survivors, pair for pair, with 8 and with 64 persistent
    # workgroups as with one per CU (ORR_SCREEN_GRID is read at every launch)
    def run_with_grid(g, nbq):
        if g is None:
            os.environ.pop("ORR_SCREEN_GRID", None)
        else:
            os.environ["ORR_SCREEN_GRID"] = str(g)
        try:
            idx.reset_search_stats()
            out = idx.search(qs[:nbq], terms[:nbq], NOW, 10, candidate_limit=n)
            return out, idx.search_stats()["survivors_total"]
        finally:
            os.environ.pop("ORR_SCREEN_GRID", None)
    for nbq in (64, 128, 256):                                            # eight-wave form, four-wave form, 16 x 16 x 64 form
        (r_a, s_a, c_a), surv_a = run_with_grid(None, nbq)
        for g in (8, 64):
            (r_g, s_g, c_g), surv_g = run_with_grid(g, nbq)
            assert surv_g == surv_a, (nbq, g, surv_g, surv_a)
            assert np.array_equal(c_g, c_a) and np.array_equal(r_g, r_a) and np.array_equal(s_g, s_a), (nbq, g)
    # the same answers with the two-stage pass switched off (exact kernels only)
    idx.set_option("two_stage", 0)
    rows0, scores0, counts0 = idx.search(qs[:130], terms[:130], NOW, 10, candidate_limit=n)
    idx.set_option("two_stage", 1)
    rows1, scores1, counts1 = idx.search(qs[:130], terms[:130], NOW, 10, candidate_limit=n)
    assert np.array_equal(counts0, counts1) and np.array_equal(rows0, rows1) and np.array_equal(scores0, scores1)
    idx.close()


def test_shard_records_are_complete_when_orr_search_shard_returns():
    """The stream contract the record exchange leans on (sharded.py, include/omnirecall_hip.h conventions): a DEVICE-resident
    `out` of orr_search_shard(_ex) is complete when the call returns, although the library wrote it on its own non-blocking
    stream.  Here the buffer is read back on a fresh stream of the caller's immediately after the call (what a collective
    on the process group's stream does) and must equal the records of the same search written to host memory."""
    import importlib
    import torch
    P = pkg()
    sh = importlib.import_module("omni_recall_rag_amd.sharded")
    rng = np.random.default_rng(99)
    n, dim = 300_000, 128
    emb = rng.standard_normal((n, dim)).astype(np.float32)
    created = np.sort(NOW - rng.integers(0, 300 * DAY, n))[::-1].astype(np.int64)
    words = np.array(["alpha", "beta", "gamma", "delta", "kubernetes", "helm"])
    contents = [" ".join(w).encode() for w in words[rng.integers(0, len(words), (n, 3))]]
    idx = P.RecallIndex(dim=dim, row_base=1_000_000)
    for r0 in range(0, n, 100_000):
        idx.append(emb[r0:r0 + 100_000], created[r0:r0 + 100_000], contents[r0:r0 + 100_000])
    idx.seal()
    front = sh.ShardedRecallSearch(idx, dim, "cuda:0")          # no process group: one rank, the same code path minus the collective
    snaps = []
    side = torch.cuda.Stream()

    def snapshot(mine, nb, kprime):
        with torch.cuda.stream(side):                           # NOT the library's stream, and nothing synchronises with it
            host = torch.empty(mine.numel(), dtype=torch.uint8, pin_memory=True)
            host.copy_(mine, non_blocking=True)
        side.synchronize()
        snaps.append((host.numpy().copy(), nb, kprime))

    front._shard_search_done = snapshot
    B, kprime = 48, 32
    q = torch.from_numpy(rng.standard_normal((B, dim)).astype(np.float32)).cuda()
    terms = [P.text.query_terms(QUERY_TEXTS[b % len(QUERY_TEXTS)]) for b in range(B)]
    for trial in range(3):
        snaps.clear()
        rows, scores, counts = front.search_from(0, q, terms, NOW, 10, 1_000_000 + n, kprime=kprime)
        assert snaps and snaps[0][1] == B
        want = idx.search_shard(q, terms, NOW, kprime, 1_000_000 + n, topk=10, shard_pass=0)       # the same search into host memory
        got = snaps[0][0].view(P.CAND_DTYPE).reshape(B, kprime + 1)
        for f in ("order_key", "row_id", "matches", "flags", "created_ticks"):
            assert np.array_equal(got[f], want[f]), (trial, f)
        assert np.array_equal(got["dot"], want["dot"]) and np.array_equal(got["norm_b"], want["norm_b"])
        r2, s2, c2 = idx.search(q, terms, NOW, 10, candidate_limit=1_000_000 + n)
        assert np.array_equal(rows, r2) and np.array_equal(scores, s2) and np.array_equal(counts, c2)
    idx.close()
