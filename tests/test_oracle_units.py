"""Unit checks of the CPU oracle: each function against the independent Python
restatement and against hand-computed values for the edge cases SURVEY.md §4/§8c
lists (guards, ties, NaN order, rounding, topK floor, candidate_limit)."""
import math

import numpy as np
import pytest

from oracle import oracle_py as orc

NOW = 639144000000000000
DAY = 864000000000


def test_cosine_guards():
    assert orc.cosine([], [1, 2]) == 0.0                   # empty query        :71
    assert orc.cosine([1, 2], None) == 0.0                 # null chunk vector
    assert orc.cosine([1, 2], []) == 0.0                   # empty chunk vector
    assert orc.cosine([1, 2], [1, 2, 3]) == 0.0            # dimension mismatch
    assert orc.cosine([0, 0], [1, 2]) == 0.0               # normA <= 0         :84
    assert orc.cosine([1, 2], [0, 0]) == 0.0               # normB <= 0
    assert math.isnan(orc.cosine([1, float("nan")], [1, 2]))   # NaN passes the <= 0 guard


@pytest.mark.parametrize("d", [1, 2, 3, 17, 768, 3072])
def test_cosine_matches_python_restatement(d):
    rng = np.random.default_rng(d)
    for scale in (1.0, 1e-3, 37.0):
        a = (rng.standard_normal(d) * scale).astype(np.float32)
        b = (rng.standard_normal(d) * scale).astype(np.float32)
        assert orc.cosine(a, b) == orc.py_cosine(a, b)


def test_products_are_rounded_to_binary32_before_widening():
    # 16777217 = 2^24+1 is not a float; (2^12+1)^2 = 2^24 + 2^13 + 1 rounds to 2^24 + 2^13 in binary32
    a = np.array([4097.0], np.float32)
    assert orc.dot(a, a) == float(np.float32(4097.0) * np.float32(4097.0)) == 16785408.0
    assert 4097.0 * 4097.0 == 16785409.0


def test_dot_is_sequential_fp64():
    rng = np.random.default_rng(3)
    a = (rng.standard_normal(3072) * 1e3).astype(np.float32)
    b = (rng.standard_normal(3072) * 1e-3).astype(np.float32)
    acc = 0.0
    for p in (a * b):
        acc += float(p)
    assert orc.dot(a, b) == acc


def test_recency():
    assert orc.recency(NOW, NOW) == 1.0
    assert orc.recency(NOW + 5 * DAY, NOW) == 1.0                   # created in the future: age clamped :117
    assert orc.recency(NOW - 30 * DAY, NOW) == math.exp(-1.0)
    for age in (1, 12345678901, 365 * DAY, 4000 * DAY):
        assert orc.recency(NOW - age, NOW) == orc.py_recency(NOW - age, NOW)


def test_round4_is_bankers_rounding_on_the_scaled_double():
    assert orc.round4(0.30000000000000004) == 0.3
    assert orc.round4(0.9999999999999999) == 1.0
    assert orc.round4(0.00005) == round(0.00005 * 1e4) / 1e4       # Python round() is also half-even
    assert orc.round4(0.12345) == 0.1234 or orc.round4(0.12345) == 0.1235
    for x in (0.12345, 0.12355, 0.5, 2.5e-4, 3.5e-4, -0.00025, 1e17, 123.456789):
        want = x if abs(x) >= 1e16 else float(np.rint(x * 1e4) / 1e4)
        assert orc.round4(x) == want


def test_query_terms_split_lower_distinct_stopwords():
    assert orc.query_terms("  Azure\tAZURE azure  Cosmos ") == [b"azure", b"cosmos"]
    assert orc.query_terms("the of and") == [b"the", b"of", b"and"]         # all stop words: raw terms kept :107
    assert orc.query_terms("The Kubernetes") == [b"kubernetes"]            # stop test is on the lowercased term
    assert orc.query_terms("a b c　d") == [b"b", b"c", b"d"]  # Unicode spaces split; "a" is a stop word
    assert orc.query_terms("x\x1cy") == [b"x\x1cy"]                        # U+001C is not .NET whitespace
    assert orc.query_terms("ÄRGER ärger") == ["ärger".encode()]
    assert orc.query_terms("   ") == []
    for q in ("what is the kubernetes", "Hello hello HELLO world", "  a an  the ", "İstanbul i̇stanbul"):
        assert [t.decode() for t in orc.query_terms(q)] == orc.py_query_terms(q)


def test_keyword_is_substring_not_token_match():
    assert orc.keyword_score("net", "Kubernetes networking") == 1.0            # substring inside a token (F6)
    assert orc.keyword_score("kube net", "KUBERNETES") == 1.0
    assert orc.keyword_score("kube zzz", "kubernetes") == 0.5
    assert orc.keyword_score("kubernetes", "   \n ") == 0.0                    # blank content :93
    assert orc.keyword_score("  ", "kubernetes") == 0.0                        # blank query   :92
    assert orc.keyword_score("ab", "a b") == 0.0                               # no match across whitespace
    rng = np.random.default_rng(11)
    words = ["alpha", "Beta", "GAMMA", "delta", "the", "of", "Zeta", "Été", "x1", "q?"]
    for _ in range(200):
        q = " ".join(rng.choice(words, size=rng.integers(1, 6)))
        c = " ".join(rng.choice(words, size=rng.integers(0, 12)))
        assert orc.keyword_score(q, c) == orc.py_keyword_score(q, c)


def test_lower_invariant_and_blank():
    assert orc.lower_invariant("ABC xyz ÄÖ İ") == "abc xyz äö İ".encode()
    assert orc.is_blank("") and orc.is_blank(" \t\r\n ") and not orc.is_blank(" x ")


def test_snippet():
    assert orc.snippet("  a\nb\r\nc  ") == b"a b  c"
    long = "w" * 200
    assert orc.snippet(long) == b"w" * 180 + b"..."
    assert orc.snippet("w" * 180) == b"w" * 180


def _toy_corpus():
    emb = [[1, 0], [0, 1], [1, 1], None, [2, 0], [1, 0, 0]]
    created = [NOW - 3 * DAY, NOW - 1 * DAY, NOW - 1 * DAY, NOW, NOW - 9 * DAY, NOW - 2 * DAY]
    contents = ["alpha beta", "beta gamma", "gamma delta", "alpha", "", "alpha"]
    return emb, created, contents


def test_recent_chunks_is_stable_created_desc_with_floor_of_one():
    emb, created, contents = _toy_corpus()
    c = orc.OracleCorpus(emb, created, contents)
    assert list(c.recent_chunks(300)) == [3, 1, 2, 5, 0, 4]      # rows 1,2 tie on created: enumeration order kept
    assert list(c.recent_chunks(2)) == [3, 1]
    assert list(c.recent_chunks(0)) == [3]                       # Math.Max(1, maxCount)
    assert list(c.recent_chunks(-5)) == [3]


@pytest.mark.parametrize("topk", [-1, 0, 1, 3, 10])
@pytest.mark.parametrize("limit", [1, 2, 300])
def test_search_matches_python_restatement(topk, limit):
    emb, created, contents = _toy_corpus()
    c = orc.OracleCorpus(emb, created, contents)
    for qvec, q in (([1, 0], "alpha"), ([], "beta gamma"), ([1, 0, 0], "the alpha"), ([0, 0], "zzz")):
        rows, scores, _ = c.search(qvec, q, NOW, topk, candidate_limit=limit)
        prow, pscores = orc.py_search(emb, created, contents, qvec, q, NOW, topk, candidate_limit=limit)
        assert list(rows) == prow and list(scores) == pscores
        assert len(rows) == min(max(1, topk), min(max(1, limit), len(created)))


def test_exact_ties_keep_candidate_order_and_nan_sorts_last():
    emb = [[1, 0], [1, 0], [float("nan"), 0], [1, 0]]
    created = [NOW, NOW, NOW, NOW]
    contents = ["x", "x", "x", "x"]
    c = orc.OracleCorpus(emb, created, contents)
    rows, scores, _ = c.search([1, 0], "x", NOW, 10)
    assert list(rows) == [0, 1, 3, 2]
    assert math.isnan(scores[3]) and scores[0] == scores[1] == scores[2]
    # ThenByDescending(CreatedAtUtc) only matters for ties in score
    created2 = [NOW - DAY, NOW, NOW, NOW - DAY]
    c2 = orc.OracleCorpus([None] * 4, created2, ["q"] * 4)
    rows2, _, _ = c2.search([], "zz", NOW, 4)
    assert list(rows2) == [1, 2, 0, 3]


def test_threads_do_not_change_results():
    rng = np.random.default_rng(5)
    n, d = 500, 64
    emb = rng.standard_normal((n, d)).astype(np.float32)
    created = (NOW - rng.integers(0, 400 * DAY, n)).astype(np.int64)
    contents = [" ".join(rng.choice(["alpha", "beta", "gamma", "delta"], 5)) for _ in range(n)]
    c = orc.OracleCorpus(emb, created, contents)
    q = rng.standard_normal(d).astype(np.float32)
    r1 = c.search(q, "alpha delta", NOW, 10, candidate_limit=n, threads=1)
    r4 = c.search(q, "alpha delta", NOW, 10, candidate_limit=n, threads=4)
    assert all((a == b).all() for a, b in zip(r1, r4))
