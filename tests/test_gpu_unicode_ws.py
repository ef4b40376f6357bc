"""The token index splits content on char.IsWhiteSpace; a term must match iff the reference's
plain substring test does, whatever separates the words (NBSP, ideographic space, tabs, ...)
and wherever the term sits inside a token.  All special characters are written as escapes."""
import numpy as np
import pytest

from helpers import DAY, NOW, assert_same_ranking, build_index, oracle_corpus

pytestmark = pytest.mark.gpu

SEPS = [" ", "\t", "\n", "\r\n", " ", "　", " ", " ", " ", " ", "", "  \t "]
NOT_SEPS = ["​", "", "᠎", "-", "_", "/"]          # zero-width space etc. do NOT split
WORDS = ["alpha", "beta", "gamma", "kubernetes", "naïve", "Été", "x", "漢字", "azure-functions"]
QUERIES = ["alpha", "beta​gamma", "kubernetes naïve", "été 漢字", "functions azure-", "ab x",
           "mm ta᠎alpha", "azure-functions_alpha", "x/", "-", "字", "ab alpha"]


def test_unicode_whitespace_tokenisation_matches_contains_semantics():
    rng = np.random.default_rng(5)
    contents = []
    for r in range(400):
        parts = []
        for _ in range(int(rng.integers(1, 12))):
            parts.append(str(rng.choice(WORDS)))
            parts.append(str(rng.choice(SEPS if rng.random() < 0.8 else NOT_SEPS)))
        contents.append("".join(parts))
    n = len(contents)
    c = {"emb": [None] * n, "created": (NOW - rng.integers(0, 50 * DAY, n)).astype(np.int64), "contents": contents, "dim": 0}
    idx = build_index(c)
    corpus = oracle_corpus(c)
    for text in QUERIES:
        for topk in (7, 400):
            assert_same_ranking(idx, corpus, c, None, text, topk, n)
    idx.close()
