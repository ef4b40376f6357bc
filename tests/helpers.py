"""Shared test helpers: build the same corpus for the oracle (checker) and for the
HIP library (through the C ABI), and compare ranked results bit for bit."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft            # noqa: E402
from oracle import oracle_py as orc        # noqa: E402  (checker only)

NOW = 639144000000000000
DAY = 864000000000

WORDS = ["alpha", "Beta", "GAMMA", "delta", "epsilon", "zeta", "kubernetes", "helm", "azure", "cosmos",
         "vector", "search", "the", "of", "and", "what", "is", "a", "Été", "naïve", "x1", "q?", "db",
         "Deployment", "yaml", "chart", "net", "working", "ab", "abc", "abcd", "abcde", "abcdefghij"]


def pkg():
    return graft.load_package()


def has_gpu() -> bool:
    try:
        return pkg().native.hip.orr_device_count() > 0
    except Exception:
        return False


def random_corpus(rng, n, dim, p_null=0.1, created_spread_days=400, sorted_created=False, n_words=(0, 40),
                  dup_frac=0.05):
    """Returns dict(emb list (None for null rows), created int64[n], contents list[str])."""
    emb = []
    for r in range(n):
        if dim == 0 or rng.random() < p_null:
            emb.append(None)
        else:
            emb.append((rng.standard_normal(dim) * rng.choice([1.0, 1e-2, 30.0])).astype(np.float32))
    # exact duplicates -> exact score ties
    for _ in range(int(n * dup_frac)):
        a, b = rng.integers(0, n, 2)
        emb[b] = None if emb[a] is None else emb[a].copy()
    n_docs = max(1, n // 4)
    doc_created = NOW - rng.integers(-2 * DAY, created_spread_days * DAY, n_docs)
    doc_of = rng.integers(0, n_docs, n)
    created = doc_created[doc_of].astype(np.int64)
    if sorted_created:
        created = np.sort(created)[::-1].copy()
    contents = []
    for r in range(n):
        k = int(rng.integers(n_words[0], n_words[1] + 1))
        contents.append(" ".join(rng.choice(WORDS, size=k)) if k else "")
    for _ in range(int(n * dup_frac)):
        a, b = rng.integers(0, n, 2)
        contents[b] = contents[a]
        created[b] = created[a]
    return {"emb": emb, "created": created, "contents": contents, "dim": dim}


def oracle_corpus(c):
    return orc.OracleCorpus(c["emb"], c["created"], c["contents"])


def build_index(c, device=0, chunk=None, row_base=0):
    """Appends in store enumeration order (runs of rows with / without embedding)."""
    P = pkg()
    n, dim = len(c["created"]), c["dim"]
    idx = P.RecallIndex(dim=dim, device=device, row_base=row_base)
    lower = [P.text.lower_invariant(s) for s in c["contents"]]
    r = 0
    while r < n:
        has = c["emb"][r] is not None
        e = r
        while e < n and (c["emb"][e] is not None) == has and (chunk is None or e - r < chunk):
            e += 1
        emb = np.stack(c["emb"][r:e]).astype(np.float32) if has and dim > 0 else None
        idx.append(emb, c["created"][r:e], lower[r:e])
        r = e
    idx.seal()
    return idx


def assert_same_ranking(idx, corpus, c, qvec, query, topk, limit, now=NOW):
    """HIP path vs oracle: identical row ids, identical order, bit-identical fp64 scores."""
    P = pkg()
    terms = [P.text.query_terms(query)]
    q = None if qvec is None else np.asarray(qvec, dtype=np.float32).reshape(1, -1)
    rows, scores, counts = idx.search(q, terms, now, topk, candidate_limit=limit)
    orows, oscores, _ = corpus.search([] if qvec is None else qvec, query, now, topk, candidate_limit=limit)
    k = int(counts[0])
    assert k == len(orows), (k, len(orows))
    got_rows = list(rows[0, :k])
    assert got_rows == list(orows), f"rows differ: {got_rows[:12]} vs {list(orows)[:12]} (q={query!r}, k={topk}, limit={limit})"
    a, b = scores[0, :k], oscores
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    assert same.all(), f"scores differ: {a[~same][:5]} vs {b[~same][:5]}"
    return rows[0, :k], scores[0, :k]
