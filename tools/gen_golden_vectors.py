#!/usr/bin/env python3
"""Generates tests/golden/oracle_vectors.json: small frozen input/output vectors of the hybrid recall scorer
("additional golden vectors the build must generate itself").

The outputs come from the CPU oracle (oracle/recall_oracle.c, the restatement of
RecallSearchService.cs:20-119 + InMemoryIngestionStore.cs:57-65 that tests/golden/reference_kats.json pins
against the reference's own tests).  Freezing them pins the oracle against regressions and gives the HIP path
fixed expected values.  Scores are stored as C99 hex doubles (bit-exact) next to the 4-decimal rounded ones; a query's "vector" is an
index into its case's "queryVectors" (null = no embedding).

Covered on purpose: D in {2, 3, 16, 768}; rows without an embedding; exact duplicates (exact score ties, broken
by candidate order) and near-duplicates; a query of another dimension, an empty query vector, a zero-norm
query; all-stop-word, mixed-case and non-ASCII query texts and contents; created > now; topK in
{-1, 0, 1, 10, N + 5}; candidate_limit in {300, N, small}; NaN and infinity in an embedding.

Usage: python tools/gen_golden_vectors.py   (rewrites the fixture; commit the result)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as orc  # noqa: E402

NOW = 639144000000000000            # 2026-05-15T00:00:00Z in .NET ticks
DAY = 864000000000
WORDS = ["azure", "cosmos", "db", "vector", "search", "kubernetes", "helm", "chart", "Deployment", "YAML", "naïve",
         "café", "ÉTÉ", "中文", "straße", "the", "and", "of", "what", "is", "alpha", "beta", "x1", "k8s"]
TEXTS = ["azure", "the kubernetes helm", "what is the", "VECTOR Search cosmos", "été naïve", "中文 straße k8s", "zzz",
         "  alpha   beta  alpha ", "deployment yaml", ""]


def f32(x):
    """A float32 as a short JSON number: 9 significant digits identify it (the decimal stays ~100 times closer to the
    value than to a rounding boundary, so reading it as a double and rounding to float32 gives it back)."""
    v = float("%.9g" % np.float32(x))
    assert np.float32(v) == np.float32(x)
    return v


def make_case(rng, n, dim, name):
    emb = []
    for r in range(n):
        if dim == 0 or rng.random() < 0.12:
            emb.append(None)
        else:
            emb.append((rng.standard_normal(dim) * rng.choice([1.0, 1e-3, 40.0])).astype(np.float32))
    for _ in range(max(1, n // 8)):                       # exact duplicates and near-duplicates
        a, b = rng.integers(0, n, 2)
        if emb[a] is not None:
            emb[b] = emb[a].copy() if rng.random() < 0.5 else (emb[a] * np.float32(1.0 + 1e-6)).astype(np.float32)
    docs = rng.integers(0, max(1, n // 4), n)
    doc_created = NOW - rng.integers(-2 * DAY, 200 * DAY, max(1, n // 4))       # some in the future
    created = doc_created[docs].astype(np.int64)
    contents = [" ".join(rng.choice(WORDS, size=int(rng.integers(0, 9)))) for _ in range(n)]
    for _ in range(max(1, n // 8)):
        a, b = rng.integers(0, n, 2)
        contents[b], created[b] = contents[a], created[a]
    return {"name": name, "dim": dim, "emb": emb, "created": created, "contents": contents}


def main():
    rng = np.random.default_rng(20260515)
    cases = []
    shapes = [(5, 2), (17, 3), (48, 16), (64, 16), (40, 0), (6, 768), (33, 2), (300, 3), (310, 16)]
    for ci, (n, dim) in enumerate(shapes):
        c = make_case(rng, n, dim, "case%02d_n%d_d%d" % (ci, n, dim))
        if ci == 2:
            c["emb"][3] = np.full(dim, np.nan, np.float32)
            c["emb"][7][0] = np.float32(np.inf)
        corpus = orc.OracleCorpus(c["emb"], c["created"], c["contents"])
        queries = []
        real = [e for e in c["emb"] if e is not None]
        qvecs = [None, np.zeros(max(dim, 1), np.float32)]
        if dim:
            qvecs += [rng.standard_normal(dim).astype(np.float32), rng.standard_normal(dim + 1).astype(np.float32)]
            if real:
                qvecs.append(real[0].copy())                                    # cosine 1 on a stored row and its duplicates
        for qi, qv in enumerate(qvecs):
            for ti in (qi, qi + 3, qi + 6):
                text = TEXTS[ti % len(TEXTS)]
                for topk, limit in ((10, n), (1, 300), (-1, n), (0, 2), (n + 5, n), (3, 7)):
                    if (qi + ti + topk) % 3 and topk not in (10,):             # thin the grid, keep it varied
                        continue
                    if topk > 64 and (qi, ti) != (2, 5):                       # the long answers of the larger corpora: once
                        continue
                    rows, scores, rounded = corpus.search([] if qv is None else qv, text, NOW, topk, candidate_limit=limit)
                    queries.append({"vector": None if qv is None else qi, "text": text, "topK": topk,
                                    "candidateLimit": limit, "rows": [int(r) for r in rows],
                                    "scoresHex": [float(s).hex() for s in scores],
                                    "rounded": [None if r != r else float(r) for r in rounded]})
        cases.append({"name": c["name"], "dim": dim,
                      "queryVectors": [None if qv is None else [f32(x) for x in qv] for qv in qvecs],
                      "embeddings": [None if e is None else [None if x != x else ("inf" if x == np.inf else f32(x)) for x in e] for e in c["emb"]],
                      "createdTicks": [int(t) for t in c["created"]], "contents": c["contents"], "queries": queries})
    out = {"nowTicks": NOW, "generator": "tools/gen_golden_vectors.py (CPU oracle)", "cases": cases}
    path = os.path.join(ROOT, "tests", "golden", "oracle_vectors.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, separators=(",", ":"))
    print(path, os.path.getsize(path), "bytes,", sum(len(c["queries"]) for c in cases), "queries in", len(cases), "cases")


if __name__ == "__main__":
    main()
