"""Frozen input/output vectors (tests/golden/oracle_vectors.json, written by tools/gen_golden_vectors.py from the
CPU oracle): the oracle must keep reproducing them (no GPU needed), and the HIP path must produce exactly the
same rows and bit-identical fp64 scores.  What the cases cover is listed in the generator's docstring."""
import json
import os

import numpy as np
import pytest

from helpers import ROOT, orc, pkg


def _load():
    with open(os.path.join(ROOT, "tests", "golden", "oracle_vectors.json"), encoding="utf-8") as f:
        return json.load(f)


def _corpus(case):
    emb = []
    for row in case["embeddings"]:
        if row is None:
            emb.append(None)
        else:
            emb.append(np.asarray([np.nan if x is None else np.inf if x == "inf" else x for x in row], dtype=np.float32))
    return emb, np.asarray(case["createdTicks"], dtype=np.int64), case["contents"]


def _qvec(case, q):
    return None if q["vector"] is None else np.asarray(case["queryVectors"][q["vector"]], dtype=np.float32)


def _same_scores(got, hexes):
    want = np.asarray([float.fromhex(h) for h in hexes], dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    return got.shape == want.shape and bool(((got == want) | (np.isnan(got) & np.isnan(want))).all())


CASES = _load()


@pytest.mark.parametrize("ci", range(len(CASES["cases"])))
def test_oracle_reproduces_the_frozen_vectors(ci):
    case = CASES["cases"][ci]
    emb, created, contents = _corpus(case)
    corpus = orc.OracleCorpus(emb, created, contents)
    for q in case["queries"]:
        qv = _qvec(case, q)
        rows, scores, rounded = corpus.search([] if qv is None else qv, q["text"], CASES["nowTicks"], q["topK"],
                                              candidate_limit=q["candidateLimit"])
        assert [int(r) for r in rows] == q["rows"], (case["name"], q["text"], q["topK"])
        assert _same_scores(scores, q["scoresHex"]), (case["name"], q["text"], q["topK"])
        assert [None if r != r else float(r) for r in rounded] == q["rounded"]


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(len(CASES["cases"])))
def test_hip_path_reproduces_the_frozen_vectors(ci):
    P = pkg()
    case = CASES["cases"][ci]
    emb, created, contents = _corpus(case)
    n, dim = len(contents), case["dim"]
    idx = P.RecallIndex(dim=dim)
    lower = [P.text.lower_invariant(s) for s in contents]
    r = 0
    while r < n:                                              # runs of rows with / without an embedding, in store order
        has = emb[r] is not None
        e = r
        while e < n and (emb[e] is not None) == has:
            e += 1
        idx.append(np.stack(emb[r:e]) if has and dim > 0 else None, created[r:e], lower[r:e])
        r = e
    idx.seal()
    for q in case["queries"]:
        qv = _qvec(case, q)
        rows, scores, counts = idx.search(None if qv is None else qv[None, :], [P.text.query_terms(q["text"])], CASES["nowTicks"],
                                          q["topK"], candidate_limit=q["candidateLimit"])
        k = int(counts[0])
        assert [int(x) for x in rows[0, :k]] == q["rows"], (case["name"], q["text"], q["topK"], q["candidateLimit"])
        assert _same_scores(scores[0, :k], q["scoresHex"]), (case["name"], q["text"], q["topK"])
        assert [None if s != s else P.text.round4(float(s)) for s in scores[0, :k]] == q["rounded"]
    idx.close()
