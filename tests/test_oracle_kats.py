"""Pins the CPU oracle against every known-answer case the reference's own tests
hold for the recall-search path.  CPU only."""
import numpy as np
import pytest

from oracle import oracle_py as orc


def _corpus(seed):
    chunks = seed["chunks"]
    return (orc.OracleCorpus([c["embedding"] for c in chunks], [c["createdTicks"] for c in chunks],
                             [c["content"] for c in chunks]), chunks,
            {d["id"]: d["fileName"] for d in seed["documents"]})


def test_kat_file_has_all_reference_cases(kats):
    assert len(kats["cases"]) == 5


@pytest.mark.parametrize("idx", range(5))
def test_reference_known_answers_c_oracle(kats, idx):
    case = kats["cases"][idx]
    corpus, chunks, files = _corpus(case["seed"])
    rows, scores, rounded = corpus.search(case["queryVector"], case["query"], kats["nowTicks"], case["topK"])
    assert len(rows) > 0                                        # Assert.NotEmpty
    top = chunks[rows[0]]
    a = case["asserted"]
    if "rank1DocumentId" in a:
        assert top["documentId"] == a["rank1DocumentId"]
    if "rank1FileName" in a:
        assert files[top["documentId"]] == a["rank1FileName"]
    assert [chunks[r]["documentId"] for r in rows] == case["derivedOrder"]
    for r, s in zip(rows, scores):
        want = case.get("derivedScores", {}).get(chunks[r]["documentId"])
        if want is not None:
            assert s == want                                    # bit-exact binary64
    assert all(rd == orc.round4(s) for s, rd in zip(scores, rounded))


@pytest.mark.parametrize("idx", range(5))
def test_reference_known_answers_python_restatement(kats, idx):
    """The independent numpy/Python reading of the C# must agree with the C one."""
    case = kats["cases"][idx]
    corpus, chunks, _ = _corpus(case["seed"])
    rows_c, scores_c, _ = corpus.search(case["queryVector"], case["query"], kats["nowTicks"], case["topK"])
    rows_p, scores_p = orc.py_search([c["embedding"] for c in chunks], [c["createdTicks"] for c in chunks],
                                     [c["content"] for c in chunks], case["queryVector"], case["query"],
                                     kats["nowTicks"], case["topK"])
    assert list(rows_c) == rows_p
    assert list(scores_c) == scores_p


def test_kat3_discriminates_stop_word_filter(kats):
    """Without the stop-word filter doc-3 would win 0.25 vs 0.15."""
    assert orc.query_terms("what is the kubernetes") == [b"kubernetes"]
    assert orc.keyword_score("what is the kubernetes", "what is the and of for") == 0.0
    assert orc.keyword_score("what is the kubernetes", "kubernetes deployment yaml and helm chart") == 1.0


def test_kat5_keyword_score(kats):
    case = kats["cases"][4]
    c = case["seed"]["chunks"][0]
    assert orc.query_terms(case["query"]) == [b"backend", b"did", b"we", b"choose?"]
    assert orc.keyword_score(case["query"], c["content"]) == case["derivedKeywordScore"]
