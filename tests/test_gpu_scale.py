"""Parity at BASELINE.json's bench size (1M x 3072) through size-independent
properties, plus an oracle check on a sub-range the CPU finishes in seconds."""
import importlib

import numpy as np
import pytest

from helpers import orc, pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    import torch
    P = pkg()
    syn = importlib.import_module("omni_recall_rag_amd.synthetic")
    n, dim = 1_000_000, 3072
    idx = P.RecallIndex(dim=dim, capacity_rows=n)
    step = 32768
    for r0 in range(0, n, step):
        m = min(step, n - r0)
        pool, off = syn.contents(r0, m, "cuda:0")
        idx.append(syn.embeddings(r0, m, dim, "cuda:0"), syn.created_ticks(r0, m, n, "cuda:0"), pool, off)
    torch.cuda.synchronize()
    idx.seal()
    yield P, syn, idx, n, dim
    idx.close()


def test_planted_rows_win_and_batches_agree_at_1m(big):
    P, syn, idx, n, dim = big
    B = 4
    q = syn.query_vectors(0, B, dim, n, "cuda:0")
    texts = syn.query_texts(0, B, n)
    terms = [P.text.query_terms(t) for t in texts]
    rows, scores, counts = idx.search(q, terms, syn.NOW_TICKS, 10, candidate_limit=n)
    planted = syn.planted_rows(0, B, n)
    assert list(rows[:, 0]) == planted                      # the planted row is rank 1
    assert (counts == 10).all()
    assert (np.diff(scores, axis=1) <= 0).all()             # sorted by score
    for b in range(B):                                      # one by one == batched (idempotence)
        r1, s1, _ = idx.search(q[b:b + 1], terms[b:b + 1], syn.NOW_TICKS, 10, candidate_limit=n)
        assert np.array_equal(r1[0], rows[b]) and np.array_equal(s1[0], scores[b])
    # the winner's score, recomputed by the oracle from regenerated inputs
    for b in range(B):
        r = planted[b]
        e = syn.embeddings(r, 1, dim).numpy()[0]
        pool, off = syn.contents(r, 1)
        created = int(syn.created_ticks(r, 1, n)[0])
        cor = orc.OracleCorpus(e[None, :], [created], (pool.numpy(), off.numpy()))
        _, osc, _ = cor.search(q[b].cpu().numpy(), texts[b], syn.NOW_TICKS, 1, candidate_limit=1)
        assert scores[b, 0] == osc[0]


def test_candidate_limit_prefix_against_oracle_at_1m(big):
    """candidate_limit = 20000 scores only the newest 20000 rows: small enough for the oracle."""
    P, syn, idx, n, dim = big
    m = 20000
    emb = syn.embeddings(0, m, dim).numpy()
    created = syn.created_ticks(0, m, n).numpy()
    pool, off = syn.contents(0, m)
    cor = orc.OracleCorpus(emb, created, (pool.numpy(), off.numpy()))
    for b in (0, 1):
        q = syn.query_vectors(b, 1, dim, n).numpy()
        text = syn.query_texts(b, 1, n)[0]
        rows, scores, _ = idx.search(q, [P.text.query_terms(text)], syn.NOW_TICKS, 10, candidate_limit=m)
        orow, osc, _ = cor.search(q[0], text, syn.NOW_TICKS, 10, candidate_limit=m, threads=8)
        assert list(rows[0]) == list(orow) and np.array_equal(scores[0], osc)
        r300, s300, _ = idx.search(q, [P.text.query_terms(text)], syn.NOW_TICKS, 10, candidate_limit=300)
        o300, os300, _ = cor.search(q[0], text, syn.NOW_TICKS, 10, candidate_limit=300)
        assert list(r300[0]) == list(o300) and np.array_equal(s300[0], os300)


def test_batched_passes_equal_the_exact_pass_at_1m(big):
    """256 queries at the bench size: the two-stage pass (screening GEMM over the bf16 shadow, exact
    re-score), its shadow-less form and the split-bf16 pass must all return exactly what the
    reference-arithmetic kernel returns four queries at a time -- rows, order and fp64 scores."""
    P, syn, idx, n, dim = big
    B = 256
    q = syn.query_vectors(1000, B, dim, n, "cuda:0")
    texts = syn.query_texts(1000, B, n)
    terms = [P.text.query_terms(t) for t in texts]
    exact_rows = np.empty((B, 10), dtype=np.int64)
    exact_scores = np.empty((B, 10), dtype=np.float64)
    idx.set_option("two_stage", 0)                          # baseline: the reference-arithmetic kernel over every row
    for b0 in range(0, B, 4):
        r, s, c = idx.search(q[b0:b0 + 4], terms[b0:b0 + 4], syn.NOW_TICKS, 10, candidate_limit=n)
        assert (c == 10).all()
        exact_rows[b0:b0 + 4], exact_scores[b0:b0 + 4] = r, s
    assert list(exact_rows[:, 0]) == syn.planted_rows(1000, B, n)
    for mode in (1, 2, 0):
        idx.set_option("two_stage", mode)
        idx.set_profiling(True)
        r, s, c = idx.search(q, terms, syn.NOW_TICKS, 10, candidate_limit=n)
        stats = idx.kernel_stats()
        idx.set_profiling(False)
        assert np.array_equal(r, exact_rows) and np.array_equal(s, exact_scores), mode
        if mode:
            assert ("screen_i8_fused" if mode == 1 else "gemm_dot_bf16x1_fused") in stats
            if mode == 1:                                              # certified without a repeat through the split pass
                assert stats["screen_i8_fused"]["launches"] == 1 and "gemm_dot_bf16x3" not in stats, sorted(stats)
            else:
                assert stats["gemm_dot_bf16x3"]["launches"] == 1
    idx.set_option("two_stage", 1)
    # small batches take the same route: 1..4 queries stream the int8 shadow (no matrix core), 5+ use the GEMM
    for nb in (1, 2, 3, 4, 5, 8, 9, 33):
        idx.set_profiling(True)
        r, s, c = idx.search(q[:nb], terms[:nb], syn.NOW_TICKS, 10, candidate_limit=n)
        st = idx.kernel_stats()
        assert ("screen_gemv_i8" if nb <= 4 else "screen_i8_fused") in st, sorted(st)
        idx.set_profiling(False)
        assert np.array_equal(r, exact_rows[:nb]) and np.array_equal(s, exact_scores[:nb]), nb
    idx.set_option("two_stage", 0)
    r, s, c = idx.search(q[:24], terms[:24], syn.NOW_TICKS, 10, candidate_limit=n)      # streaming GEMV
    assert np.array_equal(r, exact_rows[:24]) and np.array_equal(s, exact_scores[:24])
    idx.set_option("two_stage", 1)
    # a candidate_limit that cuts the corpus goes through the same pass on the prefix
    m = 600_000
    r, s, c = idx.search(q[:96], terms[:96], syn.NOW_TICKS, 10, candidate_limit=m)
    for b0 in range(0, 8, 4):
        r4, s4, _ = idx.search(q[b0:b0 + 4], terms[b0:b0 + 4], syn.NOW_TICKS, 10, candidate_limit=m)
        assert np.array_equal(r[b0:b0 + 4], r4) and np.array_equal(s[b0:b0 + 4], s4)
    assert (r < m).all()


def test_large_topk_through_the_two_stage_pass_at_1m(big):
    """topK = 50 (k' = 64, the widest the selection kernels carry) and topK = 60 (generic sort path) agree with
    the exact kernel for a single query and for a batch."""
    P, syn, idx, n, dim = big
    B = 12
    q = syn.query_vectors(5000, B, dim, n, "cuda:0")
    terms = [P.text.query_terms(t) for t in syn.query_texts(5000, B, n)]
    for topk in (50, 60):
        idx.set_option("two_stage", 0)
        want = [idx.search(q[b:b + 1], terms[b:b + 1], syn.NOW_TICKS, topk, candidate_limit=n) for b in range(3)]
        idx.set_option("two_stage", 1)
        got1 = [idx.search(q[b:b + 1], terms[b:b + 1], syn.NOW_TICKS, topk, candidate_limit=n) for b in range(3)]
        gotB = idx.search(q, terms, syn.NOW_TICKS, topk, candidate_limit=n)
        for b in range(3):
            assert want[b][2][0] == topk
            assert all(np.array_equal(x, y) for x, y in zip(want[b], got1[b])), (topk, b)
            assert np.array_equal(gotB[0][b], want[b][0][0]) and np.array_equal(gotB[1][b], want[b][1][0]), (topk, b)
