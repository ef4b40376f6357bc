"""Parity at BASELINE.json's headline sizes, which cross boundaries the 1M tests never reach (row positions and tile
offsets beyond 2^31 bytes x 50, 64 sample segments, tens of GB of shadow):

  C3        10M x 3072 on one MI355X, 256 queries per batch (configs[2]).
  C4 / C5   the per-GPU shape of the 100M-row configs: 12.5M rows with row_base = 87.5M, through
            orr_search_shard + orr_merge_candidates as the 8-GPU job runs them.

Size-independent properties (planted row wins, batched = exact kernel, limit prefix = oracle) plus which kernels ran
and that nothing had to be repeated."""
import importlib
import os

import numpy as np
import pytest

from helpers import orc, pkg

pytestmark = pytest.mark.gpu


def _build(P, syn, rows, dim, n_total, row_base=0):
    import torch
    idx = P.RecallIndex(dim=dim, capacity_rows=rows, row_base=row_base)
    step = 32768
    for r0 in range(0, rows, step):
        m = min(step, rows - r0)
        g0 = row_base + r0
        pool, off = syn.contents(g0, m, "cuda:0")
        idx.append(syn.embeddings(g0, m, dim, "cuda:0"), syn.created_ticks(g0, m, n_total, "cuda:0"), pool, off)
    torch.cuda.synchronize()
    idx.seal()
    return idx


@pytest.fixture(scope="module")
def c3():
    P = pkg()
    syn = importlib.import_module("omni_recall_rag_amd.synthetic")
    n, dim = 10_000_000, 3072
    idx = _build(P, syn, n, dim, n)
    yield P, syn, idx, n, dim
    idx.close()


def test_c3_batch_of_256_planted_rows_win_and_equal_the_exact_kernel(c3):
    P, syn, idx, n, dim = c3
    B = 256
    q = syn.query_vectors(0, B, dim, n, "cuda:0")
    texts = syn.query_texts(0, B, n)
    terms = [P.text.query_terms(t) for t in texts]
    idx.set_profiling(True)
    idx.reset_search_stats()
    rows, scores, counts = idx.search(q, terms, syn.NOW_TICKS, 10, candidate_limit=n)
    st, ss = idx.kernel_stats(), idx.search_stats()
    idx.set_profiling(False)
    assert list(rows[:, 0]) == syn.planted_rows(0, B, n)
    assert (counts == 10).all() and (np.diff(scores, axis=1) <= 0).all()
    # the int8 two-stage pass ran once, nothing was repeated, no survivors' buffer overflowed
    # (10M rows: the screening GEMM runs as four row ranges, one launch each; one pass, no repeat)
    assert st["screen_i8_fused"]["launches"] == 4 and st["screen_i8_prefix"]["launches"] == 1, sorted(st)
    assert "dot_exact" not in st and "gemm_dot_bf16x3" not in st, sorted(st)
    assert ss["passes"] == 1 and ss["requeried"] == 0 and ss["overflowed_queries"] == 0 and ss["survivors_max"] < 8192, ss
    # which workgroup multiplies which output tiles must not show in what survives the screen (a request stream that crossed
    # to its workgroup's next tile wrongly once did, by 7 pairs in 35,000, behind identical top-10 lists)
    os.environ["ORR_SCREEN_GRID"] = "64"
    try:
        idx.reset_search_stats()
        rows_g, scores_g, counts_g = idx.search(q, terms, syn.NOW_TICKS, 10, candidate_limit=n)
        ss_g = idx.search_stats()
    finally:
        os.environ.pop("ORR_SCREEN_GRID", None)
    assert ss_g["survivors_total"] == ss["survivors_total"] and ss_g["survivors_max"] == ss["survivors_max"], (ss_g, ss)
    assert np.array_equal(rows_g, rows) and np.array_equal(scores_g, scores) and np.array_equal(counts_g, counts)
    # 32 of the queries again through the reference-arithmetic kernel over every fp32 row: identical rows, order and fp64 scores
    idx.set_option("two_stage", 0)
    try:
        for b0 in range(0, 32, 4):
            r, s, c = idx.search(q[b0:b0 + 4], terms[b0:b0 + 4], syn.NOW_TICKS, 10, candidate_limit=n)
            assert np.array_equal(r, rows[b0:b0 + 4]) and np.array_equal(s, scores[b0:b0 + 4]), b0
    finally:
        idx.set_option("two_stage", 1)
    # the same queries one at a time take the streaming screen and agree
    for b in (0, 100, 255):
        r1, s1, _ = idx.search(q[b:b + 1], terms[b:b + 1], syn.NOW_TICKS, 10, candidate_limit=n)
        assert np.array_equal(r1[0], rows[b]) and np.array_equal(s1[0], scores[b])


def test_c3_candidate_limits(c3):
    """candidate_limit = 20,000 (small enough for the oracle), and = n - 1 (the prefix ends one row short of the corpus)."""
    P, syn, idx, n, dim = c3
    m = 20_000
    emb = syn.embeddings(0, m, dim).numpy()
    created = syn.created_ticks(0, m, n).numpy()
    pool, off = syn.contents(0, m)
    cor = orc.OracleCorpus(emb, created, (pool.numpy(), off.numpy()))
    B = 8
    q = syn.query_vectors(0, B, dim, n, "cuda:0")
    texts = syn.query_texts(0, B, n)
    terms = [P.text.query_terms(t) for t in texts]
    rows, scores, counts = idx.search(q, terms, syn.NOW_TICKS, 10, candidate_limit=m)
    qh = q.cpu().numpy()
    for b in range(B):
        orow, osc, _ = cor.search(qh[b], texts[b], syn.NOW_TICKS, 10, candidate_limit=m, threads=8)
        assert list(rows[b, :counts[b]]) == list(orow) and np.array_equal(scores[b, :counts[b]], osc), b
    full = idx.search(q, terms, syn.NOW_TICKS, 10, candidate_limit=n)
    short = idx.search(q, terms, syn.NOW_TICKS, 10, candidate_limit=n - 1)
    assert (short[0] < n - 1).all()
    for b in range(B):                                      # dropping the oldest row changes nothing unless it was a result
        if n - 1 not in full[0][b]:
            assert np.array_equal(short[0][b], full[0][b]) and np.array_equal(short[1][b], full[1][b]), b
    # a limit in the middle: every result lies in the prefix, and the first row past it is never returned
    mid = 6_123_457
    r, s, c = idx.search(q, terms, syn.NOW_TICKS, 10, candidate_limit=mid)
    assert (r < mid).all() and (c == 10).all()
