"""Parity at the per-GPU shape of BASELINE.json's 100M-row configs (C4 / C5): 12.5M rows x 3072 with row_base = 87.5M
(byte offsets beyond 2^37, tile indices beyond what 1M rows reach; 153.6 GB of rows + 38.4 GB of int8 shadow on one MI355X), through
orr_search_shard + orr_merge_candidates as every rank of the 8-GPU job runs them.  (Its own module: the 10M-row
fixture of test_gpu_scale_c3.py must be gone before this one fits.)"""
import importlib

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
def c5_shard():
    P = pkg()
    syn = importlib.import_module("omni_recall_rag_amd.synthetic")
    rows, dim, n_total = 12_500_000, 3072, 100_000_000
    row_base = 87_500_000                                   # the last of eight shards
    idx = _build(P, syn, rows, dim, n_total, row_base=row_base)
    yield P, syn, idx, rows, dim, n_total, row_base
    idx.close()


def _queries_planted_in(syn, b0, B, dim, lo, hi, device):
    """q_b = e[r_b] + 0.25 noise with r_b inside [lo, hi) (global candidate positions)."""
    import torch
    r = torch.tensor([lo + p for p in syn.planted_rows(b0, B, hi - lo)], dtype=torch.int64, device=device)
    c = torch.arange(dim, dtype=torch.int64, device=device).unsqueeze(0)
    b = torch.arange(b0, b0 + B, dtype=torch.int64, device=device).unsqueeze(1)
    e = syn._unit_fixed(syn.splitmix64((r.unsqueeze(1) * dim + c) ^ syn.SEED))
    noise = syn._unit_fixed(syn.splitmix64((b * dim + c) ^ (syn.SEED + 1)))
    return (e + 0.25 * noise).contiguous(), [int(x) for x in r.cpu()]


def test_c5_shard_shape_through_search_shard_and_merge(c5_shard):
    """12.5M rows at row_base 87.5M: the records' order keys and the default row ids exceed 2^31; the shard search +
    host merge (what every rank of the 8-GPU job does) equals orr_search_batch on the same shard, planted rows win."""
    P, syn, idx, rows, dim, n_total, row_base = c5_shard
    B, kprime = 64, 32
    q, planted = _queries_planted_in(syn, 0, B, dim, row_base, row_base + rows, "cuda:0")
    terms = [[] for _ in range(B)]                          # C4: cosine (+ recency) only
    idx.set_profiling(True)
    idx.reset_search_stats()
    recs = idx.search_shard(q, terms, syn.NOW_TICKS, kprime, candidate_limit=n_total)
    st, ss = idx.kernel_stats(), idx.search_stats()
    idx.set_profiling(False)
    assert "screen_i8_fused" in st, sorted(st)
    # cosine-only scores have no steps: thousands of pairs sit within the int8 bound of the floor.  Whatever overflowed its
    # buffer was repeated INSIDE the call with larger buffers (and the next search samples a larger prefix); the caller
    # sees finished records
    assert not (recs["flags"][:, kprime] & 4).any(), ss                        # no ORR_CAND_OVERFLOW left
    print("c5 shard, 64 cosine-only queries, k' = 32:", ss)
    assert int(recs["order_key"][:, 0].min()) >= row_base and int(recs["row_id"][:, 0].max()) < row_base + rows
    assert (recs["order_key"][:, kprime] == rows).all()                       # trailer: rows that took part on this shard
    qh = q.cpu().numpy()
    mrows, mscores, mcounts, unc = P.merge_candidates(recs[None], dim, qh, terms, syn.NOW_TICKS, 10)
    assert unc == 0 and list(mrows[:, 0]) == planted and (mcounts == 10).all()
    rows2, scores2, counts2 = idx.search(q, terms, syn.NOW_TICKS, 10, candidate_limit=n_total)
    assert np.array_equal(rows2, mrows) and np.array_equal(scores2, mscores)
    # the same shard placed beyond 2^32 in the global order (a corpus of billions of rows): keys and the limit are 64-bit
    far = 5_000_000_000
    P.native.check(P.native.hip.orr_index_set_row_base(idx._h, far))
    try:
        recs_far = idx.search_shard(q[:16], terms[:16], syn.NOW_TICKS, kprime, candidate_limit=far + rows)
        assert np.array_equal(recs_far["order_key"][:, :kprime] - far, recs["order_key"][:16, :kprime] - row_base)
        assert np.array_equal(recs_far["dot"][:, :kprime], recs["dot"][:16, :kprime])
        none = idx.search_shard(q[:2], terms[:2], syn.NOW_TICKS, kprime, candidate_limit=far)      # the limit ends in front of the shard
        assert (none["order_key"][:, kprime] == 0).all() and (none["matches"][:, kprime] == 0).all()
    finally:
        P.native.check(P.native.hip.orr_index_set_row_base(idx._h, row_base))
    # a global limit that ends inside this shard: only its first 1,000,000 rows take part
    lim = row_base + 1_000_000
    r3, s3, c3_ = idx.search(q[:8], terms[:8], syn.NOW_TICKS, 10, candidate_limit=lim)
    assert (r3 < lim).all() and (r3 >= row_base).all()
    # one query with terms (the C5 form, hybrid): streaming screen, still exact against the exact kernel
    texts = [f"the {syn.vocab_word(int(t)).decode()}" for t in syn.token_ids(planted[0], 1)[0][:2]]
    t1 = [P.text.query_terms(" ".join(texts))]
    a = idx.search(q[:1], t1, syn.NOW_TICKS, 10, candidate_limit=n_total)
    idx.set_option("two_stage", 0)
    try:
        b = idx.search(q[:1], t1, syn.NOW_TICKS, 10, candidate_limit=n_total)
    finally:
        idx.set_option("two_stage", 1)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[0][0, 0] == planted[0]


def test_c5_shard_shape_hybrid_batch_of_300_does_not_depend_on_the_workgroup_count(c5_shard):
    """The C5 form on this shard: 300 hybrid queries (two query tiles per row tile, four row ranges of the 16 x 16 x 64
    screening GEMM).  What survives the screen, and the results, are the same with 64 persistent workgroups as with one per
    CU; eight of the queries equal the reference-arithmetic kernel over every row."""
    import os
    P, syn, idx, rows, dim, n_total, row_base = c5_shard
    B = 300
    q, planted = _queries_planted_in(syn, 1000, B, dim, row_base, row_base + rows, "cuda:0")
    terms = []
    for b in range(B):
        toks = syn.token_ids(planted[b], 1)[0][:2]
        terms.append(P.text.query_terms(" ".join(syn.vocab_word(int(t)).decode() for t in toks)) if b % 4 else [])
    idx.set_profiling(True)
    idx.reset_search_stats()
    r_a, s_a, c_a = idx.search(q, terms, syn.NOW_TICKS, 10, candidate_limit=n_total)
    st, ss_a = idx.kernel_stats(), idx.search_stats()
    idx.set_profiling(False)
    assert st["screen_i8_fused"]["launches"] >= 4, sorted(st)
    assert list(r_a[:, 0]) == planted
    os.environ["ORR_SCREEN_GRID"] = "64"
    try:
        idx.reset_search_stats()
        r_g, s_g, c_g = idx.search(q, terms, syn.NOW_TICKS, 10, candidate_limit=n_total)
        ss_g = idx.search_stats()
    finally:
        os.environ.pop("ORR_SCREEN_GRID", None)
    assert ss_g["survivors_total"] == ss_a["survivors_total"] and ss_g["survivors_max"] == ss_a["survivors_max"], (ss_g, ss_a)
    assert np.array_equal(r_g, r_a) and np.array_equal(s_g, s_a) and np.array_equal(c_g, c_a)
    idx.set_option("two_stage", 0)
    try:
        for b0 in (0, 296):
            r, s, c = idx.search(q[b0:b0 + 4], terms[b0:b0 + 4], syn.NOW_TICKS, 10, candidate_limit=n_total)
            assert np.array_equal(r, r_a[b0:b0 + 4]) and np.array_equal(s, s_a[b0:b0 + 4]), b0
    finally:
        idx.set_option("two_stage", 1)
