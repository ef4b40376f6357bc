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


def _hybrid_terms(P, syn, planted, every=1):
    terms = []
    for b, row in enumerate(planted):
        toks = syn.token_ids(row, 1)[0][[3, 40, 77]]
        words = " ".join(syn.vocab_word(int(t)).decode() for t in toks)
        terms.append(P.text.query_terms(f"What is the {words}") if b % every == 0 or every == 1 else [])
    return terms


def test_c5_per_gpu_shape_batch_of_1024_hybrid_queries(c5_shard):
    """BASELINE.json configs[4] as one of its eight GPUs sees it: 12.5M rows x 3072, ONE batch of 1024 full-hybrid queries
    (four 256-query tiles per row tile, eight row ranges of the 16 x 16 x 64 screening GEMM).  One pass, nothing overflows,
    nothing falls back to the exact pass; planted rows win; queries from all four query tiles equal the reference-arithmetic
    kernel over every row; survivors and results do not depend on the number of persistent workgroups."""
    import os
    P, syn, idx, rows, dim, n_total, row_base = c5_shard
    B = 1024
    q, planted = _queries_planted_in(syn, 5000, B, dim, row_base, row_base + rows, "cuda:0")
    terms = _hybrid_terms(P, syn, planted)
    assert all(len(t) == 3 for t in terms)
    idx.search(q, terms, syn.NOW_TICKS, 10, candidate_limit=n_total)         # (workspaces and the adapted sample settle)
    idx.set_profiling(True)
    idx.reset_search_stats()
    r_a, s_a, c_a = idx.search(q, terms, syn.NOW_TICKS, 10, candidate_limit=n_total)
    st, ss_a = idx.kernel_stats(), idx.search_stats()
    idx.set_profiling(False)
    print("c5 shard, 1024 hybrid queries:", ss_a, {k: round(v["total_ms"], 3) for k, v in st.items()})
    assert st["screen_i8_fused"]["launches"] == 8, {k: v["launches"] for k, v in st.items()}     # eight row ranges, one launch each
    assert "dot_exact" not in st and "gemm_dot_bf16x3" not in st, sorted(st)
    assert ss_a["passes"] == 1 and ss_a["requeried"] == 0 and ss_a["overflowed_queries"] == 0 and ss_a["exact_pass_queries"] == 0, ss_a
    assert ss_a["pass_mode"] == 1, ss_a                                      # the int8 shadow was there (two_stage = 1 ran as such)
    assert list(r_a[:, 0]) == planted and (c_a == 10).all()
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
        for b0 in (0, 252, 256, 508, 764, 1020):                              # four queries each: every query tile, both ends of two of them
            r, s, c = idx.search(q[b0:b0 + 4], terms[b0:b0 + 4], syn.NOW_TICKS, 10, candidate_limit=n_total)
            assert np.array_equal(r, r_a[b0:b0 + 4]) and np.array_equal(s, s_a[b0:b0 + 4]), b0
    finally:
        idx.set_option("two_stage", 1)


def test_c4_per_gpu_shape_one_query_and_256_cosine_only(c5_shard):
    """BASELINE.json configs[3] on one of its eight GPUs: cosine (+ recency) only, one query per step (the streaming int8
    screen over 12.5M rows) and 256 per step, through orr_search_shard + orr_merge_candidates; against orr_search_batch on
    the shard and, for a few queries, the reference-arithmetic kernel."""
    P, syn, idx, rows, dim, n_total, row_base = c5_shard
    kprime = 32
    q, planted = _queries_planted_in(syn, 9000, 256, dim, row_base, row_base + rows, "cuda:0")
    qh = q.cpu().numpy()
    none = [[] for _ in range(256)]
    # ---- one query per step
    idx.set_profiling(True)
    for b in (0, 1, 2):
        idx.reset_search_stats()
        recs = idx.search_shard(q[b:b + 1], none[:1], syn.NOW_TICKS, kprime, candidate_limit=n_total)
        assert not (recs["flags"][:, kprime] & 4).any()
        mrows, mscores, mcounts, unc = P.merge_candidates(recs[None], dim, qh[b:b + 1], none[:1], syn.NOW_TICKS, 10)
        assert unc == 0 and int(mrows[0, 0]) == planted[b] and int(mcounts[0]) == 10
        r1, s1, _ = idx.search(q[b:b + 1], none[:1], syn.NOW_TICKS, 10, candidate_limit=n_total)
        assert np.array_equal(r1, mrows) and np.array_equal(s1, mscores)
    st = idx.kernel_stats()
    idx.set_profiling(False)
    assert "screen_gemv_i8" in st and "screen_i8_fused" not in st, sorted(st)        # the stream, not the GEMM
    # ---- 256 per step (the survivors' buffers and the sampled prefix adapt over the first searches: cosine-only scores have
    # no steps)
    for _ in range(3):
        recs = idx.search_shard(q, none, syn.NOW_TICKS, kprime, candidate_limit=n_total)
    idx.set_profiling(True)
    idx.reset_search_stats()
    recs = idx.search_shard(q, none, syn.NOW_TICKS, kprime, candidate_limit=n_total)
    st, ss = idx.kernel_stats(), idx.search_stats()
    idx.set_profiling(False)
    print("c4 shard, 256 cosine-only queries:", ss)
    assert st["screen_i8_fused"]["launches"] >= 1 and "dot_exact" not in st, sorted(st)
    assert not (recs["flags"][:, kprime] & 4).any(), ss
    assert ss["exact_pass_queries"] == 0, ss
    mrows, mscores, mcounts, unc = P.merge_candidates(recs[None], dim, qh, none, syn.NOW_TICKS, 10)
    assert unc == 0 and list(mrows[:, 0]) == planted and (mcounts == 10).all()
    r2, s2, _ = idx.search(q, none, syn.NOW_TICKS, 10, candidate_limit=n_total)
    assert np.array_equal(r2, mrows) and np.array_equal(s2, mscores)
    idx.set_option("two_stage", 0)
    try:
        for b0 in (0, 252):
            r, s, _ = idx.search(q[b0:b0 + 4], none[:4], syn.NOW_TICKS, 10, candidate_limit=n_total)
            assert np.array_equal(r, mrows[b0:b0 + 4]) and np.array_equal(s, mscores[b0:b0 + 4]), b0
    finally:
        idx.set_option("two_stage", 1)
