"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports
every symbol include/*.h declares, the host string library agrees with the oracle,
and compute entry points fail loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from helpers import ROOT, has_gpu, orc, pkg


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(orrh?_[a-z0-9_]+)\s*\(", txt)))


def test_hip_library_exports_every_declared_symbol():
    P = pkg()
    names = _declared("omnirecall_hip.h")
    assert sorted(names) == sorted(P.native.EXPORTED_HIP_SYMBOLS)
    for n in names:
        assert hasattr(P.native.hip, n), n
    assert P.native.hip.orr_abi_version() == 1


def test_host_library_exports_every_declared_symbol():
    P = pkg()
    names = _declared("omnirecall_host.h")
    for n in names:
        assert hasattr(P.native.host, n), n
    assert set(P.native.EXPORTED_HOST_SYMBOLS) <= set(names)


def test_candidate_record_layout_matches_header():
    P = pkg()
    assert C.sizeof(P.native.OrrCandidate) == 56 and P.CAND_DTYPE.itemsize == 56
    assert C.sizeof(P.native.OrrConfig) == 32


def test_no_cpu_fallback_without_gpu():
    if has_gpu():
        pytest.skip("a GPU is present")
    P = pkg()
    with pytest.raises(P.OrrError) as ei:
        P.RecallIndex(dim=8)
    assert ei.value.code == P.native.ORR_EDEVICE
    assert "no CPU path" in str(ei.value) or "HIP" in str(ei.value)


def test_argument_errors_map_to_einval():
    P = pkg()
    h = C.c_void_p()
    assert P.native.hip.orr_index_create(None, C.byref(h)) == P.native.ORR_EINVAL
    assert b"null" in P.native.hip.orr_last_error()
    cfg = P.native.OrrConfig(4, 0, 8, 0, 0, 0)       # wrong struct_size
    assert P.native.hip.orr_index_create(C.byref(cfg), C.byref(h)) == P.native.ORR_EINVAL
    assert P.native.hip.orr_index_seal(None) == P.native.ORR_EINVAL
    assert P.native.hip.orr_index_rows(None) == 0


QUERIES = ["azure", "what is the kubernetes", "  Azure\tAZURE azure  Cosmos ", "the of and", "a b c　d",
           "x\x1cy", "ÄRGER ärger", "İstanbul i̇stanbul", "What backend did we choose?", "   ", "",
           "ǅ Ǆ ǆ ΑΒΓ Σίσυφος ЖЁ", "tab sep line"]


@pytest.mark.parametrize("q", QUERIES)
def test_host_query_terms_match_oracle(q):
    P = pkg()
    assert P.text.query_terms(q) == orc.query_terms(q)
    assert P.text.is_blank(q) == orc.is_blank(q)


def test_host_lower_snippet_round_match_oracle():
    P = pkg()
    rng = np.random.default_rng(2)
    pool = "aAbBzZ 09_?ÄÖÜßéÉİıǅΣσςЖжԱա\n\r\t　𐐀𐐨" 
    for _ in range(300):
        s = "".join(rng.choice(list(pool), size=int(rng.integers(0, 260))))
        assert P.text.lower_invariant(s) == orc.lower_invariant(s)
        assert P.text.build_snippet(s, 180) == orc.snippet(s, 180)
        assert P.text.build_snippet(s, 7) == orc.snippet(s, 7)
    for x in list(rng.standard_normal(200)) + [0.30000000000000004, 0.12345, 0.12355, 2.5e-4, 3.5e-4, 1e17, -0.00025]:
        assert P.text.round4(float(x)) == orc.round4(float(x))


def test_merge_candidates_is_host_only_and_exact():
    """orr_merge_candidates needs no GPU: feed it records made from oracle pieces and
    check the exact finish + ranking against the oracle's own search."""
    P = pkg()
    rng = np.random.default_rng(9)
    n, d, B, k = 200, 16, 3, 7
    emb = rng.standard_normal((n, d)).astype(np.float32)
    created = np.sort(639144000000000000 - rng.integers(0, 300 * 864000000000, n))[::-1].astype(np.int64)
    contents = [" ".join(rng.choice(["alpha", "beta", "gamma", "delta"], 4)) for _ in range(n)]
    corpus = orc.OracleCorpus(emb, created, contents)
    qs = rng.standard_normal((B, d)).astype(np.float32)
    texts = ["alpha delta", "the gamma", "zeta"]
    terms = [P.text.query_terms(t) for t in texts]
    # two "shards" holding all of their rows as candidates (kprime = shard size)
    half = n // 2
    recs = np.zeros((2, B, half + 1), dtype=P.CAND_DTYPE)
    for s, (lo, hi) in enumerate([(0, half), (half, n)]):
        for b in range(B):
            for i, r in enumerate(range(lo, hi)):
                m = sum(1 for t in terms[b] if t in P.text.lower_invariant(contents[r]))
                recs[s, b, i] = (0.0, orc.dot(qs[b], emb[r]), orc.dot(emb[r], emb[r]), created[r], r, r, m,
                                 P.native.ORR_CAND_DOT_EXACT)
            recs[s, b, half] = (-np.inf, 0, 0, 0, -1, hi - lo, hi - lo, P.native.ORR_CAND_TRAILER)
    rows, scores, counts, unc = P.merge_candidates(recs, d, qs, terms, 639144000000000000, k)
    assert unc == 0
    for b in range(B):
        orow, osc, _ = corpus.search(qs[b], texts[b], 639144000000000000, k, candidate_limit=n)
        assert list(rows[b]) == list(orow) and np.array_equal(scores[b], osc) and counts[b] == k


def test_large_merges_through_the_host_thread_pool_equal_query_by_query_merges():
    """A merge of many records is shared out over the library's persistent host threads (queries are independent),
    the exact norms of host-resident queries likewise; a second caller arriving meanwhile runs its work itself.
    Same records query by query (single-threaded path), in one call (pool), and from four threads at once."""
    import threading
    P = pkg()
    rng = np.random.default_rng(10)
    B, S, kp, d, k = 700, 3, 32, 384, 10                  # 700 x 384 floats: the norms go through the pool as well
    now = 639144000000000000
    qs = rng.standard_normal((B, d)).astype(np.float32)
    qs[5] = 0.0
    terms = [[b"alpha", b"beta"][: b % 3] for b in range(B)]
    recs = np.zeros((S, B, kp + 1), dtype=P.CAND_DTYPE)
    recs["dot"] = rng.standard_normal((S, B, kp + 1)) * 5
    recs["norm_b"] = rng.uniform(0.5, 30.0, (S, B, kp + 1))
    recs["created_ticks"] = now - rng.integers(0, 300 * 864000000000, (S, B, kp + 1))
    pos = np.arange(S * (kp + 1)).reshape(S, 1, kp + 1) + np.zeros((1, B, 1), dtype=np.int64)
    recs["row_id"] = pos * 7 + 1
    recs["order_key"] = pos
    recs["matches"] = rng.integers(0, 3, (S, B, kp + 1)) % (np.array([len(t) for t in terms])[None, :, None] + 1)
    recs["flags"] = P.native.ORR_CAND_DOT_EXACT
    for s_, i_ in ((0, 3), (1, 4)):                       # two records that would win query 7 ...
        recs["dot"][s_, 7, i_] = 1e4
        recs["norm_b"][s_, 7, i_] = 1.0
    recs["flags"][0, 7, 3] |= P.native.ORR_CAND_DEAD      # ... one of them of a deleted row: dropped by the finish
    for s in range(S):
        for b in range(B):
            recs[s, b, kp] = (-np.inf, 0, 0, 0, -1, kp, kp, P.native.ORR_CAND_TRAILER)
    whole = P.merge_candidates(recs, d, qs, terms, now, k)
    assert whole[3] == 0 and (whole[2] == k).all()
    for b in list(range(0, B, 37)) + [5, 7, B - 1]:
        one = P.merge_candidates(recs[:, b:b + 1], d, qs[b:b + 1], terms[b:b + 1], now, k)
        assert np.array_equal(one[0][0], whole[0][b]) and np.array_equal(one[1][0], whole[1][b]), b
    dead_id, twin_id = int(recs["row_id"][0, 7, 3]), int(recs["row_id"][1, 7, 4])
    assert dead_id not in set(int(r) for r in whole[0][7]) and int(whole[0][7, 0]) == twin_id
    results = [None] * 4

    def work(i):
        results[i] = P.merge_candidates(recs, d, qs, terms, now, k)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for r in results:
        assert all(np.array_equal(x, y) for x, y in zip(r[:3], whole[:3])) and r[3] == 0


def test_host_thread_pool_survives_a_fork():
    """A forked child inherits the pool object but none of its threads: large merges there run in the caller
    instead of waiting for workers that do not exist."""
    import os
    import signal
    P = pkg()
    rng = np.random.default_rng(11)
    B, kp, d, k = 400, 32, 8, 5
    now = 639144000000000000
    qs = rng.standard_normal((B, d)).astype(np.float32)
    terms = [[] for _ in range(B)]
    recs = np.zeros((1, B, kp + 1), dtype=P.CAND_DTYPE)
    recs["dot"] = rng.standard_normal((1, B, kp + 1))
    recs["norm_b"] = 2.0
    recs["created_ticks"] = now
    recs["row_id"] = np.arange(kp + 1)[None, None, :]
    recs["order_key"] = np.arange(kp + 1)[None, None, :]
    recs["flags"] = P.native.ORR_CAND_DOT_EXACT
    for b in range(B):
        recs[0, b, kp] = (-np.inf, 0, 0, 0, -1, kp, kp, P.native.ORR_CAND_TRAILER)
    want = P.merge_candidates(recs, d, qs, terms, now, k)          # starts the pool in this process
    pid = os.fork()
    if pid == 0:                                                   # child: no pytest machinery from here on
        try:
            signal.alarm(20)                                       # a wait for missing workers would end here
            got = P.merge_candidates(recs, d, qs, terms, now, k)
            ok = all(np.array_equal(x, y) for x, y in zip(got[:3], want[:3]))
            os._exit(0 if ok else 3)
        except BaseException:
            os._exit(4)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, status
