"""The survivors' exact dots when the re-score adds out of order (finish_survivors, one wave per survivor).

RecallSearchService.cs:77-82 adds the products one after the other into a double.  The re-score evaluates that sum in
another order where it can PROVE that no order of additions rounds (every product a multiple of 2^g, the sum of magnitudes
below 2^(g+53)), walks the row in slabs of 256 columns against the running sum where the whole row does not pass, and adds
a slab that does not pass in the reference's order.  Unit-scale data takes the first path for ~94 % of the rows; this file
feeds it rows and queries whose products span tens of decades, exact zeros of both signs, subnormal products, infinities
and NaNs, and compares every record's dot with the oracle's sequential sum BIT FOR BIT (and the ranked scores with the
oracle's), for the batch sizes that take groups of 16 and of 64 survivors.
"""
import numpy as np
import pytest

from helpers import NOW, DAY, orc, pkg

pytestmark = pytest.mark.gpu


def _bits(x):
    return np.asarray(x, dtype=np.float64).view(np.int64)


def _spread(rng, shape, decades):
    v = rng.standard_normal(shape).astype(np.float32)
    if decades:
        v = (v * np.power(np.float32(10.0), rng.uniform(-decades, decades, shape).astype(np.float32))).astype(np.float32)
    return v


@pytest.mark.parametrize("dim,decades", [(3072, 0), (3072, 3), (1024, 12), (256, 18), (768, 6)])
def test_survivor_dots_equal_the_sequential_sum_bit_for_bit(dim, decades):
    P = pkg()
    rng = np.random.default_rng(9000 + dim + decades)
    n = 200_000                                                            # 49 segments: the two-stage pass applies
    emb = rng.standard_normal((n, dim), dtype=np.float32)
    odd = rng.choice(n, 4000, replace=False)
    emb[odd[:2000]] = _spread(rng, (2000, dim), max(decades, 2))           # rows whose own coordinates span decades
    emb[odd[2000:2500], ::3] = 0.0
    emb[odd[2500:3000], 1::2] = -0.0
    emb[odd[3000:3200]] *= np.float32(1e-20)                               # with a 1e-20 query: subnormal and zero products
    emb[odd[3200:3400]] *= np.float32(1e19)
    created = np.sort(NOW - rng.integers(0, 300 * DAY, n))[::-1].astype(np.int64)
    idx = P.RecallIndex(dim=dim)
    for r0 in range(0, n, 50_000):
        idx.append(emb[r0:r0 + 50_000], created[r0:r0 + 50_000], [b"alpha beta"] * min(50_000, n - r0))
    idx.seal()
    B = 70
    qs = _spread(rng, (B, dim), decades)
    qs[0] = emb[odd[10]]                                                   # wide-range row as the query: squares of its coordinates
    qs[1] = emb[odd[2100]]
    qs[2] = emb[odd[2600]]
    qs[3] = emb[odd[3100]]                                                 # 1e-20 scale: products down to the subnormals
    qs[4] = emb[odd[3300]] * np.float32(1e19)                              # products up to inf
    qs[5] = 0.0
    qs[6] = -0.0
    qs[7, :] = 0.0
    qs[7, dim - 1] = 1.0                                                   # one product in the last slab only
    qs[8] = rng.standard_normal(dim).astype(np.float32)
    qs[8, 5] = np.float32(3e-30)                                           # one tiny product early in an otherwise plain row
    qs[9] = rng.standard_normal(dim).astype(np.float32)
    qs[9, dim // 2] = np.float32(1e25)                                     # one huge product in the middle: later slabs round
    terms = [[b"alpha"]] * B
    kp = 64
    checked = 0
    for b0, nb in ((0, B), (0, 3), (3, 7), (8, 2)):
        idx.set_profiling(True)
        rec = idx.search_shard(qs[b0:b0 + nb], terms[b0:b0 + nb], NOW, kp, n)
        st = idx.kernel_stats()
        idx.set_profiling(False)
        assert "finish_survivors" in st, sorted(st)
        for b in range(nb):
            for c in rec[b, :kp]:
                if c["row_id"] < 0 or not (c["flags"] & P.native.ORR_CAND_DOT_EXACT):
                    continue
                r = int(c["order_key"])
                want = orc.dot(qs[b0 + b], emb[r])
                assert _bits(c["dot"]) == _bits(want) or (np.isnan(c["dot"]) and np.isnan(want)), \
                    f"dim {dim}, decades {decades}, batch [{b0}, +{nb}), query {b0 + b}, row {r}: {c['dot']!r} != {want!r}"
                checked += 1
    assert checked > 64 * 40
    corpus = orc.OracleCorpus(emb, created, ["alpha beta"] * n)
    rows, scores, counts = idx.search(qs[:12], terms[:12], NOW, 10, candidate_limit=n)
    for b in range(12):
        orow, osc, _ = corpus.search(qs[b], "alpha", NOW, 10, candidate_limit=n, threads=8)
        assert list(rows[b, :counts[b]]) == list(orow), b
        a, o = scores[b, :counts[b]], np.asarray(osc)
        assert ((a == o) | (np.isnan(a) & np.isnan(o))).all(), b
    idx.close()
