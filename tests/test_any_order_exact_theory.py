"""The arithmetic fact the survivors' re-score relies on (orr_gemm.hip: any_order_is_exact, exact_dot_by_slabs), on the CPU.

RecallSearchService.cs:77-82 adds fp32 products one after the other into a double.  The device adds them in another order
where it can prove that no order rounds: every product is a float, a multiple of 2^(e-23); with g the smallest such exponent
every partial sum of every order is a multiple of 2^g and none exceeds M = sum |p| in magnitude; a multiple of 2^g below
2^(g+53) is a double.  This file restates the test and the slab walk in numpy / Python and checks, on data from unit scale
to 36 decades of spread with zeros, subnormals, infinities and NaNs:
  * whenever the test passes, the sequential sum, the sum in random orders and the exact sum (math.fsum) are the same double;
  * the slab walk (slabs of 256, blocks of 64, single additions where a block does not pass) returns the sequential sum's
    bits for every input, passing or not.
(The device code itself is compared with the oracle bit for bit in tests/test_gpu_rescore_any_order.py.)
"""
import math

import numpy as np
import pytest


def _g_of_products(p32):
    """Smallest e - 23 over the nonzero products (None: all zero); subnormals count as exponent -126."""
    bits = np.abs(p32).view(np.uint32)
    nz = bits != 0
    if not nz.any():
        return None
    field = np.maximum((bits[nz] >> 23).astype(np.int64), 1)
    return int(field.min()) - 127 - 23


def _g_of_double(s):
    if s == 0.0 or not math.isfinite(s):
        return None
    m, e = math.frexp(abs(s))                      # s = m 2^e, 0.5 <= m < 1
    mant = int(m * (1 << 53))                      # 53-bit integer mantissa
    tz = (mant & -mant).bit_length() - 1
    return e - 53 + tz


def any_order_is_exact(s, p32):
    """The device's test: running sum s (a double) and the float products p32 can be added in any order without rounding."""
    gp, gs = _g_of_products(p32), _g_of_double(s)
    if not math.isfinite(s):
        return False
    gs_all = [g for g in (gp, gs) if g is not None]
    if not gs_all:
        return True
    g = min(gs_all)
    M = abs(s) + float(np.abs(p32).astype(np.float64).sum())
    if not math.isfinite(M):
        return False
    e = min(g + 53, 1023)
    return M <= math.ldexp(1.0, e) * (1.0 - 2.0 ** -20)


def sequential(p32, s=0.0):
    for x in p32.astype(np.float64):
        s = s + float(x)
    return s


def by_slabs(p32):
    """exact_dot_by_slabs restated: returns (sum, number of single additions that were needed)."""
    s, singles = 0.0, 0
    for k in range(0, len(p32), 256):
        slab = p32[k:k + 256]
        if any_order_is_exact(s, slab):
            s = s + math.fsum(float(x) for x in slab)           # any order: here the exact sum, which is a double
            continue
        for q in range(0, len(slab), 64):
            blk = slab[q:q + 64]
            if any_order_is_exact(s, blk):
                s = s + math.fsum(float(x) for x in blk)
            else:
                s = sequential(blk, s)
                singles += len(blk)
    return s, singles


def _bits(x):
    return np.float64(x).view(np.int64)


def _same(a, b):
    return _bits(a) == _bits(b) or (math.isnan(a) and math.isnan(b))


def _rows(rng, dim, decades, n):
    q = rng.standard_normal((n, dim)).astype(np.float32)
    e = rng.standard_normal((n, dim)).astype(np.float32)
    if decades:
        q = (q * np.power(np.float32(10.0), rng.uniform(-decades, decades, (n, dim)).astype(np.float32))).astype(np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        return (q * e).astype(np.float32)


@pytest.mark.parametrize("dim,decades", [(3072, 0), (3072, 2), (1024, 6), (256, 18), (768, 1)])
def test_passing_rows_sum_to_the_same_double_in_every_order(dim, decades):
    rng = np.random.default_rng(100 + dim + decades)
    P = _rows(rng, dim, decades, 60)
    passed = 0
    for p in P:
        if not any_order_is_exact(0.0, p):
            continue
        passed += 1
        want = sequential(p)
        assert _same(want, math.fsum(float(x) for x in p))              # no addition of the reference's chain rounded
        for _ in range(3):
            assert _same(want, sequential(p[rng.permutation(dim)]))
        # the device's order: 64 lanes x (4 consecutive columns of every slab), then a butterfly over the lanes
        lanes = [0.0] * 64
        for k in range(0, dim, 256):
            for lane in range(64):
                for x in p[k + 4 * lane:k + 4 * lane + 4]:
                    lanes[lane] += float(x)
        off = 32
        while off:
            lanes = [lanes[i] + lanes[i ^ off] for i in range(64)]
            off >>= 1
        assert all(_same(want, v) for v in lanes)
    if decades == 0:
        assert passed >= 50                                              # unit-scale rows: ~94 % pass as a whole
    if decades >= 6:
        assert passed <= 5


@pytest.mark.parametrize("dim,decades", [(3072, 0), (3072, 3), (1024, 12), (256, 18), (512, 30)])
def test_slab_walk_returns_the_sequential_sum_for_any_input(dim, decades):
    rng = np.random.default_rng(200 + dim + decades)
    P = _rows(rng, dim, decades, 40)
    P[1, ::3] = 0.0
    P[2, 1::2] = -0.0
    P[3] = 0.0
    P[4] = -0.0                                                          # the reference's sum starts at +0.0 and stays there
    P[5, dim // 2] = np.float32(1e30)                                    # later slabs round at every binade they climb
    P[6, 5] = np.float32(3e-38)
    P[7, :4] = np.float32([1e-45, -1e-45, 2e-45, 1e-40])                 # subnormal floats
    P[8, 100] = np.inf
    P[9, 7] = np.nan
    P[10, 3] = np.inf
    P[10, 900 % dim] = -np.inf
    singles_total = 0
    for i, p in enumerate(P):
        want = sequential(p)
        got, singles = by_slabs(p)
        singles_total += singles
        assert _same(want, got), (i, want, got)
    assert _bits(by_slabs(P[4])[0]) == _bits(0.0)                        # +0.0, not -0.0
    if decades >= 12:
        assert singles_total > 0                                         # the single additions are exercised


def test_the_test_itself_is_tight_enough_to_matter_and_never_wrong_near_the_limit():
    """Rows built to sit just inside and just outside the limit: inside, all orders agree; outside, a row exists whose
    orders differ (so the test is not vacuous)."""
    rng = np.random.default_rng(7)
    big = np.float32(1.0)
    differ = 0
    for shift in range(20, 40):
        tiny = np.float32(2.0 ** -shift) * np.float32(1.0 + 2.0 ** -23)  # full mantissa: ulp = 2^(-shift-23)
        p = np.array([big] * 200 + [tiny] * 5 + [big] * 200, dtype=np.float32)
        ok = any_order_is_exact(0.0, p)
        a, b = sequential(p), sequential(p[::-1].copy())
        c = sequential(p[rng.permutation(len(p))])
        if ok:
            assert _same(a, b) and _same(a, c) and _same(a, math.fsum(float(x) for x in p)), shift
        else:
            differ += int(not (_same(a, math.fsum(float(x) for x in p))))
    assert differ > 0
