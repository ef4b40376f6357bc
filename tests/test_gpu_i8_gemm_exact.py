"""Bit-exact check of the INTEGER work of the int8 screening GEMM (K2j), form by form.

The screening pass multiplies int8 images of queries and rows on the matrix cores in place of the fp32 products of
RecallSearchService.cs:77-82; everything it keeps is re-scored exactly, so an end-to-end parity test only sees a faulty
product when a row of the true top-k drops below the floor -- round 2's first 16 x 16 x 64 kernel multiplied one K-tile of a
neighbouring output tile's rows in every tile but a workgroup's first and passed 104 GPU tests.  Here the kernels run with
an epilogue that stores the raw int32 accumulators (orr_index_screen_i8_dots: same K loop, operand rings, request streams
and persistent walk of the output tiles as the fused launches) and every element must equal numpy's integer product of
the int8 images the library hands back -- for every form of the tile, batch sizes around the query-tile boundaries, a
partial last row tile, and 8 / 64 / one-per-CU persistent workgroups (with 8, every workgroup multiplies 3+ output tiles).
"""
import os

import numpy as np
import pytest

from helpers import NOW, DAY, pkg

pytestmark = pytest.mark.gpu

FORM_NAMES = {0: "eight-wave 32x32x32", 1: "four-wave 32x32x32", 2: "four-wave 16x16x64"}


def _quantise_rows(emb):
    """i8_shadow_kernel restated: se = max|e| / 127 (fp32), ie = rint(e * (1 / se)) clipped to +-127."""
    mx = np.abs(emb).max(axis=1).astype(np.float32)
    se = (mx / np.float32(127.0)).astype(np.float32)
    inv = np.where(se > 0, np.float32(1.0) / np.where(se > 0, se, np.float32(1.0)), np.float32(0.0)).astype(np.float32)
    q = np.rint((emb * inv[:, None]).astype(np.float32))
    return np.clip(q, -127, 127).astype(np.int8)


def _quantise_queries(qs):
    """i8_queries_kernel, first level: s1 = max|q| / 127, iq = rint(q / s1) clipped."""
    mx = np.abs(qs).max(axis=1).astype(np.float32)
    s1 = (mx / np.float32(127.0)).astype(np.float32)
    safe = np.where(s1 > 0, s1, np.float32(1.0))
    a = np.rint((qs / safe[:, None]).astype(np.float32))
    a = np.where(s1[:, None] > 0, a, 0.0)
    return np.clip(a, -127, 127).astype(np.int8)


def _build(P, n, dim, seed):
    rng = np.random.default_rng(seed)
    emb = rng.standard_normal((n, dim)).astype(np.float32)
    # rows that drive the accumulator to its extremes (|I| = D * 127^2 = 49.5M at D = 3072: beyond fp32's 2^24 integers),
    # in the first output tile, in later tiles of a workgroup's walk, and in the partial last tile
    for r in (0, 256 * 9 + 5, n - 3):
        emb[r] = 1.0
        emb[r + 1] = np.where(np.arange(dim) % 2 == 0, 1.0, -1.0)
    emb[7] = 0.0                                                          # a zero row: scale 0, image 0
    emb[300, :] = 0.0
    emb[300, dim - 1] = 5.0                                               # one coordinate only, in the LAST K-tile
    emb[600, :] = 0.0
    emb[600, 0] = -3.0                                                    # ... in the FIRST K-tile
    created = np.sort(NOW - rng.integers(0, 300 * DAY, n))[::-1].astype(np.int64)
    idx = P.RecallIndex(dim=dim)
    idx.append(emb, created, [b"x"] * n)
    idx.seal()
    return idx, emb, rng


def _queries(rng, B, dim, emb):
    qs = rng.standard_normal((B, dim)).astype(np.float32)
    qs[0] = 1.0                                                           # with row 0: the largest accumulator there is
    if B > 1:
        qs[1] = -1.0
    if B > 3:
        qs[2] = 0.0
        qs[3] = 0.0
        qs[3, dim - 64:] = rng.standard_normal(64)                        # one K-tile only (the last)
    if B > 70:
        qs[B - 1] = emb[300]                                              # the batch's last query (a partly filled query tile)
        qs[64] = 0.0
        qs[64, :64] = rng.standard_normal(64)                             # one K-tile only (the first)
    return qs


def _check(idx, qs, form, nt, ie_ref, what, cache=None):
    dots, iq, ie = idx.screen_i8_dots(qs, form, nt_rows=nt)
    assert np.array_equal(ie, ie_ref), f"{what}: the int8 shadow differs from the restated quantisation"
    iq_ref = _quantise_queries(qs)
    assert np.array_equal(iq, iq_ref), f"{what}: the int8 queries differ from the restated quantisation"
    key = qs.shape
    if cache is not None and key in cache:
        want = cache[key]                                                 # (same images as the run that made it: both just checked)
    else:
        # exact in binary64: |sum| <= 3072 * 127^2 < 2^53 (BLAS; numpy's integer matmul is a scalar loop)
        want = (iq_ref.astype(np.float64) @ ie_ref.astype(np.float64).T).astype(np.int64)
        if cache is not None:
            cache[key] = want
    got = dots.astype(np.int64)
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        b, r = bad[0]
        tiles = sorted({int(x) // 256 for x in bad[:, 1]})[:12]
        raise AssertionError(f"{what}: {len(bad)} of {got.size} accumulators differ; first at query {b}, row {r} "
                             f"(row tile {r // 256}): {got[b, r]} != {want[b, r]}; row tiles affected: {tiles}")
    return int(np.abs(want).max())


@pytest.mark.parametrize("dim", [3072, 512])
def test_int8_gemm_accumulators_bit_exact_every_form(dim):
    P = pkg()
    n = 8 * 3 * 256 + 100                                                 # 25 row tiles, the last one of 100 rows; 8 workgroups: 3+ tiles each
    idx, emb, rng = _build(P, n, dim, 20260515 + dim)
    ie_ref = _quantise_rows(emb)
    old = os.environ.get("ORR_SCREEN_GRID")
    biggest = 0
    try:
        for B in (65, 129, 256, 300, 1024):
            qs = _queries(rng, B, dim, emb)
            cache = {}
            for grid in ("8", "64", None):
                if grid is None:
                    os.environ.pop("ORR_SCREEN_GRID", None)
                else:
                    os.environ["ORR_SCREEN_GRID"] = grid
                for form in (0, 1, 2):
                    for nt in ((False, True) if B <= 256 else (False,)):
                        what = f"dim {dim}, {B} queries, {FORM_NAMES[form]}, grid {grid or 'one per CU'}, nt_rows {nt}"
                        biggest = max(biggest, _check(idx, qs, form, nt, ie_ref, what, cache))
    finally:
        if old is None:
            os.environ.pop("ORR_SCREEN_GRID", None)
        else:
            os.environ["ORR_SCREEN_GRID"] = old
    assert biggest == dim * 127 * 127                                     # the extreme really occurred (and came back exactly)
    idx.close()


def test_int8_gemm_accumulators_small_batches_and_short_rows():
    """The eight-wave form with 1, 2, 4 and 8 live query tiles (what batches of up to 64 queries and the sampled prefix
    run), dim 128 (two K-tiles: the non-persistent grid) and dim 3072."""
    P = pkg()
    for dim in (128, 3072):
        n = 8 * 3 * 256 + 100
        idx, emb, rng = _build(P, n, dim, 777 + dim)
        ie_ref = _quantise_rows(emb)
        for B in (1, 5, 32, 33, 64, 65, 128, 129, 256, 300):
            qs = _queries(rng, B, dim, emb)
            cache = {}
            for nt in (False, True):
                _check(idx, qs, 0, nt, ie_ref, f"dim {dim}, {B} queries, eight-wave form, nt_rows {nt}", cache)
        if dim == 128:
            with pytest.raises(P.native.OrrError):                        # the four-wave forms need more K-tiles than their ring holds
                idx.screen_i8_dots(qs, 2)
        idx.close()
