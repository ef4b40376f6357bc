"""RecallIndex: a thin Python handle over the C ABI (include/omnirecall_hip.h),
used by the tests, bench.py and the sharded front-end.  It adds no arithmetic:
every score comes out of libomnirecall_hip.so."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _native as N

CAND_DTYPE = np.dtype([("approx_score", "<f8"), ("dot", "<f8"), ("norm_b", "<f8"), ("created_ticks", "<i8"),
                       ("row_id", "<i8"), ("order_key", "<i8"), ("matches", "<i4"), ("flags", "<i4")])
assert CAND_DTYPE.itemsize == C.sizeof(N.OrrCandidate) == 56


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _ptr(x) -> Optional[int]:
    """Raw address of a numpy array or torch tensor (host or device).  The library reads device memory on its own
    streams (include/omnirecall_hip.h, conventions): whatever torch still has queued for a CUDA tensor is waited for."""
    if x is None:
        return None
    if _is_torch(x):
        assert x.is_contiguous()
        if x.is_cuda:
            import torch
            torch.cuda.current_stream(x.device).synchronize()
        return x.data_ptr()
    assert x.flags["C_CONTIGUOUS"]
    return x.ctypes.data


class PackedTerms:
    """(terms_utf8, term_off, query_term_off) already in the ABI's form; accepted wherever a list of
    per-query term lists is (saves re-packing the same batch for several calls).  Indexing gives one
    query's terms as a list of bytes, like the list form."""

    def __init__(self, arrays):
        self.arrays = tuple(arrays)

    def __len__(self):
        return int(self.arrays[2].shape[0]) - 1

    def __getitem__(self, b: int):
        pool, toff, qoff = self.arrays
        return [bytes(pool[int(toff[i]):int(toff[i + 1])]) for i in range(int(qoff[b]), int(qoff[b + 1]))]

    def __iter__(self):
        return (self[b] for b in range(len(self)))


def pack_terms(queries_terms) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """[[term bytes]] per query -> (terms_utf8, term_off, query_term_off) of the ABI."""
    if isinstance(queries_terms, PackedTerms):
        return queries_terms.arrays
    flat = [bytes(t) for terms in queries_terms for t in terms]
    pool_arr = np.frombuffer(b"".join(flat) + b"\0", dtype=np.uint8).copy()
    term_off = np.zeros(len(flat) + 1, dtype=np.uint32)
    if flat:
        np.cumsum(np.fromiter(map(len, flat), dtype=np.int64, count=len(flat)), out=term_off[1:])
    qoff = np.zeros(len(queries_terms) + 1, dtype=np.uint32)
    if len(queries_terms):
        np.cumsum(np.fromiter(map(len, queries_terms), dtype=np.int64, count=len(queries_terms)), out=qoff[1:])
    return pool_arr, term_off, qoff


def pack_contents(contents: Sequence[bytes]) -> Tuple[np.ndarray, np.ndarray]:
    off = np.zeros(len(contents) + 1, dtype=np.uint64)
    if len(contents):
        off[1:] = np.cumsum([len(c) for c in contents], dtype=np.uint64)
    pool = np.frombuffer(b"".join(bytes(c) for c in contents) + b"\0", dtype=np.uint8).copy()
    return pool, off


class RecallIndex:
    """One corpus shard resident on one GPU (orr_index)."""

    def __init__(self, dim: int, device: int = 0, capacity_rows: int = 0, row_base: int = 0):
        cfg = N.OrrConfig(C.sizeof(N.OrrConfig), device, dim, 0, capacity_rows, row_base)
        h = C.c_void_p()
        N.check(N.hip.orr_index_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self.dim = dim
        self.row_base = row_base

    def close(self) -> None:
        if getattr(self, "_h", None):
            N.hip.orr_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def rows(self) -> int:
        return int(N.hip.orr_index_rows(self._h))

    def append(self, emb, created_ticks, content_lower, content_off=None, row_ids=None) -> None:
        """emb: [n, dim] float32 (numpy or torch, host or device) or None for rows without an
        embedding; created_ticks: [n] int64; content_lower: list of lowercased bytes, or a uint8
        pool with content_off [n+1] uint64 (numpy or torch)."""
        if content_off is None:
            content_lower, content_off = pack_contents(content_lower)
        n = int(content_off.shape[0]) - 1
        if not _is_torch(created_ticks):
            created_ticks = np.ascontiguousarray(created_ticks, dtype=np.int64)
        if emb is not None and not _is_torch(emb):
            emb = np.ascontiguousarray(emb, dtype=np.float32)
        if row_ids is not None and not _is_torch(row_ids):
            row_ids = np.ascontiguousarray(row_ids, dtype=np.int64)
        dim = 0 if emb is None else int(emb.shape[1])
        N.check(N.hip.orr_index_append(self._h, n, dim, _ptr(emb), _ptr(created_ticks), _ptr(content_lower),
                                       _ptr(content_off), _ptr(row_ids)))

    def seal(self) -> None:
        N.check(N.hip.orr_index_seal(self._h))

    def delete_rows(self, row_ids) -> int:
        """orr_index_delete_rows: the rows with these ids stop taking part in later searches (no reseal).
        Returns how many rows were newly deleted."""
        ids = np.ascontiguousarray(row_ids, dtype=np.int64).reshape(-1)
        done = C.c_int64(0)
        N.check(N.hip.orr_index_delete_rows(self._h, int(ids.shape[0]), _ptr(ids), C.cast(C.byref(done), C.c_void_p)))
        return int(done.value)

    @property
    def live_rows(self) -> int:
        return int(N.hip.orr_index_live_rows(self._h))

    def compact(self) -> int:
        """orr_index_compact: the shard rebuilt in place without its deleted rows.  Returns the rows removed."""
        done = C.c_int64(0)
        N.check(N.hip.orr_index_compact(self._h, C.cast(C.byref(done), C.c_void_p)))
        return int(done.value)

    def save(self, path: str) -> None:
        """orr_index_save: the sealed shard as one binary file."""
        N.check(N.hip.orr_index_save(self._h, path.encode()))

    @classmethod
    def load(cls, path: str, device: int = 0, row_base: int = 0) -> "RecallIndex":
        """orr_index_load: a sealed shard from a file written by save()."""
        cfg = N.OrrConfig(C.sizeof(N.OrrConfig), device, 0, 0, 0, row_base)
        h = C.c_void_p()
        N.check(N.hip.orr_index_load(C.byref(cfg), path.encode(), C.byref(h)))
        self = cls.__new__(cls)
        self._h = h
        self.dim = int(N.hip.orr_index_dim(h))
        self.row_base = row_base
        return self

    @staticmethod
    def _query_args(qvecs, n_queries: int):
        if qvecs is None:
            return 0, None, None
        if not _is_torch(qvecs):
            qvecs = np.ascontiguousarray(qvecs, dtype=np.float32).reshape(n_queries, -1)
        dim = int(qvecs.shape[1])
        return dim, (qvecs if dim > 0 else None), qvecs

    def search(self, qvecs, queries_terms: Sequence[Sequence[bytes]], now_ticks: int, topk: int,
               candidate_limit: int = 300):
        """orr_search_batch.  Returns (rows [B,k] int64, scores [B,k] float64, counts [B] int32)."""
        B = len(queries_terms)
        dim, q, _keep = self._query_args(qvecs, B)
        pool, toff, qoff = pack_terms(queries_terms)
        k = max(1, int(topk))
        rows = np.full((B, k), -1, dtype=np.int64)
        scores = np.zeros((B, k), dtype=np.float64)
        counts = np.zeros(B, dtype=np.int32)
        N.check(N.hip.orr_search_batch(self._h, B, dim, _ptr(q), _ptr(pool), _ptr(toff), _ptr(qoff), now_ticks,
                                       int(topk), int(candidate_limit), _ptr(rows), _ptr(scores), _ptr(counts)))
        return rows, scores, counts

    def search_shard(self, qvecs, queries_terms, now_ticks: int, kprime: int, candidate_limit: int, out=None,
                     topk: Optional[int] = None, shard_pass: Optional[int] = None):
        """orr_search_shard (or, with topk / shard_pass given, orr_search_shard_ex: the caller's k and the pass as call
        arguments instead of sticky index options).  Returns a [B, kprime+1] structured array (or fills `out`, which may be
        a torch uint8 tensor on the device with B*(kprime+1)*56 bytes)."""
        B = len(queries_terms)
        dim, q, _keep = self._query_args(qvecs, B)
        pool, toff, qoff = pack_terms(queries_terms)
        if out is None:
            out = np.zeros((B, kprime + 1), dtype=CAND_DTYPE)
        if topk is None and shard_pass is None:
            N.check(N.hip.orr_search_shard(self._h, B, dim, _ptr(q), _ptr(pool), _ptr(toff), _ptr(qoff), now_ticks,
                                           int(kprime), int(candidate_limit), _ptr(out)))
        else:
            N.check(N.hip.orr_search_shard_ex(self._h, B, dim, _ptr(q), _ptr(pool), _ptr(toff), _ptr(qoff), now_ticks,
                                              int(kprime), int(candidate_limit), max(0, int(topk or 0)), int(shard_pass or 0), _ptr(out)))
        return out

    def view(self) -> "RecallIndex":
        """orr_index_view: a second search lane over this sealed shard (own streams and workspaces, shared
        corpus).  Searches on the index and on its views may run concurrently from different threads."""
        h = C.c_void_p()
        N.check(N.hip.orr_index_view(self._h, C.byref(h)))
        v = RecallIndex.__new__(RecallIndex)
        v.__dict__.update(self.__dict__)
        v._h = h
        v._parent = self                     # keeps the owner alive; close views before it
        return v

    def set_option(self, name: str, value: int) -> None:
        N.check(N.hip.orr_index_set_option(self._h, name.encode(), int(value)))

    def screen_dots(self, qvecs) -> np.ndarray:
        """orr_index_screen_dots: the two-stage pass's plain-bf16 screening dots, [B, rows] fp32 (diagnostic)."""
        q = np.ascontiguousarray(qvecs, dtype=np.float32)
        q = q.reshape(1, -1) if q.ndim == 1 else q
        out = np.empty((q.shape[0], self.rows), dtype=np.float32)
        N.check(N.hip.orr_index_screen_dots(self._h, int(q.shape[0]), int(q.shape[1]), _ptr(q), _ptr(out)))
        return out

    def screen_i8_dots(self, qvecs, form: int, nt_rows: bool = False, images: bool = True):
        """orr_index_screen_i8_dots: raw int32 accumulators [B, rows] of one form of the int8 screening GEMM (0 eight-wave,
        1 four-wave 32x32x32, 2 four-wave 16x16x64) and, with images, the int8 images it multiplied ([B, dim], [rows, dim])."""
        q = np.ascontiguousarray(qvecs, dtype=np.float32)
        q = q.reshape(1, -1) if q.ndim == 1 else q
        B, dim, n = int(q.shape[0]), int(q.shape[1]), self.rows
        dots = np.empty((B, n), dtype=np.int32)
        iq = np.empty((B, dim), dtype=np.int8) if images else None
        ie = np.empty((n, dim), dtype=np.int8) if images else None
        N.check(N.hip.orr_index_screen_i8_dots(self._h, B, dim, _ptr(q), int(form), 1 if nt_rows else 0, _ptr(dots), _ptr(iq), _ptr(ie)))
        return dots, iq, ie

    def set_profiling(self, on) -> None:
        """False/0 off, True/1 every kernel, 2 only the launch that streams every row (orr_index_set_profiling)."""
        N.check(N.hip.orr_index_set_profiling(self._h, int(on)))

    def search_stats(self, reset: bool = False) -> dict:
        """orr_index_search_stats: repeats of the cheap passes and what the screening pass kept, since the last reset."""
        st = N.OrrSearchStats()
        N.check(N.hip.orr_index_search_stats(self._h, C.byref(st), 1 if reset else 0))
        d = {n: int(getattr(st, n)) for n, _ in N.OrrSearchStats._fields_ if n != "reserved"}
        d["survivors_per_query"] = d["survivors_total"] / d["survivor_samples"] if d["survivor_samples"] else None
        return d

    def reset_search_stats(self) -> None:
        N.check(N.hip.orr_index_search_stats(self._h, None, 1))

    def kernel_stats(self) -> dict:
        arr = (N.OrrKernelStat * 32)()
        n = N.hip.orr_index_kernel_stats(self._h, C.cast(arr, C.c_void_p), 32)
        return {arr[i].name.decode(): {"launches": int(arr[i].launches), "total_ms": float(arr[i].total_ms),
                                       "algo_bytes": float(arr[i].algo_bytes)} for i in range(min(n, 32))}


class _BorrowedIndex(RecallIndex):
    """A shard owned by a RecallCluster: usable like a RecallIndex, never destroyed through this object."""

    def __init__(self, handle, dim: int, owner):
        self._h = handle
        self.dim = dim
        self.row_base = 0
        self._owner = owner                  # keeps the cluster alive

    def close(self) -> None:
        self._h = None


class RecallCluster:
    """orr_cluster: several shards (one per entry of `devices`) behind one handle in this process; searches answer as
    one index over all the rows would.  shard(i) is a borrowed RecallIndex to append to (rows of shard i newer than or
    as new as those of shard i + 1; pass explicit row_ids)."""

    def __init__(self, devices: Sequence[int], dim: int, capacity_rows_per_shard: int = 0):
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        h = C.c_void_p()
        N.check(N.hip.orr_cluster_create(_ptr(devs), int(devs.shape[0]), dim, int(capacity_rows_per_shard), C.byref(h)))
        self._h = h
        self.dim = dim

    def close(self) -> None:
        if getattr(self, "_h", None):
            N.hip.orr_cluster_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_shards(self) -> int:
        return int(N.hip.orr_cluster_shards(self._h))

    @property
    def rows(self) -> int:
        return int(N.hip.orr_cluster_rows(self._h))

    def shard(self, i: int) -> RecallIndex:
        p = N.hip.orr_cluster_shard(self._h, int(i))
        if not p:
            N.check(N.ORR_EINVAL)
        return _BorrowedIndex(C.c_void_p(p), self.dim, self)

    def seal(self) -> None:
        N.check(N.hip.orr_cluster_seal(self._h))

    def search(self, qvecs, queries_terms, now_ticks: int, topk: int, candidate_limit: int = 300):
        """orr_cluster_search_batch; qvecs in host memory (numpy) or None."""
        B = len(queries_terms)
        if qvecs is None:
            dim, q = 0, None
        else:
            q = np.ascontiguousarray(qvecs, dtype=np.float32).reshape(B, -1)
            dim = int(q.shape[1])
            q = q if dim > 0 else None
        pool, toff, qoff = pack_terms(queries_terms)
        k = max(1, int(topk))
        rows = np.full((B, k), -1, dtype=np.int64)
        scores = np.zeros((B, k), dtype=np.float64)
        counts = np.zeros(B, dtype=np.int32)
        N.check(N.hip.orr_cluster_search_batch(self._h, B, dim, _ptr(q), _ptr(pool), _ptr(toff), _ptr(qoff), now_ticks, int(topk),
                                               int(candidate_limit), _ptr(rows), _ptr(scores), _ptr(counts)))
        return rows, scores, counts

    def set_option(self, name: str, value: int) -> None:
        """orr_cluster_set_option ("exchange": 0 pinned host memory, 1 RCCL all-gather)."""
        N.check(N.hip.orr_cluster_set_option(self._h, name.encode(), int(value)))

    def compact(self) -> int:
        """orr_cluster_compact: every shard without its deleted rows, placed in the global order again."""
        done = C.c_int64(0)
        N.check(N.hip.orr_cluster_compact(self._h, C.cast(C.byref(done), C.c_void_p)))
        return int(done.value)

    def search_stats(self, reset: bool = False) -> dict:
        st = N.OrrSearchStats()
        N.check(N.hip.orr_cluster_search_stats(self._h, C.byref(st), 1 if reset else 0))
        d = {n: int(getattr(st, n)) for n, _ in N.OrrSearchStats._fields_ if n != "reserved"}
        d["survivors_per_query"] = d["survivors_total"] / d["survivor_samples"] if d["survivor_samples"] else None
        d["rccl_exchanges"] = int(st.reserved[0])
        return d


def merge_candidates(all_records: np.ndarray, index_dim: int, qvecs, queries_terms, now_ticks: int, topk: int, with_certificates: bool = False):
    """orr_merge_candidates over [n_shards, B, kprime+1] records (host).  Returns
    (rows, scores, counts, uncertified) -- with_certificates: (..., uncertified, certified [B] bool) through
    orr_merge_candidates_ex."""
    assert all_records.dtype == CAND_DTYPE and all_records.ndim == 3
    n_shards, B, kp1 = all_records.shape
    all_records = np.ascontiguousarray(all_records)
    if qvecs is None:
        dim, q = 0, None
    else:
        q = np.ascontiguousarray(qvecs, dtype=np.float32).reshape(B, -1)
        dim = int(q.shape[1])
        q = q if dim > 0 else None
    _, _, qoff = pack_terms(queries_terms)
    k = max(1, int(topk))
    rows = np.full((B, k), -1, dtype=np.int64)
    scores = np.zeros((B, k), dtype=np.float64)
    counts = np.zeros(B, dtype=np.int32)
    unc = C.c_int32(0)
    if with_certificates:
        cert = np.zeros(B, dtype=np.uint8)
        N.check(N.hip.orr_merge_candidates_ex(n_shards, B, kp1 - 1, _ptr(all_records), index_dim, dim, _ptr(q), _ptr(qoff),
                                              now_ticks, int(topk), _ptr(rows), _ptr(scores), _ptr(counts),
                                              C.cast(C.byref(unc), C.c_void_p), _ptr(cert)))
        return rows, scores, counts, int(unc.value), cert.astype(bool)
    N.check(N.hip.orr_merge_candidates(n_shards, B, kp1 - 1, _ptr(all_records), index_dim, dim, _ptr(q), _ptr(qoff),
                                       now_ticks, int(topk), _ptr(rows), _ptr(scores), _ptr(counts),
                                       C.cast(C.byref(unc), C.c_void_p)))
    return rows, scores, counts, int(unc.value)


class MicroBatcher:
    """orrh_batcher: coalesces concurrent single-query searches on one RecallIndex into batched
    orr_search_batch calls (include/omnirecall_host.h).  search() blocks and is thread-safe
    (ctypes releases the GIL during the call)."""

    def __init__(self, index: RecallIndex, max_batch: int = 64, max_wait_us: int = 200):
        self._index = index                    # keep the index alive
        self._h = C.c_void_p(N.host.orrh_batcher_create(index._h, max_batch, max_wait_us))
        if not self._h:
            raise ValueError("orrh_batcher_create failed")

    def search(self, qvec, terms: Sequence[bytes], now_ticks: int, topk: int, candidate_limit: int = 300, with_clock: bool = False):
        """One request.  with_clock: also return the clock its batch was answered at (the latest now_ticks of the batch)."""
        q = None if qvec is None else np.ascontiguousarray(qvec, dtype=np.float32).reshape(-1)
        dim = 0 if q is None else int(q.shape[0])
        pool, toff, _ = pack_terms([terms])
        k = max(1, int(topk))
        rows = np.full(k, -1, dtype=np.int64)
        scores = np.zeros(k, dtype=np.float64)
        cnt, clock = C.c_int32(0), C.c_int64(0)
        st = N.host.orrh_batcher_search_at(self._h, dim, _ptr(q) if dim else None, _ptr(pool), _ptr(toff), len(terms),
                                           now_ticks, int(topk), int(candidate_limit), _ptr(rows), _ptr(scores),
                                           C.cast(C.byref(cnt), C.c_void_p), C.cast(C.byref(clock), C.c_void_p))
        if st != N.ORR_OK:
            raise N.OrrError(st, (N.host.orrh_last_error() or b"").decode("utf-8", "replace"))
        if with_clock:
            return rows[:cnt.value], scores[:cnt.value], int(clock.value)
        return rows[:cnt.value], scores[:cnt.value]

    def stats(self):
        b, r, l = C.c_int64(0), C.c_int64(0), C.c_int32(0)
        N.host.orrh_batcher_stats(self._h, C.cast(C.byref(b), C.c_void_p), C.cast(C.byref(r), C.c_void_p),
                                  C.cast(C.byref(l), C.c_void_p))
        return {"batches": b.value, "requests": r.value, "largest_batch": l.value}

    def close(self):
        if getattr(self, "_h", None):
            N.host.orrh_batcher_destroy(self._h)
            self._h = None
