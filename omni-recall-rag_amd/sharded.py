"""Row-sharded search across the GPUs of one node: one process per GPU, the corpus
split into contiguous ranges of the global candidate order.

Per step:  all-gather the queries (every rank may originate some)  ->  every rank
scores ALL queries against its own shard (orr_search_shard, HBM-bound: the shard is
read once for the whole batch)  ->  ONE all-gather of the per-shard candidate
records over RCCL/xGMI  ->  every rank finishes the queries on the host
(orr_merge_candidates) and keeps the ones it originated.  No all-reduce anywhere.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch
import torch.distributed as dist

from .index import CAND_DTYPE, PackedTerms, RecallIndex, merge_candidates, pack_terms

TERM_SLOT = 256        # bytes reserved per query for its packed terms


def _slots_from_packed(pool: np.ndarray, toff: np.ndarray, qoff: np.ndarray) -> np.ndarray:
    """ABI term arrays -> one TERM_SLOT-byte slot per query: [n][len_0 .. len_{n-1}][bytes...], zero padded.
    Vectorised: a batch of 1024 queries costs a fraction of a millisecond; a handful of queries (the
    one-query-per-rank step of the bench) is quicker in plain Python than through a dozen numpy calls."""
    B = int(qoff.shape[0]) - 1
    slots = np.zeros((B, TERM_SLOT), dtype=np.uint8)
    if B == 0:
        return slots
    if B <= 4:
        to, qo = toff.tolist(), qoff.tolist()
        raw = pool.tobytes()
        for b in range(B):
            t0, t1 = qo[b], qo[b + 1]
            lens = [to[t + 1] - to[t] for t in range(t0, t1)]
            if t1 - t0 > 255 or (lens and max(lens) > 255):
                raise ValueError("term longer than 255 bytes" if lens and max(lens) > 255 else
                                 "query terms do not fit the %d-byte exchange slot" % TERM_SLOT)
            body = bytes([t1 - t0]) + bytes(lens) + raw[to[t0]:to[t1]]
            if len(body) > TERM_SLOT:
                raise ValueError("query terms do not fit the %d-byte exchange slot" % TERM_SLOT)
            slots[b, :len(body)] = np.frombuffer(body, dtype=np.uint8)
        return slots
    toff = toff.astype(np.int64)
    qoff = qoff.astype(np.int64)
    lens = np.diff(toff)                                    # per term
    n = np.diff(qoff)                                       # terms per query
    qbytes = toff[qoff[1:]] - toff[qoff[:-1]]               # term bytes per query
    if lens.size and int(lens.max()) > 255:
        raise ValueError("term longer than 255 bytes")
    if int(n.max()) > 255 or int((1 + n + qbytes).max()) > TERM_SLOT:
        raise ValueError("query terms do not fit the %d-byte exchange slot" % TERM_SLOT)
    slots[:, 0] = n
    tq = np.repeat(np.arange(B), n)                         # query of each term
    slots[tq, 1 + np.arange(lens.size) - qoff[tq]] = lens
    total = int(toff[-1])
    bq = np.repeat(np.arange(B), qbytes)                    # query of each term byte
    slots[bq, 1 + n[bq] + np.arange(total) - toff[qoff[bq]]] = pool[:total]
    return slots


def _packed_from_slots(slots: np.ndarray) -> PackedTerms:
    """The inverse, for all gathered queries at once."""
    B = int(slots.shape[0])
    if B <= 16:                                             # a few queries: plain Python beats the masked gathers below
        pool, toff, qoff = bytearray(), [0], [0]
        for row in slots:
            raw = row.tobytes()
            n = raw[0]
            at = 1 + n
            for ln in raw[1:1 + n]:
                pool += raw[at:at + ln]
                at += ln
                toff.append(len(pool))
            qoff.append(len(toff) - 1)
        pool.append(0)
        return PackedTerms((np.frombuffer(bytes(pool), dtype=np.uint8), np.asarray(toff, dtype=np.uint32),
                            np.asarray(qoff, dtype=np.uint32)))
    n = slots[:, 0].astype(np.int64)
    col = np.arange(TERM_SLOT, dtype=np.int64)[None, :]
    len_mask = (col >= 1) & (col < 1 + n[:, None])
    lens = slots[len_mask].astype(np.int64)                 # row-major: query order, then term order
    qoff = np.zeros(B + 1, dtype=np.int64)
    np.cumsum(n, out=qoff[1:])
    toff = np.zeros(lens.size + 1, dtype=np.int64)
    np.cumsum(lens, out=toff[1:])
    qbytes = toff[qoff[1:]] - toff[qoff[:-1]]
    byte_mask = (col >= 1 + n[:, None]) & (col < 1 + n[:, None] + qbytes[:, None])
    pool = np.concatenate([slots[byte_mask], np.zeros(1, dtype=np.uint8)])
    return PackedTerms((np.ascontiguousarray(pool), toff.astype(np.uint32), qoff.astype(np.uint32)))


def _pack_terms_fixed(terms: Sequence[bytes]) -> np.ndarray:
    """One query's slot (kept for tests and small callers)."""
    return _slots_from_packed(*pack_terms([terms]))[0]


def _unpack_terms_fixed(slot: np.ndarray) -> List[bytes]:
    return _packed_from_slots(slot[None, :])[0]


class ShardedRecallSearch:
    """`index` is this rank's shard (row_base = its first global row).  `device` is the
    torch device of this rank, or "cpu" for gloo rehearsals with a stub shard search."""

    def __init__(self, index, index_dim: int, device, group=None, shard_search=None):
        self.index = index
        self.index_dim = index_dim
        self.device = torch.device(device)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # shard_search(qvecs [B,dim] np/torch, terms, now, kprime, limit) -> records [B,kprime+1]
        self._shard_search = shard_search or (lambda q, t, now, kp, lim, out=None:
                                              self.index.search_shard(q, t, now, kp, lim, out=out))
        self._pinned = {}                      # name -> pinned host staging tensor (device runs only)

    def _to_host(self, name: str, t: torch.Tensor) -> np.ndarray:
        """Device tensor -> numpy through a reused pinned buffer (one asynchronous copy + one stream sync instead of
        a pageable synchronous copy); plain .numpy() on CPU rehearsals."""
        if t.device.type != "cuda":
            return t.numpy()
        buf = self._pinned.get(name)
        if buf is None or buf.numel() < t.numel() or buf.dtype != t.dtype:
            buf = torch.empty(max(t.numel(), 1), dtype=t.dtype, pin_memory=True)
            self._pinned[name] = buf
        view = buf[:t.numel()].view(t.shape)
        view.copy_(t, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        return view.numpy()

    def search(self, q_local: torch.Tensor, terms_local: Sequence[Sequence[bytes]], now_ticks: int, topk: int,
               candidate_limit: int, kprime: int = 32):
        """q_local: [B_local, dim] float32 on self.device (dim may be 0).  Every rank must call with
        the same B_local, dim, now_ticks, topk, candidate_limit.  Returns this rank's
        (rows [B_local,k], scores [B_local,k], counts [B_local])."""
        B_local = len(terms_local)
        dim = int(q_local.shape[1]) if q_local is not None and q_local.numel() else 0
        W = self.world
        # ---- exchange 1: queries (vector bytes + packed terms in one buffer per query)
        vec_bytes = 4 * dim
        slot = vec_bytes + TERM_SLOT
        send = torch.empty((B_local, slot), dtype=torch.uint8, device=self.device)
        if dim:
            send[:, :vec_bytes] = q_local.contiguous().view(torch.uint8).reshape(B_local, vec_bytes)
        tslots = _slots_from_packed(*pack_terms(terms_local))
        send[:, vec_bytes:] = torch.from_numpy(tslots).to(self.device, non_blocking=True)
        if W > 1:
            allq = torch.empty((W * B_local, slot), dtype=torch.uint8, device=self.device)
            dist.all_gather_into_tensor(allq, send, group=self.group)
        else:
            allq = send
        B = W * B_local
        q_all = allq[:, :vec_bytes].contiguous().view(torch.float32).reshape(B, dim) if dim else None
        allq_host = self._to_host("queries", allq)   # ONE download: vectors (for the exact normA) + terms
        q_host = np.ascontiguousarray(allq_host[:, :vec_bytes]).view(np.float32).reshape(B, dim) if dim else None
        terms_all = _packed_from_slots(allq_host[:, vec_bytes:])   # ABI form, packed once for both calls below

        while True:
            # ---- local scoring of every query against this shard
            rec_bytes = B * (kprime + 1) * CAND_DTYPE.itemsize
            mine = torch.empty(rec_bytes, dtype=torch.uint8, device=self.device)
            self._shard_search(q_all, terms_all, now_ticks, kprime, candidate_limit, out=mine)
            # ---- exchange 2: ONE all-gather of the per-shard top-k' records
            if W > 1:
                allrec = torch.empty(W * rec_bytes, dtype=torch.uint8, device=self.device)
                dist.all_gather_into_tensor(allrec, mine, group=self.group)
            else:
                allrec = mine
            recs = self._to_host("records", allrec).view(CAND_DTYPE).reshape(W, B, kprime + 1)
            # Every rank finishes EVERY query from the same gathered bytes, so all ranks reach the
            # same "escalate or not" decision without another collective.
            rows, scores, counts, unc = merge_candidates(recs, self.index_dim, q_host, terms_all, now_ticks, topk)
            lo = self.rank * B_local
            rows, scores, counts = rows[lo:lo + B_local], scores[lo:lo + B_local], counts[lo:lo + B_local]
            total = int(recs[:, 0, kprime]["order_key"].sum())
            if unc == 0 or kprime >= total:
                return rows, scores, counts
            kprime = min(max(total, 1), kprime * 4)
