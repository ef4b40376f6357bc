"""Row-sharded search across the GPUs of one node: one process per GPU, the corpus
split into contiguous ranges of the global candidate order.

Per step:  the queries reach every rank (all-gather of what each rank originates, or a
broadcast from the one rank that originates the batch)  ->  every rank scores ALL queries
against its own shard (orr_search_shard: the shard is read once for the whole batch)  ->
ONE all-gather of the per-shard candidate records over RCCL/xGMI  ->  every rank finishes
the queries on the host (orr_merge_candidates) from identical bytes, so the
escalate-or-not decision needs no further collective.  No all-reduce anywhere.

Query exchange format (length-prefixed, no per-query size limit): one block per rank,
    [4 x int64: queries, dim, term-section bytes, reserved]
    [queries x dim fp32]                                   the vectors
    [u32 query_term_off[queries+1]] [u32 term_off[T+1]] [term bytes]   the ABI's packed terms
The first collective carries the header, the vectors and the first TERM_BUDGET bytes per query
of the term section; only when some rank's term section is longer does a second collective
carry the remainders (every rank sees every header, so all take the same branch).
"""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch
import torch.distributed as dist

from .index import CAND_DTYPE, PackedTerms, merge_candidates, pack_terms

TERM_BUDGET = 256      # bytes of packed terms per query that ride in the first collective
HEADER_BYTES = 32


def _term_section(packed) -> np.ndarray:
    """ABI term arrays -> one byte string [qoff u32 x (B+1)][toff u32 x (T+1)][pool]."""
    pool, toff, qoff = packed
    qoff = np.ascontiguousarray(qoff, dtype=np.uint32)
    t_lo, t_hi = int(qoff[0]), int(qoff[-1])
    toff = np.ascontiguousarray(toff[t_lo:t_hi + 1], dtype=np.uint32)
    p_lo, p_hi = (int(toff[0]), int(toff[-1])) if toff.size else (0, 0)
    parts = [(qoff - np.uint32(t_lo)).view(np.uint8), (toff - np.uint32(p_lo)).view(np.uint8),
             np.ascontiguousarray(pool[p_lo:p_hi], dtype=np.uint8)]
    return np.concatenate(parts)


def _parse_term_section(sec: np.ndarray, n_queries: int):
    """The inverse: (pool, toff, qoff) of one rank."""
    qoff = sec[:4 * (n_queries + 1)].view(np.uint32)
    n_terms = int(qoff[-1])
    at = 4 * (n_queries + 1)
    toff = sec[at:at + 4 * (n_terms + 1)].view(np.uint32)
    at += 4 * (n_terms + 1)
    pool = sec[at:at + int(toff[-1])]
    return pool, toff, qoff


def _concat_packed(parts) -> PackedTerms:
    """Packed term arrays of several ranks -> those of the concatenated batch."""
    pools, toffs, qoffs = [], [np.zeros(1, dtype=np.int64)], [np.zeros(1, dtype=np.int64)]
    p_base = t_base = 0
    for pool, toff, qoff in parts:
        pools.append(pool)
        toffs.append(toff[1:].astype(np.int64) + p_base)
        qoffs.append(qoff[1:].astype(np.int64) + t_base)
        p_base += int(toff[-1])
        t_base += int(qoff[-1])
    pools.append(np.zeros(1, dtype=np.uint8))              # the ABI never reads it; keeps the pointer valid for empty pools
    return PackedTerms((np.ascontiguousarray(np.concatenate(pools)), np.concatenate(toffs).astype(np.uint32),
                        np.concatenate(qoffs).astype(np.uint32)))


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
        # shard_search(qvecs [B,dim] np/torch, terms, now, kprime, limit, out=, mode=, topk=) -> records [B,kprime+1]
        self._shard_search = shard_search or self._index_shard_search
        self._pinned = {}                      # name -> pinned host staging tensor (device runs only)
        self.collectives = 0                   # data-path collectives issued so far (diagnostic)
        self.escalations = 0                   # passes repeated with the exact pass or a larger k' (for the uncertified queries only)
        self.escalated_queries = 0             # queries that went through such a repeat (summed over repeats)
        self._shard_search_done = None         # test hook: called with (records tensor, queries, k') between the shard search and the all-gather
        self.always_collect = False            # rehearsals: issue the collectives with ONE rank too (RCCL itself on a one-GPU box)

    def _index_shard_search(self, q, terms, now, kprime, limit, out=None, mode=0, topk=0):
        # mode 0: the library picks the pass; 2: the exact pass (escalation after a failed certificate).
        # topk: the caller's k, so that the two-stage floor comes from the k-th best of the sample, not the k'-th.
        # Both travel as CALL ARGUMENTS (orr_search_shard_ex): nothing sticky is left on the index for another caller.
        return self.index.search_shard(q, terms, now, kprime, limit, out=out, topk=max(0, int(topk)), shard_pass=int(mode))

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

    def rccl_ranks_seen(self) -> int:
        """Ranks that answer one all-gather on this group's backend (bench evidence that the RCCL path ran)."""
        if self.world == 1 and not self.always_collect:
            return 1
        mine = torch.tensor([self.rank], dtype=torch.int64, device=self.device)
        allr = torch.empty(self.world, dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(allr, mine, group=self.group)
        return int(torch.unique(allr.cpu()).numel())

    # ---- exchange 1: the queries
    def _block(self, q_local, packed, B_local: int, dim: int, first_bytes: int):
        sec = _term_section(packed)
        vec_bytes = 4 * dim * B_local
        head = np.array([B_local, dim, sec.size, 0], dtype=np.int64).view(np.uint8)
        total = HEADER_BYTES + vec_bytes + sec.size
        block = torch.zeros(max(first_bytes, total), dtype=torch.uint8, device=self.device)
        block[:HEADER_BYTES] = torch.from_numpy(head.copy()).to(self.device, non_blocking=True)
        if vec_bytes:
            block[HEADER_BYTES:HEADER_BYTES + vec_bytes] = q_local.contiguous().view(torch.uint8).reshape(-1)
        if sec.size:
            block[HEADER_BYTES + vec_bytes:total] = torch.from_numpy(sec).to(self.device, non_blocking=True)
        return block, total

    def _gather_queries(self, q_local, terms_local, B_local: int, dim: int):
        """Every rank originates B_local queries.  Returns (q_all device [B,dim] or None, q_host, terms_all)."""
        W = self.world
        first = HEADER_BYTES + 4 * dim * B_local + TERM_BUDGET * B_local
        first = (first + 15) // 16 * 16
        block, total = self._block(q_local, pack_terms(terms_local), B_local, dim, first)
        if W > 1 or self.always_collect:
            got = torch.empty(W * first, dtype=torch.uint8, device=self.device)
            dist.all_gather_into_tensor(got, block[:first].contiguous(), group=self.group)
            got = got.reshape(W, first)
            self.collectives += 1
        else:
            got = block[:first].reshape(1, first)
        host = self._to_host("queries", got)
        heads = host[:, :HEADER_BYTES].copy().view(np.int64).reshape(W, 4)
        totals = HEADER_BYTES + 4 * heads[:, 1] * heads[:, 0] + heads[:, 2]
        rest_max = int(max(0, (totals - first).max()))
        rest_host = None
        if rest_max > 0:                                     # some rank's terms did not fit: one more collective, same on every rank
            rest_max = (rest_max + 15) // 16 * 16
            mine = torch.zeros(rest_max, dtype=torch.uint8, device=self.device)
            if total > first:
                mine[:total - first] = block[first:total]
            if W > 1 or self.always_collect:
                rest = torch.empty(W * rest_max, dtype=torch.uint8, device=self.device)
                dist.all_gather_into_tensor(rest, mine, group=self.group)
                rest = rest.reshape(W, rest_max)
                self.collectives += 1
            else:
                rest = mine.reshape(1, rest_max)
            rest_host = self._to_host("queries_rest", rest)
        parts, vecs = [], []
        for r in range(W):
            b_r, d_r, sec_bytes = int(heads[r, 0]), int(heads[r, 1]), int(heads[r, 2])
            if b_r != B_local or d_r != dim:
                raise ValueError("every rank must call search() with the same batch size and dimension")
            raw = host[r] if rest_host is None else np.concatenate([host[r], rest_host[r]])
            vb = 4 * dim * b_r
            if dim:
                vecs.append(raw[HEADER_BYTES:HEADER_BYTES + vb])
            parts.append(_parse_term_section(np.ascontiguousarray(raw[HEADER_BYTES + vb:HEADER_BYTES + vb + sec_bytes]), b_r))
        B = W * B_local
        q_host = np.ascontiguousarray(np.concatenate(vecs)).view(np.float32).reshape(B, dim) if dim else None
        if dim and self.device.type == "cuda":
            q_all = got[:, HEADER_BYTES:HEADER_BYTES + 4 * dim * B_local].contiguous().view(torch.float32).reshape(B, dim)
        else:
            q_all = torch.from_numpy(q_host) if dim else None
        return q_all, q_host, _concat_packed(parts)

    def _broadcast_queries(self, q, terms, origin: int):
        """Rank `origin` holds the whole batch (the others pass None): header, then one block."""
        W = self.world
        head = torch.zeros(4, dtype=torch.int64, device=self.device)
        block = None
        if self.rank == origin:
            B = len(terms)
            dim = int(q.shape[1]) if q is not None and q.numel() else 0
            block, total = self._block(q, pack_terms(terms), B, dim, 0)
            head = torch.tensor([B, dim, total, 0], dtype=torch.int64, device=self.device)
        if W > 1 or self.always_collect:
            dist.broadcast(head, src=origin, group=self.group)
            self.collectives += 1
        B, dim, total = (int(x) for x in head.cpu()[:3])
        if self.rank != origin:
            block = torch.empty(total, dtype=torch.uint8, device=self.device)
        if W > 1 or self.always_collect:
            dist.broadcast(block[:total], src=origin, group=self.group)
            self.collectives += 1
        host = self._to_host("queries", block[:total])
        vb = 4 * dim * B
        q_host = np.ascontiguousarray(host[HEADER_BYTES:HEADER_BYTES + vb]).view(np.float32).reshape(B, dim).copy() if dim else None
        sec_bytes = int(host[:HEADER_BYTES].copy().view(np.int64)[2])
        terms_all = _concat_packed([_parse_term_section(np.ascontiguousarray(host[HEADER_BYTES + vb:HEADER_BYTES + vb + sec_bytes]).copy(), B)])
        if dim and self.device.type == "cuda":
            q_all = block[HEADER_BYTES:HEADER_BYTES + vb].view(torch.float32).reshape(B, dim)
        else:
            q_all = torch.from_numpy(q_host) if dim else None
        return q_all, q_host, terms_all, B

    # ---- local scoring, exchange 2, host finish
    def _score_and_merge(self, q_all, q_host, terms_all, B: int, now_ticks: int, topk: int, candidate_limit: int, kprime: int):
        """Stream contract of the record exchange: orr_search_shard_ex writes this rank's records into `mine` on the library's
        OWN stream and returns only when they are complete (include/omnirecall_hip.h, conventions: device-resident outputs
        are complete when the call returns); the collective then reads `mine` on the process group's stream, which was idle
        or behind work issued before the call.  Nothing of torch's is pending on `mine` either: it is allocated here and
        never written by torch.  `_shard_search_done` (test hook) sees the buffer between the two."""
        W = self.world
        k = max(1, int(topk))
        rows = np.full((B, k), -1, dtype=np.int64)
        scores = np.zeros((B, k), dtype=np.float64)
        counts = np.zeros(B, dtype=np.int32)
        ids = np.arange(B)                                   # queries still to be answered, in the batch's numbering
        whole = True
        mode = 0
        while True:
            nb = int(ids.shape[0])
            if whole:
                q_sub, qh_sub, t_sub = q_all, q_host, terms_all
            else:
                # ONLY the queries that could not be certified go through the next, more exact pass, as a compacted sub-batch
                # (the exact pass keeps 8 bytes per (query,row): 1024 queries x 12.5M rows would be 102 GB).  Every rank holds
                # the same gathered bytes, hence the same `ids`: no collective is needed to agree on the sub-batch.
                qh_sub = None if q_host is None else np.ascontiguousarray(q_host[ids])
                if q_all is None:
                    q_sub = None
                elif q_all.device.type == "cuda":
                    q_sub = q_all.index_select(0, torch.as_tensor(ids, dtype=torch.int64, device=q_all.device)).contiguous()
                else:
                    q_sub = torch.from_numpy(qh_sub)
                t_sub = PackedTerms(pack_terms([terms_all[int(b)] for b in ids]))
            rec_bytes = nb * (kprime + 1) * CAND_DTYPE.itemsize
            mine = torch.empty(rec_bytes, dtype=torch.uint8, device=self.device)
            self._shard_search(q_sub, t_sub, now_ticks, kprime, candidate_limit, out=mine, mode=mode, topk=k)
            if self._shard_search_done is not None:
                self._shard_search_done(mine, nb, kprime)
            if W > 1 or self.always_collect:
                allrec = torch.empty(W * rec_bytes, dtype=torch.uint8, device=self.device)
                dist.all_gather_into_tensor(allrec, mine, group=self.group)
                self.collectives += 1
            else:
                allrec = mine
            recs = self._to_host("records", allrec).view(CAND_DTYPE).reshape(W, nb, kprime + 1)
            # Every rank finishes EVERY query from the same gathered bytes, so all ranks reach the
            # same "escalate or not" decision without another collective.
            r, s, c, unc, cert = merge_candidates(recs, self.index_dim, qh_sub, t_sub, now_ticks, topk, with_certificates=True)
            total = int(recs[:, 0, kprime]["order_key"].sum())
            done = cert if (unc > 0 and kprime < total) else np.ones(nb, dtype=bool)
            rows[ids[done]], scores[ids[done]], counts[ids[done]] = r[done], s[done], c[done]
            if done.all():
                return rows, scores, counts
            self.escalations += 1
            self.escalated_queries += int((~done).sum())
            ids, whole = ids[~done], False
            if mode == 0:
                mode = 2                                     # first the exact pass at the same k' (ties at the cut, overflowing survivor buffers)
            else:
                kprime = min(max(total, 1), kprime * 4)      # then a wider k' (masses of exact ties)

    def search(self, q_local: torch.Tensor, terms_local: Sequence[Sequence[bytes]], now_ticks: int, topk: int,
               candidate_limit: int, kprime: int = 32):
        """q_local: [B_local, dim] float32 on self.device (dim may be 0).  Every rank must call with
        the same B_local, dim, now_ticks, topk, candidate_limit.  Returns this rank's
        (rows [B_local,k], scores [B_local,k], counts [B_local])."""
        B_local = len(terms_local)
        dim = int(q_local.shape[1]) if q_local is not None and q_local.numel() else 0
        q_all, q_host, terms_all = self._gather_queries(q_local, terms_local, B_local, dim)
        rows, scores, counts = self._score_and_merge(q_all, q_host, terms_all, self.world * B_local, now_ticks, topk,
                                                     candidate_limit, kprime)
        lo = self.rank * B_local
        return rows[lo:lo + B_local], scores[lo:lo + B_local], counts[lo:lo + B_local]

    def search_from(self, origin: int, q, terms, now_ticks: int, topk: int, candidate_limit: int, kprime: int = 32):
        """One rank originates the whole batch (q [B, dim] on its device, terms; the others pass None, None):
        broadcast, score on every shard, one all-gather of records, finish.  Every rank returns the results of
        all B queries (rows [B,k], scores [B,k], counts [B])."""
        q_all, q_host, terms_all, B = self._broadcast_queries(q, terms, origin)
        return self._score_and_merge(q_all, q_host, terms_all, B, now_ticks, topk, candidate_limit, kprime)
